// gemm.hip -- instantiations + launcher of the grouped fp32-MFMA GEMM (see gemm.h).
#include <cstdlib>
#define GEMM_RC_MULTI_IMPL
#include "gemm_wide.h"

template <int NI>
static void launch_ni(const GemmP& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    if (a_kc && b_kc) hipLaunchKernelGGL((k_gemm<NI, true, true>), grid, dim3(GEMM_THREADS), 0, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((k_gemm<NI, true, false>), grid, dim3(GEMM_THREADS), 0, st, p);
    else if (!a_kc && !b_kc) hipLaunchKernelGGL((k_gemm<NI, false, false>), grid, dim3(GEMM_THREADS), 0, st, p);
    else hipLaunchKernelGGL((k_gemm<NI, false, true>), grid, dim3(GEMM_THREADS), 0, st, p);
}

int launch_gemm(const GemmP& p_in, bool a_kc, bool b_kc, hipStream_t st) {
    GemmP p = p_in;
    AR_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.G > 0, "gemm: empty problem M=%d N=%d K=%d G=%d", p.M, p.N, p.K, p.G);
    AR_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.a_gs % 4 == 0 && p.b_gs % 4 == 0,
                 "gemm: leading dimensions / group strides must be multiples of 4 (lda=%lld ldb=%lld)",
                 (long long)p.lda, (long long)p.ldb);
    AR_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "gemm: operands must be 16-byte aligned");
    if (p.k_split < 1) p.k_split = 1;
    if (p.k_split == 1) p.k_chunk = p.K;
    AR_CHECK_ARG(p.k_split == 1 || p.k_chunk % TILE_M == 0, "gemm: k_chunk must be a multiple of %d", TILE_M);
    AR_CHECK_ARG(p.gate_axis == 0 || p.tile_seg != nullptr, "gemm: gating needs tile_seg");
    const int ni = gemm_ni(p.N);
    dim3 grid(cdiv(p.N, 16 * ni), cdiv(p.M, 64), p.G * p.k_split);
    switch (ni) {
        case 8: launch_ni<8>(p, a_kc, b_kc, grid, st); break;
        case 6: launch_ni<6>(p, a_kc, b_kc, grid, st); break;
        case 4: launch_ni<4>(p, a_kc, b_kc, grid, st); break;
        case 2: launch_ni<2>(p, a_kc, b_kc, grid, st); break;
        default: launch_ni<1>(p, a_kc, b_kc, grid, st); break;
    }
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

int launch_gemm_bf3(const GemmP& p_in, hipStream_t st) {
    GemmP p = p_in;
    AR_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.G > 0, "gemm_bf3: empty problem");
    AR_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.a_gs % 4 == 0 && p.b_gs % 4 == 0, "gemm_bf3: strides must be multiples of 4");
    AR_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "gemm_bf3: operands must be 16-byte aligned");
    AR_CHECK_ARG(p.gate_axis != 2, "gemm_bf3: k-gating is not supported");
    AR_CHECK_ARG(p.gate_axis == 0 || p.tile_seg != nullptr, "gemm_bf3: gating needs tile_seg");
    p.k_split = 1; p.k_chunk = p.K; p.c_ks = 0;
    const int ni = gemm_ni(p.N);
    dim3 grid(cdiv(p.N, 16 * ni), cdiv(p.M, 64), p.G);
    switch (ni) {
        case 8: hipLaunchKernelGGL((k_gemm_bf3<8>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 6: hipLaunchKernelGGL((k_gemm_bf3<6>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 4: hipLaunchKernelGGL((k_gemm_bf3<4>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 2: hipLaunchKernelGGL((k_gemm_bf3<2>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        default: hipLaunchKernelGGL((k_gemm_bf3<1>), grid, dim3(GEMM_THREADS), 0, st, p); break;
    }
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

int launch_gemm_bf3_rc(const GemmP& p_in, hipStream_t st) {
    GemmP p = p_in;
    AR_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.G > 0, "gemm_bf3_rc: empty problem");
    AR_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.a_gs % 4 == 0 && p.b_gs % 4 == 0, "gemm_bf3_rc: strides must be multiples of 4");
    AR_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "gemm_bf3_rc: operands must be 16-byte aligned");
    if (p.k_split < 1) p.k_split = 1;
    if (p.k_split == 1) p.k_chunk = p.K;
    AR_CHECK_ARG(p.k_split == 1 || p.k_chunk % TILE_M == 0, "gemm_bf3_rc: k_chunk must be a multiple of %d", TILE_M);
    AR_CHECK_ARG(p.gate_axis == 0 || p.gate_axis == 2, "gemm_bf3_rc: only k-tile gating is supported");
    AR_CHECK_ARG(p.gate_axis == 0 || p.tile_seg != nullptr, "gemm_bf3_rc: gating needs tile_seg");
    const int ni = gemm_ni(p.N);
    dim3 grid(cdiv(p.N, 16 * ni), cdiv(p.M, 64), p.G * p.k_split);
    switch (ni) {
        case 8: hipLaunchKernelGGL((k_gemm_bf3_rc<8>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 6: hipLaunchKernelGGL((k_gemm_bf3_rc<6>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 4: hipLaunchKernelGGL((k_gemm_bf3_rc<4>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        case 2: hipLaunchKernelGGL((k_gemm_bf3_rc<2>), grid, dim3(GEMM_THREADS), 0, st, p); break;
        default: hipLaunchKernelGGL((k_gemm_bf3_rc<1>), grid, dim3(GEMM_THREADS), 0, st, p); break;
    }
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

int launch_gemm_bf3_rc_multi(const GemmP* ps, int n, hipStream_t st) {
    if (n <= 0) return AREAD_OK;
    if (n == 1) return launch_gemm_bf3_rc(ps[0], st);
    for (int lo = 0; lo < n; lo += RC_MULTI_MAX) {
        const int cnt = n - lo < RC_MULTI_MAX ? n - lo : RC_MULTI_MAX;
        RcMultiP a = {};
        a.n = cnt;
        int blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            GemmP p = ps[lo + i];
            AR_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.G > 0, "gemm_bf3_rc_multi: empty problem");
            AR_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0 && p.a_gs % 4 == 0 && p.b_gs % 4 == 0, "gemm_bf3_rc_multi: strides must be multiples of 4");
            AR_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "gemm_bf3_rc_multi: operands must be 16-byte aligned");
            if (p.k_split < 1) p.k_split = 1;
            if (p.k_split == 1) p.k_chunk = p.K;
            AR_CHECK_ARG(p.k_split == 1 || p.k_chunk % TILE_M == 0, "gemm_bf3_rc_multi: k_chunk must be a multiple of %d", TILE_M);
            AR_CHECK_ARG(p.ldc == p.N && !p.accumulate && !p.bias && !p.stat_part, "gemm_bf3_rc_multi: plain split-K slabs only");
            AR_CHECK_ARG(p.gate_axis == ps[lo].gate_axis && p.tile_seg == ps[lo].tile_seg && p.active_ld == ps[lo].active_ld,
                         "gemm_bf3_rc_multi: the GEMMs of one launch share the row plan");
            AR_CHECK_ARG(p.gate_axis == 0 || (p.gate_axis == 2 && p.tile_seg != nullptr), "gemm_bf3_rc_multi: only k-tile gating is supported");
            RcDesc& d = a.d[i];
            d.A = p.A; d.lda = p.lda; d.a_gs = p.a_gs; d.B = p.B; d.ldb = p.ldb; d.b_gs = p.b_gs;
            d.C = p.C; d.c_gs = p.c_gs; d.c_ks = p.c_ks; d.active = p.active;
            d.M = p.M; d.N = p.N; d.K = p.K; d.G = p.G; d.k_split = p.k_split; d.k_chunk = p.k_chunk;
            d.ni = gemm_ni(p.N); d.nx = cdiv(p.N, 16 * d.ni); d.ny = cdiv(p.M, 64); d.nz = p.G * p.k_split;
            a.first[i] = blocks;
            blocks += (d.nx * d.ny * d.nz + 7) / 8 * 8;
        }
        a.first[cnt] = blocks;
        a.tile_seg = ps[lo].tile_seg; a.active_ld = ps[lo].active_ld; a.gate_axis = ps[lo].gate_axis;
        hipLaunchKernelGGL(k_gemm_bf3_rc_multi, dim3(blocks), dim3(GEMM_THREADS), 0, st, a);
        AR_LAUNCH_CHECK();
    }
    return AREAD_OK;
}

extern "C" int aread_gemm_bf16x3_rc(const float* A, int64_t lda, int64_t a_gs, const float* B, int64_t ldb, int64_t b_gs, float* C,
                                    int64_t ldc, int64_t c_gs, int M, int N, int K, int G, int accumulate, void* stream) {
    GemmP p = {};
    p.A = A; p.lda = lda; p.a_gs = a_gs;
    p.B = B; p.ldb = ldb; p.b_gs = b_gs;
    p.C = C; p.ldc = ldc; p.c_gs = c_gs;
    p.M = M; p.N = N; p.K = K; p.G = G;
    p.accumulate = accumulate;
    AR_CHECK_ARG(A && B && C, "aread_gemm_bf16x3_rc: null pointer");
    return launch_gemm_bf3_rc(p, (hipStream_t)stream);
}

extern "C" int aread_gemm(const float* A, int64_t lda, int64_t a_gs, int a_kc, const float* B, int64_t ldb,
                          int64_t b_gs, int b_kc, float* C, int64_t ldc, int64_t c_gs, const float* bias,
                          int64_t bias_gs, int M, int N, int K, int G, int accumulate, void* stream) {
    GemmP p = {};
    p.A = A; p.lda = lda; p.a_gs = a_gs;
    p.B = B; p.ldb = ldb; p.b_gs = b_gs;
    p.C = C; p.ldc = ldc; p.c_gs = c_gs; p.c_ks = 0;
    p.bias = bias; p.bias_gs = bias_gs;
    p.M = M; p.N = N; p.K = K; p.G = G;
    p.accumulate = accumulate;
    p.k_split = 1; p.k_chunk = K;
    AR_CHECK_ARG(A && B && C, "aread_gemm: null pointer");
    return launch_gemm(p, a_kc != 0, b_kc != 0, (hipStream_t)stream);
}

extern "C" int aread_gemm_bf16x3(const float* A, int64_t lda, int64_t a_gs, const float* B, int64_t ldb, int64_t b_gs, float* C,
                                 int64_t ldc, int64_t c_gs, const float* bias, int64_t bias_gs, int M, int N, int K, int G,
                                 int accumulate, void* stream) {
    GemmP p = {};
    p.A = A; p.lda = lda; p.a_gs = a_gs;
    p.B = B; p.ldb = ldb; p.b_gs = b_gs;
    p.C = C; p.ldc = ldc; p.c_gs = c_gs;
    p.bias = bias; p.bias_gs = bias_gs;
    p.M = M; p.N = N; p.K = K; p.G = G;
    p.accumulate = accumulate;
    AR_CHECK_ARG(A && B && C, "aread_gemm_bf16x3: null pointer");
    return launch_gemm_bf3(p, (hipStream_t)stream);
}

// ---- pre-tiled split-bf16 weight images (gemm_wide.h) --------------------------------------------------------------------
int launch_prep_wimg(const WPrepAllP& a, hipStream_t st) {
    if (a.n <= 0) return AREAD_OK;
    int mx = 1;
    for (int i = 0; i < a.n; ++i) {
        const int b = a.d[i].G * a.d[i].NT * a.d[i].KS;
        if (b > mx) mx = b;
    }
    if (mx > 128) mx = 128;
    hipLaunchKernelGGL(k_prep_wimg, dim3(mx, a.n), dim3(256), 0, st, a);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
