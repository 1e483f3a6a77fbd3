// gemm_wide.h -- split-bf16 ("bf16x3") GEMM with 128-row tiles for the expert / tower Linear layers
// (forward Y = X W^T + b and dgrad dX = dY W of model/layer.py:203-229), the dense contraction that dominates the step.
//
//   C[g][m][n] (+)= sum_k A[g](m,k) * Wt[g](n,k)  (+ bias[g][n])
//
// What is different from k_gemm_bf3 (gemm.h), and why (measured there: 0.20 MFMA-busy, LDS-bound):
//   * 128 x (32*NF) workgroup tile, 4 waves as 2 (M) x 2 (N), a wave owns 64 x (16*NF): one LDS fragment read feeds
//     4x the MFMAs of the 16-row waves of k_gemm_bf3 (8 + 2*NF ds_read_b128 for 12*NF MFMAs per wave and k-step).
//   * The weight operand never passes through registers or the fp32 -> (hi, lo) conversion inside the GEMM: once per step
//     k_prep_wimg writes every weight as a PRE-TILED split-bf16 image -- per (group, n-tile, 32-wide k-step) one block
//     [hi | lo] laid out exactly like the LDS tile -- and the GEMM moves a block with NF*4 global_load_lds_dwordx4
//     (LDS-DMA, 1 KiB per wave-instruction, lane-linear on both sides), double-buffered across the k-loop.
//   * Only the activation tile (A) is converted in flight: 4 float4 per thread and k-step, one ds_write_b64 pair each.
//   * MFMA operands are swapped (weights as the A input, activations as the B input), so a lane's 4 accumulator
//     registers are 4 consecutive COLUMNS of one output row: the epilogue stores 16 bytes per lane straight from the
//     registers (no LDS staging) and the BatchNorm statistics of a 64-row tile are a register + 16-lane reduction inside
//     ONE wave (no cross-wave step).
// LDS: A hi+lo 16 KB + 2 x W block (NF*4 KB) = 48 KB at NF = 4 -> three workgroups per CU hide each other's barriers.
#pragma once
#include "gemm.h"

struct WImgDesc {
    const __bf16* img;      // [G][NT][KS][2 (hi, lo)][TN*32] bf16, TN = 32*NF
    int NF, NT, KS;
};
static inline int64_t wimg_elems(int G, int N, int K, int NF) {
    const int TN = 32 * NF;
    return (int64_t)G * ((N + TN - 1) / TN) * ((K + 31) / 32) * 2 * TN * 32;
}
// n-fragments per wave: the widest tile that wastes the fewest padded columns
static inline int wide_nf(int N) {
    if (N <= 64) return 2;
    const int p4 = (N + 127) / 128 * 128, p3 = (N + 95) / 96 * 96;
    return p3 < p4 ? 3 : 4;
}

// ---- weight images ------------------------------------------------------------------------------------------
// element (n, k) of group g is W[g*gs + n*sn + k*sk]: (sn, sk) = (K, 1) for the forward (torch Linear weight [out, in]),
// (1, in_dim) for the dgrad view (n = input feature, k = output feature).
struct WPrepOne { const float* W; __bf16* img; int G, N, K, NF, NT, KS; int64_t gs, sn, sk; };
#define WPREP_MAX 32
struct WPrepAllP { int n; WPrepOne d[WPREP_MAX]; };

static __global__ __launch_bounds__(256) void k_prep_wimg(const WPrepAllP a) {
    const WPrepOne& p = a.d[blockIdx.y];
    const int TN = 32 * p.NF;
    const int blocks = p.G * p.NT * p.KS;
    for (int b = blockIdx.x; b < blocks; b += gridDim.x) {
        const int s = b % p.KS, j = (b / p.KS) % p.NT, g = b / (p.KS * p.NT);
        __bf16* hi = p.img + (int64_t)b * 2 * TN * 32;
        __bf16* lo = hi + TN * 32;
        for (int idx = threadIdx.x; idx < TN * 4; idx += 256) {
            const int row = idx >> 2, plane = idx & 3;
            const int n = j * TN + row, k0 = s * 32 + plane * 8;
            bf16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                const float x = (n < p.N && k < p.K) ? p.W[(int64_t)g * p.gs + (int64_t)n * p.sn + (int64_t)k * p.sk] : 0.f;
                h[e] = (__bf16)x;
                l[e] = (__bf16)(x - (float)h[e]);
            }
            const int o = bf3_off(TN, row, plane);
            *(bf16x8*)(hi + o) = h;
            *(bf16x8*)(lo + o) = l;
        }
    }
}

// ---- the GEMM ------------------------------------------------------------------------------------------------
typedef float gw_v4f __attribute__((ext_vector_type(4)));
// diagnostics (AREAD_GEMM_DBG & 1): s_memrealtime stamps of four workgroups' wave 0, read back by aread_debug_gemm_stamps
static __device__ unsigned long long g_gw_stamps[4][256];
#define GW_STAMP()                                                                                       \
    do {                                                                                                 \
        if (stamping) {                                                                                  \
            unsigned long long t_;                                                                       \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            if (n_st < 256) g_gw_stamps[st_slot][n_st] = t_;                                             \
            ++n_st;                                                                                      \
        }                                                                                                \
    } while (0)

// ADB: double-buffered A image (64 KB of LDS at NF = 4: two workgroups per CU, one barrier per k-step); without it the wave
// holds the step's fragments in registers across a second barrier and overwrites the single A image while its MFMAs run
// (48 KB: three workgroups per CU -- what a 608-tile launch needs to stay in one residency round).
template <int NF, bool ADB>
__global__ __launch_bounds__(GEMM_THREADS, ADB ? 2 : 3) void k_gemm_bf3w(const GemmP p, const WImgDesc wd) {
    constexpr int TM = 128, TN = 32 * NF;
    constexpr int A_ELEMS = TM * 32, W_ELEMS = TN * 32;                 // bf16 elements of one (hi or lo) image
    constexpr int NAB = ADB ? 2 : 1;
    __shared__ __attribute__((aligned(1024))) char s_lds[(2 * NAB * A_ELEMS + 4 * W_ELEMS) * 2];
    __bf16* Ab = (__bf16*)s_lds;                                        // [NAB buffers][hi | lo][A_ELEMS]
    __bf16* Wb = Ab + 2 * NAB * A_ELEMS;                                // [2 buffers][hi | lo][W_ELEMS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int g = blockIdx.z;
    // XCD-aware tile mapping (workgroups are dealt round-robin over the 8 XCDs): all column tiles of a row tile on one XCD
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int nx = gridDim.x, ny8 = (gridDim.y / 8) * 8;
        const int id = blockIdx.x + nx * blockIdx.y;
        if (id < nx * ny8) {
            const int xcd = id & 7, slot = id >> 3;
            bx = slot % nx;
            by = (slot / nx) * 8 + xcd;
        }
    }
    const int m0 = by * TM, n0 = bx * TN;
    // a 128-row tile is two 64-row plan tiles, each inside one segment; either may be unused / an inactive tower's
    bool live[2];
    int nvalid[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = 2 * by + h;
        live[h] = (int64_t)t * TILE_M < p.M;
        nvalid[h] = TILE_M;
        if (live[h] && p.gate_axis == 1) {
            const int seg = p.tile_seg[t];
            live[h] = seg >= 0 && !(p.active && !p.active[seg * p.active_ld + g]);
            nvalid[h] = p.tile_valid[t];
        }
    }
    if (!live[0] && !live[1]) return;

    const float* Ag = p.A + (int64_t)g * p.a_gs;
    const __bf16* wsrc = wd.img + ((int64_t)(g * wd.NT + bx) * wd.KS) * 2 * W_ELEMS + wave * NF * 512 + lane * 8;
    const int KS = wd.KS;

    f32x4 acc[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A rows travel global -> registers two k-steps ahead of their use (the load latency under load is ~2 us, one k-step of
    // MFMA work is ~1 us at three workgroups per CU).  Loads are unconditional (rows and the k offset clamped into the row)
    // so that the counted s_waitcnt below always sees exactly four younger loads; a ragged K tail is zeroed in registers.
    gw_v4f avA[4], avB[4];
    const int a_row = tid >> 3, a_kq = tid & 7;                         // + 32 rows per q
    const float* a_ptr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = min(m0 + a_row + 32 * q, p.M - 1);
        a_ptr[q] = Ag + (int64_t)r * p.lda + 4 * a_kq;
    }
    const int k_last = (p.K + 3) / 4 * 4 - 4 - 4 * a_kq;                // clamp: the last float4 of the row's K columns (may be < 0: a_ptr holds +4*a_kq)
    // The A loads are inline asm so that the compiler keeps no scoreboard entry for them (it would drain vmcnt, LDS-DMA
    // included, before their first use); the counted waits at the step ends cover them, and `tie` pins the consumers of a
    // register set behind the wait that completed it.
    auto loadA = [&](gw_v4f (&av)[4], int k0) {
        const int kc = min(k0, k_last);
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(av[q]) : "v"(a_ptr[q] + kc) : "memory");
    };
    auto tie = [&](gw_v4f (&av)[4]) { asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3])); };
    auto storeA1 = [&](const gw_v4f (&av)[4], int k0, int buf, int q) {
        float x[4] = {av[q][0], av[q][1], av[q][2], av[q][3]};
        const int kleft = p.K - k0 - 4 * a_kq;                          // columns of this float4 inside K (selects, no branch)
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = i < kleft ? x[i] : 0.f;
        bf16x4 h, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = (__bf16)x[i];
            l[i] = (__bf16)(x[i] - (float)h[i]);
        }
        const int o = (ADB ? buf : 0) * 2 * A_ELEMS + bf3_off(TM, a_row + 32 * q, a_kq >> 1) + 4 * (a_kq & 1);
        *(bf16x4*)(Ab + o) = h;
        *(bf16x4*)(Ab + A_ELEMS + o) = l;
    };
    auto dmaW = [&](int s, int buf) {
        const __bf16* src = wsrc + (int64_t)s * 2 * W_ELEMS;
        __bf16* dst = Wb + buf * 2 * W_ELEMS + wave * NF * 512;
#pragma unroll
        for (int q = 0; q < NF; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 512),
                                             (__attribute__((address_space(3))) void*)(dst + q * 512), 16, 0, 0);
    };

    const int fr = lane & 15, fk = lane >> 4;
    const int a_off = bf3_off(TM, wr * 64 + fr, fk);                    // + mi*16 rows: the XOR only touches row bits 1..2
    const int w_off = bf3_off(TN, wc * 16 * NF + fr, fk);

    const int lin_wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int st_slot = lin_wg == 0 ? 0 : lin_wg == 101 ? 1 : lin_wg == 303 ? 2 : lin_wg == 520 ? 3 : -1;
    const bool stamping = (p.dbg & 1) && st_slot >= 0 && tid == 0;
    int n_st = 0;
    GW_STAMP();
    // One barrier per k-step: while the MFMAs of step s run, the wave converts and stores the A rows of step s+1 into the
    // other A buffer and the LDS-DMA brings W block s+1 into the other W buffer.  Register sets: step s+1 sits in
    // av[(s+1)&1] (loaded two steps ago), av[s&1] is loading step s+2, and av[(s+1)&1] is reloaded with step s+3 after its store.
    // The step is straight-line code (prefetches past the last step are clamped repeats whose results nobody reads), so the
    // compiler can count its own waits instead of draining vmcnt.
    auto step = [&](int s, gw_v4f (&avn)[4], gw_v4f (&avo)[4]) {
        GW_STAMP();
        dmaW(min(s + 1, KS - 1), (s + 1) & 1);
        const __bf16* Ah = Ab + (ADB ? (s & 1) : 0) * 2 * A_ELEMS;
        const __bf16* Al = Ah + A_ELEMS;
        const __bf16* Wh = Wb + (s & 1) * 2 * W_ELEMS;
        const __bf16* Wl = Wh + W_ELEMS;
        bf16x8 ah[4], al[4], wh[NF], wl[NF];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ah[j] = *(const bf16x8*)(Ah + a_off + j * 128);
            al[j] = *(const bf16x8*)(Al + a_off + j * 128);
        }
        if (ADB) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                wh[i] = *(const bf16x8*)(Wh + w_off + i * 128);
                wl[i] = *(const bf16x8*)(Wl + w_off + i * 128);
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave holds its A fragments: the A image is free
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (!ADB) {
                wh[i] = *(const bf16x8*)(Wh + w_off + i * 128);
                wl[i] = *(const bf16x8*)(Wl + w_off + i * 128);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], ah[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], al[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], ah[j], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) storeA1(avn, (s + 1) * 32, (s + 1) & 1, q);
        GW_STAMP();
        loadA(avn, (s + 3) * 32);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");     // W block s+1 and A step s+2 landed; the 4 loads above fly
        GW_STAMP();
        asm volatile("s_barrier" ::: "memory");
        tie(avo);
    };
    dmaW(0, 0);
    loadA(avA, 0);
    loadA(avB, 32);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    tie(avA);
#pragma unroll
    for (int q = 0; q < 4; ++q) storeA1(avA, 0, 0, q);
    loadA(avA, 64);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    tie(avB);
    for (int s = 0; s < KS; s += 2) {
        step(s, avB, avA);                                              // stores step s+1 (odd) from avB
        if (s + 1 < KS) step(s + 1, avA, avB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the clamped prefetches of the last steps

    GW_STAMP();
    // ---- epilogue: lane holds C[m = .. + mi*16 + fr][n = .. + ni*16 + fk*4 + 0..3] ---------------------------------
    if (!live[wr]) return;
    const int nw0 = n0 + wc * 16 * NF;
    float* Cg = p.C + (int64_t)g * p.c_gs;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int n = nw0 + i * 16 + fk * 4;
        if (p.bias && n < p.N) {
            const float4 b = *(const float4*)(p.bias + (int64_t)g * p.bias_gs + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[i][j][0] += b.x; acc[i][j][1] += b.y; acc[i][j][2] += b.z; acc[i][j][3] += b.w; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wr * 64 + j * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int n = nw0 + i * 16 + fk * 4;
            if (n >= p.N) continue;
            float4* dst = (float4*)(Cg + (int64_t)m * p.ldc + n);
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
        }
    }
    GW_STAMP();
    if (stamping) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GW_STAMP(); }
    if (p.stat_part) {
        // (mean, M2) over the valid rows of this wave's 64-row tile, per column: rows live on (j, fr), columns on (i, fk, reg)
        const int nv = nvalid[wr];
        const float inv = 1.0f / (float)nv;
        const int tile = 2 * by + wr;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            float mean[4], m2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) s += (j * 16 + fr < nv) ? acc[i][j][r] : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
                mean[r] = s * inv;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = acc[i][j][r] - mean[r];
                    s += (j * 16 + fr < nv) ? d * d : 0.f;
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
                m2[r] = s;
            }
            const int n = nw0 + i * 16 + fk * 4;
            if (fr == 0 && n < p.N) {
                float* o = p.stat_part + ((int64_t)tile * p.stat_ld + (int64_t)g * p.N + n) * 2;
                *(float4*)o = make_float4(mean[0], m2[0], mean[1], m2[1]);
                *(float4*)(o + 4) = make_float4(mean[2], m2[2], mean[3], m2[3]);
            }
        }
    }
}

int launch_gemm_bf3w(const GemmP& p, const WImgDesc& w, hipStream_t st);
int launch_prep_wimg(const WPrepAllP& a, hipStream_t st);
