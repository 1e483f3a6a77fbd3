// optim.hip -- fused Adam for the path's parameters (torch.optim.Adam as configured at run.py:830-831:
// coupled weight decay, bias correction, no amsgrad), one streaming pass per buffer.
//
//   k_adam        flat buffer with a materialised gradient (the 0.58 M dense floats); `active` mirrors autograd's
//                 grad=None rule: Adam skips tensors that took no part in the step (no decay, no moment update)
//   k_adam_table  the embedding table WITHOUT a materialised dense gradient: g = 2*l2*w (+ the batch's row
//                 gradient when the row was looked up, found through the routing workspace's flag/slot arrays),
//                 sum(w^2) for the L2 loss term in the same pass.  Traffic 6 x 4 B per element (w, m, v in and out)
//                 instead of 9 x (L2 pass: read w, write g; optimizer: read w, g, m, v, write w, m, v).
#include "common.h"
#include "route.h"

// no implicit fma contraction in this file: phase 0 and phases 1 + 2 of the table update are different kernels and
// must round identically (explicit __fmaf_rn where a fused op is meant)
#pragma clang fp contract(off)

#define AD_THREADS 256
#define AD_BLOCKS 2048          // == aread_l2_partials(): the sum(w^2) partials reduce exactly like k_l2_table's

typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
    const nt_f4 v = __builtin_nontemporal_load((const nt_f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(float4* p, const float4 v) {
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (nt_f4*)p);
}

struct AdamK {
    float b1w, b2, b2w, eps, wd, step_size, sqrt_bc2;
};

__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, const AdamK k) {
    g = g + k.wd * w;
    m = m + k.b1w * (g - m);
    v = v * k.b2 + (k.b2w * g) * g;
    const float denom = sqrtf(v) / k.sqrt_bc2 + k.eps;
    w = w - k.step_size * (m / denom);
}

static int adam_consts(const aread_adam_cfg* c, AdamK* k) {
    if (!c || c->step < 1 || !(c->beta1 >= 0.f && c->beta1 < 1.f) || !(c->beta2 >= 0.f && c->beta2 < 1.f)) return -1;
    const double bc1 = 1.0 - pow((double)c->beta1, (double)c->step);
    const double bc2 = 1.0 - pow((double)c->beta2, (double)c->step);
    k->b1w = 1.0f - c->beta1;
    k->b2 = c->beta2;
    k->b2w = 1.0f - c->beta2;
    k->eps = c->eps;
    k->wd = c->weight_decay;
    k->step_size = (float)((double)c->lr / bc1);
    k->sqrt_bc2 = (float)sqrt(bc2);
    return 0;
}

__global__ __launch_bounds__(AD_THREADS) void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, int64_t n, const uint8_t* __restrict__ active,
                                                     AdamK k) {
    const int64_t stride = (int64_t)gridDim.x * AD_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * AD_THREADS + threadIdx.x; i < n; i += stride) {
        if (active && !active[i]) continue;
        float wi = w[i], mi = m[i], vi = v[i];
        adam1(wi, mi, vi, g[i], k);
        w[i] = wi; m[i] = mi; v[i] = vi;
    }
}

// PHASE 0: every row (looked-up rows add their gradient row).  PHASE 1: only rows NOT looked up by the batch -- runs
// concurrently with the forward/backward, which read exactly the other rows.
template <int PHASE>
__global__ __launch_bounds__(AD_THREADS) void k_adam_table(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                           int64_t n4, int e4, uint8_t* __restrict__ flags,
                                                           const int32_t* __restrict__ slotmap, const float4* __restrict__ g_rows,
                                                           float l2x2, AdamK k, float* __restrict__ partial) {
    float4* w4 = (float4*)w;
    float4* m4 = (float4*)m;
    float4* v4 = (float4*)v;
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * AD_THREADS;
    for (int64_t i = (int64_t)blockIdx.x * AD_THREADS + threadIdx.x; i < n4; i += stride) {
        const int64_t row = i / e4;
        const int c = (int)(i - row * e4);
        if (PHASE == 1 && flags[row]) continue;
        float4 wi = nt_load(w4 + i), mi = nt_load(m4 + i), vi = nt_load(v4 + i);   // 1 GB stream: keep it out of L2/MALL
        acc += wi.x * wi.x + wi.y * wi.y + wi.z * wi.z + wi.w * wi.w;
        float4 g = make_float4(l2x2 * wi.x, l2x2 * wi.y, l2x2 * wi.z, l2x2 * wi.w);
        if (PHASE == 0 && flags && flags[row]) {
            const float4 s = g_rows[(int64_t)slotmap[row] * e4 + c];
            g = make_float4(__fmaf_rn(l2x2, wi.x, s.x), __fmaf_rn(l2x2, wi.y, s.y), __fmaf_rn(l2x2, wi.z, s.z),
                            __fmaf_rn(l2x2, wi.w, s.w));
        }
        adam1(wi.x, mi.x, vi.x, g.x, k);
        adam1(wi.y, mi.y, vi.y, g.y, k);
        adam1(wi.z, mi.z, vi.z, g.z, k);
        adam1(wi.w, mi.w, vi.w, g.w, k);
        nt_store(w4 + i, wi); nt_store(m4 + i, mi); nt_store(v4 + i, vi);
    }
    if (!partial) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ float s[AD_THREADS / WAVE];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < AD_THREADS / WAVE; ++i) t += s[i];
        partial[blockIdx.x] = t;
    }
}

// PHASE 2: the looked-up rows only (after the backward): row uniq_rows[s] takes gradient row s; clears the flags.
#define AD_ROW_BLOCKS 512
__global__ __launch_bounds__(AD_THREADS) void k_adam_rows(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                          int e4, const int32_t* __restrict__ uniq_rows,
                                                          const int32_t* __restrict__ edges, const float4* __restrict__ g_rows,
                                                          uint8_t* __restrict__ flags, float l2x2, AdamK k,
                                                          float* __restrict__ partial) {
    float4* w4 = (float4*)w;
    float4* m4 = (float4*)m;
    float4* v4 = (float4*)v;
    const int64_t n = (int64_t)edges[1] * e4;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * AD_THREADS + threadIdx.x; i < n; i += (int64_t)AD_ROW_BLOCKS * AD_THREADS) {
        const int64_t s = i / e4;
        const int c = (int)(i - s * e4);
        const int64_t row = uniq_rows[s];
        const int64_t j = row * e4 + c;
        float4 wi = w4[j], mi = m4[j], vi = v4[j];
        acc += wi.x * wi.x + wi.y * wi.y + wi.z * wi.z + wi.w * wi.w;
        const float4 gs = g_rows[i];
        float4 g = make_float4(__fmaf_rn(l2x2, wi.x, gs.x), __fmaf_rn(l2x2, wi.y, gs.y), __fmaf_rn(l2x2, wi.z, gs.z),
                               __fmaf_rn(l2x2, wi.w, gs.w));
        adam1(wi.x, mi.x, vi.x, g.x, k);
        adam1(wi.y, mi.y, vi.y, g.y, k);
        adam1(wi.z, mi.z, vi.z, g.z, k);
        adam1(wi.w, mi.w, vi.w, g.w, k);
        w4[j] = wi; m4[j] = mi; v4[j] = vi;
        if (c == 0) flags[row] = 0;
    }
    if (!partial) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    __shared__ float sm[AD_THREADS / WAVE];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < AD_THREADS / WAVE; ++i) t += sm[i];
        partial[blockIdx.x] = t;
    }
}

// flags of the rows this batch touched are cleared AFTER the update pass (a row's 8 float4 lanes may sit in
// different waves, so the pass itself must not clear them)
__global__ __launch_bounds__(256) void k_route_clear(const int32_t* __restrict__ uniq_rows, const int32_t* __restrict__ edges,
                                                     int n_ranks, uint8_t* __restrict__ flags) {
    const int n = edges[n_ranks];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) flags[uniq_rows[i]] = 0;
}

extern "C" int aread_adam_step(float* w, const float* g, float* m, float* v, int64_t n, const uint8_t* active,
                               const aread_adam_cfg* cfg, void* stream) {
    AR_CHECK_ARG(w && g && m && v && n > 0, "aread_adam_step: bad arguments");
    AdamK k;
    AR_CHECK_ARG(adam_consts(cfg, &k) == 0, "aread_adam_step: bad optimizer configuration");
    int64_t blocks = (n + AD_THREADS - 1) / AD_THREADS;
    if (blocks > AD_BLOCKS) blocks = AD_BLOCKS;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(AD_THREADS), 0, (hipStream_t)stream, w, g, m, v, n, active, k);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

extern "C" int aread_adam_table_l2(float* w, float* m, float* v, int64_t n_rows, int E, void* route_ws,
                                   const int32_t* uniq_rows, const int32_t* edges, const float* g_rows, float l2,
                                   const aread_adam_cfg* cfg, int phase, float* partial, void* stream) {
    AR_CHECK_ARG(w && m && v && n_rows > 0 && E > 0 && E % 4 == 0, "aread_adam_table_l2: bad arguments");
    AR_CHECK_ARG((((uintptr_t)w | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g_rows) & 15) == 0, "aread_adam_table_l2: alignment");
    AR_CHECK_ARG(phase >= 0 && phase <= 2, "aread_adam_table_l2: phase=%d", phase);
    if (phase == 0)
        AR_CHECK_ARG((route_ws == nullptr) == (g_rows == nullptr) && (route_ws == nullptr) == (uniq_rows == nullptr) &&
                     (route_ws == nullptr) == (edges == nullptr),
                     "aread_adam_table_l2: route_ws, uniq_rows, edges and g_rows go together");
    else
        AR_CHECK_ARG(route_ws && (phase == 1 || (uniq_rows && edges && g_rows)), "aread_adam_table_l2: phase %d needs the routing "
                     "workspace%s", phase, phase == 2 ? ", uniq_rows, edges and g_rows" : "");
    AdamK k;
    AR_CHECK_ARG(adam_consts(cfg, &k) == 0, "aread_adam_table_l2: bad optimizer configuration");
    uint8_t* flags = nullptr;
    const int32_t* slotmap = nullptr;
    if (route_ws) {
        RouteWs L;
        AR_CHECK_ARG(route_layout(n_rows, &L) == 0, "aread_adam_table_l2: table too large");
        AR_CHECK_ARG(((uintptr_t)route_ws & 255) == 0, "aread_adam_table_l2: workspace alignment");
        flags = (uint8_t*)route_ws + L.off_flags;
        slotmap = (const int32_t*)((char*)route_ws + L.off_slotmap);
    }
    hipStream_t st = (hipStream_t)stream;
    const int e4 = E / 4;
    if (phase == 0) {
        hipLaunchKernelGGL(k_adam_table<0>, dim3(AD_BLOCKS), dim3(AD_THREADS), 0, st, w, m, v, n_rows * e4, e4, flags, slotmap,
                           (const float4*)g_rows, 2.0f * l2, k, partial);
        AR_LAUNCH_CHECK();
        if (route_ws) {
            hipLaunchKernelGGL(k_route_clear, dim3(128), dim3(256), 0, st, uniq_rows, edges, 1, flags);
            AR_LAUNCH_CHECK();
        }
    } else if (phase == 1) {
        hipLaunchKernelGGL(k_adam_table<1>, dim3(AD_BLOCKS), dim3(AD_THREADS), 0, st, w, m, v, n_rows * e4, e4, flags, slotmap,
                           (const float4*)nullptr, 2.0f * l2, k, partial);
        AR_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_adam_rows, dim3(AD_ROW_BLOCKS), dim3(AD_THREADS), 0, st, w, m, v, e4, uniq_rows, edges,
                           (const float4*)g_rows, flags, 2.0f * l2, k, partial);
        AR_LAUNCH_CHECK();
    }
    return AREAD_OK;
}

extern "C" int aread_adam_row_partials(void) { return AD_ROW_BLOCKS; }
