// gemm_wide.h -- pre-tiled split-bf16 weight images: every Linear weight of the tower stacks once per step as [hi | lo] bf16
// blocks laid out exactly like the LDS tile of its consumer (the fused tower kernels, tower_fused*.h, stream them with
// global_load_lds).  Round 2 also had a 128-row GEMM on these images (k_gemm_bf3w); it and round 3's persistent 32x32x16
// kernel only ever tied k_gemm_bf3 (gemm.h) at every expert shape and were removed: profiles/r03_gemm_experiments.txt.
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

int launch_prep_wimg(const WPrepAllP& a, hipStream_t st);
