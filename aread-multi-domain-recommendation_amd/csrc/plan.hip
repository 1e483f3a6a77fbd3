// plan.hip -- row plan: stable bucketing of samples by BN segment (domain) with tile padding.
// Replaces the per-domain loaders and the host-side domain loop of run.py:310-353,609-611.
#include "common.h"

#define PLAN_THREADS 1024

// One workgroup.  Pass 1: integer histogram (LDS atomics: order-independent, deterministic).
// Pass 2: stable rank of every sample inside its segment via wave ballots + per-wave LDS counts.
__global__ __launch_bounds__(PLAN_THREADS) void k_plan_build(const int32_t* __restrict__ x, int B, int f_in,
                                                              int seg_col, int n_seg, int32_t* __restrict__ plan,
                                                              int max_rows, int max_tiles) {
    __shared__ int s_count[MAX_SEG];
    __shared__ int s_start[MAX_SEG];
    __shared__ int s_run[MAX_SEG];                      // rows already placed per segment
    __shared__ int s_wave[PLAN_THREADS / WAVE][MAX_SEG];  // per-wave counts of the current chunk
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* seg_count = plan + PLAN_HDR;
    int32_t* seg_start = seg_count + MAX_SEG;
    int32_t* tile_seg = seg_start + MAX_SEG;
    int32_t* tile_valid = tile_seg + max_tiles;
    int32_t* row_sample = tile_valid + max_tiles;
    int32_t* sample_row = row_sample + max_rows;

    if (tid < MAX_SEG) { s_count[tid] = 0; s_run[tid] = 0; }
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < max_rows; i += PLAN_THREADS) row_sample[i] = -1;
    for (int i = tid; i < max_tiles; i += PLAN_THREADS) { tile_seg[i] = -1; tile_valid[i] = 0; }
    __syncthreads();
    for (int b = tid; b < B; b += PLAN_THREADS) {
        int s = 0;
        if (seg_col >= 0) {
            s = x[(int64_t)b * f_in + seg_col];
            if (s < 0 || s >= n_seg) { atomicAdd(&s_bad, 1); s = s < 0 ? 0 : n_seg - 1; }
        }
        atomicAdd(&s_count[s], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int row = 0;
        for (int s = 0; s < MAX_SEG; ++s) {
            int c = s < n_seg ? s_count[s] : 0;
            s_start[s] = row;
            seg_count[s] = c;
            seg_start[s] = row;
            int nt = (c + TILE_M - 1) / TILE_M;
            for (int t = 0; t < nt; ++t) {
                int ti = row / TILE_M + t;
                tile_seg[ti] = s;
                int v = c - t * TILE_M;
                tile_valid[ti] = v > TILE_M ? TILE_M : v;
            }
            row += nt * TILE_M;
        }
        plan[PLAN_B] = B;
        plan[PLAN_NSEG] = n_seg;
        plan[PLAN_ROWS] = row;
        plan[PLAN_NTILES] = row / TILE_M;
        plan[PLAN_NBAD] = s_bad;
    }
    __syncthreads();
    for (int base = 0; base < B; base += PLAN_THREADS) {
        const int b = base + tid;
        int s = -1;
        if (b < B) {
            s = 0;
            if (seg_col >= 0) {
                s = x[(int64_t)b * f_in + seg_col];
                s = s < 0 ? 0 : (s >= n_seg ? n_seg - 1 : s);
            }
        }
        for (int i = lane; i < MAX_SEG; i += WAVE) s_wave[wave][i] = 0;
        // rank inside the wave among equal segment ids
        int rank_in_wave = 0;
        unsigned long long todo = __ballot(s >= 0);
        while (todo) {
            int leader = __ffsll((long long)todo) - 1;
            int s0 = __shfl(s, leader);
            unsigned long long same = __ballot(s == s0);
            if (s == s0) rank_in_wave = __popcll(same & ((1ull << lane) - 1ull));
            if (lane == leader) s_wave[wave][s0] = __popcll(same);
            todo &= ~same;
        }
        __syncthreads();
        if (b < B) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += s_wave[w][s];
            int r = s_start[s] + s_run[s] + before + rank_in_wave;
            row_sample[r] = b;
            sample_row[b] = r;
        }
        __syncthreads();
        if (tid < MAX_SEG) {
            int add = 0;
            for (int w = 0; w < PLAN_THREADS / WAVE; ++w) add += s_wave[w][tid];
            s_run[tid] += add;
        }
        __syncthreads();
    }
}

extern "C" int aread_plan_layout_get(int64_t B, int n_seg, aread_plan_layout* L) {
    AR_CHECK_ARG(L != nullptr, "aread_plan_layout_get: null output");
    AR_CHECK_ARG(B > 0 && n_seg >= 1 && n_seg <= MAX_SEG, "aread_plan_layout_get: bad B=%lld n_seg=%d", (long long)B, n_seg);
    const int64_t mr = plan_max_rows(B, n_seg);
    L->max_rows = mr;
    L->max_tiles = mr / TILE_M;
    L->off_seg_count = PLAN_HDR;
    L->off_seg_start = L->off_seg_count + MAX_SEG;
    L->off_tile_seg = L->off_seg_start + MAX_SEG;
    L->off_tile_valid = L->off_tile_seg + L->max_tiles;
    L->off_row_sample = L->off_tile_valid + L->max_tiles;
    L->off_sample_row = L->off_row_sample + mr;
    L->words = L->off_sample_row + B;
    return AREAD_OK;
}

extern "C" int aread_plan_build(const int32_t* x, int64_t B, int f_in, int seg_col, int n_seg, int32_t* plan,
                                void* stream) {
    AR_CHECK_ARG(plan != nullptr, "aread_plan_build: plan is null");
    AR_CHECK_ARG(B > 0 && B < (1ll << 30), "aread_plan_build: bad B=%lld", (long long)B);
    AR_CHECK_ARG(n_seg >= 1 && n_seg <= MAX_SEG, "aread_plan_build: n_seg=%d not in [1,%d]", n_seg, MAX_SEG);
    AR_CHECK_ARG(seg_col < f_in, "aread_plan_build: seg_col=%d >= f_in=%d", seg_col, f_in);
    AR_CHECK_ARG(seg_col < 0 || x != nullptr, "aread_plan_build: x is null");
    int64_t mr = plan_max_rows(B, n_seg);
    hipLaunchKernelGGL(k_plan_build, dim3(1), dim3(PLAN_THREADS), 0, (hipStream_t)stream, x, (int)B, f_in, seg_col,
                       n_seg, plan, (int)mr, (int)(mr / TILE_M));
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
