// plan.hip -- row plan: stable bucketing of samples by BN segment (domain) with tile padding.
// Replaces the per-domain loaders and the host-side domain loop of run.py:310-353,609-611.
#include "common.h"

#define PLAN_THREADS 1024

// One workgroup of 16 waves.  Wave w owns the contiguous sample range [w*C, (w+1)*C): pass 1 counts its samples per
// segment (ballot + popcount, no atomics), a prefix over waves gives every wave its first row per segment, pass 2
// assigns rows in sample order with wave-private running counters.  Stable and deterministic; three block barriers.
__global__ __launch_bounds__(PLAN_THREADS) void k_plan_build(const int32_t* __restrict__ x, int B, int f_in,
                                                              int seg_col, int n_seg, int32_t* __restrict__ plan,
                                                              int max_rows, int max_tiles) {
    constexpr int NW = PLAN_THREADS / WAVE;
    __shared__ int s_cnt[NW][MAX_SEG];       // pass 1: samples of wave w in segment s; pass 2: next row
    __shared__ int s_start[MAX_SEG];
    __shared__ int s_total[MAX_SEG];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* seg_count = plan + PLAN_HDR;
    int32_t* seg_start = seg_count + MAX_SEG;
    int32_t* tile_seg = seg_start + MAX_SEG;
    int32_t* tile_valid = tile_seg + max_tiles;
    int32_t* row_sample = tile_valid + max_tiles;
    int32_t* sample_row = row_sample + max_rows;
    const int chunk = ((B + NW - 1) / NW + WAVE - 1) / WAVE * WAVE;
    const int b0 = wave * chunk, b1 = min(B, b0 + chunk);

    for (int i = lane; i < MAX_SEG; i += WAVE) s_cnt[wave][i] = 0;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < max_rows; i += PLAN_THREADS) row_sample[i] = -1;
    for (int i = tid; i < max_tiles; i += PLAN_THREADS) { tile_seg[i] = -1; tile_valid[i] = 0; }
    __syncthreads();
    auto seg_of = [&](int b) -> int {
        if (b >= b1) return -1;
        if (seg_col < 0) return 0;
        int s = x[(int64_t)b * f_in + seg_col];
        if (s < 0 || s >= n_seg) { atomicAdd(&s_bad, 1); s = s < 0 ? 0 : n_seg - 1; }
        return s;
    };
    for (int base = b0; base < b1; base += WAVE) {                 // pass 1
        const int s = seg_of(base + lane);
        unsigned long long todo = __ballot(s >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int s0 = __shfl(s, leader);
            const unsigned long long same = __ballot(s == s0);
            if (lane == leader) s_cnt[wave][s0] += __popcll(same);
            todo &= ~same;
        }
    }
    __syncthreads();
    if (tid < MAX_SEG) {                                           // per segment: total and exclusive prefix over waves
        int run = 0;
        for (int w = 0; w < NW; ++w) { const int c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
        s_total[tid] = tid < n_seg ? run : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int row = 0;
        for (int s = 0; s < MAX_SEG; ++s) {
            const int c = s_total[s];
            s_start[s] = row;
            seg_count[s] = c;
            seg_start[s] = row;
            const int nt = (c + TILE_M - 1) / TILE_M;
            for (int t = 0; t < nt; ++t) {
                const int ti = row / TILE_M + t;
                tile_seg[ti] = s;
                const int v = c - t * TILE_M;
                tile_valid[ti] = v > TILE_M ? TILE_M : v;
            }
            row += nt * TILE_M;
        }
        plan[PLAN_B] = B;
        plan[PLAN_NSEG] = n_seg;
        plan[PLAN_ROWS] = row;
        plan[PLAN_NTILES] = row / TILE_M;
        plan[PLAN_NBAD] = s_bad / 2;                               // every sample is classified twice
    }
    __syncthreads();
    for (int base = b0; base < b1; base += WAVE) {                 // pass 2
        const int b = base + lane;
        const int s = seg_of(b);
        unsigned long long todo = __ballot(s >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int s0 = __shfl(s, leader);
            const unsigned long long same = __ballot(s == s0);
            if (s == s0) {
                const int r = s_start[s0] + s_cnt[wave][s0] + __popcll(same & ((1ull << lane) - 1ull));
                row_sample[r] = b;
                sample_row[b] = r;
            }
            if (lane == leader) s_cnt[wave][s0] += __popcll(same);
            todo &= ~same;
        }
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
