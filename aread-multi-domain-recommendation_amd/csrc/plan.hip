// plan.hip -- row plan: stable bucketing of samples by BN segment (domain) with tile padding.
// Replaces the per-domain loaders and the host-side domain loop of run.py:310-353,609-611.
#include "common.h"

#define PLAN_THREADS 1024
#define PLAN_REGS 8

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
    // the wave's segment ids are fetched in batches of PLAN_REGS x 64 with all loads in flight at once (the id column
    // is strided by f_in*4 bytes: one load latency per batch instead of one per 64 samples) and reused by both passes
    auto load_batch = [&](int base, int (&sg)[PLAN_REGS]) {
#pragma unroll
        for (int j = 0; j < PLAN_REGS; ++j) {
            const int b = base + j * WAVE + lane;
            int s = -1;
            if (b < b1) {
                s = 0;
                if (seg_col >= 0) {
                    s = x[(int64_t)b * f_in + seg_col];
                    if (s < 0 || s >= n_seg) { atomicAdd(&s_bad, 1); s = s < 0 ? 0 : n_seg - 1; }
                }
            }
            sg[j] = s;
        }
    };
    const bool single_batch = (b1 - b0) <= PLAN_REGS * WAVE;
    int sg[PLAN_REGS];
    // Lane L keeps the counter of segment L in a register (MAX_SEG == wave size); one ballot per segment value.
    int my_cnt = 0;
    for (int base = b0; base < b1; base += PLAN_REGS * WAVE) {     // pass 1: count
        load_batch(base, sg);
#pragma unroll
        for (int j = 0; j < PLAN_REGS; ++j)
            for (int v = 0; v < n_seg; ++v) {
                const unsigned long long m = __ballot(sg[j] == v);
                if (lane == v) my_cnt += __popcll(m);
            }
    }
    s_cnt[wave][lane] = my_cnt;
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
        plan[PLAN_NBAD] = single_batch ? s_bad : s_bad / 2;        // re-loaded batches classify every sample twice
    }
    __syncthreads();
    int run = s_start[lane] + s_cnt[wave][lane];                   // next row of segment `lane` for this wave
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = b0; base < b1; base += PLAN_REGS * WAVE) {     // pass 2: assign rows in sample order
        if (!single_batch) load_batch(base, sg);
#pragma unroll
        for (int j = 0; j < PLAN_REGS; ++j) {
            const int b = base + j * WAVE + lane;
            const int s = sg[j];
            unsigned long long mine = 0ull;
            int add = 0;
            for (int v = 0; v < n_seg; ++v) {
                const unsigned long long m = __ballot(s == v);
                if (s == v) mine = m;
                if (lane == v) add = __popcll(m);
            }
            const int first = __shfl(run, s < 0 ? 0 : s);
            if (s >= 0) {
                const int r = first + __popcll(mine & lt);
                row_sample[r] = b;
                sample_row[b] = r;
            }
            run += add;
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
