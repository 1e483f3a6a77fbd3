// plan.hip -- row plan: stable bucketing of samples by BN segment (domain) with tile padding.
// Replaces the per-domain loaders and the host-side domain loop of run.py:310-353,609-611.
#include "common.h"
#include <cstdlib>

#define PLAN_WAVES_PER_BLOCK 4
#define PLAN_MAX_WAVES 2048

// Three short kernels instead of one long single-workgroup kernel (the plan is the first thing on the step's critical
// path).  The samples are cut into W contiguous ranges, one per wave, W = min(ceil(B/64), 2048):
//   k_plan_count  : wave w counts its samples per segment (one ballot per segment value, lane L keeps the counter of
//                   segment L in a register: MAX_SEG == wave size; no atomics) -> cnt[w][seg]; also clears the tables
//   k_plan_prefix : one workgroup: per segment an exclusive prefix over the waves, the tile-padded segment starts,
//                   tile tables and the header
//   k_plan_rank   : wave w assigns rows in sample order with wave-private running counters
// Stable and deterministic.  cnt / bad live in a scratch area behind sample_row in the plan buffer.
struct PlanGeom {
    int B, f_in, seg_col, n_seg, max_rows, max_tiles, n_waves, per_wave;
};
static inline __host__ __device__ int plan_waves(int64_t B) {
    const int64_t w = (B + WAVE - 1) / WAVE;
    return (int)(w < PLAN_MAX_WAVES ? w : PLAN_MAX_WAVES);
}
static inline __host__ __device__ int plan_per_wave(int64_t B, int n_waves) {
    return (int)(((B + n_waves - 1) / n_waves + WAVE - 1) / WAVE * WAVE);
}
__device__ __forceinline__ int plan_seg_of(const int32_t* __restrict__ x, const PlanGeom& g, int b, int b1, int& bad) {
    int s = -1;
    if (b < b1) {
        s = 0;
        if (g.seg_col >= 0) {
            s = x[(int64_t)b * g.f_in + g.seg_col];
            if (s < 0 || s >= g.n_seg) { ++bad; s = s < 0 ? 0 : g.n_seg - 1; }
        }
    }
    return s;
}

__global__ __launch_bounds__(PLAN_WAVES_PER_BLOCK * WAVE) void k_plan_count(const int32_t* __restrict__ x, PlanGeom g,
                                                                             int32_t* __restrict__ plan) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * PLAN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    int32_t* tile_seg = plan + PLAN_HDR + 2 * MAX_SEG;
    int32_t* tile_valid = tile_seg + g.max_tiles;
    int32_t* row_sample = tile_valid + g.max_tiles;
    int32_t* cnt = row_sample + g.max_rows + g.B;                  // [n_waves][MAX_SEG], then bad[n_waves]
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int i = gtid; i < g.max_rows; i += gsz) row_sample[i] = -1;
    for (int i = gtid; i < g.max_tiles; i += gsz) { tile_seg[i] = -1; tile_valid[i] = 0; }
    if (gw >= g.n_waves) return;
    const int b0 = gw * g.per_wave, b1 = min(g.B, b0 + g.per_wave);
    int my_cnt = 0, bad = 0;
    for (int base = b0; base < b1; base += WAVE) {
        const int s = plan_seg_of(x, g, base + lane, b1, bad);
        for (int v = 0; v < g.n_seg; ++v) {
            const unsigned long long m = __ballot(s == v);
            if (lane == v) my_cnt += __popcll(m);
        }
    }
    cnt[(int64_t)gw * MAX_SEG + lane] = my_cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    if (lane == 0) cnt[(int64_t)g.n_waves * MAX_SEG + gw] = bad;
}

__global__ __launch_bounds__(1024) void k_plan_prefix(PlanGeom g, int32_t* __restrict__ plan) {
    __shared__ int s_part[16][MAX_SEG];
    __shared__ int s_total[MAX_SEG];
    __shared__ int s_bad[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* seg_count = plan + PLAN_HDR;
    int32_t* seg_start = seg_count + MAX_SEG;
    int32_t* tile_seg = seg_start + MAX_SEG;
    int32_t* tile_valid = tile_seg + g.max_tiles;
    int32_t* cnt = tile_valid + g.max_tiles + g.max_rows + g.B;
    const int per = (g.n_waves + 15) / 16;
    const int w0 = wave * per, w1 = min(g.n_waves, w0 + per);
    int sum = 0, bad = 0;
#pragma unroll 8
    for (int w = w0; w < w1; ++w) sum += cnt[(int64_t)w * MAX_SEG + lane];
    for (int w = w0 + lane; w < w1; w += WAVE) bad += cnt[(int64_t)g.n_waves * MAX_SEG + w];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    s_part[wave][lane] = sum;
    if (lane == 0) s_bad[wave] = bad;
    __syncthreads();
    if (tid < MAX_SEG) {
        int run = 0;
        for (int j = 0; j < 16; ++j) { const int c = s_part[j][tid]; s_part[j][tid] = run; run += c; }
        s_total[tid] = tid < g.n_seg ? run : 0;
    }
    __syncthreads();
    {
        int run = s_part[wave][lane];                              // exclusive prefix over the waves, in place
        for (int w = w0; w < w1; ++w) {
            const int c = cnt[(int64_t)w * MAX_SEG + lane];
            cnt[(int64_t)w * MAX_SEG + lane] = run;
            run += c;
        }
    }
    if (tid < MAX_SEG) {                                       // wave 0: lane = segment; tile ranges by a wave scan of the tile counts
        const int c = s_total[tid];
        const int nt = (c + TILE_M - 1) / TILE_M;
        int incl = nt;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int first = incl - nt;
        seg_count[tid] = c;
        seg_start[tid] = first * TILE_M;
        for (int t = 0; t < nt; ++t) {
            tile_seg[first + t] = tid;
            const int v = c - t * TILE_M;
            tile_valid[first + t] = v > TILE_M ? TILE_M : v;
        }
        const int total_tiles = __shfl(incl, WAVE - 1);
        if (tid == 0) {
            int nbad = 0;
            for (int j = 0; j < 16; ++j) nbad += s_bad[j];
            plan[PLAN_B] = g.B;
            plan[PLAN_NSEG] = g.n_seg;
            plan[PLAN_ROWS] = total_tiles * TILE_M;
            plan[PLAN_NTILES] = total_tiles;
            plan[PLAN_NBAD] = nbad;
        }
    }
}

__global__ __launch_bounds__(PLAN_WAVES_PER_BLOCK * WAVE) void k_plan_rank(const int32_t* __restrict__ x, PlanGeom g,
                                                                            int32_t* __restrict__ plan) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * PLAN_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (gw >= g.n_waves) return;
    const int32_t* seg_start = plan + PLAN_HDR + MAX_SEG;
    int32_t* row_sample = plan + PLAN_HDR + 2 * MAX_SEG + 2 * g.max_tiles;
    int32_t* sample_row = row_sample + g.max_rows;
    const int32_t* cnt = sample_row + g.B;
    const int b0 = gw * g.per_wave, b1 = min(g.B, b0 + g.per_wave);
    int run = seg_start[lane] + cnt[(int64_t)gw * MAX_SEG + lane];  // next row of segment `lane` for this wave
    const unsigned long long lt = (1ull << lane) - 1ull;
    int bad = 0;
    for (int base = b0; base < b1; base += WAVE) {
        const int b = base + lane;
        const int s = plan_seg_of(x, g, b, b1, bad);
        unsigned long long mine = 0ull;
        int add = 0;
        for (int v = 0; v < g.n_seg; ++v) {
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

// The same plan in ONE launch for batches of up to PLAN_SINGLE_MAX samples: a single 1024-thread workgroup (16 waves, each a
// contiguous range of samples whose segment ids stay in registers between the counting and the ranking pass), the wave x segment
// prefix in LDS.  Bit-identical tables.  MEASURED SLOWER than the three short launches (step +18 us: one workgroup walks the
// whole strided id column and clears the tables alone), so it is off by default (AREAD_PLAN_SINGLE=1 for A/B); the plan test
// covers it through the environment switch.
#define PLAN_SINGLE_MAX 16384
#define PLAN_SINGLE_IT (PLAN_SINGLE_MAX / 1024)
__global__ __launch_bounds__(1024) void k_plan_single(const int32_t* __restrict__ x, PlanGeom g, int32_t* __restrict__ plan) {
    __shared__ int s_part[16][MAX_SEG];
    __shared__ int s_start[MAX_SEG];
    __shared__ int s_bad[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* seg_count = plan + PLAN_HDR;
    int32_t* seg_start = seg_count + MAX_SEG;
    int32_t* tile_seg = seg_start + MAX_SEG;
    int32_t* tile_valid = tile_seg + g.max_tiles;
    int32_t* row_sample = tile_valid + g.max_tiles;
    int32_t* sample_row = row_sample + g.max_rows;
    for (int i = tid; i < g.max_rows; i += 1024) row_sample[i] = -1;
    for (int i = tid; i < g.max_tiles; i += 1024) { tile_seg[i] = -1; tile_valid[i] = 0; }
    const int per = ((g.B + 15) / 16 + WAVE - 1) / WAVE * WAVE;       // samples per wave, a multiple of 64
    const int b0 = wave * per, b1 = min(g.B, b0 + per);
    int seg[PLAN_SINGLE_IT];
    int bad = 0;
#pragma unroll
    for (int it = 0; it < PLAN_SINGLE_IT; ++it) seg[it] = plan_seg_of(x, g, b0 + it * WAVE + lane, b1, bad);
    int my_cnt = 0;
#pragma unroll
    for (int it = 0; it < PLAN_SINGLE_IT; ++it) {
        if (b0 + it * WAVE >= b1) break;                               // wave-uniform
        for (int v = 0; v < g.n_seg; ++v) {
            const unsigned long long m = __ballot(seg[it] == v);
            if (lane == v) my_cnt += __popcll(m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    s_part[wave][lane] = my_cnt;
    if (lane == 0) s_bad[wave] = bad;
    // the table clears above and the table / row writes below hit the same addresses from different waves: have every
    // clear acknowledged by the L2 before any wave passes the barrier (the workgroup-scope fence of __syncthreads alone
    // relies on the in-order memory path of one CU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < MAX_SEG) {                                               // wave 0: lane = segment
        int run = 0;
        for (int j = 0; j < 16; ++j) { const int c = s_part[j][tid]; s_part[j][tid] = run; run += c; }
        const int c = tid < g.n_seg ? run : 0;
        const int nt = (c + TILE_M - 1) / TILE_M;
        int incl = nt;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int first = incl - nt;
        seg_count[tid] = c;
        seg_start[tid] = first * TILE_M;
        s_start[tid] = first * TILE_M;
        for (int t = 0; t < nt; ++t) {
            tile_seg[first + t] = tid;
            const int v = c - t * TILE_M;
            tile_valid[first + t] = v > TILE_M ? TILE_M : v;
        }
        const int total_tiles = __shfl(incl, WAVE - 1);
        if (tid == 0) {
            int nbad = 0;
            for (int j = 0; j < 16; ++j) nbad += s_bad[j];
            plan[PLAN_B] = g.B;
            plan[PLAN_NSEG] = g.n_seg;
            plan[PLAN_ROWS] = total_tiles * TILE_M;
            plan[PLAN_NTILES] = total_tiles;
            plan[PLAN_NBAD] = nbad;
        }
    }
    __syncthreads();
    int run = s_start[lane] + s_part[wave][lane];                      // next row of segment `lane` for this wave
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int it = 0; it < PLAN_SINGLE_IT; ++it) {
        if (b0 + it * WAVE >= b1) break;
        const int b = b0 + it * WAVE + lane;
        const int s = seg[it];
        unsigned long long mine = 0ull;
        int add = 0;
        for (int v = 0; v < g.n_seg; ++v) {
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
    L->words = L->off_sample_row + B + (int64_t)plan_waves(B) * (MAX_SEG + 1);   // + scratch: cnt[W][MAX_SEG], bad[W]
    return AREAD_OK;
}

int g_plan_single = -1;     // AREAD_PLAN_SINGLE / aread_debug_set("plan_single", v)
extern "C" int aread_plan_build(const int32_t* x, int64_t B, int f_in, int seg_col, int n_seg, int32_t* plan,
                                void* stream) {
    AR_CHECK_ARG(plan != nullptr, "aread_plan_build: plan is null");
    AR_CHECK_ARG(B > 0 && B < (1ll << 30), "aread_plan_build: bad B=%lld", (long long)B);
    AR_CHECK_ARG(n_seg >= 1 && n_seg <= MAX_SEG, "aread_plan_build: n_seg=%d not in [1,%d]", n_seg, MAX_SEG);
    AR_CHECK_ARG(seg_col < f_in, "aread_plan_build: seg_col=%d >= f_in=%d", seg_col, f_in);
    AR_CHECK_ARG(seg_col < 0 || x != nullptr, "aread_plan_build: x is null");
    const int64_t mr = plan_max_rows(B, n_seg);
    PlanGeom g;
    g.B = (int)B; g.f_in = f_in; g.seg_col = seg_col; g.n_seg = n_seg; g.max_rows = (int)mr; g.max_tiles = (int)(mr / TILE_M);
    g.n_waves = plan_waves(B); g.per_wave = plan_per_wave(B, g.n_waves);
    const hipStream_t st = (hipStream_t)stream;
    int& single = g_plan_single;
    if (single < 0) { const char* e = getenv("AREAD_PLAN_SINGLE"); single = e ? atoi(e) : 0; }   // measured +18 us per step: off
    if (single && B <= PLAN_SINGLE_MAX) {
        hipLaunchKernelGGL(k_plan_single, dim3(1), dim3(1024), 0, st, x, g, plan);
        AR_LAUNCH_CHECK();
        return AREAD_OK;
    }
    const unsigned blocks = (unsigned)((g.n_waves + PLAN_WAVES_PER_BLOCK - 1) / PLAN_WAVES_PER_BLOCK);
    hipLaunchKernelGGL(k_plan_count, dim3(blocks), dim3(PLAN_WAVES_PER_BLOCK * WAVE), 0, st, x, g, plan);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_plan_prefix, dim3(1), dim3(1024), 0, st, g, plan);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_plan_rank, dim3(blocks), dim3(PLAN_WAVES_PER_BLOCK * WAVE), 0, st, x, g, plan);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
