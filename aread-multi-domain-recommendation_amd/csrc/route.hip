// route.hip -- lookup routing for the row-sharded embedding table (multi-GPU):
// deduplicate one batch's table rows and group them by owner rank without sorting.
//
// Global row g lives on rank g % P at local row g / P; key = (g % P) * Rp + g / P is owner-major.  The key space
// (P*Rp ~ 1.39 M for the Amazon table) is small enough for direct addressing:
//   k_route_mark   flags[key] = 1 for every lookup                      (scattered byte stores, benign races)
//   k_route_count  per 4096-key block: number of set flags              (streams the 1.4 MB flag array)
//   k_route_prefix exclusive scan of the block counts (one block)
//   k_route_assign slot of every set key (= its rank among the set keys, i.e. unique rows come out sorted by
//                  (owner, local row)), the list of unique local rows, and the per-owner boundaries
//   k_route_slot   slot[i] = slot_of_key[key_i]; flags are cleared again (the workspace stays zero between calls)
// Five short launches instead of a sort + unique; everything is integer work, bit-exact by construction.
#include "common.h"
#include "route.h"

__device__ __forceinline__ int route_key(int g, int n_rows, int P, int Rp) {
    g = g < 0 ? 0 : (g >= n_rows ? n_rows - 1 : g);                      // ids are not validated upstream: never store OOB
    return (g % P) * Rp + g / P;
}

__global__ __launch_bounds__(256) void k_route_mark(const int32_t* __restrict__ x, const int32_t* __restrict__ offsets,
                                                    int64_t n, int f_in, int n_rows, int P, int Rp,
                                                    uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i % f_in);
    flags[route_key(x[i] + offsets[j], n_rows, P, Rp)] = 1;
}

__device__ __forceinline__ int route_count16(const uint4 v) {          // flags are 0/1 bytes
    return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
}

__global__ __launch_bounds__(RT_THREADS) void k_route_count(const uint8_t* __restrict__ flags, int32_t* __restrict__ bsum) {
    __shared__ int s_w[RT_THREADS / WAVE];
    const uint4 v = ((const uint4*)flags)[(int64_t)blockIdx.x * RT_THREADS + threadIdx.x];
    int c = route_count16(v);
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & (WAVE - 1)) == 0) s_w[threadIdx.x / WAVE] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < RT_THREADS / WAVE; ++w) t += s_w[w];
        bsum[blockIdx.x] = t;
    }
}

// exclusive scan of bsum[0..n) in place, bsum[n] = total; one block, chunks of 1024 with a running carry
__global__ __launch_bounds__(1024) void k_route_prefix(int32_t* __restrict__ bsum, int n) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? bsum[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == WAVE - 1) s_w[wave] = inc;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (i < n) bsum[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[n] = s_carry;
}

__global__ __launch_bounds__(RT_THREADS) void k_route_assign(const uint8_t* __restrict__ flags, const int32_t* __restrict__ bsum,
                                                             int64_t n_keys, int n_blk, int P, int Rp,
                                                             int32_t* __restrict__ slotmap, int32_t* __restrict__ uniq_rows,
                                                             int32_t* __restrict__ edges) {
    __shared__ int s_w[RT_THREADS / WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const uint4 v = ((const uint4*)flags)[(int64_t)blockIdx.x * RT_THREADS + threadIdx.x];
    const int c = route_count16(v);
    int inc = c;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == WAVE - 1) s_w[wave] = inc;
    __syncthreads();
    int pos = bsum[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    const int64_t start = ((int64_t)blockIdx.x * RT_THREADS + threadIdx.x) * RT_ITEMS;
    // first owner boundary q*Rp at or after `start`
    int64_t q = (start + Rp - 1) / Rp;
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < RT_ITEMS; ++k) {
        const int64_t key = start + k;
        if (q < P && key == q * Rp) { edges[q] = pos; ++q; }
        if ((words[k >> 2] >> ((k & 3) * 8)) & 1u) {
            slotmap[key] = pos;
            uniq_rows[pos] = (int32_t)(key % Rp);
            ++pos;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) edges[P] = bsum[n_blk];
}

__global__ __launch_bounds__(256) void k_route_slot(const int32_t* __restrict__ x, const int32_t* __restrict__ offsets,
                                                    int64_t n, int f_in, int n_rows, int P, int Rp,
                                                    const int32_t* __restrict__ slotmap, uint8_t* __restrict__ flags,
                                                    int32_t* __restrict__ slot, int keep_flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i % f_in);
    const int key = route_key(x[i] + offsets[j], n_rows, P, Rp);
    slot[i] = slotmap[key];
    if (!keep_flags) flags[key] = 0;                                     // every set flag belongs to some lookup
}

extern "C" int64_t aread_route_ws_bytes(int64_t n_table_rows, int n_ranks) {
    if (n_table_rows <= 0 || n_ranks <= 0) return -1;
    const int64_t Rp = (n_table_rows + n_ranks - 1) / n_ranks;
    RouteWs L;
    if (route_layout(Rp * n_ranks, &L)) return -1;
    return L.total;
}

extern "C" int aread_route_build(const int32_t* x, int64_t B, int f_in, const int32_t* offsets, int64_t n_table_rows,
                                 int n_ranks, void* ws, int32_t* slot_out, int32_t* uniq_rows_out, int32_t* edges_out,
                                 int keep_flags, void* stream) {
    AR_CHECK_ARG(x && offsets && ws && slot_out && uniq_rows_out && edges_out, "aread_route_build: null pointer");
    AR_CHECK_ARG(B > 0 && f_in > 0 && n_ranks > 0 && n_ranks <= 1024, "aread_route_build: bad sizes");
    AR_CHECK_ARG(n_table_rows > 0 && n_table_rows < (1ll << 31), "aread_route_build: bad table size");
    AR_CHECK_ARG(((uintptr_t)ws & 255) == 0, "aread_route_build: workspace alignment");
    const int64_t Rp = (n_table_rows + n_ranks - 1) / n_ranks;
    RouteWs L;
    AR_CHECK_ARG(route_layout(Rp * n_ranks, &L) == 0, "aread_route_build: key space too large");
    AR_CHECK_ARG(L.n_blk < (1ll << 31), "aread_route_build: grid too large");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)ws;
    uint8_t* flags = (uint8_t*)(base + L.off_flags);
    int32_t* slotmap = (int32_t*)(base + L.off_slotmap);
    int32_t* bsum = (int32_t*)(base + L.off_bsum);
    const int64_t n = B * f_in;
    const unsigned g_n = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_route_mark, dim3(g_n), dim3(256), 0, st, x, offsets, n, f_in, (int)n_table_rows, n_ranks, (int)Rp,
                       flags);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_route_count, dim3((unsigned)L.n_blk), dim3(RT_THREADS), 0, st, flags, bsum);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_route_prefix, dim3(1), dim3(1024), 0, st, bsum, (int)L.n_blk);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_route_assign, dim3((unsigned)L.n_blk), dim3(RT_THREADS), 0, st, flags, bsum, L.n_keys, (int)L.n_blk,
                       n_ranks, (int)Rp, slotmap, uniq_rows_out, edges_out);
    AR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_route_slot, dim3(g_n), dim3(256), 0, st, x, offsets, n, f_in, (int)n_table_rows, n_ranks, (int)Rp,
                       slotmap, flags, slot_out, keep_flags);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}
