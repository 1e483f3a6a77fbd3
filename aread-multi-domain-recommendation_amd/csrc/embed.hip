// embed.hip -- sparse embedding lookup (gather + history pooling) and its deterministic
// scatter-add backward.  Replaces FeaturesEmbedding.forward (model/layer.py:160-183) and the
// autograd of nn.Embedding + view/mean/cat behind it.
//
// HBM-bound.  Forward: one "worker" of E/4 lanes per (output row, output field); every lane moves
// 16 B, a worker reads one whole table row (128 B at E=32) per lookup, a wave holds 8 workers, and
// the five history rows of a pooled field are issued back-to-back before the in-order fp32 sum
// (((r0+r1)+r2)+r3)+r4 and the true division by seq_len that make the result bit-identical to ATen.
// Backward: (row, slot) pairs are radix-sorted by table row, then summed by a two-level wavefront
// segmented reduction in a fixed order (no float atomics; hot rows such as the pad-alias row are
// split over workers and combined through LDS / a boundary list).
#include <cstring>
#include "common.h"
#include <rocprim/rocprim.hpp>

#define EMB_THREADS 256
#define MAX_SEQ 8

__global__ __launch_bounds__(EMB_THREADS) void k_embed_fwd(
    const int32_t* __restrict__ x, const int32_t* __restrict__ offsets, const float4* __restrict__ table,
    const int32_t* __restrict__ row_sample, float4* __restrict__ out, int32_t* __restrict__ bag_out,
    int B, int n_rows_out, int f_in, int f_out, int n_oh, int S, int pool, int e4) {
    const int64_t gid = (int64_t)blockIdx.x * EMB_THREADS + threadIdx.x;
    const int64_t item = gid / e4;
    const int c4 = (int)(gid - item * e4);
    const int64_t p = item / f_out;
    const int fo = (int)(item - p * f_out);
    if (p >= n_rows_out) return;
    const int b = row_sample ? row_sample[p] : (int)p;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b >= 0) {
        const int32_t* xr = x + (int64_t)b * f_in;
        if (fo < n_oh) {
            const int32_t g = xr[fo] + offsets[fo];               // int32 index bag (layer.py:165)
            acc = table[(int64_t)g * e4 + c4];
            if (bag_out && c4 == 0) bag_out[(int64_t)b * f_in + fo] = g;
        } else {
            const int j0 = n_oh + (fo - n_oh) * S;
            float4 v[MAX_SEQ];
#pragma unroll
            for (int s = 0; s < MAX_SEQ; ++s) {
                if (s < S) {
                    const int32_t g = xr[j0 + s] + offsets[j0 + s];
                    v[s] = table[(int64_t)g * e4 + c4];
                    if (bag_out && c4 == 0) bag_out[(int64_t)b * f_in + j0 + s] = g;
                }
            }
            acc = v[0];
#pragma unroll
            for (int s = 1; s < MAX_SEQ; ++s) {
                if (s < S) { acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w; }
            }
            if (pool == 2) {
                const float d = (float)S;                          // true division: matches torch.mean
                acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
            }
        }
    }
    out[((int64_t)p * f_out + fo) * e4 + c4] = acc;
}

extern "C" int aread_embed_fwd(const int32_t* x, int64_t B, int f_in, const int32_t* offsets, const float* table,
                               int64_t n_table_rows, int E, int n_onehot, int n_mh_fields, int seq_len, int pool,
                               const int32_t* row_sample, int64_t n_rows_out, float* out, int32_t* bag_out,
                               void* stream) {
    AR_CHECK_ARG(x && offsets && table && out, "aread_embed_fwd: null pointer");
    AR_CHECK_ARG(E > 0 && E % 4 == 0 && E <= 1024, "aread_embed_fwd: E=%d must be a multiple of 4", E);
    AR_CHECK_ARG(pool >= 0 && pool <= 2, "aread_embed_fwd: pool=%d", pool);
    AR_CHECK_ARG(B > 0 && n_table_rows > 0, "aread_embed_fwd: empty input");
    if (pool == 0) { n_onehot = f_in; n_mh_fields = 0; seq_len = 1; }
    AR_CHECK_ARG(seq_len >= 1 && seq_len <= MAX_SEQ, "aread_embed_fwd: seq_len=%d not in [1,%d]", seq_len, MAX_SEQ);
    AR_CHECK_ARG(n_onehot + n_mh_fields * seq_len == f_in, "aread_embed_fwd: %d one-hot + %d x %d history != f_in=%d",
                 n_onehot, n_mh_fields, seq_len, f_in);
    AR_CHECK_ARG(((uintptr_t)table & 15) == 0 && ((uintptr_t)out & 15) == 0, "aread_embed_fwd: 16-byte alignment");
    AR_CHECK_ARG(row_sample != nullptr || n_rows_out == B, "aread_embed_fwd: n_rows_out=%lld != B=%lld without a plan",
                 (long long)n_rows_out, (long long)B);
    const int f_out = n_onehot + n_mh_fields;
    const int e4 = E / 4;
    const int64_t threads = n_rows_out * f_out * e4;
    AR_CHECK_ARG(threads / EMB_THREADS < (1ll << 31), "aread_embed_fwd: grid too large");
    hipLaunchKernelGGL(k_embed_fwd, dim3(cdiv(threads, EMB_THREADS)), dim3(EMB_THREADS), 0, (hipStream_t)stream, x,
                       offsets, (const float4*)table, row_sample, (float4*)out, bag_out, (int)B, (int)n_rows_out, f_in,
                       f_out, n_onehot, seq_len, pool, e4);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

// ================================================================================================
// backward
// ================================================================================================
__global__ __launch_bounds__(256) void k_embed_bwd_keys(const int32_t* __restrict__ x,
                                                        const int32_t* __restrict__ offsets,
                                                        const int32_t* __restrict__ sample_row,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        int64_t n, int f_in, int f_out, int n_oh, int S, int pool) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / f_in;
    const int j = (int)(i - b * f_in);
    const int fo = j < n_oh ? j : n_oh + (j - n_oh) / S;
    const int64_t p = sample_row ? sample_row[b] : b;
    keys[i] = (uint32_t)(x[i] + offsets[j]);
    vals[i] = (uint32_t)(p * f_out + fo) | ((j >= n_oh && pool == 2) ? 0x80000000u : 0u);
}

static __device__ __forceinline__ void f4_add(float4& a, const float4& b) {
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// Shared skeleton of both reduction levels.  Each worker (e4 lanes) walks `per_worker` consecutive
// sorted entries; complete runs inside a worker are added straight to the table gradient, the first
// and last run of every worker go through LDS and are combined per block in entry order.  Runs that
// touch the block's first/last key may continue in a neighbour block: level 1 hands them to the
// boundary list, level 2 (one block, sees everything) writes them out.
template <bool LEVEL1, int THREADS>
__device__ __forceinline__ void segreduce_block(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                const float4* __restrict__ src, int64_t n, int64_t blk_begin,
                                                int64_t blk_end, int per_worker, int e4, float inv_div,
                                                float4* __restrict__ grad, int32_t* __restrict__ bnd_keys,
                                                float4* __restrict__ bnd_vals, int32_t* s_key, float4* s_val) {
    const int tid = threadIdx.x;
    const int w = tid / e4, c4 = tid - w * e4;
    const int n_workers = THREADS / e4;
    const bool live = w < n_workers;
    int32_t fk = -1, lk = -1;
    float4 fv = make_float4(0.f, 0.f, 0.f, 0.f), lv = fv;
    if (live) {
        const int64_t i0 = blk_begin + (int64_t)w * per_worker;
        int64_t i1 = i0 + per_worker;
        if (i1 > blk_end) i1 = blk_end;
        int32_t cur = -1;
        int nseg = 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = i0; i < i1; ++i) {
            const int32_t k = (int32_t)keys[i];
            if (k < 0) continue;                                  // level 2: unused boundary slot
            if (k != cur) {
                if (cur >= 0) {
                    if (nseg == 0) { fk = cur; fv = acc; }
                    else f4_add(grad[(int64_t)cur * e4 + c4], acc);   // run complete inside this worker
                    ++nseg;
                }
                cur = k;
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 g;
            if (LEVEL1) {
                const uint32_t v = vals[i];
                g = src[(int64_t)(v & 0x7FFFFFFFu) * e4 + c4];
                if (v >> 31) { g.x *= inv_div; g.y *= inv_div; g.z *= inv_div; g.w *= inv_div; }
            } else {
                g = src[i * e4 + c4];
            }
            f4_add(acc, g);
        }
        if (cur >= 0) {
            if (nseg == 0) { fk = cur; fv = acc; }
            else { lk = cur; lv = acc; }
        }
        if (c4 == 0) { s_key[2 * w] = fk; s_key[2 * w + 1] = lk; }
        s_val[(2 * w) * e4 + c4] = fv;
        s_val[(2 * w + 1) * e4 + c4] = lv;
    }
    __syncthreads();
    if (!live) return;
    int32_t first_key = -1, last_key = -1;
    if (LEVEL1) {
        first_key = (int32_t)keys[blk_begin];
        last_key = (int32_t)keys[blk_end - 1];
    }
    const int n_ent = 2 * n_workers;
    for (int ei = 2 * w; ei < 2 * w + 2; ++ei) {
        const int32_t k = s_key[ei];
        if (k < 0) continue;
        int prev = ei - 1;
        while (prev >= 0 && s_key[prev] < 0) --prev;
        if (prev >= 0 && s_key[prev] == k) continue;              // not the head of its run
        float4 acc = s_val[ei * e4 + c4];
        for (int j = ei + 1; j < n_ent; ++j) {
            const int32_t kj = s_key[j];
            if (kj < 0) continue;
            if (kj != k) break;
            f4_add(acc, s_val[j * e4 + c4]);
        }
        if (LEVEL1 && k == first_key) {
            if (c4 == 0) bnd_keys[2 * blockIdx.x] = k;
            bnd_vals[(int64_t)(2 * blockIdx.x) * e4 + c4] = acc;
        } else if (LEVEL1 && k == last_key) {
            if (c4 == 0) bnd_keys[2 * blockIdx.x + 1] = k;
            bnd_vals[(int64_t)(2 * blockIdx.x + 1) * e4 + c4] = acc;
        } else {
            f4_add(grad[(int64_t)k * e4 + c4], acc);
        }
    }
}

#define SR_THREADS 256
#define SR_PER_WORKER 16
#define SR2_THREADS 1024

__global__ __launch_bounds__(SR_THREADS) void k_embed_bwd_reduce1(const uint32_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ vals,
                                                                  const float4* __restrict__ dout, int64_t n, int e4,
                                                                  float inv_div, float4* __restrict__ grad,
                                                                  int32_t* __restrict__ bnd_keys,
                                                                  float4* __restrict__ bnd_vals) {
    extern __shared__ float4 smem[];
    const int n_workers = SR_THREADS / e4;
    float4* s_val = smem;
    int32_t* s_key = (int32_t*)(smem + 2 * n_workers * e4);
    const int64_t per_block = (int64_t)n_workers * SR_PER_WORKER;
    const int64_t b0 = (int64_t)blockIdx.x * per_block;
    int64_t b1 = b0 + per_block;
    if (b1 > n) b1 = n;
    if (threadIdx.x < 2) bnd_keys[2 * blockIdx.x + threadIdx.x] = -1;
    __syncthreads();
    segreduce_block<true, SR_THREADS>(keys, vals, dout, n, b0, b1, SR_PER_WORKER, e4, inv_div, grad, bnd_keys, bnd_vals,
                                      s_key, s_val);
}

__global__ __launch_bounds__(SR2_THREADS) void k_embed_bwd_reduce2(const int32_t* __restrict__ bnd_keys,
                                                                   const float4* __restrict__ bnd_vals, int64_t n_bnd,
                                                                   int e4, float4* __restrict__ grad) {
    extern __shared__ float4 smem[];
    const int n_workers = SR2_THREADS / e4;
    float4* s_val = smem;
    int32_t* s_key = (int32_t*)(smem + 2 * n_workers * e4);
    const int per_worker = (int)((n_bnd + n_workers - 1) / n_workers);
    segreduce_block<false, SR2_THREADS>((const uint32_t*)bnd_keys, nullptr, bnd_vals, n_bnd, 0, n_bnd, per_worker, e4,
                                        1.f, grad, nullptr, nullptr, s_key, s_val);
}

static inline int key_bits(int64_t n_rows) {
    int b = 1;
    while (b < 32 && (1ll << b) < n_rows) ++b;
    return b;
}
static inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }

struct EmbBwdWs {
    int64_t n, n_blk, off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_bkeys, off_bvals, off_temp, temp_bytes, total;
};
static int emb_bwd_layout(int64_t B, int f_in, int E, EmbBwdWs* L) {
    const int e4 = E / 4;
    L->n = B * f_in;
    const int64_t per_block = (int64_t)(SR_THREADS / e4) * SR_PER_WORKER;
    L->n_blk = (L->n + per_block - 1) / per_block;
    int64_t o = 0;
    L->off_keys_a = o; o = align256(o + L->n * 4);
    L->off_keys_b = o; o = align256(o + L->n * 4);
    L->off_vals_a = o; o = align256(o + L->n * 4);
    L->off_vals_b = o; o = align256(o + L->n * 4);
    L->off_bkeys = o;  o = align256(o + L->n_blk * 2 * 4);
    L->off_bvals = o;  o = align256(o + L->n_blk * 2 * (int64_t)E * 4);
    size_t tb = 0;
    hipError_t e = rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, uint32_t*, uint32_t*>(
        nullptr, tb, nullptr, nullptr, nullptr, nullptr, (size_t)L->n, 0u, 32u, (hipStream_t)0);
    if (e != hipSuccess) return -1;
    L->temp_bytes = (int64_t)tb;
    L->off_temp = o; o = align256(o + L->temp_bytes);
    L->total = o;
    return 0;
}

extern "C" int64_t aread_embed_bwd_ws_bytes(int64_t B, int f_in, int E) {
    if (B <= 0 || f_in <= 0 || E <= 0 || E % 4) return -1;
    EmbBwdWs L;
    if (emb_bwd_layout(B, f_in, E, &L)) return -1;
    return L.total;
}

// phase 1 (depends only on the ids and the row map): build (table row, dout slot) pairs and sort them by row
extern "C" int aread_embed_bwd_sort(const int32_t* x, int64_t B, int f_in, const int32_t* offsets, int64_t n_table_rows,
                                    int E, int n_onehot, int n_mh_fields, int seq_len, int pool, const int32_t* sample_row,
                                    void* ws, void* stream) {
    AR_CHECK_ARG(x && offsets && ws, "aread_embed_bwd_sort: null pointer");
    AR_CHECK_ARG(E > 0 && E % 4 == 0 && E <= 256, "aread_embed_bwd: E=%d must be a multiple of 4, <= 256", E);
    AR_CHECK_ARG(pool >= 0 && pool <= 2, "aread_embed_bwd: pool=%d", pool);
    if (pool == 0) { n_onehot = f_in; n_mh_fields = 0; seq_len = 1; }
    AR_CHECK_ARG(n_onehot + n_mh_fields * seq_len == f_in, "aread_embed_bwd: field layout does not add up to f_in=%d", f_in);
    AR_CHECK_ARG(n_table_rows > 0 && n_table_rows < (1ll << 31), "aread_embed_bwd: bad table size");
    AR_CHECK_ARG(((uintptr_t)ws & 255) == 0, "aread_embed_bwd: workspace alignment");
    hipStream_t st = (hipStream_t)stream;
    EmbBwdWs L;
    AR_CHECK_ARG(emb_bwd_layout(B, f_in, E, &L) == 0, "aread_embed_bwd: workspace layout failed");
    char* base = (char*)ws;
    uint32_t* keys_a = (uint32_t*)(base + L.off_keys_a);
    uint32_t* keys_b = (uint32_t*)(base + L.off_keys_b);
    uint32_t* vals_a = (uint32_t*)(base + L.off_vals_a);
    uint32_t* vals_b = (uint32_t*)(base + L.off_vals_b);
    const int f_out = n_onehot + n_mh_fields;
    hipLaunchKernelGGL(k_embed_bwd_keys, dim3(cdiv(L.n, 256)), dim3(256), 0, st, x, offsets, sample_row, keys_a, vals_a,
                       L.n, f_in, f_out, n_onehot, seq_len, pool);
    AR_LAUNCH_CHECK();
    size_t tb = (size_t)L.temp_bytes;
    AR_HIP((rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, uint32_t*, uint32_t*>(
        base + L.off_temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)L.n, 0u, (unsigned)key_bits(n_table_rows), st)));
    return AREAD_OK;
}

// phase 2: segmented reduction of the sorted pairs into table_grad
extern "C" int aread_embed_bwd_reduce(int64_t B, int f_in, int E, int seq_len, const float* dout, float* table_grad, void* ws,
                                      void* stream) {
    AR_CHECK_ARG(dout && table_grad && ws, "aread_embed_bwd_reduce: null pointer");
    AR_CHECK_ARG(E > 0 && E % 4 == 0 && E <= 256 && seq_len >= 1, "aread_embed_bwd_reduce: bad E/seq_len");
    AR_CHECK_ARG(((uintptr_t)dout & 15) == 0 && ((uintptr_t)table_grad & 15) == 0 && ((uintptr_t)ws & 255) == 0,
                 "aread_embed_bwd_reduce: alignment");
    hipStream_t st = (hipStream_t)stream;
    EmbBwdWs L;
    AR_CHECK_ARG(emb_bwd_layout(B, f_in, E, &L) == 0, "aread_embed_bwd: workspace layout failed");
    char* base = (char*)ws;
    uint32_t* keys_b = (uint32_t*)(base + L.off_keys_b);
    uint32_t* vals_b = (uint32_t*)(base + L.off_vals_b);
    int32_t* bkeys = (int32_t*)(base + L.off_bkeys);
    float4* bvals = (float4*)(base + L.off_bvals);
    const int e4 = E / 4;
    const int nw1 = SR_THREADS / e4;
    const size_t lds1 = (size_t)2 * nw1 * e4 * 16 + (size_t)2 * nw1 * 4;
    hipLaunchKernelGGL(k_embed_bwd_reduce1, dim3((unsigned)L.n_blk), dim3(SR_THREADS), lds1, st, keys_b, vals_b,
                       (const float4*)dout, L.n, e4, 1.0f / (float)seq_len, (float4*)table_grad, bkeys, bvals);
    AR_LAUNCH_CHECK();
    const int nw2 = SR2_THREADS / e4;
    const size_t lds2 = (size_t)2 * nw2 * e4 * 16 + (size_t)2 * nw2 * 4;
    hipLaunchKernelGGL(k_embed_bwd_reduce2, dim3(1), dim3(SR2_THREADS), lds2, st, bkeys, (const float4*)bvals,
                       2 * L.n_blk, e4, (float4*)table_grad);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
}

extern "C" int aread_embed_bwd(const int32_t* x, int64_t B, int f_in, const int32_t* offsets, int64_t n_table_rows,
                               int E, int n_onehot, int n_mh_fields, int seq_len, int pool, const int32_t* sample_row,
                               const float* dout, float* table_grad, void* ws, void* stream) {
    int st = aread_embed_bwd_sort(x, B, f_in, offsets, n_table_rows, E, n_onehot, n_mh_fields, seq_len, pool, sample_row, ws,
                                  stream);
    if (st != AREAD_OK) return st;
    if (pool == 0) seq_len = 1;
    return aread_embed_bwd_reduce(B, f_in, E, seq_len, dout, table_grad, ws, stream);
}
