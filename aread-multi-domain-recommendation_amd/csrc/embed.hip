// embed.hip -- sparse embedding lookup (gather + history pooling) and its deterministic
// scatter-add backward.  Replaces FeaturesEmbedding.forward (model/layer.py:160-183) and the
// autograd of nn.Embedding + view/mean/cat behind it.
//
// HBM / latency bound.  Forward: a "worker" of E/4 lanes (16 B per lane: one whole 128-byte table row per lookup at
// E = 32) owns EMB_R consecutive output rows of ONE output field; workers are numbered field-major, so a wave's workers
// all take the same branch (one-hot or pooled) and every lane has EMB_R (one-hot) or EMB_R x seq_len (pooled)
// independent 16-byte loads in flight before the first use.  The history rows of a pooled field are summed in slot
// order (((r0+r1)+r2)+r3)+r4 and truly divided by seq_len: bit-identical to ATen.
// Backward: (table row, slot) pairs are sorted by table row with a hand-written LSD radix sort (8-bit digits, wave-level
// multisplit by ballots: stable, no atomics in the ranking), then summed by a segmented reduction in a fixed order
// (no float atomics, bitwise reproducible): per worker 16 sorted entries in registers (segmented prefix, all gathers and
// all read-modify-writes of finished rows in flight together), per workgroup a log-step segmented scan through LDS,
// across workgroups a boundary list reduced by the same kernel.  Hot rows (the pad-alias row takes ~46 % of the lookups)
// cost log steps, not a serial chain.
#include <cstring>
#include "common.h"

#define EMB_THREADS 256
typedef float v4f __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct arrays went to scratch)
#define MAX_SEQ 8

#define EMB_R 4
// SP: compile-time seq_len of the pooled fields (0 = no pooled field is ever taken)
template <int SP>
__global__ __launch_bounds__(EMB_THREADS) void k_embed_fwd(
    const int32_t* __restrict__ x, const int32_t* __restrict__ offsets, const v4f* __restrict__ table,
    const int32_t* __restrict__ row_sample, v4f* __restrict__ out, int32_t* __restrict__ bag_out,
    int B, int n_rows_out, int f_in, int f_out, int n_oh, int pool, int e4, int n_rg) {
    const int64_t gid = (int64_t)blockIdx.x * EMB_THREADS + threadIdx.x;
    const int64_t item = gid / e4;
    const int c4 = (int)(gid - item * e4);
    const int fo = (int)(item / n_rg);
    const int rg = (int)(item - (int64_t)fo * n_rg);
    if (fo >= f_out) return;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    int b[EMB_R];
#pragma unroll
    for (int k = 0; k < EMB_R; ++k) {
        const int p = rg * EMB_R + k;
        b[k] = p < n_rows_out ? (row_sample ? row_sample[p] : p) : -1;
    }
    if (fo < n_oh) {
        const int32_t off = offsets[fo];
        int32_t g[EMB_R];
#pragma unroll
        for (int k = 0; k < EMB_R; ++k) g[k] = b[k] >= 0 ? x[(int64_t)b[k] * f_in + fo] + off : -1;   // int32 index bag (layer.py:165)
        v4f v[EMB_R];
#pragma unroll
        for (int k = 0; k < EMB_R; ++k) v[k] = g[k] >= 0 ? table[(int64_t)g[k] * e4 + c4] : zero;
#pragma unroll
        for (int k = 0; k < EMB_R; ++k) {
            const int p = rg * EMB_R + k;
            if (p < n_rows_out) out[((int64_t)p * f_out + fo) * e4 + c4] = v[k];
            if (bag_out && c4 == 0 && b[k] >= 0) bag_out[(int64_t)b[k] * f_in + fo] = g[k];
        }
    } else if (SP > 0) {
        constexpr int S = SP > 0 ? SP : 1;
        const int j0 = n_oh + (fo - n_oh) * S;
        int32_t g[EMB_R][S];
#pragma unroll
        for (int k = 0; k < EMB_R; ++k)
#pragma unroll
            for (int s = 0; s < S; ++s) g[k][s] = b[k] >= 0 ? x[(int64_t)b[k] * f_in + j0 + s] + offsets[j0 + s] : -1;
        v4f v[EMB_R][S];
#pragma unroll
        for (int k = 0; k < EMB_R; ++k)
#pragma unroll
            for (int s = 0; s < S; ++s) v[k][s] = g[k][s] >= 0 ? table[(int64_t)g[k][s] * e4 + c4] : zero;
#pragma unroll
        for (int k = 0; k < EMB_R; ++k) {
            v4f acc = v[k][0];
#pragma unroll
            for (int s = 1; s < S; ++s) acc += v[k][s];
            if (pool == 2) acc /= (float)S;                         // true division: matches torch.mean
            const int p = rg * EMB_R + k;
            if (p < n_rows_out) out[((int64_t)p * f_out + fo) * e4 + c4] = acc;
            if (bag_out && c4 == 0 && b[k] >= 0)
#pragma unroll
                for (int s = 0; s < S; ++s) bag_out[(int64_t)b[k] * f_in + j0 + s] = g[k][s];
        }
    }
}

// ---- measurement only: the most favourable form of the gather's memory traffic -------------------------------------------
// n_read random 128-byte-class rows (E floats each) read through PRE-RESOLVED row indices and n_write rows written as a stream:
// no id -> row dependency beyond one index load per row, no field / pooling logic, every load of a thread in flight before the
// first use (RD rows per thread), 16 bytes per lane.  What k_embed_fwd could reach if its id decoding, history pooling and
// plan lookup were free: bench.py quotes the gather against this beside the 8 TB/s figure (gather_roofline.achievable_us).
template <int RD>
__global__ __launch_bounds__(256) void k_gather_roof(const int32_t* __restrict__ rows, int64_t n_read, const v4f* __restrict__ table,
                                                     v4f* __restrict__ out, int64_t n_write, int e4) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t item = tid / e4;
    const int c4 = (int)(tid - item * e4);
    const int64_t r0 = item * RD;
    if (r0 >= n_read) return;
    int32_t idx[RD];
#pragma unroll
    for (int j = 0; j < RD; ++j) idx[j] = r0 + j < n_read ? rows[r0 + j] : rows[r0];
    v4f v[RD];
#pragma unroll
    for (int j = 0; j < RD; ++j) v[j] = table[(int64_t)idx[j] * e4 + c4];
    // the same rows-in : rows-out ratio as the real gather (n_write / n_read): sums of consecutive reads become one output row
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < RD; ++j) {
        acc += v[j];
        const int64_t r = r0 + j;
        if (r < n_read && ((r + 1) * n_write / n_read) != (r * n_write / n_read)) {
            out[(r * n_write / n_read) * e4 + c4] = acc;
            acc = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
}
extern "C" int aread_debug_gather_roof(const int32_t* rows, int64_t n_read, const float* table, int E, float* out, int64_t n_write,
                                       void* stream) {
    AR_CHECK_ARG(rows && table && out && n_read > 0 && n_write > 0 && n_write <= n_read && E > 0 && E % 4 == 0, "aread_debug_gather_roof: bad arguments");
    constexpr int RD = 8;
    const int e4 = E / 4;
    const int64_t threads = (n_read + RD - 1) / RD * e4;
    hipLaunchKernelGGL(k_gather_roof<RD>, dim3(cdiv(threads, 256)), dim3(256), 0, (hipStream_t)stream, rows, n_read, (const v4f*)table,
                       (v4f*)out, n_write, e4);
    AR_LAUNCH_CHECK();
    return AREAD_OK;
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
    const int n_rg = (int)((n_rows_out + EMB_R - 1) / EMB_R);
    const int64_t threads = (int64_t)n_rg * f_out * e4;
    AR_CHECK_ARG(threads / EMB_THREADS < (1ll << 31) && n_rows_out < (1ll << 31), "aread_embed_fwd: grid too large");
    const dim3 grid(cdiv(threads, EMB_THREADS)), block(EMB_THREADS);
    const hipStream_t st = (hipStream_t)stream;
#define EMB_FWD(SP)                                                                                                      \
    hipLaunchKernelGGL(k_embed_fwd<SP>, grid, block, 0, st, x, offsets, (const v4f*)table, row_sample, (v4f*)out,        \
                       bag_out, (int)B, (int)n_rows_out, f_in, f_out, n_onehot, pool, e4, n_rg)
    switch (n_mh_fields > 0 ? seq_len : 0) {
        case 0: EMB_FWD(0); break;
        case 1: EMB_FWD(1); break;
        case 2: EMB_FWD(2); break;
        case 3: EMB_FWD(3); break;
        case 4: EMB_FWD(4); break;
        case 5: EMB_FWD(5); break;
        case 6: EMB_FWD(6); break;
        case 7: EMB_FWD(7); break;
        default: EMB_FWD(8); break;
    }
#undef EMB_FWD
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


// ------------------------------------------------------------------------------------------------
// LSD radix sort of (key, value) pairs, 8-bit digits.  Per pass: per-block digit histogram -> exclusive scan over
// [digit][block] -> stable scatter.  A block covers RS_PER_BLOCK consecutive pairs, wave w of it RS_ITEMS rounds of 64
// consecutive pairs; the rank of a pair inside its (wave, round) comes from a ballot-built match mask, the running
// per-wave digit counters live in LDS rows private to the wave (no atomics, nothing order-dependent).
// ------------------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_PER_BLOCK (RS_THREADS * RS_ITEMS)
#define RS_BITS 8
#define RS_BUCKETS (1 << RS_BITS)

__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, int64_t n, int shift, int nb,
                                                        int32_t* __restrict__ hist) {
    __shared__ int s_h[RS_BUCKETS];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_PER_BLOCK;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = base + i * RS_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&s_h[(keys[idx] >> shift) & (RS_BUCKETS - 1)], 1);      // integer counts: order-free
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nb + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive scan of `total` ints in place, one workgroup of 1024 threads.  Thread t owns the contiguous elements
// [t*per, (t+1)*per): up to RS_SCAN_REG of them are read with independent loads into registers (a dependent loop would pay
// one memory latency per element); the 1024 thread sums are scanned with wave shuffles + one 16-entry LDS pass.
#define RS_SCAN_REG 24
__global__ __launch_bounds__(1024) void k_rs_scan(int32_t* __restrict__ h, int64_t total) {
    __shared__ int s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (total + 1023) / 1024, i0 = (int64_t)tid * per;
    const int64_t i1 = i0 + per < total ? i0 + per : total;
    int sum = 0;
    int reg[RS_SCAN_REG];
    const bool in_reg = per <= RS_SCAN_REG;
    if (in_reg) {
#pragma unroll
        for (int j = 0; j < RS_SCAN_REG; ++j) reg[j] = i0 + j < i1 ? h[i0 + j] : 0;
#pragma unroll
        for (int j = 0; j < RS_SCAN_REG; ++j) sum += reg[j];
    } else {
        for (int64_t i = i0; i < i1; ++i) sum += h[i];
    }
    int incl = sum;                                        // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) base += w < wave ? s_wave[w] : 0;
    int run = base + incl - sum;
    if (in_reg) {
#pragma unroll
        for (int j = 0; j < RS_SCAN_REG; ++j) {
            if (i0 + j < i1) h[i0 + j] = run;
            run += reg[j];
        }
    } else {
        for (int64_t i = i0; i < i1; ++i) { const int c = h[i]; h[i] = run; run += c; }
    }
}

__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                           int64_t n, int shift, int nb, const int32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ int s_cnt[RS_THREADS / WAVE][RS_BUCKETS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < RS_THREADS / WAVE; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_PER_BLOCK + (int64_t)wave * (RS_ITEMS * WAVE);
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t k[RS_ITEMS], v[RS_ITEMS];
    int pos[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = base + i * WAVE + lane;
        const bool ok = idx < n;
        k[i] = ok ? keys[idx] : 0u;
        v[i] = ok ? vals[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const bool ok = base + i * WAVE + lane < n;
        const int d = (int)((k[i] >> shift) & (RS_BUCKETS - 1));
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int bit = 0; bit < RS_BITS; ++bit) {
            const bool set = (d >> bit) & 1;
            const unsigned long long bm = __ballot(set);
            m &= set ? bm : ~bm;
        }
        const int rank = __popcll(m & lt);
        int old = 0;
        if (ok && rank == 0) {                             // the lowest lane of every digit group keeps the wave's counter
            old = s_cnt[wave][d];
            s_cnt[wave][d] = old + __popcll(m);
        }
        const int leader = ok ? __ffsll((long long)m) - 1 : 0;
        old = __shfl(old, leader);
        pos[i] = old + rank;
    }
    __syncthreads();
    {   // digit d = tid: block offset from the scanned histogram + exclusive prefix over the waves
        int run = hist[(int64_t)tid * nb + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_THREADS / WAVE; ++w) { const int c = s_cnt[w][tid]; s_cnt[w][tid] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        if (base + i * WAVE + lane < n) {
            const int d = (int)((k[i] >> shift) & (RS_BUCKETS - 1));
            const int64_t dst = (int64_t)s_cnt[wave][d] + pos[i];
            keys_out[dst] = k[i];
            vals_out[dst] = v[i];
        }
    }
}

static inline int key_bits(int64_t n_rows) {
    int b = 1;
    while (b < 32 && (1ll << b) < n_rows) ++b;
    return b;
}
// number of passes: odd, so that the sorted pairs always end in the "b" buffers
static inline int rs_passes(int64_t n_rows) {
    int p = (key_bits(n_rows) + RS_BITS - 1) / RS_BITS;
    if ((p & 1) == 0) ++p;
    return p;
}

// ------------------------------------------------------------------------------------------------
// segmented reduction of sorted (key, 16-byte x e4 value) entries.
//   GATHER: entry i's value is dout row (vals[i] & 0x7fffffff), scaled by inv_div when bit 31 is set (level 1);
//           otherwise the value is src[i] (boundary lists of a previous level).
//   FINAL : one workgroup sees every entry: all runs go to the table gradient; otherwise the runs that touch the
//           workgroup's first / last key go to the boundary list (2 entries per workgroup, always both written, sorted).
// ------------------------------------------------------------------------------------------------
#ifndef SR_PW
#define SR_PW 16                       // entries per worker
#endif
#define SR_SENT 0x7fffffff             // key of "no entry" (table rows are < 2^31 - 1)

template <bool GATHER, bool FINAL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_segreduce(const int32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                       const v4f* __restrict__ src, const v4f* __restrict__ src2, int64_t n, int e4, float inv_div,
                                                       v4f* __restrict__ grad, int32_t* __restrict__ bnd_keys,
                                                       v4f* __restrict__ bnd_vals) {
    extern __shared__ v4f smem[];
    const int tid = threadIdx.x;
    const int w = tid / e4, c4 = tid - w * e4;
    const int nw = THREADS / e4, n_ent = 2 * nw;
    v4f* s_val = smem;                                   // [n_ent][e4]
    int32_t* s_key = (int32_t*)(smem + (size_t)n_ent * e4); // [n_ent]
    const bool live = w < nw;
    const int64_t blk_begin = (int64_t)blockIdx.x * nw * SR_PW;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    int32_t first_key = SR_SENT, last_key = SR_SENT;
    if (!FINAL) {
        int64_t blk_end = blk_begin + (int64_t)nw * SR_PW;
        if (blk_end > n) blk_end = n;
        first_key = keys[blk_begin];
        last_key = keys[blk_end - 1];
        if (w == 0) {                                       // both boundary entries always exist: (key, 0) unless a run total replaces it below
            if (c4 == 0) { bnd_keys[2 * blockIdx.x] = first_key; bnd_keys[2 * blockIdx.x + 1] = last_key; }
            bnd_vals[(size_t)(2 * blockIdx.x) * e4 + c4] = zero;
            bnd_vals[(size_t)(2 * blockIdx.x + 1) * e4 + c4] = zero;
        }
    }
    if (live) {
        const int64_t i0 = blk_begin + (int64_t)w * SR_PW;
        int32_t k[SR_PW];
        v4f g[SR_PW];
        if (GATHER) {
            uint32_t v[SR_PW];
#pragma unroll
            for (int j = 0; j < SR_PW; ++j) {
                const bool ok = i0 + j < n;
                k[j] = ok ? keys[i0 + j] : SR_SENT;
                v[j] = ok ? vals[i0 + j] : 0u;
            }
#pragma unroll
            for (int j = 0; j < SR_PW; ++j) {
                g[j] = k[j] != SR_SENT ? src[(int64_t)(v[j] & 0x7FFFFFFFu) * e4 + c4] : zero;
                if (src2 && k[j] != SR_SENT) g[j] += src2[(int64_t)(v[j] & 0x7FFFFFFFu) * e4 + c4];   // second addend of dL/de (aread_call.de_rw)
                if (v[j] >> 31) g[j] *= inv_div;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SR_PW; ++j) {
                const bool ok = i0 + j < n;
                k[j] = ok ? keys[i0 + j] : SR_SENT;
                g[j] = ok ? src[(i0 + j) * e4 + c4] : zero;
            }
        }
        // segmented inclusive prefix in registers: g[j] = sum of the run of k[j] up to j
#pragma unroll
        for (int j = 1; j < SR_PW; ++j)
            if (k[j] == k[j - 1]) g[j] += g[j - 1];
        // runs that begin and end inside this worker (not its first, not its last run): straight to the gradient;
        // the loads of all of them are issued before the first add
        v4f o[SR_PW - 1];
        bool interior[SR_PW - 1];
#pragma unroll
        for (int j = 0; j < SR_PW - 1; ++j) {
            interior[j] = k[j] != k[j + 1] && k[j] != k[0] && k[j] != k[SR_PW - 1];
            o[j] = interior[j] ? grad[(int64_t)k[j] * e4 + c4] : zero;
        }
#pragma unroll
        for (int j = 0; j < SR_PW - 1; ++j)
            if (interior[j]) grad[(int64_t)k[j] * e4 + c4] = o[j] + g[j];
        // first run (may continue in the previous worker) and last run (may continue in the next one)
        v4f fv = g[SR_PW - 1];
#pragma unroll
        for (int j = SR_PW - 2; j >= 0; --j)
            if (k[j] == k[0] && k[j + 1] != k[0]) fv = g[j];
        const bool single = k[SR_PW - 1] == k[0];
        if (c4 == 0) { s_key[2 * w] = k[0]; s_key[2 * w + 1] = k[SR_PW - 1]; }
        s_val[(size_t)(2 * w) * e4 + c4] = fv;
        s_val[(size_t)(2 * w + 1) * e4 + c4] = single ? zero : g[SR_PW - 1];
    }
    __syncthreads();
    // log-step segmented inclusive scan over the n_ent sorted entries (entry e adds entry e-d when the keys agree)
    for (int d = 1; d < n_ent; d <<= 1) {
        v4f a0 = zero, a1 = zero;
        if (live) {
            const int e0 = 2 * w, e1 = 2 * w + 1;
            if (e0 >= d && s_key[e0 - d] == s_key[e0]) a0 = s_val[(size_t)(e0 - d) * e4 + c4];
            if (e1 >= d && s_key[e1 - d] == s_key[e1]) a1 = s_val[(size_t)(e1 - d) * e4 + c4];
        }
        __syncthreads();
        if (live) {
            s_val[(size_t)(2 * w) * e4 + c4] += a0;
            s_val[(size_t)(2 * w + 1) * e4 + c4] += a1;
        }
        __syncthreads();
    }
    if (!live) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 2 * w + q;
        const int32_t key = s_key[e];
        if (key == SR_SENT) continue;
        if (e + 1 < n_ent && s_key[e + 1] == key) continue;       // not the last entry of its run
        const v4f tot = s_val[(size_t)e * e4 + c4];
        if (!FINAL && key == first_key) {
            bnd_vals[(size_t)(2 * blockIdx.x) * e4 + c4] = tot;
        } else if (!FINAL && key == last_key) {
            bnd_vals[(size_t)(2 * blockIdx.x + 1) * e4 + c4] = tot;
        } else {
            grad[(int64_t)key * e4 + c4] += tot;
        }
    }
}

#define SR_THREADS 256
#define SRF_THREADS 512
static inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }

struct EmbBwdWs {
    int64_t n, n_blk, n_blk2, rs_nb, off_keys_a, off_keys_b, off_vals_a, off_vals_b, off_bkeys, off_bvals, off_bkeys2, off_bvals2,
        off_hist, total;
};
static int emb_bwd_layout(int64_t B, int f_in, int E, EmbBwdWs* L) {
    const int e4 = E / 4;
    L->n = B * f_in;
    const int64_t per_block = (int64_t)(SR_THREADS / e4) * SR_PW;
    L->n_blk = (L->n + per_block - 1) / per_block;
    L->n_blk2 = (2 * L->n_blk + per_block - 1) / per_block;         // second level (only when the final kernel cannot take level 1's list)
    L->rs_nb = (L->n + RS_PER_BLOCK - 1) / RS_PER_BLOCK;
    int64_t o = 0;
    L->off_keys_a = o; o = align256(o + L->n * 4);
    L->off_keys_b = o; o = align256(o + L->n * 4);
    L->off_vals_a = o; o = align256(o + L->n * 4);
    L->off_vals_b = o; o = align256(o + L->n * 4);
    L->off_bkeys = o;  o = align256(o + L->n_blk * 2 * 4);
    L->off_bvals = o;  o = align256(o + L->n_blk * 2 * (int64_t)E * 4);
    L->off_bkeys2 = o; o = align256(o + L->n_blk2 * 2 * 4);
    L->off_bvals2 = o; o = align256(o + L->n_blk2 * 2 * (int64_t)E * 4);
    L->off_hist = o;   o = align256(o + L->rs_nb * RS_BUCKETS * 4);
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
    AR_CHECK_ARG(n_table_rows > 0 && n_table_rows < (1ll << 31) - 1, "aread_embed_bwd: bad table size");
    AR_CHECK_ARG(((uintptr_t)ws & 255) == 0, "aread_embed_bwd: workspace alignment");
    AR_CHECK_ARG(B * f_in < (1ll << 31), "aread_embed_bwd: too many lookups");
    hipStream_t st = (hipStream_t)stream;
    EmbBwdWs L;
    AR_CHECK_ARG(emb_bwd_layout(B, f_in, E, &L) == 0, "aread_embed_bwd: workspace layout failed");
    char* base = (char*)ws;
    uint32_t* ka = (uint32_t*)(base + L.off_keys_a);
    uint32_t* kb = (uint32_t*)(base + L.off_keys_b);
    uint32_t* va = (uint32_t*)(base + L.off_vals_a);
    uint32_t* vb = (uint32_t*)(base + L.off_vals_b);
    int32_t* hist = (int32_t*)(base + L.off_hist);
    const int f_out = n_onehot + n_mh_fields;
    hipLaunchKernelGGL(k_embed_bwd_keys, dim3(cdiv(L.n, 256)), dim3(256), 0, st, x, offsets, sample_row, ka, va,
                       L.n, f_in, f_out, n_onehot, seq_len, pool);
    AR_LAUNCH_CHECK();
    const int passes = rs_passes(n_table_rows), nb = (int)L.rs_nb;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * RS_BITS < 32 ? p * RS_BITS : 31;       // passes beyond the key width see digit 0: identity
        const bool pad_pass = p * RS_BITS >= 32;
        hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(RS_THREADS), 0, st, ka, L.n, pad_pass ? 31 : shift, nb, hist);
        AR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, st, hist, (int64_t)nb * RS_BUCKETS);
        AR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(RS_THREADS), 0, st, ka, va, L.n, pad_pass ? 31 : shift, nb, hist, kb, vb);
        AR_LAUNCH_CHECK();
        uint32_t* t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    return AREAD_OK;
}

// phase 2: segmented reduction of the sorted pairs into table_grad
extern "C" int aread_embed_bwd_reduce(int64_t B, int f_in, int E, int seq_len, const float* dout, float* table_grad, void* ws,
                                      void* stream) {
    return aread_embed_bwd_reduce2(B, f_in, E, seq_len, dout, nullptr, table_grad, ws, stream);
}
extern "C" int aread_embed_bwd_reduce2(int64_t B, int f_in, int E, int seq_len, const float* dout, const float* dout2, float* table_grad,
                                       void* ws, void* stream) {
    AR_CHECK_ARG(dout && table_grad && ws, "aread_embed_bwd_reduce: null pointer");
    AR_CHECK_ARG(((uintptr_t)dout2 & 15) == 0, "aread_embed_bwd_reduce: alignment");
    AR_CHECK_ARG(E > 0 && E % 4 == 0 && E <= 256 && seq_len >= 1, "aread_embed_bwd_reduce: bad E/seq_len");
    AR_CHECK_ARG(((uintptr_t)dout & 15) == 0 && ((uintptr_t)table_grad & 15) == 0 && ((uintptr_t)ws & 255) == 0,
                 "aread_embed_bwd_reduce: alignment");
    hipStream_t st = (hipStream_t)stream;
    EmbBwdWs L;
    AR_CHECK_ARG(emb_bwd_layout(B, f_in, E, &L) == 0, "aread_embed_bwd: workspace layout failed");
    char* base = (char*)ws;
    const int32_t* keys_b = (const int32_t*)(base + L.off_keys_b);
    const uint32_t* vals_b = (const uint32_t*)(base + L.off_vals_b);
    int32_t* bkeys = (int32_t*)(base + L.off_bkeys);
    v4f* bvals = (v4f*)(base + L.off_bvals);
    int32_t* bkeys2 = (int32_t*)(base + L.off_bkeys2);
    v4f* bvals2 = (v4f*)(base + L.off_bvals2);
    const int e4 = E / 4;
    const int nw1 = SR_THREADS / e4, nwf = SRF_THREADS / e4;
    const size_t lds1 = (size_t)2 * nw1 * e4 * 16 + (size_t)2 * nw1 * 4;
    const size_t ldsf = (size_t)2 * nwf * e4 * 16 + (size_t)2 * nwf * 4;
    const int64_t cap_final = (int64_t)nwf * SR_PW;                  // entries the one-workgroup final level can take
    const float inv = 1.0f / (float)seq_len;
    if (L.n <= cap_final) {
        hipLaunchKernelGGL((k_segreduce<true, true, SRF_THREADS>), dim3(1), dim3(SRF_THREADS), ldsf, st, keys_b, vals_b,
                           (const v4f*)dout, (const v4f*)dout2, L.n, e4, inv, (v4f*)table_grad, (int32_t*)nullptr, (v4f*)nullptr);
        AR_LAUNCH_CHECK();
        return AREAD_OK;
    }
    hipLaunchKernelGGL((k_segreduce<true, false, SR_THREADS>), dim3((unsigned)L.n_blk), dim3(SR_THREADS), lds1, st, keys_b,
                       vals_b, (const v4f*)dout, (const v4f*)dout2, L.n, e4, inv, (v4f*)table_grad, bkeys, bvals);
    AR_LAUNCH_CHECK();
    int64_t n_cur = 2 * L.n_blk;
    const int32_t* ck = bkeys; const v4f* cv = bvals;
    int32_t* nk = bkeys2; v4f* nv = bvals2;
    while (n_cur > cap_final) {                                      // (B > ~30 k lookups-per-final-capacity: one more level)
        const int64_t per_block = (int64_t)nw1 * SR_PW, blocks = (n_cur + per_block - 1) / per_block;
        hipLaunchKernelGGL((k_segreduce<false, false, SR_THREADS>), dim3((unsigned)blocks), dim3(SR_THREADS), lds1, st, ck,
                           (const uint32_t*)nullptr, cv, (const v4f*)nullptr, n_cur, e4, 1.f, (v4f*)table_grad, nk, nv);
        AR_LAUNCH_CHECK();
        n_cur = 2 * blocks;
        const int32_t* tk = ck; const v4f* tv = cv;
        ck = nk; cv = nv; nk = (int32_t*)tk; nv = (v4f*)tv;
    }
    hipLaunchKernelGGL((k_segreduce<false, true, SRF_THREADS>), dim3(1), dim3(SRF_THREADS), ldsf, st, ck, (const uint32_t*)nullptr,
                       cv, (const v4f*)nullptr, n_cur, e4, 1.f, (v4f*)table_grad, (int32_t*)nullptr, (v4f*)nullptr);
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
