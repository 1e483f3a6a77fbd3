// dense_bwd_kernels.h -- backward kernels of the dense path (non-GEMM part) and the reductions that turn
// per-tile partial sums into parameter gradients in a fixed order (bitwise reproducible, no float atomics).
#pragma once
#include "dense_kernels.h"
#define MAX_BN_LAYERS_DECL (AREAD_MAX_LAYER * (AREAD_MAX_LEVEL + 1))

// ------------------------------------------------------------------------------------------------
// generic reductions
// ------------------------------------------------------------------------------------------------
// out[(c / gw) * gs + c % gw] (+)= sum over live partial slots of part[slot*ld + c]
// A slot covers 64/sub rows (sub partial slots per 64-row tile).  Block = 8 slot-groups x 32 columns; the
// 8 partial sums of a column are combined through LDS in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void k_reduce_tiles(const float* part, int64_t ld, int ncols, float* out, int gw,
                                                      int64_t gs, int accumulate, int sub, RowsP r) {
    __shared__ float s_acc[8][33];
    const int cl = threadIdx.x & 31, tg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < ncols) {
        const int n_slots = r.hdr[PLAN_NTILES] * sub;          // live tiles are contiguous: no per-slot test
#pragma unroll 8
        for (int t = tg; t < n_slots; t += 8) s += part[(int64_t)t * ld + c];
    }
    s_acc[tg][cl] = s;
    __syncthreads();
    if (tg == 0 && c < ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += s_acc[k][cl];
        float* o = out + (int64_t)(c / gw) * gs + (c % gw);
        *o = accumulate ? *o + tot : tot;
    }
}

// The same reduction for up to 4 column ranges of one partial buffer with separate destinations, in one launch
// (the row-wise backward's cn_w / cn_b / lin_w / lin_b partials): column c of range j goes to out[j][c - c0[j]].
struct ReduceMultiP {
    const float* part; int64_t ld; int n, sub;
    int c0[4], c1[4]; float* out[4];
    RowsP r;
};
__global__ __launch_bounds__(256) void k_reduce_tiles_multi(const ReduceMultiP p) {
    __shared__ float s_acc[8][33];
    const int cl = threadIdx.x & 31, tg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int ncols = p.c1[p.n - 1];
    float s = 0.f;
    if (c < ncols) {
        const int n_slots = p.r.hdr[PLAN_NTILES] * p.sub;
#pragma unroll 8
        for (int t = tg; t < n_slots; t += 8) s += p.part[(int64_t)t * p.ld + c];
    }
    s_acc[tg][cl] = s;
    __syncthreads();
    if (tg == 0 && c < ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += s_acc[k][cl];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < p.n && c >= p.c0[j] && c < p.c1[j]) p.out[j][c - p.c0[j]] = tot;
    }
}

// out[seg][c] = sum over the tiles of segment seg of part[tile*ld + c]   (grid: (n_seg, ceil(ncols/16)))
__global__ __launch_bounds__(256) void k_seg_reduce(const float* part, int64_t ld, int ncols, float* out, int sub, RowsP r) {
    __shared__ float s_acc[16][17];
    const int seg = blockIdx.x;
    const int cl = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int c = blockIdx.y * 16 + cl;
    const int cnt = r.seg_count[seg];
    const int t0 = (r.seg_start[seg] / TILE_M) * sub, nt = ((cnt + TILE_M - 1) / TILE_M) * sub;
    float s = 0.f;
    if (c < ncols)
        for (int t = tg; t < nt; t += 16) s += part[(int64_t)(t0 + t) * ld + c];
    s_acc[tg][cl] = s;
    __syncthreads();
    if (tg == 0 && c < ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_acc[k][cl];
        out[(int64_t)seg * ncols + c] = tot;
    }
}

// part[tile][c] = sum over the valid rows of the tile of X[row][c]     (16 row groups x 16 columns per pass)
__global__ __launch_bounds__(256) void k_colsum(const float* X, int64_t ldx, int ncols, float* part, int64_t ldp, RowsP r) {
    __shared__ float s_acc[16][17];
    const int tile = blockIdx.x;
    if (r.tile_seg[tile] < 0) return;
    const int nvalid = r.tile_valid[tile];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    for (int cb = 0; cb < ncols; cb += 16) {
        const int c = cb + cl;
        float s = 0.f;
        if (c < ncols)
            for (int rr = rg; rr < nvalid; rr += 16) s += X[((int64_t)tile * TILE_M + rr) * ldx + c];
        s_acc[rg][cl] = s;
        __syncthreads();
        if (rg == 0 && c < ncols) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += s_acc[k][cl];
            part[(int64_t)tile * ldp + c] = tot;
        }
        __syncthreads();
    }
}

// k_colsum for up to 4 inputs in one launch (grid (n_tiles, n)): the bias gradients of the gate layers
struct ColsumOne { const float* X; int64_t ldx; int ncols; float* part; int64_t ldp; };
struct ColsumAllP { int n; ColsumOne d[4]; RowsP r; };
__global__ __launch_bounds__(256) void k_colsum_all(const ColsumAllP a) {
    __shared__ float s_acc[16][17];
    const ColsumOne& p = a.d[blockIdx.y];
    const int tile = blockIdx.x;
    if (a.r.tile_seg[tile] < 0) return;
    const int nvalid = a.r.tile_valid[tile];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    for (int cb = 0; cb < p.ncols; cb += 16) {
        const int c = cb + cl;
        float s = 0.f;
        if (c < p.ncols)
            for (int rr = rg; rr < nvalid; rr += 16) s += p.X[((int64_t)tile * TILE_M + rr) * p.ldx + c];
        s_acc[rg][cl] = s;
        __syncthreads();
        if (rg == 0 && c < p.ncols) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tot += s_acc[k][cl];
            p.part[(int64_t)tile * p.ldp + c] = tot;
        }
        __syncthreads();
    }
}

// k_reduce_tiles for up to 4 partial buffers in one launch (grid (max column blocks, n)); the sums are ADDED to out
struct ReduceOne { const float* part; int64_t ld; int ncols; float* out; int gw; int64_t gs; int sub; };
struct ReduceAllP { int n; ReduceOne d[4]; RowsP r; };
__global__ __launch_bounds__(256) void k_reduce_tiles_all(const ReduceAllP a) {
    __shared__ float s_acc[8][33];
    const ReduceOne& p = a.d[blockIdx.y];
    if ((int)blockIdx.x * 32 >= p.ncols) return;
    const int cl = threadIdx.x & 31, tg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < p.ncols) {
        const int n_slots = a.r.hdr[PLAN_NTILES] * p.sub;
#pragma unroll 8
        for (int t = tg; t < n_slots; t += 8) s += p.part[(int64_t)t * p.ld + c];
    }
    s_acc[tg][cl] = s;
    __syncthreads();
    if (tg == 0 && c < p.ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += s_acc[k][cl];
        float* o = p.out + (int64_t)(c / p.gw) * p.gs + (c % p.gw);
        *o += tot;
    }
}

// batched weight transposes: WT[g][i][o] = W[g][o][i]   (grid.y = layer; weights are small)
struct TransOne { const float* W; float* WT; int G, out, in; };
struct TransAllP { int n; TransOne d[MAX_BN_LAYERS_DECL]; };
__global__ __launch_bounds__(256) void k_transpose_weights(const TransAllP a) {
    const TransOne& p = a.d[blockIdx.y];
    const int64_t n = (int64_t)p.G * p.out * p.in;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int o = (int)(idx % p.out);
        const int64_t r = idx / p.out;
        const int i = (int)(r % p.in), g = (int)(r / p.in);
        p.WT[idx] = p.W[((int64_t)g * p.out + o) * p.in + i];
    }
}

// batched split-K reduction: for every wgrad d, out[g][m][n] = sum_ks slab[ks][g][m][n]  (grid.y = d)
struct SplitKOne { const float* slab; float* out; int k_split, G, M, N; int64_t ldo, o_gs; int transposed; };
#define MAX_WGRADS 24
struct SplitKAllP { int n; SplitKOne d[MAX_WGRADS]; };
__device__ __forceinline__ void splitk_reduce_body(const SplitKAllP& a, int job) {
    // block = 32 groups of 4 consecutive elements x 8 interleaved slab groups: a thread sums the slabs k = kg, kg+8, ... (four
    // independent 16-byte loads in flight), the 8 partial sums are combined through LDS in a fixed order.  Small weight
    // gradients are cut into many slabs (up to rows/64): the slab loop must not be one serial chain.
    __shared__ float4 s_acc[8][32];
    const SplitKOne& p = a.d[job];
    const int64_t per = (int64_t)p.G * p.M * p.N;
    const int el = threadIdx.x & 31, kg = threadIdx.x >> 5;
    const bool vec = (per & 3) == 0 && (p.N & 3) == 0;
    const int64_t n_items = vec ? (per >> 2) : per;            // items of 4 elements (vector path) or single elements
    for (int64_t base = (int64_t)blockIdx.x * 32; base < n_items; base += (int64_t)gridDim.x * 32) {
        const int64_t it = base + el;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < n_items) {
            if (vec) {
                const float4* src = (const float4*)p.slab + it;
                const int64_t per4 = per >> 2;
                int k = kg;
                for (; k + 24 < p.k_split; k += 32) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(k + 8 * u) * per4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                for (; k < p.k_split; k += 8) { const float4 v = src[(int64_t)k * per4]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            } else {
                for (int k = kg; k < p.k_split; k += 8) acc.x += p.slab[(int64_t)k * per + it];
            }
        }
        s_acc[kg][el] = acc;
        __syncthreads();
        if (kg == 0 && it < n_items) {
            float4 t = s_acc[0][el];
#pragma unroll
            for (int q = 1; q < 8; ++q) { const float4 b = s_acc[q][el]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
            const int64_t idx = vec ? (it << 2) : it;
            const int g = (int)(idx / ((int64_t)p.M * p.N));
            const int rem = (int)(idx - (int64_t)g * p.M * p.N);
            const int m = rem / p.N, n = rem - m * p.N;
            // the gradient buffer holds its initial value (zero, or the dense L2 term 2*coef*w): the sums are ADDED to it
            if (!vec) {
                if (p.transposed) p.out[(int64_t)g * p.o_gs + (int64_t)n * p.ldo + m] += t.x;
                else p.out[(int64_t)g * p.o_gs + (int64_t)m * p.ldo + n] += t.x;
            } else if (p.transposed) {                         // the slab holds the transposed product
                float* o = p.out + (int64_t)g * p.o_gs + (int64_t)n * p.ldo + m;
                o[0] += t.x; o[p.ldo] += t.y; o[2 * p.ldo] += t.z; o[3 * p.ldo] += t.w;
            } else {
                float* o = p.out + (int64_t)g * p.o_gs + (int64_t)m * p.ldo + n;
                if ((p.ldo & 3) == 0 && (p.o_gs & 3) == 0 && (((uintptr_t)p.out) & 15) == 0) {
                    const float4 b = *(const float4*)o;
                    *(float4*)o = make_float4(b.x + t.x, b.y + t.y, b.z + t.z, b.w + t.w);
                } else { o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_splitk_reduce_all(const SplitKAllP a) { splitk_reduce_body(a, blockIdx.y); }

// batched per-layer column reductions at the end of the backward (grid.y = layer):
//   db[c]     = sum over live tiles of cpart[tile][c]
//   dbeta[c]  = sum over tiles of BatchNorm-applied, active segments of bpart[tile][c][0]
//   dgamma[c] = ... of bpart[tile][c][1]
struct BiasOne { const float* cpart; const float* bpart; float* db; float* dgamma; float* dbeta; int ncols, h, level; };
struct BiasAllP { int n; BiasOne d[MAX_BN_LAYERS_DECL]; RowsP r; ModeP mp; };
__device__ __forceinline__ void bias_reduce_body(const BiasAllP& a, int job) {
    __shared__ float s_acc[3][16][17];
    __shared__ int s_seg[1024];                  // tile -> segment (or -1), with bit 30 set when the segment's BatchNorm applies
    const BiasOne& p = a.d[job];
    const int cl = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    if (blockIdx.x * 16 >= p.ncols) return;
    const int n_tiles = a.r.n_tiles < 1024 ? a.r.n_tiles : 1024;
    for (int t = threadIdx.x; t < n_tiles; t += 256) {          // tile metadata once per block: the sums below then issue
        const int sg = a.r.tile_seg[t];                         // nothing but independent loads
        s_seg[t] = sg < 0 ? -1 : (sg | (a.r.seg_count[sg] > 1 ? (1 << 30) : 0));
    }
    __syncthreads();
    float sb = 0.f, s1 = 0.f, s2 = 0.f;
    if (c < p.ncols) {
        const uint8_t* act = p.level >= 0 ? active_level(a.mp, p.level) : nullptr;
        const int grp = c / p.h;
#pragma unroll 4
        for (int t = tg; t < a.r.n_tiles; t += 16) {
            const int v = t < 1024 ? s_seg[t] : (a.r.tile_seg[t] < 0 ? -1 : (a.r.tile_seg[t] | (a.r.seg_count[a.r.tile_seg[t]] > 1 ? (1 << 30) : 0)));
            if (v < 0) continue;
            sb += p.cpart[(int64_t)t * p.ncols + c];
            if (!(v >> 30)) continue;
            if (act && !act[(v & 0xFFFF) * MAX_TOWER + grp]) continue;
            const float2 bp = *(const float2*)(p.bpart + ((int64_t)t * p.ncols + c) * 2);
            s1 += bp.x; s2 += bp.y;
        }
    }
    s_acc[0][tg][cl] = sb; s_acc[1][tg][cl] = s1; s_acc[2][tg][cl] = s2;
    __syncthreads();
    if (tg < 3 && c < p.ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_acc[tg][k][cl];
        float* o = tg == 0 ? p.db : (tg == 1 ? p.dbeta : p.dgamma);
        o[c] += tot;
    }
}
__global__ __launch_bounds__(256) void k_bias_reduce_all(const BiasAllP a) { bias_reduce_body(a, blockIdx.y); }
// both batches in ONE launch (grid.y = split-K jobs, then the column-reduction jobs): they add into disjoint parts of the gradient
// buffer and sit at the tail of the side stream's chain, where a launch boundary is ~5 us of the step
__global__ __launch_bounds__(256) void k_reduce_tail_all(const SplitKAllP a, const BiasAllP b) {
    if ((int)blockIdx.y < a.n) splitk_reduce_body(a, blockIdx.y);
    else bias_reduce_body(b, blockIdx.y - a.n);
}

// ------------------------------------------------------------------------------------------------
// heads backward: dAct_last = dz * v_tail, dlin = sum_i dz, per-tile partials of dv_tail
// ------------------------------------------------------------------------------------------------
struct HeadsBwdP {
    const float* dz; const float* act; const float* head_w; int head_ld, D, h, n_heads, ld_h;
    float* dact; float* dlin; float* part; int64_t ldp;
    RowsP r;
};

__global__ __launch_bounds__(256) void k_heads_bwd(const HeadsBwdP p) {
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    if (p.r.tile_seg[tile] < 0) return;
    const int ncols = p.n_heads * p.h;
    for (int it = threadIdx.x; it < SUB_ROWS * ncols; it += 256) {
        const int rr = r_lo + it / ncols, col = it % ncols;
        const int i = col / p.h, c = col - i * p.h;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        p.dact[row * ncols + col] = p.dz[row * p.ld_h + i] * p.head_w[(int64_t)i * p.head_ld + p.D + c];
    }
    for (int rl = threadIdx.x; rl < SUB_ROWS; rl += 256) {
        const int64_t row = (int64_t)tile * TILE_M + r_lo + rl;
        float s = 0.f;
        for (int i = 0; i < p.n_heads; ++i) s += p.dz[row * p.ld_h + i];
        p.dlin[row] = s;
    }
    for (int col = threadIdx.x; col < ncols; col += 256) {
        const int i = col / p.h;
        float s = 0.f;
        for (int rl = 0; rl < SUB_ROWS; ++rl) {
            const int64_t row = (int64_t)tile * TILE_M + r_lo + rl;
            s += p.dz[row * p.ld_h + i] * p.act[row * ncols + col];
        }
        p.part[(int64_t)blockIdx.x * p.ldp + col] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// activation backward: dAct -> dyhat (in place) through dropout and ReLU; per-tile sums of dyhat and
// dyhat*xhat for the BatchNorm backward.  Block = (tile, 64-column chunk); thread = (row group, float4).
// ------------------------------------------------------------------------------------------------
struct ActBwdP {
    float* d; const float* H; const float* mean; const float* rstd; const float* gamma; const float* beta;
    float* bpart;                       // [n_tiles][ncols][2]
    int ncols, h, level, stack, layer, train;
    uint32_t seed, thr; float keep_scale; const uint32_t* seed_dev;
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_act_bwd(const ActBwdP p) {
    __shared__ float s1[16][64], s2[16][64];
    const int tile = blockIdx.x, c0 = blockIdx.y * 64;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int c = c0 + cq * 4;
    const bool col_ok = c < p.ncols;
    const int g = col_ok ? c / p.h : 0;
    bool act = col_ok;
    if (act && p.level >= 0) act = active_level(p.mp, p.level)[seg * MAX_TOWER + g] != 0;
    const bool bn = p.r.seg_count[seg] > 1;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4], rs[4], ga[4], be[4];
    if (col_ok) {
        const int64_t so = (int64_t)seg * p.ncols + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) { mu[i] = p.mean[so + i]; rs[i] = p.rstd[so + i]; ga[i] = p.gamma[c + i]; be[i] = p.beta[c + i]; }
    }
    const uint32_t site = (uint32_t)((p.stack * 8 + p.layer) * 64 + g);
    const int cg = c - g * p.h;
    for (int rr = rg; rr < TILE_M; rr += 16) {
        if (!col_ok) break;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float4* dp = (float4*)(p.d + row * p.ncols + c);
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act && rr < nvalid) {
            const float4 dv = *dp;
            const float4 hv = *(const float4*)(p.H + row * p.ncols + c);
            float d[4] = {dv.x, dv.y, dv.z, dv.w}, hh[4] = {hv.x, hv.y, hv.z, hv.w};
            const uint32_t key = (p.train && p.thr) ? drop_row_key(drop_seed_of(p.seed, p.seed_dev), (uint32_t)p.r.row_sample[row]) : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (hh[i] - mu[i]) * rs[i];
                const float y = bn ? xh * ga[i] + be[i] : hh[i];
                float dy = d[i];
                if (p.train && p.thr) dy = drop_keep(key, site, (uint32_t)(cg + i), p.thr) ? dy * p.keep_scale : 0.f;
                dy = y > 0.f ? dy : 0.f;
                d[i] = dy;
                a1[i] += dy;
                a2[i] += dy * xh;
            }
            out = make_float4(d[0], d[1], d[2], d[3]);
        }
        *dp = out;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[rg][cq * 4 + i] = a1[i]; s2[rg][cq * 4 + i] = a2[i]; }
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < p.ncols) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < 16; ++k) { t1 += s1[k][threadIdx.x]; t2 += s2[k][threadIdx.x]; }
        float* o = p.bpart + ((int64_t)tile * p.ncols + c0 + threadIdx.x) * 2;
        o[0] = t1; o[1] = t2;
    }
}

// dyhat -> dH (in place): dH = gamma*rstd*(dyhat - s1/n - xhat*s2/n); per-tile column sums of dH (bias grads)
struct BnBwdApplyP {
    float* d; const float* H; const float* mean; const float* rstd; const float* gamma; const float* bpart;
    float* cpart;                       // [n_tiles][ncols]
    int ncols, h, level, train;         // train == 0: statistics are constants (running stats): dH = gamma*rstd*dyhat
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const BnBwdApplyP p) {
    __shared__ float s1[16][64];
    __shared__ float s_half[2][128];
    __shared__ float s_sum[128];                 // [col][2]: (sum dyhat, sum dyhat*xhat) of this block's segment
    const int tile = blockIdx.x, c0 = blockIdx.y * 64;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int c = c0 + cq * 4;
    const bool col_ok = c < p.ncols;
    const int ncol_here = (p.ncols - c0 < 64 ? p.ncols - c0 : 64) * 2;
    const int cnt = p.r.seg_count[seg];
    {   // per-segment sums for this block's 64 columns, fixed order: two interleaved halves, then combined
        const int t0 = p.r.seg_start[seg] / TILE_M, nt = (cnt + TILE_M - 1) / TILE_M;
        const int v = threadIdx.x & 127, half = threadIdx.x >> 7;
        float acc = 0.f;
        if (v < ncol_here)
            for (int t = half; t < nt; t += 2) acc += p.bpart[((int64_t)(t0 + t) * p.ncols + c0) * 2 + v];
        s_half[half][v] = acc;
        __syncthreads();
        if (threadIdx.x < 128) s_sum[threadIdx.x] = s_half[0][threadIdx.x] + s_half[1][threadIdx.x];
        __syncthreads();
    }
    bool act = col_ok;
    if (act && p.level >= 0) act = active_level(p.mp, p.level)[seg * MAX_TOWER + c / p.h] != 0;
    const bool bn = cnt > 1 && act;     // inactive towers: d is already zero and H was never written
    const float inv_n = 1.0f / (float)cnt;
    float a1[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4], rs[4], ga[4], m1[4], m2[4];
    if (col_ok) {
        const int64_t so = (int64_t)seg * p.ncols + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mu[i] = p.mean[so + i]; rs[i] = p.rstd[so + i]; ga[i] = p.gamma[c + i];
            m1[i] = p.train ? s_sum[(cq * 4 + i) * 2] * inv_n : 0.f;
            m2[i] = p.train ? s_sum[(cq * 4 + i) * 2 + 1] * inv_n : 0.f;
        }
    }
    for (int rr = rg; rr < nvalid; rr += 16) {
        if (!col_ok) break;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float4* dp = (float4*)(p.d + row * p.ncols + c);
        const float4 dv = *dp;
        float d[4] = {dv.x, dv.y, dv.z, dv.w};
        if (bn) {
            const float4 hv = *(const float4*)(p.H + row * p.ncols + c);
            const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (hh[i] - mu[i]) * rs[i];
                d[i] = ga[i] * rs[i] * (d[i] - m1[i] - xh * m2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] += d[i];
        *dp = make_float4(d[0], d[1], d[2], d[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[rg][cq * 4 + i] = a1[i];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < p.ncols) {
        float t1 = 0.f;
        for (int k = 0; k < 16; ++k) t1 += s1[k][threadIdx.x];
        p.cpart[(int64_t)tile * p.ncols + c0 + threadIdx.x] = t1;
    }
}

// ------------------------------------------------------------------------------------------------
// k_act_bwd + k_bn_bwd_apply in ONE launch for the wide (expert) layers: the dropout / ReLU backward keeps dyhat and xhat of
// its 64x64 block in registers, hands the block's column sums to the other tiles of the segment inside the kernel (the
// data-tagged granules of tower_fused.h: one 8-byte {tag, value} store per sum, swept by the merging threads until every tag
// matches), merges the segment's partials in the order of k_bn_bwd_apply and applies the BatchNorm backward from the
// registers: d and H are read once instead of twice and one launch boundary disappears (step -12 us; the first version with
// drain + counter + poll hand-offs was +40 us).
// Grid (tile, 64-column chunk): the workgroups that wait for each other differ only in the tile index and are dispatched
// next to each other, so they are co-resident whenever the device holds 2 x n_tiles workgroups (checked by the launcher).
// ------------------------------------------------------------------------------------------------
struct ActBnBwdP {
    ActBwdP a;                          // d, H, mean, rstd, gamma, beta, bpart, dropout site ...
    float* cpart;                       // [n_tiles][ncols]
    tf_u64* tags;                       // [n_tiles][ncols][2] data-tagged (sum dyhat, sum dyhat*xhat) granules, zeroed per step
    unsigned* err;
    tf_u64* fin;                        // [MAX_SEG][ncols][2] the two sums of a long segment, published by the granule's owner tile
    int two_hop_nt;                     // segments of more tiles take the two-hop merge
};

__global__ __launch_bounds__(256) void k_act_bn_bwd(const ActBnBwdP q) {
    const ActBwdP& p = q.a;
    __shared__ float s1[16][64], s2[16][64];
    __shared__ float s_half[2][128];
    __shared__ float s_sum[128];
    const int tile = blockIdx.x, c0 = blockIdx.y * 64;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int c = c0 + cq * 4;
    const bool col_ok = c < p.ncols;
    const int g = col_ok ? c / p.h : 0;
    bool act = col_ok;
    if (act && p.level >= 0) act = active_level(p.mp, p.level)[seg * MAX_TOWER + g] != 0;
    const int cnt = p.r.seg_count[seg];
    const bool bn = cnt > 1;
    const bool sync_stats = p.train && bn;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (col_ok) {
        const int64_t so = (int64_t)seg * p.ncols + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) { mu[i] = p.mean[so + i]; rs[i] = p.rstd[so + i]; ga[i] = p.gamma[c + i]; be[i] = p.beta[c + i]; }
    }
    const uint32_t site = (uint32_t)((p.stack * 8 + p.layer) * 64 + g);
    const int cg = c - g * p.h;
    // ---- 1. dropout / ReLU backward: dyhat, xhat of this thread's four rows stay in registers --------------------------
    float dy[4][4], xh[4][4];
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 dv[4], hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rr = rg + 16 * k;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        dv[k] = make_float4(0.f, 0.f, 0.f, 0.f); hv[k] = dv[k];
        if (act && rr < nvalid) { dv[k] = *(const float4*)(p.d + row * p.ncols + c); hv[k] = *(const float4*)(p.H + row * p.ncols + c); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rr = rg + 16 * k;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        const float d[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w}, hh[4] = {hv[k].x, hv[k].y, hv[k].z, hv[k].w};
        const bool on = act && rr < nvalid;
        const uint32_t key = (on && p.train && p.thr) ? drop_row_key(drop_seed_of(p.seed, p.seed_dev), (uint32_t)p.r.row_sample[row]) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = 0.f, v = 0.f;
            if (on) {
                x = (hh[i] - mu[i]) * rs[i];
                const float y = bn ? x * ga[i] + be[i] : hh[i];
                v = d[i];
                if (p.train && p.thr) v = drop_keep(key, site, (uint32_t)(cg + i), p.thr) ? v * p.keep_scale : 0.f;
                v = y > 0.f ? v : 0.f;
                a1[i] += v;
                a2[i] += v * x;
            }
            dy[k][i] = v; xh[k][i] = x;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[rg][cq * 4 + i] = a1[i]; s2[rg][cq * 4 + i] = a2[i]; }
    __syncthreads();
    const int cnt_ = cnt;
    const bool sync_now = sync_stats;
    if (threadIdx.x < 64 && c0 + threadIdx.x < p.ncols) {
        float t1 = 0.f, t2 = 0.f;
        for (int k = 0; k < 16; ++k) { t1 += s1[k][threadIdx.x]; t2 += s2[k][threadIdx.x]; }
        float* o = p.bpart + ((int64_t)tile * p.ncols + c0 + threadIdx.x) * 2;     // (read later by the bias / gamma / beta reduction)
        o[0] = t1; o[1] = t2;
        if (sync_now) tf_put_tagged(q.tags + ((int64_t)tile * p.ncols + c0 + threadIdx.x) * 2, t1, t2);
    }
    // ---- 2. hand-off: sweep the data-tagged granules of the segment's tiles for this chunk's columns ------------------------
    const int ncol_here = (p.ncols - c0 < 64 ? p.ncols - c0 : 64) * 2;
    const int nt_seg = (cnt_ + TILE_M - 1) / TILE_M;
    if (sync_now && nt_seg > q.two_hop_nt) {
        // long segment (a one-domain batch: 128 tiles): two hops instead of every tile summing every tile's partials (tower_fused.h).
        // Tile i of the segment owns the granules v = i (mod nt) of this column chunk; one wave per owned granule adds the nt
        // partials (lane = tile, then a butterfly, lower lane first) and publishes the sum; then 128 threads read the 128 sums.
        const int t0 = p.r.seg_start[seg] / TILE_M, nt = nt_seg, ti = tile - t0;
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        tf_u64* fin = q.fin + ((int64_t)seg * p.ncols + c0) * 2;
        for (int v = ti + nt * wv; v < ncol_here; v += nt * 4) {                   // (wave-uniform)
            const tf_u64* src = q.tags + ((int64_t)t0 * p.ncols + c0) * 2 + v;
            float acc = 0.f;
            for (int t = ln; t < nt; t += 64) {
                tf_u64 x;
                unsigned spins = 0;
                do { x = __hip_atomic_load(src + (int64_t)t * p.ncols * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                while ((unsigned)(x >> 32) != TF_TAG && ++spins < (1u << 22));
                if ((unsigned)(x >> 32) != TF_TAG) __hip_atomic_store(q.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __uint_as_float((unsigned)x);
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float u = __shfl_xor(acc, o);
                acc = (ln & o) ? u + acc : acc + u;
            }
            if (ln == 0) __hip_atomic_store(fin + v, ((tf_u64)TF_TAG << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x < 128) {
            float sum = 0.f;
            if ((int)threadIdx.x < ncol_here) {
                tf_u64 x;
                unsigned spins = 0;
                do { x = __hip_atomic_load(fin + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                while ((unsigned)(x >> 32) != TF_TAG && ++spins < (1u << 22));
                if ((unsigned)(x >> 32) != TF_TAG) __hip_atomic_store(q.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sum = __uint_as_float((unsigned)x);
            }
            s_sum[threadIdx.x] = sum;
        }
        __syncthreads();
    } else if (sync_now) {
        const int t0 = p.r.seg_start[seg] / TILE_M, nt = (cnt_ + TILE_M - 1) / TILE_M;
        // per-segment sums of this block's 64 columns, the order of k_bn_bwd_apply: two interleaved halves, then combined
        const int v = threadIdx.x & 127, half = threadIdx.x >> 7;
        float acc = 0.f;
        if (v < ncol_here) {
            const tf_u64* src = q.tags + ((int64_t)t0 * p.ncols + c0) * 2 + v;       // granule (column v/2, component v&1) of tile t0
            // 12 partials in flight per round (a one-domain batch of 128 tiles takes six rounds), summed in tile order
            for (int kb = 0; half + 2 * kb < nt; kb += 12) {
                float b[12];
                bool have[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) { b[k] = 0.f; have[k] = half + 2 * (kb + k) >= nt; }
                for (unsigned spins = 0;;) {
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        if (!have[k]) {
                            const tf_u64 x = __hip_atomic_load(src + (int64_t)(half + 2 * (kb + k)) * p.ncols * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            have[k] = (unsigned)(x >> 32) == TF_TAG;
                            b[k] = __uint_as_float((unsigned)x);
                        }
                        all = all && have[k];
                    }
                    if (all) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { __hip_atomic_store(q.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
#pragma unroll
                for (int k = 0; k < 12; ++k) acc += b[k];
            }
        }
        s_half[half][v] = acc;
        __syncthreads();
        if (threadIdx.x < 128) s_sum[threadIdx.x] = s_half[0][threadIdx.x] + s_half[1][threadIdx.x];
        __syncthreads();
    }
    // ---- 3. BatchNorm backward from the registers: dH = gamma*rstd*(dyhat - s1/n - xhat*s2/n); column sums -> cpart ----
    const bool bna = bn && act;
    const float inv_n = 1.0f / (float)cnt;
    float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
    if (col_ok && sync_stats) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { m1[i] = s_sum[(cq * 4 + i) * 2] * inv_n; m2[i] = s_sum[(cq * 4 + i) * 2 + 1] * inv_n; }
    }
    float a3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rr = rg + 16 * k;
        if (!col_ok) break;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = dy[k][i];
            if (bna && rr < nvalid) v = ga[i] * rs[i] * (v - m1[i] - xh[k][i] * m2[i]);
            o[i] = v;
            a3[i] += rr < nvalid ? v : 0.f;
        }
        *(float4*)(p.d + row * p.ncols + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) s1[rg][cq * 4 + i] = a3[i];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < p.ncols) {
        float t1 = 0.f;
        for (int k = 0; k < 16; ++k) t1 += s1[k][threadIdx.x];
        q.cpart[(int64_t)tile * p.ncols + c0 + threadIdx.x] = t1;
    }
}

// ------------------------------------------------------------------------------------------------
// gate-mix backward, level l >= 1: dIn[row][t][:] -> dprev[row][s][:] and dglogT
// ------------------------------------------------------------------------------------------------
struct MixLBwdP {
    const float* glog; float* dglog; int ld_g, goff;
    const float* prev; const float* dIn; float* dprev;
    int n_src, n_t, w, level, mask_off;
    RowsP r; ModeP mp;
};

// Block = SUB_ROWS rows of one tile.  dIn and prev rows are staged in LDS with 16-byte coalesced loads, every
// later access is LDS; phase A (thread per (row, tower)) produces the gate-logit gradients and keeps the
// renormalised gates in LDS, phase B (thread per (row, source, float4)) produces dprev.
#define MIX_LDS_FLOATS 4096      // per staged operand: SUB_ROWS * (towers * width) <= 4096 floats
__global__ __launch_bounds__(256) void k_mixl_bwd(const MixLBwdP p) {
    __shared__ __attribute__((aligned(16))) float s_din[MIX_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_prev[MIX_LDS_FLOATS];
    __shared__ float s_ah[SUB_ROWS][MAX_TOWER * MAX_TOWER / 2 + 1];
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const uint8_t* act = active_level(p.mp, p.level) + seg * MAX_TOWER;
    const uint8_t* mk = p.mp.masks ? p.mp.masks + (size_t)p.mp.seg_dom[seg] * p.mp.edge_count + p.mask_off : nullptr;
    const int wt = p.n_t * p.w, ws_ = p.n_src * p.w;          // floats per row of dIn / prev
    const int64_t row0 = (int64_t)tile * TILE_M + r_lo;
    for (int i = threadIdx.x; i < SUB_ROWS * wt / 4; i += 256) {
        const int rl = i / (wt / 4), c4 = i - rl * (wt / 4);
        *(float4*)(s_din + rl * wt + c4 * 4) = *(const float4*)(p.dIn + (row0 + rl) * wt + c4 * 4);
    }
    for (int i = threadIdx.x; i < SUB_ROWS * ws_ / 4; i += 256) {
        const int rl = i / (ws_ / 4), c4 = i - rl * (ws_ / 4);
        *(float4*)(s_prev + rl * ws_ + c4 * 4) = *(const float4*)(p.prev + (row0 + rl) * ws_ + c4 * 4);
    }
    __syncthreads();
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_t; it += 256) {
        const int rl = it / p.n_t, t = it - rl * p.n_t, rr = r_lo + rl;
        const int64_t row = row0 + rl;
        const bool on = rr < nvalid && act[t];
        float a[MAX_TOWER], am[MAX_TOWER], ah[MAX_TOWER], dah[MAX_TOWER], S = 1.f;
        float* dgl = p.dglog + row * p.ld_g + p.goff + t * p.n_src;
        if (!on) {
            for (int s = 0; s < p.n_src; ++s) { dgl[s] = 0.f; s_ah[rl][t * p.n_src + s] = 0.f; }
            continue;
        }
        gate_weights(p.glog + row * p.ld_g + p.goff + t * p.n_src, p.n_src, mk, p.n_t, t, p.mp.mode, a, am, ah, &S);
        const float* din = s_din + rl * wt + t * p.w;
        float dot_ah = 0.f;
        for (int s = 0; s < p.n_src; ++s) {
            const float* src = s_prev + rl * ws_ + s * p.w;
            float acc = 0.f;
            for (int c = 0; c < p.w; c += 4) {
                const float4 d4 = *(const float4*)(din + c), x4 = *(const float4*)(src + c);
                acc += d4.x * x4.x + d4.y * x4.y + d4.z * x4.z + d4.w * x4.w;
            }
            dah[s] = acc;
            dot_ah += acc * ah[s];
            s_ah[rl][t * p.n_src + s] = ah[s];
        }
        float da[MAX_TOWER], dot_a = 0.f;
        for (int s = 0; s < p.n_src; ++s) {
            if (p.mp.mode == 1) da[s] = dah[s];
            else da[s] = mk[s * p.n_t + t] ? (dah[s] - dot_ah) / S : 0.f;
            dot_a += da[s] * a[s];
        }
        for (int s = 0; s < p.n_src; ++s) dgl[s] = a[s] * (da[s] - dot_a);
    }
    __syncthreads();
    const int w4 = p.w / 4;
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_src * w4; it += 256) {
        const int rl = it / (p.n_src * w4), rem = it - rl * (p.n_src * w4);
        const int s = rem / w4, c = (rem - s * w4) * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r_lo + rl < nvalid)
            for (int t = 0; t < p.n_t; ++t) {
                const float wgt = s_ah[rl][t * p.n_src + s];
                if (wgt != 0.f) {
                    const float4 x = *(const float4*)(s_din + rl * wt + t * p.w + c);
                    o.x += wgt * x.x; o.y += wgt * x.y; o.z += wgt * x.z; o.w += wgt * x.w;
                }
            }
        *(float4*)(p.dprev + (row0 + rl) * ws_ + s * p.w + c) = o;
    }
}

// MMoE mix backward: dU[row][t][:] -> dX[row][k][:] and dglogE
struct Mix0BwdP {
    const float* glog; float* dglog; int ld_g; const float* X; const float* dU; float* dX;
    int n_t, n_exp, h;
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_mix0_bwd(const Mix0BwdP p) {
    __shared__ __attribute__((aligned(16))) float s_du[MIX_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_x[2 * MIX_LDS_FLOATS];
    __shared__ float s_pi[SUB_ROWS][MAX_TOWER * 8 + 1];
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const uint8_t* act = active_level(p.mp, 0) + seg * MAX_TOWER;
    const int wu = p.n_t * p.h, wx = p.n_exp * p.h;
    const int64_t row0 = (int64_t)tile * TILE_M + r_lo;
    for (int i = threadIdx.x; i < SUB_ROWS * wu / 4; i += 256) {
        const int rl = i / (wu / 4), c4 = i - rl * (wu / 4);
        *(float4*)(s_du + rl * wu + c4 * 4) = *(const float4*)(p.dU + (row0 + rl) * wu + c4 * 4);
    }
    for (int i = threadIdx.x; i < SUB_ROWS * wx / 4; i += 256) {
        const int rl = i / (wx / 4), c4 = i - rl * (wx / 4);
        *(float4*)(s_x + rl * wx + c4 * 4) = *(const float4*)(p.X + (row0 + rl) * wx + c4 * 4);
    }
    __syncthreads();
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_t; it += 256) {
        const int rl = it / p.n_t, t = it - rl * p.n_t, rr = r_lo + rl;
        const int64_t row = row0 + rl;
        float* dgl = p.dglog + row * p.ld_g + t * p.n_exp;
        if (!(rr < nvalid && act[t])) {
            for (int k = 0; k < p.n_exp; ++k) { dgl[k] = 0.f; s_pi[rl][t * p.n_exp + k] = 0.f; }
            continue;
        }
        const float* gl = p.glog + row * p.ld_g + t * p.n_exp;
        float pi[8], dpi[8];
        float mx = gl[0];
        for (int k = 1; k < p.n_exp; ++k) mx = fmaxf(mx, gl[k]);
        float den = 0.f;
        for (int k = 0; k < p.n_exp; ++k) { pi[k] = __expf(gl[k] - mx); den += pi[k]; }
        const float* du = s_du + rl * wu + t * p.h;
        float dot = 0.f;
        for (int k = 0; k < p.n_exp; ++k) {
            pi[k] /= den;
            const float* xk = s_x + rl * wx + k * p.h;
            float acc = 0.f;
            for (int c = 0; c < p.h; c += 4) {
                const float4 d4 = *(const float4*)(du + c), x4 = *(const float4*)(xk + c);
                acc += d4.x * x4.x + d4.y * x4.y + d4.z * x4.z + d4.w * x4.w;
            }
            dpi[k] = acc;
            dot += acc * pi[k];
            s_pi[rl][t * p.n_exp + k] = pi[k];
        }
        for (int k = 0; k < p.n_exp; ++k) dgl[k] = pi[k] * (dpi[k] - dot);
    }
    __syncthreads();
    const int h4 = p.h >> 2;
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_exp * h4; it += 256) {
        const int rl = it / (p.n_exp * h4), rem = it - rl * (p.n_exp * h4);
        const int k = rem / h4, c = (rem - k * h4) * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r_lo + rl < nvalid)
            for (int t = 0; t < p.n_t; ++t) {
                const float wgt = s_pi[rl][t * p.n_exp + k];
                if (wgt != 0.f) {
                    const float4 x = *(const float4*)(s_du + rl * wu + t * p.h + c);
                    o.x += wgt * x.x; o.y += wgt * x.y; o.z += wgt * x.z; o.w += wgt * x.w;
                }
            }
        *(float4*)(p.dX + (row0 + rl) * wx + k * p.h + c) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// row-wise trunk backward: cross network (layer.py:517-537), linear term, gate input.
//
// With c_i = e (1 + X_i) + B_i  (X_i = sum_{j<i} x_j.w_j saved by the forward, B_i = sum_{j<i} b_j) the whole backward of
// a row collapses onto a handful of per-row SCALARS:
//     ds_i = dcn.e + sum_{j>i} ds_j (w_j.e)          (three dot products per row, one interleaved reduction)
//     u_i  = ds_i (1 + X_i),   alpha = 1 + sum_i xw_i
//     de  += alpha dcn + sum_i u_i w_i + dl w_lin + deg (+ dq on the domain field's columns)
// and the parameter gradients are column sums over the rows:
//     dw_i = sum_r u_i e + B_i sum_r ds_i,   db_i = sum_r dcn + sum_{j>i} w_j sum_r ds_j,   dw_lin = sum_r dl e,  db_lin = sum_r dl
// so a workgroup only accumulates  EU_i = sum u_i e (NC vectors), EL = sum dl e, DC = sum dcn  and NC+1 scalars; the
// finishing kernel (k_rowwise_finish) applies the B_i / w_j terms once.  No recomputation of c_i, a third of the
// accumulator registers of a literal transcription, every load of a row independent of every other.
//
// Half a wave (32 lanes) per row, lane hl owns float4 chunks hl, hl+32, ...; a workgroup covers RWB_ROWS rows.
// part layout per workgroup: [NC][D] EU | [D] EL | [D] DC | NC x sum ds_i | sum dl
// ------------------------------------------------------------------------------------------------
#define RWB_ROWS 16                    // 2 rows per half-wave: more workgroups in flight beat fewer partial slots (measured)
#define RWB_SUB (TILE_M / RWB_ROWS)
struct RowwiseBwdP {
    const float* e; const float* xw; const float* dcn; const float* dlin; const float* dq; const float* deg;
    const float* lin_w; const float* cn_w; const float* cn_b;
    float* de; float* part; int64_t part_ld; float* dgrp_part;
    int D, E, n_cross, dom_field; int64_t rows;
    int de_init;                        // 1: de is written (every row of a live tile, zeros on its padding rows), not added to
    RowsP r;
};

template <int NC, int NV>
__global__ __launch_bounds__(256) void k_rowwise_bwd(const RowwiseBwdP p) {
    __shared__ float4 s_red[4][NV][32];
    __shared__ float s_sc[8][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, hw = wave * 2 + (lane >> 5);
    const int tile = blockIdx.x / RWB_SUB, r_lo = (blockIdx.x % RWB_SUB) * RWB_ROWS;
    if (p.r.tile_seg[tile] < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int r_hi = (r_lo + RWB_ROWS < nvalid) ? r_lo + RWB_ROWS : nvalid;
    const int d4 = p.D >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NCA = NC > 0 ? NC : 1;
    float4 eu[NCA][NV], el[NV], dcs[NV];
    float sds[NCA], sdl = 0.f;
#pragma unroll
    for (int i = 0; i < NCA; ++i) sds[i] = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        el[v] = zero; dcs[v] = zero;
#pragma unroll
        for (int i = 0; i < NCA; ++i) eu[i][v] = zero;
    }
    const int dom_c0 = p.dom_field * p.E, dom_c1 = dom_c0 + p.E;
    for (int rr = r_lo + hw; rr < r_lo + RWB_ROWS; rr += 8) {   // (RWB_ROWS / 8 rows per half-wave)
        const bool live = rr < r_hi;                   // (the whole half-wave agrees; dead rows only keep the shuffles uniform)
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float4 e[NV], dc[NV], dg[NV], o[NV];
        const float4* e4 = (const float4*)(p.e + row * p.D);
        const float4* g4 = (const float4*)(p.dcn + row * p.D);
        const float4* q4 = (const float4*)(p.deg + row * p.D);
        float4* o4 = (float4*)(p.de + row * p.D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int ch = hl + 32 * v;
            const bool on = live && ch < d4;
            e[v] = on ? e4[ch] : zero;
            dc[v] = on ? g4[ch] : zero;
            dg[v] = on ? q4[ch] : zero;
            o[v] = (on && !p.de_init) ? o4[ch] : zero;
        }
        float xw[NCA];
#pragma unroll
        for (int i = 0; i < NC; ++i) xw[i] = live ? p.xw[(int64_t)i * p.rows + row] : 0.f;
        const float dl = live ? p.dlin[row] : 0.f;
        float dots[NCA];                               // [0] = dcn.e, [j] = w_j.e (j >= 1)
#pragma unroll
        for (int j = 0; j < NCA; ++j) dots[j] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int ch = hl + 32 * v;
            dots[0] += dot4(dc[v], e[v]);
#pragma unroll
            for (int j = 1; j < NC; ++j)
                if (ch < d4) dots[j] += dot4(((const float4*)(p.cn_w + (int64_t)j * p.D))[ch], e[v]);
        }
#pragma unroll
        for (int ofs = 16; ofs > 0; ofs >>= 1) {
#pragma unroll
            for (int j = 0; j < NCA; ++j) dots[j] += __shfl_xor(dots[j], ofs);
        }
        float ds[NCA], u[NCA], alpha = 1.f, X = 0.f;
#pragma unroll
        for (int i = NC - 1; i >= 0; --i) {
            float t = dots[0];
#pragma unroll
            for (int j = NC - 1; j > i; --j) t += ds[j] * dots[j];
            ds[i] = t;
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            u[i] = ds[i] * (1.f + X);
            X += xw[i];
            alpha += xw[i];
            sds[i] += ds[i];
        }
        sdl += dl;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int ch = hl + 32 * v;
            if (ch < d4) {
                const float4 wl = ((const float4*)p.lin_w)[ch];
                float4 t = o[v];
                t.x += dg[v].x + alpha * dc[v].x + dl * wl.x; t.y += dg[v].y + alpha * dc[v].y + dl * wl.y;
                t.z += dg[v].z + alpha * dc[v].z + dl * wl.z; t.w += dg[v].w + alpha * dc[v].w + dl * wl.w;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const float4 w = ((const float4*)(p.cn_w + (int64_t)i * p.D))[ch];
                    t.x += u[i] * w.x; t.y += u[i] * w.y; t.z += u[i] * w.z; t.w += u[i] * w.w;
                    eu[i][v].x += u[i] * e[v].x; eu[i][v].y += u[i] * e[v].y; eu[i][v].z += u[i] * e[v].z; eu[i][v].w += u[i] * e[v].w;
                }
                const int col = ch * 4;
                if (col >= dom_c0 && col < dom_c1 && live) {       // domain-embedding part of the gate input
                    const float4 dqv = *(const float4*)(p.dq + row * 2 * p.E + (col - dom_c0));
                    t.x += dqv.x; t.y += dqv.y; t.z += dqv.z; t.w += dqv.w;
                }
                if (live) o4[ch] = t;
                else if (p.de_init) o4[ch] = zero;       // padding row of a live tile: the dgrad that accumulates onto de reads it
                el[v].x += dl * e[v].x; el[v].y += dl * e[v].y; el[v].z += dl * e[v].z; el[v].w += dl * e[v].w;
                dcs[v].x += dc[v].x; dcs[v].y += dc[v].y; dcs[v].z += dc[v].z; dcs[v].w += dc[v].w;
            }
        }
    }
    // ---- combine the eight half-waves in a fixed order and write the workgroup's partial -------------------
    float* out = p.part + (int64_t)blockIdx.x * p.part_ld;
    auto flush = [&](float4 (&acc)[NV], int64_t off) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 a = acc[v];
            a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
            if (lane < 32) s_red[wave][v][hl] = a;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NV * 32; i += 256) {
            const int v = i >> 5, l = i & 31, ch = l + 32 * v;
            if (ch < d4) {
                float4 t = s_red[0][v][l];
#pragma unroll
                for (int w = 1; w < 4; ++w) { const float4 b = s_red[w][v][l]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
                *(float4*)(out + off + ch * 4) = t;
            }
        }
        __syncthreads();
    };
#pragma unroll
    for (int i = 0; i < NC; ++i) flush(eu[i], (int64_t)i * p.D);
    flush(el, (int64_t)NC * p.D);
    flush(dcs, (int64_t)(NC + 1) * p.D);
    if (hl == 0) {
#pragma unroll
        for (int i = 0; i < NC; ++i) s_sc[hw][i] = sds[i];
        s_sc[hw][NC] = sdl;
    }
    __syncthreads();
    if (threadIdx.x <= NC) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_sc[k][threadIdx.x];
        out[(int64_t)(NC + 2) * p.D + threadIdx.x] = t;
    }
    // group-embedding gradient partial: sum over valid rows of dq[:, E:2E]
    for (int cidx = threadIdx.x; cidx < p.E; cidx += 256) {
        float s = 0.f;
        for (int rr = r_lo; rr < r_hi; ++rr) s += p.dq[((int64_t)tile * TILE_M + rr) * 2 * p.E + p.E + cidx];
        p.dgrp_part[(int64_t)blockIdx.x * p.E + cidx] = s;
    }
}

// parameter gradients of the row-wise trunk from the workgroup partials (fixed order: 8 interleaved slot groups, then
// combined): dw_i = EU_i + B_i S_i, db_i = DC + sum_{j>i} w_j S_j, dw_lin = EL, db_lin = SL.  Block = 32 columns.
struct RowwiseFinP {
    const float* part; int64_t ld; const float* cn_w; const float* cn_b;
    float* g_cn_w; float* g_cn_b; float* g_lin_w; float* g_lin_b;
    int D, n_cross;
    RowsP r;
};
template <int NC>
__global__ __launch_bounds__(256) void k_rowwise_finish(const RowwiseFinP p) {
    // block = 8 columns x 32 interleaved slot groups (the partial list is short and wide: spread the slots over the
    // lanes so that a thread sees ~10 slots, loaded four at a time)
    constexpr int Q = NC + 2;
    __shared__ float s_acc[32][Q][9];
    __shared__ float s_sc[32][NC + 1];
    const int cl = threadIdx.x & 7, tg = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    const int n_slots = p.r.hdr[PLAN_NTILES] * RWB_SUB;
    const bool col = c < p.D;
    float a[Q], sc[NC + 1];
#pragma unroll
    for (int q = 0; q < Q; ++q) a[q] = 0.f;
#pragma unroll
    for (int q = 0; q <= NC; ++q) sc[q] = 0.f;
#pragma unroll 4
    for (int t = tg; t < n_slots; t += 32) {
        const float* pp = p.part + (int64_t)t * p.ld;
#pragma unroll
        for (int q = 0; q < Q; ++q) a[q] += col ? pp[(int64_t)q * p.D + c] : 0.f;
#pragma unroll
        for (int q = 0; q <= NC; ++q) sc[q] += pp[(int64_t)Q * p.D + q];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) s_acc[tg][q][cl] = a[q];
    if (cl == 0)
#pragma unroll
        for (int q = 0; q <= NC; ++q) s_sc[tg][q] = sc[q];
    __syncthreads();
    if (tg == 0 && col) {
        float tot[Q], S[NC + 1];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += s_acc[k][q][cl];
            tot[q] = t;
        }
#pragma unroll
        for (int q = 0; q <= NC; ++q) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += s_sc[k][q];
            S[q] = t;
        }
        float Bi = 0.f;                                   // B_i = sum_{j<i} b_j
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            p.g_cn_w[(int64_t)i * p.D + c] += tot[i] + Bi * S[i];
            Bi += p.cn_b[(int64_t)i * p.D + c];
            float db = tot[NC + 1];
#pragma unroll
            for (int j = i + 1; j < NC; ++j) db += p.cn_w[(int64_t)j * p.D + c] * S[j];
            p.g_cn_b[(int64_t)i * p.D + c] += db;
        }
        p.g_lin_w[c] += tot[NC];
        if (c == 0) p.g_lin_b[0] += S[NC];
    }
}

// group_embedding gradient (autograd of aread.py:226-229) from the per-segment sums of dq[:, E:2E]
__global__ __launch_bounds__(256) void k_grp_bwd(const float* dgrp_seg, float* dgroup, int n_t0, int E, RowsP r, ModeP mp) {
    for (int i = threadIdx.x; i < n_t0 * E; i += 256) {
        const int t = i / E, c = i - t * E;
        float acc = 0.f;
        if (mp.mode == 0)
            for (int s0 = 0; s0 < r.n_seg; s0 += 8) {              // (loads of eight segments in flight, sums in segment order)
                float sv[8]; int n0[8]; bool on[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int seg = s0 + u;
                    const bool in = seg < r.n_seg;
                    on[u] = in && r.seg_count[seg] != 0 && active_level(mp, 0)[seg * MAX_TOWER + t] != 0;
                    sv[u] = dgrp_seg[(int64_t)(in ? seg : 0) * E + c];
                    n0[u] = mp.n0act[in ? seg : 0];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (on[u]) acc += n0[u] > 1 ? sv[u] / (float)n0[u] : sv[u];
            }
        dgroup[i] += acc;
    }
}

// running statistics (momentum 0.1, unbiased variance), applied segment by segment in domain order,
// exactly like the reference's sequence of per-domain calls.
struct BnRunP {
    const float* mean; const float* var; float* rmean; float* rvar; int64_t* nbt; int ncols, h, level;
};
#define MAX_BN_LAYERS (AREAD_MAX_LAYER * (AREAD_MAX_LEVEL + 1))
struct BnRunAllP {
    int n_layers;
    BnRunP L[MAX_BN_LAYERS];
    RowsP r; ModeP mp;
};
__global__ __launch_bounds__(256) void k_bn_running(const BnRunAllP a) {
    const BnRunP& p = a.L[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.ncols) return;
    float rm = p.rmean[c], rv = p.rvar[c];
    int n_upd = 0;
    // the EMA is a serial recurrence in domain order, its inputs are not: eight segments' loads are issued together (the
    // version with one dependent round trip per segment took 30 us for 25 segments)
    for (int s0 = 0; s0 < a.r.n_seg; s0 += 8) {
        int cnt[8]; float mu[8], va[8]; bool on[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int seg = s0 + u;
            const bool in = seg < a.r.n_seg;
            cnt[u] = in ? a.r.seg_count[seg] : 0;
            on[u] = in && (p.level < 0 || active_level(a.mp, p.level)[seg * MAX_TOWER + c / p.h] != 0);
            const int64_t o = (int64_t)(in ? seg : 0) * p.ncols + c;
            mu[u] = p.mean[o]; va[u] = p.var[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cnt[u] <= 1 || !on[u]) continue;
            rm = (1.0f - BN_MOMENTUM) * rm + BN_MOMENTUM * mu[u];
            rv = (1.0f - BN_MOMENTUM) * rv + BN_MOMENTUM * (va[u] * ((float)cnt[u] / (float)(cnt[u] - 1)));
            ++n_upd;
        }
    }
    p.rmean[c] = rm; p.rvar[c] = rv;
    if (c % p.h == 0) p.nbt[c / p.h] += n_upd;
}

// dense L2: partial sums of coef*w^2 and grads += 2*coef*w
// One launch: block partials of sum coef*w^2 (float4 lanes), grads += 2*coef*w, and the LAST block to finish (self-resetting
// ticket counter) adds the 256 partials in index order into loss_out[0] and, when asked, forms total = loss_in + loss_out[0]
// (the step's final scalar) -- what used to be k_l2_dense + k_l2_finish + a torch.add at the serial tail of the step.
__device__ unsigned g_l2_dense_ticket = 0;
// 2*coef*w as one rounded product: the empty asm keeps hipcc from contracting it into an fma with whatever it is added to
// (HIP's __fmul_rn is a plain multiply and would be contracted)
__device__ __forceinline__ float l2_term(float c, float v) {
    float t = 2.0f * c * v;
    asm volatile("" : "+v"(t));
    return t;
}
// init != 0: grads is WRITTEN (= 2*coef*w, zero where coef is zero) instead of added to -- the step's gradient buffer starts
// from the dense L2 term and every reduction of the backward adds onto it (same sums, commuted: bitwise the same result),
// so the pass runs at the head of the step beside the forward instead of at its serial tail.
__global__ __launch_bounds__(256) void k_l2_dense(const float* w, const float* coef, int64_t n, float* grads, float* partial,
                                                  float* loss_out, int accumulate, const float* loss_in, float* total_out, int init) {
    float acc = 0.f;
    const int64_t n4 = ((((uintptr_t)w | (uintptr_t)coef | (uintptr_t)grads) & 15) == 0) ? n >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        // all three loads in flight together (the gradient load used to wait for the coefficient test: two dependent round trips
        // per iteration on the serial tail of the step)
        const float4 c = ((const float4*)coef)[i], v = ((const float4*)w)[i];
        float4 g = (grads && !init) ? ((const float4*)grads)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        acc += c.x * v.x * v.x; acc += c.y * v.y * v.y; acc += c.z * v.z * v.z; acc += c.w * v.w * v.w;
        if (grads && (init || c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f)) {
            // (l2_term: no contraction into an fma with the running gradient -- the term is the same rounded product whether
            // it initialises the buffer or is added last, so both orders give bitwise the same gradient)
            g.x += l2_term(c.x, v.x); g.y += l2_term(c.y, v.y); g.z += l2_term(c.z, v.z); g.w += l2_term(c.w, v.w);
            ((float4*)grads)[i] = g;
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float c = coef[i], v = w[i];
        acc += c * v * v;
        if (grads && init) grads[i] = l2_term(c, v);
        else if (grads && c != 0.f) grads[i] += l2_term(c, v);
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    __shared__ bool s_last;
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, (s[0] + s[1]) + (s[2] + s[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicInc(&g_l2_dense_ticket, gridDim.x - 1) == gridDim.x - 1;       // wraps to 0: ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    float v = threadIdx.x < gridDim.x ? __hip_atomic_load(partial + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    __shared__ float s_p[256];
    s_p[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < (int)gridDim.x && k < 256; ++k) tot += s_p[k];
        const float r = accumulate ? loss_out[0] + tot : tot;
        loss_out[0] = r;
        if (total_out) total_out[0] = (loss_in ? loss_in[0] : 0.f) + r;
    }
}

// the step's final scalars once every contribution is known: reg[0] = reg_table (already in reg[0]) + reg_dense[0],
// total = loss + reg[0]   (one thread; the sums are formed in the order the tail kernel k_l2_dense used to form them)
__global__ void k_step_total(const float* loss, const float* reg_dense, float* reg, float* total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float r = reg[0] + reg_dense[0];
        reg[0] = r;
        total[0] = loss[0] + r;
    }
}
