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
__global__ __launch_bounds__(256) void k_splitk_reduce_all(const SplitKAllP a) {
    __shared__ float s_acc[4][64];
    const SplitKOne& p = a.d[blockIdx.y];
    const int64_t per = (int64_t)p.G * p.M * p.N;
    const int el = threadIdx.x & 63, kg = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < per; base += (int64_t)gridDim.x * 64) {
        const int64_t idx = base + el;
        float s = 0.f;
        if (idx < per) {
#pragma unroll 4
            for (int k = kg; k < p.k_split; k += 4) s += p.slab[(int64_t)k * per + idx];
        }
        s_acc[kg][el] = s;
        __syncthreads();
        if (kg == 0 && idx < per) {
            const float tot = (s_acc[0][el] + s_acc[1][el]) + (s_acc[2][el] + s_acc[3][el]);
            const int g = (int)(idx / ((int64_t)p.M * p.N));
            const int rem = (int)(idx - (int64_t)g * p.M * p.N);
            const int m = rem / p.N, n = rem - m * p.N;
            if (p.transposed) p.out[(int64_t)g * p.o_gs + (int64_t)n * p.ldo + m] = tot;   // slab holds the transposed product
            else p.out[(int64_t)g * p.o_gs + (int64_t)m * p.ldo + n] = tot;
        }
        __syncthreads();
    }
}

// batched per-layer column reductions at the end of the backward (grid.y = layer):
//   db[c]     = sum over live tiles of cpart[tile][c]
//   dbeta[c]  = sum over tiles of BatchNorm-applied, active segments of bpart[tile][c][0]
//   dgamma[c] = ... of bpart[tile][c][1]
struct BiasOne { const float* cpart; const float* bpart; float* db; float* dgamma; float* dbeta; int ncols, h, level; };
struct BiasAllP { int n; BiasOne d[MAX_BN_LAYERS_DECL]; RowsP r; ModeP mp; };
__global__ __launch_bounds__(256) void k_bias_reduce_all(const BiasAllP a) {
    __shared__ float s_acc[3][16][17];
    const BiasOne& p = a.d[blockIdx.y];
    const int cl = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    if (blockIdx.x * 16 >= p.ncols) return;
    float sb = 0.f, s1 = 0.f, s2 = 0.f;
    if (c < p.ncols)
        for (int t = tg; t < a.r.n_tiles; t += 16) {
            const int sg = a.r.tile_seg[t];
            if (sg < 0) continue;
            sb += p.cpart[(int64_t)t * p.ncols + c];
            if (a.r.seg_count[sg] <= 1) continue;
            if (p.level >= 0 && !active_level(a.mp, p.level)[sg * MAX_TOWER + c / p.h]) continue;
            const float* bp = p.bpart + ((int64_t)t * p.ncols + c) * 2;
            s1 += bp[0]; s2 += bp[1];
        }
    s_acc[0][tg][cl] = sb; s_acc[1][tg][cl] = s1; s_acc[2][tg][cl] = s2;
    __syncthreads();
    if (tg < 3 && c < p.ncols) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += s_acc[tg][k][cl];
        float* o = tg == 0 ? p.db : (tg == 1 ? p.dbeta : p.dgamma);
        o[c] = tot;
    }
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
    uint32_t seed, thr; float keep_scale;
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
            const uint32_t key = (p.train && p.thr) ? drop_row_key(p.seed, (uint32_t)p.r.row_sample[row]) : 0u;
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
// row-wise trunk backward: cross network, linear term, gate input.  de_out += ...; per-tile partials
// of the parameter gradients (dw_i, db_i, dw_lin, db_lin) and of the group-embedding gradient.
// One workgroup per tile; wave w walks rows w, w+4, ...
// part layout per tile: [n_cross][D] dw | [n_cross][D] db | [D] dw_lin | 1 db_lin
// ------------------------------------------------------------------------------------------------
struct RowwiseBwdP {
    const float* e; const float* xw; const float* dcn; const float* dlin; const float* dq; const float* deg;
    const float* lin_w; const float* cn_w; const float* cn_b;
    float* de; float* part; int64_t part_ld; float* dgrp_part;
    int D, E, n_cross, dom_field; int64_t rows;
    RowsP r;
};

template <int NC, int RW_MAXV>
__global__ __launch_bounds__(256) void k_rowwise_bwd(const RowwiseBwdP p) {
    __shared__ float4 s_red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    if (p.r.tile_seg[tile] < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int r_hi = (r_lo + SUB_ROWS < nvalid) ? r_lo + SUB_ROWS : nvalid;
    const int d4 = p.D >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NCA = NC > 0 ? NC : 1;
    float4 adw[NCA][RW_MAXV], adb[NCA][RW_MAXV], adl[RW_MAXV];
    float adbl = 0.f;
#pragma unroll
    for (int v = 0; v < RW_MAXV; ++v) {
        adl[v] = zero;
#pragma unroll
        for (int i = 0; i < NCA; ++i) { adw[i][v] = zero; adb[i][v] = zero; }
    }
    for (int rr = r_lo + wave; rr < r_hi; rr += 4) {
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float4 e[RW_MAXV], c[NCA][RW_MAXV], dc[RW_MAXV], de[RW_MAXV];
        const float4* e4 = (const float4*)(p.e + row * p.D);
        const float4* g4 = (const float4*)(p.dcn + row * p.D);
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            e[v] = ch < d4 ? e4[ch] : zero;
            dc[v] = ch < d4 ? g4[ch] : zero;
            de[v] = zero;
            c[0][v] = e[v];
        }
        // recompute c_1 .. c_{n-1} from the saved x.w scalars
#pragma unroll
        for (int i = 0; i + 1 < NC; ++i) {
            {
                const float xw = p.xw[(int64_t)i * p.rows + row];
                const float4* b4 = (const float4*)(p.cn_b + (int64_t)i * p.D);
#pragma unroll
                for (int v = 0; v < RW_MAXV; ++v) {
                    const int ch = lane + 64 * v;
                    if (ch < d4) {
                        const float4 b = b4[ch];
                        c[i + 1][v].x = e[v].x * xw + b.x + c[i][v].x;
                        c[i + 1][v].y = e[v].y * xw + b.y + c[i][v].y;
                        c[i + 1][v].z = e[v].z * xw + b.z + c[i][v].z;
                        c[i + 1][v].w = e[v].w * xw + b.w + c[i][v].w;
                    } else c[i + 1][v] = zero;
                }
            }
        }
        // ds_i = dc_i . e with dc_i = dcn + sum_{j>i} w_j ds_j  =>  ds_i = (dcn . e) + sum_{j>i} ds_j (w_j . e): the row's
        // dot products are independent of the recursion, so one interleaved wave reduction replaces NC chained ones
        float dots[NCA];                       // [0] = dcn . e, [j] = w_j . e (j >= 1)
#pragma unroll
        for (int j = 0; j < NCA; ++j) dots[j] = 0.f;
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            dots[0] += dot4(dc[v], e[v]);
#pragma unroll
            for (int j = 1; j < NC; ++j)
                if (ch < d4) dots[j] += dot4(((const float4*)(p.cn_w + (int64_t)j * p.D))[ch], e[v]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int j = 0; j < NCA; ++j) dots[j] += __shfl_xor(dots[j], o);
        }
        float dsv[NCA];
#pragma unroll
        for (int i = NC - 1; i >= 0; --i) {
            float t = dots[0];
#pragma unroll
            for (int j = NC - 1; j > i; --j) t += dsv[j] * dots[j];
            dsv[i] = t;
        }
#pragma unroll
        for (int i = NC - 1; i >= 0; --i) {
            {
                const float xw = p.xw[(int64_t)i * p.rows + row];
                const float4* w4 = (const float4*)(p.cn_w + (int64_t)i * p.D);
                const float ds = dsv[i];
#pragma unroll
                for (int v = 0; v < RW_MAXV; ++v) {
                    const int ch = lane + 64 * v;
                    if (ch < d4) {
                        const float4 w = w4[ch];
                        de[v].x += dc[v].x * xw; de[v].y += dc[v].y * xw; de[v].z += dc[v].z * xw; de[v].w += dc[v].w * xw;
                        adb[i][v].x += dc[v].x; adb[i][v].y += dc[v].y; adb[i][v].z += dc[v].z; adb[i][v].w += dc[v].w;
                        adw[i][v].x += ds * c[i][v].x; adw[i][v].y += ds * c[i][v].y;
                        adw[i][v].z += ds * c[i][v].z; adw[i][v].w += ds * c[i][v].w;
                        dc[v].x += w.x * ds; dc[v].y += w.y * ds; dc[v].z += w.z * ds; dc[v].w += w.w * ds;
                    }
                }
            }
        }
        const float dl = p.dlin[row];
        adbl += dl;
        float4* o4 = (float4*)(p.de + row * p.D);
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            if (ch < d4) {
                const float4 wl = ((const float4*)p.lin_w)[ch];
                float4 t = o4[ch];
                const float4 gq = ((const float4*)(p.deg + row * p.D))[ch];       // MMoE-gate part of dE (side stream)
                t.x += gq.x; t.y += gq.y; t.z += gq.z; t.w += gq.w;
                t.x += de[v].x + dc[v].x + dl * wl.x; t.y += de[v].y + dc[v].y + dl * wl.y;
                t.z += de[v].z + dc[v].z + dl * wl.z; t.w += de[v].w + dc[v].w + dl * wl.w;
                // domain-embedding part of the gate input
                const int col = ch * 4;
                if (col >= p.dom_field * p.E && col < (p.dom_field + 1) * p.E) {
                    const float4 dqv = *(const float4*)(p.dq + row * 2 * p.E + (col - p.dom_field * p.E));
                    t.x += dqv.x; t.y += dqv.y; t.z += dqv.z; t.w += dqv.w;
                }
                o4[ch] = t;
                adl[v].x += dl * e[v].x; adl[v].y += dl * e[v].y; adl[v].z += dl * e[v].z; adl[v].w += dl * e[v].w;
            }
        }
    }
    // ---- combine the four waves through LDS, vector by vector, and write the tile partial ----------
    float* out = p.part + (int64_t)blockIdx.x * p.part_ld;
    auto flush = [&](float4 (&acc)[RW_MAXV], int64_t off) {
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            __syncthreads();
            if (ch < d4) s_red[wave][lane] = acc[v];
            __syncthreads();
            if (wave == 0 && ch < d4) {
                float4 t = s_red[0][lane];
                for (int w = 1; w < 4; ++w) { t.x += s_red[w][lane].x; t.y += s_red[w][lane].y; t.z += s_red[w][lane].z; t.w += s_red[w][lane].w; }
                *(float4*)(out + off + ch * 4) = t;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        flush(adw[i], (int64_t)i * p.D);
        flush(adb[i], (int64_t)(NC + i) * p.D);
    }
    flush(adl, (int64_t)2 * p.n_cross * p.D);
    __syncthreads();
    if (lane == 0) s_red[wave][0].x = adbl;
    __syncthreads();
    if (threadIdx.x == 0) out[(int64_t)(2 * p.n_cross + 1) * p.D] = (s_red[0][0].x + s_red[1][0].x) + (s_red[2][0].x + s_red[3][0].x);
    // group-embedding gradient partial: sum over valid rows of dq[:, E:2E]
    for (int cidx = threadIdx.x; cidx < p.E; cidx += 256) {
        float s = 0.f;
        for (int rr = r_lo; rr < r_hi; ++rr) s += p.dq[((int64_t)tile * TILE_M + rr) * 2 * p.E + p.E + cidx];
        p.dgrp_part[(int64_t)blockIdx.x * p.E + cidx] = s;
    }
}

// group_embedding gradient (autograd of aread.py:226-229) from the per-segment sums of dq[:, E:2E]
__global__ __launch_bounds__(256) void k_grp_bwd(const float* dgrp_seg, float* dgroup, int n_t0, int E, RowsP r, ModeP mp) {
    for (int i = threadIdx.x; i < n_t0 * E; i += 256) {
        const int t = i / E, c = i - t * E;
        float acc = 0.f;
        if (mp.mode == 0)
            for (int seg = 0; seg < r.n_seg; ++seg) {
                if (r.seg_count[seg] == 0 || !active_level(mp, 0)[seg * MAX_TOWER + t]) continue;
                const float s = dgrp_seg[(int64_t)seg * E + c];
                const int n0 = mp.n0act[seg];
                acc += n0 > 1 ? s / (float)n0 : s;
            }
        dgroup[i] = acc;
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
    for (int seg = 0; seg < a.r.n_seg; ++seg) {
        const int cnt = a.r.seg_count[seg];
        if (cnt <= 1) continue;
        if (p.level >= 0 && !active_level(a.mp, p.level)[seg * MAX_TOWER + c / p.h]) continue;
        const int64_t o = (int64_t)seg * p.ncols + c;
        rm = (1.0f - BN_MOMENTUM) * rm + BN_MOMENTUM * p.mean[o];
        rv = (1.0f - BN_MOMENTUM) * rv + BN_MOMENTUM * (p.var[o] * ((float)cnt / (float)(cnt - 1)));
        ++n_upd;
    }
    p.rmean[c] = rm; p.rvar[c] = rv;
    if (c % p.h == 0) p.nbt[c / p.h] += n_upd;
}

// dense L2: partial sums of coef*w^2 and grads += 2*coef*w
__global__ __launch_bounds__(256) void k_l2_dense(const float* w, const float* coef, int64_t n, float* grads, float* partial) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float c = coef[i], v = w[i];
        acc += c * v * v;
        if (grads && c != 0.f) grads[i] += 2.0f * c * v;
    }
    acc = wave_sum(acc);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
