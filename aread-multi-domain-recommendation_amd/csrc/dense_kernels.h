// dense_kernels.h -- the non-GEMM kernels of the dense path.  Everything here is row-local or a
// per-(segment, column) reduction: HBM/L2-bandwidth or latency bound, fp32 throughout.
//
// Row layout: rows follow the row plan (segments = domains, padded to 64-row tiles).  A tile is
// entirely inside one segment, so the domain mask, the BatchNorm statistics and the "is this tower
// active for this domain" predicate are wave-uniform: there is no host-side branching anywhere.
#pragma once
#include "common.h"
#include "model.h"

#define BN_EPS 1e-5f
#define BN_MOMENTUM 0.1f
#define GATE_EPS 1e-8f
#define SUB 4                        // row-local kernels split a 64-row tile into SUB workgroups ...
#define SUB_ROWS (TILE_M / SUB)      // ... of SUB_ROWS rows; per-workgroup partial sums are indexed by "slot"

struct RowsP {                       // the parts of the row plan a kernel needs
    const int32_t* tile_seg;
    const int32_t* tile_valid;
    const int32_t* row_sample;
    const int32_t* seg_count;
    const int32_t* seg_start;
    const int32_t* hdr;              // plan header: hdr[PLAN_NTILES] = number of live tiles (they are contiguous from 0)
    int n_tiles, n_seg;
};

struct ModeP {                       // per-call mask tables (built by k_mask_prep)
    const uint8_t* active;           // [n_level][MAX_SEG][MAX_TOWER]
    const int32_t* kact;             // [MAX_SEG] active heads
    const int32_t* n0act;            // [MAX_SEG] active level-0 towers
    const int32_t* seg_dom;          // [MAX_SEG] domain of a segment
    const uint8_t* masks;            // [n_domain][edge_count] (nullable in wo_mask)
    int edge_count, mode;
};

__device__ __forceinline__ const uint8_t* active_level(const ModeP& mp, int l) {
    return mp.active + (size_t)l * MAX_SEG * MAX_TOWER;
}

// ------------------------------------------------------------------------------------------------
// mask tables + group embedding  (aread.py:225-230,266-268)
// ------------------------------------------------------------------------------------------------
struct MaskPrepP {
    const uint8_t* masks; int n_seg, domain, mode, n_level, n_domain, edge_count, E;
    int n_tower[AREAD_MAX_LEVEL]; int mask_off[AREAD_MAX_LEVEL + 1];
    const float* group_emb;
    uint8_t* active; int32_t* kact; int32_t* n0act; int32_t* seg_dom; float* grp;
};

__global__ __launch_bounds__(256) void k_mask_prep(const MaskPrepP p) {
    // the tables are built in LDS and written out once: every phase used to re-read the previous phase's bytes from global
    // memory (this single-workgroup kernel heads the side chain the tower kernel waits for)
    __shared__ uint8_t s_active[AREAD_MAX_LEVEL * MAX_SEG * MAX_TOWER];
    __shared__ int s_dom[MAX_SEG], s_n0[MAX_SEG];
    const int tid = threadIdx.x;
    int n_tot = 0;
    for (int l = 0; l < p.n_level; ++l) n_tot += p.n_tower[l];
    for (int i = tid; i < AREAD_MAX_LEVEL * MAX_SEG * MAX_TOWER; i += 256) s_active[i] = 0;
    for (int s = tid; s < MAX_SEG; s += 256) {
        int dom = p.n_seg == 1 ? p.domain : s;
        dom = dom < 0 ? 0 : (dom >= p.n_domain ? p.n_domain - 1 : dom);
        s_dom[s] = dom;
        p.seg_dom[s] = dom;
    }
    __syncthreads();
    // one thread per (segment, tower): is the tower fed by any edge of the segment's domain mask?
    for (int i = tid; i < p.n_seg * n_tot; i += 256) {
        const int s = i / n_tot;
        int t = i - s * n_tot, l = 0;
        while (t >= p.n_tower[l]) { t -= p.n_tower[l]; ++l; }
        int a = 1;
        if (p.mode == 0) {
            const uint8_t* mk = p.masks + (size_t)s_dom[s] * p.edge_count + p.mask_off[l];
            const int n_src = l == 0 ? 1 : p.n_tower[l - 1], nt = p.n_tower[l];
            a = 0;
#pragma unroll 4
            for (int src = 0; src < n_src; ++src) a |= mk[src * nt + t] ? 1 : 0;
        }
        s_active[((size_t)l * MAX_SEG + s) * MAX_TOWER + t] = (uint8_t)a;
    }
    __syncthreads();
    for (int i = tid; i < AREAD_MAX_LEVEL * MAX_SEG * MAX_TOWER / 4; i += 256) ((uint32_t*)p.active)[i] = ((const uint32_t*)s_active)[i];
    for (int s = tid; s < MAX_SEG; s += 256) {
        int k = 0, n0 = 0;
        if (s < p.n_seg) {
            for (int t = 0; t < p.n_tower[0]; ++t) n0 += s_active[(size_t)s * MAX_TOWER + t];
            for (int t = 0; t < p.n_tower[p.n_level - 1]; ++t) k += s_active[((size_t)(p.n_level - 1) * MAX_SEG + s) * MAX_TOWER + t];
        }
        p.kact[s] = k;
        p.n0act[s] = n0;
        s_n0[s] = n0;
    }
    __syncthreads();
    // group embedding of each segment: mean of the rows of the active level-0 towers (0 in wo_mask)
    for (int i = tid; i < p.n_seg * p.E; i += 256) {
        const int s = i / p.E, c = i - s * p.E;
        float acc = 0.f;
        if (p.mode == 0) {
            for (int t = 0; t < p.n_tower[0]; ++t)
                if (s_active[(size_t)s * MAX_TOWER + t]) acc += p.group_emb[t * p.E + c];
            const int n0 = s_n0[s];
            if (n0 > 1) acc = acc / (float)n0;
        }
        p.grp[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// row-wise trunk: linear term, cross network, gate input  (layer.py:115-126, 529-537; aread.py:132,230)
// one wave per row; lane i owns float4 chunks i, i+64, ...
// ------------------------------------------------------------------------------------------------
struct RowwiseP {
    const float* e; float* cn; float* lin; float* xw; float* q; const float* grp;
    const float* lin_w; const float* lin_b; const float* cn_w; const float* cn_b;
    int D, E, n_cross, dom_field; int64_t rows;
    RowsP r;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

template <int RW_MAXV>
__global__ __launch_bounds__(256) void k_rowwise_fwd(const RowwiseP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int d4 = p.D >> 2;
    for (int rr = r_lo + wave; rr < r_lo + SUB_ROWS; rr += 4) {
        const int64_t row = (int64_t)tile * TILE_M + rr;
        const bool valid = rr < nvalid;
        float4 e[RW_MAXV], c[RW_MAXV];
        const float4* e4 = (const float4*)(p.e + row * p.D);
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            e[v] = (ch < d4 && valid) ? e4[ch] : make_float4(0.f, 0.f, 0.f, 0.f);
            c[v] = e[v];
        }
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            if (ch < d4) s += dot4(e[v], ((const float4*)p.lin_w)[ch]);
        }
        s = wave_sum(s) + p.lin_b[0];
        if (lane == 0) p.lin[row] = valid ? s : 0.f;
        for (int i = 0; i < p.n_cross; ++i) {
            const float4* w4 = (const float4*)(p.cn_w + (int64_t)i * p.D);
            const float4* b4 = (const float4*)(p.cn_b + (int64_t)i * p.D);
            float xw = 0.f;
#pragma unroll
            for (int v = 0; v < RW_MAXV; ++v) {
                const int ch = lane + 64 * v;
                if (ch < d4) xw += dot4(c[v], w4[ch]);
            }
            xw = wave_sum(xw);
            if (lane == 0) p.xw[(int64_t)i * p.rows + row] = xw;
#pragma unroll
            for (int v = 0; v < RW_MAXV; ++v) {
                const int ch = lane + 64 * v;
                if (ch < d4) {
                    const float4 b = b4[ch];
                    c[v].x = e[v].x * xw + b.x + c[v].x;
                    c[v].y = e[v].y * xw + b.y + c[v].y;
                    c[v].z = e[v].z * xw + b.z + c[v].z;
                    c[v].w = e[v].w * xw + b.w + c[v].w;
                }
            }
        }
        float4* cn4 = (float4*)(p.cn + row * p.D);
#pragma unroll
        for (int v = 0; v < RW_MAXV; ++v) {
            const int ch = lane + 64 * v;
            if (ch < d4) cn4[ch] = valid ? c[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // gate input q = [domain embedding, group embedding]
        const int E = p.E;
        for (int cidx = lane; cidx < 2 * E; cidx += 64) {
            float v = 0.f;
            if (valid) v = cidx < E ? p.e[row * p.D + p.dom_field * E + cidx] : p.grp[seg * E + (cidx - E)];
            p.q[row * 2 * E + cidx] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics + apply + ReLU + dropout: H -> Act   (layer.py:209-229), one launch.
// Block = (64-row tile, 64-column chunk).  Every block first merges the per-tile (mean, M2) partials of its
// segment for its 64 columns (Chan, fixed order: 4 interleaved tile groups, then combined), the block of the
// segment's first tile also publishes mean / rstd / var for the backward pass and the running statistics.
// ------------------------------------------------------------------------------------------------
struct BnActP {
    const float* H; float* Act; const float* part; float* mean; float* rstd; float* var;
    const float* rmean; const float* rvar; const float* gamma; const float* beta;
    int ncols, h, level, stack, layer, train;
    uint32_t seed, thr; float keep_scale; const uint32_t* seed_dev;
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_bn_act(const BnActP p) {
    __shared__ float s_n[4][64], s_m[4][64], s_q[4][64];
    __shared__ float s_mean[64], s_rstd[64];
    const int tile = blockIdx.x, c0 = blockIdx.y * 64;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const int cnt = p.r.seg_count[seg];
    const int t0 = p.r.seg_start[seg] / TILE_M, nt = (cnt + TILE_M - 1) / TILE_M;
    {   // ---- statistics of this segment for columns c0 .. c0+63 -----------------------------------
        const int cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
        const int cc = c0 + cl;
        bool act = cc < p.ncols;
        if (act && p.level >= 0) act = active_level(p.mp, p.level)[seg * MAX_TOWER + cc / p.h] != 0;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        if (act && cnt > 1 && p.train) {
            // six tiles of the group in flight per round (the loop used to pay one global round trip per tile before the apply
            // pass could start; a one-domain batch of 128 tiles takes six rounds): same arithmetic, same order
            for (int kb = 0; tg + 4 * kb < nt; kb += 6) {
                float nbv[6], mbv[6], m2v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int t = tg + 4 * (kb + k);
                    nbv[k] = 0.f; mbv[k] = 0.f; m2v[k] = 0.f;
                    if (t < nt) {
                        const float2 pp = *(const float2*)(p.part + ((int64_t)(t0 + t) * p.ncols + cc) * 2);
                        nbv[k] = (float)p.r.tile_valid[t0 + t]; mbv[k] = pp.x; m2v[k] = pp.y;
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (tg + 4 * (kb + k) < nt) {
                        const float nb = nbv[k];
                        const float tot = n + nb, delta = mbv[k] - mean;
                        mean += delta * (nb / tot);
                        m2 += m2v[k] + delta * delta * (n * nb / tot);
                        n = tot;
                    }
            }
        }
        s_n[tg][cl] = n; s_m[tg][cl] = mean; s_q[tg][cl] = m2;
        __syncthreads();
        if (tg == 0) {
            float rstd = 1.f, var = 0.f;
            mean = 0.f;
            if (act && cnt > 1) {
                if (p.train) {
                    n = 0.f; m2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float nb = s_n[k][cl];
                        if (nb > 0.f) {
                            const float tot = n + nb, delta = s_m[k][cl] - mean;
                            mean += delta * (nb / tot);
                            m2 += s_q[k][cl] + delta * delta * (n * nb / tot);
                            n = tot;
                        }
                    }
                    var = m2 / (float)cnt;
                } else {
                    mean = p.rmean[cc];
                    var = p.rvar[cc];
                }
                rstd = 1.0f / sqrtf(var + BN_EPS);
            }
            s_mean[cl] = mean; s_rstd[cl] = rstd;
            if (tile == t0 && cc < p.ncols) {
                const int64_t o = (int64_t)seg * p.ncols + cc;
                p.mean[o] = mean; p.rstd[o] = rstd; p.var[o] = var;
            }
        }
        __syncthreads();
    }
    // ---- apply: thread = (row group, float4) ----------------------------------------------------------
    const int rg = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int c = c0 + cq * 4;
    if (c >= p.ncols) return;
    const int g = c / p.h;
    bool act = true;
    if (p.level >= 0) act = active_level(p.mp, p.level)[seg * MAX_TOWER + g] != 0;
    const bool bn = cnt > 1;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu[i] = s_mean[cq * 4 + i]; rs[i] = s_rstd[cq * 4 + i]; ga[i] = p.gamma[c + i]; be[i] = p.beta[c + i]; }
    const uint32_t site = (uint32_t)((p.stack * 8 + p.layer) * 64 + g);
    const int cg = c - g * p.h;
    for (int rr = rg; rr < TILE_M; rr += 16) {
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act && rr < nvalid) {
            const float4 hv = *(const float4*)(p.H + row * p.ncols + c);
            float y[4] = {hv.x, hv.y, hv.z, hv.w};
            if (bn) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (y[i] - mu[i]) * rs[i] * ga[i] + be[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = y[i] > 0.f ? y[i] : 0.f;
            if (p.train && p.thr) {
                const uint32_t key = drop_row_key(drop_seed_of(p.seed, p.seed_dev), (uint32_t)p.r.row_sample[row]);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = drop_keep(key, site, (uint32_t)(cg + i), p.thr) ? y[i] * p.keep_scale : 0.f;
            }
            out = make_float4(y[0], y[1], y[2], y[3]);
        }
        *(float4*)(p.Act + row * p.ncols + c) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// MMoE mix (aread.py:152-153): In0[row][t][:] = sum_k softmax(glogE[row][t][:])_k * X[row][k][:]
// ------------------------------------------------------------------------------------------------
struct Mix0P {
    const float* glog; int ld_g; const float* X; float* In0;
    int n_t, n_exp, h;      // X: [rows][n_exp*h], In0: [rows][n_t*h]
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_mix0(const Mix0P p) {
    const int h4 = p.h >> 2, per_row = p.n_t * h4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / per_row;
    const int rem = (int)(idx - row * per_row);
    const int t = rem / h4, c = (rem - t * h4) * 4;
    const int tile = (int)(row / TILE_M);
    if (tile >= p.r.n_tiles) return;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool act = (row - (int64_t)tile * TILE_M) < p.r.tile_valid[tile] && active_level(p.mp, 0)[seg * MAX_TOWER + t];
    if (act) {
        const float* gl = p.glog + row * p.ld_g + t * p.n_exp;
        float mx = gl[0];
        for (int k = 1; k < p.n_exp; ++k) mx = fmaxf(mx, gl[k]);
        float den = 0.f;
        for (int k = 0; k < p.n_exp; ++k) den += __expf(gl[k] - mx);
        for (int k = 0; k < p.n_exp; ++k) {
            const float w = __expf(gl[k] - mx) / den;
            const float4 x = *(const float4*)(p.X + row * (p.n_exp * p.h) + k * p.h + c);
            out.x += w * x.x; out.y += w * x.y; out.z += w * x.z; out.w += w * x.w;
        }
    }
    *(float4*)(p.In0 + row * (p.n_t * p.h) + t * p.h + c) = out;
}

// ------------------------------------------------------------------------------------------------
// masked gate mix of level l >= 1 (aread.py:282-295): one thread per (row, tower t).
// Also emits per-tile sums of gate*mask for the HEMP gate statistics.
// ------------------------------------------------------------------------------------------------
struct MixLP {
    const float* glog; int ld_g, goff;        // gate logits of this level start at column goff
    const float* prev; float* In;             // prev: [rows][n_src*w]  In: [rows][n_t*w]
    int n_src, n_t, w, level, mask_off;
    float* gate_part;                          // nullable: [n_tiles*SUB][ld_g] per-slot sums of gate*mask
    RowsP r; ModeP mp;
};

typedef unsigned long long tf_u64;
// Data-tagged hand-off granules (MI355X guide, Guideline 16 R2): one naturally aligned 8-byte {tag, value} word written by ONE
// relaxed agent-scope (sc1) store and read by relaxed sc1 loads until the tag matches -- the data is the flag: no drain, no
// counter, no poll.  A (mean, M2) resp. (sum, sum) pair is two granules.  The tag words are zeroed before every step.
#define TF_TAG 0x5A17u
__device__ __forceinline__ void tf_put_tagged(tf_u64* g, float a, float b) {
    __hip_atomic_store(g, ((tf_u64)TF_TAG << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((tf_u64)TF_TAG << 32) | __float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool tf_get_tagged(const tf_u64* g, float& a, float& b) {
    const tf_u64 x0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const tf_u64 x1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __uint_as_float((unsigned)x0); b = __uint_as_float((unsigned)x1);
    return (unsigned)(x0 >> 32) == TF_TAG && (unsigned)(x1 >> 32) == TF_TAG;
}


__device__ __forceinline__ void gate_weights(const float* gl, int n_src, const uint8_t* mk, int n_t, int t, int mode,
                                             float* a, float* am, float* ah, float* S) {
    float mx = gl[0];
    for (int s = 1; s < n_src; ++s) mx = fmaxf(mx, gl[s]);
    float den = 0.f;
    for (int s = 0; s < n_src; ++s) { a[s] = __expf(gl[s] - mx); den += a[s]; }
    float sum = 0.f;
    for (int s = 0; s < n_src; ++s) {
        a[s] = a[s] / den;
        am[s] = (mode == 1 || mk[s * n_t + t]) ? a[s] : 0.f;
        sum += am[s];
    }
    if (mode == 1) { for (int s = 0; s < n_src; ++s) ah[s] = a[s]; *S = 1.f; }
    else { *S = sum + GATE_EPS; for (int s = 0; s < n_src; ++s) ah[s] = am[s] / *S; }
}

__global__ __launch_bounds__(256) void k_mixl(const MixLP p) {
    // block = SUB_ROWS rows of one tile; threads loop over (row, t) pairs
    __shared__ float s_gate[SUB_ROWS][MAX_TOWER * MAX_TOWER / 2 + 1];
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const uint8_t* act = active_level(p.mp, p.level) + seg * MAX_TOWER;
    const uint8_t* mk = p.mp.masks ? p.mp.masks + (size_t)p.mp.seg_dom[seg] * p.mp.edge_count + p.mask_off : nullptr;
    const int ngate = p.n_t * p.n_src;
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_t; it += 256) {
        const int rl = it / p.n_t, t = it - rl * p.n_t, rr = r_lo + rl;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float* dst = p.In + row * (p.n_t * p.w) + t * p.w;
        const bool on = rr < nvalid && act[t];
        float a[MAX_TOWER], am[MAX_TOWER], ah[MAX_TOWER], S;
        if (on) gate_weights(p.glog + row * p.ld_g + p.goff + t * p.n_src, p.n_src, mk, p.n_t, t, p.mp.mode, a, am, ah, &S);
        if (p.gate_part && ngate <= MAX_TOWER * MAX_TOWER / 2)
            for (int s = 0; s < p.n_src; ++s) s_gate[rl][t * p.n_src + s] = on ? am[s] : 0.f;
        const float* src = p.prev + row * (p.n_src * p.w);
        for (int c = 0; c < p.w; c += 4) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on)
                for (int s = 0; s < p.n_src; ++s) {
                    const float4 x = *(const float4*)(src + s * p.w + c);
                    o.x += ah[s] * x.x; o.y += ah[s] * x.y; o.z += ah[s] * x.z; o.w += ah[s] * x.w;
                }
            *(float4*)(dst + c) = o;
        }
    }
    if (p.gate_part && ngate <= MAX_TOWER * MAX_TOWER / 2) {
        __syncthreads();
        for (int gcol = threadIdx.x; gcol < ngate; gcol += 256) {
            float s = 0.f;
            for (int rl = 0; rl < SUB_ROWS; ++rl) s += s_gate[rl][gcol];
            p.gate_part[(int64_t)blockIdx.x * p.ld_g + p.goff + gcol] = s;
        }
    }
}

// gate statistics: mean over the segment of gate*mask  -> gate_stats[seg][gate_rows]
__global__ __launch_bounds__(256) void k_gate_stats(const float* gate_part, int ld_g, int gate_rows, float* out, RowsP r) {
    const int seg = blockIdx.x;
    const int cnt = r.seg_count[seg];
    const int t0 = r.seg_start[seg] / TILE_M, nt = (cnt + TILE_M - 1) / TILE_M;
    for (int c = threadIdx.x; c < gate_rows; c += 256) {
        float s = 0.f;
        for (int t = 0; t < nt * SUB; ++t) s += gate_part[(int64_t)(t0 * SUB + t) * ld_g + c];
        out[(int64_t)seg * gate_rows + c] = cnt > 0 ? s / (float)cnt : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// heads: z = cn.v[:D] + lin + tower_out.v[D:], p = sigmoid(z)  (aread.py:304-310), fused bagging BCE
// (run.py:672-677) and its gradient.  One thread per (row, head).
// ------------------------------------------------------------------------------------------------
struct HeadsP {
    const float* hc; const float* lin; const float* act; const float* head_w; int head_ld, D, h, n_heads, ld_h;
    float* z; float* prob; float* dz; float* probs_out; int64_t B;
    const float* y; const float* seg_weight; const float* dprobs; float* loss_part; int level;
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(256) void k_heads_fwd(const HeadsP p) {
    __shared__ float s_loss[256];
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const uint8_t* act = active_level(p.mp, p.level) + seg * MAX_TOWER;
    const float cnt = (float)p.r.seg_count[seg];
    const float kact = (float)p.mp.kact[seg];
    float loss = 0.f;
    for (int it = threadIdx.x; it < SUB_ROWS * p.n_heads; it += 256) {
        const int rr = r_lo + it / p.n_heads, i = it % p.n_heads;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        const bool on = rr < nvalid && act[i];
        float z = 0.f, pr = 0.f, dz = 0.f;
        if (on) {
            z = p.hc[row * p.ld_h + i] + p.lin[row];
            const float* a = p.act + row * (p.n_heads * p.h) + i * p.h;
            const float* v = p.head_w + (int64_t)i * p.head_ld + p.D;
            for (int c = 0; c < p.h; ++c) z += a[c] * v[c];
            pr = 1.0f / (1.0f + __expf(-z));
            const int b = p.r.row_sample[row];
            if (p.probs_out) p.probs_out[(int64_t)i * p.B + b] = pr;
            if (p.y) {
                const float yv = p.y[b];
                const float lp = fmaxf(__logf(pr), -100.f), lq = fmaxf(__logf(1.0f - pr), -100.f);
                loss += -(yv * lp + (1.0f - yv) * lq) / (cnt * kact);
            }
        }
        p.z[row * p.ld_h + i] = z;
        p.prob[row * p.ld_h + i] = pr;
        if (p.y) {                                    // fused loss gradient (BCELoss backward * sigmoid backward)
            if (on) {
                const float wseg = p.seg_weight ? p.seg_weight[seg] : 1.f;
                const float dp = wseg / (cnt * kact) * (pr - p.y[p.r.row_sample[row]]) / fmaxf((1.0f - pr) * pr, 1e-12f);
                dz = dp * pr * (1.0f - pr);
            }
            p.dz[row * p.ld_h + i] = dz;
        }
    }
    if (p.y)                                          // pad columns of dz
        for (int it = threadIdx.x; it < SUB_ROWS * (p.ld_h - p.n_heads); it += 256) {
            const int rr = r_lo + it / (p.ld_h - p.n_heads), i = p.n_heads + it % (p.ld_h - p.n_heads);
            p.dz[((int64_t)tile * TILE_M + rr) * p.ld_h + i] = 0.f;
        }
    if (p.loss_part) {
        s_loss[threadIdx.x] = loss;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) s_loss[threadIdx.x] += s_loss[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) p.loss_part[blockIdx.x] = s_loss[0];
    }
}

// loss_out[0] = sum_seg w_seg * bag_seg, loss_out[1+seg] = bag_seg.  `fault` (nullable): the error word of the fused tower
// kernel -- a segment hand-off that gave up poisons the loss with NaN, so the failure is loud without a host round trip.
__global__ __launch_bounds__(64) void k_loss_finish(const float* loss_part, const float* seg_weight, float* loss_out, RowsP r,
                                                    const unsigned* fault) {
    __shared__ float s_bag[MAX_SEG];
    const int seg = threadIdx.x;
    float bag = 0.f;
    if (seg < r.n_seg) {
        const int cnt = r.seg_count[seg];
        const int t0 = r.seg_start[seg] / TILE_M, nt = (cnt + TILE_M - 1) / TILE_M;
        // (same summation order as a plain loop; eight independent loads in flight instead of one dependent chain)
        const float* lp = loss_part + t0 * SUB;
        const int n = nt * SUB;
        int t = 0;
        for (; t + 8 <= n; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = lp[t + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) bag += v[u];
        }
        for (; t < n; ++t) bag += lp[t];
        loss_out[1 + seg] = bag;
        bag *= seg_weight ? seg_weight[seg] : 1.f;
    }
    s_bag[seg] = bag;
    __syncthreads();
    if (seg == 0) {
        float tot = 0.f;
        for (int s = 0; s < r.n_seg; ++s) tot += s_bag[s];
        if (fault && *fault) tot = __builtin_nanf("");
        loss_out[0] = tot;
    }
}

// dz = dL/dz for every (row, head): from labels (fused loss) or from an external dL/dprobs.
__global__ __launch_bounds__(256) void k_heads_dz(const HeadsP p) {
    const int tile = blockIdx.x / SUB, r_lo = (blockIdx.x % SUB) * SUB_ROWS;
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;
    const int nvalid = p.r.tile_valid[tile];
    const uint8_t* act = active_level(p.mp, p.level) + seg * MAX_TOWER;
    const float cnt = (float)p.r.seg_count[seg], kact = (float)p.mp.kact[seg];
    const float wseg = p.seg_weight ? p.seg_weight[seg] : 1.f;
    for (int it = threadIdx.x; it < SUB_ROWS * p.ld_h; it += 256) {
        const int rr = r_lo + it / p.ld_h, i = it % p.ld_h;
        const int64_t row = (int64_t)tile * TILE_M + rr;
        float dz = 0.f;
        if (i < p.n_heads && rr < nvalid && act[i]) {
            const float pr = p.prob[row * p.ld_h + i];
            const int b = p.r.row_sample[row];
            float dp;
            if (p.dprobs) dp = p.dprobs[(int64_t)i * p.B + b];
            else dp = wseg / (cnt * kact) * (pr - p.y[b]) / fmaxf((1.0f - pr) * pr, 1e-12f);   // BCELoss backward
            dz = dp * pr * (1.0f - pr);                                                           // sigmoid backward
        }
        p.dz[row * p.ld_h + i] = dz;
    }
}
