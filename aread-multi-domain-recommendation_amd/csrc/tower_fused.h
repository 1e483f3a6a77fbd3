// tower_fused.h -- the whole HEI tower pyramid of the FORWARD pass in one launch (split-bf16 mode):
//   MMoE mix (aread.py:152-153) -> per level { masked gate mix (aread.py:282-295), [Linear -> BatchNorm -> ReLU -> Dropout] x
//   n_layers (layer.py:203-229) } -> heads + sigmoid + bagging BCE and its gradient (aread.py:304-310, run.py:672-677).
// It replaces k_mix0, k_mixl, the tower GEMMs, k_bn_act and k_heads_fwd: 19 dependent launches of 5-12 us each for 7 % of
// the step's FLOPs become one launch whose BatchNorm statistics points are segment-scoped hand-offs INSIDE the kernel.
//
// One workgroup (512 threads, 8 waves) owns one 64-row plan tile for every tower of every level; a tile lies inside one
// segment (domain), so the edge mask and the "tower is active" predicate are workgroup-uniform.  Per layer:
//   A image : the layer input of all towers as split-bf16 (hi, lo) tiles in LDS, per (tower, 32-wide k-step) one block in
//             the layout of gemm.h::bf3_off (conflict-free ds_read_b128 fragments); K is zero-padded to 32
//   weights : fragments straight from the pre-tiled images k_prep_wimg wrote for this step (L2-resident, 139 KB in all)
//   MFMA    : v_mfma_f32_16x16x32_bf16 x3 (hi*hi + hi*lo + lo*hi), operands swapped (weights as the A input) so that a
//             lane holds 4 consecutive columns of one row; a "unit" = (tower, 16-column fragment) x 64 rows, units are
//             dealt round-robin to the 8 waves, so the column statistics of a unit never leave its wave
//   stats   : (mean, M2) of the tile per column -> two data-tagged 8-byte granules {tag, value} per column, each ONE relaxed
//             agent-scope (sc1) store; every workgroup of the segment then merges the partials itself (Chan, tile order): a
//             thread sweeps the granules of its (column, tile group) item with sc1 loads until every tag matches -- MI355X
//             guide, Guideline 16 R2: the data is the flag, no drain / counter / poll / barrier; the sweep is bounded and
//             raises an error word instead of hanging; the tags are zeroed before every step (aread_forward, side stream)
//   apply   : normalise + ReLU + dropout in registers; H (pre-BN) and Act go to the workspace exactly where the backward
//             pass expects them; Act also stays in LDS (fp32) for the next A image, the next level's mix, or the heads.
// All workgroups of a segment must be resident together: the launcher only takes this path when the tile count fits the
// CU count (one workgroup per CU, ~150 KB of LDS); otherwise the layer-by-layer path of dense.hip runs.
#pragma once
#include "dense_kernels.h"
#include "gemm.h"

#ifndef TF_THREADS
#define TF_THREADS 512                  // measured: 256 -> 152 us, 512 -> 117 us, 1024 -> 127 us (spills) for the BASELINE batch
#endif
#define TF_WAVES (TF_THREADS / 64)
#define TF_DIRECT_IMG 0                 // next A image written from the normalise registers: measured 0.4 us slower per layer
#define TF_MAX_UNITS ((16 + TF_WAVES - 1) / TF_WAVES)   // units per wave and layer (16 in all)
#define TF_SPIN_LIMIT (1u << 22)
#define TF_MAX_SEG_TILES 256           // tiles of one segment whose row counts are cached in LDS
#define TF_MERGE_Q 12                  // partial loads in flight per (column, tile group): covers segments of <= 24 tiles at two groups
#define TF_TWO_HOP_NT 32               // default of AREAD_TWO_HOP_NT: segments of more tiles merge their statistics in two hops (owner per column, then all read the result)

// the "tower t of this level is active for the tile's segment" bytes as a register bit mask: act[t] in an inner loop was a
// dependent global load per item (the gate / mix / dot-product phases of both kernels spent most of their time waiting for it)
struct TFActBits {
    unsigned b;
    __device__ __forceinline__ bool operator[](int t) const { return (b >> t) & 1u; }
};
__device__ __forceinline__ TFActBits tf_act_bits(const uint8_t* a) {      // a: MAX_TOWER (16) bytes, 16-byte aligned
    static_assert(MAX_TOWER == 16, "one uint4 load");
    const uint4 v = *(const uint4*)a;
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    TFActBits r; r.b = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.b |= (((w[k] >> (8 * j)) & 0xFFu) ? 1u : 0u) << (4 * k + j);
    return r;
}

struct TFLayer {
    int n_t, in_w, out_w, ncols;        // towers of the level, per-tower input / output width, n_t*out_w
    int ks, nfr, pk;                    // 32-wide k-steps, 16-wide column fragments per tower, stored 8-wide planes per k-step
                                        // (pk = 4, or in_w/8 when in_w < 32: the A image keeps no all-zero planes)
    int stack, layer;                   // dropout site ids
    const __bf16* wimg;                 // forward weight image (gemm_wide.h), NF = 2: [n_t][ks][hi | lo][64*32]
    const float* bias; const float* gamma; const float* beta;
    const float* rmean; const float* rvar;
    float* H; float* Act; float* part; float* mean; float* rstd; float* var;
    tf_u64* tags;                       // [n_tiles][ncols][2] data-tagged (mean, M2) granules of the in-kernel hand-off
    tf_u64* fin;                        // [MAX_SEG][ncols][2] data-tagged (mean, rstd) of a long segment, published by the column's owner tile
};

struct TFwdP {
    int n_level, n_layers, train, mode, two_hop_nt, prio;
    uint32_t seed, thr; float keep_scale; const uint32_t* seed_dev;
    TFLayer L[AREAD_MAX_LEVEL][AREAD_MAX_LAYER];
    int n_t[AREAD_MAX_LEVEL], mask_off[AREAD_MAX_LEVEL], gate_off[AREAD_MAX_LEVEL];
    const float* X; int n_exp, xw;                       // expert outputs [rows][n_exp*xw]
    const float* glogE; int ld_ge; const float* glogT; int ld_gt;
    float* In[AREAD_MAX_LEVEL];                          // level inputs [rows][n_t*in_w] (the backward reads them)
    float* gate_part;                                    // nullable: [n_tiles*SUB][ld_gt]
    const float* hc; const float* lin; const float* head_w; int head_ld, D, n_heads, ld_h, h_last;
    float* z; float* prob; float* dz; float* probs_out; int64_t B;
    const float* y; const float* seg_weight; float* loss_part;
    unsigned* err;                                       // set to 1 when a bounded spin gives up
    unsigned long long* stamps;                          // diagnostics only (nullable): [n_tiles][64] s_memrealtime at phase ends
    int lds_aimg, lds_actf, lds_gate, max_ngate;         // byte offsets of the LDS regions; widest n_t*n_src
    RowsP r; ModeP mp;
};

__device__ __forceinline__ void tf_store_sc1(float* p, float a, float b) {
    union { float f[2]; tf_u64 u; } v;
    v.f[0] = a; v.f[1] = b;
    __hip_atomic_store((tf_u64*)p, v.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tf_load_sc1(const float* p, float& a, float& b) {
    union { float f[2]; tf_u64 u; } v;
    v.u = __hip_atomic_load((const tf_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = v.f[0]; b = v.f[1];
}

// (hi, lo) split of 8 consecutive floats into one 16-byte slot of each image
__device__ __forceinline__ void tf_put8(__bf16* hi, __bf16* lo, int off, const float (&x)[8]) {
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        bf16x2 h2, l2;
        bf3_split2(x[e], x[e + 1], h2, l2);
        h[e] = h2[0]; h[e + 1] = h2[1];
        l[e] = l2[0]; l[e + 1] = l2[1];
    }
    *(bf16x8*)(hi + off) = h;
    *(bf16x8*)(lo + off) = l;
}

__global__ __launch_bounds__(TF_THREADS) void k_tower_fwd(const TFwdP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    __bf16* Aimg = (__bf16*)(smem + p.lds_aimg);          // [(tower, k-step)][hi | lo][64*32]
    float* actf = (float*)(smem + p.lds_actf);            // [64][<= 256]: activations of the last finished layer, fp32
    float* s_gate = (float*)(smem + p.lds_gate);          // [64][n_t*n_src] renormalised gates of a level transition
    float* s_gam = s_gate + TILE_M * p.max_ngate;         // [64][n_t*n_src] masked, un-renormalised gates (statistics)
    __shared__ float s_mean[256], s_rstd[256];
    __shared__ float s_red[TF_THREADS];
    __shared__ float s_cn[4][256], s_cm[4][256], s_cq[4][256];      // merge scratch: per tile quarter (count, mean, M2)
    __shared__ float s_tv[TF_MAX_SEG_TILES];
    __shared__ __attribute__((aligned(16))) __bf16 s_zero[8];
    const __bf16* zslot = s_zero;
    if (threadIdx.x < 8) s_zero[threadIdx.x] = (__bf16)0.f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int tile = blockIdx.x;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    const int seg = p.r.tile_seg[tile];
    if (seg < 0) return;                                   // unused tile: takes part in no hand-off
    const int nvalid = p.r.tile_valid[tile];
    const int cnt = p.r.seg_count[seg];
    const int t0 = p.r.seg_start[seg] / TILE_M, nt = (cnt + TILE_M - 1) / TILE_M;
    const int64_t row0 = (int64_t)tile * TILE_M;
    const bool bn = cnt > 1;
    // the segment's edge-mask bytes in LDS: the gate phases read them per (row, tower, source) item
    __shared__ uint8_t s_mask[512];
    const uint8_t* gmasks = p.mp.masks ? p.mp.masks + (size_t)p.mp.seg_dom[seg] * p.mp.edge_count : nullptr;
    const bool mask_lds = gmasks && p.mp.edge_count <= 512;
    if (mask_lds) for (int i = threadIdx.x; i < p.mp.edge_count; i += TF_THREADS) s_mask[i] = gmasks[i];
    const uint8_t* masks = mask_lds ? s_mask : gmasks;

    // dropout keys of this lane's four rows (m = mi*16 + fr), rows-per-tile of the segment's tiles: read once
    uint32_t dkey[4] = {0u, 0u, 0u, 0u};
    if (p.train && p.thr) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int sm = p.r.row_sample[row0 + mi * 16 + fr];
            dkey[mi] = drop_row_key(drop_seed_of(p.seed, p.seed_dev), (uint32_t)sm);
        }
    }
    for (int t = tid; t < nt && t < TF_MAX_SEG_TILES; t += TF_THREADS) s_tv[t] = (float)p.r.tile_valid[t0 + t];

    int n_stamp = 0;
#define TF_STAMP()                                                                                       \
    do {                                                                                                 \
        if (p.stamps && tid == 0 && n_stamp < 64) p.stamps[(size_t)tile * 64 + n_stamp] = __builtin_amdgcn_s_memrealtime(); \
        ++n_stamp;                                                                                       \
    } while (0)
    // gates of level l's input mix: one thread per (row, tower) -> s_gate (renormalised) / s_gam (masked, HEMP statistics).
    // They depend on the gate logits and masks only, so levels > 0 are computed between the arrive and the poll of the
    // previous level's last hand-off, where the workgroup would otherwise idle.
    int gates_ready = -1;
    auto compute_gates = [&](int l) {
        const int n_t = p.L[l][0].n_t;
        const TFActBits act = tf_act_bits(active_level(p.mp, l) + seg * MAX_TOWER);
        const int n_src = l == 0 ? p.n_exp : p.n_t[l - 1];
        const int ngate = n_t * n_src;
        for (int it = tid; it < TILE_M * n_t; it += TF_THREADS) {
            const int m = it / n_t, t = it - m * n_t;
            const int64_t row = row0 + m;
            const bool on = m < nvalid && act[t];
            // no per-thread arrays (they would live in scratch at this register pressure): the few logits are re-read
            // and the softmax terms recomputed with the same operations as gate_weights()
            float* gdst = s_gate + m * ngate + t * n_src;
            float* sdst = s_gam + m * ngate + t * n_src;
            const bool want_am = l > 0 && p.gate_part;      // un-renormalised masked gate: HEMP statistics (aread.py:290-295)
            if (!on) {
                for (int s = 0; s < n_src; ++s) { gdst[s] = 0.f; if (want_am) sdst[s] = 0.f; }
                continue;
            }
            const float* gl = l == 0 ? p.glogE + row * p.ld_ge + t * n_src : p.glogT + row * p.ld_gt + p.gate_off[l] + t * n_src;
            const bool masked = l > 0 && p.mp.mode != 1;
            const uint8_t* mk = masked ? masks + p.mask_off[l] : nullptr;
            if (n_src <= 8) {
                // up to 8 sources: logits and mask bytes in flight together, fully unrolled (registers, no dependent load chain)
                float g[8];
                bool keep[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    g[s] = s < n_src ? gl[s] : -INFINITY;
                    keep[s] = s < n_src && (!masked || mk[s * n_t + t]);
                }
                float mx = g[0];
#pragma unroll
                for (int s = 1; s < 8; ++s) mx = s < n_src ? fmaxf(mx, g[s]) : mx;
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) { g[s] = s < n_src ? __expf(g[s] - mx) : 0.f; den += s < n_src ? g[s] : 0.f; }
                float sum = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) { g[s] = keep[s] ? g[s] / den : 0.f; sum += s < n_src ? g[s] : 0.f; }
                const float S = sum + GATE_EPS;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < n_src) {
                        gdst[s] = masked ? g[s] / S : g[s];
                        if (want_am) sdst[s] = g[s];
                    }
                continue;
            }
            float mx = gl[0];
            for (int s = 1; s < n_src; ++s) mx = fmaxf(mx, gl[s]);
            float den = 0.f;
            for (int s = 0; s < n_src; ++s) den += __expf(gl[s] - mx);
            if (!masked) {                                  // MMoE / unmasked mode: plain softmax
                for (int s = 0; s < n_src; ++s) {
                    const float a = __expf(gl[s] - mx) / den;
                    gdst[s] = a;
                    if (want_am) sdst[s] = a;
                }
            } else {
                float sum = 0.f;
                for (int s = 0; s < n_src; ++s) sum += mk[s * n_t + t] ? __expf(gl[s] - mx) / den : 0.f;
                const float S = sum + GATE_EPS;
                for (int s = 0; s < n_src; ++s) {
                    const float am = mk[s * n_t + t] ? __expf(gl[s] - mx) / den : 0.f;
                    gdst[s] = am / S;
                    if (want_am) sdst[s] = am;
                }
            }
        }
        gates_ready = l;
    };
    TF_STAMP();                                            // 0: start
    int prev_cols = 0;                                     // width of actf (previous layer's n_t*out_w)
    for (int l = 0; l < p.n_level; ++l) {
        const TFLayer& L0 = p.L[l][0];
        const int n_t = L0.n_t, in_w = L0.in_w, ks0 = L0.ks;
        const TFActBits act = tf_act_bits(active_level(p.mp, l) + seg * MAX_TOWER);
        // ---------------- level input: gate mix of the previous level (or of the experts) -> In[l], A image ----------------
        {
            const int n_src = l == 0 ? p.n_exp : p.n_t[l - 1];
            const int ngate = n_t * n_src;
            if (gates_ready != l) compute_gates(l);          // (level 0; later levels were done while waiting for the segment)
            __syncthreads();
            TF_STAMP();                                      // gates
            if (l > 0 && p.gate_part)
                for (int gcol = tid; gcol < ngate; gcol += TF_THREADS) {
                    float sum = 0.f;
                    for (int m = 0; m < TILE_M; ++m) sum += s_gam[m * ngate + gcol];
                    for (int q = 0; q < SUB; ++q) p.gate_part[((int64_t)tile * SUB + q) * p.ld_gt + p.gate_off[l] + gcol] = q == 0 ? sum : 0.f;
                }
            // weighted sums: one thread per (row, tower, 8 columns)
            const int pk0 = L0.pk, bstride0 = 2 * pk0 * 512; // planes per k-step; elements per (tower, k-step) block [hi | lo]
            const int g8 = in_w >> 3;                        // 8-column groups per tower (in_w % 8 == 0)
            const int planes = ks0 * pk0;                    // 16-byte slots per tower row in the A image
            for (int it = tid; it < TILE_M * n_t * planes; it += TF_THREADS) {
                const int m = it / (n_t * planes), rem = it - m * (n_t * planes);
                const int t = rem / planes, pl = rem - t * planes;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (pl < g8) {
                    const float* gw = s_gate + m * ngate + t * n_src;
                    if (l == 0) {
                        const float* src = p.X + (row0 + m) * (int64_t)(n_src * p.xw) + pl * 8;
                        if (n_src <= 4) {                    // every source row piece in flight at once
                            float4 x0[4], x1[4];
                            float w[4];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                w[s] = s < n_src ? gw[s] : 0.f;
                                x0[s] = x1[s] = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (w[s] != 0.f) { x0[s] = *(const float4*)(src + s * p.xw); x1[s] = *(const float4*)(src + s * p.xw + 4); }
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                if (w[s] != 0.f) {
                                    v[0] += w[s] * x0[s].x; v[1] += w[s] * x0[s].y; v[2] += w[s] * x0[s].z; v[3] += w[s] * x0[s].w;
                                    v[4] += w[s] * x1[s].x; v[5] += w[s] * x1[s].y; v[6] += w[s] * x1[s].z; v[7] += w[s] * x1[s].w;
                                }
                        } else
                        for (int s = 0; s < n_src; ++s) {
                            const float w = gw[s];
                            if (w != 0.f) {
                                const float4 x0 = *(const float4*)(src + s * p.xw), x1 = *(const float4*)(src + s * p.xw + 4);
                                v[0] += w * x0.x; v[1] += w * x0.y; v[2] += w * x0.z; v[3] += w * x0.w;
                                v[4] += w * x1.x; v[5] += w * x1.y; v[6] += w * x1.z; v[7] += w * x1.w;
                            }
                        }
                    } else {
                        const float* src = actf + m * prev_cols + pl * 8;
                        for (int s = 0; s < n_src; ++s) {
                            const float w = gw[s];
                            if (w != 0.f) {
                                const float4 x0 = *(const float4*)(src + s * in_w), x1 = *(const float4*)(src + s * in_w + 4);
                                v[0] += w * x0.x; v[1] += w * x0.y; v[2] += w * x0.z; v[3] += w * x0.w;
                                v[4] += w * x1.x; v[5] += w * x1.y; v[6] += w * x1.z; v[7] += w * x1.w;
                            }
                        }
                    }
                    float* dst = p.In[l] + (row0 + m) * (int64_t)(n_t * in_w) + t * in_w + pl * 8;
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                const int blk = t * ks0 + pl / pk0;
                tf_put8(Aimg + (size_t)blk * bstride0, Aimg + (size_t)blk * bstride0 + pk0 * 512, bf3_off(TILE_M, m, pl % pk0), v);
            }
            __syncthreads();
            TF_STAMP();                                      // mix + A image
        }
        // ---------------- the level's layers -----------------------------------------------------------------------------
        for (int j = 0; j < p.n_layers; ++j) {
            const TFLayer& L = p.L[l][j];
            const int ks = L.ks, nfr = L.nfr, out_w = L.out_w, ncols = L.ncols, pk = L.pk, bstride = 2 * pk * 512;
            const int n_units = n_t * nfr;
            f32x4 acc[TF_MAX_UNITS][4];
            int ut[TF_MAX_UNITS], uf[TF_MAX_UNITS];
            bool uon[TF_MAX_UNITS];
#pragma unroll
            for (int u = 0; u < TF_MAX_UNITS; ++u) {
                const int unit = wave + TF_WAVES * u;
                ut[u] = unit < n_units ? unit / nfr : 0;
                uf[u] = unit < n_units ? unit - ut[u] * nfr : 0;
                uon[u] = unit < n_units && act[ut[u]];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[u][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (ks <= 2) {
                // all weight fragments of the layer in flight before the first MFMA (two L2 round trips otherwise serialise per unit)
                bf16x8 wh[TF_MAX_UNITS][2], wl[TF_MAX_UNITS][2];
#pragma unroll
                for (int u = 0; u < TF_MAX_UNITS; ++u)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (uon[u] && s < ks) {
                            const __bf16* wb = L.wimg + ((size_t)(ut[u] * ks + s)) * 4096 + bf3_off(64, uf[u] * 16 + fr, fk);
                            wh[u][s] = *(const bf16x8*)wb; wl[u][s] = *(const bf16x8*)(wb + 2048);
                        }
#pragma unroll
                for (int u = 0; u < TF_MAX_UNITS; ++u) {
                    if (!uon[u]) continue;                   // wave-uniform
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        if (s >= ks) continue;
                        const bool stored = fk < pk;         // planes >= pk of a narrow input are all zero and not stored
                        const __bf16* ab = stored ? Aimg + ((size_t)(ut[u] * ks + s)) * bstride + bf3_off(TILE_M, fr, fk) : zslot;
                        const int mstep = stored ? 128 : 0, lo_off = stored ? pk * 512 : 0;
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            const bf16x8 ah = *(const bf16x8*)(ab + mi * mstep), al = *(const bf16x8*)(ab + lo_off + mi * mstep);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[u][s], ah, acc[u][mi], 0, 0, 0);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u][s], al, acc[u][mi], 0, 0, 0);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[u][s], ah, acc[u][mi], 0, 0, 0);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < TF_MAX_UNITS; ++u) {
                    if (!uon[u]) continue;                   // wave-uniform
                    for (int s = 0; s < ks; ++s) {
                        const __bf16* wb = L.wimg + ((size_t)(ut[u] * ks + s)) * 4096 + bf3_off(64, uf[u] * 16 + fr, fk);
                        const bf16x8 wh = *(const bf16x8*)wb, wl = *(const bf16x8*)(wb + 2048);
                        const bool stored = fk < pk;
                        const __bf16* ab = stored ? Aimg + ((size_t)(ut[u] * ks + s)) * bstride + bf3_off(TILE_M, fr, fk) : zslot;
                        const int mstep = stored ? 128 : 0, lo_off = stored ? pk * 512 : 0;
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            const bf16x8 ah = *(const bf16x8*)(ab + mi * mstep), al = *(const bf16x8*)(ab + lo_off + mi * mstep);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah, acc[u][mi], 0, 0, 0);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al, acc[u][mi], 0, 0, 0);
                            acc[u][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah, acc[u][mi], 0, 0, 0);
                        }
                    }
                }
            }
            TF_STAMP();                                      // MFMA
            // ---- bias; per-column (mean, M2) of the tile -> partials; arrive; only then H -> workspace -----------------------
            const bool sync_stats = p.train && bn;
            float4 gam[TF_MAX_UNITS], bet[TF_MAX_UNITS];     // BatchNorm affine of this lane's columns: in flight across the hand-off
#pragma unroll
            for (int u = 0; u < TF_MAX_UNITS; ++u) {
                gam[u] = make_float4(1.f, 1.f, 1.f, 1.f); bet[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!uon[u]) continue;
                const int cw = uf[u] * 16 + fk * 4;          // column inside the tower
                const bool cok = cw < out_w;
                const int col = ut[u] * out_w + cw;
                if (cok) {
                    const float4 b = *(const float4*)(L.bias + col);
                    gam[u] = *(const float4*)(L.gamma + col); bet[u] = *(const float4*)(L.beta + col);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) { acc[u][mi][0] += b.x; acc[u][mi][1] += b.y; acc[u][mi][2] += b.z; acc[u][mi][3] += b.w; }
                }
                if (sync_stats) {
                    const float inv = 1.0f / (float)nvalid;
                    float mean[4], m2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = 0.f;
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) s += (mi * 16 + fr < nvalid) ? acc[u][mi][r] : 0.f;
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
                        mean[r] = s * inv;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = 0.f;
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            const float d = acc[u][mi][r] - mean[r];
                            s += (mi * 16 + fr < nvalid) ? d * d : 0.f;
                        }
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
                        m2[r] = s;
                    }
                    if (fr == 0 && cok) {
                        tf_u64* o = L.tags + ((int64_t)tile * ncols + col) * 2;
#pragma unroll
                        for (int r = 0; r < 4; ++r) tf_put_tagged(o + 2 * r, mean[r], m2[r]);
                    }
                }
            }
            TF_STAMP();                                      // stats published
            if (j + 1 == p.n_layers && l + 1 < p.n_level) compute_gates(l + 1);
            // H (pre-BatchNorm, the backward reads it) drains while the other tiles of the segment arrive
#pragma unroll
            for (int u = 0; u < TF_MAX_UNITS; ++u) {
                if (!uon[u]) continue;
                const int cw = uf[u] * 16 + fk * 4;
                if (cw >= out_w) continue;
                const int col = ut[u] * out_w + cw;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    *(float4*)(L.H + (row0 + mi * 16 + fr) * ncols + col) = make_float4(acc[u][mi][0], acc[u][mi][1], acc[u][mi][2], acc[u][mi][3]);
            }
            // ---- statistics of the segment -------------------------------------------------------------------------------
            if (sync_stats && nt > p.two_hop_nt) {
                TF_STAMP();                                  // H stores issued + poll
                // Long segment (the reference's per-domain batches are ONE segment: 128 tiles at B = 8192): every tile merging every
                // tile's partials is nt^2 granule loads per column -- 27 us per layer at 128 tiles.  Two hops instead: tile i of the
                // segment owns the columns c = i (mod nt); one wave per owned column gathers the nt partials (lane = tile: Chan in
                // lane order, then a butterfly whose pairs are combined lower lane first, so every lane ends with the same bits) and
                // publishes the segment's (mean, rstd) as one tagged pair; then every tile picks up the ncols finished pairs.
                // All tiles use the owner's numbers, so the statistics are identical across the segment by construction.
                const int ti = tile - t0;
                for (int c = ti + nt * wave; c < ncols; c += nt * TF_WAVES) {        // (wave-uniform)
                    const bool on = act[c / out_w];
                    float n = 0.f, mean = 0.f, m2 = 0.f;
                    if (on) {
                        for (int t = lane; t < nt; t += 64) {
                            float mb = 0.f, qb = 0.f;
                            for (unsigned spins = 0; !tf_get_tagged(L.tags + ((int64_t)(t0 + t) * ncols + c) * 2, mb, qb);) {
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            }
                            const float nb = t < TF_MAX_SEG_TILES ? s_tv[t] : (float)p.r.tile_valid[t0 + t];
                            const float tot = n + nb, delta = mb - mean, rt = __builtin_amdgcn_rcpf(tot);
                            mean += delta * (nb * rt);
                            m2 += qb + delta * delta * (n * nb * rt);
                            n = tot;
                        }
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const float n2 = __shfl_xor(n, o), me2 = __shfl_xor(mean, o), q2 = __shfl_xor(m2, o);
                            const bool hi = (lane & o) != 0;
                            const float nA = hi ? n2 : n, mA = hi ? me2 : mean, qA = hi ? q2 : m2;
                            const float nB = hi ? n : n2, mB = hi ? mean : me2, qB = hi ? m2 : q2;
                            const float tot = nA + nB;
                            mean = mA; m2 = qA;
                            if (nB > 0.f) {
                                const float delta = mB - mA, rt = __builtin_amdgcn_rcpf(tot);
                                mean = mA + delta * (nB * rt);
                                m2 = qA + qB + delta * delta * (nA * nB * rt);
                            }
                            n = tot;
                        }
                    }
                    if (lane == 0) {
                        float rstd = 1.f, var = 0.f;
                        if (on) {
                            var = m2 / (float)cnt;
                            rstd = 1.0f / sqrtf(var + BN_EPS);
                            tf_put_tagged(L.fin + ((int64_t)seg * ncols + c) * 2, mean, rstd);
                        }
                        const int64_t o = (int64_t)seg * ncols + c;
                        L.mean[o] = mean; L.rstd[o] = rstd; L.var[o] = var;
                    }
                }
                for (int c = tid; c < ncols; c += TF_THREADS) {
                    float mean = 0.f, rstd = 1.f;
                    if (act[c / out_w])
                        for (unsigned spins = 0; !tf_get_tagged(L.fin + ((int64_t)seg * ncols + c) * 2, mean, rstd);) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                    s_mean[c] = mean; s_rstd[c] = rstd;
                }
            } else if (sync_stats) {
                TF_STAMP();                                  // H stores issued + poll
                // merge the partials of the segment's tiles (Chan): item = (column, tile quarter q: tiles q, q+4, ...), every
                // load of an item in flight at once, then the four quarters are combined in order -- the order of k_bn_act
                // one item per thread: (column, tile group q of nq: tiles q, q+nq, ...), every partial of the item in flight at once
                const int nq = 4 * ncols <= TF_THREADS ? 4 : 2 * ncols <= TF_THREADS ? 2 : 1;
                for (int base = 0; base < nq * ncols; base += TF_THREADS) {
                    const int item = base + tid;
                    const int c = item % ncols, q = item / ncols;
                    float n = 0.f, mean = 0.f, m2 = 0.f;
                    if (item < nq * ncols && act[c / out_w]) {
                        // TF_MERGE_Q partials of the item in flight per round (one round covers segments of <= nq*TF_MERGE_Q tiles; a
                        // one-domain batch of 128 tiles takes three): merged in tile order whatever the round size
                        for (int ib = 0; q + nq * ib < nt; ib += TF_MERGE_Q) {
                            float mb[TF_MERGE_Q], qb[TF_MERGE_Q];
                            bool have[TF_MERGE_Q];
#pragma unroll
                            for (int i = 0; i < TF_MERGE_Q; ++i) { mb[i] = 0.f; qb[i] = 0.f; have[i] = q + nq * (ib + i) >= nt; }
                            for (unsigned spins = 0;;) {         // sweep this round's granules until every tag matches
                                bool all = true;
#pragma unroll
                                for (int i = 0; i < TF_MERGE_Q; ++i) {
                                    if (!have[i]) have[i] = tf_get_tagged(L.tags + ((int64_t)(t0 + q + nq * (ib + i)) * ncols + c) * 2, mb[i], qb[i]);
                                    all = all && have[i];
                                }
                                if (all) break;
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            }
#pragma unroll
                            for (int i = 0; i < TF_MERGE_Q; ++i) {
                                const int t = q + nq * (ib + i);
                                if (t < nt) {
                                    const float nb = t < TF_MAX_SEG_TILES ? s_tv[t] : (float)p.r.tile_valid[t0 + t];
                                    const float tot = n + nb, delta = mb[i] - mean, rt = __builtin_amdgcn_rcpf(tot);
                                    mean += delta * (nb * rt);
                                    m2 += qb[i] + delta * delta * (n * nb * rt);
                                    n = tot;
                                }
                            }
                        }
                    }
                    if (item < nq * ncols) { s_cn[q][c] = n; s_cm[q][c] = mean; s_cq[q][c] = m2; }
                }
                __syncthreads();
                for (int c = tid; c < ncols; c += TF_THREADS) {
                    float mean = 0.f, rstd = 1.f, var = 0.f;
                    if (act[c / out_w]) {
                        float n = 0.f, m2 = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float nb = k < nq ? s_cn[k][c] : 0.f;
                            if (nb > 0.f) {
                                const float tot = n + nb, delta = s_cm[k][c] - mean, rt = __builtin_amdgcn_rcpf(tot);
                                mean += delta * (nb * rt);
                                m2 += s_cq[k][c] + delta * delta * (n * nb * rt);
                                n = tot;
                            }
                        }
                        var = m2 / (float)cnt;
                        rstd = 1.0f / sqrtf(var + BN_EPS);
                    }
                    s_mean[c] = mean; s_rstd[c] = rstd;
                    if (tile == t0) {
                        const int64_t o = (int64_t)seg * ncols + c;
                        L.mean[o] = mean; L.rstd[o] = rstd; L.var[o] = var;
                    }
                }
            } else {
                for (int c = tid; c < ncols; c += TF_THREADS) {
                    float mean = 0.f, rstd = 1.f, var = 0.f;
                    if (bn && act[c / out_w]) {                          // eval mode: running statistics
                        mean = L.rmean[c]; var = L.rvar[c];
                        rstd = 1.0f / sqrtf(var + BN_EPS);
                    }
                    s_mean[c] = mean; s_rstd[c] = rstd;
                    if (tile == t0) {
                        const int64_t o = (int64_t)seg * ncols + c;
                        L.mean[o] = mean; L.rstd[o] = rstd; L.var[o] = var;
                    }
                }
            }
            __syncthreads();
            TF_STAMP();                                      // merge
            // ---- normalise + ReLU + dropout -> Act (workspace + LDS) ----------------------------------------------------
            // (a following layer of the same level whose A image has no padding planes gets it directly from the registers)
            bool direct_img = false;
            if (j + 1 < p.n_layers) {
                const TFLayer& N = p.L[l][j + 1];
                direct_img = TF_DIRECT_IMG && N.ks * N.pk == (N.in_w >> 3) && (out_w & 15) == 0 && N.in_w == out_w;
            }
#pragma unroll
            for (int u = 0; u < TF_MAX_UNITS; ++u) {
                const int unit = wave + TF_WAVES * u;
                if (unit >= n_units) continue;
                const int cw = uf[u] * 16 + fk * 4;
                if (cw >= out_w) continue;
                const int col = ut[u] * out_w + cw;
                float mu[4], rs[4];
                const float ga[4] = {gam[u].x, gam[u].y, gam[u].z, gam[u].w}, be[4] = {bet[u].x, bet[u].y, bet[u].z, bet[u].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) { mu[r] = s_mean[col + r]; rs[r] = s_rstd[col + r]; }
                const uint32_t site = (uint32_t)((L.stack * 8 + L.layer) * 64 + ut[u]);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int m = mi * 16 + fr;
                    float y[4] = {0.f, 0.f, 0.f, 0.f};
                    if (uon[u] && m < nvalid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[u][mi][r];
                            if (bn) v = (v - mu[r]) * rs[r] * ga[r] + be[r];
                            y[r] = v > 0.f ? v : 0.f;
                        }
                        if (p.train && p.thr) {
                            const uint32_t key = dkey[mi];
#pragma unroll
                            for (int r = 0; r < 4; ++r) y[r] = drop_keep(key, site, (uint32_t)(cw + r), p.thr) ? y[r] * p.keep_scale : 0.f;
                        }
                    }
                    const float4 o = make_float4(y[0], y[1], y[2], y[3]);
                    *(float4*)(L.Act + (row0 + m) * ncols + col) = o;
                    if (direct_img) {
                        // the next layer's A image straight from registers: the lane pair (fk, fk^1) holds 8 consecutive columns
                        float v8[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v8[r] = y[r]; v8[4 + r] = __shfl_xor(y[r], 16); }
                        if (!(fk & 1)) {
                            const TFLayer& N = p.L[l][j + 1];
                            const int pl = cw >> 3, blk = ut[u] * N.ks + pl / N.pk, nbs = 2 * N.pk * 512;
                            tf_put8(Aimg + (size_t)blk * nbs, Aimg + (size_t)blk * nbs + N.pk * 512, bf3_off(TILE_M, m, pl % N.pk), v8);
                        }
                    } else {
                        *(float4*)(actf + m * ncols + col) = o;
                    }
                }
            }
            prev_cols = ncols;
            __syncthreads();
            TF_STAMP();                                      // normalise + Act
            // ---- next layer of the same level: its A image is this layer's activation, tower by tower -------------------
            if (j + 1 < p.n_layers && !direct_img) {
                const TFLayer& N = p.L[l][j + 1];
                const int npk = N.pk, nbs = 2 * npk * 512;
                const int planes = N.ks * npk, g8 = N.in_w >> 3;
                for (int it = tid; it < TILE_M * n_t * planes; it += TF_THREADS) {
                    const int m = it / (n_t * planes), rem = it - m * (n_t * planes);
                    const int t = rem / planes, pl = rem - t * planes;
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (pl < g8) {
                        const float* src = actf + m * ncols + t * N.in_w + pl * 8;
                        const float4 x0 = *(const float4*)src, x1 = *(const float4*)(src + 4);
                        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                    }
                    const int blk = t * N.ks + pl / npk;
                    tf_put8(Aimg + (size_t)blk * nbs, Aimg + (size_t)blk * nbs + npk * 512, bf3_off(TILE_M, m, pl % npk), v);
                }
                __syncthreads();
            }
        }
    }
    // ---------------- heads: z = cn.v[:D] + lin + act.v[D:], sigmoid, bagging BCE and its gradient ------------------------
    {
        const int LL = p.n_level - 1;
        const TFActBits act = tf_act_bits(active_level(p.mp, LL) + seg * MAX_TOWER);
        const float cntf = (float)cnt, kact = (float)p.mp.kact[seg];
        const float wseg = p.seg_weight ? p.seg_weight[seg] : 1.f;
        float loss = 0.f;
        for (int it = tid; it < TILE_M * p.ld_h; it += TF_THREADS) {
            const int m = it / p.ld_h, i = it - m * p.ld_h;
            const int64_t row = row0 + m;
            float z = 0.f, pr = 0.f, dz = 0.f;
            if (i < p.n_heads && m < nvalid && act[i]) {
                z = p.hc[row * p.ld_h + i] + p.lin[row];
                const float* a = actf + m * prev_cols + i * p.h_last;
                const float* v = p.head_w + (int64_t)i * p.head_ld + p.D;
                for (int c = 0; c < p.h_last; ++c) z += a[c] * v[c];
                pr = 1.0f / (1.0f + __expf(-z));
                const int b = p.r.row_sample[row];
                if (p.probs_out) p.probs_out[(int64_t)i * p.B + b] = pr;
                if (p.y) {
                    const float yv = p.y[b];
                    const float lp = fmaxf(__logf(pr), -100.f), lq = fmaxf(__logf(1.0f - pr), -100.f);
                    loss += -(yv * lp + (1.0f - yv) * lq) / (cntf * kact);
                    const float dp = wseg / (cntf * kact) * (pr - yv) / fmaxf((1.0f - pr) * pr, 1e-12f);
                    dz = dp * pr * (1.0f - pr);
                }
            }
            if (i < p.n_heads) { p.z[row * p.ld_h + i] = z; p.prob[row * p.ld_h + i] = pr; }
            if (p.y) p.dz[row * p.ld_h + i] = dz;
        }
        if (p.loss_part) {
            s_red[tid] = loss;
            __syncthreads();
            for (int o = TF_THREADS / 2; o > 0; o >>= 1) {
                if (tid < o) s_red[tid] += s_red[tid + o];
                __syncthreads();
            }
            if (tid < SUB) p.loss_part[(int64_t)tile * SUB + tid] = tid == 0 ? s_red[0] : 0.f;
        }
    }
    TF_STAMP();                                            // heads
#undef TF_STAMP
}
