// tower_fused_bwd.h -- the BACKWARD pass of the whole HEI tower pyramid in one launch (split-bf16 mode): the mirror of
// tower_fused.h.  heads backward -> per level, top down { [dropout/ReLU backward -> BatchNorm backward -> dgrad] x n_layers ->
// gate-mix backward (aread.py:282-295 differentiated) } -> MMoE mix backward.  It replaces k_heads_bwd, k_act_bwd,
// k_bn_bwd_apply, the tower dgrad GEMMs, k_mixl_bwd and k_mix0_bwd (22 dependent launches); the weight gradients, the
// bias / gamma / beta reductions and the gate-input GEMMs stay on the side stream and read exactly the buffers the
// layer-by-layer path writes: dH of every layer (LayerWs::dAct), the per-tile partials bpart / cpart, dglogT / dglogE, dlin,
// the head partials, and dX for the expert backward.
//
// One workgroup (TF_THREADS threads) owns one 64-row plan tile.  The BatchNorm backward needs, per column, the sums of
// dyhat and dyhat*xhat over the SEGMENT: the same in-kernel hand-off as the forward (data-tagged 8-byte granules, swept by the
// merging threads until every tag matches; bpart is written as well, for the reductions on the side stream).
// LDS: two fp32 row buffers D0 / D1 [64][ldd] that alternate between "current gradient" and "xhat / next gradient", the
// split-bf16 A image of dH for the dgrad MFMA, and small per-column arrays; gate scratch and merge scratch alias the A
// image region, the expert-output tile of the MMoE mix backward spans D1 + the A image region.
#pragma once
#include "tower_fused.h"

struct TBLayer {
    int n_t, in_w, out_w, ncols;        // towers, per-tower input / output width of the FORWARD layer, n_t*out_w
    int ks, nfr, pk;                    // dgrad: 32-wide k-steps over out_w, 16-wide fragments of in_w, stored 8-wide planes per k-step
    int stack, layer;
    const __bf16* wimg;                 // dgrad weight image, NF = 2: [n_t][ks][hi | lo][64*32] (n = input column, k = output column)
    const float* gamma; const float* beta;
    const float* H; const float* mean; const float* rstd;
    float* dH; float* bpart; float* cpart;
    tf_u64* tags;                       // [n_tiles][ncols][2] data-tagged (sum dyhat, sum dyhat*xhat) granules of the hand-off
    tf_u64* fin;                        // [MAX_SEG][ncols][2] the two sums of a long segment, published by the column's owner tile
};

struct TBwdP {
    int n_level, n_layers, train, mode, two_hop_nt, prio;
    uint32_t seed, thr; float keep_scale; const uint32_t* seed_dev;
    TBLayer L[AREAD_MAX_LEVEL][AREAD_MAX_LAYER];
    int n_t[AREAD_MAX_LEVEL], mask_off[AREAD_MAX_LEVEL], gate_off[AREAD_MAX_LEVEL];
    const float* prevAct[AREAD_MAX_LEVEL];               // l > 0: activations of level l-1's last layer [rows][n_t[l-1]*in_w(l)]
    const float* X; int n_exp, xw; float* dX;            // expert outputs [rows][n_exp*xw] and their gradient
    const float* glogE; float* dglogE; int ld_ge; const float* glogT; float* dglogT; int ld_gt;
    const float* dz; const float* actLast; const float* head_w; int head_ld, D, n_heads, ld_h, h_last;
    float* dlin; float* head_part; int64_t ld_hp;        // [n_tiles*SUB][ld_hp]: sub-block 0 carries the tile's sums
    unsigned* err;
    unsigned long long* stamps;
    int lds_d0, lds_d1, lds_aimg, ldd, ldx, first_buf;   // byte offsets, row strides (floats) of D0/D1 and of the X tile
    RowsP r; ModeP mp;
};

__global__ __launch_bounds__(TF_THREADS) void k_tower_bwd(const TBwdP p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    float* const D0 = (float*)(smem + p.lds_d0);
    float* const D1 = (float*)(smem + p.lds_d1);
    __bf16* const Aimg = (__bf16*)(smem + p.lds_aimg);
    float* const s_scr = (float*)(smem + p.lds_aimg);     // gate / merge scratch: alive only while the A image is dead
    __shared__ float s_m1[256], s_m2[256];
    __shared__ uint32_t s_key[TILE_M];
    __shared__ float s_tvb[4];
    __shared__ __attribute__((aligned(16))) __bf16 s_zero[8];
    const __bf16* zslot = s_zero;
    if (threadIdx.x < 8) s_zero[threadIdx.x] = (__bf16)0.f;
    (void)s_tvb;

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
    const float inv_n = 1.0f / (float)cnt;
    // the segment's edge-mask bytes and tower-active bytes in LDS: the gate phases read them per (row, tower, source) item
    __shared__ uint8_t s_mask[512];
    const uint8_t* gmasks = p.mp.masks ? p.mp.masks + (size_t)p.mp.seg_dom[seg] * p.mp.edge_count : nullptr;
    const bool mask_lds = gmasks && p.mp.edge_count <= 512;
    if (mask_lds) for (int i = threadIdx.x; i < p.mp.edge_count; i += TF_THREADS) s_mask[i] = gmasks[i];
    const uint8_t* masks = mask_lds ? s_mask : gmasks;
    const int ldd = p.ldd;
    const bool drop = p.train && p.thr;
    if (tid < TILE_M) s_key[tid] = drop ? drop_row_key(drop_seed_of(p.seed, p.seed_dev), (uint32_t)p.r.row_sample[row0 + tid]) : 0u;

    int n_stamp = 0;
#define TB_STAMP()                                                                                       \
    do {                                                                                                 \
        if (p.stamps && tid == 0 && n_stamp < 64) p.stamps[(size_t)tile * 64 + n_stamp] = __builtin_amdgcn_s_memrealtime(); \
        ++n_stamp;                                                                                       \
    } while (0)
    TB_STAMP();                                            // 0: start

    float* dcur = p.first_buf ? D1 : D0;                   // current gradient rows [64][ldd]
    float* dalt = p.first_buf ? D0 : D1;

    // ---------------- heads backward: dAct_last = dz * v_tail, dlin = sum_i dz, per-tile partial of dv_tail ---------------
    {
        const int LL = p.n_level - 1;
        const int ncols = p.n_heads * p.h_last;
        const TFActBits act = tf_act_bits(active_level(p.mp, LL) + seg * MAX_TOWER);
        // dz tile -> LDS; thread = (column quad, row group): dAct rows and the dv_tail partial (shuffle-reduced) together
        float* s_dz = s_scr;                                // [64][n_heads]
        for (int it = tid; it < TILE_M * p.n_heads; it += TF_THREADS) {
            const int m = it / p.n_heads, i = it - m * p.n_heads;
            s_dz[it] = (m < nvalid && act[i]) ? p.dz[(row0 + m) * p.ld_h + i] : 0.f;
        }
        __syncthreads();
        for (int m = tid; m < TILE_M; m += TF_THREADS) {
            float sum = 0.f;
            for (int i = 0; i < p.n_heads; ++i) sum += s_dz[m * p.n_heads + i];
            p.dlin[row0 + m] = sum;
        }
        {
            const int nq = ncols >> 2;                      // h_last % 4 == 0: a quad lies inside one head
            int R = 64;
            while (R * nq > TF_THREADS) R >>= 1;
            const int cq = tid / R, rg = tid - cq * R;
            const bool qon = cq < nq;
            const int c = cq * 4, i = qon ? c / p.h_last : 0;
            float4 hw = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qon) {
                const float* hp = p.head_w + (int64_t)i * p.head_ld + p.D + (c - i * p.h_last);
                hw = make_float4(hp[0], hp[1], hp[2], hp[3]);
            }
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            if (qon) {
                float4 av[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = rg + k * R;
                    av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < nvalid) av[k] = *(const float4*)(p.actLast + (row0 + m) * ncols + c);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = rg + k * R;
                    if (m < TILE_M) {
                        const float dzv = s_dz[m * p.n_heads + i];
                        *(float4*)(dcur + m * ldd + c) = make_float4(dzv * hw.x, dzv * hw.y, dzv * hw.z, dzv * hw.w);
                        s4[0] += dzv * av[k].x; s4[1] += dzv * av[k].y; s4[2] += dzv * av[k].z; s4[3] += dzv * av[k].w;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                for (int o = 1; o < R; o <<= 1) s4[e] += __shfl_xor(s4[e], o);
            if (qon && rg == 0)
                for (int q = 0; q < SUB; ++q)
                    *(float4*)(p.head_part + ((int64_t)tile * SUB + q) * p.ld_hp + c) =
                        q == 0 ? make_float4(s4[0], s4[1], s4[2], s4[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
    }
    TB_STAMP();                                            // heads

    for (int l = p.n_level - 1; l >= 0; --l) {
        const int n_t = p.n_t[l];
        const TFActBits act = tf_act_bits(active_level(p.mp, l) + seg * MAX_TOWER);
        for (int j = p.n_layers - 1; j >= 0; --j) {
            const TBLayer& L = p.L[l][j];
            const int ncols = L.ncols, out_w = L.out_w, in_w = L.in_w;
            // ---- A. dropout / ReLU backward -> dyhat (in place), xhat -> dalt, per-tile column sums -> bpart ----------------
            // thread = (column quad cq, row group rg); the R row groups of a quad are adjacent lanes (shuffle reduction)
            const int nq = ncols >> 2;
            int R = 64;
            while (R * nq > TF_THREADS) R >>= 1;
            const int cq = tid / R, rg = tid - cq * R;
            const bool qon = cq < nq;
            const int c = cq * 4;
            const int g = qon ? c / out_w : 0, cg = c - g * out_w;
            const bool aon = qon && act[g];
            const bool sync_stats = p.train && bn;
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f}, ga[4] = {1.f, 1.f, 1.f, 1.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
            if (aon) {
                const float4 m4 = *(const float4*)(L.mean + (int64_t)seg * ncols + c), r4 = *(const float4*)(L.rstd + (int64_t)seg * ncols + c);
                const float4 g4 = *(const float4*)(L.gamma + c), b4 = *(const float4*)(L.beta + c);
                mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w; rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
                ga[0] = g4.x; ga[1] = g4.y; ga[2] = g4.z; ga[3] = g4.w; be[0] = b4.x; be[1] = b4.y; be[2] = b4.z; be[3] = b4.w;
            }
            const uint32_t site = (uint32_t)((L.stack * 8 + L.layer) * 64 + g);
            float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
            if (qon) {
                float4 hv8[8];                               // R >= 8: at most 8 rows per thread; all H loads in flight first
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = rg + k * R;
                    hv8[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (aon && m < nvalid) hv8[k] = *(const float4*)(L.H + (row0 + m) * ncols + c);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int m = rg + k * R;
                    if (m >= TILE_M) continue;
                    float4 dyo = make_float4(0.f, 0.f, 0.f, 0.f), xho = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (aon && m < nvalid) {
                        const float4 dv = *(const float4*)(dcur + m * ldd + c);
                        const float4 hv = hv8[k];
                        float d[4] = {dv.x, dv.y, dv.z, dv.w}, xh[4];
                        const float hh[4] = {hv.x, hv.y, hv.z, hv.w};
                        const uint32_t key = s_key[m];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            xh[i] = (hh[i] - mu[i]) * rs[i];
                            const float y = bn ? xh[i] * ga[i] + be[i] : hh[i];
                            float dy = d[i];
                            if (drop) dy = drop_keep(key, site, (uint32_t)(cg + i), p.thr) ? dy * p.keep_scale : 0.f;
                            dy = y > 0.f ? dy : 0.f;
                            d[i] = dy;
                            a1[i] += dy;
                            a2[i] += dy * xh[i];
                        }
                        dyo = make_float4(d[0], d[1], d[2], d[3]);
                        xho = make_float4(xh[0], xh[1], xh[2], xh[3]);
                    }
                    *(float4*)(dcur + m * ldd + c) = dyo;
                    *(float4*)(dalt + m * ldd + c) = xho;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int o = 1; o < R; o <<= 1) { a1[i] += __shfl_xor(a1[i], o); a2[i] += __shfl_xor(a2[i], o); }
            if (qon && rg == 0) {
                float* o = L.bpart + ((int64_t)tile * ncols + c) * 2;      // (the bias / gamma / beta reductions read these later)
                *(float4*)o = make_float4(a1[0], a2[0], a1[1], a2[1]);
                *(float4*)(o + 4) = make_float4(a1[2], a2[2], a1[3], a2[3]);
                if (p.train && bn) {
                    tf_u64* tg = L.tags + ((int64_t)tile * ncols + c) * 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i) tf_put_tagged(tg + 2 * i, a1[i], a2[i]);
                }
            }
            TB_STAMP();                                      // act backward + arrive
            // ---- B. segment sums ---------------------------------------------------------------------------------------------
            if (sync_stats && nt > p.two_hop_nt) {
                TB_STAMP();                                  // poll
                // long segment: two hops (tower_fused.h): tile i owns the columns c = i (mod nt), one wave per owned column adds the nt
                // partial pairs (lane = tile, then a butterfly, lower lane first) and publishes the sums; every tile reads ncols pairs
                const int ti = tile - t0, wv = tid >> 6, ln = tid & 63;
                for (int cc = ti + nt * wv; cc < ncols; cc += nt * TF_WAVES) {       // (wave-uniform)
                    if (!act[cc / out_w]) continue;
                    float t1 = 0.f, t2 = 0.f;
                    for (int t = ln; t < nt; t += 64) {
                        float x1 = 0.f, x2 = 0.f;
                        for (unsigned spins = 0; !tf_get_tagged(L.tags + ((int64_t)(t0 + t) * ncols + cc) * 2, x1, x2);) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        t1 += x1; t2 += x2;
                    }
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const float u1 = __shfl_xor(t1, o), u2 = __shfl_xor(t2, o);
                        const bool hi = (ln & o) != 0;
                        t1 = hi ? u1 + t1 : t1 + u1;
                        t2 = hi ? u2 + t2 : t2 + u2;
                    }
                    if (ln == 0) tf_put_tagged(L.fin + ((int64_t)seg * ncols + cc) * 2, t1, t2);
                }
                for (int cc = tid; cc < ncols; cc += TF_THREADS) {
                    float t1 = 0.f, t2 = 0.f;
                    if (act[cc / out_w])
                        for (unsigned spins = 0; !tf_get_tagged(L.fin + ((int64_t)seg * ncols + cc) * 2, t1, t2);) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                    s_m1[cc] = t1 * inv_n; s_m2[cc] = t2 * inv_n;
                }
                __syncthreads();
            } else if (sync_stats) {
                TB_STAMP();                                  // poll
                // item = (column, tile group q of nqg): plain sums in tile order inside a group, groups combined in order
                const int nqg = 4 * ncols <= TF_THREADS ? 4 : 2 * ncols <= TF_THREADS ? 2 : 1;
                float* s_c1 = s_scr, * s_c2 = s_scr + 4 * 256;
                for (int base = 0; base < nqg * ncols; base += TF_THREADS) {
                    const int item = base + tid;
                    const int cc = item % ncols, q = item / ncols;
                    float t1 = 0.f, t2 = 0.f;
                    if (item < nqg * ncols && act[cc / out_w]) {
                        // TF_MERGE_Q partials of the item in flight per round (a one-domain batch of 128 tiles takes three rounds),
                        // summed in tile order whatever the round size
                        for (int ib = 0; q + nqg * ib < nt; ib += TF_MERGE_Q) {
                            float b1[TF_MERGE_Q], b2[TF_MERGE_Q];
                            bool have[TF_MERGE_Q];
#pragma unroll
                            for (int i = 0; i < TF_MERGE_Q; ++i) { b1[i] = 0.f; b2[i] = 0.f; have[i] = q + nqg * (ib + i) >= nt; }
                            for (unsigned spins = 0;;) {         // sweep this round's granules until every tag matches
                                bool all = true;
#pragma unroll
                                for (int i = 0; i < TF_MERGE_Q; ++i) {
                                    if (!have[i]) have[i] = tf_get_tagged(L.tags + ((int64_t)(t0 + q + nqg * (ib + i)) * ncols + cc) * 2, b1[i], b2[i]);
                                    all = all && have[i];
                                }
                                if (all) break;
                                __builtin_amdgcn_s_sleep(1);
                                if (++spins > TF_SPIN_LIMIT) { __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                            }
#pragma unroll
                            for (int i = 0; i < TF_MERGE_Q; ++i) { t1 += b1[i]; t2 += b2[i]; }
                        }
                    }
                    if (item < nqg * ncols) { s_c1[q * 256 + cc] = t1; s_c2[q * 256 + cc] = t2; }
                }
                __syncthreads();
                for (int cc = tid; cc < ncols; cc += TF_THREADS) {
                    float t1 = 0.f, t2 = 0.f;
                    for (int k = 0; k < nqg; ++k) { t1 += s_c1[k * 256 + cc]; t2 += s_c2[k * 256 + cc]; }
                    s_m1[cc] = t1 * inv_n; s_m2[cc] = t2 * inv_n;
                }
                __syncthreads();
            } else {
                __syncthreads();                             // dcur / dalt of phase A are complete
                TB_STAMP();                                  // (keeps the stamp count uniform)
            }
            TB_STAMP();                                      // merge
            // ---- C. BatchNorm backward: dH = gamma*rstd*(dyhat - s1/n - xhat*s2/n) -> workspace + LDS; column sums -> cpart ---
            float a3[4] = {0.f, 0.f, 0.f, 0.f};
            if (qon) {
                float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
                if (sync_stats) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { m1[i] = s_m1[c + i]; m2[i] = s_m2[c + i]; }
                }
                for (int m = rg; m < TILE_M; m += R) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (aon && m < nvalid) {
                        const float4 dv = *(const float4*)(dcur + m * ldd + c);
                        float d[4] = {dv.x, dv.y, dv.z, dv.w};
                        if (bn) {
                            const float4 xv = *(const float4*)(dalt + m * ldd + c);
                            const float xh[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) d[i] = ga[i] * rs[i] * (d[i] - m1[i] - xh[i] * m2[i]);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) a3[i] += d[i];
                        o = make_float4(d[0], d[1], d[2], d[3]);
                    }
                    *(float4*)(dcur + m * ldd + c) = o;
                    *(float4*)(L.dH + (row0 + m) * ncols + c) = o;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int o = 1; o < R; o <<= 1) a3[i] += __shfl_xor(a3[i], o);
            if (qon && rg == 0) *(float4*)(L.cpart + (int64_t)tile * ncols + c) = make_float4(a3[0], a3[1], a3[2], a3[3]);
            __syncthreads();
            TB_STAMP();                                      // apply
            // ---- D. A image of dH (K = out_w per tower), dgrad: d_in = dH . W -> dalt ------------------------------------
            const int ks = L.ks, nfr = L.nfr, pk = L.pk, bstride = 2 * pk * 512;
            {
                const int planes = ks * pk, g8 = out_w >> 3;
                for (int it = tid; it < TILE_M * n_t * planes; it += TF_THREADS) {
                    const int m = it / (n_t * planes), rem = it - m * (n_t * planes);
                    const int t = rem / planes, pl = rem - t * planes;
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (pl < g8) {
                        const float* src = dcur + m * ldd + t * out_w + pl * 8;
                        const float4 x0 = *(const float4*)src, x1 = *(const float4*)(src + 4);
                        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                    }
                    const int blk = t * ks + pl / pk;
                    tf_put8(Aimg + (size_t)blk * bstride, Aimg + (size_t)blk * bstride + pk * 512, bf3_off(TILE_M, m, pl % pk), v);
                }
                __syncthreads();
            }
            {
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
                // d_in rows: [64][n_t*in_w] into dalt (the xhat rows are dead)
#pragma unroll
                for (int u = 0; u < TF_MAX_UNITS; ++u) {
                    const int unit = wave + TF_WAVES * u;
                    if (unit >= n_units) continue;
                    const int cw = uf[u] * 16 + fk * 4;
                    if (cw >= in_w) continue;
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int m = mi * 16 + fr;
                        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (uon[u] && m < nvalid) o = make_float4(acc[u][mi][0], acc[u][mi][1], acc[u][mi][2], acc[u][mi][3]);
                        *(float4*)(dalt + m * ldd + ut[u] * in_w + cw) = o;
                    }
                }
            }
            __syncthreads();
            { float* t = dcur; dcur = dalt; dalt = t; }
            TB_STAMP();                                      // dgrad
        }
        // ---------------- level input backward: dcur = dIn[l] [64][n_t*w] -------------------------------------------------
        const int w = p.L[l][0].in_w;
        // gate backward of one (row, tower): softmax / mask renormalisation differentiated (aread.py:282-295); dah[s] = dIn . src_s.
        // Writes the logit gradients and leaves the forward's mixing weights in ahd (for the dprev pass).
        auto gate_bwd = [&](const float* gl, const float* dah, int n_src, bool masked, const uint8_t* mk, int t, float* dgl, float* ahd) {
            if (n_src <= 8) {                                // registers, every load in flight together
                float g[8], d[8];
                bool keep[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    g[s] = s < n_src ? gl[s] : -INFINITY;
                    d[s] = s < n_src ? dah[s] : 0.f;
                    keep[s] = s < n_src && (!masked || mk[s * n_t + t]);
                }
                float mx = g[0];
#pragma unroll
                for (int s = 1; s < 8; ++s) mx = s < n_src ? fmaxf(mx, g[s]) : mx;
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s) { g[s] = s < n_src ? __expf(g[s] - mx) : 0.f; den += g[s]; }
                float sum = 0.f, am[8];
                const float inv_den = 1.0f / den;
#pragma unroll
                for (int s = 0; s < 8; ++s) { g[s] = g[s] * inv_den; am[s] = keep[s] ? g[s] : 0.f; sum += am[s]; }
                const float S = masked ? sum + GATE_EPS : 1.f;
                const float inv_S = 1.0f / S;
                float dot_ah = 0.f, ah[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) { ah[s] = masked ? am[s] * inv_S : g[s]; dot_ah += d[s] * ah[s]; }
                float dot_a = 0.f, da[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    da[s] = !masked ? d[s] : keep[s] ? (d[s] - dot_ah) * inv_S : 0.f;
                    dot_a += s < n_src ? da[s] * g[s] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < n_src) { dgl[s] = g[s] * (da[s] - dot_a); ahd[s] = ah[s]; }
                return;
            }
            float a[MAX_TOWER], am[MAX_TOWER], ah[MAX_TOWER], da[MAX_TOWER], S = 1.f;
            gate_weights(gl, n_src, mk, n_t, t, masked ? 0 : 1, a, am, ah, &S);
            float dot_ah = 0.f;
            for (int s = 0; s < n_src; ++s) dot_ah += dah[s] * ah[s];
            float dot_a = 0.f;
            for (int s = 0; s < n_src; ++s) {
                da[s] = !masked ? dah[s] : mk[s * n_t + t] ? (dah[s] - dot_ah) / S : 0.f;
                dot_a += da[s] * a[s];
            }
            for (int s = 0; s < n_src; ++s) { dgl[s] = a[s] * (da[s] - dot_a); ahd[s] = ah[s]; }
        };
        // level l > 0: sources = towers of level l-1 (their activations); level 0: sources = experts (MMoE, plain softmax)
        const int n_src = l > 0 ? p.n_t[l - 1] : p.n_exp;
        const int wsrc = n_src * w, ngate = n_t * n_src;
        const bool masked = l > 0 && p.mp.mode != 1;
        const uint8_t* mk = masked ? masks + p.mask_off[l] : nullptr;
        const float* src_g = l > 0 ? p.prevAct[l] : p.X;
        float* srcs = l > 0 ? dalt : D1;                     // source rows [64][lds]: level 0 spans D1 + the A image region
        const int lds_ = l > 0 ? ldd : p.ldx;
        float* s_ah = l > 0 ? s_scr : srcs + TILE_M * lds_;  // [64][ngate] mixing weights, then [64][ngate] dot products
        float* s_dah = s_ah + TILE_M * ngate;
        for (int i = tid; i < TILE_M * (wsrc >> 2); i += TF_THREADS) {
            const int m = i / (wsrc >> 2), c4 = i - m * (wsrc >> 2);
            *(float4*)(srcs + m * lds_ + c4 * 4) = *(const float4*)(src_g + (row0 + m) * wsrc + c4 * 4);
        }
        __syncthreads();
        TB_STAMP();                                          // mix: sources staged
        // dah[m][t][s] = dIn[m][t][:] . src[m][s][:], one lane per product with the ROW as the fastest lane index: a wave reads
        // 64 different rows of one (tower, source) pair, rows are ldd = 4 (mod 32) floats apart -> at most 2-way bank conflicts
        // (tower-fastest order put 12 towers 64 B apart on two bank groups: 9 us instead of 3 at the last level; spreading a
        // product over w/4 lanes with shuffles was slower still: 13 us)
        for (int it = tid; it < TILE_M * ngate; it += TF_THREADS) {
            const int m = it & (TILE_M - 1), rem = it >> 6;
            const int t = rem / n_src, sidx = rem - t * n_src;
            float accd = 0.f;
            if (m < nvalid && act[t]) {
                const float* din = dcur + m * ldd + t * w;
                const float* sp = srcs + m * lds_ + sidx * w;
                for (int cc = 0; cc < w; cc += 4) {
                    const float4 d4 = *(const float4*)(din + cc), x4 = *(const float4*)(sp + cc);
                    accd += d4.x * x4.x + d4.y * x4.y + d4.z * x4.z + d4.w * x4.w;
                }
            }
            s_dah[m * ngate + rem] = accd;
        }
        __syncthreads();
        TB_STAMP();                                          // mix: dot products
        for (int it = tid; it < TILE_M * n_t; it += TF_THREADS) {
            const int m = it / n_t, t = it - m * n_t;
            const int64_t row = row0 + m;
            float* dgl = l > 0 ? p.dglogT + row * p.ld_gt + p.gate_off[l] + t * n_src : p.dglogE + row * p.ld_ge + t * n_src;
            float* ahd = s_ah + m * ngate + t * n_src;
            if (!(m < nvalid && act[t])) {
                for (int sidx = 0; sidx < n_src; ++sidx) { dgl[sidx] = 0.f; ahd[sidx] = 0.f; }
                continue;
            }
            const float* gl = l > 0 ? p.glogT + row * p.ld_gt + p.gate_off[l] + t * n_src : p.glogE + row * p.ld_ge + t * n_src;
            gate_bwd(gl, s_dah + m * ngate + t * n_src, n_src, masked, mk, t, dgl, ahd);
        }
        __syncthreads();
        TB_STAMP();                                          // mix: gate backward
        // d_src[m][s][:] = sum_t ah[m][t][s] * dIn[m][t][:]  -> over the source rows (level > 0: the next dcur) or dX
        const int w4 = w >> 2;
        for (int it = tid; it < TILE_M * n_src * w4; it += TF_THREADS) {
            const int m = it / (n_src * w4), rem = it - m * (n_src * w4);
            const int sidx = rem / w4, cc = (rem - sidx * w4) * 4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < nvalid)
                for (int t = 0; t < n_t; ++t) {
                    const float wgt = s_ah[m * ngate + t * n_src + sidx];
                    if (wgt != 0.f) {
                        const float4 x = *(const float4*)(dcur + m * ldd + t * w + cc);
                        o.x += wgt * x.x; o.y += wgt * x.y; o.z += wgt * x.z; o.w += wgt * x.w;
                    }
                }
            if (l > 0) *(float4*)(srcs + m * lds_ + sidx * w + cc) = o;
            else *(float4*)(p.dX + (row0 + m) * wsrc + sidx * w + cc) = o;
        }
        if (l > 0) {
            __syncthreads();
            { float* t = dcur; dcur = dalt; dalt = t; }
        }
        TB_STAMP();                                          // mix backward
    }
#undef TB_STAMP
}
