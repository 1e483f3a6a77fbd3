// model.h -- host-side layouts of the dense model: flat parameter / statistics buffers and the
// per-call workspace.  Pure bookkeeping; no device memory is owned here.
#pragma once
#include <cstdlib>
#include <string>
#include <vector>
#include "common.h"

#define MAX_TOWER 16      // towers per level
#define MAX_CROSS 4

struct LayerL {              // one (grouped) Linear+BatchNorm layer of an MLP stack
    int G, in_dim, out_dim;  // G groups (experts / towers of a level)
    int ncols;               // G*out_dim
    int in_ld;               // leading dimension of the input activation buffer
    int64_t in_gs;           // per-group offset in the input buffer (0 = all groups share the input)
    int stack, layer;        // dropout site ids (stack 0 = experts, 1+l = tower level l)
    // params (floats into the flat buffer)
    int64_t w, b, gamma, beta;
    // stats buffer
    int64_t rmean, rvar;
    int nbt0;                // first num_batches_tracked counter (one per group)
};

struct LayerWs {             // workspace of one layer (float offsets)
    int64_t H, Act, dAct, part, mean, rstd, var, bpart, s12, cpart;
    int64_t wT;              // transposed weights [G][in][out] (split-bf16 dgrad wants k-contiguous operands)
    int64_t wimg_f, wimg_d;  // pre-tiled split-bf16 weight images for the wide GEMM (gemm_wide.h): forward, dgrad view
    int64_t tag_f = -1, tag_b = -1;   // tower layers: data-tagged hand-off granules of the fused kernels, [tiles][ncols][2] x 8 bytes
    int64_t fin_f = -1, fin_b = -1;   // the finished per-segment pairs of a long segment's two-hop merge, [MAX_SEG][ncols][2] x 8 bytes
};

struct StackL {
    int n_layers;
    LayerL L[AREAD_MAX_LAYER];
};

struct aread_model {
    aread_model_cfg cfg;
    int D, E, n_heads, h_last, head_ld;     // head_ld = D + h_last
    StackL experts;
    StackL towers[AREAD_MAX_LEVEL];
    // params
    int64_t lin_w, lin_b, cn_w, cn_b, gate_w, gate_b, group_emb, tgate_w, tgate_b, head_w;
    int64_t n_params, n_stats;
    int n_bn;
    // masks
    int edge_count;
    int mask_off[AREAD_MAX_LEVEL + 1];      // byte offset of each level inside one domain's mask
    int gate_rows;                           // sum_l n_l * n_{l-1}
    int gate_off[AREAD_MAX_LEVEL];           // first gate row of level l (l >= 1)
    int ld_ge, ld_gt, ld_h;                  // padded leading dims: MMoE gate logits, tower gate logits, heads
    std::vector<aread_tensor_desc> tensors;
    // side stream + event pool for fork-join concurrency inside one call (created on first use, per device)
    // stand-alone MLP handle (aread_mlp_create): experts stack with G = 1, optional Linear(last, 1)
    bool is_mlp = false;
    int mlp_in = 0, mlp_out_layer = 0;
    int64_t out_w = 0, out_b = 0;
    mutable hipStream_t side = nullptr;
    mutable hipStream_t side2 = nullptr;     // second fork-join stream: the row-wise trunk backward beside the expert backward
    mutable hipEvent_t ev[64] = {};
    mutable hipEvent_t ev_prep = nullptr;    // forward: the backward-only preparations (transposed weights, dgrad images) are done on the side stream
    mutable bool prep_pending = false;
    mutable bool side2_pending = false;      // aread_backward left parameter-gradient work on side2: aread_join waits for it too
    mutable hipEvent_t ev_join2 = nullptr;
    mutable bool fwd_tail_deferred = false;  // fused step (async_tail bit 1): the forward left its loss / running-statistics tail to aread_backward
    mutable const void* fwd_prepared = nullptr;   // aread_prepare queued the plan-independent side work of the next forward on this workspace
    mutable int n_ev = 0;
    mutable bool ab_tags_clean = false;    // k_act_bn_bwd's tags were zeroed by the last forward
    mutable bool bwd_tags_clean = false;   // the fused tower backward's hand-off tags were zeroed by the last forward and not used yet
};
int model_streams_init(const aread_model* m);

struct WsLayout {                            // float offsets into the workspace (computed per (B, n_seg))
    int64_t max_rows, n_tiles;
    int64_t cn, lin, xw, q, glogE, glogT, hc, z, prob, dz, dlin, dcn, dq, deg, dglogE, dglogT, grp, dgrp_part;
    int64_t In[AREAD_MAX_LEVEL], dIn[AREAD_MAX_LEVEL];
    LayerWs ex[AREAD_MAX_LAYER];
    LayerWs tw[AREAD_MAX_LEVEL][AREAD_MAX_LAYER];
    int64_t active;                          // bytes region (as float offset): [n_level][MAX_SEG][MAX_TOWER]
    int64_t kact, seg_dom;                   // ints: active heads per seg, domain of each seg
    int64_t loss_part, gate_part, rw_part, misc_part;
    int64_t cs_part_e, cs_part_t;            // per-tile column sums of the gate-logit gradients (gate bias gradients)
    int64_t tf_sync;                         // fused tower kernels: arrival counters [MAX_LEVEL*MAX_LAYER][MAX_SEG] x 2 (fwd, bwd) + error words
    int64_t tf_tags = -1, tf_tags_floats = 0, ab_tags_floats = 0;   // all tag_f / tag_b buffers, contiguous: zeroed by one memset per forward
    int64_t ab_sync = -1;         // k_act_bn_bwd arrival counters (-1: layout without them, e.g. the stand-alone MLP)
    int64_t slab_ex[AREAD_MAX_LAYER], slab_tw[AREAD_MAX_LEVEL][AREAD_MAX_LAYER], slab_head, slab_gate, slab_tgate;
    int64_t total;                           // floats
};

void ws_layout(const aread_model* m, int64_t B, int n_seg, WsLayout* w);
void mlp_ws_layout(const aread_model* m, int64_t B, WsLayout* w);

// split-K geometry of a wgrad GEMM (K = padded batch rows): enough slices to fill the chip
struct KSplit { int k_split, k_chunk; };
static inline KSplit wgrad_ksplit(int64_t rows, int G, int M, int N) {
    int tn = N > 32 ? 64 : (N > 16 ? 32 : 16);
    if (N > 64) tn = ((N + 95) / 96 * 96 < (N + 127) / 128 * 128) ? 96 : 128;
    const int64_t blocks_mn = (int64_t)G * ((M + 63) / 64) * ((N + tn - 1) / tn);
    // small weight gradients (towers, gates, heads) share ONE launch with a dozen others (k_gemm_bf3_rc_multi): ~128 workgroups
    // each fill the chip together, and 5-6x fewer slabs to write and to reduce than a chip-filling split of every one of them
    static int small_target = -1;                // AREAD_WGRAD_TARGET: A/B of the small weight gradients' workgroup count
    if (small_target < 0) { const char* e = getenv("AREAD_WGRAD_TARGET"); small_target = e ? atoi(e) : 128; }
    const int64_t target = (int64_t)G * M * N <= 32768 ? small_target : 768;
    int64_t want = (target + blocks_mn - 1) / blocks_mn;
    const int64_t max_split = rows / TILE_M;
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    int64_t chunk = ((rows + want - 1) / want + TILE_M - 1) / TILE_M * TILE_M;
    KSplit k;
    k.k_chunk = (int)chunk;
    k.k_split = (int)((rows + chunk - 1) / chunk);
    return k;
}
