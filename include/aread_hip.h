/*
 * aread_hip.h -- C ABI of libaread_hip.so: the MI355X (gfx950) implementation of AREAD's
 * CTR forward/backward hot path.
 *
 * The reference has no FFI/plugin layer for this path: it sits behind torch.nn.Module.forward
 * (SURVEY.md 8b).  The entry points below are therefore what a ctypes binding added to the
 * reference's modules would call; each one cites the reference code it replaces
 * (paths relative to the reference repository).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns an int status: 0 = AREAD_OK, anything else is an error whose text
 *     is returned by aread_last_error() (thread-local).  Nothing throws across the boundary.
 *   - all pointers are DEVICE pointers unless the parameter is called *_host; the caller owns
 *     every buffer.  No function allocates, frees, synchronises or copies to the host, so any
 *     call sequence can be captured into a hipGraph.
 *   - `stream` is a hipStream_t passed as void*.  aread_forward/aread_backward additionally fork independent
 *     work (weight-gradient GEMMs, gate logits, reductions) onto one internal non-blocking side stream and join it
 *     back before returning (fork-join with events: legal inside stream capture); every other entry point
 *     launches on `stream` only.
 *   - row-major fp32 tensors, int32 indices (run.py:251-258 keeps ids as torch.int).
 */
#ifndef AREAD_HIP_H
#define AREAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AREAD_OK 0
#define AREAD_ERR_ARG 1      /* bad argument (shape, alignment, null pointer) */
#define AREAD_ERR_HIP 2      /* a HIP runtime call failed */
#define AREAD_ERR_UNSUPPORTED 3

#define AREAD_TILE_M 64      /* row-tile of every dense kernel; BN segments are padded to it */
#define AREAD_MAX_SEG 64     /* max BN segments (= domains) in one call */
#define AREAD_MAX_LEVEL 4
#define AREAD_MAX_LAYER 4

int aread_version(void);
const char* aread_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Row plan: bucket the samples of one call by BN segment (= domain) and pad every segment to a
 * multiple of AREAD_TILE_M rows, so that every 64-row tile of every dense kernel belongs to one
 * domain (wave-uniform masks, per-segment BatchNorm statistics).
 *
 * Replaces: the per-domain DataLoaders + Python loop over domains (run.py:310-353, 609-611) and the
 * host-side `if not this_level_active_tower[t]` branches (aread.py:272,309,320).
 *
 * Plan buffer layout (int32 words), sized by aread_plan_layout_get(B, n_seg).words:
 *   [0] B  [1] n_seg  [2] rows_padded (device-computed)  [3] n_tiles (device-computed)  [4..15] reserved
 *   seg_count[AREAD_MAX_SEG]  seg_start[AREAD_MAX_SEG]      (start row, tile aligned)
 *   tile_seg[max_tiles]  (-1 = unused tile)   tile_valid[max_tiles] (valid rows in the tile)
 *   row_sample[max_rows] (sample index of a padded row, -1 = padding)   sample_row[B]
 *   scratch of aread_plan_build (per-wave segment counts), not part of the contract
 * max_rows (aread_plan_layout.max_rows) is the host-known upper bound used for grid sizes.
 * ------------------------------------------------------------------------------------------- */
typedef struct aread_plan_layout {   /* host-side description of the plan buffer (int32 word offsets) */
    int64_t max_rows, max_tiles, words;
    int64_t off_seg_count, off_seg_start, off_tile_seg, off_tile_valid, off_row_sample, off_sample_row;
} aread_plan_layout;
int aread_plan_layout_get(int64_t B, int n_seg, aread_plan_layout* out_host);
/* seg_col < 0: the whole call is one segment (a single-domain batch, or 'wo_mask').
 * Otherwise segment id = x[b, seg_col] (must be in [0, n_seg)). Stable within a segment. */
int aread_plan_build(const int32_t* x, int64_t B, int f_in, int seg_col, int n_seg,
                     int32_t* plan, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse embedding lookup.  Replaces FeaturesEmbedding.forward (model/layer.py:160-183):
 *   bag g = x + offsets (int32), rows = table[g], history slots summed in slot order and divided by
 *   seq_len ('mean', padding included), one-hot fields first then pooled fields.
 * pool: 0 = no multi-hot pooling (all f_in columns are one-hot), 1 = 'sum', 2 = 'mean'.
 * row_sample == NULL: output row p = sample p and n_rows_out must equal B.  Otherwise output row p
 * holds sample row_sample[p] (the plan's row_sample array; -1 = padding row, written as zeros).
 * out: [n_rows_out, f_out*E];  bag_out (optional): int32 [B, f_in].
 * ------------------------------------------------------------------------------------------- */
int aread_embed_fwd(const int32_t* x, int64_t B, int f_in, const int32_t* offsets,
                    const float* table, int64_t n_table_rows, int E,
                    int n_onehot, int n_mh_fields, int seq_len, int pool,
                    const int32_t* row_sample, int64_t n_rows_out,
                    float* out, int32_t* bag_out, void* stream);

/* Backward of the lookup (autograd of layer.py:166-178): table_grad[g] += c * dout row, c = 1 for
 * one-hot fields, 1/seq_len for 'mean' history slots.  Deterministic: contributions are radix-sorted
 * by table row (hand-written LSD sort) and summed in a fixed order by a segmented reduction; no float atomics.
 * table_grad must already hold the value to accumulate onto (zeros, or the dense L2 term).
 * sample_row (optional): row of dout that holds sample b (the plan's sample_row array); NULL = b.
 * ws: workspace of aread_embed_bwd_ws_bytes(B, f_in, E) bytes. */
int64_t aread_embed_bwd_ws_bytes(int64_t B, int f_in, int E);
int aread_embed_bwd(const int32_t* x, int64_t B, int f_in, const int32_t* offsets,
                    int64_t n_table_rows, int E, int n_onehot, int n_mh_fields, int seq_len, int pool,
                    const int32_t* sample_row, const float* dout, float* table_grad,
                    void* ws, void* stream);

/* The same in two phases, so that the index sort (which depends only on the ids) can run on another stream
 * while the dense backward is still producing dout: aread_embed_bwd == sort followed by reduce. */
int aread_embed_bwd_sort(const int32_t* x, int64_t B, int f_in, const int32_t* offsets,
                         int64_t n_table_rows, int E, int n_onehot, int n_mh_fields, int seq_len, int pool,
                         const int32_t* sample_row, void* ws, void* stream);
int aread_embed_bwd_reduce(int64_t B, int f_in, int E, int seq_len, const float* dout, float* table_grad,
                           void* ws, void* stream);
/* The same with the gradient given as the sum of two buffers (dout2 may be NULL): row r contributes dout[r] + dout2[r].
 * aread_call.de_rw keeps the row-wise trunk's share of dL/de apart from the expert stack's. */
int aread_embed_bwd_reduce2(int64_t B, int f_in, int E, int seq_len, const float* dout, const float* dout2, float* table_grad,
                            void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Lookup routing for a row-sharded table (multi-GPU extension; no counterpart in the single-device
 * reference -- it sits between FeaturesEmbedding's index bag, layer.py:165-166, and the gather).
 * Global row g = x + offsets lives on rank g % n_ranks at local row g / n_ranks.  Deduplicates the batch's rows
 * and groups them by owner, without sorting (direct-addressed flag array over the key space):
 *   slot_out      [B*f_in]  lookup -> index of its row in the unique list
 *   uniq_rows_out [<= B*f_in] LOCAL row of every unique row, ordered by (owner, local row)
 *   edges_out     [n_ranks+1] unique rows of owner q are uniq_rows_out[edges[q] .. edges[q+1]); edges[n_ranks] = count
 * ws: aread_route_ws_bytes() bytes, ZERO-FILLED by the caller before the first call; every call leaves it zeroed
 * where it matters, so the same buffer is reused step after step.  Integer work, bit-exact.
 * keep_flags != 0 (n_ranks == 1 only): the "row was looked up" flags stay set for aread_adam_table_l2, which
 * consumes and clears them. */
int64_t aread_route_ws_bytes(int64_t n_table_rows, int n_ranks);
int aread_route_build(const int32_t* x, int64_t B, int f_in, const int32_t* offsets, int64_t n_table_rows,
                      int n_ranks, void* ws, int32_t* slot_out, int32_t* uniq_rows_out, int32_t* edges_out,
                      int keep_flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense L2 term of the embedding table.  Replaces the table part of
 * BaseModel.get_regularization_loss (model/layer.py:96-112, registered at layer.py:31):
 *   loss += l2 * sum(w^2);   grad = grad_scale * 2*l2*w   (written, not accumulated)
 * One streaming pass: 2*n*4 bytes of HBM traffic.  partial: float[aread_l2_partials()] block sums,
 * reduced in fixed order by aread_l2_finish into *loss_out (+= when accumulate != 0).
 * grad may be NULL (loss only, e.g. under no_grad) and partial may be NULL (gradient only).
 * ------------------------------------------------------------------------------------------- */
int aread_l2_partials(void);
int aread_l2_table(const float* w, int64_t n, float l2, float grad_scale, float* grad,
                   float* partial, void* stream);
/* The same pass on at most max_workgroups workgroups (0 = full width): a background sweep that leaves HBM bandwidth and
 * compute units to the kernels of the critical path (the fused step runs it beside the forward).  Bitwise the same
 * gradient and partial sums as aread_l2_table at any width. */
int aread_l2_table_throttled(const float* w, int64_t n, float l2, float grad_scale, float* grad,
                             float* partial, int max_workgroups, void* stream);
/* Gradient only, scaled by a DEVICE scalar: grad[i] = 2*l2*grad_scale_dev[0]*w[i].  The backward of get_regularization_loss
 * under autograd (layer.py:96-112): dL/dreg arrives as a device tensor, reading it on the host would stall the stream. */
int aread_l2_table_dev(const float* w, int64_t n, float l2, const float* grad_scale_dev, float* grad, void* stream);
int aread_l2_finish(const float* partial, int n_partial, float l2, float* loss_out, int accumulate,
                    void* stream);

/* ---------------------------------------------------------------------------------------------
 * Grouped fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products):
 *     C[g][m][n] (+)= sum_k A[g](m,k) * B[g](n,k) (+ bias[g][n])
 * a_kc / b_kc != 0: the operand is k-contiguous (element (r,k) at base + g*gs + r*ld + k);
 *           == 0: it is row-contiguous     (element (r,k) at base + g*gs + k*ld + r).
 * Building block of every Linear layer on the path: torch.nn.Linear inside MultiLayerPerceptron
 * (model/layer.py:209-229), the MMoE / tower gates (model/aread.py:96-99,111-114) and their autograd.
 * ld and group strides must be multiples of 4 floats, base pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int aread_gemm(const float* A, int64_t lda, int64_t a_gs, int a_kc,
               const float* B, int64_t ldb, int64_t b_gs, int b_kc,
               float* C, int64_t ldc, int64_t c_gs, const float* bias, int64_t bias_gs,
               int M, int N, int K, int G, int accumulate, void* stream);

/* The same product with both operands k-contiguous on the bf16 matrix cores, fp32 operands split on the fly
 * into hi + lo bf16 (3 products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate):
 * ~1e-6 relative error, 3/16 of the fp32-MFMA matrix-pipe time.  Used for the forward and dgrad GEMMs of the
 * expert / tower layers when aread_model_cfg.precision == 1. */
int aread_gemm_bf16x3(const float* A, int64_t lda, int64_t a_gs, const float* B, int64_t ldb, int64_t b_gs,
                      float* C, int64_t ldc, int64_t c_gs, const float* bias, int64_t bias_gs,
                      int M, int N, int K, int G, int accumulate, void* stream);
/* The same split-bf16 product for two ROW-contiguous operands (element (r,k) at base + g*gs + k*ld + r): the shape of
 * every weight gradient dW = dH^T X of the path (autograd of torch.nn.Linear, model/layer.py:209-229), K = batch rows.
 * The k-major tiles are staged untransposed and read with the CDNA4 transposing LDS read (ds_read_b64_tr_b16). */
int aread_gemm_bf16x3_rc(const float* A, int64_t lda, int64_t a_gs, const float* B, int64_t ldb, int64_t b_gs,
                         float* C, int64_t ldc, int64_t c_gs, int M, int N, int K, int G, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense part of the model: linear term, cross network, MMoE bottom, masked HEI tower pyramid, heads,
 * bagging loss -- forward and backward.  Replaces AREAD.forward / hier_tower_mask_forward
 * (model/aread.py:129-322), MultiLayerPerceptron.forward (model/layer.py:221-229),
 * CrossNetwork.forward (model/layer.py:529-537), FeaturesLinear (model/layer.py:115-126), the dense
 * terms of get_regularization_loss (model/layer.py:96-112) and the step closure (run.py:672-680).
 *
 * aread_model is a host-side handle that only holds layouts (it owns no device memory):
 *   params : one flat fp32 buffer with every trainable dense tensor (layout: aread_model_tensor)
 *   stats  : one flat fp32 buffer with the BatchNorm running_mean / running_var vectors
 *   nbt    : int64 num_batches_tracked counters, one per BatchNorm module
 *   ws     : workspace of aread_model_workspace_bytes(m, B, n_seg) bytes (activations, partials)
 * Every tensor of the reference's state_dict is a contiguous slice of params/stats/nbt.
 * ------------------------------------------------------------------------------------------- */
typedef struct aread_model_cfg {
    int32_t embed_dim, f_out, domain_field;       /* D = f_out*embed_dim; output-field index of the domain id */
    int32_t n_expert, n_expert_layers, expert_dims[AREAD_MAX_LAYER];
    int32_t n_level, n_tower[AREAD_MAX_LEVEL], n_tower_layers, tower_dims[AREAD_MAX_LEVEL][AREAD_MAX_LAYER];
    int32_t n_cross, n_domain;
    float dropout;
    float l2_linear, l2_dnn, l2_cross;
    int32_t precision;      /* 0: every GEMM on the exact fp32 MFMA (parity mode); 1: forward and dgrad GEMMs of the
                               expert/tower layers on split-bf16 (3 products, ~4e-6 rms relative), wgrad stays fp32 */
} aread_model_cfg;

typedef struct aread_tensor_desc {
    char name[96];          /* the reference's state_dict key */
    int32_t kind;           /* 0 = params, 1 = stats, 2 = nbt counter */
    int32_t ndim;
    int64_t offset;         /* element offset inside the buffer named by kind */
    int64_t shape[2];
    float l2;               /* L2 coefficient of this tensor in the regulariser (0 = not regularised) */
} aread_tensor_desc;

typedef struct aread_model aread_model;
int aread_model_create(const aread_model_cfg* cfg_host, aread_model** out_host);
void aread_model_destroy(aread_model* m);
int64_t aread_model_param_floats(const aread_model* m);
int64_t aread_model_stat_floats(const aread_model* m);
int aread_model_n_bn(const aread_model* m);
int aread_model_n_tensors(const aread_model* m);
int aread_model_tensor(const aread_model* m, int i, aread_tensor_desc* out_host);
int aread_model_edge_count(const aread_model* m);      /* bytes per domain in `masks` */
int aread_model_gate_rows(const aread_model* m);       /* sum_l n_l*n_{l-1}: columns of gate_stats */
int64_t aread_model_workspace_bytes(const aread_model* m, int64_t B, int n_seg);

typedef struct aread_call {
    int64_t B;
    int32_t n_seg;            /* 1: one BN segment (single-domain batch or wo_mask); else n_domain */
    int32_t mode;             /* 0 = masked HEI (domain_mask_bagging / domain_with_mask), 1 = wo_mask */
    int32_t train;            /* 1: batch statistics + dropout (model.train()); 0: running statistics */
    int32_t update_running;   /* 1: update running_mean/var/num_batches_tracked (train only) */
    int32_t domain;           /* n_seg == 1: domain_i of the call (selects the mask); ignored otherwise */
    uint32_t drop_seed;
    const int32_t* plan;      /* aread_plan_build(x, B, ..., n_seg) */
    const uint8_t* masks;     /* [n_domain][edge_count] bytes, level-major, row-major (mode 0) */
    const float* params;
    float* stats;
    int64_t* nbt;
    void* ws;
    float* probs;             /* out: [n_heads][B] in sample order; 0 where the head is inactive */
    float* gate_stats;        /* out (optional): [n_seg][gate_rows] mean over the segment of gate*mask */
    const float* y;           /* optional labels [B] (float 0/1): enables the fused loss + its gradient */
    const float* seg_weight;  /* optional [n_seg] weights w_d of the per-domain bagging losses (default 1) */
    float* loss_out;          /* optional out: [1 + n_seg]: sum_d w_d*bag_d, then bag_d per segment */
    int32_t async_tail;       /* bit 0 (aread_backward): return with de_out complete on `stream` but the parameter
                                 gradients still finishing on the model's internal side stream; the caller overlaps
                                 its own work and then calls aread_join(m, stream).
                                 bit 1 (aread_forward): return with probs complete on `stream` but loss_out and the
                                 running statistics still finishing on the side stream (a following aread_backward
                                 queues behind them; aread_join covers both) */
    /* Optional (aread_backward only): the embedding table's L2 pass (aread_l2_table_throttled + aread_l2_finish: grad =
     * 2*coef*w written to l2_grad, coef*sum w^2 to l2_reg_out[0]) issued by the backward itself on an internal stream at
     * the moment the latency-bound tower backward starts, so that its 356 MB of HBM traffic hides behind it instead of
     * competing with the bandwidth-bound kernels around the forward/backward boundary.  Complete on `stream` when
     * aread_backward returns (stream order).  l2_table == NULL: nothing is launched. */
    const float* l2_table; int64_t l2_n; float l2_coef; int32_t l2_workgroups;
    float* l2_grad; float* l2_partial; float* l2_reg_out;
    /* L2 coefficient per dense parameter (device vector like params; aread_model_l2_coef fills a host copy): read by
     * aread_prepare for the dense L2 terms at the head of the step (init_grads below). */
    const float* l2_dense_coef;
    /* aread_backward: non-zero = `grads` already holds an initial value (the fused step starts it from the dense L2 term,
     * aread_l2_dense_init) and every parameter gradient is ADDED to it; 0 = the buffer is cleared first. */
    int32_t grads_init;
    /* Optional (aread_prepare): together with l2_dense_coef, the dense L2 terms at the head of the step on the side stream:
     * init_grads[i] = 2*coef[i]*w[i] (written), init_reg_out[0] = sum coef*w^2 (257 floats).  Pass the same buffer as `grads`
     * to aread_backward with grads_init = 1. */
    float* init_grads; float* init_reg_out;
    /* Optional (aread_backward): where the row-wise trunk backward (linear term, cross network, gate inputs) writes ITS share
     * of dL/de_in, [plan.max_rows][D].  de_out then holds the expert stack's share only (gradient = de_out + de_rw; the
     * embedding backward aread_embed_bwd_reduce2 adds them on the fly) and the first expert layer's dgrad no longer waits for
     * the row-wise chain.  NULL: de_out holds the whole gradient. */
    float* de_rw;
    /* aread_forward, train == 0 only: non-zero = no backward will follow this forward (torch.no_grad evaluation, run.py:712-763):
     * the expert layers apply BatchNorm (running statistics) + ReLU in their GEMM epilogue and do not keep the pre-BatchNorm H. */
    int32_t inference;
    /* Optional: the dropout seed in DEVICE memory (one uint32).  When set, the kernels read it at run time and `drop_seed` is
     * ignored -- a step captured in a hipGraph then draws a new dropout mask on every replay if the caller changes the word
     * between replays (a kernel argument is baked into the capture).  The forward and the backward of one step must see the
     * same value. */
    const uint32_t* drop_seed_dev;
} aread_call;

/* Optional, before the row plan / gather of a step: queues the part of the forward's preparation that depends only on the
 * parameters, the masks and the call's scalars (mask tables, pre-tiled weight images, hand-off tag memsets, probs memset,
 * the backward's transposed weights) on the model's side stream, forked from `stream` at this point.  call_host->plan is
 * not read.  The next aread_forward on the same workspace then skips it.  (The reference has no counterpart: it rebuilds
 * its Python-side mask logic inside every forward, aread.py:263-322.) */
int aread_prepare(const aread_model* m, const aread_call* call_host, void* stream);
/* e_in: embedding output in plan order [plan.max_rows][D] (aread_embed_fwd with the plan's row_sample). */
int aread_forward(const aread_model* m, const aread_call* call_host, const float* e_in, void* stream);
/* Backward of the same call (the workspace must be untouched since aread_forward).
 * dprobs: gradient w.r.t. probs [n_heads][B] (sample order), or NULL to use the fused loss gradient
 * (requires call.y).  grads: flat buffer like params, OVERWRITTEN with the gradient.
 * de_out: [plan.max_rows][D], overwritten with the gradient w.r.t. e_in. */
int aread_backward(const aread_model* m, const aread_call* call_host, const float* e_in, const float* dprobs,
                   float* grads, float* de_out, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Stand-alone MLP block.  Replaces MultiLayerPerceptron (model/layer.py:203-229):
 *   [Linear -> BatchNorm1d -> ReLU -> Dropout] x n_layers (+ Linear(last, 1) when output_layer != 0),
 *   BatchNorm skipped for a one-row batch (layer.py:226).  One BN segment (the whole batch).
 * The handle is an aread_model whose tensor table uses the keys of that module's state_dict
 * ("layers.0.weight", "layers.1.running_mean", ...); params / stats / nbt / ws follow the same conventions.
 * plan: aread_plan_build(NULL, B, 0, -1, 1, ...).  x: [B, in_dim], out: [B, dims[last]] or [B, 1].
 * ------------------------------------------------------------------------------------------- */
typedef struct aread_mlp_cfg {
    int32_t in_dim, n_layers, dims[AREAD_MAX_LAYER], output_layer, precision;
    float dropout;
} aread_mlp_cfg;
int aread_mlp_create(const aread_mlp_cfg* cfg_host, aread_model** out_host);
int64_t aread_mlp_workspace_bytes(const aread_model* m, int64_t B);
typedef struct aread_mlp_call {
    int64_t B;
    int32_t train, update_running;
    uint32_t drop_seed;
    const int32_t* plan;
    const float* params;
    float* stats;
    int64_t* nbt;
    void* ws;
} aread_mlp_call;
int aread_mlp_forward(const aread_model* m, const aread_mlp_call* call_host, const float* x, float* out, void* stream);
/* grads: flat like params, overwritten; dx: [B, in_dim] or NULL. */
int aread_mlp_backward(const aread_model* m, const aread_mlp_call* call_host, const float* x, const float* dout,
                       float* grads, float* dx, void* stream);

/* Test / A-B switch of the launch strategy (results agree to rounding): key "fused_towers" (0 = one launch per tower
 * layer, 1 = the fused tower pyramid of csrc/tower_fused.h), "wide_gemm" (0 / 1 / 2, csrc/gemm_wide.h).  Process-wide. */
int aread_debug_set(const char* key, int value);
/* Diagnostics: "fused_fwd_calls" / "fused_bwd_calls" = launches of the fused tower kernels so far in this process (-1: unknown key). */
long long aread_debug_get(const char* key);
/* Diagnostics: after aread_debug_set("phase_events", 1) the forward / backward record events at their phase boundaries on
 * the caller's stream; this returns the elapsed GPU time (ms) between consecutive boundaries of the last call. */
int aread_debug_phase_times(float* out_ms, int n);
/* Measurement only (bench.py gather_roofline.achievable_us, tools/mem_roof.py): n_read table rows of E floats read through
 * pre-resolved row indices (eight in flight per lane, 16 bytes per lane) and n_write <= n_read output rows streamed out (sums
 * of consecutive reads) -- the gather's memory traffic without its id decoding, pooling and plan lookups. */
int aread_debug_gather_roof(const int32_t* rows, int64_t n_read, const float* table, int E, float* out, int64_t n_write, void* stream);
/* Makes `stream` wait for the model's internal side stream (see aread_call.async_tail). */
int aread_join(const aread_model* m, void* stream);
/* Dense L2 terms: loss_out[0] (+)= sum_i coef[i]*w[i]^2, grads[i] += 2*coef[i]*w[i] (grads may be NULL).
 * coef: device vector like params (aread_model_l2_coef fills a host copy).
 * loss_out must have room for 257 floats: [0] is the result, [1..256] is scratch for block partials. */
int aread_model_l2_coef(const aread_model* m, float* coef_host);
/* Test/debug introspection: float offset of a named workspace buffer (e.g. "cn", "ex0.H", "tw1.0.Act"), -1 if unknown. */
int64_t aread_debug_ws_offset(const aread_model* m, int64_t B, int n_seg, const char* name);
int aread_l2_dense(const float* params, const float* coef, int64_t n, float* grads, float* loss_out,
                   int accumulate, void* stream);
/* The same in one launch with the step's final scalar: total_out[0] = loss_in[0] + loss_out[0] (both optional).  The last
 * block to finish does the fixed-order sum of the block partials (a process-wide self-resetting ticket: one call at a time). */
int aread_l2_dense_total(const float* params, const float* coef, int64_t n, float* grads, float* loss_out,
                         int accumulate, const float* loss_in, float* total_out, void* stream);
/* The fused step's head: grads[i] = 2*coef[i]*w[i] (WRITTEN: the buffer the backward then adds every gradient to, see
 * aread_call.grads_init) and reg_out[0] = sum_i coef[i]*w[i]^2 (257 floats, as loss_out above).  get_regularization_loss's
 * dense half (layer.py:96-112) moved from the serial tail of the step to its start. */
int aread_l2_dense_init(const float* params, const float* coef, int64_t n, float* grads, float* reg_out, void* stream);
/* The fused step's last launch: reg[0] += reg_dense[0]; total[0] = loss[0] + reg[0]  (run.py:678 `loss = loss + reg`). */
int aread_step_total(const float* loss, const float* reg_dense, float* reg, float* total, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimizer.  torch.optim.Adam as the reference configures it (run.py:830-831: lr 1e-3,
 * betas (0.9, 0.99), eps 1e-8, COUPLED weight_decay 1e-8, no amsgrad); `step` is 1-based.
 *   aread_adam_step     : w, m, v updated in place from a materialised gradient g (flat buffer of n floats).
 *                         active (optional, uint8[n]): 0 = the element belongs to a tensor whose grad is None in
 *                         the reference (torch skips it entirely: no decay, no moment update).
 *   aread_adam_table_l2 : the embedding table with the dense L2 term of layer.py:31,96-112 folded in, no dense
 *                         gradient buffer: g = 2*l2*w + (row looked up ? g_rows[slot(row)] : 0).  route_ws /
 *                         uniq_rows / edges come from aread_route_build(n_ranks = 1, keep_flags = 1) on this batch,
 *                         g_rows [<= B*f_in, E] from aread_embed_bwd over the slot ids; all four NULL = L2 only.
 *                         partial (optional): float[aread_l2_partials()] block sums of w^2 BEFORE the update, for
 *                         aread_l2_finish (the step's loss value).  The flags in route_ws are cleared.
 *                         phase 0 = the whole table in one pass (after the backward);
 *                         phase 1 = only the rows the batch did NOT look up: needs nothing from the backward and
 *                                   touches no row the gather reads, so it overlaps the forward/backward on another
 *                                   stream (g_rows/uniq_rows/edges unused);
 *                         phase 2 = the looked-up rows only (after the backward; clears the flags; partial holds
 *                                   aread_adam_row_partials() entries).  Phases 1 + 2 == phase 0. */
typedef struct aread_adam_cfg {
    float lr, beta1, beta2, eps, weight_decay;
    int32_t step;
} aread_adam_cfg;
int aread_adam_step(float* w, const float* g, float* m, float* v, int64_t n, const uint8_t* active,
                    const aread_adam_cfg* cfg, void* stream);
int aread_adam_table_l2(float* w, float* m, float* v, int64_t n_rows, int E, void* route_ws,
                        const int32_t* uniq_rows, const int32_t* edges, const float* g_rows, float l2,
                        const aread_adam_cfg* cfg, int phase, float* partial, void* stream);
int aread_adam_row_partials(void);

#ifdef __cplusplus
}
#endif
#endif /* AREAD_HIP_H */
