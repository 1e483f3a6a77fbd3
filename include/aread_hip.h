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
 *   - `stream` is a hipStream_t passed as void*; kernels are launched on it and nowhere else.
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
 * Plan buffer layout (int32 words), sized by aread_plan_words(B, n_seg):
 *   [0] B  [1] n_seg  [2] rows_padded (device-computed)  [3] n_tiles (device-computed)  [4..15] reserved
 *   seg_count[AREAD_MAX_SEG]  seg_start[AREAD_MAX_SEG]      (start row, tile aligned)
 *   tile_seg[max_tiles]  (-1 = unused tile)   tile_valid[max_tiles] (valid rows in the tile)
 *   row_sample[max_rows] (sample index of a padded row, -1 = padding)   sample_row[B]
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
 * by table row and summed in a fixed order by a wavefront segmented reduction; no float atomics.
 * table_grad must already hold the value to accumulate onto (zeros, or the dense L2 term).
 * sample_row (optional): row of dout that holds sample b (the plan's sample_row array); NULL = b.
 * ws: workspace of aread_embed_bwd_ws_bytes(B, f_in, E) bytes. */
int64_t aread_embed_bwd_ws_bytes(int64_t B, int f_in, int E);
int aread_embed_bwd(const int32_t* x, int64_t B, int f_in, const int32_t* offsets,
                    int64_t n_table_rows, int E, int n_onehot, int n_mh_fields, int seq_len, int pool,
                    const int32_t* sample_row, const float* dout, float* table_grad,
                    void* ws, void* stream);

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
int aread_l2_finish(const float* partial, int n_partial, float l2, float* loss_out, int accumulate,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AREAD_HIP_H */
