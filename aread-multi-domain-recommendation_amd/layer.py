"""Host-side mirror of the reference's model/layer.py for the hot path: same class names,
constructor arguments, attributes, forward() signatures and state_dict keys; the math runs in
libaread_hip.so (hand-written HIP for gfx950) through the C ABI of include/aread_hip.h."""
import numpy as np
import torch
from torch import nn

from . import _lib as L

_POOL = {None: 0, "sum": 1, "mean": 2}


class _EmbedFn(torch.autograd.Function):
    """autograd node around aread_embed_fwd / aread_embed_bwd (dense table gradient, as nn.Embedding)."""

    @staticmethod
    def forward(ctx, table, x, mod, row_sample, sample_row, n_rows_out):
        L.require_device(table, x)
        L.require(x, torch.int32, "x (the reference keeps ids as torch.int, run.py:251-258)")
        L.require(table, torch.float32, "embedding table")
        B = x.shape[0]
        n_out = B if row_sample is None else int(n_rows_out)
        out = torch.empty((n_out, mod.output_dim0, mod.embed_dim), dtype=torch.float32, device=x.device)
        L.check(L.lib().aread_embed_fwd(L.ptr(x), B, x.shape[1], L.ptr(mod._offsets_dev(x.device)), L.ptr(table),
                                        table.shape[0], mod.embed_dim, mod.one_hot_field_num, mod.multi_hot_field_num,
                                        mod.seq_maxlen, mod._pool, L.ptr(row_sample), n_out, L.ptr(out), None,
                                        L.stream()))
        ctx.mod, ctx.sample_row = mod, sample_row
        ctx.save_for_backward(x)
        ctx.table_shape = tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        dout = dout.contiguous()
        grad = torch.zeros(ctx.table_shape, dtype=torch.float32, device=x.device)
        mod.scatter_grad(x, dout, grad, ctx.sample_row)
        return grad, None, None, None, None, None


class FeaturesEmbedding(nn.Module):
    """model/layer.py:129-183.  One shared table, per-field offsets, history slots share the itemid
    sub-table and are mean/sum pooled over seq_maxlen (padding included)."""

    def __init__(self, one_hot_field_dims, embed_dim, multi_hot_dict=None):
        super().__init__()
        one_hot_field_dims = [int(d) for d in one_hot_field_dims]
        self.multi_hot_flag = np.array(multi_hot_dict["multi_hot_flag"])
        self.one_hot_field_num = len(one_hot_field_dims)
        self.seq_maxlen = multi_hot_dict["seq_maxlen"]
        self.multi_hot_field_num = int(sum(self.multi_hot_flag)) // self.seq_maxlen
        self.multi_hot_method = multi_hot_dict["method"]
        if self.multi_hot_method not in {"sum", "mean", None}:
            raise ValueError(f"Invalid multi-hot method '{self.multi_hot_method}'. "
                             "Method must be 'mean', 'sum', or None.")
        pooled = self.multi_hot_method in {"sum", "mean"}
        self.output_dim0 = self.one_hot_field_num + (self.multi_hot_field_num if pooled
                                                     else int(sum(self.multi_hot_flag)))
        self.embed_dim = embed_dim
        self.embedding_dict = nn.Embedding(sum(one_hot_field_dims), embed_dim)
        self.offsets = np.array((0, *np.cumsum(one_hot_field_dims)[:-1]), dtype=np.longlong)
        if self.multi_hot_field_num > 0:
            mh = [self.offsets[multi_hot_dict["itemid_idx"]]] * int(sum(self.multi_hot_flag))
            self.offsets = np.concatenate((self.offsets, mh))
        n_flag = int(sum(self.multi_hot_flag))
        if n_flag and not (list(self.multi_hot_flag[:self.one_hot_field_num]) == [False] * self.one_hot_field_num
                           and all(self.multi_hot_flag[self.one_hot_field_num:])):
            raise ValueError("multi_hot_flag must list the one-hot columns first, then the history slots "
                             "(the layout run.py:254-258 produces)")
        self._pool = _POOL[self.multi_hot_method] if self.multi_hot_field_num > 0 else 0
        self._off_cache = {}
        self._bwd_ws = None

    # ---- device-side helpers ------------------------------------------------------------------
    def _offsets_dev(self, device):
        key = str(device)
        if key not in self._off_cache:
            self._off_cache[key] = torch.from_numpy(self.offsets.astype(np.int32)).to(device)
        return self._off_cache[key]

    def index_bag(self, x):
        """int32 bag g = x + offsets exactly as the kernel forms it (bit-exact parity target)."""
        L.require_device(x)
        bag = torch.empty_like(x)
        out = torch.empty((x.shape[0], self.output_dim0, self.embed_dim), dtype=torch.float32, device=x.device)
        w = self.embedding_dict.weight
        L.check(L.lib().aread_embed_fwd(L.ptr(x), x.shape[0], x.shape[1], L.ptr(self._offsets_dev(x.device)),
                                        L.ptr(w), w.shape[0], self.embed_dim, self.one_hot_field_num,
                                        self.multi_hot_field_num, self.seq_maxlen, self._pool, None, x.shape[0],
                                        L.ptr(out), L.ptr(bag), L.stream()))
        return bag

    def scatter_grad(self, x, dout, grad, sample_row=None):
        """grad[g] += c * dout rows (deterministic sort + segmented reduction)."""
        B, f_in = x.shape
        need = L.lib().aread_embed_bwd_ws_bytes(B, f_in, self.embed_dim)
        if need < 0:
            raise RuntimeError("aread_embed_bwd_ws_bytes failed")
        if self._bwd_ws is None or self._bwd_ws.numel() < need or self._bwd_ws.device != x.device:
            self._bwd_ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        L.check(L.lib().aread_embed_bwd(L.ptr(x), B, f_in, L.ptr(self._offsets_dev(x.device)), grad.shape[0],
                                        self.embed_dim, self.one_hot_field_num, self.multi_hot_field_num,
                                        self.seq_maxlen, self._pool, L.ptr(sample_row), L.ptr(dout), L.ptr(grad),
                                        L.ptr(self._bwd_ws), L.stream()))

    def _ws_for(self, x):
        B, f_in = x.shape
        need = L.lib().aread_embed_bwd_ws_bytes(B, f_in, self.embed_dim)
        if need < 0:
            raise RuntimeError("aread_embed_bwd_ws_bytes failed")
        if self._bwd_ws is None or self._bwd_ws.numel() < need or self._bwd_ws.device != x.device:
            self._bwd_ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        return self._bwd_ws

    def sort_lookups(self, x, sample_row=None):
        """phase 1 of scatter_grad: depends only on the ids, can run on a side stream."""
        B, f_in = x.shape
        L.check(L.lib().aread_embed_bwd_sort(L.ptr(x), B, f_in, L.ptr(self._offsets_dev(x.device)),
                                             self.embedding_dict.weight.shape[0], self.embed_dim, self.one_hot_field_num,
                                             self.multi_hot_field_num, self.seq_maxlen, self._pool, L.ptr(sample_row),
                                             L.ptr(self._ws_for(x)), L.stream()))

    def reduce_sorted(self, x, dout, grad):
        """phase 2 of scatter_grad: grad[g] += contributions, in the order fixed by sort_lookups."""
        B, f_in = x.shape
        seq = self.seq_maxlen if self._pool != 0 else 1
        L.check(L.lib().aread_embed_bwd_reduce(B, f_in, self.embed_dim, seq, L.ptr(dout), L.ptr(grad), L.ptr(self._ws_for(x)),
                                               L.stream()))

    def forward(self, x, squeeze_dim=False, row_plan=None):
        """x: int32 [B, F_in] on the HIP device -> [B, output_dim0, E] (or [B, output_dim0*E])."""
        if row_plan is None:
            out = _EmbedFn.apply(self.embedding_dict.weight, x, self, None, None, 0)
        else:
            out = _EmbedFn.apply(self.embedding_dict.weight, x, self, row_plan.row_sample, row_plan.sample_row,
                                 row_plan.max_rows)
        if squeeze_dim:
            out = torch.flatten(out, start_dim=1)
        return out
