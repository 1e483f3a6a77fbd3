"""Host-side mirror of the reference's model/layer.py for the hot path: same class names,
constructor arguments, attributes, forward() signatures and state_dict keys; the math runs in
libaread_hip.so (hand-written HIP for gfx950) through the C ABI of include/aread_hip.h."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import _lib as L

_POOL = {None: 0, "sum": 1, "mean": 2}
_MAX_LAYER = 4


class _MlpCfg(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("n_layers", C.c_int32), ("dims", C.c_int32 * _MAX_LAYER), ("output_layer", C.c_int32),
                ("precision", C.c_int32), ("dropout", C.c_float)]


class _MlpCall(C.Structure):
    _fields_ = [("B", C.c_int64), ("train", C.c_int32), ("update_running", C.c_int32), ("drop_seed", C.c_uint32),
                ("plan", C.c_void_p), ("params", C.c_void_p), ("stats", C.c_void_p), ("nbt", C.c_void_p), ("ws", C.c_void_p)]


for _n, _r, _a in [
    ("aread_mlp_create", C.c_int, [C.POINTER(_MlpCfg), C.POINTER(C.c_void_p)]),
    ("aread_mlp_workspace_bytes", C.c_int64, [C.c_void_p, C.c_int64]),
    ("aread_mlp_forward", C.c_int, [C.c_void_p, C.POINTER(_MlpCall), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("aread_mlp_backward", C.c_int, [C.c_void_p, C.POINTER(_MlpCall), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
]:
    L.register(_n, _r, _a)


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

    def bwd_ws_bytes(self, x):
        need = L.lib().aread_embed_bwd_ws_bytes(x.shape[0], x.shape[1], self.embed_dim)
        if need < 0:
            raise RuntimeError("aread_embed_bwd_ws_bytes failed")
        return int(need)

    def sort_lookups(self, x, sample_row=None, ws=None):
        """phase 1 of scatter_grad: depends only on the ids, can run on a side stream (ws: a caller-owned workspace of
        bwd_ws_bytes(x) bytes instead of the module's own, e.g. the prefetched batch of AREAD.prepare_batch)."""
        B, f_in = x.shape
        L.check(L.lib().aread_embed_bwd_sort(L.ptr(x), B, f_in, L.ptr(self._offsets_dev(x.device)),
                                             self.embedding_dict.weight.shape[0], self.embed_dim, self.one_hot_field_num,
                                             self.multi_hot_field_num, self.seq_maxlen, self._pool, L.ptr(sample_row),
                                             L.ptr(ws if ws is not None else self._ws_for(x)), L.stream()))

    def reduce_sorted(self, x, dout, grad, ws=None, dout2=None):
        """phase 2 of scatter_grad: grad[g] += contributions, in the order fixed by sort_lookups (dout2: a second addend of
        the gradient, row for row)."""
        B, f_in = x.shape
        seq = self.seq_maxlen if self._pool != 0 else 1
        L.check(L.lib().aread_embed_bwd_reduce2(B, f_in, self.embed_dim, seq, L.ptr(dout), L.ptr(dout2), L.ptr(grad),
                                                L.ptr(ws if ws is not None else self._ws_for(x)), L.stream()))

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



class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dense, mod):
        L.require_device(x, dense)
        x = L.require(x.contiguous(), torch.float32, "x")
        lib = L.lib()
        B = x.shape[0]
        from .plan import RowPlan
        plan = RowPlan(torch.empty((B, 1), dtype=torch.int32, device=x.device), -1, 1)
        ws = torch.empty(lib.aread_mlp_workspace_bytes(mod._handle, B), dtype=torch.uint8, device=x.device)
        out = torch.empty((B, mod.out_features), dtype=torch.float32, device=x.device)
        call = _MlpCall()
        call.B, call.train, call.update_running = B, int(mod.training), int(mod.training)
        mod._calls += 1
        call.drop_seed = (mod.drop_seed if mod.drop_seed is not None else mod._calls * 0x9E3779B1) & 0xFFFFFFFF
        call.plan, call.params, call.stats, call.nbt, call.ws = (L.ptr(plan.buf), L.ptr(dense), L.ptr(mod.bn_stats),
                                                                 L.ptr(mod.bn_nbt), L.ptr(ws))
        L.check(lib.aread_mlp_forward(mod._handle, C.byref(call), L.ptr(x), L.ptr(out), L.stream()))
        ctx.mod, ctx.keep = mod, (plan, ws, call, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        plan, ws, call, x = ctx.keep
        dout = dout.contiguous()
        grads = torch.empty_like(mod.dense)
        dx = torch.empty_like(x)
        L.check(L.lib().aread_mlp_backward(mod._handle, C.byref(call), L.ptr(x), L.ptr(dout), L.ptr(grads), L.ptr(dx), L.stream()))
        return dx, grads, None


class MultiLayerPerceptron(nn.Module):
    """model/layer.py:203-229: [Linear -> BatchNorm1d -> ReLU -> Dropout] x len(layer_dims) (+ Linear(last, 1)).
    Same constructor, forward(x) and state_dict keys ('layers.0.weight', 'layers.1.running_mean', ...); the
    trainable tensors are slices of one flat parameter `dense`.  BatchNorm is skipped for a one-row batch."""

    def __init__(self, input_dim, layer_dims, dropout, output_layer=True, bn=True, precision="f32"):
        super().__init__()
        if not bn:
            raise NotImplementedError("bn=False is not on the accelerated path (AREAD always builds its MLPs with bn=True)")
        layer_dims = [int(d) for d in layer_dims]
        if not 1 <= len(layer_dims) <= _MAX_LAYER:
            raise ValueError(f"between 1 and {_MAX_LAYER} hidden layers are supported")
        cfg = _MlpCfg()
        cfg.in_dim, cfg.n_layers, cfg.output_layer = int(input_dim), len(layer_dims), int(bool(output_layer))
        cfg.precision, cfg.dropout = (1 if precision == "bf16x3" else 0), float(dropout)
        for j, d in enumerate(layer_dims):
            cfg.dims[j] = d
        h = C.c_void_p()
        L.check(L.lib().aread_mlp_create(C.byref(cfg), C.byref(h)))
        self._handle = h
        self.out_features = 1 if output_layer else layer_dims[-1]
        from .aread import TensorDesc
        lib = L.lib()
        self._tensors = []
        d = TensorDesc()
        for i in range(lib.aread_model_n_tensors(h)):
            L.check(lib.aread_model_tensor(h, i, C.byref(d)))
            self._tensors.append((d.name.decode(), int(d.kind), int(d.offset), tuple(int(d.shape[k]) for k in range(d.ndim))))
        self.dense = nn.Parameter(torch.zeros(lib.aread_model_param_floats(h)))
        self.register_buffer("bn_stats", torch.zeros(lib.aread_model_stat_floats(h)))
        self.register_buffer("bn_nbt", torch.zeros(lib.aread_model_n_bn(h), dtype=torch.int64))
        self._calls, self.drop_seed = 0, None
        with torch.no_grad():
            for name, v in self.named_views():
                if name.endswith("running_var"):
                    v.fill_(1.0)
                elif v.dim() == 2:
                    v.uniform_(-1.0 / np.sqrt(v.shape[1]), 1.0 / np.sqrt(v.shape[1]))
                elif int(name.split(".")[1]) % 4 == 1 and name.endswith("weight"):
                    v.fill_(1.0)
                elif name.endswith("bias") and int(name.split(".")[1]) % 4 == 0:
                    v.uniform_(-0.05, 0.05)
        self._register_state_dict_hook(MultiLayerPerceptron._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                L.lib().aread_model_destroy(self._handle)
        except Exception:
            pass

    def named_views(self):
        bufs = {0: self.dense.data, 1: self.bn_stats, 2: self.bn_nbt}
        for name, kind, off, shape in self._tensors:
            n = int(np.prod(shape)) if shape else 1
            yield name, bufs[kind][off:off + n].view(shape)

    @staticmethod
    def _sd_hook(self, sd, prefix, local_metadata):
        bufs = {0: sd.pop(prefix + "dense"), 1: sd.pop(prefix + "bn_stats"), 2: sd.pop(prefix + "bn_nbt")}
        for name, kind, off, shape in self._tensors:
            n = int(np.prod(shape)) if shape else 1
            sd[prefix + name] = bufs[kind][off:off + n].view(shape)
        return sd

    def _load_hook(self, sd, prefix, local_metadata, strict, missing, unexpected, errors):
        if not any(prefix + t[0] in sd for t in self._tensors):
            return
        bufs = {0: self.dense.detach().clone(), 1: self.bn_stats.clone(), 2: self.bn_nbt.clone()}
        for name, kind, off, shape in self._tensors:
            k = prefix + name
            if k in sd:
                v = sd.pop(k)
                n = int(np.prod(shape)) if shape else 1
                bufs[kind][off:off + n] = v.reshape(-1).to(bufs[kind].device, bufs[kind].dtype)
            elif strict:
                missing.append(k)
        sd[prefix + "dense"], sd[prefix + "bn_stats"], sd[prefix + "bn_nbt"] = bufs[0], bufs[1], bufs[2]

    def forward(self, x):
        """x: float32 [B, input_dim] on the HIP device."""
        return _MlpFn.apply(x, self.dense, self)
