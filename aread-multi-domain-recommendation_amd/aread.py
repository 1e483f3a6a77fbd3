"""Host-side mirror of the reference's model/aread.py (class AREAD) for the hot path.

Same constructor, forward() signature/modes, state_dict keys and HEMP-facing attributes; the math
runs in libaread_hip.so.  Differences that are deliberate (and documented in DESIGN.md):
  * the dead attention branch (aread.py:139-140) is never computed; its parameters exist only so
    that a reference checkpoint loads with strict=True;
  * every trainable dense tensor lives in ONE flat nn.Parameter (`dense`); state_dict()/load_state_dict()
    expose it under the reference's per-tensor keys;
  * two extensions: mode='with_mask' is the fused multi-domain call the reference intended
    (aread.py:203-223 raises NameError there), and train_step() is the fused, graph-capturable
    forward + loss + backward used by bench.py.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import os
import torch
from torch import nn

from . import _lib as L
from .hemp import HempMixin
from .layer import FeaturesEmbedding
from .plan import RowPlan

MAX_LEVEL, MAX_LAYER = 4, 4


class ModelCfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("f_out", C.c_int32), ("domain_field", C.c_int32),
                ("n_expert", C.c_int32), ("n_expert_layers", C.c_int32), ("expert_dims", C.c_int32 * MAX_LAYER),
                ("n_level", C.c_int32), ("n_tower", C.c_int32 * MAX_LEVEL), ("n_tower_layers", C.c_int32),
                ("tower_dims", (C.c_int32 * MAX_LAYER) * MAX_LEVEL), ("n_cross", C.c_int32), ("n_domain", C.c_int32),
                ("dropout", C.c_float), ("l2_linear", C.c_float), ("l2_dnn", C.c_float), ("l2_cross", C.c_float),
                ("precision", C.c_int32)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("kind", C.c_int32), ("ndim", C.c_int32), ("offset", C.c_int64),
                ("shape", C.c_int64 * 2), ("l2", C.c_float)]


class Call(C.Structure):
    _fields_ = [("B", C.c_int64), ("n_seg", C.c_int32), ("mode", C.c_int32), ("train", C.c_int32),
                ("update_running", C.c_int32), ("domain", C.c_int32), ("drop_seed", C.c_uint32),
                ("plan", C.c_void_p), ("masks", C.c_void_p), ("params", C.c_void_p), ("stats", C.c_void_p),
                ("nbt", C.c_void_p), ("ws", C.c_void_p), ("probs", C.c_void_p), ("gate_stats", C.c_void_p),
                ("y", C.c_void_p), ("seg_weight", C.c_void_p), ("loss_out", C.c_void_p), ("async_tail", C.c_int32),
                ("l2_table", C.c_void_p), ("l2_n", C.c_int64), ("l2_coef", C.c_float), ("l2_workgroups", C.c_int32),
                ("l2_grad", C.c_void_p), ("l2_partial", C.c_void_p), ("l2_reg_out", C.c_void_p), ("l2_dense_coef", C.c_void_p),
                ("grads_init", C.c_int32), ("init_grads", C.c_void_p), ("init_reg_out", C.c_void_p), ("de_rw", C.c_void_p),
                ("inference", C.c_int32), ("drop_seed_dev", C.c_void_p)]


for _n, _r, _a in [
    ("aread_model_create", C.c_int, [C.POINTER(ModelCfg), C.POINTER(C.c_void_p)]),
    ("aread_model_destroy", None, [C.c_void_p]),
    ("aread_model_param_floats", C.c_int64, [C.c_void_p]),
    ("aread_model_stat_floats", C.c_int64, [C.c_void_p]),
    ("aread_model_n_bn", C.c_int, [C.c_void_p]),
    ("aread_model_n_tensors", C.c_int, [C.c_void_p]),
    ("aread_model_tensor", C.c_int, [C.c_void_p, C.c_int, C.POINTER(TensorDesc)]),
    ("aread_model_edge_count", C.c_int, [C.c_void_p]),
    ("aread_model_gate_rows", C.c_int, [C.c_void_p]),
    ("aread_model_workspace_bytes", C.c_int64, [C.c_void_p, C.c_int64, C.c_int]),
    ("aread_model_l2_coef", C.c_int, [C.c_void_p, C.c_void_p]),
    ("aread_prepare", C.c_int, [C.c_void_p, C.POINTER(Call), C.c_void_p]),
    ("aread_l2_dense_init", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("aread_step_total", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("aread_forward", C.c_int, [C.c_void_p, C.POINTER(Call), C.c_void_p, C.c_void_p]),
    ("aread_backward", C.c_int, [C.c_void_p, C.POINTER(Call), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("aread_join", C.c_int, [C.c_void_p, C.c_void_p]),
    ("aread_debug_ws_offset", C.c_int64, [C.c_void_p, C.c_int64, C.c_int, C.c_char_p]),
    ("aread_l2_dense", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("aread_l2_dense_total", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
]:
    L.register(_n, _r, _a)


def pack_masks(masks, n_domain, edge_count, device):
    """list (per domain) of masks (list of n_level+1 bool tensors/arrays, or None) -> uint8 [n_domain, edges]."""
    out = np.zeros((n_domain, edge_count), dtype=np.uint8)
    for d, mk in enumerate(masks):
        if mk is None:
            continue
        flat = np.concatenate([(m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m))
                               .astype(np.uint8).reshape(-1) for m in mk])
        if flat.size != edge_count:
            raise ValueError(f"mask of domain {d} has {flat.size} edges, expected {edge_count}")
        out[d] = flat
    return torch.from_numpy(out).to(device)


class _CallState:
    """Everything one forward leaves behind for its backward (buffers are owned here)."""
    __slots__ = ("plan", "e", "ws", "call", "keep", "probs", "x")


class PreparedBatch:
    """The ids-only stage of a step (AREAD.prepare_batch): row plan + the embedding backward's index sort of one batch,
    computed on the model's prefetch stream while the previous step runs; `ready` is recorded behind both."""
    __slots__ = ("x", "plan", "sort_ws", "ready", "n_seg")


class _AreadFn(torch.autograd.Function):
    """autograd node of one forward call.  `table` and `anchor` (the first dense parameter) only tie the node into
    the graph; the dense gradients are written by the library into one flat buffer and handed to the per-tensor
    parameters as views (AREAD._accumulate_dense), with grad=None where the reference's autograd leaves None."""

    @staticmethod
    def forward(ctx, table, anchor, model, x, st, present):
        ctx.model, ctx.st, ctx.present = model, st, present
        ctx.table_shape = tuple(table.shape)
        return st.probs

    @staticmethod
    def backward(ctx, dprobs):
        model, st = ctx.model, ctx.st
        dprobs = dprobs.contiguous()
        # The flat dense gradient lives in the model's gradient arena (every parameter's .grad is a cached view of it):
        #   no gradient held yet (zero_grad -> backward)      : the library writes the arena
        #   gradients held in the arena (the regulariser's node ran first, or gradient accumulation over several backwards):
        #                                                       the library ADDS onto the arena (aread_call.grads_init)
        #   gradients held elsewhere (a buffer adopted from train_step): a temporary, added by torch
        fresh = model._grads_fresh()
        arena = model._grad_arena(st.e.device)
        in_place = fresh or model._gflat is arena
        grads = arena if in_place else torch.empty_like(model.dense)
        de = model._de_arena(st.e)
        st.call.grads_init = 0 if fresh or not in_place else 1
        try:
            L.check(L.lib().aread_backward(model._handle, C.byref(st.call), L.ptr(st.e), L.ptr(dprobs), L.ptr(grads),
                                           L.ptr(de), L.stream()))
        finally:
            st.call.grads_init = 0
        model._accumulate_dense(grads, ctx.present, take=True, in_arena=in_place)
        # The table gradient.  When the regulariser's node of the SAME backward pass has already produced its dense 2*l2*W (it runs
        # first: it was added to the loss last), the looked-up rows are scatter-added into THAT buffer and this node contributes
        # nothing of its own: no zero-filled 178 MB buffer and no 178 MB addition by the autograd engine per step.
        pend = model.__dict__.get("_gtab_pending")
        if pend is not None and tuple(pend.shape) == ctx.table_shape and pend.device == de.device:
            model.__dict__["_gtab_pending"] = None          # (the engine may then take the buffer as .grad without a copy)
            model.embedding.scatter_grad(st.x, de, pend, st.plan.sample_row)
            return None, None, None, None, None, None
        gtab = torch.zeros(ctx.table_shape, dtype=torch.float32, device=de.device)
        model.embedding.scatter_grad(st.x, de, gtab, st.plan.sample_row)
        return gtab, None, None, None, None, None


class _RegFn(torch.autograd.Function):
    """get_regularization_loss (layer.py:96-112) as one streaming pass per direction."""

    @staticmethod
    def forward(ctx, table, anchor, model):
        ctx.model = model
        ctx.save_for_backward(table)
        out = torch.zeros(257, dtype=torch.float32, device=table.device)
        lib = L.lib()
        part = model._l2_partials(table.device)
        L.check(lib.aread_l2_table(L.ptr(table), table.numel(), model.l2_reg_embedding, 1.0, None, L.ptr(part), L.stream()))
        L.check(lib.aread_l2_finish(L.ptr(part), part.numel(), model.l2_reg_embedding, L.ptr(out), 0, L.stream()))
        L.check(lib.aread_l2_dense(L.ptr(model.dense), L.ptr(model._l2_coef(table.device)), model.dense.numel(), None,
                                   L.ptr(out), 1, L.stream()))
        return out[:1].clone()

    @staticmethod
    def backward(ctx, gout):
        (table,) = ctx.saved_tensors
        model = ctx.model
        g = gout.reshape(-1)[:1].contiguous()      # scalar scale of the regulariser in the caller's loss: stays on the device
        gtab = torch.empty_like(table)
        L.check(L.lib().aread_l2_table_dev(L.ptr(table), table.numel(), model.l2_reg_embedding, L.ptr(g), L.ptr(gtab), L.stream()))
        if model._grads_fresh():               # (this node runs before the forward's: the regulariser was added to the loss last)
            arena = model._grad_arena(table.device)
            torch.mul(model._l2_coef2(table.device), model.dense, out=arena)
            arena.mul_(g)
            model._accumulate_dense(arena, model._reg_present, take=True, in_arena=True)
        else:
            model._accumulate_dense(model._l2_coef2(table.device) * model.dense * g, model._reg_present, take=True)
        # offered to the forward's node of this backward pass (see _AreadFn.backward); withdrawn when the pass ends
        model.__dict__["_gtab_pending"] = gtab
        torch.autograd.Variable._execution_engine.queue_callback(model._drop_pending_table_grad)
        return gtab, None, None


class _MaskList(list):
    """model.domain_mask: a plain list for every reader (aread.py:61), but every assignment to or through it bumps
    owner.mask_version, so caches derived from the masks (which dense tensors receive a gradient) cannot go stale when HEMP
    rewrites an entry in place (aread.py:330-341)."""

    def __init__(self, it, owner):
        super().__init__(it)
        self._owner = owner

    def _touch(self):
        self._owner.__dict__["mask_version"] = self._owner.__dict__.get("mask_version", 0) + 1

    def __setitem__(self, k, v):
        super().__setitem__(k, v); self._touch()

    def __delitem__(self, k):
        super().__delitem__(k); self._touch()

    def append(self, v):
        super().append(v); self._touch()

    def extend(self, it):
        super().extend(it); self._touch()

    def insert(self, i, v):
        super().insert(i, v); self._touch()

    def __reduce__(self):                     # checkpoints store a plain list (run.py:459-484 keeps 'domain_mask')
        return (list, (list(self),))


class AREAD(HempMixin, nn.Module):
    """Adaptive REcommendation for All Domains -- model/aread.py:15-322 on MI355X."""

    def __init__(self, one_hot_feature_dims, embed_dim, multi_hot_dict, n_tower, n_domain, base_model,
                 expert_dims, tower_dims, domain_idx, domain2group=None, n_cross_layers=3, dropout=0.2, device=None,
                 l2_reg_embedding=1e-5, l2_reg_linear=1e-5, l2_reg_dnn=1e-5, l2_reg_cross=1e-5, config=None):
        super().__init__()
        if base_model != "mmoe":
            raise NotImplementedError("only base_model='mmoe' is on the accelerated path (SURVEY 2.1 #11)")
        if not getattr(config, "use_dcn", False):
            raise ValueError("AREAD's masked path needs use_dcn=True (aread.py:231,307 use cn_out unconditionally)")
        self.model_name = "aread"
        self.base_model = base_model
        self.embedding = FeaturesEmbedding(one_hot_feature_dims, embed_dim, multi_hot_dict)
        self.embed_dim = embed_dim
        self.embed_output_dim = self.embedding.output_dim0 * embed_dim
        self.field_num = self.embedding.one_hot_field_num + self.embedding.multi_hot_field_num
        self.domain_idx = domain_idx
        self.n_tower = tuple(int(t) for t in n_tower)
        self.n_level = len(self.n_tower)
        self.edge_num = self.n_tower[0] + sum(self.n_tower[l - 1] * self.n_tower[l] for l in range(1, self.n_level)) \
            + self.n_tower[-1]
        self.n_domain = n_domain
        self.tower_dims = tuple(tuple(int(v) for v in t) for t in tower_dims)
        self.expert_dims = tuple(int(v) for v in expert_dims)
        self.bottom_level = len(self.expert_dims)
        self.device = device
        self.domain2group = np.array([domain2group[d] for d in range(n_domain)]) if domain2group is not None else None
        self.domain_mask = [None for _ in range(n_domain)]
        self.candidate_domain_mask = None
        self.tower2cluster = [[None for _ in range(self.n_tower[l])] for l in range(self.n_level)]
        self.model_state = None
        self.domain_tower_gate_values = None
        self.tmp_tower_gate_values = [[None for _ in range(self.n_tower[l])] for l in range(self.n_level)]
        self.gate_value_threshold = None
        self.eval_loss = None
        self.domain_size = np.array(config.domain_size[config.dataset_name])
        self.use_dcn, self.use_atten = True, getattr(config, "use_atten", False)
        self.dropout = float(dropout)
        self.l2_reg_embedding, self.l2_reg_linear = float(l2_reg_embedding), float(l2_reg_linear)
        self.l2_reg_dnn, self.l2_reg_cross = float(l2_reg_dnn), float(l2_reg_cross)
        if domain_idx >= self.embedding.one_hot_field_num:
            raise ValueError("domain_idx must name a one-hot column")

        cfg = ModelCfg()
        cfg.embed_dim, cfg.f_out, cfg.domain_field = embed_dim, self.embedding.output_dim0, domain_idx
        cfg.n_expert, cfg.n_expert_layers = int(config.mmoe_n_expert), len(self.expert_dims)
        for j, v in enumerate(self.expert_dims):
            cfg.expert_dims[j] = v
        cfg.n_level, cfg.n_tower_layers = self.n_level, len(self.tower_dims[0])
        for l in range(self.n_level):
            cfg.n_tower[l] = self.n_tower[l]
            if len(self.tower_dims[l]) != cfg.n_tower_layers:
                raise ValueError("every tower level must have the same number of layers")
            for j, v in enumerate(self.tower_dims[l]):
                cfg.tower_dims[l][j] = v
        cfg.n_cross, cfg.n_domain, cfg.dropout = int(config.n_cross_layers), n_domain, self.dropout
        cfg.l2_linear, cfg.l2_dnn, cfg.l2_cross = self.l2_reg_linear, self.l2_reg_dnn, self.l2_reg_cross
        # extension knob (not in the reference's config.py): 'f32' = exact fp32 MFMA everywhere (default),
        # 'bf16x3' = split-bf16 forward/dgrad GEMMs in the expert and tower layers
        self.precision = getattr(config, "aread_precision", "f32")
        if self.precision not in ("f32", "bf16x3"):
            raise ValueError("config.aread_precision must be 'f32' or 'bf16x3'")
        cfg.precision = 1 if self.precision == "bf16x3" else 0
        self._cfg = cfg
        h = C.c_void_p()
        L.check(L.lib().aread_model_create(C.byref(cfg), C.byref(h)))
        self._handle = h
        lib = L.lib()
        self.n_heads = self.n_tower[-1]
        self._edge_count = lib.aread_model_edge_count(h)
        self._gate_rows = lib.aread_model_gate_rows(h)
        assert self._edge_count == self.edge_num
        self._tensors = []
        d = TensorDesc()
        for i in range(lib.aread_model_n_tensors(h)):
            L.check(lib.aread_model_tensor(h, i, C.byref(d)))
            shape = tuple(int(d.shape[k]) for k in range(d.ndim))
            self._tensors.append((d.name.decode(), int(d.kind), int(d.offset), shape, float(d.l2)))
        # storage: ONE flat fp32 buffer read by the kernels; every reference tensor is a trainable nn.Parameter whose
        # storage is a slice of it (so optimizers see the reference's per-tensor parameters and grad=None semantics)
        self.register_buffer("dense", torch.zeros(lib.aread_model_param_floats(h)))
        self._ptensors = [t for t in self._tensors if t[1] == 0]
        self.dense_params = nn.ParameterList([nn.Parameter(self._view_of(self.dense, t)) for t in self._ptensors])
        self.__dict__["_dparams"] = list(self.dense_params)      # plain list: iterating an nn.ParameterList costs ~2 us per item
        self._gflat, self._garena, self._gviews, self._dearena, self._ext_views = None, None, None, None, None
        self._mask_cache, self._presence_wo = {}, None
        self._reg_present = [t[4] > 0 for t in self._ptensors]
        self.register_buffer("bn_stats", torch.zeros(lib.aread_model_stat_floats(h)))
        self.register_buffer("bn_nbt", torch.zeros(lib.aread_model_n_bn(h), dtype=torch.int64))
        # parameters that never influence the output (kept for checkpoint compatibility only)
        self.group_embedding_dead = None
        self.final_gate = nn.Sequential(nn.Linear(2 * embed_dim, self.n_tower[-1], bias=False), nn.Softmax(dim=1))
        if self.use_atten:
            a = getattr(config, "atten_embed_dim", embed_dim)
            self.atten_embedding = nn.Linear(embed_dim, a)
            self.self_attns = nn.ModuleList([nn.MultiheadAttention(a, config.att_head_num, dropout=dropout)
                                             for _ in range(config.att_layer_num)])
            if config.att_res:
                self.V_res_embedding = nn.Linear(embed_dim, a)
            self.atten_linear = nn.Linear(self.embedding.output_dim0 * a, 1, bias=False)
        self._init_dense()
        self._coef, self._part, self._ws_cache, self._streams = {}, {}, {}, {}
        self._drop_calls = 0
        self._pending_dense_l2 = False
        self.drop_seed_base = 0
        import os
        self.l2_pass_workgroups = int(os.environ.get("AREAD_L2_WG", "0"))   # width of the table L2 sweep inside the fused step
        self.l2_pass_early = os.environ.get("AREAD_L2_EARLY", "0") == "1"    # A/B: the sweep right after the row plan (round 1)
        self.l2_pass_in_backward = os.environ.get("AREAD_L2_IN_BWD", "1") == "1"   # default: issued by aread_backward beside the tower backward
        self.sort_early = os.environ.get("AREAD_SORT_EARLY", "0") == "1"                  # A/B: index sort issued before the forward (graph replay: +10 us)
        # aread_prepare ahead of the row plan, and with it the dense L2 terms at the head of train_step (gradient buffer initialised with
        # them): both were wins while the step's tail waited for the side stream; with the tail balanced they cost the head more than
        # they save (0.7353 -> 0.7226 ms/step with both off, profiles/r03_ab_variants.txt section 16).  Off by default: the preparation
        # forks inside aread_forward, the dense L2 terms are formed with the step's total in the last launch.
        self.prepare_early = os.environ.get("AREAD_PREPARE_EARLY", "0") == "1"
        self.l2_dense_first = os.environ.get("AREAD_L2_DENSE_FIRST", "1") == "1"          # (only with AREAD_PREPARE_EARLY=1)
        self._dense_l2_done_first = False
        self.split_de = os.environ.get("AREAD_SPLIT_DE", "1") == "1"                      # row-wise share of dL/de in its own buffer (A/B: 0 = accumulated into de)
        self._train_step_owner = False
        self._de_split = False
        self.drop_seed = None          # set to an int to pin the dropout stream (tests)
        self.drop_seed_dev = None      # device_dropout_seed(): the seed lives in device memory (graph replays draw new masks)
        self._marks = None             # tools/step_anatomy.py: list collecting (name, event) at the step's host-level boundaries
        self._register_state_dict_hook(AREAD._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    @property
    def domain_mask(self):
        return self.__dict__["_domain_mask"]

    @domain_mask.setter
    def domain_mask(self, value):
        self.__dict__["_domain_mask"] = _MaskList(value, self)
        self.__dict__["mask_version"] = self.__dict__.get("mask_version", 0) + 1

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                L.lib().aread_model_destroy(self._handle)
        except Exception:
            pass

    # ---- parameters -----------------------------------------------------------------------------
    @staticmethod
    def _view_of(flat, t):
        n = int(np.prod(t[3])) if t[3] else 1
        return flat[t[2]:t[2] + n].view(t[3])

    def _apply(self, fn, *args, **kwargs):
        """Module.to()/cuda()/cpu(): move the flat buffer, then re-point every parameter at its slice."""
        super()._apply(fn, *args, **kwargs)
        for p, t in zip(self.dense_params, self._ptensors):
            p.data = self._view_of(self.dense, t)
            p.grad = None
        self._gflat, self._garena, self._gviews, self._dearena, self._ext_views = None, None, None, None, None
        self.__dict__["_ghave"] = 0
        self._mask_cache = {}
        return self

    def named_dense_parameters(self):
        """(reference state_dict key, nn.Parameter) for every dense tensor."""
        return [(t[0], p) for t, p in zip(self._ptensors, self.dense_params)]

    def zero_grad(self, set_to_none: bool = True):
        """nn.Module.zero_grad over cached parameter lists (the generic walk over 300 parameters costs 0.4 ms per step)"""
        if "_all_params" not in self.__dict__:
            self.__dict__["_all_params"] = list(self.parameters())
        for p in self._all_params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:                                   # (the gradients are views of one flat buffer: no detach_() on a view)
                    g = p.grad
                    if g.grad_fn is not None:
                        p.grad = g = g.detach()
                    g.zero_()
        # what the module knows about the dense parameters' .grad without reading 300 attributes: a bit per parameter that holds a
        # gradient view THIS module assigned since the last all-None state; None = unknown (grads zeroed in place, external edits)
        self.__dict__["_ghave"] = 0 if set_to_none else None

    _GRAD_SCAN = os.environ.get("AREAD_GRAD_SCAN", "0") == "1"     # 1: always read every parameter's .grad (no bookkeeping shortcuts)

    @staticmethod
    def _bits(present):
        b = 0
        for i, pres in enumerate(present):
            if pres:
                b |= 1 << i
        return b

    def _present_bits(self, present):
        """bit mask + index list of a presence list, cached by the list object (the lists themselves are cached per mask version)"""
        c = self.__dict__.setdefault("_pbits", {})
        ent = c.get(id(present))
        if ent is None or ent[0] is not present:
            if len(c) > 256:
                c.clear()
            ent = (present, self._bits(present), [i for i, pres in enumerate(present) if pres])
            c[id(present)] = ent
        return ent[1], ent[2]

    def _grads_fresh(self):
        """no dense parameter holds a gradient (zero_grad(set_to_none=True) or a new model).
        Fast path: the module's own record (`_ghave`), confirmed on one parameter -- model.zero_grad() and the module's own
        assignments keep it exact; an optimizer's zero_grad() is noticed on the sampled parameter and answered by a full scan."""
        have = self.__dict__.get("_ghave")
        dp = self._dparams
        if have is not None and not self._GRAD_SCAN:
            if have == 0:
                if dp[0].grad is None and dp[-1].grad is None:
                    return True
            else:
                k = (have & -have).bit_length() - 1           # a parameter this module gave a gradient to
                if dp[k].grad is not None:
                    return False
        for p in dp:                                            # unknown state (or the sample disagreed): look at everything
            if p.grad is not None:
                self.__dict__["_ghave"] = None
                return False
        self.__dict__["_ghave"] = 0
        return True

    def _drop_pending_table_grad(self):
        self.__dict__["_gtab_pending"] = None

    def _grad_presence(self):
        """bool array: which dense parameters hold a gradient right now (what an optimizer's `p.grad is None` test sees)"""
        have = self.__dict__.get("_ghave")
        dp = self._dparams
        if have is not None and have != 0 and not self._GRAD_SCAN:
            k = (have & -have).bit_length() - 1
            if dp[k].grad is not None:                          # the record is live (nobody set the gradients to None behind it)
                c = self.__dict__.setdefault("_hbits", {})
                arr = c.get(have)
                if arr is None:
                    if len(c) > 256:
                        c.clear()
                    arr = c[have] = np.array([(have >> i) & 1 for i in range(len(dp))], dtype=bool)
                return arr
        return np.array([p.grad is not None for p in dp], dtype=bool)

    def _grad_arena(self, device):
        """one flat gradient buffer per model, reused by every backward; `_gviews[i]` is parameter i's view of it"""
        if self._garena is None or self._garena.device != torch.device(device):
            self._garena = torch.zeros_like(self.dense)
            self._gviews = [self._view_of(self._garena, t) for t in self._ptensors]
        return self._garena

    def _de_arena(self, e):
        if self._dearena is None or self._dearena.shape != e.shape or self._dearena.device != e.device:
            self._dearena = torch.empty_like(e)
        return self._dearena

    def _accumulate_dense(self, flat, present, take=False, in_arena=False):
        """Add a flat gradient contribution.  Tensors marked present get `.grad` (a view of the flat gradient buffer);
        the others keep grad=None exactly as the reference's autograd leaves them (Adam then skips them).
        in_arena: `flat` IS the model's gradient arena and no gradient was held before (the caller checked)."""
        dp = self._dparams
        if in_arena:
            # `flat` is the arena and already holds the sum (written by a fresh backward, or added onto in place)
            self._gflat, views = flat, self._gviews
            bits, idx = self._present_bits(present)
            have = self.__dict__.get("_ghave")
            if have is not None and not self._GRAD_SCAN:
                need = bits & ~have
                if need:
                    for i in idx:
                        if (need >> i) & 1:
                            dp[i].grad = views[i]
                self.__dict__["_ghave"] = have | bits
            else:
                for i in idx:
                    if dp[i].grad is None:
                        dp[i].grad = views[i]
            return
        fresh = self._grads_fresh()
        if fresh and take:                      # adopt the caller's buffer (train_step's persistent bufs['gdense'], a fresh temporary)
            self._gflat = flat
        elif fresh:
            arena = self._grad_arena(flat.device)
            arena.copy_(flat)
            self._gflat = arena
        else:
            self._gflat.add_(flat)
        if self._gflat is self._garena:
            views = self._gviews
        else:                                   # views of an adopted buffer, cached by its address (one entry: the step buffers persist)
            key = (self._gflat.data_ptr(), self._gflat.numel())
            if self._ext_views is None or self._ext_views[0] != key:
                self._ext_views = (key, [self._view_of(self._gflat, t) for t in self._ptensors])
            views = self._ext_views[1]
        bits, idx = self._present_bits(present)
        have = self.__dict__.get("_ghave")
        if have is not None and not self._GRAD_SCAN:
            # the module knows which parameters already hold a view of this buffer: only the new ones are touched
            need = bits & ~have
            if need:
                for i in idx:
                    if (need >> i) & 1:
                        dp[i].grad = views[i]
            self.__dict__["_ghave"] = have | bits
            return
        for i in idx:
            if dp[i].grad is None:
                dp[i].grad = views[i]

    def _presence(self, mode_id, masks):
        """Which dense tensors are on a gradient path of a forward call (what the reference's autograd would reach)."""
        if mode_id == 1:
            act = [[True] * n for n in self.n_tower]
        else:
            act = [[False] * n for n in self.n_tower]
            for mk in masks:
                if mk is None:                       # no mask installed for this domain: treat every tower as reachable
                    act = [[True] * n for n in self.n_tower]
                    break
                for l in range(self.n_level):
                    a = np.asarray(mk[l].cpu() if isinstance(mk[l], torch.Tensor) else mk[l]).any(axis=0)
                    for t in range(self.n_tower[l]):
                        act[l][t] = act[l][t] or bool(a[t])
        out = []
        for name, *_ in self._ptensors:
            k = name.split(".")
            if k[0] == "mmoe_gates":
                out.append(act[0][int(k[1])])
            elif k[0] == "group_embedding":
                out.append(mode_id == 0)
            elif k[0] == "towers":
                out.append(act[int(k[1])][int(k[2])])
            elif k[0] == "tower_gates":
                out.append(act[int(k[1]) + 1][int(k[2])])
            elif k[0] == "towers_linear":
                out.append(act[-1][int(k[1])])
            else:
                out.append(True)             # linear, cn, mmoe_experts
        return out

    def named_views(self):
        """(reference key, tensor view) for every tensor that lives in dense / bn_stats / bn_nbt."""
        bufs = {0: self.dense, 1: self.bn_stats, 2: self.bn_nbt}
        for name, kind, off, shape, _ in self._tensors:
            n = int(np.prod(shape)) if shape else 1
            yield name, bufs[kind][off:off + n].view(shape)

    def _init_dense(self):
        """torch defaults of the reference's layers: Linear kaiming-uniform(a=sqrt(5)), BN gamma=1/beta=0,
        running stats 0/1, cn.b = 0 (layer.py:525-527), Embedding N(0,1)."""
        with torch.no_grad():
            for name, v in self.named_views():
                if name.endswith("running_var"):
                    v.fill_(1.0)
                elif name.endswith(("running_mean", "num_batches_tracked")) or name.startswith("cn.b."):
                    v.zero_()
                elif name == "group_embedding.weight":
                    v.normal_()
                elif ".layers." in name and v.dim() == 1 and name.endswith("weight"):
                    v.fill_(1.0)                       # BatchNorm gamma
                elif ".layers." in name and v.dim() == 1 and int(name.split(".layers.")[1].split(".")[0]) % 4 == 1:
                    v.zero_()                          # BatchNorm beta
                elif v.dim() == 2:
                    bound = 1.0 / np.sqrt(v.shape[1])
                    v.uniform_(-bound, bound)
                else:                                  # Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                    v.uniform_(-0.05, 0.05)

    @staticmethod
    def _sd_hook(self, sd, prefix, local_metadata):
        out = OrderedDict()
        for k, v in sd.items():
            if k in (prefix + "dense", prefix + "bn_stats", prefix + "bn_nbt") or k.startswith(prefix + "dense_params."):
                continue
            out[k] = v
        bufs = {0: sd[prefix + "dense"], 1: sd[prefix + "bn_stats"], 2: sd[prefix + "bn_nbt"]}
        for name, kind, off, shape, _ in self._tensors:
            n = int(np.prod(shape)) if shape else 1
            out[prefix + name] = bufs[kind][off:off + n].view(shape)
        sd.clear()
        sd.update(out)
        return sd

    def _load_hook(self, sd, prefix, local_metadata, strict, missing, unexpected, errors):
        names = {prefix + t[0] for t in self._tensors}
        if not any(k in names for k in sd):
            return
        bufs = {0: self.dense.detach().clone(), 1: self.bn_stats.clone(), 2: self.bn_nbt.clone()}
        for name, kind, off, shape, _ in self._tensors:
            k = prefix + name
            if k in sd:
                v = sd.pop(k)
                n = int(np.prod(shape)) if shape else 1
                if tuple(v.shape) != shape:
                    errors.append(f"size mismatch for {k}: {tuple(v.shape)} vs {shape}")
                    continue
                bufs[kind][off:off + n] = v.reshape(-1).to(bufs[kind].device, bufs[kind].dtype)
            elif strict:
                missing.append(k)
        sd[prefix + "dense"], sd[prefix + "bn_stats"], sd[prefix + "bn_nbt"] = bufs[0], bufs[1], bufs[2]
        for i, t in enumerate(self._ptensors):
            sd[prefix + f"dense_params.{i}"] = self._view_of(bufs[0], t)

    def _l2_coef(self, device):
        key = str(device)
        if key not in self._coef:
            host = torch.empty(self.dense.numel(), dtype=torch.float32)
            L.check(L.lib().aread_model_l2_coef(self._handle, host.data_ptr()))
            self._coef[key] = host.to(device)
        return self._coef[key]

    def _l2_coef2(self, device):
        key = "2:" + str(device)
        if key not in self._coef:
            self._coef[key] = 2.0 * self._l2_coef(device)
        return self._coef[key]

    def _l2_partials(self, device):
        key = str(device)
        if key not in self._part:
            self._part[key] = torch.empty(L.lib().aread_l2_partials(), dtype=torch.float32, device=device)
        return self._part[key]

    def get_regularization_loss(self, device=None):
        return _RegFn.apply(self.embedding.embedding_dict.weight, self.dense_params[0], self)

    # ---- one call ---------------------------------------------------------------------------------
    def _masks_dev(self, masks, device):
        return pack_masks(masks, self.n_domain, self._edge_count, device)

    def _run(self, x, mode_id, n_seg, domain, masks_dev, want_gates, y=None, seg_weight=None, loss_out=None,
             ws=None, plan=None, probs=None, e=None, e_ready=False, async_fwd=False, pre=None):
        """plan + embedding + dense forward.  Returns a _CallState (buffers owned by it).
        e_ready: `e` already holds the pooled embedding in plan order (row-sharded table, dist.ShardedTableStep).
        pre: the tuple _make_call returned (the fused step builds the call first and hands it to aread_prepare)."""
        L.require_device(x, self.dense, self.embedding.embedding_dict.weight)
        L.require(x, torch.int32, "x")
        lib = L.lib()
        B = x.shape[0]
        st = _CallState()
        st.x = x
        st.plan = plan if plan is not None else RowPlan(x, self.domain_idx if n_seg > 1 else -1, n_seg)
        table = self.embedding.embedding_dict.weight
        emb = self.embedding
        st.e = e if e is not None else torch.empty((st.plan.max_rows, self.embed_output_dim), dtype=torch.float32,
                                                   device=x.device)
        if not e_ready:
            L.check(lib.aread_embed_fwd(L.ptr(x), B, x.shape[1], L.ptr(emb._offsets_dev(x.device)), L.ptr(table),
                                        table.shape[0], emb.embed_dim, emb.one_hot_field_num, emb.multi_hot_field_num,
                                        emb.seq_maxlen, emb._pool, L.ptr(st.plan.row_sample), st.plan.max_rows,
                                        L.ptr(st.e), None, L.stream()))
        if pre is None:
            pre = self._make_call(x.device, B, mode_id, n_seg, domain, masks_dev, want_gates, y, seg_weight, loss_out, ws, probs,
                                  async_fwd)
        call, st.ws, st.probs, gate, st.keep = pre
        call.plan = L.ptr(st.plan.buf)
        st.call = call
        L.check(lib.aread_forward(self._handle, C.byref(call), L.ptr(st.e), L.stream()))
        return st, gate

    def _make_call(self, device, B, mode_id, n_seg, domain, masks_dev, want_gates, y=None, seg_weight=None, loss_out=None,
                   ws=None, probs=None, async_fwd=False):
        """The aread_call of one forward (everything but the row plan) and the buffers it points to."""
        lib = L.lib()
        if ws is None:
            nbytes = lib.aread_model_workspace_bytes(self._handle, B, n_seg)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        probs = probs if probs is not None else torch.empty((self.n_heads, B), dtype=torch.float32, device=device)
        gate = torch.zeros((n_seg, max(self._gate_rows, 1)), dtype=torch.float32, device=device) if want_gates else None
        train = self.training
        if train and self.dropout > 0:
            self._drop_calls += 1
        call = Call()
        call.B, call.n_seg, call.mode, call.train = B, n_seg, mode_id, int(train)
        call.update_running, call.domain = int(train), int(domain if domain is not None else 0)
        call.drop_seed = ((self.drop_seed_base + self._drop_calls * 0x9E3779B1) if self.drop_seed is None
                          else int(self.drop_seed)) & 0xFFFFFFFF
        call.drop_seed_dev = L.ptr(self.drop_seed_dev) if self.drop_seed_dev is not None else None
        call.masks = L.ptr(masks_dev)
        call.params, call.stats, call.nbt = L.ptr(self.dense), L.ptr(self.bn_stats), L.ptr(self.bn_nbt)
        call.ws, call.probs, call.gate_stats = L.ptr(ws), L.ptr(probs), L.ptr(gate)
        call.y, call.seg_weight, call.loss_out = L.ptr(y), L.ptr(seg_weight), L.ptr(loss_out)
        call.async_tail = 2 if async_fwd else 0          # fused step: loss / running stats finish on the library's side stream
        call.inference = int(not train and not torch.is_grad_enabled())   # eval under no_grad: BatchNorm + ReLU inside the expert GEMMs
        return call, ws, probs, gate, (masks_dev, gate, y, seg_weight, loss_out)

    def _mask_info(self, d, mask, device):
        """Host-side facts about one domain's mask, cached: packed device copy, active heads, gradient-presence list.
        The reference decides these with host-side `if`s on device booleans in EVERY forward (aread.py:272,309,320: 33 device
        reads per call); here one device read per (mask object, in-place version).  Key: the mask list's identity and every
        tensor's autograd version counter (bumped by any in-place write), or the bytes of a numpy mask."""
        key = (d, id(mask)) + tuple((id(t), t._version) if isinstance(t, torch.Tensor) else hash(np.asarray(t).tobytes()) for t in mask)
        ent = self._mask_cache.get(key)
        if ent is None or ent["mask"] is not mask:
            host = [np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t).astype(bool) for t in mask]
            if not host[0].reshape(-1).any():
                raise ValueError("mask[0] has no active level-0 tower")
            active = np.nonzero(host[self.n_level - 1].any(axis=0))[0]
            if active.size == 0:
                raise RuntimeError("mask has no active last-level tower (aread.py:312-322)")
            masks = [None] * self.n_domain
            masks[d] = host
            if len(self._mask_cache) > 512:
                self._mask_cache.clear()
            ent = {"mask": mask, "dev": self._masks_dev(masks, device), "present": self._presence(0, [host]),
                   "active": torch.from_numpy(active).to(device)}
            self._mask_cache[key] = ent
        return ent

    def _record_gates(self, gate_row, d, memory_gate_value, tmp_memory_gate_value):
        """Side outputs of aread.py:187-200,275-295 from the kernel's [gate_rows] vector."""
        off = 0
        for l in range(1, self.n_level):
            for t in range(self.n_tower[l]):
                v = gate_row[off:off + self.n_tower[l - 1]].clone()
                if self.device is not None:
                    v = v.to(self.device)          # the HEMP host logic runs where the masks live
                off += self.n_tower[l - 1]
                if tmp_memory_gate_value:
                    self.tmp_tower_gate_values[l][t] = v
                if memory_gate_value:
                    self.domain_tower_gate_values[d][l][t].append(v)

    def forward(self, x, mode="wo_mask", targets=None, memory_gate_value=False, domain_i=None, current_mask=None,
                tmp_memory_gate_value=False):
        """Same contract as aread.py:129-261 for 'wo_mask' ([B,1]), 'domain_with_mask' ([B]) and
        'domain_mask_bagging' ([K_active,B]).  'with_mask' (extension) = every sample uses the mask of its
        own domain; returns (mean over active heads, targets), both in domain order."""
        want_gates = bool(memory_gate_value or tmp_memory_gate_value) and self._gate_rows > 0
        table = self.embedding.embedding_dict.weight
        if mode == "wo_mask":
            st, gate = self._run(x, 1, 1, domain_i, None, want_gates and domain_i is not None)
            if self._presence_wo is None:
                self._presence_wo = self._presence(1, None)
            probs = (_AreadFn.apply(table, self.dense_params[0], self, x, st, self._presence_wo)
                     if torch.is_grad_enabled() else st.probs)
            if want_gates and domain_i is not None:
                self._record_gates(gate[0], domain_i, memory_gate_value, False)
            return probs.mean(dim=0).unsqueeze(-1)
        if mode in ("domain_with_mask", "domain_mask_bagging"):
            mask = self.domain_mask[domain_i] if current_mask is None else current_mask
            if mask is None:
                raise ValueError("no mask for this domain (domain_mask is filled by the first regroup)")
            d = 0 if domain_i is None else int(domain_i)
            info = self._mask_info(d, mask, x.device)
            st, gate = self._run(x, 0, 1, d, info["dev"], want_gates)
            probs = (_AreadFn.apply(table, self.dense_params[0], self, x, st, info["present"])
                     if torch.is_grad_enabled() else st.probs)
            if want_gates:
                self._record_gates(gate[0], d, memory_gate_value, tmp_memory_gate_value)
            y_stack = probs[info["active"]]
            return y_stack if mode == "domain_mask_bagging" else y_stack.mean(dim=0)
        if mode == "with_mask":
            if any(m is None for m in self.domain_mask):
                raise ValueError("with_mask needs a mask for every domain")
            st, _ = self._run(x, 0, self.n_domain, None, self._masks_dev(self.domain_mask, x.device), False)
            probs = (_AreadFn.apply(table, self.dense_params[0], self, x, st, self._presence(0, self.domain_mask))
                     if torch.is_grad_enabled() else st.probs)
            kact = torch.tensor([float(np.asarray(m[self.n_level - 1].cpu() if isinstance(m[0], torch.Tensor)
                                                  else m[self.n_level - 1]).any(axis=0).sum())
                                 for m in self.domain_mask], device=x.device)
            dom = x[:, self.domain_idx].long()
            y = probs.sum(dim=0) / kact[dom]
            order = torch.argsort(dom, stable=True)
            return y[order], (targets[order] if targets is not None else None)
        raise NotImplementedError(f"mode {mode!r} is not on the accelerated path (SURVEY 8a: dead in the reference)")

    def debug_ws(self, st, name, cols):
        """Test helper: view of a named workspace buffer of a finished call as [max_rows, cols] floats."""
        off = L.lib().aread_debug_ws_offset(self._handle, st.call.B, st.call.n_seg, name.encode())
        if off < 0:
            raise KeyError(name)
        f = st.ws.view(torch.float32)
        return f[off:off + st.plan.max_rows * cols].view(st.plan.max_rows, cols)

    def reset_for_mask_update(self, d=None):
        """aread.py:383-401: (re)allocate the gate-value / candidate-mask / eval-loss bookkeeping lists."""
        fresh = lambda: [[[] for _ in range(self.n_tower[l])] for l in range(self.n_level)] + \
            [[[] for _ in range(self.n_tower[-1])]]
        if d is None:
            self.domain_tower_gate_values = [fresh() for _ in range(self.n_domain)]
            self.gate_value_threshold = [None for _ in range(self.n_domain)]
            self.candidate_domain_mask = [[] for _ in range(self.n_domain)]
            self.eval_loss = [[] for _ in range(self.n_domain)]
        else:
            self.domain_tower_gate_values[d] = fresh()
            self.gate_value_threshold[d] = None
            self.candidate_domain_mask[d] = []
            self.eval_loss[d] = []

    # ---- fused training step (extension; what bench.py times) -------------------------------------------
    def make_step_buffers(self, B, multi_domain=True, device=None, with_table_grad=True):
        device = device or self.dense.device
        n_seg = self.n_domain if multi_domain else 1
        lib = L.lib()
        lay = L.PlanLayout()
        L.check(lib.aread_plan_layout_get(B, n_seg, lay))
        bufs = dict(
            n_seg=n_seg, B=B,
            ws=torch.empty(lib.aread_model_workspace_bytes(self._handle, B, n_seg), dtype=torch.uint8, device=device),
            e=torch.empty((int(lay.max_rows), self.embed_output_dim), dtype=torch.float32, device=device),
            de=torch.empty((int(lay.max_rows), self.embed_output_dim), dtype=torch.float32, device=device),
            de_rw=torch.empty((int(lay.max_rows), self.embed_output_dim), dtype=torch.float32, device=device),
            probs=torch.empty((self.n_heads, B), dtype=torch.float32, device=device),
            loss=torch.zeros(1 + n_seg, dtype=torch.float32, device=device),
            reg=torch.zeros(257, dtype=torch.float32, device=device),
            reg_dense=torch.zeros(257, dtype=torch.float32, device=device),
            total=torch.zeros(1, dtype=torch.float32, device=device),
            gdense=torch.zeros_like(self.dense),
            gtable=torch.empty_like(self.embedding.embedding_dict.weight.data) if with_table_grad else None,
        )
        return bufs

    def prepare_batch(self, x, multi_domain=True, sort=True, reuse=None):
        """Input-pipeline stage of the fused step: everything of a batch that depends only on its ids -- the row plan
        (run.py:310-353's per-domain loaders become one bucketing pass) and the index sort of the embedding backward -- on the
        model's prefetch stream, forked from the current stream HERE, so that it runs while the previous step computes.
        Returns a PreparedBatch for train_step(..., prepared=).  reuse: a PreparedBatch of the same shape whose buffers are
        overwritten (two of them alternate in a training loop)."""
        L.require_device(x)
        n_seg = self.n_domain if multi_domain else 1
        emb = self.embedding
        pb = PreparedBatch()
        pb.x, pb.n_seg = x, n_seg
        need = emb.bwd_ws_bytes(x) if sort else 0
        buf = reuse.plan.buf if reuse is not None else None
        pb.sort_ws = None
        if sort:
            pb.sort_ws = reuse.sort_ws if (reuse is not None and reuse.sort_ws is not None and reuse.sort_ws.numel() >= need) \
                else torch.empty(need, dtype=torch.uint8, device=x.device)
        if buf is None:                      # allocated on the caller's stream (the step that consumes it runs there)
            lay = L.PlanLayout()
            L.check(L.lib().aread_plan_layout_get(x.shape[0], n_seg, lay))
            buf = torch.empty(int(lay.words), dtype=torch.int32, device=x.device)
        main = torch.cuda.current_stream()
        key = "pf:" + str(x.device)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=x.device)
        pf = self._streams[key]
        pf.wait_stream(main)
        with torch.cuda.stream(pf):
            pb.plan = RowPlan(x, self.domain_idx if n_seg > 1 else -1, n_seg, buf=buf)
            if sort:
                emb.sort_lookups(x, pb.plan.sample_row, ws=pb.sort_ws)
            pb.ready = torch.cuda.Event()
            pb.ready.record(pf)
        return pb

    _SEED_STEP = 0x9E3779B1 - (1 << 32)     # the host-side seed sequence's increment, as an int32

    def device_dropout_seed(self, seed=0):
        """Keep the dropout seed in device memory (aread_call.drop_seed_dev) from now on: the kernels read it at run time and
        train_step advances it at its end ON THE STREAM, so a step captured into a hipGraph draws a new dropout mask on every
        replay (a host-side seed is a kernel argument and would be baked into the capture).  Returns the int32[1] tensor
        (assign to it to pin a value); `model.drop_seed_dev = None` goes back to the host-side sequence."""
        self.drop_seed_dev = torch.tensor([int(seed) - (1 << 32) if int(seed) >= (1 << 31) else int(seed)], dtype=torch.int32,
                                          device=self.dense.device)
        return self.drop_seed_dev

    def _mark(self, name):
        """diagnostics only (tools/step_anatomy.py): a timing event on the current stream when a collector is installed"""
        if self._marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._marks.append((name, ev))

    def _side_stream(self, device):
        key = str(device)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def step_local(self, x, y, bufs, masks_dev=None, domain_i=None, seg_weight=None, with_reg=True,
                   with_dense_l2=True, want_gates=False, presort=True, plan=None, e_ready=False, l2_target=None,
                   table_pass=True, prepared=None, dense_l2_first=False):
        """Rank-local part of the step: row plan, gather, dense forward + bagging BCE + backward on the current
        stream; concurrently on a side stream (fork-join, capturable): the table L2 pass
        (bufs['gtable'] = 2*l2*W, bufs['reg'] = l2 terms) and, with presort, the index sort of the embedding
        backward.  Leaves bufs['de'] (gradient w.r.t. the pooled embedding, plan order) for step_scatter.
        No host sync, no allocation besides the row plan.
        e_ready / l2_target=(rows, grad): row-sharded table -- bufs['e'] is already filled and the L2 pass runs over
        this rank's shard instead of the full table.  table_pass=False: no dense table gradient at all (the fused
        optimizer folds the L2 term into its own pass, optim.FusedAdam)."""
        lib = L.lib()
        n_seg = bufs["n_seg"]
        table, gtable = (self.embedding.embedding_dict.weight, bufs["gtable"]) if l2_target is None else l2_target
        if masks_dev is None:
            masks_dev = self._masks_dev(self.domain_mask, x.device)
        main, side = torch.cuda.current_stream(), self._side_stream(x.device)
        part = self._l2_partials(x.device)
        self.embedding._ws_for(x)                      # allocate on the main stream's pool before forking
        # Host issue order matters (eager launches): row plan, then the table L2 pass on the side stream, then the main
        # stream gets its whole forward; only then the index sort (which needs the row plan, not the forward) is queued
        # on the side stream, so the main stream never waits for the host.
        self._mark("step start")
        # the call first: its plan-independent preparation (mask tables, weight images, tag memsets, transposed weights) forks
        # onto the library's side stream NOW, ahead of the row plan and the gather
        pre = self._make_call(x.device, x.shape[0], 0, n_seg, domain_i, masks_dev, want_gates, y, seg_weight, bufs["loss"],
                              bufs["ws"], bufs["probs"], True)
        dense_first = bool(dense_l2_first and with_reg and with_dense_l2 and self.prepare_early)
        if dense_first:
            # dense half of get_regularization_loss at the HEAD of the step, on the library's side stream (aread_prepare): gdense
            # starts as 2*coef*w and the backward adds every gradient onto it (bitwise the same sums as adding the L2 term last)
            pre[0].l2_dense_coef = L.ptr(self._l2_coef(self.dense.device))
            pre[0].init_grads, pre[0].init_reg_out = L.ptr(bufs["gdense"]), L.ptr(bufs["reg_dense"])
        if self.prepare_early:
            L.check(lib.aread_prepare(self._handle, C.byref(pre[0]), L.stream()))
        pre[0].init_grads = None
        if prepared is not None:
            if prepared.n_seg != n_seg or prepared.x.data_ptr() != x.data_ptr():
                raise ValueError("train_step: the PreparedBatch belongs to another batch / segment layout")
            # a wait on an event that has already completed still costs the main stream ~5 us: the plan was queued a whole step ago
            # on the idle prefetch stream, so it usually HAS completed by the time the host gets here -- ask before waiting
            if torch.cuda.is_current_stream_capturing() or not prepared.ready.query():
                main.wait_event(prepared.ready)
            plan = prepared.plan
            presort = False                            # the index sort is part of the prepared batch
        if plan is None:                               # first on the main stream: everything else waits for it, and it is
            plan = RowPlan(x, self.domain_idx if n_seg > 1 else -1, n_seg)     # latency-bound (slow next to the HBM-saturating L2 pass)
        self._mark("row plan")
        # The table L2 pass (356 MB of HBM traffic, needed only by the embedding reduction at the very end) goes on the side
        # stream BEHIND the forward (AREAD_L2_EARLY=1: right after the row plan, as in round 1): next to the latency-bound head
        # of the step (gather, first GEMM) it cost those kernels 2-3x their isolated time.
        used = {"side": False}                         # was anything queued on the torch side stream in this step?

        def l2_pass():
            used["side"] = True
            side.wait_stream(main)
            with torch.cuda.stream(side):
                if not table_pass:
                    bufs["reg"].zero_()
                elif with_reg:
                    L.check(lib.aread_l2_table_throttled(L.ptr(table), table.numel(), self.l2_reg_embedding, 1.0, L.ptr(gtable),
                                                         L.ptr(part), self.l2_pass_workgroups, L.stream()))
                    L.check(lib.aread_l2_finish(L.ptr(part), part.numel(), self.l2_reg_embedding, L.ptr(bufs["reg"]), 0,
                                                L.stream()))
                else:
                    gtable.zero_()
                    bufs["reg"].zero_()
        if self.l2_pass_early:
            l2_pass()
        if presort and self.sort_early:
            # the index sort of the embedding backward needs only the row plan: issued (captured) BEFORE the forward so that a
            # graph replay schedules it beside the expert forward instead of behind everything else at the tail of the step
            used["side"] = True
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.embedding.sort_lookups(x, plan.sample_row)
        elif presort:
            plan_ready = torch.cuda.Event()
            plan_ready.record(main)
        st, gate = self._run(x, 0, n_seg, domain_i, masks_dev, want_gates, e=bufs["e"], plan=plan, e_ready=e_ready, pre=pre)
        self._mark("gather + forward")
        in_bwd = self.l2_pass_in_backward and not self.l2_pass_early and table_pass and with_reg
        if not self.l2_pass_early and not in_bwd:
            l2_pass()
        if presort and not self.sort_early:
            used["side"] = True
            side.wait_event(plan_ready)
            with torch.cuda.stream(side):
                self.embedding.sort_lookups(x, plan.sample_row)
        st.call.async_tail = 3          # parameter gradients finish on the library's side stream: see step_finish
        if in_bwd:                      # the table L2 sweep is issued by the backward, beside the latency-bound tower backward
            st.call.l2_table, st.call.l2_n, st.call.l2_coef = L.ptr(table), table.numel(), self.l2_reg_embedding
            st.call.l2_workgroups = self.l2_pass_workgroups or 1024  # four workgroups per CU.  (256 was the choice while the sweep stole issue
            # slots from the tower kernel it runs beside; since that kernel raises its wave priority the wider sweep is done sooner and
            # costs it less: 0.7481 -> 0.7421 ms/step, profiles/r03_ab_variants.txt section 15)
            st.call.l2_grad, st.call.l2_partial, st.call.l2_reg_out = L.ptr(gtable), L.ptr(part), L.ptr(bufs["reg"])
        st.call.grads_init = 1 if dense_first else 0      # gdense = 2*coef*w was queued on the library's side stream by aread_prepare
        split_de = self.split_de and self._train_step_owner and bufs.get("de_rw") is not None    # only train_step adds the two shares (other callers read bufs["de"] whole)
        st.call.de_rw = L.ptr(bufs["de_rw"]) if split_de else None
        self._de_split = bool(split_de)
        L.check(lib.aread_backward(self._handle, C.byref(st.call), L.ptr(st.e), None, L.ptr(bufs["gdense"]),
                                   L.ptr(bufs["de"]), L.stream()))
        st.call.async_tail = 0
        st.call.grads_init = 0
        st.call.de_rw = None
        st.call.l2_table = None
        st.call.l2_dense_coef = None
        self._mark("backward (main stream: to the last dgrad)")
        if used["side"]:                # table L2 pass / index sort queued there (a wait on an idle stream still costs the
            main.wait_stream(side)      # main stream ~5 us: with a prepared batch and the sweep inside the backward nothing is)
        self._pending_dense_l2 = bool(with_reg and with_dense_l2) and not dense_first
        self._dense_l2_done_first = dense_first
        self._last = (st, gate)
        return st

    def step_finish(self, bufs):
        """Join the library's side stream (dense parameter gradients complete) and add the dense L2 term."""
        L.check(L.lib().aread_join(self._handle, L.stream()))
        if self._pending_dense_l2:
            self.add_dense_l2(bufs)
            self._pending_dense_l2 = False

    def add_dense_l2(self, bufs):
        """reg += sum coef*w^2 over the dense tensors, gdense += 2*coef*w (once per step, after any all-reduce)."""
        L.check(L.lib().aread_l2_dense(L.ptr(self.dense), L.ptr(self._l2_coef(self.dense.device)), self.dense.numel(),
                                       L.ptr(bufs["gdense"]), L.ptr(bufs["reg"]), 1, L.stream()))

    def step_scatter(self, x, de, sample_row, gtable):
        """gtable[g] += contributions of (x, de): the embedding backward (sorted segmented reduction)."""
        self.embedding.scatter_grad(x, de, gtable, sample_row)

    def train_step(self, x, y, bufs, masks_dev=None, domain_i=None, seg_weight=None, with_reg=True, set_grads=True,
                   want_gates=False, prepared=None):
        """forward + bagging BCE + L2 + backward to every parameter gradient, no host sync, no allocation
        besides the row plan (run.py:668-680 without the optimizer).  y: float32 [B] on the device.
        prepared: the batch's PreparedBatch (prepare_batch(x), issued one step ahead): row plan and index sort are then not
        part of this step's dependency chain.  Returns the device scalar loss = sum_d w_d*bag_d + reg."""
        self._train_step_owner = True       # (step_local may keep the row-wise share of dL/de apart: reduce_sorted below adds the two)
        try:
            st = self.step_local(x, y, bufs, masks_dev, domain_i, seg_weight, with_reg, True, want_gates, presort=True,
                                 prepared=prepared, dense_l2_first=self.l2_dense_first)
        finally:
            self._train_step_owner = False
        # tail, all on the main stream (every cross-stream hop costs more than the few small kernels it could overlap): the
        # segmented reduction into the table gradient, then the join with the library's parameter-gradient reductions
        # (long finished by then) and the dense L2 term
        self._mark("join index sort")
        self.embedding.reduce_sorted(x, bufs["de"], bufs["gtable"], ws=prepared.sort_ws if prepared is not None else None,
                                     dout2=bufs["de_rw"] if self._de_split else None)
        self._mark("table-gradient segmented reduction")
        if self._dense_l2_done_first:   # join + the step's two final scalars (the dense L2 terms were formed at the head)
            L.check(L.lib().aread_join(self._handle, L.stream()))
            L.check(L.lib().aread_step_total(L.ptr(bufs["loss"]), L.ptr(bufs["reg_dense"]), L.ptr(bufs["reg"]), L.ptr(bufs["total"]),
                                             L.stream()))
        elif self._pending_dense_l2:    # join + dense L2 terms + total = loss + reg in one launch
            L.check(L.lib().aread_join(self._handle, L.stream()))
            L.check(L.lib().aread_l2_dense_total(L.ptr(self.dense), L.ptr(self._l2_coef(self.dense.device)), self.dense.numel(),
                                                 L.ptr(bufs["gdense"]), L.ptr(bufs["reg"]), 1, L.ptr(bufs["loss"]),
                                                 L.ptr(bufs["total"]), L.stream()))
            self._pending_dense_l2 = False
        else:
            self.step_finish(bufs)
            torch.add(bufs["loss"][:1], bufs["reg"][:1], out=bufs["total"])
        self._mark("join parameter gradients + dense L2 + total")
        if self.drop_seed_dev is not None and self.training and self.dropout > 0:
            self.drop_seed_dev.add_(self._SEED_STEP)        # (behind every reader of this step's value, stream-ordered; captured with the step)
        if set_grads:
            for p in self._dparams:
                p.grad = None
            self.__dict__["_ghave"] = 0
            present = [a or b for a, b in zip(self._presence(0, self.domain_mask), self._reg_present)] if with_reg \
                else self._presence(0, self.domain_mask)
            self._accumulate_dense(bufs["gdense"], present, take=True)
            self.embedding.embedding_dict.weight.grad = bufs["gtable"]
        return bufs["total"]
