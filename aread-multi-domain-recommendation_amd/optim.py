"""Fused training step with the optimizer inside: forward + bagging BCE + L2 + backward + Adam,
no dense 178 MB table gradient and no per-tensor optimizer launches.

Same update as the reference's step closure (run.py:668-682) followed by
torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd) (run.py:830-831):
  * embedding table: g = 2*l2*W + scatter(dE) is never materialised -- aread_route_build (one rank) deduplicates the
    batch's rows, aread_embed_bwd reduces dE into one gradient row per looked-up table row, aread_adam_table_l2
    streams W, m, v once (6 x 4 B per element instead of 9 x) and also yields sum(W^2) for the loss value;
  * dense tensors: aread_adam_step over the flat buffer; tensors the reference's autograd would leave at grad=None
    (towers no mask reaches) are skipped exactly as torch.optim.Adam skips them, with per-tensor step counts."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .plan import RowPlan


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step", C.c_int32)]


class FusedAdam:
    def __init__(self, model, B, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8):
        self.model = model
        self.hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay))
        table = model.embedding.embedding_dict.weight.data
        dev = table.device
        lib = L.lib()
        self.bufs = model.make_step_buffers(B, multi_domain=True, with_table_grad=False)
        self.m_table, self.v_table = torch.zeros_like(table), torch.zeros_like(table)
        self.m_dense, self.v_dense = torch.zeros_like(model.dense), torch.zeros_like(model.dense)
        self.t_table = 0
        self.t_dense = np.zeros(len(model._ptensors), dtype=np.int64)          # torch keeps one step count per tensor
        f_in = model.embedding.offsets.shape[0]
        self.cap = B * f_in                                                     # unique rows <= lookups
        self.route_ws = torch.zeros(int(lib.aread_route_ws_bytes(table.shape[0], 1)), dtype=torch.uint8, device=dev)
        self.slot = torch.empty((B, f_in), dtype=torch.int32, device=dev)
        self.uniq = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.edges = torch.empty(2, dtype=torch.int32, device=dev)
        self.g_rows = torch.empty((self.cap, table.shape[1]), dtype=torch.float32, device=dev)
        self.zero_off = torch.zeros(f_in, dtype=torch.int32, device=dev)
        self.sort_ws = torch.empty(int(lib.aread_embed_bwd_ws_bytes(B, f_in, table.shape[1])), dtype=torch.uint8, device=dev)
        self.part = torch.empty(lib.aread_l2_partials(), dtype=torch.float32, device=dev)
        self.part_rows = torch.empty(lib.aread_adam_row_partials(), dtype=torch.float32, device=dev)
        self.total = torch.zeros(1, dtype=torch.float32, device=dev)
        self._side = torch.cuda.Stream(device=dev)
        self._masks_key, self._present = None, None
        self._active = {}

    # ---- which dense tensors get a gradient (host logic, cached per mask set) -----------------------------------------
    def _present_for(self, masks):
        # model.domain_mask is rewritten IN PLACE by HEMP (hemp.py update_all_mask: self.domain_mask[d] = ...), so the list's
        # identity says nothing: the model counts every assignment (AREAD.mask_version).  Any other mask list is keyed by its
        # content (device tensors cost a copy each: pass numpy / CPU masks, or install them as model.domain_mask).
        m_ = self.model
        if masks is m_.domain_mask:
            key = ("version", m_.mask_version)
        else:
            key = b"|".join(b"-" if mk is None else
                            b"".join(np.packbits(np.asarray(t.cpu() if isinstance(t, torch.Tensor) else t, dtype=bool)).tobytes() for t in mk)
                            for mk in masks)
        if key != self._masks_key:
            m = self.model
            self._present = np.array([a or b for a, b in zip(m._presence(0, masks), m._reg_present)], dtype=bool)
            self._masks_key = key
        return self._present

    def _active_mask(self, sel):
        """uint8 [n_dense]: 1 on the elements of the selected tensors"""
        key = sel.tobytes()
        if key not in self._active:
            m = self.model
            a = np.zeros(m.dense.numel(), dtype=np.uint8)
            for on, (name, kind, off, shape, l2) in zip(sel, m._ptensors):
                if on:
                    a[off:off + (int(np.prod(shape)) if shape else 1)] = 1
            self._active[key] = torch.from_numpy(a).to(m.dense.device)
        return self._active[key]

    def _cfg(self, step):
        c = AdamCfg()
        c.lr, c.beta1, c.beta2, c.eps, c.weight_decay = self.hyper
        c.step = int(step)
        return c

    # ---- one training step ---------------------------------------------------------------------------------------------
    def step(self, x, y, masks_dev=None, masks=None):
        """forward + loss + backward + Adam on every parameter; returns the device scalar loss (pre-update weights).
        masks: the per-domain mask list that masks_dev packs (default: model.domain_mask)."""
        m, b = self.model, self.bufs
        lib = L.lib()
        emb = m.embedding
        table = emb.embedding_dict.weight.data
        masks = m.domain_mask if masks is None else masks
        if masks_dev is None:
            masks_dev = m._masks_dev(masks, x.device)
        Bn, f_in = x.shape
        E = emb.embed_dim
        main, side = torch.cuda.current_stream(), self._side
        plan = RowPlan(x, m.domain_idx, m.n_domain)
        # side stream: dedupe of the batch's rows + index sort of the row-gradient reduction (ids only)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            L.check(lib.aread_route_build(L.ptr(x), Bn, f_in, L.ptr(emb._offsets_dev(x.device)), table.shape[0], 1,
                                          L.ptr(self.route_ws), L.ptr(self.slot), L.ptr(self.uniq), L.ptr(self.edges), 1,
                                          L.stream()))
            L.check(lib.aread_embed_bwd_sort(L.ptr(self.slot), Bn, f_in, L.ptr(self.zero_off), self.cap, E,
                                             emb.one_hot_field_num, emb.multi_hot_field_num, emb.seq_maxlen, emb._pool,
                                             L.ptr(plan.sample_row), L.ptr(self.sort_ws), L.stream()))
            self.g_rows.zero_()
            # rows the batch does not touch: L2-only Adam update now, concurrently with the forward/backward (the
            # gather reads exactly the other rows)
            self.t_table += 1
            cfg = self._cfg(self.t_table)
            L.check(lib.aread_adam_table_l2(L.ptr(table), L.ptr(self.m_table), L.ptr(self.v_table), table.shape[0], E,
                                            L.ptr(self.route_ws), None, None, None, m.l2_reg_embedding, C.byref(cfg), 1,
                                            L.ptr(self.part), L.stream()))
        m.step_local(x, y, b, masks_dev=masks_dev, with_dense_l2=False, presort=False, plan=plan, table_pass=False)
        main.wait_stream(side)
        seq = emb.seq_maxlen if emb._pool != 0 else 1
        L.check(lib.aread_embed_bwd_reduce(Bn, f_in, E, seq, L.ptr(b["de"]), L.ptr(self.g_rows), L.ptr(self.sort_ws),
                                           L.stream()))
        L.check(lib.aread_adam_table_l2(L.ptr(table), L.ptr(self.m_table), L.ptr(self.v_table), table.shape[0], E,
                                        L.ptr(self.route_ws), L.ptr(self.uniq), L.ptr(self.edges), L.ptr(self.g_rows),
                                        m.l2_reg_embedding, C.byref(cfg), 2, L.ptr(self.part_rows), L.stream()))
        L.check(lib.aread_l2_finish(L.ptr(self.part), self.part.numel(), m.l2_reg_embedding, L.ptr(b["reg"]), 0, L.stream()))
        L.check(lib.aread_l2_finish(L.ptr(self.part_rows), self.part_rows.numel(), m.l2_reg_embedding, L.ptr(b["reg"]), 1,
                                    L.stream()))
        m.step_finish(b)                                    # dense gradients complete
        m.add_dense_l2(b)                                   # reg += dense terms, gdense += 2*coef*w
        torch.add(b["loss"][:1], b["reg"][:1], out=self.total)
        present = self._present_for(masks)
        self.t_dense[present] += 1
        for t in np.unique(self.t_dense[present]):
            sel = present & (self.t_dense == t)
            act = None if sel.all() else self._active_mask(sel)
            cfg = self._cfg(t)
            L.check(lib.aread_adam_step(L.ptr(m.dense), L.ptr(b["gdense"]), L.ptr(self.m_dense), L.ptr(self.v_dense),
                                        m.dense.numel(), L.ptr(act), C.byref(cfg), L.stream()))
        return self.total


class Adam(torch.optim.Optimizer):
    """Drop-in for `torch.optim.Adam(model.parameters(), lr, betas, eps, weight_decay)` in the reference's step loop
    (run.py:830-831, 680-681): the same update, but the 178 MB table and the flat dense buffer take ONE kernel launch each
    instead of ~300 per-tensor launch groups.  Construct it with the MODEL (it needs to know that the dense parameters are
    views of one buffer); `zero_grad` / `step` behave as usual, `state_dict` / `load_state_dict` carry the flat moment buffers
    and torch's per-tensor step counts (a resumed run continues bit for bit).  Tensors whose grad is None are skipped
    exactly as torch does (no decay, no moment update, no step count)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(list(model.parameters()), defaults)
        self.model = model
        self._t_table = 0
        self._t_dense = np.zeros(len(model._ptensors), dtype=np.int64)
        self._m_table = self._v_table = self._m_dense = self._v_dense = None
        self._active = {}

    def _cfg(self, step):
        g = self.param_groups[0]
        c = AdamCfg()
        c.lr, c.beta1, c.beta2, c.eps, c.weight_decay = g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]
        c.step = int(step)
        return c

    def zero_grad(self, set_to_none: bool = True):
        """the optimizer holds exactly the model's parameters: the model's zero_grad keeps its gradient bookkeeping exact"""
        self.model.zero_grad(set_to_none=set_to_none)

    # ---- checkpointing (run.py:459-484 stores optimizer.state_dict(); harness.is_continuable does the same) -------------
    def state_dict(self):
        """torch's layout ({'state', 'param_groups'}) with the moments where they really live: 'state' holds the two flat
        moment buffers of the table and of the dense parameters plus torch's step counts (one for the table, one per
        dense tensor: tensors whose grad was None never advanced)."""
        sd = super().state_dict()
        sd["state"] = {"flat": {
            "m_table": None if self._m_table is None else self._m_table.clone(),
            "v_table": None if self._v_table is None else self._v_table.clone(),
            "m_dense": None if self._m_dense is None else self._m_dense.clone(),
            "v_dense": None if self._v_dense is None else self._v_dense.clone(),
            "t_table": int(self._t_table), "t_dense": torch.from_numpy(self._t_dense.copy())}}
        return sd

    def load_state_dict(self, state_dict):
        flat = state_dict.get("state", {}).get("flat")
        if flat is None:
            raise ValueError("aread_amd.Adam.load_state_dict: not a state_dict of aread_amd.Adam (no 'flat' moment buffers)")
        groups = state_dict["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, src in zip(self.param_groups, groups):
            g.update({k: v for k, v in src.items() if k != "params"})
        dev = self.model.dense.device
        put = lambda t: None if t is None else t.detach().to(dev, torch.float32).clone()
        self._m_table, self._v_table = put(flat["m_table"]), put(flat["v_table"])
        self._m_dense, self._v_dense = put(flat["m_dense"]), put(flat["v_dense"])
        self._t_table = int(flat["t_table"])
        t = np.asarray(flat["t_dense"].cpu() if isinstance(flat["t_dense"], torch.Tensor) else flat["t_dense"], dtype=np.int64)
        if t.shape != self._t_dense.shape:
            raise ValueError("loaded state dict does not match this model's dense tensors")
        self._t_dense = t.copy()

    def _active_mask(self, sel):
        key = sel.tobytes()
        if key not in self._active:
            m = self.model
            a = np.zeros(m.dense.numel(), dtype=np.uint8)
            for on, (name, kind, off, shape, l2) in zip(sel, m._ptensors):
                if on:
                    a[off:off + (int(np.prod(shape)) if shape else 1)] = 1
            self._active[key] = torch.from_numpy(a).to(m.dense.device)
        return self._active[key]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self.model
        lib = L.lib()
        table = m.embedding.embedding_dict.weight
        if table.grad is not None:
            if self._m_table is None:
                self._m_table, self._v_table = torch.zeros_like(table.data), torch.zeros_like(table.data)
            g = table.grad.contiguous()
            self._t_table += 1
            cfg = self._cfg(self._t_table)
            L.check(lib.aread_adam_step(L.ptr(table.data), L.ptr(g), L.ptr(self._m_table), L.ptr(self._v_table), table.numel(),
                                        None, C.byref(cfg), L.stream()))
        present = m._grad_presence()
        if present.any():
            if self._m_dense is None:
                self._m_dense, self._v_dense = torch.zeros_like(m.dense), torch.zeros_like(m.dense)
            gflat = m._gflat                                   # every present .grad is a view of this flat buffer
            k = int(np.argmax(present))
            if gflat is None or m._dparams[k].grad.data_ptr() != gflat.data_ptr() + 4 * m._ptensors[k][2]:
                raise RuntimeError("aread_amd.Adam: dense gradients must come from this module's backward (views of one buffer)")
            self._t_dense[present] += 1
            for t in np.unique(self._t_dense[present]):
                sel = present & (self._t_dense == t)
                act = None if sel.all() else self._active_mask(sel)
                cfg = self._cfg(t)
                L.check(lib.aread_adam_step(L.ptr(m.dense), L.ptr(gflat), L.ptr(self._m_dense), L.ptr(self._v_dense),
                                            m.dense.numel(), L.ptr(act), C.byref(cfg), L.stream()))
        return loss
