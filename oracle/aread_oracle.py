"""CPU oracle for the AREAD hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path
(``aread_amd``) never routes through it and fails loudly when ``libaread_hip.so``
is missing.

What it is: a *functional* (no nn.Module) fp32 restatement, on torch CPU tensors,
of the math of the reference's CTR forward/backward path, written from the
specification in SURVEY.md Appendix A.  Gradients come from torch autograd on the
CPU.  Integer index-bag arithmetic is done in numpy int32.

Reference sites restated (paths relative to /root/reference):
  * index bag + pooling ............ model/layer.py:150-178   -> index_bag(), embed_pool()
  * linear term .................... model/layer.py:115-126   -> trunk()
  * cross network .................. model/layer.py:529-537   -> trunk()
  * MLP block (Linear,BN,ReLU,Drop)  model/layer.py:209-229   -> mlp_stack()
  * MMoE bottom .................... model/aread.py:150-153   -> trunk()
  * masked HEI towers + heads ...... model/aread.py:263-322   -> hei_masked()
  * unmasked HEI (warm-up) ......... model/aread.py:156-202   -> hei_plain()
  * mode dispatch .................. model/aread.py:224-244   -> forward()
  * L2 regulariser ................. model/layer.py:96-112, model/aread.py:102-104,123-127 -> reg_loss()
  * step closure ................... run.py:672-682           -> bagging_loss(), step()

Parity pin: tests/golden/*.npz were produced by tests/golden/make_golden.py, which
imports the real reference in the build container and records its outputs for the
parameters of init_params(); tests/test_oracle_golden.py checks this file against
them.  (Dropout cannot be RNG-matched to torch, so the goldens use p=0; for p>0 this
oracle and the HIP kernels share the counter-based hash in dropout_keep().)
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
GATE_EPS = 1e-8


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class Spec:
    """Shape contract of the path (values: config.py / run.py:377-457 of the reference)."""
    field_dims: Sequence[int]                 # cardinality of each one-hot field
    embed_dim: int = 32
    multi_hot_flag: Sequence[bool] = ()       # len F_in; one-hot columns first, then history slots
    itemid_idx: int = 0
    seq_maxlen: int = 5
    method: Optional[str] = "mean"            # 'mean' | 'sum' | None
    n_tower: Sequence[int] = (3, 6, 12)
    n_domain: int = 25
    n_expert: int = 4
    expert_dims: Sequence[int] = (256, 128, 64)
    tower_dims: Sequence[Sequence[int]] = ((64, 32), (32, 16), (16, 8))
    domain_idx: int = 2
    n_cross: int = 3
    dropout: float = 0.0
    l2_embedding: float = 1e-5
    l2_linear: float = 1e-5
    l2_dnn: float = 1e-5
    l2_cross: float = 1e-5
    # dead attention branch: parameters exist in the reference state_dict, never affect outputs
    atten_embed_dim: int = 64
    att_layer_num: int = 3
    with_dead_attention: bool = True

    def __post_init__(self):
        if not self.multi_hot_flag:
            self.multi_hot_flag = [False] * len(self.field_dims)
        if self.method not in ("mean", "sum", None):
            raise ValueError(f"Invalid multi-hot method '{self.method}'.")

    # derived
    @property
    def n_onehot(self): return len(self.field_dims)
    @property
    def n_mh_slots(self): return int(sum(self.multi_hot_flag))
    @property
    def n_mh_fields(self): return self.n_mh_slots // self.seq_maxlen if self.n_mh_slots else 0
    @property
    def f_in(self): return len(self.multi_hot_flag)
    @property
    def f_out(self):
        if self.method in ("mean", "sum"):
            return self.n_onehot + self.n_mh_fields
        return self.n_onehot + self.n_mh_slots
    @property
    def d(self): return self.f_out * self.embed_dim
    @property
    def rows(self): return int(sum(self.field_dims))
    @property
    def n_level(self): return len(self.n_tower)
    @property
    def edge_num(self):
        n = self.n_tower
        return n[0] + sum(n[l - 1] * n[l] for l in range(1, len(n))) + n[-1]

    def offsets(self) -> np.ndarray:
        """layer.py:151-157: exclusive cumsum for one-hot fields; history slots reuse the itemid offset."""
        off = np.concatenate(([0], np.cumsum(self.field_dims)[:-1])).astype(np.int64)
        if self.n_mh_fields > 0:
            off = np.concatenate((off, np.full(self.n_mh_slots, off[self.itemid_idx], dtype=np.int64)))
        return off


def amazon_spec(**kw) -> Spec:
    """BASELINE configs 2-4: Amazon-like 25-domain layout."""
    base = dict(field_dims=[1368287, 7, 25, 45, 11, 22356, 10], embed_dim=32,
                multi_hot_flag=[False] * 7 + [True] * 10, itemid_idx=0, seq_maxlen=5, method="mean",
                n_tower=(3, 6, 12), n_domain=25, domain_idx=2)
    base.update(kw)
    return Spec(**base)


def aliccp_spec(**kw) -> Spec:
    """BASELINE config 5: AliCCP-like 30-domain layout, no multi-hot."""
    dims = [211161, 95, 14, 3, 8, 4, 4, 3, 5, 41775, 30, 284915, 81491, 112993, 1929, 118091, 54472, 34677,
            5821, 106908, 54295, 31716, 4]
    base = dict(field_dims=dims, embed_dim=32, multi_hot_flag=[False] * 23, itemid_idx=9, method=None,
                n_tower=(3, 6, 12), n_domain=30, domain_idx=10)
    base.update(kw)
    return Spec(**base)


# --------------------------------------------------------------------------------------
# parameters (names = the reference's state_dict keys)
# --------------------------------------------------------------------------------------
def param_shapes(spec: Spec) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, kind). kind in {emb, w, b, gamma, beta, rmean, rvar, count, zero}."""
    E, D = spec.embed_dim, spec.d
    out: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    out["embedding.embedding_dict.weight"] = ((spec.rows, E), "emb")
    out["linear.fc.weight"] = ((1, D), "w")
    out["linear.fc.bias"] = ((1,), "b")
    out["group_embedding.weight"] = ((spec.n_tower[0], E), "emb")
    out["final_gate.0.weight"] = ((spec.n_tower[-1], 2 * E), "w")
    for i in range(spec.n_cross):
        out[f"cn.w.{i}.weight"] = ((1, D), "w")
        out[f"cn.b.{i}"] = ((D,), "b")
    if spec.with_dead_attention:
        A = spec.atten_embed_dim
        out["atten_embedding.weight"] = ((A, E), "w")
        out["atten_embedding.bias"] = ((A,), "b")
        for i in range(spec.att_layer_num):
            out[f"self_attns.{i}.in_proj_weight"] = ((3 * A, A), "w")
            out[f"self_attns.{i}.in_proj_bias"] = ((3 * A,), "b")
            out[f"self_attns.{i}.out_proj.weight"] = ((A, A), "w")
            out[f"self_attns.{i}.out_proj.bias"] = ((A,), "b")
        out["V_res_embedding.weight"] = ((A, E), "w")
        out["V_res_embedding.bias"] = ((A,), "b")
        out["atten_linear.weight"] = ((1, spec.f_out * A), "w")

    def mlp(prefix, in_dim, dims):
        for j, h in enumerate(dims):
            out[f"{prefix}.layers.{4 * j}.weight"] = ((h, in_dim), "w")
            out[f"{prefix}.layers.{4 * j}.bias"] = ((h,), "b")
            out[f"{prefix}.layers.{4 * j + 1}.weight"] = ((h,), "gamma")
            out[f"{prefix}.layers.{4 * j + 1}.bias"] = ((h,), "beta")
            out[f"{prefix}.layers.{4 * j + 1}.running_mean"] = ((h,), "rmean")
            out[f"{prefix}.layers.{4 * j + 1}.running_var"] = ((h,), "rvar")
            out[f"{prefix}.layers.{4 * j + 1}.num_batches_tracked"] = ((), "count")
            in_dim = h

    for k in range(spec.n_expert):
        mlp(f"mmoe_experts.{k}", D, spec.expert_dims)
    for t in range(spec.n_tower[0]):
        out[f"mmoe_gates.{t}.0.weight"] = ((spec.n_expert, D), "w")
        out[f"mmoe_gates.{t}.0.bias"] = ((spec.n_expert,), "b")
    tin = spec.expert_dims[-1]
    for l in range(spec.n_level):
        for t in range(spec.n_tower[l]):
            mlp(f"towers.{l}.{t}", tin, spec.tower_dims[l])
            if l > 0:
                out[f"tower_gates.{l - 1}.{t}.0.weight"] = ((spec.n_tower[l - 1], 2 * E), "w")
                out[f"tower_gates.{l - 1}.{t}.0.bias"] = ((spec.n_tower[l - 1],), "b")
        tin = spec.tower_dims[l][-1]
    for i in range(spec.n_tower[-1]):
        out[f"towers_linear.{i}.weight"] = ((1, D + spec.tower_dims[-1][-1]), "w")
    return out


def init_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int) -> torch.Tensor:
    """Deterministic, order-independent initialiser: the stream is keyed by (seed, crc32(name))."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if kind == "count":
        return torch.zeros((), dtype=torch.int64)
    if kind == "emb":
        a = rng.standard_normal(shape) * 0.5
    elif kind == "w":
        fan_in = shape[-1]
        a = rng.uniform(-1.0, 1.0, shape) * (1.5 / np.sqrt(fan_in))
    elif kind == "b":
        a = rng.uniform(-0.2, 0.2, shape)
    elif kind == "gamma":
        a = rng.uniform(0.6, 1.4, shape)
    elif kind == "beta":
        a = rng.uniform(-0.3, 0.3, shape)
    elif kind == "rmean":
        a = rng.standard_normal(shape) * 0.2
    elif kind == "rvar":
        a = rng.uniform(0.5, 1.5, shape)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def init_params(spec: Spec, seed: int = 123) -> Dict[str, torch.Tensor]:
    return {n: init_tensor(n, s, k, seed) for n, (s, k) in param_shapes(spec).items()}


def reg_groups(spec: Spec) -> List[Tuple[str, float]]:
    """(name, l2) of every tensor in the reference's regularization_weight lists.

    layer.py:31-33: table and linear.fc.weight; aread.py:102-104: every '*weight*' named parameter
    of mmoe_experts (Linear weights AND BatchNorm gammas: the 'bn' filter never matches because BN
    layers are called layers.N); aread.py:123-124 same for towers; aread.py:125-127 cn.w.*.weight.
    """
    g: List[Tuple[str, float]] = [("embedding.embedding_dict.weight", spec.l2_embedding),
                                  ("linear.fc.weight", spec.l2_linear)]
    for k in range(spec.n_expert):
        for j in range(len(spec.expert_dims)):
            g.append((f"mmoe_experts.{k}.layers.{4 * j}.weight", spec.l2_dnn))
            g.append((f"mmoe_experts.{k}.layers.{4 * j + 1}.weight", spec.l2_dnn))
    for l in range(spec.n_level):
        for t in range(spec.n_tower[l]):
            for j in range(len(spec.tower_dims[l])):
                g.append((f"towers.{l}.{t}.layers.{4 * j}.weight", spec.l2_dnn))
                g.append((f"towers.{l}.{t}.layers.{4 * j + 1}.weight", spec.l2_dnn))
    for i in range(spec.n_cross):
        g.append((f"cn.w.{i}.weight", spec.l2_cross))
    return g


def trainable_names(spec: Spec) -> List[str]:
    return [n for n, (_, k) in param_shapes(spec).items() if k in ("emb", "w", "b", "gamma", "beta")]


# --------------------------------------------------------------------------------------
# masks
# --------------------------------------------------------------------------------------
def full_mask(spec: Spec, value: bool = True) -> List[np.ndarray]:
    n = spec.n_tower
    shapes = [(1, n[0])] + [(n[l - 1], n[l]) for l in range(1, len(n))] + [(n[-1], 1)]
    return [np.full(s, value, dtype=bool) for s in shapes]


def pack_mask(spec: Spec, mask: Sequence) -> np.ndarray:
    """Flatten a mask (list of n_level+1 bool arrays/tensors) to uint8[edge_num], level-major, row-major."""
    parts = []
    for m in mask:
        a = m.cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m)
        parts.append(a.astype(np.uint8).reshape(-1))
    out = np.concatenate(parts)
    assert out.size == spec.edge_num
    return out


def unpack_mask(spec: Spec, flat: np.ndarray) -> List[np.ndarray]:
    ref = full_mask(spec)
    out, p = [], 0
    for m in ref:
        out.append(np.asarray(flat[p:p + m.size]).astype(bool).reshape(m.shape))
        p += m.size
    return out


def random_valid_mask(spec: Spec, rng: np.random.Generator, p_active: float = 0.7) -> List[np.ndarray]:
    """A random mask closed under the validity rules (inputs for used towers, outputs for fed heads,
    dead hidden towers cut) with at least one active head.  Used for synthetic workloads only."""
    n = spec.n_tower
    while True:
        m = [rng.random(s.shape) < p_active for s in full_mask(spec)]
        changed = True
        while changed:
            changed = False
            for l in range(1, len(n)):
                for t in range(n[l - 1]):          # tower t of level l-1: inputs m[l-1][:,t], outputs m[l][t,:]
                    has_in = m[l - 1][:, t].any()
                    has_out = m[l][t, :].any()
                    if l - 1 == 0:
                        if has_out and not has_in:
                            m[0][:, t] = True; changed = True
                        if has_in and not has_out:
                            m[0][:, t] = False; changed = True
                    else:
                        if has_in and not has_out:
                            m[l - 1][:, t] = False; changed = True
                        if has_out and not has_in:
                            m[l][t, :] = False; changed = True
            for t in range(n[-1]):
                want = m[-2][:, t].any()
                if m[-1][t, 0] != want:
                    m[-1][t, 0] = want; changed = True
        if m[-1].any():
            return m


# --------------------------------------------------------------------------------------
# dropout: counter-based hash shared bit-for-bit with the HIP kernels (csrc/common.h)
# --------------------------------------------------------------------------------------
def _mix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x7FEB352D)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    h = (h * np.uint32(0x846CA68B)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_site(stack: int, layer: int, group: int) -> int:
    """Site id of one (MLP stack, layer, group): experts are stack 0, tower level l is stack 1+l."""
    return ((stack * 8 + layer) * 64 + group)


def dropout_keep(seed: int, site: int, sample_ids: np.ndarray, n_cols: int, p: float) -> np.ndarray:
    """keep[b, c] for dropout probability p; bool array [len(sample_ids), n_cols]."""
    with np.errstate(over="ignore"):
        s = np.asarray(sample_ids, dtype=np.uint32)[:, None]
        c = (np.uint32(site) * np.uint32(4096) + np.arange(n_cols, dtype=np.uint32))[None, :]
        h = _mix32((s * np.uint32(0x9E3779B1)) ^ np.uint32(seed & 0xFFFFFFFF))
        h = _mix32(h ^ (c * np.uint32(0x85EBCA77)))
    thr = np.uint32(min(int(round(p * 4294967296.0)), 0xFFFFFFFF))
    return h >= thr


# --------------------------------------------------------------------------------------
# embedding
# --------------------------------------------------------------------------------------
def index_bag(x: np.ndarray, spec: Spec) -> np.ndarray:
    """g = x + offsets, computed in the dtype of x (int32), layer.py:165."""
    x = np.asarray(x)
    return (x + spec.offsets().astype(x.dtype)[None, :]).astype(x.dtype)


def embed_pool(table: torch.Tensor, g: torch.Tensor, spec: Spec) -> torch.Tensor:
    """[B,F_in] global rows -> [B,F_out,E]; history slots are summed in slot order then divided
    by seq_maxlen ('mean', padding included) - layer.py:166-178."""
    rows = table[g.long()]                                    # [B,F_in,E]
    if spec.n_mh_fields > 0 and spec.method in ("mean", "sum"):
        flag = torch.as_tensor(np.asarray(spec.multi_hot_flag, dtype=bool))
        one = rows[:, ~flag, :]
        mh = rows[:, flag, :].reshape(rows.shape[0], spec.n_mh_fields, spec.seq_maxlen, spec.embed_dim)
        acc = mh[:, :, 0, :]
        for s in range(1, spec.seq_maxlen):                   # sequential fp32 sum (bit-exact with ATen)
            acc = acc + mh[:, :, s, :]
        if spec.method == "mean":
            acc = acc / float(spec.seq_maxlen)
        rows = torch.cat((one, acc), dim=1)
    return rows


# --------------------------------------------------------------------------------------
# dense blocks
# --------------------------------------------------------------------------------------
class Ctx:
    """Per-call context: training flag, dropout, running-stat updates, captured intermediates."""
    def __init__(self, spec: Spec, train: bool, sample_ids: Optional[np.ndarray], drop_seed: int,
                 buffers: Dict[str, torch.Tensor]):
        self.spec, self.train, self.sample_ids, self.drop_seed = spec, train, sample_ids, drop_seed
        self.buffers = buffers            # running stats; updated in place on this (copied) dict
        self.cap: Dict[str, torch.Tensor] = {}


def mlp_stack(P, ctx: Ctx, prefix: str, x: torch.Tensor, dims: Sequence[int], stack: int, group: int):
    """[Linear -> BatchNorm1d -> ReLU -> Dropout] x len(dims)  (layer.py:209-229).
    BN is skipped when the call has exactly one row (layer.py:226)."""
    spec = ctx.spec
    for j, h in enumerate(dims):
        W, b = P[f"{prefix}.layers.{4 * j}.weight"], P[f"{prefix}.layers.{4 * j}.bias"]
        x = x @ W.t() + b
        if x.shape[0] != 1:
            bn = f"{prefix}.layers.{4 * j + 1}"
            gamma, beta = P[bn + ".weight"], P[bn + ".bias"]
            if ctx.train:
                mu = x.mean(dim=0)
                var = x.var(dim=0, unbiased=False)
                n = x.shape[0]
                with torch.no_grad():
                    ctx.buffers[bn + ".running_mean"] = ((1 - BN_MOMENTUM) * ctx.buffers[bn + ".running_mean"]
                                                         + BN_MOMENTUM * mu.detach())
                    ctx.buffers[bn + ".running_var"] = ((1 - BN_MOMENTUM) * ctx.buffers[bn + ".running_var"]
                                                        + BN_MOMENTUM * var.detach() * (n / (n - 1)))
                    ctx.buffers[bn + ".num_batches_tracked"] = ctx.buffers[bn + ".num_batches_tracked"] + 1
            else:
                mu, var = ctx.buffers[bn + ".running_mean"], ctx.buffers[bn + ".running_var"]
            x = (x - mu) / torch.sqrt(var + BN_EPS) * gamma + beta
        x = torch.relu(x)
        if ctx.train and spec.dropout > 0.0:
            keep = dropout_keep(ctx.drop_seed, dropout_site(stack, j, group), ctx.sample_ids, h, spec.dropout)
            x = x * torch.from_numpy(keep.astype(np.float32)) / (1.0 - spec.dropout)
    return x


def trunk(P, ctx: Ctx, x_idx: np.ndarray):
    """aread.py:131-153 (attention branch omitted: its result is never read)."""
    spec = ctx.spec
    g = torch.from_numpy(index_bag(x_idx, spec).astype(np.int64))
    emb = embed_pool(P["embedding.embedding_dict.weight"], g, spec)      # [B,F_out,E]
    dom = emb[:, spec.domain_idx, :]
    e = emb.flatten(start_dim=1)
    lin = e @ P["linear.fc.weight"].t() + P["linear.fc.bias"]           # [B,1]
    c = e
    for i in range(spec.n_cross):
        c = e * (c @ P[f"cn.w.{i}.weight"].t()) + P[f"cn.b.{i}"] + c
    experts = [mlp_stack(P, ctx, f"mmoe_experts.{k}", e, spec.expert_dims, 0, k) for k in range(spec.n_expert)]
    X = torch.stack(experts, dim=1)                                      # [B,n_exp,H]
    u = []
    for t in range(spec.n_tower[0]):
        pi = torch.softmax(e @ P[f"mmoe_gates.{t}.0.weight"].t() + P[f"mmoe_gates.{t}.0.bias"], dim=1)
        u.append((pi.unsqueeze(-1) * X).sum(dim=1))
    ctx.cap.update(embed=emb, e=e, lin=lin, cn=c, experts=X, u=torch.stack(u, dim=1))
    return dom, lin, c, u


def _tower(P, ctx, l, t, x):
    return mlp_stack(P, ctx, f"towers.{l}.{t}", x, ctx.spec.tower_dims[l], 1 + l, t)


def hei_masked(P, ctx: Ctx, mask: Sequence[np.ndarray], u, q, cn, lin, want_gate_stats: bool):
    """aread.py:263-322.  Returns (P_active [K,B], active head ids, gate stats per level)."""
    spec = ctx.spec
    n, B = spec.n_tower, lin.shape[0]
    gate_stats: List[Optional[torch.Tensor]] = [None] * spec.n_level
    outs = None
    for l in range(spec.n_level):
        active = np.asarray(mask[l]).any(axis=0)
        width_out = spec.tower_dims[l][-1]
        if l == 0:
            ins = [u[t] if active[t] else torch.zeros(B, width_out) for t in range(n[0])]
        else:
            ins, stats = [], torch.zeros(n[l - 1], n[l])
            for t in range(n[l]):
                if not active[t]:
                    ins.append(torch.zeros(B, width_out))
                    continue
                a = torch.softmax(q @ P[f"tower_gates.{l - 1}.{t}.0.weight"].t()
                                  + P[f"tower_gates.{l - 1}.{t}.0.bias"], dim=1)
                col = torch.from_numpy(np.asarray(mask[l])[:, t].astype(np.float32))
                am = a * col
                ah = am / (am.sum(dim=1, keepdim=True) + GATE_EPS)
                ins.append((ah.unsqueeze(-1) * outs).sum(dim=1))
                stats[:, t] = am.mean(dim=0).detach()
            if want_gate_stats:
                gate_stats[l] = stats
        if l == spec.n_level - 1:
            heads, ids = [], []
            for i in range(n[l]):
                if not active[i]:
                    continue
                h = _tower(P, ctx, l, i, ins[i])
                z = torch.cat([cn, h], dim=1) @ P[f"towers_linear.{i}.weight"].t() + lin
                heads.append(z.squeeze(-1)); ids.append(i)
            if not heads:
                raise RuntimeError("mask has no active last-level tower (aread.py:312-322)")
            logits = torch.stack(heads, dim=0)
            ctx.cap["logits"] = logits
            return torch.sigmoid(logits), ids, gate_stats
        outs = torch.stack([_tower(P, ctx, l, t, ins[t]) if active[t] else ins[t] for t in range(n[l])], dim=1)
        ctx.cap[f"tower_out{l}"] = outs


def hei_plain(P, ctx: Ctx, u, q, cn, lin, want_gate_stats: bool):
    """aread.py:164-186: all towers, plain softmax gates, all heads."""
    spec = ctx.spec
    n = spec.n_tower
    gate_stats: List[Optional[torch.Tensor]] = [None] * spec.n_level
    ins, outs = list(u), None
    for l in range(spec.n_level):
        if l > 0:
            ins, stats = [], torch.zeros(n[l - 1], n[l])
            for t in range(n[l]):
                a = torch.softmax(q @ P[f"tower_gates.{l - 1}.{t}.0.weight"].t()
                                  + P[f"tower_gates.{l - 1}.{t}.0.bias"], dim=1)
                ins.append((a.unsqueeze(-1) * outs).sum(dim=1))
                stats[:, t] = a.mean(dim=0).detach()
            if want_gate_stats:
                gate_stats[l] = stats
        hs = [_tower(P, ctx, l, t, ins[t]) for t in range(n[l])]
        if l == spec.n_level - 1:
            logits = torch.stack([(torch.cat([cn, hs[i]], dim=1) @ P[f"towers_linear.{i}.weight"].t() + lin
                                   ).squeeze(-1) for i in range(n[l])], dim=0)
            ctx.cap["logits"] = logits
            return torch.sigmoid(logits), list(range(n[l])), gate_stats
        outs = torch.stack(hs, dim=1)


def forward(P: Dict[str, torch.Tensor], buffers: Dict[str, torch.Tensor], spec: Spec, x_idx: np.ndarray,
            mode: str = "wo_mask", mask: Optional[Sequence[np.ndarray]] = None, train: bool = True,
            sample_ids: Optional[np.ndarray] = None, drop_seed: int = 0, want_gate_stats: bool = False):
    """One call of the reference's AREAD.forward on one (single-domain) batch.

    Returns dict(y=..., probs=[K,B], heads=[ids], gate_stats=[per level], buffers=new running stats, cap=...).
      mode 'wo_mask'             -> y [B,1]   (aread.py:186)
      mode 'domain_with_mask'    -> y [B]     (aread.py:233)
      mode 'domain_mask_bagging' -> y [K,B]   (aread.py:244)
    """
    x_idx = np.asarray(x_idx)
    if sample_ids is None:
        sample_ids = np.arange(x_idx.shape[0])
    ctx = Ctx(spec, train, sample_ids, drop_seed, dict(buffers))
    dom, lin, cn, u = trunk(P, ctx, x_idx)
    if mode == "wo_mask":
        q = torch.cat([dom, torch.zeros_like(dom)], dim=1)
        probs, ids, gs = hei_plain(P, ctx, u, q, cn, lin, want_gate_stats)
        y = probs.mean(dim=0).unsqueeze(-1)
    elif mode in ("domain_with_mask", "domain_mask_bagging"):
        if mask is None:
            raise ValueError("masked modes need a mask")
        act0 = np.nonzero(np.asarray(mask[0]).reshape(-1))[0]
        if act0.size == 0:
            raise ValueError("mask[0] has no active level-0 tower")
        grp = P["group_embedding.weight"][torch.from_numpy(act0)]
        grp = grp.mean(dim=0, keepdim=True) if grp.shape[0] > 1 else grp
        q = torch.cat([dom, grp.expand(dom.shape[0], -1)], dim=1)
        probs, ids, gs = hei_masked(P, ctx, mask, u, q, cn, lin, want_gate_stats)
        y = probs if mode == "domain_mask_bagging" else probs.mean(dim=0)
    else:
        raise ValueError(f"mode {mode!r} is not on the path")
    return dict(y=y, probs=probs, heads=ids, gate_stats=gs, buffers=ctx.buffers, cap=ctx.cap)


# --------------------------------------------------------------------------------------
# losses and the step closure
# --------------------------------------------------------------------------------------
def bce_mean(p: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """torch.nn.BCELoss(reduction='mean'): logs clamped at -100."""
    return -(y * torch.clamp(torch.log(p), min=-100.0)
             + (1.0 - y) * torch.clamp(torch.log(1.0 - p), min=-100.0)).mean()


def bagging_loss(probs: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """run.py:672-677: (1/K) sum_k BCE_mean(preds[k], y)."""
    return sum(bce_mean(probs[k], y) for k in range(probs.shape[0])) / probs.shape[0]


def reg_loss(P, spec: Spec) -> torch.Tensor:
    """layer.py:96-112: sum over groups of sum(l2 * w^2); shape [1]."""
    total = torch.zeros(1)
    for name, l2 in reg_groups(spec):
        if l2 > 0:
            total = total + torch.sum(l2 * torch.square(P[name]))
    return total


def split_buffers(P):
    keys = [k for k in P if k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    return {k: P[k] for k in keys}


def step(P, spec: Spec, x_idx: np.ndarray, y: np.ndarray, masks: Sequence[Sequence[np.ndarray]],
         mode: str = "domain_mask_bagging", domain_weights: Optional[Sequence[float]] = None,
         drop_seed: int = 0, train: bool = True, with_reg: bool = True, want_grads: bool = True,
         want_gate_stats: bool = False):
    """The 'N-domain batch' step = the reference's per-domain calls in domain order on one model
    instance, one summed loss, one backward:

        L = sum_d w_d * bagging_loss(forward(X[dom==d], domain_i=d, mask_d), y[dom==d]) + reg

    A single-domain batch is the special case with one non-empty domain.  Running BN statistics are
    threaded through the calls in domain order.  Returns loss, per-sample probabilities in a
    [n_heads, B] array (nan where a head is inactive for the sample's domain), grads, new buffers.
    """
    x_idx = np.asarray(x_idx)
    B = x_idx.shape[0]
    names = trainable_names(spec)
    leaves = {n: (P[n].clone().requires_grad_(want_grads)) for n in names}
    Pw = dict(P); Pw.update(leaves)
    buffers = split_buffers(P)
    dom_col = x_idx[:, spec.domain_idx]
    n_heads = spec.n_tower[-1]
    probs_full = np.full((n_heads, B), np.nan, dtype=np.float32)
    logits_full = np.full((n_heads, B), np.nan, dtype=np.float32)
    total = torch.zeros(1)
    gate_stats = {}
    bag_by_domain = {}
    yt = torch.from_numpy(np.asarray(y, dtype=np.float32).reshape(-1))
    for d in range(spec.n_domain):
        idx = np.nonzero(dom_col == d)[0]
        if idx.size == 0:
            continue
        r = forward(Pw, buffers, spec, x_idx[idx], mode="domain_mask_bagging" if mode != "wo_mask" else mode,
                    mask=masks[d] if masks is not None else None, train=train, sample_ids=idx,
                    drop_seed=drop_seed, want_gate_stats=want_gate_stats)
        buffers = r["buffers"]
        w = 1.0 if domain_weights is None else float(domain_weights[d])
        if mode == "wo_mask":
            bag = bce_mean(r["y"].squeeze(-1), yt[idx])
        else:
            bag = bagging_loss(r["probs"], yt[idx])
        bag_by_domain[d] = float(bag.detach())
        total = total + w * bag
        probs_full[np.ix_(r["heads"], idx)] = r["probs"].detach().numpy()
        logits_full[np.ix_(r["heads"], idx)] = r["cap"]["logits"].detach().numpy()
        gate_stats[d] = r["gate_stats"]
    bag_total = float(total.detach())
    reg = reg_loss(Pw, spec) if with_reg else torch.zeros(1)
    loss = total + reg
    grads = None
    if want_grads:
        gl = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
        grads = {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, gl)}
    return dict(loss=float(loss.detach()), bag=bag_total, reg=float(reg.detach()), probs=probs_full,
                logits=logits_full, grads=grads, buffers=buffers, gate_stats=gate_stats,
                bag_by_domain=bag_by_domain)
