"""TEST INFRASTRUCTURE: CPU restatement of the reference's MMoE baseline (model/mmoe.py:14-73 with BaseModel.tower_forward,
model/layer.py:36-54) for BASELINE configs[0] -- "MMoE on the bundled AliCCP sample, CPU-only plumbing".  Only tests/ may
import this module.  It reuses the primitives of oracle/aread_oracle.py (index bag + pooled lookup, the MLP block, BCE); the
MMoE bottom is the same math as AREAD's (aread.py:150-153).

The reference's multi-tower baselines only run with use_dcn=False (y_logits[B,1] += cn_out[B,D] raises), so
that is what is restated: y[:, t] = sigmoid(Tower_t(sum_k softmax(G_t e)_k Expert_k(e)) + Linear(e)).
Pinned by tests/golden/mmoe_aliccp.npz, recorded from the reference itself (tests/golden/make_golden_mmoe.py)."""
from typing import Dict, Sequence

import numpy as np
import torch

from . import aread_oracle as O


def param_shapes(dims: Sequence[int], embed_dim=32, n_tower=3, n_expert=4, expert_dims=(256, 128, 64), tower_dims=(64, 32)):
    """state_dict of model/mmoe.py (use_dcn=False, use_atten=False): name -> (shape, init kind of aread_oracle.init_tensor)."""
    D = len(dims) * embed_dim
    out = {"embedding.embedding_dict.weight": ((int(sum(dims)), embed_dim), "emb"),
           "linear.fc.weight": ((1, D), "w"), "linear.fc.bias": ((1,), "b")}

    def mlp(prefix, d_in, hidden, output_layer):
        for j, h in enumerate(hidden):
            out[f"{prefix}.layers.{4 * j}.weight"] = ((h, d_in), "w")
            out[f"{prefix}.layers.{4 * j}.bias"] = ((h,), "b")
            bn = f"{prefix}.layers.{4 * j + 1}"
            out[bn + ".weight"], out[bn + ".bias"] = ((h,), "gamma"), ((h,), "beta")
            out[bn + ".running_mean"], out[bn + ".running_var"] = ((h,), "rmean"), ((h,), "rvar")
            out[bn + ".num_batches_tracked"] = ((), "count")
            d_in = h
        if output_layer:
            out[f"{prefix}.layers.{4 * len(hidden)}.weight"] = ((1, d_in), "w")
            out[f"{prefix}.layers.{4 * len(hidden)}.bias"] = ((1,), "b")
    for k in range(n_expert):
        mlp(f"experts.{k}", D, expert_dims, False)
    for t in range(n_tower):
        out[f"gates.{t}.0.weight"], out[f"gates.{t}.0.bias"] = ((n_expert, D), "w"), ((n_expert,), "b")
        mlp(f"towers.{t}", expert_dims[-1], tower_dims, True)
    return out


def init_params(shapes, seed=7) -> Dict[str, torch.Tensor]:
    return {n: O.init_tensor(n, s, k, seed) for n, (s, k) in shapes.items()}


def forward(P, spec: O.Spec, x_idx: np.ndarray, n_tower=3, n_expert=4, expert_dims=(256, 128, 64), tower_dims=(64, 32),
            train=True, buffers=None):
    """mmoe.py:51-73 -> probabilities [B, n_tower]."""
    ctx = O.Ctx(spec=spec, train=train, buffers=buffers if buffers is not None else O.split_buffers(P), drop_seed=0,
                sample_ids=np.arange(x_idx.shape[0], dtype=np.uint32))
    g = torch.from_numpy(O.index_bag(x_idx, spec).astype(np.int64))
    e = O.embed_pool(P["embedding.embedding_dict.weight"], g, spec).flatten(start_dim=1)
    X = torch.stack([O.mlp_stack(P, ctx, f"experts.{k}", e, expert_dims, 0, k) for k in range(n_expert)], dim=1)
    lin = e @ P["linear.fc.weight"].t() + P["linear.fc.bias"]
    ys = []
    for t in range(n_tower):
        pi = torch.softmax(e @ P[f"gates.{t}.0.weight"].t() + P[f"gates.{t}.0.bias"], dim=1)
        u = (pi.unsqueeze(-1) * X).sum(dim=1)
        h = O.mlp_stack(P, ctx, f"towers.{t}", u, tower_dims, 1, t)
        j = 4 * len(tower_dims)
        z = h @ P[f"towers.{t}.layers.{j}.weight"].t() + P[f"towers.{t}.layers.{j}.bias"] + lin      # layer.py:48-53
        ys.append(torch.sigmoid(z))
    return torch.cat(ys, dim=1), ctx.buffers


def reg_loss(P, l2=1e-5):
    """layer.py:96-112 with the groups mmoe.py:44-50 registers: table, linear weights, every '*weight' of experts and towers
    (BatchNorm gammas included: the 'bn' filter never matches 'layers.N')."""
    tot = torch.zeros(1)
    for n, v in P.items():
        if n == "embedding.embedding_dict.weight" or n == "linear.fc.weight" or \
                (n.split(".")[0] in ("experts", "towers") and n.endswith("weight")):
            tot = tot + torch.sum(l2 * torch.square(v))
    return tot


def step(P, spec, x_idx, y, group, **kw):
    """run.py:496-505: loss = BCE(pred.gather(1, group), y) + reg, one backward."""
    leaves = {n: v.clone().requires_grad_(True) for n, v in P.items() if v.is_floating_point() and "running" not in n}
    Pw = dict(P); Pw.update(leaves)
    pred, buffers = forward(Pw, spec, x_idx, train=True, **kw)
    picked = pred.gather(1, torch.from_numpy(np.asarray(group, dtype=np.int64)).reshape(-1, 1)).squeeze(1)
    loss = O.bce_mean(picked, torch.from_numpy(np.asarray(y, dtype=np.float32))) + reg_loss(Pw)
    loss.backward()
    return {"pred": pred.detach().numpy(), "loss": float(loss.detach()), "grads": {n: v.grad for n, v in leaves.items()}, "buffers": buffers}
