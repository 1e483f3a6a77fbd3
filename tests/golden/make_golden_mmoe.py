#!/usr/bin/env python
"""Record tests/golden/mmoe_aliccp.npz: BASELINE configs[0] -- the reference's MMoE baseline (model/mmoe.py, imported from
/root/reference, use_dcn=False: SURVEY App. B.6/B.14) on the reference's bundled AliCCP sample CSV, tensorised by
aread_amd.data.read_split_data.  Build container only.  Stored: the tensorised inputs (the first 512 training rows: ids,
labels, the harness's per-sample tower column of run.py:499-500), the parameter seed, and the reference's outputs: the
[B, 3] predictions in train and eval mode, the loss of the training step of run.py:496-505 and some of its gradients."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import aread_amd.data as D                     # noqa: E402  (host-side tensorisation, no GPU)
from oracle import mmoe_oracle as MO           # noqa: E402

CSV = "/root/reference/dataset/aliccp/thresh15_ndomain30_modeinterval_random.csv"
GROUPS = [1, 0, 1, 0, 0, 0, 0, 0, 0, 2, 1, 0, 0, 0, 1, 2, 1, 0, 0, 0, 2, 0, 0, 2, 2, 2, 1, 1, 1, 1]   # config.py:72 dcn_3groups_kl
SEED, N = 7, 512


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        from model.mmoe import MMoE                                        # reference
    t = D.read_split_data(CSV, "aliccp")
    X, y = t.splits["train"]
    X, y = X[:N].numpy().astype(np.int32), y[:N].numpy().reshape(-1).astype(np.float32)
    dims = [int(d) for d in t.one_hot_feature_dims]
    group = np.asarray(GROUPS)[X[:, t.domain_idx]].astype(np.int64)
    cfg = types.SimpleNamespace(use_dcn=False, use_atten=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model = MMoE(np.array(dims), 32, t.multi_hot_dict, 3, 4, (256, 128, 64), (64, 32), dropout=0.0, config=cfg)
    shapes = MO.param_shapes(dims)
    sd = model.state_dict()
    assert set(sd) == set(shapes), sorted(set(sd) ^ set(shapes))[:6]
    P = MO.init_params(shapes, SEED)
    model.load_state_dict(P, strict=True)
    xt = torch.from_numpy(X)
    model.train()
    pred = model(xt)
    loss = torch.nn.BCELoss()(pred.gather(1, torch.from_numpy(group).reshape(-1, 1)).squeeze(1), torch.from_numpy(y)) \
        + model.get_regularization_loss(device=torch.device("cpu"))
    model.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().numpy() for n, p in model.named_parameters()}
    rm = model.state_dict()["experts.2.layers.5.running_mean"].numpy().copy()
    model.eval()
    with torch.no_grad():
        pred_eval = model(xt).numpy()
    keep = ["linear.fc.weight", "experts.1.layers.4.weight", "experts.3.layers.9.weight", "gates.2.0.weight", "towers.0.layers.0.weight",
            "towers.1.layers.5.bias", "towers.2.layers.8.weight"]
    tab = grads["embedding.embedding_dict.weight"]
    np.savez_compressed(os.path.join(HERE, "mmoe_aliccp.npz"), x=X, y=y, group=group, dims=np.array(dims), seed=SEED,
                        itemid_idx=t.itemid_idx, domain_idx=t.domain_idx, n_domain=t.n_domain,
                        pred_train=pred.detach().numpy(), pred_eval=pred_eval, loss=np.array([float(loss)]),
                        running_mean_after=rm, table_grad_rows=tab[::997], table_grad_sum=np.array([tab.sum(), np.abs(tab).sum()]),
                        **{"grad/" + k: grads[k] for k in keep})
    print(f"dims {dims[:4]}... n_domain {t.n_domain}; loss {float(loss):.6f}; pred range {pred.min().item():.3f}..{pred.max().item():.3f}")


if __name__ == "__main__":
    main()
