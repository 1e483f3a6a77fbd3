"""CPU: BASELINE configs[0] -- the MMoE baseline on the reference's bundled AliCCP sample ("plumbing, no GPU").  The oracle's
restatement (oracle/mmoe_oracle.py: model/mmoe.py:14-73 + layer.py:36-54 with use_dcn=False) against outputs recorded from
the reference itself on inputs tensorised by aread_amd.data (tests/golden/make_golden_mmoe.py)."""
import os

import numpy as np
import torch

from oracle import aread_oracle as O
from oracle import mmoe_oracle as MO

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mmoe_aliccp.npz"))


def _spec():
    dims = [int(d) for d in G["dims"]]
    return O.Spec(field_dims=dims, embed_dim=32, multi_hot_flag=[False] * len(dims), itemid_idx=int(G["itemid_idx"]), method=None,
                  n_domain=int(G["n_domain"]), domain_idx=int(G["domain_idx"])), dims


def test_geometry_of_the_bundled_sample():
    spec, dims = _spec()
    assert len(dims) == 23 and spec.n_domain == 30 and spec.domain_idx == 10 and spec.itemid_idx == 9      # SURVEY 8d, run.py:57-59
    assert G["x"].shape == (512, 23) and G["x"].dtype == np.int32 and (G["x"] < np.array(dims)[None, :]).all()
    assert set(np.unique(G["group"])) <= {0, 1, 2}


def test_mmoe_train_step_and_eval_match_reference():
    spec, dims = _spec()
    torch.set_num_threads(4)
    P = MO.init_params(MO.param_shapes(dims), int(G["seed"]))
    r = MO.step(P, spec, G["x"], G["y"], G["group"])
    np.testing.assert_allclose(r["pred"], G["pred_train"], rtol=2e-5, atol=2e-6)
    assert r["pred"].shape == (512, 3)
    assert abs(r["loss"] - float(G["loss"][0])) <= 2e-6 * abs(float(G["loss"][0]))
    for k in G.files:
        if k.startswith("grad/"):
            ref = G[k]
            np.testing.assert_allclose(r["grads"][k[5:]].numpy(), ref, rtol=2e-3, atol=2e-6 * max(np.abs(ref).max(), 1e-6), err_msg=k)
    tab = r["grads"]["embedding.embedding_dict.weight"].numpy()
    np.testing.assert_allclose(tab[::997], G["table_grad_rows"], rtol=1e-3, atol=1e-9)
    assert abs(tab.sum() - G["table_grad_sum"][0]) <= 1e-3 * G["table_grad_sum"][1]
    np.testing.assert_allclose(r["buffers"]["experts.2.layers.5.running_mean"].numpy(), G["running_mean_after"], rtol=1e-5, atol=1e-6)
    # eval mode on the statistics the training step left behind
    P2 = dict(P); P2.update(r["buffers"])
    pe, _ = MO.forward(P2, spec, G["x"], train=False)
    np.testing.assert_allclose(pe.numpy(), G["pred_eval"], rtol=2e-5, atol=2e-6)
