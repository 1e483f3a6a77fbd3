"""GPU: one epoch of the training harness (warm-up, HEMP regroups with fast updates + pruning + evaluation, bagging
steps, validation metrics) driving the HIP model, against the trace recorded from the REFERENCE model driven by the
same harness on the CPU (tests/golden/harness.npz)."""
import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu


def test_one_epoch_trace_matches_reference():
    spec = U.spec_full()
    G = U.load_golden("harness.npz")
    model, _ = U.build_model(spec, 123, device="cuda")
    model.device = torch.device("cpu")            # HEMP host logic (mask RNG, thresholds) on the CPU generator
    out = U.run_harness(model, spec, torch.device("cuda"))
    assert list(out["trace_tags"]) == list(G["trace_tags"])          # same schedule: warm-up, regroups, steps
    tags, got, ref = G["trace_tags"], out["trace_vals"], G["trace_vals"]
    # warm-up (Adam lr 1e-3, every tensor on the gradient path) tracks the reference to fp32 round-off
    np.testing.assert_allclose(got[tags == "warmup_loss"], ref[tags == "warmup_loss"], rtol=1e-5)
    # the regroup phase uses Adam(lr=1e-2) fast updates: elements whose gradient is ~eps (data term cancelling the L2
    # term) make g/(|g|+eps) sensitive to the last bits, so trajectories decorrelate slowly -- bounded, not bit-equal
    e0, e1 = got[tags == "mask_edges"], ref[tags == "mask_edges"]
    assert e0[0] == e1[0]                                              # masks picked by the first regroup agree
    assert np.abs(e0 - e1).max() <= 6
    for tag in ("train_loss", "eval_loss"):
        sel = tags == tag
        assert np.abs(got[sel] - ref[sel]).max() < 4e-2, tag
        assert np.abs(got[sel] - ref[sel]).mean() < 1e-2, tag
    assert (out["masks"] != G["masks"]).mean() < 0.15                  # a flipped argmin reroutes later random streams
    np.testing.assert_allclose(out["valid"], G["valid"], atol=3e-2)


def test_grad_none_pattern_and_one_adam_step_match_oracle():
    """Tensors the reference's autograd does not reach keep grad=None (so Adam skips them, weight decay included);
    after one Adam(lr=1e-2) step every tensor is within a fraction of one step of the oracle-driven update."""
    from oracle import aread_oracle as O
    spec = U.spec_full()
    G = U.load_golden("aread_full.npz")
    masks = U.golden_masks(spec, G, "sparse")
    p = "single_sparse"
    d = int(G[f"{p}/domain"]); x = G[f"{p}/x"]; y = G[f"{p}/y"].astype(np.float32)
    model, P = U.build_model(spec, 123)
    model.train()
    tm = [torch.tensor(m, dtype=torch.bool, device="cuda") for m in masks[d]]
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    names = O.trainable_names(spec)
    leaves = {n: torch.nn.Parameter(P[n].clone()) for n in names}
    Pw = dict(P); Pw.update(leaves)
    oopt = torch.optim.Adam(list(leaves.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    crit = torch.nn.BCELoss()
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    preds = model(xt, mode="domain_mask_bagging", domain_i=d, current_mask=tm)
    loss = sum(crit(pr, yt) for pr in preds.unbind(0)) / preds.shape[0] + model.get_regularization_loss()
    model.zero_grad(); loss.backward()
    r = O.forward(Pw, O.split_buffers(P), spec, x, mode="domain_mask_bagging", mask=masks[d], train=True)
    ol = O.bagging_loss(r["probs"], torch.from_numpy(y)) + O.reg_loss(Pw, spec)
    oopt.zero_grad(); ol.backward()
    mine_none = {n: q.grad is None for n, q in model.named_dense_parameters()}
    assert any(mine_none.values())                                    # the sparse mask leaves towers/heads unused
    for n in names:
        if n in mine_none:
            assert mine_none[n] == (leaves[n].grad is None), n
    opt.step(); oopt.step()
    sd = model.state_dict()
    for n in names:
        a, b = sd[n].detach().cpu().numpy(), leaves[n].detach().numpy()
        assert np.abs(a - b).max() < 1e-3, n                          # lr = 1e-2: < 10 % of one step, only on |g|~eps elements
        pre_bn_bias = ".layers." in n and n.endswith("bias") and int(n.split(".layers.")[1].split(".")[0]) % 4 == 0
        if not pre_bn_bias:                                               # d(bias before BN) is pure round-off noise
            assert np.abs(a - b).mean() < 1e-4, n                         # 1 % of one step
        if leaves[n].grad is None:
            assert np.array_equal(a, P[n].numpy()), n                 # untouched


def test_metrics_match_sklearn():
    from sklearn.metrics import log_loss, roc_auc_score
    from aread_amd.harness import _auc, _logloss
    rng = np.random.default_rng(0)
    t = (rng.random(500) < 0.4).astype(np.int64)
    p = np.round(rng.random(500), 2).astype(np.float32)     # ties on purpose
    assert abs(_auc(t, p) - roc_auc_score(t, p)) < 1e-12
    assert abs(_logloss(t, p) - log_loss(t, p)) < 1e-9
