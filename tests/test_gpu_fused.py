"""GPU: the fused tower pyramid (csrc/tower_fused.h) against the layer-by-layer launch sequence of the same library, same
parameters, same batch (split-bf16 mode): both are implementations of aread.py:152-153,263-322 + layer.py:203-229 and must
agree to fp32 rounding, in training mode (segment-scoped batch statistics, dropout, gate statistics) and in eval mode.
The layer-by-layer path itself is pinned to the reference goldens by test_gpu_aread.py."""
import numpy as np
import pytest
import torch

from oracle import aread_oracle as O
from tests.util import build_model, spec_full

pytestmark = pytest.mark.gpu


def _batch(spec, rng, B, ragged=True):
    x = np.stack([rng.integers(0, d, B) for d in spec.field_dims]
                 + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
    if ragged:                                   # a big segment (many tiles), small ones, a one-row one, an empty one
        dom = rng.choice(spec.n_domain, B, p=[0.62, 0.25, 0.12, 0.01, 0.0][:spec.n_domain])
        dom[0] = 3
        dom[1:][dom[1:] == 3] = 0
        x[:, spec.domain_idx] = dom
    y = (rng.random(B) < 0.5).astype(np.float32)
    return x, y


def _run(model, x, y, md, fused, names, fused_bwd=None):
    from aread_amd import _lib as L
    L.check(L.lib().aread_debug_set(b"fused_towers", int(fused)))
    L.check(L.lib().aread_debug_set(b"fused_towers_bwd", int(fused if fused_bwd is None else fused_bwd)))
    model.drop_seed = 1234
    bufs = model.make_step_buffers(x.shape[0])
    bufs["ws"].zero_()                          # (the tests read whole buffers: what a step never writes -- H of a tower no mask reaches -- is 0, not stale memory)
    model.bn_stats.copy_(model._stats0); model.bn_nbt.zero_()
    loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md, want_gates=True)
    torch.cuda.synchronize()
    st, gate = model._last
    err = int(st.ws.view(torch.int32)[L.lib().aread_debug_ws_offset(model._handle, st.call.B, st.call.n_seg, b"tf_err")])
    out = {"loss": float(loss), "probs": bufs["probs"].cpu().numpy().copy(), "gdense": bufs["gdense"].cpu().numpy().copy(),
           "gate": gate.cpu().numpy().copy(), "stats": model.bn_stats.cpu().numpy().copy(), "nbt": model.bn_nbt.cpu().numpy().copy(),
           "gtable": model.embedding.embedding_dict.weight.grad.cpu().numpy().copy(), "err": err,
           "rows": int(st.plan.header[2])}                      # rows of live tiles: what lies behind them is never written
    for n, c in names:
        out[n] = model.debug_ws(st, n, c).cpu().numpy().copy()
    return out


@pytest.mark.parametrize("dropout,B", [(0.0, 700), (0.2, 3000), (0.2, 5200)])      # 5200: a 51-tile segment (two-hop statistics merge)
def test_fused_towers_match_layerwise_training_step(dropout, B):
    import aread_amd
    spec = spec_full(dropout=dropout)
    rng = np.random.default_rng(11)
    x, y = _batch(spec, rng, B)
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 123, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    names = [("In0", 3 * 64), ("tw0.0.H", 3 * 64), ("tw0.1.Act", 3 * 32), ("In1", 6 * 32), ("tw1.1.Act", 6 * 16), ("In2", 12 * 16),
             ("tw2.0.H", 12 * 16), ("tw2.1.Act", 12 * 8), ("tw2.1.mean", 12 * 8), ("prob", 12), ("dz", 12)]
    a = _run(model, x, y, md, 0, names)
    b = _run(model, x, y, md, 1, names)
    assert b["err"] == 0, "a segment hand-off of the fused kernel timed out"
    live_seg = np.bincount(x[:, spec.domain_idx], minlength=spec.n_domain) > 0      # statistics rows of empty segments are never written
    for n, _ in names:
        ref, got = (a[n][:a["rows"]], b[n][:a["rows"]]) if "mean" not in n else (a[n][:spec.n_domain][live_seg], b[n][:spec.n_domain][live_seg])
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=n)
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(a["loss"])
    np.testing.assert_allclose(b["probs"], a["probs"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["gate"], a["gate"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["stats"], a["stats"], rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(b["nbt"], a["nbt"])
    for k in ("gdense", "gtable"):
        d = np.abs(b[k] - a[k]).max()
        assert d <= 5e-4 * np.abs(a[k]).max() + 1e-9, (k, d)


@pytest.mark.parametrize("dropout,B,p_active", [(0.0, 700, 0.6), (0.2, 3000, 0.6), (0.2, 1500, 0.3), (0.2, 5200, 0.6)])
def test_fused_tower_backward_matches_layerwise(dropout, B, p_active):
    """csrc/tower_fused_bwd.h against the layer-by-layer backward on the same (fused) forward: every buffer the side stream
    and the expert backward read -- dH of every tower layer, the gate-logit gradients, dX, dlin -- and the final gradients."""
    import aread_amd
    spec = spec_full(dropout=dropout)
    rng = np.random.default_rng(23)
    x, y = _batch(spec, rng, B)
    masks = [O.random_valid_mask(spec, rng, p_active) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 321, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    names = [("tw2.1.dAct", 12 * 8), ("tw2.0.dAct", 12 * 16), ("tw1.1.dAct", 6 * 16), ("tw1.0.dAct", 6 * 32), ("tw0.1.dAct", 3 * 32),
             ("tw0.0.dAct", 3 * 64), ("ex2.dAct", 4 * 64), ("dcn", None)]
    names = [(n, c) for n, c in names if c is not None]
    a = _run(model, x, y, md, 1, names, fused_bwd=0)
    b = _run(model, x, y, md, 1, names, fused_bwd=1)
    assert a["err"] == 0 and b["err"] == 0, "a segment hand-off timed out"
    for n, _ in names:
        ref, got = a[n][:a["rows"]], b[n][:a["rows"]]
        np.testing.assert_allclose(got, ref, rtol=5e-4, atol=2e-5 * max(1e-30, np.abs(ref).max()), err_msg=n)
    assert abs(a["loss"] - b["loss"]) == 0.0
    for k in ("gdense", "gtable"):
        d = np.abs(b[k] - a[k]).max()
        assert d <= 2e-4 * np.abs(a[k]).max() + 1e-12, (k, d)


@pytest.mark.parametrize("geom", [dict(n_tower=(2, 4, 8), n_expert=3, expert_dims=(128, 64, 32), tower_dims=((32, 16), (16, 8), (8, 8)),
                                       n_domain=7),
                                  dict(n_tower=(4, 4, 4), n_expert=5, expert_dims=(64, 64, 16), tower_dims=((16, 16), (16, 16), (16, 8)),
                                       n_domain=3, field_dims=[40, 7, 3, 9, 11, 30, 10])])
def test_fused_towers_other_geometries(geom):
    """other tower / expert widths (8-wide layers, one k-step planes, 5 experts, equal-sized levels): fused forward + backward
    against the layer-by-layer path on the whole step's outputs."""
    import aread_amd
    spec = spec_full(dropout=0.2, **geom)
    rng = np.random.default_rng(41)
    B = 1800
    x = np.stack([rng.integers(0, d, B) for d in spec.field_dims]
                 + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
    y = (rng.random(B) < 0.4).astype(np.float32)
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 99, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    from aread_amd import _lib as L
    a = _run(model, x, y, md, 0, [])
    n0 = (L.lib().aread_debug_get(b"fused_fwd_calls"), L.lib().aread_debug_get(b"fused_bwd_calls"))
    b = _run(model, x, y, md, 1, [])
    n1 = (L.lib().aread_debug_get(b"fused_fwd_calls"), L.lib().aread_debug_get(b"fused_bwd_calls"))
    assert n1[0] == n0[0] + 1 and n1[1] == n0[1] + 1, "the fused kernels did not take this geometry"
    assert b["err"] == 0
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(a["loss"])
    np.testing.assert_allclose(b["probs"], a["probs"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["gate"], a["gate"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["stats"], a["stats"], rtol=1e-4, atol=1e-6)
    for k in ("gdense", "gtable"):
        d = np.abs(b[k] - a[k]).max()
        assert d <= 1e-3 * np.abs(a[k]).max() + 1e-9, (k, d)       # 8-wide towers: two split-bf16 roundings apart


@pytest.mark.parametrize("mode,B", [("wo_mask", 900), ("domain_with_mask", 900), ("domain_with_mask", 4500)])
def test_fused_towers_dropin_autograd_path(mode, B):
    """the drop-in forward()/autograd path (one segment, external dL/dprobs through k_heads_dz, wo_mask = unmasked gates):
    fused forward + backward against the layer-by-layer kernels on predictions and every gradient.
    B = 4500: ONE segment of 71 tiles -- the two-hop statistics merge with more tiles than a wave has lanes, in the tower kernels
    and (expert layers) in k_act_bn_bwd; this is the shape of the reference's per-domain batches."""
    from aread_amd import _lib as L
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(17)
    x, _ = _batch(spec, rng, B, ragged=False)
    x[:, spec.domain_idx] = 2
    y = torch.from_numpy((rng.random(B) < 0.5).astype(np.float32)).cuda()
    masks = [O.random_valid_mask(spec, rng, 0.5) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 123, precision="bf16x3")
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    xd = torch.from_numpy(x).cuda()
    stats0 = model.bn_stats.clone()
    res = []
    n0 = (L.lib().aread_debug_get(b"fused_fwd_calls"), L.lib().aread_debug_get(b"fused_bwd_calls"))
    try:
        for fused in (0, 1):
            L.check(L.lib().aread_debug_set(b"fused_towers", fused))
            L.check(L.lib().aread_debug_set(b"fused_towers_bwd", fused))
            model.bn_stats.copy_(stats0)
            model.train(); model.drop_seed = 5
            model.reset_for_mask_update()
            model.zero_grad()
            pred = model(xd, mode=mode, domain_i=2, memory_gate_value=(mode == "wo_mask"))
            loss = torch.nn.BCELoss()(pred.squeeze(), y) + model.get_regularization_loss(device="cuda")
            loss.backward()
            torch.cuda.synchronize()
            grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
            res.append((pred.detach().cpu().numpy().copy(), float(loss.detach()), grads))
    finally:
        L.check(L.lib().aread_debug_set(b"fused_towers", 1))
        L.check(L.lib().aread_debug_set(b"fused_towers_bwd", 1))
    n1 = (L.lib().aread_debug_get(b"fused_fwd_calls"), L.lib().aread_debug_get(b"fused_bwd_calls"))
    assert n1[0] > n0[0] and n1[1] > n0[1], "the fused kernels did not run on the drop-in path"
    (pa, la, ga), (pb, lb, gb) = res
    np.testing.assert_allclose(pb, pa, rtol=1e-4, atol=1e-6)
    assert abs(la - lb) <= 2e-6 * abs(la)
    assert set(ga) == set(gb)
    for n in ga:
        d = np.abs(gb[n] - ga[n]).max()
        # (4500 rows: the two-hop merge combines the tile statistics in a tree, the layer-by-layer path in four interleaved chains;
        # the small gate-weight gradients see that rounding at 1e-3 of their largest element)
        if B < 1000:
            assert d <= 5e-4 * np.abs(ga[n]).max() + 1e-9, (n, d)
        else:
            rel = float(np.linalg.norm(gb[n] - ga[n]) / (np.linalg.norm(ga[n]) + 1e-30))
            # (a handful of activations sit within rounding of the ReLU threshold and switch sides between the two statistics orders:
            # each one changes its gradient contribution wholesale, which the small tensors see at the 1e-3 level)
            assert (rel <= 2e-2 or d <= 1e-9) and d <= 5e-2 * np.abs(ga[n]).max() + 1e-9, (n, rel, d)


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606])
def test_fused_towers_random_batches(seed):
    """randomised sweep: batch size, domain mix (empty, one-row, sub-tile and many-tile segments), mask density and dropout
    drawn per seed; the fused kernels (in-kernel segment hand-offs through data-tagged granules) against the layer-by-layer
    path on everything the step produces.  Two steps per configuration: the second reuses the hand-off buffers."""
    import aread_amd
    from aread_amd import _lib as L
    rng = np.random.default_rng(seed)
    dropout = float(rng.choice([0.0, 0.2, 0.5]))
    spec = spec_full(dropout=dropout)
    B = int(rng.integers(65, 6000))
    x = np.stack([rng.integers(0, d, B) for d in spec.field_dims]
                 + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
    p = rng.dirichlet(np.ones(spec.n_domain) * float(rng.choice([0.2, 1.0, 5.0])))
    dom = rng.choice(spec.n_domain, B, p=p)
    if seed % 2:
        dom[dom == 4] = 0                                          # an empty segment
        dom[0] = 4                                                 # ... turned into a one-row segment (BatchNorm skipped)
    x[:, spec.domain_idx] = dom
    y = (rng.random(B) < 0.5).astype(np.float32)
    masks = [O.random_valid_mask(spec, rng, float(rng.uniform(0.3, 0.9))) for _ in range(spec.n_domain)]
    model, P = build_model(spec, seed, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    a = _run(model, x, y, md, 0, [])
    n0 = L.lib().aread_debug_get(b"fused_bwd_calls")
    b = _run(model, x, y, md, 1, [])
    b2 = _run(model, x, y, md, 1, [])                              # same inputs, same seed: the fused path is deterministic
    assert L.lib().aread_debug_get(b"fused_bwd_calls") == n0 + 2
    assert b["err"] == 0 and b2["err"] == 0
    for k in ("probs", "gdense", "gtable", "stats"):
        np.testing.assert_array_equal(b2[k], b[k], err_msg=f"run-to-run {k}")
    assert abs(a["loss"] - b["loss"]) <= 2e-6 * abs(a["loss"])
    np.testing.assert_allclose(b["probs"], a["probs"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["gate"], a["gate"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b["stats"], a["stats"], rtol=1e-4, atol=1e-6)
    for k in ("gdense", "gtable"):
        d = np.abs(b[k] - a[k]).max()
        assert d <= 1e-3 * np.abs(a[k]).max() + 1e-9, (k, d)


@pytest.mark.parametrize("B,one_domain", [(2500, False), (5200, False), (4500, True)])
def test_fused_act_bn_backward_matches_two_pass(B, one_domain):
    """k_act_bn_bwd (dropout/ReLU backward + BatchNorm backward of an expert layer in one launch, segment sums handed off in the
    kernel through data-tagged granules) against the two-kernel sequence: identical gradients up to summation order.
    B = 5200: the largest segment has 51 tiles and takes the two-hop merge (owner tile per granule), the others the flat one;
    one_domain: every sample in one domain, 71 tiles (more tiles than a wave has lanes: the owner's lanes take two tiles each)."""
    import aread_amd
    from aread_amd import _lib as L
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(31)
    x, y = _batch(spec, rng, B)
    if one_domain:
        x[:, spec.domain_idx] = 1
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 77, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    names = [("ex0.dAct", 4 * 256), ("ex2.dAct", 4 * 64)]
    res = []
    try:
        for v in (0, 1):
            L.check(L.lib().aread_debug_set(b"fused_act_bn", v))
            res.append(_run(model, x, y, md, 1, names))
    finally:
        L.check(L.lib().aread_debug_set(b"fused_act_bn", 1))
    a, b = res
    assert b["err"] == 0
    for n, _ in names:
        ref, got = a[n][:a["rows"]], b[n][:a["rows"]]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-5 * np.abs(ref).max(), err_msg=n)
    for k in ("gdense", "gtable"):
        d = np.abs(b[k] - a[k]).max()
        assert d <= 1e-4 * np.abs(a[k]).max() + 1e-12, (k, d)


@pytest.mark.parametrize("B", [4500, 8192])
def test_two_hop_statistics_merge_matches_the_flat_merge(B):
    """Segments of more than AREAD_TWO_HOP_NT tiles merge their BatchNorm statistics (forward) and their (sum dyhat, sum dyhat*xhat)
    (backward) in two hops -- an owner tile per column / granule, then every tile reads the finished pair -- in k_tower_fwd,
    k_tower_bwd and k_act_bn_bwd.  Same kernels, same one-domain batch (71 resp. 128 tiles in ONE segment: the reference's
    per-domain batches), threshold 32 against a threshold no segment reaches.
    A different summation order moves the statistics by rounding; downstream a few activations within rounding of the ReLU
    threshold switch sides and, through the segment sums of the BatchNorm backward, move every later gradient at the 1e-3
    level -- so each merge is checked where its inputs are still identical:
      towers : every forward buffer, and the LAST tower layer's dH (the first BatchNorm backward: inputs equal to rounding);
      experts: k_act_bn_bwd's own threshold alone, the towers' merge held fixed -- dH of all three expert layers."""
    import aread_amd
    from aread_amd import _lib as L
    lib = L.lib()
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(23)
    x, y = _batch(spec, rng, B, ragged=False)
    x[:, spec.domain_idx] = 3
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 55, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    names = [("tw0.0.H", 3 * 64), ("tw0.1.Act", 3 * 32), ("tw1.1.Act", 6 * 16), ("tw2.1.Act", 12 * 8), ("dz", 12), ("tw2.1.dAct", 12 * 8),
             ("ex2.dAct", 4 * 64), ("ex1.dAct", 4 * 128), ("ex0.dAct", 4 * 256)]
    FLAT = 1 << 20

    def run(towers, experts):
        L.check(lib.aread_debug_set(b"two_hop_nt", towers))
        L.check(lib.aread_debug_set(b"two_hop_nt_act_bn", experts))
        return _run(model, x, y, md, 1, names)

    def quant(u, v, n):
        ref, got = u[n][:u["rows"]], v[n][:u["rows"]]
        nz = ref != 0
        rel = np.abs(got - ref)[nz] / np.abs(ref)[nz]
        return float(np.quantile(rel, 0.5)), float(np.quantile(rel, 0.99)), float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    try:
        flat, tow, exp = run(FLAT, FLAT), run(32, FLAT), run(FLAT, 32)
    finally:
        L.check(lib.aread_debug_set(b"two_hop_nt", 32)); L.check(lib.aread_debug_set(b"two_hop_nt_act_bn", -1))
    for r in (flat, tow, exp):
        assert r["err"] == 0 and np.isfinite(r["loss"])
    # towers, two hops against one: forward + the first BatchNorm backward
    assert abs(tow["loss"] - flat["loss"]) <= 1e-6 * abs(flat["loss"])
    np.testing.assert_allclose(tow["probs"], flat["probs"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(tow["stats"], flat["stats"], rtol=1e-5, atol=1e-7)
    for n in ("tw0.0.H", "tw0.1.Act", "tw1.1.Act", "tw2.1.Act", "dz", "tw2.1.dAct"):
        q50, q99, l2 = quant(flat, tow, n)
        assert q50 <= 2e-5 and q99 <= 2e-3 and l2 <= 1e-4, (n, q50, q99, l2)
    for k in ("gdense", "gtable"):                       # (everything downstream of the ReLU switches: the 1e-3 level)
        rel = np.linalg.norm(tow[k] - flat[k]) / np.linalg.norm(flat[k])
        assert rel <= 1e-2, (k, rel)
    # experts, two hops against one, identical inputs: more than half of dH's elements have dyhat = 0 and are nothing but
    # -gamma*rstd*(m1 + xhat*m2), i.e. the segment sums themselves -- a tile missed or counted twice would be 1e-2 everywhere
    assert exp["loss"] == flat["loss"]
    for n in ("ex2.dAct", "ex1.dAct", "ex0.dAct"):
        q50, q99, l2 = quant(flat, exp, n)
        assert q50 <= 2e-6 and q99 <= 1e-3 and l2 <= 2e-5, (n, q50, q99, l2)
    for k in ("gdense", "gtable"):
        rel = np.linalg.norm(exp[k] - flat[k]) / np.linalg.norm(flat[k])
        assert rel <= 2e-5, (k, rel)


def test_prepare_early_and_dense_l2_first_give_the_default_steps_bits():
    """aread_prepare ahead of the row plan (model.prepare_early) with the dense L2 terms initialising the gradient buffer
    (aread_call.init_grads / grads_init) is off by default since the end of round 3 (it costs the head of the step more than it
    saves) but stays part of the C ABI: same loss, probabilities and gradients, bit for bit, as the default order."""
    import aread_amd
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(8)
    x, y = _batch(spec, rng, 2200)
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 21, precision="bf16x3")
    model.train()
    model._stats0 = model.bn_stats.clone()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    res = []
    try:
        for early, first in ((False, False), (True, True), (True, False)):
            model.prepare_early, model.l2_dense_first = early, first
            res.append(_run(model, x, y, md, 1, []))
    finally:
        model.prepare_early, model.l2_dense_first = False, True
    a = res[0]
    for b in res[1:]:
        assert b["err"] == 0 and b["loss"] == a["loss"]
        for k in ("probs", "gdense", "gtable", "stats"):
            np.testing.assert_array_equal(b[k], a[k], err_msg=k)


def test_fused_towers_eval_and_wo_mask_forward():
    """eval mode (running statistics, no hand-off) and the unmasked warm-up mode through the drop-in forward()."""
    from aread_amd import _lib as L
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(5)
    x, _ = _batch(spec, rng, 500, ragged=False)
    masks = [O.random_valid_mask(spec, rng, 0.5) for _ in range(spec.n_domain)]
    model, P = build_model(spec, 123, precision="bf16x3")
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    xd = torch.from_numpy(x).cuda()
    res = {}
    stats0 = model.bn_stats.clone()
    for fused in (0, 1):
        L.check(L.lib().aread_debug_set(b"fused_towers", fused))
        model.bn_stats.copy_(stats0)                       # the train-mode forward below moves the running statistics
        model.eval()
        with torch.no_grad():
            e = model(xd, mode="domain_with_mask", domain_i=2).cpu().numpy()
            model.train(); model.drop_seed = 7
            model.reset_for_mask_update()
            w = model(xd, mode="wo_mask", domain_i=1, memory_gate_value=True).cpu().numpy()
        g = torch.stack([torch.stack(v[-1:]).mean(0) for v in model.domain_tower_gate_values[1][1]]).cpu().numpy()
        res[fused] = (e, w, g)
    L.check(L.lib().aread_debug_set(b"fused_towers", 1))
    for a, b in zip(res[0], res[1]):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-6)
