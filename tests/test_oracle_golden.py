"""CPU: the oracle (oracle/aread_oracle.py) against the golden vectors recorded from the real reference.

This is the pin required before the oracle may be trusted as the checker for the HIP path.
"""
import numpy as np
import pytest
import torch

from oracle import aread_oracle as O
from tests import util as U


def test_embedding_golden_bitexact():
    G = U.load_golden("embedding.npz")
    spec = U.spec_full()
    W = O.init_tensor("embedding.embedding_dict.weight", (spec.rows, 32), "emb", 123)
    for method in ("mean", "sum"):
        sp = U.spec_full(method=method)
        x = G[f"{method}/x"]
        np.testing.assert_array_equal(sp.offsets(), G[f"{method}/offsets"])
        bag = O.index_bag(x, sp)
        assert bag.dtype == np.int32
        np.testing.assert_array_equal(bag, G[f"{method}/bag"])                 # bit-exact index bag
        Wg = W.clone().requires_grad_(True)
        out = O.embed_pool(Wg, torch.from_numpy(bag.astype(np.int64)), sp)
        np.testing.assert_array_equal(out.detach().numpy(), G[f"{method}/out"])  # bit-exact pooled rows
        out.backward(torch.from_numpy(G[f"{method}/dout"]))
        np.testing.assert_allclose(Wg.grad.numpy(), G[f"{method}/dtable"], rtol=1e-4, atol=2e-4)  # hot pad row: sum order
    # pad id aliases row 0 of the next field
    pad_rows = G["mean/bag"][G["mean/x"] == spec.field_dims[0]]
    assert pad_rows.size > 0 and (pad_rows == spec.offsets()[1]).all()
    sp = O.Spec(field_dims=[13, 4, 6, 3, 17], embed_dim=32, multi_hot_flag=[False] * 5, method=None, n_domain=4,
                domain_idx=1)
    W = O.init_tensor("embedding.embedding_dict.weight", (sp.rows, 32), "emb", 123)
    out = O.embed_pool(W, torch.from_numpy(O.index_bag(G["flat/x"], sp).astype(np.int64)), sp).flatten(1)
    np.testing.assert_array_equal(out.numpy(), G["flat/out"])


@pytest.mark.parametrize("model", ["full", "tiny"])
@pytest.mark.parametrize("mname", ["ones", "rand", "sparse"])
def test_single_domain_step(model, mname):
    fn, mk, seed = U.GOLDEN_MODELS[model]
    G, spec = U.load_golden(fn), mk()
    P = O.init_params(spec, seed)
    p = f"single_{mname}"
    masks = U.golden_masks(spec, G, mname)
    r = O.step(P, spec, G[f"{p}/x"], G[f"{p}/y"], masks, want_gate_stats=True)
    ref_probs = G[f"{p}/probs"]
    assert (np.isnan(ref_probs) == np.isnan(r["probs"])).all()
    ok = ~np.isnan(ref_probs)
    np.testing.assert_allclose(r["logits"][ok], G[f"{p}/logits"][ok], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r["probs"][ok], ref_probs[ok], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([r["loss"], r["bag"], r["reg"]], G[f"{p}/loss"], rtol=1e-5)
    d = int(G[f"{p}/domain"])
    for l in range(1, spec.n_level):
        np.testing.assert_allclose(r["gate_stats"][d][l].numpy(), G[f"{p}/gate{l}"], rtol=1e-5, atol=1e-7)
    U.check_grads(G, f"{p}/grad", {k: v.numpy() for k, v in r["grads"].items()})
    for k, v in r["buffers"].items():
        np.testing.assert_allclose(v.numpy(), G[f"{p}/buf/{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("model", ["full", "tiny"])
def test_multi_domain_step(model):
    fn, mk, seed = U.GOLDEN_MODELS[model]
    G, spec = U.load_golden(fn), mk()
    P = O.init_params(spec, seed)
    masks = U.golden_masks(spec, G, "rand")
    r = O.step(P, spec, G["multi_rand/x"], G["multi_rand/y"], masks, want_gate_stats=True)
    ref = G["multi_rand/probs"]
    assert (np.isnan(ref) == np.isnan(r["probs"])).all()
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(r["logits"][ok], G["multi_rand/logits"][ok], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose([r["loss"], r["bag"], r["reg"]], G["multi_rand/loss"], rtol=1e-5)
    for d, gs in r["gate_stats"].items():
        for l in range(1, spec.n_level):
            np.testing.assert_allclose(gs[l].numpy(), G[f"multi_rand/gate{l}/d{d}"], rtol=1e-5, atol=1e-7)
    U.check_grads(G, "multi_rand/grad", {k: v.numpy() for k, v in r["grads"].items()})
    for k, v in r["buffers"].items():
        np.testing.assert_allclose(v.numpy(), G[f"multi_rand/buf/{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("model", ["full", "tiny"])
def test_eval_with_mask(model):
    fn, mk, seed = U.GOLDEN_MODELS[model]
    G, spec = U.load_golden(fn), mk()
    P = O.init_params(spec, seed)
    d = int(G["eval_with_mask/domain"])
    masks = U.golden_masks(spec, G, "rand")
    with torch.no_grad():
        r = O.forward(P, O.split_buffers(P), spec, G["eval_with_mask/x"], mode="domain_with_mask", mask=masks[d],
                      train=False)
    np.testing.assert_allclose(r["y"].numpy(), G["eval_with_mask/y"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("model", ["full", "tiny"])
def test_wo_mask_warmup_step(model):
    fn, mk, seed = U.GOLDEN_MODELS[model]
    G, spec = U.load_golden(fn), mk()
    P = O.init_params(spec, seed)
    r = O.step(P, spec, G["wo_mask/x"], G["wo_mask/y"], None, mode="wo_mask", want_gate_stats=True)
    np.testing.assert_allclose([r["loss"], r["bag"], r["reg"]], G["wo_mask/loss"], rtol=1e-5)
    d = int(G["wo_mask/domain"])
    for l in range(1, spec.n_level):
        np.testing.assert_allclose(r["gate_stats"][d][l].numpy(), G[f"wo_mask/gate{l}"], rtol=1e-5, atol=1e-7)
    U.check_grads(G, "wo_mask/grad", {k: v.numpy() for k, v in r["grads"].items()})
    rf = O.forward(P, O.split_buffers(P), spec, G["wo_mask/x"], mode="wo_mask")
    np.testing.assert_allclose(rf["y"].detach().numpy(), G["wo_mask/pred"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("model", ["full", "tiny"])
def test_one_row_skips_batchnorm(model):
    fn, mk, seed = U.GOLDEN_MODELS[model]
    G, spec = U.load_golden(fn), mk()
    P = O.init_params(spec, seed)
    masks = U.golden_masks(spec, G, "rand")
    r = O.step(P, spec, G["one_row/x"], G["one_row/y"], masks)
    ok = ~np.isnan(G["one_row/probs"])
    np.testing.assert_allclose(r["probs"][ok], G["one_row/probs"][ok], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([r["loss"], r["bag"], r["reg"]], G["one_row/loss"], rtol=1e-5)
    U.check_grads(G, "one_row/grad", {k: v.numpy() for k, v in r["grads"].items()})


def test_dropout_hash_statistics_and_determinism():
    keep = O.dropout_keep(2000, O.dropout_site(0, 1, 2), np.arange(4096), 256, 0.2)
    assert keep.shape == (4096, 256)
    assert abs(keep.mean() - 0.8) < 0.005
    keep2 = O.dropout_keep(2000, O.dropout_site(0, 1, 2), np.arange(4096)[::-1].copy(), 256, 0.2)
    assert (keep2 == keep[::-1]).all()                       # keyed by sample id, not by position
    other = O.dropout_keep(2000, O.dropout_site(0, 1, 3), np.arange(4096), 256, 0.2)
    assert abs((other == keep).mean() - (0.8 * 0.8 + 0.2 * 0.2)) < 0.01


def test_random_valid_mask_is_closed():
    spec = U.spec_full()
    rng = np.random.default_rng(0)
    for p in (0.2, 0.5, 0.8):
        for _ in range(20):
            m = O.random_valid_mask(spec, rng, p)
            assert m[-1].any()
            n = spec.n_tower
            for l in range(1, spec.n_level):
                for t in range(n[l - 1]):
                    assert bool(m[l - 1][:, t].any()) == bool(m[l][t, :].any())
            assert O.unpack_mask(spec, O.pack_mask(spec, m))[2].shape == (n[1], n[2])
