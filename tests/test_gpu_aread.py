"""GPU parity of the whole path (embedding + dense model, forward and backward) through the C ABI,
against the golden vectors recorded from the reference and against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import aread_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)      # north_star: logits within 1e-4 relative


def tmask(mask, dev="cuda"):
    return [torch.tensor(np.asarray(m), dtype=torch.bool, device=dev) for m in mask]


def all_grads(model):
    g = U.dense_grads(model)
    g["embedding.embedding_dict.weight"] = model.embedding.embedding_dict.weight.grad.detach().cpu().numpy()
    return g


def logits_of(p):
    return np.log(p) - np.log1p(-p)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_state_dict_roundtrip(which):
    fn, mk, seed = U.GOLDEN_MODELS[which]
    spec = mk()
    model, P = U.build_model(spec, seed)
    sd = model.state_dict()
    assert set(sd.keys()) == set(P.keys())
    for k, v in P.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert torch.equal(sd[k].cpu(), v), k


@pytest.mark.parametrize("which", ["full", "tiny"])
@pytest.mark.parametrize("mname", ["ones", "rand", "sparse"])
def test_single_domain_bagging_step_autograd(which, mname):
    """The drop-in path: model(X, 'domain_mask_bagging') + BCELoss per head + reg, loss.backward() (run.py:668-680)."""
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.train()
    model.reset_for_mask_update() if hasattr(model, "reset_for_mask_update") else None
    p = f"single_{mname}"
    d = int(G[f"{p}/domain"])
    masks = U.golden_masks(spec, G, mname)
    x = torch.from_numpy(G[f"{p}/x"]).cuda()
    y = torch.from_numpy(G[f"{p}/y"].astype(np.float32)).cuda()
    preds = model(x, mode="domain_mask_bagging", domain_i=d, current_mask=tmask(masks[d]), tmp_memory_gate_value=True)
    crit = torch.nn.BCELoss()
    bag = sum(crit(pr, y) for pr in preds.unbind(dim=0)) / preds.shape[0]
    reg = model.get_regularization_loss(device="cuda")
    loss = bag + reg
    model.zero_grad()
    loss.backward()
    ref = G[f"{p}/probs"]
    act = ~np.isnan(ref[:, 0])
    assert preds.shape[0] == act.sum()
    got = preds.detach().cpu().numpy()
    np.testing.assert_allclose(logits_of(got), G[f"{p}/logits"][act], **TOL)
    np.testing.assert_allclose(got, ref[act], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([float(loss), float(bag), float(reg)], G[f"{p}/loss"], rtol=2e-5)
    for l in range(1, spec.n_level):
        g = torch.stack(model.tmp_tower_gate_values[l], dim=1).cpu().numpy()
        np.testing.assert_allclose(g, G[f"{p}/gate{l}"], rtol=1e-4, atol=1e-6)
    U.check_grads(G, f"{p}/grad", all_grads(model), rtol=5e-4, atol_scale=5e-5)
    sd = model.state_dict()
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            np.testing.assert_allclose(sd[k].cpu().numpy(), G[f"{p}/buf/{k}"], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_multi_domain_fused_step(which):
    """The fused 'N-domain batch' step (ragged: one empty domain, one single-row domain) vs the reference's
    per-domain calls."""
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.train()
    masks = U.golden_masks(spec, G, "rand")
    model.domain_mask = [tmask(m) for m in masks]
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    bufs = model.make_step_buffers(x.shape[0], multi_domain=True)
    loss = model.train_step(x, y, bufs, want_gates=True)
    torch.cuda.synchronize()
    ref = G["multi_rand/probs"]
    ok = ~np.isnan(ref)
    got = bufs["probs"].cpu().numpy()
    assert (got[~ok] == 0).all()
    np.testing.assert_allclose(logits_of(got[ok]), G["multi_rand/logits"][ok], **TOL)
    np.testing.assert_allclose([float(loss), float(bufs["loss"][0]), float(bufs["reg"][0])], G["multi_rand/loss"], rtol=2e-5)
    st, gate = model._last
    gate = gate.cpu().numpy()
    for d in range(spec.n_domain):
        if f"multi_rand/gate1/d{d}" not in G:
            continue
        off = 0
        for l in range(1, spec.n_level):
            n = spec.n_tower[l] * spec.n_tower[l - 1]
            g = gate[d, off:off + n].reshape(spec.n_tower[l], spec.n_tower[l - 1]).T
            np.testing.assert_allclose(g, G[f"multi_rand/gate{l}/d{d}"], rtol=1e-4, atol=1e-6)
            off += n
    U.check_grads(G, "multi_rand/grad", all_grads(model), rtol=5e-4, atol_scale=5e-5)
    sd = model.state_dict()
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            np.testing.assert_allclose(sd[k].cpu().numpy(), G[f"multi_rand/buf/{k}"], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_eval_domain_with_mask(which):
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.eval()
    d = int(G["eval_with_mask/domain"])
    masks = U.golden_masks(spec, G, "rand")
    with torch.no_grad():
        yv = model(torch.from_numpy(G["eval_with_mask/x"]).cuda(), mode="domain_with_mask", domain_i=d,
                   current_mask=tmask(masks[d]))
    np.testing.assert_allclose(yv.cpu().numpy(), G["eval_with_mask/y"], rtol=1e-5, atol=1e-6)
    sd = model.state_dict()
    P = O.init_params(spec, seed)
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            assert torch.equal(sd[k].cpu(), P[k]), k            # eval never touches the running statistics


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_eval_inference_epilogue_matches_unfolded_eval(precision):
    """eval under torch.no_grad() (aread_call.inference: BatchNorm with running statistics + ReLU inside the expert GEMMs'
    epilogue, no k_bn_act, H not kept) == eval with autograd enabled (the layer-by-layer expert path that keeps H for a
    backward), on a many-row batch and on a ONE-row batch (BatchNorm skipped, layer.py:226)."""
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed, precision=precision)
    model.eval()
    model.bn_stats.uniform_(0.5, 1.5)                         # non-trivial running statistics
    d = int(G["eval_with_mask/domain"])
    mask = tmask(U.golden_masks(spec, G, "rand")[d])
    x = torch.from_numpy(G["eval_with_mask/x"]).cuda()
    for xb in (x, x[:1].contiguous()):
        with torch.no_grad():
            a = model(xb, mode="domain_with_mask", domain_i=d, current_mask=mask)
        b = model(xb, mode="domain_with_mask", domain_i=d, current_mask=mask).detach()
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_wo_mask_warmup_step(which):
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.train()
    model.reset_for_mask_update()
    d = int(G["wo_mask/domain"])
    x = torch.from_numpy(G["wo_mask/x"]).cuda()
    y = torch.from_numpy(G["wo_mask/y"].astype(np.float32)).cuda()
    pred = model(x, mode="wo_mask", domain_i=d, memory_gate_value=True)
    assert tuple(pred.shape) == (x.shape[0], 1)
    loss = torch.nn.BCELoss()(pred.squeeze(), y)
    reg = model.get_regularization_loss(device="cuda")
    model.zero_grad()
    (loss + reg).backward()
    np.testing.assert_allclose(pred.detach().cpu().numpy(), G["wo_mask/pred"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([float(loss + reg), float(loss), float(reg)], G["wo_mask/loss"], rtol=2e-5)
    for l in range(1, spec.n_level):
        g = torch.stack([model.domain_tower_gate_values[d][l][t][0] for t in range(spec.n_tower[l])], dim=1).cpu().numpy()
        np.testing.assert_allclose(g, G[f"wo_mask/gate{l}"], rtol=1e-4, atol=1e-6)
    U.check_grads(G, "wo_mask/grad", all_grads(model), rtol=5e-4, atol_scale=5e-5)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_one_row_call_skips_batchnorm(which):
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.train()
    masks = U.golden_masks(spec, G, "rand")
    model.domain_mask = [tmask(m) for m in masks]
    x = torch.from_numpy(G["one_row/x"]).cuda()
    y = torch.from_numpy(G["one_row/y"].astype(np.float32)).cuda()
    bufs = model.make_step_buffers(1, multi_domain=True)
    loss = model.train_step(x, y, bufs)
    ok = ~np.isnan(G["one_row/probs"])
    np.testing.assert_allclose(bufs["probs"].cpu().numpy()[ok], G["one_row/probs"][ok], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(loss), G["one_row/loss"][0], rtol=2e-5)
    U.check_grads(G, "one_row/grad", all_grads(model), rtol=5e-4, atol_scale=5e-5)


def test_dropout_matches_oracle_hash():
    """p = 0.2 in train mode: the kernels and the oracle share the counter-based keep mask bit for bit."""
    spec = U.spec_full(dropout=0.2)
    seed = 123
    G = U.load_golden("aread_full.npz")
    model, P = U.build_model(spec, seed)
    model.train()
    model.drop_seed = 777
    masks = U.golden_masks(spec, G, "rand")
    model.domain_mask = [tmask(m) for m in masks]
    x, y = G["multi_rand/x"], G["multi_rand/y"]
    bufs = model.make_step_buffers(x.shape[0], multi_domain=True)
    loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y.astype(np.float32)).cuda(), bufs)
    r = O.step(P, spec, x, y, masks, drop_seed=777)
    ok = ~np.isnan(r["probs"])
    np.testing.assert_allclose(logits_of(bufs["probs"].cpu().numpy()[ok]), r["logits"][ok], **TOL)
    np.testing.assert_allclose(float(loss), r["loss"], rtol=2e-5)
    got = all_grads(model)
    for k, v in r["grads"].items():
        ref = v.numpy()
        if k not in got:                      # dead parameters (attention branch, final_gate): no gradient
            assert not ref.any(), k
            continue
        np.testing.assert_allclose(got[k], ref, rtol=1e-3, atol=5e-5 * max(np.abs(ref).max(), 1e-4), err_msg=k)


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_split_bf16_mode_within_tolerance(which):
    """precision='bf16x3' (split-bf16 forward/dgrad GEMMs): logits within the north star's 1e-4, gradients within 2e-3."""
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed, precision="bf16x3")
    model.train()
    masks = U.golden_masks(spec, G, "rand")
    model.domain_mask = [tmask(m) for m in masks]
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    bufs = model.make_step_buffers(x.shape[0], multi_domain=True)
    n0 = _fused_calls()
    loss = model.train_step(x, y, bufs)
    if which == "full":        # the configuration the fused tower kernels take (tiny: in_dim % 8 != 0 -> layer-by-layer path)
        assert _fused_calls() == (n0[0] + 1, n0[1] + 1), "k_tower_fwd / k_tower_bwd did not produce these numbers"
    ref = G["multi_rand/probs"]
    ok = ~np.isnan(ref)
    got = bufs["probs"].cpu().numpy()
    refl = G["multi_rand/logits"][ok]
    assert np.abs(logits_of(got[ok]) - refl).max() <= 1e-4 * max(np.abs(refl).max(), 1.0)
    np.testing.assert_allclose(float(loss), G["multi_rand/loss"][0], rtol=5e-5)
    # gradients: BatchNorm backward on few-row segments and ReLU sign flips amplify the ~1e-5 product error of the split
    U.check_grads(G, "multi_rand/grad", all_grads(model), rtol=1e-2, atol_scale=1e-2)
    # ... and norm-wise per tensor (measured: median 5e-5, 90th percentile 1e-3, worst tower tensor 5e-3; the exact-fp32
    # mode sits at 3e-6 on the same fixtures, test_fp32_mode_gradients_norm_wise below)
    e = _rel_l2(U.rel_l2_vs_golden(G, "multi_rand/grad", all_grads(model)))
    print(f"[bf16x3 vs golden, {which}] gradient rel-L2 per tensor: median {np.median(e):.2e} p90 {np.quantile(e, 0.9):.2e} max {e.max():.2e}")
    # bounds = 2x the measured triple (r3d: full 4.7e-5 / 1.0e-3 / 5.1e-3, tiny 2.8e-5 / 5.8e-5 / 1.3e-4)
    assert np.median(e) <= 1e-4 and np.quantile(e, 0.9) <= 2e-3 and e.max() <= 1e-2, (np.median(e), np.quantile(e, 0.9), e.max())


def _fused_calls():
    """(k_tower_fwd launches, k_tower_bwd launches) so far: the oracle comparisons on the timed configuration assert that the
    fused kernels -- not the layer-by-layer fallback -- produced the numbers they check"""
    from aread_amd import _lib as L
    return (L.lib().aread_debug_get(b"fused_fwd_calls"), L.lib().aread_debug_get(b"fused_bwd_calls"))


def _pre_bn_bias(name):
    """Linear bias in front of a BatchNorm (layers.{0,4,8}.bias): the true gradient is exactly zero, every implementation
    (the reference included) returns ~1e-10 of rounding noise there -- a relative error is meaningless"""
    k = name.split(".")
    return k[-1] == "bias" and "layers" in k and int(k[k.index("layers") + 1]) % 4 == 0


def _rel_l2(d):
    return np.array([v for n, v in d.items() if not _pre_bn_bias(n)])


@pytest.mark.parametrize("which", ["full", "tiny"])
def test_fp32_mode_gradients_norm_wise(which):
    """exact-fp32 mode: every parameter gradient within 1e-4 relative L2 of the reference's (measured 3e-6 at worst)."""
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed)
    model.train()
    model.domain_mask = [tmask(m) for m in U.golden_masks(spec, G, "rand")]
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    bufs = model.make_step_buffers(x.shape[0], multi_domain=True)
    model.train_step(x, y, bufs)
    e = _rel_l2(U.rel_l2_vs_golden(G, "multi_rand/grad", all_grads(model)))
    assert len(e) > 100 and e.max() <= 1e-4, e.max()


def test_baseline_size_proportional_gradients_vs_fp64_oracle():
    """The workload bench.py runs (Amazon dims, B = 8192, domains ~ config.py:60-61) with the domains of fewer than 8 rows
    removed (11 samples: BatchNorm over 1-3 rows is ill-conditioned in any precision): EVERY parameter gradient, norm-wise,
    against an fp64 run of the oracle.  The fp32 oracle itself is 3e-6 (median) / 5e-4 (90th percentile) / 2e-3 (worst) away
    from fp64 -- the per-domain BatchNorm backward amplifies rounding by ~1e3 -- so the bounds are: exact-fp32 mode the
    oracle's own distance, split-bf16 mode that times the ~1e3 larger product error.  Measured (r3d, printed by the test):
    f32 median 3.2e-6 / p90 3.5e-4 / max 1.0e-3, split-bf16 2.4e-3 / 7.4e-3 / 4.1e-2; the bounds are 2x those."""
    import aread_amd
    from tools import synth
    spec = O.amazon_spec(dropout=0.0)
    rng = np.random.default_rng(2000)
    masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
    x, y = synth.amazon_batch(spec, rng, 8192, domain="proportional")
    cnt = np.bincount(x[:, spec.domain_idx], minlength=spec.n_domain)
    keep = cnt[x[:, spec.domain_idx]] >= 8
    assert keep.sum() > 0.99 * len(keep)
    x, y = x[keep], y[keep]
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    P = O.init_params(spec, 123)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    r64 = O.step(P64, spec, x, y.astype(np.float64), masks)
    for precision, med, p90, worst in (("f32", 1e-5, 7e-4, 2e-3), ("bf16x3", 5e-3, 1.5e-2, 8e-2)):     # <= 2x the measured triples below
        model, _ = U.build_model(spec, 123, precision=precision)
        model.train()
        md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
        model.domain_mask = [tmask(m) for m in masks]
        bufs = model.make_step_buffers(x.shape[0])
        n0 = _fused_calls()
        loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md)
        if precision == "bf16x3":
            assert _fused_calls() == (n0[0] + 1, n0[1] + 1), "the timed configuration must run k_tower_fwd / k_tower_bwd"
        assert abs(float(loss) - r64["loss"]) <= 5e-5 * abs(r64["loss"]), precision
        g = all_grads(model)
        e = {}
        for n, ref in r64["grads"].items():
            ref = ref.numpy()
            if n in g and np.linalg.norm(ref) > 0:
                e[n] = float(np.linalg.norm(g[n].astype(np.float64) - ref) / np.linalg.norm(ref))
        v = _rel_l2(e)
        assert len(v) > 200
        print(f"[B=8192 proportional vs fp64 oracle, {precision}] gradient rel-L2 per tensor: median {np.median(v):.2e} "
              f"p90 {np.quantile(v, 0.9):.2e} max {v.max():.2e} ({max((n for n in e if not _pre_bn_bias(n)), key=e.get)})")
        assert np.median(v) <= med and np.quantile(v, 0.9) <= p90 and v.max() <= worst, (precision, np.median(v), np.quantile(v, 0.9), v.max())
        assert e["embedding.embedding_dict.weight"] <= (1e-4 if precision == "f32" else 5e-3), (precision, e["embedding.embedding_dict.weight"])
        del model, bufs
        torch.cuda.empty_cache()


def test_dp_path_single_rank_matches_fused_step():
    """aread_amd.dist.DataParallelStep on RCCL with one rank == the fused single-GPU step, bit for bit."""
    import torch.distributed as dist
    import aread_amd
    from aread_amd.dist import DataParallelStep
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    masks = U.golden_masks(spec, G, "rand")
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    res = []
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1"); os_env.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        for use_dp in (False, True, "overlap"):
            model, _ = U.build_model(spec, seed)
            model.train()
            md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
            if use_dp:
                dp = DataParallelStep(model, x.shape[0], force_overlap=(use_dp == "overlap"))
                loss = dp.step(x, y, md); bufs = dp.bufs
            else:
                bufs = model.make_step_buffers(x.shape[0])
                loss = model.train_step(x, y, bufs, masks_dev=md, set_grads=False)
            res.append((float(loss), bufs["gdense"].clone(), bufs["gtable"].clone()))
    finally:
        dist.destroy_process_group()
    for r in res[1:]:
        assert res[0][0] == r[0]
        assert torch.equal(res[0][1], r[1]) and torch.equal(res[0][2], r[2])


@pytest.mark.parametrize("B", [1, 100, 700])
@pytest.mark.parametrize("output_layer", [False, True])
def test_multilayer_perceptron_vs_torch(B, output_layer):
    """Drop-in MultiLayerPerceptron (layer.py:203-229) against the same block built from torch.nn on the CPU (fp32)."""
    import aread_amd
    torch.manual_seed(3)
    dims, in_dim = (48, 24, 12), 40
    ref = torch.nn.ModuleList()
    d = in_dim
    for h in dims:
        ref += [torch.nn.Linear(d, h), torch.nn.BatchNorm1d(h), torch.nn.ReLU(), torch.nn.Dropout(0.0)]
        d = h
    if output_layer:
        ref.append(torch.nn.Linear(d, 1))
    for m in ref:
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    sd = {f"layers.{k}": v for k, v in ref.state_dict().items()}
    mlp = aread_amd.MultiLayerPerceptron(in_dim, dims, 0.0, output_layer=output_layer)
    mlp.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    mlp = mlp.cuda()
    assert set(mlp.state_dict().keys()) == set(sd.keys())

    def ref_fwd(x):
        for m in ref:
            if isinstance(m, torch.nn.BatchNorm1d) and x.shape[0] == 1:
                continue
            x = m(x)
        return x

    for train in (True, False):
        ref.train(train); mlp.train(train)
        x = torch.randn(B, in_dim)
        xr = x.clone().requires_grad_(True)
        xg = x.clone().cuda().requires_grad_(True)
        yr = ref_fwd(xr)
        yg = mlp(xg)
        assert tuple(yg.shape) == tuple(yr.shape)
        np.testing.assert_allclose(yg.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-4, atol=2e-5)
        w = torch.randn_like(yr)
        ref.zero_grad(); mlp.zero_grad()
        (yr * w).sum().backward()
        (yg * w.cuda()).sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=2e-5)
        got = {n: v for n, v in zip([t[0] for t in mlp._tensors if t[1] == 0],
                                    [mlp.dense.grad[t[2]:t[2] + int(np.prod(t[3]))].view(t[3]).cpu().numpy()
                                     for t in mlp._tensors if t[1] == 0])}
        for k, p in ref.named_parameters():
            g = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)   # B == 1: BN skipped
            idx = int(k.split(".")[0])
            pre_bn_bias = k.endswith("bias") and idx % 4 == 0 and idx < 4 * len(dims) and train and B > 1
            atol = 1e-3 if pre_bn_bias else max(2e-5, 3e-5 * np.abs(g).max())     # d(bias before BN) is exactly 0: fp noise
            np.testing.assert_allclose(got[f"layers.{k}"], g, rtol=2e-3, atol=atol, err_msg=k)
    sd2 = mlp.state_dict()
    for k, v in ref.state_dict().items():
        if "running" in k or "num_batches" in k:
            np.testing.assert_allclose(sd2[f"layers.{k}"].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("domain_dist", ["proportional", "uniform"])
def test_baseline_size_step_vs_oracle_both_precisions(domain_dist):
    """BASELINE config (Amazon-like dims, 1.39 M-row table, 25 domains, B=8192): fused step vs the CPU oracle.
    Logits inside the north star's tolerance |d| <= 1e-4 * max(|ref|, 1) in the exact-fp32 mode AND in the
    split-bf16 mode bench.py runs; loss to 5e-5; gradients on the uniform-domain batch.

    With domains drawn proportionally to config.py:60-61, a B=8192 batch contains domains of 1-3 rows.  BatchNorm
    over 2-3 rows is ill-conditioned in ANY fp32 implementation (x_hat = +-1 for two rows; rstd up to 1/sqrt(eps)
    per layer): the fp32 oracle itself differs from an fp64 oracle by 100 % on those gradients
    (tests/devtools/debug_small_segments.py).  Logits are therefore asserted on domains with >= 8 rows (> 99 % of the
    samples) and gradients on the uniform batch."""
    import aread_amd
    from tools import synth
    spec = O.amazon_spec(dropout=0.0)
    rng = np.random.default_rng(2000)
    masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
    x, y = synth.amazon_batch(spec, rng, 8192, domain=domain_dist)
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    P = O.init_params(spec, 123)
    r = O.step(P, spec, x, y, masks, want_grads=(domain_dist == "uniform"))
    # yardstick for the ill-conditioned few-row domains: an fp64 run of the oracle (forward only)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    r64 = O.step(P64, spec, x, y.astype(np.float64), masks, want_grads=False)
    ok = ~np.isnan(r["probs"])
    refl = r["logits"][ok]
    cnt = np.bincount(x[:, spec.domain_idx], minlength=spec.n_domain)
    big = np.broadcast_to((cnt[x[:, spec.domain_idx]] >= 8)[None, :], ok.shape)
    sel = ok & big
    assert sel.sum() > 0.99 * ok.sum()
    # gradient spot checks at this size are sanity bounds (25 per-domain BatchNorm backward passes with cancellation:
    # the fp32 oracle itself is ~1 % of max away from fp64); strict gradient parity is pinned by the golden tests
    for precision, gtol, kfac in (("f32", 3e-2, 10.0), ("bf16x3", 6e-2, 150.0)):
        model, _ = U.build_model(spec, 123, precision=precision)
        model.train()
        md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
        model.domain_mask = [tmask(m) for m in masks]
        bufs = model.make_step_buffers(8192)
        n0 = _fused_calls()
        loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md)
        if precision == "bf16x3":
            assert _fused_calls() == (n0[0] + 1, n0[1] + 1), "the timed configuration must run k_tower_fwd / k_tower_bwd"
        got = bufs["probs"].cpu().numpy()
        assert (got[~ok] == 0).all()
        dl = np.abs(logits_of(got.astype(np.float64)) - r["logits"].astype(np.float64))
        assert dl[sel].max() <= 1e-4 * max(np.abs(refl).max(), 1.0), (precision, dl[sel].max())
        assert np.isfinite(got[ok]).all()
        small = ok & ~big
        if small.any():
            # domains of 1-7 rows (BatchNorm over 2-3 rows: x_hat = +-1, rstd up to 1/sqrt(eps)): the fp64 oracle is the
            # yardstick -- the HIP probabilities may be kfac times as far from it as the fp32 oracle is (f32: the same
            # arithmetic in another summation order; split-bf16: 4e-6 per product against fp32's 6e-8, ~70x)
            e_ref = np.abs(r["probs"][small].astype(np.float64) - r64["probs"][small].astype(np.float64))
            e_hip = np.abs(got[small].astype(np.float64) - r64["probs"][small].astype(np.float64))
            q = lambda v: (float(np.median(v)), float(np.quantile(v, 0.9)), float(v.max()))
            print(f"[B=8192 {domain_dist}, {precision}] few-row domains ({int(small.sum())} probabilities): |hip - fp64| median/p90/max "
                  f"{q(e_hip)}, |oracle32 - fp64| {q(e_ref)}; logits of the other domains: max |d| {dl[sel].max():.2e}")
            # measured (r3e): f32 median 3.0e-7 / p90 1.0e-5 / max 1.8e-2 against the fp32 oracle's own 1.8e-7 / 1.4e-6 / 4.0e-4;
            # split-bf16 median 7.0e-6 / p90 1.5e-2 / max 4.7e-2.  The typical (median) error obeys the yardstick; the tail does
            # not in ANY implementation: two nearly equal rows give x_hat = 158 * (x1 - x2) per layer (eps = 1e-5), nine
            # BatchNorm layers deep, so a 1e-7 difference in summation order decides single probabilities -- bounded absolutely,
            # at half the previous slack
            assert q(e_hip)[0] <= kfac * q(e_ref)[0] + 1e-6, (precision, q(e_hip), q(e_ref))
            if precision == "f32":
                assert q(e_hip)[1] <= kfac * q(e_ref)[1] + 1e-6, (precision, q(e_hip), q(e_ref))
            assert q(e_hip)[2] <= 0.1, (precision, q(e_hip))
        bags = bufs["loss"].cpu().numpy()[1:]
        for d_, b_ in r["bag_by_domain"].items():
            if cnt[d_] >= 8:
                assert abs(bags[d_] - b_) <= 1e-4 * b_, (precision, d_)       # per-domain bagging loss
        assert abs(float(bufs["reg"][0]) - r["reg"]) <= 2e-5 * r["reg"]
        if domain_dist == "uniform":
            assert abs(float(loss) - r["loss"]) <= 5e-5 * abs(r["loss"]), precision
            g = all_grads(model)
            for name in ("linear.fc.weight", "cn.w.1.weight", "mmoe_experts.2.layers.4.weight", "mmoe_gates.1.0.weight",
                         "towers.1.3.layers.0.weight", "tower_gates.1.7.0.weight", "towers_linear.5.weight"):
                ref = r["grads"][name].numpy()
                assert np.abs(g[name] - ref).max() <= gtol * np.abs(ref).max(), (precision, name)
            bag = np.unique(O.index_bag(x, spec))[:4000]
            gt = model.embedding.embedding_dict.weight.grad[torch.from_numpy(bag.astype(np.int64)).cuda()].cpu().numpy()
            reft = r["grads"]["embedding.embedding_dict.weight"].numpy()[bag]
            assert np.abs(gt - reft).max() <= gtol * np.abs(reft).max(), precision
        del model, bufs
        torch.cuda.empty_cache()


@pytest.mark.parametrize("precision,ltol,gfac", [("f32", 2e-5, 3.0), ("bf16x3", 1e-4, 150.0)])
def test_aliccp_layout_step_vs_oracle(precision, ltol, gfac):
    """BASELINE config 5 layout (AliCCP-like: 23 one-hot fields, no history pooling, D = 736, 30 domains, the domain
    id in column 10), table dims capped at 600 rows per field so the CPU oracle finishes in seconds: fused
    multi-domain step vs the oracle -- logits, loss, reg, and the gradient of every tensor.

    Gradients: 30 per-domain BatchNorm backward passes over ~100-row segments lose digits in ANY fp32 implementation,
    so the yardstick is the oracle itself: the HIP gradient must be as close to an fp64 run of the oracle as the
    fp32 oracle is (x gfac; split-bf16 GEMMs carry 4e-6 per product instead of fp32's 6e-8, i.e. ~70x the unit
    roundoff, and the same conditioning amplifies both)."""
    import aread_amd
    full = O.aliccp_spec()
    dims = [min(d, 600) for d in full.field_dims]
    spec = O.aliccp_spec(field_dims=dims, dropout=0.0)
    assert spec.n_domain == 30 and spec.d == 23 * 32 and spec.f_in == 23
    rng = np.random.default_rng(77)
    B = 3000
    x = np.stack([rng.integers(0, d, B) for d in dims], axis=1).astype(np.int32)
    x[:, spec.domain_idx] = rng.integers(0, spec.n_domain, B)
    y = (rng.random(B) < 0.2).astype(np.float32)
    masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    P = O.init_params(spec, 11)
    r = O.step(P, spec, x, y, masks, want_grads=True)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    r64 = O.step(P64, spec, x, y.astype(np.float64), masks, want_grads=True)
    model, _ = U.build_model(spec, 11, precision=precision)
    model.train()
    model.domain_mask = [tmask(m) for m in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    bufs = model.make_step_buffers(B)
    loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md)
    got = bufs["probs"].cpu().numpy()
    ok = ~np.isnan(r["probs"])
    assert (got[~ok] == 0).all()
    refl = r["logits"][ok]
    assert np.abs(logits_of(got[ok]) - refl).max() <= ltol * max(np.abs(refl).max(), 1.0)
    assert abs(float(loss) - r["loss"]) <= 5e-5 * abs(r["loss"])
    assert abs(float(bufs["reg"][0]) - r["reg"]) <= 2e-5 * r["reg"]
    g = all_grads(model)
    checked, num, den = 0, 0.0, 0.0
    for name, ref in r["grads"].items():
        ref, ref64 = ref.numpy(), r64["grads"][name].numpy()
        scale = np.abs(ref64).max()
        if name not in g or scale < 1e-7:
            continue
        e_ref = np.abs(ref - ref64).max()
        e_hip = np.abs(g[name] - ref64).max()
        if precision == "f32":
            assert e_hip <= gfac * e_ref + 2e-5 * scale, (name, e_hip, e_ref, scale)
        num += float(((g[name] - ref64) ** 2).sum())
        den += float((ref64 ** 2).sum())
        checked += 1
    # split-bf16: a 4e-6 perturbation also flips ReLU units sitting at zero (one of ~100 rows of a segment moves single
    # entries by percents), so the bound is norm-wise over all gradients together
    assert (num / den) ** 0.5 <= (1e-3 if precision == "f32" else 3e-2), (num / den) ** 0.5
    assert checked > 200


def test_multilayer_perceptron_split_bf16_backward():
    """MultiLayerPerceptron(precision='bf16x3'): forward, input gradient (the dgrad uses transposed weight copies made in
    the forward) and weight gradients against the fp32-mode module."""
    import aread_amd
    torch.manual_seed(5)
    dims, in_dim, B = (64, 32, 16), 96, 500
    a = aread_amd.MultiLayerPerceptron(in_dim, dims, 0.0, output_layer=True).cuda()
    b = aread_amd.MultiLayerPerceptron(in_dim, dims, 0.0, output_layer=True, precision="bf16x3").cuda()
    b.load_state_dict(a.state_dict(), strict=True)
    a.train(); b.train()
    x = torch.randn(B, in_dim, device="cuda")
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = a(xa), b(xb)
    np.testing.assert_allclose(yb.detach().cpu().numpy(), ya.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    w = torch.randn_like(ya)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    ga, gb = xa.grad.cpu().numpy(), xb.grad.cpu().numpy()
    assert np.abs(gb - ga).max() <= 2e-3 * np.abs(ga).max()
    da, db = a.dense.grad.cpu().numpy(), b.dense.grad.cpu().numpy()
    assert np.linalg.norm(db - da) <= 2e-3 * np.linalg.norm(da)


def test_aliccp_full_size_layout_properties():
    """BASELINE configs[4] at FULL size: the AliCCP layout (23 one-hot fields, 1 140 414 table rows, D = 736, 30 domains drawn
    ~ config.py:62-64) at B = 8192, split-bf16 mode, one step on an 'original' batch (positive rate 0.043) and one on an
    'augmented' batch (0.13).  Too large for a full oracle step in the test budget, so:
      * eval-mode probabilities of a 512-row sub-batch against the oracle (eval rows are independent: SURVEY 8e);
      * size-independent properties of the training step: loss = sum_d bag_d + reg, reg against an fp64 sum, every gradient
        finite, table-gradient rows that no sample looked up equal 2*l2*w exactly, looked-up rows differ from it, and the
        step is bitwise reproducible (sorted segmented reduction, fixed-order partial sums)."""
    import aread_amd
    from tools import synth
    spec = O.aliccp_spec(dropout=0.2)
    rng = np.random.default_rng(5)
    masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
    model, P = U.build_model(spec, 11, precision="bf16x3")
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    model.domain_mask = [tmask(m) for m in masks]
    x, y = synth.generic_batch(spec, rng, 8192, domain_p=synth.ALICCP_DOMAIN_SIZE, pos_rate=0.043)
    xa, ya = synth.generic_batch(spec, rng, 8192, domain_p=synth.ALICCP_DOMAIN_SIZE, pos_rate=0.13)
    # ---- eval forward of the whole batch (every sample under its own domain's mask) vs the oracle on 512 of its rows
    model.eval()
    with torch.no_grad():
        got, _ = model(torch.from_numpy(x).cuda(), mode="with_mask")
    order = np.argsort(x[:, spec.domain_idx], kind="stable")
    sub = order[::16]                                               # 512 rows across all domains
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    probs = O.step(P, spec, x[sub], y[sub], masks, train=False, want_grads=False, with_reg=False)["probs"]
    mean_ref = np.nanmean(probs, axis=0)                            # mean over the active heads (aread.py:233)
    # a domain with ONE row in the sub-batch skips BatchNorm in the oracle's per-domain call (layer.py:226), also in eval
    # mode, while its many rows in the full batch do not: compare the others
    dsub = x[sub, spec.domain_idx]
    multi = np.bincount(dsub, minlength=spec.n_domain)[dsub] >= 2
    assert multi.sum() > 480
    np.testing.assert_allclose(got.cpu().numpy()[::16][multi], mean_ref[multi], rtol=1e-4, atol=2e-6)
    # ---- training steps
    model.train()
    bufs = model.make_step_buffers(8192)
    w = model.embedding.embedding_dict.weight.detach()
    reg64 = spec.l2_embedding * float((w.double() ** 2).sum())
    outs = []
    for xb, yb in ((x, y), (xa, ya), (xa, ya)):
        model.drop_seed = 99
        loss = model.train_step(torch.from_numpy(xb).cuda(), torch.from_numpy(yb).cuda(), bufs, masks_dev=md)
        torch.cuda.synchronize()
        bag = bufs["loss"].cpu().numpy()
        assert np.isfinite(float(loss)) and abs(float(loss) - (bag[0] + float(bufs["reg"][0]))) <= 1e-6 * abs(float(loss))
        assert abs(bag[0] - bag[1:1 + spec.n_domain].sum()) <= 1e-5 * abs(bag[0])
        assert float(bufs["reg"][0]) >= reg64 * (1 - 1e-5) and float(bufs["reg"][0]) <= reg64 * 1.02     # table term + the small dense terms
        gt = model.embedding.embedding_dict.weight.grad
        assert torch.isfinite(gt).all() and torch.isfinite(bufs["gdense"]).all()
        touched = torch.zeros(gt.shape[0], dtype=torch.bool, device="cuda")
        touched[torch.from_numpy(np.unique(O.index_bag(xb, spec)).astype(np.int64)).cuda()] = True
        l2g = 2.0 * spec.l2_embedding * w
        assert torch.equal(gt[~touched], l2g[~touched])
        assert ((gt[touched] - l2g[touched]).abs().amax(dim=1) > 0).float().mean() > 0.99
        outs.append((float(loss), gt.clone(), bufs["gdense"].clone()))
    assert outs[1][0] == outs[2][0] and torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[1][2], outs[2][2])   # reproducible
    assert outs[0][0] != outs[1][0]
