"""GPU: fused optimizer.  aread_adam_step / aread_adam_table_l2 against torch.optim.Adam with the
reference's hyper-parameters (run.py:830-831), and the whole fused training step (aread_amd.FusedAdam) against
train_step + torch.optim.Adam over model.parameters()."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu
HYPER = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)


def _cfg(step, lr=1e-3, wd=1e-8):
    from aread_amd.optim import AdamCfg
    c = AdamCfg()
    c.lr, c.beta1, c.beta2, c.eps, c.weight_decay, c.step = lr, 0.9, 0.99, 1e-8, wd, step
    return c


@pytest.mark.parametrize("n", [1, 1000, 300001])
def test_adam_step_matches_torch(n):
    from aread_amd import _lib as L
    g0 = torch.Generator(device="cuda").manual_seed(n)
    w = torch.randn(n, device="cuda", generator=g0)
    ref = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([ref], **HYPER)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    active = (torch.rand(n, device="cuda", generator=g0) < 0.7).to(torch.uint8) if n > 1 else None
    w_act = w.clone()
    m_act, v_act = torch.zeros_like(w), torch.zeros_like(w)
    for step in range(1, 5):
        g = torch.randn(n, device="cuda", generator=g0) * (10.0 ** float(-step))      # down to 1e-4-sized gradients
        ref.grad = g.clone()
        opt.step()
        c = _cfg(step)
        L.check(L.lib().aread_adam_step(L.ptr(w), L.ptr(g), L.ptr(m), L.ptr(v), n, None, C.byref(c), L.stream()))
        if active is not None:
            L.check(L.lib().aread_adam_step(L.ptr(w_act), L.ptr(g), L.ptr(m_act), L.ptr(v_act), n, L.ptr(active), C.byref(c),
                                            L.stream()))
    torch.testing.assert_close(w, ref.data, rtol=2e-6, atol=2e-7)
    st = opt.state[ref]
    torch.testing.assert_close(m, st["exp_avg"], rtol=1e-5, atol=1e-8)       # m ~ 0.1*g: cancellation near 0
    torch.testing.assert_close(v, st["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    if active is not None:
        on = active.bool()
        assert torch.equal(w_act[on], w[on])
        assert torch.equal(w_act[~on], w.new_tensor(0) + torch.randn(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))[~on])
        assert float(m_act[~on].abs().max()) == 0.0 and float(v_act[~on].abs().max()) == 0.0


def test_adam_table_l2_matches_l2_pass_plus_torch_adam():
    """table kernel == (aread_l2_table dense gradient + scatter) fed to torch Adam; sum(w^2) partials identical to the L2 pass"""
    from aread_amd import _lib as L
    import aread_amd.dist as D
    lib = L.lib()
    rng = np.random.default_rng(3)
    dims, E, B = [5000, 7, 30], 32, 512
    R = sum(dims)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int32)).cuda()
    w = torch.randn(R, E, device="cuda")
    ref = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([ref], **HYPER)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    w2, m2, v2 = w.clone(), m.clone(), v.clone()                     # the same update in two phases
    part1 = torch.empty(lib.aread_l2_partials(), device="cuda")
    part2 = torch.empty(lib.aread_adam_row_partials(), device="cuda")
    ws = torch.zeros(int(lib.aread_route_ws_bytes(R, 1)), dtype=torch.uint8, device="cuda")
    part, part_ref = (torch.empty(lib.aread_l2_partials(), device="cuda") for _ in range(2))
    l2 = 1e-5
    for step in range(1, 4):
        x = torch.from_numpy(np.stack([rng.integers(0, d, B) for d in dims], axis=1).astype(np.int32)).cuda()
        slot = torch.empty_like(x)
        uniq = torch.empty(x.numel(), dtype=torch.int32, device="cuda")
        edges = torch.empty(2, dtype=torch.int32, device="cuda")
        L.check(lib.aread_route_build(L.ptr(x), B, 3, L.ptr(off), R, 1, L.ptr(ws), L.ptr(slot), L.ptr(uniq), L.ptr(edges), 1,
                                      L.stream()))
        n_u = int(edges[1])
        g_rows = torch.zeros(x.numel(), E, device="cuda")
        g_rows[:n_u] = torch.randn(n_u, E, device="cuda") * 1e-3
        # reference: dense gradient = L2 pass + rows
        gd = torch.empty_like(w)
        L.check(lib.aread_l2_table(L.ptr(ref.data), ref.numel(), l2, 1.0, L.ptr(gd), L.ptr(part_ref), L.stream()))
        gd[uniq[:n_u].long()] += g_rows[:n_u]
        ref.grad = gd
        opt.step()
        c = _cfg(step)
        # phase 1 (rows not looked up) + phase 2 (looked-up rows) first: they need the flags that phase 0 clears
        L.check(lib.aread_adam_table_l2(L.ptr(w2), L.ptr(m2), L.ptr(v2), R, E, L.ptr(ws), None, None, None, l2, C.byref(c), 1,
                                        L.ptr(part1), L.stream()))
        flags_between = int(ws[:R].sum())
        L.check(lib.aread_adam_table_l2(L.ptr(w2), L.ptr(m2), L.ptr(v2), R, E, L.ptr(ws), L.ptr(uniq), L.ptr(edges),
                                        L.ptr(g_rows), l2, C.byref(c), 2, L.ptr(part2), L.stream()))
        assert flags_between == n_u and int(ws[:R].sum()) == 0
        L.check(lib.aread_route_build(L.ptr(x), B, 3, L.ptr(off), R, 1, L.ptr(ws), L.ptr(slot), L.ptr(uniq), L.ptr(edges), 1,
                                      L.stream()))
        L.check(lib.aread_adam_table_l2(L.ptr(w), L.ptr(m), L.ptr(v), R, E, L.ptr(ws), L.ptr(uniq), L.ptr(edges), L.ptr(g_rows),
                                        l2, C.byref(c), 0, L.ptr(part), L.stream()))
        torch.cuda.synchronize()
        assert torch.equal(w, w2) and torch.equal(m, m2) and torch.equal(v, v2)
        np.testing.assert_allclose(float(part1.double().sum() + part2.double().sum()), float(part.double().sum()), rtol=1e-6)
        assert int(ws[:R].sum()) == 0                                  # flags consumed
        assert torch.equal(part, part_ref) or step > 1                 # same weights only before the first update
        torch.testing.assert_close(w, ref.data, rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(m, opt.state[ref]["exp_avg"], rtol=1e-5, atol=1e-9)


def test_fused_training_step_matches_train_step_plus_torch_adam():
    import aread_amd
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    masks = U.golden_masks(spec, G, "rand")
    x0 = torch.from_numpy(G["multi_rand/x"]).cuda()
    y0 = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    batches = [(x0, y0)]
    for k in (1, 2):
        perm = torch.from_numpy(np.random.default_rng(k).permutation(x0.shape[0])).cuda()
        xk = x0[perm].clone()
        xk[:, 0] = (xk[:, 0] + 7 * k) % spec.field_dims[0]
        batches.append((xk.contiguous(), (1.0 - y0[perm]).contiguous()))

    def fresh():
        model, _ = U.build_model(spec, seed, dropout=0.2)
        model.train()
        model.drop_seed = 99
        model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk_] for mk_ in masks]
        return model

    a = fresh()
    md = aread_amd.pack_masks(masks, spec.n_domain, a.edge_num, "cuda")
    opt = torch.optim.Adam(a.parameters(), **HYPER)
    bufs = a.make_step_buffers(x0.shape[0])
    ref_losses = []
    for x, y in batches:
        a.zero_grad(set_to_none=True)
        ref_losses.append(float(a.train_step(x, y, bufs, masks_dev=md)))
        opt.step()
    b = fresh()
    fused = aread_amd.FusedAdam(b, x0.shape[0], **HYPER)
    losses = [float(fused.step(x, y, md)) for x, y in batches]
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses[0], ref_losses[0], rtol=1e-6)
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)          # later steps: see the dense tolerance below
    wa, wb = a.embedding.embedding_dict.weight.data, b.embedding.embedding_dict.weight.data
    # every element moved by ~lr per step; both paths must agree far inside that.  Isolated entries of hot rows whose
    # ~B/10 contributions cancel to |g| ~ 1e-9 are renormalised by Adam, so the two summation orders (L2 term first vs
    # last) can differ there by a fraction of one lr step: bound the count of such entries, not only the maximum
    dw = (wa - wb).abs()
    assert float(dw.max()) <= 3e-4 and float(dw.mean()) <= 1e-6 and int((dw > 1e-5).sum()) <= max(20, dw.numel() // 10000)
    assert float((wb - U.build_model(spec, seed)[0].embedding.embedding_dict.weight.data).abs().max()) > 2e-3
    # dense: the bias of a Linear that feeds BatchNorm has a zero true gradient; what is computed is rounding noise
    # (~1e-10..1e-7) that Adam renormalises to up to +-lr per step and that decorrelates between any two
    # implementations after the first update.  Those biases cannot influence the forward (BN removes them).
    da, db = a.dense.data, b.dense.data
    dd = (da - db).abs()
    assert float(dd.mean()) <= 2e-7
    # ... and the few hot-row entries above feed back into the next forward: isolated dense entries with near-zero
    # gradients then differ by a fraction of one lr step too (both paths are bit-reproducible run to run,
    # tests/devtools/debug_fused_adam.py); everything else agrees to 5e-5 of ~3e-3 of movement
    assert int((dd > 5e-5).sum()) <= dd.numel() // 2000 and float(dd.max()) <= 2 * HYPER["lr"] * len(batches)
    # tensors the masks never reach keep their initial values and zero optimizer state in both
    present = fused._present_for(b.domain_mask)
    for on, (name, kind, off, shape, l2) in zip(present, b._ptensors):
        if not on:
            n = int(np.prod(shape)) if shape else 1
            assert float(fused.m_dense[off:off + n].abs().max()) == 0.0


def test_fused_step_accepts_a_smaller_last_batch():
    """buffers sized for B = 512 drive a 300-row batch (the last batch of an epoch) to exactly the same update as buffers
    sized for 300, and a full-size step still works afterwards"""
    import aread_amd
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    masks = U.golden_masks(spec, G, "rand")
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    n_small = min(300, x.shape[0] - 1)
    xs, ys = x[:n_small].contiguous(), y[:n_small].contiguous()
    rep = (512 + x.shape[0] - 1) // x.shape[0]
    xb, yb = x.repeat(rep, 1)[:512].contiguous(), y.repeat(rep)[:512].contiguous()

    def run(cap, then_full):
        model, _ = U.build_model(spec, seed, dropout=0.2)
        model.train()
        model.drop_seed = 5
        model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk_] for mk_ in masks]
        md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
        fused = aread_amd.FusedAdam(model, cap, **HYPER)
        loss = float(fused.step(xs, ys, md))
        state = (model.embedding.embedding_dict.weight.data.clone(), model.dense.data.clone())
        if then_full:
            assert np.isfinite(float(fused.step(xb, yb, md)))
        torch.cuda.synchronize()
        return loss, state

    loss_big, (w_big, d_big) = run(512, then_full=True)
    loss_small, (w_small, d_small) = run(n_small, then_full=False)
    assert loss_big == loss_small
    assert torch.equal(w_big, w_small) and torch.equal(d_big, d_small)


def test_dropin_adam_matches_torch_adam_on_the_reference_step_loop():
    """aread_amd.Adam(model) in place of torch.optim.Adam(model.parameters()) in the reference's per-domain step closure
    (run.py:668-681): three steps with different masks (so some towers keep grad=None on some steps and their step counts
    differ), table and dense parameters compared."""
    import aread_amd
    from tests.test_gpu_aread import tmask
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    crit = torch.nn.BCELoss()
    hyper = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)

    def run(make_opt):
        model, _ = U.build_model(spec, seed, dropout=0.0)
        model.train()
        opt = make_opt(model)
        losses = []
        for mname in ("sparse", "rand", "ones"):
            p = f"single_{mname}"
            d = int(G[f"{p}/domain"])
            masks = U.golden_masks(spec, G, mname)
            x = torch.from_numpy(G[f"{p}/x"]).cuda()
            y = torch.from_numpy(G[f"{p}/y"].astype(np.float32)).cuda()
            preds = model(x, mode="domain_mask_bagging", domain_i=d, current_mask=tmask(masks[d]))
            loss = sum(crit(pr, y) for pr in preds.unbind(dim=0)) / preds.shape[0] + model.get_regularization_loss(device="cuda")
            model.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return losses, model.embedding.embedding_dict.weight.data.clone(), model.dense.data.clone(), model, opt

    l_ref, w_ref, d_ref, _, _ = run(lambda m: torch.optim.Adam(m.parameters(), **hyper))
    l_new, w_new, d_new, model, opt = run(lambda m: aread_amd.Adam(m, **hyper))
    np.testing.assert_allclose(l_new[0], l_ref[0], rtol=1e-6)
    np.testing.assert_allclose(l_new, l_ref, rtol=1e-4)
    dw = (w_new - w_ref).abs()
    assert float(dw.max()) <= 3e-4 and float(dw.mean()) <= 1e-6
    dd = (d_new - d_ref).abs()
    assert float(dd.mean()) <= 2e-7 and int((dd > 5e-5).sum()) <= dd.numel() // 2000
    assert len(set(opt._t_dense.tolist())) > 1                    # the masks really produced different step counts
    # tensors that never got a gradient kept their initial value in both runs
    init = U.build_model(spec, seed, dropout=0.0)[0].dense.data
    never = opt._t_dense == 0
    for on, (name, kind, off, shape, l2) in zip(never, model._ptensors):
        if on:
            n = int(np.prod(shape)) if shape else 1
            assert torch.equal(d_new[off:off + n], init[off:off + n]) and torch.equal(d_ref[off:off + n], init[off:off + n]), name


def test_dropin_adam_checkpoint_resume_is_bit_identical():
    """aread_amd.Adam.state_dict() must carry the flat moment buffers and step counts.  Two steps, checkpoint
    (through torch.save / torch.load like run.py:459-484), a fresh model + optimizer loaded from it, a third step: bit for bit
    the third step of the uninterrupted run."""
    import io
    import aread_amd
    from tests.test_gpu_aread import tmask
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    crit = torch.nn.BCELoss()
    hyper = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    steps = ("sparse", "rand", "ones")

    def one(model, opt, mname):
        p = f"single_{mname}"
        d = int(G[f"{p}/domain"])
        masks = U.golden_masks(spec, G, mname)
        x = torch.from_numpy(G[f"{p}/x"]).cuda()
        y = torch.from_numpy(G[f"{p}/y"].astype(np.float32)).cuda()
        preds = model(x, mode="domain_mask_bagging", domain_i=d, current_mask=tmask(masks[d]))
        loss = sum(crit(pr, y) for pr in preds.unbind(dim=0)) / preds.shape[0] + model.get_regularization_loss(device="cuda")
        model.zero_grad()
        loss.backward()
        opt.step()
        return float(loss.detach())

    a, _ = U.build_model(spec, seed, dropout=0.0); a.train()
    oa = aread_amd.Adam(a, **hyper)
    for s_ in steps[:2]:
        one(a, oa, s_)
    buf = io.BytesIO()
    torch.save({"state_dict": a.state_dict(), "optimizer": oa.state_dict()}, buf)
    sd = oa.state_dict()
    assert sd["state"]["flat"]["m_table"] is not None and int(sd["state"]["flat"]["t_table"]) == 2
    assert len(set(sd["state"]["flat"]["t_dense"].tolist())) > 1          # tensors skipped on some steps keep their own count
    last_a = one(a, oa, steps[2])
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    b, _ = U.build_model(spec, seed + 1, dropout=0.0); b.train()           # different initial values: everything must come from the checkpoint
    b.load_state_dict(ck["state_dict"], strict=True)
    ob = aread_amd.Adam(b, lr=5.0)                                          # wrong hyper-parameters too
    ob.load_state_dict(ck["optimizer"])
    last_b = one(b, ob, steps[2])
    torch.cuda.synchronize()
    assert last_a == last_b
    assert torch.equal(a.dense.data, b.dense.data)
    assert torch.equal(a.embedding.embedding_dict.weight.data, b.embedding.embedding_dict.weight.data)
    assert torch.equal(oa._m_dense, ob._m_dense) and torch.equal(oa._v_table, ob._v_table)
    import pytest
    with pytest.raises(ValueError):
        ob.load_state_dict(torch.optim.Adam(b.parameters()).state_dict())


def test_fused_adam_follows_an_in_place_mask_change():
    """HEMP rewrites model.domain_mask[d] IN PLACE; the cached 'which tensors get a gradient' set must follow.
    Two fused steps with a regroup between them (one domain's mask replaced by one that reaches other towers) against
    train_step + torch.optim.Adam, which sees the change through grad=None."""
    import aread_amd
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    sparse, ones = U.golden_masks(spec, G, "sparse"), U.golden_masks(spec, G, "ones")
    reach = [sum(int(np.asarray(l).any(axis=0).sum()) for l in mk_[:-1]) for mk_ in sparse]
    sparse = [sparse[int(np.argmin(reach))]] * spec.n_domain               # every domain on the one mask that reaches the fewest towers
    x = torch.from_numpy(G["multi_rand/x"]).cuda()
    y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    dev = lambda mk_: [torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk_]

    def fresh():
        model, _ = U.build_model(spec, seed, dropout=0.0)
        model.train()
        model.domain_mask = [dev(m) for m in sparse]
        return model

    def regroup(model):
        v = model.mask_version
        for d in range(spec.n_domain):
            model.domain_mask[d] = dev(ones[d])                            # what update_all_mask does (aread.py:330-341)
        assert model.mask_version > v
        return aread_amd.pack_masks(ones, spec.n_domain, model.edge_num, "cuda")

    a = fresh()
    opt = torch.optim.Adam(a.parameters(), **HYPER)
    bufs = a.make_step_buffers(x.shape[0])
    md = aread_amd.pack_masks(sparse, spec.n_domain, a.edge_num, "cuda")
    for k in range(2):
        a.zero_grad(set_to_none=True)
        a.train_step(x, y, bufs, masks_dev=md)
        opt.step()
        if k == 0:
            md = regroup(a)
    b = fresh()
    fused = aread_amd.FusedAdam(b, x.shape[0], **HYPER)
    md = aread_amd.pack_masks(sparse, spec.n_domain, b.edge_num, "cuda")
    p0 = fused._present_for(b.domain_mask).copy()
    fused.step(x, y, md)
    md = regroup(b)
    fused.step(x, y, md)
    p1 = fused._present_for(b.domain_mask)
    torch.cuda.synchronize()
    assert p1.sum() > p0.sum()                                             # the all-ones masks reach towers the sparse ones do not
    newly = p1 & ~p0
    assert newly.any()
    for on, t, (name, kind, off, shape, l2) in zip(newly, fused.t_dense, b._ptensors):
        if on:
            assert t == 1, name                                            # first update on the second step, like torch
            n = int(np.prod(shape)) if shape else 1
            assert float((a.dense.data[off:off + n] - b.dense.data[off:off + n]).abs().max()) <= 5e-5, name
    dd = (a.dense.data - b.dense.data).abs()
    assert float(dd.mean()) <= 2e-7 and int((dd > 5e-5).sum()) <= dd.numel() // 2000


def test_gradient_bookkeeping_matches_a_full_scan_of_the_parameters():
    """The module keeps a record of which dense parameters hold a gradient it assigned (so that a backward does not read
    ~300 `.grad` attributes three times).  The same scripted sequence -- model.zero_grad(), an optimizer's zero_grad() behind
    the module's back, zero_grad(set_to_none=False), two backwards without zeroing in between (accumulation), masks that
    change which towers take part -- must leave the same None-pattern and the same gradient values as the mode that reads
    every attribute (AREAD._GRAD_SCAN)."""
    import aread_amd
    from tests.test_gpu_aread import tmask
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    crit = torch.nn.BCELoss()

    def loss_of(model, mname, reg=True):
        p = f"single_{mname}"
        d = int(G[f"{p}/domain"])
        masks = U.golden_masks(spec, G, mname)
        x = torch.from_numpy(G[f"{p}/x"]).cuda()
        y = torch.from_numpy(G[f"{p}/y"].astype(np.float32)).cuda()
        preds = model(x, mode="domain_mask_bagging", domain_i=d, current_mask=tmask(masks[d]))
        bce = sum(crit(pr, y) for pr in preds.unbind(dim=0)) / preds.shape[0]
        return bce + model.get_regularization_loss(device="cuda") if reg else bce

    def script(scan):
        aread_amd.AREAD._GRAD_SCAN = scan
        try:
            model, _ = U.build_model(spec, seed, dropout=0.0)
            model.train()
            topt = torch.optim.SGD(model.parameters(), lr=0.0)
            snaps = []

            def snap():
                torch.cuda.synchronize()
                tg = model.embedding.embedding_dict.weight.grad
                snaps.append([None if p.grad is None else p.grad.detach().cpu().numpy().copy() for p in model.dense_params]
                             + [None if tg is None else tg.detach().cpu().numpy().copy()])
            model.zero_grad(); loss_of(model, "sparse").backward(); snap()              # 0: fresh, few towers
            model.zero_grad(); loss_of(model, "ones").backward(); snap()                # 1: fresh, every tower
            topt.zero_grad(set_to_none=True); loss_of(model, "rand").backward(); snap() # 2: zeroed behind the module's back
            loss_of(model, "sparse").backward(); snap()                                 # 3: accumulation onto 2
            model.zero_grad(set_to_none=False); loss_of(model, "sparse").backward(); snap()   # 4: zeroed in place
            model.zero_grad(); loss_of(model, "sparse").backward(); loss_of(model, "ones").backward(); snap()   # 5: two in a row
            model.dense_params[0].grad = None                                           # an external edit of one gradient ...
            topt.zero_grad(set_to_none=True); loss_of(model, "rand").backward(); snap() # 6: ... followed by an external zero_grad
            # the regulariser alone offers its table gradient to a forward node that never comes; the next pass must not see it
            model.zero_grad(); model.get_regularization_loss(device="cuda").backward(); snap()            # 7
            model.zero_grad(); loss_of(model, "rand", reg=False).backward(); snap()                        # 8: no regulariser in the loss
            assert model.__dict__.get("_gtab_pending") is None
            (0.5 * loss_of(model, "ones")).backward(); snap()                                              # 9: scaled loss, accumulated onto 8
            return snaps
        finally:
            aread_amd.AREAD._GRAD_SCAN = False

    ref, got = script(True), script(False)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert [x is None for x in a] == [x is None for x in b], f"snapshot {k}: different None pattern"
        for i, (x, z) in enumerate(zip(a, b)):
            if x is not None:
                np.testing.assert_array_equal(z, x, err_msg=f"snapshot {k}, parameter {i}")
    assert any(x is None for x in ref[0]) and sum(x is None for x in ref[0]) > sum(x is None for x in ref[1])
