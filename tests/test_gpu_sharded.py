"""GPU: row-sharded embedding table step (aread_amd.dist.ShardedTableStep, SURVEY 8e north-star partitioning).

 * one rank, collectives forced through RCCL: pooled embedding and loss bit-identical to the fused
   single-GPU step, shard gradient == table gradient, owned dense chunk == dense gradient (incl. L2);
 * two ranks sharing this box's one GPU (gloo rehearsal of the very same code, staged through host memory):
   every rank's shard gradient / dense chunk equals the sum over ranks of the unsharded per-rank steps,
   and one ZeRO-1 Adam step leaves identical dense parameters on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_batch(spec, G, rank):
    """rank 0 takes the golden multi-domain batch as is; other ranks a seeded row permutation with flipped labels"""
    x, y = G["multi_rand/x"], G["multi_rand/y"].astype(np.float32)
    if rank:
        perm = np.random.default_rng(rank).permutation(x.shape[0])
        x, y = x[perm], 1.0 - y[perm]
        x = x.copy()
        x[:, 0] = (x[:, 0] + 3 * rank) % spec.field_dims[0]           # different items -> different unique rows
    return torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(np.ascontiguousarray(y)).cuda()


def _unsharded(spec, seed, masks, x, y, drop_seed):
    import aread_amd
    model, _ = U.build_model(spec, seed, dropout=0.2)
    model.train()
    model.drop_seed = drop_seed
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    bufs = model.make_step_buffers(x.shape[0])
    model.step_local(x, y, bufs, masks_dev=md, with_dense_l2=False, presort=True)
    model.embedding.reduce_sorted(x, bufs["de"], bufs["gtable"])
    model.step_finish(bufs)
    torch.cuda.synchronize()
    return model, bufs, md


def test_sharded_single_rank_matches_fused_step():
    import torch.distributed as dist
    import aread_amd
    import aread_amd.dist as D
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    masks = U.golden_masks(spec, G, "rand")
    x, y = _rank_batch(spec, G, 0)
    ref_model, ref, md = _unsharded(spec, seed, masks, x, y, 77)
    ref_model.add_dense_l2(ref)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    D.FORCE_COLLECTIVES = True
    try:
        model, _ = U.build_model(spec, seed, dropout=0.2)
        model.train()
        model.drop_seed = 77
        sh = D.ShardedTableStep(model, x.shape[0])
        total = sh.step(x, y, md)
        torch.cuda.synchronize()
        assert torch.equal(sh.bufs["e"], ref["e"])                       # same rows, same pooling order
        assert torch.equal(sh.bufs["loss"], ref["loss"])
        np.testing.assert_allclose(float(total), float(ref["loss"][0] + ref["reg"][0]), rtol=1e-6)
        gt = ref["gtable"]
        assert sh.gshard.shape == gt.shape
        scale = float(gt.abs().max())
        assert float((sh.gshard - gt).abs().max()) <= 2e-6 * scale      # two-level reduction order
        n = model.dense.numel()
        gd = ref["gdense"]
        assert float((sh.gchunk[:n] - gd).abs().max()) <= 1e-6 * float(gd.abs().max())
        assert torch.equal(sh.full_table(), model.embedding.embedding_dict.weight.data)
        # route prefetch: the routing of the next batch made on its own stream during this step gives the same step;
        # a prefetched route is only used for the batch it was made for (x2 here), never for another one
        g1, e1 = sh.gshard.clone(), sh.bufs["e"].clone()
        x2 = torch.flip(x, dims=[0]).contiguous()
        y2 = torch.flip(y, dims=[0]).contiguous()
        model.drop_seed = 77
        sh.step(x, y, md, next_x=x2)                                     # prefetches x2's routing
        torch.cuda.synchronize()
        assert torch.equal(sh.gshard, g1) and torch.equal(sh.bufs["e"], e1)
        assert sh._pref is not None
        sh.step(x, y, md)                                                # x again: the prefetched route must be ignored
        torch.cuda.synchronize()
        assert sh._pref is None and torch.equal(sh.gshard, g1) and torch.equal(sh.bufs["e"], e1)
        sh.step(x2, y2, md)                                              # reference for x2, routed inline
        torch.cuda.synchronize()
        g2, e2 = sh.gshard.clone(), sh.bufs["e"].clone()
        sh.prefetch_route(x2)
        xs = x2.clone()                                                  # a staging copy, matched through x_key
        sh.step(xs, y2, md, x_key=x2.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(sh.bufs["e"], e2) and torch.equal(sh.gshard, g2)
        # fixed-capacity exchange (no host read: constant split sizes, padded unique-row buffer): the same pooled embedding and
        # shard gradient; a capacity below the batch's unique count is reported when the NEXT step is issued
        cap = sh.calibrate_capacity(x)
        assert cap % 64 == 0 and cap >= int(sh._keep[0].n_unique)
        model.drop_seed = 77
        sh.step(x, y, md)
        torch.cuda.synchronize()
        assert sh._keep[0].n_unique == cap and not bool(sh._keep[0].overflow)
        assert torch.equal(sh.bufs["e"], e1)
        assert float((sh.gshard - g1).abs().max()) <= 1e-6 * float(g1.abs().max())     # (padded slots reorder the owner-side reduction)
        sh.capacity = 64
        sh.step(x, y, md)
        with pytest.raises(RuntimeError, match="exceeded capacity"):
            sh.step(x, y, md)
        sh.capacity, sh._ovf_prev = None, None
    finally:
        D.FORCE_COLLECTIVES = False
        dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import aread_amd.dist as D
        fn, mk, seed = U.GOLDEN_MODELS["full"]
        G, spec = U.load_golden(fn), mk()
        masks = U.golden_masks(spec, G, "rand")
        # expected: every rank's unsharded step, summed (each process recomputes all of them: deterministic kernels)
        gt_sum, gd_sum, losses = None, None, []
        for r in range(world):
            xr, yr = _rank_batch(spec, G, r)
            mr, br, md = _unsharded(spec, seed, masks, xr, yr, 77 + r)
            sc = br["gtable"] - 2.0 * mr.l2_reg_embedding * mr.embedding.embedding_dict.weight.data   # scatter part
            gt_sum = sc if gt_sum is None else gt_sum + sc
            gd_sum = br["gdense"].clone() if gd_sum is None else gd_sum + br["gdense"]
            losses.append(float(br["loss"][0]))
            if r == world - 1:
                reg_model, reg_bufs = mr, br
        reg_bufs["gdense"].copy_(gd_sum)
        reg_model.add_dense_l2(reg_bufs)                                   # dense L2 once, on the summed gradient
        gd_exp = reg_bufs["gdense"]
        W = reg_model.embedding.embedding_dict.weight.data
        gt_exp = gt_sum + 2.0 * reg_model.l2_reg_embedding * W
        reg_exp = float(reg_bufs["reg"][0])

        x, y = _rank_batch(spec, G, rank)
        model, _ = U.build_model(spec, seed, dropout=0.2)
        model.train()
        model.drop_seed = 77 + rank
        sh = D.ShardedTableStep(model, x.shape[0])
        total = sh.step(x, y, md)
        torch.cuda.synchronize()
        ok = {}
        ok["loss"] = abs(float(total) - (losses[rank] + reg_exp)) <= 2e-6 * abs(losses[rank] + reg_exp)
        exp_shard = sh.router.shard_of(gt_exp)
        ok["table"] = float((sh.gshard - exp_shard).abs().max()) <= 4e-6 * float(gt_exp.abs().max())
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]                  # this rank's tensors (ZeRO chunks cut on tensor boundaries)
        ok["dense"] = float((sh.gchunk[:hi - lo] - gd_exp[lo:hi]).abs().max()) <= 2e-6 * float(gd_exp.abs().max()) \
            and (hi - lo == sh.chunk or float(sh.gchunk[hi - lo:].abs().max()) == 0.0)
        starts = {t[2] for t in model._ptensors} | {0, model.dense.numel()}
        ok["bounds"] = all(v in starts for v in sh.bounds) and sh.bounds[0] == 0 and sh.bounds[-1] == model.dense.numel()
        ok["split"] = sum(sh._keep[0].send) == sh._keep[0].n_unique and len(sh._keep[0].recv) == world
        # ZeRO-1 Adam step: replicas agree on the dense parameters, the table shards reassemble
        before = model.dense.data.clone()
        sh.adam_step()
        mine = model.dense.data.cpu()
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        ok["adam_same"] = all(torch.equal(outs[0], o) for o in outs)
        ok["adam_moved"] = float((model.dense.data - before).abs().max()) > 1e-4
        full = sh.full_table()
        ok["table_rows"] = tuple(full.shape) == tuple(W.shape) and float((full - W).abs().max()) <= 1.1e-3   # lr step
        q.put((rank, ok))
    except Exception as e:                                                 # surface the failure in the parent
        import traceback
        q.put((rank, {"exception: " + traceback.format_exc(): False}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert sorted(r[0] for r in res) == list(range(world))
    for _, ok in res:
        assert all(ok.values()), ok


@pytest.mark.parametrize("world", [1, 3, 8])
@pytest.mark.parametrize("shape", ["tiny", "amazon"])
def test_route_build_matches_tensor_ops(world, shape):
    """aread_route_build (sort-free dedupe on the device) == torch.unique statement of the same index math, bit-exact;
    ragged shards (rows % world != 0), rows_per_rank < 16, and workspace reuse across calls."""
    import aread_amd.dist as D
    rng = np.random.default_rng(5)
    if shape == "tiny":
        dims, B = [37, 3, 5], 50
    else:
        dims, B = [1368287, 7, 25, 45, 11, 22356, 10], 8192
    n_rows = sum(dims)
    off = np.concatenate([[0], np.cumsum(dims)[:-1]]).astype(np.int32)
    router = D.ShardRouter(n_rows, world=world, rank=0)
    offsets = torch.from_numpy(off).cuda()
    for it in range(3):                                          # the flag workspace must come back zeroed
        x = np.stack([rng.integers(0, d, B) for d in dims], axis=1).astype(np.int32)
        x[rng.random(x.shape[0]) < 0.5, 0] = dims[0] - 1          # hot row
        xd = torch.from_numpy(x).cuda()
        slot, uniq, edges = router.dedupe_hip(xd, offsets)
        rslot, runiq, redges = router.dedupe(xd + offsets)
        n = int(redges[-1])
        assert torch.equal(edges, redges)
        assert torch.equal(uniq[:n], runiq)
        assert torch.equal(slot, rslot)
    assert int(router._ws[:router.rows_per_rank * world].sum()) == 0


def test_sharded_adam_skips_unreached_tensors_like_torch():
    """ShardedTableStep.adam_step must not touch tensors whose gradient the reference leaves at None (towers /
    gates / heads no mask reaches): two steps (one rank, RCCL) against train_step + torch.optim.Adam(model.parameters()),
    every domain on the mask that reaches the fewest towers."""
    import torch.distributed as dist
    import aread_amd
    import aread_amd.dist as D
    fn, mk, seed = U.GOLDEN_MODELS["full"]
    G, spec = U.load_golden(fn), mk()
    sparse = U.golden_masks(spec, G, "sparse")
    reach = [sum(int(np.asarray(l).any(axis=0).sum()) for l in mk_[:-1]) for mk_ in sparse]
    masks = [sparse[int(np.argmin(reach))]] * spec.n_domain
    x, y = _rank_batch(spec, G, 0)
    hyper = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)

    def fresh():
        model, _ = U.build_model(spec, seed, dropout=0.0)
        model.train()
        model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk_] for mk_ in masks]
        return model, aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")

    a, md = fresh()
    init = a.dense.data.clone()
    opt = torch.optim.Adam(a.parameters(), **hyper)
    bufs = a.make_step_buffers(x.shape[0])
    for _ in range(2):
        a.zero_grad(set_to_none=True)
        a.train_step(x, y, bufs, masks_dev=md)
        opt.step()
    never = np.array([p.grad is None for p in a.dense_params])
    assert never.any()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    D.FORCE_COLLECTIVES = True
    try:
        b, md = fresh()
        sh = D.ShardedTableStep(b, x.shape[0])
        for _ in range(2):
            sh.step(x, y, md)
            sh.adam_step(**hyper)
        torch.cuda.synchronize()
        for on, (name, kind, off, shape, l2) in zip(never, b._ptensors):
            n = int(np.prod(shape)) if shape else 1
            if on:                                                    # untouched: bit for bit the initial values, zero moments
                assert torch.equal(b.dense.data[off:off + n], init[off:off + n]), name
                assert float(sh._opt["m_chunk"][off:off + n].abs().max()) == 0.0, name
        dd = (a.dense.data - b.dense.data).abs()
        assert float(dd.mean()) <= 2e-7 and int((dd > 5e-5).sum()) <= dd.numel() // 2000
        dw = (a.embedding.embedding_dict.weight.data - sh.full_table()).abs()
        assert float(dw.max()) <= 3e-4 and float(dw.mean()) <= 1e-6
    finally:
        D.FORCE_COLLECTIVES = False
        dist.destroy_process_group()
