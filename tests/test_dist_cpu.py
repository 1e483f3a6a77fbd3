"""CPU, gloo, world_size 2: the multi-GPU exchange logic (aread_amd/dist.py).

The collectives and row-map rebasing run for real; the local compute is stood in by the oracle's
scatter (tests may use the oracle, the product path may not)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import aread_oracle as O
from tests import util as U


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_data(spec, rank, B, rows):
    rng = np.random.default_rng(100 + rank)
    x = np.stack([rng.integers(0, d, B) for d in spec.field_dims]
                 + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
    perm = rng.permutation(rows)[:B].astype(np.int32)            # a row plan: sample b lives in row perm[b]
    de = rng.standard_normal((rows, spec.d)).astype(np.float32)
    g = rng.standard_normal(50).astype(np.float32)
    return x, perm, de, g


def _scatter_ref(spec, x, sample_row, de):
    """oracle scatter: table_grad[bag[b,j]] += c_j * de[sample_row[b], out_field(j)]"""
    bag = O.index_bag(x, spec).astype(np.int64)
    out = np.zeros((spec.rows, spec.embed_dim), np.float64)
    de3 = de.reshape(de.shape[0], spec.f_out, spec.embed_dim).astype(np.float64)
    for j in range(spec.f_in):
        mh = j >= spec.n_onehot
        fo = spec.n_onehot + (j - spec.n_onehot) // spec.seq_maxlen if mh else j
        c = 1.0 / spec.seq_maxlen if (mh and spec.method == "mean") else 1.0
        np.add.at(out, bag[:, j], c * de3[sample_row, fo, :])
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import aread_amd.dist as D
        spec = U.spec_full()
        B, rows = 40, 64
        x, perm, de, g = _rank_data(spec, rank, B, rows)
        x_all, sr_all, de_all = D.gather_sparse_grad_inputs(torch.from_numpy(x), torch.from_numpy(perm),
                                                            torch.from_numpy(de))
        gsum = D.reduce_dense_grad(torch.from_numpy(g.copy()))
        got = _scatter_ref(spec, x_all.numpy(), sr_all.numpy().astype(np.int64), de_all.numpy())
        # expected: the sum of every rank's own scatter (rank-local data regenerated from the seeds)
        exp = np.zeros_like(got)
        gexp = np.zeros_like(g)
        for r in range(world):
            xr, pr, dr, gr = _rank_data(spec, r, B, rows)
            exp += _scatter_ref(spec, xr, pr.astype(np.int64), dr)
            gexp += gr
        ok = (np.allclose(got, exp, rtol=1e-12, atol=1e-12) and np.allclose(gsum.numpy(), gexp, rtol=1e-6)
              and x_all.shape == (world * B, spec.f_in) and de_all.shape == (world * rows, spec.d))
        # replicas must agree bit for bit: gather every rank's result on rank 0
        t = torch.from_numpy(got.astype(np.float32))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        same = all(torch.equal(outs[0], o) for o in outs)
        q.put((rank, bool(ok), bool(same)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_and_reduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), res


# ---- row-sharded table: ShardRouter (all-to-all lookup / gradient push), reduce-scatter helpers ------------------------
def _shard_data(spec, rank, B, n_rows, E):
    rng = np.random.default_rng(500 + rank)
    bag = rng.integers(0, n_rows, (B, 17)).astype(np.int32)
    bag[rng.random((B, 17)) < 0.4] = n_rows - 3               # a hot (pad-like) row, deduplicated before sending
    g_lookup = rng.standard_normal((B * 17, E)).astype(np.float32)
    return bag, g_lookup


def _shard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import aread_amd.dist as D
        n_rows, E, B = 1001, 8, 64                               # 1001 rows: the shards are ragged (334/334/333)
        table = torch.from_numpy(np.random.default_rng(7).standard_normal((n_rows, E)).astype(np.float32))
        router = D.ShardRouter(n_rows)
        shard = router.shard_of(table)
        bag, g_lookup = _shard_data(None, rank, B, n_rows, E)
        route = router.route(torch.from_numpy(bag))
        urows = router.fetch(route, shard[route.recv_rows.long()])
        ok_fwd = torch.equal(urows[route.slot.long()], table[torch.from_numpy(bag).long()])
        ok_dedupe = route.n_unique == len(np.unique(bag)) and sum(route.send) == route.n_unique
        g_unique = torch.zeros(route.n_unique, E, dtype=torch.float64).index_add_(
            0, route.slot.reshape(-1).long(), torch.from_numpy(g_lookup).double())
        g_recv = router.push(route, g_unique)
        gshard = torch.zeros(shard.shape, dtype=torch.float64).index_add_(0, route.recv_rows.long(), g_recv)
        shards = [torch.empty_like(gshard) for _ in range(world)]
        dist.all_gather(shards, gshard)
        got = router.unshard(shards).numpy()
        exp = np.zeros((n_rows, E))
        for r in range(world):
            br, gr = _shard_data(None, r, B, n_rows, E)
            np.add.at(exp, br.reshape(-1), gr.astype(np.float64))
        ok_bwd = np.allclose(got, exp, rtol=1e-12, atol=1e-12)
        # fixed-capacity exchange (constant split sizes, nothing read on the host): same rows, same gradients; overflow flag
        cap = 64 * 17
        rf = router.route(torch.from_numpy(bag), capacity=cap)
        uf = router.fetch(rf, shard[rf.recv_rows.long()])
        ok_fwd = ok_fwd and rf.n_unique == world * cap and not bool(rf.overflow) and \
            torch.equal(uf[rf.slot.long()], table[torch.from_numpy(bag).long()])
        gu = torch.zeros(rf.n_unique, E, dtype=torch.float64).index_add_(0, rf.slot.reshape(-1).long(), torch.from_numpy(g_lookup).double())
        gr = router.push(rf, gu)
        gshard_f = torch.zeros(shard.shape, dtype=torch.float64).index_add_(0, rf.recv_rows.long(), gr)
        ok_bwd = ok_bwd and torch.allclose(gshard_f, gshard, rtol=1e-12, atol=1e-12)
        ok_dedupe = ok_dedupe and bool(router.route(torch.from_numpy(bag), capacity=8).overflow)
        # shard_of / unshard round trip over every rank's shard
        tabs = [torch.empty_like(shard) for _ in range(world)]
        dist.all_gather(tabs, shard)
        ok_rt = torch.equal(router.unshard(tabs), table)
        # flat reduce-scatter / all-gather helpers
        chunk = 5
        flat = torch.arange(world * chunk, dtype=torch.float32) * (rank + 1)
        mine = D.reduce_scatter_flat(torch.empty(chunk), flat)
        tot = sum(range(1, world + 1))
        ok_rs = torch.equal(mine, torch.arange(rank * chunk, (rank + 1) * chunk, dtype=torch.float32) * tot)
        back = D.all_gather_flat(torch.empty(world * chunk), mine)
        ok_ag = torch.equal(back, torch.arange(world * chunk, dtype=torch.float32) * tot)
        q.put((rank, bool(ok_fwd), bool(ok_dedupe), bool(ok_bwd), bool(ok_rt), bool(ok_rs), bool(ok_ag)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_shard_router(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(all(r[1:]) for r in res), res


def test_shard_router_single_process():
    """no process group: the router degenerates to dedupe + identity exchange"""
    import aread_amd.dist as D
    router = D.ShardRouter(50)
    bag = torch.tensor([[3, 3, 49], [0, 3, 7]], dtype=torch.int32)
    route = router.route(bag)
    assert route.n_unique == 4 and route.send == [4] and route.recv == [4]
    assert route.recv_rows.tolist() == [0, 3, 7, 49]
    assert route.slot.tolist() == [[1, 1, 3], [0, 1, 2]]
    rf = router.route(bag, capacity=6)
    assert rf.n_unique == 6 and rf.recv_rows.tolist() == [0, 3, 7, 49, 0, 0] and rf.slot.tolist() == route.slot.tolist() and not bool(rf.overflow)
    assert bool(router.route(bag, capacity=3).overflow)


def test_zero_bounds_on_tensor_boundaries():
    """ZeRO-1 chunks: boundaries are tensor starts, monotone, cover [0, n], near the equal split"""
    import aread_amd.dist as D
    rng = np.random.default_rng(1)
    sizes = rng.integers(1, 5000, 200)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    tensors = [("t%d" % i, 0, int(offs[i]), (int(sizes[i]),), 0.0) for i in range(len(sizes))]
    n = int(offs[-1])
    for P in (1, 2, 3, 8):
        b = D.zero_bounds(tensors, n, P)
        assert len(b) == P + 1 and b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(P))
        assert all(v in set(offs.tolist()) for v in b)
        assert max(abs(b[q] - q * n / P) for q in range(P + 1)) <= 5000
