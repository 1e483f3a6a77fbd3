"""Single-node multi-GPU step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Variant implemented (SURVEY 8e "fallback", stated in DESIGN.md): the 178 MB embedding table is REPLICATED
(it is 0.06 % of one MI355X's HBM) and the samples are data-parallel.  Forward needs no communication.
Backward exchanges only what is sparse:

  * all_gather of the id matrix x [B,F_in] int32, of the row map and of dE [rows,D] (gradient w.r.t. the
    pooled embedding output) -- fixed sizes, no host sync; every rank then runs the same deterministic
    sorted segmented reduction over the global batch, so all replicas hold bit-identical table gradients
    (dense L2 term 2*l2*W is computed locally, it is identical everywhere);
  * all_reduce (sum) of the flat 2.3 MB dense gradient, then the dense L2 term is added once.

BatchNorm statistics stay per replica and per domain segment (DDP without SyncBN).
The collectives and index bookkeeping are plain tensor code, exercised on CPU with gloo in
tests/test_dist_cpu.py; the compute is the HIP library (no fallback).
"""
import torch
import torch.distributed as dist


_REBASE = {}


def _rebase(world, B, rows, dtype, device):
    """[r*rows for r in range(world) for _ in range(B)] -- constant per configuration, built once"""
    key = (world, B, rows, dtype, str(device))
    if key not in _REBASE:
        _REBASE[key] = (torch.arange(world, device=device, dtype=dtype) * rows).repeat_interleave(B)
    return _REBASE[key]


def gather_ids(x, sample_row, rows, group=None):
    """all_gather the id matrix and the row map; the row map of rank r is rebased by r*rows so that it indexes the
    concatenation of every rank's dE buffer.  Returns (x_all [P*B,F], sample_row_all [P*B])."""
    world = dist.get_world_size(group)
    B = x.shape[0]
    x_all = torch.empty((world * B, x.shape[1]), dtype=x.dtype, device=x.device)
    sr_all = torch.empty(world * B, dtype=sample_row.dtype, device=x.device)
    dist.all_gather_into_tensor(x_all, x.contiguous(), group=group)
    dist.all_gather_into_tensor(sr_all, sample_row.contiguous(), group=group)
    sr_all += _rebase(world, B, int(rows), sr_all.dtype, x.device)
    return x_all, sr_all


def gather_rows(de, group=None):
    """all_gather dE [rows, D] -> [P*rows, D]"""
    world = dist.get_world_size(group)
    de_all = torch.empty((world * de.shape[0], de.shape[1]), dtype=de.dtype, device=de.device)
    dist.all_gather_into_tensor(de_all, de.contiguous(), group=group)
    return de_all


def gather_sparse_grad_inputs(x, sample_row, de, group=None):
    """(x, sample_row, de) of every rank, concatenated, with rebased row maps.  Any backend/device.
    Returns (x_all [P*B,F], sample_row_all [P*B], de_all [P*rows,D])."""
    x_all, sr_all = gather_ids(x, sample_row, de.shape[0], group)
    return x_all, sr_all, gather_rows(de, group)


def reduce_dense_grad(g, group=None):
    """Sum of the flat dense gradient over the ranks (in place)."""
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


class DataParallelStep:
    """Weak-scaling training step: B samples per rank, replicated parameters.

    Stream plan of one step (all eager, fork-join):
      main : row plan -> [gather/forward/backward of the local batch] -> all_gather(dE) -> join -> all_reduce(dense
             grads) -> dense L2 -> segmented reduction of the GLOBAL sorted lookups into the table gradient
      side2: all_gather(ids, row maps) -> radix sort of the global lookups        (overlaps the local compute)
    (the library's own side stream runs the weight-gradient GEMMs, the table L2 pass, ... as in the 1-GPU step)
    """

    def __init__(self, model, B, group=None, force_overlap=False):
        self.model, self.group = model, group
        self.force_overlap = force_overlap      # tests: take the multi-rank code path even with one rank
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bufs = model.make_step_buffers(B, multi_domain=True)
        self._side2 = None

    # ---- single-phase API (kept for tests / graph capture of the local part) -------------------------------
    def local(self, x, y, masks_dev):
        return self.model.step_local(x, y, self.bufs, masks_dev=masks_dev, with_dense_l2=False, presort=self.world == 1)

    def exchange_and_scatter(self, x, st):
        m, b = self.model, self.bufs
        if self.world > 1:
            x_all, sr_all, de_all = gather_sparse_grad_inputs(x, st.plan.sample_row, b["de"], self.group)
            m.step_finish(b)                                        # dense gradients complete
            reduce_dense_grad(b["gdense"], self.group)
            m.add_dense_l2(b)
            m.step_scatter(x_all, de_all, sr_all, b["gtable"])
        else:
            m.embedding.reduce_sorted(x, b["de"], b["gtable"])      # the sort already ran on the side stream
            m.step_finish(b)
            m.add_dense_l2(b)
        torch.add(b["loss"][:1], b["reg"][:1], out=b["total"])
        return b["total"]

    # ---- overlapped step ---------------------------------------------------------------------------------------
    def step(self, x, y, masks_dev):
        from .plan import RowPlan
        m, b = self.model, self.bufs
        if self.world == 1 and not self.force_overlap:
            return self.exchange_and_scatter(x, self.local(x, y, masks_dev))
        if self._side2 is None:
            self._side2 = torch.cuda.Stream(device=x.device)
        main, side2 = torch.cuda.current_stream(), self._side2
        plan = RowPlan(x, m.domain_idx, m.n_domain)
        side2.wait_stream(main)
        with torch.cuda.stream(side2):
            x_all, sr_all = gather_ids(x, plan.sample_row, plan.max_rows, self.group)
            m.embedding.sort_lookups(x_all, sr_all)
            x_all.record_stream(side2); sr_all.record_stream(side2)
            sorted_ev = torch.cuda.Event()
            sorted_ev.record(side2)
        st = m.step_local(x, y, b, masks_dev=masks_dev, with_dense_l2=False, presort=False, plan=plan)
        de_all = gather_rows(b["de"], self.group)
        # dense gradients (join, all_reduce, L2) on the side stream, concurrently with the global table reduction
        side2.wait_stream(main)
        with torch.cuda.stream(side2):
            m.step_finish(b)
            reduce_dense_grad(b["gdense"], self.group)
            m.add_dense_l2(b)
        main.wait_event(sorted_ev)                                   # global lookups sorted
        m.embedding.reduce_sorted(x_all, de_all, b["gtable"])
        main.wait_stream(side2)
        torch.add(b["loss"][:1], b["reg"][:1], out=b["total"])
        self._keep = (x_all, sr_all, de_all, st)
        return b["total"]


# ======================================================================================================================
# Row-sharded embedding table (SURVEY 8e "partitioning (north star)"): all-to-all lookup, sharded gradients
# ======================================================================================================================
def _backend(group=None):
    return dist.get_backend(group) if dist.is_initialized() else None


FORCE_COLLECTIVES = False          # tests: issue the collectives even in a one-rank group


def _world(group=None):
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _alone(group=None):
    return not dist.is_initialized() or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES)


def _rank(group=None):
    return dist.get_rank(group) if dist.is_initialized() else 0


def all_to_all_rows(inp, out_splits, in_splits, group=None):
    """Variable-size all-to-all along dim 0 (splits are host ints).  RCCL takes device tensors directly; gloo
    (CPU rehearsal of the same code) is staged through host memory when the tensors live on a device."""
    out = torch.empty((int(sum(out_splits)),) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    if _alone(group):
        out.copy_(inp)
        return out
    if _backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), list(out_splits), list(in_splits), group=group)
        out.copy_(o)
        return out
    dist.all_to_all_single(out, inp.contiguous(), list(out_splits), list(in_splits), group=group)
    return out


def reduce_scatter_flat(chunk_out, flat_in, group=None):
    """chunk_out = sum over ranks of this rank's chunk of flat_in (flat_in.numel() == world * chunk_out.numel())."""
    if _alone(group):
        chunk_out.copy_(flat_in)
    elif _backend(group) == "gloo":                      # gloo has no reduce_scatter: all_reduce + slice (tests only)
        t = flat_in.cpu() if flat_in.is_cuda else flat_in.clone()
        dist.all_reduce(t, group=group)
        n = chunk_out.numel()
        chunk_out.copy_(t[_rank(group) * n:(_rank(group) + 1) * n])
    else:
        dist.reduce_scatter_tensor(chunk_out, flat_in, op=dist.ReduceOp.SUM, group=group)
    return chunk_out


def all_gather_flat(flat_out, chunk_in, group=None):
    if _alone(group):
        flat_out.copy_(chunk_in)
    elif _backend(group) == "gloo" and chunk_in.is_cuda:
        o = torch.empty(flat_out.shape, dtype=flat_out.dtype)
        dist.all_gather_into_tensor(o, chunk_in.cpu(), group=group)
        flat_out.copy_(o)
    else:
        dist.all_gather_into_tensor(flat_out, chunk_in.contiguous(), group=group)
    return flat_out


def all_reduce_any(t, group=None):
    if _alone(group):
        return t
    if _backend(group) == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


class Route:
    """Result of ShardRouter.route for one batch.  overflow: device bool (fixed-capacity exchange only) -- a peer's share of
    the unique rows exceeded the capacity, the step's results are invalid (ShardedTableStep checks it one step later)."""
    __slots__ = ("slot", "n_unique", "send", "recv", "recv_rows", "overflow")


def all_to_all_equal(inp, group=None):
    """Equal-split all-to-all along dim 0 (inp.shape[0] % world == 0): no split sizes, nothing read on the host."""
    if _alone(group):
        return inp.clone()
    out = torch.empty_like(inp)
    if _backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(inp.shape, dtype=inp.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), group=group)
        out.copy_(o)
        return out
    dist.all_to_all_single(out, inp.contiguous(), group=group)
    return out


class ShardRouter:
    """Index bookkeeping + collectives of the row-sharded table; pure tensor code, any backend / device.

    Global table row r lives on rank r % P at local row r // P (interleaved, so the hot neighbours of the pad /
    domain rows spread over the ranks).  route() deduplicates this rank's lookups (the pad row alone is ~37 % of
    them), groups the unique rows by owner and exchanges the requests:
        all_to_all #0  per-peer counts (P ints)            -> one host read of 2P ints (variable split sizes)
        all_to_all #1  local row ids of the requested rows
    fetch(): all_to_all #2 the requested rows come back in request order; push(): all_to_all #3 sends the
    per-unique-row gradients to the owners in the same order (splits of #1 reversed)."""

    def __init__(self, n_rows, group=None, world=None, rank=None):
        self.group = group
        self.P = _world(group) if world is None else int(world)      # world/rank overrides: index logic tests
        self.p = _rank(group) if rank is None else int(rank)
        self.n_rows = int(n_rows)
        self.rows_per_rank = (self.n_rows + self.P - 1) // self.P
        self._ws = None

    def shard_of(self, table):
        """this rank's rows of a full [R, E] table, zero-padded to rows_per_rank"""
        part = table[self.p::self.P]
        out = torch.zeros((self.rows_per_rank,) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
        out[:part.shape[0]].copy_(part)
        return out

    def unshard(self, shards, n_rows=None):
        """inverse of shard_of over a list of every rank's shard (tests / checkpointing)"""
        n_rows = self.n_rows if n_rows is None else n_rows
        out = torch.empty((n_rows,) + tuple(shards[0].shape[1:]), dtype=shards[0].dtype, device=shards[0].device)
        for q, s in enumerate(shards):
            n = out[q::self.P].shape[0]
            out[q::self.P] = s[:n]
        return out

    # ---- dedupe: (slot per lookup, unique local rows grouped by owner, owner boundaries) ---------------------------
    def dedupe(self, bag):
        """tensor-op statement of the routing index math (any device; the CPU/gloo rehearsal uses it)"""
        P, Rp = self.P, self.rows_per_rank
        g = bag.reshape(-1).to(torch.int64)
        key = (g % P) * Rp + g // P                                  # owner-major: unique() output is grouped by owner
        uniq, inv = torch.unique(key, return_inverse=True)
        edges = torch.searchsorted(uniq, torch.arange(P + 1, device=uniq.device, dtype=torch.int64) * Rp)
        return inv.to(torch.int32).reshape(bag.shape), (uniq % Rp).to(torch.int32), edges.to(torch.int32)

    def dedupe_hip(self, x, offsets):
        """the same on the device without a sort: aread_route_build (csrc/route.hip); bag = x + offsets"""
        from . import _lib as L
        L.require_device(x, offsets)
        L.require(x, torch.int32, "x")
        lib = L.lib()
        if self._ws is None or self._ws.device != x.device:
            self._ws = torch.zeros(int(lib.aread_route_ws_bytes(self.n_rows, self.P)), dtype=torch.uint8, device=x.device)
        slot = torch.empty(x.shape, dtype=torch.int32, device=x.device)
        uniq = torch.empty(x.numel(), dtype=torch.int32, device=x.device)
        edges = torch.empty(self.P + 1, dtype=torch.int32, device=x.device)
        L.check(lib.aread_route_build(L.ptr(x), x.shape[0], x.shape[1], L.ptr(offsets), self.n_rows, self.P, L.ptr(self._ws),
                                      L.ptr(slot), L.ptr(uniq), L.ptr(edges), 0, L.stream()))
        return slot, uniq, edges

    def _exchange(self, slot, uniq_rows, edges):
        P = self.P
        send_cnt = (edges[1:] - edges[:-1]).to(torch.int64)
        both = torch.stack([send_cnt, all_to_all_rows(send_cnt, [1] * P, [1] * P, self.group)]).tolist()   # host read
        r = Route()
        r.overflow = None
        r.send, r.recv = [int(v) for v in both[0]], [int(v) for v in both[1]]
        r.n_unique = sum(r.send)
        r.slot = slot                                                # lookup -> slot in this rank's unique-row buffer
        r.recv_rows = all_to_all_rows(uniq_rows[:r.n_unique], r.recv, r.send, self.group)        # local rows asked of me
        return r

    def _exchange_fixed(self, slot, uniq_rows, edges, cap):
        """Fixed-capacity exchange: every rank asks every peer for exactly `cap` rows -- its unique rows of that owner, padded
        with the owner's local row 0 -- so the split sizes are constants: NO host read, the whole sharded step can be queued
        (and captured) without waiting for the routing kernels.  The unique-row buffer then has P*cap slots (owner-major,
        `cap` per owner); padded slots are never referenced by a lookup, fetch whatever row 0 holds and push zero gradients.
        A peer share above the capacity sets route.overflow (device flag): the caller checks it off the critical path."""
        P = self.P
        dev = uniq_rows.device
        e = edges.to(torch.int64)                                           # [P+1] owner boundaries of the compact unique list
        n = uniq_rows.numel()
        ar = torch.arange(n, device=dev, dtype=torch.int64)
        owner = torch.bucketize(ar, e[1:].contiguous(), right=True).clamp_(max=P - 1)      # owner of compact entry i (garbage past n_unique)
        pos = ar - e[owner]
        ok = (ar < e[P]) & (pos < cap)
        dst = torch.where(ok, owner * cap + pos, torch.full_like(ar, P * cap))               # dropped entries go to a dummy slot
        req = torch.zeros(P * cap + 1, dtype=uniq_rows.dtype, device=dev)
        req.scatter_(0, dst, uniq_rows)
        s64 = slot.reshape(-1).to(torch.int64)
        so = torch.bucketize(s64, e[1:].contiguous(), right=True).clamp_(max=P - 1)
        r = Route()
        r.slot = (s64 - e[so] + so * cap).clamp_(max=P * cap - 1).to(torch.int32).reshape(slot.shape)
        r.overflow = ((e[1:] - e[:-1]) > cap).any()
        r.send, r.recv = [cap] * P, [cap] * P
        r.n_unique = P * cap
        r.recv_rows = all_to_all_equal(req[:P * cap].contiguous(), self.group)             # local rows asked of me, cap per peer
        return r

    def route(self, bag, capacity=None):
        if capacity is not None:
            return self._exchange_fixed(*self.dedupe(bag), int(capacity))
        return self._exchange(*self.dedupe(bag))

    def route_hip(self, x, offsets, capacity=None):
        if capacity is not None:
            return self._exchange_fixed(*self.dedupe_hip(x, offsets), int(capacity))
        return self._exchange(*self.dedupe_hip(x, offsets))

    def peer_counts(self, edges):
        """unique rows this rank asks of each owner (device tensor [P]); capacity calibration reads its maximum once"""
        return (edges[1:] - edges[:-1]).to(torch.int64)

    def fetch(self, route, rows):
        """rows [sum(recv), E] (this rank's rows for route.recv_rows) -> [n_unique, E] in slot order"""
        if getattr(route, "overflow", None) is not None:
            return all_to_all_equal(rows, self.group)
        return all_to_all_rows(rows, route.send, route.recv, self.group)

    def push(self, route, g_unique):
        """g_unique [n_unique, E] -> [sum(recv), E] aligned with route.recv_rows"""
        if getattr(route, "overflow", None) is not None:
            return all_to_all_equal(g_unique, self.group)
        return all_to_all_rows(g_unique, route.recv, route.send, self.group)


def zero_bounds(tensors, n, P):
    """ZeRO-1 chunk boundaries of the flat dense parameter cut ON TENSOR BOUNDARIES: b[0] = 0 <= b[1] <= ... <= b[P] = n with
    every b[q] the start offset of a tensor (the start nearest to q*n/P), so that a whole expert / tower / gate tensor -- weights,
    its Adam moments, its step count -- lives on one rank (north star: "domain-expert towers partition across the GPUs").
    tensors: (name, kind, offset, shape, l2) of the model's trainable tensors."""
    import numpy as np
    starts = np.array(sorted({int(t[2]) for t in tensors} | {0}), dtype=np.int64)
    b = [0]
    for q in range(1, P):
        target = q * n / P
        cand = int(starts[np.argmin(np.abs(starts - target))])
        b.append(max(cand, b[-1]))
    b.append(int(n))
    return b


class ShardedTableStep:
    """Weak-scaling training step with the embedding table row-sharded over the ranks and ZeRO-1 style dense
    parameters (SURVEY 8e north-star partitioning; BASELINE config 4).

      lookup    : ids -> ShardRouter.route (dedupe, all_to_all ids) -> owners gather their rows (aread_embed_fwd,
                  one column) -> all_to_all rows -> aread_embed_fwd over the received unique rows (slot ids, zero
                  offsets, the row plan) -> pooled embedding e, bit-identical to the unsharded gather
      dense     : rank-local forward / bagging BCE / backward (libaread_hip), BatchNorm per replica
      table grad: aread_embed_bwd into the unique-row buffer -> all_to_all to the owners -> aread_embed_bwd (one
                  column) into the shard gradient, which the shard-local L2 pass initialised with 2*l2*W_shard
      dense grad: reduce_scatter of the flat gradient (equal chunks), dense L2 on the owned chunk; adam_step()
                  updates shard + owned chunk and all_gathers the dense parameters.
    The loss convention is the DataParallelStep's: objective = sum over ranks of the rank losses + reg (once).
    Two host reads per step (unique count, split sizes) -- the usual price of a variable-size all-to-all."""

    def __init__(self, model, B, group=None, capacity=None):
        """capacity: rows per peer of the FIXED-capacity all-to-all (ShardRouter._exchange_fixed): no host read anywhere in the
        step (capturable); None = variable-size exchange with one host read of 2P counts (prefetched one batch ahead).
        calibrate_capacity(x) measures a value on a sample batch."""
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.model, self.group = model, group
        self.P, self.p = _world(group), _rank(group)
        self.capacity = None if capacity is None else int(capacity)
        self._ovf_prev = None
        emb = model.embedding
        table = emb.embedding_dict.weight.data
        dev = table.device
        self.router = ShardRouter(table.shape[0], group)
        self.shard = torch.nn.Parameter(self.router.shard_of(table))
        self.gshard = torch.empty_like(self.shard.data)
        self.bufs = model.make_step_buffers(B, multi_domain=True, with_table_grad=False)
        n = model.dense.numel()
        # ZeRO-1 chunks cut on tensor boundaries (zero_bounds); the collectives want equal sizes, so every rank's chunk is
        # padded to the longest one: packed layout [P][chunk], rank q's tensors at q*chunk .. q*chunk + (b[q+1] - b[q])
        self.bounds = zero_bounds(model._ptensors, n, self.P)
        self.chunk = (max(self.bounds[q + 1] - self.bounds[q] for q in range(self.P)) + 3) // 4 * 4
        self.gpad = torch.zeros(self.chunk * self.P, dtype=torch.float32, device=dev)
        self.dense_pad = torch.zeros(self.chunk * self.P, dtype=torch.float32, device=dev)
        self.coef_pad = torch.zeros(self.chunk * self.P, dtype=torch.float32, device=dev)
        self._pack(self.dense_pad, model.dense.data)
        self._pack(self.coef_pad, model._l2_coef(dev))
        lo = self.p * self.chunk
        self.dense_chunk = torch.nn.Parameter(self.dense_pad[lo:lo + self.chunk].clone())
        self.gchunk = torch.zeros(self.chunk, dtype=torch.float32, device=dev)
        self.reg = torch.zeros(257, dtype=torch.float32, device=dev)
        self.total = torch.zeros(1, dtype=torch.float32, device=dev)
        self._zero17 = torch.zeros(emb.offsets.shape[0], dtype=torch.int32, device=dev)
        self._zero1 = torch.zeros(1, dtype=torch.int32, device=dev)
        self._ws = {}
        self._pref, self._route_stream = None, None
        self._opt = None
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def _pack(self, padded, flat):
        """flat [n] (the library's tensor order) -> packed [P][chunk]"""
        for q in range(self.P):
            lo, hi = self.bounds[q], self.bounds[q + 1]
            if hi > lo:
                padded[q * self.chunk:q * self.chunk + hi - lo].copy_(flat[lo:hi])
        return padded

    def _unpack(self, flat, padded):
        for q in range(self.P):
            lo, hi = self.bounds[q], self.bounds[q + 1]
            if hi > lo:
                flat[lo:hi].copy_(padded[q * self.chunk:q * self.chunk + hi - lo])
        return flat

    def calibrate_capacity(self, x, margin=1.25):
        """rows per peer for the fixed-capacity exchange from a sample batch: the largest per-owner unique count over all
        ranks x margin, rounded up to 64 (ONE host read, outside the training loop); sets and returns self.capacity"""
        emb = self.model.embedding
        _, _, edges = self.router.dedupe_hip(x, emb._offsets_dev(x.device))
        mx = self.router.peer_counts(edges).max().reshape(1).to(torch.float32)
        if not _alone(self.group):
            if _backend(self.group) == "gloo" and mx.is_cuda:
                c = mx.cpu(); dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group); mx = c
            else:
                dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
        self.capacity = (int(float(mx) * margin) + 63) // 64 * 64
        return self.capacity

    # ---- C-ABI helpers -------------------------------------------------------------------------------------------
    def _bwd_ws(self, tag, B, f_in):
        L, E = self._L, self.model.embedding.embed_dim
        need = int(L.lib().aread_embed_bwd_ws_bytes(B, f_in, E))
        if need < 0:
            raise RuntimeError("aread_embed_bwd_ws_bytes failed")
        ws = self._ws.get(tag)
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, int(need * 1.25)), dtype=torch.uint8, device=self.shard.device)
            self._ws[tag] = ws
        return ws

    def _gather_owned(self, local_rows):
        """rows of this rank's shard for the requests of every peer: [n, E]"""
        L, emb = self._L, self.model.embedding
        n = int(local_rows.shape[0])
        out = torch.empty((n, emb.embed_dim), dtype=torch.float32, device=self.shard.device)
        if n:
            L.check(L.lib().aread_embed_fwd(L.ptr(local_rows), n, 1, L.ptr(self._zero1), L.ptr(self.shard.data),
                                            self.shard.shape[0], emb.embed_dim, 1, 0, 1, 0, None, n, L.ptr(out), None,
                                            L.stream()))
        return out

    def prefetch_route(self, x_next):
        """Routing of the NEXT step's batch on its own stream while the current step runs: the variable-size all-to-all needs
        2P counts on the host, and read inline that host read waits for everything queued before it -- the host can never
        run ahead of the GPU and every step pays the launch latencies un-overlapped.  Read from the route stream it waits
        only for the few routing kernels of the next batch.  x_next must keep its contents until the step that uses it.
        Collective order: every rank calls prefetch_route at the same point of its step."""
        emb = self.model.embedding
        if self._route_stream is None:
            self._route_stream = torch.cuda.Stream(device=x_next.device)
        rs = self._route_stream
        rs.wait_stream(torch.cuda.current_stream())                  # (x_next may have been produced on the main stream)
        with torch.cuda.stream(rs):
            route = self.router.route_hip(x_next, emb._offsets_dev(x_next.device), self.capacity)   # (variable-size: host read, waits for rs only)
            ev = torch.cuda.Event()
            ev.record(rs)
        self._pref = ((x_next.data_ptr(), tuple(x_next.shape)), route, ev)

    def lookup(self, x, plan, x_key=None):
        """-> (route, unique rows); fills bufs['e'] (plan order).  A prefetched route is used only when it was made for this
        batch: x is the tensor given to prefetch_route, or x_key is that tensor's data_ptr() (x being a copy of it)."""
        L, emb, b = self._L, self.model.embedding, self.bufs
        pref, self._pref = self._pref, None
        if pref is not None and pref[0] == ((x.data_ptr() if x_key is None else int(x_key)), tuple(x.shape)):
            route = pref[1]
            torch.cuda.current_stream().wait_event(pref[2])
            for t in (route.slot, route.recv_rows):
                t.record_stream(torch.cuda.current_stream())
        else:
            route = self.router.route_hip(x, emb._offsets_dev(x.device), self.capacity)
        if route.overflow is not None:
            # fixed capacity: the flag of THIS batch is looked at when the next step is issued (by then it was computed long ago)
            if self._ovf_prev is not None and bool(self._ovf_prev):
                raise RuntimeError(f"ShardedTableStep: a peer's share of the previous batch's unique rows exceeded capacity={self.capacity}: "
                                   "that step's lookup was incomplete -- raise the capacity (calibrate_capacity) and redo it")
            self._ovf_prev = route.overflow
        urows = self.router.fetch(route, self._gather_owned(route.recv_rows))
        L.check(L.lib().aread_embed_fwd(L.ptr(route.slot), x.shape[0], x.shape[1], L.ptr(self._zero17), L.ptr(urows),
                                        urows.shape[0], emb.embed_dim, emb.one_hot_field_num, emb.multi_hot_field_num,
                                        emb.seq_maxlen, emb._pool, L.ptr(plan.row_sample), plan.max_rows, L.ptr(b["e"]),
                                        None, L.stream()))
        return route, urows

    def presort(self, x, route, plan):
        """index sorts of both segmented reductions of table_grad (they depend only on the ids): side stream"""
        L, emb = self._L, self.model.embedding
        lib = L.lib()
        B, f_in = x.shape
        n = int(route.recv_rows.shape[0])
        ws_u, ws_o = self._bwd_ws("u", B, f_in), self._bwd_ws("o", max(n, 1), 1)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            L.check(lib.aread_embed_bwd_sort(L.ptr(route.slot), B, f_in, L.ptr(self._zero17), route.n_unique, emb.embed_dim,
                                             emb.one_hot_field_num, emb.multi_hot_field_num, emb.seq_maxlen, emb._pool,
                                             L.ptr(plan.sample_row), L.ptr(ws_u), L.stream()))
            if n:
                L.check(lib.aread_embed_bwd_sort(L.ptr(route.recv_rows), n, 1, L.ptr(self._zero1), self.shard.shape[0],
                                                 emb.embed_dim, 1, 0, 1, 0, None, L.ptr(ws_o), L.stream()))
            self._sorted_ev = torch.cuda.Event()
            self._sorted_ev.record(self._side)

    def table_grad(self, x, route, plan):
        """bufs['de'] -> self.gshard (+= on top of the shard's L2 gradient); presort() must have run"""
        L, emb, b = self._L, self.model.embedding, self.bufs
        lib = L.lib()
        B, f_in = x.shape
        seq = emb.seq_maxlen if emb._pool != 0 else 1
        g_unique = torch.zeros((route.n_unique, emb.embed_dim), dtype=torch.float32, device=x.device)
        torch.cuda.current_stream().wait_event(self._sorted_ev)     # both index sorts (presort) are done
        L.check(lib.aread_embed_bwd_reduce(B, f_in, emb.embed_dim, seq, L.ptr(b["de"]), L.ptr(g_unique),
                                           L.ptr(self._bwd_ws("u", B, f_in)), L.stream()))
        g_recv = self.router.push(route, g_unique)
        n = int(g_recv.shape[0])
        if n:
            L.check(lib.aread_embed_bwd_reduce(n, 1, emb.embed_dim, 1, L.ptr(g_recv), L.ptr(self.gshard),
                                               L.ptr(self._bwd_ws("o", n, 1)), L.stream()))

    def step(self, x, y, masks_dev, next_x=None, x_key=None):
        """next_x: the id tensor the NEXT call's batch comes from: its routing is prefetched while this step runs
        (prefetch_route), which removes the host read from the step's critical path.  x_key: data_ptr() of the tensor this
        call's x was copied from, when x is a staging copy (the prefetched route is matched by it)."""
        from .plan import RowPlan
        L, m, b = self._L, self.model, self.bufs
        lib = L.lib()
        plan = RowPlan(x, m.domain_idx, m.n_domain)
        route, urows = self.lookup(x, plan, x_key)
        if next_x is not None:
            self.prefetch_route(next_x)
        self.presort(x, route, plan)
        st = m.step_local(x, y, b, masks_dev=masks_dev, with_dense_l2=False, presort=False, plan=plan, e_ready=True,
                          l2_target=(self.shard.data, self.gshard))
        # dense gradients (join, reduce_scatter, dense L2 on the owned chunk, reg all_reduce) on the side stream,
        # concurrently with the table-gradient exchange on the main stream
        main, side = torch.cuda.current_stream(), self._side
        lo = self.p * self.chunk
        side.wait_stream(main)
        with torch.cuda.stream(side):
            m.step_finish(b)                                           # dense gradients of the local batch complete
            self._pack(self.gpad, b["gdense"])                         # tensor-aligned chunks, padded to equal length
            reduce_scatter_flat(self.gchunk, self.gpad, self.group)
            self.reg.copy_(b["reg"])                                   # table L2 of this shard
            L.check(lib.aread_l2_dense(L.ptr(self.dense_chunk.data), L.ptr(self.coef_pad[lo:lo + self.chunk]), self.chunk,
                                       L.ptr(self.gchunk), L.ptr(self.reg), 1, L.stream()))
            reg = all_reduce_any(self.reg[:1].clone(), self.group)     # reg = sum over shards / chunks
            reg.record_stream(side)
        self.table_grad(x, route, plan)
        main.wait_stream(side)
        torch.add(b["loss"][:1], reg, out=self.total)
        self._keep = (route, urows, st, plan)
        return self.total

    # ---- ZeRO-1 optimizer step (outside the fwd+bwd metric) ---------------------------------------------------------
    def adam_step(self, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, masks=None):
        """Adam (run.py:830-831 hyper-parameters) on the owned table shard and dense chunk with the fused kernels of
        csrc/optim.hip, then all_gather of the dense parameters into every replica's model.dense.  Like torch.optim.Adam
        (and aread_amd.FusedAdam) it SKIPS the tensors whose gradient the reference's autograd leaves at None -- towers,
        gates and heads that no domain mask reaches: no decay, no moment update, no step count -- through a per-element
        `active` mask cut to this rank's chunk, with torch's per-tensor step counts."""
        import numpy as np
        from .optim import AdamCfg
        C, L, m = self._C, self._L, self.model
        lib = L.lib()
        if self._opt is None:
            z = torch.zeros_like
            self._opt = dict(m_shard=z(self.shard.data), v_shard=z(self.shard.data), m_chunk=z(self.gchunk), v_chunk=z(self.gchunk),
                             t_table=0, t_dense=np.zeros(len(m._ptensors), dtype=np.int64), act={}, key=None, present=None)
        o = self._opt

        def cfg(step):
            c = AdamCfg()
            c.lr, c.beta1, c.beta2, c.eps, c.weight_decay, c.step = lr, betas[0], betas[1], eps, weight_decay, int(step)
            return c
        o["t_table"] += 1
        c = cfg(o["t_table"])
        L.check(lib.aread_adam_step(L.ptr(self.shard.data), L.ptr(self.gshard), L.ptr(o["m_shard"]), L.ptr(o["v_shard"]),
                                    self.shard.numel(), None, C.byref(c), L.stream()))
        masks = m.domain_mask if masks is None else masks
        key = ("version", m.mask_version) if masks is m.domain_mask else id(masks)
        if key != o["key"]:
            o["present"] = np.array([a or b for a, b in zip(m._presence(0, masks), m._reg_present)], dtype=bool)
            o["key"] = key
        present = o["present"]
        o["t_dense"][present] += 1
        lo = self.p * self.chunk
        for t in np.unique(o["t_dense"][present]):
            sel = present & (o["t_dense"] == t)
            k = sel.tobytes()
            if k not in o["act"]:                                   # uint8 [chunk]: 1 on this rank's elements of the selected tensors
                a = np.zeros(m.dense.numel(), dtype=np.uint8)
                for on, (name, kind, off, shape, l2) in zip(sel, m._ptensors):
                    if on:
                        a[off:off + (int(np.prod(shape)) if shape else 1)] = 1
                mine = np.zeros(self.chunk, dtype=np.uint8)              # this rank's tensors (zero_bounds), padded
                mine[:self.bounds[self.p + 1] - self.bounds[self.p]] = a[self.bounds[self.p]:self.bounds[self.p + 1]]
                o["act"][k] = torch.from_numpy(mine).to(self.gchunk.device)
            c = cfg(t)
            L.check(lib.aread_adam_step(L.ptr(self.dense_chunk.data), L.ptr(self.gchunk), L.ptr(o["m_chunk"]), L.ptr(o["v_chunk"]),
                                        self.chunk, L.ptr(o["act"][k]), C.byref(c), L.stream()))
        all_gather_flat(self.dense_pad, self.dense_chunk.data, self.group)
        self._unpack(self.model.dense.data, self.dense_pad)

    def full_table(self):
        """gathers the shards back into a [R, E] table (checkpointing / tests)"""
        flat = torch.empty((self.P,) + tuple(self.shard.shape), dtype=torch.float32, device=self.shard.device)
        all_gather_flat(flat.view(-1), self.shard.data.reshape(-1), self.group)
        return self.router.unshard(list(flat))
