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


def gather_ids(x, sample_row, rows, group=None):
    """all_gather the id matrix and the row map; the row map of rank r is rebased by r*rows so that it indexes the
    concatenation of every rank's dE buffer.  Returns (x_all [P*B,F], sample_row_all [P*B])."""
    world = dist.get_world_size(group)
    B = x.shape[0]
    x_all = torch.empty((world * B, x.shape[1]), dtype=x.dtype, device=x.device)
    sr_all = torch.empty(world * B, dtype=sample_row.dtype, device=x.device)
    dist.all_gather_into_tensor(x_all, x.contiguous(), group=group)
    dist.all_gather_into_tensor(sr_all, sample_row.contiguous(), group=group)
    sr_all += (torch.arange(world, device=x.device, dtype=sr_all.dtype) * rows).repeat_interleave(B)
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
        st = m.step_local(x, y, b, masks_dev=masks_dev, with_dense_l2=False, presort=False, plan=plan)
        de_all = gather_rows(b["de"], self.group)
        m.step_finish(b)
        reduce_dense_grad(b["gdense"], self.group)
        m.add_dense_l2(b)
        main.wait_stream(side2)
        m.embedding.reduce_sorted(x_all, de_all, b["gtable"])
        torch.add(b["loss"][:1], b["reg"][:1], out=b["total"])
        self._keep = (x_all, sr_all, de_all, st)
        return b["total"]
