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


def gather_sparse_grad_inputs(x, sample_row, de, group=None):
    """all_gather (x, sample_row, de) over the ranks and rebase each rank's row map into the concatenated
    dE buffer.  Works on any backend/device.  Returns (x_all [P*B,F], sample_row_all [P*B], de_all [P*rows,D])."""
    world = dist.get_world_size(group)
    B, rows = x.shape[0], de.shape[0]
    x_all = torch.empty((world * B, x.shape[1]), dtype=x.dtype, device=x.device)
    sr_all = torch.empty(world * B, dtype=sample_row.dtype, device=x.device)
    de_all = torch.empty((world * rows, de.shape[1]), dtype=de.dtype, device=de.device)
    dist.all_gather_into_tensor(x_all, x.contiguous(), group=group)
    dist.all_gather_into_tensor(sr_all, sample_row.contiguous(), group=group)
    dist.all_gather_into_tensor(de_all, de.contiguous(), group=group)
    sr_all += (torch.arange(world, device=x.device, dtype=sr_all.dtype) * rows).repeat_interleave(B)
    return x_all, sr_all, de_all


def reduce_dense_grad(g, group=None):
    """Sum of the flat dense gradient over the ranks (in place)."""
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g


class DataParallelStep:
    """Weak-scaling training step: B samples per rank, replicated parameters."""

    def __init__(self, model, B, group=None):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bufs = model.make_step_buffers(B, multi_domain=True)

    def local(self, x, y, masks_dev):
        """graph-capturable: everything that needs no communication"""
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

    def step(self, x, y, masks_dev):
        st = self.local(x, y, masks_dev)
        return self.exchange_and_scatter(x, st)
