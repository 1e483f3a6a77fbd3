"""GPU: the hipGraph-replayed training step -- the launch mode bench.py times -- computes exactly what the eager step does.

train_step (fork-join over the library's side streams, the fused tower kernels with their tag memsets, the embedding
backward's sort on a torch side stream) is captured into a torch.cuda.CUDAGraph the way bench.py captures it, then replayed
on two DIFFERENT batches copied into the static input tensors; loss, the flat dense gradient and the dense table gradient
must be bitwise equal to an eager step on the same inputs (the kernels are deterministic: fixed-order reductions, no float
atomics).  This is also the regression test for the capture abort of round 2 (gpurun_out/r2i.log: hipStreamEndCapture
recursed without end when two forked streams waited on each other; csrc/dense.hip keeps side2 waiting on side only): a
SIGSEGV inside capture_end fails this test's process."""
import numpy as np
import pytest
import torch

from oracle import aread_oracle as O
from tests.util import build_model, spec_full

pytestmark = pytest.mark.gpu


def _batch(spec, rng, B):
    x = np.stack([rng.integers(0, d, B) for d in spec.field_dims]
                 + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
    dom = rng.choice(spec.n_domain, B, p=[0.5, 0.3, 0.15, 0.05, 0.0][:spec.n_domain])
    dom[0] = 4                                   # a one-row segment; the replayed batches differ in their segment sizes
    x[:, spec.domain_idx] = dom
    return x, (rng.random(B) < 0.4).astype(np.float32)


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_graph_replay_equals_eager_step(precision):
    import aread_amd
    from aread_amd import _lib as L
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(77)
    B = 2500
    batches = [_batch(spec, rng, B) for _ in range(3)]
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, _ = build_model(spec, 123, precision=precision)
    model.train()
    model.drop_seed = 99                          # the seed is a launch argument: a captured graph replays it
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    stats0, nbt0 = model.bn_stats.clone(), model.bn_nbt.clone()
    lib = L.lib()

    def eager(x, y):
        model.bn_stats.copy_(stats0); model.bn_nbt.copy_(nbt0)
        bufs = model.make_step_buffers(B)
        loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md, set_grads=False)
        torch.cuda.synchronize()
        return (loss.clone(), bufs["gdense"].clone(), bufs["gtable"].clone(), bufs["probs"].clone(), model.bn_stats.clone())

    ref = [eager(x, y) for x, y in batches]

    xs = torch.from_numpy(batches[0][0]).cuda()
    ys = torch.from_numpy(batches[0][1]).cuda()
    bufs = model.make_step_buffers(B)
    step = lambda: model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
    step()                                        # sizes the embedding-backward workspace outside the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    n0 = lib.aread_debug_get(b"fused_fwd_calls")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                 # (round 2: Fatal Python error: Segmentation fault in capture_end)
        step()
    torch.cuda.synchronize()
    if precision == "bf16x3":
        assert lib.aread_debug_get(b"fused_fwd_calls") == n0 + 1, "the captured step did not take the fused tower kernels"
    for rep in range(2):                          # second round: the hand-off tags / workspaces are reused by later replays
        for (x, y), r in zip(batches, ref):
            xs.copy_(torch.from_numpy(x)); ys.copy_(torch.from_numpy(y))
            model.bn_stats.copy_(stats0); model.bn_nbt.copy_(nbt0)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(bufs["total"], r[0]), (rep, float(bufs["total"]), float(r[0]))
            assert torch.equal(bufs["gdense"], r[1]), rep
            assert torch.equal(bufs["gtable"], r[2]), rep
            assert torch.equal(bufs["probs"], r[3]), rep
            assert torch.equal(model.bn_stats, r[4]), rep


def test_graph_replay_draws_a_new_dropout_mask_with_the_device_seed():
    """ADVICE r1 / VERDICT r2: a host-side dropout seed is a kernel argument, so a captured step replays ONE dropout mask for
    ever.  With model.device_dropout_seed() the kernels read the seed from device memory and train_step advances it at its
    end inside the captured work: replay k must equal the eager step run with the k-th value of the sequence pinned."""
    import aread_amd
    spec = spec_full(dropout=0.2)
    rng = np.random.default_rng(5)
    B = 2000
    x, y = _batch(spec, rng, B)
    masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
    model, _ = build_model(spec, 321, precision="bf16x3")
    model.train()
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    stats0, nbt0 = model.bn_stats.clone(), model.bn_nbt.clone()
    xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    seed0 = 1234567
    step_inc = 0x9E3779B1

    def eager(seed):
        model.drop_seed_dev = None
        model.drop_seed = seed & 0xFFFFFFFF
        model.bn_stats.copy_(stats0); model.bn_nbt.copy_(nbt0)
        bufs = model.make_step_buffers(B)
        loss = model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
        torch.cuda.synchronize()
        return loss.clone(), bufs["gdense"].clone()

    ref = [eager(seed0 + k * step_inc) for k in range(3)]
    assert not torch.equal(ref[0][0], ref[1][0])                 # different seeds really give different masks
    model.drop_seed = None
    sd = model.device_dropout_seed(seed0)
    bufs = model.make_step_buffers(B)
    step = lambda: model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
    step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    torch.cuda.synchronize()
    sd.fill_(seed0)                                              # (warm-up and capture advanced it)
    for k in range(3):
        model.bn_stats.copy_(stats0); model.bn_nbt.copy_(nbt0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(bufs["total"], ref[k][0]), (k, float(bufs["total"]), float(ref[k][0]))
        assert torch.equal(bufs["gdense"], ref[k][1]), k
    want = (seed0 + 3 * step_inc) & 0xFFFFFFFF
    assert (int(sd.item()) & 0xFFFFFFFF) == want
