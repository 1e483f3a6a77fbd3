"""Which part of the step breaks hipGraph capture?  Captures progressively larger pieces."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import _lib as L
from oracle import aread_oracle as O
from tools import synth
from tests.util import build_model

spec = O.amazon_spec(dropout=0.2)
rng = np.random.default_rng(0); mr = np.random.default_rng(2000)
masks = [O.random_valid_mask(spec, mr, 0.7) for _ in range(25)]
model, P = build_model(spec, 123, precision="bf16x3"); model.train()
md = aread_amd.pack_masks(masks, 25, model.edge_num, "cuda")
x, y = synth.amazon_batch(spec, rng, 8192)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
bufs = model.make_step_buffers(8192)
model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
torch.cuda.synchronize()


def capture(name, fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", name, flush=True)
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("  ok", name, flush=True)


emb = model.embedding
from aread_amd.plan import RowPlan
plan = RowPlan(xs, model.domain_idx, 25)
if os.environ.get("CAP_ALL"): capture("plan", lambda: RowPlan(xs, model.domain_idx, 25))
if os.environ.get("CAP_ALL"): capture("sort", lambda: emb.sort_lookups(xs, plan.sample_row))
if os.environ.get("CAP_ALL"): capture("sort+reduce", lambda: (emb.sort_lookups(xs, plan.sample_row), emb.reduce_sorted(xs, bufs["de"], bufs["gtable"])))
if os.environ.get("CAP_ALL"): capture("forward", lambda: model._run(xs, 0, 25, None, md, False, y=ys, loss_out=bufs["loss"], ws=bufs["ws"], probs=bufs["probs"], e=bufs["e"]))
capture("step_local", lambda: model.step_local(xs, ys, bufs, md, presort=False) and model.step_finish(bufs))
capture("train_step", lambda: model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False))
