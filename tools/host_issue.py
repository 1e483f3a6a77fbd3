"""How long does the host take to ENQUEUE one eager step (no sync)?  If close to the step time we are launch-bound."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import presets
from tools import synth
spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
model = presets.build_model(spec, "cuda", precision=os.environ.get("PRECISION", "bf16x3")); model.train()
masks = presets.random_masks(model, 0.7, seed=2000)
md = aread_amd.pack_masks(masks, 25, model.edge_num, "cuda")
x, y = synth.amazon_batch(spec, rng, 8192)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
bufs = model.make_step_buffers(8192)
for _ in range(5): model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/N:.3f} ms/step ; total {1e3*(t2-t0)/N:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
