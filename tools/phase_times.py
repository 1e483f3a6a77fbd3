"""GPU-side anatomy of one eager fused step from events on the main stream (no profiler in the way)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import _lib as L
from aread_amd import presets
from tools import synth

spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
model = presets.build_model(spec, "cuda", precision="bf16x3"); model.train()
masks = presets.random_masks(model, 0.7, seed=2000)
md = aread_amd.pack_masks(masks, 25, model.edge_num, "cuda")
if os.environ.get("FUSED") == "1":
    L.check(L.lib().aread_debug_set(b"fused_towers", 1))
x, y = synth.amazon_batch(spec, rng, 8192)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
bufs = model.make_step_buffers(8192)
L.check(L.lib().aread_debug_set(b"phase_events", 1))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
acc = np.zeros(10); tot = []
N = 30
for i in range(N + 5):
    ev[0].record()
    model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
    ev[1].record()
    torch.cuda.synchronize()
    out = (C.c_float * 16)()
    L.check(L.lib().aread_debug_phase_times(out, 16))
    if i >= 5:
        acc += np.maximum(np.array(out[:10]), 0) * 1e3
        tot.append(ev[0].elapsed_time(ev[1]) * 1e3)
names = ["experts fwd", "MMoE mix + towers + heads fwd", "(fwd tail -> bwd start)", "grads memset + dcn GEMM", "heads + towers + mixes bwd",
         "side batch A fork (gate GEMMs)", "experts bwd", "-", "-", "-"]
print(f"step (event to event, synced each step): {np.mean(tot):.1f} us")
for n, v in zip(names, acc / N):
    print(f"  {n:28s} {v:8.1f} us")
print(f"  sum of the phases above      {np.sum(acc[:8]) / N:8.1f} us")
