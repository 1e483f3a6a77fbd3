"""Anatomy of the fused step as bench.py runs it (eager launches, the host running ahead of the GPU: NO synchronisation
between the steps): timing events at the step's host-level boundaries on the main stream plus the library's phase events,
averaged over the steps.  usage: python tools/step_anatomy.py [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import _lib as L
from aread_amd import presets
from tools import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
model = presets.build_model(spec, "cuda", precision="bf16x3"); model.train()
masks = presets.random_masks(model, 0.7, seed=2000)
md = aread_amd.pack_masks(masks, 25, model.edge_num, "cuda")
batches = []
for _ in range(8):
    x, y = synth.amazon_batch(spec, rng, 8192)
    batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))
bufs = model.make_step_buffers(8192)
for i in range(5):
    model.train_step(*batches[i % 8], bufs, masks_dev=md, set_grads=False)
torch.cuda.synchronize()
all_marks = []
prefetch = os.environ.get("PREFETCH", "1") == "1"
cur, spare = None, None
for i in range(N):
    model._marks = []
    xb, yb = batches[i % 8]
    if prefetch:
        if cur is None:
            cur = model.prepare_batch(xb)
        nxt = model.prepare_batch(batches[(i + 1) % 8][0], reuse=spare)
        model.train_step(xb, yb, bufs, masks_dev=md, set_grads=False, prepared=cur)
        spare, cur = cur, nxt
    else:
        model.train_step(xb, yb, bufs, masks_dev=md, set_grads=False)
    all_marks.append(model._marks)
model._marks = None
torch.cuda.synchronize()
names = [n for n, _ in all_marks[0]]
acc = np.zeros(len(names))
for k in range(1, N):                                  # interval ending at mark j of step k (mark 0: since the previous step's last mark)
    prev = all_marks[k - 1][-1][1]
    for j, (n, ev) in enumerate(all_marks[k]):
        acc[j] += prev.elapsed_time(ev) * 1e3
        prev = ev
acc /= (N - 1)
print(f"eager step, host running ahead, {N - 1} steps: {acc.sum():.1f} us per step (events on the main stream)")
for n, v in zip(names, acc):
    print(f"  -> {n:52s} {v:8.1f} us")
