"""Phase anatomy of the fused tower BACKWARD kernel: s_memrealtime stamps (100 MHz) per tile -> median / max microseconds."""
import os, sys
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
# `one` on the command line: every sample in one domain (the reference's per-domain batches: one segment of 128 tiles)
x, y = synth.amazon_batch(spec, rng, 8192, domain=3) if "one" in sys.argv[1:] else synth.amazon_batch(spec, rng, 8192)
xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
bufs = model.make_step_buffers(8192)
L.check(L.lib().aread_debug_set(b"tf_stamps", 2))
for _ in range(3):
    model.train_step(xs, ys, bufs, masks_dev=md, set_grads=False)
torch.cuda.synchronize()
st, _ = model._last
off = L.lib().aread_debug_ws_offset(model._handle, 8192, 25, b"gate_part")
nt = int(st.plan.header[3])
raw = st.ws.view(torch.float32)[off:off + nt * 128].view(torch.int64).view(nt, 64).cpu().numpy()
names = ["start", "heads"]
for l in (2, 1, 0):
    for j in (1, 0):
        names += [f"l{l}.{j} act+arrive", f"l{l}.{j} poll", f"l{l}.{j} merge", f"l{l}.{j} apply", f"l{l}.{j} img+dgrad"]
    names += [f"l{l} mix: stage", f"l{l} mix: dots", f"l{l} mix: gates", f"l{l} mix: d_src"]
t = (raw[:, :len(names)] - raw[:, :1]) / 100.0
d = np.diff(t, axis=1)
print(f"{nt} tiles; kernel span (first start -> last end): {(raw[:, len(names) - 1].max() - raw[:, 0].min()) / 100.0:.1f} us; "
      f"start skew {(raw[:, 0].max() - raw[:, 0].min()) / 100.0:.1f} us")
for i, n in enumerate(names[1:]):
    print(f"{n:22s} median {np.median(d[:, i]):6.2f}  max {d[:, i].max():6.2f}   (cumulative median {np.median(t[:, i + 1]):7.2f})")
