"""Phase anatomy of the fused tower forward kernel: s_memrealtime stamps (100 MHz) per tile -> median / max microseconds."""
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
L.check(L.lib().aread_debug_set(b"fused_towers", 1))
L.check(L.lib().aread_debug_set(b"tf_stamps", 1))
for _ in range(3):
    st, _ = model._run(xs, 0, 25, None, md, False, y=ys, loss_out=bufs["loss"], ws=bufs["ws"], probs=bufs["probs"], e=bufs["e"])
torch.cuda.synchronize()
off = L.lib().aread_debug_ws_offset(model._handle, 8192, 25, b"misc_part")
nt = int(st.plan.header[3])
raw = bufs["ws"].view(torch.float32)[off:off + nt * 128].view(torch.int64).view(nt, 64).cpu().numpy()
names = ["start"]
for l in range(3):
    names += [f"l{l} gates", f"l{l} mix"]
    for j in range(2):
        names += [f"l{l}.{j} mfma", f"l{l}.{j} stats+arrive", f"l{l}.{j} H+poll", f"l{l}.{j} merge", f"l{l}.{j} norm"]
names += ["heads"]
t = (raw[:, :len(names)] - raw[:, :1]) / 100.0          # us since the tile's own start
d = np.diff(t, axis=1)
print(f"{nt} tiles; kernel span (first start -> last end): {(raw[:, len(names) - 1].max() - raw[:, 0].min()) / 100.0:.1f} us; "
      f"start skew {(raw[:, 0].max() - raw[:, 0].min()) / 100.0:.1f} us")
for i, n in enumerate(names[1:]):
    print(f"{n:22s} median {np.median(d[:, i]):6.2f}  max {d[:, i].max():6.2f}   (cumulative median {np.median(t[:, i + 1]):7.2f})")
