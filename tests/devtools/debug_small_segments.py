"""Are tiny BatchNorm segments (2-3 rows) handled correctly?  HIP fp32 vs oracle fp32 vs oracle fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import aread_amd
from oracle import aread_oracle as O
from tests import util as U
spec = U.spec_full()
rng = np.random.default_rng(3)
sizes = [2, 3, 60, 100, 75]
doms = np.concatenate([np.full(s, d) for d, s in enumerate(sizes)]); rng.shuffle(doms)
B = len(doms)
x = np.stack([rng.integers(0, d, B) for d in spec.field_dims] + [rng.integers(0, 41, B) for _ in range(10)], axis=1).astype(np.int32)
x[:, spec.domain_idx] = doms
y = (rng.random(B) < 0.5).astype(np.float32)
masks = [O.full_mask(spec) for _ in range(5)]
P = O.init_params(spec, 123)
r32 = O.step(P, spec, x, y, masks)
P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
r64 = O.step(P64, spec, x, y.astype(np.float64), masks)
model, _ = U.build_model(spec, 123); model.train()
model.domain_mask = [[torch.tensor(m, dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
bufs = model.make_step_buffers(B)
loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs)
g = U.dense_grads(model)
print("loss hip", float(loss), "o32", r32["loss"], "o64", r64["loss"])
for n in ("mmoe_experts.2.layers.4.weight", "mmoe_experts.0.layers.0.weight", "towers.0.1.layers.0.weight", "linear.fc.weight", "towers_linear.3.weight"):
    t = r64["grads"][n].numpy(); a = g[n]; b = r32["grads"][n].numpy()
    print(f"{n:36s} |g64|max {np.abs(t).max():.3e}  hip-g64 {np.abs(a - t).max():.3e}  o32-g64 {np.abs(b - t).max():.3e}")
pl = bufs["probs"].cpu().numpy(); ok = ~np.isnan(r64["probs"])
for d, s in enumerate(sizes):
    sel = np.broadcast_to((doms == d)[None, :], ok.shape) & ok
    print(f"domain {d} n={s}: prob err hip-o64 {np.abs(pl[sel] - r64['probs'][sel]).max():.2e}  o32-o64 {np.abs(r32['probs'][sel] - r64['probs'][sel]).max():.2e}")
