"""smoke() inputs, all dense gradients vs the oracle, repeated (race hunting aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from oracle import aread_oracle as O
from tests.util import build_model, dense_grads, spec_full
spec = spec_full()
rng = np.random.default_rng(0)
B = 300
x = np.stack([rng.integers(0, d, B) for d in spec.field_dims] + [rng.integers(0, spec.field_dims[0] + 1, B) for _ in range(10)], axis=1).astype(np.int32)
y = (rng.random(B) < 0.5).astype(np.float32)
masks = [O.random_valid_mask(spec, rng, 0.6) for _ in range(spec.n_domain)]
r = None
prev = None
for rep in range(4):
    model, P = build_model(spec, 123, device="cuda:0"); model.train()
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda:0")
    bufs = model.make_step_buffers(B)
    loss = model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md)
    torch.cuda.synchronize()
    if r is None:
        r = O.step(P, spec, x, y, masks)
    g = dense_grads(model, bufs["gdense"])
    bad = []
    for name, ref in r["grads"].items():
        if name not in g: continue
        ref = ref.numpy(); d = np.abs(g[name] - ref).max(); m = np.abs(ref).max()
        if d > 1e-3 * max(m, 1e-6): bad.append((name, float(d), float(m)))
    print(rep, "loss", float(loss), r["loss"], "bad:", bad[:6], len(bad))
    flat = bufs["gdense"].clone()
    if prev is not None: print("   identical to previous run:", bool(torch.equal(prev, flat)))
    prev = flat
