"""Which switch of the fused step changes its results bitwise?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import aread_amd
from tests import util as U

fn, mk, seed = U.GOLDEN_MODELS["full"]
G, spec = U.load_golden(fn), mk()
masks = U.golden_masks(spec, G, "rand")
x = torch.from_numpy(G["multi_rand/x"]).cuda()
y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
res = {}
for name, kw in [("base", dict(split_de=False, l2_dense_first=False, prepare_early=False)),
                 ("prepare", dict(split_de=False, l2_dense_first=False, prepare_early=True)),
                 ("dense_first", dict(split_de=False, l2_dense_first=True, prepare_early=True)),
                 ("split_de", dict(split_de=True, l2_dense_first=False, prepare_early=True)),
                 ("all", dict(split_de=True, l2_dense_first=True, prepare_early=True)),
                 ("all+prefetch", dict(split_de=True, l2_dense_first=True, prepare_early=True, prefetch=True))]:
    model, _ = U.build_model(spec, seed, precision=prec)
    model.train()
    pf = kw.pop("prefetch", False)
    for k, v in kw.items():
        setattr(model, k, v)
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    bufs = model.make_step_buffers(x.shape[0])
    pb = model.prepare_batch(x) if pf else None
    loss = model.train_step(x, y, bufs, masks_dev=md, set_grads=False, prepared=pb)
    torch.cuda.synchronize()
    res[name] = (float(loss), bufs["gdense"].clone(), bufs["gtable"].clone(), float(bufs["reg"][0]))
b = res["base"]
for name, r in res.items():
    dg = (r[1] != b[1]).nonzero().flatten()
    dt = (r[2] != b[2]).any(dim=1).nonzero().flatten()
    where = ""
    if len(dg):
        i = int(dg[0])
        t = [t_ for t_ in model._tensors if t_[1] == 0 and t_[2] <= i < t_[2] + max(1, int(np.prod(t_[3])))]
        where = f" first dense diff in {t[0][0] if t else i}: {float(r[1][i])!r} vs {float(b[1][i])!r}"
    print(f"{name:14s} loss {r[0]!r} reg {r[3]!r}  dense diffs {len(dg)}  table rows differing {len(dt)}{where}")
