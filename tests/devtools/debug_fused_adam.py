import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util as U
import tests.test_gpu_optim as T
import aread_amd
HYPER = T.HYPER
fn, mk, seed = U.GOLDEN_MODELS["full"]
G, spec = U.load_golden(fn), mk()
masks = U.golden_masks(spec, G, "rand")
x0 = torch.from_numpy(G["multi_rand/x"]).cuda(); y0 = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
batches = [(x0, y0)]
for k in (1, 2):
    perm = torch.from_numpy(np.random.default_rng(k).permutation(x0.shape[0])).cuda()
    xk = x0[perm].clone(); xk[:, 0] = (xk[:, 0] + 7 * k) % spec.field_dims[0]
    batches.append((xk.contiguous(), (1.0 - y0[perm]).contiguous()))
def fresh():
    model, _ = U.build_model(spec, seed, dropout=0.2); model.train(); model.drop_seed = 99
    model.domain_mask = [[torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk_] for mk_ in masks]
    return model
res = []
for rep in range(4):
    a = fresh(); md = aread_amd.pack_masks(masks, spec.n_domain, a.edge_num, "cuda")
    opt = torch.optim.Adam(a.parameters(), **HYPER); bufs = a.make_step_buffers(x0.shape[0])
    for x, y in batches:
        a.zero_grad(set_to_none=True); a.train_step(x, y, bufs, masks_dev=md); opt.step()
    b = fresh(); fused = aread_amd.FusedAdam(b, x0.shape[0], **HYPER)
    for x, y in batches: fused.step(x, y, md)
    torch.cuda.synchronize()
    wa, wb = a.embedding.embedding_dict.weight.data, b.embedding.embedding_dict.weight.data
    d = (wa - wb).abs()
    i = int(d.argmax()) // wa.shape[1]
    res.append((wa.clone(), wb.clone(), a.dense.data.clone(), b.dense.data.clone()))
    dd = (a.dense.data - b.dense.data).abs()
    print("   dense max %.3e mean %.3e n>5e-5: %d" % (float(dd.max()), float(dd.mean()), int((dd > 5e-5).sum())))
    print(rep, float(d.max()), float(d.mean()), "row", i, "n>1e-5:", int((d > 1e-5).sum()))
print("a reproducible:", all(torch.equal(res[0][0], r[0]) and torch.equal(res[0][2], r[2]) for r in res),
      " b reproducible:", all(torch.equal(res[0][1], r[1]) and torch.equal(res[0][3], r[3]) for r in res))
