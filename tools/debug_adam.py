"""Two Adam(lr=1e-2) steps on a masked single-domain batch: HIP model vs oracle (torch autograd, per-tensor params)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import aread_oracle as O
from tests import util as U
spec = U.spec_full()
G = U.load_golden("aread_full.npz")
masks = U.golden_masks(spec, G, "sparse")
p = "single_sparse"
d = int(G[f"{p}/domain"]); x = G[f"{p}/x"]; y = G[f"{p}/y"].astype(np.float32)
model, P = U.build_model(spec, 123)
model.train()
tm = [torch.tensor(m, dtype=torch.bool, device="cuda") for m in masks[d]]
opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
# oracle side
names = O.trainable_names(spec)
leaves = {n: torch.nn.Parameter(P[n].clone()) for n in names}
Pw = dict(P); Pw.update(leaves)
bufs = O.split_buffers(P)
oopt = torch.optim.Adam(list(leaves.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
crit = torch.nn.BCELoss()
xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for it in range(3):
    preds = model(xt, mode="domain_mask_bagging", domain_i=d, current_mask=tm)
    loss = sum(crit(pr, yt) for pr in preds.unbind(0)) / preds.shape[0] + model.get_regularization_loss()
    model.zero_grad(); loss.backward(); opt.step()
    r = O.forward(Pw, bufs, spec, x, mode="domain_mask_bagging", mask=masks[d], train=True)
    bufs = r["buffers"]
    ol = O.bagging_loss(r["probs"], torch.from_numpy(y)) + O.reg_loss(Pw, spec)
    oopt.zero_grad(); ol.backward(); oopt.step()
    print(f"step {it}: loss mine {float(loss):.7f} oracle {float(ol):.7f}")
    sd = model.state_dict()
    worst = []
    for n in names:
        a = sd[n].detach().cpu().numpy(); b = leaves[n].detach().numpy()
        worst.append((np.abs(a - b).max(), n, leaves[n].grad is None))
    worst.sort(reverse=True)
    for w in worst[:8]:
        print(f"   {w[0]:.3e} {w[1]} grad_none_in_oracle={w[2]}")
    # None-ness agreement
    mine_none = {n: p.grad is None for n, p in model.named_dense_parameters()}
    bad = [n for n in names if n in mine_none and mine_none[n] != (leaves[n].grad is None)]
    print("   None-mismatch:", bad[:10], len(bad))
