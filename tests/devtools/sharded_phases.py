"""Phase timing of ShardedTableStep on one GPU (host clock with synchronisation after each phase)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import aread_oracle as O
from tools import synth
from tests.util import build_model
import aread_amd
import aread_amd.dist as D
from aread_amd.plan import RowPlan

spec = O.amazon_spec(dropout=0.2)
dev = torch.device("cuda")
model, P = build_model(spec, 123, device=dev, precision="bf16x3")
model.train()
rng = np.random.default_rng(2000)
masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, dev)
x, y = synth.amazon_batch(spec, rng, 8192)
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
sh = D.ShardedTableStep(model, 8192)
for _ in range(3):
    sh.step(x, y, md)
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6, r


emb = model.embedding
t, plan = timed(lambda: RowPlan(x, model.domain_idx, model.n_domain)); print(f"plan          {t:8.1f} us")
t, bag = timed(lambda: x + emb._offsets_dev(dev)); print(f"bag           {t:8.1f} us")
g = bag.reshape(-1).to(torch.int64)
t, key = timed(lambda: (g % 1) * sh.router.rows_per_rank + g // 1); print(f"key           {t:8.1f} us")
t, _ = timed(lambda: torch.unique(key, return_inverse=True)); print(f"torch.unique  {t:8.1f} us")
t, route = timed(lambda: sh.router.route(bag)); print(f"route (all)   {t:8.1f} us   unique={route.n_unique}")
t, rows = timed(lambda: sh._gather_owned(route.recv_rows)); print(f"gather owned  {t:8.1f} us")
t, _ = timed(lambda: sh.lookup(x, plan)); print(f"lookup (all)  {t:8.1f} us")
t, _ = timed(lambda: sh.presort(x, route, plan)); print(f"presort       {t:8.1f} us")
t, _ = timed(lambda: sh.table_grad(x, route, plan)); print(f"table_grad    {t:8.1f} us")
t, _ = timed(lambda: sh.step(x, y, md)); print(f"step          {t:8.1f} us")
bufs = model.make_step_buffers(8192)
t, _ = timed(lambda: model.train_step(x, y, bufs, masks_dev=md, set_grads=False)); print(f"fused step    {t:8.1f} us")
