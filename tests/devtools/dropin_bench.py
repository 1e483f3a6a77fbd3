"""Throughput of the DROP-IN path: the reference's own training-step code (run.py:668-682) -- model(X, mode=
'domain_mask_bagging', domain_i=d), per-head BCELoss, get_regularization_loss, loss.backward(), torch Adam -- on a
single-domain batch of 8192 Amazon-like samples, plus the eval path of Run.test."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import aread_oracle as O
from tools import synth
from tests.util import build_model

spec = O.amazon_spec(dropout=0.2)
rng = np.random.default_rng(0); mr = np.random.default_rng(2000)
masks = [O.random_valid_mask(spec, mr, 0.7) for _ in range(25)]
import aread_amd
for precision, opt_name in (("f32", "torch"), ("bf16x3", "torch"), ("bf16x3", "aread_amd.Adam")):
    model, P = build_model(spec, 123, precision=precision); model.train()
    model.domain_mask = [[torch.tensor(m, dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
    hyper = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    opt = torch.optim.Adam(model.parameters(), **hyper) if opt_name == "torch" else aread_amd.Adam(model, **hyper)
    crit = torch.nn.BCELoss()
    batches = []
    for d in (3, 6, 12):
        x, y = synth.amazon_batch(spec, rng, 8192, domain=d)
        batches.append((d, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))

    def train_step(i):
        d, X, y = batches[i % 3]
        preds = model(X, mode="domain_mask_bagging", domain_i=d)
        loss = sum(crit(p, y) for p in preds.unbind(0)) / preds.shape[0] + model.get_regularization_loss(device="cuda")
        model.zero_grad(); loss.backward(); opt.step()
        return loss

    for i in range(5): train_step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 30
    for i in range(N): l = train_step(i)
    float(l); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    print(f"[{precision}, {opt_name}] drop-in train step (fwd + loss + bwd + Adam over table and {len(list(model.parameters()))} tensors): "
          f"{dt*1e3:.2f} ms/step = {8192/dt/1e6:.2f} M samples/s")
    model.eval()
    with torch.no_grad():
        for i in range(5): model(batches[0][1], mode="domain_with_mask", domain_i=3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(N): p = model(batches[i % 3][1], mode="domain_with_mask", domain_i=batches[i % 3][0])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    print(f"[{precision}] drop-in eval forward: {dt*1e3:.3f} ms/batch = {8192/dt/1e6:.2f} M samples/s")
    del model, opt
    torch.cuda.empty_cache()
