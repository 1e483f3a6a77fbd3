"""cProfile of the drop-in training step (host side)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import aread_oracle as O
from tools import synth
from tests.util import build_model
import aread_amd
spec = O.amazon_spec(dropout=0.2)
rng = np.random.default_rng(0); mr = np.random.default_rng(2000)
masks = [O.random_valid_mask(spec, mr, 0.7) for _ in range(25)]
model, P = build_model(spec, 123, precision="bf16x3"); model.train()
model.domain_mask = [[torch.tensor(m, dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
opt = aread_amd.Adam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
crit = torch.nn.BCELoss()
x, y = synth.amazon_batch(spec, rng, 8192, domain=3)
X, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
def step():
    preds = model(X, mode="domain_mask_bagging", domain_i=3)
    loss = sum(crit(p, y) for p in preds.unbind(0)) / preds.shape[0] + model.get_regularization_loss(device="cuda")
    model.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
