"""Host profile of the drop-in training step (run.py:668-682 on the nn.Module API, the loop bench.py's `dropin_step` times):
cProfile of 40 steps per optimizer + the step time with the host synchronised / running ahead.  usage: dropin_profile.py [steps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import aread_amd
from aread_amd import presets
from tools import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
B = 8192
spec = presets.amazon_workload(0.2)
rng = np.random.default_rng(0)
model = presets.build_model(spec, "cuda", precision="bf16x3"); model.train()
masks = presets.random_masks(model, 0.7, seed=2000)
model.domain_mask = [[m if isinstance(m, torch.Tensor) else torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk] for mk in masks]
batches = []
for d in (3, 6, 12):
    x, y = synth.amazon_batch(spec, rng, B, domain=d)
    batches.append((d, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))
# wall-clock timers around the pieces of the two autograd nodes (they run on the autograd engine's thread: cProfile does not see them)
from aread_amd import aread as _A
_acc = {}
def _timed(owner, name, label):
    fn = getattr(owner, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            _acc[label] = _acc.get(label, 0.0) + time.perf_counter() - t
    setattr(owner, name, staticmethod(w) if isinstance(owner, type) and name == "backward" else w)
_timed(_A._AreadFn, "backward", "_AreadFn.backward")
_timed(_A._RegFn, "backward", "_RegFn.backward")
_timed(model, "_accumulate_dense", "  _accumulate_dense (both)")
_timed(model.embedding, "scatter_grad", "  embedding.scatter_grad")
_lib_bwd = _A.L.lib().aread_backward
crit = torch.nn.BCELoss()
hyper = dict(lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
for name in (("aread_amd.Adam",) if "ours" in sys.argv[1:] else ("aread_amd.Adam", "torch.optim.Adam")):
    opt = torch.optim.Adam(model.parameters(), **hyper) if name == "torch.optim.Adam" else aread_amd.Adam(model, **hyper)
    T = {}

    def one(i, split=False):
        d, X, y = batches[i % 3]
        t = [time.perf_counter()]
        preds = model(X, mode="domain_mask_bagging", domain_i=d); t.append(time.perf_counter())
        loss = sum(crit(p, y) for p in preds.unbind(0)) / preds.shape[0]; t.append(time.perf_counter())
        loss = loss + model.get_regularization_loss(device="cuda"); t.append(time.perf_counter())
        model.zero_grad(); t.append(time.perf_counter())
        loss.backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        if split:
            for k, n in enumerate(["forward", "BCE per head", "reg loss", "zero_grad", "backward", "optimizer.step"]):
                T[n] = T.get(n, 0.0) + t[k + 1] - t[k]
        return loss
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = one(i, True)
    t_host = (time.perf_counter() - t0) / steps
    float(last); torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps
    # host time of one step with an EMPTY queue in front of it (no back-pressure from the GPU), and the GPU time of that step alone
    th, tg = 0.0, 0.0
    for i in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter(); one(i); t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        th += t2 - t1; tg += t3 - t1
    print(f"== {name}: {t_all * 1e3:.3f} ms/step (host enqueue {t_host * 1e3:.3f} ms/step); one step from an idle GPU: host {th / 10 * 1e3:.3f} ms, "
          f"until the GPU is done {tg / 10 * 1e3:.3f} ms")
    for n, v in T.items():
        print(f"   host {n:16s} {v / steps * 1e3:7.3f} ms")
    for n, v in _acc.items():
        print(f"        {n:28s} {v / (steps + 5) * 1e3:7.3f} ms")
    _acc.clear()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        one(i)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
    del opt
    model.zero_grad(set_to_none=True)
