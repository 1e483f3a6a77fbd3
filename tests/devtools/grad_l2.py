"""Per-tensor relative L2 error of the gradients: (a) golden multi-domain step (reference fixtures), (b) BASELINE-size
proportional batch restricted to domains with >= 8 rows, against an fp64 run of the oracle.  Both GEMM precisions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import aread_amd
from oracle import aread_oracle as O
from tests import util as U
from tools import synth

tmask = lambda mk: [torch.tensor(np.asarray(m), dtype=torch.bool, device="cuda") for m in mk]


def pre_bn_bias(name):
    """Linear bias in front of a BatchNorm (layers.{0,4,8}.bias): its true gradient is exactly zero, what any implementation
    computes is rounding noise -- no relative error to speak of"""
    k = name.split(".")
    return k[-1] == "bias" and "layers" in k and int(k[k.index("layers") + 1]) % 4 == 0


def show(tag, e):
    v = np.array([x for n, x in e.items() if not pre_bn_bias(n)])
    worst = sorted(((n, x) for n, x in e.items() if not pre_bn_bias(n)), key=lambda kv: -kv[1])[:4]
    print(f"{tag}: n={len(v)} median {np.median(v):.2e} p90 {np.quantile(v, 0.9):.2e} max {v.max():.2e}  worst {[(k, f'{x:.1e}') for k, x in worst]}", flush=True)


def all_grads(model):
    g = U.dense_grads(model)
    g["embedding.embedding_dict.weight"] = model.embedding.embedding_dict.weight.grad.cpu().numpy()
    return g


def golden_case(which, precision):
    fn, mk, seed = U.GOLDEN_MODELS[which]
    G, spec = U.load_golden(fn), mk()
    model, _ = U.build_model(spec, seed, precision=precision); model.train()
    model.domain_mask = [tmask(m) for m in U.golden_masks(spec, G, "rand")]
    x = torch.from_numpy(G["multi_rand/x"]).cuda(); y = torch.from_numpy(G["multi_rand/y"].astype(np.float32)).cuda()
    bufs = model.make_step_buffers(x.shape[0], multi_domain=True)
    model.train_step(x, y, bufs)
    return U.rel_l2_vs_golden(G, "multi_rand/grad", all_grads(model))


def baseline_case(precision, r64, x, y, masks, spec):
    model, _ = U.build_model(spec, 123, precision=precision); model.train()
    md = aread_amd.pack_masks(masks, spec.n_domain, model.edge_num, "cuda")
    model.domain_mask = [tmask(m) for m in masks]
    bufs = model.make_step_buffers(x.shape[0])
    model.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), bufs, masks_dev=md)
    g = all_grads(model)
    out = {}
    for n, ref in r64["grads"].items():
        ref = ref.numpy().astype(np.float64)
        if n not in g or np.linalg.norm(ref) == 0:
            continue
        out[n] = float(np.linalg.norm(g[n].astype(np.float64) - ref) / np.linalg.norm(ref))
    return out


if __name__ == "__main__":
    for which in ("full", "tiny"):
        for prec in ("f32", "bf16x3"):
            e = golden_case(which, prec)
            show(f"golden {which:5s} {prec:7s}", e)
    spec = O.amazon_spec(dropout=0.0)
    rng = np.random.default_rng(2000)
    masks = [O.random_valid_mask(spec, rng, 0.7) for _ in range(spec.n_domain)]
    x, y = synth.amazon_batch(spec, rng, 8192, domain="proportional")
    cnt = np.bincount(x[:, spec.domain_idx], minlength=spec.n_domain)
    keep = cnt[x[:, spec.domain_idx]] >= 8
    x, y = x[keep], y[keep]
    print("rows kept", keep.sum(), "of 8192; domains kept", int((cnt >= 8).sum()), flush=True)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    P = O.init_params(spec, 123)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    r64 = O.step(P64, spec, x, y.astype(np.float64), masks)
    r32 = O.step(P, spec, x, y, masks)
    e32 = {n: float(np.linalg.norm(r32["grads"][n].numpy().astype(np.float64) - r64["grads"][n].numpy()) / max(np.linalg.norm(r64["grads"][n].numpy()), 1e-300))
           for n in r64["grads"] if np.linalg.norm(r64["grads"][n].numpy()) > 0}
    show("oracle fp32 vs fp64", e32)
    for prec in ("f32", "bf16x3"):
        e = baseline_case(prec, r64, x, y, masks, spec)
        show(f"baseline-size {prec:7s} vs fp64", e)
