"""Shared helpers for the tests: golden loading, spec factories, comparison of packed gradients."""
import os

import numpy as np

from oracle import aread_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GRAD_SAMPLE_STRIDE = 37


def spec_full(**kw):
    base = dict(field_dims=[40, 7, 5, 9, 11, 30, 10], embed_dim=32, multi_hot_flag=[False] * 7 + [True] * 10,
                itemid_idx=0, seq_maxlen=5, method="mean", n_tower=(3, 6, 12), n_domain=5, domain_idx=2)
    base.update(kw)
    return O.Spec(**base)


def spec_tiny(**kw):
    base = dict(field_dims=[23, 4, 6, 3, 17], embed_dim=16, multi_hot_flag=[False] * 5, itemid_idx=0,
                method=None, n_tower=(2, 3, 5), n_domain=4, domain_idx=1, n_expert=3,
                expert_dims=(40, 24, 12), tower_dims=((12, 8), (8, 8), (8, 4)), n_cross=2, atten_embed_dim=64)
    base.update(kw)
    return O.Spec(**base)


GOLDEN_MODELS = {"full": ("aread_full.npz", spec_full, 123), "tiny": ("aread_tiny.npz", spec_tiny, 321)}

_cache = {}


def load_golden(name):
    if name not in _cache:
        with np.load(os.path.join(GOLDEN, name)) as z:
            _cache[name] = {k: z[k] for k in z.files}
    return _cache[name]


def golden_masks(spec, G, which):
    return [O.unpack_mask(spec, row) for row in G[f"masks/{which}"]]


def check_grads(G, prefix, grads, rtol=2e-4, atol_scale=2e-5, skip=(), floor=2e-8):
    """Compare a dict name->array with packed golden gradients (full / strided sample + sums)."""
    n = 0
    for key in G:
        if not key.startswith(prefix + "/full/") and not key.startswith(prefix + "/samp/"):
            continue
        name = key[len(prefix) + 6:]
        if name in skip or name not in grads:
            continue
        got = np.asarray(grads[name], dtype=np.float32).reshape(-1)
        ref = G[key]
        if key.startswith(prefix + "/samp/"):
            sums = G[f"{prefix}/sum/{name}"]
            scale = max(float(sums[1]) / got.size, 1e-12)
            np.testing.assert_allclose(got[::GRAD_SAMPLE_STRIDE], ref, rtol=rtol, atol=max(floor, atol_scale * max(scale, np.abs(ref).max())),
                                       err_msg=name)
            assert abs(got.astype(np.float64).sum() - sums[0]) <= 1e-3 * max(sums[1], 1e-9), name
        else:
            atol = max(floor, atol_scale * float(np.abs(ref).max()))  # pre-BN bias grads are ~1e-10 noise
            np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol, err_msg=name)
        n += 1
    assert n > 0, f"no golden gradients under {prefix}"
    return n


def rel_l2_vs_golden(G, prefix, grads, skip=()):
    """name -> ||got - ref||_2 / ||ref||_2 against packed golden gradients (full tensors, or their strided samples)."""
    out = {}
    for key in G:
        full, samp = key.startswith(prefix + "/full/"), key.startswith(prefix + "/samp/")
        if not (full or samp):
            continue
        name = key[len(prefix) + 6:]
        if name in skip or name not in grads:
            continue
        got = np.asarray(grads[name], dtype=np.float64).reshape(-1)
        ref = np.asarray(G[key], dtype=np.float64).reshape(-1)
        if samp:
            got = got[::GRAD_SAMPLE_STRIDE]
        nrm = np.linalg.norm(ref)
        if nrm > 0:
            out[name] = float(np.linalg.norm(got - ref) / nrm)
    return out


# ---- helpers for the HIP-side model (GPU tests, smoke, bench) ---------------------------------------
def model_config(spec, precision="f32"):
    import types
    cfg = types.SimpleNamespace()
    cfg.aread_precision = precision
    cfg.dataset_name = "synthetic"
    cfg.domain_size = {"synthetic": [100 + d for d in range(spec.n_domain)]}
    cfg.use_dcn, cfg.use_atten = True, bool(spec.with_dead_attention)
    cfg.n_cross_layers, cfg.mmoe_n_expert = spec.n_cross, spec.n_expert
    cfg.atten_embed_dim, cfg.att_layer_num, cfg.att_head_num, cfg.att_res = spec.atten_embed_dim, spec.att_layer_num, 2, True
    return cfg


def build_model(spec, seed, device="cuda", dropout=None, precision="f32"):
    """aread_amd.AREAD with the parameters of oracle.init_params(spec, seed) (strict state_dict load)."""
    import aread_amd
    mh = {"multi_hot_flag": list(spec.multi_hot_flag), "itemid_idx": spec.itemid_idx, "seq_maxlen": spec.seq_maxlen,
          "method": spec.method}
    model = aread_amd.AREAD(list(spec.field_dims), spec.embed_dim, mh, tuple(spec.n_tower), spec.n_domain, "mmoe",
                            tuple(spec.expert_dims), tuple(tuple(t) for t in spec.tower_dims), spec.domain_idx,
                            n_cross_layers=spec.n_cross, dropout=spec.dropout if dropout is None else dropout,
                            device=device, l2_reg_embedding=spec.l2_embedding, l2_reg_linear=spec.l2_linear,
                            l2_reg_dnn=spec.l2_dnn, l2_reg_cross=spec.l2_cross, config=model_config(spec, precision))
    P = O.init_params(spec, seed)
    model.load_state_dict(P, strict=True)
    return model.to(device), P


def dense_grads(model, flat=None):
    """name -> numpy gradient for every tensor that lives in the flat dense parameter."""
    if flat is not None:
        g = flat.detach().cpu().numpy()
        out = {}
        for name, kind, off, shape, _ in model._tensors:
            if kind == 0:
                n = int(np.prod(shape)) if shape else 1
                out[name] = g[off:off + n].reshape(shape)
        return out
    return {name: (p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
            for name, p in model.named_dense_parameters()}


# ---- HEMP host-logic sequence: driven identically on the reference (golden generation) and on aread_amd ----
def hemp_sequence(model, spec, seed=7):
    """Run a fixed sequence of mask operations under fixed numpy/torch seeds; returns {name: uint8 array}.
    Inputs that are not part of the model's own random stream come from a private Generator."""
    import contextlib, io
    import torch
    out = {}
    rng = np.random.default_rng(seed + 100)
    np.random.seed(seed)
    torch.manual_seed(seed)
    pk = lambda m: O.pack_mask(spec, m)
    n = spec.n_tower
    with contextlib.redirect_stdout(io.StringIO()):
        model.reset_for_mask_update()
        for i, p in enumerate([0.7, 0.7, 0.5, 0.3, 0.3, 0.15]):
            out[f"rand/{i}"] = pk(model.generate_mask("rand", init_active_percent=p))
        for i in range(24):                                           # validity closure on raw random masks
            p = [0.15, 0.3, 0.5, 0.8][i % 4]
            raw = [rng.random(s.shape) < p for s in O.full_mask(spec)]
            m = [r.copy() for r in raw] if i % 2 == 0 else [torch.tensor(r) for r in raw]
            out[f"validate/{i}/in"] = pk(raw)
            out[f"validate/{i}/out"] = pk(model.validate_mask(m))
        # gate statistics recorded by warm-up steps -> 'mask_max_gate' candidates (run.py:628-630)
        for d in range(3):
            for l in range(1, spec.n_level):
                for t in range(n[l]):
                    for _ in range(3):
                        g = rng.random(n[l - 1]).astype(np.float32)
                        g = g / g.sum() * (rng.random() < 0.85)
                        model.domain_tower_gate_values[d][l][t].append(torch.from_numpy(g.astype(np.float32)))
        for d in range(3):
            m1 = model.generate_mask("mask_max_gate", d=d, init_active_percent=0.7, random_modify_sigma=0.2)
            out[f"mmg/{d}/first"] = pk(m1)
            model.domain_mask[d] = m1
            out[f"mmg/{d}/second"] = pk(model.generate_mask("mask_max_gate", d=d, init_active_percent=0.4,
                                                             random_modify_sigma=0.3))
            out[f"mnr/{d}"] = pk(model.generate_mask("mask_norm_rand", d=d, random_modify_sigma=0.25))
            out[f"mgnr/{d}"] = pk(model.generate_mask("max_gate_norm_rand", d=d, init_active_percent=0.6,
                                                       random_modify_sigma=0.2))
        # a domain with no recorded statistics falls back to 'rand' inside mask_max_gate
        out["mmg/empty"] = pk(model.generate_mask("mask_max_gate", d=4 % spec.n_domain, init_active_percent=0.7,
                                                  random_modify_sigma=0.2))
        # pruning with the gate means of one fast-update step
        for i in range(6):
            cur = model.generate_mask("rand", init_active_percent=0.8)
            for l in range(1, spec.n_level):
                for t in range(n[l]):
                    col = cur[l][:, t].numpy().astype(np.float32)
                    g = rng.random(n[l - 1]).astype(np.float32) * col
                    model.tmp_tower_gate_values[l][t] = torch.from_numpy(g / max(g.sum(), 1e-6) * (g.sum() > 0))
            out[f"prune/{i}/in"] = pk(cur)
            res = model.prun_single_mask(0, cur, prun_ratio=[0.05, 0.3, 0.6][i % 3])
            out[f"prune/{i}/out"] = pk(res)
        # selection
        model.reset_for_mask_update()
        for d in range(spec.n_domain):
            for z in range(3):
                model.candidate_domain_mask[d].append(model.generate_mask("rand", init_active_percent=0.6))
                for _ in range(4):
                    model.add_eval_loss(float(rng.random()), d=d, mask_z=z)
        model.update_all_mask(regroup_times=1)
        for d in range(spec.n_domain):
            out[f"select/{d}"] = pk(model.domain_mask[d])
        out["active_ratio"] = np.array([model.count_current_active_ratio()])
    return out



# ---- harness run: driven identically on the reference (golden generation, CPU) and on aread_amd (GPU) -----------
def harness_config():
    import types
    c = types.SimpleNamespace()
    c.bs, c.lr, c.wd, c.update_lr = 64, 1e-3, 1e-8, 1e-2
    c.warm_up_interval, c.regroup_interval = 0.25, 0.5          # -> 4 warm-up steps, regroup every 8 steps
    c.regroup_update_step, c.regroup_eval_step = 2, 2
    c.candidate_mask_num, c.random_modify_sigma, c.init_active_percent = 2.5, 0.2, 0.7
    c.early_stop = 2
    return c


def harness_data(spec, seed=11):
    rng = np.random.default_rng(seed)
    def make(n):
        x = np.stack([rng.integers(0, d, n) for d in spec.field_dims]
                     + [rng.integers(0, spec.field_dims[0] + 1, n) for _ in range(spec.n_mh_slots)], axis=1).astype(np.int32)
        w = rng.standard_normal(spec.f_in)
        y = ((x * w).sum(1) % 7 < 3).astype(np.int16)             # a learnable-ish pattern
        return x, y
    return make(1000), make(300)


def run_harness(model, spec, device, seed=5):
    """One epoch of Trainer.main on the tiny dataset; returns the trace + final masks + validation result."""
    import contextlib, io
    import torch
    from aread_amd.harness import DomainStreams, Trainer
    (xt, yt), (xv, yv) = harness_data(spec)
    np.random.seed(seed)
    torch.manual_seed(seed)
    cfg = harness_config()
    tr = DomainStreams(torch.from_numpy(xt), torch.from_numpy(yt), spec.n_domain, spec.domain_idx, cfg.bs, device)
    va = DomainStreams(torch.from_numpy(xv), torch.from_numpy(yv), spec.n_domain, spec.domain_idx, cfg.bs, device)
    t = Trainer(model, cfg, tr, va, device=device, log=lambda *a: None)
    with contextlib.redirect_stdout(io.StringIO()):
        model.reset_for_mask_update()
        res = t.main(epochs=1)
    out = {"trace_tags": np.array([k for k, _ in t.trace]), "trace_vals": np.array([v for _, v in t.trace], dtype=np.float64)}
    out["masks"] = np.stack([O.pack_mask(spec, m) for m in model.domain_mask])
    out["valid"] = np.array([res[0]["total_auc"], res[0]["total_loss"], res[0]["mean_auc"], res[0]["mean_loss"]])
    return out
