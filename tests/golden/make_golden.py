#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/make_golden.py

Every fixture stores inputs + the reference's outputs.  Parameters are not stored: they are
re-created from oracle.aread_oracle.init_params(spec, seed) (a numpy stream keyed by tensor name), and this
script loads exactly those tensors into the reference model with load_state_dict(strict=True).
Large gradient tensors are stored as (strided sample, sum, abs-sum) to keep fixtures small.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import aread_oracle as O  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from model.aread import AREAD  # noqa: E402  (reference)
    from model.layer import FeaturesEmbedding  # noqa: E402  (reference)

GRAD_SAMPLE_STRIDE = 37
FULL_LIMIT = 8192


def ref_config(spec: O.Spec):
    cfg = types.SimpleNamespace()
    cfg.dataset_name = "synthetic"
    cfg.domain_size = {"synthetic": [100 + d for d in range(spec.n_domain)]}
    cfg.use_dcn = True
    cfg.use_atten = True
    cfg.n_cross_layers = spec.n_cross
    cfg.mmoe_n_expert = spec.n_expert
    cfg.atten_embed_dim = spec.atten_embed_dim
    cfg.att_layer_num = spec.att_layer_num
    cfg.att_head_num = 2
    cfg.att_res = True
    return cfg


def build_reference(spec: O.Spec, seed: int):
    mh = {"multi_hot_flag": list(spec.multi_hot_flag), "itemid_idx": spec.itemid_idx,
          "seq_maxlen": spec.seq_maxlen, "method": spec.method}
    with contextlib.redirect_stdout(io.StringIO()):
        model = AREAD(np.array(spec.field_dims), spec.embed_dim, mh, tuple(spec.n_tower), spec.n_domain, "mmoe",
                      tuple(spec.expert_dims), tuple(tuple(t) for t in spec.tower_dims), spec.domain_idx,
                      n_cross_layers=spec.n_cross, dropout=0.0, device=torch.device("cpu"),
                      config=ref_config(spec))
    sd = model.state_dict()
    P = O.init_params(spec, seed)
    assert set(sd.keys()) == set(P.keys()), (set(sd.keys()) ^ set(P.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    model.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    model.reset_for_mask_update()
    logits = {}
    for i, layer in enumerate(model.output_layers):
        layer.register_forward_hook(lambda m, inp, out, i=i: logits.__setitem__(i, inp[0].detach().clone()))
    return model, logits


def tmask(mask):
    return [torch.tensor(np.asarray(m), dtype=torch.bool) for m in mask]


def synth_x(spec: O.Spec, rng, B, domains=None):
    x = np.zeros((B, spec.f_in), dtype=np.int32)
    for j, dim in enumerate(spec.field_dims):
        x[:, j] = rng.integers(0, dim, B)
    if domains is not None:
        x[:, spec.domain_idx] = domains
    pad = spec.field_dims[spec.itemid_idx]
    for f in range(spec.n_mh_fields):
        L = rng.choice([0, 1, 2, 3, 5], size=B, p=[0.5, 0.2, 0.1, 0.1, 0.1])
        for s in range(spec.seq_maxlen):
            col = spec.n_onehot + f * spec.seq_maxlen + s
            ids = rng.integers(0, pad, B)
            x[:, col] = np.where(s < L, ids, pad)       # pad id aliases the next field's row 0
    return x


def pack_grads(out, prefix, grads):
    for name, g in grads.items():
        a = g.detach().numpy().astype(np.float32).reshape(-1)
        if a.size <= FULL_LIMIT:
            out[f"{prefix}/full/{name}"] = a
        else:
            out[f"{prefix}/samp/{name}"] = a[::GRAD_SAMPLE_STRIDE].copy()
            out[f"{prefix}/sum/{name}"] = np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum()])


def buffers_of(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}


def pack_buffers(out, prefix, bufs):
    for k, v in bufs.items():
        out[f"{prefix}/{k}"] = v.numpy()


def ref_grads(model):
    return {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
            for n, p in model.named_parameters()}


def ref_step(model, logits_hook, spec, x, y, masks, want_gate=True):
    """The reference as its own training loop drives it (run.py:668-682), one call per domain."""
    model.train()
    model.zero_grad()
    crit = torch.nn.BCELoss()
    total = torch.zeros(1)
    n_heads = spec.n_tower[-1]
    probs = np.full((n_heads, x.shape[0]), np.nan, np.float32)
    logit = np.full((n_heads, x.shape[0]), np.nan, np.float32)
    gates = {}
    dom = x[:, spec.domain_idx]
    for d in range(spec.n_domain):
        idx = np.nonzero(dom == d)[0]
        if idx.size == 0:
            continue
        m = tmask(masks[d])
        preds = model(torch.from_numpy(x[idx]), mode="domain_mask_bagging", domain_i=d, current_mask=m,
                      tmp_memory_gate_value=want_gate)
        yt = torch.from_numpy(y[idx].astype(np.float32))
        losses = [crit(p, yt) for p in preds.unbind(dim=0)]
        total = total + sum(losses) / preds.shape[0]
        act = np.nonzero(np.asarray(masks[d][spec.n_level - 1]).any(axis=0))[0]
        probs[np.ix_(act, idx)] = preds.detach().numpy()
        for k, i in enumerate(act):
            logit[i, idx] = logits_hook[i].reshape(-1).numpy()
        if want_gate:
            gs = []
            for l in range(1, spec.n_level):
                gs.append(torch.stack([g for g in model.tmp_tower_gate_values[l]], dim=1).numpy())
            gates[d] = gs
    reg = model.get_regularization_loss(device=torch.device("cpu"))
    loss = total + reg
    loss.backward()
    return dict(loss=float(loss), bag=float(total), reg=float(reg), probs=probs, logits=logit,
                grads=ref_grads(model), buffers=buffers_of(model), gates=gates)


def gen_embedding(path):
    spec = O.Spec(field_dims=[40, 7, 5, 9, 11, 30, 10], embed_dim=32, multi_hot_flag=[False] * 7 + [True] * 10,
                  itemid_idx=0, seq_maxlen=5, method="mean", n_domain=5, domain_idx=2)
    rng = np.random.default_rng(7)
    out = {}
    for method in ("mean", "sum"):
        mh = {"multi_hot_flag": list(spec.multi_hot_flag), "itemid_idx": 0, "seq_maxlen": 5, "method": method}
        with contextlib.redirect_stdout(io.StringIO()):
            emb = FeaturesEmbedding(list(spec.field_dims), 32, mh)
        W = O.init_tensor("embedding.embedding_dict.weight", (spec.rows, 32), "emb", 123)
        emb.embedding_dict.weight.data.copy_(W)
        x = synth_x(spec, rng, 67)
        xt = torch.from_numpy(x)
        bag = (xt + xt.new_tensor(emb.offsets).unsqueeze(0))
        o = emb(xt)
        dout = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
        o.backward(dout)
        out[f"{method}/x"] = x
        out[f"{method}/bag"] = bag.numpy()
        out[f"{method}/out"] = o.detach().numpy()
        out[f"{method}/dout"] = dout.numpy()
        out[f"{method}/dtable"] = emb.embedding_dict.weight.grad.numpy()
        out[f"{method}/offsets"] = np.asarray(emb.offsets)
    # no multi-hot (AliCCP-like) layout
    dims = [13, 4, 6, 3, 17]
    mh = {"multi_hot_flag": [False] * 5, "itemid_idx": 0, "seq_maxlen": 5, "method": None}
    with contextlib.redirect_stdout(io.StringIO()):
        emb = FeaturesEmbedding(dims, 32, mh)
    W = O.init_tensor("embedding.embedding_dict.weight", (sum(dims), 32), "emb", 123)
    emb.embedding_dict.weight.data.copy_(W)
    x = np.stack([rng.integers(0, d, 33) for d in dims], axis=1).astype(np.int32)
    o = emb(torch.from_numpy(x), squeeze_dim=True)
    out["flat/x"], out["flat/out"] = x, o.detach().numpy()
    np.savez_compressed(path, **out)


def gen_model(path, spec: O.Spec, seed, tag):
    rng = np.random.default_rng(1000 + seed)
    out = {}
    nd = spec.n_domain
    masks_sets = {
        "ones": [O.full_mask(spec) for _ in range(nd)],
        "rand": [O.random_valid_mask(spec, rng, 0.7) for _ in range(nd)],
        "sparse": [O.random_valid_mask(spec, rng, 0.35) for _ in range(nd)],
    }
    for mname, masks in masks_sets.items():
        out[f"masks/{mname}"] = np.stack([O.pack_mask(spec, m) for m in masks])

    # ---- case 1: single-domain bagging train step, three masks -------------------------------------
    for mname, masks in masks_sets.items():
        model, hook = build_reference(spec, seed)
        d = 3 % nd
        x = synth_x(spec, rng, 96, domains=d)
        y = (rng.random(96) < 0.5).astype(np.int16)
        r = ref_step(model, hook, spec, x, y, masks)
        p = f"single_{mname}"
        out[f"{p}/x"], out[f"{p}/y"], out[f"{p}/domain"] = x, y, np.array(d)
        out[f"{p}/probs"], out[f"{p}/logits"] = r["probs"], r["logits"]
        out[f"{p}/loss"] = np.array([r["loss"], r["bag"], r["reg"]])
        for l, g in enumerate(r["gates"][d]):
            out[f"{p}/gate{l + 1}"] = g
        pack_grads(out, f"{p}/grad", r["grads"])
        pack_buffers(out, f"{p}/buf", r["buffers"])

    # ---- case 2: multi-domain step (ragged: one domain empty, one with a single row) ---------------
    model, hook = build_reference(spec, seed)
    B = 230
    sizes = rng.multinomial(B - 1, np.ones(nd - 2) / (nd - 2))
    doms = np.concatenate([np.full(s, d) for d, s in enumerate(sizes)] + [np.full(1, nd - 2)])  # nd-1 is empty
    rng.shuffle(doms)
    x = synth_x(spec, rng, B, domains=doms)
    y = (rng.random(B) < 0.5).astype(np.int16)
    r = ref_step(model, hook, spec, x, y, masks_sets["rand"])
    p = "multi_rand"
    out[f"{p}/x"], out[f"{p}/y"] = x, y
    out[f"{p}/probs"], out[f"{p}/logits"] = r["probs"], r["logits"]
    out[f"{p}/loss"] = np.array([r["loss"], r["bag"], r["reg"]])
    for d, gs in r["gates"].items():
        for l, g in enumerate(gs):
            out[f"{p}/gate{l + 1}/d{d}"] = g
    pack_grads(out, f"{p}/grad", r["grads"])
    pack_buffers(out, f"{p}/buf", r["buffers"])

    # ---- case 3: eval-mode domain_with_mask (running stats) ----------------------------------------
    model, hook = build_reference(spec, seed)
    model.eval()
    d = 1
    x = synth_x(spec, rng, 64, domains=d)
    with torch.no_grad():
        yv = model(torch.from_numpy(x), mode="domain_with_mask", domain_i=d,
                   current_mask=tmask(masks_sets["rand"][d]))
    out["eval_with_mask/x"], out["eval_with_mask/domain"], out["eval_with_mask/y"] = x, np.array(d), yv.numpy()

    # ---- case 4: wo_mask warm-up step with gate recording -------------------------------------------
    model, hook = build_reference(spec, seed)
    model.train(); model.zero_grad()
    d = 2
    x = synth_x(spec, rng, 80, domains=d)
    y = (rng.random(80) < 0.5).astype(np.int16)
    pred = model(torch.from_numpy(x), mode="wo_mask", domain_i=d, memory_gate_value=True)
    loss = torch.nn.BCELoss()(pred.squeeze(), torch.from_numpy(y.astype(np.float32)))
    reg = model.get_regularization_loss(device=torch.device("cpu"))
    (loss + reg).backward()
    p = "wo_mask"
    out[f"{p}/x"], out[f"{p}/y"], out[f"{p}/domain"] = x, y, np.array(d)
    out[f"{p}/pred"] = pred.detach().numpy()
    out[f"{p}/loss"] = np.array([float(loss + reg), float(loss), float(reg)])
    for l in range(1, spec.n_level):
        out[f"{p}/gate{l}"] = torch.stack([model.domain_tower_gate_values[d][l][t][0]
                                          for t in range(spec.n_tower[l])], dim=1).numpy()
    pack_grads(out, f"{p}/grad", ref_grads(model))
    pack_buffers(out, f"{p}/buf", buffers_of(model))

    # ---- case 5: one-row call (BatchNorm skipped, layer.py:226) --------------------------------------
    model, hook = build_reference(spec, seed)
    d = 0
    x = synth_x(spec, rng, 1, domains=d)
    y = np.array([1], dtype=np.int16)
    r = ref_step(model, hook, spec, x, y, masks_sets["rand"], want_gate=False)
    p = "one_row"
    out[f"{p}/x"], out[f"{p}/y"] = x, y
    out[f"{p}/probs"], out[f"{p}/loss"] = r["probs"], np.array([r["loss"], r["bag"], r["reg"]])
    pack_grads(out, f"{p}/grad", r["grads"])

    out["meta/seed"] = np.array(seed)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays  [{tag}]")


def spec_full():
    """Real layer widths of the path (config.py:22,39,57), small tables."""
    return O.Spec(field_dims=[40, 7, 5, 9, 11, 30, 10], embed_dim=32, multi_hot_flag=[False] * 7 + [True] * 10,
                  itemid_idx=0, seq_maxlen=5, method="mean", n_tower=(3, 6, 12), n_domain=5, domain_idx=2)


def spec_tiny():
    """Odd widths to exercise tails / generality; no multi-hot; different tower counts."""
    return O.Spec(field_dims=[23, 4, 6, 3, 17], embed_dim=16, multi_hot_flag=[False] * 5, itemid_idx=0,
                  method=None, n_tower=(2, 3, 5), n_domain=4, domain_idx=1, n_expert=3,
                  expert_dims=(40, 24, 12), tower_dims=((12, 8), (8, 8), (8, 4)), n_cross=2,
                  atten_embed_dim=64)


def gen_hemp(path):
    """Mask sequences of the reference's HEMP host logic under fixed seeds (tests/util.py::hemp_sequence)."""
    from tests.util import hemp_sequence
    spec = spec_full()
    model, _ = build_reference(spec, 123)
    out = hemp_sequence(model, spec)
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays")


def gen_harness(path):
    """One epoch of the training harness (aread_amd/harness.py) driving the REFERENCE model on the CPU."""
    from tests.util import run_harness
    spec = spec_full()
    model, _ = build_reference(spec, 123)
    out = run_harness(model, spec, torch.device("cpu"))
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out['trace_vals'])} trace points, valid auc/logloss/mean_auc/mean_loss = {out['valid']}")


if __name__ == "__main__":
    torch.manual_seed(2000)
    np.random.seed(2000)
    torch.set_num_threads(4)
    gen_embedding(os.path.join(HERE, "embedding.npz"))
    gen_model(os.path.join(HERE, "aread_full.npz"), spec_full(), 123, "full widths")
    gen_model(os.path.join(HERE, "aread_tiny.npz"), spec_tiny(), 321, "tiny/odd widths")
    gen_hemp(os.path.join(HERE, "hemp.npz"))
    gen_harness(os.path.join(HERE, "harness.npz"))
