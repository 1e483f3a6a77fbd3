"""Synthetic CTR batches shaped like the reference's datasets.  Bench/test infrastructure."""
import numpy as np

AMAZON_DOMAIN_SIZE = [69360, 282546, 776105, 3001846, 88496, 449031, 2859592, 1893, 1437340, 16454, 601698, 1802,
                      2416380, 197170, 202176, 6931, 317131, 132650, 602500, 585227, 845268, 1107407, 997451, 623565,
                      44843]                                     # config.py:60-61 (training-set domain sizes)
ALICCP_DOMAIN_SIZE = [2695782, 1433175, 925817, 584726, 461755, 358265, 166869, 113621, 78692, 65313, 54483, 45808,
                      40975, 37939, 34079, 31703, 29551, 27084, 25027, 23464, 21764, 19857, 18390, 16712, 15852, 14914,
                      13653, 12265, 11179, 9760]                 # config.py:62-64
# `spec` below is anything with field_dims / f_in / n_domain / itemid_idx / domain_idx / n_mh_fields / n_onehot / seq_maxlen:
# oracle.aread_oracle.Spec (tests) or aread_amd.presets.Workload (bench.py)
HIST_LEN_P = [0.62, 0.14, 0.07, 0.04, 0.03, 0.10]               # history length 0..5 (bundled sample histogram)


def _skewed(rng, n, card, top1):
    """ids with mass `top1` on id 0 and the rest uniform."""
    ids = rng.integers(0, card, n)
    return np.where(rng.random(n) < top1, 0, ids)


def amazon_batch(spec, rng, B, domain="proportional", items="zipf"):
    """int32 [B, 17] batch + float32 labels, Amazon-like: 7 one-hot columns then 2x5 history slots whose
    padding id (= itemid cardinality) aliases row 0 of the next field."""
    dims = spec.field_dims
    x = np.zeros((B, spec.f_in), dtype=np.int32)
    draw_item = (lambda n: np.floor(dims[0] * rng.random(n) ** 3).astype(np.int64)) if items == "zipf" \
        else (lambda n: rng.integers(0, dims[0], n))
    x[:, 0] = draw_item(B)
    x[:, 1] = rng.integers(0, dims[1], B)
    if domain == "proportional":
        p = np.asarray(AMAZON_DOMAIN_SIZE[:spec.n_domain], dtype=np.float64)
        x[:, 2] = rng.choice(spec.n_domain, size=B, p=p / p.sum())
    elif domain == "uniform":
        x[:, 2] = rng.integers(0, spec.n_domain, B)
    else:
        x[:, 2] = int(domain)
    x[:, 3] = _skewed(rng, B, dims[3], 0.50)
    x[:, 4] = _skewed(rng, B, dims[4], 0.59)
    x[:, 5] = _skewed(rng, B, dims[5], 0.29)
    x[:, 6] = _skewed(rng, B, dims[6], 0.51)
    pad = dims[spec.itemid_idx]
    for f in range(spec.n_mh_fields):
        Lh = rng.choice(6, size=B, p=HIST_LEN_P)
        for s in range(spec.seq_maxlen):
            x[:, spec.n_onehot + f * spec.seq_maxlen + s] = np.where(s < Lh, draw_item(B), pad)
    y = (rng.random(B) < 0.5).astype(np.float32)
    return x, y


def generic_batch(spec, rng, B, domain_p=None, pos_rate=0.5):
    """Any spec (e.g. AliCCP-like, no multi-hot): zipf-ish ids per field, domain ~ domain_p."""
    x = np.zeros((B, spec.f_in), dtype=np.int32)
    for j, dim in enumerate(spec.field_dims):
        x[:, j] = np.floor(dim * rng.random(B) ** 3).astype(np.int64)
    if domain_p is not None:
        p = np.asarray(domain_p, dtype=np.float64)
        x[:, spec.domain_idx] = rng.choice(spec.n_domain, size=B, p=p / p.sum())
    else:
        x[:, spec.domain_idx] = rng.integers(0, spec.n_domain, B)
    pad = spec.field_dims[spec.itemid_idx]
    for f in range(spec.n_mh_fields):
        Lh = rng.choice(6, size=B, p=HIST_LEN_P)
        for s in range(spec.seq_maxlen):
            x[:, spec.n_onehot + f * spec.seq_maxlen + s] = np.where(s < Lh, np.floor(pad * rng.random(B) ** 3), pad)
    y = (rng.random(B) < pos_rate).astype(np.float32)
    return x, y
