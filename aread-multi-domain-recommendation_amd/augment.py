"""Popularity-based counterfactual augmenter (second half): the host logic of the reference's
DataPreprocessing.make_augmentation (preprocess.py:368-474), whose output CSV (`*_aug{ratio}.csv`) is the `aug` stream
of the HEMP fast-update steps (run.py:634-648).  Offline pandas work in the reference and here: no kernel.

What it does (preprocess.py line numbers):
  * item popularity = (positives + 1) / (exposures + 2) per itemid                                   (:389-395)
  * cold items: exposures <= 4 (amazon) / popularity < 0.05 (aliccp) / popularity < 0.2 (cloudtheme)  (:402-437)
  * minority domains: rows <= int(n * 0.02) (amazon) resp. int(n * 0.015); majority domains: more than 1.5 x that
    (amazon, cloudtheme) resp. more than that (aliccp)
  * source pool = positive rows of cold items in majority domains                                     (:441-442)
  * int(n * aug_ratio) rows drawn from the pool with replacement, weights proportional to 1 / popularity   (:447-450)
  * every drawn row is re-assigned to a minority domain drawn with weights exp(w / quantile_0.3(w)),
    w = max(100, target - size), target = (sum of minority sizes + n_aug) / #minority domains           (:452-458)
  * output = original rows (is_augmented False) followed by the drawn rows (is_augmented True)         (:461,469)

The random draws consume numpy's GLOBAL stream exactly like the reference (DataFrame.sample(random_state=None) and
np.random.choice), so under the same np.random.seed the output rows are identical: tests/golden/augment_*.npz were
recorded from the reference itself (tests/golden/make_golden_aug.py)."""
import os

import numpy as np
import pandas as pd

# (cold-item rule, minority fraction, majority factor) per dataset -- preprocess.py:402-437
_RULES = {
    "amazon": dict(cold=("exposure", 4), small_frac=0.02, large_factor=1.5, label="label", count_col=None),
    "aliccp": dict(cold=("popularity", 0.05), small_frac=0.015, large_factor=1.0, label="click", count_col=None),
    "cloudtheme": dict(cold=("popularity", 0.2), small_frac=0.015, large_factor=1.5, label="click", count_col="clk_cnt"),
}
POSITIVE_SMOOTHING, TOTAL_SMOOTHING = 1, 2          # preprocess.py:381-386 (the same for every dataset)
MIN_DOMAIN_WEIGHT = 100                             # preprocess.py:455


def item_popularity(data, count_col):
    """per itemid: total_count, positive_count, popularity = (pos + 1) / (total + 2)   (preprocess.py:389-395)"""
    grp = data.groupby("itemid")[count_col]
    pop = pd.DataFrame({"total_count": grp.count(), "positive_count": grp.sum()})
    pop["popularity"] = (pop["positive_count"] + POSITIVE_SMOOTHING) / (pop["total_count"] + TOTAL_SMOOTHING)
    return pop


def domain_weights(domain_counts, small_domains, aug_len):
    """minority domains further below the common target get exponentially more of the augmented rows (:452-457)"""
    small = domain_counts.loc[small_domains]
    target = (small.sum() + aug_len) / len(small_domains)
    w = target - small
    w.loc[w < MIN_DOMAIN_WEIGHT] = MIN_DOMAIN_WEIGHT
    w = np.exp(w / w.quantile(0.3))
    return w / w.sum()


def make_augmentation(data, dataset_name, aug_ratio, rng=None):
    """data: the prepared frame (preprocess_path CSV).  Returns the augmented frame.  rng: anything with numpy's
    RandomState.choice signature; default = numpy's global stream, like the reference."""
    if dataset_name not in _RULES:
        raise ValueError(f"unknown dataset {dataset_name!r}")
    if not aug_ratio or aug_ratio <= 0:
        raise ValueError("aug_ratio must be greater than 0")                 # preprocess.py:84-87
    rule = _RULES[dataset_name]
    rng = np.random if rng is None else rng
    label = rule["label"]
    data = data.copy()
    n = data.shape[0]
    aug_len = int(n * aug_ratio)
    pop = item_popularity(data, rule["count_col"] or label)
    domain_counts = data["domain"].value_counts()
    data["is_augmented"] = False
    kind, thr = rule["cold"]
    cold_items = (pop.index[pop["total_count"] <= thr] if kind == "exposure" else pop.index[pop["popularity"] < thr]).to_numpy()
    small_thr = int(n * rule["small_frac"])
    large_domains = domain_counts.index[domain_counts > rule["large_factor"] * small_thr]
    small_domains = domain_counts.index[domain_counts <= small_thr]
    # the reference writes `a & b & data[label] == 1`, i.e. ((a & b & label) == 1): identical to "all three" for 0/1 labels
    pool = data[(data["itemid"].isin(cold_items) & data["domain"].isin(large_domains) & data[label]) == 1]
    if len(pool) == 0 or len(small_domains) == 0:
        raise ValueError("nothing to augment from: no positive cold-item row in a majority domain, or no minority domain")
    inv = 1 / pop.loc[pool["itemid"], "popularity"]
    w_item = np.asarray((inv / inv.sum()).tolist(), dtype=np.float64)
    w_item = w_item / w_item.sum()                                          # DataFrame.sample renormalises once more
    picked = rng.choice(len(pool), size=aug_len, replace=True, p=w_item)
    aug = pool.iloc[picked].copy()
    aug["domain"] = rng.choice(small_domains, size=aug_len, p=domain_weights(domain_counts, small_domains, aug_len))
    aug["is_augmented"] = True
    return pd.concat([data, aug])


def write_augmentation(preprocess_path, aug_path, dataset_name, aug_ratio, rng=None):
    """File-level behaviour of preprocess.py:373-374,468-469: nothing happens when aug_path exists."""
    if os.path.exists(aug_path):
        return False
    make_augmentation(pd.read_csv(preprocess_path), dataset_name, aug_ratio, rng).to_csv(aug_path, index=False)
    return True
