"""CPU: the counterfactual augmenter (aread_amd/augment.py) against outputs recorded from the reference's own
DataPreprocessing.make_augmentation (preprocess.py:368-474) on its bundled sample CSVs (tests/golden/make_golden_aug.py)."""
import os

import numpy as np
import pandas as pd
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(name):
    z = np.load(os.path.join(GOLD, f"augment_{name}.npz"))
    cols = [str(c) for c in z["cols"]]
    return z, cols, pd.DataFrame(z["base"], columns=cols)


@pytest.mark.parametrize("name", ["amazon", "aliccp"])
def test_augmenter_matches_reference_rows(name):
    from aread_amd.augment import make_augmentation
    z, cols, base = _case(name)
    np.random.seed(int(z["seed"]))
    out = make_augmentation(base, name, float(z["ratio"]))
    n = int(z["n_base"])
    assert len(out) == int(z["n_total"]) and len(out) - n == int(n * float(z["ratio"]))
    assert not out.iloc[:n]["is_augmented"].any() and out.iloc[n:]["is_augmented"].all()
    np.testing.assert_array_equal(out.iloc[:n][cols].to_numpy(dtype=np.int64), z["base"])        # originals untouched, in order
    np.testing.assert_array_equal(out.iloc[n:][cols].to_numpy(dtype=np.int64), z["rows"])        # bit-exact draws


def test_augmenter_rules_on_a_hand_case():
    """popularity (pos+1)/(cnt+2), cold = exposures <= 4, minority = rows <= int(n*0.02), sources are POSITIVE rows of cold
    items in majority domains, every augmented row lands in a minority domain."""
    from aread_amd.augment import item_popularity, make_augmentation
    rng = np.random.RandomState(0)
    n = 1000
    dom = np.where(np.arange(n) < 985, 0, np.where(np.arange(n) < 993, 1, 2))          # sizes 985 / 8 / 7: threshold int(20)
    item = np.where(np.arange(n) < 900, np.arange(n) % 3, 100 + np.arange(n))           # items 0..2 are hot, the rest seen once
    lab = (np.arange(n) % 2).astype(np.int64)
    df = pd.DataFrame({"itemid": item, "domain": dom, "label": lab})
    pop = item_popularity(df, "label")
    assert pop.loc[0, "total_count"] == 300 and abs(pop.loc[0, "popularity"] - (pop.loc[0, "positive_count"] + 1) / 302) < 1e-15
    out = make_augmentation(df, "amazon", 0.05, rng=rng)
    aug = out.iloc[n:]
    assert len(aug) == 50 and set(aug["domain"]) <= {1, 2}
    assert (aug["label"] == 1).all() and (aug["itemid"] >= 100).all()                    # positive, cold
    src = df[(df["itemid"] >= 100) & (df["domain"] == 0) & (df["label"] == 1)]
    assert set(aug["itemid"]) <= set(src["itemid"])                                      # drawn from the majority domain only
    with pytest.raises(ValueError):
        make_augmentation(df, "amazon", 0.0)
    with pytest.raises(ValueError):
        make_augmentation(df, "movielens", 0.1)


def test_write_augmentation_skips_existing(tmp_path):
    from aread_amd.augment import write_augmentation
    z, cols, base = _case("aliccp")
    src, dst = tmp_path / "base.csv", tmp_path / "base_aug0.1.csv"
    base.to_csv(src, index=False)
    np.random.seed(int(z["seed"]))
    assert write_augmentation(str(src), str(dst), "aliccp", 0.1) is True
    got = pd.read_csv(dst)
    np.testing.assert_array_equal(got.iloc[int(z["n_base"]):][cols].to_numpy(dtype=np.int64), z["rows"])
    assert write_augmentation(str(src), str(dst), "aliccp", 0.1) is False                 # preprocess.py:373-374
