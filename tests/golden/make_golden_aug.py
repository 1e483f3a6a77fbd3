#!/usr/bin/env python
"""Record tests/golden/augment_{amazon,aliccp}.npz by running the REAL reference augmenter
(DataPreprocessing.make_augmentation, /root/reference/preprocess.py:368-474) on the reference's bundled sample CSVs under
a fixed numpy seed.  Build container only; the reference writes its output to a scratch directory (never into
/root/reference).  Stored: the seed, the id columns of the base CSV (the input) and the same columns of every augmented row (the output)."""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)

CASES = {
    "amazon": dict(data_path=os.path.join(REF, "dataset", "amazon"), kw=dict(prepare2train_month=12),
                   ids=["userid", "itemid", "domain", "label", "timestamp"]),
    "aliccp": dict(data_path=os.path.join(REF, "dataset", "aliccp"), kw=dict(),
                   ids=["userid", "itemid", "domain", "click", "207"]),
}
SEED, RATIO = 2000, 0.1


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        from preprocess import DataPreprocessing                       # reference
    for name, case in CASES.items():
        tmp = tempfile.mkdtemp(prefix="aread_aug_")
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                dp = DataPreprocessing(case["data_path"], name, None, is_aug=True, aug_ratio=RATIO, **case["kw"])
            base = dp.preprocess_path
            assert os.path.exists(base), base
            dp.preprocess_aug_path = os.path.join(tmp, "aug.csv")      # keep the reference tree untouched
            np.random.seed(SEED)
            dp.make_augmentation()
            sys.stdout = sys.__stdout__
            out = pd.read_csv(dp.preprocess_aug_path)
            n = pd.read_csv(base).shape[0]
            aug = out.iloc[n:]
            assert bool(aug["is_augmented"].all()) and not bool(out.iloc[:n]["is_augmented"].any())
            np.savez_compressed(os.path.join(HERE, f"augment_{name}.npz"), seed=SEED, ratio=RATIO, n_base=n, n_total=len(out),
                                base_name=os.path.basename(base), cols=np.array(case["ids"]),
                                base=out.iloc[:n][case["ids"]].to_numpy(dtype=np.int64),      # the input (id columns only)
                                rows=aug[case["ids"]].to_numpy(dtype=np.int64))                # the reference's output
            print(f"{name}: {n} base rows, {len(aug)} augmented rows, domains {sorted(aug['domain'].unique().tolist())[:8]} ...")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
