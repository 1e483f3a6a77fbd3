"""CSV -> device tensors for the harness (tensorisation part): the host logic of the reference's
Run.get_data / read_split_data / save_tensor_from_data / convert2domain_data_loader (run.py:105-111, 112-236, 237-265,
310-353, 355-392) without wandb / keras / .pth caches.  The counterfactual augmenter (preprocess.py:368-474) is an
offline pandas job and stays out of scope; its output CSV is read here like the reference reads it.

Parity note: run.py cannot be imported in the build container (it imports wandb), so this module is pinned by
hand-computed cases in tests/test_data_cpu.py, not by vectors recorded from the reference ("parity unpinned")."""
import ast

import numpy as np
import torch

from .harness import DomainStreams

FEATURES = {
    # run.py:374-392
    "amazon": dict(features=["itemid", "weekday", "domain", "sales_chart", "sales_rank", "brand", "price",
                             "user_pos_6month_seq", "user_neg_6month_seq"], label="label", split="timestamp"),
    "aliccp": dict(features=["userid", "121", "122", "124", "125", "126", "127", "128", "129", "itemid", "domain", "207", "210",
                             "216", "508", "509", "702", "853", "109_14", "110_14", "127_14", "150_14", "301"],
                   label="click", split="train_tag"),
    "cloudtheme": dict(features=["userid", "itemid", "domain", "leaf_cate_id", "cate_level1_id"], label="click",
                       split="train_tag"),
}


def seq_extractor(seq, maxlen, padding_value):
    """run.py:105-111: the LAST maxlen ids of the history list, post-padded with the pad id (= itemid_all)."""
    seq = ast.literal_eval(seq) if isinstance(seq, str) else list(seq)
    if len(seq) >= maxlen:
        return np.array(seq[-maxlen:], dtype=np.int64)
    return np.array(list(seq) + [padding_value] * (maxlen - len(seq)), dtype=np.int64)


class Tensorised:
    """What Run.get_data leaves on `self` (run.py:146-158) plus the split tensors."""

    def __init__(self):
        self.x_cols = self.label = None
        self.itemid_idx = self.domain_idx = self.n_domain = None
        self.one_hot_feature_dims = self.multi_hot_flag = None
        self.multi_hot_dict = None
        self.splits = {}                      # 'train' | 'valid' | 'test' | 'aug' -> (X int32 [N, F_in], y int16 [N, 1])


def tensorise(frame, x_cols, label, seq_maxlen, itemid_all):
    """run.py:237-265: one-hot id columns first, then every history column as seq_maxlen slots; int32 / int16."""
    seq_cols = [c for c in x_cols if "seq" in c]
    id_cols = [c for c in x_cols if "seq" not in c]
    X = torch.tensor(frame[id_cols].values.astype(np.int64), dtype=torch.int)
    for c in seq_cols:
        seq = np.stack([seq_extractor(s, seq_maxlen, itemid_all) for s in frame[c]]) if len(frame) else \
            np.zeros((0, seq_maxlen), dtype=np.int64)
        X = torch.cat([X, torch.tensor(seq, dtype=torch.int)], dim=1)
    y = torch.tensor(frame[[label]].values.astype(np.int64), dtype=torch.short)
    return X, y


def read_split_data(path, dataset_name, aug_path=None, seq_maxlen=5, itemid_all=1368287, history=True, domain_filter=None):
    """run.py:112-236 + 237-265: read the prepared CSV, split it (Amazon: timestamp quantiles 0.9 / 0.95, run.py:142;
    AliCCP / cloudtheme: train_tag 0/1/2, run.py:144), derive the model geometry, tensorise every split."""
    import pandas as pd
    spec = FEATURES[dataset_name]
    x_cols = [f for f in spec["features"] if "seq" not in f]
    if history:
        x_cols += [f for f in spec["features"] if "seq" in f]
    label, split_col = spec["label"], spec["split"]
    cols = x_cols + [label, split_col]
    data = pd.read_csv(path, usecols=cols)
    if dataset_name == "amazon":
        train_valid, valid_test = data[split_col].quantile(0.9), data[split_col].quantile(0.95)
    else:
        train_valid, valid_test = 1, 2
    if domain_filter is not None:
        data = data.loc[data["domain"].isin(domain_filter)].copy()
    out = Tensorised()
    out.x_cols, out.label = x_cols, label
    out.itemid_idx, out.domain_idx = x_cols.index("itemid"), x_cols.index("domain")
    one_hot_cols = [c for c in x_cols if "seq" not in c]
    dims = np.max(data[one_hot_cols].values, axis=0).astype(np.int64) + 1
    if dataset_name == "amazon":
        dims[out.itemid_idx] = itemid_all                     # history ids may exceed the max of the itemid column
    out.one_hot_feature_dims = dims
    n_seq = len(x_cols) - len(one_hot_cols)
    out.multi_hot_flag = [False] * len(one_hot_cols) + [True] * n_seq * seq_maxlen
    out.n_domain = int(data["domain"].nunique())
    out.multi_hot_dict = {"multi_hot_flag": out.multi_hot_flag, "itemid_idx": out.itemid_idx, "seq_maxlen": seq_maxlen,
                          "method": "mean" if n_seq else None}                       # run.py:378-381
    parts = {"train": data[data[split_col] < train_valid],
             "valid": data[(data[split_col] >= train_valid) & (data[split_col] < valid_test)],
             "test": data[data[split_col] >= valid_test]}
    if aug_path is not None:
        aug = pd.read_csv(aug_path, usecols=cols)
        parts["aug"] = aug[aug[split_col] < train_valid]
    for name, frame in parts.items():
        out.splits[name] = tensorise(frame, x_cols, label, seq_maxlen, itemid_all)
    return out


def domain_streams(t, bs, device, shuffle_seq=True):
    """run.py:310-353: per-domain shuffled loaders + the shuffled list of domain ids that drives an epoch."""
    return {name: DomainStreams(X, y, t.n_domain, t.domain_idx, bs, device, shuffle_seq=shuffle_seq)
            for name, (X, y) in t.splits.items() if X.shape[0] > 0}
