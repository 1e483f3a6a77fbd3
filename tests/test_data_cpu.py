"""CPU: CSV tensorisation (aread_amd/data.py) on small synthetic CSVs with hand-computed expectations
(run.py itself cannot be imported here -- wandb -- so this host logic is "parity unpinned", see data.py)."""
import numpy as np
import pandas as pd
import torch

from aread_amd import data as D


def test_seq_extractor_last_ids_post_padded():
    assert D.seq_extractor("[3, 5]", 5, 99).tolist() == [3, 5, 99, 99, 99]
    assert D.seq_extractor("[]", 5, 99).tolist() == [99] * 5
    assert D.seq_extractor("[1, 2, 3, 4, 5, 6, 7]", 5, 99).tolist() == [3, 4, 5, 6, 7]       # the LAST five
    assert D.seq_extractor([8, 9], 3, 0).tolist() == [8, 9, 0]


def _amazon_csv(path, n=40):
    rng = np.random.default_rng(0)
    rows = []
    for i in range(n):
        pos = rng.integers(0, 50, rng.integers(0, 8)).tolist()
        neg = rng.integers(0, 50, rng.integers(0, 3)).tolist()
        rows.append(dict(userid=i, itemid=int(rng.integers(0, 30)), weekday=i % 7, domain=i % 4, sales_chart=i % 5,
                         sales_rank=i % 3, brand=i % 11, price=i % 6, user_pos_1month_seq="[]", user_neg_1month_seq="[]",
                         user_pos_6month_seq=str(pos), user_neg_6month_seq=str(neg), label=i % 2, timestamp=1000 + i))
    pd.DataFrame(rows).to_csv(path, index=False)
    return rows


def test_amazon_layout_and_split(tmp_path):
    p = tmp_path / "amazon.csv"
    rows = _amazon_csv(p)
    t = D.read_split_data(str(p), "amazon", seq_maxlen=5, itemid_all=60)
    assert t.x_cols[:7] == ["itemid", "weekday", "domain", "sales_chart", "sales_rank", "brand", "price"]
    assert (t.itemid_idx, t.domain_idx, t.n_domain) == (0, 2, 4)
    assert t.multi_hot_flag == [False] * 7 + [True] * 10
    assert t.one_hot_feature_dims.tolist() == [60, 7, 4, 5, 3, 11, 6]                 # itemid dim forced to itemid_all
    assert t.multi_hot_dict == {"multi_hot_flag": t.multi_hot_flag, "itemid_idx": 0, "seq_maxlen": 5, "method": "mean"}
    Xtr, ytr = t.splits["train"]
    Xva, _ = t.splits["valid"]
    Xte, _ = t.splits["test"]
    assert Xtr.dtype == torch.int32 and ytr.dtype == torch.int16 and ytr.shape == (Xtr.shape[0], 1)
    assert Xtr.shape[1] == 17
    ts = np.array([r["timestamp"] for r in rows])
    q90, q95 = np.quantile(ts, 0.9), np.quantile(ts, 0.95)
    assert Xtr.shape[0] == (ts < q90).sum() and Xva.shape[0] == ((ts >= q90) & (ts < q95)).sum()
    assert Xte.shape[0] == (ts >= q95).sum()
    r = rows[3]
    want = [r["itemid"], r["weekday"], r["domain"], r["sales_chart"], r["sales_rank"], r["brand"], r["price"]]
    want += D.seq_extractor(r["user_pos_6month_seq"], 5, 60).tolist() + D.seq_extractor(r["user_neg_6month_seq"], 5, 60).tolist()
    assert Xtr[3].tolist() == want and int(ytr[3]) == r["label"]


def test_aliccp_layout_and_streams(tmp_path):
    rng = np.random.default_rng(1)
    cols = D.FEATURES["aliccp"]["features"]
    n = 90
    frame = {c: rng.integers(0, 7, n) for c in cols}
    frame["domain"] = np.arange(n) % 3
    frame["click"] = rng.integers(0, 2, n)
    frame["purchase"] = np.zeros(n, dtype=np.int64)
    frame["train_tag"] = np.array([0] * 60 + [1] * 15 + [2] * 15)
    p = tmp_path / "aliccp.csv"
    pd.DataFrame(frame).to_csv(p, index=False)
    aug = tmp_path / "aliccp_aug.csv"
    pd.DataFrame({k: v[:30] for k, v in frame.items()}).to_csv(aug, index=False)
    t = D.read_split_data(str(p), "aliccp", aug_path=str(aug))
    assert (t.itemid_idx, t.domain_idx, t.n_domain) == (9, 10, 3)
    assert t.multi_hot_flag == [False] * 23 and t.multi_hot_dict["method"] is None
    assert [t.splits[k][0].shape[0] for k in ("train", "valid", "test", "aug")] == [60, 15, 15, 30]
    assert t.splits["train"][0][:, 10].tolist() == (np.arange(60) % 3).tolist()
    np.random.seed(0); torch.manual_seed(0)
    s = D.domain_streams(t, bs=8, device="cpu")
    tr = s["train"]
    assert sorted(tr.batch_seq) == [0] * 3 + [1] * 3 + [2] * 3                       # ceil(20 / 8) batches per domain
    assert abs(tr.domain_cnt_weight.sum() - 1.0) < 1e-12
    X, y = tr.next(1)
    assert X.shape == (8, 23) and (X[:, 10] == 1).all() and y.shape == (8, 1)
    for _ in range(5):                                                               # restart on exhaustion
        X, y = tr.next(1)
    assert (X[:, 10] == 1).all()
