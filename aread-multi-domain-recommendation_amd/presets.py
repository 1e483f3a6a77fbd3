"""Workload presets of the reference's two datasets (shapes only: config.py:7-8,22,39,57-65 and run.py:57-59,157,377-457 of
the reference) and a model factory for them: what main.py / run.py derive from the prepared CSVs, as constants, so that a
model of the BASELINE shapes can be built without any data file (bench.py, smoke tests, multi-GPU rehearsals)."""
import types
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import torch

AMAZON_DOMAIN_SIZE = [69360, 282546, 776105, 3001846, 88496, 449031, 2859592, 1893, 1437340, 16454, 601698, 1802,
                      2416380, 197170, 202176, 6931, 317131, 132650, 602500, 585227, 845268, 1107407, 997451, 623565,
                      44843]                                     # config.py:60-61 (training-set domain sizes)
ALICCP_DOMAIN_SIZE = [2695782, 1433175, 925817, 584726, 461755, 358265, 166869, 113621, 78692, 65313, 54483, 45808,
                      40975, 37939, 34079, 31703, 29551, 27084, 25027, 23464, 21764, 19857, 18390, 16712, 15852, 14914,
                      13653, 12265, 11179, 9760]                 # config.py:62-64


@dataclass
class Workload:
    name: str
    field_dims: Sequence[int]
    multi_hot_flag: Sequence[bool]
    itemid_idx: int
    domain_idx: int
    n_domain: int
    domain_size: Sequence[int]
    embed_dim: int = 32
    seq_maxlen: int = 5
    method: Optional[str] = "mean"
    n_tower: Sequence[int] = (3, 6, 12)           # run.py:438-440: 3 * 2^l
    n_expert: int = 4                             # config.py:39
    expert_dims: Sequence[int] = (256, 128, 64)   # config.py:22
    tower_dims: Sequence[Sequence[int]] = ((64, 32), (32, 16), (16, 8))   # config.py:57
    n_cross: int = 3
    dropout: float = 0.2
    l2: float = 1e-5
    pos_rate: float = 0.5
    extra: dict = field(default_factory=dict)

    @property
    def n_onehot(self): return len(self.field_dims)
    @property
    def n_mh_slots(self): return int(sum(self.multi_hot_flag))
    @property
    def n_mh_fields(self): return self.n_mh_slots // self.seq_maxlen if self.n_mh_slots else 0
    @property
    def f_in(self): return len(self.multi_hot_flag)
    @property
    def f_out(self): return self.n_onehot + (self.n_mh_fields if self.method in ("mean", "sum") else self.n_mh_slots)
    @property
    def n_table_rows(self): return int(sum(self.field_dims))


def amazon_workload(dropout=0.2) -> Workload:
    """BASELINE configs[1..3]: Amazon 25 domains, 7 one-hot columns + 2 histories of 5 slots."""
    return Workload("amazon", [1368287, 7, 25, 45, 11, 22356, 10], [False] * 7 + [True] * 10, itemid_idx=0, domain_idx=2,
                    n_domain=25, domain_size=AMAZON_DOMAIN_SIZE, dropout=dropout, pos_rate=0.5)


def aliccp_workload(dropout=0.2) -> Workload:
    """BASELINE configs[4]: AliCCP 30 domains, 23 one-hot columns, no history (run.py:57-59; dims = bundled-sample max + 1)."""
    dims = [211161, 95, 14, 3, 8, 4, 4, 3, 5, 41775, 30, 284915, 81491, 112993, 1929, 118091, 54472, 34677,
            5821, 106908, 54295, 31716, 4]
    return Workload("aliccp", dims, [False] * 23, itemid_idx=9, domain_idx=10, n_domain=30, domain_size=ALICCP_DOMAIN_SIZE,
                    method=None, dropout=dropout, pos_rate=0.043)


def model_config(w: Workload, precision="f32"):
    """The attributes AREAD.__init__ reads from the reference's config namespace (aread.py:71-73,95; config.py)."""
    cfg = types.SimpleNamespace()
    cfg.aread_precision = precision
    cfg.dataset_name = w.name
    cfg.domain_size = {w.name: list(w.domain_size)}
    cfg.use_dcn, cfg.use_atten = True, True
    cfg.n_cross_layers, cfg.mmoe_n_expert = w.n_cross, w.n_expert
    cfg.atten_embed_dim, cfg.att_layer_num, cfg.att_head_num, cfg.att_res = 64, 3, 2, True
    return cfg


def build_model(w: Workload, device="cuda", precision="f32", seed=123):
    """aread_amd.AREAD of the workload's shapes with the module's own (torch-default) initialisation."""
    from .aread import AREAD
    torch.manual_seed(seed)
    mh = {"multi_hot_flag": list(w.multi_hot_flag), "itemid_idx": w.itemid_idx, "seq_maxlen": w.seq_maxlen, "method": w.method}
    model = AREAD(list(w.field_dims), w.embed_dim, mh, tuple(w.n_tower), w.n_domain, "mmoe", tuple(w.expert_dims),
                  tuple(tuple(t) for t in w.tower_dims), w.domain_idx, n_cross_layers=w.n_cross, dropout=w.dropout, device=device,
                  l2_reg_embedding=w.l2, l2_reg_linear=w.l2, l2_reg_dnn=w.l2, l2_reg_cross=w.l2, config=model_config(w, precision))
    return model.to(device)


def random_masks(model, p_active=0.7, seed=2000):
    """One random valid mask per domain from the model's own generator (generate_mask('rand'), aread.py:432-446) and the
    installed model.domain_mask; returns them (lists of bool tensors on the model's device)."""
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        masks = [model.generate_mask("rand", init_active_percent=p_active) for _ in range(model.n_domain)]
    finally:
        np.random.set_state(state)
    dev = model.dense.device
    masks = [[m.to(dev) if isinstance(m, torch.Tensor) else torch.as_tensor(np.asarray(m), dtype=torch.bool, device=dev) for m in mk]
             for mk in masks]
    model.domain_mask = masks
    return masks
