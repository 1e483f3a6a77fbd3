"""HEMP (hierarchical expert mask pruning) host logic: candidate generation, validity closure, quantile
pruning, selection and the in-memory parameter snapshot -- the methods `Run.train_aread` calls on the model
(run.py:612-661).  Behavioural mirror of model/aread.py:324-680; tiny host tensors, no kernels.

The masks themselves are consumed by the HIP path as wave-uniform predicates; the per-tower gate
statistics these methods read are produced by the kernels (k_mixl -> gate_stats).

Random streams: the same numpy / torch generator calls, in the same order and with the same shapes as
the reference, so that a run seeded like main.py:51-57 produces the same mask sequence
(tests/test_hemp_cpu.py pins this against sequences recorded from the reference).
"""
import copy
import re

import numpy as np
import torch

_SNAPSHOT_PREFIXES = ("cn", "cgc_layers", "towers", "tower_gates", "towers_linear", "output_layers", "embedding",
                      "linear", "reg_loss", "regularization_weight")     # aread.py:535-536 (mmoe_* is NOT in the list)


def _any(v):
    return bool(v.any().item()) if isinstance(v, torch.Tensor) else bool(np.any(v))


class HempMixin:
    # ---- shapes -------------------------------------------------------------------------------------
    def _mask_shapes(self):
        n = self.n_tower
        return [(1, n[0])] + [(n[l - 1], n[l]) for l in range(1, self.n_level)] + [(n[-1], 1)]

    def create_single_full_mask(self, fill_value=0):
        """aread.py:548-568: all-False / all-True / Bernoulli(fill_value) numpy masks, one array per level."""
        shapes = self._mask_shapes()
        if fill_value == 0:
            return [np.zeros(s, dtype=bool) for s in shapes]
        if fill_value == 1:
            return [np.ones(s, dtype=bool) for s in shapes]
        if 0 < fill_value < 1:
            return [np.random.choice([True, False], s, p=[fill_value, 1 - fill_value]) for s in shapes]
        raise ValueError("fill_value in mask must be 0 or 1 or (0, 1)")

    # ---- validity closure ----------------------------------------------------------------------------
    def validate_mask(self, mask, add_input=True, add_output=True, remove_hidden=True):
        """aread.py:570-605.  In place on `mask` (list of arrays or tensors); returns it.
        (1) a level-0 tower that feeds someone gets its input edge; (2) a last-level tower that is fed gets
        its output edge; (3) worklist over the towers of levels >= 1: no inputs -> cut outputs; no outputs ->
        re-examine its feeders (levels >= 2) and cut an input column.
        Reference quirk kept on purpose (aread.py:601-604): when feeders were re-queued, the column that is
        cut is indexed by the LAST FEEDER's index, not by the tower itself."""
        n = self.n_tower
        if add_input:
            for t in range(n[0]):
                if _any(mask[1][t, :]):
                    mask[0][:, t] = True
        if add_output:
            for t in range(n[-1]):
                if _any(mask[-2][:, t]):
                    mask[-1][t, :] = True
        if remove_hidden:
            work = [(l, t) for l in range(1, self.n_level) for t in range(n[l])]
            while work:
                l, t = work.pop(0)
                if not _any(mask[l][:, t]):
                    mask[l + 1][t, :] = False
                if not _any(mask[l + 1][t, :]):
                    cut = t
                    if l > 1:
                        col = mask[l][:, t]
                        feeders = (col.nonzero()[:, 0].tolist() if isinstance(col, torch.Tensor)
                                   else np.nonzero(col)[0].tolist())
                        for f in feeders:
                            if (l - 1, f) not in work:
                                work.append((l - 1, f))
                        if feeders:
                            cut = feeders[-1]
                    mask[l][:, cut] = False
        return mask

    # ---- bookkeeping ---------------------------------------------------------------------------------
    def add_eval_loss(self, loss_mean, d, mask_z):
        if len(self.eval_loss[d]) <= mask_z:
            self.eval_loss[d].append([loss_mean])
        else:
            self.eval_loss[d][mask_z].append(loss_mean)

    def count_active_edge(self, d=None, d_mask=None):
        mask = d_mask if d_mask is not None else self.domain_mask[d]
        if isinstance(mask[0], torch.Tensor):
            return sum(torch.sum(m).cpu().item() for m in mask)
        return sum(np.sum(m) for m in mask)

    def count_current_active_ratio(self):
        return sum(self.count_active_edge(d=d) * 1.0 / self.edge_num for d in range(self.n_domain)) / self.n_domain

    def print_domain_mask(self, d_mask=None, d=None, all_edges=False):
        mask = d_mask if d_mask is not None else self.domain_mask[d]
        mask = [m.cpu().numpy() if isinstance(m, torch.Tensor) else m for m in mask]
        print("level 0 towers:", np.nonzero(mask[0])[1])
        for l in range(1, self.n_level):
            if all_edges:
                print(f"========= level {l} =========")
                for t in range(self.n_tower[l]):
                    src = np.nonzero(mask[l][:, t])[0]
                    print(f"last level input towers of tower {t}:", src if src.size else None)
            else:
                print(f"level {l} used last level towers:", np.nonzero(np.any(mask[l], axis=1))[0])
        print("the last level output towers:", np.nonzero(mask[-1])[0])

    # ---- gate statistics -> thresholds -------------------------------------------------------------------
    def _quantile_floor(self, mats, q):
        """min over the given matrices of quantile(values > 1e-8, q); 1 when nothing qualifies."""
        thr = 1
        for m in mats:
            if (m > 1e-8).any():
                thr = min(thr, torch.quantile(m[m > 1e-8].flatten(), q))
        return thr

    def mean_domain_tower_gate_values(self, d, get_threshold=None):
        """aread.py:403-430: collapse the recorded per-step gate means of domain d to one matrix per level and,
        if asked, derive the activity threshold (quantile 1 - get_threshold of the non-zero means)."""
        rec = self.domain_tower_gate_values[d]
        if not isinstance(rec[0], list):
            return
        dev = self.device
        mats = [torch.zeros(1, self.n_tower[0], dtype=torch.float32, device=dev)]
        for l in range(1, self.n_level):
            cols = []
            for t in range(self.n_tower[l]):
                hist = rec[l][t]
                cols.append(torch.mean(torch.stack(hist, dim=0), dim=0) if len(hist) else
                            torch.zeros(self.n_tower[l - 1], dtype=torch.float32, device=dev))
            mats.append(torch.stack(cols, dim=1))
        mats.append(torch.zeros(self.n_tower[-1], 1, dtype=torch.float32, device=dev))
        self.domain_tower_gate_values[d] = mats
        if get_threshold is not None:
            thr = self._quantile_floor(mats[1:-1], 1 - get_threshold)
            self.gate_value_threshold[d] = None if thr == 1 else thr

    # ---- pruning ---------------------------------------------------------------------------------------
    def prun_single_mask(self, d, current_mask, prun_ratio=0.05):
        """aread.py:357-381: drop the edges whose recorded gate mean is below the prun_ratio-quantile, re-validate,
        keep the previous mask if no head survives.  Mutates current_mask (as the reference does)."""
        gates = [torch.stack(self.tmp_tower_gate_values[l], dim=1) for l in range(1, self.n_level)]
        thr = self._quantile_floor(gates, prun_ratio)
        if thr == 1:
            self.print_domain_mask(current_mask, all_edges=True)
            raise ValueError("no valid tmp_tower_gate_values in candidate mask")
        before = copy.deepcopy(current_mask)
        for l in range(1, self.n_level):
            current_mask[l] = current_mask[l] & (gates[l - 1] >= thr)
        valid = self.validate_mask(current_mask)
        self.tmp_tower_gate_values = [[None for _ in range(self.n_tower[l])] for l in range(self.n_level)]
        return valid if _any(valid[-1]) else before

    # ---- candidates ------------------------------------------------------------------------------------
    def _to_tensors(self, mask):
        return [torch.tensor(m, dtype=torch.bool, device=self.device) for m in mask]

    def generate_mask(self, generate_mode="rand", d=None, init_active_percent=0.7, random_modify_sigma=0.2):
        """aread.py:432-532."""
        nl = self.n_level + 1
        if generate_mode == "rand":
            while True:
                valid = self.validate_mask(self.create_single_full_mask(fill_value=init_active_percent))
                if _any(valid[-1]):
                    return self._to_tensors(valid)
        if generate_mode == "mask_norm_rand":
            origin = [self.domain_mask[d][l].cpu().numpy() for l in range(nl)]
            n_active = self.count_active_edge(d_mask=origin)
            while True:
                pct = min(1, np.abs(np.random.normal(0, random_modify_sigma)))
                grow = n_active < self.edge_num * pct
                cand = []
                for l in range(nl):
                    flip = np.random.rand(*origin[l].shape) < pct
                    cand.append(origin[l] | flip if grow else origin[l] ^ flip)
                valid = self.validate_mask(cand)
                changed = any(not np.all(valid[l] == origin[l]) for l in range(nl))
                if changed and _any(valid[-1]):
                    return self._to_tensors(valid)
        if generate_mode in ("max_gate", "max_gate_norm_rand", "mask_max_gate"):
            if not any(self.domain_tower_gate_values):
                raise ValueError("tower_gate_values is None")
            self.mean_domain_tower_gate_values(d, get_threshold=init_active_percent)
            thr = self.gate_value_threshold[d]
            if generate_mode != "mask_max_gate":
                if thr is None:
                    return self.generate_mask("rand", d, init_active_percent, random_modify_sigma)
                strong = [t >= thr for t in self.domain_tower_gate_values[d]]
                if generate_mode == "max_gate":
                    valid = self.validate_mask(strong)
                    if not _any(valid[-1]):
                        raise ValueError(f"mask generated for domain {d} in the 'max_gate' mode has no output")
                    return valid
                pct = min(1, np.abs(np.random.normal(0, random_modify_sigma)))
                while True:
                    cand = [strong[l] ^ (torch.rand(strong[l].shape, device=self.device) < pct) for l in range(nl)]
                    valid = self.validate_mask(cand)
                    if _any(valid[-1]):
                        return valid
            # 'mask_max_gate' (the mode the training loop uses, run.py:628-630)
            if thr is None:
                strong = self.generate_mask("rand", d, init_active_percent, random_modify_sigma)
            else:
                strong = [t >= thr for t in self.domain_tower_gate_values[d]]
            pct = min(1, np.abs(np.random.normal(0, random_modify_sigma)))
            origin = self.domain_mask[d] if self.domain_mask[d] is not None else strong
            dense = (self.count_active_edge(d_mask=origin) * 1.0 / self.edge_num) > init_active_percent
            while True:
                cand = []
                for l in range(nl):
                    flip = torch.rand(strong[l].shape, device=self.device) < pct
                    merged = origin[l] | strong[l]
                    cand.append(merged ^ flip if dense else merged | flip)
                valid = self.validate_mask(cand)
                changed = any(not torch.all(valid[l] == origin[l]) for l in range(nl))
                if changed and _any(valid[-1]):
                    return valid
        raise ValueError(f"unknown generate_mode {generate_mode!r}")

    # ---- selection ---------------------------------------------------------------------------------------
    def update_all_mask(self, regroup_times=None, update_mode="best4single_domain"):
        """aread.py:330-355: per domain keep the candidate with the lowest mean evaluation loss."""
        if update_mode != "best4single_domain":
            return
        means, stds = [], []
        for d in range(self.n_domain):
            per_mask = [np.mean(self.eval_loss[d][z]) for z in range(len(self.candidate_domain_mask[0]))]
            self.domain_mask[d] = self.candidate_domain_mask[d][int(np.argmin(per_mask))]
            means.append(np.mean(per_mask))
            stds.append(np.std(per_mask))
        print("\n============Update Mask============")
        print("regroup_times: ", regroup_times, "current domain mask active ratio: ", self.count_current_active_ratio())
        print(f"loss_mean of different domain masks: {means}")
        print(f"loss_std of different domain masks: {stds}")
        users = [[] for _ in range(self.n_tower[1])] if self.n_level > 1 else []
        for d in range(self.n_domain):
            if self.n_level > 1:
                for t in torch.nonzero(torch.any(self.domain_mask[d][1], dim=0))[:, 0].cpu().numpy():
                    users[t].append(d)
        print(f"active domain num of each tower in the middle layer: {[len(u) for u in users]}")
        print(f"sample size training each tower in the middle layer: {[sum(self.domain_size[u]) for u in users]}")
        print("============Finish Update Mask============")

    # ---- snapshot ------------------------------------------------------------------------------------------
    def save_model_state(self):
        """aread.py:534-543: deep copy of the state_dict entries whose key starts with one of the listed
        prefixes.  mmoe_experts / mmoe_gates are not listed, so fast-update steps leak into the MMoE bottom
        (SURVEY 0.9): reproduced, not fixed."""
        pat = re.compile("^(" + "|".join(_SNAPSHOT_PREFIXES) + ")")
        self.model_state = copy.deepcopy({k: v for k, v in self.state_dict().items() if pat.match(k)})

    def load_model_state(self):
        self.load_state_dict(self.model_state, strict=False)
