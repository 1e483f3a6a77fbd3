"""Training / evaluation harness for AREAD: the counterpart of the reference's Run.train_aread, Run.test,
Run.is_continuable and Run.main (run.py:459-484, 578-686, 712-808, 821-852) without pandas/wandb/tqdm.

It drives any model that exposes the reference's AREAD interface (forward modes, get_regularization_loss,
generate_mask / prun_single_mask / add_eval_loss / update_all_mask / save|load_model_state,
reset_for_mask_update), so the very same loop runs the HIP model on the GPU and -- in
tests/golden/make_golden.py only -- the reference model on the CPU to record golden traces.

Schedule reproduced (run.py:578-686):
  * epoch 0 warm-up: (warm_up_interval*1024)//bs steps, domains round-robin from the end of the list,
    mode='wo_mask' with gate recording, BCE + reg;
  * every (regroup_interval*1024)//bs steps and at step 0 of epoch 0: HEMP regroup -- decay sigma*0.99,
    active%*0.95 (floor 0.1), candidates*0.99; per domain and candidate: generate_mask('mask_max_gate'),
    restore snapshot, regroup_update_step fast Adam(update_lr) steps on the augmented stream with 5 % pruning,
    regroup_eval_step no-grad eval steps in train() mode; then select, reset, restore;
  * otherwise: mode='domain_mask_bagging', loss = mean_k BCE(p_k, y) + reg, Adam step.
"""
import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


class DomainStreams:
    """Per-domain shuffled loaders with restart-on-exhaustion (run.py:310-353, 540-575)."""

    def __init__(self, X, y, n_domain, domain_idx, bs, device, shuffle_seq=True):
        self.n_domain, self.bs = n_domain, bs
        self.loaders, self.batch_seq = [], []
        for d in range(n_domain):
            m = X[:, domain_idx] == d
            Xd, yd = X[m].to(device), y[m].to(device)
            self.loaders.append(DataLoader(TensorDataset(Xd, yd), bs, shuffle=True))
            self.batch_seq.extend([d] * int(np.ceil(Xd.shape[0] * 1.0 / bs)))
        cnt = torch.bincount(X[:, domain_idx].long(), minlength=n_domain)
        self.domain_cnt_weight = np.array([float(cnt[i]) / X.shape[0] for i in range(n_domain)])
        if shuffle_seq:
            np.random.shuffle(self.batch_seq)
        self.iters = [iter(l) for l in self.loaders]

    def next(self, d):
        try:
            return next(self.iters[d])
        except StopIteration:
            self.iters[d] = iter(self.loaders[d])
            return next(self.iters[d])


def _auc(t, p):
    """ROC AUC by rank statistic (ties averaged): same value as sklearn.metrics.roc_auc_score."""
    t = np.asarray(t, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    n_pos, n_neg = t.sum(), (1 - t).sum()
    if n_pos == 0 or n_neg == 0:
        raise ValueError("only one class present")
    order = np.argsort(p, kind="mergesort")
    ranks = np.empty(len(p), dtype=np.float64)
    sp = p[order]
    i = 0
    while i < len(sp):
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[t == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def _logloss(t, p):
    """sklearn.metrics.log_loss for binary labels (probabilities clipped to float eps)."""
    t = np.asarray(t, dtype=np.float64)
    eps = np.finfo(np.asarray(p).dtype if np.asarray(p).dtype.kind == "f" else np.float64).eps
    p = np.clip(np.asarray(p, dtype=np.float64), eps, 1 - eps)
    if len(np.unique(t)) < 2:
        raise ValueError("only one class present")
    return float(-(t * np.log(p) + (1 - t) * np.log(1 - p)).mean())


class Trainer:
    def __init__(self, model, cfg, train, valid, test=None, aug=None, device="cuda", log=print):
        self.model, self.cfg, self.device, self.log = model, cfg, device, log
        self.train_s, self.valid_s, self.test_s, self.aug_s = train, valid, test, aug if aug is not None else train
        self.n_domain = train.n_domain
        self.random_modify_sigma = cfg.random_modify_sigma
        self.init_active_percent = cfg.init_active_percent
        self.candidate_mask_num = cfg.candidate_mask_num
        self.regroup_times = 0
        self.best_mean_auc, self.best_auc, self.best_loss = 0, 0, 0
        self.trial_counter, self.num_trials = 0, getattr(cfg, "early_stop", 2)
        self.trace = []                       # (tag, value) pairs: the golden trace

    def _bagging_loss(self, preds, y, criterion):
        y = y.squeeze().float()
        return sum(criterion(p, y) for p in preds.unbind(dim=0)) / preds.shape[0]

    def train_aread(self, criterion, optimizer, epoch_i):
        """run.py:578-686"""
        model, cfg = self.model, self.cfg
        model.train()
        warm_up = int((cfg.warm_up_interval * 1024) // cfg.bs)
        regroup = int((cfg.regroup_interval * 1024) // cfg.bs)
        if epoch_i == 0:
            domain_list = list(range(self.n_domain))
            for _ in range(warm_up):
                if not domain_list:
                    domain_list = list(range(self.n_domain))
                d = domain_list.pop()
                X, y = self.train_s.next(d)
                pred = model(X, mode="wo_mask", domain_i=d, memory_gate_value=True)
                loss = criterion(pred.squeeze(), y.squeeze().float()) + model.get_regularization_loss(device=self.device)
                model.zero_grad()
                loss.backward()
                optimizer.step()
                self.trace.append(("warmup_loss", float(loss.detach())))
        for i, d in enumerate(self.train_s.batch_seq):
            X, y = self.train_s.next(d)
            if (epoch_i == 0 and i == 0) or ((i + 1) % regroup == 0):
                self._regroup(criterion)
                # reference quirk kept (run.py:625,663): the regroup loop re-binds `d`, so the step right after a
                # regroup runs this batch through the mask of the LAST domain
                d = self.n_domain - 1
            preds = model(X, mode="domain_mask_bagging", domain_i=d)
            loss = self._bagging_loss(preds, y, criterion) + model.get_regularization_loss(device=self.device)
            model.zero_grad()
            loss.backward()
            optimizer.step()
            self.trace.append(("train_loss", float(loss.detach())))

    def _regroup(self, criterion):
        """run.py:612-661"""
        model, cfg = self.model, self.cfg
        model.save_model_state()
        self.random_modify_sigma *= 0.99
        self.init_active_percent = max(0.1, self.init_active_percent * 0.95)
        self.candidate_mask_num *= 0.99
        n_cand = max(1, int(self.candidate_mask_num))
        self.regroup_times += 1
        for d in range(self.n_domain):
            for z in range(n_cand):
                mask = model.generate_mask(generate_mode="mask_max_gate", d=d, init_active_percent=self.init_active_percent,
                                           random_modify_sigma=self.random_modify_sigma)
                model.load_model_state()
                fast = torch.optim.Adam(model.parameters(), lr=cfg.update_lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=cfg.wd)
                for _ in range(cfg.regroup_update_step):
                    X, y = self.aug_s.next(d)
                    preds = model(X, mode="domain_mask_bagging", current_mask=mask, tmp_memory_gate_value=True, domain_i=d)
                    loss = self._bagging_loss(preds, y, criterion) + model.get_regularization_loss(device=self.device)
                    model.zero_grad()
                    loss.backward()
                    fast.step()
                    mask = model.prun_single_mask(d, mask, prun_ratio=0.05)
                model.candidate_domain_mask[d].append(mask)
                with torch.no_grad():
                    for _ in range(cfg.regroup_eval_step):
                        X, y = self.train_s.next(d)
                        pred = model(X, mode="domain_with_mask", current_mask=mask, domain_i=d)
                        loss = criterion(pred.squeeze(), y.squeeze().float()) + model.get_regularization_loss(device=self.device)
                        model.add_eval_loss(loss.mean().item(), d=d, mask_z=z)
                        self.trace.append(("eval_loss", float(loss.detach())))
        model.update_all_mask(regroup_times=self.regroup_times)
        model.reset_for_mask_update()
        model.load_model_state()
        self.trace.append(("mask_edges", float(sum(model.count_active_edge(d=d) for d in range(self.n_domain)))))

    def test(self, mode="valid"):
        """run.py:712-763 + evaluate_multi_domain (run.py:787-808)"""
        model = self.model
        streams = self.valid_s if mode == "valid" else self.test_s
        model.eval()
        targets, predicts, domains = [], [], []
        with torch.no_grad():
            for d in streams.batch_seq:
                X, y = streams.next(d)
                pred = model(X, mode="domain_with_mask", domain_i=d)
                targets.append(y.reshape(-1).cpu().numpy())
                predicts.append(pred.reshape(-1).cpu().numpy())
                domains.append(X[:, model.domain_idx].cpu().numpy())
        t, p, dm = np.concatenate(targets), np.concatenate(predicts), np.concatenate(domains)
        res = {"total_auc": _auc(t, p), "total_loss": _logloss(t, p), "domain_auc": {}, "domain_loss": {}}
        mean_auc = mean_loss = 0.0
        for d in np.unique(dm):
            sel = dm == d
            try:
                a, l = _auc(t[sel], p[sel]), _logloss(t[sel], p[sel])
            except ValueError:
                a, l = np.nan, np.nan
            res["domain_auc"][int(d)], res["domain_loss"][int(d)] = a, l
            mean_auc += self.train_s.domain_cnt_weight[int(d)] * a
            mean_loss += self.train_s.domain_cnt_weight[int(d)] * l
        res["mean_auc"], res["mean_loss"] = mean_auc, mean_loss
        return res

    def is_continuable(self, result, epoch_i, optimizer, save_path=None):
        """run.py:459-484: checkpoint on a better train-frequency-weighted mean AUC, patience num_trials."""
        if result["mean_auc"] > self.best_mean_auc:
            self.trial_counter = 0
            self.best_auc, self.best_loss = result["total_auc"], result["total_loss"]
            self.best_mean_auc, self.best_mean_loss = result["mean_auc"], result["mean_loss"]
            if save_path:
                torch.save({"epoch": epoch_i + 1, "state_dict": self.model.state_dict(), "best_auc": self.best_auc,
                            "best_result": result, "preprocess_path": None, "optimizer": optimizer.state_dict(),
                            "best_mean_auc": self.best_mean_auc, "best_mean_loss": self.best_mean_loss,
                            "domain_mask": self.model.domain_mask}, save_path)
            return True
        if self.trial_counter + 1 < self.num_trials:
            self.trial_counter += 1
            return True
        return False

    def main(self, epochs, save_path=None):
        """run.py:821-852"""
        cfg = self.cfg
        optimizer = torch.optim.Adam(self.model.parameters(), lr=cfg.lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=cfg.wd)
        criterion = torch.nn.BCELoss(reduction="mean")
        results = []
        for epoch_i in range(epochs):
            self.train_aread(criterion, optimizer, epoch_i)
            res = self.test("valid")
            results.append(res)
            self.trace.append(("valid_auc", res["total_auc"]))
            self.trace.append(("valid_logloss", res["total_loss"]))
            self.log(f"epoch {epoch_i + 1}: auc {res['total_auc']:.4f} loss {res['total_loss']:.4f} "
                     f"mean_auc {res['mean_auc']:.4f}")
            if not self.is_continuable(res, epoch_i, optimizer, save_path):
                break
        return results
