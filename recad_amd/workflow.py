"""Train -> attack -> inject -> retrain -> evaluate driver around the victim hot path.

The build's thin counterpart of recad/workflow/normal.py: same steps, same
configuration keys and the same calls into the victim (``.I(dataset=)``, ``.to(device)``,
``train_step(**info, progress_bar=)``, ``reset()``), so any attacker object exposing the
reference's ``generate_fake(target_id_list=...)`` / optional ``train_step`` plugs in
unmodified.  Evaluation uses the batched device path (recad_amd.evaluate) when the victim
offers ``scoring_tables()`` and the reference's per-user ``forward`` loop otherwise.
``execute()`` RETURNS the metrics (the reference only prints them).
"""
import time
from collections import OrderedDict
from copy import copy

import numpy as np
import torch

from .default import WORKFLOW
from .evaluate import eligible_users, eligible_users_device, full_catalog_topk, hit_counts, hr_rows, pred_shift
from .utils import NullProgress, get_logger


class RandomAttack:
    """Heuristic random-filler attacker (semantics of recad/model/attacker/heuristic.py:10-68):
    attack_num fake profiles, each target rated 5 by its share of the profiles, filler_num
    other items rated round(N(mean, std)) clipped to [1, 5].  Never touches the victim."""

    model_name = "random"

    def __init__(self, n_items, attack_num=50, filler_num=36, rating_mean=3.6, rating_std=1.1, seed=None):
        self.n_items, self.attack_num, self.filler_num = n_items, attack_num, filler_num
        self.mean, self.std = rating_mean, rating_std
        self.rng = np.random.default_rng(np.random.randint(0, 2 ** 31 - 1) if seed is None else seed)

    def I(self, **kw):
        return self

    def to(self, device):
        return self

    def input_describe(self):
        return {"generate_fake": {"target_id_list": (list, [])}}

    def generate_fake(self, target_id_list, **config):
        fake = np.zeros((self.attack_num, self.n_items), dtype=float)
        rate = int(self.attack_num / len(target_id_list))
        for k, t in enumerate(target_id_list):
            fake[k * rate:(k + 1) * rate, t] = 5
        pool = np.setdiff1d(np.arange(self.n_items), np.asarray(target_id_list))
        for r in range(self.attack_num):
            cols = self.rng.choice(pool, size=self.filler_num, replace=False)
            vals = np.clip(np.round(self.rng.normal(self.mean, self.std, self.filler_num)), 1, 5)
            fake[r, cols] = vals
        return fake


class Normal:
    def __init__(self, **config):
        self.c = config
        self.attacker = config["attacker"].I(dataset=config["attack_data"])
        self.victim = config["victim"].I(dataset=config["victim_data"])
        self.victim_data = config["victim_data"]
        self.logger = get_logger(__name__, level=config["logging_level"])
        self.timings = OrderedDict()

    @classmethod
    def from_config(cls, **kwargs):
        need = ("victim_data", "attack_data", "victim", "attacker")
        if any(k not in kwargs for k in need):
            raise TypeError(f"Expect for user arguments [{', '.join(need)}]")  # workflow/base.py:17-18
        config = {k: copy(v) for k, v in WORKFLOW["no defense"].items()}
        for k, v in kwargs.items():
            if k in config or k in need:
                config[k] = v
        return cls(**config)

    def info_describe(self):
        return {"target_id_list": self.c["target_id_list"], "input_describe": {}}

    # ------------------------------------------------------------------ train (normal.py:95-109)
    def normal_train(self, model, epoch):
        progress = NullProgress()
        losses = []
        for _ in range(epoch):
            loss = model.train_step(**self.info_describe(), progress_bar=progress)
            out_des = model.output_describe()["train_step"]
            assert len(loss) == len(out_des), \
                f"The output describe is not aligned with the actual output of train_step for {model.model_name}"
            losses.append(loss)
        return losses

    # ------------------------------------------------------------------ evaluate (normal.py:57-160)
    def _rows(self, model, dataset, users, targets, topks):
        ptr, idx = dataset.train_csr_sorted()
        if hasattr(model, "scoring_tables") or hasattr(model, "score_matrix"):
            res = full_catalog_topk(model, users, ptr, idx, targets, K=max(100, max(topks)))
            return hr_rows(users, res, topks), res
        # foreign victim: the reference's own per-user forward loop
        rows = []
        dev = self.c["device"]
        with torch.no_grad():
            for u in users:
                seen = np.zeros(dataset.n_items, dtype=bool)
                seen[idx[ptr[u]:ptr[u + 1]]] = True
                iids = torch.from_numpy(np.nonzero(~seen)[0]).to(dev)
                s = model(torch.full_like(iids, int(u)), iids).cpu().numpy()
                for t in targets:
                    st = float(s[np.nonzero(iids.cpu().numpy() == t)[0][0]])
                    rank = int((s > st).sum())
                    rows.append([u, st] + [1.0 if rank < k else 0.0 for k in topks])
        return np.asarray(rows, dtype=np.float64), None

    @staticmethod
    def _batched(model):
        return hasattr(model, "scoring_tables") or hasattr(model, "score_matrix")

    def normal_evaluate(self, model, model_fake, dataset, target_id_list, topks):
        for m in (model, model_fake):
            fwd = m.input_describe()["forward"]
            assert len(fwd) == 2 and "users" in fwd and "items" in fwd, "Expect forward(users, items)"
        ptr, idx = dataset.train_csr_sorted()
        if self._batched(model) and self._batched(model_fake):
            return self._evaluate_on_device(model, model_fake, ptr, idx, target_id_list, topks)
        users = eligible_users(ptr, idx, target_id_list)
        rows, _ = self._rows(model, dataset, users, target_id_list, topks)
        rows_fake, _ = self._rows(model_fake, dataset, users, target_id_list, topks)
        assert np.allclose(rows[:, 0], rows_fake[:, 0]), "Users are not aligned"
        results = OrderedDict()
        results["pred_shift"] = float(np.mean(rows_fake[:, 1] - rows[:, 1]))
        for i, k in enumerate(topks):
            results[f"HR@{k}"] = float(np.mean(rows[:, 2 + i]))
            results[f"HR@{k} after attack"] = float(np.mean(rows_fake[:, 2 + i]))
        results["n_eval_users"] = int(len(users))
        return results

    def _evaluate_on_device(self, model, model_fake, ptr, idx, target_id_list, topks):
        """normal.py:111-160 without host round trips of per-user data: the target-present filter
        (rk_eligible_users), scoring + top-K + target ranks (rk_score_topk / rk_topk_rows), the HR@k numerators
        (rk_hit_counts) and pred_shift (rk_pred_shift) all stay in HBM; the host reads back the eligible-user
        count and, at the end, 2*T*len(topks) integers and one double."""
        dev = next(model.parameters()).device
        users, ptr_d, idx_d, tg_d = eligible_users_device(ptr, idx, target_id_list, dev)
        K = max(100, max(topks))
        n, T = int(users.numel()), int(tg_d.numel())
        results = OrderedDict()
        if n == 0:
            results["pred_shift"] = float("nan")
            for k in topks:
                results[f"HR@{k}"] = results[f"HR@{k} after attack"] = float("nan")
            results["n_eval_users"] = 0
            return results
        res = full_catalog_topk(model, users, ptr_d, idx_d, tg_d, K=K, to_host=False)
        res_f = full_catalog_topk(model_fake, users, ptr_d, idx_d, tg_d, K=K, to_host=False)
        hits, hits_f = hit_counts(res["target_rank"], topks), hit_counts(res_f["target_rank"], topks)
        shift = pred_shift(res["target_score"], res_f["target_score"])
        packed = torch.cat([hits.double().view(-1), hits_f.double().view(-1), shift]).cpu().numpy()   # the ONE read-back
        h = packed[: T * len(topks)].reshape(T, len(topks)).sum(axis=0)
        hf = packed[T * len(topks): 2 * T * len(topks)].reshape(T, len(topks)).sum(axis=0)
        results["pred_shift"] = float(packed[-2])
        for i, k in enumerate(topks):
            results[f"HR@{k}"] = float(h[i] / (n * T))
            results[f"HR@{k} after attack"] = float(hf[i] / (n * T))
        results["n_eval_users"] = n
        self.last_eval = {"users": users, "clean": res, "poisoned": res_f}
        return results

    # ------------------------------------------------------------------ normal.py:162-225
    def execute(self):
        dev = self.c["device"]
        tick = time.time
        self.victim = self.victim.to(dev)
        self.attacker = self.attacker.to(dev)
        t0 = tick()
        self.losses = self.normal_train(self.victim, self.c["rec_epoch"])
        self.timings["train_clean_s"] = tick() - t0
        if "train_step" in self.attacker.input_describe():
            self.normal_train(self.attacker, self.c["attack_epoch"])
        fake_array = self.attacker.generate_fake(**self.info_describe())
        fake_dataset = self.victim_data.inject_data("explicit", fake_array, filter_num=self.c["filter_num"])
        fake_victim = self.victim.reset().I(dataset=fake_dataset).to(dev)
        t0 = tick()
        self.losses_fake = self.normal_train(fake_victim, self.c["rec_epoch"])
        self.timings["train_poisoned_s"] = tick() - t0
        self.fake_victim, self.fake_dataset = fake_victim, fake_dataset
        t0 = tick()
        results = self.normal_evaluate(self.victim, fake_victim, self.victim_data, self.c["target_id_list"], self.c["topks"])
        self.timings["evaluate_s"] = tick() - t0
        self.results = results
        try:
            from tabulate import tabulate
            print(tabulate([(k, v) for k, v in results.items()], headers=["metric", "value"], tablefmt="fancy_grid"))
        except Exception:
            print(results)
        return results


class Defense(Normal):
    """Counterpart of recad/workflow/defense.py:177-303: Normal's steps, then the defender flags
    users (``defense_step() -> ids``), the flagged users are dropped from the poisoned dataset
    (``delete_data``), the victim is retrained a THIRD time from a fresh init and evaluated again.
    RNGs are re-seeded before the retrains like the reference does (defense.py:102-106,222,281).
    The defender is any object with the reference's interface (``I``, ``to``, ``input_describe``,
    optional ``train_step``, ``defense_step``); it never calls the victim."""

    def __init__(self, **config):
        super().__init__(**config)
        self.defender = config["defender"].I(dataset=config.get("defense_data", config["attack_data"]))

    @classmethod
    def from_config(cls, **kwargs):
        need = ("victim_data", "attack_data", "victim", "attacker", "defender")
        if any(k not in kwargs for k in need):
            raise TypeError(f"Expect for user arguments [{', '.join(need)}]")
        config = {k: copy(v) for k, v in WORKFLOW["defense"].items()}
        for k, v in kwargs.items():
            if k in config or k in need or k == "defense_data":
                config[k] = v
        return cls(**config)

    @staticmethod
    def random_seed_set(seed=None):
        import random
        from .default import SEED
        seed = SEED if seed is None else seed
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)

    def execute(self):
        dev = self.c["device"]
        self.victim = self.victim.to(dev)
        self.attacker = self.attacker.to(dev)
        self.defender = self.defender.to(dev)
        self.losses = self.normal_train(self.victim, self.c["rec_epoch"])
        if "train_step" in self.attacker.input_describe():
            self.normal_train(self.attacker, self.c["attack_epoch"])
        fake_array = self.attacker.generate_fake(**self.info_describe())
        fake_dataset = self.victim_data.inject_data("explicit", fake_array, filter_num=self.c["filter_num"])
        self.random_seed_set()
        fake_victim = self.victim.reset().I(dataset=fake_dataset).to(dev)
        self.normal_train(fake_victim, self.c["rec_epoch"])
        results = OrderedDict()
        results["attacked"] = self.normal_evaluate(self.victim, fake_victim, self.victim_data, self.c["target_id_list"], self.c["topks"])
        if "train_step" in self.defender.input_describe():
            self.normal_train(self.defender, self.c["defense_epoch"])
        flagged = list(self.defender.defense_step())
        cleaned = self.victim_data.delete_data("explicit", flagged, fake_array, filter_num=self.c["filter_num"])
        self.random_seed_set()
        defended_victim = self.victim.reset().I(dataset=cleaned).to(dev)
        self.normal_train(defended_victim, self.c["rec_epoch"])
        results["defended"] = self.normal_evaluate(self.victim, defended_victim, self.victim_data, self.c["target_id_list"], self.c["topks"])
        results["n_flagged"] = len(flagged)
        self.fake_dataset, self.cleaned_dataset, self.defended_victim = fake_dataset, cleaned, defended_victim
        self.results = results
        return results


factories = {"no defense": Normal, "defense": Defense}


def from_config(name, **kwargs):
    """workflow.from_config("no defense", **cfg) (recad/workflow/__init__.py:6-7)."""
    return factories[name].from_config(**kwargs)
