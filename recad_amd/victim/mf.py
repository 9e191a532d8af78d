"""MF victim on MI355X: same interface as recad/model/victim/mf.py, hot path in HIP."""
import torch
import torch.nn as nn

from .. import _lib
from ..utils import VarDim, pick_optim
from .base import BaseVictim


class MF(BaseVictim):
    victim_name = "mf"

    def _build(self, factor_num, embedding_size, dropout, **config):
        info = config["dataset"].info_describe()
        num_users, num_items = info["n_users"], info["n_items"]
        self.dataset = config["dataset"]
        self.config = config
        if dropout:
            raise ValueError("dropout is not supported by the HIP path (reference default is 0)")
        # same construction / RNG order as mf.py:16-24
        self.user_emb = nn.Embedding(num_users, embedding_size)
        self.user_bias = nn.Embedding(num_users, 1)
        self.item_emb = nn.Embedding(num_items, embedding_size)
        self.item_bias = nn.Embedding(num_items, 1)
        self.user_emb.weight.data.uniform_(0, 0.005)
        self.user_bias.weight.data.uniform_(-0.01, 0.01)
        self.item_emb.weight.data.uniform_(0, 0.005)
        self.item_bias.weight.data.uniform_(-0.01, 0.01)
        self.mean = nn.Parameter(torch.FloatTensor([factor_num]), False)  # mf.py:26: constant offset
        self.dropout = nn.Dropout(dropout)
        self.optimizer = pick_optim(config["optim"])(self.parameters(), lr=config["lr"])
        self.num_users, self.num_items, self.dim = num_users, num_items, embedding_size
        self._mom = None
        self._t = 0

    def _tables(self):
        ts = (self.user_emb.weight, self.item_emb.weight, self.user_bias.weight, self.item_bias.weight)
        if ts[0].device.type != "cuda":
            raise _lib.HipCallError("MF parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        return ts

    def _flat_state(self, dev):
        tot = (self.num_users + self.num_items) * (self.dim + 1)
        if self._mom is None or self._mom[0].device != dev:
            self._mom = tuple(torch.zeros(tot, device=dev, dtype=torch.float32) for _ in range(3))  # m, v, grads
        return self._mom

    def forward(self, users, items):
        ue, ie, ub, ib = self._tables()
        out = torch.empty(users.numel(), device=ue.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_pair_scores(
            self.dim, _lib.ptr(ue.data), _lib.ptr(ie.data), _lib.ptr(ub.data), _lib.ptr(ib.data), float(self.mean.item()),
            _lib.ptr(users.long().contiguous()), _lib.ptr(items.long().contiguous()), users.numel(), _lib.ptr(out),
            _lib.stream_ptr()), "rk_pair_scores")
        return out

    def _run_epoch(self, users, items, labels, batch, apply_update=True):
        ue, ie, ub, ib = self._tables()
        if not isinstance(self.optimizer, torch.optim.Adam):
            raise NotImplementedError("the HIP MF path fuses torch.optim.Adam (default options)")
        grp = self.optimizer.param_groups[0]
        b1, b2 = grp.get("betas", (0.9, 0.999))
        m, v, grads = self._flat_state(ue.device)
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        lp = torch.empty(n_steps * _lib.RK_LOSS_PARTIALS, device=ue.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_mf_train_epoch(
            self.num_users, self.num_items, self.dim, _lib.ptr(ue.data), _lib.ptr(ie.data), _lib.ptr(ub.data),
            _lib.ptr(ib.data), float(self.mean.item()), _lib.ptr(m), _lib.ptr(v), _lib.ptr(grads), _lib.ptr(users),
            _lib.ptr(items), _lib.ptr(labels), n, batch, self._t, float(grp["lr"]), float(b1), float(b2),
            float(grp.get("eps", 1e-8)), _lib.ptr(lp), 1 if apply_update else 0, _lib.stream_ptr()), "rk_mf_train_epoch")
        if apply_update:
            self._t += n_steps
        return lp.view(n_steps, _lib.RK_LOSS_PARTIALS)

    def train_step(self, **config):
        """One epoch of pointwise BCE training (mf.py:49-69) -> (mean step loss,)."""
        self.train()
        pbar = config.get("progress_bar", None)
        (users, items, labels), batch = self._collect_epoch(self.dataset, ("users", "items", "labels"))
        dev = self.user_emb.weight.device
        users, items, labels = (t.to(dev).long().contiguous() for t in (users, items, labels))
        partials = self._run_epoch(users, items, labels, batch)
        step_losses = partials.sum(dim=1).double().cpu()
        mean_loss = float(step_losses.sum().item() / len(step_losses))
        if pbar:
            pbar.set_description(f"loss: {mean_loss:.4f}")
        return (mean_loss,)

    def scoring_tables(self):
        ue, ie, ub, ib = self._tables()
        return ue.data, ie.data, ub.data.reshape(-1), ib.data.reshape(-1), float(self.mean.item())

    def input_describe(self):
        return {
            "train_step": {
                "users": (torch.int64, (VarDim(comment="batch"))),
                "items": (torch.int64, (VarDim(comment="batch"))),
                "labels": (torch.int64, (VarDim(comment="batch"))),
            },
            "forward": {"users": (torch.int64, (VarDim(comment="batch"))), "items": (torch.int64, (VarDim(comment="batch")))},
        }

    def output_describe(self):
        return {
            "train_step": {"loss": (float, [])},
            "forward": {"unnormalized_scores": (torch.float32, [VarDim(comment="batch")])},
        }
