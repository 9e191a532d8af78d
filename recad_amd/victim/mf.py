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
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        self.drop_p = float(dropout)
        self._drop_seed = None   # drawn from torch's global RNG on first use, like nn.Dropout draws its masks from it
        self._drop_calls = 0
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
        self._fused_adam = self._adam_is_fused()
        self._flat = None   # {"m","v","g"}: flat [U*d | I*d | U | I] buffers; optimizer.state holds views of m / v

    def _tables(self):
        ts = (self.user_emb.weight, self.item_emb.weight, self.user_bias.weight, self.item_bias.weight)
        if ts[0].device.type != "cuda":
            raise _lib.HipCallError("MF parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        return ts

    def _flat_state(self, dev):
        """Flat moment / gradient buffers in the kernel's layout [user_emb | item_emb | user_bias | item_bias].
        The Adam moments live in optimizer.state[p] (exp_avg / exp_avg_sq / step, what torch.optim.Adam keeps) as
        VIEWS of the flat buffers, so optimizer.state_dict() / load_state_dict() and .to(device) carry them; the
        views are re-homed lazily whenever they are not where the kernel expects them."""
        ps = self._tables()
        sizes = [p.numel() for p in ps]
        offs = [0]
        for n_ in sizes:
            offs.append(offs[-1] + n_)
        fl = self._flat
        slots = [self._adam_slot(p) if self._fused_adam else None for p in ps]

        def in_place(buf, t, off):
            return t.device == buf.device and t.is_contiguous() and t.data_ptr() == buf.data_ptr() + 4 * off

        ok = fl is not None and fl["m"].device == dev and fl["m"].numel() == offs[-1]
        if ok and self._fused_adam:
            ok = all(in_place(fl["m"], st["exp_avg"], o) and in_place(fl["v"], st["exp_avg_sq"], o) for st, o in zip(slots, offs))
        if not ok:
            fl = {k: torch.zeros(offs[-1], device=dev, dtype=torch.float32) for k in ("m", "v", "g")}
            if self._fused_adam:
                for p, st, o, n_ in zip(ps, slots, offs, sizes):
                    fl["m"][o:o + n_].copy_(st["exp_avg"].reshape(-1).to(dev))
                    fl["v"][o:o + n_].copy_(st["exp_avg_sq"].reshape(-1).to(dev))
                    st["exp_avg"], st["exp_avg_sq"] = fl["m"][o:o + n_].view_as(p), fl["v"][o:o + n_].view_as(p)
            self._flat = fl
        self._offs = offs
        return fl["m"], fl["v"], fl["g"]

    def _steps_done(self):
        st = self.optimizer.state.get(self.user_emb.weight, {})
        return int(st["step"]) if "step" in st else 0

    def _dropout_now(self):
        """(p, seed) of the nn.Dropout on the logit (mf.py:27,47): active whenever the module is in training mode --
        also while the workflows score under no_grad without .eval() (normal.py:61-67).  A fresh mask per call."""
        if not (self.drop_p > 0.0 and self.training):
            return 0.0, 0
        if self._drop_seed is None:
            self._drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._drop_calls += 1
        return self.drop_p, (self._drop_seed + 0x9E3779B97F4A7C15 * self._drop_calls) % (1 << 63)

    def forward(self, users, items):
        ue, ie, ub, ib = self._tables()
        out = torch.empty(users.numel(), device=ue.device, dtype=torch.float32)
        p, seed = self._dropout_now()
        _lib.check(_lib.lib().rk_pair_scores(
            self.dim, _lib.ptr(ue.data), _lib.ptr(ie.data), _lib.ptr(ub.data), _lib.ptr(ib.data), float(self.mean.item()),
            _lib.ptr(users.long().contiguous()), _lib.ptr(items.long().contiguous()), users.numel(), _lib.ptr(out),
            p, seed, _lib.stream_ptr()), "rk_pair_scores")
        return out

    def score_matrix(self, user_ids, out):
        """out[len(user_ids), n_items] <- MF.forward over the full catalogue of each user, through the module's
        dropout when it is active (rk_score_matrix); the batched evaluation then runs rk_topk_rows on it."""
        ue, ie, ub, ib = self._tables()
        p, seed = self._dropout_now()
        _lib.check(_lib.lib().rk_score_matrix(
            self.dim, _lib.ptr(ue.data), user_ids.numel(), _lib.ptr(user_ids), _lib.ptr(ie.data), self.num_items,
            _lib.ptr(ub.data.reshape(-1)), _lib.ptr(ib.data.reshape(-1)), float(self.mean.item()), p, seed, _lib.ptr(out),
            _lib.stream_ptr()), "rk_score_matrix")

    def _run_epoch(self, users, items, labels, batch, apply_update=True):
        ue, ie, ub, ib = self._tables()
        if apply_update and not self._fused_adam:
            raise _lib.HipCallError("fused update requested with a non-default optimizer (internal error)")
        grp = self.optimizer.param_groups[0]
        b1, b2 = grp.get("betas", (0.9, 0.999))
        m, v, grads = self._flat_state(ue.device)
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        lp = torch.empty(n_steps * _lib.RK_LOSS_PARTIALS, device=ue.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_mf_train_epoch(
            self.num_users, self.num_items, self.dim, _lib.ptr(ue.data), _lib.ptr(ie.data), _lib.ptr(ub.data),
            _lib.ptr(ib.data), float(self.mean.item()), _lib.ptr(m), _lib.ptr(v), _lib.ptr(grads), _lib.ptr(users),
            _lib.ptr(items), _lib.ptr(labels), n, batch, self._steps_done(), float(grp.get("lr", 1e-3)), float(b1), float(b2),
            float(grp.get("eps", 1e-8)), _lib.ptr(lp), 1 if apply_update else 0, *self._dropout_now(), _lib.stream_ptr()),
            "rk_mf_train_epoch")
        if apply_update:
            for p in (ue, ie, ub, ib):
                self.optimizer.state[p]["step"] += n_steps
        return lp.view(n_steps, _lib.RK_LOSS_PARTIALS)

    def _grad_step(self, cols):
        """loss partials and the four dense gradients of ONE minibatch (no update): mf.py:59-63 without opt.step()."""
        u, i, y = cols
        part = self._run_epoch(u, i, y, max(u.numel(), 1), apply_update=False)
        g, o = self._flat["g"], self._offs
        ps = self._tables()
        return part.clone(), {p: g[o[k]:o[k + 1]].view_as(p).clone() for k, p in enumerate(ps)}

    def train_step(self, **config):
        """One epoch of pointwise BCE training (mf.py:49-69) -> (mean step loss,)."""
        self.train()
        pbar = config.get("progress_bar", None)
        (users, items, labels), batch = self._collect_epoch(self.dataset, ("users", "items", "labels"))
        dev = self.user_emb.weight.device
        users, items, labels = (t.to(dev).long().contiguous() for t in (users, items, labels))
        if self._fused_adam:
            step_losses = self._run_epoch(users, items, labels, batch).sum(dim=1).double().cpu()
        else:
            step_losses = self._unfused_epoch((users, items, labels), batch, self._grad_step)
        mean_loss = float(step_losses.sum().item() / len(step_losses))
        if pbar:
            pbar.set_description(f"loss: {mean_loss:.4f}")
        return (mean_loss,)

    scoring_tables_static = True   # views of the parameters, no launches: evaluate.EvalSession asks once and keeps them

    def scoring_tables(self):
        """Tables for the dot-product scoring path; None while a logit dropout is active (the evaluation then goes
        through score_matrix())."""
        if self.drop_p > 0.0 and self.training:
            return None
        ue, ie, ub, ib = self._tables()
        return ue.data, ie.data, ub.data.reshape(-1), ib.data.reshape(-1), float(self.mean.item())

    def scoring_tables_key(self):
        """What a cached scoring_tables() answer depends on, without a device read: storage and shape of the four tables, the
        storage and in-place version of `mean`, and whether a logit dropout is active.  evaluate.EvalSession compares it on every
        run(): re-homed parameters (.to(), load_state_dict into new tensors) or a dropout switched on must not replay a graph that
        holds the old pointers."""
        ue, ie, ub, ib = self._tables()
        return (tuple((t.data_ptr(), tuple(t.shape)) for t in (ue, ie, ub, ib)), self.mean.data_ptr(), self.mean._version,
                bool(self.drop_p > 0.0 and self.training))

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
