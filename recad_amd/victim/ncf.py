"""NCF victim on MI355X (model = 'NeuMF-end' | 'NeuMF-pre' | 'MLP' | 'GMF'): same interface as
recad/model/victim/ncf.py, hot path in HIP (recad_amd/csrc/ncf.hip: fp32-MFMA tower GEMMs, fused predict/BCE,
counter-based dropout masks, multi-tensor Adam)."""
import ctypes as C

import torch
from torch import nn

from .. import _lib
from ..utils import VarDim, pick_optim
from .base import BaseVictim


class NCF(BaseVictim):
    victim_name = "ncf"

    def _build(self, factor_num, num_layers, dropout, model, GMF_model, MLP_model, **config):
        self.dataset = config["dataset"]
        self.config = config
        info = self.dataset.info_describe()
        user_num, item_num = info["n_users"], info["n_items"]
        if model not in ("NeuMF-end", "NeuMF-pre", "MLP", "GMF"):
            raise ValueError(f"model must be one of 'MLP', 'GMF', 'NeuMF-end', 'NeuMF-pre' (ncf.py:19), got {model!r}")
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        if model == "NeuMF-pre" and (GMF_model is None or MLP_model is None):
            raise ValueError("model='NeuMF-pre' needs the pre-trained GMF_model and MLP_model (ncf.py:78-110)")
        if not 1 <= num_layers <= 8:
            raise ValueError("num_layers must be in [1, 8]")
        self.dropout, self.model, self.GMF_model, self.MLP_model = dropout, model, GMF_model, MLP_model
        self.factor_num, self.num_layers = factor_num, num_layers
        self.num_users, self.num_items = user_num, item_num
        E = factor_num * (2 ** (num_layers - 1))
        # construction and initialisation in the reference's order (ncf.py:33-58,60-77) so that a
        # seeded run starts from the same weights
        self.embed_user_GMF = nn.Embedding(user_num, factor_num)
        self.embed_item_GMF = nn.Embedding(item_num, factor_num)
        self.embed_user_MLP = nn.Embedding(user_num, E)
        self.embed_item_MLP = nn.Embedding(item_num, E)
        mods = []
        for i in range(num_layers):
            width = factor_num * (2 ** (num_layers - i))
            mods += [nn.Dropout(p=dropout), nn.Linear(width, width // 2), nn.ReLU()]
        self.MLP_layers = nn.Sequential(*mods)
        self.predict_layer = nn.Linear(factor_num if model in ("MLP", "GMF") else factor_num * 2, 1)   # ncf.py:49-53
        if model != "NeuMF-pre":
            for emb in (self.embed_user_GMF, self.embed_user_MLP, self.embed_item_GMF, self.embed_item_MLP):
                nn.init.normal_(emb.weight, std=0.01)
            for m in self.MLP_layers:
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
            nn.init.kaiming_uniform_(self.predict_layer.weight, a=1, nonlinearity="sigmoid")
            for m in self.modules():
                if isinstance(m, nn.Linear) and m.bias is not None:
                    m.bias.data.zero_()
        else:
            # ncf.py:78-110: embeddings and tower from the pre-trained GMF / MLP models, the predict layer the
            # concatenation of theirs, halved
            with torch.no_grad():
                self.embed_user_GMF.weight.copy_(GMF_model.embed_user_GMF.weight.to("cpu"))
                self.embed_item_GMF.weight.copy_(GMF_model.embed_item_GMF.weight.to("cpu"))
                self.embed_user_MLP.weight.copy_(MLP_model.embed_user_MLP.weight.to("cpu"))
                self.embed_item_MLP.weight.copy_(MLP_model.embed_item_MLP.weight.to("cpu"))
                for m1, m2 in zip(self.MLP_layers, MLP_model.MLP_layers):
                    if isinstance(m1, nn.Linear) and isinstance(m2, nn.Linear):
                        m1.weight.copy_(m2.weight.to("cpu"))
                        m1.bias.copy_(m2.bias.to("cpu"))
                pw = torch.cat([GMF_model.predict_layer.weight.to("cpu"), MLP_model.predict_layer.weight.to("cpu")], dim=1)
                self.predict_layer.weight.copy_(0.5 * pw)
                self.predict_layer.bias.copy_(0.5 * (GMF_model.predict_layer.bias.to("cpu") + MLP_model.predict_layer.bias.to("cpu")))
        self.optimizer = pick_optim(config["optim"])(self.parameters(), lr=config["lr"])
        self.loss_func = nn.BCEWithLogitsLoss()
        self._E = E
        self._ws = None
        self._fused_adam = self._adam_is_fused()
        self._mode = {"NeuMF-end": 0, "NeuMF-pre": 0, "MLP": 1, "GMF": 2}[model]
        self.drop_p = float(dropout)
        self._drop_seed = None   # drawn from torch's global RNG on first use (nn.Dropout draws its masks from it)
        self._drop_calls = 0
        self._mask_steps = 0   # minibatches trained so far (numbers the dropout masks of gradient-only calls)
        self.max_batch = 16384  # forward chunk of the full-catalog evaluation: large enough for the 128x128-tile GEMM

    # ------------------------------------------------------------------ C-ABI descriptor
    def _tensors(self):
        lin = [m for m in self.MLP_layers if isinstance(m, nn.Linear)]
        return ([self.embed_user_GMF.weight, self.embed_item_GMF.weight, self.embed_user_MLP.weight, self.embed_item_MLP.weight]
                + [m.weight for m in lin] + [m.bias for m in lin] + [self.predict_layer.weight, self.predict_layer.bias])

    def _desc(self, batch):
        ts = self._tensors()
        dev = ts[0].device
        if dev.type != "cuda":
            raise _lib.HipCallError("NCF parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        L, f = self.num_layers, self.factor_num
        mb = max(int(batch), self.max_batch)
        key = (tuple(t.data_ptr() for t in ts), mb)
        if self._ws is None or self._ws["key"] != key:
            widths = sum(f * 2 ** (L - l) for l in range(L + 1))
            ws = {"key": key,
                  "grad": [torch.zeros_like(t) for t in ts],
                  "acts": torch.empty(mb * widths, device=dev), "dacts": torch.empty(mb * widths, device=dev),
                  "d0": torch.empty(mb, device=dev), "max_batch": mb,
                  # split-K slices of the dX GEMMs and the 8 k-blocks of the blocked training forward ([8][batch, widest output])
                  "gemm_scratch": torch.empty(max(8 << 20, 8 * mb * (f * 2 ** (L - 1))), device=dev),
                  "wgrad_part": torch.empty(((mb + 63) // 64) * (2 * f + 1), device=dev)}
            self._ws = ws
        ws = self._ws
        grp = self.optimizer.param_groups[0]
        b1, b2 = grp.get("betas", (0.9, 0.999))
        # Adam's moments live where torch.optim.Adam keeps them (optimizer.state[p]: exp_avg / exp_avg_sq / step), so
        # optimizer.state_dict() / load_state_dict() carry them; they follow the parameters across .to(device)
        if self._fused_adam:
            slots = [self._adam_slot(t) for t in ts]
            for st, t in zip(slots, ts):
                for k_ in ("exp_avg", "exp_avg_sq"):
                    if st[k_].device != dev or not st[k_].is_contiguous():
                        st[k_] = st[k_].to(dev).contiguous()
            mom_m, mom_v = [st["exp_avg"] for st in slots], [st["exp_avg_sq"] for st in slots]
        else:  # foreign optimizer: the library only computes gradients; unused moment buffers keep the descriptor whole
            if ws.get("dummy") is None:
                ws["dummy"] = ([torch.zeros_like(t) for t in ts], [torch.zeros_like(t) for t in ts])
            mom_m, mom_v = ws["dummy"]
        d = _lib.NCFDesc(n_users=self.num_users, n_items=self.num_items, factor=f, n_layers=L, lr=float(grp.get("lr", 1e-3)),
                         beta1=float(b1), beta2=float(b2), eps=float(grp.get("eps", 1e-8)),
                         ug=_lib.ptr(ts[0].data), ig=_lib.ptr(ts[1].data), um=_lib.ptr(ts[2].data), im=_lib.ptr(ts[3].data),
                         pw=_lib.ptr(ts[-2].data), pb=_lib.ptr(ts[-1].data), acts=_lib.ptr(ws["acts"]), dacts=_lib.ptr(ws["dacts"]),
                         d0=_lib.ptr(ws["d0"]), max_batch=ws["max_batch"], gemm_scratch=_lib.ptr(ws["gemm_scratch"]),
                         gemm_scratch_floats=ws["gemm_scratch"].numel(), wgrad_part=_lib.ptr(ws["wgrad_part"]), mode=self._mode)
        if self.drop_p > 0.0 and self.training:   # also while the workflows score without .eval() (normal.py:61-67)
            if self._drop_seed is None:
                self._drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            self._drop_calls += 1
            d.dropout, d.drop_seed, d.drop_call = self.drop_p, self._drop_seed, self._drop_calls & 0x7FFFFFFF
        for l in range(L):
            d.W[l] = ts[4 + l].data.data_ptr()
            d.b[l] = ts[4 + L + l].data.data_ptr()
        for k in range(len(ts)):
            d.grad[k] = ws["grad"][k].data_ptr()
            d.m[k] = mom_m[k].data_ptr()
            d.v[k] = mom_v[k].data_ptr()
        return d

    # ------------------------------------------------------------------ reference API
    def forward(self, user, item):
        d = self._desc(0)
        out = torch.empty(user.numel(), device=self.predict_layer.weight.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_ncf_forward(C.byref(d), _lib.ptr(user.long().contiguous()), _lib.ptr(item.long().contiguous()),
                                             None, 0, user.numel(), _lib.ptr(out), _lib.stream_ptr()), "rk_ncf_forward")
        return out.view(-1)

    def _run_epoch(self, users, items, labels, batch, apply_update=True):
        d = self._desc(batch)
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        lp = torch.empty(n_steps * _lib.RK_LOSS_PARTIALS, device=users.device, dtype=torch.float32)
        if apply_update and not self._fused_adam:
            raise _lib.HipCallError("fused update requested with a non-default optimizer (internal error)")
        st0 = self.optimizer.state.get(self._tensors()[0], {})
        t0 = int(st0["step"]) if "step" in st0 else 0
        if not apply_update:
            # gradient-only calls (a foreign optimizer applies the update): the library numbers the train-time dropout masks
            # by adam_t0 + step, and optimizers such as SGD keep no 'step' -- count the minibatches here so that every one
            # draws a fresh mask like nn.Dropout does (ncf.py:44)
            t0 = self._mask_steps
        self._mask_steps += n_steps
        _lib.check(_lib.lib().rk_ncf_train_epoch(C.byref(d), _lib.ptr(users), _lib.ptr(items), _lib.ptr(labels), n, batch,
                                                 t0, _lib.ptr(lp), 1 if apply_update else 0, _lib.stream_ptr()),
                   "rk_ncf_train_epoch")
        if apply_update:
            for t in self._tensors():
                self.optimizer.state[t]["step"] += n_steps
        return lp.view(n_steps, _lib.RK_LOSS_PARTIALS)

    def _grad_step(self, cols):
        """loss partials and every tensor's dense gradient for ONE minibatch (no update): ncf.py:143-147 without
        optimizer.step().  The gradient buffers accumulate (scatter-adds, split-K atomics): cleared after the copy."""
        u, i, y = cols
        part = self._run_epoch(u, i, y, max(u.numel(), 1), apply_update=False)
        grads = {t: g.clone() for t, g in zip(self._tensors(), self._ws["grad"])}
        for g in self._ws["grad"]:
            g.zero_()
        return part.clone(), grads

    def train_step(self, **config):
        """One epoch of pointwise BCE training (ncf.py:133-153) -> (mean step loss,)."""
        self.train()
        pbar = config.get("progress_bar", None)
        (users, items, labels), batch = self._collect_epoch(self.dataset, ("users", "items", "labels"))
        dev = self.predict_layer.weight.device
        users, items, labels = (t.to(dev).long().contiguous() for t in (users, items, labels))
        if self._fused_adam:
            step_losses = self._run_epoch(users, items, labels, batch).sum(dim=1).double().cpu()
        else:
            step_losses = self._unfused_epoch((users, items, labels), batch, self._grad_step)
        mean_loss = float(step_losses.sum().item() / len(step_losses))
        if pbar:
            pbar.set_description(f"loss: {mean_loss:.5f}")
        return (mean_loss,)

    def score_matrix(self, user_ids, out):
        """out[len(user_ids), n_items] <- forward over the full catalog of each user (evaluation)."""
        d = self._desc(0)
        _lib.check(_lib.lib().rk_ncf_forward(C.byref(d), None, None, _lib.ptr(user_ids), self.num_items,
                                             user_ids.numel() * self.num_items, _lib.ptr(out), _lib.stream_ptr()),
                   "rk_ncf_forward")

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
