"""LightGCN victim on MI355X: same interface as recad/model/victim/lightgcn.py, hot path in HIP.

forward/backward/Adam run in librecad_hip.so (recad_amd/csrc/lightgcn.hip); this module is
orchestration only: parameter ownership (nn.Embedding, so .to()/state_dict() behave as in
the reference), the optimizer object, and the C-ABI handle.
"""
import time
import ctypes as C

import torch
from torch import nn

from .. import _lib
from ..graph import CsrGraph
from ..utils import VarDim, get_logger, pick_optim
from .base import BaseVictim


class LightGCN(BaseVictim):
    victim_name = "lightgcn"

    @classmethod
    def extra_user_args(cls, kwargs):
        # recad/model/victim/lightgcn.py:21-28 (pretrained tables are supplied by the caller)
        # + this build's own option: deterministic (ordered gradient scatter, rk_lightgcn_set_deterministic)
        return "user_emb, item_emb, deterministic" if kwargs.get("pretrain", False) else "deterministic"

    def _build(self, **config):
        self.config = config
        self.dataset = config["dataset"]
        self.logger = get_logger(__name__, config["logging_level"])
        info = self.dataset.info_describe()
        self.num_users, self.num_items = info["n_users"], info["n_items"]
        self.Graph = info.get("graph_csr", None) or info["graph"]
        self.latent_dim = config["latent_dim_rec"]
        self.n_layers = config["lightGCN_n_layers"]
        self.keep_prob = config["keep_prob"]
        self.A_split = config["A_split"]
        if self.A_split:
            raise ValueError("A_split is not support in LightGCN yet")  # lightgcn.py:38-39
        # graph dropout (lightgcn.py:62-80,91-95): `dropout` truthy switches it on, `keep_prob` is the rate
        self.graph_dropout = bool(config["dropout"])
        if self.graph_dropout and not (0.0 < float(self.keep_prob) <= 1.0):
            raise ValueError(f"keep_prob must be in (0, 1], got {self.keep_prob}")
        self._drop_seed = None     # drawn from torch's global RNG on first use (the reference draws torch.rand per step)
        self._drop_calls = 0       # one fresh mask per computer() call outside train_step
        # same RNG consumption order as lightgcn.py:40-48
        self.embedding_user = nn.Embedding(self.num_users, self.latent_dim)
        self.embedding_item = nn.Embedding(self.num_items, self.latent_dim)
        if config["pretrain"] == 0:
            nn.init.normal_(self.embedding_user.weight, std=0.1)
            nn.init.normal_(self.embedding_item.weight, std=0.1)
        else:
            self.embedding_user.weight.data.copy_(torch.from_numpy(config["user_emb"]))
            self.embedding_item.weight.data.copy_(torch.from_numpy(config["item_emb"]))
        self.f = nn.Sigmoid()
        self.optimizer = pick_optim(config["optim"])(self.parameters(), lr=config["lr"])
        self._fused_adam = self._adam_is_fused()
        self._handle = None
        self._handle_key = None
        self._fingerprint = None
        self._ws = None
        self.graph_steps = 32  # steps per hipGraph replay of a long epoch (<= 64-step epochs are one replay); 0/1 = plain launches
        # last forward layer only on the minibatch's rows (-4 us of 18 on ml1m)
        self.use_batch_sparsity = True
        # LDS-resident sliced SpMM (csrc/spmm_lds.h) on graphs that qualify: True / False force it on / off, "auto" takes it
        # when it is the faster kernel -- staging the slice tables costs ~2 us per launch whatever the graph holds, so sparse
        # graphs (the reference's as-is test-edge graphs: 10 nonzeros per row, 7.3 vs 6.6 us per launch) stay on the row gather
        self.use_lds = "auto"
        # LDS path, opt-in: the L layers of a forward / backward pass as ONE launch whose workgroups hand the layers over to each
        # other per column group (csrc/spmm_lds.h, spmm_lds_multi_kernel) instead of L launches.  Same bits; measured SLOWER on
        # MI355X (ml1m: 90.2 vs 72.7 us per step -- an in-launch hand-off costs 2.5-3 us per seam against 1.9 us for the kernel
        # boundary it removes), so the default is one launch per layer
        self.fuse_layers = False
        # ordered (bit-reproducible) gradient scatter instead of float atomics: one more launch per step and a sort of
        # the epoch's triplets (rk_lightgcn_set_deterministic); not a reference option (its CUDA path is atomic too)
        self.deterministic = bool(config.get("deterministic", False))

    # ------------------------------------------------------------------ C-ABI handle
    def _adam_state(self, p):
        if self._fused_adam:
            return self._adam_slot(p)
        # foreign optimizer: its own state stays untouched; the handle still wants moment buffers (unused: the
        # library is only asked for gradients then)
        if getattr(self, "_dummy_mom", None) is None:
            self._dummy_mom = {}
        st = self._dummy_mom.setdefault(id(p), {"step": torch.zeros((), dtype=torch.float32)})
        if "exp_avg" not in st or st["exp_avg"].shape != p.shape:
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        return st

    def _fuse_tables(self):
        """The kernels address E0 = [users; items] (and the Adam moments) through ONE base
        pointer, i.e. torch.cat (lightgcn.py:88) costs nothing.  Re-home the two Parameters'
        storage (and their moments) into one [N,d] allocation whenever they are not adjacent
        (fresh model, after .to(), after load_state_dict into new tensors).  Values and
        Parameter identities are preserved, so optimizers/state_dicts keep working."""
        wu, wi = self.embedding_user.weight, self.embedding_item.weight
        su, si = self._adam_state(wu), self._adam_state(wi)
        U, d = self.num_users, self.latent_dim

        def adjacent(a, b):
            return (a.device == b.device and a.is_contiguous() and b.is_contiguous()
                    and b.data_ptr() == a.data_ptr() + U * d * 4)

        def fuse(a, b):
            flat = torch.empty(self.num_users + self.num_items, d, device=wu.device, dtype=torch.float32)
            flat[:U].copy_(a)
            flat[U:].copy_(b)
            return flat[:U], flat[U:]

        if not adjacent(wu.data, wi.data):
            wu.data, wi.data = fuse(wu.data, wi.data)
        for k in ("exp_avg", "exp_avg_sq"):
            if not adjacent(su[k], si[k]) or su[k].device != wu.device:
                su[k], si[k] = fuse(su[k].to(wu.device), si[k].to(wu.device))
        return wu, wi, su, si

    def _csr(self, device):
        if not isinstance(self.Graph, CsrGraph):
            self.Graph = CsrGraph.from_torch_coo(self.Graph, device, class_split=self.num_users)
        return self.Graph.to(device)

    def _handle_fingerprint(self, want_grad):
        """What a live handle depends on, read the cheap way (no storage checks): the four base pointers the kernels were given
        and every value the handle copied.  Equal to the one taken when the handle was built <=> the slow path's key is equal
        (a Parameter or moment that was re-homed, moved or replaced has a new data_ptr)."""
        wu, wi = self.embedding_user.weight, self.embedding_item.weight
        st = self.optimizer.state if self._fused_adam else None
        su = st.get(wu) if st is not None else None
        si = st.get(wi) if st is not None else None
        if st is not None and (not su or not si or "exp_avg" not in su or "exp_avg" not in si):
            return None
        grp = self.optimizer.param_groups[0]
        betas = grp.get("betas", (0.9, 0.999))
        return (wu.data_ptr(), wi.data_ptr(), su["exp_avg"].data_ptr() if su else 0, si["exp_avg"].data_ptr() if si else 0,
                su["exp_avg_sq"].data_ptr() if su else 0, si["exp_avg_sq"].data_ptr() if si else 0, bool(want_grad),
                float(grp["lr"]), float(betas[0]), float(betas[1]), float(grp.get("eps", 1e-8)), float(self.config["lambda"]),
                self.graph_dropout, float(self.keep_prob) if self.graph_dropout else 0.0,
                bool(self.deterministic), str(self.use_lds), bool(self.fuse_layers))

    def _ensure_handle(self, want_grad=False):
        # fast path of a live handle (every epoch call, every evaluation): one tuple of pointers and scalars against the one taken
        # when the handle was built -- the full check below re-walks the storages (adjacency, devices) and costs 3x as much,
        # with the device waiting behind the caller's synchronize in a short epoch
        if self._handle is not None and self._fingerprint is not None and self._fingerprint == self._handle_fingerprint(want_grad):
            return self._handle
        _lib.require_gpu()
        dev = self.embedding_user.weight.device
        if dev.type != "cuda":
            raise _lib.HipCallError("LightGCN parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        wu, wi, su, si = self._fuse_tables()
        grp = self.optimizer.param_groups[0]
        betas = grp.get("betas", (0.9, 0.999))
        # everything the handle (and the hipGraph captured inside it) copies BY VALUE is part of the key: a later
        # change of lr / betas / eps (a scheduler, a manual decay) or of config["lambda"] rebuilds the handle, as
        # the reference re-reads them every step
        key = (wu.data_ptr(), wi.data_ptr(), su["exp_avg"].data_ptr(), si["exp_avg"].data_ptr(),
               su["exp_avg_sq"].data_ptr(), si["exp_avg_sq"].data_ptr(), bool(want_grad),   # (every base pointer the descriptor holds)
               self.graph_dropout, float(self.keep_prob) if self.graph_dropout else 0.0,
               float(grp["lr"]), float(betas[0]), float(betas[1]), float(grp.get("eps", 1e-8)), float(self.config["lambda"]),
               bool(self.deterministic), str(self.use_lds), bool(self.fuse_layers))
        if self._handle is not None and self._handle_key == key:
            self._fingerprint = self._handle_fingerprint(want_grad)
            return self._handle
        self._drop_handle()
        g = self._csr(dev)
        N, d = self.num_users + self.num_items, self.latent_dim
        ws = {k: torch.zeros(N, d, device=dev, dtype=torch.float32) for k in ("buf_a", "buf_b", "light", "gprop", "gego")}
        ws["grad"] = torch.zeros(N, d, device=dev, dtype=torch.float32) if want_grad else None
        ws["state"] = torch.zeros(16, device=dev, dtype=torch.int32)
        ws["coef"] = torch.zeros(3 * _lib.RK_MAX_GRAPH_STEPS, device=dev, dtype=torch.float32)
        ws["row_bits"] = torch.zeros((N + 31) // 32, device=dev, dtype=torch.int32) if self.use_batch_sparsity else None
        if self.graph_dropout and self._drop_seed is None:
            self._drop_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ws["tpos"] = g.transpose_index() if self.graph_dropout else None
        # LDS-resident sliced propagation (csrc/spmm_lds.h) when the graph qualifies: bipartite normalised binary adjacency
        # whose class tables fit a CU's LDS (ml1m / Amazon-game size); use_lds = False keeps the row-gather kernel (bench.py --spmm csr)
        lds = None
        want_lds = self.use_lds if self.use_lds != "auto" else g.nnz >= 24 * N
        if want_lds and self.n_layers >= 1 and not self.graph_dropout:
            lds = g.lds_plan(d)
        ws["lds"] = lds
        # the row-gather kernel's schedule (host-built) only when this handle will launch it
        wave_desc, n_blocks = (None, 0) if lds else g.schedule(d)
        ws["spmm_scratch"] = None if lds else g.new_scratch(d)  # this handle's own long-row counters / partial slots
        # the row-filtered last forward layer starts only the schedule's workgroups that hold a minibatch row (listed here by the
        # launch before it): int32[4 + blocks]; the long-row pieces bound the list beyond 3 * batch (scratch words / dim >= pieces)
        n_sched_blocks = (int(n_blocks) & 0x07ffffff) if not lds else 0   # (bits 27-30 of the opaque launch parameter are flags)
        use_list = (not lds) and ws["row_bits"] is not None and self.n_layers >= 3 and getattr(self, "use_block_list", True)
        ws["row_blocks"] = torch.zeros(4 + n_sched_blocks, device=dev, dtype=torch.int32) if use_list else None
        row_blocks_extra = (ws["spmm_scratch"].numel() // max(d, 1) + 1) if (use_list and ws["spmm_scratch"] is not None) else 0
        ws["lsum"] = torch.zeros(N, d, device=dev, dtype=torch.float32) if lds else None
        ws["cnt"] = torch.zeros(N, device=dev, dtype=torch.int32) if lds else None   # per-node incidence counts of a minibatch
        fuse = bool(lds) and bool(self.fuse_layers) and self.n_layers >= 2   # (bench.py --fuse-layers for A/B runs)
        ws["lds_sync"] = torch.zeros(_lib.RK_LDS_SYNC_WORDS, device=dev, dtype=torch.int32) if fuse else None   # this handle's own hand-off counters
        for k in ("e0s", "ms", "vs"):   # sliced working copies of E0 and the Adam moments
            ws[k] = torch.zeros(N, d, device=dev, dtype=torch.float32) if lds else None
        desc = _lib.LightGCNDesc(
            n_users=self.num_users, n_items=self.num_items, dim=d, n_layers=self.n_layers,
            lam=float(self.config["lambda"]), lr=float(grp["lr"]), beta1=float(betas[0]), beta2=float(betas[1]),
            eps=float(grp.get("eps", 1e-8)),
            rowptr=_lib.ptr(g.rowptr), col=_lib.ptr(g.col), val=_lib.ptr(g.val), wave_desc=_lib.ptr(wave_desc),
            n_blocks=n_blocks,
            user_emb=_lib.ptr(wu.data), item_emb=_lib.ptr(wi.data),
            m_user=_lib.ptr(su["exp_avg"]), v_user=_lib.ptr(su["exp_avg_sq"]),
            m_item=_lib.ptr(si["exp_avg"]), v_item=_lib.ptr(si["exp_avg_sq"]),
            buf_a=_lib.ptr(ws["buf_a"]), buf_b=_lib.ptr(ws["buf_b"]), light=_lib.ptr(ws["light"]),
            gprop=_lib.ptr(ws["gprop"]), gego=_lib.ptr(ws["gego"]), grad=_lib.ptr(ws["grad"]),
            state=_lib.ptr(ws["state"]), coef=_lib.ptr(ws["coef"]),
            spmm_scratch=_lib.ptr(ws["spmm_scratch"]),
            row_bits=_lib.ptr(ws["row_bits"]), row_blocks=_lib.ptr(ws["row_blocks"]), row_blocks_extra=int(row_blocks_extra),
            keep_prob=float(self.keep_prob) if self.graph_dropout else 0.0,
            drop_seed=self._drop_seed if self.graph_dropout else 0, tpos=_lib.ptr(ws["tpos"]),
            lds_plan=_lib.ptr(lds[0]) if lds else None, lds_info=lds[1] if lds else _lib.LdsInfo(),
            lsum=_lib.ptr(ws["lsum"]), e0s=_lib.ptr(ws["e0s"]), ms=_lib.ptr(ws["ms"]), vs=_lib.ptr(ws["vs"]), cnt=_lib.ptr(ws["cnt"]),
            lds_sync=_lib.ptr(ws["lds_sync"]))
        h = C.c_void_p()
        _lib.check(_lib.lib().rk_lightgcn_create(C.byref(desc), C.byref(h)), "rk_lightgcn_create")
        if self.deterministic:
            _lib.check(_lib.lib().rk_lightgcn_set_deterministic(h, 1), "rk_lightgcn_set_deterministic")
        self._handle, self._handle_key, self._ws = h, key, ws
        self._fingerprint = self._handle_fingerprint(want_grad)
        return h

    def check_handoffs(self):
        """Raise if an in-launch hand-off of the fused multi-layer propagation ever gave up waiting (rk_lightgcn_sync_status);
        synchronises the stream, so it is called where the host waits anyway (after an epoch's loss read-back)."""
        if self._handle is not None and self._ws is not None and self._ws.get("lds_sync") is not None:
            st = C.c_int32(0)
            _lib.check(_lib.lib().rk_lightgcn_sync_status(self._handle, C.byref(st), _lib.stream_ptr()), "rk_lightgcn_sync_status")
            if st.value:
                raise _lib.HipCallError("LightGCN: a hand-off wait of the multi-layer propagation launch timed out (results invalid)")

    def _drop_handle(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().rk_lightgcn_destroy(self._handle)
            self._handle = None
            self._handle_key = None
            self._fingerprint = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    def computer(self):
        """lightgcn.py:82-113 -> (users[U,d], items[I,d]).  Fresh tensors like the reference's: the handle's
        `light` workspace is overwritten by the next propagation / train step, so the public call copies it
        (2 x [N,d] floats); the module's own consumers use the views of _propagate()."""
        lu, li = self._propagate()
        return lu.clone(), li.clone()

    def _propagate(self):
        """computer() as VIEWS of the handle's `light` workspace: valid until the next propagation or train step
        on this module (same stream)."""
        h = self._ensure_handle(want_grad=self._ws is not None and self._ws.get("grad") is not None)
        if self.graph_dropout and self.training:
            # lightgcn.py:91-95: a module in training mode propagates through a freshly dropped-out graph -- also
            # when the workflows score under no_grad without calling .eval() (normal.py:61-67)
            self._drop_calls += 1
            _lib.check(_lib.lib().rk_lightgcn_propagate_dropout(h, (1 << 40) + self._drop_calls, _lib.stream_ptr()),
                       "rk_lightgcn_propagate_dropout")
        else:
            _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "rk_lightgcn_propagate")
        if self.fuse_layers:
            self.check_handoffs()   # (opt-in multi-phase launch only; synchronises) a timed-out hand-off means `light` is invalid
        light = self._ws["light"]
        return light[: self.num_users], light[self.num_users:]

    def getUsersRating(self, users):
        """lightgcn.py:115-120: sigmoid(users_emb . items_emb^T) -> [len(users), n_items]; the dot products on the
        build's own fp32-MFMA GEMM (rk_users_rating), the user rows gathered inside its tile loads."""
        all_users, all_items = self._propagate()
        ids = users.to(device=all_users.device, dtype=torch.int32).contiguous()
        out = torch.empty(ids.numel(), self.num_items, device=all_users.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_users_rating(self.latent_dim, _lib.ptr(all_users), ids.numel(), _lib.ptr(ids), _lib.ptr(all_items),
                                              self.num_items, _lib.ptr(out), _lib.stream_ptr()), "rk_users_rating")
        return out

    def getEmbedding(self, users, pos_items, neg_items):
        all_users, all_items = self._propagate()   # the indexing below copies the rows out
        return (all_users[users], all_items[pos_items], all_items[neg_items], self.embedding_user(users),
                self.embedding_item(pos_items), self.embedding_item(neg_items))

    def _loss_buffer(self, n, batch, device):
        """Per-step loss partials of an epoch.  (The triplet tensors are handed to the library as they are: their pointers
        reach the kernels through the device state block, so the captured hipGraphs do not depend on them.)"""
        n_steps = (n + batch - 1) // batch
        buf = self._ws.get("loss")
        if buf is None or buf.numel() < n_steps * _lib.RK_LOSS_PARTIALS or buf.device != device:
            buf = torch.empty(max(n_steps, int(1.25 * n_steps) if buf is not None else n_steps) * _lib.RK_LOSS_PARTIALS,
                              device=device, dtype=torch.float32)
            self._ws["loss"] = buf
        return buf

    def reserve(self, n_triplets, batch, want_grad=False):
        """Capture + upload the hipGraphs an epoch of n_triplets will replay now, so that no later epoch (or timed region)
        pays for it."""
        h = self._ensure_handle(want_grad=want_grad)
        self._loss_buffer(int(n_triplets), int(batch), self.embedding_user.weight.device)
        if self._fused_adam and int(self.graph_steps) > 1 and not self.deterministic:   # (deterministic: the graph follows the epoch's plan)
            _lib.check(_lib.lib().rk_lightgcn_prepare(h, int(n_triplets), int(batch), 1, int(self.graph_steps), _lib.stream_ptr()),
                       "rk_lightgcn_prepare")

    def _run_epoch(self, users, pos, neg, batch, apply_update=True, want_grad=False):
        ta = time.perf_counter()
        h = self._ensure_handle(want_grad=want_grad)
        tb = time.perf_counter()
        n = users.numel()
        n_steps = (n + batch - 1) // batch
        users, pos, neg = (t.contiguous() for t in (users, pos, neg))
        loss_partials = self._loss_buffer(n, batch, users.device)
        su = self._adam_state(self.embedding_user.weight)
        t0 = int(su["step"].item()) if apply_update else 0
        args = (h, _lib.ptr(users), _lib.ptr(pos), _lib.ptr(neg), n, batch, t0, _lib.ptr(loss_partials),
                1 if apply_update else 0, int(self.graph_steps) if apply_update else 0, _lib.stream_ptr())
        fn = _lib.lib().rk_lightgcn_train_epoch
        tc = time.perf_counter()
        rc = fn(*args)
        # host-side stamps of this call: entered, handle checked, C call entered / left (bench.py itemises its timed call with them)
        self.last_call_seconds = (ta, tb, tc, time.perf_counter())
        _lib.check(rc, "rk_lightgcn_train_epoch")
        self._ws["epoch_inputs"] = (users, pos, neg)   # alive until the next call: the stream may still be reading them
        if apply_update:
            for p in (self.embedding_user.weight, self.embedding_item.weight):
                self.optimizer.state[p]["step"] += n_steps
        return loss_partials[: n_steps * _lib.RK_LOSS_PARTIALS].view(n_steps, _lib.RK_LOSS_PARTIALS)

    def _grad_step(self, cols):
        """loss partials and dLoss/dE0 of ONE minibatch (no update): the gradient half of lightgcn.py:149-167."""
        u, p, n = cols
        part = self._run_epoch(u, p, n, max(u.numel(), 1), apply_update=False, want_grad=True)
        if self.fuse_layers:
            self.check_handoffs()
        g = self._ws["grad"]
        U = self.num_users
        return part.clone(), {self.embedding_user.weight: g[:U].clone(), self.embedding_item.weight: g[U:].clone()}

    def train_step(self, **config):
        """One epoch over dataset.generate_batch() (lightgcn.py:132-172) -> (mean step loss,)."""
        self.train()
        pbar = config.get("progress_bar", None)
        (users, pos, neg), batch = self._collect_epoch(self.dataset, ("users", "positive_items", "negative_items"))
        dev = self.embedding_user.weight.device
        users, pos, neg = (t.to(dev).long().contiguous() for t in (users, pos, neg))
        if self._fused_adam:
            partials = self._run_epoch(users, pos, neg, batch)
            step_losses = partials.sum(dim=1).double().cpu()  # ONE device->host sync per epoch
            self.check_handoffs()
        else:
            step_losses = self._unfused_epoch((users, pos, neg), batch, self._grad_step)
        mean_loss = float(step_losses.sum().item() / len(step_losses))
        if pbar:
            pbar.set_description(f"loss {mean_loss:.5f}")
        return (mean_loss,)

    def forward(self, users, items):
        all_users, all_items = self._propagate()
        out = torch.empty(users.numel(), device=all_users.device, dtype=torch.float32)
        _lib.check(_lib.lib().rk_pair_scores(
            self.latent_dim, _lib.ptr(all_users), _lib.ptr(all_items), None, None, 0.0, _lib.ptr(users.long().contiguous()),
            _lib.ptr(items.long().contiguous()), users.numel(), _lib.ptr(out), 0.0, 0, _lib.stream_ptr()), "rk_pair_scores")
        return out

    # ------------------------------------------------------------------ batched evaluation hook
    def scoring_tables(self):
        """(user_rows[U,d], item_rows[I,d], user_bias|None, item_bias|None, mean) for rk_score_topk: views of the
        handle's workspace, consumed by the caller before anything else runs on this module."""
        all_users, all_items = self._propagate()
        return all_users, all_items, None, None, 0.0

    def input_describe(self):
        return {
            "train_step": {
                "users": (torch.int64, (VarDim(comment="batch"))),
                "positive_items": (torch.int64, (VarDim(comment="batch"))),
                "negative_items": (torch.int64, (VarDim(comment="batch"))),
            },
            "forward": {"users": (torch.int64, (VarDim(comment="batch"))), "items": (torch.int64, (VarDim(comment="batch")))},
        }

    def output_describe(self):
        return {
            "train_step": {"loss": (float, [])},
            "forward": {"unnormalized_scores": (torch.float32, [VarDim(comment="batch")])},
        }
