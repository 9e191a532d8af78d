"""Row-sharded LightGCN training over the GPUs of one node (north_star).

One process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  The N = U+I node rows are dealt round-robin to the W ranks (row r -> rank r % W, local
row r // W), which balances nonzeros for power-law graphs without a reordering pass.  Rank g
owns rows R_g of E0 and of Adam's moments and the CSR slab A[R_g, :] (columns re-labelled to
the gathered layout).  Per train step:

    forward   l = 1..L : all-gather X_{l-1}  ->  local SpMM on the slab (fused layer sum)
              all-gather light
    BPR       replicated on every rank (B is tiny next to the graph): gprop/gego are full
              and identical everywhere, so the backward needs NO reduce-scatter
    backward  j = 1..L : local SpMM (A symmetric: the same row slab serves the transpose),
              all-gather t_j between layers; the last one fuses Adam on the owned rows

=> 2L all-gathers of (N/W)*d*4 bytes per rank and step, each rank sending its shard to the
other W-1 ranks over its own xGMI links.  Results are identical for every W (fixed summation
order inside a row; tests/test_sharded_gloo.py checks W=2 against W=1 and the oracle).

The compute calls go through an `ops` object: `HipOps` (the C-ABI) in the product; the CPU
tests inject an oracle-backed stand-in -- this module never imports it.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class HipOps:
    """The product implementation: librecad_hip.so through include/recad_hip.h."""

    name = "hip"

    def make_slab(self, rowptr, col, val, device):
        return {"n_rows": len(rowptr) - 1, "rowptr": torch.as_tensor(rowptr, dtype=torch.int32, device=device).contiguous(),
                "col": torch.as_tensor(col, dtype=torch.int32, device=device).contiguous(),
                "val": torch.as_tensor(val, dtype=torch.float32, device=device).contiguous(), "sched": {},
                "coef": torch.zeros(2, device=device)}

    @staticmethod
    def _sched(slab, dim):
        if dim not in slab["sched"]:
            sched, n_blocks, n_words = C.c_void_p(), C.c_int32(0), C.c_int64(0)
            _lib.check(_lib.lib().rk_csr_schedule_build(slab["n_rows"], _lib.ptr(slab["rowptr"]), 0, dim, _lib.stream_ptr(), C.byref(sched),
                                                        C.byref(n_blocks), C.byref(n_words)), "rk_csr_schedule_build")
            try:
                desc = torch.zeros(int(n_words.value), device=slab["rowptr"].device, dtype=torch.int32)
                _lib.check(_lib.lib().rk_csr_schedule_upload(sched, _lib.ptr(desc), _lib.stream_ptr()), "rk_csr_schedule_upload")
            finally:
                _lib.lib().rk_csr_schedule_destroy(sched)
            slab["sched"][dim] = (desc, int(n_blocks.value))
        return slab["sched"][dim]

    def spmm(self, slab, x, add=None, y=None, sum_in=None, sum_out=None, sum_scale=1.0, adam=None):
        e = _lib.SpmmEpilogue(add=_lib.ptr(add), y=_lib.ptr(y), sum_in=_lib.ptr(sum_in), sum_out=_lib.ptr(sum_out),
                              sum_scale=float(sum_scale))
        if adam is not None:
            e.adam_t, e.adam_p, e.adam_m, e.adam_v = adam["t"], _lib.ptr(adam["p"]), _lib.ptr(adam["m"]), _lib.ptr(adam["v"])
            e.coef_scratch = _lib.ptr(slab["coef"])
            e.lr, e.beta1, e.beta2, e.eps = adam["lr"], adam["b1"], adam["b2"], adam["eps"]
        desc, n_blocks = self._sched(slab, x.shape[1])
        _lib.check(_lib.lib().rk_spmm_csr_ex(slab["n_rows"], _lib.ptr(slab["rowptr"]), _lib.ptr(slab["col"]), _lib.ptr(slab["val"]),
                                             _lib.ptr(desc), n_blocks, x.shape[1], _lib.ptr(x), x.shape[0],
                                             C.byref(e), _lib.stream_ptr()), "rk_spmm_csr_ex")

    def bpr(self, dim, n_layers, lam, light, emb, gprop, gego, ru, rp, rn, loss_partials):
        _lib.check(_lib.lib().rk_bpr_rows(dim, n_layers, float(lam), _lib.ptr(light), _lib.ptr(emb), _lib.ptr(gprop),
                                          _lib.ptr(gego), _lib.ptr(ru), _lib.ptr(rp), _lib.ptr(rn), ru.numel(),
                                          _lib.ptr(loss_partials), _lib.stream_ptr()), "rk_bpr_rows")


def shard_rows(n_rows, world):
    """rows-per-rank M and the round-robin relabelling: new id = (r % W) * M + r // W."""
    M = (n_rows + world - 1) // world
    return M, (lambda r: (r % world) * M + r // world)


def build_slab(rowptr, col, val, rank, world):
    """CSR slab of the rows owned by `rank` (local order), columns in the gathered layout."""
    rowptr, col, val = (np.asarray(a) for a in (rowptr, col, val))
    N = len(rowptr) - 1
    M, relabel = shard_rows(N, world)
    rows = np.arange(rank, N, world)
    deg = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    lp = np.zeros(M + 1, dtype=np.int64)
    lp[1:len(rows) + 1] = np.cumsum(deg)
    lp[len(rows) + 1:] = lp[len(rows)]
    take = np.concatenate([np.arange(rowptr[r], rowptr[r + 1]) for r in rows]) if len(rows) else np.zeros(0, dtype=np.int64)
    return lp.astype(np.int32), relabel(col[take].astype(np.int64)).astype(np.int32), val[take].astype(np.float32)


class ShardedLightGCN:
    """Trains a LightGCN victim's tables with the node rows sharded over the process group."""

    def __init__(self, n_users, n_items, dim, n_layers, csr, user_emb, item_emb, lam=1e-4, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, group=None, ops=None, device=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ops = ops or HipOps()
        self.U, self.I, self.d, self.L = n_users, n_items, dim, n_layers
        self.lam, self.lr, self.betas, self.eps = lam, lr, betas, eps
        self.N = n_users + n_items
        self.M, self.relabel = shard_rows(self.N, self.world)
        self.device = device if device is not None else user_emb.device
        rowptr, col, val = csr
        self.slab = self.ops.make_slab(*build_slab(rowptr, col, val, self.rank, self.world), self.device)
        dev, M, W, d = self.device, self.M, self.world, dim
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.e0, self.m, self.v = z(M, d), z(M, d), z(M, d)
        self.full = [z(W * M, d) for _ in range(2)]     # gathered X / t buffers (ping-pong)
        self.e0_full, self.light_full = z(W * M, d), z(W * M, d)
        self.y, self.s = z(M, d), z(M, d)
        self.gprop, self.gego = z(W * M, d), z(W * M, d)
        self.t = 0
        self.load_tables(user_emb, item_emb)

    # ------------------------------------------------------------------ table movement
    def _own_rows(self):
        return torch.arange(self.rank, self.N, self.world, device=self.device)

    def load_tables(self, user_emb, item_emb):
        full = torch.cat([user_emb.detach().to(self.device), item_emb.detach().to(self.device)])
        own = self._own_rows()
        self.e0.zero_()
        self.e0[: len(own)] = full[own]

    def _gather(self, local, out):
        if self.world == 1:
            out.copy_(local)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def tables(self):
        """(users[U,d], items[I,d]) in the original order, on every rank."""
        g = self._gather(self.e0, torch.empty_like(self.e0_full))
        idx = self.relabel(torch.arange(self.N, device=self.device))
        full = g[idx]
        return full[: self.U].contiguous(), full[self.U:].contiguous()

    # ------------------------------------------------------------------ one step
    def step(self, users, pos, neg):
        ops, L, M, r = self.ops, self.L, self.M, self.rank
        inv = 1.0 / (L + 1)
        lo = slice(r * M, (r + 1) * M)
        # forward
        x = self._gather(self.e0, self.e0_full)
        for l in range(1, L + 1):
            last = l == L
            ops.spmm(self.slab, x, y=None if last else self.y, sum_in=self.e0 if l == 1 else self.s, sum_out=self.s,
                     sum_scale=inv if last else 1.0)
            if not last:
                x = self._gather(self.y, self.full[l & 1])
        light = self._gather(self.s, self.light_full)
        # BPR, replicated (rows in the gathered layout)
        ru = self.relabel(users)
        rp = self.relabel(pos + self.U)
        rn = self.relabel(neg + self.U)
        lp = torch.zeros(_lib.RK_LOSS_PARTIALS, device=self.device, dtype=torch.float32)
        ops.bpr(self.d, L, self.lam, light, self.e0_full, self.gprop, self.gego, ru.contiguous(), rp.contiguous(), rn.contiguous(), lp)
        # backward + Adam on the owned rows
        self.t += 1
        adam = {"t": self.t, "p": self.e0, "m": self.m, "v": self.v, "lr": self.lr, "b1": self.betas[0], "b2": self.betas[1],
                "eps": self.eps}
        x = self.gprop
        for j in range(1, L + 1):
            last = j == L
            ops.spmm(self.slab, x, add=(self.gego if last else self.gprop)[lo], y=None if last else self.y,
                     adam=adam if last else None)
            if not last:
                x = self._gather(self.y, self.full[j & 1])
        self.gprop.zero_()
        self.gego.zero_()
        return lp

    def train_epoch(self, users, pos, neg, batch):
        """All ranks pass the SAME triplets.  Returns the per-step losses (float64 tensor, host)."""
        n = users.numel()
        parts = []
        for s in range(0, n, batch):
            parts.append(self.step(users[s:s + batch], pos[s:s + batch], neg[s:s + batch]))
        return torch.stack(parts).sum(dim=1).double().cpu()
