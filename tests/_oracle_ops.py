"""Oracle-backed stand-in for recad_amd.sharded.HipOps so the row-sharded trainer's
partitioning and collectives can be exercised on CPU (gloo).  Test infrastructure only."""
import ctypes as C

import numpy as np
import torch

from oracle import oracle as orc


class OracleOps:
    name = "oracle"

    def make_slab(self, rowptr, col, val, device):
        a = lambda x, dt: np.ascontiguousarray(torch.as_tensor(x).cpu().numpy(), dtype=dt)
        return {"n_rows": len(rowptr) - 1, "csr": (a(rowptr, np.int32), a(col, np.int32), a(val, np.float32))}

    def spmm(self, slab, x, add=None, y=None, sum_in=None, sum_out=None, sum_scale=1.0, adam=None):
        rp, c, v = slab["csr"]
        n = slab["n_rows"]
        # the oracle's SpMM looks X rows up by column id, so X is passed as is
        X = np.ascontiguousarray(x.numpy(), dtype=np.float32)
        out = np.zeros((n, X.shape[1]), dtype=np.float32)
        orc.lib().orc_spmm(C.c_int32(n), orc._p(rp), orc._p(c), orc._p(v), C.c_int32(X.shape[1]), orc._p(X), orc._p(out))
        t = torch.from_numpy(out)
        if add is not None:
            t = t + add
        if y is not None:
            y.copy_(t)
        if sum_out is not None:
            sum_out.copy_((sum_in + t) * np.float32(sum_scale))
        if adam is not None:
            # p/m/v are row-slices (views) of the trainer's buffers: update them in place
            p, m, vv = (np.ascontiguousarray(adam[k].numpy()) for k in ("p", "m", "v"))
            orc.adam(p, np.ascontiguousarray(t.numpy()), m, vv, adam["t"], adam["lr"], adam["b1"], adam["b2"], adam["eps"])
            for k, arr in (("p", p), ("m", m), ("v", vv)):
                adam[k].copy_(torch.from_numpy(arr))

    def adam(self, p, g, m, v, t, lr, b1, b2, eps, coef=None):
        pn, mn, vn = (np.ascontiguousarray(a.numpy()) for a in (p, m, v))
        orc.adam(pn, np.ascontiguousarray(g.numpy()), mn, vn, t, lr, b1, b2, eps)
        for dst, arr in ((p, pn), (m, mn), (v, vn)):
            dst.copy_(torch.from_numpy(arr))

    def gather_rows(self, src, idx, mask, out):
        rows = src[idx]
        out.copy_(rows if mask is None else rows * mask.view(-1, 1))

    def zero_rows(self, a, b, idx):
        a[idx] = 0.0
        if b is not None:
            b[idx] = 0.0

    def bpr(self, dim, n_layers, lam, light_rows, emb, gprop, gego, ru, rp, rn, loss_partials, keys=None):   # keys: the HIP ops' ordered plan; this loop is ordered anyway
        """light_rows: compact [3B, d] (users, positives, negatives); emb/gprop/gego indexed by ru/rp/rn."""
        R = light_rows.numpy()
        E = emb.numpy()
        gp, ge = gprop.numpy(), gego.numpy()
        ru, rp, rn = ru.numpy(), rp.numpy(), rn.numpy()
        B = len(ru)
        Lu, Lp, Ln = R[:B], R[B:2 * B], R[2 * B:3 * B]
        inv = np.float32(1.0 / (n_layers + 1))
        ps = (Lu * Lp).sum(1)
        ns = (Lu * Ln).sum(1)
        x = (ns - ps).astype(np.float32)
        sp = np.where(x > 20, x, np.log1p(np.exp(x)))
        reg = (E[ru] ** 2).sum() + (E[rp] ** 2).sum() + (E[rn] ** 2).sum()
        dx = (np.where(x > 20, 1.0, 1.0 / (1.0 + np.exp(-x))) / B * inv).astype(np.float32)[:, None]
        c = np.float32(lam / B)
        for rows, g in ((ru, dx * (Ln - Lp)), (rp, -dx * Lu), (rn, dx * Lu)):
            np.add.at(gp, rows, g)
            np.add.at(ge, rows, g + c * E[rows])
        loss_partials.zero_()
        loss_partials[0] = float(sp.sum() / B + lam * 0.5 * reg / B)

    def score_topk(self, utab, user_rows, itab, seen_ptr, seen_idx, targets, K):
        ut, it = utab.numpy(), np.ascontiguousarray(itab.numpy())
        rows = user_rows.numpy()
        ptr, idx, tg = seen_ptr.numpy(), seen_idx.numpy(), targets.numpy()
        n, T = len(rows), len(tg)
        ts, tr, ids = np.zeros((n, T), np.float32), np.zeros((n, T), np.int32), np.full((n, K), -1, np.int32)
        for k, q in enumerate(rows):
            s = orc.score_rows(np.ascontiguousarray(ut[q:q + 1]), it)[0]
            top, _, tsc, trk = orc.topk_row(s, idx[ptr[q]:ptr[q + 1]], K, tg)
            ids[k, :len(top)] = top[:K]
            ts[k], tr[k] = tsc, trk
        return torch.from_numpy(ts), torch.from_numpy(tr), torch.from_numpy(ids)
