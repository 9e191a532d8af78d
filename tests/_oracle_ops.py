"""Oracle-backed stand-in for recad_amd.sharded.HipOps so the row-sharded trainer's
partitioning and collectives can be exercised on CPU (gloo).  Test infrastructure only."""
import numpy as np
import torch

from oracle import oracle as orc


class OracleOps:
    name = "oracle"

    def make_slab(self, rowptr, col, val, device):
        return {"n_rows": len(rowptr) - 1, "csr": (np.ascontiguousarray(rowptr, dtype=np.int32),
                                                   np.ascontiguousarray(col, dtype=np.int32), np.ascontiguousarray(val, dtype=np.float32))}

    def spmm(self, slab, x, add=None, y=None, sum_in=None, sum_out=None, sum_scale=1.0, adam=None):
        rp, c, v = slab["csr"]
        n = slab["n_rows"]
        # the oracle's SpMM takes a square-ish X: rows are looked up by column id, so pass X as is
        X = np.ascontiguousarray(x.numpy(), dtype=np.float32)
        out = np.zeros((n, X.shape[1]), dtype=np.float32)
        import ctypes as C
        orc.lib().orc_spmm(C.c_int32(n), orc._p(rp), orc._p(c), orc._p(v), C.c_int32(X.shape[1]), orc._p(X), orc._p(out))
        t = torch.from_numpy(out)
        if add is not None:
            t = t + add
        if y is not None:
            y.copy_(t)
        if sum_out is not None:
            sum_out.copy_((sum_in + t) * np.float32(sum_scale))
        if adam is not None:
            p, m, vv = (adam[k].numpy() for k in ("p", "m", "v"))
            orc.adam(p, t.numpy(), m, vv, adam["t"], adam["lr"], adam["b1"], adam["b2"], adam["eps"])

    def bpr(self, dim, n_layers, lam, light, emb, gprop, gego, ru, rp, rn, loss_partials):
        L = light.numpy()
        E = emb.numpy()
        gp, ge = gprop.numpy(), gego.numpy()
        ru, rp, rn = ru.numpy(), rp.numpy(), rn.numpy()
        B = len(ru)
        inv = np.float32(1.0 / (n_layers + 1))
        ps = (L[ru] * L[rp]).sum(1)
        ns = (L[ru] * L[rn]).sum(1)
        x = (ns - ps).astype(np.float32)
        sp = np.where(x > 20, x, np.log1p(np.exp(x)))
        reg = (E[ru] ** 2).sum() + (E[rp] ** 2).sum() + (E[rn] ** 2).sum()
        dx = (np.where(x > 20, 1.0, 1.0 / (1.0 + np.exp(-x))) / B * inv).astype(np.float32)[:, None]
        c = np.float32(lam / B)
        for rows, g in ((ru, dx * (L[rn] - L[rp])), (rp, -dx * L[ru]), (rn, dx * L[ru])):
            np.add.at(gp, rows, g)
            np.add.at(ge, rows, g + c * E[rows])
        loss_partials.zero_()
        loss_partials[0] = float(sp.sum() / B + lam * 0.5 * reg / B)
