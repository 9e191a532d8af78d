"""ctypes/numpy front-end of the CPU oracle (oracle/recad_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  Parity status: pinned against
golden vectors captured from the reference (tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "recad_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_lightgcn_step.restype = C.c_float
        _LIB.orc_lightgcn_step_general.restype = C.c_float
        _LIB.orc_mf_step.restype = C.c_float
        _LIB.orc_ncf_grads.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f(x):
    return C.c_float(float(x))


def c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def coo_to_csr(n, row, col, val):
    """Coalesced COO -> CSR (rowptr int32[n+1], col int32, val fp32)."""
    row = np.ascontiguousarray(row, dtype=np.int32)
    assert np.all(np.diff(row.astype(np.int64)) >= 0), "COO must be row-sorted (coalesced)"
    rowptr = np.zeros(n + 1, dtype=np.int32)
    lib().orc_coo_to_csr(C.c_int32(n), C.c_int64(len(row)), _p(row), _p(rowptr))
    return rowptr, np.ascontiguousarray(col, dtype=np.int32), c32(val)


def build_norm_adj(U, I, rptr, ridx):
    rptr = np.ascontiguousarray(rptr, dtype=np.int32)
    ridx = np.ascontiguousarray(ridx, dtype=np.int32)
    E = int(rptr[U])
    rowptr = np.zeros(U + I + 1, dtype=np.int32)
    col = np.zeros(2 * E, dtype=np.int32)
    val = np.zeros(2 * E, dtype=np.float32)
    lib().orc_build_norm_adj(C.c_int32(U), C.c_int32(I), _p(rptr), _p(ridx), _p(rowptr), _p(col), _p(val))
    return rowptr, col, val


def spmm(rowptr, col, val, X):
    X = c32(X)
    Y = np.empty_like(X)
    lib().orc_spmm(C.c_int32(len(rowptr) - 1), _p(rowptr), _p(col), _p(val), C.c_int32(X.shape[1]), _p(X), _p(Y))
    return Y


def lightgcn_propagate(csr, user, item, L):
    rowptr, col, val = csr
    user, item = c32(user), c32(item)
    U, d = user.shape
    I = item.shape[0]
    light = np.empty((U + I, d), dtype=np.float32)
    lib().orc_lightgcn_propagate(C.c_int32(U), C.c_int32(I), C.c_int32(d), C.c_int32(L), _p(rowptr), _p(col), _p(val),
                                 _p(user), _p(item), _p(light))
    return light


class AdamState:
    def __init__(self, *shapes):
        self.m = [np.zeros(s, dtype=np.float32) for s in shapes]
        self.v = [np.zeros(s, dtype=np.float32) for s in shapes]
        self.t = 0


def adam(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    lib().orc_adam(C.c_int64(p.size), _p(p), _p(c32(g)), _p(m), _p(v), C.c_int32(t), _f(lr), _f(b1), _f(b2), _f(eps))


def lightgcn_step(csr, user, item, state, users, pos, neg, L, lam=1e-4, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8,
                  want_grads=False, apply_update=True, csr_t=None):
    """In-place on user/item/state.  Returns loss (and grads when asked).  csr_t: the transpose of a
    non-symmetric (dropped-out) forward graph, applied in the backward; None = symmetric."""
    rowptr, col, val = csr
    rowptr_t, col_t, val_t = csr if csr_t is None else csr_t
    U, d = user.shape
    I = item.shape[0]
    users = np.ascontiguousarray(users, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    neg = np.ascontiguousarray(neg, dtype=np.int64)
    gu = np.empty_like(user) if want_grads else None
    gi = np.empty_like(item) if want_grads else None
    if apply_update:
        state.t += 1
    loss = lib().orc_lightgcn_step_general(
        C.c_int32(U), C.c_int32(I), C.c_int32(d), C.c_int32(L), _p(rowptr), _p(col), _p(val), _p(rowptr_t), _p(col_t),
        _p(val_t), _p(user), _p(item),
        _p(state.m[0]), _p(state.v[0]), _p(state.m[1]), _p(state.v[1]), C.c_int32(max(state.t, 1)), _p(users), _p(pos),
        _p(neg), C.c_int32(len(users)), _f(lam), _f(lr), _f(b1), _f(b2), _f(eps), _p(gu), _p(gi),
        C.c_int32(1 if apply_update else 0))
    return (float(loss), gu, gi) if want_grads else float(loss)


def pair_scores(utab, itab, users, items, ubias=None, ibias=None, mean=0.0):
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    out = np.empty(len(users), dtype=np.float32)
    lib().orc_pair_scores(C.c_int32(utab.shape[1]), _p(c32(utab)), _p(c32(itab)), _p(ubias), _p(ibias), _f(mean),
                          _p(users), _p(items), C.c_int64(len(users)), _p(out))
    return out


class MFParams:
    """ue[U,d], ie[I,d], ub[U], ib[I]; moments in one flat buffer like the C side expects."""

    def __init__(self, ue, ie, ub, ib, mean):
        self.ue, self.ie = c32(ue).copy(), c32(ie).copy()
        self.ub, self.ib = c32(ub).reshape(-1).copy(), c32(ib).reshape(-1).copy()
        self.mean = float(mean)
        self.tot = self.ue.size + self.ie.size + self.ub.size + self.ib.size
        self.mom = np.zeros(2 * self.tot, dtype=np.float32)
        self.t = 0


def mf_step(P, users, items, labels, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, want_grads=False, apply_update=True):
    U, d = P.ue.shape
    I = P.ie.shape[0]
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    g = np.empty(P.tot, dtype=np.float32) if want_grads else None
    if apply_update:
        P.t += 1
    loss = lib().orc_mf_step(C.c_int32(U), C.c_int32(I), C.c_int32(d), _p(P.ue), _p(P.ie), _p(P.ub), _p(P.ib),
                             _f(P.mean), _p(P.mom), C.c_int32(max(P.t, 1)), _p(users), _p(items), _p(labels),
                             C.c_int32(len(users)), _f(lr), _f(b1), _f(b2), _f(eps), _p(g),
                             C.c_int32(1 if apply_update else 0))
    if want_grads:
        a, b = P.ue.size, P.ue.size + P.ie.size
        return float(loss), (g[:a].reshape(U, d), g[a:b].reshape(I, d), g[b:b + U], g[b + U:])
    return float(loss)


class NCFParams:
    def __init__(self, f, L, ug, ig, um, im, W, b, pw, pb):
        self.f, self.L = f, L
        self.ug, self.ig, self.um, self.im = (c32(x).copy() for x in (ug, ig, um, im))
        self.W = [c32(w).copy() for w in W]
        self.b = [c32(x).copy() for x in b]
        self.pw = c32(pw).reshape(-1).copy()
        self.pb = np.asarray([float(np.asarray(pb).reshape(-1)[0])], dtype=np.float32)
        self.t = 0
        self.mom = None

    def tensors(self):
        return [self.ug, self.ig, self.um, self.im] + self.W + self.b + [self.pw, self.pb]


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def ncf_forward(P, users, items):
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    out = np.empty(len(users), dtype=np.float32)
    lib().orc_ncf_forward(C.c_int32(P.f), C.c_int32(P.L), _p(P.ug), _p(P.ig), _p(P.um), _p(P.im), _ptr_array(P.W),
                          _ptr_array(P.b), _p(P.pw), _f(P.pb[0]), _p(users), _p(items), C.c_int64(len(users)), _p(out))
    return out


def ncf_step(P, users, items, labels, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, apply_update=True):
    users = np.ascontiguousarray(users, dtype=np.int64)
    items = np.ascontiguousarray(items, dtype=np.int64)
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    g = [np.empty_like(t) for t in P.tensors()]
    gW, gb = g[4:4 + P.L], g[4 + P.L:4 + 2 * P.L]
    loss = lib().orc_ncf_grads(C.c_int32(P.ug.shape[0]), C.c_int32(P.ig.shape[0]), C.c_int32(P.f), C.c_int32(P.L),
                               _p(P.ug), _p(P.ig), _p(P.um), _p(P.im), _ptr_array(P.W), _ptr_array(P.b), _p(P.pw),
                               _f(P.pb[0]), _p(users), _p(items), _p(labels), C.c_int32(len(users)), _p(g[0]), _p(g[1]),
                               _p(g[2]), _p(g[3]), _ptr_array(gW), _ptr_array(gb), _p(g[-2]), _p(g[-1]))
    if apply_update:
        if P.mom is None:
            P.mom = [(np.zeros_like(t), np.zeros_like(t)) for t in P.tensors()]
        P.t += 1
        for t, gt, (m, v) in zip(P.tensors(), g, P.mom):
            adam(t, gt, m, v, P.t, lr, b1, b2, eps)
    return float(loss), g


def score_rows(urows, itab, ubias_rows=None, ibias=None, mean=0.0):
    urows, itab = c32(urows), c32(itab)
    out = np.empty((urows.shape[0], itab.shape[0]), dtype=np.float32)
    lib().orc_score_rows(C.c_int32(urows.shape[1]), _p(urows), C.c_int32(urows.shape[0]), _p(itab),
                         C.c_int32(itab.shape[0]), _p(ubias_rows), _p(ibias), _f(mean), _p(out))
    return out


def topk_row(scores, seen, K, targets):
    scores = c32(scores)
    seen = np.ascontiguousarray(seen, dtype=np.int32)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    top_ids = np.empty(K, dtype=np.int32)
    top_scores = np.empty(K, dtype=np.float32)
    ts = np.empty(len(targets), dtype=np.float32)
    tr = np.empty(len(targets), dtype=np.int32)
    lib().orc_topk_row(C.c_int32(len(scores)), _p(scores), _p(seen), C.c_int32(len(seen)), C.c_int32(K), _p(top_ids),
                       _p(top_scores), _p(targets), C.c_int32(len(targets)), _p(ts), _p(tr))
    return top_ids, top_scores, ts, tr


def evaluate(score_fn, n_items, train_ptr, train_idx, targets, topks, K=100, users=None):
    """Restatement of normal_evaluate's per-model half (normal.py:111-160): eligible
    users = users with a train list that does not contain any target; rows =
    [user, score(target), hit@k...].  score_fn(u) -> fp32[n_items]."""
    rows, tops = [], {}
    tset = set(int(t) for t in targets)
    U = len(train_ptr) - 1
    for u in (range(U) if users is None else users):
        seen = train_idx[train_ptr[u]:train_ptr[u + 1]]
        if len(seen) == 0 or tset & set(int(x) for x in seen):
            continue
        ids, sc, ts, tr = topk_row(score_fn(u), seen, K, targets)
        tops[u] = (ids, sc)
        for t in range(len(targets)):
            rows.append([u, float(ts[t])] + [1.0 if tr[t] < k else 0.0 for k in topks])
    return np.asarray(rows, dtype=np.float64), tops
