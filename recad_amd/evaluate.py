"""Batched full-catalog scoring + top-K + HR@K on device.

Replaces the per-user Python loop + pandas sort of Normal.user_item_model_generate
(recad/workflow/normal.py:57-93) with rk_score_topk: propagate ONCE, then for blocks of
users one fp32-MFMA GEMM and one selection kernel.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def score_plan(nb, n_items, dim, K, n_targets, request=None):
    """rk_score_topk_plan: the library's path choice for this request, or the requested one.  request: None / dict with any of
    path ("gemm" | "panel" | "auto"), panel_rows (16 | 32), panel_ntw (8 | 15), panel_safe (bool) -- what the tests and the
    probes force; the product passes None."""
    req = _lib.ScorePlan()
    if request:
        req.path = {"auto": _lib.RK_SCORE_AUTO, "gemm": _lib.RK_SCORE_GEMM, "panel": _lib.RK_SCORE_PANEL}[request.get("path", "auto")]
        req.panel_rows, req.panel_ntw = int(request.get("panel_rows", 0)), int(request.get("panel_ntw", 0))
        req.panel_safe = 1 if request.get("panel_safe") else 0
    out = _lib.ScorePlan()
    _lib.check(_lib.lib().rk_score_topk_plan(int(nb), int(n_items), int(dim), int(K), int(n_targets), C.byref(req) if request else None,
                                             C.byref(out)), "rk_score_topk_plan")
    return out


# what full_catalog_topk asks the library for when its caller does not say: None = the library's choice.  The GPU tests and the
# probes set it (tests/conftest-style fixtures) -- the library itself reads no environment variable for this any more.
SCORE_REQUEST = None


def full_catalog_topk(victim, user_ids, seen_ptr, seen_idx, targets, K=100, chunk=4096, to_host=True, request=None, extra_scratch_floats=0):
    """For every user in user_ids (int array): top-K unseen items and the score/rank of each target.

    seen_ptr/seen_idx: CSR (indexed by user id) of the items to exclude (the train items), item ids ASCENDING
    within a user (dataset.train_csr_sorted()): the fused sweep walks each list with a cursor.  Host arrays are
    checked and sorted here when needed; device tensors are taken as sorted.
    Returns dict of arrays top_ids[n,K], top_scores[n,K], target_score[n,T], target_rank[n,T]: host numpy
    arrays, or (to_host=False) device tensors left in HBM, no synchronisation.
    """
    _lib.require_gpu()
    if request is None:
        request = SCORE_REQUEST
    tabs = victim.scoring_tables() if hasattr(victim, "scoring_tables") else None
    dot = tabs is not None
    if dot:
        utab, itab, ubias, ibias, mean = tabs
        dev = itab.device
        utab, itab = utab.contiguous(), itab.contiguous()
        if ubias is not None:
            ubias, ibias = ubias.contiguous().view(-1), ibias.contiguous().view(-1)
        n_items, d = itab.shape
    else:  # score_matrix(user_ids, out) victims (NCF): scores are not a dot product
        dev = next(victim.parameters()).device
        n_items = victim.num_items
    as_dev = lambda a: (a.to(device=dev, dtype=torch.int32) if torch.is_tensor(a)
                        else torch.as_tensor(np.asarray(a), dtype=torch.int32, device=dev)).contiguous()
    if not torch.is_tensor(seen_idx) and len(seen_idx) > 1:
        sp, si = np.asarray(seen_ptr).astype(np.int64), np.asarray(seen_idx)
        row = np.repeat(np.arange(len(sp) - 1, dtype=np.int64), np.diff(sp))
        key = row * (int(si.max()) + 1) + si
        if not bool(np.all(key[1:] >= key[:-1])):
            seen_idx = si[np.argsort(key, kind="stable")]
    user_ids_t, seen_ptr_t, seen_idx_t = as_dev(user_ids), as_dev(seen_ptr), as_dev(seen_idx)
    if seen_idx_t.numel() == 0:
        seen_idx_t = torch.zeros(1, dtype=torch.int32, device=dev)
    targets_t = as_dev(targets)
    n, T = user_ids_t.numel(), targets_t.numel()
    top_ids = torch.empty(n, K, dtype=torch.int32, device=dev)
    top_scores = torch.empty(n, K, dtype=torch.float32, device=dev)
    tscore = torch.empty(n, max(T, 1), dtype=torch.float32, device=dev)
    trank = torch.empty(n, max(T, 1), dtype=torch.int32, device=dev)
    chunk = max(1, min(chunk, n))
    plan = None
    if dot:
        # the scoring path is an argument of the call: a plan for the whole user list first -- if the library (or the request)
        # takes the panel form, its scratch does not grow with the block, so everything goes in ONE call; the GEMM + selection
        # path works through blocks of `chunk` users (its scratch is the block's score matrix)
        plan = score_plan(n, n_items, d, K, T, request)
        if plan.path != _lib.RK_SCORE_PANEL:
            plan = score_plan(chunk, n_items, d, K, T, request)
        else:
            chunk = n
        scratch = torch.empty(int(plan.scratch_floats) + extra_scratch_floats, dtype=torch.float32, device=dev)
    else:
        scratch = torch.empty(chunk * n_items, dtype=torch.float32, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        ids = user_ids_t[s:e]
        if not dot:
            victim.score_matrix(ids, scratch[: (e - s) * n_items])
            _lib.check(_lib.lib().rk_topk_rows(
                _lib.ptr(scratch), e - s, n_items, _lib.ptr(ids), _lib.ptr(seen_ptr_t), _lib.ptr(seen_idx_t), K,
                _lib.ptr(top_ids[s:e]), _lib.ptr(top_scores[s:e]), _lib.ptr(targets_t), T, _lib.ptr(tscore[s:e]),
                _lib.ptr(trank[s:e]), _lib.stream_ptr()), "rk_topk_rows")
            continue
        _lib.check(_lib.lib().rk_score_topk(
            d, _lib.ptr(utab), e - s, _lib.ptr(ids), _lib.ptr(itab), n_items, _lib.ptr(ubias), _lib.ptr(ibias),
            float(mean), _lib.ptr(seen_ptr_t), _lib.ptr(seen_idx_t), K, _lib.ptr(top_ids[s:e]), _lib.ptr(top_scores[s:e]),
            _lib.ptr(targets_t), T, _lib.ptr(tscore[s:e]), _lib.ptr(trank[s:e]), C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()),
            "rk_score_topk")
    if not to_host:
        return {"top_ids": top_ids, "top_scores": top_scores, "target_score": tscore[:, :T], "target_rank": trank[:, :T]}
    return {
        "top_ids": top_ids.cpu().numpy(), "top_scores": top_scores.cpu().numpy(),
        "target_score": tscore[:, :T].cpu().numpy(), "target_rank": trank[:, :T].cpu().numpy(),
    }


class EvalSession:
    """A full evaluation of ONE victim on a FIXED user list, set up once and run many times (normal.py:57-93,111-160: propagate,
    score every unseen item, top-K, target score / rank, HR@k numerators): every buffer, the scoring plan and the argument
    marshalling are made in __init__, run() only launches -- and, from its second call on (graph=True), replays ONE hipGraph
    holding the victim's propagation, the GEMM + selection launches of every user block and the HR@k reduction, so an
    evaluation costs one enqueue instead of a Python loop (the perturb-retrain loops that re-score a victim after every
    epoch, bench.py's scorings / s).  Results are device tensors owned by the session, overwritten by the next run().

    The graph holds the victim's table / workspace pointers: it is re-captured when the victim's native handle changes
    (new graph, new dimension), and never used for victims without scoring_tables() (NCF: score_matrix) or with a
    host-synchronising propagation (fuse_layers)."""

    def __init__(self, victim, user_ids, seen_ptr, seen_idx, targets, K=100, topks=(10, 20, 50, 100), chunk=None, request=None, graph=True):
        _lib.require_gpu()
        if not hasattr(victim, "scoring_tables"):
            raise TypeError("EvalSession needs a victim with scoring_tables() (dot-product scoring); use full_catalog_topk")
        self.victim, self.K, self.topks = victim, int(K), tuple(int(k) for k in topks)
        self.request = SCORE_REQUEST if request is None else request
        tabs = victim.scoring_tables()
        if tabs is None:
            raise TypeError("EvalSession: the victim scores through score_matrix() right now (logit dropout active); use full_catalog_topk")
        # victims whose tables are plain parameter views (MF: scoring_tables_static) are asked once -- their scoring_tables() reads
        # a scalar back, which a stream capture does not allow; LightGCN's launches the propagation and is called per run
        self._static = bool(getattr(victim, "scoring_tables_static", False))
        self._static_tabs = tabs if self._static else None
        self._static_key = self._tables_key()
        utab, itab = tabs[0], tabs[1]
        dev = itab.device
        self.dev = dev
        as_dev = lambda a: (a.to(device=dev, dtype=torch.int32) if torch.is_tensor(a)
                            else torch.as_tensor(np.asarray(a), dtype=torch.int32, device=dev)).contiguous()
        self.users, self.seen_ptr, self.seen_idx, self.targets = as_dev(user_ids), as_dev(seen_ptr), as_dev(seen_idx), as_dev(targets)
        if self.seen_idx.numel() == 0:
            self.seen_idx = torch.zeros(1, dtype=torch.int32, device=dev)
        n, T = self.users.numel(), self.targets.numel()
        self.n, self.T = n, T
        n_items, d = itab.shape
        if chunk is None:
            chunk = max(256, min(8192, (1 << 31) // max(n_items, 1)))
        chunk = max(1, min(int(chunk), max(n, 1)))
        plan = score_plan(max(n, 1), n_items, d, self.K, T, self.request)
        if plan.path != _lib.RK_SCORE_PANEL:
            plan = score_plan(chunk, n_items, d, self.K, T, self.request)
        else:
            chunk = max(n, 1)
        self.plan, self.chunk = plan, chunk
        self.scratch = torch.empty(int(plan.scratch_floats), dtype=torch.float32, device=dev)
        self.top_ids = torch.empty(n, self.K, dtype=torch.int32, device=dev)
        self.top_scores = torch.empty(n, self.K, dtype=torch.float32, device=dev)
        self.tscore = torch.empty(n, max(T, 1), dtype=torch.float32, device=dev)
        self.trank = torch.empty(n, max(T, 1), dtype=torch.int32, device=dev)
        self.ks = torch.as_tensor(list(self.topks), dtype=torch.int32, device=dev)
        self.counts = torch.zeros(max(T, 1), len(self.topks), dtype=torch.int32, device=dev)
        self.want_graph = bool(graph) and not getattr(victim, "fuse_layers", False)
        self._graph, self._graph_key, self._runs = None, None, 0

    def _tables_key(self):
        """Static victims (MF): what the cached tables -- and a captured graph's pointers and baked `mean` -- depend on
        (victim.scoring_tables_key(): storages, shapes, `mean`'s version, dropout state; no device read)."""
        if not self._static:
            return None
        fn = getattr(self.victim, "scoring_tables_key", None)
        return fn() if fn is not None else tuple((t.data_ptr(), tuple(t.shape)) for t in self._static_tabs[:4] if t is not None)

    def _refresh_static(self):
        """Called by run() in front of every launch / replay: re-query a static victim's tables when their key moved."""
        if not self._static:
            return
        key = self._tables_key()
        if key != self._static_key:
            tabs = self.victim.scoring_tables()
            if tabs is None:
                raise RuntimeError("EvalSession: the victim scores through score_matrix() now (logit dropout active); use full_catalog_topk")
            if tuple(tabs[1].shape) != (self.plan.n_items, self.plan.dim):
                raise RuntimeError("EvalSession: the victim's item table changed shape; build a new session")
            self._static_tabs, self._static_key = tabs, key
            self._graph, self._graph_key = None, None     # the captured pointers / mean are stale: direct launches, re-capture

    def _launch(self):
        v, L, plan = self.victim, _lib.lib(), self.plan
        utab, itab, ubias, ibias, mean = self._static_tabs if self._static_tabs is not None else v.scoring_tables()   # (LightGCN: the propagation's launches)
        utab, itab = utab.contiguous(), itab.contiguous()
        if ubias is not None:
            ubias, ibias = ubias.contiguous().view(-1), ibias.contiguous().view(-1)
        n_items, d = itab.shape
        st = _lib.stream_ptr()
        for s in range(0, self.n, self.chunk):
            e = min(self.n, s + self.chunk)
            _lib.check(L.rk_score_topk(
                d, _lib.ptr(utab), e - s, _lib.ptr(self.users[s:e]), _lib.ptr(itab), n_items, _lib.ptr(ubias), _lib.ptr(ibias), float(mean),
                _lib.ptr(self.seen_ptr), _lib.ptr(self.seen_idx), self.K, _lib.ptr(self.top_ids[s:e]), _lib.ptr(self.top_scores[s:e]),
                _lib.ptr(self.targets), self.T, _lib.ptr(self.tscore[s:e]), _lib.ptr(self.trank[s:e]), C.byref(plan), _lib.ptr(self.scratch), st),
                "rk_score_topk")
        if self.T and self.n:
            _lib.check(L.rk_hit_counts(_lib.ptr(self.trank), self.n, self.T, _lib.ptr(self.ks), len(self.topks), _lib.ptr(self.counts), st), "rk_hit_counts")

    def run(self):
        """-> dict of device tensors (top_ids [n, K], top_scores, target_score [n, T], target_rank, hit_counts [T, len(topks)]);
        no synchronisation."""
        self._runs += 1
        self._refresh_static()
        key = (getattr(self.victim, "_handle_key", None), self._static_key)
        if self.want_graph and self._graph is not None and self._graph_key == key:
            self._graph.replay()
        else:
            self._launch()
            if self.want_graph and self._runs >= 2 and (self._graph is None or self._graph_key != key):
                # capture on the SECOND run (a single evaluation never pays for a capture; handles, plans and lazily built
                # schedules exist by now); a failed capture keeps the eager path
                try:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._launch()
                    self._graph, self._graph_key = g, (getattr(self.victim, "_handle_key", None), self._static_key)
                    self._graph.replay()     # (the capture itself ran nothing: results must come from an execution)
                except Exception as exc:    # noqa: BLE001
                    import warnings
                    warnings.warn(f"EvalSession: hipGraph capture unavailable ({type(exc).__name__}: {exc}); evaluating with direct launches")
                    self.want_graph, self._graph = False, None
                    torch.cuda.synchronize()
        return {"top_ids": self.top_ids, "top_scores": self.top_scores, "target_score": self.tscore[:, :self.T],
                "target_rank": self.trank[:, :self.T], "hit_counts": self.counts[: max(self.T, 0)]}


_KS_CACHE = {}


def hit_counts(target_rank, topks):
    """HR@k numerators on the device: int32 tensor [n_targets, len(topks)] with the number of users whose
    target ranks below k (normal.py:86-92); divide by the number of users for HR@k.  No synchronisation."""
    n, T = target_rank.shape
    key = (target_rank.device, tuple(int(k) for k in topks))
    ks = _KS_CACHE.get(key)
    if ks is None:  # a host->device copy per call would stall the stream
        ks = _KS_CACHE[key] = torch.as_tensor(list(key[1]), dtype=torch.int32, device=target_rank.device)
    counts = torch.empty(T, len(ks), dtype=torch.int32, device=target_rank.device)
    _lib.check(_lib.lib().rk_hit_counts(_lib.ptr(target_rank.contiguous()), n, T, _lib.ptr(ks), len(ks), _lib.ptr(counts),
                                        _lib.stream_ptr()), "rk_hit_counts")
    return counts


def eligible_users_device(seen_ptr, seen_idx, targets, device):
    """eligible_users() on the device (rk_eligible_users): -> (user_ids int32 device tensor [n], the seen CSR and the
    targets as int32 device tensors).  One scalar read-back (the count sizes every later buffer)."""
    _lib.require_gpu()
    as_dev = lambda a: (a.to(device=device, dtype=torch.int32) if torch.is_tensor(a)
                        else torch.as_tensor(np.asarray(a), dtype=torch.int32, device=device)).contiguous()
    ptr, idx, tg = as_dev(seen_ptr), as_dev(seen_idx), as_dev(targets)
    if idx.numel() == 0:
        idx = torch.zeros(1, dtype=torch.int32, device=device)
    U = ptr.numel() - 1
    flags = torch.empty(max(U, 1), dtype=torch.int32, device=device)
    ids = torch.empty(max(U, 1), dtype=torch.int32, device=device)
    count = torch.zeros(1, dtype=torch.int32, device=device)
    _lib.check(_lib.lib().rk_eligible_users(U, _lib.ptr(ptr), _lib.ptr(idx), _lib.ptr(tg), tg.numel(), _lib.ptr(flags), _lib.ptr(ids),
                                            _lib.ptr(count), _lib.stream_ptr()), "rk_eligible_users")
    return ids[: int(count.item())], ptr, idx, tg


def pred_shift(score_before, score_after):
    """mean(score_after - score_before) over all (user, target) rows (normal.py:147-149) as a device double[2]
    tensor {mean, sum}; no synchronisation."""
    a, b = score_before.contiguous().view(-1), score_after.contiguous().view(-1)
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().rk_pred_shift(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(out), _lib.stream_ptr()), "rk_pred_shift")
    return out


def eligible_users(train_ptr, train_idx, targets):
    """Users the reference evaluates (normal.py:133-143): every user with a train list that
    contains none of the targets."""
    train_ptr = np.asarray(train_ptr)
    train_idx = np.asarray(train_idx)
    U = len(train_ptr) - 1
    deg = np.diff(train_ptr)
    has_target = np.zeros(U, dtype=bool)
    rows = np.repeat(np.arange(U), deg)
    hit = np.isin(train_idx, np.asarray(targets))
    has_target[rows[hit]] = True
    return np.nonzero((deg > 0) & ~has_target)[0].astype(np.int32)


def hr_rows(user_ids, res, topks):
    """[user, score(target), hit@k...] rows like normal.py:89-92 (one target per row)."""
    rows = []
    T = res["target_score"].shape[1]
    for t in range(T):
        hits = [(res["target_rank"][:, t] < k).astype(np.float64) for k in topks]
        rows.append(np.stack([np.asarray(user_ids, dtype=np.float64), res["target_score"][:, t].astype(np.float64)] + hits, 1))
    return np.concatenate(rows, 0) if rows else np.zeros((0, 2 + len(topks)))
