"""Loading of the golden fixtures (tests/golden/*.npz, made by make_golden.py) and
regeneration of the seeded inputs that script used instead of storing them."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def seeded_table(seed, rows, dim, std):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((rows, dim), dtype=np.float32) * np.float32(std)).astype(np.float32)


def seeded_uniform(seed, rows, dim, lo, hi):
    rng = np.random.default_rng(seed)
    return (rng.random((rows, dim), dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def lightgcn_init(g):
    U, I, d = int(g["n_users"]), int(g["n_items"]), int(g["dim"])
    return seeded_table(11, U, d, 0.1), seeded_table(12, I, d, 0.1)


def mf_init(g):
    U, I, d = int(g["n_users"]), int(g["n_items"]), int(g["dim"])
    return (seeded_uniform(21, U, d, 0, 0.005), seeded_uniform(22, I, d, 0, 0.005),
            seeded_uniform(23, U, 1, -0.01, 0.01), seeded_uniform(24, I, 1, -0.01, 0.01))


def ncf_init(g):
    U, I, f, L = int(g["n_users"]), int(g["n_items"]), int(g["factor"]), int(g["layers"])
    E = f * 2 ** (L - 1)
    emb = (seeded_table(31, U, f, 0.01), seeded_table(32, I, f, 0.01), seeded_table(33, U, E, 0.01),
           seeded_table(34, I, E, 0.01))
    W, b = [], []
    names = [str(n) for n in g["dense_names"]]
    pw = None
    for j, n in enumerate(names):
        if n.startswith("MLP_layers"):
            l = len(W)
            inn = f * 2 ** (L - l)
            out = inn // 2
            a = float(np.sqrt(6.0 / (inn + out)))
            W.append(seeded_uniform(40 + j, out, inn, -a, a))
            b.append(np.zeros(out, dtype=np.float32))
        else:
            a = float(np.sqrt(6.0 / (1 + 2 * f)))
            pw = seeded_uniform(40 + j, 1, 2 * f, -a, a)
    return emb, W, b, pw, np.zeros(1, dtype=np.float32)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


TIE_RTOL = 2e-6  # adjacent scores closer than this (relative) are "tied" for ranking purposes


def compare_topk_lists(mine, ref_ids, ref_scores, score_rtol=1e-5):
    """Top-K parity as north_star states it: index lists identical on tie-free
    prefixes.  `mine` = [(ids, scores)] per user.  Every position where the ids
    differ must sit inside a run of reference scores that are tied to within
    TIE_RTOL (the reference's own order there is unspecified); the
    score at every rank must agree to score_rtol.  Returns #users whose list is
    identical outright."""
    exact = 0
    for r, (ids, sc) in enumerate(mine):
        rid, rsc = ref_ids[r], ref_scores[r].astype(np.float64)
        valid = rid >= 0
        assert np.array_equal(np.asarray(ids)[~valid], rid[~valid])
        assert np.allclose(np.asarray(sc)[valid], rsc[valid], rtol=score_rtol, atol=1e-7), r
        diff = np.nonzero((np.asarray(ids) != rid) & valid)[0]
        if len(diff) == 0:
            exact += 1
            continue
        scale = np.maximum(np.abs(rsc), 1e-30)
        for k in diff:
            lo = abs(rsc[k] - rsc[k - 1]) / scale[k] if k > 0 else np.inf
            hi = abs(rsc[k] - rsc[k + 1]) / scale[k] if k + 1 < len(rsc) and rid[k + 1] >= 0 else np.inf
            # the last slot can also trade places with the (unrecorded) 101st item
            edge = k == len(rsc) - 1
            assert min(lo, hi) <= TIE_RTOL or edge, (r, k, lo, hi)
    return exact


def adam_close(a, b, lr, steps, rtol=1e-4, outlier_frac=1e-3, travel_frac=0.25):
    """Parity of Adam-trained tensors.  Adam's m/(sqrt(v)+eps) maps summation-order
    noise on near-cancelling gradient entries to O(lr) differences, so a pure
    max-norm bound is not meaningful for every entry: require (1) all but
    `outlier_frac` of the entries within rtol*max|b|, and (2) no entry further than
    travel_frac of the distance Adam can move it in `steps` steps."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    diff = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-30)
    n_bad = int((diff > rtol * scale).sum())
    # small tensors: a handful of near-cancelling entries is already above any sensible fraction
    return n_bad <= max(outlier_frac * diff.size, 8) and diff.max() <= travel_frac * lr * steps, (n_bad, diff.size, diff.max())


def check_eval_rows_multi(g, users, target_score, target_rank, top_ids, top_scores, score_rtol=1e-5, min_exact=0.85, lists=True):
    """An evaluation (per user: scores of the targets, their ranks among the unseen items, the top-101 list) against a golden that
    holds the REFERENCE's evaluation rows for several targets -- eval_rows[t] = user_item_model_generate's [user, score(target),
    hit@k...] (recad/workflow/normal.py:57-93), one target at a time -- and its top-100 lists.  North_star's bar: target scores to
    `score_rtol`; hit flags identical on every row whose target is not within the observed score noise of the cutoff; HR@k within
    1e-4 relative + those cutoff-ambiguous rows (asserted to stay a handful); lists identical on tie-free prefixes.  target_rank /
    top_* may cover only the first rows of `users` when the caller scored the full catalogue for fewer users (the oracle at L = 5);
    rows without a rank are compared on the target scores alone.  Returns (#lists identical outright, #lists, #ambiguous rows)."""
    ref = g["eval_rows"]                       # [T, n, 2 + len(topks)]
    T, n = ref.shape[0], ref.shape[1]
    topks = [int(k) for k in g["topks"]]
    assert np.array_equal(np.asarray(users), g["eval_users"]) and T == len(g["target_ids"])
    ts = np.asarray(target_score, dtype=np.float64)
    assert ts.shape == (n, T)
    for t in range(T):
        assert np.array_equal(ref[t][:, 0], np.asarray(users, dtype=np.float64))
        # (an untrained victim's scores are sums of cancelling terms around 0: "1e-5" is of the largest target score, like every
        # table / gradient tolerance of this suite, not of each -- possibly tiny -- entry)
        assert np.abs(ts[:, t] - ref[t][:, 1]).max() <= score_rtol * np.abs(ref[:, :, 1]).max(), (t, np.abs(ts[:, t] - ref[t][:, 1]).max())
    n_rank = 0 if target_rank is None else len(target_rank)
    n_amb_total = 0
    if n_rank:
        K = top_scores.shape[1] - 1
        sc = np.asarray(top_scores, dtype=np.float64)
        tol = 4.0 * max(np.abs(ts[:, t] - ref[t][:, 1]).max() for t in range(T)) + 1e-12
        for t in range(T):
            rank = np.asarray(target_rank)[:, t]
            for q, k in enumerate(topks):
                mine = (rank < k).astype(np.float64)
                other = np.where(rank < k, sc[:, min(k, K)], sc[:, k - 1])   # first item outside the cutoff / last item inside it
                amb = np.isfinite(other) & (np.abs(ts[:n_rank, t] - other) <= tol)
                rflag = ref[t][:n_rank, 2 + q]
                assert np.array_equal(mine[~amb], rflag[~amb]), (t, k, np.nonzero(mine != rflag)[0])
                if n_rank == n:
                    assert abs(mine.mean() - rflag.mean()) <= 1e-4 * max(rflag.mean(), 1e-12) + amb.sum() / n, (t, k, int(amb.sum()))
                assert amb.sum() <= 0.02 * n_rank + 2, (t, k, int(amb.sum()))   # a handful, not a blanket
                n_amb_total += int(amb.sum())
    exact = n_lists = 0
    if lists and n_rank:
        mine = [(np.asarray(top_ids)[r][:100], np.asarray(top_scores)[r][:100]) for r in range(n_rank)]
        n_lists = n_rank
        exact = compare_topk_lists(mine, g["top_ids"][:n_rank], g["top_scores"][:n_rank], score_rtol=score_rtol)
        assert exact >= min_exact * n_lists, (exact, n_lists)
    return exact, n_lists, n_amb_total
