"""Synthetic implicit-feedback datasets of a given shape: ml1m/yelp are not
available offline, so the benchmark and the large-size tests use interaction matrices with
log-normal user degrees and rank^-0.8 item popularity, no duplicate (user, item)."""
import numpy as np

SHAPES = {
    # name: (n_users, n_items, train, valid, test edges, seed)
    "ml1m": (5950, 3702, 468649, 49390, 49494, 0),      # readme.md:168-176 of the reference
    "yelp": (54632, 34474, 1600000, 170000, 170000, 1),  # data/readme.md:62; edge counts assumed
    "tiny": (300, 200, 6000, 600, 600, 3),
    "c4s": (250000, 125000, 25000000, 100000, 100000, 2),        # config 4 scaled down 4x per side
    "config4": (1000000, 500000, 100000000, 100000, 100000, 2),  # BASELINE.json config 4
}


def user_item_edges(n_users, n_items, n_edges, seed, min_deg=10, alpha=0.8):
    """CSR (ptr int64[U+1], idx int32[E]) of about n_edges distinct (user, item) pairs, item ids
    sorted within a user."""
    rng = np.random.default_rng(seed)
    deg = rng.lognormal(4.0, 1.0, n_users)
    deg = np.maximum(min_deg, np.round(deg * (n_edges / deg.sum()))).astype(np.int64)
    deg = np.minimum(deg, n_items // 2)
    pop = np.arange(1, n_items + 1, dtype=np.float64) ** (-alpha)
    cdf = np.cumsum(pop / pop.sum())
    perm = rng.permutation(n_items)  # popularity rank -> item id
    out_u, out_i = [], []
    need = deg.copy()
    have = np.zeros(0, dtype=np.int64)
    for _ in range(12):
        todo = np.nonzero(need > 0)[0]
        if len(todo) == 0:
            break
        draws = (need[todo] * 1.3 + 4).astype(np.int64)
        u = np.repeat(todo, draws)
        it = perm[np.minimum(np.searchsorted(cdf, rng.random(len(u))), n_items - 1)]
        keys = np.unique(np.concatenate([have, u * n_items + it]))
        have = keys
        cnt = np.bincount(keys // n_items, minlength=n_users)
        need = np.maximum(deg - cnt, 0)
    users = have // n_items
    items = (have % n_items).astype(np.int32)
    # trim users that overshot (keep a random subset of their items)
    cnt = np.bincount(users, minlength=n_users)
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(cnt)
    over = np.nonzero(cnt > deg)[0]
    keep = np.ones(len(have), dtype=bool)
    for uu in over:
        drop = rng.choice(cnt[uu], size=cnt[uu] - deg[uu], replace=False)
        keep[ptr[uu] + drop] = False
    users, items = users[keep], items[keep]
    cnt = np.bincount(users, minlength=n_users)
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(cnt)
    return ptr, items


def split_edges(ptr, idx, fractions, seed):
    """Random per-edge split of a user->item CSR into len(fractions) CSRs (same row count)."""
    rng = np.random.default_rng(seed + 1000)
    n_users = len(ptr) - 1
    r = rng.random(len(idx))
    edges = np.cumsum(np.asarray(fractions, dtype=np.float64) / np.sum(fractions))
    part = np.searchsorted(edges, r, side="right").clip(0, len(fractions) - 1)
    users = np.repeat(np.arange(n_users), np.diff(ptr))
    out = []
    for k in range(len(fractions)):
        m = part == k
        cnt = np.bincount(users[m], minlength=n_users)
        p = np.zeros(n_users + 1, dtype=np.int64)
        p[1:] = np.cumsum(cnt)
        out.append((p, idx[m].astype(np.int32)))
    return out


def make(name):
    """-> dict(n_users, n_items, train=(ptr, idx), valid=(ptr, idx), test=(ptr, idx))."""
    U, I, tr, va, te, seed = SHAPES[name]
    ptr, idx = user_item_edges(U, I, tr + va + te, seed)
    train, valid, test = split_edges(ptr, idx, (tr, va, te), seed)
    return {"name": name, "n_users": U, "n_items": I, "train": train, "valid": valid, "test": test}


def make_device(name, device):
    """The same family of shapes generated ON the device with torch (seconds instead of minutes for
    the 25 M / 100 M-edge shapes; same degree / popularity model, a different -- torch Philox -- random
    stream, identical on every rank that uses the same seed).  Returns the dict `make` returns, with
    CSRs as (ptr int64, idx int32) device tensors."""
    import torch

    U, I, tr, va, te, seed = SHAPES[name]
    n_edges = tr + va + te
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    deg = torch.exp(4.0 + torch.randn(U, device=dev, generator=g, dtype=torch.float64))
    deg = torch.clamp(torch.round(deg * (n_edges / float(deg.sum()))), min=10).to(torch.int64)
    deg = torch.minimum(deg, torch.full_like(deg, I // 2))
    pop = torch.arange(1, I + 1, device=dev, dtype=torch.float64) ** (-0.8)
    cdf = torch.cumsum(pop / pop.sum(), 0)
    perm = torch.randperm(I, device=dev, generator=g)
    have = torch.zeros(0, dtype=torch.int64, device=dev)
    need = deg.clone()
    for _ in range(12):
        todo = torch.nonzero(need > 0).view(-1)
        if todo.numel() == 0:
            break
        draws = (need[todo].double() * 1.3 + 4).to(torch.int64)
        u = torch.repeat_interleave(todo, draws)
        r = torch.rand(u.numel(), device=dev, generator=g, dtype=torch.float64)
        it = perm[torch.clamp(torch.searchsorted(cdf, r), max=I - 1)]
        have = torch.unique(torch.cat([have, u * I + it]))
        del u, r, it
        cnt = torch.bincount(have // I, minlength=U)
        need = torch.clamp(deg - cnt, min=0)
    users = have // I
    cnt = torch.bincount(users, minlength=U)
    # trim users that overshot: keep a random subset of deg[u] of their items
    prio = torch.rand(have.numel(), device=dev, generator=g, dtype=torch.float64)
    order = torch.argsort(users.double() + prio * 0.999999)     # by user, random inside a user
    start = torch.cumsum(cnt, 0) - cnt
    rank_in_user = torch.arange(have.numel(), device=dev) - start[users[order]]
    keep = order[rank_in_user < deg[users[order]]]
    have = torch.sort(have[keep]).values
    users, items = have // I, (have % I).to(torch.int32)
    part_r = torch.rand(have.numel(), device=dev, generator=g, dtype=torch.float64)
    edges = torch.cumsum(torch.tensor([tr, va, te], dtype=torch.float64, device=dev) / float(n_edges), 0)
    part = torch.clamp(torch.searchsorted(edges, part_r, right=True), max=2)
    out = {"name": name, "n_users": U, "n_items": I}
    for k, key in enumerate(("train", "valid", "test")):
        m = part == k
        c = torch.bincount(users[m], minlength=U)
        p = torch.zeros(U + 1, dtype=torch.int64, device=dev)
        p[1:] = torch.cumsum(c, 0)
        out[key] = (p, items[m].contiguous())
    return out
