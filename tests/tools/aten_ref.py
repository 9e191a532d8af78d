"""The reference's LightGCN op sequence restated against ATen on the host cores -- SURVEY.md 8d's CPU baseline.

Test infrastructure (like oracle/): tests/test_aten_ref.py pins it against the reference's goldens, and bench.py's
cpu_baseline leg times it on the GPU box's host with k threads.  Never imported by recad_amd/.

Follows /root/reference/recad/model/victim/lightgcn.py line by line:
  computer()        :82-113   cat, L x torch.sparse.mm on the coalesced COO graph, stack, mean, split
  getEmbedding()    :122-130  six index gathers
  train_step() body :137-169  reg = 1/2 (|u0|^2 + |p0|^2 + |n0|^2) / B; softplus(neg - pos) mean; + lambda * reg;
                              zero_grad, backward, torch.optim.Adam(lr) step
and the per-user evaluation loop of /root/reference/recad/workflow/normal.py:57-93 (model(users, items) = computer()
per user, pair scores of the unseen items, sort).
"""
import time

import numpy as np
import torch


def norm_adj_coo(n_users, n_items, ptr, idx):
    """D^-1/2 A D^-1/2 of the bipartite graph as a coalesced torch sparse COO tensor (implicit.py:259-277,320-326)."""
    U, I = n_users, n_items
    ptr, idx = np.asarray(ptr), np.asarray(idx)
    uu = np.repeat(np.arange(U, dtype=np.int64), np.diff(ptr))
    ii = idx.astype(np.int64) + U
    deg = np.zeros(U + I)
    deg[:U] = np.diff(ptr)
    np.add.at(deg, ii, 1.0)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, (deg + 1e-14) ** -0.5, 0.0).astype(np.float32)
    rows = np.concatenate([uu, ii])
    cols = np.concatenate([ii, uu])
    vals = (dinv[rows] * dinv[cols]).astype(np.float32)
    return torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (U + I, U + I)).coalesce()


class AtenLightGCN:
    def __init__(self, graph, user_emb, item_emb, n_layers, lam=1e-4, lr=1e-3):
        self.G = graph
        self.eu = torch.nn.Parameter(torch.as_tensor(user_emb, dtype=torch.float32).clone())
        self.ei = torch.nn.Parameter(torch.as_tensor(item_emb, dtype=torch.float32).clone())
        self.U, self.I = self.eu.shape[0], self.ei.shape[0]
        self.L, self.lam = n_layers, lam
        self.opt = torch.optim.Adam([self.eu, self.ei], lr=lr)

    def computer(self):
        all_emb = torch.cat([self.eu, self.ei])
        embs = [all_emb]
        for _ in range(self.L):
            all_emb = torch.sparse.mm(self.G, all_emb)
            embs.append(all_emb)
        light = torch.mean(torch.stack(embs, dim=1), dim=1)
        return torch.split(light, [self.U, self.I])

    def loss(self, u, p, n):
        lu, li = self.computer()
        ue, pe, ne = lu[u], li[p], li[n]
        u0, p0, n0 = self.eu[u], self.ei[p], self.ei[n]
        reg = 0.5 * (u0.norm(2).pow(2) + p0.norm(2).pow(2) + n0.norm(2).pow(2)) / float(len(u))
        pos, neg = (ue * pe).sum(1), (ue * ne).sum(1)
        return torch.mean(torch.nn.functional.softplus(neg - pos)) + self.lam * reg

    def step(self, u, p, n):
        u, p, n = (torch.as_tensor(np.asarray(t), dtype=torch.int64) for t in (u, p, n))
        loss = self.loss(u, p, n)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return float(loss.item())

    @torch.no_grad()
    def eval_user(self, u, seen_items, k=100):
        """normal.py:61-84 for one user: the scores of the unseen items through model(users, items), sorted."""
        mask = np.ones(self.I, dtype=bool)
        mask[np.asarray(seen_items)] = False
        items = torch.from_numpy(np.nonzero(mask)[0])
        lu, li = self.computer()
        scores = (lu[u].unsqueeze(0) * li[items]).sum(1)
        order = torch.argsort(scores, descending=True)[:k]
        return items[order].numpy(), scores[order].numpy()


def time_baseline(n_users, n_items, ptr, idx, train_ptr, train_idx, dim, layers, batch, triplets, budget_s=12.0,
                  eval_budget_s=6.0, thread_choices=(8, 16, 32, 64)):
    """bench.py's cpu_baseline: train steps/s and per-user evaluations/s of the sequence above with k host threads
    (ATen's sparse kernels do not scale to hundreds of threads: one step is timed at a few counts, the fastest is kept
    and reported)."""
    import os
    ncpu = os.cpu_count() or 1
    prev = torch.get_num_threads()
    try:
        G = norm_adj_coo(n_users, n_items, ptr, idx)
        g = torch.Generator().manual_seed(2023)
        m = AtenLightGCN(G, torch.randn(n_users, dim, generator=g) * 0.1, torch.randn(n_items, dim, generator=g) * 0.1, layers)
        users, pos, neg = (np.asarray(t) for t in triplets)
        avail = len(users) // batch
        best = None
        for k in sorted({min(ncpu, c) for c in thread_choices}):
            torch.set_num_threads(k)
            m.step(users[:batch], pos[:batch], neg[:batch])
            t0 = time.perf_counter()
            m.step(users[:batch], pos[:batch], neg[:batch])
            one = time.perf_counter() - t0
            if best is None or one < best[0]:
                best = (one, k)
            if one > 3.0:
                break
        one, k = best
        torch.set_num_threads(k)
        n = int(max(1, min(avail - 1, budget_s / max(one, 1e-3))))
        t0 = time.perf_counter()
        last = 0.0
        for s in range(1, n + 1):
            sl = slice(s * batch, (s + 1) * batch)
            last = m.step(users[sl], pos[sl], neg[sl])
        el = time.perf_counter() - t0
        train_ptr, train_idx = np.asarray(train_ptr), np.asarray(train_idx)
        n_eval = 0
        t1 = time.perf_counter()
        for u in range(0, n_users, max(1, n_users // 24)):
            m.eval_user(u, train_idx[train_ptr[u]:train_ptr[u + 1]])
            n_eval += 1
            if time.perf_counter() - t1 > eval_budget_s:
                break
        el_eval = time.perf_counter() - t1
        return {"steps": n, "seconds": el, "threads": k, "host_threads": ncpu, "last_loss": last,
                "interactions_per_s": n * batch / el, "eval_users": n_eval, "eval_seconds": el_eval,
                "eval_users_per_s": n_eval / el_eval}
    finally:
        torch.set_num_threads(prev)
