"""Row-sharded LightGCN training + evaluation over the GPUs of one node (north_star).

One process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests and in
the 2-ranks-on-one-GPU box check).  The N = U+I node rows are dealt round-robin to the W ranks (row r ->
rank r % W, local row q = r // W), which balances nonzeros of power-law graphs without a reordering pass.
Rank g owns rows R_g of E0 and of Adam's moments, and the CSR slab A[R_g, :] whose columns are re-labelled
to the *gathered layout* below.  The local rows are cut into C contiguous chunks (M = C * Mc rows per rank,
padded): chunk c of every rank is all-gathered as ONE contiguous block, so the gathered position of node r is

    pos(r) = c * (W * Mc) + (r % W) * Mc + o,      q = r // W,  c = q // Mc,  o = q % Mc.

Per train step:

    forward   l = 1..L : for c in chunks:  local SpMM on slab chunk c (fused layer sum)
                                           -> async all-gather of Y_l chunk c on the collective's own stream,
                                              overlapping the SpMM of chunk c+1            (l < L only)
              light rows of the minibatch: every rank contributes the rows it owns to a [3B, d] buffer,
              one small all-reduce (768 KB at B=1024, d=64) -- NOT an all-gather of the [N, d] light table
    BPR       replicated on every rank (B is tiny next to the graph): the gradients w.r.t. light / E0 are
              scattered into full-size, identical gprop / gego buffers on every rank
    backward  j = 1..L : chunked local SpMM (A symmetric: the same row slab serves the transpose), the
              all-gathers of t_j overlapped the same way; the last layer fuses Adam on the owned rows

=> 2L-1 all-gathers of (N/W)*d*4 bytes per rank and one 3B*d*4-byte all-reduce per step.  north_star's
"reduce-scatter on the item gradients" is what a batch-SPLIT BPR would need (each rank holding a partial
gradient over all rows); with the minibatch replicated every rank already holds the complete gradient rows,
so the reduce-scatter -- and L more collectives of the same volume per step -- drop out.  The price: every
rank runs the (B-sized) BPR kernel, keeps two full-size [N, d] gradient buffers and gathers from a
locally-held full copy of gprop in the first backward layer.

Evaluation (normal.py:57-160) is user-sharded: one propagation (L-1 all-gathers), ONE all-gather of the
light table from which the item block is taken, then every rank scores the users it owns against the whole
catalogue (fp32-MFMA GEMM + top-K, no further exchange) and the HR@K numerators are all-reduced (a few ints).

The compute calls go through an `ops` object: `HipOps` (the C-ABI) in the product; the CPU tests inject an
oracle-backed stand-in -- this module never imports it.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


# ------------------------------------------------------------------------------------------------ layout
class RowLayout:
    """Round-robin row partition with C chunks per rank (see the module docstring)."""

    def __init__(self, n_rows, world, chunks=1):
        self.N, self.W = int(n_rows), int(world)
        per = (self.N + self.W - 1) // self.W
        self.C = max(1, min(int(chunks), per))
        self.Mc = (per + self.C - 1) // self.C
        self.M = self.Mc * self.C

    def pos(self, r):
        """gathered position of node ids r (numpy array or torch tensor, any integer dtype)."""
        q = r // self.W
        return (q // self.Mc) * (self.W * self.Mc) + (r % self.W) * self.Mc + q % self.Mc

    def owner(self, r):
        return r % self.W

    def local(self, r):
        return r // self.W

    def own_rows(self, rank):
        return np.arange(rank, self.N, self.W)

    def chunk_range(self, rank, c):
        """(first, last+1) gathered positions of chunk c's rows owned by `rank`."""
        lo = c * (self.W * self.Mc) + rank * self.Mc
        return lo, lo + self.Mc


def build_slab_chunks(rowptr, col, val, rank, layout, tiled=False):
    """CSR slabs (one per chunk) of the rows owned by `rank`, columns in the gathered layout, entry order
    inside a row unchanged.  Inputs: host arrays or torch tensors (any device); the work is vectorised
    (repeat / cumsum / gather), on the inputs' device.  Returns a list of (rowptr int32[Mc+1], col int32,
    val float32) torch tensors -- or, tiled, a list of lists: entry [c][k] holds the entries of row chunk c
    whose columns lie in gathered chunk k (positions [k W Mc, (k+1) W Mc)), entry order inside a tile row
    unchanged, so that y_c = sum_k tile[c][k] . x can start on the column blocks that have arrived."""
    rp = torch.as_tensor(rowptr).long()
    cl = torch.as_tensor(col)
    vl = torch.as_tensor(val)
    dev = rp.device
    L = layout
    rows = torch.arange(rank, L.N, L.W, device=dev)
    deg_own = (rp[rows + 1] - rp[rows])
    deg = torch.zeros(L.M, dtype=torch.long, device=dev)
    deg[: rows.numel()] = deg_own
    lp = torch.zeros(L.M + 1, dtype=torch.long, device=dev)
    lp[1:] = torch.cumsum(deg, 0)
    total = int(lp[-1].item())
    # source position of every slab entry: start of its row + offset inside the row
    row_of = torch.repeat_interleave(torch.arange(L.M, device=dev), deg)
    starts = torch.zeros(L.M, dtype=torch.long, device=dev)
    starts[: rows.numel()] = rp[rows]
    src = starts[row_of] + (torch.arange(total, device=dev) - lp[row_of])
    c_new = L.pos(cl[src].long()).to(torch.int32)
    v_new = vl[src].to(torch.float32)
    out = []
    for c in range(L.C):
        a, b = int(lp[c * L.Mc].item()), int(lp[(c + 1) * L.Mc].item())
        if not tiled:
            out.append(((lp[c * L.Mc:(c + 1) * L.Mc + 1] - a).to(torch.int32).contiguous(), c_new[a:b].contiguous(), v_new[a:b].contiguous()))
            continue
        cc, vv = c_new[a:b], v_new[a:b]
        rows_c = (row_of[a:b] - c * L.Mc)
        blk = cc.long() // (L.W * L.Mc)                                  # source chunk of every entry
        order = torch.argsort(blk * L.Mc + rows_c, stable=True)          # by (block, row), entry order kept inside
        cnt = torch.bincount(blk * L.Mc + rows_c, minlength=L.C * L.Mc).view(L.C, L.Mc)
        cs, vs = cc[order], vv[order]
        start = 0
        tiles = []
        for k in range(L.C):
            rp_k = torch.zeros(L.Mc + 1, dtype=torch.long, device=dev)
            rp_k[1:] = torch.cumsum(cnt[k], 0)
            nk = int(rp_k[-1].item())
            tiles.append((rp_k.to(torch.int32).contiguous(), cs[start:start + nk].contiguous(), vs[start:start + nk].contiguous()))
            start += nk
        out.append(tiles)
    return out


# ------------------------------------------------------------------------------------------------ ops (product)
class HipOps:
    """The product implementation: librecad_hip.so through include/recad_hip.h."""

    name = "hip"

    def make_slab(self, rowptr, col, val, device):
        t = lambda a, dt: torch.as_tensor(a).to(device=device, dtype=dt).contiguous()
        return {"n_rows": len(rowptr) - 1, "rowptr": t(rowptr, torch.int32), "col": t(col, torch.int32),
                "val": t(val, torch.float32), "sched": {}, "coef": torch.zeros(2, device=device)}

    @staticmethod
    def _sched(slab, dim):
        if dim not in slab["sched"]:
            sched, n_blocks, n_words, s_words = C.c_void_p(), C.c_int32(0), C.c_int64(0), C.c_int64(0)
            _lib.check(_lib.lib().rk_csr_schedule_build(slab["n_rows"], _lib.ptr(slab["rowptr"]), 0, dim, _lib.stream_ptr(), C.byref(sched),
                                                        C.byref(n_blocks), C.byref(n_words), C.byref(s_words)), "rk_csr_schedule_build")
            try:
                desc = torch.zeros(int(n_words.value), device=slab["rowptr"].device, dtype=torch.int32)
                _lib.check(_lib.lib().rk_csr_schedule_upload(sched, _lib.ptr(desc), _lib.stream_ptr()), "rk_csr_schedule_upload")
            finally:
                _lib.lib().rk_csr_schedule_destroy(sched)
            scratch = torch.zeros(int(s_words.value), device=slab["rowptr"].device, dtype=torch.int32) if s_words.value else None
            slab["sched"][dim] = (desc, int(n_blocks.value), scratch)   # the slab is this trainer's own: one stream
        return slab["sched"][dim]

    def spmm(self, slab, x, add=None, y=None, sum_in=None, sum_out=None, sum_scale=1.0, adam=None, src_filter=None):
        if slab["col"].numel() == 0 and slab["n_rows"] == 0:
            return
        # A step repeats the same launches on the same persistent buffers: the marshalled argument list of every distinct
        # (slab, operands) combination is built once and replayed (the step is host-bound on small graphs: 2 L C^2 launches)
        dp = lambda t: 0 if t is None else t.data_ptr()
        key = (id(slab), dp(x), dp(add), dp(y), dp(sum_in), dp(sum_out), float(sum_scale), dp(src_filter),
               None if adam is None else (dp(adam["p"]), dp(adam["m"]), dp(adam["v"]), dp(adam.get("coef"))))
        memo = self.__dict__.setdefault("_calls", {})
        call = memo.get(key)
        if call is None:
            if slab["col"].numel() == 0:
                # a chunk / tile without nonzeros (n_rows > 0): the launch needs col / val pointers, so give it one unused zero entry
                slab = dict(slab, col=torch.zeros(1, dtype=torch.int32, device=x.device), val=torch.zeros(1, dtype=torch.float32, device=x.device))
            e = _lib.SpmmEpilogue(add=_lib.ptr(add), y=_lib.ptr(y), sum_in=_lib.ptr(sum_in), sum_out=_lib.ptr(sum_out),
                                  sum_scale=float(sum_scale), src_filter=_lib.ptr(src_filter))
            if adam is not None:
                e.adam_p, e.adam_m, e.adam_v = _lib.ptr(adam["p"]), _lib.ptr(adam["m"]), _lib.ptr(adam["v"])
                e.coef_scratch = _lib.ptr(adam.get("coef", slab["coef"]))
                e.lr, e.beta1, e.beta2, e.eps = adam["lr"], adam["b1"], adam["b2"], adam["eps"]
            desc, n_blocks, scratch = self._sched(slab, x.shape[1])
            args = (slab["n_rows"], _lib.ptr(slab["rowptr"]), _lib.ptr(slab["col"]), _lib.ptr(slab["val"]), _lib.ptr(desc), n_blocks,
                    _lib.ptr(scratch), x.shape[1], _lib.ptr(x), x.shape[0], C.byref(e))
            call = memo[key] = (args, e, (slab, desc, scratch, x, add, y, sum_in, sum_out, dict(adam) if adam else None, src_filter))   # (keeps the tensors alive)
        args, e, _ = call
        if adam is not None:
            e.adam_t = adam["t"]
            e.lr, e.beta1, e.beta2, e.eps = adam["lr"], adam["b1"], adam["b2"], adam["eps"]
        _lib.check(_lib.lib().rk_spmm_csr_ex(*args, _lib.stream_ptr()), "rk_spmm_csr_ex")

    def gather_rows(self, src, idx, mask, out):
        """out[i] = mask[i] * src[idx[i]] (mask None = 1): the minibatch's light rows this rank owns, zeros elsewhere."""
        _lib.check(_lib.lib().rk_rows_gather_masked(src.shape[1], _lib.ptr(src), _lib.ptr(idx), _lib.ptr(mask), idx.numel(), _lib.ptr(out),
                                                    _lib.stream_ptr()), "rk_rows_gather_masked")

    def zero_rows(self, a, b, idx):
        _lib.check(_lib.lib().rk_rows_zero(a.shape[1], _lib.ptr(a), _lib.ptr(b), _lib.ptr(idx), idx.numel(), _lib.stream_ptr()), "rk_rows_zero")

    def adam_advance(self, coef, counter, lr, b1, b2):
        """counter += 1 and this step's Adam coefficients into coef, on the device (a captured step cannot take t from the host)"""
        _lib.check(_lib.lib().rk_adam_coef_advance(_lib.ptr(coef), _lib.ptr(counter), float(lr), float(b1), float(b2), _lib.stream_ptr()),
                   "rk_adam_coef_advance")

    def adam(self, p, g, m, v, t, lr, b1, b2, eps, coef=None):
        """dense torch.optim.Adam step t (1-based) on a contiguous block (the 2-D trainer's owned rows: its gradient only exists
        after a reduce-scatter, so it cannot ride in an SpMM epilogue); coef: device float[2] holding the step's coefficients
        (adam_advance) instead of t -- what a captured step uses"""
        if coef is not None:
            _lib.check(_lib.lib().rk_adam_step_dev(p.numel(), _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(coef), float(b1),
                                                   float(b2), float(eps), _lib.stream_ptr()), "rk_adam_step_dev")
            return
        _lib.check(_lib.lib().rk_adam_step(p.numel(), _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), int(t), float(lr), float(b1),
                                           float(b2), float(eps), _lib.stream_ptr()), "rk_adam_step")

    def new_row_bits(self, n_rows, device):
        return torch.zeros((n_rows + 31) // 32, dtype=torch.int32, device=device)

    def mark_rows(self, bits, idx, on):
        """frontier bitmap of the first backward layer: set the bits of the minibatch's gathered rows / clear their words"""
        _lib.check(_lib.lib().rk_rows_mark_bits(_lib.ptr(bits), _lib.ptr(idx), idx.numel(), 1 if on else 0, _lib.stream_ptr()), "rk_rows_mark_bits")

    def bpr(self, dim, n_layers, lam, light_rows, emb, gprop, gego, ru, rp, rn, loss_partials, keys=None):
        """light_rows: compact [3*nb, d] (users, positives, negatives of the minibatch, in that order);
        emb / gprop / gego are indexed by the gathered positions ru / rp / rn.  keys (int64[3*nb], the batch's sorted
        incidence plan): the ordered scatter instead of float atomics."""
        if keys is not None:
            _lib.check(_lib.lib().rk_bpr_rows_ordered(dim, n_layers, float(lam), _lib.ptr(light_rows), _lib.ptr(emb), _lib.ptr(gprop),
                                                      _lib.ptr(gego), _lib.ptr(ru), _lib.ptr(rp), _lib.ptr(rn), ru.numel(), _lib.ptr(keys),
                                                      _lib.ptr(loss_partials), _lib.stream_ptr()), "rk_bpr_rows_ordered")
            return
        _lib.check(_lib.lib().rk_bpr_rows(dim, n_layers, float(lam), _lib.ptr(light_rows), 1, _lib.ptr(emb), _lib.ptr(gprop),
                                          _lib.ptr(gego), _lib.ptr(ru), _lib.ptr(rp), _lib.ptr(rn), ru.numel(),
                                          _lib.ptr(loss_partials), _lib.stream_ptr()), "rk_bpr_rows")

    def score_topk(self, utab, user_rows, itab, seen_ptr, seen_idx, targets, K):
        """-> (target_score [n,T] float32, target_rank [n,T] int32, top_ids [n,K] int32) for the users whose
        rows in utab are user_rows; seen_ptr/seen_idx are indexed by those same row numbers."""
        from .evaluate import full_catalog_topk

        class _Tables:
            def scoring_tables(self_inner):
                return utab, itab, None, None, 0.0

        chunk = max(256, min(8192, (1 << 31) // max(itab.shape[0], 1)))
        res = full_catalog_topk(_Tables(), user_rows, seen_ptr, seen_idx, targets, K=K, chunk=chunk, to_host=False)
        return res["target_score"], res["target_rank"], res["top_ids"]


# ------------------------------------------------------------------------------------------------ trainer
class ShardedLightGCN:
    """Trains / evaluates a LightGCN victim's tables with the node rows sharded over the process group.

    csr: (rowptr, col, val) host arrays / tensors, or a recad_amd.graph.CsrGraph (device tensors are used
    in place: the slab is cut on the device)."""

    def __init__(self, n_users, n_items, dim, n_layers, csr, user_emb, item_emb, lam=1e-4, lr=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, group=None, ops=None, device=None, chunks=None, gather="collective", force_collectives=False,
                 deterministic=False, overlap=None, capture=None):
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.ops = ops or HipOps()
        self.U, self.I, self.d, self.L = int(n_users), int(n_items), int(dim), int(n_layers)
        if self.L < 1:
            raise ValueError("ShardedLightGCN needs n_layers >= 1")
        self.lam, self.lr, self.betas, self.eps = lam, lr, betas, eps
        self.N = self.U + self.I
        if chunks is None:
            # Overlap only pays when there is a collective to hide and the slab is big.  The gather of the last chunk of a
            # layer is exposed (about 1/C of the layer's communication) and the local SpMM pays for being cut: measured on one
            # GPU for rank 0 of W = 8 at config 4 (scripts/shard_probe.py): 1.00 / 1.12 / 1.26 ms per layer at C = 2 / 4 / 8,
            # against >= 0.62 ms to receive the other ranks' 336 MB over 7 xGMI links => C = 4 for slabs of >= 128 K rows.
            per_rank = self.N // self.world
            collective = self.world > 1 or force_collectives
            chunks = 4 if (collective and per_rank >= (1 << 17)) else 2 if (collective and per_rank >= 4096) else 1
        self.layout = RowLayout(self.N, self.world, chunks)
        self.gather_mode = gather
        # W == 1 normally short-circuits every collective to a copy; force_collectives keeps them (a one-rank RCCL
        # group on a single-GPU box exercises the real collective calls, their async handles and stream ordering)
        self.force_collectives = bool(force_collectives) and on
        self.deterministic = bool(deterministic)   # ordered gradient scatter (rk_bpr_rows_ordered): bit-identical on every rank and run
        self.device = torch.device(device) if device is not None else user_emb.device
        if hasattr(csr, "rowptr"):
            rowptr, col, val = csr.rowptr, csr.col, csr.val
        else:
            rowptr, col, val = csr
        # Consumer-side overlap (SURVEY 8e "SpMM on already-arrived column blocks"): every row chunk's slab is tiled by source
        # chunk and a layer runs source chunk by source chunk, chaining the partial sums through the SpMM's `add` epilogue, so
        # layer l+1 starts on the column blocks whose all-gather has completed while the last chunk of layer l is still in
        # flight.  Off for one chunk, and in deterministic mode (the tiled sum is not in CSR order, so the tables would stop
        # being bit-identical across world sizes).
        if overlap is None:
            overlap = "consumer" if (self.layout.C > 1 and not deterministic) else "producer"
        if overlap not in ("consumer", "producer"):
            raise ValueError("overlap must be 'consumer' or 'producer'")
        self.tiled = overlap == "consumer" and self.layout.C > 1
        built = build_slab_chunks(rowptr, col, val, self.rank, self.layout, tiled=self.tiled)
        if self.tiled:
            self.tiles = [[self.ops.make_slab(rp, c, v, self.device) for rp, c, v in row] for row in built]
            self.slabs = None
        else:
            self.slabs = [self.ops.make_slab(rp, c, v, self.device) for rp, c, v in built]
            self.tiles = [[sl] for sl in self.slabs]
        dev, M, W, d = self.device, self.layout.M, self.world, self.d
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.e0, self.m, self.v = z(M, d), z(M, d), z(M, d)
        self.xfull = [z(W * M, d) for _ in range(2)]      # gathered X_l / t_j (ping-pong)
        self.e0_full = z(W * M, d)
        self.y, self.s = z(M, d), z(M, d)
        self.ybuf = [self.y, z(M, d)] if self.tiled else [self.y, self.y]   # tiled: a layer's output is still being gathered while the next one accumulates
        self._ysel = 0
        self.gprop, self.gego = z(W * M, d), z(W * M, d)  # replicated, zero outside a step
        self.t = 0
        self._plan = None
        # Step capture: one full-batch train step -- its ~2 L C^2 + 6 kernel launches AND its collectives -- recorded once into a
        # graph (torch.cuda.CUDAGraph on the launch stream: the C-ABI launches go to torch's current stream, so they are
        # captured with the RCCL calls) and replayed per step; the step's indices are copied into fixed staging buffers first
        # and Adam's step number lives on the device (rk_adam_coef_advance).  None = try it wherever it can work (HIP ops,
        # device tensors, RCCL or no collective at all) and fall back to the eager step if the capture fails.
        self.capture = capture
        self._graph = None
        self._graph_failed = False
        # frontier bitmap over the gathered rows (first backward layer); ops without it (the CPU stand-in) gather everything
        self.row_bits = self.ops.new_row_bits(W * M, dev) if hasattr(self.ops, "new_row_bits") else None
        self.load_tables(user_emb, item_emb)

    def describe(self):
        L = self.layout
        return (f"node rows sharded round-robin over {self.world} GPUs ({L.M} rows/rank in {L.C} chunk(s)); per step "
                f"{2 * self.L - 1} all-gathers of {L.M * self.d * 4 / 1e6:.1f} MB/rank ({self.gather_mode}) overlapped chunk-wise "
                f"with the local SpMM ({'consumer-side: tiles by source chunk' if self.tiled else 'producer-side'}) + one [3B,d] "
                f"all-reduce; BPR replicated, no reduce-scatter")

    # ------------------------------------------------------------------ table movement
    def load_tables(self, user_emb, item_emb):
        full = torch.cat([user_emb.detach().to(self.device), item_emb.detach().to(self.device)])
        own = torch.arange(self.rank, self.N, self.world, device=self.device)
        self.e0.zero_()
        self.e0[: own.numel()] = full[own]

    def _host_staged(self):
        """gloo with device tensors (the 2-ranks-on-one-GPU box check): collectives go through host copies."""
        return self.world > 1 and self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"

    def _all_reduce(self, t):
        if self.world == 1 and not self.force_collectives:
            return
        if self._host_staged():
            h = t.cpu()
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    def _all_gather_chunk(self, local_chunk, out_full, c, async_op):
        """all-gather chunk c (rows [c*Mc, (c+1)*Mc) of a local [M,d] buffer) into its block of out_full."""
        L, W = self.layout, self.world
        blk = out_full[c * W * L.Mc:(c + 1) * W * L.Mc]
        if W == 1 and not self.force_collectives:
            blk.copy_(local_chunk)
            return None
        if self._host_staged():
            h = torch.empty(blk.shape, dtype=blk.dtype)
            dist.all_gather_into_tensor(h, local_chunk.cpu().contiguous(), group=self.group)
            blk.copy_(h)
            return None
        if self.gather_mode == "direct" and W > 1:
            # one-shot: every rank sends its shard to each peer and receives theirs, all pairs at once
            # (one RCCL group of 2(W-1) point-to-point ops: each rides its own xGMI link)
            blk[self.rank * L.Mc:(self.rank + 1) * L.Mc].copy_(local_chunk)
            ops_ = []
            for k in range(1, W):
                dst, src = (self.rank + k) % W, (self.rank - k) % W
                ops_.append(dist.P2POp(dist.isend, local_chunk, dst, self.group))
                ops_.append(dist.P2POp(dist.irecv, blk[src * L.Mc:(src + 1) * L.Mc], src, self.group))
            works = dist.batch_isend_irecv(ops_)
            if not async_op:
                for w_ in works:
                    w_.wait()
                return None
            return works
        work = dist.all_gather_into_tensor(blk, local_chunk, group=self.group, async_op=async_op)
        return [work] if async_op else None

    @staticmethod
    def _wait(pending):
        for works in pending:
            if works:
                for w_ in works:
                    w_.wait()   # nccl: the current stream waits for the collective's stream; gloo: host wait
        pending.clear()

    def _gather_full(self, local, out_full):
        for c in range(self.layout.C):
            self._all_gather_chunk(local[c * self.layout.Mc:(c + 1) * self.layout.Mc], out_full, c, False)
        return out_full

    def tables(self):
        """(users[U,d], items[I,d]) in the original order, on every rank."""
        g = self._gather_full(self.e0, torch.empty_like(self.e0_full))
        full = g[self.layout.pos(torch.arange(self.N, device=self.device))]
        return full[: self.U].contiguous(), full[self.U:].contiguous()

    # ------------------------------------------------------------------ propagation (shared by train / eval)
    def _layer(self, x, ready, first_add, final_kw, gather_into, src_filter=None):
        """One propagation layer on the owned rows: y_c = sum_k tile[c][k] . x (+ first_add_c), the partial sums chained through
        the SpMM's `add` epilogue in self.y.  Source chunk k is the OUTER loop: its tiles only need gathered chunk k of x
        (`ready[k]`: the pending all-gather works of that chunk, or None), so they run while the later chunks are still in
        flight; row chunk c is complete after its last tile, and -- gather_into given -- its own all-gather starts at once.
        first_add(c) -> addend of row chunk c or None; final_kw(c) -> epilogue of its last tile (sum_in/sum_out/sum_scale/adam;
        `y` is filled in here).  Returns the pending works of the produced chunks (or None)."""
        ops, lay = self.ops, self.layout
        K = len(self.tiles[0])
        pending = [None] * lay.C
        self._ysel ^= 1
        ybuf = self.ybuf[self._ysel]   # (not the buffer whose chunks the previous layer's all-gathers may still be reading)
        for k in range(K):
            if ready is not None and ready[k]:
                self._wait([ready[k]])   # gathered chunk k of x has landed (the later chunks may still be in flight)
            for c in range(lay.C):
                rs = slice(c * lay.Mc, (c + 1) * lay.Mc)
                first, final = k == 0, k == K - 1
                add = first_add(c) if first else ybuf[rs]
                kw = dict(final_kw(c)) if final else {}
                want_y = kw.pop("want_y", True)
                if src_filter is not None:
                    kw["src_filter"] = src_filter   # (an argument, not instance state: an exception mid-step cannot leave it set)
                ops.spmm(self.tiles[c][k], x, add=add, y=ybuf[rs] if (not final or want_y) else None, **kw)
                if final and gather_into is not None:
                    pending[c] = self._all_gather_chunk(ybuf[rs], gather_into, c, True)
        return pending if gather_into is not None else None

    def _ready_all(self, ready):
        """Wait for every chunk (the untiled layer needs all of x)."""
        if ready is None:
            return None
        self._wait([w for w in ready if w])
        return None

    def _forward(self):
        """L propagation layers with the all-gathers of layers < L in flight under the local SpMMs (tiled: also under the
        NEXT layer's tiles of the chunks that have arrived); leaves light rows (owned) in self.s."""
        L, lay = self.L, self.layout
        inv = 1.0 / (L + 1)
        x = self._gather_full(self.e0, self.e0_full)
        ready = None
        for l in range(1, L + 1):
            last = l == L
            nxt = self.xfull[l & 1]
            if not self.tiled:
                ready = self._ready_all(ready)

            def final_kw(c, l=l, last=last):
                rs = slice(c * lay.Mc, (c + 1) * lay.Mc)
                return {"sum_in": (self.e0 if l == 1 else self.s)[rs], "sum_out": self.s[rs], "sum_scale": inv if last else 1.0,
                        "want_y": not last}
            ready = self._layer(x, ready, lambda c: None, final_kw, None if last else nxt)
            x = nxt

    # ------------------------------------------------------------------ training
    def reserve(self, n_triplets, batch):
        """Preallocate the per-epoch index plan and loss buffers for epochs of up to n_triplets."""
        n_steps = (int(n_triplets) + batch - 1) // batch
        dev = self.device
        if self._plan is None or self._plan["cap_steps"] < n_steps or self._plan["batch"] != batch:
            # `rows` (the minibatch's compact light rows) is baked into the captured step graph: it is allocated once per
            # batch size and survives a growing epoch; only the per-step loss rows (copied out after a replay, never captured)
            # are re-sized.  A new batch size drops the captured graph with the buffer it was recorded on.
            rows = None if (self._plan is None or self._plan["batch"] != batch) else self._plan["rows"]
            if rows is None:
                rows = torch.zeros(3 * batch, self.d, device=dev, dtype=torch.float32)
                self._graph = None
            self._plan = {"cap_steps": n_steps, "batch": batch,
                          "loss": torch.zeros(n_steps, _lib.RK_LOSS_PARTIALS, device=dev, dtype=torch.float32),
                          "rows": rows}
        return self._plan

    def _bpr_pos(self, nodes):
        """row of a node in the replicated gradient buffers the BPR kernel scatters into"""
        return self.layout.pos(nodes)

    def _epoch_plan(self, users, pos, neg, batch):
        """Everything index-shaped a step needs, for the whole epoch at once (no per-step host arithmetic):
        gathered positions of the minibatch nodes, and which compact rows this rank owns."""
        lay, dev = self.layout, self.device
        nodes = torch.stack([users.to(dev).long(), pos.to(dev).long() + self.U, neg.to(dev).long() + self.U])  # [3, n]
        posn = self._bpr_pos(nodes).contiguous()                                                  # gathered positions
        own_f = ((nodes % self.world) == self.rank).to(torch.float32)
        local = (nodes // self.world)
        ep = {"pos": posn, "own_f": own_f, "local": local}
        # per step: the 3*nb (role-major) indices as ONE contiguous row each, for the row kernels
        n_all = posn.shape[1]
        n_st = (n_all + batch - 1) // batch

        def per_step(t, fill):
            pad = torch.full((3, n_st * batch), fill, dtype=t.dtype, device=dev)
            pad[:, :n_all] = t
            full = pad.view(3, n_st, batch).permute(1, 0, 2).contiguous()           # [step, role, b]
            out = full.reshape(n_st, 3 * batch).clone()
            last = n_all - (n_st - 1) * batch
            if last < batch:   # ragged last step: its 3*nb entries must be contiguous at the front
                out[n_st - 1, : 3 * last] = full[n_st - 1, :, :last].reshape(-1)
            return out
        ep["local3"], ep["pos3"], ep["own3"] = per_step(local, 0), per_step(posn, 0), per_step(own_f, 0.0)
        if self.deterministic:
            # ordered scatter: every step's 3*nb incidences (gathered row << 20 | 3*b + role) sorted once per epoch, the ragged
            # last step padded with keys that sort behind everything (the kernel reads the first 3*nb of a step)
            n = posn.shape[1]
            n_steps = (n + batch - 1) // batch
            if 3 * batch >= (1 << 20):
                raise ValueError("deterministic scatter: batches of < 349525 triplets")
            if self.world * lay.M >= (1 << 24):
                # the kernel (rk_bpr_rows_ordered, plan_row) keeps 24 bits of the gathered row in a key
                raise ValueError("deterministic scatter: fewer than 2^24 gathered rows (world * rows per rank)")
            b_in = torch.arange(n, device=dev) % batch
            keys = (posn << 20) | (3 * b_in.unsqueeze(0) + torch.arange(3, device=dev).unsqueeze(1))          # [3, n]
            padded = torch.full((3, n_steps * batch), torch.iinfo(torch.int64).max, dtype=torch.int64, device=dev)
            padded[:, :n] = keys
            per_step = padded.view(3, n_steps, batch).permute(1, 0, 2).reshape(n_steps, 3 * batch)
            ep["keys"] = torch.sort(per_step, dim=1).values.contiguous()
        return ep

    def _step_core(self, nb, idx_local3, idx_own3, idx_pos3, lp, keys, adam):
        """One train step given the minibatch's 3*nb role-major index rows (gathered positions idx_pos3, local rows idx_local3,
        ownership mask idx_own3) -- slices of the epoch plan (eager) or the fixed staging buffers (captured)."""
        ops, L, lay, r = self.ops, self.L, self.layout, self.rank
        self._forward()
        # light rows of the minibatch: own rows in place, zeros elsewhere, summed over the ranks (x + 0 is exact)
        rows = self._plan["rows"][: 3 * nb]
        collective = self.world > 1 or self.force_collectives
        # r // W is a valid local row for every node; rows this rank does not own come out as exact zeros
        ops.gather_rows(self.s, idx_local3, idx_own3 if collective else None, rows)
        if collective:
            self._all_reduce(rows)
        ru, rp, rn = idx_pos3[:nb], idx_pos3[nb:2 * nb], idx_pos3[2 * nb:3 * nb]
        ops.bpr(self.d, L, self.lam, rows, self.e0_full, self.gprop, self.gego, ru, rp, rn, lp, keys=keys)
        # backward + Adam on the owned rows
        x = self.gprop
        ready = None
        frontier = getattr(self, "row_bits", None)
        if frontier is not None:
            ops.mark_rows(frontier, idx_pos3, True)   # gprop is zero outside these rows: the first layer skips the rest
        for j in range(1, L + 1):
            last = j == L
            nxt = self.xfull[j & 1]
            if not self.tiled:
                ready = self._ready_all(ready)

            def first_add(c, last=last):
                lo, hi = lay.chunk_range(r, c)
                return (self.gego if last else self.gprop)[lo:hi]

            def final_kw(c, last=last):
                rs = slice(c * lay.Mc, (c + 1) * lay.Mc)
                return {"adam": dict(adam, p=self.e0[rs], m=self.m[rs], v=self.v[rs]), "want_y": False} if last else {}
            ready = self._layer(x, ready, first_add, final_kw, None if last else nxt, src_filter=frontier if j == 1 else None)
            x = nxt
        # only the minibatch's rows of the replicated gradient buffers are non-zero
        ops.zero_rows(self.gprop, self.gego, idx_pos3)
        if frontier is not None:
            ops.mark_rows(frontier, idx_pos3, False)

    def step(self, plan, ep, s0, nb, k):
        """One train step on triplets [s0, s0+nb) of the epoch plan; writes its loss partials to plan['loss'][k]."""
        lp = plan["loss"][k]
        self.t += 1
        if nb == plan["batch"] and self._capture_ok():
            if self._replay(plan, ep, k, lp):
                return lp
        adam = {"t": self.t, "lr": self.lr, "b1": self.betas[0], "b2": self.betas[1], "eps": self.eps}
        self._step_core(nb, ep["local3"][k][: 3 * nb], ep["own3"][k][: 3 * nb], ep["pos3"][k][: 3 * nb], lp,
                        ep["keys"][k] if self.deterministic else None, adam)
        return lp

    # ------------------------------------------------------------------ captured step
    def _capture_ok(self):
        if self.capture is False or self._graph_failed or self.device.type != "cuda" or not hasattr(self.ops, "adam_advance"):
            return False
        if self._host_staged():
            return False
        return True

    def _replay(self, plan, ep, k, lp):
        """Replay (after capturing it on first use) the full-batch step graph for step k of the epoch; False = not available."""
        B = plan["batch"]
        if self._graph is None or self._graph["batch"] != B:
            if self.t <= 1:
                return False      # the first step of a trainer runs eagerly: it builds the schedules and the memoised launches
            st = {"local3": torch.zeros(3 * B, dtype=torch.int64, device=self.device),
                  "pos3": torch.zeros(3 * B, dtype=torch.int64, device=self.device),
                  "own3": torch.zeros(3 * B, dtype=torch.float32, device=self.device),
                  "keys": torch.zeros(3 * B, dtype=torch.int64, device=self.device) if self.deterministic else None,
                  "loss": torch.zeros(_lib.RK_LOSS_PARTIALS, dtype=torch.float32, device=self.device),
                  "coef": torch.zeros(2, dtype=torch.float32, device=self.device),
                  "t_dev": torch.zeros(1, dtype=torch.int32, device=self.device), "t_host": -1, "batch": B}
            adam = {"t": -1, "coef": st["coef"], "lr": self.lr, "b1": self.betas[0], "b2": self.betas[1], "eps": self.eps}
            ysel = self._ysel
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    self.ops.adam_advance(st["coef"], st["t_dev"], self.lr, self.betas[0], self.betas[1])
                    self._step_core(B, st["local3"], st["own3"], st["pos3"], st["loss"], st["keys"], adam)
            except Exception as exc:    # RCCL / runtime without capture support: keep the eager step
                self._graph_failed = True
                self._ysel = ysel
                import warnings
                warnings.warn(f"ShardedLightGCN: step capture unavailable ({type(exc).__name__}: {exc}); running eagerly")
                return False
            if (self._ysel ^ ysel) & 1:
                raise RuntimeError("internal: a step must flip the partial-sum buffer an even number of times")
            st["graph"] = g
            self._graph = st
        st = self._graph
        st["local3"].copy_(ep["local3"][k], non_blocking=True)
        st["pos3"].copy_(ep["pos3"][k], non_blocking=True)
        st["own3"].copy_(ep["own3"][k], non_blocking=True)
        if self.deterministic:
            st["keys"].copy_(ep["keys"][k], non_blocking=True)
        if st["t_host"] != self.t - 1:
            st["t_dev"].fill_(self.t - 1)     # eager steps ran in between: the device counter follows the host's
        st["graph"].replay()
        st["t_host"] = self.t
        lp.copy_(st["loss"], non_blocking=True)
        return True

    def train_epoch(self, users, pos, neg, batch):
        """All ranks pass the SAME triplets.  Returns the per-step losses (float64 tensor, host)."""
        n = users.numel()
        plan = self.reserve(n, batch)
        ep = self._epoch_plan(users, pos, neg, batch)
        k = 0
        for s0 in range(0, n, batch):
            self.step(plan, ep, s0, min(batch, n - s0), k)
            k += 1
        return plan["loss"][:k].sum(dim=1).double().cpu()

    # ------------------------------------------------------------------ evaluation (user-sharded)
    def evaluate(self, seen_ptr, seen_idx, targets, K=100, topks=(10, 20, 50, 100), reps=1):
        """Full-catalog scoring + top-K + HR@K (normal.py:57-160) with the users sharded like the rows:
        one propagation, one all-gather of the light table, local scoring of the owned users, all-reduced
        hit counts.  seen_ptr/seen_idx: host CSR user -> sorted train items; users whose list is empty or
        holds a target are skipped (normal.py:133-143).  Returns a dict (identical on every rank)."""
        import time

        lay, dev, W, r = self.layout, self.device, self.world, self.rank
        seen_ptr = np.asarray(seen_ptr).astype(np.int64)
        seen_idx = np.asarray(seen_idx).astype(np.int32)
        targets = np.asarray(targets, dtype=np.int32)
        mine = np.arange(r, self.U, W)                                  # global user ids owned by this rank
        deg = seen_ptr[mine + 1] - seen_ptr[mine]
        # local seen CSR indexed by the LOCAL row number q = u // W (utab is this rank's light rows)
        lptr = np.zeros(lay.M + 1, dtype=np.int64)
        lptr[1:len(mine) + 1] = np.cumsum(deg)
        lptr[len(mine) + 1:] = lptr[len(mine)]
        take = np.repeat(seen_ptr[mine], deg) + (np.arange(int(deg.sum())) - np.repeat(lptr[:len(mine)], deg))
        lidx = seen_idx[take]
        if len(lidx) > 1:   # ids ascending inside a user (the fused scoring sweep walks the lists with a cursor)
            key = np.repeat(np.arange(len(mine), dtype=np.int64), deg) * (int(lidx.max()) + 1) + lidx
            if not bool(np.all(key[1:] >= key[:-1])):
                lidx = lidx[np.argsort(key, kind="stable")]
        has_t = np.zeros(len(mine), dtype=bool)
        if len(lidx):
            hit = np.isin(lidx, targets)
            has_t[np.repeat(np.arange(len(mine)), deg)[hit]] = True
        q_elig = np.nonzero((deg > 0) & ~has_t)[0].astype(np.int32)
        t_ = lambda a, dt: torch.as_tensor(a).to(device=dev, dtype=dt).contiguous()
        q_dev, lptr_dev, lidx_dev, tg_dev = t_(q_elig, torch.int32), t_(lptr, torch.int32), t_(lidx if len(lidx) else np.zeros(1, np.int32), torch.int32), t_(targets, torch.int32)
        ks = torch.as_tensor(list(topks), device=dev, dtype=torch.int32)
        item_pos = lay.pos(torch.arange(self.U, self.N, device=dev))
        light_full = torch.empty_like(self.e0_full)
        out = None
        if dev.type == "cuda":
            torch.cuda.synchronize()
        if W > 1:
            dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(max(1, reps)):
            self._forward()
            self._gather_full(self.s, light_full)
            itab = light_full.index_select(0, item_pos)
            stats = torch.zeros(1 + len(targets) * (1 + len(topks)), device=dev, dtype=torch.float64)
            if len(q_elig):
                tscore, trank, _ = self.ops.score_topk(self.s, q_dev, itab, lptr_dev, lidx_dev, tg_dev, K)
                hits = (trank.unsqueeze(2) < ks.view(1, 1, -1)).sum(dim=0).double()        # [T, nk]
                stats[0] = float(len(q_elig))
                stats[1:1 + len(targets)] = tscore.double().sum(dim=0)
                stats[1 + len(targets):] = hits.reshape(-1)
            self._all_reduce(stats)
            out = stats
        host = out.cpu().numpy()
        el = (time.perf_counter() - t0) / max(1, reps)
        n_users = int(host[0])
        T = len(targets)
        hits = host[1 + T:].reshape(T, len(topks))
        res = {"value": n_users / el, "unit": "users/s", "eligible_users": n_users, "seconds": el, "evaluations_timed": max(1, reps),
               "target_score_mean": [float(x) / max(n_users, 1) for x in host[1:1 + T]],
               "includes": "sharded propagate + light all-gather + per-rank fp32-MFMA GEMM / top-K over the owned users + all-reduced HR counts"}
        for qi, k in enumerate(topks):
            res[f"hr@{k}"] = float(hits[0, qi]) / max(n_users, 1)
        res["hit_counts"] = hits.tolist()
        return res
