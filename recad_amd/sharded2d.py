"""2-D (Pr x Pc) tiling of the row-sharded LightGCN trainer (north_star: "all-gather ... before scoring and reduce-scatter on
the item gradients"): `--parallel rows2d`.

The 1-D partition (recad_amd/sharded.py) makes every rank receive the whole gathered table per layer -- (W-1)/W of N*d*4 bytes:
336 MB at config 4 with W = 8 -- and walk all of it with the gathers of its slab.  Here the W = Pr * Pc ranks form a grid; node
row r belongs to block b = r % W (round-robin: balances a power-law graph), block b sits at grid position (i, j) = (b // Pc,
b % Pc), and rank (i, j)

  * owns block (i, j) of every vector (E0, Adam's moments, the layer outputs): N / W rows;
  * holds the TILE A[R_i, C_j] of the normalised adjacency: rows R_i = the Pc blocks of grid row i, columns C_j = the Pr blocks
    of grid column j (A is symmetric, so the same tile serves the backward pass).

One propagation layer  y = A x :
    all-gather of x within the COLUMN group (the Pr ranks (., j)): every member then holds x[C_j]   -- receives (Pr-1) blocks
    local SpMM  partial = A[R_i, C_j] . x[C_j]                                                        -- N/Pr rows, nnz/W entries
    reduce-scatter of the partials within the ROW group (the Pc ranks (i, .)): sum over j, rank (i, j) keeps block (i, j)
                                                                                                      -- receives (Pc-1) blocks
=> (Pr - 1 + Pc - 1) blocks of (N/W)*d*4 bytes received per rank and layer: 2 x 4 at config 4 = 4 x 48 MB = 192 MB instead of 336,
and the tile's gathers touch a column part of N/Pc rows instead of the whole table.  The price: a second collective on every
layer's critical path (none for Pr = 1, the default: see grid_shape), and a summation order that depends on the grid
(`reduce="ordered"` fixes it: an all-to-all of the partial blocks and a sum in group-rank order -- same bytes as a direct
reduce-scatter).  The blocks' rows are cut into C chunks; the reduce-scatter of chunk c is issued asynchronously and runs under
the tile SpMM of chunk c+1 (producer-side overlap), so a layer exposes about 1/C of its communication.

BPR runs replicated like in the 1-D trainer (B is tiny next to the graph): the minibatch's light rows AND ego rows travel in ONE
[6B, d] all-reduce (every rank contributes the rows it owns, exact zeros elsewhere), the gradient rows are scattered into
replicated, grid-column-major gprop / gego buffers, so the first backward layer's x[C_j] is a local view (no gather).

Per step: L all-gathers + L reduce-scatters (forward), L reduce-scatters + (L-1) all-gathers (backward), one [6B, d] all-reduce.
Evaluation is user-sharded exactly like the 1-D trainer's (inherited), and so is the step capture (a full-batch step with its
collectives recorded once, replayed per step; eager fallback)."""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .sharded import HipOps, ShardedLightGCN


def grid_shape(world, pr=None):
    """(Pr, Pc) with Pr * Pc == world.  Default Pr = 1: COLUMN slabs -- rank j holds A[:, C_j] with C_j = its own block, so a
    layer needs no all-gather at all (x[C_j] is local), its gathers touch N/W rows (48 MB at config 4: cache-resident) and
    the one collective is a reduce-scatter whose W-1 blocks arrive over W-1 different xGMI links at once.  On the fully
    connected mesh that beats the 2 x 4 grid in TIME although 2 x 4 receives fewer BYTES (192 vs 336 MB per layer at config
    4): every block is N/W rows whatever the grid, a collective with k peers uses k links, and 2 x 4 puts TWO collectives
    (1 link, then 3 links) on every layer's critical path -- measured tile SpMM 0.764 ms (1 x 8) / 0.745 (2 x 4) / 1.09 (the
    1-D row partition) per layer at config 4 (profiles/r04_shard_probe_config4.txt), each collective >= 0.63 ms."""
    if pr is None:
        pr = 1
    if pr < 1 or world % pr:
        raise ValueError(f"grid rows {pr} do not divide the world size {world}")
    return pr, world // pr


class GridLayout:
    """Blocks of Mb = ceil(N / W) rows; block b = r % W holds node rows r = q * W + b at local row q."""

    def __init__(self, n_rows, world, pr=None, chunks=1):
        self.N, self.W = int(n_rows), int(world)
        self.Pr, self.Pc = grid_shape(self.W, pr)
        per = (self.N + self.W - 1) // self.W
        self.C = max(1, min(int(chunks), per))            # row chunks of a block: the reduce-scatter of chunk c runs under the SpMM of c+1
        self.Mc = (per + self.C - 1) // self.C
        self.Mb = self.Mc * self.C                        # rows per block, padded
        self.M = self.Mb                                  # (what the inherited evaluation reads: rows per rank)

    def coords(self, b):
        return b // self.Pc, b % self.Pc

    def pos(self, r):
        """world-gathered position (rank order): what tables() / evaluate() index an all-gather over ALL ranks with"""
        return (r % self.W) * self.Mb + r // self.W

    def col_pos(self, r):
        """position of node r inside x[C_j] (the column group's gathered blocks, group-rank order i = 0..Pr-1)"""
        return ((r % self.W) // self.Pc) * self.Mb + r // self.W

    def row_pos(self, r):
        """position of node r inside a partial y[R_i]: chunk-major -- chunk c of every block of the row group (group-rank order
        j = 0..Pc-1) is ONE contiguous run, what a reduce-scatter of that chunk takes"""
        q = r // self.W
        return (q // self.Mc) * (self.Pc * self.Mc) + ((r % self.W) % self.Pc) * self.Mc + q % self.Mc

    def full_pos(self, r):
        """position in the replicated, grid-column-major [W * Mb, d] buffers (gprop / gego / e0_full): C_j is rows
        [j * Pr * Mb, (j + 1) * Pr * Mb)"""
        b = r % self.W
        return ((b % self.Pc) * self.Pr + b // self.Pc) * self.Mb + r // self.W


def build_tile(rowptr, col, val, rank, layout):
    """CSR (rowptr int32[Pc*Mb + 1], col int32, val fp32) of the tile A[R_i, C_j] of `rank` = (i, j): rows in row_pos order,
    columns relabelled to col_pos, entry order inside a row unchanged.  Vectorised on the inputs' device."""
    rp = torch.as_tensor(rowptr).long()
    cl = torch.as_tensor(col).long()
    vl = torch.as_tensor(val)
    dev = rp.device
    L = layout
    i, j = L.coords(rank)
    # row_pos order: [chunk c][block j'][row o of the chunk]  ->  node id (c * Mc + o) * W + (i * Pc + j')
    qq = (torch.arange(L.C, device=dev).view(-1, 1, 1) * L.Mc + torch.arange(L.Mc, device=dev).view(1, 1, -1))
    rows = (qq * L.W + (i * L.Pc + torch.arange(L.Pc, device=dev)).view(1, -1, 1)).reshape(-1)
    valid = rows < L.N
    rsafe = torch.where(valid, rows, torch.zeros_like(rows))
    deg = torch.where(valid, rp[rsafe + 1] - rp[rsafe], torch.zeros_like(rows))
    n_rows = rows.numel()
    row_of = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
    first = torch.zeros(n_rows + 1, dtype=torch.long, device=dev)
    first[1:] = torch.cumsum(deg, 0)
    src = rp[rsafe][row_of] + (torch.arange(int(first[-1].item()), device=dev) - first[row_of])
    c = cl[src]
    keep = ((c % L.W) % L.Pc) == j
    row_k, c_k, v_k = row_of[keep], c[keep], vl[src][keep]
    cnt = torch.bincount(row_k, minlength=n_rows)
    out_rp = torch.zeros(n_rows + 1, dtype=torch.long, device=dev)
    out_rp[1:] = torch.cumsum(cnt, 0)
    return out_rp.to(torch.int32).contiguous(), L.col_pos(c_k).to(torch.int32).contiguous(), v_k.to(torch.float32).contiguous()


class Grid2DLightGCN(ShardedLightGCN):
    """Same interface as ShardedLightGCN (train_epoch / tables / evaluate / describe), 2-D tiled propagation."""

    def __init__(self, n_users, n_items, dim, n_layers, csr, user_emb, item_emb, lam=1e-4, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 ops=None, device=None, grid_rows=None, reduce="collective", deterministic=False, chunks=None, force_collectives=False, capture=None, probe_rank_world=None):
        on = dist.is_available() and dist.is_initialized()
        self.group = None
        self.rank = dist.get_rank() if on else 0
        self.world = dist.get_world_size() if on else 1
        # measurement only (scripts/shard_probe.py): act as rank r of a W-rank job WITHOUT a process group -- every collective
        # becomes a local copy of this rank's own share, so a step costs exactly what the rank computes and launches
        self._probe = probe_rank_world is not None
        if self._probe:
            self.rank, self.world = int(probe_rank_world[0]), int(probe_rank_world[1])
            on = False
        self.ops = ops or HipOps()
        self.U, self.I, self.d, self.L = int(n_users), int(n_items), int(dim), int(n_layers)
        if self.L < 1:
            raise ValueError("Grid2DLightGCN needs n_layers >= 1")
        if reduce not in ("collective", "ordered"):
            raise ValueError("reduce must be 'collective' (reduce_scatter_tensor) or 'ordered' (all-to-all + sum in group-rank order)")
        self.lam, self.lr, self.betas, self.eps = lam, lr, betas, eps
        self.N = self.U + self.I
        if chunks is None:
            # (like the 1-D trainer: overlap only pays when there is a collective to hide and the blocks are big)
            per_rank = self.N // self.world
            coll = self.world > 1 or (bool(force_collectives) and on)
            chunks = 4 if (coll and per_rank >= (1 << 17)) else 2 if (coll and per_rank >= 4096) else 1
        self.layout = lay = GridLayout(self.N, self.world, grid_rows, chunks)
        self.reduce_mode = "ordered" if deterministic else reduce
        self.deterministic = bool(deterministic)
        # W == 1 (and groups of one rank) normally short-circuit every collective to a copy; force_collectives keeps them: a one-rank
        # RCCL group on a single-GPU box then exercises the real reduce_scatter / all_gather / all_reduce calls and their async handles
        self.force_collectives = bool(force_collectives) and on
        self.gather_mode = "collective"
        self.device = torch.device(device) if device is not None else user_emb.device
        self.gi, self.gj = lay.coords(self.rank)
        # EVERY rank creates EVERY group, in the same order (torch.distributed requires it); keep the two this rank is in
        self.col_group = self.row_group = None
        if on and (self.world > 1 or self.force_collectives):
            for j in range(lay.Pc):
                g_ = dist.new_group([i * lay.Pc + j for i in range(lay.Pr)])
                if j == self.gj:
                    self.col_group = g_
            for i in range(lay.Pr):
                g_ = dist.new_group([i * lay.Pc + j for j in range(lay.Pc)])
                if i == self.gi:
                    self.row_group = g_
        rowptr, col, val = (csr.rowptr, csr.col, csr.val) if hasattr(csr, "rowptr") else csr
        trp, tcl, tvl = build_tile(rowptr, col, val, self.rank, lay)
        self.tiles = []                          # one slab per row chunk: rows [c * Pc * Mc, (c + 1) * Pc * Mc) of the tile
        for c in range(lay.C):
            r0, r1 = c * lay.Pc * lay.Mc, (c + 1) * lay.Pc * lay.Mc
            a, b = int(trp[r0].item()), int(trp[r1].item())
            self.tiles.append(self.ops.make_slab((trp[r0:r1 + 1] - a).contiguous(), tcl[a:b].contiguous(), tvl[a:b].contiguous(), self.device))
        dev, Mb, W, d = self.device, lay.Mb, self.world, self.d
        z = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.e0, self.m, self.v = z(Mb, d), z(Mb, d), z(Mb, d)
        self.s = z(Mb, d)
        self.ybuf = [z(Mb, d), z(Mb, d)]
        self.xcol = z(lay.Pr * Mb, d)            # x[C_j] gathered within the column group
        self.partial = z(lay.Pc * Mb, d)         # A[R_i, C_j] . x[C_j]
        self.recv = z(lay.Pc * Mb, d) if self.reduce_mode == "ordered" else None
        self.gprop, self.gego, self.e0_full = z(W * Mb, d), z(W * Mb, d), z(W * Mb, d)   # replicated, grid-column-major
        own0 = (self.gj * lay.Pr + self.gi) * Mb
        self.own_full = slice(own0, own0 + Mb)                                           # this rank's block inside them
        self.col_full = slice(self.gj * lay.Pr * Mb, (self.gj + 1) * lay.Pr * Mb)        # C_j inside them
        # frontier bitmap over x[C_j] (+ one spill row for the minibatch nodes outside this column part): the first backward layer
        # gathers the minibatch's rows only (gprop is zero elsewhere); ops without it (the CPU stand-in) gather everything
        self.row_bits = self.ops.new_row_bits(lay.Pr * Mb + 1, dev) if hasattr(self.ops, "new_row_bits") else None
        self.t = 0
        self._plan = None
        self._graph = None
        self._graph_failed = False
        self._ysel = 0
        # step capture like the 1-D trainer's (kernels + collectives recorded once into a torch.cuda.CUDAGraph, replayed per step with
        # staged indices and device-resident Adam coefficients); None = wherever it can work, eager fallback if the capture fails
        self.capture = capture
        self.load_tables(user_emb, item_emb)

    def describe(self):
        L = self.layout
        blk = L.Mb * self.d * 4 / 1e6
        return (f"node rows in {self.world} round-robin blocks on a {L.Pr} x {L.Pc} grid ({L.Mb} rows/rank in {L.C} chunk(s)); per layer an all-gather within "
                f"the column group ({L.Pr - 1} x {blk:.1f} MB received) + a local tile SpMM + a reduce-scatter ({self.reduce_mode}) within "
                f"the row group ({L.Pc - 1} x {blk:.1f} MB received); per step {2 * self.L - 1} all-gathers, {2 * self.L} reduce-scatters, "
                f"one [6B,d] all-reduce; BPR replicated")

    # ------------------------------------------------------------------ collectives
    def _host_staged(self):
        return False if self._probe else super()._host_staged()

    def _host(self):
        return not self._probe and self.world > 1 and self.device.type == "cuda" and dist.get_backend() == "gloo"

    def _all_reduce(self, t):
        if (self.world == 1 and not self.force_collectives) or self._probe:
            return
        if self._host():
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)

    def _gather_full(self, local, out_full):
        """all-gather over ALL ranks in rank order (tables / evaluation): out_full[b * Mb + q] = row q of rank b"""
        if self._probe:
            out_full[self.rank * self.layout.Mb:(self.rank + 1) * self.layout.Mb].copy_(local)
        elif self.world == 1 and not self.force_collectives:
            out_full.copy_(local)
        elif self._host():
            h = torch.empty(out_full.shape, dtype=out_full.dtype)
            dist.all_gather_into_tensor(h, local.cpu().contiguous())
            out_full.copy_(h)
        else:
            dist.all_gather_into_tensor(out_full, local.contiguous())
        return out_full

    def _gather_col(self, x_own):
        """x[C_j] from the column group's blocks"""
        if self.layout.Pr == 1 and not self.force_collectives:
            return x_own
        if self._probe:
            self.xcol[self.gi * self.layout.Mb:(self.gi + 1) * self.layout.Mb].copy_(x_own)
            return self.xcol
        if self._host():
            h = torch.empty(self.xcol.shape, dtype=self.xcol.dtype)
            dist.all_gather_into_tensor(h, x_own.cpu().contiguous(), group=self.col_group)
            self.xcol.copy_(h)
        else:
            dist.all_gather_into_tensor(self.xcol, x_own, group=self.col_group)
        return self.xcol

    def _reduce_rows(self, partial, out, async_op=False):
        """out = sum over the row group of the members' partial blocks destined to this rank (partial: [Pc * rows, d], out:
        [rows, d]).  async_op (RCCL / gloo collective form only): returns the work handle instead of waiting."""
        lay = self.layout
        if lay.Pc == 1 and not self.force_collectives:
            out.copy_(partial)
            return None
        if self._probe:
            n = out.shape[0]
            out.copy_(partial[self.gj * n:(self.gj + 1) * n])
            return None
        if self.reduce_mode == "ordered":
            # direct reduce-scatter with a FIXED summation order: block k of every member goes to member k, which adds the Pc
            # blocks it receives in group-rank order (its own among them)
            recv = self.recv[: partial.shape[0]]
            if self._host():
                h = torch.empty(recv.shape, dtype=recv.dtype)
                dist.all_to_all_single(h, partial.cpu().contiguous(), group=self.row_group)
                recv.copy_(h)
            else:
                dist.all_to_all_single(recv, partial, group=self.row_group)
            r = recv.view(lay.Pc, out.shape[0], self.d)
            out.copy_(r[0])
            for k in range(1, lay.Pc):
                out.add_(r[k])
            return None
        if self._host():
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.reduce_scatter_tensor(h, partial.cpu().contiguous(), group=self.row_group)
            out.copy_(h)
            return None
        return dist.reduce_scatter_tensor(out, partial, group=self.row_group, async_op=async_op) if async_op else \
            dist.reduce_scatter_tensor(out, partial, group=self.row_group)

    def _layer(self, x_col, out, src_filter=None):
        """out (own block) = (A x)[block]: per row chunk a tile SpMM on x[C_j], then that chunk's reduce-scatter within the row
        group -- issued asynchronously (it runs on the collective's stream behind the SpMM that produced the chunk) while the
        next chunk's SpMM runs; the layer returns when every chunk has landed."""
        lay = self.layout
        pending = []
        for c in range(lay.C):
            part = self.partial[c * lay.Pc * lay.Mc:(c + 1) * lay.Pc * lay.Mc]
            if src_filter is not None:
                self.ops.spmm(self.tiles[c], x_col, y=part, src_filter=src_filter)   # (only the marked rows of x_col are non-zero)
            else:
                self.ops.spmm(self.tiles[c], x_col, y=part)
            w = self._reduce_rows(part, out[c * lay.Mc:(c + 1) * lay.Mc], async_op=lay.C > 1 or self.force_collectives)
            if w is not None:
                pending.append(w)
        for w in pending:
            w.wait()      # nccl: the launch stream waits for the collective's stream; gloo: host wait
        return out

    # ------------------------------------------------------------------ propagation
    def _forward(self):
        """light rows of the owned block into self.s"""
        L = self.L
        inv = 1.0 / (L + 1)
        x = self.e0
        for l in range(1, L + 1):
            y = self._layer(self._gather_col(x), self.ybuf[l & 1])
            if l == 1:
                torch.add(self.e0, y, out=self.s)
            else:
                self.s.add_(y)
            x = y
        self.s.mul_(inv)

    # ------------------------------------------------------------------ training
    def reserve(self, n_triplets, batch):
        n_steps = (int(n_triplets) + batch - 1) // batch
        if self._plan is None or self._plan["cap_steps"] < n_steps or self._plan["batch"] != batch:
            # `rows` is baked into the captured step graph: allocated once per batch size (see ShardedLightGCN.reserve)
            rows = None if (self._plan is None or self._plan["batch"] != batch) else self._plan["rows"]
            if rows is None:
                rows = torch.zeros(6 * batch, self.d, device=self.device, dtype=torch.float32)   # light rows | ego rows of the minibatch
                self._graph = None
            self._plan = {"cap_steps": n_steps, "batch": batch,
                          "loss": torch.zeros(n_steps, _lib.RK_LOSS_PARTIALS, device=self.device, dtype=torch.float32), "rows": rows}
        return self._plan

    def _bpr_pos(self, nodes):
        return self.layout.full_pos(nodes)     # the replicated buffers are grid-column-major

    def _step_core(self, nb, idx_local3, idx_own3, idx_pos3, lp, keys, adam):
        """One train step given the minibatch's 3*nb role-major index rows (slices of the epoch plan when eager, the fixed staging
        buffers when captured: the inherited step() / _replay() drive it exactly like the 1-D trainer's)."""
        ops, L = self.ops, self.L
        self._forward()
        rows = self._plan["rows"]
        light_rows, ego_rows = rows[: 3 * nb], rows[3 * nb: 6 * nb]
        mask = idx_own3 if (self.world > 1 or self.force_collectives) else None
        ops.gather_rows(self.s, idx_local3, mask, light_rows)
        ops.gather_rows(self.e0, idx_local3, mask, ego_rows)
        self._all_reduce(rows[: 6 * nb])
        self.e0_full.index_copy_(0, idx_pos3, ego_rows)            # the minibatch's rows of E0 (reg term); duplicates carry equal values
        ru, rp, rn = idx_pos3[:nb], idx_pos3[nb:2 * nb], idx_pos3[2 * nb:3 * nb]
        ops.bpr(self.d, L, self.lam, light_rows, self.e0_full, self.gprop, self.gego, ru, rp, rn, lp, keys=keys)
        # backward: t_1 = g + A g, ..., grad = gego + A t_{L-1}; the first layer's x[C_j] is a view of the replicated gprop
        x_col = self.gprop[self.col_full]
        frontier, cidx = self.row_bits, None
        if frontier is not None:
            n_col = self.layout.Pr * self.layout.Mb
            cidx = idx_pos3 - self.col_full.start
            cidx = torch.where((cidx >= 0) & (cidx < n_col), cidx, torch.full_like(cidx, n_col))
            ops.mark_rows(frontier, cidx, True)
        for j in range(1, L + 1):
            last = j == L
            y = self._layer(x_col, self.ybuf[j & 1], src_filter=frontier if j == 1 else None)
            y.add_((self.gego if last else self.gprop)[self.own_full])
            if last:
                ops.adam(self.e0, y, self.m, self.v, adam["t"], adam["lr"], adam["b1"], adam["b2"], adam["eps"], coef=adam.get("coef"))
            else:
                x_col = self._gather_col(y)
        ops.zero_rows(self.gprop, self.gego, idx_pos3)
        if frontier is not None:
            # the bitmap describes THIS minibatch only (like ShardedLightGCN._step_core): stale bits would widen the next step's
            # filtered gather and make its summation order depend on the minibatches that came before
            ops.mark_rows(frontier, cidx, False)
