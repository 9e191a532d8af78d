"""CPU checks of the LDS-resident SpMM's plan (recad_amd/csrc/spmm_lds.hip): rk_lds_plan_build_host is pure host code, so
the schedule can be validated without a GPU -- a numpy walk of the plan in exactly the order the kernel consumes it
(tasks -> chunk partials -> per-row sums -> dinv scaling) must reproduce the normalised adjacency product, every stored
entry must appear exactly once in the column stream, and graphs that do not qualify must be refused."""
import ctypes as C

import numpy as np
import pytest

from recad_amd import _lib, synth

H = dict(MAGIC=0, NWG=1, U=2, I=3, D=4, LSU=5, LSI=6, NBLK0=7, NBLK1=8, WG_OFS=9, BLK_OFS=10, DINV_OFS=11, LDS_BYTES=12,
         CHUNK=13, NWORDS=14, PERM0=15, PERM1=16, MQ_OFS=17, WGX_OFS=18)
LB = dict(ROW0=0, NROWS=1, NPART=2, NTASKS=3, TASK_OFS=4, DST_OFS=5, PP_OFS=6, STREAM_OFS=7, WORDS=8)


def norm_adj_csr(U, I, ptr, idx):
    """rowptr, col, val (float32, val = dinv[r]*dinv[c] like implicit.py:259-277) of the bipartite adjacency."""
    N = U + I
    users = np.repeat(np.arange(U), np.diff(ptr))
    items = idx.astype(np.int64)
    rows = np.concatenate([users, U + items])
    cols = np.concatenate([U + items, users])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    deg = np.bincount(rows, minlength=N)
    rowptr = np.zeros(N + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(deg)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, 1.0 / np.sqrt(deg.astype(np.float64)), 0.0).astype(np.float32)
    val = (dinv[rows] * dinv[cols]).astype(np.float32)
    return rowptr, cols.astype(np.int32), val


def build_plan(U, I, rowptr, col, val, dim, n_cu=256):
    plan, n_words, info = C.c_void_p(), C.c_int64(0), _lib.LdsInfo()
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    cc = np.ascontiguousarray(col, dtype=np.int32)
    vv = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    rc = _lib.lib().rk_lds_plan_build_host(U, I, rp.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p),
                                           None if vv is None else vv.ctypes.data_as(C.c_void_p), dim, n_cu,
                                           C.byref(plan), C.byref(n_words), C.byref(info))
    assert rc == 0, _lib.lib().rk_last_error()
    if n_words.value == 0:
        return None, info
    words = np.zeros(n_words.value, dtype=np.int32)
    _lib.check(_lib.lib().rk_lds_plan_words(plan, words.ctypes.data_as(C.c_void_p)), "rk_lds_plan_words")
    _lib.lib().rk_lds_plan_destroy(plan)
    return words, info


B128_GROUPS = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])


def emulate(words, x, check_banks=True):
    """y = A.x computed the way spmm_lds_kernel does, from the plan alone.  x, y: row-major [N, d] float32."""
    U, I, d = int(words[H["U"]]), int(words[H["I"]]), int(words[H["D"]])
    N = U + I
    S = {0: 1 << int(words[H["LSI"]]), 1: 1 << int(words[H["LSU"]])}   # half 0 gathers the items table
    dinv = words[int(words[H["DINV_OFS"]]): int(words[H["DINV_OFS"]]) + N].view(np.float32)
    stream16 = words.view(np.uint16)
    y = np.full((N, d), np.nan, dtype=np.float32)
    seen_rows = np.zeros((N, d), dtype=np.int32)
    wg = words[int(words[H["WG_OFS"]]): int(words[H["WG_OFS"]]) + 4 * int(words[H["NWG"]])].reshape(-1, 4)
    entries = 0
    for half, sl, rb, _ in wg:
        half, sl, rb = int(half), int(sl), int(rb)
        Sh = S[half]
        LPn = Sh // 4
        SL = 64 // LPn
        n_src = U if half else I
        src0 = 0 if half else U
        dst0 = U if half else 0
        bd = words[int(words[H["BLK_OFS"]]) + ((int(words[H["NBLK0"]]) if half else 0) + rb) * LB["WORDS"]:][: LB["WORDS"]]
        cols = slice(sl * Sh, (sl + 1) * Sh)
        K = 16 // LPn
        table = np.zeros((n_src + K, Sh), dtype=np.float32)
        perm = words[int(words[H["PERM1" if half else "PERM0"]]):][:n_src]
        assert np.array_equal(np.sort(perm), np.arange(n_src))
        table[perm] = x[src0: src0 + n_src, cols] * dinv[src0: src0 + n_src, None]
        part = np.zeros((max(int(bd[LB["NPART"]]), 1), Sh), dtype=np.float32)
        written = np.zeros(part.shape[0], dtype=np.int32)
        for t in range(int(bd[LB["NTASKS"]])):
            ofs, nb = (int(v) for v in words[int(bd[LB["TASK_OFS"]]) + 2 * t: int(bd[LB["TASK_OFS"]]) + 2 * t + 2])
            dst = words[int(bd[LB["DST_OFS"]]) + t * SL: int(bd[LB["DST_OFS"]]) + (t + 1) * SL]
            base = (int(bd[LB["STREAM_OFS"]]) + ofs) * 8
            blk = stream16[base: base + nb * SL * 8].reshape(nb, SL, 8).astype(np.int64)
            assert blk.max(initial=0) < n_src + K
            if check_banks:
                # every ds_read_b128 of the walk is conflict-free: the K slots of a 16-lane group read K different
                # bank classes (row index mod K)
                for grp in B128_GROUPS:
                    slots = sorted({lane // LPn for lane in grp})
                    cl = blk[:, slots, :] % K                       # [nb, K, 8]
                    srt = np.sort(cl, axis=1)
                    assert (srt[:, 1:, :] != srt[:, :-1, :]).all()
            acc = np.zeros((SL, Sh), dtype=np.float32)
            for b in range(nb):
                for e in range(8):
                    acc = acc + table[blk[b, :, e]]
            if half == 0 and sl == 0 or half == 1 and sl == 0:
                entries += int((blk < n_src).sum())
            live = dst >= 0
            assert (blk[:, ~live, :] >= n_src).all()      # empty slots only read zero rows
            part[dst[live]] = acc[live]
            written[dst[live]] += 1
        assert (written[: int(bd[LB["NPART"]])] == 1).all()
        pp = words[int(bd[LB["PP_OFS"]]): int(bd[LB["PP_OFS"]]) + int(bd[LB["NROWS"]]) + 1]
        for lr in range(int(bd[LB["NROWS"]])):
            acc = np.zeros(Sh, dtype=np.float32)
            for p in range(int(pp[lr]), int(pp[lr + 1])):
                acc = acc + part[p]
            r = dst0 + int(bd[LB["ROW0"]]) + lr
            y[r, cols] = acc * dinv[r]
            seen_rows[r, cols] += 1
    assert (seen_rows == 1).all()     # every (row, column) of the output is produced by exactly one workgroup
    return y, entries


@pytest.mark.parametrize("shape,dim", [("tiny", 64), ("tiny", 32), ("tiny", 128)])
def test_plan_reproduces_the_product(shape, dim):
    data = synth.make(shape)
    U, I = data["n_users"], data["n_items"]
    rowptr, col, val = norm_adj_csr(U, I, *data["train"])
    words, info = build_plan(U, I, rowptr, col, val, dim, n_cu=64)
    assert words is not None and int(words[H["MAGIC"]]) == 0x4c445331
    assert info.n_wg == int(words[H["NWG"]]) and info.lds_bytes <= 160 * 1024 - 64
    rng = np.random.default_rng(5)
    x = rng.standard_normal((U + I, dim)).astype(np.float32)
    y, entries = emulate(words, x)
    assert entries == len(col)        # every stored entry exactly once (per slice)
    ref = np.zeros((U + I, dim), dtype=np.float64)
    rows = np.repeat(np.arange(U + I), np.diff(rowptr))
    np.add.at(ref, rows, val.astype(np.float64)[:, None] * x[col].astype(np.float64))
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max()


def test_plan_ml1m_shape_fits_and_balances():
    data = synth.make("ml1m")
    U, I = data["n_users"], data["n_items"]
    rowptr, col, val = norm_adj_csr(U, I, *data["train"])
    words, info = build_plan(U, I, rowptr, col, val, 64, n_cu=256)
    assert words is not None
    assert info.n_wg == 256 and (info.lpa, info.lpb) == (1, 1)     # 4-float slices of both tables: one lane per entry
    # per-block work (stream units) within 15 % of the mean in each half
    nb0, nb1 = int(words[H["NBLK0"]]), int(words[H["NBLK1"]])
    for lo, hi in ((0, nb0), (nb0, nb0 + nb1)):
        units = []
        for bi in range(lo, hi):
            bd = words[int(words[H["BLK_OFS"]]) + bi * LB["WORDS"]:][: LB["WORDS"]]
            t = words[int(bd[LB["TASK_OFS"]]): int(bd[LB["TASK_OFS"]]) + 2 * int(bd[LB["NTASKS"]])].reshape(-1, 2)
            units.append(int(t[:, 1].sum()))
        units = np.asarray(units, dtype=np.float64)
        assert units.max() <= 1.15 * units.mean(), units


def test_plan_refuses_graphs_that_do_not_qualify():
    data = synth.make("tiny")
    U, I = data["n_users"], data["n_items"]
    rowptr, col, val = norm_adj_csr(U, I, *data["train"])
    bad = val.copy()
    bad[7] *= 1.01                       # not dinv[r]*dinv[c] any more
    assert build_plan(U, I, rowptr, col, bad, 64)[0] is None
    col2 = col.copy()
    col2[0] = 0                          # a user row pointing at a user column: not bipartite
    assert build_plan(U, I, rowptr, col2, val, 64)[0] is None
    assert build_plan(U, I, rowptr, col, val, 6)[0] is None      # dim % 4
    # a class table that cannot fit 160 KB of LDS at the narrowest slice
    Ub, Ib = 70000, 200
    ptr = np.arange(Ub + 1, dtype=np.int64)
    idx = (np.arange(Ub) % Ib).astype(np.int32)
    rp, cc, vv = norm_adj_csr(Ub, Ib, ptr, idx)
    assert build_plan(Ub, Ib, rp, cc, vv, 64)[0] is None


def _multi_queues(words):
    """The multi-phase launch's work-item queues (plan words at LP_MQ_OFS, csrc/spmm_lds.h)."""
    o = int(words[H["MQ_OFS"]])
    n_queues, n_groups, G = (int(v) for v in words[o: o + 3])
    queues = []
    for q in range(n_queues):
        n_items, first = int(words[o + 4 + 2 * q]), int(words[o + 4 + 2 * q + 1])
        queues.append(words[first * 4: first * 4 + 4 * n_items].reshape(-1, 4))
    members = words[o + 4 + 2 * n_queues: o + 4 + 2 * n_queues + n_groups]
    return n_queues, n_groups, G, queues, members


@pytest.mark.parametrize("shape,dim,n_cu", [("tiny", 64, 64), ("tiny", 32, 64), ("tiny", 16, 64), ("tiny", 128, 256), ("tiny", 256, 256), ("ml1m", 64, 256)])
def test_multi_phase_queues_cover_the_launch_and_cannot_deadlock(shape, dim, n_cu):
    """spmm_lds_multi_kernel's contract with the plan: the queue lists hold every (half, slice, block) workgroup of the
    single-phase table exactly once; a column group (the workgroups that exchange data between consecutive layers) never
    straddles queues and its member count is what the arrival counter waits for; and -- simulated with FEWER resident
    workgroups than the grid, in adversarial order -- handing a queue's items out in phase-major ticket order never leaves a
    running workgroup waiting for an item nobody has taken."""
    data = synth.make(shape)
    U, I = data["n_users"], data["n_items"]
    rowptr, col, val = norm_adj_csr(U, I, *data["train"])
    words, info = build_plan(U, I, rowptr, col, val, dim, n_cu=n_cu)
    assert words is not None
    n_queues, n_groups, G, queues, members = _multi_queues(words)
    S = {0: 1 << int(words[H["LSI"]]), 1: 1 << int(words[H["LSU"]])}
    assert G == max(S.values()) and n_groups == dim // G and 1 <= n_queues <= 8 and n_groups <= 64
    wg = words[int(words[H["WG_OFS"]]): int(words[H["WG_OFS"]]) + 4 * int(words[H["NWG"]])].reshape(-1, 4)
    # the 64-byte workgroup records: table entry + its block descriptor (what the kernels read instead of the header chain)
    rec = words[int(words[H["WGX_OFS"]]): int(words[H["WGX_OFS"]]) + 16 * int(words[H["NWG"]])].reshape(-1, 16)
    assert np.array_equal(rec[:, :3], wg[:, :3])
    for b, (h, s_, rb, _) in enumerate(wg):
        bd = words[int(words[H["BLK_OFS"]]) + ((int(words[H["NBLK0"]]) if h else 0) + int(rb)) * LB["WORDS"]:][: LB["WORDS"]]
        assert np.array_equal(rec[b, 4:12], bd) and int(rec[b, 3]) == int(s_) * S[int(h)] // G
    multi = sorted(int(b) for qv in queues for b, _, _, _ in qv)
    assert multi == list(range(int(words[H["NWG"]])))          # every record exactly once
    count = np.zeros(n_groups, dtype=np.int64)
    for q, qv in enumerate(queues):
        for b, _, _, g in qv:
            assert int(g) == int(rec[int(b), 3]) and int(g) % n_queues == q      # the group of its slice; its home queue
            count[int(g)] += 1
    assert np.array_equal(count, members)
    # ticket-order simulation: R < grid workgroups, each pulls from queue (b % 8) % n_queues; an item of phase p needs all
    # members of its group to have FINISHED phase p - 1.  A workgroup that cannot start spins (keeps its slot).
    n_phases, rng = 3, np.random.default_rng(dim)
    for resident in (1, 3, max(2, int(words[H["NWG"]]) // 5)):
        head = [0] * n_queues
        arrived = np.zeros(n_groups, dtype=np.int64)
        blocks = list(range(int(words[H["NWG"]])))
        running = {}                                       # block -> (queue, ticket) it holds
        waiting_blocks = blocks[::-1]
        done_items = 0
        total = sum(len(qv) for qv in queues) * n_phases
        for _ in range(50 * total + 100):
            while len(running) < resident and waiting_blocks:      # admit blocks in an arbitrary (reversed) order
                b = waiting_blocks.pop()
                q = (b % 8) % n_queues
                running[b] = (q, head[q]); head[q] += 1
            if not running:
                break
            b = list(running)[int(rng.integers(len(running)))]   # an arbitrary running block makes progress
            q, t = running[b]
            if t >= len(queues[q]) * n_phases:
                del running[b]                                   # queue exhausted: the block exits
                continue
            phase, (_, _, _, g) = t // len(queues[q]), queues[q][t % len(queues[q])]
            if phase > 0 and arrived[int(g)] < members[int(g)] * phase:
                continue                                         # spins
            arrived[int(g)] += 1
            done_items += 1
            running[b] = (q, head[q]); head[q] += 1
        assert done_items == total and not running, (resident, done_items, total)


def test_plan_builder_leaves_the_callers_affinity_alone():
    """The builder places its pool's short-lived workers on CPUs of the caller's mask (host/lds_plan_host.h: place_self).  A first
    version set the workers' affinity from OUTSIDE: a worker that had already finished has thread id 0, and the call then pinned
    the CALLER -- every later thread and child process of the host program inherited one CPU.  Tiny graphs (more threads than
    work: workers exit at once) are the case that showed it."""
    import os
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity on this platform")
    before = os.sched_getaffinity(0)
    rng = np.random.default_rng(7)
    for k in range(40):
        U, I = int(rng.integers(20, 400)), int(rng.integers(20, 400))
        deg = rng.integers(1, min(I, 12), U)
        ptr = np.zeros(U + 1, dtype=np.int64)
        ptr[1:] = np.cumsum(deg)
        idx = np.concatenate([np.sort(rng.choice(I, size=int(n), replace=False)) for n in deg])
        words, _ = build_plan(U, I, *norm_adj_csr(U, I, ptr, idx), 64)
        assert words is not None
        assert os.sched_getaffinity(0) == before, k
