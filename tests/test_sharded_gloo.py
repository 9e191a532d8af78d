"""N>1 path on CPU: world_size 1/2/3 gloo runs of the row-sharded LightGCN trainer + user-sharded
evaluation (oracle ops injected) must reproduce the single-process oracle and the goldens; and the
self-launching `bench.py --gpus N` front end must rendezvous N workers."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from tests import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOPKS = (10, 20, 50, 100)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, out_path, chunks, gather, grid=None):
    """grid: None = the 1-D row partition (ShardedLightGCN); (grid rows, reduce mode) = the 2-D tiling (Grid2DLightGCN)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from recad_amd.sharded import ShardedLightGCN
    from recad_amd.sharded2d import Grid2DLightGCN
    from tests._oracle_ops import OracleOps
    g = G.load(name)
    U, I, d, L = int(g["n_users"]), int(g["n_items"]), int(g["dim"]), int(g["layers"])
    csr = orc.coo_to_csr(U + I, g["graph_row"], g["graph_col"], g["graph_val"])
    u0, i0 = G.lightgcn_init(g)
    if grid is None:
        tr = ShardedLightGCN(U, I, d, L, csr, torch.from_numpy(u0), torch.from_numpy(i0), ops=OracleOps(), device=torch.device("cpu"),
                             chunks=chunks, gather=gather)
        assert tr.layout.C == chunks
    else:
        tr = Grid2DLightGCN(U, I, d, L, csr, torch.from_numpy(u0), torch.from_numpy(i0), ops=OracleOps(), device=torch.device("cpu"),
                            grid_rows=grid[0], reduce=grid[1], chunks=chunks)
        assert tr.layout.Pr * tr.layout.Pc == world and (grid[0] is None or tr.layout.Pr == grid[0]) and tr.layout.C == chunks
    losses = []
    for s in range(min(4, len(g["batch_len"]))):
        n = int(g["batch_len"][s])
        u, p, ng = (torch.from_numpy(g["batches"][s, k, :n].astype(np.int64)) for k in range(3))
        losses.append(float(tr.train_epoch(u, p, ng, n)[0]))
    # two steps in ONE epoch call with a short last batch (the per-epoch index plan)
    n = int(g["batch_len"][0])
    u, p, ng = (torch.from_numpy(g["batches"][0, k, :n].astype(np.int64)) for k in range(3))
    two = tr.train_epoch(u, p, ng, (n + 1) // 2 + 1)
    assert two.numel() == 2 and bool(torch.isfinite(two).all())
    users, items = tr.tables()
    ev = tr.evaluate(g["train_ptr"], g["train_idx"], g["target_ids"], K=100, topks=TOPKS)
    if rank == 0:
        np.savez(out_path, losses=np.asarray(losses), users=users.numpy(), items=items.numpy(), hits=np.asarray(ev["hit_counts"]),
                 n_users=ev["eligible_users"], tmean=np.asarray(ev["target_score_mean"]))
    dist.barrier()
    dist.destroy_process_group()


# world 8 = the target machine's only world size: the 1-D partition (collective and one-shot gathers) and the 2-D tiling on its
# 2 x 4 default grid, 4 x 2, and the degenerate 1 x W / W x 1 grids; N % W != 0 throughout
@pytest.mark.parametrize("world,chunks,gather,grid", [
    (1, 1, "collective", None), (2, 1, "collective", None), (2, 3, "collective", None), (3, 2, "collective", None), (3, 2, "direct", None),
    (2, 4, "direct", None), (8, 2, "collective", None), (8, 1, "direct", None),
    (1, 1, "collective", (None, "collective")), (2, 1, "collective", (1, "collective")), (2, 1, "collective", (2, "ordered")),
    (4, 1, "collective", (None, "collective")), (6, 1, "collective", (3, "ordered")),
    (8, 1, "collective", (None, "collective")), (8, 1, "collective", (2, "ordered")), (8, 1, "collective", (4, "collective")),
    (8, 3, "collective", (2, "collective")), (8, 2, "collective", (1, "ordered")), (3, 2, "collective", (None, "collective"))])
def test_sharded_matches_oracle(tmp_path, world, chunks, gather, grid):
    name = "lightgcn_game_d64_tg"
    out = str(tmp_path / f"w{world}.npz")
    mp.spawn(_worker, args=(world, _free_port(), name, out, chunks, gather, grid), nprocs=world, join=True)
    res = np.load(out)
    g = G.load(name)
    U, I, L = int(g["n_users"]), int(g["n_items"]), int(g["layers"])
    assert (U + I) % world != 0 or world in (1, 4)   # the ragged case (N % W != 0) is the one exercised
    csr = orc.coo_to_csr(U + I, g["graph_row"], g["graph_col"], g["graph_val"])
    u, i = G.lightgcn_init(g)
    st = orc.AdamState(u.shape, i.shape)
    for s in range(len(res["losses"])):
        n = int(g["batch_len"][s])
        ref = orc.lightgcn_step(csr, u, i, st, *(g["batches"][s, k, :n] for k in range(3)), L)
        assert abs(res["losses"][s] - ref) <= 1e-5 * abs(ref), (s, res["losses"][s], ref)
        assert abs(res["losses"][s] - g["losses"][s]) <= 1e-5 * abs(g["losses"][s])
    # the extra two-step epoch of the worker
    n = int(g["batch_len"][0])
    b = (n + 1) // 2 + 1
    for lo in (0, b):
        orc.lightgcn_step(csr, u, i, st, *(g["batches"][0, k, lo:min(n, lo + b)] for k in range(3)), L)
    assert G.relerr(res["users"], u) < 1e-5 and G.relerr(res["items"], i) < 1e-5
    # user-sharded evaluation == single-process oracle evaluation of the same tables
    light = orc.lightgcn_propagate(csr, u, i, L)
    rows, _ = orc.evaluate(lambda uu: orc.score_rows(light[uu:uu + 1], light[U:])[0], I, g["train_ptr"], g["train_idx"],
                           g["target_ids"], TOPKS, K=100)
    T = len(g["target_ids"])
    assert int(res["n_users"]) == len(rows) // T
    ref_hits = rows[:, 2:].reshape(-1, T, len(TOPKS)).sum(axis=0) if T > 1 else rows[:, 2:].sum(axis=0, keepdims=True)
    # a rank flip needs two scores within summation noise of each other: allow one per threshold
    assert np.abs(res["hits"] - ref_hits).max() <= 1, (res["hits"], ref_hits)
    assert abs(res["tmean"][0] - rows[0::T, 1].mean()) <= 1e-5 * max(1e-3, abs(rows[0::T, 1].mean()))


def test_slab_partition_covers_graph():
    from recad_amd.sharded import RowLayout, build_slab_chunks
    rng = np.random.default_rng(0)
    n = 101
    deg = rng.integers(0, 9, n)
    rowptr = np.zeros(n + 1, dtype=np.int32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int32)
    val = rng.random(rowptr[-1]).astype(np.float32)
    for W, Cc in ((1, 1), (2, 1), (4, 3), (8, 2), (3, 5)):
        lay = RowLayout(n, W, Cc)
        inv = {int(lay.pos(np.int64(r))): r for r in range(n)}
        assert len(inv) == n and max(inv) < W * lay.M and lay.M == lay.C * lay.Mc
        seen = 0
        for g in range(W):
            chunks = build_slab_chunks(rowptr, col, val, g, lay)
            assert len(chunks) == lay.C
            own = list(range(g, n, W))
            for c, (lp, lc, lv) in enumerate(chunks):
                lp, lc, lv = lp.numpy(), lc.numpy(), lv.numpy()
                assert len(lp) == lay.Mc + 1 and lp[0] == 0
                lo, hi = lay.chunk_range(g, c)
                for k in range(lay.Mc):
                    q = c * lay.Mc + k
                    a, b = lp[k], lp[k + 1]
                    if q >= len(own):
                        assert a == b
                        continue
                    r = own[q]
                    assert int(lay.pos(np.int64(r))) == lo + k
                    assert [inv[int(x)] for x in lc[a:b]] == col[rowptr[r]:rowptr[r + 1]].tolist()
                    assert np.array_equal(lv[a:b], val[rowptr[r]:rowptr[r + 1]])
                    seen += b - a
        assert seen == rowptr[-1]


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` must start its own two workers (no torchrun) and print one JSON object."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] and d["n_gpus"] == 2 and d["rank_sum"] == 1.0 and d["parallel"] == "replicas" and d["scaling"] == "weak"
    # the strong-scaling jobs live in `also`: config 4 in the default form, then the SAME job in the three exchange forms the first
    # hardware run has to choose between (reduce_scatter_tensor / one-shot all-to-all + ordered sum / 1-D one-shot all-gather), each
    # with 1 and 4 row chunks, then the yelp shape
    assert d["also_sharded_legs"] == ["config4_rows2d", "config4_collective_c1", "config4_collective_c4", "config4_ordered_c1",
                                      "config4_ordered_c4", "config4_rows_direct_c1", "config4_rows_direct_c4", "config3_yelp_rows2d"]
    import bench
    assert d["also_sharded_legs"] == bench.also_sharded_leg_names() and bench.also_sharded_leg_names(True) == ["config3_yelp_rows2d"]
    assert {(v[1], v[2], v[3], v[4]) for v in bench.SHARDED_VARIANTS} == {
        ("rows2d", "collective", "collective", 1), ("rows2d", "collective", "collective", 4), ("rows2d", "ordered", "collective", 1),
        ("rows2d", "ordered", "collective", 4), ("rows", "collective", "direct", 1), ("rows", "collective", "direct", 4)}
    assert bench.Deadline.EXIT_CODE != 0
    # a world that disagrees with --gpus is refused
    env2 = dict(env, RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=120, env=env2)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_line_is_one_quantity_for_every_n():
    """SCALE-readiness: `bench.py --gpus N` reports BASELINE.json's metric on BASELINE.json's workload for N = 1 and N = 8 alike
    (replicas, weak scaling), so the driver's 1 -> 8 series is one curve; a sharded job as the top-level line is a labelled extra
    mode whose metric says so."""
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    lines = {}
    for n in (1, 2, 4, 8):
        a = bench.parse(["--gpus", str(n), "--steps", "20", "--warmup", "5"])
        assert (a.workload, a.parallel, a.dim, a.steps, a.warmup) == ("ml1m", "replicas", 64, 20, 5)
        lines[n] = (bench.metric_for(a), a.workload, a.dim, a.layers, a.batch, a.graph)
        assert bench.metric_for(a) == base["metric"]
    assert len(set(lines.values())) == 1
    a = bench.parse(["--gpus", "8"])
    assert (a.steps, a.warmup) == (bench.parse([]).steps, bench.parse([]).warmup)       # same defaults for every N
    a = bench.parse(["--gpus", "8", "--parallel", "rows2d", "--workload", "config4"])
    assert bench.metric_for(a) != base["metric"] and "extra mode" in bench.metric_for(a) and a.steps == 20
    a = bench.parse(["--gpus", "1", "--workload", "yelp"])
    assert bench.metric_for(a) != base["metric"] and a.dim == 128
    # the dry run prints the same metric / workload string for N = 1 and N = 2 (what the real line carries)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    got = []
    for n in ("1", "2"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--dry-run"], capture_output=True, text=True,
                           timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        got.append((d["metric"], d["config"]["workload"], d["scaling"]))
    assert got[0] == got[1] and got[0][0] == base["metric"] and "ml1m-shaped" in got[0][1]


def test_grid_tiles_cover_graph():
    """Grid2D layout: every stored entry of A lands in exactly one rank's tile, at the row / column positions the layer's
    all-gather (column group) and reduce-scatter (row group) use; the replicated buffers' column parts are contiguous."""
    from recad_amd.sharded2d import GridLayout, build_tile
    rng = np.random.default_rng(1)
    n = 103
    deg = rng.integers(0, 9, n)
    rowptr = np.zeros(n + 1, dtype=np.int32); rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int32)
    val = rng.random(rowptr[-1]).astype(np.float32)
    dense = np.zeros((n, n), dtype=np.float64)
    np.add.at(dense, (np.repeat(np.arange(n), deg), col), val)
    for W, pr, Cc in ((1, None, 1), (2, 1, 2), (2, 2, 1), (4, None, 3), (8, 2, 2), (8, 4, 1), (6, 3, 4), (8, 1, 3)):
        lay = GridLayout(n, W, pr, Cc)
        nodes = np.arange(n, dtype=np.int64)
        assert len(set(lay.full_pos(nodes).tolist())) == n and len(set(lay.pos(nodes).tolist())) == n
        rebuilt = np.zeros_like(dense)
        for rank in range(W):
            i, j = lay.coords(rank)
            rp, cc, vv = (t.numpy() for t in build_tile(rowptr, col, val, rank, lay))
            assert len(rp) == lay.Pc * lay.Mb + 1
            rows_i = [r for r in range(n) if (r % W) // lay.Pc == i]
            cols_j = [c for c in range(n) if (c % W) % lay.Pc == j]
            inv_row = {int(lay.row_pos(np.int64(r))): r for r in rows_i}
            inv_col = {int(lay.col_pos(np.int64(c))): c for c in cols_j}
            assert len(inv_row) == len(rows_i) and len(inv_col) == len(cols_j)
            fp = lay.full_pos(np.asarray(cols_j, dtype=np.int64)) if cols_j else np.zeros(0, dtype=np.int64)
            assert all(j * lay.Pr * lay.Mb <= p < (j + 1) * lay.Pr * lay.Mb for p in fp)          # C_j is one contiguous run
            assert all(int(lay.full_pos(np.int64(c))) - j * lay.Pr * lay.Mb == int(lay.col_pos(np.int64(c))) for c in cols_j)
            for k in range(lay.Pc * lay.Mb):
                a, b = rp[k], rp[k + 1]
                if k not in inv_row:
                    assert a == b
                    continue
                r = inv_row[k]
                for e in range(a, b):
                    rebuilt[r, inv_col[int(cc[e])]] += vv[e]
        assert np.allclose(rebuilt, dense, rtol=0, atol=1e-12)


def test_bench_deadline_prints_the_line_then_leaves_nonzero():
    """bench.Deadline (the N > 1 `also` legs run under it): when a collective never completes, rank 0 prints the already-measured line
    with also.error FIRST, then the process leaves with a NON-zero code -- the line survives, the code tells the driver something hung
    (round 5 left with 0)."""
    code = (
        "import sys, time, json; sys.path.insert(0, %r); import bench\n"
        "def emit(err=None): print(json.dumps({'value': 1.0, 'also': {'error': err}}), flush=True)\n"
        "dl = bench.Deadline(0.5, 0, emit)\n"
        "time.sleep(30)   # a collective that never completes\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    import bench
    assert p.returncode == bench.Deadline.EXIT_CODE != 0, (p.returncode, p.stderr[-500:])
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] == 1.0 and "deadline" in d["also"]["error"]
    # a cancelled deadline does nothing
    code2 = ("import sys, time; sys.path.insert(0, %r); import bench\n"
             "dl = bench.Deadline(0.3, 0, lambda err=None: print('fired'))\n"
             "dl.cancel(); time.sleep(1.0); print('ok')\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "ok"
