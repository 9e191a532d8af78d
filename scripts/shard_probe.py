"""Per-rank compute of the row-sharded step, measured on ONE GPU: rank 0's slab of a W-way round-robin partition
(recad_amd/sharded.py: RowLayout + build_slab_chunks + HipOps.spmm), W = 1, 2, 4, 8.  What a rank computes per layer
(its chunked local SpMMs over the full gathered table) is timed with no collective at all -- the compute side of the
strong-scaling curve that the driver's multi-GPU run would complete with the all-gather side.
    python3 scripts/shard_probe.py [workload=config4] [dim=64] [reps=10] [chunks=2]
Also timed (W > 1): the same layer as C x C tiles chained through the `add` epilogue, source chunk outermost -- the
consumer-side-overlap order of ShardedLightGCN._layer -- i.e. the compute price of starting on arrived column blocks."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, synth
from recad_amd.sharded import HipOps, RowLayout, build_slab_chunks

name = sys.argv[1] if len(sys.argv) > 1 else "config4"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n_chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 2
only_step = len(sys.argv) > 5 and sys.argv[5] == "step"      # just the whole-step section
dev = torch.device("cuda:0")
big = name in ("c4s", "config4")
if big:
    dd = synth.make_device(name, dev)
    d = {k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
    del dd
else:
    d = synth.make(name)
ds = dataset.from_config("implicit", name, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
g = ds.graph_csr()
N = ds.n_users + ds.n_items
ops = HipOps()
out = {"workload": name, "dim": dim, "n_rows": N, "nnz": int(g.col.numel()), "per_layer_ms": {}}
x_full = None
for W in (() if only_step else (1, 2, 4, 8)):
    lay = RowLayout(N, W, chunks=n_chunks if W > 1 else 1)
    slabs = [ops.make_slab(rp, cl, vl, dev) for rp, cl, vl in build_slab_chunks(g.rowptr, g.col, g.val, 0, lay)]
    x = torch.randn(W * lay.M, dim, device=dev) * 0.1          # the gathered table every rank reads
    ys = [torch.empty(lay.Mc, dim, device=dev) for _ in slabs]  # this rank's chunk outputs
    for s, y in zip(slabs, ys):
        ops.spmm(s, x, y=y)                                      # builds the schedules
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s, y in zip(slabs, ys):
            ops.spmm(s, x, y=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nnz_local = sum(int(s["col"].numel()) for s in slabs)
    ms_tiled = None
    if W > 1 and lay.C > 1:
        tiles = [[ops.make_slab(rp, cl, vl, dev) for rp, cl, vl in row] for row in build_slab_chunks(g.rowptr, g.col, g.val, 0, lay, tiled=True)]

        def tiled_layer():
            for k in range(lay.C):
                for c in range(lay.C):
                    ops.spmm(tiles[c][k], x, add=None if k == 0 else ys[c], y=ys[c])
        tiled_layer()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            tiled_layer()
        e1.record()
        torch.cuda.synchronize()
        ms_tiled = e0.elapsed_time(e1) / reps
        del tiles
    out["per_layer_ms"][str(W)] = {"ms": ms, "ms_tiled": ms_tiled, "local_rows": lay.M, "local_nnz": nnz_local, "chunks": len(slabs),
                                   "allgather_bytes_per_rank_out": lay.M * dim * 4, "allgather_bytes_per_rank_in": (W - 1) * lay.M * dim * 4}
    print(f"W={W}: rank-0 local SpMM per layer {ms:.3f} ms" + (f" (tiled {lay.C} x {lay.C}: {ms_tiled:.3f} ms)" if ms_tiled else "") +
          f"  ({nnz_local} nnz, {lay.M} rows, {len(slabs)} chunk(s)); all-gather receives {(W - 1) * lay.M * dim * 4 / 1e6:.1f} MB per layer", flush=True)
    del slabs, x, ys
    torch.cuda.empty_cache()
# the 2-D tiling (recad_amd/sharded2d.py): rank 0's tile A[R_0, C_0] over x[C_0] (Pr * Mb rows), partial of Pc * Mb rows
from recad_amd.sharded2d import GridLayout, build_tile  # noqa: E402
out["grid2d_per_layer_ms"] = {}
for W, pr in (() if only_step else ((4, 2), (8, 2), (8, 4), (8, 1))):
    lay = GridLayout(N, W, pr)
    tile = ops.make_slab(*build_tile(g.rowptr, g.col, g.val, 0, lay), dev)
    x = torch.randn(lay.Pr * lay.Mb, dim, device=dev) * 0.1
    y = torch.empty(lay.Pc * lay.Mb, dim, device=dev)
    ops.spmm(tile, x, y=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.spmm(tile, x, y=y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    blk = lay.Mb * dim * 4
    out["grid2d_per_layer_ms"][f"{W}:{lay.Pr}x{lay.Pc}"] = {"ms": ms, "tile_nnz": int(tile["col"].numel()), "x_rows": lay.Pr * lay.Mb, "partial_rows": lay.Pc * lay.Mb,
                                                          "bytes_received_per_rank": (lay.Pr - 1 + lay.Pc - 1) * blk}
    print(f"W={W} grid {lay.Pr} x {lay.Pc}: rank-0 tile SpMM per layer {ms:.3f} ms ({int(tile['col'].numel())} nnz, x {lay.Pr * lay.Mb} rows, partial {lay.Pc * lay.Mb} rows); "
          f"receives {(lay.Pr - 1 + lay.Pc - 1) * blk / 1e6:.1f} MB per layer (all-gather {(lay.Pr - 1) * blk / 1e6:.1f} + reduce-scatter {(lay.Pc - 1) * blk / 1e6:.1f})", flush=True)
    del tile, x, y
    torch.cuda.empty_cache()
# a whole train step of rank 0 of a W-rank job with every collective replaced by a local copy of the rank's own share
# (Grid2DLightGCN(probe_rank_world=...)): what the rank computes and launches per step, captured like the real step
from recad_amd.sharded2d import Grid2DLightGCN  # noqa: E402
out["rank0_step_ms"] = {}
B, steps = 1024, 8
gen = torch.Generator(device=dev).manual_seed(5)
tu = torch.randint(0, ds.n_users, (B * (steps + 3),), device=dev, generator=gen)
tp = torch.randint(0, ds.n_items, (B * (steps + 3),), device=dev, generator=gen)
tn = torch.randint(0, ds.n_items, (B * (steps + 3),), device=dev, generator=gen)
ue = torch.randn(ds.n_users, dim, device=dev) * 0.1
ie = torch.randn(ds.n_items, dim, device=dev) * 0.1
for W, pr in ((1, 1), (8, 1), (8, 2)):
    tr = Grid2DLightGCN(ds.n_users, ds.n_items, dim, 3, g, ue, ie, device=dev, grid_rows=pr, probe_rank_world=(0, W))
    tr.train_epoch(tu[: 3 * B], tp[: 3 * B], tn[: 3 * B], B)      # eager first step, capture, one replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_epoch(tu[3 * B:], tp[3 * B:], tn[3 * B:], B)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out["rank0_step_ms"][f"{W}:{tr.layout.Pr}x{tr.layout.Pc}"] = {"ms": ms, "chunks": tr.layout.C, "captured": tr._graph is not None}
    print(f"W={W} grid {tr.layout.Pr} x {tr.layout.Pc}: rank-0 train step WITHOUT communication {ms:.3f} ms (captured: {tr._graph is not None}, {tr.layout.C} chunk(s))", flush=True)
    del tr
    torch.cuda.empty_cache()
if not only_step:
    base = out["per_layer_ms"]["1"]["ms"]
    for W, v in out["per_layer_ms"].items():
        v["compute_speedup"] = base / v["ms"]
print(json.dumps(out))
