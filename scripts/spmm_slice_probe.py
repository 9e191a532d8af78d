"""Review item 8, measured first: does the row-gather SpMM get faster when its source table is cut into column slices that
fit an XCD's 4 MiB L2?  Y[:, s] = A . X[:, s] holds per slice, so the d-wide launch becomes d/D launches over slice-major
tables [d/D][N][D] -- every launch gathers D*4-byte pieces out of an N*D*4-byte table (yelp shape, d = 128: 45.6 MB as one
table; 22.8 / 11.4 MB as D = 64 / 32 slices).  Uses the product kernel as it is (spmm_csr_kernel<D>), nothing else changes.
    python3 scripts/spmm_slice_probe.py [workload=yelp] [dim=128] [reps=20]"""
import json
import sys

import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, synth
from recad_amd.sharded import HipOps

name = sys.argv[1] if len(sys.argv) > 1 else "yelp"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
if name in ("c4s", "config4"):
    dd = synth.make_device(name, dev)
    d = {k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
    del dd
else:
    d = synth.make(name)
ds = dataset.from_config("implicit", name, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
g = ds.graph_csr()
N = ds.n_users + ds.n_items
ops = HipOps()
slab = ops.make_slab(g.rowptr, g.col, g.val, dev)
x = torch.randn(N, dim, device=dev) * 0.1
out = {"workload": name, "dim": dim, "n_rows": N, "nnz": int(g.col.numel()), "us": {}}


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


y = torch.empty_like(x)
ref_us = timed(lambda: ops.spmm(slab, x, y=y))
out["us"][str(dim)] = ref_us
print(f"{name} d={dim}: one launch, {dim * 4}-byte rows out of a {N * dim * 4 / 1e6:.1f} MB table: {ref_us:.1f} us", flush=True)
for D in (64, 32, 16):
    if D >= dim or D * 4 < dim:
        continue
    ns = dim // D
    xs = x.view(N, ns, D).permute(1, 0, 2).contiguous()          # [ns][N][D]
    ys = torch.empty_like(xs)

    def sliced():
        for s in range(ns):
            ops.spmm(slab, xs[s], y=ys[s])
    us = timed(sliced)
    err = float((ys.permute(1, 0, 2).reshape(N, dim) - y).abs().max())
    out["us"][f"{ns}x{D}"] = us
    print(f"{name} d={dim}: {ns} launches over slice tables of {N * D * 4 / 1e6:.1f} MB ({D * 4}-byte pieces): {us:.1f} us = {ref_us / us:.2f} x the single launch"
          f" (max |difference| {err:.1e})", flush=True)
print(json.dumps(out))

# ---- the dedicated slice kernel (scripts/spmm_slice_probe.hip): lane groups own rows, one launch walks the slices in order
import ctypes  # noqa: E402
import os  # noqa: E402
import subprocess  # noqa: E402

import numpy as np  # noqa: E402

so = os.path.join("recad_amd", "lib", "libspmm_slice_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-munsafe-fp-atomics", "scripts/spmm_slice_probe.hip", "-o", so])
lib = ctypes.CDLL(so)
lib.slice_spmm.restype = ctypes.c_int
lib.slice_spmm.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
rp = g.rowptr.cpu().numpy().astype(np.int64)
U = ds.n_users


def build_pieces(cap):
    """{row, e_begin, e_end, split} per piece: user rows first, then item rows, each by length descending"""
    out_p, split_rows = [], []
    for lo, hi in ((0, U), (U, N)):
        rows = np.arange(lo, hi)
        deg = rp[rows + 1] - rp[rows]
        npc = np.maximum(1, -(-deg // cap))
        r = np.repeat(rows, npc)
        k = np.concatenate([np.arange(n) for n in npc]) if len(npc) else np.zeros(0, np.int64)
        e0 = rp[r] + k * cap
        e1 = np.minimum(e0 + cap, rp[r + 1])
        sp = (np.repeat(npc, npc) > 1).astype(np.int64)
        order = np.argsort(-(e1 - e0), kind="stable")
        out_p.append(np.stack([r, e0, e1, sp], 1)[order])
        split_rows.append(rows[npc > 1])
    return np.concatenate(out_p).astype(np.int32), np.concatenate(split_rows).astype(np.int32)


stream = torch.cuda.current_stream().cuda_stream
for D, un, cap in ((16, 2, 256), (16, 4, 256), (16, 1, 256), (16, 2, 64), (8, 4, 256), (8, 2, 256), (32, 1, 256), (32, 2, 256), (64, 1, 256)):
    if D >= dim or (dim < 128 and D * 4 < dim):
        continue
    ns = dim // D
    pieces, split_rows = build_pieces(cap)
    pt = torch.from_numpy(pieces).to(dev).contiguous()
    st = torch.from_numpy(split_rows if len(split_rows) else np.zeros(1, np.int32)).to(dev)
    xs = x.view(N, ns, D).permute(1, 0, 2).contiguous()
    ys = torch.full_like(xs, float("nan"))

    def run():
        rc = lib.slice_spmm(D, un, pt.data_ptr(), pieces.shape[0], st.data_ptr(), len(split_rows), g.col.data_ptr(), g.val.data_ptr(), N, ns,
                            xs.data_ptr(), ys.data_ptr(), stream)
        assert rc == 0, rc
    us = timed(run)
    err = float((ys.permute(1, 0, 2).reshape(N, dim) - y).abs().max())
    out["us"][f"slice_kernel_D{D}_un{un}_cap{cap}"] = us
    print(f"{name} d={dim}: slice kernel, ONE launch over {ns} slice tables of {N * D * 4 / 1e6:.1f} MB ({D * 4}-byte pieces, {un * D // 4} gathers in flight per lane, "
          f"rows cut at {cap}: {pieces.shape[0]} pieces, {len(split_rows)} split rows): {us:.1f} us = {ref_us / us:.2f} x the product launch (max |difference| {err:.1e})", flush=True)
print(json.dumps(out))
