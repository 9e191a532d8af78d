"""Probe (tuning only): SpMM time on a big graph as ONE launch vs P column-panel launches chained through the
`add` epilogue (y += A[:, panel] x), and with user-column / item-column panels run one after the other.
    python3 scripts/spmm_panel_probe.py <workload> <dim> [reps]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib, synth
from recad_amd.graph import CsrGraph

name, dim = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda:0')
d = synth.make_device(name, dev) if name in ("c4s", "config4") else synth.make(name)
U, I = d["n_users"], d["n_items"]
ptr, idx = d["train"]
g = CsrGraph.from_user_item_csr(U, I, torch.as_tensor(ptr), torch.as_tensor(idx), dev)
N, nnz = g.n_rows, g.nnz
print(name, "N", N, "nnz", nnz, "dim", dim, flush=True)
x = torch.randn(N, dim, device=dev)
rows = torch.repeat_interleave(torch.arange(N, device=dev), (g.rowptr[1:] - g.rowptr[:-1]).long())


def sub_graph(mask):
    cnt = torch.bincount(rows[mask], minlength=N)
    rp = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    rp[1:] = torch.cumsum(cnt, 0).to(torch.int32)
    return CsrGraph(N, rp, g.col[mask].contiguous(), g.val[mask].contiguous(), g.class_split)


def run(graphs, label):
    y = torch.empty(N, dim, device=dev)
    scheds = [gg.schedule(dim) + (gg.new_scratch(dim),) for gg in graphs]

    def once():
        for k, (gg, (wd, nb, sc)) in enumerate(zip(graphs, scheds)):
            e = _lib.SpmmEpilogue(add=_lib.ptr(y) if k else None, y=_lib.ptr(y), sum_scale=1.0)
            _lib.check(_lib.lib().rk_spmm_csr_ex(N, _lib.ptr(gg.rowptr), _lib.ptr(gg.col), _lib.ptr(gg.val), _lib.ptr(wd), nb, _lib.ptr(sc), dim,
                                                 _lib.ptr(x), N, C.byref(e), _lib.stream_ptr()), "spmm")
    once(); once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 8 * nnz + 4 * (N + 1) + 8 * N * dim
    print(f"{label:34s} {ms:9.3f} ms   algorithmic {alg / ms / 1e6:8.1f} GB/s   gather {(4 * nnz * dim) / ms / 1e6:8.1f} GB/s", flush=True)
    return y.clone()


ref = run([g], "one launch")
col = g.col.long()
# user columns (item rows gather them) then item columns (user rows gather them): one table in the caches at a time
ys = run([sub_graph(col < U), sub_graph(col >= U)], "2 panels: user cols | item cols")
print("   max |diff| vs one launch", float((ys - ref).abs().max()))
for pu, pi in ((2, 1), (4, 1), (4, 2), (8, 2)):
    bounds = [U * k // pu for k in range(pu + 1)] + [U + I * k // pi for k in range(1, pi + 1)]
    gs = [sub_graph((col >= bounds[k]) & (col < bounds[k + 1])) for k in range(len(bounds) - 1)]
    ys = run(gs, f"{pu} user-col panels + {pi} item-col")
    print("   max |diff| vs one launch", float((ys - ref).abs().max()))
    del gs
