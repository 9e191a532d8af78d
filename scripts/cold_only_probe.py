"""Probe for DESIGN section 7 item 1: SpMM time on the ml1m train graph with the edges of the H most popular items removed
(what the gather kernel would be left with if a dense bit-panel product covered those edges)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import synth, _lib
from recad_amd.graph import CsrGraph
dev = torch.device('cuda:0')
d = synth.make("ml1m")
ptr, idx = d["train"][0].astype(np.int64), d["train"][1].astype(np.int64)
U, I = len(ptr) - 1, int(idx.max()) + 1
deg_i = np.bincount(idx, minlength=I)
rows = np.repeat(np.arange(U), np.diff(ptr))
for H in (0, 512, 1024, 2048):
    hot = np.zeros(I, dtype=bool)
    if H: hot[np.argsort(-deg_i)[:H]] = True
    keep = ~hot[idx]
    r, c = rows[keep], idx[keep]
    p2 = np.zeros(U + 1, dtype=np.int32); np.add.at(p2, r + 1, 1); p2 = np.cumsum(p2).astype(np.int32)
    g = CsrGraph.from_user_item_csr(U, I, p2, c.astype(np.int32), dev)
    x = torch.randn(U + I, 64, device=dev)
    for _ in range(5): g.spmm(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.spmm(x)
    e1.record(); torch.cuda.synchronize()
    print(f"H={H}: nnz {g.nnz} ({g.nnz / (2 * len(idx)):.2f} of all) spmm {e0.elapsed_time(e1) / 200 * 1e3:.2f} us (includes torch output alloc)")
