"""SpMM launches of one workload for profiling (kernel-trace / PMC passes): scripts/spmm_sweep.py <graph> [workload] [dim] [reps]"""
import sys, os, torch, numpy as np
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import synth, dataset, model, _lib
dev = torch.device('cuda:0')
which = sys.argv[1]
work = sys.argv[2] if len(sys.argv) > 2 else "ml1m"
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
if work in ("c4s", "config4"):
    dd = synth.make_device(work, dev)
    d = {k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
    del dd
else:
    d = synth.make(work)
ds = dataset.from_config("implicit", work, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source=which)
m = model.from_config("victim", "lightgcn", latent_dim_rec=dim).I(dataset=ds).to(dev)
h = m._ensure_handle()
for _ in range(3): _lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): _lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr())
e1.record(); torch.cuda.synchronize()
g = ds.graph_csr()
print(work, which, "N", g.n_rows, "nnz", g.nnz, "dim", dim, "blocks", g.schedule(dim)[1] & 0x7ffffff, "us/spmm %.2f" % (e0.elapsed_time(e1) * 1000 / (3 * reps)))
