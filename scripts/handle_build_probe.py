"""Where the host time of a LightGCN handle build goes (the perturb-retrain loop builds one per injected graph): graph CSR,
LDS plan (device -> host copy of the graph, host build, upload), workspace allocations, rk_lightgcn_create.
    python3 scripts/handle_build_probe.py [workload=ml1m]"""
import sys
import time

import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model, synth

name = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
dev = torch.device("cuda:0")
d = synth.make(name)


def t_sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for rep in range(4):
    ds = dataset.from_config("implicit", name, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
    t0 = t_sync()
    g = ds.graph_csr()
    t1 = t_sync()
    plan = g.lds_plan(64)
    t2 = t_sync()
    torch.manual_seed(1)
    v = model.from_config("victim", "lightgcn", latent_dim_rec=64, lightGCN_n_layers=3).I(dataset=ds).to(dev)
    t3 = t_sync()
    v._ensure_handle()
    t4 = t_sync()
    print(f"{name} build {rep}: graph_csr {1e3 * (t1 - t0):.1f} ms | lds_plan (copy + host build + upload) {1e3 * (t2 - t1):.1f} ms | victim instantiate {1e3 * (t3 - t2):.1f} ms | "
          f"_ensure_handle with the plan cached (workspace + create) {1e3 * (t4 - t3):.1f} ms", flush=True)
    del v, g, ds
