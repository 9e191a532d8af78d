import sys, os, torch, numpy as np
sys.path.insert(0, '.')
from recad_amd import synth, dataset, model, _lib
dev = torch.device('cuda:0')
which = sys.argv[1]
d = synth.make("ml1m")
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source=which)
m = model.from_config("victim", "lightgcn", latent_dim_rec=64).I(dataset=ds).to(dev)
h = m._ensure_handle()
for _ in range(20): _lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): _lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr())
e1.record(); torch.cuda.synchronize()
print(which, "variant", os.environ.get("RK_SPMM_VARIANT", "0"), "seg", os.environ.get("RK_SEG_NNZ", "64"), "blocks", ds.graph_csr().schedule(64)[1], "us/spmm %.2f" % (e0.elapsed_time(e1) * 1000 / 900))
