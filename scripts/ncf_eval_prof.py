"""NCF full-catalog evaluation only (for rocprofv3 --stats): scripts/ncf_eval_prof.py <factor> <layers>"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model
from recad_amd.evaluate import eligible_users, full_catalog_topk
from tests import _golden as G
f, L = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda:0')
g = G.load("lightgcn_game_d64")
ds = dataset.from_config("implicit", "game", train_csr=(g["train_ptr"].astype(np.int64), g["train_idx"]), test_csr=(g["test_ptr"].astype(np.int64), g["test_idx"]),
                         need_graph=False, device=dev, sample="pointwise", seed=1)
m = model.from_config("victim", "ncf", factor_num=f, num_layers=L).I(dataset=ds).to(dev)
ptr, idx = ds.train_csr_sorted()
ev = eligible_users(ptr, idx, [0])[:1024]
full_catalog_topk(m, ev, ptr, idx, [0], K=100, chunk=512)
torch.cuda.synchronize()
t = time.perf_counter()
full_catalog_topk(m, ev, ptr, idx, [0], K=100, chunk=512)
torch.cuda.synchronize()
print("users/s", len(ev) / (time.perf_counter() - t))
