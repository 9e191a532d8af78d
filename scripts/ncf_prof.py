import sys, numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model
from tests import _golden as G
dev = torch.device('cuda:0')
g = G.load("lightgcn_game_d64")
ds = dataset.from_config("implicit", "game", train_csr=(g["train_ptr"].astype(np.int64), g["train_idx"]), test_csr=(g["test_ptr"].astype(np.int64), g["test_idx"]),
                         need_graph=False, device=dev, sample="pointwise", seed=1)
f, L = int(sys.argv[1]), int(sys.argv[2])
m = model.from_config("victim", "ncf", factor_num=f, num_layers=L).I(dataset=ds).to(dev)
ep = ds.generate_epoch()
u, i, l = (ep[k][:1024 * 40] for k in ("users", "items", "labels"))
m._run_epoch(u, i, l, 1024); torch.cuda.synchronize()
