"""Row multiplicities inside the minibatches the bench trains on (tuning aid for the ordered scatter)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from recad_amd import dataset, synth
dev = torch.device('cuda:0')
d = synth.make("ml1m")
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
e = ds.generate_epoch()
keys = list(e.keys()); print(keys)
u, p, n = (e[k].cpu().numpy() for k in keys[:3])
U = ds.n_users
mx = []
for s in range(0, len(u) - 1024, 1024):
    r = np.concatenate([u[s:s+1024], U + p[s:s+1024], U + n[s:s+1024]])
    c = np.bincount(r)
    mx.append(c.max())
mx = np.array(mx)
print("steps", len(mx), "max multiplicity per batch: mean %.1f min %d max %d" % (mx.mean(), mx.min(), mx.max()), "hist of last batch", np.bincount(c)[:16])
