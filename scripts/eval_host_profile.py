"""Host-side cost of one full evaluation (ml1m-shaped LightGCN): cProfile of full_catalog_topk."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model, synth
from recad_amd.evaluate import eligible_users, full_catalog_topk, hit_counts
dev = torch.device('cuda:0')
d = synth.make("ml1m")
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
victim = model.from_config("victim", "lightgcn", latent_dim_rec=64).I(dataset=ds).to(dev)
ptr, idx = ds.train_csr_sorted()
targets = np.array([0], dtype=np.int32)
ev = eligible_users(ptr, idx, targets)
t = lambda a: torch.as_tensor(a, dtype=torch.int32, device=dev)
ev_d, ptr_d, idx_d, tg_d = t(ev), t(ptr), t(idx), t(targets)
def once():
    res = full_catalog_topk(victim, ev_d, ptr_d, idx_d, tg_d, K=100, chunk=8192, to_host=False)
    return hit_counts(res["target_rank"], (10, 20, 50, 100))
for _ in range(3): once()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); h = once(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e6*(t1-t0):.0f} us, total {1e6*(t2-t0):.0f} us")
t0 = time.perf_counter()
for _ in range(20): h = once()
torch.cuda.synchronize()
print(f"20 back-to-back: {1e6*(time.perf_counter()-t0)/20:.0f} us per evaluation")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): once()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
