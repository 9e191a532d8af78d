"""Where the Python in front of (and behind) rk_lightgcn_train_epoch goes: cProfile over 400 five-step calls of the bench's run_steps.
usage: python scripts/host_path_profile.py"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
import bench
from recad_amd import dataset, model, synth
dev = torch.device("cuda:0")
d = synth.make("ml1m")
B = 1024
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True, device=dev, graph_source="train", pairwise_batch_size=B, seed=1234)
torch.manual_seed(2023)
v = model.from_config("victim", "lightgcn", latent_dim_rec=64, lightGCN_n_layers=3).I(dataset=ds).to(dev)
trip = bench.resident_triplets(ds, 30 * B)
v.reserve(20 * B, B)
for _ in range(5):
    bench.run_steps(v, trip, B, 0, 5)
torch.cuda.synchronize()
n = 400
t = time.perf_counter()
pre = post = 0.0
for _ in range(n):
    t0 = time.perf_counter()
    bench.run_steps(v, trip, B, 0, 5)
    t1 = time.perf_counter()
    tc = v.last_call_seconds
    pre += tc[2] - t0; post += t1 - tc[3]
    torch.cuda.synchronize()
print("per call: Python in front of the C call %.1f us, behind it %.1f us" % (pre / n * 1e6, post / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    bench.run_steps(v, trip, B, 0, 5)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(28)
