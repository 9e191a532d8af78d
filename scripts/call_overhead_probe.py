"""Where the driver-style 20-step call spends its host time: enqueue of victim._run_epoch (Python + ctypes + hipGraphLaunch), GPU
span by events, completion detection.  usage: python scripts/call_overhead_probe.py [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
import bench
from recad_amd import dataset, model, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
d = synth.make("ml1m")
B = 1024
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True, device=dev, graph_source="train", pairwise_batch_size=B, seed=1234)
torch.manual_seed(2023)
v = model.from_config("victim", "lightgcn", latent_dim_rec=64, lightGCN_n_layers=3).I(dataset=ds).to(dev)
trip = bench.resident_triplets(ds, (steps + 5) * B)
v.reserve(steps * B, B)
stream = torch.cuda.current_stream()
# time the C call inside victim._run_epoch (state_init launch + hipGraphLaunch) apart from the Python around it
from recad_amd import _lib
_real = _lib.lib().rk_lightgcn_train_epoch
c_us = []
def _timed(*a):
    t = time.perf_counter(); rc = _real(*a); c_us.append((time.perf_counter() - t) * 1e6); return rc
class _L:
    def __getattr__(self, k): return _timed if k == "rk_lightgcn_train_epoch" else getattr(_lib._lib if hasattr(_lib, "_lib") and _lib._lib is not None else _lib.lib(), k)
_orig_lib = _lib.lib
_lib.lib = lambda: _L()
first = os.environ.get("PROBE_FIRST") == "1"   # time the FIRST launches of the reserved 20-step graph (what bench.py's timed call is)
if first:
    bench.run_steps(v, trip, B, 0, 5)      # the bench's warm-up: a 5-step call (its own whole-call graph)
else:
    for _ in range(3):
        bench.run_steps(v, trip, B, 0, steps)
torch.cuda.synchronize()
rows = []
prewarm_ms = float(os.environ.get("PROBE_PREWARM_MS", "0"))   # device kept busy (5-step calls back to back) right up to each timed call
idle_ms = float(os.environ.get("PROBE_IDLE_MS", "0"))         # host sleep in front of each timed call (device idle)
for rep in range(12):
    if prewarm_ms > 0:
        for _ in range(int(prewarm_ms * 1000 / (5 * 66)) + 1):
            bench.run_steps(v, trip, B, 0, 5)
    torch.cuda.synchronize()
    if idle_ms > 0:
        time.sleep(idle_ms / 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    t1 = time.perf_counter()
    bench.run_steps(v, trip, B, 5, steps)
    t2 = time.perf_counter()
    e1.record(stream)
    bench.wait_done(stream)
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6, e0.elapsed_time(e1) * 1e3))
print("C call rk_lightgcn_train_epoch (us) per call:", " ".join("%.0f" % x for x in c_us[-14:]))
if first:
    for i, r in enumerate(rows[:12]):
        print("call %d after reserve: _run_epoch returns after %.1f us | total %.1f us = %.2f us/step | GPU span %.1f us" % (i, r[1], r[3], r[3] / steps, r[4]))
a = np.median(np.array(rows[2:]), axis=0)
print("steps %d: event record %.1f us | _run_epoch returns after %.1f us | completion seen %.1f us later | total %.1f us = %.2f us/step | GPU span by events %.1f us" % (steps, *a[:4], a[3] / steps, a[4]))
