"""EvalSession (one hipGraph replay per evaluation) under rocprofv3 --kernel-trace: 12 evaluations back to back.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/eval_session_probe.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
import bench
from recad_amd import dataset, model, synth
from recad_amd.evaluate import EvalSession, eligible_users
dev = torch.device('cuda:0')
d = synth.make("ml1m")
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
victim = model.from_config("victim", "lightgcn", latent_dim_rec=64).I(dataset=ds).to(dev)
ptr, idx = ds.train_csr_sorted()
targets = np.array([0], dtype=np.int32)
ev = eligible_users(ptr, idx, targets)
t = lambda a: torch.as_tensor(a, dtype=torch.int32, device=dev)
sess = EvalSession(victim, t(ev), t(ptr), t(idx), t(targets), K=100, topks=(10, 20, 50, 100))
for _ in range(4):
    sess.run()
torch.cuda.synchronize()
stream = torch.cuda.current_stream()
for reps in (1, 8, 12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sess.run()
    t1 = time.perf_counter()
    bench.wait_done(stream)
    t2 = time.perf_counter()
    print(f"{reps} evaluations back to back: host enqueue {1e6 * (t1 - t0) / reps:.1f} us each, {1e6 * (t2 - t0) / reps:.1f} us per evaluation", flush=True)
