"""Probe (tuning only): rk_score_topk per-call time, panel form (PROBE_ROWS=16|32 forces its workgroup shape) vs GEMM + selection, on random tables (PROBE_MODES=panel,unfused).
    python3 scripts/score_probe.py <n_users> <n_items> <dim> [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib
from recad_amd.evaluate import score_plan

nu, I, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
utab = torch.randn(nu, d, device=dev, generator=g) * 0.1
itab = torch.randn(I, d, device=dev, generator=g) * 0.1
rng = np.random.default_rng(0)
deg = rng.integers(10, 150, nu)
ptr = np.zeros(nu + 1, dtype=np.int32)
ptr[1:] = np.cumsum(deg)
idx = np.concatenate([np.sort(rng.choice(I, size=k, replace=False)) for k in deg]).astype(np.int32)
ids = torch.arange(nu, dtype=torch.int32, device=dev)
sp, si = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
NT = int(os.environ.get("PROBE_TARGETS", "1"))   # target items per user block (the panel form's multi-target instantiation from 2 on)
tg = torch.tensor([(7 * t * t + 3 * t) % I for t in range(NT)], dtype=torch.int32, device=dev)
K = 100
top_ids = torch.empty(nu, K, dtype=torch.int32, device=dev)
top_sc = torch.empty(nu, K, device=dev)
ts = torch.empty(nu, NT, device=dev)
tr = torch.empty(nu, NT, dtype=torch.int32, device=dev)
out = {}
modes = os.environ.get("PROBE_MODES", "panel,unfused").split(",")
for mode in modes:
    chunk = nu if mode != "unfused" else max(256, min(8192, (1 << 31) // I))
    req = {"path": "gemm"} if mode == "unfused" else dict({"path": "panel"}, **({"panel_rows": int(os.environ["PROBE_ROWS"])} if os.environ.get("PROBE_ROWS") else {}))
    plan = score_plan(min(chunk, nu), I, d, K, NT, req)
    scratch = torch.empty(int(plan.scratch_floats) + 16384, device=dev)

    def once():
        for s in range(0, nu, chunk):
            e = min(nu, s + chunk)
            _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(utab), e - s, _lib.ptr(ids[s:e]), _lib.ptr(itab), I, None, None, 0.0, _lib.ptr(sp), _lib.ptr(si),
                                                K, _lib.ptr(top_ids[s:e]), _lib.ptr(top_sc[s:e]), _lib.ptr(tg), NT, _lib.ptr(ts[s:e]), _lib.ptr(tr[s:e]),
                                                C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()), "rk_score_topk")
    once(); once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[mode] = (top_ids.clone(), tr.clone())
    print(f"{mode:8s} {nu} x {I} x {d}: {ms * 1e3:9.1f} us per evaluation   {2.0 * nu * I * d / ms / 1e9:7.1f} TFLOP/s", flush=True)
for m in modes[1:]:
    print(f"{modes[0]} vs {m}: identical lists:", bool(torch.equal(out[modes[0]][0], out[m][0])), "identical ranks:", bool(torch.equal(out[modes[0]][1], out[m][1])))
