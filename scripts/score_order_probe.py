"""Probe (tuning only): rk_score_topk when the scores are ORDERED by item id (an item bias that grows / falls / steps with the
id on top of random embeddings) -- the case in which the panel form's bound from earlier panels is too low for a later one.
    python3 scripts/score_order_probe.py <random|descending|blocky|ascending> [n_users=16384] [n_items=34474] [dim=64]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib
from recad_amd.evaluate import score_plan

kind = sys.argv[1]
nu, I, d = (int(sys.argv[k]) if len(sys.argv) > k else v for k, v in ((2, 16384), (3, 34474), (4, 64)))
K = 100
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
utab = torch.randn(nu, d, device=dev, generator=g) * 0.1
itab = torch.randn(I, d, device=dev, generator=g) * 0.1
ub = torch.zeros(nu, device=dev)
ar = torch.arange(I, device=dev, dtype=torch.float32)
ib = {"ascending": ar * 1e-4, "descending": -ar * 1e-4, "random": torch.zeros(I, device=dev), "blocky": ((ar // 1920) % 3) * 0.5}[kind]
ptr = torch.zeros(nu + 1, dtype=torch.int32, device=dev)
idx = torch.zeros(1, dtype=torch.int32, device=dev)
ids = torch.arange(nu, dtype=torch.int32, device=dev)
tg = torch.tensor([0], dtype=torch.int32, device=dev)
out = {}
for mode in ("panel", "unfused"):
    chunk = nu if mode == "panel" else 8192
    plan = score_plan(chunk, I, d, K, 1, {"path": "panel" if mode == "panel" else "gemm"})
    top_ids = torch.empty(nu, K, dtype=torch.int32, device=dev)
    top_sc = torch.empty(nu, K, device=dev)
    ts = torch.empty(nu, 1, device=dev)
    tr = torch.empty(nu, 1, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(plan.scratch_floats) + 64, device=dev)

    def once():
        for s in range(0, nu, chunk):
            e = min(nu, s + chunk)
            _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(utab), e - s, _lib.ptr(ids[s:e]), _lib.ptr(itab), I, _lib.ptr(ub), _lib.ptr(ib), 0.0, _lib.ptr(ptr),
                                                _lib.ptr(idx), K, _lib.ptr(top_ids[s:e]), _lib.ptr(top_sc[s:e]), _lib.ptr(tg), 1, _lib.ptr(ts[s:e]),
                                                _lib.ptr(tr[s:e]), C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()), "rk_score_topk")
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    once()
    once()
    e1.record()
    torch.cuda.synchronize()
    out[mode] = top_ids.clone()
    print(f"{kind:10s} {mode:8s} {nu} x {I} x {d}: {e0.elapsed_time(e1) / 2 * 1e3:10.1f} us", flush=True)
print("identical lists:", bool(torch.equal(out["panel"], out["unfused"])))
