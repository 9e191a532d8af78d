"""Diagnostic: where a workgroup of the register-resident panel form (score_panel.h) spends its time (RK_PAN_STAMPS=1).
    python3 scripts/pan_stamps.py [n_users=5893] [n_items=3702] [dim=64]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
os.environ["RK_PAN_STAMPS"] = "1"
from recad_amd import _lib
from recad_amd.evaluate import score_plan

nu, I, d = (int(sys.argv[k]) if len(sys.argv) > k else v for k, v in ((1, 5893), (2, 3702), (3, 64)))
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
tg = torch.tensor([0], dtype=torch.int32, device=dev)
K = 100
top_ids = torch.empty(nu, K, dtype=torch.int32, device=dev)
top_sc = torch.zeros(nu, K, device=dev)
ts = torch.empty(nu, 1, device=dev)
tr = torch.empty(nu, 1, dtype=torch.int32, device=dev)
rows = int(os.environ.get("PROBE_ROWS", 32 if nu >= 4096 else 16))
plan = score_plan(nu, I, d, K, 1, {"path": "panel", "panel_rows": rows})   # (stamps: tuning build + RK_PAN_STAMPS=1)
need = int(plan.scratch_floats)
n_wg = (nu + rows - 1) // rows
scratch = torch.zeros(need + n_wg * 72 + 64, device=dev)
for _ in range(3):
    _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(utab), nu, _lib.ptr(ids), _lib.ptr(itab), I, None, None, 0.0, _lib.ptr(sp), _lib.ptr(si), K, _lib.ptr(top_ids),
                                        _lib.ptr(top_sc), _lib.ptr(tg), 1, _lib.ptr(ts), _lib.ptr(tr), C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()), "x")
torch.cuda.synchronize()
off = ((scratch.data_ptr() + need * 4 + 63) & ~63) - scratch.data_ptr()
st = scratch.view(torch.uint8)[off: off + n_wg * 288].view(torch.int64).view(n_wg, 36).cpu().numpy().astype(np.float64)
# [0] start, [1] after the prologue, [2 + 8 p + k] end of phase k of panel p (p < 4), [34] ranks, [35] end
phases = ["mfma + loads", "bitmap + barrier", "pass 1 (mask, counts, maxima)", "bound (first panel)", "collect + overflow check", "tau / prune + barrier"]
n_pan = min(4, (I + 1919) // 1920 if I > 1024 else 1)
tot = st[:, 35] - st[:, 0]
print(f"workgroups {n_wg} of {rows} rows; median workgroup lifetime {np.median(tot) / 100:.2f} us, kernel span {(st[:, 35].max() - st[:, 0].min()) / 100:.2f} us")
print(f"  {'prologue':32s} median {np.median(st[:, 1] - st[:, 0]) / 100:7.2f} us")
prev = st[:, 1]
for p_ in range(n_pan):
    for k, nm in enumerate(phases):
        cur = st[:, 2 + 8 * p_ + k]
        print(f"  panel {p_} {nm:32s} median {np.median(cur - prev) / 100:7.2f} us")
        prev = cur
print(f"  {'ranks':32s} median {np.median(st[:, 34] - prev) / 100:7.2f} us   (after panel {n_pan - 1}; later panels of a longer sweep are in here)")
print(f"  {'sort + output':32s} median {np.median(st[:, 35] - st[:, 34]) / 100:7.2f} us")
print(f"  start spread {(st[:, 0].max() - st[:, 0].min()) / 100:.2f} us, end spread {(st[:, 35].max() - st[:, 35].min()) / 100:.2f} us")
print(f"  candidates held at the end: mean per row {st[:, 32].mean() / rows:.1f}, largest row {st[:, 33].max():.0f}")
