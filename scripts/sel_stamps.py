"""Diagnostic (RK_SEL_STAMPS build only): where a workgroup of the fused sweep spends its cycles."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from recad_amd import _lib
nu, I, d = 5893, 3702, 64
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
utab = torch.randn(nu, d, device=dev, generator=g) * 0.1
itab = torch.randn(I, d, device=dev, generator=g) * 0.1
rng = np.random.default_rng(0)
deg = rng.integers(10, 150, nu)
ptr = np.zeros(nu + 1, dtype=np.int32); ptr[1:] = np.cumsum(deg)
idx = np.concatenate([np.sort(rng.choice(I, size=k, replace=False)) for k in deg]).astype(np.int32)
ids = torch.arange(nu, dtype=torch.int32, device=dev)
sp, si = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
tg = torch.tensor([0], dtype=torch.int32, device=dev)
K = 100
top_ids = torch.empty(nu, K, dtype=torch.int32, device=dev)
top_sc = torch.zeros(nu, K, device=dev)
ts = torch.empty(nu, 1, device=dev); tr = torch.empty(nu, 1, dtype=torch.int32, device=dev)
need = int(_lib.lib().rk_score_topk_scratch_floats(nu, I, d, K, 1))
scratch = torch.zeros(need + 32768, device=dev)
for _ in range(3):
    _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(utab), nu, _lib.ptr(ids), _lib.ptr(itab), I, None, None, 0.0, _lib.ptr(sp), _lib.ptr(si), K, _lib.ptr(top_ids),
                                        _lib.ptr(top_sc), _lib.ptr(tg), 1, _lib.ptr(ts), _lib.ptr(tr), _lib.ptr(scratch), _lib.stream_ptr()), "x")
torch.cuda.synchronize()
n_wg = (nu + 15) // 16
off = ((scratch.data_ptr() + need * 4 + 63) & ~63) - scratch.data_ptr()
st = scratch.view(torch.uint8)[off: off + n_wg * 64].view(torch.int64).view(n_wg, 8).cpu().numpy().astype(np.float64)
names = ["mfma+loads", "store_b", "chunk barrier", "epilogue", "epi barrier", "check/compact"]
tot = st[:, 7] - st[:, 6]
print("workgroups", n_wg, "median loop cycles (s_memtime ticks = shader cycles / 100MHz units?)", np.median(tot))
for k, nme in enumerate(names):
    print(f"  {nme:16s} median {np.median(st[:, k]):12.0f}  share {np.median(st[:, k] / tot):6.3f}")
print("  start spread", st[:, 6].max() - st[:, 6].min(), "end spread", st[:, 7].max() - st[:, 7].min())
