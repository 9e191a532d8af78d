"""Times the scoring GEMM + top-K selection kernels (rk_score_topk) on synthetic tables."""
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib
from recad_amd.evaluate import score_plan
dev = torch.device('cuda:0')
nb, I, d, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 100
g = torch.Generator(device=dev); g.manual_seed(0)
u = torch.randn(nb, d, device=dev, generator=g); it = torch.randn(I, d, device=dev, generator=g)
ids = torch.arange(nb, dtype=torch.int32, device=dev)
seen_ptr = torch.arange(0, (nb + 1) * 50, 50, dtype=torch.int32, device=dev)
seen_idx = torch.randint(0, I, (nb * 50,), dtype=torch.int32, device=dev, generator=g).view(nb, 50).sort(1).values.reshape(-1).contiguous()
tg = torch.zeros(1, dtype=torch.int32, device=dev)
top_ids = torch.empty(nb, K, dtype=torch.int32, device=dev); top_sc = torch.empty(nb, K, device=dev)
ts = torch.empty(nb, 1, device=dev); tr = torch.empty(nb, 1, dtype=torch.int32, device=dev)
plan = score_plan(nb, I, d, K, 1, {"path": "gemm"})
scratch = torch.empty(int(plan.scratch_floats), device=dev)
def run():
    _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(u), nb, _lib.ptr(ids), _lib.ptr(it), I, None, None, 0.0, _lib.ptr(seen_ptr), _lib.ptr(seen_idx), K,
                                        _lib.ptr(top_ids), _lib.ptr(top_sc), _lib.ptr(tg), 1, _lib.ptr(ts), _lib.ptr(tr), C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()), "x")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"nb={nb} I={I} d={d}: {ms:.3f} ms per call, {2.0*nb*I*d/ms/1e9:.1f} TFLOP/s end-to-end (gemm+select)")
