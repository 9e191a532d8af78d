"""Does the scoring GEMM of one block of users overlap with the selection kernel of the previous block?  (The GEMM is MFMA /
store-bound with 2 workgroups of 256 VGPRs per CU, the selection is LDS / latency-bound: complementary on paper.)
rk_score_matrix on stream A, rk_topk_rows on stream B, events between them; C = 1 is today's sequence.
    python3 scripts/score_overlap_probe.py [n_users=5893] [n_items=3702] [dim=64]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib

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
L = _lib.lib()
P = _lib.ptr
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
ref = None
for C in (1, 2, 3, 4, 6, 8, 12):
    top_ids = torch.empty(nu, K, dtype=torch.int32, device=dev)
    top_sc = torch.zeros(nu, K, device=dev)
    ts = torch.empty(nu, 1, device=dev)
    tr = torch.empty(nu, 1, dtype=torch.int32, device=dev)
    scores = torch.empty(nu * I, device=dev)
    per = -(-nu // C)
    blocks = [(s, min(nu, s + per)) for s in range(0, nu, per)]
    evs = [torch.cuda.Event() for _ in blocks]

    def run():
        sB.wait_stream(sA)
        for (s, e), ev in zip(blocks, evs):
            _lib.check(L.rk_score_matrix(d, P(utab), e - s, P(ids[s:e]), P(itab), I, None, None, 0.0, 0.0, 0, P(scores[s * I:e * I]), sA.cuda_stream), "gemm")
            ev.record(sA)
            sB.wait_event(ev)
            _lib.check(L.rk_topk_rows(P(scores[s * I:e * I]), e - s, I, P(ids[s:e]), P(sp), P(si), K, P(top_ids[s:e]), P(top_sc[s:e]), P(tg), 1, P(ts[s:e]), P(tr[s:e]),
                                      sB.cuda_stream), "topk")
        sA.wait_stream(sB)
    torch.cuda.synchronize()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(sA)
    for _ in range(reps):
        run()
    e1.record(sA)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    if ref is None:
        ref = (top_ids.clone(), tr.clone())
    same = bool((top_ids == ref[0]).all()) and bool((tr == ref[1]).all())
    print(f"{nu} x {I} x {d}: {C:2d} block(s) of {per} users, GEMM on stream A | selection on stream B: {us:7.1f} us per evaluation (identical lists and ranks: {same})", flush=True)
