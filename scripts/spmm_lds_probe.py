"""A/B of the LDS-resident sliced SpMM (rk_spmm_lds) against the row-gather kernel (rk_spmm_csr) on one graph:
max difference, and microseconds per launch with HIP events around back-to-back launches.
usage: python scripts/spmm_lds_probe.py [ml1m|tiny|game-shaped U I E] [--dim 64] [--iters 300]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib, synth  # noqa: E402
from recad_amd.graph import CsrGraph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml1m")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--no-stamps", action="store_true", help="timed launches only (what bench.py's live PMC passes run)")
    ap.add_argument("--lds-only", action="store_true", help="skip the row-gather kernel's launches")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    data = synth.make(a.shape)
    U, I = data["n_users"], data["n_items"]
    g = CsrGraph.from_user_item_csr(U, I, data["train"][0], data["train"][1], dev)
    N, d = U + I, a.dim
    torch.manual_seed(0)
    x = torch.randn(N, d, device=dev)
    add = torch.randn(N, d, device=dev)
    y_ref = g.spmm(x, add)
    got = g.lds_plan(d)
    out = {"shape": a.shape, "dim": d, "nnz": g.nnz}
    if got is None:
        print(json.dumps({**out, "lds": None}))
        return
    plan, info = got
    y = g.spmm_lds(x, add)
    y2 = g.spmm_lds(x, add)
    err = (y - y_ref).abs().max().item() / y_ref.abs().max().item()
    out.update(rel_err=err, bit_reproducible=bool(torch.equal(y, y2)), n_wg=info.n_wg, lds_bytes=info.lds_bytes,
               lpa=info.lpa, lpb=info.lpb, chunk=[info.chunk & 0xffff, info.chunk >> 16], plan_mb=plan.numel() * 4 / 1e6)
    L = _lib.lib()
    xs, ys, adds = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    _lib.check(L.rk_lds_pack(C.byref(info), _lib.ptr(x), _lib.ptr(xs), 1, 0, _lib.stream_ptr()), "pack")
    _lib.check(L.rk_lds_pack(C.byref(info), _lib.ptr(add), _lib.ptr(adds), 1, 0, _lib.stream_ptr()), "pack")
    epi = _lib.LdsEpilogue(add=_lib.ptr(adds), y=_lib.ptr(ys), sum_scale=1.0)
    wave_desc, n_blocks = g.schedule(d)
    scratch = g.new_scratch(d)
    yy = torch.empty_like(x)

    def run_lds():
        _lib.check(L.rk_spmm_lds(C.byref(info), _lib.ptr(plan), _lib.ptr(xs), C.byref(epi), _lib.stream_ptr()), "rk_spmm_lds")

    def run_csr():
        _lib.check(L.rk_spmm_csr(N, _lib.ptr(g.rowptr), _lib.ptr(g.col), _lib.ptr(g.val), _lib.ptr(wave_desc), n_blocks,
                                 _lib.ptr(scratch), d, _lib.ptr(x), _lib.ptr(add), _lib.ptr(yy), _lib.stream_ptr()), "rk_spmm_csr")

    for name, fn in (("lds_us", run_lds),) + (() if a.lds_only else (("csr_us", run_csr),)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) * 1e3 / a.iters
    if a.no_stamps:
        print(json.dumps(out))
        return
    # in-kernel wall-clock stamps of one launch (100 MHz counter): start, staged, gathered, done per workgroup
    st = torch.zeros(info.n_wg * 4, device=dev, dtype=torch.int64)
    epi_s = _lib.LdsEpilogue(add=_lib.ptr(adds), y=_lib.ptr(ys), sum_scale=1.0, stamps=_lib.ptr(st))
    for _ in range(3):
        _lib.check(L.rk_spmm_lds(C.byref(info), _lib.ptr(plan), _lib.ptr(xs), C.byref(epi_s), _lib.stream_ptr()), "rk_spmm_lds")
    torch.cuda.synchronize()
    t = st.view(-1, 4).cpu().numpy().astype(np.int64)
    t0 = t[:, 0].min()
    ph = {"start_skew": t[:, 0] - t0, "stage": t[:, 1] - t[:, 0], "gather": t[:, 2] - t[:, 1], "rows": t[:, 3] - t[:, 2]}
    wgt = plan[int(plan[9].item()): int(plan[9].item()) + 4 * info.n_wg].view(-1, 4)[:, 0].cpu().numpy()
    out["stamps_us"] = {k: {"mean": float(v.mean()) / 100.0, "max": float(v.max()) / 100.0,
                            "half0_mean": float(v[wgt == 0].mean()) / 100.0, "half1_mean": float(v[wgt == 1].mean()) / 100.0}
                        for k, v in ph.items()}
    out["stamps_us"]["span"] = float(t[:, 3].max() - t0) / 100.0
    alg = 8 * g.nnz + 4 * (N + 1) + 2 * 4 * N * d
    out["alg_bytes"] = alg
    out["lds_frac_of_8TBs"] = alg / (out["lds_us"] * 1e-6) / 8e12
    if "csr_us" in out:
        out["csr_frac_of_8TBs"] = alg / (out["csr_us"] * 1e-6) / 8e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
