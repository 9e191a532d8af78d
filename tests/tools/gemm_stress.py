"""Randomised scoring-GEMM stress: rk_score_topk's score matrix must equal the oracle's k-ordered fmaf chain bit for bit
(random block sizes, catalog sizes, dims, gathered user ids, with and without biases)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from recad_amd import _lib
from recad_amd.evaluate import score_plan


def run(seed=0, n_cases=100):
    _run(seed, n_cases)   # (the score matrix is what this tool checks: every call asks for the GEMM + selection path)


def _run(seed, n_cases):
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(seed)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=dev).contiguous() if a is not None else None
    for case in range(n_cases):
        nu = int(rng.integers(1, 900)); nb = int(rng.integers(1, nu + 1)); I = int(rng.integers(1, 3000)); d = int(rng.choice([1, 3, 8, 31, 32, 33, 64, 100, 128, 200, 256]))
        bias = bool(rng.integers(0, 2))
        utab = rng.standard_normal((nu, d), dtype=np.float32); itab = rng.standard_normal((I, d), dtype=np.float32)
        ub = rng.standard_normal(nu, dtype=np.float32) if bias else None
        ib = rng.standard_normal(I, dtype=np.float32) if bias else None
        ids = rng.permutation(nu)[:nb].astype(np.int32)
        K = min(100, 256)
        sp = np.zeros(nu + 1, dtype=np.int32); si = np.zeros(1, dtype=np.int32)
        top_ids = torch.empty(nb, K, dtype=torch.int32, device=dev); top_sc = torch.empty(nb, K, device=dev)
        ts_ = torch.empty(nb, 1, device=dev); tr = torch.empty(nb, 1, dtype=torch.int32, device=dev)
        plan = score_plan(nb, I, d, K, 1, {"path": "gemm"})
        scratch = torch.empty(int(plan.scratch_floats), device=dev)   # [nb, plan.ld_scores]: rows padded to 128-byte lines
        tu, ti, tub, tib, tid, tsp, tsi, tg = t(utab, torch.float32), t(itab, torch.float32), t(ub, torch.float32), t(ib, torch.float32), t(ids, torch.int32), t(sp, torch.int32), t(si, torch.int32), t(np.zeros(1, dtype=np.int32), torch.int32)
        _lib.check(_lib.lib().rk_score_topk(d, _lib.ptr(tu), nb, _lib.ptr(tid), _lib.ptr(ti), I, _lib.ptr(tub), _lib.ptr(tib), 0.5 if bias else 0.0,
                                            _lib.ptr(tsp), _lib.ptr(tsi), K, _lib.ptr(top_ids), _lib.ptr(top_sc), _lib.ptr(tg), 1, _lib.ptr(ts_), _lib.ptr(tr),
                                            C.byref(plan), _lib.ptr(scratch), _lib.stream_ptr()), "rk_score_topk")
        got = scratch.view(nb, int(plan.ld_scores))[:, :I].cpu().numpy()
        ref = orc.score_rows(utab[ids], itab, ub[ids] if bias else None, ib, 0.5 if bias else 0.0)
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            print("MISMATCH case", case, dict(nu=nu, nb=nb, I=I, d=d, bias=bias), "first", bad[0], got[tuple(bad[0])], ref[tuple(bad[0])], "n_bad", len(bad)); raise AssertionError('mismatch')
    print(f"{n_cases} cases ok (bit-exact)")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
