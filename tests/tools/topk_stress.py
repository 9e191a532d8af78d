"""Randomised rk_topk_rows-vs-oracle stress: random catalog sizes (LDS and L2 forms), K, targets, score
distributions with and without heavy ties, seen lists from empty to almost everything."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from recad_amd import _lib


def run(seed=0, n_cases=100):
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(seed)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=dev).contiguous()
    for case in range(n_cases):
        I = int(rng.choice([rng.integers(1, 64), rng.integers(64, 4000), rng.integers(4000, 11000), rng.integers(11000, 40000)]))
        nb = int(rng.integers(1, 12))
        K = int(rng.choice([1, 5, 10, 50, 100, 200, 256]))
        kind = int(rng.integers(0, 6))
        if kind == 0: scores = rng.standard_normal((nb, I), dtype=np.float32)
        elif kind == 1: scores = (rng.integers(-5, 5, (nb, I)) / 4.0).astype(np.float32)
        elif kind == 2: scores = np.full((nb, I), 1.25, dtype=np.float32)
        elif kind == 3: scores = (np.float32(2.0) + np.spacing(np.float32(2.0)) * rng.integers(0, 4, (nb, I))).astype(np.float32)
        elif kind == 4: scores = (rng.standard_normal((nb, I)) * 1e-3).astype(np.float32); scores[:, ::7] = 0.0; scores[:, 3::11] = -0.0
        else: scores = np.exp(rng.standard_normal((nb, I)) * 3).astype(np.float32) * rng.choice([-1, 1], (nb, I)).astype(np.float32)
        seen = []
        for b in range(nb):
            m = int(rng.choice([0, rng.integers(0, min(I, 50) + 1), max(0, I - int(rng.integers(0, 20)))]))
            seen.append(np.sort(rng.choice(I, size=min(m, I), replace=False)).astype(np.int32))
        nt = int(rng.integers(0, 7))
        targets = rng.choice(I, size=min(nt, I), replace=False).astype(np.int32)
        nt = len(targets)
        sp = np.zeros(nb + 1, dtype=np.int32); sp[1:] = np.cumsum([len(x) for x in seen])
        si = np.concatenate(seen + [np.zeros(1, dtype=np.int32)]).astype(np.int32)
        sc = t(scores.copy(), torch.float32)
        top_ids = torch.empty(nb, K, dtype=torch.int32, device=dev); top_sc = torch.empty(nb, K, device=dev)
        ts_ = torch.empty(nb, max(nt, 1), device=dev); tr = torch.empty(nb, max(nt, 1), dtype=torch.int32, device=dev)
        tg = t(targets if nt else np.zeros(1, dtype=np.int32), torch.int32)
        uid, sp_d, si_d = torch.arange(nb, dtype=torch.int32, device=dev), t(sp, torch.int32), t(si, torch.int32)
        _lib.check(_lib.lib().rk_topk_rows(_lib.ptr(sc), nb, I, _lib.ptr(uid), _lib.ptr(sp_d), _lib.ptr(si_d), K, _lib.ptr(top_ids),
                                           _lib.ptr(top_sc), _lib.ptr(tg), nt, _lib.ptr(ts_), _lib.ptr(tr), _lib.stream_ptr()), "rk_topk_rows")
        torch.cuda.synchronize()
        ti, tsn, tsc, trn = top_ids.cpu().numpy(), top_sc.cpu().numpy(), ts_.cpu().numpy(), tr.cpu().numpy()
        for b in range(nb):
            rid, rsc, rts, rtr = orc.topk_row(scores[b], seen[b], K, targets)
            ok = np.array_equal(ti[b], rid) and np.array_equal(tsn[b], rsc)
            if nt: ok = ok and np.array_equal(tsc[b, :nt], rts) and np.array_equal(trn[b, :nt], rtr)
            if not ok:
                print("MISMATCH case", case, dict(I=I, nb=nb, K=K, kind=kind, row=b, n_seen=len(seen[b]), nt=nt)); raise AssertionError('mismatch')
    print(f"{n_cases} cases ok")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
