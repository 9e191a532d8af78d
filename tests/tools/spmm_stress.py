"""Randomised SpMM-vs-oracle stress over graph shapes, degree distributions, dims and schedule choices."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from recad_amd.graph import CsrGraph


def run(seed=0, n_cases=100):
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(n_cases):
        n = int(rng.integers(20, 6000))
        kind = rng.integers(0, 4)
        if kind == 0: deg = rng.poisson(rng.uniform(0.5, 30), n)
        elif kind == 1: deg = (rng.pareto(1.2, n) * rng.uniform(1, 40)).astype(np.int64)
        elif kind == 2: deg = rng.integers(60, 300, n)
        else: deg = np.where(rng.random(n) < 0.02, rng.integers(500, max(501, n)), rng.poisson(8, n))
        deg = np.minimum(deg, n).astype(np.int64)
        rowptr = np.zeros(n + 1, dtype=np.int32); rowptr[1:] = np.cumsum(deg)
        col = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in deg] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
        val = rng.random(len(col), dtype=np.float32)
        d = int(rng.choice([32, 64, 128, 256, 48, 8, 100]))
        split = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
        g = CsrGraph(n, torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev), class_split=split)
        x = rng.standard_normal((n, d), dtype=np.float32); add = rng.standard_normal((n, d), dtype=np.float32)
        y = g.spmm(torch.from_numpy(x).to(dev), torch.from_numpy(add).to(dev)).cpu().numpy()
        ref = orc.spmm(rowptr, col, val, x) + add
        err = float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))
        worst = max(worst, err)
        if err > 1e-6 + 2e-7 * float(np.sqrt(max(int(deg.max()), 1))):  # fp32 accumulation of maxdeg terms in a different order
            print("MISMATCH case", case, dict(n=n, kind=int(kind), d=d, split=split, nnz=len(col), maxdeg=int(deg.max())), err); raise AssertionError('mismatch')
    print(f"{n_cases} cases ok, worst relative error {worst:.2e}")


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100)
