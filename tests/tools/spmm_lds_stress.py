"""Randomised check of the LDS-resident sliced SpMM (rk_spmm_lds) against the oracle's CSR product:
    python tests/tools/spmm_lds_stress.py <seed> <n_cases>
Random bipartite graphs (tiny classes, empty rows, rows longer than every chunk cap, dense and sparse), every dim that is a
multiple of 4 up to 256, through CsrGraph.lds_plan -> pack -> rk_spmm_lds -> unpack.  Prints the worst relative error."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle as orc  # noqa: E402
from recad_amd.graph import CsrGraph  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    worst, done, skipped = 0.0, 0, 0
    for case in range(n_cases):
        U = int(rng.choice([1, 3, 17, 64, 300, 1500, 5000, 9000]))
        I = int(rng.choice([1, 2, 16, 33, 200, 1200, 3700, 9000]))
        d = int(rng.choice([4, 8, 12, 32, 48, 64, 100, 128, 256]))
        dens = float(rng.choice([0.002, 0.02, 0.2, 0.9]))
        deg = rng.binomial(I, dens, U)
        if rng.random() < 0.5:
            deg[rng.integers(0, U)] = I            # one full row
        if rng.random() < 0.5:
            deg[rng.integers(0, U, max(1, U // 10))] = 0   # empty rows
        if deg.sum() == 0:
            deg[0] = min(I, 1)
        ptr = np.zeros(U + 1, dtype=np.int64)
        ptr[1:] = np.cumsum(deg)
        idx = np.concatenate([np.sort(rng.choice(I, size=int(k), replace=False)) for k in deg]).astype(np.int32) if deg.sum() else np.zeros(0, np.int32)
        g = CsrGraph.from_user_item_csr(U, I, ptr, idx, dev)
        got = g.lds_plan(d)
        if got is None:
            skipped += 1
            continue
        N = U + I
        x = rng.standard_normal((N, d), dtype=np.float32)
        add = rng.standard_normal((N, d), dtype=np.float32)
        y = g.spmm_lds(torch.from_numpy(x).to(dev), torch.from_numpy(add).to(dev)).cpu().numpy()
        y2 = g.spmm_lds(torch.from_numpy(x).to(dev), torch.from_numpy(add).to(dev)).cpu().numpy()
        ref = orc.spmm(g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy(), x) + add
        err = float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))
        assert np.array_equal(y, y2), (case, U, I, d, "not reproducible")
        assert err < 3e-6, (case, U, I, d, dens, err)
        worst = max(worst, err)
        done += 1
    print(f"seed {seed}: {done} cases clean ({skipped} graphs did not qualify), worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
