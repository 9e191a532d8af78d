"""Where a phase of the multi-phase propagation launch spends its time (tuning build: make -C recad_amd/csrc tuning, or the
-DRK_TUNING objects; run with RECAD_TUNING_LIB=.../librecad_hip_tuning.so RK_LDS_MSTAMPS=1): per (workgroup, item) wall-clock
stamps {ticket known, wait over, staged, gathered, rows done, stores drained} of ONE forward pass on the ml1m-shaped graph."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib, dataset, model, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    _lib.RK_LDS_SYNC_WORDS = 2560 + 2 * 8 * 4 * 256 + 64   # room for the stamps (uint64 x 8 per item, 4 items per workgroup)
    d = synth.make("ml1m")
    ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True,
                             device=dev, graph_source="train", seed=3)
    torch.manual_seed(1)
    m = model.from_config("victim", "lightgcn", latent_dim_rec=64, lightGCN_n_layers=3).I(dataset=ds).to(dev)
    m.use_lds, m.fuse_layers = True, True
    h = m._ensure_handle()
    for _ in range(20):
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
    torch.cuda.synchronize()
    raw = m._ws["lds_sync"][2560: 2560 + 2 * 8 * 4 * 256].cpu().numpy().view(np.uint64).reshape(256, 4, 8)
    st = raw[:, :3, :6].astype(np.float64) / 100.0          # us (100 MHz wall clock)
    t0 = st[:, 0, 0].min()
    st -= t0
    names = ["ticket", "wait over", "staged", "gathered", "rows done", "drained"]
    for p in range(3):
        seg = st[:, p, :]
        print(f"phase {p}: " + "  ".join(f"{n} {seg[:, k].mean():7.2f} (max {seg[:, k].max():7.2f})" for k, n in enumerate(names)))
        dur = np.diff(seg, axis=1)
        print("   durations: " + "  ".join(f"{a}->{b} {dur[:, k].mean():5.2f}" for k, (a, b) in enumerate(zip(names[:-1], names[1:]))))
    print("pass: first ticket -> last drained %.2f us; per phase %.2f us" % (st[:, 2, 5].max(), st[:, 2, 5].max() / 3))


if __name__ == "__main__":
    main()
