"""Soak: the same training run (ml1m-shaped, d=64, 3 layers, ordered scatter so that only the SpMM form differs) through the
LDS-resident sliced SpMM and through the row-gather SpMM: mean epoch losses, final tables, HR@K of one target after E epochs.
north_star's bar for the floating-point side is 1e-4 relative on BPR loss / HR@50.
    python3 scripts/lds_vs_csr_soak.py [epochs=3]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model, synth  # noqa: E402
from recad_amd.evaluate import eligible_users, full_catalog_topk, hit_counts  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
d = synth.make("ml1m")
out = {}
runs = {}
for form in ("lds", "row_gather"):
    ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True,
                             device=dev, graph_source="train", pairwise_batch_size=1024, seed=77)
    torch.manual_seed(2023)
    m = model.from_config("victim", "lightgcn", latent_dim_rec=64, lightGCN_n_layers=3, deterministic=True).I(dataset=ds).to(dev)
    m.use_lds = form == "lds"
    losses = [m.train_step(progress_bar=None)[0] for _ in range(epochs)]
    assert (m._ws.get("lds") is not None) == (form == "lds")
    ptr, idx = ds.train_csr_sorted()
    target = int(np.argsort(np.bincount(idx, minlength=ds.n_items))[ds.n_items // 2])   # a mid-popularity item
    users = eligible_users(ptr, idx, [target])
    res = full_catalog_topk(m, users, ptr, idx, [target], K=100)
    hits = [(res["target_rank"][:, 0] < k).mean() for k in (10, 20, 50, 100)]
    runs[form] = {"losses": losses, "user": m.embedding_user.weight.detach().cpu().numpy(), "item": m.embedding_item.weight.detach().cpu().numpy(),
                  "hr": hits, "top": res["top_ids"], "tscore": res["target_score"][:, 0]}
a, b = runs["lds"], runs["row_gather"]
rel = lambda x, y: float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).max() / max(np.abs(np.asarray(y, np.float64)).max(), 1e-30))
out = {"epochs": epochs, "steps": epochs * 459,
       "mean_epoch_loss": {"lds": a["losses"], "row_gather": b["losses"], "max_rel_diff": max(abs(x - y) / abs(y) for x, y in zip(a["losses"], b["losses"]))},
       "tables_max_rel_diff": {"user": rel(a["user"], b["user"]), "item": rel(a["item"], b["item"])},
       "hr@10,20,50,100": {"lds": [float(h) for h in a["hr"]], "row_gather": [float(h) for h in b["hr"]]},
       "target_score_max_rel_diff": rel(a["tscore"], b["tscore"]),
       "top100_lists_identical_users": float((a["top"] == b["top"]).all(axis=1).mean()),
       "top100_same_set_users": float(np.mean([set(x) == set(y) for x, y in zip(a["top"][::17], b["top"][::17])]))}
print(json.dumps(out))
