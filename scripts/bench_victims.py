"""Secondary measurements (not the driver's bench line): MF and NCF victims.
  MF  : ml1m-shaped synthetic, embedding 64 (BASELINE.json config 1 shape, on the GPU)
  NCF : Amazon-game interactions (train edges of the golden fixture, 3179 x 5600, 34 439 edges),
        factor 32 / 5 layers (reference default), factor 256 / 3 layers and factor 256 / 5 layers (config 5's dim = 256 at the
        reference's default depth, default.py:123-125: tower 8192 -> 4096 -> ... -> 256, 89.4 MFLOP per scored pair)
Prints samples/s for a training epoch (device sampler included and excluded) and users/s for the
full-catalog evaluation."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, model, synth
from recad_amd.evaluate import eligible_users, full_catalog_topk
from tests import _golden as G

dev = torch.device('cuda:0')
out = {}

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

def run(name, ds, m, flops_per_sample=None, eval_flops_per_pair=None, eval_flops_per_user=0):
    m = m.to(dev)
    ep = ds.generate_epoch()
    users, items, labels = (ep[k] for k in ("users", "items", "labels"))
    n = users.numel()
    t_body = timed(lambda: m._run_epoch(users, items, labels, 1024))
    t_full = timed(lambda: m.train_step())
    ptr, idx = ds.train_csr_sorted()
    ev = eligible_users(ptr, idx, [0])
    t_eval = timed(lambda: full_catalog_topk(m, ev, ptr, idx, [0], K=100, chunk=512 if name.startswith("ncf") else 8192), n=1 if name.endswith("f256_l5") else 2)
    r = {"samples": n, "train_samples_per_s_body": n / t_body, "train_samples_per_s_with_sampler": n / t_full,
         "us_per_step": t_body / ((n + 1023) // 1024) * 1e6, "executed_gflop_per_step": (3 * flops_per_sample * 1024 / 1e9) if flops_per_sample else None, "eval_users_per_s": len(ev) / t_eval, "eval_users": int(len(ev)),
         "pair_scorings_per_s": float((ds.n_items - np.diff(ptr)[ev]).sum()) / t_eval}
    if flops_per_sample:
        r["train_tflops"] = 3 * flops_per_sample * n / t_body / 1e12
        # EXECUTED flops of the evaluation: rk_ncf_forward computes the user half of tower layer 0 once per user (a
        # [1, E] x [E, out0] product) and only the item half per (user, item) pair, so the nominal per-pair count
        # overstates the work done (it read 178.9 TF/s -- above the 157.3 fp32 MFMA peak -- at f = 256 in round 2)
        r["eval_tflops"] = ((eval_flops_per_pair or flops_per_sample) * len(ev) * ds.n_items + eval_flops_per_user * len(ev)) / t_eval / 1e12
        r["eval_tflops_nominal"] = flops_per_sample * len(ev) * ds.n_items / t_eval / 1e12
        r["eval_mfma_frac"] = r["eval_tflops"] / 157.3
    out[name] = r
    print(name, json.dumps(r))

d = synth.make("ml1m")
ds = dataset.from_config("implicit", "ml1m", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=False, device=dev, sample="pointwise", seed=1)
run("mf_ml1m_e64", ds, model.from_config("victim", "mf", embedding_size=64).I(dataset=ds))
g = G.load("lightgcn_game_d64")
ds = dataset.from_config("implicit", "game", train_csr=(g["train_ptr"].astype(np.int64), g["train_idx"]), test_csr=(g["test_ptr"].astype(np.int64), g["test_idx"]),
                         need_graph=False, device=dev, sample="pointwise", seed=1)
for f, L in ((32, 5), (256, 3), (256, 5)):
    fl = 2 * sum((f * 2 ** (L - l)) * (f * 2 ** (L - l)) // 2 for l in range(L)) + 3 * f
    in0 = f * 2 ** L                      # tower layer 0: Linear(in0 -> in0 / 2) over [user emb | item emb], each in0 / 2 wide
    half0 = 2 * (in0 // 2) * (in0 // 2)   # flops of ONE half of layer 0 for one row
    run(f"ncf_game_f{f}_l{L}", ds, model.from_config("victim", "ncf", factor_num=f, num_layers=L).I(dataset=ds), fl,
        eval_flops_per_pair=fl - half0, eval_flops_per_user=half0)
json.dump(out, open("gpurun_out/bench_victims.json", "w"), indent=1)
