#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: BPR train interactions/s (+ full-catalog top-K scorings/s) of the
LightGCN victim on synthetic interaction matrices, on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE.json config[1] -- ml1m-shaped (5950 x 3702, ~470K train edges, intended train
graph), dim=64, 3 layers, B=1024.  A "step" = one pass of the hot path over one minibatch: forward
propagation (L SpMM), gather + BPR softplus + L2 reg, backward (L SpMM), dense Adam; K steps are timed with
the pre-sampled triplets already resident in HBM, after W untimed steps on the SAME buffers and the SAME
captured hipGraph (victim.reserve(): nothing is allocated, captured or instantiated inside the timed
region).  The full-catalog scoring + top-100 + HR@K pass is timed after it ("topk").

N > 1: `python bench.py --gpus N` starts its own N worker processes (one per GPU, before anything touches a
GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is a worker itself.
The TOP-LEVEL line is the SAME quantity for every N: the metric's own workload (ml1m-shaped, the one `--gpus 1`
reports) as N independent victim replicas -- one retrain job per GPU, no data-path collective, "scaling":
"weak", value = N x steps x B / max-over-ranks time -- so that the 1 -> 8 series is one curve and N = 1 equals
BENCH.  The strong-scaling measurements of ONE training job sharded over the N GPUs live in `also`, each with
its own single-GPU denominator and a `ranks_seen` record: `also.config4_rows2d` (1M x 500K x 100M edges, the
1 x N column-slab form of recad_amd/sharded2d.py: per propagation layer one tile SpMM on the GPU's own block
and one chunk-overlapped RCCL reduce-scatter) and `also.config3_yelp_rows2d`.  `--parallel rows2d | rows`
(with `--workload`) makes such a sharded job the top-level line instead -- a labelled extra mode whose
`metric` names its workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # same guide: v_mfma_f32_32x32x2_f32 dense peak
LDS_PEAK_GBS = 256 * 256 * 2.4   # same guide, "LDS": ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 TB/s
METRIC = "BPR train interactions/sec + full-catalog top-K scorings/sec, LightGCN ml1m dim=64"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: one ml1m epoch = 458; 20 for the larger shapes / sharded modes)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=["ml1m", "yelp", "tiny", "c4s", "config4"],
                    help="default: ml1m (BASELINE.json config[1]) for every N")
    ap.add_argument("--graph", default="train", choices=["train", "reference"],
                    help="adjacency from the train edges (BASELINE '~470K edges') or the reference's as-is test-edge graph")
    ap.add_argument("--dim", type=int, default=None, help="default 64 (128 for --workload yelp, BASELINE config 3)")
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--graph-steps", type=int, default=32, help="train steps per hipGraph replay (0 = plain launches)")
    ap.add_argument("--spmm", default="auto", choices=["auto", "lds", "csr"],
                    help="LightGCN propagation kernel: auto (the library's choice), lds (LDS-resident sliced SpMM where the graph qualifies), "
                         "csr (row gather) -- A/B of the two forms on one graph")
    ap.add_argument("--no-block-list", action="store_true", help="row-gather path: start every workgroup in the row-filtered last forward layer (A/B of the marked-block list)")
    ap.add_argument("--fuse-layers", action="store_true", help="LDS path: the L layers of a pass as one multi-phase launch (opt-in form, measured slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle replay of the timed path (parity object, cpu_baseline_port)")
    ap.add_argument("--parity-steps", type=int, default=0, help="oracle steps laid next to the timed run (0 = 25 at ml1m size, 3 at yelp size)")
    ap.add_argument("--no-topk", action="store_true", help="skip the evaluation leg")
    ap.add_argument("--deterministic", action="store_true",
                    help="ordered (bit-reproducible) gradient scatter instead of float atomics (fused and row-sharded paths; labelled in config)")
    ap.add_argument("--eval-users", type=int, default=0, help="evaluate only the first n eligible users (0 = all)")
    ap.add_argument("--parallel", default=None, choices=["rows", "rows2d", "replicas"],
                    help="N>1 default: replicas = one independent victim replica per GPU on the metric's workload (weak scaling, no data-path "
                         "collective; the sharded legs go to `also`).  As the TOP-LEVEL line instead (labelled extra mode): rows2d = the Pr x Pc "
                         "tiling (recad_amd/sharded2d.py; default grid 1 x N: column slabs, no all-gather, one chunk-overlapped reduce-scatter "
                         "per layer -- strong scaling); rows: the 1-D row partition with RCCL all-gathers")
    ap.add_argument("--grid-rows", type=int, default=0, help="rows2d: Pr of the Pr x Pc grid (0 = 1: column slabs)")
    ap.add_argument("--reduce", default="collective", choices=["collective", "ordered"],
                    help="rows2d: reduce_scatter_tensor (RCCL picks the algorithm) or all-to-all + sum in group-rank order (fixed order)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the workers")
    ap.add_argument("--gather", default="collective", choices=["collective", "direct"],
                    help="rows mode: all_gather_into_tensor (RCCL picks the algorithm) or one batched group of W-1 sends / receives "
                         "per rank (one-shot all-gather over the xGMI mesh)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N=1 with --parallel rows: keep every collective of the sharded trainer (one-rank RCCL group): "
                         "exercises the real all-gather / all-reduce calls, async handles and stream ordering on a 1-GPU box")
    ap.add_argument("--workflow", action="store_true",
                    help="labelled extra mode (SURVEY 8d config 3): time train(clean) + inject 50 fake users + train(poisoned) + 2 x "
                         "evaluation through workflow.execute(), with the graph / schedule rebuild of the retrain loop itemised")
    ap.add_argument("--rec-epoch", type=int, default=2, help="--workflow: training epochs per (re)train")
    ap.add_argument("--no-also", action="store_true", help="N = 1 default run: skip the short yelp-shaped measurement (`also`)")
    ap.add_argument("--no-also-config4", action="store_true",
                    help="N = 1 default run: skip the config-4-shaped measurement (1M x 500K x 100M edges, ~6 s) inside `also`")
    ap.add_argument("--no-also-sharded", action="store_true",
                    help="N > 1 default run: skip the strong-scaling legs (ONE job sharded over the N GPUs: config-4 and yelp shapes, rows2d) in `also`")
    ap.add_argument("--also-timeout", type=float, default=480.0,
                    help="N > 1: seconds the sharded `also` legs may take before the (already measured) top-level line is printed without them")
    ap.add_argument("--share-gpu", action="store_true", help="box check: several nccl / gloo ranks on ONE GPU (rank -> device rank %% n_devices)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not run the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this run")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: workers rendezvous over gloo, all-reduce their ranks and exit")
    a = ap.parse_args(argv)
    if a.workload is None:
        a.workload = "ml1m"          # the metric's own workload for EVERY N: the 1 -> 8 series is one quantity
    if a.parallel is None:
        a.parallel = "rows" if (a.force_collectives and a.gpus == 1) else "replicas"
    if a.dim is None:
        a.dim = 128 if a.workload == "yelp" else 64
    small = a.workload in ("ml1m", "tiny") and a.parallel == "replicas"
    if a.steps is None:
        a.steps = 458 if small else 20
    if a.warmup is None:
        a.warmup = 32 if small else 5
    return a


# ------------------------------------------------------------------------------------------------ launcher
def launch_workers(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N workers (fresh interpreters, so no
    process that has touched a GPU is ever re-executed), one per GPU, rendezvous on 127.0.0.1.  Rank 0's
    stdout (the JSON line) is passed through; the exit code is the first non-zero worker code."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "LOCAL_WORLD_SIZE": str(args.gpus)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(args.gpus))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:      # one worker failed: the others would wait in a collective forever
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run_worker(args, rank, world):
    """Launcher check without a GPU: the workers rendezvous over gloo; rank 0 prints what the real run's line would carry as
    `metric` / `config.workload` / mode for this N (built from the synthetic shape on the CPU)."""
    import torch
    import torch.distributed as dist

    if "MASTER_ADDR" not in os.environ:   # (N = 1 without a launcher)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(sk.getsockname()[1])})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "rank_sum": float(t.item()), "parallel": args.parallel, "workload": args.workload,
               "metric": metric_for(args), "steps": args.steps, "warmup": args.warmup,
               "scaling": "strong" if (args.parallel in ("rows", "rows2d") and (world > 1 or args.force_collectives)) else "weak",
               "also_sharded_legs": (also_sharded_leg_names(args.no_also_config4) if (world > 1 and args.parallel == "replicas" and not args.no_also_sharded) else [])}
        if args.workload not in ("c4s", "config4"):
            from recad_amd import synth
            d = synth.make(args.workload)
            ptr, idx = d["train"] if args.graph == "train" else d["test"]
            out["config"] = {"workload": workload_string(args.workload, d["n_users"], d["n_items"], int(len(d["train"][1])), args.graph, 2 * int(len(idx)),
                                                         args.dim, args.layers, args.batch),
                             "mode": args.parallel if world > 1 else "single"}
        print(json.dumps(out))
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ CPU baselines
def oracle_replay(d, graph, layers, batch, triplets, user0, item0, n_steps):
    """n_steps train steps of the CPU oracle (1-thread C restatement of lightgcn.py:137-169) on the SAME triplets, from the
    SAME initial tables as the GPU run.  Checker code: -> per-step losses, final tables, seconds (graph build excluded)."""
    from oracle import oracle as orc

    U, I = d["n_users"], d["n_items"]
    ptr, idx = d["train"] if graph == "train" else d["test"]
    csr = orc.build_norm_adj(U, I, np.asarray(ptr).astype(np.int32), np.asarray(idx).astype(np.int32))
    user, item = np.array(user0, dtype=np.float32, copy=True), np.array(item0, dtype=np.float32, copy=True)
    st = orc.AdamState(user.shape, item.shape)
    users, pos, neg = triplets
    losses = []
    t0 = time.perf_counter()
    for s in range(n_steps):
        sl = slice(s * batch, (s + 1) * batch)
        losses.append(orc.lightgcn_step(csr, user, item, st, users[sl], pos[sl], neg[sl], layers))
    return {"losses": np.asarray(losses, dtype=np.float64), "user": user, "item": item, "seconds": time.perf_counter() - t0,
            "steps": n_steps}


def relerr_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run_steps(victim, triplets, B, lo, n_steps):
    """Steps [lo, lo + n_steps) of the resident triplets through the victim's epoch call -- the timed call of this bench
    (after victim.reserve(): one whole-call hipGraph replay for <= 64 steps, chunk graphs beyond)."""
    users, pos, neg = triplets
    sl = slice(lo * B, (lo + n_steps) * B)
    return victim._run_epoch(users[sl], pos[sl], neg[sl], B)


def parity_object(rep, gpu_losses, gpu_tables, tables_from):
    """The self-validation every N = 1 run carries: the timed path next to the oracle on its own inputs."""
    n = min(len(gpu_losses), rep["steps"])
    lo = np.abs(np.asarray(gpu_losses[:n], dtype=np.float64) - rep["losses"][:n]) / np.abs(rep["losses"][:n])
    out = {"steps": int(n), "max_rel_loss_err": float(lo.max()), "loss_tol": 1e-5, "tables_relerr": None, "tables_tol": 1e-4,
           "tables_from": tables_from, "oracle": "oracle/recad_oracle.c (orc_lightgcn_step_general), same triplets, same initial tables"}
    if gpu_tables is not None:
        out["tables_relerr"] = max(relerr_max(gpu_tables[0], rep["user"]), relerr_max(gpu_tables[1], rep["item"]))
    out["ok"] = bool(out["max_rel_loss_err"] <= out["loss_tol"] and (out["tables_relerr"] is None or out["tables_relerr"] <= out["tables_tol"]))
    return out


def mfma_gemm_probe(dev, nb=8192, n_items=34474, dim=256, reps=40):
    """The scoring GEMM at north_star's MFMA shape (a block of 8192 users x the yelp-sized catalogue at d = 256), timed live
    with events on the launch stream: rk_score_matrix without biases is exactly the LightGCN scoring GEMM (gathered user
    rows, fp32 in / fp32 accumulate).  Reported next to the SpMM roofline; profiles/r02_gemm_d256_wide_pmc.json has the
    rocprofv3 / PMC view of the same launch."""
    import torch
    from recad_amd import _lib
    g = torch.Generator(device=dev).manual_seed(7)
    utab = torch.randn(nb, dim, device=dev, generator=g) * 0.1
    itab = torch.randn(n_items, dim, device=dev, generator=g) * 0.1
    ids = torch.randperm(nb, device=dev, generator=g).to(torch.int32)
    out = torch.empty(nb, n_items, device=dev)

    def once():
        _lib.check(_lib.lib().rk_score_matrix(dim, _lib.ptr(utab), nb, _lib.ptr(ids), _lib.ptr(itab), n_items, None, None, 0.0, 0.0, 0,
                                              _lib.ptr(out), _lib.stream_ptr()), "rk_score_matrix")
    for _ in range(8):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * nb * n_items * dim / us / 1e6
    del out
    return {"bound": "mfma", "kernel": "gemm_f32_wide_kernel<1,1>", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
            "shape": f"C[{nb},{n_items}] = U[ids][{nb},{dim}] . I[{n_items},{dim}]^T, fp32 in / fp32 accumulate", "avg_launch_us": us,
            "flops_per_launch": 2 * nb * n_items * dim,
            "note": "secondary roofline: the scoring GEMM at d = 256 (dense fp32 MFMA peak 157.3 TFLOP/s at 2.4 GHz)"}


def cpu_baseline_aten(d, graph, dim, layers, batch, triplets, budget_s=12.0):
    """SURVEY.md 8d's CPU baseline: the reference's own op sequence (recad/model/victim/lightgcn.py:82-113,137-169 and
    the per-user loop of recad/workflow/normal.py:57-93) on ATen with k host threads -- tests/tools/aten_ref.py, pinned
    against the reference's goldens by tests/test_aten_ref.py.  Baseline only; never on the product path."""
    from tests.tools import aten_ref

    ptr, idx = d["train"] if graph == "train" else d["test"]
    r = aten_ref.time_baseline(d["n_users"], d["n_items"], ptr, idx, d["train"][0], d["train"][1], dim, layers, batch, triplets,
                               budget_s=budget_s)
    return {"value": r["interactions_per_s"], "unit": "interactions/s", "cores": r["threads"], "kind": "port", "impl": "aten",
            "sample": f"{r['steps']} train steps of {batch} triplets through the reference's ATen op sequence (torch.sparse.mm on the "
                      f"coalesced COO graph, autograd, torch.optim.Adam; tests/tools/aten_ref.py, validated against the reference's "
                      f"goldens) on {r['threads']} of {r['host_threads']} host threads (fastest of 8/16/32/64), {r['seconds']:.1f} s; "
                      f"last loss {r['last_loss']:.5f}",
            "eval_users_per_s": r["eval_users_per_s"],
            "eval_sample": f"{r['eval_users']} users through the reference's per-user loop (computer() = {layers} sparse mm per user, "
                           f"pair scores of the unseen items, sort, top-100), same {r['threads']} threads, {r['eval_seconds']:.1f} s"}


def also_measure(dev, workload, dim, layers, B, steps=20, warmup=5, eval_users=0, parity_steps=0):
    """A short measurement of ANOTHER BASELINE.json config next to the headline one, so that the driver's record carries it:
    ms/step through the same reserve() -> epoch-call path, the SpMM's per-launch time and roofline (with the no-reuse gather
    figure as `effective`), and one full evaluation.  Config 3 = yelp-shaped d = 128; config 4 = 1M x 500K x 100M edges d = 64."""
    import torch
    from recad_amd import _lib, dataset, model, synth
    from recad_amd.evaluate import eligible_users, full_catalog_topk, hit_counts

    t_all = time.perf_counter()
    big = workload in ("c4s", "config4")
    if big:
        dd = synth.make_device(workload, dev)
        d = {k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
        del dd
    else:
        d = synth.make(workload)
    ds = dataset.from_config("implicit", workload, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True,
                             device=dev, graph_source="train", pairwise_batch_size=B, seed=1234)
    torch.manual_seed(2023)
    v = model.from_config("victim", "lightgcn", latent_dim_rec=dim, lightGCN_n_layers=layers).I(dataset=ds).to(dev)
    g = ds.graph_csr()
    N, nnz = g.n_rows, g.nnz
    ep = ds.generate_epoch()
    trip = tuple(ep[k][: (steps + warmup) * B].contiguous() for k in ("users", "positive_items", "negative_items"))
    init_tables = tuple(p_.detach().cpu().numpy().copy() for p_ in (v.embedding_user.weight, v.embedding_item.weight)) if parity_steps else None
    v.reserve(max(steps, warmup) * B, B)
    warm_losses = run_steps(v, trip, B, 0, warmup).sum(dim=1).double().cpu().numpy()   # (read now: the call returns a view of the handle's loss buffer, which the next call overwrites)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    part = run_steps(v, trip, B, warmup, steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert np.isfinite(float(part[-1].sum().item()))
    h = v._ensure_handle()
    reps = 30 if nnz < 20_000_000 else 4
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
    e0.record()
    for _ in range(reps):
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
    e1.record()
    torch.cuda.synchronize()
    lds = v._ws.get("lds") is not None
    spmm_ms = e0.elapsed_time(e1) / (reps * layers)
    alg = 8 * nnz + 4 * (N + 1) + 8 * N * dim
    gather = 8 * nnz + 4 * nnz * dim + 4 * N * dim
    ptr, idx = ds.train_csr_sorted()
    users = eligible_users(ptr, idx, np.array([0], dtype=np.int32))
    if eval_users:
        users = users[:eval_users]
    ud = torch.as_tensor(users, dtype=torch.int32, device=dev)
    pd_, id_ = torch.as_tensor(ptr, dtype=torch.int32, device=dev), torch.as_tensor(idx, dtype=torch.int32, device=dev)
    tg = torch.as_tensor(np.array([0], dtype=np.int32), device=dev)
    chunk = max(256, min(8192, (1 << 31) // max(ds.n_items, 1)))
    full_catalog_topk(v, ud, pd_, id_, tg, K=100, chunk=chunk, to_host=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = full_catalog_topk(v, ud, pd_, id_, tg, K=100, chunk=chunk, to_host=False)
    hits = hit_counts(res["target_rank"], (10, 20, 50, 100))
    torch.cuda.synchronize()
    ev = time.perf_counter() - t1
    parity = None
    if parity_steps:
        # the large-shape train step next to the oracle in the driver's own record (round-5 review): the first `parity_steps` steps
        # of this leg's triplets from this leg's initial tables -- losses from the run above (its warm-up call), tables through a second
        # victim run through the same reserve() -> whole-call hipGraph sequence for exactly those steps (the run's own tables are
        # `steps + warmup` steps old; the oracle needs ~2 s per step at this shape)
        n_par = min(int(parity_steps), warmup)
        torch.manual_seed(2023)
        v2 = model.from_config("victim", "lightgcn", latent_dim_rec=dim, lightGCN_n_layers=layers).I(dataset=ds).to(dev)
        v2.embedding_user.weight.data.copy_(torch.from_numpy(init_tables[0]))
        v2.embedding_item.weight.data.copy_(torch.from_numpy(init_tables[1]))
        v2.reserve(n_par * B, B)
        l2 = run_steps(v2, trip, B, 0, n_par).sum(dim=1).double().cpu().numpy()
        run_losses = warm_losses[:n_par]
        tabs2 = tuple(p_.detach().cpu().numpy().copy() for p_ in (v2.embedding_user.weight, v2.embedding_item.weight))
        rb = v2._ws.get("row_blocks")
        used_list = int(rb[0].item()) if rb is not None else None
        del v2
        rep = oracle_replay(d, "train", layers, B, tuple(t[: n_par * B].cpu().numpy() for t in trip), init_tables[0], init_tables[1], n_par)
        parity = parity_object(rep, run_losses, tabs2, f"a second victim from the same initial tables through the same call sequence ({n_par} steps)")
        parity["second_victim_max_rel_loss_diff"] = float(np.abs(l2 - run_losses).max() / np.abs(run_losses).max())
        parity["ok"] = bool(parity["ok"] and parity["second_victim_max_rel_loss_diff"] <= 1e-5)
        parity["spmm"] = "lds" if lds else "csr"
        parity["marked_block_list_count"] = used_list      # > 0: the row-filtered last forward layer ran from the list
        parity["frontier_filter"] = bool(v._ws.get("row_bits") is not None)
        parity["oracle_seconds"] = rep["seconds"]
    return {"workload": f"LightGCN victim, {workload}-shaped synthetic {ds.n_users}x{ds.n_items}, {ds.traindataSize} train edges (nnz {nnz}), "
                        f"dim={dim}, layers={layers}, batch={B}",
            "value": steps * B / el, "unit": "interactions/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup,
            "roofline": {"bound": "hbm", "kernel": "spmm_lds_kernel" if lds else f"spmm_csr_kernel<{dim}>", "avg_launch_us": spmm_ms * 1e3,
                         "bytes_per_launch": alg, "achieved": alg / spmm_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / spmm_ms / 1e6 / HBM_PEAK_GBS,
                         "gather_bytes_per_launch": gather, "effective_gbs": gather / spmm_ms / 1e6,
                         "effective_frac": gather / spmm_ms / 1e6 / HBM_PEAK_GBS,
                         "note": "per-launch time = propagate (L launches, HIP events) / L.  `frac` prices SURVEY 8d's COMPULSORY bytes "
                                 "(A, X, Y once); `effective_frac` the no-reuse gather figure (one X row per nonzero): on i.i.d. synthetic "
                                 "graphs there is no row reuse to harvest beyond the caches, so the kernel moves the gather bytes at about "
                                 "the fabric rate while `frac` stays low (DESIGN 4.1)"},
            "topk": {"eligible_users": int(len(users)), "seconds": ev, "value": len(users) / ev, "unit": "users/s",
                     "hr@50": float(hits[0, 2].item()) / max(len(users), 1),
                     "gemm_tflops_e2e": 2.0 * len(users) * ds.n_items * dim / ev / 1e12},
            "parity": parity,
            "seconds_total": time.perf_counter() - t_all}


# ------------------------------------------------------------------------------------------------ worker pieces
def workload_string(name, n_users, n_items, train_edges, graph, nnz, dim, layers, B):
    return (f"LightGCN victim, {name}-shaped synthetic {n_users}x{n_items}, {train_edges} train edges, graph={graph} (nnz {nnz}), "
            f"dim={dim}, layers={layers}, batch={B}, Adam lr 1e-3, lambda 1e-4")


def metric_for(args):
    """BASELINE.json's metric string for its own workload; any other workload / a sharded job as the top-level line is a labelled
    extra mode and says so in `metric`."""
    sharded_top = args.parallel in ("rows", "rows2d") and (args.gpus > 1 or args.force_collectives)
    if args.workload == "ml1m" and args.dim == 64 and not sharded_top:
        return METRIC
    return (f"BPR train interactions/sec + full-catalog top-K scorings/sec, LightGCN {args.workload}-shaped dim={args.dim}"
            + (f", ONE job sharded over the GPUs ({args.parallel})" if sharded_top else "") + " [extra mode: not BASELINE.json's headline workload]")


def load_workload(workload, dev, graph, B, seed):
    """Synthetic interactions of the named shape + the dataset object on `dev` (same seed => identical on every rank)."""
    from recad_amd import dataset, synth
    if workload in ("c4s", "config4"):
        dd = synth.make_device(workload, dev)
        d = {k: (tuple(t.cpu().numpy() for t in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
        del dd
    else:
        d = synth.make(workload)
    ds = dataset.from_config("implicit", workload, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], need_graph=True,
                             device=dev, graph_source=graph, pairwise_batch_size=B, seed=seed)
    return d, ds


def resident_triplets(ds, need):
    import torch
    cols = [[], [], []]
    have = 0
    while have < need:
        ep = ds.generate_epoch()
        for c, k in zip(cols, ("users", "positive_items", "negative_items")):
            c.append(ep[k])
        have += len(ep["users"])
    return tuple(torch.cat(c)[:need].contiguous() for c in cols)


def ranks_seen(dev, world, collectives):
    """What actually took part: the process group's world size and every rank's device, all-gathered (N > 1 or a one-rank
    group), so a line cannot claim more GPUs than ran."""
    import torch
    import torch.distributed as dist
    props = torch.cuda.get_device_properties(dev)
    mine = {"device_index": int(dev.index), "name": props.name, "pci_bus_id": int(getattr(props, "pci_bus_id", -1))}
    if not collectives:
        return {"world_size": 1, "devices": [mine]}
    t = torch.tensor([int(os.environ.get("RANK", 0)), int(dev.index), mine["pci_bus_id"]], device=dev, dtype=torch.int64)
    if dist.get_backend() == "gloo":
        t = t.cpu()
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return {"world_size": int(dist.get_world_size()), "backend": dist.get_backend(),
            "devices": [{"rank": int(o[0]), "device_index": int(o[1]), "pci_bus_id": int(o[2])} for o in out], "name": props.name}


def wait_done(stream):
    """Spin on a completion event before the contract's torch.cuda.synchronize(): the blocking wait's wake-up latency (tens of
    microseconds on this stack) would otherwise sit inside a 20-step timed region of ~1.4 ms."""
    import torch
    ev = torch.cuda.Event()
    ev.record(stream)
    while not ev.query():
        pass
    torch.cuda.synchronize()


def spmm_roofline(args, victim, N, nnz, traffic_live=None):
    """The dominant kernel: per-launch time by HIP events on ITS stream (captured + replayed for the ~9 us LDS kernel), SURVEY 8d's
    algorithmic bytes, and -- for the LDS-resident kernel -- the bound it actually runs against (`lds_frac`)."""
    import ctypes as C
    import torch
    from recad_amd import _lib
    reps = 200 if nnz < 20_000_000 else 10
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h = victim._ensure_handle()
    lds = victim._ws.get("lds")
    if lds is not None:
        # the LDS-resident sliced kernel (csrc/spmm_lds.h), launched exactly as the first forward layer of a train step
        ws = victim._ws
        plan, info = lds
        # the operand of the timed launches is the packed E0 of the real tables, not the zero-initialised workspace: zeros in the
        # gather table let the chip hold a ~4 % higher clock (8.85 vs 9.2 us per launch, profiles/r05f: same box, both orders)
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
        epi = _lib.LdsEpilogue(y=_lib.ptr(ws["buf_a"]), sum_in=_lib.ptr(ws["e0s"]), sum_out=_lib.ptr(ws["lsum"]), sum_scale=1.0)

        def spmm_once():
            _lib.check(_lib.lib().rk_spmm_lds(C.byref(info), _lib.ptr(plan), _lib.ptr(ws["e0s"]), C.byref(epi), _lib.stream_ptr()), "rk_spmm_lds")
        per_call, kname = 1, f"spmm_lds_kernel<{info.lpa}, {info.lpb}>"
    else:
        def spmm_once():
            _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
        per_call, kname = args.layers, f"spmm_csr_kernel<{args.dim}>"
    for _ in range(3):
        spmm_once()
    # The LDS kernel runs ~9 us: 200 separate ctypes calls can be HOST-bound (10-14 us each on a busy host: the same tree gave
    # 9.2 and 14.1 us on two boxes), which would time Python, not the kernel.  So the launches are captured once and replayed
    # (same kernel, same arguments, back to back on the stream the events are recorded on); a failed capture falls back to
    # the plain loop and says so.
    timed_as = "direct launches"
    replay = None
    if lds is not None:
        try:
            torch.cuda.synchronize()
            gcap = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gcap):
                for _ in range(reps):
                    spmm_once()
            gcap.replay()
            torch.cuda.synchronize()
            replay, timed_as = gcap, f"{reps} launches captured in one hipGraph, replayed"
        except Exception as e:   # noqa: BLE001
            replay, timed_as = None, f"direct launches (capture failed: {type(e).__name__})"
            torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    rounds = 7 if nnz < 20_000_000 else 1    # median of 7 windows of `reps` launches (a 2 ms window alone moved 9.7 - 10.7 us run to run)
    windows = []
    for _ in range(rounds):
        ev0.record(stream)
        if replay is not None:
            replay.replay()
        else:
            for _ in range(reps):
                spmm_once()
        ev1.record(stream)
        wait_done(stream)   # (spinning, not the blocking wait: this probe ends at the timed region's opening barrier, and a host core that slept
        #                      through seven 2 ms waits runs the timed call's Python ~4x slower -- 25 us per stage instead of 6)
        windows.append(ev0.elapsed_time(ev1) / (reps * per_call))
    spmm_ms = sorted(windows)[len(windows) // 2]
    spmm_bytes = 8 * nnz + 4 * (N + 1) + 2 * 4 * N * args.dim  # A once, X once, Y once
    achieved = spmm_bytes / (spmm_ms * 1e-3) / 1e9
    traffic = traffic_src = None
    tkey = f"{args.workload}_{args.graph}_d{args.dim}" + ("_lds" if lds is not None else "")
    if traffic_live and traffic_live.get("bytes") and traffic_live.get("key") == tkey:
        traffic, traffic_src = traffic_live["bytes"], traffic_live["source"]
    else:
        for tname in ("r05_spmm_traffic.json", "r04_spmm_traffic.json", "r03_spmm_traffic.json", "r02_spmm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(tkey)
                    traffic_src = f"profiles/{tname} (rocprofv3 PMC passes on a builder-run box, not measured in this run)" if traffic is not None else None
                except Exception:
                    traffic = None
        if traffic_live and traffic_live.get("error"):
            traffic_src = (traffic_src or "none") + f"; live PMC passes unavailable in this run: {traffic_live['error']}"
    roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": spmm_bytes, "avg_launch_us": spmm_ms * 1e3,
                "gather_bytes_per_launch": 8 * nnz + 4 * nnz * args.dim + 4 * N * args.dim,
                "timed_as": timed_as + (f"; median of {rounds} windows ({min(windows) * 1e3:.2f} - {max(windows) * 1e3:.2f} us)" if rounds > 1 else ""),
                "note": "per-launch time from HIP events on the launch stream around back-to-back launches (includes the "
                        "inter-kernel boundary); algorithmic bytes = SURVEY 8d's 8 nnz + 4 (N+1) + 8 N d"}
    if traffic:
        # the bytes the launch actually MOVED against the same peak: a row gather on a graph without reuse (every nonzero fetches its
        # d-float row: i.i.d. synthetic graphs) sits at the fabric's rate here while `frac`, priced on each operand being read once, is small
        roofline["traffic_frac"] = traffic / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        roofline["traffic_frac_note"] = "counter traffic / launch time / peak (fabric side, Infinity-Cache hits included): how busy the memory system is, not a quality claim"
    if traffic_live and traffic_live.get("detail"):
        roofline["traffic_detail"] = traffic_live["detail"]
    if lds is not None:
        on_chip = 4 * nnz * args.dim   # one 16-byte ds_read_b128 per nonzero and 4-float slice: what the LDS pipes deliver instead of the L2 -> L1 gather
        roofline["on_chip_bytes_per_launch"] = on_chip
        roofline["lds_peak_GBs"] = LDS_PEAK_GBS
        roofline["lds_frac"] = on_chip / (spmm_ms * 1e-3) / 1e9 / LDS_PEAK_GBS
        roofline["lds_note"] = ("the bound this kernel runs against: the graph's tables are LDS-resident, every nonzero costs one ds_read_b128; "
                                "lds_frac = on-chip gather bytes / WHOLE launch time / (256 CUs x 256 B/clk x 2.4 GHz, MI355X_MICROARCH.md 'LDS'); "
                                "the gather phase alone is ~45 % of the launch (stage, row reduction and the launch boundary are the rest: DESIGN 4.1b)")
    return roofline


def live_traffic_probe(args, timeout_s=150):
    """HBM-side traffic of the dominant kernel FROM THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one counter per
    pass (MI355X_MICROARCH.md 'HBM'; on gfx950 FETCH_SIZE counts a 128-byte request as 64 B => x2; WRITE_SIZE as is), around
    scripts/spmm_lds_probe.py (20 launches of the same kernel on the same synthetic graph), as CHILD processes started before
    this process touches the GPU (the program goes directly after `--`).  -> {"key", "bytes", "source", "detail"} or {"error"}."""
    import csv
    import glob
    import shutil
    import statistics
    import tempfile
    if args.workload not in ("ml1m", "tiny") or args.graph != "train" or args.spmm == "csr":
        return {"error": "live PMC passes are wired for the headline (LDS-resident) kernel only"}
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    kern = "spmm_lds_kernel"
    vals = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix="recad_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--", sys.executable,
               os.path.join(ROOT, "scripts", "spmm_lds_probe.py"), "--shape", args.workload, "--dim", str(args.dim), "--iters", "20", "--no-stamps", "--lds-only"]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)      # exactly the process group this call started
                p.wait()
                shutil.rmtree(out_dir, ignore_errors=True)
                return {"error": f"rocprofv3 --pmc {counter} did not finish within {timeout_s} s"}
            files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {counter}: rc {p.returncode}, {len(files)} counter file(s)"}
            got = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not got:
                return {"error": f"no {kern} dispatch in the {counter} pass"}
            vals[counter] = (statistics.mean(got), len(got))
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    fetch_kib, write_kib = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
    nbytes = int((2.0 * fetch_kib + write_kib) * 1024.0)
    return {"key": f"{args.workload}_{args.graph}_d{args.dim}_lds", "bytes": nbytes,
            "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, run by this bench.py as child processes before its own GPU "
                      "work (scripts/spmm_lds_probe.py: the same kernel on the same synthetic graph); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB "
                      "(gfx950: FETCH_SIZE tallies 128-byte requests at 64 B); fabric-side traffic, Infinity-Cache hits included",
            "detail": {"FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "dispatches": vals["FETCH_SIZE"][1], "seconds": time.perf_counter() - t0}}


# The A/B the first hardware SCALE run has to decide by itself (round-5 review, next #5): config 4 as ONE job over the N GPUs in the
# three exchange forms this repository holds, each with 1 and 4 row chunks (chunk c's collective runs under chunk c + 1's SpMM).
SHARDED_VARIANTS = (
    # name suffix,            parallel, reduce,       gather,        chunks
    ("collective_c1", "rows2d", "collective", "collective", 1),    # column slabs; reduce_scatter_tensor (RCCL picks the algorithm)
    ("collective_c4", "rows2d", "collective", "collective", 4),
    ("ordered_c1", "rows2d", "ordered", "collective", 1),          # column slabs; direct all-to-all + sum in group-rank order (one shot on the mesh)
    ("ordered_c4", "rows2d", "ordered", "collective", 4),
    ("rows_direct_c1", "rows", "collective", "direct", 1),         # 1-D rows; one batched group of W-1 sends / receives per rank (one-shot all-gather)
    ("rows_direct_c4", "rows", "collective", "direct", 4),
)


def also_sharded_leg_names(no_config4=False):
    """The `also` legs of an N > 1 default run, in the order they run (--dry-run lists them; tests/test_sharded_gloo.py asserts them)."""
    names = [] if no_config4 else ["config4_rows2d"] + [f"config4_{v[0]}" for v in SHARDED_VARIANTS]
    return names + ["config3_yelp_rows2d"]


def sharded_context(args, dev, rank, world, workload, dim, steps, warmup):
    """What every sharded variant of one workload shares: the synthetic graph (generated by every rank from the same seed and
    cross-checked), the victim whose initial tables every variant starts from, the resident triplets (broadcast from rank 0)."""
    import torch
    import torch.distributed as dist
    from recad_amd import model

    collectives = world > 1 or args.force_collectives
    B = args.batch
    d, ds = load_workload(workload, dev, args.graph, B, 1234)
    torch.manual_seed(2023)
    victim = model.from_config("victim", "lightgcn", latent_dim_rec=dim, lightGCN_n_layers=args.layers,
                               deterministic=bool(args.deterministic)).I(dataset=ds).to(dev)
    victim.graph_steps = args.graph_steps
    g = ds.graph_csr()
    users, pos, neg = resident_triplets(ds, (steps + warmup) * B)
    # every rank must see the same triplets and start from the same tables
    if collectives:
        for t in (users, pos, neg):
            dist.broadcast(t, src=0)
        for p_ in victim.parameters():
            dist.broadcast(p_.data, src=0)
        # every rank generated the workload itself (same seed): refuse to go on if the graphs differ
        chk = torch.stack([g.col.long().sum(), g.rowptr.long().sum(), torch.tensor(g.nnz, device=dev)]).double()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not bool(torch.equal(lo, hi)):
            raise RuntimeError("the ranks generated different graphs (synthetic workload not reproducible across ranks)")
    # the tables every variant starts from (the victim itself is trained by the denominators' runs, on rank 0 more than elsewhere)
    init = tuple(p_.detach().clone() for p_ in (victim.embedding_user.weight, victim.embedding_item.weight))
    return {"workload": workload, "dim": dim, "d": d, "ds": ds, "victim": victim, "g": g, "triplets": (users, pos, neg), "init": init,
            "collectives": collectives, "same_1gpu": None, "replicas": None}


def sharded_denominators(args, dev, rank, world, ctx, steps, warmup, with_replicas=True):
    """A sharded job's OWN denominators on the same workload, measured once per workload: the fused single-GPU step on rank 0
    (`same_workload_1gpu`) and N independent replicas (`same_workload_replicas`)."""
    import torch
    import torch.distributed as dist
    victim, (users, pos, neg), B, collectives = ctx["victim"], ctx["triplets"], args.batch, ctx["collectives"]

    def barrier():
        torch.cuda.synchronize()
        if collectives:
            dist.barrier()
        torch.cuda.synchronize()

    # like-for-like single-GPU reference on rank 0 (the others wait), then the same workload as N independent replicas
    if rank == 0:
        victim.reserve(max(steps, warmup) * B, B)
        s1 = min(steps, 10)
        w1 = min(warmup, 3) * B or B
        victim._run_epoch(users[:w1], pos[:w1], neg[:w1], B)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        victim._run_epoch(users[: s1 * B], pos[: s1 * B], neg[: s1 * B], B)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        ctx["same_1gpu"] = {"value": s1 * B / e1, "unit": "interactions/s", "ms_per_step": e1 / s1 * 1e3, "steps": s1,
                            "note": "the single-GPU fused path (hipGraph) on the same workload, timed on rank 0 while the other ranks wait: "
                                    "the denominator of this workload's strong-scaling legs"}
    barrier()
    if world > 1 and with_replicas:
        # what the node delivers when the perturb-retrain loop is parallelised over jobs instead of inside one -- the comparison
        # north_star's 6x on the yelp shape has to be read against (the rows modes are communication-bound there: DESIGN 6)
        if rank != 0:
            victim.reserve(max(steps, warmup) * B, B)
        sr = min(steps, 10)
        victim._run_epoch(users[: 3 * B], pos[: 3 * B], neg[: 3 * B], B)
        barrier()
        t1 = time.perf_counter()
        victim._run_epoch(users[: sr * B], pos[: sr * B], neg[: sr * B], B)
        barrier()
        tr_ = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tr_, op=dist.ReduceOp.MAX)
        ctx["replicas"] = {"value": world * sr * B / float(tr_.item()), "unit": "interactions/s", "ms_per_step": float(tr_.item()) / sr * 1e3,
                           "steps": sr, "note": f"{world} independent victims (one per GPU, fused single-GPU path) on this workload, all at once"}


def sharded_variant(args, dev, rank, world, ctx, parallel, steps, warmup, reduce=None, gather=None, chunks=None, with_eval=True,
                    with_exposed=True):
    """ONE LightGCN training job whose node rows are dealt over the process group's ranks (recad_amd/sharded.py: 1-D row
    partition with all-gathers; recad_amd/sharded2d.py: Pr x Pc tiling / column slabs with reduce-scatters), timed like the
    headline (warm-up, barrier + synchronize, K steps, barrier + synchronize, MAX over ranks), in ONE exchange form
    (`reduce`, `gather`, `chunks`; None = the trainer's default) -- plus the exposed-communication share: the same rank's step
    with every collective replaced by a copy of its own share (Grid2DLightGCN probe_rank_world: what the rank computes and
    launches), so step - compute_only is what the chunk pipeline failed to hide.  Every rank calls it; rank 0 gets the dict."""
    import torch
    import torch.distributed as dist

    collectives, B = ctx["collectives"], args.batch
    ds, g, victim, dim = ctx["ds"], ctx["g"], ctx["victim"], ctx["dim"]
    users, pos, neg = ctx["triplets"]
    N, nnz = g.n_rows, g.nnz
    reduce = reduce or args.reduce
    gather = gather or args.gather
    t_leg = time.perf_counter()

    def barrier():
        torch.cuda.synchronize()
        if collectives:
            dist.barrier()
        torch.cuda.synchronize()

    def make(probe=None):
        if parallel == "rows2d":
            from recad_amd.sharded2d import Grid2DLightGCN
            return Grid2DLightGCN(ds.n_users, ds.n_items, dim, args.layers, g, ctx["init"][0],
                                  ctx["init"][1], device=dev, grid_rows=args.grid_rows or None, reduce=reduce,
                                  deterministic=bool(args.deterministic), force_collectives=args.force_collectives and probe is None,
                                  chunks=chunks, probe_rank_world=probe)
        from recad_amd.sharded import ShardedLightGCN
        return ShardedLightGCN(ds.n_users, ds.n_items, dim, args.layers, g, ctx["init"][0],
                               ctx["init"][1], device=dev, gather=gather, chunks=chunks,
                               force_collectives=args.force_collectives, deterministic=bool(args.deterministic))

    def timed(tr):
        tr.reserve(max(steps, warmup) * B, B)

        def run(lo, n_steps):
            sl = slice(lo * B, (lo + n_steps) * B)
            return tr.train_epoch(users[sl], pos[sl], neg[sl], B)
        if warmup > 0:
            run(0, warmup)
        barrier()
        t0 = time.perf_counter()
        losses = run(warmup, steps)
        barrier()
        el = time.perf_counter() - t0
        if collectives:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, float(losses[-1])

    sharded = make()
    elapsed, last_loss = timed(sharded)
    assert np.isfinite(last_loss), "training diverged"
    ev = None
    if with_eval:
        ptr, idx = ds.train_csr_sorted()
        ev = sharded.evaluate(ptr, idx, np.array([0], dtype=np.int32), K=100, topks=(10, 20, 50, 100), reps=2)
    describe, captured = sharded.describe(), bool(getattr(sharded, "_graph", None) is not None)
    lay = getattr(sharded, "layout", None)
    grid = (lay.Pr, lay.Pc) if parallel == "rows2d" else None
    n_chunks = int(getattr(lay, "C", 0)) or None
    del sharded
    torch.cuda.empty_cache()
    exposed = None
    if with_exposed and parallel == "rows2d" and world > 1:
        # the same rank's share of the same job without a process group: every collective is a local copy of this rank's own block
        probe = make(probe=(rank, world))
        el_p, _ = timed(probe)       # (MAX over ranks like the step itself: the slowest rank's compute)
        del probe
        torch.cuda.empty_cache()
        exposed = {"compute_only_ms_per_step": el_p / steps * 1e3, "exposed_comm_ms_per_step": (elapsed - el_p) / steps * 1e3,
                   "exposed_comm_share": max(0.0, (elapsed - el_p) / elapsed),
                   "note": "compute_only = this job's step on every rank with each collective replaced by a copy of the rank's own share "
                           "(Grid2DLightGCN probe_rank_world), slowest rank; exposed = step - compute_only: the communication the chunk "
                           "pipeline did not hide"}
    seen = ranks_seen(dev, world, collectives)
    out = None
    if rank == 0:
        # the arithmetic the rows modes live under: bytes a rank must RECEIVE per propagation layer over its xGMI links
        # (7 links x ~77 GB/s one direction on MI355X) against the local SpMM time a layer needs
        blk = (N + world - 1) // world * dim * 4
        recv = (grid[0] - 1 + grid[1] - 1) * blk if parallel == "rows2d" else (world - 1) * blk
        comm_model = {"bytes_received_per_rank_and_layer": recv, "xgmi_in_GBps_assumed": 7 * 76.5,
                      "receive_floor_us_per_layer": recv / (7 * 76.5e3) if world > 1 else 0.0,
                      "layers_with_exchange_per_step": 2 * args.layers - 1,
                      "note": "a layer cannot finish before its inputs have arrived: when receive_floor_us_per_layer exceeds the local SpMM "
                              "time (same_workload_1gpu's step / (2 L) / N), the mode is communication-bound at this N"}
        out = {"workload": workload_string(ctx["workload"], ds.n_users, ds.n_items, ds.traindataSize, args.graph, nnz, dim, args.layers, B),
               "mode": parallel, "exchange": {"reduce": reduce if parallel == "rows2d" else None, "gather": gather if parallel == "rows" else None,
                                              "chunks": n_chunks, "chunks_requested": chunks},
               "parallelism": describe, "backend": (dist.get_backend() if collectives else None),
               "scaling": "strong", "value": steps * B / elapsed, "unit": "interactions/s", "ms_per_step": elapsed / steps * 1e3,
               "steps": steps, "warmup": warmup, "n_gpus": world, "ranks_seen": seen, "last_step_loss": last_loss,
               "step_captured": captured, "exposed_communication": exposed,
               "same_workload_1gpu": dict(ctx["same_1gpu"]) if ctx["same_1gpu"] else None,
               "same_workload_replicas": ctx["replicas"], "rows_comm_model": comm_model, "topk": ev,
               "seconds_total": time.perf_counter() - t_leg}
        if out["same_workload_1gpu"]:
            out["same_workload_1gpu"]["speedup_of_this_leg"] = out["value"] / out["same_workload_1gpu"]["value"]
    return out


def sharded_leg(args, dev, rank, world, workload, dim, parallel, steps, warmup, with_eval=True, with_replicas=True):
    """One workload, one sharded job in the command line's exchange form (--reduce / --gather, the trainer's default chunk count),
    with its denominators: what `--parallel rows | rows2d` prints as the top-level line and what the `*_rows2d` legs of `also` hold."""
    import torch
    ctx = sharded_context(args, dev, rank, world, workload, dim, steps, warmup)
    sharded_denominators(args, dev, rank, world, ctx, steps, warmup, with_replicas=with_replicas)
    out = sharded_variant(args, dev, rank, world, ctx, parallel, steps, warmup, with_eval=with_eval)
    del ctx
    torch.cuda.empty_cache()
    return out


class Deadline:
    """The sharded `also` legs have never met a real 8-GPU RCCL fabric in this repository's history: if one of them hangs (a
    collective that never completes), the top-level line -- already measured -- must still come out.  Every rank arms the
    same timer; on expiry rank 0 prints the line with what it has (`also.error` says what happened) and THEN every rank leaves with
    os._exit(EXIT_CODE) -- non-zero, so that the launcher / driver sees that a collective hung while the measured line is already on
    stdout (a rank stuck inside a collective cannot unwind; rank 0 fires first, the others 3 s later)."""
    EXIT_CODE = 75   # EX_TEMPFAIL

    def __init__(self, seconds, rank, emit):
        import threading
        self.done = self.printed = False
        self._t = threading.Timer(seconds + (0.0 if rank == 0 else 3.0), self._fire)
        self._t.daemon = True
        self.rank, self.emit, self.seconds = rank, emit, seconds
        self._t.start()

    def _fire(self):
        if self.done:
            return
        try:
            if self.rank == 0 and not self.printed:
                self.emit(f"the sharded legs did not finish within {self.seconds:.0f} s (deadline; the top-level line was measured before them)")
        finally:
            sys.stdout.flush()
            os._exit(self.EXIT_CODE)   # non-zero: the line (printed above by rank 0) survives, the code tells the driver a collective hung

    def cancel(self):
        self.done = True
        self._t.cancel()


# ------------------------------------------------------------------------------------------------ worker
def worker(args, traffic_live=None):
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch `python bench.py --gpus N` "
                         "(self-launching) or torch.distributed.run with --nproc-per-node equal to --gpus")
    if args.dry_run:
        return dry_run_worker(args, rank, world)

    import torch
    import torch.distributed as dist

    # gloo workers may share one GPU (the 1-GPU box check of the sharded path: --share-gpu); nccl needs one GPU per rank
    n_dev = torch.cuda.device_count()
    share = args.backend == "gloo" or args.share_gpu
    local_dev = local % max(n_dev, 1) if share else local
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    collectives = world > 1 or args.force_collectives
    if collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import recad_amd  # noqa: F401
    from recad_amd import _lib, model
    from recad_amd.evaluate import eligible_users, full_catalog_topk, hit_counts

    B = args.batch
    metric = metric_for(args)

    # ================= labelled extra mode: ONE sharded job as the top-level line (--parallel rows | rows2d)
    if collectives and args.parallel in ("rows", "rows2d"):
        leg = sharded_leg(args, dev, rank, world, args.workload, args.dim, args.parallel, args.steps, args.warmup, with_eval=not args.no_topk)
        if rank == 0:
            out = {"metric": metric, "value": leg["value"], "unit": "interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                   "data": "synthetic",
                   "config": {"workload": leg["workload"], "parallelism": leg["parallelism"], "mode": args.parallel, "backend": args.backend,
                              "graph_steps": args.graph_steps, "scatter": "ordered" if args.deterministic else "float atomics"},
                   "topk": leg["topk"], "roofline": None, "cpu_baseline": None, "ranks_seen": leg["ranks_seen"],
                   "same_workload_1gpu": leg["same_workload_1gpu"], "same_workload_replicas": leg["same_workload_replicas"],
                   "rows_comm_model": leg["rows_comm_model"], "last_step_loss": leg["last_step_loss"], "step_captured": leg["step_captured"]}
            if out["same_workload_1gpu"] is not None:
                out["same_workload_1gpu"]["speedup_of_this_run"] = out["same_workload_1gpu"].pop("speedup_of_this_leg")
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ================= the headline: the metric's workload on one GPU, or as N independent replicas (one per GPU)
    big = args.workload in ("c4s", "config4")
    d, ds = load_workload(args.workload, dev, args.graph, B, 1234 + rank)
    torch.manual_seed(2023)
    victim = model.from_config("victim", "lightgcn", latent_dim_rec=args.dim, lightGCN_n_layers=args.layers,
                               deterministic=bool(args.deterministic)).I(dataset=ds).to(dev)
    victim.graph_steps = args.graph_steps
    victim.use_lds = {"auto": "auto", "lds": True, "csr": False}[args.spmm]
    victim.fuse_layers = bool(args.fuse_layers)
    victim.use_block_list = not args.no_block_list
    g = ds.graph_csr()
    N, nnz = g.n_rows, g.nnz
    users, pos, neg = resident_triplets(ds, (args.steps + args.warmup) * B)
    host_triplets = tuple(t[: B * 64].cpu().numpy() for t in (users, pos, neg))  # cpu_baseline / parity sample
    want_parity = world == 1 and not collectives and not big and not args.no_parity
    n_par = min(args.warmup + args.steps, args.parity_steps or (25 if args.workload in ("ml1m", "tiny") else 3), 64) if want_parity else 0
    init_tables = tuple(p_.detach().cpu().numpy().copy() for p_ in (victim.embedding_user.weight, victim.embedding_item.weight)) if want_parity else None
    victim.reserve(max(args.steps, args.warmup) * B, B)   # staging + hipGraph capture/upload, before any timing
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if collectives:      # (one process, no group: the barrier is vacuous and ONE synchronize brackets the region -- the second one
            dist.barrier()   #  only re-synchronises behind a real barrier)
            torch.cuda.synchronize()

    # ---------------- what runs in front of the timed region, and why in this order (profiles/r05e_call_clock.txt, r05g_host_path.txt):
    # (1) the dominant-kernel probe: the SpMM's per-launch time by HIP events on its stream, ~13 ms of the step's own kernel
    #     (workspace buffers only, never the tables) -- it also brings the device to its working clocks: the same 20-step call
    #     takes 1 417 us of device time after 50 ms of idle, 1 350 us 5 ms after a busy period, 1 305-1 330 us right behind one;
    # (2) the W warm-up steps, through the same call as the timed ones: five steps (0.3 ms) do not move the device's clocks,
    #     but they leave the HOST path hot -- the first pass through that Python after 100 ms of other code ran every stage
    #     ~4x slower (25 us instead of 6 for the handle check alone: 150 us of a 1.4 ms region with the device waiting);
    # (3) nothing else: the reduction that copies the warm-up's losses is run once on a dummy BEFORE (1), so its code object
    #     is loaded by then (loading it between warm-up and timed call left the device idle for milliseconds).
    if want_parity and args.warmup > 0:
        torch.zeros(args.warmup, _lib.RK_LOSS_PARTIALS, device=dev).sum(dim=1).double()
        torch.cuda.synchronize()
    roofline = spmm_roofline(args, victim, N, nnz, traffic_live)
    warm_losses = None
    if args.warmup > 0:
        wp = run_steps(victim, (users, pos, neg), B, 0, args.warmup)
        if want_parity:
            warm_losses = wp.sum(dim=1).double()   # device-side copy now (the loss buffer is reused by the timed call), read back after the timing
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timed_cols = tuple(t[args.warmup * B: (args.warmup + args.steps) * B] for t in (users, pos, neg))   # (views of the resident triplets)
    barrier()
    t0 = time.perf_counter()
    partials = victim._run_epoch(*timed_cols, B)   # = run_steps(victim, (users, pos, neg), B, args.warmup, args.steps)
    t_enq = time.perf_counter()
    wait_done(stream)
    t_seen = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    tc = getattr(victim, "last_call_seconds", (t0,) * 4)
    timed_partials = partials.clone()     # (the handle's loss buffer is reused by the next call)
    timed_tables = (tuple(p_.detach().cpu().numpy().copy() for p_ in (victim.embedding_user.weight, victim.embedding_item.weight))
                    if want_parity and args.warmup + args.steps == n_par else None)
    # the GPU-side span of such a call, measured on a REPEAT of it behind the timed one (round 5 recorded its start event inside the
    # timed region: 6.6 us of host time in front of the launch with the device waiting; the span is diagnostic, the repeat costs nothing)
    span_cols = timed_cols if args.steps <= 64 else tuple(t[: 64 * B] for t in timed_cols)
    ev_a.record(stream)
    victim._run_epoch(*span_cols, B)
    ev_b.record(stream)
    wait_done(stream)
    partials = timed_partials
    timed_region = {"host_total_us": elapsed * 1e6,
                    "host_path_us": {"before_the_call": (tc[0] - t0) * 1e6, "handle_check": (tc[1] - tc[0]) * 1e6, "argument_marshalling": (tc[2] - tc[1]) * 1e6,
                                     "c_call": (tc[3] - tc[2]) * 1e6, "behind_the_c_call": (t_enq - tc[3]) * 1e6},
                    "enqueue_returns_after_us": (t_enq - t0) * 1e6, "completion_seen_after_us": (t_seen - t0) * 1e6,
                    "closing_barrier_us": (elapsed - (t_seen - t0)) * 1e6, "gpu_span_us": ev_a.elapsed_time(ev_b) * 1e3,
                    "gpu_span_of": f"a repeat of the call ({span_cols[0].numel() // B} steps) right behind the timed one, HIP events around it on the launch stream "
                                   "(includes the launch latency between the start event and the first kernel)",
                    "note": "the K timed steps are ONE call (one whole-call hipGraph replay for K <= 64); "
                            "host_total - gpu_span = launch latency + completion detection + the contract's barrier / synchronize pair",
                    "in_front": "the roofline probe (~13 ms of the dominant kernel, workspace buffers only), then the W warm-up steps, then the opening barrier: "
                                "the device is at its working clocks (the same call is 5-8 % slower behind an idle period, profiles/r05e_call_clock.txt) and the host path is hot"}
    if args.fuse_layers:
        victim.check_handoffs()
    run_losses = run_tables = None
    if want_parity:   # what the timed path produced, to be laid next to the oracle below (outside the timed region)
        tl = partials.sum(dim=1).double().cpu().numpy()
        run_losses = tl if warm_losses is None else np.concatenate([warm_losses.cpu().numpy(), tl])
        if args.warmup + args.steps == n_par:
            run_tables = timed_tables      # (taken right behind the timed call, before the span-measuring repeat trained further)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last_loss = float(partials[-1].sum().item())
    assert np.isfinite(last_loss), "training diverged"
    seen = ranks_seen(dev, world, collectives)

    # ---------------- secondary: one whole train_step() epoch, the build's device sampler included
    epoch_obj = None
    if args.workload in ("ml1m", "tiny"):
        victim.train_step(progress_bar=None)  # warm: sampler kernels, staging buffers
        barrier()
        te = time.perf_counter()
        victim.train_step(progress_bar=None)  # samples an epoch on the device, runs it, reads the losses back
        torch.cuda.synchronize()
        te = time.perf_counter() - te
        if world > 1:
            t = torch.tensor([te], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            te = float(t.item())
        n_ep = int(len(ds.generate_epoch()["users"]))  # triplets of one epoch (traindataSize draws with a positive)
        epoch_obj = {"seconds": te, "includes": "device BPR sampler + every step of one epoch + loss read-back (model.train_step())"}
        if n_ep:
            epoch_obj.update({"value": world * n_ep / te, "unit": "interactions/s", "triplets": n_ep})

    # ---------------- second half of the metric: full-catalog scoring + top-100 + HR@K
    topk = None
    if not args.no_topk:
        ptr, idx = ds.train_csr_sorted()
        targets = np.array([0], dtype=np.int32)
        ev_users = eligible_users(ptr, idx, targets)
        if args.eval_users:
            ev_users = ev_users[: args.eval_users]
        # inputs (eligible users, seen-item CSR, targets) and outputs (top-100 lists, target ranks) stay in HBM
        ev_dev = torch.as_tensor(ev_users, dtype=torch.int32, device=dev)
        ptr_dev, idx_dev = torch.as_tensor(ptr, dtype=torch.int32, device=dev), torch.as_tensor(idx, dtype=torch.int32, device=dev)
        tg_dev = torch.as_tensor(targets, dtype=torch.int32, device=dev)
        ev_chunk = max(256, min(8192, (1 << 31) // max(ds.n_items, 1)))
        # evaluate.EvalSession: buffers + plan made once; from the second run on ONE hipGraph replay per evaluation (propagate + GEMM +
        # selection + HR@k counts) -- what a loop that re-scores the victim after every epoch runs
        from recad_amd.evaluate import EvalSession
        sess = EvalSession(victim, ev_dev, ptr_dev, idx_dev, tg_dev, K=100, topks=(10, 20, 50, 100), chunk=ev_chunk)
        for _ in range(3):   # eager, capture + replay, replay
            warm = sess.run()
        int(warm["hit_counts"][0, 2].item())
        ref = full_catalog_topk(victim, ev_dev, ptr_dev, idx_dev, tg_dev, K=100, chunk=ev_chunk, to_host=False)   # the session against the plain call
        assert torch.equal(ref["top_ids"], warm["top_ids"]) and torch.equal(ref["target_rank"], warm["target_rank"]), "EvalSession != full_catalog_topk"
        assert torch.equal(hit_counts(ref["target_rank"], (10, 20, 50, 100)), warm["hit_counts"])
        del ref
        # EV_REPS complete evaluations back to back, one sync at the end
        EV_REPS = 8 if nnz < 20_000_000 else 2
        for _ in range(4 if nnz < 20_000_000 else 1):   # (the comparison above left the device idle: replays in front of the timed ones, as a
            sess.run()                                  #  loop that evaluates after every epoch finds it -- device clocks and host path hot)
        barrier()
        t1 = time.perf_counter()
        for _ in range(EV_REPS):
            res = sess.run()
        wait_done(stream)
        ev_el = (time.perf_counter() - t1) / EV_REPS
        if world > 1:
            t = torch.tensor([ev_el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ev_el = float(t.item())
        hr50 = float(res["hit_counts"][0, 2].item()) / max(len(ev_users), 1)
        t1 = time.perf_counter()  # one evaluation on an idle device through the plain call, host enqueue included (latency, not throughput)
        res = full_catalog_topk(victim, ev_dev, ptr_dev, idx_dev, tg_dev, K=100, chunk=ev_chunk, to_host=False)
        hit_counts(res["target_rank"], (10, 20, 50, 100)).cpu()
        ev_single = time.perf_counter() - t1
        deg = np.diff(ptr)
        pairs = float((ds.n_items - deg[ev_users]).sum())
        flops = 2.0 * len(ev_users) * ds.n_items * args.dim
        topk = {"value": world * len(ev_users) / ev_el, "unit": "users/s", "pair_scorings_per_s": world * pairs / ev_el,
                "eligible_users": int(len(ev_users)), "seconds": ev_el, "evaluations_timed": EV_REPS,
                "single_evaluation_seconds": ev_single, "hr@50": hr50,
                "gemm_tflops_e2e": flops / ev_el / 1e12,
                "graph_replay": bool(sess._graph is not None),
                "includes": "propagate + fp32-MFMA GEMM + seen mask + top-100 + target rank + HR@{10,20,50,100} counts (evaluate.EvalSession: one hipGraph "
                            "replay per evaluation); inputs and outputs resident in HBM"
                            + (f"; {world} replicas, one per GPU, slowest rank's time" if world > 1 else "")}

    mfma = None
    if rank == 0 and world == 1 and not args.no_topk and not big and not collectives:
        mfma = mfma_gemm_probe(dev)
    also = None
    if rank == 0 and world == 1 and not collectives and args.workload == "ml1m" and not args.no_also:
        also = {"config3_yelp": also_measure(dev, "yelp", 128, args.layers, B, parity_steps=0 if args.no_parity else 3)}
        if not args.no_also_config4:
            also["config4"] = also_measure(dev, "config4", 64, args.layers, B, steps=10, warmup=3, eval_users=65536)
    cpu = cpu_aten = parity = None
    if want_parity:
        # the oracle on the timed run's own triplets from the victim's own initial tables: parity of the timed path AND the
        # 1-thread C baseline (cpu_baseline_port) from one replay
        rep = oracle_replay(d, args.graph, args.layers, B, host_triplets, init_tables[0], init_tables[1], n_par)
        tables_from = "the timed run (warm-up + timed steps)"
        if run_tables is None:
            # a longer run: its first n_par losses are compared as they are; the tables through a second victim started from
            # the same initial tables and run through the same reserve() -> whole-call hipGraph sequence for n_par steps
            torch.manual_seed(2023)
            v2 = model.from_config("victim", "lightgcn", latent_dim_rec=args.dim, lightGCN_n_layers=args.layers,
                                   deterministic=bool(args.deterministic)).I(dataset=ds).to(dev)
            v2.graph_steps = args.graph_steps
            v2.use_lds, v2.fuse_layers = victim.use_lds, victim.fuse_layers
            v2.embedding_user.weight.data.copy_(torch.from_numpy(init_tables[0]))
            v2.embedding_item.weight.data.copy_(torch.from_numpy(init_tables[1]))
            w2 = min(args.warmup, max(n_par // 5, 0))
            v2.reserve(max(w2, n_par - w2) * B, B)
            l2 = []
            if w2:
                l2.append(run_steps(v2, (users, pos, neg), B, 0, w2).sum(dim=1).double().cpu().numpy())
            l2.append(run_steps(v2, (users, pos, neg), B, w2, n_par - w2).sum(dim=1).double().cpu().numpy())
            l2 = np.concatenate(l2)
            run_tables = tuple(p_.detach().cpu().numpy().copy() for p_ in (v2.embedding_user.weight, v2.embedding_item.weight))
            tables_from = f"a second victim from the same initial tables through the same call sequence ({w2} + {n_par - w2} steps)"
            assert np.allclose(l2, run_losses[:n_par], rtol=1e-5), "the replay leg must reproduce the timed run's losses"
            del v2
        parity = parity_object(rep, run_losses, run_tables, tables_from)
        parity["spmm"] = "lds" if victim._ws.get("lds") is not None else "csr"
        cpu = {"value": rep["steps"] * B / rep["seconds"], "unit": "interactions/s", "cores": 1, "kind": "port", "impl": "c-oracle",
               "sample": f"{rep['steps']} train steps of {B} triplets on the same graph from the victim's initial tables "
                         f"(oracle/recad_oracle.c, 1 thread, {rep['seconds']:.1f} s) -- the replay the parity object is computed from"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not big:
        cpu_aten = cpu_baseline_aten(d, args.graph, args.dim, args.layers, B, host_triplets)

    out = None
    if rank == 0:
        par = "single GPU" if world == 1 else f"{world} independent victim replicas, one per GPU (no data-path collective)"
        out = {
            "metric": metric,
            "value": world * args.steps * B / elapsed, "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_string(args.workload, ds.n_users, ds.n_items, ds.traindataSize, args.graph, nnz, args.dim, args.layers, B),
                       "parallelism": par, "mode": "replicas" if world > 1 else "single",
                       "backend": args.backend if collectives else None,
                       "graph_steps": args.graph_steps, "scatter": "ordered" if args.deterministic else "float atomics"},
            "timed_region": timed_region, "epoch_with_sampler": epoch_obj, "topk": topk, "roofline": roofline, "cpu_baseline": cpu_aten,
            "cpu_baseline_port": cpu, "parity": parity, "also": also, "mfma_gemm": mfma, "ranks_seen": seen, "last_step_loss": last_loss,
        }

    # ================= N > 1: the strong-scaling legs (ONE job sharded over the ranks) go into `also`, under a deadline
    if world > 1 and not args.no_also_sharded:
        del victim
        torch.cuda.empty_cache()
        legs = {}

        def emit(err=None):
            if rank == 0:
                out["also"] = dict(legs)
                if err:
                    out["also"]["error"] = err
                print(json.dumps(out), flush=True)

        dl = Deadline(args.also_timeout, rank, emit)

        def leg_of(name, fn):
            try:
                leg = fn()
                if rank == 0:
                    legs[name] = leg
            except Exception as e:   # noqa: BLE001 -- a failed leg is recorded; the ranks re-synchronise at the next leg's first collective
                legs[name] = {"error": f"{type(e).__name__}: {e}"}

        # config 4 first (north_star's row-sharded shape): the default form with its denominators and the user-sharded evaluation, then
        # the SAME job in the three exchange forms x {1, 4} chunks (training step only, each with speedup_of_this_leg, ranks_seen and
        # its exposed-communication share), so that the first run on real xGMI decides reduce_scatter_tensor vs the one-shot
        # all-to-all vs the 1-D one-shot all-gather, and the chunk count, by itself; then the yelp shape in the default form.
        if not args.no_also_config4:
            ctx = None
            try:
                ctx = sharded_context(args, dev, rank, world, "config4", 64, 10, 3)
                sharded_denominators(args, dev, rank, world, ctx, 10, 3)
            except Exception as e:   # noqa: BLE001
                legs["config4_rows2d"] = {"error": f"{type(e).__name__}: {e}"}
                ctx = None
            if ctx is not None:
                leg_of("config4_rows2d", lambda: sharded_variant(args, dev, rank, world, ctx, "rows2d", 10, 3))
                for suffix, par_, red_, gat_, ch_ in SHARDED_VARIANTS:
                    leg_of("config4_" + suffix, lambda: sharded_variant(args, dev, rank, world, ctx, par_, 6, 2, reduce=red_, gather=gat_,
                                                                        chunks=ch_, with_eval=False))
                if rank == 0:
                    ab = {k: {"ms_per_step": v["ms_per_step"], "speedup_vs_1gpu": (v.get("same_workload_1gpu") or {}).get("speedup_of_this_leg"),
                              "exposed_comm_share": (v.get("exposed_communication") or {}).get("exposed_comm_share")}
                          for k, v in legs.items() if k.startswith("config4_") and isinstance(v, dict) and "ms_per_step" in v}
                    if ab:
                        legs["config4_exchange_ab"] = dict(ab, best=min(ab, key=lambda k: ab[k]["ms_per_step"]))
                del ctx
                torch.cuda.empty_cache()
        leg_of("config3_yelp_rows2d", lambda: sharded_leg(args, dev, rank, world, "yelp", 128, "rows2d", 20, 5))
        emit()
        dl.printed = True
        dist.barrier()       # (still under the deadline: a rank whose leg failed half-way may never arrive)
        dl.cancel()
        dist.destroy_process_group()
        return
    if rank == 0:
        print(json.dumps(out), flush=True)
    if collectives:
        dist.barrier()
        dist.destroy_process_group()


def workflow_bench(args):
    """SURVEY 8d config 3's timed quantity: Normal.execute (recad/workflow/normal.py:193-225) = train(clean) + attacker +
    inject (recad/dataset/implicit.py:482-494) + fresh victim on the poisoned graph + train(poisoned) + the two evaluations,
    with the host-side costs of the perturb-retrain loop itemised: dataset / graph rebuild, SpMM schedule or LDS plan build
    (first handle of the new victim), hipGraph capture."""
    import torch
    from recad_amd import dataset, model, synth, workflow

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    def sync_time():
        torch.cuda.synchronize()
        return time.perf_counter()

    t = sync_time()
    big = args.workload in ("c4s", "config4")
    if big:
        dd = synth.make_device(args.workload, dev)
        d = {k: (tuple(x.cpu().numpy() for x in v) if isinstance(v, tuple) else v) for k, v in dd.items()}
        del dd
    else:
        d = synth.make(args.workload)
    t_synth = sync_time() - t
    t = sync_time()
    ds = dataset.from_config("implicit", args.workload, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"],
                             need_graph=True, device=dev, graph_source=args.graph, pairwise_batch_size=args.batch, seed=1234)
    g = ds.graph_csr()
    t_dataset = sync_time() - t
    torch.manual_seed(2023)
    victim = model.from_config("victim", "lightgcn", latent_dim_rec=args.dim, lightGCN_n_layers=args.layers)
    wf = workflow.from_config("no defense", victim_data=ds, attack_data=None, victim=victim,
                              attacker=workflow.RandomAttack(ds.n_items, attack_num=50, filler_num=36, seed=2),
                              rec_epoch=args.rec_epoch, attack_epoch=0, device=dev, target_id_list=[0])
    items = {}

    def staged(name, fn):
        t0 = sync_time()
        out = fn()
        items[name] = sync_time() - t0
        return out

    t_all = sync_time()
    wf.victim = wf.victim.to(dev)
    staged("clean_handle_schedule_build_s", lambda: wf.victim._ensure_handle())
    staged("train_clean_s", lambda: wf.normal_train(wf.victim, args.rec_epoch))
    fake = staged("attacker_generate_fake_s", lambda: wf.attacker.generate_fake(**wf.info_describe()))
    fake_ds = staged("inject_dataset_rebuild_s", lambda: ds.inject_data("explicit", fake, filter_num=wf.c["filter_num"]))
    staged("inject_graph_rebuild_s", lambda: fake_ds.graph_csr())
    fake_victim = staged("reset_instantiate_s", lambda: wf.victim.reset().I(dataset=fake_ds).to(dev))
    staged("poisoned_handle_schedule_build_s", lambda: fake_victim._ensure_handle())
    staged("train_poisoned_s", lambda: wf.normal_train(fake_victim, args.rec_epoch))
    res = staged("evaluate_2x_s", lambda: wf.normal_evaluate(wf.victim, fake_victim, ds, [0], wf.c["topks"]))
    total = sync_time() - t_all
    retrain = items["inject_dataset_rebuild_s"] + items["inject_graph_rebuild_s"] + items["reset_instantiate_s"] + \
        items["poisoned_handle_schedule_build_s"] + items["train_poisoned_s"]
    host = items["inject_dataset_rebuild_s"] + items["inject_graph_rebuild_s"] + items["poisoned_handle_schedule_build_s"]
    n_ep = int(len(ds.generate_epoch()["users"]))
    out = {"metric": "config-3 workflow: train(clean) + inject 50 users + train(poisoned) + 2 x evaluation, seconds",
           "value": total, "unit": "s", "higher_is_better": False, "n_gpus": 1, "data": "synthetic", "dtype": "f32",
           "config": {"workload": f"LightGCN victim, {args.workload}-shaped synthetic {ds.n_users}x{ds.n_items}, {ds.traindataSize} train edges, "
                                  f"graph={args.graph} (nnz {g.nnz}), dim={args.dim}, layers={args.layers}, batch={args.batch}, "
                                  f"rec_epoch={args.rec_epoch}, random attacker 50 x 36, device sampler included",
                      "spmm": "lds" if wf.victim._ws.get("lds") is not None else "csr"},
           "items_s": items, "synth_s": t_synth, "dataset_and_graph_build_s": t_dataset,
           "retrain_s": retrain, "retrain_host_rebuild_s": host, "retrain_host_share": host / retrain,
           "triplets_per_epoch": n_ep, "train_interactions_per_s": 2 * args.rec_epoch * n_ep / (items["train_clean_s"] + items["train_poisoned_s"]),
           "results": {k: (float(v) if isinstance(v, (int, float)) else v) for k, v in res.items()}}
    print(json.dumps(out), flush=True)


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.workflow:
        return workflow_bench(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_workers(args, argv))
    traffic_live = None
    if args.gpus == 1 and not args.dry_run and not args.no_live_traffic and not args.force_collectives:
        # BEFORE this process touches the GPU: the PMC passes are child processes (rocprofv3 ... -- python3 probe)
        try:
            traffic_live = live_traffic_probe(args)
        except Exception as e:   # noqa: BLE001 -- the bench line must come out; the roofline says why traffic is not live
            traffic_live = {"error": f"{type(e).__name__}: {e}"}
    worker(args, traffic_live)


if __name__ == "__main__":
    main()
