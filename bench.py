#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its config[1]: LightGCN victim, ml1m-shaped synthetic
interactions (5950 x 3702, ~470K train edges, intended train graph), dim=64, 3 layers, B=1024.

A "step" = one pass of the hot path over one minibatch: forward propagation (3 SpMM),
gather + BPR softplus + L2 reg, backward (3 SpMM), dense Adam -- K steps are timed with the
pre-sampled triplets already resident in HBM.  After the timed region the full-catalog
scoring + top-100 + HR@K pass is timed too (the second half of the metric) and reported in
the same JSON line under "topk".

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

N>1: every rank retrains its own victim replica on its own poisoned copy of the dataset
(the unit the perturb-retrain loop parallelises at ml1m scale; no data-path collective), so
scaling is "weak"; value = triplets of all ranks / max-over-ranks time.  DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # same guide: v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=458)   # one ml1m epoch = ceil(468649/1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="ml1m", choices=["ml1m", "yelp", "tiny", "c4s", "config4"])
    ap.add_argument("--graph", default="train", choices=["train", "reference"],
                    help="adjacency from the train edges (BASELINE '~470K edges') or the reference's as-is test-edge graph")
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--graph-steps", type=int, default=8, help="train steps per hipGraph replay (0 = plain launches)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="oracle steps for cpu_baseline (0 = auto, ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-users", type=int, default=0, help="evaluate only the first n eligible users (0 = all)")
    ap.add_argument("--parallel", default="replicas", choices=["replicas", "rows"],
                    help="N>1: one victim replica per GPU (weak) or node rows sharded over the GPUs with RCCL all-gathers (strong)")
    return ap.parse_args()


def cpu_baseline(d, graph, dim, layers, batch, triplets, n_steps_req):
    """The CPU oracle (a 1-thread C port of the reference path) on a bounded sample of the same
    workload: the first few train steps of the same epoch.  Checker code, timed -- not shipped."""
    from oracle import oracle as orc

    U, I = d["n_users"], d["n_items"]
    ptr, idx = d["train"] if graph == "train" else d["test"]
    csr = orc.build_norm_adj(U, I, ptr.astype(np.int32), idx.astype(np.int32))
    rng = np.random.default_rng(2023)
    user = (rng.standard_normal((U, dim), dtype=np.float32) * 0.1).astype(np.float32)
    item = (rng.standard_normal((I, dim), dtype=np.float32) * 0.1).astype(np.float32)
    st = orc.AdamState(user.shape, item.shape)
    users, pos, neg = triplets
    avail = len(users) // batch
    t0 = time.perf_counter()
    orc.lightgcn_step(csr, user, item, st, users[:batch], pos[:batch], neg[:batch], layers)
    one = time.perf_counter() - t0
    n = int(min(avail - 1, n_steps_req or max(2, 15.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    for s in range(1, n + 1):
        orc.lightgcn_step(csr, user, item, st, users[s * batch:(s + 1) * batch], pos[s * batch:(s + 1) * batch],
                          neg[s * batch:(s + 1) * batch], layers)
    el = time.perf_counter() - t0
    return {"value": n * batch / el, "unit": "interactions/s", "cores": 1, "kind": "port",
            "sample": f"{n} train steps of {batch} triplets on the same graph (oracle/recad_oracle.c, 1 thread, {el:.1f} s)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import recad_amd  # noqa: F401
    from recad_amd import _lib, dataset, model, synth
    from recad_amd.evaluate import eligible_users, full_catalog_topk, hit_counts

    # ---------------- workload: synthetic interactions of the named shape, resident on the GPU
    d = synth.make(args.workload)
    ds = dataset.from_config("implicit", args.workload, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"],
                             need_graph=True, device=dev, graph_source=args.graph, pairwise_batch_size=args.batch,
                             seed=1234 + rank)
    torch.manual_seed(2023)
    victim = model.from_config("victim", "lightgcn", latent_dim_rec=args.dim, lightGCN_n_layers=args.layers).I(dataset=ds).to(dev)
    victim.graph_steps = args.graph_steps
    g = ds.graph_csr()
    N, nnz = g.n_rows, g.nnz
    B = args.batch
    need = (args.steps + args.warmup) * B
    cols = [[], [], []]
    have = 0
    while have < need:
        ep = ds.generate_epoch()
        for c, k in zip(cols, ("users", "positive_items", "negative_items")):
            c.append(ep[k])
        have += len(ep["users"])
    users, pos, neg = (torch.cat(c)[:need].contiguous() for c in cols)
    host_triplets = tuple(t[: B * 160].cpu().numpy() for t in (users, pos, neg))  # cpu_baseline sample

    sharded = None
    if args.parallel == "rows":
        # every rank must see the same triplets and start from the same tables
        from recad_amd.sharded import ShardedLightGCN
        if world > 1:
            for t in (users, pos, neg):
                dist.broadcast(t, src=0)
            for p_ in victim.parameters():
                dist.broadcast(p_.data, src=0)
        gg = ds.graph_csr()
        sharded = ShardedLightGCN(ds.n_users, ds.n_items, args.dim, args.layers,
                                  (gg.rowptr.cpu().numpy(), gg.col.cpu().numpy(), gg.val.cpu().numpy()),
                                  victim.embedding_user.weight, victim.embedding_item.weight, device=dev)

    def run(lo, n_steps):
        sl = slice(lo * B, (lo + n_steps) * B)
        if sharded is not None:
            return sharded.train_epoch(users[sl], pos[sl], neg[sl], B)
        return victim._run_epoch(users[sl], pos[sl], neg[sl], B)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(0, args.warmup)
    barrier()
    t0 = time.perf_counter()
    partials = run(args.warmup, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    last_loss = float(partials[-1].sum().item()) if sharded is None else float(partials[-1])
    if sharded is not None:  # hand the trained tables back to the victim for the evaluation leg
        tu, ti = sharded.tables()
        victim.embedding_user.weight.data.copy_(tu)
        victim.embedding_item.weight.data.copy_(ti)
    work_ranks = 1 if sharded is not None else world  # rows: all ranks work on ONE training job
    assert np.isfinite(last_loss), "training diverged"

    # ---------------- secondary: one whole train_step() epoch, the build's device sampler included
    epoch_obj = None
    if sharded is None and args.workload in ("ml1m", "tiny"):
        victim.train_step(progress_bar=None)  # warm: sampler kernels, staging buffers
        barrier()
        te = time.perf_counter()
        victim.train_step(progress_bar=None)  # samples an epoch on the device, runs it, reads the losses back
        torch.cuda.synchronize()
        te = time.perf_counter() - te
        n_ep = int(len(ds.generate_epoch()["users"]))  # triplets of one epoch (traindataSize draws with a positive)
        epoch_obj = {"seconds": te, "includes": "device BPR sampler + every step of one epoch + loss read-back (model.train_step())"}
        if n_ep:
            epoch_obj.update({"value": world * n_ep / te, "unit": "interactions/s", "triplets": n_ep})

    # ---------------- dominant kernel: the CSR SpMM; per-launch time by HIP events on its stream
    stream = torch.cuda.current_stream()
    reps = 200
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h = victim._ensure_handle()
    for _ in range(10):
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
    ev0.record(stream)
    for _ in range(reps):
        _lib.check(_lib.lib().rk_lightgcn_propagate(h, _lib.stream_ptr()), "propagate")
    ev1.record(stream)
    torch.cuda.synchronize()
    spmm_ms = ev0.elapsed_time(ev1) / (reps * args.layers)
    spmm_bytes = 8 * nnz + 4 * (N + 1) + 2 * 4 * N * args.dim  # A once, X once, Y once
    achieved = spmm_bytes / (spmm_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_spmm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{args.workload}_{args.graph}_d{args.dim}")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": f"spmm_csr_kernel<{args.dim}>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": spmm_bytes, "avg_launch_us": spmm_ms * 1e3,
                "gather_bytes_per_launch": 8 * nnz + 4 * nnz * args.dim + 4 * N * args.dim,
                "note": "working set is L2/MALL resident at this size; time includes the inter-kernel boundary",
                # what actually bounds the kernel: no-reuse row gathers served by L2 (MI355X_MICROARCH.md, 'Indexed rows':
                # 16.8-18.8 TB/s chip-wide for L2-served row gathers; 7.4-8.6 TB/s from the Infinity Cache)
                "gather_achieved_GBps": (8 * nnz + 4 * nnz * args.dim + 4 * N * args.dim) / (spmm_ms * 1e-3) / 1e9,
                "gather_ceiling_GBps": 18800.0 if 4 * N * args.dim <= 32 * 2 ** 20 else 8600.0}

    # ---------------- second half of the metric: full-catalog scoring + top-100 + HR@K
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
    warm = full_catalog_topk(victim, ev_dev, ptr_dev, idx_dev, tg_dev, K=100, chunk=ev_chunk, to_host=False)  # warm (allocator, torch kernels)
    int(hit_counts(warm["target_rank"], (10, 20, 50, 100))[0, 2].item())
    del warm
    # EV_REPS complete evaluations back to back (each: propagate + GEMM + select + HR reduction), one sync at the end
    EV_REPS = 8
    barrier()
    t1 = time.perf_counter()
    for _ in range(EV_REPS):
        res = full_catalog_topk(victim, ev_dev, ptr_dev, idx_dev, tg_dev, K=100, chunk=ev_chunk, to_host=False)
        hits_t = hit_counts(res["target_rank"], (10, 20, 50, 100))
    torch.cuda.synchronize()
    ev_el = (time.perf_counter() - t1) / EV_REPS
    hr50 = float(hits_t[0, 2].item()) / max(len(ev_users), 1)
    t1 = time.perf_counter()  # one evaluation on an idle device, host enqueue included (latency, not throughput)
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
            "includes": "propagate + fp32-MFMA GEMM + seen mask + top-100 + target rank + HR@{10,20,50,100} counts; inputs and outputs resident in HBM"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(d, args.graph, args.dim, args.layers, B, host_triplets, args.cpu_steps)

    if rank == 0:
        out = {
            "metric": "BPR train interactions/sec + full-catalog top-K scorings/sec, LightGCN ml1m dim=64",
            "value": work_ranks * args.steps * B / elapsed, "unit": "interactions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if sharded is None else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LightGCN victim, {args.workload}-shaped synthetic {ds.n_users}x{ds.n_items}, "
                                   f"{ds.traindataSize} train edges, graph={args.graph} (nnz {nnz}), dim={args.dim}, "
                                   f"layers={args.layers}, batch={B}, Adam lr 1e-3, lambda 1e-4",
                       "parallelism": ("single GPU" if world == 1 else "1 victim replica per GPU" if sharded is None
                                       else f"node rows sharded over {world} GPUs, 2L all-gathers/step (RCCL)"),
                       "graph_steps": args.graph_steps},
            "epoch_with_sampler": epoch_obj, "topk": topk, "roofline": roofline, "cpu_baseline": cpu, "last_step_loss": last_loss,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
