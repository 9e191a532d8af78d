"""Review item 8, measured (probe, not product): the panelled LDS SpMM with register accumulators (scripts/spmm_panel_probe2.hip)
against the product's row-gather launch on the yelp-shaped graph, d = 128.  The plan is built here with numpy.
    python3 scripts/spmm_panel_probe2.py [workload=yelp] [dim=128] [B0=6] [B1=10] [cap=1024] [reps=20]"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import dataset, synth
from recad_amd.sharded import HipOps

def build_plan(rp, col, U, I, B, CAP, PANEL_MAX=8800):
    """-> units [n][16] int32, tables int32, stream uint32 (as int64), split rows, JM, stats"""
    N = U + I
    deg = np.diff(rp)
    def balanced(lo, hi, nb):
        tot = rp[hi] - rp[lo]
        cuts = [lo]
        for k in range(1, nb):
            cuts.append(int(np.searchsorted(rp[lo:hi + 1], rp[lo] + tot * k // nb)) + lo)
        cuts.append(hi)
        return cuts


    units, tables, streams, split_rows = [], [], [], []
    n_words = 0          # stream position in units of 64 words
    JMAX_used = 0
    stats = []
    for h, (lo, hi, src0, n_src, nb) in enumerate(((0, U, U, I, B[0]), (U, N, 0, U, B[1]))):
        P = -(-n_src // PANEL_MAX)
        PR = -(-n_src // P)
        cuts = balanced(lo, hi, nb)
        for b in range(nb):
            r_lo, r_hi = cuts[b], cuts[b + 1]
            rows = np.arange(r_lo, r_hi)
            # virtual rows: pieces of <= CAP consecutive nonzeros
            npc = np.maximum(1, -(-deg[rows] // CAP))
            vr = np.repeat(rows, npc)
            k = np.concatenate([np.arange(n) for n in npc])
            e0 = rp[vr] + k * CAP
            e1 = np.minimum(e0 + CAP, rp[vr + 1])
            split = np.repeat(npc > 1, npc)
            split_rows.append(rows[npc > 1])
            nv = len(vr)
            # per virtual row and panel: count and first entry (columns are sorted inside a row)
            ent_row = np.repeat(np.arange(nv), e1 - e0)
            ent_idx = np.concatenate([np.arange(a, b_) for a, b_ in zip(e0, e1)]) if nv else np.zeros(0, np.int64)
            ent_pan = (col[ent_idx] - src0) // PR
            cnt = np.zeros((nv, P), np.int64)
            np.add.at(cnt, (ent_row, ent_pan), 1)
            first = e0[:, None] + np.concatenate([np.zeros((nv, 1), np.int64), np.cumsum(cnt, 1)[:, :-1]], 1)
            order = np.argsort(-(e1 - e0), kind="stable")
            J = -(-nv // 1024)
            JMAX_used = max(JMAX_used, J)
            pad = J * 1024 - nv
            order = np.concatenate([order, np.full(pad, -1)])
            # sorted groups of 64 dealt round-robin over the 16 waves: group q -> wave q % 16, slot q // 16
            grp = order.reshape(J * 16, 64)
            slot_of, wave_of = np.arange(J * 16) // 16, np.arange(J * 16) % 16
            rowtab = np.full((J, 1024), -1, np.int64)
            for q in range(J * 16):
                ids = grp[q]
                ok = ids >= 0
                rr = np.where(ok, vr[np.maximum(ids, 0)] | (np.where(split[np.maximum(ids, 0)], 1, 0) << 30), -1)
                rowtab[slot_of[q], wave_of[q] * 64: wave_of[q] * 64 + 64] = rr
            gcnt = np.where(grp[:, :, None] >= 0, cnt[np.maximum(grp, 0)], 0)          # [groups, 64, P]
            gfirst = first[np.maximum(grp, 0)]
            steps = -(-gcnt.max(1) // 2)                                               # [groups, P] pair-steps
            hdr = np.zeros((P, 16, 1 + 20), np.int64)
            unit_words = 0
            real = padded = 0
            for p in range(P):
                for w in range(16):
                    qs = np.arange(J) * 16 + w                                          # this wave's groups, slot order
                    n_st = steps[qs, p]
                    tot = int(n_st.sum())
                    hdr[p, w, 0] = n_words + unit_words
                    hdr[p, w, 1:1 + J] = n_st
                    blockw = np.full((tot * 2, 64), PR, np.int64)                       # entries: [step * 2 + half][lane]; pad = the zero row
                    base = np.concatenate([[0], np.cumsum(n_st)[:-1]]) * 2
                    for j, q in enumerate(qs):
                        c, f = gcnt[q, :, p], gfirst[q, :, p]
                        for t in range(int(c.max()) if len(c) else 0):
                            m = c > t
                            blockw[base[j] + t, m] = col[f[m] + t] - src0 - p * PR
                        real += int(c.sum())
                    padded += tot * 2 * 64
                    if os.environ.get("PROBE_CONFLICT_FREE"):
                        # timing experiment only (results are WRONG): lane l reads bank class l % 16, so no ds_read_b128 of a
                        # 16-lane service group meets a bank conflict -- what a coloured schedule would buy
                        real_e = blockw != PR
                        blockw = np.where(real_e, np.minimum((blockw // 16) * 16 + (np.arange(64) % 16)[None, :], PR - 1), blockw)
                    words = (blockw[0::2] | (blockw[1::2] << 16)).astype(np.uint32)     # [tot][64]
                    tot4 = -(-tot // 4) * 4                                              # whole chunks: a lane reads 4 pair-steps as one 16-byte word
                    if tot4 > tot:
                        words = np.concatenate([words, np.full((tot4 - tot, 64), PR | (PR << 16), np.uint32)])
                    streams.append(words.reshape(tot4 // 4, 4, 64).transpose(0, 2, 1).reshape(-1))   # [chunk][lane][4]
                    unit_words += tot4
            n_words += unit_words
            stats.append((h, b, nv, J, real, padded))
            units.append([src0, n_src, PR, P, J, 0, 0] + [0] * 9)
            tables.append((hdr, rowtab))
    JM = 8 if JMAX_used <= 8 else 12 if JMAX_used <= 12 else 16 if JMAX_used <= 16 else 20
    tab_flat, ofs = [], 0
    for un, (hdr, rowtab) in zip(units, tables):
        P, J = un[3], un[4]
        hd = np.zeros((P, 16, 1 + JM), np.int64)
        hd[:, :, : 1 + min(JM, 20)] = hdr[:, :, : 1 + JM]
        un[5] = ofs
        tab_flat.append(hd.reshape(-1))
        ofs += hd.size
        rt = np.full((JM, 1024), -1, np.int64)
        rt[:J] = rowtab
        un[6] = ofs
        tab_flat.append(rt.reshape(-1))
        ofs += rt.size
    return np.asarray(units, np.int32), np.concatenate(tab_flat).astype(np.int32), np.concatenate(streams).astype(np.int64), np.concatenate(split_rows).astype(np.int32), JM, stats


if __name__ != "__main__":
    raise SystemExit
name = sys.argv[1] if len(sys.argv) > 1 else "yelp"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = [int(sys.argv[3]) if len(sys.argv) > 3 else 6, int(sys.argv[4]) if len(sys.argv) > 4 else 10]
CAP = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
PANEL_MAX = 8800
dev = torch.device("cuda:0")
d = synth.make(name)
ds = dataset.from_config("implicit", name, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source="train")
g = ds.graph_csr()
U, I = ds.n_users, ds.n_items
N = U + I
rp = g.rowptr.cpu().numpy().astype(np.int64)
col = g.col.cpu().numpy().astype(np.int64)
deg = np.diff(rp)
dinv = np.where(deg > 0, 1.0 / np.sqrt(np.maximum(deg, 1).astype(np.float64)), 0.0).astype(np.float32)


units, tables_np, stream_np, split, JM, stats = build_plan(rp, col, U, I, B, CAP, PANEL_MAX)
units_t = torch.tensor(units, device=dev)
tables_t = torch.tensor(tables_np, device=dev)
stream_t = torch.tensor(stream_np, device=dev).to(torch.int32)   # (uint32 bit patterns)
split_t = torch.tensor(split if len(split) else np.zeros(1, np.int32), device=dev)
dinv_t = torch.tensor(dinv, device=dev)
real = sum(s[4] for s in stats)
padded = sum(s[5] for s in stats)
for h in (0, 1):
    r_, p_ = sum(s[4] for s in stats if s[0] == h), sum(s[5] for s in stats if s[0] == h)
    print(f"half {h}: {B[h]} row blocks, J <= {max(s[3] for s in stats if s[0] == h)}, {units[0 if h == 0 else B[0]][3]} panels of {units[0 if h == 0 else B[0]][2]} rows; "
          f"nonzeros {r_}, lock-step entries {p_} (x {p_ / max(r_, 1):.2f})", flush=True)

so = os.path.join("recad_amd", "lib", "libspmm_panel_probe2.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-munsafe-fp-atomics", "scripts/spmm_panel_probe2.hip", "-o", so])
lib = ctypes.CDLL(so)
lib.panel_spmm.restype = ctypes.c_int
lib.panel_spmm.argtypes = [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
ns = dim // 4
x = torch.randn(N, dim, device=dev) * 0.1
xs = x.view(N, ns, 4).permute(1, 0, 2).contiguous()
ys = torch.full_like(xs, float("nan"))
lds_bytes = (max(un[2] for un in units) + 1) * 16
stream = torch.cuda.current_stream().cuda_stream
n_wg = len(units) * ns
stamps = torch.zeros(n_wg * 4, dtype=torch.int64, device=dev)


def run(st=None):
    rc = lib.panel_spmm(JM, units_t.data_ptr(), len(units), tables_t.data_ptr(), stream_t.data_ptr(), dinv_t.data_ptr(), N, ns, xs.data_ptr(), ys.data_ptr(),
                        split_t.data_ptr(), len(split), lds_bytes, st, stream)
    assert rc == 0, rc


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ops = HipOps()
slab = ops.make_slab(g.rowptr, g.col, g.val, dev)
y = torch.empty_like(x)
ref_us = timed(lambda: ops.spmm(slab, x, y=y))
us = timed(run)
got = ys.permute(1, 0, 2).reshape(N, dim)
err = float((got - y).abs().max() / y.abs().max())
run(stamps.data_ptr())
torch.cuda.synchronize()
st = stamps.view(n_wg, 4).cpu().numpy().astype(np.float64) / 100.0
life = st[:, 3] - st[:, 0]
unit_of = np.arange(n_wg) % len(units)
print(f"{name} d={dim}: product row-gather launch {ref_us:.1f} us | panelled LDS form {us:.1f} us = {ref_us / us:.2f} x  (relative difference {err:.1e}; "
      f"{n_wg} workgroups, {lds_bytes} B of LDS, J <= {JM}, {len(split)} split rows; lock-step entries x {padded / real:.2f})", flush=True)
for h, sel in ((0, unit_of < B[0]), (1, unit_of >= B[0])):
    print(f"  half {h} workgroups: lifetime mean {life[sel].mean():.1f} us, max {life[sel].max():.1f}; gather (from the last staged panel on) {np.mean(st[sel, 2] - st[sel, 1]):.1f} us; "
          f"rows {np.mean(st[sel, 3] - st[sel, 2]):.1f} us", flush=True)
print(f"  kernel span by stamps {st[:, 3].max() - st[:, 0].min():.1f} us", flush=True)
print(json.dumps({"workload": name, "dim": dim, "B": B, "cap": CAP, "product_us": ref_us, "panel_us": us, "rel_diff": err, "pad": padded / real}))
