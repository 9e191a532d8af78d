"""Review item 8, priced before building (CPU only): the LDS SpMM "one size up" for the yelp shape keeps a workgroup's output rows as
REGISTER accumulators (a lane owns `J` rows, 4 floats each) while the source class streams through LDS in panels of <= 8 832
rows (138 KB of 16-byte slice rows).  A lane's J rows are walked slot by slot and the 64 lanes of a wave run in lockstep, so a
(slot, panel) costs the MAXIMUM count over its 64 rows.  This script counts those padded steps on the synthetic yelp graph for
the two halves, rows grouped by total degree ("deg") or by a k-d split over the per-panel counts ("kd"), counts rounded up to
`quant` (the stream's load granule).  ratio = padded / real nonzeros; the last figure is the heaviest wave against the mean
(rows are dealt to waves in sorted order: a long item row is one lane's job unless it is split).
    python3 scripts/lds_panel_pad_sim.py  ->  profiles/r04_lds_panel_pad_sim.txt"""
import sys, numpy as np, time
sys.path.insert(0, ".")
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import synth
d = synth.make("yelp")
rp, ci = d["train"][0], d["train"][1]
U = len(rp)-1; I = int(ci.max())+1
print(U, I, len(ci))
import scipy.sparse as sp
R = sp.csr_matrix((np.ones(len(ci),np.int8), ci, rp), shape=(U, I))
RT = R.T.tocsr()
PANEL_ROWS = 8832   # 138 KB / 16 B minus a zero row
def sim(M, nblocks, label, quant=2, strategy="deg"):
    # M: dest x src csr ; panels over src columns
    nd, ns = M.shape
    P = -(-ns // PANEL_ROWS)
    pr = -(-ns // P)
    # per-row per-panel counts
    coo = M.tocoo()
    cnt = np.zeros((nd, P), np.int32)
    np.add.at(cnt, (coo.row, coo.col // pr), 1)
    real = cnt.sum()
    tot_steps = 0; wave_steps=[]
    bs = -(-nd // nblocks)
    for b in range(nblocks):
        c = cnt[b*bs:(b+1)*bs]
        n = len(c)
        if strategy == "deg":
            order = np.argsort(-c.sum(1), kind="stable")
        elif strategy == "kd":
            # recursive k-d split: levels over panels in order of total weight
            pw = np.argsort(-c.sum(0))
            groups = [np.arange(n)]
            ngroups_target = -(-n // 64)
            lv = 0
            while len(groups) < ngroups_target and lv < len(pw):
                # split factor per level
                remaining_levels = len(pw) - lv
                f = max(2, int(round((ngroups_target / len(groups)) ** (1.0 / remaining_levels))))
                new = []
                for g in groups:
                    if len(g) <= 64: new.append(g); continue
                    o = g[np.argsort(-c[g, pw[lv]], kind="stable")]
                    k = min(f, -(-len(g)//64))
                    # split into k pieces with sizes multiple of 64
                    per = -(-(-(-len(g)//64)) // k) * 64
                    for s in range(0, len(o), per): new.append(o[s:s+per])
                groups = new; lv += 1
            # final: inside each group sort by total
            order = np.concatenate([g[np.argsort(-c[g].sum(1), kind="stable")] for g in groups])
        cs = c[order]
        pad = (-len(cs)) % 1024
        cs = np.vstack([cs, np.zeros((pad, P), np.int32)])
        J = len(cs)//1024
        g = cs.reshape(J, 16, 64, P)        # slot, wave, lane, panel
        mx = g.max(2)                        # slot, wave, panel
        mx = -(-mx // quant) * quant
        tot_steps += mx.sum() * 64
        wave_steps.append(mx.sum((0,2)))     # per wave
    ws = np.concatenate(wave_steps)
    print(f"{label}: P={P} blocks={nblocks} J={J} strategy={strategy} quant={quant}: real {real} padded {tot_steps} ratio {tot_steps/real:.2f}; per-wave steps mean {ws.mean():.0f} max {ws.max()} (block max/mean {ws.max()/ws.mean():.2f})")
for strat in ("deg","kd"):
    for q in (1,2,4):
        sim(R, 4, "user rows <- item panels", q, strat)
        sim(RT, 4, "item rows <- user panels", q, strat)
