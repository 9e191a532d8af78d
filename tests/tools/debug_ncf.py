import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from recad_amd import dataset, model, synth
from tests import _golden as G
dev = torch.device('cuda:0')
f, L = int(sys.argv[1]), int(sys.argv[2])
dd = synth.make("tiny")
ds = dataset.from_config("implicit", "tiny", train_csr=dd["train"], valid_csr=dd["valid"], test_csr=dd["test"], need_graph=False, device=dev, sample="pointwise", seed=f * 10 + L, pointwise_batch_size=1000)
torch.manual_seed(f * 10 + L)
m = model.from_config("victim", "ncf", factor_num=f, num_layers=L).I(dataset=ds).to(dev)
ts = [t.detach().cpu().numpy().copy() for t in m._tensors()]
P = orc.NCFParams(f, L, ts[0], ts[1], ts[2], ts[3], ts[4:4 + L], ts[4 + L:4 + 2 * L], ts[-2], ts[-1])
e = ds.generate_epoch(); cols = [e[k][:8000] for k in ("users", "items", "labels")]
un, it, lb = (c.cpu().numpy() for c in cols)
for s in range(8):
    sl = slice(s * 1000, (s + 1) * 1000)
    # gradients of this step on the CURRENT (shared) parameters
    m._run_epoch(cols[0][sl], cols[1][sl], cols[2][sl], 1000, apply_update=False)
    gg = [g.cpu().numpy().copy() for g in m._ws["grad"]]
    for g in m._ws["grad"]: g.zero_()
    loss_ref, gr = orc.ncf_step(P, un[sl], it[sl], lb[sl], apply_update=False)
    ge = [G.relerr(a, b.reshape(a.shape)) for a, b in zip(gg, gr)]
    # now update both
    lg = float(m._run_epoch(cols[0][sl], cols[1][sl], cols[2][sl], 1000).sum())
    lr_, _ = orc.ncf_step(P, un[sl], it[sl], lb[sl])
    te = [G.relerr(a.detach().cpu().numpy(), b.reshape(a.shape)) for a, b in zip(m._tensors(), P.tensors())]
    print("step", s, "loss rel %.1e" % (abs(lg - lr_) / abs(lr_)), "grad relerr max %.1e" % max(ge), "table relerr", " ".join("%.0e" % x for x in te))
