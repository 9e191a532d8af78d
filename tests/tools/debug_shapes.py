import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from recad_amd import dataset, model, synth
from tests import _golden as G
dev = torch.device('cuda:0')
d, L, gs_, graph_steps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
dd = synth.make("tiny")
ds = dataset.from_config("implicit", "tiny", train_csr=dd["train"], valid_csr=dd["valid"], test_csr=dd["test"], device=dev, graph_source=gs_, seed=d + L, pairwise_batch_size=512)
torch.manual_seed(d * 10 + L)
m = model.from_config("victim", "lightgcn", latent_dim_rec=d, lightGCN_n_layers=L).I(dataset=ds).to(dev)
m.graph_steps = graph_steps
u0 = m.embedding_user.weight.detach().cpu().numpy().copy(); i0 = m.embedding_item.weight.detach().cpu().numpy().copy()
g = ds.graph_csr(); csr = (g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy())
st = orc.AdamState(u0.shape, i0.shape)
for ep in range(3):
    e = ds.generate_epoch(); users, pos, neg = (e[k] for k in ("users", "positive_items", "negative_items"))
    un, pn, nn_ = users.cpu().numpy(), pos.cpu().numpy(), neg.cpu().numpy()
    n = len(un); errs = []
    if graph_steps == 0:
        for s in range((n + 511) // 512):
            sl = slice(s * 512, (s + 1) * 512)
            part = m._run_epoch(users[sl], pos[sl], neg[sl], 512)   # one step per call
            l = float(part.sum())
            ref = orc.lightgcn_step(csr, u0, i0, st, un[sl], pn[sl], nn_[sl], L)
            errs.append("%.1e/%.1e" % (abs(l - ref) / abs(ref), G.relerr(m.embedding_user.weight.detach().cpu().numpy(), u0)))
    else:
        losses = m._run_epoch(users, pos, neg, 512).sum(1).double().cpu().numpy()
        for s in range(len(losses)):
            sl = slice(s * 512, (s + 1) * 512)
            ref = orc.lightgcn_step(csr, u0, i0, st, un[sl], pn[sl], nn_[sl], L)
            errs.append("%.1e" % (abs(losses[s] - ref) / abs(ref)))
        errs.append("tab %.1e" % G.relerr(m.embedding_user.weight.detach().cpu().numpy(), u0))
    print("epoch", ep, " ".join(errs))
