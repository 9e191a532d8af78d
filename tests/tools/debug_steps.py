import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import oracle as orc
from tests import _golden as G
from tests._stub import LGN_KEYS, ReplayDataset
from recad_amd import model
dev = torch.device('cuda:0')
for name in sys.argv[1:]:
    g = G.load(name)
    ds = ReplayDataset(g, LGN_KEYS, device=dev)
    m = model.from_config("victim", "lightgcn", latent_dim_rec=int(g["dim"]), lightGCN_n_layers=int(g["layers"])).I(dataset=ds)
    u, i = G.lightgcn_init(g)
    m.embedding_user.weight.data.copy_(torch.from_numpy(u)); m.embedding_item.weight.data.copy_(torch.from_numpy(i))
    m = m.to(dev); m.graph_steps = 0
    csr = orc.coo_to_csr(int(g["n_users"]) + int(g["n_items"]), g["graph_row"], g["graph_col"], g["graph_val"])
    st = orc.AdamState(u.shape, i.shape)
    B = int(g["batch_len"][0])
    steps = [s for s in range(len(g["batch_len"])) if int(g["batch_len"][s]) == B][:6]
    users, pos, neg = (torch.cat([torch.from_numpy(g["batches"][s, k, :B].astype(np.int64)) for s in steps]).to(dev) for k in range(3))
    # run step by step with separate calls, and also all-at-once on a clone
    import copy
    for mode in ("one_call",):
        part = m._run_epoch(users, pos, neg, B)
        losses = part.sum(1).double().cpu().numpy()
    ol = []
    for s in steps:
        ol.append(orc.lightgcn_step(csr, u, i, st, *(g["batches"][s, k, :B] for k in range(3)), int(g["layers"])))
    print(name, "gpu losses", losses, "\n   oracle   ", np.array(ol), "\n   golden   ", g["losses"][steps])
    print("   table relerr", G.relerr(m.embedding_user.weight.detach().cpu().numpy(), u), G.relerr(m.embedding_item.weight.detach().cpu().numpy(), i))
    print("   gprop/gego residual", float(m._ws["gprop"].abs().max()), float(m._ws["gego"].abs().max()))
