"""How close does the HIP NCF path get to the REFERENCE's golden at factor 256 / L 3 (BASELINE config 5): step-1 gradients,
tables after step 1 and after the golden's steps (adam_close's counts), evaluation rows.  Prints numbers; asserts nothing."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import model  # noqa: E402
from tests import _golden as G  # noqa: E402
from tests._stub import PW_KEYS, ReplayDataset  # noqa: E402

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["ncf_game_f256_l3"]:
    g = G.load(name)
    f, L = int(g["factor"]), int(g["layers"])
    rs, dstr = int(g["row_stride"]), int(g["dense_stride"])
    ds = ReplayDataset(g, PW_KEYS, device=dev, with_graph=False, steps=[0])
    m = model.from_config("victim", "ncf", factor_num=f, num_layers=L).I(dataset=ds)
    (ug, ig, um, im), W, b, pw, pb = G.ncf_init(g)
    for p, a in zip((m.embed_user_GMF, m.embed_item_GMF, m.embed_user_MLP, m.embed_item_MLP), (ug, ig, um, im)):
        p.weight.data.copy_(torch.from_numpy(a))
    for l, x in enumerate([x for x in m.MLP_layers if isinstance(x, torch.nn.Linear)]):
        x.weight.data.copy_(torch.from_numpy(W[l]))
    m.predict_layer.weight.data.copy_(torch.from_numpy(pw))
    m = m.to(dev)
    names = ["embed_user_GMF.weight", "embed_item_GMF.weight", "embed_user_MLP.weight", "embed_item_MLP.weight"]
    names += [f"MLP_layers.{3 * l + 1}.weight" for l in range(L)] + [f"MLP_layers.{3 * l + 1}.bias" for l in range(L)]
    names += ["predict_layer.weight", "predict_layer.bias"]
    pick = lambda n, a: (a.detach().cpu().numpy() if torch.is_tensor(a) else a)[::rs] if n.startswith("embed_") else (a.detach().cpu().numpy() if torch.is_tensor(a) else a).reshape(-1)[::dstr]
    n0 = int(g["batch_len"][0])
    b0 = next(ds.generate_batch())
    part = m._run_epoch(b0["users"], b0["items"], b0["labels"], n0, apply_update=False)
    print(name, "loss", float(part.sum()), g["losses"][0])
    for nme, gr in zip(names, m._ws["grad"]):
        got, ref = pick(nme, gr), g["grad1_" + nme]
        print("  grad1 %-26s err %.3e" % (nme, np.abs(got - ref.reshape(got.shape)).max() / np.abs(ref).max()))
    for gr in m._ws["grad"]:
        gr.zero_()
    params = dict(m.named_parameters())
    steps = len(g["batch_len"])
    for s in range(steps):
        ds.steps = [s]
        (loss,) = m.train_step()
        print("  step", s, "loss", loss, g["losses"][s], "rel", abs(loss - g["losses"][s]) / g["losses"][s])
        if s == 0:
            for nme in names:
                a_, b_ = pick(nme, params[nme]), g["after1_" + nme]
                d_ = np.abs(a_ - b_.reshape(a_.shape))
                print("    after1 %-26s relerr %.3e  frac>1e-4*max %.4f" % (nme, d_.max() / np.abs(b_).max(), float((d_ > 1e-4 * np.abs(b_).max()).mean())))
    for nme in names:
        ok, info = G.adam_close(pick(nme, params[nme]), g["final_" + nme], 1e-3, steps, outlier_frac=5e-3, travel_frac=0.5)
        print("  final %-26s adam_close %s n_bad %d of %d (%.4f) max diff %.3e" % (nme, ok, info[0], info[1], info[0] / info[1], info[2]))
