"""Wall time of a whole no-defense workflow (train -> random attack -> inject -> retrain -> evaluate)
on ml1m-shaped data, LightGCN d=64, rec_epoch epochs per training."""
import sys, time, torch
sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
import recad_amd
from recad_amd import dataset, model, synth, workflow
dev = torch.device('cuda:0')
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shape = sys.argv[2] if len(sys.argv) > 2 else "ml1m"
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 64
d = synth.make(shape)
for gs in ("train", "reference"):
    t0 = time.time()
    ds = dataset.from_config("implicit", shape, train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"], device=dev, graph_source=gs, seed=1)
    wf = workflow.from_config("no defense", victim_data=ds, attack_data=None, victim=model.from_config("victim", "lightgcn", latent_dim_rec=dim),
                              attacker=workflow.RandomAttack(ds.n_items, seed=2), rec_epoch=epochs, attack_epoch=0, device=dev)
    res = wf.execute(); torch.cuda.synchronize()
    print(f"{shape} d={dim} graph={gs} rec_epoch={epochs}: total {time.time()-t0:.2f} s", {k: round(v, 3) for k, v in wf.timings.items()}, {k: round(v, 4) for k, v in res.items()})
