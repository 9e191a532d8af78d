"""INTEGRATION.md option A, executed as far as a machine without a GPU allows: the build's victims registered in the
REAL reference's factory (imported from /root/reference, present in the build container only), constructed through the
reference's own `recad.model.from_config` / lazy `.I(dataset=...)` with the reference's own ImplicitData object, and
handed to the reference's `Normal` workflow constructor.  Training needs the HIP device and must refuse loudly here."""
import os
import shutil
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "recad")), reason="the reference checkout is not on this machine")


@pytest.fixture(scope="module")
def recad(tmp_path_factory):
    scratch = tmp_path_factory.mktemp("recad_scratch")
    os.makedirs(scratch / "data", exist_ok=True)
    shutil.copytree(os.path.join(REF, "data", "dev"), scratch / "data" / "dev")
    cwd = os.getcwd()
    os.chdir(scratch)                      # the reference resolves ./data and ./generated from the cwd
    sys.path.insert(0, REF)
    try:
        import recad as ref
        ref.utils.TQDM = False
        yield ref
    finally:
        os.chdir(cwd)
        sys.path.remove(REF)


def test_victims_register_in_the_reference_factory(recad):
    import recad_amd
    from recad_amd import _lib
    from recad_amd.victim import MF, NCF, LightGCN
    saved = dict(recad.model.factories["victim"])
    try:
        recad.model.factories["victim"].update({"lightgcn": LightGCN, "mf": MF, "ncf": NCF})
        ds = recad.dataset.from_config("implicit", "dev", need_graph=True, download=False)
        lazy = recad.model.from_config("victim", "lightgcn", latent_dim_rec=32)       # recad/model/__init__.py:3-21
        assert isinstance(lazy, LightGCN) and lazy.model_name == "lightgcn"
        with pytest.raises(recad_amd.utils.NotInstantiatedError):
            lazy.train_step()
        m = lazy.I(dataset=ds)                                                          # the reference's dataset object
        info = ds.info_describe()
        assert tuple(m.embedding_user.weight.shape) == (info["n_users"], 32)
        assert m.Graph.shape == (info["n_users"] + info["n_items"],) * 2 and m.Graph.is_sparse      # implicit.py:320-326
        assert set(m.input_describe()["forward"]) == {"users", "items"}                # normal.py:122-131
        assert len(m.output_describe()["train_step"]) == 1
        assert isinstance(m.reset(), LightGCN)
        # the reference's own workflow object accepts it (normal.py:25-40); running it needs the device
        attacker = recad.model.from_config("attacker", "random")
        ds_attack = recad.dataset.from_config("explicit", "dev", download=False)
        wf = recad.workflow.from_config("no defense", victim_data=ds, attack_data=ds_attack, victim=recad.model.from_config("victim", "lightgcn"),
                                        attacker=attacker, rec_epoch=1, attack_epoch=0, device=torch.device("cpu"))
        assert isinstance(wf.victim, LightGCN)
        with pytest.raises(_lib.HipCallError):        # no CPU fallback: the path fails loudly without its HIP device
            wf.execute()
        for name, cls in (("mf", MF), ("ncf", NCF)):
            v = recad.model.from_config("victim", name).I(dataset=ds)
            assert isinstance(v, cls) and set(v.input_describe()["train_step"]) == {"users", "items", "labels"}
    finally:
        recad.model.factories["victim"].clear()
        recad.model.factories["victim"].update(saved)
