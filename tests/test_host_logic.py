"""CPU tests: the host-side mirror of the reference interface (lazy init, defaults, dataset
contract, workflow plumbing) and the C-ABI surface.  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import recad_amd
from recad_amd import _lib, dataset, model, synth, workflow
from recad_amd.utils import InstantiateFail, NotInstantiatedError
from tests import _golden as G
from tests._stub import LGN_KEYS, PW_KEYS, ReplayDataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "recad_hip.h")).read()
    declared = set(re.findall(r"\b(rk_[a-z0-9_]+)\s*\(", hdr))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/recad_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.lib().rk_abi_version() == _lib.ABI_VERSION == 9


def test_no_cpu_fallback():
    g = G.load("lightgcn_dev_d64")
    ds = ReplayDataset(g, LGN_KEYS)
    m = model.from_config("victim", "lightgcn", latent_dim_rec=64).I(dataset=ds)
    with pytest.raises(_lib.HipCallError):
        m.train_step()  # parameters on the CPU: must fail loudly, never fall back
    with pytest.raises(_lib.HipCallError):
        m(torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "recad_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"
    # helper scripts neither; bench.py only inside the oracle replay its parity / cpu_baseline_port leg runs (the checker,
    # timed -- never the thing measured), __graft_entry__.py only inside smoke()
    for f in os.listdir(os.path.join(ROOT, "scripts")):
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "scripts", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"scripts/{f} imports the oracle"
    for f, fn in (("bench.py", "oracle_replay"), ("__graft_entry__.py", "smoke")):
        src = open(os.path.join(ROOT, f)).read()
        hits = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle", src, re.M)]
        assert len(hits) == 1, (f, hits)
        head = src[: hits[0]]
        assert head.rfind("def " + fn + "(") == max(head.rfind("\ndef "), head.rfind("\n    def ")) + 1 or \
            head[head.rfind("\ndef "):].startswith("\ndef " + fn + "("), f"{f}: the oracle import must live inside {fn}()"


def test_product_reads_no_environment_variable():
    """paths and shapes are arguments.  No Python file of the package touches os.environ / os.getenv, and the C side's
    only getenv sits behind -DRK_TUNING (csrc/host/layout.h: compiled out of the shipped library).  A variant build is bound
    through the explicit _lib.load(path) (scripts/_tune.py), never through the environment."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "recad_amd")):
        for f in files:
            p = os.path.join(dirpath, f)
            if f.endswith(".py"):
                code = "\n".join(l.split("#", 1)[0] for l in open(p).read().split("\n"))
                assert not re.search(r"\bos\.environ\b|\bgetenv\b|\benviron\b", code), f"{p} reads the environment"
            elif f.endswith((".hip", ".h")):
                src = open(p).read()
                for m in re.finditer(r"\bgetenv\s*\(", src):
                    head = src[: m.start()]
                    assert head.rfind("#ifdef RK_TUNING") > head.rfind("#endif") or head.rfind("#if defined(RK_TUNING)") > head.rfind("#endif"), \
                        f"{p}: getenv outside an RK_TUNING block"
    assert _lib.LIB_PATH == os.path.join(ROOT, "recad_amd", "lib", "librecad_hip.so")
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "from recad_amd import _lib; print(_lib.LIB_PATH)"], cwd=ROOT, capture_output=True, text=True,
                         env=dict(os.environ, RECAD_HIP_LIB="/nonexistent/x.so", RECAD_TUNING_LIB="/nonexistent/y.so"))
    assert out.stdout.strip() == _lib.LIB_PATH, out
    with pytest.raises(_lib.HipCallError):
        _lib.lib()
        _lib.load("librecad_hip_tuning.so")   # too late: the product library is bound


# ------------------------------------------------------------------ lazy-init contract
def test_lazy_init_contract():
    lazy = model.from_config("victim", "lightgcn", latent_dim_rec=32, not_a_key=1)
    assert lazy.model_name == "lightgcn"
    assert lazy._init_config["latent_dim_rec"] == 32 and "not_a_key" not in lazy._init_config
    assert lazy._init_config["lightGCN_n_layers"] == 3 and lazy._init_config["lambda"] == 1e-4
    for call in (lambda: lazy.train_step(), lambda: lazy.input_describe(), lambda: lazy(torch.zeros(1), torch.zeros(1))):
        with pytest.raises(NotInstantiatedError):
            call()
    g = G.load("lightgcn_dev_d64")
    ds = ReplayDataset(g, LGN_KEYS)
    real = lazy.I(dataset=ds)
    assert real is not lazy and real.I() is real
    assert real.model_name == "lightgcn" and real._init_config == lazy._init_config
    assert tuple(real.embedding_user.weight.shape) == (512, 32)
    assert set(real.input_describe()["forward"]) == {"users", "items"}
    assert len(real.output_describe()["train_step"]) == 1
    again = real.reset()
    with pytest.raises(NotInstantiatedError):
        again.train_step()
    assert again._init_config == lazy._init_config
    with pytest.raises(ValueError):
        real.reset(bogus=1)
    with pytest.raises(InstantiateFail):
        model.from_config("victim", "lightgcn", A_split=True).I(dataset=ds)
    with pytest.raises(InstantiateFail):
        model.from_config("victim", "mf").I(dataset=object())
    assert isinstance(real, torch.nn.Module) and isinstance(real.optimizer, torch.optim.Adam)


def test_defaults_match_reference_registry():
    d = recad_amd.default.MODEL["victim"]
    assert d["lightgcn"]["latent_dim_rec"] == 128 and d["lightgcn"]["lr"] == 1e-3 and d["lightgcn"]["keep_prob"] == 0.6
    assert d["mf"]["factor_num"] == 3 and d["mf"]["embedding_size"] == 128
    assert recad_amd.default.WORKFLOW["no defense"]["topks"] == [10, 20, 50, 100]
    assert recad_amd.default.WORKFLOW["no defense"]["filter_num"] == 4


def test_init_rng_order_matches_reference():
    """Same torch RNG consumption as lightgcn.py:40-48 / mf.py:16-24 => same initial tables."""
    g = G.load("lightgcn_dev_d64")
    torch.manual_seed(2023)
    m = model.from_config("victim", "lightgcn", latent_dim_rec=64).I(dataset=ReplayDataset(g, LGN_KEYS))
    assert np.array_equal(m.embedding_user.weight.detach().numpy(), g["ref_init_user"])
    assert np.array_equal(m.embedding_item.weight.detach().numpy(), g["ref_init_item"])
    g = G.load("mf_dev_e64")
    torch.manual_seed(2023)
    m = model.from_config("victim", "mf", embedding_size=64).I(dataset=ReplayDataset(g, PW_KEYS, with_graph=False))
    for nm, p in (("user_emb", m.user_emb), ("item_emb", m.item_emb), ("user_bias", m.user_bias), ("item_bias", m.item_bias)):
        assert np.array_equal(p.weight.detach().numpy(), g["ref_init_" + nm]), nm
    assert float(m.mean.item()) == float(g["mean"]) and not m.mean.requires_grad


# ------------------------------------------------------------------ dataset contract
@pytest.fixture(scope="module")
def tiny():
    d = synth.make("tiny")
    return d, dataset.from_config("implicit", "tiny", train_csr=d["train"], valid_csr=d["valid"], test_csr=d["test"],
                                  need_graph=False, device=torch.device("cpu"), seed=5)


def test_dataset_sizes_and_quirk(tiny):
    d, ds = tiny
    info = ds.info_describe()
    assert info["n_users"] == 300 and info["n_items"] == 200
    assert info["train_interactions"] == len(d["train"][1])
    assert sum(len(v) for v in info["train_dict"].values()) == len(d["train"][1])
    # reference quirk: positives / adjacency come from the TEST edges by default
    assert ds._net[0][-1] == len(np.unique(np.repeat(np.arange(300), np.diff(d["test"][0])) * 200 + d["test"][1]))
    ds_t = ds.reset(graph_source="train")
    assert ds_t._net[0][-1] == len(d["train"][1])


def test_pairwise_sampler_semantics(tiny):
    _, ds = tiny
    ds = ds.reset(graph_source="train")
    ep = ds.generate_epoch()
    u, p, n = (ep[k].numpy() for k in LGN_KEYS)
    keys = set(ds._net_keys.tolist())
    assert len(u) <= ds.traindataSize and len(u) > 0.9 * ds.traindataSize
    assert all((a * ds.n_items + b) in keys for a, b in zip(u, p))
    assert not any((a * ds.n_items + b) in keys for a, b in zip(u, n))
    # uniform over users (with replacement): chi-square-ish sanity
    cnt = np.bincount(u, minlength=ds.n_users)
    assert abs(cnt.mean() - len(u) / ds.n_users) < 1e-9 and cnt.std() < 3 * np.sqrt(cnt.mean()) + 1
    batches = list(ds.generate_batch())
    assert all(len(b["users"]) == 1024 for b in batches[:-1]) and 0 < len(batches[-1]["users"]) <= 1024
    assert batches[0]["users"].dtype == torch.int64 and set(batches[0]) == set(LGN_KEYS)


def test_pointwise_sampler_semantics(tiny):
    _, ds = tiny
    ds = ds.reset(sample="pointwise")
    ep = ds.generate_epoch()
    u, i, l = (ep[k].numpy() for k in PW_KEYS)
    tp, ti = ds.train_csr_sorted()
    tk = set((np.repeat(np.arange(ds.n_users), np.diff(tp)) * ds.n_items + ti).tolist())
    assert len(u) == 5 * ds.traindataSize and l.sum() == ds.traindataSize  # 1 pos + negative_ratio=4 negs
    assert all(((a * ds.n_items + b) in tk) == (c == 1) for a, b, c in zip(u, i, l))
    # negatives are uniform over the complement: every free item of a busy user gets drawn
    uu = int(np.argmax(np.diff(tp)))
    negs = i[(u == uu) & (l == 0)]
    assert negs.min() >= 0 and negs.max() < ds.n_items and len(np.unique(negs)) > 0.5 * min(len(negs), ds.n_items - np.diff(tp)[uu])


def test_inject_matches_reference_golden():
    """inject_data / fake_array2dict semantics against what the reference itself produced."""
    g = G.load("inject_dev")
    U, I = int(g["n_users"]), int(g["n_items"])
    base = G.load("lightgcn_dev_d64")
    ds = dataset.from_config("implicit", "dev", train_csr=(base["train_ptr"].astype(np.int64), base["train_idx"]),
                             test_csr=(base["test_ptr"].astype(np.int64), base["test_idx"]), need_graph=False,
                             device=torch.device("cpu"))
    assert (ds.n_users, ds.n_items) == (U, I)
    ds2 = ds.inject_data("explicit", g["fake"], filter_num=4)
    assert ds2.n_users == int(g["n_users_after"]) and ds2.n_items == int(g["n_items_after"])
    assert ds2.traindataSize == int(g["traindata_size_after"])
    ptr, idx = ds2._csr["train"]
    assert np.array_equal(ptr.astype(np.int32), g["train_ptr_after"]) and np.array_equal(idx, g["train_idx_after"])
    assert ds.n_users == U, "the original dataset must be untouched"


def test_csv_loading(tmp_path):
    p = tmp_path / "x_train.csv"
    p.write_text("user_id,item_id,rating,timestamp\n0,5,5,30\n0,2,4,10\n0,2,5,20\n1,7,3,5\n2,1,4,1\n")
    ds = dataset.from_config("implicit", "x", path_train=str(p), need_graph=False, device=torch.device("cpu"), graph_source="train")
    assert ds.train_dict == {0: [2, 5], 2: [1]} and ds.n_users == 3 and ds.n_items == 6


# ------------------------------------------------------------------ workflow plumbing with a stub victim
class _StubVictim(torch.nn.Module):
    """Duck-typed victim (oracle-free, numpy scores) to exercise the driver on CPU."""
    model_name = "stub"

    def __init__(self, dataset=None):
        super().__init__()
        self.dataset = dataset
        self.steps = 0

    def I(self, dataset=None, **kw):
        return _StubVictim(dataset)

    def reset(self, **kw):
        return _StubVictim()

    def train_step(self, **config):
        assert "progress_bar" in config and "target_id_list" in config
        self.steps += sum(1 for _ in self.dataset.generate_batch())
        return (0.5,)

    def forward(self, users, items):
        return ((users * 7 + items * 13) % 101).float() + (items == 0).float() * (50.0 if self.dataset.n_users > 300 else 0.0)

    def input_describe(self):
        return {"forward": {"users": None, "items": None}}

    def output_describe(self):
        return {"train_step": {"loss": None}}


def test_workflow_plumbing_cpu(tiny):
    _, ds = tiny
    wf = workflow.from_config("no defense", victim_data=ds, attack_data=None, victim=_StubVictim(),
                              attacker=workflow.RandomAttack(ds.n_items, attack_num=10, filler_num=5, seed=1),
                              rec_epoch=2, attack_epoch=0, device=torch.device("cpu"))
    res = wf.execute()
    assert wf.fake_dataset.n_users == ds.n_users + 10 and wf.fake_dataset.traindataSize > ds.traindataSize
    assert set(res) >= {"pred_shift", "HR@10", "HR@10 after attack", "HR@100", "n_eval_users"}
    assert res["pred_shift"] == pytest.approx(50.0) and res["HR@10 after attack"] >= res["HR@10"]
    with pytest.raises(TypeError):
        workflow.from_config("no defense", victim=_StubVictim())


def test_eligible_users_and_rows():
    from recad_amd.evaluate import eligible_users, hr_rows
    ptr = np.array([0, 2, 2, 5, 6]); idx = np.array([0, 3, 1, 2, 4, 0])
    assert eligible_users(ptr, idx, [0]).tolist() == [2]
    assert eligible_users(ptr, idx, [9]).tolist() == [0, 2, 3]
    res = {"target_score": np.array([[1.5], [2.5]], dtype=np.float32), "target_rank": np.array([[9], [10]], dtype=np.int32)}
    rows = hr_rows([4, 7], res, [10, 20])
    assert rows.tolist() == [[4.0, 1.5, 1.0, 1.0], [7.0, 2.5, 0.0, 1.0]]


class _StubDefender:
    """Flags the injected users (ids >= the clean user count) plus one genuine user."""
    model_name = "stub-defender"

    def __init__(self, n_clean):
        self.n_clean = n_clean

    def I(self, **kw):
        return self

    def to(self, device):
        return self

    def input_describe(self):
        return {}

    def defense_step(self, **kw):
        return list(range(self.n_clean, self.n_clean + 10)) + [3]


def test_defense_workflow_plumbing_cpu(tiny):
    _, ds = tiny
    wf = workflow.from_config("defense", victim_data=ds, attack_data=None, victim=_StubVictim(),
                              attacker=workflow.RandomAttack(ds.n_items, attack_num=10, filler_num=5, seed=1),
                              defender=_StubDefender(ds.n_users), rec_epoch=1, attack_epoch=0, device=torch.device("cpu"))
    res = wf.execute()
    assert res["n_flagged"] == 11 and set(res) == {"attacked", "defended", "n_flagged"}
    # every fake user and the flagged genuine user are gone from the cleaned train set
    ptr, _ = wf.cleaned_dataset._csr["train"]
    deg = np.diff(ptr)
    assert deg[3] == 0 and (len(deg) <= ds.n_users or deg[ds.n_users:].sum() == 0)
    assert wf.cleaned_dataset.traindataSize == ds.traindataSize - np.diff(ds._csr["train"][0])[3]
    assert res["attacked"]["pred_shift"] == pytest.approx(50.0) and res["defended"]["pred_shift"] == pytest.approx(0.0)
    with pytest.raises(TypeError):
        workflow.from_config("defense", victim_data=ds, attack_data=None, victim=_StubVictim(), attacker=None)


def test_ctypes_descriptors_match_the_header(tmp_path):
    """The ctypes mirrors in recad_amd/_lib.py must have the C header's size and field offsets: a shorter or
    shifted structure would hand the library garbage for the trailing fields."""
    import ctypes as C
    import shutil
    import subprocess
    from recad_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    probes = {
        "rk_lightgcn_desc": ["n_users", "lambda", "rowptr", "n_blocks", "user_emb", "grad", "state", "coef", "spmm_scratch",
                             "row_bits", "keep_prob", "drop_seed", "tpos", "row_blocks", "row_blocks_extra", "lds_sync"],
        "rk_ncf_desc": ["n_users", "lr", "ug", "pw", "grad", "m", "v", "acts", "d0", "max_batch", "gemm_scratch",
                        "gemm_scratch_floats", "wgrad_part"],
        "rk_spmm_epilogue": [],
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "recad_hip.h"', 'int main(void) {']
    for st, fields in probes.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fields:
            lines.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines += ['return 0; }']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirrors = {"rk_lightgcn_desc": _lib.LightGCNDesc, "rk_ncf_desc": _lib.NCFDesc, "rk_spmm_epilogue": _lib.SpmmEpilogue}
    rename = {"lambda": "lam"}
    for st, fields in probes.items():
        cls = mirrors[st]
        assert C.sizeof(cls) == int(out[st]), (st, C.sizeof(cls), out[st])
        for f in fields:
            assert getattr(cls, rename.get(f, f)).offset == int(out[f"{st}.{f}"]), (st, f)


def test_score_topk_plan_names_the_path():
    """rk_score_topk_plan (host code of the C ABI, no GPU needed) names the path rk_score_topk will take and sizes its scratch:
    GEMM + selection (a [nb, n_items] matrix) below 8 192 items or for user blocks too small to fill the chip, the
    register-resident panel form (a k-permuted copy of the item table) otherwise; a request forces a path and the panel form's
    shape knobs, and a path that cannot take the request is refused.  No environment variable is consulted."""
    import os
    from recad_amd import _lib
    from recad_amd.evaluate import score_plan
    for k in ("RK_SEL_OFF", "RK_SEL_FORCE", "RK_PAN_OFF", "RK_PAN_FORCE"):   # (the round-3 switches: must be inert now)
        os.environ[k] = "1"
    try:
        def f(nb, I, d, K=100, T=1, req=None):
            p = score_plan(nb, I, d, K, T, req)
            assert (p.nb, p.n_items, p.dim, p.K, p.n_targets) == (nb, I, d, K, T)
            return p.path, int(p.scratch_floats)
        panel = lambda I, d: (_lib.RK_SCORE_PANEL, I * 16 * (2 if d <= 32 else 4 if d <= 64 else 8 if d <= 128 else 16) + 4)
        gemm = lambda nb, I: (_lib.RK_SCORE_GEMM, nb * ((I + 31) // 32 * 32))   # rows padded to 128-byte lines (plan.ld_scores)
        assert f(5893, 3702, 64) == gemm(5893, 3702)                  # ml1m: GEMM + selection
        assert f(16384, 34474, 64) == panel(34474, 64)
        assert f(8192, 8192, 64) == panel(8192, 64)                   # a full machine of workgroups: from 8 192 items on (219 vs 241 us measured)
        assert f(8192, 6144, 64) == gemm(8192, 6144)                  # (181 vs 161 us)
        assert f(4096, 8192, 64) == gemm(4096, 8192)
        assert f(8192, 34474, 256, 100, 4) == gemm(8192, 34474)       # several targets: the panel form only at dim <= 64 ...
        assert f(16384, 34474, 64, 100, 4) == panel(34474, 64)
        assert score_plan(16384, 131072, 64, 100, 4, None).panel_rows == 16   # ... and in 16-row workgroups
        assert f(54617, 34474, 128) == panel(34474, 128)
        assert f(8192, 34474, 100) == panel(34474, 100)               # k padded to a multiple of 16
        assert f(8192, 34474, 256) == panel(34474, 256)
        assert f(4096, 34474, 64) == panel(34474, 64)                 # 4096 users fill the chip at dim <= 64 ...
        assert f(4096, 34474, 128) == gemm(4096, 34474)               # ... not beyond
        assert f(2048, 131072, 64) == gemm(2048, 131072)              # few users: the GEMM parallelises over the items too
        assert f(2048, 500000, 64) == gemm(2048, 500000)              # (round 3 sent this corner to the fused sweep: 3x slower)
        assert f(16384, 500000, 64) == panel(500000, 64)
        assert f(8192, 500000, 64, 100, 5) == gemm(8192, 500000)      # five targets: not the panel form
        assert f(16384, 34474, 64, req={"path": "gemm"}) == gemm(16384, 34474)
        assert f(5893, 3702, 64, req={"path": "panel"}) == panel(3702, 64)
        p = score_plan(16384, 131072, 64, 100, 1, None)
        assert (p.panel_rows, p.panel_ntw, p.panel_safe) == (32, 15, 0)                   # operand-bound sweep: 32-row workgroups
        p = score_plan(16384, 34474, 64, 100, 1, {"path": "panel", "panel_rows": 32, "panel_safe": True})
        assert (p.panel_rows, p.panel_ntw, p.panel_safe) == (32, 15, 1)
        p = score_plan(100, 900, 64, 100, 1, {"path": "panel"})
        assert (p.panel_rows, p.panel_ntw) == (16, 8)                                      # small catalogue: narrow panels
        with pytest.raises(_lib.HipCallError):
            score_plan(8192, 500000, 64, 100, 5, {"path": "panel"})                       # five targets
        with pytest.raises(_lib.HipCallError):
            score_plan(8192, 34474, 64, 300, 1, None)                                      # K > 256
    finally:
        for k in ("RK_SEL_OFF", "RK_SEL_FORCE", "RK_PAN_OFF", "RK_PAN_FORCE"):
            os.environ.pop(k, None)
