"""Pins tests/tools/aten_ref.py -- the k-thread ATen restatement bench.py times as `cpu_baseline` -- against
the goldens captured from the reference itself: the same recorded minibatches replayed through it must reproduce the
reference's per-step losses (1e-5) and trained tables (1e-4).  CPU only."""
import numpy as np
import pytest
import torch

from tests import _golden as G
from tests.tools import aten_ref

LOSS_RTOL, TABLE_RTOL = 1e-5, 1e-4


@pytest.mark.parametrize("name", ["lightgcn_dev_d64", "lightgcn_game_d64_tg"])
def test_aten_ref_replays_the_reference_goldens(name):
    g = G.load(name)
    U, I = int(g["n_users"]), int(g["n_items"])
    graph = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g["graph_row"], g["graph_col"]]).astype(np.int64)),
                                    torch.from_numpy(g["graph_val"].astype(np.float32)), (U + I, U + I)).coalesce()
    u0, i0 = G.lightgcn_init(g)
    prev = torch.get_num_threads()
    torch.set_num_threads(1)          # the goldens were taken single-threaded (the only bit-deterministic mode)
    try:
        m = aten_ref.AtenLightGCN(graph, u0, i0, int(g["layers"]))
        rs = int(g["row_stride"])
        lu, li = m.computer()
        assert G.relerr(torch.cat([lu, li]).detach().numpy()[::rs], g["light0"]) < 1e-6
        for s in range(len(g["batch_len"])):
            n = int(g["batch_len"][s])
            loss = m.step(*(g["batches"][s, k, :n] for k in range(3)))
            assert abs(loss - g["losses"][s]) <= LOSS_RTOL * abs(g["losses"][s]), (s, loss, g["losses"][s])
            if s == 0:
                assert G.relerr(m.eu.detach().numpy()[::rs], g["after1_user"]) < 1e-5
        assert G.relerr(m.eu.detach().numpy()[::rs], g["final_user"]) < TABLE_RTOL
        assert G.relerr(m.ei.detach().numpy()[::rs], g["final_item"]) < TABLE_RTOL
    finally:
        torch.set_num_threads(prev)


def test_norm_adj_coo_matches_the_golden_graph():
    """aten_ref.norm_adj_coo (what the timed baseline propagates through) against the reference's own graph tensor."""
    g = G.load("lightgcn_game_d64_tg")
    U, I = int(g["n_users"]), int(g["n_items"])
    coo = aten_ref.norm_adj_coo(U, I, g["train_ptr"], g["train_idx"])
    idx = coo.indices().numpy()
    assert np.array_equal(idx[0], g["graph_row"]) and np.array_equal(idx[1], g["graph_col"])
    assert np.allclose(coo.values().numpy(), g["graph_val"], rtol=1e-6, atol=0)
