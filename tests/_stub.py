"""Minimal dataset stand-ins that replay golden minibatches through the dataset contract
the victims consume (info_describe() + generate_batch(), recad/dataset/implicit.py:397-458)."""
import numpy as np
import torch


class ReplayDataset:
    def __init__(self, g, keys, device="cpu", with_graph=True, steps=None):
        self.g, self.keys, self.device = g, keys, device
        self.U, self.I = int(g["n_users"]), int(g["n_items"])
        self.steps = range(len(g["batch_len"])) if steps is None else steps
        self.graph = None
        if with_graph:
            idx = torch.from_numpy(np.stack([g["graph_row"], g["graph_col"]]).astype(np.int64))
            self.graph = torch.sparse_coo_tensor(idx, torch.from_numpy(g["graph_val"]), (self.U + self.I,) * 2).coalesce()

    def info_describe(self):
        d = {"n_users": self.U, "n_items": self.I}
        if self.graph is not None:
            d["graph"] = self.graph
        return d

    def generate_batch(self):
        g = self.g
        for s in self.steps:
            n = int(g["batch_len"][s])
            yield {k: torch.from_numpy(g["batches"][s, j, :n].astype(np.int64)).to(self.device) for j, k in enumerate(self.keys)}


LGN_KEYS = ("users", "positive_items", "negative_items")
PW_KEYS = ("users", "items", "labels")
