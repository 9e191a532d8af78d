"""Implicit-feedback dataset feeding the victim hot path.

This is the build's own thin counterpart of recad/dataset/implicit.py: it
exposes the same object contract the victims and the workflow consume --
``info_describe()`` (n_users, n_items, train/valid/test dicts, graph), ``generate_batch()``
(dicts of int64 device tensors, last batch short), ``inject_data``, ``reset``,
``partial_sample`` -- but keeps interactions as CSR arrays, samples with vectorised numpy
instead of per-draw Python loops, and builds the normalised adjacency on the GPU
(rk_build_norm_adj) instead of scipy dok/lil.
"""
import random
from copy import copy

import numpy as np
import torch

from .default import DATASET_IMPLICIT
from .utils import VarDim, get_logger


def _dict_to_csr(d, n_rows=None):
    users = np.fromiter((int(u) for u in d.keys()), dtype=np.int64, count=len(d))
    n = int(users.max()) + 1 if len(users) else 0
    n_rows = max(n, n_rows or 0)
    cnt = np.zeros(n_rows, dtype=np.int64)
    for u, items in d.items():
        cnt[int(u)] = len(items)
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(cnt)
    idx = np.zeros(ptr[-1], dtype=np.int32)
    for u, items in d.items():
        if len(items):
            idx[ptr[int(u)]:ptr[int(u) + 1]] = np.asarray(items, dtype=np.int32)
    return ptr, idx


def _csr_to_dict(ptr, idx):
    return {int(u): idx[ptr[u]:ptr[u + 1]].tolist() for u in range(len(ptr) - 1) if ptr[u + 1] > ptr[u]}


def _pad_csr(csr, n_rows):
    ptr, idx = csr
    if len(ptr) - 1 >= n_rows:
        return np.asarray(ptr, dtype=np.int64), np.asarray(idx, dtype=np.int32)
    out = np.full(n_rows + 1, ptr[-1], dtype=np.int64)
    out[: len(ptr)] = ptr
    return out, np.asarray(idx, dtype=np.int32)


def _sorted_unique_rows(ptr, idx):
    """Sort item ids within each user and drop duplicates (the scipy CSR the reference builds
    at implicit.py:206-209 sums duplicates; interactions are unique per user anyway)."""
    n = len(ptr) - 1
    users = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
    width = int(idx.max()) + 1 if len(idx) else 1
    keys = users * width + idx.astype(np.int64)
    if len(keys) > 1 and not bool(np.all(keys[1:] > keys[:-1])):   # (lists that are already sorted and unique skip the sort: 0.6 vs 10.6 ms at ml1m size)
        keys = np.unique(keys)
    cnt = np.bincount(keys // width, minlength=n)
    out = np.zeros(n + 1, dtype=np.int64)
    out[1:] = np.cumsum(cnt)
    return out, (keys % width).astype(np.int32)


def csv_to_dict(path, rating_filter=4):
    """recad/dataset/implicit.py:94-104: keep rating >= filter, time order, first occurrence."""
    import pandas as pd

    df = pd.read_csv(path).sort_values("timestamp")
    df = df[df["rating"] >= rating_filter]
    out = {}
    for u, i in zip(df["user_id"].to_numpy(), df["item_id"].to_numpy()):
        lst = out.setdefault(int(u), [])
        if int(i) not in lst:
            lst.append(int(i))
    return out


class ImplicitData:
    def __init__(self, **config):
        self.config = config
        self.logger = get_logger(f"{__name__}:{self.dataset_name}", level=config["logging_level"])
        self._mode = "train"
        self._dicts = {}
        splits = {}
        for split in ("train", "valid", "test"):
            if config.get(f"{split}_csr") is not None:
                splits[split] = tuple(np.asarray(a) for a in config[f"{split}_csr"])
            elif config.get(f"{split}_dict") is not None:
                self._dicts[split] = config[f"{split}_dict"]
                splits[split] = _dict_to_csr(config[f"{split}_dict"])
            elif config.get(f"path_{split}"):
                self._dicts[split] = csv_to_dict(config[f"path_{split}"], config["rating_filter"])
                splits[split] = _dict_to_csr(self._dicts[split])
            else:
                splits[split] = (np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32))
        # n_users / n_items = max id + 1 over train, valid, test (implicit.py:229-230,194-195)
        self.n_users = max(len(p) - 1 for p, _ in splits.values())
        # users with empty lists at the tail do not count in the reference (only non-empty rows do)
        self.n_users = max((int(np.nonzero(np.diff(p))[0].max()) + 1 if len(i) else 0) for p, i in splits.values())
        self.n_items = max((int(i.max()) + 1 if len(i) else 0) for _, i in splits.values())
        self._csr = {k: _pad_csr(v, self.n_users) for k, v in splits.items()}
        self._csr = {k: (p[: self.n_users + 1], i[: p[self.n_users]]) for k, (p, i) in self._csr.items()}
        self.traindataSize = int(self._csr["train"][0][-1])
        self.validDataSize = int(self._csr["valid"][0][-1])
        self.testDataSize = int(self._csr["test"][0][-1])
        # The reference builds UserItemNet (hence the graph AND the BPR positives) from whichever
        # split read_data() saw last, i.e. the TEST edges (implicit.py:173-192,206-209,233-235).
        src = "test" if config["graph_source"] == "reference" else "train"
        self._net = _sorted_unique_rows(*self._csr[src])
        self._net_keys = (np.repeat(np.arange(self.n_users, dtype=np.int64), np.diff(self._net[0])) * self.n_items
                          + self._net[1].astype(np.int64))
        self._train_sorted = None
        self._graph = None
        self._graph_coo = None
        seed = config.get("seed", None)
        self._rng = np.random.default_rng(np.random.randint(0, 2 ** 31 - 1) if seed is None else seed)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_config(cls, name, **user_config):
        config = {k: copy(v) for k, v in DATASET_IMPLICIT.items()}
        for k in ("train_csr", "valid_csr", "test_csr", "seed"):
            config[k] = None
        for k, v in user_config.items():
            if k == "download":
                continue
            if k not in config:
                get_logger(__name__).debug(f"Unexpected key [{k}] for {cls}")
            config[k] = v
        inst = object.__new__(cls)
        inst._dataset_name = name
        inst._init_config = config
        inst.__init__(**config)
        return inst

    @property
    def dataset_name(self):
        return getattr(self, "_dataset_name", type(self).__name__)

    def reset(self, **kwargs):
        config = copy(self._init_config)
        for k, v in kwargs.items():
            if k not in config:
                raise ValueError(f"reset arg {k} should be in {list(config)}")
            config[k] = v
            if k.endswith("_dict") and v is not None:
                config[k.replace("_dict", "_csr")] = None
            if k.endswith("_csr") and v is not None:
                config[k.replace("_csr", "_dict")] = None
        # splits not overridden carry over as CSR (cheap, no dict rebuild)
        for split in ("train", "valid", "test"):
            if config.get(f"{split}_csr") is None and config.get(f"{split}_dict") is None:
                config[f"{split}_csr"] = self._csr[split]
        return type(self).from_config(self.dataset_name, **config)

    # ------------------------------------------------------------------ description
    def _dict(self, split):
        if split not in self._dicts:
            self._dicts[split] = _csr_to_dict(*self._csr[split])
        return self._dicts[split]

    @property
    def train_dict(self):
        return self._dict("train")

    @property
    def valid_dict(self):
        return self._dict("valid")

    @property
    def test_dict(self):
        return self._dict("test")

    def train_csr_sorted(self):
        """user -> sorted unique train items (seen-item lists for evaluation)."""
        if self._train_sorted is None:
            self._train_sorted = _sorted_unique_rows(*self._csr["train"])
        return self._train_sorted

    @property
    def allPos(self):
        ptr, idx = self._net
        return [idx[ptr[u]:ptr[u + 1]] for u in range(self.n_users)]

    def graph_csr(self):
        """Normalised adjacency as a device CSR (built once, on the GPU)."""
        if self._graph is None:
            from .graph import CsrGraph
            self._graph = CsrGraph.from_user_item_csr(self.n_users, self.n_items, self._net[0], self._net[1],
                                                      self.config["device"])
        return self._graph

    def getSparseGraph(self):
        """The graph in the reference's format (coalesced torch sparse COO, implicit.py:243-298)."""
        if self._graph_coo is None:
            self._graph_coo = self.graph_csr().to_torch_coo()
        return self._graph_coo

    def batch_describe(self):
        if self._mode == "train" and self.config["sample"] == "pairwise":
            b = VarDim(max=self.config["pairwise_batch_size"], comment="batch")
            return {"users": (torch.int64, b), "positive_items": (torch.int64, b), "negative_items": (torch.int64, b)}
        if self._mode == "train" and self.config["sample"] == "pointwise":
            b = VarDim(max=self.config["pointwise_batch_size"], comment="batch")
            return {"users": (torch.int64, b), "items": (torch.int64, b), "labels": (torch.int64, b)}
        b = VarDim(max=self.config["test_batch_size"], comment="batch")
        return {"users": (torch.int64, b), "positive_items": (list, b), "ground_truth": (list, b)}

    class _Info(dict):
        """info_describe() result; the dict/graph views are materialised on first access."""

        def __init__(self, ds, base):
            super().__init__(base)
            self._ds = ds

        def __missing__(self, key):
            ds = self._ds
            lazy = {"train_dict": lambda: ds.train_dict, "valid_dict": lambda: ds.valid_dict,
                    "test_dict": lambda: ds.test_dict}
            if ds.config["need_graph"]:
                lazy["graph"] = ds.getSparseGraph
                lazy["graph_csr"] = ds.graph_csr
            if key not in lazy:
                raise KeyError(key)
            self[key] = lazy[key]()
            return self[key]

        def get(self, key, default=None):
            try:
                return self[key]
            except KeyError:
                return default

    def info_describe(self):
        return ImplicitData._Info(self, {
            "n_users": self.n_users, "n_items": self.n_items, "train_interactions": self.traindataSize,
            "valid_interactions": self.validDataSize, "test_interactions": self.testDataSize,
            "batch_describe": self.batch_describe(),
        })

    def mode(self):
        return self._mode

    def switch_mode(self, mode):
        assert mode in ["train", "test", "validate"]
        self._mode = mode

    # ------------------------------------------------------------------ samplers
    def pairwise_sample(self):
        """BPR triplets with the semantics of recad/dataset/implicit.py:50-74: traindataSize
        uniform user draws (with replacement), users without positives skipped, uniform
        positive, negative rejection-sampled outside the user's positives."""
        rng = self._rng
        ptr, idx = self._net
        deg = np.diff(ptr)
        users = rng.integers(0, self.n_users, self.traindataSize)
        users = users[deg[users] > 0]
        pos = idx[ptr[users] + (rng.random(len(users)) * deg[users]).astype(np.int64)]
        neg = rng.integers(0, self.n_items, len(users))
        todo = np.arange(len(users))
        while len(todo):
            keys = users[todo] * self.n_items + neg[todo]
            p = np.searchsorted(self._net_keys, keys)
            p = np.minimum(p, len(self._net_keys) - 1)
            clash = self._net_keys[p] == keys
            todo = todo[clash]
            neg[todo] = rng.integers(0, self.n_items, len(todo))
        return users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)

    def pointwise_sample(self):
        """(user, item, label) rows as recad/dataset/implicit.py:77-91: every train positive
        plus negative_ratio*deg negatives per user drawn WITH replacement from the items the
        user has not interacted with in train."""
        rng = self._rng
        ptr, idx = self.train_csr_sorted()
        deg = np.diff(ptr)
        ratio = self.config["negative_ratio"]
        pu = np.repeat(np.arange(self.n_users, dtype=np.int64), deg)
        pi = idx.astype(np.int64)
        nu = np.repeat(np.arange(self.n_users, dtype=np.int64), deg * ratio)
        # k-th free item of user u: draw a rank in the complement and skip over the sorted positives
        free = (self.n_items - deg)[nu]
        r = (rng.random(len(nu)) * free).astype(np.int64)
        # r-th free item of user u, exactly: idx[k] - k (k = position inside the user's sorted list) is the number
        # of free items below positive k and is non-decreasing, so j = #{k : idx[k] - k <= r} positives precede
        # the answer and the item is r + j (one vectorised binary search over per-user keys).
        k_in_row = np.arange(len(idx), dtype=np.int64) - np.repeat(ptr[:-1], deg)
        free_below = pu * self.n_items + (pi - k_in_row)          # sorted: user-major, non-decreasing inside a user
        j = np.searchsorted(free_below, nu * self.n_items + r, side="right") - ptr[nu]
        ni = r + j
        users = np.concatenate([pu, nu])
        items = np.concatenate([pi, ni])
        labels = np.concatenate([np.ones(len(pu), dtype=np.int64), np.zeros(len(nu), dtype=np.int64)])
        return users, items, labels

    def _device_epoch(self):
        """Samplers on the GPU (rk_bpr_sample / rk_pointwise_sample); shuffling by torch.randperm."""
        from . import _lib
        dev = torch.device(self.config["device"])
        seed = int(self._rng.integers(0, 2 ** 62))
        pairwise = self.config["sample"] == "pairwise"
        key = "_dev_net" if pairwise else "_dev_train"
        if not hasattr(self, key):
            ptr, idx = self._net if pairwise else self.train_csr_sorted()
            setattr(self, key, (torch.as_tensor(ptr, dtype=torch.int32).to(dev), torch.as_tensor(idx, dtype=torch.int32).to(dev)))
        ptr, idx = getattr(self, key)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed & 0x7FFFFFFF)
        if pairwise:
            n = self.traindataSize
            u, p, ng = (torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3))
            valid = torch.empty(n, dtype=torch.int32, device=dev)
            _lib.check(_lib.lib().rk_bpr_sample(self.n_users, self.n_items, _lib.ptr(ptr), _lib.ptr(idx), n, seed, _lib.ptr(u),
                                                _lib.ptr(p), _lib.ptr(ng), _lib.ptr(valid), _lib.stream_ptr()), "rk_bpr_sample")
            keep = valid.bool()
            cols = [t[keep] for t in (u, p, ng)]
            names, bs = ("users", "positive_items", "negative_items"), self.config["pairwise_batch_size"]
        else:
            ratio = self.config["negative_ratio"]
            E = idx.numel()
            n = E * (ratio + 1)
            cols = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3)]
            _lib.check(_lib.lib().rk_pointwise_sample(self.n_users, self.n_items, _lib.ptr(ptr), _lib.ptr(idx), E, ratio, seed,
                                                      _lib.ptr(cols[0]), _lib.ptr(cols[1]), _lib.ptr(cols[2]), _lib.stream_ptr()),
                       "rk_pointwise_sample")
            names, bs = ("users", "items", "labels"), self.config["pointwise_batch_size"]
        perm = torch.randperm(cols[0].numel(), device=dev, generator=gen)
        out = {k: c[perm].contiguous() for k, c in zip(names, cols)}
        out["batch_size"] = bs
        return out

    def generate_epoch(self):
        """One shuffled epoch as three int64 device tensors (+ batch size): the fast path the
        HIP victims consume; generate_batch() slices the same tensors."""
        dev = self.config["device"]
        smp = self.config["sampler"]
        if smp == "auto":
            smp = "device" if torch.device(dev).type == "cuda" else "numpy"
        if smp == "device" and self.config["sample"] in ("pairwise", "pointwise"):
            return self._device_epoch()
        if self.config["sample"] == "pairwise":
            cols, names, bs = self.pairwise_sample(), ("users", "positive_items", "negative_items"), self.config["pairwise_batch_size"]
        elif self.config["sample"] == "pointwise":
            cols, names, bs = self.pointwise_sample(), ("users", "items", "labels"), self.config["pointwise_batch_size"]
        else:
            raise NotImplementedError("Not implemented yet")
        perm = self._rng.permutation(len(cols[0]))
        out = {n: torch.from_numpy(c[perm]).to(dev) for n, c in zip(names, cols)}
        out["batch_size"] = bs
        return out

    def generate_batch(self, **config):
        if self._mode != "train":
            test = self.valid_dict if self._mode == "validate" else self.test_dict
            users = list(test.keys())
            bs = self.config["test_batch_size"]
            ap = self.allPos
            for i in range(0, len(users), bs):
                bu = users[i:i + bs]
                yield {"users": torch.tensor(bu, dtype=torch.int64, device=self.config["device"]),
                       "positive_items": [ap[u] for u in bu], "ground_truth": [test[u] for u in bu]}
            return
        ep = self.generate_epoch()
        bs = ep.pop("batch_size")
        n = len(next(iter(ep.values())))
        for i in range(0, n, bs):
            yield {k: v[i:i + bs] for k, v in ep.items()}

    # ------------------------------------------------------------------ perturbation
    def inject_data(self, data_mode, data, **kwargs):
        """recad/dataset/implicit.py:482-494 + fake_array2dict (:107-114): fake users are appended
        at ids n_users.., keeping the items rated STRICTLY above filter_num."""
        if data_mode != "explicit":
            raise NotImplementedError(f"Injection not supported in {data_mode} mode")
        assert len(data.shape) == 2, "Expect a user-item 2D rating matrix"
        uids, iids = np.where(np.asarray(data) > kwargs["filter_num"])
        ptr, idx = self._csr["train"]
        cnt = np.bincount(uids, minlength=data.shape[0])
        last = int(np.nonzero(cnt)[0].max()) + 1 if cnt.any() else 0
        new_ptr = np.concatenate([ptr, ptr[-1] + np.cumsum(cnt[:last])])
        order = np.argsort(uids, kind="stable")
        new_idx = np.concatenate([idx, iids[order].astype(np.int32)])
        return self.reset(train_csr=(new_ptr, new_idx), if_cache=False)

    def delete_data(self, data_mode, user_id, data, **kwargs):
        """recad/dataset/implicit.py:496-513: inject, then drop the flagged users."""
        ds = self.inject_data(data_mode, data, **kwargs)
        ptr, idx = ds._csr["train"]
        cnt = np.diff(ptr).copy()
        users = np.repeat(np.arange(len(cnt)), cnt)
        drop = np.isin(users, np.asarray(list(user_id)))
        cnt[np.asarray([u for u in user_id if u < len(cnt)], dtype=np.int64)] = 0
        new_ptr = np.zeros(len(cnt) + 1, dtype=np.int64)
        new_ptr[1:] = np.cumsum(cnt)
        return self.reset(train_csr=(new_ptr, idx[~drop]), if_cache=False)

    def partial_sample(self, **kwargs):
        assert "user_ratio" in kwargs, "Expect to have [user_ratio]"
        ratio = kwargs["user_ratio"]
        if abs(ratio - 1) < 1e-9:
            return self
        ptr, idx = self._csr["train"]
        users = [u for u in range(self.n_users) if ptr[u + 1] > ptr[u]]
        random.shuffle(users)
        keep = np.zeros(self.n_users, dtype=bool)
        keep[users[: int(len(users) * ratio)]] = True
        cnt = np.diff(ptr) * keep
        mask = np.repeat(keep, np.diff(ptr))
        new_ptr = np.zeros(self.n_users + 1, dtype=np.int64)
        new_ptr[1:] = np.cumsum(cnt)
        return self.reset(train_csr=(new_ptr, idx[mask]))

    def print_help(self, **kwargs):
        from pprint import pprint

        pprint({**self.info_describe(), "dataset_name": self.dataset_name})


factories = {"implicit": ImplicitData}


def from_config(scope, *args, **kwargs):
    """dataset.from_config("implicit", name, **kw) (recad/dataset/__init__.py:13-17)."""
    return factories[scope].from_config(*args, **kwargs)
