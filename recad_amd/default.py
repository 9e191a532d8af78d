"""Default configuration registry for the victim path.

Mirrors the keys and values of the reference's registry for the three victims
(recad/default.py:103-132) and the implicit dataset / workflow knobs the hot path reads
(recad/default.py:49-62,247-267).  Only what the path needs is present.
"""
import logging
import os

import torch

SEED = 2023  # recad/default.py:21
DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")

MODEL = {
    "victim": {
        "lightgcn": {
            "latent_dim_rec": 128, "lightGCN_n_layers": 3, "A_split": False, "pretrain": False, "keep_prob": 0.6,
            "dropout": 0.0, "lambda": 1e-4, "optim": "adam", "lr": 1e-3,
        },
        "mf": {"factor_num": 3, "embedding_size": 128, "dropout": 0, "optim": "adam", "lr": 1e-3},
        "ncf": {
            "factor_num": 32, "num_layers": 5, "dropout": 0, "model": "NeuMF-end", "GMF_model": None, "MLP_model": None,
            "optim": "adam", "lr": 1e-3,
        },
    },
    "attacker": {"random": {"attack_num": 50, "filler_num": 36}},
}
for _scope in MODEL.values():
    for _cfg in _scope.values():
        _cfg["logging_level"] = logging.INFO
        _cfg["device"] = DEVICE

DATASET_IMPLICIT = {
    "path_train": None, "path_valid": None, "path_test": None,
    "test_batch_size": 400, "A_split": False, "A_n_fold": 100,
    "pairwise_batch_size": 1024, "pointwise_batch_size": 1024, "sample": "pairwise", "negative_ratio": 4,
    "need_graph": True, "rating_filter": 4,
    "logging_level": logging.INFO, "train_dict": None, "valid_dict": None, "test_dict": None,
    "device": DEVICE, "if_cache": False, "cache_dir": os.path.join(".", "generated"),
    # build-specific (no reference counterpart):
    "graph_source": "reference",  # "reference" = adjacency/positives from the LAST split read (test); "train"
    "sampler": "auto",            # "numpy" (vectorised host) | "device" (HIP) | "auto" = device when on a GPU
}

WORKFLOW = {
    "no defense": {
        "rec_epoch": 400, "attack_epoch": 100, "target_id_list": [0], "filter_num": 4, "topks": [10, 20, 50, 100],
        "logging_level": logging.INFO, "device": DEVICE, "cache_dir": os.path.join(".", "workflows_results"),
    },
    "defense": {
        "rec_epoch": 400, "attack_epoch": 100, "target_id_list": [0], "filter_num": 4, "topks": [10, 20, 50, 100],
        "defense_epoch": 1,
        "logging_level": logging.INFO, "device": DEVICE, "cache_dir": os.path.join(".", "workflows_results"),
    },
}


def set_device_id(cuda_id):
    """recad/default.py:285-290."""
    device = torch.device(f"cuda:{cuda_id}" if torch.cuda.is_available() else "cpu")
    for scope in MODEL.values():
        for cfg in scope.values():
            cfg["device"] = device
    DATASET_IMPLICIT["device"] = device
    for cfg in WORKFLOW.values():
        cfg["device"] = device
    return device
