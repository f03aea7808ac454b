"""Run configuration with the reference's defaults.

The reference parses sys.argv at import time into a module-global dict
(/root/reference/utility/word.py:7, utility/utils.py:18-62) and merges a per-model
dict (utility/config.py:1-81).  Here `get_config(model, **overrides)` builds the
same dict explicitly; `CFG` is the process-wide default the models read when no
config is passed, exactly as the reference's models read `utility.word.CFG`.
"""
import torch

_BASE = {
    "model": "lightgcn", "dataset": "synthetic",
    "train_batch": 512, "test_batch": 512, "has_val": False, "use_tag": True,
    "patient_epoch": 10, "test_interval": 5, "early_stop_key": "ndcg",
    "topks": [10, 20], "lr": 0.01, "reg": 0.0, "cor_reg": 0.0,
    "epochs": 1000, "dim_latent": 64, "dim_layer_list": [64, 32, 16],
    "message_drop_list": [0.0, 0.0, 0.0], "node_drop": 0.0,
    "seed": 2020, "cpu_core": 4, "split_adj_k": 1,
    "hip_graph": False,       # Basic_train: replay each phase's step as one captured HIP graph (train.GraphedStep)
    "all_gather": "collective",   # row-sharded models (dist.py): "direct" = one grouped send / receive pair per peer
}

# utility/config.py:1-12, 41-52
_PER_MODEL = {
    "ngcf": {"norm_type": "ngcf", "agg_type": "bi_agg", "mul_loss_func": "logsigmoid"},
    "lightgcn": {"mul_loss_func": "softplus", "norm_type": "bi_norm", "cor_batch": 100},
    "tgcn": {"dim_weight": 10, "dim_atten": 32, "num_bit_conv": 32, "num_vec_conv": 8, "margin": 1,
             "transtag_batch": 512, "neighbor_k": 25, "transtag_reg": 0.0001, "mul_loss_func": "logsigmoid"},
    # utility/config.py:14-30
    "dgcf": {"mul_loss_func": "softplus", "norm_type": "plain", "factor_k": 4, "iterate_k": 2, "cor_batch": 100},
    "disengcn": {"mul_loss_func": "softplus", "norm_type": "plain", "factor_k": 4, "iterate_k": 2, "cor_batch": 100},
    # utility/config.py:54-61 -- note the default agg_type "bi_agg" switches KGAT's propagation off (kgat.py:100)
    "kgat": {"dim_relation": 64, "transe_reg": 0.0001, "transe_batch": 1024, "agg_type": "bi_agg", "mul_loss_func": "softplus"},
}


def get_config(model="lightgcn", **overrides):
    if model not in _PER_MODEL:
        raise KeyError(f"model {model!r} is outside the hot-path scope (have {sorted(_PER_MODEL)})")
    cfg = dict(_BASE)
    cfg["model"] = model
    cfg.update(_PER_MODEL[model])
    cfg["device"] = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    cfg.update(overrides)
    return cfg


CFG = get_config("lightgcn")


def init_seed(seed):
    """utility/utils.py:10-15."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
