"""tagrec_amd -- MI355X-native hot path for tag-aware graph recommenders.

Python host code (this package) mirrors the reference's operator / model / step
surface and calls a C-ABI HIP library
(`csrc/` -> `libtagrec_hip.so`, declared in `include/tagrec.h`) through ctypes.
There is no CPU fallback: every op raises if the library or a GPU is missing.

Reference name                      here
  model.help.split_mm / mul_loss ...  tagrec_amd.help
  model.LightGCN / NGCF / TGCN        tagrec_amd.LightGCN / NGCF / TGCN
  model.DGCF / DisenGCN / KGAT        tagrec_amd.DGCF / DisenGCN / KGAT
  train_data.BPR_training_data        tagrec_amd.BPR_training_data
  training.Basic_train / Basic_test   tagrec_amd.Basic_train / Basic_test
  training.basic_train.epoch_training tagrec_amd.epoch_training
  utility.word.CFG                    tagrec_amd.CFG (get_config(model, **kw))
"""
from . import synth  # noqa: F401
from . import _lib, config, data, graph, help  # noqa: F401
from ._lib import TagrecError  # noqa: F401
from .config import CFG, get_config, init_seed  # noqa: F401
from .evaluate import Basic_test  # noqa: F401
from .graph import Graph, creat_adj  # noqa: F401
from .lightgcn import LightGCN  # noqa: F401
from .ngcf import NGCF  # noqa: F401
from .tgcn import TGCN  # noqa: F401
from .dgcf import DGCF  # noqa: F401
from .disengcn import DisenGCN  # noqa: F401
from .kgat import KGAT  # noqa: F401
from .train import Adam, Basic_train, Early_stop, GraphedStep, epoch_training  # noqa: F401
from .train_data import (Abstract_training_data, BPR_training_data, DGCF_training_data,  # noqa: F401
                         Fixed_training_data, KGAT_training_data, TransTag_training_data)
