"""Registries and config records with the reference's names and fields
(graph_hscn/config/config.py:13-152, defaults.py:1-39).

pydantic is not used: under the installed pydantic 2.x the reference's
``@root_validator`` raises at class creation, and wandb
is optional here (the reference makes it mandatory, config.py:146-152).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional

from torch.optim import Adagrad, Adam, AdamW

from ..nn.conv import GATConv, GCNConv
from ..nn.functional import Activation

# defaults.py:1-39
BATCH_SIZE = 32
NUM_WORKERS = 0
NUM_LAYERS = 3
HIDDEN_CHANNELS = 16
BATCH_ACCUMULATION = 1
CLIP_GRAD_NORM = False
LR = 0.01
WEIGHT_DECAY = 5e-4
EPOCHS = 500
EVAL_PERIOD = 10
MIN_DELTA = 0.01
PATIENCE = 2
NUM_CLUSTERS = 4
CLUSTER_EPOCHS = 10

ACT_DICT: dict[str, Callable] = {  # config.py:13-18
    "elu": Activation("elu"),
    "relu": Activation("relu"),
    "tanh": Activation("tanh"),
    "identity": Activation("identity"),
}
# config.py:19-23 also lists "gin": GINConv(dim, hidden, add_self_loops=...) raises a
# TypeError in PyG (GINConv takes an nn, not channel counts), so only gcn/gat can be
# built through build_conv_relation (model/hscn.py:117-125).
CONV_DICT: dict[str, type] = {"gcn": GCNConv, "gat": GATConv}
OPTIM_DICT: dict[str, type] = {"adagrad": Adagrad, "adam": Adam, "adamW": AdamW}  # config.py:24-28
DATASETS_NUM_FEATURES: dict[str, int] = {"peptides_func": 9, "peptides_struct": 9}


@dataclass
class DataConfig:  # config.py:32-46
    dataset_name: str
    pe: bool = False
    batch_size: int = BATCH_SIZE
    num_workers: int = NUM_WORKERS
    task_level: Optional[str] = None

    def __post_init__(self):
        self.task_level = "graph" if "peptides" in self.dataset_name else (self.task_level or "graph")


DROPOUT = 0.2  # defaults.py:6
USE_BATCH_NORM = False
USE_LAYER_NORM = False


@dataclass
class MPNNConfig:  # config.py:49-73
    conv_type: str
    activation: str
    hidden_channels: int = HIDDEN_CHANNELS
    num_layers: int = NUM_LAYERS
    dropout: float = DROPOUT
    use_batch_norm: bool = USE_BATCH_NORM
    use_layer_norm: bool = USE_LAYER_NORM

    def __post_init__(self):
        if self.dropout and not (0.0 <= self.dropout <= 1.0):
            raise ValueError(f"{self.dropout} must be between 0.0 and 1.0.")
        for v in (self.num_layers, self.hidden_channels):
            if v < 0:
                raise ValueError(f"{v} must be non-negative.")


@dataclass
class HSCNConfig:  # config.py:76-93 (+ mp_units, read at main.py:102 but absent there)
    activation: str
    lv_conv_type: str = "GAT"
    ll_conv_type: str = "GCN"
    vv_conv_type: str = "GCN"
    hidden_channels: int = HIDDEN_CHANNELS
    num_layers: int = NUM_LAYERS
    num_clusters: int = NUM_CLUSTERS
    cluster_epochs: int = CLUSTER_EPOCHS
    mp_units: list = field(default_factory=lambda: [16])

    def __post_init__(self):
        for v in (self.num_layers, self.hidden_channels):
            if v < 0:
                raise ValueError(f"{v} must be non-negative.")


@dataclass
class OptimConfig:  # config.py:96-112
    optim_type: str
    batch_accumulation: int = BATCH_ACCUMULATION
    clip_grad_norm: bool = CLIP_GRAD_NORM
    lr: float = LR
    weight_decay: float = WEIGHT_DECAY

    def __post_init__(self):
        for v in (self.lr, self.weight_decay):
            if v and not (0.0 <= v <= 1.0):
                raise ValueError(f"{v} must be between 0.0 and 1.0.")


@dataclass
class PEConfig:  # config.py:115-130 (defaults.py:19-28)
    dim_in: int
    dim_emb: int
    dim_pe: int
    model: str = "DeepSet"
    layers: int = 1
    post_layers: int = 1
    eigen_max_freqs: int = 10
    eigvec_norm: str = "L2"
    eigen_laplacian_norm: str = "sym"
    phi_hidden_dim: int = 32
    phi_out_dim: int = 4
    pass_as_var: bool = False
    use_bn: bool = False


@dataclass
class TrainingConfig:  # config.py:133-152
    model_type: str
    loss_fn: str
    metric: str
    epochs: int = EPOCHS
    eval_period: int = EVAL_PERIOD
    min_delta: float = MIN_DELTA
    patience: int = PATIENCE
    use_wandb: bool = False
    wandb_proj_name: Optional[str] = None
