"""
ghn3_amd -- MI355X-native implementation of the GHN-3 parameter-prediction hot path.

Mirrors the package surface of the reference for this path (/root/reference/ghn3/__init__.py:8-13):
``Graph, GraphBatch, from_pretrained, GHN3, ConvDecoder3, SequentialMultipleInOut, norm_check, get_metadata`` plus
the DDP helpers, the trainer and the target networks (``Network``, ``NetworkLight``: ghn3/ops.py:584-585).
"""

from .graph import Graph, GraphBatch
from .nn import from_pretrained, GHN3, ConvDecoder3, SequentialMultipleInOut, norm_check, get_metadata
from .utils import log, Logger, print_grads
from .ddp_utils import (setup_ddp, is_ddp, get_ddp_rank, clean_ddp, avg_ddp_metric, all_reduce_flat_grads,
                        sync_parameters)
from .optim import FusedAdamW, save_checkpoint
from .trainer import Trainer
from .ops import Network, NetworkLight
from .deepnets1m import DeepNets1MDDP, NetBatchSamplerDDP

__all__ = ['Graph', 'GraphBatch', 'from_pretrained', 'GHN3', 'ConvDecoder3', 'SequentialMultipleInOut', 'log', 'Logger', 'print_grads', 'norm_check', 'get_metadata',
           'setup_ddp', 'is_ddp', 'get_ddp_rank', 'clean_ddp', 'avg_ddp_metric', 'all_reduce_flat_grads', 'sync_parameters', 'FusedAdamW', 'save_checkpoint', 'Trainer',
           'Network', 'NetworkLight', 'DeepNets1MDDP', 'NetBatchSamplerDDP']
