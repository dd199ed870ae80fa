"""gnn_uds_amd -- MI355X (gfx950) message-passing engine for the GNN-UDS graph-convolution hot path.

(Importable spelling of the repo's `gnn-uds_amd` package.)  Scope: the drainage-network spatial
block of Zhiyu014/GNN-UDS's surrogate (`surrogate/emulator.py:215-235,264-288`) -- fusion MLPs,
learned-weight node<->link incidence aggregation, GAT neighbour softmax/aggregation -- as hand-written
HIP kernels behind the C ABI in include/uds_hip.h, wrapped in torch.nn.Modules that keep the
reference's layer signatures.  See DESIGN.md.
"""
from . import graph                                                   # noqa: F401
from .graph import CSR, DrainageGraph, synthetic_drainage_network     # noqa: F401
from .layers import (Dense, DiffusionConv, GATConv, GCNConv, MixedGAT, NodeEdge,     # noqa: F401
                     SpatialBlock, SpatialLayer)

from .emulator import GRU, LSTM, Conv1D, Emulator                     # noqa: F401,E402
from . import inp                                                     # noqa: F401,E402
from .agent import ConvNet, GlobalAttnSumPool                         # noqa: F401,E402
from . import mpc                                                     # noqa: F401,E402
from . import mbrl                                                    # noqa: F401,E402

__version__ = '0.1.0'
