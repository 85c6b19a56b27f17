"""spgnn_amd — MI355X-native (gfx950) graph-convolution hot path for airway-tree GNNs.

Drop-in for the DGL layers and graph calls DIAGNijmegen/spgnn uses:

    from spgnn_amd.nn import GATConv, GraphConv, SAGEConv, GINConv      # was: dgl.nn.pytorch
    from spgnn_amd import dgl_compat as dgl                               # DGLGraph, batch, unbatch, ...
    from spgnn_amd import models                                          # GCN/GAT/GIN/SAGE/GATPSPGNN*/…Net

The compute path is libspgnn_hip.so (hand-written HIP, C ABI in include/spgnn_hip.h); there is
no CPU fallback — ops raise if the library is missing or tensors are not on a ROCm device.
"""
from . import graph  # noqa: F401  (pure numpy/torch host code; safe without a GPU)

__version__ = "0.1.0"
