"""The slice of the ``dgl`` module namespace the reference touches, so
``import dgl`` / ``from dgl import DGLGraph`` in reference-style runner code can be pointed here."""
from .graph import DGLGraph, TreeGraph, batch, unbatch, remove_self_loop, to_networkx, graph_from_adj  # noqa: F401
from .graph import Block, to_block  # noqa: F401
from .nn import DGLError  # noqa: F401
from . import dataloading  # noqa: F401  (dgl.dataloading.MultiLayerNeighborSampler / NodeDataLoader, job_runner.py:1487-1497)
from .dataloading import in_subgraph, seed  # noqa: F401

NID, EID = "_ID", "_ID"


class sampling:  # dgl.sampling.sample_neighbors
    sample_neighbors = staticmethod(dataloading.sample_neighbors)


class random:  # dgl.random.seed
    seed = staticmethod(dataloading.seed)


def add_self_loop(g):
    import numpy as np
    out = TreeGraph((g._src, g._dst), g.number_of_nodes(), g.device)
    for k, v in g.ndata.items():
        out.ndata[k] = v
    out.add_edges(np.arange(g.number_of_nodes()), np.arange(g.number_of_nodes()))
    return out


class backend:  # dgl.backend.asnumpy (reference job_runner.py:1815)
    @staticmethod
    def asnumpy(t):
        return t.detach().cpu().numpy()
