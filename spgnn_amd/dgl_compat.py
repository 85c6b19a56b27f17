"""The slice of the ``dgl`` module namespace the reference touches, so
``import dgl`` / ``from dgl import DGLGraph`` in reference-style runner code can be pointed here."""
from .graph import DGLGraph, TreeGraph, batch, unbatch, remove_self_loop, to_networkx, graph_from_adj  # noqa: F401
from .nn import DGLError  # noqa: F401


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
