"""Experiment configurations of the GNN stage.

``load_settings(path)`` reads a reference-style settings module (every UPPERCASE name becomes an
attribute; reference utils.py:34-61) so the reference's ``exp_settings/*.py`` files drive this
package unchanged.  ``CONFIGS`` embeds the GNN-relevant values of those files (they are data, not
code; file:line cited per entry) for bench.py and the tests, which run on machines without the
reference checkout.
"""
from __future__ import annotations

import importlib.util
from types import SimpleNamespace
from typing import Dict

# CNN-trunk kwargs, identical in all 12 exp_settings/*.py (e.g. st_pgat_spgnn_3.py:88-99)
_TRUNK = dict(n_layers=3, in_ch_list=[1, 32, 64, 128], base_ch_list=[24, 32, 64, 128], end_ch_list=[32, 64, 128, 256],
              kernel_sizes=[3, 3, 3, 3], checkpoint_layers=[0, 1, 1, 0, 1, 1, 1],
              padding_list=[(1, 1, 1)] * 4, conv_strides=[[1, 2], [1, 2], [1, 2]], dropout=0.0, spatial_size=10,
              norm_method="bn", act_method="relu", out_ch=22, fv_dim=1024, node_embed_dim=1024)

# CLASS_WEIGHTS (st_pgat_spgnn_3.py:70-74), identical in every GNN config
CLASS_WEIGHTS = {0: 0.1, 1: 0.2, **{k: 0.8 for k in range(2, 23)}}
_OPT = dict(OPTIMIZER={"method": "torch.optim.SGD", "momentum": 0.9, "lr": 0.0001},
            SCHEDULER={"method": "torch.optim.lr_scheduler.ExponentialLR", "gamma": 0.9},
            GCN_STEPS=300, TRAIN_BATCH_SIZE=64, NR_CLASS=22, CLASS_WEIGHTS=CLASS_WEIGHTS, GRAPH_MODE="all_connected")


def _gat(layers, hiddens, **kw):
    return {"method": "models.GATNet", **_TRUNK, "num_gat_layers": layers, "num_heads": 2, "num_out_heads": 2,
            "feat_drop": 0.1, "attn_drop": 0.1, "num_hiddens": hiddens, "negative_slope": 0.2, **kw}


_SPGNN = {"method": "models.GATPositionSPGNNNet", **_TRUNK, "num_gat_layers": 3, "num_heads": 2, "num_out_heads": 2,
          "feat_drop": 0.1, "attn_drop": 0.1, "num_hiddens": [256, 128, 64], "num_pos_heads": 1,
          "pos_hiddens": [256, 128, 64], "negative_slope": 0.2, "pos_enc_dim": 39}

CONFIGS: Dict[str, dict] = {
    # exp_settings/st_gcn_3.py:80-101
    "st_gcn_3": dict(MODEL={"method": "models.GCNNet", **_TRUNK, "num_gcn_layers": 3, "num_hiddens": [256, 128, 64]},
                     SAMPLING_RATE=0.05, POS_ENC_DIM=None, KIND="gcn", CONV_LAYERS=4, **_OPT),
    # exp_settings/st_gat_1.py, st_gat_3.py:83-108, st_gat_6.py:81-106 (+ _nr: "res": False)
    "st_gat_1": dict(MODEL=_gat(1, [256]), SAMPLING_RATE=0.05, POS_ENC_DIM=None, KIND="gat", CONV_LAYERS=2, **_OPT),
    "st_gat_3": dict(MODEL=_gat(3, [256, 128, 64]), SAMPLING_RATE=0.3, POS_ENC_DIM=None, KIND="gat", CONV_LAYERS=4, **_OPT),
    "st_gat_6": dict(MODEL=_gat(6, [256, 128, 64, 64, 64, 64]), SAMPLING_RATE=0.15, POS_ENC_DIM=None, KIND="gat",
                     CONV_LAYERS=7, **_OPT),
    "st_gat_1_nr": dict(MODEL=_gat(1, [256], res=False), SAMPLING_RATE=0.05, POS_ENC_DIM=None, KIND="gat", CONV_LAYERS=2, **_OPT),
    "st_gat_3_nr": dict(MODEL=_gat(3, [256, 128, 64], res=False), SAMPLING_RATE=0.3, POS_ENC_DIM=None, KIND="gat",
                        CONV_LAYERS=4, **_OPT),
    "st_gat_6_nr": dict(MODEL=_gat(6, [256, 128, 64, 64, 64, 64], res=False), SAMPLING_RATE=0.15, POS_ENC_DIM=None,
                        KIND="gat", CONV_LAYERS=7, **_OPT),
    # exp_settings/st_gin_3.py:77-98
    "st_gin_3": dict(MODEL={"method": "models.GINNet", **_TRUNK, "num_gin_layers": 3, "num_hiddens": [256, 128, 64]},
                     SAMPLING_RATE=0.05, POS_ENC_DIM=None, KIND="gin", CONV_LAYERS=4, **_OPT),
    # exp_settings/st_sage_3.py:80-107
    "st_sage_3": dict(MODEL={"method": "models.SAGENet", **_TRUNK, "num_layers": 3, "node_ks": [2, 2, 2, 2],
                             "feat_drop": 0.1, "node_sample_rate": 0.3, "num_hiddens": [256, 128, 64]},
                      SAMPLING_RATE=0.05, POS_ENC_DIM=None, KIND="sage", CONV_LAYERS=4, **_OPT),
    # exp_settings/st_pgat_spgnn_3.py:27,86-115 ; st_pgat_spgnnnl_3.py (mode PENL)
    "st_pgat_spgnn_3": dict(MODEL=dict(_SPGNN), SAMPLING_RATE=0.15, POS_ENC_DIM=39, KIND="spgnn_pel", CONV_LAYERS=7, **_OPT),
    "st_pgat_spgnnnl_3": dict(MODEL={**_SPGNN, "mode": "PENL"}, SAMPLING_RATE=0.15, POS_ENC_DIM=39, KIND="spgnn_penl",
                              CONV_LAYERS=4, **_OPT),
}


def get_config(name: str) -> SimpleNamespace:
    import copy
    return SimpleNamespace(**copy.deepcopy(CONFIGS[name]), NAME=name)


def load_settings(path: str) -> SimpleNamespace:
    """Exec a settings module and expose its UPPERCASE names (reference utils.Settings, utils.py:34-61)."""
    spec = importlib.util.spec_from_file_location("settings", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ns = SimpleNamespace(**{k: getattr(mod, k) for k in dir(mod) if k.isupper()})
    ns.settings_module_path = path
    return ns


def class_weight_list(class_weights: dict):
    """``[w[k] for k in sorted(w)][1:]`` (reference job_runner.py:1372, 1867): weights of classes 0..21."""
    return [class_weights[k] for k in sorted(class_weights.keys())][1:]
