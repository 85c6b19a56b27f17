"""Cached-embedding reader and batch assembler against the synthetic generator."""
import numpy as np
import pytest
import torch

from spgnn_amd import data, synthetic


def _write(tmp_path, n=5):
    samples = synthetic.synthetic_trees(n, rank=9, n_lo=21, n_hi=60)
    uids = [f"1.2.840.{i}" for i in range(n)]
    for uid, s in zip(uids, samples):
        data.write_embedding(str(tmp_path), uid, s)
    return samples, uids


def test_dataset_reads_reference_schema_and_collates(tmp_path):
    samples, uids = _write(tmp_path)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    assert len(ds) == 5
    item = ds[2]
    assert item["fvs"].dtype == np.float64 and item["adj"].dtype == np.uint8 and item["labels"].dtype == np.uint8
    assert np.array_equal(item["adj"], samples[2]["adj"]) and np.allclose(item["fvs"], samples[2]["fvs"])
    loader = torch.utils.data.DataLoader(ds, batch_size=3, collate_fn=data.collate_native, num_workers=0)
    batches = list(loader)
    assert [len(b["adj"]) for b in batches] == [3, 2] and batches[0]["meta"]["uid"] == uids[:3]
    with open(tmp_path / "derived" / "conv_embedding" / "bad.pkl", "wb") as fp:
        import pickle; pickle.dump({"fvs": 1}, fp)
    with pytest.raises(KeyError):
        data.ConvEmbeddingDataset(str(tmp_path), ["bad"])[0]


def test_assemble_batch_on_cpu_matches_generator(tmp_path):
    samples, uids = _write(tmp_path)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    g = data.assemble_batch(data.collate_native([ds[i] for i in range(5)]), device="cpu")
    ref = synthetic.batch_from_samples(samples, "cpu", 39)
    assert np.array_equal(g._src, ref._src) and np.array_equal(g._dst, ref._dst)
    assert g.batch_num_nodes_list == ref.batch_num_nodes_list and g.batch_num_edges_list == ref.batch_num_edges_list
    for k in ("fvs", "fvs_out", "y", "pos_enc"):
        assert torch.equal(g.ndata[k], ref.ndata[k]), k


@pytest.mark.gpu
def test_assemble_batch_on_gpu_matches_host_path(tmp_path):
    samples, uids = _write(tmp_path, n=12)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    g = data.assemble_batch([ds[i] for i in range(12)], device="cuda")
    ref = synthetic.batch_from_samples(samples, "cpu", 39)
    for k in ("fvs", "fvs_out", "y", "pos_enc"):
        assert torch.equal(g.ndata[k].cpu(), ref.ndata[k]), k
    csc, rcsc = g.csc(), ref.csc("cpu")
    for name in ("indptr", "indices", "eid", "out_indptr", "out_indices", "out_pos"):
        assert torch.equal(getattr(csc, name).cpu(), getattr(rcsc, name)), name
    assert g.ndata["pos_enc"].stride(0) % 4 == 0           # rows ready for the vector / MFMA kernels
