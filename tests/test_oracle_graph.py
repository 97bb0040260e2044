"""CPU: the graph oracle and the product's graph drop-in are bit-exact with the adjacency produced by
the reference's own graph package (tests/golden/make_golden_graph.py)."""
import hashlib
import os

import numpy as np

from oracle.graph import spatial_adjacency, gin_adjacency

SHA = "52b0058b1d83ca6aa24487064fb786136ba05f3047644f6e10e54799457827a5"


def test_oracle_adjacency_bit_exact(golden_dir):
    A = spatial_adjacency().astype(np.float32)
    G = np.load(os.path.join(golden_dir, "adjacency_f32.npy"))
    assert A.shape == (3, 25, 25)
    assert A.tobytes() == G.tobytes()
    assert hashlib.sha256(A.tobytes()).hexdigest() == SHA


def test_oracle_gin_adjacency_bit_exact(golden_dir):
    G = np.load(os.path.join(golden_dir, "adjacency_gin_f32.npy"))
    assert gin_adjacency().astype(np.float32).tobytes() == G.tobytes()


def test_product_graph_dropin_bit_exact(golden_dir):
    from graph.ntu_rgb_d import Graph, num_node, inward, outward, neighbor, self_link
    g = Graph()
    assert g.A.dtype == np.float64 and g.num_node == num_node == 25
    assert g.A.astype(np.float32).tobytes() == np.load(os.path.join(golden_dir, "adjacency_f32.npy")).tobytes()
    assert Graph("GIN").A.astype(np.float32).tobytes() == np.load(os.path.join(golden_dir, "adjacency_gin_f32.npy")).tobytes()
    assert len(inward) == 24 and outward == [(j, i) for i, j in inward] and neighbor == inward + outward
    assert self_link == [(i, i) for i in range(25)]
    assert g.get_adjacency_matrix() is g.A
    import pytest
    with pytest.raises(ValueError):
        Graph("nope")


def test_adjacency_structure():
    A = spatial_adjacency()
    nnz = [(A[k] != 0).sum() for k in range(3)]
    assert nnz == [25, 24, 24]
    assert set(np.unique(A[2][A[2] != 0])) == {1.0, 0.5, 0.25}
    # column normalisation: non-empty columns sum to 1
    for k in (1, 2):
        cs = A[k].sum(0)
        assert np.all((cs == 0) | (cs == 1))


def test_gather_lists_reproduce_dense_einsum():
    from sar_amd.graph_tables import gather_lists
    A = spatial_adjacency()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 25)).astype(np.float32)
    for transpose in (False, True):
        idx, wt, nz, colsum = gather_lists(A, transpose)
        assert nz == ([1, 4, 1] if transpose else [1, 1, 4])
        Ad = np.transpose(A, (0, 2, 1)) if transpose else A
        for k in range(3):
            dense = x @ Ad[k].astype(np.float32)
            g = np.zeros_like(dense)
            for j in range(4):
                g += x[:, idx[k, :, j]] * wt[k, :, j]
            assert np.array_equal(g, dense)   # weights are exact binary fractions -> bit-exact
            assert np.array_equal(colsum[k], Ad[k].sum(0).astype(np.float32))


def test_gather_lists_reject_dense():
    import pytest
    from sar_amd.graph_tables import gather_lists
    with pytest.raises(ValueError):
        gather_lists(np.ones((3, 25, 25)))
