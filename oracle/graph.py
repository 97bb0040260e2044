"""Oracle: NTU-RGB+D skeleton graph (restates graph/tools.py:4-30 and
graph/ntu_rgb_d.py:6-40 of the reference).  numpy float64, integer indexing.

Parity status: PINNED.  tests/test_oracle_graph.py compares the float32 bytes
against tests/golden/adjacency_f32.npy, produced by importing the reference's
own ``graph`` package (tests/golden/make_golden_graph.py), sha256
52b0058b1d83ca6aa24487064fb786136ba05f3047644f6e10e54799457827a5.
"""
import numpy as np

NUM_NODE = 25
# graph/ntu_rgb_d.py:8-11 -- 1-based (child, parent) pairs of the Kinect v2 skeleton
INWARD_ORI_INDEX = [(1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6),
                    (8, 7), (9, 21), (10, 9), (11, 10), (12, 11), (13, 1),
                    (14, 13), (15, 14), (16, 15), (17, 1), (18, 17), (19, 18),
                    (20, 19), (22, 23), (23, 8), (24, 25), (25, 12)]


def edge2mat(link, num_node):
    """graph/tools.py:4-8: A[j, i] = 1 for every link (i, j)."""
    A = np.zeros((num_node, num_node))
    for i, j in link:
        A[j, i] = 1
    return A


def normalize_digraph(A):
    """graph/tools.py:11-19: A @ diag(1/colsum); all-zero columns stay zero."""
    Dl = np.sum(A, 0)
    _, w = A.shape
    Dn = np.zeros((w, w))
    for i in range(w):
        if Dl[i] > 0:
            Dn[i, i] = Dl[i] ** (-1)
    return np.dot(A, Dn)


def spatial_adjacency(normalize=True):
    """graph/tools.py:22-30 + graph/ntu_rgb_d.py:6-14,31-32 -> (3,25,25) float64."""
    self_link = [(i, i) for i in range(NUM_NODE)]
    inward = [(i - 1, j - 1) for (i, j) in INWARD_ORI_INDEX]
    outward = [(j, i) for (i, j) in inward]
    I = edge2mat(self_link, NUM_NODE)
    In = edge2mat(inward, NUM_NODE)
    Out = edge2mat(outward, NUM_NODE)
    if normalize:
        In = normalize_digraph(In)
        Out = normalize_digraph(Out)
    return np.stack((I, In, Out))


def gin_adjacency():
    """graph/ntu_rgb_d.py:33-39 ('GIN' mode): un-normalised, identity slice dropped."""
    return spatial_adjacency(normalize=False)[1:]
