"""Adjacency helpers with the reference's names and semantics (graph/tools.py:4-30), vectorised.

edge2mat(link, n)[j, i] = 1 for every link (i, j); normalize_digraph divides each column by its sum
(all-zero columns stay zero); get_spatial_graph stacks (I, In, Out).  float64 like the reference.
"""
import numpy as np


def edge2mat(link, num_node):
    A = np.zeros((num_node, num_node))
    if len(link):
        src, dst = np.asarray(link, dtype=np.int64).T
        A[dst, src] = 1
    return A


def normalize_digraph(A):
    col = A.sum(axis=0)
    inv = np.zeros_like(col)
    np.divide(1.0, col, out=inv, where=col > 0)   # Dl[i] ** (-1) where Dl[i] > 0
    return A @ np.diag(inv)


def get_spatial_graph(num_node, self_link, inward, outward, normalize=True):
    mats = [edge2mat(self_link, num_node), edge2mat(inward, num_node), edge2mat(outward, num_node)]
    if normalize:
        mats[1], mats[2] = normalize_digraph(mats[1]), normalize_digraph(mats[2])
    return np.stack(mats)
