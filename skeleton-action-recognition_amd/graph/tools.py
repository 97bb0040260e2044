"""Adjacency helpers with the names and semantics of the reference's graph/tools.py:4-30, vectorised (float64).

edge2mat(link, n)[j, i] = 1 for every link (i, j); normalize_digraph scales every column by the reciprocal of its sum
(an all-zero column stays zero); get_spatial_graph stacks (identity, inward, outward).
"""
import numpy as np


def edge2mat(link, num_node):
    mat = np.zeros((num_node, num_node))
    pairs = np.asarray(list(link), dtype=np.int64).reshape(-1, 2)
    mat[pairs[:, 1], pairs[:, 0]] = 1
    return mat


def normalize_digraph(A):
    colsum = A.sum(axis=0)
    scale = np.zeros_like(colsum)
    np.divide(1.0, colsum, out=scale, where=colsum > 0)
    return A @ np.diag(scale)          # the reference's matrix product, so that the rounding is identical


def get_spatial_graph(num_node, self_link, inward, outward, normalize=True):
    post = normalize_digraph if normalize else (lambda m: m)
    return np.stack([edge2mat(self_link, num_node), post(edge2mat(inward, num_node)), post(edge2mat(outward, num_node))])
