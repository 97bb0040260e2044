"""NTU-RGB+D 25-joint skeleton graph; same public names as the reference graph/ntu_rgb_d.py:6-40.

The skeleton is stored as the 1-based parent of every Kinect-v2 joint (0 = root, joint 21 "spine
shoulder"); `inward` lists (child, parent) 0-based pairs in child order, `outward` the reversed pairs.
"""
import numpy as np

from graph import tools

num_node = 25
#            1   2   3  4   5  6  7  8   9 10  11  12 13  14  15  16 17  18  19  20 21  22 23  24  25
_PARENT = [2, 21, 21, 3, 21, 5, 6, 7, 21, 9, 10, 11, 1, 13, 14, 15, 1, 17, 18, 19, 0, 23, 8, 25, 12]
self_link = [(i, i) for i in range(num_node)]
inward_ori_index = [(c + 1, p) for c, p in enumerate(_PARENT) if p]
inward = [(i - 1, j - 1) for (i, j) in inward_ori_index]
outward = [(j, i) for (i, j) in inward]
neighbor = inward + outward


class Graph:
    def __init__(self, labeling_mode='spatial'):
        self.A = self.get_adjacency_matrix(labeling_mode)
        self.num_node = num_node
        self.self_link = self_link
        self.inward = inward
        self.outward = outward
        self.neighbor = neighbor

    def get_adjacency_matrix(self, labeling_mode=None):
        if labeling_mode is None:
            return self.A
        if labeling_mode == 'spatial':
            return tools.get_spatial_graph(num_node, self_link, inward, outward)
        if labeling_mode == 'GIN':
            return tools.get_spatial_graph(num_node, self_link, inward, outward, normalize=False)[1:]
        raise ValueError()
