"""NTU-RGB+D 25-joint skeleton graph with the public names of the reference's graph/ntu_rgb_d.py:6-40
(`num_node`, `self_link`, `inward_ori_index`, `inward`, `outward`, `neighbor`, `Graph(labeling_mode).A`).

The skeleton is kept as the 1-based parent of every Kinect-v2 joint (0 marks the root, joint 21 "spine shoulder");
the (child, parent) link lists are derived from it in child order, which is the order of the reference's literal.
"""
from graph import tools

#          joint:  1   2   3  4   5  6  7  8   9 10  11  12 13  14  15  16 17  18  19  20 21  22 23  24  25
_PARENT_1BASED = (2, 21, 21, 3, 21, 5, 6, 7, 21, 9, 10, 11, 1, 13, 14, 15, 1, 17, 18, 19, 0, 23, 8, 25, 12)
num_node = len(_PARENT_1BASED)
inward_ori_index = [(child, parent) for child, parent in enumerate(_PARENT_1BASED, start=1) if parent]
inward = [(child - 1, parent - 1) for child, parent in inward_ori_index]
outward = [link[::-1] for link in inward]
self_link = list(zip(range(num_node), range(num_node)))
neighbor = inward + outward

_BUILDERS = {
    'spatial': lambda: tools.get_spatial_graph(num_node, self_link, inward, outward),
    # 'GIN' (graph/ntu_rgb_d.py:33-39): un-normalised inward / outward slices, no identity slice
    'GIN': lambda: tools.get_spatial_graph(num_node, self_link, inward, outward, normalize=False)[1:],
}


class Graph:
    num_node, self_link, inward, outward, neighbor = num_node, self_link, inward, outward, neighbor

    def __init__(self, labeling_mode='spatial'):
        self.A = self.get_adjacency_matrix(labeling_mode)

    def get_adjacency_matrix(self, labeling_mode=None):
        if labeling_mode is None:
            return self.A
        if labeling_mode not in _BUILDERS:
            raise ValueError()
        return _BUILDERS[labeling_mode]()
