"""Joint -> bone stream (reference data_gen/gen_bone_data.py:7-41), fused into the data_bn prologue
of the HIP path instead of being materialised offline.

bone[..., v1-1, :] = joint[..., v1-1, :] - joint[..., v2-1, :] for the 25 (1-based) pairs below; the
pair (21, 21) makes joint 21 (the root, "spine shoulder") identically zero.  The pairs are the
skeleton's (child, parent) edges (graph/ntu_rgb_d.py:8-11) plus that root self-pair; the reference
lists the same pairs for 'xview' and 'xsub'.
"""
import numpy as np

from graph.ntu_rgb_d import inward_ori_index

NTU_BONE_PAIRS = tuple(inward_ori_index[:21]) + ((21, 21),) + tuple(inward_ori_index[21:])


def bone_parent_array(num_node=25, pairs=NTU_BONE_PAIRS):
    """0-based `v2` for every joint (or -1 = keep the joint) as the int32 table the kernels take."""
    bp = np.full(num_node, -1, dtype=np.int32)
    for v1, v2 in pairs:
        bp[v1 - 1] = v2 - 1
    return bp
