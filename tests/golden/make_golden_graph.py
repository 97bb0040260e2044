"""Generates tests/golden/adjacency_f32.npy by IMPORTING the reference's graph
package (numpy only).  Runs only in the build container (needs /root/reference)."""
import hashlib, sys, os
import numpy as np
sys.path.insert(0, "/root/reference")
from graph.ntu_rgb_d import Graph  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
A = Graph().A.astype(np.float32)
np.save(os.path.join(here, "adjacency_f32.npy"), A)
G = Graph("GIN").A.astype(np.float32)
np.save(os.path.join(here, "adjacency_gin_f32.npy"), G)
print(A.shape, hashlib.sha256(A.tobytes()).hexdigest())
