"""Host logic: turn the dense stacked adjacency A (K,V,V) into the per-column gather
lists the HIP kernels consume.

out[w] = sum_v x[v] * A[k, v, w]   (einsum 'nkctv,kvw->nctw', reference models/gcn.py:208)

For the NTU graph (reference graph/ntu_rgb_d.py:6-14 through graph/tools.py:22-30) every column
of every slice has at most 4 non-zeros (73 of 1875 entries), so each output joint is a <=4-entry
weighted gather.  The transposed lists (A_k^T) drive the data gradient.
Pure integer / exact-float bookkeeping: the gather weights are the adjacency entries themselves.
"""
import numpy as np

NZMAX = 4


def gather_lists(A, transpose=False):
    """A: (K,V,V) array.  Returns idx int32 (K,V,NZMAX), wt float32 (K,V,NZMAX), nz list[int] (K),
    colsum float32 (K,V).  Unused entries have weight 0 and index = the column itself."""
    A = np.asarray(A, dtype=np.float32)
    K, V, _ = A.shape
    if transpose:
        A = np.transpose(A, (0, 2, 1))
    idx = np.zeros((K, V, NZMAX), dtype=np.int32)
    wt = np.zeros((K, V, NZMAX), dtype=np.float32)
    nz = []
    for k in range(K):
        worst = 1
        for w in range(V):
            rows = np.nonzero(A[k, :, w])[0]
            if len(rows) > NZMAX:
                raise ValueError(
                    "adjacency slice %d column %d has %d non-zeros; the HIP graph kernels support at most %d "
                    "per column" % (k, w, len(rows), NZMAX))
            idx[k, w, :] = w
            for j, v in enumerate(rows):
                idx[k, w, j] = v
                wt[k, w, j] = A[k, v, w]
            worst = max(worst, len(rows))
        nz.append(int(worst))
    colsum = A.sum(axis=1).astype(np.float32)  # sum over v of A[k, v, w]
    return idx, wt, nz, colsum
