#!/usr/bin/env python3
"""Score fusion of separately trained streams (joint / bone / motion): the usual 2-stream evaluation for ST-GCN-family
models -- not part of the reference, which trains the streams independently and stops there.
    python tools/fuse_scores.py --labels val_label.pkl run_joint/scores-50.npy run_bone/scores-50.npy [--weights 1 1]
prints top-1 / top-5 of each stream and of the weighted sum of their class probabilities."""
import argparse
import pickle

import numpy as np


def topk(scores, labels, k):
    idx = np.argsort(-scores, axis=1)[:, :k]
    return float((idx == labels[:, None]).any(1).mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scores", nargs="+")
    ap.add_argument("--labels", required=True, help="*_label.pkl of the test split (data_gen/gen_joint_data.py)")
    ap.add_argument("--weights", type=float, nargs="*")
    a = ap.parse_args()
    with open(a.labels, "rb") as f:
        _, labels = pickle.load(f, encoding="latin1")
    labels = np.asarray(labels, dtype=np.int64)
    S = [np.load(p) for p in a.scores]
    w = a.weights or [1.0] * len(S)
    assert len(w) == len(S) and all(s.shape == S[0].shape for s in S) and len(labels) >= len(S[0])
    labels = labels[:len(S[0])]
    for p, s in zip(a.scores, S):
        print("%-40s top1 %.4f top5 %.4f" % (p, topk(s, labels, 1), topk(s, labels, 5)))
    fused = sum(wi * s for wi, s in zip(w, S))
    print("%-40s top1 %.4f top5 %.4f" % ("fused " + str(w), topk(fused, labels, 1), topk(fused, labels, 5)))


if __name__ == "__main__":
    main()
