"""Shared helpers for the parity tests (oracle side is numpy/torch-CPU; device side goes through the C ABI)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class SyntheticData(object):
    """Raw_data-shaped synthetic LTR set: features ~ U(-1,1), labels categorical
    (0.52,0.32,0.13,0.02,0.01) with >= 1 relevant doc per list, optional short (padded) lists."""

    def __init__(self, n_queries, list_len, feature_size, seed=1234, pad_frac=0.0, sparse=0.0):
        rng = np.random.RandomState(seed)
        self.feature_size = feature_size
        self.rank_list_size = list_len
        n_docs = n_queries * list_len
        feats = rng.uniform(-1, 1, size=(n_docs, feature_size)).astype(np.float32)
        if sparse > 0:
            feats[rng.uniform(size=feats.shape) < sparse] = 0.0
        labels = rng.choice(5, size=(n_queries, list_len), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
        none = labels.sum(axis=1) == 0
        labels[none, rng.randint(0, list_len, size=int(none.sum()))] = 1.0
        il = np.arange(n_docs, dtype=np.int32).reshape(n_queries, list_len)
        if pad_frac > 0:
            short = rng.uniform(size=n_queries) < pad_frac
            cut = rng.randint(2, list_len, size=n_queries)
            for q in np.nonzero(short)[0]:
                il[q, cut[q]:] = -1
                labels[q, cut[q]:] = 0.0
                if labels[q].sum() == 0:
                    labels[q, 0] = 1.0
        # the zero pad row that Raw_data.pad appends
        self.features = np.concatenate([feats, np.zeros((1, feature_size), np.float32)], axis=0)
        self.initial_list = il
        self.labels = labels
        self.qids = [str(i) for i in range(n_queries)]


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def relerr_floor(a, b, floor=1e-4):
    """L2-relative error with an absolute floor of `floor` per element: tensors whose true value is identically
    zero (e.g. the output-layer beta under a shift-invariant loss) are compared on an absolute scale."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), floor * np.sqrt(b.size)))
