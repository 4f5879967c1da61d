"""find_k_largest with the reference's signature (util/algorithm.py:155-167; numba heap there)."""
import numpy as np


def find_k_largest(K, candidates):
    c = np.asarray(candidates)
    K = min(K, len(c))
    part = np.argpartition(-c, K - 1)[:K]
    order = part[np.argsort(-c[part], kind='stable')]
    return [int(i) for i in order], [float(c[i]) for i in order]
