"""Icosahedron-based direction basis for integrated positional encoding (host-side, run once).

Mirrors the reference buffer `pos_basis_t` (helper.py:457-531 `generate_basis`): tessellate the
icosahedron faces by `subdivision`, normalise to the sphere, merge duplicate vertices, keep one of
each antipodal pair, reverse the xyz order.  21 directions for subdivision 2.
"""
from __future__ import annotations

import math

import numpy as np
import torch

_GOLDEN = (math.sqrt(5.0) + 1.0) / 2.0
_ICO_VERTS = np.array(
    [(-1, 0, _GOLDEN), (1, 0, _GOLDEN), (-1, 0, -_GOLDEN), (1, 0, -_GOLDEN), (0, _GOLDEN, 1), (0, _GOLDEN, -1),
     (0, -_GOLDEN, 1), (0, -_GOLDEN, -1), (_GOLDEN, 1, 0), (-_GOLDEN, 1, 0), (_GOLDEN, -1, 0), (-_GOLDEN, -1, 0)],
    dtype=np.float64) / math.sqrt(_GOLDEN + 2.0)
_ICO_FACES = np.array(
    [(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3), (2, 7, 3),
     (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11), (9, 11, 2), (9, 2, 5), (7, 2, 11)])


def _pairwise_sqdist(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a, b: [3, n] column sets -> [na, nb] squared distances (via the norm expansion, clipped at 0)."""
    return np.maximum(0, np.sum(a**2, 0)[:, None] + np.sum(b**2, 0)[None, :] - 2 * a.T @ b)


def generate_basis(base_shape: str = "icosahedron", subdivision: int = 2, tol: float = 1e-4) -> torch.Tensor:
    if base_shape != "icosahedron":
        raise NotImplementedError("only the icosahedron basis is used by the reference configs")
    v = int(subdivision)
    weights = np.array([(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)], dtype=np.float64) / v
    cloud = []
    for tri in _ICO_FACES:
        q = weights @ _ICO_VERTS[tri, :]
        cloud.append(q / np.sqrt(np.sum(q**2, 1, keepdims=True)))
    cloud = np.concatenate(cloud, 0)
    near = _pairwise_sqdist(cloud.T, cloud.T) <= tol
    owner = np.array([np.flatnonzero(row)[0] for row in near])       # first duplicate wins
    cloud = cloud[np.unique(owner), :]
    anti = _pairwise_sqdist(cloud.T, -cloud.T) < tol
    cloud = cloud[np.any(np.triu(anti), 1), :]                        # one of each antipodal pair
    return torch.from_numpy(np.ascontiguousarray(cloud[:, ::-1].T)).to(torch.float32)
