"""Synthetic tetrahedral grids, query clouds and `.tet` text IO.

The reference obtains its grids from the external QuarTet binary
(/root/reference/utils/dataloder_helper.py:30-69), which is not available; only
res-40/50 grids ship.  SURVEY.md section 8(d) therefore defines a synthetic "res=R"
grid: (R/2)^3 cubes on [0,1]^3, each split into six Kuhn tetrahedra around the main
diagonal, every tet positively oriented (the convention asserted by
/root/reference/utils/mesh_utils.py:197-219), shifted by -0.5
(/root/reference/train_multigpu.py:65-66).  Per shape the interior vertices
(mask rule of dataloder_helper.py:66-68) are jittered by U(-0.1, 0.1)*h per
coordinate, h = 2/R, seed 1000+b.  Queries follow
/root/reference/dataloader.py:108: 1.05 * (U[0,1)^3 - 0.5), seed 2000+b.

All generators are numpy-only and deterministic so the CPU oracle and the HIP
path see bit-identical inputs.
"""
from __future__ import annotations

import itertools

import numpy as np

__all__ = [
    "kuhn_grid", "jittered_positions", "gather_tets", "random_queries",
    "make_case", "read_tet", "write_tet", "tet_orientation", "project_faces", "pixel_grid",
]


def _perm_sign(p):
    s = 1
    p = list(p)
    for i in range(len(p)):
        while p[i] != i:
            j = p[i]
            p[i], p[j] = p[j], p[i]
            s = -s
    return s


def kuhn_grid(res: int):
    """Vertices [V,3] float64 on [0,1]^3 and tets [T,4] int32 of the res=R Kuhn grid.

    T = 6*(R/2)^3 = 0.75*R^3, V = (R/2+1)^3.  Vertex id = (ix*n1 + iy)*n1 + iz.
    Tets are enumerated cube-major (ix, iy, iz) then by axis permutation, so
    consecutive tet ids are spatially adjacent.
    """
    if res % 2 or res < 2:
        raise ValueError("res must be an even integer >= 2")
    n = res // 2
    n1 = n + 1
    ax = np.arange(n1, dtype=np.float64) / n
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)

    ci, cj, ck = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    corner = np.stack([ci, cj, ck], -1).reshape(-1, 3)          # [C,3]
    eye = np.eye(3, dtype=np.int64)
    tets = []
    for perm in itertools.permutations(range(3)):
        v0 = corner
        v1 = v0 + eye[perm[0]]
        v2 = v1 + eye[perm[1]]
        v3 = v2 + eye[perm[2]]
        quad = [v0, v1, v2, v3]
        # det[(v1-v0),(v2-v0),(v3-v0)] = sign(perm); make every tet positive
        if _perm_sign(perm) < 0:
            quad[2], quad[3] = quad[3], quad[2]
        ids = [(q[:, 0] * n1 + q[:, 1]) * n1 + q[:, 2] for q in quad]
        tets.append(np.stack(ids, -1))
    tets = np.stack(tets, 1).reshape(-1, 4).astype(np.int32)     # cube-major
    return verts, tets


def jittered_positions(verts01: np.ndarray, res: int, batch: int, jitter: float = 0.1,
                       seed0: int = 1000) -> np.ndarray:
    """[B,V,3] float32 vertex positions: verts-0.5 plus per-shape interior jitter."""
    h = 2.0 / res
    interior = np.logical_and(verts01 > 0, verts01 < 1)           # per coordinate
    out = np.empty((batch,) + verts01.shape, dtype=np.float32)
    for b in range(batch):
        rng = np.random.default_rng(seed0 + b)
        d = rng.uniform(-jitter, jitter, size=verts01.shape) * h
        out[b] = (verts01 - 0.5 + d * interior).astype(np.float32)
    return out


def gather_tets(pos_bxvx3: np.ndarray, tets: np.ndarray) -> np.ndarray:
    """[B,T,4,3] float32 — same gather as /root/reference/layers/DefTet/deftet.py:65-68."""
    return np.ascontiguousarray(pos_bxvx3[:, tets.astype(np.int64), :])


def random_queries(batch: int, n_query: int, seed0: int = 2000) -> np.ndarray:
    """[B,Q,3] float32 in [-0.525, 0.525)^3 (/root/reference/dataloader.py:108)."""
    out = np.empty((batch, n_query, 3), dtype=np.float32)
    for b in range(batch):
        rng = np.random.default_rng(seed0 + b)
        out[b] = (1.05 * (rng.random((n_query, 3)) - 0.5)).astype(np.float32)
    return out


def tet_orientation(tet_bxtx4x3: np.ndarray) -> np.ndarray:
    """((b-a) x (c-a)) . (d-a) in float64, shape [B,T]."""
    t = tet_bxtx4x3.astype(np.float64)
    a, b, c, d = t[..., 0, :], t[..., 1, :], t[..., 2, :], t[..., 3, :]
    return np.einsum("...i,...i->...", np.cross(b - a, c - a), d - a)


def make_case(res: int, n_query: int, batch: int, jitter: float = 0.1):
    """Returns (tet_bxtx4x3 f32, queries_bxqx3 f32, tets int32 [T,4], n_vertex)."""
    verts, tets = kuhn_grid(res)
    pos = jittered_positions(verts, res, batch, jitter)
    tet_pos = gather_tets(pos, tets)
    if not (tet_orientation(tet_pos) > 0).all():
        raise AssertionError("jitter inverted a tetrahedron")
    return tet_pos, random_queries(batch, n_query), tets, verts.shape[0]


# ---------------------------------------------------------------------------------
# `.tet` text format (/root/reference/utils/tet_utils.py:378-400,
# /root/reference/utils/dataloder_helper.py:45-59): header "tet <n_vert> <n_tet>",
# then n_vert lines "x y z", then n_tet lines "i j k l" (0-based).
# ---------------------------------------------------------------------------------
def read_tet(path: str):
    with open(path, "r") as f:
        head = f.readline().strip().split(" ")
        n_vert, n_tet = int(head[1]), int(head[2])
        body = np.loadtxt(f, dtype=np.float64, max_rows=n_vert, ndmin=2)
        tets = np.loadtxt(f, dtype=np.int64, max_rows=n_tet, ndmin=2)
    if body.shape != (n_vert, 3) or tets.shape != (n_tet, 4):
        raise ValueError("malformed .tet file %s" % path)
    return body, tets


def write_tet(path: str, verts: np.ndarray, tets: np.ndarray) -> None:
    with open(path, "w") as f:
        f.write("tet %d %d\n" % (verts.shape[0], tets.shape[0]))
        np.savetxt(f, verts, fmt="%.9g")
        np.savetxt(f, tets, fmt="%d")


# ---------------------------------------------------------------------------------
# Rasterizer inputs (BASELINE configs[4]): the grid scaled by `coef`
# (diff_render/diftet_6_subdiv expconfig.py:53-56), one camera at radius `cam_z` looking down -z with the
# NeRF-blender projection vector [2f/W, 2f/H, -1] (2_data/load_blender.py:189-190), image coordinates
# multiplied by 1000 (3_model/deftet.py:459-468); per-vertex RGBA features U[0,1).
# ---------------------------------------------------------------------------------
def project_faces(verts01, face_fx3, rot=(0.35, 0.5), cam_z=4.0, focal=1111.0 / 800.0 * 2.0, mult=1000.0, coef=2.5, seed=0):
    """face_z [1,F,3], face_xy [1,F,3,2], feat [1,F,3,4] (float32) of the indexed triangles `face_fx3`."""
    p = (np.asarray(verts01, np.float64) - 0.5) * coef
    ax, ay = rot
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    p = p @ (Rx @ Ry).T
    pc = p - np.array([0, 0, cam_z])
    xy3 = pc * np.array([focal, focal, -1.0])
    xy = xy3[:, :2] / xy3[:, 2:3] * mult
    feat_v = np.random.default_rng(seed).random((p.shape[0], 4))
    f3 = np.asarray(face_fx3, np.int64)
    return (pc[f3][:, :, 2][None].astype(np.float32), xy[f3][None].astype(np.float32), feat_v[f3][None].astype(np.float32))


def pixel_grid(n, mult=1000.0):
    """pixel centres [1,n*n,2] on [-1,1]^2 * mult and render ranges [1,n*n,2] = [-1000, 0] (3_model/deftet.py:459-468)."""
    a = (np.arange(n) + 0.5) / n * 2 - 1
    X, Y = np.meshgrid(a, a, indexing="xy")
    pix = np.stack([X, Y], -1).reshape(1, -1, 2) * mult
    rngs = np.zeros_like(pix)
    rngs[..., 0] = -1000.0
    return pix.astype(np.float32), rngs.astype(np.float32)
