"""ctypes front-end for the CPU oracle (oracle/libdeftet_oracle.so) and oracle/_ref.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing in deftet_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdeftet_oracle.so")
LIB_FMA_PATH = os.path.join(HERE, "libdeftet_oracle_fma.so")    # same source, -mfma -ffp-contract=fast (flip counting only)
REF_DIR = os.path.join(HERE, "_ref")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> None:
    """Compile the restatement and (when /root/reference is present) oracle/_ref."""
    src = [os.path.join(HERE, f) for f in ("deftet_oracle.c", "deftet_oracle_surface.c", "deftet_oracle_render.c", "deftet_oracle_sign.c", "Makefile")]
    # (the FMA variant is a diagnostic: it is built along with the oracle when the compiler can, and its absence — a host
    # without -mfma — neither makes the oracle stale nor fails the build; lib_fma() says so)
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    need_ref = os.path.isdir("/root/reference/utils/lib") and not all(
        os.path.exists(os.path.join(REF_DIR, n + "_run.so"))
        for n in ("tet_adj_share", "tet_face_adj", "tet_point_adj", "colaps_v"))
    if force or stale or need_ref:
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_point_in_tet_f32_omp.restype = C.c_int
    return _lib


_lib_fma = None


class FmaOracleUnavailable(RuntimeError):
    """the -mfma build of the restatement (a flip-counting diagnostic, never the parity oracle) cannot be used on this host"""


def _cpu_has_fma():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return " fma " in line + " "
    except OSError:
        pass
    return True                                               # unknown: let the loader decide


def lib_fma():
    global _lib_fma
    if _lib_fma is None:
        build()
        if not os.path.exists(LIB_FMA_PATH) or os.path.getmtime(LIB_FMA_PATH) < os.path.getmtime(os.path.join(HERE, "deftet_oracle.c")):
            subprocess.call(["make", "-s", "-C", HERE, "fma"])
        if not os.path.exists(LIB_FMA_PATH):
            raise FmaOracleUnavailable("libdeftet_oracle_fma.so could not be built on this host (needs -mfma)")
        if not _cpu_has_fma():
            raise FmaOracleUnavailable("this CPU has no FMA instructions: libdeftet_oracle_fma.so would fault")
        _lib_fma = C.CDLL(LIB_FMA_PATH)
        _lib_fma.oracle_point_in_tet_f32_omp.restype = C.c_int
    return _lib_fma


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------- A1 / A1b
def point_in_tet(tet_bxtx4x3, pts_bxqx3, omp=False, return_executed=False):
    tet = _c(tet_bxtx4x3, np.float32)
    pts = _c(pts_bxqx3, np.float32)
    B, T = tet.shape[0], tet.shape[1]
    Q = pts.shape[1]
    out = np.empty((B, Q, 1), np.float32)
    if omp:
        n = lib().oracle_point_in_tet_f32_omp(_p(tet, _f32p), _p(pts, _f32p), _p(out, _f32p), B, T, Q)
        return (out, n) if return_executed else out
    ex = C.c_longlong(0)
    lib().oracle_point_in_tet_f32(_p(tet, _f32p), _p(pts, _f32p), _p(out, _f32p), B, T, Q, C.byref(ex))
    return (out, ex.value) if return_executed else out


def point_in_tet_contracted(tet_bxtx4x3, pts_bxqx3):
    """The scan of point_in_tet(omp=True) from the build with fused multiply-adds (what nvcc's default contraction does to
    check_condition_tet_for.cu:105-121).  For counting flipped decisions; NOT the parity oracle."""
    tet = _c(tet_bxtx4x3, np.float32)
    pts = _c(pts_bxqx3, np.float32)
    out = np.empty((tet.shape[0], pts.shape[1], 1), np.float32)
    lib_fma().oracle_point_in_tet_f32_omp(_p(tet, _f32p), _p(pts, _f32p), _p(out, _f32p), tet.shape[0], tet.shape[1], pts.shape[1])
    return out


def point_in_tet_margin(tet_tx4x3, pts_qx3):
    tet = _c(tet_tx4x3, np.float32)
    pts = _c(pts_qx3, np.float32)
    out = np.empty(pts.shape[0], np.float64)
    lib().oracle_point_in_tet_margin_f32(_p(tet, _f32p), _p(pts, _f32p), _p(out, _f64p),
                                         tet.shape[0], pts.shape[0])
    return out


def bary(tet_bxtx4x3, pts_bxqx3, cond_bxq):
    tet = _c(tet_bxtx4x3, np.float32)
    pts = _c(pts_bxqx3, np.float32)
    cond = _c(cond_bxq, np.float32).reshape(pts.shape[0], pts.shape[1])
    out = np.empty(pts.shape[:2] + (4,), np.float32)
    lib().oracle_bary_f32(_p(tet, _f32p), _p(pts, _f32p), _p(cond, _f32p), _p(out, _f32p),
                          tet.shape[0], tet.shape[1], pts.shape[1])
    return out


def bary_bwd(tet_bxtx4x3, pts_bxqx3, cond_bxq, grad_w_bxqx4):
    """dL/dtet [B,T,4,3] in plain fp32 C (scatter-add in query order): the CPU leg of bench.py's fwd+bwd figure."""
    tet = _c(tet_bxtx4x3, np.float32)
    pts = _c(pts_bxqx3, np.float32)
    cond = _c(cond_bxq, np.float32).reshape(pts.shape[0], pts.shape[1])
    gw = _c(grad_w_bxqx4, np.float32)
    out = np.zeros(tet.shape, np.float32)
    lib().oracle_bary_bwd_f32(_p(tet, _f32p), _p(pts, _f32p), _p(cond, _f32p), _p(gw, _f32p), _p(out, _f32p),
                              tet.shape[0], tet.shape[1], pts.shape[1])
    return out


def bary_torch(a, b, c, d, p):
    """Restatement of /root/reference/utils/tet_utils.py:25-45 on torch tensors
    (any dtype) — the autograd oracle for A1b."""
    import torch

    def triple(x, y, z):
        return torch.sum(x * torch.cross(y, z, dim=-1), dim=-1)

    vap, vbp = p - a, p - b
    vab, vac, vad = b - a, c - a, d - a
    vbc, vbd = c - b, d - b
    va6 = triple(vbp, vbd, vbc)
    vb6 = triple(vap, vac, vad)
    vc6 = triple(vap, vad, vab)
    vd6 = triple(vap, vab, vac)
    v6 = 1 / triple(vab, vac, vad)
    return va6 * v6, vb6 * v6, vc6 * v6, vd6 * v6


def point_in_tet_bwd_torch(tet_bxtx4x3, pts_bxqx3, cond_bxq, grad_w_bxqx4, dtype=None):
    """dL/dtet [B,T,4,3] by torch autograd through bary_torch + gather.
    Misses (cond < 0) contribute nothing."""
    import torch
    dtype = dtype or torch.float64
    tet = torch.as_tensor(np.asarray(tet_bxtx4x3)).to(dtype).clone().requires_grad_(True)
    pts = torch.as_tensor(np.asarray(pts_bxqx3)).to(dtype)
    cond = torch.as_tensor(np.asarray(cond_bxq)).reshape(pts.shape[0], pts.shape[1])
    gw = torch.as_tensor(np.asarray(grad_w_bxqx4)).to(dtype)
    hit = cond >= 0
    idx = cond.clamp(min=0).long()
    g = torch.gather(tet, 1, idx[:, :, None, None].expand(-1, -1, 4, 3))
    w = torch.stack(bary_torch(g[:, :, 0], g[:, :, 1], g[:, :, 2], g[:, :, 3], pts), -1)
    loss = (w * gw * hit[..., None].to(dtype)).sum()
    loss.backward()
    return w.detach().numpy(), tet.grad.numpy()


# --------------------------------------------------------------------------- builders
def tet_adj_share(tet_list, n_point):
    tet = _c(tet_list, np.int32)
    T = tet.shape[0]
    out = np.zeros((T * 8, 3), np.int32)
    n = np.zeros(1, np.int32)
    lib().oracle_tet_adj_share(_p(tet, _i32p), _p(out, _i32p), _p(n, _i32p), int(n_point), T)
    return out[: n[0] * 2]


def tet_face_adj(tet_list, n_point, wrap32=True):
    tet = _c(tet_list, np.int32)
    T = tet.shape[0]
    cnt = C.c_longlong(0)
    lib().oracle_tet_face_adj(_p(tet, _i32p), None, C.byref(cnt), int(n_point), T, int(wrap32))
    out = np.zeros((max(cnt.value, 1), 2), np.int32)
    lib().oracle_tet_face_adj(_p(tet, _i32p), _p(out, _i32p), C.byref(cnt), int(n_point), T, int(wrap32))
    return out[: cnt.value]


def tet_point_adj(tet_list, n_point):
    tet = _c(tet_list, np.int32)
    T = tet.shape[0]
    out = np.zeros((max(T * 12, 1), 2), np.int32)
    n = np.zeros(1, np.int32)
    lib().oracle_tet_point_adj(_p(tet, _i32p), _p(out, _i32p), _p(n, _i32p), int(n_point), T)
    return out[: n[0]]


def colaps_v(points_nx3):
    pts = _c(points_nx3, np.float32)
    N = pts.shape[0]
    m = np.zeros(N, np.int32)
    inv = np.zeros(N, np.int32)
    n = np.zeros(1, np.int32)
    lib().oracle_colaps_v(_p(pts, _f32p), _p(m, _i32p), _p(inv, _i32p), _p(n, _i32p), N)
    return m, inv[: n[0]]


def tet_to_face(tet_list, n_point, with_boundary=False):
    tet = _c(tet_list, np.int32)
    T = tet.shape[0]
    f3 = np.zeros((T * 4, 3), np.int64)
    t2 = np.zeros((T * 4, 2), np.int64)
    tf2 = np.zeros((T * 4, 2), np.int64)
    b3 = np.zeros((T * 4, 3), np.int64)
    nf, nb, nm = (np.zeros(1, np.int32) for _ in range(3))
    lib().oracle_tet_to_face(_p(tet, _i32p), int(n_point), T, int(with_boundary), _p(f3, _i64p), _p(t2, _i64p),
                             _p(tf2, _i64p), _p(b3, _i64p), _p(nf, _i32p), _p(nb, _i32p), _p(nm, _i32p))
    return f3[: nf[0]], t2[: nf[0]], tf2[: nf[0]], b3[: nb[0]], int(nm[0])


def tet_neighbours(tet_list, n_point):
    """(tet_neighbour_idx [T,4], tet_face_tetidx [4T,2]) restated in plain loops from
    diff_render/diftet_6_subdiv/3_model/utils_tetsv.py:41-58 and utils/tet_utils.py:288-298, driven by the
    first-seen unique-face table (oracle tet_to_face with_boundary=True): shared faces are visited in table
    order and append the partner to both owners' rows; a lone owner's face row is padded with 0."""
    tet = _c(tet_list, np.int32)
    T = tet.shape[0]
    _, t2, tf2, _, n_multi = tet_to_face(tet, n_point, with_boundary=True)
    if n_multi:
        raise ValueError("face shared by more than two tetrahedra")
    nbr = -np.ones((T, 4), np.int64)
    fill = np.zeros(T, np.int64)
    owners = np.zeros((T * 4, 2), np.int64)
    for (t0, t1), (l0, l1) in zip(t2, tf2):
        if t1 >= 0:
            nbr[t0, fill[t0]] = t1
            fill[t0] += 1
            nbr[t1, fill[t1]] = t0
            fill[t1] += 1
            owners[4 * t0 + l0] = (t0, t1)
            owners[4 * t1 + l1] = (t0, t1)
        else:
            owners[4 * t0 + l0] = (t0, 0)
    return nbr, owners


# --------------------------------------------------------------------------- surface ops
def face_edge_adj(face_fx3x3, n_max_nei=30):
    face = _c(face_fx3x3, np.float32)
    F = face.shape[0]
    adj = -np.ones((F, n_max_nei), np.float32)
    lib().oracle_face_edge_adj_f32(_p(face, _f32p), _p(adj, _f32p), F, n_max_nei)
    return adj


def tri_dist_fwd(pts_bxpx3, face_bxfx3x3, n_face_b):
    pts = _c(pts_bxpx3, np.float32)
    face = _c(face_bxfx3x3, np.float32)
    nfb = _c(n_face_b, np.float32)
    B, P = pts.shape[:2]
    d = np.zeros((B, P, 1), np.float32)
    f = np.zeros((B, P, 1), np.float32)
    lib().oracle_tri_dist_fwd_f32(_p(pts, _f32p), _p(face, _f32p), _p(nfb, _f32p), _p(d, _f32p), _p(f, _f32p),
                                  B, P, face.shape[1])
    return d, f


def tri_dist_bwd(pts_bxpx3, face_bxfx3x3, closest_f, dl_dd):
    pts = _c(pts_bxpx3, np.float32)
    face = _c(face_bxfx3x3, np.float32)
    cf = _c(closest_f, np.float32)
    g = _c(dl_dd, np.float32)
    B, P = pts.shape[:2]
    out = np.zeros(face.shape, np.float32)
    lib().oracle_tri_dist_bwd_f32(_p(pts, _f32p), _p(face, _f32p), _p(cf, _f32p), _p(g, _f32p), _p(out, _f32p),
                                  B, P, face.shape[1])
    return out


def nn_index(queries_bxnx3, points_bxmx3):
    q = _c(queries_bxnx3, np.float32)
    p = _c(points_bxmx3, np.float32)
    B, N = q.shape[:2]
    out = np.zeros((B, N), np.int32)
    lib().oracle_nn_index_f32(_p(q, _f32p), _p(p, _f32p), _p(out, _i32p), B, N, p.shape[1])
    return out


# --------------------------------------------------------------------------- rasterizer (parity unpinned)
RASTER_NEAREST, RASTER_FIRST = 0, 1     # which kept faces a saturated pixel records (deftet_oracle_render.c header)


def sparse_render_fwd(pixel_bxpx2, range_bxpx2, face_z_bxfx3, face_xy_bxfx3x2, face_feat_bxfx3xd, knum=300, eps=1e-8,
                      policy=RASTER_NEAREST):
    pix, rng = _c(pixel_bxpx2, np.float32), _c(range_bxpx2, np.float32)
    fz, fxy, ff = _c(face_z_bxfx3, np.float32), _c(face_xy_bxfx3x2, np.float32), _c(face_feat_bxfx3xd, np.float32)
    B, P = pix.shape[:2]
    F, D = fz.shape[1], ff.shape[3]
    feat = np.zeros((B, P, knum, D), np.float32)
    face = np.zeros((B, P, knum), np.int64)
    w = np.zeros((B, P, knum, 3), np.float32)
    lib().oracle_sparse_render_fwd_policy_f32(_p(pix, _f32p), _p(rng, _f32p), _p(fz, _f32p), _p(fxy, _f32p), _p(ff, _f32p),
                                              _p(feat, _f32p), _p(face, _i64p), _p(w, _f32p), B, P, F, D, int(knum), C.c_float(eps),
                                              int(policy))
    return feat, face, w


def sparse_render_torch(pixel, face_xy, face_feat, face_idx, eps=1e-8):
    """Differentiable (torch, any dtype) re-evaluation of the interpolated features for GIVEN
    face indices — the autograd oracle for the rasterizer backward."""
    import torch
    B, P, K = face_idx.shape
    valid = face_idx >= 0
    fi = face_idx.clamp(min=0)
    xy = torch.gather(face_xy, 1, fi.reshape(B, P * K, 1, 1).expand(-1, -1, 3, 2)).reshape(B, P, K, 3, 2)
    ft = torch.gather(face_feat, 1, fi.reshape(B, P * K, 1, 1).expand(-1, -1, 3, face_feat.shape[-1])).reshape(B, P, K, 3, -1)
    ax, ay, bx, by, cx, cy = xy[..., 0, 0], xy[..., 0, 1], xy[..., 1, 0], xy[..., 1, 1], xy[..., 2, 0], xy[..., 2, 1]
    px, py = pixel[..., 0:1], pixel[..., 1:2]
    m, pp, n, q, s, t = bx - ax, by - ay, cx - ax, cy - ay, px - ax, py - ay
    k1, k2, k3 = s * q - n * t, m * t - s * pp, m * q - n * pp
    w1, w2 = k1 / (k3 + eps), k2 / (k3 + eps)
    w0 = 1 - w1 - w2
    feat = w0[..., None] * ft[..., 0, :] + w1[..., None] * ft[..., 1, :] + w2[..., None] * ft[..., 2, :]
    return feat * valid[..., None].to(feat.dtype)


# --------------------------------------------------------------------------- oracle/_ref
class RefBuilders:
    """The reference's own native builders (utils/lib/*/run.cpp) compiled by
    oracle/Makefile into oracle/_ref/.  Same buffer sizing as the reference's
    interface.py files."""

    def __init__(self):
        def load(n):
            path = os.path.join(REF_DIR, n + "_run.so")
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            return C.CDLL(path)
        self.adj_share = load("tet_adj_share")
        self.face_adj = load("tet_face_adj")
        self.point_adj = load("tet_point_adj")
        self.colaps = load("colaps_v")

    @staticmethod
    def available():
        return all(os.path.exists(os.path.join(REF_DIR, n + "_run.so"))
                   for n in ("tet_adj_share", "tet_face_adj", "tet_point_adj", "colaps_v"))

    def tet_adj_share(self, tet_list, n_point):           # utils/lib/tet_adj_share/interface.py:19-37
        tet = _c(tet_list, np.int32)
        out = np.zeros((tet.shape[0] * 8, 3), np.int32)
        n = np.zeros(1, np.int32)
        self.adj_share.run(_p(tet, _i32p), _p(out, _i32p), _p(n, _i32p), C.c_int(int(n_point)), C.c_int(tet.shape[0]))
        return out[: n[0] * 2]

    def tet_face_adj(self, tet_list, n_point):            # utils/lib/tet_face_adj/interface.py:20-32
        tet = _c(tet_list, np.int32)
        out = np.zeros((tet.shape[0] * 4 * 50, 2), np.int32)
        n = np.zeros(1, np.int32)
        self.face_adj.run(_p(tet, _i32p), _p(out, _i32p), _p(n, _i32p), C.c_int(int(n_point)), C.c_int(tet.shape[0]))
        return out[: n[0]]

    def tet_point_adj(self, tet_list, n_point):           # utils/lib/tet_point_adj/interface.py:20-39
        tet = _c(tet_list, np.int32)
        out = np.zeros((tet.shape[0] * 12, 2), np.int32)
        n = np.zeros(1, np.int32)
        self.point_adj.run(_p(tet, _i32p), _p(out, _i32p), _p(n, _i32p), C.c_int(int(n_point)), C.c_int(tet.shape[0]))
        return out[: n[0]]

    def colaps_v(self, points_nx3):                       # utils/lib/colaps_v/interface.py:20-35
        pts = _c(points_nx3, np.float32)
        N = pts.shape[0]
        m = np.zeros(N, np.int32)
        inv = np.zeros(N, np.int32)
        n = np.zeros(1, np.int32)
        self.colaps.run(_p(pts, _f32p), _p(m, _i32p), _p(inv, _i32p), _p(n, _i32p), C.c_int(N))
        return m, inv[: n[0]]


# ----------------------------------------------------------------------------- N2 vertex <-> tet gather
def tet_gather(pos_bxvx3, tet_idx):
    """numpy restatement of torch.gather(vertice_pos, tetrahedron_bxfx4)
    (/root/reference/layers/DefTet/deftet.py:65-68): [B,T,4,3]."""
    pos = np.asarray(pos_bxvx3, dtype=np.float32)
    idx = np.asarray(tet_idx, dtype=np.int64)
    if idx.ndim == 2:
        idx = np.broadcast_to(idx[None], (pos.shape[0],) + idx.shape)
    return np.stack([pos[b][idx[b]] for b in range(pos.shape[0])]).astype(np.float32)


def tet_gather_bwd(grad_tet_bxtx4x3, tet_idx, n_vertex):
    """Backward of the gather in the FIXED fp32 summation order of the HIP kernel
    (deftet_amd/csrc/vertex_ops.hip k_gather_bwd), so the comparison is bit-exact: the incidences
    4*t+corner of a vertex are taken in ascending order; position j of that list goes to partial
    sum j % 4 (each accumulated sequentially from 0, np.add.at applies updates in index order),
    and the result is (s0 + s1) + (s2 + s3).  torch's scatter-add computes the same sum in an
    unspecified order; equality with it is checked to fp32 round-off in the tests."""
    g = np.asarray(grad_tet_bxtx4x3, dtype=np.float32)
    idx = np.asarray(tet_idx, dtype=np.int64)
    B = g.shape[0]
    if idx.ndim == 2:
        idx = np.broadcast_to(idx[None], (B,) + idx.shape)
    out = np.zeros((B, n_vertex, 3), dtype=np.float32)
    for b in range(B):
        flat = idx[b].reshape(-1)
        order = np.argsort(flat, kind="stable")                       # ascending slot inside each vertex
        sv = flat[order]
        start = np.searchsorted(sv, sv, side="left")                  # first list position of the vertex
        lane = (np.arange(sv.size) - start) % 4
        part = np.zeros((n_vertex, 4, 3), dtype=np.float32)
        np.add.at(part, (sv, lane), g[b].reshape(-1, 3)[order])
        out[b] = (part[:, 0] + part[:, 1]) + (part[:, 2] + part[:, 3])
    return out


# ----------------------------------------------------------------------------- N3 render-side rebuilds
# Vectorised numpy restatements of /root/reference/diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py
# (the reference versions loop in Python, matchedgelist is O(E*T), generate_point_adj is a dense
# P x P matrix); pinned by tests/golden/n3_*.npz, produced by the reference functions themselves.
_EDGE_ENDS = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])          # :190


def generate_edge(tet_tx4):
    """:184-203 — unique (min,max) rows, np.unique(axis=0) order."""
    t = np.asarray(tet_tx4, dtype=np.int64)
    a, b = t[:, _EDGE_ENDS[:, 0]].T.reshape(-1), t[:, _EDGE_ENDS[:, 1]].T.reshape(-1)
    e = np.stack([np.minimum(a, b), np.maximum(a, b)], axis=1)
    return np.unique(e, axis=0) if e.size else e.reshape(0, 2)


def generate_tet_edge_idx(tet_tx4, edges_ex2):
    """:206-236 — row of every tet edge in the unique list (matchedgelist as a key lookup)."""
    t = np.asarray(tet_tx4, dtype=np.int64)
    n = int(max(t.max(initial=-1), np.asarray(edges_ex2).max(initial=-1))) + 1
    ekey = edges_ex2[:, 0] * n + edges_ex2[:, 1]                                 # ascending, unique
    a, b = t[:, _EDGE_ENDS[:, 0]], t[:, _EDGE_ENDS[:, 1]]
    key = np.minimum(a, b) * n + np.maximum(a, b)
    return np.searchsorted(ekey, key).astype(np.int64)


def generate_subdivision(tet_tx4, points_px3, feat_pxk, sig=None):
    """:255-301."""
    t = np.asarray(tet_tx4, dtype=np.int64)
    pts, feat = np.asarray(points_px3, dtype=np.float32), np.asarray(feat_pxk, dtype=np.float32)
    edges = generate_edge(t)
    te = generate_tet_edge_idx(t, edges)
    pn = np.concatenate([pts, (pts[edges[:, 0]] + pts[edges[:, 1]]) / 2], axis=0)
    fn = np.concatenate([feat, (feat[edges[:, 0]] + feat[edges[:, 1]]) / 2], axis=0)
    P = pts.shape[0]
    a, b, c, d = t.T
    ab, ac, ad, bc, bd, cd = (te + P).T
    ch = np.stack([np.stack(x, axis=1) for x in ([a, ab, ac, ad], [b, bc, ab, bd], [c, ac, bc, cd], [d, ad, cd, bd],
                                                 [ab, ac, ad, bd], [ab, ac, bd, bc], [cd, ac, bd, ad], [cd, ac, bc, bd])], axis=1)
    if sig is None:
        tn = ch.reshape(-1, 4)
    else:
        sig = np.asarray(sig, dtype=bool)
        tn = np.concatenate([t[~sig], ch[sig].reshape(-1, 4)], axis=0)
    return pn, fn, tn


def generate_point_adj_idx(n_point, tet_tx4):
    """:108-146 — ascending neighbour table padded with -1, and the degrees as float32 [P,1]."""
    t = np.asarray(tet_tx4, dtype=np.int64)
    i = np.repeat(t, 3, axis=1).reshape(-1)                                       # every ordered pair of distinct corners
    j = t[:, [1, 2, 3, 0, 2, 3, 1, 0, 3, 1, 0, 2]].reshape(-1)
    pair = np.unique(np.stack([i, j], axis=1), axis=0) if t.size else np.zeros((0, 2), dtype=np.int64)
    deg = np.bincount(pair[:, 0], minlength=n_point).astype(np.int64)
    m = int(deg.max()) if n_point else 0
    table = -np.ones((n_point, m), dtype=np.int64)
    start = np.concatenate([[0], np.cumsum(deg)[:-1]]) if n_point else np.zeros(0, dtype=np.int64)
    col = np.arange(pair.shape[0]) - start[pair[:, 0]]
    table[pair[:, 0], col] = pair[:, 1]
    return table, deg.astype(np.float32).reshape(-1, 1)


def delete_tet(tet_tx4, weights_txk, thres=0.01):
    """:171-180."""
    w = np.asarray(weights_txk, dtype=np.float32)
    return np.asarray(tet_tx4)[np.max(w, axis=1) > thres]


def tetweights2tetneighbourweights(weights_txk, nei_tx4, neilevel=1):
    """3_model/deftet.py:316-331."""
    w = np.asarray(weights_txk, dtype=np.float32)
    nei = np.asarray(nei_tx4, dtype=np.int64)
    for _ in range(neilevel):
        t, k = w.shape
        w1 = np.zeros((t + 1, k), dtype=np.float32)
        w1[1:] = w
        w = w1[nei.reshape(-1) + 1].reshape(t, -1)
    return w


# ----------------------------------------------------------------------------- N1 check_sign (parity unpinned)
def check_sign(verts_bxvx3, faces_fx3, points_bxnx3, return_count=False):
    """kal.ops.mesh.check_sign restated (oracle/deftet_oracle_sign.c): bool [B,N]."""
    v, f, p = _c(verts_bxvx3, np.float32), _c(faces_fx3, np.int64), _c(points_bxnx3, np.float32)
    B, V, N, F = v.shape[0], v.shape[1], p.shape[1], f.shape[0]
    out = np.zeros((B, N), np.uint8)
    cnt = np.zeros((B, N), np.int32)
    lib().oracle_check_sign_f32(_p(v, _f32p), _p(f, C.POINTER(C.c_int64)), _p(p, _f32p), _p(out, C.POINTER(C.c_uint8)),
                                _p(cnt, _i32p), B, V, F, N)
    return (out.astype(bool), cnt) if return_count else out.astype(bool)
