"""CPU tests of the host side: the C-ABI library exports what include/deftet_hip.h declares,
the ctypes table covers it, synthetic grids / .tet IO, shape sharding over gloo (world 2)."""
import ctypes
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

from deftet_amd import grids, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "deftet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(deftet_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from deftet_amd import _lib, build
    build.build()
    syms = header_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libdeftet_hip.so does not export %s" % s
    loaded = _lib.load()
    assert loaded.deftet_version() >= 200
    # argument errors are reported through the status code + deftet_last_error, no GPU needed
    assert loaded.deftet_point_in_tet_workspace_bytes(8, 257250, 100000, 0) > 0
    st = loaded.deftet_point_in_tet_f32(None, None, None, None, None, None, None, -1, 1, 1, 0, None, 0, None)
    assert st == -1 and b"negative" in loaded.deftet_last_error()
    st = loaded.deftet_point_in_tet_f32(None, None, None, None, None, None, None, 1, 1 << 24, 1, 0, None, 0, None)
    assert st == -4


def test_ops_fail_loudly_without_gpu_tensors():
    from deftet_amd import _lib
    from deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils import check_condition_f_base
    with pytest.raises(_lib.DefTetHipError):
        check_condition_f_base(torch.zeros(1, 2, 4, 3), torch.zeros(1, 3, 3))
    # the round-5 entry points too: the traversal order and the query-box hint never run anywhere but on the GPU
    from deftet_amd import hip_ops
    with pytest.raises(_lib.DefTetHipError):
        hip_ops.tet_spatial_order(torch.zeros(10, 4, 3))
    with pytest.raises(_lib.DefTetHipError):
        hip_ops.point_in_tet(torch.zeros(1, 2, 4, 3), torch.zeros(1, 3, 3), order="auto", query_box="track")
    with pytest.raises(ValueError):
        hip_ops.pit_kernel_name(hip_ops.PIT_AUTO)               # the kernel AUTO runs depends on the sizes: no silent default
    # the backward reads hit records up to 2 queries per tet (include/deftet_hip.h, deftet_point_in_tet_bwd_f32): the autograd ops ask for them only there
    assert hip_ops.bwd_uses_records(257250, 100000) and hip_ops.bwd_uses_records(48000, 96000) and not hip_ops.bwd_uses_records(48000, 96001)


def test_tracked_query_boxes_alternate_between_two_buffers():
    """host logic of query_box="track" (no kernel involved): the first call of a (device, B, Q) measures and writes buffer 0, every later
    call reads what the previous one wrote and writes the other buffer; DEFTET_PIT_BOX=off disables it."""
    from deftet_amd import hip_ops
    hip_ops.clear_query_box_cache()
    dev = torch.device("cpu", 0)                                 # (the helper only allocates and hands out tensors)
    i0, o0, m0 = hip_ops._tracked_boxes(dev, 2, 100)
    i1, o1, m1 = hip_ops._tracked_boxes(dev, 2, 100)
    i2, o2, m2 = hip_ops._tracked_boxes(dev, 2, 100)
    assert i0 is None and o0.shape == (2, 6) and m0 is None     # a measuring call has no hint to miss
    assert i1 is o0 and o1 is not o0
    assert i2 is o1 and o2 is o0
    assert m1 is m2 and m1.dtype == torch.int32 and m1.numel() == 2 and m1.device.type == "cpu"   # the mailbox the kernels write
    j0, p0, _ = hip_ops._tracked_boxes(dev, 2, 101)              # another query count: its own state
    assert j0 is None and p0 is not o0 and p0 is not o1
    os.environ["DEFTET_PIT_BOX"] = "off"
    try:
        assert hip_ops._tracked_boxes(dev, 2, 100) == (None, None, None)
    finally:
        del os.environ["DEFTET_PIT_BOX"]
    hip_ops.clear_query_box_cache()


def test_tracked_query_boxes_fall_back_to_measuring_when_queries_miss_the_box():
    """host logic of the tracker's feedback loop: the kernels report the regular queries that fell outside the hinted box in a
    pinned int32 mailbox; more than `limit` of them => the next calls measure again, for a number of calls that quadruples with
    every relapse (two alternating query distributions end up measuring all the time) and shrinks after a long clean run."""
    from deftet_amd import hip_ops
    hip_ops.clear_query_box_cache()
    dev = torch.device("cpu", 0)
    B, Q = 3, 1 << 17

    def call():
        return hip_ops._tracked_boxes(dev, B, Q)

    def state():
        return hip_ops.query_box_trackers()[hip_ops.query_box_key(dev, B, Q)]

    assert call()[0] is None                                     # first call measures
    box_in, _, miss = call()
    assert box_in is not None and miss is not None
    limit = max(4, Q >> 14)
    miss[1] = limit                                              # a few stragglers: not worth a reaction
    assert call()[0] is not None and state()["backoffs"] == 0
    miss[1] = limit + 1                                          # what a tracked call whose box did not fit reports
    i, o, m = call()
    assert i is None and m is None and o is not None             # ... the next call measures
    assert int(miss.max()) == 0 and state()["backoffs"] == 1 and state()["backoff"] == 4
    i, o2, m = call()                                            # one measuring call (back-off 1), then tracking resumes from its box
    assert i is o and m is miss
    miss[0] = 1000                                               # relapse: four measuring calls
    for _ in range(4):
        assert call()[0] is None
    assert call()[0] is not None and state()["backoffs"] == 2 and state()["backoff"] == 16
    for k in range(2, 6):                                        # a caller that keeps missing: 16, 64, ... measuring calls per tracked one
        miss[2] = 77
        n = 0
        while call()[0] is None:
            n += 1
        assert n == 4 ** k
    tracked = state()["tracked"]
    assert state()["measured"] > 20 * tracked                    # i.e. it measures practically always
    # a long clean run earns the short back-off again
    b0 = state()["backoff"]
    for _ in range(64 * b0):
        assert call()[0] is not None
    assert state()["backoff"] == b0 // 4
    hip_ops.clear_query_box_cache()


@pytest.mark.parametrize("res", [2, 4, 20])
def test_kuhn_grid_counts_and_orientation(res):
    verts, tets = grids.kuhn_grid(res)
    assert tets.shape == (6 * (res // 2) ** 3, 4) and verts.shape == ((res // 2 + 1) ** 3, 3)
    assert tets.shape[0] == round(0.75 * res ** 3)
    pos = grids.jittered_positions(verts, res, 2)
    tp = grids.gather_tets(pos, tets)
    vol = grids.tet_orientation(tp)
    assert (vol > 0).all()
    # the six tets of every cube tile it: volumes add up to the cube volume
    assert np.isclose(grids.tet_orientation(grids.gather_tets((verts - 0.5)[None].astype(np.float32), tets)).sum() / 6, 1.0)
    # boundary vertices are not jittered
    on_bnd = ((verts == 0) | (verts == 1))
    assert np.array_equal(pos[0][on_bnd], (verts - 0.5).astype(np.float32)[on_bnd])
    q = grids.random_queries(2, 1000)
    assert q.dtype == np.float32 and q.min() >= -0.525 and q.max() < 0.525
    assert np.array_equal(q, grids.random_queries(2, 1000))       # seeded


def test_tet_file_roundtrip(tmp_path):
    verts, tets = grids.kuhn_grid(4)
    p = str(tmp_path / "cube_0.250000_tet.tet")
    grids.write_tet(p, verts, tets)
    assert open(p).readline().strip() == "tet %d %d" % (verts.shape[0], tets.shape[0])
    v2, t2 = grids.read_tet(p)
    assert np.allclose(v2, verts) and np.array_equal(t2, tets)


def test_shard_ranges():
    for n, w in [(64, 8), (8, 1), (10, 4), (3, 8), (0, 2)]:
        spans = [sharding.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, w)
    with pytest.raises(ValueError):
        sharding.shard_range(4, 4, 4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(n_items, rank, world)
        # every rank computes the per-shape "losses" of the shapes it owns
        local = torch.stack([torch.tensor([float(i), float(i) * 0.5 + 1.0]) for i in range(lo, hi)]) if hi > lo \
            else torch.zeros(0, 2)
        full = sharding.all_gather_losses(local, n_items)
        q.put((rank, full.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 5])
def test_all_gather_losses_gloo_world2(n_items):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([[float(i), float(i) * 0.5 + 1.0] for i in range(n_items)]).astype(np.float32)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want)


def _worker_pipelined(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.LossGather()
        outs = []
        for step in range(4):                                       # submit returns the PREVIOUS step's result
            local = torch.full((3,), float(10 * step + rank))
            outs.append(g.submit(local))
        outs.append(g.flush())
        assert g.flush() is None
        q.put((rank, [None if o is None else o.numpy() for o in outs]))
    finally:
        dist.destroy_process_group()


def test_loss_gather_pipelined_gloo_world2():
    """the asynchronous all-gather bench.py uses for N > 1: results arrive one step late, in order"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        outs = got[r]
        assert outs[0] is None and len(outs) == 5
        for step in range(4):
            want = np.concatenate([np.full(3, 10.0 * step + k, np.float32) for k in range(2)])
            assert np.array_equal(outs[step + 1], want)
    # without a process group the class degrades to a one-step delay line
    g = sharding.LossGather()
    assert g.submit(torch.ones(2)) is None and torch.equal(g.submit(torch.zeros(2)), torch.ones(2))


def test_read_tetrahedron_mirror(tmp_path):
    """N4: .tet text format + the boundary snapping rule of utils/dataloder_helper.py:30-69."""
    from deftet_amd.utils.dataloder_helper import read_tetrahedron, tet_file_name
    from deftet_amd import grids
    root = str(tmp_path)
    v, t, mask = read_tetrahedron(res=8, root=root)            # file missing -> synthetic grid is written
    fn, r = tet_file_name(8, root)
    assert os.path.exists(fn) and fn.endswith("cube_0.125000_tet.tet") and r == 0.125
    head = open(fn).readline().split()
    assert head == ["tet", str(v.shape[0]), str(t.shape[0])]
    v0, t0 = grids.kuhn_grid(8)
    assert np.array_equal(t, t0) and np.allclose(v, v0) and t.dtype == np.int64 and v.dtype == np.float64
    assert mask.shape == v.shape
    interior = np.logical_and(v0 > 0, v0 < 1)
    assert np.array_equal(mask, interior)
    # snapping: coordinates within res/4 of the faces of the unit cube move onto them
    w = v0.copy()
    w[w == 0] = 0.03
    w[w == 1] = 0.97
    grids.write_tet(fn, w, t0)
    v2, _, mask2 = read_tetrahedron(res=8, root=root)
    assert np.array_equal(v2, v0) and np.array_equal(mask2, interior)
    # second call reads the existing file; a malformed one raises
    open(fn, "w").write("tet 2 1\n0 0 0\n1 1 1\n0 1\n")
    with pytest.raises(ValueError):
        read_tetrahedron(res=8, root=root)
    with pytest.raises(FileNotFoundError):
        read_tetrahedron(res=10, root=root, generate_missing=False)


def test_read_tetrahedron_vs_reference_outputs_on_shipped_grid(tmp_path):
    """N4 pin: what the reference's read_tetrahedron (utils/dataloder_helper.py:30-69) returned for the shipped
    cube_40_tet.tet (tests/golden/n4_read_tetrahedron.npz, written by gen_golden.py) == what the mirror returns for
    the same grid written in the same text format."""
    import hashlib
    from deftet_amd import grids
    from deftet_amd.utils.dataloder_helper import read_tetrahedron, tet_file_name
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "n4_read_tetrahedron.npz"))
    g40 = np.load(os.path.join(os.path.dirname(__file__), "golden", "cube40_grid.npz"))
    fn, res = tet_file_name(40, str(tmp_path))
    assert res == float(gold["res"])
    os.makedirs(os.path.dirname(fn))
    with open(fn, "w") as f:                                       # 17 significant digits: the float64 values survive the text format
        f.write("tet %d %d\n" % (g40["verts"].shape[0], g40["tets"].shape[0]))
        np.savetxt(f, g40["verts"], fmt="%.17g")
        np.savetxt(f, g40["tets"], fmt="%d")
    v, t, mask = read_tetrahedron(res=40, root=str(tmp_path), generate_missing=False)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()
    assert list(gold["shape"]) == [v.shape[0], t.shape[0]]
    assert sha(v.astype(np.float64)) == gold["verts_sha"].tobytes()
    assert sha(t.astype(np.int64)) == gold["tets_sha"].tobytes()
    assert sha(mask.astype(np.uint8)) == gold["mask_sha"].tobytes()
    assert int(mask.sum()) == int(gold["n_interior"]) and int((v == 0).sum()) == int(gold["n_snapped_to_0"])
    assert np.array_equal(v[:16], gold["first_rows"])


def test_integration_overlay_names_exist():
    """every module of the overlay imports on a CPU-only host and exports the
    reference's names (the operators themselves refuse to run without a GPU)."""
    import importlib
    names = {
        "deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils": ["check_condition_f_base", "point_in_tet_bary", "paste_occ"],
        "deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils": ["tet_face_adj_m_f_idx"],
        "deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils": ["tet_analytic_distance_f_batch"],
        "deftet_amd.layers.nearest_neighbor": ["NearestNeighbor"],
        "deftet_amd.layers.DefTet.deftet": ["DefTet", "TetTopology"],
        "deftet_amd.utils.lib.tet_point_adj.interface": ["Tet_point_adj"],
        "deftet_amd.utils.lib.tet_face_adj.interface": ["Tet_face_adj"],
        "deftet_amd.utils.lib.tet_adj_share.interface": ["Tet_adj_share"],
        "deftet_amd.utils.lib.colaps_v.interface": ["Tet_point_adj"],
        "deftet_amd.utils.tet_utils": ["tet_to_face"],
        "deftet_amd.utils.dataloder_helper": ["read_tetrahedron"],
        "deftet_amd.render.deftet_sparse_render": ["deftet_sparse_render"],
        "deftet_amd.render.compositing": ["alpha_composite"],
        "deftet_amd.surface_losses": ["normal_consistency", "sample_on_faces", "cloud_to_cloud", "cloud_to_surface", "surface_terms"],
        "deftet_amd.render.prepare_for_wz": ["generate_edge", "generate_tet_edge_idx", "generate_subdivision", "generate_point_adj_idx",
                                             "delete_tet", "tetweights2tetneighbourweights"],
    }
    for mod, attrs in names.items():
        m = importlib.import_module(mod)
        for a in attrs:
            assert hasattr(m, a), "%s lacks %s" % (mod, a)
    from deftet_amd.layers.DefTet.deftet import DefTet
    for meth in ("check_tet_inside_sdfs", "gather_tet_pos", "get_boundary_index", "get_internal_index", "paste_occ", "volume_variance",
                 "amips_energy", "edge_length", "tet_inverse_v", "forward", "forward_surface_align", "laplacian_sparse"):
        assert callable(getattr(DefTet, meth))


def _clean_overlay_names():
    for k in [k for k in sys.modules if k.split(".")[0] in ("layers", "utils", "kaolin", "cv2")]:
        del sys.modules[k]


def test_overlay_installs_and_reference_shaped_callers_resolve_to_hip_ops():
    """INTEGRATION.md section A, executed: after overlay.install() the reference's own import statements
    (tests/ref_shaped_callers.py mirrors them) bind to this repository's operators, incl. the two Kaolin
    entry points; nothing is imported from a CPU fallback."""
    import inspect
    import deftet_amd.overlay as overlay
    saved = dict(sys.modules)
    _clean_overlay_names()
    try:
        names = overlay.install(kaolin=True)
        assert set(overlay.L1_MODULES) <= set(names) and "kaolin.ops.mesh" in names and "kaolin.render.mesh" in names
        from tests import ref_shaped_callers as C
        got = C.import_like_reference()
        import deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils as ours
        assert got["check_condition_f_base"] is ours.check_condition_f_base
        import deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils as ours_b
        import deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils as ours_c
        assert got["tet_face_adj_m_f_idx"] is ours_b.tet_face_adj_m_f_idx
        assert got["tet_analytic_distance_f_batch"] is ours_c.tet_analytic_distance_f_batch
        for k in ("NearestNeighbor", "Tet_point_adj", "Tet_face_adj", "Tet_adj_share"):
            assert got[k].__module__.startswith("deftet_amd."), k
        kal = got["kal"]
        assert inspect.signature(kal.ops.mesh.check_sign).parameters.keys() >= {"verts", "faces", "points", "hash_resolution"}
        sig = inspect.signature(kal.render.mesh.deftet_sparse_render)
        assert list(sig.parameters)[:5] == ["pixel_coords", "render_ranges", "face_vertices_z", "face_vertices_image", "face_features"]
        assert sig.parameters["knum"].default == 300 and sig.parameters["eps"].default == 1e-8
        # the operators refuse CPU tensors instead of falling back
        import torch
        from deftet_amd import _lib
        with pytest.raises(_lib.DefTetHipError):
            C.query_and_paste(torch.zeros(1, 2, 4, 3), torch.zeros(1, 3, 3), torch.zeros(1, 2))
        with pytest.raises(_lib.DefTetHipError):
            kal.ops.mesh.check_sign(torch.zeros(1, 4, 3), torch.zeros(2, 3, dtype=torch.long), torch.zeros(1, 5, 3))
        overlay.uninstall(names)
    finally:
        _clean_overlay_names()
        sys.modules.update({k: v for k, v in saved.items() if k not in sys.modules})


@pytest.mark.skipif(not os.path.isdir("/root/reference/layers"), reason="needs the reference checkout (authoring container only)")
def test_unchanged_reference_modules_import_through_the_overlay():
    """The reference's own layers/DefTet/deftet.py, utils/mesh_utils.py and utils/tet_utils.py — unmodified, from
    /root/reference — import with the overlay installed (no nvcc, no THC, no cv2, no kaolin, no run.so in the CWD)
    and end up holding this repository's operators."""
    import deftet_amd.overlay as overlay
    saved, saved_path = dict(sys.modules), list(sys.path)
    _clean_overlay_names()
    try:
        names = overlay.install(kaolin=True)
        sys.path.insert(0, "/root/reference")
        import importlib
        ref_deftet = importlib.import_module("layers.DefTet.deftet")
        ref_mesh = importlib.import_module("utils.mesh_utils")
        ref_tet = importlib.import_module("utils.tet_utils")
        assert ref_deftet.__file__.startswith("/root/reference/") and ref_tet.__file__.startswith("/root/reference/")
        import deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils as a
        import deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils as b
        import deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils as c
        import deftet_amd.layers.nearest_neighbor as d
        assert ref_deftet.check_condition_f_base is a.check_condition_f_base
        assert ref_mesh.tet_face_adj_m_f_idx is b.tet_face_adj_m_f_idx
        assert ref_mesh.tet_analytic_distance_f_batch is c.tet_analytic_distance_f_batch
        assert ref_mesh.NearestNeighbor is d.NearestNeighbor
        assert type(ref_tet.c_tet_point_adj).__module__ == "deftet_amd.utils.lib.tet_point_adj.interface"
        assert type(ref_tet.c_tet_face_adj).__module__ == "deftet_amd.utils.lib.tet_face_adj.interface"
        assert type(ref_tet.c_obj_tet_adj_share).__module__ == "deftet_amd.utils.lib.tet_adj_share.interface"
        assert getattr(ref_deftet.kal, "__deftet_amd_shim__", False)
        m = ref_deftet.DefTet()                                  # the reference's module object, our operators underneath
        assert hasattr(m, "forward_surface_align")
        overlay.uninstall(names)
    finally:
        sys.path[:] = saved_path
        _clean_overlay_names()
        sys.modules.update({k: v for k, v in saved.items() if k not in sys.modules})


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` with no torch.distributed environment must become the launcher the driver would
    otherwise be: torch.distributed.run, one node, N ranks, loopback rendezvous, the original arguments passed on."""
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    args = type("A", (), {"gpus": 4})()
    assert bench.self_launch(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # every BASELINE configuration has a workload definition (5 = the full geometry step, not a BASELINE entry)
    assert sorted(bench.CONFIGS) == [1, 2, 3, 4, 5] and bench.CONFIGS[2]["res"] == 70 and bench.CONFIGS[2]["n_query"] == 100_000


def test_profiles_readme_quotes_what_the_evidence_files_hold():
    """profiles/README.md's rows for the evidence pass (bench lines, kernel tables, pmc summaries) are generated from the files: a
    refreshed pass that is not followed by tools/profiles_readme_rows.py leaves stale figures, and this fails."""
    import subprocess
    for script in ("profiles_readme_rows.py", "profiles_readme_r06.py"):       # round 5's rows; the round-6 section
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--check"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, script + ": " + r.stdout + r.stderr


def test_order_rule_and_topology_keys_are_pure_functions():
    """host logic of order="auto" (round 6): a fixed rule on two far-step fractions, and what identifies a tet list."""
    from deftet_amd import hip_ops
    d = hip_ops._decide_order
    # the measured cases the rule was fitted on (profiles/r06_order_rule.jsonl, r06_order_breaks.jsonl)
    assert not d(0.0047, 0.0047)            # Kuhn grid, any enumeration
    assert not d(0.047, 0.027)              # the shipped QuarTet grid: the computed order does not even halve its far steps
    assert d(0.0244, 0.0047)                # 1 % of the positions shuffled: already 25 % slower as it is
    assert d(0.995, 0.0047)                 # a shuffled list
    assert not d(0.5, 0.4)                  # nothing to gain from a permutation that is incoherent itself
    # the watch compares what is traversed now with what the decision was made on
    assert not hip_ops._moved(0.0050, 0.0047) and not hip_ops._moved(0.052, 0.047)
    assert hip_ops._moved(0.99, 0.0047) and hip_ops._moved(0.0047, 0.99)
    k = hip_ops._topology_key
    assert k(None) is None and k(7) == 7 and k("grid") == "grid" and k((1, 2)) == (1, 2)

    class Topo:
        serial = 3
    assert k(Topo()) == ("topology", 3)
    a = torch.arange(48, dtype=torch.int64).reshape(12, 4)
    assert k(a) == k(a.clone()) and k(a) != k(a.flip(0).contiguous()) and k(a)[0] == "tensor"
    assert k(a.to(torch.int32)) == k(a)                                     # content, not dtype
    with pytest.raises(RuntimeError):
        k(object())


def test_sparse_render_default_policy_is_none():
    """the rasterizer's saturation policy is an open question (Kaolin absent): the default is NOT a silent choice"""
    import inspect
    from deftet_amd.render.deftet_sparse_render import deftet_sparse_render, NEAREST, FIRST
    assert inspect.signature(deftet_sparse_render).parameters["policy"].default is None
    assert (NEAREST, FIRST) == (0, 1)
