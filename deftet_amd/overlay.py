"""INTEGRATION.md section A as code: make an UNCHANGED DefTet checkout import this repository's
HIP-backed operators.

    import deftet_amd.overlay as overlay
    overlay.install()            # before `import layers`, `import utils.tet_utils`, `import kaolin`
    import train_multigpu        # the reference's own scripts, unmodified

`install()` registers, under the reference's module names, the drop-in modules of this package for
every import the hot path goes through:

    layers.DefTet.check_condition_tetrahedron_base.utils   check_condition_f_base
    layers.DefTet.tet_face_adj_m_idx.utils                 tet_face_adj_m_f_idx
    layers.DefTet.tet_analytic_distance_batch.utils        tet_analytic_distance_f_batch
    layers.nearest_neighbor                                NearestNeighbor
    utils.lib.{tet_point_adj,tet_face_adj,tet_adj_share,colaps_v}.interface
    kaolin.ops.mesh.check_sign, kaolin.render.mesh.deftet_sparse_render   (only when Kaolin itself is not
                                                           importable, or with kaolin=True; parity unpinned)
    cv2                                                    empty stub (imported, never used: check_condition.../utils.py:14)

With `deftet_module=True` also `layers.DefTet.deftet` (the `DefTet` nn.Module built on the fused
operators) — otherwise the reference's own module runs on top of the replaced L1 operators.
Nothing here touches a CPU fallback: every replaced entry point raises on non-GPU tensors.
"""
import importlib
import importlib.util
import sys
import types

L1_MODULES = {
    "layers.DefTet.check_condition_tetrahedron_base.utils": "deftet_amd.layers.DefTet.check_condition_tetrahedron_base.utils",
    "layers.DefTet.tet_face_adj_m_idx.utils": "deftet_amd.layers.DefTet.tet_face_adj_m_idx.utils",
    "layers.DefTet.tet_analytic_distance_batch.utils": "deftet_amd.layers.DefTet.tet_analytic_distance_batch.utils",
    "layers.nearest_neighbor": "deftet_amd.layers.nearest_neighbor",
    "utils.lib.tet_point_adj.interface": "deftet_amd.utils.lib.tet_point_adj.interface",
    "utils.lib.tet_face_adj.interface": "deftet_amd.utils.lib.tet_face_adj.interface",
    "utils.lib.tet_adj_share.interface": "deftet_amd.utils.lib.tet_adj_share.interface",
    "utils.lib.colaps_v.interface": "deftet_amd.utils.lib.colaps_v.interface",
}


def _kaolin_check_sign(verts, faces, points, hash_resolution=512):
    """kal.ops.mesh.check_sign(verts [B,V,3], faces [F,3], points [B,N,3]) -> bool [B,N]
    (`hash_resolution` only steers Kaolin's own acceleration structure)."""
    from deftet_amd import hip_ops
    return hip_ops.check_sign(verts, faces, points)


def kaolin_shim():
    """A module tree exposing exactly the two Kaolin entry points the hot path calls
    (layers/DefTet/deftet.py:46, diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:97-100)."""
    from deftet_amd.render.deftet_sparse_render import deftet_sparse_render
    kal = types.ModuleType("kaolin")
    kal.__path__ = []                                  # a package, so `import kaolin.ops.mesh` resolves through sys.modules
    ops, render = types.ModuleType("kaolin.ops"), types.ModuleType("kaolin.render")
    ops.__path__, render.__path__ = [], []
    ops_mesh, render_mesh = types.ModuleType("kaolin.ops.mesh"), types.ModuleType("kaolin.render.mesh")
    ops_mesh.check_sign = _kaolin_check_sign
    render_mesh.deftet_sparse_render = deftet_sparse_render
    kal.ops, kal.render, ops.mesh, render.mesh = ops, render, ops_mesh, render_mesh
    kal.__deftet_amd_shim__ = True
    return {"kaolin": kal, "kaolin.ops": ops, "kaolin.render": render, "kaolin.ops.mesh": ops_mesh,
            "kaolin.render.mesh": render_mesh}


def install(kaolin=None, deftet_module=False, stub_cv2=True):
    """Register the overlay in sys.modules; returns the list of names it registered.
    kaolin: True = always shim, False = never, None = shim only if `import kaolin` would fail."""
    done = []
    for ref_name, ours in L1_MODULES.items():
        sys.modules[ref_name] = importlib.import_module(ours)
        done.append(ref_name)
    if deftet_module:
        sys.modules["layers.DefTet.deftet"] = importlib.import_module("deftet_amd.layers.DefTet.deftet")
        done.append("layers.DefTet.deftet")
    if kaolin is None:
        kaolin = "kaolin" not in sys.modules and importlib.util.find_spec("kaolin") is None
    if kaolin:
        for name, mod in kaolin_shim().items():
            sys.modules[name] = mod
            done.append(name)
    if stub_cv2 and "cv2" not in sys.modules and importlib.util.find_spec("cv2") is None:
        sys.modules["cv2"] = types.ModuleType("cv2")
        done.append("cv2")
    return done


def uninstall(names):
    for n in names:
        sys.modules.pop(n, None)
