#!/usr/bin/env python
"""Generates tests/golden/render_glue.npz by IMPORTING the reference's render-side glue in the authoring container
(inputs and outputs only; no reference source is stored).

    python tests/golden/gen_render_glue.py          # needs /root/reference

What is pinned (all pure torch on CPU; `kaolin` and `config` are stub modules, the reference never calls into Kaolin
for these functions except through `kal.render.mesh.deftet_sparse_render`, which the stub replaces by a recorder):
  peel2mask        diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:31-64, with and without depth layers
  vertex2face      diff_render/diftet_6_subdiv/4_render/vertex2face.py:12-28
  perspective      diff_render/diftet_6_subdiv/3_model/cameraop.py:19-33
  rendermeshcolor  5_rendereq/deftetrneder.py:67-113 around a rasterizer stub that RECORDS its five arguments and returns
                   seeded layers: pins how the call site prepares the rasterizer's inputs (z = last coordinate of the
                   camera-space corners, [B,F,3,2] image corners, [B,F,3,D] features after the sigmoid) and what it does
                   with the result (depth unwrap, compositing).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/diff_render/diftet_6_subdiv"


def main():
    rec = {}

    def fake_render(xy, xydep, z_bxfx3, img_bxfx3x2, feat_bxfx3xd):
        rec["args"] = [t.detach().clone() for t in (xy, xydep, z_bxfx3, img_bxfx3x2, feat_bxfx3xd)]
        g = torch.Generator().manual_seed(99)
        B, P, K, D = z_bxfx3.shape[0], xy.shape[1], 6, feat_bxfx3xd.shape[-1]
        layers = torch.rand(B, P, K, D, generator=g)
        rec["layers"] = layers.clone()
        return layers, torch.zeros(B, P, K, dtype=torch.long)

    kal = types.ModuleType("kaolin")
    kal.render = types.SimpleNamespace(mesh=types.SimpleNamespace(deftet_sparse_render=fake_render))
    cfg = types.ModuleType("config")
    cfg.rootdir = REF
    sys.modules["kaolin"], sys.modules["config"] = kal, cfg
    for sub in ("5_rendereq", "4_render", "3_model"):
        sys.path.insert(0, os.path.join(REF, sub))
    import deftetrneder as R
    from vertex2face import vertex2face
    from cameraop import perspective

    g = torch.Generator().manual_seed(7)
    out = {}
    # peel2mask
    ims = torch.rand(2, 40, 6, 4, generator=g)
    ims[0, :5, :, 0] = 0.0                                   # opacities at and beyond the clamp
    ims[0, 5:9, :, 0] = 1.0
    dep = -torch.rand(2, 40, 6, 1, generator=g) * 4
    c, v, d = R.peel2mask(ims, dep)
    c2, v2, d2 = R.peel2mask(ims)
    assert d2 is None
    out.update(peel_ims=ims.numpy(), peel_depth=dep.numpy(), peel_color=c.numpy(), peel_vis=v.numpy(), peel_dep=d.numpy(),
               peel_color_nodepth=c2.numpy(), peel_vis_nodepth=v2.numpy())
    # vertex2face
    vf = torch.rand(3, 11, 5, generator=g)
    faces = torch.randint(0, 11, (17, 3), generator=g)
    out.update(v2f_features=vf.numpy(), v2f_faces=faces.numpy(), v2f_out=vertex2face(vf, faces).numpy())
    # perspective
    pts = torch.randn(2, 13, 3, generator=g)
    rot = torch.linalg.qr(torch.randn(2, 3, 3, generator=g))[0]
    pos = torch.randn(2, 3, generator=g) * 3
    proj = torch.tensor([[1.3], [1.7], [-1.0]])
    cam3d, xy = perspective(pts, (rot, pos, proj))
    out.update(persp_points=pts.numpy(), persp_rot=rot.numpy(), persp_pos=pos.numpy(), persp_proj=proj.numpy(),
               persp_cam=cam3d.numpy(), persp_xy=xy.numpy())
    # rendermeshcolor, with and without the depth channel
    B, Pv, F, npx = 2, 9, 12, 10
    p3 = torch.randn(B, Pv, 3, generator=g)
    p2 = torch.randn(B, Pv, 2, generator=g)
    feat = torch.randn(B, Pv, 5, generator=g)                 # [depth, opacity, r, g, b] before the sigmoid
    fc = torch.randint(0, Pv, (F, 3), generator=g)
    pix = torch.rand(1, npx, 2, generator=g)
    rngs = torch.tensor([[[-10.0, 10.0]]]).expand(1, npx, 2).contiguous()
    for tag, depth in (("d", True), ("n", False)):
        f_in = feat if depth else feat[:, :, 1:]
        col, msk, dp = R.rendermeshcolor(pix, rngs, p3, p2, f_in, fc, viewdir=False, depth=depth)
        a = rec["args"]
        out.update({"rmc_%s_arg_z" % tag: a[2].numpy(), "rmc_%s_arg_img" % tag: a[3].numpy(), "rmc_%s_arg_feat" % tag: a[4].numpy(),
                    "rmc_%s_layers" % tag: rec["layers"].numpy(), "rmc_%s_color" % tag: col.numpy(), "rmc_%s_mask" % tag: msk.numpy()})
        if depth:
            out["rmc_d_depth"] = dp.numpy()
        else:
            assert dp is None
    out.update(rmc_points3d=p3.numpy(), rmc_points2d=p2.numpy(), rmc_feat=feat.numpy(), rmc_faces=fc.numpy(), rmc_pix=pix.numpy(),
               rmc_ranges=rngs.numpy())
    np.savez_compressed(os.path.join(HERE, "render_glue.npz"), **out)
    print("wrote", os.path.join(HERE, "render_glue.npz"), os.path.getsize(os.path.join(HERE, "render_glue.npz")), "bytes")


if __name__ == "__main__":
    main()
