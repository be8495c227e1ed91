"""Front-to-back alpha compositing of the k depth-sorted layers the rasterizer returns per pixel —
the step the reference applies to `deftet_sparse_render`'s output
(/root/reference/diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:31-64, 102-113).
Standard emission-absorption quadrature: layer i contributes alpha_i * prod_{j<i} (1 - alpha_j)."""
import torch

ALPHA_EPS = 1e-10


def alpha_composite(layers_bxpxkxd, depth_bxpxkx1=None, background=1.0, far_depth=-6.0):
    """layers[..., 0] = opacity, layers[..., 1:] = colour; layers are ordered nearest first.
    Returns (colour [B,P,D-1] over a `background`-coloured backdrop, coverage [B,P,1],
    expected depth [B,P,1] or None)."""
    alpha = layers_bxpxkxd[..., :1].clamp(ALPHA_EPS, 1.0 - ALPHA_EPS)
    through = torch.cumprod(1.0 - alpha, dim=2)
    reach = torch.cat([torch.ones_like(through[:, :, :1]), through[:, :, :-1]], dim=2)   # transmittance BEFORE each layer
    weight = alpha * reach
    coverage = weight.sum(2)
    colour = (weight * layers_bxpxkxd[..., 1:]).sum(2) + background * (1.0 - coverage)
    depth = None
    if depth_bxpxkx1 is not None:
        depth = (weight * depth_bxpxkx1).sum(2) + far_depth * (1.0 - coverage)
    return colour, coverage, depth
