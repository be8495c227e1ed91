"""`read_tetrahedron` with the reference's name, signature and return values
(/root/reference/utils/dataloder_helper.py:30-69), without the QuarTet binary.

The reference shells out to `quartet/quartet meshes/cube.obj <res> ...` when
`<root>/quartet/meshes/cube_%f_tet.tet` is missing.  QuarTet is a third-party executable that is
not part of the reference tree; here the missing file is written from the synthetic Kuhn grid of
SURVEY.md section 8(d) (deftet_amd.grids.kuhn_grid: same size class, T = 0.75 R^3) in the same
text format, so `train_multigpu.py:60-66` style callers run unchanged:

    tet <n_vert> <n_tet>
    x y z            (n_vert lines)
    i j k l          (n_tet lines, 0-based)

Host-side file I/O, like the reference: no GPU involved.
"""
import os

import numpy as np

from deftet_amd import grids


def tet_file_name(res, root=".."):
    if res > 1.0:
        res = 1.0 / res                                      # :33-34
    return os.path.join(root, "quartet/meshes", "cube_%f_tet.tet" % res), res


def read_tetrahedron(res=50, root="..", generate_missing=True):
    file_name, res = tet_file_name(res, root)
    if not os.path.exists(file_name):                        # :40-43 runs QuarTet here
        if not generate_missing:
            raise FileNotFoundError(file_name)
        r = int(round(1.0 / res))
        verts, tets = grids.kuhn_grid(r + (r % 2))
        os.makedirs(os.path.dirname(file_name), exist_ok=True)
        grids.write_tet(file_name, verts, tets)
    vertices, tetrahedrons = grids.read_tet(file_name)      # :45-60 (asserts on malformed files -> ValueError)
    vertices = np.asarray(vertices, dtype=np.float64)
    vertices[vertices <= (0 + res / 4.0)] = 0                # :66  determine the boundary point
    vertices[vertices >= (1 - res / 4.0)] = 1                # :67
    mask = np.logical_and(vertices < 1, vertices > 0)        # :68
    return vertices, np.asarray(tetrahedrons), mask
