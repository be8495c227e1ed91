from .deftet_sparse_render import deftet_sparse_render  # noqa: F401
from .compositing import alpha_composite  # noqa: F401
from .camera import perspective, face_attributes, render_mesh_color  # noqa: F401
