from .deftet_sparse_render import deftet_sparse_render, peel2mask, vertex2face, perspective  # noqa: F401
