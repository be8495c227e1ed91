// stubs.cpp — entry points declared in include/deftet_hip.h whose kernels are not written yet.
// Each returns DEFTET_EINVAL with a message; the file shrinks as the real implementations land.
#include "common.hpp"
#define NYI(name) return deftet::set_error(DEFTET_EINVAL, name ": not implemented yet")
extern "C" {
size_t deftet_face_edge_adj_workspace_bytes(int) { return 0; }
int deftet_face_edge_adj_f32(const float *, float *, int, int, void *, size_t, void *) { NYI("deftet_face_edge_adj_f32"); }
int deftet_tri_dist_fwd_f32(const float *, const float *, const float *, float *, float *, int, int, int, void *) { NYI("deftet_tri_dist_fwd_f32"); }
int deftet_tri_dist_bwd_f32(const float *, const float *, const float *, const float *, float *, int, int, int, int, void *) { NYI("deftet_tri_dist_bwd_f32"); }
int deftet_nn_index_f32(const float *, const float *, int32_t *, int, int, int, void *) { NYI("deftet_nn_index_f32"); }
size_t deftet_sparse_render_workspace_bytes(int, int, int, int) { return 0; }
int deftet_sparse_render_fwd_f32(const float *, const float *, const float *, const float *, const float *, float *, int64_t *, float *, int, int, int, int, int, float, void *, size_t, void *) { NYI("deftet_sparse_render_fwd_f32"); }
int deftet_sparse_render_bwd_f32(const float *, const float *, const float *, const int64_t *, const float *, const float *, float *, float *, int, int, int, int, int, float, void *) { NYI("deftet_sparse_render_bwd_f32"); }
}
