// stubs.cpp — entry points declared in include/deftet_hip.h whose kernels are not written yet.
// Each returns DEFTET_EINVAL with a message; the file shrinks as the real implementations land.
#include "common.hpp"
#define NYI(name) return deftet::set_error(DEFTET_EINVAL, name ": not implemented yet")
extern "C" {
size_t deftet_sparse_render_workspace_bytes(int, int, int, int) { return 0; }
int deftet_sparse_render_fwd_f32(const float *, const float *, const float *, const float *, const float *, float *, int64_t *, float *, int, int, int, int, int, float, void *, size_t, void *) { NYI("deftet_sparse_render_fwd_f32"); }
int deftet_sparse_render_bwd_f32(const float *, const float *, const float *, const int64_t *, const float *, const float *, float *, float *, int, int, int, int, int, float, void *) { NYI("deftet_sparse_render_bwd_f32"); }
}
