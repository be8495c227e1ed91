// reduce.hip — per-shape loss scalars: out[r] = sum_c a[r,c] * b[r,c]  (b may be NULL: plain row sum).
// Deterministic (fixed reduction tree, no atomics): one 1024-thread workgroup per row.
#include "common.hpp"

namespace deftet {
namespace red {

__global__ __launch_bounds__(1024) void k_rowdot(const float *__restrict__ a, const float *__restrict__ b, float *out,
                                                 long long n_cols)
{
    __shared__ float wsum[16];
    const long long r = blockIdx.x;
    const float *pa = a + r * n_cols;
    const float *pb = b ? b + r * n_cols : nullptr;
    float acc = 0.f;
    const bool vec = (n_cols % 4 == 0) && (((uintptr_t)pa & 15) == 0) && (!pb || ((uintptr_t)pb & 15) == 0);
    if (vec) {
        const long long n4 = n_cols / 4;
        for (long long i = threadIdx.x; i < n4; i += 1024) {
            const float4 x = reinterpret_cast<const float4 *>(pa)[i];
            if (pb) {
                const float4 y = reinterpret_cast<const float4 *>(pb)[i];
                acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
            } else {
                acc += (x.x + x.y) + (x.z + x.w);
            }
        }
    } else {
        for (long long i = threadIdx.x; i < n_cols; i += 1024) acc += pb ? pa[i] * pb[i] : pa[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? wsum[threadIdx.x] : 0.f;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0) out[r] = v;
    }
}

}  // namespace red
}  // namespace deftet

extern "C" int deftet_rowdot_f32(const float *a, const float *b, float *out, int n_rows, long long n_cols, void *stream_)
{
    DEFTET_CHECK_ARG(n_rows >= 0 && n_cols >= 0, "negative size");
    if (n_rows == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(a && out, "null pointer");
    DEFTET_LAUNCH(deftet::red::k_rowdot, dim3(n_rows), dim3(1024), deftet::as_stream(stream_), a, b, out, n_cols);
    return DEFTET_OK;
}
