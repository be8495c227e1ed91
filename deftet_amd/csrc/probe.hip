// probe.hip — what this GPU's HBM delivers to a plain streaming kernel: the yardstick bench.py quotes beside the 8 TB/s
// datasheet figure (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy).  Not part of any operator.
#include "common.hpp"

namespace deftet {
namespace probe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every workgroup streams one contiguous piece of 256 * 16 * U bytes; consecutive threads touch consecutive 16-byte words,
// U independent loads per thread are in flight before the first store.  Nontemporal: every byte is touched once.
template <int U, bool COPY>
__global__ __launch_bounds__(256) void k_stream(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, float *sums)
{
    const size_t base = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    f32x4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(src + base + (size_t)k * 256);
    if (COPY) {
#pragma unroll
        for (int k = 0; k < U; ++k) __builtin_nontemporal_store(v[k], dst + base + (size_t)k * 256);
    } else {
        f32x4 s = v[0];
#pragma unroll
        for (int k = 1; k < U; ++k) s += v[k];
        const float t = (s.x + s.y) + (s.z + s.w);
        if (t == 1.2345678e30f) sums[blockIdx.x] = t;              // keeps the loads alive; practically never true
    }
}

}  // namespace probe
}  // namespace deftet

// mode 0: copy src -> dst; mode 1: read src only (dst: >= one float per 32 KB of src, may be written).  n_bytes is rounded
// DOWN to a multiple of 32 KB; returns the number of bytes actually streamed per direction in *done (may be NULL).
extern "C" int deftet_bandwidth_probe(const void *src, void *dst, size_t n_bytes, int mode, size_t *done, void *stream_)
{
    using namespace deftet;
    using namespace deftet::probe;
    DEFTET_CHECK_ARG(src && dst && (mode == 0 || mode == 1), "bad argument");
    DEFTET_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pointers must be 16-byte aligned");
    constexpr int U = 8;
    const size_t piece = (size_t)256 * 16 * U, blocks = n_bytes / piece;
    if (done) *done = blocks * piece;
    if (blocks == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(blocks <= 0x7FFFFFFFull, "too large for one launch");
    hipStream_t st = as_stream(stream_);
    if (mode == 0)
        DEFTET_LAUNCH((k_stream<U, true>), dim3((unsigned)blocks), dim3(256), st, (const f32x4 *)src, (f32x4 *)dst, (float *)nullptr);
    else
        DEFTET_LAUNCH((k_stream<U, false>), dim3((unsigned)blocks), dim3(256), st, (const f32x4 *)src, (f32x4 *)nullptr, (float *)dst);
    return DEFTET_OK;
}
