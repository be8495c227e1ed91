// reduce.hip — per-shape loss scalars: out[r] = sum_c a[r,c] * b[r,c]  (b may be NULL: plain row sum),
// optionally plus a second term sum_c a2[r,c] * b2[r,c] with its own width (one launch pair for both).
// Deterministic (fixed reduction tree): kParts workgroups per row produce partial sums.  Up to kTicketRows rows take ONE
// launch: every workgroup hands its partial to the memory side (returning RMW atomics bypass the per-XCD L2s) and draws
// a ticket; the one that draws the last ticket of its row adds the partials up in index order.  Larger row counts (and
// processes that use more than kTicketSlots streams) use the two-launch form (partials into the caller's workspace, one
// wave per row adds them up).

#include <stdlib.h>

#include "common.hpp"

namespace deftet {
namespace red {

constexpr int kParts = 128;

// SQRT: the terms are sqrt(a + eps) instead (b unused) — the surface terms' "mean of sqrt(d^2 + 1e-10)" in one launch
template <bool SQRT = false>
__device__ __forceinline__ float rowdot_slice(const float *__restrict__ a, const float *__restrict__ b, long long r,
                                              long long n_cols, float eps = 0.f)
{
    const float *pa = a + r * n_cols;
    const float *pb = b ? b + r * n_cols : nullptr;
    float acc = 0.f;
    const bool vec = (n_cols % 4 == 0) && (((uintptr_t)pa & 15) == 0) && (!pb || ((uintptr_t)pb & 15) == 0);
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    if (vec) {
        const long long n4 = n_cols / 4;
        // four strided items per trip, all loads issued before the first is used (one load per trip left the row's
        // three trips as three memory latencies in a row); the items are added in index order, trip by trip.  (Batching
        // the tail trips as well — clamped indices, items past the end not added — measured slower: 16 -> 21 us at
        // 8 rows x 100,000 x (4 + 1) columns, where most threads have three items.)
        long long i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            float4 x[4], y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = reinterpret_cast<const float4 *>(pa)[i + k * stride];
            if (!SQRT && pb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = reinterpret_cast<const float4 *>(pb)[i + k * stride];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (SQRT) acc += (sqrtf(x[k].x + eps) + sqrtf(x[k].y + eps)) + (sqrtf(x[k].z + eps) + sqrtf(x[k].w + eps));
                else if (pb) acc += (x[k].x * y[k].x + x[k].y * y[k].y) + (x[k].z * y[k].z + x[k].w * y[k].w);
                else acc += (x[k].x + x[k].y) + (x[k].z + x[k].w);
            }
        }
        for (; i < n4; i += stride) {
            const float4 x = reinterpret_cast<const float4 *>(pa)[i];
            if (SQRT) {
                acc += (sqrtf(x.x + eps) + sqrtf(x.y + eps)) + (sqrtf(x.z + eps) + sqrtf(x.w + eps));
            } else if (pb) {
                const float4 y = reinterpret_cast<const float4 *>(pb)[i];
                acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
            } else {
                acc += (x.x + x.y) + (x.z + x.w);
            }
        }
    } else {
        for (long long i = tid; i < n_cols; i += stride) acc += SQRT ? sqrtf(pa[i] + eps) : pb ? pa[i] * pb[i] : pa[i];
    }
    return acc;
}

template <bool SQRT>
__global__ __launch_bounds__(256) void k_rowdot_partial(const float *__restrict__ a, const float *__restrict__ b,
                                                        float *part, long long n_cols, const float *__restrict__ a2,
                                                        const float *__restrict__ b2, long long n_cols2, float eps)
{
    __shared__ float wsum[4];
    const long long r = blockIdx.y;
    float acc = rowdot_slice<SQRT>(a, b, r, n_cols, eps);
    if (!SQRT && a2) acc += rowdot_slice(a2, b2, r, n_cols2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[r * kParts + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (blockIdx.x == 0 && threadIdx.x < kParts && threadIdx.x >= gridDim.x) part[r * kParts + threadIdx.x] = 0.f;   // k_rowdot_final adds all kParts
}

constexpr int kTicketRows = 1024, kTicketSlots = 32;
// zero at load; every launch leaves its slot zero again.  A slot belongs to one (device, stream) pair
// (ticket_slot_for_stream): launches that share it are ordered by their stream.
__device__ int g_tickets[kTicketSlots][kTicketRows];

template <bool SQRT>
__global__ __launch_bounds__(256) void k_rowdot_fused(const float *__restrict__ a, const float *__restrict__ b, float *part,
                                                      long long n_cols, const float *__restrict__ a2, const float *__restrict__ b2,
                                                      long long n_cols2, float *out, int slot, float eps)
{
    __shared__ float wsum[4];
    __shared__ float vals[kParts];
    __shared__ int s_last;
    const long long r = blockIdx.y;
    float acc = rowdot_slice<SQRT>(a, b, r, n_cols, eps);
    if (!SQRT && a2) acc += rowdot_slice(a2, b2, r, n_cols2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        const int old = atomicExch(reinterpret_cast<int *>(part + r * kParts + blockIdx.x), __float_as_int(v));
        asm volatile("" ::"v"(old));                               // returning form: complete (at the memory side) once it has returned
        __builtin_amdgcn_s_waitcnt(0);
        s_last = atomicAdd(&g_tickets[slot][r], 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < kParts)                                      // (fewer workgroups than kParts: the missing partials are zeros)
        vals[threadIdx.x] = threadIdx.x < gridDim.x ? __int_as_float(atomicOr(reinterpret_cast<int *>(part + r * kParts + threadIdx.x), 0)) : 0.f;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = vals[threadIdx.x] + vals[64 + threadIdx.x];      // same tree as k_rowdot_final
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0) {
            out[r] = v;
            atomicExch(&g_tickets[slot][r], 0);
        }
    }
}

__global__ __launch_bounds__(64) void k_rowdot_final(const float *__restrict__ part, float *out)
{
    const long long r = blockIdx.x;
    float v = part[r * kParts + threadIdx.x] + part[r * kParts + 64 + threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (threadIdx.x == 0) out[r] = v;
}

// backward of out[r] = sum_c sqrt(x[r,c] + eps):  gx[r,c] = g[r] * 0.5 / sqrt(x[r,c] + eps)
__global__ __launch_bounds__(256) void k_sqrt_rowsum_bwd(const float *__restrict__ x, const float *__restrict__ g, float *gx,
                                                         long long n_cols, float eps)
{
    const long long r = blockIdx.y;
    const float gr = g[r] * 0.5f;
    const float *px = x + r * n_cols;
    float *po = gx + r * n_cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_cols; i += (long long)gridDim.x * 256) po[i] = gr / sqrtf(px[i] + eps);
}

}  // namespace red
}  // namespace deftet

// Workgroups per row: all kParts.  (Measured at 8 rows x 4 MB: 16 / 32 / 64 / 128 workgroups per row take 35 / 22 / 18 / 17 us —
// the streams, not the 128 same-address ticket draws per row, are what the launch waits for.  DEFTET_ROWDOT_PARTS: experiments.)
static int parts_for(long long n_cols, long long n_cols2)
{
    static const int forced = [] { const char *e = getenv("DEFTET_ROWDOT_PARTS"); return e ? atoi(e) : 0; }();
    (void)n_cols; (void)n_cols2;
    return forced >= 1 && forced <= deftet::red::kParts ? forced : deftet::red::kParts;
}

extern "C" size_t deftet_rowdot_workspace_bytes(int n_rows) { return (size_t)(n_rows > 0 ? n_rows : 0) * deftet::red::kParts * 4; }

extern "C" int deftet_rowdot2_f32(const float *a, const float *b, long long n_cols, const float *a2, const float *b2,
                                  long long n_cols2, float *out, int n_rows, void *workspace, size_t workspace_bytes,
                                  void *stream_)
{
    DEFTET_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && n_cols2 >= 0 && n_rows <= 65535, "bad size");
    if (n_rows == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(a && out, "null pointer");
    DEFTET_CHECK_ARG(workspace && workspace_bytes >= deftet_rowdot_workspace_bytes(n_rows), "workspace null or too small");
    hipStream_t st = deftet::as_stream(stream_);
    float *part = static_cast<float *>(workspace);
    const int slot = n_rows <= deftet::red::kTicketRows ? deftet::ticket_slot_for_stream(st, deftet::red::kTicketSlots) : -1;
    if (slot >= 0) {
        DEFTET_LAUNCH(deftet::red::k_rowdot_fused<false>, dim3(parts_for(n_cols, n_cols2), n_rows), dim3(256), st, a, b, part, n_cols, a2, b2, n_cols2,
                      out, slot, 0.f);
        return DEFTET_OK;
    }
    DEFTET_LAUNCH(deftet::red::k_rowdot_partial<false>, dim3(parts_for(n_cols, n_cols2), n_rows), dim3(256), st, a, b, part, n_cols, a2, b2, n_cols2, 0.f);
    DEFTET_LAUNCH(deftet::red::k_rowdot_final, dim3(n_rows), dim3(64), st, part, out);
    return DEFTET_OK;
}

extern "C" int deftet_rowdot_f32(const float *a, const float *b, float *out, int n_rows, long long n_cols, void *workspace,
                                 size_t workspace_bytes, void *stream_)
{
    return deftet_rowdot2_f32(a, b, n_cols, nullptr, nullptr, 0, out, n_rows, workspace, workspace_bytes, stream_);
}

// out[r] = sum_c sqrt(x[r,c] + eps), one launch for up to 1,024 rows (the ticket form of the row sums above); its backward
// gx[r,c] = g[r] / (2 sqrt(x[r,c] + eps)).  What the surface terms of utils/mesh_utils.py:14 ("sqrt(d^2 + 1e-10)", then
// the mean over the points) reduce to per shape.
extern "C" int deftet_sqrt_rowsum_f32(const float *x, float eps, float *out, int n_rows, long long n_cols, void *workspace,
                                      size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && n_rows <= 65535, "bad size");
    if (n_rows == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(x && out, "null pointer");
    DEFTET_CHECK_ARG(workspace && workspace_bytes >= deftet_rowdot_workspace_bytes(n_rows), "workspace null or too small");
    hipStream_t st = deftet::as_stream(stream_);
    float *part = static_cast<float *>(workspace);
    const int slot = n_rows <= deftet::red::kTicketRows ? deftet::ticket_slot_for_stream(st, deftet::red::kTicketSlots) : -1;
    if (slot >= 0) {
        DEFTET_LAUNCH(deftet::red::k_rowdot_fused<true>, dim3(parts_for(n_cols, 0), n_rows), dim3(256), st, x, (const float *)nullptr, part, n_cols,
                      (const float *)nullptr, (const float *)nullptr, 0LL, out, slot, eps);
        return DEFTET_OK;
    }
    DEFTET_LAUNCH(deftet::red::k_rowdot_partial<true>, dim3(parts_for(n_cols, 0), n_rows), dim3(256), st, x, (const float *)nullptr, part, n_cols,
                  (const float *)nullptr, (const float *)nullptr, 0LL, eps);
    DEFTET_LAUNCH(deftet::red::k_rowdot_final, dim3(n_rows), dim3(64), st, part, out);
    return DEFTET_OK;
}

extern "C" int deftet_sqrt_rowsum_bwd_f32(const float *x, float eps, const float *grad_out, float *grad_x, int n_rows,
                                          long long n_cols, void *stream_)
{
    DEFTET_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && n_rows <= 65535, "bad size");
    if (n_rows == 0 || n_cols == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(x && grad_out && grad_x, "null pointer");
    const long long blocks = (n_cols + 1023) / 1024;
    DEFTET_LAUNCH(deftet::red::k_sqrt_rowsum_bwd, dim3((unsigned)(blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks), n_rows), dim3(256),
                  deftet::as_stream(stream_), x, grad_out, grad_x, n_cols, eps);
    return DEFTET_OK;
}

// A few host integers (per-shape counts, offsets) to device memory WITHOUT a host-to-device copy: the values travel in the
// kernel's argument block.  A copy from pageable host memory — what torch.tensor(list, device=...) issues — blocks the host
// until the stream has drained; two of them per geometry step kept the Python thread waiting for the GPU and the GPU waiting
// for the Python thread (tools/probes/geometry_cpu_probe.py).  Asynchronous, and legal inside a graph capture.
namespace deftet {
namespace red {
constexpr int kHostInts = 64;
struct HostInts {
    long long v[kHostInts];
};
__global__ void k_put_ints(HostInts h, int n, int *o32, long long *o64, float *of)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    if (o32) o32[i] = (int)h.v[i];
    if (o64) o64[i] = h.v[i];
    if (of) of[i] = (float)h.v[i];
}
}  // namespace red
}  // namespace deftet

extern "C" int deftet_put_host_ints(const long long *host_values, int n, int32_t *out_i32, long long *out_i64, float *out_f32, void *stream_)
{
    DEFTET_CHECK_ARG(n >= 0, "negative count");
    if (n == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(host_values && (out_i32 || out_i64 || out_f32), "null pointer");
    for (int at = 0; at < n; at += deftet::red::kHostInts) {
        deftet::red::HostInts h;
        const int m = n - at < deftet::red::kHostInts ? n - at : deftet::red::kHostInts;
        for (int i = 0; i < m; ++i) h.v[i] = host_values[at + i];
        for (int i = m; i < deftet::red::kHostInts; ++i) h.v[i] = 0;
        DEFTET_LAUNCH(deftet::red::k_put_ints, dim3(1), dim3(deftet::red::kHostInts), deftet::as_stream(stream_), h, m, out_i32 ? out_i32 + at : nullptr,
                      out_i64 ? out_i64 + at : nullptr, out_f32 ? out_f32 + at : nullptr);
    }
    return DEFTET_OK;
}
