// tet_order.hip — a traversal order for the point-in-tet query that does not depend on how the caller numbers its tets.
//
// k_tet_scan_wave (point_in_tet.hip) stages the candidates of 64 CONSECUTIVE tets: it is fast when consecutive tets are
// spatial neighbours (a Kuhn grid enumerated cube by cube) and degrades to per-lane walks of the global cell table when they
// are not (a shuffled list; whatever order a mesher happened to emit — the grids the reference trains on are QuarTet outputs,
// utils/dataloder_helper.py:30-43).  The topology of a DefTet grid is static (layers/DefTet/deftet.py:65-68: one index list,
// gathered every step), so the remedy is computed ONCE per topology from any set of positions: a permutation that walks the
// tets column by column — (x, y) columns one mean tet extent wide, ascending z inside a column, which is the shape of the
// footprints the traversal groups lanes by (a few cells in x and y, any number of z slabs).  The traversal kernels read their
// tet through the permutation and publish the ORIGINAL index, so "lowest tet index" (check_condition_tet_for.cu:176-178) and
// every output are unchanged: the order is a matter of speed only.
//
// deftet_tet_spatial_order_f32 also measures how coherent both orders are — the number of places inside a 64-tet group where
// the column changes or z jumps — so that the caller can keep the identity when the list is already coherent (no indirection,
// coalesced 48-byte loads).
#pragma clang fp contract(off)
#include "prims.hpp"

#include "common.hpp"

namespace deftet {
namespace order {

constexpr int kStatBlocks = 128;
constexpr int kStatWords = 10;         // lo xyz, hi xyz of the finite centroids; sum of the box extents xyz; count
constexpr int kColBits = 10, kZBits = 10;
constexpr unsigned kBadKey = (1u << (2 * kColBits + kZBits)) - 1u;      // non-finite tets: behind everything else
constexpr int kZJump = 1 << (kZBits - 4);                                // a z step of more than 1/16 of the range breaks a run

__device__ __forceinline__ bool centroid_of(const float *__restrict__ tet, int t, float *c, float *ext)
{
    const float4 *src = reinterpret_cast<const float4 *>(tet + (size_t)t * 12);
    const float4 a = src[0], b = src[1], d = src[2];
    const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
        const float hi = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
        c[k] = ((v[k] + v[3 + k]) + (v[6 + k] + v[9 + k])) * 0.25f;
        ext[k] = hi - lo;
        ok = ok && fabsf(c[k]) <= 3.0e38f && ext[k] <= 3.0e38f;       // NaN fails both
    }
    return ok;
}

__global__ __launch_bounds__(256) void k_order_stats(const float *__restrict__ tet, int T, float *part)
{
    __shared__ float sh[4][kStatWords];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, se[3] = {0.f, 0.f, 0.f}, n = 0.f;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        float c[3], e[3];
        if (!centroid_of(tet, t, c, e)) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], c[k]);
            hi[k] = fmaxf(hi[k], c[k]);
            se[k] += e[k];
        }
        n += 1.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            se[k] += __shfl_xor(se[k], off);
        }
        n += __shfl_xor(n, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sh[w][k] = lo[k]; sh[w][3 + k] = hi[k]; sh[w][6 + k] = se[k]; }
        sh[w][9] = n;
    }
    __syncthreads();
    if (threadIdx.x < kStatWords) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, sh[i][k]) : k < 6 ? fmaxf(v, sh[i][k]) : v + sh[i][k];
        part[blockIdx.x * kStatWords + k] = v;
    }
}

// key = (x column, y column, z bin): columns one mean box extent wide (at most 2^kColBits of them per axis), z in 2^kZBits
// bins over the centroids' range
__global__ __launch_bounds__(256) void k_order_keys(const float *__restrict__ tet, int T, const float *__restrict__ part, unsigned *key)
{
    __shared__ float st[kStatWords];
    if (threadIdx.x < kStatWords) {
        const int k = threadIdx.x;
        float v = part[k];
        for (int i = 1; i < kStatBlocks; ++i) {
            const float p = part[i * kStatWords + k];
            v = k < 3 ? fminf(v, p) : k < 6 ? fmaxf(v, p) : v + p;
        }
        st[k] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float c[3], e[3];
    if (!centroid_of(tet, t, c, e)) {
        key[t] = kBadKey;
        return;
    }
    const float n = fmaxf(st[9], 1.f);
    unsigned q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float range = fmaxf(st[3 + k] - st[k], 0.f);
        const int bits = k < 2 ? kColBits : kZBits;
        const float cells = (float)((1 << bits) - 1);
        // columns: one mean extent wide, but never more than the key has room for; z: the finest the key holds
        const float width = k < 2 ? fmaxf(st[6 + k] / n, range / cells) : range / cells;
        const float f = width > 0.f ? (c[k] - st[k]) / width : 0.f;
        q[k] = (unsigned)__builtin_amdgcn_fmed3f(f, 0.f, cells);
    }
    key[t] = min((q[0] << (kColBits + kZBits)) | (q[1] << kZBits) | q[2], kBadKey - 1u);
}

// breaks[0] / breaks[1]: places inside a group of 64 consecutive tets of the NATIVE / the SORTED order where the column
// changes or z jumps by more than kZJump bins
__global__ __launch_bounds__(256) void k_order_breaks(const unsigned *__restrict__ key, const unsigned *__restrict__ skey, int T, int *breaks)
{
    __shared__ int sh[2][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b0 = 0, b1 = 0;
    if (i < T && (i & 63) != 0) {
        auto brk = [](unsigned a, unsigned b) {
            const int dz = (int)(a & ((1u << kZBits) - 1u)) - (int)(b & ((1u << kZBits) - 1u));
            return ((a >> kZBits) != (b >> kZBits) || dz > kZJump || dz < -kZJump) ? 1 : 0;
        };
        b0 = brk(key[i], key[i - 1]);
        b1 = brk(skey[i], skey[i - 1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        b0 += __shfl_xor(b0, off);
        b1 += __shfl_xor(b1, off);
    }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = b0; sh[1][threadIdx.x >> 6] = b1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int s = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
        if (s) atomicAdd(&breaks[threadIdx.x], s);
    }
}

struct Layout {
    float *part;
    unsigned *key, *skey;
    void *sortTmp;
    size_t sortBytes, bytes;
};

static Layout make_layout(int T, void *ws, size_t wsb)
{
    Layout L{};
    Arena A(ws, wsb);
    L.part = A.take<float>((size_t)kStatBlocks * kStatWords);
    L.key = A.take<unsigned>((size_t)T);
    L.skey = A.take<unsigned>((size_t)T);
    L.sortBytes = prims::radix_sort_temp_bytes<unsigned, unsigned>((size_t)T);
    L.sortTmp = A.take<char>(L.sortBytes);
    L.bytes = align_up(A.off, 256);
    return L;
}

}  // namespace order
}  // namespace deftet

using namespace deftet;

extern "C" size_t deftet_tet_spatial_order_workspace_bytes(int n_tet)
{
    if (n_tet <= 0) return 0;
    return order::make_layout(n_tet, nullptr, 0).bytes;
}

extern "C" int deftet_tet_spatial_order_f32(const float *tet, int n_tet, int32_t *order_out, int32_t *breaks, void *workspace,
                                            size_t workspace_bytes, void *stream_)
{
    using namespace deftet::order;
    DEFTET_CHECK_ARG(n_tet >= 0, "negative size");
    hipStream_t st = as_stream(stream_);
    if (breaks) DEFTET_HIP(hipMemsetAsync(breaks, 0, 8, st));
    if (n_tet == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(tet && order_out && ((uintptr_t)tet & 15) == 0, "null or misaligned pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    Layout L = make_layout(n_tet, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    const int nblk = (n_tet + 255) / 256;
    DEFTET_LAUNCH(k_order_stats, dim3(kStatBlocks), dim3(256), st, tet, n_tet, L.part);
    DEFTET_LAUNCH(k_order_keys, dim3(nblk), dim3(256), st, tet, n_tet, (const float *)L.part, L.key);
    // stable: tets with equal keys keep their original order
    int rc = prims::radix_sort_from<unsigned, unsigned>(prims::PtrLoad<unsigned>{L.key}, L.skey, prims::IotaLoad{},
                                                        reinterpret_cast<unsigned *>(order_out), (size_t)n_tet,
                                                        2 * kColBits + kZBits, L.sortTmp, L.sortBytes, st);
    if (rc != DEFTET_OK) return rc;
    if (breaks) DEFTET_LAUNCH(k_order_breaks, dim3(nblk), dim3(256), st, (const unsigned *)L.key, (const unsigned *)L.skey, n_tet, breaks);
    return DEFTET_OK;
}
