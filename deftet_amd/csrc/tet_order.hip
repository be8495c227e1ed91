// tet_order.hip — a traversal order for the point-in-tet query that does not depend on how the caller numbers its tets.
//
// k_tet_scan_wave (point_in_tet.hip) stages the candidates of 64 CONSECUTIVE tets: it is fast when consecutive tets are
// spatial neighbours (a Kuhn grid enumerated cube by cube) and degrades to per-lane walks of the global cell table when they
// are not (a shuffled list; whatever order a mesher happened to emit — the grids the reference trains on are QuarTet outputs,
// utils/dataloder_helper.py:30-43).  The topology of a DefTet grid is static (layers/DefTet/deftet.py:65-68: one index list,
// gathered every step), so the remedy is computed ONCE per topology from any set of positions: a permutation that walks the
// tets column by column — (x, y) columns about one mean tet extent wide, ascending z inside a column, which is the shape of the
// footprints the traversal groups lanes by (a few cells in x and y, any number of z slabs).  The column boundaries are taken
// from the DATA: the lower box corners of a lattice-like grid (the reference's grids, QuarTet or Kuhn) cluster around the
// lattice planes, and a boundary that cuts through a cluster splits every column along it into two half-empty ones (measured:
// 145 us with fixed-width columns against 100 with the clusters kept whole, configs[2] with a shuffled tet list); so columns
// end at the empty stretches of a 1,024-bin histogram of the corners, and only where there is none at their nominal width.
// The traversal kernels read their tet through the permutation and publish the ORIGINAL index, so "lowest tet index"
// (check_condition_tet_for.cu:176-178) and every output are unchanged: the order is a matter of speed only.
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
constexpr int kFine = 1024;            // histogram bins per axis the column boundaries are chosen on
constexpr int kStatWords = 10;         // lo xyz, hi xyz of the lower box corners of the finite tets; sum of the box extents xyz; count
constexpr int kColBits = 10, kZBits = 10;
constexpr unsigned kBadKey = (1u << (2 * kColBits + kZBits)) - 1u;      // non-finite tets: behind everything else
constexpr int kZJump = 1 << (kZBits - 2);                                // a z step of more than a quarter of the range breaks a run

__device__ __forceinline__ bool centroid_of(const float *__restrict__ tet, int t, float *c, float *ext, float *blo = nullptr)
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
        if (blo) blo[k] = lo;
        ok = ok && fabsf(c[k]) <= 3.0e38f && ext[k] <= 3.0e38f;       // NaN fails both
    }
    return ok;
}

__global__ __launch_bounds__(256) void k_order_stats(const float *__restrict__ tet, int T, float *part)
{
    __shared__ float sh[4][kStatWords];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, se[3] = {0.f, 0.f, 0.f}, n = 0.f;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        float c[3], e[3], m[3];
        if (!centroid_of(tet, t, c, e, m)) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], m[k]);                                 // range of the lower box corners
            hi[k] = fmaxf(hi[k], m[k]);
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

// statistics of the whole list from the per-block partials (every thread of a block gets them)
__device__ __forceinline__ void reduce_stats(const float *__restrict__ part, float *st /* shared, kStatWords */)
{
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
}

__device__ __forceinline__ int fine_bin(float x, float lo, float hi)
{
    const float range = hi - lo;
    const float f = range > 0.f ? (x - lo) / range * (float)(kFine - 1) : 0.f;
    return (int)__builtin_amdgcn_fmed3f(f, 0.f, (float)(kFine - 1));
}

// histograms of the lower box corners along x and y, kFine bins over their range
__global__ __launch_bounds__(256) void k_order_hist(const float *__restrict__ tet, int T, const float *__restrict__ part, unsigned *hist)
{
    __shared__ float st[kStatWords];
    __shared__ unsigned h[2][kFine];
    reduce_stats(part, st);
    for (int i = threadIdx.x; i < 2 * kFine; i += blockDim.x) (&h[0][0])[i] = 0u;
    __syncthreads();
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        float c[3], e[3], m[3];
        if (!centroid_of(tet, t, c, e, m)) continue;
        atomicAdd(&h[0][fine_bin(m[0], st[0], st[3])], 1u);
        atomicAdd(&h[1][fine_bin(m[1], st[1], st[4])], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kFine; i += blockDim.x) {
        const unsigned v = (&h[0][0])[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// Column boundaries from the histograms, one thread per axis (a walk over 1,024 bins, once per topology): a column starts at
// a non-empty bin and runs until an empty stretch of at least a tenth of the nominal width (the clusters of a lattice stay
// whole) or, without one, until it is one nominal width (the mean box extent of the axis) wide.  lut[axis][bin] = column.
__global__ __launch_bounds__(64) void k_order_lut(const float *__restrict__ part, const unsigned *__restrict__ hist, unsigned short *lut)
{
    __shared__ float st[kStatWords];
    reduce_stats(part, st);
    const int a = threadIdx.x;
    if (a >= 2) return;
    const float n = fmaxf(st[9], 1.f), range = fmaxf(st[3 + a] - st[a], 0.f);
    const float binw = range / (float)(kFine - 1);
    // nominal column width in bins: the mean extent, at least range / (2^kColBits - 1) (the key has room for that many columns)
    const float wnom = fmaxf(st[6 + a] / n, range / (float)((1 << kColBits) - 1));
    const int wb = binw > 0.f ? max(1, (int)(wnom / binw + 0.5f)) : kFine;
    const int gap = max(1, wb / 10);
    int col = 0, start = -1, empty = 0;
    for (int i = 0; i < kFine; ++i) {
        const bool occ = hist[a * kFine + i] != 0u;
        if (start < 0) {
            if (occ) { start = i; empty = 0; }
        } else if (!occ) {
            ++empty;
        } else {
            if (empty >= gap || i - start >= wb) {                      // a new column begins at this bin
                col = min(col + 1, (1 << kColBits) - 1);
                start = i;
            }
            empty = 0;
        }
        lut[a * kFine + i] = (unsigned short)col;                      // (empty bins carry the column before them: never looked up)
    }
}

// key = (x column, y column, z bin): columns from the look-up tables, z in 2^kZBits bins over the range of the lower corners
__global__ __launch_bounds__(256) void k_order_keys(const float *__restrict__ tet, int T, const float *__restrict__ part,
                                                    const unsigned short *__restrict__ lut, unsigned *key)
{
    __shared__ float st[kStatWords];
    reduce_stats(part, st);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float c[3], e[3], m[3];
    if (!centroid_of(tet, t, c, e, m)) {
        key[t] = kBadKey;
        return;
    }
    const unsigned qx = lut[fine_bin(m[0], st[0], st[3])], qy = lut[kFine + fine_bin(m[1], st[1], st[4])];
    const float rz = st[5] - st[2], cells = (float)((1 << kZBits) - 1);
    const unsigned qz = (unsigned)__builtin_amdgcn_fmed3f(rz > 0.f ? (m[2] - st[2]) / rz * cells : 0.f, 0.f, cells);
    key[t] = min((qx << (kColBits + kZBits)) | (qy << kZBits) | qz, kBadKey - 1u);
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

// How coherent a tet numbering is FOR THE TRAVERSAL: far[0] = places inside a group of 64 consecutive tets (of the caller's
// order, or of `order` when given) where the next tet lies FAR from the one before — the centroids differ by more than
// kFarExt mean box extents along some axis — and far[1] = the number of places looked at.  (The `breaks` of
// deftet_tet_spatial_order_f32 count every change of column; the shipped QuarTet grid alternates between NEIGHBOURING columns
// all the time — 76 % breaks — and is still the faster order as it is: what the wave kernel cares about is whether the 64 tets
// of a wave share a neighbourhood, not whether they share a column.)  Non-finite tets count as far from everything.
constexpr float kFarExt = 3.0f;
constexpr int kFarBlocks = 256;

__global__ __launch_bounds__(256) void k_order_far(const float *__restrict__ tet, int T, const int *__restrict__ order,
                                                   const float *__restrict__ part, int *blockFar)
{
    __shared__ float st[kStatWords];
    __shared__ int sh[4];
    reduce_stats(part, st);
    const float n = fmaxf(st[9], 1.f);
    const float lim[3] = {kFarExt * st[6] / n, kFarExt * st[7] / n, kFarExt * st[8] / n};
    int far = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) {
        if ((i & 63) == 0) continue;
        float c0[3], c1[3], e[3];
        const bool ok0 = centroid_of(tet, order ? order[i - 1] : i - 1, c0, e), ok1 = centroid_of(tet, order ? order[i] : i, c1, e);
        bool f = !(ok0 && ok1);
#pragma unroll
        for (int k = 0; k < 3; ++k) f = f || !(fabsf(c1[k] - c0[k]) <= lim[k]);
        far += f ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) far += __shfl_xor(far, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = far;
    __syncthreads();
    if (threadIdx.x == 0) blockFar[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// one workgroup adds the partials up and stores the two words with plain stores (`out` may be host-mapped memory)
__global__ __launch_bounds__(kFarBlocks) void k_order_far_sum(const int *__restrict__ blockFar, int T, int *out)
{
    __shared__ int sh[kFarBlocks / 64];
    int v = blockFar[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < kFarBlocks / 64; ++k) tot += sh[k];
        out[0] = tot;
        out[1] = T - (T + 63) / 64;
    }
}

struct Layout {
    float *part;
    unsigned *hist;
    unsigned short *lut;
    unsigned *key, *skey;
    void *sortTmp;
    size_t sortBytes, bytes;
};

static Layout make_layout(int T, void *ws, size_t wsb)
{
    Layout L{};
    Arena A(ws, wsb);
    L.part = A.take<float>((size_t)kStatBlocks * kStatWords);
    L.hist = A.take<unsigned>((size_t)2 * kFine);
    L.lut = A.take<unsigned short>((size_t)2 * kFine);
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
    DEFTET_HIP(hipMemsetAsync(L.hist, 0, (size_t)2 * kFine * 4, st));
    DEFTET_LAUNCH(k_order_hist, dim3(kStatBlocks), dim3(256), st, tet, n_tet, (const float *)L.part, L.hist);
    DEFTET_LAUNCH(k_order_lut, dim3(1), dim3(64), st, (const float *)L.part, (const unsigned *)L.hist, L.lut);
    DEFTET_LAUNCH(k_order_keys, dim3(nblk), dim3(256), st, tet, n_tet, (const float *)L.part, (const unsigned short *)L.lut, L.key);
    // stable: tets with equal keys keep their original order
    int rc = prims::radix_sort_from<unsigned, unsigned>(prims::PtrLoad<unsigned>{L.key}, L.skey, prims::IotaLoad{},
                                                        reinterpret_cast<unsigned *>(order_out), (size_t)n_tet,
                                                        2 * kColBits + kZBits, L.sortTmp, L.sortBytes, st);
    if (rc != DEFTET_OK) return rc;
    if (breaks) DEFTET_LAUNCH(k_order_breaks, dim3(nblk), dim3(256), st, (const unsigned *)L.key, (const unsigned *)L.skey, n_tet, breaks);
    return DEFTET_OK;
}

// Coherence of a tet numbering for the traversal (k_order_far above): out2[0] = far steps inside groups of 64 consecutive tets
// of `order` (NULL: the caller's own numbering), out2[1] = steps looked at.  out2 is written with two plain 4-byte stores by
// one thread and may be host-mapped memory (a caller can poll it without synchronising).  Three small launches.
extern "C" size_t deftet_tet_order_coherence_workspace_bytes(int n_tet)
{
    if (n_tet <= 0) return 0;
    return align_up((size_t)deftet::order::kStatBlocks * deftet::order::kStatWords * 4, 256) + align_up((size_t)deftet::order::kFarBlocks * 4, 256);
}

extern "C" int deftet_tet_order_coherence_f32(const float *tet, int n_tet, const int32_t *order, int32_t *out2, void *workspace,
                                              size_t workspace_bytes, void *stream_)
{
    using namespace deftet::order;
    DEFTET_CHECK_ARG(n_tet >= 0 && out2, "negative size or null output");
    hipStream_t st = as_stream(stream_);
    if (n_tet == 0) {
        DEFTET_HIP(hipMemsetAsync(out2, 0, 8, st));
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(tet && ((uintptr_t)tet & 15) == 0, "null or misaligned pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && workspace_bytes >= deftet_tet_order_coherence_workspace_bytes(n_tet),
                     "workspace null, misaligned or too small");
    float *part = static_cast<float *>(workspace);
    int *blockFar = reinterpret_cast<int *>(static_cast<char *>(workspace) + align_up((size_t)kStatBlocks * kStatWords * 4, 256));
    DEFTET_LAUNCH(k_order_stats, dim3(kStatBlocks), dim3(256), st, tet, n_tet, part);
    DEFTET_LAUNCH(k_order_far, dim3(kFarBlocks), dim3(256), st, tet, n_tet, (const int *)order, (const float *)part, blockFar);
    DEFTET_LAUNCH(k_order_far_sum, dim3(1), dim3(kFarBlocks), st, (const int *)blockFar, n_tet, out2);
    return DEFTET_OK;
}
