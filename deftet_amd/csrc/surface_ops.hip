// surface_ops.hip — the per-shape surface operators layers.DefTet.forward calls
//  for CDNA4 / gfx950.
//
//   A8  deftet_face_edge_adj_f32   layers/DefTet/tet_face_adj_m_idx/tet_face_adj_m_for.cu:15-130
//   A9  deftet_tri_dist_fwd/bwd    layers/DefTet/tet_analytic_distance_batch/tet_analytic_distance_for.cu:15-334
//                                  layers/DefTet/tet_analytic_distance_batch/tet_analytic_distance_back.cu:15-715
//   A10 deftet_nn_index_f32        layers/nearest_neighbor/nearest_neighbor_cuda.cu:15-80
//
// All three reference kernels are "one thread per query, loop over every primitive from
// global memory".  Here the primitive stream is wave-uniform: it is read once per wave through
// the scalar cache (s_load) and broadcast as SGPR operands, one query per lane, so the vector
// memory pipe only carries the queries and the results.  The arithmetic follows the reference
// operation by operation in fp32 with FMA contraction off (the argmin / neighbour indices
// are decided by exact fp32 comparisons), double-typed literals promoted as C++ does.
#pragma clang fp contract(off)
#include <cstdlib>
#include <cstring>

#include "common.hpp"

#include "prims.hpp"

namespace deftet {
namespace surf {

// ---------------------------------------------------------------------------- A10 nearest neighbour
// result = index of the first point with the strictly smallest ((dx*dx + dy*dy) + dz*dz)
__global__ __launch_bounds__(256) void k_nn(const float *__restrict__ queries, const float *__restrict__ points, int N,
                                            int M, int *result)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < N;
    const float *qq = queries + ((size_t)b * N + (live ? q : 0)) * 3;
    const float qx = qq[0], qy = qq[1], qz = qq[2];
    const float *__restrict__ pp = points + (size_t)b * M * 3;      // wave-uniform stream -> s_load
    float best = 1e20f;                                             // nearest_neighbor_cuda.cu:28
    int besti = 0;
    int i = 0;
    for (; i + 4 <= M; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dx = pp[(i + k) * 3] - qx, dy = pp[(i + k) * 3 + 1] - qy, dz = pp[(i + k) * 3 + 2] - qz;
            float d = 0.f;
            d += dx * dx;                                           // :42
            d += dy * dy;                                           // :44
            d += dz * dz;                                           // :46
            if (d < best) { best = d; besti = i + k; }              // :48-51
        }
    }
    for (; i < M; ++i) {
        const float dx = pp[i * 3] - qx, dy = pp[i * 3 + 1] - qy, dz = pp[i * 3 + 2] - qz;
        float d = 0.f;
        d += dx * dx;
        d += dy * dy;
        d += dz * dz;
        if (d < best) { best = d; besti = i; }
    }
    if (live) result[(size_t)b * N + q] = besti;
}

// --- batching across shapes ----------------------------------------------------------------------------
// The grid-accelerated operators keep every per-shape scratch array in a workspace SLICE of `slice` bytes; slice s starts
// s * slice bytes after slice 0.  A batched launch covers kBatchShapes shapes with its y (or z) grid dimension: the
// kernels get the pointers of slice 0 and rebase them with SHAPE(ptr), their inputs / outputs with their own strides.
// The per-shape element counts travel by value.  (One launch sequence for a batch instead of one per shape: at ~40
// launches per shape and operator the command processor was what bounded a training step's surface terms.)
constexpr int kBatchShapes = 8;
struct ShapeCounts { int n[kBatchShapes]; };
#define SHAPE(ptr) (ptr) = reinterpret_cast<decltype(ptr)>(reinterpret_cast<uintptr_t>(ptr) + (size_t)sb * slice)

// exclusive prefix sums of n ints per shape, one launch: workgroup (j, shape) scans tile j (kScanThreads * kScanPer
// elements) after adding up everything before its tile itself — the arrays are a few hundred thousand ints, so the
// redundant reads (L2 hits) cost less than a carry chain through one workgroup (94 us per call in round 2) or a second
// launch.  out[n] receives the total when with_total.
constexpr int kScanThreads = 1024, kScanPer = 16, kScanTile = kScanThreads * kScanPer;
static inline unsigned scan_tiles(long long n) { return (unsigned)((n + kScanTile - 1) / kScanTile > 0 ? (n + kScanTile - 1) / kScanTile : 1); }
__global__ __launch_bounds__(kScanThreads) void k_scan_excl(const int *in, int *out, int n, size_t slice, int with_total)
{
    __shared__ int wsum[kScanThreads / 64];
    const int sb = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    SHAPE(in); SHAPE(out);
    const int base = blockIdx.x * kScanTile;
    // carry: the sum of in[0, base) (base is a multiple of the tile, the arrays are 256-byte aligned)
    int carry = 0;
    for (int i = tid * 4; i < base; i += kScanThreads * 4) {
        const int4 q = *reinterpret_cast<const int4 *>(in + i);
        carry += (q.x + q.y) + (q.z + q.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) carry += __shfl_xor(carry, off);
    if (lane == 0) wsum[w] = carry;
    __syncthreads();
    carry = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) carry += wsum[k];
    __syncthreads();
    int v[kScanPer], sum = 0;
    const int i0 = base + tid * kScanPer;
    if (i0 + kScanPer <= n) {                                       // 16-byte loads
#pragma unroll
        for (int k = 0; k < kScanPer; k += 4) {
            const int4 q = *reinterpret_cast<const int4 *>(in + i0 + k);
            v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) sum += v[k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int run = carry + incl - sum, tot = 0;
#pragma unroll
    for (int k = 0; k < kScanThreads / 64; ++k) { if (k < w) run += wsum[k]; tot += wsum[k]; }
    if (i0 + kScanPer <= n) {
#pragma unroll
        for (int k = 0; k < kScanPer; k += 4) {
            int4 q;
            q.x = run; q.y = q.x + v[k]; q.z = q.y + v[k + 1]; q.w = q.z + v[k + 2];
            run = q.w + v[k + 3];
            *reinterpret_cast<int4 *>(out + i0 + k) = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPer; ++k) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    }
    if (with_total && tid == 0 && blockIdx.x == gridDim.x - 1) out[n] = carry + tot;
}

// rank = atomicAdd(&counter[key], 1) for every lane with key >= 0, with ONE atomic per run of consecutive lanes that share
// a key: point clouds sampled face by face put ~20 consecutive points into the same cell, and returning atomics on a
// handful of addresses were what the binning kernels spent their time on (k_tri_point_keys 76 us, k_nn_bin 42 us per
// 8 x 97 k points).  Must be called by all lanes of the wave (key < 0: no count).
__device__ __forceinline__ int run_atomic_rank(int *counter, int key)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const bool head = key >= 0 && (lane == 0 || prev != key);
    const unsigned long long heads = __ballot(head || key < 0);     // a lane without a key also ends the run before it
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes 0..lane
    const int start = 63 - __clzll((long long)(heads & upto));
    const unsigned long long above = heads & ~upto;
    const int end = above ? __ffsll((long long)above) - 1 : 64;
    int base = 0;
    if (head) base = atomicAdd(&counter[key], end - start);
    base = __shfl(base, start);
    return key >= 0 ? base + (lane - start) : 0;
}

// --- A10, grid-accelerated (exact) -------------------------------------------------------------------
// Points are counting-sorted into a uniform G^3 grid over their bounding box; a query walks the
// cell shells around its own (virtual) cell in increasing Chebyshev radius r and stops as soon as
// the best distance found is smaller than a certified lower bound for everything outside the
// shell box.  Distances are evaluated exactly as the reference does, and candidates are combined
// lexicographically (distance, index), which equals "first strict minimum of an ascending scan".
constexpr int kNNBlocks = 64;

constexpr int kNNCoarse = 4;           // coarse cells are 4x4x4 fine cells
struct NNGrid { float o[3], inv[3], cs[3], slack[3]; int G, Gc; };

// (the same launch clears what the later kernels accumulate into: cell counters, coarse-cell representatives, nRep)
__global__ __launch_bounds__(256) void k_nn_bbox(const float *__restrict__ pts, int M, float *part, size_t slice, int *cells, int *rep,
                                                 int *nRep, int nc)
{
    __shared__ float sh[4][6];
    const int sb = blockIdx.y;
    pts += (size_t)sb * M * 3;
    SHAPE(part); SHAPE(cells); SHAPE(rep); SHAPE(nRep);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) { cells[i] = 0; rep[i] = -1; }
    if (blockIdx.x == 0 && threadIdx.x < 4) nRep[threadIdx.x] = 0;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        if (fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY) {     // finite (NaN fails)
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) { sh[w][k] = lo[k]; sh[w][3 + k] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, sh[i][k]) : fmaxf(v, sh[i][k]);
        part[blockIdx.x * 6 + k] = v;
    }
}

__global__ __launch_bounds__(64) void k_nn_grid(const float *__restrict__ part, int G, NNGrid *g, size_t slice)
{
    const int sb = blockIdx.x;
    SHAPE(part); SHAPE(g);
    const int lane = threadIdx.x;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = part[lane * 6 + k]; hi[k] = part[lane * 6 + 3 + k]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    if (lane == 0) {
        NNGrid r;
        r.G = G;
        r.Gc = (G + kNNCoarse - 1) / kNNCoarse;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool ok = hi[k] >= lo[k];
            const float l = ok ? lo[k] : 0.f, h = ok ? hi[k] : 0.f;
            const float ext = h - l;
            r.o[k] = l;
            const bool flat = !(ext > 1e-30f) || !(ext < 1e30f);
            r.inv[k] = flat ? 0.f : (float)G / ext;
            r.cs[k] = flat ? INFINITY : ext / (float)G;          // cell size (inf: one slab, no bound on this axis)
            // absolute uncertainty of a cell boundary position as seen through the fp32 cell assignment
            r.slack[k] = 8e-6f * (fabsf(l) + fabsf(h)) + 1e-30f;
        }
        *g = r;
    }
}

__device__ __forceinline__ int nn_cell(float x, float o, float inv, int G)
{
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.f), (float)(G - 1));
    return (int)f;
}

__global__ __launch_bounds__(256) void k_nn_bin(const float *__restrict__ pts, int M, const NNGrid *__restrict__ gp, int *cells,
                                                int2 *pcell, int *rep, size_t slice)
{
    const int sb = blockIdx.y;
    pts += (size_t)sb * M * 3;
    SHAPE(gp); SHAPE(cells); SHAPE(pcell); SHAPE(rep);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < M;                                       // (no early return: run_atomic_rank is a wave operation)
    const NNGrid g = *gp;
    const int ii = live ? i : 0;
    const float x = pts[ii * 3], y = pts[ii * 3 + 1], z = pts[ii * 3 + 2];
    int2 r = make_int2(-1, 0);
    int cx = 0, cy = 0, cz = 0;
    if (live && fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY) {
        cx = nn_cell(x, g.o[0], g.inv[0], g.G); cy = nn_cell(y, g.o[1], g.inv[1], g.G); cz = nn_cell(z, g.o[2], g.inv[2], g.G);
        r.x = (cz * g.G + cy) * g.G + cx;
    }
    r.y = run_atomic_rank(cells, r.x);
    // any point of the coarse cell will do as its representative: the first arrival of every fine cell offers itself
    // (one atomic per occupied fine cell — one per POINT put hundreds of them on the same address: 0.33 ms per 8 shapes)
    if (r.x >= 0 && r.y == 0) atomicMax(&rep[((cz / kNNCoarse) * g.Gc + cy / kNNCoarse) * g.Gc + cx / kNNCoarse], i);
    if (live) pcell[i] = r;                                        // non-finite points can never be nearest (d is inf/NaN)
}

__global__ __launch_bounds__(256) void k_nn_scatter(const float *__restrict__ pts, int M, const int2 *__restrict__ pcell,
                                                    const int *__restrict__ start, float4 *sorted, size_t slice)
{
    const int sb = blockIdx.y;
    pts += (size_t)sb * M * 3;
    SHAPE(pcell); SHAPE(start); SHAPE(sorted);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int2 r = pcell[i];
    if (r.x < 0) return;
    sorted[start[r.x] + r.y] = make_float4(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], __int_as_float(i));
}

// Sort key of the queries k_nn_query answered itself: the first power of two above every Morton key of the grid (three
// interleaved coordinates 0..G+1), so that the far queries sort to the front on 3*bits + 1 key bits (16 for G <= 30).
static int nn_far_key_bits(int G)
{
    int bits = 1;
    while ((1 << bits) < G + 2) ++bits;
    return 3 * bits;
}
#ifndef NN_FAR_ROWS
#define NN_FAR_ROWS 25
#endif
constexpr int kFarRows = NN_FAR_ROWS;

__device__ __forceinline__ unsigned spread3(unsigned v)            // bit i of an 8-bit value -> bit 3i
{
    v = (v | (v << 8)) & 0x0000F00Fu;
    v = (v | (v << 4)) & 0x000C30C3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}

// Two phases per query (one lane each):
//  1. an UPPER bound U on the nearest distance from the coarse grid: shells of coarse cells are
//     searched until one holds a representative point; U = smallest exact distance to those;
//  2. every fine-grid row (cz,cy) whose slab can intersect the ball of radius sqrt(U) is visited
//     once: the x-interval of the ball in that row is one contiguous slice of the sorted points.
// All points with fp32 distance <= U lie in the enumerated slices (margins below), so the
// lexicographic (distance, index) minimum over them equals the reference's ascending scan.
__global__ __launch_bounds__(256) void k_nn_query(const float *__restrict__ queries, int N, const NNGrid *__restrict__ gp,
                                                  const int *__restrict__ start, const float4 *__restrict__ sorted,
                                                  const int *__restrict__ rep, const float *__restrict__ pts, int M, int *result,
                                                  unsigned *farKey, unsigned notFar, size_t slice, ShapeCounts cnt, int Nst,
                                                  unsigned shapeShift)
{
    // N = the query stride; shape sb has cnt.n[sb] <= N queries.  farKey is ONE array of Nst keys per shape (not in the
    // slices: the far keys of all shapes are sorted by one call, with the shape index above the key bits).
    const int sb = blockIdx.y;
    queries += (size_t)sb * N * 3; pts += (size_t)sb * M * 3; result += (size_t)sb * N; farKey += (size_t)sb * Nst;
    SHAPE(gp); SHAPE(start); SHAPE(sorted); SHAPE(rep);
    const unsigned shapeBits = (unsigned)sb << shapeShift;
    notFar |= shapeBits;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt.n[sb]) {
        if (q < Nst) farKey[q] = notFar;                            // padding keys sort behind the shape's far queries
        return;
    }
    const NNGrid g = *gp;
    const int G = g.G, Gc = g.Gc;
    const float qx = queries[q * 3], qy = queries[q * 3 + 1], qz = queries[q * 3 + 2];
    const float qq[3] = {qx, qy, qz};
    farKey[q] = notFar;
    if (!(fabsf(qx) < INFINITY && fabsf(qy) < INFINITY && fabsf(qz) < INFINITY)) { result[q] = 0; return; }   // every d is inf/NaN
    // a far query is keyed by the Morton code of its (clamped) fine cell: sorted by it, the lanes of k_nn_far are neighbours
    auto far_key = [&]() {
        unsigned key = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float f = floorf((qq[k] - g.o[k]) * g.inv[k]);
            f = fminf(fmaxf(f, -1.f), (float)G) + 1.f;              // 0 .. G+1 <= 161: eight bits
            key |= spread3((unsigned)f) << k;
        }
        return key;
    };
    auto dist = [&](float px, float py, float pz) {
        const float dx = px - qx, dy = py - qy, dz = pz - qz;
        float d = 0.f;
        d += dx * dx;                                               // nearest_neighbor_cuda.cu:42
        d += dy * dy;                                               // :44
        d += dz * dz;                                               // :46
        return d;
    };
    // ---- phase 0: tight upper bound from the 3x3x3 fine cells around the query (near queries)
    float U = INFINITY;
    {
        int c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float f = floorf((qq[k] - g.o[k]) * g.inv[k]);
            f = fminf(fmaxf(f, -2.f), (float)(G + 1));
            c[k] = (int)f;
        }
        const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, G - 1);
        if (x0 <= x1)
            for (int cz = max(c[2] - 1, 0); cz <= min(c[2] + 1, G - 1); ++cz)
                for (int cy = max(c[1] - 1, 0); cy <= min(c[1] + 1, G - 1); ++cy) {
                    const int row = (cz * G + cy) * G;
                    const int s = start[row + x0], e = start[row + x1 + 1];
                    for (int j = s; j < e; j += 4) {
                        float4 p[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) p[k] = sorted[min(j + k, e - 1)];
#pragma unroll
                        for (int k = 0; k < 4; ++k) U = fminf(U, dist(p[k].x, p[k].y, p[k].z));
                    }
                }
    }
    // ---- phase 1: otherwise a loose bound from the coarse grid (one representative point per 4^3 cells)
    int cc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float f = floorf((qq[k] - g.o[k]) * g.inv[k] * (1.0f / kNNCoarse));
        f = fminf(fmaxf(f, -1.f), (float)Gc);
        cc[k] = (int)f;
    }
    constexpr int kCoarseRings = 1;                                  // beyond that the query is "far": streaming scan
    for (int r = 0; r <= kCoarseRings && !(U < INFINITY); ++r) {
        for (int dz = -r; dz <= r; ++dz) {
            const int z = cc[2] + dz;
            if (z < 0 || z >= Gc) continue;
            for (int dy = -r; dy <= r; ++dy) {
                const int y = cc[1] + dy;
                if (y < 0 || y >= Gc) continue;
                const bool face = max(abs(dz), abs(dy)) == r;
                for (int dx = -r; dx <= r; dx += (face ? 1 : (r > 0 ? 2 * r : 1))) {
                    const int x = cc[0] + dx;
                    if (x < 0 || x >= Gc) continue;
                    const int i = rep[(z * Gc + y) * Gc + x];
                    if (i >= 0) U = fminf(U, dist(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]));
                }
            }
        }
    }
    float best = 1e20f;                                             // :28
    int besti = 0;
    if (!(U < INFINITY)) {                                           // nothing within one coarse ring: far (or no finite point)
        farKey[q] = far_key() | shapeBits;
        return;
    }
    U = fminf(U, 1e20f);                                            // nothing farther than the initial best can win
    // ---- phase 2: rows intersecting the ball of radius R (inflated for fp32 rounding of d and of the cell map)
    const float R = sqrtf(U) * 1.00001f;
    int lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = nn_cell(qq[k] - R - g.slack[k], g.o[k], g.inv[k], G);
        hi[k] = nn_cell(qq[k] + R + g.slack[k], g.o[k], g.inv[k], G);
    }
    // distance from q to the slab of cell layer c on axis k (0 inside; a flat axis is one unbounded slab)
    auto slab = [&](int k, int c) -> float {
        if (!(g.cs[k] < INFINITY)) return 0.f;
        const float l = g.o[k] + (float)c * g.cs[k] - g.slack[k], h = g.o[k] + (float)(c + 1) * g.cs[k] + g.slack[k];
        return fmaxf(fmaxf(l - qq[k], qq[k] - h), 0.f);
    };
    const float R2 = R * R;
    // A query whose ball spans many cell rows pays a dependent lookup per row and gathers points at
    // ~4x the per-point cost of a wave-uniform stream: hand it to the k_nn_far_* kernels.  (Threshold measured on 80k
    // queries against 100k points on a sphere: uniform queries 1.03 / 1.39 / 2.66 ms for 9 / 25 / 100 rows; queries sampled
    // on a nearby surface 0.59 / 0.47 / 0.47 ms — the far path has ~0.15 ms of dependent-load latency of its own.)
    if ((long long)(hi[2] - lo[2] + 1) * (hi[1] - lo[1] + 1) > kFarRows) {
        farKey[q] = far_key() | shapeBits;
        return;
    }
    for (int cz = lo[2]; cz <= hi[2]; ++cz) {
        const float dz = slab(2, cz);
        for (int cy = lo[1]; cy <= hi[1]; ++cy) {
            const float dy = slab(1, cy);
            const float rem = R2 - dy * dy - dz * dz;
            if (rem < 0.f) continue;                                 // the whole row is farther than R
            const float rx = sqrtf(rem) * 1.00001f;
            const int x0 = nn_cell(qx - rx - g.slack[0], g.o[0], g.inv[0], G), x1 = nn_cell(qx + rx + g.slack[0], g.o[0], g.inv[0], G);
            const int row = (cz * G + cy) * G;
            const int s = start[row + x0], e = start[row + x1 + 1];
            for (int j = s; j < e; j += 4) {                         // four gathers in flight per lane
                float4 p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = sorted[min(j + k, e - 1)];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = dist(p[k].x, p[k].y, p[k].z);
                    const int idx = __float_as_int(p[k].w);
                    if (j + k < e && (d < best || (d == best && idx < besti && best < 1e20f))) { best = d; besti = idx; }
                }
            }
        }
    }
    result[q] = besti;
}

// compact helper tables for k_nn_far: the representatives as (x, y, z, index) records and the first sorted slot of
// every cell row; both padded so that k_nn_far can read them eight entries at a time
constexpr int kNNBatch = 8;

__global__ __launch_bounds__(256) void k_nn_far_tables(const int *__restrict__ rep, const float *__restrict__ pts, int Gc3,
                                                       const int *__restrict__ start, int G, float4 *repList, int *nRep,
                                                       int *rowStart, float4 *sorted, size_t slice, int M)
{
    const int sb = blockIdx.y;
    pts += (size_t)sb * M * 3;
    SHAPE(rep); SHAPE(start); SHAPE(repList); SHAPE(nRep); SHAPE(rowStart); SHAPE(sorted);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Gc3) {
        const int r = rep[i];
        const unsigned long long m = __ballot(r >= 0);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(nRep, __popcll(m));
        base = __shfl(base, 0);
        if (r >= 0)
            repList[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] =
                make_float4(pts[r * 3], pts[r * 3 + 1], pts[r * 3 + 2], __int_as_float(r));
    }
    const int total = start[G * G * G];
    if (i < G * G + 2 * kNNBatch) rowStart[i] = i < G * G ? start[i * G] : total;
    if (i < 64) sorted[total + i] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // tile padding (never considered)
}

__global__ __launch_bounds__(64) void k_nn_far_pad(float4 *repList, const int *__restrict__ nRep, size_t slice)
{
    const int sb = blockIdx.x;
    SHAPE(repList); SHAPE(nRep);
    repList[*nRep + threadIdx.x] = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // tile padding (64 threads)
}

// Far queries (answer farther than one coarse ring, or a ball over more than kFarRows cell rows).  One lane per query,
// the queries SORTED by the Morton code of their cell so that a wave holds 64 neighbours, and everything a wave reads
// is wave-uniform (scalar loads, eight records per batch, points as SGPR operands — the vector memory pipe is idle):
//   k_nn_far_bound  A. every coarse cell's representative point bounds the answer from above;
//                   B. the points of the coarse cells holding the lanes' best representatives tighten the bound;
//   k_nn_far_rows   C. the cell rows (cz,cy) are visited; a row is skipped when its slab is farther than the current
//                      best of EVERY lane (same certified margins as k_nn_query), otherwise the x-range the lanes can
//                      still need is streamed;
//   k_nn_far_final  scatters the answers back to query order.
// Any point is a valid candidate for any lane (reading a few records past a row's end is harmless), and candidates are
// combined as the 64-bit word (distance bits, index) under min — distances are non-negative, so that is the
// lexicographic (distance, index) minimum = the reference's first strict minimum of an ascending scan — first in
// registers, then across the four waves of a block in LDS, then across blocks with one atomicMin per lane.
// A group of 64 queries at the centre of a sphere needs every point: 6.4 M distance evaluations that must not land on
// one CU, so the rows of every group are split over kFarSlices blocks.
// (Measured history, 80,640 uniform queries against 100,000 points on a sphere, plain scan 5.5 ms: far list appended
// with one same-address atomic per query and every far query scanning ALL points 7.2 ms; pruned rows, one wave per 64
// queries, one scalar load per record 14.3 ms; batches of eight and 4 / 16 waves per group 4.5 / 2.4 ms; A and B shared
// between the waves 1.55 ms — the group that needs every point then kept one CU busy for 1 ms.)
#ifndef NN_FAR_SLICES
#define NN_FAR_SLICES 8
#endif
constexpr int kFarSlices = NN_FAR_SLICES;   // blocks per group of 64 far queries in k_nn_far_rows
constexpr int kFarWaves = 4;            // waves per block; they split the block's records

struct FarLane {
    float qx, qy, qz, best;
    int besti;
    __device__ __forceinline__ void consider(const float4 p)
    {
        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        const int idx = __float_as_int(p.w);
        float d = 0.f;
        d += dx * dx;                                               // nearest_neighbor_cuda.cu:42
        d += dy * dy;                                               // :44
        d += dz * dz;                                               // :46
        if (d < best || (d == best && idx < besti && best < 1e20f)) { best = d; besti = idx; }
    }
    // records [s, e) of a wave-uniform table (and up to seven more: the tables are padded with d = inf records)
    __device__ __forceinline__ void stream(const float4 *__restrict__ src, int s, int e)
    {
        for (int j = s; j < e; j += kNNBatch) {
            float4 p[kNNBatch];
#pragma unroll
            for (int k = 0; k < kNNBatch; ++k) p[k] = src[j + k];
#pragma unroll
            for (int k = 0; k < kNNBatch; ++k) consider(p[k]);
        }
    }
    __device__ __forceinline__ unsigned long long packed() const
    {
        return ((unsigned long long)(unsigned)__float_as_int(best) << 32) | (unsigned)besti;
    }
    __device__ __forceinline__ void unpack(unsigned long long v)
    {
        best = __int_as_float((int)(v >> 32));
        besti = (int)(unsigned)v;
    }
    // every wave of the block leaves with the block's best answer so far
    __device__ __forceinline__ void share(unsigned long long (*s_pack)[64], int part, int lane)
    {
        s_pack[part][lane] = packed();
        __syncthreads();
        unsigned long long v = s_pack[0][lane];
#pragma unroll
        for (int w = 1; w < kFarWaves; ++w) v = min(v, s_pack[w][lane]);
        unpack(v);
        __syncthreads();
    }
};

// (keys carry the shape index above the key bits: within a shape's segment the comparison with its own notFar is the same)
__device__ __forceinline__ int far_count(const unsigned *__restrict__ farKeyS, int N, unsigned notFar)
{
    int lo = 0, hi = N;                                            // first sorted key == notFar
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (farKeyS[mid] < notFar) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kFarWaves * 64) void k_nn_far_bound(const float *__restrict__ queries, const NNGrid *__restrict__ gp,
                                                                 const int *__restrict__ start, const float4 *__restrict__ sorted,
                                                                 const float4 *__restrict__ repList, const int *__restrict__ nRepP,
                                                                 const float *__restrict__ pts, int N,
                                                                 const unsigned *__restrict__ farKeyS, const unsigned *__restrict__ farList,
                                                                 unsigned long long *bound, int *nFar, unsigned notFar, size_t slice,
                                                                 int M, int Nst, unsigned shapeShift)
{
    __shared__ unsigned long long s_pack[kFarWaves][64];
    const int sb = blockIdx.y;
    queries += (size_t)sb * N * 3; pts += (size_t)sb * M * 3; farKeyS += (size_t)sb * Nst; farList += (size_t)sb * Nst;
    SHAPE(gp); SHAPE(start); SHAPE(sorted); SHAPE(repList); SHAPE(nRepP); SHAPE(bound); SHAPE(nFar);
    const unsigned qbase = (unsigned)sb * (unsigned)Nst;            // farList holds positions in the all-shapes key array
    const int n = far_count(farKeyS, Nst, notFar | ((unsigned)sb << shapeShift));
    if (blockIdx.x == 0 && threadIdx.x == 0) *nFar = n;
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform for the compiler too: scalar loads below
    const int i = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;                              // whole block idle
    const int q = (int)(farList[i < n ? i : n - 1] - qbase);        // idle lanes shadow the last far query
    const NNGrid g = *gp;
    const int G = g.G, Gc = g.Gc;
    FarLane L;
    L.qx = queries[q * 3]; L.qy = queries[q * 3 + 1]; L.qz = queries[q * 3 + 2];
    L.best = 1e20f;                                                 // nearest_neighbor_cuda.cu:28
    L.besti = 0;
    // ---- A: representatives, one batch per wave and turn
    {
        const int nRep = *nRepP;
        for (int j = part * kNNBatch; j < nRep; j += kFarWaves * kNNBatch) L.stream(repList, j, j + kNNBatch);
    }
    L.share(s_pack, part, lane);
    // ---- B: the coarse cells of the lanes' best representatives (at most four distinct ones per block); a coarse cell
    //         is 4 x 4 rows of four cells: four rows per wave
    {
        const bool have = L.best < 1e20f;
        const int bi = have ? L.besti : 0;
        const float bx = pts[bi * 3], by = pts[bi * 3 + 1], bz = pts[bi * 3 + 2];
        const int mine = ((nn_cell(bz, g.o[2], g.inv[2], G) / kNNCoarse) * Gc + nn_cell(by, g.o[1], g.inv[1], G) / kNNCoarse) * Gc +
                         nn_cell(bx, g.o[0], g.inv[0], G) / kNNCoarse;
        unsigned long long todo = __ballot(have);
        for (int round = 0; round < 4 && todo; ++round) {
            const int c = __builtin_amdgcn_readlane(mine, __ffsll((long long)todo) - 1);
            todo &= ~__ballot(mine == c);
            const int x0 = (c % Gc) * kNNCoarse, x1 = min(x0 + kNNCoarse - 1, G - 1);
            const int cy0 = ((c / Gc) % Gc) * kNNCoarse, cz = (c / (Gc * Gc)) * kNNCoarse + part;
            int se[2 * kNNCoarse];
#pragma unroll
            for (int k = 0; k < kNNCoarse; ++k) {
                const int row = (min(cz, G - 1) * G + min(cy0 + k, G - 1)) * G;
                se[2 * k] = start[row + x0]; se[2 * k + 1] = start[row + x1 + 1];
            }
#pragma unroll
            for (int k = 0; k < kNNCoarse; ++k)
                if (cz < G && cy0 + k < G) L.stream(sorted, se[2 * k], se[2 * k + 1]);
        }
    }
    L.share(s_pack, part, lane);
    if (part == 0 && i < n) bound[i] = L.packed();
}

__global__ __launch_bounds__(kFarWaves * 64) void k_nn_far_rows(const float *__restrict__ queries, const NNGrid *__restrict__ gp,
                                                                const int *__restrict__ start, const float4 *__restrict__ sorted,
                                                                const int *__restrict__ rowStart, const int *__restrict__ nFar,
                                                                const unsigned *__restrict__ farList, unsigned long long *bound,
                                                                size_t slice, int N, int Nst)
{
    __shared__ unsigned long long s_pack[kFarWaves][64];
    const int sb = blockIdx.z;
    queries += (size_t)sb * N * 3; farList += (size_t)sb * Nst;
    SHAPE(gp); SHAPE(start); SHAPE(sorted); SHAPE(rowStart); SHAPE(nFar); SHAPE(bound);
    const unsigned qbase = (unsigned)sb * (unsigned)Nst;
    const int n = *nFar;
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;                              // whole block idle
    const int ii = i < n ? i : n - 1;                              // idle lanes shadow the last far query
    const int q = (int)(farList[ii] - qbase);
    const NNGrid g = *gp;
    const int G = g.G;
    FarLane L;
    L.qx = queries[q * 3]; L.qy = queries[q * 3 + 1]; L.qz = queries[q * 3 + 2];
    L.unpack(bound[ii]);                                           // as k_nn_far_bound left it (other slices may have improved it)
    const float qq[3] = {L.qx, L.qy, L.qz};
    auto slab = [&](int k, int c) -> float {                        // as in k_nn_query
        if (!(g.cs[k] < INFINITY)) return 0.f;
        const float l = g.o[k] + (float)c * g.cs[k] - g.slack[k], h = g.o[k] + (float)(c + 1) * g.cs[k] + g.slack[k];
        return fmaxf(fmaxf(l - qq[k], qq[k] - h), 0.f);
    };
    const int nb = (G + kNNBatch - 1) / kNNBatch;                   // work item = (layer cz, batch of eight rows cy0..cy0+7)
    for (int item = blockIdx.y * kFarWaves + part; item < G * nb; item += kFarWaves * kFarSlices) {
        const int cz = item / nb, cy0 = (item % nb) * kNNBatch;
        const float dz = slab(2, cz);
        if (!__any(!(L.best * 1.00003f - dz * dz < 0.f))) continue; // the whole layer is out of reach of every lane
        int rs[kNNBatch + 1];
#pragma unroll
        for (int k = 0; k <= kNNBatch; ++k) rs[k] = rowStart[cz * G + cy0 + k];   // padded: reads past G*G return the total
        // first the slice [s, e) of each of the eight rows that some lane can still need (lane k of vs/ve keeps row k's;
        // the eight pairs of cell-start loads are independent), then ONE copy of the streaming loop walks them
        int vs = 0, ve = 0;
#pragma unroll
        for (int k = 0; k < kNNBatch; ++k) {
            const int cy = cy0 + k;
            int s = 0, e = 0;
            if (cy < G && rs[k] != rs[k + 1]) {
                const float dy = slab(1, cy);
                // squared reach, inflated for the fp32 rounding of d and of the cell map (k_nn_query: R = sqrt(U) * 1.00001)
                const float rem = L.best * 1.00003f - dy * dy - dz * dz;
                const bool need = !(rem < 0.f);
                if (__any(need)) {
                    s = rs[k]; e = rs[k + 1];
                    if (e - s > 3 * kNNBatch) {                     // long row: only the x-range the lanes can reach
                        const float rx = sqrtf(fmaxf(rem, 0.f)) * 1.00001f;
                        int x0 = need ? nn_cell(L.qx - rx - g.slack[0], g.o[0], g.inv[0], G) : G - 1;
                        int x1 = need ? nn_cell(L.qx + rx + g.slack[0], g.o[0], g.inv[0], G) : 0;
                        if (need && !(rx < INFINITY)) { x0 = 0; x1 = G - 1; }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            x0 = min(x0, __shfl_xor(x0, off));
                            x1 = max(x1, __shfl_xor(x1, off));
                        }
                        x0 = __builtin_amdgcn_readfirstlane(x0);
                        x1 = __builtin_amdgcn_readfirstlane(x1);
                        const int row = (cz * G + cy) * G;
                        s = start[row + x0]; e = start[row + x1 + 1];
                    }
                }
            }
            vs = lane == k ? s : vs;
            ve = lane == k ? e : ve;
        }
#pragma unroll 1
        for (int k = 0; k < kNNBatch; ++k) L.stream(sorted, __builtin_amdgcn_readlane(vs, k), __builtin_amdgcn_readlane(ve, k));
    }
    L.share(s_pack, part, lane);
    if (part == 0 && i < n) atomicMin(&bound[i], L.packed());
}

__global__ __launch_bounds__(256) void k_nn_far_final(const unsigned long long *__restrict__ bound, const int *__restrict__ nFar,
                                                      const unsigned *__restrict__ farList, int *result, size_t slice, int N, int Nst)
{
    const int sb = blockIdx.y;
    farList += (size_t)sb * Nst; result += (size_t)sb * N;
    SHAPE(bound); SHAPE(nFar);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *nFar) result[farList[i] - (unsigned)sb * (unsigned)Nst] = (int)(unsigned)bound[i];
}

// ---------------------------------------------------------------------------- chamfer term of the surface loss
// layers/DefTet/deftet.py:174-177 composed (utils/mesh_utils.py:290-299, :360-374): K area-uniform samples per predicted
// face, each measured against its nearest ground-truth point (operator A10).  As torch expressions these are ~45 small
// launches per step (forward and backward); here: one to place the samples, one for the distances, one for the gradient.
constexpr float kChamferEps = 1e-10f;   // inside the square root (utils/mesh_utils.py:14)

// samples f32 [B, F*K, 3]: sample j of face f in row f*K + j = wa*a + wb*b + wc*c with s = sqrt(r0), wa = 1 - s,
// wb = s (1 - r1), wc = s r1 (square-root warp of two uniform numbers r f32 [2, B, F, K])
__global__ __launch_bounds__(256) void k_face_samples(const float *__restrict__ tri, const float *__restrict__ r, float *samples, int K,
                                                      long long total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float *t = tri + (i / K) * 9;
    const float s = sqrtf(r[i]), r1 = r[total + i];
    const float wa = 1.0f - s, wb = s * (1.0f - r1), wc = s * r1;
#pragma unroll
    for (int k = 0; k < 3; ++k) samples[i * 3 + k] = (wa * t[k] + wb * t[3 + k]) + wc * t[6 + k];
}

// d f32 [B,N] = sqrt(|sample - gt[idx]|^2 + eps) for the first n_valid[b] rows of shape b, 0 beyond
__global__ __launch_bounds__(256) void k_chamfer_fwd(const float *__restrict__ samples, const float *__restrict__ gt,
                                                     const int *__restrict__ idx, const int *__restrict__ n_valid, float *d, int N, int M)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const size_t i = (size_t)b * N + j;
    float v = 0.f;
    if (j < n_valid[b]) {
        const float *sp = samples + i * 3, *g = gt + ((size_t)b * M + idx[i]) * 3;
        const float dx = sp[0] - g[0], dy = sp[1] - g[1], dz = sp[2] - g[2];
        v = sqrtf(((dx * dx + dy * dy) + dz * dz) + kChamferEps);
    }
    d[i] = v;
}

// grad_tri f32 [B,F,3,3]: one lane per face adds up its K samples' gradients (no atomics); gscale f32 [B] = dL/d(sum of d)
__global__ __launch_bounds__(256) void k_chamfer_bwd(const float *__restrict__ samples, const float *__restrict__ gt,
                                                     const int *__restrict__ idx, const int *__restrict__ n_valid,
                                                     const float *__restrict__ d, const float *__restrict__ r,
                                                     const float *__restrict__ gscale, float *grad_tri, int F, int K, int M, long long total)
{
    const int b = blockIdx.y;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f}, gc[3] = {0.f, 0.f, 0.f};
    const int N = F * K;
    const float gs = gscale[b];
    for (int j = 0; j < K; ++j) {
        const int row = f * K + j;
        if (row >= n_valid[b]) break;
        const size_t i = (size_t)b * N + row;
        const float *sp = samples + i * 3, *g = gt + ((size_t)b * M + idx[i]) * 3;
        const float s = sqrtf(r[i]), r1 = r[total + i];
        const float wa = 1.0f - s, wb = s * (1.0f - r1), wc = s * r1;
        const float w = gs / d[i];                                     // d(sqrt(q + eps))/d(sample) = (sample - near) / d
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gk = w * (sp[k] - g[k]);
            ga[k] += wa * gk; gb[k] += wb * gk; gc[k] += wc * gk;
        }
    }
    float *o = grad_tri + ((size_t)b * F + f) * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = ga[k]; o[3 + k] = gb[k]; o[6 + k] = gc[k]; }
}

// ---------------------------------------------------------------------------- A8 face edge adjacency
__device__ __forceinline__ bool pos_equal(const float *a, const float *b)
{   // equal(), tet_face_adj_m_for.cu:26-35
    float diff = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float d = a[i] - b[i];
        if (d < 0) d = -d;
        diff += d;
    }
    return (double)diff <= 1e-15;
}

__global__ __launch_bounds__(256) void k_face_edge_adj(const float *__restrict__ face, float *__restrict__ adj, int F, int max_nei)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = f < F;
    float fa[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fa[i] = face[(size_t)(live ? f : 0) * 9 + i];
    int found = live ? 0 : max_nei;
    for (int g = 0; g < F; ++g) {                                   // ascending g, :95
        if (__ballot(found < max_nei) == 0ull) break;               // every lane of the wave is full (:104-106)
        float fb[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) fb[i] = face[(size_t)g * 9 + i];   // wave-uniform -> scalar loads
        // check_share (:38-69) through the 3x3 vertex-equality matrix: equal() is a pure
        // function of its two vertices, so evaluating each pair once gives the same boolean
        bool E[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) E[i][j] = pos_equal(fa + 3 * i, fb + 3 * j);
        bool share = false;
#pragma unroll
        for (int ia = 0; ia < 3; ++ia)
#pragma unroll
            for (int ib = 0; ib < 3; ++ib) {
                const int ia2 = (ia + 1) % 3, ib2 = (ib + 1) % 3;
                share = share || (E[ia][ib] && E[ia2][ib2]) || (E[ia][ib2] && E[ia2][ib]);   // :60, :63
            }
        if (share && g != f && found < max_nei) {                   // :96, :100-103
            adj[(size_t)f * max_nei + found] = (float)g;
            ++found;
        }
    }
}

// --- A8, hash-based (exact): O(F) instead of O(F^2) --------------------------------------------------
// equal(a,b) (L1 distance <= 1e-15 in fp32) can only hold if, coordinate by coordinate, the two
// floats are bitwise identical or both "tiny" (|x| < 2^-24: one ulp there is still > 1e-15 above
// that magnitude, so distinct non-tiny floats differ by >= 7e-15).  NaN/Inf coordinates never
// compare equal (the difference is NaN).  So a NECESSARY condition for two faces to share an edge
// is equality of a 192-bit edge key built from per-coordinate keys (float bits, or one tag for
// all tiny values, or a unique tag for non-finite ones).  Edge records are chained per slot of a hash table
// on that key (one atomicExch each), and every face runs the EXACT check_share() only on the chain
// entries whose full key equals the key of one of its three edges, keeping the 30 smallest neighbour
// ids in ascending order.  Two launches for a whole batch of surfaces (rounds 1-2 radix-sorted the
// records by the 192-bit key: three 64-bit passes, ~38 launches per surface).
using u64 = unsigned long long;
using u32 = unsigned int;

struct VKey { u32 x, y, z; };

__device__ __forceinline__ u32 coord_key(float v, u32 unique, bool &bad)
{
    const u32 b = __float_as_uint(v);
    if ((b & 0x7F800000u) == 0x7F800000u) { bad = true; return unique; }   // NaN / Inf: never equal to anything
    if (fabsf(v) < 5.9604645e-08f) return 0u;                                // tiny (|x| < 2^-24), includes +-0
    return b;
}
__device__ __forceinline__ VKey vertex_key(const float *v, u32 unique)
{
    bool bad = false;
    VKey k{coord_key(v[0], unique, bad), coord_key(v[1], unique, bad), coord_key(v[2], unique, bad)};
    if (bad) { k.x = 0xFFFFFFFFu; k.y = unique; k.z = 0u; }                 // 0xFFFFFFFF is itself a NaN pattern: no finite float has it
    return k;
}
__device__ __forceinline__ bool vkey_less(const VKey &a, const VKey &b)
{
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
}

// 64-bit hash of an edge key (a <= b in vkey order): the chains of a table slot hold every edge with that key plus
// the few that collide with it; keys are compared in full before check_share() runs.
__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
struct EKey { VKey a, b; };
__device__ __forceinline__ EKey edge_key(const float *fa, int f, int e)
{
    VKey a = vertex_key(fa + 3 * e, (u32)(f * 3 + e)), b = vertex_key(fa + 3 * ((e + 1) % 3), (u32)(f * 3 + (e + 1) % 3));
    if (vkey_less(b, a)) { const VKey t = a; a = b; b = t; }
    return EKey{a, b};
}
__device__ __forceinline__ bool ekey_equal(const EKey &p, const EKey &q)
{
    return p.a.x == q.a.x && p.a.y == q.a.y && p.a.z == q.a.z && p.b.x == q.b.x && p.b.y == q.b.y && p.b.z == q.b.z;
}
__device__ __forceinline__ u32 ekey_slot(const EKey &k, u32 mask)
{
    u64 h = mix64(((u64)k.a.x << 32) | k.a.y);
    h = mix64(h ^ (((u64)k.a.z << 32) | k.b.x));
    h = mix64(h ^ (((u64)k.b.y << 32) | k.b.z));
    return (u32)h & mask;
}

constexpr int kA8Shapes = 32;          // shapes per launch (their face counts travel by value)
struct A8Counts { int n[kA8Shapes]; };

// every edge record r = 3 f + e of shape blockIdx.y is pushed onto the chain of its table slot (one atomicExch)
__global__ __launch_bounds__(256) void k_edge_insert(const float *__restrict__ face, int Fmax, A8Counts cnt, u32 mask, int *head, int *next)
{
    const int b = blockIdx.y, F = cnt.n[b];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= F * 3) return;
    const int f = r / 3, e = r % 3;
    const EKey k = edge_key(face + ((size_t)b * Fmax + f) * 9, f, e);
    int *hb = head + (size_t)b * (mask + 1), *nb = next + (size_t)b * Fmax * 3;
    nb[r] = atomicExch(&hb[ekey_slot(k, mask)], r);
}

__device__ __forceinline__ bool check_share_exact(const float *fa, const float *fb)
{   // check_share, tet_face_adj_m_for.cu:38-69 (through the 3x3 vertex-equality matrix)
    bool E[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) E[i][j] = pos_equal(fa + 3 * i, fb + 3 * j);
    bool share = false;
#pragma unroll
    for (int ia = 0; ia < 3; ++ia)
#pragma unroll
        for (int ib = 0; ib < 3; ++ib) {
            const int ia2 = (ia + 1) % 3, ib2 = (ib + 1) % 3;
            share = share || (E[ia][ib] && E[ia2][ib2]) || (E[ia][ib2] && E[ia2][ib]);
        }
    return share;
}

constexpr int kMaxNeiFast = 32;        // the sorted path keeps its candidate list in registers/scratch


// every face walks the chains of its three edges: candidates with an identical 192-bit key get the EXACT check_share(),
// the max_nei smallest neighbour ids are kept in ascending order (the chain order does not matter)
__global__ __launch_bounds__(256) void k_face_neighbors(const float *__restrict__ face, int Fmax, A8Counts cnt, u32 mask,
                                                        const int *__restrict__ head, const int *__restrict__ next,
                                                        float *__restrict__ adj, int max_nei)
{
    const int b = blockIdx.y, F = cnt.n[b];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float *fbase = face + (size_t)b * Fmax * 9;
    const int *hb = head + (size_t)b * (mask + 1), *nxt = next + (size_t)b * Fmax * 3;
    float fa[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) fa[i] = fbase[(size_t)f * 9 + i];
    int nb[kMaxNeiFast];
    int n = 0;
    for (int e = 0; e < 3; ++e) {
        const EKey k = edge_key(fa, f, e);
        for (int rj = hb[ekey_slot(k, mask)]; rj >= 0; rj = nxt[rj]) {
            const int g = rj / 3;
            if (g == f) continue;
            if (n == max_nei && g > nb[n - 1]) continue;             // cannot be among the max_nei smallest
            float fb[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) fb[i] = fbase[(size_t)g * 9 + i];
            if (!ekey_equal(k, edge_key(fb, g, rj % 3))) continue;  // a hash collision, or another key in the same slot
            if (!check_share_exact(fa, fb)) continue;
            // insert g into the ascending list (no duplicates), keep the max_nei smallest
            int pos = 0;
            while (pos < n && nb[pos] < g) ++pos;
            if (pos < n && nb[pos] == g) continue;
            if (n < max_nei) ++n;
            for (int kk = n - 1; kk > pos; --kk) nb[kk] = nb[kk - 1];
            if (pos < n) nb[pos] = g;
        }
    }
    float *ab = adj + ((size_t)b * Fmax + f) * max_nei;
    for (int kk = 0; kk < n; ++kk) ab[kk] = (float)nb[kk];
}

// --- normal consistency on the A8 table (utils/mesh_utils.py:28-39 composed: unit normals, pairs, mean of 1 - cos) ----
// loss_b = mean over the valid entries (i, j = adj[i][k] >= 0) of 1 - <n_i, n_j>,  n = c / sqrt(|c|^2 + 1e-12),
// c = (v1 - v0) x (v2 - v0); 0 when there is no pair.  One workgroup per shape (fixed reduction order: deterministic
// forward); the backward adds the two adjacency directions with float atomics and chains through the normalisation and
// the cross product.  (The torch composition of the same thing gathers a [B, F, 30, 3] tensor and scatters it back in
// the backward: 1.8 ms per call at F = 4,056, B = 8, against ~0.03 ms here.)
constexpr int kNCThreads = 1024;
constexpr float kNormalEps = 1e-12f;

__device__ __forceinline__ void face_cross(const float *t, float *c)
{
    const float e1[3] = {t[3] - t[0], t[4] - t[1], t[5] - t[2]}, e2[3] = {t[6] - t[0], t[7] - t[1], t[8] - t[2]};
    c[0] = e1[1] * e2[2] - e1[2] * e2[1];
    c[1] = e1[2] * e2[0] - e1[0] * e2[2];
    c[2] = e1[0] * e2[1] - e1[1] * e2[0];
}

__device__ __forceinline__ float block_sum(float v, float *sh)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float tot = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += sh[k];     // fixed order
    return tot;
}

constexpr int kNCLdsFaces = 4096;       // normals of a shape kept in LDS up to this many faces (48 KB)
__global__ __launch_bounds__(kNCThreads) void k_normal_consistency_fwd(const float *__restrict__ tri, const float *__restrict__ adj,
                                                                       const int *__restrict__ n_face, float *loss, float *nrm,
                                                                       float *count, int Fmax, int max_nei)
{
    __shared__ float sh[kNCThreads / 64];
    __shared__ float s_n[kNCLdsFaces * 3];
    const int b = blockIdx.x, F = min(n_face[b], Fmax);
    const float *tb = tri + (size_t)b * Fmax * 9, *ab = adj + (size_t)b * Fmax * max_nei;
    float *nb = nrm + (size_t)b * Fmax * 3;
    const bool lds = F <= kNCLdsFaces;                               // block-uniform
    for (int f = threadIdx.x; f < F; f += kNCThreads) {
        float t[9], c[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = tb[(size_t)f * 9 + k];
        face_cross(t, c);
        const float r = 1.0f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + kNormalEps);
        const float n0 = c[0] * r, n1 = c[1] * r, n2 = c[2] * r;
        nb[f * 3] = n0; nb[f * 3 + 1] = n1; nb[f * 3 + 2] = n2;     // kept for the backward
        if (lds) { s_n[f * 3] = n0; s_n[f * 3 + 1] = n1; s_n[f * 3 + 2] = n2; }
    }
    __syncthreads();                                                 // the shape's normals are this workgroup's own writes
    // One (face, neighbour slot) pair per thread and step: the table is read with consecutive addresses, the normals come
    // from LDS, and nothing in a step depends on the previous one, so the unrolled steps overlap.  (One face per thread
    // walking its row — 30 dependent loads behind a branch, four faces after each other — took 65 us for 8 x 4,056
    // faces; the pair loop with the normals gathered from global memory 91 us.)  Surfaces of more than 4,096 faces keep
    // the row walk.
    float s = 0.f, cnt = 0.f;
    if (lds) {
        const int nPair = F * max_nei;
#pragma unroll 4
        for (int i = threadIdx.x; i < nPair; i += kNCThreads) {
            const int f = i / max_nei;
            const float a = ab[i];
            const bool valid = a >= 0.f && a < (float)F;            // (entries outside [0, F) — or NaN — count as "no neighbour")
            const int j = valid ? (int)a : f;
            const float d = s_n[f * 3] * s_n[j * 3] + s_n[f * 3 + 1] * s_n[j * 3 + 1] + s_n[f * 3 + 2] * s_n[j * 3 + 2];
            s += valid ? 1.0f - d : 0.f;
            cnt += valid ? 1.0f : 0.f;
        }
    } else {
        for (int f = threadIdx.x; f < F; f += kNCThreads) {
            const float n0 = nb[f * 3], n1 = nb[f * 3 + 1], n2 = nb[f * 3 + 2];
            for (int k = 0; k < max_nei; ++k) {
                const float a = ab[(size_t)f * max_nei + k];
                if (!(a >= 0.f && a < (float)F)) continue;
                const int j = (int)a;
                s += 1.0f - (n0 * nb[j * 3] + n1 * nb[j * 3 + 1] + n2 * nb[j * 3 + 2]);
                cnt += 1.0f;
            }
        }
    }
    const float S = block_sum(s, sh), C = block_sum(cnt, sh);
    if (threadIdx.x == 0) {
        count[b] = C;
        loss[b] = C > 0.f ? S / C : 0.f;
    }
}

// backward, two launches over (faces, shapes) grids (one workgroup per shape, as the forward still is, ran its ~4 faces x 30
// dependent neighbour loads per thread unhidden: 0.13 ms per 8 shapes):
//  scatter: G_k = d(sum of 1 - <n_i, n_j>)/d n_k = -(sum over row k of n_j) - (sum over the rows i that list k of n_i),
//           formed in acc (zeroed by the caller) with float atomics — the table need not be symmetric (rows are cut at
//           max_nei entries);
//  final:   chain through the normalisation and the cross product, one lane per face.
__global__ __launch_bounds__(256) void k_normal_consistency_bwd_scatter(const float *__restrict__ adj, const int *__restrict__ n_face,
                                                                        const float *__restrict__ nrm, float *acc, int Fmax, int max_nei)
{
    const int b = blockIdx.y, F = min(n_face[b], Fmax);
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float *ab = adj + ((size_t)b * Fmax + f) * max_nei, *nb = nrm + (size_t)b * Fmax * 3;
    float *accb = acc + (size_t)b * Fmax * 3;
    const float n0 = nb[f * 3], n1 = nb[f * 3 + 1], n2 = nb[f * 3 + 2];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < max_nei; ++k) {
        const float a = ab[k];
        if (!(a >= 0.f && a < (float)F)) continue;                  // same validity rule as the forward
        const int j = (int)a;
        s0 += nb[j * 3]; s1 += nb[j * 3 + 1]; s2 += nb[j * 3 + 2];
        unsafeAtomicAdd(&accb[j * 3], -n0); unsafeAtomicAdd(&accb[j * 3 + 1], -n1); unsafeAtomicAdd(&accb[j * 3 + 2], -n2);
    }
    unsafeAtomicAdd(&accb[f * 3], -s0); unsafeAtomicAdd(&accb[f * 3 + 1], -s1); unsafeAtomicAdd(&accb[f * 3 + 2], -s2);
}

__global__ __launch_bounds__(256) void k_normal_consistency_bwd_final(const float *__restrict__ tri, const int *__restrict__ n_face,
                                                                      const float *__restrict__ count, const float *__restrict__ gloss,
                                                                      const float *__restrict__ acc, float *gtri, int Fmax)
{
    const int b = blockIdx.y, F = min(n_face[b], Fmax);
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Fmax) return;
    const float C = count[b], w = C > 0.f ? gloss[b] / C : 0.f;
    float g[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = 0.f;
    if (f < F) {
        const float *tb = tri + ((size_t)b * Fmax + f) * 9, *accf = acc + ((size_t)b * Fmax + f) * 3;
        float t[9], c[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = tb[k];
        face_cross(t, c);
        const float r2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + kNormalEps, r = 1.0f / sqrtf(r2);
        const float G[3] = {accf[0] * w, accf[1] * w, accf[2] * w};
        const float cg = (c[0] * G[0] + c[1] * G[1] + c[2] * G[2]) * r * r * r;          // n = c r:  dL/dc = G r - c (c.G) r^3
        const float dc[3] = {G[0] * r - c[0] * cg, G[1] * r - c[1] * cg, G[2] * r - c[2] * cg};
        const float e1[3] = {t[3] - t[0], t[4] - t[1], t[5] - t[2]}, e2[3] = {t[6] - t[0], t[7] - t[1], t[8] - t[2]};
        // c = e1 x e2:  dL/de1 = e2 x dc,  dL/de2 = dc x e1
        const float d1[3] = {e2[1] * dc[2] - e2[2] * dc[1], e2[2] * dc[0] - e2[0] * dc[2], e2[0] * dc[1] - e2[1] * dc[0]};
        const float d2[3] = {dc[1] * e1[2] - dc[2] * e1[1], dc[2] * e1[0] - dc[0] * e1[2], dc[0] * e1[1] - dc[1] * e1[0]};
#pragma unroll
        for (int k = 0; k < 3; ++k) { g[k] = -(d1[k] + d2[k]); g[3 + k] = d1[k]; g[6 + k] = d2[k]; }
    }
    float *gb = gtri + ((size_t)b * Fmax + f) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) gb[k] = g[k];
}

// ---------------------------------------------------------------------------- A9 point -> triangle distance
__device__ __forceinline__ float divide_non_zero(float a)
{   // tet_analytic_distance_for.cu:40-52: `eps` is a double literal, the sum is formed in double
    if (a == 0) return (float)1e-10;
    if (a < 0) return (float)((double)a - 1e-10);
    if (a > 0) return (float)((double)a + 1e-10);
    return (float)1e-10;
}
__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float abs_ref(float a) { return a > 0.0f ? a : -a; }                 // :20-28
__device__ __forceinline__ float min3(float a, float b, float c) { float m = a; if (b < m) m = b; if (c < m) m = c; return m; }
__device__ __forceinline__ float min3_idx(float a, float b, float c)
{   // tet_analytic_distance_back.cu:139-152
    float m = a, i = 0.f;
    if (b < m) { m = b; i = 1.f; }
    if (c < m) { m = c; i = 2.f; }
    return i;
}
__device__ __forceinline__ float dist_point_sq(const float *a, const float *b)
{   // :139-146
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}
__device__ __forceinline__ float distance_line_square(const float *A, const float *B, const float *P)
{   // :148-170
    float PA[3], BA[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { PA[k] = P[k] - A[k]; BA[k] = B[k] - A[k]; }
    const float t = dot3(PA, BA) / divide_non_zero(dot3(BA, BA));
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float tmp = BA[k] * t; d[k] = PA[k] - tmp; }
    const float distance = dot3(d, d);
    if (t >= 0 && t <= 1) return distance;
    return -distance;
}

template <bool WANT_IDX>
__device__ __forceinline__ void line_distance(const float *a, const float *b, const float *c, const float *p, float *ret,
                                              float max_dis)
{   // cuda_line_distance, for.cu:172-220 / back.cu:348-403
    const float k1 = (b[1] - c[1]) * (p[0] - c[0]) + (c[0] - b[0]) * (p[1] - c[1]);
    const float k2 = (a[0] - c[0]) * (p[1] - c[1]) + (c[1] - a[1]) * (p[0] - c[0]);
    const float k3 = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1]);
    if (k3 == 0) { ret[0] = -1; return; }
    const float l1 = k1 / k3, l2 = k2 / k3, l3 = 1 - l1 - l2;
    float dis12 = distance_line_square(a, b, p);
    float dis23 = distance_line_square(b, c, p);
    float dis13 = distance_line_square(a, c, p);
    if (l1 >= 0 && l2 >= 0 && l3 >= 0) {
        ret[0] = 0;
        ret[1] = min3(abs_ref(dis12), abs_ref(dis23), abs_ref(dis13));
        if (WANT_IDX) ret[2] = min3_idx(abs_ref(dis12), abs_ref(dis23), abs_ref(dis13));
        return;
    }
    if (dis12 <= 0) dis12 = max_dis;
    if (dis23 <= 0) dis23 = max_dis;
    if (dis13 <= 0) dis13 = max_dis;
    const float min_line = min3(dis12, dis23, dis13);
    const float d1 = dist_point_sq(a, p), d2 = dist_point_sq(b, p), d3 = dist_point_sq(c, p);
    const float min_pt = min3(d1, d2, d3);
    if (min_line < min_pt) {
        ret[0] = 1; ret[1] = min_line;
        if (WANT_IDX) ret[2] = min3_idx(dis12, dis23, dis13);
    } else {
        ret[0] = 2; ret[1] = min_pt;
        if (WANT_IDX) ret[2] = min3_idx(d1, d2, d3);
    }
}

__device__ __forceinline__ void plane_project(const float *a, const float *b, const float *c, const float *p, float *ip,
                                              float &t_out)
{   // for.cu:227-238 / back.cu:411-421
    float r1[3], r2[3], n[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r1[k] = b[k] - a[k]; r2[k] = c[k] - a[k]; }
    n[0] = r1[1] * r2[2] - r1[2] * r2[1];
    n[1] = r1[2] * r2[0] - r1[0] * r2[2];
    n[2] = r1[0] * r2[1] - r1[1] * r2[0];
    float length = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);   // cuda_normalize, :128-137
    length = divide_non_zero(length);
    n[0] = n[0] / length; n[1] = n[1] / length; n[2] = n[2] / length;
    const float t = dot3(n, a) - dot3(n, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float m = n[k] * t; ip[k] = p[k] + m; }
    t_out = t;
}

template <bool WANT_IDX>
__device__ __forceinline__ float min_triangle_distance(const float *a, const float *b, const float *c, const float *p,
                                                       float *ret, float *ip, float max_dis)
{   // for.cu:222-254 (MAX_DIS 10000) / back.cu:406-436 (MAX_DIS 9999999)
    float t;
    plane_project(a, b, c, p, ip, t);
    const float distance_1 = t * t;
    line_distance<WANT_IDX>(a, b, c, ip, ret, max_dis);
    if (ret[0] == 0) return distance_1;
    if (ret[0] < 0) return max_dis;
    return distance_1 + ret[1];
}

// The same evaluation with a wave-level early out between its two halves.  Whatever the inside test decides, the value is
// t^2, t^2 + (a non-negative term) or max_dis >= every best-so-far, i.e. never below the fp32 square of the plane offset t
// computed first; so when t^2 > best holds for every voting lane the face can neither lower nor tie any of them and the
// expensive half (three point-line distances, ~350 of ~400 instructions) is skipped.  Bit-identical results.
// Returns false when skipped.
__device__ __forceinline__ bool min_triangle_distance_voted(const float *a, const float *b, const float *c, const float *p,
                                                            float best, bool voter, float &dis)
{
    float t, ip[3], ret[3] = {0.f, 0.f, 0.f};
    plane_project(a, b, c, p, ip, t);
    const float distance_1 = t * t;
    if (!__any(voter && distance_1 <= best)) return false;
    line_distance<false>(a, b, c, ip, ret, 10000.0f);
    dis = ret[0] == 0 ? distance_1 : (ret[0] < 0 ? 10000.0f : distance_1 + ret[1]);
    return true;
}

__global__ __launch_bounds__(256) void k_tri_dist_fwd(const float *__restrict__ pts, const float *__restrict__ face,
                                                      const float *__restrict__ n_face_b, float *closest_d,
                                                      float *closest_f, int P, int Fmax)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < P;
    const float *pp = pts + ((size_t)b * P + (live ? q : 0)) * 3;
    const float p[3] = {pp[0], pp[1], pp[2]};
    const int nf = (int)n_face_b[b];                                // for.cu:285
    const float *__restrict__ fb = face + (size_t)b * Fmax * 9;     // wave-uniform stream
    float min_d = 10000.0f;                                         // :277
    int min_idx = -1;
    for (int f = 0; f < nf; ++f) {
        float fc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) fc[i] = fb[(size_t)f * 9 + i];
        float ret[3] = {0.f, 0.f, 0.f}, ip[3];
        const float dis = min_triangle_distance<false>(fc, fc + 3, fc + 6, p, ret, ip, 10000.0f);
        if (min_d > dis) { min_d = dis; min_idx = f; }              // :300-303
    }
    if (live) {
        closest_d[(size_t)b * P + q] = min_d;
        closest_f[(size_t)b * P + q] = (float)min_idx;              // __int2float_rz, :306
    }
}

// --- A9 forward, grid-accelerated (exact) -------------------------------------------------------------
// Faces are binned by bounding box into a uniform grid whose cells are about one mean face extent
// wide; a point evaluates the reference formula only on the faces listed in the 3x3x3 (then 5x5x5)
// cells around it, plus a "wide" list of faces that every point must see, and stops once the best
// value is below a certified lower bound for every face it has not seen.  Faces on the wide list:
// non-finite, spanning > 64 cells, |n| < 1e-5 (the reference's normalisation adds 1e-10 to |n|),
// or nearly vertical (|k3| < |n|/64): for those the reference's xy-only inside test can misfire
// and report a plane distance for a far-away point, so their value is not bounded below by the
// true distance.  For every other face the reference value is the true squared distance up to
// rounding, which is what the pruning bound needs.  Points that are not settled
// within two shells go to the streaming scan (k_tri_far).  Results are combined lexicographically
// (value, face index) == "first strict minimum of the ascending scan" (for.cu:300-303).
constexpr int kTGMax = 64;             // cells per axis (upper bound)
constexpr int kTMaxCells = 64;         // faces overlapping more cells go to the wide list
constexpr int kTParts = 64;

#ifndef TRI_CELL_SCALE
#define TRI_CELL_SCALE 1.0f
#endif
constexpr float kTCellScale = TRI_CELL_SCALE;   // cell width in mean face extents

struct TGrid { float o[3], inv[3], cs[3], slack[3]; int g[3]; float abs_slack; };

__device__ __forceinline__ bool face_regular(const float *fc, float &lox, float &loy, float &loz, float &hix, float &hiy, float &hiz)
{
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) finite = finite && (fabsf(fc[k]) <= 1048576.0f);
    lox = fminf(fc[0], fminf(fc[3], fc[6])); hix = fmaxf(fc[0], fmaxf(fc[3], fc[6]));
    loy = fminf(fc[1], fminf(fc[4], fc[7])); hiy = fmaxf(fc[1], fmaxf(fc[4], fc[7]));
    loz = fminf(fc[2], fminf(fc[5], fc[8])); hiz = fmaxf(fc[2], fmaxf(fc[5], fc[8]));
    const float r1[3] = {fc[3] - fc[0], fc[4] - fc[1], fc[5] - fc[2]}, r2[3] = {fc[6] - fc[0], fc[7] - fc[1], fc[8] - fc[2]};
    const float nx = r1[1] * r2[2] - r1[2] * r2[1], ny = r1[2] * r2[0] - r1[0] * r2[2], nz = r1[0] * r2[1] - r1[1] * r2[0];
    const float len = sqrtf(nx * nx + ny * ny + nz * nz);
    const float k3 = (fc[4] - fc[7]) * (fc[0] - fc[6]) + (fc[6] - fc[3]) * (fc[1] - fc[7]);    // as cuda_line_distance computes it
    return finite && len >= 1e-5f && fabsf(k3) * 64.0f >= len && fabsf(nz) * 64.0f >= len;
}

// (the same launch clears what the later kernels accumulate into: cell counters, fill cursors, the counters block and the
// coarse-cell representatives)
__global__ __launch_bounds__(256) void k_tri_face_stats(const float *__restrict__ face, const float *__restrict__ nfb, float *part,
                                                        size_t slice, int Fmax, int *cnt0, int *fill0, int *counters, int *rep, int nc,
                                                        int nRepCells, int *pcnt0)
{
    __shared__ float sh[4][8];
    const int sb = blockIdx.y;
    face += (size_t)sb * Fmax * 9; nfb += sb;
    SHAPE(part); SHAPE(cnt0); SHAPE(fill0); SHAPE(counters); SHAPE(rep); SHAPE(pcnt0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) { cnt0[i] = 0; fill0[i] = 0; pcnt0[i] = 0; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nRepCells; i += gridDim.x * blockDim.x) rep[i] = -1;
    if (blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;
    const int nf = (int)nfb[0];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, sw = 0.f, cnt = 0.f;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
        float fc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) fc[k] = face[(size_t)f * 9 + k];
        float a, b, c, d, e, g;
        if (face_regular(fc, a, b, c, d, e, g)) {
            lo[0] = fminf(lo[0], a); lo[1] = fminf(lo[1], b); lo[2] = fminf(lo[2], c);
            hi[0] = fmaxf(hi[0], d); hi[1] = fmaxf(hi[1], e); hi[2] = fmaxf(hi[2], g);
            sw += fmaxf(fmaxf(d - a, e - b), g - c);
            cnt += 1.f;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        sw += __shfl_xor(sw, off);
        cnt += __shfl_xor(cnt, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sh[w][k] = lo[k]; sh[w][3 + k] = hi[k]; }
        sh[w][6] = sw; sh[w][7] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, sh[i][k]) : (k < 6 ? fmaxf(v, sh[i][k]) : v + sh[i][k]);
        part[blockIdx.x * 8 + k] = v;
    }
}

__global__ __launch_bounds__(64) void k_tri_grid(const float *__restrict__ part, TGrid *gp, size_t slice)
{
    const int sb = blockIdx.x;
    SHAPE(part); SHAPE(gp);
    const int lane = threadIdx.x;
    float lo[3], hi[3], sw = part[lane * 8 + 6], cnt = part[lane * 8 + 7];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = part[lane * 8 + k]; hi[k] = part[lane * 8 + 3 + k]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        sw += __shfl_xor(sw, off);
        cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) {
        TGrid r;
        const float meanw = cnt > 0.f ? sw / cnt : 0.f;
        float diag2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool ok = hi[k] >= lo[k];
            const float l = ok ? lo[k] : 0.f, h = ok ? hi[k] : 0.f, ext = h - l;
            float n = (meanw > 0.f && ext > 0.f) ? ceilf(ext / (meanw * kTCellScale)) : 1.f;
            n = fminf(fmaxf(n, 1.f), (float)kTGMax);
            r.g[k] = (int)n;
            r.o[k] = l;
            r.inv[k] = ext > 1e-30f ? n / ext : 0.f;
            r.cs[k] = ext > 1e-30f ? ext / n : INFINITY;
            r.slack[k] = 8e-6f * (fabsf(l) + fabsf(h)) + 1e-30f;
            diag2 += (fabsf(l) + fabsf(h)) * (fabsf(l) + fabsf(h));
        }
        r.abs_slack = 1e-8f * diag2;                               // see the bound in k_tri_query_coop
        *gp = r;
    }
}

__device__ __forceinline__ int t_cell(float x, float o, float inv, int G)
{
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.f), (float)(G - 1));
    return (int)f;
}

// mode 0: count cells / append to the wide list; mode 1: fill the cell lists
constexpr int kTCoarse = 4;            // coarse cells (4x4x4 cells) keep one representative face each for the far path
constexpr int kTGc = kTGMax / kTCoarse;

__global__ __launch_bounds__(256) void k_tri_face_bin(const float *__restrict__ face, const float *__restrict__ nfb,
                                                      const TGrid *__restrict__ gp, int mode, int *cellCount,
                                                      const int *__restrict__ cellStart, int *cellFill, int *list, int *wide,
                                                      int *nWide, int *rep, size_t slice, int Fmax, float4 *sph, uint2 *frange)
{
    const int sb = blockIdx.y;
    face += (size_t)sb * Fmax * 9; nfb += sb;
    SHAPE(gp); SHAPE(cellCount); SHAPE(cellStart); SHAPE(cellFill); SHAPE(list); SHAPE(wide); SHAPE(nWide); SHAPE(rep); SHAPE(sph); SHAPE(frange);
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= (int)nfb[0]) return;
    const TGrid g = *gp;
    float fc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) fc[k] = face[(size_t)f * 9 + k];
    float lox, loy, loz, hix, hiy, hiz;
    bool tiles = face_regular(fc, lox, loy, loz, hix, hiy, hiz);
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0, z0 = 0, z1 = 0;
    if (tiles) {
        x0 = t_cell(lox - g.slack[0], g.o[0], g.inv[0], g.g[0]); x1 = t_cell(hix + g.slack[0], g.o[0], g.inv[0], g.g[0]);
        y0 = t_cell(loy - g.slack[1], g.o[1], g.inv[1], g.g[1]); y1 = t_cell(hiy + g.slack[1], g.o[1], g.inv[1], g.g[1]);
        z0 = t_cell(loz - g.slack[2], g.o[2], g.inv[2], g.g[2]); z1 = t_cell(hiz + g.slack[2], g.o[2], g.inv[2], g.g[2]);
        tiles = (x1 - x0 + 1) * (y1 - y0 + 1) * (z1 - z0 + 1) <= kTMaxCells;
    }
    if (!tiles) {
        if (mode == 0) {
            wide[atomicAdd(nWide, 1)] = f;
            // the face's plane exactly as plane_project forms it (-ffp-contract=off: the same fp32 operations give the same
            // bits here and there): k_tri_query_coop gets the plane offset t of a point, which bounds the reference value
            // from below, from five operations instead of the whole projection
            float r1[3], r2[3], n[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { r1[k] = fc[3 + k] - fc[k]; r2[k] = fc[6 + k] - fc[k]; }
            n[0] = r1[1] * r2[2] - r1[2] * r2[1];
            n[1] = r1[2] * r2[0] - r1[0] * r2[2];
            n[2] = r1[0] * r2[1] - r1[1] * r2[0];
            float length = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            length = divide_non_zero(length);
            n[0] = n[0] / length; n[1] = n[1] / length; n[2] = n[2] / length;
            sph[f] = make_float4(n[0], n[1], n[2], dot3(n, fc));      // t(p) = w - dot3(n, p), as plane_project forms it
        }
        return;
    }
    if (mode == 0) {
        // bounding sphere about the centroid (a point OF the triangle): |p - c| - R <= distance(p, triangle) <= |p - c|.
        // k_tri_query_coop prunes with the lower bound; rounding of c and R is covered by the grid's per-axis slack there.
        float c[3], r2 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = (fc[k] + fc[3 + k] + fc[6 + k]) * (1.0f / 3.0f);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const float dx = fc[v * 3] - c[0], dy = fc[v * 3 + 1] - c[1], dz = fc[v * 3 + 2] - c[2];
            r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
        }
        sph[f] = make_float4(c[0], c[1], c[2], sqrtf(r2) * 1.00001f);
        // its cell range, six bits per bound: what k_tri_query_coop's canonical-cell rule needs of the face
        frange[f] = make_uint2((unsigned)x0 | (unsigned)x1 << 6 | (unsigned)y0 << 12 | (unsigned)y1 << 18 | (unsigned)z0 << 24, (unsigned)z1);
    }
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                const int c = (z * g.g[1] + y) * g.g[0] + x;
                if (mode == 0) atomicAdd(&cellCount[c], 1);
                else list[cellStart[c] + atomicAdd(&cellFill[c], 1)] = f | (x << 24);   // face | x cell (kTFaceBits)
            }
    if (mode == 0)                                                   // any face overlapping the coarse cell will do
        for (int z = z0 / kTCoarse; z <= z1 / kTCoarse; ++z)
            for (int y = y0 / kTCoarse; y <= y1 / kTCoarse; ++y)
                for (int x = x0 / kTCoarse; x <= x1 / kTCoarse; ++x) atomicMax(&rep[(z * kTGc + y) * kTGc + x], f);
}

// Points are sorted by their (clamped) grid cell — a counting sort: count + rank (one returning atomic per point), scan,
// scatter — so that the 64 lanes of a wave are neighbours in space, walk the same cells and fetch the same face records;
// the GT point cloud itself comes in arbitrary order.  (Round 2 used a library radix sort: ~20 launches per call.)  The
// order of the points inside a cell is whatever the atomics gave; no result depends on it.
__global__ __launch_bounds__(256) void k_tri_point_keys(const float *__restrict__ pts, int P, const TGrid *__restrict__ gp, int *pcount,
                                                        int2 *prank, size_t slice)
{
    const int sb = blockIdx.y;
    pts += (size_t)sb * P * 3;
    SHAPE(gp); SHAPE(pcount); SHAPE(prank);
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < P;                                         // (no early return: run_atomic_rank is a wave operation)
    const TGrid g = *gp;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float f = floorf((pts[(live ? q : 0) * 3 + k] - g.o[k]) * g.inv[k]);
        f = fminf(fmaxf(f, 0.f), (float)(g.g[k] - 1));               // NaN -> 0
        c[k] = (int)f;
    }
    const int cell = live ? (c[2] * kTGMax + c[1]) * kTGMax + c[0] : -1;   // 18 bits: z, y, x
    const int rank = run_atomic_rank(pcount, cell);                    // one atomic per run of lanes in the same cell
    if (live) prank[q] = make_int2(cell, rank);
}

__global__ __launch_bounds__(256) void k_tri_point_scatter(int P, const int2 *__restrict__ prank, const int *__restrict__ pstart,
                                                           unsigned *order, unsigned *skey, size_t slice)
{
    // order / skey: ONE array of P entries per shape (outside the slices)
    const int sb = blockIdx.y;
    order += (size_t)sb * P; skey += (size_t)sb * P;
    SHAPE(prank); SHAPE(pstart);
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const int2 r = prank[q];
    const int slot = pstart[r.x] + r.y;
    order[slot] = (unsigned)q;
    skey[slot] = (unsigned)r.x;
}

// ---- the grid query, wave-cooperative ------------------------------------------------------------
// Points arrive sorted by grid cell (z, y, x with x fastest); a CHUNK is up to 64 consecutive points of one (y, z) ROW of
// cells, i.e. of the x-adjacent cells [xa, xb].  The search box of shell r is [xa-r, xb+r] x [y-r, y+r] x [z-r, z+r]; per
// (y', z') row of the box the face-list entries of its cells are ONE contiguous range of `list` (cells are stored x
// fastest), which the wave walks 64 entries at a time: lane k loads entry k, the entries that survive the filters are
// broadcast one by one (v_readlane) and every lane tests its own point against the broadcast bounding sphere.
// (Round 2 gave a wave the points of ONE cell — ~10 of 64 lanes at the training-time density — and walked the box cell
// by cell, 27 or 125 short lists per chunk: the per-list overhead was what the kernel spent its time on, 2.2 ms per 8
// shapes; see profiles/r03_pmc_tri_query.json.)  The boxes are taken around the CHUNK's cells; a lane's termination test
// uses its own distance to that box, so the bound holds unchanged.  Extra evaluations cannot change the lexicographic
// (distance, index) minimum.
constexpr int kTRows = kTGMax * kTGMax;     // (y, z) rows of the grid
constexpr int kTFaceBits = 24;              // list entries: face index | x cell << 24 (faces < 2^24 is checked at the boundary)
constexpr int kTFaceMask = (1 << kTFaceBits) - 1;

__global__ __launch_bounds__(256) void k_tri_chunks(const int *__restrict__ pstart, int *ptStart, int *chunkCount, size_t slice)
{
    const int sb = blockIdx.y;
    SHAPE(pstart); SHAPE(ptStart); SHAPE(chunkCount);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;               // row = cell >> 6
    if (r > kTRows) return;
    const int s0 = pstart[r << 6];                                     // pstart has 64^3 + 1 entries, the last = P
    ptStart[r] = s0;
    chunkCount[r] = r < kTRows ? (pstart[(r + 1) << 6] - s0 + 63) >> 6 : 0;
}

__device__ __forceinline__ float bcastf(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

#ifdef TRI_STATS          // probe builds only (tools/probes/tri_stats_probe.py): where k_tri_query_coop spends its instructions
__device__ unsigned long long g_tri_stats[16];
#define TRI_STAT(i, v) do { if (lane == 0) atomicAdd(&g_tri_stats[i], (unsigned long long)(v)); } while (0)
#else
#define TRI_STAT(i, v)
#endif
constexpr int kTriChunkWaves = 4;      // waves per chunk of 64 points in k_tri_query_coop: they split the rows of the search box
constexpr int kTriCand = 8;            // per-lane candidate slots (LDS) between two drains in k_tri_query_coop

__device__ __forceinline__ void tri_query_chunk(int W, int part, float (*s_lb)[64], int (*s_cf)[64], int *s_q, unsigned long long *s_best,
                                                const uint2 *__restrict__ frange, const float4 *__restrict__ sph, const unsigned *__restrict__ skey,
                                                const float *__restrict__ pts, const float *__restrict__ face,
                                                const float *__restrict__ nfb, int P, const TGrid *__restrict__ gp,
                                                const int *__restrict__ cellStart, const int *__restrict__ list,
                                                const int *__restrict__ wide, const int *__restrict__ nWide, float *closest_d,
                                                float *closest_f, int *farFlag, const unsigned *__restrict__ order,
                                                const int *__restrict__ ptStart, const int *__restrict__ chunkStart,
                                                const int *__restrict__ rep)
{
    const int lane = threadIdx.x & 63;
    int lo = 0, hi = kTRows;                                         // largest row with chunkStart[row] <= W
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunkStart[mid] <= W) lo = mid; else hi = mid;
    }
    const int row = lo;
    const int slot = ptStart[row] + (W - chunkStart[row]) * 64 + lane;
    const bool live = slot < ptStart[row + 1];
    const int q = live ? (int)order[slot] : 0;
    int xa = live ? (int)(skey[slot] & 63u) : 63, xb = live ? (int)(skey[slot] & 63u) : 0;   // the chunk's cells on x
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        xa = min(xa, __shfl_xor(xa, off));
        xb = max(xb, __shfl_xor(xb, off));
    }
    xa = __builtin_amdgcn_readfirstlane(xa);
    xb = __builtin_amdgcn_readfirstlane(xb);
    const TGrid g = *gp;
    const float p[3] = {pts[q * 3], pts[q * 3 + 1], pts[q * 3 + 2]};
    const int nf = (int)nfb[0];
    if (live && part == 0) farFlag[slot] = 0;                      // every sorted slot belongs to exactly one live lane
    if (nf <= 0) {
        if (live && part == 0) { closest_d[q] = 10000.0f; closest_f[q] = -1.0f; }
        return;
    }
    const bool tame = fabsf(p[0]) <= 1048576.0f && fabsf(p[1]) <= 1048576.0f && fabsf(p[2]) <= 1048576.0f;
    float min_d = 10000.0f;                                         // for.cu:277
    int min_idx = -1;
    auto take = [&](int f, float dis) {
        if (min_d > dis || (min_d == dis && f < min_idx)) { min_d = dis; min_idx = f; }   // lexicographic (value, index)
    };
    // The four waves of the block work on the same 64 points (different rows of the box) and keep ONE best per point in
    // LDS, as a packed (value, index) word under atomicMin: a wave that only sees far rows prunes with what the wave on the
    // near rows has found.  Any published word is the value of an evaluated face, so pruning against it is as valid as
    // against the wave's own best, whenever it arrives; the final minimum does not depend on the timing.
    auto pack = [](float d, int f) { return ((unsigned long long)(unsigned)__float_as_int(d) << 32) | (unsigned)f; };
    auto publish = [&]() { atomicMin(&s_best[lane], pack(min_d, min_idx)); };
    auto refresh = [&]() {
        const unsigned long long v = s_best[lane];
        min_d = __int_as_float((int)(v >> 32));
        min_idx = (int)(unsigned)v;
    };
    // wave-uniform evaluation (the wide list): every lane evaluates the broadcast face, with the plane-offset vote
    auto eval = [&](int f, const float *fc) {
        float dis;
        if (min_triangle_distance_voted(fc, fc + 3, fc + 6, p, min_d, live, dis)) take(f, dis);
    };
    // Regular faces are pruned per LANE with their bounding sphere: the reference value of a regular face is its true
    // squared distance up to rounding (the certificate the shell test below already relies on: value >= 0.9998 d^2 -
    // abs_slack), and d >= |p - c| - R.  A lane keeps the faces it cannot rule out in kTriCand LDS slots and evaluates
    // them itself (its own gather of the 36-byte record), nearest sphere first, re-testing each against the best so far:
    // a handful of evaluations per point instead of one per face of the neighbourhood and wave.  A face ruled out has a
    // value strictly above the lane's best, so it can neither lower nor tie it: same lexicographic minimum.
    const float sph_slack = g.slack[0] + g.slack[1] + g.slack[2];
    int ncand = 0;
    auto ruled_out = [&](float lb2) { return 0.9998f * lb2 - g.abs_slack > min_d; };
    auto eval_own = [&](int f) {
        const float *src = face + (size_t)f * 9;
        float fc[9], ret[3] = {0.f, 0.f, 0.f}, ip[3];
#pragma unroll
        for (int j = 0; j < 9; ++j) fc[j] = src[j];
        take(f, min_triangle_distance<false>(fc, fc + 3, fc + 6, p, ret, ip, 10000.0f));
    };
    auto drain = [&]() {
        publish();
        refresh();
        // round 1: every lane evaluates its nearest candidate — after it the bests are tight
        int first = 0;
        float lbmin = INFINITY;
        for (int j = 0; j < kTriCand; ++j) {
            const float v = j < ncand ? s_lb[j][lane] : INFINITY;
            if (v < lbmin) { lbmin = v; first = j; }
        }
        if (ncand > 0 && !ruled_out(lbmin)) {
            eval_own(s_cf[first][lane]);
            publish();
        }
        // the candidates that survive the re-test go to a wave-wide queue of (lane, face) pairs and are evaluated 64 per
        // round whoever owns them: rounds = ceil(survivors / 64) instead of the largest per-lane count
        unsigned surv = 0u;
        for (int j = 0; j < kTriCand; ++j)
            if (j < ncand && j != first && !ruled_out(s_lb[j][lane])) surv |= 1u << j;
        const int mine = __popc(surv);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        TRI_STAT(4, 1);                                             // [4] drains
        TRI_STAT(6, (total + 63) >> 6);                             // [6] queue rounds
        TRI_STAT(7, total);                                         // [7] queued pair evaluations
        if (total > 0) {
            int pos = incl - mine;
            for (int j = 0; j < kTriCand; ++j)
                if (surv >> j & 1u) s_q[pos++] = (lane << kTFaceBits) | s_cf[j][lane];
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < total; base += 64) {
                const int e = base + lane < total ? s_q[base + lane] : -1;
                const int pl = e >= 0 ? e >> kTFaceBits : lane;
                const float pp[3] = {__shfl(p[0], pl), __shfl(p[1], pl), __shfl(p[2], pl)};
                if (e >= 0) {
                    const int f = e & kTFaceMask;
                    const float *src = face + (size_t)f * 9;
                    float fc[9], ret[3] = {0.f, 0.f, 0.f}, ip[3];
#pragma unroll
                    for (int j = 0; j < 9; ++j) fc[j] = src[j];
                    const float dis = min_triangle_distance<false>(fc, fc + 3, fc + 6, pp, ret, ip, 10000.0f);
                    atomicMin(&s_best[pl], pack(dis, f));           // lexicographic (value, index): values are >= +0, NaN packs above all
                }
            }
            __builtin_amdgcn_wave_barrier();
            refresh();
        }
        ncand = 0;
    };
    // A face overlapping several cells is listed in each of them: inside a search box it is taken only from its CANONICAL
    // cell — of its cells inside the box, the one nearest to the chunk's cells on every axis — once per distinct face.
    // reach2 / plo / phi (set per shell below): a regular face whose box is farther than sqrt(reach2) from the box of the
    // wave's unsettled points cannot bring any of them under its certification threshold, so it is not looked at here —
    // either the lane is settled by a nearer face or it goes to the far path, which is exact.
    float reach2 = INFINITY, plo[3] = {0.f, 0.f, 0.f}, phi[3] = {0.f, 0.f, 0.f};
    float bestmax = INFINITY;                                       // largest best among the wave's live points (wide list only)
    int boxn[3] = {1, 1, 1};                                        // cells per axis of the current search box
    const int Clo[3] = {xa, row & (kTGMax - 1), row >> 6}, Chi[3] = {xb, row & (kTGMax - 1), row >> 6};   // the chunk's (clamped) cells
    // entries [s, e): the lists of the cells [bx0, bx1] of row (cy, cz) when filter, else a stretch of the wide list
    // (rmod, rsel): of the 64-entry rounds of the range, this wave takes those with index % rmod == rsel
    auto list_run = [&](const int *__restrict__ lst, int s, int e, bool filter, int cy, int cz, int bx0, int by0, int bz0, int rmod, int rsel) {
        for (int base = s + rsel * 64; base < e; base += 64 * rmod) {
            TRI_STAT(1, 1);                                         // [1] 64-entry rounds
            TRI_STAT(2, min(64, e - base));                         // [2] list entries loaded
            const int idx = base + lane;
            const bool have = idx < e;
            const int ent = have ? lst[idx] : 0;
            const int fm = ent & kTFaceMask;
            float fv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool use = have;
            float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);             // regular faces: sphere; wide faces: plane
            float blo[3] = {0.f, 0.f, 0.f}, bhi[3] = {0.f, 0.f, 0.f}; // the face's bounding box
            if (have && filter) {
                // canonical cell: of the face's cells (as k_tri_face_bin listed them) inside the box, the one NEAREST to the
                // chunk's cells on every axis (so that a skipped row implies that all the face's cells are out of reach)
                const uint2 rg = frange[fm];
                const int f0[3] = {(int)(rg.x & 63u), (int)(rg.x >> 12 & 63u), (int)(rg.x >> 24 & 63u)};
                const int f1[3] = {(int)(rg.x >> 6 & 63u), (int)(rg.x >> 18 & 63u), (int)(rg.y & 63u)};
                const int cc[3] = {(int)((unsigned)ent >> kTFaceBits), cy, cz}, bb0[3] = {bx0, by0, bz0};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int lo_k = max(f0[k], bb0[k]), hi_k = min(f1[k], bb0[k] + boxn[k] - 1);
                    const int canon = lo_k > Chi[k] ? lo_k : (hi_k < Clo[k] ? hi_k : max(lo_k, Clo[k]));
                    use = use && cc[k] == canon;
                }
                if (use) {                                            // one listing in ~8 survives: only those fetch the record
#pragma unroll
                    for (int k = 0; k < 9; ++k) fv[k] = face[(size_t)fm * 9 + k];
                    sp = sph[fm];
                    float d2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        blo[k] = fminf(fv[k], fminf(fv[3 + k], fv[6 + k]));
                        bhi[k] = fmaxf(fv[k], fmaxf(fv[3 + k], fv[6 + k]));
                        const float d = fmaxf(fmaxf(plo[k] - bhi[k], blo[k] - phi[k]), 0.f);
                        d2 += d * d;
                    }
                    use = !(d2 * 0.9999f > reach2);
                }
            }
            if (have && !filter) {
#pragma unroll
                for (int k = 0; k < 9; ++k) fv[k] = face[(size_t)fm * 9 + k];
                sp = sph[fm];
                // wide face: its plane offset t over the box of the wave's points lies in [tlo, thi] (up to rounding, mag
                // bounds the operands); when even the smallest |t| squared exceeds every lane's best, the vote below fails
                // for every lane — decided here for 64 faces at once instead of one broadcast each
                float smax = 0.f, smin = 0.f, mag = fabsf(sp.w);
                const float nn[3] = {sp.x, sp.y, sp.z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float a = nn[k] * plo[k], b = nn[k] * phi[k];
                    smax += fmaxf(a, b); smin += fminf(a, b);
                    mag += fmaxf(fabsf(a), fabsf(b));
                }
                const float tlo = sp.w - smax, thi = sp.w - smin;
                const float tabs = fmaxf((tlo > 0.f ? tlo : (thi < 0.f ? -thi : 0.f)) - 1e-5f * mag, 0.f);
                use = !(tabs * tabs > bestmax);                           // NaN / Inf planes or boxes: looked at
            }
            unsigned long long todo = __ballot(use);
            TRI_STAT(filter ? 3 : 8, __popcll(todo));               // [3] faces broadcast (cell lists), [8] (wide list)
            while (todo) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int f = __builtin_amdgcn_readlane(fm, k);
                if (filter) {
                    const float dx = p[0] - bcastf(sp.x, k), dy = p[1] - bcastf(sp.y, k), dz = p[2] - bcastf(sp.z, k);
                    const float lb = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz) - bcastf(sp.w, k) - sph_slack, 0.f);
                    float box2 = 0.f;                                  // ... and the point's distance to its bounding box
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float d = fmaxf(fmaxf(bcastf(blo[a], k) - p[a], p[a] - bcastf(bhi[a], k)), 0.f);
                        box2 += d * d;
                    }
                    const float lb2 = fmaxf(lb * lb, box2);
                    if (live && tame && !ruled_out(lb2)) {
                        s_lb[ncand][lane] = lb2;
                        s_cf[ncand][lane] = f;
                        ++ncand;
                    }
                    if (__any(ncand == kTriCand)) drain();
                } else {
                    // the vote of eval() from the precomputed plane (the same fp32 operations as plane_project: same t)
                    const float n[3] = {bcastf(sp.x, k), bcastf(sp.y, k), bcastf(sp.z, k)};
                    const float t = bcastf(sp.w, k) - dot3(n, p);
                    if (!__any(live && t * t <= min_d)) continue;
                    float fc[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) fc[j] = bcastf(fv[j], k);
                    eval(f, fc);
                }
            }
        }
    };
    {   // No face within the coarse cells (4x4x4 cells each) around the chunk's cells: the two shells below cannot
        // settle anything — the whole wave goes to the far path at once (the wide list is evaluated there as well).
        bool any_face = false;
        for (int z = max(Clo[2] / kTCoarse - 1, 0); z <= min(Clo[2] / kTCoarse + 1, kTGc - 1); ++z)
            for (int y = max(Clo[1] / kTCoarse - 1, 0); y <= min(Clo[1] / kTCoarse + 1, kTGc - 1); ++y)
                for (int x = max(xa / kTCoarse - 1, 0); x <= min(xb / kTCoarse + 1, kTGc - 1); ++x)
                    any_face = any_face || rep[(z * kTGc + y) * kTGc + x] >= 0;
        // Likewise when every point of the wave lies more than three cells outside the grid (the faces' bounding box, onto
        // whose boundary cells such points are clamped): the shells reach two cells.  Sending a point to the far path is
        // always exact, only the cost differs.
        float out2 = 0.f, csmax = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!(g.cs[k] < INFINITY)) continue;
            const float d = fmaxf(fmaxf(g.o[k] - p[k], p[k] - (g.o[k] + (float)g.g[k] * g.cs[k])), 0.f);
            out2 += d * d;
            csmax = fmaxf(csmax, g.cs[k]);
        }
        const bool outside = out2 > 9.f * csmax * csmax;
        if (!any_face || __all(!live || outside)) {
            if (live && part == 0) farFlag[slot] = 1;
            return;
        }
    }
    if (part == 0) s_best[lane] = pack(10000.0f, -1);               // for.cu:277
    __syncthreads();
    bool done = false;
    TRI_STAT(0, 1);                                                 // [0] wave-chunks that search
    {
        const int nlive = __popcll(__ballot(live));
        TRI_STAT(10, nlive);                                        // [10] live lanes
        (void)nlive;
    }
    TRI_STAT(11, xb - xa + 1);                                      // [11] cells spanned on x
    for (int r = 1; r <= 2; ++r) {
        if (__all(done || !live || !tame)) break;
        TRI_STAT(11 + r, 1);                                        // [12] shells r = 1, [13] shells r = 2
        // the whole box [Clo-r, Chi+r] (clipped to the grid); r == 2 revisits the inner cells, which is cheap
        // with one look per distinct face and keeps the canonical rule simple
        const int b0[3] = {max(Clo[0] - r, 0), max(Clo[1] - r, 0), max(Clo[2] - r, 0)};
        const int b1[3] = {min(Chi[0] + r, g.g[0] - 1), min(Chi[1] + r, g.g[1] - 1), min(Chi[2] + r, g.g[2] - 1)};
        // every face not seen after this shell lies outside the box of cells [Clo-r, Chi+r]: distance >= m (per lane)
        float m = INFINITY;
        bool more = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!(g.cs[k] < INFINITY)) continue;                     // flat axis: one slab
            if (Clo[k] - r > 0) {
                more = true;
                m = fminf(m, fmaxf(p[k] - (g.o[k] + (float)(Clo[k] - r) * g.cs[k]) - g.slack[k], 0.f));
            }
            if (Chi[k] + r < g.g[k] - 1) {
                more = true;
                m = fminf(m, fmaxf((g.o[k] + (float)(Chi[k] + r + 1) * g.cs[k]) - p[k] - g.slack[k], 0.f));
            }
        }
        {   // what the unsettled lanes of this wave can still use: the largest threshold radius and the box of their points
            const bool act = live && tame && !done;
            float r2 = act ? (more ? m * m : INFINITY) : 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { plo[k] = act ? p[k] : INFINITY; phi[k] = act ? p[k] : -INFINITY; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                r2 = fmaxf(r2, __shfl_xor(r2, off));
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    plo[k] = fminf(plo[k], __shfl_xor(plo[k], off));
                    phi[k] = fmaxf(phi[k], __shfl_xor(phi[k], off));
                }
            }
            reach2 = r2;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) boxn[k] = b1[k] - b0[k] + 1;
        // distance^2 from the box of the unsettled points to cell layer c on axis k (wave-uniform values)
        auto gap2 = [&](int k, int c) -> float {
            if (!(g.cs[k] < INFINITY)) return 0.f;
            const float l = g.o[k] + (float)c * g.cs[k] - g.slack[k], h = g.o[k] + (float)(c + 1) * g.cs[k] + g.slack[k];
            const float d = fmaxf(fmaxf(l - phi[k], plo[k] - h), 0.f);
            return d * d;
        };
        {   // the chunk's own row first, its rounds dealt to the four waves: it holds the nearest faces of most points, and
            // what they give is in the shared bests before the other rows are pruned against them
            const int rowc = (Clo[2] * g.g[1] + Clo[1]) * g.g[0];
            const int s0 = cellStart[rowc + b0[0]], e0 = cellStart[rowc + b1[0] + 1];
            if (s0 < e0) list_run(list, s0, e0, true, Clo[1], Clo[2], b0[0], b0[1], b0[2], kTriChunkWaves, part);
            drain();
            __syncthreads();
            refresh();
        }
        int nth = 0;
        for (int z = b0[2]; z <= b1[2]; ++z)
            for (int y = b0[1]; y <= b1[1]; ++y) {
                if (y == Clo[1] && z == Clo[2]) continue;
                if ((nth++ % kTriChunkWaves) != part) continue;                               // this wave's rows of the box
                if (__all((gap2(1, y) + gap2(2, z)) * 0.9999f > reach2)) continue;             // the whole row is out of reach
                const int rowc = (z * g.g[1] + y) * g.g[0];
                const int s0 = cellStart[rowc + b0[0]], e0 = cellStart[rowc + b1[0] + 1];       // the row's cells: one range
                if (s0 < e0) list_run(list, s0, e0, true, y, z, b0[0], b0[1], b0[2], 1, 0);
            }
        if (r == 1) {
            // the wide list, AFTER the first shell: the lanes' bests are then small and most wide faces (typically the nearly
            // vertical ones of a grid-like surface, hundreds of them) fail the plane-offset vote of eval() at once
            drain();
            bestmax = (live && tame) ? min_d : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bestmax = fmaxf(bestmax, __shfl_xor(bestmax, off));
            const int nw = *nWide;                                  // every wave of the block takes every kTriChunkWaves-th batch
            list_run(wide, 0, nw, false, 0, 0, 0, 0, 0, kTriChunkWaves, part);
        }
        drain();
        publish();
        __syncthreads();                                            // every wave's work of this shell is in the shared bests
        refresh();
        __syncthreads();
        if (!more) done = true;                                      // the box covers the grid: every listed face was seen
        else if (min_d < (m * m) * 0.9998f - g.abs_slack) done = true;
    }
    if (!live || part != 0) return;
    if (!tame || !done) { farFlag[slot] = 1; return; }              // NaN / Inf / huge points, unresolved ones: the far path
    closest_d[q] = min_d;
    closest_f[q] = (float)min_idx;
}

// The number of chunks is only known on the device and at most P/64 + (number of rows); a fixed grid strides over them.
constexpr int kTriQueryBlocks = 8192;

// (Register budget of FOUR waves per SIMD: the allocator takes 145 VGPRs = three waves when left alone; held to 128 it
// spills nine dwords and the kernel runs 0.76 -> 0.63 ms at 8 x 97 k points — five waves, 96 VGPRs with 176 bytes of
// scratch, and six are slower again: geometry step 2.80 / 2.68 / 2.85 / 3.24 ms for 3 / 4 / 5 / 6 waves.)
#ifndef TRI_WAVES
#define TRI_WAVES 4
#endif
__attribute__((amdgpu_waves_per_eu(TRI_WAVES, 8)))
__global__ __launch_bounds__(kTriChunkWaves * 64) void k_tri_query_coop(const float *__restrict__ pts, const float *__restrict__ face,
                                                                         const float *__restrict__ nfb, int P, const TGrid *__restrict__ gp,
                                                                         const int *__restrict__ cellStart, const int *__restrict__ list,
                                                                         const int *__restrict__ wide, const int *__restrict__ nWide,
                                                                         float *closest_d, float *closest_f, int *farFlag,
                                                                         const unsigned *__restrict__ order, const int *__restrict__ ptStart,
                                                                         const int *__restrict__ chunkStart, const int *__restrict__ rep,
                                                                         size_t slice, int Fmax, const float4 *__restrict__ sph,
                                                                         const unsigned *__restrict__ skey, const uint2 *__restrict__ frange)
{
    const int sb = blockIdx.y;
    pts += (size_t)sb * P * 3; face += (size_t)sb * Fmax * 9; nfb += sb; closest_d += (size_t)sb * P; closest_f += (size_t)sb * P;
    order += (size_t)sb * P; skey += (size_t)sb * P;
    SHAPE(gp); SHAPE(cellStart); SHAPE(list); SHAPE(wide); SHAPE(nWide); SHAPE(farFlag); SHAPE(ptStart); SHAPE(chunkStart); SHAPE(rep); SHAPE(sph); SHAPE(frange);
    // One block per chunk of 64 points, its waves splitting the rows of the search box: one wave per chunk ran its chain of
    // dependent loads (cell starts -> list entries -> vertices) unhidden.
    __shared__ float s_lb[kTriChunkWaves][kTriCand][64];
    __shared__ int s_cf[kTriChunkWaves][kTriCand][64];
    __shared__ int s_q[kTriChunkWaves][kTriCand * 64];
    __shared__ unsigned long long s_best[64];                         // one (value, index) word per point of the chunk, all waves
    const int total = chunkStart[kTRows];
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int W = blockIdx.x; W < total; W += gridDim.x)
        tri_query_chunk(W, part, s_lb[part], s_cf[part], s_q[part], s_best, frange, sph, skey, pts, face, nfb, P, gp, cellStart, list, wide, nWide, closest_d,
                        closest_f, farFlag, order, ptStart, chunkStart, rep);
}

// ---- A9 far path: points the two shells did not settle (a surface still far from the cloud, early in training) ------
// Same shape as the A10 far path.  The unsettled points stay in cell order (flags + exclusive scan), 64 neighbours per
// group.  k_tri_far_bound evaluates the wide list and one representative face per coarse cell: an upper bound for every
// lane.  k_tri_far_rows walks the cell rows (cz,cy): a cell is skipped when 0.9997 dist^2 - abs_slack exceeds the current
// best of EVERY lane (the certified bound of k_tri_query_coop: every face listed only in skipped cells is a regular face
// farther than that), otherwise the list slice of the x-range the lanes can still reach is loaded cooperatively (lane k
// holds face k) and broadcast.  A face listed in several visited cells is evaluated once per block (bitset in LDS).  The
// rows of a group are split over kTriSlices blocks x 4 waves; answers are combined as 64-bit words (value bits, index)
// under min — the values are non-negative, so that is the reference's "first strict minimum of an ascending scan".
// (Before: one same-address atomic per unsettled point and the streaming scan over ALL faces for each of them —
// 5.2-5.7 ms against 4.8 ms for the plain scan when the 100 k points sit 30 % off a 4,032-face surface.)
constexpr int kTriSlices = 8;
constexpr int kTriWaves = 4;
constexpr int kTriBitWords = 4096;      // LDS bitset (16 KB): 131,072 faces; beyond that duplicates are simply evaluated again
static_assert(kTriBitWords * 4 + kTriWaves * 64 * 8 <= 32 * 1024, "k_tri_far_rows: keep the static LDS of the far path small");

struct TriLane {
    float p[3], min_d;
    int min_idx;
    __device__ __forceinline__ void eval(int f, const float *fc)
    {
        float dis;
        if (!min_triangle_distance_voted(fc, fc + 3, fc + 6, p, min_d, true, dis)) return;
        if (min_d > dis || (min_d == dis && f < min_idx)) { min_d = dis; min_idx = f; }   // lexicographic (value, index)
    }
    __device__ __forceinline__ unsigned long long packed() const
    {
        return ((unsigned long long)(unsigned)__float_as_int(min_d) << 32) | (unsigned)min_idx;   // index -1 is the largest
    }
    __device__ __forceinline__ void unpack(unsigned long long v)
    {
        min_d = __int_as_float((int)(v >> 32));
        min_idx = (int)(unsigned)v;
    }
    // entries [s, e) of a face list: 64 per round trip, lane k loads face k, the wave evaluates them one by one
    template <bool DEDUPE>
    __device__ __forceinline__ void run(const float *__restrict__ face, const int *__restrict__ lst, int s, int e, unsigned *bits, int lane)
    {
        for (int base = s; base < e; base += 64) {
            const int idx = base + lane;
            const int fm = idx < e ? (lst[idx] & 0xFFFFFF) : -1;       // cell lists carry their x cell above bit 24
            bool use = fm >= 0;
            if (DEDUPE && use) {
                const unsigned bit = 1u << (fm & 31);
                use = (atomicOr(&bits[fm >> 5], bit) & bit) == 0u;
            }
            float fv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) fv[k] = use ? face[(size_t)fm * 9 + k] : 0.f;
            unsigned long long todo = __ballot(use);
            while (todo) {
                const int k = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                float fc[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) fc[j] = bcastf(fv[j], k);
                eval(__builtin_amdgcn_readlane(fm, k), fc);
            }
        }
    }
    template <int W>
    __device__ __forceinline__ void share(unsigned long long (*s_pack)[64], int part, int lane)
    {
        s_pack[part][lane] = packed();
        __syncthreads();
        unsigned long long v = s_pack[0][lane];
#pragma unroll
        for (int w = 1; w < W; ++w) v = min(v, s_pack[w][lane]);
        unpack(v);
        __syncthreads();
    }
};

// unsettled points, still in cell order (off = exclusive scan of the flags); the last slot publishes their number
__global__ __launch_bounds__(256) void k_tri_compact(const int *__restrict__ flag, const int *__restrict__ off, const unsigned *__restrict__ order,
                                                     int P, int *farList, int *nFar, size_t slice)
{
    const int sb = blockIdx.y;
    order += (size_t)sb * P;
    SHAPE(flag); SHAPE(off); SHAPE(farList); SHAPE(nFar);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (flag[i]) farList[off[i]] = (int)order[i];
    if (i == P - 1) *nFar = off[i] + flag[i];
}

__global__ __launch_bounds__(kTriWaves * 64) void k_tri_far_bound(const float *__restrict__ pts, const float *__restrict__ face,
                                                                  const TGrid *__restrict__ gp, const int *__restrict__ wide,
                                                                  const int *__restrict__ nWide, const int *__restrict__ rep,
                                                                  const int *__restrict__ farList, const int *__restrict__ nFar,
                                                                  unsigned long long *bound, size_t slice, int P, int Fmax)
{
    __shared__ unsigned long long s_pack[kTriWaves][64];
    const int sb = blockIdx.y;
    pts += (size_t)sb * P * 3; face += (size_t)sb * Fmax * 9;
    SHAPE(gp); SHAPE(wide); SHAPE(nWide); SHAPE(rep); SHAPE(farList); SHAPE(nFar); SHAPE(bound);
    const int n = *nFar;
    if ((int)blockIdx.x * 64 >= n) return;                          // whole block idle
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    const int q = farList[i < n ? i : n - 1];                       // idle lanes shadow the last far point
    const TGrid g = *gp;
    TriLane L;
    L.p[0] = pts[q * 3]; L.p[1] = pts[q * 3 + 1]; L.p[2] = pts[q * 3 + 2];
    L.min_d = 10000.0f;                                             // for.cu:277
    L.min_idx = -1;
    const int nw = *nWide;
    for (int s = part * 64; s < nw; s += kTriWaves * 64) L.run<false>(face, wide, s, min(s + 64, nw), nullptr, lane);
    // One representative face per coarse cell (4x4x4 cells), in raster order, every fourth one per wave.  A representative
    // is evaluated only while its coarse cell is within reach of some lane's current best (same certified test as the rows
    // kernel; the face overlaps the cell, so it may also be closer: it is an upper bound either way), which leaves a few
    // dozen evaluations of the ~400-instruction distance formula instead of one per occupied coarse cell.
    const int gc[3] = {(g.g[0] + kTCoarse - 1) / kTCoarse, (g.g[1] + kTCoarse - 1) / kTCoarse, (g.g[2] + kTCoarse - 1) / kTCoarse};
    for (int cz = 0; cz < gc[2]; ++cz)
        for (int cy = 0; cy < gc[1]; ++cy)
            for (int cx0 = part * kNNBatch; cx0 < gc[0]; cx0 += kTriWaves * kNNBatch) {
                int r[kNNBatch];
#pragma unroll
                for (int k = 0; k < kNNBatch; ++k) r[k] = cx0 + k < gc[0] ? rep[(cz * kTGc + cy) * kTGc + cx0 + k] : -1;
#pragma unroll
                for (int k = 0; k < kNNBatch; ++k) {
                    if (r[k] < 0) continue;
                    float d2 = 0.f;
                    const int cc[3] = {cx0 + k, cy, cz};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (!(g.cs[a] < INFINITY)) continue;
                        const float l = g.o[a] + (float)(cc[a] * kTCoarse) * g.cs[a] - g.slack[a];
                        const float h = g.o[a] + (float)min((cc[a] + 1) * kTCoarse, g.g[a]) * g.cs[a] + g.slack[a];
                        const float d = fmaxf(fmaxf(l - L.p[a], L.p[a] - h), 0.f);
                        d2 += d * d;
                    }
                    if (!__any(!(d2 * 0.9997f - g.abs_slack > L.min_d))) continue;
                    float fc[9];
#pragma unroll
                    for (int j = 0; j < 9; ++j) fc[j] = face[(size_t)r[k] * 9 + j];       // wave-uniform: scalar loads
                    L.eval(r[k], fc);
                }
            }
    L.share<kTriWaves>(s_pack, part, lane);
    if (part == 0 && i < n) bound[i] = L.packed();
}

__global__ __launch_bounds__(kTriWaves * 64) void k_tri_far_rows(const float *__restrict__ pts, const float *__restrict__ face,
                                                                 const float *__restrict__ nfb, const TGrid *__restrict__ gp,
                                                                 const int *__restrict__ cellStart, const int *__restrict__ list,
                                                                 const int *__restrict__ farList, const int *__restrict__ nFar,
                                                                 unsigned long long *bound, size_t slice, int P, int Fmax)
{
    __shared__ unsigned long long s_pack[kTriWaves][64];
    __shared__ unsigned bits[kTriBitWords];
    const int sb = blockIdx.z;
    pts += (size_t)sb * P * 3; face += (size_t)sb * Fmax * 9; nfb += sb;
    SHAPE(gp); SHAPE(cellStart); SHAPE(list); SHAPE(farList); SHAPE(nFar); SHAPE(bound);
    const int n = *nFar;
    if ((int)blockIdx.x * 64 >= n) return;                          // whole block idle
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 64 + lane;
    const int ii = i < n ? i : n - 1;                               // idle lanes shadow the last far point
    const int q = farList[ii];
    const TGrid g = *gp;
    const int nf = (int)nfb[0];
    const bool dedupe = nf <= kTriBitWords * 32;
    if (dedupe) {
        for (int w = threadIdx.x; w < (nf + 31) / 32; w += kTriWaves * 64) bits[w] = 0u;
        __syncthreads();
    }
    TriLane L;
    L.p[0] = pts[q * 3]; L.p[1] = pts[q * 3 + 1]; L.p[2] = pts[q * 3 + 2];
    L.unpack(bound[ii]);                                           // as k_tri_far_bound left it (other slices may have improved it)
    auto slab = [&](int k, int c) -> float {                        // distance to the slab of cell layer c on axis k
        if (!(g.cs[k] < INFINITY)) return 0.f;
        const float l = g.o[k] + (float)c * g.cs[k] - g.slack[k], h = g.o[k] + (float)(c + 1) * g.cs[k] + g.slack[k];
        return fmaxf(fmaxf(l - L.p[k], L.p[k] - h), 0.f);
    };
    const int rows = g.g[1] * g.g[2];
    for (int r = blockIdx.y * kTriWaves + part; r < rows; r += kTriWaves * kTriSlices) {
        const int cy = r % g.g[1], cz = r / g.g[1];
        const float dy = slab(1, cy), dz = slab(2, cz);
        // a cell at distance d is out of reach when 0.9998 d^2 - abs_slack > best (k_tri_query_coop); 0.9997 covers the
        // rounding of this test itself
        const float rem = (L.min_d + g.abs_slack) * (1.0f / 0.9997f) - dy * dy - dz * dz;
        const bool need = !(rem < 0.f);                             // NaN points need everything (and select nothing)
        if (!__any(need)) continue;
        const float rx = sqrtf(fmaxf(rem, 0.f)) * 1.00001f;
        int x0 = need ? t_cell(L.p[0] - rx - g.slack[0], g.o[0], g.inv[0], g.g[0]) : g.g[0] - 1;
        int x1 = need ? t_cell(L.p[0] + rx + g.slack[0], g.o[0], g.inv[0], g.g[0]) : 0;
        if (need && !(rx < INFINITY)) { x0 = 0; x1 = g.g[0] - 1; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            x0 = min(x0, __shfl_xor(x0, off));
            x1 = max(x1, __shfl_xor(x1, off));
        }
        x0 = __builtin_amdgcn_readfirstlane(x0);
        x1 = __builtin_amdgcn_readfirstlane(x1);
        const int c0 = r * g.g[0];
        const int s = cellStart[c0 + x0], e = cellStart[c0 + x1 + 1];
        if (dedupe) L.run<true>(face, list, s, e, bits, lane);
        else L.run<false>(face, list, s, e, bits, lane);
    }
    L.share<kTriWaves>(s_pack, part, lane);
    if (part == 0 && i < n) atomicMin(&bound[i], L.packed());
}

__global__ __launch_bounds__(256) void k_tri_far_final(const unsigned long long *__restrict__ bound, const int *__restrict__ nFar,
                                                       const int *__restrict__ farList, float *closest_d, float *closest_f, size_t slice, int P)
{
    const int sb = blockIdx.y;
    closest_d += (size_t)sb * P; closest_f += (size_t)sb * P;
    SHAPE(bound); SHAPE(nFar); SHAPE(farList);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nFar) return;
    const unsigned long long v = bound[i];
    const int q = farList[i];
    closest_d[q] = __int_as_float((int)(v >> 32));
    closest_f[q] = (float)(int)(unsigned)v;
}

// per-point gradient contributions of the backward kernel (back.cu:591-686): up to 9 values
// for up to 3 vertices of the saved face.  Returns the number of (slot,value) pairs written.
// Gradient of one point's value with respect to the nine coordinates of its nearest face: g9[s], and the mask of the
// entries the reference's backward ADDS to (it also adds the zeros of the far edge endpoint, which matter only when the
// incoming gradient is not finite).  Everything is indexed statically — a (slot, value) list walked with a run-time count
// and corners addressed as fc + i * 3 put the lists and the corner array into scratch memory (48 bytes per lane, and 81
// compares per point to undo the list) in all three backward kernels.
__device__ __forceinline__ unsigned tri_dist_point_grad9(const float *fc, const float *p, float gp, float *g9)
{
    float ret[3] = {0.f, 0.f, 0.f}, ip[3];
    min_triangle_distance<true>(fc, fc + 3, fc + 6, p, ret, ip, 9999999.0f);     // back.cu:628
#pragma unroll
    for (int s = 0; s < 9; ++s) g9[s] = 0.f;
    auto corner = [&](int c, int k) { return c == 0 ? fc[k] : (c == 1 ? fc[3 + k] : fc[6 + k]); };
    if (ret[0] == 0) {                                              // :630-651, cuda_gradient_triangle_distance :439-483
        const float *a = fc, *b = fc + 3, *c = fc + 6;
        float ip2[3], t;
        plane_project(a, b, c, p, ip2, t);
        const float k1 = (b[1] - c[1]) * (ip2[0] - c[0]) + (c[0] - b[0]) * (ip2[1] - c[1]);
        const float k2 = (a[0] - c[0]) * (ip2[1] - c[1]) + (c[1] - a[1]) * (ip2[0] - c[0]);
        const float k3 = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1]);
        float grad[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (k3 != 0) {
            const float l1 = k1 / k3, l2 = k2 / k3, l3 = 1 - l1 - l2;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                grad[k] = 2 * (ip2[k] - p[k]) * l1;
                grad[3 + k] = 2 * (ip2[k] - p[k]) * l2;
                grad[6 + k] = 2 * (ip2[k] - p[k]) * l3;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) g9[k] = gp * grad[k];
        return 0x1FFu;
    }
    if (ret[0] == 1) {                                              // :652-670
        const int i1 = (int)ret[2], i2 = (i1 + 1) % 3;
        // cuda_gradient_line_distance (:291-317): the second assignment of grad[0..2] wins, grad[3..5] stays 0
        float A[3], B[3], PA[3], BA[3], v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { A[k] = corner(i1, k); B[k] = corner(i2, k); PA[k] = p[k] - A[k]; BA[k] = B[k] - A[k]; }
        const float t = dot3(PA, BA) / divide_non_zero(dot3(BA, BA));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float tmp = B[k] * t;
            float ipk = A[k] * (1 - t);
            ipk = ipk + tmp;
            v[k] = gp * (2 * (ipk - p[k]) * (t));
        }
        const float z = gp * 0.0f;
#pragma unroll
        for (int s = 0; s < 9; ++s) g9[s] = s / 3 == i1 ? v[s % 3] : (s / 3 == i2 ? z : 0.f);
        return (7u << (3 * i1)) | (7u << (3 * i2));
    }
    if (ret[0] == 2) {                                              // :671-685
        const int iv = (int)ret[2];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            float gl = corner(s / 3, s % 3) - p[s % 3];
            gl = gl * 1.0f;
            g9[s] = s / 3 == iv ? 2 * gp * gl : 0.f;
        }
        return 7u << (3 * iv);
    }
    return 0u;
}

__global__ __launch_bounds__(256) void k_tri_dist_bwd_atomic(const float *__restrict__ pts, const float *__restrict__ face,
                                                             const float *__restrict__ closest_f,
                                                             const float *__restrict__ dl_dd, float *dldface, int P, int F)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= P) return;
    const size_t i = (size_t)b * P + q;
    const int fi = (int)closest_f[i];                               // back.cu:618
    if (fi < 0 || fi >= F) return;                                  // the reference would read out of bounds
    float fc[9];
    const float *src = face + ((size_t)b * F + fi) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) fc[k] = src[k];
    const float p[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    float g9[9];
    const unsigned touched = tri_dist_point_grad9(fc, p, dl_dd[i], g9);
    float *g = dldface + ((size_t)b * F + fi) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k)
        if (touched >> k & 1u) unsafeAtomicAdd(g + k, g9[k]);                   // back.cu:640-683
}

// The same, walking the points in the FORWARD's order (sorted by grid cell): the 64 points of a wave are neighbours in
// space and share a handful of closest faces, so the wave adds up the contributions per distinct face first (butterfly over
// the lanes holding that face) and issues nine atomics per distinct face instead of up to nine per point — the atomics on
// ~36 k addresses per shape were what bounded the kernel above (0.37 ms per 8 x 97 k points).
__global__ __launch_bounds__(256) void k_tri_dist_bwd_grouped(const float *__restrict__ pts, const float *__restrict__ face,
                                                              const float *__restrict__ closest_f, const float *__restrict__ dl_dd,
                                                              const int *__restrict__ order, float *dldface, int P, int F)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = slot < P ? order[(size_t)b * P + slot] : -1;
    const size_t i = (size_t)b * P + (q >= 0 ? q : 0);
    int fi = q >= 0 ? (int)closest_f[i] : -1;                       // back.cu:618
    if (fi >= F) fi = -1;                                           // the reference would read out of bounds
    float gsum[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fi >= 0) {
        float fc[9];
        const float *src = face + ((size_t)b * F + fi) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) fc[k] = src[k];
        const float p[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
        float g9[9];
        const unsigned touched = tri_dist_point_grad9(fc, p, dl_dd[i], g9);
#pragma unroll
        for (int s = 0; s < 9; ++s)
            if (touched >> s & 1u) gsum[s] += g9[s];
    }
    unsigned long long todo = __ballot(fi >= 0);
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        const int key = __builtin_amdgcn_readlane(fi, L);
        const bool mine = fi == key;
        todo &= ~__ballot(mine);
        float v[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            v[s] = mine ? gsum[s] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[s] += __shfl_xor(v[s], off);
        }
        if (lane == L) {
            float *g = dldface + ((size_t)b * F + key) * 9;
#pragma unroll
            for (int s = 0; s < 9; ++s)
                if (v[s] != 0.f) unsafeAtomicAdd(g + s, v[s]);           // back.cu:640-683
        }
    }
}

// deterministic backward: (face, point) pairs sorted by face then point; one lane per face adds
// its points' contributions in ascending point order == the serial order of the CPU oracle.
__global__ __launch_bounds__(256) void k_bwd_keys(const float *__restrict__ closest_f, long long n, int P, int F,
                                                  unsigned long long *key)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / P);
    const int fi = (int)closest_f[i];
    const unsigned long long fkey = (fi < 0 || fi >= F) ? 0xFFFFFFFFull : (unsigned long long)((long long)b * F + fi);
    key[i] = (fkey << 32) | (unsigned long long)(i % P);
}

__global__ __launch_bounds__(256) void k_tri_dist_bwd_sorted(const float *__restrict__ pts, const float *__restrict__ face,
                                                             const float *__restrict__ dl_dd,
                                                             const unsigned long long *__restrict__ skey, long long n, int P,
                                                             int F, float *dldface)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = skey[i];
    const unsigned long long fkey = k >> 32;
    if (fkey == 0xFFFFFFFFull) return;
    if (i > 0 && (skey[i - 1] >> 32) == fkey) return;               // not the head of this face's segment
    const int b = (int)(fkey / (unsigned long long)F);
    float fc[9];
    const float *src = face + (size_t)fkey * 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) fc[j] = src[j];
    float acc[9];
    float *g = dldface + (size_t)fkey * 9;
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = g[j];
    for (long long y = i; y < n && (skey[y] >> 32) == fkey; ++y) {
        const size_t pi = (size_t)b * P + (size_t)(skey[y] & 0xFFFFFFFFull);
        const float p[3] = {pts[pi * 3], pts[pi * 3 + 1], pts[pi * 3 + 2]};
        float g9[9];
        const unsigned touched = tri_dist_point_grad9(fc, p, dl_dd[pi], g9);
#pragma unroll
        for (int s = 0; s < 9; ++s)
            if (touched >> s & 1u) acc[s] += g9[s];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) g[j] = acc[j];
}

}  // namespace surf
}  // namespace deftet

using namespace deftet;
using namespace deftet::surf;

#ifdef TRI_STATS
extern "C" int deftet_debug_tri_stats(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tri_stats), sizeof(g_tri_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tri_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

static int nn_pick_G(int M)
{
    // 1.5x the cells per axis that ~4 points per cell of a VOLUME-filling cloud would give (round 2): the clouds of this
    // operator are surface samples, which leaves ~10 points in an occupied cell (round 2: ~35, and the 3x3x3 neighbourhood
    // every query scans first held ~300 points).  Measured on the geometry step (8 x 97 k points, 8 x 80 k queries):
    // 3.49 / 3.32 / 3.27 / 3.22 ms per step at 1 / 1.4 / 1.8 / 2.4 x; a single shape (one call, 80 k queries) 0.32 / 0.29 /
    // 0.41 / 0.39 ms — finer grids cost a lone call more in rows visited per query than they save in points per row.
    // DEFTET_NN_GSCALE (read once; experiments) scales the cells per axis further.
    static const double scale = [] {
        const char *e = std::getenv("DEFTET_NN_GSCALE");
        const double v = e ? std::atof(e) : 1.0;
        return v > 0.1 && v < 10.0 ? v : 1.0;
    }();
    int G = (int)llround(scale * 1.5 * cbrt((double)(M > 0 ? M : 1) / 4.0));
    if (G < 1) G = 1;
    if (G > 160) G = 160;
    return G;
}

// per-shape scratch of the grid search (one slice per shape of a launch group), and the all-shapes far-key arrays + sort
// scratch behind the slices
static size_t nn_slice_bytes(int N, int M)
{
    const int G = nn_pick_G(M);
    const size_t nc = (size_t)G * G * G + 1;
    return align_up(nc * 4 * 3 + (size_t)(M > 0 ? M : 0) * (8 + 16) + (size_t)(N > 0 ? N : 0) * 8 + nc * 8 + nc + 8192 + (size_t)G * G * 4 +
                        ((size_t)1 << 20), 256);
}
static size_t nn_sort_bytes(int nShapes, int N)
{
    const size_t sortTmp = prims::radix_sort_temp_bytes<unsigned, unsigned>((size_t)nShapes * ((size_t)(N > 0 ? N : 0) + 1));
    return align_up(sortTmp, 256) + 4 * align_up((size_t)nShapes * ((size_t)(N > 0 ? N : 0) + 1) * 4, 256) + 1024;
}
extern "C" size_t deftet_nn_index_workspace_bytes(int B, int N, int M)
{
    const int g = B < 1 ? 1 : (B < kBatchShapes ? B : kBatchShapes);
    return nn_slice_bytes(N, M) * (size_t)g + nn_sort_bytes(g, N);
}

__global__ __launch_bounds__(256) void k_iota(unsigned *v, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (unsigned)i;
}

// the grid search for a GROUP of nS <= kBatchShapes shapes: one launch per kernel for the whole group
static int nn_group(const float *queries, const float *points, int32_t *result, int nS, int N, int M, const ShapeCounts &cnt, void *ws,
                    size_t wsb, hipStream_t st)
{
    const int G = nn_pick_G(M), keyBits = nn_far_key_bits(G);
    const size_t nc = (size_t)G * G * G + 1, slice = nn_slice_bytes(N, M);
    int nmax = 0;
    for (int i = 0; i < nS; ++i) nmax = std::max(nmax, cnt.n[i]);
    if (nmax == 0) return DEFTET_OK;
    Arena A(ws, slice);                                               // layout of slice 0; the kernels rebase to their shape
    float *part = A.take<float>(kNNBlocks * 6);
    NNGrid *grid = A.take<NNGrid>(1);
    int *cells = A.take<int>(nc), *start = A.take<int>(nc), *rep = A.take<int>(nc);
    int2 *pcell = A.take<int2>(M);
    float4 *sorted = A.take<float4>((size_t)M + 64);                                  // k_nn_far reads whole 64-record tiles
    float4 *repList = A.take<float4>(nc / (kNNCoarse * kNNCoarse) + 64 + 64);         // >= Gc^3 + pad
    int *rowStart = A.take<int>((size_t)G * G + 2 * kNNBatch), *nRep = A.take<int>(4);
    unsigned long long *bound = A.take<unsigned long long>((size_t)N + 1);
    int *nFar = A.take<int>(4);
    if (A.off > slice) return set_error(DEFTET_EINVAL, "nn slice layout exceeds its size");
    // behind the slices: far keys / sorted keys / iota / sorted positions of ALL shapes of the group, then the sort scratch
    const int Nst = N + 1;
    const size_t nAll = (size_t)nS * Nst;
    Arena T(static_cast<char *>(ws) + slice * nS, wsb - slice * nS);
    unsigned *farKey = T.take<unsigned>(nAll), *farKeyS = T.take<unsigned>(nAll), *iota = T.take<unsigned>(nAll), *farList = T.take<unsigned>(nAll);
    void *tmp = T.base + align_up(T.off, 256);
    const size_t left = (wsb - slice * nS) - align_up(T.off, 256);
    const unsigned shapeShift = (unsigned)keyBits + 1;
    const dim3 blk(256);
    DEFTET_LAUNCH(k_nn_bbox, dim3(kNNBlocks, nS), blk, st, points, M, part, slice, cells, rep, nRep, (int)nc);
    DEFTET_LAUNCH(k_nn_grid, dim3(nS), dim3(64), st, (const float *)part, G, grid, slice);
    DEFTET_LAUNCH(k_nn_bin, dim3((M + 255) / 256, nS), blk, st, points, M, (const NNGrid *)grid, cells, pcell, rep, slice);
    DEFTET_LAUNCH(k_scan_excl, dim3(scan_tiles(nc), nS), dim3(kScanThreads), st, (const int *)cells, start, (int)nc, slice, 0);
    DEFTET_LAUNCH(k_nn_scatter, dim3((M + 255) / 256, nS), blk, st, points, M, (const int2 *)pcell, (const int *)start, sorted, slice);
    DEFTET_LAUNCH(k_nn_query, dim3((Nst + 255) / 256, nS), blk, st, queries, N, (const NNGrid *)grid, (const int *)start,
                  (const float4 *)sorted, (const int *)rep, points, M, result, farKey, 1u << keyBits, slice, cnt, Nst, shapeShift);
    DEFTET_LAUNCH(k_iota, dim3((unsigned)((nAll + 255) / 256)), blk, st, iota, (long long)nAll);
    int shapeBitsN = 0;
    while ((1 << shapeBitsN) < nS) ++shapeBitsN;
    {
        const int rc = prims::radix_sort<unsigned, unsigned>(farKey, farKeyS, iota, reinterpret_cast<unsigned *>(farList), nAll,
                                                             keyBits + 1 + shapeBitsN, tmp, left, st);
        if (rc != DEFTET_OK) return rc;
    }
    const int Gc = (G + kNNCoarse - 1) / kNNCoarse, Gc3 = Gc * Gc * Gc, nt = std::max(Gc3, G * G + 2 * kNNBatch);
    DEFTET_LAUNCH(k_nn_far_tables, dim3((nt + 255) / 256, nS), blk, st, (const int *)rep, points, Gc3, (const int *)start, G, repList, nRep,
                  rowStart, sorted, slice, M);
    DEFTET_LAUNCH(k_nn_far_pad, dim3(nS), dim3(64), st, repList, (const int *)nRep, slice);
    DEFTET_LAUNCH(k_nn_far_bound, dim3((nmax + 63) / 64, nS), dim3(kFarWaves * 64), st, queries, (const NNGrid *)grid, (const int *)start,
                  (const float4 *)sorted, (const float4 *)repList, (const int *)nRep, points, N, (const unsigned *)farKeyS,
                  (const unsigned *)farList, bound, nFar, 1u << keyBits, slice, M, Nst, shapeShift);
    DEFTET_LAUNCH(k_nn_far_rows, dim3((nmax + 63) / 64, kFarSlices, nS), dim3(kFarWaves * 64), st, queries, (const NNGrid *)grid,
                  (const int *)start, (const float4 *)sorted, (const int *)rowStart, (const int *)nFar, (const unsigned *)farList, bound, slice,
                  N, Nst);
    DEFTET_LAUNCH(k_nn_far_final, dim3((nmax + 255) / 256, nS), blk, st, (const unsigned long long *)bound, (const int *)nFar,
                  (const unsigned *)farList, result, slice, N, Nst);
    return DEFTET_OK;
}

// n_query_host == NULL: every shape has N queries.  Otherwise shape b has n_query_host[b] <= N queries (rows beyond
// that are left untouched); the counts are HOST integers (the caller knows its face counts), strides stay N.
static int nn_index_impl(const float *queries, const float *points, int32_t *result, int B, int N, int M, const int *n_query_host,
                         void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && N >= 0 && M >= 0 && B <= 65535, "bad size");
    if (B == 0 || N == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(queries && result && (M == 0 || points), "null pointer");
    DEFTET_CHECK_ARG((long long)M * 3 < 2147483647LL && (long long)N * 3 < 2147483647LL, "too many points per shape");
    if (n_query_host)
        for (int b = 0; b < B; ++b) DEFTET_CHECK_ARG(n_query_host[b] >= 0 && n_query_host[b] <= N, "n_query[%d]=%d outside [0,%d]", b, n_query_host[b], N);
    hipStream_t st = as_stream(stream_);
    if (!workspace || M == 0) {
        if (!n_query_host) {
            DEFTET_LAUNCH(k_nn, dim3((N + 255) / 256, B), dim3(256), st, queries, points, N, M, result);
        } else {
            for (int b = 0; b < B; ++b)
                if (n_query_host[b] > 0)
                    DEFTET_LAUNCH(k_nn, dim3((n_query_host[b] + 255) / 256, 1), dim3(256), st, queries + (size_t)b * N * 3, points + (size_t)b * M * 3,
                                  n_query_host[b], M, result + (size_t)b * N);
        }
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && wsb >= deftet_nn_index_workspace_bytes(B, N, M),
                     "workspace misaligned or too small");
    for (int b0 = 0; b0 < B; b0 += kBatchShapes) {                   // groups of kBatchShapes shapes reuse the workspace (stream order)
        const int nS = std::min(kBatchShapes, B - b0);
        ShapeCounts cnt{};
        for (int i = 0; i < nS; ++i) cnt.n[i] = n_query_host ? n_query_host[b0 + i] : N;
        const int rc = nn_group(queries + (size_t)b0 * N * 3, points + (size_t)b0 * M * 3, result + (size_t)b0 * N, nS, N, M, cnt, workspace,
                                wsb, st);
        if (rc != DEFTET_OK) return rc;
    }
    return DEFTET_OK;
}

// workspace == NULL: the scalar-stream brute force; otherwise the grid search (both exact).
extern "C" int deftet_nn_index_f32(const float *queries, const float *points, int32_t *result, int B, int N, int M,
                                   void *workspace, size_t wsb, void *stream_)
{
    return nn_index_impl(queries, points, result, B, N, M, nullptr, workspace, wsb, stream_);
}

extern "C" int deftet_nn_index_ragged_f32(const float *queries, const float *points, int32_t *result, int B, int N_max, int M,
                                          const int *n_query_host, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(n_query_host || B == 0, "null n_query_host");
    return nn_index_impl(queries, points, result, B, N_max, M, n_query_host, workspace, wsb, stream_);
}

static u32 a8_table_mask(int F)
{
    u32 n = 64;
    while (n < (u32)(F > 0 ? F : 0) * 6u) n <<= 1;                   // >= 2 slots per edge record
    return n - 1;
}
static size_t a8_bytes(int B, int F_max)
{
    const size_t b = (size_t)(B > 0 ? B : 0);
    return align_up(b * ((size_t)a8_table_mask(F_max) + 1) * 4, 256) + align_up(b * (size_t)(F_max > 0 ? F_max : 0) * 3 * 4, 256) + 256;
}
extern "C" size_t deftet_face_edge_adj_workspace_bytes(int F) { return a8_bytes(1, F); }
extern "C" size_t deftet_face_edge_adj_ragged_workspace_bytes(int B, int F_max) { return a8_bytes(B, F_max); }

// A8 for a batch of surfaces with different face counts: face f32 [B, F_max, 3, 3], adj f32 [B, F_max, max_nei] (pre-filled
// with -1 by the caller), shape b has n_face_host[b] <= F_max faces (HOST integers); neighbour indices are local to the
// shape.  workspace == NULL (or max_nei > 32): the O(F^2) scan per shape; otherwise three launches per 32 shapes.
extern "C" int deftet_face_edge_adj_ragged_f32(const float *face, float *adj, int B, int F_max, const int *n_face_host, int max_nei,
                                               void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && F_max >= 0 && max_nei >= 0 && B <= 65535, "bad size");
    if (F_max >= (1 << 24)) return set_error(DEFTET_ELIMIT, "n_face=%d does not fit a float-encoded index", F_max);
    if (B == 0 || F_max == 0 || max_nei == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(face && adj && n_face_host, "null pointer");
    for (int b = 0; b < B; ++b) DEFTET_CHECK_ARG(n_face_host[b] >= 0 && n_face_host[b] <= F_max, "n_face[%d]=%d outside [0,%d]", b, n_face_host[b], F_max);
    hipStream_t st = as_stream(stream_);
    if (!workspace || max_nei > kMaxNeiFast) {
        for (int b = 0; b < B; ++b)
            if (n_face_host[b] > 0)
                DEFTET_LAUNCH(k_face_edge_adj, dim3((n_face_host[b] + 255) / 256), dim3(256), st, face + (size_t)b * F_max * 9,
                              adj + (size_t)b * F_max * max_nei, n_face_host[b], max_nei);
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && wsb >= a8_bytes(B, F_max), "workspace misaligned or too small");
    const u32 mask = a8_table_mask(F_max);
    Arena A(workspace, wsb);
    int *head = A.take<int>((size_t)B * (mask + 1)), *next = A.take<int>((size_t)B * F_max * 3);
    DEFTET_HIP(hipMemsetAsync(head, 0xFF, (size_t)B * (mask + 1) * 4, st));   // -1 = empty chain
    for (int b0 = 0; b0 < B; b0 += kA8Shapes) {
        const int nb = std::min(kA8Shapes, B - b0);
        A8Counts cnt{};
        int fmax = 0;
        for (int i = 0; i < nb; ++i) { cnt.n[i] = n_face_host[b0 + i]; fmax = std::max(fmax, cnt.n[i]); }
        if (fmax == 0) continue;
        const float *fb = face + (size_t)b0 * F_max * 9;
        DEFTET_LAUNCH(k_edge_insert, dim3((fmax * 3 + 255) / 256, nb), dim3(256), st, fb, F_max, cnt, mask, head + (size_t)b0 * (mask + 1),
                      next + (size_t)b0 * F_max * 3);
        DEFTET_LAUNCH(k_face_neighbors, dim3((fmax + 255) / 256, nb), dim3(256), st, fb, F_max, cnt, mask,
                      (const int *)(head + (size_t)b0 * (mask + 1)), (const int *)(next + (size_t)b0 * F_max * 3),
                      adj + (size_t)b0 * F_max * max_nei, max_nei);
    }
    return DEFTET_OK;
}

// workspace == NULL (or max_nei > 32): the scalar-stream brute force; otherwise the hash-based path.
extern "C" int deftet_face_edge_adj_f32(const float *face, float *adj, int F, int max_nei, void *workspace, size_t wsb,
                                        void *stream_)
{
    DEFTET_CHECK_ARG(F >= 0 && max_nei >= 0, "negative size");
    return deftet_face_edge_adj_ragged_f32(face, adj, 1, F, &F, max_nei, workspace, wsb, stream_);
}

// Normal consistency of B surfaces on their A8 tables (see k_normal_consistency_fwd).  tri f32 [B,F_max,3,3], adj f32
// [B,F_max,max_nei] (-1 padded, indices local to the shape), n_face int32 [B] ON THE DEVICE; loss f32 [B];
// nrm f32 [B,F_max,3] and count f32 [B] are saved for the backward.  grad_tri f32 [B,F_max,3,3] is fully overwritten
// (zeros for the padding faces); acc f32 [B,F_max,3] is scratch.
extern "C" int deftet_normal_consistency_fwd_f32(const float *tri, const float *adj, const int32_t *n_face, float *loss, float *nrm,
                                                 float *count, int B, int F_max, int max_nei, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && F_max >= 0 && max_nei >= 0, "bad size");
    if (B == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(loss && count && n_face && (F_max == 0 || (tri && nrm)) && (F_max == 0 || max_nei == 0 || adj), "null pointer");
    DEFTET_LAUNCH(k_normal_consistency_fwd, dim3(B), dim3(kNCThreads), as_stream(stream_), tri, adj, n_face, loss, nrm, count, F_max, max_nei);
    return DEFTET_OK;
}

extern "C" int deftet_normal_consistency_bwd_f32(const float *tri, const float *adj, const int32_t *n_face, const float *nrm,
                                                 const float *count, const float *grad_loss, float *grad_tri, float *acc, int B,
                                                 int F_max, int max_nei, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && F_max >= 0 && max_nei >= 0 && B <= 65535, "bad size");
    if (B == 0 || F_max == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(tri && n_face && nrm && count && grad_loss && grad_tri && acc && (max_nei == 0 || adj), "null pointer");
    hipStream_t st = as_stream(stream_);
    DEFTET_HIP(hipMemsetAsync(acc, 0, (size_t)B * F_max * 3 * sizeof(float), st));
    DEFTET_LAUNCH(k_normal_consistency_bwd_scatter, dim3((F_max + 255) / 256, B), dim3(256), st, adj, n_face, nrm, acc, F_max, max_nei);
    DEFTET_LAUNCH(k_normal_consistency_bwd_final, dim3((F_max + 255) / 256, B), dim3(256), st, tri, n_face, count, grad_loss, (const float *)acc,
                  grad_tri, F_max);
    return DEFTET_OK;
}

extern "C" int deftet_face_samples_f32(const float *tri, const float *r, float *samples, int B, int F, int K, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && F >= 0 && K >= 1, "bad size");
    const long long total = (long long)B * F * K;
    if (total == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(tri && r && samples, "null pointer");
    DEFTET_CHECK_ARG(total * 3 < (1LL << 40), "too many samples");
    DEFTET_LAUNCH(k_face_samples, dim3((unsigned)((total + 255) / 256)), dim3(256), as_stream(stream_), tri, r, samples, K, total);
    return DEFTET_OK;
}

extern "C" int deftet_chamfer_fwd_f32(const float *samples, const float *gt, const int32_t *idx, const int32_t *n_valid, float *d, int B,
                                      int N, int M, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && N >= 0 && M >= 1 && B <= 65535, "bad size");
    if (B == 0 || N == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(samples && gt && idx && n_valid && d, "null pointer");
    DEFTET_LAUNCH(k_chamfer_fwd, dim3((N + 255) / 256, B), dim3(256), as_stream(stream_), samples, gt, idx, n_valid, d, N, M);
    return DEFTET_OK;
}

extern "C" int deftet_chamfer_bwd_f32(const float *samples, const float *gt, const int32_t *idx, const int32_t *n_valid, const float *d,
                                      const float *r, const float *gscale, float *grad_tri, int B, int F, int K, int M, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && F >= 0 && K >= 1 && M >= 1 && B <= 65535, "bad size");
    if (B == 0 || F == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(samples && gt && idx && n_valid && d && r && gscale && grad_tri, "null pointer");
    DEFTET_LAUNCH(k_chamfer_bwd, dim3((F + 255) / 256, B), dim3(256), as_stream(stream_), samples, gt, idx, n_valid, d, r, gscale, grad_tri, F, K,
                  M, (long long)B * F * K);
    return DEFTET_OK;
}

// the per-shape scratch slice of the grid search: ONE description of the layout, used for the size and for the pointers
struct TriSlice {
    float *part; TGrid *grid;
    int *cnt, *start, *fill, *list, *wide, *farList, *counters, *farFlag, *farOff;
    unsigned long long *bound;
    int *rep, *ptStart, *chunkCount, *chunkStart;
    float4 *sph;
    uint2 *frange;
    int *pcount, *pstart;
    int2 *prank;
    size_t lay(void *base, int P, int Fmax)
    {
        const size_t nc = (size_t)kTGMax * kTGMax * kTGMax + 1, F = (size_t)(Fmax > 0 ? Fmax : 0), Pn = (size_t)(P > 0 ? P : 0);
        Arena A(base, ~(size_t)0);
        part = A.take<float>(kTParts * 8);
        grid = A.take<TGrid>(1);
        cnt = A.take<int>(nc); start = A.take<int>(nc); fill = A.take<int>(nc);
        list = A.take<int>(F * kTMaxCells + 1); wide = A.take<int>(F + 1);
        farList = A.take<int>(Pn + 1); counters = A.take<int>(8);
        farFlag = A.take<int>(Pn + 1); farOff = A.take<int>(Pn + 1);
        bound = A.take<unsigned long long>(Pn + 1);
        rep = A.take<int>(kTGc * kTGc * kTGc);
        ptStart = A.take<int>(kTRows + 2); chunkCount = A.take<int>(kTRows + 2); chunkStart = A.take<int>(kTRows + 2);
        sph = A.take<float4>(F + 1);
        frange = A.take<uint2>(F + 1);
        pcount = A.take<int>(nc); pstart = A.take<int>(nc + 1);
        prank = A.take<int2>(Pn + 1);
        return align_up(A.off, 256);
    }
};
static size_t tri_slice_bytes(int P, int Fmax)
{
    TriSlice L;
    return L.lay(nullptr, P, Fmax);
}
// behind the slices: the sorted cell keys and the point order of all shapes of a group (P entries per shape each)
static size_t tri_sort_bytes(int nShapes, int P)
{
    const size_t n = (size_t)nShapes * (size_t)(P > 0 ? P : 0);
    return 2 * align_up(n * 4 + 4, 256) + 1024;
}
// per-shape scratch slices for a launch group of <= kBatchShapes shapes + the all-shapes point keys and sort scratch
extern "C" size_t deftet_tri_dist_workspace_bytes(int B, int P, int Fmax)
{
    const int g = B < 1 ? 1 : (B < kBatchShapes ? B : kBatchShapes);
    return tri_slice_bytes(P, Fmax) * (size_t)g + tri_sort_bytes(g, P);
}

// the grid search for a GROUP of nS <= kBatchShapes shapes: one launch per kernel for the whole group
static int tri_dist_group(const float *pts, const float *face, const float *nfb, float *cd, float *cf, int nS, int P, int Fmax, void *ws,
                          size_t wsb, hipStream_t st, int *order_out)
{
    const size_t nc = (size_t)kTGMax * kTGMax * kTGMax + 1;
    TriSlice L;                                                       // layout of slice 0; the kernels rebase to their shape
    const size_t slice = L.lay(ws, P, Fmax);
    float *part = L.part;
    TGrid *grid = L.grid;
    int *cnt = L.cnt, *start = L.start, *fill = L.fill, *list = L.list, *wide = L.wide, *farList = L.farList, *counters = L.counters;
    int *farFlag = L.farFlag, *farOff = L.farOff, *rep = L.rep, *ptStart = L.ptStart, *chunkCount = L.chunkCount, *chunkStart = L.chunkStart;
    unsigned long long *bound = L.bound;
    float4 *sph = L.sph;
    uint2 *frange = L.frange;
    int *pcount = L.pcount, *pstart = L.pstart;
    int2 *prank = L.prank;
    const size_t nAll = (size_t)nS * P;
    if (slice * nS + tri_sort_bytes(nS, P) > wsb) return set_error(DEFTET_EINVAL, "tri_dist workspace smaller than its layout");
    Arena T(static_cast<char *>(ws) + slice * nS, wsb - slice * nS);
    unsigned *pskey = T.take<unsigned>(nAll + 1), *order = T.take<unsigned>(nAll + 1);
    const dim3 blk(256);
    DEFTET_LAUNCH(k_tri_face_stats, dim3(kTParts, nS), blk, st, face, nfb, part, slice, Fmax, cnt, fill, counters, rep, (int)nc, kTGc * kTGc * kTGc,
                  pcount);
    DEFTET_LAUNCH(k_tri_grid, dim3(nS), dim3(64), st, (const float *)part, grid, slice);
    DEFTET_LAUNCH(k_tri_face_bin, dim3((Fmax + 255) / 256, nS), blk, st, face, nfb, (const TGrid *)grid, 0, cnt, (const int *)start, fill, list, wide,
                  counters, rep, slice, Fmax, sph, frange);
    DEFTET_LAUNCH(k_scan_excl, dim3(scan_tiles(nc), nS), dim3(kScanThreads), st, (const int *)cnt, start, (int)nc, slice, 0);
    DEFTET_LAUNCH(k_tri_face_bin, dim3((Fmax + 255) / 256, nS), blk, st, face, nfb, (const TGrid *)grid, 1, cnt, (const int *)start, fill, list, wide,
                  counters, rep, slice, Fmax, sph, frange);
    DEFTET_LAUNCH(k_tri_point_keys, dim3((P + 255) / 256, nS), blk, st, pts, P, (const TGrid *)grid, pcount, prank, slice);
    DEFTET_LAUNCH(k_scan_excl, dim3(scan_tiles(nc), nS), dim3(kScanThreads), st, (const int *)pcount, pstart, (int)nc, slice, 0);
    DEFTET_LAUNCH(k_tri_point_scatter, dim3((P + 255) / 256, nS), blk, st, P, (const int2 *)prank, (const int *)pstart, order, pskey, slice);
    if (order_out) DEFTET_HIP(hipMemcpyAsync(order_out, order, nAll * 4, hipMemcpyDeviceToDevice, st));   // for the grouped backward
    DEFTET_LAUNCH(k_tri_chunks, dim3((kTRows + 1 + 255) / 256, nS), blk, st, (const int *)pstart, ptStart, chunkCount, slice);
    DEFTET_LAUNCH(k_scan_excl, dim3(scan_tiles(kTRows + 1), nS), dim3(kScanThreads), st, (const int *)chunkCount, chunkStart, kTRows + 1, slice, 0);
    {
        const long long maxChunks = (long long)(P + 63) / 64 + (long long)kTRows;
        DEFTET_LAUNCH(k_tri_query_coop, dim3((unsigned)std::min<long long>(maxChunks, kTriQueryBlocks), nS), dim3(kTriChunkWaves * 64), st, pts,
                      face, nfb, P, (const TGrid *)grid, (const int *)start, (const int *)list, (const int *)wide, (const int *)counters, cd, cf,
                      farFlag, (const unsigned *)order, (const int *)ptStart, (const int *)chunkStart, (const int *)rep, slice, Fmax, (const float4 *)sph,
                      (const unsigned *)pskey, (const uint2 *)frange);
    }
    // the far path (counters: [0] wide faces, [1] far points)
    DEFTET_LAUNCH(k_scan_excl, dim3(scan_tiles(P), nS), dim3(kScanThreads), st, (const int *)farFlag, farOff, P, slice, 0);
    DEFTET_LAUNCH(k_tri_compact, dim3((P + 255) / 256, nS), blk, st, (const int *)farFlag, (const int *)farOff, (const unsigned *)order, P, farList,
                  counters + 1, slice);
    DEFTET_LAUNCH(k_tri_far_bound, dim3((P + 63) / 64, nS), dim3(kTriWaves * 64), st, pts, face, (const TGrid *)grid, (const int *)wide,
                  (const int *)counters, (const int *)rep, (const int *)farList, (const int *)(counters + 1), bound, slice, P, Fmax);
    DEFTET_LAUNCH(k_tri_far_rows, dim3((P + 63) / 64, kTriSlices, nS), dim3(kTriWaves * 64), st, pts, face, nfb, (const TGrid *)grid,
                  (const int *)start, (const int *)list, (const int *)farList, (const int *)(counters + 1), bound, slice, P, Fmax);
    DEFTET_LAUNCH(k_tri_far_final, dim3((P + 255) / 256, nS), blk, st, (const unsigned long long *)bound, (const int *)(counters + 1),
                  (const int *)farList, cd, cf, slice, P);
    return DEFTET_OK;
}

// workspace == NULL: the scalar-stream brute force; otherwise the grid search (both exact).
// order_out (int32 [B,P], may be NULL): the points of every shape in the order the grid search walked them (sorted by
// grid cell) — what deftet_tri_dist_bwd_order_f32 wants; written only on the grid path with n_max_face > 0.
extern "C" int deftet_tri_dist_fwd_order_f32(const float *pts, const float *face, const float *n_face_b, float *closest_d,
                                             float *closest_f, int32_t *order_out, int B, int P, int Fmax, void *workspace, size_t wsb,
                                             void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && Fmax >= 0 && B <= 65535, "bad size");
    if (Fmax >= (1 << 24)) return set_error(DEFTET_ELIMIT, "n_face=%d does not fit a float-encoded index", Fmax);
    if (B == 0 || P == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts && n_face_b && closest_d && closest_f && (Fmax == 0 || face), "null pointer");
    hipStream_t st = as_stream(stream_);
    if (!workspace || Fmax == 0) {
        DEFTET_CHECK_ARG(!order_out, "the point order is only produced by the grid search (workspace and faces needed)");
        DEFTET_LAUNCH(k_tri_dist_fwd, dim3((P + 255) / 256, B), dim3(256), st, pts, face, n_face_b, closest_d, closest_f, P, Fmax);
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && wsb >= deftet_tri_dist_workspace_bytes(B, P, Fmax),
                     "workspace misaligned or too small");
    DEFTET_CHECK_ARG((long long)Fmax * kTMaxCells < 2147483647LL && (long long)P * 3 < 2147483647LL, "too many faces / points");
    for (int b0 = 0; b0 < B; b0 += kBatchShapes) {                   // groups of kBatchShapes shapes reuse the workspace (stream order)
        const int nS = std::min(kBatchShapes, B - b0);
        const int rc = tri_dist_group(pts + (size_t)b0 * P * 3, face + (size_t)b0 * Fmax * 9, n_face_b + b0, closest_d + (size_t)b0 * P,
                                      closest_f + (size_t)b0 * P, nS, P, Fmax, workspace, wsb, st,
                                      order_out ? order_out + (size_t)b0 * P : nullptr);
        if (rc != DEFTET_OK) return rc;
    }
    return DEFTET_OK;
}

extern "C" int deftet_tri_dist_fwd_f32(const float *pts, const float *face, const float *n_face_b, float *closest_d,
                                       float *closest_f, int B, int P, int Fmax, void *workspace, size_t wsb, void *stream_)
{
    return deftet_tri_dist_fwd_order_f32(pts, face, n_face_b, closest_d, closest_f, nullptr, B, P, Fmax, workspace, wsb, stream_);
}

// The atomic backward with the forward's point order (deftet_tri_dist_fwd_order_f32): same sums, up to the order of the
// floating-point additions, from ~7x fewer atomics.
extern "C" int deftet_tri_dist_bwd_order_f32(const float *pts, const float *face, const float *closest_f, const float *dl_dd,
                                             const int32_t *order, float *dldface, int B, int P, int F, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && B <= 65535, "bad size");
    if (B == 0 || P == 0 || F == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts && face && closest_f && dl_dd && dldface && order, "null pointer");
    DEFTET_LAUNCH(k_tri_dist_bwd_grouped, dim3((P + 255) / 256, B), dim3(256), as_stream(stream_), pts, face, closest_f, dl_dd, order, dldface,
                  P, F);
    return DEFTET_OK;
}

extern "C" int deftet_tri_dist_bwd_f32(const float *pts, const float *face, const float *closest_f, const float *dl_dd,
                                       float *dldface, int B, int P, int F, int deterministic, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && B <= 65535, "bad size");
    if (B == 0 || P == 0 || F == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts && face && closest_f && dl_dd && dldface, "null pointer");
    hipStream_t st = as_stream(stream_);
    if (!deterministic) {
        DEFTET_LAUNCH(k_tri_dist_bwd_atomic, dim3((P + 255) / 256, B), dim3(256), st, pts, face, closest_f, dl_dd, dldface, P, F);
        return DEFTET_OK;
    }
    // deterministic mode allocates its own scratch (stream-ordered), it is a test/debug path
    const long long n = (long long)B * P;
    DEFTET_CHECK_ARG((long long)B * F < 0xFFFFFFFFLL, "too many faces for the deterministic path");
    unsigned long long *key = nullptr, *skey = nullptr;
    void *tmp = nullptr;
    const size_t need = prims::radix_sort_temp_bytes<unsigned long long, unsigned>((size_t)n, false);
    DEFTET_HIP(hipMallocAsync((void **)&key, (size_t)n * 8, st));
    DEFTET_HIP(hipMallocAsync((void **)&skey, (size_t)n * 8, st));
    DEFTET_HIP(hipMallocAsync(&tmp, need, st));
    DEFTET_LAUNCH(k_bwd_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), st, closest_f, n, P, F, key);
    {
        const int rc = prims::radix_sort_keys<unsigned long long>(key, skey, (size_t)n, 64, tmp, need, st);
        if (rc != DEFTET_OK) return rc;
    }
    DEFTET_LAUNCH(k_tri_dist_bwd_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), st, pts, face, dl_dd, skey, n, P, F, dldface);
    DEFTET_HIP(hipFreeAsync(key, st));
    DEFTET_HIP(hipFreeAsync(skey, st));
    DEFTET_HIP(hipFreeAsync(tmp, st));
    return DEFTET_OK;
}
