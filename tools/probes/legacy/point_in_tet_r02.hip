// point_in_tet.hip — A1 point-in-tet occupancy query (+ A1b weights/backward, paste_occ)
// for CDNA4 / gfx950.  See DESIGN.md section "A1" for the algorithm and its error analysis.
//
// Semantics follow /root/reference/layers/DefTet/check_condition_tetrahedron_base/
// check_condition_tet_for.cu:105-189: for every query the LOWEST tet index t whose four
// same-side tests agree (all true or all false), evaluated in fp32 in the source's
// operation order WITHOUT fused multiply-add; else -1.
//
// Design (tet-centric, not a translation of the one-thread-per-query scan):
//   1. queries are counting-sorted into a uniform G^3 grid spanning their own bounding box;
//   2. one lane per tet: the lane computes the tet's four face planes once, visits the
//      grid cells overlapped by its (slightly enlarged) bounding box and runs the exact
//      predicate on the queries stored there; hits are combined with atomicMin, which
//      makes the result independent of evaluation order;
//   3. tets that fail a conditioning test ("irregular": tiny/flat/inverted-inconsistent,
//      non-finite, huge) are tested against ALL queries, and queries that are
//      non-finite/huge are tested against ALL tets, by two brute-force side kernels, so
//      the result equals the reference scan for every input, not just for nice meshes.
//
// The whole file is compiled with -ffp-contract=off; the pragma below repeats that.
#pragma clang fp contract(off)

#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.hpp"

// The round-2 algorithm ids (the product header now has AUTO / BRUTE / EXACT only).  This file is NOT part of
// libdeftet_hip.so: tools/probes/build_variant.sh <out> --legacy builds it into a probe library for A/B timing.
#undef DEFTET_PIT_EXACT
#define DEFTET_PIT_STAGED 2
#define DEFTET_PIT_ROWS 3
#define DEFTET_PIT_FMA 4
#define DEFTET_PIT_FMA2 5
#define DEFTET_PIT_GRP2 6
#define DEFTET_PIT_GRP4 7
#define DEFTET_PIT_GRP6 8
#define DEFTET_PIT_LDSB 9
#define DEFTET_PIT_LDS 10
#define DEFTET_PIT_EXACT 11

namespace deftet {
namespace pit {

constexpr float kBig = 1048576.0f;           // 2^20: coordinates beyond this go to the brute side paths
constexpr float kTau = 1.0f / 128.0f;        // regular tet: min |6V| >= tau * w^3
constexpr float kMargin = 1.0f / 64.0f;      // bounding-box enlargement, in units of w
constexpr float kWMin = 9.3132257e-10f;      // 2^-30
constexpr int kMiss = 0x7F7F7F7F;            // result sentinel (> any tet index)
constexpr int kMaxG = 96;
constexpr int kHitOverflow = -2;            // hits[.].w: this tet has accepted queries that are not recorded

// "hit record" buffer written by the forward and consumed by the backward (int32 words):
//   [0, 4*B*T)            int4 per tet: the (<= 4) queries the tet accepted, or w == kHitOverflow
//   [4*B*T, +pad)         per shape: number of uncovered queries (pad = B rounded up to kHitPad)
//   [.., + B*Q)           per shape: uncovered queries = hits that are NOT in their tet's record
//                         (tet overflowed / irregular tet / NaN-Inf-huge query)
constexpr int kHitPad = 64;
__host__ __device__ inline size_t hit_cnt_off(int B, int T) { return (size_t)B * T * 4; }
__host__ __device__ inline size_t hit_list_off(int B, int T) { return (size_t)B * T * 4 + (size_t)((B + kHitPad - 1) / kHitPad) * kHitPad; }
// Tets whose hit record overflowed (> 4 accepted queries; ~1e-4 of the tets at BASELINE configs[2]) are also LISTED, so
// that k_finalize can tell "this hit is not in its tet's record" from a short wave-uniform list instead of gathering the
// winning tet's 16-byte record for every query of the shape.  The list lives behind the counter block:
//   counters[0, 4B) counters | [4B, 8B) statistics ([.][2] = number of overflowed tets) | [8B, 8B + B*kOvfCap) the lists.
constexpr int kOvfCap = 128;
__device__ __forceinline__ void note_overflow(int *counters, int nB, int b, int t)
{
    counters[b * 4 + 2] = 1;                                           // some record overflowed (benign race: all write 1)
    const int k = atomicAdd(&counters[nB * 4 + b * 4 + 2], 1);
    if (k < kOvfCap) counters[nB * 8 + b * kOvfCap + k] = t;
}

#ifndef PIT_XFINE
#define PIT_XFINE 6
#endif
#ifndef PIT_GDIV
#define PIT_GDIV 6.0
#endif
constexpr int kXFine = PIT_XFINE;           // default: cells are kXFine times finer along x (the run direction)
constexpr int kMaxXFine = 8;                // upper bound of the runtime override DEFTET_PIT_XFINE (LDS sizing of k_row_fine)

// ------------------------------------------------------------------------------------
// exact predicate pieces (check_condition_tet_for.cu:105-121, :172-176)
// ------------------------------------------------------------------------------------
struct Planes {
    float n[4][3];   // (b-a) x (c-a) for the four vertex orderings
    float a[4][3];   // base vertex of each ordering (= vertex i)
    unsigned sv;     // bit i: dotv4_i > 0
    float dv[4];     // dotv4_i
};

__device__ __forceinline__ void make_planes(const float *v /*12*/, Planes &P)
{
    // orderings (a,b,c,d),(b,a,d,c),(c,d,a,b),(d,c,b,a): check_condition_tet_for.cu:172-175
    constexpr int ord[4][4] = {{0, 1, 2, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {3, 2, 1, 0}};
    P.sv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *a = v + 3 * ord[i][0], *b = v + 3 * ord[i][1], *c = v + 3 * ord[i][2], *d = v + 3 * ord[i][3];
        float r1x = b[0] - a[0], r1y = b[1] - a[1], r1z = b[2] - a[2];        // :111
        float r2x = c[0] - a[0], r2y = c[1] - a[1], r2z = c[2] - a[2];        // :112
        float nx = r1y * r2z - r1z * r2y;                                       // :63
        float ny = r1z * r2x - r1x * r2z;                                       // :64
        float nz = r1x * r2y - r1y * r2x;                                       // :65
        float dx = d[0] - a[0], dy = d[1] - a[1], dz = d[2] - a[2];            // :114
        float dotv4 = nx * dx + ny * dy + nz * dz;                              // :115
        P.n[i][0] = nx; P.n[i][1] = ny; P.n[i][2] = nz;
        P.a[i][0] = a[0]; P.a[i][1] = a[1]; P.a[i][2] = a[2];
        P.dv[i] = dotv4;
        P.sv |= (dotv4 > 0 ? 1u : 0u) << i;                                     // :119
    }
}

__device__ __forceinline__ bool accept(const Planes &P, float px, float py, float pz)
{
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float rx = px - P.a[i][0], ry = py - P.a[i][1], rz = pz - P.a[i][2];   // :116
        float dotp = P.n[i][0] * rx + P.n[i][1] * ry + P.n[i][2] * rz;          // :117
        m |= (dotp > 0 ? 1u : 0u) << i;                                          // :118
    }
    unsigned x = m ^ P.sv;          // bit i set <=> sign_p != sign_v  (:120)
    return x == 0u || x == 15u;     // all four equal (:176)
}

// ------------------------------------------------------------------------------------
// grid helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int f2ord(float f)
{
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

struct Grid {
    float o[3], inv[3], lo[3], hi[3];
};
constexpr int kBoxBlocks = 64;

// reduce the per-block query boxes of one shape into grid parameters; called by every wave of
// k_row_count (64 partials, a few shuffles) so that no separate launch is needed
__device__ __forceinline__ Grid reduce_grid(const float *__restrict__ part, int nPart, int G, int Gx)
{
    const int lane = threadIdx.x & 63;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lane; i < nPart; i += 64) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], part[(size_t)i * 6 + k]);
            hi[k] = fmaxf(hi[k], part[(size_t)i * 6 + 3 + k]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    Grid g;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = lo[k], h = hi[k];
        const bool ok = h >= l;                   // false when no regular query was seen
        const float ext = h - l;
        l = ok ? l : 0.f;
        h = ok ? h : 0.f;
        g.o[k] = l;
        g.inv[k] = (ok && ext > 1e-30f) ? (float)(k == 0 ? Gx : G) / ext : 0.f;   // cells per unit
        g.lo[k] = l;
        g.hi[k] = h;
    }
    return g;
}

__device__ __forceinline__ Grid load_grid(const float *__restrict__ gp)
{
    Grid g;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.o[k] = gp[k];
        g.inv[k] = gp[3 + k];
        g.lo[k] = gp[6 + k];
        g.hi[k] = gp[9 + k];
    }
    return g;
}

// monotone non-decreasing in x for fixed (o, inv >= 0): rounding, floor and clamp are monotone
__device__ __forceinline__ int cell_of(float x, float o, float inv, int G)
{
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.f), (float)(G - 1));
    return (int)f;
}

__device__ __forceinline__ bool query_regular(float x, float y, float z)
{
    return fabsf(x) <= kBig && fabsf(y) <= kBig && fabsf(z) <= kBig;   // NaN fails
}

// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
// query bounding box per shape (per-block partials, no contended atomics).
// part layout: [B][kBoxBlocks][6] floats (lo xyz, hi xyz); a block that saw no regular query
// writes (+inf, -inf).
// The same launch also clears what the later kernels accumulate into (counters, result
// sentinels): nothing in THIS kernel reads them.
__global__ __launch_bounds__(256) void k_query_bbox(const float *__restrict__ pts, int Q, float *part, int *counters,
                                                    int *result, int nB, long long nQ)
{
    __shared__ float sh[4][6];
    const int b = blockIdx.y;
    {
        const long long nblk = (long long)gridDim.x * gridDim.y, bid = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        const long long i = bid * blockDim.x + threadIdx.x, stride = nblk * blockDim.x;
        if (i < nB * 8) counters[i] = 0;                             // [0, 4B): counters; [4B, 8B): traversal statistics
        for (long long j = i; j < nQ; j += stride) result[j] = kMiss;
    }
    const float *p = pts + (size_t)b * Q * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += gridDim.x * blockDim.x) {
        float x = p[q * 3], y = p[q * 3 + 1], z = p[q * 3 + 2];
        if (query_regular(x, y, z)) {                              // irregular queries are listed by k_row_count
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sh[w][k] = lo[k]; sh[w][3 + k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, sh[i][k]) : fmaxf(v, sh[i][k]);
        part[((size_t)b * kBoxBlocks + blockIdx.x) * 6 + k] = v;
    }
}

// ------------------------------------------------------------------------------------
// Counting sort of the regular queries into grid cells WITHOUT global atomics.
// (Round-1 history: one returning global atomicAdd per query = 800 k fabric transactions = 41 us;
// random-address global atomics run at ~23-26 G/s chip-wide at ANY scope, tools/probes/.)
// Two levels, both with LDS atomics only:
//   rows  (cz*G+cy, <= 96^2): k_row_count (per-block LDS histogram + rank inside the block),
//         k_row_colscan (prefix over blocks per row), k_row_scatter (queries -> row order);
//   cells (cx inside a row, <= 384): k_row_fine, one wave per row (count, scan, place).
// The order of queries inside a cell is arbitrary (as it was with global atomics); nothing
// downstream depends on it except which four accepted queries a hit record keeps.
// ------------------------------------------------------------------------------------
constexpr int kMaxRowBlocks = 256;   // blocks per shape in k_row_count / k_row_scatter
#ifndef PIT_ROWTILE
#define PIT_ROWTILE 2048
#endif
constexpr int kRowTile = PIT_ROWTILE;  // smallest query chunk per block

__global__ __launch_bounds__(256) void k_row_count(const float *__restrict__ pts, int Q, const float *__restrict__ bboxPart,
                                                   float *gparam, int G, int Gx, int nblk, int chunkQ, int2 *qkey,
                                                   int *blockHist, int *counters, int *irregQ)
{
    __shared__ int hist[kMaxG * kMaxG];
    const int b = blockIdx.y, blk = blockIdx.x, R = G * G;
    const Grid g = reduce_grid(bboxPart + (size_t)b * kBoxBlocks * 6, kBoxBlocks, G, Gx);
    if (blk == 0 && threadIdx.x < 3) {                             // publish for k_row_fine / k_tet_scan
        const int k = threadIdx.x;
        gparam[b * 12 + k] = g.o[k]; gparam[b * 12 + 3 + k] = g.inv[k]; gparam[b * 12 + 6 + k] = g.lo[k]; gparam[b * 12 + 9 + k] = g.hi[k];
    }
    for (int i = threadIdx.x; i < R; i += 256) hist[i] = 0;
    __syncthreads();
    const int q0 = blk * chunkQ, q1 = min(Q, q0 + chunkQ);
    for (int qb = q0 + threadIdx.x; qb < q1; qb += 256 * 4) {
        int row[4];
        float3 pp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = qb + k * 256;
            if (q < q1) {
                const float *p = pts + ((size_t)b * Q + q) * 3;
                pp[k] = make_float3(p[0], p[1], p[2]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = qb + k * 256;
            if (q >= q1) continue;
            int rank = 0;
            if (query_regular(pp[k].x, pp[k].y, pp[k].z)) {
                row[k] = cell_of(pp[k].z, g.o[2], g.inv[2], G) * G + cell_of(pp[k].y, g.o[1], g.inv[1], G);
                rank = atomicAdd(&hist[row[k]], 1);                // LDS
            } else {                                               // NaN / Inf / huge: tested by every tet lane at the end of k_tet_scan
                row[k] = -1;
                irregQ[(size_t)b * Q + atomicAdd(&counters[b * 4 + 1], 1)] = q;
            }
            qkey[(size_t)b * Q + q] = make_int2(row[k], rank);
        }
    }
    __syncthreads();
    int *out = blockHist + ((size_t)b * nblk + blk) * R;
    for (int i = threadIdx.x; i < R; i += 256) out[i] = hist[i];
}

// per row: exclusive prefix of the block histograms over the blocks (in place) + row total
__global__ __launch_bounds__(256) void k_row_colscan(int *blockHist, int nblk, int R, int *rowTotal)
{
    const int b = blockIdx.y, row = blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    int *p = blockHist + (size_t)b * nblk * R + row;
    int run = 0, blk = 0;
    for (; blk + 8 <= nblk; blk += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(blk + k) * R];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            p[(size_t)(blk + k) * R] = run;
            run += v[k];
        }
    }
    for (; blk < nblk; ++blk) {
        const int v = p[(size_t)blk * R];
        p[(size_t)blk * R] = run;
        run += v;
    }
    rowTotal[(size_t)b * R + row] = run;
}

// queries -> row order.  Every block first rebuilds the row starts (exclusive scan of <= 9216
// row totals in LDS: cheaper than one more launch); block 0 of a shape publishes them.
__global__ __launch_bounds__(256) void k_row_scatter(const float *__restrict__ pts, int Q, const int2 *__restrict__ qkey,
                                                     const int *__restrict__ blockHist, const int *__restrict__ rowTotal,
                                                     int *rowStart, int G, int nblk, int chunkQ, float4 *rowSorted)
{
    __shared__ int rs[kMaxG * kMaxG + 1];
    __shared__ int wsum[4];
    const int b = blockIdx.y, blk = blockIdx.x, R = G * G;
    {
        const int per = (R + 255) / 256, r0 = threadIdx.x * per, r1 = min(R, r0 + per);
        const int *rt = rowTotal + (size_t)b * R;
        int sum = 0;
        for (int r = r0; r < r1; ++r) sum += rt[r];
        int incl = sum;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < w) run += wsum[k];
        for (int r = r0; r < r1; ++r) {
            rs[r] = run;
            run += rt[r];
        }
        if (r0 < R && r1 == R) rs[R] = run;
        __syncthreads();
        if (blk == 0)
            for (int i = threadIdx.x; i <= R; i += 256) rowStart[(size_t)b * (R + 1) + i] = rs[i];
    }
    const int *bh = blockHist + ((size_t)b * nblk + blk) * R;
    const int q0 = blk * chunkQ, q1 = min(Q, q0 + chunkQ);
    for (int q = q0 + threadIdx.x; q < q1; q += 256) {
        const int2 k = qkey[(size_t)b * Q + q];
        if (k.x < 0) continue;
        const float *p = pts + ((size_t)b * Q + q) * 3;
        const int pos = rs[k.x] + bh[k.x] + k.y;
        rowSorted[(size_t)b * Q + pos] = make_float4(p[0], p[1], p[2], __int_as_float(q));
    }
}

// one wave per row: count the row's queries per x-cell, scan, write the cell starts and place the
// queries in cell order
__global__ __launch_bounds__(256) void k_row_fine(const float4 *__restrict__ rowSorted, int Q, const float *__restrict__ gparam,
                                                  int G, int Gx, const int *__restrict__ rowStart, long long cellStride, int *cells,
                                                  float4 *sortedQ)
{
    __shared__ int cnt[4][kMaxG * kMaxXFine];
    const int b = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63, R = G * G;
    const int row = blockIdx.x * 4 + wv;
    const bool live = row < R;
    const float o = gparam[b * 12 + 0], inv = gparam[b * 12 + 3];
    int s = 0, e = 0;
    if (live) {
        s = rowStart[(size_t)b * (R + 1) + row];
        e = rowStart[(size_t)b * (R + 1) + row + 1];
    }
    for (int i = lane; i < Gx; i += 64) cnt[wv][i] = 0;
    __syncthreads();
    const float4 *src = rowSorted + (size_t)b * Q;
    for (int i = s + lane; i < e; i += 64) atomicAdd(&cnt[wv][cell_of(src[i].x, o, inv, Gx)], 1);
    __syncthreads();
    if (live) {
        const int per = (Gx + 63) / 64, c0 = lane * per, c1 = min(Gx, c0 + per);
        int sum = 0;
        for (int c = c0; c < c1; ++c) sum += cnt[wv][c];
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        int run = s + incl - sum;
        int *cb = cells + (size_t)b * cellStride + (size_t)row * Gx;
        for (int c = c0; c < c1; ++c) {
            const int n = cnt[wv][c];
            cnt[wv][c] = run;                                      // becomes the placement cursor
            cb[c] = run;
            run += n;
        }
        if (row == R - 1 && lane == 0) cb[Gx] = e;                 // end sentinel: cells has Gx*G*G+1 entries
    }
    __syncthreads();
    float4 *dst = sortedQ + (size_t)b * Q;
    for (int i = s + lane; i < e; i += 64) {
        const float4 q = src[i];
        dst[atomicAdd(&cnt[wv][cell_of(q.x, o, inv, Gx)], 1)] = q;
    }
}

// Irregular queries (NaN / Inf / |x| > 2^20: not in the grid; normally none) are tested by every
// tet lane against its own planes at the end of the tet kernels.
__device__ __forceinline__ void irregular_queries_tail(const Planes &P, int t, int b, int Q, const float *__restrict__ pts,
                                                       const int *__restrict__ counters, const int *__restrict__ irregQ, int *result)
{
    const int n = counters[b * 4 + 1];
    for (int k = 0; k < n; ++k) {
        const int q = irregQ[(size_t)b * Q + k];
        const float *p = pts + ((size_t)b * Q + q) * 3;
        if (accept(P, p[0], p[1], p[2])) atomicMin(&result[(size_t)b * Q + q], t);
    }
}

// The main kernel: one lane per tet, exact test inline.
// (A two-phase variant — ballot-compacted candidate ring in LDS + dense exact test — was built
// and measured in round 1: 14 % fewer VALU instructions but 1.5x slower, because the kernel is
// bound by vector-memory issue/latency, not by VALU: SQ_WAIT_ANY = 66 % of SQ_WAVE_CYCLES,
// ~55 gather instructions per wave.  A per-lane LDS candidate queue that defers the exact test so
// that it is issued max-over-lanes times per wave instead of once per slot was also measured:
// 133 us vs 114 us — same conclusion.  See DESIGN.md "A1 kernel anatomy" and profiles/.)
#ifndef PIT_WAVES
#define PIT_WAVES 6
#endif
#ifndef PIT_BATCH
#define PIT_BATCH 2
#endif
__global__ __launch_bounds__(256, PIT_WAVES) void k_tet_scan(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int4 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount)
{
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;   // uncovered-hit counter of the hit buffer (k_finalize appends)
    const int b = blockIdx.y;
    // XCD-aware mapping (workgroup i is observed to run on XCD i % 8, each XCD has a private
    // L2): give every XCD one CONTIGUOUS eighth of the tet range, so a mesh whose tet order is
    // spatially coherent makes each L2 pull only its own part of the sorted queries instead of
    // all eight pulling all of it (measured: 201 MB -> see profiles/).  Speed only; any
    // placement is correct.
    const int nblk = gridDim.x;
    const int per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int t = vb * blockDim.x + threadIdx.x;
    if (vb >= nblk || t >= T) return;
    float v[12];
    {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + t) * 12);
        float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
    Planes P;
    make_planes(v, P);
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
        hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
    }
    float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    // conditioning test; every comparison is written so that NaN yields "irregular"
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k]) <= kBig);
    float mn = fminf(fminf(fabsf(P.dv[0]), fabsf(P.dv[1])), fminf(fabsf(P.dv[2]), fabsf(P.dv[3])));
    bool regular = finite && (P.sv == 0u || P.sv == 15u) && (w >= kWMin) && (mn >= kTau * ((w * w) * w));
    // sv==0 with a zero dotv4 is excluded by mn >= tau*w^3 > 0
    // hits[b,t] (optional): the queries this tet ACCEPTED (up to 4; w == kHitOverflow marks "more
    // than fit / not recorded") — the backward filters them by condition == t, so no per-hit
    // atomics or linked lists are needed there.
    int4 hrec = make_int4(-1, -1, -1, -1);
    int hcnt = 0;
    if (!regular) {
        int k = atomicAdd(&counters[b * 4 + 0], 1);
        irregT[(size_t)b * T + k] = t;
        if (hits) hits[(size_t)b * T + t] = make_int4(-1, -1, -1, kHitOverflow);   // accepted by k_finalize, not recorded
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
        return;
    }
    const Grid g = load_grid(gparam + b * 12);
    const float m = w * kMargin;
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = lo[k] - m;
        ehi[k] = hi[k] + m;
    }
    // no regular query can lie in the enlarged box -> nothing to do
    if (ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]) {
        if (hits) hits[(size_t)b * T + t] = hrec;
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
        return;
    }
    const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
    const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    // Walk the (cz, cy) rows of the cell range; the x-run of a row is contiguous in sortedQ.
    // Latency hiding per lane: the next row's [start,end) is fetched before the current
    // row's queries are tested, and queries are fetched PIT_BATCH at a time (measured on the
    // BASELINE workload: 2 predicated = 104 us, 4 predicated = 135 us, 4 clamped = 113 us, 2 clamped = 118 us).
    auto test = [&](const float4 &q) {
        if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2]) {
            if (accept(P, q.x, q.y, q.z)) {
                const int qi = __float_as_int(q.w);
                atomicMin(&res[qi], t);
                if (hcnt == 0) hrec.x = qi;
                else if (hcnt == 1) hrec.y = qi;
                else if (hcnt == 2) hrec.z = qi;
                else if (hcnt == 3) hrec.w = qi;
                ++hcnt;
            }
        }
    };
    int cy = cy0, cz = cz0;
    auto bounds = [&](int row, int &s_, int &e_) {
        s_ = cb[row + cx0];
        e_ = cb[row + cx1 + 1];                         // cells has Gx*G*G+1 valid entries
    };
    int s, e;
    bounds((cz * G + cy) * Gx, s, e);
    for (;;) {
        int ny = cy + 1, nz = cz;
        if (ny > cy1) { ny = cy0; nz = cz + 1; }
        const bool more = nz <= cz1;
        int s2 = 0, e2 = 0;
        if (more) bounds((nz * G + ny) * Gx, s2, e2);
        for (int j = s; j < e; j += PIT_BATCH) {
            float4 qq[PIT_BATCH];
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k)
                if (k == 0 || j + k < e) qq[k] = sq[j + k];     // no clamped duplicate gathers: lane-gathers are the cost
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k)
                if (k == 0 || j + k < e) test(qq[k]);
        }
        if (!more) break;
        s = s2; e = e2; cy = ny; cz = nz;
    }
    if (hits) {
        if (hcnt > 4) {
            hrec.w = kHitOverflow;
            note_overflow(counters, gridDim.y, b, t);
        }
        hits[(size_t)b * T + t] = hrec;
    }
    irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
}

// ------------------------------------------------------------------------------------
// k_tet_scan_fma (DEFTET_PIT_FMA / DEFTET_PIT_FMA2): same traversal as k_tet_scan, but the per-candidate work
// — box test (6 compares) + exact predicate (4 x [3 sub, 3 mul, 2 add, 1 cmp]) — is replaced by a CERTIFIED
// fused filter: one 3-FMA chain per face plane, one min over the four, two compares.  The exact predicate runs
// only for candidates the filter cannot decide (a band of a few fp32 ulps around the face planes: ~1e-4 of the
// candidates on the BASELINE workload), so the result is still bit-exact.  k_tet_scan is bound by VALU issue
// (44.8 M wave-instructions x 4 cycles = 83 us of its 105 us; profiles/r01_pmc_k_tet_scan_variants.json), and
// ~80 % of those instructions are the per-candidate tests.
//
// For a regular tet (see k_tet_scan) with sigma = sign of its four dotv4, the reference accepts p iff
//     sigma * dotp_i(p) > 0  for i = 0..3        (dotp_i = fl(n_i . fl(p - a_i)), source operation order)
// ("all four false" cannot happen for regular tets).  With D_i = n_i . (p - a_i) in exact
// arithmetic over the COMPUTED normals:  |dotp_i - D_i| <= 4.0001 u * sum_k |n_ik| (|p_k| + |a_ik|),  u = 2^-24.
// The filter evaluates  A_i = fma(N_i0, x, fma(N_i1, y, fma(N_i2, z, C_i))),  N_i = sigma n_i,
// C_i = fl(-sigma c_i - E_i),  c_i = fma(n_i0, a_i0, fma(n_i1, a_i1, n_i2 a_i2)),  which equals
// sigma D_i - E_i up to  3u sum|n||p| + 7u sum|n||a| + 4u E_i.  With
//     E_i = 16 u * sum_k |n_ik| (P_k + 2 M_k) + 2^-120,   P_k >= |p_k| for every regular query (grid box),
//                                                         M_k >= |vertex coordinate k| of this tet,
// E_i exceeds the sum of both error bounds (7u + 4u on |p|, 11u + 4u... on |a|, with a factor ~2 to spare for the
// fp32 evaluation of E_i itself), hence
//     min_i A_i > 0              =>  every sigma * dotp_i > 0        =>  the reference accepts   (certain)
//     min_i A_i < -2 max_i E_i   =>  some  sigma * dotp_j < 0        =>  the reference rejects   (certain)
// and anything in between is handed to the exact predicate.  No box test is needed: a point outside the tet
// violates at least one plane.  2^-120 absorbs products that underflow in either evaluation.
// PACKED: two candidates per instruction (v_pk_fma_f32).
// ------------------------------------------------------------------------------------
constexpr float kErrScale = 9.5367431640625e-07f;      // 16 u = 2^-20
constexpr float kErrAbs = 7.5231638e-37f;              // 2^-120

struct Filter {
    float N[4][3];
    float C[4];
    float twoEmax;
};

// Select with the lane mask in an SGPR pair (VOP3 v_cndmask_b32_e64).  hipcc likes to shrink selects whose mask sits in
// VCC to the VOP2 form `v_cndmask_b32_e32 …, vcc`, which gfx950 issues ~7.5x slower than an FMA (9.4 vs 1.25 ns per
// wave-instruction per SIMD, tools/probes/valu_rate_probe.hip; the SGPR-pair form: 1.85 ns).  The traversal loops carry
// ten selects per iteration, so the form matters more than the count.
typedef unsigned long long lanemask_t;
__device__ __forceinline__ int sel(lanemask_t m, int if_set, int if_clear)
{
    int d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(m));
    return d;
}
__device__ __forceinline__ unsigned sel(lanemask_t m, unsigned if_set, unsigned if_clear)
{
    return (unsigned)sel(m, (int)if_set, (int)if_clear);
}
__device__ __forceinline__ lanemask_t mask_of(bool c) { return __builtin_amdgcn_ballot_w64(c); }

#ifdef PIT_PHASE_TIMING
// Diagnostic build only (tools/probes/build_variant.sh … -DPIT_PHASE_TIMING): per-phase wall cycles of the traversal
// kernels.  Lane 0 of every wave adds its s_memtime deltas to a slot of its own (no atomics: same-address atomics from
// 32 k waves would dominate what is being measured); deftet_debug_phase_read sums the slots.
constexpr int kPhaseWaves = 1 << 16;
__device__ unsigned long long g_phase[kPhaseWaves][16];
#define PHASE_DECL                                                                                      \
    long long ph_t_ = clock64();                                                                        \
    const unsigned ph_w_ = (((unsigned)blockIdx.y * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6)) & (kPhaseWaves - 1)
#define PHASE_MARK(i)                                                                  \
    do {                                                                               \
        const long long ph_n_ = clock64();                                             \
        if ((threadIdx.x & 63) == 0) g_phase[ph_w_][i] += (unsigned long long)(ph_n_ - ph_t_); \
        ph_t_ = ph_n_;                                                                 \
    } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(i)
#endif

// load at a 32-bit unsigned BYTE offset from a (wave-uniform) base pointer: scalar-base + vector-offset addressing
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Exact re-scan of ONE tet's candidates (box test + reference predicate, as k_tet_scan does), used by k_tet_scan_fma for
// the rare tets whose traversal met the filter's undecided band or more than four acceptances.  Out of line and called
// AFTER the traversal loop, so that nothing of it is scheduled (or kept in registers) inside the loop.  Publishes every
// accepted query with atomicMin (idempotent w.r.t. the ones the filter already accepted) and returns the hit record.
__device__ __noinline__ int4 exact_rescan(const float *__restrict__ tv, int t, const int *__restrict__ cb, const float4 *__restrict__ sq,
                                          int *res, int G, int Gx, int cx0, int cx1, int cy0, int cy1, int cz0, int cz1,
                                          float m, int *counters, int nB, int b)
{
    float vv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) vv[k] = tv[k];
    Planes P;
    make_planes(vv, P);
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = fminf(fminf(vv[k], vv[3 + k]), fminf(vv[6 + k], vv[9 + k])) - m;
        ehi[k] = fmaxf(fmaxf(vv[k], vv[3 + k]), fmaxf(vv[6 + k], vv[9 + k])) + m;
    }
    int4 hrec = make_int4(-1, -1, -1, -1);
    int hcnt = 0;
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) {
            const int row = (cz * G + cy) * Gx;
            const int s = cb[row + cx0], e = cb[row + cx1 + 1];
            for (int j = s; j < e; ++j) {
                const float4 q = sq[j];
                if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2] &&
                    accept(P, q.x, q.y, q.z)) {
                    const int qi = __float_as_int(q.w);
                    atomicMin(&res[qi], t);
                    if (hcnt == 0) hrec.x = qi;
                    else if (hcnt == 1) hrec.y = qi;
                    else if (hcnt == 2) hrec.z = qi;
                    else if (hcnt == 3) hrec.w = qi;
                    ++hcnt;
                }
            }
        }
    if (hcnt > 4) {
        hrec.w = kHitOverflow;
        note_overflow(counters, nB, b, t);
    }
    return hrec;
}

// irregular queries (NaN / Inf / huge; normally none): re-load the tet so that its vertices need not stay in
// registers across the traversal loop
__device__ __noinline__ void fma_irregular_tail_slow(const float *__restrict__ tet, int t, int b, int T, int Q, const float *__restrict__ pts,
                                                     const int *__restrict__ counters, const int *__restrict__ irregQ, int *result)
{
    float v[12];
    const float *src = tet + ((size_t)b * T + t) * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = src[k];
    Planes P;
    make_planes(v, P);
    irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
}
__device__ __forceinline__ void fma_irregular_tail(const float *__restrict__ tet, int t, int b, int T, int Q, const float *__restrict__ pts,
                                                   const int *__restrict__ counters, const int *__restrict__ irregQ, int *result)
{
    if (counters[b * 4 + 1] > 0) fma_irregular_tail_slow(tet, t, b, T, Q, pts, counters, irregQ, result);
}

template <bool PACKED>
__global__ __launch_bounds__(256, PIT_WAVES) void k_tet_scan_fma(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int4 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount, const int *__restrict__ list)
{
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;
    const int b = blockIdx.y;
    const int nblk = gridDim.x;
    const int per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware mapping, see k_tet_scan
    int t = vb * blockDim.x + threadIdx.x;
    if (vb >= nblk || t >= T) return;
    PHASE_DECL;
    if (list) {                                                        // list mode: the tets k_tet_scan_grp deferred (count in counters[.][3])
        if (t >= counters[b * 4 + 3]) return;
        t = list[(size_t)b * T + t];
    }
    float v[12];
    {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + t) * 12);
        float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
        hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
    }
    const float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    Filter F;
    unsigned sv;
    float sigma;
    bool regular;
    const Grid g = load_grid(gparam + b * 12);
    {
        Planes P;
        make_planes(v, P);
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k]) <= kBig);
        const float mn = fminf(fminf(fabsf(P.dv[0]), fabsf(P.dv[1])), fminf(fabsf(P.dv[2]), fabsf(P.dv[3])));
        regular = finite && (P.sv == 0u || P.sv == 15u) && (w >= kWMin) && (mn >= kTau * ((w * w) * w));
        sv = P.sv;
        sigma = P.sv == 15u ? 1.0f : -1.0f;
        if (!regular) {
            int k = atomicAdd(&counters[b * 4 + 0], 1);
            irregT[(size_t)b * T + k] = t;
            if (hits) hits[(size_t)b * T + t] = make_int4(-1, -1, -1, kHitOverflow);
            irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
            return;
        }
        float S[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            S[k] = kErrScale * (fmaxf(fabsf(g.lo[k]), fabsf(g.hi[k])) + 2.0f * fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        float emax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n0 = P.n[i][0], n1 = P.n[i][1], n2 = P.n[i][2];
            const float c = fmaf(n0, P.a[i][0], fmaf(n1, P.a[i][1], n2 * P.a[i][2]));
            const float E = fmaf(fabsf(n0), S[0], fmaf(fabsf(n1), S[1], fabsf(n2) * S[2])) + kErrAbs;
            F.N[i][0] = sigma * n0; F.N[i][1] = sigma * n1; F.N[i][2] = sigma * n2;
            F.C[i] = -sigma * c - E;
            emax = fmaxf(emax, E);
        }
        F.twoEmax = 2.0f * emax;
    }
    const float m = w * kMargin;
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = lo[k] - m;
        ehi[k] = hi[k] + m;
    }
    int hcnt = 0;
    if (ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]) {
        if (hits) hits[(size_t)b * T + t] = make_int4(-1, -1, -1, -1);
        fma_irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
        return;
    }
    const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
    const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    PHASE_MARK(0);                                                       // [0] load + setup
    // The loop body is branch-free.  Accepted queries go into a four-deep shift register (h0 = newest) and are
    // published with atomicMin once, after the traversal.  A candidate in the filter's undecided band, or a fifth
    // acceptance, only raises a flag; such tets (~1e-4 of them) are re-scanned exactly afterwards.
    int h0 = -1, h1 = -1, h2 = -1, h3 = -1;
    float amin = INFINITY;                                             // smallest |filter value| met: <= 2 Emax <=> the undecided band was touched
    auto decide = [&](float a, int qi, bool live) {
        const lanemask_t acc = mask_of(live && a > 0.f);
        amin = fminf(amin, fabsf(a));                                  // a dead second slot repeats the first candidate: no mask needed
        h3 = sel(acc, h2, h3);
        h2 = sel(acc, h1, h2);
        h1 = sel(acc, h0, h1);
        h0 = sel(acc, qi, h0);
        hcnt += sel(acc, 1, 0);
    };
    // per-lane cursor over (row, position), see k_tet_scan_grp: every wave-iteration each lane takes ITS next two candidates.
    // All addresses are 32-bit BYTE offsets from wave-uniform bases (scalar base + vector offset addressing: no 64-bit
    // address arithmetic in the loop); the row offset advances by additions (no integer multiply).
    const unsigned rowStepB = (unsigned)Gx * 4u;                                        // next cy
    const unsigned rowWrapB = (unsigned)((G - (cy1 - cy0)) * Gx) * 4u;                 // cy wraps to cy0, cz + 1
    const unsigned x0B = (unsigned)cx0 * 4u, x1B = (unsigned)(cx1 + 1) * 4u;
    unsigned rowB = (unsigned)((cz0 * G + cy0) * Gx) * 4u;                              // the row whose bounds sit in (s2, e2)
    int cy = cy0, cz = cz0;
    int j = 0, e = 0;
    int s2 = ld_off<int>(cb, rowB + x0B), e2 = ld_off<int>(cb, rowB + x1B);
    bool haveNext = true;
    while (j < e || haveNext) {
        if (j >= e) {                                                   // enter the prefetched row, prefetch the one after it
            j = s2;
            e = e2;
            const lanemask_t wrap = mask_of(cy == cy1);
            cy = sel(wrap, cy0, cy + 1);
            cz += sel(wrap, 1, 0);
            rowB += sel(wrap, rowWrapB, rowStepB);
            haveNext = cz <= cz1;
            if (haveNext) {
                s2 = ld_off<int>(cb, rowB + x0B);
                e2 = ld_off<int>(cb, rowB + x1B);
            }
        }
        if (j < e) {
            const bool two = j + 1 < e;
            const float4 q0 = ld_off<float4>(sq, (unsigned)j * 16u);
            const float4 q1 = ld_off<float4>(sq, (unsigned)sel(mask_of(two), j + 1, j) * 16u);   // dead slot: the same candidate again (never recorded)
            if constexpr (PACKED) {
                const f32x2 X = {q0.x, q1.x}, Y = {q0.y, q1.y}, Z = {q0.z, q1.z};
                f32x2 A[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 n0 = {F.N[i][0], F.N[i][0]}, n1 = {F.N[i][1], F.N[i][1]}, n2 = {F.N[i][2], F.N[i][2]}, cc = {F.C[i], F.C[i]};
                    A[i] = __builtin_elementwise_fma(n0, X, __builtin_elementwise_fma(n1, Y, __builtin_elementwise_fma(n2, Z, cc)));
                }
                const float a0 = fminf(fminf(A[0].x, A[1].x), fminf(A[2].x, A[3].x));
                const float a1 = fminf(fminf(A[0].y, A[1].y), fminf(A[2].y, A[3].y));
                decide(a0, __float_as_int(q0.w), true);
                decide(a1, __float_as_int(q1.w), two);
            } else {
                float a0, a1;
                {
                    float A[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) A[i] = fmaf(F.N[i][0], q0.x, fmaf(F.N[i][1], q0.y, fmaf(F.N[i][2], q0.z, F.C[i])));
                    a0 = fminf(fminf(A[0], A[1]), fminf(A[2], A[3]));
                }
                {
                    float A[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) A[i] = fmaf(F.N[i][0], q1.x, fmaf(F.N[i][1], q1.y, fmaf(F.N[i][2], q1.z, F.C[i])));
                    a1 = fminf(fminf(A[0], A[1]), fminf(A[2], A[3]));
                }
                decide(a0, __float_as_int(q0.w), true);
                decide(a1, __float_as_int(q1.w), two);
            }
            j += 2;
        }
    }
    PHASE_MARK(1);                                                       // [1] traversal loop
    if (amin <= F.twoEmax || hcnt > 4) {
        atomicAdd(&counters[gridDim.y * 4 + b * 4 + 1], 1);                   // statistics: tets re-scanned
        const int4 r = exact_rescan(tet + ((size_t)b * T + t) * 12, t, cb, sq, res, G, Gx, cx0, cx1, cy0, cy1, cz0, cz1, m,
                                    counters, gridDim.y, b);
        if (hits) hits[(size_t)b * T + t] = r;
    } else {
        if (hcnt > 0) atomicMin(&res[h0], t);
        if (hcnt > 1) atomicMin(&res[h1], t);
        if (hcnt > 2) atomicMin(&res[h2], t);
        if (hcnt > 3) atomicMin(&res[h3], t);
        if (hits) hits[(size_t)b * T + t] = make_int4(h0, h1, h2, h3);
    }
    fma_irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
    PHASE_MARK(2);                                                       // [2] publish (atomics, record store) / re-scan
}

// ------------------------------------------------------------------------------------
// k_tet_scan_grp<NP> (DEFTET_PIT_GRP2/4/6): K = 2*NP CONSECUTIVE tets per lane share one candidate stream.
//
// What bounds the one-tet-per-lane kernels is the divergent-gather path: every lane fetches its own row bounds and
// its own candidate queries, ~20 lane-requests per tet at BASELINE configs[2], and the texture-address unit retires
// about one divergent lane-request per clock per CU (measured: traversal time = requests / (256 CU x ~2 GHz) within
// 20 % at configs[1..3]; cutting the VALU work by 40 % with the fused filter moved the time by 7 %,
// profiles/r02_scan_variants_fma_sweep.jsonl).  Consecutive tets of a mesh are almost always neighbours (the six
// Kuhn tets of a cube have the same bounding box; in the shipped QuarTet grid the median pair of consecutive tets
// needs 1.25x the cells of one), so a lane that owns K consecutive tets walks the UNION of their cell ranges once
// and tests every fetched candidate against all K tets: requests per tet drop by up to K.
// The K-tet test is where packed fp32 pays: two tets sit in the two halves of a register pair, so the fused filter
// (see k_tet_scan_fma) costs 12 v_pk_fma_f32 per candidate per PAIR of tets, and the whole per-tet setup (planes,
// conditioning test, filter coefficients) runs as packed arithmetic too — with the same operation order and
// rounding per half, so every decision is bit-identical to the one-tet kernels.
//
// Accepted (candidate, tet-set) pairs go into an 8-deep shift register per lane (query id | K-bit tet mask); after
// the traversal each entry is published with ONE atomicMin (the lowest accepting tet of the group is all that can
// win) and decoded into the per-tet hit records.  Groups whose union range has more cells than the sum of its
// members' ranges (mesh-order jumps; ~2 % of the groups of the shipped grid) are appended to a list that a second,
// one-tet-per-lane launch of k_tet_scan_fma processes.  Candidates in the filter's undecided band, or more than
// eight accepting candidates, send the group to the exact re-scan.
// ------------------------------------------------------------------------------------
constexpr int kGrpDepth = 8;

struct TetCells { int cx0, cx1, cy0, cy1, cz0, cz1; };

// Setup of one PAIR of tets in packed arithmetic (.x = first tet, .y = second).  Returns per half: regular, active
// (regular + live + box meets the query grid), cell range; fills the filter coefficients (zero / -inf for halves that
// must never accept) and raises twoE to 2 * max E.
__device__ __forceinline__ void pair_setup(const float *__restrict__ recA, const float *__restrict__ recB, bool liveA, bool liveB,
                                           const Grid &g, int G, int Gx, f32x2 (&N)[4][3], f32x2 (&C)[4], float &twoE,
                                           bool (&regular)[2], bool (&active)[2], TetCells (&cells)[2])
{
    f32x2 v[12];
    {
        const float4 *sa = reinterpret_cast<const float4 *>(recA), *sb = reinterpret_cast<const float4 *>(recB);
        const float4 a0 = sa[0], a1 = sa[1], a2 = sa[2], b0 = sb[0], b1 = sb[1], b2 = sb[2];
        v[0] = f32x2{a0.x, b0.x}; v[1] = f32x2{a0.y, b0.y}; v[2] = f32x2{a0.z, b0.z}; v[3] = f32x2{a0.w, b0.w};
        v[4] = f32x2{a1.x, b1.x}; v[5] = f32x2{a1.y, b1.y}; v[6] = f32x2{a1.z, b1.z}; v[7] = f32x2{a1.w, b1.w};
        v[8] = f32x2{a2.x, b2.x}; v[9] = f32x2{a2.y, b2.y}; v[10] = f32x2{a2.z, b2.z}; v[11] = f32x2{a2.w, b2.w};
    }
    constexpr int ord[4][4] = {{0, 1, 2, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {3, 2, 1, 0}};    // check_condition_tet_for.cu:172-175
    f32x2 n[4][3], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 *a = v + 3 * ord[i][0], *b = v + 3 * ord[i][1], *c = v + 3 * ord[i][2], *d = v + 3 * ord[i][3];
        const f32x2 r1x = b[0] - a[0], r1y = b[1] - a[1], r1z = b[2] - a[2];               // :111
        const f32x2 r2x = c[0] - a[0], r2y = c[1] - a[1], r2z = c[2] - a[2];               // :112
        n[i][0] = r1y * r2z - r1z * r2y;                                                     // :63
        n[i][1] = r1z * r2x - r1x * r2z;                                                     // :64
        n[i][2] = r1x * r2y - r1y * r2x;                                                     // :65
        const f32x2 dx = d[0] - a[0], dy = d[1] - a[1], dz = d[2] - a[2];                   // :114
        dv[i] = n[i][0] * dx + n[i][1] * dy + n[i][2] * dz;                                  // :115
    }
    f32x2 lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = __builtin_elementwise_min(__builtin_elementwise_min(v[k], v[3 + k]), __builtin_elementwise_min(v[6 + k], v[9 + k]));
        hi[k] = __builtin_elementwise_max(__builtin_elementwise_max(v[k], v[3 + k]), __builtin_elementwise_max(v[6 + k], v[9 + k]));
    }
    const f32x2 w = __builtin_elementwise_max(__builtin_elementwise_max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    const f32x2 mn = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_abs(dv[0]), __builtin_elementwise_abs(dv[1])),
                                               __builtin_elementwise_min(__builtin_elementwise_abs(dv[2]), __builtin_elementwise_abs(dv[3])));
    const f32x2 thr = kTau * ((w * w) * w);
    const f32x2 mg = w * kMargin;
    f32x2 sigma, S[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f32x2 M = __builtin_elementwise_max(__builtin_elementwise_abs(lo[k]), __builtin_elementwise_abs(hi[k]));
        S[k] = kErrScale * (fmaxf(fabsf(g.lo[k]), fabsf(g.hi[k])) + 2.0f * M);
    }
    const bool live[2] = {liveA, liveB};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k][h]) <= kBig);            // NaN fails
        const unsigned sv = (dv[0][h] > 0 ? 1u : 0u) | (dv[1][h] > 0 ? 2u : 0u) | (dv[2][h] > 0 ? 4u : 0u) | (dv[3][h] > 0 ? 8u : 0u);   // :119
        regular[h] = finite && (sv == 0u || sv == 15u) && (w[h] >= kWMin) && (mn[h] >= thr[h]);
        sigma[h] = sv == 15u ? 1.0f : -1.0f;
        const float elo0 = lo[0][h] - mg[h], ehi0 = hi[0][h] + mg[h], elo1 = lo[1][h] - mg[h], ehi1 = hi[1][h] + mg[h],
                    elo2 = lo[2][h] - mg[h], ehi2 = hi[2][h] + mg[h];
        const bool ingrid = !(ehi0 < g.lo[0] || elo0 > g.hi[0] || ehi1 < g.lo[1] || elo1 > g.hi[1] || ehi2 < g.lo[2] || elo2 > g.hi[2]);
        active[h] = live[h] && regular[h] && ingrid;
        cells[h].cx0 = cell_of(elo0, g.o[0], g.inv[0], Gx); cells[h].cx1 = cell_of(ehi0, g.o[0], g.inv[0], Gx);
        cells[h].cy0 = cell_of(elo1, g.o[1], g.inv[1], G);  cells[h].cy1 = cell_of(ehi1, g.o[1], g.inv[1], G);
        cells[h].cz0 = cell_of(elo2, g.o[2], g.inv[2], G);  cells[h].cz1 = cell_of(ehi2, g.o[2], g.inv[2], G);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 *a = v + 3 * ord[i][0];
        const f32x2 c = __builtin_elementwise_fma(n[i][0], a[0], __builtin_elementwise_fma(n[i][1], a[1], n[i][2] * a[2]));
        const f32x2 E = __builtin_elementwise_fma(__builtin_elementwise_abs(n[i][0]), S[0],
                                                  __builtin_elementwise_fma(__builtin_elementwise_abs(n[i][1]), S[1],
                                                                            __builtin_elementwise_abs(n[i][2]) * S[2])) + kErrAbs;
        f32x2 Nx = sigma * n[i][0], Ny = sigma * n[i][1], Nz = sigma * n[i][2];
        f32x2 Cc = -sigma * c - E;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // inactive half: never accepts, never "undecided" (selects, not branches)
            Nx[h] = active[h] ? Nx[h] : 0.f; Ny[h] = active[h] ? Ny[h] : 0.f; Nz[h] = active[h] ? Nz[h] : 0.f;
            Cc[h] = active[h] ? Cc[h] : -INFINITY;
            twoE = fmaxf(twoE, active[h] ? 2.0f * E[h] : 0.f);
        }
        N[i][0] = Nx; N[i][1] = Ny; N[i][2] = Nz; C[i] = Cc;
    }
}

// exact re-scan of one tet with its OWN cell range (for the grouped kernel's rare slow path)
__device__ __noinline__ void exact_rescan_tet(const float *__restrict__ tet, int t, int b, int T, const float *__restrict__ gparam, int G, int Gx,
                                              const int *__restrict__ cb, const float4 *__restrict__ sq, int *res, int4 *hits, int *counters)
{
    const float *tv = tet + ((size_t)b * T + t) * 12;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(tv[k], tv[3 + k]), fminf(tv[6 + k], tv[9 + k]));
        hi[k] = fmaxf(fmaxf(tv[k], tv[3 + k]), fmaxf(tv[6 + k], tv[9 + k]));
    }
    const float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    const float m = w * kMargin;
    const Grid g = load_grid(gparam + b * 12);
    const int cx0 = cell_of(lo[0] - m, g.o[0], g.inv[0], Gx), cx1 = cell_of(hi[0] + m, g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(lo[1] - m, g.o[1], g.inv[1], G), cy1 = cell_of(hi[1] + m, g.o[1], g.inv[1], G);
    const int cz0 = cell_of(lo[2] - m, g.o[2], g.inv[2], G), cz1 = cell_of(hi[2] + m, g.o[2], g.inv[2], G);
    const int4 r = exact_rescan(tv, t, cb, sq, res, G, Gx, cx0, cx1, cy0, cy1, cz0, cz1, m, counters, gridDim.y, b);
    if (hits) hits[(size_t)b * T + t] = r;
}

template <int NP>
__global__ __launch_bounds__(256, NP == 1 ? 5 : (NP == 2 ? 4 : 3)) void k_tet_scan_grp(
    const float *__restrict__ tet, int T, int Q, const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
    long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters, int *irregT, int4 *hits,
    const float *__restrict__ pts, const int *__restrict__ irregQ, int *ucount, int *deferT)
{
    constexpr int K = 2 * NP;
    constexpr int kShift = 32 - K;                          // entry = query id | tet mask << kShift   (host guarantees Q < 2^kShift)
    constexpr unsigned kIdMask = (1u << kShift) - 1u;
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;
    const int b = blockIdx.y;
    const int nblk = gridDim.x, per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware mapping, see k_tet_scan
    const long long grp = (long long)vb * blockDim.x + threadIdx.x;
    const long long t0l = grp * K;
    if (vb >= nblk || t0l >= T) return;
    const int t0 = (int)t0l;
    const Grid g = load_grid(gparam + b * 12);
    f32x2 N[NP][4][3], C[NP][4];
    float twoE = 0.f;
    unsigned regM = 0, actM = 0, liveM = 0;                 // bit k: tet t0 + k is regular / active / exists
    int ux0 = 0x7FFFFFFF, ux1 = -1, uy0 = 0x7FFFFFFF, uy1 = -1, uz0 = 0x7FFFFFFF, uz1 = -1;
    long long vsum = 0;
    int nact = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int tA = t0 + 2 * p, tB = tA + 1;
        const bool liveA = tA < T, liveB = tB < T;
        const float *recA = tet + ((size_t)b * T + (liveA ? tA : t0)) * 12, *recB = tet + ((size_t)b * T + (liveB ? tB : t0)) * 12;
        bool r2[2], a2[2];
        TetCells c2[2];
        pair_setup(recA, recB, liveA, liveB, g, G, Gx, N[p], C[p], twoE, r2, a2, c2);
        const bool lv[2] = {liveA, liveB};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = 2 * p + h;
            if (lv[h]) liveM |= 1u << k;
            if (lv[h] && r2[h]) regM |= 1u << k;
            if (a2[h]) {
                actM |= 1u << k;
                ux0 = min(ux0, c2[h].cx0); ux1 = max(ux1, c2[h].cx1);
                uy0 = min(uy0, c2[h].cy0); uy1 = max(uy1, c2[h].cy1);
                uz0 = min(uz0, c2[h].cz0); uz1 = max(uz1, c2[h].cz1);
                vsum += (long long)(c2[h].cx1 - c2[h].cx0 + 1) * (c2[h].cy1 - c2[h].cy0 + 1) * (c2[h].cz1 - c2[h].cz0 + 1);
                ++nact;
            }
        }
    }
    // irregular tets (normally none): listed for k_finalize's brute-force pass, never recorded
    if (regM != liveM) {
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            if (((liveM & ~regM) >> k) & 1u) {
                const int i = atomicAdd(&counters[b * 4 + 0], 1);
                irregT[(size_t)b * T + i] = t0 + k;
                if (hits) hits[(size_t)b * T + t0 + k] = make_int4(-1, -1, -1, kHitOverflow);
            }
        }
    }
    // regular members whose box misses the query grid: empty record
    if (hits && (regM & ~actM)) {
#pragma unroll 1
        for (int k = 0; k < K; ++k)
            if (((regM & ~actM) >> k) & 1u) hits[(size_t)b * T + t0 + k] = make_int4(-1, -1, -1, -1);
    }
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    int e0 = -1, e1 = -1, e2 = -1, e3 = -1, e4 = -1, e5 = -1, e6 = -1, e7 = -1;
    int cnt = 0, nslow = 0;
    bool traversed = false;
    if (nact > 0) {
        const long long vu = (long long)(ux1 - ux0 + 1) * (uy1 - uy0 + 1) * (uz1 - uz0 + 1);
        if (vu > vsum) {
            // a jump in the mesh order inside this group: its members go to the one-tet-per-lane pass
            const int base = atomicAdd(&counters[b * 4 + 3], nact);
            int o = 0;
#pragma unroll 1
            for (int k = 0; k < K; ++k)
                if ((actM >> k) & 1u) deferT[(size_t)b * T + base + (o++)] = t0 + k;
        } else {
            traversed = true;
            auto test = [&](const float4 &q, bool live) {
                unsigned bits = 0;
                bool unc = false;
                const f32x2 X = {q.x, q.x}, Y = {q.y, q.y}, Z = {q.z, q.z};
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    f32x2 A[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        A[i] = __builtin_elementwise_fma(N[p][i][0], X, __builtin_elementwise_fma(N[p][i][1], Y,
                                                                                                    __builtin_elementwise_fma(N[p][i][2], Z, C[p][i])));
                    const float ax = fminf(fminf(A[0].x, A[1].x), fminf(A[2].x, A[3].x));
                    const float ay = fminf(fminf(A[0].y, A[1].y), fminf(A[2].y, A[3].y));
                    const bool accx = ax > 0.f, accy = ay > 0.f;
                    unc = unc | ((!accx) & (ax >= -twoE)) | ((!accy) & (ay >= -twoE));   // bitwise on purpose: no branches in this loop
                    bits = (bits << 2) | (accx ? 2u : 0u) | (accy ? 1u : 0u);           // first tet of the group = most significant bit
                }
                nslow += (live & unc) ? 1 : 0;
                const bool push = live & (bits != 0u);
                const int entry = (int)(((unsigned)__float_as_int(q.w) & kIdMask) | (bits << kShift));
                e7 = push ? e6 : e7; e6 = push ? e5 : e6; e5 = push ? e4 : e5; e4 = push ? e3 : e4;
                e3 = push ? e2 : e3; e2 = push ? e1 : e2; e1 = push ? e0 : e1; e0 = push ? entry : e0;
                cnt += push ? 1 : 0;
            };
            // Per-lane cursor over (row, position): in every wave-iteration each lane takes ITS next two candidates,
            // wherever they are — a lane whose run is exhausted enters its next row while the others keep testing.
            // (Looping row by row instead makes the wave spend max-over-lanes iterations on EVERY row: ~30 wave-
            // iterations for ~6.5 per lane at configs[2], i.e. 20 % lane utilisation and 30 dependent gather round trips.)
            int cy = uy0, cz = uz0;                                   // the row whose bounds sit in (s2, e2r)
            int j = 0, e = 0;
            int s2 = cb[(cz * G + cy) * Gx + ux0], e2r = cb[(cz * G + cy) * Gx + ux1 + 1];
            bool haveNext = true;
            while (j < e || haveNext) {
                if (j >= e) {                                         // enter the prefetched row, prefetch the one after it
                    j = s2;
                    e = e2r;
                    ++cy;
                    if (cy > uy1) { cy = uy0; ++cz; }
                    haveNext = cz <= uz1;
                    if (haveNext) {
                        const int row2 = (cz * G + cy) * Gx;
                        s2 = cb[row2 + ux0];
                        e2r = cb[row2 + ux1 + 1];
                    }
                }
                if (j < e) {
                    const bool two = j + 1 < e;
                    const float4 q0 = sq[j];
                    float4 q1;
                    q1.x = __builtin_nondeterministic_value(q0.x); q1.y = __builtin_nondeterministic_value(q0.y);
                    q1.z = __builtin_nondeterministic_value(q0.z); q1.w = __builtin_nondeterministic_value(q0.w);
                    if (two) q1 = sq[j + 1];
                    test(q0, true);
                    test(q1, two);
                    j += 2;
                }
            }
        }
    }
    if (traversed) {
        if (nslow > 0 || cnt > kGrpDepth) {
            // undecided band met, or more accepting candidates than the register holds: exact re-scan of every active member
            atomicAdd(&counters[gridDim.y * 4 + b * 4 + 0], 1);              // statistics: groups re-scanned
#pragma unroll 1
            for (int k = 0; k < K; ++k)
                if ((actM >> k) & 1u) exact_rescan_tet(tet, t0 + k, b, T, gparam, G, Gx, cb, sq, res, hits, counters);
        } else {
            const int ent[kGrpDepth] = {e0, e1, e2, e3, e4, e5, e6, e7};
            // one atomicMin per entry: only the lowest accepting tet of the group can be the query's answer
#pragma unroll
            for (int i = 0; i < kGrpDepth; ++i) {
                if (i < cnt) {
                    const unsigned bits = (unsigned)ent[i] >> kShift;
                    const int first = K - 1 - (31 - __clz((int)bits));          // most significant set bit = first tet
                    atomicMin(&res[(unsigned)ent[i] & kIdMask], t0 + first);
                }
            }
            if (hits) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (!((actM >> k) & 1u)) continue;
                    int r0 = -1, r1 = -1, r2 = -1, r3 = -1, c = 0;
#pragma unroll
                    for (int i = 0; i < kGrpDepth; ++i) {
                        const bool has = i < cnt && (((unsigned)ent[i] >> (kShift + K - 1 - k)) & 1u);
                        const int qi = (int)((unsigned)ent[i] & kIdMask);
                        r3 = has ? r2 : r3; r2 = has ? r1 : r2; r1 = has ? r0 : r1; r0 = has ? qi : r0;
                        c += has ? 1 : 0;
                    }
                    if (c > 4) {
                        r3 = kHitOverflow;
                        note_overflow(counters, gridDim.y, b, t0 + k);
                    }
                    hits[(size_t)b * T + t0 + k] = make_int4(r0, r1, r2, r3);
                }
            }
        }
    }
    if (counters[b * 4 + 1] > 0) {                                     // irregular queries (normally none)
#pragma unroll 1
        for (int k = 0; k < K; ++k)
            if ((regM >> k) & 1u) fma_irregular_tail_slow(tet, t0 + k, b, T, Q, pts, counters, irregQ, result);
    }
}

// ------------------------------------------------------------------------------------
// k_tet_scan_lds<STAGE_Q> (DEFTET_PIT_LDSB / DEFTET_PIT_LDS): the fused-filter traversal with the block's slice of the
// grid staged in LDS.  What the one-tet-per-lane kernels pay for is the divergent-gather path: ~8 row-bound loads and ~12
// candidate loads per tet, each lane with its own address (TA_TA_BUSY ~85 % of the kernel time, profiles/r02_pmc_*).  The
// 256 consecutive tets of a workgroup are neighbours in any sensibly ordered mesh, so the workgroup
//   1. reduces its lanes' cell ranges to one union box (six LDS atomics per lane),
//   2. copies the box's cell starts into LDS with coalesced loads                       (<= kCapB ints), and, with STAGE_Q,
//   3. the queries of the box's row runs as well: per-row LDS offsets by a block scan   (<= kCapQ queries),
// after which every lane walks ITS OWN cell range exactly as k_tet_scan_fma does, but out of LDS.  Same candidates, same
// certified filter, same exact fallback, same records: results are bit-identical.  A workgroup whose box does not fit
// (incoherent tet order, list mode) keeps the per-lane global loads for whatever did not fit.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

constexpr int kCapB = 2048;        // staged cell starts per workgroup (8 KB)
constexpr int kCapQ = 1024;        // staged queries per workgroup (16 KB)
constexpr int kCapRows = 512;      // rows of the union box (2 per thread in the offset scan)

template <bool STAGE_Q>
__global__ __launch_bounds__(256, 5) void k_tet_scan_lds(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int4 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount)
{
    __shared__ int s_box4[24];
    __shared__ int s_cb[kCapB];
    __shared__ int s_delta[STAGE_Q ? kCapRows : 1];        // LDS position of a row's first staged query minus its global position
    __shared__ float4 s_q[STAGE_Q ? kCapQ : 1];
    __shared__ int s_wsum[5];
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int nblk = gridDim.x;
    const int per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware mapping, see k_tet_scan
    if (vb >= nblk || vb * 256 >= T) return;                           // whole workgroup out of range (uniform)
    const int t = vb * 256 + tid;
    const bool live = t < T;
    PHASE_DECL;
    const Grid g = load_grid(gparam + b * 12);
    Filter F;
    bool regular = false, active = false;
    int cx0 = 0, cx1 = -1, cy0 = 0, cy1 = -1, cz0 = 0, cz1 = -1;
    float m = 0.f;
    {
        float v[12];
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + (live ? t : vb * 256)) * 12);
        const float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
            hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
        }
        const float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        Planes P;
        make_planes(v, P);
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k]) <= kBig);
        const float mn = fminf(fminf(fabsf(P.dv[0]), fabsf(P.dv[1])), fminf(fabsf(P.dv[2]), fabsf(P.dv[3])));
        regular = finite && (P.sv == 0u || P.sv == 15u) && (w >= kWMin) && (mn >= kTau * ((w * w) * w));
        const float sigma = P.sv == 15u ? 1.0f : -1.0f;
        float S[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            S[k] = kErrScale * (fmaxf(fabsf(g.lo[k]), fabsf(g.hi[k])) + 2.0f * fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        float emax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n0 = P.n[i][0], n1 = P.n[i][1], n2 = P.n[i][2];
            const float c2 = fmaf(n0, P.a[i][0], fmaf(n1, P.a[i][1], n2 * P.a[i][2]));
            const float E = fmaf(fabsf(n0), S[0], fmaf(fabsf(n1), S[1], fabsf(n2) * S[2])) + kErrAbs;
            F.N[i][0] = sigma * n0; F.N[i][1] = sigma * n1; F.N[i][2] = sigma * n2;
            F.C[i] = -sigma * c2 - E;
            emax = fmaxf(emax, E);
        }
        F.twoEmax = 2.0f * emax;
        m = w * kMargin;
        const float e0 = lo[0] - m, e1 = hi[0] + m, e2 = lo[1] - m, e3 = hi[1] + m, e4 = lo[2] - m, e5 = hi[2] + m;
        const bool ingrid = !(e1 < g.lo[0] || e0 > g.hi[0] || e3 < g.lo[1] || e2 > g.hi[1] || e5 < g.lo[2] || e4 > g.hi[2]);
        active = live && regular && ingrid;
        if (active) {
            cx0 = cell_of(e0, g.o[0], g.inv[0], Gx); cx1 = cell_of(e1, g.o[0], g.inv[0], Gx);
            cy0 = cell_of(e2, g.o[1], g.inv[1], G);  cy1 = cell_of(e3, g.o[1], g.inv[1], G);
            cz0 = cell_of(e4, g.o[2], g.inv[2], G);  cz1 = cell_of(e5, g.o[2], g.inv[2], G);
        }
        if (live && !regular) {                                        // irregular tet (normally none): k_finalize tests it against every query
            const int k = atomicAdd(&counters[b * 4 + 0], 1);
            irregT[(size_t)b * T + k] = t;
            if (hits) hits[(size_t)b * T + t] = make_int4(-1, -1, -1, kHitOverflow);
        }
        if (live && regular && !ingrid && hits) hits[(size_t)b * T + t] = make_int4(-1, -1, -1, -1);
    }
    PHASE_MARK(4);                                                       // [4] load + setup
    // ---- 1. union box of the workgroup's active lanes: wave butterflies, then four partials per bound through LDS
    //         (64 lanes hitting one LDS word with an atomic serialise: measured 2x the whole kernel)
    {
        constexpr int kBigI = 0x7FFFFFFF;
        const int m0 = wave_min_i(active ? cx0 : kBigI), m1 = wave_min_i(active ? cy0 : kBigI), m2 = wave_min_i(active ? cz0 : kBigI);
        const int m3 = wave_max_i(active ? cx1 : -1), m4 = wave_max_i(active ? cy1 : -1), m5 = wave_max_i(active ? cz1 : -1);
        if ((tid & 63) == 0) {
            int *dst = s_box4 + (tid >> 6) * 6;
            dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3; dst[4] = m4; dst[5] = m5;
        }
    }
    __syncthreads();
    const int ux0 = min(min(s_box4[0], s_box4[6]), min(s_box4[12], s_box4[18])), uy0 = min(min(s_box4[1], s_box4[7]), min(s_box4[13], s_box4[19]));
    const int uz0 = min(min(s_box4[2], s_box4[8]), min(s_box4[14], s_box4[20])), ux1 = max(max(s_box4[3], s_box4[9]), max(s_box4[15], s_box4[21]));
    const int uy1 = max(max(s_box4[4], s_box4[10]), max(s_box4[16], s_box4[22])), uz1 = max(max(s_box4[5], s_box4[11]), max(s_box4[17], s_box4[23]));
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    PHASE_MARK(5);                                                       // [5] union box (butterflies + barrier)
    bool stB = false, stQ = false;
    int nx1 = 1, ny = 1;
    if (ux1 >= ux0) {                                                  // some lane is active (uniform)
        nx1 = ux1 - ux0 + 2;
        ny = uy1 - uy0 + 1;
        const int nz = uz1 - uz0 + 1;
        const long long rowsl = (long long)ny * nz;
        stB = rowsl <= kCapRows && rowsl * nx1 <= kCapB;
        if (stB) {
            // ---- 2. cell starts of the box -> LDS (row r = (cz - uz0) * ny + (cy - uy0), nx1 starts per row)
            const int rows = (int)rowsl, n = rows * nx1;
            const float inv_nx1 = 1.0f / (float)nx1, inv_ny = 1.0f / (float)ny;
            for (int i = tid; i < n; i += 256) {
                const int r = (int)(((float)i + 0.5f) * inv_nx1), x = i - r * nx1;          // exact for these small integers
                const int rz = (int)(((float)r + 0.5f) * inv_ny), ry = r - rz * ny;
                s_cb[i] = cb[((uz0 + rz) * G + (uy0 + ry)) * Gx + ux0 + x];
            }
            __syncthreads();
            PHASE_MARK(6);                                               // [6] cell starts -> LDS (+ barrier)
            if (STAGE_Q) {
                // ---- 3. per-row LDS offsets (block exclusive scan of the run lengths, two rows per thread)
                const int r0 = tid * 2, r1 = r0 + 1;
                const int st0 = r0 < rows ? s_cb[r0 * nx1] : 0, st1 = r1 < rows ? s_cb[r1 * nx1] : 0;
                const int len0 = r0 < rows ? s_cb[r0 * nx1 + nx1 - 1] - st0 : 0;
                const int len1 = r1 < rows ? s_cb[r1 * nx1 + nx1 - 1] - st1 : 0;
                const int sum = len0 + len1;
                int incl = sum;
                const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int u = __shfl_up(incl, off);
                    if (lane >= off) incl += u;
                }
                if (lane == 63) s_wsum[wv] = incl;
                __syncthreads();
                int base = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < wv) base += s_wsum[k];
                const int total = (s_wsum[0] + s_wsum[1]) + (s_wsum[2] + s_wsum[3]);
                const int excl = base + incl - sum;
                stQ = total <= kCapQ;                                   // uniform
                if (stQ) {
                    if (r0 < rows) s_delta[r0] = excl - st0;
                    if (r1 < rows) s_delta[r1] = excl + len0 - st1;
                }
                __syncthreads();
                if (stQ) {
                    // the queries of the box's row runs -> LDS, eight lanes per row (128 contiguous bytes per step)
                    for (int r = tid >> 3; r < rows; r += 32) {
                        const int st = s_cb[r * nx1], nq = s_cb[r * nx1 + nx1 - 1] - st, d = s_delta[r];
                        for (int k = tid & 7; k < nq; k += 8) s_q[st + k + d] = sq[st + k];
                    }
                    __syncthreads();
                }
            }
        }
    }
    PHASE_MARK(7);                                                       // [7] offsets scan + queries -> LDS (+ barriers)
    // ---- traversal (per-lane cursor, see k_tet_scan_grp), operands from LDS where staged.  Three specialisations of
    //      one loop (MODE 0: bounds and queries from global memory; 1: bounds from LDS; 2: both from LDS), picked by a
    //      workgroup-uniform branch OUTSIDE the loop.  Row positions advance by additions; selects use SGPR-pair masks.
    int hcnt = 0;
    int h0 = -1, h1 = -1, h2 = -1, h3 = -1;
    float amin = INFINITY;
    if (active) {
        auto decide = [&](float a, int qi, bool lv) {
            const lanemask_t acc = mask_of(lv && a > 0.f);
            amin = fminf(amin, fabsf(a));                              // a dead second slot repeats the first candidate
            h3 = sel(acc, h2, h3);
            h2 = sel(acc, h1, h2);
            h1 = sel(acc, h0, h1);
            h0 = sel(acc, qi, h0);
            hcnt += sel(acc, 1, 0);
        };
        auto filter = [&](const float4 &q) {
            float A[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) A[i] = fmaf(F.N[i][0], q.x, fmaf(F.N[i][1], q.y, fmaf(F.N[i][2], q.z, F.C[i])));
            return fminf(fminf(A[0], A[1]), fminf(A[2], A[3]));
        };
        auto traverse = [&](auto modeTag) {
            constexpr int MODE = decltype(modeTag)::value;
            // global addressing (MODE 0): byte offsets from the scalar base, as in k_tet_scan_fma
            const unsigned rowStepB = (unsigned)Gx * 4u, rowWrapB = (unsigned)((G - (cy1 - cy0)) * Gx) * 4u;
            const unsigned x0B = (unsigned)cx0 * 4u, x1B = (unsigned)(cx1 + 1) * 4u;
            unsigned rowB = (unsigned)((cz0 * G + cy0) * Gx) * 4u;
            // LDS addressing (MODE 1, 2): staged row r = (cz - uz0) * ny + (cy - uy0); its starts sit at s_cb[r * nx1 ...]
            const int rStep = 1, rWrap = ny - (cy1 - cy0);
            int r = (cz0 - uz0) * ny + (cy0 - uy0);
            int rb0 = r * nx1 + (cx0 - ux0), rb1 = r * nx1 + (cx1 + 1 - ux0);          // positions of this lane's two bounds
            const int rbStep = nx1, rbWrap = rWrap * nx1;
            int cy = cy0, cz = cz0;                                     // the row whose bounds sit in (s2, e2)
            int j = 0, e = 0, d = 0, d2 = 0;
            int s2, e2;
            if (MODE == 0) {
                s2 = ld_off<int>(cb, rowB + x0B);
                e2 = ld_off<int>(cb, rowB + x1B);
            } else {
                s2 = s_cb[rb0];
                e2 = s_cb[rb1];
                if (MODE == 2) d2 = s_delta[r];
            }
            bool haveNext = true;
            while (j < e || haveNext) {
                if (j >= e) {                                           // enter the prefetched row, prefetch the one after it
                    j = s2;
                    e = e2;
                    d = d2;
                    const lanemask_t wrap = mask_of(cy == cy1);
                    cy = sel(wrap, cy0, cy + 1);
                    cz += sel(wrap, 1, 0);
                    haveNext = cz <= cz1;
                    if (MODE == 0) {
                        rowB += sel(wrap, rowWrapB, rowStepB);
                        if (haveNext) {
                            s2 = ld_off<int>(cb, rowB + x0B);
                            e2 = ld_off<int>(cb, rowB + x1B);
                        }
                    } else {
                        const int inc = sel(wrap, rbWrap, rbStep);
                        rb0 += inc;
                        rb1 += inc;
                        if (MODE == 2) r += sel(wrap, rWrap, rStep);
                        if (haveNext) {
                            s2 = s_cb[rb0];
                            e2 = s_cb[rb1];
                            if (MODE == 2) d2 = s_delta[r];
                        }
                    }
                }
                if (j < e) {
                    const bool two = j + 1 < e;
                    const int j1 = sel(mask_of(two), j + 1, j);         // dead slot: the same candidate again (never recorded)
                    float4 q0, q1;
                    if (MODE == 2) {
                        q0 = s_q[j + d];
                        q1 = s_q[j1 + d];
                    } else {
                        q0 = ld_off<float4>(sq, (unsigned)j * 16u);
                        q1 = ld_off<float4>(sq, (unsigned)j1 * 16u);
                    }
                    const float a0 = filter(q0), a1 = filter(q1);
                    decide(a0, __float_as_int(q0.w), true);
                    decide(a1, __float_as_int(q1.w), two);
                    j += 2;
                }
            }
        };
        if (stQ) traverse(std::integral_constant<int, 2>{});
        else if (stB) traverse(std::integral_constant<int, 1>{});
        else traverse(std::integral_constant<int, 0>{});
        PHASE_MARK(8);                                                   // [8] traversal loop
        if (amin <= F.twoEmax || hcnt > 4) {
            atomicAdd(&counters[gridDim.y * 4 + b * 4 + 1], 1);               // statistics: tets re-scanned
            const int4 r4 = exact_rescan(tet + ((size_t)b * T + t) * 12, t, cb, sq, res, G, Gx, cx0, cx1, cy0, cy1, cz0, cz1, m,
                                         counters, gridDim.y, b);
            if (hits) hits[(size_t)b * T + t] = r4;
        } else {
            if (hcnt > 0) atomicMin(&res[h0], t);
            if (hcnt > 1) atomicMin(&res[h1], t);
            if (hcnt > 2) atomicMin(&res[h2], t);
            if (hcnt > 3) atomicMin(&res[h3], t);
            if (hits) hits[(size_t)b * T + t] = make_int4(h0, h1, h2, h3);
        }
    }
    if (live) fma_irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
    PHASE_MARK(9);                                                       // [9] publish / re-scan
}

// ------------------------------------------------------------------------------------
// Wave-cooperative variant of k_tet_scan for spatially coherent tet orders (DEFTET_PIT_STAGED).
// MEASURED NO FASTER than k_tet_scan on the BASELINE workload (0.322 vs 0.324 ms per step: 37 %
// fewer vector-memory instructions, 16 % fewer L1 accesses, but 18 % more VALU work for the
// staging bookkeeping; profiles/r01_pmc_k_tet_scan_variants.json) — kept selectable, not default.
// The 64 tets of a wave mostly visit the same few cell rows, yet every lane fetches its row
// bounds and candidate queries with its own gather instructions (~55 per wave, the kernel's
// bottleneck).  Here a wave first reduces its lanes' cell ranges to one union box; if that box
// is small (<= 64 rows, <= kSubMax cell bounds, <= kStageQ queries) the wave copies the box's
// cell bounds and queries into LDS with a handful of full-width loads, and each lane then walks
// ITS OWN rows out of LDS.  Same candidates, same exact test, same atomicMin — only the source of
// the operands changes.  Waves whose box is too large (incoherent tet order, or a wave that
// straddles two grid columns) take the per-lane gather path of k_tet_scan.
// ------------------------------------------------------------------------------------
#ifndef PIT_STAGE_BATCH
#define PIT_STAGE_BATCH 4
#endif
constexpr int kSubMax = 448;           // staged cell bounds per wave
constexpr int kStageQ = 224;           // staged queries per wave

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef PIT_STAGE_WAVES
#define PIT_STAGE_WAVES 5
#endif
__global__ __launch_bounds__(256, PIT_STAGE_WAVES) void k_tet_scan_staged(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int4 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount)
{
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;   // uncovered-hit counter of the hit buffer (k_finalize appends)
    __shared__ int s_cs[4][kSubMax];
    __shared__ float4 s_q[4][kStageQ];
    __shared__ int s_off[4][65];
    __shared__ int s_rs[4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = gridDim.x, per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware mapping, see k_tet_scan
    const int t = vb * blockDim.x + threadIdx.x;
    if (t - lane >= T) return;                                       // whole wave out of range
    const bool intet = t < T;
    float v[12];
    {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + (intet ? t : t - lane)) * 12);
        float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
    Planes P;
    make_planes(v, P);
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
        hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
    }
    const float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k]) <= kBig);
    const float mn = fminf(fminf(fabsf(P.dv[0]), fabsf(P.dv[1])), fminf(fabsf(P.dv[2]), fabsf(P.dv[3])));
    const bool regular = finite && (P.sv == 0u || P.sv == 15u) && (w >= kWMin) && (mn >= kTau * ((w * w) * w));
    int4 hrec = make_int4(-1, -1, -1, -1);
    int hcnt = 0;
    if (intet && !regular) {
        const int k = atomicAdd(&counters[b * 4 + 0], 1);
        irregT[(size_t)b * T + k] = t;
        hrec.w = kHitOverflow;                                       // accepted by k_finalize, not recorded
    }
    const Grid g = load_grid(gparam + b * 12);
    const float m = w * kMargin;
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { elo[k] = lo[k] - m; ehi[k] = hi[k] + m; }
    const bool active = intet && regular &&
                        !(ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]);
    const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
    const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    auto test = [&](const float4 &q) {
        if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2]) {
            if (accept(P, q.x, q.y, q.z)) {
                const int qi = __float_as_int(q.w);
                atomicMin(&res[qi], t);
                if (hcnt == 0) hrec.x = qi;
                else if (hcnt == 1) hrec.y = qi;
                else if (hcnt == 2) hrec.z = qi;
                else if (hcnt == 3) hrec.w = qi;
                ++hcnt;
            }
        }
    };
    if (__any(active)) {
        constexpr int kBigI = 1 << 30;
        const int ux0 = wave_min_i(active ? cx0 : kBigI), ux1 = wave_max_i(active ? cx1 : -1);
        const int uy0 = wave_min_i(active ? cy0 : kBigI), uy1 = wave_max_i(active ? cy1 : -1);
        const int uz0 = wave_min_i(active ? cz0 : kBigI), uz1 = wave_max_i(active ? cz1 : -1);
        const int nx1 = ux1 - ux0 + 2, ny = uy1 - uy0 + 1, nz = uz1 - uz0 + 1, rows = ny * nz;   // nx1: bounds per row
        bool staged = rows <= 64 && rows * nx1 <= kSubMax;           // wave-uniform
        int total = 0;
        if (staged) {
            const float inv_nx1 = 1.0f / (float)nx1, inv_ny = 1.0f / (float)ny;
            for (int i = lane; i < rows * nx1; i += 64) {
                const int r = (int)(((float)i + 0.5f) * inv_nx1), x = i - r * nx1;      // exact for these small integers
                const int rz = (int)(((float)r + 0.5f) * inv_ny), ry = r - rz * ny;
                s_cs[wv][i] = cb[((uz0 + rz) * G + (uy0 + ry)) * Gx + ux0 + x];
            }
            wave_fence();
            const int len = lane < rows ? s_cs[wv][lane * nx1 + nx1 - 1] - s_cs[wv][lane * nx1] : 0;
            int incl = len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(incl, off);
                if (lane >= off) incl += u;
            }
            total = __shfl(incl, 63);
            s_off[wv][lane] = incl - len;
            if (lane == 63) s_off[wv][64] = total;
            if (lane < rows) s_rs[wv][lane] = s_cs[wv][lane * nx1];
            staged = total <= kStageQ;
        }
        if (staged) {
            wave_fence();
            for (int i = lane; i < total; i += 64) {
                int lo_r = 0, hi_r = rows;                              // largest r with s_off[r] <= i
                while (hi_r - lo_r > 1) {
                    const int mid = (lo_r + hi_r) >> 1;
                    if (s_off[wv][mid] <= i) lo_r = mid; else hi_r = mid;
                }
                s_q[wv][i] = sq[s_rs[wv][lo_r] + (i - s_off[wv][lo_r])];
            }
            wave_fence();
            if (active) {
                for (int cz = cz0; cz <= cz1; ++cz)
                    for (int cy = cy0; cy <= cy1; ++cy) {
                        const int r = (cz - uz0) * ny + (cy - uy0), base = r * nx1;
                        const int s = s_cs[wv][base + cx0 - ux0], e = s_cs[wv][base + cx1 + 1 - ux0];
                        const int l0 = s_off[wv][r] + (s - s_rs[wv][r]), n = e - s;
                        for (int j = 0; j < n; j += PIT_STAGE_BATCH) {
                            float4 qq[PIT_STAGE_BATCH];
#pragma unroll
                            for (int k = 0; k < PIT_STAGE_BATCH; ++k) qq[k] = s_q[wv][l0 + min(j + k, n - 1)];
#pragma unroll
                            for (int k = 0; k < PIT_STAGE_BATCH; ++k)
                                if (k == 0 || j + k < n) test(qq[k]);
                        }
                    }
            }
        } else if (active) {
            // per-lane gather path (identical to k_tet_scan)
            int cy = cy0, cz = cz0;
            int s = cb[(cz * G + cy) * Gx + cx0];
            int e = cb[(cz * G + cy) * Gx + cx1 + 1];
            for (;;) {
                int ny2 = cy + 1, nz2 = cz;
                if (ny2 > cy1) { ny2 = cy0; nz2 = cz + 1; }
                const bool more = nz2 <= cz1;
                int s2 = 0, e2 = 0;
                if (more) {
                    const int row2 = (nz2 * G + ny2) * Gx;
                    s2 = cb[row2 + cx0];
                    e2 = cb[row2 + cx1 + 1];
                }
                for (int j = s; j < e; j += PIT_STAGE_BATCH) {
                    const int last = e - 1;
                    float4 qq[PIT_STAGE_BATCH];
#pragma unroll
                    for (int k = 0; k < PIT_STAGE_BATCH; ++k) qq[k] = sq[min(j + k, last)];
#pragma unroll
                    for (int k = 0; k < PIT_STAGE_BATCH; ++k)
                        if (k == 0 || j + k < e) test(qq[k]);
                }
                if (!more) break;
                s = s2; e = e2; cy = ny2; cz = nz2;
            }
        }
    }
    if (hits && intet) {
        if (hcnt > 4) {
            hrec.w = kHitOverflow;
            note_overflow(counters, gridDim.y, b, t);
        }
        hits[(size_t)b * T + t] = hrec;
    }
    if (intet) irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
}

// ------------------------------------------------------------------------------------
// Row-balanced variant of k_tet_scan (DEFTET_PIT_ROWS).  In k_tet_scan a lane walks ALL cell rows of
// its tet, so a wave issues max-over-lanes(rows) x max-over-lanes(batches) gather rounds while the
// average lane needs 2.25 rows (measured lane utilisation 42 %).  Here the unit of work is one
// (tet, row) pair: the 64 tets of a wave publish their plane records in LDS, the rows are numbered
// by a wave prefix sum, and the wave processes them 64 at a time, each lane fetching the planes of
// the row's owner from LDS.  Same candidates, same exact test, same atomicMin; hit records are
// collected per tet in LDS.
// ------------------------------------------------------------------------------------
constexpr int kRowWords = 33;          // n[12] | a[12] | sv | elo[3] | ehi[3] | cx0,cx1 | cy0,ny,cz0

__global__ __launch_bounds__(256, 4) void k_tet_scan_rows(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ cells,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int4 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount)
{
    if (ucount && blockIdx.x == 0 && threadIdx.x == 0) ucount[blockIdx.y] = 0;   // uncovered-hit counter of the hit buffer (k_finalize appends)
    __shared__ float s_rec[4][kRowWords][64];
    __shared__ int s_off[4][65];
    __shared__ int s_hcnt[4][64];
    __shared__ int s_hrec[4][4][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nblk = gridDim.x, per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);       // XCD-aware mapping, see k_tet_scan
    const int t = vb * blockDim.x + threadIdx.x;
    const int tw = t - lane;                                         // first tet of this wave
    if (vb >= nblk || tw >= T) return;                               // whole wave out of range
    const bool intet = t < T;
    const Grid g = load_grid(gparam + b * 12);
    int nrows = 0;
    bool irregularTet = false;
    {
        float v[12];
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + (intet ? t : tw)) * 12);
        float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        Planes P;
        make_planes(v, P);
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
            hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
        }
        const float w = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
        bool finite = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) finite = finite && (fabsf(v[k]) <= kBig);
        const float mn = fminf(fminf(fabsf(P.dv[0]), fabsf(P.dv[1])), fminf(fabsf(P.dv[2]), fabsf(P.dv[3])));
        const bool regular = finite && (P.sv == 0u || P.sv == 15u) && (w >= kWMin) && (mn >= kTau * ((w * w) * w));
        irregularTet = intet && !regular;
        const float m = w * kMargin;
        float elo[3], ehi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { elo[k] = lo[k] - m; ehi[k] = hi[k] + m; }
        const bool active = intet && regular &&
                            !(ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]);
        const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
        const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
        const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
        const int ny = cy1 - cy0 + 1;
        nrows = active ? ny * (cz1 - cz0 + 1) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_rec[wv][i * 3 + k][lane] = P.n[i][k];
                s_rec[wv][12 + i * 3 + k][lane] = P.a[i][k];
            }
        s_rec[wv][24][lane] = __int_as_float((int)P.sv);
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_rec[wv][25 + k][lane] = elo[k]; s_rec[wv][28 + k][lane] = ehi[k]; }
        s_rec[wv][31][lane] = __int_as_float(cx0 | (cx1 << 16));
        s_rec[wv][32][lane] = __int_as_float(cy0 | (ny << 8) | (cz0 << 16));
    }
    if (irregularTet) irregT[(size_t)b * T + atomicAdd(&counters[b * 4 + 0], 1)] = t;
    s_hcnt[wv][lane] = 0;
    // exclusive prefix of the row counts
    int incl = nrows;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    const int M = __shfl(incl, 63);
    s_off[wv][lane] = incl - nrows;
    if (lane == 63) s_off[wv][64] = M;
    wave_fence();
    const int *cb = cells + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    for (int base = 0; base < M; base += 64) {
        const int item = base + lane;
        if (item < M) {
            int lo_j = 0, hi_j = 64;                                  // largest j with s_off[j] <= item
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo_j + hi_j) >> 1;
                if (s_off[wv][mid] <= item) lo_j = mid; else hi_j = mid;
            }
            const int j = lo_j;
            int r = item - s_off[wv][j];
            Planes P;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    P.n[i][k] = s_rec[wv][i * 3 + k][j];
                    P.a[i][k] = s_rec[wv][12 + i * 3 + k][j];
                }
            P.sv = (unsigned)__float_as_int(s_rec[wv][24][j]);
            float elo[3], ehi[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { elo[k] = s_rec[wv][25 + k][j]; ehi[k] = s_rec[wv][28 + k][j]; }
            const int pk1 = __float_as_int(s_rec[wv][31][j]), pk2 = __float_as_int(s_rec[wv][32][j]);
            const int cx0 = pk1 & 0xFFFF, cx1 = pk1 >> 16, cy0 = pk2 & 0xFF, ny = (pk2 >> 8) & 0xFF, cz0 = pk2 >> 16;
            const int rz = (int)(((float)r + 0.5f) * (1.0f / (float)ny));     // exact for these small integers
            const int cy = cy0 + (r - rz * ny), cz = cz0 + rz;
            const int row = (cz * G + cy) * Gx;
            const int s = cb[row + cx0], e = cb[row + cx1 + 1];
            const int tg = tw + j;
            for (int jq = s; jq < e; jq += 2) {
                float4 qq[2];
                qq[0] = sq[jq];
                if (jq + 1 < e) qq[1] = sq[jq + 1];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k == 1 && jq + 1 >= e) break;
                    const float4 q = qq[k];
                    if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2]) {
                        if (accept(P, q.x, q.y, q.z)) {
                            const int qi = __float_as_int(q.w);
                            atomicMin(&res[qi], tg);
                            const int slot = atomicAdd(&s_hcnt[wv][j], 1);
                            if (slot < 4) s_hrec[wv][slot][j] = qi;
                        }
                    }
                }
            }
        }
    }
    wave_fence();
    if (hits && intet) {
        int4 hrec = make_int4(-1, -1, -1, -1);
        const int hcnt = s_hcnt[wv][lane];
        if (hcnt > 0) hrec.x = s_hrec[wv][0][lane];
        if (hcnt > 1) hrec.y = s_hrec[wv][1][lane];
        if (hcnt > 2) hrec.z = s_hrec[wv][2][lane];
        if (hcnt > 3) hrec.w = s_hrec[wv][3][lane];
        if (irregularTet) hrec = make_int4(-1, -1, -1, kHitOverflow);          // accepted by k_finalize, not recorded
        if (hcnt > 4) {
            hrec.w = kHitOverflow;
            note_overflow(counters, gridDim.y, b, t);
        }
        hits[(size_t)b * T + t] = hrec;
    }
    if (intet && counters[b * 4 + 1] > 0) {                           // irregular queries (normally none)
        Planes P;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                P.n[i][k] = s_rec[wv][i * 3 + k][lane];
                P.a[i][k] = s_rec[wv][12 + i * 3 + k][lane];
            }
        P.sv = (unsigned)__float_as_int(s_rec[wv][24][lane]);
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
    }
}

// barycentric weights, utils/tet_utils.py:25-45 (same association as the torch expression)
__device__ __forceinline__ float triple(const float *a, const float *b, const float *c)
{
    float x0 = b[1] * c[2] - b[2] * c[1];
    float x1 = b[2] * c[0] - b[0] * c[2];
    float x2 = b[0] * c[1] - b[1] * c[0];
    return (a[0] * x0 + a[1] * x1) + a[2] * x2;
}

__global__ __launch_bounds__(256) void k_finalize(const float *__restrict__ tet, const float *__restrict__ pts, int T,
                                                  int Q, const int *__restrict__ result, float *cond, float *bary,
                                                  const float *__restrict__ pred, float *occ, const int4 *__restrict__ hits,
                                                  int *ucount, int *ulist, const int *__restrict__ counters,
                                                  const int *__restrict__ irregT)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const size_t i = (size_t)b * Q + q;
    int r = result[i];
    // irregular tets (not certified for the grid filter; normally none) are tested here against
    // every query: query-centric, so no atomics and no extra launch
    const int nIrregT = counters ? counters[b * 4 + 0] : 0;
    if (nIrregT > 0) {
        const float *p = pts + i * 3;
        const float x = p[0], y = p[1], z = p[2];
        for (int k = 0; k < nIrregT; ++k) {
            const int t = irregT[(size_t)b * T + k];
            float v[12];
            const float *src = tet + ((size_t)b * T + t) * 12;
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = src[j];
            Planes P;
            make_planes(v, P);
            if (accept(P, x, y, z)) r = min(r, t);
        }
    }
    const bool hit = r != kMiss;
    cond[i] = hit ? (float)r : -1.0f;                               // :177, :149
    if (occ) occ[i] = pred[(size_t)b * T + (hit ? r : 0)];          // paste_occ: misses alias tet 0 (deftet.py:133-135)
    if (hits && hit) {
        // is this hit in its tet's record?  (not if the tet overflowed / is irregular, or if the
        // query took the irregular-query side path, which records nothing)
        const float *pq = pts + i * 3;
        // records are complete unless some tet is irregular or overflowed (wave-uniform test: the
        // per-hit gather of the record is skipped for ordinary meshes)
        bool covered = query_regular(pq[0], pq[1], pq[2]);
        if (counters) {
            const int nB = gridDim.y, nOvf = counters[nB * 4 + b * 4 + 2];
            if (counters[b * 4 + 0] > 0 || nOvf > kOvfCap) {
                // irregular tets exist, or more overflowed tets than the list holds: read the winning tet's record
                covered = covered && hits[(size_t)b * T + r].w != kHitOverflow;
            } else {
                // the usual case: a handful of overflowed tets per shape, listed; wave-uniform scalar reads, no gather
                const int *ovf = counters + nB * 8 + b * kOvfCap;
                for (int k = 0; k < nOvf; ++k) covered = covered && ovf[k] != r;
            }
        }
        if (!covered) ulist[(size_t)b * Q + atomicAdd(&ucount[b], 1)] = q;
    }
    if (!bary) return;
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hit) {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + r) * 12);
        float4 t0 = src[0], t1 = src[1], t2 = src[2];
        const float A[3] = {t0.x, t0.y, t0.z}, Bv[3] = {t0.w, t1.x, t1.y}, Cv[3] = {t1.z, t1.w, t2.x}, D[3] = {t2.y, t2.z, t2.w};
        const float *pp = pts + i * 3;
        float vap[3], vbp[3], vab[3], vac[3], vad[3], vbc[3], vbd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vap[k] = pp[k] - A[k]; vbp[k] = pp[k] - Bv[k];
            vab[k] = Bv[k] - A[k]; vac[k] = Cv[k] - A[k]; vad[k] = D[k] - A[k];
            vbc[k] = Cv[k] - Bv[k]; vbd[k] = D[k] - Bv[k];
        }
        float v6 = 1.0f / triple(vab, vac, vad);
        wq.x = triple(vbp, vbd, vbc) * v6;
        wq.y = triple(vap, vac, vad) * v6;
        wq.z = triple(vap, vad, vab) * v6;
        wq.w = triple(vap, vab, vac) * v6;
    }
    reinterpret_cast<float4 *>(bary)[i] = wq;
}

// ------------------------------------------------------------------------------------
// brute force (DEFTET_PIT_BRUTE): the algorithmic equivalent of the reference kernel —
// every query meets every tet in index order — restructured for CDNA4: plane records are
// computed once per tet, read through the scalar cache as wave-uniform operands, one
// query per lane, wave-wide early exit once all 64 lanes have their first hit.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_records(const float *__restrict__ tet, long long n, float *rec)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[12];
    const float4 *src = reinterpret_cast<const float4 *>(tet + i * 12);
    float4 a = src[0], bq = src[1], c = src[2];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
    v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    Planes P;
    make_planes(v, P);
    float4 *dst = reinterpret_cast<float4 *>(rec + i * 32);
    dst[0] = make_float4(P.n[0][0], P.n[0][1], P.n[0][2], P.a[0][0]);
    dst[1] = make_float4(P.a[0][1], P.a[0][2], P.n[1][0], P.n[1][1]);
    dst[2] = make_float4(P.n[1][2], P.a[1][0], P.a[1][1], P.a[1][2]);
    dst[3] = make_float4(P.n[2][0], P.n[2][1], P.n[2][2], P.a[2][0]);
    dst[4] = make_float4(P.a[2][1], P.a[2][2], P.n[3][0], P.n[3][1]);
    dst[5] = make_float4(P.n[3][2], P.a[3][0], P.a[3][1], P.a[3][2]);
    dst[6] = make_float4(__int_as_float((int)P.sv), 0.f, 0.f, 0.f);
    dst[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_brute(const float *__restrict__ rec, const float *__restrict__ pts, int T, int Q,
                                               int *result)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < Q;
    const float *p = pts + ((size_t)b * Q + (live ? q : 0)) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float *__restrict__ r = rec + (size_t)b * T * 32;
    int found = live ? kMiss : 0;
    for (int t = 0; t < T; ++t) {
        const float *__restrict__ s = r + (size_t)t * 32;      // wave-uniform address -> scalar loads
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float nx = s[i * 6 + 0], ny = s[i * 6 + 1], nz = s[i * 6 + 2];
            const float ax = s[i * 6 + 3], ay = s[i * 6 + 4], az = s[i * 6 + 5];
            float rx = px - ax, ry = py - ay, rz = pz - az;
            float dotp = nx * rx + ny * ry + nz * rz;
            m |= (dotp > 0 ? 1u : 0u) << i;
        }
        const unsigned x = m ^ (unsigned)__float_as_int(s[24]);
        if ((x == 0u || x == 15u) && found == kMiss) found = t;
        if ((t & 7) == 7 && __ballot(found == kMiss) == 0ull) break;   // whole wave done
    }
    if (live) result[(size_t)b * Q + q] = found;
}

// ------------------------------------------------------------------------------------
// A1b backward: dL/dtet = -w_k * G,  G = sum_i g_i * grad_p(w_i)   
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__global__ __launch_bounds__(256) void k_bary_bwd(const float *__restrict__ tet, const float *__restrict__ pts,
                                                  const float *__restrict__ cond, const float *__restrict__ grad_w, int T,
                                                  int Q, float *grad_tet, float *grad_pts)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const size_t i = (size_t)b * Q + q;
    const float cf = cond[i];
    const bool hit = cf >= 0.f;
    float G3[3] = {0.f, 0.f, 0.f};
    if (hit) {
        const int t = (int)cf;
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + t) * 12);
        float4 t0 = src[0], t1 = src[1], t2 = src[2];
        const float A[3] = {t0.x, t0.y, t0.z}, Bv[3] = {t0.w, t1.x, t1.y}, Cv[3] = {t1.z, t1.w, t2.x}, D[3] = {t2.y, t2.z, t2.w};
        const float *pp = pts + i * 3;
        float vap[3], vbp[3], vab[3], vac[3], vad[3], vbc[3], vbd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vap[k] = pp[k] - A[k]; vbp[k] = pp[k] - Bv[k];
            vab[k] = Bv[k] - A[k]; vac[k] = Cv[k] - A[k]; vad[k] = D[k] - A[k];
            vbc[k] = Cv[k] - Bv[k]; vbd[k] = D[k] - Bv[k];
        }
        float na[3], nb[3], nc[3], nd[3];
        cross3(vbd, vbc, na);    // grad_p va6
        cross3(vac, vad, nb);    // grad_p vb6
        cross3(vad, vab, nc);    // grad_p vc6
        cross3(vab, vac, nd);    // grad_p vd6
        const float v6 = 1.0f / (vab[0] * nb[0] + vab[1] * nb[1] + vab[2] * nb[2]);
        float w[4];
        w[0] = (vbp[0] * na[0] + vbp[1] * na[1] + vbp[2] * na[2]) * v6;
        w[1] = (vap[0] * nb[0] + vap[1] * nb[1] + vap[2] * nb[2]) * v6;
        w[2] = (vap[0] * nc[0] + vap[1] * nc[1] + vap[2] * nc[2]) * v6;
        w[3] = (vap[0] * nd[0] + vap[1] * nd[1] + vap[2] * nd[2]) * v6;
        const float4 g = reinterpret_cast<const float4 *>(grad_w)[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) G3[k] = (g.x * na[k] + g.y * nb[k] + g.z * nc[k] + g.w * nd[k]) * v6;
        float *gt = grad_tet + ((size_t)b * T + t) * 12;
#pragma unroll
        for (int vtx = 0; vtx < 4; ++vtx)
#pragma unroll
            for (int k = 0; k < 3; ++k) unsafeAtomicAdd(gt + vtx * 3 + k, -w[vtx] * G3[k]);
    }
    if (grad_pts) {
        grad_pts[i * 3 + 0] = G3[0];
        grad_pts[i * 3 + 1] = G3[1];
        grad_pts[i * 3 + 2] = G3[2];
    }
}

// --- atomic-free backward -----------------------------------------------------------------
// Device-scope fp32 atomics are fabric transactions on MI355X (~20 G/s measured); 12 per hit
// made the scatter above the slowest kernel of the step.  Instead: thread the hits of every
// tet into a linked list (ONE returning atomicExch per hit), then one lane per tet walks its
// list, accumulates the 12 partials in registers and stores its 48-byte gradient record
// once, coalesced.  No memset of grad_tet is needed: every tet is written.
constexpr int kLinkPer = 4;    // queries per thread (returning atomicExch: keep 4 in flight per lane)

// gocc != NULL: additionally sum the paste_occ gradient of the MISSES (they alias tet 0,
// deftet.py:133) per shape into missSum[b] — one atomic per workgroup.
__global__ __launch_bounds__(256) void k_hit_link(const float *__restrict__ cond, int T, int Q, int *head, int *next,
                                                  const float *__restrict__ gocc, float *missSum)
{
    __shared__ float wsum[4];
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * (256 * kLinkPer) + threadIdx.x;
    int tgt[kLinkPer];
    float gm = 0.f;
#pragma unroll
    for (int k = 0; k < kLinkPer; ++k) {
        const int q = q0 + k * 256;
        tgt[k] = -1;
        if (q < Q) {
            const float c = cond[(size_t)b * Q + q];
            if (c >= 0.f) tgt[k] = (int)c;
            else if (gocc) gm += gocc[(size_t)b * Q + q];
        }
    }
    int prev[kLinkPer];
#pragma unroll
    for (int k = 0; k < kLinkPer; ++k) prev[k] = tgt[k] >= 0 ? atomicExch(&head[(size_t)b * T + tgt[k]], q0 + k * 256) : -1;
#pragma unroll
    for (int k = 0; k < kLinkPer; ++k)
        if (tgt[k] >= 0) next[(size_t)b * Q + q0 + k * 256] = prev[k];
    if (gocc) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm += __shfl_xor(gm, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = gm;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
            if (tot != 0.f) unsafeAtomicAdd(&missSum[b], tot);
        }
    }
}

__global__ __launch_bounds__(256) void k_bary_bwd_gather(const float *__restrict__ tet, const float *__restrict__ pts,
                                                         const float *__restrict__ grad_w, const int *__restrict__ head,
                                                         const int *__restrict__ next, int T, int Q, float *grad_tet,
                                                         float *grad_pts, int accumulate, const float *__restrict__ gocc,
                                                         const float *__restrict__ missSum, float *grad_pred)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    int q = head[(size_t)b * T + t];
    float gp = (grad_pred && t == 0) ? missSum[b] : 0.f;             // clamped misses paste from tet 0
    if (q >= 0) {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + t) * 12);
        float4 t0 = src[0], t1 = src[1], t2 = src[2];
        const float A[3] = {t0.x, t0.y, t0.z}, Bv[3] = {t0.w, t1.x, t1.y}, Cv[3] = {t1.z, t1.w, t2.x}, D[3] = {t2.y, t2.z, t2.w};
        float vab[3], vac[3], vad[3], vbc[3], vbd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            vab[k] = Bv[k] - A[k]; vac[k] = Cv[k] - A[k]; vad[k] = D[k] - A[k];
            vbc[k] = Cv[k] - Bv[k]; vbd[k] = D[k] - Bv[k];
        }
        float na[3], nb[3], nc[3], nd[3];
        cross3(vbd, vbc, na);
        cross3(vac, vad, nb);
        cross3(vad, vab, nc);
        cross3(vab, vac, nd);
        const float v6 = 1.0f / (vab[0] * nb[0] + vab[1] * nb[1] + vab[2] * nb[2]);
        while (q >= 0) {
            const size_t i = (size_t)b * Q + q;
            const float *pp = pts + i * 3;
            const float4 g = reinterpret_cast<const float4 *>(grad_w)[i];
            const int qn = next[i];
            float vap[3], vbp[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { vap[k] = pp[k] - A[k]; vbp[k] = pp[k] - Bv[k]; }
            float w[4];
            w[0] = (vbp[0] * na[0] + vbp[1] * na[1] + vbp[2] * na[2]) * v6;
            w[1] = (vap[0] * nb[0] + vap[1] * nb[1] + vap[2] * nb[2]) * v6;
            w[2] = (vap[0] * nc[0] + vap[1] * nc[1] + vap[2] * nc[2]) * v6;
            w[3] = (vap[0] * nd[0] + vap[1] * nd[1] + vap[2] * nd[2]) * v6;
            float G3[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) G3[k] = (g.x * na[k] + g.y * nb[k] + g.z * nc[k] + g.w * nd[k]) * v6;
#pragma unroll
            for (int vtx = 0; vtx < 4; ++vtx)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[vtx * 3 + k] += -w[vtx] * G3[k];
            if (grad_pts) { grad_pts[i * 3] = G3[0]; grad_pts[i * 3 + 1] = G3[1]; grad_pts[i * 3 + 2] = G3[2]; }
            if (grad_pred) gp += gocc[i];
            q = qn;
        }
    }
    if (grad_pred) grad_pred[(size_t)b * T + t] = accumulate ? grad_pred[(size_t)b * T + t] + gp : gp;
    float4 *dst = reinterpret_cast<float4 *>(grad_tet + ((size_t)b * T + t) * 12);
    float4 o0 = make_float4(acc[0], acc[1], acc[2], acc[3]), o1 = make_float4(acc[4], acc[5], acc[6], acc[7]),
           o2 = make_float4(acc[8], acc[9], acc[10], acc[11]);
    if (accumulate) {
        float4 p0 = dst[0], p1 = dst[1], p2 = dst[2];
        o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
        o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        o2.x += p2.x; o2.y += p2.y; o2.z += p2.z; o2.w += p2.w;
    }
    dst[0] = o0; dst[1] = o1; dst[2] = o2;
}

// --- backward from the forward's hit records: no atomics, no lists, no memsets ---------------------
// per-block partial sums of the paste_occ gradient of the misses (they alias tet 0, deftet.py:133)
constexpr int kMissParts = 64;

struct TetGrad {
    float A[3], Bv[3];
    float na[3], nb[3], nc[3], nd[3];
    float v6;
};
__device__ __forceinline__ void tet_grad_setup(const float *__restrict__ tet, size_t i, TetGrad &g)
{
    const float4 *src = reinterpret_cast<const float4 *>(tet + i * 12);
    const float4 t0 = src[0], t1 = src[1], t2 = src[2];
    const float C[3] = {t1.z, t1.w, t2.x}, D[3] = {t2.y, t2.z, t2.w};
    g.A[0] = t0.x; g.A[1] = t0.y; g.A[2] = t0.z; g.Bv[0] = t0.w; g.Bv[1] = t1.x; g.Bv[2] = t1.y;
    float vab[3], vac[3], vad[3], vbc[3], vbd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vab[k] = g.Bv[k] - g.A[k]; vac[k] = C[k] - g.A[k]; vad[k] = D[k] - g.A[k];
        vbc[k] = C[k] - g.Bv[k]; vbd[k] = D[k] - g.Bv[k];
    }
    cross3(vbd, vbc, g.na);
    cross3(vac, vad, g.nb);
    cross3(vad, vab, g.nc);
    cross3(vab, vac, g.nd);
    g.v6 = 1.0f / (vab[0] * g.nb[0] + vab[1] * g.nb[1] + vab[2] * g.nb[2]);
}
// contribution of one hit query to its tet's 12 gradient components; returns dL/dp in G3
__device__ __forceinline__ void tet_grad_add(const TetGrad &g, const float *pp, const float4 gw, float *acc, float *G3)
{
    float vap[3], vbp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { vap[k] = pp[k] - g.A[k]; vbp[k] = pp[k] - g.Bv[k]; }
    float w[4];
    w[0] = (vbp[0] * g.na[0] + vbp[1] * g.na[1] + vbp[2] * g.na[2]) * g.v6;
    w[1] = (vap[0] * g.nb[0] + vap[1] * g.nb[1] + vap[2] * g.nb[2]) * g.v6;
    w[2] = (vap[0] * g.nc[0] + vap[1] * g.nc[1] + vap[2] * g.nc[2]) * g.v6;
    w[3] = (vap[0] * g.nd[0] + vap[1] * g.nd[1] + vap[2] * g.nd[2]) * g.v6;
#pragma unroll
    for (int k = 0; k < 3; ++k) G3[k] = (gw.x * g.na[k] + gw.y * g.nb[k] + gw.z * g.nc[k] + gw.w * g.nd[k]) * g.v6;
#pragma unroll
    for (int vtx = 0; vtx < 4; ++vtx)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[vtx * 3 + k] += -w[vtx] * G3[k];
}

__global__ __launch_bounds__(256) void k_bary_bwd_hits(const float *__restrict__ tet, const float *__restrict__ pts,
                                                       const float *__restrict__ cond, const float *__restrict__ grad_w,
                                                       const int4 *__restrict__ hits, int T, int Q, float *grad_tet,
                                                       float *grad_pts, int accumulate, const float *__restrict__ gocc,
                                                       float *grad_pred, float *missPart, int nMissParts)
{
    const int b = blockIdx.y;
    // paste_occ sends every miss to tet 0 (deftet.py:133-135), so grad_pred[b,0] also gets the sum
    // of grad_occ over the misses.  The first nMissParts blocks of a shape each sum a slice of the
    // queries on the side (hidden under this kernel's own traffic); k_bary_bwd_tail adds the
    // partials up in a fixed order.
    if (grad_pred && (int)blockIdx.x < nMissParts) {
        __shared__ float wsum[4];
        float gm = 0.f;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += nMissParts * blockDim.x)
            if (cond[(size_t)b * Q + q] < 0.f) gm += gocc[(size_t)b * Q + q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm += __shfl_xor(gm, off);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = gm;
        __syncthreads();
        if (threadIdx.x == 0) missPart[b * kMissParts + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    float gp = 0.f;
    const int4 h = hits[(size_t)b * T + t];
    if (h.x >= 0 && h.w != kHitOverflow) {                         // slots fill in order: x < 0 means no accepted query
        TetGrad g;
        tet_grad_setup(tet, (size_t)b * T + t, g);
        // the record lists the accepted queries in traversal order, which depends on the (arbitrary)
        // order of queries inside a grid cell: sort the four ids so that the fp32 sums below are
        // added in the same order on every run (empty slots, -1, go last)
        unsigned hu[4] = {(unsigned)h.x, (unsigned)h.y, (unsigned)h.z, (unsigned)h.w};
#define DEFTET_CSWAP(a, b) { const unsigned lo_ = min(hu[a], hu[b]), hi_ = max(hu[a], hu[b]); hu[a] = lo_; hu[b] = hi_; }
        DEFTET_CSWAP(0, 1) DEFTET_CSWAP(2, 3) DEFTET_CSWAP(0, 2) DEFTET_CSWAP(1, 3) DEFTET_CSWAP(1, 2)
#undef DEFTET_CSWAP
        const int hq[4] = {(int)hu[0], (int)hu[1], (int)hu[2], (int)hu[3]};
        const float tf = (float)t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = hq[k];
            if (q < 0) continue;
            const size_t i = (size_t)b * Q + q;
            if (cond[i] != tf) continue;                           // accepted here, but a lower-index tet won the query
            float G3[3];
            tet_grad_add(g, pts + i * 3, reinterpret_cast<const float4 *>(grad_w)[i], acc, G3);
            if (grad_pts) { grad_pts[i * 3] = G3[0]; grad_pts[i * 3 + 1] = G3[1]; grad_pts[i * 3 + 2] = G3[2]; }
            if (grad_pred) gp += gocc[i];
        }
    }
    if (grad_pred) grad_pred[(size_t)b * T + t] = accumulate ? grad_pred[(size_t)b * T + t] + gp : gp;
    float4 *dst = reinterpret_cast<float4 *>(grad_tet + ((size_t)b * T + t) * 12);
    float4 o0 = make_float4(acc[0], acc[1], acc[2], acc[3]), o1 = make_float4(acc[4], acc[5], acc[6], acc[7]),
           o2 = make_float4(acc[8], acc[9], acc[10], acc[11]);
    if (accumulate) {
        float4 p0 = dst[0], p1 = dst[1], p2 = dst[2];
        o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
        o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        o2.x += p2.x; o2.y += p2.y; o2.z += p2.z; o2.w += p2.w;
    }
    dst[0] = o0; dst[1] = o1; dst[2] = o2;
}

// After k_bary_bwd_hits, one launch for the two leftovers:
//  * grad_pred[b,0] += sum of the miss partials (one wave, fixed order -> deterministic);
//  * the (normally empty) list of hits that are in no tet record: float atomics.
__global__ __launch_bounds__(256) void k_bary_bwd_tail(const float *__restrict__ tet, const float *__restrict__ pts,
                                                       const float *__restrict__ cond, const float *__restrict__ grad_w,
                                                       const int *__restrict__ ucount, const int *__restrict__ ulist, int T,
                                                       int Q, float *grad_tet, float *grad_pts,
                                                       const float *__restrict__ gocc, float *grad_pred,
                                                       const float *__restrict__ missPart, int nMissParts)
{
    const int b = blockIdx.y;
    if (grad_pred && blockIdx.x == 0 && threadIdx.x < 64) {
        float v = (int)threadIdx.x < nMissParts ? missPart[b * kMissParts + threadIdx.x] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (threadIdx.x == 0) unsafeAtomicAdd(&grad_pred[(size_t)b * T], v);   // atomic only because of the list below
    }
    const int n = ucount[b];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const int q = ulist[(size_t)b * Q + k];
        const size_t i = (size_t)b * Q + q;
        const int t = (int)cond[i];
        TetGrad g;
        tet_grad_setup(tet, (size_t)b * T + t, g);
        float acc[12], G3[3];
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] = 0.f;
        tet_grad_add(g, pts + i * 3, reinterpret_cast<const float4 *>(grad_w)[i], acc, G3);
        float *gt = grad_tet + ((size_t)b * T + t) * 12;
#pragma unroll
        for (int j = 0; j < 12; ++j) unsafeAtomicAdd(gt + j, acc[j]);
        if (grad_pts) { grad_pts[i * 3] = G3[0]; grad_pts[i * 3 + 1] = G3[1]; grad_pts[i * 3 + 2] = G3[2]; }
        if (grad_pred) unsafeAtomicAdd(&grad_pred[(size_t)b * T + t], gocc[i]);
    }
}

// paste_occ, layers/DefTet/deftet.py:132-136
__global__ __launch_bounds__(256) void k_paste_fwd(const float *__restrict__ pred, float *cond, float *out, int T, int Q,
                                                   int clamp_inplace)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    const size_t i = (size_t)b * Q + q;
    float c = cond[i];
    if (c < 0) {                                   // condition[condition < 0] = 0
        c = 0.f;
        if (clamp_inplace) cond[i] = 0.f;
    }
    out[i] = pred[(size_t)b * T + (long long)c];   // torch.gather(..., index=condition.long())
}

__global__ __launch_bounds__(256) void k_paste_bwd(const float *__restrict__ cond, const float *__restrict__ gout,
                                                   float *gpred, int T, int Q)
{
    const int b = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = q < Q;
    const size_t i = (size_t)b * Q + (live ? q : 0);
    const float c = live ? cond[i] : 0.f;
    const float g = live ? gout[i] : 0.f;
    // misses alias tet 0 (deftet.py:133; the caller may already have clamped them to 0.0):
    // thousands of queries per shape hit one address, so everything destined for tet 0 is
    // summed across the wave first and sent as one atomic
    const bool miss = live && c < 1.0f;
    float gm = miss ? g : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gm += __shfl_xor(gm, off);
    __shared__ float wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (tot != 0.f) unsafeAtomicAdd(&gpred[(size_t)b * T], tot);
    }
    if (live && !miss) unsafeAtomicAdd(&gpred[(size_t)b * T + (long long)c], g);
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static int pick_G(int T, int Q)
{
    double gdiv = PIT_GDIV;
    if (const char *e = getenv("DEFTET_PIT_GDIV")) {            // experiments only: tets per cell
        const double v = atof(e);
        if (v >= 0.25 && v <= 4096.0) gdiv = v;
    }
    double qdiv = 2.0;
    if (const char *e = getenv("DEFTET_PIT_QDIV")) {            // experiments only: queries per cell
        const double v = atof(e);
        if (v >= 0.03 && v <= 4096.0) qdiv = v;
    }
    double a = T / gdiv, bq = (Q > 0 ? Q : 1) / qdiv;
    double m = a < bq ? a : bq;
    int G = (int)llround(cbrt(m < 1 ? 1 : m));
    if (G < 1) G = 1;
    if (G > kMaxG) G = kMaxG;
    return G;
}

struct Layout {
    int G, Gx, nRowBlk, chunkQ;
    long long cellStride;   // padded cells per shape (>= Gx*G*G + 1)
    size_t bytes;
    float *bboxPart;
    int *counters, *cells, *blockHist, *rowTotal, *rowStart, *result, *irregT, *irregQ, *deferT;
    int2 *qkey;
    float4 *rowSorted, *sortedQ;
    float *rec, *gparam;
};

static Layout make_layout(int B, int T, int Q, int algo, void *ws, size_t wsBytes)
{
    Layout L{};
    Arena A(ws, wsBytes);
    L.result = A.take<int>((size_t)B * Q);
    if (algo == DEFTET_PIT_BRUTE) {
        L.rec = A.take<float>((size_t)B * T * 32);
    } else {
        L.G = pick_G(T, Q);
        int xfine = kXFine;
        if (const char *e = getenv("DEFTET_PIT_XFINE")) {         // experiments only: x-refinement of the cells
            const int v = atoi(e);
            if (v >= 1 && v <= kMaxXFine) xfine = v;
        }
        L.Gx = L.G * xfine;
        const long long n = (long long)L.Gx * L.G * L.G + 1, R = (long long)L.G * L.G;
        L.cellStride = (n + 63) / 64 * 64;
        L.nRowBlk = (Q + kRowTile - 1) / kRowTile;
        if (L.nRowBlk > kMaxRowBlocks) L.nRowBlk = kMaxRowBlocks;
        if (L.nRowBlk < 1) L.nRowBlk = 1;
        L.chunkQ = ((Q + L.nRowBlk - 1) / L.nRowBlk + 255) / 256 * 256;
        L.bboxPart = A.take<float>((size_t)B * kBoxBlocks * 6);
        L.counters = A.take<int>((size_t)B * (8 + kOvfCap));  // 4 counters + 4 statistics words per shape, then the overflowed-tet lists
        L.gparam = A.take<float>((size_t)B * 12);
        L.cells = A.take<int>((size_t)B * L.cellStride);
        L.blockHist = A.take<int>((size_t)B * L.nRowBlk * R);
        L.rowTotal = A.take<int>((size_t)B * R);
        L.rowStart = A.take<int>((size_t)B * (R + 1));
        L.qkey = A.take<int2>((size_t)B * Q);
        L.rowSorted = A.take<float4>((size_t)B * Q);
        L.sortedQ = A.take<float4>((size_t)B * Q);
        L.irregT = A.take<int>((size_t)B * T);
        L.irregQ = A.take<int>((size_t)B * Q);
        L.deferT = A.take<int>((size_t)B * T);
    }
    L.bytes = align_up(A.off, 256);
    return L;
}

}  // namespace pit
}  // namespace deftet

using namespace deftet;
using namespace deftet::pit;

extern "C" size_t deftet_point_in_tet_workspace_bytes(int B, int T, int Q, int algo)
{
    if (B <= 0 || T < 0 || Q < 0) return 0;
    return make_layout(B, T, Q, algo, nullptr, 0).bytes;
}

extern "C" size_t deftet_point_in_tet_hits_ints(int B, int T, int Q)
{
    if (B <= 0 || T < 0 || Q < 0) return 0;
    return hit_list_off(B, T) + (size_t)B * Q;
}

static int pit_check(const float *tet, const float *pts, const float *cond, const float *bary, const float *pred, const float *occ,
                     const int32_t *hit_buf, int B, int T, int Q, int algo, const void *workspace)
{
    DEFTET_CHECK_ARG(!hit_buf || (((uintptr_t)hit_buf & 15) == 0 && algo != DEFTET_PIT_BRUTE), "hit_buf must be 16-byte aligned and needs a binned algo");
    DEFTET_CHECK_ARG((pred == nullptr) == (occ == nullptr), "pred and occ must be given together");
    DEFTET_CHECK_ARG(!occ || T > 0, "paste_occ needs at least one tet");
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0, "negative size (B=%d T=%d Q=%d)", B, T, Q);
    DEFTET_CHECK_ARG(algo == DEFTET_PIT_AUTO || algo == DEFTET_PIT_BRUTE || algo == DEFTET_PIT_STAGED || algo == DEFTET_PIT_ROWS ||
                         algo == DEFTET_PIT_FMA || algo == DEFTET_PIT_FMA2 || algo == DEFTET_PIT_GRP2 || algo == DEFTET_PIT_GRP4 ||
                         algo == DEFTET_PIT_GRP6 || algo == DEFTET_PIT_LDSB || algo == DEFTET_PIT_LDS || algo == DEFTET_PIT_EXACT,
                     "unknown algo %d", algo);
    if (T >= (1 << 24)) return set_error(DEFTET_ELIMIT, "n_tet=%d does not fit a float-encoded index (2^24)", T);
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds the grid-y limit 65535", B);
    if (Q >= (1 << 27)) return set_error(DEFTET_ELIMIT, "n_query=%d: 16-byte query records are addressed with 32-bit byte offsets (limit 2^27)", Q);
    if (B == 0 || Q == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts && cond, "null pts/cond pointer");
    DEFTET_CHECK_ARG(T == 0 || tet, "null tet pointer");
    DEFTET_CHECK_ARG(((uintptr_t)tet & 15) == 0, "tet must be 16-byte aligned");
    DEFTET_CHECK_ARG(!bary || ((uintptr_t)bary & 15) == 0, "bary must be 16-byte aligned");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    return DEFTET_OK;
}

// query side: resets + bounding box + counting sort of the queries into grid cells (depends on pts, B, T, Q only)
static int pit_prepare(const Layout &L, const float *pts, int B, int Q, hipStream_t st)
{
    const dim3 blk(256);
    const int R = L.G * L.G;
    DEFTET_LAUNCH(k_query_bbox, dim3(kBoxBlocks, B), blk, st, pts, Q, L.bboxPart, L.counters, L.result, B, (long long)B * Q);
    DEFTET_LAUNCH(k_row_count, dim3(L.nRowBlk, B), blk, st, pts, Q, L.bboxPart, L.gparam, L.G, L.Gx, L.nRowBlk, L.chunkQ, L.qkey,
                  L.blockHist, L.counters, L.irregQ);
    DEFTET_LAUNCH(k_row_colscan, dim3((R + 255) / 256, B), blk, st, L.blockHist, L.nRowBlk, R, L.rowTotal);
    DEFTET_LAUNCH(k_row_scatter, dim3(L.nRowBlk, B), blk, st, pts, Q, L.qkey, L.blockHist, L.rowTotal, L.rowStart, L.G, L.nRowBlk,
                  L.chunkQ, L.rowSorted);
    DEFTET_LAUNCH(k_row_fine, dim3((R + 3) / 4, B), blk, st, L.rowSorted, Q, L.gparam, L.G, L.Gx, L.rowStart, L.cellStride, L.cells,
                  L.sortedQ);
    return DEFTET_OK;
}

// tet side: traversal + finalize; consumes the prepared state (result sentinels, counters)
static int pit_scan(const Layout &L, const float *tet, const float *pts, float *cond, float *bary, const float *pred, float *occ,
                    int32_t *hit_buf, int B, int T, int Q, int algo, hipStream_t st)
{
    if ((algo == DEFTET_PIT_GRP2 && Q >= (1 << 30)) || (algo == DEFTET_PIT_GRP4 && Q >= (1 << 28)) || (algo == DEFTET_PIT_GRP6 && Q >= (1 << 26)))
        algo = DEFTET_PIT_FMA;                                       // query id + tet mask no longer fit one 32-bit entry
    const dim3 blk(256);
    const dim3 gq((Q + 255) / 256, B), gt((((T + 255) / 256 + 7) / 8) * 8, B);   // gt: multiple of 8 for the XCD mapping
    int *ucount = hit_buf ? hit_buf + hit_cnt_off(B, T) : nullptr;
    if (T > 0) {
        if (algo == DEFTET_PIT_ROWS) {
            DEFTET_LAUNCH(k_tet_scan_rows, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount);
        } else if (algo == DEFTET_PIT_FMA || algo == DEFTET_PIT_AUTO) {
            DEFTET_LAUNCH(k_tet_scan_fma<false>, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount, (const int *)nullptr);
        } else if (algo == DEFTET_PIT_FMA2) {
            DEFTET_LAUNCH(k_tet_scan_fma<true>, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount, (const int *)nullptr);
        } else if (algo == DEFTET_PIT_LDSB) {
            DEFTET_LAUNCH(k_tet_scan_lds<false>, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount);
        } else if (algo == DEFTET_PIT_LDS) {
            DEFTET_LAUNCH(k_tet_scan_lds<true>, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount);
        } else if (algo == DEFTET_PIT_GRP2 || algo == DEFTET_PIT_GRP4 || algo == DEFTET_PIT_GRP6) {
            const int K = algo == DEFTET_PIT_GRP2 ? 2 : (algo == DEFTET_PIT_GRP4 ? 4 : 6);
            const int ng = (T + K - 1) / K;
            const dim3 gg((((ng + 255) / 256 + 7) / 8) * 8, B);
            if (K == 2) {
                DEFTET_LAUNCH(k_tet_scan_grp<1>, gg, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                              L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount, L.deferT);
            } else if (K == 4) {
                DEFTET_LAUNCH(k_tet_scan_grp<2>, gg, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                              L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount, L.deferT);
            } else {
                DEFTET_LAUNCH(k_tet_scan_grp<3>, gg, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                              L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount, L.deferT);
            }
            // the groups with a mesh-order jump inside (normally a few per cent, possibly none), one tet per lane
            DEFTET_LAUNCH(k_tet_scan_fma<false>, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, (int *)nullptr, (const int *)L.deferT);
        } else if (algo == DEFTET_PIT_EXACT) {
            DEFTET_LAUNCH(k_tet_scan, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount);
        } else {
            DEFTET_LAUNCH(k_tet_scan_staged, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.cells, L.cellStride, L.sortedQ,
                          L.result, L.counters, L.irregT, (int4 *)hit_buf, pts, L.irregQ, ucount);
        }
    } else if (ucount) {
        DEFTET_HIP(hipMemsetAsync(ucount, 0, (size_t)B * 4, st));
    }
    DEFTET_LAUNCH(k_finalize, gq, blk, st, tet, pts, T, Q, L.result, cond, bary, pred, occ, (const int4 *)hit_buf, ucount,
                  hit_buf ? hit_buf + hit_list_off(B, T) : nullptr, L.counters, L.irregT);
    return DEFTET_OK;
}

extern "C" int deftet_point_in_tet_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                       float *occ, int32_t *hit_buf, int B, int T, int Q, int algo, void *workspace,
                                       size_t workspace_bytes, void *stream_)
{
    int rc = pit_check(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace);
    if (rc != DEFTET_OK || B == 0 || Q == 0) return rc;
    Layout L = make_layout(B, T, Q, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    hipStream_t st = as_stream(stream_);
    if (algo == DEFTET_PIT_BRUTE) {
        const dim3 blk(256), gq((Q + 255) / 256, B);
        if (T > 0) {
            long long n = (long long)B * T;
            DEFTET_LAUNCH(k_prep_records, dim3((unsigned)((n + 255) / 256)), blk, st, tet, n, L.rec);
        }
        DEFTET_LAUNCH(k_brute, gq, blk, st, L.rec, pts, T, Q, L.result);
        DEFTET_LAUNCH(k_finalize, gq, blk, st, tet, pts, T, Q, L.result, cond, bary, pred, occ, (const int4 *)nullptr, (int *)nullptr,
                      (int *)nullptr, (const int *)nullptr, L.irregT);
        return DEFTET_OK;
    }
    rc = pit_prepare(L, pts, B, Q, st);
    if (rc != DEFTET_OK) return rc;
    return pit_scan(L, tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, st);
}

// The same operator in two calls: the QUERY side (bounding box + counting sort: depends on pts and on
// the sizes only) can be enqueued ahead of time — e.g. on another stream while the previous step's
// backward is still running — and the TET side consumes it.  One prepare feeds exactly one scan
// (the scan uses up the result sentinels and counters the prepare resets); both must see the same
// pts, sizes, algo and workspace.
extern "C" int deftet_point_in_tet_prepare_f32(const float *pts, int B, int T, int Q, int algo, void *workspace,
                                               size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0 && B <= 65535, "bad size (B=%d T=%d Q=%d)", B, T, Q);
    DEFTET_CHECK_ARG(algo != DEFTET_PIT_BRUTE && algo >= 0 && algo <= DEFTET_PIT_EXACT, "prepare needs a binned algo (got %d)", algo);
    if (B == 0 || Q == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts, "null pts pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    Layout L = make_layout(B, T, Q, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    return pit_prepare(L, pts, B, Q, as_stream(stream_));
}

extern "C" int deftet_point_in_tet_scan_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                            float *occ, int32_t *hit_buf, int B, int T, int Q, int algo, void *workspace,
                                            size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(algo != DEFTET_PIT_BRUTE, "scan needs a binned algo");
    int rc = pit_check(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace);
    if (rc != DEFTET_OK || B == 0 || Q == 0) return rc;
    Layout L = make_layout(B, T, Q, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    return pit_scan(L, tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, as_stream(stream_));
}

// Diagnostics: copies the 8 int32 words per shape that the last forward on this workspace left behind —
// [0] irregular tets, [1] irregular queries, [2] hit-record overflow flag, [3] tets deferred by the grouped traversal,
// [4] groups re-scanned exactly, [5] tets re-scanned exactly (one-tet filter kernel), [6..7] unused — to host memory.
// Synchronises the stream.
extern "C" int deftet_point_in_tet_read_stats(const void *workspace, size_t workspace_bytes, int B, int T, int Q, int algo,
                                              int32_t *out_host, void *stream_)
{
    DEFTET_CHECK_ARG(workspace && out_host && B > 0 && algo != DEFTET_PIT_BRUTE, "bad argument");
    Layout L = make_layout(B, T, Q, algo, const_cast<void *>(workspace), workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small");
    hipStream_t st = as_stream(stream_);
    std::vector<int32_t> tmp((size_t)B * 8);
    DEFTET_HIP(hipMemcpyAsync(tmp.data(), L.counters, (size_t)B * 32, hipMemcpyDeviceToHost, st));
    DEFTET_HIP(hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k) {
            out_host[b * 8 + k] = tmp[(size_t)b * 4 + k];
            out_host[b * 8 + 4 + k] = tmp[(size_t)B * 4 + (size_t)b * 4 + k];
        }
    return DEFTET_OK;
}

#ifdef PIT_PHASE_TIMING
extern "C" int deftet_debug_phase_read(unsigned long long *out16, int reset)
{
    std::vector<unsigned long long> h((size_t)kPhaseWaves * 16);
    DEFTET_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(deftet::pit::g_phase), h.size() * sizeof(unsigned long long)));
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (size_t w = 0; w < (size_t)kPhaseWaves; ++w)
        for (int i = 0; i < 16; ++i) out16[i] += h[w * 16 + i];
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        DEFTET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(deftet::pit::g_phase), h.data(), h.size() * sizeof(unsigned long long)));
    }
    return DEFTET_OK;
}
#endif

extern "C" size_t deftet_point_in_tet_bwd_workspace_bytes(int B, int T, int Q)
{
    if (B <= 0 || T < 0 || Q < 0) return 0;
    return align_up((size_t)B * T * 4, 256) + align_up((size_t)B * Q * 4, 256) + align_up((size_t)B * 4, 256);
}

extern "C" int deftet_point_in_tet_bwd_f32(const float *tet, const float *pts, const float *cond, const float *grad_w,
                                           float *grad_tet, float *grad_pts, const float *grad_occ, float *grad_pred,
                                           const int32_t *hit_buf, int B, int T, int Q, int accumulate, void *workspace,
                                           size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0, "negative size");
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds 65535", B);
    DEFTET_CHECK_ARG((grad_occ == nullptr) == (grad_pred == nullptr), "grad_occ and grad_pred must be given together");
    hipStream_t st = as_stream(stream_);
    if (B == 0) return DEFTET_OK;
    if (grad_pts && Q > 0) DEFTET_HIP(hipMemsetAsync(grad_pts, 0, (size_t)B * Q * 12, st));   // also when T == 0: no tet, zero gradient
    if (T == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(grad_tet && ((uintptr_t)grad_tet & 15) == 0, "grad_tet null or not 16-byte aligned");
    if (Q == 0) {
        if (!accumulate) {
            DEFTET_HIP(hipMemsetAsync(grad_tet, 0, (size_t)B * T * 48, st));
            if (grad_pred) DEFTET_HIP(hipMemsetAsync(grad_pred, 0, (size_t)B * T * 4, st));
        }
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(tet && pts && cond && grad_w, "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)tet & 15) == 0 && ((uintptr_t)grad_w & 15) == 0, "tet/grad_w must be 16-byte aligned");
    if (hit_buf) {
        // fastest path: the forward's hit records (needs only kMissParts floats per shape of workspace)
        DEFTET_CHECK_ARG(((uintptr_t)hit_buf & 15) == 0, "hit_buf must be 16-byte aligned");
        float *missPart = nullptr;
        if (grad_pred) {
            DEFTET_CHECK_ARG(workspace && workspace_bytes >= (size_t)B * kMissParts * 4, "workspace needed for the miss sums");
            missPart = static_cast<float *>(workspace);
        }
        const int tblocks = (T + 255) / 256, nMissParts = tblocks < kMissParts ? tblocks : kMissParts;
        DEFTET_LAUNCH(k_bary_bwd_hits, dim3(tblocks, B), dim3(256), st, tet, pts, cond, grad_w, (const int4 *)hit_buf, T, Q,
                      grad_tet, grad_pts, accumulate, grad_occ, grad_pred, missPart, nMissParts);
        static_assert(kMissParts == 64, "k_bary_bwd_tail sums the partials with one wave");
        DEFTET_LAUNCH(k_bary_bwd_tail, dim3(64, B), dim3(256), st, tet, pts, cond, grad_w, hit_buf + hit_cnt_off(B, T),
                      hit_buf + hit_list_off(B, T), T, Q, grad_tet, grad_pts, grad_occ, grad_pred, missPart, nMissParts);
    } else if (workspace) {
        const size_t need = deftet_point_in_tet_bwd_workspace_bytes(B, T, Q);
        DEFTET_CHECK_ARG(workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0,
                         "backward workspace too small (%zu < %zu) or misaligned", workspace_bytes, need);
        char *w = static_cast<char *>(workspace);
        int *head = reinterpret_cast<int *>(w);
        int *next = reinterpret_cast<int *>(w + align_up((size_t)B * T * 4, 256));
        float *missSum = reinterpret_cast<float *>(w + align_up((size_t)B * T * 4, 256) + align_up((size_t)B * Q * 4, 256));
        DEFTET_HIP(hipMemsetAsync(head, 0xFF, (size_t)B * T * 4, st));      // -1 = empty list
        if (grad_pred) DEFTET_HIP(hipMemsetAsync(missSum, 0, (size_t)B * 4, st));
        DEFTET_LAUNCH(k_hit_link, dim3((Q + 256 * kLinkPer - 1) / (256 * kLinkPer), B), dim3(256), st, cond, T, Q, head, next,
                      grad_occ, missSum);
        DEFTET_LAUNCH(k_bary_bwd_gather, dim3((T + 255) / 256, B), dim3(256), st, tet, pts, grad_w, head, next, T, Q,
                      grad_tet, grad_pts, accumulate, grad_occ, missSum, grad_pred);
    } else {
        // no workspace: atomic scatter (slow on this chip; kept for callers that cannot provide one)
        if (!accumulate) DEFTET_HIP(hipMemsetAsync(grad_tet, 0, (size_t)B * T * 48, st));
        DEFTET_LAUNCH(k_bary_bwd, dim3((Q + 255) / 256, B), dim3(256), st, tet, pts, cond, grad_w, T, Q, grad_tet,
                      grad_pts);
        if (grad_pred) {
            if (!accumulate) DEFTET_HIP(hipMemsetAsync(grad_pred, 0, (size_t)B * T * 4, st));
            DEFTET_LAUNCH(k_paste_bwd, dim3((Q + 255) / 256, B), dim3(256), st, cond, grad_occ, grad_pred, T, Q);
        }
    }
    return DEFTET_OK;
}

extern "C" int deftet_paste_occ_fwd_f32(const float *pred, float *cond, float *out, int B, int T, int Q,
                                        int clamp_cond_inplace, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0 && B <= 65535, "bad size");
    if (B == 0 || Q == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(T > 0, "paste_occ needs at least one tet");
    DEFTET_CHECK_ARG(pred && cond && out, "null pointer");
    DEFTET_LAUNCH(k_paste_fwd, dim3((Q + 255) / 256, B), dim3(256), as_stream(stream_), pred, cond, out, T, Q,
                       clamp_cond_inplace);
    return DEFTET_OK;
}

extern "C" int deftet_paste_occ_bwd_f32(const float *cond, const float *grad_out, float *grad_pred, int B, int T, int Q,
                                        int zero_grad_pred, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0 && B <= 65535, "bad size");
    hipStream_t st = as_stream(stream_);
    if (zero_grad_pred && B > 0 && T > 0) {
        DEFTET_CHECK_ARG(grad_pred, "null grad_pred");
        DEFTET_HIP(hipMemsetAsync(grad_pred, 0, (size_t)B * T * 4, st));
    }
    if (B == 0 || Q == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(T > 0 && cond && grad_out && grad_pred, "null pointer / empty tet set");
    DEFTET_LAUNCH(k_paste_bwd, dim3((Q + 255) / 256, B), dim3(256), st, cond, grad_out, grad_pred, T, Q);
    return DEFTET_OK;
}
