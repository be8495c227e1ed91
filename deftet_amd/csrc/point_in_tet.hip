// point_in_tet.hip — A1 point-in-tet occupancy query (+ A1b weights/backward, paste_occ)
// for CDNA4 / gfx950.  See DESIGN.md section "A1" for the algorithm and its error analysis.
//
// Semantics follow /root/reference/layers/DefTet/check_condition_tetrahedron_base/
// check_condition_tet_for.cu:105-189: for every query the LOWEST tet index t whose four
// same-side tests agree (all true or all false), evaluated in fp32 in the source's
// operation order WITHOUT fused multiply-add; else -1.
//
// Design (tet-centric, not a translation of the one-thread-per-query scan):
//   1. queries are counting-sorted into a uniform Gx*G*G grid spanning their own bounding box
//      (slab = z cell; inside a slab (y cell, x cell) order), with a TRANSPOSED cell-start table
//      table[cz][cx][cy] so that the bounds of up to four y-adjacent cell runs are one 16-byte load;
//   2. one lane per tet: the lane computes the tet's four face planes once, walks the z slabs
//      overlapped by its (slightly enlarged) bounding box, flattens the <= 4 runs of a slab into
//      one candidate range and runs a certified fused plane filter on them; hits are combined
//      with atomicMin, which makes the result independent of evaluation order;
//   3. tets that fail a conditioning test ("irregular": tiny/flat/inverted-inconsistent,
//      non-finite, huge) are tested against ALL queries, and queries that are
//      non-finite/huge are tested against ALL tets, by two brute-force side paths, so
//      the result equals the reference scan for every input, not just for nice meshes.
//
// The traversal variants of rounds 1-2 (staged / rows / grouped / LDS / packed) live in
// tools/probes/legacy/point_in_tet_r02.hip and are built only by tools/probes/build_variant.sh.
//
// The whole file is compiled with -ffp-contract=off; the pragma below repeats that.
#pragma clang fp contract(off)

#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace deftet {
namespace pit {

constexpr float kBig = 1048576.0f;           // 2^20: coordinates beyond this go to the brute side paths
constexpr float kTau = 1.0f / 128.0f;        // regular tet: min |6V| >= tau * w^3
constexpr float kMargin = 1.0f / 64.0f;      // bounding-box enlargement, in units of w
constexpr float kWMin = 9.3132257e-10f;      // 2^-30
constexpr int kMiss = 0x7F7F7F7F;            // result sentinel (> any tet index)
constexpr int kMaxG = 112;                   // y/z cells per axis (slab key has 7 bits)
constexpr int kMaxBins = 15 * 1024;          // (cy, cx) counters of one slab quarter must fit k_slab_sort's LDS (60 KB)
constexpr int kHitOverflow = -2;            // hits[.].y: this tet has accepted queries that are not recorded

// "hit record" buffer written by the forward and consumed by the backward (int32 words):
//   [0, 2*B*T)            int2 per tet (round 6; int4 before): the first two queries the tet accepted (-1: none; the slots fill in
//                         order), or y == kHitOverflow: accepted queries that are not recorded.  kHitSpilled set in .x: up to
//                         four more are in the spill record.  71 % of the tets of BASELINE configs[2] accept nothing, 24 % one
//                         query, 4.1 % two, 0.5 % more: 8 bytes per tet instead of 16 take 16.5 MB out of the traversal's writes
//                         and out of the backward's reads (1.5 + 3.3 us, measured with a probe build before the format changed:
//                         profiles/r06_traversal_memory_ops.jsonl).
//   [2*B*T, +3*pad)       three words per shape (pad = B rounded up to kHitPad): [0, pad) number of uncovered queries,
//                         [pad, 2 pad) ticket of the backward's miss-sum reduction (zero between calls),
//                         [2 pad, 3 pad) flag: some uncovered entry belongs to a NaN/Inf/huge query
//   [.., + B*Q)           per shape: uncovered queries = hits that are NOT in their tet's record
//                         (tet overflowed / irregular tet / NaN-Inf-huge query)
//   [.., + 4*B*T)         int4 per tet: the SPILL record — accepted queries three to six (or -1); a seventh acceptance makes
//                         the tet "overflowed" (its hits are then carried by the uncovered list).  (With four slots in all, a
//                         dense query set — configs[1]: one query per tet on average — overflowed 130 tets per shape, and their
//                         ~650 uncovered hits cost k_finalize and the backward 20 us each.)  Untouched for tets that never spill.
constexpr int kHitPad = 64;
__host__ __device__ inline int hit_pad(int B) { return (B + kHitPad - 1) / kHitPad * kHitPad; }
__host__ __device__ inline size_t hit_cnt_off(int B, int T) { return (size_t)B * T * 2; }
__host__ __device__ inline size_t hit_list_off(int B, int T) { return (size_t)B * T * 2 + (size_t)3 * hit_pad(B); }
// second record of the tets that accepted three to six queries (int4 per tet, 16-byte aligned)
__host__ __device__ inline size_t hit_spill_off(int B, int T, int Q) { return (hit_list_off(B, T) + (size_t)B * Q + 3) / 4 * 4; }
constexpr int kHitSpilled = 1 << 30;       // flag in hits[.].x: up to four more accepted queries (or -1) are in the spill record
constexpr int kRecSlots = 2, kSpillSlots = 4;   // accepted queries a tet records: 2 + 4
// Tets whose hit record overflowed (> 6 accepted queries; ~1e-4 of the tets at BASELINE configs[2]) are also LISTED, so
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

// The record of a tet from its first n accepted queries hq[0 .. 5] (-1 beyond n); more than the record and its spill record
// hold: the overflow marker (and the tet is listed).  Plain stores (the traversal kernels have their own, tuned ones).
__device__ __forceinline__ void put_hit_record(int2 *hits, int4 *spill, size_t idx, const int (&hq)[kRecSlots + kSpillSlots], int n,
                                               int *counters, int nB, int b, int t)
{
    if (n > kRecSlots + kSpillSlots || (n > kRecSlots && !spill)) {
        hits[idx] = make_int2(-1, kHitOverflow);
        note_overflow(counters, nB, b, t);
        return;
    }
    if (n > kRecSlots) spill[idx] = make_int4(hq[2], hq[3], hq[4], hq[5]);
    hits[idx] = make_int2(n > kRecSlots ? (hq[0] | kHitSpilled) : hq[0], hq[1]);
}

#ifdef PIT_PHASE_TIMING
// Diagnostic build only (tools/probes/build_variant.sh … -DPIT_PHASE_TIMING): per-phase wall cycles of the traversal
// kernel.  Lane 0 of every wave adds its s_memtime deltas to a slot of its own (no atomics: same-address atomics from
// 32 k waves would dominate what is being measured); deftet_debug_phase_read sums the slots.
constexpr int kPhaseWaves = 1 << 17;
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
// (the argument is evaluated by the whole wave BEFORE the lane-0 branch: written inside it, a ballot in `v` saw lane 0 only —
// the "0.04 ungrouped lanes per wave" of round 4 were 1.9)
#define PHASE_COUNT(i, v)                                                              \
    do {                                                                               \
        const unsigned long long pc_ = (unsigned long long)(v);                        \
        if ((threadIdx.x & 63) == 0) g_phase[ph_w_][i] += pc_;                         \
    } while (0)
// when every wave of the traversal starts and ends (constant-rate wall clock, 100 MHz): the launch's occupancy over time
__device__ unsigned long long g_span[kPhaseWaves][2];
#define SPAN_MARK(k)                                                                                                     \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0) g_span[ph_w_][k] = wall_clock64();                                                  \
    } while (0)
// why lanes end up outside every footprint group (lanes per reason, summed over the waves of all launches)
__device__ unsigned long long g_reason[8];
#define REASON_COUNT(i, mask)                                                                                        \
    do {                                                                                                             \
        const unsigned long long rm_ = (mask);                                                                       \
        if (rm_ != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&g_reason[i], (unsigned long long)__popcll(rm_));     \
    } while (0)
#else
#define SPAN_MARK(k)
#define REASON_COUNT(i, mask)
#define PHASE_DECL
#define PHASE_MARK(i)
#define PHASE_COUNT(i, v)
#endif

// ------------------------------------------------------------------------------------
// exact predicate pieces (check_condition_tet_for.cu:105-121, :172-176)
// ------------------------------------------------------------------------------------
struct Planes {
    float n[4][3];   // (b-a) x (c-a) for the four vertex orderings
    float a[4][3];   // base vertex of each ordering (= vertex i)
    unsigned sv;     // bit i: dotv4_i > 0
    float dv[4];     // dotv4_i
};

// cross product in the source's operation order (:63-65)
__device__ __forceinline__ void cross_ref(float r1x, float r1y, float r1z, float r2x, float r2y, float r2z, float *n)
{
    n[0] = r1y * r2z - r1z * r2y;
    n[1] = r1z * r2x - r1x * r2z;
    n[2] = r1x * r2y - r1y * r2x;
}

// The four orderings (a,b,c,d),(b,a,d,c),(c,d,a,b),(d,c,b,a) (:172-175) only need the SIX edge vectors of the tet:
// fl(x - y) = -fl(y - x) exactly, and a product / difference of negated operands is the exact negation (or the same
// value), so feeding the shared edges with the right signs reproduces every intermediate of the four source-order
// evaluations bit for bit — with 18 subtractions instead of 36.
__device__ __forceinline__ void make_planes(const float *v /*12*/, Planes &P)
{
    const float abx = v[3] - v[0], aby = v[4] - v[1], abz = v[5] - v[2];
    const float acx = v[6] - v[0], acy = v[7] - v[1], acz = v[8] - v[2];
    const float adx = v[9] - v[0], ady = v[10] - v[1], adz = v[11] - v[2];
    const float bcx = v[6] - v[3], bcy = v[7] - v[4], bcz = v[8] - v[5];
    const float bdx = v[9] - v[3], bdy = v[10] - v[4], bdz = v[11] - v[5];
    const float cdx = v[9] - v[6], cdy = v[10] - v[7], cdz = v[11] - v[8];
    // i = 0: a=v0 b=v1 c=v2 d=v3: r1 = ab, r2 = ac, d-a = ad
    cross_ref(abx, aby, abz, acx, acy, acz, P.n[0]);
    P.dv[0] = P.n[0][0] * adx + P.n[0][1] * ady + P.n[0][2] * adz;                     // :115
    // i = 1: a=v1 b=v0 c=v3 d=v2: r1 = -ab, r2 = bd, d-a = bc
    cross_ref(-abx, -aby, -abz, bdx, bdy, bdz, P.n[1]);
    P.dv[1] = P.n[1][0] * bcx + P.n[1][1] * bcy + P.n[1][2] * bcz;
    // i = 2: a=v2 b=v3 c=v0 d=v1: r1 = cd, r2 = -ac, d-a = -bc
    cross_ref(cdx, cdy, cdz, -acx, -acy, -acz, P.n[2]);
    P.dv[2] = P.n[2][0] * -bcx + P.n[2][1] * -bcy + P.n[2][2] * -bcz;
    // i = 3: a=v3 b=v2 c=v1 d=v0: r1 = -cd, r2 = -bd, d-a = -ad
    cross_ref(-cdx, -cdy, -cdz, -bdx, -bdy, -bdz, P.n[3]);
    P.dv[3] = P.n[3][0] * -adx + P.n[3][1] * -ady + P.n[3][2] * -adz;
    P.sv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        P.a[i][0] = v[3 * i]; P.a[i][1] = v[3 * i + 1]; P.a[i][2] = v[3 * i + 2];
        P.sv |= (P.dv[i] > 0 ? 1u : 0u) << i;                                           // :119
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
constexpr float kErrScale = 4.76837158203125e-07f;      // 8 u = 2^-21
constexpr float kErrAbs = 7.5231638e-37f;              // 2^-120
constexpr int kGridWords = 20;                         // floats of grid parameters per shape

struct Grid {
    float o[3], inv[3], lo[3], hi[3];
    float pe[3];     // kErrScale * max(|lo|, |hi|): the query-side part of the filter's error radius
    float cs[3];     // upper bound of a cell's edge (1.001 * extent / cells, >= 1e-30): k_tet_scan_wave's candidate-distance bound
    int wide;        // a tet whose cell box holds more cells than this is WIDE (wide_tet below); set by k_slab_local, which knows Q
};
constexpr int kBoxBlocks = 64;

// grid parameters from the box of the regular queries (lo > hi: none seen)
__device__ __forceinline__ Grid make_grid(const float *lo, const float *hi, int G, int Gx)
{
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
        g.pe[k] = kErrScale * fmaxf(fabsf(l), fabsf(h));
        // n cells of axis k span at most n * cs[k]: with inv > 0 a cell is ext / cells wide up to the three roundings of
        // cell_of (<< 0.1 %); with inv == 0 every regular query sits in cell 0 and the box is <= 1e-30 wide
        g.cs[k] = fmaxf(1.001f * ((h - l) / (float)(k == 0 ? Gx : G)), 1e-30f);
    }
    g.wide = 0x7FFFFFFF;                           // (k_slab_local publishes the real threshold)
    return g;
}

// reduce the per-block query boxes of one shape into grid parameters; called by every wave of
// k_slab_count (64 partials, a few shuffles) so that no separate launch is needed
__device__ __forceinline__ void reduce_box(const float *__restrict__ part, int nPart, float *lo, float *hi)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
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
}
__device__ __forceinline__ Grid reduce_grid(const float *__restrict__ part, int nPart, int G, int Gx)
{
    float lo[3], hi[3];
    reduce_box(part, nPart, lo, hi);
    return make_grid(lo, hi, G, Gx);
}

// The grid of a call that was handed a box instead of measuring its queries' own (query_box_in of the *_ex_* entry points: the
// sampler's box, or the box an earlier call measured).  The box is a HINT: it is enlarged by 1/32 of its extent per side, the
// grid spans it, and a query outside it is not binned but listed with the NaN / Inf / huge ones and tested against every tet
// by the side path, so the result is exact for any box — a box that fits saves the launch that measures (k_query_bbox),
// one that does not costs time.  A box that is not a box (NaN, lo > hi) holds no query.
__device__ __forceinline__ Grid hint_grid(const float *__restrict__ box, int G, int Gx)
{
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float l = box[k], h = box[3 + k];
        const float e = (h - l) * (1.0f / 32.0f);
        const bool ok = h >= l && fabsf(l) <= kBig && fabsf(h) <= kBig;      // (NaN fails)
        lo[k] = ok ? fmaxf(l - e, -kBig) : INFINITY;
        hi[k] = ok ? fminf(h + e, kBig) : -INFINITY;
    }
    return make_grid(lo, hi, G, Gx);
}
// a query the grid bins: regular coordinates inside the grid's box (always true for the box measured from the queries themselves)
__device__ __forceinline__ bool query_binned(float x, float y, float z, const Grid &g)
{
    return x >= g.lo[0] && x <= g.hi[0] && y >= g.lo[1] && y <= g.hi[1] && z >= g.lo[2] && z <= g.hi[2];
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
        g.pe[k] = gp[12 + k];
        g.cs[k] = gp[15 + k];
    }
    g.wide = __float_as_int(gp[18]);
    return g;
}

// monotone non-decreasing in x for fixed (o, inv >= 0): rounding, clamp and truncation of a non-negative value are
// monotone.  (Truncation instead of floor: both give 0 for every value the clamp maps to [0, 1).)
__device__ __forceinline__ int cell_of(float x, float o, float inv, int G)
{
    const float f = __builtin_amdgcn_fmed3f((x - o) * inv, 0.f, (float)(G - 1));
    return (int)f;
}

// A WIDE tet: its cell box holds so large a part of the query grid that its lane would walk thousands of candidates on its own
// while the rest of the chip has nothing to do — queries packed into a box the size of a tet (or smaller: every tet that
// touches the box then covers ALL cells) cost 59 ms that way where brute force needs 16 (tools/probes/regime_probe.py).  Such a
// tet is handed to k_finalize with the irregular ones: every QUERY lane tests it with the reference predicate, n_query tests in
// parallel instead of a serial walk, and "lowest index" is decided there as for any irregular tet.  The estimate is the cell
// count times the mean occupancy of a cell (the grid spans the queries' own box); the bound keeps both sides cheap: a lane
// never walks more than ~max(4096, Q/32) candidates, and at most ~8 x 32 tets of a shape can be that wide.
// (An axis along which all queries coincide has one occupied cell layer, inv == 0: it does not dilute the occupancy.)
// The threshold is a number of CELLS, worked out once per shape by the thread that publishes the grid (k_slab_local) and read
// from the grid parameters: the traversal pays two multiplications and a compare per tet.
__device__ __forceinline__ int wide_cells(const Grid &g, int Q, int G, int Gx)
{
    const float layers = (g.inv[0] > 0.f ? (float)Gx : 1.0f) * (g.inv[1] > 0.f ? (float)G : 1.0f) * (g.inv[2] > 0.f ? (float)G : 1.0f);
    const float cells = fmaxf(4096.0f, (float)Q * (1.0f / 32.0f)) * layers / fmaxf((float)Q, 1.0f);
    return cells >= 2.0e9f ? 0x7FFFFFFF : (int)cells;                 // (few queries: no tet is wide)
}
__device__ __forceinline__ bool wide_tet(int cx0, int cx1, int cy0, int cy1, int cz0, int cz1, const Grid &g)
{
    return (cx1 - cx0 + 1) * ((cy1 - cy0 + 1) * (cz1 - cz0 + 1)) > g.wide;   // <= 128 x 64 x 64 cells: no overflow
}

__device__ __forceinline__ bool query_regular(float x, float y, float z)
{
    return fabsf(x) <= kBig && fabsf(y) <= kBig && fabsf(z) <= kBig;   // NaN fails
}

// Transposed cell-start table of one shape: entry (cz, cx, cy) = position in sortedQ of the first query of cell
// (cx, cy, cz); cx = Gx is the end of row (cz, cy); Gp = G + 3 entries per (cz, cx) so that a 4-wide read starting at any
// cy < G stays inside its (cz, cx) line.
__host__ __device__ inline int table_pitch(int G) { return G + 3; }
__device__ __forceinline__ unsigned table_off(int cz, int cx, int cy, int Gx, int Gp)
{
    return (unsigned)((cz * (Gx + 1) + cx) * Gp + cy);
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
    // kBoxKeep loads in flight per thread: left as a plain strided loop the compiler waits for each 12-byte load before it
    // issues the next (seven dependent trips to HBM per thread at 100,000 queries: 10.5 us for 9.6 MB).  The clamped
    // duplicates of the last query change no minimum.
    constexpr int kBoxKeep = 8;
    const int stride = gridDim.x * blockDim.x;
    for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < Q; q0 += stride * kBoxKeep) {
        float x[kBoxKeep], y[kBoxKeep], z[kBoxKeep];
#pragma unroll
        for (int k = 0; k < kBoxKeep; ++k) {
            const float *pq = p + (size_t)min(q0 + k * stride, Q - 1) * 3;
            x[k] = pq[0]; y[k] = pq[1]; z[k] = pq[2];
        }
#pragma unroll
        for (int k = 0; k < kBoxKeep; ++k)
            if (query_regular(x[k], y[k], z[k])) {                 // irregular queries are listed by k_slab_local
                lo[0] = fminf(lo[0], x[k]); lo[1] = fminf(lo[1], y[k]); lo[2] = fminf(lo[2], z[k]);
                hi[0] = fmaxf(hi[0], x[k]); hi[1] = fmaxf(hi[1], y[k]); hi[2] = fmaxf(hi[2], z[k]);
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

// Placement of a (chunk, shape) grid for the kernels whose gathers go to PER-SHAPE arrays in arbitrary order (the
// queries' coordinates, weights' gradients, winners, predictions): workgroup i is observed to run on XCD i % 8, so
// with shape = i % B all workgroups of an XCD work on the same shape(s) and the gathered arrays of a shape (3.6 MB in
// the backward at 100,000 queries) stay in that XCD's 4 MiB L2, instead of every L2 pulling every shape's arrays.
// Returns (shape, chunk).  Speed only; any placement is correct.
__device__ __forceinline__ int2 shape_block(int pin)
{
    if (!pin) return make_int2(blockIdx.y, blockIdx.x);
    const int L = blockIdx.y * gridDim.x + blockIdx.x, nB = gridDim.y;
    return make_int2(L % nB, L / nB);
}
// ... when a shape's gathered arrays fit an L2.  Beyond that the shapes are better taken one after the other by the
// whole chip (the arrays of ONE shape then live in the 256 MiB Infinity Cache; eight at a time do not: configs[3], one
// million queries per shape, backward 153 -> 165 us with the pinned placement; configs[1]: 33.4 -> 22.1 us, configs[2]:
// 62.7 -> 60.1 us, k_finalize 30.5 -> 27.5 us).
static inline int pin_shapes(int Q) { return (size_t)Q * 36 <= ((size_t)4 << 20); }

// ------------------------------------------------------------------------------------
// Counting sort of the regular queries into grid cells WITHOUT global atomics.
// (Round-1 history: one returning global atomicAdd per query = 800 k fabric transactions = 41 us;
// random-address global atomics run at ~23-26 G/s chip-wide at ANY scope, tools/probes/.)
// Two levels, both with LDS atomics only, two launches:
//   k_slab_local: every workgroup sorts ITS chunk of <= kRowTile queries by first-level bin (slab cz, y-eighth) into its
//         own segment of `localQ` and publishes the exclusive bin prefix of the chunk (pre[bin][block]);
//   k_slab_sort:  one workgroup per (slab, shape) gathers the slab's runs from all chunk segments (a run per chunk,
//         contiguous), counts per (cy, cx) cell in LDS, scans, writes its plane of the transposed cell-start table with
//         coalesced stores and places the queries.
// (Rounds 1-2 sorted into (cz, cy) ROWS with G^2-bin histograms, a column-scan kernel over [nblk][G^2] words, a global
// scatter and one wave per row: five launches, 44 us at configs[2].  The first slab version of round 3 kept a global
// scatter between the levels and lost 12 us in it to the column sums of a [nblk][bins] table — a chain of memory
// latencies; here no workgroup ever needs another one's histogram before the second level.)
// The order of queries inside a cell is arbitrary (as it was with global atomics); nothing
// downstream depends on it except which four accepted queries a hit record keeps.
// ------------------------------------------------------------------------------------
constexpr int kMaxRowBlocks = 256;   // chunks per shape (k_slab_sort handles one run per chunk with one thread each)
#ifndef PIT_ROWTILE
#define PIT_ROWTILE 2048
#endif
constexpr int kRowTile = PIT_ROWTILE;  // smallest query chunk per workgroup
constexpr int kLocalKeep = kRowTile / 256;   // queries a thread keeps in registers between the two passes
// First-level bins: (slab, y-eighth of the cell rows).  The second level works on y-quarters of a slab (two adjacent
// sub-bins = one contiguous run per chunk): a quarter's (cy, cx) counters need ~10 KB of LDS, so several workgroups
// share a compute unit (a whole slab needs 40 KB at configs[2], and only 64 KB per CU are handed out to kernels that do
// not opt into the large-LDS mode: one workgroup per CU, 17 us instead of 6).
constexpr int kSub = 8;
constexpr int kParts = 4;                  // second-level workgroups per slab
constexpr int kSubPerPart = kSub / kParts;
constexpr int kMaxBin1 = kMaxG * kSub;     // 896

// HINT (query_box_in given): the grid comes from the box the caller handed in (hint_grid) and k_query_bbox is not launched,
// so this kernel also does what that launch did besides measuring — the result sentinels of its chunk — and measures for
// the NEXT call: the box of its chunk's regular queries goes to partOut[chunk], k_slab_sort reduces the chunks' boxes into
// query_box_out.  Queries the grid does not bin (outside the hint, NaN / Inf / huge) are listed per chunk without a global
// counter (nobody has zeroed one): irregQ[q0 + i], i < irrCnt[chunk]; k_slab_sort compacts the lists and sets the counters.
// irrCnt[nShapes * kMaxRowBlocks + chunk] counts the chunk's regular queries OUTSIDE the hint on their own (query_box_misses).
template <bool HINT>
__global__ __launch_bounds__(256) void k_slab_local(const float *__restrict__ pts, int Q, const float *__restrict__ bboxPart,
                                                    float *gparam, int G, int Gx, int nblk, int nblkPad, int chunkQ,
                                                    float4 *localQ, int *pre, int *counters, int *irregQ,
                                                    const float *__restrict__ boxIn, float *partOut, int *irrCnt, int *result)
{
    __shared__ int hist[kMaxBin1 + 1];
    __shared__ int wtot[4];
    __shared__ int s_irr, s_out;
    __shared__ float s_box[4][6];
    const int b = blockIdx.y, blk = blockIdx.x, R1 = G * kSub, tid = threadIdx.x;
    const int q0 = blk * chunkQ, q1 = min(Q, q0 + chunkQ);
    const bool keep = chunkQ <= kRowTile;                          // launch-uniform: the chunk fits the register file
    float3 kp[kLocalKeep];
    if (keep) {                                                    // issued BEFORE the box reduction: its loads and shuffles wait
#pragma unroll                                                     // on nothing these need
        for (int k = 0; k < kLocalKeep; ++k) {                     // unconditional (clamped) loads: all in flight at once
            const float *p = pts + ((size_t)b * Q + max(min(q0 + tid + k * 256, q1 - 1), 0)) * 3;
            kp[k] = make_float3(p[0], p[1], p[2]);
        }
    }
    if (HINT) {
        for (int q = q0 + tid; q < q1; q += 256) result[(size_t)b * Q + q] = kMiss;
        if (tid == 0) { s_irr = 0; s_out = 0; }
    }
    const Grid g = HINT ? hint_grid(boxIn + (size_t)b * 6, G, Gx) : reduce_grid(bboxPart + (size_t)b * kBoxBlocks * 6, kBoxBlocks, G, Gx);
    float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};   // HINT: box of this chunk's regular queries
    auto box_add = [&](float x, float y, float z) {
        if (HINT && query_regular(x, y, z)) {
            blo[0] = fminf(blo[0], x); blo[1] = fminf(blo[1], y); blo[2] = fminf(blo[2], z);
            bhi[0] = fmaxf(bhi[0], x); bhi[1] = fmaxf(bhi[1], y); bhi[2] = fmaxf(bhi[2], z);
        }
    };
    // a query the grid does not bin: with a measured box that is a NaN / Inf / huge one, with a hint also one outside the hint
    auto binned = [&](float x, float y, float z) { return HINT ? query_binned(x, y, z, g) : query_regular(x, y, z); };
    auto list_irregular = [&](int q, float x, float y, float z) {
        if (HINT) {
            irregQ[(size_t)b * Q + q0 + atomicAdd(&s_irr, 1)] = q;                            // LDS counter, the chunk's own segment
            if (query_regular(x, y, z)) atomicAdd(&s_out, 1);                                 // a miss of the hint, not a NaN / Inf / huge query
        } else {
            irregQ[(size_t)b * Q + atomicAdd(&counters[b * 4 + 1], 1)] = q;
        }
    };
    if (blk == 0 && tid == 0) {                                    // publish for k_slab_sort / the traversal
        float *gp = gparam + b * kGridWords;                       // (constant indices: a dynamic one sends g through LDS — 13 us)
#pragma unroll
        for (int k = 0; k < 3; ++k) { gp[k] = g.o[k]; gp[3 + k] = g.inv[k]; gp[6 + k] = g.lo[k]; gp[9 + k] = g.hi[k]; gp[12 + k] = g.pe[k]; gp[15 + k] = g.cs[k]; }
        gp[18] = __int_as_float(wide_cells(g, Q, G, Gx));
        gp[19] = HINT ? 1.0f : 0.f;                                  // k_finalize: with a measured box every regular query is binned (no test needed)
    }
    for (int i = tid; i <= R1; i += 256) hist[i] = 0;
    __syncthreads();
    const float rcpG = 1.0f / (float)G;
    int kbin[kLocalKeep], krank[kLocalKeep];
    auto bin_of = [&](float x, float y, float z) {                 // (cz, y-eighth of the CELL row: sub = (cy * 8) / G exactly,
        const int cy = cell_of(y, g.o[1], g.inv[1], G);            //  +0.5 keeps the quotient away from the integers)
        return cell_of(z, g.o[2], g.inv[2], G) * kSub + (int)(((float)(cy * kSub) + 0.5f) * rcpG);
    };
    if (keep) {
#pragma unroll
        for (int k = 0; k < kLocalKeep; ++k) {
            const int q = q0 + tid + k * 256;
            kbin[k] = -1;
            if (q < q1) {
                box_add(kp[k].x, kp[k].y, kp[k].z);
                if (binned(kp[k].x, kp[k].y, kp[k].z)) {
                    kbin[k] = bin_of(kp[k].x, kp[k].y, kp[k].z);
                    krank[k] = atomicAdd(&hist[kbin[k]], 1);                          // LDS
                } else {                                           // NaN / Inf / huge (/ outside the hint): tested by every tet lane at the end of the traversal
                    list_irregular(q, kp[k].x, kp[k].y, kp[k].z);
                }
            }
        }
    } else {
        for (int q = q0 + tid; q < q1; q += 256) {
            const float *p = pts + ((size_t)b * Q + q) * 3;
            const float x = p[0], y = p[1], z = p[2];
            box_add(x, y, z);
            if (binned(x, y, z)) atomicAdd(&hist[bin_of(x, y, z)], 1);
            else list_irregular(q, x, y, z);
        }
    }
    if (HINT) {                                                    // the chunk's box for the next call's hint
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                blo[k] = fminf(blo[k], __shfl_xor(blo[k], off));
                bhi[k] = fmaxf(bhi[k], __shfl_xor(bhi[k], off));
            }
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { s_box[tid >> 6][k] = blo[k]; s_box[tid >> 6][3 + k] = bhi[k]; }
        }
    }
    __syncthreads();
    if (HINT) {
        if (tid < 6) {
            float v = s_box[0][tid];
            for (int i = 1; i < 4; ++i) v = tid < 3 ? fminf(v, s_box[i][tid]) : fmaxf(v, s_box[i][tid]);
            partOut[((size_t)b * kMaxRowBlocks + blk) * 6 + tid] = v;
        }
        if (tid == 0) {
            irrCnt[b * kMaxRowBlocks + blk] = s_irr;
            irrCnt[(gridDim.y + b) * kMaxRowBlocks + blk] = s_out;
        }
    }
    {   // exclusive scan of the R1 counts (<= 4 consecutive bins per thread); hist[R1] = number of regular queries
        constexpr int kPer = (kMaxBin1 + 255) / 256;               // 4
        const int per = (R1 + 255) / 256, i0 = tid * per;
        int n[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            n[j] = (j < per && i0 + j < R1) ? hist[i0 + j] : 0;
            sum += n[j];
        }
        int incl = sum;
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        int run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < w) run += wtot[k];
        int *col = pre + (size_t)b * (R1 + 1) * nblkPad + blk;     // pre[bin][chunk]: a slab workgroup reads rows of it
#pragma unroll
        for (int j = 0; j < kPer; ++j)
            if (j < per && i0 + j < R1) {
                hist[i0 + j] = run;
                col[(size_t)(i0 + j) * nblkPad] = run;
                run += n[j];
            }
        if (tid == 255) col[(size_t)R1 * nblkPad] = run;           // thread 255 holds the grand total after its own bins
    }
    __syncthreads();
    float4 *dst = localQ + (size_t)b * Q + q0;
    if (keep) {
#pragma unroll
        for (int k = 0; k < kLocalKeep; ++k)
            if (kbin[k] >= 0) dst[hist[kbin[k]] + krank[k]] = make_float4(kp[k].x, kp[k].y, kp[k].z, __int_as_float(q0 + tid + k * 256));
    } else {
        for (int q = q0 + tid; q < q1; q += 256) {
            const float *p = pts + ((size_t)b * Q + q) * 3;
            const float x = p[0], y = p[1], z = p[2];
            if (binned(x, y, z)) dst[atomicAdd(&hist[bin_of(x, y, z)], 1)] = make_float4(x, y, z, __int_as_float(q));
        }
    }
}

// One workgroup per (slab, y-quarter, shape).  Thread k < nblk owns the part's run in chunk k (pre[.][k] gives its offset
// and length); the block sums the offsets (= queries in lower bins = the part's start in sortedQ) and scans the lengths.
// Then: gather, count per (cy, cx) cell in LDS, exclusive scan in (cy, cx) order, placement, and the part's rows of the
// slab's plane of the transposed cell-start table.  Parts of up to kSortThreads * kSortKeep queries (the usual case) keep
// their queries, cells and ranks in registers between the counting and the placement pass; larger ones gather twice.
#ifndef PIT_SORT_THREADS
#define PIT_SORT_THREADS 256
#endif
constexpr int kSortThreads = PIT_SORT_THREADS;
constexpr int kSortKeep = 1024 / kSortThreads;                        // queries a thread keeps in registers
constexpr int kRunsPer = kMaxRowBlocks / kSortThreads;               // chunk runs per thread (consecutive chunks)
static_assert(kRunsPer * kSortThreads == kMaxRowBlocks && kSortThreads % 64 == 0, "every chunk run has a thread");
// boxPart / boxOut (query_box_out): the workgroup of a shape's first slab part also reduces the nPart per-block boxes of the
// queries (k_query_bbox's, or k_slab_local<true>'s) into the box the caller gets back.  hint: k_slab_local<true> ran instead
// of k_query_bbox — the same workgroup compacts the chunks' lists of unbinned queries (forward copy: a list's place in the
// compact list never lies behind its own segment) and sets the counters the traversal and k_finalize read.
__global__ __launch_bounds__(kSortThreads) void k_slab_sort(const float4 *__restrict__ localQ, int Q, const float *__restrict__ gparam,
                                                            int G, int Gx, const int *__restrict__ pre, int nblk, int nblkPad,
                                                            int chunkQ, long long cellStride, int *table, float4 *sortedQ, int pin,
                                                            const float *__restrict__ boxPart, int nPart, int partStride, float *boxOut,
                                                            int hint, const int *__restrict__ irrCnt, int *irregQ, int *counters,
                                                            int *missOut)
{
    if (shape_block(pin).y == 0 && (boxOut || hint)) {               // (block-uniform; before the sort proper: nothing below depends on it)
        const int b0 = shape_block(pin).x, nB = gridDim.y;
        if (boxOut && threadIdx.x < 64) {
            float lo[3], hi[3];
            reduce_box(boxPart + (size_t)b0 * partStride * 6, nPart, lo, hi);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { boxOut[b0 * 6 + k] = lo[k]; boxOut[b0 * 6 + 3 + k] = hi[k]; }   // lo > hi: no regular query
            }
        }
        if (hint && threadIdx.x < 64) {
            // One wave, 64 entries per step.  A list's place in the compact list never lies behind its own segment (n <= c *
            // chunkQ) and every step's loads return before its stores leave (the stores carry the loaded values), so the copy
            // may overlap its source.
            int n = 0, nOut = 0;
            for (int c = 0; c < nblk; ++c) {
                const int cnt = irrCnt[b0 * kMaxRowBlocks + c];
                nOut += irrCnt[(nB + b0) * kMaxRowBlocks + c];
                const int *src = irregQ + (size_t)b0 * Q + (size_t)c * chunkQ;
                for (int i = (int)threadIdx.x; i < cnt; i += 64) {
                    const int v = src[i];
                    irregQ[(size_t)b0 * Q + n + i] = v;
                }
                n += cnt;
            }
            if (threadIdx.x == 0) {
                counters[b0 * 4 + 0] = 0; counters[b0 * 4 + 1] = n; counters[b0 * 4 + 2] = 0; counters[b0 * 4 + 3] = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) counters[nB * 4 + b0 * 4 + k] = 0;
                if (missOut) missOut[b0] = nOut;                     // (may be host-mapped memory: one posted store per shape)
            }
        }
    }
    extern __shared__ __attribute__((aligned(16))) int cnt[];       // [rows of the part][GxP] counts -> starts (-> placement cursors)
    __shared__ int wsum[kSortThreads / 64], wsum2[kSortThreads / 64];
    __shared__ int runStart[kMaxRowBlocks + 1], runSrc[kMaxRowBlocks];
    const int2 sb = shape_block(pin);                               // the runs are gathered from all over the shape's localQ
    const int b = sb.x, cz = sb.y / kParts, part = sb.y % kParts, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int GxP = Gx | 1;                                         // odd pitch: the transposed read below is conflict-free
    // cell rows of this part: (cy * kSub) / G in [part * kSubPerPart, (part + 1) * kSubPerPart)
    const int cyLo = (part * kSubPerPart * G + kSub - 1) / kSub, cyHi = ((part + 1) * kSubPerPart * G + kSub - 1) / kSub;
    const int nb = (cyHi - cyLo) * GxP, R1 = G * kSub;
    const int binLo = cz * kSub + part * kSubPerPart;
    PHASE_DECL;
    SPAN_MARK(0);
    // runs of this part, one per chunk
    int a[kRunsPer], len[kRunsPer], asum = 0, lsum = 0;
#pragma unroll
    for (int j = 0; j < kRunsPer; ++j) {
        const int r = tid * kRunsPer + j;
        a[j] = 0; len[j] = 0;
        if (r < nblk) {
            const int *row = pre + (size_t)b * (R1 + 1) * nblkPad;
            a[j] = row[(size_t)binLo * nblkPad + r];
            len[j] = row[(size_t)(binLo + kSubPerPart) * nblkPad + r] - a[j];
        }
        asum += a[j];
        lsum += len[j];
    }
    int s0, n;
    {
        int ia = asum, il = lsum;                                   // inclusive scans over the threads
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int ta = __shfl_up(ia, off), tl = __shfl_up(il, off);
            if (lane >= off) { ia += ta; il += tl; }
        }
        if (lane == 63) { wsum[w] = ia; wsum2[w] = il; }
        __syncthreads();
        int ta = 0, tl = 0, before = 0;
#pragma unroll
        for (int k = 0; k < kSortThreads / 64; ++k) {
            ta += wsum[k];
            tl += wsum2[k];
            if (k < w) before += wsum2[k];
        }
        s0 = ta;
        n = tl;
        int at = before + il - lsum;
#pragma unroll
        for (int j = 0; j < kRunsPer; ++j) {
            const int r = tid * kRunsPer + j;
            if (r < nblk) {
                runStart[r] = at;
                runSrc[r] = r * chunkQ + a[j];
            }
            at += len[j];
        }
        if (tid == 0) runStart[nblk] = n;
    }
    for (int i = tid; i < ((nb + 3) & ~3); i += kSortThreads) cnt[i] = 0;
    __syncthreads();
    const float ox = gparam[b * kGridWords + 0], oy = gparam[b * kGridWords + 1];
    const float ix = gparam[b * kGridWords + 3], iy = gparam[b * kGridWords + 4];
    const int s1 = s0 + n;
    const bool keep = n <= kSortThreads * kSortKeep;                // block-uniform
    const float4 *src = localQ + (size_t)b * Q;
    auto source_of = [&](int i) {                                   // i-th query of the slab -> position in localQ (binary search over the runs)
        int lo = 0, hi = nblk;                                      // runStart[lo] <= i < runStart[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (runStart[mid] <= i) lo = mid; else hi = mid;
        }
        return min(runSrc[lo] + (i - runStart[lo]), Q - 1);       // (the clamp only matters for the dummy load of an empty slab)
    };
    float4 kq[kSortKeep];
    int kbin[kSortKeep], krank[kSortKeep];
    if (keep) {
#pragma unroll
        for (int k = 0; k < kSortKeep; ++k) kq[k] = src[source_of(max(min(tid + k * kSortThreads, n - 1), 0))];   // clamped, unconditional
    }
    PHASE_MARK(8);                                                  // [8] runs + loads issued + LDS cleared
    if (keep) {
#pragma unroll
        for (int k = 0; k < kSortKeep; ++k) {
            if (tid + k * kSortThreads < n) {
                kbin[k] = (cell_of(kq[k].y, oy, iy, G) - cyLo) * GxP + cell_of(kq[k].x, ox, ix, Gx);
                krank[k] = atomicAdd(&cnt[kbin[k]], 1);
            }
        }
    } else {
        for (int i = tid; i < n; i += kSortThreads) {
            const float4 q = src[source_of(i)];
            atomicAdd(&cnt[(cell_of(q.y, oy, iy, G) - cyLo) * GxP + cell_of(q.x, ox, ix, Gx)], 1);
        }
    }
    __syncthreads();
    PHASE_MARK(9);                                                  // [9] counting pass
    {   // exclusive scan over the nb words (pad words hold 0), result offset by s0; 16-byte LDS accesses
        const int nb4 = (nb + 3) >> 2;                              // the dynamic LDS block is sized to a multiple of 16 bytes
        const int per = (nb4 + kSortThreads - 1) / kSortThreads, i0 = min(nb4, tid * per), i1 = min(nb4, i0 + per);
        int4 *c4 = reinterpret_cast<int4 *>(cnt);
        int sum = 0;
        for (int i = i0; i < i1; ++i) {
            const int4 v = c4[i];
            sum += (v.x + v.y) + (v.z + v.w);
        }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        __syncthreads();                                            // wsum is reused
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int run = s0 + incl - sum;
#pragma unroll
        for (int k = 0; k < kSortThreads / 64; ++k)
            if (k < w) run += wsum[k];
        for (int i = i0; i < i1; ++i) {
            const int4 v = c4[i];
            int4 o;
            o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
            run = o.w + v.w;
            c4[i] = o;
        }
    }
    __syncthreads();
    PHASE_MARK(10);                                                 // [10] scan
    float4 *dst = sortedQ + (size_t)b * Q;
    if (keep) {                                                     // placement straight from the registers: start + rank
#pragma unroll
        for (int k = 0; k < kSortKeep; ++k)
            if (tid + k * kSortThreads < n) dst[cnt[kbin[k]] + krank[k]] = kq[k];
    }
    PHASE_MARK(11);                                                 // [11] placement
    {   // table[cz][cx][cy] for cy in [cyLo, cyHi) (the last part also writes the padding cy >= G): the threads walk
        // (cx, cy - cyLo) in order, so every cx line gets one short contiguous store burst
        const int Gp = table_pitch(G);
        int *tb = table + (size_t)b * cellStride + (size_t)cz * (Gx + 1) * Gp;
        const int cyEnd = part == kParts - 1 ? Gp : cyHi, ny = cyEnd - cyLo;
        const int total = (Gx + 1) * ny, stepx = kSortThreads / ny, stepy = kSortThreads % ny;
        int cx = tid / ny, ry = tid % ny;
        for (int i = tid; i < total; i += kSortThreads) {
            const int cy = cyLo + ry;
            int v;
            if (cy >= G) v = s1;                                                        // padding: an empty run at the slab's end
            else if (cx < Gx) v = cnt[ry * GxP + cx];
            else v = cy + 1 < cyHi ? cnt[(ry + 1) * GxP] : s1;                          // end of row (cz, cy) = start of the next row
            tb[cx * Gp + cy] = v;
            cx += stepx;
            ry += stepy;
            if (ry >= ny) { ry -= ny; ++cx; }
        }
    }
    PHASE_MARK(12);                                                 // [12] table plane
    SPAN_MARK(1);
    if (keep) return;
    __syncthreads();
    for (int i = tid; i < n; i += kSortThreads) {
        const float4 q = src[source_of(i)];
        dst[atomicAdd(&cnt[(cell_of(q.y, oy, iy, G) - cyLo) * GxP + cell_of(q.x, ox, ix, Gx)], 1)] = q;
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

#ifndef PIT_WAVES
#define PIT_WAVES 5      // 102 VGPRs: at 6 waves (80) the traversal kernel spills around its publish phase (+15 us)
#endif


// bounding box of a tet and the classification shared by the traversal kernels.  Every comparison is written so that
// NaN yields "irregular": a NaN coordinate is dropped by fmin/fmax but poisons all four dotv4 (each involves all four
// vertices), and |NaN| >= thr is false; Inf and huge values show up in the box.
struct TetBox {
    float lo[3], hi[3], w;
};
__device__ __forceinline__ bool classify(const float *v, const Planes &P, TetBox &bx)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bx.lo[k] = fminf(fminf(v[k], v[3 + k]), fminf(v[6 + k], v[9 + k]));
        bx.hi[k] = fmaxf(fmaxf(v[k], v[3 + k]), fmaxf(v[6 + k], v[9 + k]));
    }
    bx.w = fmaxf(fmaxf(bx.hi[0] - bx.lo[0], bx.hi[1] - bx.lo[1]), bx.hi[2] - bx.lo[2]);
    const float big = fmaxf(fmaxf(fmaxf(fabsf(bx.lo[0]), fabsf(bx.hi[0])), fmaxf(fabsf(bx.lo[1]), fabsf(bx.hi[1]))),
                            fmaxf(fabsf(bx.lo[2]), fabsf(bx.hi[2])));
    const float thr = kTau * ((bx.w * bx.w) * bx.w);
    const bool cond = fabsf(P.dv[0]) >= thr && fabsf(P.dv[1]) >= thr && fabsf(P.dv[2]) >= thr && fabsf(P.dv[3]) >= thr;
    // sv == 0 with a zero dotv4 is excluded by |dotv4| >= tau * w^3 > 0
    return big <= kBig && (P.sv == 0u || P.sv == 15u) && bx.w >= kWMin && cond;
}

struct CellBox {
    int cx0, cx1, cy0, cy1, cz0, cz1;
};

// The round-1 kernel (DEFTET_PIT_EXACT): one lane per tet, box test + the exact predicate on every candidate.
__global__ __launch_bounds__(256, PIT_WAVES) void k_tet_scan(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ table,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int2 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount, int hpad, int4 *spill)
{
    // per-shape words of the hit buffer: uncovered-hit counter (k_finalize appends), backward ticket, irregular-query flag.
    // (hpad is an argument: deriving it from gridDim.y here made the compiler fetch the dispatch packet with vector loads:
    // +20 us.  The branch is WAVE-UNIFORM — all lanes of the first wave store the same zeros: behind a one-thread branch the
    // compiler carried the shape index, and the pointers derived from it, in VGPRs through the whole kernel.)
    if (ucount && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) {
        ucount[blockIdx.y] = 0;
        ucount[hpad + blockIdx.y] = 0;
        ucount[2 * hpad + blockIdx.y] = 0;
    }
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
    TetBox bx;
    const bool regular = classify(v, P, bx);
    // hits[b,t] (optional): the queries this tet ACCEPTED (two in the record, four in the spill record; y == kHitOverflow
    // marks "more than fit / not recorded") — the backward filters them by condition == t, so no per-hit
    // atomics or linked lists are needed there.
    int hq[kRecSlots + kSpillSlots] = {-1, -1, -1, -1, -1, -1};
    int hcnt = 0;
    if (!regular) {
        int k = atomicAdd(&counters[b * 4 + 0], 1);
        irregT[(size_t)b * T + k] = t;
        if (hits) hits[(size_t)b * T + t] = make_int2(-1, kHitOverflow);   // accepted by k_finalize, not recorded
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
        return;
    }
    const Grid g = load_grid(gparam + b * kGridWords);
    const float m = bx.w * kMargin;
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = bx.lo[k] - m;
        ehi[k] = bx.hi[k] + m;
    }
    // no regular query can lie in the enlarged box -> nothing to do
    if (ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]) {
        if (hits) hits[(size_t)b * T + t] = make_int2(-1, -1);
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
        return;
    }
    const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
    const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
    if (wide_tet(cx0, cx1, cy0, cy1, cz0, cz1, g)) {            // k_finalize tests it against every query instead
        int k = atomicAdd(&counters[b * 4 + 0], 1);
        irregT[(size_t)b * T + k] = t;
        if (hits) hits[(size_t)b * T + t] = make_int2(-1, kHitOverflow);
        irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
        return;
    }
    const int Gp = table_pitch(G);
    const int *tb = table + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    int *res = result + (size_t)b * Q;
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) {
            const int s = tb[table_off(cz, cx0, cy, Gx, Gp)], e = tb[table_off(cz, cx1 + 1, cy, Gx, Gp)];
            for (int j = s; j < e; ++j) {
                const float4 q = sq[j];
                if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2] &&
                    accept(P, q.x, q.y, q.z)) {
                    const int qi = __float_as_int(q.w);
                    atomicMin(&res[qi], t);
#pragma unroll
                    for (int k = 0; k < kRecSlots + kSpillSlots; ++k)
                        if (hcnt == k) hq[k] = qi;
                    ++hcnt;
                }
            }
        }
    if (hits) put_hit_record(hits, spill, (size_t)b * T + t, hq, hcnt, counters, gridDim.y, b, t);
    irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
}

// ------------------------------------------------------------------------------------
// k_tet_scan_slab (DEFTET_PIT_AUTO): the per-candidate work — box test (6 compares) + exact predicate
// (4 x [3 sub, 3 mul, 2 add, 1 cmp]) — is replaced by a CERTIFIED fused filter: one 3-FMA chain per face plane, one
// min over the four, two compares.  The exact predicate runs only for candidates the filter cannot decide (a band of
// a few fp32 ulps around the face planes: ~1e-4 of the candidates on the BASELINE workload), so the result is still
// bit-exact.
//
// For a regular tet (see classify) with sigma = sign of its four dotv4, the reference accepts p iff
//     sigma * dotp_i(p) > 0  for i = 0..3        (dotp_i = fl(n_i . fl(p - a_i)), source operation order)
// ("all four false" cannot happen for regular tets).  With D_i = n_i . (p - a_i) in exact
// arithmetic over the COMPUTED normals:  |dotp_i - D_i| <= 4.0001 u * sum_k |n_ik| (|p_k| + |a_ik|),  u = 2^-24.
// The filter evaluates  A_i = fma(N_i0, x, fma(N_i1, y, fma(N_i2, z, C_i))),  N_i = sigma n_i,
// C_i = fl(-sigma c_i - E_i),  c_i = fma(n_i0, a_i0, fma(n_i1, a_i1, n_i2 a_i2)),  which equals
// sigma D_i - E_i up to  3u sum|n||p| + 7u sum|n||a| + 4u E_i.  With
//     E_i = 8 u * sum_k |n_ik| (P_k + 2 M_k) + 2^-120,    P_k >= |p_k| for every regular query (grid box),
//                                                         M_k >= |vertex coordinate k| of this tet,
// the sum of both bounds is (7.0001 u) sum|n|P + (11.0002 u) sum|n|M + 4u E_i, which E_i (1 - 4u) exceeds: 8 > 7.0001 and
// 16 > 11.0002 leave 12 % / 31 % to spare against the <= 10 u relative error of evaluating E_i itself in fp32 (three FMAs,
// one addition, S_k = fma(M_k, 16u, 8u P_k) with 8u P_k rounded once).  (Rounds 1-2 used 16 u; halving it halves the
// undecided band and with it the number of tets that need the exact re-scan.)  Hence
//     min_i A_i > 0              =>  every sigma * dotp_i > 0        =>  the reference accepts   (certain)
//     min_i A_i < -2 max_i E_i   =>  some  sigma * dotp_j < 0        =>  the reference rejects   (certain)
// and anything in between is handed to the exact predicate.  No box test is needed: a point outside the tet
// violates at least one plane.  2^-120 absorbs products that underflow in either evaluation.
//
// Traversal: the cell box of the tet is walked slab by slab (cz).  The bounds of the <= 4 y-adjacent runs of a slab
// come from two 16-byte loads of the transposed table (starts at cx0, ends at cx1 + 1); the runs are flattened into
// one candidate range [0, P4) by their prefix sums, so empty runs cost nothing and a wave-iteration always offers two
// candidates to every lane that has any left in its slab (rounds 1-2 walked (cz, cy) rows with 2 x 4-byte bounds each:
// 10.5 wave-iterations for 11.5 candidates per tet at configs[2]; slabs of half-size cells: 7 for 7).
// ------------------------------------------------------------------------------------
struct Filter {
    float N[4][3];
    float C[4];
    float twoEmax;
};

// Select with the lane mask in an SGPR pair (VOP3 v_cndmask_b32_e64).  hipcc likes to shrink selects whose mask sits in
// VCC to the VOP2 form `v_cndmask_b32_e32 …, vcc`, which gfx950 issues ~7.5x slower than an FMA (9.4 vs 1.25 ns per
// wave-instruction per SIMD, tools/probes/valu_rate_probe.hip; the SGPR-pair form: 1.85 ns).  The traversal loop carries
// a dozen selects per iteration, so the form matters more than the count.
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


// load at a 32-bit unsigned BYTE offset from a (wave-uniform) base pointer: scalar-base + vector-offset addressing
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
// 4-byte-aligned 16-byte load (the table reads start at an arbitrary cy)
struct __attribute__((packed, aligned(4))) int4u { int x, y, z, w; };
__device__ __forceinline__ int4u ld_off_u4(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const int4u *>(static_cast<const char *>(base) + byte_off);
}

// Publish helpers with the addressing fixed in the instruction: wave-uniform base in an SGPR pair, 32-bit byte offset in
// one VGPR.  (Left to itself the compiler keeps 64-bit per-lane addresses alive across the traversal loop, runs out of
// its 80 registers and spills them: four scratch reloads in front of the four atomics cost the launch 20 us.)
__device__ __forceinline__ void atomic_smin_off(int *base, unsigned byte_off, int v)
{
    // s_nop 4: the scalar base usually comes straight out of v_readfirstlane (uniform_ptr), and a vector-memory instruction
    // that reads an SGPR written by a VALU instruction needs five wait states in between.  The compiler pads that hazard
    // for its own instructions but cannot see into an asm block: without the padding the instruction used the OLD
    // contents of the register pair (round 3: stores to address 0 + offset once the code around the call no longer
    // happened to put five instructions in between).
    asm volatile("s_nop 4\n\tglobal_atomic_smin %0, %1, %2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}
// a wave-uniform pointer, re-materialised in scalar registers at the point of use
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2s __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// streaming accesses (each byte touched once per launch): the nt hint keeps them from evicting the per-query arrays the
// gathers of k_finalize / k_bary_bwd_hits live on in the 4 MiB L2 of each XCD
__device__ __forceinline__ void stream_store(float4 *p, const float4 v)
{
    const f32x4 x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<f32x4 *>(p));
}
__device__ __forceinline__ void stream_store(float *p, float v) { __builtin_nontemporal_store(v, p); }
// probe switches (tools/probes/build_variant.sh): the tet records of a launch are read once — do they belong in the L2s?
#ifndef PIT_SCAN_TET_NT
#define PIT_SCAN_TET_NT 0
#endif
#ifndef PIT_BWD_TET_NT
#define PIT_BWD_TET_NT 0
#endif
#ifndef PIT_BWD_REV
#define PIT_BWD_REV 0       // the backward walks the tets from the last to the first (what the traversal touched last comes first)
#endif
#ifndef PIT_REC_NT
#define PIT_REC_NT 1        // hit records of the wave kernel leave with the streaming hint: the backward is their only reader, and
#endif                      // k_finalize, which runs in between, finds more of the tets it gathers in the caches (28.2 -> 26.8 us)
template <bool NT>
__device__ __forceinline__ float4 load_f4(const float4 *p)
{
    if (!NT) return *p;
    const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void store_rec(int2 *p, const int2 v)
{
    if (PIT_REC_NT) {
        const i32x2s x = {v.x, v.y};
        __builtin_nontemporal_store(x, reinterpret_cast<i32x2s *>(p));
    } else {
        *p = v;
    }
}
__device__ __forceinline__ int2 stream_load(const int2 *p)
{
    const i32x2s x = __builtin_nontemporal_load(reinterpret_cast<const i32x2s *>(p));
    return make_int2(x.x, x.y);
}
__device__ __forceinline__ void store_b128_off(void *base, unsigned byte_off, int x, int y, int z, int w)
{
    const i32x4 v = {x, y, z, w};
    // s_nop: a store of more than 64 bits keeps reading its data registers for a few cycles; the compiler pads that
    // hazard for its own stores, not inside asm (without it the next VALU write clobbered the record of 4 lanes in 16)
    // (leading s_nop 4: VALU-written SGPR base read by a vector-memory instruction, see atomic_smin_off)
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}

typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_b64_off(void *base, unsigned byte_off, int x, int y)
{
    const i32x2 v = {x, y};
    asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}

// Exact re-scan of ONE tet's candidates (box test + reference predicate, as k_tet_scan does), used by k_tet_scan_slab for
// the very rare tets that met the filter's undecided band more than twice.  Out of line and called
// AFTER the traversal loop, so that nothing of it is scheduled (or kept in registers) inside the loop.  Publishes every
// accepted query with atomicMin (idempotent w.r.t. the ones the filter already accepted) and writes the hit record.
__device__ __noinline__ void exact_rescan(const float *__restrict__ tv, int t, const int *__restrict__ tb, const float4 *__restrict__ sq,
                                          int *res, int G, int Gx, int cx0, int cx1, int cy0, int cy1, int cz0, int cz1,
                                          float m, int *counters, int nB, int b, int2 *hits, int4 *spill, size_t idx)
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
    const int Gp = table_pitch(G);
    int hq[kRecSlots + kSpillSlots] = {-1, -1, -1, -1, -1, -1};
    int hcnt = 0;
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) {
            const int s = tb[table_off(cz, cx0, cy, Gx, Gp)], e = tb[table_off(cz, cx1 + 1, cy, Gx, Gp)];
            for (int j = s; j < e; ++j) {
                const float4 q = sq[j];
                if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2] &&
                    accept(P, q.x, q.y, q.z)) {
                    const int qi = __float_as_int(q.w);
                    atomicMin(&res[qi], t);
#pragma unroll
                    for (int k = 0; k < kRecSlots + kSpillSlots; ++k)
                        if (hcnt == k) hq[k] = qi;
                    ++hcnt;
                }
            }
        }
    if (hits) put_hit_record(hits, spill, idx, hq, hcnt, counters, nB, b, t);
}

// The reference predicate for ONE (tet, candidate) pair, used by k_tet_scan_slab for the rare candidates inside the
// filter's undecided band.  Out of line and re-loading the tet, so that neither the exact planes nor the vertices are
// kept in registers by the traversal loop.  (For a regular tet the predicate itself is the reference's decision: every
// point it accepts lies in the enlarged box, DESIGN.md section 3, so no box test is needed.)
__device__ __noinline__ float exact_accept(const float *__restrict__ tv, float x, float y, float z)
{
    float vv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) vv[k] = tv[k];
    Planes P;
    make_planes(vv, P);
    return accept(P, x, y, z) ? 1.0f : -1.0f;
}

// irregular queries (NaN / Inf / huge; normally none): re-load the tet so that its vertices need not stay in
// registers across the traversal loop
__device__ __noinline__ void irregular_tail_slow(const float *__restrict__ tet, int t, int b, int T, int Q, const float *__restrict__ pts,
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
__device__ __forceinline__ void irregular_tail(const float *__restrict__ tet, int t, int b, int T, int Q, const float *__restrict__ pts,
                                               const int *__restrict__ counters, const int *__restrict__ irregQ, int *result)
{
    if (counters[b * 4 + 1] > 0) irregular_tail_slow(tet, t, b, T, Q, pts, counters, irregQ, result);
}

#ifndef PIT_BATCH
#define PIT_BATCH 3
#endif
#if PIT_BATCH < 1 || PIT_BATCH > 4
#error "PIT_BATCH must be 1..4 (a wave-iteration may add at most four acceptances to the four-deep hit register)"
#endif
// ORD: the lane at position p of the launch works on tet order[p] (a permutation of [0, T) shared by the shapes of the batch,
// deftet_tet_spatial_order_f32) and publishes / records under that ORIGINAL index, so every output is the same as without
// the permutation; a template argument, so that the unordered instance is the round-4 kernel unchanged.
template <bool ORD>
__global__ __launch_bounds__(256, PIT_WAVES) void k_tet_scan_slab(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ table,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int2 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount, int hpad, int4 *spill,
                                                  const int *__restrict__ order)
{
    // per-shape words of the hit buffer: uncovered-hit counter (k_finalize appends), backward ticket, irregular-query flag.
    // (hpad is an argument: deriving it from gridDim.y here made the compiler fetch the dispatch packet with vector loads:
    // +20 us.  The branch is WAVE-UNIFORM — all lanes of the first wave store the same zeros: behind a one-thread branch the
    // compiler carried the shape index, and the pointers derived from it, in VGPRs through the whole kernel.)
    if (ucount && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) {
        ucount[blockIdx.y] = 0;
        ucount[hpad + blockIdx.y] = 0;
        ucount[2 * hpad + blockIdx.y] = 0;
    }
    const int b = blockIdx.y;
    const int nblk = gridDim.x;
    const int per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware mapping, see k_tet_scan
    const int pos = vb * blockDim.x + threadIdx.x;
    if (vb >= nblk || pos >= T) return;
    const int t = ORD ? order[pos] : pos;
    PHASE_DECL;
    float v[12];
    {
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + t) * 12);
        float4 a = src[0], bq = src[1], c = src[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
    Filter F;
    TetBox bx;
    const Grid g = load_grid(gparam + b * kGridWords);
    {
        Planes P;
        make_planes(v, P);
        if (!classify(v, P, bx)) {
            int k = atomicAdd(&counters[b * 4 + 0], 1);
            irregT[(size_t)b * T + k] = t;
            if (hits) hits[(size_t)b * T + t] = make_int2(-1, kHitOverflow);
            irregular_queries_tail(P, t, b, Q, pts, counters, irregQ, result);
            return;
        }
        const float sigma = P.sv == 15u ? 1.0f : -1.0f;
        float S[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) S[k] = fmaf(fmaxf(fabsf(bx.lo[k]), fabsf(bx.hi[k])), 2.0f * kErrScale, g.pe[k]);
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
    typedef float pk2 __attribute__((ext_vector_type(2)));
    const pk2 Px01 = {F.N[0][0], F.N[1][0]}, Py01 = {F.N[0][1], F.N[1][1]}, Pz01 = {F.N[0][2], F.N[1][2]}, Pc01 = {F.C[0], F.C[1]};
    const pk2 Px23 = {F.N[2][0], F.N[3][0]}, Py23 = {F.N[2][1], F.N[3][1]}, Pz23 = {F.N[2][2], F.N[3][2]}, Pc23 = {F.C[2], F.C[3]};
    const float m = bx.w * kMargin;
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = bx.lo[k] - m;
        ehi[k] = bx.hi[k] + m;
    }
    if (ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]) {
        if (hits) hits[(size_t)b * T + t] = make_int2(-1, -1);
        irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
        return;
    }
    const int cx0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), cx1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
    const int cy0 = cell_of(elo[1], g.o[1], g.inv[1], G), cy1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
    const int cz0 = cell_of(elo[2], g.o[2], g.inv[2], G), cz1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
    if (wide_tet(cx0, cx1, cy0, cy1, cz0, cz1, g)) {            // k_finalize tests it against every query instead
        int k = atomicAdd(&counters[b * 4 + 0], 1);
        irregT[(size_t)b * T + k] = t;
        if (hits) hits[(size_t)b * T + t] = make_int2(-1, kHitOverflow);
        irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
        return;
    }
    const int Gp = table_pitch(G);
    const int *tb = table + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    PHASE_MARK(0);                                                       // [0] load + setup
    // Accepted queries go into a four-deep shift register (h0 = newest) and are published with atomicMin once, after the
    // traversal.  Two rare events are handled by wave-uniform branches inside the loop, so that no tet is ever walked twice
    // (rounds 1-2 re-scanned such tets after the loop — a serial chain of ~20 memory round trips that kept the whole wave
    // alive; 1.3e-3 of the tets, 4 us of the launch):
    //  * a candidate in the filter's undecided band is remembered (two slots) and gets the reference predicate after the
    //    loop (exact_accept; a third one — never seen on the BASELINE workloads — falls back to the exact re-scan);
    //  * a lane about to hold more than four acceptances publishes what it holds; the first time it also writes the four
    //    into its record (flagged kHitSpilled) and goes on collecting into the spill record, the second time it marks the
    //    tet "overflowed" (its hits are then carried by the uncovered list).
    int h0 = -1, h1 = -1, h2 = -1, h3 = -1, hcnt = 0;
    int full = 0;                                                      // 0: nothing yet, 1: first batch spilled, 2: overflowed
    int cap = 4, scnt = 0;                                             // acceptances the register may hold now; spill slots already used
    auto on_full = [&]() {                                             // the lane holds PUBLISHED acceptances and gets more than fit
        // (scalar base + 32-bit offset, like the record store at the end: a 64-bit per-lane address kept alive across the
        // loop for this rare store was spilt to scratch)
        // first time: the first two held go to the record (flagged), the others (if any) to the first slots of the spill record,
        // and the register goes on collecting what the spill record still has room for (six recorded in all, whatever the
        // batches the acceptances arrive in); second time: the tet is marked overflowed at the end
        if (hits && spill && full == 0) {
            store_b64_off(uniform_ptr(hits + (size_t)b * T), (unsigned)t * 8u, h0 | kHitSpilled, h1);
            store_b128_off(uniform_ptr(spill + (size_t)b * T), (unsigned)t * 16u, h2, h3, -1, -1);
            scnt = max(hcnt - kRecSlots, 0);
            cap = kSpillSlots - scnt;
        }
        full = (hits && spill) ? min(full + 1, 2) : 2;
        h0 = -1; h1 = -1; h2 = -1; h3 = -1;
        hcnt = 0;
    };
    int pend0 = 0, pend1 = 0, npend = 0;                               // positions in sortedQ of undecided candidates
    // Slab cursor.  A "slab step" is (cz, chunk of four cy rows); tets that span more than four y cells (rare: needles)
    // take several chunks per cz.  All table addresses are 32-bit BYTE offsets from the wave-uniform table base.
    const int ny = cy1 - cy0 + 1;
    const unsigned lineB = (unsigned)Gp * 4u;                                             // one (cz, cx) line
    const unsigned slabB = (unsigned)(Gx + 1) * lineB;                                    // one cz plane
    const unsigned dxB = (unsigned)(cx1 + 1 - cx0) * lineB;                               // start line -> end line
    unsigned offS = ((unsigned)cz0 * (unsigned)(Gx + 1) + (unsigned)cx0) * lineB + (unsigned)cy0 * 4u;   // bounds of the step held in (nS, nE)
    int cz = cz0, yoff = 0;                                                              // position of that step
    int4u nS = ld_off_u4(tb, offS), nE = ld_off_u4(tb, offS + dxB);
    bool haveNext = true;
    int c = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    while (c < P4 || haveNext) {
        if (c >= P4) {                                                  // enter the prefetched step, prefetch the one after it
            const int rem = ny - yoff;                                  // rows of this chunk: min(rem, 4)
            const int n0 = nE.x - nS.x;
            const int n1 = sel(mask_of(rem > 1), nE.y - nS.y, 0);
            const int n2 = sel(mask_of(rem > 2), nE.z - nS.z, 0);
            const int n3 = sel(mask_of(rem > 3), nE.w - nS.w, 0);
            P1 = n0; P2 = P1 + n1; P3 = P2 + n2; P4 = P3 + n3;
            o0 = nS.x; o1 = nS.y - P1; o2 = nS.z - P2; o3 = nS.w - P3;
            c = 0;
            // pin the eight values here: every use of (nS, nE) then precedes the prefetch below, which can load into the
            // same registers (without this the compiler sinks the arithmetic below the loads, copies the bounds and waits
            // for the prefetch right after issuing it)
            asm volatile("" : "+v"(P1), "+v"(P2), "+v"(P3), "+v"(P4), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            const lanemask_t wrap = mask_of(rem <= 4);                  // last chunk of this cz
            offS += sel(wrap, slabB - (unsigned)yoff * 4u, 16u);
            yoff = sel(wrap, 0, yoff + 4);
            cz += sel(wrap, 1, 0);
            haveNext = cz <= cz1;
            if (haveNext) {
                nS = ld_off_u4(tb, offS);
                nE = ld_off_u4(tb, offS + dxB);
            }
        }
        if (c < P4) {
            // PIT_BATCH candidates per lane and wave-iteration; a dead slot repeats the first candidate (never recorded)
            int cc[PIT_BATCH], jq[PIT_BATCH];
            bool live[PIT_BATCH];
            float4 q[PIT_BATCH];
            cc[0] = c; live[0] = true;
#pragma unroll
            for (int k = 1; k < PIT_BATCH; ++k) {
                live[k] = c + k < P4;
                cc[k] = sel(mask_of(live[k]), c + k, c);
            }
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k) {
                jq[k] = cc[k] + sel(mask_of(cc[k] < P1), o0, sel(mask_of(cc[k] < P2), o1, sel(mask_of(cc[k] < P3), o2, o3)));
                q[k] = ld_off<float4>(sq, (unsigned)jq[k] * 16u);
            }
            __builtin_amdgcn_sched_barrier(0);                         // all gathers in flight before the first filter value is needed
            float av[PIT_BATCH];
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k) {                       // two planes per v_pk_fma_f32 (same fmas, same order)
                const pk2 qx = {q[k].x, q[k].x}, qy = {q[k].y, q[k].y}, qz = {q[k].z, q[k].z};
                const pk2 A01 = __builtin_elementwise_fma(Px01, qx, __builtin_elementwise_fma(Py01, qy, __builtin_elementwise_fma(Pz01, qz, Pc01)));
                const pk2 A23 = __builtin_elementwise_fma(Px23, qx, __builtin_elementwise_fma(Py23, qy, __builtin_elementwise_fma(Pz23, qz, Pc23)));
                av[k] = fminf(fminf(A01.x, A01.y), fminf(A23.x, A23.y));
            }
            float am = fabsf(av[0]);                                   // dead slots repeat candidate 0: no mask needed
#pragma unroll
            for (int k = 1; k < PIT_BATCH; ++k) am = fminf(am, fabsf(av[k]));
            if (__builtin_amdgcn_ballot_w64(am <= F.twoEmax) != 0ull) {       // rare: someone met the undecided band
#pragma unroll
                for (int k = 0; k < PIT_BATCH; ++k) {                          // remember the candidate; decided after the loop
                    const lanemask_t bd = mask_of(live[k] && fabsf(av[k]) <= F.twoEmax);
                    pend1 = sel(bd, pend0, pend1);
                    pend0 = sel(bd, jq[k], pend0);
                    npend += sel(bd, 1, 0);
                    av[k] = __int_as_float(sel(bd, __float_as_int(-1.0f), __float_as_int(av[k])));
                }
            }
            lanemask_t acc[PIT_BATCH];
            int cnew = hcnt;
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k) {
                acc[k] = mask_of(live[k] && av[k] > 0.f);
                cnew += sel(acc[k], 1, 0);
            }
            if (__builtin_amdgcn_ballot_w64(cnew > cap) != 0ull) {            // rare: more acceptances than the register holds
                if (cnew > cap) {
                    int *resb = result + (size_t)b * Q;
                    if (hcnt > 0) atomicMin(&resb[h0], t);
                    if (hcnt > 1) atomicMin(&resb[h1], t);
                    if (hcnt > 2) atomicMin(&resb[h2], t);
                    if (hcnt > 3) atomicMin(&resb[h3], t);
                    on_full();
                }
            }
#pragma unroll
            for (int k = 0; k < PIT_BATCH; ++k) {
                const int qi = __float_as_int(q[k].w);
                h3 = sel(acc[k], h2, h3);
                h2 = sel(acc[k], h1, h2);
                h1 = sel(acc[k], h0, h1);
                h0 = sel(acc[k], qi, h0);
                hcnt += sel(acc[k], 1, 0);
            }
            c += PIT_BATCH;
        }
    }
    PHASE_MARK(1);                                                       // [1] traversal loop
    if (npend > 0) {                                                      // undecided candidates (rare)
        const float *tv = tet + ((size_t)b * T + t) * 12;
        atomicAdd(&counters[gridDim.y * 4 + b * 4 + 1], npend);              // statistics: candidates decided exactly
        if (npend > 2) {
            exact_rescan(tv, t, tb, sq, result + (size_t)b * Q, G, Gx, cx0, cx1, cy0, cy1, cz0, cz1, m, counters, gridDim.y, b, hits, spill, (size_t)b * T + t);
            irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
            return;
        }
        for (int k = 0; k < npend; ++k) {
            const float4 q = sq[k == 0 ? pend0 : pend1];
            if (exact_accept(tv, q.x, q.y, q.z) > 0.f) {
                const int qi = __float_as_int(q.w);
                atomicMin(&result[(size_t)b * Q + qi], t);
                if (hcnt >= cap) {                                           // (the held ones are not published yet)
                    int *resb = result + (size_t)b * Q;
                    atomicMin(&resb[h0], t); atomicMin(&resb[h1], t);
                    if (hcnt > 2) atomicMin(&resb[h2], t);
                    if (hcnt > 3) atomicMin(&resb[h3], t);
                    on_full();
                }
                h3 = h2; h2 = h1; h1 = h0; h0 = qi;                          // (qi itself was published above)
                ++hcnt;
            }
        }
    }
    int *resb = uniform_ptr(result + (size_t)b * Q);
    if (hcnt > 0) atomic_smin_off(resb, (unsigned)h0 * 4u, t);
    if (hcnt > 1) atomic_smin_off(resb, (unsigned)h1 * 4u, t);
    if (hcnt > 2) atomic_smin_off(resb, (unsigned)h2 * 4u, t);
    if (hcnt > 3) atomic_smin_off(resb, (unsigned)h3 * 4u, t);
    if (full == 1 && hcnt > cap) full = 2;                             // (the last wave-iteration brought more than the spill record still holds)
    if (full == 2) note_overflow(counters, gridDim.y, b, t);
    if (hits) {
        // full == 0: the record (+ the low half of the spill record when three or four are held); 1: the high half of the spill
        // record (record and low half were written when the register spilled); 2: overflow marker
        if (full == 0) {
            const lanemask_t sp = mask_of(hcnt > 2);
            store_b64_off(uniform_ptr(hits + (size_t)b * T), (unsigned)t * 8u, sel(sp, h0 | kHitSpilled, h0), h1);
            if (hcnt > 2) store_b128_off(uniform_ptr(spill + (size_t)b * T), (unsigned)t * 16u, h2, h3, -1, -1);
        } else if (full == 1) {                                          // (rare: one dword per held acceptance, behind the slots in use)
            int *sp = reinterpret_cast<int *>(spill + (size_t)b * T + t) + scnt;
            if (hcnt > 0) sp[0] = h0;
            if (hcnt > 1) sp[1] = h1;
            if (hcnt > 2) sp[2] = h2;
            if (hcnt > 3) sp[3] = h3;
        } else {
            store_b64_off(uniform_ptr(hits + (size_t)b * T), (unsigned)t * 8u, -1, kHitOverflow);
        }
    }
    irregular_tail(tet, t, b, T, Q, pts, counters, irregQ, result);
    PHASE_MARK(2);                                                       // [2] publish (atomics, record store) / re-scan
}

// ------------------------------------------------------------------------------------
// k_tet_scan_wave (DEFTET_PIT_AUTO since round 4).  Same result and hit records as k_tet_scan_slab; three changes:
//
//  1. FILTER-ONLY SETUP.  The lane no longer evaluates the reference's four cross products: it builds the filter from
//     the three faces at v0 (m0 = e1 x e2, m1 = e3 x e1, m2 = e2 x e3, e_i = v_i - v0), the fourth as
//     m3 = -(m0 + m1 + m2) (exact identity for exact normals) and ONE determinant det = m0 . e3.  With N*_i the exact
//     normal of the reference's ordering i and D*_i(p) = N*_i . (p - a_i) (exact arithmetic on the fp32 vertices), w_k the
//     box extents of the tet and G_k = 2 w_l w_m (>= sum of the absolute products of any face normal's component k):
//       reference:  |dotp_i - D*_i(p)| <= sum_k |p_k - a_k| (4.03 u |N*_k| + 4.02 u G_k) <= 8.05 u sum_k G_k |p_k - a_k|
//                   (its normal: two rounded products, one difference, rounded edges: 4.01 u G_k per component; the
//                   dot product: 3.01 u; the rounded p - a: u)
//       here:       |m_ik - N*_ik| <= 4.01 u G_k for i < 3 (fma form: not worse), <= 15.05 u G_k for the derived one
//                   (three such errors + two rounded additions of terms <= 2 G_k), base point v0 (faces 0-2) / v1 (face 3)
//                   lies on the exact plane, so sigma m_i . (p - base) differs from sigma D*_i(p) by <= 15.05 u sum_k G_k |p_k - base_k|;
//                   evaluating it as A_i = fma(N_i0, x, fma(N_i1, y, fma(N_i2, z, C_i))), C_i = fma(-sigma, c_i, -E_i):
//                   3.01 u sum|m||p| + 7.02 u sum|m||base| + 4.01 u E_i    (as for k_tet_scan_slab).
//     Hence  E_i = 24 u sum_k G_k R_k  +  8 u sum_k |m_ik| (P_k + 2 M_k)  + 2^-120   with R_k >= |p_k - vertex_k| for every
//     candidate this lane is ever offered covers all of it (24 > 23.1, 8 > 7.02; >= 3.7 % to spare against the ~10 u relative
//     error of evaluating E_i itself), and   min_i A_i > 0 => the reference accepts,  min_i A_i < -2 max_i E_i => it rejects;
//     in between the reference predicate decides (exact_accept), so the result stays bit-exact.
//     R_k: the candidates come from grid cells c with c in [first, last] along axis k, a range that contains the cells
//     of the tet's enlarged box; a vertex and a candidate are then both within (box edge + n cells) of a common point:
//     R_k = (ehi_k - elo_k) + (n_k + 1) cs_k, n_k = number of cells of that range (the lane's own, or its group's
//     footprint, see 2), cs_k >= cell edge (Grid::cs).
//     "Regular" needs the reference's four dotv4 to share a strict sign and |dotv4| >= 2^-7 w^3: |dotv4_i - det| <= 97 u w^3
//     (same error terms), so |det| >= 2^-7 (1 + 2^-7) w^3 implies it (2^-14 = 1024 u).
//  2. THE CANDIDATES OF A WAVE ARE STAGED IN LDS.  64 consecutive tets of a spatially coherent mesh share their
//     neighbourhood.  The wave picks a pivot lane, groups the lanes whose (x, y) cell footprint is within two / one cells
//     of the pivot's (up to two groups: a wave that straddles two columns of the mesh), reduces the group's cell box
//     (packed 16-bit max over DPP), reads the bounds of the box's (cz, cy) rows over the group's x range — one LANE per
//     slab, two 16-byte loads of the transposed table (start column, end column) for the slab's <= 4 rows; a wave scan
//     gives the rows' positions in LDS — and copies the rows' queries with coalesced loads, balanced over the lanes
//     through an owner scan.  A lane's candidates are then ONE contiguous LDS range — all staged rows of its own z slabs —
//     walked with ds_read_b128, two candidates per trip, the four planes as two packed pairs (v_pk_fma_f32: six
//     instructions per candidate).  Accepted queries are kept in the lane's LDS slots (kWvSlots) and published after the
//     loops with one atomicMin instruction per slot level (an atomic per acceptance inside the loop was a vector-memory
//     instruction with one or two live lanes: 101 vs 84 us).  Footprints with more queries than a chunk holds are staged
//     slab range by slab range.  Everything here is per wave: no barrier.
//  3. Lanes that fit no group (incoherent tet order, over-long footprints) walk the global table as k_tet_scan_slab does.
// ------------------------------------------------------------------------------------
constexpr int kWvRows = 192;                            // (cz, cy) rows of a staged footprint (three per lane)
#ifndef PIT_WVCAP
#define PIT_WVCAP 96
#endif
constexpr int kWvCap = PIT_WVCAP;                       // staged queries per chunk
constexpr int kWvSlots = 6;                             // accepted queries a lane keeps: record + half a spill record
#ifndef PIT_WG
#define PIT_WG 128     // 256: 64.4-64.8 / 144.5-145.8 us at configs[2] / [3]; 128: 63.3 / 142.8; 64: 64.2 / 143.1 (a workgroup frees its slots when its slowest wave ends)
#endif
constexpr int kWvThreads = PIT_WG;                      // threads per workgroup of the wave-staged kernels (nothing in them needs more than a wave)
static_assert(kWvThreads % 64 == 0 && kWvThreads >= 64 && kWvThreads <= 256, "whole waves");
static_assert(kWvRows % 64 == 0 && kWvSlots >= 4 && kWvSlots <= 8, "whole waves of row ids; the records hold four + four");
// two tets per lane (k_tet_scan_pair): the wave's footprint holds the rows and queries of 128 tets
#ifndef PIT_WVCAP_PAIR
#define PIT_WVCAP_PAIR 128
#endif
constexpr int kWvCapPair = PIT_WVCAP_PAIR, kWvRowsPair = 256;
// timing experiments only (wrong results when 0): leave out the global walk of the lanes no group covers / the exact decisions
#ifndef PIT_DBG_WALK
#define PIT_DBG_WALK 1
#endif
#ifndef PIT_DBG_PEND
#define PIT_DBG_PEND 1
#endif
// probe builds only (-DPIT_STOP=n): the traversal ends after stage n (1 setup, 2 groups + radius, 3 row bounds + scan, 4 staging
// without the candidate loop but with the tail, 5 everything but the publish / record tail) with what it computed kept alive, so that instruction counters can be read per stage
#ifndef PIT_STOP
#define PIT_STOP 0
#endif
#define PIT_KEEP(x) asm volatile("" ::"v"(x))
#ifndef PIT_PROBE_SKIP
#define PIT_PROBE_SKIP 0      // probe builds only (wrong results): 1 = no publish atomics, 2 = no hit-record store in k_tet_scan_wave
#endif
#ifndef PIT_PROBE_PREFILL
#define PIT_PROBE_PREFILL 0
#endif
#ifndef PIT_PIN_SHAPES
#define PIT_PIN_SHAPES 0
#endif
#ifndef PIT_PROBE_SCATTER
#define PIT_PROBE_SCATTER 0   // probe builds only (1 / 2): per-hit scattered stores beside the publish atomics (timing only)
#endif
#ifndef PIT_WAVES_PAIR
#define PIT_WAVES_PAIR 5
#endif
#ifndef PIT_TOLX
#define PIT_TOLX 2          // x cells a lane's footprint may differ from the pivot's
#endif
#ifndef PIT_MINGROUP
#define PIT_MINGROUP 4      // 16: 68.7 us at configs[2], 150 at configs[3]; 8: 66.2 / 148; 4: 65.2 / 143; 2, 1: the same
#endif
constexpr int kWvMinGroup = PIT_MINGROUP;               // lanes a footprint group must have to be worth staging
#ifndef PIT_ZREACH
#define PIT_ZREACH 24
#endif
constexpr int kWvZReach = PIT_ZREACH;                    // slabs a lane's first slab may lie from its pivot's (two groups of 2 * 24 + a few slabs fit 64 lanes... mostly)
constexpr float kRelScale = 2.86102294921875e-06f;      // 48 u = 24 u * (the 2 of G_k = 2 w_l w_m)
constexpr float kTauSlim = kTau * (1.0f + 1.0f / 128.0f);
#ifndef PIT_WAVES2
#define PIT_WAVES2 7     // 72 registers, scratch only on the rare exact-decision path; 69.0 us (six waves, 76 registers: 72.2; eight, 64: 70.5)
#endif
template <int CAP, int ROWS>
struct __attribute__((aligned(16))) WaveStageT {       // one tet per lane: 2.8 KB per wave, with the hit slots 19.4 KB per workgroup
    typedef typename std::conditional<(ROWS <= 254), unsigned char, unsigned short>::type marker_t;
    float4 q[CAP];                                      // the chunk's queries, rows in (cz, cy) order
    int delta[ROWS];                                    // position of the row's first query in sortedQ - rowBase
    unsigned short rowBase[ROWS + 4];                   // exclusive prefix of the row lengths; [R] = total (< 2^16, checked)
    marker_t marker[CAP];                               // row id + 1 at the LDS position where a non-empty row starts
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp0(int v)             // lanes without a source (or masked rows) read 0
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
// inclusive scans over the 64 lanes (row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 / 31)
__device__ __forceinline__ int wave_scan_add(int x)
{
    x += dpp0<0x111, 0xf>(x); x += dpp0<0x112, 0xf>(x); x += dpp0<0x114, 0xf>(x); x += dpp0<0x118, 0xf>(x);
    x += dpp0<0x142, 0xa>(x); x += dpp0<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ int wave_scan_max(int x)    // x >= 0
{
    x = max(x, dpp0<0x111, 0xf>(x)); x = max(x, dpp0<0x112, 0xf>(x)); x = max(x, dpp0<0x114, 0xf>(x)); x = max(x, dpp0<0x118, 0xf>(x));
    x = max(x, dpp0<0x142, 0xa>(x)); x = max(x, dpp0<0x143, 0xc>(x));
    return x;
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(unsigned, r);
}
// max of both 16-bit halves over all lanes (max is idempotent: the row masks of a scan are not needed); wave-uniform
__device__ __forceinline__ unsigned wave_pk_max(unsigned v)
{
    v = pk_max(v, (unsigned)dpp0<0x111, 0xf>((int)v)); v = pk_max(v, (unsigned)dpp0<0x112, 0xf>((int)v));
    v = pk_max(v, (unsigned)dpp0<0x114, 0xf>((int)v)); v = pk_max(v, (unsigned)dpp0<0x118, 0xf>((int)v));
    v = pk_max(v, (unsigned)dpp0<0x142, 0xf>((int)v)); v = pk_max(v, (unsigned)dpp0<0x143, 0xf>((int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// wave-private LDS hand-off between lanes: DS operations of a wave execute in order, the fence keeps the compiler from
// moving them
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void atomic_smin_off_nh(int *base, unsigned byte_off, int v)   // base: SGPR pair written long before (no hazard)
{
    asm volatile("global_atomic_smin %0, %1, %2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}

__device__ __forceinline__ float sel(lanemask_t m, float if_set, float if_clear)
{
    return __int_as_float(sel(m, __float_as_int(if_set), __float_as_int(if_clear)));
}
__device__ __forceinline__ void cross_fma(float ax, float ay, float az, float bx, float by, float bz, float *n)
{
    n[0] = fmaf(ay, bz, -(az * by));
    n[1] = fmaf(az, bx, -(ax * bz));
    n[2] = fmaf(ax, by, -(ay * bx));
}

// irregular tet (flat / needle / non-finite / huge): listed for k_finalize, its hits are not recorded
__device__ __noinline__ void irregular_tet_slow(const float *__restrict__ tet, int t, int b, int T, int Q, const float *__restrict__ pts,
                                                int *counters, int *irregT, const int *__restrict__ irregQ, int *result, int2 *hits)
{
    const int k = atomicAdd(&counters[b * 4 + 0], 1);
    irregT[(size_t)b * T + k] = t;
    if (hits) hits[(size_t)b * T + t] = make_int2(-1, kHitOverflow);
    if (counters[b * 4 + 1] > 0) irregular_tail_slow(tet, t, b, T, Q, pts, counters, irregQ, result);
}

// The cell box of a regular tet, from its vertices: the same operations, in the same order, as the setup of
// k_tet_scan_wave (so the same cells).  The kernel does not keep its cells across the staged loop — the few lanes that need
// them afterwards (global walk, exact re-scan) recompute them from the reloaded tet: six registers less in the loop.
__device__ __forceinline__ void tet_cell_box(const float *__restrict__ tv, const Grid &g, int G, int Gx, CellBox &cb, float &mrg)
{
    float blo[3], bhi[3], wk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        blo[k] = fminf(fminf(tv[k], tv[3 + k]), fminf(tv[6 + k], tv[9 + k]));
        bhi[k] = fmaxf(fmaxf(tv[k], tv[3 + k]), fmaxf(tv[6 + k], tv[9 + k]));
        wk[k] = bhi[k] - blo[k];
    }
    const float w = fmaxf(fmaxf(wk[0], wk[1]), wk[2]);
    mrg = w * kMargin;
    cb.cx0 = cell_of(blo[0] - mrg, g.o[0], g.inv[0], Gx); cb.cx1 = cell_of(bhi[0] + mrg, g.o[0], g.inv[0], Gx);
    cb.cy0 = cell_of(blo[1] - mrg, g.o[1], g.inv[1], G); cb.cy1 = cell_of(bhi[1] + mrg, g.o[1], g.inv[1], G);
    cb.cz0 = cell_of(blo[2] - mrg, g.o[2], g.inv[2], G); cb.cz1 = cell_of(bhi[2] + mrg, g.o[2], g.inv[2], G);
}

// exact re-scan of one tet (more than two undecided candidates, or more acceptances than slots: practically never): every
// accepted query is published and the lane's LDS slots are refilled from scratch; returns the number of accepted queries
__device__ __noinline__ int exact_rescan_slots(const float *__restrict__ tv, int t, const int *__restrict__ tb, const float4 *__restrict__ sq,
                                               int *res, const float *__restrict__ gp, int G, int Gx, int *slotCol)
{
    float vv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) vv[k] = tv[k];
    Planes P;
    make_planes(vv, P);
    const Grid g = load_grid(gp);
    CellBox cb;
    float m;
    tet_cell_box(vv, g, G, Gx, cb, m);
    float elo[3], ehi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        elo[k] = fminf(fminf(vv[k], vv[3 + k]), fminf(vv[6 + k], vv[9 + k])) - m;
        ehi[k] = fmaxf(fmaxf(vv[k], vv[3 + k]), fmaxf(vv[6 + k], vv[9 + k])) + m;
    }
    const int Gp = table_pitch(G);
    int hcnt = 0;
    for (int cz = cb.cz0; cz <= cb.cz1; ++cz)
        for (int cy = cb.cy0; cy <= cb.cy1; ++cy) {
            const int s = tb[table_off(cz, cb.cx0, cy, Gx, Gp)], e = tb[table_off(cz, cb.cx1 + 1, cy, Gx, Gp)];
            for (int j = s; j < e; ++j) {
                const float4 q = sq[j];
                if (q.x >= elo[0] && q.x <= ehi[0] && q.y >= elo[1] && q.y <= ehi[1] && q.z >= elo[2] && q.z <= ehi[2] &&
                    accept(P, q.x, q.y, q.z)) {
                    const int qi = __float_as_int(q.w);
                    atomicMin(&res[qi], t);
                    slotCol[min(hcnt, kWvSlots) * kWvThreads] = qi;
                    ++hcnt;
                }
            }
        }
    return hcnt;
}

// barycentric weights, utils/tet_utils.py:25-45 (same association as the torch expression)
__device__ __forceinline__ float triple(const float *a, const float *b, const float *c)
{
    float x0 = b[1] * c[2] - b[2] * c[1];
    float x1 = b[2] * c[0] - b[0] * c[2];
    float x2 = b[0] * c[1] - b[1] * c[0];
    return (a[0] * x0 + a[1] * x1) + a[2] * x2;
}
// the four weights of point p in the tet (t0, t1, t2 = its 48-byte record): utils/tet_utils.py:28-45, operation for operation
__device__ __forceinline__ float4 bary_weights(const float4 t0, const float4 t1, const float4 t2, float px, float py, float pz)
{
    const float A[3] = {t0.x, t0.y, t0.z}, Bv[3] = {t0.w, t1.x, t1.y}, Cv[3] = {t1.z, t1.w, t2.x}, D[3] = {t2.y, t2.z, t2.w};
    const float pp[3] = {px, py, pz};
    float vap[3], vbp[3], vab[3], vac[3], vad[3], vbc[3], vbd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vap[k] = pp[k] - A[k]; vbp[k] = pp[k] - Bv[k];
        vab[k] = Bv[k] - A[k]; vac[k] = Cv[k] - A[k]; vad[k] = D[k] - A[k];
        vbc[k] = Cv[k] - Bv[k]; vbd[k] = D[k] - Bv[k];
    }
    const float v6 = 1.0f / triple(vab, vac, vad);
    float4 wq;
    wq.x = triple(vbp, vbd, vbc) * v6;
    wq.y = triple(vap, vac, vad) * v6;
    wq.z = triple(vap, vad, vab) * v6;
    wq.w = triple(vap, vab, vac) * v6;
    return wq;
}

// NT tets per lane (1: k_tet_scan_wave, 2: k_tet_scan_pair).  With two, the lane owns the tets at positions 2p and 2p + 1 — in a
// coherent list neighbours, usually with the same candidates — and everything that is done once per WAVE (footprint groups,
// row bounds, scan, staging: 59 % of the instructions of the one-tet kernel at configs[2]) serves 128 tets instead of 64:
// the lane's footprint is the union of its two cell boxes, every staged candidate is tested against both filters (one LDS
// read, 2 x 6 packed FMAs), each tet has its own slots, record and publish.  A tet of the pair that has nothing to traverse
// (irregular, outside the query box, past the end of the list) gets a filter that rejects everything.
template <bool ORD, int NT>                                             // ORD: see k_tet_scan_slab
__device__ __forceinline__ void tet_scan_wave_body(const float *__restrict__ tet, int T, int Q,
                                                  const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ table,
                                                  long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters,
                                                  int *irregT, int2 *hits, const float *__restrict__ pts,
                                                  const int *__restrict__ irregQ, int *ucount, int hpad, int4 *spill,
                                                  const int *__restrict__ order)
{
    constexpr int CAP = NT == 1 ? kWvCap : kWvCapPair, ROWS = NT == 1 ? kWvRows : kWvRowsPair;
    typedef WaveStageT<CAP, ROWS> Stage;
    __shared__ Stage s_w[kWvThreads / 64];
    __shared__ int s_hit[NT][kWvSlots + 2][kWvThreads];                // [tet of the lane][slot][thread]; the last two rows swallow the overflow
#if PIT_PIN_SHAPES
    // probe builds: shape-per-XCD placement (shape_block): workgroup L runs on XCD L % 8, so with shape = L % B every XCD walks ONE
    // shape (B = 8) and the per-query arrays it scatters into live in one L2
    const int pinL = blockIdx.y * gridDim.x + blockIdx.x, pinB = gridDim.y;
#define PIT_BLK_B (pinL % pinB)
#define PIT_BLK_X (pinL / pinB)
#else
#define PIT_BLK_B blockIdx.y
#define PIT_BLK_X blockIdx.x
#endif
    if (ucount && PIT_BLK_X == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) {   // per-shape words of the hit buffer
        ucount[PIT_BLK_B] = 0;                                         // (wave-uniform branch, see k_tet_scan_slab: behind a
        ucount[hpad + PIT_BLK_B] = 0;                                  // one-thread branch: 80 registers and an 8-byte scratch store
        ucount[2 * hpad + PIT_BLK_B] = 0;                              // per lane = 16 MB of HBM writes per launch; so: 76, none)
    }
    const int b = PIT_BLK_B, tid = threadIdx.x, lane = tid & 63;
    Stage &W = s_w[tid >> 6];
    const int nblk = gridDim.x;
#if PIT_PIN_SHAPES
    const int vb = PIT_BLK_X;
#else
    const int per = (nblk + 7) >> 3;
    const int vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // XCD-aware mapping, see k_tet_scan
#endif
#undef PIT_BLK_B
#undef PIT_BLK_X
    const int t = vb * blockDim.x + tid;                                // the lane's position in the launch: tets at NT * t + k
    const bool valid = vb < nblk && t < (T + NT - 1) / NT;
    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;           // (every other lane stays: the wave works together)
    // (re-read where it is needed instead of kept in a register across the staged loop; clamped for the lanes past the end)
    auto tet_id = [&](int k) -> int {
        const int p = min(NT * t + k, T - 1);
        return ORD ? order[p] : p;
    };
    auto tet_exists = [&](int k) -> bool { return valid && (NT == 1 || NT * t + k < T); };
    PHASE_DECL;
    SPAN_MARK(0);
    const Grid g = load_grid(gparam + b * kGridWords);
    // --- filter-only setup -------------------------------------------------------------------------------------------
    // Everything that does not depend on the lane's group is finished here, so that the vertices and normals are dead
    // before the wave-level part starts: N_i, cE_i = fl(-sigma c_i - Eabs_i), max Eabs; the group-dependent part of the
    // radius is subtracted afterwards (C_i = fl(cE_i - erel): one more rounding of <= u (|c_i| + E_i), covered by the
    // 16 u of the |base| term where 8.02 u are needed).
    Filter F[NT];
    float eabsMax[NT], mrg[NT], wk[NT][3];
    bool regular[NT], works[NT];
    int cx0 = 0x7FFF, cx1 = 0, cy0 = 0x7FFF, cy1 = 0, cz0 = 0x7FFF, cz1 = 0;   // the lane's cell box: the union over its working tets
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        float v[12];
        {
            const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + tet_id(k)) * 12);
            float4 a = load_f4<PIT_SCAN_TET_NT != 0>(src), bq = load_f4<PIT_SCAN_TET_NT != 0>(src + 1), c = load_f4<PIT_SCAN_TET_NT != 0>(src + 2);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
            v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        }
        float mN[4][3], det;
        {
            const float e1x = v[3] - v[0], e1y = v[4] - v[1], e1z = v[5] - v[2];
            const float e2x = v[6] - v[0], e2y = v[7] - v[1], e2z = v[8] - v[2];
            const float e3x = v[9] - v[0], e3y = v[10] - v[1], e3z = v[11] - v[2];
            cross_fma(e1x, e1y, e1z, e2x, e2y, e2z, mN[0]);            // face (v0 v1 v2): the reference's ordering 0
            cross_fma(e3x, e3y, e3z, e1x, e1y, e1z, mN[1]);            // face (v1 v0 v3): ordering 1
            cross_fma(e2x, e2y, e2z, e3x, e3y, e3z, mN[2]);            // face (v2 v3 v0): ordering 2
#pragma unroll
            for (int j = 0; j < 3; ++j) mN[3][j] = -((mN[0][j] + mN[1][j]) + mN[2][j]);   // face (v3 v2 v1): ordering 3
            det = fmaf(mN[0][0], e3x, fmaf(mN[0][1], e3y, mN[0][2] * e3z));
        }
        float blo[3], bhi[3], mx[3];                                   // box; largest |coordinate| per axis
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            blo[j] = fminf(fminf(v[j], v[3 + j]), fminf(v[6 + j], v[9 + j]));
            bhi[j] = fmaxf(fmaxf(v[j], v[3 + j]), fmaxf(v[6 + j], v[9 + j]));
            wk[k][j] = bhi[j] - blo[j];
            mx[j] = fmaxf(fabsf(blo[j]), fabsf(bhi[j]));
        }
        const float w = fmaxf(fmaxf(wk[k][0], wk[k][1]), wk[k][2]);
        // every comparison is written so that NaN yields "irregular" (NaN poisons det; Inf / huge values show up in mx)
        regular[k] = fmaxf(fmaxf(mx[0], mx[1]), mx[2]) <= kBig && w >= kWMin && fabsf(det) >= kTauSlim * ((w * w) * w);
        mrg[k] = w * kMargin;
        float elo[3], ehi[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            elo[j] = blo[j] - mrg[k];
            ehi[j] = bhi[j] + mrg[k];
        }
        // no regular query can lie in the enlarged box -> nothing to traverse
        const bool ingrid = !(ehi[0] < g.lo[0] || elo[0] > g.hi[0] || ehi[1] < g.lo[1] || elo[1] > g.hi[1] || ehi[2] < g.lo[2] || elo[2] > g.hi[2]);
        {
            const int ax0 = cell_of(elo[0], g.o[0], g.inv[0], Gx), ax1 = cell_of(ehi[0], g.o[0], g.inv[0], Gx);
            const int ay0 = cell_of(elo[1], g.o[1], g.inv[1], G), ay1 = cell_of(ehi[1], g.o[1], g.inv[1], G);
            const int az0 = cell_of(elo[2], g.o[2], g.inv[2], G), az1 = cell_of(ehi[2], g.o[2], g.inv[2], G);
            // a wide tet (wide_tet above) goes the way of the irregular ones: listed for k_finalize, not traversed, not recorded
            regular[k] = regular[k] && !(ingrid && wide_tet(ax0, ax1, ay0, ay1, az0, az1, g));
            works[k] = tet_exists(k) && regular[k] && ingrid;
            if (NT == 1) {
                cx0 = ax0; cx1 = ax1; cy0 = ay0; cy1 = ay1; cz0 = az0; cz1 = az1;
            } else if (works[k]) {
                cx0 = min(cx0, ax0); cx1 = max(cx1, ax1); cy0 = min(cy0, ay0); cy1 = max(cy1, ay1); cz0 = min(cz0, az0); cz1 = max(cz1, az1);
            }
        }
        const float sigma = det > 0.f ? 1.0f : -1.0f;
        float S[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) S[j] = fmaf(mx[j], 2.0f * kErrScale, g.pe[j]);                 // 8 u P_k + 16 u M_k
        eabsMax[k] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float n0 = mN[i][0], n1 = mN[i][1], n2 = mN[i][2];
            const float *a = i < 3 ? v : v + 3;                        // a point of the face: v0 (faces 0-2), v1 (face 3)
            const float c = fmaf(n0, a[0], fmaf(n1, a[1], n2 * a[2]));
            const float E = fmaf(fabsf(n0), S[0], fmaf(fabsf(n1), S[1], fabsf(n2) * S[2]));
            F[k].N[i][0] = sigma * n0; F[k].N[i][1] = sigma * n1; F[k].N[i][2] = sigma * n2;
            F[k].C[i] = fmaf(-sigma, c, -E);
            eabsMax[k] = fmaxf(eabsMax[k], E);
        }
    }
    bool work = works[0];
#pragma unroll
    for (int k = 1; k < NT; ++k) work = work || works[k];
    if (PIT_STOP == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { PIT_KEEP(F[0].N[i][0]); PIT_KEEP(F[0].N[i][1]); PIT_KEEP(F[0].N[i][2]); PIT_KEEP(F[0].C[i]); }
        PIT_KEEP(eabsMax[0]); PIT_KEEP(mrg[0]); PIT_KEEP(wk[0][0]); PIT_KEEP(wk[0][1]); PIT_KEEP(wk[0][2]);
        PIT_KEEP(cx0); PIT_KEEP(cx1); PIT_KEEP(cy0); PIT_KEEP(cy1); PIT_KEEP(cz0); PIT_KEEP(cz1); PIT_KEEP((int)work);
        return;
    }
    const int Gp = table_pitch(G);
    const int *tb = table + (size_t)b * cellStride;
    const float4 *sq = sortedQ + (size_t)b * Q;
    // --- footprint groups (wave-uniform scalars) ------------------------------------------------------------------------
    // group g: cell box [gx0, gx1] x [gy0, gy0 + gny) x [gz0, gz0 + gnz).  Its rows are numbered (slab << gsh) + y with a
    // power-of-two y pitch <= 4 (ids with y >= gny are empty rows); group 1's follow group 0's.  One LANE per slab reads
    // the slab's row bounds below (one 16-byte load per column), so the two groups have at most 64 slabs together.
    int gx0[2], gx1[2], gy0[2], gz0[2], gny[2], gnz[2], gsh[2];
    int gid = -1;                                                      // the lane's group
    int ncx = cx1 - cx0 + 1, ncy = cy1 - cy0 + 1;                      // cells of the lane's candidate source along x / y
    int rowsTot = 0, rowOff1 = 0, slabsTot = 0;                        // row ids in use; first id of group 1; slabs of both groups
    {
        // Up to three pivots for up to two groups.  A pivot's group holds the lanes whose (x, y) footprint is within PIT_TOLX /
        // one cell of the pivot's AND whose first slab is within kWvZReach of the pivot's: without the z condition the last
        // tets of one mesh column and the first of the next (y-adjacent: their footprints can differ by a single cell when the
        // jitter falls that way) formed ONE group spanning the whole z range — five rows, or more than 64 slabs, or more rows
        // than the stage holds — which was dropped as a whole, and all its lanes walked the global table by themselves:
        // 61,000 of 2.06 M lanes per launch at configs[2], 212,000 of 6 M at configs[3], whole waves of them, 8 of 68 us.
        // A pivot whose group is too small to be worth staging (the last one or two tets of a column at the head of a wave)
        // gives up only ITS lanes; the others get the next pivot (it used to send the whole rest of the wave to the global walk).
        lanemask_t rem = __builtin_amdgcn_ballot_w64(work);
        int ng = 0;                                                    // groups accepted so far (wave-uniform)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            gx0[p] = gx1[p] = gy0[p] = gz0[p] = 0;
            gny[p] = gnz[p] = gsh[p] = 0;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (ng == 2 || __popcll(rem) < kWvMinGroup) break;         // (lanes left over walk the global table)
            const int pl = __ffsll((long long)rem) - 1;
            const int px0 = __builtin_amdgcn_readlane(cx0, pl), px1 = __builtin_amdgcn_readlane(cx1, pl);
            const int py0 = __builtin_amdgcn_readlane(cy0, pl), py1 = __builtin_amdgcn_readlane(cy1, pl);
            const int pz0 = __builtin_amdgcn_readlane(cz0, pl);
            // (unsigned compare of the shifted difference: one instruction per bound)
            const bool in = ((rem >> lane) & 1ull) != 0ull && (unsigned)(cx0 - px0 + PIT_TOLX) <= 2u * PIT_TOLX && (unsigned)(cx1 - px1 + PIT_TOLX) <= 2u * PIT_TOLX &&
                            (unsigned)(cy0 - py0 + 1) <= 2u && (unsigned)(cy1 - py1 + 1) <= 2u && (unsigned)(cz0 - pz0 + kWvZReach) <= 2u * kWvZReach;
            const lanemask_t gm = __builtin_amdgcn_ballot_w64(in);
            rem &= ~gm;
            if (__popcll(gm) < kWvMinGroup) { REASON_COUNT(2, gm); continue; }   // its few lanes walk the global table; the rest gets the next pivot
            const unsigned ax = wave_pk_max(in ? ((unsigned)cx1 | ((0xFFFFu - (unsigned)cx0) << 16)) : 0u);
            const unsigned ay = wave_pk_max(in ? ((unsigned)cy1 | ((0xFFFFu - (unsigned)cy0) << 16)) : 0u);
            const unsigned az = wave_pk_max(in ? ((unsigned)cz1 | ((0xFFFFu - (unsigned)cz0) << 16)) : 0u);
            const int ux0 = (int)(0xFFFFu - (ax >> 16)), uy0 = (int)(0xFFFFu - (ay >> 16)), uz0 = (int)(0xFFFFu - (az >> 16));
            const int ny = (int)(ay & 0xFFFFu) - uy0 + 1, nz = (int)(az & 0xFFFFu) - uz0 + 1;
            const int sh = ny <= 1 ? 0 : 32 - __builtin_clz((unsigned)(ny - 1));              // pitch 2^sh >= ny
            if (sh > 2 || slabsTot + nz > 64 || rowsTot + (nz << sh) > ROWS) {              // too many rows: its lanes walk the global table
                REASON_COUNT(sh > 2 ? 3 : slabsTot + nz > 64 ? 4 : 5, gm);
                continue;
            }
            const bool second = ng != 0;                                // (static indices, wave-uniform selects)
            gx0[0] = second ? gx0[0] : ux0; gx1[0] = second ? gx1[0] : (int)(ax & 0xFFFFu); gy0[0] = second ? gy0[0] : uy0; gz0[0] = second ? gz0[0] : uz0;
            gny[0] = second ? gny[0] : ny; gnz[0] = second ? gnz[0] : nz; gsh[0] = second ? gsh[0] : sh;
            gx0[1] = second ? ux0 : 0; gx1[1] = second ? (int)(ax & 0xFFFFu) : 0; gy0[1] = second ? uy0 : 0; gz0[1] = second ? uz0 : 0;
            gny[1] = second ? ny : 0; gnz[1] = second ? nz : 0; gsh[1] = second ? sh : 0;
            if (second) rowOff1 = rowsTot;
            rowsTot += nz << sh;
            slabsTot += nz;
            if (in) { gid = ng; ncx = (int)(ax & 0xFFFFu) - ux0 + 1; ncy = ny; }
            ++ng;
        }
        REASON_COUNT(6, rem);                                            // [6]: left over after the last pivot
    }
    // --- the group-dependent part of the error radius ---------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        // |candidate - vertex| per axis: box edge + the cells the candidates may come from (+ 1 for the roundings of cell_of);
        // with two tets per lane the cells are those of the lane's UNION box, which contains the tet's own
        const float R0 = fmaf(2.0f, mrg[k], wk[k][0]) + (float)(ncx + 1) * g.cs[0];
        const float R1 = fmaf(2.0f, mrg[k], wk[k][1]) + (float)(ncy + 1) * g.cs[1];
        const float R2 = fmaf(2.0f, mrg[k], wk[k][2]) + (float)(cz1 - cz0 + 2) * g.cs[2];
        const float erel = fmaf(kRelScale, fmaf(wk[k][1] * wk[k][2], R0, fmaf(wk[k][0] * wk[k][2], R1, (wk[k][0] * wk[k][1]) * R2)), kErrAbs);
#pragma unroll
        for (int i = 0; i < 4; ++i) F[k].C[i] -= erel;
        F[k].twoEmax = 2.0f * (eabsMax[k] + erel);
        if (NT > 1 && !works[k]) {                                      // nothing to traverse for this tet: A_i = -inf, never accepted, never in the band
#pragma unroll
            for (int i = 0; i < 4; ++i) { F[k].N[i][0] = F[k].N[i][1] = F[k].N[i][2] = 0.f; F[k].C[i] = -INFINITY; }
            F[k].twoEmax = 0.f;
        }
    }
    if (PIT_STOP == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { PIT_KEEP(F[0].N[i][0]); PIT_KEEP(F[0].N[i][1]); PIT_KEEP(F[0].N[i][2]); PIT_KEEP(F[0].C[i]); }
        PIT_KEEP(F[0].twoEmax); PIT_KEEP(gid); PIT_KEEP(rowsTot); PIT_KEEP(rowOff1); PIT_KEEP(cz0); PIT_KEEP(cz1);
        return;
    }
    PHASE_MARK(0);                                                       // [0] load + setup + grouping
    int *resb = uniform_ptr(result + (size_t)b * Q);
    asm volatile("s_nop 4" ::: "memory");                               // VALU-written SGPR base -> vector memory: five wait states, paid once
    // accepted queries go to the tet's LDS slots: `slotA` is the byte address of the next one, saturating two rows past the
    // last real slot (so that "more than kWvSlots" stays visible); they are published after the loops
    unsigned slot0[NT], slotEnd[NT], slotA[NT];                         // byte offsets into s_hit
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        slot0[k] = (unsigned)(k * (kWvSlots + 2) * kWvThreads + tid) * 4u;
        slotEnd[k] = slot0[k] + (unsigned)(kWvSlots + 1) * (unsigned)(kWvThreads * 4);
        slotA[k] = slot0[k];
    }
    int pend0 = -1, pend1 = -1, npend = 0;                              // undecided candidates: query id, tet of the lane in bit 30
    // the four planes as two packed pairs: v_pk_fma_f32 evaluates two of them per instruction (six instructions per
    // candidate where twelve v_fma_f32 stood; every half is the same IEEE fma in the same order, so A_i is unchanged)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 Px01[NT], Py01[NT], Pz01[NT], Pc01[NT], Px23[NT], Py23[NT], Pz23[NT], Pc23[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        Px01[k] = f32x2{F[k].N[0][0], F[k].N[1][0]}; Py01[k] = f32x2{F[k].N[0][1], F[k].N[1][1]};
        Pz01[k] = f32x2{F[k].N[0][2], F[k].N[1][2]}; Pc01[k] = f32x2{F[k].C[0], F[k].C[1]};
        Px23[k] = f32x2{F[k].N[2][0], F[k].N[3][0]}; Py23[k] = f32x2{F[k].N[2][1], F[k].N[3][1]};
        Pz23[k] = f32x2{F[k].N[2][2], F[k].N[3][2]}; Pc23[k] = f32x2{F[k].C[2], F[k].C[3]};
    }
    auto test = [&](const float4 q) {
        const f32x2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
        const int qi = __float_as_int(q.w);
        float av[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const f32x2 A01 = __builtin_elementwise_fma(Px01[k], qx, __builtin_elementwise_fma(Py01[k], qy, __builtin_elementwise_fma(Pz01[k], qz, Pc01[k])));
            const f32x2 A23 = __builtin_elementwise_fma(Px23[k], qx, __builtin_elementwise_fma(Py23[k], qy, __builtin_elementwise_fma(Pz23[k], qz, Pc23[k])));
            av[k] = fminf(fminf(A01.x, A01.y), fminf(A23.x, A23.y));
        }
#pragma unroll
        for (int k = 0; k < NT; ++k)
            if (av[k] > F[k].twoEmax) {                                 // certain (twoEmax >= 0): kept in the tet's LDS slots, published
                // after the loops (an atomic per acceptance here is a vector-memory instruction with one or two live lanes in
                // most wave-iterations: 28 of them per wave kept the address unit as busy as the round-3 gathers did)
                *reinterpret_cast<int *>(reinterpret_cast<char *>(&s_hit[0][0][0]) + slotA[k]) = qi;
                slotA[k] = min(slotA[k] + (unsigned)(kWvThreads * 4), slotEnd[k]);
            }
        bool band = fabsf(av[0]) <= F[0].twoEmax;                       // rare: decided by the reference predicate after the loops
#pragma unroll
        for (int k = 1; k < NT; ++k) band = band || fabsf(av[k]) <= F[k].twoEmax;
        if (__builtin_amdgcn_ballot_w64(band) != 0ull) {                // (wave-uniform branch: the selects stay out of the common path)
            asm volatile("" ::: "memory");                              // (... and the compiler from turning the branch into selects)
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const bool bk = fabsf(av[k]) <= F[k].twoEmax;
                pend1 = bk ? pend0 : pend1;
                pend0 = bk ? (qi | (k << 30)) : pend0;
                npend = bk ? npend + 1 : npend;
            }
        }
    };
    // --- the groups' rows and queries are staged together, the lanes of both groups walk their ranges in one loop --------
    if (rowsTot > 0) {
        // One lane per slab of a footprint: the bounds of the slab's (<= 8) y rows over the group's x range are two 16-byte
        // loads of the transposed table per four rows (start column, end column) — 50 addresses per wave where one lane
        // per ROW fetched 200 four-byte words, and that fetch, not arithmetic, was the longest stage of the kernel (stage
        // timing, DESIGN.md section 3: 26 of 80 us).  Group 1's slabs follow group 0's (when group 0 was dropped: none of it).
        const int nzA = gnz[0];
        const bool second = lane >= nzA, slabLane = lane < slabsTot;
        const lanemask_t m2 = mask_of(second);
        const int zl = sel(m2, lane - nzA, lane), sh = sel(m2, gsh[1], gsh[0]), ny = sel(m2, gny[1], gny[0]);
        const int rowId0 = sel(m2, rowOff1, 0) + (zl << sh);
        int a[4], len[4];
        {
            const unsigned zrow = (unsigned)(sel(m2, gz0[1], gz0[0]) + (slabLane ? zl : 0)) * (unsigned)(Gx + 1);
            const unsigned la = ((zrow + (unsigned)sel(m2, gx0[1], gx0[0])) * (unsigned)Gp + (unsigned)sel(m2, gy0[1], gy0[0])) * 4u;
            const unsigned le = ((zrow + (unsigned)sel(m2, gx1[1], gx1[0]) + 1u) * (unsigned)Gp + (unsigned)sel(m2, gy0[1], gy0[0])) * 4u;
            const int4u a0 = ld_off_u4(tb, la), e0 = ld_off_u4(tb, le);
            a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
            len[0] = e0.x - a0.x; len[1] = e0.y - a0.y; len[2] = e0.z - a0.z; len[3] = e0.w - a0.w;
        }
        int pre[5];                                                     // exclusive prefix of the slab's rows
        pre[0] = 0;
        int lmax = 0;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            len[y] = sel(mask_of(slabLane && y < ny), len[y], 0);
            pre[y + 1] = pre[y] + len[y];
            lmax = max(lmax, len[y]);
        }
        const int incl = wave_scan_add(pre[4]);
        const int lb = incl - pre[4];                                   // staged position of the slab's first query
        const int N = __builtin_amdgcn_readlane(incl, 63);
        const int Rtot = rowsTot;
        // a row that does not fit a chunk by itself (very dense queries): nothing is staged, everybody walks the global table
        const bool fitsRows = __builtin_amdgcn_ballot_w64(lmax > CAP) == 0ull && N < 65536;
        if (slabLane) {
#pragma unroll
            for (int y = 0; y < 4; ++y)
                if (y < (1 << sh)) {                                    // (ids past the footprint's height: empty rows at the slab's end)
                    W.rowBase[rowId0 + y] = (unsigned short)(lb + pre[y]);
                    W.delta[rowId0 + y] = a[y] - (lb + pre[y]);
                }
        }
        if (lane == 0) W.rowBase[Rtot] = (unsigned short)N;
        wave_sync();
        if (PIT_STOP == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { PIT_KEEP(F[0].N[i][0]); PIT_KEEP(F[0].N[i][1]); PIT_KEEP(F[0].N[i][2]); PIT_KEEP(F[0].C[i]); }
            PIT_KEEP(F[0].twoEmax); PIT_KEEP(gid); PIT_KEEP(N); PIT_KEEP(lb); PIT_KEEP(cz0); PIT_KEEP(cz1);
            return;
        }
        PHASE_MARK(4);                                                   // [4] row bounds, scan
        // the lane's rows: the slabs [cz0, cz1] of its group's footprint, all of the footprint's y rows
        // (sel: the SGPR-pair form of the select; written as ?: the compiler emits the VCC form, which gfx950 issues eight times slower)
        const lanemask_t mine2 = mask_of(gid == 1);
        const int rowOff = sel(mine2, rowOff1, 0), gz = sel(mine2, gz0[1], gz0[0]), gs = sel(mine2, gsh[1], gsh[0]);
        int rn = rowOff + ((cz0 - gz) << gs);                          // next row to walk
        const int re = rowOff + ((cz1 + 1 - gz) << gs);
        if (!fitsRows) { REASON_COUNT(7, __builtin_amdgcn_ballot_w64(gid >= 0)); gid = -1; }
        int r0 = fitsRows ? 0 : Rtot;
#pragma unroll 1
        while (r0 < Rtot) {                                              // chunks of whole rows
            const int B0 = __builtin_amdgcn_readfirstlane(W.rowBase[r0]);
            int r1 = Rtot;
            if (N - B0 > CAP) {
                r1 = r0;
#pragma unroll 1
                for (int j = 0; j < ROWS / 64; ++j) {
                    const int rc = r0 + 1 + lane + 64 * j;
                    const int cnt = rc <= Rtot ? W.rowBase[rc] - B0 : 0x7FFFFFFF;
                    const lanemask_t fits = __builtin_amdgcn_ballot_w64(cnt <= CAP);  // a prefix of the lanes (counts are monotone)
                    if (fits == ~0ull) { r1 += 64; continue; }
                    r1 += __ffsll((long long)~fits) - 1;
                    break;
                }
            }
            const int Nc = __builtin_amdgcn_readfirstlane(W.rowBase[r1]) - B0;
            // owner of every staged position: non-empty rows mark their first position, a max-scan spreads the marks
            for (int i = lane; i < Nc; i += 64) W.marker[i] = 0;
            wave_sync();
            if (slabLane) {                                              // (row bounds re-read from LDS: 17 registers less across the loop)
                int p0 = W.rowBase[rowId0];
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    if (y < (1 << sh)) {
                        const int r = rowId0 + y, p1 = W.rowBase[r + 1];
                        if (p1 > p0 && r >= r0 && r < r1) W.marker[p0 - B0] = (typename Stage::marker_t)(r + 1);
                        p0 = p1;
                    }
                }
            }
            wave_sync();
            int carry = 0;
#pragma unroll
            for (int j = 0; j < (CAP + 63) / 64; ++j)
                if (j * 64 < Nc) {
                    const int i = lane + 64 * j;
                    const int own = max(wave_scan_max(i < Nc ? W.marker[i] : 0), carry);
                    carry = __builtin_amdgcn_readlane(own, 63);
                    if (i < Nc) W.q[i] = ld_off<float4>(sq, (unsigned)(B0 + i + W.delta[own - 1]) * 16u);
                }
            wave_sync();
            PHASE_MARK(5);                                               // [5] owners, copy
            PHASE_COUNT(7, Nc);
            PHASE_COUNT(15, 1);
            // the lane's rows inside the chunk: one contiguous range of staged queries.  Two candidates per trip of the
            // source loop, each in its own registers, so that the next one is on its way while this one is tested and nothing
            // has to be moved.
            // (A matrix-pipe version of this loop — 16 tets at a time, four lanes per tet, v_mfma_f32_4x4x1 giving the four plane
            // values of a candidate to one lane — was measured: the vector instructions around the matrix ones (dead-slot masks,
            // minima, acceptance bookkeeping, the transposition through LDS) are as many as the 14 it replaces: 94 vs 82 us.)
            const int lo = max(rn, r0), hi = min(re, r1);
            if (PIT_STOP != 4 && gid >= 0 && lo < hi) {
                unsigned c = (unsigned)(W.rowBase[lo] - B0) * 16u;      // byte offsets into W.q
                const unsigned e = (unsigned)(W.rowBase[hi] - B0) * 16u;
                PHASE_COUNT(13, (e - c) >> 4);
                rn = hi;
                const char *qb0 = reinterpret_cast<const char *>(&W.q[0]);
                if (c < e) {
                    float4 qa = *reinterpret_cast<const float4 *>(qb0 + c);
                    for (;;) {
                        const float4 qb = *reinterpret_cast<const float4 *>(qb0 + c + 16);   // (one entry past the range: harmless)
                        test(qa);
                        c += 16;
                        if (c >= e) break;
                        qa = *reinterpret_cast<const float4 *>(qb0 + c + 16);
                        test(qb);
                        c += 16;
                        if (c >= e) break;
                    }
                }
            }
            wave_sync();
            PHASE_MARK(6);                                               // [6] traversal of the chunk
            r0 = r1;
        }
    }
    PHASE_COUNT(14, __popcll(__builtin_amdgcn_ballot_w64(work && gid < 0)));
    PHASE_MARK(1);                                                       // [1] staged traversal
    // --- global walk for the lanes (slabs) no group covered: the slab cursor of k_tet_scan_slab ---------------------------
    if (PIT_DBG_WALK && work && gid < 0) {
        CellBox cb;
        {
            cb.cx0 = cb.cy0 = cb.cz0 = 0x7FFF;
            cb.cx1 = cb.cy1 = cb.cz1 = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (NT > 1 && !works[k]) continue;
                float tv[12], mrg2;
                const float *src = tet + ((size_t)b * T + tet_id(k)) * 12;
#pragma unroll
                for (int j = 0; j < 12; ++j) tv[j] = src[j];
                CellBox c1;
                tet_cell_box(tv, g, G, Gx, c1, mrg2);
                cb.cx0 = min(cb.cx0, c1.cx0); cb.cx1 = max(cb.cx1, c1.cx1); cb.cy0 = min(cb.cy0, c1.cy0); cb.cy1 = max(cb.cy1, c1.cy1);
                cb.cz0 = min(cb.cz0, c1.cz0); cb.cz1 = max(cb.cz1, c1.cz1);
            }
        }
        const int cx0 = cb.cx0, cx1 = cb.cx1, cy0 = cb.cy0, cy1 = cb.cy1, cz1 = cb.cz1, czn = cb.cz0;
        const int ny = cy1 - cy0 + 1;
        const unsigned lineB = (unsigned)Gp * 4u;
        const unsigned slabB = (unsigned)(Gx + 1) * lineB;
        const unsigned dxB = (unsigned)(cx1 + 1 - cx0) * lineB;
        unsigned offS = ((unsigned)czn * (unsigned)(Gx + 1) + (unsigned)cx0) * lineB + (unsigned)cy0 * 4u;
        int cz = czn, yoff = 0;
        int4u nS = ld_off_u4(tb, offS), nE = ld_off_u4(tb, offS + dxB);
        bool haveNext = true;
        int c = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        while (c < P4 || haveNext) {
            if (c >= P4) {
                const int rem = ny - yoff;
                const int n0 = nE.x - nS.x;
                const int n1 = rem > 1 ? nE.y - nS.y : 0;
                const int n2 = rem > 2 ? nE.z - nS.z : 0;
                const int n3 = rem > 3 ? nE.w - nS.w : 0;
                P1 = n0; P2 = P1 + n1; P3 = P2 + n2; P4 = P3 + n3;
                o0 = nS.x; o1 = nS.y - P1; o2 = nS.z - P2; o3 = nS.w - P3;
                c = 0;
                asm volatile("" : "+v"(P1), "+v"(P2), "+v"(P3), "+v"(P4), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                const bool wrap = rem <= 4;
                offS += wrap ? slabB - (unsigned)yoff * 4u : 16u;
                yoff = wrap ? 0 : yoff + 4;
                cz += wrap ? 1 : 0;
                haveNext = cz <= cz1;
                if (haveNext) {
                    nS = ld_off_u4(tb, offS);
                    nE = ld_off_u4(tb, offS + dxB);
                }
            }
            if (c < P4) {
                const int jq = c + (c < P1 ? o0 : c < P2 ? o1 : c < P3 ? o2 : o3);
                test(ld_off<float4>(sq, (unsigned)jq * 16u));
                ++c;
            }
        }
    }
    PHASE_MARK(2);                                                       // [2] global walk
    if (PIT_STOP == 5) { PIT_KEEP(slotA[0]); PIT_KEEP(pend0); PIT_KEEP(pend1); PIT_KEEP(npend); return; }   // (everything but the publish / record tail)
    if (!valid) return;
    int hcnt[NT];                                                        // accepted (kWvSlots + 1 stands for "more than kWvSlots")
#pragma unroll
    for (int k = 0; k < NT; ++k) hcnt[k] = (int)((slotA[k] - slot0[k]) / (unsigned)(kWvThreads * 4));
    if (PIT_DBG_PEND && npend > 0) {                                     // undecided candidates (rare)
        atomicAdd(&counters[gridDim.y * 4 + b * 4 + 1], npend);         // statistics: candidates decided exactly
        if (npend > 2) {
#pragma unroll
            for (int k = 0; k < NT; ++k)
                if (works[k]) {
                    const int te = tet_id(k);
                    hcnt[k] = exact_rescan_slots(tet + ((size_t)b * T + te) * 12, te, tb, sq, result + (size_t)b * Q, gparam + b * kGridWords, G, Gx,
                                                 &s_hit[k][0][tid]);
                }
        } else {
            for (int j = 0; j < npend; ++j) {
                const int pq = j == 0 ? pend0 : pend1;
                const int k = NT == 1 ? 0 : (pq >> 30) & 1, qi = pq & 0x3FFFFFFF;
                const int te = tet_id(k);
                const float *pp = pts + ((size_t)b * Q + qi) * 3;
                if (exact_accept(tet + ((size_t)b * T + te) * 12, pp[0], pp[1], pp[2]) > 0.f) {
                    atomicMin(&result[(size_t)b * Q + qi], te);
                    if (NT == 1 || k == 0) { s_hit[0][min(hcnt[0], kWvSlots)][tid] = qi; ++hcnt[0]; }
                    else { s_hit[NT - 1][min(hcnt[NT - 1], kWvSlots)][tid] = qi; ++hcnt[NT - 1]; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (NT > 1 && !tet_exists(k)) continue;                          // (odd list: the last lane has one tet)
        // (the addresses of this tet, its record, ... are formed from `te` HERE: hoisted to the top of the kernel they are
        // six more live registers across the staged loop — the difference between five and six waves per SIMD)
        int te = tet_id(k);
        asm volatile("" : "+v"(te));
        if (!regular[k]) {                                               // (out-of-line call placed where almost nothing is live)
            irregular_tet_slow(tet, te, b, T, Q, pts, counters, irregT, irregQ, result, hits);
            continue;
        }
        int hc = hcnt[k];
        if (hc > kWvSlots) {                                             // more acceptances than slots (dense queries): the out-of-line exact
            const float *tv = tet + ((size_t)b * T + te) * 12;           // walk publishes every one of them
            hc = exact_rescan_slots(tv, te, tb, sq, result + (size_t)b * Q, gparam + b * kGridWords, G, Gx, &s_hit[k][0][tid]);
        }
#pragma unroll 1
        for (int i = 0; i < kWvSlots; ++i) {                             // publish: one atomic instruction per slot level in use
            if (__builtin_amdgcn_ballot_w64(hc > i) == 0ull) break;
#if PIT_PROBE_SCATTER == 0
            if (hc > i && !(PIT_PROBE_SKIP & 1)) atomic_smin_off_nh(resb, (unsigned)s_hit[k][i][tid] * 4u, te);   // (PIT_PROBE_SKIP: timing probes)
#else
            // timing probe only (wrong spill records): what do per-hit scattered output stores cost next to the atomics?
            if (hc > i && spill) {
                const int qi = s_hit[k][i][tid];
                char *pb = reinterpret_cast<char *>(spill);
                const size_t nq = (size_t)gridDim.y * Q, iq = (size_t)b * Q + qi;
                bool win = true;
                if (PIT_PROBE_SCATTER == 1) atomic_smin_off_nh(resb, (unsigned)qi * 4u, te);
                else win = atomicMin(&result[iq], te) > te;
                if (win && PIT_PROBE_SCATTER == 3) {
                    const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + te) * 12);
                    const float *pp = pts + iq * 3;
                    *reinterpret_cast<float4 *>(pb + iq * 16) = bary_weights(src[0], src[1], src[2], pp[0], pp[1], pp[2]);
                    *reinterpret_cast<float *>(pb + nq * 16 + iq * 4) = (float)te;
                    *reinterpret_cast<float *>(pb + nq * 20 + iq * 4) = (float)qi;
                } else if (win) {
                    *reinterpret_cast<float4 *>(pb + iq * 16) = make_float4((float)te, (float)qi, 0.25f, 0.5f);
                    *reinterpret_cast<float *>(pb + nq * 16 + iq * 4) = (float)te;
                    *reinterpret_cast<float *>(pb + nq * 20 + iq * 4) = (float)qi;
                }
            }
#endif
        }
        if (hits) {
            // <= 2 accepted: the record; 3..kWvSlots: the record (flagged) + the spill record; more: overflow marker (the hits are
            // then carried by the uncovered list, see k_finalize)
            int h[kRecSlots + kSpillSlots];
#pragma unroll
            for (int i = 0; i < kRecSlots + kSpillSlots; ++i) h[i] = -1;
#pragma unroll
            for (int i = 0; i < kRecSlots; ++i)
                if (i < hc) h[i] = s_hit[k][i][tid];
            const bool spilled = hc > kRecSlots && hc <= kWvSlots && spill != nullptr && PIT_PROBE_SCATTER == 0;   // (probe builds scribble over the spill records)
            const bool over = hc > kRecSlots && !spilled;
            if (spilled) {
#pragma unroll
                for (int i = kRecSlots; i < kWvSlots; ++i)
                    if (i < hc) h[i] = s_hit[k][i][tid];
                spill[(size_t)b * T + te] = make_int4(h[2], h[3], h[4], h[5]);
                h[0] |= kHitSpilled;
            }
            if (over) note_overflow(counters, gridDim.y, b, te);
            if (!(PIT_PROBE_SKIP & 2)) store_rec(hits + (size_t)b * T + te, over ? make_int2(-1, kHitOverflow) : make_int2(h[0], h[1]));
        }
        irregular_tail(tet, te, b, T, Q, pts, counters, irregQ, result);
    }
    PHASE_MARK(3);                                                       // [3] exact decisions, records
    SPAN_MARK(1);
}

#define PIT_SCAN_PARAMS                                                                                                               \
    const float *__restrict__ tet, int T, int Q, const float *__restrict__ gparam, int G, int Gx, const int *__restrict__ table,      \
        long long cellStride, const float4 *__restrict__ sortedQ, int *result, int *counters, int *irregT, int2 *hits,                \
        const float *__restrict__ pts, const int *__restrict__ irregQ, int *ucount, int hpad, int4 *spill, const int *__restrict__ order
#define PIT_SCAN_FWD tet, T, Q, gparam, G, Gx, table, cellStride, sortedQ, result, counters, irregT, hits, pts, irregQ, ucount, hpad, spill, order
template <bool ORD>
__global__ __launch_bounds__(kWvThreads, PIT_WAVES2) void k_tet_scan_wave(PIT_SCAN_PARAMS)
{
    tet_scan_wave_body<ORD, 1>(PIT_SCAN_FWD);
}
template <bool ORD>
__global__ __launch_bounds__(kWvThreads, PIT_WAVES_PAIR) void k_tet_scan_pair(PIT_SCAN_PARAMS)
{
    tet_scan_wave_body<ORD, 2>(PIT_SCAN_FWD);
}
#undef PIT_SCAN_PARAMS
#undef PIT_SCAN_FWD

// QPT queries per thread (round 6 experiment; 1 is what ships).  A query's work is two memory round trips in a row — its winner,
// then the winner's 48-byte record and prediction — so two or more chains per thread, every gather issued before the first is
// used, looked like free memory-level parallelism.  Measured inside the configs[2] step (tools/probes/sort_probe.py, two runs
// each): QPT 1 / 2 / 3 / 4 = 25.1-25.8 / 26.8-27.5 / 28.6 / 29.8-30.0 us — the kernel is bound by how fast the texture path
// takes 64 scattered 16-byte requests per instruction, not by how many are in flight; more per thread only adds registers.
// Query j of the thread is (chunk * QPT + j) * 256 + tid, so every access stays coalesced per j.
#ifndef PIT_FIN_QPT
#define PIT_FIN_QPT 1
#endif
template <int QPT>
__global__ __launch_bounds__(256) void k_finalize(const float *__restrict__ tet, const float *__restrict__ pts, int T,
                                                  int Q, const int *__restrict__ result, float *cond, float *bary,
                                                  const float *__restrict__ pred, float *occ, const int2 *__restrict__ hits,
                                                  int *ucount, int *ulist, const int *__restrict__ counters,
                                                  const int *__restrict__ irregT, int hpad, int pin, const float *__restrict__ gparam)
{
    __shared__ int s_cnt[4], s_base;
    const int2 sb = shape_block(pin);                                  // shape-per-XCD placement
    const int b = sb.x;
    int q[QPT], r[QPT];
    bool live[QPT];                                                 // (no early return: the append below has barriers)
    size_t i[QPT];
    float px[QPT], py[QPT], pz[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        q[j] = (sb.y * QPT + j) * (int)blockDim.x + (int)threadIdx.x;
        live[j] = q[j] < Q;
        i[j] = (size_t)b * Q + (live[j] ? q[j] : 0);
        r[j] = live[j] ? result[i[j]] : kMiss;
    }
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        const float *p = pts + i[j] * 3;
        px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
    }
    // irregular tets (not certified for the grid filter; normally none) are tested here against
    // every query: query-centric, so no atomics and no extra launch
    const int nIrregT = counters ? counters[b * 4 + 0] : 0;
    if (nIrregT > 0) {
        for (int k = 0; k < nIrregT; ++k) {
            const int t = irregT[(size_t)b * T + k];
            float v[12];
            const float *src = tet + ((size_t)b * T + t) * 12;
#pragma unroll
            for (int jj = 0; jj < 12; ++jj) v[jj] = src[jj];
            Planes P;
            make_planes(v, P);
#pragma unroll
            for (int j = 0; j < QPT; ++j)
                if (live[j] && accept(P, px[j], py[j], pz[j])) r[j] = min(r[j], t);
        }
    }
    // every gather of the thread leaves here, before anything waits for one of them
    bool hit[QPT];
    float4 t0[QPT], t1[QPT], t2[QPT];
    float pr[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        hit[j] = r[j] != kMiss;
        const float4 *src = reinterpret_cast<const float4 *>(tet + ((size_t)b * T + (hit[j] ? r[j] : 0)) * 12);
        if (bary && hit[j]) { t0[j] = src[0]; t1[j] = src[1]; t2[j] = src[2]; }
        pr[j] = (occ && live[j]) ? pred[(size_t)b * T + (hit[j] ? r[j] : 0)] : 0.f;    // paste_occ: misses alias tet 0 (deftet.py:133-135)
    }
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        if (live[j]) {
            stream_store(cond + i[j], hit[j] ? (float)r[j] : -1.0f);  // :177, :149
            if (occ) stream_store(occ + i[j], pr[j]);
        }
        bool uncovered = false;
        if (hits && hit[j]) {
            // is this hit in its tet's record?  (not if the tet overflowed / is irregular, or if the
            // query took the irregular-query side path, which records nothing)
            // records are complete unless some tet is irregular or overflowed (wave-uniform test: the
            // per-hit gather of the record is skipped for ordinary meshes)
            // (a query the grid did not bin — NaN / Inf / huge, or outside the box the grid was given — took the side path, which
            // records nothing; with the box measured from the queries themselves the second test is true for every regular query:
            // gparam[19] says whether the grid came from a hint)
            bool covered = query_regular(px[j], py[j], pz[j]);
            if (gparam && gparam[b * kGridWords + 19] != 0.f) covered = covered && query_binned(px[j], py[j], pz[j], load_grid(gparam + b * kGridWords));
            const bool sidePath = !covered;
            if (counters) {
                const int nB = gridDim.y, nOvf = counters[nB * 4 + b * 4 + 2];
                if (counters[b * 4 + 0] > 0 || nOvf > kOvfCap) {
                    // irregular tets exist, or more overflowed tets than the list holds: read the winning tet's record
                    covered = covered && hits[(size_t)b * T + r[j]].y != kHitOverflow;
                } else {
                    // the usual case: a handful of overflowed tets per shape, listed; wave-uniform scalar reads, no gather
                    const int *ovf = counters + nB * 8 + b * kOvfCap;
                    for (int k = 0; k < nOvf; ++k) covered = covered && ovf[k] != r[j];
                }
            }
            if (!covered) {
                uncovered = true;
                if (sidePath) ucount[2 * hpad + b] = 1;              // its tet's record may be complete: every lane must look
            }
        }
        if (hits) {
            // Append the uncovered hits of this workgroup with ONE atomic: the counter is a single address per shape, and
            // same-address atomics serialise at the memory side (~50 ns each) — where many tets overflow their four-slot
            // record (configs[1]: one query per tet on average, ~650 uncovered hits per shape) one atomic per hit made this
            // kernel 50 us instead of 25.
            const unsigned long long m = __ballot(uncovered);
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (j > 0) __syncthreads();                              // s_cnt / s_base of the previous query row are done with
            if (lane == 0) s_cnt[wave] = __popcll(m);
            __syncthreads();
            if (threadIdx.x == 0) {
                const int tot = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
                s_base = tot > 0 ? atomicAdd(&ucount[b], tot) : 0;
            }
            __syncthreads();
            if (uncovered) {
                int off = s_base + __popcll(m & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; ++w) off += s_cnt[w];
                ulist[(size_t)b * Q + off] = q[j];
            }
        }
    }
    if (!bary) return;
#pragma unroll
    for (int j = 0; j < QPT; ++j) {
        if (!live[j]) continue;
        float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hit[j]) wq = bary_weights(t0[j], t1[j], t2[j], px[j], py[j], pz[j]);
        stream_store(reinterpret_cast<float4 *>(bary) + i[j], wq);
    }
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
    const float4 t0 = load_f4<PIT_BWD_TET_NT != 0>(src), t1 = load_f4<PIT_BWD_TET_NT != 0>(src + 1), t2 = load_f4<PIT_BWD_TET_NT != 0>(src + 2);
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

// bit-exact read / write of a float at the memory side (device-scope RMW atomics bypass the per-XCD L2s): what the
// ticket reduction below exchanges between workgroups that may sit on different XCDs
__device__ __forceinline__ float mem_read_f32(float *p) { return __int_as_float(atomicOr(reinterpret_cast<int *>(p), 0)); }
__device__ __forceinline__ void mem_write_f32(float *p, float v)
{
    const int old = atomicExch(reinterpret_cast<int *>(p), __float_as_int(v));
    asm volatile("" ::"v"(old));                                   // keep the RETURNING form: its completion is what s_waitcnt observes
}

constexpr int kMissStride = 80;    // floats of workspace per shape: kMissParts partials + the tet-0 lane's own sum

// One launch for the whole backward:
//  * every tet lane adds up the (<= 4) hits of its record in ascending query order — no atomics;
//  * a tet whose record is marked overflowed (or any tet, when the uncovered list holds NaN/Inf/huge queries) gets
//    its hits from the forward's uncovered list: the WAVE scans the list for the lane (64 entries per step) and adds
//    the tet's hits up in ASCENDING QUERY ORDER, the order the record path uses — the list itself is in whatever order
//    k_finalize's workgroups appended to it, so (round 5) the wave takes the smallest query id not yet added, one pass
//    over the list per hit; no atomics, and the whole gradient is bit-reproducible from run to run.  (Rounds 2-4 let
//    the matching lanes evaluate one hit each and added the lanes up with a butterfly: fixed tree, but which lane held
//    which hit followed the list order — the last bits of an overflowed tet's gradient changed from run to run.  A tet
//    with so many unrecorded hits that (hits) x (list length / 64) exceeds kRescanOrderedSteps still takes that path:
//    one pass per hit is then too slow.)
//  * paste_occ sends every miss to tet 0 (deftet.py:133-135), so grad_pred[b,0] also gets the sum of grad_occ over
//    the misses: the first nMissParts workgroups of a shape each sum a slice of the queries on the side (hidden under
//    the kernel's own traffic), hand their partial to memory and draw a ticket; the workgroup that draws the last one
//    adds the partials up in index order, together with the tet-0 lane's own sum, and writes grad_pred[b,0].
//    (Rounds 1-2 needed a second launch, k_bary_bwd_tail, for the last two items.)
// The rare part of k_bary_bwd_hits: for every tet of the wave in `need`, the wave walks the shape's list of unrecorded
// hits and leaves the tet's 12 gradient sums + its grad_pred sum in park[.][tid] of the tet's lane (LDS), to be added
// after the record's own hits.  It runs BEFORE the kernel's accumulators exist, so that the two phases' registers do
// not add up: 110 -> 96 VGPRs, 4 -> 5 waves per SIMD, 69.5 -> 63.1 us inside the configs[2] step.  (Out of line with
// the register budget of 6 waves: the same time, its callee-saved spills cost what the occupancy gains; forcing 6
// waves on the inlined form spills on the main path: 84 us.)  Returns whether this lane's tet was one of them.
// One pass over the list per hit is affordable while (hits of the tet) x (64-entry steps of the list) stays below this; beyond
// it (thousands of queries inside one tet) the matching lanes evaluate one hit each and a butterfly adds the lanes up
constexpr long long kRescanOrderedSteps = 16384;
__device__ __forceinline__ bool bwd_rescan(const float *__restrict__ tet, const float *__restrict__ pts, const float *__restrict__ cond,
                                        const float *__restrict__ grad_w, const float *__restrict__ gocc, float *grad_pts,
                                        bool want_pred, const int *__restrict__ ulist, int b, int T, int Q, int nU,
                                        unsigned long long need, int t, float (*park)[256])
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int *__restrict__ lst = ulist + (size_t)b * Q;
    bool parked = false;
    while (need) {
        const int L = __ffsll((long long)need) - 1;
        need &= need - 1;
        const int tL = __shfl(t, L);
        const float tLf = (float)tL;
        // smallest listed query > cur that this tet won (a hit recorded by the overflowing tet's own record cannot be in the
        // list: overflowed records are ignored by the caller); the first call also counts the tet's listed hits
        int m = 0;
        auto next_hit = [&](int cur, bool count) {
            int best = 0x7FFFFFFF;
            for (int base = 0; base < nU; base += 64) {
                const int e = base + lane;
                const int q = e < nU ? lst[e] : -1;
                const bool mine = q > cur && cond[(size_t)b * Q + q] == tLf;
                if (mine) best = min(best, q);
                if (count) m += __popcll(__ballot(mine));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off));
            return best;
        };
        int best = next_hit(-1, true);
        if (m == 0) continue;                                       // (the usual answer when every lane has to look: irregular queries)
        const bool ordered = (long long)m * ((nU + 63) >> 6) <= kRescanOrderedSteps;   // wave-uniform
        float part[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) part[k] = 0.f;
        TetGrad g;
        tet_grad_setup(tet, (size_t)b * T + tL, g);
        // ONE instance of the evaluation for both modes.  ordered: the whole wave evaluates hit `best` (wave-uniform values: every
        // lane holds the same sums), then looks for the next; otherwise: every lane evaluates its own entry of the next 64
        for (int base = 0;;) {
            int q;
            bool mine;
            if (ordered) {
                if (best == 0x7FFFFFFF) break;
                q = best;
                mine = true;
            } else {
                if (base >= nU) break;
                const int e = base + lane;
                q = e < nU ? lst[e] : -1;
                mine = q >= 0 && cond[(size_t)b * Q + q] == tLf;
                base += 64;
            }
            if (mine) {
                const size_t i = (size_t)b * Q + q;
                float G3[3];
                tet_grad_add(g, pts + i * 3, reinterpret_cast<const float4 *>(grad_w)[i], part, G3);
                if (grad_pts) { grad_pts[i * 3] = G3[0]; grad_pts[i * 3 + 1] = G3[1]; grad_pts[i * 3 + 2] = G3[2]; }
                if (want_pred) part[12] += gocc[i];
            }
            if (ordered) best = next_hit(best, false);
        }
        if (!ordered) {
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                float v = part[k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);  // fixed tree; which lane holds which hit follows the list order
                part[k] = v;
            }
        }
        if (lane == L) {                     // parked in LDS until the record's own hits are summed
#pragma unroll
            for (int k = 0; k < 13; ++k) park[k][tid] = part[k];
            parked = true;
        }
    }
    return parked;
}

#ifndef PIT_BWD_WAVES
#define PIT_BWD_WAVES 7
#endif
// SPARSE (deftet_point_in_tet_bwd_to_vertices_f32): grad_tet is a workspace whose only reader is the vertex gather
// (vertex_ops.hip: k_gather_bwd_rows).  71 % of the rows are zero at BASELINE configs[2] (0.34 hits per tet), so a wave writes
// only the rows of the tets that accepted something, COMPACTED to the start of its 64-row block in lane order, and one 64-bit
// word per wave (rowMask[b][t / 64], the ballot of those lanes) tells the reader which tets they belong to: rank = popcount of
// the mask below the tet's bit.  29 MB of contiguous stores instead of 99 MB, and the reader fetches a third of the lines.
template <bool SPARSE>
__global__ __launch_bounds__(256, PIT_BWD_WAVES) void k_bary_bwd_hits(const float *__restrict__ tet, const float *__restrict__ pts,
                                                       const float *__restrict__ cond, const float *__restrict__ grad_w,
                                                       const int2 *__restrict__ hits, int T, int Q, float *grad_tet,
                                                       float *grad_pts, int accumulate, const float *__restrict__ gocc,
                                                       float *grad_pred, float *missPart, int nMissParts, int *hitWords,
                                                       const int *__restrict__ ulist, int pad, const int4 *__restrict__ spill, int pin,
                                                       unsigned long long *rowMask)
{
    __shared__ float wsum[4];
    __shared__ float s_vals[kMissStride];
    __shared__ int s_last;
    // One buffer for two uses that never overlap in time: `s_park` (the rare rescan parks its sums there until the record's own
    // hits have been added) and `s_rows` (the wave's 48-byte rows on their way out).  The rescan only runs when the shape has
    // uncovered hits (nU > 0, the same for every thread of the workgroup), and then a barrier separates the last read of s_park
    // from the first write of s_rows; without it the kernel held 25.9 KB per workgroup — six per compute unit where its 70
    // registers allow seven.
    __shared__ __attribute__((aligned(16))) float s_buf[13 * 256];
    float4(*s_rows)[192] = reinterpret_cast<float4(*)[192]>(s_buf);   // [4][192] float4 = 12 KB of the 13 KB
    float(*s_park)[256] = reinterpret_cast<float(*)[256]>(s_buf);     // [13][256] float
    const int2 sb = shape_block(pin);                                  // shape-per-XCD placement
    const int b = sb.x, bx = PIT_BWD_REV ? (int)gridDim.x - 1 - sb.y : sb.y, tid = threadIdx.x, lane = tid & 63;
    const bool side = grad_pred && bx < nMissParts;                // block-uniform
    float missPartial = 0.f;
    if (side) {
        float gm = 0.f;
        for (int q = bx * blockDim.x + tid; q < Q; q += nMissParts * blockDim.x)
            if (cond[(size_t)b * Q + q] < 0.f) gm += gocc[(size_t)b * Q + q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gm += __shfl_xor(gm, off);
        if (lane == 0) wsum[tid >> 6] = gm;
        __syncthreads();
        // (the same value in every thread of the workgroup: kept in a scalar register — the kernel has not one vector
        // register to spare at seven waves per SIMD, and a spill here is a scratch store + load in EVERY wave of the launch)
        missPartial = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]))));
    }
    const int t = bx * blockDim.x + tid;
    const bool live = t < T;
    int2 h = live ? stream_load(hits + (size_t)b * T + t) : make_int2(-1, -1);
    const bool spilled = h.y != kHitOverflow && h.x >= 0 && (h.x & kHitSpilled) != 0;   // up to four more in the spill record
    if (spilled) h.x &= ~kHitSpilled;
    bool parked = false;
    // hits that are in no record (overflowed / irregular tets; NaN/Inf/huge queries): the forward listed them
    const int nU = hitWords[b];
    if (nU > 0) {                                                  // wave-uniform (scalar load)
        const bool anyQ = hitWords[2 * pad + b] != 0;              // entries of irregular queries: their tets' records look complete
        const unsigned long long need = __ballot(live && (h.y == kHitOverflow || anyQ));
        if (need) parked = bwd_rescan(tet, pts, cond, grad_w, gocc, grad_pts, grad_pred != nullptr, ulist, b, T, Q, nU, need, t, s_park);
    }
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    float gp = 0.f;
    if (h.x >= 0 && h.y != kHitOverflow) {                         // slots fill in order: x < 0 means no accepted query
        TetGrad g;
        tet_grad_setup(tet, (size_t)b * T + t, g);
        // the records list the accepted queries in traversal order, which depends on the (arbitrary)
        // order of queries inside a grid cell: sort the ids so that the fp32 sums below are
        // added in the same order on every run (empty slots, -1, go last)
        unsigned hu[8] = {(unsigned)h.x, (unsigned)h.y, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
        const float tf = (float)t;
        auto add_hit = [&](int q) {
            if (q < 0) return;
            const size_t i = (size_t)b * Q + q;
            // all four gathers leave together: the winner test below only selects (waiting for `cond` before asking for
            // the rest made every slot two memory latencies long)
            const float c = cond[i], go = grad_pred ? gocc[i] : 0.f;
            const float pq[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
            const float4 gw = reinterpret_cast<const float4 *>(grad_w)[i];
            const bool won = c == tf;                              // else: accepted here, but a lower-index tet won the query
            float G3[3], a2[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) a2[k] = acc[k];
            tet_grad_add(g, pq, gw, a2, G3);
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k] = won ? a2[k] : acc[k];
            if (won && grad_pts) { grad_pts[i * 3] = G3[0]; grad_pts[i * 3 + 1] = G3[1]; grad_pts[i * 3 + 2] = G3[2]; }
            gp = won ? gp + go : gp;
        };
#define DEFTET_CSWAP(a, b) { const unsigned lo_ = min(hu[a], hu[b]), hi_ = max(hu[a], hu[b]); hu[a] = lo_; hu[b] = hi_; }
        if (!__any(spilled)) {                                       // the usual wave: two slots, one comparator
            DEFTET_CSWAP(0, 1)
#pragma unroll 1
            for (int k = 0; k < 2 && __any(hu[0] != ~0u); ++k) {     // sorted: empty slots are last, most records hold one hit
                add_hit((int)hu[0]);
                hu[0] = hu[1]; hu[1] = ~0u;
            }
        } else {
            if (spilled) {
                const int4 h2 = spill[(size_t)b * T + t];
                hu[2] = (unsigned)h2.x; hu[3] = (unsigned)h2.y; hu[4] = (unsigned)h2.z; hu[5] = (unsigned)h2.w;
            }
            // 19-comparator sorting network for eight keys (empty slots are 0xFFFFFFFF and end up last)
            DEFTET_CSWAP(0, 1) DEFTET_CSWAP(2, 3) DEFTET_CSWAP(4, 5) DEFTET_CSWAP(6, 7)
            DEFTET_CSWAP(0, 2) DEFTET_CSWAP(1, 3) DEFTET_CSWAP(4, 6) DEFTET_CSWAP(5, 7)
            DEFTET_CSWAP(1, 2) DEFTET_CSWAP(5, 6) DEFTET_CSWAP(0, 4) DEFTET_CSWAP(3, 7)
            DEFTET_CSWAP(1, 5) DEFTET_CSWAP(2, 6)
            DEFTET_CSWAP(1, 4) DEFTET_CSWAP(3, 6)
            DEFTET_CSWAP(2, 4) DEFTET_CSWAP(3, 5)
            DEFTET_CSWAP(3, 4)
#pragma unroll 1
            for (int k = 0; k < 6 && __any(hu[0] != ~0u); ++k) {
                add_hit((int)hu[0]);
#pragma unroll
                for (int j = 0; j < 7; ++j) hu[j] = hu[j + 1];
                hu[7] = ~0u;
            }
        }
#undef DEFTET_CSWAP
    }
    if (parked) {                                                  // the rescan's sums, added after the record's hits as before
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += s_park[k][tid];
        gp += s_park[12][tid];
    }
    if (nU > 0) __syncthreads();                                   // (workgroup-uniform) s_park is done with before s_rows overwrites it
    if (live) {
        const bool deferred = grad_pred && t == 0;                 // tet 0 of the shape: written by the ticket winner below
        if (grad_pred && !deferred) stream_store(grad_pred + (size_t)b * T + t, accumulate ? grad_pred[(size_t)b * T + t] + gp : gp);
    }
    {   // the 48-byte rows of a wave go through LDS so that every store instruction writes 1 KB of consecutive addresses
        // (a lane storing its own row wrote 16 bytes every 48: three partial-line streaming stores per line)
        const int w = tid >> 6;
        const int t0 = t - lane;                                    // the wave's first tet
        // SPARSE: the rows of the tets that accepted a query (the only ones that can be non-zero), in lane order
        const bool has = live && ((h.x >= 0 && h.y != kHitOverflow) || parked);
        const unsigned long long hm = SPARSE ? __ballot(has) : 0ull;
        const int slot = SPARSE ? __popcll(hm & ((1ull << lane) - 1ull)) : lane;
        if (!SPARSE || has) {
            s_rows[w][slot * 3] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            s_rows[w][slot * 3 + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            s_rows[w][slot * 3 + 2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
        }
        if (SPARSE && lane == 0 && t0 < T) rowMask[(size_t)b * ((T + 63) >> 6) + (t0 >> 6)] = hm;
        __syncthreads();
        float4 *dst = reinterpret_cast<float4 *>(grad_tet + ((size_t)b * T + t0) * 12);
        const int nRow = (SPARSE ? __popcll(hm) : min(64, T - t0)) * 3;   // float4 pieces of the wave's live rows (<= 0: none)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = k * 64 + lane;
            if (j < nRow) {
                float4 o = s_rows[w][j];
                if (accumulate && !SPARSE) { const float4 p = dst[j]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }   // (SPARSE: the rows are a workspace; `accumulate` is about grad_pred)
                stream_store(dst + j, o);
            }
        }
    }
    if (!side) return;
    // miss-sum reduction across the side workgroups: partials and ticket live at the memory side (RMW atomics), so no
    // cache write-back is needed — a release fence here would flush the whole L2 behind ~100 MB of gradient stores
    float *mp = missPart + (size_t)b * kMissStride;
    int *ticket = hitWords + pad + b;
    if (tid == 0) {
        mem_write_f32(mp + bx, missPartial);
        if (bx == 0) mem_write_f32(mp + kMissParts, gp);   // thread 0 of workgroup 0 is the lane of tet 0
        __builtin_amdgcn_s_waitcnt(0);                             // both exchanges have returned: they are at the memory side
        s_last = atomicAdd(ticket, 1) == nMissParts - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid <= kMissParts) s_vals[tid] = (tid < nMissParts || tid == kMissParts) ? mem_read_f32(mp + tid) : 0.f;
    __syncthreads();
    if (tid == 0) {
        float tot = 0.f;
        for (int k = 0; k < kMissParts; ++k) tot += s_vals[k];     // index order: deterministic
        tot += s_vals[kMissParts];
        float *dst = grad_pred + (size_t)b * T;
        *dst = accumulate ? *dst + tot : tot;
        atomicExch(ticket, 0);                                     // ready for the next backward on this hit buffer
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
// Grid tunables.  Environment overrides (experiments only) are read ONCE per process, so a prepare and the scan that
// consumes it always agree on the layout.
struct Tunables {
    double gdiv = 6.0;      // tets per coarse cell
    double qdiv = 2.0;      // queries per coarse cell
    double yzfine = 2.0;    // y/z cells per coarse cell edge
    double xfine = 2.0;     // x cells per y/z cell edge (the run direction)
    Tunables()
    {
        auto get = [](const char *name, double lo, double hi, double &dst) {
            if (const char *e = getenv(name)) {
                const double v = atof(e);
                if (v >= lo && v <= hi) dst = v;
            }
        };
        get("DEFTET_PIT_GDIV", 0.25, 4096.0, gdiv);
        get("DEFTET_PIT_QDIV", 0.03, 4096.0, qdiv);
        get("DEFTET_PIT_YZFINE", 0.25, 8.0, yzfine);
        get("DEFTET_PIT_XFINE", 0.25, 16.0, xfine);
    }
};
static const Tunables &tunables()
{
    static const Tunables t;
    return t;
}

// DEFTET_PIT_AUTO: the wave-staged traversal pays for its staging (union box, row bounds, copy) with a walk that costs a
// third per candidate; with about one query per tet or more (BASELINE configs[1]) the candidates dominate either way and
// the round-3 walk, which takes three of them per wave-iteration, is faster (measured: 35 vs 53 us at configs[1], 83 vs 80
// at configs[2], 204 vs 185 at configs[3]).
static int resolve_auto(int algo, int T, int Q)
{
    if (algo != DEFTET_PIT_AUTO) return algo;
    // (DEFTET_PIT_PAIR, two tets per lane, executes 4 % fewer instructions but lives at five waves per SIMD with waves that
    // take 1.5x as long: 79-89 against 63-67 us at configs[2], 173-187 against 141-153 at configs[3]; round 5)
    return (double)Q <= 0.6 * (double)T ? DEFTET_PIT_WAVE : DEFTET_PIT_SLAB;
}

static void pick_grid(int T, int Q, int &G, int &Gx)
{
    const Tunables &tn = tunables();
    const double a = T / tn.gdiv, bq = (Q > 0 ? Q : 1) / tn.qdiv;
    const double m = a < bq ? a : bq;
    long long g = llround(tn.yzfine * cbrt(m < 1 ? 1 : m));
    if (g < 1) g = 1;
    if (g > kMaxG) g = kMaxG;
    long long gx = llround(tn.xfine * (double)g);
    if (gx < 1) gx = 1;
    while (gx > 1 && (g / 4 + 2) * (gx | 1) > kMaxBins) --gx;       // a slab quarter's (cy, cx) counters must fit k_slab_sort's LDS
    G = (int)g;
    Gx = (int)gx;
}

struct Layout {
    int G, Gx, nRowBlk, chunkQ;
    long long cellStride;   // padded table words per shape (>= G * (Gx + 1) * (G + 3))
    size_t bytes;
    float *bboxPart, *chunkBox;
    int *irrCnt;
    int nblkPad;
    int *counters, *table, *pre, *result, *irregT, *irregQ;
    float4 *localQ, *sortedQ;
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
        pick_grid(T, Q, L.G, L.Gx);
        const long long n = (long long)L.G * (L.Gx + 1) * table_pitch(L.G);
        L.cellStride = (n + 63) / 64 * 64;
        L.nRowBlk = (Q + kRowTile - 1) / kRowTile;
        if (L.nRowBlk > kMaxRowBlocks) L.nRowBlk = kMaxRowBlocks;
        if (L.nRowBlk < 1) L.nRowBlk = 1;
        L.chunkQ = ((Q + L.nRowBlk - 1) / L.nRowBlk + 255) / 256 * 256;
        L.bboxPart = A.take<float>((size_t)B * kBoxBlocks * 6);
        L.chunkBox = A.take<float>((size_t)B * kMaxRowBlocks * 6);  // k_slab_local<true>: the chunks' boxes, the chunks' unbinned counts
        L.irrCnt = A.take<int>((size_t)B * kMaxRowBlocks * 2);      // unbinned queries per chunk, then those of them that are regular
        L.counters = A.take<int>((size_t)B * (8 + kOvfCap));  // 4 counters + 4 statistics words per shape, then the overflowed-tet lists
        L.gparam = A.take<float>((size_t)B * kGridWords);
        L.table = A.take<int>((size_t)B * L.cellStride);
        L.nblkPad = (L.nRowBlk + 15) / 16 * 16;
        L.pre = A.take<int>((size_t)B * (L.G * kSub + 1) * L.nblkPad);
        L.localQ = A.take<float4>((size_t)B * Q);
        L.sortedQ = A.take<float4>((size_t)B * Q);
        L.irregT = A.take<int>((size_t)B * T);
        L.irregQ = A.take<int>((size_t)B * Q);
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
    return hit_spill_off(B, T, Q) + (size_t)B * T * 4;
}

extern "C" int deftet_point_in_tet_resolve_algo(int algo, int T, int Q) { return resolve_auto(algo, T, Q); }

extern "C" int deftet_point_in_tet_grid_dims(int T, int Q, int *G_yz, int *G_x)
{
    DEFTET_CHECK_ARG(T >= 0 && Q >= 0 && G_yz && G_x, "bad argument");
    pick_grid(T, Q, *G_yz, *G_x);
    return DEFTET_OK;
}

static int pit_check(const float *tet, const float *pts, const float *cond, const float *bary, const float *pred, const float *occ,
                     const int32_t *hit_buf, int B, int T, int Q, int algo, const void *workspace)
{
    DEFTET_CHECK_ARG(!hit_buf || (((uintptr_t)hit_buf & 15) == 0 && algo != DEFTET_PIT_BRUTE), "hit_buf must be 16-byte aligned and needs a binned algo");
    DEFTET_CHECK_ARG((pred == nullptr) == (occ == nullptr), "pred and occ must be given together");
    DEFTET_CHECK_ARG(!occ || T > 0, "paste_occ needs at least one tet");
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0, "negative size (B=%d T=%d Q=%d)", B, T, Q);
    DEFTET_CHECK_ARG(algo == DEFTET_PIT_AUTO || algo == DEFTET_PIT_BRUTE || algo == DEFTET_PIT_EXACT || algo == DEFTET_PIT_SLAB || algo == DEFTET_PIT_WAVE || algo == DEFTET_PIT_PAIR, "unknown algo %d", algo);
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
static int pit_prepare(const Layout &L, const float *pts, int B, int Q, hipStream_t st, const float *boxIn = nullptr, float *boxOut = nullptr,
                       int32_t *missOut = nullptr)
{
    const dim3 blk(256);
    if (boxIn) {                                                         // two launches: the grid spans the box handed in
        DEFTET_LAUNCH(k_slab_local<true>, dim3(L.nRowBlk, B), blk, st, pts, Q, L.bboxPart, L.gparam, L.G, L.Gx, L.nRowBlk, L.nblkPad, L.chunkQ,
                      L.localQ, L.pre, L.counters, L.irregQ, boxIn, L.chunkBox, L.irrCnt, L.result);
    } else {
        DEFTET_LAUNCH(k_query_bbox, dim3(kBoxBlocks, B), blk, st, pts, Q, L.bboxPart, L.counters, L.result, B, (long long)B * Q);
        DEFTET_LAUNCH(k_slab_local<false>, dim3(L.nRowBlk, B), blk, st, pts, Q, L.bboxPart, L.gparam, L.G, L.Gx, L.nRowBlk, L.nblkPad, L.chunkQ,
                      L.localQ, L.pre, L.counters, L.irregQ, (const float *)nullptr, (float *)nullptr, (int *)nullptr, (int *)nullptr);
    }
    const size_t shm = align_up((size_t)((L.G + kParts - 1) / kParts + 1) * (L.Gx | 1) * sizeof(int), 16);   // rows of a y-quarter
    DEFTET_LAUNCH_SHM(k_slab_sort, dim3(L.G * kParts, B), dim3(kSortThreads), shm, st, L.localQ, Q, L.gparam, L.G, L.Gx, L.pre, L.nRowBlk, L.nblkPad,
                      L.chunkQ, L.cellStride, L.table, L.sortedQ, (size_t)Q * 16 <= ((size_t)4 << 20),
                      (const float *)(boxIn ? L.chunkBox : L.bboxPart), boxIn ? L.nRowBlk : kBoxBlocks, boxIn ? kMaxRowBlocks : kBoxBlocks, boxOut,
                      boxIn ? 1 : 0, (const int *)L.irrCnt, L.irregQ, L.counters, (int *)(boxIn ? missOut : nullptr));
    return DEFTET_OK;
}

// tet side: traversal + finalize; consumes the prepared state (result sentinels, counters)
static int pit_scan(const Layout &L, const float *tet, const float *pts, float *cond, float *bary, const float *pred, float *occ,
                    int32_t *hit_buf, int B, int T, int Q, int algo, hipStream_t st, const int32_t *order = nullptr)
{
    const dim3 blk(256);
    const dim3 gf((Q + 256 * PIT_FIN_QPT - 1) / (256 * PIT_FIN_QPT), B), gt((((T + 255) / 256 + 7) / 8) * 8, B);   // gt: multiple of 8 for the XCD mapping
    int *ucount = hit_buf ? hit_buf + hit_cnt_off(B, T) : nullptr;
    if (T > 0) {
        if (algo == DEFTET_PIT_EXACT) {
            DEFTET_LAUNCH(k_tet_scan, gt, blk, st, tet, T, Q, L.gparam, L.G, L.Gx, L.table, L.cellStride, L.sortedQ, L.result,
                          L.counters, L.irregT, (int2 *)hit_buf, pts, L.irregQ, ucount, hit_pad(B), hit_buf ? (int4 *)(hit_buf + hit_spill_off(B, T, Q)) : (int4 *)nullptr);
        } else {
            int4 *spill = hit_buf ? (int4 *)(hit_buf + hit_spill_off(B, T, Q)) : (int4 *)nullptr;
            const int kern = resolve_auto(algo, T, Q);
            const bool slab = kern == DEFTET_PIT_SLAB;
#if PIT_PROBE_SCATTER && PIT_PROBE_PREFILL
            // probe: the lines the per-hit stores will hit are written in full first (do they then sit in the Infinity Cache?)
            if (spill) DEFTET_HIP(hipMemsetAsync(spill, 0, (size_t)B * Q * 24, st));
#endif
            const dim3 bw(kWvThreads);                                                           // the wave-staged kernels' workgroup
            const dim3 gw((((T + kWvThreads - 1) / kWvThreads + 7) / 8) * 8, B);
            const dim3 gp(((((T + 1) / 2 + kWvThreads - 1) / kWvThreads + 7) / 8) * 8, B);       // two tets per lane
#define PIT_SCAN_ARGS tet, T, Q, L.gparam, L.G, L.Gx, L.table, L.cellStride, L.sortedQ, L.result, L.counters, L.irregT, (int2 *)hit_buf, pts, \
                      L.irregQ, ucount, hit_pad(B), spill, (const int *)order
            if (slab && order) DEFTET_LAUNCH(k_tet_scan_slab<true>, gt, blk, st, PIT_SCAN_ARGS);
            else if (slab) DEFTET_LAUNCH(k_tet_scan_slab<false>, gt, blk, st, PIT_SCAN_ARGS);
            else if (kern == DEFTET_PIT_PAIR && order) DEFTET_LAUNCH(k_tet_scan_pair<true>, gp, bw, st, PIT_SCAN_ARGS);
            else if (kern == DEFTET_PIT_PAIR) DEFTET_LAUNCH(k_tet_scan_pair<false>, gp, bw, st, PIT_SCAN_ARGS);
            else if (order) DEFTET_LAUNCH(k_tet_scan_wave<true>, gw, bw, st, PIT_SCAN_ARGS);
            else DEFTET_LAUNCH(k_tet_scan_wave<false>, gw, bw, st, PIT_SCAN_ARGS);
#undef PIT_SCAN_ARGS
        }
    } else if (ucount) {
        DEFTET_HIP(hipMemsetAsync(ucount, 0, (size_t)3 * hit_pad(B) * 4, st));
    }
    DEFTET_LAUNCH(k_finalize<PIT_FIN_QPT>, gf, blk, st, tet, pts, T, Q, L.result, cond, bary, pred, occ, (const int2 *)hit_buf, ucount,
                  hit_buf ? hit_buf + hit_list_off(B, T) : nullptr, L.counters, L.irregT, hit_pad(B), pin_shapes(Q), (const float *)L.gparam);
    return DEFTET_OK;
}

static int pit_forward(const float *tet, const float *pts, float *cond, float *bary, const float *pred, float *occ, int32_t *hit_buf,
                       int B, int T, int Q, int algo, void *workspace, size_t workspace_bytes, void *stream_, const int32_t *order,
                       const float *boxIn = nullptr, float *boxOut = nullptr, int32_t *missOut = nullptr)
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
        DEFTET_LAUNCH(k_finalize<PIT_FIN_QPT>, dim3((Q + 256 * PIT_FIN_QPT - 1) / (256 * PIT_FIN_QPT), B), blk, st, tet, pts, T, Q, L.result, cond, bary, pred, occ, (const int2 *)nullptr, (int *)nullptr,
                      (int *)nullptr, (const int *)nullptr, L.irregT, 0, pin_shapes(Q), (const float *)nullptr);
        return DEFTET_OK;
    }
    rc = pit_prepare(L, pts, B, Q, st, boxIn, boxOut, missOut);
    if (rc != DEFTET_OK) return rc;
    return pit_scan(L, tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, st, order);
}

extern "C" int deftet_point_in_tet_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                       float *occ, int32_t *hit_buf, int B, int T, int Q, int algo, void *workspace,
                                       size_t workspace_bytes, void *stream_)
{
    return pit_forward(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace, workspace_bytes, stream_, nullptr);
}

// The same with a traversal order (int32 [T] on the device, a permutation of [0, T) — deftet_tet_spatial_order_f32 — shared by
// the shapes of the batch; NULL = the caller's own order).  Outputs are identical with and without it: the filter kernels
// (DEFTET_PIT_AUTO / _SLAB / _WAVE / _PAIR) walk the tets in that order and publish the original indices; the other ids ignore it.
// ... and with a box for the query grid (query_box_in, f32 [B,6] = lo xyz, hi xyz on the device, or NULL): the grid spans that box
// (enlarged by 1/32 per side) instead of the measured box of this call's queries, which saves the measuring launch.  The box is
// a hint: queries outside it are answered exactly by the side path that handles NaN / Inf / huge queries, at brute-force cost
// each — hand in the sampler's box, or the box an earlier call with the same query distribution measured: query_box_out
// (f32 [B,6] or NULL, must not alias query_box_in) receives the box of THIS call's regular queries (lo > hi when there is none).
// query_box_misses (int32 [B] or NULL; any memory the device can write, e.g. host-mapped so that the caller can poll it
// without synchronising) receives, when query_box_in is given, how many regular queries of each shape fell outside it: the
// feedback a caller that reuses boxes needs to notice that its query distribution moved (hip_ops.point_in_tet's "track"
// mode goes back to measuring when it sees misses).  The brute-force algorithm ignores all three.
extern "C" int deftet_point_in_tet_ex_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                          float *occ, int32_t *hit_buf, int B, int T, int Q, int algo, const int32_t *tet_order,
                                          const float *query_box_in, float *query_box_out, int32_t *query_box_misses,
                                          void *workspace, size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(!query_box_in || query_box_in != query_box_out, "query_box_in and query_box_out must not alias");
    return pit_forward(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace, workspace_bytes, stream_, tet_order,
                       query_box_in, query_box_out, query_box_misses);
}

// The same operator in two calls: the QUERY side (bounding box + counting sort: depends on pts and on
// the sizes only) can be enqueued ahead of time — e.g. on another stream while the previous step's
// backward is still running — and the TET side consumes it.  One prepare feeds exactly one scan
// (the scan uses up the result sentinels and counters the prepare resets); both must see the same
// pts, sizes, algo and workspace.
static int pit_prepare_entry(const float *pts, int B, int T, int Q, int algo, void *workspace, size_t workspace_bytes, void *stream_,
                             const float *boxIn, float *boxOut, int32_t *missOut);
extern "C" int deftet_point_in_tet_prepare_f32(const float *pts, int B, int T, int Q, int algo, void *workspace,
                                               size_t workspace_bytes, void *stream_)
{
    return pit_prepare_entry(pts, B, T, Q, algo, workspace, workspace_bytes, stream_, nullptr, nullptr, nullptr);
}
extern "C" int deftet_point_in_tet_prepare_ex_f32(const float *pts, int B, int T, int Q, int algo, const float *query_box_in,
                                                  float *query_box_out, int32_t *query_box_misses, void *workspace,
                                                  size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(!query_box_in || query_box_in != query_box_out, "query_box_in and query_box_out must not alias");
    return pit_prepare_entry(pts, B, T, Q, algo, workspace, workspace_bytes, stream_, query_box_in, query_box_out, query_box_misses);
}
static int pit_prepare_entry(const float *pts, int B, int T, int Q, int algo, void *workspace, size_t workspace_bytes, void *stream_,
                             const float *boxIn, float *boxOut, int32_t *missOut)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Q >= 0 && B <= 65535, "bad size (B=%d T=%d Q=%d)", B, T, Q);
    DEFTET_CHECK_ARG(algo == DEFTET_PIT_AUTO || algo == DEFTET_PIT_EXACT || algo == DEFTET_PIT_SLAB || algo == DEFTET_PIT_WAVE || algo == DEFTET_PIT_PAIR, "prepare needs a binned algo (got %d)", algo);
    if (Q >= (1 << 27)) return set_error(DEFTET_ELIMIT, "n_query=%d exceeds 2^27", Q);
    if (B == 0 || Q == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts, "null pts pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    Layout L = make_layout(B, T, Q, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    return pit_prepare(L, pts, B, Q, as_stream(stream_), boxIn, boxOut, missOut);
}

static int pit_scan_entry(const float *tet, const float *pts, float *cond, float *bary, const float *pred, float *occ, int32_t *hit_buf,
                          int B, int T, int Q, int algo, void *workspace, size_t workspace_bytes, void *stream_, const int32_t *order)
{
    DEFTET_CHECK_ARG(algo != DEFTET_PIT_BRUTE, "scan needs a binned algo");
    int rc = pit_check(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace);
    if (rc != DEFTET_OK || B == 0 || Q == 0) return rc;
    Layout L = make_layout(B, T, Q, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    return pit_scan(L, tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, as_stream(stream_), order);
}

extern "C" int deftet_point_in_tet_scan_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                            float *occ, int32_t *hit_buf, int B, int T, int Q, int algo, void *workspace,
                                            size_t workspace_bytes, void *stream_)
{
    return pit_scan_entry(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int deftet_point_in_tet_scan_ex_f32(const float *tet, const float *pts, float *cond, float *bary, const float *pred,
                                                    float *occ, int32_t *hit_buf, int B, int T, int Q, int algo,
                                                    const int32_t *tet_order, void *workspace, size_t workspace_bytes, void *stream_)
{
    return pit_scan_entry(tet, pts, cond, bary, pred, occ, hit_buf, B, T, Q, algo, workspace, workspace_bytes, stream_, tet_order);
}

// Diagnostics: copies the 8 int32 words per shape that the last forward on this workspace left behind —
// [0] irregular tets, [1] irregular queries, [2] hit-record overflow flag, [3] unused,
// [4] unused, [5] tets re-scanned exactly, [6] overflowed tets, [7] unused — to host memory.
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
// probe builds: what the runtime says about the residency of the traversal kernels (workgroups of 256 threads per compute unit)
extern "C" int deftet_debug_occupancy(int *out8)
{
    int n = 0;
    DEFTET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)deftet::pit::k_tet_scan_wave<false>, 256, 0)); out8[0] = n;
    DEFTET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)deftet::pit::k_tet_scan_pair<false>, 256, 0)); out8[1] = n;
    DEFTET_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)deftet::pit::k_tet_scan_slab<false>, 256, 0)); out8[2] = n;
    hipFuncAttributes a;
    DEFTET_HIP(hipFuncGetAttributes(&a, (const void *)deftet::pit::k_tet_scan_wave<false>));
    out8[3] = (int)a.sharedSizeBytes; out8[4] = a.numRegs; out8[5] = (int)a.maxDynamicSharedSizeBytes;
    hipDeviceProp_t pr;
    DEFTET_HIP(hipGetDeviceProperties(&pr, 0));
    out8[6] = (int)pr.sharedMemPerMultiprocessor; out8[7] = (int)pr.sharedMemPerBlock;
    return DEFTET_OK;
}
extern "C" int deftet_debug_span_read(unsigned long long *out, int n_waves)     // out[2 * n_waves]: start, end per wave id
{
    DEFTET_CHECK_ARG(out && n_waves > 0 && n_waves <= deftet::pit::kPhaseWaves, "bad argument");
    DEFTET_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(deftet::pit::g_span), (size_t)n_waves * 16));
    return DEFTET_OK;
}
extern "C" int deftet_debug_reason_read(unsigned long long *out8, int reset)
{
    DEFTET_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(deftet::pit::g_reason), 64));
    if (reset) {
        unsigned long long z[8] = {0};
        DEFTET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(deftet::pit::g_reason), z, 64));
    }
    return DEFTET_OK;
}
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

constexpr long long kDenseQueriesPerTet = 2;
extern "C" size_t deftet_point_in_tet_bwd_workspace_bytes(int B, int T, int Q)
{
    if (B <= 0 || T < 0 || Q < 0) return 0;
    const size_t lists = align_up((size_t)B * T * 4, 256) + align_up((size_t)B * Q * 4, 256) + align_up((size_t)B * 4, 256);
    const size_t miss = align_up((size_t)B * kMissStride * 4, 256);   // hit-record path: partial sums of the miss gradient
    return lists > miss ? lists : miss;
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
    // More than kDenseQueriesPerTet queries per tet: more and more tets accept more queries than a record and its spill record
    // hold (six to eight), every one of them scans the shape's list of unrecorded hits, and that is quadratic — 111 ms against
    // 0.055 ms for the per-tet lists at 100,000 queries on 6,000 tets; the crossover is measured (tools/probes/
    // dense_queries_probe.py --sweep, 8 shapes x 48,000 tets: records 0.024 / 0.028 / 0.035 / 0.058 / 0.189 / 0.501 ms against
    // lists 0.036 / 0.052 / 0.069 / 0.090 / 0.107 / 0.120 ms at 0.5 / 1.0 / 1.6 / 2.1 / 2.6 / 3.0 queries per tet).  The records
    // are for the sparse case (BASELINE: 0.4 to 1 query per tet); with a workspace the dense case takes the lists.
    const bool dense = (long long)Q > kDenseQueriesPerTet * (long long)T && workspace && ((uintptr_t)workspace & 255) == 0 &&
                       workspace_bytes >= deftet_point_in_tet_bwd_workspace_bytes(B, T, Q);
    if (hit_buf && !dense) {
        // fastest path: the forward's hit records (needs only kMissParts floats per shape of workspace)
        DEFTET_CHECK_ARG(((uintptr_t)hit_buf & 15) == 0, "hit_buf must be 16-byte aligned");
        float *missPart = nullptr;
        if (grad_pred) {
            DEFTET_CHECK_ARG(workspace && workspace_bytes >= (size_t)B * kMissStride * 4, "workspace needed for the miss sums (80 floats per shape)");
            missPart = static_cast<float *>(workspace);
        }
        const int tblocks = (T + 255) / 256, nMissParts = tblocks < kMissParts ? tblocks : kMissParts;
        int32_t *words = const_cast<int32_t *>(hit_buf) + hit_cnt_off(B, T);    // counters / ticket / flag: the buffer is this library's own
        DEFTET_LAUNCH(k_bary_bwd_hits<false>, dim3(tblocks, B), dim3(256), st, tet, pts, cond, grad_w, (const int2 *)hit_buf, T, Q,
                      grad_tet, grad_pts, accumulate, grad_occ, grad_pred, missPart, nMissParts, words,
                      (const int *)(hit_buf + hit_list_off(B, T)), hit_pad(B), (const int4 *)(hit_buf + hit_spill_off(B, T, Q)), pin_shapes(Q),
                      (unsigned long long *)nullptr);
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

// A1b backward fused with the backward of the vertex -> tet gather (N2): dL/dpos [B,V,3] without the dense dL/dtet.
// The reference's dataflow is vertice_pos -> torch.gather -> tet_bxfx4x3 -> operators (layers/DefTet/deftet.py:65-68), so the
// gradient of a caller that owns both the gather and the query ends on the vertices; deftet_point_in_tet_bwd_f32 followed by
// deftet_tet_gather_bwd_f32 writes 48 bytes per tet (71 % of them zeros at 0.34 hits per tet) and reads them back.  Here the
// per-tet pass keeps only the rows of tets that accepted a query (k_bary_bwd_hits<true>: compacted per wave + one mask word
// per 64 tets) and the per-vertex pass (vertex_ops.hip, k_gather_bwd<true>) adds them up in the incidence CSR's order: the
// same additions in the same order as the two-call form, bit for bit, no floating-point atomics.
extern "C" size_t deftet_point_in_tet_bwd_to_vertices_workspace_bytes(int B, int T, int Q)
{
    if (B <= 0 || T < 0 || Q < 0) return 0;
    return align_up((size_t)B * T * 48, 256) + align_up((size_t)B * ((T + 63) / 64) * 8, 256) +
           align_up(deftet_point_in_tet_bwd_workspace_bytes(B, T, Q), 256);
}

extern "C" int deftet_point_in_tet_bwd_to_vertices_f32(const float *tet, const float *pts, const float *cond, const float *grad_w,
                                                       const float *grad_occ, const int32_t *hit_buf, const int32_t *csr_offsets,
                                                       const int32_t *csr_slots, int idx_batch, float *grad_pos, float *grad_pts,
                                                       float *grad_pred, int B, int V, int T, int Q, int accumulate, void *workspace,
                                                       size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && V >= 0 && T >= 0 && Q >= 0, "negative size");
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds 65535", B);
    DEFTET_CHECK_ARG((grad_occ == nullptr) == (grad_pred == nullptr), "grad_occ and grad_pred must be given together");
    DEFTET_CHECK_ARG(idx_batch == 1 || idx_batch == B, "CSR batch must be 1 or n_batch (got %d)", idx_batch);
    hipStream_t st = as_stream(stream_);
    if (B == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(V == 0 || grad_pos, "null grad_pos");
    if (T == 0 || Q == 0 || V == 0) {                                 // no tet or no query: zero gradients
        if (grad_pts && Q > 0) DEFTET_HIP(hipMemsetAsync(grad_pts, 0, (size_t)B * Q * 12, st));
        if (!accumulate) {
            if (V > 0) DEFTET_HIP(hipMemsetAsync(grad_pos, 0, (size_t)B * V * 12, st));
            if (grad_pred && T > 0) DEFTET_HIP(hipMemsetAsync(grad_pred, 0, (size_t)B * T * 4, st));
        }
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(tet && pts && cond && grad_w && csr_offsets && csr_slots, "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)tet & 15) == 0 && ((uintptr_t)grad_w & 15) == 0, "tet/grad_w must be 16-byte aligned");
    const size_t need = deftet_point_in_tet_bwd_to_vertices_workspace_bytes(B, T, Q);
    DEFTET_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0,
                     "workspace null, misaligned or too small (%zu < %zu)", workspace_bytes, need);
    Arena A(workspace, workspace_bytes);
    float *rows = A.take<float>((size_t)B * T * 12);
    unsigned long long *rowMask = A.take<unsigned long long>((size_t)B * ((T + 63) / 64));
    const size_t innerBytes = deftet_point_in_tet_bwd_workspace_bytes(B, T, Q);
    void *inner = A.take<char>(innerBytes);
    const bool dense = (long long)Q > kDenseQueriesPerTet * (long long)T;
    if (hit_buf && !dense) {
        DEFTET_CHECK_ARG(((uintptr_t)hit_buf & 15) == 0, "hit_buf must be 16-byte aligned");
        if (grad_pts) DEFTET_HIP(hipMemsetAsync(grad_pts, 0, (size_t)B * Q * 12, st));
        const int tblocks = (T + 255) / 256, nMissParts = tblocks < kMissParts ? tblocks : kMissParts;
        int32_t *words = const_cast<int32_t *>(hit_buf) + hit_cnt_off(B, T);
        DEFTET_LAUNCH(k_bary_bwd_hits<true>, dim3(tblocks, B), dim3(256), st, tet, pts, cond, grad_w, (const int2 *)hit_buf, T, Q,
                      rows, grad_pts, accumulate, grad_occ, grad_pred, static_cast<float *>(inner), nMissParts, words,
                      (const int *)(hit_buf + hit_list_off(B, T)), hit_pad(B), (const int4 *)(hit_buf + hit_spill_off(B, T, Q)), pin_shapes(Q),
                      rowMask);
        return vtx::gather_bwd_rows(rows, rowMask, csr_offsets, csr_slots, grad_pos, B, V, T, idx_batch, accumulate, st);
    }
    // no records (the forward was asked for none), or the dense case where the records are not the fast path: the per-tet lists
    // into dense rows, then the dense gather
    if (accumulate) DEFTET_HIP(hipMemsetAsync(rows, 0, (size_t)B * T * 48, st));     // (the inner call's `accumulate` covers grad_pred too)
    const int rc = deftet_point_in_tet_bwd_f32(tet, pts, cond, grad_w, rows, grad_pts, grad_occ, grad_pred, hit_buf, B, T, Q, accumulate, inner,
                                               innerBytes, stream_);
    if (rc != DEFTET_OK) return rc;
    return vtx::gather_bwd_rows(rows, nullptr, csr_offsets, csr_slots, grad_pos, B, V, T, idx_batch, accumulate, st);
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
