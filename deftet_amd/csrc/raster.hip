// raster.hip — A12: differentiable tet-face rasterizer with the contract of
// kaolin.render.mesh.deftet_sparse_render as the reference calls it
// (diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:97-100).
//
// PARITY UNPINNED: Kaolin is a third-party, un-vendored, un-pinned dependency of the reference
// (README.md:30); its source is not part of the reference tree.  The arithmetic below is this
// build's statement of the documented contract (same as oracle/deftet_oracle_render.c):
//     m = bx-ax; pp = by-ay; n = cx-ax; q = cy-ay; s = px-ax; t = py-ay
//     k1 = s*q - n*t;  k2 = m*t - s*pp;  k3 = m*q - n*pp
//     w1 = k1/(k3+eps); w2 = k2/(k3+eps); w0 = 1 - w1 - w2;   covered iff w0,w1,w2 >= 0
//     z = (w0*az + w1*bz) + w2*cz;                            kept iff zmin <= z <= zmax
//   per pixel at most `knum` kept faces are recorded and ordered by z descending (ties: ascending face
//   index); which ones when more are kept is the saturation POLICY: the `knum` that come first in that
//   order (NEAREST, the default — an insertion-sorted list of bounded length) or the first `knum` in
//   ascending face index (FIRST, rounds 1-2); features = (w0*f0 + w1*f1) + w2*f2.
//
// MI355X design (the brute-force formulation is pixels x faces = 1.4e11 tests at 512x512 over a
// res-70 grid): pixels define a uniform 2-D tile grid; faces are binned into the tiles their
// (slightly enlarged) image-space box overlaps by one stable radix sort (lists come out ascending
// in face index), pixels are sorted by tile, and one wave rasterises 64 pixels of one tile against
// that tile's list with cooperative loads and lane broadcasts (merged with a short list of "wide"
// faces — degenerate, non-finite or spanning more than kMaxTiles tiles — which every pixel tests).
// The backward is atomic-free: hits are threaded into per-face linked lists and one lane per face
// accumulates its gradients in registers.
#pragma clang fp contract(off)
#include <cstring>
#include <stdlib.h>

#include "common.hpp"

#include "prims.hpp"

namespace deftet {
namespace rast {

constexpr int kMaxTiles = 16;          // faces overlapping more tiles go to the wide list
constexpr int kBoxBlocks = 512;         // workgroups of the two statistics kernels (64 of them were one dependent load per trip, 32 trips: 19 us for 19 MB)
constexpr float kBig = 1048576.0f;     // 2^20
constexpr float kTau = 1.0f / 128.0f;
constexpr float kMargin = 1.0f / 64.0f;

struct Grid2 { float ox, oy, ix, iy, lox, loy, hix, hiy; int gx, gy; };
constexpr int kG2Max = 90;             // tiles per axis (upper bound; the actual count is chosen on the device).  90^2 < 2^13: the tile keys of
                                       // the two sorts below fit 13 (+ 2) bits = two radix passes each (512 per axis, rounds 1-5: 19 / 21 bits = three),
                                       // and the per-tile tables and the chunk launch shrink from 262 k to 8 k entries.  BASELINE configs[4] picks 32 x 32.

__device__ __forceinline__ int cell_of(float x, float o, float inv, int G)
{
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.f), (float)(G - 1));
    return (int)f;
}

__global__ __launch_bounds__(256) void k_pix_bbox(const float *__restrict__ pix, int P, float *part)
{
    __shared__ float sh[4][4];
    float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const float x = pix[p * 2], y = pix[p * 2 + 1];
        if (fabsf(x) <= kBig && fabsf(y) <= kBig) {
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = lo[0]; sh[w][1] = lo[1]; sh[w][2] = hi[0]; sh[w][3] = hi[1]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = k < 2 ? fminf(v, sh[i][k]) : fmaxf(v, sh[i][k]);
        part[blockIdx.x * 4 + k] = v;
    }
}

// mean image-space extent of the finite faces and the largest finite |corner depth| (per-block partials: sum of w, count,
// max |z|; k_pix_grid finishes them)
__global__ __launch_bounds__(256) void k_face_stats(const float *__restrict__ xy, const float *__restrict__ fz, int F, float *part)
{
    __shared__ float sh[4][3];
    float sw = 0.f, cnt = 0.f, zm = 0.f;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
        const float m = fmaxf(fabsf(fz[f * 3]), fmaxf(fabsf(fz[f * 3 + 1]), fabsf(fz[f * 3 + 2])));
        if (m < INFINITY) zm = fmaxf(zm, m);                           // NaN / Inf depths do not scale the margin
        const float2 a = reinterpret_cast<const float2 *>(xy)[f * 3], b = reinterpret_cast<const float2 *>(xy)[f * 3 + 1],
                     c = reinterpret_cast<const float2 *>(xy)[f * 3 + 2];
        const float w = fmaxf(fmaxf(a.x, fmaxf(b.x, c.x)) - fminf(a.x, fminf(b.x, c.x)),
                              fmaxf(a.y, fmaxf(b.y, c.y)) - fminf(a.y, fminf(b.y, c.y)));
        if (w <= 2.f * kBig && w > 0.f) { sw += w; cnt += 1.f; }      // NaN fails
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sw += __shfl_xor(sw, off); cnt += __shfl_xor(cnt, off);
        zm = fmaxf(zm, __shfl_xor(zm, off));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wv][0] = sw; sh[wv][1] = cnt; sh[wv][2] = zm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 3] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        part[blockIdx.x * 3 + 1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
        part[blockIdx.x * 3 + 2] = fmaxf(fmaxf(sh[0][2], sh[1][2]), fmaxf(sh[2][2], sh[3][2]));
    }
}

// pixel box + mean face size -> tile grid: tiles are about one mean face extent wide, so a
// typical face overlaps 2x2..3x3 tiles; never more than kG2Max tiles per axis
__global__ __launch_bounds__(64) void k_pix_grid(const float *__restrict__ part, const float *__restrict__ fpart, Grid2 *g, unsigned *zAbsMax)
{
    const int lane = threadIdx.x;
    float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    float sw = 0.f, cnt = 0.f, zm = 0.f;
    for (int i = lane; i < kBoxBlocks; i += 64) {                   // fixed order: the same tile grid on every run
        lo[0] = fminf(lo[0], part[i * 4]); lo[1] = fminf(lo[1], part[i * 4 + 1]);
        hi[0] = fmaxf(hi[0], part[i * 4 + 2]); hi[1] = fmaxf(hi[1], part[i * 4 + 3]);
        sw += fpart[i * 3]; cnt += fpart[i * 3 + 1]; zm = fmaxf(zm, fpart[i * 3 + 2]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        sw += __shfl_xor(sw, off);
        cnt += __shfl_xor(cnt, off);
        zm = fmaxf(zm, __shfl_xor(zm, off));
    }
    if (lane == 0) {
        *zAbsMax = __float_as_uint(zm);                             // scales the rounding margin of the NEAREST walk's stop test
        Grid2 r;
        const bool okx = hi[0] >= lo[0], oky = hi[1] >= lo[1];
        r.lox = okx ? lo[0] : 0.f; r.hix = okx ? hi[0] : 0.f;
        r.loy = oky ? lo[1] : 0.f; r.hiy = oky ? hi[1] : 0.f;
        r.ox = r.lox; r.oy = r.loy;
        const float ex = r.hix - r.lox, ey = r.hiy - r.loy;
        const float meanw = cnt > 0.f ? sw / cnt : 0.f;
        float nx = meanw > 0.f ? ceilf(ex / meanw) : 1.f, ny = meanw > 0.f ? ceilf(ey / meanw) : 1.f;
        nx = fminf(fmaxf(nx, 1.f), (float)kG2Max);                  // NaN -> 1
        ny = fminf(fmaxf(ny, 1.f), (float)kG2Max);
        r.gx = (int)nx; r.gy = (int)ny;
        r.ix = ex > 1e-30f ? nx / ex : 0.f;
        r.iy = ey > 1e-30f ? ny / ey : 0.f;
        *g = r;
    }
}

// face classification + tile range.  A face is "regular" iff its six coordinates are finite and
// <= 2^20, |k3| >= 2^-7 * w^2 (w = largest box extent) and |k3| >= 2^10 * eps: then every pixel
// the fp32 test can accept lies within w/64 of the face's box (same argument as DESIGN.md A1).
// SLIVER faces (round 6).  A face whose area is small against its extent — |k3| < kTau w^2, an edge-on triangle — used to go to
// the "wide" list with the truly degenerate ones, and every wave tested every wide face against its 64 pixels: at BASELINE
// configs[4] one diagonal plane family of the res-70 grid passes within a degree of the camera, 846 of 521,850 faces are such
// slivers, and 4,096 waves x 846 faces = 3.5 M broadcast tests stood beside the 1.6 M of all tile lists together
// (profiles/r06_raster_probes.jsonl).  They are not unbounded: with u = 2^-24, W >= every edge component, K1, K2, K3 the exact
// values of the contract's expressions on the rounded edges and (s, t) = fl(p - a):
//     (s, t) = alpha (m, pp) + beta (n, q),  alpha = K1 / K3, beta = K2 / K3                          (Cramer)
//     k_i = K_i + e_i, |e_1|, |e_2| <= 3.01 u W (|s| + |t|), |e_3| <= 6.02 u W^2, den = fl(k3 + eps), |eps| <= |k3| / 1024
//     covered  =>  k1/den, k2/den >= -2^-150  and  k1/den + k2/den <= 1 + 2.1 u          (correctly rounded /, fl(fl(1 - w1) - w2) >= 0)
// and |k3| >= 2^-16 w^2 gives rho = 6.02 u W^2 / |k3| <= 0.0236, den / K3 = 1 + theta with |theta| <= 0.0252, and, with
// S = |alpha| + |beta| and |s| + |t| <= 2 W S:  S <= 1.0252 (1 + 2.2 u) + 2 (rho / (1 - rho)) S  =>  S <= 1.078,
// alpha, beta >= -0.026, alpha + beta <= 1.078.  So (alpha, beta) lies within L1 distance 0.182 of the triangle's own
// parameter simplex, i.e. p lies within 0.182 W (1 + 2u) of the face's box along each axis: the box enlarged by kSliverMargin = 1/4
// of its extent holds every pixel the contract can accept (measured on 400 random slivers x 400,000 pixels along their lines, numpy
// fp32: at most 0.002 w outside).  The enlarged box must survive its own rounding, so a sliver also needs
// w >= 2^-16 max|coordinate| (then u |coordinate| <= 2^-8 w).  Such faces stay on the wide list — their boxes span dozens of tiles
// — but the list now carries a certified box per face (regular faces that are wide by SPAN have theirs too) and a wave skips the
// faces whose box misses its pixels' box: one scalar 16-byte load and four compares instead of the whole test.
constexpr float kTauSliver = 1.0f / 65536.0f;      // 2^-16
constexpr float kSliverMargin = 0.25f;
struct FaceBox { int tx0, tx1, ty0, ty1, mode; float elx, ehx, ely, ehy; };   // mode 0: skip, 1: tiles, 2: wide; e*: certified box (+-inf: none)

__device__ __forceinline__ FaceBox face_box(const float *__restrict__ xy, int f, const Grid2 &g, float eps)
{
    const float2 a = reinterpret_cast<const float2 *>(xy)[f * 3], b = reinterpret_cast<const float2 *>(xy)[f * 3 + 1],
                 c = reinterpret_cast<const float2 *>(xy)[f * 3 + 2];
    FaceBox r{0, 0, 0, 0, 2, -INFINITY, INFINITY, -INFINITY, INFINITY};
    const bool finite = fabsf(a.x) <= kBig && fabsf(a.y) <= kBig && fabsf(b.x) <= kBig && fabsf(b.y) <= kBig &&
                        fabsf(c.x) <= kBig && fabsf(c.y) <= kBig;
    const float m = b.x - a.x, pp = b.y - a.y, n = c.x - a.x, q = c.y - a.y;
    const float k3 = m * q - n * pp;
    const float lox = fminf(a.x, fminf(b.x, c.x)), hix = fmaxf(a.x, fmaxf(b.x, c.x));
    const float loy = fminf(a.y, fminf(b.y, c.y)), hiy = fmaxf(a.y, fmaxf(b.y, c.y));
    const float w = fmaxf(hix - lox, hiy - loy);
    const bool regular = finite && fabsf(k3) >= kTau * (w * w) && fabsf(k3) >= 1024.0f * fabsf(eps) && w > 0.f;
    if (!regular) {
        const float cmax = fmaxf(fmaxf(fabsf(lox), fabsf(hix)), fmaxf(fabsf(loy), fabsf(hiy)));
        const bool sliver = finite && fabsf(k3) >= kTauSliver * (w * w) && fabsf(k3) >= 1024.0f * fabsf(eps) && w > 0.f && w >= kTauSliver * cmax;
        if (sliver) {
            const float mg = w * kSliverMargin;
            r.elx = lox - mg; r.ehx = hix + mg; r.ely = loy - mg; r.ehy = hiy + mg;
            if (r.ehx < g.lox || r.elx > g.hix || r.ehy < g.loy || r.ely > g.hiy) r.mode = 0;       // no tame pixel can be covered
        }
        return r;
    }
    const float mg = w * kMargin;
    const float elx = lox - mg, ehx = hix + mg, ely = loy - mg, ehy = hiy + mg;
    if (ehx < g.lox || elx > g.hix || ehy < g.loy || ely > g.hiy) { r.mode = 0; return r; }
    r.tx0 = cell_of(elx, g.ox, g.ix, g.gx); r.tx1 = cell_of(ehx, g.ox, g.ix, g.gx);
    r.ty0 = cell_of(ely, g.oy, g.iy, g.gy); r.ty1 = cell_of(ehy, g.oy, g.iy, g.gy);
    r.mode = ((r.tx1 - r.tx0 + 1) * (r.ty1 - r.ty0 + 1) <= kMaxTiles) ? 1 : 2;
    r.elx = elx; r.ehx = ehx; r.ely = ely; r.ehy = ehy;
    return r;
}

// Tile lists WITHOUT atomics and without a per-tile sort (round-1 history: atomic count + atomic
// fill + a bitonic sort per tile = 0.58 + 0.59 + 1.71 ms at configs[4]): every face reports how
// many tiles it overlaps, an exclusive scan turns that into pair offsets, the (tile, face) pairs
// are written in ascending face order and ONE stable radix sort by tile (prims.hpp, 13 key bits)
// leaves every tile's faces ascending.  The wide list is a stream compaction (ascending by
// construction).  Unused pair slots carry the key kPadKey and sort to the end.
constexpr unsigned kPadKey = 8191u;                // > any tile id (kG2Max^2 = 8,100 tiles at most), 13 bits
static_assert((unsigned)(kG2Max * kG2Max) <= kPadKey, "the pad key sorts behind every tile");

// NEAREST binning order: faces by descending depth of their nearest corner (zhi = largest corner z; the camera looks
// down -z), so that every tile list comes out near-first and the walk of k_pix_raster can stop at the first batch that lies
// behind every lane's worst record.  key = order-preserving bits of -zhi (ascending sort = descending zhi); a face with a
// NaN corner sorts first (its depth bound is unknown).  (The largest finite |corner z|, which scales the rounding margin of
// that stop test, comes from k_face_stats / k_pix_grid.)
__global__ __launch_bounds__(256) void k_face_depth_keys(const float *__restrict__ fz, int F, unsigned *key, unsigned *val)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float a = fz[f * 3], b = fz[f * 3 + 1], c = fz[f * 3 + 2];
    const bool nan = !(a == a) || !(b == b) || !(c == c);
    const float zhi = fmaxf(a, fmaxf(b, c));
    unsigned u = __float_as_uint(-zhi);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                  // total order of the floats as unsigned
    // the top 24 bits (round 6: three radix passes instead of four): faces whose keys tie differ by less than 2^-15 of their depth,
    // which the walk's stopping rule allows for (zMargin in k_pix_raster)
    key[f] = nan ? 0u : (u >> 8);
    val[f] = (unsigned)f;
}

__global__ __launch_bounds__(256) void k_face_span(const float *__restrict__ xy, int F, const Grid2 *__restrict__ gp, float eps,
                                                   const unsigned *__restrict__ perm,
                                                   int *span, int *isWide)
{
    // position i of the binning order holds face perm[i] (NEAREST: faces by descending depth) or face i (perm == NULL)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F) return;
    const int f = perm ? (int)perm[i] : i;
    const Grid2 g = *gp;
    const FaceBox fb = face_box(xy, f, g, eps);
    span[i] = fb.mode == 1 ? (fb.tx1 - fb.tx0 + 1) * (fb.ty1 - fb.ty0 + 1) : 0;
    isWide[i] = fb.mode == 2 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_face_pairs(const float *__restrict__ xy, int F, const Grid2 *__restrict__ gp, float eps,
                                                    const int *__restrict__ pairOff, const int *__restrict__ wideOff,
                                                    unsigned *key, unsigned *val, int *wide, int *nWide, long long cap,
                                                    const unsigned *__restrict__ perm, float4 *wideBox)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F) {
        const int f = perm ? (int)perm[i] : (int)i;                  // pairs come out in binning order: the stable sort by tile keeps it
        const Grid2 g = *gp;
        const FaceBox fb = face_box(xy, f, g, eps);
        if (fb.mode == 2) {
            wide[wideOff[i]] = f;
            wideBox[wideOff[i]] = make_float4(fb.elx, fb.ehx, fb.ely, fb.ehy);      // certified box of the face, or the whole plane
        }
        if (fb.mode == 1) {
            int o = pairOff[i];
            for (int ty = fb.ty0; ty <= fb.ty1; ++ty)
                for (int tx = fb.tx0; tx <= fb.tx1; ++tx) {
                    key[o] = (unsigned)(ty * g.gx + tx);
                    val[o] = (unsigned)f;
                    ++o;
                }
        }
        if (i == F - 1) *nWide = wideOff[F];
    }
    // pad the unused tail of the pair arrays (grid covers max(F, cap))
    const int used = pairOff[F];
    if (i >= used && i < cap) key[i] = kPadKey;
}

// tileStart[t] = first sorted position whose key is >= t (binary search, one lane per tile;
// a boundary scan would leave one lane writing the whole run of empty tiles)
__global__ __launch_bounds__(256) void k_tile_starts(const unsigned *__restrict__ skey, const int *__restrict__ nUsed, int nTilesCap,
                                                     int *tileStart)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nTilesCap) return;
    const long long n = *nUsed;                                    // the pairs really produced (only those were sorted)
    long long lo = 0, hi = n;                                      // first i in [0,n] with skey[i] >= t
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (skey[mid] < (unsigned)t) lo = mid + 1; else hi = mid;
    }
    tileStart[t] = (int)lo;
}

// pixels keyed by their tile, so that the lanes of a wave walk the same face list
__global__ __launch_bounds__(256) void k_pix_keys(const float *__restrict__ pix, int P, const Grid2 *__restrict__ gp, unsigned *key,
                                                  unsigned *val)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const Grid2 g = *gp;
    const float px = pix[p * 2], py = pix[p * 2 + 1];
    const bool tame = fabsf(px) <= kBig && fabsf(py) <= kBig;
    unsigned k = (unsigned)(kG2Max * kG2Max) * 4u;                   // the pseudo-tile nTilesCap of the NaN / Inf / huge pixels
    if (tame) {
        const int tx = cell_of(px, g.ox, g.ix, g.gx), ty = cell_of(py, g.oy, g.iy, g.gy);
        // two more key bits: the quadrant of the tile, so that 64 consecutive pixels form a compact patch
        const float fx = (px - g.ox) * g.ix - (float)tx, fy = (py - g.oy) * g.iy - (float)ty;
        k = (unsigned)(ty * g.gx + tx) * 4u + (fy >= 0.5f ? 2u : 0u) + (fx >= 0.5f ? 1u : 0u);
    }
    key[p] = k;
    val[p] = (unsigned)p;
}

// pixStart[t] = first tile-sorted pixel slot whose key is >= t (t in [0, nTilesCap + 1]; the key
// nTilesCap = kG2Max^2 collects the NaN/Inf/huge pixels), and the number of 64-pixel chunks of tile t
__global__ __launch_bounds__(256) void k_pix_chunks(const unsigned *__restrict__ xskey, int P, int nTilesCap, int *pixStart,
                                                    int *chunkCount)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nTilesCap + 1) return;
    auto lower = [&](unsigned k) {
        int lo = 0, hi = P;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (xskey[mid] < k) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int s0 = lower((unsigned)t * 4u);                       // keys carry two quadrant bits below the tile id
    pixStart[t] = s0;
    if (t <= nTilesCap) chunkCount[t] = (lower(((unsigned)t + 1u) * 4u) - s0 + 63) >> 6;
    else chunkCount[t] = 0;
}

struct Hit { int f; float z, w1, w2; };
#ifdef RAST_STATS        // probe builds only: where k_pix_raster spends its time under the NEAREST policy
__device__ unsigned long long g_rast_stats[8];
#define RAST_STAT(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicAdd(&g_rast_stats[i], v_); } while (0)
#define RAST_STAT_MAX(i, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane == 0) atomicMax(&g_rast_stats[i], v_); } while (0)
#else
#define RAST_STAT(i, v)
#define RAST_STAT_MAX(i, v)
#endif
#ifndef RAST_ILP
#define RAST_ILP 1                 // faces evaluated per trip of the face loop before they are committed in list order (1 / 2 / 4: 352 / 360 / 370 us)
#endif
#ifndef RAST_LDS_BCAST
#define RAST_LDS_BCAST 1
#endif
#ifndef RAST_PROBE_NOSTORE
#define RAST_PROBE_NOSTORE 0      // probe builds only (wrong results): the hit-record store of the NEAREST face loop left out
#endif
constexpr int kPendDepth = 4;          // admitted hits a full lane queues before the wave works the queues off (NEAREST)

// One wave per 64 pixels OF ONE TILE: the face list is then the same for every lane, so the wave
// loads 64 list entries and their 36 bytes of face data with one coalesced round trip (lane k holds
// face k) and broadcasts them one after the other with v_readlane — the per-lane formulation spent
// two dependent gather latencies on every candidate (1.5 ms at configs[4], lists of ~1,700 faces).
// FIRST policy: faces are visited in ascending index (tile list merged with the wide list), every lane keeps its first
// `knum` hits and the wave stops when all its lanes are full.  NEAREST policy: the whole list is visited (from the end
// whose faces are nearer); a full lane replaces its worst record (smallest z, then largest face index) by a better hit.
// The chunk of NaN/Inf/huge pixels (pseudo-tile nTilesCap) visits every face.
__device__ __forceinline__ float bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// a after b in the output order (z descending, then face index ascending)
__device__ __forceinline__ bool rast_worse(float za, int fa, float zb, int fb) { return za < zb || (za == zb && fa > fb); }

struct Worst { float z; int f, at; };

// NEAREST: the wave works its lanes' queues of admitted hits off together (see k_pix_raster).  Out of line on purpose:
// inlined at the three places a hit can be queued it grew the kernel sixfold, and the instruction fetch of the hot loop,
// not the work, became what a wave waited for (1.98 ms against 0.74 ms for FIRST with the same number of tests).
__device__ __noinline__ Worst rast_work_off(const int4 (*pend)[64], int npend, Worst w, int4 *hits, int p, int knum, int lane)
{
    int4 *out = hits + (size_t)p * knum;
    for (int j = 0; j < kPendDepth; ++j) {
        int4 rec = make_int4(0, 0, 0, 0);
        bool go = false;
        if (j < npend) {
            rec = pend[j][lane];
            go = rast_worse(w.z, w.f, __int_as_float(rec.y), rec.x);
        }
        if (go) out[w.at] = rec;
        unsigned long long need = __ballot(go);
        RAST_STAT(5, __popcll(need));                                   // [5] records replaced
        // The new worst of a lane that replaced: the WAVE reads that lane's knum records (lane i reads record i: one
        // coalesced request instead of knum dependent ones from a single lane), the record just replaced comes from
        // registers, and a butterfly finds the worst.
        while (need) {
            const int L = __ffsll((long long)need) - 1;
            need &= need - 1;
            const int4 *oL = hits + (size_t)__builtin_amdgcn_readlane(p, L) * knum;
            const int atL = __builtin_amdgcn_readlane(w.at, L);
            const int newF = __builtin_amdgcn_readlane(rec.x, L);
            const float newZ = __int_as_float(__builtin_amdgcn_readlane(rec.y, L));
            float wz = INFINITY;
            int wf = -1, wat = -1;
            for (int i = lane; i < knum; i += 64) {
                const int2 o = *reinterpret_cast<const int2 *>(oL + i);
                const float zi = i == atL ? newZ : __int_as_float(o.y);
                const int fi = i == atL ? newF : o.x;
                if (wat < 0 || rast_worse(zi, fi, wz, wf)) { wz = zi; wf = fi; wat = i; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float oz = __shfl_xor(wz, off);
                const int of = __shfl_xor(wf, off), oat = __shfl_xor(wat, off);
                if (oat >= 0 && (wat < 0 || rast_worse(oz, of, wz, wf))) { wz = oz; wf = of; wat = oat; }
            }
            if (lane == L) { w.z = wz; w.f = wf; w.at = wat; }
        }
    }
    return w;
}

template <bool nearest>   // the saturation policy: a template argument, so that FIRST carries none of NEAREST's state
__global__ __launch_bounds__(256) void k_pix_raster(const float *__restrict__ pix, const float *__restrict__ rng,
                                                    const float *__restrict__ fz, const float *__restrict__ fxy, int P,
                                                    int nTilesCap, const int *__restrict__ tileStart,
                                                    const int *__restrict__ list, const int *__restrict__ wide,
                                                    const int *__restrict__ nWide, int F, int knum, float eps, int4 *hits,
                                                    int *nhit, const unsigned *__restrict__ pixOrder,
                                                    const int *__restrict__ pixStart, const int *__restrict__ chunkStart,
                                                    const unsigned *__restrict__ zAbsMax, const float4 *__restrict__ wideBox)
{
    const int lane = threadIdx.x & 63;
    const int W = blockIdx.x * 4 + (threadIdx.x >> 6);              // chunk id (wave-uniform)
    if (W >= chunkStart[nTilesCap + 1]) return;
    int lo = 0, hi = nTilesCap + 1;                                 // largest tile with chunkStart[tile] <= W
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunkStart[mid] <= W) lo = mid; else hi = mid;
    }
#ifdef RAST_STATS
    const unsigned long long tStart = __builtin_amdgcn_s_memtime();
    int nBroadcast = 0;
#endif
    const int tile = lo;
    const int slot = pixStart[tile] + (W - chunkStart[tile]) * 64 + lane;
    const bool live = slot < pixStart[tile + 1];
    const int p = live ? (int)pixOrder[slot] : 0;
    const float px = pix[p * 2], py = pix[p * 2 + 1];
    const float zmin = rng[p * 2], zmax = rng[p * 2 + 1];
    int nh = live ? 0 : knum;                                       // dead lanes count as full
    int4 *out = hits + (size_t)p * knum;
    // The face-only terms (m, pp, n, q, den = k3 + eps) are computed once by the lane that loaded the face and broadcast;
    // same fp32 operations as the contract, evaluated in another lane.
    // (A certified wave-level early out before the two divisions — every lane surely outside by the signs of k1, k2 and
    // k1 + k2 - den with rounding margins — was measured: it rarely holds for all 64 pixels and its dozen instructions
    // made the kernel slower, 850 vs 739 us.)
    // NEAREST: a full lane admits a hit that beats its worst record (smallest z, then largest face index).  Finding the new
    // worst is a pass over the lane's knum records, and the lanes of a wave saturate at different faces, so admitted hits
    // wait in a short per-lane queue (LDS) and the wave works the queues off together, every lane replacing its j-th
    // pending hit in the same pass.  A queued hit is re-tested against the lane's current worst when it is taken out (the
    // threshold only tightens in between).
    __shared__ int4 s_pend[4][kPendDepth][64];
#if RAST_LDS_BCAST
    __shared__ float4 s_face[4][64][3];                              // per wave: the batch's faces, 48 bytes each
#endif
    int4(*pend)[64] = s_pend[threadIdx.x >> 6];
    float worstZ = INFINITY;                                        // the record a better hit would replace
    int worstF = -1, worstAt = 0, npend = 0;
    auto worse = [](float za, int fa, float zb, int fb) { return rast_worse(za, fa, zb, fb); };
    auto work_off = [&]() __attribute__((always_inline)) {
        const Worst w = rast_work_off(pend, npend, Worst{worstZ, worstF, worstAt}, hits, p, knum, lane);
        worstZ = w.z; worstF = w.f; worstAt = w.at;
        npend = 0;
    };
    // One (face, pixel) test in two halves: `eval` is pure arithmetic (the contract's operations, nothing else), `commit` records
    // the hit.  The face loop below evaluates RAST_ILP faces per trip before it commits them in list order: a wave's trip is one
    // long dependent chain (eleven lane reads, two IEEE division sequences of ten dependent instructions each, five compares),
    // there are only four waves per SIMD to interleave (4,096 chunks on 1,024 SIMDs), and by the counters the vector ALU was
    // "busy" 0.9 k cycles per trip for ~60 instructions — waiting for its own results (round 6: profiles/r06_raster_probes.jsonl;
    // the hit-record stores, suspected first, are not it: without them the kernel takes 0.87 instead of 0.89 ms).
    struct Ev { float z, w1, w2; bool ok; };
    auto eval = [&](float ax, float ay, float m, float pp, float n, float q, float den, float az, float bz, float cz) __attribute__((always_inline)) {
        const float s_ = px - ax, t = py - ay;
        const float k1 = s_ * q - n * t, k2 = m * t - s_ * pp;
        Ev e;
        e.w1 = k1 / den; e.w2 = k2 / den;
        const float w0 = 1 - e.w1 - e.w2;
        e.z = (w0 * az + e.w1 * bz) + e.w2 * cz;
        e.ok = (w0 >= 0 && e.w1 >= 0 && e.w2 >= 0) && (e.z >= zmin && e.z <= zmax);
        return e;
    };
    auto commit = [&](int f, const Ev &e) __attribute__((always_inline)) {
        if (!nearest) {                                               // FIRST: the round-2 body, nothing else
            if (e.ok && nh < knum) {
                out[nh] = make_int4(f, __float_as_int(e.z), __float_as_int(e.w1), __float_as_int(e.w2));
                ++nh;
            }
            return;
        }
        // (the two counters advance OUTSIDE the branches: written as ++nh / ++npend inside them the compiler merges the two
        // increments into one store through a selected address, which puts both counters into scratch memory — a scratch
        // load, a dependent scratch store and another load with s_waitcnt vmcnt(0) in every iteration of the face loop)
        int dnh = 0, dnp = 0;
        if (e.ok) {
            const int4 rec = make_int4(f, __float_as_int(e.z), __float_as_int(e.w1), __float_as_int(e.w2));
            if (nh < knum) {
#if RAST_PROBE_NOSTORE == 0
                out[nh] = rec;
#elif RAST_PROBE_NOSTORE == 2
                if (nh == 1000000) out[nh] = rec;                     // timing probe: the store stays in the code, never executes
#endif
                if (nh == 0 || worse(e.z, f, worstZ, worstF)) { worstZ = e.z; worstF = f; worstAt = nh; }
                dnh = 1;
            } else if (live && knum > 0 && worse(worstZ, worstF, e.z, f)) {   // full (dead lanes only count as full)
                pend[npend][lane] = rec;
                dnp = 1;
            }
        }
        nh += dnh;
        npend += dnp;
        if (nearest && __any(npend == kPendDepth)) work_off();
    };
    auto test = [&](int f, float ax, float ay, float m, float pp, float n, float q, float den, float az, float bz, float cz) __attribute__((always_inline)) {
        if (!nearest && nh >= knum) return;
        commit(f, eval(ax, ay, m, pp, n, q, den, az, bz, cz));
    };
    auto face_terms = [&](float2 a, float2 b, float2 c, float &m, float &pp, float &n, float &q, float &den) {
        m = b.x - a.x; pp = b.y - a.y; n = c.x - a.x; q = c.y - a.y;
        const float k3 = m * q - n * pp;
        den = k3 + eps;
    };
    const bool allFaces = tile == nTilesCap;                        // the non-tame pixels: every face, no lists
    const int ib = allFaces ? 0 : tileStart[tile], ie = allFaces ? F : tileStart[tile + 1];
    const int je = allFaces ? 0 : *nWide;
    // image-space box of this wave's pixels: a listed (regular) face whose enlarged box misses it
    // cannot be accepted by any lane (the certified-box argument of face_box) and is skipped
    float cxl = live ? px : INFINITY, cxh = live ? px : -INFINITY, cyl = live ? py : INFINITY, cyh = live ? py : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cxl = fminf(cxl, __shfl_xor(cxl, off)); cxh = fmaxf(cxh, __shfl_xor(cxh, off));
        cyl = fminf(cyl, __shfl_xor(cyl, off)); cyh = fmaxf(cyh, __shfl_xor(cyh, off));
    }
    // Wide faces: the list is scanned 64 entries at a time — lane l holds the certified box of entry jw + l (face_box; the whole
    // plane for the truly degenerate ones) and a ballot leaves the entries whose box meets this wave's pixels; only those are
    // tested, in list order.  (A face that is pruned can be dropped at any time: it records nothing.  One entry per trip through
    // scalar loads cost a memory latency per entry: 846 entries per wave at configs[4], two of them survivors.)
    int jw = -64;
    unsigned long long wmask = 0ull;
    auto wide_before = [&](int fLimit) __attribute__((always_inline)) {                            // wide faces with index < fLimit
        for (;;) {
            while (wmask == 0ull && jw + 64 < je) {
                jw += 64;
                const int idx = jw + lane;
                bool meets = false;
                if (idx < je) {
                    const float4 wb = wideBox[idx];
                    meets = !(wb.y < cxl || wb.x > cxh || wb.w < cyl || wb.z > cyh);
                }
                wmask = __ballot(meets);
            }
            if (wmask == 0ull) return;
            const int jj = jw + __ffsll((long long)wmask) - 1;
            const int f = wide[jj];
            if (f >= fLimit) return;
            wmask &= wmask - 1ull;
            const float2 a = reinterpret_cast<const float2 *>(fxy)[f * 3], b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1],
                         c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
            float m, pp, n, q, den;
            face_terms(a, b, c, m, pp, n, q, den);
            test(f, a.x, a.y, m, pp, n, q, den, fz[f * 3], fz[f * 3 + 1], fz[f * 3 + 2]);
            RAST_STAT(4, 1);                                            // [4] wide faces tested (per wave)
        }
    };
    // Software pipeline over the list, 64 entries per stage: while batch k is tested, the face data of batch k+1 (whose
    // list entries arrived during batch k-1) and the list entries of batch k+2 are in flight — the two dependent round
    // trips per batch were exposed with only ~4.5 waves per SIMD.
    struct Batch { float2 a, b, c; float az, bz, cz; };
    auto load_entry = [&](int idx) { return idx < ie ? (allFaces ? idx : list[idx]) : -1; };
    auto load_faces = [&](int fm) {
        Batch d;
        const int f = fm < 0 ? 0 : fm;                               // idle lanes read face 0 (F > 0 whenever ib < ie)
        d.a = reinterpret_cast<const float2 *>(fxy)[f * 3]; d.b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1];
        d.c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
        d.az = fz[f * 3]; d.bz = fz[f * 3 + 1]; d.cz = fz[f * 3 + 2];
        return d;
    };
    // NEAREST: the tile lists are in descending order of the faces' nearest corner depth (k_face_depth_keys), so the walk
    // stops at the first batch whose first face lies behind every lane's worst record — all later faces do as well.  (With
    // lists in face order the walk had to cover all ~1,700 faces of a tile, and their gathered 36-byte records, not the
    // tests, were what bounded the kernel: 1.92 ms against 0.92 ms for FIRST at configs[4].)
    const int nb = (ie - ib + 63) >> 6;
    const bool rev = false;
    // (1e-5: rounding of the interpolated depth; 2^-15 = 3.1e-5: the lists are sorted on the top 24 bits of the depth, so a later
    // face may lie that much of its depth in front of an earlier one)
    const float zMargin = nearest ? 4.2e-5f * __uint_as_float(*zAbsMax) : 0.f;
    auto batch_base = [&](int k) { return ib + ((rev ? nb - 1 - k : k) << 6); };
    int fmCur = -1, fmNext = -1;
    Batch cur = {}, nxt = {};
    if (nb > 0) {
        fmCur = load_entry(batch_base(0) + lane);
        cur = load_faces(fmCur);
        if (nb > 1) fmNext = load_entry(batch_base(1) + lane);
    }
    for (int kb = 0; kb < nb; ++kb) {
        if (!nearest && __all(nh >= knum)) break;
        if (kb + 1 < nb) nxt = load_faces(fmNext);
        const int fmAfter = kb + 2 < nb ? load_entry(batch_base(kb + 2) + lane) : -1;
        const int fm = fmCur;
        const float2 a = cur.a, b = cur.b, c = cur.c;
        const float az = cur.az, bz = cur.bz, cz = cur.cz;
        bool cand = fm >= 0;
        if (nearest) {
            // Depth culling: the interpolated z of a covered pixel is a convex combination of the corner depths (weights in
            // [0, 1] up to rounding), so it cannot exceed their maximum by more than a few ulps of the largest |depth| — a
            // face whose corners all lie behind the worst record of EVERY lane of the wave can be admitted by none of them.
            // (Lanes that are not full yet admit anything: -inf.  The bound is refreshed per batch; a stale one is only
            // less sharp.)
            float mw = live ? (nh >= knum ? worstZ : -INFINITY) : INFINITY;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mw = fminf(mw, __shfl_xor(mw, off));
            if (!allFaces) {                                         // sorted list: nothing from here on can be admitted
                const bool nanz = !(az == az) || !(bz == bz) || !(cz == cz);     // such faces sort first: no bound from them
                const float zfirst = bcast(nanz ? INFINITY : fmaxf(az, fmaxf(bz, cz)), 0);
                if (zfirst + zMargin < mw) break;                    // (a NaN bound compares false: walk on)
            }
            const float zhi = fmaxf(az, fmaxf(bz, cz)), zabs = fmaxf(fabsf(az), fmaxf(fabsf(bz), fabsf(cz)));
            cand = cand && !(zhi + 1e-5f * zabs < mw);               // NaN depths: tested as before
        }
        if (cand && !allFaces) {                                    // same enlarged box as face_box()
            const float lox = fminf(a.x, fminf(b.x, c.x)), hix = fmaxf(a.x, fmaxf(b.x, c.x));
            const float loy = fminf(a.y, fminf(b.y, c.y)), hiy = fmaxf(a.y, fmaxf(b.y, c.y));
            const float mg = fmaxf(hix - lox, hiy - loy) * kMargin;
            cand = !(hix + mg < cxl || lox - mg > cxh || hiy + mg < cyl || loy - mg > cyh);
        }
        float fm_, fpp, fn, fq, fden;
        face_terms(a, b, c, fm_, fpp, fn, fq, fden);
        unsigned long long todo = __ballot(cand);
        RAST_STAT(1, 1);                                                // [1] batches of 64 entries
        RAST_STAT(2, __popcll(todo));                                   // [2] faces broadcast
#ifdef RAST_STATS
        nBroadcast += __popcll(todo);
#endif
        int since = 0;
#if RAST_LDS_BCAST
        // The surviving faces are handed to the 64 pixel lanes through LDS: every lane parks the eleven words of ITS face, the
        // wave reads them back one face at a time with three 16-byte reads at a wave-uniform address (a broadcast: one LDS
        // cycle each, on the LDS pipe) — the next face's reads are in flight while this one is evaluated.  Eleven v_readlane
        // per face were 8 cycles of the VECTOR pipe each on this chip (tools/probes/valu_rate_probe.hip), a quarter of a trip.
        {
            float4 *mine = &s_face[threadIdx.x >> 6][lane][0];
            mine[0] = make_float4(a.x, a.y, fm_, fpp);
            mine[1] = make_float4(fn, fq, fden, az);
            mine[2] = make_float4(bz, cz, __int_as_float(fm), 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        auto fetch = [&](int k, float4 &q0, float4 &q1, float4 &q2) __attribute__((always_inline)) {
            const float4 *src = &s_face[threadIdx.x >> 6][k][0];
            q0 = src[0]; q1 = src[1]; q2 = src[2];
        };
        float4 c0, c1, c2, n0 = {}, n1 = {}, n2 = {};
        if (todo) fetch(__ffsll((long long)todo) - 1, c0, c1, c2);
        while (todo) {
            todo &= todo - 1;
            if (todo) fetch(__ffsll((long long)todo) - 1, n0, n1, n2);
            const int f = __float_as_int(c2.z);
            const Ev e = eval(c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y);
            if (je > 0 && !nearest) wide_before(f);                   // FIRST: wide faces merged in face order
            commit(f, e);
            if (!nearest && (++since & 7) == 0 && __all(nh >= knum)) break;
            c0 = n0; c1 = n1; c2 = n2;
        }
        __builtin_amdgcn_wave_barrier();                               // (the next batch overwrites s_face: DS operations of a wave execute in order)
#else
        while (todo) {
            int kk[RAST_ILP], ff[RAST_ILP];
            Ev ev[RAST_ILP];
            bool have[RAST_ILP];
#pragma unroll
            for (int u = 0; u < RAST_ILP; ++u) {                      // (past the end of the batch: the last face again, not committed)
                have[u] = todo != 0ull;
                kk[u] = have[u] ? __ffsll((long long)todo) - 1 : kk[u > 0 ? u - 1 : 0];
                todo &= todo - 1;
                ff[u] = __builtin_amdgcn_readlane(fm, kk[u]);
            }
#pragma unroll
            for (int u = 0; u < RAST_ILP; ++u)
                ev[u] = eval(bcast(a.x, kk[u]), bcast(a.y, kk[u]), bcast(fm_, kk[u]), bcast(fpp, kk[u]), bcast(fn, kk[u]), bcast(fq, kk[u]),
                             bcast(fden, kk[u]), bcast(az, kk[u]), bcast(bz, kk[u]), bcast(cz, kk[u]));
            bool stop = false;
#pragma unroll
            for (int u = 0; u < RAST_ILP; ++u) {
                if (!have[u] || stop) continue;                       // (wave-uniform)
                if (je > 0 && !nearest) wide_before(ff[u]);           // FIRST: wide faces merged in face order
                commit(ff[u], ev[u]);
                if (!nearest && (++since & 7) == 0 && __all(nh >= knum)) stop = true;
            }
            if (stop) break;
        }
#endif
        fmCur = fmNext; cur = nxt; fmNext = fmAfter;
    }
    if (je > 0) wide_before(0x7FFFFFFF);
    if (nearest) work_off();
    RAST_STAT(0, 1);                                                    // [0] wave-chunks
#ifdef RAST_STATS
    RAST_STAT_MAX(6, nBroadcast);                                       // [6] most faces broadcast by one wave
    RAST_STAT_MAX(7, __builtin_amdgcn_s_memtime() - tStart);            // [7] longest wave, shader cycles
#endif
    RAST_STAT(3, __popcll(__ballot(live)));                             // [3] live lanes
    if (live) nhit[p] = (nearest && RAST_PROBE_NOSTORE) ? 0 : nh;      // (probe builds wrote no records: nothing for k_pix_emit to read)
}

// one wave per pixel: rank by (z descending, face ascending), write the sorted outputs
__global__ __launch_bounds__(256) void k_pix_emit(const int4 *__restrict__ hits, const int *__restrict__ nhit,
                                                  const float *__restrict__ feat, int P, int D, int knum, float *out_feat,
                                                  long long *out_face, float *out_w)
{
    // The (face, z) keys of the pixel's hits are parked in LDS, 64 at a time, and every lane ranks its own hit against them with
    // wave-uniform 8-byte reads (LDS broadcasts, all in flight together).  Rounds 2-5 read them through the scalar cache with a
    // wait after every pair of loads: 32 dependent scalar-cache latencies per wave (round 6: 0.30 -> 0.245 ms at configs[4]).
    __shared__ int2 s_key[4][64];
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (p >= P) return;
    const int n = nhit[p];
    const int4 *h = hits + (size_t)p * knum;
    const bool vec4 = D == 4 && (((uintptr_t)feat | (uintptr_t)out_feat) & 15) == 0;   // (launch-uniform)
    for (int i0 = 0; i0 < knum; i0 += 64) {                          // (wave-uniform trip count)
        const int i = i0 + lane;
        const bool mine = i < n;
        int4 me = make_int4(0, 0, 0, 0);
        if (mine) me = h[i];
        const float zi = __int_as_float(me.y);
        int r = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {                          // the pixel's hits, 64 keys per round
            __builtin_amdgcn_wave_barrier();
            const int jl = j0 + lane;
            if (jl < n) s_key[wv][lane] = (jl >= i0 && jl < i0 + 64) ? make_int2(me.x, me.y) : *reinterpret_cast<const int2 *>(h + jl);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int m = min(64, n - j0);
            int j = 0;
            for (; j + 8 <= m; j += 8) {
                int2 o[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = s_key[wv][j + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float zj = __int_as_float(o[u].y);
                    r += (zj > zi || (zj == zi && o[u].x < me.x)) ? 1 : 0;
                }
            }
            for (; j < m; ++j) {
                const int2 o = s_key[wv][j];
                const float zj = __int_as_float(o.y);
                r += (zj > zi || (zj == zi && o.x < me.x)) ? 1 : 0;
            }
        }
        if (mine) {
            const size_t o = (size_t)p * knum + r;
            const float w1 = __int_as_float(me.z), w2 = __int_as_float(me.w), w0 = 1 - w1 - w2;
            out_face[o] = me.x;
            if (out_w) { out_w[o * 3] = w0; out_w[o * 3 + 1] = w1; out_w[o * 3 + 2] = w2; }
            const float *ff = feat + (size_t)me.x * 3 * D;
            if (vec4) {                                            // D = 4, 16-byte aligned: three 16-byte gathers, one 16-byte store (same expression per channel)
                const float4 f0 = reinterpret_cast<const float4 *>(ff)[0], f1 = reinterpret_cast<const float4 *>(ff)[1], f2 = reinterpret_cast<const float4 *>(ff)[2];
                reinterpret_cast<float4 *>(out_feat)[o] = make_float4((w0 * f0.x + w1 * f1.x) + w2 * f2.x, (w0 * f0.y + w1 * f1.y) + w2 * f2.y,
                                                                       (w0 * f0.z + w1 * f1.z) + w2 * f2.z, (w0 * f0.w + w1 * f1.w) + w2 * f2.w);
            } else {
                for (int d = 0; d < D; ++d) out_feat[o * D + d] = (w0 * ff[d] + w1 * ff[D + d]) + w2 * ff[2 * D + d];
            }
        } else if (i < knum) {
            const size_t o = (size_t)p * knum + i;                 // slots n..knum-1 stay empty
            out_face[o] = -1;
            if (out_w) { out_w[o * 3] = 0.f; out_w[o * 3 + 1] = 0.f; out_w[o * 3 + 2] = 0.f; }
            if (vec4) reinterpret_cast<float4 *>(out_feat)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int d = 0; d < D; ++d) out_feat[o * D + d] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------- backward
// Hits are grouped by face with ONE stable radix sort of (face, slot) pairs (prims.hpp): the first pass reads its keys
// straight from face_idx through a loader (empty slots get the padding key F and sort to the end) and takes the slot
// index as value.  Then one lane per SORTED hit (k_bwd_sorted): the slot list
// is read coalesced, the hit's pixel and output gradient are the only divergent gathers (the face data of
// neighbouring lanes coincide), the barycentric weights are recomputed exactly as the forward computed
// them, and the per-hit contributions are combined by a segmented scan across the wave (DPP moves) and a
// carry across the 16 waves of the block (LDS), all in a fixed order.  A face whose hits lie inside one
// block is written with plain stores; one that straddles a block boundary (1,024 sorted hits) adds its
// partial sums atomically into the cleared output, so only faces with more than 1,024 hits can see
// different summation orders between runs.
// Measured history at configs[4] (15.4 M hits, 521,850 faces): per-face linked lists built with atomicExch
// + one lane per face chasing them 0.45 + 2.12 ms (a chain of dependent, fully divergent gathers); sorted
// hits with ds_bpermute shuffles 1.07 ms (LDS pipe); DPP moves 0.89 ms; wave-local runs flushed with 36
// float atomics per wave -> carried through LDS instead: 0.46 ms.  The sort itself costs 0.49 ms.
// first-pass key loader of the hit sort: the face of slot i straight from face_idx (empty slots get the padding key F)
struct FaceKeyLoad {
    const long long *face;
    int F;
    __device__ __forceinline__ unsigned operator()(size_t i) const
    {
        const long long f = face[i];
        return (f >= 0 && f < (long long)F) ? (unsigned)f : (unsigned)F;
    }
};

constexpr int kDChunk = 4;

// Select with the lane mask in an SGPR pair (v_cndmask_b32_e64).  Written as `c ? a : b` on a per-lane bool the compiler emits the
// VCC form where it can, and gfx950 issues that form about eight times slower than any other VALU instruction (23 cycles
// against 3-4.5: tools/probes/valu_rate_probe.hip) — 21 of them were a seventh of a wave's time in k_bwd_sorted.
typedef unsigned long long lanemask_t;
__device__ __forceinline__ float sel_f(lanemask_t m, float if_set, float if_clear)
{
    float d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if_clear), "v"(if_set), "s"(m));
    return d;
}

// Data-parallel-primitive moves (full-rate VALU; ds_bpermute shuffles made this kernel LDS-pipe bound: 1.07 ms):
// row_shr:n inside the 16-lane rows, then row_bcast:15 (lane 15 of each row to the next row; rows 1 and 3
// written) and row_bcast:31 (lane 31 to rows 2 and 3).  Lanes without a source receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, false); }

struct SegFlags { bool s1, s2, s4, s8, b15, b31; };

// which lanes of the sorted key sequence continue the run of the lane the move reads from
__device__ __forceinline__ SegFlags seg_flags(unsigned key, int lane)
{
    const int k = (int)key, r = lane & 15;
    SegFlags f;
    const int k1 = dpp_mov<0x111, 0xf>(k), k2 = dpp_mov<0x112, 0xf>(k), k4 = dpp_mov<0x114, 0xf>(k), k8 = dpp_mov<0x118, 0xf>(k);
    const int k15 = dpp_mov<0x142, 0xa>(k), k31 = dpp_mov<0x143, 0xc>(k);
    f.s1 = r >= 1 && k1 == k; f.s2 = r >= 2 && k2 == k; f.s4 = r >= 4 && k4 == k; f.s8 = r >= 8 && k8 == k;
    f.b15 = (lane & 16) != 0 && k15 == k;
    f.b31 = (lane & 32) != 0 && k31 == k;
    return f;
}

// inclusive segmented scan over the wave (segments = runs of equal sorted keys), fixed order of additions
template <int N>
__device__ __forceinline__ void seg_scan(float (&v)[N], const SegFlags &f)
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float x = v[k], t;                                           // every move is executed by all lanes, then selected
        t = __int_as_float(dpp_mov<0x111, 0xf>(__float_as_int(x))); x += f.s1 ? t : 0.f;
        t = __int_as_float(dpp_mov<0x112, 0xf>(__float_as_int(x))); x += f.s2 ? t : 0.f;
        t = __int_as_float(dpp_mov<0x114, 0xf>(__float_as_int(x))); x += f.s4 ? t : 0.f;
        t = __int_as_float(dpp_mov<0x118, 0xf>(__float_as_int(x))); x += f.s8 ? t : 0.f;
        t = __int_as_float(dpp_mov<0x142, 0xa>(__float_as_int(x))); x += f.b15 ? t : 0.f;
        t = __int_as_float(dpp_mov<0x143, 0xc>(__float_as_int(x))); x += f.b31 ? t : 0.f;
        v[k] = x;
    }
}

constexpr int kBwdWaves = 16;           // waves per block of k_bwd_sorted

// What a wave knows about the runs that cross its two ends (wave-uniform).
struct WaveEnds {
    bool contL;     // the run holding lane 0 started before this wave
    bool contR;     // the run holding lane 63 goes on in the next wave
    int w;          // wave index in the block
};

// Carry the partial sums of runs across the waves of a block through LDS.  On entry v[] holds the wave's
// inclusive segmented scan.  On return the lane that ENDS a run (knext != key) holds the run's sum over the whole
// block in v[]; `ext` tells it whether the run began in an earlier block.  Fixed order of additions.
template <int N>
__device__ __forceinline__ void block_carry(float (&v)[N], float (*s_sum)[20], const int *s_one, const WaveEnds &e, int lane,
                                            bool inFirstRun, bool &ext)
{
    __syncthreads();                                                // LDS free again (previous use)
    if (lane == 63)
#pragma unroll
        for (int k = 0; k < N; ++k) s_sum[e.w][k] = v[k];
    __syncthreads();
    float c = 0.f;
    bool fromBefore = e.contL && e.w == 0;
    if (e.contL && e.w > 0 && lane < N) {
        for (int u = e.w - 1; u >= 0; --u) {
            c += s_sum[u][lane];
            if (!s_one[u]) break;
        }
    }
    if (e.contL && e.w > 0) {
        int u = e.w - 1;
        while (u > 0 && s_one[u]) --u;
        fromBefore = u == 0 && s_one[0];
    }
    const lanemask_t mFirst = __ballot(inFirstRun);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float ck = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), k));
        v[k] += sel_f(mFirst, ck, 0.f);
    }
    ext = inFirstRun && fromBefore;
}

template <int DT>   // DT = number of feature channels when it is 4 (16-byte loads and stores, loops unroll), 0 = run-time D
__global__ __launch_bounds__(kBwdWaves * 64) void k_bwd_sorted(const float *__restrict__ pix, const float *__restrict__ fxy,
                                                               const float *__restrict__ feat, const float *__restrict__ gout,
                                                               const unsigned *__restrict__ skey, const unsigned *__restrict__ sval,
                                                               long long n, int F, int D, int knum, float eps, float *gxy, float *gfeat)
{
    __shared__ float s_sum[kBwdWaves][20];
    __shared__ int s_one[kBwdWaves];
    const long long i0 = (long long)blockIdx.x * blockDim.x;
    if (skey[i0] >= (unsigned)F) return;                            // the padding tail (keys are sorted): whole block idle
    const long long i = i0 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (DT) D = DT;
    const unsigned key = i < n ? skey[i] : (unsigned)F;
    const bool valid = key < (unsigned)F;
    const unsigned kprev = (valid && i > 0) ? skey[i - 1] : 0xFFFFFFFFu;
    const unsigned knext = (valid && i + 1 < n) ? skey[i + 1] : 0xFFFFFFFFu;
    const SegFlags same = seg_flags(key, lane);
    const unsigned long long heads = __ballot(!valid || kprev != key);
    const bool inFirstRun = valid && (heads & ((2ull << lane) - 1ull)) == 0ull;   // no run starts at or below this lane
    const bool ends = valid && knext != key;                                        // the run's last hit
    WaveEnds e;
    e.w = threadIdx.x >> 6;
    e.contL = (heads & 1ull) == 0ull;
    e.contR = (__ballot(valid && knext == key) >> 63) != 0ull;
    if (lane == 0) s_one[e.w] = (e.contL && heads == 0ull && e.contR) ? 1 : 0;      // one run from before the wave to after it
    const bool lastWave = e.w == kBwdWaves - 1;
    // who writes: the lane that ends a run, plus lane 63 of the block's last wave for a run that goes on in the next block
    const bool spill = lastWave && lane == 63 && valid && knext == key;
    const int f = valid ? (int)key : 0;
    const unsigned h = valid ? sval[i] : 0u;
    const int p = (int)(h / (unsigned)knum);
    const float2 px = reinterpret_cast<const float2 *>(pix)[p];
    const float2 a = reinterpret_cast<const float2 *>(fxy)[f * 3], b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1],
                 c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
    const float m = b.x - a.x, pp = b.y - a.y, nn = c.x - a.x, q = c.y - a.y, s = px.x - a.x, t = px.y - a.y;
    const float k1 = s * q - nn * t, k2 = m * t - s * pp, k3 = m * q - nn * pp;
    const float den = k3 + eps;
    const float w1 = k1 / den, w2 = k2 / den, w0 = 1 - w1 - w2;     // the forward's expressions, bit for bit
    const float *g = gout + (size_t)h * D;
    const float *ff = feat + (size_t)f * 3 * D;
    {
        float gw1 = 0.f, gw2 = 0.f;                                  // dL/dw1, dL/dw2 (w0 = 1 - w1 - w2)
#pragma unroll
        for (int d = 0; d < (DT ? DT : D); ++d) {
            const float gd = g[d];
            gw1 += gd * (ff[D + d] - ff[d]);
            gw2 += gd * (ff[2 * D + d] - ff[d]);
        }
        const float gk1 = gw1 / den, gk2 = gw2 / den, gk3 = -(gw1 * w1 + gw2 * w2) / den;
        const float gm = gk2 * t + gk3 * q, gp_ = -gk2 * s - gk3 * nn, gn = -gk1 * t - gk3 * pp, gq = gk1 * s + gk3 * m;
        const float gs = gk1 * q - gk2 * pp, gt = -gk1 * nn + gk2 * m;
        float v[6] = {-(gm + gn + gs), -(gp_ + gq + gt), gm, gp_, gn, gq};
        {
            const lanemask_t mValid = __ballot(valid);
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = sel_f(mValid, v[k], 0.f);
        }
        seg_scan(v, same);
        bool ext;
        block_carry(v, s_sum, s_one, e, lane, inFirstRun, ext);
        float *o = gxy + (size_t)f * 6;
        if (ends && !ext) {
#pragma unroll
            for (int k = 0; k < 3; ++k) reinterpret_cast<float2 *>(o)[k] = make_float2(v[2 * k], v[2 * k + 1]);
        } else if ((ends && ext) || spill) {
#pragma unroll
            for (int k = 0; k < 6; ++k) unsafeAtomicAdd(o + k, v[k]);
        }
    }
    for (int c0 = 0; c0 < (DT ? DT : D); c0 += kDChunk) {         // one trip when DT == 4
        float v[3 * kDChunk];
#pragma unroll
        for (int d = 0; d < kDChunk; ++d) {
            const float gd = (valid && c0 + d < D) ? g[c0 + d] : 0.f;
            v[d] = w0 * gd; v[kDChunk + d] = w1 * gd; v[2 * kDChunk + d] = w2 * gd;
        }
        seg_scan(v, same);
        bool ext;
        block_carry(v, s_sum, s_one, e, lane, inFirstRun, ext);
        float *o = gfeat + (size_t)f * 3 * D + c0;
        if (ends && !ext) {
            if (DT == 4) {
#pragma unroll
                for (int vtx = 0; vtx < 3; ++vtx)
                    reinterpret_cast<float4 *>(o)[vtx] = make_float4(v[vtx * 4], v[vtx * 4 + 1], v[vtx * 4 + 2], v[vtx * 4 + 3]);
            } else {
#pragma unroll
                for (int vtx = 0; vtx < 3; ++vtx)
#pragma unroll
                    for (int d = 0; d < kDChunk; ++d)
                        if (c0 + d < D) o[(size_t)vtx * D + d] = v[vtx * kDChunk + d];
            }
        } else if ((ends && ext) || spill) {
#pragma unroll
            for (int vtx = 0; vtx < 3; ++vtx)
#pragma unroll
                for (int d = 0; d < kDChunk; ++d)
                    if (c0 + d < D) unsafeAtomicAdd(o + (size_t)vtx * D + d, v[vtx * kDChunk + d]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_bwd_runs (round 6, D = 4): the same reduction by face over the sorted hits with FOUR CONSECUTIVE hits per lane.  k_bwd_sorted
// spends two thirds of its time issuing vector instructions (181 M per launch at configs[4], 690 per wave: counters in
// profiles/r06_pmc_bwd_sorted.json), and half of those are the segmented scan — eighteen values through six DPP steps for every
// hit.  Here a lane first adds up its own four hits in registers (a run has ~30 hits: 87 % of the lanes hold one run only) and the
// wave scans ONE value set per lane, a quarter of the scans per hit.  What a lane holds:
//   * its TAIL — the hits at its end that share the lane's last key kl (all four when the lane is "uniform"); the tails are what
//     the wave's segmented scan runs over, with kl as the segment key (sorted keys: equal kl <=> same run, and a lane that is not
//     uniform has kl > the key before it, so its tail starts a segment by itself);
//   * its HEAD — when the lane is not uniform, the leading hits with the lane's first key kf: that run ENDS inside the lane, its
//     total is the head plus the scanned tail of the lane before (when that lane's kl == kf) and is written by this lane;
//   * runs that begin and end inside the lane (faces with one or two hits): written at once.
// Runs that cross wave boundaries are carried through LDS, runs that cross block boundaries add their parts atomically into the
// cleared output — as in k_bwd_sorted.  Every order of additions is fixed: bit-reproducible from run to run.
constexpr int kRunHPL = 4, kRunWaves = 4, kRunBlock = kRunWaves * 64 * kRunHPL;   // 1,024 sorted hits per block

// the value of the lane below (wave_shr:1: across the 16-lane rows too; lane 0 receives 0)
__device__ __forceinline__ float lane_below(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int lane_below(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); }

__global__ __launch_bounds__(kRunWaves * 64) void k_bwd_runs(const float *__restrict__ pix, const float *__restrict__ fxy,
                                                             const float *__restrict__ feat, const float *__restrict__ gout,
                                                             const unsigned *__restrict__ skey, const unsigned *__restrict__ sval,
                                                             long long n, int F, int knum, float eps, float *gxy, float *gfeat)
{
    constexpr int D = 4, NV = 6 + 3 * D;                            // values per run: dL/dxy (6) and dL/dfeat (3 x 4)
    __shared__ float s_sum[kRunWaves][NV];
    __shared__ int s_one[kRunWaves];
    const long long i0 = (long long)blockIdx.x * kRunBlock;
    if (skey[i0] >= (unsigned)F) return;                            // the padding tail (keys are sorted): whole block idle
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long base = i0 + (long long)tid * kRunHPL;
    unsigned k[kRunHPL], h[kRunHPL];
    if (base + kRunHPL <= n) {                                      // (n is a multiple of nothing in particular: the last lanes load one by one)
        const uint4 K = *reinterpret_cast<const uint4 *>(skey + base), H = *reinterpret_cast<const uint4 *>(sval + base);
        k[0] = K.x; k[1] = K.y; k[2] = K.z; k[3] = K.w;
        h[0] = H.x; h[1] = H.y; h[2] = H.z; h[3] = H.w;
    } else {
#pragma unroll
        for (int j = 0; j < kRunHPL; ++j) {
            k[j] = base + j < n ? skey[base + j] : (unsigned)F;
            h[j] = base + j < n ? sval[base + j] : 0u;
        }
    }
    const unsigned kprev = base == 0 ? 0xFFFFFFFFu : base - 1 < n ? skey[base - 1] : (unsigned)F;   // the key before the lane's first hit (past the end: padding)
    const unsigned knext = base + kRunHPL < n ? skey[base + kRunHPL] : 0xFFFFFFFFu;    // the key after its last
    // ---- the lane's four hits, added up run by run ------------------------------------------------------------------------
    float acc[NV], A[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) A[q] = 0.f;
    bool uniform = true;
    unsigned kcur = k[0];
#pragma unroll
    for (int j = 0; j < kRunHPL; ++j) {
        const bool valid = k[j] < (unsigned)F;
        const int f = valid ? (int)k[j] : 0;
        const unsigned hh = valid ? h[j] : 0u;
        const int p = (int)(hh / (unsigned)knum);
        const float2 px = reinterpret_cast<const float2 *>(pix)[p];
        const float2 a = reinterpret_cast<const float2 *>(fxy)[f * 3], b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1],
                     c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
        const float4 g = reinterpret_cast<const float4 *>(gout)[hh];
        const float4 f0 = reinterpret_cast<const float4 *>(feat)[f * 3], f1 = reinterpret_cast<const float4 *>(feat)[f * 3 + 1],
                     f2 = reinterpret_cast<const float4 *>(feat)[f * 3 + 2];
        const float m = b.x - a.x, pp = b.y - a.y, nn = c.x - a.x, q = c.y - a.y, s = px.x - a.x, t = px.y - a.y;
        const float k1 = s * q - nn * t, k2 = m * t - s * pp, k3 = m * q - nn * pp;
        const float den = k3 + eps;
        const float w1 = k1 / den, w2 = k2 / den, w0 = 1 - w1 - w2;     // the forward's expressions, bit for bit
        float gw1 = 0.f, gw2 = 0.f;                                  // dL/dw1, dL/dw2 (w0 = 1 - w1 - w2); channel order as k_bwd_sorted
        gw1 += g.x * (f1.x - f0.x); gw2 += g.x * (f2.x - f0.x);
        gw1 += g.y * (f1.y - f0.y); gw2 += g.y * (f2.y - f0.y);
        gw1 += g.z * (f1.z - f0.z); gw2 += g.z * (f2.z - f0.z);
        gw1 += g.w * (f1.w - f0.w); gw2 += g.w * (f2.w - f0.w);
        const float gk1 = gw1 / den, gk2 = gw2 / den, gk3 = -(gw1 * w1 + gw2 * w2) / den;
        const float gm = gk2 * t + gk3 * q, gp_ = -gk2 * s - gk3 * nn, gn = -gk1 * t - gk3 * pp, gq = gk1 * s + gk3 * m;
        const float gs = gk1 * q - gk2 * pp, gt = -gk1 * nn + gk2 * m;
        float cv[NV] = {-(gm + gn + gs), -(gp_ + gq + gt), gm, gp_, gn, gq,
                        w0 * g.x, w0 * g.y, w0 * g.z, w0 * g.w, w1 * g.x, w1 * g.y, w1 * g.z, w1 * g.w, w2 * g.x, w2 * g.y, w2 * g.z, w2 * g.w};
        if (!valid) {
#pragma unroll
            for (int q2 = 0; q2 < NV; ++q2) cv[q2] = 0.f;
        }
        if (j == 0) {
#pragma unroll
            for (int q2 = 0; q2 < NV; ++q2) acc[q2] = cv[q2];
        } else {
            const bool same = k[j] == kcur;
            if (!same && !uniform && kcur < (unsigned)F) {          // rare: a run that began AND ended inside the lane (a face with <= 2 hits)
                float *o = gxy + (size_t)kcur * 6, *of = gfeat + (size_t)kcur * 3 * D;
#pragma unroll
                for (int q2 = 0; q2 < 3; ++q2) reinterpret_cast<float2 *>(o)[q2] = make_float2(acc[2 * q2], acc[2 * q2 + 1]);
#pragma unroll
                for (int v3 = 0; v3 < 3; ++v3) reinterpret_cast<float4 *>(of)[v3] = make_float4(acc[6 + v3 * 4], acc[7 + v3 * 4], acc[8 + v3 * 4], acc[9 + v3 * 4]);
            }
            const bool firstEnds = !same && uniform;               // the lane's first run ends here: its sum waits for the carry
#pragma unroll
            for (int q2 = 0; q2 < NV; ++q2) {
                A[q2] = firstEnds ? acc[q2] : A[q2];
                acc[q2] = same ? acc[q2] + cv[q2] : cv[q2];
            }
            uniform = uniform && same;
            kcur = k[j];
        }
    }
    const unsigned kf = k[0], kl = kcur;
    const bool tailValid = kl < (unsigned)F;
    // ---- the tails: segmented scan over the wave, carry across the block's waves ------------------------------------------------
    const SegFlags same = seg_flags(kl, lane);
    const unsigned long long heads = __ballot(kprev != kl);         // a tail segment starts at this lane (kprev: the key before the lane)
    const bool inFirstRun = (heads & ((2ull << lane) - 1ull)) == 0ull;
    const bool contL = (heads & 1ull) == 0ull;                       // lane 0's tail continues the run of the wave before
    const bool contR = (__ballot(knext == kl) >> 63) != 0ull;        // lane 63's tail goes on in the next wave
    if (lane == 0) s_one[w] = (contL && heads == 0ull && contR) ? 1 : 0;
    // the first HIT of the wave continues a run of the wave before (lane 0's head when it is not uniform, else = contL)
    const bool contAny = __builtin_amdgcn_readfirstlane((int)(kf == kprev)) != 0;
    seg_scan(acc, same);                                            // acc: inclusive sums of the tails inside the wave
    __syncthreads();
    if (lane == 63)
#pragma unroll
        for (int q = 0; q < NV; ++q) s_sum[w][q] = acc[q];
    __syncthreads();
    // what the waves before hold of the run that enters this wave (fixed order: nearest wave first, as k_bwd_sorted)
    float cin = 0.f;
    bool fromBefore = contAny && w == 0;
    if (contAny && w > 0) {
        if (lane < NV) {
            for (int u = w - 1; u >= 0; --u) {
                cin += s_sum[u][lane];
                if (!s_one[u]) break;
            }
        }
        int u = w - 1;
        while (u > 0 && s_one[u]) --u;
        fromBefore = u == 0 && s_one[0];
    }
    const lanemask_t mFirst = __ballot(inFirstRun && contL);
    float cinv[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        cinv[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cin), q));
        acc[q] += sel_f(mFirst, cinv[q], 0.f);
    }
    const bool tailExt = inFirstRun && contL && fromBefore;          // the tail's run began in an earlier block
    // ---- the heads of the lanes that are not uniform: head + what the lanes before hold of that run ----------------------------
    const bool headCont = kf == kprev;                               // the head continues the run of the lane (or wave) before
    const int belowExt = lane_below((int)tailExt);
    const bool headExt = lane == 0 ? (headCont && fromBefore) : (headCont && belowExt != 0);
    const lanemask_t mHead = __ballot(!uniform);
    if (mHead != 0ull) {                                            // (wave-uniform: most waves of large faces have none)
        const lanemask_t mCarry = __ballot(!uniform && headCont), mLane0 = __ballot(lane == 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float below = sel_f(mLane0, cinv[q], lane_below(acc[q]));   // lane 0: the carry of the waves before
            A[q] += sel_f(mCarry, below, 0.f);
        }
        if (!uniform && kf < (unsigned)F) {
            float *o = gxy + (size_t)kf * 6, *of = gfeat + (size_t)kf * 3 * D;
            if (!headExt) {
#pragma unroll
                for (int q = 0; q < 3; ++q) reinterpret_cast<float2 *>(o)[q] = make_float2(A[2 * q], A[2 * q + 1]);
#pragma unroll
                for (int v3 = 0; v3 < 3; ++v3) reinterpret_cast<float4 *>(of)[v3] = make_float4(A[6 + v3 * 4], A[7 + v3 * 4], A[8 + v3 * 4], A[9 + v3 * 4]);
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) unsafeAtomicAdd(o + q, A[q]);
#pragma unroll
                for (int q = 0; q < 3 * D; ++q) unsafeAtomicAdd(of + q, A[6 + q]);
            }
        }
    }
    // ---- the tails that end with their lane; the block's last lane hands an unfinished run to the next block -------------------
    const bool ends = tailValid && knext != kl;
    const bool spill = tid == kRunWaves * 64 - 1 && tailValid && knext == kl;
    if (ends || spill) {
        float *o = gxy + (size_t)kl * 6, *of = gfeat + (size_t)kl * 3 * D;
        if (ends && !tailExt) {
#pragma unroll
            for (int q = 0; q < 3; ++q) reinterpret_cast<float2 *>(o)[q] = make_float2(acc[2 * q], acc[2 * q + 1]);
#pragma unroll
            for (int v3 = 0; v3 < 3; ++v3) reinterpret_cast<float4 *>(of)[v3] = make_float4(acc[6 + v3 * 4], acc[7 + v3 * 4], acc[8 + v3 * 4], acc[9 + v3 * 4]);
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) unsafeAtomicAdd(o + q, acc[q]);
#pragma unroll
            for (int q = 0; q < 3 * D; ++q) unsafeAtomicAdd(of + q, acc[6 + q]);
        }
    }
}

struct BwdLayout {
    size_t bytes, tmpBytes;
    unsigned *skey, *sval;
    void *tmp;
    int bits;
};

static BwdLayout make_bwd_layout(int P, int F, int knum, void *ws, size_t wsb)
{
    BwdLayout L{};
    Arena A(ws, wsb);
    const size_t n = (size_t)P * knum;
    L.skey = A.take<unsigned>(n + 1);
    L.sval = A.take<unsigned>(n + 1);
    L.bits = 1;
    while (L.bits < 32 && (1ull << L.bits) <= (unsigned long long)F) ++L.bits;       // keys are 0..F inclusive
    L.tmpBytes = prims::radix_sort_temp_bytes<unsigned, unsigned>(n);
    L.tmp = A.take<char>(L.tmpBytes);
    L.bytes = align_up(A.off, 256);
    return L;
}

struct Layout {
    int nTiles;
    size_t bytes;
    float *part, *fpart;
    Grid2 *grid;
    int *tileStart, *wide, *nWide, *span, *isWide, *pairOff, *wideOff, *nhit;
    unsigned *pkey, *pval, *skey, *list, *xkey, *xval, *xskey, *pixOrder, *perm, *zAbsMax;
    int *pixStart, *chunkCount, *chunkStart;
    long long cap;
    int4 *hits;
    float4 *wideBox;
    void *tmp;
    size_t tmpBytes;
};

static Layout make_layout(int P, int F, int knum, void *ws, size_t wsb)
{
    Layout L{};
    Arena A(ws, wsb);
    L.nTiles = kG2Max * kG2Max;                              // capacity; the device picks gx*gy <= this
    L.part = A.take<float>(kBoxBlocks * 4);
    L.fpart = A.take<float>(kBoxBlocks * 3);
    L.grid = A.take<Grid2>(1);
    L.tileStart = A.take<int>((size_t)L.nTiles + 2);
    L.nWide = A.take<int>(4);
    L.wide = A.take<int>((size_t)F + 1);
    L.wideBox = A.take<float4>((size_t)F + 1);
    L.span = A.take<int>((size_t)F + 1);
    L.isWide = A.take<int>((size_t)F + 1);
    L.pairOff = A.take<int>((size_t)F + 1);
    L.wideOff = A.take<int>((size_t)F + 1);
    L.cap = (long long)F * kMaxTiles + 1;
    L.pkey = A.take<unsigned>((size_t)L.cap);
    L.pval = A.take<unsigned>((size_t)L.cap);
    L.skey = A.take<unsigned>((size_t)L.cap);
    L.list = A.take<unsigned>((size_t)L.cap);
    L.xkey = A.take<unsigned>((size_t)P + 1);
    L.xval = A.take<unsigned>((size_t)P + 1);
    L.xskey = A.take<unsigned>((size_t)P + 1);
    L.pixOrder = A.take<unsigned>((size_t)P + 1);
    L.pixStart = A.take<int>((size_t)L.nTiles + 3);
    L.chunkCount = A.take<int>((size_t)L.nTiles + 3);
    L.chunkStart = A.take<int>((size_t)L.nTiles + 3);
    L.nhit = A.take<int>((size_t)P + 1);
    L.perm = A.take<unsigned>((size_t)F + 1);
    L.zAbsMax = A.take<unsigned>(4);
    L.hits = A.take<int4>((size_t)P * knum + 1);
    {   // temporary storage of the sorts and scans (prims.hpp): the largest of them
        const size_t a1 = prims::radix_sort_temp_bytes<unsigned, unsigned>((size_t)L.cap);
        const size_t a2 = prims::radix_sort_temp_bytes<unsigned, unsigned>((size_t)P + 1);
        const size_t a3 = prims::scan_temp_bytes<int>((size_t)(F > L.nTiles ? F : L.nTiles) + 3);
        L.tmpBytes = a1 > a2 ? a1 : a2;
        if (a3 > L.tmpBytes) L.tmpBytes = a3;
    }
    L.tmp = A.take<char>(L.tmpBytes);
    L.bytes = align_up(A.off, 256);
    return L;
}

}  // namespace rast
}  // namespace deftet

using namespace deftet;
using namespace deftet::rast;

#ifdef RAST_STATS
extern "C" int deftet_debug_rast_stats(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rast_stats), sizeof(g_rast_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rast_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" size_t deftet_sparse_render_workspace_bytes(int B, int P, int F, int knum)
{
    if (P < 0 || F < 0 || knum < 0) return 0;
    return make_layout(P, F, knum, nullptr, 0).bytes;      // shapes are processed one after another
}

extern "C" int deftet_sparse_render_fwd_policy_f32(const float *pix, const float *rng, const float *fz, const float *fxy,
                                                   const float *feat, float *out_feat, int64_t *out_face, float *out_w, int B, int P,
                                                   int F, int D, int knum, float eps, int policy, void *workspace, size_t wsb,
                                                   void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && D >= 0 && knum >= 0, "negative size");
    DEFTET_CHECK_ARG(policy == DEFTET_RASTER_NEAREST || policy == DEFTET_RASTER_FIRST, "unknown saturation policy %d", policy);
    DEFTET_CHECK_ARG((long long)P * knum < 2147483647LL && (long long)F * kMaxTiles < 2147483647LL, "P*knum or F too large");
    if (B == 0 || P == 0 || knum == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pix && rng && out_feat && out_face && (F == 0 || (fz && fxy && feat)), "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)fxy & 7) == 0, "face_vertices_image must be 8-byte aligned");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or misaligned");
    Layout L = make_layout(P, F, knum, workspace, wsb);
    DEFTET_CHECK_ARG(L.bytes <= wsb, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
    hipStream_t st = as_stream(stream_);
    for (int b = 0; b < B; ++b) {
        const float *pb = pix + (size_t)b * P * 2, *rb = rng + (size_t)b * P * 2;
        const float *zb = fz + (size_t)b * F * 3, *xb = fxy + (size_t)b * F * 6, *fb = feat + (size_t)b * F * 3 * D;
        DEFTET_LAUNCH(k_pix_bbox, dim3(kBoxBlocks), dim3(256), st, pb, P, L.part);
        DEFTET_LAUNCH(k_face_stats, dim3(kBoxBlocks), dim3(256), st, xb, zb, F, L.fpart);
        DEFTET_LAUNCH(k_pix_grid, dim3(1), dim3(64), st, L.part, L.fpart, L.grid, L.zAbsMax);
#define RAST_TRY(call)                     \
    do {                                   \
        const int rc_ = (call);            \
        if (rc_ != DEFTET_OK) return rc_;  \
    } while (0)
        const bool nearest = policy == DEFTET_RASTER_NEAREST;
        const unsigned *perm = nullptr;
        if (F > 0 && nearest) {
            DEFTET_LAUNCH(k_face_depth_keys, dim3((F + 255) / 256), dim3(256), st, zb, F, L.pkey, L.pval);
            RAST_TRY((prims::radix_sort<unsigned, unsigned>(L.pkey, L.skey, L.pval, L.perm, (size_t)F, 24, L.tmp, L.tmpBytes, st)));
            perm = L.perm;
        }
        if (F > 0) {
            DEFTET_LAUNCH(k_face_span, dim3((F + 255) / 256), dim3(256), st, xb, F, L.grid, eps, perm, L.span, L.isWide);
            DEFTET_HIP(hipMemsetAsync(L.span + F, 0, 4, st));
            DEFTET_HIP(hipMemsetAsync(L.isWide + F, 0, 4, st));
            RAST_TRY((prims::scan<int, prims::Plus, true>(L.span, L.pairOff, (size_t)F + 1, 0, prims::Plus(), L.tmp, L.tmpBytes, st)));
            RAST_TRY((prims::scan<int, prims::Plus, true>(L.isWide, L.wideOff, (size_t)F + 1, 0, prims::Plus(), L.tmp, L.tmpBytes, st)));
            DEFTET_LAUNCH(k_face_pairs, dim3((unsigned)((L.cap + 255) / 256)), dim3(256), st, xb, F, L.grid, eps, L.pairOff, L.wideOff, L.pkey,
                          L.pval, L.wide, L.nWide, L.cap, perm, L.wideBox);
            // stable sort of the (tile, face) pairs by tile — of the pairs really produced (pairOff[F], known on the device
            // only): the workgroups beyond that count find nothing to do (rounds 1-2 sorted the whole F * 16 capacity)
            RAST_TRY((prims::radix_sort<unsigned, unsigned>(L.pkey, L.skey, L.pval, L.list, (size_t)L.cap, 13, L.tmp, L.tmpBytes, st,
                                                            (const int *)(L.pairOff + F))));
            DEFTET_LAUNCH(k_tile_starts, dim3((L.nTiles + 256) / 256), dim3(256), st, (const unsigned *)L.skey, (const int *)(L.pairOff + F), L.nTiles,
                          L.tileStart);
        } else {
            DEFTET_HIP(hipMemsetAsync(L.tileStart, 0, ((size_t)L.nTiles + 2) * 4, st));
            DEFTET_HIP(hipMemsetAsync(L.nWide, 0, 16, st));
        }
        DEFTET_LAUNCH(k_pix_keys, dim3((P + 255) / 256), dim3(256), st, pb, P, L.grid, L.xkey, L.xval);
        RAST_TRY((prims::radix_sort<unsigned, unsigned>(L.xkey, L.xskey, L.xval, L.pixOrder, (size_t)P, 15, L.tmp, L.tmpBytes, st)));
        DEFTET_LAUNCH(k_pix_chunks, dim3((L.nTiles + 2 + 255) / 256), dim3(256), st, (const unsigned *)L.xskey, P, L.nTiles, L.pixStart,
                      L.chunkCount);
        RAST_TRY((prims::scan<int, prims::Plus, true>(L.chunkCount, L.chunkStart, (size_t)L.nTiles + 2, 0, prims::Plus(), L.tmp, L.tmpBytes, st)));
#undef RAST_TRY
        {
            const long long maxChunks = (long long)(P + 63) / 64 + L.nTiles + 1;     // every tile may end with a partial chunk
            if (nearest)
                DEFTET_LAUNCH(k_pix_raster<true>, dim3((unsigned)((maxChunks + 3) / 4)), dim3(256), st, pb, rb, zb, xb, P, L.nTiles, L.tileStart,
                              (const int *)L.list, L.wide, L.nWide, F, knum, eps, L.hits, L.nhit, (const unsigned *)L.pixOrder,
                              (const int *)L.pixStart, (const int *)L.chunkStart, (const unsigned *)L.zAbsMax, (const float4 *)L.wideBox);
            else
                DEFTET_LAUNCH(k_pix_raster<false>, dim3((unsigned)((maxChunks + 3) / 4)), dim3(256), st, pb, rb, zb, xb, P, L.nTiles, L.tileStart,
                              (const int *)L.list, L.wide, L.nWide, F, knum, eps, L.hits, L.nhit, (const unsigned *)L.pixOrder,
                              (const int *)L.pixStart, (const int *)L.chunkStart, (const unsigned *)L.zAbsMax, (const float4 *)L.wideBox);
        }
        DEFTET_LAUNCH(k_pix_emit, dim3((P + 3) / 4), dim3(256), st, L.hits, L.nhit, fb, P, D, knum,
                      out_feat + (size_t)b * P * knum * D, (long long *)out_face + (size_t)b * P * knum,
                      out_w ? out_w + (size_t)b * P * knum * 3 : nullptr);
    }
    return DEFTET_OK;
}

extern "C" int deftet_sparse_render_fwd_f32(const float *pix, const float *rng, const float *fz, const float *fxy,
                                            const float *feat, float *out_feat, int64_t *out_face, float *out_w, int B, int P,
                                            int F, int D, int knum, float eps, void *workspace, size_t wsb, void *stream_)
{
    return deftet_sparse_render_fwd_policy_f32(pix, rng, fz, fxy, feat, out_feat, out_face, out_w, B, P, F, D, knum, eps,
                                               DEFTET_RASTER_NEAREST, workspace, wsb, stream_);
}

extern "C" size_t deftet_sparse_render_bwd_workspace_bytes(int B, int P, int F, int knum)
{
    if (P < 0 || F < 0 || knum < 0 || (long long)P * knum >= 2147483647LL) return 0;
    return make_bwd_layout(P, F, knum, nullptr, 0).bytes;  // shapes are processed one after another
}

extern "C" int deftet_sparse_render_bwd_f32(const float *pix, const float *fxy, const float *feat, const int64_t *face_idx,
                                            const float *w, const float *gout, float *gxy, float *gfeat, int B, int P, int F,
                                            int D, int knum, float eps, void *workspace, size_t wsb, void *stream_)
{
    (void)w;                                                   // recomputed from the pixel and the face (may be NULL)
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && D >= 0 && knum >= 0, "negative size");
    DEFTET_CHECK_ARG((long long)P * knum < 2147483647LL, "P*knum too large");
    if (B == 0 || F == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(fxy && feat && gxy && gfeat, "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)fxy & 7) == 0 && ((uintptr_t)pix & 7) == 0, "face_vertices_image and pixel_coords must be 8-byte aligned");
    hipStream_t st = as_stream(stream_);
    const long long n = (long long)P * knum;
    DEFTET_HIP(hipMemsetAsync(gxy, 0, (size_t)B * F * 6 * sizeof(float), st));
    DEFTET_HIP(hipMemsetAsync(gfeat, 0, (size_t)B * F * 3 * D * sizeof(float), st));
    if (n == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pix && face_idx && gout, "null pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "backward workspace null or misaligned");
    BwdLayout L = make_bwd_layout(P, F, knum, workspace, wsb);
    DEFTET_CHECK_ARG(L.bytes <= wsb, "backward workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
    for (int b = 0; b < B; ++b) {
        {
            const int rc = prims::radix_sort_from<unsigned, unsigned>(FaceKeyLoad{(const long long *)face_idx + (size_t)b * n, F}, L.skey,
                                                                      prims::IotaLoad(), L.sval, (size_t)n, L.bits, L.tmp, L.tmpBytes, st);
            if (rc != DEFTET_OK) return rc;
        }
#define RAST_BWD(DT)                                                                                                                  \
    DEFTET_LAUNCH(k_bwd_sorted<DT>, dim3((unsigned)((n + kBwdWaves * 64 - 1) / (kBwdWaves * 64))), dim3(kBwdWaves * 64), st, pix + (size_t)b * P * 2, fxy + (size_t)b * F * 6, \
                  feat + (size_t)b * F * 3 * D, gout + (size_t)b * n * D, (const unsigned *)L.skey, (const unsigned *)L.sval, n, F, D,   \
                  knum, eps, gxy + (size_t)b * F * 6, gfeat + (size_t)b * F * 3 * D)
        // DEFTET_RAST_BWD=sorted: the round-5 kernel for D = 4 as well (A/B runs)
        static const bool runs = [] { const char *e = getenv("DEFTET_RAST_BWD"); return !(e && e[0] == 's'); }();
        const bool d4 = D == 4 && (((uintptr_t)feat | (uintptr_t)gout | (uintptr_t)gfeat) & 15) == 0;
        if (d4 && runs)
            DEFTET_LAUNCH(k_bwd_runs, dim3((unsigned)((n + kRunBlock - 1) / kRunBlock)), dim3(kRunWaves * 64), st, pix + (size_t)b * P * 2,
                          fxy + (size_t)b * F * 6, feat + (size_t)b * F * 3 * D, gout + (size_t)b * n * D, (const unsigned *)L.skey,
                          (const unsigned *)L.sval, n, F, knum, eps, gxy + (size_t)b * F * 6, gfeat + (size_t)b * F * 3 * D);
        else if (d4) RAST_BWD(4);
        else RAST_BWD(0);
#undef RAST_BWD
    }
    return DEFTET_OK;
}
