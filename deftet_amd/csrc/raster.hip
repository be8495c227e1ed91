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
//   per pixel the first `knum` kept faces in ascending face index are recorded, then ordered by
//   z descending (ties: ascending face index); features = (w0*f0 + w1*f1) + w2*f2.
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

#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace deftet {
namespace rast {

constexpr int kMaxTiles = 16;          // faces overlapping more tiles go to the wide list
constexpr int kBoxBlocks = 64;
constexpr float kBig = 1048576.0f;     // 2^20
constexpr float kTau = 1.0f / 128.0f;
constexpr float kMargin = 1.0f / 64.0f;

struct Grid2 { float ox, oy, ix, iy, lox, loy, hix, hiy; int gx, gy; };
constexpr int kG2Max = 512;            // tiles per axis (upper bound; the actual count is chosen on the device)

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

// mean image-space extent of the finite faces (per-block partials: sum of w, count)
__global__ __launch_bounds__(256) void k_face_stats(const float *__restrict__ xy, int F, float *part)
{
    __shared__ float sh[4][2];
    float sw = 0.f, cnt = 0.f;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < F; f += gridDim.x * blockDim.x) {
        const float2 a = reinterpret_cast<const float2 *>(xy)[f * 3], b = reinterpret_cast<const float2 *>(xy)[f * 3 + 1],
                     c = reinterpret_cast<const float2 *>(xy)[f * 3 + 2];
        const float w = fmaxf(fmaxf(a.x, fmaxf(b.x, c.x)) - fminf(a.x, fminf(b.x, c.x)),
                              fmaxf(a.y, fmaxf(b.y, c.y)) - fminf(a.y, fminf(b.y, c.y)));
        if (w <= 2.f * kBig && w > 0.f) { sw += w; cnt += 1.f; }      // NaN fails
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sw += __shfl_xor(sw, off); cnt += __shfl_xor(cnt, off); }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wv][0] = sw; sh[wv][1] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x * 2] = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
        part[blockIdx.x * 2 + 1] = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
    }
}

// pixel box + mean face size -> tile grid: tiles are about one mean face extent wide, so a
// typical face overlaps 2x2..3x3 tiles; never more than kG2Max tiles per axis
__global__ __launch_bounds__(64) void k_pix_grid(const float *__restrict__ part, const float *__restrict__ fpart, Grid2 *g)
{
    const int lane = threadIdx.x;
    float lo[2] = {part[lane * 4], part[lane * 4 + 1]}, hi[2] = {part[lane * 4 + 2], part[lane * 4 + 3]};
    float sw = fpart[lane * 2], cnt = fpart[lane * 2 + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        sw += __shfl_xor(sw, off);
        cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) {
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
struct FaceBox { int tx0, tx1, ty0, ty1, mode; };   // mode 0: skip, 1: tiles, 2: wide

__device__ __forceinline__ FaceBox face_box(const float *__restrict__ xy, int f, const Grid2 &g, float eps)
{
    const float2 a = reinterpret_cast<const float2 *>(xy)[f * 3], b = reinterpret_cast<const float2 *>(xy)[f * 3 + 1],
                 c = reinterpret_cast<const float2 *>(xy)[f * 3 + 2];
    FaceBox r{0, 0, 0, 0, 2};
    const bool finite = fabsf(a.x) <= kBig && fabsf(a.y) <= kBig && fabsf(b.x) <= kBig && fabsf(b.y) <= kBig &&
                        fabsf(c.x) <= kBig && fabsf(c.y) <= kBig;
    const float m = b.x - a.x, pp = b.y - a.y, n = c.x - a.x, q = c.y - a.y;
    const float k3 = m * q - n * pp;
    const float lox = fminf(a.x, fminf(b.x, c.x)), hix = fmaxf(a.x, fmaxf(b.x, c.x));
    const float loy = fminf(a.y, fminf(b.y, c.y)), hiy = fmaxf(a.y, fmaxf(b.y, c.y));
    const float w = fmaxf(hix - lox, hiy - loy);
    const bool regular = finite && fabsf(k3) >= kTau * (w * w) && fabsf(k3) >= 1024.0f * fabsf(eps) && w > 0.f;
    if (!regular) return r;
    const float mg = w * kMargin;
    const float elx = lox - mg, ehx = hix + mg, ely = loy - mg, ehy = hiy + mg;
    if (ehx < g.lox || elx > g.hix || ehy < g.loy || ely > g.hiy) { r.mode = 0; return r; }
    r.tx0 = cell_of(elx, g.ox, g.ix, g.gx); r.tx1 = cell_of(ehx, g.ox, g.ix, g.gx);
    r.ty0 = cell_of(ely, g.oy, g.iy, g.gy); r.ty1 = cell_of(ehy, g.oy, g.iy, g.gy);
    r.mode = ((r.tx1 - r.tx0 + 1) * (r.ty1 - r.ty0 + 1) <= kMaxTiles) ? 1 : 2;
    return r;
}

// Tile lists WITHOUT atomics and without a per-tile sort (round-1 history: atomic count + atomic
// fill + a bitonic sort per tile = 0.58 + 0.59 + 1.71 ms at configs[4]): every face reports how
// many tiles it overlaps, an exclusive scan turns that into pair offsets, the (tile, face) pairs
// are written in ascending face order and ONE stable radix sort by tile (rocPRIM, 19 key bits)
// leaves every tile's faces ascending.  The wide list is a stream compaction (ascending by
// construction).  Unused pair slots carry the key kPadKey and sort to the end.
constexpr unsigned kPadKey = 1u << 18;             // > any tile id (kG2Max^2 = 2^18 tiles at most)

__global__ __launch_bounds__(256) void k_face_span(const float *__restrict__ xy, int F, const Grid2 *__restrict__ gp, float eps,
                                                   int *span, int *isWide)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const Grid2 g = *gp;
    const FaceBox fb = face_box(xy, f, g, eps);
    span[f] = fb.mode == 1 ? (fb.tx1 - fb.tx0 + 1) * (fb.ty1 - fb.ty0 + 1) : 0;
    isWide[f] = fb.mode == 2 ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_face_pairs(const float *__restrict__ xy, int F, const Grid2 *__restrict__ gp, float eps,
                                                    const int *__restrict__ pairOff, const int *__restrict__ wideOff,
                                                    unsigned *key, unsigned *val, int *wide, int *nWide, long long cap)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F) {
        const int f = (int)i;
        const Grid2 g = *gp;
        const FaceBox fb = face_box(xy, f, g, eps);
        if (fb.mode == 2) wide[wideOff[f]] = f;
        if (fb.mode == 1) {
            int o = pairOff[f];
            for (int ty = fb.ty0; ty <= fb.ty1; ++ty)
                for (int tx = fb.tx0; tx <= fb.tx1; ++tx) {
                    key[o] = (unsigned)(ty * g.gx + tx);
                    val[o] = (unsigned)f;
                    ++o;
                }
        }
        if (f == F - 1) *nWide = wideOff[F];
    }
    // pad the unused tail of the pair arrays (grid covers max(F, cap))
    const int used = pairOff[F];
    if (i >= used && i < cap) key[i] = kPadKey;
}

// tileStart[t] = first sorted position whose key is >= t (binary search, one lane per tile;
// a boundary scan would leave one lane writing the whole run of empty tiles)
__global__ __launch_bounds__(256) void k_tile_starts(const unsigned *__restrict__ skey, long long n, int nTilesCap, int *tileStart)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nTilesCap) return;
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
    unsigned k = kPadKey * 4u;
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
// nTilesCap = kPadKey collects the NaN/Inf/huge pixels), and the number of 64-pixel chunks of tile t
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

// One wave per 64 pixels OF ONE TILE: the face list is then the same for every lane, so the wave
// loads 64 list entries and their 36 bytes of face data with one coalesced round trip (lane k holds
// face k) and broadcasts them one after the other with v_readlane — the per-lane formulation spent
// two dependent gather latencies on every candidate (1.5 ms at configs[4], lists of ~1,700 faces).
// Faces are visited in ascending index (tile list merged with the wide list), every lane keeps its
// first `knum` hits, the wave stops when all its lanes are full.  The chunk of NaN/Inf/huge pixels
// (pseudo-tile nTilesCap) visits every face.
__device__ __forceinline__ float bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

__global__ __launch_bounds__(256) void k_pix_raster(const float *__restrict__ pix, const float *__restrict__ rng,
                                                    const float *__restrict__ fz, const float *__restrict__ fxy, int P,
                                                    int nTilesCap, const int *__restrict__ tileStart,
                                                    const int *__restrict__ list, const int *__restrict__ wide,
                                                    const int *__restrict__ nWide, int F, int knum, float eps, int4 *hits,
                                                    int *nhit, const unsigned *__restrict__ pixOrder,
                                                    const int *__restrict__ pixStart, const int *__restrict__ chunkStart)
{
    const int lane = threadIdx.x & 63;
    const int W = blockIdx.x * 4 + (threadIdx.x >> 6);              // chunk id (wave-uniform)
    if (W >= chunkStart[nTilesCap + 1]) return;
    int lo = 0, hi = nTilesCap + 1;                                 // largest tile with chunkStart[tile] <= W
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunkStart[mid] <= W) lo = mid; else hi = mid;
    }
    const int tile = lo;
    const int slot = pixStart[tile] + (W - chunkStart[tile]) * 64 + lane;
    const bool live = slot < pixStart[tile + 1];
    const int p = live ? (int)pixOrder[slot] : 0;
    const float px = pix[p * 2], py = pix[p * 2 + 1];
    const float zmin = rng[p * 2], zmax = rng[p * 2 + 1];
    int nh = live ? 0 : knum;                                       // dead lanes count as full
    int4 *out = hits + (size_t)p * knum;
    auto test = [&](int f, float ax, float ay, float bx, float by, float cx, float cy, float az, float bz, float cz) {
        if (nh >= knum) return;
        const float m = bx - ax, pp = by - ay, n = cx - ax, q = cy - ay, s_ = px - ax, t = py - ay;
        const float k1 = s_ * q - n * t, k2 = m * t - s_ * pp, k3 = m * q - n * pp;
        const float den = k3 + eps;
        const float w1 = k1 / den, w2 = k2 / den, w0 = 1 - w1 - w2;
        if (!(w0 >= 0 && w1 >= 0 && w2 >= 0)) return;
        const float z = (w0 * az + w1 * bz) + w2 * cz;
        if (!(z >= zmin && z <= zmax)) return;
        out[nh] = make_int4(f, __float_as_int(z), __float_as_int(w1), __float_as_int(w2));
        ++nh;
    };
    const bool allFaces = tile == nTilesCap;                        // the non-tame pixels: every face, no lists
    const int ib = allFaces ? 0 : tileStart[tile], ie = allFaces ? F : tileStart[tile + 1];
    int j = 0;
    const int je = allFaces ? 0 : *nWide;
    auto wide_before = [&](int fLimit) {                            // wide faces with index < fLimit (normally none)
        while (j < je) {
            const int f = wide[j];
            if (f >= fLimit) break;
            const float2 a = reinterpret_cast<const float2 *>(fxy)[f * 3], b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1],
                         c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
            test(f, a.x, a.y, b.x, b.y, c.x, c.y, fz[f * 3], fz[f * 3 + 1], fz[f * 3 + 2]);
            ++j;
        }
    };
    // image-space box of this wave's pixels: a listed (regular) face whose enlarged box misses it
    // cannot be accepted by any lane (the certified-box argument of face_box) and is skipped
    float cxl = live ? px : INFINITY, cxh = live ? px : -INFINITY, cyl = live ? py : INFINITY, cyh = live ? py : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cxl = fminf(cxl, __shfl_xor(cxl, off)); cxh = fmaxf(cxh, __shfl_xor(cxh, off));
        cyl = fminf(cyl, __shfl_xor(cyl, off)); cyh = fmaxf(cyh, __shfl_xor(cyh, off));
    }
    for (int base = ib; base < ie; base += 64) {
        if (__all(nh >= knum)) break;
        const int idx = base + lane;
        const bool have = idx < ie;
        const int fm = have ? (allFaces ? idx : list[idx]) : 0;
        float2 a = make_float2(0.f, 0.f), b = a, c = a;
        float az = 0.f, bz = 0.f, cz = 0.f;
        bool cand = have;
        if (have) {
            a = reinterpret_cast<const float2 *>(fxy)[fm * 3]; b = reinterpret_cast<const float2 *>(fxy)[fm * 3 + 1];
            c = reinterpret_cast<const float2 *>(fxy)[fm * 3 + 2];
            az = fz[fm * 3]; bz = fz[fm * 3 + 1]; cz = fz[fm * 3 + 2];
            if (!allFaces) {                                        // same enlarged box as face_box()
                const float lox = fminf(a.x, fminf(b.x, c.x)), hix = fmaxf(a.x, fmaxf(b.x, c.x));
                const float loy = fminf(a.y, fminf(b.y, c.y)), hiy = fmaxf(a.y, fmaxf(b.y, c.y));
                const float mg = fmaxf(hix - lox, hiy - loy) * kMargin;
                cand = !(hix + mg < cxl || lox - mg > cxh || hiy + mg < cyl || loy - mg > cyh);
            }
        }
        unsigned long long todo = __ballot(cand);
        int since = 0;
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int f = __builtin_amdgcn_readlane(fm, k);
            if (je > 0) wide_before(f);
            test(f, bcast(a.x, k), bcast(a.y, k), bcast(b.x, k), bcast(b.y, k), bcast(c.x, k), bcast(c.y, k), bcast(az, k), bcast(bz, k),
                 bcast(cz, k));
            if ((++since & 7) == 0 && __all(nh >= knum)) break;
        }
    }
    if (je > 0) wide_before(0x7FFFFFFF);
    if (live) nhit[p] = nh;
}

// one wave per pixel: rank by (z descending, face ascending), write the sorted outputs
__global__ __launch_bounds__(256) void k_pix_emit(const int4 *__restrict__ hits, const int *__restrict__ nhit,
                                                  const float *__restrict__ feat, int P, int D, int knum, float *out_feat,
                                                  long long *out_face, float *out_w)
{
    // wave-uniform pixel: the inner-loop reads of hits[j] then go through the scalar cache
    const int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (p >= P) return;
    const int n = nhit[p];
    const int4 *h = hits + (size_t)p * knum;
    for (int i = lane; i < knum; i += 64) {
        if (i < n) {
            const int4 me = h[i];
            const float zi = __int_as_float(me.y);
            int r = 0;
            for (int j = 0; j < n; ++j) {
                const int2 o = *reinterpret_cast<const int2 *>(h + j);   // (face, z): wave-uniform address, one scalar-width load
                const float zj = __int_as_float(o.y);
                r += (zj > zi || (zj == zi && o.x < me.x)) ? 1 : 0;
            }
            const size_t o = (size_t)p * knum + r;
            const float w1 = __int_as_float(me.z), w2 = __int_as_float(me.w), w0 = 1 - w1 - w2;
            out_face[o] = me.x;
            out_w[o * 3] = w0; out_w[o * 3 + 1] = w1; out_w[o * 3 + 2] = w2;
            const float *ff = feat + (size_t)me.x * 3 * D;
            for (int d = 0; d < D; ++d) out_feat[o * D + d] = (w0 * ff[d] + w1 * ff[D + d]) + w2 * ff[2 * D + d];
        } else {
            const size_t o = (size_t)p * knum + i;                 // slots n..knum-1 stay empty
            out_face[o] = -1;
            out_w[o * 3] = 0.f; out_w[o * 3 + 1] = 0.f; out_w[o * 3 + 2] = 0.f;
            for (int d = 0; d < D; ++d) out_feat[o * D + d] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------- backward
__global__ __launch_bounds__(256) void k_link(const long long *__restrict__ face_idx, long long n, int F, int *head, int *next)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long f = face_idx[i];
    if (f >= 0 && f < F) next[i] = atomicExch(&head[f], (int)i);
}

constexpr int kDChunk = 8;

__global__ __launch_bounds__(256) void k_bwd_gather(const float *__restrict__ pix, const float *__restrict__ fxy,
                                                    const float *__restrict__ feat, const float *__restrict__ w,
                                                    const float *__restrict__ gout, const int *__restrict__ head,
                                                    const int *__restrict__ next, int F, int D, int knum, float eps,
                                                    float *gxy, float *gfeat)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float2 a = reinterpret_cast<const float2 *>(fxy)[f * 3], b = reinterpret_cast<const float2 *>(fxy)[f * 3 + 1],
                 c = reinterpret_cast<const float2 *>(fxy)[f * 3 + 2];
    const float m = b.x - a.x, pp = b.y - a.y, n = c.x - a.x, q = c.y - a.y;
    const float den = (m * q - n * pp) + eps;
    const float *ff = feat + (size_t)f * 3 * D;
    float gax = 0.f, gay = 0.f, gbx = 0.f, gby = 0.f, gcx = 0.f, gcy = 0.f;
    for (int c0 = 0; c0 < D; c0 += kDChunk) {
        float acc[3][kDChunk];
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int d = 0; d < kDChunk; ++d) acc[v][d] = 0.f;
        for (int h = head[f]; h >= 0; h = next[h]) {
            const float w0 = w[(size_t)h * 3], w1 = w[(size_t)h * 3 + 1], w2 = w[(size_t)h * 3 + 2];
            const float *g = gout + (size_t)h * D;
#pragma unroll
            for (int d = 0; d < kDChunk; ++d)
                if (c0 + d < D) {
                    const float gd = g[c0 + d];
                    acc[0][d] += w0 * gd; acc[1][d] += w1 * gd; acc[2][d] += w2 * gd;
                }
            if (c0 == 0) {
                float gw1 = 0.f, gw2 = 0.f;                         // dL/dw1, dL/dw2 (w0 = 1 - w1 - w2)
                for (int d = 0; d < D; ++d) {
                    const float gd = g[d];
                    gw1 += gd * (ff[D + d] - ff[d]);
                    gw2 += gd * (ff[2 * D + d] - ff[d]);
                }
                const int p = h / knum;
                const float s = pix[p * 2] - a.x, t = pix[p * 2 + 1] - a.y;
                const float gk1 = gw1 / den, gk2 = gw2 / den, gk3 = -(gw1 * w1 + gw2 * w2) / den;
                const float gm = gk2 * t + gk3 * q, gp_ = -gk2 * s - gk3 * n, gn = -gk1 * t - gk3 * pp, gq = gk1 * s + gk3 * m;
                const float gs = gk1 * q - gk2 * pp, gt = -gk1 * n + gk2 * m;
                gbx += gm; gby += gp_; gcx += gn; gcy += gq;
                gax += -(gm + gn + gs); gay += -(gp_ + gq + gt);
            }
        }
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int d = 0; d < kDChunk; ++d)
                if (c0 + d < D) gfeat[((size_t)f * 3 + v) * D + c0 + d] = acc[v][d];
    }
    float *o = gxy + (size_t)f * 6;
    o[0] = gax; o[1] = gay; o[2] = gbx; o[3] = gby; o[4] = gcx; o[5] = gcy;
}

struct Layout {
    int nTiles;
    size_t bytes;
    float *part, *fpart;
    Grid2 *grid;
    int *tileStart, *wide, *nWide, *span, *isWide, *pairOff, *wideOff, *nhit;
    unsigned *pkey, *pval, *skey, *list, *xkey, *xval, *xskey, *pixOrder;
    int *pixStart, *chunkCount, *chunkStart;
    long long cap;
    int4 *hits;
    void *tmp;
    size_t tmpBytes;
};

static Layout make_layout(int P, int F, int knum, void *ws, size_t wsb)
{
    Layout L{};
    Arena A(ws, wsb);
    L.nTiles = kG2Max * kG2Max;                              // capacity; the device picks gx*gy <= this
    L.part = A.take<float>(kBoxBlocks * 4);
    L.fpart = A.take<float>(kBoxBlocks * 2);
    L.grid = A.take<Grid2>(1);
    L.tileStart = A.take<int>((size_t)L.nTiles + 2);
    L.nWide = A.take<int>(4);
    L.wide = A.take<int>((size_t)F + 1);
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
    L.hits = A.take<int4>((size_t)P * knum + 1);
    {
        size_t a1 = 0, a2 = 0, a3 = 0;
        unsigned *u = nullptr;
        int *ip = nullptr;
        (void)rocprim::radix_sort_pairs(nullptr, a1, u, u, u, u, (size_t)L.cap, 0, 19, (hipStream_t) nullptr);
        (void)rocprim::radix_sort_pairs(nullptr, a2, u, u, u, u, (size_t)P + 1, 0, 21, (hipStream_t) nullptr);
        (void)rocprim::exclusive_scan(nullptr, a3, ip, ip, 0, (size_t)(F > L.nTiles ? F : L.nTiles) + 3, rocprim::plus<int>(), (hipStream_t) nullptr);
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

extern "C" size_t deftet_sparse_render_workspace_bytes(int B, int P, int F, int knum)
{
    if (P < 0 || F < 0 || knum < 0) return 0;
    return make_layout(P, F, knum, nullptr, 0).bytes;      // shapes are processed one after another
}

extern "C" int deftet_sparse_render_fwd_f32(const float *pix, const float *rng, const float *fz, const float *fxy,
                                            const float *feat, float *out_feat, int64_t *out_face, float *out_w, int B, int P,
                                            int F, int D, int knum, float eps, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && D >= 0 && knum >= 0, "negative size");
    DEFTET_CHECK_ARG((long long)P * knum < 2147483647LL && (long long)F * kMaxTiles < 2147483647LL, "P*knum or F too large");
    if (B == 0 || P == 0 || knum == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pix && rng && out_feat && out_face && out_w && (F == 0 || (fz && fxy && feat)), "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)fxy & 7) == 0, "face_vertices_image must be 8-byte aligned");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or misaligned");
    Layout L = make_layout(P, F, knum, workspace, wsb);
    DEFTET_CHECK_ARG(L.bytes <= wsb, "workspace too small: need %zu bytes, got %zu", L.bytes, wsb);
    hipStream_t st = as_stream(stream_);
    for (int b = 0; b < B; ++b) {
        const float *pb = pix + (size_t)b * P * 2, *rb = rng + (size_t)b * P * 2;
        const float *zb = fz + (size_t)b * F * 3, *xb = fxy + (size_t)b * F * 6, *fb = feat + (size_t)b * F * 3 * D;
        DEFTET_LAUNCH(k_pix_bbox, dim3(kBoxBlocks), dim3(256), st, pb, P, L.part);
        DEFTET_LAUNCH(k_face_stats, dim3(kBoxBlocks), dim3(256), st, xb, F, L.fpart);
        DEFTET_LAUNCH(k_pix_grid, dim3(1), dim3(64), st, L.part, L.fpart, L.grid);
        size_t need = L.tmpBytes;
        hipError_t e;
#define RAST_RP(call)                                                                              \
    do {                                                                                           \
        need = L.tmpBytes;                                                                         \
        e = (call);                                                                                \
        if (e != hipSuccess) return set_error(DEFTET_ELAUNCH, "%s: %s", #call, hipGetErrorString(e)); \
    } while (0)
        if (F > 0) {
            DEFTET_LAUNCH(k_face_span, dim3((F + 255) / 256), dim3(256), st, xb, F, L.grid, eps, L.span, L.isWide);
            DEFTET_HIP(hipMemsetAsync(L.span + F, 0, 4, st));
            DEFTET_HIP(hipMemsetAsync(L.isWide + F, 0, 4, st));
            RAST_RP(rocprim::exclusive_scan(L.tmp, need, L.span, L.pairOff, 0, (size_t)F + 1, rocprim::plus<int>(), st));
            RAST_RP(rocprim::exclusive_scan(L.tmp, need, L.isWide, L.wideOff, 0, (size_t)F + 1, rocprim::plus<int>(), st));
            DEFTET_LAUNCH(k_face_pairs, dim3((unsigned)((L.cap + 255) / 256)), dim3(256), st, xb, F, L.grid, eps, L.pairOff, L.wideOff, L.pkey,
                          L.pval, L.wide, L.nWide, L.cap);
            RAST_RP(rocprim::radix_sort_pairs(L.tmp, need, L.pkey, L.skey, L.pval, L.list, (size_t)L.cap, 0, 19, st));
            DEFTET_LAUNCH(k_tile_starts, dim3((L.nTiles + 256) / 256), dim3(256), st, L.skey, L.cap, L.nTiles, L.tileStart);
        } else {
            DEFTET_HIP(hipMemsetAsync(L.tileStart, 0, ((size_t)L.nTiles + 2) * 4, st));
            DEFTET_HIP(hipMemsetAsync(L.nWide, 0, 16, st));
        }
        DEFTET_LAUNCH(k_pix_keys, dim3((P + 255) / 256), dim3(256), st, pb, P, L.grid, L.xkey, L.xval);
        RAST_RP(rocprim::radix_sort_pairs(L.tmp, need, L.xkey, L.xskey, L.xval, L.pixOrder, (size_t)P, 0, 21, st));
#undef RAST_RP
        DEFTET_LAUNCH(k_pix_chunks, dim3((L.nTiles + 2 + 255) / 256), dim3(256), st, (const unsigned *)L.xskey, P, L.nTiles, L.pixStart,
                      L.chunkCount);
        need = L.tmpBytes;
        e = rocprim::exclusive_scan(L.tmp, need, L.chunkCount, L.chunkStart, 0, (size_t)L.nTiles + 2, rocprim::plus<int>(), st);
        if (e != hipSuccess) return set_error(DEFTET_ELAUNCH, "rocprim::exclusive_scan: %s", hipGetErrorString(e));
        {
            const long long maxChunks = (long long)(P + 63) / 64 + L.nTiles + 1;     // every tile may end with a partial chunk
            DEFTET_LAUNCH(k_pix_raster, dim3((unsigned)((maxChunks + 3) / 4)), dim3(256), st, pb, rb, zb, xb, P, L.nTiles, L.tileStart,
                          (const int *)L.list, L.wide, L.nWide, F, knum, eps, L.hits, L.nhit, (const unsigned *)L.pixOrder,
                          (const int *)L.pixStart, (const int *)L.chunkStart);
        }
        DEFTET_LAUNCH(k_pix_emit, dim3((P + 3) / 4), dim3(256), st, L.hits, L.nhit, fb, P, D, knum,
                      out_feat + (size_t)b * P * knum * D, (long long *)out_face + (size_t)b * P * knum, out_w + (size_t)b * P * knum * 3);
    }
    return DEFTET_OK;
}

extern "C" size_t deftet_sparse_render_bwd_workspace_bytes(int B, int P, int F, int knum)
{
    if (P < 0 || F < 0 || knum < 0) return 0;
    return align_up((size_t)F * 4, 256) + align_up((size_t)P * knum * 4, 256);
}

extern "C" int deftet_sparse_render_bwd_f32(const float *pix, const float *fxy, const float *feat, const int64_t *face_idx,
                                            const float *w, const float *gout, float *gxy, float *gfeat, int B, int P, int F,
                                            int D, int knum, float eps, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && F >= 0 && D >= 0 && knum >= 0, "negative size");
    DEFTET_CHECK_ARG((long long)P * knum < 2147483647LL, "P*knum too large");
    if (B == 0 || F == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(fxy && feat && gxy && gfeat, "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)fxy & 7) == 0, "face_vertices_image must be 8-byte aligned");
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && wsb >= deftet_sparse_render_bwd_workspace_bytes(B, P, F, knum),
                     "backward workspace null, misaligned or too small");
    int *head = static_cast<int *>(workspace);
    int *next = reinterpret_cast<int *>(static_cast<char *>(workspace) + align_up((size_t)F * 4, 256));
    const long long n = (long long)P * knum;
    for (int b = 0; b < B; ++b) {
        DEFTET_HIP(hipMemsetAsync(head, 0xFF, (size_t)F * 4, st));
        if (n > 0) {
            DEFTET_CHECK_ARG(pix && face_idx && w && gout, "null pointer");
            DEFTET_LAUNCH(k_link, dim3((unsigned)((n + 255) / 256)), dim3(256), st, (const long long *)face_idx + (size_t)b * n, n, F, head, next);
        }
        DEFTET_LAUNCH(k_bwd_gather, dim3((F + 255) / 256), dim3(256), st, pix + (size_t)b * P * 2, fxy + (size_t)b * F * 6,
                      feat + (size_t)b * F * 3 * D, w + (size_t)b * n * 3, gout + (size_t)b * n * D, head, next, F, D, knum, eps,
                      gxy + (size_t)b * F * 6, gfeat + (size_t)b * F * 3 * D);
    }
    return DEFTET_OK;
}
