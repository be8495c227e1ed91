// check_sign.hip — N1: ground-truth occupancy of points by ray parity,
//     kal.ops.mesh.check_sign(verts, faces, points, hash_resolution=512)
// as called at layers/DefTet/deftet.py:46 (all T tet centroids of every shape, every training step),
// eval.py:239 and dataloader.py:92.  PARITY UNPINNED: Kaolin is third-party, un-vendored and
// un-pinned; the contract implemented here (ray p + t*(1,0,0), Moller-Trumbore in fp32 in a fixed
// operation order, eps = 1e-7, inside <=> odd number of crossings) is stated in
// oracle/deftet_oracle_sign.c and DESIGN.md section 6b, and the kernels reproduce that oracle bit
// for bit.
//
// Two exact paths behind one entry point:
//  * DEFTET_CS_BRUTE — every point against every face; the face records are wave-uniform and
//    come through the scalar cache (the restructured equivalent of Kaolin's CUDA kernel);
//  * DEFTET_CS_AUTO  — faces are binned into a G x G grid over the (y,z) plane, the plane
//    perpendicular to the ray; a point tests only the faces of its cell.  The u/v conditions of the
//    test depend on (y,z) only, so a face can accept a point only if the point lies in the face's
//    projected triangle up to rounding.  As for the tets this is made exact
//    by certification: a face is REGULAR iff its coordinates are finite and <= 2^20 in magnitude,
//    its (y,z) box has extent w >= 2^-30 and |a| >= 2^-12 w^2 (a = twice the projected area);
//    then every accepted point lies within w/32 of the box (error of the fp32 barycentrics
//    <= ~6u(|s|/w)/tau + 6u/tau < 2^-7, |s| <= 2.03 w), so the box enlarged by w/32 is a certified
//    filter and the monotone cell map keeps it one.  Irregular faces (projected slivers, i.e.
//    silhouette faces, non-finite or huge ones) and faces spanning more than kMaxSpan cells go to a
//    per-shape list that every point tests; non-finite / huge points test every face.
#include <cstring>
#include "prims.hpp"

#include "common.hpp"

namespace deftet {
namespace cs {

constexpr float kEps = 1e-7f;
constexpr float kBig = 1048576.0f;          // 2^20
constexpr float kTau = 1.0f / 4096.0f;      // regular face: |a| >= tau * w^2
constexpr float kMargin = 1.0f / 32.0f;     // box enlargement in units of w
constexpr float kWMin = 9.3132257e-10f;     // 2^-30
constexpr int kMaxG = 512;
constexpr int kMaxSpan = 16;                // cells per binned face; larger faces join the "all points" list
constexpr int kParts = 128;                 // box partials per shape

// 48-byte record: v1 (3) | e1 (3) | e2 (3) | a | 2 pad
struct Rec {
    float4 r0, r1, r2;
};

// identical operation order to oracle/deftet_oracle_sign.c cs_hit (no FMA: -ffp-contract=off)
__device__ __forceinline__ int hit(const float4 r0, const float4 r1, const float4 r2, float px, float py, float pz)
{
    const float e1x = r0.w, e1y = r1.x, e1z = r1.y, e2x = r1.z, e2y = r1.w, e2z = r2.x, a = r2.y;
    if (a > -kEps && a < kEps) return 0;
    const float f = 1.0f / a;
    const float sx = px - r0.x, sy = py - r0.y, sz = pz - r0.z;
    const float u = f * (sy * (-e2z) + sz * e2y);
    if (u < 0.0f || u > 1.0f) return 0;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = f * qx;
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    return t > kEps ? 1 : 0;
}

__device__ __forceinline__ int cell_of(float x, float o, float inv, int G)
{
    float f = floorf((x - o) * inv);
    f = fminf(fmaxf(f, 0.f), (float)(G - 1));          // NaN -> 0 (fmaxf/fminf drop NaN)
    return (int)f;
}

// one shape's slice of the vertex / face / record arrays.  Uniform batches: V and F per shape, ONE face
// list shared by all shapes (Kaolin's signature); ragged batches (a different ground-truth mesh per shape,
// as in layers/DefTet/deftet.py:44-47): offsets into concatenated arrays.
struct Mesh {
    long long vBase, fBase, rBase;
    int V, F;
};
__device__ __forceinline__ Mesh mesh_of(int b, int V, int F, const int *__restrict__ vOff, const int *__restrict__ fOff)
{
    Mesh m;
    if (fOff) {
        m.vBase = vOff[b]; m.V = vOff[b + 1] - vOff[b];
        m.fBase = fOff[b]; m.F = fOff[b + 1] - fOff[b];
        m.rBase = m.fBase;
    } else {
        m.vBase = (long long)b * V; m.V = V;
        m.fBase = 0; m.F = F;
        m.rBase = (long long)b * F;
    }
    return m;
}

struct Box {
    float ylo, yhi, zlo, zhi;                           // enlarged (y,z) box of a regular face
};

// counters per shape: [0] irregular/wide faces, [1] spare
// kind: 0 = never hits (|a| < eps), 1 = regular (box valid), 2 = irregular
__device__ __forceinline__ int classify(const float *v1, const float *v2, const float *v3, float a, Box &bx)
{
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) finite = finite && fabsf(v1[k]) <= kBig && fabsf(v2[k]) <= kBig && fabsf(v3[k]) <= kBig;   // NaN fails
    if (!finite) return 2;
    if (a > -kEps && a < kEps) return 0;
    const float ylo = fminf(v1[1], fminf(v2[1], v3[1])), yhi = fmaxf(v1[1], fmaxf(v2[1], v3[1]));
    const float zlo = fminf(v1[2], fminf(v2[2], v3[2])), zhi = fmaxf(v1[2], fmaxf(v2[2], v3[2]));
    const float w = fmaxf(yhi - ylo, zhi - zlo);
    if (!(w >= kWMin) || !(fabsf(a) >= kTau * (w * w))) return 2;
    const float m = w * kMargin;
    bx.ylo = ylo - m; bx.yhi = yhi + m; bx.zlo = zlo - m; bx.zhi = zhi + m;
    return 1;
}

// records + classification + per-block partial of the regular boxes; irregular faces are listed
__global__ __launch_bounds__(256) void k_prep(const float *__restrict__ verts, const long long *__restrict__ faces, int V, int F,
                                              float4 *rec, signed char *kind, float4 *box, float *part, int *counters, int *irreg,
                                              int *bad, const int *__restrict__ vOff, const int *__restrict__ fOff)
{
    __shared__ float sh[4][4];
    const int b = blockIdx.y;
    const Mesh M = mesh_of(b, V, F, vOff, fOff);
    V = M.V;
    F = M.F;
    const float *vb = verts + (size_t)M.vBase * 3;
    faces += (size_t)M.fBase * 3;
    float lo[2] = {INFINITY, INFINITY}, hi[2] = {-INFINITY, -INFINITY};
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < F; k += gridDim.x * blockDim.x) {
        long long i0 = faces[(size_t)k * 3], i1 = faces[(size_t)k * 3 + 1], i2 = faces[(size_t)k * 3 + 2];
        if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) {      // torch.index_select raises
            *bad = 1;
            i0 = i1 = i2 = 0;
        }
        const float v1[3] = {vb[i0 * 3], vb[i0 * 3 + 1], vb[i0 * 3 + 2]};
        const float v2[3] = {vb[i1 * 3], vb[i1 * 3 + 1], vb[i1 * 3 + 2]};
        const float v3[3] = {vb[i2 * 3], vb[i2 * 3 + 1], vb[i2 * 3 + 2]};
        const float e1x = v2[0] - v1[0], e1y = v2[1] - v1[1], e1z = v2[2] - v1[2];
        const float e2x = v3[0] - v1[0], e2y = v3[1] - v1[1], e2z = v3[2] - v1[2];
        const float a = e1y * (-e2z) + e1z * e2y;
        float4 *r = rec + ((size_t)M.rBase + k) * 3;
        r[0] = make_float4(v1[0], v1[1], v1[2], e1x);
        r[1] = make_float4(e1y, e1z, e2x, e2y);
        r[2] = make_float4(e2z, a, 0.f, 0.f);
        if (!kind) continue;                                         // brute path: records only
        Box bx = {0.f, 0.f, 0.f, 0.f};
        const int kd = classify(v1, v2, v3, a, bx);
        kind[(size_t)M.rBase + k] = (signed char)kd;
        box[(size_t)M.rBase + k] = make_float4(bx.ylo, bx.yhi, bx.zlo, bx.zhi);
        if (kd == 1) {
            lo[0] = fminf(lo[0], bx.ylo); hi[0] = fmaxf(hi[0], bx.yhi);
            lo[1] = fminf(lo[1], bx.zlo); hi[1] = fmaxf(hi[1], bx.zhi);
        } else if (kd == 2) {
            irreg[(size_t)M.rBase + atomicAdd(&counters[b * 2], 1)] = k;
        }
    }
    if (!kind) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = lo[0]; sh[w][1] = hi[0]; sh[w][2] = lo[1]; sh[w][3] = hi[1]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        float v = sh[0][k];
        for (int i = 1; i < 4; ++i) v = (k & 1) ? fmaxf(v, sh[i][k]) : fminf(v, sh[i][k]);
        part[((size_t)b * kParts + blockIdx.x) * 4 + k] = v;
    }
}

struct Dom {
    float oy, oz, invy, invz;
};

// every wave reduces the <= kParts box partials of its shape (cheaper than one more launch)
__device__ __forceinline__ Dom reduce_domain(const float *__restrict__ part, int nPart, int G)
{
    const int lane = threadIdx.x & 63;
    float v[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = lane; i < nPart; i += 64) {
        v[0] = fminf(v[0], part[i * 4]); v[1] = fmaxf(v[1], part[i * 4 + 1]);
        v[2] = fminf(v[2], part[i * 4 + 2]); v[3] = fmaxf(v[3], part[i * 4 + 3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v[0] = fminf(v[0], __shfl_xor(v[0], off)); v[1] = fmaxf(v[1], __shfl_xor(v[1], off));
        v[2] = fminf(v[2], __shfl_xor(v[2], off)); v[3] = fmaxf(v[3], __shfl_xor(v[3], off));
    }
    Dom d;
    const bool oky = v[1] >= v[0], okz = v[3] >= v[2];
    d.oy = oky ? v[0] : 0.f;
    d.oz = okz ? v[2] : 0.f;
    d.invy = (oky && v[1] - v[0] > 1e-30f) ? (float)G / (v[1] - v[0]) : 0.f;
    d.invz = (okz && v[3] - v[2] > 1e-30f) ? (float)G / (v[3] - v[2]) : 0.f;
    return d;
}

// pass 0: count the faces per cell (wide faces join the irregular list); pass 1: fill the lists
template <int PASS>
__global__ __launch_bounds__(256) void k_bin(const signed char *kind, const float4 *__restrict__ box,
                                             const float *__restrict__ part, int nPart, int F, int G, float *dom, int *cellCount,
                                             const int *__restrict__ cellStart, int *cellFill, int *list, int *counters, int *irreg,
                                             signed char *kindOut, const int *__restrict__ fOff)
{
    const int b = blockIdx.y;
    const Mesh M = mesh_of(b, 0, F, fOff, fOff);
    F = M.F;
    const Dom d = reduce_domain(part + (size_t)b * kParts * 4, nPart, G);
    if (PASS == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        dom[b * 4] = d.oy; dom[b * 4 + 1] = d.oz; dom[b * 4 + 2] = d.invy; dom[b * 4 + 3] = d.invz;
    }
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= F) return;
    const size_t i = (size_t)M.rBase + k;
    if (kind[i] != 1) return;
    const float4 bx = box[i];
    const int cy0 = cell_of(bx.x, d.oy, d.invy, G), cy1 = cell_of(bx.y, d.oy, d.invy, G);
    const int cz0 = cell_of(bx.z, d.oz, d.invz, G), cz1 = cell_of(bx.w, d.oz, d.invz, G);
    const int span = (cy1 - cy0 + 1) * (cz1 - cz0 + 1);
    if (PASS == 0 && span > kMaxSpan) {
        irreg[(size_t)M.rBase + atomicAdd(&counters[b * 2], 1)] = k;
        kindOut[i] = 3;                                              // wide: handled with the irregular ones
        return;
    }
    const size_t cb = (size_t)b * ((size_t)G * G + 1);
    for (int cz = cz0; cz <= cz1; ++cz)
        for (int cy = cy0; cy <= cy1; ++cy) {
            const size_t c = cb + (size_t)cz * G + cy;
            if (PASS == 0) atomicAdd(&cellCount[c], 1);
            else list[cellStart[c] + atomicAdd(&cellFill[c], 1)] = k;
        }
}

__global__ __launch_bounds__(256) void k_query(const float *__restrict__ points, const float4 *__restrict__ rec, int N, int F, int G,
                                               const float *__restrict__ dom, const int *__restrict__ cellStart,
                                               const int *__restrict__ list, const int *__restrict__ counters,
                                               const int *__restrict__ irreg, unsigned char *out, int *count,
                                               const int *__restrict__ fOff)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Mesh M = mesh_of(b, 0, F, fOff, fOff);
    F = M.F;
    const float *p = points + ((size_t)b * N + i) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float4 *rb = rec + (size_t)M.rBase * 3;
    int c = 0;
    const bool regular = fabsf(px) <= kBig && fabsf(py) <= kBig && fabsf(pz) <= kBig;     // NaN fails
    if (!regular) {                                                 // not certified for the grid: every face
        for (int k = 0; k < F; ++k) c += hit(rb[k * 3], rb[k * 3 + 1], rb[k * 3 + 2], px, py, pz);
    } else {
        const int cy = cell_of(py, dom[b * 4], dom[b * 4 + 2], G), cz = cell_of(pz, dom[b * 4 + 1], dom[b * 4 + 3], G);
        const size_t cc = (size_t)b * ((size_t)G * G + 1) + (size_t)cz * G + cy;
        const int s = cellStart[cc], e = cellStart[cc + 1];
        int j = s;
        for (; j + 2 <= e; j += 2) {                                 // two record gathers in flight
            const int k0 = list[j], k1 = list[j + 1];
            const float4 a0 = rb[k0 * 3], a1 = rb[k0 * 3 + 1], a2 = rb[k0 * 3 + 2];
            const float4 b0 = rb[k1 * 3], b1 = rb[k1 * 3 + 1], b2 = rb[k1 * 3 + 2];
            c += hit(a0, a1, a2, px, py, pz);
            c += hit(b0, b1, b2, px, py, pz);
        }
        if (j < e) {
            const int k0 = list[j];
            c += hit(rb[k0 * 3], rb[k0 * 3 + 1], rb[k0 * 3 + 2], px, py, pz);
        }
        const int nI = counters[b * 2];                              // wave-uniform: irregular + wide faces
        for (int q = 0; q < nI; ++q) {
            const int k = irreg[(size_t)M.rBase + q];
            c += hit(rb[k * 3], rb[k * 3 + 1], rb[k * 3 + 2], px, py, pz);
        }
    }
    out[(size_t)b * N + i] = (unsigned char)(c & 1);
    if (count) count[(size_t)b * N + i] = c;
}

// brute force: the face index is wave-uniform, records arrive as scalar loads
__global__ __launch_bounds__(256) void k_brute(const float *__restrict__ points, const float4 *__restrict__ rec, int N, int F,
                                               unsigned char *out, int *count, const int *__restrict__ fOff)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < N;
    const Mesh M = mesh_of(b, 0, F, fOff, fOff);
    F = M.F;
    const float *p = points + ((size_t)b * N + (live ? i : 0)) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    const float4 *rb = rec + (size_t)M.rBase * 3;
    int c = 0;
    for (int k = 0; k < F; ++k) c += hit(rb[k * 3], rb[k * 3 + 1], rb[k * 3 + 2], px, py, pz);
    if (live) {
        out[(size_t)b * N + i] = (unsigned char)(c & 1);
        if (count) count[(size_t)b * N + i] = c;
    }
}

static int pick_G(int F)
{
    int G = (int)ceil(sqrt((double)(F > 0 ? F : 1)) / 2.0);
    if (G < 1) G = 1;
    if (G > kMaxG) G = kMaxG;
    return G;
}

struct Layout {
    int G;
    size_t bytes, scanTmpBytes;
    float4 *rec, *box;
    signed char *kind;
    float *part, *dom;
    int *counters, *irreg, *cellCount, *cellStart, *cellFill, *list;
    void *scanTmp;
};

static Layout make_layout(int B, int F, long long nRec, int algo, void *ws, size_t wsBytes)   // F: largest mesh, nRec: all faces
{
    Layout L{};
    Arena A(ws, wsBytes);
    L.rec = A.take<float4>((size_t)nRec * 3);
    if (algo != 1) {
        L.G = pick_G(F);
        const size_t nc = (size_t)B * ((size_t)L.G * L.G + 1);
        L.box = A.take<float4>((size_t)nRec);
        L.kind = A.take<signed char>((size_t)nRec);
        L.part = A.take<float>((size_t)B * kParts * 4);
        L.dom = A.take<float>((size_t)B * 4);
        L.irreg = A.take<int>((size_t)nRec);
        L.counters = A.take<int>((size_t)B * 2);
        L.cellCount = A.take<int>(nc);                               // counters, cellCount, cellFill are cleared together
        L.cellFill = A.take<int>(nc);
        L.cellStart = A.take<int>(nc);
        L.list = A.take<int>((size_t)nRec * kMaxSpan);
        L.scanTmpBytes = prims::scan_temp_bytes<int>(nc);
        L.scanTmp = A.take<char>(L.scanTmpBytes);
    }
    L.bytes = align_up(A.off, 256);
    return L;
}

}  // namespace cs
}  // namespace deftet

using namespace deftet;
using namespace deftet::cs;

// shared implementation: uniform batches (vOff == fOff == NULL: V, F per shape, one shared face list)
// and ragged ones (concatenated meshes with int32 offsets [B+1], F = largest mesh, nRec = all faces)
static int check_sign_run(const float *verts, const int64_t *faces, const int *vOff, const int *fOff, const float *points,
                          uint8_t *inside, int32_t *count, int32_t *bad_flag, int B, int V, int F, long long nRec, int N, int algo,
                          void *workspace, size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && V >= 0 && F >= 0 && N >= 0 && nRec >= 0, "negative size");
    DEFTET_CHECK_ARG(algo == 0 || algo == 1, "unknown algo %d", algo);
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds 65535", B);
    DEFTET_CHECK_ARG(nRec * kMaxSpan < 0x7FFFFFFFLL, "too many faces");
    if (B == 0 || N == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(points && inside && bad_flag, "null pointer");
    hipStream_t st = as_stream(stream_);
    DEFTET_HIP(hipMemsetAsync(bad_flag, 0, 4, st));
    const dim3 blk(256), gn((N + 255) / 256, B);
    if (F == 0 || nRec == 0) {                                       // no surface: everything is outside
        DEFTET_HIP(hipMemsetAsync(inside, 0, (size_t)B * N, st));
        if (count) DEFTET_HIP(hipMemsetAsync(count, 0, (size_t)B * N * 4, st));
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(verts && faces, "null mesh");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    Layout L = make_layout(B, F, nRec, algo, workspace, workspace_bytes);
    DEFTET_CHECK_ARG(L.bytes <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", L.bytes, workspace_bytes);
    int pb = (F + 255) / 256;
    if (pb > kParts) pb = kParts;
    if (algo == 1) {
        DEFTET_LAUNCH(k_prep, dim3(pb, B), blk, st, verts, (const long long *)faces, V, F, L.rec, (signed char *)nullptr,
                      (float4 *)nullptr, (float *)nullptr, (int *)nullptr, (int *)nullptr, bad_flag, vOff, fOff);
        DEFTET_LAUNCH(k_brute, gn, blk, st, points, (const float4 *)L.rec, N, F, inside, count, fOff);
        return DEFTET_OK;
    }
    const size_t nc = (size_t)B * ((size_t)L.G * L.G + 1);
    DEFTET_HIP(hipMemsetAsync(L.counters, 0, (size_t)((char *)L.cellStart - (char *)L.counters), st));   // counters, cellCount, cellFill
    DEFTET_LAUNCH(k_prep, dim3(pb, B), blk, st, verts, (const long long *)faces, V, F, L.rec, L.kind, L.box, L.part, L.counters, L.irreg,
                  bad_flag, vOff, fOff);
    const dim3 gf((F + 255) / 256, B);
    DEFTET_LAUNCH(k_bin<0>, gf, blk, st, (const signed char *)L.kind, (const float4 *)L.box, (const float *)L.part, pb, F, L.G, L.dom,
                  L.cellCount, (const int *)nullptr, (int *)nullptr, (int *)nullptr, L.counters, L.irreg, L.kind, fOff);
    {
        const int rc = prims::scan<int, prims::Plus, true>(L.cellCount, L.cellStart, nc, 0, prims::Plus(), L.scanTmp, L.scanTmpBytes, st);
        if (rc != DEFTET_OK) return rc;
    }
    DEFTET_LAUNCH(k_bin<1>, gf, blk, st, (const signed char *)L.kind, (const float4 *)L.box, (const float *)L.part, pb, F, L.G, L.dom,
                  L.cellCount, (const int *)L.cellStart, L.cellFill, L.list, L.counters, L.irreg, L.kind, fOff);
    DEFTET_LAUNCH(k_query, gn, blk, st, points, (const float4 *)L.rec, N, F, L.G, (const float *)L.dom, (const int *)L.cellStart,
                  (const int *)L.list, (const int *)L.counters, (const int *)L.irreg, inside, count, fOff);
    return DEFTET_OK;
}

extern "C" size_t deftet_check_sign_workspace_bytes(int B, int F, int algo)
{
    if (B <= 0 || F < 0) return 0;
    return make_layout(B, F, (long long)B * F, algo, nullptr, 0).bytes;
}

extern "C" int deftet_check_sign_f32(const float *verts, const int64_t *faces, const float *points, uint8_t *inside, int32_t *count,
                                     int32_t *bad_flag, int B, int V, int F, int N, int algo, void *workspace, size_t workspace_bytes,
                                     void *stream_)
{
    return check_sign_run(verts, faces, nullptr, nullptr, points, inside, count, bad_flag, B, V, F, (long long)B * F, N, algo, workspace,
                          workspace_bytes, stream_);
}

extern "C" size_t deftet_check_sign_ragged_workspace_bytes(int B, long long n_face_total, int n_face_max, int algo)
{
    if (B <= 0 || n_face_total < 0 || n_face_max < 0) return 0;
    return make_layout(B, n_face_max, n_face_total, algo, nullptr, 0).bytes;
}

extern "C" int deftet_check_sign_ragged_f32(const float *verts_cat, const int32_t *vert_offsets, const int64_t *faces_cat,
                                            const int32_t *face_offsets, const float *points, uint8_t *inside, int32_t *count,
                                            int32_t *bad_flag, int B, long long n_face_total, int n_face_max, int N, int algo,
                                            void *workspace, size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(B == 0 || (vert_offsets && face_offsets), "null offset arrays");
    return check_sign_run(verts_cat, faces_cat, vert_offsets, face_offsets, points, inside, count, bad_flag, B, 0, n_face_max,
                          n_face_total, N, algo, workspace, workspace_bytes, stream_);
}
