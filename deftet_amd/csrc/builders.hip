// builders.hip — A2-A6 tet adjacency / face-table builders on the GPU (gfx950).
//
// The reference builds these tables on the host with std::map / unordered_set / Python
// dicts (utils/lib/*/run.cpp, utils/tet_utils.py:208-256).  Here every builder is
//     key generation (one lane per tet-face / tet-edge incidence)
//  -> stable LSD radix sort of (key, insertion index)              [prims.hpp]
//  -> run detection + prefix sums + ordered emission               [hand-written kernels]
// which reproduces the reference's output ORDER as well as its content: std::map iterates
// keys ascending and each key's vector keeps insertion order == a stable sort by key of
// records generated in insertion order; Python dicts iterate in first-seen order == the
// order of each group's first record.
//
// Integer work only: results are bit-exact with the oracle / oracle/_ref.
#include <cstring>

#include "common.hpp"

#include "prims.hpp"

namespace deftet {
namespace bld {

using u64 = unsigned long long;
using u32 = unsigned int;

__constant__ int kFaceIdx[4][3] = {{0, 1, 2}, {1, 0, 3}, {2, 3, 0}, {3, 2, 1}};   // run.cpp:42-45

// face key = min*n^2 + max*n + mid (note: max before mid), tet_adj_share/run.cpp:57-68:
// `c` starts as the third vertex and is overwritten by every vertex that is neither min nor max
__device__ __forceinline__ u64 face_key_share(int v0, int v1, int v2, u64 n)
{
    int a = min(v0, min(v1, v2)), b = max(v0, max(v1, v2)), c = v2;
    if (a != v0 && b != v0) c = v0;
    if (a != v1 && b != v1) c = v1;
    if (a != v2 && b != v2) c = v2;
    return (u64)a * n * n + (u64)b * n + (u64)c;
}
// absolute face id of tet_face_adj/run.cpp:48-65: `c` is seeded with triangle[0]
__device__ __forceinline__ u64 face_key_native(int v0, int v1, int v2, u64 n)
{
    int a = min(v0, min(v1, v2)), b = max(v0, max(v1, v2)), c = v0;
    if (v0 != a && v0 != b) c = v0;
    if (v1 != a && v1 != b) c = v1;
    if (v2 != a && v2 != b) c = v2;
    return (u64)a * n * n + (u64)b * n + (u64)c;
}

__global__ __launch_bounds__(256) void k_face_keys(const int *__restrict__ tet, int T, u64 n, u64 *key, u32 *owner,
                                                   u64 *fkey_native)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // i = t*4 + f (insertion order)
    if (i >= T * 4) return;
    const int t = i >> 2, f = i & 3;
    const int4 v = reinterpret_cast<const int4 *>(tet)[t];
    const int vv[4] = {v.x, v.y, v.z, v.w};
    const int v0 = vv[kFaceIdx[f][0]], v1 = vv[kFaceIdx[f][1]], v2 = vv[kFaceIdx[f][2]];
    if (key) {
        key[i] = face_key_share(v0, v1, v2, n);
        owner[i] = (u32)i;
    }
    if (fkey_native) fkey_native[i] = face_key_native(v0, v1, v2, n);
}

// ---------------------------------------------------------------------------- A2
__global__ __launch_bounds__(256) void k_share_flag(const u64 *__restrict__ key, int n, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = key[i];
    const bool start = i == 0 || key[i - 1] != k;
    const bool two = start && i + 1 < n && key[i + 1] == k && (i + 2 >= n || key[i + 2] != k);   // f.size()==2, run.cpp:83
    flag[i] = two ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_share_emit(const int *__restrict__ flag, const int *__restrict__ pos,
                                                    const u32 *__restrict__ owner, int n, int *out, int *n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        const u32 o0 = owner[i], o1 = owner[i + 1];
        int *o = out + (size_t)pos[i] * 6;
        o[0] = o0 >> 2; o[1] = o1 >> 2; o[2] = o0 & 3;          // run.cpp:84-86
        o[3] = o1 >> 2; o[4] = o0 >> 2; o[5] = o1 & 3;          // run.cpp:88-90
    }
    if (i == n - 1) *n_out = pos[i] + flag[i];
}

// ---------------------------------------------------------------------------- A6
__global__ __launch_bounds__(256) void k_group_info(const u64 *__restrict__ key, const u32 *__restrict__ owner, int n,
                                                    int *gsize_by_owner, int *second_by_owner)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = key[i];
    if (i > 0 && key[i - 1] == k) return;                       // not a group head
    int len = 1;
    while (i + len < n && key[i + len] == k) ++len;
    const u32 o = owner[i];                                     // first owner (stable sort keeps insertion order)
    gsize_by_owner[o] = len;
    second_by_owner[o] = len >= 2 ? (int)owner[i + 1] : -1;
}

__global__ __launch_bounds__(256) void k_face_flags(const int *__restrict__ gsize, int n, int with_boundary, int *f_in,
                                                    int *f_bd, int *f_multi)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const int g = gsize[o];
    f_in[o] = (g == 2 || (with_boundary && g == 1)) ? 1 : 0;     // tet_utils.py:245-249 / prepare_for_wz.py:84-95
    f_bd[o] = g == 1 ? 1 : 0;                                    // tet_utils.py:250-252
    f_multi[o] = g > 2 ? 1 : 0;                                  // cnt_n_tet[2]
}

__global__ __launch_bounds__(256) void k_face_emit(const int *__restrict__ tet, const int *__restrict__ gsize,
                                                   const int *__restrict__ second, const int *__restrict__ f_in,
                                                   const int *__restrict__ p_in, const int *__restrict__ f_bd,
                                                   const int *__restrict__ p_bd, const int *__restrict__ f_multi,
                                                   const int *__restrict__ p_multi, int n, long long *face_fx3,
                                                   long long *tetidx_fx2, long long *tetfaceidx_fx2, long long *boundary_fx3,
                                                   int *counts)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;          // first-seen order == owner order
    if (o >= n) return;
    const int t = o >> 2, f = o & 3;
    if (f_in[o] || f_bd[o]) {
        const int4 v = reinterpret_cast<const int4 *>(tet)[t];
        const int vv[4] = {v.x, v.y, v.z, v.w};
        const long long tri[3] = {vv[kFaceIdx[f][0]], vv[kFaceIdx[f][1]], vv[kFaceIdx[f][2]]};   // as oriented in the first owner
        if (f_in[o]) {
            const size_t r = p_in[o];
            face_fx3[r * 3] = tri[0]; face_fx3[r * 3 + 1] = tri[1]; face_fx3[r * 3 + 2] = tri[2];
            const bool two = gsize[o] == 2;
            tetidx_fx2[r * 2] = t;
            tetidx_fx2[r * 2 + 1] = two ? second[o] >> 2 : -1;
            tetfaceidx_fx2[r * 2] = f;
            tetfaceidx_fx2[r * 2 + 1] = two ? second[o] & 3 : -1;
        }
        if (f_bd[o] && boundary_fx3) {
            const size_t r = p_bd[o];
            boundary_fx3[r * 3] = tri[0]; boundary_fx3[r * 3 + 1] = tri[1]; boundary_fx3[r * 3 + 2] = tri[2];
        }
    }
    if (o == n - 1) {
        counts[0] = p_in[o] + f_in[o];
        counts[1] = p_bd[o] + f_bd[o];
        counts[2] = p_multi[o] + f_multi[o];
    }
}

// ---------------------------------------------------------------------------- A3
__global__ __launch_bounds__(256) void k_edge_keys(const int *__restrict__ tet, int T, u64 n, int wrap32, u64 *key, u32 *ord)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;          // m = (t*4+f)*3 + e : push_back order, run.cpp:45
    if (m >= T * 12) return;
    const int e = m % 3, tf = m / 3, t = tf >> 2, f = tf & 3;
    const int4 v = reinterpret_cast<const int4 *>(tet)[t];
    const int vv[4] = {v.x, v.y, v.z, v.w};
    const int p = vv[kFaceIdx[f][e]], q = vv[kFaceIdx[f][(e + 1) % 3]];
    const int pa = min(p, q), pb = max(p, q);                     // run.cpp:37-38
    u64 k;
    if (wrap32) {
        const u32 w = (u32)pa * (u32)n + (u32)pb;                 // `int e = point_a * n_point + point_b`, run.cpp:39
        k = (u64)(w ^ 0x80000000u);                               // std::map<int> orders it as a signed int
    } else {
        k = (u64)pa * n + (u64)pb;                                // utils/tet_utils.py:176 (unbounded ints)
    }
    key[m] = k;
    ord[m] = (u32)m;
}

__device__ __forceinline__ void edge_run(const u64 *__restrict__ key, int n, int i, int &rs, int &re)
{
    const u64 k = key[i];
    rs = i;
    while (rs > 0 && key[rs - 1] == k) --rs;
    re = i + 1;
    while (re < n && key[re] == k) ++re;
}

__global__ __launch_bounds__(256) void k_edge_count(const u64 *__restrict__ key, const u32 *__restrict__ ord,
                                                    const u64 *__restrict__ fkey, int n, long long *cnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int rs, re;
    edge_run(key, n, i, rs, re);
    const u32 fa = ord[i] / 3;
    const u64 ka = fkey[fa];
    int c = 0;
    for (int y = rs; y < re; ++y) {
        const u32 fb = ord[y] / 3;
        if (fb != fa && fkey[fb] != ka) ++c;                      // run.cpp:79-80
    }
    cnt[i] = c;
}

__global__ __launch_bounds__(256) void k_edge_emit(const u64 *__restrict__ key, const u32 *__restrict__ ord,
                                                   const u64 *__restrict__ fkey, const long long *__restrict__ cnt,
                                                   const long long *__restrict__ pos, int n, long long capacity, int *out,
                                                   long long *n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *n_out = pos[i] + cnt[i];                     // total (may exceed capacity: caller checks)
    if (cnt[i] == 0) return;
    int rs, re;
    edge_run(key, n, i, rs, re);
    const u32 fa = ord[i] / 3;
    const u64 ka = fkey[fa];
    long long w = pos[i];
    for (int y = rs; y < re; ++y) {                               // inner loop of run.cpp:78
        const u32 fb = ord[y] / 3;
        if (fb != fa && fkey[fb] != ka) {
            if (w < capacity) { out[w * 2] = (int)fa; out[w * 2 + 1] = (int)fb; }   // run.cpp:81-82
            ++w;
        }
    }
}

// ---------------------------------------------------------------------------- A4
__global__ __launch_bounds__(256) void k_pt_keys(const int *__restrict__ tet, int T, u64 n, u64 *key)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= T * 12) return;
    const int t = m / 12, r = m % 12, i = r / 3, jj = r % 3, j = jj + (jj >= i ? 1 : 0);   // 12 ordered pairs i != j
    const int4 v = reinterpret_cast<const int4 *>(tet)[t];
    const int vv[4] = {v.x, v.y, v.z, v.w};
    key[m] = (u64)vv[i] * n + (u64)vv[j];                         // get_i, run.cpp:17-19
}

__global__ __launch_bounds__(256) void k_unique_flag(const u64 *__restrict__ key, int n, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || key[i - 1] != key[i]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_pt_emit(const u64 *__restrict__ key, const int *__restrict__ flag,
                                                 const int *__restrict__ pos, int n, u64 np, int *out, int *n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) {
        out[(size_t)pos[i] * 2] = (int)(key[i] / np);            // run.cpp:47
        out[(size_t)pos[i] * 2 + 1] = (int)(key[i] % np);        // run.cpp:48
    }
    if (i == n - 1) *n_out = pos[i] + flag[i];
}

// ---------------------------------------------------------------------------- A5
// Integer image of printf("%.5f", (double)x): sign bit kept even when the magnitude rounds
// to zero ("-0.00000" != "0.00000"); magnitude = round-half-even(|x| * 10^5) computed exactly
// from the binary value; |x| >= 2^24 (already an integer: the decimal string is injective on
// the float) and inf/nan get tagged raw-bit keys.
__device__ __forceinline__ u64 decimal_key(float x)
{
    const u32 bits = __float_as_uint(x);
    const u64 sign = (u64)(bits >> 31) << 63;
    const u32 ex = (bits >> 23) & 0xFF;
    const u32 man = bits & 0x7FFFFF;
    if (ex == 0xFF) return sign | (1ull << 62) | (man ? 1ull : 0ull);          // "inf" / "nan" ("-nan" keeps the sign)
    if (ex >= 127 + 24) return sign | (1ull << 61) | (u64)(bits & 0x7FFFFFFF);   // integer-valued floats
    // value = m * 2^e with m < 2^24
    const u64 m = ex ? (u64)(man | 0x800000) : (u64)man;
    const int e = (ex ? (int)ex : 1) - 127 - 23;                                // e <= 0 here (ex < 151 -> e < 1) or small positive
    const u64 N = m * 100000ull;                                               // < 2^41
    if (e >= 0) return sign | (N << e);                                        // e <= 0 in practice; kept for completeness
    const int s = -e;
    if (s >= 64) return sign;                                                  // N / 2^s < 2^-23: rounds to 0
    const u64 q = N >> s, rem = N & ((1ull << s) - 1), half = 1ull << (s - 1);
    u64 r = q;
    if (rem > half || (rem == half && (q & 1))) ++r;                           // round half to even (glibc printf)
    return sign | r;
}

__global__ __launch_bounds__(256) void k_dec_keys(const float *__restrict__ pts, int n, u64 *kx, u64 *ky, u64 *kz, u32 *idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    kx[i] = decimal_key(pts[i * 3]);
    ky[i] = decimal_key(pts[i * 3 + 1]);
    kz[i] = decimal_key(pts[i * 3 + 2]);
    idx[i] = (u32)i;
}

__global__ __launch_bounds__(256) void k_gather_u64(const u64 *__restrict__ src, const u32 *__restrict__ idx, int n, u64 *dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// head position of each group in sorted order (0 for non-heads), to be max-scanned
__global__ __launch_bounds__(256) void k_colaps_heads(const u64 *__restrict__ kx, const u64 *__restrict__ ky,
                                                      const u64 *__restrict__ kz, const u32 *__restrict__ idx, int n,
                                                      int *headpos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool head = i == 0;
    if (!head) {
        const u32 a = idx[i], b = idx[i - 1];
        head = kx[a] != kx[b] || ky[a] != ky[b] || kz[a] != kz[b];
    }
    headpos[i] = head ? i : 0;
}

__global__ __launch_bounds__(256) void k_colaps_first(const int *__restrict__ headpos_scanned, const u32 *__restrict__ idx,
                                                      int n, int *first, int *isfirst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 me = idx[i], hd = idx[headpos_scanned[i]];          // stable sorts: the head holds the smallest index
    first[me] = (int)hd;
    isfirst[me] = me == hd ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_colaps_emit(const int *__restrict__ first, const int *__restrict__ isfirst,
                                                     const int *__restrict__ newid, int n, int *map_array, int *inverse_idx,
                                                     int *n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    map_array[i] = newid[first[i]];                               // run.cpp:49,53
    if (isfirst[i]) inverse_idx[newid[i]] = i;                    // run.cpp:48
    if (i == n - 1) *n_out = newid[i] + isfirst[i];
}

// ---------------------------------------------------------------------------- host helpers
// ---------------------------------------------------------------------------- N3
// Render-side geometry rebuilds, diff_render/diftet_6_subdiv/3_model/prepare_for_wz.py:
// generate_edge (:184-203), generate_tet_edge_idx (:223-236, the O(E*T) matchedgelist :206-221
// becomes one sort), generate_subdivision (:255-301), generate_point_adj_idx (:108-146, the
// dense P x P matrix becomes the sorted pair list of A4), delete_tet (:171-180),
// tetweights2tetneighbourweights (3_model/deftet.py:316-331).
__constant__ int kEdgeEnds[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};   // edges_all_connect_6x2, :190

// incidence m = t*6 + e: key = min*n + max (np.unique(axis=0) sorts rows lexicographically = by this key)
__global__ __launch_bounds__(256) void k_uedge_keys(const long long *__restrict__ tet, int T, u64 n, u64 *key, u32 *inc, int *bad)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= T * 6) return;
    const int t = m / 6, e = m % 6;
    const long long a = tet[(size_t)t * 4 + kEdgeEnds[e][0]], b = tet[(size_t)t * 4 + kEdgeEnds[e][1]];
    if (a < 0 || b < 0 || a >= (long long)n || b >= (long long)n) *bad = 1;      // caller raises; keys stay defined
    const u64 lo = (u64)(a < b ? a : b), hi = (u64)(a < b ? b : a);
    key[m] = lo * n + hi;
    inc[m] = (u32)m;
}

__global__ __launch_bounds__(256) void k_uedge_emit(const u64 *__restrict__ key, const u32 *__restrict__ inc,
                                                    const int *__restrict__ flag, const int *__restrict__ pos, int n, u64 np,
                                                    long long *edges, long long *tet_edge, int *n_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = pos[i] + flag[i] - 1;                          // index of this incidence's edge in the unique list
    if (flag[i]) {
        edges[(size_t)u * 2] = (long long)(key[i] / np);
        edges[(size_t)u * 2 + 1] = (long long)(key[i] % np);
    }
    tet_edge[inc[i]] = u;                                        // [T,6] row-major: inc = t*6 + e
    if (i == n - 1) *n_out = u + 1;
}

// new vertices: first the old ones, then one midpoint per edge, (a + b) / 2 in fp32 (:238-252)
__global__ __launch_bounds__(256) void k_subdiv_points(const float *__restrict__ src, const long long *__restrict__ edges, int P, int E,
                                                       int K, float *dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)(P + E) * K) return;
    const int r = (int)(i / K), c = (int)(i % K);
    if (r < P) {
        dst[i] = src[i];
    } else {
        const long long a = edges[(size_t)(r - P) * 2], b = edges[(size_t)(r - P) * 2 + 1];
        dst[i] = (src[(size_t)a * K + c] + src[(size_t)b * K + c]) / 2.0f;
    }
}

__global__ __launch_bounds__(256) void k_subdiv_flags(const unsigned char *__restrict__ sig, int T, int *keepOld, int *split)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int s = sig ? (sig[t] != 0) : 1;
    keepOld[t] = 1 - s;
    split[t] = s;
}

// the eight children of a tet (:270-281) in the reference's order; unsplit tets are copied first
__global__ __launch_bounds__(256) void k_subdiv_tets(const long long *__restrict__ tet, const long long *__restrict__ tet_edge,
                                                     const unsigned char *__restrict__ sig, const int *__restrict__ posOld,
                                                     const int *__restrict__ posSplit, int T, int P, long long *out, int *n_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int s = sig ? (sig[t] != 0) : 1;
    const int nOld = sig ? posOld[T - 1] + (sig[T - 1] ? 0 : 1) : 0;
    if (t == T - 1) *n_out = nOld + 8 * (posSplit[T - 1] + s);
    const long long a = tet[(size_t)t * 4], b = tet[(size_t)t * 4 + 1], c = tet[(size_t)t * 4 + 2], d = tet[(size_t)t * 4 + 3];
    if (!s) {
        long long *o = out + (size_t)posOld[t] * 4;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d;
        return;
    }
    const long long *te = tet_edge + (size_t)t * 6;
    const long long ab = te[0] + P, ac = te[1] + P, ad = te[2] + P, bc = te[3] + P, bd = te[4] + P, cd = te[5] + P;
    const long long ch[8][4] = {{a, ab, ac, ad}, {b, bc, ab, bd}, {c, ac, bc, cd}, {d, ad, cd, bd},
                                {ab, ac, ad, bd}, {ab, ac, bd, bc}, {cd, ac, bd, ad}, {cd, ac, bc, bd}};
    long long *o = out + ((size_t)nOld + (size_t)posSplit[t] * 8) * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        o[k * 4] = ch[k][0]; o[k * 4 + 1] = ch[k][1]; o[k * 4 + 2] = ch[k][2]; o[k * 4 + 3] = ch[k][3];
    }
}

// point adjacency table from the sorted unique ordered pairs (i, j) of A4: row starts by key boundaries
__global__ __launch_bounds__(256) void k_adj_rowstart(const int *__restrict__ pairs, int n, int P, int *rowStart)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const int prev = i == 0 ? -1 : pairs[(size_t)(i - 1) * 2];
    const int cur = i == n ? P : pairs[(size_t)i * 2];
    for (int k = prev + 1; k <= cur; ++k) rowStart[k] = i;
}

__global__ __launch_bounds__(256) void k_adj_degree(const int *__restrict__ rowStart, int P, float *adjsum, int *maxDeg)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    int d = 0;
    if (p < P) {
        d = rowStart[p + 1] - rowStart[p];
        adjsum[p] = (float)d;                                     // np.sum of a float32 0/1 row, :139
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
    if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(maxDeg, d);
}

__global__ __launch_bounds__(256) void k_adj_fill(const int *__restrict__ pairs, const int *__restrict__ rowStart, int P, int M,
                                                  long long *table)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)P * M) return;
    const int p = (int)(i / M), c = (int)(i % M);
    const int s = rowStart[p], d = rowStart[p + 1] - s;
    table[i] = c < d ? (long long)pairs[(size_t)(s + c) * 2 + 1] : -1;   // ascending neighbours, then -1 (:141-145)
}

// delete_tet: keep = (max over the row, NaN-propagating like np.max) > thres
__global__ __launch_bounds__(256) void k_delete_flags(const float *__restrict__ w, int T, int K, float thres, int *keep)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float m = -INFINITY;
    bool nan = false;
    for (int k = 0; k < K; ++k) {
        const float x = w[(size_t)t * K + k];
        nan = nan || (x != x);
        m = fmaxf(m, x);
    }
    keep[t] = (!nan && K > 0 && m > thres) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_compact_tets(const long long *__restrict__ tet, const int *__restrict__ keep,
                                                      const int *__restrict__ pos, int T, long long *out, int *n_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (keep[t]) {
        long long *o = out + (size_t)pos[t] * 4;
        const long long *s = tet + (size_t)t * 4;
        o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
    }
    if (t == T - 1) *n_out = pos[t] + keep[t];
}

// one level of tetweights2tetneighbourweights: out[t, j*K + c] = w[nei[t,j], c], zero row for nei = -1
__global__ __launch_bounds__(256) void k_neighbour_weights(const float *__restrict__ w, const long long *__restrict__ nei, int T, int K,
                                                           float *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)T * 4 * K) return;
    const int c = (int)(i % K);
    const long long tj = i / K;
    const long long n = nei[tj];
    out[i] = (n >= 0 && n < T) ? w[(size_t)n * K + c] : 0.f;
}

static int key_bits(u64 max_key)
{
    int b = 1;
    while (b < 64 && (max_key >> b)) ++b;
    return b;
}

// temporary storage of the sorts and scans (prims.hpp) is carved from the tail of the caller's workspace
struct Ws {
    Arena A;
    Ws(void *p, size_t n) : A(p, n) {}
    void *tail() { A.off = align_up(A.off, 256); return A.base + A.off; }
    size_t left() const { return A.cap > A.off ? A.cap - A.off : 0; }
};

template <typename V>
static int sort_pairs(Ws &W, const u64 *kin, u64 *kout, const V *vin, V *vout, size_t n, int bits, hipStream_t st)
{
    void *tmp = W.tail();
    return prims::radix_sort<u64, V>(kin, kout, vin, vout, n, bits, tmp, W.left(), st);
}

static int sort_keys(Ws &W, const u64 *kin, u64 *kout, size_t n, int bits, hipStream_t st)
{
    void *tmp = W.tail();
    return prims::radix_sort_keys<u64>(kin, kout, n, bits, tmp, W.left(), st);
}

template <typename T>
static int ex_scan(Ws &W, const T *in, T *out, size_t n, hipStream_t st)
{
    void *tmp = W.tail();
    return prims::scan<T, prims::Plus, true>(in, out, n, T(0), prims::Plus(), tmp, W.left(), st);
}

static int max_scan(Ws &W, const int *in, int *out, size_t n, hipStream_t st)
{
    void *tmp = W.tail();
    return prims::scan<int, prims::Max, false>(in, out, n, (int)0x80000000, prims::Max(), tmp, W.left(), st);
}

#define TRY(x)                     \
    do {                           \
        int s_ = (x);              \
        if (s_ != DEFTET_OK) return s_; \
    } while (0)

static int check_common(const void *tet, int n_point, int n_tet, void *ws)
{
    DEFTET_CHECK_ARG(n_tet >= 0 && n_point >= 0, "negative size");
    DEFTET_CHECK_ARG(n_point <= 2097151, "n_point=%d: face keys min*n^2+max*n+mid would overflow 63 bits", n_point);
    DEFTET_CHECK_ARG(n_tet <= 100000000, "n_tet too large");
    DEFTET_CHECK_ARG(n_tet == 0 || tet, "null tet_list");
    DEFTET_CHECK_ARG(n_tet == 0 || (((uintptr_t)tet & 15) == 0), "tet_list must be 16-byte aligned");
    DEFTET_CHECK_ARG(n_tet == 0 || (ws && ((uintptr_t)ws & 255) == 0), "workspace null or not 256-byte aligned");
    return DEFTET_OK;
}

static inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + 255) / 256)); }


// ------------------------------------------------------------------------------------
// Per-tet neighbour table and per-tet-face owner table from the unique-face table of
// deftet_tet_to_face_i32(with_boundary=1):
//   tet_neighbour_idx [T,4]  diff_render/diftet_6_subdiv/3_model/utils_tetsv.py:41-58 — the shared faces are
//       visited in first-seen order and each appends the partner to both owners' rows, so a tet's row lists
//       its partners in ascending order of the shared face's position in the unique-face table; -1 padded;
//   tet_face_tetidx [4T,2]   utils/tet_utils.py:259-300 (tet_to_face_withtet): for global face 4t+i the owners
//       of its key in insertion order, a lone owner padded with 0.
// One lane per unique face scatters into per-(tet, local face) slots (each slot belongs to exactly one face
// key), then one lane per tet orders its four slots with a sorting network.
// ------------------------------------------------------------------------------------
// slot filler for the face position = 0x7F7F7F7F (memset pattern): sorts after every real face position

__global__ __launch_bounds__(256) void k_owner_scatter(const long long *__restrict__ tetidx_fx2, const long long *__restrict__ tetfaceidx_fx2,
                                                       int F, int *slotFace, int *slotNbr, long long *withtet_4tx2)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const long long t0 = tetidx_fx2[2 * f], t1 = tetidx_fx2[2 * f + 1];
    const long long l0 = tetfaceidx_fx2[2 * f], l1 = tetfaceidx_fx2[2 * f + 1];
    const long long g0 = t0 * 4 + l0;
    if (t1 >= 0) {
        const long long g1 = t1 * 4 + l1;
        slotFace[g0] = f; slotNbr[g0] = (int)t1;
        slotFace[g1] = f; slotNbr[g1] = (int)t0;
        if (withtet_4tx2) {
            withtet_4tx2[2 * g0] = t0; withtet_4tx2[2 * g0 + 1] = t1;
            withtet_4tx2[2 * g1] = t0; withtet_4tx2[2 * g1 + 1] = t1;
        }
    } else if (withtet_4tx2) {                                           // boundary face: single owner, padded with 0 (:293-294)
        withtet_4tx2[2 * g0] = t0; withtet_4tx2[2 * g0 + 1] = 0;
    }
}

__global__ __launch_bounds__(256) void k_neighbour_rows(const int *__restrict__ slotFace, const int *__restrict__ slotNbr, int T,
                                                        long long *nbr_tx4)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int4 f4 = reinterpret_cast<const int4 *>(slotFace)[t], n4 = reinterpret_cast<const int4 *>(slotNbr)[t];
    // order the four slots by (position of the face in the unique-face table, local face id); the local id only
    // matters when one tet owns both sides of a face (repeated vertices): the reference then appends the tet to
    // its own row twice in a row
    unsigned long long key[4] = {((unsigned long long)(unsigned)f4.x << 2) | 0u, ((unsigned long long)(unsigned)f4.y << 2) | 1u,
                                 ((unsigned long long)(unsigned)f4.z << 2) | 2u, ((unsigned long long)(unsigned)f4.w << 2) | 3u};
    int nb[4] = {n4.x, n4.y, n4.z, n4.w};
#define DEFTET_CSWAP_KV(a, b)                                                                     \
    if (key[b] < key[a]) {                                                                        \
        const unsigned long long tk_ = key[a]; key[a] = key[b]; key[b] = tk_;                     \
        const int tn_ = nb[a]; nb[a] = nb[b]; nb[b] = tn_;                                        \
    }
    DEFTET_CSWAP_KV(0, 1) DEFTET_CSWAP_KV(2, 3) DEFTET_CSWAP_KV(0, 2) DEFTET_CSWAP_KV(1, 3) DEFTET_CSWAP_KV(1, 2)
#undef DEFTET_CSWAP_KV
#pragma unroll
    for (int i = 0; i < 4; ++i) nbr_tx4[(size_t)t * 4 + i] = (long long)nb[i];     // unfilled slots hold -1 (memset 0xFF) and sort last
}

}  // namespace bld
}  // namespace deftet

using namespace deftet;
using namespace deftet::bld;

// Upper bound of what any builder needs for (n_point, n_tet): our arrays over 12T records
// plus the sorts' spare buffers and digit tables (prims.hpp).
extern "C" size_t deftet_builder_workspace_bytes(int n_point, int n_tet)
{
    size_t n = (size_t)(n_tet > 0 ? n_tet : 0) * 12 + (size_t)(n_point > 0 ? n_point : 0) + 1024;
    return n * 96 + ((size_t)8 << 20);
}

extern "C" int deftet_tet_adj_share_i32(const int32_t *tet, int32_t *out_rows, int32_t *n_out, int n_point, int T,
                                        void *workspace, size_t wsb, void *stream_)
{
    TRY(check_common(tet, n_point, T, workspace));
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(n_out, "null n_out");
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_out, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(out_rows, "null out_rows");
    const size_t n = (size_t)T * 4;
    Ws W(workspace, wsb);
    u64 *key = W.A.take<u64>(n), *skey = W.A.take<u64>(n);
    u32 *own = W.A.take<u32>(n), *sown = W.A.take<u32>(n);
    int *flag = W.A.take<int>(n), *pos = W.A.take<int>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    const u64 np = (u64)n_point;
    DEFTET_LAUNCH(k_face_keys, grid_for(n), dim3(256), st, tet, T, np, key, own, (u64 *)nullptr);
    TRY(sort_pairs(W, key, skey, own, sown, n, key_bits(np * np * np), st));
    DEFTET_LAUNCH(k_share_flag, grid_for(n), dim3(256), st, skey, (int)n, flag);
    TRY(ex_scan(W, flag, pos, n, st));
    DEFTET_LAUNCH(k_share_emit, grid_for(n), dim3(256), st, flag, pos, sown, (int)n, out_rows, n_out);
    return DEFTET_OK;
}

extern "C" int deftet_tet_to_face_i32(const int32_t *tet, int64_t *face_fx3, int64_t *tetidx_fx2, int64_t *tetfaceidx_fx2,
                                      int64_t *boundary_fx3, int32_t *counts, int n_point, int T, int with_boundary,
                                      void *workspace, size_t wsb, void *stream_)
{
    TRY(check_common(tet, n_point, T, workspace));
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(counts, "null counts");
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(counts, 0, 12, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(face_fx3 && tetidx_fx2 && tetfaceidx_fx2, "null output");
    const size_t n = (size_t)T * 4;
    Ws W(workspace, wsb);
    u64 *key = W.A.take<u64>(n), *skey = W.A.take<u64>(n);
    u32 *own = W.A.take<u32>(n), *sown = W.A.take<u32>(n);
    int *gsize = W.A.take<int>(n), *second = W.A.take<int>(n);
    int *f_in = W.A.take<int>(n), *f_bd = W.A.take<int>(n), *f_mu = W.A.take<int>(n);
    int *p_in = W.A.take<int>(n), *p_bd = W.A.take<int>(n), *p_mu = W.A.take<int>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    const u64 np = (u64)n_point;
    DEFTET_LAUNCH(k_face_keys, grid_for(n), dim3(256), st, tet, T, np, key, own, (u64 *)nullptr);
    TRY(sort_pairs(W, key, skey, own, sown, n, key_bits(np * np * np), st));
    DEFTET_HIP(hipMemsetAsync(gsize, 0, n * 4, st));
    DEFTET_LAUNCH(k_group_info, grid_for(n), dim3(256), st, skey, sown, (int)n, gsize, second);
    DEFTET_LAUNCH(k_face_flags, grid_for(n), dim3(256), st, gsize, (int)n, with_boundary, f_in, f_bd, f_mu);
    TRY(ex_scan(W, f_in, p_in, n, st));
    TRY(ex_scan(W, f_bd, p_bd, n, st));
    TRY(ex_scan(W, f_mu, p_mu, n, st));
    DEFTET_LAUNCH(k_face_emit, grid_for(n), dim3(256), st, tet, gsize, second, f_in, p_in, f_bd, p_bd, f_mu, p_mu, (int)n,
                  (long long *)face_fx3, (long long *)tetidx_fx2, (long long *)tetfaceidx_fx2, (long long *)boundary_fx3,
                  counts);
    return DEFTET_OK;
}

extern "C" size_t deftet_tet_neighbours_workspace_bytes(int n_tet)
{
    return align_up((size_t)(n_tet > 0 ? n_tet : 0) * 16, 256) * 2 + 256;
}

extern "C" int deftet_tet_neighbours_i64(const int64_t *tetidx_fx2, const int64_t *tetfaceidx_fx2, int n_face, int T,
                                         int64_t *nbr_tx4, int64_t *withtet_4tx2, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(n_face >= 0 && T >= 0, "negative size");
    if (T == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(nbr_tx4, "null nbr_tx4");
    DEFTET_CHECK_ARG(n_face == 0 || (tetidx_fx2 && tetfaceidx_fx2), "null face table");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && wsb >= deftet_tet_neighbours_workspace_bytes(T), "workspace null, misaligned or too small");
    hipStream_t st = as_stream(stream_);
    Arena A(workspace, wsb);
    int *slotFace = A.take<int>((size_t)T * 4), *slotNbr = A.take<int>((size_t)T * 4);
    DEFTET_HIP(hipMemsetAsync(slotFace, 0x7F, (size_t)T * 16, st));
    DEFTET_HIP(hipMemsetAsync(slotNbr, 0xFF, (size_t)T * 16, st));
    if (n_face > 0)
        DEFTET_LAUNCH(k_owner_scatter, grid_for((size_t)n_face), dim3(256), st, (const long long *)tetidx_fx2, (const long long *)tetfaceidx_fx2,
                      n_face, slotFace, slotNbr, (long long *)withtet_4tx2);
    DEFTET_LAUNCH(k_neighbour_rows, grid_for((size_t)T), dim3(256), st, slotFace, slotNbr, T, (long long *)nbr_tx4);
    return DEFTET_OK;
}

extern "C" int deftet_tet_face_adj_i32(const int32_t *tet, int32_t *out_rows, long long capacity, long long *n_out,
                                       int n_point, int T, int wrap32, void *workspace, size_t wsb, void *stream_)
{
    TRY(check_common(tet, n_point, T, workspace));
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(n_out && capacity >= 0, "null n_out / negative capacity");
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_out, 0, 8, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(out_rows || capacity == 0, "null out_rows");
    const size_t n = (size_t)T * 12, nf = (size_t)T * 4;
    Ws W(workspace, wsb);
    u64 *key = W.A.take<u64>(n), *skey = W.A.take<u64>(n), *fkey = W.A.take<u64>(nf);
    u32 *ord = W.A.take<u32>(n), *sord = W.A.take<u32>(n);
    long long *cnt = W.A.take<long long>(n), *pos = W.A.take<long long>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    const u64 np = (u64)n_point;
    DEFTET_LAUNCH(k_face_keys, grid_for(nf), dim3(256), st, tet, T, np, (u64 *)nullptr, (u32 *)nullptr, fkey);
    DEFTET_LAUNCH(k_edge_keys, grid_for(n), dim3(256), st, tet, T, np, wrap32, key, ord);
    TRY(sort_pairs(W, key, skey, ord, sord, n, wrap32 ? 32 : key_bits(np * np), st));
    DEFTET_LAUNCH(k_edge_count, grid_for(n), dim3(256), st, skey, sord, fkey, (int)n, cnt);
    TRY(ex_scan(W, cnt, pos, n, st));
    DEFTET_LAUNCH(k_edge_emit, grid_for(n), dim3(256), st, skey, sord, fkey, cnt, pos, (int)n, capacity, out_rows, n_out);
    return DEFTET_OK;
}

extern "C" int deftet_tet_point_adj_i32(const int32_t *tet, int32_t *out_edges, int32_t *n_out, int n_point, int T,
                                        void *workspace, size_t wsb, void *stream_)
{
    TRY(check_common(tet, n_point, T, workspace));
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(n_out, "null n_out");
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_out, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(out_edges, "null out_edges");
    const size_t n = (size_t)T * 12;
    Ws W(workspace, wsb);
    u64 *key = W.A.take<u64>(n), *skey = W.A.take<u64>(n);
    int *flag = W.A.take<int>(n), *pos = W.A.take<int>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    const u64 np = (u64)n_point;
    DEFTET_LAUNCH(k_pt_keys, grid_for(n), dim3(256), st, tet, T, np, key);
    TRY(sort_keys(W, key, skey, n, key_bits(np * np), st));
    DEFTET_LAUNCH(k_unique_flag, grid_for(n), dim3(256), st, skey, (int)n, flag);
    TRY(ex_scan(W, flag, pos, n, st));
    DEFTET_LAUNCH(k_pt_emit, grid_for(n), dim3(256), st, skey, flag, pos, (int)n, np, out_edges, n_out);
    return DEFTET_OK;
}

extern "C" int deftet_colaps_v_f32(const float *pts, int32_t *map_array, int32_t *inverse_idx, int32_t *n_out, int N,
                                   void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(N >= 0, "negative size");
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(n_out, "null n_out");
    if (N == 0) { DEFTET_HIP(hipMemsetAsync(n_out, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(pts && map_array && inverse_idx, "null pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or misaligned");
    const size_t n = (size_t)N;
    Ws W(workspace, wsb);
    u64 *kx = W.A.take<u64>(n), *ky = W.A.take<u64>(n), *kz = W.A.take<u64>(n), *ka = W.A.take<u64>(n), *kb = W.A.take<u64>(n);
    u32 *i0 = W.A.take<u32>(n), *i1 = W.A.take<u32>(n);
    int *hp = W.A.take<int>(n), *hps = W.A.take<int>(n), *first = W.A.take<int>(n), *isf = W.A.take<int>(n), *nid = W.A.take<int>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    DEFTET_LAUNCH(k_dec_keys, grid_for(n), dim3(256), st, pts, N, kx, ky, kz, i0);
    // LSD over the three coordinate keys (stable): z, then y, then x
    TRY(sort_pairs(W, kz, kb, i0, i1, n, 64, st));
    DEFTET_LAUNCH(k_gather_u64, grid_for(n), dim3(256), st, ky, i1, N, ka);
    TRY(sort_pairs(W, ka, kb, i1, i0, n, 64, st));
    DEFTET_LAUNCH(k_gather_u64, grid_for(n), dim3(256), st, kx, i0, N, ka);
    TRY(sort_pairs(W, ka, kb, i0, i1, n, 64, st));
    DEFTET_LAUNCH(k_colaps_heads, grid_for(n), dim3(256), st, kx, ky, kz, i1, N, hp);
    TRY(max_scan(W, hp, hps, n, st));
    DEFTET_LAUNCH(k_colaps_first, grid_for(n), dim3(256), st, hps, i1, N, first, isf);
    TRY(ex_scan(W, isf, nid, n, st));
    DEFTET_LAUNCH(k_colaps_emit, grid_for(n), dim3(256), st, first, isf, nid, N, map_array, inverse_idx, n_out);
    return DEFTET_OK;
}

// ---------------------------------------------------------------------------------------------
// N3 entry points
// ---------------------------------------------------------------------------------------------
static int check_i64(const void *tet, int n_point, int n_tet, void *ws)
{
    DEFTET_CHECK_ARG(n_tet >= 0 && n_point >= 0, "negative size");
    DEFTET_CHECK_ARG(n_point <= 2000000000 && n_tet <= 100000000, "size too large");
    DEFTET_CHECK_ARG(n_tet == 0 || tet, "null tet list");
    DEFTET_CHECK_ARG(n_tet == 0 || (ws && ((uintptr_t)ws & 255) == 0), "workspace null or not 256-byte aligned");
    return DEFTET_OK;
}

extern "C" int deftet_tet_edges_i64(const int64_t *tet, int64_t *edges_ex2, int64_t *tet_edge_tx6, int32_t *n_edge, int32_t *bad_flag,
                                    int n_point, int T, void *workspace, size_t wsb, void *stream_)
{
    TRY(check_i64(tet, n_point, T, workspace));
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(n_edge && bad_flag, "null n_edge / bad_flag");
    DEFTET_HIP(hipMemsetAsync(bad_flag, 0, 4, st));
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_edge, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(edges_ex2 && tet_edge_tx6, "null output");
    const size_t n = (size_t)T * 6;
    Ws W(workspace, wsb);
    u64 *key = W.A.take<u64>(n), *skey = W.A.take<u64>(n);
    u32 *inc = W.A.take<u32>(n), *sinc = W.A.take<u32>(n);
    int *flag = W.A.take<int>(n), *pos = W.A.take<int>(n);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    const u64 np = (u64)(n_point > 0 ? n_point : 1);
    DEFTET_LAUNCH(k_uedge_keys, grid_for(n), dim3(256), st, (const long long *)tet, T, np, key, inc, bad_flag);
    TRY(sort_pairs(W, key, skey, inc, sinc, n, key_bits(np * np), st));
    DEFTET_LAUNCH(k_unique_flag, grid_for(n), dim3(256), st, skey, (int)n, flag);
    TRY(ex_scan(W, flag, pos, n, st));
    DEFTET_LAUNCH(k_uedge_emit, grid_for(n), dim3(256), st, skey, sinc, flag, pos, (int)n, np, (long long *)edges_ex2,
                  (long long *)tet_edge_tx6, n_edge);
    return DEFTET_OK;
}

extern "C" int deftet_subdivide_f32(const int64_t *tet, const int64_t *tet_edge_tx6, const int64_t *edges_ex2, const float *points,
                                    const float *feat, const uint8_t *subdiv_sig, float *points_new, float *feat_new,
                                    int64_t *tet_new, int32_t *n_tet_new, int n_point, int T, int n_edge, int n_feat,
                                    void *workspace, size_t wsb, void *stream_)
{
    TRY(check_i64(tet, n_point, T, workspace));
    DEFTET_CHECK_ARG(n_edge >= 0 && n_feat >= 0, "negative size");
    DEFTET_CHECK_ARG(n_tet_new, "null n_tet_new");
    hipStream_t st = as_stream(stream_);
    if (n_point + n_edge > 0) {
        DEFTET_CHECK_ARG(points && points_new && (n_edge == 0 || edges_ex2), "null point arrays");
        DEFTET_LAUNCH(k_subdiv_points, grid_for((size_t)(n_point + n_edge) * 3), dim3(256), st, points, (const long long *)edges_ex2,
                      n_point, n_edge, 3, points_new);
        if (n_feat > 0) {
            DEFTET_CHECK_ARG(feat && feat_new, "null feature arrays");
            DEFTET_LAUNCH(k_subdiv_points, grid_for((size_t)(n_point + n_edge) * n_feat), dim3(256), st, feat,
                          (const long long *)edges_ex2, n_point, n_edge, n_feat, feat_new);
        }
    }
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_tet_new, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(tet_edge_tx6 && tet_new, "null tet arrays");
    Ws W(workspace, wsb);
    int *keepOld = W.A.take<int>(T), *split = W.A.take<int>(T), *posOld = W.A.take<int>(T), *posSplit = W.A.take<int>(T);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    DEFTET_LAUNCH(k_subdiv_flags, grid_for(T), dim3(256), st, subdiv_sig, T, keepOld, split);
    TRY(ex_scan(W, keepOld, posOld, (size_t)T, st));
    TRY(ex_scan(W, split, posSplit, (size_t)T, st));
    DEFTET_LAUNCH(k_subdiv_tets, grid_for(T), dim3(256), st, (const long long *)tet, (const long long *)tet_edge_tx6, subdiv_sig,
                  posOld, posSplit, T, n_point, (long long *)tet_new, n_tet_new);
    return DEFTET_OK;
}

extern "C" int deftet_point_adj_table_i64(const int32_t *pairs_nx2, int n_pairs, int n_point, int64_t *table_pxm, int width,
                                          float *adjsum_px1, int32_t *max_degree, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(n_pairs >= 0 && n_point >= 0 && width >= 0, "negative size");
    DEFTET_CHECK_ARG(n_pairs == 0 || pairs_nx2, "null pairs");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && wsb >= ((size_t)n_point + 1) * 4, "workspace null, misaligned or too small");
    hipStream_t st = as_stream(stream_);
    int *rowStart = static_cast<int *>(workspace);
    DEFTET_LAUNCH(k_adj_rowstart, grid_for((size_t)n_pairs + 1), dim3(256), st, pairs_nx2, n_pairs, n_point, rowStart);
    if (width == 0) {                                             // phase A: degrees and the table width
        DEFTET_CHECK_ARG(max_degree && (n_point == 0 || adjsum_px1), "null adjsum / max_degree");
        DEFTET_HIP(hipMemsetAsync(max_degree, 0, 4, st));
        if (n_point > 0) DEFTET_LAUNCH(k_adj_degree, grid_for(n_point), dim3(256), st, rowStart, n_point, adjsum_px1, max_degree);
    } else if (n_point > 0) {                                     // phase B: the padded table
        DEFTET_CHECK_ARG(table_pxm, "null table");
        DEFTET_LAUNCH(k_adj_fill, grid_for((size_t)n_point * width), dim3(256), st, pairs_nx2, rowStart, n_point, width,
                      (long long *)table_pxm);
    }
    return DEFTET_OK;
}

extern "C" int deftet_delete_tet_i64(const int64_t *tet, const float *weights_txk, float thres, int64_t *tet_kept, int32_t *n_kept,
                                     int T, int K, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(T >= 0 && K >= 0 && n_kept, "bad argument");
    hipStream_t st = as_stream(stream_);
    if (T == 0) { DEFTET_HIP(hipMemsetAsync(n_kept, 0, 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(tet && tet_kept && (K == 0 || weights_txk), "null pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "workspace null or not 256-byte aligned");
    Ws W(workspace, wsb);
    int *keep = W.A.take<int>(T), *pos = W.A.take<int>(T);
    DEFTET_CHECK_ARG(W.A.ok(), "workspace too small");
    DEFTET_LAUNCH(k_delete_flags, grid_for(T), dim3(256), st, weights_txk, T, K, thres, keep);
    TRY(ex_scan(W, keep, pos, (size_t)T, st));
    DEFTET_LAUNCH(k_compact_tets, grid_for(T), dim3(256), st, (const long long *)tet, keep, pos, T, (long long *)tet_kept, n_kept);
    return DEFTET_OK;
}

extern "C" int deftet_tet_neighbour_weights_f32(const float *weights_txk, const int64_t *nei_tx4, float *out_tx4k, int T, int K,
                                                void *stream_)
{
    DEFTET_CHECK_ARG(T >= 0 && K >= 0, "negative size");
    if (T == 0 || K == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(weights_txk && nei_tx4 && out_tx4k, "null pointer");
    DEFTET_LAUNCH(k_neighbour_weights, grid_for((size_t)T * 4 * K), dim3(256), as_stream(stream_), weights_txk,
                  (const long long *)nei_tx4, T, K, out_tx4k);
    return DEFTET_OK;
}

// ---------------------------------------------------------------------------- host-pointer variants
// Same argument lists as the reference's `extern "C" void run(...)` (utils/lib/*/run.cpp), plus
// an int status.  Synchronous: data comes from and returns to host memory.
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(&p, n ? n : 1) == hipSuccess ? 0 : -1; }
};
}  // namespace

#define HOST_PROLOGUE(T_)                                                                      \
    DEFTET_CHECK_ARG((T_) >= 0 && n_point >= 0, "negative size");                              \
    DevBuf d_tet, d_ws;                                                                        \
    const size_t wsb = deftet_builder_workspace_bytes(n_point, (T_));                          \
    if (d_tet.alloc((size_t)(T_) * 16) || d_ws.alloc(wsb)) return set_error(DEFTET_ELAUNCH, "hipMalloc failed"); \
    DEFTET_HIP(hipMemcpy(d_tet.p, tet_list, (size_t)(T_) * 16, hipMemcpyHostToDevice));

extern "C" int deftet_tet_adj_share_host(int *tet_list, int *face_edge_p, int *n_face_edge_p, int n_point, int n_tet)
{
    HOST_PROLOGUE(n_tet)
    DevBuf d_out, d_n;
    if (d_out.alloc((size_t)n_tet * 8 * 12) || d_n.alloc(4)) return set_error(DEFTET_ELAUNCH, "hipMalloc failed");
    TRY(deftet_tet_adj_share_i32((const int32_t *)d_tet.p, (int32_t *)d_out.p, (int32_t *)d_n.p, n_point, n_tet, d_ws.p, wsb, nullptr));
    DEFTET_HIP(hipDeviceSynchronize());
    DEFTET_HIP(hipMemcpy(n_face_edge_p, d_n.p, 4, hipMemcpyDeviceToHost));
    DEFTET_HIP(hipMemcpy(face_edge_p, d_out.p, (size_t)n_face_edge_p[0] * 24, hipMemcpyDeviceToHost));
    return DEFTET_OK;
}

extern "C" int deftet_tet_face_adj_host(int *tet_list, int *face_edge_p, int *n_face_edge_p, int n_point, int n_tet)
{
    HOST_PROLOGUE(n_tet)
    const long long cap = (long long)n_tet * 4 * 50;             // interface.py:27-28
    DevBuf d_out, d_n;
    if (d_out.alloc((size_t)cap * 8) || d_n.alloc(8)) return set_error(DEFTET_ELAUNCH, "hipMalloc failed");
    TRY(deftet_tet_face_adj_i32((const int32_t *)d_tet.p, (int32_t *)d_out.p, cap, (long long *)d_n.p, n_point, n_tet, 1,
                                d_ws.p, wsb, nullptr));
    DEFTET_HIP(hipDeviceSynchronize());
    long long cnt = 0;
    DEFTET_HIP(hipMemcpy(&cnt, d_n.p, 8, hipMemcpyDeviceToHost));
    if (cnt > cap) return set_error(DEFTET_ELIMIT, "face adjacency has %lld rows, more than the reference's 4*n_tet*50 buffer", cnt);
    n_face_edge_p[0] = (int)cnt;
    DEFTET_HIP(hipMemcpy(face_edge_p, d_out.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    return DEFTET_OK;
}

extern "C" int deftet_tet_point_adj_host(int *tet_list, int *edge_p, int *n_edge, int n_point, int n_tet)
{
    HOST_PROLOGUE(n_tet)
    DevBuf d_out, d_n;
    if (d_out.alloc((size_t)n_tet * 12 * 8) || d_n.alloc(4)) return set_error(DEFTET_ELAUNCH, "hipMalloc failed");
    TRY(deftet_tet_point_adj_i32((const int32_t *)d_tet.p, (int32_t *)d_out.p, (int32_t *)d_n.p, n_point, n_tet, d_ws.p, wsb, nullptr));
    DEFTET_HIP(hipDeviceSynchronize());
    DEFTET_HIP(hipMemcpy(n_edge, d_n.p, 4, hipMemcpyDeviceToHost));
    DEFTET_HIP(hipMemcpy(edge_p, d_out.p, (size_t)n_edge[0] * 8, hipMemcpyDeviceToHost));
    return DEFTET_OK;
}

extern "C" int deftet_colaps_v_host(float *point_p, int *map_array_p, int *inverse_idx_p, int *n_colaps_v_p, int n_point)
{
    DEFTET_CHECK_ARG(n_point >= 0, "negative size");
    DevBuf d_pts, d_map, d_inv, d_n, d_ws;
    const size_t wsb = deftet_builder_workspace_bytes(n_point, 0);
    if (d_pts.alloc((size_t)n_point * 12) || d_map.alloc((size_t)n_point * 4) || d_inv.alloc((size_t)n_point * 4) ||
        d_n.alloc(4) || d_ws.alloc(wsb))
        return set_error(DEFTET_ELAUNCH, "hipMalloc failed");
    DEFTET_HIP(hipMemcpy(d_pts.p, point_p, (size_t)n_point * 12, hipMemcpyHostToDevice));
    TRY(deftet_colaps_v_f32((const float *)d_pts.p, (int32_t *)d_map.p, (int32_t *)d_inv.p, (int32_t *)d_n.p, n_point, d_ws.p, wsb, nullptr));
    DEFTET_HIP(hipDeviceSynchronize());
    DEFTET_HIP(hipMemcpy(n_colaps_v_p, d_n.p, 4, hipMemcpyDeviceToHost));
    DEFTET_HIP(hipMemcpy(map_array_p, d_map.p, (size_t)n_point * 4, hipMemcpyDeviceToHost));
    DEFTET_HIP(hipMemcpy(inverse_idx_p, d_inv.p, (size_t)n_colaps_v_p[0] * 4, hipMemcpyDeviceToHost));
    return DEFTET_OK;
}
