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
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

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

// per-point gradient contributions of the backward kernel (back.cu:591-686): up to 9 values
// for up to 3 vertices of the saved face.  Returns the number of (slot,value) pairs written.
__device__ __forceinline__ int tri_dist_point_grad(const float *fc, const float *p, float gp, int *slot, float *val)
{
    float ret[3] = {0.f, 0.f, 0.f}, ip[3];
    min_triangle_distance<true>(fc, fc + 3, fc + 6, p, ret, ip, 9999999.0f);     // back.cu:628
    int n = 0;
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
        for (int k = 0; k < 9; ++k) { slot[n] = k; val[n] = gp * grad[k]; ++n; }
    } else if (ret[0] == 1) {                                       // :652-670
        const int i1 = (int)ret[2], i2 = (i1 + 1) % 3;
        const float *A = fc + i1 * 3, *B = fc + i2 * 3;
        // cuda_gradient_line_distance (:291-317): the second assignment of grad[0..2] wins, grad[3..5] stays 0
        float PA[3], BA[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { PA[k] = p[k] - A[k]; BA[k] = B[k] - A[k]; }
        const float t = dot3(PA, BA) / divide_non_zero(dot3(BA, BA));
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float tmp = B[k] * t;
            float ipk = A[k] * (1 - t);
            ipk = ipk + tmp;
            slot[n] = i1 * 3 + k; val[n] = gp * (2 * (ipk - p[k]) * (t)); ++n;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { slot[n] = i2 * 3 + k; val[n] = gp * 0.0f; ++n; }
    } else if (ret[0] == 2) {                                       // :671-685
        const int iv = (int)ret[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float gl = fc[iv * 3 + k] - p[k];
            gl = gl * 1.0f;
            slot[n] = iv * 3 + k; val[n] = 2 * gp * gl; ++n;
        }
    }
    return n;
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
    int slot[9];
    float val[9];
    const int n = tri_dist_point_grad(fc, p, dl_dd[i], slot, val);
    float *g = dldface + ((size_t)b * F + fi) * 9;
    for (int k = 0; k < n; ++k) unsafeAtomicAdd(g + slot[k], val[k]);           // back.cu:640-683
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
        int slot[9];
        float val[9];
        const int m = tri_dist_point_grad(fc, p, dl_dd[pi], slot, val);
        for (int j = 0; j < m; ++j) {
#pragma unroll
            for (int s = 0; s < 9; ++s)
                if (slot[j] == s) acc[s] += val[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) g[j] = acc[j];
}

}  // namespace surf
}  // namespace deftet

using namespace deftet;
using namespace deftet::surf;

extern "C" int deftet_nn_index_f32(const float *queries, const float *points, int32_t *result, int B, int N, int M,
                                   void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && N >= 0 && M >= 0 && B <= 65535, "bad size");
    if (B == 0 || N == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(queries && result && (M == 0 || points), "null pointer");
    DEFTET_CHECK_ARG((long long)M * 3 < 2147483647LL, "too many points per shape");
    DEFTET_LAUNCH(k_nn, dim3((N + 255) / 256, B), dim3(256), as_stream(stream_), queries, points, N, M, result);
    return DEFTET_OK;
}

extern "C" size_t deftet_face_edge_adj_workspace_bytes(int) { return 0; }

extern "C" int deftet_face_edge_adj_f32(const float *face, float *adj, int F, int max_nei, void *, size_t, void *stream_)
{
    DEFTET_CHECK_ARG(F >= 0 && max_nei >= 0, "negative size");
    if (F >= (1 << 24)) return set_error(DEFTET_ELIMIT, "n_face=%d does not fit a float-encoded index", F);
    if (F == 0 || max_nei == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(face && adj, "null pointer");
    DEFTET_LAUNCH(k_face_edge_adj, dim3((F + 255) / 256), dim3(256), as_stream(stream_), face, adj, F, max_nei);
    return DEFTET_OK;
}

extern "C" int deftet_tri_dist_fwd_f32(const float *pts, const float *face, const float *n_face_b, float *closest_d,
                                       float *closest_f, int B, int P, int Fmax, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && P >= 0 && Fmax >= 0 && B <= 65535, "bad size");
    if (Fmax >= (1 << 24)) return set_error(DEFTET_ELIMIT, "n_face=%d does not fit a float-encoded index", Fmax);
    if (B == 0 || P == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pts && n_face_b && closest_d && closest_f && (Fmax == 0 || face), "null pointer");
    DEFTET_LAUNCH(k_tri_dist_fwd, dim3((P + 255) / 256, B), dim3(256), as_stream(stream_), pts, face, n_face_b, closest_d,
                  closest_f, P, Fmax);
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
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, need, key, skey, (size_t)n, 0, 64, st);
    if (e != hipSuccess) return set_error(DEFTET_ELAUNCH, "radix_sort_keys(size): %s", hipGetErrorString(e));
    DEFTET_HIP(hipMallocAsync((void **)&key, (size_t)n * 8, st));
    DEFTET_HIP(hipMallocAsync((void **)&skey, (size_t)n * 8, st));
    DEFTET_HIP(hipMallocAsync(&tmp, need ? need : 1, st));
    DEFTET_LAUNCH(k_bwd_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), st, closest_f, n, P, F, key);
    e = rocprim::radix_sort_keys(tmp, need, key, skey, (size_t)n, 0, 64, st);
    if (e != hipSuccess) return set_error(DEFTET_ELAUNCH, "radix_sort_keys: %s", hipGetErrorString(e));
    DEFTET_LAUNCH(k_tri_dist_bwd_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), st, pts, face, dl_dd, skey, n, P, F, dldface);
    DEFTET_HIP(hipFreeAsync(key, st));
    DEFTET_HIP(hipFreeAsync(skey, st));
    DEFTET_HIP(hipFreeAsync(tmp, st));
    return DEFTET_OK;
}
