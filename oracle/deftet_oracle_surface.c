/*
 * deftet_oracle_surface.c — CPU restatement of the per-shape surface operators that
 * layers.DefTet.forward calls.
 *
 * TEST INFRASTRUCTURE ONLY (see deftet_oracle.c header).  The reference kernels are CUDA
 * (unbuildable here, no golden vectors) ⇒ parity unpinned at the rounding level; the
 * restatement follows the cited kernel text operation by operation in fp32, FMA
 * contraction disabled, double-typed literals promoted exactly as C++ would.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * A8  surface-face edge adjacency by position
 * layers/DefTet/tet_face_adj_m_idx/tet_face_adj_m_for.cu:15-108
 * ---------------------------------------------------------------------------------- */
static int pos_equal(const float *a, const float *b)
{   /* equal(), :26-35 — float accumulation, compared against the double 1e-15 */
    float diff = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float d = a[i] - b[i];
        if (d < 0) d = -d;                         /* abs(), :17-23 */
        diff += d;
    }
    return (double)diff <= 1e-15;
}

static int check_share(const float *fa, const float *fb)
{   /* :38-69 */
    int find = 0;
    for (int ia = 0; ia < 3; ++ia) {
        const float *aa = fa + ia * 3, *ab = fa + ((ia + 1) % 3) * 3;
        for (int ib = 0; ib < 3; ++ib) {
            const float *ba = fb + ib * 3, *bb = fb + ((ib + 1) % 3) * 3;
            if (pos_equal(aa, ba) && pos_equal(ab, bb)) find = 1;      /* :60 */
            if (pos_equal(aa, bb) && pos_equal(ab, ba)) find = 1;      /* :63 */
        }
    }
    return find;
}

/* adj_fxm pre-filled with -1 by the caller (utils.py:47); kernel :72-108 */
void oracle_face_edge_adj_f32(const float *face_fx3x3, float *adj_fxm, int n_face, int n_max_nei)
{
    for (int f = 0; f < n_face; ++f) {
        int found = 0;
        for (int g = 0; g < n_face; ++g) {
            if (g == f) continue;                                      /* :96 */
            if (check_share(face_fx3x3 + (size_t)f * 9, face_fx3x3 + (size_t)g * 9)) {
                adj_fxm[(size_t)f * n_max_nei + found] = (float)g;     /* :101 */
                ++found;
            }
            if (found >= n_max_nei) break;                             /* :104-106 */
        }
    }
}

/* ------------------------------------------------------------------------------------
 * A9  point -> triangle-soup squared distance, forward and backward
 * layers/DefTet/tet_analytic_distance_batch/tet_analytic_distance_for.cu:15-307
 * layers/DefTet/tet_analytic_distance_batch/tet_analytic_distance_back.cu:15-686
 * ---------------------------------------------------------------------------------- */
static float divide_non_zero(float a)
{   /* for.cu:40-52 — `eps` is the double literal 1e-10: the sum is formed in double
     * and rounded back to scalar_t on return */
    if (a == 0) return (float)1e-10;
    if (a < 0) return (float)((double)a - 1e-10);
    if (a > 0) return (float)((double)a + 1e-10);
    return (float)1e-10;
}

static float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }  /* :82-84 */
static float fabs_ref(float a) { return a > 0.0 ? a : -a; }                                             /* :20-28 */

static float min3(float a, float b, float c) { float m = a; if (b < m) m = b; if (c < m) m = c; return m; }        /* :69-79 */
static float min3_idx(float a, float b, float c)
{   /* back.cu:139-152 */
    float m = a, i = 0;
    if (b < m) { m = b; i = 1; }
    if (c < m) { m = c; i = 2; }
    return i;
}

static float dist_point_sq(const float *a, const float *b)
{   /* :139-146 */
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

static float distance_line_square(const float *A, const float *B, const float *P)
{   /* :148-170 */
    float PA[3], BA[3], d[3];
    for (int k = 0; k < 3; ++k) { PA[k] = P[k] - A[k]; BA[k] = B[k] - A[k]; }
    float t = dot3(PA, BA) / divide_non_zero(dot3(BA, BA));
    for (int k = 0; k < 3; ++k) { float tmp = BA[k] * t; d[k] = PA[k] - tmp; }
    float distance = dot3(d, d);
    if (t >= 0 && t <= 1) return distance;
    return -distance;
}

/* cuda_line_distance, for.cu:172-220 / back.cu:348-403; ret = {case, dist, idx} */
static void line_distance(const float *a, const float *b, const float *c, const float *p,
                          float *ret, float max_dis)
{
    float k1 = (b[1] - c[1]) * (p[0] - c[0]) + (c[0] - b[0]) * (p[1] - c[1]);
    float k2 = (a[0] - c[0]) * (p[1] - c[1]) + (c[1] - a[1]) * (p[0] - c[0]);
    float k3 = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1]);
    if (k3 == 0) { ret[0] = -1; return; }
    float l1 = k1 / k3, l2 = k2 / k3, l3 = 1 - l1 - l2;
    float dis12 = distance_line_square(a, b, p);
    float dis23 = distance_line_square(b, c, p);
    float dis13 = distance_line_square(a, c, p);
    if (l1 >= 0 && l2 >= 0 && l3 >= 0) {
        ret[0] = 0;
        ret[1] = min3(fabs_ref(dis12), fabs_ref(dis23), fabs_ref(dis13));
        ret[2] = min3_idx(fabs_ref(dis12), fabs_ref(dis23), fabs_ref(dis13));
        return;
    }
    if (dis12 <= 0) dis12 = max_dis;
    if (dis23 <= 0) dis23 = max_dis;
    if (dis13 <= 0) dis13 = max_dis;
    float min_line = min3(dis12, dis23, dis13), min_line_idx = min3_idx(dis12, dis23, dis13);
    float d1 = dist_point_sq(a, p), d2 = dist_point_sq(b, p), d3 = dist_point_sq(c, p);
    float min_pt = min3(d1, d2, d3), min_pt_idx = min3_idx(d1, d2, d3);
    if (min_line < min_pt) { ret[0] = 1; ret[1] = min_line; ret[2] = min_line_idx; }
    else { ret[0] = 2; ret[1] = min_pt; ret[2] = min_pt_idx; }
}

static void plane_project(const float *a, const float *b, const float *c, const float *p,
                          float *ip, float *t_out)
{   /* for.cu:227-238 / back.cu:411-421 */
    float r1[3], r2[3], n[3];
    for (int k = 0; k < 3; ++k) { r1[k] = b[k] - a[k]; r2[k] = c[k] - a[k]; }
    n[0] = r1[1] * r2[2] - r1[2] * r2[1];
    n[1] = r1[2] * r2[0] - r1[0] * r2[2];
    n[2] = r1[0] * r2[1] - r1[1] * r2[0];
    float length = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);    /* cuda_normalize :128-137 */
    length = divide_non_zero(length);
    n[0] = n[0] / length; n[1] = n[1] / length; n[2] = n[2] / length;
    float t = dot3(n, a) - dot3(n, p);
    for (int k = 0; k < 3; ++k) { float m = n[k] * t; ip[k] = p[k] + m; }
    *t_out = t;
}

static float min_triangle_distance(const float *a, const float *b, const float *c, const float *p,
                                   float *ret, float *ip, float max_dis)
{   /* for.cu:222-254 (max_dis 10000) / back.cu:406-436 (max_dis 9999999) */
    float t;
    plane_project(a, b, c, p, ip, &t);
    float distance_1 = t * t;
    line_distance(a, b, c, ip, ret, max_dis);
    if (ret[0] == 0) return distance_1;
    if (ret[0] < 0) return max_dis;
    return distance_1 + ret[1];
}

/* forward kernel, for.cu:256-307 */
void oracle_tri_dist_fwd_f32(const float *pts_bxpx3, const float *face_bxfx3x3, const float *n_face_b,
                             float *closest_d, float *closest_f, int n_batch, int n_point, int n_max_face)
{
    for (int b = 0; b < n_batch; ++b) {
        int nf = (int)n_face_b[b];                                     /* :285 */
        for (int q = 0; q < n_point; ++q) {
            const float *p = pts_bxpx3 + ((size_t)b * n_point + q) * 3;
            float min_d = 10000.0f;                                    /* :277 */
            int min_idx = -1;
            for (int f = 0; f < nf; ++f) {
                const float *fc = face_bxfx3x3 + ((size_t)b * n_max_face + f) * 9;
                float ret[3] = {0, 0, 0}, ip[3];
                float dis = min_triangle_distance(fc, fc + 3, fc + 6, p, ret, ip, 10000.0f);
                if (min_d > dis) { min_d = dis; min_idx = f; }         /* :300-303 */
            }
            closest_d[(size_t)b * n_point + q] = min_d;
            closest_f[(size_t)b * n_point + q] = (float)min_idx;
        }
    }
}

/* backward kernel, back.cu:591-686.  dldface must be zero-initialised by the caller
 * (utils.py:65).  Accumulation is serial in point order (the reference's atomicAdd order
 * is nondeterministic).  A saved face index < 0 is skipped (the reference would read
 * out of bounds). */
void oracle_tri_dist_bwd_f32(const float *pts_bxpx3, const float *face_bxfx3x3, const float *closest_f,
                             const float *dl_dd, float *dldface_bxfx3x3, int n_batch, int n_point, int n_face)
{
    for (int b = 0; b < n_batch; ++b)
        for (int q = 0; q < n_point; ++q) {
            size_t i = (size_t)b * n_point + q;
            const float *p = pts_bxpx3 + i * 3;
            int fi = (int)closest_f[i];                                /* :618 */
            if (fi < 0 || fi >= n_face) continue;
            const float *fc = face_bxfx3x3 + ((size_t)b * n_face + fi) * 9;
            float *g = dldface_bxfx3x3 + ((size_t)b * n_face + fi) * 9;
            float ret[3] = {0, 0, 0}, ip[3];
            min_triangle_distance(fc, fc + 3, fc + 6, p, ret, ip, 9999999.0f);   /* :628 */
            float gp = dl_dd[i];                                       /* :629 */
            if (ret[0] == 0) {                                         /* :630-651 */
                /* cuda_gradient_triangle_distance, back.cu:439-483 */
                const float *a = fc, *bb = fc + 3, *c = fc + 6;
                float ip2[3], t;
                plane_project(a, bb, c, p, ip2, &t);
                float k1 = (bb[1] - c[1]) * (ip2[0] - c[0]) + (c[0] - bb[0]) * (ip2[1] - c[1]);
                float k2 = (a[0] - c[0]) * (ip2[1] - c[1]) + (c[1] - a[1]) * (ip2[0] - c[0]);
                float k3 = (bb[1] - c[1]) * (a[0] - c[0]) + (c[0] - bb[0]) * (a[1] - c[1]);
                float grad[9] = {0};
                if (k3 != 0) {
                    float l1 = k1 / k3, l2 = k2 / k3, l3 = 1 - l1 - l2;
                    for (int k = 0; k < 3; ++k) {
                        grad[k] = 2 * (ip2[k] - p[k]) * l1;
                        grad[3 + k] = 2 * (ip2[k] - p[k]) * l2;
                        grad[6 + k] = 2 * (ip2[k] - p[k]) * l3;
                    }
                }
                for (int k = 0; k < 9; ++k) g[k] += gp * grad[k];
            }
            if (ret[0] == 1) {                                         /* :652-670 */
                int i1 = (int)ret[2], i2 = (i1 + 1) % 3;
                const float *A = fc + i1 * 3, *B = fc + i2 * 3;
                /* cuda_gradient_line_distance, back.cu:291-317: only grad[0..2] is written
                 * (second assignment wins); grad[3..5] keeps its zero initialiser */
                float PA[3], BA[3];
                for (int k = 0; k < 3; ++k) { PA[k] = p[k] - A[k]; BA[k] = B[k] - A[k]; }
                float t = dot3(PA, BA) / divide_non_zero(dot3(BA, BA));
                float gl[6] = {0};
                for (int k = 0; k < 3; ++k) {
                    float tmp = B[k] * t;
                    float ipk = A[k] * (1 - t);
                    ipk = ipk + tmp;
                    gl[k] = 2 * (ipk - p[k]) * (t);
                }
                for (int k = 0; k < 3; ++k) { g[i1 * 3 + k] += gp * gl[k]; g[i2 * 3 + k] += gp * gl[3 + k]; }
            }
            if (ret[0] == 2) {                                         /* :671-685 */
                int iv = (int)ret[2];
                for (int k = 0; k < 3; ++k) {
                    float gl = fc[iv * 3 + k] - p[k];
                    gl = gl * 1.0f;
                    g[iv * 3 + k] += 2 * gp * gl;
                }
            }
        }
}

/* ------------------------------------------------------------------------------------
 * A10  brute-force nearest neighbour index
 * layers/nearest_neighbor/nearest_neighbor_cuda.cu:17-55
 * ---------------------------------------------------------------------------------- */
void oracle_nn_index_f32(const float *queries_bxnx3, const float *points_bxmx3, int32_t *result_bxn,
                         int n_batch, int n_query, int n_point)
{
    for (int b = 0; b < n_batch; ++b)
        for (int q = 0; q < n_query; ++q) {
            const float *qq = queries_bxnx3 + ((size_t)b * n_query + q) * 3;
            const float *pp = points_bxmx3 + (size_t)b * n_point * 3;
            float min_distance = 1e20f;                                /* :28 */
            int min_point = 0;
            for (int i = 0; i < n_point; ++i) {
                float distance = 0;
                float d0 = pp[i * 3] - qq[0], d1 = pp[i * 3 + 1] - qq[1], d2 = pp[i * 3 + 2] - qq[2];
                distance += d0 * d0;                                   /* :42 */
                distance += d1 * d1;                                   /* :44 */
                distance += d2 * d2;                                   /* :46 */
                if (distance < min_distance) { min_point = i; min_distance = distance; }
            }
            result_bxn[(size_t)b * n_query + q] = min_point;
        }
}
