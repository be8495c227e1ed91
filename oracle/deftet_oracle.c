/*
 * deftet_oracle.c — CPU restatement of the DefTet per-tetrahedron hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under deftet_amd/ may import, link or call this
 * file; it exists so tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * can check / time the HIP path against an independent plain-C statement of the
 * reference algorithm.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).
 *
 * PARITY STATUS
 *   - builders (tet_adj_share / tet_face_adj / tet_point_adj / colaps_v / tet_to_face):
 *     PINNED — checked in tests/ against the reference's own run.cpp compiled by
 *     oracle/Makefile into oracle/_ref/ and against golden vectors produced by
 *     importing the reference's Python twins (tests/golden/gen_golden.py).
 *   - point_in_tet / tri_dist / nn / face_edge_adj (CUDA kernels of the reference):
 *     the CUDA sources need nvcc + THC headers and are unbuildable in this image and
 *     the reference ships no tests or golden vectors for them ⇒ "parity unpinned" at
 *     the rounding level.  The restatement follows the cited kernel text operation by
 *     operation in fp32 with FMA contraction disabled (build with -ffp-contract=off),
 *     and is pinned at the semantic level against the reference's Python
 *     bary_centric_tet (utils/tet_utils.py:28-45) by tests/test_cpu_oracle_golden.py.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------
 * A1  point-in-tet occupancy query
 * layers/DefTet/check_condition_tetrahedron_base/check_condition_tet_for.cu
 * ---------------------------------------------------------------------------------- */

/* cuda_check_sign, check_condition_tet_for.cu:105-121 */
static int check_sign(const float *a, const float *b, const float *c, const float *d,
                      const float *p)
{
    float r1[3], r2[3], n[3];
    r1[0] = b[0] - a[0]; r1[1] = b[1] - a[1]; r1[2] = b[2] - a[2];      /* :111 */
    r2[0] = c[0] - a[0]; r2[1] = c[1] - a[1]; r2[2] = c[2] - a[2];      /* :112 */
    n[0] = r1[1] * r2[2] - r1[2] * r2[1];                               /* :63-65 */
    n[1] = r1[2] * r2[0] - r1[0] * r2[2];
    n[2] = r1[0] * r2[1] - r1[1] * r2[0];
    r1[0] = d[0] - a[0]; r1[1] = d[1] - a[1]; r1[2] = d[2] - a[2];      /* :114 */
    float dotv4 = n[0] * r1[0] + n[1] * r1[1] + n[2] * r1[2];           /* :115, :57 */
    r1[0] = p[0] - a[0]; r1[1] = p[1] - a[1]; r1[2] = p[2] - a[2];      /* :116 */
    float dotp = n[0] * r1[0] + n[1] * r1[1] + n[2] * r1[2];            /* :117 */
    int sign_p = dotp > 0;                                              /* :118 */
    int sign_v = dotv4 > 0;                                             /* :119 */
    return sign_p == sign_v;                                            /* :120 */
}

/* one query against the tets of one shape; dr_cuda_forward_kernel_batch :146-188 */
static float query_one(const float *tet_tx4x3, const float *p, int n_tet, long long *n_tested)
{
    float target = -1.0f;                                               /* :149 */
    int t;
    for (t = 0; t < n_tet; ++t) {                                       /* :152 */
        const float *a = tet_tx4x3 + (size_t)t * 12;                    /* :167-170 */
        const float *b = a + 3, *c = a + 6, *d = a + 9;
        int s1 = check_sign(a, b, c, d, p);                             /* :172 */
        int s2 = check_sign(b, a, d, c, p);                             /* :173 */
        int s3 = check_sign(c, d, a, b, p);                             /* :174 */
        int s4 = check_sign(d, c, b, a, p);                             /* :175 */
        if (s1 == s2 && s2 == s3 && s3 == s4) {                         /* :176 */
            target = (float)t;                                          /* :177 */
            ++t;
            break;                                                      /* :178 */
        }
    }
    if (n_tested) *n_tested += t;
    return target;
}

/* Whole batch, serial (the reference has no host threading).  Returns the number of
 * tet-point tests actually executed (early exit included) through *executed. */
void oracle_point_in_tet_f32(const float *tet_bxtx4x3, const float *pts_bxqx3,
                             float *cond_bxqx1, int n_batch, int n_tet, int n_query,
                             long long *executed)
{
    long long cnt = 0;
    for (int b = 0; b < n_batch; ++b)
        for (int q = 0; q < n_query; ++q) {
            const float *p = pts_bxqx3 + ((size_t)b * n_query + q) * 3;           /* :142 */
            cond_bxqx1[(size_t)b * n_query + q] =
                query_one(tet_bxtx4x3 + (size_t)b * n_tet * 12, p, n_tet, &cnt);  /* :188 */
        }
    if (executed) *executed = cnt;
}

/* Same, OpenMP over queries (all host cores); returns the thread count used. */
int oracle_point_in_tet_f32_omp(const float *tet_bxtx4x3, const float *pts_bxqx3,
                                float *cond_bxqx1, int n_batch, int n_tet, int n_query)
{
    int nthreads = 1;
    long long total = (long long)n_batch * n_query;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (long long i = 0; i < total; ++i) {
        int b = (int)(i / n_query);
        const float *p = pts_bxqx3 + (size_t)i * 3;
        cond_bxqx1[i] = query_one(tet_bxtx4x3 + (size_t)b * n_tet * 12, p, n_tet, NULL);
    }
    return nthreads;
}

/* Decision margin of a query against its tets: min over tets (scanned until the hit)
 * and faces of |dotp| / (|n|*|p-a| + tiny), in double.  Used by tests to report how
 * many decisions sit within rounding distance of a face (hard parts). */
void oracle_point_in_tet_margin_f32(const float *tet_tx4x3, const float *pts_qx3,
                                    double *margin_q, int n_tet, int n_query)
{
    static const int ord[4][4] = {{0, 1, 2, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {3, 2, 1, 0}};
    for (int q = 0; q < n_query; ++q) {
        const float *p = pts_qx3 + (size_t)q * 3;
        double best = 1e300;
        float hit = query_one(tet_tx4x3, p, n_tet, NULL);
        int last = hit < 0 ? n_tet - 1 : (int)hit;
        for (int t = 0; t <= last; ++t) {
            const float *v = tet_tx4x3 + (size_t)t * 12;
            for (int f = 0; f < 4; ++f) {
                const float *a = v + 3 * ord[f][0], *b = v + 3 * ord[f][1], *c = v + 3 * ord[f][2];
                double e1[3], e2[3], n[3], r[3];
                for (int k = 0; k < 3; ++k) { e1[k] = (double)b[k] - a[k]; e2[k] = (double)c[k] - a[k]; r[k] = (double)p[k] - a[k]; }
                n[0] = e1[1] * e2[2] - e1[2] * e2[1];
                n[1] = e1[2] * e2[0] - e1[0] * e2[2];
                n[2] = e1[0] * e2[1] - e1[1] * e2[0];
                double dot = n[0] * r[0] + n[1] * r[1] + n[2] * r[2];
                double den = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) *
                             sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) + 1e-300;
                double m = fabs(dot) / den;
                if (m < best) best = m;
            }
        }
        margin_q[q] = best;
    }
}

/* ------------------------------------------------------------------------------------
 * A1b  barycentric weights of a given tet (utils/tet_utils.py:25-45), fp32, same
 * association as the torch expression: sum(a * cross(b, c)) then * (1 / v).
 * idx < 0 -> weights 0.
 * ---------------------------------------------------------------------------------- */
static float triple(const float *a, const float *b, const float *c)
{   /* scaler_triplet_produt, tet_utils.py:25-26 */
    float x0 = b[1] * c[2] - b[2] * c[1];
    float x1 = b[2] * c[0] - b[0] * c[2];
    float x2 = b[0] * c[1] - b[1] * c[0];
    return (a[0] * x0 + a[1] * x1) + a[2] * x2;
}

void oracle_bary_f32(const float *tet_bxtx4x3, const float *pts_bxqx3, const float *cond_bxq,
                     float *w_bxqx4, int n_batch, int n_tet, int n_query)
{
    for (long long i = 0; i < (long long)n_batch * n_query; ++i) {
        int b = (int)(i / n_query);
        float *w = w_bxqx4 + i * 4;
        int t = (int)cond_bxq[i];
        if (t < 0) { w[0] = w[1] = w[2] = w[3] = 0.f; continue; }
        const float *A = tet_bxtx4x3 + ((size_t)b * n_tet + t) * 12, *B = A + 3, *C = A + 6, *D = A + 9;
        const float *p = pts_bxqx3 + i * 3;
        float vap[3], vbp[3], vab[3], vac[3], vad[3], vbc[3], vbd[3];
        for (int k = 0; k < 3; ++k) {
            vap[k] = p[k] - A[k]; vbp[k] = p[k] - B[k];                 /* :29-30 */
            vab[k] = B[k] - A[k]; vac[k] = C[k] - A[k]; vad[k] = D[k] - A[k];  /* :32-34 */
            vbc[k] = C[k] - B[k]; vbd[k] = D[k] - B[k];                 /* :36-37 */
        }
        float va6 = triple(vbp, vbd, vbc);                              /* :39 */
        float vb6 = triple(vap, vac, vad);                              /* :40 */
        float vc6 = triple(vap, vad, vab);                              /* :41 */
        float vd6 = triple(vap, vab, vac);                              /* :42 */
        float v6 = 1.0f / triple(vab, vac, vad);                        /* :43 */
        w[0] = va6 * v6; w[1] = vb6 * v6; w[2] = vc6 * v6; w[3] = vd6 * v6;   /* :45 */
    }
}

/* ------------------------------------------------------------------------------------
 * A1b backward, plain C (bench.py's CPU fwd+bwd figure; checked against the torch-autograd oracle in
 * tests/test_cpu_oracle_golden.py).  The weights of utils/tet_utils.py:28-45 are affine in p, w_i(p) = s_i . (p - base_i) / v,
 * with gradients  grad_p w_a = (vbd x vbc)/v, w_b: (vac x vad)/v, w_c: (vad x vab)/v, w_d: (vab x vac)/v;  from
 * p = sum_i w_i v_i, sum_i w_i = 1:   dL/dv_k = -w_k G,  G = sum_i g_i grad_p w_i.   A scatter-add in query order.
 * grad_tet must be zeroed by the caller.  Misses (cond < 0) contribute nothing.
 * ---------------------------------------------------------------------------------- */
static void cross3(const float *b, const float *c, float *x)
{
    x[0] = b[1] * c[2] - b[2] * c[1];
    x[1] = b[2] * c[0] - b[0] * c[2];
    x[2] = b[0] * c[1] - b[1] * c[0];
}

void oracle_bary_bwd_f32(const float *tet_bxtx4x3, const float *pts_bxqx3, const float *cond_bxq, const float *grad_w_bxqx4,
                         float *grad_tet_bxtx4x3, int n_batch, int n_tet, int n_query)
{
    for (long long i = 0; i < (long long)n_batch * n_query; ++i) {
        int b = (int)(i / n_query);
        int t = (int)cond_bxq[i];
        if (t < 0) continue;
        const float *A = tet_bxtx4x3 + ((size_t)b * n_tet + t) * 12, *B = A + 3, *C = A + 6, *D = A + 9;
        const float *p = pts_bxqx3 + i * 3, *g = grad_w_bxqx4 + i * 4;
        float vap[3], vbp[3], vab[3], vac[3], vad[3], vbc[3], vbd[3];
        for (int k = 0; k < 3; ++k) {
            vap[k] = p[k] - A[k]; vbp[k] = p[k] - B[k];
            vab[k] = B[k] - A[k]; vac[k] = C[k] - A[k]; vad[k] = D[k] - A[k];
            vbc[k] = C[k] - B[k]; vbd[k] = D[k] - B[k];
        }
        float na[3], nb[3], nc[3], nd[3];
        cross3(vbd, vbc, na); cross3(vac, vad, nb); cross3(vad, vab, nc); cross3(vab, vac, nd);
        float v6 = 1.0f / ((vab[0] * nb[0] + vab[1] * nb[1]) + vab[2] * nb[2]);      /* triple(vab, vac, vad) */
        float w[4];
        w[0] = ((vbp[0] * na[0] + vbp[1] * na[1]) + vbp[2] * na[2]) * v6;
        w[1] = ((vap[0] * nb[0] + vap[1] * nb[1]) + vap[2] * nb[2]) * v6;
        w[2] = ((vap[0] * nc[0] + vap[1] * nc[1]) + vap[2] * nc[2]) * v6;
        w[3] = ((vap[0] * nd[0] + vap[1] * nd[1]) + vap[2] * nd[2]) * v6;
        float G[3];
        for (int k = 0; k < 3; ++k) G[k] = (g[0] * na[k] + g[1] * nb[k] + g[2] * nc[k] + g[3] * nd[k]) * v6;
        float *gt = grad_tet_bxtx4x3 + ((size_t)b * n_tet + t) * 12;
        for (int v = 0; v < 4; ++v)
            for (int k = 0; k < 3; ++k) gt[v * 3 + k] -= w[v] * G[k];
    }
}

/* ------------------------------------------------------------------------------------
 * Builders.  Shared local-face table (utils/lib/tet_adj_share/run.cpp:42-45,
 * utils/tet_utils.py:160-163).
 * ---------------------------------------------------------------------------------- */
static const int FACE_IDX[4][3] = {{0, 1, 2}, {1, 0, 3}, {2, 3, 0}, {3, 2, 1}};

typedef struct { uint64_t key; int32_t owner; int32_t aux; } krec_t;

static int cmp_krec(const void *x, const void *y)
{   /* ascending key, then ascending insertion order (owner = tet*4+face or similar) */
    const krec_t *a = (const krec_t *)x, *b = (const krec_t *)y;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return (a->owner > b->owner) - (a->owner < b->owner);
}

/* face key = min*n^2 + max*n + mid, run.cpp:60-68 (note: max before mid) */
static uint64_t face_key(const int32_t *tet, int f, int n_point)
{
    int v0 = tet[FACE_IDX[f][0]], v1 = tet[FACE_IDX[f][1]], v2 = tet[FACE_IDX[f][2]];
    int a = v0, b = v0, c = v2;
    if (v1 < a) a = v1;
    if (v2 < a) a = v2;
    if (v1 > b) b = v1;
    if (v2 > b) b = v2;
    /* run.cpp:57-66: c starts as the third vertex and is overwritten by every vertex
     * that is neither min nor max (last one wins) */
    if (a != v0 && b != v0) c = v0;
    if (a != v1 && b != v1) c = v1;
    if (a != v2 && b != v2) c = v2;
    return (uint64_t)a * (uint64_t)n_point * (uint64_t)n_point + (uint64_t)b * (uint64_t)n_point + (uint64_t)c;
}

/* A2  tet_adj_share — utils/lib/tet_adj_share/run.cpp:40-97.
 * out rows [t0,t1,f0],[t1,t0,f1] per key with exactly two owners, ascending key;
 * returns the number of shared faces (*n_out = cnt, rows = 2*cnt). */
void oracle_tet_adj_share(const int32_t *tet_list, int32_t *out_rows_x3, int32_t *n_out,
                          int n_point, int n_tet)
{
    size_t n = (size_t)n_tet * 4;
    krec_t *r = (krec_t *)malloc((n ? n : 1) * sizeof(krec_t));
    for (int t = 0; t < n_tet; ++t)
        for (int f = 0; f < 4; ++f) {
            r[(size_t)t * 4 + f].key = face_key(tet_list + (size_t)t * 4, f, n_point);
            r[(size_t)t * 4 + f].owner = t * 4 + f;       /* insertion order, :74 */
            r[(size_t)t * 4 + f].aux = 0;
        }
    qsort(r, n, sizeof(krec_t), cmp_krec);                /* std::map order, :80 */
    int cnt = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && r[j].key == r[i].key) ++j;
        if (j - i == 2) {                                  /* :83 */
            int t0 = r[i].owner / 4, f0 = r[i].owner % 4;
            int t1 = r[i + 1].owner / 4, f1 = r[i + 1].owner % 4;
            int32_t *o = out_rows_x3 + (size_t)cnt * 6;
            o[0] = t0; o[1] = t1; o[2] = f0;               /* :84-86 */
            o[3] = t1; o[4] = t0; o[5] = f1;               /* :88-90 */
            ++cnt;
        }
        i = j;
    }
    *n_out = cnt;
    free(r);
}

/* A3  tet_face_adj — utils/lib/tet_face_adj/run.cpp:18-92.
 * wrap32 != 0 reproduces the native library's 32-bit edge key (`int e`, :39, wraps
 * modulo 2^32 and is ordered as a signed int by std::map<int,...>); wrap32 == 0 is the
 * Python twin (utils/tet_utils.py:155-201, unbounded ints, first-seen edge order —
 * canonicalise before comparing).  Rows [fa,fb]; *n_out = number of rows.
 * If out_rows_x2 is NULL only the count is produced. */
void oracle_tet_face_adj(const int32_t *tet_list, int32_t *out_rows_x2, long long *n_out,
                         int n_point, int n_tet, int wrap32)
{
    size_t n = (size_t)n_tet * 12;
    krec_t *r = (krec_t *)malloc((n ? n : 1) * sizeof(krec_t));
    uint64_t *fkey = (uint64_t *)malloc(((size_t)n_tet * 4 + 1) * sizeof(uint64_t));
    size_t m = 0;
    for (int t = 0; t < n_tet; ++t)
        for (int f = 0; f < 4; ++f) {
            const int32_t *tet = tet_list + (size_t)t * 4;
            int tri[3] = {tet[FACE_IDX[f][0]], tet[FACE_IDX[f][1]], tet[FACE_IDX[f][2]]};
            for (int e = 0; e < 3; ++e) {                                  /* :36-46 */
                int pa = tri[e] < tri[(e + 1) % 3] ? tri[e] : tri[(e + 1) % 3];
                int pb = tri[e] < tri[(e + 1) % 3] ? tri[(e + 1) % 3] : tri[e];
                uint64_t k;
                if (wrap32) {
                    uint32_t w = (uint32_t)pa * (uint32_t)n_point + (uint32_t)pb;   /* :39 */
                    k = (uint64_t)(w ^ 0x80000000u);      /* signed order of std::map<int> */
                } else {
                    k = (uint64_t)pa * (uint64_t)n_point + (uint64_t)pb;
                }
                r[m].key = k;
                r[m].owner = (int32_t)m;                   /* push_back order, :45 */
                r[m].aux = t * 4 + f;
                ++m;
            }
            /* absolute face id, :48-65.  NOTE the native code seeds face_p_c with
             * triangle[0] and keeps it when no vertex is strictly between (repeated
             * vertices); identical to face_key() for proper triangles. */
            int fa = tri[0], fb = tri[0], fc = tri[0];
            for (int i = 0; i < 3; ++i) { if (tri[i] < fa) fa = tri[i]; if (tri[i] > fb) fb = tri[i]; }
            for (int i = 0; i < 3; ++i) if (tri[i] != fa && tri[i] != fb) fc = tri[i];
            fkey[(size_t)t * 4 + f] = (uint64_t)fa * (uint64_t)n_point * (uint64_t)n_point +
                                      (uint64_t)fb * (uint64_t)n_point + (uint64_t)fc;
        }
    qsort(r, n, sizeof(krec_t), cmp_krec);
    long long cnt = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && r[j].key == r[i].key) ++j;
        for (size_t x = i; x < j; ++x)                                     /* :77 */
            for (size_t y = i; y < j; ++y) {                               /* :78 */
                int fa = r[x].aux, fb = r[y].aux;
                if (fa == fb) continue;                                    /* :79 */
                if (fkey[fa] == fkey[fb]) continue;                        /* :80 */
                if (out_rows_x2) { out_rows_x2[cnt * 2] = fa; out_rows_x2[cnt * 2 + 1] = fb; }
                ++cnt;
            }
        i = j;
    }
    *n_out = cnt;
    free(r);
    free(fkey);
}

/* A4  tet_point_adj — utils/lib/tet_point_adj/run.cpp:20-56.  The reference dumps an
 * unordered_set in hash-iteration order (unspecified); the oracle emits the same SET
 * sorted by (a,b), which is what the tests canonicalise both sides to. */
static int cmp_u64(const void *x, const void *y)
{
    uint64_t a = *(const uint64_t *)x, b = *(const uint64_t *)y;
    return (a > b) - (a < b);
}

void oracle_tet_point_adj(const int32_t *tet_list, int32_t *out_edges_x2, int32_t *n_out,
                          int n_point, int n_tet)
{
    size_t n = (size_t)n_tet * 12, m = 0;
    uint64_t *k = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (int t = 0; t < n_tet; ++t) {
        const int32_t *tet = tet_list + (size_t)t * 4;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (i != j) k[m++] = (uint64_t)tet[i] * (uint64_t)n_point + (uint64_t)tet[j];  /* :17-40 */
    }
    qsort(k, n, sizeof(uint64_t), cmp_u64);
    int cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i && k[i] == k[i - 1]) continue;
        out_edges_x2[(size_t)cnt * 2] = (int32_t)(k[i] / (uint64_t)n_point);          /* :47 */
        out_edges_x2[(size_t)cnt * 2 + 1] = (int32_t)(k[i] % (uint64_t)n_point);      /* :48 */
        ++cnt;
    }
    *n_out = cnt;
    free(k);
}

/* A5  colaps_v — utils/lib/colaps_v/run.cpp:18-59.  Key = "%.5f-%.5f-%.5f" of the three
 * floats promoted to double (ostream << float with std::fixed/setprecision(5) prints
 * exactly what printf("%.5f", (double)x) prints). */
typedef struct { char s[96]; int32_t idx; } srec_t;
static int cmp_srec(const void *x, const void *y)
{
    const srec_t *a = (const srec_t *)x, *b = (const srec_t *)y;
    int c = strcmp(a->s, b->s);
    if (c) return c;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

void oracle_colaps_v(const float *point_nx3, int32_t *map_array, int32_t *inverse_idx,
                     int32_t *n_colaps, int n_point)
{
    srec_t *r = (srec_t *)malloc(((size_t)n_point + 1) * sizeof(srec_t));
    int32_t *first = (int32_t *)malloc(((size_t)n_point + 1) * sizeof(int32_t));
    for (int i = 0; i < n_point; ++i) {
        snprintf(r[i].s, sizeof(r[i].s), "%.5f-%.5f-%.5f", (double)point_nx3[i * 3],
                 (double)point_nx3[i * 3 + 1], (double)point_nx3[i * 3 + 2]);          /* :18-38 */
        r[i].idx = i;
    }
    qsort(r, (size_t)n_point, sizeof(srec_t), cmp_srec);
    /* first[i] = smallest original index sharing i's key */
    for (int i = 0; i < n_point;) {
        int j = i;
        while (j < n_point && strcmp(r[j].s, r[i].s) == 0) { first[r[j].idx] = r[i].idx; ++j; }
        i = j;
    }
    int cnt = 0;
    for (int i = 0; i < n_point; ++i) {                                               /* :43-56 */
        if (first[i] == i) { inverse_idx[cnt] = i; map_array[i] = cnt; ++cnt; }
        else map_array[i] = map_array[first[i]];
    }
    *n_colaps = cnt;
    free(r);
    free(first);
}

/* A6  tet_to_face — utils/tet_utils.py:208-256 (with_boundary = 0) and the render-side
 * tet_to_face_idx(with_boundary=True), diff_render/diftet_6_subdiv/3_model/
 * prepare_for_wz.py:49-104 (with_boundary = 1: boundary faces stay inline at their
 * first-seen position with partner tet/face = -1).
 * Faces are listed in first-seen order while enumerating tets then local faces.
 * Outputs (caller allocates 4*n_tet rows each): face_fx3 (triangle as oriented in the
 * first owner), tetidx_fx2, tetfaceidx_fx2, boundary_fx3.  Keys with >2 owners are
 * dropped exactly like the reference (cnt_n_tet[2]) and counted in *n_multi. */
void oracle_tet_to_face(const int32_t *tet_list, int n_point, int n_tet, int with_boundary,
                        int64_t *face_fx3, int64_t *tetidx_fx2, int64_t *tetfaceidx_fx2,
                        int64_t *boundary_fx3, int32_t *n_face, int32_t *n_boundary,
                        int32_t *n_multi)
{
    size_t n = (size_t)n_tet * 4;
    krec_t *r = (krec_t *)malloc((n ? n : 1) * sizeof(krec_t));
    for (size_t i = 0; i < n; ++i) {
        r[i].key = face_key(tet_list + (i / 4) * 4, (int)(i % 4), n_point);
        r[i].owner = (int32_t)i;
        r[i].aux = 0;
    }
    qsort(r, n, sizeof(krec_t), cmp_krec);
    /* group size and second owner, indexed by first owner */
    int32_t *gsize = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    int32_t *second = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && r[j].key == r[i].key) ++j;
        gsize[r[i].owner] = (int32_t)(j - i);
        second[r[i].owner] = (j - i >= 2) ? r[i + 1].owner : -1;
        i = j;
    }
    int nf = 0, nb = 0, nm = 0;
    for (size_t o = 0; o < n; ++o) {          /* dict insertion order == first-seen order */
        if (!gsize[o]) continue;
        const int32_t *tet = tet_list + (o / 4) * 4;
        int f = (int)(o % 4);
        int64_t tri[3] = {tet[FACE_IDX[f][0]], tet[FACE_IDX[f][1]], tet[FACE_IDX[f][2]]};
        if (gsize[o] == 2 || (with_boundary && gsize[o] == 1)) {
            memcpy(face_fx3 + (size_t)nf * 3, tri, sizeof(tri));
            tetidx_fx2[(size_t)nf * 2] = (int64_t)(o / 4);
            tetidx_fx2[(size_t)nf * 2 + 1] = gsize[o] == 2 ? second[o] / 4 : -1;
            tetfaceidx_fx2[(size_t)nf * 2] = f;
            tetfaceidx_fx2[(size_t)nf * 2 + 1] = gsize[o] == 2 ? second[o] % 4 : -1;
            ++nf;
        }
        if (gsize[o] == 1) { memcpy(boundary_fx3 + (size_t)nb * 3, tri, sizeof(tri)); ++nb; }
        if (gsize[o] > 2) ++nm;
    }
    *n_face = nf; *n_boundary = nb; *n_multi = nm;
    free(r); free(gsize); free(second);
}
