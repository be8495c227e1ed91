/* deftet_oracle_sign.c — TEST INFRASTRUCTURE (never linked or called by the product path).
 *
 * N1: ground-truth occupancy by ray parity,
 *     kal.ops.mesh.check_sign(verts, faces, points, hash_resolution=512)
 * as called at /root/reference/layers/DefTet/deftet.py:46, eval.py:239, dataloader.py:92.
 *
 * PARITY UNPINNED.  Kaolin is a third-party dependency that is neither vendored in the reference
 * tree nor pinned to a version (README.md:30 only says "install kaolin"), and it cannot be
 * installed here.  What is restated below is Kaolin's PUBLISHED algorithm for CUDA inputs
 * (kaolin/ops/mesh/check_sign.py + csrc/ops/mesh/check_sign_cuda.cu, releases 0.9-0.12, where
 * `hash_resolution` only affects the CPU fallback): for every point count the triangles hit by the
 * ray p + t*(1,0,0), t > eps, with the Moller-Trumbore test; inside <=> the count is odd.
 * The exact operation order and eps = 1e-7 below are THIS repository's contract (fp32, no FMA,
 * compiled with -ffp-contract=off); the HIP kernels reproduce it bit for bit.  It is pinned
 * semantically in tests (centroids of a closed tet-boundary surface: inside <=> tet selected).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define CS_EPS 1e-7f

/* 1 if the ray from p along +x crosses triangle (v1,v2,v3) */
static int cs_hit(const float *v1, const float *v2, const float *v3, float px, float py, float pz)
{
    const float e1x = v2[0] - v1[0], e1y = v2[1] - v1[1], e1z = v2[2] - v1[2];
    const float e2x = v3[0] - v1[0], e2y = v3[1] - v1[1], e2z = v3[2] - v1[2];
    /* h = dir x e2 = (0, -e2z, e2y);  a = e1 . h */
    const float a = e1y * (-e2z) + e1z * e2y;
    if (a > -CS_EPS && a < CS_EPS) return 0;          /* ray parallel to the triangle */
    const float f = 1.0f / a;
    const float sx = px - v1[0], sy = py - v1[1], sz = pz - v1[2];
    const float u = f * (sy * (-e2z) + sz * e2y);
    if (u < 0.0f || u > 1.0f) return 0;
    /* q = s x e1 */
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = f * qx;                           /* dir . q */
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    return t > CS_EPS;
}

/* verts f32 [B,V,3], faces i64 [F,3], points f32 [B,N,3] -> out u8 [B,N] (1 = inside),
 * count i32 [B,N] (number of crossings; may be NULL) */
void oracle_check_sign_f32(const float *verts, const int64_t *faces, const float *points, uint8_t *out, int32_t *count,
                           int B, int V, int F, int N)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            const float *p = points + ((size_t)b * N + i) * 3;
            int c = 0;
            for (int k = 0; k < F; ++k) {
                const float *vb = verts + (size_t)b * V * 3;
                c += cs_hit(vb + faces[k * 3] * 3, vb + faces[k * 3 + 1] * 3, vb + faces[k * 3 + 2] * 3, p[0], p[1], p[2]);
            }
            out[(size_t)b * N + i] = (uint8_t)(c & 1);
            if (count) count[(size_t)b * N + i] = c;
        }
}
