/*
 * deftet_oracle_render.c — CPU statement of the differentiable tet rasterizer contract
 * (SURVEY.md section 8 row A12).  TEST INFRASTRUCTURE ONLY (see deftet_oracle.c header).
 *
 * PARITY UNPINNED: the reference calls kaolin.render.mesh.deftet_sparse_render
 * (diff_render/diftet_6_subdiv/5_rendereq/deftetrneder.py:97-100); Kaolin is a third-party,
 * un-vendored, un-pinned dependency (README.md:30) that is not under /root/reference and not
 * installed here.  This file states the contract as recalled from Kaolin's public
 * documentation, and is what the HIP rasterizer is tested against:
 *
 *   for every pixel p and every face f in ascending index (a,b,c = image-space vertices):
 *       m = bx-ax; pp = by-ay; n = cx-ax; q = cy-ay; s = px-ax; t = py-ay
 *       k1 = s*q - n*t;  k2 = m*t - s*pp;  k3 = m*q - n*pp
 *       w1 = k1/(k3+eps); w2 = k2/(k3+eps); w0 = 1 - w1 - w2
 *       covered iff w0 >= 0 and w1 >= 0 and w2 >= 0
 *       z = w0*az + w1*bz + w2*cz;  kept iff range_min <= z <= range_max
 *   which kept faces are RECORDED when a pixel has more than `knum` of them is a policy (nothing in the reference tree
 *   settles it; at the reference's call site knum = 300 against ~60 covering faces, so both give the same images):
 *       policy 0, NEAREST: the `knum` kept faces that come first in the output order below (what an insertion-sorted
 *                 list of bounded length keeps — this build's recollection of Kaolin's kernel);
 *       policy 1, FIRST:   the first `knum` kept faces in ascending face index (rounds 1-2 of this build)
 *   the recorded faces are ordered by z descending (the camera looks down -z: nearest first),
 *   ties by ascending face index; unused slots: face -1, weights 0, features 0
 *   features = (w0*f0 + w1*f1) + w2*f2
 * fp32, operation order as written, no FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float z; int32_t f; float w0, w1, w2; } hit_t;

static int cmp_hit(const void *x, const void *y)
{
    const hit_t *a = (const hit_t *)x, *b = (const hit_t *)y;
    if (a->z != b->z) return a->z > b->z ? -1 : 1;
    return (a->f > b->f) - (a->f < b->f);
}

void oracle_sparse_render_fwd_policy_f32(const float *pixel_bxpx2, const float *range_bxpx2, const float *face_z_bxfx3,
                                         const float *face_xy_bxfx3x2, const float *face_feat_bxfx3xd,
                                         float *out_feat_bxpxkxd, int64_t *out_face_bxpxk, float *out_w_bxpxkx3,
                                         int B, int P, int F, int D, int knum, float eps, int policy)
{
    /* NEAREST: all kept faces are collected and sorted, the first knum of the sorted list are the record */
    const size_t cap = policy == 0 ? (size_t)(F > 0 ? F : 1) : (size_t)(knum > 0 ? knum : 1);
    hit_t *h = (hit_t *)malloc(cap * sizeof(hit_t));
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const float px = pixel_bxpx2[((size_t)b * P + p) * 2], py = pixel_bxpx2[((size_t)b * P + p) * 2 + 1];
            const float zmin = range_bxpx2[((size_t)b * P + p) * 2], zmax = range_bxpx2[((size_t)b * P + p) * 2 + 1];
            int nh = 0;
            for (int f = 0; f < F && (policy == 0 || nh < knum); ++f) {
                const float *xy = face_xy_bxfx3x2 + ((size_t)b * F + f) * 6;
                const float *zz = face_z_bxfx3 + ((size_t)b * F + f) * 3;
                const float ax = xy[0], ay = xy[1], bx = xy[2], by = xy[3], cx = xy[4], cy = xy[5];
                const float m = bx - ax, pp = by - ay, n = cx - ax, q = cy - ay, s = px - ax, t = py - ay;
                const float k1 = s * q - n * t, k2 = m * t - s * pp, k3 = m * q - n * pp;
                const float den = k3 + eps;
                const float w1 = k1 / den, w2 = k2 / den, w0 = 1 - w1 - w2;
                if (!(w0 >= 0 && w1 >= 0 && w2 >= 0)) continue;
                const float z = (w0 * zz[0] + w1 * zz[1]) + w2 * zz[2];
                if (!(z >= zmin && z <= zmax)) continue;
                h[nh].z = z; h[nh].f = f; h[nh].w0 = w0; h[nh].w1 = w1; h[nh].w2 = w2;
                ++nh;
            }
            qsort(h, (size_t)nh, sizeof(hit_t), cmp_hit);
            if (nh > knum) nh = knum;
            for (int j = 0; j < knum; ++j) {
                const size_t o = ((size_t)b * P + p) * knum + j;
                if (j < nh) {
                    out_face_bxpxk[o] = h[j].f;
                    out_w_bxpxkx3[o * 3] = h[j].w0; out_w_bxpxkx3[o * 3 + 1] = h[j].w1; out_w_bxpxkx3[o * 3 + 2] = h[j].w2;
                    const float *ff = face_feat_bxfx3xd + ((size_t)b * F + h[j].f) * 3 * D;
                    for (int d = 0; d < D; ++d)
                        out_feat_bxpxkxd[o * D + d] = (h[j].w0 * ff[d] + h[j].w1 * ff[D + d]) + h[j].w2 * ff[2 * D + d];
                } else {
                    out_face_bxpxk[o] = -1;
                    out_w_bxpxkx3[o * 3] = out_w_bxpxkx3[o * 3 + 1] = out_w_bxpxkx3[o * 3 + 2] = 0.f;
                    for (int d = 0; d < D; ++d) out_feat_bxpxkxd[o * D + d] = 0.f;
                }
            }
        }
    free(h);
}

void oracle_sparse_render_fwd_f32(const float *pixel_bxpx2, const float *range_bxpx2, const float *face_z_bxfx3,
                                  const float *face_xy_bxfx3x2, const float *face_feat_bxfx3xd,
                                  float *out_feat_bxpxkxd, int64_t *out_face_bxpxk, float *out_w_bxpxkx3,
                                  int B, int P, int F, int D, int knum, float eps)
{
    oracle_sparse_render_fwd_policy_f32(pixel_bxpx2, range_bxpx2, face_z_bxfx3, face_xy_bxfx3x2, face_feat_bxfx3xd, out_feat_bxpxkxd,
                                        out_face_bxpxk, out_w_bxpxkx3, B, P, F, D, knum, eps, 0);
}
