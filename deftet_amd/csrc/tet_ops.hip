// tet_ops.hip — A7 boundary-face extraction and A11 fused per-tet energies (gfx950).
//
//   A7  DefTet.get_boundary_index / get_internal_index   layers/DefTet/deftet.py:186-203
//   A11 DefTet.volume_variance / amips_energy / edge_length   layers/DefTet/deftet.py:239-338
//       (det_m: utils/matrix_utils.py:42-47)
//
// The reference runs A7 as B gathers + boolean-mask compactions (one host sync per shape) and
// A11 as ~20 elementwise torch kernels that each stream [B,T,4,3].  Here A7 is flag -> prefix
// sum -> ordered emit for the whole batch (row-major mask order preserved, one sync for the
// offsets), and A11 reads each 48-byte tet record once per pass and reduces in fp64.
#include <cstring>

#include "common.hpp"
#include <atomic>

#include "prims.hpp"

namespace deftet {
namespace tops {

// ---------------------------------------------------------------------------- A7
// mode 1: faces whose two tets have occupancy sum == 1 (boundary, winding flipped when the
// FIRST tet is the occupied one, deftet.py:190-194); mode 2: sum == 2 (internal, :197-203).
__global__ __launch_bounds__(256) void k_bnd_flag(const long long *__restrict__ tetidx, const float *__restrict__ occ, int T,
                                                  int Fi, int mode, int *flag)
{
    const int b = blockIdx.y;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Fi) return;
    const float o0 = occ[(size_t)b * T + tetidx[f * 2]], o1 = occ[(size_t)b * T + tetidx[f * 2 + 1]];
    const float s = o0 + o1;                                          // tet_face_occ_bxfx2.sum(dim=-1), :189
    flag[(size_t)b * Fi + f] = (s == (float)mode) ? 1 : 0;            // t == 1 / t == 2
}

__global__ __launch_bounds__(256) void k_bnd_emit(const long long *__restrict__ face, const long long *__restrict__ tetidx,
                                                  const float *__restrict__ occ, const int *__restrict__ flag,
                                                  const int *__restrict__ pos, int T, int Fi, int B, int mode, long long *out,
                                                  int *offsets)
{
    const int b = blockIdx.y;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= Fi) return;
    const size_t i = (size_t)b * Fi + f;
    if (flag[i]) {
        long long v0 = face[f * 3], v1 = face[f * 3 + 1], v2 = face[f * 3 + 2];
        if (mode == 1 && occ[(size_t)b * T + tetidx[f * 2]] == 1.0f) {    // change_idx: t[:,0] == 1, :191
            const long long t = v0; v0 = v2; v2 = t;                     // b.flip(dims=[1]), :193
        }
        long long *o = out + (size_t)pos[i] * 3;
        o[0] = v0; o[1] = v1; o[2] = v2;
    }
    if (f == 0) offsets[b] = pos[i];                                   // start of shape b in the flat output
    if (b == B - 1 && f == Fi - 1) offsets[B] = pos[i] + flag[i];
}

// ---------------------------------------------------------------------------- A11
struct TetV { float A[3], Bv[3], C[3], D[3]; };

__device__ __forceinline__ TetV load_tet(const float *__restrict__ tet, size_t i)
{
    const float4 *src = reinterpret_cast<const float4 *>(tet + i * 12);
    const float4 t0 = src[0], t1 = src[1], t2 = src[2];
    TetV v;
    v.A[0] = t0.x; v.A[1] = t0.y; v.A[2] = t0.z; v.Bv[0] = t0.w; v.Bv[1] = t1.x; v.Bv[2] = t1.y;
    v.C[0] = t1.z; v.C[1] = t1.w; v.C[2] = t2.x; v.D[0] = t2.y; v.D[1] = t2.z; v.D[2] = t2.w;
    return v;
}
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float ipow(float x, int p)
{
    float r = 1.f;
    for (int i = 0; i < p; ++i) r *= x;
    return r;
}
// The same product with the exponent known at compile time (P4: both exponents are the reference's default 4,
// deftet.py:27): identical multiplication order, no loop.  Kernels take P4 as a template argument.
template <bool P4>
__device__ __forceinline__ float ipow_t(float x, int p) { return P4 ? ((x * x) * x) * x : ipow(x, p); }
template <bool P4>
__device__ __forceinline__ float ipow_m1_t(float x, int p) { return P4 ? (x * x) * x : ipow(x, p - 1); }

// V = -det([A-D; B-D; C-D]) / 6,  deftet.py:247-253
__device__ __forceinline__ float tet_volume(const TetV &v, float *a, float *b, float *c)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] = v.A[k] - v.D[k]; b[k] = v.Bv[k] - v.D[k]; c[k] = v.C[k] - v.D[k]; }
    float bc[3];
    cross3(b, c, bc);
    return -dot3(a, bc) / 6.0f;
}

// AMIPS term of one tet (deftet.py:266-285), optionally its gradient w.r.t. the 12 coordinates
__device__ __forceinline__ float tet_amips(const TetV &v, const float *__restrict__ inv /*9, row-major*/, float scale,
                                           float *grad /*12 or null*/, float gscale)
{
    float off[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        off[0][k] = v.Bv[k] * scale - v.A[k] * scale;
        off[1][k] = v.C[k] * scale - v.A[k] * scale;
        off[2][k] = v.D[k] * scale - v.A[k] * scale;
    }
    float J[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) J[r][c] = off[r][0] * inv[c] + off[r][1] * inv[3 + c] + off[r][2] * inv[6 + c];   // bmm, :278
    float trace = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) trace += J[r][c] * J[r][c];                                                    // :279
    float c12[3], c20[3], c01[3];
    cross3(J[1], J[2], c12);
    cross3(J[2], J[0], c20);
    cross3(J[0], J[1], c01);
    const float det = dot3(J[0], c12);                                                                              // det_m, :281
    const float pos = det >= 0.0f ? 1.f : 0.f;                                                                      // :282
    const float base = det * det + 1e-10f;
    const float bottom = powf(base, -1.0f / 3.0f);                                                                  // :283
    const float energy = trace * bottom * pos;                                                                      // :284
    if (grad) {
        // dE/dJ = pos * (2 J bottom + trace * dbottom/ddet * cof),  dbottom/ddet = -(2/3) det base^(-4/3)
        const float db = -(2.0f / 3.0f) * det * bottom / base;
        float gJ[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gJ[0][c] = pos * (2.f * J[0][c] * bottom + trace * db * c12[c]);
            gJ[1][c] = pos * (2.f * J[1][c] * bottom + trace * db * c20[c]);
            gJ[2][c] = pos * (2.f * J[2][c] * bottom + trace * db * c01[c]);
        }
        // d/d off = gJ @ inv^T ; off rows = (B-A, C-A, D-A) * scale
        float go[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) go[r][k] = (gJ[r][0] * inv[k * 3] + gJ[r][1] * inv[k * 3 + 1] + gJ[r][2] * inv[k * 3 + 2]) * scale * gscale;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            grad[k] += -(go[0][k] + go[1][k] + go[2][k]);
            grad[3 + k] += go[0][k];
            grad[6 + k] += go[1][k];
            grad[9 + k] += go[2][k];
        }
    }
    return energy;
}

// sum of the six edge terms of one tet (deftet.py:326-337)
template <bool P4>
__device__ __forceinline__ float tet_edges(const TetV &v, float scale, int pw, float *grad, float gscale)
{
    const float *P[4] = {v.A, v.Bv, v.C, v.D};
    const int e[6][2] = {{0, 3}, {1, 3}, {2, 3}, {0, 1}, {0, 2}, {1, 2}};          // (A-D),(B-D),(C-D),(A-B),(A-C),(B-C)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float d = P[e[i][0]][k] * scale - P[e[i][1]][k] * scale;
            sum += ipow_t<P4>(d, pw);
            if (grad) {
                const float g = (float)pw * ipow_m1_t<P4>(d, pw) * scale * gscale;
                grad[e[i][0] * 3 + k] += g;
                grad[e[i][1] * 3 + k] -= g;
            }
        }
    return sum;
}

// The two reduction passes run kEParts workgroups of kEThreads threads per shape: few, large workgroups, because every
// workgroup ends with one ticket draw on its shape's counter and same-address atomics serialise at the memory side
// (256 draws per shape cost more than the 8 MB second pass itself; 64 do not show).
constexpr int kEParts = 64, kEThreads = 1024, kEWaves = kEThreads / 64;
constexpr int kETicketShapes = 1024, kETicketSlots = 32;
// zero at load; every launch leaves its slot zero again; one slot per (device, stream) pair (ticket_slot_for_stream)
__device__ int g_energy_tickets[kETicketSlots][2][kETicketShapes];

// exact 64-bit exchange at the memory side (device-scope RMW atomics bypass the per-XCD L2s): what the ticket reductions
// below pass between workgroups that may sit on different XCDs
__device__ __forceinline__ void mem_write_f64(double *p, double v)
{
    const unsigned long long old = atomicExch(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v));
    asm volatile("" ::"v"(old));                                   // keep the RETURNING form: its completion is what s_waitcnt observes
}
__device__ __forceinline__ double mem_read_f64(double *p)
{
    return __longlong_as_double((long long)atomicOr(reinterpret_cast<unsigned long long *>(p), 0ull));
}

// Per-workgroup sums of N doubles -> part[(b * kEParts + blockIdx.x) * N + k].  With slot >= 0 the partials go to the
// memory side and the workgroup draws a ticket; returns true (to every thread) in the one that drew the last ticket of
// shape b, which then finishes the reduction itself — no separate launch.  slot < 0: plain stores, returns false.
template <int N>
__device__ __forceinline__ bool block_reduce_store(double (&v)[N], double *part, int b, int slot, int phase)
{
    __shared__ double sh[kEWaves][N];
    __shared__ int s_last;
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < N; ++k) sh[w][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 64) {                                        // wave 0: one lane per component finishes the sum and hands it over
        if (threadIdx.x < N) {
            double sum = 0.0;
#pragma unroll
            for (int i = 0; i < kEWaves; ++i) sum += sh[i][threadIdx.x];
            double *dst = part + ((size_t)b * kEParts + blockIdx.x) * N + threadIdx.x;
            if (slot >= 0) mem_write_f64(dst, sum);
            else *dst = sum;
        }
        if (slot >= 0) {
            __builtin_amdgcn_s_waitcnt(0);                         // the N exchanges of this wave have returned
            if (threadIdx.x == 0) s_last = atomicAdd(&g_energy_tickets[slot][phase][b], 1) == kEParts - 1;
        }
    }
    if (slot < 0) return false;
    __syncthreads();
    return s_last != 0;
}

// sum over the kEParts partials of shape b of component k: wave 0 of the workgroup, one partial per lane (the same tree
// in the fused and the stand-alone form); every thread of wave 0 gets the result
template <int N>
__device__ __forceinline__ void sum_parts(double *part, int b, bool at_memory, double (&out)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double *src = part + ((size_t)b * kEParts + (threadIdx.x & 63)) * N + k;
        double v = at_memory ? mem_read_f64(src) : *src;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        out[k] = v;
    }
}
static_assert(kEParts == 64, "sum_parts reads one partial per lane of a wave");

// stats[b] = {mean V, amips mean, edge mean, sum (V-mean)^pow, sum pow (V-mean)^(pow-1) / T}
__device__ __forceinline__ void finish_means(double *part, int b, int T, bool at_memory, double *stats)
{
    if (threadIdx.x >= 64) return;
    double v[3];
    sum_parts<3>(part, b, at_memory, v);
    if (threadIdx.x == 0) {
        stats[b * 8 + 0] = v[0] / (double)T;                 // torch.mean(V), :258
        stats[b * 8 + 1] = v[1] / (double)T;                 // torch.mean(energy), :298
        stats[b * 8 + 2] = v[2] / (6.0 * (double)T);         // sum_edge / (6 * T), :338
    }
}

__device__ __forceinline__ void finish_energies(double *part2, int b, int T, bool at_memory, double *stats, float *out)
{
    if (threadIdx.x >= 64) return;
    double v[2];
    sum_parts<2>(part2, b, at_memory, v);
    if (threadIdx.x == 0) {
        stats[b * 8 + 3] = v[0];
        stats[b * 8 + 4] = v[1] / (double)T;                 // mean of d var / d V (enters through mean_v)
        out[b * 3 + 0] = (float)v[0];                        // volume variance
        out[b * 3 + 1] = (float)stats[b * 8 + 1];            // amips
        out[b * 3 + 2] = (float)stats[b * 8 + 2];            // edge length
    }
}

// pass 1: per-workgroup partial sums of (V, amips, edge); the volume of every tet is also kept (4 bytes per tet) so that
// the second pass — sum of (V - mean)^p, which needs the mean first — reads 4 bytes per tet instead of the 48-byte record
template <bool P4>
__attribute__((amdgpu_waves_per_eu(8, 8)))   // 72 -> 64 VGPRs (two dwords of scratch in the P4 instance): forward 48.8 -> 45.6 us (the backward is faster left alone at 82)
__global__ __launch_bounds__(kEThreads) void k_energy_pass1(const float *__restrict__ tet, const float *__restrict__ inv_v, int T,
                                                      float scale, int pow_e, double *part, float *vol, double *stats, int slot)
{
    const int b = blockIdx.y;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const TetV v = load_tet(tet, (size_t)b * T + t);
        float a[3], bb[3], c[3];
        const float V = tet_volume(v, a, bb, c);
        vol[(size_t)b * T + t] = V;
        acc[0] += (double)V;
        if (inv_v) acc[1] += (double)tet_amips(v, inv_v + (size_t)t * 9, scale, nullptr, 0.f);
        acc[2] += (double)tet_edges<P4>(v, scale, pow_e, nullptr, 0.f);
    }
    if (!block_reduce_store<3>(acc, part, b, slot, 0)) return;
    finish_means(part, b, T, true, stats);
    if (threadIdx.x == 0) atomicExch(&g_energy_tickets[slot][0][b], 0);
}

__global__ __launch_bounds__(64) void k_energy_mean(double *part, int T, double *stats) { finish_means(part, blockIdx.x, T, false, stats); }

// pass 2 over the saved volumes (same per-tet values as a second pass over the tets would compute)
template <bool P4>
__global__ __launch_bounds__(kEThreads) void k_energy_pass2(const float *__restrict__ vol, int T, int pow_v, double *stats,
                                                      double *part, float *out, int slot)
{
    const int b = blockIdx.y;
    const float mean = (float)stats[b * 8 + 0];
    double acc[2] = {0.0, 0.0};
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const float d = vol[(size_t)b * T + t] - mean;
        if (!P4 && pow_v == 1) {
            acc[0] += (double)fabsf(d);                                          // :260
            acc[1] += (double)(d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        } else {
            acc[0] += (double)ipow_t<P4>(d, pow_v);                              // :262
            acc[1] += (double)((float)pow_v * ipow_m1_t<P4>(d, pow_v));
        }
    }
    if (!block_reduce_store<2>(acc, part, b, slot, 1)) return;
    finish_energies(part, b, T, true, stats, out);
    if (threadIdx.x == 0) atomicExch(&g_energy_tickets[slot][1][b], 0);
}

__global__ __launch_bounds__(64) void k_energy_final(double *part2, int T, double *stats, float *out)
{
    finish_energies(part2, blockIdx.x, T, false, stats, out);
}

// backward: grad_tet[b,t] = g_var * dvar/dtet + g_amips * damips/dtet + g_edge * dedge/dtet
template <bool P4>
__global__ __launch_bounds__(256) void k_energy_bwd(const float *__restrict__ tet, const float *__restrict__ inv_v, int T,
                                                    float scale, int pow_v, int pow_e, const double *__restrict__ stats,
                                                    const float *__restrict__ gout, float *grad_tet)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const TetV v = load_tet(tet, (size_t)b * T + t);
    float g[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = 0.f;
    const float gv = gout[b * 3 + 0], ga = gout[b * 3 + 1], ge = gout[b * 3 + 2];
    {
        float a[3], bb[3], c[3];
        const float d = tet_volume(v, a, bb, c) - (float)stats[b * 8 + 0];
        float dvar = (!P4 && pow_v == 1) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : (float)pow_v * ipow_m1_t<P4>(d, pow_v);
        dvar -= (float)stats[b * 8 + 4];                      // through mean_v
        // V = -(a . (b x c)) / 6 with a=A-D, b=B-D, c=C-D
        float bc[3], ca[3], ab[3];
        cross3(bb, c, bc);
        cross3(c, a, ca);
        cross3(a, bb, ab);
        const float s = -gv * dvar / 6.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g[k] += s * bc[k];
            g[3 + k] += s * ca[k];
            g[6 + k] += s * ab[k];
            g[9 + k] += -s * (bc[k] + ca[k] + ab[k]);
        }
    }
    if (inv_v) tet_amips(v, inv_v + (size_t)t * 9, scale, g, ga / (float)T);
    tet_edges<P4>(v, scale, pow_e, g, ge / (6.0f * (float)T));
    float4 *dst = reinterpret_cast<float4 *>(grad_tet + ((size_t)b * T + t) * 12);
    dst[0] = make_float4(g[0], g[1], g[2], g[3]);
    dst[1] = make_float4(g[4], g[5], g[6], g[7]);
    dst[2] = make_float4(g[8], g[9], g[10], g[11]);
}

}  // namespace tops
}  // namespace deftet

using namespace deftet;
using namespace deftet::tops;

extern "C" size_t deftet_boundary_index_workspace_bytes(int B, int Fi)
{
    const size_t n = (size_t)(B > 0 ? B : 0) * (size_t)(Fi > 0 ? Fi : 0);
    return align_up(n * 4, 256) * 2 + n * 8 + ((size_t)1 << 20);
}

extern "C" int deftet_boundary_index_i64(const int64_t *face_fx3, const int64_t *tetidx_fx2, const float *occ_bxt,
                                         int64_t *out_rows, int32_t *offsets, int B, int T, int Fi, int mode, void *workspace,
                                         size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T >= 0 && Fi >= 0 && B <= 65535 && (mode == 1 || mode == 2), "bad argument");
    DEFTET_CHECK_ARG((long long)B * Fi < 2147483647LL, "B*F too large");
    hipStream_t st = as_stream(stream_);
    DEFTET_CHECK_ARG(offsets, "null offsets");
    if (B == 0 || Fi == 0) { DEFTET_HIP(hipMemsetAsync(offsets, 0, ((size_t)B + 1) * 4, st)); return DEFTET_OK; }
    DEFTET_CHECK_ARG(face_fx3 && tetidx_fx2 && occ_bxt && out_rows, "null pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && wsb >= deftet_boundary_index_workspace_bytes(B, Fi),
                     "workspace null, misaligned or too small");
    const size_t n = (size_t)B * Fi;
    Arena A(workspace, wsb);
    int *flag = A.take<int>(n), *pos = A.take<int>(n);
    void *tmp = A.base + align_up(A.off, 256);
    const size_t left = wsb - align_up(A.off, 256);
    DEFTET_LAUNCH(k_bnd_flag, dim3((Fi + 255) / 256, B), dim3(256), st, (const long long *)tetidx_fx2, occ_bxt, T, Fi, mode, flag);
    {
        const int rc = prims::scan<int, prims::Plus, true>(flag, pos, n, 0, prims::Plus(), tmp, left, st);
        if (rc != DEFTET_OK) return rc;
    }
    DEFTET_LAUNCH(k_bnd_emit, dim3((Fi + 255) / 256, B), dim3(256), st, (const long long *)face_fx3, (const long long *)tetidx_fx2,
                  occ_bxt, flag, pos, T, Fi, B, mode, (long long *)out_rows, offsets);
    return DEFTET_OK;
}

// B * kEParts * 5 doubles of partial sums (+ with n_tet > 0: one float per tet, the saved volumes of the forward)
extern "C" size_t deftet_tet_energies_workspace_bytes2(int B, int T)
{
    return align_up((size_t)(B > 0 ? B : 0) * kEParts * 5 * 8 + 256, 256) + align_up((size_t)(B > 0 ? B : 0) * (size_t)(T > 0 ? T : 0) * 4, 256);
}
static size_t energies_partials_bytes(int B) { return deftet_tet_energies_workspace_bytes2(B, 0); }

// out f32 [B,3] = {volume_variance(pow_v), amips_energy (0 if inv_v == NULL), edge_length(pow_e)};
// stats f64 [B,8] is kept by the caller for the backward.
extern "C" int deftet_tet_energies_fwd_f32(const float *tet, const float *inv_v, float *out, double *stats, int B, int T,
                                           int pow_v, int pow_e, float scale, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T > 0 && B <= 65535 && pow_v >= 1 && pow_e >= 1 && pow_v <= 16 && pow_e <= 16, "bad argument");
    if (B == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(tet && out && stats && ((uintptr_t)tet & 15) == 0, "null or misaligned pointer");
    DEFTET_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0 && wsb >= deftet_tet_energies_workspace_bytes2(B, T),
                     "workspace null, misaligned or smaller than deftet_tet_energies_workspace_bytes2(B, T)");
    hipStream_t st = as_stream(stream_);
    double *part1 = static_cast<double *>(workspace), *part2 = part1 + (size_t)B * kEParts * 3;
    float *vol = reinterpret_cast<float *>(static_cast<char *>(workspace) + energies_partials_bytes(B));
    // B <= kETicketShapes: two launches, each pass finishes its own reduction on the counters of this stream's slot
    const int slot = B <= kETicketShapes ? ticket_slot_for_stream(st, kETicketSlots) : -1;
    const bool p4 = pow_v == 4 && pow_e == 4;
    if (p4) DEFTET_LAUNCH(k_energy_pass1<true>, dim3(kEParts, B), dim3(kEThreads), st, tet, inv_v, T, scale, pow_e, part1, vol, stats, slot);
    else DEFTET_LAUNCH(k_energy_pass1<false>, dim3(kEParts, B), dim3(kEThreads), st, tet, inv_v, T, scale, pow_e, part1, vol, stats, slot);
    if (slot < 0) DEFTET_LAUNCH(k_energy_mean, dim3(B), dim3(64), st, part1, T, stats);
    if (p4) DEFTET_LAUNCH(k_energy_pass2<true>, dim3(kEParts, B), dim3(kEThreads), st, (const float *)vol, T, pow_v, stats, part2, out, slot);
    else DEFTET_LAUNCH(k_energy_pass2<false>, dim3(kEParts, B), dim3(kEThreads), st, (const float *)vol, T, pow_v, stats, part2, out, slot);
    if (slot < 0) DEFTET_LAUNCH(k_energy_final, dim3(B), dim3(64), st, part2, T, stats, out);
    return DEFTET_OK;
}

extern "C" int deftet_tet_energies_bwd_f32(const float *tet, const float *inv_v, const double *stats, const float *grad_out,
                                           float *grad_tet, int B, int T, int pow_v, int pow_e, float scale, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && T > 0 && B <= 65535 && pow_v >= 1 && pow_e >= 1, "bad argument");
    if (B == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(tet && stats && grad_out && grad_tet && ((uintptr_t)tet & 15) == 0 && ((uintptr_t)grad_tet & 15) == 0,
                     "null or misaligned pointer");
    if (pow_v == 4 && pow_e == 4)
        DEFTET_LAUNCH(k_energy_bwd<true>, dim3((T + 255) / 256, B), dim3(256), as_stream(stream_), tet, inv_v, T, scale, pow_v, pow_e,
                      stats, grad_out, grad_tet);
    else
        DEFTET_LAUNCH(k_energy_bwd<false>, dim3((T + 255) / 256, B), dim3(256), as_stream(stream_), tet, inv_v, T, scale, pow_v, pow_e,
                      stats, grad_out, grad_tet);
    return DEFTET_OK;
}
