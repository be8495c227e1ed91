// vertex_ops.hip — the gather either side of the per-tet operators:
//   forward   tet_bxfx4x3 = torch.gather(vertice_pos, tetrahedron_bxfx4)   layers/DefTet/deftet.py:65-68
//   backward  grad_pos[b,v] = sum of grad_tet over the (tet, corner) incidences of vertex v
// torch's backward of that gather is a scatter-add with one float atomic per component:
// 12·T atomics per shape = 24.7 M at BASELINE configs[2], and global atomics run at ~25 G/s
// chip-wide on this part.  Here the topology is turned ONCE into a CSR of
// incidences per vertex (stable radix sort ⇒ ascending slot order), and the backward is a
// segmented gather-sum: no atomics, deterministic (fixed summation order, see k_gather_bwd).
#include <cstring>
#include "prims.hpp"

#include "common.hpp"

namespace deftet {
namespace vtx {

// one lane per tet: two 16-byte index loads, four 12-byte vertex gathers, three 16-byte stores
__global__ __launch_bounds__(256) void k_gather_fwd(const float *__restrict__ pos, const int64_t *__restrict__ idx, float *out,
                                                    int V, int T, int idxBatch, int *bad)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const longlong2 *ip = reinterpret_cast<const longlong2 *>(idx + ((idxBatch > 1 ? (size_t)b * T : 0) + t) * 4);
    const longlong2 i01 = ip[0], i23 = ip[1];
    const long long vi[4] = {i01.x, i01.y, i23.x, i23.y};
    float c[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (vi[k] < 0 || vi[k] >= V) {                       // torch.gather raises; here: NaN + flag
            c[3 * k] = c[3 * k + 1] = c[3 * k + 2] = __int_as_float(0x7FC00000);
            if (bad) *bad = 1;
        } else {
            const float *p = pos + ((size_t)b * V + vi[k]) * 3;
            c[3 * k] = p[0]; c[3 * k + 1] = p[1]; c[3 * k + 2] = p[2];
        }
    }
    float4 *o = reinterpret_cast<float4 *>(out + ((size_t)b * T + t) * 12);
    o[0] = make_float4(c[0], c[1], c[2], c[3]);
    o[1] = make_float4(c[4], c[5], c[6], c[7]);
    o[2] = make_float4(c[8], c[9], c[10], c[11]);
}

__global__ __launch_bounds__(256) void k_csr_keys(const int64_t *__restrict__ idx, long long n, long long slotsPerShape, int V,
                                                  unsigned *key, unsigned *val, int *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long bi = i / slotsPerShape, s = i - bi * slotsPerShape;
    const long long v = idx[i];
    const bool ok = v >= 0 && v < V;
    if (!ok) *bad = 1;
    key[i] = ok ? (unsigned)(bi * V + v) : 0xFFFFFFFFu;      // invalid incidences sort to the end and are never referenced
    val[i] = (unsigned)s;
}

// offsets[k] = first sorted position whose key is >= k, for k in [0, nKeys]
__global__ __launch_bounds__(256) void k_csr_offsets(const unsigned *__restrict__ skey, long long n, long long nKeys, int *offsets)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const long long prev = i == 0 ? -1 : (long long)skey[i - 1];
    long long cur = i == n ? nKeys : (long long)skey[i];
    if (cur > nKeys) cur = nKeys;                            // the invalid tail
    for (long long k = prev + 1; k <= cur; ++k) offsets[k] = (int)i;
}

// Four lanes per (shape, vertex): lane k adds the incidences at list positions k, k+4, k+8, ...
// one after the other, then the four partial sums are combined as (s0 + s1) + (s2 + s3).  A fixed
// order (restated by oracle.tet_gather_bwd), four times the memory parallelism of one lane per
// vertex (a vertex has ~24 incidences: one lane per vertex measured 115 us for 99 MB).
// MASKED: the rows come compacted from k_bary_bwd_hits<true> (common.hpp: gather_bwd_rows) — a 64-bit word per 64 tets says
// which tets have one; an absent row is an exact zero and adding it would not change a partial sum, so it is skipped: the
// same additions in the same order as the dense form.  The words of a shape (32 KB at T = 257,250) live in the L2.
template <bool MASKED>
__global__ __launch_bounds__(256) void k_gather_bwd(const float *__restrict__ grad_tet, const int *__restrict__ offsets,
                                                    const int *__restrict__ slots, float *grad_pos, int V, long long slotsPerShape,
                                                    int idxBatch, int accumulate, const unsigned long long *__restrict__ rowMask)
{
    const int b = blockIdx.y;
    // XCD-aware placement (workgroup i runs on XCD i % 8, each XCD has its own L2): every XCD takes one CONTIGUOUS eighth of
    // the vertex range, so the tets shared by vertices of neighbouring workgroups are fetched into one L2 instead of up to
    // four.  The launch rounds the grid up to a multiple of 8.  Speed only; any placement is correct.
    const int per = gridDim.x >> 3, vb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const int gid = vb * blockDim.x + threadIdx.x;
    const int v = gid >> 2, k = gid & 3;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (v < V) {
        const size_t row = (idxBatch > 1 ? (size_t)b * V : 0) + v;
        const int s0 = offsets[row], s1 = offsets[row + 1];
        const float *g = grad_tet + (size_t)b * slotsPerShape * 3;
        int i = s0 + k;
        if (MASKED) {
            const unsigned long long *mk = rowMask + (size_t)b * (size_t)((slotsPerShape / 4 + 63) >> 6);
            // slot s = 4 t + corner -> the float offset of the corner in the compacted rows, or -1 when tet t has no row
            auto where = [&](int s) -> long long {
                const int t = s >> 2;
                const unsigned long long w = mk[t >> 6], bit = 1ull << (t & 63);
                return (w & bit) ? ((long long)((t & ~63) + __popcll(w & (bit - 1ull))) * 4 + (s & 3)) * 3 : -1ll;
            };
            // four incidences in flight per lane (slot -> mask word -> row is a chain of three loads; a vertex of a Kuhn grid has
            // ~22 incidences, five or six per lane); the additions keep the order of the dense form: i, i + 4, i + 8, ...
            for (; i < s1; i += 16) {
                long long o[4];
                float r[4][3];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = i + 4 * j < s1 ? where(slots[i + 4 * j]) : -1ll;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o[j] >= 0) { r[j][0] = g[o[j]]; r[j][1] = g[o[j] + 1]; r[j][2] = g[o[j] + 2]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (o[j] >= 0) { ax += r[j][0]; ay += r[j][1]; az += r[j][2]; }
            }
        } else {
            for (; i + 4 < s1; i += 8) {                     // two independent gathers in flight per lane
                const float *p = g + (size_t)slots[i] * 3, *q = g + (size_t)slots[i + 4] * 3;
                const float px = p[0], py = p[1], pz = p[2], qx = q[0], qy = q[1], qz = q[2];
                ax += px; ay += py; az += pz;
                ax += qx; ay += qy; az += qz;
            }
            if (i < s1) {
                const float *p = g + (size_t)slots[i] * 3;
                ax += p[0]; ay += p[1]; az += p[2];
            }
        }
    }
    // (s0 + s1) + (s2 + s3); lanes of one vertex are adjacent, the whole wave takes part
    ax += __shfl_xor(ax, 1); ay += __shfl_xor(ay, 1); az += __shfl_xor(az, 1);
    ax += __shfl_xor(ax, 2); ay += __shfl_xor(ay, 2); az += __shfl_xor(az, 2);
    if (v < V && k == 0) {
        float *o = grad_pos + ((size_t)b * V + v) * 3;
        if (accumulate) { ax += o[0]; ay += o[1]; az += o[2]; }
        o[0] = ax; o[1] = ay; o[2] = az;
    }
}

static size_t sort_tmp_bytes(size_t n) { return prims::radix_sort_temp_bytes<unsigned, unsigned>(n); }

}  // namespace vtx
}  // namespace deftet

using namespace deftet;

extern "C" int deftet_tet_gather_fwd_f32(const float *pos, const int64_t *tet_idx, float *out, int32_t *bad_flag, int B, int V,
                                         int T, int idx_batch, void *stream_)
{
    DEFTET_CHECK_ARG(B >= 0 && V >= 0 && T >= 0, "negative size");
    DEFTET_CHECK_ARG(idx_batch == 1 || idx_batch == B, "tet_idx batch must be 1 or n_batch (got %d)", idx_batch);
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds 65535", B);
    if (B == 0 || T == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(pos && tet_idx && out, "null pointer");
    DEFTET_CHECK_ARG(((uintptr_t)tet_idx & 15) == 0 && ((uintptr_t)out & 15) == 0, "tet_idx/out must be 16-byte aligned");
    DEFTET_LAUNCH(vtx::k_gather_fwd, dim3((T + 255) / 256, B), dim3(256), as_stream(stream_), pos, tet_idx, out, V, T, idx_batch, bad_flag);
    return DEFTET_OK;
}

extern "C" size_t deftet_tet_vertex_csr_workspace_bytes(int idx_batch, int V, int T)
{
    if (idx_batch <= 0 || V < 0 || T < 0) return 0;
    const size_t n = (size_t)idx_batch * T * 4;
    return 3 * align_up(n * 4, 256) + align_up(vtx::sort_tmp_bytes(n), 256) + 256;
}

extern "C" int deftet_tet_vertex_csr_i32(const int64_t *tet_idx, int32_t *offsets, int32_t *slots, int32_t *bad_flag,
                                         int idx_batch, int V, int T, void *workspace, size_t workspace_bytes, void *stream_)
{
    DEFTET_CHECK_ARG(idx_batch >= 1 && V >= 0 && T >= 0, "bad size");
    DEFTET_CHECK_ARG((long long)idx_batch * V < 0xFFFFFFFFLL && (long long)idx_batch * T * 4 < 0x7FFFFFFFLL, "topology too large for 32-bit keys");
    DEFTET_CHECK_ARG(offsets && bad_flag && (T == 0 || (tet_idx && slots)), "null pointer");
    hipStream_t st = as_stream(stream_);
    const long long n = (long long)idx_batch * T * 4, nKeys = (long long)idx_batch * V;
    DEFTET_HIP(hipMemsetAsync(bad_flag, 0, 4, st));
    if (n == 0) {
        DEFTET_HIP(hipMemsetAsync(offsets, 0, (size_t)(nKeys + 1) * 4, st));
        return DEFTET_OK;
    }
    DEFTET_CHECK_ARG(workspace && workspace_bytes >= deftet_tet_vertex_csr_workspace_bytes(idx_batch, V, T) &&
                         ((uintptr_t)workspace & 255) == 0,
                     "workspace null, misaligned or too small");
    Arena A(workspace, workspace_bytes);
    unsigned *key = A.take<unsigned>((size_t)n), *val = A.take<unsigned>((size_t)n), *skey = A.take<unsigned>((size_t)n);
    size_t tmpBytes = vtx::sort_tmp_bytes((size_t)n);
    void *tmp = A.take<char>(tmpBytes);
    const unsigned gb = (unsigned)((n + 255) / 256);
    DEFTET_LAUNCH(vtx::k_csr_keys, dim3(gb), dim3(256), st, tet_idx, n, (long long)T * 4, V, key, val, bad_flag);
    // all 32 key bits: invalid incidences carry the key 0xFFFFFFFF and must sort behind every vertex
    const int rc = prims::radix_sort<unsigned, unsigned>(key, skey, val, reinterpret_cast<unsigned *>(slots), (size_t)n, 32, tmp, tmpBytes, st);
    if (rc != DEFTET_OK) return rc;
    DEFTET_LAUNCH(vtx::k_csr_offsets, dim3((unsigned)((n + 256) / 256)), dim3(256), st, skey, n, nKeys, offsets);
    return DEFTET_OK;
}

int deftet::vtx::gather_bwd_rows(const float *rows, const unsigned long long *rowMask, const int32_t *offsets, const int32_t *slots,
                                 float *grad_pos, int B, int V, int T, int idx_batch, int accumulate, hipStream_t st)
{
    DEFTET_CHECK_ARG(B >= 0 && V >= 0 && T >= 0, "negative size");
    DEFTET_CHECK_ARG(idx_batch == 1 || idx_batch == B, "CSR batch must be 1 or n_batch (got %d)", idx_batch);
    DEFTET_CHECK_ARG(B <= 65535, "n_batch=%d exceeds 65535", B);
    if (B == 0 || V == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(offsets && grad_pos && (T == 0 || (rows && slots)), "null pointer");
    const dim3 grid((((V + 63) / 64 + 7) / 8) * 8, B);
    if (rowMask) DEFTET_LAUNCH(vtx::k_gather_bwd<true>, grid, dim3(256), st, rows, offsets, slots, grad_pos, V, (long long)T * 4, idx_batch, accumulate, rowMask);
    else DEFTET_LAUNCH(vtx::k_gather_bwd<false>, grid, dim3(256), st, rows, offsets, slots, grad_pos, V, (long long)T * 4, idx_batch, accumulate, rowMask);
    return DEFTET_OK;
}

extern "C" int deftet_tet_gather_bwd_f32(const float *grad_tet, const int32_t *offsets, const int32_t *slots, float *grad_pos, int B,
                                         int V, int T, int idx_batch, int accumulate, void *stream_)
{
    return vtx::gather_bwd_rows(grad_tet, nullptr, offsets, slots, grad_pos, B, V, T, idx_batch, accumulate, as_stream(stream_));
}
