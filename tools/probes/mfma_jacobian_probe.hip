// MFMA experiment for the per-tet Jacobian of the AMIPS energy (J = off(3x3) . inv_v(3x3), deftet.py:283-290):
// the same quantity — the squared Frobenius norm of J of every tet — computed
//   valu : one lane per tet, 27 FMAs in registers (what k_energy_pass1 / k_energy_bwd do),
//   mfma : v_mfma_f32_4x4x1_16b_f32, 16 tets per wave-instruction, 4 lanes per tet (lane (tet, i) feeds row i of off and
//          column i of inv_v, padded 3 -> 4), three instructions (k = 0..2) per 16 tets,
// in two regimes: `mem` (one pass over the records, the product's regime) and `alu` (the Jacobian recomputed `rep` times
// from registers with a data dependence through the inputs, so the arithmetic pipes are what is timed).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_jacobian_probe.hip && /tmp/mfma_probe
//
// Prints one JSON line; the two kernels' outputs are compared (max relative difference) before timing.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));         \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- one lane per tet ----
template <int REP>
__global__ __launch_bounds__(256) void k_valu(const float *__restrict__ tet, const float *__restrict__ inv_v, int T, int N,
                                              float scale, float *__restrict__ out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 *src = reinterpret_cast<const float4 *>(tet + (size_t)n * 12);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2];
    const float A[3] = {r0.x, r0.y, r0.z}, Bv[3] = {r0.w, r1.x, r1.y}, C[3] = {r1.z, r1.w, r2.x}, D[3] = {r2.y, r2.z, r2.w};
    const float *iv = inv_v + (size_t)(n % T) * 9;
    float inv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[k] = iv[k];
    float off[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        off[0][k] = (Bv[k] - A[k]) * scale;
        off[1][k] = (C[k] - A[k]) * scale;
        off[2][k] = (D[k] - A[k]) * scale;
    }
    float fro = 0.f;
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        float f = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float Jij = off[i][0] * inv[j] + off[i][1] * inv[3 + j] + off[i][2] * inv[6 + j];
                f += Jij * Jij;
            }
        fro += f;
        if (REP > 1) off[0][0] += f * 1e-30f;                       // carries a dependence from one repetition to the next
    }
    out[n] = fro;
}

// ---- four lanes per tet, J through the matrix pipe ----
template <int REP>
__global__ __launch_bounds__(256) void k_mfma(const float *__restrict__ tet, const float *__restrict__ inv_v, int T, int N,
                                              float scale, float *__restrict__ out)
{
    const int lane4 = blockIdx.x * blockDim.x + threadIdx.x;        // 4 lanes per tet
    const int n = lane4 >> 2, i = lane4 & 3;
    const bool live = n < N && i < 3;
    const int nc = n < N ? n : N - 1;
    const float *p = tet + (size_t)nc * 12;
    // lane i: row i of off = (vertex i+1 - vertex 0) * scale; lane 3: the zero padding row
    float a[3], b[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v0 = p[k], vi = p[(i < 3 ? i + 1 : 0) * 3 + k];
        a[k] = live ? (vi - v0) * scale : 0.f;
        b[k] = live ? inv_v[(size_t)(nc % T) * 9 + k * 3 + i] : 0.f;  // column i of inv_v: B operand of step k
    }
    float fro = 0.f;
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], acc, 0, 0, 0);
        // lane (tet, j) now holds column j of J (rows in acc[0..3]); Frobenius: own column, then across the 4 lanes
        float f = acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2];
        f += __shfl_xor(f, 1);
        f += __shfl_xor(f, 2);
        fro += f;
        if (REP > 1) a[0] += f * 1e-30f;
    }
    if (n < N && i == 0) out[n] = fro;
}

template <typename F>
static float time_ms(F launch, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) launch();
    CK(hipEventRecord(e0));
    for (int k = 0; k < iters; ++k) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main()
{
    const int T = 257250, B = 8, N = T * B;                          // BASELINE configs[2]: res 70, batch 8
    const float scale = 70.f;
    std::vector<float> h_tet((size_t)N * 12), h_inv((size_t)T * 9);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : h_tet) v = rnd();
    for (auto &v : h_inv) v = rnd() * 2.f - 1.f;
    float *tet, *inv, *o1, *o2;
    CK(hipMalloc(&tet, h_tet.size() * 4));
    CK(hipMalloc(&inv, h_inv.size() * 4));
    CK(hipMalloc(&o1, (size_t)N * 4));
    CK(hipMalloc(&o2, (size_t)N * 4));
    CK(hipMemcpy(tet, h_tet.data(), h_tet.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(inv, h_inv.data(), h_inv.size() * 4, hipMemcpyHostToDevice));
    const int gv = (N + 255) / 256, gm = (N * 4 + 255) / 256;
    hipLaunchKernelGGL(k_valu<1>, dim3(gv), dim3(256), 0, 0, tet, inv, T, N, scale, o1);
    hipLaunchKernelGGL(k_mfma<1>, dim3(gm), dim3(256), 0, 0, tet, inv, T, N, scale, o2);
    CK(hipDeviceSynchronize());
    std::vector<float> h1(N), h2(N);
    CK(hipMemcpy(h1.data(), o1, (size_t)N * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), o2, (size_t)N * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int k = 0; k < N; ++k) worst = std::fmax(worst, std::fabs((double)h1[k] - h2[k]) / std::fmax(1e-30, std::fabs((double)h1[k])));
    constexpr int R = 256;
    const float valu_mem = time_ms([&] { hipLaunchKernelGGL(k_valu<1>, dim3(gv), dim3(256), 0, 0, tet, inv, T, N, scale, o1); }, 50);
    const float mfma_mem = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(gm), dim3(256), 0, 0, tet, inv, T, N, scale, o2); }, 50);
    const float valu_alu = time_ms([&] { hipLaunchKernelGGL(k_valu<R>, dim3(gv), dim3(256), 0, 0, tet, inv, T, N, scale, o1); }, 10);
    const float mfma_alu = time_ms([&] { hipLaunchKernelGGL(k_mfma<R>, dim3(gm), dim3(256), 0, 0, tet, inv, T, N, scale, o2); }, 10);
    const double bytes = (double)N * (48 + 4) + (double)T * 36;
    std::printf("{\"probe\": \"mfma_jacobian\", \"n_tet\": %d, \"batch\": %d, \"max_rel_diff\": %.3g, "
                "\"mem_regime_us\": {\"valu\": %.1f, \"mfma_4x4x1\": %.1f}, \"mem_regime_GBs\": {\"valu\": %.0f, \"mfma_4x4x1\": %.0f}, "
                "\"alu_regime_rep\": %d, \"alu_regime_us\": {\"valu\": %.1f, \"mfma_4x4x1\": %.1f}, "
                "\"alu_regime_ns_per_tet_jacobian\": {\"valu\": %.4f, \"mfma_4x4x1\": %.4f}}\n",
                T, B, worst, valu_mem * 1e3, mfma_mem * 1e3, bytes / (valu_mem * 1e-3) / 1e9, bytes / (mfma_mem * 1e-3) / 1e9, R,
                valu_alu * 1e3, mfma_alu * 1e3, valu_alu * 1e6 / ((double)N * R), mfma_alu * 1e6 / ((double)N * R));
    return 0;
}
