// valu_rate_probe.hip — issue rate of the VALU instructions the traversal kernels are made of, on gfx950:
// v_fma_f32, v_pk_fma_f32, v_cndmask_b32, v_min3_f32, v_cmp_lt_f32 — N independent dependency chains per lane,
// all 256 CUs busy, wave64.  Prints wave-instructions per ns per CU and cycles per wave-instruction per SIMD
// (at the clock hipDeviceProp reports).  Build: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o gpurun_out/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kChains = 8;

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, float seed)
{
    float a[kChains];
    f32x2 p[kChains];
    const float m = seed * 0.999f, c = seed * 1e-3f;
    __shared__ float4 s_q[64];
    float4 q4[2] = {};
    const unsigned ldsAddr = (unsigned)(size_t)&s_q[(int)seed & 63];
    if (threadIdx.x < 64) s_q[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    asm volatile("s_mov_b32 s30, 5" ::: "s30");
#pragma unroll
    for (int i = 0; i < kChains; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 0.5f}; }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{m, m}), "v"(f32x2{c, c}));
            if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : );
            if (KIND == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{m, m}));
            if (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(m), "v"(c) : "vcc");
            if (KIND == 7) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
            if (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
            if (KIND == 10) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 13) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 14) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 15) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 16) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 17) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 18) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            // round 6: what the rasterizer's face loop is made of
            if (KIND == 19) asm volatile("v_readlane_b32 s20, %0, s30" : : "v"(a[i]) : "s20");                       // lane select in an SGPR
            if (KIND == 20) asm volatile("v_readlane_b32 s20, %1, s30\n\tv_add_f32 %0, s20, %0" : "+v"(a[i]) : "v"(m) : "s20");   // + a VALU reader of the SGPR
            if (KIND == 21) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 22) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc");
            if (KIND == 23) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (KIND == 24) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 25) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(m) : "s20", "s21");
            if (KIND == 26) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[i & 1]) : "v"(ldsAddr) : "memory");          // wave-uniform address: LDS broadcast
            if (KIND == 27) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[i]) : "s20");
            if (KIND == 28) { a[i] = a[i] / m; }                                                                     // the compiler's IEEE division sequence
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kChains; ++i) s += a[i] + p[i].x + p[i].y;
    if (KIND == 26) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); s += q4[0].x + q4[1].y; }
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
static int run(const char *name, float *out, double clkGHz, int nCU)
{
    const int blocks = nCU * 8;                       // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double waveInstr = (double)blocks * 4 * kIters * kChains;
    const double perSimd = waveInstr / (nCU * 4.0);
    printf("{\"instr\": \"%s\", \"ms\": %.4f, \"wave_instr_per_simd\": %.0f, \"ns_per_wave_instr_per_simd\": %.3f, \"cycles_at_%.2fGHz\": %.2f}\n",
           name, ms, perSimd, ms * 1e6 / perSimd, clkGHz, ms * 1e6 / perSimd * clkGHz);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e-6;
    const int nCU = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"CUs\": %d, \"clock_GHz\": %.3f}\n", prop.name, nCU, clk);
    float *out;
    CK(hipMalloc(&out, 64));
    if (run<0>("v_fma_f32", out, clk, nCU)) return 1;
    if (run<1>("v_pk_fma_f32", out, clk, nCU)) return 1;
    if (run<5>("v_pk_mul_f32", out, clk, nCU)) return 1;
    if (run<2>("v_cndmask_b32", out, clk, nCU)) return 1;
    if (run<3>("v_min3_f32", out, clk, nCU)) return 1;
    if (run<4>("v_cmp_lt_f32", out, clk, nCU)) return 1;
    if (run<6>("v_mad_u64_u32", out, clk, nCU)) return 1;
    if (run<7>("v_mul_lo_u32", out, clk, nCU)) return 1;
    if (run<8>("v_cndmask_b32_e64 (sgpr pair mask)", out, clk, nCU)) return 1;
    if (run<16>("v_cndmask_b32 (no dst dependency)", out, clk, nCU)) return 1;
    if (run<9>("v_mov_b32", out, clk, nCU)) return 1;
    if (run<10>("v_addc_co_u32", out, clk, nCU)) return 1;
    if (run<11>("v_add_u32", out, clk, nCU)) return 1;
    if (run<12>("v_max_f32", out, clk, nCU)) return 1;
    if (run<13>("v_lshl_or_b32", out, clk, nCU)) return 1;
    if (run<14>("v_bfi_b32", out, clk, nCU)) return 1;
    if (run<15>("v_fmac_f32", out, clk, nCU)) return 1;
    if (run<17>("v_sub_f32", out, clk, nCU)) return 1;
    if (run<18>("v_mul_f32", out, clk, nCU)) return 1;
    if (run<19>("v_readlane_b32 (sgpr lane)", out, clk, nCU)) return 1;
    if (run<20>("v_readlane_b32 + v_add_f32 reading it (2 instr)", out, clk, nCU)) return 1;
    if (run<27>("v_readfirstlane_b32", out, clk, nCU)) return 1;
    if (run<21>("v_rcp_f32", out, clk, nCU)) return 1;
    if (run<22>("v_div_scale_f32", out, clk, nCU)) return 1;
    if (run<23>("v_div_fmas_f32", out, clk, nCU)) return 1;
    if (run<24>("v_div_fixup_f32", out, clk, nCU)) return 1;
    if (run<25>("v_cmp_le_f32_e64 (sgpr pair)", out, clk, nCU)) return 1;
    if (run<26>("ds_read_b128 (uniform address)", out, clk, nCU)) return 1;
    if (run<28>("a / m (IEEE division sequence)", out, clk, nCU)) return 1;
    return 0;
}
