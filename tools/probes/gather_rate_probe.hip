// gather_rate_probe.hip — what does a divergent per-lane gather cost on gfx950, by width?
// Every lane reads `steps` records of W dwords (W = 1, 2, 3, 4) from a 12.8 MB table (L2-resident, like sortedQ of one
// shape) at addresses that mimic the traversal kernels: groups of 6 consecutive lanes share an address (the six tets of a
// cube), neighbouring groups are a few records apart, every step jumps to another pseudo-random place.  Record stride is
// 16 bytes for W = 1, 2, 4 (aligned) and 12 bytes for W = 3.  Prints ns per wave-instruction per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/gather_rate_probe.hip -o tools/probes/bin/gather_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int W, int SHARE>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ table, unsigned nrec, int steps, float *out)
{
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const unsigned grp = lane / SHARE;                     // lanes of a group read the same record
    const unsigned strideB = W == 3 ? 12u : 16u;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        const unsigned base = hash(wave * 977u + s) % (nrec - 4096u);
        const unsigned rec = base + grp * 5u + (hash(wave + s * 31u + grp) & 3u);
        const char *p = reinterpret_cast<const char *>(table) + (size_t)rec * strideB;
        if (W == 1) acc += *reinterpret_cast<const float *>(p);
        if (W == 2) { const float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
        if (W == 3) { struct F3 { float x, y, z; }; const F3 v = *reinterpret_cast<const F3 *>(p); acc += v.x + v.y + v.z; }
        if (W == 4) { const float4 v = *reinterpret_cast<const float4 *>(p); acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int W, int SHARE>
static int run(const float *table, unsigned nrec, float *out, int nCU)
{
    const int blocks = nCU * 24, steps = 64;               // 96 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<W, SHARE>), dim3(blocks), dim3(256), 0, 0, table, nrec, steps, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<W, SHARE>), dim3(blocks), dim3(256), 0, 0, table, nrec, steps, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instrPerCU = (double)blocks * 4 * steps / nCU;
    printf("{\"width_dwords\": %d, \"lanes_sharing_an_address\": %d, \"ms\": %.4f, \"ns_per_wave_instr_per_CU\": %.2f}\n", W, SHARE, ms,
           ms * 1e6 / instrPerCU);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nCU = prop.multiProcessorCount;
    const unsigned nrec = 800000;
    float *table, *out;
    CK(hipMalloc(&table, (size_t)nrec * 16 + 65536));
    CK(hipMemset(table, 0, (size_t)nrec * 16 + 65536));
    CK(hipMalloc(&out, 64));
    if (run<1, 6>(table, nrec, out, nCU)) return 1;
    if (run<2, 6>(table, nrec, out, nCU)) return 1;
    if (run<3, 6>(table, nrec, out, nCU)) return 1;
    if (run<4, 6>(table, nrec, out, nCU)) return 1;
    if (run<1, 1>(table, nrec, out, nCU)) return 1;
    if (run<2, 1>(table, nrec, out, nCU)) return 1;
    if (run<4, 1>(table, nrec, out, nCU)) return 1;
    if (run<4, 64>(table, nrec, out, nCU)) return 1;
    return 0;
}
