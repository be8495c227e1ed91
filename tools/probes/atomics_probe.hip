// atomics_probe.hip — how fast are global atomics on gfx950 at each memory scope?
// Measures N returning / non-returning atomicAdd on pseudo-random addresses of a table,
// (a) agent (device) scope from all XCDs, (b) workgroup scope with every table owned by ONE XCD
// (the block reads HW_REG_XCC_ID and only touches the table of its own XCD), plus LDS atomics.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/atomics_probe.hip -o gpurun_out/atomics_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int SCOPE, bool RET>
__global__ __launch_bounds__(256) void k_atomics(int *table, unsigned tableLen, int perThread, int *sink, int *xcdSeen, int ownByXcd)
{
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned x = ownByXcd ? xcc_id() : 0;
    if (threadIdx.x == 0 && xcdSeen) atomicAdd(&xcdSeen[xcc_id() * 16 + (blockIdx.x & 7)], 1);
    int *tb = table + (size_t)x * tableLen;
    int acc = 0;
#pragma unroll 4
    for (int i = 0; i < perThread; ++i) {
        unsigned a = hash(gid * 131u + i) % tableLen;
        if (RET) acc += __hip_atomic_fetch_add(&tb[a], 1, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(&tb[a], 1, __ATOMIC_RELAXED, SCOPE);
    }
    if (RET && acc == -12345) sink[0] = acc;
}

__global__ void k_sum(const int *table, size_t n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += table[i];
    atomicAdd(out, s);
}

template <int SCOPE, bool RET>
static int run(const char *name, int *table, unsigned tableLen, int nTables, int blocks, int perThread, int own, int *sink, int *seen,
               unsigned long long *dsum)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(table, 0, (size_t)tableLen * nTables * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_atomics<SCOPE, RET>), dim3(blocks), dim3(256), 0, 0, table, tableLen, perThread, sink, rep == 0 ? seen : nullptr, own);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipMemset(dsum, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, 0, table, (size_t)tableLen * nTables, dsum);
    unsigned long long s; CK(hipMemcpy(&s, dsum, 8, hipMemcpyDeviceToHost));
    const double n = (double)blocks * 256 * perThread;
    printf("%-44s n=%9.0f  %8.2f us  %7.2f G/s  sum_ok=%d\n", name, n, best * 1e3, n / (best * 1e-3) * 1e-9, (double)s == n);
    return 0;
}

int main()
{
    const unsigned tableLen = 171520 * 8;      // 1.37 M ints = 5.5 MB: the cell table of BASELINE configs[2]
    const int nTables = 8;
    int *table, *sink, *seen; unsigned long long *dsum;
    CK(hipMalloc(&table, (size_t)tableLen * nTables * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&seen, 16 * 16 * 4)); CK(hipMalloc(&dsum, 8));
    CK(hipMemset(seen, 0, 16 * 16 * 4));
    const int blocks = 784, per = 4;             // 800 k atomics, like k_query_bin
    run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, returning, 1 table", table, tableLen, 1, blocks, per, 0, sink, seen, dsum);
    run<__HIP_MEMORY_SCOPE_AGENT, false>("agent scope, no return, 1 table", table, tableLen, 1, blocks, per, 0, sink, nullptr, dsum);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, returning, table per XCD", table, tableLen, nTables, blocks, per, 1, sink, nullptr, dsum);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("workgroup scope, no return, table per XCD", table, tableLen, nTables, blocks, per, 1, sink, nullptr, dsum);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, returning, SHARED (unsafe)", table, tableLen, 1, blocks, per, 0, sink, nullptr, dsum);
    run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope, returning, 8M atomics", table, tableLen, 1, blocks * 10, per, 0, sink, nullptr, dsum);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope, returning, per XCD, 8M", table, tableLen, nTables, blocks * 10, per, 1, sink, nullptr, dsum);
    std::vector<int> h(256);
    CK(hipMemcpy(h.data(), seen, 256 * 4, hipMemcpyDeviceToHost));
    printf("blocks seen per (xcc_id, blockIdx&7):\n");
    for (int x = 0; x < 16; ++x) {
        int tot = 0; for (int j = 0; j < 16; ++j) tot += h[x * 16 + j];
        if (!tot) continue;
        printf("  xcc %2d:", x); for (int j = 0; j < 8; ++j) printf(" %5d", h[x * 16 + j]); printf("\n");
    }
    return 0;
}
