// prims.hip — C entry points of the library's own device-wide primitives (prims.hpp): what the operators sort and scan
// with, exported so that it can be tested by itself (tests/test_prims_gpu.py) and used by a caller that needs the same
// stable order the operators produce.
#pragma clang fp contract(off)
#include "prims.hpp"

using namespace deftet;

extern "C" size_t deftet_radix_sort_workspace_bytes(long long n, int key_bytes, int value_bytes)
{
    if (n < 0 || (key_bytes != 4 && key_bytes != 8) || (value_bytes != 0 && value_bytes != 4 && value_bytes != 8)) return 0;
    const size_t N = (size_t)n;
    if (key_bytes == 4) return value_bytes == 8 ? prims::radix_sort_temp_bytes<unsigned, unsigned long long>(N) : prims::radix_sort_temp_bytes<unsigned, unsigned>(N, value_bytes != 0);
    return value_bytes == 8 ? prims::radix_sort_temp_bytes<unsigned long long, unsigned long long>(N) : prims::radix_sort_temp_bytes<unsigned long long, unsigned>(N, value_bytes != 0);
}

// Stable ascending sort of unsigned keys (4 or 8 bytes) on their low `bits` bits, optionally carrying values of 4 or 8
// bytes.  keys_in / values_in are not modified; the outputs must not alias the inputs.  n_dev (device pointer, may be
// NULL): only the first min(n, *n_dev) elements exist.
extern "C" int deftet_radix_sort(const void *keys_in, void *keys_out, const void *values_in, void *values_out, long long n, int key_bytes,
                                 int value_bytes, int bits, const int32_t *n_dev, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(n >= 0 && (key_bytes == 4 || key_bytes == 8) && (value_bytes == 0 || value_bytes == 4 || value_bytes == 8), "bad size");
    DEFTET_CHECK_ARG(bits >= 1 && bits <= key_bytes * 8, "bits=%d out of range for %d-byte keys", bits, key_bytes);
    if (n == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(keys_in && keys_out && keys_in != keys_out && (value_bytes == 0 || (values_in && values_out && values_in != values_out)),
                     "null or aliased pointer");
    DEFTET_CHECK_ARG(workspace && wsb >= deftet_radix_sort_workspace_bytes(n, key_bytes, value_bytes), "workspace null or too small");
    hipStream_t st = as_stream(stream_);
    typedef unsigned u32;
    typedef unsigned long long u64;
    const size_t N = (size_t)n;
    if (key_bytes == 4) {
        if (value_bytes == 0) return prims::radix_sort_impl<u32, u32, false>(prims::PtrLoad<u32>{(const u32 *)keys_in}, (u32 *)keys_out, prims::PtrLoad<u32>{nullptr}, (u32 *)nullptr, N, bits, workspace, wsb, st, n_dev);
        if (value_bytes == 4) return prims::radix_sort<u32, u32>((const u32 *)keys_in, (u32 *)keys_out, (const u32 *)values_in, (u32 *)values_out, N, bits, workspace, wsb, st, n_dev);
        return prims::radix_sort<u32, u64>((const u32 *)keys_in, (u32 *)keys_out, (const u64 *)values_in, (u64 *)values_out, N, bits, workspace, wsb, st, n_dev);
    }
    if (value_bytes == 0) return prims::radix_sort_impl<u64, u32, false>(prims::PtrLoad<u64>{(const u64 *)keys_in}, (u64 *)keys_out, prims::PtrLoad<u32>{nullptr}, (u32 *)nullptr, N, bits, workspace, wsb, st, n_dev);
    if (value_bytes == 4) return prims::radix_sort<u64, u32>((const u64 *)keys_in, (u64 *)keys_out, (const u32 *)values_in, (u32 *)values_out, N, bits, workspace, wsb, st, n_dev);
    return prims::radix_sort<u64, u64>((const u64 *)keys_in, (u64 *)keys_out, (const u64 *)values_in, (u64 *)values_out, N, bits, workspace, wsb, st, n_dev);
}

extern "C" size_t deftet_scan_workspace_bytes(long long n, int elem_bytes)
{
    if (n < 0) return 0;
    return elem_bytes == 8 ? prims::scan_temp_bytes<long long>((size_t)n) : prims::scan_temp_bytes<int>((size_t)n);
}

// kind 0: exclusive sum, 1: inclusive sum, 2: inclusive running maximum; int32 (elem_bytes 4) or int64 (8) elements;
// in == out allowed.
extern "C" int deftet_scan(const void *in, void *out, long long n, int elem_bytes, int kind, void *workspace, size_t wsb, void *stream_)
{
    DEFTET_CHECK_ARG(n >= 0 && (elem_bytes == 4 || elem_bytes == 8) && kind >= 0 && kind <= 2, "bad argument");
    if (n == 0) return DEFTET_OK;
    DEFTET_CHECK_ARG(in && out && workspace && ((uintptr_t)workspace & 15) == 0, "null or misaligned pointer");
    hipStream_t st = as_stream(stream_);
    const size_t N = (size_t)n;
    if (elem_bytes == 4) {
        const int *i = (const int *)in;
        int *o = (int *)out;
        if (kind == 0) return prims::scan<int, prims::Plus, true>(i, o, N, 0, prims::Plus(), workspace, wsb, st);
        if (kind == 1) return prims::scan<int, prims::Plus, false>(i, o, N, 0, prims::Plus(), workspace, wsb, st);
        return prims::scan<int, prims::Max, false>(i, o, N, (int)0x80000000, prims::Max(), workspace, wsb, st);
    }
    const long long *i = (const long long *)in;
    long long *o = (long long *)out;
    if (kind == 0) return prims::scan<long long, prims::Plus, true>(i, o, N, 0LL, prims::Plus(), workspace, wsb, st);
    if (kind == 1) return prims::scan<long long, prims::Plus, false>(i, o, N, 0LL, prims::Plus(), workspace, wsb, st);
    return prims::scan<long long, prims::Max, false>(i, o, N, (long long)0x8000000000000000ULL, prims::Max(), workspace, wsb, st);
}
