// common.hpp — shared helpers for libdeftet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deftet_hip.h"

namespace deftet {

// thread-local last-error message (deftet_last_error)
char *err_buf();
int set_error(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define DEFTET_CHECK_ARG(cond, ...)                                  \
    do {                                                              \
        if (!(cond)) return deftet::set_error(DEFTET_EINVAL, __VA_ARGS__); \
    } while (0)

#define DEFTET_HIP(call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return deftet::set_error(DEFTET_ELAUNCH, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define DEFTET_LAUNCH_CHECK(name)                                                          \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess)                                                               \
            return deftet::set_error(DEFTET_ELAUNCH, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// Optional per-kernel timing (deftet_profile_select / deftet_profile_read): when a kernel
// name is selected, every launch of that kernel is bracketed by a pair of hipEvents recorded
// on the launch stream.  Off by default; costs one strcmp-free pointer compare per launch.
bool prof_match(const char *name);
void prof_begin(hipStream_t st);
void prof_end(hipStream_t st);

#define DEFTET_LAUNCH(kern, grid, block, stream, ...)                          \
    do {                                                                       \
        const bool prof_ = deftet::prof_match(#kern);                          \
        if (prof_) deftet::prof_begin(stream);                                 \
        hipLaunchKernelGGL(kern, grid, block, 0, stream, __VA_ARGS__);         \
        if (prof_) deftet::prof_end(stream);                                   \
        DEFTET_LAUNCH_CHECK(#kern);                                            \
    } while (0)

// same, with dynamic LDS bytes
#define DEFTET_LAUNCH_SHM(kern, grid, block, shm, stream, ...)                 \
    do {                                                                       \
        const bool prof_ = deftet::prof_match(#kern);                          \
        if (prof_) deftet::prof_begin(stream);                                 \
        hipLaunchKernelGGL(kern, grid, block, shm, stream, __VA_ARGS__);       \
        if (prof_) deftet::prof_end(stream);                                   \
        DEFTET_LAUNCH_CHECK(#kern);                                            \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char *base;
    size_t off, cap;
    Arena(void *p, size_t bytes) : base(static_cast<char *>(p)), off(0), cap(bytes) {}
    template <typename T>
    T *take(size_t n) {
        off = align_up(off, 256);
        T *r = reinterpret_cast<T *>(base + off);
        off += n * sizeof(T);
        return r;
    }
    bool ok() const { return off <= cap; }
};

constexpr int kWave = 64;  // CDNA4 wavefront

// Ticket reductions (reduce.hip, tet_ops.hip) finish inside one launch by counting workgroups on device-global counters.
// Two launches may only share a set of counters if they cannot run at the same time: launches on ONE stream are ordered, so
// every (device, stream) pair gets its own slot of counters, at most `n_slots` of them per process; -1 when they are used up
// (the caller then takes its multi-launch form, which needs no counters).
int ticket_slot_for_stream(hipStream_t st, int n_slots);

// vertex_ops.hip: grad_pos[b,v] = sum over the (tet, corner) incidences of vertex v of the rows of a [B,T,4,3] gradient, in the
// incidence CSR's order (k_gather_bwd).  rowMask == nullptr: dense rows.  Otherwise the rows are in the COMPACTED form
// k_bary_bwd_hits<true> writes (point_in_tet.hip): rowMask[b][t / 64] has bit t % 64 set iff tet t has a row, stored at row
// (t & ~63) + popcount(mask below the bit); absent rows are zero and are skipped (x + 0 = x: the same sums, bit for bit).
namespace vtx {
int gather_bwd_rows(const float *rows, const unsigned long long *rowMask, const int32_t *offsets, const int32_t *slots, float *grad_pos,
                    int B, int V, int T, int idx_batch, int accumulate, hipStream_t st);
}

}  // namespace deftet
