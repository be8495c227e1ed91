// prims.hpp — this library's own device-wide primitives for gfx950: stable LSD radix sort (keys / key-value pairs) and
// prefix scans.  They replace the rocPRIM calls of the builders, the vertex operators, the boundary extraction and
// check_sign (rounds 1-2): every instantiated rocPRIM sort brought 1-2 MB of kernels into libdeftet_hip.so (20 MB in
// round 2), and the sorts are part of what this library is supposed to have written itself.
//
//   radix_sort<K, V>(kin, kout, vin, vout, n, bits, tmp, tmp_bytes, stream)      K: u32 / u64, V: 4- or 8-byte values, or
//   radix_sort_keys<K>(kin, kout, n, bits, tmp, tmp_bytes, stream)               keys only; kin is not modified
//   scan<T, Op, EXCLUSIVE>(in, out, n, identity, tmp, tmp_bytes, stream)         in == out allowed
//
// Sort: 8 bits per pass, (bits + 7) / 8 passes.  Per pass: k_rs_hist (LDS histogram of the digit per 2,048-key tile ->
// hist[tile][digit]), an exclusive scan of that table in (digit, tile) order (= where the keys of (digit, tile) start;
// k_cs_partials / k_cs_spine / k_cs_apply: coalesced rows only), k_rs_scatter.  Sorts of up
// to kFusedTiles tiles (1 M keys) skip the scan launches: the table is written tile-major and every scatter workgroup sums
// the rows of the earlier tiles itself (two launches per pass instead of three to five).  The
// scatter is stable without atomics: a tile is four waves x eight rounds x 64 lanes in key order; per round the lanes
// with equal digits find each other with eight ballots (match-any), the first of them advances the wave's own counter
// of that digit, so a key's rank among the wave's keys of its digit is known after one walk; one pass over the 4 x 256
// counters turns them into rank bases inside the tile; the tile is staged in digit order in LDS and written out with
// consecutive threads on consecutive addresses of every digit's run.
// Temporary storage: one spare key (and value) array for the ping-pong, the two digit tables and the scan's partials —
// radix_sort_temp_bytes<K, V>(n).
#pragma once
#include "common.hpp"

namespace deftet {
namespace prims {

constexpr int kTileThreads = 256, kTileItems = 8, kTile = kTileThreads * kTileItems;   // 2,048 elements per workgroup
#ifndef PRIMS_FUSED_TILES
#define PRIMS_FUSED_TILES 512
#endif
constexpr unsigned kFusedTiles = PRIMS_FUSED_TILES;       // sorts of up to this many tiles (1 M keys) run two launches per pass (no scan)

// ---------------------------------------------------------------------------- scans
struct Plus {
    template <typename T>
    __host__ __device__ T operator()(T a, T b) const { return a + b; }
};
struct Max {
    template <typename T>
    __host__ __device__ T operator()(T a, T b) const { return a > b ? a : b; }
};

template <typename T, typename Op>
__device__ __forceinline__ T wave_incl(T v, Op op, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T t = __shfl_up(v, off);
        if (lane >= off) v = op(v, t);
    }
    return v;
}

// tile totals: part[tile] = op over the tile's elements
template <typename T, typename Op>
__global__ __launch_bounds__(kTileThreads) void k_scan_partials(const T *__restrict__ in, size_t n, T identity, Op op, T *part)
{
    __shared__ T sh[kTileThreads / 64];
    const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kTileItems;
    T v = identity;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k)
        if (base + k < n) v = op(v, in[base + k]);
    const int lane = threadIdx.x & 63;
    v = wave_incl(v, op, lane);
    if (lane == 63) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = sh[0];
        for (int w = 1; w < kTileThreads / 64; ++w) t = op(t, sh[w]);
        part[blockIdx.x] = t;
    }
}

// exclusive scan of the tile totals by ONE workgroup (a few thousand tiles at most for the sizes of this library)
template <typename T, typename Op>
__global__ __launch_bounds__(1024) void k_scan_spine(T *part, size_t nt, T identity, Op op)
{
    __shared__ T sh[16];
    __shared__ T s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = identity;
    __syncthreads();
    for (size_t base = 0; base < nt; base += 1024) {
        const size_t i = base + threadIdx.x;
        const T x = i < nt ? part[i] : identity;
        const T incl = wave_incl(x, op, lane);
        if (lane == 63) sh[w] = incl;
        __syncthreads();
        T pre = s_carry, tot = identity;
        for (int k = 0; k < 16; ++k) {
            if (k < w) pre = op(pre, sh[k]);
            tot = op(tot, sh[k]);
        }
        // exclusive value of element i: carry + earlier waves + earlier lanes
        const T before = __shfl_up(incl, 1);
        if (i < nt) part[i] = lane == 0 ? pre : op(pre, before);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = op(s_carry, tot);
        __syncthreads();
    }
}

template <typename T, typename Op, bool EXCLUSIVE>
__global__ __launch_bounds__(kTileThreads) void k_scan_apply(const T *in, T *out, size_t n, T identity, Op op, const T *__restrict__ part)
{
    __shared__ T sh[kTileThreads / 64];
    const size_t base = (size_t)blockIdx.x * kTile + (size_t)threadIdx.x * kTileItems;
    T x[kTileItems];
    T v = identity;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        x[k] = base + k < n ? in[base + k] : identity;
        v = op(v, x[k]);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T incl = wave_incl(v, op, lane);
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    T run = part[blockIdx.x];
    for (int k = 0; k < w; ++k) run = op(run, sh[k]);
    const T before = __shfl_up(incl, 1);
    if (lane > 0) run = op(run, before);
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        if (EXCLUSIVE) {
            if (base + k < n) out[base + k] = run;
            run = op(run, x[k]);
        } else {
            run = op(run, x[k]);
            if (base + k < n) out[base + k] = run;
        }
    }
}

// short inputs: the whole scan by ONE workgroup of 1,024 threads in one launch (eight elements per thread and trip, a running
// carry between the trips).  The three-launch form above costs ~16 us of launches for a table of a few thousand entries
// — the rasterizer's tile sort scans thirteen such tables per step.
constexpr size_t kScanSmall = 8 * 1024;                     // (one trip; four trips over 32 K entries measured slower than the three launches)
template <typename T, typename Op, bool EXCLUSIVE>
__global__ __launch_bounds__(1024) void k_scan_small(const T *in, T *out, size_t n, T identity, Op op)
{
    __shared__ T sh[16];
    __shared__ T s_carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = identity;
    __syncthreads();
    for (size_t base = 0; base < n; base += 1024 * 8) {
        const size_t i0 = base + (size_t)threadIdx.x * 8;
        T x[8];
        T v = identity;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x[k] = i0 + k < n ? in[i0 + k] : identity;
            v = op(v, x[k]);
        }
        const T incl = wave_incl(v, op, lane);
        if (lane == 63) sh[w] = incl;
        __syncthreads();
        T run = s_carry, tot = identity;
        for (int k = 0; k < 16; ++k) {
            if (k < w) run = op(run, sh[k]);
            tot = op(tot, sh[k]);
        }
        const T before = __shfl_up(incl, 1);
        if (lane > 0) run = op(run, before);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (EXCLUSIVE) {
                if (i0 + k < n) out[i0 + k] = run;
                run = op(run, x[k]);
            } else {
                run = op(run, x[k]);
                if (i0 + k < n) out[i0 + k] = run;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = op(s_carry, tot);
        __syncthreads();
    }
}

inline size_t scan_tiles(size_t n) { return (n + kTile - 1) / kTile; }
template <typename T>
inline size_t scan_temp_bytes(size_t n) { return align_up((scan_tiles(n) + 1) * sizeof(T), 256); }

template <typename T, typename Op, bool EXCLUSIVE>
int scan(const T *in, T *out, size_t n, T identity, Op op, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    if (n == 0) return DEFTET_OK;
    if (!tmp || tmp_bytes < scan_temp_bytes<T>(n)) return set_error(DEFTET_EINVAL, "scan: temporary storage too small");
    if (n <= kScanSmall) {
        DEFTET_LAUNCH((k_scan_small<T, Op, EXCLUSIVE>), dim3(1), dim3(1024), st, in, out, n, identity, op);
        return DEFTET_OK;
    }
    T *part = static_cast<T *>(tmp);
    const size_t nt = scan_tiles(n);
    if (nt > 0x7FFFFFFFull) return set_error(DEFTET_ELIMIT, "scan: too many elements");
    DEFTET_LAUNCH((k_scan_partials<T, Op>), dim3((unsigned)nt), dim3(kTileThreads), st, in, n, identity, op, part);
    DEFTET_LAUNCH((k_scan_spine<T, Op>), dim3(1), dim3(1024), st, part, nt, identity, op);
    DEFTET_LAUNCH((k_scan_apply<T, Op, EXCLUSIVE>), dim3((unsigned)nt), dim3(kTileThreads), st, in, out, n, identity, op, (const T *)part);
    return DEFTET_OK;
}

// ---------------------------------------------------------------------------- radix sort
// where a pass reads its keys / values from: an array, or (first pass only) anything indexable — e.g. keys derived from
// another tensor, or the identity as values — so that such inputs need not be written out first
template <typename T>
struct PtrLoad {
    const T *p;
    __device__ __forceinline__ T operator()(size_t i) const { return p[i]; }
};
struct IotaLoad {
    __device__ __forceinline__ unsigned operator()(size_t i) const { return (unsigned)i; }
};

// n_dev (may be NULL): the number of elements actually present, known on the device only (n is then the capacity the
// grid is sized for); workgroups beyond it find nothing to do, so the cost follows the real count.
// The table is tile-major, hist[tile][digit]: a workgroup writes ONE 1 KB row.  (Rounds 3-4 wrote hist[digit][tile] for a flat
// scan: 256 four-byte stores per workgroup, each to another line — 2.1 M scattered stores per pass of the 16.8 M-pair sort,
// which took 50 us to read 67 MB — and the scatter read its 256 offsets back the same way.)
template <typename K, typename KL>
__global__ __launch_bounds__(kTileThreads) void k_rs_hist(KL keys, size_t n, int shift, unsigned dmask, unsigned nblk, unsigned *hist,
                                                          const int *__restrict__ n_dev)
{
    if (n_dev) n = min(n, (size_t)max(*n_dev, 0));
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kTile;
#pragma unroll
    for (int k = 0; k < kTileItems; ++k) {
        const size_t i = base + (size_t)k * kTileThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys(i) >> shift) & dmask], 1u);
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// Exclusive scan of the tile-major table in (digit, tile) order — where the keys of (digit, tile) start in the output — with
// coalesced rows only.  Thread d owns column d everywhere: chunk sums of kColChunk rows, one workgroup that turns the
// chunk sums into chunk bases (exclusive over the chunks of a digit, plus the exclusive scan of the digits' totals),
// and the rows of every chunk rewritten as running offsets.
constexpr unsigned kColChunk = 64;
static __global__ __launch_bounds__(256) void k_cs_partials(const unsigned *__restrict__ hist, unsigned nblk, unsigned *part)
{
    const unsigned t0 = blockIdx.x * kColChunk, t1 = min(nblk, t0 + kColChunk), d = threadIdx.x;
    unsigned sum = 0u;
    for (unsigned j0 = t0; j0 < t1; j0 += 16) {
        unsigned a[16];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) a[k] = hist[(size_t)min(j0 + k, t1 - 1u) * 256 + d];      // clamped, unconditional
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) sum += j0 + k < t1 ? a[k] : 0u;
    }
    part[(size_t)blockIdx.x * 256 + d] = sum;
}
// one workgroup of 1,024 threads: thread (q, d) owns a quarter of the chunks of digit d (reads and writes in batches of 16,
// all loads of a batch in flight; a plain loop of read-modify-writes over 128 chunks took 14 us)
static __global__ __launch_bounds__(1024) void k_cs_spine(unsigned *part, unsigned nchunk)
{
    __shared__ unsigned qsum[4][256], wtot[4];
    const unsigned d = threadIdx.x & 255u, q = threadIdx.x >> 8, lane = threadIdx.x & 63u;
    const unsigned per = (nchunk + 3u) / 4u, c0 = min(nchunk, q * per), c1 = min(nchunk, c0 + per);
    unsigned total = 0u;
    for (unsigned j0 = c0; j0 < c1; j0 += 16) {
        unsigned a[16];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) a[k] = part[(size_t)min(j0 + k, c1 - 1u) * 256 + d];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) total += j0 + k < c1 ? a[k] : 0u;
    }
    qsum[q][d] = total;
    __syncthreads();
    unsigned before = 0u, dtot = 0u;                                // earlier quarters of this digit; the digit's total
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
        const unsigned v = qsum[k][d];
        before += k < q ? v : 0u;
        dtot += v;
    }
    unsigned incl = dtot;                                           // exclusive scan of the totals over the digits (every quarter
#pragma unroll                                                      // computes it: waves 0-3 of a quarter hold digits 0-255)
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(incl, off);
        if (lane >= (unsigned)off) incl += t;
    }
    if (q == 0 && lane == 63u) wtot[d >> 6] = incl;
    __syncthreads();
    unsigned run = incl - dtot + before;
    for (unsigned k = 0; k < (d >> 6); ++k) run += wtot[k];
    for (unsigned j0 = c0; j0 < c1; j0 += 16) {
        unsigned a[16];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) a[k] = part[(size_t)min(j0 + k, c1 - 1u) * 256 + d];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k)
            if (j0 + k < c1) {
                part[(size_t)(j0 + k) * 256 + d] = run;             // (own column, own quarter: read above, nobody else touches it)
                run += a[k];
            }
    }
}
static __global__ __launch_bounds__(256) void k_cs_apply(const unsigned *__restrict__ hist, unsigned nblk, const unsigned *__restrict__ part,
                                                  unsigned *offs)
{
    const unsigned t0 = blockIdx.x * kColChunk, t1 = min(nblk, t0 + kColChunk), d = threadIdx.x;
    unsigned run = part[(size_t)blockIdx.x * 256 + d];
    for (unsigned j0 = t0; j0 < t1; j0 += 16) {
        unsigned a[16];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) a[k] = hist[(size_t)min(j0 + k, t1 - 1u) * 256 + d];
#pragma unroll
        for (unsigned k = 0; k < 16; ++k)
            if (j0 + k < t1) {
                offs[(size_t)(j0 + k) * 256 + d] = run;
                run += a[k];
            }
    }
}

// FUSED is a template argument: the 32 rows its column sums keep in flight are 32 more registers (100 instead of 68), which
// as a run-time branch cost the large sorts — the 16.8 M-pair hit sort of the rasterizer's backward — a workgroup per CU
template <typename K, typename V, bool HAS_V, typename KL, typename VL, bool FUSED>
__global__ __launch_bounds__(kTileThreads) void k_rs_scatter(KL kin, K *kout, VL vin, V *vout, size_t n,
                                                             int shift, unsigned dmask, unsigned nblk, const unsigned *__restrict__ offs,
                                                             const int *__restrict__ n_dev)
{
    constexpr bool fused = FUSED;
    if (n_dev) n = min(n, (size_t)max(*n_dev, 0));
    if ((size_t)blockIdx.x * kTile >= n) return;
    __shared__ unsigned cnt[kTileThreads / 64][256];                // per wave and digit: count, then rank base inside the tile
    __shared__ unsigned lbase[256], gbase[256], wtot[kTileThreads / 64], wtot2[kTileThreads / 64];
    __shared__ K s_key[kTile];
    __shared__ V s_val[HAS_V ? kTile : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kTileThreads / 64; ++k) cnt[k][threadIdx.x] = 0u;
    __syncthreads();
    // wave w owns the tile's keys [w * 512, (w + 1) * 512) in rounds of 64: tile order = (wave, round, lane)
    const size_t tile0 = (size_t)blockIdx.x * kTile, base = tile0 + (size_t)w * (kTileItems * 64);
    K key[kTileItems];
    V val[kTileItems];
    unsigned dig[kTileItems], rank[kTileItems];
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const size_t i = base + (size_t)r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? kin(i) : K(0);
        if (HAS_V) val[r] = valid ? vin(i) : V(0);
        const unsigned d = (unsigned)(key[r] >> shift) & dmask;
        dig[r] = d;
        unsigned long long m = __ballot(valid);                     // lanes of this round with MY digit (match-any by ballots)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const unsigned below = __popcll(m & ((1ull << lane) - 1ull));
        const unsigned seen = cnt[w][d];                            // keys of digit d in the wave's earlier rounds
        rank[r] = seen + below;
        if (valid && below == 0u) cnt[w][d] = seen + (unsigned)__popcll(m);   // one lane per distinct digit, after every lane has read
    }
    __syncthreads();
    {   // thread d: the tile's keys of digit d start at lbase[d] inside the tile (exclusive scan over the digits) and at gbase[d]
        // in the output (the scanned table); the wave counters become rank bases inside the tile
        const unsigned d = threadIdx.x;
        unsigned tot = 0u;
#pragma unroll
        for (int k = 0; k < kTileThreads / 64; ++k) tot += cnt[k][d];
        unsigned incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        // fused: `offs` is the UNSCANNED table hist[tile][digit] of k_rs_hist: thread d sums its column over the earlier tiles
        // (where the tile's keys of digit d start inside the digit's run) and over all tiles (the digit's total); an exclusive
        // scan of the totals over the digits gives the run's start.  A few hundred coalesced 1 KB rows from L2 per workgroup
        // instead of three scan launches per pass (small sorts: the launches were most of the pass).
        unsigned below = 0u, total = 0u, gincl = 0u;
        if (fused) {
            const unsigned ntile = (unsigned)((n + kTile - 1) / kTile);          // tiles that hold keys (n is the device-side count)
            constexpr unsigned kInFlight = 32;                                   // rows per trip, all loads issued before the first is used
            for (unsigned j0 = 0; j0 < ntile; j0 += kInFlight) {
                unsigned a[kInFlight];
#pragma unroll
                for (unsigned k = 0; k < kInFlight; ++k) a[k] = offs[(size_t)min(j0 + k, ntile - 1u) * 256 + d];   // clamped, unconditional
#pragma unroll
                for (unsigned k = 0; k < kInFlight; ++k) {
                    const unsigned v = j0 + k < ntile ? a[k] : 0u;
                    total += v;
                    below += j0 + k < blockIdx.x ? v : 0u;
                }
            }
            gincl = total;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned t = __shfl_up(gincl, off);
                if (lane >= off) gincl += t;
            }
            if (lane == 63) wtot2[w] = gincl;
        }
        if (lane == 63) wtot[w] = incl;
        __syncthreads();
        unsigned run = incl - tot;
        for (int k = 0; k < w; ++k) run += wtot[k];
        lbase[d] = run;
        if (fused) {
            unsigned g = gincl - total + below;
            for (int k = 0; k < w; ++k) g += wtot2[k];
            gbase[d] = g;
        } else {
            gbase[d] = offs[(size_t)blockIdx.x * 256 + d];           // the column scan's table is tile-major as well
        }
#pragma unroll
        for (int k = 0; k < kTileThreads / 64; ++k) {
            const unsigned t = cnt[k][d];
            cnt[k][d] = run;
            run += t;
        }
    }
    __syncthreads();
    // stage the tile in digit order in LDS (stable), then write it out with consecutive threads on consecutive addresses
    // inside every digit's run — placing the keys straight from the registers made every lane of a store go to a
    // different digit's run (4- or 8-byte segments): the 16.8 M-pair sort of the rasterizer's backward took 0.16 ms per pass
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const size_t i = base + (size_t)r * 64 + lane;
        if (i < n) {
            const unsigned lp = cnt[w][dig[r]] + rank[r];
            s_key[lp] = key[r];
            if (HAS_V) s_val[lp] = val[r];
        }
    }
    __syncthreads();
    const unsigned nTile = (unsigned)min((size_t)kTile, n - tile0);
#pragma unroll
    for (int r = 0; r < kTileItems; ++r) {
        const unsigned j = (unsigned)r * kTileThreads + threadIdx.x;
        if (j < nTile) {
            const K k = s_key[j];
            const unsigned d = (unsigned)(k >> shift) & dmask;
            const size_t pos = (size_t)gbase[d] + (j - lbase[d]);
            kout[pos] = k;
            if (HAS_V) vout[pos] = s_val[j];
        }
    }
}

template <typename K, typename V>
inline size_t radix_sort_temp_bytes(size_t n, bool has_values = true)
{
    const size_t nblk = (n + kTile - 1) / kTile;
    return align_up(n * sizeof(K), 256) + (has_values ? align_up(n * sizeof(V), 256) : 0) + 2 * align_up(256 * nblk * 4 + 4, 256) +
           align_up(256 * ((nblk + kColChunk - 1) / kColChunk + 1) * 4, 256) + 256;
}

template <typename K, typename V, bool HAS_V, typename KL, typename VL>
int radix_sort_impl(KL kin, K *kout, VL vin, V *vout, size_t n, int bits, void *tmp, size_t tmp_bytes, hipStream_t st, const int *n_dev = nullptr)
{
    if (n == 0) return DEFTET_OK;
    if (n > 0xFFFFFFFFull) return set_error(DEFTET_ELIMIT, "radix_sort: more than 2^32 - 1 elements");
    if (!tmp || ((uintptr_t)tmp & 255) != 0 || tmp_bytes < radix_sort_temp_bytes<K, V>(n, HAS_V))
        return set_error(DEFTET_EINVAL, "radix_sort: temporary storage missing, misaligned or too small");
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;
    const unsigned nblk = (unsigned)((n + kTile - 1) / kTile);
    Arena A(tmp, tmp_bytes);
    K *tk = A.take<K>(n);
    V *tv = HAS_V ? A.take<V>(n) : nullptr;
    unsigned *hist = A.take<unsigned>((size_t)256 * nblk + 1), *offs = A.take<unsigned>((size_t)256 * nblk + 1);
    const unsigned nchunk = (nblk + kColChunk - 1) / kColChunk;
    unsigned *part = A.take<unsigned>((size_t)256 * (nchunk + 1));
    const K *sk = nullptr;
    const V *sv = nullptr;
    // sorts of up to kFusedTiles tiles skip the scan: the scatter sums the histogram table itself (see k_rs_scatter)
    const int fused = nblk <= kFusedTiles ? 1 : 0;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;            // the last pass lands in kout / vout
        K *dk = to_out ? kout : tk;
        V *dv = to_out ? vout : tv;
        const int left = bits - p * 8;                               // the last digit may be narrower: bits above `bits` do not count
        const unsigned dmask = left >= 8 || bits <= 0 ? 255u : (1u << left) - 1u;
        if (p == 0) DEFTET_LAUNCH((k_rs_hist<K, KL>), dim3(nblk), dim3(kTileThreads), st, kin, n, 0, dmask, nblk, hist, n_dev);
        else DEFTET_LAUNCH((k_rs_hist<K, PtrLoad<K>>), dim3(nblk), dim3(kTileThreads), st, PtrLoad<K>{sk}, n, p * 8, dmask, nblk, hist, n_dev);
        const unsigned *table = hist;
        if (!fused) {
            DEFTET_LAUNCH(k_cs_partials, dim3(nchunk), dim3(256), st, (const unsigned *)hist, nblk, part);
            DEFTET_LAUNCH(k_cs_spine, dim3(1), dim3(1024), st, part, nchunk);
            DEFTET_LAUNCH(k_cs_apply, dim3(nchunk), dim3(256), st, (const unsigned *)hist, nblk, (const unsigned *)part, offs);
            table = offs;
        }
        if (p == 0 && fused)
            DEFTET_LAUNCH((k_rs_scatter<K, V, HAS_V, KL, VL, true>), dim3(nblk), dim3(kTileThreads), st, kin, dk, vin, dv, n, 0, dmask, nblk,
                          table, n_dev);
        else if (p == 0)
            DEFTET_LAUNCH((k_rs_scatter<K, V, HAS_V, KL, VL, false>), dim3(nblk), dim3(kTileThreads), st, kin, dk, vin, dv, n, 0, dmask, nblk,
                          table, n_dev);
        else if (fused)
            DEFTET_LAUNCH((k_rs_scatter<K, V, HAS_V, PtrLoad<K>, PtrLoad<V>, true>), dim3(nblk), dim3(kTileThreads), st, PtrLoad<K>{sk}, dk,
                          PtrLoad<V>{sv}, dv, n, p * 8, dmask, nblk, table, n_dev);
        else
            DEFTET_LAUNCH((k_rs_scatter<K, V, HAS_V, PtrLoad<K>, PtrLoad<V>, false>), dim3(nblk), dim3(kTileThreads), st, PtrLoad<K>{sk}, dk,
                          PtrLoad<V>{sv}, dv, n, p * 8, dmask, nblk, table, n_dev);
        sk = dk;
        sv = dv;
    }
    return DEFTET_OK;
}

template <typename K, typename V>
int radix_sort(const K *kin, K *kout, const V *vin, V *vout, size_t n, int bits, void *tmp, size_t tmp_bytes, hipStream_t st,
               const int *n_dev = nullptr)
{
    static_assert(sizeof(V) == 4 || sizeof(V) == 8, "values of 4 or 8 bytes");
    return radix_sort_impl<K, V, true>(PtrLoad<K>{kin}, kout, PtrLoad<V>{vin}, vout, n, bits, tmp, tmp_bytes, st, n_dev);
}

// the same with the first pass reading its keys / values through loaders (see PtrLoad / IotaLoad)
template <typename K, typename V, typename KL, typename VL>
int radix_sort_from(KL kin, K *kout, VL vin, V *vout, size_t n, int bits, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    return radix_sort_impl<K, V, true, KL, VL>(kin, kout, vin, vout, n, bits, tmp, tmp_bytes, st);
}

template <typename K>
int radix_sort_keys(const K *kin, K *kout, size_t n, int bits, void *tmp, size_t tmp_bytes, hipStream_t st)
{
    return radix_sort_impl<K, unsigned, false>(PtrLoad<K>{kin}, kout, PtrLoad<unsigned>{nullptr}, (unsigned *)nullptr, n, bits, tmp, tmp_bytes, st);
}

}  // namespace prims
}  // namespace deftet
