// common.cpp — error channel and misc entry points of libdeftet_hip.so
#include "common.hpp"

#include <string.h>

#include <mutex>
#include <vector>

namespace deftet {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-kernel event timing -------------------------------------------------------
namespace {
struct Prof {
    char name[64] = {0};
    bool on = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t used = 0;
} g_prof;
std::mutex g_prof_mu;
}  // namespace

// the selected name matches a kernel of that name and every template instance of it ("k_pix_raster" selects "k_pix_raster<true>")
bool prof_match(const char *name)
{
    if (!g_prof.on) return false;
    const size_t n = strlen(g_prof.name);
    return strncmp(name, g_prof.name, n) == 0 && (name[n] == 0 || name[n] == '<');
}

void prof_begin(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.used + 2 > g_prof.ev.size()) {
        // grow by 64 launches at a time and record every new event once: what the runtime allocates lazily behind an event
        // then exists after the first (warm-up) launch instead of appearing launch by launch inside a timed region
        for (int i = 0; i < 128; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            (void)hipEventRecord(e, st);
            g_prof.ev.push_back(e);
        }
        if (g_prof.used + 2 > g_prof.ev.size()) return;
    }
    (void)hipEventRecord(g_prof.ev[g_prof.used], st);        // timing aid: a failed record only loses a sample
}

void prof_end(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.used + 2 > g_prof.ev.size()) return;
    (void)hipEventRecord(g_prof.ev[g_prof.used + 1], st);
    g_prof.used += 2;
}

int ticket_slot_for_stream(hipStream_t st, int n_slots)
{
    struct Key { int dev; hipStream_t st; };
    static std::mutex mu;
    static std::vector<Key> keys;                             // index = slot
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    // A launch that is being CAPTURED into a hipGraph is not ordered by the stream it was captured on: two graphs captured on
    // one stream may be replayed concurrently on different streams and would then share that stream's counters (a row sum
    // corrupted, an output left unwritten).  Captured launches therefore take the counter-free multi-launch form.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();                                  // (not this launch's error: keep it out of the launch check)
        return -1;
    }
    if (cap != hipStreamCaptureStatusNone) return -1;
    // Slots are never recycled (a stream handle cannot be observed dying): after n_slots distinct (device, stream) pairs of a
    // process every further stream gets -1 for good, i.e. the multi-launch form — slower by one launch, never wrong.
    std::lock_guard<std::mutex> lk(mu);
    for (size_t i = 0; i < keys.size(); ++i)
        if (keys[i].dev == dev && keys[i].st == st) return (int)i;
    if ((int)keys.size() >= n_slots) return -1;
    keys.push_back(Key{dev, st});
    return (int)keys.size() - 1;
}

}  // namespace deftet

// select the kernel to time ("" or NULL switches timing off); resets the accumulated samples
extern "C" int deftet_profile_select(const char *kernel_name)
{
    std::lock_guard<std::mutex> lk(deftet::g_prof_mu);
    deftet::g_prof.used = 0;
    deftet::g_prof.on = kernel_name && kernel_name[0];
    snprintf(deftet::g_prof.name, sizeof(deftet::g_prof.name), "%s", kernel_name ? kernel_name : "");
    return DEFTET_OK;
}

// synchronises the recorded events; total_ms = sum of launch durations, count = launches
extern "C" int deftet_profile_read(double *total_ms, long long *count)
{
    std::lock_guard<std::mutex> lk(deftet::g_prof_mu);
    double tot = 0;
    long long n = 0;
    for (size_t i = 0; i + 1 < deftet::g_prof.used; i += 2) {
        float ms = 0;
        if (hipEventSynchronize(deftet::g_prof.ev[i + 1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, deftet::g_prof.ev[i], deftet::g_prof.ev[i + 1]) != hipSuccess) continue;
        tot += ms;
        ++n;
    }
    deftet::g_prof.used = 0;
    if (total_ms) *total_ms = tot;
    if (count) *count = n;
    return DEFTET_OK;
}

extern "C" int deftet_version(void) { return 221; }   // 221: round 6, second half (8-byte hit records, deftet_point_in_tet_bwd_to_vertices_f32, deftet_put_host_ints); 220: round 6 (deftet_tet_order_coherence_f32); 210: round 5 (the *_ex_* entry points, tet order, query box + misses)

extern "C" const char *deftet_last_error(void) { return deftet::err_buf(); }

extern "C" int deftet_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return deftet::set_error(DEFTET_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
