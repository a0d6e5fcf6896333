/*
 * runtime.cpp — device / stream ownership for libsaf_hip.
 * All work of the library is enqueued on ONE stream per process (operators are
 * called one at a time per handle, like the reference: SURVEY §8b "Threading").
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"
#include <mutex>
#include <sched.h>

namespace saf {

/* Operators are called from several host threads (initCodec on a worker thread while process runs on the audio thread,
 * SURVEY §8b "Threading"): the first call of each may arrive together, so device check and stream creation are serialised. */
static std::mutex g_rt_mutex;
static hipStream_t g_stream = nullptr;
static bool g_own_stream = false;
static bool g_checked = false;

static void ensure_device_locked()
{
    if (g_checked) return;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        SAF_FATAL("no usable HIP device (hipGetDeviceCount: %s, %d devices). libsaf_hip has no CPU fallback; "
                  "run on an MI355X (gfx950).", hipGetErrorString(e), n);
    g_checked = true;
}

void ensure_device()
{
    std::lock_guard<std::mutex> lk(g_rt_mutex);
    ensure_device_locked();
}

/* A calling thread may redirect the library's launches to a stream of its own for the duration of a call (StreamScope): the
 * host-pointer ambi_dec_process gives every handle its own stream, so that N host threads driving N handles do not queue behind
 * one another on the process-wide stream. */
static thread_local hipStream_t t_stream = nullptr;
StreamScope::StreamScope(hipStream_t s) : prev(t_stream) { if (s) t_stream = s; }
StreamScope::~StreamScope() { t_stream = prev; }
bool on_private_stream() { return t_stream != nullptr; }
/* End of a host-pointer call: wait for the handle's stream.  hipStreamSynchronize spins; with more calling threads than the
 * process may use CPUs (a 16-CPU cgroup driving 32 handles) the spinning threads take the CPUs from the ones that have work to
 * launch.  So: poll for the time a lone call needs, then give the CPU away between polls. */
void wait_stream(hipStream_t s)
{
    for (int i = 0; i < 4000; i++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIP_CHECK(e);
        if (i >= 64) sched_yield();
    }
    HIP_CHECK(hipStreamSynchronize(s));
}

hipStream_t new_stream()
{
    ensure_device();
    hipStream_t ns = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
    return ns;
}

hipStream_t stream()
{
    if (t_stream) return t_stream;
    /* fast path: the pointer is written once under the lock (and by set_stream, which callers do not race with launches) */
    hipStream_t s = __atomic_load_n(&g_stream, __ATOMIC_ACQUIRE);
    if (s) return s;
    std::lock_guard<std::mutex> lk(g_rt_mutex);
    if (!g_stream) {
        ensure_device_locked();
        hipStream_t ns = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        g_own_stream = true;
        __atomic_store_n(&g_stream, ns, __ATOMIC_RELEASE);
    }
    return g_stream;
}

/* The side stream carries kernels that run BESIDE one on the main stream (the decode kernel beside the filterbank equaliser of
 * ambi_dec).  fork: the side stream waits for everything enqueued on the main stream so far; join: the main stream waits for
 * everything enqueued on the side stream so far.  One event each, re-recorded per use (HIP events may be reused once the wait
 * that names them has been enqueued). */
static hipStream_t g_side = nullptr;
static hipEvent_t g_ev_fork = nullptr, g_ev_join = nullptr;
hipStream_t side_stream()
{
    hipStream_t s = __atomic_load_n(&g_side, __ATOMIC_ACQUIRE);
    if (s) return s;
    (void)stream();
    std::lock_guard<std::mutex> lk(g_rt_mutex);
    if (!g_side) {
        hipStream_t ns = nullptr;
        HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&g_ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming));
        __atomic_store_n(&g_side, ns, __ATOMIC_RELEASE);
    }
    return g_side;
}
void side_fork()
{
    hipStream_t s2 = side_stream();
    HIP_CHECK(hipEventRecord(g_ev_fork, stream()));
    HIP_CHECK(hipStreamWaitEvent(s2, g_ev_fork, 0));
}
void side_join()
{
    hipStream_t s2 = side_stream();
    HIP_CHECK(hipEventRecord(g_ev_join, s2));
    HIP_CHECK(hipStreamWaitEvent(stream(), g_ev_join, 0));
}

void set_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_rt_mutex);
    ensure_device_locked();
    if (g_side) HIP_CHECK(hipStreamSynchronize(g_side));
    if (g_stream && g_own_stream) { HIP_CHECK(hipStreamSynchronize(g_stream)); HIP_CHECK(hipStreamDestroy(g_stream)); }
    g_own_stream = false;
    hipStream_t ns = s;
    if (!ns) { HIP_CHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking)); g_own_stream = true; }
    __atomic_store_n(&g_stream, ns, __ATOMIC_RELEASE);
}

/* ---- host-pointer entry points: zero-copy I/O ---- */
static int g_zero_copy = []() { const char* e = getenv("SAF_HIP_ZERO_COPY"); return e ? atoi(e) != 0 : 1; }();
bool zero_copy_io() { return g_zero_copy != 0; }

/* ---- per-kernel timing ---- */
struct ProfRec { const char* name; hipEvent_t a, b; hipStream_t s; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;
static std::mutex g_prof_mutex;          /* launches may come from several host threads while profiling is on */

KernelTimer::KernelTimer(const char* name, hipStream_t on) : slot(-1)
{
    if (!g_prof) return;
    ProfRec r; r.name = name; r.s = on ? on : stream();
    HIP_CHECK(hipEventCreate(&r.a)); HIP_CHECK(hipEventCreate(&r.b));
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    HIP_CHECK(hipEventRecord(r.a, r.s));
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
KernelTimer::~KernelTimer()
{
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    if (slot < (int)g_recs.size()) HIP_CHECK(hipEventRecord(g_recs[slot].b, g_recs[slot].s));
}

}  // namespace saf

extern "C" {

void saf_hip_profile_enable(int on) { saf::g_prof = on != 0; }
void saf_hip_profile_reset(void)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    std::lock_guard<std::mutex> lk(saf::g_prof_mutex);
    for (auto& r : saf::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    saf::g_recs.clear();
}
/* Sums the recorded launches of kernel `name`; returns the launch count. */
int saf_hip_profile_read(const char* name, double* total_ms)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    std::lock_guard<std::mutex> lk(saf::g_prof_mutex);
    int n = 0; double t = 0.0;
    for (auto& r : saf::g_recs)
        if (!strcmp(r.name, name)) { float ms = 0.f; HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b)); t += ms; n++; }
    if (total_ms) *total_ms = t;
    return n;
}

/* a stop-watch on the library stream (bench.py corroborates its host-clock timing with it) */
static hipEvent_t g_sw0 = nullptr, g_sw1 = nullptr;
void saf_hip_stopwatch_start(void)
{
    if (!g_sw0) { HIP_CHECK(hipEventCreate(&g_sw0)); HIP_CHECK(hipEventCreate(&g_sw1)); }
    HIP_CHECK(hipEventRecord(g_sw0, saf::stream()));
}
double saf_hip_stopwatch_stop_ms(void)
{
    if (!g_sw0) return 0.0;
    HIP_CHECK(hipEventRecord(g_sw1, saf::stream()));
    HIP_CHECK(hipEventSynchronize(g_sw1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, g_sw0, g_sw1));
    return (double)ms;
}

void saf_hip_setZeroCopyIO(int enable) { saf::g_zero_copy = enable ? 1 : 0; }
int saf_hip_getZeroCopyIO(void) { return saf::g_zero_copy; }
void saf_hip_set_stream(void* hipStream) { saf::set_stream((hipStream_t)hipStream); }
void* saf_hip_get_stream(void) { return (void*)saf::stream(); }
void saf_hip_synchronize(void) { HIP_CHECK(hipStreamSynchronize(saf::stream())); }
int saf_hip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
void saf_hip_set_device(int dev) { HIP_CHECK(hipSetDevice(dev)); }
const char* saf_hip_version(void) { return "saf_hip 0.1 (gfx950; SAF 1.3.0 hot-path API)"; }

}
