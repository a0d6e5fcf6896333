/*
 * runtime.cpp — device / stream ownership for libsaf_hip.
 * All work of the library is enqueued on ONE stream per process (operators are
 * called one at a time per handle, like the reference: SURVEY §8b "Threading").
 */
#include "saf_hip_common.h"
#include "../../include/saf_hip.h"

namespace saf {

static hipStream_t g_stream = nullptr;
static bool g_own_stream = false;
static bool g_checked = false;

void ensure_device()
{
    if (g_checked) return;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        SAF_FATAL("no usable HIP device (hipGetDeviceCount: %s, %d devices). libsaf_hip has no CPU fallback; "
                  "run on an MI355X (gfx950).", hipGetErrorString(e), n);
    g_checked = true;
}

hipStream_t stream()
{
    if (!g_stream) {
        ensure_device();
        HIP_CHECK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        g_own_stream = true;
    }
    return g_stream;
}

void set_stream(hipStream_t s)
{
    ensure_device();
    if (g_stream && g_own_stream) { HIP_CHECK(hipStreamSynchronize(g_stream)); HIP_CHECK(hipStreamDestroy(g_stream)); }
    g_stream = s;
    g_own_stream = false;
    if (!g_stream) { HIP_CHECK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking)); g_own_stream = true; }
}

/* ---- host-pointer entry points: zero-copy I/O ---- */
static int g_zero_copy = []() { const char* e = getenv("SAF_HIP_ZERO_COPY"); return e ? atoi(e) != 0 : 1; }();
bool zero_copy_io() { return g_zero_copy != 0; }

/* ---- per-kernel timing ---- */
struct ProfRec { const char* name; hipEvent_t a, b; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;

KernelTimer::KernelTimer(const char* name) : slot(-1)
{
    if (!g_prof) return;
    ProfRec r; r.name = name;
    HIP_CHECK(hipEventCreate(&r.a)); HIP_CHECK(hipEventCreate(&r.b));
    HIP_CHECK(hipEventRecord(r.a, stream()));
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
KernelTimer::~KernelTimer()
{
    if (slot >= 0) HIP_CHECK(hipEventRecord(g_recs[slot].b, stream()));
}

}  // namespace saf

extern "C" {

void saf_hip_profile_enable(int on) { saf::g_prof = on != 0; }
void saf_hip_profile_reset(void)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    for (auto& r : saf::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    saf::g_recs.clear();
}
/* Sums the recorded launches of kernel `name`; returns the launch count. */
int saf_hip_profile_read(const char* name, double* total_ms)
{
    HIP_CHECK(hipStreamSynchronize(saf::stream()));
    int n = 0; double t = 0.0;
    for (auto& r : saf::g_recs)
        if (!strcmp(r.name, name)) { float ms = 0.f; HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b)); t += ms; n++; }
    if (total_ms) *total_ms = t;
    return n;
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
