// Library-level plumbing of libmsnet_hip.so: error strings and the per-launch HIP-event profiler.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace msnet {

static thread_local char g_err[512] = "";
static thread_local unsigned* g_oflag = nullptr;

static thread_local bool g_exact_tails = false;

unsigned* overflow_flag() { return g_oflag; }
bool exact_tails() { return g_exact_tails; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

struct Rec {
    const char* name;
    hipEvent_t a, b;
    double flops, bytes;
};
static std::mutex g_mu;
static std::vector<Rec*> g_recs;
static volatile int g_prof = 0;
static char g_prof_prefix[256] = "";     // comma-separated name prefixes; only matching launches are timed ("" = all)

static bool prof_selected(const char* n) {
    if (!g_prof_prefix[0]) return true;
    for (const char* p = g_prof_prefix; *p;) {
        const char* e = strchr(p, ',');
        const size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len && strncmp(n, p, len) == 0) return true;
        if (!e) break;
        p = e + 1;
    }
    return false;
}

LaunchScope::LaunchScope(const char* n, hipStream_t s, double flops, double bytes, bool attach_)
    : name(n), stream(s), rec(nullptr), attach(attach_) {
    if (!g_prof) return;
    if (!prof_selected(n)) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;      // a launch being captured into a graph is not timed (its events
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;      // would belong to the capture)
    Rec* r = new Rec{n, nullptr, nullptr, flops, bytes};
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return; }
    if (!attach) (void)hipEventRecord(r->a, s);
    rec = r;
}

bool LaunchScope::events(hipEvent_t* start, hipEvent_t* stop) const {
    if (!rec || !attach) return false;
    Rec* r = static_cast<Rec*>(rec);
    *start = r->a; *stop = r->b;
    launched = true;
    return true;
}

LaunchScope::~LaunchScope() {
    if (!rec) return;
    Rec* r = static_cast<Rec*>(rec);
    if (attach && !launched) {
        // an early return between the scope and its one launch: the pair was never recorded -- querying it later would leave a
        // sticky HIP error for the next check_launch() to report as a failure that never happened
        (void)hipEventDestroy(r->a);
        (void)hipEventDestroy(r->b);
        delete r;
        return;
    }
    if (!attach) (void)hipEventRecord(r->b, stream);
    std::lock_guard<std::mutex> lk(g_mu);
    g_recs.push_back(r);
}

}  // namespace msnet

using namespace msnet;


extern "C" int msnet_version(void) { return 1; }
extern "C" const char* msnet_last_error(void) { return g_err; }

extern "C" int msnet_set_exact_tails(int on) {
    g_exact_tails = on != 0;
    return 0;
}

extern "C" int msnet_set_overflow_flag(void* device_u32) {
    g_oflag = static_cast<unsigned*>(device_u32);
    return 0;
}

extern "C" int msnet_prof_enable(int on) {
    g_prof = on ? 1 : 0;
    return 0;
}

extern "C" int msnet_prof_select(const char* name_prefix) {
    if (name_prefix && strlen(name_prefix) >= sizeof(g_prof_prefix)) return fail("msnet_prof_select: prefix too long");
    strncpy(g_prof_prefix, name_prefix ? name_prefix : "", sizeof(g_prof_prefix) - 1);
    g_prof_prefix[sizeof(g_prof_prefix) - 1] = 0;
    return 0;
}

extern "C" long msnet_prof_collect(char* buf, size_t n) {
    std::vector<Rec*> recs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        recs.swap(g_recs);
    }
    struct Agg { long calls = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (Rec* r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r->b) == hipSuccess && hipEventElapsedTime(&ms, r->a, r->b) == hipSuccess) {
            Agg& g = agg[r->name];
            g.calls++; g.ms += ms; g.flops += r->flops; g.bytes += r->bytes;
        } else {
            (void)hipGetLastError();          // a row that cannot be read is dropped, not left behind as a sticky error
        }
        (void)hipEventDestroy(r->a);
        (void)hipEventDestroy(r->b);
        delete r;
    }
    size_t off = 0;
    for (auto& kv : agg) {
        char line[256];
        int k = snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.calls, kv.second.ms,
                         kv.second.flops, kv.second.bytes);
        if (k < 0) continue;
        if (!buf || off + (size_t)k + 1 > n) { fail("msnet_prof_collect: buffer too small"); return -1; }
        memcpy(buf + off, line, (size_t)k);
        off += (size_t)k;
    }
    if (buf && off < n) buf[off] = 0;
    return (long)off;
}
