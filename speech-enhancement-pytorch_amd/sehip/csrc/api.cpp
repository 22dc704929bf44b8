// libsehip C ABI plumbing: error string, version, device probe.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"
#include "det.h"

static thread_local char g_err[512] = "";

int sehip_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static thread_local char g_kernel[128] = "";

void sehip_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

static int g_deterministic = 0;
int sehip_deterministic(void) { return g_deterministic; }
extern "C" int sehip_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return 0; }
extern "C" int sehip_get_deterministic(void) { return g_deterministic; }

// ---- the deterministic schedule's per-stream partial arrays (csrc/det.h).  Launches of one stream are ordered, so one array per
// stream serves them all; an array that has to grow waits for its stream first.  Not inside a stream capture (Solver refuses
// cudnn_deterministic + use_graph for the same reason).
namespace {
struct DetSlot { hipStream_t st; double* p; size_t doubles; };
DetSlot det_pool[32];
}  // namespace

DetCtx sehip_det_ctx(hipStream_t st, size_t ndoubles, bool* ok) {
    *ok = true;
    if (!g_deterministic) return DetCtx{nullptr};
    *ok = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        sehip_set_error(-2, "deterministic schedule: its partial arrays cannot be set up inside a stream capture");
        return DetCtx{nullptr};
    }
    DetSlot* e = nullptr;
    for (auto& q : det_pool)
        if (q.p && q.st == st) { e = &q; break; }
    if (!e)
        for (auto& q : det_pool)
            if (!q.p) { e = &q; break; }
    if (!e) {       // every slot belongs to some stream: give them all back once the device is idle
        if (hipDeviceSynchronize() != hipSuccess) { sehip_set_error(-2, "deterministic schedule: device synchronize failed"); return DetCtx{nullptr}; }
        for (auto& q : det_pool) { (void)hipFree(q.p); q.p = nullptr; q.doubles = 0; q.st = nullptr; }
        e = &det_pool[0];
    }
    if (!e->p || e->doubles < ndoubles) {
        if (e->p) {
            if (hipStreamSynchronize(st) != hipSuccess) { sehip_set_error(-2, "deterministic schedule: stream synchronize failed"); return DetCtx{nullptr}; }
            (void)hipFree(e->p);
            e->p = nullptr;
        }
        const size_t want = ndoubles < (1u << 16) ? (1u << 16) : ndoubles * 2;
        double* p = nullptr;
        if (hipMalloc(&p, want * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            sehip_set_error(-2, "deterministic schedule: could not allocate %zu partial sums", want);
            return DetCtx{nullptr};
        }
        e->st = st; e->p = p; e->doubles = want;
    }
    *ok = true;
    return DetCtx{e->p};
}

extern "C" const char* sehip_last_error(void) { return g_err; }
extern "C" const char* sehip_last_kernel(void) { return g_kernel; }
extern "C" int sehip_version(void) { return 100; }

// Returns 0 when device `dev` is a gfx950 part this library was built for.
extern "C" int sehip_check_device(int dev) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return sehip_set_error(-3, "check_device: %s", hipGetErrorString(e));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return sehip_set_error(-3, "check_device: libsehip is built for gfx950 only, device %d is %s", dev, p.gcnArchName);
    return 0;
}

// ---- cheap cross-stream dependency (see include/sehip.h)
extern "C" void* sehip_event_create(void) {
    hipEvent_t e = nullptr;
    hipError_t st = hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
    if (st != hipSuccess) {
        sehip_set_error(-2, "event_create: %s", hipGetErrorString(st));
        return nullptr;
    }
    return (void*)e;
}

extern "C" int sehip_event_destroy(void* event) {
    if (!event) return 0;
    hipError_t st = hipEventDestroy((hipEvent_t)event);
    if (st != hipSuccess) return sehip_set_error(-2, "event_destroy: %s", hipGetErrorString(st));
    return 0;
}

extern "C" int sehip_stream_depend(void* to_stream, void* from_stream, void* event) {
    if (!event) return sehip_set_error(-1, "stream_depend: null event");
    hipError_t st = hipEventRecord((hipEvent_t)event, (hipStream_t)from_stream);
    if (st != hipSuccess) return sehip_set_error(-2, "stream_depend: record: %s", hipGetErrorString(st));
    st = hipStreamWaitEvent((hipStream_t)to_stream, (hipEvent_t)event, 0);
    if (st != hipSuccess) return sehip_set_error(-2, "stream_depend: wait: %s", hipGetErrorString(st));
    return 0;
}

// A stream of a given priority class: -1 = the device's highest, 0 = default, 1 = the device's lowest.  The weight-gradient side
// stream is filler work beside the dependent chain; with the lowest priority the dispatcher hands a freed CU to the chain first.
extern "C" void* sehip_stream_create(int priority_class) {
    int least = 0, greatest = 0;
    hipError_t st = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (st != hipSuccess) {
        sehip_set_error(-2, "stream_create: priority range: %s", hipGetErrorString(st));
        return nullptr;
    }
    const int prio = priority_class < 0 ? greatest : (priority_class > 0 ? least : 0);
    hipStream_t s = nullptr;
    st = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
    if (st != hipSuccess) {
        sehip_set_error(-2, "stream_create: %s", hipGetErrorString(st));
        return nullptr;
    }
    return (void*)s;
}

extern "C" int sehip_stream_destroy(void* stream) {
    if (!stream) return 0;
    hipError_t st = hipStreamDestroy((hipStream_t)stream);
    if (st != hipSuccess) return sehip_set_error(-2, "stream_destroy: %s", hipGetErrorString(st));
    return 0;
}

// the two halves of sehip_stream_depend, for a dependency that is recorded now and waited for later
extern "C" int sehip_event_record(void* event, void* stream) {
    if (!event) return sehip_set_error(-1, "event_record: null event");
    hipError_t st = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    if (st != hipSuccess) return sehip_set_error(-2, "event_record: %s", hipGetErrorString(st));
    return 0;
}

extern "C" int sehip_stream_wait_event(void* stream, void* event) {
    if (!event) return sehip_set_error(-1, "stream_wait_event: null event");
    hipError_t st = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    if (st != hipSuccess) return sehip_set_error(-2, "stream_wait_event: %s", hipGetErrorString(st));
    return 0;
}
