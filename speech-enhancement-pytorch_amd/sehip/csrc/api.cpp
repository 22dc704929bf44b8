// libsehip C ABI plumbing: error string, version, device probe.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

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
