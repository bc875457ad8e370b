// oc_internal.h -- helpers shared by the translation units of liboc_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

int oc_internal_fail(int code, const char *fmt, ...);      // overcooked_engine.hip: sets oc_last_error()
struct oc_group;
struct oc_engine;
oc_engine *oc_internal_group_engine(oc_group *g, int i);    // overcooked_engine.hip: engine i of a group (NULL if out of range)
int oc_internal_set_cu_budget(oc_engine *e, int cus);        // size the engine's launches for a CU-masked stream (0 = whole device)

// Every launching entry point runs with the GPU that owns its buffers current, whatever the caller's (or torch's)
// current device is, and puts the caller's device back on return: kernels are enqueued on streams of that GPU, and
// CU counts / LDS opt-ins are taken from it.  Costs one hipGetDevice when the device is already current.
struct OcDeviceScope {
    int prev = -1;
    bool switched = false;
    bool ok = true;
    explicit OcDeviceScope(int dev) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            switched = ok;
        }
    }
    ~OcDeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    OcDeviceScope(const OcDeviceScope &) = delete;
    OcDeviceScope &operator=(const OcDeviceScope &) = delete;
};

// GPU that owns a device allocation (-1: not a device pointer / unknown): one runtime query, well under a
// microsecond -- used by the entry points that have no engine handle to take the device from.
inline int oc_device_of_pointer(const void *p) {
    if (!p) return -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? attr.device : -1;
}
