// Error plumbing, device check and small utilities of libvoicemap_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace vm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    return VM_OK;
}

}  // namespace vm

extern "C" const char* vm_last_error(void) { return vm::g_err; }

extern "C" int vm_abi_version(void) { return 11; }

extern "C" int vm_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        vm::set_error("hipGetDevice: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        vm::set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vm::set_error("device %d is %s; libvoicemap_hip.so is built for gfx950 only", dev, prop.gcnArchName);
        return VM_ERR_UNSUPPORTED;
    }
    return VM_OK;
}

extern "C" int vm_fill_zero(void* ptr, int64_t bytes, void* stream) {
    VM_REQUIRE(ptr && bytes >= 0, "vm_fill_zero: bad argument");
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        vm::set_error("vm_fill_zero: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    return VM_OK;
}

// ---- stream-ordering primitives for a caller that replays a recorded launch sequence (voicemap_amd/engine.py: the training step's
// C-ABI calls, event records and waits are recorded once per configuration and replayed without the host logic that produced them).
// Events are created without timing: ordering only. ----
extern "C" int vm_event_create(void** event_out) {
    VM_REQUIRE(event_out, "vm_event_create: null pointer");
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        vm::set_error("vm_event_create: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    *event_out = (void*)ev;
    return VM_OK;
}

extern "C" int vm_event_destroy(void* event) {
    if (event == nullptr) return VM_OK;
    hipError_t e = hipEventDestroy((hipEvent_t)event);
    if (e != hipSuccess) {
        vm::set_error("vm_event_destroy: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    return VM_OK;
}

extern "C" int vm_event_record(void* event, void* stream) {
    VM_REQUIRE(event, "vm_event_record: null event");
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    if (e != hipSuccess) {
        vm::set_error("vm_event_record: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    return VM_OK;
}

extern "C" int vm_stream_wait_event(void* stream, void* event) {
    VM_REQUIRE(event, "vm_stream_wait_event: null event");
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
    if (e != hipSuccess) {
        vm::set_error("vm_stream_wait_event: %s", hipGetErrorString(e));
        return VM_ERR_LAUNCH;
    }
    return VM_OK;
}
