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

extern "C" int vm_abi_version(void) { return 7; }

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
