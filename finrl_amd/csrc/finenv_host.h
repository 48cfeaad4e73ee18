// finenv_host.h -- host-side helpers shared by the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>

namespace finenv_host {

// HIP device that owns a device pointer, or -1 (host pointer, no device, unknown).
inline int pointer_device(const void *p)
{
    hipPointerAttribute_t a;
    if (p != nullptr && hipPointerGetAttributes(&a, p) == hipSuccess &&
        a.type == hipMemoryTypeDevice)
        return a.device;
    (void)hipGetLastError();
    return -1;
}

// Launches go to the device that owns the handle's state block, whatever the calling thread's
// current device is (a caller holding tensors on cuda:3 without a set_device would otherwise
// launch on device 0 with device-3 pointers).  Restores the previous device on scope exit.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int want)
    {
        int cur = -1;
        if (want >= 0 && hipGetDevice(&cur) == hipSuccess && cur != want &&
            hipSetDevice(want) == hipSuccess)
            prev = cur;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

}  // namespace finenv_host
