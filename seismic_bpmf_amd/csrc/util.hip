// Error reporting and device diagnostics of libbpmf_hip.so.
#include "common.h"
#include "../../include/bpmf_hip.h"

#include <cstring>

namespace bpmf {

char* last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
}

}  // namespace bpmf

extern "C" const char* bpmf_last_error(void) { return bpmf::last_error_buf(); }

// HIP analogue of the reference's device probe (BPMF/GPU.cu:8-24).
extern "C" int bpmf_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        bpmf::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return -2;
    }
    return n;
}

extern "C" int bpmf_device_info(int device, char* name, size_t name_len, size_t* total_mem_bytes,
                                int* compute_units)
{
    hipDeviceProp_t prop;
    BPMF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (total_mem_bytes) *total_mem_bytes = prop.totalGlobalMem;
    if (compute_units) *compute_units = prop.multiProcessorCount;
    return 0;
}
