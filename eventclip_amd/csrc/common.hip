// Error reporting, version and device info for libeventclip_hip.so.
#include "common.h"

#include <cstring>

namespace ec {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ec

extern "C" {

EC_API const char *ec_last_error(void) { return ec::err_buf(); }

EC_API int ec_version(void) { return 100; }

EC_API int ec_device_info(int *cu_count, char *name, int name_len)
{
    int dev = 0;
    EC_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    EC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return EC_OK;
}

}  // extern "C"
