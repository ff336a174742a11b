// runtime.hip -- library management entry points of libohevc_hip.so (error text, device selection).
#include <stdarg.h>
#include <string.h>
#include "common.hpp"

namespace ohevc {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ohevc

extern "C" const char *ohevc_last_error(void) { return ohevc::g_err; }

extern "C" int ohevc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ohevc_set_device(int device)
{
    using namespace ohevc;
    OHEVC_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    OHEVC_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return OHEVC_ERR_NODEV;
    }
    return OHEVC_OK;
}

extern "C" const char *ohevc_version(void) { return "ohevc_hip 0.1 (gfx950)"; }
