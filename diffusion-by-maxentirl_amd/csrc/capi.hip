// C-ABI glue: error text, version, device probe.  No torch types, no exceptions.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void dxmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dxmi_last_error(void) { return g_err; }

extern "C" int dxmi_version(void) { return 100; }

extern "C" int dxmi_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        dxmi_set_error("dxmi_device_check: no HIP device visible");
        return DXMI_ENODEV;
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        dxmi_set_error("dxmi_device_check: hipGetDeviceProperties failed");
        return DXMI_ENODEV;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        dxmi_set_error("dxmi_device_check: device arch %s is not gfx950", prop.gcnArchName);
        return DXMI_ENODEV;
    }
    return DXMI_OK;
}
