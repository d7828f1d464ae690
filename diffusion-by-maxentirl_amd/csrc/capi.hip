// C-ABI glue: error text, version, device probe.  No torch types, no exceptions.
#include "common.h"
#include <stdarg.h>
#include <stdlib.h>
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

// Kernel-selection knobs (dxmi_set_tuning / dxmi_get_tuning): process-wide, first read from the environment.
namespace {
struct Knob {
    const char* name;
    const char* env;
    int value;
    bool init;
};
Knob g_knobs[] = {
    // Defaults keep an image's result independent of the batch it rides in (one kernel per layer shape whatever N is); the training
    // entry points switch to the throughput values 96 / 13 (dxmi_hip.ops.tune_for_throughput): small grids on smaller tiles.
    {"conv_ws_min_tiles", "DXMI_CONV_WS_MIN_TILES", 0, false},    // conv_ws.hip: fewer (256-pixel, 128-cout) tiles -> conv_pipe_kernel
    {"gn_bwd_fused", "DXMI_GN_BWD_FUSED", 1, false},              // groupnorm.hip: generic GroupNorm backward as one launch on <= 256-pixel maps (2: everywhere it fits, 0: reduce + apply launches)
    {"conv_sm_mask", "DXMI_CONV_SM", 9, false},                   // conv_sm.hip: bit 0 4x4 maps, bit 1 every 8x8 map, bit 2 8x8 maps on under-filled grids, bit 3 8x8 maps with >= 1024 channels
};
Knob* find_knob(const char* name) {
    if (!name) return nullptr;
    for (Knob& k : g_knobs)
        if (strcmp(k.name, name) == 0) {
            if (!k.init) {
                const char* e = getenv(k.env);
                if (e) k.value = atoi(e);
                k.init = true;
            }
            return &k;
        }
    return nullptr;
}
}  // namespace

int dxmi_tuning(const char* name) {       // internal: the launch paths read their knob through this
    Knob* k = find_knob(name);
    return k ? k->value : 0;
}

extern "C" int dxmi_set_tuning(const char* name, int32_t value) {
    Knob* k = find_knob(name);
    DXMI_CHECK_ARG(k != nullptr, "dxmi_set_tuning: unknown knob '%s'", name ? name : "(null)");
    DXMI_CHECK_ARG(value >= 0, "dxmi_set_tuning: %s = %d must be >= 0", name, value);
    k->value = value;
    return DXMI_OK;
}

extern "C" int dxmi_get_tuning(const char* name, int32_t* value) {
    Knob* k = find_knob(name);
    DXMI_CHECK_ARG(k != nullptr && value != nullptr, "dxmi_get_tuning: unknown knob '%s' / null result pointer", name ? name : "(null)");
    *value = k->value;
    return DXMI_OK;
}

// Compute units of the current device (cached per device; 0 when there is none): residency bound of the kernels that hand data
// over between workgroups inside a launch (groupnorm.hip).
int dxmi_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus[dev] = prop.multiProcessorCount;
    }
    return cus[dev];
}

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
