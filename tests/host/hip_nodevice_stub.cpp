// A HIP runtime with no device, for the host-only sanitizer build of the C-ABI (make asan): the dozen runtime entry
// points the library's host side calls, each answering as a box without a GPU would (hipErrorNoDevice), so that the
// malformed-argument driver is hermetic: nothing parses a code object, nothing touches a driver.  Test infrastructure
// only; the product library links the real libamdhip64.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <string.h>

static thread_local hipError_t g_last = hipSuccess;
static void* g_module = nullptr;

extern "C" {
void** __hipRegisterFatBinary(const void*) { return &g_module; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
// a launch fails where it starts: the configuration push reports the missing device (the kernel's host stub is then skipped,
// as `kernel<<<...>>>(...)` prescribes) and hipGetLastError() hands the error to DXMI_CHECK_LAUNCH
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return g_last = hipErrorNoDevice; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* s, hipStream_t* st) {
    *g = dim3(1); *b = dim3(1); *s = 0; *st = nullptr;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return g_last = hipErrorNoDevice; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "no ROCm-capable device is detected (stub)"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipErrorNoDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipErrorNoDevice; }
hipError_t hipGetDeviceCount(int* n) { *n = 0; return hipErrorNoDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) { memset(p, 0, sizeof *p); return hipErrorNoDevice; }
hipError_t hipGetSymbolAddress(void** p, const void*) { *p = nullptr; return hipErrorNoDevice; }
hipError_t hipMemcpyFromSymbol(void*, const void*, size_t, size_t, hipMemcpyKind) { return hipErrorNoDevice; }
}
