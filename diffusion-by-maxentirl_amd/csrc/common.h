// Shared device/host helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dxmi_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

void dxmi_set_error(const char* fmt, ...);
int dxmi_device_cus();                  // capi.hip: compute units of the current device (0: none)
int dxmi_tuning(const char* name);      // capi.hip: value of a kernel-selection knob (dxmi_set_tuning)

#define DXMI_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            dxmi_set_error(__VA_ARGS__);          \
            return DXMI_EINVAL;                   \
        }                                         \
    } while (0)

#define DXMI_CHECK_LAUNCH(name)                                                     \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            dxmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return DXMI_ELAUNCH;                                                    \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float dxmi_act(float v, int act) {
    switch (act) {
        case DXMI_ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        case DXMI_ACT_RELU:    return v > 0.f ? v : 0.f;
        case DXMI_ACT_SILU:    return v / (1.f + __expf(-v));
        default:               return v;
    }
}

// x * sigmoid(x) with the hardware reciprocal: v_exp_f32 + v_rcp_f32 + 2 VALU per value.  `v / (1 + expf(-v))` compiles
// to the IEEE division sequence (div_scale / rcp / 4 fma / div_fmas / div_fixup: ~10 more instructions), which made the
// GroupNorm+SiLU kernel VALU-bound: ~17 us of arithmetic per 24 us of HBM time at 128 ch x 32x32 x 256 images, none of
// it overlapped (one round of workgroups).  v_rcp_f32 is accurate to 1 ulp; the result is rounded to bf16 (2^-9).
__device__ __forceinline__ float dxmi_silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// Branch-free form for the epilogues of the conv kernels: `act` is uniform, but a switch per element costs a ladder of
// scalar branches per value (measured: the tile epilogue of conv_pipe_kernel spent thousands of cycles in them).
// slope = 1 (none), 0.2 (leaky), 0 (relu): v > 0 ? v : slope * v.  SiLU keeps its own (hoisted) path.
// One v_mul + v_cmp + v_cndmask; a NaN stays a NaN (NaN > 0 is false and slope * NaN = NaN) — the fmaxf/fminf form
// used before returned 0 for NaN inputs and hid diverged activations from the non-finite checks downstream.
// (relu of -inf gives NaN rather than 0: a non-finite activation stays non-finite.)
__device__ __forceinline__ float dxmi_act_slope(int act) { return act == DXMI_ACT_LEAKY02 ? 0.2f : (act == DXMI_ACT_RELU ? 0.f : 1.f); }
__device__ __forceinline__ float dxmi_act_lin(float v, float slope) { return v > 0.f ? v : slope * v; }

// Workgroup barrier for LDS hand-offs that does NOT drain the vector-memory queue: __syncthreads()
// makes hipcc emit s_waitcnt vmcnt(0) first, which stalls on every prefetched global load and on every
// outstanding global store (CDNA4's vmcnt counts stores too).  LDS traffic of this wave is complete
// (lgkmcnt(0)) before the barrier; global memory ordering is NOT implied.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
