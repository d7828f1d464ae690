// Narrow-head 3x3 convolution for gfx950: 128 channels -> a few output channels (conv_out of the U-Nets: 128 -> 3,
// models/DxMI/unet_small.py:144,329-331; models/cm/unet.py `out`), NHWC bf16 in, NCHW fp32 out (the image the sampler
// consumes).  The layer is a pure read of the activation (67 MB at 256 x 32x32) with trivial arithmetic; the generic MFMA
// kernel it replaces pads Cout to 32, stages 32 channels at a time through registers and took 66 us.
//
// One workgroup = one 4x32-pixel tile of one image.  The tile's WHOLE input halo (6 x 34 pixels x 128 channels, 51 KB: two
// workgroups share a CU, so one's load burst overlaps the other's MFMAs; an 8-row tile with one workgroup per CU measured 36.6 us)
// travels global -> LDS by DMA in one burst (no chunk loop, no barriers inside the K loop); all 36 weight fragments
// (9 taps x 4 chunks of 32 channels, couts padded to 16) are loaded once into registers from the packed weights the other conv
// kernels use (the 16x16x32 A fragment is a lane subset of the 32x32x16 fragments).  Each wave then runs 72
// v_mfma_f32_16x16x32_bf16 over its 32 pixels (one tile row) with B fragments read by ds_read_b128 at a per-lane base plus
// a compile-time offset.  Pixel rows are 256 B = 16 slots of 16 B; channel piece s of halo pixel hp sits in slot s ^ (hp & 15).
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define HD_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HD_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int HD_TH = 4;                                          // tile rows (one per wave): 51 KB of LDS, two workgroups per CU (register-limited)
constexpr int HD_HP = 34, HD_HH = HD_TH + 2, HD_PIX = HD_HP * HD_HH;     // halo pitch / rows / pixels
constexpr int HD_BLOCKS = (HD_PIX + 3) / 4;                      // 1-KiB DMA blocks (4 pixels each)

__device__ uint4 hd_zero16 = {0u, 0u, 0u, 0u};

__global__ __launch_bounds__(256, 2) void conv_head_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int txn = p.OW / 32, tyn = p.OH / HD_TH;
    const int tx = blockIdx.x % txn, ty = (blockIdx.x / txn) % tyn, n = blockIdx.x / (txn * tyn);
    const int oy0 = ty * HD_TH, ox0 = tx * 32;

    // ---- halo: block b = pixels 4b .. 4b+3 (row-major over the 6 x 34 halo), 256 B each
    for (int b = wave; b < HD_BLOCKS; b += 4) {
        const int hp = b * 4 + (lane >> 4);
        const int hy = hp / HD_HP, hx = hp - hy * HD_HP;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const int s = (lane & 15) ^ (hp & 15);
        const bool ok = hp < HD_PIX && iy >= 0 && ix >= 0 && iy < p.IH && ix < p.IW;
        const void* g = ok ? (const void*)(p.in0 + (((size_t)n * p.IH + iy) * p.IW + ix) * 128 + s * 8) : (const void*)&hd_zero16;
        __builtin_amdgcn_global_load_lds(HD_GPTR(g), HD_LPTR(smem + b * 1024), 16, 0, 0);
    }
    // ---- weight fragments: A of (tap t, chunk c) for couts 0..15: lane (co = lane & 15, channel group kg = lane >> 4) holds
    // lanes co + 32 (kg & 1) of the packed 32x32x16 fragment of k-step 2c + (kg >> 1), cout block 0
    bf16x8 A[36];
    {
        const int kg = lane >> 4;
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.w) + (lane & 15) + 32 * (kg & 1);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) A[t * 4 + c] = wf[(size_t)((t * p.KST + c * 2 + (kg >> 1)) * p.CB) * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- 32 pixels per wave: tile row w, two 16-pixel blocks
    constexpr int NBW = HD_TH / 2;       // 16-pixel blocks per wave
    const int px = lane & 15, kg = lane >> 4;
    f32x4 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nb][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t % 3;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int hp = (wave * (HD_TH / 4) + (nb >> 1) + ky) * HD_HP + (nb & 1) * 16 + px + kx;
            const char* row = smem + hp * 256;
            const int sw = hp & 15;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(row + (((c * 4 + kg) ^ sw) << 4));
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t * 4 + c], b, acc[nb], 0, 0, 0);
            }
        }
    }
    // ---- D[co][pixel]: lane = pixel (lane & 15), couts 4 kg .. 4 kg + 3 -> NCHW fp32, 64-byte runs per cout
    const float slope = dxmi_act_slope(p.act);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int oy = oy0 + wave * (HD_TH / 4) + (nb >> 1), ox = ox0 + (nb & 1) * 16 + px;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = 4 * kg + e;
            if (co < p.Cout) {
                float v = acc[nb][e] + (p.bias ? p.bias[co] : 0.f);
                v = dxmi_act_lin(v, slope);
                reinterpret_cast<float*>(p.out)[(((size_t)n * p.Cout + co) * p.OH + oy) * p.OW + ox] = v;
            }
        }
    }
}

}  // namespace

// Launches the narrow-head kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv_head_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV_HEAD") ? atoi(getenv("DXMI_CONV_HEAD")) : 1;   // 0: generic kernel
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NCHW_F32) return 1;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups != 0 || a.mask_src || a.addvec || a.residual || a.act == DXMI_ACT_SILU) return 1;
    if (a.C0 != 128 || a.C1 != 0 || a.Cout > 16 || a.OW % 32 != 0 || a.OH % HD_TH != 0) return 1;
    if (kernel_id) {
        *kernel_id = 600000;
        return DXMI_OK;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = a.N * (a.OH / HD_TH) * (a.OW / 32);
    hipLaunchKernelGGL(conv_head_kernel, dim3(grid), dim3(256), (size_t)HD_BLOCKS * 1024, st, a);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(head)");
    return DXMI_OK;
}
