// 3x3 convolution 64 -> 64 channels with REGISTER-RESIDENT weights for gfx950: the first two ResBlockV2s of the value network
// (IGEBMEncoderV2, models/modules.py:39-101,142-158: 64 -> 64 at 32x32 and 16x16) forward and their data gradients (same shape on
// transposed-flipped weights, LeakyReLU mask in the epilogue).  conv_pipe_kernel gave these layers a 128-cout tile (half of every MFMA
// wasted at Cout = 64) and an 18-step K loop per tile that is all prologue and epilogue: 80 us for a layer whose 67 MB of traffic is
// worth 15 us (round-3 train trace: 41 launches, 3.3 ms per step).
//
// Here the whole weight tensor (9 taps x 64 x 64 bf16 = 73.7 KB) lives in the VGPRs of a persistent workgroup for its whole life:
// wave (ch, ph) owns couts 32 ch .. + 31 (36 A fragments of v_mfma_f32_16x16x32_bf16 = 144 registers, taken from the packed
// weights the other conv kernels use) and the 64 pixels 64 ph .. + 63 of every 128-pixel tile.  The only thing that streams is the
// input: a tile's whole halo for all 64 channels (6 x 34 or 10 x 18 pixels x 128 B) travels global -> LDS by DMA in one burst into
// one of two buffers, one tile ahead of the MFMAs, so a tile is 144 MFMAs per wave between two barriers with B fragments read by
// ds_read_b128 at a per-lane base + compile-time offsets (pixel rows are 8 slots of 16 B; slot s of halo pixel hp sits at
// s ^ (hp & 7): sixteen consecutive pixels of one channel piece touch every bank once).  Epilogue in two 64-pixel passes through a
// 17 KB fp32 slab: bias in the accumulator layout, then every thread owns 16-byte output pieces of whole 128-byte NHWC rows:
// + residual, x activation mask, activation, ONE rounding — the order of conv_epilogue_lds.  Two workgroups share a CU (68 KB of
// LDS, 2 x 4 waves), so one's epilogue and halo wait overlap the other's MFMAs.
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define C6_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define C6_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int C6_SLAB_PITCH = 272;               // 64 couts fp32 + 16 B: conflict-free b128 writes (accumulator layout) and reads (rows)
constexpr int C6_SLAB = 64 * C6_SLAB_PITCH;      // one 64-pixel pass

__device__ uint4 c6_zero16 = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void c6_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// TW: tile width (32: 4 x 32-pixel tiles, 16: 8 x 16); the halo is (TH + 2) x (TW + 2) pixels
template <int TW, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_c64_kernel(ConvArgs p) {
    constexpr int TH = 128 / TW, HP = TW + 2, HH = TH + 2, PIX = HP * HH;
    constexpr int BLOCKS = (PIX + 7) / 8;                 // 1-KiB DMA blocks: 8 pixels x 128 B
    constexpr int HALO = BLOCKS * 1024;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const slab = smem + 2 * HALO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, ph = wave >> 1;
    const int txn = p.OW / TW, tyn = p.OH / TH;
    const int ntiles = p.N * tyn * txn;
    const char* const zero_page = reinterpret_cast<const char*>(&c6_zero16);

    auto issue_halo = [&](int tile, char* buf) {
        const int tx = tile % txn, ty = (tile / txn) % tyn, n = tile / (txn * tyn);
        const int oy0 = ty * TH, ox0 = tx * TW;
        for (int b = wave; b < BLOCKS; b += 4) {
            const int hp = b * 8 + (lane >> 3);
            const int hy = hp / HP, hx = hp - hy * HP;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const int s = (lane & 7) ^ (hp & 7);
            const bool ok = hp < PIX && iy >= 0 && ix >= 0 && iy < p.IH && ix < p.IW;
            const void* g = ok ? (const void*)(p.in0 + (((size_t)n * p.IH + iy) * p.IW + ix) * 64 + s * 8) : (const void*)zero_page;
            __builtin_amdgcn_global_load_lds(C6_GPTR(g), C6_LPTR(buf + b * 1024), 16, 0, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue_halo(tile, smem);

    // ---- resident weights: A of (tap t, chunk c, 16-cout sub-block sb) of this wave's 32 couts: lane (co = lane & 15, kg = lane >> 4)
    // holds lanes co + 16 sb + 32 (kg & 1) of the packed 32x32x16 fragment (k-step 2 c + (kg >> 1), 32-cout block ch)
    bf16x8 A[36];
    {
        const int kg = lane >> 4;
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.w) + (lane & 15) + 32 * (kg & 1);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
                    A[(t * 2 + c) * 2 + sb] = wf[(size_t)((t * p.KST + c * 2 + (kg >> 1)) * p.CB + ch) * 64 + 16 * sb];
    }
    const int px = lane & 15, kg = lane >> 4;
    const float slope = dxmi_act_slope(p.act);
    f32x4 bv[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        bv[sb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv[sb] = *reinterpret_cast<const f32x4*>(p.bias + ch * 32 + sb * 16 + 4 * kg);
    }

    int cur = 0;
    for (;;) {
        const int next = tile + gridDim.x;
        // vmcnt retires in order: everything older than this wave's four youngest operations — the previous tile's four row stores —
        // has landed, i.e. its halo blocks of `tile` (vmcnt(0) would also wait for those stores to be acknowledged: ~2 us per tile)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        c6_barrier();                                          // ... and every other wave's; the other buffer and the slab are free
        if (next < ntiles) issue_halo(next, smem + (cur ^ 1) * HALO);
        const char* const halo = smem + cur * HALO;
        // this tile's residual / mask pieces are requested now and consumed in the epilogue
        const int tx = tile % txn, ty = (tile / txn) % tyn, n = tile / (txn * tyn);
        size_t ooff[4];
        bf16x8 rv[4], mv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pix = (q >> 1) * 64 + (tid >> 3) + 32 * (q & 1);
            const int y = pix / TW, x = pix % TW;
            ooff[q] = ((((size_t)n * p.OH + ty * TH + y) * p.OW) + tx * TW + x) * 64 + (tid & 7) * 8;
            if (p.residual) rv[q] = *reinterpret_cast<const bf16x8*>(p.residual + ooff[q]);
            if (p.mask_src) mv[q] = *reinterpret_cast<const bf16x8*>(p.mask_src + ooff[q]);
        }

        // ---- 64 pixels x 32 couts per wave: 4 pixel blocks x 2 cout sub-blocks x 9 taps x 2 chunks
        f32x4 acc[2][4];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[sb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // operand reads run one tap ahead of the MFMAs that consume them (read, wait, MFMA per fragment left the MFMA pipe idle for an
        // LDS latency 36 times per tile: 12 k cycles per tile instead of 2.3 k)
        bf16x8 B[2][4][2];
        auto read_tap = [&](int t, bf16x8 (&Bt)[4][2]) {
            const int ky = t / 3, kx = t % 3;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int blk = ph * 4 + nb;                                  // 16-pixel block of the tile
                const int y = (blk * 16) / TW, x0 = (blk * 16) % TW;
                const int hp = (y + ky) * HP + x0 + px + kx;
                const char* row = halo + hp * 128;
                const int sw = hp & 7;
#pragma unroll
                for (int c = 0; c < 2; ++c) Bt[nb][c] = *reinterpret_cast<const bf16x8*>(row + (((c * 4 + kg) ^ sw) << 4));
            }
        };
        read_tap(0, B[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) read_tap(t + 1, B[(t + 1) & 1]);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[sb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[(t * 2 + c) * 2 + sb], B[t & 1][nb][c], acc[sb][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue, two passes of 64 pixels: D[co][pixel]: lane = pixel px of its block, couts 4 kg .. 4 kg + 3 of sub-block sb
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) c6_barrier();                           // pass 0's rows have been read
            if (ph == pass) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        f32x4 v = acc[sb][nb];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[sb][e];
                        *reinterpret_cast<f32x4*>(slab + (nb * 16 + px) * C6_SLAB_PITCH + (ch * 32 + sb * 16 + 4 * kg) * 4) = v;
                    }
            }
            c6_barrier();
            // thread -> 16-byte output pieces: pixel lp = (tid >> 3) + 32 k of the pass, couts 8 (tid & 7) .. + 7
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int lp = (tid >> 3) + 32 * k, pc = tid & 7, q = pass * 2 + k;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + lp * C6_SLAB_PITCH + pc * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + lp * C6_SLAB_PITCH + pc * 32 + 16);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                if (p.residual) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[q][e];
                }
                if (p.mask_src) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= ((float)mv[q][e] > 0.f ? 1.f : p.mask_slope);
                }
                bf16x8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (bf16)dxmi_act_lin(v[e], slope);
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + ooff[q]) = ov;
            }
        }
        if (next >= ntiles) break;
        tile = next;
        cur ^= 1;
    }
}

}  // namespace

// Launches the 64 -> 64 kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv_c64_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV_C64") ? atoi(getenv("DXMI_CONV_C64")) : 1;   // 0: conv_pipe_kernel as before
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups != 0 || a.addvec || a.act == DXMI_ACT_SILU || a.gn_stats || a.gn_out) return 1;
    if (a.C0 != 64 || a.C1 != 0 || a.Cout != 64 || a.IH != a.OH || a.IW != a.OW) return 1;
    const int TW = a.OW % 32 == 0 ? 32 : (a.OW % 16 == 0 ? 16 : 0);
    if (TW == 0 || a.OH % (128 / TW) != 0) return 1;
    if ((long)a.N * a.IH * a.IW * 64 * 2 >= (1L << 31)) return 1;
    if (kernel_id) {
        *kernel_id = 460000 + TW;        // conv_c64_kernel<TW>
        return DXMI_OK;
    }
    const int ntiles = a.N * (a.OH / (128 / TW)) * (a.OW / TW);
    static const int occ_g = getenv("DXMI_CONV_C64_OCC") ? atoi(getenv("DXMI_CONV_C64_OCC")) : 1;
    const int maxg = occ_g == 1 ? 256 : 512;
    const int grid = ntiles < maxg ? ntiles : maxg;           // persistent workgroups: two (one) per CU
    const int HPIX = (TW + 2) * (128 / TW + 2);
    const size_t lds = (size_t)2 * ((HPIX + 7) / 8) * 1024 + C6_SLAB;
    static const int occ = getenv("DXMI_CONV_C64_OCC") ? atoi(getenv("DXMI_CONV_C64_OCC")) : 1;       // tuning: workgroups per CU the kernel is compiled for (2 spills the weights)
#define C6_LAUNCH(TW_, OCC_)                                                                                                              \
    do {                                                                                                                                  \
        static bool attr_set = false;                                                                                                     \
        if (!attr_set) {                                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c64_kernel<TW_, OCC_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                              \
        }                                                                                                                                 \
        hipLaunchKernelGGL((conv_c64_kernel<TW_, OCC_>), dim3(grid), dim3(256), lds, st, a);                                                \
    } while (0)
    if (TW == 32) { if (occ == 1) C6_LAUNCH(32, 1); else C6_LAUNCH(32, 2); }
    else { if (occ == 1) C6_LAUNCH(16, 1); else C6_LAUNCH(16, 2); }
#undef C6_LAUNCH
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(c64)");
    return DXMI_OK;
}
