// 1x1 convolution with REGISTER-RESIDENT weights for gfx950 (q|k|v, proj_out, nin_shortcut of the U-Net:
// models/DxMI/unet_small.py:110-187) — the HBM-bound GEMMs [pixels x K] . [K x Cout], K <= 512.
//
// conv1x1_stream_kernel gives every (64-pixel, 128-cout) tile its own workgroup, which re-fetches that tile's
// 128 x K weight slab from L2 (64 KB for 32 KB of input at K = 256: two thirds of the bytes a CU pulls are weights) and
// pays the address set-up per tile (PMC: 16 VALU instructions per MFMA).  Here a workgroup is persistent and keeps ONE
// cout tile for its whole life:
//   * wave w owns couts 32w..32w+31: its K/16 weight fragments are loaded once and stay in VGPRs (64 registers at
//     K = 256) — the only thing that streams is the input;
//   * the input stream (64-pixel x 128-channel chunks, 16 KB, two full 128-byte lines per pixel) travels global -> LDS by
//     DMA (global_load_lds) through a ring of R chunks, R-1 chunks ahead of the MFMAs and across tile boundaries, so the
//     next tile's input is in flight during this tile's epilogue; ONE barrier per chunk (16 MFMAs per wave) is the only
//     workgroup synchronisation; two workgroups share a CU (one for K > 384);
//   * the epilogue is wave-private: a wave's 64 px x 32 cout slice goes through its own 4 KB of LDS (accumulator layout ->
//     pixel rows), the residual slice rides the same wave's DMA queue into that LDS, and the wave stores 64-byte row
//     segments (its neighbour wave writes the other half of each 128-byte line at about the same time).
// vmcnt retires in order: every wait below counts exactly the DMAs / stores that are younger than the chunk it needs.
// The CT cout tiles of one pixel stream sit on one XCD, adjacent in dispatch order (they read the same input from that
// XCD's L2 at about the same time: measured 18 MB of HBM fetch for the 33.5 MB q|k|v input read by six cout tiles).
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define RW_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RW_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define RW_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void rw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int RW_CHUNK = 64 * 256;      // 64 pixels x 128 channels bf16
constexpr int RW_RO = 64 * 256;         // 64 pixels x 128 couts bf16 (4 KB per wave)

// NCH: 128-channel chunks per tile (K / 128); R: ring slots (R - 1 <= 2 NCH: the stream runs up to two tiles ahead);
// RES: residual present
template <int NCH, int R, bool RES>
__global__ __launch_bounds__(256, ((NCH > 3 || R > 4) ? 1 : 2)) void conv1x1_rw_kernel(ConvArgs p) {
    static_assert(R - 1 <= 4 * NCH && R >= 3 && 4 * (R - 2) + 4 * (RES ? 8 : 4) < 64, "ring depth");
    constexpr int BOPS = RES ? 8 : 4;            // per-wave vmem ops at a tile boundary: 4 row stores (+ 4 residual DMAs)
    constexpr int NYOUNG = 4 * (R - 2);          // DMAs of the R-2 chunks issued after the one being awaited
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const ro = smem + R * RW_CHUNK + wave * 4096;       // this wave's [64 px][32 co] slice
    // block -> (cout tile, pixel stream): blocks b, b+8, .. share an XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cot = j % p.CT;
    const int nstreams = (gridDim.x >> 3) / p.CT * 8;
    const int stream = (j / p.CT) * 8 + xcd;
    const int ntiles = stream < p.PT ? (p.PT - stream + nstreams - 1) / nstreams : 0;   // tiles stream, stream + nstreams, ..
    if (ntiles == 0) return;

    // ---- resident weight fragments (A operand) and bias of this wave's 32 couts
    bf16x8 A[NCH * 8];
    {
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.w) + (size_t)(cot * 4 + wave) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NCH * 8; ++ks) A[ks] = wf[(size_t)ks * p.CB * 64];
    }
    f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[g][e] = 0.f;
        if (p.bias) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + cot * 128 + wave * 32 + 8 * g + 4 * h);
    }
    const float slope = dxmi_act_slope(p.act);

    // ---- DMA roles.  Chunk image: pixel row = 256 B = 16 slots; channel piece s of pixel px sits in slot s ^ (px & 15)
    // (b128 fragment reads conflict-free).  Instruction u (0..3) of wave w moves pixels (4w + u) * 4 .. + 3.
    int dpx[4], ds8[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        dpx[u] = (wave * 4 + u) * 4 + (lane >> 4);
        ds8[u] = ((lane & 15) ^ (dpx[u] & 15)) * 8;
    }
    auto issue_chunk = [&](int tile, int c, int slot) {          // chunk c of the stream's tile-th tile -> ring slot
        const size_t P0 = (size_t)(stream + tile * nstreams) * 64;
        const int cbase = c * 128;
        const bool first = cbase < p.C0;
        const bf16* src = first ? p.in0 : p.in1;
        const int Cs = first ? p.C0 : p.C1;
        const int coff = first ? cbase : cbase - p.C0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_amdgcn_global_load_lds(RW_GPTR(src + (P0 + dpx[u]) * Cs + coff + ds8[u]),
                                             RW_LPTR(ring + slot * RW_CHUNK + (wave * 4 + u) * 1024), 16, 0, 0);
    };
    // wave-private output / residual slice: pixel row = 64 B = 4 slots, cout piece c of pixel px in slot c ^ ((px >> 1) & 3);
    // instruction i (0..3) moves pixels 16 i .. 16 i + 15
    int rpx[4], rc8[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rpx[i] = i * 16 + (lane >> 2);
        rc8[i] = ((lane & 3) ^ ((rpx[i] >> 1) & 3)) * 8;
    }
    auto tile_off = [&](int tile, int i) -> size_t {
        return ((size_t)(stream + tile * nstreams) * 64 + rpx[i]) * p.Cout + cot * 128 + wave * 32 + rc8[i];
    };
    auto issue_residual = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(RW_GPTR(p.residual + tile_off(tile, i)), RW_LPTR(ro + i * 1024), 16, 0, 0);
    };

    // ---- prologue: residual of tile 0, then chunks 0 .. R-2 of the stream
    const int gtot = ntiles * NCH;               // chunks of the whole stream
    if (RES) issue_residual(0);
#pragma unroll
    for (int g = 0; g < R - 1; ++g)
        if (g < gtot) issue_chunk(g / NCH, g % NCH, g);

    // B-operand read offsets: pixel nb*32 + (lane & 31), k-step ks -> slot (ks*2 + h) ^ (px & 15); px & 15 is the same for
    // both pixel blocks, so block 1 is block 0 + 32 rows
    int boff[8];
    {
        const int px = lane & 31;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) boff[ks] = px * 256 + (((ks * 2 + h) ^ (px & 15)) << 4);
    }

    int slot = 0;                                // ring slot of the chunk being consumed
    for (int ti = 0; ti < ntiles; ++ti) {
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int g = ti * NCH + c;
            // chunk g landed: in order behind it are the R-2 younger chunks and the stores / residual DMAs of every tile
            // boundary passed since chunk g was issued R-1 steps ago (boundary d tiles back counts when d*NCH < R-1-c: at most
            // two); near the end of the stream fewer chunks are younger
            const int kmax = R - 1 - c > 0 ? (R - 1 - c + NCH - 1) / NCH : 0;
            const int nbnd = ti < kmax ? ti : kmax;
            if (g + R - 2 >= gtot) RW_WAIT_VM(0);
            else if (nbnd == 0) RW_WAIT_VM(NYOUNG);
            else if (nbnd == 1) RW_WAIT_VM(NYOUNG + BOPS);
            else if (nbnd == 2) RW_WAIT_VM(NYOUNG + 2 * BOPS);
            else if (nbnd == 3) RW_WAIT_VM(NYOUNG + 3 * BOPS);
            else RW_WAIT_VM(NYOUNG + 4 * BOPS);
            rw_barrier();                        // every wave's pieces of chunk g landed; every wave is done with chunk g-1
            {
                const int gi = g + R - 1;        // refill the slot chunk g-1 vacated
                const int si = slot == 0 ? R - 1 : slot - 1;
                if (gi < gtot) issue_chunk(ti + (c + R - 1) / NCH, (c + R - 1) % NCH, si);
            }
            const char* img = ring + slot * RW_CHUNK;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(img + boff[ks] + nb * (32 * 256));
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c * 8 + ks], b, acc[nb], 0, 0, 0);
                }
            slot = slot + 1 == R ? 0 : slot + 1;
        }
        // ---- tile end, wave-private.  The residual DMAs of this tile are older than the 4*NCH chunk DMAs issued since
        // the boundary (fewer near the end of the stream).
        if (RES) {
            if ((ti + 1) * NCH + R - 1 > gtot) RW_WAIT_VM(0);
            else RW_WAIT_VM(4 * NCH);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int px = nb * 32 + (lane & 31);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                char* a = ro + px * 64 + ((g4 ^ ((px >> 1) & 3)) << 4) + 8 * h;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[nb][4 * g4 + e] + bv[g4][e];
                if (RES) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)dxmi_act_lin(v[e], slope);
                *reinterpret_cast<bf16x4*>(a) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slice a wave drains is the slice it wrote
        {
            bf16x8 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const bf16x8*>(ro + i * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + tile_off(ti, i)) = v[i];
        }
        if (RES && ti + 1 < ntiles) issue_residual(ti + 1);     // into the slice just drained (its reads are complete)
    }
}

template <int NCH, int R>
int rw_launch(const ConvArgs& b, int grid, hipStream_t st) {
    const size_t lds = (size_t)R * RW_CHUNK + RW_RO;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_rw_kernel<NCH, R, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_rw_kernel<NCH, R, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (b.residual) hipLaunchKernelGGL((conv1x1_rw_kernel<NCH, R, true>), dim3(grid), dim3(256), lds, st, b);
    else hipLaunchKernelGGL((conv1x1_rw_kernel<NCH, R, false>), dim3(grid), dim3(256), lds, st, b);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(1x1 rw)");
    return DXMI_OK;
}

}  // namespace

// Launches the register-weights 1x1 kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv1x1_rw_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV1X1_RW") ? atoi(getenv("DXMI_CONV1X1_RW")) : 1;   // 0: conv1x1_stream_kernel for every shape
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.ksize != 1 || a.stride != 1 || a.pad != 0 || a.ups != 0 || a.mask_src || a.addvec || a.act == DXMI_ACT_SILU) return 1;
    const int K = a.C0 + a.C1;
    if (K % 128 != 0 || a.C0 % 128 != 0 || K > 512 || a.Cout % 128 != 0) return 1;
    const long px = (long)a.N * a.OH * a.OW;
    if (px % 64 != 0 || px / 64 < 512) return 1;          // small maps: the per-tile kernel fills the chip better
    if (kernel_id) {
        const int nch = K / 128;
        *kernel_id = 500000 + nch * 1000 + (nch > 3 ? 6 : 3) * 10 + (a.residual ? 1 : 0);   // conv1x1_rw_kernel<NCH, R, RES>
        return DXMI_OK;
    }
    ConvArgs b = a;
    b.PT = (int)(px / 64);
    b.CT = a.Cout / 128;
    b.tile_px = 64;
    if (b.CT > 32) return 1;
    const int per_xcd = K > 384 ? 32 : 64;                // co-resident workgroups per XCD (one / two per CU)
    const int grid = 8 * b.CT * (per_xcd / b.CT);         // whole XCD groups of CT cout tiles
    switch (K / 128) {
    // two chunks (32 KB) in flight per workgroup, 64 KB of LDS -> two workgroups per CU.  Measured at 256 -> 768 @16x16, B = 256:
    // one workgroup per CU with seven chunks in flight 60 us, two with two chunks each 42 us — a workgroup's own issue
    // chain (DMA issue + 16 MFMAs + epilogue per chunk), not load latency, sets the tile time, so occupancy wins over depth
    case 1: return rw_launch<1, 3>(b, grid, st);
    case 2: return rw_launch<2, 3>(b, grid, st);
    case 3: return rw_launch<3, 3>(b, grid, st);
    default: return rw_launch<4, 6>(b, grid, st);         // K = 512: 128 weight registers, one workgroup per CU, deeper ring
    }
}
