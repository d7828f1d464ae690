// 1x1 convolution with register-resident weights, eight waves x 32 couts (256-cout tiles), for K = 384 / 512 / 576 (round 5): the
// q|k|v / proj_out / skip_connection GEMMs [pixels x K] . [K x Cout] of the ADM nets that conv1x1_rw_kernel does not take
// (/root/reference models/cm/unet.py:240-260 skip_connection, :320-332 AttentionBlock qkv / proj_out).
//
// conv1x1_rw_kernel (K <= 512, four waves x 32 couts, TWO workgroups per CU) left these shapes to conv1x1_stream_kernel (64-pixel x
// 128-cout tiles that re-fetch their 128 x K weight slab per tile: 0.19-0.27 of the MFMA peak).  The register-resident form needs the
// second wave per SIMD that conv1x1_rw_kernel gets from its second workgroup: a wave issuing its chunk DMAs (~190 cycles each)
// stalls its own MFMA stream, and only a sibling wave's MFMAs fill the matrix pipe meanwhile (a one-workgroup, one-wave-per-SIMD
// version of this file measured SLOWER than the stream kernel: DESIGN 5.5).  Here ONE 512-thread workgroup per CU:
//   * wave w keeps the K / 16 fragments of couts 32w .. 32w+31 of a 256-cout tile in VGPRs (144 at K = 576; K = 768 would be 192
//     and spills: not enabled);
//   * the input (64-pixel x 128-channel chunks, 16 KB) travels global -> LDS by DMA through a ring of six chunks, TWO DMAs per wave
//     and chunk, one barrier per chunk for 16 MFMAs per wave; a chunk feeds 256 couts, so the L2 -> LDS feed per MFMA is half of
//     conv1x1_rw_kernel's;
//   * 16-byte pieces are addressed per lane: the virtual concat [in0 | in1] may split anywhere (C0 % 8 == 0), K % 128 == 64 leaves
//     the last chunk half full (its unused pieces re-read a valid address);
//   * Cout % 256 != 0 (multiples of 32): waves past the last cout keep moving data and meeting barriers, run the MFMAs on clamped
//     fragments (a branch around them makes hipcc shuffle the accumulators between register files) and skip epilogue and stores;
//   * bias through an LDS table, wave-private 4 KB output / residual slice, exact in-order vmcnt counts (conv1x1_rw_kernel's).
#include "conv_common.h"
#include <stdlib.h>

namespace {

#define R8_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define R8_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define R8_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void r8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int R8_CHUNK = 64 * 256;      // 64 pixels x 128 channels bf16
constexpr int R8_RMAX = 6;              // ring slots (fewer when a tile has fewer chunks: at most one tile boundary inside the ring)
constexpr int r8_ring(int nk) { return (nk + 1) / 2 + 1 < R8_RMAX ? (nk + 1) / 2 + 1 : R8_RMAX; }
constexpr int r8_lds(int nk) { return r8_ring(nk) * R8_CHUNK + 8 * 4096 + 1024; }      // ring + eight output slices + bias[256]

// NK: K / 64 (6, 8, 9); RES: residual present
template <int NK, bool RES>
__global__ __launch_bounds__(512, 1) void conv1x1_rw8_kernel(ConvArgs p) {
    constexpr int NCH = (NK + 1) / 2;            // 128-channel chunks; the last one is half full when NK is odd
    constexpr bool TAIL = (NK & 1) != 0;
    constexpr int R = r8_ring(NK);
    constexpr int BOPS = RES ? 8 : 4;            // per-wave vmem ops at a tile boundary: 4 row stores (+ 4 residual DMAs)
    constexpr int NYOUNG = 2 * (R - 2);          // DMAs of the R-2 chunks issued after the one being awaited
    static_assert(NCH >= 3 && R - 1 <= NCH && NYOUNG + BOPS < 64, "ring depth: at most one tile boundary inside the ring");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;
    float* const btab = reinterpret_cast<float*>(smem + R * R8_CHUNK + 8 * 4096);

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const ro = smem + R * R8_CHUNK + wave * 4096;       // this wave's [64 px][32 co] slice
    // block -> (cout tile, pixel stream): blocks b, b+8, .. share an XCD
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int cot = j % p.CT;
    const int nstreams = (gridDim.x >> 3) / p.CT * 8;
    const int stream = (j / p.CT) * 8 + xcd;
    const int ntiles = stream < p.PT ? (p.PT - stream + nstreams - 1) / nstreams : 0;   // tiles stream, stream + nstreams, ..
    if (ntiles == 0) return;
    const int co0 = cot * 256 + wave * 32;
    const bool active = co0 < p.Cout;            // wave-uniform

    if (tid < 256) btab[tid] = (p.bias && cot * 256 + tid < p.Cout) ? p.bias[cot * 256 + tid] : 0.f;
    // ---- resident weight fragments (A operand) of this wave's 32 couts
    bf16x8 A[NK * 4];
    {
        const int cb = active ? cot * 8 + wave : p.CB - 1;
        const bf16x8* wf = reinterpret_cast<const bf16x8*>(p.w) + (size_t)cb * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NK * 4; ++ks) A[ks] = wf[(size_t)ks * p.CB * 64];
    }
    const float slope = dxmi_act_slope(p.act);

    // ---- DMA roles.  Chunk image: pixel row = 256 B = 16 slots; channel piece s of pixel px sits in slot s ^ (px & 15)
    // (b128 fragment reads conflict-free).  Instruction u (0, 1) of wave w moves pixels (2w + u) * 4 .. + 3.
    const int dpx0 = wave * 8 + (lane >> 4);                     // u = 1: + 4
    const int K = NK * 64;
    auto issue_chunk = [&](int tile, int c, int slot) {          // chunk c of the stream's tile-th tile -> ring slot
        const size_t P0 = (size_t)(stream + tile * nstreams) * 64;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int dpx = dpx0 + 4 * u;
            int ch = c * 128 + (((lane & 15) ^ (dpx & 15)) << 3);
            if (TAIL) ch = ch < K - 8 ? ch : K - 8;              // the empty half of the last chunk: any valid address
            const bf16* g = ch < p.C0 ? p.in0 + (P0 + dpx) * p.C0 + ch : p.in1 + (P0 + dpx) * p.C1 + (ch - p.C0);
            __builtin_amdgcn_global_load_lds(R8_GPTR(g), R8_LPTR(ring + slot * R8_CHUNK + (wave * 2 + u) * 1024), 16, 0, 0);
        }
    };
    // wave-private output / residual slice: pixel row = 64 B = 4 slots, cout piece c of pixel px in slot c ^ ((px >> 1) & 3);
    // instruction i (0..3) moves pixels 16 i + (lane >> 2): (px >> 1) & 3 = (lane >> 3) & 3 for every i
    const int rpx0 = lane >> 2, rc8 = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;
    auto tile_off = [&](int tile, int i) -> size_t {
        return ((size_t)(stream + tile * nstreams) * 64 + rpx0 + 16 * i) * p.Cout + co0 + rc8;
    };
    auto issue_residual = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(R8_GPTR(p.residual + tile_off(tile, i)), R8_LPTR(ro + i * 1024), 16, 0, 0);
    };

    // ---- prologue: residual of tile 0, then chunks 0 .. R-2 of the stream
    const int gtot = ntiles * NCH;               // chunks of the whole stream
    if (RES && active) issue_residual(0);
#pragma unroll
    for (int g = 0; g < R - 1; ++g)
        if (g < gtot) issue_chunk(g / NCH, g % NCH, g);

    // B-operand read base: pixel nb*32 + (lane & 31), k-step ks -> slot (ks*2 + h) ^ (px & 15)
    const int bpx = lane & 31, bsw = bpx & 15;
    const int brow = bpx * 256;

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
            // chunk g landed: in order behind it are the R-2 younger chunks and — for a wave that stores — the stores / residual DMAs
            // of the tile boundary passed since chunk g was issued R-1 steps ago; near the end of the stream fewer chunks are younger
            const bool bnd = active && ti > 0 && c < R - 1;
            if (g + R - 2 >= gtot) R8_WAIT_VM(0);
            else if (!bnd) R8_WAIT_VM(NYOUNG);
            else R8_WAIT_VM(NYOUNG + BOPS);
            r8_barrier();                        // every wave's pieces of chunk g landed; every wave is done with chunk g-1
            {
                const int gi = g + R - 1;        // refill the slot chunk g-1 vacated
                const int si = slot == 0 ? R - 1 : slot - 1;
                if (gi < gtot) issue_chunk(ti + (c + R - 1) / NCH, (c + R - 1) % NCH, si);
            }
            {
                const char* img = ring + slot * R8_CHUNK + brow;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (TAIL && c == NCH - 1 && ks >= 4) continue;
                    // (keeps hipcc from hoisting all 16 B-fragment reads of a chunk in front of its MFMAs)
                    if (ks == 4) __builtin_amdgcn_sched_barrier(0);
                    const int so = ((ks * 2 + h) ^ bsw) << 4;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(img + so + nb * (32 * 256));
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[c * 8 + ks], b, acc[nb], 0, 0, 0);
                    }
                }
            }
            slot = slot + 1 == R ? 0 : slot + 1;
        }
        if (!active) continue;
        // ---- tile end, wave-private.  The residual DMAs of this tile are older than the 2*NCH chunk DMAs issued since
        // the boundary (fewer near the end of the stream).
        if (RES) {
            if ((ti + 1) * NCH + R - 1 > gtot) R8_WAIT_VM(0);
            else R8_WAIT_VM(2 * NCH);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int px = nb * 32 + (lane & 31);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                char* a = ro + px * 64 + ((g4 ^ ((px >> 1) & 3)) << 4) + 8 * h;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(btab + wave * 32 + 8 * g4 + 4 * h);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[nb][4 * g4 + e] + bv[e];
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

template <int NK>
int r8_launch(const ConvArgs& b, int grid, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_rw8_kernel<NK, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_rw8_kernel<NK, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (b.residual) hipLaunchKernelGGL((conv1x1_rw8_kernel<NK, true>), dim3(grid), dim3(512), (size_t)r8_lds(NK), st, b);
    else hipLaunchKernelGGL((conv1x1_rw8_kernel<NK, false>), dim3(grid), dim3(512), (size_t)r8_lds(NK), st, b);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(1x1 rw8)");
    return DXMI_OK;
}

}  // namespace

// Launches the eight-wave register-weights 1x1 kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv1x1_rw8_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV1X1_RW8") ? atoi(getenv("DXMI_CONV1X1_RW8")) : 1;   // 0: conv1x1_stream_kernel for these shapes
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.ksize != 1 || a.stride != 1 || a.pad != 0 || a.ups != 0 || a.mask_src || a.addvec || a.act == DXMI_ACT_SILU || a.gn_stats || a.gn_out) return 1;
    const int K = a.C0 + a.C1;
    const long px_ = (long)a.N * a.OH * a.OW;
    // K = 576: always (conv1x1_stream_kernel otherwise: 85 -> 70 us on 576 -> 1728 @16x16, 206 -> 124 us on 576 -> 192 @64x64, B = 100).
    // K = 384 / 512: where conv1x1_rw_kernel does not take the shape (Cout % 128 != 0: 384 -> 192 @64x64 151 -> 101 us) or, at K = 512,
    // Cout >= 1024 (512 -> 1536 @32x32: 283 -> 189 us); on conv1x1_rw_kernel's other shapes its two workgroups per CU win
    // (384 -> 1152 @32x32: 119 us there, 131 us here).  K = 768 spills 76-118 registers here (41 -> 76 us): conv1x1_stream_kernel.
    const bool rw_takes = K % 128 == 0 && a.C0 % 128 == 0 && K <= 512 && a.Cout % 128 == 0 && px_ % 64 == 0 && px_ / 64 >= 512;
    const bool kok = K == 576 || ((K == 384 || K == 512) && (!rw_takes || (K == 512 && a.Cout >= 1024)));
    if (!kok || a.C0 % 8 != 0 || a.C1 % 8 != 0 || a.Cout % 32 != 0) return 1;
    const long px = (long)a.N * a.OH * a.OW;
    if (px % 64 != 0) return 1;
    ConvArgs b = a;
    b.PT = (int)(px / 64);
    b.CT = (a.Cout + 255) / 256;
    b.tile_px = 64;
    if (b.CT > 32) return 1;
    const int grid = 8 * b.CT * (32 / b.CT);              // one workgroup per CU, whole XCD groups of CT cout tiles
    const int nstreams = (grid >> 3) / b.CT * 8;
    // every pixel stream needs a few tiles to pay for its 256 x K weight load (small maps / batches stay on the per-tile kernel)
    if (b.PT < 2 * nstreams) return 1;
    if (kernel_id) {
        *kernel_id = 550000 + (K / 64) * 10 + (a.residual ? 1 : 0);   // conv1x1_rw8_kernel<NK, RES>
        return DXMI_OK;
    }
    switch (K / 64) {
    case 6: return r8_launch<6>(b, grid, st);
    case 8: return r8_launch<8>(b, grid, st);
    default: return r8_launch<9>(b, grid, st);
    }
}
