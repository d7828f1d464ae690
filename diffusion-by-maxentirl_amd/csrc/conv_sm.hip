// 3x3 convolutions on the small maps of the U-Nets (4x4 and 8x8) for gfx950: tiled for the L2 -> CU FEED, not for the MFMAs.
//
// Why (measured on MI355X): at 256 images a 4x4-map conv is a 256 x 4096 x 2304 GEMM — 2 us of MFMA time —
// and every kernel that served it (conv_pipe 64-pixel x 128-cout tiles: 19-26 us; an input-resident 4-image x 64-cout variant:
// 25 us) ran at the rate its workgroups could PULL their operands through one CU's memory pipe: a workgroup needs
// Cin*2*(9*Mt + Nt) bytes for an Mt-cout x Nt-pixel tile, and with Mt*Nt fixed by "one tile per CU" that is minimal at
// Mt ~ Nt/9.  Hence:
//   * tile = 32 couts x 128 pixels (eight 4x4 images / two 8x8 images): 213 KB per tile at Cin = 256, a third of the 64 x 128
//     tiling's bytes; the cout tile of a workgroup is blockIdx % 8, so each XCD's L2 serves ONE 32-cout slice of the weights
//     (147 KB) to its 32 workgroups;
//   * every operand reaches LDS by DMA (`global_load_lds`, no staging registers) exactly once per workgroup and is shared by
//     its four waves: per 32-channel chunk the 18 pre-packed weight fragments of the nine taps (18 KB, 32x32x16 A-fragment
//     order of pack_conv_weight, read as they lie) and the tile's 128 input pixels x 64 B (8 KB; zero padding is a per-lane
//     pointer to a zero slot, no halo copy);
//   * a ring of five chunk slots: four chunks (104 KB) are in flight per CU while one is consumed, each wave issues its share
//     (6-7 DMAs per chunk) and waits with an exact in-order vmcnt count; ONE barrier per chunk (18 MFMAs per wave);
//   * wave w owns pixels 32 w .. 32 w + 31 of the tile: one 32x32 accumulator block, `v_mfma_f32_32x32x16_bf16`;
//   * persistent over (pixel tile, cout tile) pairs with the DMA stream running across tile ends: residual tile, bias and
//     the images' temb rows arrive by DMA with the tile's last chunk (a global load in the epilogue would wait for every
//     older DMA in flight: vmcnt retires in order).
// Input pixel ps sits at ps * 64 B with channel piece j in 16-byte slot (j + (ps >> 2)) & 3: the ds_read_b128 lane groups of
// the B fragments (32 pixels x one 8-channel piece) are then bank-conflict free for every tap on both map sizes.
// Results do not depend on the batch an image rides in (fixed K order per output element).
//
// Scope: 3x3 / stride 1 / pad 1, no upsample, 4x4 or 8x8 maps, NHWC bf16 in (virtual concat) and out, C0 % 32 == 0,
// C1 % 32 == 0, Cout % 32 == 0, >= 5 chunks, bias / temb / residual / linear activation.  Everything else: conv_ws8 / conv_pipe.
#include "conv_common.h"
#include <stdlib.h>

namespace {

// per cout-tile width MT (32 or 64): weight fragments of a chunk = 9 taps x 2 k16-steps x MT/32 blocks of 1 KiB
template <int MT>
struct SmCfg {
    static constexpr int CBT = MT / 32;                      // 32-cout blocks per tile
    static constexpr int NWP = 18 * CBT;                     // weight pieces per chunk
    static constexpr int WB = NWP * 1024;                    // bytes of a chunk's weights
    static constexpr int IB = 8 * 1024;                      // input block of a chunk: 128 pixels x 64 B
    static constexpr int ZERO = WB + IB;                     // 64 zero bytes inside a chunk slot
    static constexpr int SLOT = WB + IB + 1024;              // weights | input | zero slot | pad
    static constexpr int R = MT == 32 ? 5 : 3;               // ring slots (chunks)
    static constexpr int NRP = 8 * CBT;                      // residual pieces: 128 pixels x MT couts bf16
    static constexpr int RES = NRP * 1024;
    static constexpr int TB = 3 * 1024;                      // bias[MT] | temb[images][MT] fp32: <= 9 rows of 256 B
    static constexpr int LDS = R * SLOT + RES + TB;
};

__device__ uint4 sm_zero16 = {0u, 0u, 0u, 0u};   // DMA source of out-of-range images and absent bias / temb rows

#define SM_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SM_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void sm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(n) for a wave-uniform run-time n <= 30, rounded DOWN to an even count (waits for more, never less)
__device__ __forceinline__ void sm_wait_vm(int n) {
    switch (n >> 1) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    }
}

// ML = log2(map width): 2 (4x4, 8 images per tile) or 3 (8x8, 2 images per tile); NL loader waves behind the 4 MFMA waves.
// Roles are split as in conv_ws.hip: a DMA costs its wave 60-185 cycles of ISSUE, so a wave that both loads and computes spends
// more time issuing its 6-7 pieces per chunk than on the chunk's 18 MFMAs (first version of this kernel, all four waves
// doing both: 2 600 cycles per chunk = 10 B/clk per CU).
template <int ML, int NL, int MT>
__global__ __launch_bounds__(256 + 64 * NL) void conv_sm_kernel(ConvArgs p) {
    typedef SmCfg<MT> Cfg;
    constexpr int MW = 1 << ML, HW = MW * MW, IMGS = 128 / HW, CBT = Cfg::CBT;
    constexpr int SM_R = Cfg::R, SM_SLOT = Cfg::SLOT, SM_WB = Cfg::WB, SM_ZERO = Cfg::ZERO;
    constexpr int TROW = MT * 4;                                  // bytes of a table row (MT floats)
    constexpr int NTP = ((1 + IMGS) * TROW + 1023) / 1024;        // table pieces
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const resb = smem + SM_R * SM_SLOT;
    float* const tb = reinterpret_cast<float*>(resb + Cfg::RES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = (p.C0 + p.C1) >> 5;
    const int ntiles = p.PT * p.CT;                  // (128-pixel tile, 32-cout tile) pairs, cout fastest
    const int q0 = blockIdx.x, qstride = gridDim.x;
    if (q0 >= ntiles) return;
    const int mytiles = (ntiles - 1 - q0) / qstride + 1;
    const int total = mytiles * nchunks;             // chunks this workgroup consumes, in stream order
    const bool has_res = p.residual != nullptr;

    // zero slots of the ring (read by lanes whose tap falls outside the map)
    if (tid < SM_R * 4) reinterpret_cast<uint4*>(smem + (tid >> 2) * SM_SLOT + SM_ZERO)[tid & 3] = uint4{0u, 0u, 0u, 0u};

    if (wave >= 4) {
        // ============================================================ loader waves (NL = 8)
        // pieces of a chunk (1 KiB each): 18 weight fragments f = tap * 2 + k16-step, 8 input pieces of 16 pixels; with a tile's
        // LAST chunk also the 8 pieces of the residual tile and the 2 pieces of the bias / temb table.  Loader l owns weight
        // fragments l, l + 8 (and l + 16 for l < 2), input piece l, residual piece l, table piece l (l < 2): 3-4 pieces per
        // ordinary chunk.  One wave sustains one DMA per ~190 cycles whatever it keeps in flight (tools/dma_probe.hip: 12.9 GB/s
        // per wave, 50 GB/s per CU at 4 waves, 67 at 8, 75 at 12), so the loaders' instruction stream is kept to the DMAs: every
        // per-lane offset is formed once, per-tile bases once per tile, per-chunk sources are one 64-bit add.
        static_assert(NL == 8, "static piece roles are written for 8 loader waves");
        const int l = wave - 4;
        const int NW = (Cfg::NWP - l + 7) / 8 + 1;                                 // pieces of an ordinary chunk: weights + one input piece
        const char* const zero_page = reinterpret_cast<const char*>(p.mask_src);   // 16 zero bytes in global memory (host: &sm_zero16)
        // input piece l: lane -> pixel ps of the tile; LDS slot (lane & 3) holds channel piece (slot - (ps >> 2)) & 3
        const int ps = l * 16 + (lane >> 2), img = ps >> (2 * ML);
        const int jin = ((lane & 3) - (ps >> 2)) & 3;
        const int in_off0 = (ps * p.C0 + jin * 8) * 2, in_off1 = (ps * p.C1 + jin * 8) * 2;
        // residual pieces l, l + 8 (MT = 64): 1 KiB = (1024 / (MT * 2)) pixels x MT couts; lane -> (pixel, 8-cout piece)
        constexpr int RLP = MT / 8;                                                // lanes per residual pixel row
        int res_off[CBT], res_img[CBT];
#pragma unroll
        for (int k = 0; k < CBT; ++k) {
            const int rp = (l + 8 * k) * (64 / RLP) + lane / RLP;
            res_off[k] = (rp * p.Cout + (lane % RLP) * 8) * 2;
            res_img[k] = rp >> (2 * ML);
        }
        // table piece l (l < NTP): rows of MT floats, row 0 bias, row 1 + i the temb row of image i; lane -> (row, 4 floats)
        constexpr int TLP = MT / 4;                                                // lanes per table row
        const int trow = l * (64 / TLP) + lane / TLP, tcol = (lane % TLP) * 4;
        const size_t tap_stride = (size_t)p.KST * p.CB * 1024, ks_stride = (size_t)p.CB * 1024;
        // issue-side position in the stream
        int ic = 0, iq = q0;
        const char *w_t = nullptr, *in0_t = nullptr, *in1_t = nullptr, *res_t = nullptr;
        bool img_ok = false;
        int cot = 0, n0 = 0;
        auto plan_tile = [&]() {
            cot = iq % p.CT;
            n0 = (iq / p.CT) * IMGS;
            img_ok = n0 + img < p.N;
            w_t = reinterpret_cast<const char*>(p.w) + (size_t)cot * CBT * 1024 + (size_t)lane * 16;
            in0_t = reinterpret_cast<const char*>(p.in0) + (size_t)n0 * HW * p.C0 * 2;
            in1_t = p.in1 ? reinterpret_cast<const char*>(p.in1) + (size_t)n0 * HW * p.C1 * 2 : nullptr;
            res_t = has_res ? reinterpret_cast<const char*>(p.residual) + ((size_t)n0 * HW * p.Cout + cot * MT) * 2 : nullptr;
        };
        plan_tile();
        auto issue = [&](int gslot) {
            char* const slot = smem + gslot * SM_SLOT;
            const char* const wc = w_t + (size_t)(ic * 2) * ks_stride;
            // weight fragments f = l, l + 8, ...: f = (tap * 2 + ks) * CBT + cout block
#pragma unroll
            for (int k = 0; k < (Cfg::NWP + 7) / 8; ++k) {
                const int f = l + 8 * k;
                if (f < Cfg::NWP) {
                    const int tk = f / CBT, cb = f % CBT;
                    __builtin_amdgcn_global_load_lds(SM_GPTR(wc + (size_t)(tk >> 1) * tap_stride + (size_t)(tk & 1) * ks_stride + cb * 1024),
                                                     SM_LPTR(slot + f * 1024), 16, 0, 0);
                }
            }
            {
                const int cbase = ic * 32;
                const bool first = cbase < p.C0;
                const char* src = first ? in0_t + cbase * 2 + in_off0 : in1_t + (cbase - p.C0) * 2 + in_off1;
                if (!img_ok) src = zero_page;
                __builtin_amdgcn_global_load_lds(SM_GPTR(src), SM_LPTR(slot + SM_WB + l * 1024), 16, 0, 0);
            }
            if (ic == nchunks - 1) {
                if (has_res) {
#pragma unroll
                    for (int k = 0; k < CBT; ++k) {
                        const char* src = n0 + res_img[k] < p.N ? res_t + res_off[k] : zero_page;
                        __builtin_amdgcn_global_load_lds(SM_GPTR(src), SM_LPTR(resb + (l + 8 * k) * 1024), 16, 0, 0);
                    }
                }
                if (l < NTP) {
                    const char* src = zero_page;
                    if (trow == 0) { if (p.bias) src = reinterpret_cast<const char*>(p.bias + cot * MT + tcol); }
                    else if (trow <= IMGS && p.addvec && n0 + trow - 1 < p.N)
                        src = reinterpret_cast<const char*>(p.addvec + (size_t)(n0 + trow - 1) * p.addvec_ld + cot * MT + tcol);
                    __builtin_amdgcn_global_load_lds(SM_GPTR(src), SM_LPTR(reinterpret_cast<char*>(tb) + l * 1024), 16, 0, 0);
                }
            }
            if (++ic == nchunks) {
                ic = 0;
                iq += qstride;
                if (iq < ntiles) plan_tile();
            }
        };
        int islot = 0, issued = 0;                                              // ring slot of / number of chunks issued so far
        auto issue_next = [&]() {
            issue(islot);
            islot = islot + 1 == SM_R ? 0 : islot + 1;
            ++issued;
        };
        // Only chunk 0 is issued before the first barrier (a loader needs ~190 cycles per DMA: issuing the whole ring first
        // would hold the MFMA waves back by ~1.5 us); the ring then fills two chunks per iteration until R - 1 are in flight.
        issue_next();
        for (int gc = 0; gc < total; ++gc) {
            // this loader's pieces of chunk gc have landed when at most the pieces of the younger issued chunks are outstanding
            // (their ordinary count: the extra pieces of a tile's last chunk only make this wait for a little more)
            sm_wait_vm((issued - 1 - gc) * NW);
            sm_barrier();                       // B_gc: chunk gc landed; the MFMA waves are done reading chunk gc - 1
            // chunks up to gc + R - 1 may be in the ring now (the slot of chunk gc - 1 is free)
#pragma unroll 1
            for (int k = 0; k < 2 && issued < total && issued < gc + SM_R; ++k) issue_next();
        }
        return;
    }

    // ================================================================ MFMA waves: pixels 32 wave .. + 31 of the tile, 32 couts
    // B fragment (32x32x16): lane -> pixel (lane & 31) of this wave's 32, 8-channel piece j = 2 ks + (lane >> 5); per tap the
    // byte offset of the source pixel's slot for ks = 0 inside a chunk slot (ks = 1: offset ^ 32), or the zero slot.
    int boff[9];
    {
        const int px = wave * 32 + (lane & 31);
        const int img = px >> (2 * ML), y = (px >> ML) & (MW - 1), x = px & (MW - 1), jh = lane >> 5;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = yy >= 0 && yy < MW && xx >= 0 && xx < MW;
            const int ps = (img << (2 * ML)) + yy * MW + xx;
            boff[t] = ok ? SM_WB + ps * 64 + (((jh + (ps >> 2)) & 3) << 4) : SM_ZERO;
        }
    }
    f32x16 acc[CBT];
#pragma unroll
    for (int cb = 0; cb < CBT; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const float slope = dxmi_act_slope(p.act);
    const bool plain = p.act == DXMI_ACT_NONE;
    // fused GroupNorm of the output (4x4 maps, 32-cout tiles): an image's 16 pixels are one DPP row, a group's 8 couts the two
    // half-waves' 4 accumulator registers of one g.  The workgroup keeps its cout tile, so gamma / beta live in registers.
    constexpr bool CAN_GN = ML == 2 && MT == 32;
    const bool fuse_gn = CAN_GN && p.gn_out != nullptr;
    f32x4 gam[4], bet[4];
    if (fuse_gn) {
        const int co0 = (q0 % p.CT) * MT + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            gam[g] = *reinterpret_cast<const f32x4*>(p.gn_gamma + co0 + 8 * g);
            bet[g] = *reinterpret_cast<const f32x4*>(p.gn_beta + co0 + 8 * g);
        }
    }
    int c = 0, ti = 0;
    for (int gc = 0; gc < total; ++gc) {
        sm_barrier();                           // B_gc
        const char* const slot = smem + (gc % SM_R) * SM_SLOT;
        const char* const ab = slot + lane * 16;
        // operand reads run AH taps ahead of the MFMAs that consume them (left to itself hipcc emits read, wait, MFMA per
        // fragment: one LDS latency per MFMA, 2 800 cycles per chunk instead of ~600)
        constexpr int AH = CBT == 1 ? 4 : 3;
        bf16x8 A[18 * CBT], B[18];
        auto read_tap = [&](int t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int cb = 0; cb < CBT; ++cb) A[(2 * t + ks) * CBT + cb] = *reinterpret_cast<const bf16x8*>(ab + ((2 * t + ks) * CBT + cb) * 1024);
                // k16 step 1: slot + 2 = offset ^ 32 (the zero slot is 64 bytes)
                B[2 * t + ks] = *reinterpret_cast<const bf16x8*>(slot + (ks ? boff[t] ^ 32 : boff[t]));
            }
        };
#pragma unroll
        for (int t = 0; t < AH; ++t) read_tap(t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + AH < 9) read_tap(t + AH);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int cb = 0; cb < CBT; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(2 * t + ks) * CBT + cb], B[2 * t + ks], acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (++c == nchunks) {
            // ---- tile end: D[cout][pixel]: lane -> pixel (lane & 31), couts 32 cb + 8 g + 4 (lane >> 5) + {0..3}
            const int qq = q0 + ti * qstride;
            const int cot = qq % p.CT, pt = qq / p.CT;
            const int px = wave * 32 + (lane & 31), h = lane >> 5;
            const int n = pt * IMGS + (px >> (2 * ML));
            const float* const trow = tb + MT + (px >> (2 * ML)) * MT;
            bf16* const orow = reinterpret_cast<bf16*>(p.out) + ((size_t)n * HW + (px & (HW - 1))) * p.Cout + cot * MT;
#pragma unroll
            for (int cb = 0; cb < CBT; ++cb) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * cb + 8 * g + 4 * h;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(tb + co), tv = *reinterpret_cast<const f32x4*>(trow + co);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[cb][4 * g + e] + (bv[e] + tv[e]);
                    if (has_res) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(resb + px * (MT * 2) + co * 2);
                        if (p.res_is_mask) {              // data gradient through a LeakyReLU: the fetched tile is the mask source
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= ((float)rv[e] > 0.f ? 1.f : p.mask_slope);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                    }
                    if (!plain) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dxmi_act_lin(v[e], slope);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                    if (n < p.N && !(fuse_gn && (p.gn_flags & 2))) *reinterpret_cast<bf16x4*>(orow + co) = o;
                    if constexpr (CAN_GN) {
                        if (fuse_gn) {
                            // statistics of the ROUNDED output (what a separate GroupNorm launch would read), two-pass, in a
                            // fixed order: 4 in-lane values, the 16 pixels of the DPP row, the other half-wave
                            float x[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = (float)o[e];
                            float s = dxmi_row16_sum((x[0] + x[1]) + (x[2] + x[3]));
                            s += __shfl_xor(s, 32);
                            const float mean = s * (1.f / 128.f);
                            float d[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e] = x[e] - mean;
                            float qv = dxmi_row16_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                            qv += __shfl_xor(qv, 32);
                            const float rstd = rsqrtf(qv * (1.f / 128.f) + p.gn_eps);
                            bf16x4 y;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float ga = gam[g][e] * rstd;           // same form as groupnorm_silu_kernel
                                float t = x[e] * ga + (bet[g][e] - mean * ga);
                                if (p.gn_flags & 1) t = dxmi_silu_fast(t);
                                y[e] = (bf16)t;
                            }
                            if (n < p.N)
                                *reinterpret_cast<bf16x4*>(p.gn_out + ((size_t)n * HW + (px & (HW - 1))) * p.Cout + cot * MT + co) = y;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[cb][4 * g + e] = 0.f;
                }
            }
            c = 0;
            ++ti;
        }
    }
}

}  // namespace

// Launches the small-map kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv_sm_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    // bit 0: 4x4 maps (32-cout tiles), bit 1: every 8x8 map, bit 3: see below, bit 2: 8x8 maps whose conv_ws8 grid (256-pixel x 128-cout tiles) would
    // leave more than half of the CUs idle (round 4: the EDM nets at the train batch of 16 - 24 tiles; 768 -> 768 measured 40.0 us
    // on conv_ws8_kernel, 18.6 us here with 192 workgroups; at 150 tiles conv_ws8_kernel is 5-9 % ahead); 0: conv_ws8 / conv_pipe
    // (knob "conv_sm_mask" / DXMI_CONV_SM)
    const int enabled = dxmi_tuning("conv_sm_mask");
    const bool small_grid8 = a.OW == 8 && (long)((a.N * 64 + 255) / 256) * ((a.Cout + 127) / 128) < 128;
    // bit 3 (on by default, a rule on the LAYER SHAPE only, so an image's result stays independent of its batch): 8x8 maps with
    // >= 1024 channels in and out — the bottom of the LSUN-256 net, whose per-GPU batch of 16 is 32 conv_ws8 tiles; at batch 100
    // this kernel is ~5 % behind conv_ws8_kernel on such layers, at batch 16 it is 2x ahead
    const bool wide8 = a.OW == 8 && a.C0 + a.C1 >= 1024 && a.Cout >= 1024;
    if (!((a.OW == 4 && (enabled & 1)) || (a.OW == 8 && ((enabled & 2) || ((enabled & 4) && small_grid8) || ((enabled & 8) && wide8))))) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups != 0 || (a.mask_src && a.residual) || a.act == DXMI_ACT_SILU || a.gn_stats) return 1;   // a mask alone rides the residual path
    if (a.OH != a.OW || (a.OW != 4 && a.OW != 8) || a.IH != a.OH || a.IW != a.OW) return 1;
    // tile width: 64 couts where that still gives every CU a tile (8x8 maps at batch 256), else 32
    const int imgs = 128 / (a.OH * a.OW);
    const int PT = (a.N + imgs - 1) / imgs;
    const int MT = (a.Cout % 64 == 0 && (long)PT * (a.Cout / 64) >= 256) ? 64 : 32;
    if (a.Cout % MT != 0 || a.C0 % 32 != 0 || a.C1 % 32 != 0) return 1;
    if (a.gn_out && !(a.OW == 4 && MT == 32)) return 1;      // the fused GroupNorm needs whole images per DPP row
    const int nchunks = (a.C0 + a.C1) / 32;
    if (nchunks < 5) return 1;               // the residual / table buffers are refilled R - 1 chunks ahead of their tile's end
    if ((long)a.N * a.OH * a.OW * (a.C0 > a.C1 ? a.C0 : a.C1) * 2 >= (1L << 31)) return 1;
    if (kernel_id) {
        *kernel_id = 450000 + a.OW * 100 + MT;      // conv_sm_kernel<log2 OW, 8, MT>
        return DXMI_OK;
    }
    static const void* zero_page = nullptr;
    if (!zero_page) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(sm_zero16)) != hipSuccess || !zp) {
            dxmi_set_error("dxmi_conv2d_fwd(sm): hipGetSymbolAddress(sm_zero16) failed");
            return DXMI_EINVAL;
        }
        zero_page = zp;
    }
    ConvArgs b = a;
    if (a.mask_src) {
        b.residual = a.mask_src;
        b.res_is_mask = 1;
    }
    b.mask_src = reinterpret_cast<const bf16*>(zero_page);    // the field carries the zero page (a mask source travels in `residual`)
    b.PT = PT;
    b.CT = a.Cout / MT;
    b.tile_px = 128;
    const int ntiles = b.PT * b.CT;
    // persistent grid: one workgroup per CU, a multiple of the cout-tile count so that a workgroup keeps its cout tile (and,
    // at 8 cout tiles, every XCD serves one slice of the weights)
    int grid = ntiles < 256 ? ntiles : (256 / b.CT) * b.CT;
    if (grid < 1) grid = ntiles < b.CT ? ntiles : b.CT;
#define SM_LAUNCH(ML_, MT_)                                                                                                   \
    do {                                                                                                                      \
        static bool attr = false;                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sm_kernel<ML_, 8, MT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv_sm_kernel<ML_, 8, MT_>), dim3(grid), dim3(256 + 64 * 8), SmCfg<MT_>::LDS, st, b);            \
    } while (0)
    if (a.OW == 4) { if (MT == 32) SM_LAUNCH(2, 32); else SM_LAUNCH(2, 64); }
    else { if (MT == 32) SM_LAUNCH(3, 32); else SM_LAUNCH(3, 64); }
#undef SM_LAUNCH
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(sm)");
    return DXMI_OK;
}
