// conv_ws_kernel's design (conv_ws.hip: MFMA waves fed only from LDS, dedicated mover waves, 16x16x32 MFMAs) for the 8x8
// maps of the U-Nets (level 2 of the CIFAR-10 DDPM net: 13 of its 49 convs), where a 256-pixel x 128-cout tile would leave
// only 128 workgroups for 256 CUs.  Here a tile is FOUR whole 8x8 images x 64 couts:
//   * 256 images x 256 couts = 256 tiles: one per CU, same 512-thread workgroup (waves 0-3 MFMA, 4-5 weights, 6-7 the rest);
//   * MFMA wave w owns image w of the tile: 64 couts x 64 pixels (4 x 4 accumulator blocks of 16 x 16), operands by
//     ds_read_b128 at a per-lane base + compile-time offset; a step (one tap of one 32-channel chunk) is 16 MFMAs;
//   * halo image: 4 x (10 x 10) pixels x 64 B = 25 KB per chunk, channel piece j of halo column hx in slot j ^ (hx & 2)
//     (bank-conflict free for every tap at row pitch 10, brute-forced); weight ring 6 x 4 KB; output / residual tile 32 KB;
//     bias + one temb row per image;
//   * everything else as in conv_ws.hip: one barrier per step, movers issue a few DMAs per step, exact in-order vmcnt counts.
// Scope: 3x3 / stride 1 / pad 1 on 8x8 maps, NHWC bf16 in (virtual concat) and out, any batch, Cout % 64 == 0,
// an even number (>= 4) of 32-channel chunks.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>


namespace {

constexpr int W8_HP = 10, W8_HS = 100;              // halo row pitch / pixels per image halo
constexpr int W8_HALO_BLOCKS = 25;                  // 4 x 100 pixels x 64 B = 25 KiB exactly
constexpr int W8_HALO = W8_HALO_BLOCKS * 1024;
constexpr int W8_A_SLOT = 4096;                     // one (chunk, tap): 2 k-steps x 2 cout blocks x 1 KiB fragments
#ifndef W8_RING_SLOTS
#define W8_RING_SLOTS 6
#endif
constexpr int W8_RING = W8_RING_SLOTS;        // must divide 18 (the MFMA waves index slots by the step inside a chunk pair)
constexpr int W8_A_RING = W8_RING * W8_A_SLOT;
constexpr int W8_RO = 256 * 128;                    // 256 px x 64 co bf16
constexpr int W8_TB = 2048;                         // bias[64] | temb[4][64] fp32

__device__ uint4 w8_zero16 = {0u, 0u, 0u, 0u};

#define W8_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define W8_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void w8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 8 MFMAs (16 cycles each) with 2 / 6 operand reads spread between them
#define W8_INTERLEAVE_2()                                       \
    do {                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
        }                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      \
    } while (0)
#define W8_INTERLEAVE_6()                                       \
    do {                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
        }                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      \
    } while (0)

__device__ __forceinline__ void w8_wait_vm(int n) {      // s_waitcnt vmcnt(n), n wave-uniform, rounded DOWN to even (waits for more)
    switch (n >> 1) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

struct W8Tile {
    int cot, n0;
};

// GN: GroupNorm(+SiLU) of the output written INSTEAD of the raw output (dxmi_conv_desc.gn_out with gn_flags bit 1): MFMA wave w
// holds all 64 pixels of its image for the tile's 64 couts = eight whole groups of 8 channels, so the statistics are 16
// in-lane values, one DPP row sum and one exchange with lane ^ 16 per group.
template <bool GN>
__global__ __launch_bounds__(512, 1) void conv_ws8_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const halo0 = smem;
    char* const aring = smem + 2 * W8_HALO;
    char* const ro = aring + W8_A_RING;
    float* const tb = reinterpret_cast<float*>(ro + W8_RO);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = (p.C0 + p.C1) / 32;
    const int S = 9 * nchunks;                       // steps per tile
    const int ntiles = p.PT * p.CT;                  // (4-image tile, 64-cout tile) pairs, cout fastest

    auto tile_of = [&](int q, W8Tile& t) {
        t.cot = q % p.CT;
        t.n0 = (q / p.CT) * 4;
    };

    int q = dxmi_xcd_logical(blockIdx.x, gridDim.x, p.xcd_order);
    if (q >= ntiles) return;
    const int qstride = gridDim.x;

    if (wave < 4) {
        // ================================================================ MFMA waves: image `wave` of the tile, 64 couts
        const int px = lane & 15, kg = lane >> 4;      // B: pixel / channel piece; A: cout / 8-channel group; D: pixel / 4-cout group
        int bbase[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int hx = (px & 7) + kx;
            bbase[kx] = ((wave * W8_HS + (px >> 3) * W8_HP + hx) * 64) + ((kg ^ (hx & 2)) << 4);
        }
        const char* const abase = aring + ((kg >> 1) * 2048) + ((px + 32 * (kg & 1)) << 4);   // + slot*4096 + (cb16>>1)*1024 + (cb16&1)*256
        f32x4 acc[4][4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[cb][nb][r] = 0.f;
        bf16x8 A0[4], A1[4], Bx[2], By[2];
        // operands of step u of a chunk PAIR (u = 0..17: chunk parity u / 9, tap u % 9; u = 18 wraps)
        auto read_a = [&](int u, bf16x8 (&A)[4]) {
            const char* as = abase + (u % W8_RING) * W8_A_SLOT;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) A[cb] = *reinterpret_cast<const bf16x8*>(as + (cb >> 1) * 1024 + (cb & 1) * 256);
        };
        auto read_b = [&](int u, int half, bf16x8 (&B)[2]) {      // 16-pixel blocks 2 half, 2 half + 1: image rows 2 nb, 2 nb + 1
            const int t = u % 9, ky = t / 3, kx = t % 3;
            const char* hb = halo0 + ((u / 9) & 1) * W8_HALO + bbase[kx];
#pragma unroll
            for (int i = 0; i < 2; ++i) B[i] = *reinterpret_cast<const bf16x8*>(hb + ((2 * (half * 2 + i) + ky) * W8_HP) * 64);
        };
        auto mfma8 = [&](const bf16x8 (&A)[4], const bf16x8 (&B)[2], int half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb][half * 2 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb], B[i], acc[cb][half * 2 + i], 0, 0, 0);
        };
        const float slope = dxmi_act_slope(p.act);
        const bool has_res = p.residual != nullptr;

        w8_barrier();                                   // P0: tap 0 and the first halo chunk have landed
        read_a(0, A0);
        read_b(0, 0, Bx);
        for (;;) {
            const bool more = q + qstride < ntiles;
            f32x4 gam[4], bet[4];
            if constexpr (GN) {          // this tile's affine parameters: in flight under the K loop
                const int co0 = (q % p.CT) * 64 + 4 * kg;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    gam[cb] = *reinterpret_cast<const f32x4*>(p.gn_gamma + co0 + cb * 16);
                    bet[cb] = *reinterpret_cast<const f32x4*>(p.gn_beta + co0 + cb * 16);
                }
            }
            for (int c = 0; c < nchunks; c += 2) {
#pragma unroll
                for (int u = 0; u < 18; ++u) {
                    read_b(u, 1, By);
                    if (u & 1) mfma8(A1, Bx, 0); else mfma8(A0, Bx, 0);
                    W8_INTERLEAVE_2();
                    if (u % 3 == 2) w8_barrier();       // end of a group of three steps: the next group's weights (at a chunk end: the next halo image) landed
                    if (u & 1) read_a(u + 1, A0); else read_a(u + 1, A1);
                    read_b(u + 1, 0, Bx);
                    if (u & 1) mfma8(A1, By, 1); else mfma8(A0, By, 1);
                    W8_INTERLEAVE_6();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            w8_barrier();                               // E1: residual tile + bias / temb table of this tile landed
            {
                // per-lane base + compile-time offsets, the residual pieces of a cout block requested together, no activation
                // code when there is none (as conv_ws.hip)
                const bool plain = p.act == DXMI_ACT_NONE;
                char* const rowb = ro + ((wave * 64 + px) << 7) + 8 * (kg & 1);      // pixel row of block 0: blocks are 2 KiB apart
                // run-time switches hoisted out of the 16 (cb, nb) bodies (as conv_ws.hip, round 4): straight-line code per combination
                auto epi = [&](auto RES, auto MASK, auto ACT) {
                    constexpr bool kRes = decltype(RES)::value, kMask = decltype(MASK)::value, kAct = decltype(ACT)::value;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const int co = cb * 16 + 4 * kg;
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(tb + co), t0 = *reinterpret_cast<const f32x4*>(tb + 64 + wave * 64 + co);
                        f32x4 bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = b0[e] + t0[e];
                        char* const a0 = rowb + (((cb * 2 + (kg >> 1)) ^ (px & 7)) << 4);
                        bf16x4 r[4];
                        if constexpr (kRes) {
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) r[nb] = *reinterpret_cast<const bf16x4*>(a0 + nb * 2048);
                        }
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[cb][nb][e] + bv[e];
                            if constexpr (kRes) {
                                if constexpr (kMask) {          // data gradient through a LeakyReLU: the fetched tile is the mask source
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] *= ((float)r[nb][e] > 0.f ? 1.f : p.mask_slope);
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += (float)r[nb][e];
                                }
                            }
                            if constexpr (kAct) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = dxmi_act_lin(v[e], slope);
                            }
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                            if constexpr (GN) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[cb][nb][e] = (float)o[e];     // the value a separate launch would read
                            } else {
                                *reinterpret_cast<bf16x4*>(a0 + nb * 2048) = o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[cb][nb][e] = 0.f;
                            }
                        }
                        if constexpr (GN) {
                            // group = couts cb * 16 + 8 (kg >> 1) .. + 7 over the image's 64 pixels: two-pass statistics, fixed order
                            float s = 0.f;
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) s += (acc[cb][nb][0] + acc[cb][nb][1]) + (acc[cb][nb][2] + acc[cb][nb][3]);
                            s = dxmi_row16_sum(s);
                            s += __shfl_xor(s, 16, 64);
                            const float mean = s * (1.f / 512.f);
                            float qv = 0.f;
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) {
                                float d[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) d[e] = acc[cb][nb][e] - mean;
                                qv += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
                            }
                            qv = dxmi_row16_sum(qv);
                            qv += __shfl_xor(qv, 16, 64);
                            const float rstd = rsqrtf(qv * (1.f / 512.f) + p.gn_eps);
                            float ga[4], be[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                ga[e] = gam[cb][e] * rstd;               // same form as groupnorm_silu_kernel
                                be[e] = bet[cb][e] - mean * ga[e];
                            }
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) {
                                bf16x4 y;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float t = acc[cb][nb][e] * ga[e] + be[e];
                                    if (p.gn_flags & 1) t = dxmi_silu_fast(t);
                                    y[e] = (bf16)t;
                                    acc[cb][nb][e] = 0.f;
                                }
                                *reinterpret_cast<bf16x4*>(a0 + nb * 2048) = y;
                            }
                        }
                    }
                };
                using T_ = std::true_type;
                using F_ = std::false_type;
                const bool is_mask = has_res && p.res_is_mask;
                if (plain && !is_mask) { if (has_res) epi(T_{}, F_{}, F_{}); else epi(F_{}, F_{}, F_{}); }
                else if (is_mask) { if (plain) epi(T_{}, T_{}, F_{}); else epi(T_{}, T_{}, T_{}); }
                else if (has_res) epi(T_{}, F_{}, T_{});
                else epi(F_{}, F_{}, T_{});
            }
            w8_barrier();                               // E2: output tile complete, the bulk movers may drain it
            if (!more) break;
            q += qstride;
        }
        return;
    }

    if (wave < 6) {
        // ================================================================ weight loaders (waves 4, 5): k-step lw of every tap
        const int lw = wave - 4;
        const char* const wb = reinterpret_cast<const char*>(p.w) + (size_t)lane * 16;
        W8Tile cur;
        tile_of(q, cur);
        auto issue_tap = [&](int cot, int g) {
            const int c = g / 9, t = g - c * 9;
            char* dst = aring + (g % W8_RING) * W8_A_SLOT + lw * 2048;
            const char* src = wb + ((size_t)(t * p.KST + c * 2 + lw) * p.CB + cot * 2) * 1024;
#pragma unroll
            for (int f = 0; f < 2; ++f)
                __builtin_amdgcn_global_load_lds(W8_GPTR(src + f * 1024), W8_LPTR(dst + f * 1024), 16, 0, 0);
        };
        // register-staged weight stream handed over in groups of three steps (see conv_ws.hip, round 4): the ring's six slots are
        // two groups; group G + 1 is written from registers (ds_write_b128) while the MFMA waves read group G; its fragments were
        // requested two group steps earlier (two register sets of 6 x 4 registers).  One barrier per three steps.
        static_assert(W8_RING == 6, "group hand-over needs the six-slot ring");
        u32x4 fq[2][3][2];
        auto load_tap = [&](int cot, int g, u32x4 (&f)[2]) {
            const int c = g / 9, t = g - c * 9;
            const char* src = wb + ((size_t)(t * p.KST + c * 2 + lw) * p.CB + cot * 2) * 1024;
#pragma unroll
            for (int k = 0; k < 2; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f[k]) : "v"(src + k * 1024) : "memory");
        };
        auto store_tap = [&](int g, const u32x4 (&f)[2]) {
            char* dst = aring + (g % W8_RING) * W8_A_SLOT + lw * 2048 + lane * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k) *reinterpret_cast<u32x4*>(dst + k * 1024) = f[k];
        };
        const int NG = S / 3;                           // groups per tile (even)
        auto load_group = [&](int G, int cot_cur, int cot_nxt, u32x4 (&f)[3][2]) {
            const bool nx = G >= NG;
            const int g0 = (nx ? G - NG : G) * 3, cot = nx ? cot_nxt : cot_cur;
#pragma unroll
            for (int j = 0; j < 3; ++j) load_tap(cot, g0 + j, f[j]);
        };
        auto store_group = [&](int G, const u32x4 (&f)[3][2]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) store_tap((G & 1) * 3 + j, f[j]);
        };
        // residual pieces of a tile switch (see the bulk movers: piece L = k*128 + t2, t2 = lw*64 + lane): fetched here one
        // group step after the bulk movers drained the piece (conv_ws.hip, round 4)
        const int t2l = lw * 64 + lane;
        const int kpc_l = (16 + nchunks - 2) / (nchunks - 1), pps_l = (kpc_l + 7) >> 3;
        const char* const zero_page_l = reinterpret_cast<const char*>(p.mask_src);
        auto piece_off_l = [&](const W8Tile& t, int k) -> long {
            const int lp = k * 16 + (t2l >> 3);
            const int c8 = (t2l & 7) ^ (lp & 7);
            const int n = t.n0 + (lp >> 6);
            if (n >= p.N) return -1;
            return (long)((((size_t)n * 8 + ((lp >> 3) & 7)) * 8 + (lp & 7)) * p.Cout + t.cot * 64 + c8 * 8);
        };
        bool first_tile_l = true;
        load_group(0, cur.cot, cur.cot, fq[0]);
        load_group(1, cur.cot, cur.cot, fq[1]);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        store_group(0, fq[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        load_group(2, cur.cot, cur.cot, fq[0]);
        w8_barrier();                                       // P0
        for (;;) {
            const bool more = q + qstride < ntiles;
            W8Tile nxt = cur;
            if (more) tile_of(q + qstride, nxt);
            const bool res_here = !first_tile_l && p.residual != nullptr;
            first_tile_l = false;
            for (int G0 = 0; G0 < NG; G0 += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int G = G0 + j;
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // group G + 1 landed (group G + 2's loads stay in flight)
                    store_group(G + 1, fq[(j + 1) & 1]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (res_here && G > 0) {
                        const int cg = (G - 1) / 3, g3 = (G - 1) - cg * 3;     // the group step whose pieces were drained
                        if (cg + 1 < nchunks) {
                            const int k0 = cg * kpc_l < 16 ? cg * kpc_l : 16, k1 = k0 + kpc_l < 16 ? k0 + kpc_l : 16;
                            const int s0 = 3 * g3, s1 = g3 < 2 ? 3 * g3 + 3 : 8;       // steps of the group that carry pieces (t < 8)
                            const int ka = k0 + s0 * pps_l < k1 ? k0 + s0 * pps_l : k1, kb = k0 + s1 * pps_l < k1 ? k0 + s1 * pps_l : k1;
#pragma unroll 1
                            for (int k = ka; k < kb; ++k) {
                                const long o = piece_off_l(cur, k);
                                const void* g = o >= 0 ? (const void*)(p.residual + o) : (const void*)zero_page_l;
                                __builtin_amdgcn_global_load_lds(W8_GPTR(g), W8_LPTR(ro + (k * 128 + lw * 64) * 16), 16, 0, 0);
                            }
                        }
                    }
                    load_group(G + 3, cur.cot, more ? nxt.cot : cur.cot, fq[(j + 1) & 1]);
                    w8_barrier();                                           // end of group step G
                }
            }
            w8_barrier();                                        // E1
            w8_barrier();                                        // E2
            if (!more) break;
            q += qstride;
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ==================================================================== bulk movers (waves 6, 7)
    {
        const int bw = wave - 6;
        const int t2 = bw * 64 + lane;                     // 0..127 over the two waves
        constexpr int HB2 = (W8_HALO_BLOCKS + 1) / 2;      // halo blocks per wave (block bw + 2k)
        // per tile each lane keeps the BYTE offset of its source piece inside either concat part (-1: zero padding): a DMA
        // address per chunk is then one 64-bit add and a select (no 64-bit multiply, no GOT load of the zero page per DMA)
        const char* const zero_page = reinterpret_cast<const char*>(p.mask_src);    // 16 zero bytes (host: &w8_zero16)
        int hoff0[HB2], hoff1[HB2];
        auto halo_plan = [&](const W8Tile& t) {
#pragma unroll
            for (int k = 0; k < HB2; ++k) {
                const int hp = (bw + 2 * k) * 16 + (lane >> 2);
                const int sub = hp / W8_HS, r = hp - sub * W8_HS;
                const int hy = r / W8_HP, hx = r - hy * W8_HP;
                const int iy = hy - 1, ix = hx - 1;
                const bool ok = sub < 4 && t.n0 + sub < p.N && iy >= 0 && ix >= 0 && iy < 8 && ix < 8;
                // nearest x2 upsample in front of the conv (Upsample 4x4 -> 8x8): virtual pixel (iy, ix) is source pixel (iy / 2, ix / 2)
                const int pix = p.ups ? ((t.n0 + sub) * 4 + (iy >> 1)) * 4 + (ix >> 1) : ((t.n0 + sub) * 8 + iy) * 8 + ix;
                const int j8 = ((lane & 3) ^ (hx & 2)) * 8;
                hoff0[k] = ok ? (pix * p.C0 + j8) * 2 : -1;
                hoff1[k] = ok ? (pix * p.C1 + j8) * 2 : -1;
            }
        };
        auto halo_issue = [&](int c, char* buf, int ka, int kb) {     // blocks ka .. kb-1 of this wave's share (compile-time range)
            const int cbase = c * 32;
            const bool first = cbase < p.C0;
            const char* base = reinterpret_cast<const char*>(first ? p.in0 : p.in1) + (first ? cbase : cbase - p.C0) * 2;
#pragma unroll
            for (int k = 0; k < HB2; ++k) {
                if (k < ka || k >= kb) continue;
                const int blk = bw + 2 * k;
                const int off = first ? hoff0[k] : hoff1[k];
                const char* g = off >= 0 ? base + off : zero_page;
                if (blk < W8_HALO_BLOCKS)
                    __builtin_amdgcn_global_load_lds(W8_GPTR(g), W8_LPTR(buf + blk * 1024), 16, 0, 0);
            }
        };
        // output-tile pieces of this thread: L = k*128 + t2 (k = 0..15): pixel lp = k*16 + (t2 >> 3), cout piece (t2 & 7) ^ (lp & 7)
        // (-1: the piece belongs to an image past the end of the batch — the last tile of a batch that is not a multiple of 4)
        auto piece_off = [&](const W8Tile& t, int k) -> long {
            const int lp = k * 16 + (t2 >> 3);
            const int c8 = (t2 & 7) ^ (lp & 7);
            const int n = t.n0 + (lp >> 6);
            if (n >= p.N) return -1;
            return (long)((((size_t)n * 8 + ((lp >> 3) & 7)) * 8 + (lp & 7)) * p.Cout + t.cot * 64 + c8 * 8);
        };
        auto fetch_table = [&](const W8Tile& t) {            // bias[64] and the four images' temb rows: 4-byte DMA, 256 B per instruction
            if (bw != 0) return;
            if (p.bias) __builtin_amdgcn_global_load_lds(W8_GPTR(p.bias + t.cot * 64 + lane), W8_LPTR(tb), 4, 0, 0);
            if (p.addvec) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const int n = t.n0 + sub < p.N ? t.n0 + sub : p.N - 1;      // images past the batch: any valid row (results discarded)
                    __builtin_amdgcn_global_load_lds(W8_GPTR(p.addvec + (size_t)n * p.addvec_ld + t.cot * 64 + lane),
                                                     W8_LPTR(tb + 64 + sub * 64), 4, 0, 0);
                }
            }
        };
        bool do_res = p.residual != nullptr;       // prologue only: from the second tile on the loaders fetch the residual tile
        auto fetch_residual = [&](const W8Tile& t, int k0, int k1) {
            if (!do_res) return;
#pragma unroll 1
            for (int k = k0; k < k1; ++k) {
                const long o = piece_off(t, k);
                const void* g = o >= 0 ? (const void*)(p.residual + o) : (const void*)zero_page;
                __builtin_amdgcn_global_load_lds(W8_GPTR(g), W8_LPTR(ro + (k * 128 + bw * 64) * 16), 16, 0, 0);
            }
        };
        // pieces k0 .. k1-1 of the tile switch: previous tile's output pieces out, this tile's residual pieces into the same
        // LDS rows (a wave refills exactly the rows it drained).  Returns the vector-memory operations it issued.
        auto tile_switch = [&](const W8Tile& pt, const W8Tile& ct, int k0, int k1) -> int {
#pragma unroll 1
            for (int k = k0; k < k1; k += 2) {
                const int n = k1 - k < 2 ? k1 - k : 2;
                bf16x8 v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (u < n) v[u] = *reinterpret_cast<const bf16x8*>(ro + ((k + u) * 128 + t2) * 16);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const long o = u < n ? piece_off(pt, k + u) : -1;
                    // predicated per lane; the store still counts once in vmcnt for the wave whenever u < n
                    if (u < n) {
                        if (o >= 0) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + o) = v[u];
                    }
                }
                fetch_residual(ct, k, k + n);
            }
            // a store whose lanes all belong to images past the batch is skipped entirely: such tiles' stores are not counted
            // (a count that is too LOW only waits for more)
            return (k1 - k0) * ((pt.n0 + 4 <= p.N ? 1 : 0) + (do_res ? 1 : 0));
        };

        W8Tile cur;
        tile_of(q, cur);
        // table rows that no DMA fills stay zero
        if (bw == 0) {
            if (!p.bias) tb[lane] = 0.f;
            if (!p.addvec) { tb[64 + lane] = 0.f; tb[128 + lane] = 0.f; tb[192 + lane] = 0.f; tb[256 + lane] = 0.f; }
        }
        halo_plan(cur);
        halo_issue(0, halo0, 0, HB2);
        fetch_table(cur);
        fetch_residual(cur, 0, 16);
        // P0 needs the halo image and the table, not the residual tile (16 younger DMAs; E1's vmcnt(0) covers them)
        if (do_res) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w8_barrier();                                           // P0
        do_res = false;
        bool have_prev = false;
        W8Tile prev = cur;
        const int kpc = (16 + nchunks - 2) / (nchunks - 1);      // tile-switch pieces per chunk (first nchunks-1 chunks)
        const int pps = (kpc + 7) >> 3;                          // ... per step
        for (;;) {
            const bool more = q + qstride < ntiles;
            W8Tile nxt = cur;
            if (more) tile_of(q + qstride, nxt);
            for (int c = 0; c < nchunks; ++c) {
                const bool wrap = c + 1 == nchunks;
                if (c == 0 && have_prev) fetch_table(cur);       // older than this chunk's halo DMAs: complete at its last barrier
                if (wrap && more) halo_plan(nxt);
                const bool do_halo = !wrap || more;
                char* const hbuf = halo0 + ((c + 1) & 1) * W8_HALO;
                const int hc = wrap ? 0 : c + 1;
                int k0 = 16, k1 = 16;
                if (have_prev && !wrap) {
                    k0 = c * kpc < 16 ? c * kpc : 16;
                    k1 = k0 + kpc < 16 ? k0 + kpc : 16;
                }
                int young = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t < 7 && do_halo) halo_issue(hc, hbuf, 2 * t, 2 * t + 2);
                    if (t < 8) {
                        const int ka = k0 + t * pps < k1 ? k0 + t * pps : k1, kb = ka + pps < k1 ? ka + pps : k1;
                        if (ka < kb) {
                            const int n = tile_switch(prev, cur, ka, kb);
                            if (t >= 6) young += n;              // issued after the last halo block of this chunk
                        }
                    }
                    if (t == 8) {
                        // halo image (and everything older) landed; at the last chunk E1 needs the whole residual tile
                        if (wrap) w8_wait_vm(0);
                        else w8_wait_vm(young);
                    }
                    if (t % 3 == 2) w8_barrier();                // end of a group step
                }
            }
            w8_barrier();                                        // E1
            w8_barrier();                                        // E2
            prev = cur;
            have_prev = true;
            if (!more) break;
            q += qstride;
            cur = nxt;
        }
        // last tile out
#pragma unroll 1
        for (int k = 0; k < 16; k += 4) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(ro + ((k + u) * 128 + t2) * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = piece_off(prev, k + u);
                if (o >= 0) *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + o) = v[u];
            }
        }
    }
}

}  // namespace

// Launches the 8x8-map wave-specialised kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv_ws8_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV_WS8") ? atoi(getenv("DXMI_CONV_WS8")) : 1;   // 0: conv_pipe_kernel for the 8x8 maps
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups == 2 || (a.mask_src && a.residual) || a.act == DXMI_ACT_SILU) return 1;   // a mask alone rides the residual path
    if (a.OH != 8 || a.OW != 8) return 1;
    if (a.ups ? (a.IH != 4 || a.IW != 4) : (a.IH != 8 || a.IW != 8)) return 1;     // ups: nearest x2 upsample of a 4x4 map in front
    if (a.Cout % 64 != 0 || (a.C0 + a.C1) % 32 != 0 || a.C0 % 32 != 0) return 1;
    const int nchunks = (a.C0 + a.C1) / 32;
    if (nchunks < 4 || nchunks % 2 != 0 || (9 * nchunks) % W8_RING != 0) return 1;
    // no batch-size condition: an image's result must not depend on the batch it rides in (the kernels differ in summation order)
    if ((long)a.N * 64 * (a.C0 > a.C1 ? a.C0 : a.C1) * 2 >= (1L << 31)) return 1;   // 32-bit byte offsets inside either input part
    if (a.gn_out && !(a.gn_flags & 2)) return 1;     // the fused GroupNorm replaces the raw output (one output tile in LDS)
    if (kernel_id) {
        *kernel_id = 400008;    // conv_ws8_kernel
        return DXMI_OK;
    }
    static const void* zero_page = nullptr;
    if (!zero_page) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(w8_zero16)) != hipSuccess || !zp) {
            dxmi_set_error("dxmi_conv2d_fwd(ws8): hipGetSymbolAddress(w8_zero16) failed");
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
    b.SUBS = 4;
    b.PT = (a.N + 3) / 4;      // the last tile may hold fewer than four images (masked)
    b.CT = a.Cout / 64;
    b.tile_px = 256;
    static const int xcd_env = getenv("DXMI_CONV_WS_XCD") ? atoi(getenv("DXMI_CONV_WS_XCD")) : 1;
    b.xcd_order = xcd_env;
    const size_t lds = 2 * W8_HALO + W8_A_RING + W8_RO + W8_TB;
    int grid = b.PT * b.CT;
    if (grid > 256) grid = 256;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (a.gn_out) {
        b.out = a.gn_out;       // the movers drain the (normalised) tile to the fused output
        hipLaunchKernelGGL(conv_ws8_kernel<true>, dim3(grid), dim3(512), lds, st, b);
    } else {
        hipLaunchKernelGGL(conv_ws8_kernel<false>, dim3(grid), dim3(512), lds, st, b);
    }
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(ws8)");
    return DXMI_OK;
}
