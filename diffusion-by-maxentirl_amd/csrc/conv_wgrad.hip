// Convolution weight gradient on MFMA for gfx950 (value-network / U-Net training path).
//
//   dW[tap][co][ci] = sum_p dY[p][co] * X[p + tap][ci]        p over all N*H*W output pixels
// is a GEMM whose K dimension is the PIXEL index, while both operands live in NHWC (channel
// contiguous).  The MFMA fragments need 8 consecutive k per lane, i.e. a transposed view of both
// tiles: gfx950's ds_read_b64_tr_b16 delivers exactly that from row-major LDS images (a 16-lane
// group reads a 4-pixel x 16-channel block and every lane receives one channel's 4 pixels; lane
// semantics verified on hardware with tools/tr_probe.hip).  No transposed copies are ever made.
//
// Workgroup = 4 waves = a 64 co x 64 ci block of dW for ALL taps (accumulators: 9 x 32x32 fp32 per
// wave), looping over a strided list of 128-pixel tiles: per tile the dY tile [128 px][64 co] and the
// X halo tile [halo px][64 ci] are staged in LDS (pitch 192 B: the 4 rows x 64 B of a tr-read half
// wave land in distinct banks), then per 16-pixel k-step ONE A fragment (2 tr reads) feeds 9 MFMAs
// whose B fragments are the 9 shifted windows of the halo image.  Split-K over pixel tiles: every
// workgroup writes its partial block with plain coalesced stores, a second kernel sums the partials
// in a fixed order (bitwise reproducible), converts to OIHW and optionally accumulates into .grad.
//
// Replaces autograd's conv weight gradient for models/modules.py:71-101,142-145 and
// models/DxMI/unet_small.py convs (reference: torch.nn.Conv2d backward).
#include "conv_common.h"
#include <stdlib.h>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

struct WgradArgs {
    const bf16* x0;      // forward input, NHWC [N,IH,IW,C0]
    const bf16* x1;      // second concat source or null
    const bf16* dy;      // output gradient, NHWC [N,OH,OW,Cout]
    float* partial;      // [S][taps][Cout][Cin] fp32
    float* bpart;        // optional [S][4][Cout] fp32: column sums of dY (bias gradient), written by the ci-block-0 workgroups
    int N, IH, IW, C0, C1, OH, OW, Cout;
    int ksize, pad, ups, stride;
    int TWl, THl, SUBS, HH, HWd;  // 128-pixel tile geometry + halo
    int PT, S;                    // pixel tiles, pixel splits
    int CIB, COB;                 // 64-wide ci / co blocks
    int xcd;                      // XCD-aware block order (round 6; DXMI_WGRAD_XCD=0: hardware order)
};

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8), each with its own L2.  The (co, ci) blocks of one
// pixel split stream the SAME dY and X tiles; with consecutive block ids they sat on different XCDs and every tile was fetched from
// HBM once per block that needs it (128 -> 128 @32x32, 256 images: 208 MB per launch against 67 MB of operands, round-5 PMC pass —
// the kernel ran at the HBM rate of that traffic, not at its MFMA rate).  Logical ids that are consecutive WITHIN an XCD put the
// blocks of a split behind one L2 (any grid size: XCD j owns G / 8 + (j < G % 8) workgroups).
__device__ __forceinline__ int wg_logical_block(int b, int G, int on) {
    if (!on) return b;
    const int x = b & 7, k = b >> 3;
    const int q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + k;
}

constexpr int WG_PITCH = 192;  // bytes per LDS pixel row (64 channels bf16 + 64 pad)


__device__ __forceinline__ bf16x8 tr_frag(const char* row_lo, const char* row_hi) {
    // two transposing reads: k (pixel) 0..3 and 4..7 of this lane's channel
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row_lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row_hi));
    bf16x8 r;
    short* rs = reinterpret_cast<short*>(&r);
#pragma unroll
    for (int e = 0; e < 4; ++e) { rs[e] = a[e]; rs[4 + e] = b[e]; }
    return r;
}

// PF: the next pixel tile's dY and X pieces are fetched into registers while this tile's MFMAs run (needs <= 8 halo
// pieces per thread, i.e. stride-1 tiles); otherwise tiles are staged synchronously.
template <int KS, bool PF>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
    constexpr int TAPS = KS * KS;
    constexpr int TP = 128;  // pixels per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ydy = smem;                       // [TP][64 co]
    char* xim = smem + TP * WG_PITCH;       // [halo px][64 ci]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;  // 32-co / 32-ci sub-block of the 64x64 block
    // block -> (split, co block, ci block)
    int b = wg_logical_block(blockIdx.x, gridDim.x, p.xcd);
    const int cib = b % p.CIB; b /= p.CIB;
    const int cob = b % p.COB; b /= p.COB;
    const int split = b;

    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
    const int HHW = p.HH * p.HWd;
    const int Cin = p.C0 + p.C1;
    const int ci0 = cib * 64, co0 = cob * 64;
    const bool first = ci0 < p.C0;
    const bf16* xsrc = first ? p.x0 : p.x1;
    const int Cs = first ? p.C0 : p.C1;
    const int cis = first ? ci0 : ci0 - p.C0;

    // tr-read lane roles: group g = lane>>4 -> channel half (g&1), k half (g>>1); lane 4q+p -> row q, cols 4p..
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int ch_off = (16 * (g & 1) + 4 * pp) * 2;  // byte offset of this lane's 4-channel chunk in a 32-ch sub-block
    const int krow = 8 * (g >> 1) + q;               // + 4 for the second read

    f32x16 acc[TAPS];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int DYP = TP * 8 / 256, XP = 8;
    bf16x8 dyr[PF ? DYP : 1], xr[PF ? XP : 1];
    const int nxp = p.SUBS * HHW * 8;   // 16-byte pieces of the X halo tile
    auto fetch = [&](int pt) {
        const int tx = pt % txn, ty = (pt / txn) % tyn;
        const int n0 = (pt / (txn * tyn)) * p.SUBS;
        const int oy0 = ty << p.THl, ox0 = tx << p.TWl;
#pragma unroll
        for (int q = 0; q < DYP; ++q) {
            const int i = tid + q * 256;
            const int pix = i >> 3, pc = i & 7;
            const int x = pix & (TW - 1), y = (pix >> p.TWl) & (TH - 1), n = n0 + (pix >> (p.TWl + p.THl));
#pragma unroll
            for (int e = 0; e < 8; ++e) dyr[q][e] = (bf16)0.f;
            if (n < p.N) dyr[q] = *reinterpret_cast<const bf16x8*>(p.dy + (((size_t)n * p.OH + oy0 + y) * p.OW + ox0 + x) * p.Cout + co0 + pc * 8);
        }
#pragma unroll
        for (int q = 0; q < XP; ++q) {
            const int i = tid + q * 256;
            const int hp = i >> 3, pc = i & 7;
            const int sub = hp / HHW, rem = hp - sub * HHW;
            const int hy = rem / p.HWd, hx = rem - hy * p.HWd;
            const int iy = oy0 * p.stride - p.pad + hy, ix = ox0 * p.stride - p.pad + hx, n = n0 + sub;
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[q][e] = (bf16)0.f;
            if (i < nxp && n < p.N && iy >= 0 && ix >= 0 && iy < (p.IH << p.ups) && ix < (p.IW << p.ups))
                xr[q] = *reinterpret_cast<const bf16x8*>(xsrc + (((size_t)n * p.IH + (iy >> p.ups)) * p.IW + (ix >> p.ups)) * Cs + cis + pc * 8);
        }
    };
    if constexpr (PF) fetch(split);

    for (int pt = split; pt < p.PT; pt += p.S) {
        const int tx = pt % txn, ty = (pt / txn) % tyn;
        const int n0 = (pt / (txn * tyn)) * p.SUBS;
        const int oy0 = ty << p.THl, ox0 = tx << p.TWl;
        __syncthreads();
        if constexpr (PF) {
#pragma unroll
            for (int q = 0; q < DYP; ++q) {
                const int i = tid + q * 256;
                *reinterpret_cast<bf16x8*>(ydy + (i >> 3) * WG_PITCH + (i & 7) * 16) = dyr[q];
            }
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const int i = tid + q * 256;
                if (i < nxp) *reinterpret_cast<bf16x8*>(xim + (i >> 3) * WG_PITCH + (i & 7) * 16) = xr[q];
            }
            __syncthreads();
            if (pt + p.S < p.PT) fetch(pt + p.S);
        } else {
        // ---- stage dY tile: 128 px x 64 co = 8 pieces / px
        for (int i = tid; i < TP * 8; i += 256) {
            const int pix = i >> 3, pc = i & 7;
            const int x = pix & (TW - 1), y = (pix >> p.TWl) & (TH - 1), n = n0 + (pix >> (p.TWl + p.THl));
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (n < p.N) v = *reinterpret_cast<const bf16x8*>(p.dy + (((size_t)n * p.OH + oy0 + y) * p.OW + ox0 + x) * p.Cout + co0 + pc * 8);
            *reinterpret_cast<bf16x8*>(ydy + pix * WG_PITCH + pc * 16) = v;
        }
        // ---- stage X halo tile: (SUBS*HH*HWd) px x 64 ci
        for (int i = tid; i < p.SUBS * HHW * 8; i += 256) {
            const int hp = i >> 3, pc = i & 7;
            const int sub = hp / HHW, rem = hp - sub * HHW;
            const int hy = rem / p.HWd, hx = rem - hy * p.HWd;
            const int iy = oy0 * p.stride - p.pad + hy, ix = ox0 * p.stride - p.pad + hx, n = n0 + sub;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (n < p.N && iy >= 0 && ix >= 0 && iy < (p.IH << p.ups) && ix < (p.IW << p.ups))
                v = *reinterpret_cast<const bf16x8*>(xsrc + (((size_t)n * p.IH + (iy >> p.ups)) * p.IW + (ix >> p.ups)) * Cs + cis + pc * 8);
            *reinterpret_cast<bf16x8*>(xim + hp * WG_PITCH + pc * 16) = v;
        }
        __syncthreads();
        }
        if (p.bpart && cib == 0) {   // bias gradient rides along: this workgroup's 64 couts x 128 staged pixels
            const int col = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int px = part * 32; px < part * 32 + 32; ++px) bsum += (float)*reinterpret_cast<const bf16*>(ydy + px * WG_PITCH + col * 2);
        }
        // ---- 8 k-steps of 16 pixels.  The 20 transposing reads of k-step kb + 1 (one A fragment, nine B windows) are issued
        // BEFORE the nine MFMAs of k-step kb (double-buffered fragments): left to itself hipcc reads a fragment one or two
        // MFMAs ahead of its use and the wave — alone on its SIMD at 288 registers — exposes an LDS latency per MFMA
        // (round-2 profile: 0.13 of the MFMA peak).
        auto read_step = [&](int kb, bf16x8& a, bf16x8 (&bw)[TAPS]) {
            const int plo = kb * 16 + krow, phi = plo + 4;
            a = tr_frag(ydy + plo * WG_PITCH + wr * 64 + ch_off, ydy + phi * WG_PITCH + wr * 64 + ch_off);
            // halo offsets of the two pixel rows this lane addresses
            const int xl = plo & (TW - 1), yl = (plo >> p.TWl) & (TH - 1), sl = plo >> (p.TWl + p.THl);
            const int xh = phi & (TW - 1), yh = (phi >> p.TWl) & (TH - 1), sh = phi >> (p.TWl + p.THl);
            const char* bl = xim + ((sl * p.HH + yl * p.stride) * p.HWd + xl * p.stride) * WG_PITCH + wc * 64 + ch_off;
            const char* bh = xim + ((sh * p.HH + yh * p.stride) * p.HWd + xh * p.stride) * WG_PITCH + wc * 64 + ch_off;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int toff = ((t / KS) * p.HWd + (t % KS)) * WG_PITCH;
                bw[t] = tr_frag(bl + toff, bh + toff);
            }
        };
        bf16x8 a0, a1, b0[TAPS], b1[TAPS];
        read_step(0, a0, b0);
#pragma unroll
        for (int kb = 0; kb < TP / 16; kb += 2) {
            read_step(kb + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 2 < TP / 16) read_step(kb + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.bpart && cib == 0) p.bpart[((size_t)split * 4 + (tid >> 6)) * p.Cout + co0 + (tid & 63)] = bsum;
    // ---- partial block: D[co][ci], lane: ci = lane&31, co = 8g' + 4h + {0..3}
    const int h = lane >> 5;
    float* pb = p.partial + (size_t)split * TAPS * p.Cout * Cin;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ci = ci0 + wc * 32 + (lane & 31);
            pb[((size_t)t * p.Cout + co) * Cin + ci] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------------
// Round 3: the same 64 co x 64 ci x all-taps block per workgroup, WAVE-SPECIALISED: waves 0-3 run the MFMAs and never touch
// global memory, waves 4-7 bring the tiles into LDS by DMA (`global_load_lds`: no staging registers, no ds_write pass) —
// a DMA costs its wave ~190 cycles of issue (tools/dma_probe.hip), so the version of this kernel whose compute waves issued
// their own ~11 DMAs per tile measured 1.5x SLOWER than the register-staged kernel above, while four dedicated loaders need
// 11 x 190 = 2 100 cycles per tile against the tile's 2 304 cycles of MFMAs.
//   * double-buffered, unpadded tile images: a pixel row is 128 B (64 channels); 16-byte slot j of row R sits at slot
//     j ^ ((R & 2) << 1): the four rows x 64 B a 32-lane group of a transposing read touches then fall into four different
//     quarter-sets of the banks for ANY four consecutive rows (a DMA writes 1 KiB linearly: no padding; source addresses are free);
//   * a 1-KiB piece = 8 pixel rows; loader l moves pieces l, l + 4, ... of the dY tile (16 pieces) and of the X halo tile
//     (<= 32): per lane the tile-relative source offset and border flags of its pieces are formed once per kernel, per tile the
//     plan is a few scalar operations;
//   * ONE barrier per tile: the loaders issue tile i + 1 right behind barrier i, wait for it (vmcnt(0)) and meet the MFMA
//     waves at barrier i + 1;
//   * MFMA waves: the transposing reads run four (k-step, tap) pairs ahead of the MFMAs through a rolling window of fragments
//     (the double-buffered k-step of the kernel above needs 368 registers; eight waves per CU leave 256).
// Scope: stride 1 (optionally nearest x2 upsampled input), tile rows of >= 8 pixels (maps >= 8x8); other shapes: the kernel above.
#define WG_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WG_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ uint4 wg_zero16 = {0u, 0u, 0u, 0u};

template <int KS>
__global__ __launch_bounds__(512) void conv_wgrad_ws_kernel(WgradArgs p, const char* __restrict__ zero_page, int xpieces, int buf_bytes) {
    constexpr int TAPS = KS * KS;
    constexpr int TP = 128;
    constexpr int DYB = TP * 128;          // dY tile bytes
    constexpr int MAXXP = 8;               // X pieces per loader
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = wg_logical_block(blockIdx.x, gridDim.x, p.xcd);
    const int cib = b % p.CIB; b /= p.CIB;
    const int cob = b % p.COB; b /= p.COB;
    const int split = b;
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
    const int ci0 = cib * 64, co0 = cob * 64;
    const int ntile = split < p.PT ? (p.PT - 1 - split) / p.S + 1 : 0;      // tiles of this split

    if (wave >= 4) {
        // ============================================================ loaders
        const int l = wave - 4;
        const int HHW = p.HH * p.HWd, HPX = p.SUBS * HHW;
        const bool first = ci0 < p.C0;
        const bf16* const xsrc = (first ? p.x0 : p.x1) + (first ? ci0 : ci0 - p.C0);
        const int Cs = first ? p.C0 : p.C1;
        const bf16* const dysrc = p.dy + co0;
        // row (lane >> 3) of a piece, LDS slot (lane & 7) holds channel piece slot ^ ((R & 2) << 1)
        const int prow = lane >> 3, pslot = lane & 7;
        int dy_rel[4], dy_sub[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int R = (l + 4 * k) * 8 + prow;                        // tile pixel
            const int x = R & (TW - 1), y = (R >> p.TWl) & (TH - 1);
            dy_sub[k] = R >> (p.TWl + p.THl);
            dy_rel[k] = (((dy_sub[k] * p.OH + y) * p.OW + x) * p.Cout + ((pslot ^ ((R & 2) << 1)) << 3)) * 2;     // bytes
        }
        int x_rel[MAXXP];
        unsigned x_flag[MAXXP];   // bits 0..3: row above / below / left / right of the tile, bit 4: beyond the halo image; bits 8..: sub-image
#pragma unroll
        for (int k = 0; k < MAXXP; ++k) {
            const int R = (l + 4 * k) * 8 + prow;                        // halo pixel
            const int sub = R / HHW, rem = R - sub * HHW;
            const int hy = rem / p.HWd, hx = rem - hy * p.HWd;
            const int dyy = hy - p.pad, dxx = hx - p.pad;                // offset from the tile origin in (upsampled) input pixels
            x_rel[k] = (((sub * p.IH + (dyy >> p.ups)) * p.IW + (dxx >> p.ups)) * Cs + ((pslot ^ ((R & 2) << 1)) << 3)) * 2;
            x_flag[k] = (unsigned)(dyy < 0) | ((unsigned)(dyy >= TH) << 1) | ((unsigned)(dxx < 0) << 2) | ((unsigned)(dxx >= TW) << 3) |
                        ((unsigned)(R >= HPX) << 4) | ((unsigned)sub << 8);
        }
        const int nxw = (xpieces - l + 3) >> 2;                          // X pieces of this loader
        auto issue_tile = [&](int pt, char* buf) {
            const int tx = pt % txn, ty = (pt / txn) % tyn;
            const int n0 = (pt / (txn * tyn)) * p.SUBS;
            const int oy0 = ty << p.THl, ox0 = tx << p.TWl;
            const char* const yb = reinterpret_cast<const char*>(dysrc + ((size_t)(n0 * p.OH + oy0) * p.OW + ox0) * p.Cout);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* src = n0 + dy_sub[k] < p.N ? yb + dy_rel[k] : zero_page;
                __builtin_amdgcn_global_load_lds(WG_GPTR(src), WG_LPTR(buf + (l + 4 * k) * 1024), 16, 0, 0);
            }
            // border bits of this tile: rows above the map / below / left / right are zero padding
            const unsigned edge = (unsigned)(oy0 == 0) | ((unsigned)(oy0 + TH == p.OH) << 1) | ((unsigned)(ox0 == 0) << 2) | ((unsigned)(ox0 + TW == p.OW) << 3) | 16u;
            const char* const xb = reinterpret_cast<const char*>(xsrc + ((size_t)(n0 * p.IH + (oy0 >> p.ups)) * p.IW + (ox0 >> p.ups)) * Cs);
#pragma unroll
            for (int k = 0; k < MAXXP; ++k) {
                if (k < nxw) {
                    const char* src = ((x_flag[k] & edge & 31u) || n0 + (int)(x_flag[k] >> 8) >= p.N) ? zero_page : xb + x_rel[k];
                    __builtin_amdgcn_global_load_lds(WG_GPTR(src), WG_LPTR(buf + DYB + (l + 4 * k) * 1024), 16, 0, 0);
                }
            }
        };
        if (ntile > 0) issue_tile(split, smem);
        for (int it = 0; it < ntile; ++it) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // B_it: tile it landed; the MFMA waves are done with tile it - 1
            if (it + 1 < ntile) issue_tile(split + (it + 1) * p.S, smem + ((it + 1) & 1) * buf_bytes);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================================================ MFMA waves: 32 co x 32 ci x all taps each
    const int wr = wave >> 1, wc = wave & 1;
    const int Cin = p.C0 + p.C1;
    // tr-read lane roles (see the kernel above)
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int ch_off = (16 * (g & 1) + 4 * pp) * 2;
    const int krow = 8 * (g >> 1) + q;
    // swizzled address of (row R, byte hc < 128 of the row): (R * 128 + hc) ^ ((R & 2) << 5)
    auto saddr = [&](int R, int hc) -> int { return ((R << 7) + hc) ^ ((R & 2) << 5); };

    f32x16 acc[TAPS];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int trow[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) trow[t] = (t / KS) * p.HWd + (t % KS);

    constexpr int NK = TP / 16, STEPS = NK * TAPS, AH = TAPS > 1 ? 4 : 2, WIN = AH + 1;
    constexpr int NA = TAPS > 1 ? 2 : 3;     // A fragments in flight: the one in use + those read ahead (1x1: every step has its own)
    for (int it = 0; it < ntile; ++it) {
        const char* const ydy = smem + (it & 1) * buf_bytes;
        const char* const xim = ydy + DYB;
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): this wave's reads of the previous tile are complete
        __builtin_amdgcn_s_barrier();                    // B_it
        if (p.bpart && cib == 0) {   // bias gradient rides along: this workgroup's 64 couts x 128 staged pixels
            const int col = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int px = part * 32; px < part * 32 + 32; ++px) bsum += (float)*reinterpret_cast<const bf16*>(ydy + saddr(px, col * 2));
        }
        // flattened (k-step, tap) sequence: fragment i = kb * TAPS + t; reads run AH fragments ahead of the MFMAs
        bf16x8 afr[NA], bwin[WIN];
        int hr0 = 0;
        auto read_frag = [&](int i) {
            const int kb = i / TAPS, t = i % TAPS;
            if (t == 0) {
                const int plo = kb * 16 + krow;
                const char* al = ydy + saddr(plo, wr * 64 + ch_off);
                afr[kb % NA] = tr_frag(al, al + 512);                    // rows plo, plo + 4: same swizzle bit
                const int xl = plo & (TW - 1), yl = (plo >> p.TWl) & (TH - 1), sl = plo >> (p.TWl + p.THl);
                hr0 = (sl * p.HH + yl) * p.HWd + xl;                     // halo row of tap (0, 0); tile rows >= 8 pixels: plo + 4 is hr0 + 4
            }
#ifdef WG_DBG_SKIPB
            if (t % 3 != 0) { bwin[i % WIN] = bwin[(i + WIN - 1) % WIN]; return; }      // timing-only ablation: one B read per kernel row
#endif
            const char* bl = xim + saddr(hr0 + trow[t], wc * 64 + ch_off);
            bwin[i % WIN] = tr_frag(bl, bl + 512);
        };
#pragma unroll
        for (int i = 0; i < AH; ++i) read_frag(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            if (i + AH < STEPS) read_frag(i + AH);
            acc[i % TAPS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[(i / TAPS) % NA], bwin[i % WIN], acc[i % TAPS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.bpart && cib == 0) p.bpart[((size_t)split * 4 + (tid >> 6)) * p.Cout + co0 + (tid & 63)] = bsum;
    const int h = lane >> 5;
    float* pb = p.partial + (size_t)split * TAPS * p.Cout * Cin;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ci = ci0 + wc * 32 + (lane & 31);
            pb[((size_t)t * p.Cout + co) * Cin + ci] = acc[t][r];
        }
}

// 1x1 weight gradient with a 128 co x 128 ci block per workgroup.  The 64 x 64 block of the kernel above moves 32 KB of dY and X per
// 128-pixel tile for ONE MFLOP (8 MFMAs per wave against 8 DMAs per loader) through a two-deep buffer: one tile in flight per
// workgroup, so every tile pays its memory latency (q|k|v 256 -> 768: 24 tiles x ~3 us = the 77 us the launch takes, 20 GB/s per CU).
// With TAPS = 1 the accumulators are small, so a workgroup takes 128 x 128 (half the operand re-reads) and the tiles are 64 pixels
// deep in a ring of FOUR 32 KB slots — three tiles (96 KB) in flight per CU: four 64-channel sub-tiles per tile (dY 0 / 1, X 0 / 1,
// each in the row layout and swizzle of the kernel above), eight loader waves with counted in-order vmcnt waits, one workgroup per
// CU; MFMA wave (wr, wc) owns cout half wr of both dY sub-tiles x ci half wc of both X sub-tiles (four 32 x 32 blocks).
// A 1x1 conv's dY and X rows share their pixel index: a tile is 64 consecutive rows of the flat [N*H*W][C] tensors.
// Scope: stride 1, no upsample, Cin % 128 == 0 (both concat parts), Cout % 128 == 0, N*H*W % 64 == 0.
__global__ __launch_bounds__(768) void conv_wgrad1x1_b128_kernel(WgradArgs p) {
    constexpr int TPX = 64;                     // pixels per tile
    constexpr int SUB = TPX * 128;              // one 64-channel sub-tile: 64 rows x 128 B
    constexpr int BUF = 4 * SUB;                // dY0 | dY1 | X0 | X1 = 32 KB
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = wg_logical_block(blockIdx.x, gridDim.x, p.xcd);
    const int cib = b % p.CIB; b /= p.CIB;
    const int cob = b % p.COB; b /= p.COB;
    const int split = b;
    const int ci0 = cib * 128, co0 = cob * 128;
    const int ntile = split < p.PT ? (p.PT - 1 - split) / p.S + 1 : 0;
    const int Cin = p.C0 + p.C1;

    if (wave >= 4) {
        // ============================================================ loaders (8 waves): 32 one-KiB pieces per tile, 4 each
        const int l = wave - 4;
        const bool first = ci0 < p.C0;
        const char* const xsrc = reinterpret_cast<const char*>((first ? p.x0 : p.x1) + (first ? ci0 : ci0 - p.C0));
        const int Cs = first ? p.C0 : p.C1;
        const char* const dysrc = reinterpret_cast<const char*>(p.dy + co0);
        const int prow = lane >> 3, pslot = lane & 7;
        // piece j = l + 8 k (k = 0..3): sub-tile j >> 3 (0, 1: dY; 2, 3: X), rows 8 (j & 7) + prow; LDS slot pslot holds channel
        // piece pslot ^ ((R & 2) << 1) of the row
        long rel[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = l + 8 * k, st = j >> 3, R = (j & 7) * 8 + prow;
            const int piece = pslot ^ ((R & 2) << 1);
            rel[k] = st < 2 ? ((long)R * p.Cout + (st & 1) * 64 + piece * 8) * 2 : ((long)R * Cs + (st & 1) * 64 + piece * 8) * 2;
        }
        auto issue_tile = [&](int it) {
            const int pt = split + it * p.S;
            char* const buf = smem + (it % RING) * BUF;
            const char* const yb = dysrc + (size_t)pt * TPX * p.Cout * 2;
            const char* const xb = xsrc + (size_t)pt * TPX * Cs * 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = l + 8 * k;
                const char* src = (j >> 3) < 2 ? yb + rel[k] : xb + rel[k];
                __builtin_amdgcn_global_load_lds(WG_GPTR(src), WG_LPTR(buf + j * 1024), 16, 0, 0);
            }
        };
        for (int it = 0; it < RING - 1 && it < ntile; ++it) issue_tile(it);
        for (int it = 0; it < ntile; ++it) {
            // in-order retirement: tile `it` has landed when at most the 4 DMAs of each younger tile in flight are outstanding
            const int younger = ntile - 1 - it < RING - 2 ? ntile - 1 - it : RING - 2;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // B_it: tile it landed; the MFMA waves are done with tile it - 1
            if (it + RING - 1 < ntile) issue_tile(it + RING - 1);        // into the slot tile it - 1 vacated
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================================================ MFMA waves: (cout half wr of dY0, dY1) x (ci half wc of X0, X1)
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int ch_off = (16 * (g & 1) + 4 * pp) * 2;
    const int krow = 8 * (g >> 1) + q;
    auto saddr = [&](int R, int hc) -> int { return ((R << 7) + hc) ^ ((R & 2) << 5); };
    f32x16 acc[2][2];
    float bsum = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    for (int it = 0; it < ntile; ++it) {
        const char* const ybuf = smem + (it % RING) * BUF;
        const char* const xbuf = ybuf + 2 * SUB;
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): this wave's reads of the previous tile are complete
        __builtin_amdgcn_s_barrier();                    // B_it
        if (p.bpart && cib == 0) {   // bias gradient rides along: 128 couts x 64 staged pixels, two row halves
            const int col = tid & 127, part = tid >> 7;
            const char* const sub = ybuf + (col >> 6) * SUB;
#pragma unroll 8
            for (int px = part * 32; px < part * 32 + 32; ++px) bsum += (float)*reinterpret_cast<const bf16*>(sub + saddr(px, (col & 63) * 2));
        }
        // k-step kb = pixels 16 kb .. + 15: two dY fragments, two X fragments, four MFMAs; reads run two k-steps ahead
        bf16x8 af[3][2], bfr[3][2];
        auto read_step = [&](int kb) {
            const int plo = kb * 16 + krow;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const char* al = ybuf + u * SUB + saddr(plo, wr * 64 + ch_off);
                af[kb % 3][u] = tr_frag(al, al + 512);                   // rows plo, plo + 4: same swizzle bit
                const char* bl = xbuf + u * SUB + saddr(plo, wc * 64 + ch_off);
                bfr[kb % 3][u] = tr_frag(bl, bl + 512);
            }
        };
        read_step(0);
        read_step(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < TPX / 16; ++kb) {
            if (kb + 2 < TPX / 16) read_step(kb + 2);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kb % 3][a], bfr[kb % 3][c], acc[a][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.bpart && cib == 0) {
        // four partial rows per split (the reduce adds S * 4 rows): rows 0, 1 = the two pixel halves, rows 2, 3 = 0
        const int col = tid & 127, part = tid >> 7;
        p.bpart[((size_t)split * 4 + part) * p.Cout + co0 + col] = bsum;
        p.bpart[((size_t)split * 4 + 2 + part) * p.Cout + co0 + col] = 0.f;
    }
    const int h = lane >> 5;
    float* pb = p.partial + (size_t)split * p.Cout * Cin;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + a * 64 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ci = ci0 + c * 64 + wc * 32 + (lane & 31);
                pb[(size_t)co * Cin + ci] = acc[a][c][r];
            }
}

// sum the S partials in a fixed order -> OIHW fp32 gradient (optionally accumulated).  Four consecutive ci per thread (16-byte
// loads of every partial); the trailing blocks of the same launch fold the bias-gradient partials (bpart [S*4][Cout]) in a
// fixed order too, so a weight gradient with bias is two launches, not three.
// Flat form (one tap per position; the small layers, where the all-taps form below would be too few workgroups).
// SL: slices of the S partials per block (a block handles 256 / SL positions): 4 for the U-Net layers (S <= 64, thousands of
// positions), 16 for layers that are one or two 64 x 64 blocks, where S = 256 partials of only 36 864 positions left 145 workgroups
// adding 64 partials per thread in sequence (round-3 train trace: 60 us per launch, 12 % of all weight-gradient time).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S, int taps, int Cout,
                                                           int Cin, int accumulate, const float* __restrict__ bpart, float* __restrict__ dbias,
                                                           int wblocks) {
    constexpr int NP = 256 / SL;
    if ((int)blockIdx.x >= wblocks) {
        // bias: 16 couts x 16 slices of the S * 4 partial rows per block; slice sums added in slice order through LDS (one
        // thread per cout walking all rows in sequence was the long pole of the whole launch: S * 4 dependent L2 latencies)
        __shared__ float bs[15][16];
        const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int co = ((int)blockIdx.x - wblocks) * 16 + cl;
        const int K = S * 4;
        float t = 0.f;
        if (co < Cout) {
            const int k0 = (K * sl) >> 4, k1 = (K * (sl + 1)) >> 4;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) t += bpart[(size_t)k * Cout + co];
        }
        if (sl > 0) bs[sl - 1][cl] = t;
        __syncthreads();
        if (sl == 0 && co < Cout) {
#pragma unroll
            for (int j = 0; j < 15; ++j) t += bs[j][cl];
            dbias[co] = accumulate ? dbias[co] + t : t;
        }
        return;
    }
    // NP positions (of four consecutive ci) x SL slices of the S partials per block: a thread adds its slice's partials in
    // order, the slice sums are added in order through LDS (one thread per position looping over all S partials ran at
    // 1.3 TB/s — too few loads in flight)
    __shared__ f32x4 sm[SL - 1][NP];
    const long total = (long)taps * Cout * Cin;
    const int px = threadIdx.x % NP, sl = threadIdx.x / NP;
    const long idx = ((long)blockIdx.x * NP + px) * 4;
    const bool live = idx < total;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int k0 = (S * sl) / SL, k1 = (S * (sl + 1)) / SL;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(partial + (size_t)k * total + idx);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
    }
    if (sl > 0) sm[sl - 1][px] = s;
    __syncthreads();
    if (sl != 0 || !live) return;
#pragma unroll
    for (int j = 0; j < SL - 1; ++j) {
        const f32x4 v = sm[j][px];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int ci = idx % Cin;
    const long r = idx / Cin;
    const int co = r % Cout;
    const int t = (int)(r / Cout);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long o = ((long)co * Cin + ci + e) * taps + t;
        dw[o] = accumulate ? dw[o] + s[e] : s[e];
    }
}

// All-taps form (layers with >= 65 536 (co, ci) pairs):
// SL: slices of the S partials per block (a block handles 256 / SL positions): 4 for the U-Net layers (S <= 64, thousands of
// positions), 16 for layers that are one or two 64 x 64 blocks, where S = 256 partials of only 36 864 positions left 145 workgroups
// adding 64 partials per thread in sequence (round-3 train trace: 60 us per launch, 12 % of all weight-gradient time).
template <int SL, int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S, int taps, int Cout,
                                                           int Cin, int accumulate, const float* __restrict__ bpart, float* __restrict__ dbias,
                                                           int wblocks) {
    constexpr int NP = 256 / SL;
    if ((int)blockIdx.x >= wblocks) {
        // bias: 16 couts x 16 slices of the S * 4 partial rows per block; slice sums added in slice order through LDS (one
        // thread per cout walking all rows in sequence was the long pole of the whole launch: S * 4 dependent L2 latencies)
        __shared__ float bs[15][16];
        const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
        const int co = ((int)blockIdx.x - wblocks) * 16 + cl;
        const int K = S * 4;
        float t = 0.f;
        if (co < Cout) {
            const int k0 = (K * sl) >> 4, k1 = (K * (sl + 1)) >> 4;
#pragma unroll 8
            for (int k = k0; k < k1; ++k) t += bpart[(size_t)k * Cout + co];
        }
        if (sl > 0) bs[sl - 1][cl] = t;
        __syncthreads();
        if (sl == 0 && co < Cout) {
#pragma unroll
            for (int j = 0; j < 15; ++j) t += bs[j][cl];
            dbias[co] = accumulate ? dbias[co] + t : t;
        }
        return;
    }
    // NP positions x SL slices of the S partials per block.  A position is (co, four consecutive ci) for ALL taps: the partial
    // planes [tap][co][ci] are read as float4s (coalesced over ci) and the taps x 4 results are one contiguous run of the OIHW
    // gradient (144 B at 3x3), written as float4s.  (Round-3 first version: one tap per thread and four 4-byte stores 36 B apart —
    // every 64-byte sector of dW written in nine passes.)  A thread adds its slice's partials in order, the slice sums are added in
    // order through LDS.
    extern __shared__ __attribute__((aligned(16))) char red_smem[];
    f32x4* const sm = reinterpret_cast<f32x4*>(red_smem);            // [SL - 1][NP][TAPS]
    const long plane = (long)Cout * Cin;
    const int px = threadIdx.x % NP, sl = threadIdx.x / NP;
    const long idx = ((long)blockIdx.x * NP + px) * 4;             // (co * Cin + ci), ci % 4 == 0
    const bool live = idx < plane;
    f32x4 s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int k0 = (S * sl) / SL, k1 = (S * (sl + 1)) / SL;
        const float* src = partial + idx;
#pragma unroll 2
        for (int k = k0; k < k1; ++k) {
            f32x4 v[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) v[t] = *reinterpret_cast<const f32x4*>(src + ((size_t)k * TAPS + t) * plane);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) { s[t][0] += v[t][0]; s[t][1] += v[t][1]; s[t][2] += v[t][2]; s[t][3] += v[t][3]; }
        }
    }
    if (sl > 0) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) sm[((sl - 1) * NP + px) * TAPS + t] = s[t];
    }
    __syncthreads();
    if (sl != 0 || !live) return;
#pragma unroll
    for (int j = 0; j < SL - 1; ++j)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const f32x4 v = sm[(j * NP + px) * TAPS + t];
            s[t][0] += v[0]; s[t][1] += v[1]; s[t][2] += v[2]; s[t][3] += v[3];
        }
    // OIHW: dw[(co * Cin + ci + e) * TAPS + t], e = 0..3: 4 TAPS contiguous floats
    float o[4 * TAPS];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) o[e * TAPS + t] = s[t][e];
    f32x4* dst = reinterpret_cast<f32x4*>(dw + idx * TAPS);
#pragma unroll
    for (int i = 0; i < TAPS; ++i) {
        f32x4 v = f32x4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
        if (accumulate) {
            const f32x4 old = dst[i];
            v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
        }
        dst[i] = v;
    }
}

// column sums of a [P][C] bf16 matrix -> fp32 [C] (bias gradient); fixed-order two-level reduction
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ x, float* __restrict__ part, long P,
                                                            int Cfull, int rows_per_block) {
    // thread t handles 8-channel piece (t % (C/8)); rows strided by 256/(C/8).  gridDim.y column blocks of
    // C = Cfull / gridDim.y channels each (rows keep the full pitch) serve Cfull > 2048.
    const int C = Cfull / gridDim.y;
    x += blockIdx.y * C;
    part += blockIdx.y * C;
    const int c8n = C / 8;
    const int pc = threadIdx.x % c8n, r0 = threadIdx.x / c8n, rstep = blockDim.x / c8n;  // blockDim.x = rstep * c8n
    const long base = (long)blockIdx.x * rows_per_block;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // four rows of loads in flight per trip (one row per trip was one memory round trip per 16 bytes: 22-75 us per launch)
    const long rend = base + rows_per_block < P ? base + rows_per_block : P;
    for (long r = base + r0; r < rend; r += 4 * rstep) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r + u * rstep < rend) v[u] = *reinterpret_cast<const bf16x8*>(x + (r + u * rstep) * Cfull + pc * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r + u * rstep < rend) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[u][e];
            }
    }
    __shared__ float red[256][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < c8n) {
        float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < rstep * c8n; k += c8n)
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += red[k][e];
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(size_t)blockIdx.x * Cfull + threadIdx.x * 8 + e] = t[e];
    }
}
// 16 channels per workgroup, 16 row-lanes per channel: every lane adds its strided share of the partial rows, then one
// lane per channel adds the 16 lane sums in lane order (fixed order -> bitwise reproducible).
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblocks, int C, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < C)
        for (int k = rl; k < nblocks; k += 16) s += part[(size_t)k * C + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// out[b][c] = sum_n in[b][n][c] (fp32): 16 channels x 16 row slices per workgroup, slice sums added in slice order through LDS
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C) {
    __shared__ float bs[15][16];
    const int cblocks = (C + 15) / 16;
    const int b = blockIdx.x / cblocks, cb = blockIdx.x - b * cblocks;
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = cb * 16 + cl;
    float t = 0.f;
    if (c < C) {
        const float* src = in + (size_t)b * N * C + c;
        const int k0 = (N * sl) >> 4, k1 = (N * (sl + 1)) >> 4;
#pragma unroll 8
        for (int k = k0; k < k1; ++k) t += src[(size_t)k * C];
    }
    if (sl > 0) bs[sl - 1][cl] = t;
    __syncthreads();
    if (sl == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < 15; ++j) t += bs[j][cl];
        out[(size_t)b * C + c] = t;
    }
}

int ilog2w(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

// Upper bound of the split count any launch below may choose: the workspace is sized with it and every launch checks its S
// against it (round-3 ADVICE: the 128 x 128 1x1 path cuts 64-pixel tiles, the sizing assumed 128-pixel tiles).
static long wgrad_split_bound(long npix, int Cin, int Cout, int ksize) {
    const long PT = ksize == 1 ? (npix + 63) / 64 + 16 : (npix + 127) / 128 + 16;   // 1x1: 64-pixel tiles of the 128 x 128 kernel
    long S = (ksize == 1 ? 1024 : 512) / ((long)(Cin / 64) * (Cout / 64));          // 1x1: 128 x 128 blocks, 256 workgroups; 64 x 64: 512
    if (S < 1) S = 1;
    if (S > PT) S = PT;
    return S;
}

extern "C" int64_t dxmi_conv2d_wgrad_workspace_bytes(int32_t N, int32_t OH, int32_t OW, int32_t Cin, int32_t Cout, int32_t ksize) {
    if (N <= 0 || OH <= 0 || OW <= 0 || Cin < 64 || Cout < 64 || (ksize != 1 && ksize != 3)) return 0;    // dxmi_conv2d_wgrad rejects these
    const long S = wgrad_split_bound((long)N * OH * OW, Cin, Cout, ksize);
    return S * ksize * ksize * (int64_t)Cout * Cin * 4 + S * 4 * (int64_t)Cout * 4;   // + bias-gradient partials
}

// the fixed-order reduce of the S split partials (+ the bias partials) behind every weight-gradient kernel
static int wgrad_reduce_launch(const WgradArgs& a, void* workspace, float* dw_oihw, float* dbias, int S, int ksize, int Cout, int Cin,
                               int accumulate, hipStream_t st) {
    const long total = (long)ksize * ksize * Cout * Cin;
    const bool wide = S >= 64;                 // 16 slices of the partials per block (layers of one or two 64 x 64 blocks)
    const int np = wide ? 16 : 64;
    const int taps = ksize * ksize;
    const int bblocks = dbias ? (Cout + 15) / 16 : 0;
    if ((long)Cout * Cin >= 65536) {
        // all taps of a (co, 4 ci) position per thread: contiguous OIHW runs out (>= 256 workgroups from 256 x 256 channels up)
        const long positions = (long)Cout * Cin / 4;
        const int wblocks = (int)((positions + np - 1) / np);
        const size_t rlds = (size_t)((wide ? 15 : 3) * np * taps) * 16;
#define DXMI_WG_REDUCE(SL_, TAPS_)                                                                                                         \
    hipLaunchKernelGGL((wgrad_reduce_kernel<SL_, TAPS_>), dim3((unsigned)(wblocks + bblocks)), dim3(256), rlds, st, (const float*)workspace, \
                       dw_oihw, S, taps, Cout, Cin, accumulate, (const float*)a.bpart, dbias, wblocks)
        if (wide) {
            if (taps == 9) DXMI_WG_REDUCE(16, 9);
            else DXMI_WG_REDUCE(16, 1);
        } else {
            if (taps == 9) DXMI_WG_REDUCE(4, 9);
            else DXMI_WG_REDUCE(4, 1);
        }
#undef DXMI_WG_REDUCE
    } else {
        const long total = (long)taps * Cout * Cin;
        const int wblocks = (int)((total / 4 + np - 1) / np);
        if (wide)
            hipLaunchKernelGGL(wgrad_reduce_flat_kernel<16>, dim3((unsigned)(wblocks + bblocks)), dim3(256), 0, st, (const float*)workspace, dw_oihw, S,
                               taps, Cout, Cin, accumulate, (const float*)a.bpart, dbias, wblocks);
        else
            hipLaunchKernelGGL(wgrad_reduce_flat_kernel<4>, dim3((unsigned)(wblocks + bblocks)), dim3(256), 0, st, (const float*)workspace, dw_oihw, S,
                               taps, Cout, Cin, accumulate, (const float*)a.bpart, dbias, wblocks);
    }
    DXMI_CHECK_LAUNCH("dxmi_conv2d_wgrad(reduce)");
    return DXMI_OK;
}

static int wgrad_impl(const void* x0, int32_t C0, const void* x1, int32_t C1, const void* dy, float* dw_oihw, float* dbias,
                      void* workspace, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t Cout,
                      int32_t ksize, int32_t stride, int32_t pad, int32_t upsample, int32_t accumulate, void* stream) {
    DXMI_CHECK_ARG(stride == 1 || stride == 2, "dxmi_conv2d_wgrad: stride %d unsupported", stride);
    DXMI_CHECK_ARG(x0 && dy && dw_oihw && workspace, "dxmi_conv2d_wgrad: null pointer");
    const int Cin = C0 + C1;
    DXMI_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C0 > 0 && C1 >= 0 && Cout > 0,
                   "dxmi_conv2d_wgrad: empty or negative shape (N %d, in %dx%dx(%d+%d), out %dx%dx%d)", N, IH, IW, C0, C1, OH, OW, Cout);
    DXMI_CHECK_ARG(ksize == 1 || ksize == 3, "dxmi_conv2d_wgrad: ksize %d unsupported", ksize);
    DXMI_CHECK_ARG(Cin % 64 == 0 && Cout % 64 == 0 && C0 % 64 == 0, "dxmi_conv2d_wgrad: Cin (%d+%d) and Cout (%d) must be multiples of 64", C0, C1, Cout);
    DXMI_CHECK_ARG((OW & (OW - 1)) == 0 && (OH & (OH - 1)) == 0 && OW >= 4 && OH >= 4, "dxmi_conv2d_wgrad: OH/OW must be powers of two >= 4");
    DXMI_CHECK_ARG(C1 == 0 || x1, "dxmi_conv2d_wgrad: C1>0 but x1 NULL");
    WgradArgs a;
    static const int xcd_env = getenv("DXMI_WGRAD_XCD") ? atoi(getenv("DXMI_WGRAD_XCD")) : 1;
    // measured (tools/wgrad_time.py, same box alternating): 1x1 256 -> 768 @16x16 57.5 -> 35.7 us, 256 -> 256 28.5 -> 20.2 us, 3x3 384 -> 128
    // @32x32 216 -> 203.5 us, 256 -> 128 142.8 -> 136.9 us, 128 -> 128 unchanged; the 8x8 / 4x4 maps lose 5-10 % (their tiles already
    // share an L2 by accident of the split count): hardware order there
    a.xcd = xcd_env && (long)OH * OW >= 256;
    a.x0 = (const bf16*)x0; a.x1 = (const bf16*)x1; a.dy = (const bf16*)dy; a.partial = (float*)workspace;
    a.N = N; a.IH = IH; a.IW = IW; a.C0 = C0; a.C1 = C1; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.ksize = ksize; a.pad = pad; a.ups = upsample ? 1 : 0; a.stride = stride;
    const int TW = OW < 32 ? OW : 32;
    int TH = 128 / TW; if (TH > OH) TH = OH;
    a.TWl = ilog2w(TW); a.THl = ilog2w(TH); a.SUBS = 128 / (TW * TH);
    a.HH = (TH - 1) * stride + ksize; a.HWd = (TW - 1) * stride + ksize;
    const int ngroups = (N + a.SUBS - 1) / a.SUBS;
    a.PT = ngroups * (OH / TH) * (OW / TW);
    static const int b128_env = getenv("DXMI_WGRAD_B128") ? atoi(getenv("DXMI_WGRAD_B128")) : 1;     // 0: 64 x 64 blocks for every 1x1 layer
    const long npix = (long)N * OH * OW;
    if (b128_env && ksize == 1 && stride == 1 && !upsample && pad == 0 && IH == OH && IW == OW && Cin % 128 == 0 && C0 % 128 == 0 &&
        Cout % 128 == 0 && npix % 64 == 0 && npix * (C0 > C1 ? C0 : C1) * 2 < (1L << 31) && npix * Cout * 2 < (1L << 31)) {
        a.CIB = Cin / 128; a.COB = Cout / 128;
        a.PT = (int)(npix / 64);               // 64-pixel tiles
        int S = 256 / (a.CIB * a.COB);              // one 768-thread workgroup per CU
        if (S < 1) S = 1;
        if (S > a.PT) S = a.PT;
        DXMI_CHECK_ARG(S <= wgrad_split_bound(npix, Cin, Cout, ksize), "dxmi_conv2d_wgrad: %d splits exceed the workspace bound", S);
        a.S = S;
        a.bpart = dbias ? reinterpret_cast<float*>(workspace) + (size_t)S * Cout * Cin : nullptr;
        hipStream_t st = (hipStream_t)stream;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad1x1_b128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
        hipLaunchKernelGGL(conv_wgrad1x1_b128_kernel, dim3(S * a.CIB * a.COB), dim3(768), (size_t)4 * 4 * 64 * 128, st, a);
        DXMI_CHECK_LAUNCH("dxmi_conv2d_wgrad(1x1 b128)");
        return wgrad_reduce_launch(a, workspace, dw_oihw, dbias, S, ksize, Cout, Cin, accumulate, st);
    }
    a.CIB = Cin / 64; a.COB = Cout / 64;
    // pixel splits: the kernel holds one workgroup per CU (368 registers per lane), so 256 workgroups fill the chip; more
    // splits only add partial-sum traffic (each split writes and the reduce re-reads taps x Cout x Cin floats)
    // (the 1x1 kernel's single accumulator block lets two workgroups share a CU: 512 there; fewer, longer splits on the 4x4
    // maps measured slower: 64 workgroups instead of 256)
    static const int wgs_env = getenv("DXMI_WGRAD_WGS") ? atoi(getenv("DXMI_WGRAD_WGS")) : 0;      // tuning override
    const int wgs = wgs_env > 0 ? wgs_env : (ksize == 3 ? 256 : 512);
    int S = wgs / (a.CIB * a.COB);
    if (S < 1) S = 1;
    if (S > a.PT) S = a.PT;
    DXMI_CHECK_ARG(S <= wgrad_split_bound(npix, Cin, Cout, ksize), "dxmi_conv2d_wgrad: %d splits exceed the workspace bound (DXMI_WGRAD_WGS?)", S);
    a.S = S;
    a.bpart = dbias ? reinterpret_cast<float*>(workspace) + (size_t)S * ksize * ksize * Cout * Cin : nullptr;
    const size_t lds = (size_t)(128 + a.SUBS * a.HH * a.HWd) * WG_PITCH;
    DXMI_CHECK_ARG(lds <= 160 * 1024, "dxmi_conv2d_wgrad: LDS %zu too large", lds);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(S * a.CIB * a.COB), block(256);
    static const int pf_env = getenv("DXMI_WGRAD_PF") ? atoi(getenv("DXMI_WGRAD_PF")) : 1;   // tuning override
    const bool pf = pf_env && (long)a.SUBS * a.HH * a.HWd * 8 <= 8 * 256;   // halo pieces per thread <= 8: prefetching kernel
#define DXMI_WG_LAUNCH(KS_, PF_)                                                                                              \
    do {                                                                                                                      \
        static bool attr = false;                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<KS_, PF_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv_wgrad_kernel<KS_, PF_>), grid, block, lds, st, a);                                            \
    } while (0)
    // wave-specialised DMA-staged kernel: stride 1, tile rows of >= 8 pixels, halo tile of <= 32 one-KiB pieces
    static const int dma_env = getenv("DXMI_WGRAD_DMA") ? atoi(getenv("DXMI_WGRAD_DMA")) : 1;     // 0: register-staged kernels only
    const int hpx = a.SUBS * a.HH * a.HWd;
    const int xpieces = (hpx + 7) / 8;
    const bool dma = dma_env && stride == 1 && TW >= 8 && xpieces <= 32 && (long)N * IH * IW * (C0 > C1 ? C0 : C1) * 2 < (1L << 31) &&
                     (long)N * OH * OW * Cout * 2 < (1L << 31);
    if (dma) {
        static const void* zero_page = nullptr;
        if (!zero_page) {
            void* zp = nullptr;
            if (hipGetSymbolAddress(&zp, HIP_SYMBOL(wg_zero16)) != hipSuccess || !zp) {
                dxmi_set_error("dxmi_conv2d_wgrad: hipGetSymbolAddress(wg_zero16) failed");
                return DXMI_EINVAL;
            }
            zero_page = zp;
        }
        const int buf_bytes = 128 * 128 + xpieces * 1024;
        const size_t lds2 = 2 * (size_t)buf_bytes;
#define DXMI_WG_DMA(KS_)                                                                                                       \
    do {                                                                                                                      \
        static bool attr = false;                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ws_kernel<KS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((conv_wgrad_ws_kernel<KS_>), grid, dim3(512), lds2, st, a, (const char*)zero_page, xpieces, buf_bytes);    \
    } while (0)
        if (ksize == 3) DXMI_WG_DMA(3); else DXMI_WG_DMA(1);
#undef DXMI_WG_DMA
    } else if (ksize == 3) { if (pf) DXMI_WG_LAUNCH(3, true); else DXMI_WG_LAUNCH(3, false); }
    else { if (pf) DXMI_WG_LAUNCH(1, true); else DXMI_WG_LAUNCH(1, false); }
#undef DXMI_WG_LAUNCH
    DXMI_CHECK_LAUNCH("dxmi_conv2d_wgrad");
    return wgrad_reduce_launch(a, workspace, dw_oihw, dbias, S, ksize, Cout, Cin, accumulate, st);
}

extern "C" int dxmi_conv2d_wgrad(const void* x0, int32_t C0, const void* x1, int32_t C1, const void* dy, float* dw_oihw,
                                 void* workspace, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW, int32_t Cout,
                                 int32_t ksize, int32_t stride, int32_t pad, int32_t upsample, int32_t accumulate, void* stream) {
    return wgrad_impl(x0, C0, x1, C1, dy, dw_oihw, nullptr, workspace, N, IH, IW, OH, OW, Cout, ksize, stride, pad, upsample,
                      accumulate, stream);
}

// Same, plus the bias gradient dbias[co] = sum_p dY[p][co] (fp32 [Cout]) from the dY tiles the kernel stages anyway -
// replaces the separate column-sum pass over dY.
extern "C" int dxmi_conv2d_wgrad_bias(const void* x0, int32_t C0, const void* x1, int32_t C1, const void* dy, float* dw_oihw,
                                      float* dbias, void* workspace, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW,
                                      int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t upsample,
                                      int32_t accumulate, void* stream) {
    DXMI_CHECK_ARG(dbias, "dxmi_conv2d_wgrad_bias: dbias is NULL");
    return wgrad_impl(x0, C0, x1, C1, dy, dw_oihw, dbias, workspace, N, IH, IW, OH, OW, Cout, ksize, stride, pad, upsample,
                      accumulate, stream);
}

extern "C" int dxmi_colsum_bf16(const void* x, float* out, void* workspace, int64_t P, int32_t C, int32_t accumulate,
                                void* stream) {
    DXMI_CHECK_ARG(x && out && workspace, "dxmi_colsum_bf16: null pointer");
    int cblocks = 1;
    while (C % cblocks != 0 || (C / cblocks) % 8 != 0 || C / cblocks > 2048) {
        ++cblocks;
        DXMI_CHECK_ARG(cblocks <= 64, "dxmi_colsum_bf16: C=%d unsupported", C);
    }
    const int cthreads = (256 / (C / cblocks / 8)) * (C / cblocks / 8);
    int rows_per_block = 512;
    while ((P + rows_per_block - 1) / rows_per_block > 256) rows_per_block *= 2;  // <= 256 partial rows
    const int nblocks = (int)((P + rows_per_block - 1) / rows_per_block);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblocks, cblocks), dim3(cthreads), 0, st, (const bf16*)x, (float*)workspace, (long)P, C,
                       rows_per_block);
    DXMI_CHECK_LAUNCH("dxmi_colsum_bf16");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)workspace, out, nblocks, C,
                       accumulate);
    DXMI_CHECK_LAUNCH("dxmi_colsum_bf16(final)");
    return DXMI_OK;
}

// out[b][c] = sum_n in[b][n][c], fp32 (the per-image d(gamma) / d(beta) partials of the GroupNorm backward): fixed order.
extern "C" int dxmi_colsum_f32(const float* in, float* out, int32_t B, int32_t N, int32_t C, void* stream) {
    DXMI_CHECK_ARG(in && out && B > 0 && N > 0 && C > 0, "dxmi_colsum_f32: bad arguments");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(B * ((C + 15) / 16)), dim3(256), 0, (hipStream_t)stream, in, out, N, C);
    DXMI_CHECK_LAUNCH("dxmi_colsum_f32");
    return DXMI_OK;
}

// out[b][c] = sum over the b-th block of `rows_per_block` rows of x[P][C] (per-image sums: the gradient of
// the per-(n, channel) temb term, unet_small.py:123).
extern "C" int dxmi_colsum_blocks_bf16(const void* x, float* out, int64_t P, int32_t C, int32_t rows_per_block, void* stream) {
    DXMI_CHECK_ARG(x && out && rows_per_block > 0, "dxmi_colsum_blocks_bf16: bad arguments");
    DXMI_CHECK_ARG(C % 8 == 0 && C / 8 <= 256, "dxmi_colsum_blocks_bf16: C=%d unsupported", C);
    const int cthreads = (256 / (C / 8)) * (C / 8);
    const int nblocks = (int)((P + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblocks), dim3(cthreads), 0, (hipStream_t)stream, (const bf16*)x, out, (long)P, C,
                       rows_per_block);
    DXMI_CHECK_LAUNCH("dxmi_colsum_blocks_bf16");
    return DXMI_OK;
}
