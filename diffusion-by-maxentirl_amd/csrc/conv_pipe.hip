// Persistent, software-pipelined implicit-GEMM convolution for gfx950 (the U-Net's dominant kernel).
//
// Same GEMM mapping as conv_igemm.hip (A = weight fragments straight from global/L2 into VGPRs,
// B = input halo image in LDS, D[co][pixel] accumulators), restructured so that the MFMA pipe is
// not left waiting on memory:
//   * a workgroup (4 waves = 128 output channels x 32*NB pixels) is PERSISTENT: it walks a strided
//     list of pixel tiles for one cout tile, so the launch ramp, the first-chunk load latency and
//     the drain are paid once per workgroup instead of once per tile (measured: 42 % of a
//     128->128 @32x32 launch was per-tile prologue/epilogue);
//   * the K loop is a stream of (chunk, tap) steps that runs ACROSS tiles: every step requests its
//     successor's operands — the next tap's A fragments (global, wave-private, pre-packed in MFMA
//     order) and B fragments (ds_read_b128 from the LDS halo image) — while its own 2*NB MFMAs
//     run; two register sets ping-pong by unrolling (no copies), a sched_barrier per step keeps
//     hipcc from sinking the requests next to their uses;
//   * the next channel chunk's halo (of this tile, or chunk 0 of the NEXT tile) is loaded
//     global->registers at the first tap of a chunk and written to the other LDS image at the
//     last tap: HBM/L2 latency hides behind 8 taps of MFMAs, ONE barrier per 9*2*NB MFMAs;
//   * LDS row / sub-image pitches are chosen per tile shape so the ds_read_b128 lane groups are
//     bank-conflict free.
// NB in {8,4,2} (pixels per tile = 32*NB) is chosen on the host; 2 workgroups per CU for NB <= 4.
#include "conv_common.h"
#include <stdlib.h>

#ifdef DXMI_CONV_STAMPS
// timing-only build (make STAMPS=1): per-workgroup cycle stamps of wave 0 (s_memtime), read back with
// dxmi_debug_read_stamps() by tools/conv_stamps.py: [0] start, [1] HW_ID | XCC_ID << 32, then per tile
// (K loop start, K loop end, epilogue end)
__device__ unsigned long long g_stamps[2048][24];
#define DXMI_STAMP(i)                                                                                        \
    do {                                                                                                     \
        if (threadIdx.x == 0 && (i) < 24) g_stamps[blockIdx.x & 2047][(i)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
extern "C" int dxmi_debug_read_stamps(void* dst, int bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define DXMI_STAMP(i) do {} while (0)
#endif

namespace {

// Epilogue through LDS: the accumulators hold D[co][pixel] with only 4 consecutive couts per lane, so
// a direct store scatters 8-byte pieces over 32 pixel rows per instruction (measured: 45 % of a
// 128->128 @32x32 launch).  Instead each wave drops its fp32 tile (+bias +temb term) into a
// [64 px][128 co] fp32 LDS slab (pitch 528 B: conflict-free for the b128 writes and reads), and after a
// barrier every thread owns 16-byte output pieces: a wave then writes four full 256-byte NHWC pixel
// rows per store instruction and reads the residual the same way.  Rounding is unchanged
// (fp32 sum of conv + bias + temb + residual, activation, ONE rounding to bf16).
constexpr int EPI_PITCH = 528;
constexpr int EPI_BYTES = 64 * EPI_PITCH;

template <int NB>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& p, f32x16 (&acc)[1][NB], char* eb, int n0, int oy0,
                                                  int ox0, int cot, int wave, int lane, int tid) {
    constexpr int EPB = 2;  // 32-pixel blocks per pass (64 pixels)
    constexpr int NP = NB / EPB;
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int h = lane >> 5;
    const int co_l = wave * 32 + 4 * h;  // + 8g
    const int pc = tid & 15;
    const bool pc_ok = cot * 128 + pc * 8 < p.Cout;
    const float slope = dxmi_act_slope(p.act);      // uniform: one select, not a branch ladder per value
    const bool silu = p.act == DXMI_ACT_SILU;
    // output offset of this thread's k-th 16-byte piece of a pass (-1: outside the batch / cout range)
    auto out_off = [&](int pass, int k) -> long {
        const int lp = (tid >> 4) + 16 * k;
        const int pix = pass * 64 + lp;
        const int x = pix & (TW - 1);
        const int y = (pix >> p.TWl) & (TH - 1);
        const int n = n0 + (pix >> (p.TWl + p.THl));
        return (n < p.N && pc_ok) ? (long)((((size_t)n * p.OH + oy0 + y) * p.OW + ox0 + x) * p.Cout + cot * 128 + pc * 8) : -1;
    };
    // the residual of a pass is requested one phase early (before the slab is written / during the previous pass's
    // stores), so its latency overlaps the LDS transpose instead of following it
    bf16x8 rv[4];
    auto res_load = [&](int pass) {
        if (p.residual) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long o = out_off(pass, k);
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[k][e] = (bf16)0.f;
                if (o >= 0) rv[k] = *reinterpret_cast<const bf16x8*>(p.residual + o);
            }
        }
    };
    res_load(0);
    // optional GroupNorm block statistics of the stored tile (p.gn_stats: one image per tile, full cout tiles — host-checked):
    // every thread owns ONE 8-channel piece over its pixels of all passes
    float gst[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
#pragma unroll
        for (int e2 = 0; e2 < EPB; ++e2) {
            const int nb = pass * EPB + e2;
            const int pix = nb * 32 + (lane & 31);
            const int n = n0 + (pix >> (p.TWl + p.THl));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[0][nb][4 * g + e];
                const bool co_ok = cot * 128 + co_l + 8 * g < p.Cout;
                if (p.bias && co_ok) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cot * 128 + co_l + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                if (p.addvec && n < p.N && co_ok) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(p.addvec + (size_t)n * p.addvec_ld + cot * 128 + co_l + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += t[e];
                }
                *reinterpret_cast<f32x4*>(eb + (e2 * 32 + (lane & 31)) * EPI_PITCH + (co_l + 8 * g) * 4) = v;
            }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lp = (tid >> 4) + 16 * k;
            const long o = out_off(pass, k);
            if (o >= 0) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(eb + lp * EPI_PITCH + pc * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(eb + lp * EPI_PITCH + pc * 32 + 16);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                if (p.residual) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rv[k][e];
                }
                if (p.mask_src) {
                    const bf16x8 mv = *reinterpret_cast<const bf16x8*>(p.mask_src + o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= ((float)mv[e] > 0.f ? 1.f : p.mask_slope);
                }
                bf16x8 ov;
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (bf16)(v[e] / (1.f + __expf(-v[e])));
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov[e] = (bf16)dxmi_act_lin(v[e], slope);
                }
#if defined(STEM_DBG) && (STEM_DBG & 2)
                if (ov[0] == (bf16)12345.f)          // timing-only ablation: (almost) no output stores
#endif
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + o) = ov;
                if (p.gn_stats) {
                    const bf16x4 lo4 = {ov[0], ov[1], ov[2], ov[3]}, hi4 = {ov[4], ov[5], ov[6], ov[7]};
                    dxmi_stats4(lo4, gst[0]);
                    dxmi_stats4(hi4, gst[1]);
                }
            }
        }
        if (pass + 1 < NP) res_load(pass + 1);
        lds_barrier();
    }
    if (p.gn_stats) {
        // the 16 pixel lanes of a piece are added in lane order through the (free) slab: one partial per tile
        float* const sl = reinterpret_cast<float*>(eb) + ((tid >> 4) * 16 + pc) * 8;
        *reinterpret_cast<f32x4*>(sl) = f32x4{gst[0][0], gst[0][1], gst[0][2], gst[0][3]};
        *reinterpret_cast<f32x4*>(sl + 4) = f32x4{gst[1][0], gst[1][1], gst[1][2], gst[1][3]};
        lds_barrier();
        if (tid < 128) {
            const int j = tid & 7, c8 = tid >> 3;          // float j of piece c8
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += reinterpret_cast<const float*>(eb)[(r * 16 + c8) * 8 + j];
            const int part = (oy0 >> p.THl) * (p.OW >> p.TWl) + (ox0 >> p.TWl);
            const int P = (p.OH >> p.THl) * (p.OW >> p.TWl);
            p.gn_stats[(((size_t)n0 * P + part) * (p.Cout >> 1) + cot * 64) * 2 + tid] = t;
        }
        lds_barrier();
    }
}

// Lean epilogue: bias + linear activation only (no residual / per-image vector / activation mask / SiLU — conv_lean_ok()).
// conv_epilogue_lds carries every feature of the conv family as run-time branches (~3 000 VALU instructions: with loads AND stores
// ablated the 8-MFMA stem launch still took 28 of its 33 us, and a 64-pixel stride-2 tile spends a third of its time there).  Here:
// bias + activation in the accumulator layout, ONE rounding, bf16 into a [NB * 32 px][128 co] LDS tile (16-byte slot c8 of pixel
// row R at c8 ^ (R & 15)), then every thread moves 16-byte pieces of whole 256-byte NHWC rows; optional GroupNorm block statistics
// as in conv_epilogue_lds (one image per tile, full cout tiles: host-checked).  Same values as conv_epilogue_lds.
__device__ __forceinline__ bool conv_lean_ok(const ConvArgs& p) {
    return !p.residual && !p.addvec && !p.mask_src && p.act != DXMI_ACT_SILU;
}

template <int NB>
__device__ __forceinline__ void conv_epilogue_lean(const ConvArgs& p, f32x16 (&acc)[1][NB], char* tile, int n0, int oy0, int ox0,
                                                   int cot, int wave, int lane, int tid) {
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int h = lane >> 5, pxl = lane & 31;
    const float slope = dxmi_act_slope(p.act);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && cot * 128 + wave * 32 + 8 * g < p.Cout) bv = *reinterpret_cast<const f32x4*>(p.bias + cot * 128 + wave * 32 + 8 * g + 4 * h);
        const int c8 = wave * 4 + g;                               // 16-byte slot of couts wave * 32 + 8 g .. + 7; this lane's half: + 8 h bytes
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int R = nb * 32 + pxl;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)dxmi_act_lin(acc[0][nb][4 * g + e] + bv[e], slope);
            *reinterpret_cast<bf16x4*>(tile + R * 256 + ((c8 ^ (R & 15)) << 4) + 8 * h) = o;
        }
    }
    lds_barrier();
    const int pc = tid & 15, pr = tid >> 4;
    const bool pc_ok = cot * 128 + pc * 8 < p.Cout;
    float gst[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 2 * NB; ++k) {
        const int R = pr + 16 * k;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + R * 256 + ((pc ^ (R & 15)) << 4));
        const int x = R & (TW - 1), y = (R >> p.TWl) & (TH - 1), n = n0 + (R >> (p.TWl + p.THl));
        if (n < p.N && pc_ok) {
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16*>(p.out) + (((size_t)n * p.OH + oy0 + y) * p.OW + ox0 + x) * p.Cout + cot * 128 + pc * 8) = v;
            if (p.gn_stats) {
                const bf16x4 lo4 = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
                dxmi_stats4(lo4, gst[0]);
                dxmi_stats4(hi4, gst[1]);
            }
        }
    }
    if (p.gn_stats) {
        // the 16 pixel lanes of a piece added in lane order: one partial per tile
        lds_barrier();
        float* const sl = reinterpret_cast<float*>(tile) + (pr * 16 + pc) * 8;
        *reinterpret_cast<f32x4*>(sl) = f32x4{gst[0][0], gst[0][1], gst[0][2], gst[0][3]};
        *reinterpret_cast<f32x4*>(sl + 4) = f32x4{gst[1][0], gst[1][1], gst[1][2], gst[1][3]};
        lds_barrier();
        if (tid < 128) {
            const int j = tid & 7, c8s = tid >> 3;
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += reinterpret_cast<const float*>(tile)[(r * 16 + c8s) * 8 + j];
            const int part = (oy0 >> p.THl) * (p.OW >> p.TWl) + (ox0 >> p.TWl);
            const int P = (p.OH >> p.THl) * (p.OW >> p.TWl);
            p.gn_stats[(((size_t)n0 * P + part) * (p.Cout >> 1) + cot * 64) * 2 + tid] = t;
        }
    }
    lds_barrier();          // persistent callers reuse the tile / the buffers behind it
}

// AQ: how many steps ahead the weight fragments are requested (queue of AQ+1 fragment pairs).  Small-map layers are
// bound by the L2 latency of those requests (4 MFMAs per step at NB=2), so they run a deeper queue.
// SD: chunks of staging distance (1: the next chunk is requested at this chunk's first tap; 2: the chunk after next,
// through two register sets - the halo fetch then has 17 steps instead of 8 to come back from HBM)
template <int NB, int PMAX, int KS, int DBG = 0, int AQ = 1, int SD = 1>  // DBG: timing-only ablations (wrong results)
__global__ __launch_bounds__(256, (NB <= 4 ? 2 : 1)) void conv_pipe_kernel(ConvArgs p) {
    constexpr int CK = 32, ROWB = CK * 2 + 16, PPP = CK / 8, TAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent schedule: this workgroup owns cout tile `cot` and pixel tiles pt0, pt0+nstreams, ...
    // XCD-aware: workgroups b and b+8 share an XCD (round-robin dispatch).  The CT cout tiles of one pixel-tile
    // stream are placed on ONE XCD, adjacent in dispatch order, so the input halo they all stage is fetched from
    // HBM / Infinity Cache once and served to the others by that XCD's L2.
    const int nstreams = gridDim.x / p.CT;
    int cot, pt;
    if ((nstreams & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cot = j % p.CT;
        pt = (j / p.CT) * 8 + xcd;
    } else {
        cot = blockIdx.x % p.CT;
        pt = blockIdx.x / p.CT;
    }
    if (pt >= p.PT) return;

    // the two workgroups that share a CU start in lockstep and would stay in lockstep (same tiles, same
    // cost): both in their MFMA loops, then both in their memory-bound tile switch.  Delaying the second
    // resident half once lets one's epilogue overlap the other's MFMAs for the rest of the launch.
    if (p.stagger && blockIdx.x >= gridDim.x / 2)
        for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);

    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
    const int nchunks = (p.C0 + p.C1) / CK;  // even (host-checked)
    const int HHW = p.HH * p.HWd;
    const int npieces = p.SUBS * HHW * PPP;
    const int BUF = p.lds_buf;

    int hoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pix = nb * 32 + (lane & 31);
        const int x = pix & (TW - 1);
        const int y = (pix >> p.TWl) & (TH - 1);
        const int sub = pix >> (p.TWl + p.THl);
        hoff[nb] = sub * p.SP + (y * p.RP + x * ROWB) * p.stride + (lane >> 5) * 16;   // stride 1 or 2
    }

    // per-piece halo geometry (tile independent): packed (sub, hy, hx) and the LDS byte offset
    int geo[PMAX];
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const int i = tid + q * 256;
        geo[q] = -1;
        if (i < npieces) {
            const int hp = i / PPP;
            const int sub = hp / HHW;
            const int rem = hp - sub * HHW;
            const int hy = rem / p.HWd;
            const int hx = rem - hy * p.HWd;
            geo[q] = (sub << 20) | (hy << 10) | hx;
        }
    }
    auto tile_origin = [&](int t, int& n0, int& oy0, int& ox0) {
        const int tx = t % txn;
        const int ty = (t / txn) % tyn;
        n0 = (t / (txn * tyn)) * p.SUBS;
        oy0 = ty << p.THl;
        ox0 = tx << p.TWl;
    };
    // source pixel of every staging piece for tile t (-1 = zero padding)
    auto tile_srcpix = [&](int t, int (&sp)[PMAX]) {
        int n0, oy0, ox0;
        tile_origin(t, n0, oy0, ox0);
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int gq = geo[q];
            const int n = n0 + (gq >> 20);
            const int iy = oy0 * p.stride - p.pad + ((gq >> 10) & 1023);
            const int ix = ox0 * p.stride - p.pad + (gq & 1023);
            const int sh = p.ups ? 1 : 0;  // ups 1: nearest x2 ; ups 2: zero-stuffed x2 (stride-2 data gradient)
            const bool ok = gq >= 0 && iy >= 0 && ix >= 0 && iy < (p.IH << sh) && ix < (p.IW << sh) && n < p.N &&
                            (p.ups != 2 || (((iy | ix) & 1) == 0));
            sp[q] = ok ? (n * p.IH + (iy >> sh)) * p.IW + (ix >> sh) : -1;
        }
    };

    bf16x8 stgs[SD][PMAX];
    auto stage_load_set = [&](int c, const int (&sp)[PMAX], bf16x8 (&stg)[PMAX]) {
        const int cbase = c * CK;
        const bool first = cbase < p.C0;
        const bf16* src = first ? p.in0 : p.in1;
        const int Cs = first ? p.C0 : p.C1;
        const int coff = (first ? cbase : cbase - p.C0) + (tid % PPP) * 8;  // 256 % PPP == 0: piece column fixed per thread
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) stg[q][e] = (bf16)0.f;
            if (sp[q] >= 0) stg[q] = *reinterpret_cast<const bf16x8*>(src + (size_t)sp[q] * Cs + coff);
        }
    };
    auto stage_store_set = [&](int buf, const bf16x8 (&stg)[PMAX]) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int gq = geo[q];
            const int off = (gq >> 20) * p.SP + ((gq >> 10) & 1023) * p.RP + (gq & 1023) * ROWB + (tid % PPP) * 16;
            if (gq >= 0) *reinterpret_cast<bf16x8*>(smem + buf + off) = stg[q];
        }
    };
    auto stage_load = [&](int c, const int (&sp)[PMAX]) { stage_load_set(c, sp, stgs[0]); };
    auto stage_store = [&](int buf) { stage_store_set(buf, stgs[0]); };

    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;

    const int cb0 = cot * 4 + wave;  // this wave's 32-co block
    // Cout % 128 == 64: the last cout tile is half empty; its idle waves read a valid block and store nothing
    const int cbw = cb0 < p.CB ? cb0 : p.CB - 1;
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(p.w) + (size_t)cbw * 64 + lane;
    const int wstep = p.CB * 64;  // fragments between consecutive k-steps

    // register sets, ping-pong by the parity of the unrolled step index
    bf16x8 Aq[AQ + 1][2], B[2][2][NB];
    int srcA[PMAX];  // staging sources of the tile being loaded

    // ---- prologue: chunk 0 of the first tile into LDS image 0, operands of its first step into set 0
    tile_srcpix(pt, srcA);
    stage_load(0, srcA);
    stage_store(0);
    if constexpr (SD == 2) stage_load_set(1, srcA, stgs[1]);   // nchunks >= 2
    const int clast = nchunks - 1;
    // weight fragments of step f of the (chunk, tap) stream that starts at chunk c (wraps into the next tile)
    auto load_a = [&](int c, int f, bf16x8 (&dst)[2]) {
        int cc = c + f / TAPS;
        cc = cc > clast ? cc - nchunks : cc;
        cc = cc > clast ? cc - nchunks : cc;
        const bf16x8* w0 = wfrag + (size_t)((f % TAPS) * p.KST + cc * 2) * wstep;
        dst[0] = w0[0];
        dst[1] = w0[wstep];
    };
#pragma unroll
    for (int f = 0; f < AQ; ++f) load_a(0, f, Aq[f]);
    lds_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) B[0][ks][nb] = *reinterpret_cast<const bf16x8*>(smem + hoff[nb] + ks * 32);

    [[maybe_unused]] int stamp_i = 2;
#ifdef DXMI_CONV_STAMPS
    if (threadIdx.x == 0) {
        g_stamps[blockIdx.x & 2047][0] = __builtin_amdgcn_s_memtime();
        g_stamps[blockIdx.x & 2047][1] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (8 << 6) | 4) |
                                         ((unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 32);
    }
#endif
    for (;;) {
        DXMI_STAMP(stamp_i); ++stamp_i;
        const int pt_next = pt + nstreams;
        const bool more_tiles = pt_next < p.PT;
        // ---- K loop of this tile: two chunks (2*TAPS steps) of straight-line code per iteration.
        // Everything a step requests for its successor is unconditional, so the step body has no
        // branches and hipcc can count its waits exactly.
        for (int c = 0; c < ((DBG & 16) ? 0 : nchunks); c += 2) {
            const bool last_pair = c + 2 >= nchunks;
#pragma unroll
            for (int u = 0; u < 2 * TAPS; ++u) {
                const int half = u / TAPS;          // 0: chunk c (LDS image 0), 1: chunk c+1 (LDS image 1)
                const int tap = u % TAPS;
                const int u2 = u + 1;
                const int half2 = (u2 / TAPS) & 1;  // LDS image of the successor step
                const int tap2 = u2 % TAPS;
                int cc2 = c + u2 / TAPS;            // chunk of the successor step (wraps into the next tile)
                cc2 = cc2 > clast ? 0 : cc2;
                const int set = u & 1;
                if (SD == 2 && tap == 0) {
                    // chunk (c + half + 2): of this tile, or - wrapped - of the next one (sources recomputed on the fly)
                    int ct = c + half + 2;
                    const bool wrap = ct > clast;
                    ct = wrap ? ct - nchunks : ct;
                    int srcT[PMAX];
                    tile_srcpix(wrap ? (more_tiles ? pt_next : pt) : pt, srcT);
                    stage_load_set(ct, srcT, stgs[half]);
                }
                if (SD == 1 && tap == 0 && !(DBG & 2)) {
                    if (half == 0) stage_load(c + 1, srcA);
                    else {
                        // the chunk after this pair: c+2 of this tile, or (last pair) chunk 0 of the next
                        // tile — its sources replace srcA by select, not by branch (on the very last tile
                        // this re-loads chunk 0 of the same tile, harmlessly)
                        int srcN[PMAX];
                        tile_srcpix(more_tiles ? pt_next : pt, srcN);
#pragma unroll
                        for (int q = 0; q < PMAX; ++q) srcA[q] = last_pair ? srcN[q] : srcA[q];
                        stage_load(last_pair ? 0 : c + 2, srcA);
                    }
                }
                if (!(DBG & 1)) load_a(c, u + AQ, Aq[AQ]);
                else { Aq[AQ][0] = Aq[0][0]; Aq[AQ][1] = Aq[0][1]; }
                if (tap == TAPS - 1) {
                    if (SD == 2) stage_store_set(half ? 0 : BUF, stgs[half ^ 1]);
                    else if (!(DBG & 2)) stage_store(half ? 0 : BUF);
                    if (!(DBG & 4)) lds_barrier();
                }
                const char* nbase = smem + (half2 ? BUF : 0) + (tap2 / KS) * p.RP + (tap2 % KS) * ROWB;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if (!(DBG & 8)) B[set ^ 1][ks][nb] = *reinterpret_cast<const bf16x8*>(nbase + hoff[nb] + ks * 32);
                        else B[set ^ 1][ks][nb] = B[set][ks][nb];
                        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[0][ks], B[set][ks][nb], acc[0][nb], 0, 0, 0);
                    }
#pragma unroll
                for (int f = 0; f < AQ; ++f) { Aq[f][0] = Aq[f + 1][0]; Aq[f][1] = Aq[f + 1][1]; }
                // keep the successor's operand requests INSIDE this step: without the fence the
                // machine scheduler sinks them next to their first use and the prefetch distance is lost
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- tile done: the next tile's first operands are already in flight
        DXMI_STAMP(stamp_i); ++stamp_i;
        {
            int n0, oy0, ox0;
            tile_origin(pt, n0, oy0, ox0);
            if (!(DBG & 32)) {
                // (the lean tile is NB * 8 KB: it fits the 33 KB epilogue area up to 128-pixel tiles)
                if (NB <= 4 && conv_lean_ok(p)) conv_epilogue_lean<(NB <= 4 ? NB : 4)>(p, reinterpret_cast<f32x16(&)[1][(NB <= 4 ? NB : 4)]>(acc), smem + 2 * BUF, n0, oy0, ox0, cot, wave, lane, tid);
                else conv_epilogue_lds<NB>(p, acc, smem + 2 * BUF, n0, oy0, ox0, cot, wave, lane, tid);
            }
            else {
                // timing-only ablation: keep EVERY accumulator live (a use of one element lets hipcc delete the MFMAs that
                // feed the others: the "K loop alone" numbers of round 1 were taken that way and read 2x too fast)
                float keep = 0.f;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) keep += acc[0][nb][r];
                if (keep == 12345.f) reinterpret_cast<float*>(p.out)[0] = keep;
            }
        }
        DXMI_STAMP(stamp_i); ++stamp_i;
        if (!more_tiles) break;
        pt = pt_next;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem convolution (3 -> Cout, 3x3/s1/p1, NCHW fp32 image in, NHWC bf16 out): one k-chunk (27 taps padded to 32), so
// the launch is all epilogue — 128-pixel tiles, im2col straight from the image (thread = pixel: coalesced per tap),
// 8 MFMAs per wave, then the same LDS-transposed 256-byte-row stores as conv_pipe (the generic kernel's scattered
// 8-byte stores made this layer 5x slower than its output write).
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvArgs p) {
    constexpr int NB = 4, ROWB = 80;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cot = blockIdx.x % p.CT;
    const int pt = blockIdx.x / p.CT;
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
    const int tx = pt % txn, ty = (pt / txn) % tyn;
    const int n0 = (pt / (txn * tyn)) * p.SUBS, oy0 = ty << p.THl, ox0 = tx << p.TWl;
    {
        // the tile's 3-channel halo ((TH + 2) x (TW + 2) pixels per sub-image, <= 3.4 KB) is fetched ONCE with row-coalesced loads
        // (<= 4 per thread, all in flight) into LDS behind the epilogue slab; the im2col rows are gathered from there.  (Sixteen
        // bounds-checked 4-byte global loads per thread cost 12 of the launch's 36 us: stem ablation, tools/stem_time.py.)
        float* const himg = reinterpret_cast<float*>(smem + 128 * ROWB + EPI_BYTES);
        const float* xin = reinterpret_cast<const float*>(p.in0);
        const int HWd = TW + 2, HHt = TH + 2, HPs = HHt * HWd;
        const int total = p.SUBS * 3 * HPs;
        float hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {           // total <= 1024
            const int i = tid + u * 256;
            const int sub = i / (3 * HPs), r = i - sub * 3 * HPs;
            const int ci = r / HPs, r2 = r - ci * HPs;
            const int hy = r2 / HWd, hx = r2 - hy * HWd;
            const int n = n0 + sub, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            hv[u] = 0.f;
            if (i < total && n < p.N && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) hv[u] = xin[((n * 3 + ci) * p.IH + iy) * p.IW + ix];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (tid + u * 256 < total) himg[tid + u * 256] = hv[u];
        lds_barrier();
        // thread = (pixel, half of the 32 k-values); k = ci*9 + ky*3 + kx
        const int pix = tid & 127, half = tid >> 7;
        const int x = pix & (TW - 1);
        const int y = (pix >> p.TWl) & (TH - 1);
        const int sub = pix >> (p.TWl + p.THl);
        const float* const hb = himg + sub * 3 * HPs + y * HWd + x;
        bf16x8 v[2];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = half * 16 + j;
            const int ci = k / 9, r = k - ci * 9, ky = r / 3, kx = r - ky * 3;
            const float f = k < 27 ? hb[(ci * HHt + ky) * HWd + kx] : 0.f;
            v[j >> 3][j & 7] = (bf16)f;
        }
        *reinterpret_cast<bf16x8*>(smem + pix * ROWB + half * 32) = v[0];
        *reinterpret_cast<bf16x8*>(smem + pix * ROWB + half * 32 + 16) = v[1];
    }
    const int cb0 = cot * 4 + wave;
    const int cbw = cb0 < p.CB ? cb0 : p.CB - 1;
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(p.w) + (size_t)cbw * 64 + lane;
    bf16x8 A[2];
    A[0] = wfrag[0];
    A[1] = wfrag[(size_t)p.CB * 64];
    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;
    lds_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + (nb * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16 + ks * 32);
            acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], b, acc[0][nb], 0, 0, 0);
        }
    if (!conv_lean_ok(p)) {       // not what the stems of the nets ask for: generic epilogue
        conv_epilogue_lds<NB>(p, acc, smem + 128 * ROWB, n0, oy0, ox0, cot, wave, lane, tid);
        return;
    }
    conv_epilogue_lean<NB>(p, acc, smem + 128 * ROWB, n0, oy0, ox0, cot, wave, lane, tid);
}

// ---------------------------------------------------------------------------------------------------------
// 1x1 convolution (q|k|v, proj_out, nin_shortcut / skip_connection), streaming form: these layers are short dot products
// (K = 128..1344) over large maps - memory-bound.  One workgroup = one (64-pixel tile, 128-cout tile); RC chunks of the
// K extent (input pieces AND weight fragments) are requested at once, so a round is one memory round trip, two
// barriers and 2*RC k-steps of MFMAs; the next round's input flies during the MFMAs.  No persistence, no ring: with
// 108 registers and 33 KB of LDS FOUR workgroups share a CU and the hardware scheduler overlaps one tile's loads with
// another's MFMAs and stores (measured against the persistent register-ring variant it replaces: q|k|v 256->768 @16x16
// 86 -> 60 us, proj_out 39 -> 30 us; fewer, larger tiles were slower at every occupancy).  The epilogue slab reuses the
// input image.
template <int NB, int RC, int OCC>
__global__ __launch_bounds__(256, OCC) void conv1x1_stream_kernel(ConvArgs p) {
    constexpr int CK = 32, ROWB = CK * 2 + 16, PPP = CK / 8;
    constexpr int PM = NB / 2;           // pieces per thread per 32-channel chunk (tile_px * 4 / 256)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware: the CT cout tiles of one pixel tile sit on one XCD, adjacent in dispatch order
    const int nstreams = gridDim.x / p.CT;
    int cot, pt;
    if ((nstreams & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        cot = j % p.CT;
        pt = (j / p.CT) * 8 + xcd;
    } else {
        cot = blockIdx.x % p.CT;
        pt = blockIdx.x / p.CT;
    }
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
    const int tx = pt % txn, ty = (pt / txn) % tyn;
    const int n0 = (pt / (txn * tyn)) * p.SUBS, oy0 = ty << p.THl, ox0 = tx << p.TWl;
    const int nchunks = (p.C0 + p.C1) / CK;
    const int IMG = 32 * NB * ROWB;      // bytes of one chunk image

    int hoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) hoff[nb] = (nb * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const int pcol = tid % PPP;
    int sp[PM];
#pragma unroll
    for (int q = 0; q < PM; ++q) {
        const int pix = (tid + q * 256) / PPP;
        const int x = pix & (TW - 1);
        const int y = (pix >> p.TWl) & (TH - 1);
        const int n = n0 + (pix >> (p.TWl + p.THl));
        sp[q] = n < p.N ? (n * p.IH + oy0 + y) * p.IW + ox0 + x : -1;
    }
    const int cb0 = cot * 4 + wave;
    const int cbw = cb0 < p.CB ? cb0 : p.CB - 1;
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(p.w) + (size_t)cbw * 64 + lane;
    const size_t wstep = (size_t)p.CB * 64;

    f32x16 acc[1][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nb][r] = 0.f;

    bf16x8 stg[RC][PM], A[RC][2];
    auto request_in = [&](int c0) {      // input pieces of chunks c0 .. c0+RC-1 (clamped)
#pragma unroll
        for (int j = 0; j < RC; ++j) {
            const int c = c0 + j < nchunks ? c0 + j : nchunks - 1;
            const int cbase = c * CK;
            const bool first = cbase < p.C0;
            const bf16* src = first ? p.in0 : p.in1;
            const int Cs = first ? p.C0 : p.C1;
            const int coff = (first ? cbase : cbase - p.C0) + pcol * 8;
#pragma unroll
            for (int q = 0; q < PM; ++q) {
#pragma unroll
                for (int e = 0; e < 8; ++e) stg[j][q][e] = (bf16)0.f;
                if (sp[q] >= 0) stg[j][q] = *reinterpret_cast<const bf16x8*>(src + (size_t)sp[q] * Cs + coff);
            }
        }
    };
    auto request_w = [&](int c0) {       // weight fragments of the same chunks
#pragma unroll
        for (int j = 0; j < RC; ++j) {
            const int c = c0 + j < nchunks ? c0 + j : nchunks - 1;
            A[j][0] = wfrag[(size_t)(c * 2) * wstep];
            A[j][1] = wfrag[(size_t)(c * 2 + 1) * wstep];
        }
    };
    request_in(0);
    request_w(0);
    for (int c0 = 0; c0 < nchunks; c0 += RC) {
        if (c0) lds_barrier();           // every wave is done reading the previous round's images
#pragma unroll
        for (int j = 0; j < RC; ++j)
#pragma unroll
            for (int q = 0; q < PM; ++q) {
                const int pix = (tid + q * 256) / PPP;
                *reinterpret_cast<bf16x8*>(smem + j * IMG + pix * ROWB + pcol * 16) = stg[j][q];
            }
        lds_barrier();
        const int left = nchunks - c0;
        if (left > RC) request_in(c0 + RC);   // next round's input flies during this round's MFMAs
#pragma unroll
        for (int j = 0; j < RC; ++j) {
            if (j < left) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(smem + j * IMG + hoff[nb] + ks * 32);
                        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][ks], b, acc[0][nb], 0, 0, 0);
                    }
            }
        }
        if (left > RC) request_w(c0 + RC);
    }
    lds_barrier();                        // images are dead: the epilogue slab reuses them
    if (conv_lean_ok(p)) conv_epilogue_lean<NB>(p, acc, smem, n0, oy0, ox0, cot, wave, lane, tid);
    else conv_epilogue_lds<NB>(p, acc, smem, n0, oy0, ox0, cot, wave, lane, tid);
}

template <int NB, int RC, int OCC>
int launch_stream(const ConvArgs& a, int grid, hipStream_t st) {
    auto kern = conv1x1_stream_kernel<NB, RC, OCC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    size_t lds = (size_t)RC * 32 * NB * 80;
    if (lds < (size_t)EPI_BYTES) lds = EPI_BYTES;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(1x1 stream)");
    return DXMI_OK;
}

int ilog2p(int v);

int conv_stem_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    const int tile = 128;
    const int TW = a.OW < 32 ? a.OW : 32;
    int TH = tile / TW;
    if (TH > a.OH) TH = a.OH;
    const int SUBS = tile / (TW * TH);
    if (TW * TH * SUBS != tile) return 1;
    if (kernel_id) {
        *kernel_id = 300000;
        return DXMI_OK;
    }
    ConvArgs b = a;
    b.TWl = ilog2p(TW); b.THl = ilog2p(TH); b.SUBS = SUBS;
    const int ngroups = (a.N + SUBS - 1) / SUBS;
    b.PT = ngroups * (a.OH / TH) * (a.OW / TW);
    b.CT = (a.Cout + 127) / 128;
    b.tile_px = tile;
    hipLaunchKernelGGL(conv_stem_kernel, dim3(b.PT * b.CT), dim3(256), (size_t)128 * 80 + EPI_BYTES + 4096, st, b);     // + the 3-channel halo image
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(stem)");
    return DXMI_OK;
}

template <int NB, int PMAX, int KS, int DBG = 0, int AQ = 1, int SD = 1>
int launch_pipe(const ConvArgs& a, int grid, hipStream_t st) {
    auto kern = conv_pipe_kernel<NB, PMAX, KS, DBG, AQ, SD>;
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)2 * a.lds_buf + EPI_BYTES, st, a);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(pipe)");
    return DXMI_OK;
}

int ilog2p(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

int conv_ws_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);       // conv_ws.hip
int conv1x1_rw_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);   // conv1x1_rw.hip
int conv1x1_rw8_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);  // conv1x1_rw8.hip
int conv_head_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);    // conv_head.hip
int conv_ws8_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);     // conv_ws8.hip
int conv_sm_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);      // conv_sm.hip

// Pixels per tile of the kernels whose epilogue (conv_epilogue_lds) can emit GroupNorm block statistics, 0 for the others:
// conv_stem_kernel (300000), conv_pipe_kernel (k NB pmax = 10000 k + 100 NB + pmax) and conv1x1_stream_kernel (200000), for
// tiles inside one image and full 128-cout tiles.
int conv_pipe_stats_tile(int id, int OH, int OW, int Cout) {
    int tile = 0;
    if (id == 300000) tile = 128;
    else if (id == 200000) tile = 64;
    else if ((id >= 10000 && id < 20000) || (id >= 30000 && id < 40000)) tile = 32 * ((id % 10000) / 100);
    if (tile == 0 || Cout % 128 != 0 || OH * OW < tile || (OH * OW) % tile != 0) return 0;
    return tile;
}

int conv_pipe_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    if (a.gn_stats && !kernel_id) {
        // the statistics request must not change which kernel runs: select without it, then check that kernel can
        ConvArgs q = a;
        q.gn_stats = nullptr;
        int id = 0;
        const int rc = conv_pipe_try_launch(q, st, &id);
        if (rc < 0) return rc;
        DXMI_CHECK_ARG(rc == 0 && ((id >= 400000 && id < 400100) || conv_pipe_stats_tile(id, a.OH, a.OW, a.Cout) > 0),
                       "dxmi_conv2d_fwd: the kernel for this shape does not emit GroupNorm block statistics "
                       "(dxmi_conv2d_gn_stats_partials returns 0 for it)");
    }
    {
        int rc = 1;
        if (a.gn_out) {          // conv_sm_kernel (4x4 maps) and conv_ws8_kernel (8x8, instead of the raw output) fuse the GroupNorm of their output
            rc = conv_sm_try_launch(a, st, kernel_id);
            if (rc <= 0) return rc;
            rc = conv_ws8_try_launch(a, st, kernel_id);
            if (rc <= 0) return rc;
            DXMI_CHECK_ARG(false, "dxmi_conv2d_fwd: the kernel for this shape cannot fuse the GroupNorm of its output "
                                  "(dxmi_conv2d_gn_fuse_supported returns 0 for it)");
        }
        rc = conv_ws_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;
        // besides conv_ws_kernel only the stem / conv_pipe kernels below emit GroupNorm block statistics, for one-image tiles
        // (dxmi_conv2d_gn_stats_partials says which shapes); the other small-map kernels are skipped for such a request
        rc = conv1x1_rw8_try_launch(a, st, kernel_id);      // K = 576 (256-cout tiles, one 512-thread workgroup per CU)
        if (rc <= 0) return rc;
        rc = conv1x1_rw_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;
        rc = conv_head_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;
        rc = conv_sm_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;
        rc = conv_ws8_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;
    }
    if (a.in_mode == DXMI_IN_NCHW_F32_K27 && a.out_mode == DXMI_OUT_NHWC_BF16 && a.Cout % 64 == 0) return conv_stem_launch(a, st, kernel_id);
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    if (a.Cout % 64 != 0 || (a.C0 + a.C1) % 64 != 0 || a.C0 % 32 != 0) return 1;  // even chunk count
    if (a.stride != 1 && !(a.stride == 2 && a.ups == 0 && a.ksize == 3)) return 1;   // stride 2: Downsample convs
    const int CT = (a.Cout + 127) / 128;
    // pixel-tile size: the largest of 256/128/64 that still gives every CU a workgroup
    const long px = (long)a.N * a.OH * a.OW;
    int NB = 8;
    while (NB > 2 && (px / (32 * NB)) * CT < 512) NB >>= 1;   // two workgroups per CU
    // 128-pixel tiles (2 workgroups / CU: one's tile switch hides under the other's MFMAs) measured
    // faster than 256-pixel tiles at 1 workgroup / CU: default cap 4.
    static const int nb_env = getenv("DXMI_CONV_NB") ? atoi(getenv("DXMI_CONV_NB")) : 4;  // tuning override (2 or 4)
    const int nb_cap = nb_env >= 4 ? 4 : 2;
    while (NB > nb_cap && NB > 2) NB >>= 1;
    if (a.stride == 2) NB = 2;   // the stride-2 halo of a 64-pixel tile is 17x17 pixels (23 KB)
    if (a.ksize == 1 && a.ups == 0) NB = 2;   // 1x1: 64-pixel tiles, four workgroups per CU (conv1x1_stream_kernel)
    const int tile = 32 * NB;
    const int TW = a.OW < 32 ? a.OW : 32;
    int TH = tile / TW;
    if (TH > a.OH) TH = a.OH;
    const int SUBS = tile / (TW * TH);
    if (TW * TH * SUBS != tile) return 1;
    ConvArgs b = a;
    b.TWl = ilog2p(TW); b.THl = ilog2p(TH); b.SUBS = SUBS;
    b.HH = (TH - 1) * a.stride + a.ksize; b.HWd = (TW - 1) * a.stride + a.ksize;
    b.tile_px = tile;
    const int ngroups = (a.N + SUBS - 1) / SUBS;
    b.PT = ngroups * (a.OH / TH) * (a.OW / TW);
    b.CT = CT;
    const int HP = SUBS * b.HH * b.HWd;
    // LDS pitches: a 32-pixel MFMA block spans 32/TW halo rows; the ds_read_b128 lane groups stay
    // conflict-free when consecutive rows are offset by 0 / 128 / 64 bytes (mod 256) for TW = 16 / 8 / 4
    // (brute-forced over the b128 lane groups; TW = 32 is conflict-free at any pitch) and sub-images by
    // a multiple of 256.
    {
        const int want = TW == 16 ? 0 : (TW == 8 ? 128 : (TW == 4 ? 64 : -1));
        int rp = b.HWd * 80;
        if (want >= 0) while (rp % 256 != want) rp += 16;
        int sp = b.HH * rp;
        if (SUBS > 1) while (sp % 256 != 0) sp += 16;
        b.RP = rp; b.SP = sp;
    }
    b.lds_buf = SUBS * b.SP;
    if (2 * b.lds_buf + EPI_BYTES > 160 * 1024) return 1;
    const int npieces = HP * 4;
    if (npieces > 6 * 256) return 1;          // larger halos go to the generic kernel
    // staging pieces per thread: the small counts get the two-set (SD = 2) kernels
    const int pmax = (a.ksize == 3 && NB == 4 && npieces <= 4 * 256) ? 4 : 6;
    const bool stream1x1 = a.ksize == 1 && a.ups == 0;
    if (kernel_id) {
        // kxxyy = conv_pipe_kernel<xx, yy, k> ; 200000 = conv1x1_stream_kernel<2, 4, 4>
        *kernel_id = stream1x1 ? 200000 : 10000 * a.ksize + NB * 100 + pmax;
        return DXMI_OK;
    }
    // persistent grid: as many workgroups as are co-resident (2 per CU for NB <= 4, else 1), a
    // multiple of CT; each walks PT / nstreams pixel tiles.
    static const int wg_per_cu = getenv("DXMI_CONV_WGS") ? atoi(getenv("DXMI_CONV_WGS")) : 0;
    const int resident = 256 * (wg_per_cu > 0 ? wg_per_cu : (NB <= 4 ? 2 : 1));
    int nstreams = resident / CT;
    if (nstreams > b.PT) nstreams = b.PT;
    if (nstreams >= 8) nstreams &= ~7;   // whole XCD groups (see the kernel's block mapping)
    if (nstreams < 1) nstreams = 1;
    const int grid = nstreams * CT;
    static const int stagger = getenv("DXMI_CONV_STAGGER") ? atoi(getenv("DXMI_CONV_STAGGER")) : 2;   // measured: 0 -> 2 = -2 % conv time
    b.stagger = (grid > 256 && b.PT / nstreams >= 2) ? stagger : 0;
    if (stream1x1) return launch_stream<2, 4, 4>(b, b.PT * CT, st);
    // queue depth by shape: 3x3 at NB=2 (4x4 / 8x8 maps, latency bound) 8 ahead, 3x3 at NB=4 3 ahead with the two staging sets (4 ahead spills
    // at two workgroups per CU: measured 4.74 -> 4.63 ms of conv time per forward going from 4 to 3), 1x1 (short K loops, measured no gain) 1 ahead
    // staging distance: with few enough halo pieces per thread two register sets fit, and the halo of the chunk AFTER
    // next is requested (17 steps of cover instead of 8): -5..10 % on the K <= 2304 layers
    if (a.ksize == 3) {
        if (NB == 4) return pmax == 4 ? launch_pipe<4, 4, 3, 0, 3, 2>(b, grid, st) : launch_pipe<4, 6, 3, 0, 4>(b, grid, st);
        return launch_pipe<2, 6, 3, 0, 8>(b, grid, st);   // 64-pixel tiles: the second register set measured no gain
    }
    return NB == 4 ? launch_pipe<4, 6, 1>(b, grid, st) : launch_pipe<2, 6, 1>(b, grid, st);   // 1x1 behind an upsample (unused by the nets)
}
