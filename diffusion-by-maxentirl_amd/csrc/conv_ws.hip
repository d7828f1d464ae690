// Wave-specialised 3x3 convolution for gfx950: the MFMA waves never touch global memory.
//
// Why (measured on MI355X): in conv_pipe_kernel every wave both computes and moves data.  vmcnt retires
// in order, so a wave that has stores or residual loads in flight stalls at its next wait on a *younger* weight-fragment
// load until they are acknowledged; the tile end (residual in, output out) therefore costs more cycles than the tile's
// MFMAs, and at 256 VGPRs there is no room for a third resident workgroup to cover it.  Here ONE workgroup of 8 waves owns a
// CU, persistent over (256-pixel, 128-cout) tiles, and the roles are split (each SIMD hosts one wave of each kind):
//   waves 0-3  MFMA (v_mfma_f32_16x16x32_bf16): a 64-cout x 128-pixel accumulator tile each.  Operands come from LDS only —
//              weight fragments from a 6-slot ring, input pixels from a double-buffered halo image — at per-lane bases plus
//              compile-time offsets (no address arithmetic in the loop), with a half-step software pipeline: the reads of
//              the next 16 MFMAs fly under the current 16.  No vector-memory instruction is ever issued, so nothing a mover
//              does can stall them; the only waits are lgkmcnt for LDS reads and ONE barrier per GROUP of three (chunk, tap)
//              steps (round 4: a barrier per step cost 11-15 % by itself).
//   waves 4,5  weight loaders, register-staged (round 4): the ring is two groups of three slots; the fragments of group G + 1
//              are written from registers (ds_write_b128) while the MFMA waves read group G, after having been requested two
//              group steps earlier by plain global_load_dwordx4 (a `global_load_lds` costs its wave ~2x the issue time and
//              takes twice the issue slots from the MFMA wave on its SIMD: tools/stage_probe.hip).  They also fetch the bias /
//              temb table and, from the second tile on, the RESIDUAL tile (LDS DMA, one group step after the drain wave
//              emptied the piece), and do the next tile's first group step in front of E1.
//   wave 6     halo mover: the next 32-channel chunk's halo image (DMA, XOR-swizzled 16-byte slots instead of padding: a
//              DMA writes 1 KiB linearly), three blocks per step.  Loads only, so its vmcnt(0) at a chunk end waits for
//              nothing but the image.
//   wave 7     drain mover: the PREVIOUS tile's output LDS -> 256-byte NHWC rows, a piece or two per step.  Stores only: no
//              wait on a store's acknowledgement anywhere (round 3's two bulk movers shared both jobs and, vmcnt retiring in
//              order, waited at every chunk end for stores older than the halo blocks they needed).
// Tile end: the MFMA waves add bias + temb + residual to the accumulators in the accumulator layout (LDS reads of their
// own cout columns), apply the activation, round ONCE to bf16 and write the output tile in place in LDS; two barriers.  The
// epilogue is specialised on its launch-uniform switches (straight-line code per combination: 9.4 k -> 3.3 k cycles per tile).
// What was measured on the way, including what did NOT work (A fragments fetched by the MFMA waves, output stored from the
// accumulator layout, register-staged halo / residual with one group step or one chunk of lead): DESIGN.md 5.4.
//
// Scope: 3x3 / stride 1 / pad 1 (optionally behind a nearest x2 upsample), NHWC bf16 in (virtual concat) and out,
// Cout % 64 == 0 (half-empty last cout tile masked), an even number (>= 4) of 32-channel chunks, maps >= 16x16 (one image per 256-pixel tile); 8x8 maps: conv_ws8.hip.
// Everything else stays on conv_pipe.hip.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>


#ifdef DXMI_CONV_STAMPS
// timing-only build (make STAMPS=1): cycles MFMA wave 0 of every workgroup spends in each step barrier (tools/ws_stamps.py)
__device__ unsigned g_ws_wait[256][176];
extern "C" int dxmi_debug_read_ws_stamps(void* dst, int bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_wait), bytes, 0, hipMemcpyDeviceToHost);
}
#define WS_STAMPED_BARRIER(idx)                                                                     \
    do {                                                                                            \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();                                \
        ws_barrier();                                                                               \
        const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                                \
        if (wave == 0 && lane == 0 && (idx) < 160 && blockIdx.x < 256) g_ws_wait[blockIdx.x][(idx)] = (unsigned)(t1_ - t0_); \
    } while (0)
#define WS_CLOCKSTAMP(slot, cond)                                                                    \
    do {                                                                                            \
        if ((cond) && lane == 0 && blockIdx.x < 256) {                                              \
            g_ws_wait[blockIdx.x][(slot)] = (unsigned)__builtin_amdgcn_s_memtime();                 \
            g_ws_wait[blockIdx.x][(slot) + 1] = (unsigned)__builtin_amdgcn_s_memrealtime();         \
        }                                                                                           \
    } while (0)
#define WS_TSTAMP(slot, cond)                                                                        \
    do {                                                                                            \
        if ((cond) && lane == 0 && blockIdx.x < 256) g_ws_wait[blockIdx.x][(slot)] = (unsigned)__builtin_amdgcn_s_memtime(); \
    } while (0)
// timing-only ablations of the diagnostic build (DXMI_CONV_WS_DBG bits: 1 no weight stream, 2 no halo stream, 4 no drain,
// 8 no residual / table fetch, 16 no step barriers — wrong results); compiled out of the product library
#define WS_DBG(bit) (p.stagger & (bit))
#else
#define WS_STAMPED_BARRIER(idx) ws_barrier()
#define WS_TSTAMP(slot, cond) do {} while (0)
#define WS_CLOCKSTAMP(slot, cond) do {} while (0)
#define WS_DBG(bit) 0
#endif

// Shader clock of the last launch (round 5): workgroup 0 leaves (s_memtime, s_memrealtime) of its first and last instruction
// here; d s_memtime / d s_memrealtime x 100 MHz is the clock the chip held DURING the kernel (it lowers the clock under this
// kernel's load: 1.86-2.13 GHz).  bench.py reports it beside the roofline fraction, so that a slow box (or a throttling one) can be
// told from a slow kernel.  Two scalar reads and one 32-byte store per launch.
__device__ unsigned long long g_ws_clock[4];
extern "C" int dxmi_conv_ws_last_clock(unsigned long long* host_out4) {
    return hipMemcpyFromSymbol(host_out4, HIP_SYMBOL(g_ws_clock), 4 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost) == hipSuccess ? DXMI_OK
                                                                                                                                          : DXMI_ELAUNCH;
}

namespace {

constexpr int WS_A_SLOT = 8192;              // one (chunk, tap): 2 k-steps x 4 cout blocks x 1 KiB fragments
constexpr int WS_RING = 6;                   // ring slots (steps): five steps of weight prefetch
constexpr int WS_A_RING = WS_RING * WS_A_SLOT;
constexpr int WS_RO = 256 * 256;             // output / residual tile: 256 px x 128 co bf16
constexpr int WS_TB = 1024;                  // bias[128] | temb[128] fp32
constexpr int WS_HALO_BLOCKS = 22;           // 1-KiB blocks of a halo image (34x10 or 18x18 pixels x 64 B)
constexpr int WS_HALO = WS_HALO_BLOCKS * 1024;

__device__ uint4 ws_zero16 = {0u, 0u, 0u, 0u};   // source of zero-padding pixels

#define WS_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define WS_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 MFMAs (16 cycles each) with 4 / 8 operand reads spread between them
#define WS_INTERLEAVE_4()                                       \
    do {                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
        }                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      \
    } while (0)
#define WS_INTERLEAVE_8()                                       \
    do {                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
        }                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      \
    } while (0)

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): n is rounded DOWN to an even
// count, which only waits for more
__device__ __forceinline__ void ws_wait_vm(int n) {
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
    default: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    }
}

struct WsTile {
    int cot, n0, oy0, ox0;
};

#ifdef WS_VALU_PIECES
// Measurement build (round 5, tools/build_variant.sh ... -DWS_VALU_PIECES=n): every mover wave runs n 16-byte pieces of the
// GroupNorm(+SiLU) transform (bf16 -> fma -> SiLU -> bf16 on dummy registers: no LDS, no table loads) per group step — 2 per group
// step and wave is the VALU work of transforming every halo chunk on load (22 KiB per chunk over four waves).  What the consuming
// conv would pay AT LEAST for applying the normalisation itself; results are unchanged (the values go nowhere).
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
struct WsValuLoad {
    u32x4 r[2];
    ws_f32x2 A[4], B[4];
    unsigned sink;
    __device__ __forceinline__ void init(int lane) {
        for (int k = 0; k < 2; ++k) r[k] = u32x4{0x3f803f00u + lane + k, 0xbf803e80u + k, 0x3e003f80u + lane, 0x40003f00u - k};
        for (int i = 0; i < 4; ++i) {
            A[i] = ws_f32x2{0.9f + 0.01f * lane, 1.1f - 0.01f * i};
            B[i] = ws_f32x2{0.05f * i, -0.03f * lane};
        }
        sink = 0;
    }
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int k = 0; k < WS_VALU_PIECES; ++k) {
            u32x4& raw = r[k & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ws_f32x2 x = {__builtin_bit_cast(float, raw[i] << 16), __builtin_bit_cast(float, raw[i] & 0xffff0000u)};
                ws_f32x2 y = x * A[i] + B[i];
                const ws_f32x2 t = y * ws_f32x2{-1.4426950408889634f, -1.4426950408889634f};
                ws_f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                e = e + ws_f32x2{1.f, 1.f};
                const ws_f32x2 rc = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
                y = y * rc;
                bf16x2 o = {(bf16)y[0], (bf16)y[1]};
                sink ^= __builtin_bit_cast(unsigned, o);
            }
            raw += u32x4{0x10001u, 0x20003u, 0x10002u, 0x30001u};
        }
    }
};
#define WS_VALU_DECL(lane_) WsValuLoad vl_; vl_.init(lane_)
#define WS_VALU_RUN() vl_.run()
#define WS_VALU_DONE() do { if (vl_.sink == 0x12345u) reinterpret_cast<volatile unsigned*>(p.out)[0] = vl_.sink; } while (0)
#else
#define WS_VALU_DECL(lane_) do {} while (0)
#define WS_VALU_RUN() do {} while (0)
#define WS_VALU_DONE() do {} while (0)
#endif

// Halo image: pixel (hy, hx) of the (TH+2) x (TW+2) halo at row pitch HP pixels, 64 B (32 channels) per pixel in four
// 16-byte slots; channel piece j sits in slot j ^ (2 * ((hx >> 2) & 1)).  The MFMA shape is 16x16x32 (on random data the chip
// holds a higher clock on it than on 32x32x16 at equal cycles per FLOP: MI355X_MICROARCH.md, DVFS give-back item 7): a B
// fragment is 16 consecutive pixels x the whole 64-byte row (lane = pixel + 16 * channel piece), one MFMA consumes a whole
// 32-channel chunk of one tap.  With that swizzle the ds_read_b128 lane groups of every tap are bank-conflict free at
// HP = 34 (TW 32) and HP = 18 (TW 16) (brute-forced), and because it depends on hx only, a lane's B-operand address is one
// of three precomputed bases (kx) plus a compile-time offset: no address arithmetic in the MFMA loop.
// The weight ring keeps the 32x32x16 fragment packing of pack_conv_weight (shared with conv_pipe): the 16x16x32 A fragment
// (16 couts x 32 channels) of cout block cb16 is lanes (cb16 & 1) * 16 .. + 15 of both lane halves of the two k-step
// fragments of cout block cb16 >> 1 — a per-lane base plus a compile-time offset as well, bank-conflict free as it lies.
// (Round 4 carried a second instantiation that wrote GroupNorm(+SiLU) of a 16x16 map's output from this epilogue: parity-green, 4 %
// slower end to end — two more passes over 128 accumulators on the MFMA waves' critical path — and removed in round 5: DESIGN 5.4 / 5.5.)
template <int TW>
__device__ __forceinline__ void conv_ws_body(const ConvArgs& p) {
    constexpr int TH = 256 / TW, HP = TW == 32 ? 34 : 18, HH = TH + 2, TWl = TW == 32 ? 5 : 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const halo0 = smem;
    char* const aring = smem + 2 * WS_HALO;
    char* const ro = aring + WS_A_RING;
    float* const tb = reinterpret_cast<float*>(ro + WS_RO);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int txn = p.OW / TW, tyn = p.OH / TH;
    const int nchunks = (p.C0 + p.C1) / 32;
    const int S = 9 * nchunks;                       // steps per tile
    const int ntiles = p.PT * p.CT;                  // (pixel tile, cout tile) pairs, cout fastest
    const int ups = p.ups ? 1 : 0;

    auto tile_of = [&](int q, WsTile& t) {
        t.cot = q % p.CT;
        const int pt = q / p.CT;
        const int tx = pt % txn, ty = (pt / txn) % tyn;
        t.n0 = pt / (txn * tyn);
        t.oy0 = ty * TH;
        t.ox0 = tx * TW;
    };

    const int q0 = dxmi_xcd_logical(blockIdx.x, gridDim.x, p.xcd_order);
    int q = q0;
    if (q >= ntiles) return;
    const int qstride = gridDim.x;
    if (blockIdx.x == 0 && tid == 0) {
        g_ws_clock[0] = __builtin_amdgcn_s_memtime();
        g_ws_clock[1] = __builtin_amdgcn_s_memrealtime();
    }
    WS_CLOCKSTAMP(172, wave == 0);

    if (wave < 4) {
        // ================================================================ MFMA waves: 64 couts (ch) x 128 pixels (ph)
        const int ch = wave & 1, ph = wave >> 1;
        const int px = lane & 15, kg = lane >> 4;      // B: pixel / channel piece; A: cout / 8-channel group; D: pixel / 4-cout group
        // 16-pixel block nb (0..7) of this wave is tile pixels (ph*8 + nb)*16 ..: row (ph*8 + nb) >> (TW / 32), column ((nb & 1) * 16 at TW 32)
        constexpr int BROW = TW == 32 ? 2 : 1;         // blocks per tile row
        int bbase[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            bbase[kx] = ((ph * (8 / BROW) * HP + px + kx) * 64) + ((kg ^ (2 * (((px + kx) >> 2) & 1))) << 4);
        const char* const abase = aring + ch * 2048 + ((kg >> 1) * 4096) + ((px + 32 * (kg & 1)) << 4);   // + slot*8192 + (cb16>>1)*1024 + (cb16&1)*256
        f32x4 acc[4][8];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[cb][nb][r] = 0.f;
        bf16x8 A0[4], A1[4], Bx[4], By[4];
        // operands of step u of a chunk PAIR (u = 0..17: chunk parity u / 9, tap u % 9; u = 18 is the first step of the next
        // pair / the next tile): ring slot, halo image and tap offset are compile-time constants
        auto read_a = [&](int u, bf16x8 (&A)[4]) {
            const char* as = abase + (u % WS_RING) * WS_A_SLOT;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) A[cb] = *reinterpret_cast<const bf16x8*>(as + (cb >> 1) * 1024 + (cb & 1) * 256);
        };
        auto read_b = [&](int u, int half, bf16x8 (&B)[4]) {
            const int t = u % 9, ky = t / 3, kx = t % 3;
            const char* hb = halo0 + ((u / 9) & 1) * WS_HALO + bbase[kx];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nb = half * 4 + i;
                B[i] = *reinterpret_cast<const bf16x8*>(hb + ((nb / BROW + ky) * HP + (nb % BROW) * 16) * 64);
            }
        };
        auto mfma16 = [&](const bf16x8 (&A)[4], const bf16x8 (&B)[4], int half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb][half * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[cb], B[i], acc[cb][half * 4 + i], 0, 0, 0);
        };
        const float slope = dxmi_act_slope(p.act);
        const bool has_res = p.residual != nullptr;

        // (no s_setprio: giving the MFMA waves priority over the movers that share their SIMDs measured 5 % SLOWER on the
        // residual layers — late movers cost more than contended issue slots — and equal elsewhere)
        ws_barrier();                                   // P0: tap 0 and the first halo chunk have landed
        WS_CLOCKSTAMP(168, wave == 0);
        read_a(0, A0);
        read_b(0, 0, Bx);
        int stamp_i = 0;                                // stamp build: index of the tile's first step
        (void)stamp_i;
        for (;;) {
            const bool more = q + qstride < ntiles;
            // K loop: two chunks (18 steps) of straight-line code per iteration, no branches inside (a branch makes hipcc wait
            // lgkmcnt(0) at the join, which exposes the latency of the operand reads just issued).  Every half step (16 MFMAs)
            // requests the operands of the next one: the other 4 pixel blocks of this step, then — behind the barrier that
            // says they landed — the weights and first 4 blocks of the next step.  The first operands of the step after the
            // last one (the next tile's first step, already in LDS when the last barrier of this tile opens) are requested
            // unconditionally.
            for (int c = 0; c < nchunks; c += 2) {
#pragma unroll
                for (int u = 0; u < 18; ++u) {
                    read_b(u, 1, By);
                    if (u & 1) mfma16(A1, Bx, 0); else mfma16(A0, Bx, 0);
                    WS_INTERLEAVE_4();
                    // groups of three steps: the loaders hand over three ring slots at a time (register-staged), so the MFMA
                    // stream is interrupted by a barrier every 96 MFMAs instead of every 32
                    if (u % 3 == 2) { if (!(WS_DBG(16))) WS_STAMPED_BARRIER(stamp_i + c * 9 + u); }
                    if (u & 1) read_a(u + 1, A0); else read_a(u + 1, A1);
                    read_b(u + 1, 0, Bx);
                    if (u & 1) mfma16(A1, By, 1); else mfma16(A0, By, 1);
                    WS_INTERLEAVE_8();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            WS_TSTAMP(158, wave == 0 && q == q0);
            ws_barrier();                               // E1: residual tile + bias / temb table of this tile landed
            WS_TSTAMP(159, wave == 0 && q == q0);
            {
                // acc + bias + temb (+ residual) -> activation -> bf16 -> output tile, in place, accumulator layout:
                // lane = pixel (lane & 15) of a 16-pixel block, couts 4 * (lane >> 4) .. + 3 of a 16-cout block.  The epilogue is
                // VALU work on the MFMA waves' critical path (stamp build: 4.2 k cycles per tile, 7.4 k with a residual, before
                // this form): every address is a per-lane base + compile-time offset, the residual pieces of a cout block are
                // requested together, and the activation code is skipped when there is none (the U-Net's convs).
                const bool plain = p.act == DXMI_ACT_NONE;
                char* const rowb = ro + ((ph * 128 + px) << 8) + 8 * (kg & 1);         // pixel row of block 0: blocks are 4 KiB apart
                // GroupNorm block statistics of the output (optional): this wave's 128 pixels x its 64 couts -> (sum, sum of
                // squares) per cout PAIR, partial (pixel tile, pixel half) of the image: [pt * 2 + ph][Cout / 2][2].  A lane's
                // four accumulator values of a (cb, nb) are two pairs of one pixel: 8 pixels in registers, 16 in the DPP row.
                const bool want_stats = p.gn_stats != nullptr;
                const int cot_ = q % p.CT;
                float* const stp = want_stats ? p.gn_stats + ((size_t)((q / p.CT) * 2 + ph) * (p.Cout >> 1) + cot_ * 64 + ch * 32 + kg * 2) * 2 : nullptr;
                // The run-time switches (residual / mask / activation / statistics) are uniform for the launch: as branches inside
                // the 32 (cb, nb) bodies they cost ~130 scalar branches and their s_nop shadows per tile and keep hipcc from
                // batching the LDS traffic (round 4: the epilogue was 9.4 k of a 128-channel tile's 38 k cycles).  The U-Net's
                // four combinations get straight-line code; everything else takes the generic body.
                auto epi = [&](auto RES, auto MASK, auto ACT, auto STATS) {
                    constexpr bool kRes = decltype(RES)::value, kMask = decltype(MASK)::value, kAct = decltype(ACT)::value, kStats = decltype(STATS)::value;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        float st[4] = {0.f, 0.f, 0.f, 0.f};
                        const int co = ch * 64 + cb * 16 + 4 * kg;
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(tb + co), t0 = *reinterpret_cast<const f32x4*>(tb + 128 + co);
                        f32x4 bv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = b0[e] + t0[e];
                        char* const a0 = rowb + (((ch * 8 + cb * 2 + (kg >> 1)) ^ px) << 4);   // slot of this cout piece: (c8 ^ (pixel & 15))
                        bf16x4 r[8];
                        if constexpr (kRes) {
#pragma unroll
                            for (int nb = 0; nb < 8; ++nb) r[nb] = *reinterpret_cast<const bf16x4*>(a0 + nb * 4096);
                        }
#pragma unroll
                        for (int nb = 0; nb < 8; ++nb) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[cb][nb][e] + bv[e];
                            if constexpr (kRes) {
                                if constexpr (kMask) {    // data gradient through a LeakyReLU: the tile the movers fetched is the mask source
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
                            *reinterpret_cast<bf16x4*>(a0 + nb * 4096) = o;
                            if constexpr (kStats) dxmi_stats4(o, st);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[cb][nb][e] = 0.f;
                        }
                        if constexpr (kStats) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) st[e] = dxmi_row16_sum(st[e]);
                            if (px == 0 && cot_ * 128 + co < p.Cout) {
                                f32x4 sv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) sv[e] = st[e];
                                *reinterpret_cast<f32x4*>(stp + cb * 16) = sv;     // pairs (co / 2, co / 2 + 1): 8 pairs per 16-cout block
                            }
                        }
                    }
                };
                using T_ = std::true_type;
                using F_ = std::false_type;
                const bool is_mask = has_res && p.res_is_mask;
                if (plain && !is_mask) {                 // the U-Net's convs
                    if (has_res) { if (want_stats) epi(T_{}, F_{}, F_{}, T_{}); else epi(T_{}, F_{}, F_{}, F_{}); }
                    else { if (want_stats) epi(F_{}, F_{}, F_{}, T_{}); else epi(F_{}, F_{}, F_{}, F_{}); }
                } else if (!want_stats) {                // the value net: LeakyReLU forward / masked data gradients
                    if (is_mask) { if (plain) epi(T_{}, T_{}, F_{}, F_{}); else epi(T_{}, T_{}, T_{}, F_{}); }
                    else if (has_res) epi(T_{}, F_{}, T_{}, F_{});
                    else epi(F_{}, F_{}, T_{}, F_{});
                } else {                                 // statistics together with an activation / a mask
                    if (is_mask) epi(T_{}, T_{}, T_{}, T_{});
                    else if (has_res) epi(T_{}, F_{}, T_{}, T_{});
                    else epi(F_{}, F_{}, T_{}, T_{});
                }
            }
            WS_TSTAMP(166, wave == 0 && q == q0);
            ws_barrier();                               // E2: output tile complete, the bulk movers may drain it
            WS_TSTAMP(157, wave == 0 && q == q0);
            if (!more) {
                WS_CLOCKSTAMP(170, wave == 0);
                break;
            }
            q += qstride;
            stamp_i += S;
        }
    } else if (wave < 6) {
        // ================================================================ weight loaders (waves 4, 5)
        WS_VALU_DECL(lane);
        const int lw = wave - 4;
        const char* const wb = reinterpret_cast<const char*>(p.w) + (size_t)lane * 16;
        WsTile cur;
        tile_of(q, cur);
        // fragments lw*4 .. lw*4+3 of (chunk c, tap t) for cout tile cot -> ring slot
        auto issue_tap = [&](int cot, int g) {
            if (WS_DBG(1)) return;                  // timing-only ablation (DXMI_CONV_WS_DBG): no weight stream
            const int c = g / 9, t = g - c * 9;
            char* dst = aring + (g % WS_RING) * WS_A_SLOT + lw * 4096;
            // fragments (ks = lw, cbg = 0..3): contiguous 4 KiB of the packed weights
            const char* src = wb + ((size_t)(t * p.KST + c * 2 + lw) * p.CB) * 1024;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                // a half-empty last cout tile (Cout % 128 == 64): its missing fragments re-read the last valid one (results unused)
                const int cb = cot * 4 + f < p.CB ? cot * 4 + f : p.CB - 1;
                __builtin_amdgcn_global_load_lds(WS_GPTR(src + (size_t)cb * 1024), WS_LPTR(dst + f * 1024), 16, 0, 0);
            }
        };
        // Round 4: register-staged weight stream, handed over in GROUPS of three steps.  The ring's six slots are two groups: the
        // MFMA waves read group G (slots of parity G & 1) while this wave writes group G + 1 into the other three slots (last
        // read in group G - 1: complete at the barrier that ended it) from registers, with ds_write_b128.  The fragments were
        // requested two group steps earlier by plain global_load_dwordx4 (two register sets of 12 x 4 registers): a
        // `global_load_lds` costs its wave ~190 cycles of issue (tools/dma_probe.hip, tools/stage_probe.hip), four per step
        // and loader were more than a 512-cycle step, and one workgroup barrier per step cost 11-15 % by itself
        // (DXMI_CONV_WS_DBG=15 vs 31).  vmcnt retires in order: each wait leaves the 12 loads of the younger group in flight.
        u32x4 fq[2][3][4];
        auto load_tap = [&](int cot, int g, u32x4 (&f)[4]) {
            const int c = g / 9, t = g - c * 9;
            const char* src = wb + ((size_t)(t * p.KST + c * 2 + lw) * p.CB) * 1024;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cb = cot * 4 + k < p.CB ? cot * 4 + k : p.CB - 1;
                // inline asm: hipcc's own wait-count pass loses the count of loads in flight across the loop's back edge and
                // drains the queue (s_waitcnt vmcnt(0)) before the first ds_write of every trip; the waits below are exact
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f[k]) : "v"(src + (size_t)cb * 1024) : "memory");
            }
        };
        auto store_tap = [&](int g, const u32x4 (&f)[4]) {
            char* dst = aring + (g % WS_RING) * WS_A_SLOT + lw * 4096 + lane * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(dst + k * 1024) = f[k];
        };
        const int NG = S / 3;                           // groups per tile (even: S = 9 * nchunks, nchunks even)
        // group G of the current tile, or group G - NG of the next one (past the last tile: the current tile's again, unused)
        auto load_group = [&](int G, int cot_cur, int cot_nxt, u32x4 (&f)[3][4]) {
            const bool nx = G >= NG;
            const int g0 = (nx ? G - NG : G) * 3, cot = nx ? cot_nxt : cot_cur;
#pragma unroll
            for (int j = 0; j < 3; ++j) load_tap(cot, g0 + j, f[j]);
        };
        auto store_group = [&](int G, const u32x4 (&f)[3][4]) {     // ring slots depend on G's parity only (NG even)
#pragma unroll
            for (int j = 0; j < 3; ++j) store_tap((G & 1) * 3 + j, f[j]);
        };
        if (WS_DBG(1)) {}                               // (timing-only ablation of the weight stream: not available in this build)
        // residual pieces (see the bulk movers: piece L = k*128 + t2 with t2 = lw*64 + lane)
        const int t2l = lw * 64 + lane, prl = t2l >> 4;
        const int lpar_l0 = prl * p.Cout + (((t2l & 15) ^ prl) * 8), lpar_l1 = prl * p.Cout + (((t2l & 15) ^ (prl | 8)) * 8);
        const int c8e_l = (t2l & 15) ^ prl, c8o_l = (t2l & 15) ^ (prl | 8);
        const int kpc_l = (32 + nchunks - 2) / (nchunks - 1);
        const char* const zero_page_l = reinterpret_cast<const char*>(p.mask_src);
        bool first_tile_l = true, first_group0 = true;
        load_group(0, cur.cot, cur.cot, fq[0]);
        load_group(1, cur.cot, cur.cot, fq[1]);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        store_group(0, fq[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        load_group(2, cur.cot, cur.cot, fq[0]);
        ws_barrier();                                       // P0 (its lgkmcnt(0): group 0 is in LDS)
        bool first_tile = true;
        for (;;) {
            const bool more = q + qstride < ntiles;
            WsTile nxt = cur;
            if (more) tile_of(q + qstride, nxt);
            if (!first_tile && lw == 0 && !(WS_DBG(8))) {
                // bias / temb table of this tile: four 256-byte DMAs issued after E2 (the previous epilogue is done with the
                // table); from the second group step on every wait below covers them (they are older than the 12 youngest)
                const int co_a = cur.cot * 128 + lane, co_b = co_a + 64 < p.Cout ? co_a + 64 : p.Cout - 1;
                if (p.bias) {
                    __builtin_amdgcn_global_load_lds(WS_GPTR(p.bias + co_a), WS_LPTR(tb), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds(WS_GPTR(p.bias + co_b), WS_LPTR(tb + 64), 4, 0, 0);
                }
                if (p.addvec) {
                    const float* av = p.addvec + (size_t)cur.n0 * p.addvec_ld;
                    __builtin_amdgcn_global_load_lds(WS_GPTR(av + co_a), WS_LPTR(tb + 128), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds(WS_GPTR(av + co_b), WS_LPTR(tb + 192), 4, 0, 0);
                }
            }
            // The residual tile of THIS tile (tile switch): the bulk movers drain piece k of the previous tile's output in the
            // group step their schedule gives it; its residual replacement is fetched here one group step later (the group barrier
            // orders the two) — the loaders idle for two thirds of a group step, the bulk movers were 3-4 k cycles per chunk short
            // (stamp build: 10 k cycles of barrier waits per 128-channel tile).  The wait below leaves only the 12 youngest
            // operations in flight, so these DMAs (older than the next group step's loads) are complete two group steps later:
            // before E1, because the last chunk of a tile carries no pieces.
            const bf16* const res_t = p.residual + ((((size_t)cur.n0 * p.OH + cur.oy0) * p.OW + cur.ox0) * p.Cout + cur.cot * 128);
            const bool res_here = !first_tile_l && p.residual != nullptr && !(WS_DBG(8));
            first_tile_l = false;
            first_tile = false;
            for (int G0 = 0; G0 < NG; G0 += 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int G = G0 + j;
                    // group step G (the MFMA waves read group G): group G + 1 sits in set (j + 1) & 1 since group step G - 1.
                    // (Group step 0 of every tile but the first was done ahead, in front of E1 of the previous tile: see below.)
                    if (G > 0 || first_group0) {
                        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // ... and has landed (group G + 2's loads stay in flight)
                        store_group(G + 1, fq[(j + 1) & 1]);
                        // the barrier's lgkmcnt(0) must precede the refill: a ds_write reads its data registers when it executes
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (res_here && G > 0) {
                        const int cg = (G - 1) / 3, g3 = (G - 1) - cg * 3;     // the group step whose pieces were drained
                        if (cg + 1 < nchunks) {
                            const int k0 = cg * kpc_l < 32 ? cg * kpc_l : 32, k1 = k0 + kpc_l < 32 ? k0 + kpc_l : 32;
                            const int i0 = g3 < 2 ? 3 * g3 : 6, i1 = g3 < 2 ? 3 * g3 + 3 : 12;      // steps 3 g3 .. 3 g3 + 2 of piece_range()
                            const int ka = k0 + i0 < k1 ? k0 + i0 : k1, kb = k0 + i1 < k1 ? k0 + i1 : k1;
#pragma unroll 1
                            for (int k = ka; k < kb; ++k) {
                                const int rel = ((k >> (TWl - 3)) * p.OW + (k & (TW / 8 - 1)) * 8) * p.Cout + ((k & 1) ? lpar_l1 : lpar_l0);
                                const bool ok = cur.cot * 128 + ((k & 1) ? c8o_l : c8e_l) * 8 < p.Cout;
                                const void* g = ok ? (const void*)(res_t + rel) : (const void*)zero_page_l;
                                __builtin_amdgcn_global_load_lds(WS_GPTR(g), WS_LPTR(ro + (k * 128 + lw * 64) * 16), 16, 0, 0);
                            }
                        }
                    }
                    if (G > 0 || first_group0) load_group(G + 3, cur.cot, more ? nxt.cot : cur.cot, fq[(j + 1) & 1]);
                    WS_TSTAMP(151, wave == 4 && G == 0 && q == q0 + qstride);
                    WS_VALU_RUN();
                    if (!(WS_DBG(16))) ws_barrier();                        // end of group step G
                }
            }
            first_group0 = false;
            if (more) {
                // group step 0 of the NEXT tile, ahead of time: its ring slots (parity 1) were last read in this tile's last
                // group step, and after E2 every wave starts at once — the tile-start bookkeeping plus this block made the
                // loaders 1.1 k cycles late at the next tile's first group barrier (stamp build: 2.7 k cycles after E2)
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                store_group(1, fq[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                load_group(3, nxt.cot, nxt.cot, fq[1]);
            }
            ws_barrier();                                        // E1
            ws_barrier();                                        // E2
            WS_TSTAMP(150, wave == 4 && q == q0);
            if (!more) break;
            q += qstride;
            cur = nxt;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_VALU_DONE();
    } else if (wave == 6) {
        // ================================================================ halo mover (wave 6): loads only
        // Round 4: the two bulk movers no longer share every job.  vmcnt retires in order, so a wave that stores (the drain) and
        // loads (the halo image) waits, at the chunk end, for the acknowledgement of stores it issued long before the halo
        // blocks it actually needs (stamp build: 300-850 cycles late at every group barrier of a tile switch).  This wave issues
        // nothing but the next chunk's halo image (22 DMAs per chunk, three per step) and waits with vmcnt(0).
        WS_VALU_DECL(lane);
        constexpr int HBA = WS_HALO_BLOCKS;
        const char* const zero_page = reinterpret_cast<const char*>(p.mask_src);
        int hrel0[HBA], hrel1[HBA];
        unsigned m_in = 0, m_top = 0, m_bot = 0, m_left = 0, m_right = 0;
#pragma unroll
        for (int k = 0; k < HBA; ++k) {
            const int hp = k * 16 + (lane >> 2);
            const int hy = hp / HP, hx = hp - hy * HP;
            const int rel = ((hy - 1) >> ups) * p.IW + ((hx - 1) >> ups);
            const int j8 = ((lane & 3) ^ (2 * ((hx >> 2) & 1))) * 8;
            hrel0[k] = (rel * p.C0 + j8) * 2;
            hrel1[k] = (rel * p.C1 + j8) * 2;
            m_in |= (unsigned)(hy < HH && hx < TW + 2) << k;
            m_top |= (unsigned)(hy == 0) << k;
            m_bot |= (unsigned)(hy == HH - 1) << k;
            m_left |= (unsigned)(hx == 0) << k;
            m_right |= (unsigned)(hx == TW + 1) << k;
        }
        unsigned hmask = 0;
        int hbase0 = 0, hbase1 = 0;
        auto halo_plan = [&](const WsTile& t) {
            hmask = m_in;
            if (t.oy0 == 0) hmask &= ~m_top;
            if (t.oy0 + TH == p.OH) hmask &= ~m_bot;
            if (t.ox0 == 0) hmask &= ~m_left;
            if (t.ox0 + TW == p.OW) hmask &= ~m_right;
            const int pix = (t.n0 * p.IH + (t.oy0 >> ups)) * p.IW + (t.ox0 >> ups);
            hbase0 = pix * p.C0 * 2;
            hbase1 = pix * p.C1 * 2;
        };
        auto halo_issue = [&](int c, char* buf, int ka, int kb) {     // blocks ka .. kb-1 (compile-time range)
            if (WS_DBG(2)) return;
            const int cbase = c * 32;
            const bool first = cbase < p.C0;
            const char* base = reinterpret_cast<const char*>(first ? p.in0 : p.in1) + (first ? cbase : cbase - p.C0) * 2 + (first ? hbase0 : hbase1);
#pragma unroll
            for (int k = 0; k < HBA; ++k) {
                if (k < ka || k >= kb) continue;
                const char* g = (hmask >> k) & 1 ? base + (first ? hrel0[k] : hrel1[k]) : zero_page;
                __builtin_amdgcn_global_load_lds(WS_GPTR(g), WS_LPTR(buf + k * 1024), 16, 0, 0);
            }
        };
        WsTile cur;
        tile_of(q, cur);
        if (!p.bias) { tb[lane] = 0.f; tb[64 + lane] = 0.f; }            // table rows that no DMA fills stay zero
        if (!p.addvec) { tb[128 + lane] = 0.f; tb[192 + lane] = 0.f; }
        halo_plan(cur);
        halo_issue(0, halo0, 0, HBA);
        if (!(WS_DBG(8))) {                                              // first tile's bias / temb table (later ones: loader 0)
            const int co_a = cur.cot * 128 + lane, co_b = co_a + 64 < p.Cout ? co_a + 64 : p.Cout - 1;
            if (p.bias) {
                __builtin_amdgcn_global_load_lds(WS_GPTR(p.bias + co_a), WS_LPTR(tb), 4, 0, 0);
                __builtin_amdgcn_global_load_lds(WS_GPTR(p.bias + co_b), WS_LPTR(tb + 64), 4, 0, 0);
            }
            if (p.addvec) {
                const float* av = p.addvec + (size_t)cur.n0 * p.addvec_ld;
                __builtin_amdgcn_global_load_lds(WS_GPTR(av + co_a), WS_LPTR(tb + 128), 4, 0, 0);
                __builtin_amdgcn_global_load_lds(WS_GPTR(av + co_b), WS_LPTR(tb + 192), 4, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ws_barrier();                                           // P0
        WsTile nxt = cur;
        for (;;) {
            const bool more = q + qstride < ntiles;
            for (int c = 0; c < nchunks; ++c) {
                const bool wrap = c + 1 == nchunks;
                if (wrap && more) {
                    tile_of(q + qstride, nxt);
                    halo_plan(nxt);
                }
                const bool do_halo = !wrap || more;
                char* const hbuf = halo0 + ((c + 1) & 1) * WS_HALO;
                const int hc = wrap ? 0 : c + 1;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t < 8 && do_halo) halo_issue(hc, hbuf, 3 * t, 3 * t + 3 < HBA ? 3 * t + 3 : HBA);
                    if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the halo image landed
                    if (t % 3 == 2) { WS_VALU_RUN(); if (!(WS_DBG(16))) ws_barrier(); }            // end of a group step
                }
            }
            ws_barrier();                                        // E1
            ws_barrier();                                        // E2
            if (!more) break;
            q += qstride;
            cur = nxt;
        }
        WS_VALU_DONE();
    } else if (wave == 7) {
        // ================================================================ drain mover (wave 7): LDS -> global stores only
        // All 32 pieces (2 KiB each: both 64-lane halves) of the previous tile's output, on the schedule of piece_range(); the
        // loaders refill a piece with the residual one group step later.  No load, so no wait on a store's acknowledgement
        // anywhere but in front of E1 of the first tile (whose residual tile this wave fetched in the prologue).
        WS_VALU_DECL(lane);
        const char* const zero_page = reinterpret_cast<const char*>(p.mask_src);
        const int prA = lane >> 4, prB = 4 + (lane >> 4), lc = lane & 15;
        const int lparA[2] = {prA * p.Cout + ((lc ^ prA) * 8), prA * p.Cout + ((lc ^ (prA | 8)) * 8)};
        const int lparB[2] = {prB * p.Cout + ((lc ^ prB) * 8), prB * p.Cout + ((lc ^ (prB | 8)) * 8)};
        const int c8A[2] = {lc ^ prA, lc ^ (prA | 8)}, c8B[2] = {lc ^ prB, lc ^ (prB | 8)};
        auto tile_base = [&](const WsTile& t) -> size_t {
            return (((size_t)t.n0 * p.OH + t.oy0) * p.OW + t.ox0) * p.Cout + t.cot * 128;
        };
        auto piece_u = [&](int k) -> int { return ((k >> (TWl - 3)) * p.OW + (k & (TW / 8 - 1)) * 8) * p.Cout; };
        WsTile cur;
        tile_of(q, cur);
        if (p.residual != nullptr && !(WS_DBG(8))) {              // first tile's residual tile (later ones: the loaders)
            const bf16* rb = p.residual + tile_base(cur);
#pragma unroll 1
            for (int k = 0; k < 32; ++k) {
                const int u = piece_u(k);
                const void* ga = cur.cot * 128 + c8A[k & 1] * 8 < p.Cout ? (const void*)(rb + u + lparA[k & 1]) : (const void*)zero_page;
                const void* gb = cur.cot * 128 + c8B[k & 1] * 8 < p.Cout ? (const void*)(rb + u + lparB[k & 1]) : (const void*)zero_page;
                __builtin_amdgcn_global_load_lds(WS_GPTR(ga), WS_LPTR(ro + (k * 128) * 16), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(WS_GPTR(gb), WS_LPTR(ro + (k * 128 + 64) * 16), 16, 0, 0);
            }
        }
        ws_barrier();                                           // P0
        bool have_prev = false;
        WsTile prev = cur;
        const int kpc = (32 + nchunks - 2) / (nchunks - 1);
        auto piece_range = [&](int c, int t, int& ka, int& kb) {
            ka = kb = 32;
            if (!have_prev || c + 1 >= nchunks) return;
            const int k0 = c * kpc < 32 ? c * kpc : 32, k1 = k0 + kpc < 32 ? k0 + kpc : 32;
            const int i0 = t < 6 ? t : 6 + 2 * (t - 6), i1 = t < 6 ? t + 1 : i0 + 2;
            ka = k0 + i0 < k1 ? k0 + i0 : k1;
            kb = k0 + i1 < k1 ? k0 + i1 : k1;
        };
        const bool do_drain = !(WS_DBG(4));
        bf16* out_prev = reinterpret_cast<bf16*>(p.out);
        for (;;) {
            const bool more = q + qstride < ntiles;
            WsTile nxt = cur;
            if (more) tile_of(q + qstride, nxt);
            for (int c = 0; c < nchunks; ++c) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    int ka, kb;
                    piece_range(c, t, ka, kb);
                    if (ka < kb && do_drain) {
                        bf16x8 va[2], vb[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (ka + u < kb) {
                                va[u] = *reinterpret_cast<const bf16x8*>(ro + ((ka + u) * 128 + lane) * 16);
                                vb[u] = *reinterpret_cast<const bf16x8*>(ro + ((ka + u) * 128 + 64 + lane) * 16);
                            }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            if (ka + u < kb) {
                                const int k = ka + u, pu = piece_u(k);
                                if (prev.cot * 128 + c8A[k & 1] * 8 < p.Cout) *reinterpret_cast<bf16x8*>(out_prev + pu + lparA[k & 1]) = va[u];
                                if (prev.cot * 128 + c8B[k & 1] * 8 < p.Cout) *reinterpret_cast<bf16x8*>(out_prev + pu + lparB[k & 1]) = vb[u];
                            }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the piece is out of LDS before the group barrier
                    if (t % 3 == 2) { WS_VALU_RUN(); if (!(WS_DBG(16))) ws_barrier(); }            // end of a group step
                }
            }
            if (!have_prev) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // first tile: the residual tile of the prologue landed
            ws_barrier();                                        // E1
            ws_barrier();                                        // E2
            out_prev = reinterpret_cast<bf16*>(p.out) + tile_base(cur);
            prev = cur;
            have_prev = true;
            if (!more) break;
            q += qstride;
            cur = nxt;
        }
        WS_VALU_DONE();
    }

    // ==================================================================== last tile out: all eight waves
    // (every wave has passed the last E2; the two bulk movers alone needed ~3 us for the 64 KB)
    {
        const int qlast = q0 + ((ntiles - 1 - q0) / qstride) * qstride;
        WsTile lt;
        tile_of(qlast, lt);
        bf16* const ob = reinterpret_cast<bf16*>(p.out) + ((((size_t)lt.n0 * p.OH + lt.oy0) * p.OW + lt.ox0) * p.Cout + lt.cot * 128);
        const int t2 = tid & 127, kq = tid >> 7;      // piece L = k*128 + t2, k = 4 i + kq
        const int pr = t2 >> 4;
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(ro + (((i + u) * 4 + kq) * 128 + t2) * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (i + u) * 4 + kq;
                const int lp = k * 8 + pr;
                const int c8 = (t2 & 15) ^ (lp & 15);
                const int rel = ((lp >> TWl) * p.OW + (lp & (TW - 1))) * p.Cout + c8 * 8;
                if (lt.cot * 128 + c8 * 8 < p.Cout) *reinterpret_cast<bf16x8*>(ob + rel) = v[u];
            }
        }
        WS_CLOCKSTAMP(174, wave == 0);
    }
    if (blockIdx.x == 0 && tid == 0) {
        g_ws_clock[2] = __builtin_amdgcn_s_memtime();
        g_ws_clock[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int TW>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(ConvArgs p) {
    conv_ws_body<TW>(p);
}

}  // namespace

// Launches the wave-specialised kernel when the shape is in its scope; returns 1 otherwise (caller falls back).
int conv_ws_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id) {
    static const int enabled = getenv("DXMI_CONV_WS") ? atoi(getenv("DXMI_CONV_WS")) : 1;   // DXMI_CONV_WS=0: conv_pipe_kernel for every shape
    if (!enabled) return 1;
    if (a.in_mode != DXMI_IN_NHWC_BF16 || a.out_mode != DXMI_OUT_NHWC_BF16) return 1;
    // an activation mask (data gradient of the value net's convs: out *= mask_src > 0 ? 1 : slope) rides the residual tile's path
    // when there is no residual; both at once stay on conv_pipe_kernel
    if (a.ksize != 3 || a.stride != 1 || a.pad != 1 || a.ups == 2 || (a.mask_src && a.residual) || a.act == DXMI_ACT_SILU) return 1;
    if (a.gn_out) return 1;          // a fused GroupNorm of the output: conv_sm_kernel / conv_ws8_kernel only
    if (a.Cout % 64 != 0 || (a.C0 + a.C1) % 32 != 0 || a.C0 % 32 != 0) return 1;   // Cout % 128 == 64: the last cout tile is half empty
    const int TW = a.OW >= 32 ? 32 : a.OW;
    if (TW != 32 && TW != 16) return 1;
    const int TH = 256 / TW;
    if (a.OH % TH != 0 || a.OW % TW != 0) return 1;
    const int nchunks = (a.C0 + a.C1) / 32;
    if ((9 * nchunks) % WS_RING != 0 || nchunks % 2 != 0) return 1;   // ring slot / halo image of a step must not depend on the tile
    if (nchunks < 4) return 1;   // the tile switch is spread over nchunks-1 chunks at <= 12 pieces per chunk
    // fewer (256-pixel, 128-cout) tiles than ~a third of the CUs (the EDM nets' 16x16 maps at the train batch of 16: 80 tiles):
    // conv_pipe_kernel's 64-pixel tiles fill the chip better (576 -> 576 @16x16, B = 16: 48.1 us here, 41.1 us there; at 192
    // tiles this kernel is 40 % ahead)
    // (knob "conv_ws_min_tiles" / DXMI_CONV_WS_MIN_TILES; the kernel's own edge-case tests set it to 0)
    const int min_tiles = dxmi_tuning("conv_ws_min_tiles");
    if ((long)a.N * (a.OH / TH) * (a.OW / TW) * ((a.Cout + 127) / 128) < min_tiles) return 1;
    // 32-bit byte offsets inside either input part (the movers' per-tile source tables)
    if ((long)a.N * a.IH * a.IW * (a.C0 > a.C1 ? a.C0 : a.C1) * 2 >= (1L << 31)) return 1;
    if (kernel_id) {
        *kernel_id = 400000 + TW;    // conv_ws_kernel<TW>
        return DXMI_OK;
    }
    static const void* zero_page = nullptr;
    if (!zero_page) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(ws_zero16)) != hipSuccess || !zp) {
            dxmi_set_error("dxmi_conv2d_fwd(ws): hipGetSymbolAddress(ws_zero16) failed");
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
    b.SUBS = 1;
    b.PT = a.N * (a.OH / TH) * (a.OW / TW);
    b.CT = (a.Cout + 127) / 128;
    b.tile_px = 256;
    static const int xcd_env = getenv("DXMI_CONV_WS_XCD") ? atoi(getenv("DXMI_CONV_WS_XCD")) : 1;
    b.xcd_order = xcd_env;
#ifdef DXMI_CONV_STAMPS
    static const int dbg = getenv("DXMI_CONV_WS_DBG") ? atoi(getenv("DXMI_CONV_WS_DBG")) : 0;
    b.stagger = dbg;
#else
    b.stagger = 0;
#endif
    const size_t lds = 2 * WS_HALO + WS_A_RING + WS_RO + WS_TB;
    int grid = b.PT * b.CT;
    if (grid > 256) grid = 256;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ws_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (TW == 32) hipLaunchKernelGGL(conv_ws_kernel<32>, dim3(grid), dim3(512), lds, st, b);
    else hipLaunchKernelGGL(conv_ws_kernel<16>, dim3(grid), dim3(512), lds, st, b);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd(ws)");
    return DXMI_OK;
}
