// The CIFAR-10 U-Net's 16x16 AttnBlock (reference models/DxMI/unet_small.py:167-191) as ONE launch, inference only (round 5).
//
// Round 4 ran the block as three launches — GroupNorm apply (25 us), q|k|v 1x1 conv 256 -> 768 (40 us), attention + proj_out +
// residual (45 us) — moving ~365 MB per block at 256 images: the normalised input and q|k|v are written to HBM only to be read
// straight back.  Neither needs to exist.  With x^ = A (.) x + B the block's GroupNorm (per-image, per-channel affine from the
// producer's block statistics; no SiLU in this block) and 1x1 convs q = Wq x^ + bq, k = Wk x^ + bk, v = Wv x^ + bv:
//
//   logits   S_ij = scale q_i . k_j = x^_i^T (scale Wq^T Wk) x^_j + (scale Wk^T bq) . x^_j + [terms constant in j]
//                 = Y~_i . x_j + [constant in j],      Y~_i = A (.) (G x^_i + g),  G = scale log2(e) Wk^T Wq,  g = scale log2(e) Wk^T bq
//            (terms constant along the key axis drop out of the softmax; the key-side affine moves to the query side: K = RAW x)
//   output   out_i = x_i + bproj + Wproj sum_j P_ij (Wv x^_j + bv) = x_i + b' + W' Z_i
//            Z_i = A (.) (sum_j P_ij x_j) + B   (rows of P sum to one: V = RAW x too),  W' = Wproj Wv,  b' = bproj + Wproj bv
//
// so the raw input x (256 tokens x 256 channels bf16 = 128 KB) is the only activation the block needs: it stays in LDS as K, as V
// and as the residual, the two 256 x 256 products G and W' are folded and packed once per weight version (attn_block_fold_kernel),
// and a block is 4 instead of 6 GEMMs per image and 67 MB of HBM traffic per launch at 256 images.
//
// One 512-thread workgroup per image, wave w owns queries 32w .. 32w+31 (as attention256_kernel):
//   prologue  wave w brings ITS 32 rows of x by DMA (16 x 1 KiB, rows XOR-swizzled like attention256_kernel's blocks); while they
//             land the (A, B) table of the image is formed from the block statistics in gn_apply_kernel's order (pairs over the
//             partials, groups over their pairs: DPP quad broadcasts, no LDS round trip; the table is a static LDS object stored by
//             inline asm so that hipcc does not order it behind the DMAs); x^ of the wave's rows is formed in registers (the only
//             place it ever exists);
//   phase 1   Y^T[c][query] = G x^^T: G's 128 one-KiB fragments stream L2 -> registers (a queue of five slots per wave) -> a ring
//             of three 8 KiB LDS slots, wave w moving fragment w of every slot, one barrier per slot; a slot's fragments are read
//             into registers one step before their MFMAs, which lead each step with the ring traffic interleaved.  G's rows are
//             packed in the order that makes the accumulators of a 32-channel block the B-operand fragments of phase 2 as they
//             stand (lane half h: channels 16 j + 8 h .. + 7 of k-step j);
//   phase 2   S^T[key][query] = x Y~^T against all 256 keys in LDS (no ring, no barrier), plain two-pass softmax in the log2 domain;
//   phase 3   O^T[c][query] = x^T P^T through transposing LDS reads (attention256_kernel's scheme), Z = A (.) O / l + B;
//   phase 4   y^T[cout][query] = W' Z^T, W' through the same ring; out = y + b' + x is formed IN PLACE in the wave's own rows of
//             the x image (nobody else reads them after phase 3), whole 512-byte rows stored; optional GroupNorm block statistics
//             of the result (one partial per wave = 32 tokens) for the next Normalize().
// Rounding points: x^ (bf16, as the GroupNorm launch stored it), Y~, P, Z (bf16 MFMA operands, fp32 accumulation), out (one
// rounding).  q, k, v and the attention output are never rounded on their own: against the fp32 reference the block is no further
// than the three-launch form (tests/test_hip_round5_kernels.py).
#include "conv_common.h"
#include <stdlib.h>

#ifdef DXMI_AB_STAMPS
// timing-only build (tools/build_variant.sh ... -DDXMI_AB_STAMPS): s_memtime of every wave of the first 16 workgroups at the phase
// boundaries, kept in LDS until the kernel's end (stamps that address global memory cost the kernel its registers)
__device__ unsigned g_ab_stamps[16][8][12];
extern "C" int dxmi_debug_read_ab_stamps(void* dst, int bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ab_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#define AB_STAMP(k) do { reinterpret_cast<unsigned*>(tabs + 1024)[wave * 12 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)   /* every lane, same value: no branch */
#define AB_STAMP_WORDS 96
#define AB_RSTAMP(k) do { reinterpret_cast<unsigned*>(tabs + 1024)[wave * 12 + (k)] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)
#define AB_STAMP_DUMP() do { if (lane < 12 && blockIdx.x < 16) g_ab_stamps[blockIdx.x][wave][lane] = reinterpret_cast<unsigned*>(tabs + 1024)[wave * 12 + lane]; } while (0)
#else
#define AB_STAMP_WORDS 0
#define AB_STAMP(k) do {} while (0)
#define AB_RSTAMP(k) do {} while (0)
#define AB_STAMP_DUMP() do {} while (0)
#endif

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define AB_LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))
#define AB_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AB_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AB_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void ab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int ab_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// one ring step behind its barrier: MFMA, ring write + queue refill, then MFMA / ring read pairs
#define AB_INTERLEAVE_STEP()                                    \
    do {                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      \
        _Pragma("unroll") for (int i_ = 0; i_ < 7; ++i_) {      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
        }                                                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
    } while (0)

constexpr int AB_X = 256 * 512;            // the image: 256 tokens x 512 B
constexpr int AB_SLOT = 8192;              // ring slot: 8 weight fragments
constexpr int AB_RING = 3 * AB_SLOT;
constexpr int AB_Q = 5;                    // register-staged slots in front of the ring
constexpr int AB_LDS = AB_X + AB_RING;             // dynamic part; + 4 KB static: tables A | B | A g | b'
constexpr float AB_LOG2E = 1.4426950408889634f;

struct AttnBlockArgs {
    const bf16* x;          // [N][256][256]
    const float* stats;     // [N][P][128][2] block statistics of x
    const float* gamma;
    const float* beta;
    const bf16* wG;         // folded + packed: G fragments [8 cb][16 ks][64 lanes][8]
    const bf16* wP;         // W' fragments [8 cb][16 j][64 lanes][8] (pack_attn_proj order)
    const float* gvec;      // [256] g
    const float* bprime;    // [256] b'
    bf16* out;              // [N][256][256]
    float* out_stats;       // optional [N][8][128][2]
    int N, P;
    float eps;
};

// Row m (0..31) of the A operand of cout block cb holds channel ab_chan(cb, m): the 32x32x16 accumulator of lane half h then
// carries, in registers 0..7 / 8..15, channels 32 cb + 8 h .. + 7 / 32 cb + 16 + 8 h .. + 7 — the B fragments of k-steps 2 cb, 2 cb + 1.
__host__ __device__ inline int ab_chan(int cb, int m) {
    const int g = m >> 3, h = (m >> 2) & 1, e = m & 3;
    return 32 * cb + 16 * (g >> 1) + 8 * h + 4 * (g & 1) + e;
}

// Fold + pack, one thread per packed bf16 (2 x 65536) plus the two bias vectors; fp32 dot products of 256 terms in index order.
//   wq, wk, wv, wp: [256 cout][256 cin] fp32 (1x1 conv weights), bq, bv, bp: [256]
__global__ __launch_bounds__(256) void attn_block_fold_kernel(const float* __restrict__ wq, const float* __restrict__ bq,
                                                              const float* __restrict__ wk, const float* __restrict__ wv,
                                                              const float* __restrict__ bv, const float* __restrict__ wp,
                                                              const float* __restrict__ bp, float scale, bf16* __restrict__ dG,
                                                              bf16* __restrict__ dP, float* __restrict__ gvec, float* __restrict__ bprime) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const float s2 = scale * AB_LOG2E;
    if (t < 65536) {                    // G fragment (cb, ks), lane l, element i = G[chan(cb, l & 31)][16 ks + 8 (l >> 5) + i]
        const int i = t & 7, l = (t >> 3) & 63, ks = (t >> 9) & 15, cb = t >> 13;
        const int m = ab_chan(cb, l & 31), c = 16 * ks + 8 * (l >> 5) + i;
        float a = 0.f;
        for (int o = 0; o < 256; ++o) a = __builtin_fmaf(wk[o * 256 + m], wq[o * 256 + c], a);
        dG[t] = (bf16)(a * s2);
    } else if (t < 131072) {            // W' fragment (cb, j), lane l, element i = W'[32 cb + (l & 31)][16 j + 8 (i >> 2) + 4 (l >> 5) + (i & 3)]
        const int u = t - 65536;
        const int i = u & 7, l = (u >> 3) & 63, j = (u >> 9) & 15, cb = u >> 13;
        const int o = 32 * cb + (l & 31), c = 16 * j + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
        float a = 0.f;
        for (int m = 0; m < 256; ++m) a = __builtin_fmaf(wp[o * 256 + m], wv[m * 256 + c], a);
        dP[u] = (bf16)a;
    } else if (t < 131072 + 256) {      // g[m] = s2 sum_o Wk[o][m] bq[o]
        const int m = t - 131072;
        float a = 0.f;
        for (int o = 0; o < 256; ++o) a = __builtin_fmaf(wk[o * 256 + m], bq[o], a);
        gvec[m] = a * s2;
    } else if (t < 131072 + 512) {      // b'[o] = bp[o] + sum_m Wp[o][m] bv[m]
        const int o = t - 131072 - 256;
        float a = 0.f;
        for (int m = 0; m < 256; ++m) a = __builtin_fmaf(wp[o * 256 + m], bv[m], a);
        bprime[o] = bp[o] + a;
    }
}

__global__ __launch_bounds__(512, 1) void attn_block256_kernel(AttnBlockArgs p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // the tables are a separate (static) LDS object: hipcc orders every LDS access that MAY alias an LDS-DMA in flight behind
    // s_waitcnt vmcnt(0); stores into the dynamic block would wait for the whole image before the table could be formed
    __shared__ __attribute__((aligned(16))) float tabs[1024 + AB_STAMP_WORDS];
    char* const ring = smem + AB_X;
    float* const tabA = tabs;
    float* const tabB = tabA + 256;
    float* const tabG = tabA + 512;
    float* const tabP = tabA + 768;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int h = lane >> 5, ql = lane & 31;
    const int drow = lane >> 5, dslot = lane & 31;
    const bf16* const xn = p.x + (size_t)n * 65536;

    AB_STAMP(0);
    AB_RSTAMP(10);
    // ---- prologue.  The few small loads of the (A, B) table go first: vector-memory results return in order, so they are back
    // long before the image, and the table is formed (two LDS hand-overs, barriers that do NOT drain the vector-memory queue)
    // while the image is still landing.
    // (inline-asm loads with one counted wait: hipcc's own wait for a load that is older than LDS-DMAs in flight is vmcnt(0), and
    // so is its wait in front of any LDS access that may alias an LDS-DMA destination — the table is stored by inline asm too.)
    // Thread t < 128 owns channel pair t: its statistics partials, gamma / beta / g / b' of its two channels.
    // (64-bit scalars, not 2-vectors: hipcc miscompiles element 1 of a vector-typed inline-asm output — it reads element 0)
    uint64_t t[8], gab[4];
    {
        const int pt = tid & 127;
        const float2* const sb = reinterpret_cast<const float2*>(p.stats) + (size_t)n * p.P * 128 + pt;
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t[k]) : "v"(sb + (k < p.P ? k : 0) * 128) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gab[0]) : "v"(p.gamma + 2 * pt) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gab[1]) : "v"(p.beta + 2 * pt) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gab[2]) : "v"(p.gvec + 2 * pt) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gab[3]) : "v"(p.bprime + 2 * pt) : "memory");
    }
    // this wave's 32 rows of x (16 DMAs of two rows), then the first two ring slots of G
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = wave * 32 + 2 * i + drow;
        const bf16* g = xn + (size_t)r * 256 + ((dslot ^ ab_swz(r)) << 3);
        __builtin_amdgcn_global_load_lds(AB_GPTR(g), AB_LPTR(smem + (wave * 32 + 2 * i) * 512), 16, 0, 0);
    }
    // Weight streams (G in phase 1, W' in phase 4): 16 slots of 8 fragments through a ring of three 8 KiB LDS slots, wave w moves
    // fragment w of every slot.  An LDS-DMA ring keeps 16 KB in flight against ~1 900 cycles of L2 -> LDS latency: 8.6 B / clk of the
    // 16 B / clk the MFMAs consume (phases 1 and 4 ran at 54 % of the matrix pipe, at 16 as at 256 workgroups: latency, not L2
    // bandwidth).  Register staging adds a queue of AB_Q slots per wave IN FRONT of the ring: slot s is requested (plain
    // global_load_dwordx4, 4 VGPRs) AB_Q + 2 steps before it is read, written to the ring two steps before.
    u32x4 gq[AB_Q];
    auto load_w = [&](const bf16* w, int s) {           // fragment 8 s + wave of the stream -> queue entry s % AB_Q
        gq[s % AB_Q] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(w) + (size_t)(s * 8 + wave) * 1024 + lane * 16);
    };
    auto store_w = [&](int s) {                         // queue entry -> ring slot s % 3
        *reinterpret_cast<u32x4*>(ring + (s % 3) * AB_SLOT + wave * 1024 + lane * 16) = gq[s % AB_Q];
    };
#pragma unroll
    for (int s0 = 0; s0 < AB_Q; ++s0) load_w(p.wG, s0);
    // (A, B) of the image from the block statistics, in gn_apply_kernel's order: a pair's partials in partial order, a group's
    // four pairs in channel order, mean / rstd, A = rstd gamma, B = beta - mean A
    // the 12 small loads are older than the 16 DMAs and the AB_Q weight loads: complete when at most 16 + AB_Q operations are
    // outstanding (the operands tie the values to the wait: an ALU use must not be scheduled in front of it)
    static_assert(AB_Q == 5, "the counted waits below");
    asm volatile("s_waitcnt vmcnt(21)"
                 : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(gab[0]), "+v"(gab[1]),
                   "+v"(gab[2]), "+v"(gab[3])
                 :: "memory");
    {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < p.P) {
                s += __builtin_bit_cast(float, (uint32_t)t[k]);
                q += __builtin_bit_cast(float, (uint32_t)(t[k] >> 32));
            }
        // a group = four consecutive pairs = the four lanes of a DPP quad: every lane adds the quad's four sums in lane order
        // (bitwise the sequential sum gn_apply_kernel forms), no LDS round trip
        auto quad = [&](float v) {
            const int vi = __builtin_bit_cast(int, v);
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0x00, 0xf, 0xf, true));
            const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0x55, 0xf, 0xf, true));
            const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0xAA, 0xf, 0xf, true));
            const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0xFF, 0xf, 0xf, true));
            return ((v0 + v1) + v2) + v3;
        };
        const float sg = quad(s), qg = quad(q);
        const float cnt = 256.f * 8.f;
        const float m = sg / cnt;
        const float rstd = rsqrtf(fmaxf(qg / cnt - m * m, 0.f) + p.eps);
        float a[2], b[2], ag[2], bp[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            a[e] = rstd * __builtin_bit_cast(float, (uint32_t)(gab[0] >> (32 * e)));
            b[e] = __builtin_bit_cast(float, (uint32_t)(gab[1] >> (32 * e))) - m * a[e];
            ag[e] = a[e] * __builtin_bit_cast(float, (uint32_t)(gab[2] >> (32 * e)));
            bp[e] = __builtin_bit_cast(float, (uint32_t)(gab[3] >> (32 * e)));
        }
        auto pack2 = [](float lo, float hi) { return (uint64_t)__builtin_bit_cast(uint32_t, lo) | ((uint64_t)__builtin_bit_cast(uint32_t, hi) << 32); };
        if (tid < 128) {
            const unsigned ta = (unsigned)reinterpret_cast<uintptr_t>(AB_LPTR(tabA + 2 * tid));
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:1024\n\tds_write_b64 %0, %3 offset:2048\n\tds_write_b64 %0, %4 offset:3072"
                         :: "v"(ta), "v"(pack2(a[0], a[1])), "v"(pack2(b[0], b[1])), "v"(pack2(ag[0], ag[1])), "v"(pack2(bp[0], bp[1])) : "memory");
        }
    }
    AB_STAMP(1);
    ab_barrier();
    AB_WAIT_VM(0);          // this wave's rows of x have landed (hipcc may order the AB_Q weight loads among the DMAs: no count is safe but 0)

    // ---- x^ of this wave's queries: B operand of phase 1 (lane = query, channels 16 ks + 8 h .. + 7), rounded to bf16 exactly
    // where the GroupNorm launch stored it
    bf16x8 qf[16];
    {
        const int r = wave * 32 + ql;
        const char* xrow = smem + r * 512;
        const int sw = ab_swz(r);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(xrow + (((ks * 2 + h) ^ sw) << 4));
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(tabA + ks * 16 + 8 * h), a1 = *reinterpret_cast<const f32x4*>(tabA + ks * 16 + 8 * h + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(tabB + ks * 16 + 8 * h), b1 = *reinterpret_cast<const f32x4*>(tabB + ks * 16 + 8 * h + 4);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (bf16)((float)raw[e] * a0[e] + b0[e]);
                v[4 + e] = (bf16)((float)raw[4 + e] * a1[e] + b1[e]);
            }
            qf[ks] = v;
        }
    }

    AB_STAMP(2);
    // ---- phase 1: Y^T = G x^^T, cout block by cout block (two ring slots each); Y~ = A (.) (Y + g) replaces x^ in qf as it completes
    store_w(0);
    store_w(1);
    load_w(p.wG, AB_Q);
    load_w(p.wG, AB_Q + 1);
    bf16x8 yq[16];
    bf16x8 fr[2][8];        // the fragments of ring slot i are read into registers while the MFMAs of slot i - 1 run
    auto read_slot = [&](int i, bf16x8 (&f)[8]) {
        const char* const slot = ring + (i % 3) * AB_SLOT + lane * 16;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) f[jj] = *reinterpret_cast<const bf16x8*>(slot + jj * 1024);
    };
    {
        f32x16 y;
#pragma unroll
        for (int i = 0; i <= 16; ++i) {
            // (hipcc treats the asm barrier as a fence for memory operations only and sank the MFMAs of every other step below the
            // NEXT barrier: 1 MFMA between two barriers, 15 behind the second — the matrix pipe idled through every other barrier)
            __builtin_amdgcn_sched_barrier(0);
            if (i < 16) {
                ab_barrier();                   // slot i is in the ring (written one step ago at the latest); everybody has slot i - 1 in registers
                if (i + 2 < 16) store_w(i + 2);
                if (i + 2 + AB_Q < 16) load_w(p.wG, i + 2 + AB_Q);
                read_slot(i, fr[i & 1]);
            }
            if (i > 0) {
                const int j = i - 1;
                if ((j & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.f;
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 1][jj], qf[(j & 1) * 8 + jj], y, 0, 0, 0);
                AB_INTERLEAVE_STEP();           // the MFMAs (operands in registers since the last step) lead, this step's LDS / memory issue rides between them
                if (j & 1) {
                    const int cb = j >> 1;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int c0 = 32 * cb + 16 * hf + 8 * h;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(tabA + c0), a1 = *reinterpret_cast<const f32x4*>(tabA + c0 + 4);
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(tabG + c0), g1 = *reinterpret_cast<const f32x4*>(tabG + c0 + 4);
                        bf16x8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (bf16)(y[8 * hf + e] * a0[e] + g0[e]);
                            v[4 + e] = (bf16)(y[8 * hf + 4 + e] * a1[e] + g1[e]);
                        }
                        yq[2 * cb + hf] = v;
                    }
                }
            }
        }
    }
    AB_STAMP(3);
    ab_barrier();           // every wave's rows of x have landed (each waited for its own before phase 1) and the ring is free

    // ---- phase 2: S^T[key][query] = x . Y~^T, 8 tiles of 32 keys; logits arrive in the log2 domain (scale log2 e folded into G)
    f32x16 s[8];
    {
        const int ksw = ab_swz(ql);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x16 a16;
#pragma unroll
            for (int r = 0; r < 16; ++r) a16[r] = 0.f;
            const char* rowp = smem + (t * 32 + ql) * 512;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 2 + h) ^ ksw) << 4));
                a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, yq[ks], a16, 0, 0, 0);
            }
            s[t] = a16;
        }
    }
    AB_STAMP(4);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    bf16x8 pb[16];      // P^T fragments: k-step st covers keys 16 st .. + 15 in accumulator order
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
            l += e;
            pb[t * 2 + (r >> 3)][r & 7] = (bf16)e;
        }
    l += __shfl_xor(l, 32, 64);

    AB_STAMP(5);
    // ---- phase 3: O^T[c][query] = x^T . P^T (V = raw x: the affine is applied to the 256 sums instead of the 65536 values)
    f32x16 o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    {
        // tr-read lane roles (ds_read_b64_tr_b16): 16-lane group g -> channel half (g & 1), key half (g >> 1); lane 4q+pp -> key row q
        const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
        const int tr_krow = 4 * (trg >> 1) + trq;
        const int tr_c = 2 * (trg & 1) + (trp >> 1), tr_sub = 8 * (trp & 1);
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int r0 = 16 * st + tr_krow, r1 = r0 + 8;
            const char* row0 = smem + r0 * 512 + tr_sub;
            const char* row1 = smem + r1 * 512 + tr_sub;
            const int sw0 = ab_swz(r0), sw1 = ab_swz(r1);
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(AB_LDS_S16X4(row0 + (((db * 4 + tr_c) ^ sw0) << 4)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(AB_LDS_S16X4(row1 + (((db * 4 + tr_c) ^ sw1) << 4)));
                bf16x8 a;
                short* as = reinterpret_cast<short*>(&a);
#pragma unroll
                for (int e = 0; e < 4; ++e) { as[e] = lo[e]; as[4 + e] = hi[e]; }
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[st], o[db], 0, 0, 0);
            }
        }
    }
    AB_STAMP(6);
#pragma unroll
    for (int s0 = 0; s0 < AB_Q; ++s0) load_w(p.wP, s0);        // W' slots 0 .. AB_Q - 1 fly under the Z affine
    // Z = A (.) O / l + B -> B operand of phase 4: k-step j takes registers 8 (j & 1) .. + 7 of o[j >> 1] as they stand
    // (channels 16 j + 8 (i >> 2) + 4 h + (i & 3)); W' is packed in the same channel order
    bf16x8 of[16];
    {
        const float inv = 1.f / l;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c0 = 16 * j + 8 * g + 4 * h;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(tabA + c0), b4 = *reinterpret_cast<const f32x4*>(tabB + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) of[j][4 * g + e] = (bf16)((o[j >> 1][8 * (j & 1) + 4 * g + e] * inv) * a4[e] + b4[e]);
            }
    }
    AB_STAMP(7);
    store_w(0);
    store_w(1);
    load_w(p.wP, AB_Q);
    load_w(p.wP, AB_Q + 1);
    ab_barrier();           // every wave is done with x as K and V: a wave's own rows become its output tile

    // ---- phase 4: y^T[cout][query] = W' Z^T; out = y + b' + x in place in the wave's rows of the image
    {
        const int r = wave * 32 + ql, sw = ab_swz(r);
        char* const orow = smem + r * 512 + 8 * h;
        f32x16 y;
#pragma unroll
        for (int i = 0; i <= 16; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            if (i < 16) {
                ab_barrier();
                if (i + 2 < 16) store_w(i + 2);
                if (i + 2 + AB_Q < 16) load_w(p.wP, i + 2 + AB_Q);
                read_slot(i, fr[i & 1]);
            }
            if (i > 0) {
                const int j = i - 1, cb = j >> 1;
                if ((j & 1) == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) y[q] = 0.f;
                }
                // the residual pieces and the bias of this cout block are requested in front of the MFMAs that finish it
                bf16x4 rv[4];
                f32x4 bv[4];
                if (j & 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        rv[g] = *reinterpret_cast<const bf16x4*>(orow + (((cb * 4 + g) ^ sw) << 4));
                        bv[g] = *reinterpret_cast<const f32x4*>(tabP + cb * 32 + g * 8 + 4 * h);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 1][jj], of[(j & 1) * 8 + jj], y, 0, 0, 0);
                AB_INTERLEAVE_STEP();
                if (j & 1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 ov;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = (bf16)(y[4 * g + e] + (bv[g][e] + (float)rv[g][e]));
                        *reinterpret_cast<bf16x4*>(orow + (((cb * 4 + g) ^ sw) << 4)) = ov;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows a wave drains are the rows it wrote
    AB_STAMP(8);
    // ---- whole-row stores of the wave's 32 rows (+ their block statistics)
    bf16* const obase = p.out + (size_t)n * 65536;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(smem + (wave * 32 + 2 * (i + u)) * 512 + lane * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int r = wave * 32 + 2 * (i + u) + drow;
            asm volatile("" : "+v"(r));       // (keeps hipcc from forming the 16 store addresses in the prologue and spilling them)
            *reinterpret_cast<bf16x8*>(obase + (size_t)r * 256 + ((dslot ^ ab_swz(r)) << 3)) = v[u];
        }
    }
    if (p.out_stats) {
        // lane (16-byte piece sp, row group rg of 16 rows) adds its piece over its rows (even and odd rows in separate sums, then in
        // a fixed order), the two row groups are added across lane halves; one partial per wave
        const int sp = lane & 31, rg = lane >> 5;
        float st[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[k][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int rr = wave * 32 + rg * 16 + t;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + rr * 512 + ((sp ^ ab_swz(rr)) << 4));
            const bf16x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
            dxmi_stats4(lo, st[t & 1 ? 2 : 0]);
            dxmi_stats4(hi, st[t & 1 ? 3 : 1]);
        }
        float tot[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tot[e] = st[0][e] + st[2][e];
            tot[4 + e] = st[1][e] + st[3][e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tot[e] += __shfl_xor(tot[e], 32, 64);
        if (lane < 32) {
            float* sd = p.out_stats + (((size_t)n * 8 + wave) * 128 + sp * 4) * 2;
            *reinterpret_cast<f32x4*>(sd) = f32x4{tot[0], tot[1], tot[2], tot[3]};
            *reinterpret_cast<f32x4*>(sd + 4) = f32x4{tot[4], tot[5], tot[6], tot[7]};
        }
    }
    AB_STAMP(9);
    AB_RSTAMP(11);
    AB_STAMP_DUMP();
}

}  // namespace

extern "C" int dxmi_attn_block_supported(int32_t T, int32_t C, int32_t heads, int32_t groups) {
    return T == 256 && C == 256 && heads == 1 && groups == 32 ? 1 : 0;
}

extern "C" int64_t dxmi_attn_block_packed_bytes(void) { return 2 * 65536 * 2 + 2 * 256 * 4; }

extern "C" int dxmi_attn_block_pack(const float* wq, const float* bq, const float* wk, const float* wv, const float* bv,
                                    const float* wproj, const float* bproj, float scale, void* dst, void* stream) {
    DXMI_CHECK_ARG(wq && bq && wk && wv && bv && wproj && bproj && dst, "dxmi_attn_block_pack: null pointer");
    char* d = reinterpret_cast<char*>(dst);
    hipLaunchKernelGGL(attn_block_fold_kernel, dim3(514), dim3(256), 0, (hipStream_t)stream, wq, bq, wk, wv, bv, wproj, bproj, scale,
                       reinterpret_cast<bf16*>(d), reinterpret_cast<bf16*>(d + 131072), reinterpret_cast<float*>(d + 262144),
                       reinterpret_cast<float*>(d + 262144 + 1024));
    DXMI_CHECK_LAUNCH("dxmi_attn_block_pack");
    return DXMI_OK;
}

extern "C" int dxmi_attn_block_fwd(const void* x, const float* stats, int32_t P, const float* gamma, const float* beta, float eps,
                                   const void* packed, void* out, float* out_stats, int32_t N, int32_t T, int32_t C, void* stream) {
    DXMI_CHECK_ARG(x && stats && gamma && beta && packed && out, "dxmi_attn_block_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && P > 0 && P <= 8 && dxmi_attn_block_supported(T, C, 1, 32),
                   "dxmi_attn_block_fwd: only the single-head 256-token x 256-channel block, at most 8 statistics partials per image "
                   "(dxmi_gn_stats_fold folds more) (N=%d T=%d C=%d P=%d)", N, T, C, P);
    AttnBlockArgs a;
    const char* pk = reinterpret_cast<const char*>(packed);
    a.x = (const bf16*)x; a.stats = stats; a.gamma = gamma; a.beta = beta;
    a.wG = reinterpret_cast<const bf16*>(pk); a.wP = reinterpret_cast<const bf16*>(pk + 131072);
    a.gvec = reinterpret_cast<const float*>(pk + 262144); a.bprime = reinterpret_cast<const float*>(pk + 262144 + 1024);
    a.out = (bf16*)out; a.out_stats = out_stats; a.N = N; a.P = P; a.eps = eps;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(attn_block256_kernel, dim3(N), dim3(512), (size_t)AB_LDS, (hipStream_t)stream, a);
    DXMI_CHECK_LAUNCH("dxmi_attn_block_fwd");
    return DXMI_OK;
}
