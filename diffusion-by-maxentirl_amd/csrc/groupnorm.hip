// GroupNorm (+SiLU) over NHWC bf16 for gfx950 — the HBM-bound kernel of the U-Net.
//
// One workgroup owns one (image, channel-slice): every thread loads all of its 8- or
// 16-byte pieces up front (whole slice resident in VGPRs, >= 16 KB in flight per wave), so
// HBM sees exactly one read and one write of the tensor.  Statistics are two-pass in fp32
// (mean, then centred sum of squares) from registers: wavefront __shfl_xor reduction, then a
// fixed-order cross-wave sum through LDS, so results are bitwise reproducible.
// The input may be a virtual channel concat [in0 | in1] (U-Net skip connections); the
// output is always one dense NHWC tensor.
//
// Reference: Normalize()/nonlinearity, models/DxMI/unet_small.py:30-36,119-126,169,329-330;
// GroupNorm32, models/cm/nn.py:19-21.
#include "conv_common.h"
#include <stdlib.h>

namespace {

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.  The channel-sliced
// kernels below give the slices of ONE image consecutive block ids — 8-byte or 16-byte pieces of the same 64-byte lines — so each
// line was fetched by up to `slices` different L2s (gn_silu_bwd_kernel<4, 16>: 474 MB of HBM-side traffic per launch against 201 MB
// algorithmic, round-5 PMC pass).  This maps the hardware block id to a LOGICAL id such that the blocks of one XCD are consecutive
// logical ids: the slices of an image then share an L2 (any grid size: XCD j owns G / 8 + (j < G % 8) blocks).  Measured (same box,
// alternating, tools/gn_bwd_time.py): backward 256 x 32x32x128 104.4 -> 91.3 us, 128 + 128 concat 175.3 -> 159.4 us; shapes with one
// slice per image unchanged.
__device__ __forceinline__ int xcd_logical_block(int b, int G, int on = 1) {
    if (!on) return b;
    const int x = b & 7, k = b >> 3;
    const int q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + k;
}

struct GnArgs {
    const bf16* in0;
    const bf16* in1;
    const float* gamma;
    const float* beta;
    bf16* out;
    int C0, C1, HW, groups, slices, ppp, cpg, gps;  // ppp = pieces per pixel in slice, gps = groups per slice
    int xcd;             // XCD-aware block order (DXMI_GN_XCD=0: hardware order, A/B timing)
    float eps;
    int silu;
    int fast;  // xor-shuffle group reduction applies
};

template <int VEC>
struct PieceT;
template <>
struct PieceT<8> { typedef bf16x8 type; };
template <>
struct PieceT<4> { typedef bf16x4 type; };

// VEC channels per piece; PIECES pieces per thread; blockDim.x is a multiple of lcm(ppp,64)
template <int VEC, int PIECES>
__global__ __launch_bounds__(512) void gn_silu_kernel(GnArgs p) {
    typedef typename PieceT<VEC>::type piece_t;
    __shared__ float red[512 / 64][32];  // [wave][group in slice] partial sums
    __shared__ float stat[32];

    const int C = p.C0 + p.C1;
    const int bid = xcd_logical_block(blockIdx.x, gridDim.x, p.xcd);      // the slices of an image on ONE XCD (one L2)
    const int n = bid / p.slices;
    const int s = bid % p.slices;
    const int Csl = C / p.slices;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const int pc = tid % p.ppp;           // constant per thread since nthr % ppp == 0
    const int prow = tid / p.ppp;
    const int rows_per_iter = nthr / p.ppp;
    const int c = s * Csl + pc * VEC;     // first channel of this thread's pieces
    const int gl = (pc * VEC) / p.cpg;    // group index within the slice
    const bool from0 = c < p.C0;
    const bf16* src = from0 ? p.in0 : p.in1;
    const int Cs = from0 ? p.C0 : p.C1;
    const int cs = from0 ? c : c - p.C0;

    // The whole slice stays resident as RAW bf16 pairs; every pass re-expands them (2 VALU per
    // pair).  The empty asm keeps hipcc from hoisting the fp32 expansions out of the passes, which
    // would triple the register footprint.
    constexpr int W = VEC / 2;
    uint32_t v[PIECES][W];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
            const piece_t t = *reinterpret_cast<const piece_t*>(src + ((size_t)n * p.HW + px) * Cs + cs);
#pragma unroll
            for (int e = 0; e < W; ++e) v[q][e] = reinterpret_cast<const uint32_t*>(&t)[e];
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) v[q][e] = 0u;
        }
    }
#define GN_LO(w) __uint_as_float((w) << 16)
#define GN_HI(w) __uint_as_float((w) & 0xffff0000u)
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
#pragma unroll
        for (int e = 0; e < W; ++e) {
            asm volatile("" : "+v"(v[q][e]));
            sum += GN_LO(v[q][e]) + GN_HI(v[q][e]);  // rows past HW hold zeros
        }
    }
    const float inv_cnt = 1.f / (float)(p.HW * p.cpg);

    // ---- block reduction per group, fixed order (bitwise reproducible).
    // Fast path (ppp and pieces-per-group powers of two, ppp <= 64): lanes l, l+ppp, ... share a
    // channel piece and ppt adjacent pieces share a group -> xor-shuffles; otherwise a masked
    // wave sum per group.
    const int ppt = p.cpg / VEC;  // pieces per group per pixel
    auto group_reduce = [&](float val) -> float {
        if (p.fast) {
            float t = val;
            for (int o = 32; o >= p.ppp; o >>= 1) t += __shfl_xor(t, o, 64);
            for (int o = ppt >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane < p.ppp && (lane % ppt) == 0) red[wave][gl] = t;
        } else {
            for (int g = 0; g < p.gps; ++g) {
                float t = wave_sum(gl == g ? val : 0.f);
                if (lane == 0) red[wave][g] = t;
            }
        }
        __syncthreads();
        if (tid < p.gps) {
            float t = 0.f;
            for (int w = 0; w < nwaves; ++w) t += red[w][tid];
            stat[tid] = t;
        }
        __syncthreads();
        const float r = stat[gl];
        __syncthreads();
        return r;
    };

    const float mean = group_reduce(sum) * inv_cnt;
    float ssq = 0.f;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(v[q][e]));
                const float d0 = GN_LO(v[q][e]) - mean, d1 = GN_HI(v[q][e]) - mean;
                ssq += d0 * d0 + d1 * d1;
            }
        }
    }
    const float var = group_reduce(ssq) * inv_cnt;
    const float rstd = rsqrtf(var + p.eps);

    float ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        ga[e] = p.gamma[c + e] * rstd;
        be[e] = p.beta[c + e] - mean * ga[e];
    }
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
            piece_t o;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(v[q][e]));
                float y0 = GN_LO(v[q][e]) * ga[2 * e] + be[2 * e];
                float y1 = GN_HI(v[q][e]) * ga[2 * e + 1] + be[2 * e + 1];
                if (p.silu) {
                    y0 = dxmi_silu_fast(y0);
                    y1 = dxmi_silu_fast(y1);
                }
                o[2 * e] = (bf16)y0;
                o[2 * e + 1] = (bf16)y1;
            }
            *reinterpret_cast<piece_t*>(p.out + ((size_t)n * p.HW + px) * C + c) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic two-kernel GroupNorm(+SiLU) for shapes the register-resident kernel cannot hold (EDM U-Net:
// 6/18/30/42 channels per group, 64x64 and larger maps): (1) per-(image, row-chunk) partial sums of every
// group from fully coalesced 16-byte row pieces; (2) apply: each workgroup re-reduces its image's partials
// in a fixed order, then normalises its rows (second read mostly served by L2 / Infinity Cache).
// Optional per-image scale/shift (ADM "scale-shift norm", models/cm/unet.py:252-256):
//   y = (xh*gamma + beta) * (1 + scale[n,c]) + shift[n,c]   then SiLU.
// row chunks per image of the generic kernels: ~256 pixel rows per workgroup.  The count depends on the map only, NOT on the
// batch: the partial sums of an image are added in chunk order, and round 2's "more chunks while the launch is small" rule made
// an image's statistics (hence its output bits) depend on the batch it rode in — the reason the EDM configs were only
// batch-independent to 2e-2.
static inline int gn_gen_chunks(int HW, int N) {
    (void)N;
    if (HW >= 16384) return HW / 256 > 64 ? 64 : HW / 256;      // LSUN-size maps: 256-row chunks, at most 64
    // 64-row chunks, at most 16 per image: at batch 16 a 32x32 map is 256 workgroups instead of 64 and a 16x16 map 64 instead
    // of 16 (the EDM train step runs these launches on a fraction of the chip); every workgroup re-reads its image's partials
    // (2 C x chunks floats in the backward), which bounds the count
    const int c = HW / 64;
    return c < 1 ? 1 : (c > 16 ? 16 : c);
}

struct GnGenArgs {
    const bf16* in0;
    const bf16* in1;
    const float* gamma;
    const float* beta;
    const float* ss;     // [N][ss_ld]: scale at [c], shift at [C + c], or null
    bf16* out;
    float* part;         // [N][chunks][groups][2]
    int C0, C1, HW, groups, cpg, chunks, rows_per_chunk, ss_ld;
    float eps;
    int silu;
};

// (sum, sum of squares) of one group over the image's row chunks, added in chunk order with sixteen chunks' loads in flight (a
// plain loop is one L2 round trip per chunk: 16 of them in front of every workgroup on a 64x64 map)
__device__ __forceinline__ void gn_chunk_sums(const float* part, int groups, int chunks, float& s, float& q) {
    s = 0.f;
    q = 0.f;
    for (int k0 = 0; k0 < chunks; k0 += 16) {
        float2 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (k0 + u < chunks) t[u] = *reinterpret_cast<const float2*>(part + (size_t)(k0 + u) * groups * 2);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (k0 + u < chunks) {
                s += t[u].x;
                q += t[u].y;
            }
    }
}

__global__ __launch_bounds__(256) void gn_gen_stats_kernel(GnGenArgs p) {
    // threads = (256 / c8n) rows x c8n pieces: every thread keeps the running sum / sum of squares of its 8 channels
    // over its rows, then one thread per group adds the per-channel partials in a fixed order (bitwise reproducible).
    __shared__ float sm[2][256 * 8];
    const int C = p.C0 + p.C1, c8n = C / 8;
    const int n = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int tid = threadIdx.x;
    const int rows_par = 256 / c8n;
    const int pc = tid % c8n, rl = tid / c8n;
    const int row0 = chunk * p.rows_per_chunk;
    const int row1 = min(row0 + p.rows_per_chunk, p.HW);
    float ls[8], lq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ls[e] = 0.f; lq[e] = 0.f; }
    if (rl < rows_par) {
        const int c = pc * 8;
        const bool from0 = c < p.C0;
        const bf16* src = from0 ? p.in0 + (size_t)n * p.HW * p.C0 + c : p.in1 + (size_t)n * p.HW * p.C1 + (c - p.C0);
        const int Cs = from0 ? p.C0 : p.C1;
        for (int r = row0 + rl; r < row1; r += 4 * rows_par) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * rows_par;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = (bf16)0.f;
                if (rr < row1) v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * Cs);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    ls[e] += f; lq[e] += f * f;
                }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sm[0][rl * C + pc * 8 + e] = ls[e];
            sm[1][rl * C + pc * 8 + e] = lq[e];
        }
    }
    __syncthreads();
    if (tid < 2 * p.groups) {
        const int g = tid % p.groups, w = tid / p.groups;
        float t = 0.f;
        for (int r = 0; r < rows_par; ++r)
            for (int c = g * p.cpg; c < (g + 1) * p.cpg; ++c) t += sm[w][r * C + c];
        p.part[(((size_t)n * p.chunks + chunk) * p.groups + g) * 2 + w] = t;
    }
}

__global__ __launch_bounds__(256) void gn_gen_apply_kernel(GnGenArgs p) {
    // same thread layout as the statistics kernel: a thread owns one 8-channel piece column and strides over rows, so
    // the per-channel affine (rstd*gamma, beta - mean*rstd*gamma, FiLM scale/shift folded in) is computed once and
    // the row loop is one FMA (+ SiLU) per element, no integer division.
    __shared__ float mean_s[32], rstd_s[32];
    const int C = p.C0 + p.C1, c8n = C / 8;
    const int n = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int tid = threadIdx.x;
    if (tid < p.groups) {
        float s, q;
        gn_chunk_sums(p.part + (size_t)n * p.chunks * p.groups * 2 + tid * 2, p.groups, p.chunks, s, q);
        const float cnt = (float)p.HW * p.cpg;
        const float m = s / cnt;
        const float var = fmaxf(q / cnt - m * m, 0.f);
        mean_s[tid] = m;
        rstd_s[tid] = rsqrtf(var + p.eps);
    }
    __syncthreads();
    const int rows_par = 256 / c8n;
    const int pc = tid % c8n, rl = tid / c8n;
    if (rl >= rows_par) return;
    const int c = pc * 8;
    float A[8], Bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / p.cpg;
        float a = rstd_s[g] * p.gamma[c + e];
        float b = p.beta[c + e] - mean_s[g] * a;
        if (p.ss) {
            const float sc = 1.f + p.ss[(size_t)n * p.ss_ld + c + e];
            a *= sc;
            b = b * sc + p.ss[(size_t)n * p.ss_ld + C + c + e];
        }
        A[e] = a; Bv[e] = b;
    }
    const bool from0 = c < p.C0;
    const bf16* src = from0 ? p.in0 + (size_t)n * p.HW * p.C0 + c : p.in1 + (size_t)n * p.HW * p.C1 + (c - p.C0);
    const int Cs = from0 ? p.C0 : p.C1;
    bf16* dst = p.out + (size_t)n * p.HW * C + c;
    const int row0 = chunk * p.rows_per_chunk;
    const int row1 = min(row0 + p.rows_per_chunk, p.HW);
    // four rows per trip: the loads are issued together so each thread keeps 64 bytes in flight
    for (int r = row0 + rl; r < row1; r += 4 * rows_par) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * Cs);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = (float)v[u][e] * A[e] + Bv[e];
                    if (p.silu) y = dxmi_silu_fast(y);
                    o[e] = (bf16)y;
                }
                *reinterpret_cast<bf16x8*>(dst + (size_t)rr * C) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming GroupNorm(+SiLU) apply (round 3): the statistics pass of Normalize() is folded into whatever PRODUCED the
// tensor — the conv kernels' epilogues write (sum, sum of squares) per (image, partial, 4-channel block) of the bf16
// values they store (dxmi_conv_desc.gn_stats), gn_block_stats_kernel does the same for tensors without a producer-side
// epilogue — so the normalisation itself is ONE read + ONE write with nothing resident: a workgroup re-reduces its
// image's partials in a fixed order (32 threads, a few dozen L2 hits), forms one (scale, offset) pair per channel and
// streams its rows with U 16-byte loads in flight per thread.  The one-pass resident kernel above needed the whole
// (image, slice) in registers before the first store (all workgroups of a launch load, reduce and store in lock step:
// 3.2 TB/s in the U-Net); this one has no phase structure at all.
// 4-channel blocks are the finest group granularity of the U-Net (128 channels / 32 groups) and divide every concat
// boundary and group width it has (8, 12, 16 channels per group), so one statistics tensor per activation serves every
// GroupNorm that reads it, alone or as either half of a virtual concat.
struct GnApplyArgs {
    const bf16* in0;
    const bf16* in1;
    const float* st0;    // [N][P0][C0/2][2]
    const float* st1;    // [N][P1][C1/2][2]
    const float* gamma;
    const float* beta;
    const float* ss;     // optional FiLM scale-shift [N][ss_ld]: scale at [c], shift at [C + c] (models/cm/unet.py:252-256)
    bf16* out;
    int C0, C1, HW, groups, cpg, chunks, rows_per_chunk, P0, P1, ss_ld;
    float eps;
    int silu;
    int nt;        // tuning (DXMI_GN_APPLY_NT): bit 0 non-temporal loads of x, bit 1 non-temporal stores of the output
    float* ab;     // optional [N][C][2]: per-(image, channel) scale / offset written by gn_finalize_kernel and read here instead of
                   // the statistics prologue (dxmi_groupnorm_apply_split)
};

constexpr int GN_APPLY_MAXP = 8;        // partials per image the prologue keeps in flight (ops.MAX_APPLY_PARTIALS folds larger P)

typedef int gn_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 gn_ld(const bf16* ptr, bool nt) {
    const gn_i4* q = reinterpret_cast<const gn_i4*>(ptr);
    const gn_i4 t = nt ? __builtin_nontemporal_load(q) : *q;
    return __builtin_bit_cast(bf16x8, t);
}
__device__ __forceinline__ void gn_st(bf16* ptr, const bf16x8& v, bool nt) {
    gn_i4* q = reinterpret_cast<gn_i4*>(ptr);
    const gn_i4 t = __builtin_bit_cast(gn_i4, v);
    if (nt) __builtin_nontemporal_store(t, q);
    else *q = t;
}

template <int U>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs p) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ float2 pair_s[1024];                 // C <= 2048
    const int C = p.C0 + p.C1, c8n = C >> 3;
    const int n = blockIdx.x / p.chunks, chunk = blockIdx.x - n * p.chunks;
    const int tid = threadIdx.x;
    const int rows_par = 256 / c8n;
    const int rl = tid / c8n, pc = tid - rl * c8n;
    const bool active = rl < rows_par;
    const int c = pc * 8;
    const bool from0 = c < p.C0;
    const int Cs = from0 ? p.C0 : p.C1;
    const bf16* src = from0 ? p.in0 + (size_t)n * p.HW * p.C0 + c : p.in1 + (size_t)n * p.HW * p.C1 + (c - p.C0);
    bf16* dst = p.out + (size_t)n * p.HW * C + c;
    const int row0 = chunk * p.rows_per_chunk;
    const int row1 = min(row0 + p.rows_per_chunk, p.HW);
    // the first trip's rows are requested before the statistics prologue (independent of it)
    bf16x8 v[U];
    int r = row0 + rl;
    if (active) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) v[u] = gn_ld(src + (size_t)rr * Cs, p.nt & 1);
        }
    }
    float A[8], Bv[8];
    if (p.ab) {
        // split form (round 6): the scale / offset pairs of this image were formed once by gn_finalize_kernel (same arithmetic, same
        // order): ONE 64-byte read per thread under the first trip's loads, no LDS, no barrier — the statistics prologue below is two
        // dependent memory latencies that every workgroup of the (single) resident wave of workgroups pays at the same moment
        if (!active) return;
        const f32x4* q = reinterpret_cast<const f32x4*>(p.ab + ((size_t)n * C + c) * 2);
        const f32x4 t0 = q[0], t1 = q[1], t2 = q[2], t3 = q[3];
        A[0] = t0[0]; Bv[0] = t0[1]; A[1] = t0[2]; Bv[1] = t0[3];
        A[2] = t1[0]; Bv[2] = t1[1]; A[3] = t1[2]; Bv[3] = t1[3];
        A[4] = t2[0]; Bv[4] = t2[1]; A[5] = t2[2]; Bv[5] = t2[3];
        A[6] = t3[0]; Bv[6] = t3[1]; A[7] = t3[2]; Bv[7] = t3[3];
    } else {
    // statistics prologue, one memory latency deep: every thread sums the <= GN_APPLY_MAXP partials of one channel pair (all
    // loads in flight together, added in partial order), then 32 threads add each group's pairs in channel order through LDS
    auto pair_sums = [&](const float* st, int P, int nbs, int off) {        // uniform base + 32-bit lane offsets
        const float2* const base = reinterpret_cast<const float2*>(st) + (size_t)n * P * nbs;
        for (int b = tid; b < nbs; b += 256) {
            float2 t[GN_APPLY_MAXP];
#pragma unroll
            for (int k = 0; k < GN_APPLY_MAXP; ++k)
                if (k < P) t[k] = base[(unsigned)(k * nbs + b)];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int k = 0; k < GN_APPLY_MAXP; ++k)
                if (k < P) {
                    s += t[k].x;
                    q += t[k].y;
                }
            for (int k = GN_APPLY_MAXP; k < P; ++k) {       // more partials than the host folds to: correct, just serial
                const float2 tk = base[(unsigned)(k * nbs + b)];
                s += tk.x;
                q += tk.y;
            }
            pair_s[off + b] = make_float2(s, q);
        }
    };
    pair_sums(p.st0, p.P0, p.C0 >> 1, 0);
    if (p.C1) pair_sums(p.st1, p.P1, p.C1 >> 1, p.C0 >> 1);
    __syncthreads();
    // gamma / beta are requested before the group reduction (the partials' registers are free again)
    f32x4 g0, g1, b0, b1;
    if (active) {
        g0 = *reinterpret_cast<const f32x4*>(p.gamma + c);
        g1 = *reinterpret_cast<const f32x4*>(p.gamma + c + 4);
        b0 = *reinterpret_cast<const f32x4*>(p.beta + c);
        b1 = *reinterpret_cast<const f32x4*>(p.beta + c + 4);
    }
    // FiLM scale / shift of this image (scale-shift norm of the ADM ResBlocks): requested here as well — read after the group
    // reduction (round 3 form: sixteen scalar loads behind the barrier) they were a SECOND exposed memory latency per workgroup
    // of ~32 KB: 64x64x192 at 100 images 86.8 us with them against 61.0 us without
    f32x4 sc0 = {0.f, 0.f, 0.f, 0.f}, sc1 = sc0, sh0 = sc0, sh1 = sc0;
    if (active && p.ss) {
        const float* ssn = p.ss + (size_t)n * p.ss_ld + c;
        sc0 = *reinterpret_cast<const f32x4*>(ssn);
        sc1 = *reinterpret_cast<const f32x4*>(ssn + 4);
        sh0 = *reinterpret_cast<const f32x4*>(ssn + C);
        sh1 = *reinterpret_cast<const f32x4*>(ssn + C + 4);
    }
    if (tid < p.groups) {
        const int bpg = p.cpg >> 1;
        float s = 0.f, q = 0.f;
        for (int b = tid * bpg; b < (tid + 1) * bpg; ++b) {
            s += pair_s[b].x;
            q += pair_s[b].y;
        }
        const float cnt = (float)p.HW * (float)p.cpg;
        const float m = s / cnt;
        mean_s[tid] = m;
        rstd_s[tid] = rsqrtf(fmaxf(q / cnt - m * m, 0.f) + p.eps);
    }
    __syncthreads();
    if (!active) return;
    {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / p.cpg;
            float a = rstd_s[g] * (e < 4 ? g0[e] : g1[e - 4]);
            float b = (e < 4 ? b0[e] : b1[e - 4]) - mean_s[g] * a;
            if (p.ss) {
                const float sc = 1.f + (e < 4 ? sc0[e] : sc1[e - 4]);
                a *= sc;
                b = b * sc + (e < 4 ? sh0[e] : sh1[e - 4]);
            }
            A[e] = a;
            Bv[e] = b;
        }
    }
    }
    for (;;) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y = (float)v[u][e] * A[e] + Bv[e];
                    if (p.silu) y = dxmi_silu_fast(y);
                    o[e] = (bf16)y;
                }
                gn_st(dst + (size_t)rr * C, o, p.nt & 2);
            }
        }
        r += U * rows_par;
        if (r >= row1) break;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) v[u] = gn_ld(src + (size_t)rr * Cs, p.nt & 1);
        }
    }
}

// Block statistics of a tensor nobody produced statistics for: x [N,HW,C] -> st [N][P][C/2][2], P row chunks per image.
// Thread = one 8-channel piece column (four pairs) over the chunk's rows; the row-parallel partials are added in a fixed
// order through LDS.
__global__ __launch_bounds__(256) void gn_block_stats_kernel(const bf16* __restrict__ x, float* __restrict__ st, int HW, int C,
                                                             int chunks, int rows_per_chunk) {
    __shared__ float sm[256 * 8];         // [row lane][piece][4 pairs x (s, q)]
    const int c8n = C >> 3;
    const int n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
    const int tid = threadIdx.x;
    const int rows_par = 256 / c8n;
    const int rl = tid / c8n, pc = tid - rl * c8n;
    const int row0 = chunk * rows_per_chunk;
    const int row1 = min(row0 + rows_per_chunk, HW);
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        const bf16* src = x + (size_t)n * HW * C + pc * 8;
        for (int r = row0 + rl; r < row1; r += 4 * rows_par) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * rows_par;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = (bf16)0.f;
                if (rr < row1) v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x4 lo = {v[u][0], v[u][1], v[u][2], v[u][3]}, hi = {v[u][4], v[u][5], v[u][6], v[u][7]};
                dxmi_stats4(lo, a0);
                dxmi_stats4(hi, a1);
            }
        }
        float* d = sm + (rl * c8n + pc) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] = a0[e]; d[4 + e] = a1[e]; }
    }
    __syncthreads();
    // one thread per output float: (C/2 pairs) x (s, q) = C values, laid out exactly like a piece's 8 LDS floats
    for (int i = tid; i < C; i += 256) {
        float t = 0.f;
        for (int rr = 0; rr < rows_par; ++rr) t += sm[(rr * c8n) * 8 + i];
        st[((size_t)n * chunks + chunk) * C + i] = t;
    }
}

// Partial sums of an image folded G at a time: st [N][P][C/2][2] -> out [N][ceil(P/G)][C/2][2], fixed order.  The conv
// epilogue writes one partial per (pixel tile, pixel half): 8 at 32x32, but 512 at 256x256 (LSUN) — too many for every apply
// workgroup to re-add, and too long a dependent chain for one thread per value (a 512-deep fold by N x C threads measured
// ~0.3 ms): G <= 32 per thread, a second launch folds what is left.
__global__ __launch_bounds__(256) void gn_stats_fold_kernel(const float* __restrict__ st, float* __restrict__ out, int P, int C, int G) {
    const int PG = (P + G - 1) / G;
    const int n = blockIdx.y / PG, pg = blockIdx.y - n * PG, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C) return;
    const int k0 = pg * G, k1 = min(k0 + G, P);
    const float* s = st + ((size_t)n * P + k0) * C + i;
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < k1 - k0; ++k) t += s[(size_t)k * C];
    out[((size_t)n * PG + pg) * C + i] = t;
}

// ---------------------------------------------------------------------------------------------
// Backward of GroupNorm(+SiLU): given x (virtual concat), the upstream gradient dy w.r.t. the
// kernel's output, gamma/beta -> dx (split back into the two concat sources, optional additive
// inputs fused: skip-connection / residual gradients) and per-image partial d(gamma), d(beta).
// Same one-pass structure as the forward: x and dy slices stay resident as raw bf16, statistics are
// recomputed (two-pass) instead of being stored by the forward.
//   xh = (x-mean)*rstd ; y = xh*g + b ; z = silu(y) | y
//   dyy = dy * silu'(y) ; dg += dyy*xh ; db += dyy ; dxh = dyy*g
//   dx = rstd * (dxh - mean_grp(dxh) - xh * mean_grp(dxh*xh))
struct GnBwdArgs {
    const bf16* in0;
    const bf16* in1;
    const bf16* dy;      // [N,HW,C0+C1]
    const bf16* add0;    // optional, added to dx0
    const bf16* add1;    // optional, added to dx1
    const float* gamma;
    const float* beta;
    bf16* dx0;
    bf16* dx1;
    float* dgamma_part;  // [N][C]
    float* dbeta_part;   // [N][C]
    int C0, C1, HW, groups, slices, ppp, cpg, gps;
    int xcd;
    float eps;
    int silu;
    int fast;
};

template <int VEC, int PIECES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((VEC * PIECES <= 64 ? 4 : 1), 8))) void gn_silu_bwd_kernel(GnBwdArgs p) {
    typedef typename PieceT<VEC>::type piece_t;
    constexpr int W = VEC / 2;
    __shared__ float red[512 / 64][32];
    __shared__ float stat[32];
    extern __shared__ __attribute__((aligned(16))) char dyn[];  // [nthr][2*VEC] floats for the channel sums
    float* chan = reinterpret_cast<float*>(dyn);

    const int C = p.C0 + p.C1;
    const int bid = xcd_logical_block(blockIdx.x, gridDim.x, p.xcd);      // the slices of an image on ONE XCD (one L2)
    const int n = bid / p.slices;
    const int s = bid % p.slices;
    const int Csl = C / p.slices;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const int pc = tid % p.ppp;
    const int prow = tid / p.ppp;
    const int rows_per_iter = nthr / p.ppp;
    const int c = s * Csl + pc * VEC;
    const int gl = (pc * VEC) / p.cpg;
    const bool from0 = c < p.C0;
    const bf16* src = from0 ? p.in0 : p.in1;
    const int Cs = from0 ? p.C0 : p.C1;
    const int cs = from0 ? c : c - p.C0;

    uint32_t xv[PIECES][W], gv[PIECES][W];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
#pragma unroll
        for (int e = 0; e < W; ++e) { xv[q][e] = 0u; gv[q][e] = 0u; }
        if (px < p.HW) {
            const piece_t t = *reinterpret_cast<const piece_t*>(src + ((size_t)n * p.HW + px) * Cs + cs);
            const piece_t u = *reinterpret_cast<const piece_t*>(p.dy + ((size_t)n * p.HW + px) * C + c);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                xv[q][e] = reinterpret_cast<const uint32_t*>(&t)[e];
                gv[q][e] = reinterpret_cast<const uint32_t*>(&u)[e];
            }
        }
    }
    const int ppt = p.cpg / VEC;
    auto group_reduce = [&](float val) -> float {
        if (p.fast) {
            float t = val;
            for (int o = 32; o >= p.ppp; o >>= 1) t += __shfl_xor(t, o, 64);
            for (int o = ppt >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane < p.ppp && (lane % ppt) == 0) red[wave][gl] = t;
        } else {
            for (int g = 0; g < p.gps; ++g) {
                float t = wave_sum(gl == g ? val : 0.f);
                if (lane == 0) red[wave][g] = t;
            }
        }
        __syncthreads();
        if (tid < p.gps) {
            float t = 0.f;
            for (int w = 0; w < nwaves; ++w) t += red[w][tid];
            stat[tid] = t;
        }
        __syncthreads();
        const float r = stat[gl];
        __syncthreads();
        return r;
    };
    const float inv_cnt = 1.f / (float)(p.HW * p.cpg);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < PIECES; ++q)
#pragma unroll
        for (int e = 0; e < W; ++e) {
            asm volatile("" : "+v"(xv[q][e]));
            sum += GN_LO(xv[q][e]) + GN_HI(xv[q][e]);
        }
    const float mean = group_reduce(sum) * inv_cnt;
    float ssq = 0.f;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(xv[q][e]));
                const float d0 = GN_LO(xv[q][e]) - mean, d1 = GN_HI(xv[q][e]) - mean;
                ssq += d0 * d0 + d1 * d1;
            }
        }
    }
    const float rstd = rsqrtf(group_reduce(ssq) * inv_cnt + p.eps);

    float ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ga[e] = p.gamma[c + e]; be[e] = p.beta[c + e]; }

    // pass A: channel sums (dgamma, dbeta) and the two group sums; dyy is recomputed in pass B
    auto dyy_of = [&](float xh, float g, float gam, float bet) -> float {
        if (!p.silu) return g;
        const float y = xh * gam + bet;
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-y));
        return g * (sg * (1.f + y * (1.f - sg)));
    };
    float dgs[VEC], dbs[VEC], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { dgs[e] = 0.f; dbs[e] = 0.f; }
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(xv[q][e]));
                asm volatile("" : "+v"(gv[q][e]));
                const float xh0 = (GN_LO(xv[q][e]) - mean) * rstd, xh1 = (GN_HI(xv[q][e]) - mean) * rstd;
                const float a0 = dyy_of(xh0, GN_LO(gv[q][e]), ga[2 * e], be[2 * e]);
                const float a1 = dyy_of(xh1, GN_HI(gv[q][e]), ga[2 * e + 1], be[2 * e + 1]);
                dgs[2 * e] += a0 * xh0; dgs[2 * e + 1] += a1 * xh1;
                dbs[2 * e] += a0; dbs[2 * e + 1] += a1;
                const float h0 = a0 * ga[2 * e], h1 = a1 * ga[2 * e + 1];
                s1 += h0 + h1;
                s2 += h0 * xh0 + h1 * xh1;
            }
        }
    }
    const float m1 = group_reduce(s1) * inv_cnt;
    const float m2 = group_reduce(s2) * inv_cnt;

    // channel sums across the threads that share a channel piece, in a fixed order.  chan is a [rows][cols] matrix (rows = nthr / ppp
    // thread rows, cols = ppp x 2 VEC values): row segments are added in parallel, then the segment sums in order (the first version
    // let ppp threads walk all rows x 2 VEC values alone: 512 dependent LDS reads, ~20 us of a workgroup's life)
    constexpr int J = 2 * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { chan[tid * J + e] = dgs[e]; chan[tid * J + VEC + e] = dbs[e]; }
    __syncthreads();
    {
        const int cols = p.ppp * J, rows = nthr / p.ppp;
        int nseg = nthr / cols;
        if (nseg < 1) nseg = 1;
        if (nseg > rows) nseg = rows;
        float* const seg_part = chan + nthr * J;            // [nseg][cols], nseg * cols <= nthr (only used when nseg > 1)
        if (nseg > 1) {
            for (int i = tid; i < nseg * cols; i += nthr) {
                const int sg = i / cols, cc = i - sg * cols;
                const int r0 = rows * sg / nseg, r1 = rows * (sg + 1) / nseg;
                float t = 0.f;
                for (int r = r0; r < r1; ++r) t += chan[r * cols + cc];
                seg_part[i] = t;
            }
            __syncthreads();
        }
        for (int cc = tid; cc < cols; cc += nthr) {
            float t = 0.f;
            if (nseg > 1) {
                for (int sg = 0; sg < nseg; ++sg) t += seg_part[sg * cols + cc];
            } else {
                for (int r = 0; r < rows; ++r) t += chan[r * cols + cc];       // fewer rows than values per row: one short pass
            }
            const int pcc = cc / J, j = cc - pcc * J;
            const size_t o = (size_t)n * C + s * Csl + pcc * VEC + (j < VEC ? j : j - VEC);
            if (j < VEC) p.dgamma_part[o] = t;
            else p.dbeta_part[o] = t;
        }
    }

    // pass B: dx
    bf16* dst = from0 ? p.dx0 : p.dx1;
    const bf16* add = from0 ? p.add0 : p.add1;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const int px = prow + q * rows_per_iter;
        if (px < p.HW) {
            const size_t o = ((size_t)n * p.HW + px) * Cs + cs;
            piece_t av;
            if (add) av = *reinterpret_cast<const piece_t*>(add + o);
            piece_t ov;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                asm volatile("" : "+v"(xv[q][e]));
                asm volatile("" : "+v"(gv[q][e]));
                const float xh0 = (GN_LO(xv[q][e]) - mean) * rstd, xh1 = (GN_HI(xv[q][e]) - mean) * rstd;
                const float h0 = dyy_of(xh0, GN_LO(gv[q][e]), ga[2 * e], be[2 * e]) * ga[2 * e];
                const float h1 = dyy_of(xh1, GN_HI(gv[q][e]), ga[2 * e + 1], be[2 * e + 1]) * ga[2 * e + 1];
                float r0 = rstd * (h0 - m1 - xh0 * m2), r1 = rstd * (h1 - m1 - xh1 * m2);
                if (add) { r0 += (float)av[2 * e]; r1 += (float)av[2 * e + 1]; }
                ov[2 * e] = (bf16)r0;
                ov[2 * e + 1] = (bf16)r1;
            }
            *reinterpret_cast<piece_t*>(dst + o) = ov;
        }
    }
}

int gcd(int a, int b) { return b ? gcd(b, a % b) : a; }

template <int VEC>
int launch_gn(const GnArgs& a, int N, int pieces, int threads, hipStream_t st) {
    dim3 grid(N * a.slices), block(threads);
#define GN_CASE(P)                                                                   \
    if (pieces <= P) {                                                               \
        hipLaunchKernelGGL((gn_silu_kernel<VEC, P>), grid, block, 0, st, a);         \
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_silu_fwd");                                \
        return DXMI_OK;                                                              \
    }
    GN_CASE(1) GN_CASE(2) GN_CASE(4) GN_CASE(8) GN_CASE(12) GN_CASE(16) GN_CASE(24) GN_CASE(32)
#undef GN_CASE
    dxmi_set_error("dxmi_groupnorm_silu_fwd: %d pieces per thread unsupported", pieces);
    return DXMI_EINVAL;
}

}  // namespace

template <int VEC>
int launch_gn_bwd(const GnBwdArgs& a, int N, int pieces, int threads, hipStream_t st) {
    dim3 grid(N * a.slices), block(threads);
    const size_t dyn = (size_t)threads * (2 * VEC + 1) * sizeof(float);      // channel partials [threads][2 VEC] + segment sums (<= threads)
#define GNB_CASE(P)                                                                     \
    if (pieces <= P) {                                                                  \
        hipLaunchKernelGGL((gn_silu_bwd_kernel<VEC, P>), grid, block, dyn, st, a);      \
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_silu_bwd");                                   \
        return DXMI_OK;                                                                 \
    }
    GNB_CASE(1) GNB_CASE(2) GNB_CASE(4) GNB_CASE(8) GNB_CASE(12) GNB_CASE(16)
#undef GNB_CASE
    dxmi_set_error("dxmi_groupnorm_silu_bwd: %d pieces per thread unsupported", pieces);
    return DXMI_EINVAL;
}

// shared slicing rule: whole groups per slice, <= max_pieces pieces per thread at <= 512 threads
static int gn_plan(int C, int C0, int HW, int groups, int max8, int max4, int* VEC_, int* slices_, int* threads_,
                   int* pieces_, int* ppp_) {
    const int cpg = C / groups;
    const int VEC = (cpg % 8 == 0 && C0 % 8 == 0) ? 8 : 4;
    const int max_pieces = VEC == 8 ? max8 : max4;
    int slices = 1, threads = 0, pieces = 0, ppp = 0;
    for (;; slices *= 2) {
        if (groups % slices != 0) return -1;
        const int Csl = C / slices;
        ppp = Csl / VEC;
        const int unit = ppp / gcd(ppp, 64) * 64;
        if (unit > 512) return -2;
        long total = (long)HW * ppp;
        threads = (512 / unit) * unit;
        if (total < threads) threads = (int)((total + unit - 1) / unit) * unit;
        pieces = (int)((total + threads - 1) / threads);
        if (pieces <= max_pieces || slices == groups) break;
    }
    *VEC_ = VEC; *slices_ = slices; *threads_ = threads; *pieces_ = pieces; *ppp_ = ppp;
    return 0;
}

extern "C" int dxmi_groupnorm_silu_bwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                                       const void* add0, const void* add1, const float* gamma, const float* beta, void* dx0,
                                       void* dx1, float* dgamma_part, float* dbeta_part, int32_t N, int32_t HW,
                                       int32_t groups, float eps, int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(in0 && dy && gamma && beta && dx0 && dgamma_part && dbeta_part, "dxmi_groupnorm_silu_bwd: null pointer");
    DXMI_CHECK_ARG(C1 == 0 || (in1 && dx1), "dxmi_groupnorm_silu_bwd: C1>0 needs in1 and dx1");
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0, "dxmi_groupnorm_silu_bwd: C=%d groups=%d", C, groups);
    const int cpg = C / groups;
    DXMI_CHECK_ARG(cpg % 4 == 0 && C0 % 4 == 0, "dxmi_groupnorm_silu_bwd: channels per group (%d) must be a multiple of 4", cpg);
    int VEC, slices, threads, pieces, ppp;
    DXMI_CHECK_ARG(gn_plan(C, C0, HW, groups, 8, 16, &VEC, &slices, &threads, &pieces, &ppp) == 0 && pieces <= 16,
                   "dxmi_groupnorm_silu_bwd: cannot slice HW=%d C=%d", HW, C);
    GnBwdArgs a;
    a.in0 = (const bf16*)in0; a.in1 = (const bf16*)in1; a.dy = (const bf16*)dy; a.add0 = (const bf16*)add0;
    a.add1 = (const bf16*)add1; a.gamma = gamma; a.beta = beta; a.dx0 = (bf16*)dx0; a.dx1 = (bf16*)dx1;
    a.dgamma_part = dgamma_part; a.dbeta_part = dbeta_part;
    a.C0 = C0; a.C1 = C1; a.HW = HW; a.groups = groups; a.slices = slices; a.ppp = ppp; a.cpg = cpg;
    a.gps = groups / slices; a.eps = eps; a.silu = apply_silu;
    static const int gn_xcd_env = getenv("DXMI_GN_XCD") ? atoi(getenv("DXMI_GN_XCD")) : 1;
    a.xcd = gn_xcd_env;
    const int ppt = cpg / VEC;
    a.fast = (ppp <= 64) && ((ppp & (ppp - 1)) == 0) && ((ppt & (ppt - 1)) == 0);
    hipStream_t st = (hipStream_t)stream;
    return VEC == 8 ? launch_gn_bwd<8>(a, N, pieces, threads, st) : launch_gn_bwd<4>(a, N, pieces, threads, st);
}

extern "C" int64_t dxmi_groupnorm_generic_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
    const int chunks = gn_gen_chunks(HW, N);
    (void)C;
    return (int64_t)N * chunks * 32 * 2 * 4;
}

// Generic GroupNorm(+scale-shift)(+SiLU): any even... any C0, C1 multiples of 8, any channels per group.
extern "C" int dxmi_groupnorm_generic_fwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const float* gamma,
                                          const float* beta, const float* scale_shift, int32_t ss_ld, void* out,
                                          void* workspace, int32_t N, int32_t HW, int32_t groups, float eps,
                                          int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(in0 && out && gamma && beta && workspace, "dxmi_groupnorm_generic_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && HW > 0 && C0 > 0 && C1 >= 0, "dxmi_groupnorm_generic_fwd: empty or negative shape (N %d, HW %d, C %d+%d)", N, HW, C0, C1);
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0 && C0 % 8 == 0 && C1 % 8 == 0 && (C1 == 0 || in1) && C <= 2048,
                   "dxmi_groupnorm_generic_fwd: C0=%d C1=%d groups=%d", C0, C1, groups);
    GnGenArgs a;
    a.in0 = (const bf16*)in0; a.in1 = (const bf16*)in1; a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.ss_ld = ss_ld;
    a.out = (bf16*)out; a.part = (float*)workspace; a.C0 = C0; a.C1 = C1; a.HW = HW; a.groups = groups; a.cpg = C / groups;
    a.chunks = gn_gen_chunks(HW, N);
    a.rows_per_chunk = (HW + a.chunks - 1) / a.chunks;
    a.eps = eps; a.silu = apply_silu;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_gen_stats_kernel, dim3(N * a.chunks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_fwd(stats)");
    hipLaunchKernelGGL(gn_gen_apply_kernel, dim3(N * a.chunks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_fwd(apply)");
    return DXMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Generic GroupNorm(+FiLM scale-shift)(+SiLU) backward for the shapes of the generic forward (any channels per group,
// any map size).  With ga = gamma*(1+scale), be = beta*(1+scale)+shift (per image and channel; scale = shift = 0
// without FiLM):  xh = (x-mean)*rstd ; y = xh*ga + be ; z = silu(y) | y ; dyy = dz * silu'(y)
//   G0[n,c] = sum_px dyy ; G1[n,c] = sum_px dyy*xh                       (returned: the host folds them into
//   dgamma, dbeta, dscale, dshift — a few [N,C] element-wise ops)
//   dx = rstd * (ga*dyy - mean_grp(ga*dyy) - xh * mean_grp(ga*dyy*xh))   (+ optional additive inputs add0 | add1)
// Three launches: the forward statistics kernel (mean, rstd), a per-(image, row-chunk) partial reduction of G0/G1 in a
// fixed order, and the apply pass.  Thread layout as in the forward (one 8-channel piece column per thread).
struct GnGenBwdArgs {
    const bf16* in0;
    const bf16* in1;
    const bf16* dy;
    const bf16* add0;
    const bf16* add1;
    const float* gamma;
    const float* beta;
    const float* ss;
    bf16* dx0;
    bf16* dx1;
    const float* part;   // forward statistics partials [N][chunks][groups][2]
    float* gpart;        // [N][chunks][C][2]
    float* g_out;        // [2][N][C]: G0 then G1
    int C0, C1, HW, groups, cpg, chunks, rows_per_chunk, ss_ld, N;
    float eps;
    int silu;
    // one-launch form (gn_gen_bwd_fused_kernel): arrival counter per image, (group, channel) partials as 8-byte granules
    unsigned* cnt;                 // [N], zeroed by gn_zero_counters_kernel in front of the launch
    unsigned long long* grp;       // [N][wchunks][groups]: (sum_c ga*G0, sum_c ga*G1) of a work chunk
    unsigned long long* cpart;     // [N][wchunks][C]: (G0, G1) of a work chunk
    int wchunks, wrows;
};

__device__ __forceinline__ void gn_gen_group_stats(const GnGenBwdArgs& p, int n, float* mean_s, float* rstd_s) {
    const int tid = threadIdx.x;
    if (tid < p.groups) {
        float s, q;
        gn_chunk_sums(p.part + (size_t)n * p.chunks * p.groups * 2 + tid * 2, p.groups, p.chunks, s, q);
        const float cnt = (float)p.HW * p.cpg;
        const float m = s / cnt;
        mean_s[tid] = m;
        rstd_s[tid] = rsqrtf(fmaxf(q / cnt - m * m, 0.f) + p.eps);
    }
    __syncthreads();
}

__device__ __forceinline__ float gn_dsilu(float y) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-y));
    return sg * (1.f + y * (1.f - sg));
}

__global__ __launch_bounds__(256) void gn_gen_bwd_reduce_kernel(GnGenBwdArgs p) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ float sm[2][256 * 8];
    const int C = p.C0 + p.C1, c8n = C / 8;
    const int n = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int tid = threadIdx.x;
    gn_gen_group_stats(p, n, mean_s, rstd_s);
    const int rows_par = 256 / c8n;
    const int pc = tid % c8n, rl = tid / c8n;
    const int c = pc * 8;
    if (rl < rows_par) {
        float A[8], Bv[8], s0[8], s1[8], mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / p.cpg;
            mu[e] = mean_s[g]; rs[e] = rstd_s[g];
            float ga = p.gamma[c + e], be = p.beta[c + e];
            if (p.ss) {
                const float sc = 1.f + p.ss[(size_t)n * p.ss_ld + c + e];
                ga *= sc;
                be = be * sc + p.ss[(size_t)n * p.ss_ld + C + c + e];
            }
            A[e] = ga; Bv[e] = be; s0[e] = 0.f; s1[e] = 0.f;
        }
        const bool from0 = c < p.C0;
        const bf16* src = from0 ? p.in0 + (size_t)n * p.HW * p.C0 + c : p.in1 + (size_t)n * p.HW * p.C1 + (c - p.C0);
        const int Cs = from0 ? p.C0 : p.C1;
        const bf16* dyp = p.dy + (size_t)n * p.HW * C + c;
        const int row0 = chunk * p.rows_per_chunk;
        const int row1 = min(row0 + p.rows_per_chunk, p.HW);
        // four rows per trip, loads issued together (same row order of the sums as one row per trip)
        for (int r = row0 + rl; r < row1; r += 4 * rows_par) {
            bf16x8 v[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + u * rows_par;
                if (rr < row1) {
                    v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * Cs);
                    d[u] = *reinterpret_cast<const bf16x8*>(dyp + (size_t)rr * C);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (r + u * rows_par >= row1) break;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)v[u][e] - mu[e]) * rs[e];
                    float dyy = (float)d[u][e];
                    if (p.silu) dyy *= gn_dsilu(xh * A[e] + Bv[e]);
                    s0[e] += dyy; s1[e] += dyy * xh;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sm[0][rl * C + c + e] = s0[e];
            sm[1][rl * C + c + e] = s1[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        const int w = i / C, cc = i % C;
        float t = 0.f;
        for (int r = 0; r < rows_par; ++r) t += sm[w][r * C + cc];
        p.gpart[(((size_t)n * p.chunks + chunk) * C + cc) * 2 + w] = t;
    }
}

__global__ __launch_bounds__(256) void gn_gen_bwd_apply_kernel(GnGenBwdArgs p) {
    __shared__ float mean_s[32], rstd_s[32], m1_s[32], m2_s[32];
    __shared__ float gsum[2][2048];
    const int C = p.C0 + p.C1, c8n = C / 8;
    const int n = blockIdx.x / p.chunks, chunk = blockIdx.x % p.chunks;
    const int tid = threadIdx.x;
    gn_gen_group_stats(p, n, mean_s, rstd_s);
    for (int i = tid; i < 2 * C; i += 256) {
        const int w = i / C, cc = i % C;
        float t = 0.f;
        for (int k0 = 0; k0 < p.chunks; k0 += 16) {
            float tk[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < p.chunks) tk[u] = p.gpart[(((size_t)n * p.chunks + k0 + u) * C + cc) * 2 + w];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k0 + u < p.chunks) t += tk[u];
        }
        gsum[w][cc] = t;
        if (chunk == 0) p.g_out[((size_t)w * p.N + n) * C + cc] = t;
    }
    __syncthreads();
    if (tid < p.groups) {
        float a = 0.f, b = 0.f;
        for (int cc = tid * p.cpg; cc < (tid + 1) * p.cpg; ++cc) {
            float ga = p.gamma[cc];
            if (p.ss) ga *= 1.f + p.ss[(size_t)n * p.ss_ld + cc];
            a += ga * gsum[0][cc];
            b += ga * gsum[1][cc];
        }
        const float cnt = (float)p.HW * p.cpg;
        m1_s[tid] = a / cnt;
        m2_s[tid] = b / cnt;
    }
    __syncthreads();
    const int rows_par = 256 / c8n;
    const int pc = tid % c8n, rl = tid / c8n;
    if (rl >= rows_par) return;
    const int c = pc * 8;
    float A[8], Bv[8], mu[8], rs[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / p.cpg;
        mu[e] = mean_s[g]; rs[e] = rstd_s[g]; m1[e] = m1_s[g]; m2[e] = m2_s[g];
        float ga = p.gamma[c + e], be = p.beta[c + e];
        if (p.ss) {
            const float sc = 1.f + p.ss[(size_t)n * p.ss_ld + c + e];
            ga *= sc;
            be = be * sc + p.ss[(size_t)n * p.ss_ld + C + c + e];
        }
        A[e] = ga; Bv[e] = be;
    }
    const bool from0 = c < p.C0;
    const int Cs = from0 ? p.C0 : p.C1;
    const size_t soff = (size_t)n * p.HW * Cs + (from0 ? c : c - p.C0);
    const bf16* src = (from0 ? p.in0 : p.in1) + soff;
    const bf16* add = from0 ? p.add0 : p.add1;
    if (add) add += soff;
    bf16* dst = (from0 ? p.dx0 : p.dx1) + soff;
    const bf16* dyp = p.dy + (size_t)n * p.HW * C + c;
    const int row0 = chunk * p.rows_per_chunk;
    const int row1 = min(row0 + p.rows_per_chunk, p.HW);
    for (int r = row0 + rl; r < row1; r += 4 * rows_par) {
        bf16x8 v[4], d[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * rows_par;
            if (rr < row1) {
                v[u] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * Cs);
                d[u] = *reinterpret_cast<const bf16x8*>(dyp + (size_t)rr * C);
                if (add) av[u] = *reinterpret_cast<const bf16x8*>(add + (size_t)rr * Cs);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r + u * rows_par;
            if (rr >= row1) break;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)v[u][e] - mu[e]) * rs[e];
                float dyy = (float)d[u][e];
                if (p.silu) dyy *= gn_dsilu(xh * A[e] + Bv[e]);
                float dx = rs[e] * (A[e] * dyy - m1[e] - xh * m2[e]);
                if (add) dx += (float)av[u][e];
                o[e] = (bf16)dx;
            }
            *reinterpret_cast<bf16x8*>(dst + (size_t)rr * Cs) = o;
        }
    }
}

// One-launch form of the two kernels above (round 6).  A workgroup owns a WORK chunk of KMAX x (256 / c8n) pixel rows that its
// threads keep in registers: (1) partial G0 / G1 of its rows; per-group (sum_c ga G0, sum_c ga G1) and the per-channel pairs are
// published as 8-byte granules with agent-scope write-through stores, then ONE counter add per workgroup; (2) lane 0 polls the
// image's counter until every work chunk of the image has arrived (the chunks of an image are consecutive workgroup ids, the
// dispatcher hands workgroups out in id order and an image has <= 128 of them, far below what the chip holds: the chunks a
// workgroup waits for are always resident or ahead of every later image's in their XCD's queue); (3) the group means are added in
// chunk order from the granules (agent-scope loads: no acquire fence is needed in front of them) and dx is formed from the
// REGISTERS — x and dy are read once instead of twice, and the second launch with its statistics prologue is gone; (4) the
// image's chunk-0 workgroup adds the per-channel pairs in chunk order into g_out.  Every sum has a fixed order.  The spin is
// bounded: a workgroup that gives up writes NaN over its rows (a loss that cannot be missed) instead of hanging the device.
__device__ __forceinline__ unsigned long long gn_pack2(float a, float b) {
    return ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
}

// one arrival counter per image, 256 bytes apart: every workgroup of an image polls its counter, and sixteen counters in one 64-byte
// line put all those polls (and the arrivals they wait for) on ONE memory channel
constexpr int GN_CNT_STRIDE = 64;     // words

__global__ void gn_zero_counters_kernel(unsigned* cnt, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) __hip_atomic_store(cnt + (size_t)i * GN_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KMAX>
__global__ __launch_bounds__(256) void gn_gen_bwd_fused_kernel(GnGenBwdArgs p) {
    __shared__ float mean_s[32], rstd_s[32], m1_s[32], m2_s[32];
    __shared__ unsigned long long pool[4096];       // 32 KB: (row-group sums | per-channel sums) first, then the image's group granules
    __shared__ int ok_s;
    float (*sm)[2048] = reinterpret_cast<float (*)[2048]>(pool);
    float (*gs)[2048] = reinterpret_cast<float (*)[2048]>(reinterpret_cast<float*>(pool) + 4096);
    const int C = p.C0 + p.C1, c8n = C / 8;
    const int n = blockIdx.x / p.wchunks, w = blockIdx.x % p.wchunks;
    const int tid = threadIdx.x;
    const int rows_par = 256 / c8n;
    const int pc = tid % c8n, rl = tid / c8n;
    const bool active = rl < rows_par;
    const int c = pc * 8;
    const int row0 = w * p.wrows;
    const int row1 = min(row0 + p.wrows, p.HW);
    const bool from0 = c < p.C0;
    const int Cs = from0 ? p.C0 : p.C1;
    const size_t soff = (size_t)n * p.HW * Cs + (from0 ? c : c - p.C0);
    const bf16* src = (from0 ? p.in0 : p.in1) + soff;
    const bf16* add = from0 ? p.add0 : p.add1;
    if (add) add += soff;
    bf16* dst = (from0 ? p.dx0 : p.dx1) + soff;
    const bf16* dyp = p.dy + (size_t)n * p.HW * C + c;
    bf16x8 v[KMAX], d[KMAX];
    float A[8], Bv[8], mu[8], rs[8];
    if (active) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int rr = row0 + rl + k * rows_par;
            if (rr < row1) {
                v[k] = *reinterpret_cast<const bf16x8*>(src + (size_t)rr * Cs);
                d[k] = *reinterpret_cast<const bf16x8*>(dyp + (size_t)rr * C);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float ga = p.gamma[c + e], be = p.beta[c + e];
            if (p.ss) {
                const float sc = 1.f + p.ss[(size_t)n * p.ss_ld + c + e];
                ga *= sc;
                be = be * sc + p.ss[(size_t)n * p.ss_ld + C + c + e];
            }
            A[e] = ga; Bv[e] = be;
        }
    }
    gn_gen_group_stats(p, n, mean_s, rstd_s);       // behind the rows' and parameters' loads: its round trip (statistics partials) overlaps theirs
    if (active) {
        float s0[8], s1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / p.cpg;
            mu[e] = mean_s[g]; rs[e] = rstd_s[g];
            s0[e] = 0.f; s1[e] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (row0 + rl + k * rows_par < row1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)v[k][e] - mu[e]) * rs[e];
                    float dyy = (float)d[k][e];
                    if (p.silu) dyy *= gn_dsilu(xh * A[e] + Bv[e]);
                    s0[e] += dyy; s1[e] += dyy * xh;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sm[0][rl * C + c + e] = s0[e];
            sm[1][rl * C + c + e] = s1[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        const int h = i / C, cc = i - h * C;
        float t = 0.f;
        for (int r = 0; r < rows_par; ++r) t += sm[h][r * C + cc];
        gs[h][cc] = t;
    }
    __syncthreads();
    const size_t wg = (size_t)n * p.wchunks + w;
    for (int cc = tid; cc < C; cc += 256)
        __hip_atomic_store(p.cpart + wg * C + cc, gn_pack2(gs[0][cc], gs[1][cc]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < p.groups) {
        float a = 0.f, b = 0.f;
        for (int cc = tid * p.cpg; cc < (tid + 1) * p.cpg; ++cc) {
            float ga = p.gamma[cc];
            if (p.ss) ga *= 1.f + p.ss[(size_t)n * p.ss_ld + cc];
            a += ga * gs[0][cc];
            b += ga * gs[1][cc];
        }
        __hip_atomic_store(p.grp + wg * p.groups + tid, gn_pack2(a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the additive input's rows: requested in front of the wait (short forms only: registers)
    bf16x8 av[KMAX <= 8 ? KMAX : 1];
    if (KMAX <= 8 && add && active) {
#pragma unroll
        for (int k = 0; k < (KMAX <= 8 ? KMAX : 1); ++k)
            if (row0 + rl + k * rows_par < row1) av[k] = *reinterpret_cast<const bf16x8*>(add + (size_t)(row0 + rl + k * rows_par) * Cs);
    }
    // every storing wave drains its write-through stores, the workgroup meets, one lane signals and waits
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(p.cnt + (size_t)n * GN_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 0;
        for (unsigned spins = 0; spins < (1u << 24); ++spins) {
            if (__hip_atomic_load(p.cnt + (size_t)n * GN_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)p.wchunks) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        ok_s = ok;
    }
    __syncthreads();
    const bool ok = ok_s != 0;
    {
        // the image's wchunks x groups granules: ONE round trip (every thread up to 16 loads in flight) into LDS, then one thread per
        // group adds them in chunk order
        const int total = p.wchunks * p.groups;        // <= 128 x 32 = 4096
        const unsigned long long* gp = p.grp + (size_t)n * total;
        unsigned long long t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (tid + u * 256 < total) t[u] = __hip_atomic_load(gp + tid + u * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (tid + u * 256 < total) pool[tid + u * 256] = t[u];
    }
    __syncthreads();
    if (tid < p.groups) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < p.wchunks; ++k) {
            const unsigned long long t = pool[k * p.groups + tid];
            a += __uint_as_float((unsigned)t);
            b += __uint_as_float((unsigned)(t >> 32));
        }
        const float cnt = (float)p.HW * p.cpg;
        m1_s[tid] = a / cnt;
        m2_s[tid] = b / cnt;
    }
    __syncthreads();
    if (active) {
        float m1[8], m2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / p.cpg;
            m1[e] = m1_s[g]; m2[e] = m2_s[g];
        }
        if (KMAX > 8 && add) {
            // long form: the additive rows are fetched here, in two batches
#pragma unroll
            for (int k0 = 0; k0 < KMAX; k0 += 8) {
                bf16x8 a8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (row0 + rl + (k0 + u) * rows_par < row1)
                        a8[u] = *reinterpret_cast<const bf16x8*>(add + (size_t)(row0 + rl + (k0 + u) * rows_par) * Cs);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u;
                    const int rr = row0 + rl + k * rows_par;
                    if (k < KMAX && rr < row1) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = ((float)v[k][e] - mu[e]) * rs[e];
                            float dyy = (float)d[k][e];
                            if (p.silu) dyy *= gn_dsilu(xh * A[e] + Bv[e]);
                            float dx = rs[e] * (A[e] * dyy - m1[e] - xh * m2[e]) + (float)a8[u][e];
                            o[e] = ok ? (bf16)dx : (bf16)__builtin_nanf("");
                        }
                        *reinterpret_cast<bf16x8*>(dst + (size_t)rr * Cs) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int rr = row0 + rl + k * rows_par;
                if (rr < row1) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xh = ((float)v[k][e] - mu[e]) * rs[e];
                        float dyy = (float)d[k][e];
                        if (p.silu) dyy *= gn_dsilu(xh * A[e] + Bv[e]);
                        float dx = rs[e] * (A[e] * dyy - m1[e] - xh * m2[e]);
                        if (KMAX <= 8 && add) dx += (float)av[KMAX <= 8 ? k : 0][e];
                        o[e] = ok ? (bf16)dx : (bf16)__builtin_nanf("");
                    }
                    *reinterpret_cast<bf16x8*>(dst + (size_t)rr * Cs) = o;
                }
            }
        }
    }
    // g_out: every workgroup of the image adds the chunks' (G0, G1) pairs of ITS share of the channels — one round trip of agent-scope
    // loads into LDS, then one thread per channel adds them in chunk order
    __syncthreads();                 // pool: the group granules have been consumed
    {
        const int cpw = (C + p.wchunks - 1) / p.wchunks;
        const int c_lo = w * cpw, c_n = min(C, c_lo + cpw) - c_lo;         // may be <= 0 for the last workgroups
        const int total = c_n > 0 ? c_n * p.wchunks : 0;                    // <= C + wchunks <= 2176
        const unsigned long long* cp = p.cpart + (size_t)n * p.wchunks * C + c_lo;
        unsigned long long t[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = tid + u * 256;           // i = k * c_n + j: chunk k, channel c_lo + j (consecutive threads: consecutive channels)
            if (i < total) t[u] = __hip_atomic_load(cp + (size_t)(i / c_n) * C + (i % c_n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 9; ++u)
            if (tid + u * 256 < total) pool[tid + u * 256] = t[u];
        __syncthreads();
        for (int j = tid; j < c_n; j += 256) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < p.wchunks; ++k) {
                const unsigned long long q = pool[k * c_n + j];
                a += __uint_as_float((unsigned)q);
                b += __uint_as_float((unsigned)(q >> 32));
            }
            p.g_out[(size_t)n * C + c_lo + j] = ok ? a : __builtin_nanf("");
            p.g_out[((size_t)p.N + n) * C + c_lo + j] = ok ? b : __builtin_nanf("");
        }
    }
}

// work split of the one-launch backward: rows per workgroup = KMAX (8, else 16) x rows in parallel; 0 = not served (too many work
// chunks per image: LSUN-size maps keep the two-launch form)
static inline int gn_fused_plan(int HW, int C, int* wrows, int* wchunks) {
    const int rows_par = 256 / (C / 8);
    if (rows_par < 1) return 0;
    for (int kmax = 8; kmax <= 16; kmax *= 2) {
        const int wr = kmax * rows_par, wc = (HW + wr - 1) / wr;
        if (wc <= (kmax == 8 ? 64 : 128)) {
            *wrows = wr;
            *wchunks = wc;
            return kmax;
        }
    }
    return 0;
}

// 1 when the one-pass register-resident kernel (dxmi_groupnorm_silu_fwd) can serve the shape, else 0
// (callers then use dxmi_groupnorm_generic_fwd).  Same slicing walk as the launcher below.
extern "C" int dxmi_groupnorm_silu_supported(int32_t C0, int32_t C1, int32_t HW, int32_t groups) {
    const int C = C0 + C1;
    if (groups <= 0 || groups > 32 || C % groups != 0) return 0;
    const int cpg = C / groups;
    if (cpg % 4 != 0 || C0 % 4 != 0) return 0;
    const int VEC = (cpg % 8 == 0 && C0 % 8 == 0) ? 8 : 4;
    const int max_pieces = VEC == 8 ? 16 : 32;
    for (int slices = 1;; slices *= 2) {
        if (groups % slices != 0) return 0;
        const int ppp = C / slices / VEC;
        const int unit = ppp / gcd(ppp, 64) * 64;
        if (unit > 512) return 0;
        const long total = (long)HW * ppp;
        int threads = (512 / unit) * unit;
        if (total < threads) threads = (int)((total + unit - 1) / unit) * unit;
        const int pieces = (int)((total + threads - 1) / threads);
        if (pieces <= max_pieces) return 1;
        if (slices == groups) return pieces <= 32;
    }
}

// 1 when dxmi_groupnorm_silu_bwd can slice the shape (its plan is stricter than the forward's: x AND dy stay resident),
// else 0: callers then use dxmi_groupnorm_generic_bwd.
extern "C" int dxmi_groupnorm_silu_bwd_supported(int32_t C0, int32_t C1, int32_t HW, int32_t groups) {
    const int C = C0 + C1;
    if (groups <= 0 || groups > 32 || C % groups != 0) return 0;
    const int cpg = C / groups;
    if (cpg % 4 != 0 || C0 % 4 != 0) return 0;
    int VEC, slices, threads, pieces, ppp;
    return gn_plan(C, C0, HW, groups, 8, 16, &VEC, &slices, &threads, &pieces, &ppp) == 0 && pieces <= 16;
}

extern "C" int dxmi_groupnorm_silu_fwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const float* gamma,
                                       const float* beta, void* out, int32_t N, int32_t HW, int32_t groups, float eps,
                                       int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(in0 && out && gamma && beta, "dxmi_groupnorm_silu_fwd: null pointer");
    DXMI_CHECK_ARG(C1 == 0 || in1, "dxmi_groupnorm_silu_fwd: C1>0 but in1 NULL");
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0, "dxmi_groupnorm_silu_fwd: C=%d groups=%d", C, groups);
    const int cpg = C / groups;
    DXMI_CHECK_ARG(cpg % 4 == 0 && C0 % 4 == 0, "dxmi_groupnorm_silu_fwd: channels per group (%d) must be a multiple of 4", cpg);
    const int VEC = (cpg % 8 == 0 && C0 % 8 == 0) ? 8 : 4;
    // choose the number of channel slices (whole groups each) so a thread holds <= 16 pieces (VEC 8)
    // or <= 32 pieces (VEC 4) at ~512 threads.
    int slices = 1;
    const int max_pieces = VEC == 8 ? 16 : 32;
    int threads = 0, pieces = 0, ppp = 0;
    for (;; slices *= 2) {
        DXMI_CHECK_ARG(groups % slices == 0, "dxmi_groupnorm_silu_fwd: cannot slice %d groups for HW=%d C=%d", groups, HW, C);
        const int Csl = C / slices;
        ppp = Csl / VEC;
        const int unit = ppp / gcd(ppp, 64) * 64;  // lcm(ppp, 64)
        DXMI_CHECK_ARG(unit <= 512, "dxmi_groupnorm_silu_fwd: slice of %d channels unsupported", Csl);
        long total = (long)HW * ppp;
        threads = (512 / unit) * unit;
        if (total < threads) threads = (int)((total + unit - 1) / unit) * unit;
        pieces = (int)((total + threads - 1) / threads);
        if (pieces <= max_pieces || slices == groups) break;
    }
    DXMI_CHECK_ARG(pieces <= 32, "dxmi_groupnorm_silu_fwd: tensor slice too large (HW=%d C=%d)", HW, C);
    GnArgs a;
    a.in0 = (const bf16*)in0; a.in1 = (const bf16*)in1; a.gamma = gamma; a.beta = beta; a.out = (bf16*)out;
    a.C0 = C0; a.C1 = C1; a.HW = HW; a.groups = groups; a.slices = slices; a.ppp = ppp; a.cpg = cpg;
    a.gps = groups / slices; a.eps = eps; a.silu = apply_silu;
    static const int gn_xcd_env = getenv("DXMI_GN_XCD") ? atoi(getenv("DXMI_GN_XCD")) : 1;
    a.xcd = gn_xcd_env >= 2;      // forward: no measured gain (31.9 vs 33.6 us at 256 x 32x32x128), hardware order unless DXMI_GN_XCD=2
    const int ppt = cpg / VEC;
    a.fast = (ppp <= 64) && ((ppp & (ppp - 1)) == 0) && ((ppt & (ppt - 1)) == 0);
    hipStream_t st = (hipStream_t)stream;
    return VEC == 8 ? launch_gn<8>(a, N, pieces, threads, st) : launch_gn<4>(a, N, pieces, threads, st);
}

static inline int64_t gn_round16(int64_t b) { return (b + 15) / 16 * 16; }

extern "C" int64_t dxmi_groupnorm_generic_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
    const int chunks = gn_gen_chunks(HW, N);
    const int64_t two_launch = (int64_t)N * chunks * 32 * 2 * 4 + (int64_t)N * chunks * C * 2 * 4;
    int wrows = 0, wchunks = 0;
    if (C < 8 || !gn_fused_plan(HW, C, &wrows, &wchunks)) return two_launch;
    // one-launch form: [counters, padded to 16 bytes][statistics partials][group granules][channel granules]
    const int64_t fused = (int64_t)N * GN_CNT_STRIDE * 4 + (int64_t)N * chunks * 32 * 2 * 4 + (int64_t)N * wchunks * 32 * 8 + (int64_t)N * wchunks * C * 8;
    return fused > two_launch ? fused : two_launch;
}

// g_out: fp32 [2][N][C] (G0, G1 above).  dx1 / add0 / add1 / scale_shift may be NULL.
// fwd_stats: the statistics partials the forward left at the start of ITS workspace (dxmi_groupnorm_generic_workspace_bytes of
// them), kept by the caller — the statistics pass over the input is then skipped; NULL: recomputed here.
extern "C" int dxmi_groupnorm_generic_bwd_saved(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                                                const void* add0, const void* add1, const float* gamma, const float* beta,
                                                const float* scale_shift, int32_t ss_ld, void* dx0, void* dx1, float* g_out,
                                                const float* fwd_stats, void* workspace, int32_t N, int32_t HW, int32_t groups,
                                                float eps, int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(in0 && dy && gamma && beta && dx0 && g_out && workspace, "dxmi_groupnorm_generic_bwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && HW > 0 && C0 > 0 && C1 >= 0, "dxmi_groupnorm_generic_bwd: empty or negative shape (N %d, HW %d, C %d+%d)", N, HW, C0, C1);
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0 && C0 % 8 == 0 && C1 % 8 == 0 && (C1 == 0 || (in1 && dx1)) && C <= 2048,
                   "dxmi_groupnorm_generic_bwd: C0=%d C1=%d groups=%d", C0, C1, groups);
    const int chunks = gn_gen_chunks(HW, N);
    float* part = (float*)workspace;
    GnGenArgs f;
    f.in0 = (const bf16*)in0; f.in1 = (const bf16*)in1; f.gamma = gamma; f.beta = beta; f.ss = scale_shift; f.ss_ld = ss_ld;
    f.out = nullptr; f.part = part; f.C0 = C0; f.C1 = C1; f.HW = HW; f.groups = groups; f.cpg = C / groups;
    f.chunks = chunks; f.rows_per_chunk = (HW + chunks - 1) / chunks; f.eps = eps; f.silu = apply_silu;
    GnGenBwdArgs a;
    a.in0 = (const bf16*)in0; a.in1 = (const bf16*)in1; a.dy = (const bf16*)dy; a.add0 = (const bf16*)add0; a.add1 = (const bf16*)add1;
    a.gamma = gamma; a.beta = beta; a.ss = scale_shift; a.dx0 = (bf16*)dx0; a.dx1 = (bf16*)dx1;
    a.part = fwd_stats ? const_cast<float*>(fwd_stats) : part;      // read-only in the backward kernels
    a.gpart = part + (size_t)N * chunks * 32 * 2; a.g_out = g_out;
    a.C0 = C0; a.C1 = C1; a.HW = HW; a.groups = groups; a.cpg = C / groups; a.chunks = chunks; a.rows_per_chunk = f.rows_per_chunk;
    a.ss_ld = ss_ld; a.N = N; a.eps = eps; a.silu = apply_silu;
    hipStream_t st = (hipStream_t)stream;
    int wrows = 0, wchunks = 0;
    // knob gn_bwd_fused: 1 = where the one-launch form measured faster (maps of <= 256 pixels: every workgroup reads the image's
    // wchunks x 32 group granules with agent-scope loads, a cost that grows with the SQUARE of the chunk count — 16 x 64x64x192:
    // 52 chunks 85 us, 104 chunks 153 us, against 49 us for the two launches; 16 x 8x8x1536: 29 against 80 us), 2 = wherever it
    // fits (tests), 0 = never
    const int fused_knob = dxmi_tuning("gn_bwd_fused");
    int kmax = (fused_knob >= 2 || (fused_knob == 1 && HW <= 256)) ? gn_fused_plan(HW, C, &wrows, &wchunks) : 0;
    // ... and only while the launch is ONE round of resident workgroups (two per CU: 190 registers): it is the latency chain of a small
    // launch that the hand-off shortens; at 256 images x 16x16x384 (1 792 workgroups) it measured 121 us against 79 us
    if (fused_knob == 1 && (long)N * wchunks > 512) kmax = 0;
    // residency: the workgroups of an image wait for each other, so an image's work chunks must fit the device at one workgroup per CU
    // whatever else holds (a 32-CU partition of the chip included); otherwise the two launches
    if (kmax && wchunks > dxmi_device_cus()) kmax = 0;
    if (kmax) {
        // one launch (gn_gen_bwd_fused_kernel) behind the node that zeroes the arrival counters
        char* base = (char*)workspace;
        const int64_t cnt_bytes = (int64_t)N * GN_CNT_STRIDE * 4;
        float* part2 = (float*)(base + cnt_bytes);
        a.cnt = (unsigned*)base;
        a.grp = (unsigned long long*)(base + cnt_bytes + (int64_t)N * chunks * 32 * 2 * 4);
        a.cpart = a.grp + (size_t)N * wchunks * 32;
        a.wchunks = wchunks; a.wrows = wrows;
        if (!fwd_stats) {
            f.part = part2;
            a.part = part2;
            hipLaunchKernelGGL(gn_gen_stats_kernel, dim3(N * chunks), dim3(256), 0, st, f);
            DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(stats)");
        }
        // The counters are zeroed by a KERNEL node, not by hipMemsetAsync: behind a memset node the replayed EDM train step was not
        // reproducible (three replays of the same step: v_loss 77.3227 / 77.3596 / 77.3130; python-issued launches and the kernel node:
        // identical every time) — on this ROCm a replayed graph does not order a memset node's write against the agent-scope atomics of
        // the kernel node behind it.
        hipLaunchKernelGGL(gn_zero_counters_kernel, dim3((N + 255) / 256), dim3(256), 0, st, a.cnt, N);
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(counters)");
        if (kmax == 8) hipLaunchKernelGGL(gn_gen_bwd_fused_kernel<8>, dim3(N * wchunks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(gn_gen_bwd_fused_kernel<16>, dim3(N * wchunks), dim3(256), 0, st, a);
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(fused)");
        return DXMI_OK;
    }
    if (!fwd_stats) {
        hipLaunchKernelGGL(gn_gen_stats_kernel, dim3(N * chunks), dim3(256), 0, st, f);
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(stats)");
    }
    hipLaunchKernelGGL(gn_gen_bwd_reduce_kernel, dim3(N * chunks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(reduce)");
    hipLaunchKernelGGL(gn_gen_bwd_apply_kernel, dim3(N * chunks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_groupnorm_generic_bwd(apply)");
    return DXMI_OK;
}

extern "C" int dxmi_groupnorm_generic_bwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                                          const void* add0, const void* add1, const float* gamma, const float* beta,
                                          const float* scale_shift, int32_t ss_ld, void* dx0, void* dx1, float* g_out,
                                          void* workspace, int32_t N, int32_t HW, int32_t groups, float eps,
                                          int32_t apply_silu, void* stream) {
    return dxmi_groupnorm_generic_bwd_saved(in0, C0, in1, C1, dy, add0, add1, gamma, beta, scale_shift, ss_ld, dx0, dx1, g_out, nullptr,
                                            workspace, N, HW, groups, eps, apply_silu, stream);
}

// One workgroup per image: the statistics prologue of gn_apply_kernel run ONCE per image instead of once per workgroup of the apply
// pass — block-statistics partials -> channel-pair sums (partial order) -> group sums (channel order) -> mean / rstd -> per-channel
// (scale, offset) with gamma / beta and the FiLM scale-shift folded in, written to ab[n][c][2].  Same operations in the same order
// as the prologue: the apply pass gives bit-identical results either way.
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnApplyArgs p) {
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ float2 pair_s[1024];
    const int C = p.C0 + p.C1;
    const int n = blockIdx.x, tid = threadIdx.x;
    auto pair_sums = [&](const float* st, int P, int nbs, int off) {
        const float2* const base = reinterpret_cast<const float2*>(st) + (size_t)n * P * nbs;
        for (int b = tid; b < nbs; b += 256) {
            float2 t[GN_APPLY_MAXP];
#pragma unroll
            for (int k = 0; k < GN_APPLY_MAXP; ++k)
                if (k < P) t[k] = base[(unsigned)(k * nbs + b)];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int k = 0; k < GN_APPLY_MAXP; ++k)
                if (k < P) {
                    s += t[k].x;
                    q += t[k].y;
                }
            for (int k = GN_APPLY_MAXP; k < P; ++k) {
                const float2 tk = base[(unsigned)(k * nbs + b)];
                s += tk.x;
                q += tk.y;
            }
            pair_s[off + b] = make_float2(s, q);
        }
    };
    pair_sums(p.st0, p.P0, p.C0 >> 1, 0);
    if (p.C1) pair_sums(p.st1, p.P1, p.C1 >> 1, p.C0 >> 1);
    __syncthreads();
    if (tid < p.groups) {
        const int bpg = p.cpg >> 1;
        float s = 0.f, q = 0.f;
        for (int b = tid * bpg; b < (tid + 1) * bpg; ++b) {
            s += pair_s[b].x;
            q += pair_s[b].y;
        }
        const float cnt = (float)p.HW * (float)p.cpg;
        const float m = s / cnt;
        mean_s[tid] = m;
        rstd_s[tid] = rsqrtf(fmaxf(q / cnt - m * m, 0.f) + p.eps);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int g = c / p.cpg;
        float a = rstd_s[g] * p.gamma[c];
        float b = p.beta[c] - mean_s[g] * a;
        if (p.ss) {
            const float sc = 1.f + p.ss[(size_t)n * p.ss_ld + c];
            a *= sc;
            b = b * sc + p.ss[(size_t)n * p.ss_ld + C + c];
        }
        reinterpret_cast<float2*>(p.ab)[(size_t)n * C + c] = make_float2(a, b);
    }
}

// Block statistics (per channel pair and partial: what the conv epilogues / gn_block_stats_kernel write, what gn_apply_kernel reads)
// -> the generic kernels' statistics partials [N][chunks][groups][2] (what gn_gen_bwd_* read as `fwd_stats`): the training forward of
// the ADM nets normalises with the streaming apply pass fed by its producers' statistics (round 6) and hands the generic backward the
// same sums in its own format — the whole image's (sum, sum of squares) per group in chunk 0, zeros in the other chunks (the
// backward adds the chunks in order).  One workgroup per image; pair sums in partial order, group sums in channel order.
__global__ __launch_bounds__(256) void gn_blockstats_to_generic_kernel(const float* __restrict__ st0, int P0, int C0, const float* __restrict__ st1,
                                                                      int P1, int C1, float* __restrict__ part, int groups, int chunks) {
    __shared__ float2 pair_s[1024];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int C = C0 + C1, cpg = C / groups;
    auto pair_sums = [&](const float* st, int P, int nbs, int off) {
        const float2* const base = reinterpret_cast<const float2*>(st) + (size_t)n * P * nbs;
        for (int b = tid; b < nbs; b += 256) {
            float s = 0.f, q = 0.f;
            for (int k = 0; k < P; ++k) {
                const float2 t = base[(size_t)k * nbs + b];
                s += t.x;
                q += t.y;
            }
            pair_s[off + b] = make_float2(s, q);
        }
    };
    pair_sums(st0, P0, C0 >> 1, 0);
    if (C1) pair_sums(st1, P1, C1 >> 1, C0 >> 1);
    __syncthreads();
    float* const out = part + (size_t)n * chunks * groups * 2;
    for (int i = tid; i < chunks * groups; i += 256) {
        const int k = i / groups, g = i - k * groups;
        float s = 0.f, q = 0.f;
        if (k == 0) {
            const int bpg = cpg >> 1;
            for (int b = g * bpg; b < (g + 1) * bpg; ++b) {
                s += pair_s[b].x;
                q += pair_s[b].y;
            }
        }
        out[(size_t)i * 2] = s;
        out[(size_t)i * 2 + 1] = q;
    }
}

// ---------------------------------------------------------------------------------------------
// Streaming apply + block statistics (see gn_apply_kernel)
static inline int gn_stats_chunks(int HW) { return HW >= 512 ? HW / 256 : 1; }

extern "C" int dxmi_gn_block_stats_partials(int32_t HW) { return gn_stats_chunks(HW); }

extern "C" int dxmi_gn_block_stats(const void* x, float* stats, int32_t N, int32_t HW, int32_t C, void* stream) {
    DXMI_CHECK_ARG(x && stats && N > 0 && HW > 0, "dxmi_gn_block_stats: null pointer / empty shape");
    DXMI_CHECK_ARG(C % 8 == 0 && C >= 8 && C <= 2048, "dxmi_gn_block_stats: C=%d must be a multiple of 8, <= 2048", C);
    const int chunks = gn_stats_chunks(HW);
    const int rpc = (HW + chunks - 1) / chunks;
    hipLaunchKernelGGL(gn_block_stats_kernel, dim3(N * chunks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, stats, HW, C,
                       chunks, rpc);
    DXMI_CHECK_LAUNCH("dxmi_gn_block_stats");
    return DXMI_OK;
}

// Parameter and FiLM gradients of a scale-shift GroupNorm from the backward's per-image sums g [2][N][C] (G0 = sum dyy, G1 = sum
// dyy * xh; the kernels above): with s = 1 + scale[n, c]
//   d_ss[n, c] = G1 gamma + G0 beta ; d_ss[n, C + c] = G0 ; dgamma[c] = sum_n G1 s ; dbeta[c] = sum_n G0 s       (images in order)
// One thread per channel, one launch (it was seven torch launches per FiLM GroupNorm of the EDM backward).
__global__ __launch_bounds__(256) void gn_ss_grads_kernel(const float* __restrict__ g, const float* __restrict__ ss, int ss_ld,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ d_ss, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int N, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float ga = gamma[c], be = beta[c];
    float dg = 0.f, db = 0.f;
    for (int n0 = 0; n0 < N; n0 += 8) {
        float g0[8], g1[8], sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (n0 + u < N) {
                g0[u] = g[(size_t)(n0 + u) * C + c];
                g1[u] = g[((size_t)N + n0 + u) * C + c];
                sc[u] = 1.f + ss[(size_t)(n0 + u) * ss_ld + c];
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (n0 + u < N) {
                d_ss[(size_t)(n0 + u) * 2 * C + c] = g1[u] * ga + g0[u] * be;
                d_ss[(size_t)(n0 + u) * 2 * C + C + c] = g0[u];
                dg += g1[u] * sc[u];
                db += g0[u] * sc[u];
            }
    }
    dgamma[c] = dg;
    dbeta[c] = db;
}

extern "C" int dxmi_gn_ss_grads(const float* g, const float* scale_shift, int32_t ss_ld, const float* gamma, const float* beta,
                                float* d_scale_shift, float* dgamma, float* dbeta, int32_t N, int32_t C, void* stream) {
    DXMI_CHECK_ARG(g && scale_shift && gamma && beta && d_scale_shift && dgamma && dbeta, "dxmi_gn_ss_grads: null pointer");
    DXMI_CHECK_ARG(N > 0 && C > 0 && ss_ld >= 2 * C, "dxmi_gn_ss_grads: N=%d C=%d ss_ld=%d", N, C, ss_ld);
    hipLaunchKernelGGL(gn_ss_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, scale_shift, ss_ld, gamma, beta,
                       d_scale_shift, dgamma, dbeta, N, C);
    DXMI_CHECK_LAUNCH("dxmi_gn_ss_grads");
    return DXMI_OK;
}

extern "C" int dxmi_gn_blockstats_to_generic(const float* stats0, int32_t P0, int32_t C0, const float* stats1, int32_t P1, int32_t C1,
                                             float* generic_stats, int32_t N, int32_t HW, int32_t groups, void* stream) {
    DXMI_CHECK_ARG(stats0 && generic_stats && N > 0 && HW > 0 && P0 > 0 && C0 > 0 && C1 >= 0, "dxmi_gn_blockstats_to_generic: bad arguments");
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0 && (C / groups) % 2 == 0 && C0 % 2 == 0 && C1 % 2 == 0 && C <= 2048 &&
                   (C1 == 0 || (stats1 && P1 > 0)), "dxmi_gn_blockstats_to_generic: C0=%d C1=%d groups=%d (even channels per group)", C0, C1, groups);
    hipLaunchKernelGGL(gn_blockstats_to_generic_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, stats0, P0, C0, stats1, P1, C1, generic_stats,
                       groups, gn_gen_chunks(HW, N));
    DXMI_CHECK_LAUNCH("dxmi_gn_blockstats_to_generic");
    return DXMI_OK;
}

extern "C" int dxmi_gn_stats_fold(const float* stats, float* out, int32_t N, int32_t P, int32_t C, int32_t group, void* stream) {
    DXMI_CHECK_ARG(stats && out && N > 0 && P > 0 && C > 0 && C % 2 == 0 && group > 0, "dxmi_gn_stats_fold: bad arguments");
    const int PG = (P + group - 1) / group;
    hipLaunchKernelGGL(gn_stats_fold_kernel, dim3((C + 255) / 256, N * PG), dim3(256), 0, (hipStream_t)stream, stats, out, P, C, group);
    DXMI_CHECK_LAUNCH("dxmi_gn_stats_fold");
    return DXMI_OK;
}

static int gn_apply_impl(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                         const float* stats1, int32_t P1, const float* gamma, const float* beta,
                         const float* scale_shift, int32_t ss_ld, void* out, float* ab, int32_t N, int32_t HW, int32_t groups,
                         float eps, int32_t apply_silu, void* stream);

extern "C" int dxmi_groupnorm_apply(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                                    const float* stats1, int32_t P1, const float* gamma, const float* beta,
                                    const float* scale_shift, int32_t ss_ld, void* out, int32_t N, int32_t HW, int32_t groups,
                                    float eps, int32_t apply_silu, void* stream) {
    return gn_apply_impl(in0, C0, stats0, P0, in1, C1, stats1, P1, gamma, beta, scale_shift, ss_ld, out, nullptr, N, HW, groups, eps,
                         apply_silu, stream);
}

extern "C" int dxmi_groupnorm_apply_split(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                                          const float* stats1, int32_t P1, const float* gamma, const float* beta,
                                          const float* scale_shift, int32_t ss_ld, void* out, float* ab_workspace, int32_t N,
                                          int32_t HW, int32_t groups, float eps, int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(ab_workspace, "dxmi_groupnorm_apply_split: null workspace (N * C * 2 floats)");
    return gn_apply_impl(in0, C0, stats0, P0, in1, C1, stats1, P1, gamma, beta, scale_shift, ss_ld, out, ab_workspace, N, HW, groups,
                         eps, apply_silu, stream);
}

static int gn_apply_impl(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                         const float* stats1, int32_t P1, const float* gamma, const float* beta,
                         const float* scale_shift, int32_t ss_ld, void* out, float* ab, int32_t N, int32_t HW, int32_t groups,
                         float eps, int32_t apply_silu, void* stream) {
    DXMI_CHECK_ARG(in0 && stats0 && gamma && beta && out && P0 > 0, "dxmi_groupnorm_apply: null pointer");
    DXMI_CHECK_ARG(C1 == 0 || (in1 && stats1 && P1 > 0), "dxmi_groupnorm_apply: C1>0 needs in1 and stats1");
    const int C = C0 + C1;
    DXMI_CHECK_ARG(groups > 0 && groups <= 32 && C % groups == 0, "dxmi_groupnorm_apply: C=%d groups=%d", C, groups);
    const int cpg = C / groups;
    DXMI_CHECK_ARG(cpg % 2 == 0 && C0 % 8 == 0 && C1 % 8 == 0 && C <= 2048,
                   "dxmi_groupnorm_apply: channels per group (%d) must be even, C0/C1 (%d/%d) multiples of 8", cpg, C0, C1);
    GnApplyArgs a;
    a.in0 = (const bf16*)in0; a.in1 = (const bf16*)in1; a.st0 = stats0; a.st1 = stats1; a.gamma = gamma; a.beta = beta;
    a.ss = scale_shift; a.ss_ld = ss_ld;
    a.out = (bf16*)out; a.C0 = C0; a.C1 = C1; a.HW = HW; a.groups = groups; a.cpg = cpg; a.P0 = P0; a.P1 = P1;
    a.eps = eps; a.silu = apply_silu;
    static const int nt_env = getenv("DXMI_GN_APPLY_NT") ? atoi(getenv("DXMI_GN_APPLY_NT")) : 0;
    a.nt = nt_env;
    // a workgroup streams ~32 KB: two trips of U = 4 rows x (256 / (C/8)) row lanes
    static const int u_env = getenv("DXMI_GN_APPLY_U") ? atoi(getenv("DXMI_GN_APPLY_U")) : 4;            // tuning: loads in flight per thread
    static const int trips_env = getenv("DXMI_GN_APPLY_TRIPS") ? atoi(getenv("DXMI_GN_APPLY_TRIPS")) : 2;      // (in-situ sweep: 2 trips of 4 = 32 KB per workgroup slightly ahead of 4)
    const int U = u_env == 8 ? 8 : (u_env == 2 ? 2 : 4);
    const int rows_par = 256 / (C / 8);
    int rpc = trips_env * U * rows_par;
    if (rpc > HW) rpc = HW;
    a.chunks = (HW + rpc - 1) / rpc;
    a.rows_per_chunk = rpc;
    a.ab = ab;
    if (ab) {
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, a);
        DXMI_CHECK_LAUNCH("dxmi_groupnorm_apply_split(finalize)");
    }
    if (U == 8) hipLaunchKernelGGL(gn_apply_kernel<8>, dim3(N * a.chunks), dim3(256), 0, (hipStream_t)stream, a);
    else if (U == 2) hipLaunchKernelGGL(gn_apply_kernel<2>, dim3(N * a.chunks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(N * a.chunks), dim3(256), 0, (hipStream_t)stream, a);
    DXMI_CHECK_LAUNCH("dxmi_groupnorm_apply");
    return DXMI_OK;
}
