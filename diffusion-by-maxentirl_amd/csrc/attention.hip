// Self-attention for the U-Net attention blocks on gfx950: MFMA QK^T and PV, online softmax.
//
// Work split: a workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32
// queries.  Keys/values are streamed in blocks of 64 through LDS.
//   S^T[key][query] = K . Q^T      A = K block (LDS, row = key, k = d), B = Q (registers)
//   O^T[d][query]  += V^T . P^T    A = V^T tile, read from the ROW-MAJOR V image through transposing LDS
//                                  loads (ds_read_b64_tr_b16; staging V is plain 16-byte writes),
//                                  B = P^T taken straight from the S^T accumulators (the k
//                                  order of an accumulator tile is permuted: element j of
//                                  lane-half h is key 16s + 8(j>>2) + 4h + (j&3); the V^T
//                                  fragment is read in the same order).
// Wide heads (D >= 128) run one wave per SIMD, so block kb+1 of K/V is fetched into registers while
// block kb is computed.
// With the query on the lane (column of every accumulator), the running max / sum are
// lane-local apart from one lane<->lane+32 exchange.
//
// Reference: AttnBlock.forward models/DxMI/unet_small.py:175-187 (scale C^-0.5, softmax over
// keys); QKVAttentionLegacy models/cm/unet.py:413-441.
#include "conv_common.h"
#include <stdlib.h>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

// Workgroup b runs on XCD b % 8 (round-robin dispatch, one L2 per XCD): with consecutive block ids the query blocks of one (image,
// head) — which all stream the SAME K and V — sat on different XCDs and K / V were fetched once per query block.  Logical ids that are
// consecutive within an XCD put them behind one L2 (any grid size).
__device__ __forceinline__ int attn_xcd_logical(int b, int G, int on) {
    if (!on) return b;
    const int x = b & 7, k = b >> 3;
    const int q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + k;
}

struct AttnArgs {
    const bf16* qkv;
    bf16* out;
    int N, T, C, heads;
    int q_off, k_off, v_off, head_stride;  // channel offsets inside the 3C-wide row
    float scale;
    // attention256_kernel<true>: proj_out (1x1 conv 256 -> 256) + bias + residual fused behind the attention
    const bf16* wproj;      // packed by dxmi_pack_attn_proj_weight: [8 cout blocks][16 k-steps][64 lanes][8]
    const float* pbias;     // [256]
    const bf16* res;        // [N,256,256] (the AttnBlock's input x)
    float* gn_stats;        // optional: GroupNorm block statistics of the output, [N][8][128][2] (a partial per 32 tokens)
    int xcd;             // XCD-aware block order (round 6): the query blocks of one (image, head) share K / V behind one L2
    float* lse;          // optional (attention_kernel<D>, attention64_kernel): row log-sum-exp in the log2 domain, [N][heads][T] —
                         // max + log2(sum) of scale * log2(e) * S, what attn_bwd_dq_kernel otherwise recomputes in a sweep of its own
};

#ifndef ATTN_WPE
#define ATTN_WPE 3
#endif
template <int D, bool PF>
__global__ __launch_bounds__(256, D == 64 ? ATTN_WPE : 1) void attention_kernel(AttnArgs p) {
    constexpr int KB = 64;               // keys per block
    constexpr int DK = D / 16;           // k-steps over d
    constexpr int DB = D / 32;           // 32-row blocks of O^T
    constexpr int KPITCH = D * 2 + 16;   // bytes per key row of the K image
    constexpr int VPITCH = D * 2 + 64;   // bytes per key row of the V image (row-major; PV reads it through transposing LDS loads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + KB * KPITCH;

    const int tid = threadIdx.x, lane = tid & 63;
    // tr-read lane roles (ds_read_b64_tr_b16): 16-lane group g -> d half (g&1), key half (g>>1); lane 4q+pp -> key row q, d 4pp..
    const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
    const int tr_doff = (16 * (trg & 1) + 4 * trp) * 2;
    const int tr_krow = 4 * (trg >> 1) + trq;   // key order of a k-step follows the S^T accumulator layout: 8(j>>2) + 4h + (j&3)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = (p.T + 127) / 128;
    const int bid = attn_xcd_logical(blockIdx.x, gridDim.x, p.xcd);
    const int qb = bid % qblocks;
    const int hd = (bid / qblocks) % p.heads;
    const int n = bid / (qblocks * p.heads);
    const int C3 = 3 * p.C;
    const bf16* base = p.qkv + (size_t)n * p.T * C3;
    const int qc = p.q_off + hd * p.head_stride;
    const int kc = p.k_off + hd * p.head_stride;
    const int vc = p.v_off + hd * p.head_stride;

    const int h = lane >> 5;
    const int query = qb * 128 + wave * 32 + (lane & 31);
    const bool qvalid = query < p.T;

    // Q fragments (B operand): lane = query, elements d = ks*16 + 8h + j
    bf16x8 qf[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
        qf[ks] = qvalid ? *reinterpret_cast<const bf16x8*>(base + (size_t)query * C3 + qc + ks * 16 + 8 * h) : z;
    }

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;      // logits in the log2 domain: exp(x) = exp2(x log2 e)

    const int nkb = (p.T + KB - 1) / KB;
    constexpr int PIECES = KB * (D / 8);   // 16-byte pieces per operand block
    constexpr int NP = PIECES / 256;
    bf16x8 kreg[NP], vreg[NP];
    auto gload = [&](int kb) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int i = tid + q * 256;
            const int key = i / (D / 8), pc = i % (D / 8);
            const int gk = kb * KB + key;
#pragma unroll
            for (int e = 0; e < 8; ++e) { kreg[q][e] = (bf16)0.f; vreg[q][e] = (bf16)0.f; }
            if (gk < p.T) {
                kreg[q] = *reinterpret_cast<const bf16x8*>(base + (size_t)gk * C3 + kc + pc * 8);
                vreg[q] = *reinterpret_cast<const bf16x8*>(base + (size_t)gk * C3 + vc + pc * 8);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int i = tid + q * 256;
            const int key = i / (D / 8), pc = i % (D / 8);
            *reinterpret_cast<bf16x8*>(kimg + key * KPITCH + pc * 16) = kreg[q];
            *reinterpret_cast<bf16x8*>(vimg + key * VPITCH + pc * 16) = vreg[q];
        }
    };
    for (int kb = 0; kb < nkb; ++kb) {
        // ---- K block (row-major) and V block (transposed) in LDS, zero-filled past T.  Wide heads (D >= 128: one
        // wave per SIMD, nothing else hides latency) fetch block kb+1 into registers while block kb is computed.
        if constexpr (PF) {
            if (kb == 0) {
                gload(0);
                lstore();
                __syncthreads();
            }
            if (kb + 1 < nkb) gload(kb + 1);
        } else {
            __syncthreads();
            gload(kb);
            lstore();
            __syncthreads();
        }

        // ---- S^T = K . Q^T for the 2 x 32 keys of this block
        f32x16 s[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kimg + (kh * 32 + (lane & 31)) * KPITCH + ks * 32 + h * 16);
                s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kh], 0, 0, 0);
            }
        }
        // ---- online softmax (query on the lane; this lane holds 32 of the 64 keys).  Round 4: the kernel is VALU-bound on the
        // narrow heads (32 exponentials per lane and block beside 16 MFMAs), so the per-element work is trimmed: logits in the
        // log2 domain (scale * log2(e) folded into one multiply, v_exp_f32 directly), the key mask only in a ragged last block,
        // the accumulator rescale skipped when no lane's maximum moved.
        float mx = -INFINITY;
        if ((kb + 1) * KB <= p.T) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = s[kh][r] * sc2;
                    s[kh][r] = v;
                    mx = fmaxf(mx, v);
                }
        } else {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * KB + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float v = (key < p.T) ? s[kh][r] * sc2 : -INFINITY;
                    s[kh][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        float psum = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kh][r] - m_new);
                s[kh][r] = e;
                psum += e;
            }
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {      // some lane's running maximum moved: rescale (always on the first block)
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);  // m = -inf on the first block -> 0
            l = l * alpha + psum;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        } else {
            l += psum;
        }
        m = m_new;

        // ---- O^T += V^T . P^T ; 4 k-steps of 16 keys
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kh = st >> 1, sl = st & 1;  // accumulator tile, k-step inside it
            bf16x8 pb;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pb[jj] = (bf16)s[kh][8 * sl + jj];
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                // A = V^T tile [32 d][16 keys]: lane holds d = db*32 + (lane&31), keys 8h..8h+7 of this k-step
                const char* row = vimg + (kh * 32 + 16 * sl + tr_krow) * VPITCH + db * 64 + tr_doff;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row + 8 * VPITCH));
                bf16x8 a;
                short* as = reinterpret_cast<short*>(&a);
#pragma unroll
                for (int e = 0; e < 4; ++e) { as[e] = lo[e]; as[4 + e] = hi[e]; }
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o[db], 0, 0, 0);
            }
        }
        if constexpr (PF) {
            if (kb + 1 < nkb) {
                __syncthreads();   // every wave is done with block kb's LDS images
                lstore();
                __syncthreads();
            }
        }
    }

    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (p.lse && qvalid && h == 0) p.lse[((size_t)n * p.heads + hd) * p.T + query] = m + __log2f(l);
    if (qvalid) {
        bf16* orow = p.out + ((size_t)n * p.T + query) * p.C + hd * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16)(o[db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * h) = ov;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 64-wide heads on long sequences (the ADM nets' 32x32 and 16x16 AttentionBlocks: T = 1024 / 256, models/cm/unet.py:413-441), round 4.
// attention_kernel<64> pays per key block for two barriers and an exposed global load, with 32 queries per wave.  Here:
//   * a wave owns 64 queries (two 32-query tiles): every K fragment read from LDS feeds two MFMAs and the per-block costs
//     (barrier, staging) are shared by twice the work;
//   * the K / V images are double-buffered: block kb + 1 is requested into registers before block kb is computed and written
//     to the other buffer after it, ONE barrier per block;
//   * logits stay raw in the accumulators: the running maximum is taken on them and scale * log2(e) rides the one FMA in front
//     of the exponential (exp2(s * c - m)), a multiply and a subtract less per element.
// Measured (N = 100, T = 1024, 6 heads): 335 us on attention_kernel<64> at three waves per SIMD -> 290 us.  The kernel is bound by
// its per-wave dependency chain (S MFMAs -> maximum -> exponentials -> PV MFMAs, two waves per SIMD), not by an execution unit:
// timing-only builds without the exponentials take as long, without the S MFMAs or without staging + barrier 24 % less each.  Two
// re-orderings that hipcc accepted did not shorten it: the two tiles skewed by half a block over a three-image K
// ring (308 us), and that skew with sched_group_barrier slots "one K read, one MFMA, an eighth of the exponentials" (the
// scheduler clustered the MFMAs regardless and spilled 16 registers).
// Layouts (fragment orders, transposing V reads, padded pitches) are attention_kernel's.  T % 256 == 0 (a workgroup = 256 queries).
constexpr int AT64_KPITCH = 64 * 2 + 16, AT64_VPITCH = 64 * 2 + 64, AT64_KIMG = 64 * AT64_KPITCH, AT64_BUF = AT64_KIMG + 64 * AT64_VPITCH;
constexpr int AT64_LDS = 2 * AT64_BUF;
__global__ __launch_bounds__(256, 2) void attention64_kernel(AttnArgs p) {
    constexpr int D = 64, KB = 64, KPITCH = AT64_KPITCH, VPITCH = AT64_VPITCH, KIMG = AT64_KIMG, BUF = AT64_BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
    const int tr_doff = (16 * (trg & 1) + 4 * trp) * 2;
    const int tr_krow = 4 * (trg >> 1) + trq;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = p.T >> 8;
    const int bid = attn_xcd_logical(blockIdx.x, gridDim.x, p.xcd);
    const int qb = bid % qblocks;
    const int hd = (bid / qblocks) % p.heads;
    const int n = bid / (qblocks * p.heads);
    const int C3 = 3 * p.C;
    const bf16* base = p.qkv + (size_t)n * p.T * C3;
    const int qc = p.q_off + hd * p.head_stride, kc = p.k_off + hd * p.head_stride, vc = p.v_off + hd * p.head_stride;
    const int h = lane >> 5;
    const int query0 = qb * 256 + wave * 64 + (lane & 31);      // tile t: query0 + 32 t

    bf16x8 qf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[t][ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)(query0 + 32 * t) * C3 + qc + ks * 16 + 8 * h);

    f32x16 o[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][db][r] = 0.f;
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const float sc2 = p.scale * 1.4426950408889634f;      // exp(x * scale) = exp2(x * sc2); scale > 0

    const int nkb = p.T / KB;
    const int key0 = tid >> 3, pc = tid & 7;               // staging: pieces tid and tid + 256 (keys key0 and key0 + 32)
    const bf16* const gk = base + (size_t)key0 * C3 + kc + pc * 8;
    const bf16* const gv = base + (size_t)key0 * C3 + vc + pc * 8;
    bf16x8 kreg[2], vreg[2];
    auto gload = [&](int kb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            kreg[q] = *reinterpret_cast<const bf16x8*>(gk + (size_t)(kb * KB + 32 * q) * C3);
            vreg[q] = *reinterpret_cast<const bf16x8*>(gv + (size_t)(kb * KB + 32 * q) * C3);
        }
    };
    auto lstore = [&](char* buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<bf16x8*>(buf + (key0 + 32 * q) * KPITCH + pc * 16) = kreg[q];
            *reinterpret_cast<bf16x8*>(buf + KIMG + (key0 + 32 * q) * VPITCH + pc * 16) = vreg[q];
        }
    };
    gload(0);
    lstore(smem);
    // every load so far (the Q fragments too) has landed before the loop: hipcc's wait-count pass otherwise carries the pending Q
    // loads into the loop and waits, in every trip, for counts that drain the NEXT block's prefetch in front of the first MFMAs
    __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0)
    __syncthreads();

    for (int kb = 0; kb < nkb; ++kb) {
        const char* const kimg = smem + (kb & 1) * BUF;
        const char* const vimg = kimg + KIMG;
        if (kb + 1 < nkb) gload(kb + 1);
        // ---- S^T = K . Q^T: 2 x 32 keys against both query tiles (each K fragment is read once)
        f32x16 s[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t][kh][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kimg + (kh * 32 + (lane & 31)) * KPITCH + ks * 32 + h * 16);
                s[0][kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[0][ks], s[0][kh], 0, 0, 0);
                s[1][kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[1][ks], s[1][kh], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // ---- online softmax of tile t (query on the lane; this lane holds 32 of the block's 64 keys)
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[t][kh][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m[t], mx * sc2);
            float psum = 0.f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][kh][r], sc2, -m_new));
                    s[t][kh][r] = e;
                    psum += e;
                }
            if (__builtin_amdgcn_ballot_w64(m_new != m[t]) != 0) {      // some lane's running maximum moved (always on the first block)
                const float alpha = __builtin_amdgcn_exp2f(m[t] - m_new);
                l[t] = l[t] * alpha + psum;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][db][r] *= alpha;
            } else {
                l[t] += psum;
            }
            m[t] = m_new;
            // ---- O^T += V^T . P^T ; 4 k-steps of 16 keys
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int kh = st >> 1, sl = st & 1;
                bf16x8 pb;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pb[jj] = (bf16)s[t][kh][8 * sl + jj];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* row = vimg + (kh * 32 + 16 * sl + tr_krow) * VPITCH + db * 64 + tr_doff;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row + 8 * VPITCH));
                    bf16x8 a;
                    short* as = reinterpret_cast<short*>(&a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { as[e] = lo[e]; as[4 + e] = hi[e]; }
                    o[t][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, o[t][db], 0, 0, 0);
                }
            }
        }
        if (kb + 1 < nkb) {
            lstore(smem + ((kb + 1) & 1) * BUF);      // last read in block kb - 1: every wave passed the barrier that ended it
            __syncthreads();
        }
    }

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float lt = l[t] + __shfl_xor(l[t], 32, 64);
        const float inv = 1.f / lt;
        if (p.lse && h == 0) p.lse[((size_t)n * p.heads + hd) * p.T + query0 + 32 * t] = m[t] + __log2f(lt);
        bf16* orow = p.out + ((size_t)n * p.T + query0 + 32 * t) * p.C + hd * D;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16)(o[t][db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * h) = ov;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Single-head 256-token x 256-channel attention (the CIFAR-10 U-Net's 16x16 AttnBlocks, unet_small.py:175-187): one
// workgroup of 8 waves per image, wave w owns queries 32w..32w+31.
//   * every operand block (64 rows x 512 B of Q, K or V) travels global -> LDS by DMA (global_load_lds, no staging
//     registers) through a ring of four 32 KB slots: Q0..3, then K0..3, then V0..3 into the slot its K block vacated —
//     K and V are read ONCE per image (the generic kernel reads them once per 128 queries) and twelve blocks are in
//     flight / queued behind the MFMAs instead of one;
//   * 256 keys fit the accumulators (8 S^T tiles), so the softmax is a plain two-pass one: no running max, no rescale;
//   * an LDS row is 32 16-byte slots; slot c of row r sits at c ^ swz(r), swz(r) = ((r & 3) << 2) | ((r >> 2) & 3): the
//     b128 fragment reads of Q / K (16 rows with distinct r & 15 per lane group) and the transposing b64 reads of V
//     (4 rows x 64 B per half wave) are both bank-conflict free on that one image;
//   * the output tile goes back through the (free) ring so that a wave stores whole 512-byte rows.
#define AT_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define AT_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AT_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void at_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int at_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

template <bool PROJ>
__global__ __launch_bounds__(512, 1) void attention256_kernel(AttnArgs p) {
    constexpr int SLOT = 64 * 512;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int C3 = 3 * p.C;
    const bf16* base = p.qkv + (size_t)n * 256 * C3;
    const int h = lane >> 5;

    // block b: rows 64*(b&3).. of Q (b < 4), K (b < 8) or V; every wave moves 4 x 1 KiB (two rows each)
    const int drow = lane >> 5, dslot = lane & 31;
    auto issue_block = [&](int b, int slot) {
        const int choff = b < 4 ? p.q_off : (b < 8 ? p.k_off : p.v_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 2 + drow;                       // row inside the block
            const int c = dslot ^ at_swz(r);
            const bf16* g = base + (size_t)((b & 3) * 64 + r) * C3 + choff + c * 8;
            __builtin_amdgcn_global_load_lds(AT_GPTR(g), AT_LPTR(smem + slot * SLOT + (wave * 4 + i) * 1024), 16, 0, 0);
        }
    };
    if (PROJ && wave == 0)      // bias[256] -> LDS behind the ring (1 KiB, oldest DMA of wave 0)
        __builtin_amdgcn_global_load_lds(AT_GPTR(p.pbias + lane * 4), AT_LPTR(smem + 4 * SLOT), 16, 0, 0);
#pragma unroll
    for (int b = 0; b < 4; ++b) issue_block(b, b);
    AT_WAIT_VM(0);
    at_barrier();
    bf16x8 qf[16];      // B operand: lane = query, d = ks*16 + 8h + j
    {
        const int r = (wave & 1) * 32 + (lane & 31);
        const char* qrow = smem + (wave >> 1) * SLOT + r * 512;
        const int sw = at_swz(r);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + (((ks * 2 + h) ^ sw) << 4));
    }
    at_barrier();       // every wave holds its Q fragments: the four slots are free
#pragma unroll
    for (int b = 0; b < 4; ++b) issue_block(4 + b, b);

    // ---- S^T[key][query] = K . Q^T, 8 tiles of 32 keys
    f32x16 s[8];
    const int krow = lane & 31, ksw = at_swz(krow);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        AT_WAIT_VM(12);         // younger than K(kb): K(kb+1..3) and V(0..kb-1), four DMAs each
        at_barrier();
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            f32x16 a16;
#pragma unroll
            for (int r = 0; r < 16; ++r) a16[r] = 0.f;
            const char* rowp = smem + kb * SLOT + (kh * 32 + krow) * 512;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rowp + (((ks * 2 + h) ^ ksw) << 4));
                a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], a16, 0, 0, 0);
            }
            s[kb * 2 + kh] = a16;
        }
        at_barrier();           // every wave is done with K(kb): its slot takes V(kb)
        issue_block(8 + kb, kb);
    }

    // ---- softmax over the 256 keys of each query (lane-local + one exchange with lane ^ 32)
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[t][r] *= p.scale;
            mx = fmaxf(mx, s[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    bf16x8 pb[16];      // P^T fragments: k-step st covers keys 16*st .. +15 in accumulator order
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __expf(s[t][r] - mx);
            l += e;
            pb[t * 2 + (r >> 3)][r & 7] = (bf16)e;
        }
    l += __shfl_xor(l, 32, 64);

    // ---- O^T[d][query] = V^T . P^T
    f32x16 o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    // tr-read lane roles (ds_read_b64_tr_b16): 16-lane group g -> d half (g & 1), key half (g >> 1); lane 4q+pp -> key row q
    const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
    const int tr_krow = 4 * (trg >> 1) + trq;
    const int tr_c = 2 * (trg & 1) + (trp >> 1), tr_sub = 8 * (trp & 1);
    // PROJ: the first half of the packed proj_out weights (cout blocks 0..3: 64 one-KiB fragments) follows V(0), V(1) into
    // their slots; wave w moves fragments 4w..4w+3 of a slot
    auto issue_w = [&](int half, int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = wave * 4 + i;
            const char* g = reinterpret_cast<const char*>(p.wproj) + (size_t)((half * 2 + slot) * 32 + f) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds(AT_GPTR(g), AT_LPTR(smem + slot * SLOT + f * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int vb = 0; vb < 4; ++vb) {
        if (PROJ) {             // younger than V(vb): V(vb+1..3) and the weight fragments issued so far
            if (vb == 3) AT_WAIT_VM(8);
            else AT_WAIT_VM(12);
        } else {
            if (vb == 0) AT_WAIT_VM(12);
            else if (vb == 1) AT_WAIT_VM(8);
            else if (vb == 2) AT_WAIT_VM(4);
            else AT_WAIT_VM(0);
        }
        at_barrier();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int r0 = 16 * st + tr_krow, r1 = r0 + 8;
            const char* row0 = smem + vb * SLOT + r0 * 512 + tr_sub;
            const char* row1 = smem + vb * SLOT + r1 * 512 + tr_sub;
            const int sw0 = at_swz(r0), sw1 = at_swz(r1);
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row0 + (((db * 4 + tr_c) ^ sw0) << 4)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row1 + (((db * 4 + tr_c) ^ sw1) << 4)));
                bf16x8 a;
                short* as = reinterpret_cast<short*>(&a);
#pragma unroll
                for (int e = 0; e < 4; ++e) { as[e] = lo[e]; as[4 + e] = hi[e]; }
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[vb * 4 + st], o[db], 0, 0, 0);
            }
        }
        if (PROJ && vb < 2) {
            at_barrier();       // every wave is done with V(vb): its slot takes weight fragments
            issue_w(0, vb);
        }
    }
    at_barrier();       // every wave is done with the V images: the ring becomes the output tile
    if constexpr (PROJ) {
        // ---- Y^T[cout][query] = Wp . O^T (+ bias + x): O^T stays in registers as the B operand — k-step j of the product takes
        // the lane's accumulator registers 8 (j & 1) .. + 7 of o[j >> 1] as they stand (channels 16 j + 8 (i >> 2) + 4 h + (i & 3)),
        // and the packed weights carry the same channel order (dxmi_pack_attn_proj_weight).  Two halves of 128 couts: weights in
        // slots 0-1, the residual / output tile of the half (wave-private 32 rows x 256 B, swizzled like the Q / K / V rows) in
        // slots 2-3.  Same values as the unfused path: O rounded to bf16, fp32 accumulation, one rounding of y + bias + x.
        const float inv = 1.f / l;
        char* const rbase = smem + 2 * SLOT + wave * 8192;
        const bf16* const xrows = p.res + ((size_t)n * 256 + wave * 32) * 256;
        bf16* const orows = p.out + ((size_t)n * 256 + wave * 32) * 256;
        const int rrow = lane >> 4, rslot = lane & 15;
        auto issue_res = [&](int half) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 4 * i + rrow;
                const bf16* g = xrows + (size_t)r * 256 + half * 128 + ((rslot ^ at_swz(r)) << 3);
                __builtin_amdgcn_global_load_lds(AT_GPTR(g), AT_LPTR(rbase + i * 1024), 16, 0, 0);
            }
        };
        issue_res(0);
        bf16x8 of[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) of[j][i] = (bf16)(o[j >> 1][8 * (j & 1) + i] * inv);
        const int r = lane & 31, sw = at_swz(r);
        const float* const bias_s = reinterpret_cast<const float*>(smem + 4 * SLOT);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // in-order retirement: weights of this half are older than the 8 residual DMAs (+ 8 row stores of half 0)
            if (half == 0) AT_WAIT_VM(8);
            else if (p.gn_stats) AT_WAIT_VM(18);        // + the two statistics stores of half 0
            else AT_WAIT_VM(16);
            at_barrier();
            f32x16 y[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) y[cb][q] = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + (cb * 16 + j) * 1024 + lane * 16);
                    y[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, of[j], y[cb], 0, 0, 0);
                }
            }
            if (half == 0) {
                at_barrier();       // every wave is done with the first half of the weights
                issue_w(1, 0);
                issue_w(1, 1);
                AT_WAIT_VM(8);      // the residual rows of half 0 are older than those 8
            } else {
                AT_WAIT_VM(0);
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    char* const a = rbase + r * 256 + (((cb * 4 + g) ^ sw) << 4) + 8 * h;
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a);
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + half * 128 + cb * 32 + g * 8 + 4 * h);
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (bf16)(y[cb][4 * g + e] + (bv[e] + (float)rv[e]));
                    *reinterpret_cast<bf16x4*>(a) = ov;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                bf16x8 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const bf16x8*>(rbase + i * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = 4 * i + rrow;
                    *reinterpret_cast<bf16x8*>(orows + (size_t)rr * 256 + half * 128 + ((rslot ^ at_swz(rr)) << 3)) = v[i];
                }
            }
            if (p.gn_stats) {
                // block statistics of the stored values: lane (16-byte piece sp, row group rg) adds its piece over 8 rows, the
                // four row groups are added in a fixed order; one partial per wave (32 tokens)
                const int sp = lane & 15, rg = lane >> 4;
                float st[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[k][e] = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int rr = rg * 8 + t;
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(rbase + rr * 256 + ((sp ^ at_swz(rr)) << 4));
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
                for (int e = 0; e < 8; ++e) {
                    tot[e] += __shfl_xor(tot[e], 16, 64);
                    tot[e] += __shfl_xor(tot[e], 32, 64);
                }
                if (lane < 16) {
                    float* sd = p.gn_stats + (((size_t)n * 8 + wave) * 128 + half * 64 + sp * 4) * 2;
                    *reinterpret_cast<f32x4*>(sd) = f32x4{tot[0], tot[1], tot[2], tot[3]};
                    *reinterpret_cast<f32x4*>(sd + 4) = f32x4{tot[4], tot[5], tot[6], tot[7]};
                }
            }
            if (half == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows are in registers: the region takes half 1's x
                issue_res(1);
            }
        }
        return;
    }

    // ---- O / l -> bf16 -> LDS rows (query-major, same swizzle) -> whole-row stores
    const float inv = 1.f / l;
    {
        const int r = lane & 31;
        char* orow = smem + wave * (32 * 512) + r * 512 + 8 * h;
        const int sw = at_swz(r);
#pragma unroll
        for (int db = 0; db < 8; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16)(o[db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + (((db * 4 + g) ^ sw) << 4)) = ov;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows a wave drains are the rows it wrote
    bf16* obase = p.out + ((size_t)n * 256 + wave * 32) * p.C;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
        bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(smem + wave * (32 * 512) + (i + u) * 1024 + lane * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (i + u) * 2 + drow;
            *reinterpret_cast<bf16x8*>(obase + (size_t)r * p.C + ((dslot ^ at_swz(r)) << 3)) = v[u];
        }
    }
}

int launch_attn256(const AttnArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (a.wproj) hipLaunchKernelGGL(attention256_kernel<true>, dim3(a.N), dim3(512), (size_t)4 * 64 * 512 + 1024, st, a);
    else hipLaunchKernelGGL(attention256_kernel<false>, dim3(a.N), dim3(512), (size_t)4 * 64 * 512, st, a);
    DXMI_CHECK_LAUNCH("dxmi_attention_fwd(256x256)");
    return DXMI_OK;
}

// proj_out weight [256 cout][256 cin] fp32 -> bf16 MFMA A fragments in the channel order attention256_kernel<true> holds O^T in:
// fragment (cb, j), lane l, element i = W[32 cb + (l & 31)][16 j + 8 (i >> 2) + 4 (l >> 5) + (i & 3)]
__global__ __launch_bounds__(256) void pack_attn_proj_kernel(const float* __restrict__ w, bf16* __restrict__ dst) {
    const int t = blockIdx.x * 256 + threadIdx.x;       // one (fragment, lane) per thread: 8 * 16 * 64 threads
    const int l = t & 63, j = (t >> 6) & 15, cb = t >> 10;
    const float* row = w + (size_t)(32 * cb + (l & 31)) * 256 + 16 * j + 4 * (l >> 5);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16)row[8 * (i >> 2) + (i & 3)];
    *reinterpret_cast<bf16x8*>(dst + (size_t)t * 8) = o;
}

template <int D, bool PF>
int launch_attn_pf(const AttnArgs& a, hipStream_t st) {
    auto kern = attention_kernel<D, PF>;
    const size_t lds = 64 * (D * 2 + 16) + 64 * (D * 2 + 64);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int qblocks = (a.T + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(a.N * a.heads * qblocks), dim3(256), lds, st, a);
    DXMI_CHECK_LAUNCH("dxmi_attention_fwd");
    return DXMI_OK;
}

// PF: key block kb + 1 is fetched into registers while block kb is computed.  Wide heads (one wave per SIMD) always; narrow heads
// (D = 64: four workgroups per CU hide part of the latency themselves) by measurement — DXMI_ATTN_PF overrides (0 / 1).
template <int D>
int launch_attn(const AttnArgs& a, hipStream_t st) {
    static const int pf_env = getenv("DXMI_ATTN_PF") ? atoi(getenv("DXMI_ATTN_PF")) : -1;
    const bool pf = pf_env >= 0 ? pf_env != 0 : D >= 128;
    return pf ? launch_attn_pf<D, true>(a, st) : launch_attn_pf<D, false>(a, st);
}

}  // namespace

extern "C" int dxmi_pack_attn_proj_weight(const float* w, void* dst, void* stream) {
    DXMI_CHECK_ARG(w && dst, "dxmi_pack_attn_proj_weight: null pointer");
    hipLaunchKernelGGL(pack_attn_proj_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w, (bf16*)dst);
    DXMI_CHECK_LAUNCH("dxmi_pack_attn_proj_weight");
    return DXMI_OK;
}

extern "C" int dxmi_attention_proj_supported(int32_t T, int32_t C, int32_t heads) { return T == 256 && C == 256 && heads == 1 ? 1 : 0; }

extern "C" int dxmi_attention_proj_fwd(const void* qkv, const void* wproj_packed, const float* bias, const void* residual, void* out,
                                       float* gn_stats, int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream) {
    DXMI_CHECK_ARG(qkv && wproj_packed && bias && residual && out, "dxmi_attention_proj_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && dxmi_attention_proj_supported(T, C, heads),
                   "dxmi_attention_proj_fwd: only the single-head 256-token x 256-channel block (N=%d T=%d C=%d heads=%d)", N, T, C, heads);
    AttnArgs a;
    a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.N = N; a.T = T; a.C = C; a.heads = heads;
    a.q_off = 0; a.k_off = C; a.v_off = 2 * C; a.head_stride = C; a.scale = scale;
    a.wproj = (const bf16*)wproj_packed; a.pbias = bias; a.res = (const bf16*)residual; a.gn_stats = gn_stats;
    a.xcd = 0; a.lse = nullptr;
    return launch_attn256(a, (hipStream_t)stream);
}

static int attention_fwd_impl(const void* qkv, void* out, float* lse, int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream);

extern "C" int dxmi_attention_fwd(const void* qkv, void* out, int32_t N, int32_t T, int32_t C, int32_t heads,
                                  float scale, void* stream) {
    return attention_fwd_impl(qkv, out, nullptr, N, T, C, heads, scale, stream);
}

// 1 when dxmi_attention_fwd_lse serves the shape (the kernels that keep a running (max, sum) per query: every shape but the
// single-head 256 x 256 one of attention256_kernel).
extern "C" int dxmi_attention_fwd_lse_supported(int32_t T, int32_t C, int32_t heads) {
    if (heads <= 0 || C % heads != 0 || T <= 0) return 0;
    const int D = C / heads;
    if (D == 256 && T == 256 && heads == 1) return 0;
    return D == 64 || D == 128 || D == 256;
}

// The forward that also leaves the row log-sum-exp (log2 domain, fp32 [N][heads][T]) for dxmi_attention_bwd_lse: the saved softmax
// statistics of autograd's attention node.
extern "C" int dxmi_attention_fwd_lse(const void* qkv, void* out, float* lse, int32_t N, int32_t T, int32_t C, int32_t heads,
                                      float scale, void* stream) {
    DXMI_CHECK_ARG(lse, "dxmi_attention_fwd_lse: null lse pointer");
    DXMI_CHECK_ARG(dxmi_attention_fwd_lse_supported(T, C, heads), "dxmi_attention_fwd_lse: shape T=%d C=%d heads=%d not served", T, C, heads);
    return attention_fwd_impl(qkv, out, lse, N, T, C, heads, scale, stream);
}

static int attention_fwd_impl(const void* qkv, void* out, float* lse, int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream) {
    DXMI_CHECK_ARG(qkv && out, "dxmi_attention_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && T > 0 && heads > 0 && C % heads == 0, "dxmi_attention_fwd: bad shape N=%d T=%d C=%d heads=%d", N, T, C, heads);
    const int D = C / heads;
    AttnArgs a;
    a.qkv = (const bf16*)qkv; a.out = (bf16*)out; a.N = N; a.T = T; a.C = C; a.heads = heads;
    a.q_off = 0; a.k_off = C; a.v_off = 2 * C; a.head_stride = D; a.scale = scale;
    a.wproj = nullptr; a.pbias = nullptr; a.res = nullptr; a.gn_stats = nullptr; a.lse = lse;
    static const int xcd_env = getenv("DXMI_ATTN_XCD") ? atoi(getenv("DXMI_ATTN_XCD")) : 1;      // 0: hardware block order (A/B timing)
    a.xcd = xcd_env;
    hipStream_t st = (hipStream_t)stream;
    static const int v1 = getenv("DXMI_ATTN_GENERIC") ? atoi(getenv("DXMI_ATTN_GENERIC")) : 0;   // 1: generic kernel for every shape
    if (D == 256 && T == 256 && heads == 1 && !v1) return launch_attn256(a, st);
    if (D == 256) return launch_attn<256>(a, st);
    if (D == 128) return launch_attn<128>(a, st);
    if (D == 64) {
        // long sequences of 64-wide heads: two query tiles per wave, one barrier per key block (attention64_kernel); DXMI_ATTN64=0: generic kernel
        static const int v64 = getenv("DXMI_ATTN64") ? atoi(getenv("DXMI_ATTN64")) : 1;
        if (v64 && T % 256 == 0 && scale > 0.f) {
            static bool attr_set = false;
            if (!attr_set) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(attention64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            hipLaunchKernelGGL(attention64_kernel, dim3(N * heads * (T / 256)), dim3(256), AT64_LDS, st, a);
            DXMI_CHECK_LAUNCH("dxmi_attention_fwd(64)");
            return DXMI_OK;
        }
        return launch_attn<64>(a, st);
    }
    dxmi_set_error("dxmi_attention_fwd: head dim %d unsupported (64, 128, 256)", D);
    return DXMI_EINVAL;
}
