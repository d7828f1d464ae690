// Fused attention backward for 64-wide heads on gfx950 (round 4): no [T, T] tensor ever reaches HBM.
//
// The round-2 path (attention_bwd.hip) is five batched GEMMs and a row softmax-backward that write and re-read S and dP in
// fp32 and P and dS in bf16: 26 MB per (image, head) at T = 1024 — 2.5 GB per 1024-token block of the ImageNet-64 net at
// batch 16, 11 % of the EDM train step.  Here the probabilities are RECOMPUTED from the saved inputs, flash-attention style,
// by two kernels that reuse the forward kernel's fragment scheme (attention.hip: S^T tiles with the "lane" index on the
// accumulator column, P^T taken straight from the accumulators as the B operand of the next MFMA, the row-major LDS image of
// the other operand read through transposing loads in the accumulators' k order):
//
//   attn_bwd_dq_kernel   one workgroup = 128 queries of one (image, head), wave = 32 queries, lane = query.
//       sweep 1 over the key blocks:  S^T = K Q^T, online (max, sum)  ->  L[q] = max + log2(sum)  (log-sum-exp of the row, log2 domain)
//                                     delta[q] = <dO[q], O[q]>  (= sum_k P[q][k] dP[q][k])
//       sweep 2:  S^T again, P^T = exp(scale S^T - L), dP^T = V dO^T, dS^T = P^T o (dP^T - delta),
//                 dQ^T[d][q] += K^T dS^T        (A = K^T by transposing LDS loads, B = dS^T from the accumulators)
//       L and delta go to a small workspace for the second kernel.
//   attn_bwd_dkv_kernel  one workgroup = 128 keys of one (image, head), wave = 32 keys, lane = key; loop over query blocks:
//       S = Q K^T (row = query, column = key), P = exp(scale S - L[row]), dP = dO V^T, dS = P o (dP - delta[row]),
//       dV^T[d][k] += dO^T P,   dK^T[d][k] += Q^T dS      (A by transposing LDS loads of the row-major dO / Q images)
//
// Eight T^2 D contractions instead of five, all on MFMA (v_mfma_f32_32x32x16_bf16), against 26 MB of HBM traffic per
// (image, head) saved; every sum runs in a fixed order (no atomics): bitwise reproducible.
// Replaces autograd through QKVAttentionLegacy.forward (models/cm/unet.py:413-441) / AttnBlock.forward
// (models/DxMI/unet_small.py:175-187) for head dimension 64; other head sizes keep attention_bwd.hip.
#include "common.h"
#include <math.h>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

__device__ __forceinline__ int ab_xcd_logical(int b, int G, int on) {
    if (!on) return b;
    const int x = b & 7, k = b >> 3;
    const int q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + k;
}

struct AttnBwdArgs {
    const bf16* qkv;     // [N, T, 3C]  q | k | v, heads = contiguous channel blocks of D
    const bf16* o;       // [N, T, C]   attention output of the forward pass
    const bf16* dout;    // [N, T, C]
    bf16* dqkv;          // [N, T, 3C]
    float* lse;          // [N, heads, T]
    float* delta;        // [N, heads, T]
    const float* lse_in; // optional: the forward's row log-sum-exp (dxmi_attention_fwd_lse) — sweep 1 of attn_bwd_dq_kernel is then skipped
    int N, T, C, heads;
    float scale;
    int xcd;             // XCD-aware block order: the query / key blocks of one (image, head) behind one L2 (see attention.hip)
};

constexpr int AB_D = 64;
constexpr int AB_DK = AB_D / 16;          // k-steps over d of a 32x32x16 MFMA
constexpr int AB_DB = AB_D / 32;          // 32-row blocks of a [d][lane] accumulator
constexpr int AB_KB = 64;                 // rows of the streamed operand per block
constexpr int AB_RP = AB_D * 2 + 16;      // pitch of a row-major image read with ds_read_b128 (row = MFMA row)
constexpr int AB_TP = AB_D * 2 + 64;      // pitch of a row-major image read with transposing loads

// transposing read of A = X^T tile [32 d][16 rows] from the row-major image `img` (pitch AB_TP): rows r0 .. r0 + 15 in the
// accumulators' k order (element j of lane-half h is row 8 (j >> 2) + 4 h + (j & 3)), d block db
__device__ __forceinline__ bf16x8 tr_tile(const char* img, int r0, int db, int tr_krow, int tr_doff) {
    const char* row = img + (r0 + tr_krow) * AB_TP + db * 64 + tr_doff;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(row + 8 * AB_TP));
    bf16x8 a;
    short* as = reinterpret_cast<short*>(&a);
#pragma unroll
    for (int e = 0; e < 4; ++e) { as[e] = lo[e]; as[4 + e] = hi[e]; }
    return a;
}

// stage a 64-row x 64-channel block of `src` (row stride ld elements, rows past `rows_valid` zero) into a b128 image and / or a
// transposing image
__device__ __forceinline__ void stage_block(const bf16* src, long ld, int row0, int rows_valid, char* rimg, char* timg, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * 256;
        const int r = i >> 3, pc = i & 7;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        if (row0 + r < rows_valid) v = *reinterpret_cast<const bf16x8*>(src + (long)(row0 + r) * ld + pc * 8);
        if (rimg) *reinterpret_cast<bf16x8*>(rimg + r * AB_RP + pc * 16) = v;
        if (timg) *reinterpret_cast<bf16x8*>(timg + r * AB_TP + pc * 16) = v;
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    __shared__ __attribute__((aligned(16))) char kimg[AB_KB * AB_RP];
    __shared__ __attribute__((aligned(16))) char ktimg[AB_KB * AB_TP];
    __shared__ __attribute__((aligned(16))) char vimg[AB_KB * AB_RP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
    const int tr_doff = (16 * (trg & 1) + 4 * trp) * 2;
    const int tr_krow = 4 * (trg >> 1) + trq;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qblocks = (p.T + 127) / 128;
    const int bid = ab_xcd_logical(blockIdx.x, gridDim.x, p.xcd);
    const int qb = bid % qblocks;
    const int hd = (bid / qblocks) % p.heads;
    const int n = bid / (qblocks * p.heads);
    const int C3 = 3 * p.C;
    const bf16* base = p.qkv + (size_t)n * p.T * C3;
    const int qc = hd * AB_D, kc = p.C + hd * AB_D, vc = 2 * p.C + hd * AB_D;
    const int h = lane >> 5;
    const int query = qb * 128 + wave * 32 + (lane & 31);
    const bool qvalid = query < p.T;
    const float sc2 = p.scale * 1.4426950408889634f;      // logits in the log2 domain

    // B-operand fragments of this lane's query: elements d = ks*16 + 8h + j
    bf16x8 qf[AB_DK], dof[AB_DK];
    float delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < AB_DK; ++ks) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
        qf[ks] = z; dof[ks] = z;
        if (qvalid) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)query * C3 + qc + ks * 16 + 8 * h);
            const size_t orow = ((size_t)n * p.T + query) * p.C + hd * AB_D + ks * 16 + 8 * h;
            dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + orow);
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + orow);
#pragma unroll
            for (int e = 0; e < 8; ++e) delta += (float)dof[ks][e] * (float)ov[e];
        }
    }
    delta += __shfl_xor(delta, 32, 64);

    const int nkb = (p.T + AB_KB - 1) / AB_KB;
    // ---- sweep 1: row log-sum-exp (skipped when the forward saved it: launch-uniform branch)
    float m = -INFINITY, l = 0.f;
    for (int kb = 0; kb < (p.lse_in ? 0 : nkb); ++kb) {
        __syncthreads();
        stage_block(base + kc, C3, kb * AB_KB, p.T, kimg, nullptr, tid);
        __syncthreads();
        float mx = -INFINITY;
        f32x16 s[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kh][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < AB_DK; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kimg + (kh * 32 + (lane & 31)) * AB_RP + ks * 32 + h * 16);
                s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[kh], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * AB_KB + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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
            for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(s[kh][r] - m_new);
        l = l * __builtin_amdgcn_exp2f(m - m_new) + psum;
        m = m_new;
    }
    l += __shfl_xor(l, 32, 64);
    const size_t si = ((size_t)n * p.heads + hd) * p.T + (qvalid ? query : 0);
    const float lse = p.lse_in ? p.lse_in[si] : m + __log2f(l);          // log2 domain throughout (both kernels): P = exp2(sc2 S - L)
    if (qvalid && h == 0) {
        p.lse[si] = lse;
        p.delta[si] = delta;
    }

    // ---- sweep 2: dQ^T[d][query]
    f32x16 dq[AB_DB];
#pragma unroll
    for (int db = 0; db < AB_DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        stage_block(base + kc, C3, kb * AB_KB, p.T, kimg, ktimg, tid);
        stage_block(base + vc, C3, kb * AB_KB, p.T, vimg, nullptr, tid);
        __syncthreads();
        f32x16 s[2], dp[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[kh][r] = 0.f; dp[kh][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < AB_DK; ++ks) {
                const bf16x8 ak = *reinterpret_cast<const bf16x8*>(kimg + (kh * 32 + (lane & 31)) * AB_RP + ks * 32 + h * 16);
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(vimg + (kh * 32 + (lane & 31)) * AB_RP + ks * 32 + h * 16);
                s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[ks], s[kh], 0, 0, 0);
                dp[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dof[ks], dp[kh], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * AB_KB + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float pv = (key < p.T) ? __builtin_amdgcn_exp2f(s[kh][r] * sc2 - lse) : 0.f;
                s[kh][r] = pv * (dp[kh][r] - delta);          // dS^T
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int kh = st >> 1, sl = st & 1;
            bf16x8 pb;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pb[jj] = (bf16)s[kh][8 * sl + jj];
#pragma unroll
            for (int db = 0; db < AB_DB; ++db)
                dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_tile(ktimg, kh * 32 + 16 * sl, db, tr_krow, tr_doff), pb, dq[db], 0, 0, 0);
        }
    }
    if (qvalid) {
        bf16* orow = p.dqkv + ((size_t)n * p.T + query) * C3 + qc;
#pragma unroll
        for (int db = 0; db < AB_DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16)(dq[db][4 * g + e] * p.scale);
                *reinterpret_cast<bf16x4*>(orow + db * 32 + 8 * g + 4 * h) = ov;
            }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
    __shared__ __attribute__((aligned(16))) char qimg[AB_KB * AB_RP];
    __shared__ __attribute__((aligned(16))) char qtimg[AB_KB * AB_TP];
    __shared__ __attribute__((aligned(16))) char doimg[AB_KB * AB_RP];
    __shared__ __attribute__((aligned(16))) char dotimg[AB_KB * AB_TP];
    __shared__ __attribute__((aligned(16))) float lse_s[AB_KB];
    __shared__ __attribute__((aligned(16))) float del_s[AB_KB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int trg = lane >> 4, trq = (lane & 15) >> 2, trp = lane & 3;
    const int tr_doff = (16 * (trg & 1) + 4 * trp) * 2;
    const int tr_krow = 4 * (trg >> 1) + trq;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kblocks = (p.T + 127) / 128;
    const int bid = ab_xcd_logical(blockIdx.x, gridDim.x, p.xcd);
    const int kbk = bid % kblocks;
    const int hd = (bid / kblocks) % p.heads;
    const int n = bid / (kblocks * p.heads);
    const int C3 = 3 * p.C;
    const bf16* base = p.qkv + (size_t)n * p.T * C3;
    const bf16* dob = p.dout + (size_t)n * p.T * p.C + hd * AB_D;
    const int qc = hd * AB_D, kc = p.C + hd * AB_D, vc = 2 * p.C + hd * AB_D;
    const int h = lane >> 5;
    const int key = kbk * 128 + wave * 32 + (lane & 31);
    const bool kvalid = key < p.T;
    const float sc2 = p.scale * 1.4426950408889634f;
    const float* lse_g = p.lse + ((size_t)n * p.heads + hd) * p.T;
    const float* del_g = p.delta + ((size_t)n * p.heads + hd) * p.T;

    bf16x8 kf[AB_DK], vf[AB_DK];
#pragma unroll
    for (int ks = 0; ks < AB_DK; ++ks) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
        kf[ks] = z; vf[ks] = z;
        if (kvalid) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)key * C3 + kc + ks * 16 + 8 * h);
            vf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)key * C3 + vc + ks * 16 + 8 * h);
        }
    }
    f32x16 dk[AB_DB], dv[AB_DB];
#pragma unroll
    for (int db = 0; db < AB_DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

    const int nqb = (p.T + AB_KB - 1) / AB_KB;
    for (int qb = 0; qb < nqb; ++qb) {
        __syncthreads();
        stage_block(base + qc, C3, qb * AB_KB, p.T, qimg, qtimg, tid);
        stage_block(dob, p.C, qb * AB_KB, p.T, doimg, dotimg, tid);
        if (tid < AB_KB) {
            const int qq = qb * AB_KB + tid;
            lse_s[tid] = qq < p.T ? lse_g[qq] : INFINITY;       // exp(s - inf) = 0: rows past T contribute nothing
            del_s[tid] = qq < p.T ? del_g[qq] : 0.f;
        }
        __syncthreads();
        f32x16 s[2], dp[2];
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[qh][r] = 0.f; dp[qh][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < AB_DK; ++ks) {
                const bf16x8 aq = *reinterpret_cast<const bf16x8*>(qimg + (qh * 32 + (lane & 31)) * AB_RP + ks * 32 + h * 16);
                const bf16x8 ad = *reinterpret_cast<const bf16x8*>(doimg + (qh * 32 + (lane & 31)) * AB_RP + ks * 32 + h * 16);
                s[qh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[ks], s[qh], 0, 0, 0);       // S[query][key]
                dp[qh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp[qh], 0, 0, 0);     // dP[query][key]
            }
            // rows of accumulator group g = r >> 2: queries qh*32 + 8 g + 4 h + (0..3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 lv = *reinterpret_cast<const f32x4*>(lse_s + qh * 32 + 8 * g + 4 * h);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(del_s + qh * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pv = __builtin_amdgcn_exp2f(s[qh][r] * sc2 - lv[e]);
                    s[qh][r] = pv;                                  // P
                    dp[qh][r] = pv * (dp[qh][r] - dl[e]);           // dS
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int qh = st >> 1, sl = st & 1;
            bf16x8 pb, sb;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { pb[jj] = (bf16)s[qh][8 * sl + jj]; sb[jj] = (bf16)dp[qh][8 * sl + jj]; }
#pragma unroll
            for (int db = 0; db < AB_DB; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_tile(dotimg, qh * 32 + 16 * sl, db, tr_krow, tr_doff), pb, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_tile(qtimg, qh * 32 + 16 * sl, db, tr_krow, tr_doff), sb, dk[db], 0, 0, 0);
            }
        }
    }
    if (kvalid) {
        bf16* krow = p.dqkv + ((size_t)n * p.T + key) * C3 + kc;
        bf16* vrow = p.dqkv + ((size_t)n * p.T + key) * C3 + vc;
#pragma unroll
        for (int db = 0; db < AB_DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 ok, ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ok[e] = (bf16)(dk[db][4 * g + e] * p.scale);
                    ov[e] = (bf16)dv[db][4 * g + e];
                }
                *reinterpret_cast<bf16x4*>(krow + db * 32 + 8 * g + 4 * h) = ok;
                *reinterpret_cast<bf16x4*>(vrow + db * 32 + 8 * g + 4 * h) = ov;
            }
    }
}

}  // namespace

extern "C" int dxmi_attention_bwd_supported(int32_t T, int32_t C, int32_t heads) {
    return heads > 0 && C % heads == 0 && C / heads == AB_D && T >= 1 ? 1 : 0;
}

extern "C" int64_t dxmi_attention_bwd_workspace_bytes(int32_t N, int32_t T, int32_t heads) {
    if (N <= 0 || T <= 0 || heads <= 0) return 0;
    return (int64_t)2 * N * heads * T * 4;
}

static int attention_bwd_impl(const void* qkv, const void* o, const void* dout, void* dqkv, const float* lse_fwd, void* workspace, int32_t N,
                              int32_t T, int32_t C, int32_t heads, float scale, void* stream);

extern "C" int dxmi_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, void* workspace, int32_t N, int32_t T,
                                  int32_t C, int32_t heads, float scale, void* stream) {
    return attention_bwd_impl(qkv, o, dout, dqkv, nullptr, workspace, N, T, C, heads, scale, stream);
}

// lse_fwd: what dxmi_attention_fwd_lse left for these qkv and this scale ([N][heads][T], log2 domain): the backward then runs three
// sweeps over the keys instead of four (autograd's saved softmax statistics).
extern "C" int dxmi_attention_bwd_lse(const void* qkv, const void* o, const void* dout, void* dqkv, const float* lse_fwd, void* workspace,
                                      int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream) {
    DXMI_CHECK_ARG(lse_fwd, "dxmi_attention_bwd_lse: null lse pointer");
    return attention_bwd_impl(qkv, o, dout, dqkv, lse_fwd, workspace, N, T, C, heads, scale, stream);
}

static int attention_bwd_impl(const void* qkv, const void* o, const void* dout, void* dqkv, const float* lse_fwd, void* workspace, int32_t N,
                              int32_t T, int32_t C, int32_t heads, float scale, void* stream) {
    DXMI_CHECK_ARG(qkv && o && dout && dqkv && workspace, "dxmi_attention_bwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && T > 0 && heads > 0 && C > 0 && C % heads == 0, "dxmi_attention_bwd: bad shape N=%d T=%d C=%d heads=%d", N, T, C, heads);
    DXMI_CHECK_ARG(dxmi_attention_bwd_supported(T, C, heads), "dxmi_attention_bwd: head dimension %d unsupported (64)", C / heads);
    AttnBwdArgs a;
    a.qkv = (const bf16*)qkv; a.o = (const bf16*)o; a.dout = (const bf16*)dout; a.dqkv = (bf16*)dqkv;
    a.lse = (float*)workspace; a.delta = a.lse + (size_t)N * heads * T; a.lse_in = lse_fwd;
    a.N = N; a.T = T; a.C = C; a.heads = heads; a.scale = scale;
    static const int xcd_env = getenv("DXMI_ATTN_XCD") ? atoi(getenv("DXMI_ATTN_XCD")) : 1;      // 0: hardware block order (A/B timing)
    a.xcd = xcd_env;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (T + 127) / 128;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(N * heads * blocks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_attention_bwd(dq)");
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(N * heads * blocks), dim3(256), 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_attention_bwd(dkv)");
    return DXMI_OK;
}
