// Shared pieces of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pipe.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    const bf16* in0;
    const bf16* in1;
    const bf16* w;
    const float* bias;
    const float* addvec;
    const bf16* residual;
    const bf16* mask_src;    // activation-gradient mask source (same shape as out) or null
    void* out;
    int N, IH, IW, C0, C1, OH, OW, Cout;
    int ksize, stride, pad, ups, act, addvec_ld, in_mode, out_mode;
    int P, pre_act;          // ROWS mode: number of rows; activation applied to the input while staging
    int TWl, THl, SUBS;      // tile geometry (log2 width, log2 height, images per tile)
    int HH, HWd;             // halo height / width
    int PT, CT, CB, KST;     // pixel tiles, cout tiles, 32-co blocks (padded), total 16-ci k-steps
    int tile_px;             // output pixels per workgroup tile (256, 128 or 64)
    int lds_buf;             // bytes of one LDS halo image (16-byte multiple)
    float mask_slope;        // factor applied where mask_src <= 0 (0.2 leaky, 0 relu)
    int stagger;             // conv_pipe: start delay (units of s_sleep 127 ~ 8k cycles) of the second resident half
    int RP, SP;              // conv_pipe: LDS pitch of a halo row / of a sub-image, bytes (bank-conflict-free choice)
    float* gn_stats;         // optional: GroupNorm block statistics of the OUTPUT, fp32 [N][P][Cout/2][2] (dxmi_conv_desc.gn_stats)
    bf16* gn_out;            // optional: fused GroupNorm(+SiLU) of the output (dxmi_conv_desc.gn_out ...): conv_sm_kernel only
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int gn_flags;            // bit 0: SiLU, bit 1: skip the raw output
    int res_is_mask;         // conv_ws_kernel: `residual` carries the activation-mask source (out *= mask > 0 ? 1 : mask_slope), set by its launcher
    int xcd_order;           // conv_ws_kernel / conv_ws8_kernel (set by their launchers): tiles handed out in XCD-aware order
};

// Persistent kernels: workgroup b runs on XCD b % 8 (round-robin dispatch; each XCD has its own L2).  Logical workgroup ids that are
// consecutive WITHIN an XCD make the tiles a launch works on at the same time — the cout tiles of one pixel tile, the neighbouring
// strips of one image (shared halo rows) — neighbours behind ONE L2 (any grid size: XCD j owns G / 8 + (j < G % 8) workgroups).
__device__ __forceinline__ int dxmi_xcd_logical(int b, int G, int on) {
    if (!on) return b;
    const int x = b & 7, k = b >> 3;
    const int q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + k;
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15): every lane of the row ends with the row's total.  Each step adds
// two partial sums over disjoint, equally shaped lane sets, so all lanes hold bitwise the same value.
__device__ __forceinline__ float dxmi_row16_sum(float t) {
#define DXMI_DPP_ADD(ctrl) t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), ctrl, 0xf, 0xf, false))
    DXMI_DPP_ADD(0xB1);    // quad_perm [1,0,3,2]
    DXMI_DPP_ADD(0x4E);    // quad_perm [2,3,0,1]
    DXMI_DPP_ADD(0x141);   // row_half_mirror
    DXMI_DPP_ADD(0x140);   // row_mirror
#undef DXMI_DPP_ADD
    return t;
}

// (sum, sum of squares) of the two channel PAIRS of four bf16 values — the values as STORED, so a GroupNorm fed by these
// statistics normalises exactly the tensor the consumer reads — added to st = (s_lo, q_lo, s_hi, q_hi): four v_dot2c_f32_bf16.
// Pairs are the granularity of the statistics tensors: every group width of the U-Nets (4 ... 32 channels per group, 6 and
// 18 in the ImageNet-64 net) and every concat boundary is a whole number of pairs.
__device__ __forceinline__ void dxmi_stats4(const bf16x4& o, float (&st)[4]) {
    const bf16x2 lo = {o[0], o[1]}, hi = {o[2], o[3]};
    const bf16x2 one = {(bf16)1.f, (bf16)1.f};
    st[0] = __builtin_amdgcn_fdot2_f32_bf16(lo, one, st[0], false);
    st[1] = __builtin_amdgcn_fdot2_f32_bf16(lo, lo, st[1], false);
    st[2] = __builtin_amdgcn_fdot2_f32_bf16(hi, one, st[2], false);
    st[3] = __builtin_amdgcn_fdot2_f32_bf16(hi, hi, st[3], false);
}

// XCD-aware block mapping: blocks b and b+8 share an XCD (round-robin dispatch), so the CT
// cout-tiles of one pixel tile are made consecutive *within* an XCD and re-read the input halo
// from that XCD's L2.  Returns false for padding blocks.
__device__ __forceinline__ bool conv_block_to_tile(const ConvArgs& p, int bid, int& pt, int& cot) {
    const int xcd = bid & 7;
    const int j = bid >> 3;
    pt = (j / p.CT) * 8 + xcd;
    cot = j % p.CT;
    return pt < p.PT;
}

// Epilogue over D[co][pixel] accumulators (lane: pixel = lane&31, co = 8g + 4h + {0..3}):
// bias, per-(n,co) temb term, residual, activation, then bf16 NHWC / fp32 rows (vector path) or the
// scalar path for narrow heads and NCHW-fp32 output (compiled only when !WIDE).
template <int MB, int NB, bool WIDE>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[MB][NB], int pt, int n0, int oy0,
                                              int ox0, int pxblk0, int cb0, int lane) {
    const bool rows = p.in_mode == DXMI_IN_ROWS_F32;
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    const int h = lane >> 5;
    const bool vec_ok = WIDE || ((p.out_mode != DXMI_OUT_NCHW_F32) && ((p.Cout & 3) == 0));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pix = (pxblk0 + nb) * 32 + (lane & 31);
        size_t opix;
        int n = 0, oy = 0, ox = 0;
        bool pvalid;
        if (rows) {
            const long grow = (long)pt * p.tile_px + pix;
            pvalid = grow < p.P;
            opix = (size_t)grow;
        } else {
            const int x = pix & (TW - 1);
            const int y = (pix >> p.TWl) & (TH - 1);
            const int sub = pix >> (p.TWl + p.THl);
            n = n0 + sub; oy = oy0 + y; ox = ox0 + x;
            pvalid = n < p.N;
            opix = ((size_t)n * p.OH + oy) * p.OW + ox;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = (cb0 + mb) * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mb][nb][4 * g + e];
                if (pvalid && co < p.Cout) {
                    if (vec_ok) {
                        if (p.bias) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += bv[e];
                        }
                        if (p.addvec) {
                            const f32x4 av = *reinterpret_cast<const f32x4*>(p.addvec + (size_t)n * p.addvec_ld + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += av[e];
                        }
                        if (p.residual) {
                            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.residual + opix * p.Cout + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                        if (p.mask_src) {
                            const bf16x4 mv = *reinterpret_cast<const bf16x4*>(p.mask_src + opix * p.Cout + co);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= ((float)mv[e] > 0.f ? 1.f : p.mask_slope);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dxmi_act(v[e], p.act);
                        if (p.out_mode == DXMI_OUT_ROWS_F32) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = v[e];
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + opix * p.Cout + co) = o;
                        } else {
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.out) + opix * p.Cout + co) = o;
                        }
                    } else if constexpr (!WIDE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (co + e < p.Cout) {
                                float r = v[e] + (p.bias ? p.bias[co + e] : 0.f);
                                if (p.addvec) r += p.addvec[(size_t)n * p.addvec_ld + co + e];
                                r = dxmi_act(r, p.act);
                                if (p.out_mode == DXMI_OUT_NCHW_F32)
                                    reinterpret_cast<float*>(p.out)[(((size_t)n * p.Cout + co + e) * p.OH + oy) * p.OW + ox] = r;
                                else if (p.out_mode == DXMI_OUT_ROWS_F32)
                                    reinterpret_cast<float*>(p.out)[opix * p.Cout + co + e] = r;
                                else
                                    reinterpret_cast<bf16*>(p.out)[opix * p.Cout + co + e] = (bf16)r;
                            }
                        }
                    }
                }
            }
        }
    }
}

// conv_pipe.hip: software-pipelined kernel for stride-1 / upsampled convs with Cout % 128 == 0.
// Returns DXMI_OK after launching, or 1 when the shape is not eligible (caller falls back).
int conv_pipe_try_launch(ConvArgs& a, hipStream_t st, int* kernel_id);
