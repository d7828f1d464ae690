/*
 * dxmi_hip.h — C-ABI of libdxmi_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * the DxMI few-step sampling / training hot path.
 *
 * The reference (swyoon/Diffusion-by-MaxEntIRL) is pure Python on torch.nn and has NO FFI of
 * its own; every entry point below therefore names the reference *Python* call site whose
 * torch ops it replaces (file:line relative to the reference root).  The Python host modules
 * under diffusion-by-maxentirl_amd/models/ bind these symbols through ctypes
 * (diffusion-by-maxentirl_amd/dxmi_hip/_lib.py); INTEGRATION.md shows the stub a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = DXMI_E* ; dxmi_last_error() gives text.
 *   - pointers are raw DEVICE pointers (tensor.data_ptr()); the caller owns all memory.
 *     The library allocates nothing on the device and never synchronises.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every launch is asynchronous on it, so all calls are hipGraph-capturable.
 *   - activations between kernels: NHWC bf16.  Sampler state / network edges: NCHW fp32.
 *   - no exceptions cross the boundary; functions are re-entrant.
 *   - NOT in this header: the gradient exchange of the data-parallel train step (reference: torch DDP wrappers,
 *     train_cifar10.py:298-309).  The host stays PyTorch, so the collectives are issued through torch.distributed with the
 *     "nccl" backend (= RCCL on ROCm) by dxmi_hip/dist.py (FlatGradSync: persistent flat buffer, ~32 MB buckets all-reduced
 *     during the backward); this library only produces the gradients those buckets carry.  Generation needs no collective.
 */
#ifndef DXMI_HIP_H
#define DXMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXMI_OK            0
#define DXMI_EINVAL       -1   /* bad shape / unsupported configuration */
#define DXMI_ELAUNCH      -2   /* hipLaunch failed */
#define DXMI_ENODEV       -3   /* no gfx950 device visible */

#define DXMI_ACT_NONE      0
#define DXMI_ACT_LEAKY02   1   /* leaky_relu(x, 0.2)  — models/modules.py:84,101,145 */
#define DXMI_ACT_RELU      2   /* relu               — models/modules.py:150 */
#define DXMI_ACT_SILU      3   /* x*sigmoid(x)       — models/DxMI/unet_small.py:30 */

#define DXMI_IN_NHWC_BF16     0
#define DXMI_IN_NCHW_F32_K27  1  /* 3-channel NCHW fp32 image, 3x3/s1/p1 im2col'd to K=27(+5 zero) */
#define DXMI_IN_ROWS_F32      2  /* internal: fp32 row-major [P,K] (dxmi_linear_fwd) */
#define DXMI_OUT_NHWC_BF16    0
#define DXMI_OUT_NCHW_F32     1
#define DXMI_OUT_ROWS_F32     2  /* internal: fp32 row-major [P,M] (dxmi_linear_fwd) */

const char* dxmi_last_error(void);
int dxmi_version(void);
/* 0 when a gfx950 device is usable; DXMI_ENODEV otherwise (never falls back to CPU). */
int dxmi_device_check(void);
/* Kernel-selection knobs (process-wide; initial value from the environment variable in brackets).  They choose between kernels
 * that compute the same convolution (same operands, fp32 accumulation; the summation order inside a K extent differs between
 * kernels, i.e. results agree to rounding, not bit for bit); unknown names return DXMI_EINVAL.
 *   "conv_ws_min_tiles" [DXMI_CONV_WS_MIN_TILES, 0]: 3x3 convs with fewer (256-pixel, 128-cout) tiles run on the 64-pixel-tile
 *                        kernel instead of the wave-specialised one (small batches leave most CUs without a tile);
 *   "conv_sm_mask"      [DXMI_CONV_SM, 9]: bit 0 4x4 maps, bit 1 every 8x8 map, bit 2 8x8 maps whose 256-pixel-tile grid is
 *                        under-filled, bit 3 8x8 maps with >= 1024 channels in and out (a rule on the layer shape, not on the
 *                        batch), on the feed-tiled small-map kernel.
 *   "gn_bwd_fused"      [DXMI_GN_BWD_FUSED, 1]: dxmi_groupnorm_generic_bwd[_saved] as ONE launch that keeps a workgroup's rows
 *                        in registers across an in-launch per-image hand-off, on maps of <= 256 pixels where it measured faster
 *                        (2: on every shape it fits; 0: the reduce + apply launches of rounds 2-5 everywhere).
 * Defaults: one kernel per layer shape whatever the batch size, so an image's result does not depend on the batch it rides in.
 * The training entry points set 96 / 13 (throughput at small per-GPU batches; +5..8 % on the EDM train step).  The answers of
 * dxmi_conv2d_gn_stats_partials / dxmi_conv2d_gn_fuse_supported depend on the knobs: query again after changing one.
 * No reference counterpart (the reference leaves algorithm choice to cuDNN). */
int dxmi_set_tuning(const char* name, int32_t value);
int dxmi_get_tuning(const char* name, int32_t* value);

/* ------------------------------------------------------------------------------------------
 * Convolution as MFMA implicit GEMM (bf16 operands, fp32 accumulate), fused epilogue.
 * Replaces torch.nn.Conv2d / F.pad+conv / F.interpolate+conv / torch.cat+conv at
 *   models/DxMI/unet_small.py:50-54 (Upsample), :69-76 (Downsample, pad (0,1,0,1) + k3 s2),
 *   :117-136 (ResnetBlock conv1/conv2/nin_shortcut + temb add + residual),
 *   :167-191 (AttnBlock q,k,v,proj_out 1x1), :302, :321-322 (cat(h, skip)), :331 (conv_out);
 *   models/modules.py:71-101 (ResBlockV2 conv1/conv2/skip), :142-145 (IGEBMEncoderV2.conv1).
 * Weights are pre-packed into MFMA A-fragment order by dxmi_pack_conv_weight().
 * ---------------------------------------------------------------------------------------- */
typedef struct dxmi_conv_desc {
    const void*  in0;        /* NHWC bf16 [N,IH,IW,C0]  (or NCHW f32 [N,3,IH,IW] for K27 mode) */
    const void*  in1;        /* second concat source NHWC bf16 [N,IH,IW,C1] or NULL */
    const void*  wpacked;    /* from dxmi_pack_conv_weight */
    const float* bias;       /* [Cout] or NULL */
    const float* addvec;     /* per-(n,co) additive term (temb_proj output) [N,addvec_ld] or NULL */
    const void*  residual;   /* NHWC bf16 [N,OH,OW,Cout] added before activation, or NULL */
    void*        out;        /* NHWC bf16 [N,OH,OW,Cout] or NCHW f32 [N,Cout,OH,OW] */
    int32_t N, IH, IW, C0, C1;
    int32_t OH, OW, Cout;
    int32_t ksize;           /* 1 or 3 */
    int32_t stride;          /* 1 or 2 */
    int32_t pad;             /* top/left zero padding (bottom/right padding is implied by OH/OW) */
    int32_t upsample;        /* 1: input is nearest-upsampled x2 before the conv; 2: zero-stuffed x2 (the
                                data gradient of a stride-2 conv is a stride-1 conv over the stuffed dY) */
    int32_t act;             /* DXMI_ACT_* applied last */
    int32_t addvec_ld;       /* row stride of addvec in floats: >= Cout, or 0 = one row shared by the whole batch */
    int32_t in_mode;         /* DXMI_IN_*  */
    int32_t out_mode;        /* DXMI_OUT_* */
    int32_t variant;         /* 0 = default tiling; >0 selects an alternative (tuning / tests) */
    const void*  mask_src;   /* backward use: NHWC bf16 tensor shaped like out; result *= (mask_src > 0 ? 1 : mask_slope)
                                after bias/residual, i.e. the LeakyReLU/ReLU derivative of a saved activation; or NULL */
    float        mask_slope;
    float*       gn_stats;   /* optional output: GroupNorm block statistics of `out` for the Normalize() that consumes it
                                (unet_small.py:35-36,119-126): fp32 [N][P][Cout/2][2] = (sum, sum of squares) of the STORED bf16
                                values over the pixels of partial p of the image, per PAIR of consecutive channels, summed in a
                                fixed order inside the conv's epilogue (bitwise reproducible).  P = dxmi_conv2d_gn_stats_partials(d);
                                NULL = not wanted.  Setting it for a shape whose kernel cannot produce them is DXMI_EINVAL. */
    /* optional: GroupNorm(+SiLU) of the conv's OUTPUT fused into its epilogue, for shapes whose workgroup tile holds whole
       images and whole groups (the 4x4 maps of the U-Net, and the 8x8 maps when gn_flags bit 1 drops the raw output:
       Normalize()+nonlinearity of the next layer, unet_small.py:119-126):
       gn_out = silu?(group_norm(round_bf16(out))) with statistics in fp32 (two-pass, from registers), bitwise independent of
       the batch.  gn_out NULL = off.  dxmi_conv2d_gn_fuse_supported(d) says whether the selected kernel can; setting it for a
       shape that cannot is DXMI_EINVAL. */
    void*        gn_out;     /* NHWC bf16 [N,OH,OW,Cout] */
    const float* gn_gamma;   /* [Cout] */
    const float* gn_beta;    /* [Cout] */
    float        gn_eps;
    int32_t      gn_groups;  /* Cout / gn_groups must be 8 */
    int32_t      gn_flags;   /* bit 0: SiLU after the affine; bit 1: do not write `out` (the raw tensor has no other reader) */
} dxmi_conv_desc;

int dxmi_conv2d_fwd(const dxmi_conv_desc* d, void* stream);
/* 1 when the kernel dxmi_conv2d_fwd would launch for `d` can fuse the GroupNorm of its output (gn_out ...), else 0. */
int dxmi_conv2d_gn_fuse_supported(const dxmi_conv_desc* d);
/* Measurement aid (round 5): (s_memtime, s_memrealtime) at the first and the last instruction of workgroup 0 of the most recent
 * wave-specialised 3x3 conv launch (conv_ws_kernel), copied to host memory: host_out4 = {cycles0, ref0, cycles1, ref1}; the shader
 * clock the chip held during that launch is (cycles1 - cycles0) / (ref1 - ref0) x 100 MHz.  Synchronises (a device-to-host copy). */
int dxmi_conv_ws_last_clock(unsigned long long* host_out4);
/* Partials per image (P above) the kernel dxmi_conv2d_fwd would launch for `d` writes into d->gn_stats; 0 = that kernel
 * does not produce block statistics (use dxmi_gn_block_stats on its output, or the one-pass dxmi_groupnorm_silu_fwd). */
int dxmi_conv2d_gn_stats_partials(const dxmi_conv_desc* d);
/* Template instantiation dxmi_conv2d_fwd would launch (MB*1000+NB*100+PMAX), no launch; <0 on error. */
int dxmi_conv2d_kernel_id(const dxmi_conv_desc* d);

/* Packs an fp32 OIHW weight [Cout,Cin,k,k] (device) into bf16 A-fragment order
 * [tap][Cin/16][ceil(Cout/32)][64 lanes][8].  transpose_flip=1 packs the data-gradient
 * (flipped, Cin<->Cout) operator instead.  k27=1 packs the Cin=3,k=3 im2col form.
 * dst must hold dxmi_packed_conv_weight_bytes(...) bytes. */
int64_t dxmi_packed_conv_weight_bytes(int32_t Cout, int32_t Cin, int32_t ksize, int32_t k27);
int dxmi_pack_conv_weight(const float* w_oihw, void* dst, int32_t Cout, int32_t Cin,
                          int32_t ksize, int32_t transpose_flip, int32_t k27, void* stream);

/* Multi-tensor form of dxmi_pack_conv_weight: `count` weights (a HOST array of descriptors; the library cuts it into launches
 * of DXMI_PACK_MAX items passed by value) — after an optimiser step (trainer.py:264, :325, :389; fp16_util.py:204-223) every
 * packed weight of a net is refreshed by a handful of launches instead of one per layer. */
#define DXMI_PACK_MAX 32
typedef struct dxmi_pack_item {
    const float* w;          /* fp32 OIHW [Cout,Cin,k,k] (transpose_flip: the ORIGINAL tensor [Cin_logical][Cout_logical][k][k]) */
    void*        dst;        /* dxmi_packed_conv_weight_bytes(Cout, Cin, ksize, k27) bytes */
    int32_t Cout, Cin, ksize, transpose_flip, k27;
} dxmi_pack_item;
int dxmi_pack_conv_weights(const dxmi_pack_item* items, int32_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the convolutions (training path: models/DxMI/trainer.py:261,322,387 call
 * .backward() through the value network and the U-Net).
 *  - data gradient: dxmi_conv2d_fwd with weights packed by dxmi_pack_conv_weight(transpose_flip=1)
 *    (stride-1 convs; mask_src/mask_slope fuse the activation derivative, residual fuses the
 *    skip-path gradient).
 *  - weight gradient: dxmi_conv2d_wgrad: dW[co][ci][ky][kx] = sum_p dY[p][co] X[p+tap][ci] as an MFMA
 *    GEMM over the pixel index (ds_read_b64_tr_b16 transposed fragments, split-K over pixel tiles,
 *    fixed-order reduction).  x0|x1: forward input (virtual concat), dy: output gradient, both NHWC
 *    bf16; dw_oihw fp32 [Cout,Cin,k,k]; accumulate != 0 adds into dw_oihw.
 *  - bias gradient: dxmi_colsum_bf16 over dy viewed as [P, C].
 * ---------------------------------------------------------------------------------------- */
int64_t dxmi_conv2d_wgrad_workspace_bytes(int32_t N, int32_t OH, int32_t OW, int32_t Cin, int32_t Cout,
                                          int32_t ksize);
int dxmi_conv2d_wgrad(const void* x0, int32_t C0, const void* x1, int32_t C1, const void* dy,
                      float* dw_oihw, void* workspace, int32_t N, int32_t IH, int32_t IW, int32_t OH,
                      int32_t OW, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t upsample,
                      int32_t accumulate, void* stream);
/* Same plus the bias gradient dbias[co] = sum_p dY[p][co] (fp32 [Cout]), summed from the dY tiles the kernel stages
 * anyway (replaces a separate column-sum pass over dY; trainer.py's loss.backward() for nn.Conv2d.bias). */
int dxmi_conv2d_wgrad_bias(const void* x0, int32_t C0, const void* x1, int32_t C1, const void* dy, float* dw_oihw,
                           float* dbias, void* workspace, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW,
                           int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t upsample,
                           int32_t accumulate, void* stream);
/* workspace: ceil(P/512) * C floats */
int dxmi_colsum_bf16(const void* x, float* out, void* workspace, int64_t P, int32_t C,
                     int32_t accumulate, void* stream);
/* out[b][c] = sum of rows [b*rows_per_block, (b+1)*rows_per_block) of x[P][C] (per-image channel sums). */
int dxmi_colsum_blocks_bf16(const void* x, float* out, int64_t P, int32_t C, int32_t rows_per_block,
                            void* stream);
/* out[b][c] = sum_n in[b][n][c], fp32, fixed order (the [2][N][C] per-image d(gamma) / d(beta) partials of
 * dxmi_groupnorm_silu_bwd -> [2][C]). */
int dxmi_colsum_f32(const float* in, float* out, int32_t B, int32_t N, int32_t C, void* stream);

/* Backward of GroupNorm(+SiLU) (training path).  x = [in0 | in1] and dy (gradient w.r.t. the forward
 * kernel's output, one dense [N,HW,C0+C1] tensor) -> dx0 [N,HW,C0], dx1 [N,HW,C1] (+ add0/add1 fused:
 * skip-connection / residual gradients), and per-image partial sums dgamma_part/dbeta_part [N,C] fp32
 * (sum over dim 0 gives the parameter gradients).  Statistics are recomputed, nothing is saved by the
 * forward.  Replaces autograd through unet_small.py:119-126,169,329-330. */
int dxmi_groupnorm_silu_bwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                            const void* add0, const void* add1, const float* gamma, const float* beta,
                            void* dx0, void* dx1, float* dgamma_part, float* dbeta_part, int32_t N,
                            int32_t HW, int32_t groups, float eps, int32_t apply_silu, void* stream);
/* 1 when dxmi_groupnorm_silu_bwd can serve the shape (x and dy both stay resident, so its limits are tighter than the
 * forward's), else 0 — use dxmi_groupnorm_generic_bwd then. */
int dxmi_groupnorm_silu_bwd_supported(int32_t C0, int32_t C1, int32_t HW, int32_t groups);

/* Batched bf16 GEMM on MFMA for attention backward: C[b] = alpha * A[b] (MxK) * B[b] (KxN), batch
 * b = (outer, inner) with element strides *_b0 / *_b1.  *_kcontig != 0: element (row,k) of that operand
 * is at row*ld + k; == 0: at k*ld + row (read through transposing LDS loads).  C row-major [M,N] with
 * c_ld, fp32 or bf16.  dxmi_softmax_bwd: per row of length T, P = softmax(S), dS = P o (dP - <dP,P>). */
int dxmi_bgemm_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t a_b0,
                    int64_t a_b1, int32_t a_ld, int32_t a_kcontig, int64_t b_b0, int64_t b_b1, int32_t b_ld,
                    int32_t b_kcontig, int64_t c_b0, int64_t c_b1, int32_t c_ld, int32_t c_f32, float alpha,
                    int32_t outer, int32_t inner, void* stream);
int dxmi_softmax_bwd(const float* S, const float* dP, void* P, void* dS, int64_t rows, int32_t T,
                     void* stream);

/* Fused attention backward (round 4) for head dimension 64 (every attention block of the ADM / EDM nets,
 * models/cm/unet.py:413-441): dqkv [N,T,3C] from qkv [N,T,3C], the forward output o [N,T,C] and its gradient dout [N,T,C]
 * (bf16), without any [T,T] tensor in HBM: the probabilities are recomputed from the row log-sum-exp (first kernel, which also
 * forms delta = <dO, O> and dQ), then dK / dV per key block (second kernel).  Fixed summation orders: bitwise reproducible.
 * workspace: dxmi_attention_bwd_workspace_bytes(N, T, heads) bytes.  dxmi_attention_bwd_supported: 1 when C / heads == 64
 * (other head sizes: the dxmi_bgemm_bf16 / dxmi_softmax_bwd sequence above). */
int dxmi_attention_bwd_supported(int32_t T, int32_t C, int32_t heads);
int64_t dxmi_attention_bwd_workspace_bytes(int32_t N, int32_t T, int32_t heads);
int dxmi_attention_bwd(const void* qkv, const void* o, const void* dout, void* dqkv, void* workspace, int32_t N, int32_t T,
                       int32_t C, int32_t heads, float scale, void* stream);

/* The saved softmax statistics of autograd's attention node (round 6): dxmi_attention_fwd_lse is dxmi_attention_fwd that also leaves the
 * row log-sum-exp of the scaled logits (log2 domain: max + log2(sum) of scale * log2(e) * q.k; fp32 [N][heads][T]); dxmi_attention_bwd_lse
 * is dxmi_attention_bwd taking it, one sweep over the keys shorter.  _supported: every shape of dxmi_attention_fwd except the
 * single-head 256-token x 256-channel block.  Replaces models/cm/unet.py:413-441 under autograd. */
int dxmi_attention_fwd_lse_supported(int32_t T, int32_t C, int32_t heads);
int dxmi_attention_fwd_lse(const void* qkv, void* out, float* lse, int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream);
int dxmi_attention_bwd_lse(const void* qkv, const void* o, const void* dout, void* dqkv, const float* lse_fwd, void* workspace, int32_t N,
                           int32_t T, int32_t C, int32_t heads, float scale, void* stream);

/* Backward of dxmi_pool_act: din = (pool ? 0.25 * upsample2(g) : g), g = dout * (act_out > 0 ? 1 : slope);
 * dout/act_out: [N,OH,OW,C], din: [N,H,W,C] (H = 2*OH when pool). */
int dxmi_pool_act_bwd(const void* dout, const void* act_out, void* din, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t pool, float slope, void* stream);
/* Backward of the value head w.r.t. its feature map: dfeat[n,p,c] = dy[n] * w[c] * (feat > 0); also
 * returns s[n,c] = sum_p relu(feat[n,p,c]) (fp32) from which the caller forms the tiny parameter
 * gradients. */
int dxmi_value_head_bwd(const void* feat, const float* w, const float* dy, void* dfeat, float* s,
                        int32_t N, int32_t HW, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU) over NHWC bf16, optionally over a virtual concat of two
 * tensors.  One pass over HBM: a workgroup keeps its (image, channel-slice) in registers,
 * two-pass mean/variance in fp32 with wavefront shuffles + LDS.
 * Replaces Normalize()+nonlinearity at models/DxMI/unet_small.py:35-36,119-120,125-126,
 * 169,329-330 (eps 1e-6) and GroupNorm32 at models/cm/nn.py:19-21 (eps 1e-5).
 * ---------------------------------------------------------------------------------------- */
int dxmi_groupnorm_silu_fwd(const void* in0, int32_t C0, const void* in1, int32_t C1,
                            const float* gamma, const float* beta, void* out,
                            int32_t N, int32_t HW, int32_t groups, float eps,
                            int32_t apply_silu, void* stream);

/* GroupNorm(+SiLU) as ONE streaming read + write, given block statistics of the input(s) (the statistics pass of
 * Normalize() / GroupNorm32 is folded into whatever produced the tensor: dxmi_conv_desc.gn_stats, or dxmi_gn_block_stats):
 *   stats0 / stats1: fp32 [N][P0|P1][C0|C1 / 2][2] (sum, sum of squares per channel pair and partial).  Every workgroup
 *   re-reduces its image's partials in a fixed order (pair outer, partial inner), mean = s/cnt, var = max(q/cnt - mean^2, 0),
 *   then streams its rows: y = x*(rstd*gamma) + (beta - mean*rstd*gamma), optional FiLM scale-shift
 *   (y*(1+scale[n,c]) + shift[n,c]: scale_shift fp32 [N, ss_ld], scale at [c], shift at [C + c]; models/cm/unet.py:252-256),
 *   optional SiLU, one rounding to bf16.
 * Channels per group must be even, C0 and C1 multiples of 8.  No residency limit on HW.
 * dxmi_gn_block_stats: the statistics of a tensor that has no producer-side statistics: x [N,HW,C] -> stats
 *   [N][P][C/2][2] with P = dxmi_gn_block_stats_partials(HW) row chunks per image.
 * dxmi_gn_stats_fold: [N][P][C/2][2] -> [N][ceil(P/group)][C/2][2], `group` consecutive partials added in order (large maps:
 *   a conv writes one partial per 128-pixel half tile — 512 per 256x256 image — too many for every apply workgroup to
 *   re-add); C = floats per partial (= channels). */
int dxmi_gn_block_stats_partials(int32_t HW);
int dxmi_gn_block_stats(const void* x, float* stats, int32_t N, int32_t HW, int32_t C, void* stream);
int dxmi_gn_stats_fold(const float* stats, float* out, int32_t N, int32_t P, int32_t C, int32_t group, void* stream);
int dxmi_groupnorm_apply(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                         const float* stats1, int32_t P1, const float* gamma, const float* beta, const float* scale_shift,
                         int32_t ss_ld, void* out, int32_t N, int32_t HW, int32_t groups, float eps, int32_t apply_silu,
                         void* stream);

/* dxmi_groupnorm_apply as TWO launches (round 6): a one-workgroup-per-image kernel forms the per-(image, channel) scale / offset
 * pairs (statistics partials -> group mean / rstd -> gamma, beta and the FiLM scale-shift folded in) into ab_workspace
 * (N * (C0 + C1) * 2 floats), and the streaming pass reads them instead of running that prologue in every one of its workgroups.
 * Same operations in the same order: bit-identical output.  Pays when the extra launch is cheap (a node of a replayed hipGraph). */
int dxmi_groupnorm_apply_split(const void* in0, int32_t C0, const float* stats0, int32_t P0, const void* in1, int32_t C1,
                               const float* stats1, int32_t P1, const float* gamma, const float* beta,
                               const float* scale_shift, int32_t ss_ld, void* out, float* ab_workspace, int32_t N, int32_t HW,
                               int32_t groups, float eps, int32_t apply_silu, void* stream);

/* Block statistics ([N][P][C/2][2], as dxmi_conv_desc.gn_stats / dxmi_gn_block_stats write them; st1 for the second part of a virtual
 * concat or NULL) -> the statistics partials of the generic GroupNorm kernels (dxmi_groupnorm_generic_workspace_bytes(N, HW, C) bytes:
 * what dxmi_groupnorm_generic_bwd_saved takes as fwd_stats): a training forward that normalised with dxmi_groupnorm_apply hands the
 * generic backward the same sums without another pass over the input. */
int dxmi_gn_blockstats_to_generic(const float* stats0, int32_t P0, int32_t C0, const float* stats1, int32_t P1, int32_t C1,
                                  float* generic_stats, int32_t N, int32_t HW, int32_t groups, void* stream);

/* Parameter and FiLM gradients of a scale-shift GroupNorm from dxmi_groupnorm_generic_bwd's g_out [2][N][C] in one launch:
 * d_scale_shift [N][2C] = (G1 gamma + G0 beta | G0), dgamma [C] = sum_n G1 (1 + scale), dbeta [C] = sum_n G0 (1 + scale); scale_shift as the
 * backward took it ([N][ss_ld], scale first).  Replaces the reference's autograd of models/cm/unet.py:252-256. */
int dxmi_gn_ss_grads(const float* g, const float* scale_shift, int32_t ss_ld, const float* gamma, const float* beta,
                     float* d_scale_shift, float* dgamma, float* dbeta, int32_t N, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single-/multi-head self-attention over a fused qkv tensor, MFMA QK^T and PV with an
 * online softmax.  qkv: NHWC bf16 [N,T,3*C] laid out [q | k | v] along channels, heads are
 * contiguous channel blocks of size C/heads.  out: [N,T,C] bf16.  scale multiplies q.k.
 * Replaces models/DxMI/unet_small.py:175-187 (bmm/softmax/bmm, scale C^-0.5) and
 * models/cm/unet.py:413-441 (QKVAttentionLegacy).
 * ---------------------------------------------------------------------------------------- */
int dxmi_attention_fwd(const void* qkv, void* out, int32_t N, int32_t T, int32_t C,
                       int32_t heads, float scale, void* stream);

/* AttnBlock tail fused behind the attention (reference unet_small.py:187-190: h_ = proj_out(h_); return x + h_), for the
 * single-head 256-token x 256-channel block of the CIFAR-10 net (dxmi_attention_proj_supported): the attention output never
 * leaves the CU — it stays in registers as the MFMA operand of the 1x1 projection — and out = x + proj_out(attention) + bias is
 * written once.  Same arithmetic as dxmi_attention_fwd followed by dxmi_conv2d_fwd (attention output rounded to bf16, fp32
 * accumulation, one rounding of the sum).  wproj_packed: 128 KiB from dxmi_pack_attn_proj_weight (fp32 [256][256] OI weight).
 * Inference path only: training keeps the two launches (the projection's weight gradient needs the attention output). */
int dxmi_attention_proj_supported(int32_t T, int32_t C, int32_t heads);
int dxmi_pack_attn_proj_weight(const float* w_oi, void* dst, void* stream);
int dxmi_attention_proj_fwd(const void* qkv, const void* wproj_packed, const float* bias, const void* residual, void* out,
                            float* gn_stats, int32_t N, int32_t T, int32_t C, int32_t heads, float scale, void* stream);
/* gn_stats (optional): GroupNorm block statistics of `out`, fp32 [N][8][C/2][2] — one partial per 32 tokens, the layout
 * dxmi_groupnorm_apply reads (P = 8). */

/* The whole AttnBlock in ONE launch (round 5; reference models/DxMI/unet_small.py:167-191: h_ = norm(x); q, k, v = 1x1 convs of
 * h_; w_ = softmax(q^T k * C^-0.5); h_ = proj_out(v w_^T); return x + h_), for the single-head 256-token x 256-channel blocks of
 * the CIFAR-10 net (dxmi_attn_block_supported), inference path only.  x: NHWC bf16 [N,256,256]; stats: its GroupNorm block
 * statistics fp32 [N][P][128][2], P <= 8 (dxmi_conv_desc.gn_stats / dxmi_gn_block_stats layout; dxmi_gn_stats_fold folds more);
 * gamma / beta: the block's norm.  The
 * normalised input, q, k, v and the attention output never exist in memory: with G = scale Wk^T Wq and W' = Wproj Wv folded once
 * per weight version (dxmi_attn_block_pack: fp32 [256][256] OI weights and [256] biases -> dxmi_attn_block_packed_bytes() bytes)
 * the raw x tile in LDS is K, V and the residual at once (terms constant along the key axis drop out of the softmax; rows of the
 * softmax sum to one).  out: [N,256,256] bf16; out_stats (optional): block statistics of out, fp32 [N][8][128][2].
 * Tolerance against the reference block in fp32: the same as the three-launch form it replaces (dxmi_groupnorm_apply +
 * dxmi_conv2d_fwd + dxmi_attention_proj_fwd), see tests/test_hip_round5_kernels.py. */
int dxmi_attn_block_supported(int32_t T, int32_t C, int32_t heads, int32_t groups);
int64_t dxmi_attn_block_packed_bytes(void);
int dxmi_attn_block_pack(const float* wq, const float* bq, const float* wk, const float* wv, const float* bv,
                         const float* wproj, const float* bproj, float scale, void* dst, void* stream);
int dxmi_attn_block_fwd(const void* x, const float* stats, int32_t P, const float* gamma, const float* beta, float eps,
                        const void* packed, void* out, float* out_stats, int32_t N, int32_t T, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Timestep-embedding path.
 * dxmi_timestep_embedding: out[N,dim] fp32 sinusoid of t; freq_i = exp(-ln(max_period)*i/denom)
 *   order 0 = [sin | cos], denom = dim/2 - 1   (get_timestep_embedding, unet_small.py:9-27)
 *   order 1 = [cos | sin], denom = dim/2       (timestep_embedding, models/cm/nn.py:119-137)
 * dxmi_linear_fwd: out[P,M] = post(pre(x[P,K]) @ W[M,K]^T + b): fp32 rows in/out, bf16 MFMA
 *   operands (same kernel as the convs).  Serves temb.dense.{0,1} (unet_small.py:297-299), the
 *   concatenation of all temb_proj layers in ONE launch (unet_small.py:123; pre = swish) and
 *   the EDM time_embed / emb_layers (models/cm/unet.py:775-779, :249).
 *   wpacked = dxmi_pack_conv_weight(W as [M,K,1,1]).
 * ---------------------------------------------------------------------------------------- */
int dxmi_timestep_embedding(const float* t, float* out, int32_t N, int32_t dim, int32_t order,
                            float max_period, void* stream);
int dxmi_linear_fwd(const float* x, const void* wpacked, const float* bias, float* out,
                    int32_t P, int32_t K, int32_t M, int32_t pre_act, int32_t post_act,
                    void* stream);
/* Split-K form for skinny products with a long K (the data gradient of the ADM nets' concatenated emb_layers, models/cm/unet.py:249:
 * 16 rows x K ~ 30 000 -> 768 columns): partials[s][P][M], s < dxmi_linear_splitk_slices(P, K, M) (0: unsupported shape, 1: no
 * split pays), each slice the product over its K range; the caller sums the slices in order and adds bias / activation. */
int dxmi_linear_splitk_slices(int32_t P, int32_t K, int32_t M);
int dxmi_linear_splitk(const float* x, const void* wpacked, float* partials, int32_t P, int32_t K, int32_t M, int32_t pre_act,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused VAR sampler transition (models/DxMI/var_sampler.py:262-295 and :373-407):
 *   xs = x*xmul[b]; control = cmul[b]*eps; mean = xs+control; x' = mean + sigma[b]*z;
 *   logp[b] = mean_CHW Normal(mean, sigma).log_prob(x').
 * assoc selects the reference's floating-point association of the update: 0 = mean + sigma*z
 * (sample_step, :399), 1 = xs + (control + sigma*z) (VAR_sampling, :285).
 * Per-sample scalars are read from device vectors, so one kernel serves both the T-step
 * loop (all entries equal) and sample_step with per-sample integer t.
 * ---------------------------------------------------------------------------------------- */
int dxmi_var_step_fwd(const float* x, const float* eps, const float* z,
                      const float* xmul, const float* cmul, const float* sigma,
                      float* x_next, float* mean, float* control, float* logp,
                      int32_t N, int32_t CHW, int32_t assoc, void* stream);

/* Backward of the VAR transition for the differentiable `sample_step` of the policy update (reference var_sampler.py:357-408 under
 * torch autograd): with x' = xm x + c eps + sigma z, mean = xm x + c eps, control = c eps and logp as dxmi_var_step_fwd forms it
 * (x' detached inside), given the gradients of the loss w.r.t. x' / mean / control ([N, CHW]) and logp ([N]) — any may be NULL —
 *   d_eps = c (g_next + g_mean + g_control + g_logp z / (sigma CHW))       d_sigma[n] = sum g_next z + g_logp (mean z^2 - 1) / sigma
 * (cmul = theta multiplier x adhoc_scale1, per sample).  The sigma gradient is reduced in-kernel in a fixed order; the caller chains it
 * into log_betas (sigma = exp(log_betas[t])). */
int dxmi_var_step_bwd(const float* g_next, const float* g_mean, const float* g_control, const float* g_logp, const float* z,
                      const float* cmul, const float* sigma, float* d_eps, float* d_sigma, int32_t N, int32_t CHW, void* stream);

/* Backward of the EDM transition (openai_diffusion.py:71-94): d_model_out = -(c_out (sigma_down - sigma) / sigma) (g_sample + g_mean),
 * d_sigma_up[n] = sum g_sample z; g_sample / g_mean may be NULL. */
int dxmi_edm_step_bwd(const float* g_sample, const float* g_mean, const float* z, const float* sigma, const float* sigma_down,
                      float* d_model_out, float* d_sigma_up, int32_t N, int32_t CHW, float sigma_data, void* stream);

/* INT path: per-sample gather of schedule tables by integer timestep
 * (var_sampler.py:363-376,389; models/diffusion.py:18-22).  Writes
 *   tau[b] = continuous_steps[t[b]], xmul[b], cmul[b] (theta multiplier), sigma[b] =
 *   exp(log_betas_all[t[b]]).  Bit-exact w.r.t. the fp32 tables. */
int dxmi_var_gather_sched(const int64_t* t, const float* continuous_steps,
                          const float* xmul_tab, const float* cmul_tab,
                          const float* log_betas_all, float* tau, float* xmul, float* cmul,
                          float* sigma, int32_t N, int32_t T, void* stream);

/* Elementwise helpers of the value network (models/modules.py:96-101): 2x2 average pool
 * (optional) followed by activation, NHWC bf16. */
int dxmi_pool_act(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C,
                  int32_t pool, int32_t act, void* stream);
/* Value head (models/modules.py:150-158): relu -> sum over HxW -> Linear(C,1) -> out_scale
 * (Linear(1,1): y*out_w[0]+out_b[0], both device scalars, or both NULL). */
int dxmi_value_head(const void* in, const float* w, const float* b, const float* out_w,
                    const float* out_b, float* out, int32_t N, int32_t HW, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * EDM / ADM U-Net path (models/cm/unet.py, models/cm/karras_diffusion.py, models/DxMI/openai_diffusion.py).
 * dxmi_groupnorm_generic_fwd: GroupNorm32 (+ scale-shift norm, + SiLU) for any channels-per-group and
 *   map size (two kernels: coalesced partial statistics, then apply); scale_shift: fp32 [N, ss_ld] with
 *   scale at [c] and shift at [C + c] (unet.py:252-256) or NULL.  workspace from
 *   dxmi_groupnorm_generic_workspace_bytes.
 * dxmi_upsample2x: nearest x2 (ResBlock up, x branch; unet.py:197-198).
 * dxmi_edm_precond: x_in = c_in(sigma)*x, t = 250 ln(sigma + 1e-44)   (karras_diffusion.py:64-68,348-349).
 * dxmi_edm_step_fwd: fused Euler-ancestral transition (openai_diffusion.py:71-94): denoised =
 *   c_out F + c_skip x; d = (x - denoised)/sigma; mean = x + d (sigma_down - sigma); sample = mean + z sigma_up.
 * ---------------------------------------------------------------------------------------- */
int dxmi_groupnorm_silu_supported(int32_t C0, int32_t C1, int32_t HW, int32_t groups); /* 1: one-pass kernel serves it */
int64_t dxmi_groupnorm_generic_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int dxmi_groupnorm_generic_fwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const float* gamma,
                               const float* beta, const float* scale_shift, int32_t ss_ld, void* out,
                               void* workspace, int32_t N, int32_t HW, int32_t groups, float eps,
                               int32_t apply_silu, void* stream);
/* Backward of dxmi_groupnorm_generic_fwd: dx0 | dx1 (bf16, optional additive inputs add0 | add1 fused) and
 * g_out fp32 [2][N][C] = per-image sums of dyy and dyy*xhat, from which the caller forms dgamma, dbeta and the FiLM
 * gradients (dscale = G1*gamma + G0*beta, dshift = G0).  Replaces autograd through models/cm/unet.py:228-260. */
int64_t dxmi_groupnorm_generic_bwd_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int dxmi_groupnorm_generic_bwd(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                               const void* add0, const void* add1, const float* gamma, const float* beta,
                               const float* scale_shift, int32_t ss_ld, void* dx0, void* dx1, float* g_out,
                               void* workspace, int32_t N, int32_t HW, int32_t groups, float eps,
                               int32_t apply_silu, void* stream);
/* Same with the forward's statistics kept by the caller: fwd_stats = the first dxmi_groupnorm_generic_workspace_bytes bytes of the
 * workspace dxmi_groupnorm_generic_fwd was given (per-chunk group sums; untouched by the forward's second kernel), which saves the
 * backward its own statistics pass over the input (autograd keeps mean / rstd the same way).  fwd_stats NULL: recomputed. */
int dxmi_groupnorm_generic_bwd_saved(const void* in0, int32_t C0, const void* in1, int32_t C1, const void* dy,
                                     const void* add0, const void* add1, const float* gamma, const float* beta,
                                     const float* scale_shift, int32_t ss_ld, void* dx0, void* dx1, float* g_out,
                                     const float* fwd_stats, void* workspace, int32_t N, int32_t HW, int32_t groups, float eps,
                                     int32_t apply_silu, void* stream);
int dxmi_upsample2x(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int dxmi_edm_precond(const float* x, const float* sigma, float* x_in, float* t_out, int32_t N, int32_t CHW,
                     float sigma_data, void* stream);
int dxmi_edm_step_fwd(const float* x, const float* model_out, const float* z, const float* sigma,
                      const float* sigma_down, const float* sigma_up, float* sample, float* mean, int32_t N,
                      int32_t CHW, float sigma_data, void* stream);

/* im2col of a 3-channel NCHW fp32 image (3x3, stride 1, pad 1): out [N,H,W,64] bf16, channel
 * k = ci*9+ky*3+kx for k < 27, zero above — the activation operand of the stem convs' weight gradient
 * (dxmi_conv2d_wgrad with ksize 1). */
int dxmi_im2col27(const float* x, void* out, int32_t N, int32_t H, int32_t W, void* stream);

/* Output stage of generate_cifar10.py / generate_large.py: NCHW fp32 sampler output -> uint8 pixels (NHWC when out_nhwc,
 * else NCHW), the reference's fp32 operation order, no fused multiply-add between its steps (bit-exact pixel values):
 *   mode 0: v = (x - (-1)) / 2 -> clamp(0,1) -> * 255 -> + 0.5 -> clamp(0,255) -> truncate
 *           (rescale(), generate_cifar10.py:41-42 and :205-209; generate_large.py:36-41; torchvision.utils.save_image)
 *   mode 1: (x + 1) * 127.5 -> clamp(0,255) -> truncate   (the FID / samples_N.npz array, generate_large.py:43)
 * HW must be a multiple of 4. */
int dxmi_quantize_u8(const float* x, void* out, int32_t N, int32_t C, int32_t HW, int32_t mode, int32_t out_nhwc,
                     void* stream);

/* Layout converters at the network edge. */
int dxmi_nchw_f32_to_nhwc_bf16(const float* in, void* out, int32_t N, int32_t C, int32_t HW,
                               void* stream);
int dxmi_nhwc_bf16_to_nchw_f32(const void* in, float* out, int32_t N, int32_t C, int32_t HW,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * Train-step tail: multi-tensor optimiser steps, gradient-norm clip, dropout, replay-buffer gathers.
 * Tensor lists are HOST arrays of device pointers / element counts (any length; the library cuts them into
 * launches of DXMI_MT_MAX tensors passed by value in the kernel arguments — no descriptor upload).
 * ---------------------------------------------------------------------------------------- */
#define DXMI_MT_MAX 64

/* Workgroups one multi-tensor launch series uses over `numel[0..count)` (= fp32 partials dxmi_gradnorm_clip needs). */
int64_t dxmi_mt_blocks(const int64_t* numel, int32_t count);

/* torch.optim.Adam.step() over fp32 tensors (models/DxMI/trainer.py:264, :325, :389 with the optimisers of
 * train_cifar10.py:283-296), the arithmetic of torch's _multi_tensor_adam (no amsgrad / weight decay), every
 * intermediate rounded to fp32 where the foreach implementation stores one:
 *   m = m + (1-b1)(g - m);  v = v*b2 + (1-b2) g*g;  p = p + step_size[i] * m / (sqrt(v)/bc2_sqrt + eps)
 * step_size[i] = -(lr_i / (1 - b1^t)) and bc2_sqrt = sqrt(1 - b2^t) are formed by the caller in double, as torch does;
 * scalar hyper-parameters cross the ABI as double and are rounded to fp32 once (ATen's Scalar -> float conversion).
 * grad_scale: optional DEVICE scalar multiplied into every gradient first (the clip coefficient of
 * dxmi_gradnorm_clip: clip-then-step without a host round trip); write_back_grad also stores the scaled gradient. */
int dxmi_adam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                   const int64_t* numel, const float* step_size, int32_t count, double beta1, double beta2,
                   double eps, double bc2_sqrt, const float* grad_scale, int32_t write_back_grad, void* stream);

/* The same update for a step that is REPLAYED from a hipGraph (dxmi_hip/graph.py: kernel arguments are frozen at capture, the
 * step count is not): the step-dependent scalars are read from DEVICE memory, hyper = fp32 [bc2_sqrt, step_size[0..count)],
 * formed by the host exactly as for dxmi_adam_step and uploaded before every replay.  Bit-identical results. */
int dxmi_adam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                       const int64_t* numel, int32_t count, double beta1, double beta2, double eps, const float* hyper,
                       const float* grad_scale, int32_t write_back_grad, void* stream);

/* torch.optim.RAdam.step() (MixedPrecisionTrainer.optimize, models/cm/fp16_util.py:204-223; optimiser built at
 * train_image_large.py:153-160), arithmetic of torch/optim/radam.py _single_tensor_radam:
 *   p -= ((m/bc1) * lr_i) * (bc2_sqrt / (sqrt(v) + eps)) * rect      (rect < 0: p -= (m/bc1) * lr_i, rho_t <= 5)
 * grad_scale: optional DEVICE scalar (1 / 2^lg_loss_scale); found_inf: optional DEVICE flag, non-zero skips the whole
 * step on the device (the overflow test of fp16_util.py:208-212 without a host sync in front of the kernels). */
int dxmi_radam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                    const int64_t* numel, const float* lr, int32_t count, double beta1, double beta2, double eps,
                    double bc1, double bc2_sqrt, double rect, const float* grad_scale, const float* found_inf,
                    void* stream);

/* dxmi_radam_step for a hipGraph-replayed step: hyper = DEVICE fp32 [fp32(1/bc1), bc2_sqrt, rect, lr[0..count)]. */
int dxmi_radam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                        const int64_t* numel, int32_t count, double beta1, double beta2, double eps, const float* hyper,
                        const float* grad_scale, const float* found_inf, void* stream);

/* torch.nn.utils.clip_grad_norm_(params, max_norm) (trainer.py:388, :666-667; also MixedPrecisionTrainer._compute_norms
 * fp16_util.py:232-240 with max_norm <= 0 = "norm only"): out3[0] = global L2 norm, out3[1] = min(1, max_norm/(norm+1e-6)),
 * out3[2] = 1 if the norm is inf/nan.  Fixed-order reductions (reproducible).  partials: >= dxmi_mt_blocks() floats.
 * scale_in_place: multiply the gradients by out3[1] (otherwise hand out3+1 to dxmi_adam_step as grad_scale). */
int dxmi_gradnorm_clip(void* const* grads, const int64_t* numel, int32_t count, float max_norm, float* partials,
                       float* out3, int32_t scale_in_place, void* stream);

/* nn.Dropout(p) of ResnetBlock (models/DxMI/unet_small.py:129), NHWC bf16: y[i] = keep(i) ? bf16(x[i]/(1-p)) : 0 with
 * keep(i) = (mix32(i ^ seed) >> 8) >= p*2^24, mix32 = the 32-bit finaliser x^=x>>16; x*=0x7feb352d; x^=x>>15;
 * x*=0x846ca68b; x^=x>>16.  The backward pass calls it again on the gradient with the same seed (no stored mask). */
int dxmi_dropout_bf16(const void* x, void* y, int64_t n, float p, uint32_t seed, void* stream);
/* ... with the seed read from DEVICE memory (a dropout site of a hipGraph-replayed step: the host uploads a fresh seed
 * before every replay). */
int dxmi_dropout_bf16_dev(const void* x, void* y, int64_t n, float p, const uint32_t* seed, void* stream);

/* TD step of DxMI_Trainer.update_f_v on the replay ring, data side in one launch (reference trainer.py:271-300, :163-169): rows
 * `state_rows[b]` (and `next_rows[b]`, or the dense re-drawn next states of value_resample) of the ring's fp32 [n_src_rows, CHW]
 * trajectory block are gathered into out_state / out_next — the two halves of the batch [next_state | state] the value net
 * evaluates in one forward — and cost[b] = mean_CHW (x' - x)^2 / (2 beta) is reduced on the way.  INT path bit-exact; an index
 * outside [0, n_src_rows) fills the row with NaN.  beta: DEVICE scalar (betas_for_q[T - 1 - t]).  out_next may be NULL. */
int dxmi_td_gather_cost(const float* traj, const int64_t* state_rows, const int64_t* next_rows, const float* next_dense,
                        const float* beta, float* out_next, float* out_state, float* cost, int32_t B, int32_t CHW,
                        int64_t n_src_rows, void* stream);

/* TD loss of one step and its gradient (trainer.py:300-302: F.mse_loss(v(x_t), (v(x_t+1) + extra).detach())): v = the value
 * net's output on [next_state | state] ([2B]); grad[0..B) = 0, grad[B + i] = 2 (v[B + i] - v[i] - extra) / B;
 * logs3 = (loss, mean v(x_t), mean cost).  extra: DEVICE scalar (time-cost terms of the step) or NULL. */
int dxmi_td_loss(const float* v, const float* cost, const float* extra, float* grad, float* logs3, int32_t B, void* stream);

/* Parameter gradients of the value head (models/modules.py:146-163: relu, spatial sum, Linear(C,1), out_scale Linear(1,1)) in one
 * launch: s = the relu-sum features [N, C] dxmi_value_head_bwd returns, dy [N]; out = fp32 [C + 3] = d linear.weight |
 * d linear.bias | d out_scale.weight | d out_scale.bias (zeros when out_w is NULL).  Fixed summation order. */
int dxmi_value_head_pgrad(const float* s, const float* w, const float* b, const float* dy, const float* out_w, float* out,
                          int32_t N, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * InceptionV3 feature extractor of the FID evaluation (reference pytorch_fid/inception.py:16-163 with the FID patches
 * :193-310, called as `model(batch)[0]` by pytorch_fid/fid_score.py:170-221).  Shape-agnostic kernels (the net's maps are 149 ... 8
 * pixels wide, its kernels 1x1 ... 7x1): NHWC bf16 activations whose channel count is a multiple of 16, outputs written at a
 * channel offset of a wider tensor so that the blocks' concatenations are never materialised.
 * ---------------------------------------------------------------------------------------- */

/* bf16 elements of a packed generic-conv weight: [ceil32(Cout)][KH * KW][ceil16(Cin)]. */
int64_t dxmi_gconv_packed_elems(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW);

/* BasicConv2d's weights for dxmi_gconv_fwd (torchvision Inception3: Conv2d(bias=False) + BatchNorm2d(eps) + ReLU): w fp32
 * [Cout, Cin, KH, KW]; the BatchNorm scale gamma / sqrt(var + eps) is folded into the bf16 weights, bias[ceil32(Cout)] = beta -
 * mean * scale.  bn_gamma NULL: plain conv (bias zero). */
int dxmi_gconv_pack(const float* w, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                    float bn_eps, void* w_packed, float* bias, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, void* stream);

/* out[n, oy, ox, out_coff + co] = relu?(bias[co] + sum x[n, oy*SH - PH + ky, ox*SW - PW + kx, ci] * w[co][ky, kx][ci]) — implicit
 * GEMM on the bf16 MFMA, fp32 accumulation; x NHWC bf16 with Cin % 16 == 0 channels (the packed weight's padded count), out NHWC
 * bf16 with out_cstride channels per pixel.  OH = (IH + 2 PH - KH) / SH + 1 (likewise OW). */
int dxmi_gconv_fwd(const void* x, const void* w_packed, const float* bias, void* out, int32_t N, int32_t IH, int32_t IW,
                   int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t PH, int32_t PW,
                   int32_t out_cstride, int32_t out_coff, int32_t relu, void* stream);

/* 3x3 pooling, NHWC bf16: max (padding = -inf; F.max_pool2d) or, avg_exclude_pad != 0, the average over the window's IN-BOUNDS
 * pixels only (F.avg_pool2d(count_include_pad=False): the FID patch of inception.py:209-212).  pad 0 or 1. */
int dxmi_pool3x3(const void* x, void* out, int32_t N, int32_t IH, int32_t IW, int32_t C, int32_t stride, int32_t pad,
                 int32_t avg_exclude_pad, int32_t out_cstride, int32_t out_coff, void* stream);

/* adaptive_avg_pool2d(x, 1): [N, HW, C] bf16 -> [N, C] fp32 (pixels summed in order). */
int dxmi_global_avgpool(const void* x, float* out, int32_t N, int32_t HW, int32_t C, void* stream);

/* F.interpolate(x, (OH, OW), mode='bilinear', align_corners=False) (+ 2 x - 1 when normalize; inception.py:146-153): NCHW fp32
 * [N, 3, IH, IW] -> NHWC bf16 [N, OH, OW, 16] (channels 3..15 zero). */
int dxmi_resize_bilinear_nhwc16(const float* x, void* out, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW,
                                int32_t normalize, void* stream);

/* Replay-buffer row gather (INT path; trainer.py:278-289 `state_dict[key][indices][train_indices]`, :357-359):
 * dst[r] = src[idx[r]], rows of row_bytes (multiple of 4), idx int64 on the device, negative indices wrap; an index
 * outside [-n_src_rows, n_src_rows) fills the row with 0xFF bytes instead of reading out of bounds. */
int dxmi_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t n_rows, int64_t n_src_rows,
                     int64_t row_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * FID activation statistics (the part that needs no Inception weights).
 * Replaces, on the device, the host-side numpy statistics the reference takes of the gathered activations:
 *   m1, s1 = np.mean(act, axis=0), np.cov(act, rowvar=False)
 * at train_image_large.py:62-69 (fid()), train_cifar10.py (fid()), and pytorch_fid/fid_score.py:283-303
 * (calculate_activation_statistics).  act: fp32 [N, D] row-major (D % 4 == 0: pytorch_fid's feature sizes are 64 / 192 /
 * 768 / 2048) -> mu fp64 [D], sigma fp64 [D, D] (unbiased, N - 1, as np.cov).  Column means in fp64; the centred Gram matrix on
 * the f32-input MFMA (exact fp32 products, fp32 accumulation inside a row split), splits folded in fp64 in a fixed order
 * (bitwise reproducible).  workspace: dxmi_fid_stats_workspace_bytes(N, D) bytes.
 * The Frechet distance itself (a 2048 x 2048 matrix square root, fid_score.py:224-281) stays the reference's float64 host
 * algorithm (scipy.linalg.sqrtm) in pytorch_fid/fid_score.py of this package.
 * ---------------------------------------------------------------------------------------- */
int64_t dxmi_fid_stats_workspace_bytes(int64_t N, int32_t D);
int dxmi_fid_stats(const float* act, int64_t N, int32_t D, double* mu, double* sigma, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DXMI_HIP_H */
