// InceptionV3 feature extractor of the FID evaluation on gfx950 (reference pytorch_fid/inception.py:16-163, :193-310 on
// torchvision's Inception3; pytorch_fid/fid_score.py:170-221 `model(batch)[0]`).  Off the hot path — a generation run scores
// 50 000 images once — so these are plain, shape-agnostic kernels, not tuned ones: the network's maps are 149, 147, 73, 71, 35, 17
// and 8 pixels wide (the product conv kernels want powers of two), its kernels 1x1, 3x3, 5x5, 1x7, 7x1, 1x3 and 3x1.
//   generic_conv_kernel   implicit GEMM on v_mfma_f32_32x32x16_bf16, operands straight from global memory: lane l of a wave holds
//                         8 consecutive input channels of (pixel l % 32, tap) as the B fragment and 8 consecutive K entries of
//                         (cout l % 32) as the A fragment — NHWC bf16 activations and [cout][tap][cin] bf16 weights make both one
//                         16-byte load.  Epilogue: + bias (BatchNorm folded into the weights at pack time), ReLU, bf16, written at
//                         a channel offset of a wider tensor (the Inception blocks' concatenations are never materialised).
//   pool3x3_kernel        3x3 max pool (padding = -inf) / average pool that does NOT count the padded zeros (the FID patches:
//                         inception.py:209-212, :299-301), any stride / padding, same channel-offset output.
//   global_avgpool_kernel adaptive_avg_pool2d(x, 1): [N, HW, C] bf16 -> [N, C] fp32, pixels summed in order.
//   resize_norm_kernel    F.interpolate(x, (OH, OW), mode='bilinear', align_corners=False) and 2 x - 1 (inception.py:146-153) from
//                         NCHW fp32 to NHWC bf16 with the 3 channels padded to 16 (zeros).
#include "common.h"

namespace {

struct GConvArgs {
    const bf16* x;       // [N, IH, IW, Cin]  Cin % 16 == 0
    const bf16* w;       // [CoutP, KH * KW, Cin]  CoutP % 32 == 0 (rows >= Cout are zero)
    const float* bias;   // [CoutP]
    bf16* out;           // [N, OH, OW, out_cs] written at channel offset out_co
    int N, IH, IW, Cin, OH, OW, Cout, CoutP, KH, KW, SH, SW, PH, PW, out_cs, out_co, relu;
};

__global__ __launch_bounds__(256) void generic_conv_kernel(GConvArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wp = wave & 1, wc = wave >> 1;                 // pixel half / cout half of the 64 x 64 workgroup tile
    const long P = (long)p.N * p.OH * p.OW;
    const long pix = (long)blockIdx.x * 64 + wp * 32 + (lane & 31);
    const int co_row = blockIdx.y * 64 + wc * 32;            // first cout of this wave
    if (co_row >= p.CoutP) return;
    const int half = lane >> 5;                               // which 8 of the 16 k values of an MFMA step
    const bool pvalid = pix < P;
    int n = 0, oy = 0, ox = 0;
    if (pvalid) {
        n = (int)(pix / ((long)p.OH * p.OW));
        const int r = (int)(pix - (long)n * p.OH * p.OW);
        oy = r / p.OW;
        ox = r - oy * p.OW;
    }
    const int taps = p.KH * p.KW;
    const bf16* wrow = p.w + ((size_t)(co_row + (lane & 31)) * taps) * p.Cin + half * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (bf16)0.f;
    for (int ky = 0; ky < p.KH; ++ky) {
        const int iy = oy * p.SH - p.PH + ky;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int ix = ox * p.SW - p.PW + kx;
            const bool ok = pvalid && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const bf16* xp = p.x + (((size_t)n * p.IH + (ok ? iy : 0)) * p.IW + (ok ? ix : 0)) * p.Cin + half * 8;
            const bf16* wp_ = wrow + (size_t)(ky * p.KW + kx) * p.Cin;
            for (int c = 0; c < p.Cin; c += 16) {
                const bf16x8 b = ok ? *reinterpret_cast<const bf16x8*>(xp + c) : zero;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(wp_ + c);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
        }
    }
    if (!pvalid) return;
    bf16* op = p.out + (size_t)pix * p.out_cs + p.out_co;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co_row + (r & 3) + 8 * (r >> 2) + 4 * half;        // accumulator row of element r
        if (co < p.Cout) {
            float v = acc[r] + p.bias[co];
            if (p.relu) v = v > 0.f ? v : 0.f;
            op[co] = (bf16)v;
        }
    }
}

// weights [Cout, Cin, KH, KW] fp32 (+ BatchNorm: y = (conv - mean) * gamma / sqrt(var + eps) + beta) -> [CoutP][KH*KW][CinP] bf16 with
// the scale folded in, bias[CoutP] = beta - mean * scale; padded rows / channels are zero
__global__ void pack_gconv_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float* __restrict__ mean, const float* __restrict__ var, float eps, bf16* __restrict__ wp,
                                  float* __restrict__ bias, int Cout, int Cin, int KH, int KW, int CoutP, int CinP) {
    const long total = (long)CoutP * KH * KW * CinP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinP);
        long r = i / CinP;
        const int t = (int)(r % (KH * KW));
        const int co = (int)(r / (KH * KW));
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            const float sc = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.f;
            v = w[(((size_t)co * Cin + ci) * KH + t / KW) * KW + t % KW] * sc;
        }
        wp[i] = (bf16)v;
    }
    for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < CoutP; co += gridDim.x * blockDim.x) {
        float b = 0.f;
        if (co < Cout && gamma) b = beta[co] - mean[co] * (gamma[co] / sqrtf(var[co] + eps));
        bias[co] = b;
    }
}

__global__ void pool3x3_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int N, int IH, int IW, int C, int OH, int OW, int stride,
                               int pad, int avg, int out_cs, int out_co) {
    const int C8 = C / 8;
    const long total = (long)N * OH * OW * C8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % C8);
        long r = idx / C8;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int n = (int)(r / OH);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = avg ? 0.f : -INFINITY;
        int cnt = 0;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= IH) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= IW) continue;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + (((size_t)n * IH + iy) * IW + ix) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = avg ? v[e] + (float)a[e] : fmaxf(v[e], (float)a[e]);
                ++cnt;
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(avg ? v[e] / (float)cnt : v[e]);
        *reinterpret_cast<bf16x8*>(out + (((size_t)n * OH + oy) * OW + ox) * out_cs + out_co + c8 * 8) = o;
    }
}

__global__ void global_avgpool_kernel(const bf16* __restrict__ x, float* __restrict__ out, int N, int HW, int C) {
    const long total = (long)N * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C), n = (int)(idx / C);
        float acc = 0.f;
        for (int p = 0; p < HW; ++p) acc += (float)x[((size_t)n * HW + p) * C + c];
        out[idx] = acc / (float)HW;
    }
}

__global__ void resize_norm_kernel(const float* __restrict__ x, bf16* __restrict__ out, int N, int IH, int IW, int OH, int OW, int normalize) {
    const long total = (long)N * OH * OW;
    const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW);
        long r = idx / OW;
        const int oy = (int)(r % OH), n = (int)(r / OH);
        // torch's area_pixel_compute_source_index (align_corners = False): src = (dst + 0.5) * scale - 0.5, clamped at 0
        float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        bf16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = (bf16)0.f; hi[e] = (bf16)0.f; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = x + ((size_t)n * 3 + c) * IH * IW;
            float v = (1.f - ly) * ((1.f - lx) * pl[(size_t)y0 * IW + x0] + lx * pl[(size_t)y0 * IW + x1]) +
                      ly * ((1.f - lx) * pl[(size_t)y1 * IW + x0] + lx * pl[(size_t)y1 * IW + x1]);
            if (normalize) v = 2.f * v - 1.f;
            lo[c] = (bf16)v;
        }
        bf16* o = out + (size_t)idx * 16;
        *reinterpret_cast<bf16x8*>(o) = lo;
        *reinterpret_cast<bf16x8*>(o + 8) = hi;
    }
}

inline unsigned grid1d(long total, int block) {
    long g = (total + block - 1) / block;
    return (unsigned)(g > 65535L * 16 ? 65535L * 16 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int64_t dxmi_gconv_packed_elems(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW) {
    const int64_t CoutP = (Cout + 31) / 32 * 32, CinP = (Cin + 15) / 16 * 16;
    return CoutP * KH * KW * CinP;
}

extern "C" int dxmi_gconv_pack(const float* w, const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                               float bn_eps, void* w_packed, float* bias, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, void* stream) {
    DXMI_CHECK_ARG(w && w_packed && bias && Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "dxmi_gconv_pack: bad arguments");
    DXMI_CHECK_ARG(!bn_gamma || (bn_beta && bn_mean && bn_var), "dxmi_gconv_pack: BatchNorm needs gamma, beta, mean and var");
    const int CoutP = (Cout + 31) / 32 * 32, CinP = (Cin + 15) / 16 * 16;
    hipLaunchKernelGGL(pack_gconv_kernel, dim3(grid1d((long)CoutP * KH * KW * CinP, 256)), dim3(256), 0, (hipStream_t)stream, w, bn_gamma,
                       bn_beta, bn_mean, bn_var, bn_eps, (bf16*)w_packed, bias, Cout, Cin, KH, KW, CoutP, CinP);
    DXMI_CHECK_LAUNCH("dxmi_gconv_pack");
    return DXMI_OK;
}

extern "C" int dxmi_gconv_fwd(const void* x, const void* w_packed, const float* bias, void* out, int32_t N, int32_t IH, int32_t IW,
                              int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t SH, int32_t SW, int32_t PH, int32_t PW,
                              int32_t out_cstride, int32_t out_coff, int32_t relu, void* stream) {
    DXMI_CHECK_ARG(x && w_packed && bias && out, "dxmi_gconv_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && Cin > 0 && Cin % 16 == 0 && Cout > 0 && KH > 0 && KW > 0 && SH > 0 && SW > 0 && PH >= 0 && PW >= 0,
                   "dxmi_gconv_fwd: bad shape (N %d, in %dx%dx%d [Cin %% 16], Cout %d, k %dx%d, s %dx%d, p %dx%d)", N, IH, IW, Cin, Cout, KH, KW, SH, SW, PH, PW);
    GConvArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w_packed; a.bias = bias; a.out = (bf16*)out;
    a.N = N; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.CoutP = (Cout + 31) / 32 * 32;
    a.KH = KH; a.KW = KW; a.SH = SH; a.SW = SW; a.PH = PH; a.PW = PW;
    a.OH = (IH + 2 * PH - KH) / SH + 1; a.OW = (IW + 2 * PW - KW) / SW + 1;
    // (the window must fit the padded map: a negative numerator truncates towards zero and would pass for a 1 x 1 output)
    DXMI_CHECK_ARG(IH + 2 * PH >= KH && IW + 2 * PW >= KW && a.OH > 0 && a.OW > 0, "dxmi_gconv_fwd: the %dx%d kernel does not fit the padded %dx%d map", KH, KW, IH + 2 * PH, IW + 2 * PW);
    DXMI_CHECK_ARG(out_cstride >= out_coff + Cout && out_coff >= 0, "dxmi_gconv_fwd: channel window [%d, %d) outside the output's %d channels", out_coff, out_coff + Cout, out_cstride);
    a.out_cs = out_cstride; a.out_co = out_coff; a.relu = relu;
    const long P = (long)N * a.OH * a.OW;
    DXMI_CHECK_ARG((P + 63) / 64 < (1L << 31), "dxmi_gconv_fwd: too many pixels");
    hipLaunchKernelGGL(generic_conv_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)((a.CoutP + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    DXMI_CHECK_LAUNCH("dxmi_gconv_fwd");
    return DXMI_OK;
}

extern "C" int dxmi_pool3x3(const void* x, void* out, int32_t N, int32_t IH, int32_t IW, int32_t C, int32_t stride, int32_t pad,
                            int32_t avg_exclude_pad, int32_t out_cstride, int32_t out_coff, void* stream) {
    DXMI_CHECK_ARG(x && out && N > 0 && IH > 0 && IW > 0 && C > 0 && C % 8 == 0 && stride > 0 && pad >= 0 && pad <= 1, "dxmi_pool3x3: bad arguments");
    DXMI_CHECK_ARG(out_cstride % 8 == 0 && out_coff % 8 == 0 && out_cstride >= out_coff + C, "dxmi_pool3x3: channel window must be 8-aligned inside the output");
    const int OH = (IH + 2 * pad - 3) / stride + 1, OW = (IW + 2 * pad - 3) / stride + 1;
    DXMI_CHECK_ARG(IH + 2 * pad >= 3 && IW + 2 * pad >= 3 && OH > 0 && OW > 0, "dxmi_pool3x3: the 3x3 window does not fit the padded %dx%d map", IH + 2 * pad, IW + 2 * pad);
    hipLaunchKernelGGL(pool3x3_kernel, dim3(grid1d((long)N * OH * OW * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out,
                       N, IH, IW, C, OH, OW, stride, pad, avg_exclude_pad, out_cstride, out_coff);
    DXMI_CHECK_LAUNCH("dxmi_pool3x3");
    return DXMI_OK;
}

extern "C" int dxmi_global_avgpool(const void* x, float* out, int32_t N, int32_t HW, int32_t C, void* stream) {
    DXMI_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0, "dxmi_global_avgpool: bad arguments");
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(grid1d((long)N * C, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, out, N, HW, C);
    DXMI_CHECK_LAUNCH("dxmi_global_avgpool");
    return DXMI_OK;
}

extern "C" int dxmi_resize_bilinear_nhwc16(const float* x, void* out, int32_t N, int32_t IH, int32_t IW, int32_t OH, int32_t OW,
                                           int32_t normalize, void* stream) {
    DXMI_CHECK_ARG(x && out && N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "dxmi_resize_bilinear_nhwc16: bad arguments");
    hipLaunchKernelGGL(resize_norm_kernel, dim3(grid1d((long)N * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)out, N, IH, IW, OH, OW,
                       normalize);
    DXMI_CHECK_LAUNCH("dxmi_resize_bilinear_nhwc16");
    return DXMI_OK;
}
