// Implicit-GEMM convolution for gfx950: bf16 operands on v_mfma_f32_32x32x16_bf16, fp32
// accumulate, NHWC activations.
//
// GEMM view   D[co][pixel] = sum_{tap,ci} W[tap][co][ci] * X[pixel + tap][ci]
//   A operand = weights, pre-packed on the host side into MFMA A-fragment order so that one
//               wave-instruction loads one contiguous 1 KiB fragment straight into VGPRs
//               (weights are wave-private here, so they never go through LDS);
//   B operand = input pixels.  A workgroup owns a 256-pixel output tile and stages the
//               tile's input halo for CK input channels into LDS ONCE per channel chunk; the
//               ksize*ksize taps then read shifted windows of that one LDS image (9x reuse of
//               every staged byte for 3x3) with conflict-free ds_read_b128 (row pitch
//               CK*2+16 B: an odd number of 16-B slots).
// Fused into staging: zero padding (incl. the DDPM asymmetric (0,1,0,1) stride-2 pad),
// nearest x2 upsampling, channel concat of two sources, fp32-NCHW -> K=27 im2col for the
// 3-channel image convs.  Fused into the epilogue: bias, per-(n,co) temb term, residual
// add, activation, bf16 pack (NHWC) or fp32 NCHW store.
//
// Reference call sites: see include/dxmi_hip.h (dxmi_conv2d_fwd).
#include "conv_common.h"

namespace {

template <int MB, int NB, int CK, int PMAX>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int WAVES_PX = 8 / NB;
    constexpr int WAVES_CO = 4 / WAVES_PX;
    constexpr int BN = 32 * MB * WAVES_CO;
    constexpr int ROWB = CK * 2 + 16;
    constexpr int KSTEPS = CK / 16;
    constexpr int PPP = CK / 8;  // 16-byte pieces per staged pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wco = wave % WAVES_CO;
    const int wpx = wave / WAVES_CO;

    int pt, cot;
    if (!conv_block_to_tile(p, blockIdx.x, pt, cot)) return;

    const bool k27 = p.in_mode == DXMI_IN_NCHW_F32_K27;
    const bool rows = p.in_mode == DXMI_IN_ROWS_F32;
    const int TW = 1 << p.TWl, TH = 1 << p.THl;
    int n0 = 0, oy0 = 0, ox0 = 0;
    if (!rows) {
        const int txn = p.OW >> p.TWl, tyn = p.OH >> p.THl;
        const int tx = pt % txn;
        const int ty = (pt / txn) % tyn;
        n0 = (pt / (txn * tyn)) * p.SUBS;
        oy0 = ty << p.THl;
        ox0 = tx << p.TWl;
    }
    const int S = p.stride;
    const int Cin = p.C0 + p.C1;
    const int nchunks = k27 ? 1 : Cin / CK;
    const int ntaps = (k27 || rows) ? 1 : p.ksize * p.ksize;
    const int HHW = p.HH * p.HWd;
    const int HP = p.SUBS * HHW;
    const int npieces = HP * PPP;

    // ---- per-lane LDS base offsets of the B fragments (pixel = lane&31, k-half = lane>>5)
    int hoff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int pix = (wpx * NB + nb) * 32 + (lane & 31);
        const int x = pix & (TW - 1);
        const int y = (pix >> p.TWl) & (TH - 1);
        const int sub = pix >> (p.TWl + p.THl);
        int hp;
        if (k27 || rows) hp = pix;
        else hp = (sub * p.HH + y * S) * p.HWd + x * S;
        hoff[nb] = hp * ROWB + (lane >> 5) * 16;
    }

    // ---- source pixel of every staging piece this thread owns (same for all chunks)
    auto piece_src = [&](int i) -> int {
        const int hp = i / PPP;
        const int sub = hp / HHW;
        const int rem = hp - sub * HHW;
        const int hy = rem / p.HWd;
        const int hx = rem - hy * p.HWd;
        const int iy = oy0 * S - p.pad + hy;
        const int ix = ox0 * S - p.pad + hx;
        const int n = n0 + sub;
        const int sh = p.ups ? 1 : 0;  // ups 1: nearest x2 ; ups 2: zero-stuffed x2 (stride-2 data gradient)
        const bool ok = (iy >= 0) && (ix >= 0) && (iy < (p.IH << sh)) && (ix < (p.IW << sh)) && (n < p.N) &&
                        (p.ups != 2 || (((iy | ix) & 1) == 0));
        return ok ? ((n * p.IH + (iy >> sh)) * p.IW + (ix >> sh)) : -1;
    };
    int srcpix[PMAX > 0 ? PMAX : 1];
    if (PMAX > 0 && p.in_mode == DXMI_IN_NHWC_BF16) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int i = tid + q * 256;
            srcpix[q] = (i < npieces) ? piece_src(i) : -1;
        }
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

    const int cb0 = cot * (BN / 32) + wco * MB;  // first 32-co block of this wave
    const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(p.w);

    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // everyone is done reading the previous chunk's LDS image
        if (rows) {
            // fp32 row-major [P][Cin] -> bf16, optional pre-activation (swish on the temb path)
            const float* xin = reinterpret_cast<const float*>(p.in0);
            for (int i = tid; i < 256 * PPP; i += 256) {
                const int r = i / PPP, pc = i % PPP;
                const long grow = (long)pt * 256 + r;
                bf16x8 v;
                if (grow < p.P) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(xin + grow * Cin + c * CK + pc * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(xin + grow * Cin + c * CK + pc * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (bf16)dxmi_act(lo[e], p.pre_act);
                        v[4 + e] = (bf16)dxmi_act(hi[e], p.pre_act);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
                }
                *reinterpret_cast<bf16x8*>(smem + r * ROWB + pc * 16) = v;
            }
        } else if (k27) {
            // one thread per output pixel: gather the 27 taps (k = ci*9 + ky*3 + kx)
            const float* xin = reinterpret_cast<const float*>(p.in0);
            const int pix = tid;
            const int x = pix & (TW - 1);
            const int y = (pix >> p.TWl) & (TH - 1);
            const int sub = pix >> (p.TWl + p.THl);
            const int n = n0 + sub, oy = oy0 + y, ox = ox0 + x;
            bf16 v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                float f = 0.f;
                if (k < 27) {
                    const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
                    const int iy = oy + ky - 1, ix = ox + kx - 1;
                    if (n < p.N && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW)
                        f = xin[((n * 3 + ci) * p.IH + iy) * p.IW + ix];
                }
                v[k] = (bf16)f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk[e] = v[q * 8 + e];
                *reinterpret_cast<bf16x8*>(smem + pix * ROWB + q * 16) = pk;
            }
        } else {
            const int cbase = c * CK;
            const bool first = cbase < p.C0;
            const bf16* src = first ? p.in0 : p.in1;
            const int Cs = first ? p.C0 : p.C1;
            const int coff = first ? cbase : cbase - p.C0;
            if (PMAX > 0) {
                bf16x8 v[PMAX > 0 ? PMAX : 1];
#pragma unroll
                for (int q = 0; q < PMAX; ++q) {
                    const int i = tid + q * 256;
                    const int pc = i % PPP;
                    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    v[q] = z;
                    if (srcpix[q] >= 0)
                        v[q] = *reinterpret_cast<const bf16x8*>(src + (size_t)srcpix[q] * Cs + coff + pc * 8);
                }
#pragma unroll
                for (int q = 0; q < PMAX; ++q) {
                    const int i = tid + q * 256;
                    if (i < npieces)
                        *reinterpret_cast<bf16x8*>(smem + (i / PPP) * ROWB + (i % PPP) * 16) = v[q];
                }
            } else {
                for (int i = tid; i < npieces; i += 256) {
                    const int sp = piece_src(i);
                    const int pc = i % PPP;
                    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (sp >= 0) v = *reinterpret_cast<const bf16x8*>(src + (size_t)sp * Cs + coff + pc * 8);
                    *reinterpret_cast<bf16x8*>(smem + (i / PPP) * ROWB + pc * 16) = v;
                }
            }
        }
        __syncthreads();

        for (int tap = 0; tap < ntaps; ++tap) {
            const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
            const int toff = (k27 || rows) ? 0 : (ky * p.HWd + kx) * ROWB;
            bf16x8 a[MB][KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    a[mb][ks] = wfrag[((size_t)(tap * p.KST + c * KSTEPS + ks) * p.CB + cb0 + mb) * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8 b[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    b[nb] = *reinterpret_cast<const bf16x8*>(smem + hoff[nb] + toff + ks * 32);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mb][ks], b[nb], acc[mb][nb], 0, 0, 0);
            }
        }
    }

    conv_epilogue<MB, NB, BN == 128>(p, acc, pt, n0, oy0, ox0, wpx * NB, cb0, lane);
}

// ---- weight packing: OIHW fp32 -> [tap][kstep][cb][lane][8] bf16
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, bf16* __restrict__ dst, int Cout, int Cin,
                                        int ks, int flip, int k27, int CB, int KST, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int jj = idx & 7;
    const int lane = (idx >> 3) & 63;
    long r = idx >> 9;
    const int cb = r % CB; r /= CB;
    const int kst = r % KST; r /= KST;
    const int tap = (int)r;
    const int co = cb * 32 + (lane & 31);
    const int k = kst * 16 + 8 * (lane >> 5) + jj;
    float v = 0.f;
    if (k27) {
        // logical GEMM: Cout x 27(+5 zero); weight OIHW flattened over (ci,ky,kx)
        if (co < Cout && k < 27) v = w[(long)co * 27 + k];
    } else if (!flip) {
        if (co < Cout && k < Cin) {
            const int ky = tap / ks, kx = tap % ks;
            v = w[(((long)co * Cin + k) * ks + ky) * ks + kx];
        }
    } else {
        // data-gradient operator: output channels = original Cin (= "Cout" arg here is the
        // logical output count), taps flipped, in/out swapped.  w is [Cin_logical_out? ...]
        // Here: logical Cout = original Cin, logical Cin = original Cout, w is the ORIGINAL
        // OIHW tensor [origCout=Cin][origCin=Cout][ks][ks].
        if (co < Cout && k < Cin) {
            const int ky = tap / ks, kx = tap % ks;
            v = w[(((long)k * Cout + co) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
        }
    }
    dst[idx] = (bf16)v;
}

// Multi-tensor form: up to DXMI_PACK_MAX weights per launch (descriptors by value in the kernel arguments, blockIdx.y = item):
// after an optimiser step every weight of a net is re-packed — one launch per layer was ~330 launches per EDM policy
// iteration (30 ms of a 640 ms train step), now a handful.
struct PackItems {
    dxmi_pack_item it[DXMI_PACK_MAX];
};
// One thread = one fragment lane (cout co, eight consecutive input channels k0 .. k0 + 7) of ALL taps: in the OIHW layout those
// 8 x taps floats are one contiguous run (288 B at 3x3), read as float4s, and every tap's bf16x8 is one 16-byte store that the
// wave writes as a whole 1-KiB fragment.  (The first version moved one bf16 per thread: 4-byte loads 36 B apart, every 64-byte
// sector of a 3x3 weight fetched once per tap — 0.7 TB/s on the 1.2 GB of fp32 weights of the ImageNet-64 net.)
template <int TAPS>
__device__ __forceinline__ void pack_unit_fast(const float* __restrict__ w, bf16* __restrict__ dst, long frag_stride, int lane) {
    float buf[8 * TAPS];
    const float4* src = reinterpret_cast<const float4*>(w);
#pragma unroll
    for (int i = 0; i < 2 * TAPS; ++i) {
        const float4 t = src[i];
        buf[4 * i] = t.x; buf[4 * i + 1] = t.y; buf[4 * i + 2] = t.z; buf[4 * i + 3] = t.w;
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        bf16x8 o;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16)buf[jj * TAPS + tap];
        *reinterpret_cast<bf16x8*>(dst + tap * frag_stride + lane * 8) = o;
    }
}

// The data-gradient pack (transpose_flip: logical cout co = original Cin index, logical cin k = original Cout index, taps flipped)
// of the same unit: source run (k0 + jj, co) is TAPS contiguous floats, the runs of the 32 couts of a half wave are adjacent (a half
// wave reads 32 x TAPS x 4 contiguous bytes per jj), consecutive jj are Cout x TAPS floats apart.  All 8 x TAPS loads are issued
// before the first conversion (round 4: this form had kept the one-float-per-load path with a tap loop around it and was ~3x the
// time of the forward pack on the ADM nets: 1.35 ms against 0.4 ms per optimiser step).
struct PackF3 {
    float a, b, c;
};
template <int TAPS>
__device__ __forceinline__ void pack_unit_fast_t(const float* __restrict__ w, long kstride, bf16* __restrict__ dst, long frag_stride, int lane) {
    float buf[8][TAPS];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const float* src = w + jj * kstride;
        if constexpr (TAPS == 9) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const PackF3 t = reinterpret_cast<const PackF3*>(src)[i];
                buf[jj][3 * i] = t.a; buf[jj][3 * i + 1] = t.b; buf[jj][3 * i + 2] = t.c;
            }
        } else {
            buf[jj][0] = src[0];
        }
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        bf16x8 o;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) o[jj] = (bf16)buf[jj][TAPS - 1 - tap];
        *reinterpret_cast<bf16x8*>(dst + tap * frag_stride + lane * 8) = o;
    }
}

__global__ void pack_conv_weights_kernel(PackItems P) {
    const dxmi_pack_item d = P.it[blockIdx.y];
    const int CB = (d.Cout + 31) / 32;
    const int KST = d.k27 ? 2 : (d.Cin + 15) / 16;
    const int taps = d.k27 ? 1 : d.ksize * d.ksize;
    const float* __restrict__ w = d.w;
    bf16* __restrict__ dst = reinterpret_cast<bf16*>(d.dst);
    const int ks = d.ksize, Cout = d.Cout, Cin = d.Cin;
    const long units = (long)KST * CB * 64;                 // fragment lanes of one tap
    const long frag_stride = (long)KST * CB * 512;          // elements between the same fragment of consecutive taps
    const bool can_fast = !d.k27 && !d.transpose_flip && (taps == 9 || taps == 1) && Cin % 8 == 0;
    for (long unit = (long)blockIdx.x * blockDim.x + threadIdx.x; unit < units; unit += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(unit & 63);
        long r = unit >> 6;
        const int cb = (int)(r % CB);
        const int kst = (int)(r / CB);
        const int co = cb * 32 + (lane & 31);
        const int k0 = kst * 16 + 8 * (lane >> 5);
        bf16* const fd = dst + ((long)kst * CB + cb) * 512;
        if (can_fast && co < Cout && k0 + 8 <= Cin) {
            const float* src = w + ((long)co * Cin + k0) * taps;
            if (taps == 9) pack_unit_fast<9>(src, fd, frag_stride, lane);
            else pack_unit_fast<1>(src, fd, frag_stride, lane);
            continue;
        }
        if (!d.k27 && d.transpose_flip && (taps == 9 || taps == 1) && co < Cout && k0 + 8 <= Cin) {
            const float* src = w + ((long)k0 * Cout + co) * taps;
            if (taps == 9) pack_unit_fast_t<9>(src, (long)Cout * 9, fd, frag_stride, lane);
            else pack_unit_fast_t<1>(src, (long)Cout, fd, frag_stride, lane);
            continue;
        }
        for (int tap = 0; tap < taps; ++tap) {
            bf16x8 o;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int k = k0 + jj;
                float v = 0.f;
                if (d.k27) {
                    if (co < Cout && k < 27) v = w[(long)co * 27 + k];
                } else if (co < Cout && k < Cin) {
                    const int ky = tap / ks, kx = tap % ks;
                    v = d.transpose_flip ? w[(((long)k * Cout + co) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)]
                                         : w[(((long)co * Cin + k) * ks + ky) * ks + kx];
                }
                o[jj] = (bf16)v;
            }
            *reinterpret_cast<bf16x8*>(fd + tap * frag_stride + lane * 8) = o;
        }
    }
}

template <int MB, int NB, int CK, int PMAX>
int launch_conv(const ConvArgs& a, size_t lds, hipStream_t st) {
    auto kern = conv_igemm_kernel<MB, NB, CK, PMAX>;
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int PT8 = (a.PT + 7) / 8 * 8;
    dim3 grid(PT8 * a.CT), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
    DXMI_CHECK_LAUNCH("dxmi_conv2d_fwd");
    return DXMI_OK;
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

extern "C" int64_t dxmi_packed_conv_weight_bytes(int32_t Cout, int32_t Cin, int32_t ksize, int32_t k27) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return 0;      // dxmi_pack_conv_weight rejects these
    const int64_t CB = (Cout + 31) / 32;
    if (k27) return CB * 2 * 64 * 8 * 2;
    const int64_t KST = (Cin + 15) / 16;
    return (int64_t)ksize * ksize * KST * CB * 64 * 8 * 2;
}

extern "C" int dxmi_pack_conv_weight(const float* w, void* dst, int32_t Cout, int32_t Cin, int32_t ksize,
                                     int32_t transpose_flip, int32_t k27, void* stream) {
    DXMI_CHECK_ARG(w && dst, "dxmi_pack_conv_weight: null pointer");
    DXMI_CHECK_ARG(ksize == 1 || ksize == 3, "dxmi_pack_conv_weight: ksize must be 1 or 3");
    DXMI_CHECK_ARG(Cout > 0 && Cin > 0, "dxmi_pack_conv_weight: Cout %d / Cin %d must be positive", Cout, Cin);
    DXMI_CHECK_ARG(!k27 || (Cin == 3 && ksize == 3 && !transpose_flip), "dxmi_pack_conv_weight: k27 needs Cin=3,k=3");
    const int CB = (Cout + 31) / 32;
    const int KST = k27 ? 2 : (Cin + 15) / 16;
    const int taps = k27 ? 1 : ksize * ksize;
    const long total = (long)taps * KST * CB * 512;
    const int threads = 256;
    const long blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, w,
                       reinterpret_cast<bf16*>(dst), Cout, Cin, ksize, transpose_flip, k27, CB, KST, total);
    DXMI_CHECK_LAUNCH("dxmi_pack_conv_weight");
    return DXMI_OK;
}

extern "C" int dxmi_pack_conv_weights(const dxmi_pack_item* items, int32_t count, void* stream) {
    DXMI_CHECK_ARG(items && count > 0, "dxmi_pack_conv_weights: empty list");
    for (int base = 0; base < count; base += DXMI_PACK_MAX) {
        PackItems P;
        const int n = count - base < DXMI_PACK_MAX ? count - base : DXMI_PACK_MAX;
        long maxtotal = 0;
        for (int i = 0; i < n; ++i) {
            const dxmi_pack_item& d = items[base + i];
            DXMI_CHECK_ARG(d.w && d.dst && d.Cout > 0 && d.Cin > 0 && (d.ksize == 1 || d.ksize == 3) && (!d.k27 || (d.Cin == 3 && d.ksize == 3 && !d.transpose_flip)),
                           "dxmi_pack_conv_weights: bad item %d", base + i);
            P.it[i] = d;
            const long total = (long)(d.k27 ? 2 : (d.Cin + 15) / 16) * ((d.Cout + 31) / 32) * 64;       // fragment lanes (threads) of the item
            if (total > maxtotal) maxtotal = total;
        }
        long bx = (maxtotal + 255) / 256;
        if (bx > 512) bx = 512;              // grid-stride over the large items
        hipLaunchKernelGGL(pack_conv_weights_kernel, dim3((unsigned)bx, n), dim3(256), 0, (hipStream_t)stream, P);
        DXMI_CHECK_LAUNCH("dxmi_pack_conv_weights");
    }
    return DXMI_OK;
}

namespace {

int dispatch_conv(ConvArgs& a, int variant, hipStream_t st, int* kernel_id = nullptr) {
    if (variant == 0) {
        const int rc = conv_pipe_try_launch(a, st, kernel_id);
        if (rc <= 0) return rc;  // launched (or failed loudly); rc == 1: not eligible
    }
    DXMI_CHECK_ARG(!a.gn_stats || kernel_id, "dxmi_conv2d_fwd: the kernel for this shape does not emit GroupNorm block statistics "
                                             "(dxmi_conv2d_gn_stats_partials returns 0 for it)");
    DXMI_CHECK_ARG(!a.gn_out || kernel_id, "dxmi_conv2d_fwd: the kernel for this shape cannot fuse the GroupNorm of its output "
                                           "(dxmi_conv2d_gn_fuse_supported returns 0 for it)");
    constexpr int CK = 32;
    a.tile_px = 256;
    a.stagger = 0;
    const bool flat = a.in_mode != DXMI_IN_NHWC_BF16;
    const int HP = flat ? 256 : a.SUBS * a.HH * a.HWd;
    const size_t lds = (size_t)HP * (CK * 2 + 16);
    a.lds_buf = (int)lds;
    DXMI_CHECK_ARG(lds <= 160 * 1024, "dxmi_conv2d_fwd: LDS image %zu too large", lds);
    const int npieces = HP * (CK / 8);
    // staging pieces per thread cached in registers: 6 covers the 32x8 / 16x16 tiles, 9 the
    // multi-image 8x8 / 4x4 tiles; larger halos (stride 2) recompute their addresses per chunk.
    const int pmax = npieces <= 6 * 256 ? 6 : (npieces <= 9 * 256 ? 9 : 0);
    // Tiling variants: cout tile BN = 128 when Cout allows it, else 32 (conv_out, heads).
    if (kernel_id) {
        // id = MB*1000 + NB*100 + (small_halo ? 6 : 0): names the template instantiation that runs
        if (a.Cout % 128 == 0 && a.out_mode != DXMI_OUT_NCHW_F32) *kernel_id = (variant == 1 ? 2400 : 1800) + pmax;
        else *kernel_id = 1200 + pmax;
        return DXMI_OK;
    }
    if (a.Cout % 128 == 0 && a.out_mode != DXMI_OUT_NCHW_F32) {
        a.CT = a.Cout / 128;
#define DXMI_CONV_PM(MB_, NB_)                                           \
    (pmax == 6 ? launch_conv<MB_, NB_, CK, 6>(a, lds, st)                \
               : pmax == 9 ? launch_conv<MB_, NB_, CK, 9>(a, lds, st) : launch_conv<MB_, NB_, CK, 0>(a, lds, st))
        if (variant == 1) return DXMI_CONV_PM(2, 4);
        return DXMI_CONV_PM(1, 8);
    }
    a.CT = a.CB;  // one 32-co block per workgroup, 4 waves split the 256 pixels
    return DXMI_CONV_PM(1, 2);
#undef DXMI_CONV_PM
}

}  // namespace

static int conv2d_impl(const dxmi_conv_desc* d, void* stream, int* kernel_id) {
    DXMI_CHECK_ARG(d && d->in0 && d->wpacked && d->out, "dxmi_conv2d_fwd: null pointer");
    const bool k27 = d->in_mode == DXMI_IN_NCHW_F32_K27;
    DXMI_CHECK_ARG(d->in_mode == DXMI_IN_NHWC_BF16 || k27, "dxmi_conv2d_fwd: in_mode %d unsupported", d->in_mode);
    DXMI_CHECK_ARG(d->out_mode == DXMI_OUT_NHWC_BF16 || d->out_mode == DXMI_OUT_NCHW_F32, "dxmi_conv2d_fwd: out_mode %d unsupported", d->out_mode);
    const int Cin = d->C0 + d->C1;
    DXMI_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "dxmi_conv2d_fwd: ksize %d unsupported", d->ksize);
    DXMI_CHECK_ARG(d->stride == 1 || d->stride == 2, "dxmi_conv2d_fwd: stride %d unsupported", d->stride);
    DXMI_CHECK_ARG(d->N > 0 && d->OH > 0 && d->OW > 0 && d->Cout > 0 && d->IH > 0 && d->IW > 0 && d->C0 > 0 && d->C1 >= 0,
                   "dxmi_conv2d_fwd: empty or negative shape (N %d, in %dx%dx(%d+%d), out %dx%dx%d)", d->N, d->IH, d->IW, d->C0, d->C1, d->OH, d->OW, d->Cout);
    DXMI_CHECK_ARG(d->act >= DXMI_ACT_NONE && d->act <= DXMI_ACT_SILU, "dxmi_conv2d_fwd: act %d unknown", d->act);
    DXMI_CHECK_ARG(!d->addvec || d->addvec_ld == 0 || d->addvec_ld >= d->Cout,      // 0: one row for the whole batch (Model.temb_table)
                   "dxmi_conv2d_fwd: addvec_ld %d is neither 0 (one shared row) nor >= Cout %d", d->addvec_ld, d->Cout);
    DXMI_CHECK_ARG((d->OW & (d->OW - 1)) == 0 && (d->OH & (d->OH - 1)) == 0 && d->OW >= 4 && d->OH >= 4,
                   "dxmi_conv2d_fwd: OH/OW must be powers of two >= 4 (got %dx%d)", d->OH, d->OW);
    if (k27) {
        DXMI_CHECK_ARG(d->C0 == 3 && d->C1 == 0 && d->ksize == 3 && d->stride == 1 && d->pad == 1 && !d->upsample &&
                           d->IH == d->OH && d->IW == d->OW,
                       "dxmi_conv2d_fwd: K27 mode needs a 3-channel 3x3/s1/p1 conv");
    } else {
        DXMI_CHECK_ARG(Cin % 32 == 0 && d->C0 % 32 == 0, "dxmi_conv2d_fwd: Cin (%d+%d) must be a multiple of 32", d->C0, d->C1);
        DXMI_CHECK_ARG(d->C1 == 0 || d->in1, "dxmi_conv2d_fwd: C1>0 but in1 is NULL");
        DXMI_CHECK_ARG(d->upsample >= 0 && d->upsample <= 2, "dxmi_conv2d_fwd: upsample must be 0, 1 (nearest) or 2 (zero-stuffed)");
        const int VIH = d->IH << (d->upsample ? 1 : 0), VIW = d->IW << (d->upsample ? 1 : 0);
        DXMI_CHECK_ARG((d->OH - 1) * d->stride - d->pad < VIH && (d->OW - 1) * d->stride - d->pad < VIW && d->pad >= 0 && d->pad < d->ksize,
                       "dxmi_conv2d_fwd: output %dx%d inconsistent with input %dx%d", d->OH, d->OW, VIH, VIW);
    }
    DXMI_CHECK_ARG(d->out_mode != DXMI_OUT_NCHW_F32 || !d->residual, "dxmi_conv2d_fwd: residual unsupported with NCHW output");
    DXMI_CHECK_ARG(!d->residual || d->Cout % 4 == 0, "dxmi_conv2d_fwd: residual needs Cout%%4==0");
    DXMI_CHECK_ARG(!d->mask_src || (d->Cout % 4 == 0 && d->out_mode == DXMI_OUT_NHWC_BF16), "dxmi_conv2d_fwd: mask_src needs NHWC bf16 output, Cout%%4==0");

    ConvArgs a;
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.w = (const bf16*)d->wpacked;
    a.bias = d->bias; a.addvec = d->addvec; a.residual = (const bf16*)d->residual; a.out = d->out;
    a.mask_src = (const bf16*)d->mask_src; a.mask_slope = d->mask_slope; a.gn_stats = d->gn_stats;
    a.gn_out = (bf16*)d->gn_out; a.gn_gamma = d->gn_gamma; a.gn_beta = d->gn_beta; a.gn_eps = d->gn_eps; a.gn_flags = d->gn_flags;
    a.res_is_mask = 0;
    DXMI_CHECK_ARG(!d->gn_out || (d->gn_gamma && d->gn_beta && d->gn_groups > 0 && d->Cout == 8 * d->gn_groups && d->variant == 0 &&
                                  d->out_mode == DXMI_OUT_NHWC_BF16),
                   "dxmi_conv2d_fwd: gn_out needs gamma / beta, 8 channels per group, the default variant and NHWC bf16 output");
    DXMI_CHECK_ARG(!d->gn_stats || (d->variant == 0 && d->out_mode == DXMI_OUT_NHWC_BF16 && d->Cout % 4 == 0),
                   "dxmi_conv2d_fwd: gn_stats needs the default variant, NHWC bf16 output and Cout%%4==0");
    a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.C0 = d->C0; a.C1 = d->C1; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
    a.ksize = d->ksize; a.stride = d->stride; a.pad = d->pad; a.ups = d->upsample; a.act = d->act;
    a.addvec_ld = d->addvec_ld; a.in_mode = d->in_mode; a.out_mode = d->out_mode; a.P = 0; a.pre_act = 0;
    const int TW = d->OW < 32 ? d->OW : 32;
    int TH = 256 / TW; if (TH > d->OH) TH = d->OH;
    a.TWl = ilog2(TW); a.THl = ilog2(TH); a.SUBS = 256 / (TW * TH);
    if (k27) { a.HH = TH; a.HWd = TW; }
    else { a.HH = (TH - 1) * d->stride + d->ksize; a.HWd = (TW - 1) * d->stride + d->ksize; }
    const int ngroups = (d->N + a.SUBS - 1) / a.SUBS;
    a.PT = ngroups * (d->OH / TH) * (d->OW / TW);
    a.CB = (d->Cout + 31) / 32;
    a.KST = k27 ? 2 : Cin / 16;
    return dispatch_conv(a, d->variant, (hipStream_t)stream, kernel_id);
}

extern "C" int dxmi_conv2d_fwd(const dxmi_conv_desc* d, void* stream) { return conv2d_impl(d, stream, nullptr); }

// Partials per image of the GroupNorm block statistics the selected kernel writes (0: it writes none).
int conv_pipe_stats_tile(int id, int OH, int OW, int Cout);     // conv_pipe.hip

static int gn_stats_partials_of(const dxmi_conv_desc* d, int id) {
    if (id >= 400000 && id < 400100) {            // conv_ws_kernel<TW>: (pixel tile, pixel half) of the image
        const int TW = id - 400000, TH = 256 / TW;
        return (d->OH / TH) * (d->OW / TW) * 2;
    }
    const int tile = conv_pipe_stats_tile(id, d->OH, d->OW, d->Cout);     // stem / conv_pipe / 1x1 stream kernels: one partial per tile
    return tile ? d->OH * d->OW / tile : 0;
}

extern "C" int dxmi_conv2d_gn_stats_partials(const dxmi_conv_desc* d) {
    int id = 0;
    if (!d) return 0;
    dxmi_conv_desc q = *d;
    q.gn_stats = nullptr;
    if (q.variant != 0 || q.out_mode != DXMI_OUT_NHWC_BF16) return 0;
    const int rc = conv2d_impl(&q, nullptr, &id);
    return rc == DXMI_OK ? gn_stats_partials_of(d, id) : 0;
}

extern "C" int dxmi_conv2d_gn_fuse_supported(const dxmi_conv_desc* d) {
    int id = 0;
    if (!d) return 0;
    dxmi_conv_desc q = *d;
    q.gn_stats = nullptr;
    q.gn_out = nullptr;
    if (q.variant != 0 || q.out_mode != DXMI_OUT_NHWC_BF16 || d->gn_groups <= 0 || d->Cout != 8 * d->gn_groups) return 0;
    const int rc = conv2d_impl(&q, nullptr, &id);
    if (rc != DXMI_OK) return 0;
    if (id == 450432) return 1;                        // conv_sm_kernel<2, 8, 32>: eight whole 4x4 images x 32 couts per tile
    return id == 400008 && (d->gn_flags & 2) ? 1 : 0;  // conv_ws8_kernel: a whole 8x8 image per wave, instead of the raw output
}

// Which template instantiation dxmi_conv2d_fwd would launch for this descriptor (no launch):
// MB*1000 + NB*100 + PMAX, e.g. 1806 = conv_igemm_kernel<1,8,32,6>.  Used by bench.py to attribute
// HIP-event timings to the kernel named in the rocprof summary.
extern "C" int dxmi_conv2d_kernel_id(const dxmi_conv_desc* d) {
    int id = 0;
    const int rc = conv2d_impl(d, nullptr, &id);
    return rc == DXMI_OK ? id : rc;
}

// Small dense layers (timestep-embedding MLP, temb_proj / emb_layers: P = batch rows, K <= 2048): the conv kernel gives
// such a GEMM 1-22 workgroups walking K chunk by chunk (~100 us, pure latency).  Here every wave owns one 32-row x 32-col
// output tile, reads its weight fragments and its fp32 rows straight from global memory (no LDS, no barrier) with eight
// k16 steps of loads in flight, so the launch is P/32 x M/128 workgroups of independent MFMA chains.
namespace {
// gridDim.z > 1: split K — slice z takes k16 steps [z * ksplit, (z + 1) * ksplit) and writes its partial product to out + z * P * M
// (no bias, no activation: the caller adds the slices in slice order).  A skinny product with a long K (the data gradient of the ADM
// nets' concatenated emb_layers: 16 rows x K = 30 k x 768 columns) was 24 waves walking 1 900 dependent k steps each.
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const bf16* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int P, int K,
                                                          int M, int CB, int pre_act, int post_act, int ksplit) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb = blockIdx.y * 4 + wave;
    if (cb >= CB) return;
    out += (size_t)blockIdx.z * P * M;
    const int row = blockIdx.x * 32 + (lane & 31);
    const int h = lane >> 5;
    const bool rvalid = row < P;
    const float* xr = x + (size_t)(rvalid ? row : 0) * K + 8 * h;
    const bf16x8* wf = reinterpret_cast<const bf16x8*>(w) + (size_t)cb * 64 + lane;
    const size_t wstep = (size_t)CB * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kbeg = blockIdx.z * ksplit;
    const int ksteps = kbeg + ksplit < K / 16 ? kbeg + ksplit : K / 16;
    constexpr int LU = 8;           // k16 steps of loads in flight (4: K = 768 was twelve dependent round trips per wave)
    for (int k0 = kbeg; k0 < ksteps; k0 += LU) {
        bf16x8 a[LU];
        f32x4 lo[LU], hi[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int ks = k0 + u < ksteps ? k0 + u : ksteps - 1;
            a[u] = wf[ks * wstep];
            lo[u] = *reinterpret_cast<const f32x4*>(xr + ks * 16);
            hi[u] = *reinterpret_cast<const f32x4*>(xr + ks * 16 + 4);
        }
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            if (k0 + u < ksteps) {
                bf16x8 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    b[e] = (bf16)(rvalid ? dxmi_act(lo[u][e], pre_act) : 0.f);
                    b[4 + e] = (bf16)(rvalid ? dxmi_act(hi[u][e], pre_act) : 0.f);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b, acc, 0, 0, 0);
            }
        }
    }
    if (!rvalid) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co = cb * 32 + 8 * g + 4 * h;
        if (co + 3 < M) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = dxmi_act(acc[4 * g + e] + (bias ? bias[co + e] : 0.f), post_act);
            *reinterpret_cast<f32x4*>(out + (size_t)row * M + co) = o;
        } else {
            for (int e = 0; e < 4; ++e)
                if (co + e < M) out[(size_t)row * M + co + e] = dxmi_act(acc[4 * g + e] + (bias ? bias[co + e] : 0.f), post_act);
        }
    }
}
}  // namespace

// Dense layer on the same MFMA path: out[P][M] = post(pre(x[P][K]) @ W[M][K]^T + b), fp32 rows
// in/out, bf16 operands.  W packed by dxmi_pack_conv_weight(ksize=1).
extern "C" int dxmi_linear_fwd(const float* x, const void* wpacked, const float* bias, float* out, int32_t P,
                               int32_t K, int32_t M, int32_t pre_act, int32_t post_act, void* stream) {
    DXMI_CHECK_ARG(x && wpacked && out, "dxmi_linear_fwd: null pointer");
    DXMI_CHECK_ARG(P > 0 && M > 0 && K > 0 && K % 32 == 0, "dxmi_linear_fwd: K (%d) must be a positive multiple of 32", K);
    if (P <= 4096 && M % 4 == 0) {   // row counts of the embedding MLPs (batch); larger row counts keep the tiled conv path
        const int CB = (M + 31) / 32;
        hipLaunchKernelGGL(linear_small_kernel, dim3((P + 31) / 32, (CB + 3) / 4), dim3(256), 0, (hipStream_t)stream, x,
                           (const bf16*)wpacked, bias, out, P, K, M, CB, pre_act, post_act, K / 16);
        DXMI_CHECK_LAUNCH("dxmi_linear_fwd(small)");
        return DXMI_OK;
    }
    ConvArgs a;
    a.in0 = (const bf16*)x; a.in1 = nullptr; a.w = (const bf16*)wpacked; a.bias = bias; a.addvec = nullptr;
    a.residual = nullptr; a.out = out; a.mask_src = nullptr; a.mask_slope = 0.f; a.gn_stats = nullptr; a.gn_out = nullptr;
    a.gn_gamma = a.gn_beta = nullptr; a.gn_eps = 0.f; a.gn_flags = 0; a.res_is_mask = 0;
    a.N = 1; a.IH = 1; a.IW = P; a.C0 = K; a.C1 = 0; a.OH = 1; a.OW = P; a.Cout = M;
    a.ksize = 1; a.stride = 1; a.pad = 0; a.ups = 0; a.act = post_act; a.addvec_ld = 0;
    a.in_mode = DXMI_IN_ROWS_F32; a.out_mode = DXMI_OUT_ROWS_F32; a.P = P; a.pre_act = pre_act;
    a.TWl = 5; a.THl = 3; a.SUBS = 1; a.HH = 8; a.HWd = 32;
    a.PT = (P + 255) / 256; a.CB = (M + 31) / 32; a.KST = K / 16;
    return dispatch_conv(a, 0, (hipStream_t)stream);
}

// Split-K form for skinny products with a long K: partials[s][P][M] = pre(x[:, slice s]) @ W[:, slice s]^T, s < nsplit = dxmi_linear_splitk_slices(P, K, M)
// (>= 1); the caller sums the slices in order (deterministic) and adds bias / activation itself.  P <= 4096, M % 4 == 0, K % 32 == 0.
extern "C" int dxmi_linear_splitk_slices(int32_t P, int32_t K, int32_t M) {
    if (P <= 0 || K <= 0 || M <= 0 || K % 32 != 0 || P > 4096 || M % 4 != 0) return 0;
    const long waves = (long)((P + 31) / 32) * ((M + 31) / 32);          // one 32 x 32 output tile per wave
    if (K < 4096 || waves >= 1024) return 1;
    int s = (int)(2048 / waves);                                         // ~2 k waves in flight
    const int smax = K / 16 / 32;                                        // >= 32 k16 steps (four rounds of loads) per slice
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

extern "C" int dxmi_linear_splitk(const float* x, const void* wpacked, float* partials, int32_t P, int32_t K, int32_t M, int32_t pre_act,
                                  void* stream) {
    DXMI_CHECK_ARG(x && wpacked && partials, "dxmi_linear_splitk: null pointer");
    const int S = dxmi_linear_splitk_slices(P, K, M);
    DXMI_CHECK_ARG(S >= 1, "dxmi_linear_splitk: unsupported shape P=%d K=%d M=%d (P <= 4096, M %% 4 == 0, K %% 32 == 0)", P, K, M);
    const int CB = (M + 31) / 32;
    const int ksplit = ((K / 16 + S - 1) / S + 7) / 8 * 8;               // whole rounds of eight k16 steps
    hipLaunchKernelGGL(linear_small_kernel, dim3((P + 31) / 32, (CB + 3) / 4, S), dim3(256), 0, (hipStream_t)stream, x, (const bf16*)wpacked,
                       (const float*)nullptr, partials, P, K, M, CB, pre_act, DXMI_ACT_NONE, ksplit);
    DXMI_CHECK_LAUNCH("dxmi_linear_splitk");
    return DXMI_OK;
}
