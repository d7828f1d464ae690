// Attention backward for the U-Net attention blocks (training path) on gfx950.
//
// dQ, dK, dV are five small batched GEMMs per (image, head) plus one row-wise softmax-backward:
//   S  = scale * Q K^T          (k = d contiguous in both operands)
//   dP = dO V^T                 (k = d contiguous in both)
//   P = softmax(S) ; dS = P o (dP - rowsum(dP o P))        [dxmi_softmax_bwd]
//   dV = P^T dO                 (k = query index: strided in BOTH operands)
//   dQ = scale * dS K           (k = key index: contiguous in dS, strided in K)
//   dK = scale * dS^T Q         (k = query index: strided in both)
// One MFMA kernel serves all five: an operand whose k index is contiguous in memory is staged as
// [row][k] and read with ds_read_b128; an operand whose k index is strided is staged as [k][row] and
// read with the transposing ds_read_b64_tr_b16 (same idiom as conv_wgrad.hip) — no transposed copies.
// Workgroup = 4 waves = a 64 x 64 tile of C (2 x 2 waves of 32 x 32), K streamed in chunks of 32.
//
// Replaces autograd through models/DxMI/unet_small.py:175-187 (and models/cm/unet.py:413-441).
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_S16X4(p) ((__attribute__((address_space(3))) s16x4*)(p))

struct BgemmArgs {
    const bf16* A;
    const bf16* B;
    void* C;
    int M, N, K;
    long a_b0, a_b1, b_b0, b_b1, c_b0, c_b1;  // batch strides (outer = image, inner = head), elements
    int a_ld, b_ld, c_ld;
    int inner;       // number of inner batches (heads)
    int c_f32;
    float alpha;
};

constexpr int KC_PITCH = 80;    // [row][32 k] image: 64 B + 16 pad
constexpr int KS_PITCH = 192;   // [k][64 rows] image: 128 B + 64 pad (tr-read conflict-free)

__device__ __forceinline__ bf16x8 tr_frag2(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_S16X4(hi));
    bf16x8 r;
    short* rs = reinterpret_cast<short*>(&r);
#pragma unroll
    for (int e = 0; e < 4; ++e) { rs[e] = a[e]; rs[4 + e] = b[e]; }
    return r;
}

// AK / BK: operand's k index contiguous in memory (element (row,k) at row*ld + k) or strided ((k,row) at k*ld + row)
template <bool AK, bool BK>
__global__ __launch_bounds__(256) void bgemm_kernel(BgemmArgs p) {
    __shared__ __attribute__((aligned(16))) char sa[64 * KC_PITCH > 32 * KS_PITCH ? 64 * KC_PITCH : 32 * KS_PITCH];
    __shared__ __attribute__((aligned(16))) char sb[64 * KC_PITCH > 32 * KS_PITCH ? 64 * KC_PITCH : 32 * KS_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tn = (p.N + 63) / 64, tm = (p.M + 63) / 64;
    int b = blockIdx.x;
    const int nt = b % tn; b /= tn;
    const int mt = b % tm; b /= tm;
    const int bi = b % p.inner, bo = b / p.inner;
    const bf16* A = p.A + bo * p.a_b0 + bi * p.a_b1;
    const bf16* B = p.B + bo * p.b_b0 + bi * p.b_b1;
    const int m0 = mt * 64, n0 = nt * 64;

    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tr_col = (16 * (g & 1) + 4 * pp) * 2;
    const int tr_row = 8 * (g >> 1) + q;
    const int h = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < p.K; k0 += 32) {
        __syncthreads();
        // ---- stage A
        {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (AK) {  // [64 rows][32 k]: 4 pieces per row
                const int r = tid >> 2, pc = tid & 3;
                if (m0 + r < p.M && k0 + pc * 8 < p.K) v = *reinterpret_cast<const bf16x8*>(A + (long)(m0 + r) * p.a_ld + k0 + pc * 8);
                *reinterpret_cast<bf16x8*>(sa + r * KC_PITCH + pc * 16) = v;
            } else {   // [32 k][64 rows]: 8 pieces per k row
                const int kk = tid >> 3, pc = tid & 7;
                if (k0 + kk < p.K && m0 + pc * 8 < p.M) v = *reinterpret_cast<const bf16x8*>(A + (long)(k0 + kk) * p.a_ld + m0 + pc * 8);
                *reinterpret_cast<bf16x8*>(sa + kk * KS_PITCH + pc * 16) = v;
            }
        }
        // ---- stage B
        {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
            if (BK) {
                const int r = tid >> 2, pc = tid & 3;
                if (n0 + r < p.N && k0 + pc * 8 < p.K) v = *reinterpret_cast<const bf16x8*>(B + (long)(n0 + r) * p.b_ld + k0 + pc * 8);
                *reinterpret_cast<bf16x8*>(sb + r * KC_PITCH + pc * 16) = v;
            } else {
                const int kk = tid >> 3, pc = tid & 7;
                if (k0 + kk < p.K && n0 + pc * 8 < p.N) v = *reinterpret_cast<const bf16x8*>(B + (long)(k0 + kk) * p.b_ld + n0 + pc * 8);
                *reinterpret_cast<bf16x8*>(sb + kk * KS_PITCH + pc * 16) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af, bfr;
            if (AK) af = *reinterpret_cast<const bf16x8*>(sa + (wr * 32 + (lane & 31)) * KC_PITCH + ks * 32 + h * 16);
            else af = tr_frag2(sa + (ks * 16 + tr_row) * KS_PITCH + wr * 64 + tr_col, sa + (ks * 16 + tr_row + 4) * KS_PITCH + wr * 64 + tr_col);
            if (BK) bfr = *reinterpret_cast<const bf16x8*>(sb + (wc * 32 + (lane & 31)) * KC_PITCH + ks * 32 + h * 16);
            else bfr = tr_frag2(sb + (ks * 16 + tr_row) * KS_PITCH + wc * 64 + tr_col, sb + (ks * 16 + tr_row + 4) * KS_PITCH + wc * 64 + tr_col);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc, 0, 0, 0);
        }
    }
    // ---- D[m][n]: lane -> n = lane&31, m = (r&3) + 8*(r>>2) + 4h
    const int n = n0 + wc * 32 + (lane & 31);
    if (n < p.N) {
        const long cbase = bo * p.c_b0 + bi * p.c_b1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < p.M) {
                const float v = acc[r] * p.alpha;
                if (p.c_f32) reinterpret_cast<float*>(p.C)[cbase + (long)m * p.c_ld + n] = v;
                else reinterpret_cast<bf16*>(p.C)[cbase + (long)m * p.c_ld + n] = (bf16)v;
            }
        }
    }
}

// one wave per row: P = softmax(S) (bf16 out), dS = P o (dP - sum_j dP_j P_j) (bf16 out)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                         bf16* __restrict__ P, bf16* __restrict__ dS, long rows, int T) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* s = S + row * T;
    const float* g = dP + row * T;
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) sum += __expf(s[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) dot += g[j] * (__expf(s[j] - mx) * inv);
    dot = wave_sum(dot);
    for (int j = lane; j < T; j += 64) {
        const float pj = __expf(s[j] - mx) * inv;
        P[row * T + j] = (bf16)pj;
        dS[row * T + j] = (bf16)(pj * (g[j] - dot));
    }
}

// Same, the row held in registers: T % 4 == 0, T <= 1024.  A lane owns float4 j = 4 lane + 256 i: all loads of S and dP are issued
// up front (the loop form above re-reads S four times through dependent 4-byte loads: 160 us per launch on the 1024-token maps of
// the ImageNet-64 net), P and dS leave as 8-byte stores.
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(const float* __restrict__ S, const float* __restrict__ dP,
                                                             bf16* __restrict__ P, bf16* __restrict__ dS, long rows, int T) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(S + row * T);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(dP + row * T);
    const int n4 = T >> 2;
    f32x4 sv[4], gv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        if (j < n4) {
            sv[i] = s4[j];
            gv[i] = g4[j];
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < n4) mx = fmaxf(fmaxf(mx, fmaxf(sv[i][0], sv[i][1])), fmaxf(sv[i][2], sv[i][3]));
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sv[i][e] = __expf(sv[i][e] - mx);
                sum += sv[i][e];
            }
        }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < n4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sv[i][e] *= inv;
                dot += gv[i][e] * sv[i][e];
            }
        }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        if (j < n4) {
            bf16x4 pv, dv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pv[e] = (bf16)sv[i][e];
                dv[e] = (bf16)(sv[i][e] * (gv[i][e] - dot));
            }
            *reinterpret_cast<bf16x4*>(P + row * T + 4 * j) = pv;
            *reinterpret_cast<bf16x4*>(dS + row * T + 4 * j) = dv;
        }
    }
}

}  // namespace

extern "C" int dxmi_bgemm_bf16(const void* A, const void* B, void* C, int32_t M, int32_t N, int32_t K, int64_t a_b0,
                               int64_t a_b1, int32_t a_ld, int32_t a_kcontig, int64_t b_b0, int64_t b_b1, int32_t b_ld,
                               int32_t b_kcontig, int64_t c_b0, int64_t c_b1, int32_t c_ld, int32_t c_f32, float alpha,
                               int32_t outer, int32_t inner, void* stream) {
    DXMI_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && outer > 0 && inner > 0, "dxmi_bgemm_bf16: bad arguments");
    DXMI_CHECK_ARG(a_ld % 8 == 0 && b_ld % 8 == 0 && a_b0 % 8 == 0 && a_b1 % 8 == 0 && b_b0 % 8 == 0 && b_b1 % 8 == 0,
                   "dxmi_bgemm_bf16: operand strides must be multiples of 8 elements (16-byte vector loads)");
    DXMI_CHECK_ARG((a_kcontig ? K : M) % 8 == 0 && (b_kcontig ? K : N) % 8 == 0, "dxmi_bgemm_bf16: contiguous extents must be multiples of 8");
    BgemmArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)B; a.C = C; a.M = M; a.N = N; a.K = K;
    a.a_b0 = a_b0; a.a_b1 = a_b1; a.b_b0 = b_b0; a.b_b1 = b_b1; a.c_b0 = c_b0; a.c_b1 = c_b1;
    a.a_ld = a_ld; a.b_ld = b_ld; a.c_ld = c_ld; a.inner = inner; a.c_f32 = c_f32; a.alpha = alpha;
    const long blocks = (long)outer * inner * ((M + 63) / 64) * ((N + 63) / 64);
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (a_kcontig && b_kcontig) hipLaunchKernelGGL((bgemm_kernel<true, true>), grid, block, 0, st, a);
    else if (a_kcontig && !b_kcontig) hipLaunchKernelGGL((bgemm_kernel<true, false>), grid, block, 0, st, a);
    else if (!a_kcontig && b_kcontig) hipLaunchKernelGGL((bgemm_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bgemm_kernel<false, false>), grid, block, 0, st, a);
    DXMI_CHECK_LAUNCH("dxmi_bgemm_bf16");
    return DXMI_OK;
}

extern "C" int dxmi_softmax_bwd(const float* S, const float* dP, void* P, void* dS, int64_t rows, int32_t T, void* stream) {
    DXMI_CHECK_ARG(S && dP && P && dS && rows > 0 && T > 0, "dxmi_softmax_bwd: bad arguments");
    if (T % 4 == 0 && T <= 1024)
        hipLaunchKernelGGL(softmax_bwd_reg_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, dP,
                           (bf16*)P, (bf16*)dS, (long)rows, T);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, dP,
                           (bf16*)P, (bf16*)dS, (long)rows, T);
    DXMI_CHECK_LAUNCH("dxmi_softmax_bwd");
    return DXMI_OK;
}
