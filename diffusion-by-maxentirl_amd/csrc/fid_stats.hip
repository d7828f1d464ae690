// Activation statistics of the FID on the device (the half that needs no Inception weights).
//
// Reference: train_image_large.py:62-69 / train_cifar10.py fid(): after the all_gather of the per-rank activations rank 0 takes
//     m1, s1 = np.mean(act, axis=0), np.cov(act, rowvar=False)          (act float32 [N, dims]; np.cov works in float64)
// on the HOST: at 50 000 x 2048 that is a 420-GFLOP float64 GEMM on one core group.  Here: column means in float64, the centred
// Gram matrix on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact float32 products, float32 accumulation inside a split of
// the rows), split partials folded in float64 with the correction for the float32 rounding of the mean used for centring.
// Only the upper-triangular 128 x 128 tiles are computed; the fold mirrors them.
//
//   mean_partial_kernel   X [N, D] f32 -> partial column sums [RB][D] f64 (256 rows per block, one thread per column)
//   mean_fold_kernel      -> mu [D] f64 (+ its float32 rounding, which the Gram kernel subtracts)
//   gram_f32_kernel       tile (bi <= bj) x row split s: G_s = sum_n (x_n[i] - m[i]) (x_n[j] - m[j]); 4 waves x (64 x 64), K chunks of
//                         16 rows through LDS (centred at the store), 32 MFMAs per wave and chunk
//   cov_fold_kernel       sigma[i][j] = sigma[j][i] = (sum_s G_s[i][j] - N d_i d_j) / (N - 1) in float64, d = mu - float32(mu)
// Roofline: f32 MFMA, 157 TFLOP/s dense (MI355X_MICROARCH.md); algorithmic work N * D * (D + 128) FLOP (upper triangle).
#include "common.h"

namespace {

constexpr int FS_T = 128;          // tile edge (features)
constexpr int FS_KC = 16;          // rows (samples) per LDS chunk (2 buffers x 2 operands x 16 x 132 floats = 33 KB)

__global__ __launch_bounds__(256) void mean_partial_kernel(const float* __restrict__ x, double* __restrict__ part, long N, int D) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const long r0 = (long)blockIdx.y * 256, r1 = r0 + 256 < N ? r0 + 256 : N;
    if (col >= D) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;       // four chains: loads in flight; fixed order -> reproducible
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
        const float a = x[r * D + col], b = x[(r + 1) * D + col], c = x[(r + 2) * D + col], d = x[(r + 3) * D + col];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; r < r1; ++r) s0 += x[r * D + col];
    part[(long)blockIdx.y * D + col] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void mean_fold_kernel(const double* __restrict__ part, double* __restrict__ mu, float* __restrict__ mu32,
                                                       int RB, long N, int D) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    double s = 0.0;
    for (int b = 0; b < RB; ++b) s += part[(long)b * D + col];
    s /= (double)N;
    mu[col] = s;
    mu32[col] = (float)s;
}

// tile t of the upper triangle (bi <= bj) of an nb x nb tile grid, row-major over bi
__host__ __device__ inline void tri_tile(int t, int nb, int& bi, int& bj) {
    bi = 0;
    while (t >= nb - bi) { t -= nb - bi; ++bi; }
    bj = bi + t;
}

__global__ __launch_bounds__(256) void gram_f32_kernel(const float* __restrict__ x, const float* __restrict__ mu32, float* __restrict__ part,
                                                      long N, int D, int nb, int ntiles, long rows_per_split) {
    __shared__ float As[2][FS_KC][FS_T + 4];
    __shared__ float Bs[2][FS_KC][FS_T + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
    int bi, bj;
    tri_tile(t, nb, bi, bj);
    const long n0 = (long)split * rows_per_split, n1 = n0 + rows_per_split < N ? n0 + rows_per_split : N;
    const int wi = wave >> 1, wj = wave & 1;             // wave tile: rows (features i) wi*64.., columns (features j) wj*64..
    // loader mapping: thread -> column quad c4 (0..31), rows r8 + 8 k (k = 0..FS_KC/8 - 1)
    const int c4 = tid & 31, r8 = tid >> 5;
    const int ci = bi * FS_T + c4 * 4, cj = bj * FS_T + c4 * 4;
    f32x4 mi = {0.f, 0.f, 0.f, 0.f}, mj = {0.f, 0.f, 0.f, 0.f};
    bool vi[4], vj[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        vi[e] = ci + e < D; vj[e] = cj + e < D;
        if (vi[e]) mi[e] = mu32[ci + e];
        if (vj[e]) mj[e] = mu32[cj + e];
    }
    const bool full_i = ci + 3 < D, full_j = cj + 3 < D, diag = bi == bj;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    constexpr int LK = FS_KC / 8;
    f32x4 ra[LK], rb[LK];
    auto fetch = [&](long nbase) {
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            const long n = nbase + r8 + 8 * k;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (n < n1) {
                const float* row = x + n * D;
                if (full_i) a = *reinterpret_cast<const f32x4*>(row + ci);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (vi[e]) a[e] = row[ci + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = vi[e] ? a[e] - mi[e] : 0.f;
                if (!diag) {
                    if (full_j) b = *reinterpret_cast<const f32x4*>(row + cj);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (vj[e]) b[e] = row[cj + e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = vj[e] ? b[e] - mj[e] : 0.f;
                }
            }
            ra[k] = a; rb[k] = b;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            *reinterpret_cast<f32x4*>(&As[buf][r8 + 8 * k][c4 * 4]) = ra[k];
            if (!diag) *reinterpret_cast<f32x4*>(&Bs[buf][r8 + 8 * k][c4 * 4]) = rb[k];
        }
    };
    const long nchunks = (n1 - n0 + FS_KC - 1) / FS_KC;
    if (nchunks > 0) {
        fetch(n0);
        stage(0);
    }
    __syncthreads();
    for (long c = 0; c < nchunks; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < nchunks) fetch(n0 + (c + 1) * FS_KC);          // next chunk's loads fly under this chunk's MFMAs
        const float (*Bsrc)[FS_T + 4] = diag ? As[buf] : Bs[buf];
#pragma unroll
        for (int kk = 0; kk < FS_KC / 2; ++kk) {
            const int kr = kk * 2 + (lane >> 5), cl = lane & 31;
            float a[2], b[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[u] = As[buf][kr][wi * 64 + u * 32 + cl];
                b[u] = Bsrc[kr][wj * 64 + u * 32 + cl];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[v], acc[u][v], 0, 0, 0);
        }
        if (c + 1 < nchunks) stage(buf ^ 1);
        __syncthreads();
    }
    // partial tile: [split][t][128][128] f32; D layout of the 32x32 MFMA: column = lane % 32, rows (lane / 32) * 4 + 8 * q + e
    float* const dst = part + ((size_t)split * ntiles + t) * (FS_T * FS_T);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = wi * 64 + u * 32 + (lane >> 5) * 4 + q * 8 + e, j = wj * 64 + v * 32 + (lane & 31);
                    dst[i * FS_T + j] = acc[u][v][q * 4 + e];
                }
}

__global__ __launch_bounds__(256) void cov_fold_kernel(const float* __restrict__ part, const double* __restrict__ mu, const float* __restrict__ mu32,
                                                      double* __restrict__ sigma, long N, int D, int nb, int ntiles, int S) {
    const int t = blockIdx.x;
    int bi, bj;
    tri_tile(t, nb, bi, bj);
    const double inv = 1.0 / (double)(N - 1);
    for (int p = threadIdx.x; p < FS_T * FS_T; p += 256) {
        const int i = bi * FS_T + p / FS_T, j = bj * FS_T + p % FS_T;
        if (i >= D || j >= D) continue;
        double s = 0.0;
        for (int sp = 0; sp < S; ++sp) s += (double)part[((size_t)sp * ntiles + t) * (FS_T * FS_T) + p];
        const double di = mu[i] - (double)mu32[i], dj = mu[j] - (double)mu32[j];
        const double c = (s - (double)N * di * dj) * inv;
        sigma[(size_t)i * D + j] = c;
        if (bi != bj) sigma[(size_t)j * D + i] = c;
    }
}

struct FsPlan {
    int nb, ntiles, S, RB;
    long rows_per_split;
    size_t off_mu32, off_meanpart, off_part, total;
};

FsPlan fs_plan(long N, int D) {
    FsPlan p;
    p.nb = (D + FS_T - 1) / FS_T;
    p.ntiles = p.nb * (p.nb + 1) / 2;
    int S = (512 + p.ntiles - 1) / p.ntiles;              // >= 2 workgroups per CU
    const long maxS = (N + 4 * FS_KC - 1) / (4 * FS_KC);  // at least four chunks per split
    if (S > maxS) S = (int)maxS;
    const long minS = (N + 2047) / 2048;                  // <= 2048 rows of fp32 accumulation per split (the fold adds the splits in fp64)
    if (S < minS) S = (int)minS;
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    long rps = (N + S - 1) / S;
    rps = (rps + FS_KC - 1) / FS_KC * FS_KC;
    p.S = (int)((N + rps - 1) / rps);
    p.rows_per_split = rps;
    p.RB = (int)((N + 255) / 256);
    p.off_mu32 = 0;
    p.off_meanpart = ((size_t)D * 4 + 255) / 256 * 256;
    p.off_part = p.off_meanpart + ((size_t)p.RB * D * 8 + 255) / 256 * 256;
    p.total = p.off_part + (size_t)p.S * p.ntiles * FS_T * FS_T * 4;
    return p;
}

}  // namespace

extern "C" int64_t dxmi_fid_stats_workspace_bytes(int64_t N, int32_t D) {
    if (N < 2 || D <= 0) return 0;
    return (int64_t)fs_plan(N, D).total;
}

extern "C" int dxmi_fid_stats(const float* act, int64_t N, int32_t D, double* mu, double* sigma, void* workspace, void* stream) {
    DXMI_CHECK_ARG(act && mu && sigma && workspace, "dxmi_fid_stats: null pointer");
    DXMI_CHECK_ARG(N >= 2 && D > 0 && D <= 16384, "dxmi_fid_stats: need N >= 2 samples and 0 < D <= 16384 (N %lld, D %d)", (long long)N, D);
    DXMI_CHECK_ARG(D % 4 == 0, "dxmi_fid_stats: D %d must be a multiple of 4 (the feature sizes of pytorch_fid are 64 / 192 / 768 / 2048)", D);
    const FsPlan p = fs_plan(N, D);
    char* ws = reinterpret_cast<char*>(workspace);
    float* mu32 = reinterpret_cast<float*>(ws + p.off_mu32);
    double* mpart = reinterpret_cast<double*>(ws + p.off_meanpart);
    float* part = reinterpret_cast<float*>(ws + p.off_part);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mean_partial_kernel, dim3((D + 255) / 256, p.RB), dim3(256), 0, st, act, mpart, (long)N, D);
    DXMI_CHECK_LAUNCH("dxmi_fid_stats(mean partials)");
    hipLaunchKernelGGL(mean_fold_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const double*)mpart, mu, mu32, p.RB, (long)N, D);
    DXMI_CHECK_LAUNCH("dxmi_fid_stats(mean)");
    hipLaunchKernelGGL(gram_f32_kernel, dim3(p.ntiles * p.S), dim3(256), 0, st, act, (const float*)mu32, part, (long)N, D, p.nb, p.ntiles,
                       p.rows_per_split);
    DXMI_CHECK_LAUNCH("dxmi_fid_stats(gram)");
    hipLaunchKernelGGL(cov_fold_kernel, dim3(p.ntiles), dim3(256), 0, st, (const float*)part, (const double*)mu, (const float*)mu32, sigma,
                       (long)N, D, p.nb, p.ntiles, p.S);
    DXMI_CHECK_LAUNCH("dxmi_fid_stats(fold)");
    return DXMI_OK;
}
