// Train-step tail of the DxMI hot path on gfx950: HBM-bound multi-tensor kernels.
//   multi-tensor Adam / RAdam   torch.optim.Adam at models/DxMI/trainer.py:264,325,389 (train_cifar10.py:283-296),
//                               torch.optim.RAdam through MixedPrecisionTrainer.optimize (models/cm/fp16_util.py:204-223)
//   gradient-norm clip          torch.nn.utils.clip_grad_norm_(..., 0.1) at trainer.py:388, :666-667 — norm and clip
//                               coefficient stay on the device (no .item())
//   counter-hash dropout        nn.Dropout(0.1) inside ResnetBlock (unet_small.py:129), live in update_sampler
//   row gather                  state_dict[key][indices][train_indices] (trainer.py:278-289): INT index path
// A launch takes up to DXMI_MT_MAX tensors by value in its kernel arguments (pointers + sizes), so there is no
// descriptor upload, no pinned staging buffer and no allocation: 330 U-Net tensors = 6 launches.  Every kernel
// streams 16 bytes per lane when the tensor's pointers are 16-byte aligned and falls back to dwords otherwise.
#include "common.h"

namespace {

constexpr int MT_BLOCK = 256;
constexpr int MT_CHUNK = 4096;          // elements per workgroup: 16 per lane, four 16-byte accesses per stream

struct MtTable {
    void* ptr[4][DXMI_MT_MAX];          // [stream][tensor]: p, g, m, v (adam) / g (norm)
    int64_t numel[DXMI_MT_MAX];
    int32_t first_block[DXMI_MT_MAX + 1];   // prefix sum of chunk counts
    float lr[DXMI_MT_MAX];              // per-tensor step size term (parameter groups differ in lr)
    int32_t count;
};

__device__ __forceinline__ int mt_find(const MtTable& t, int block) {
    int lo = 0;
#pragma unroll 1
    for (int i = 1; i < t.count; ++i) lo = (block >= t.first_block[i]) ? i : lo;
    return lo;
}

struct AdamScalars {
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float bc2_sqrt;      // sqrt(1 - beta2^t)

};

// torch._multi_tensor_adam (non-capturable, no amsgrad / weight decay / maximize), op for op, each intermediate
// rounded to fp32 exactly where the foreach implementation stores one:
//   m  = lerp(m, g, 1-b1)            -> fma(w, g - m, m)            (ATen lerp, weight < 0.5)
//   v  = v * b2 ; v = v + (1-b2) * (g * g)                          (foreach_mul_, foreach_addcmul_)
//   d  = sqrt(v) / bc2_sqrt + eps                                   (foreach_sqrt, foreach_div_, foreach_add_)
//   p  = p + step_size * (m / d)                                    (foreach_addcdiv_)
// (plain operators under `fp contract(off)` + explicit __builtin_fmaf: HIP's __fdiv_rn / __fsqrt_rn intrinsics map to the
// NATIVE approximations, while `/` and sqrtf are correctly rounded by default — measured against torch on gfx950)
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamScalars& s, float step_size) {
#pragma clang fp contract(off)
    const float diff = g - m;
    m = __builtin_fmaf(s.one_minus_beta1, diff, m);
    v = v * s.beta2;
    const float gg = g * g;
    v = __builtin_fmaf(s.one_minus_beta2, gg, v);
    const float r = __builtin_sqrtf(v) / s.bc2_sqrt;
    const float d = r + s.eps;
    const float q = m / d;
    p = __builtin_fmaf(step_size, q, p);
}

struct RAdamScalars {
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float inv_bc1;       // fp32(1 / (1 - beta1^t))
    float bc2_sqrt;      // sqrt(1 - beta2^t)
    float rect;          // variance rectification term, or < 0 when rho_t <= 5 (plain momentum step)
};

// torch._single_tensor_radam (torch/optim/radam.py): bias_corrected = m / bc1;
//   rho_t > 5:  p -= ((bias_corrected * lr) * (bc2_sqrt / (sqrt(v) + eps))) * rect      else:  p -= bias_corrected * lr
// with ATen's scalar-division forms (tensor / scalar = tensor * fp32(1/scalar); scalar / tensor = reciprocal * scalar)
__device__ __forceinline__ void radam_elem(float& p, float g, float& m, float& v, const RAdamScalars& s, float lr) {
#pragma clang fp contract(off)
    const float diff = g - m;
    m = __builtin_fmaf(s.one_minus_beta1, diff, m);
    v = v * s.beta2;
    const float gg = g * g;
    v = __builtin_fmaf(s.one_minus_beta2, gg, v);
    const float bce = m * s.inv_bc1;                // tensor / python scalar: ATen multiplies by the fp32 reciprocal
    float upd = bce * lr;
    if (s.rect >= 0.f) {
        const float den = __builtin_sqrtf(v) + s.eps;
        const float rden = 1.0f / den;               // python scalar / tensor: reciprocal() * scalar
        const float adaptive = rden * s.bc2_sqrt;
        upd = upd * adaptive;
        upd = upd * s.rect;
    }
    p = p - upd;          // param.add_(update, alpha=-1.0)
}

// hyper != nullptr (the *_dev entry points: steps replayed from a hipGraph, where kernel arguments are frozen at capture):
// the step-dependent scalars come from DEVICE memory — Adam: [bc2_sqrt, step_size[0..count)], RAdam: [1/bc1, bc2_sqrt, rect,
// lr[0..count)], fp32, formed by the host exactly as for the by-value entry points and uploaded before the replay;
// hyper_first = index of this launch's first tensor in the list.
template <int RADAM>
__global__ __launch_bounds__(MT_BLOCK) void mt_adam_kernel(MtTable t, AdamScalars sa, RAdamScalars sr,
                                                          const float* __restrict__ grad_scale,
                                                          const float* __restrict__ found_inf, int write_back_grad,
                                                          const float* __restrict__ hyper, int hyper_first) {
    const int ti = mt_find(t, blockIdx.x);
    if (found_inf && *found_inf != 0.f) return;   // overflow step is skipped on the device (fp16_util.py:208-212)
    const int64_t n = t.numel[ti];
    const int64_t base = (int64_t)(blockIdx.x - t.first_block[ti]) * MT_CHUNK;
    float* __restrict__ P = (float*)t.ptr[0][ti];
    float* __restrict__ G = (float*)t.ptr[1][ti];
    float* __restrict__ M = (float*)t.ptr[2][ti];
    float* __restrict__ V = (float*)t.ptr[3][ti];
    const float gs = grad_scale ? *grad_scale : 1.f;
    float lr = t.lr[ti];
    if (hyper) {
        if (RADAM) { sr.inv_bc1 = hyper[0]; sr.bc2_sqrt = hyper[1]; sr.rect = hyper[2]; lr = hyper[3 + hyper_first + ti]; }
        else       { sa.bc2_sqrt = hyper[0]; lr = hyper[1 + hyper_first + ti]; }
    }
    const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
    if (vec && base + MT_CHUNK <= n) {
#pragma unroll
        for (int r = 0; r < MT_CHUNK / (MT_BLOCK * 4); ++r) {
            const int64_t i = base + (int64_t)(r * MT_BLOCK + threadIdx.x) * 4;
            f32x4 p = *(const f32x4*)(P + i), g = *(const f32x4*)(G + i), m = *(const f32x4*)(M + i), v = *(const f32x4*)(V + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], ge = g[e], me = m[e], ve = v[e];
                if (grad_scale) ge = ge * gs;
                if (RADAM) radam_elem(pe, ge, me, ve, sr, lr); else adam_elem(pe, ge, me, ve, sa, lr);
                p[e] = pe; g[e] = ge; m[e] = me; v[e] = ve;
            }
            *(f32x4*)(P + i) = p; *(f32x4*)(M + i) = m; *(f32x4*)(V + i) = v;
            if (grad_scale && write_back_grad) *(f32x4*)(G + i) = g;
        }
        return;
    }
    const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
    for (int64_t i = base + threadIdx.x; i < end; i += MT_BLOCK) {
        float p = P[i], g = G[i], m = M[i], v = V[i];
        if (grad_scale) g = g * gs;
        if (RADAM) radam_elem(p, g, m, v, sr, lr); else adam_elem(p, g, m, v, sa, lr);
        P[i] = p; M[i] = m; V[i] = v;
        if (grad_scale && write_back_grad) G[i] = g;
    }
}

// ---------------------------------------------------------------------------------------------------------
// squared gradient norm: one fp32 partial per workgroup (fixed order inside the block), written to
// partials[part_offset + block]; dxmi_clip_coef reduces them in a fixed order -> run-to-run reproducible.
// A workgroup takes SQ_CH consecutive 4096-element chunks and issues the loads of all of them before the first add (one chunk per
// workgroup was four 16-byte loads per thread and a workgroup that lived for one memory round trip: 2.5 TB/s over the 1.2 GB of
// gradients of the ImageNet-64 net); every chunk keeps its own partial, formed exactly as before: the norm's bits are unchanged.
constexpr int SQ_CH = 4;
__global__ __launch_bounds__(MT_BLOCK) void mt_sqnorm_kernel(MtTable t, float* __restrict__ partials, int part_offset, int nblocks) {
    __shared__ float red[SQ_CH][MT_BLOCK / 64];
    constexpr int R = MT_CHUNK / (MT_BLOCK * 4);
    f32x4 g[SQ_CH][R];
    bool fast[SQ_CH];
    float acc[SQ_CH];
#pragma unroll
    for (int c = 0; c < SQ_CH; ++c) {
        const int bid = blockIdx.x * SQ_CH + c;
        acc[c] = 0.f;
        fast[c] = false;
        if (bid >= nblocks) continue;
        const int ti = mt_find(t, bid);
        const int64_t n = t.numel[ti];
        const int64_t base = (int64_t)(bid - t.first_block[ti]) * MT_CHUNK;
        const float* __restrict__ G = (const float*)t.ptr[1][ti];
        if ((((uintptr_t)G) & 15) == 0 && base + MT_CHUNK <= n) {
            fast[c] = true;
#pragma unroll
            for (int r = 0; r < R; ++r) g[c][r] = *(const f32x4*)(G + base + (int64_t)(r * MT_BLOCK + threadIdx.x) * 4);
        } else {
            const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
            for (int64_t i = base + threadIdx.x; i < end; i += MT_BLOCK) acc[c] += G[i] * G[i];
        }
    }
#pragma unroll
    for (int c = 0; c < SQ_CH; ++c) {
        if (fast[c]) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[c] += g[c][r][0] * g[c][r][0] + g[c][r][1] * g[c][r][1] + g[c][r][2] * g[c][r][2] + g[c][r][3] * g[c][r][3];
        }
        const float a = wave_sum(acc[c]);
        if ((threadIdx.x & 63) == 0) red[c][threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x < SQ_CH) {
        const int bid = blockIdx.x * SQ_CH + threadIdx.x;
        if (bid < nblocks) partials[part_offset + bid] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// total = sqrt(sum partials); coef = min(1, max_norm / (total + 1e-6))   (torch.nn.utils.clip_grad_norm_);
// also flags a non-finite norm (MixedPrecisionTrainer's overflow test) in out[2].
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                       float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    // sixteen loads in flight per thread, added in index order (a plain loop was one round trip per 256 partials: 62 us for the
    // ImageNet-64 net's ~5 k partials)
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i0 + u * 256 < n) v[u] = partials[i0 + u * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (i0 + u * 256 < n) acc += v[u];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
        const float coef = max_norm / (total + 1e-6f);
        out[0] = total;
        out[1] = (max_norm > 0.f && coef < 1.f) ? coef : 1.f;
        out[2] = (total - total == 0.f) ? 0.f : 1.f;     // inf / nan -> 1
    }
}

__global__ __launch_bounds__(MT_BLOCK) void mt_scale_kernel(MtTable t, const float* __restrict__ scale) {
    const float s = *scale;
    if (s == 1.f) return;
    const int ti = mt_find(t, blockIdx.x);
    const int64_t n = t.numel[ti];
    const int64_t base = (int64_t)(blockIdx.x - t.first_block[ti]) * MT_CHUNK;
    float* __restrict__ G = (float*)t.ptr[1][ti];
    const int64_t end = base + MT_CHUNK < n ? base + MT_CHUNK : n;
    if ((((uintptr_t)G) & 15) == 0 && base + MT_CHUNK <= n) {
#pragma unroll
        for (int r = 0; r < MT_CHUNK / (MT_BLOCK * 4); ++r) {
            f32x4* q = (f32x4*)(G + base + (int64_t)(r * MT_BLOCK + threadIdx.x) * 4);
            f32x4 g = *q;
            g[0] *= s; g[1] *= s; g[2] *= s; g[3] *= s;
            *q = g;
        }
        return;
    }
    for (int64_t i = base + threadIdx.x; i < end; i += MT_BLOCK) G[i] *= s;
}

// ---------------------------------------------------------------------------------------------------------
// Dropout with a counter-based hash: element i of the call is kept iff (mix32(i ^ seed) >> 8) >= p * 2^24.
// The same (seed, i) regenerates the mask in the backward pass: nothing is stored.  bf16 in, bf16 out,
// y = bf16(x * 1/(1-p)) or 0.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}

__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n8,
                                                          uint32_t seed, const uint32_t* __restrict__ seed_dev, uint32_t thresh,
                                                          float scale) {
    if (seed_dev) seed = *seed_dev;      // dxmi_dropout_bf16_dev: the seed of a replayed (hipGraph) step lives in device memory
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = *(const bf16x8*)(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t idx = (uint32_t)(i * 8 + e);
            const bool keep = (mix32(idx ^ seed) >> 8) >= thresh;
            o[e] = keep ? (bf16)((float)v[e] * scale) : (bf16)0.f;
        }
        *(bf16x8*)(y + i * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------
// INT path: dst[r] = src[idx[r]] for rows of row_bytes (multiple of 16) — one wave per 1 KiB of a row,
// 16 bytes per lane, index loaded once per workgroup row.  Negative indices wrap (python semantics);
// out-of-range rows are filled with 0xFF bytes (NaN for floats) instead of reading out of bounds.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ idx,
                                                         uint8_t* __restrict__ dst, int64_t n_src_rows, int64_t row_bytes,
                                                         int blocks_per_row) {
    const int64_t r = blockIdx.x / blocks_per_row;
    const int piece = blockIdx.x % blocks_per_row;
    int64_t s = idx[r];
    if (s < 0) s += n_src_rows;
    const bool ok = s >= 0 && s < n_src_rows;
    const int64_t vecs = row_bytes / 16;
    const u32x4* __restrict__ sp = (const u32x4*)(src + (ok ? s : 0) * row_bytes);
    u32x4* __restrict__ dp = (u32x4*)(dst + r * row_bytes);
    const u32x4 poison = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int64_t i = (int64_t)piece * 256 + threadIdx.x; i < vecs; i += (int64_t)blocks_per_row * 256)
        dp[i] = ok ? sp[i] : poison;
}

// small rows (< 16 bytes, e.g. the int64 timestep / y columns, fp32 sigma / logp): one element of 4 or 8 bytes per thread
template <typename T>
__global__ __launch_bounds__(256) void gather_elems_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx,
                                                          T* __restrict__ dst, int64_t n_rows, int64_t n_src_rows, int per_row) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows * per_row) return;
    const int64_t r = i / per_row;
    int64_t s = idx[r];
    if (s < 0) s += n_src_rows;
    T v;
    if (s >= 0 && s < n_src_rows) v = src[s * per_row + i % per_row];
    else memset(&v, 0xFF, sizeof(T));
    dst[i] = v;
}

bool fill_table(MtTable& t, int count, void* const* p, void* const* g, void* const* m, void* const* v,
                const int64_t* numel, const float* lr, int* blocks) {
    t.count = count;
    int nb = 0;
    for (int i = 0; i < count; ++i) {
        t.ptr[0][i] = p ? p[i] : nullptr;
        t.ptr[1][i] = g ? g[i] : nullptr;
        t.ptr[2][i] = m ? m[i] : nullptr;
        t.ptr[3][i] = v ? v[i] : nullptr;
        t.numel[i] = numel[i];
        t.lr[i] = lr ? lr[i] : 0.f;
        t.first_block[i] = nb;
        if (numel[i] <= 0) return false;
        nb += (int)((numel[i] + MT_CHUNK - 1) / MT_CHUNK);
    }
    t.first_block[count] = nb;
    *blocks = nb;
    return true;
}

}  // namespace

extern "C" int64_t dxmi_mt_blocks(const int64_t* numel, int32_t count) {
    int64_t nb = 0;
    for (int i = 0; i < count; ++i) nb += (numel[i] + MT_CHUNK - 1) / MT_CHUNK;
    return nb;
}

static int adam_launch(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                       const int64_t* numel, const float* step_size, int32_t count, double beta1, double beta2,
                       double eps, double bc2_sqrt, const float* grad_scale, int32_t write_back_grad, const float* hyper,
                       void* stream);

extern "C" int dxmi_adam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                              const int64_t* numel, const float* step_size, int32_t count, double beta1, double beta2,
                              double eps, double bc2_sqrt, const float* grad_scale, int32_t write_back_grad, void* stream) {
    DXMI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel && step_size && count > 0, "dxmi_adam_step: null argument");
    return adam_launch(params, grads, exp_avg, exp_avg_sq, numel, step_size, count, beta1, beta2, eps, bc2_sqrt, grad_scale,
                       write_back_grad, nullptr, stream);
}

extern "C" int dxmi_adam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                  const int64_t* numel, int32_t count, double beta1, double beta2, double eps,
                                  const float* hyper, const float* grad_scale, int32_t write_back_grad, void* stream) {
    DXMI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel && hyper && count > 0, "dxmi_adam_step_dev: null argument");
    return adam_launch(params, grads, exp_avg, exp_avg_sq, numel, nullptr, count, beta1, beta2, eps, 1.0, grad_scale,
                       write_back_grad, hyper, stream);
}

static int adam_launch(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                       const int64_t* numel, const float* step_size, int32_t count, double beta1, double beta2,
                       double eps, double bc2_sqrt, const float* grad_scale, int32_t write_back_grad, const float* hyper,
                       void* stream) {
    AdamScalars sa;
    // python-double scalars rounded to fp32 once, exactly as ATen's Scalar -> opmath_t conversion does
    sa.one_minus_beta1 = (float)(1.0 - beta1);
    sa.beta2 = (float)beta2;
    sa.one_minus_beta2 = (float)(1.0 - beta2);
    sa.eps = (float)eps;
    sa.bc2_sqrt = (float)bc2_sqrt;

    RAdamScalars sr = {};
    for (int off = 0; off < count; off += DXMI_MT_MAX) {
        const int c = count - off < DXMI_MT_MAX ? count - off : DXMI_MT_MAX;
        MtTable t;
        int nb = 0;
        DXMI_CHECK_ARG(fill_table(t, c, params + off, grads + off, exp_avg + off, exp_avg_sq + off, numel + off,
                                  step_size ? step_size + off : nullptr, &nb),
                       "dxmi_adam_step: empty tensor in the list");
        hipLaunchKernelGGL(mt_adam_kernel<0>, dim3(nb), dim3(MT_BLOCK), 0, (hipStream_t)stream, t, sa, sr, grad_scale,
                           (const float*)nullptr, write_back_grad, hyper, off);
    }
    DXMI_CHECK_LAUNCH("dxmi_adam_step");
    return DXMI_OK;
}

static int radam_launch(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                        const int64_t* numel, const float* lr, int32_t count, double beta1, double beta2, double eps,
                        double bc1, double bc2_sqrt, double rect, const float* grad_scale, const float* found_inf,
                        const float* hyper, void* stream);

extern "C" int dxmi_radam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                               const int64_t* numel, const float* lr, int32_t count, double beta1, double beta2, double eps,
                               double bc1, double bc2_sqrt, double rect, const float* grad_scale, const float* found_inf,
                               void* stream) {
    DXMI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel && lr && count > 0, "dxmi_radam_step: null argument");
    return radam_launch(params, grads, exp_avg, exp_avg_sq, numel, lr, count, beta1, beta2, eps, bc1, bc2_sqrt, rect, grad_scale,
                        found_inf, nullptr, stream);
}

extern "C" int dxmi_radam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                   const int64_t* numel, int32_t count, double beta1, double beta2, double eps,
                                   const float* hyper, const float* grad_scale, const float* found_inf, void* stream) {
    DXMI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel && hyper && count > 0, "dxmi_radam_step_dev: null argument");
    return radam_launch(params, grads, exp_avg, exp_avg_sq, numel, nullptr, count, beta1, beta2, eps, 1.0, 1.0, -1.0, grad_scale,
                        found_inf, hyper, stream);
}

static int radam_launch(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                        const int64_t* numel, const float* lr, int32_t count, double beta1, double beta2, double eps,
                        double bc1, double bc2_sqrt, double rect, const float* grad_scale, const float* found_inf,
                        const float* hyper, void* stream) {
    RAdamScalars sr;
    sr.one_minus_beta1 = (float)(1.0 - beta1);
    sr.beta2 = (float)beta2;
    sr.one_minus_beta2 = (float)(1.0 - beta2);
    sr.eps = (float)eps;
    sr.inv_bc1 = (float)(1.0 / bc1);     // reciprocal formed in double, rounded once (matches ATen bit for bit)
    sr.bc2_sqrt = (float)bc2_sqrt;
    sr.rect = (float)rect;
    AdamScalars sa = {};
    for (int off = 0; off < count; off += DXMI_MT_MAX) {
        const int c = count - off < DXMI_MT_MAX ? count - off : DXMI_MT_MAX;
        MtTable t;
        int nb = 0;
        DXMI_CHECK_ARG(fill_table(t, c, params + off, grads + off, exp_avg + off, exp_avg_sq + off, numel + off, lr ? lr + off : nullptr, &nb),
                       "dxmi_radam_step: empty tensor in the list");
        hipLaunchKernelGGL(mt_adam_kernel<1>, dim3(nb), dim3(MT_BLOCK), 0, (hipStream_t)stream, t, sa, sr, grad_scale, found_inf, 0,
                           hyper, off);
    }
    DXMI_CHECK_LAUNCH("dxmi_radam_step");
    return DXMI_OK;
}

extern "C" int dxmi_gradnorm_clip(void* const* grads, const int64_t* numel, int32_t count, float max_norm, float* partials,
                                  float* out3, int32_t scale_in_place, void* stream) {
    DXMI_CHECK_ARG(grads && numel && partials && out3 && count > 0, "dxmi_gradnorm_clip: null argument");
    int total_blocks = 0;
    for (int off = 0; off < count; off += DXMI_MT_MAX) {
        const int c = count - off < DXMI_MT_MAX ? count - off : DXMI_MT_MAX;
        MtTable t;
        int nb = 0;
        DXMI_CHECK_ARG(fill_table(t, c, nullptr, grads + off, nullptr, nullptr, numel + off, nullptr, &nb),
                       "dxmi_gradnorm_clip: empty tensor in the list");
        hipLaunchKernelGGL(mt_sqnorm_kernel, dim3((nb + SQ_CH - 1) / SQ_CH), dim3(MT_BLOCK), 0, (hipStream_t)stream, t, partials, total_blocks, nb);
        total_blocks += nb;
    }
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, total_blocks, max_norm, out3);
    if (scale_in_place) {
        for (int off = 0; off < count; off += DXMI_MT_MAX) {
            const int c = count - off < DXMI_MT_MAX ? count - off : DXMI_MT_MAX;
            MtTable t;
            int nb = 0;
            fill_table(t, c, nullptr, grads + off, nullptr, nullptr, numel + off, nullptr, &nb);
            hipLaunchKernelGGL(mt_scale_kernel, dim3(nb), dim3(MT_BLOCK), 0, (hipStream_t)stream, t, out3 + 1);
        }
    }
    DXMI_CHECK_LAUNCH("dxmi_gradnorm_clip");
    return DXMI_OK;
}

static int dropout_launch(const void* x, void* y, int64_t n, float p, uint32_t seed, const uint32_t* seed_dev, void* stream);

extern "C" int dxmi_dropout_bf16(const void* x, void* y, int64_t n, float p, uint32_t seed, void* stream) {
    return dropout_launch(x, y, n, p, seed, nullptr, stream);
}

extern "C" int dxmi_dropout_bf16_dev(const void* x, void* y, int64_t n, float p, const uint32_t* seed, void* stream) {
    DXMI_CHECK_ARG(seed, "dxmi_dropout_bf16_dev: null seed pointer");
    return dropout_launch(x, y, n, p, 0u, seed, stream);
}

static int dropout_launch(const void* x, void* y, int64_t n, float p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
    DXMI_CHECK_ARG(x && y && n > 0 && n % 8 == 0 && n < ((int64_t)1 << 32), "dxmi_dropout_bf16: n (%lld) must be a multiple of 8 below 2^32", (long long)n);
    DXMI_CHECK_ARG(p >= 0.f && p < 1.f, "dxmi_dropout_bf16: p (%f) outside [0,1)", p);
    const uint32_t thresh = (uint32_t)((double)p * 16777216.0);
    const float scale = (float)(1.0 / (1.0 - (double)p));
    const int64_t n8 = n / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n8,
                       seed, seed_dev, thresh, scale);
    DXMI_CHECK_LAUNCH("dxmi_dropout_bf16");
    return DXMI_OK;
}

extern "C" int dxmi_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t n_rows, int64_t n_src_rows,
                                int64_t row_bytes, void* stream) {
    DXMI_CHECK_ARG(src && idx && dst && n_rows >= 0 && n_src_rows > 0 && row_bytes > 0, "dxmi_gather_rows: bad arguments");
    if (n_rows == 0) return DXMI_OK;
    if (row_bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        int bpr = (int)((row_bytes / 16 + 1023) / 1024);    // up to 4 x 16 B per lane per workgroup
        if (bpr < 1) bpr = 1;
        DXMI_CHECK_ARG(n_rows * bpr < ((int64_t)1 << 31), "dxmi_gather_rows: too many rows");
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(n_rows * bpr)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src,
                           idx, (uint8_t*)dst, n_src_rows, row_bytes, bpr);
    } else if (row_bytes % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 7) == 0) {
        const int per = (int)(row_bytes / 8);
        hipLaunchKernelGGL(gather_elems_kernel<uint64_t>, dim3((unsigned)((n_rows * per + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint64_t*)src, idx, (uint64_t*)dst, n_rows, n_src_rows, per);
    } else {
        DXMI_CHECK_ARG(row_bytes % 4 == 0, "dxmi_gather_rows: row_bytes (%lld) must be a multiple of 4", (long long)row_bytes);
        const int per = (int)(row_bytes / 4);
        hipLaunchKernelGGL(gather_elems_kernel<uint32_t>, dim3((unsigned)((n_rows * per + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t*)src, idx, (uint32_t*)dst, n_rows, n_src_rows, per);
    }
    DXMI_CHECK_LAUNCH("dxmi_gather_rows");
    return DXMI_OK;
}
