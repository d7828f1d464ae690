// HBM-bound / tiny kernels of the DxMI hot path for gfx950: sampler transition, schedule
// gathers (integer index path), timestep sinusoid, value-net pooling + head, layout edges.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// sinusoidal timestep embedding (unet_small.py:9-27 / models/cm/nn.py:119-137)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int N, int dim,
                                          int order, float log_period) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * half) return;
    const int n = idx / half, i = idx % half;
    const float denom = order == 0 ? (float)(half - 1) : (float)half;
    // same fp32 evaluation order as the reference: exp(arange * -(log(P)/denom)) (order 0) or
    // exp(-log(P) * arange / denom) (order 1)
    float freq;
    if (order == 0) freq = expf((float)i * -(log_period / denom));
    else freq = expf(-log_period * (float)i / denom);
    const float arg = t[n] * freq;
    const float s = sinf(arg), c = cosf(arg);
    float* o = out + (size_t)n * dim;
    if (order == 0) { o[i] = s; o[half + i] = c; }
    else { o[i] = c; o[half + i] = s; }
    if ((dim & 1) && i == 0) o[dim - 1] = 0.f;
}

// ---------------------------------------------------------------------------------------
// INT path: gather the per-sample schedule scalars by integer timestep.
__global__ void var_gather_sched_kernel(const int64_t* __restrict__ t, const float* __restrict__ cont,
                                        const float* __restrict__ xmul_tab, const float* __restrict__ cmul_tab,
                                        const float* __restrict__ log_betas_all, float* __restrict__ tau,
                                        float* __restrict__ xmul, float* __restrict__ cmul, float* __restrict__ sigma,
                                        int N, int T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N) return;
    long ti = t[b];
    if (ti < 0) ti += T;  // python-style wrap, as torch indexing does
    if (ti < 0 || ti >= T) {
        // torch indexing raises here (IndexError); a kernel cannot, and must not read out of bounds: poison the
        // sample's scalars so everything derived from them is NaN (the host wrapper checks host-resident t eagerly)
        const float q = __builtin_nanf("");
        tau[b] = q; xmul[b] = q; cmul[b] = q; sigma[b] = q;
        return;
    }
    tau[b] = cont[ti];
    xmul[b] = xmul_tab[ti];
    cmul[b] = cmul_tab[ti];
    sigma[b] = expf(log_betas_all[ti]);
}

// ---------------------------------------------------------------------------------------
// Fused sampler transition.  One workgroup per sample: x, eps, z are read once; x', mean,
// control written once; the CHW log-prob mean is reduced in-kernel (fp32, fixed order).
__global__ __launch_bounds__(256) void var_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                      const float* __restrict__ z, const float* __restrict__ xmul,
                                                      const float* __restrict__ cmul, const float* __restrict__ sigma,
                                                      float* __restrict__ x_next, float* __restrict__ mean,
                                                      float* __restrict__ control, float* __restrict__ logp, int CHW, int assoc) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float xm = xmul[b], cm = cmul[b], sg = sigma[b];
    const float var2 = 2.f * sg * sg;
    const float log_sg = logf(sg);
    const float half_log_2pi = 0.918938533204672742f;  // log(sqrt(2*pi))
    const size_t base = (size_t)b * CHW;
    float acc = 0.f;
    for (int i = threadIdx.x * 4; i < CHW; i += 256 * 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
        const f32x4 ev = *reinterpret_cast<const f32x4*>(eps + base + i);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + base + i);
        f32x4 xn, mu, ct;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xs = xv[e] * xm;
            ct[e] = cm * ev[e];
            mu[e] = xs + ct[e];
            // assoc 1: x += control + sigma*z (var_sampler.py:285); assoc 0: mean + sigma*z (:399)
            xn[e] = assoc ? xs + (ct[e] + sg * zv[e]) : mu[e] + sg * zv[e];
            const float d = xn[e] - mu[e];
            acc += -(d * d) / var2 - log_sg - half_log_2pi;
        }
        *reinterpret_cast<f32x4*>(x_next + base + i) = xn;
        if (mean) *reinterpret_cast<f32x4*>(mean + base + i) = mu;
        if (control) *reinterpret_cast<f32x4*>(control + base + i) = ct;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && logp) logp[b] = (red[0] + red[1] + red[2] + red[3]) / (float)CHW;
}

// ---------------------------------------------------------------------------------------
// value net: optional 2x2 average pool then activation, NHWC bf16, 8 channels per thread
__global__ void pool_act_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int N, int H, int W, int C,
                                int pool, int act) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int C8 = C / 8;
    const long total = (long)N * OH * OW * C8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = idx % C8;
        long r = idx / C8;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH;
        const int n = (int)(r / OH);
        float v[8];
        if (pool) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(in + (((size_t)n * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= 0.25f;
        } else {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(in + (((size_t)n * H + oy) * W + ox) * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)dxmi_act(v[e], act);
        *reinterpret_cast<bf16x8*>(out + (((size_t)n * OH + oy) * OW + ox) * C + c8 * 8) = o;
    }
}

// value head: relu -> sum over HW -> dot(w[C]) + b -> out_w*y + out_b ; one workgroup per image.  Four waves walk the channels
// (eight pixels of loads in flight, pixels added in order), the relu-sums s[c] meet in LDS and wave 0 adds s[c] w[c] in the order
// the one-wave form of rounds 1-5 did (lane: c = lane, lane + 64, ...; then the butterfly): the same bits, a quarter of the chain.
__global__ __launch_bounds__(256) void value_head_kernel(const bf16* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ b, const float* __restrict__ out_w,
                                                        const float* __restrict__ out_b, float* __restrict__ out, int HW, int C) {
    __shared__ float sw[4096];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        const bf16* p = in + (size_t)n * HW * C + c;
        float s = 0.f;
        for (int px = 0; px < HW; px += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (px + u < HW) v[u] = (float)p[(size_t)(px + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (px + u < HW) s += v[u] > 0.f ? v[u] : 0.f;
        }
        sw[c] = s;
    }
    __syncthreads();
    if (tid >= 64) return;
    float acc = 0.f;
    for (int c = tid; c < C; c += 64) acc += sw[c] * w[c];      // the expression of the one-wave form (same contraction)
    acc = wave_sum(acc);
    if (tid == 0) {
        float y = acc + b[0];
        if (out_w) y = y * out_w[0] + out_b[0];  // out_scale = Linear(1,1), models/modules.py:157-158
        out[n] = y;
    }
}

// backward of pool_act: 8 channels per thread over the INPUT-resolution grid
__global__ void pool_act_bwd_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ aout, bf16* __restrict__ din,
                                    int N, int H, int W, int C, int pool, float slope) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int C8 = C / 8;
    const long total = (long)N * H * W * C8;
    const float sc = pool ? 0.25f : 1.f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = idx % C8;
        long r = idx / C8;
        const int x = r % W; r /= W;
        const int y = r % H;
        const int n = (int)(r / H);
        const size_t o = (((size_t)n * OH + (pool ? y / 2 : y)) * OW + (pool ? x / 2 : x)) * C + c8 * 8;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + o);
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(aout + o);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)((float)g[e] * ((float)a[e] > 0.f ? sc : sc * slope));
        *reinterpret_cast<bf16x8*>(din + (((size_t)n * H + y) * W + x) * C + c8 * 8) = v;
    }
}

// backward of the value head w.r.t. the feature map (+ the relu/sum features for the parameter grads)
__global__ __launch_bounds__(256) void value_head_bwd_kernel(const bf16* __restrict__ feat, const float* __restrict__ w,
                                                            const float* __restrict__ dy, bf16* __restrict__ dfeat,
                                                            float* __restrict__ s, int HW, int C) {
    const int n = blockIdx.x;
    const float g = dy[n];
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        const float gw = g * w[c];
        for (int px = 0; px < HW; px += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (px + u < HW) v[u] = (float)feat[((size_t)n * HW + px + u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (px + u < HW) {
                    acc += v[u] > 0.f ? v[u] : 0.f;
                    dfeat[((size_t)n * HW + px + u) * C + c] = (bf16)(v[u] > 0.f ? gw : 0.f);
                }
        }
        s[(size_t)n * C + c] = acc;
    }
}

// Parameter gradients of the value head from one launch (models/value_train.py: ~10 torch elementwise / reduce launches per backward,
// twelve backwards per train step).  y = (s . w + b) * ow + ob with s = relu-sum features [N, C]; dy [N]:
//   d_linear_w[c] = sum_n (dy[n] * ow) * s[n, c]     d_linear_b = sum_n dy[n] * ow
//   d_ow = sum_n dy[n] * (s[n] . w + b)              d_ob = sum_n dy[n]
// Fixed summation order everywhere: bitwise reproducible.  out = [C + 3] fp32: d_linear_w | d_linear_b | d_ow | d_ob.  ow == nullptr: no out_scale (the last two are written as zero).
// Workgroups 0 .. ceil(C / 64) - 1: d_linear_w of 64 channels (thread = channel x one of four row slices, eight rows of loads in
// flight, the slices added in order); the last workgroup: the three scalars (a wave per row for the dot product s[n] . w — coalesced,
// a fixed butterfly over the lanes — rows in order per wave, the four waves added in order).  The first form of this kernel was ONE
// workgroup walking N rows / C channels one dependent load at a time: 96 us per launch at 256 images, 22 launches per train step.
__global__ __launch_bounds__(256) void value_head_pgrad_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                              const float* __restrict__ b, const float* __restrict__ dy,
                                                              const float* __restrict__ ow, float* __restrict__ out, int N, int C) {
    __shared__ float red[4][64];
    const int tid = threadIdx.x;
    const float scale = ow ? ow[0] : 1.f;
    const int CB = (C + 63) / 64;
    if ((int)blockIdx.x < CB) {
        const int c = blockIdx.x * 64 + (tid & 63), sl = tid >> 6;
        const int n0 = (int)((long)N * sl / 4), n1 = (int)((long)N * (sl + 1) / 4);
        float acc = 0.f;
        if (c < C) {
            for (int n = n0; n < n1; n += 8) {
                float sv[8], dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (n + u < n1) {
                        sv[u] = s[(size_t)(n + u) * C + c];
                        dv[u] = dy[n + u];
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (n + u < n1) acc += (dv[u] * scale) * sv[u];
            }
        }
        red[sl][tid & 63] = acc;
        __syncthreads();
        if (sl == 0 && c < C) out[c] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        return;
    }
    const int lane = tid & 63, wv = tid >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    // a wave owns the rows n = wv, wv + 4, ...; four of its rows per trip (their loads in flight together), added in row order
    for (int n = wv; n < N; n += 16) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = lane; c0 < C; c0 += 64 * 8) {
            float sv[4][8], wvv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u * 64 < C) {
                    wvv[u] = w[c0 + u * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + 4 * r < N) sv[r][u] = s[(size_t)(n + 4 * r) * C + c0 + u * 64];
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + 4 * r < N) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c0 + u * 64 < C) part[r] += sv[r][u] * wvv[u];
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) part[r] += __shfl_xor(part[r], o, 64);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + 4 * r < N) {
                const float y = part[r] + b[0], d = dy[n + 4 * r];
                a0 += d * scale;
                a1 += d * y;
                a2 += d;
            }
    }
    if (lane == 0) { red[0][wv] = a0; red[1][wv] = a1; red[2][wv] = a2; }
    __syncthreads();
    if (tid == 0) {
        out[C] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        out[C + 1] = ow ? ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3] : 0.f;
        out[C + 2] = ow ? ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3] : 0.f;
    }
}

// TD step of DxMI_Trainer.update_f_v on a replay ring (models/DxMI/trainer.py:278-300; reference trainer.py:271-300), data side in ONE
// launch: the rows `state[r]` and (when next_rows) `next_state[r]` of the buffered transitions are gathered straight into the two
// halves of the batch the value net evaluates ([next_state | state]: TD target and TD prediction share one forward), and the
// running cost ||x' - x||^2 / (2 beta) averaged over CHW (trainer.py:163-169) is reduced on the way.  INT path: rows are int64
// indices into the ring's [rows, CHW] fp32 trajectory block; an index outside [0, n_src_rows) poisons the row with NaN.
// next_dense: the re-drawn next states of `value_resample` (already dense [B, CHW]) instead of a gather.  beta: DEVICE scalar.
__global__ __launch_bounds__(256) void td_gather_cost_kernel(const float* __restrict__ traj, const int64_t* __restrict__ state_rows,
                                                            const int64_t* __restrict__ next_rows, const float* __restrict__ next_dense,
                                                            const float* __restrict__ beta, float* __restrict__ out_next,
                                                            float* __restrict__ out_state, float* __restrict__ cost, int CHW,
                                                            int64_t n_src_rows) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int64_t rs = state_rows[b];
    const int64_t rn = next_rows ? next_rows[b] : 0;
    const bool ok = rs >= 0 && rs < n_src_rows && (!next_rows || (rn >= 0 && rn < n_src_rows));
    const float* ps = traj + (size_t)(ok ? rs : 0) * CHW;
    const float* pn = next_rows ? traj + (size_t)(ok ? rn : 0) * CHW : next_dense + (size_t)b * CHW;
    const float two_beta = 2.f * beta[0];
    const float nanv = __builtin_nanf("");
    float acc = 0.f;
    for (int i = threadIdx.x * 4; i < CHW; i += 256 * 4) {
        f32x4 sv = *reinterpret_cast<const f32x4*>(ps + i);
        f32x4 nv = *reinterpret_cast<const f32x4*>(pn + i);
        if (!ok) { sv = f32x4{nanv, nanv, nanv, nanv}; nv = sv; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = nv[e] - sv[e];
            acc += (d * d) / two_beta;
        }
        *reinterpret_cast<f32x4*>(out_state + (size_t)b * CHW + i) = sv;
        if (out_next) *reinterpret_cast<f32x4*>(out_next + (size_t)b * CHW + i) = nv;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cost[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)CHW;
}

// TD loss of one step and its gradient (trainer.py:300-302: F.mse_loss(v(x_t), target.detach()), target = v(x_{t+1}) + extra):
// v = the value net's output on [next_state | state] ([2B]); grad[0..B) = 0 (the target half takes part in the forward only),
// grad[B + i] = 2 (v[B+i] - target_i) / B; logs = (loss, mean v(x_t), mean running cost).  extra: DEVICE scalar (time-cost terms
// of this step).  One workgroup, fixed order.
__global__ __launch_bounds__(256) void td_loss_kernel(const float* __restrict__ v, const float* __restrict__ cost,
                                                     const float* __restrict__ extra, float* __restrict__ grad,
                                                     float* __restrict__ logs, int B) {
    __shared__ float red[3][256];
    const int tid = threadIdx.x;
    const float ex = extra ? extra[0] : 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = tid; i < B; i += 256) {
        const float target = v[i] + ex;
        const float pred = v[B + i];
        const float d = pred - target;
        grad[i] = 0.f;
        grad[B + i] = 2.f * d / (float)B;
        a0 += d * d;
        a1 += pred;
        a2 += cost[i];
    }
    red[0][tid] = a0; red[1][tid] = a1; red[2][tid] = a2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
            red[2][tid] += red[2][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        logs[0] = red[0][0] / (float)B;
        logs[1] = red[1][0] / (float)B;
        logs[2] = red[2][0] / (float)B;
    }
}

// nearest-neighbour x2 upsample, NHWC bf16, 8 channels per thread (ResBlock up: x_upd, models/cm/unet.py:197-198)
__global__ void upsample2x_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int N, int H, int W, int C) {
    const int C8 = C / 8, OH = 2 * H, OW = 2 * W;
    const long total = (long)N * OH * OW * C8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c8 = idx % C8;
        long r = idx / C8;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH;
        const int n = (int)(r / OH);
        *reinterpret_cast<bf16x8*>(out + idx * 8) =
            *reinterpret_cast<const bf16x8*>(in + (((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + c8 * 8);
    }
}

// EDM preconditioning input: x_in = x / sqrt(sigma^2 + sd^2), rescaled_t = 250*ln(sigma + 1e-44)
// (KarrasDenoiser.get_scalings / denoise, models/cm/karras_diffusion.py:64-68, :348-349)
__global__ void edm_precond_kernel(const float* __restrict__ x, const float* __restrict__ sigma, float* __restrict__ x_in,
                                   float* __restrict__ t_out, int CHW, float sd) {
    const int b = blockIdx.x;
    const float s = sigma[b];
    const float c_in = 1.f / powf(s * s + sd * sd, 0.5f);
    if (threadIdx.x == 0) t_out[b] = 1000.f * 0.25f * logf(s + 1e-44f);
    const size_t base = (size_t)b * CHW;
    for (int i = threadIdx.x * 4; i < CHW; i += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = c_in * v[e];
        *reinterpret_cast<f32x4*>(x_in + base + i) = o;
    }
}

// Fused Euler-ancestral EDM transition (models/DxMI/openai_diffusion.py:71-94 with
// karras_diffusion.py:64-68,350): denoised = c_out*F + c_skip*x ; d = (x - denoised)/sigma ;
// mu = x + d*(sigma_down - sigma) ; x' = mu + z*sigma_up.
__global__ void edm_step_kernel(const float* __restrict__ x, const float* __restrict__ fout, const float* __restrict__ z,
                                const float* __restrict__ sigma, const float* __restrict__ sigma_down,
                                const float* __restrict__ sigma_up, float* __restrict__ sample, float* __restrict__ mean,
                                int CHW, float sd) {
    const int b = blockIdx.x;
    const float s = sigma[b], sdn = sigma_down[b], sup = sigma_up[b];
    const float den = s * s + sd * sd;
    const float c_skip = sd * sd / den;
    const float c_out = s * sd / powf(den, 0.5f);
    const float dt = sdn - s;
    const size_t base = (size_t)b * CHW;
    for (int i = threadIdx.x * 4; i < CHW; i += 256 * 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
        const f32x4 fv = *reinterpret_cast<const f32x4*>(fout + base + i);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + base + i);
        f32x4 mu, sm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float den_ = c_out * fv[e] + c_skip * xv[e];
            const float d = (xv[e] - den_) / s;
            mu[e] = xv[e] + d * dt;
            sm[e] = mu[e] + zv[e] * sup;
        }
        *reinterpret_cast<f32x4*>(mean + base + i) = mu;
        *reinterpret_cast<f32x4*>(sample + base + i) = sm;
    }
}

// Backward of the two sampler transitions (the differentiable `sample_step` of the policy update: reference var_sampler.py:357-408,
// openai_diffusion.py:71-94 under torch autograd).  One workgroup per sample: the incoming gradients are read once, the gradient of
// the network output written once, the per-sample sigma gradient reduced in-kernel (fp32, fixed order).  Any of g_* may be NULL.
//   VAR:  x' = xm x + c eps + sigma z ; mean = xm x + c eps ; control = c eps ; logp = mean_CHW(-(x'.detach() - mean)^2 / (2 sigma^2)) - log sigma - log sqrt(2 pi)
//         d eps = c (g_x' + g_mean + g_control + g_logp z / (sigma CHW))          (c = theta multiplier x adhoc_scale1)
//         d sigma = sum_CHW g_x' z + g_logp (mean_CHW z^2 - 1) / sigma
//   EDM:  mu = x + (x - c_out F - c_skip x) / sigma (sigma_down - sigma) ; x' = mu + z sigma_up
//         d F = -(c_out (sigma_down - sigma) / sigma) (g_x' + g_mu) ;  d sigma_up = sum_CHW g_x' z
__global__ __launch_bounds__(256) void step_bwd_kernel(const float* __restrict__ g_next, const float* __restrict__ g_mean,
                                                      const float* __restrict__ g_control, const float* __restrict__ g_logp,
                                                      const float* __restrict__ z, const float* __restrict__ cmul,
                                                      const float* __restrict__ sigma, const float* __restrict__ sigma_down, float sd,
                                                      int edm, float* __restrict__ d_eps, float* __restrict__ d_sigma, int CHW) {
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    float coef, lp_mean = 0.f, lp_sig = 0.f;
    if (edm) {
        const float s = sigma[b];
        const float c_out = s * sd / powf(s * s + sd * sd, 0.5f);
        coef = -(c_out * (sigma_down[b] - s) / s);
    } else {
        coef = cmul[b];
        if (g_logp) {
            lp_mean = g_logp[b] / (sigma[b] * (float)CHW);
            lp_sig = g_logp[b] / sigma[b];
        }
    }
    const size_t base = (size_t)b * CHW;
    float acc = 0.f, zz = 0.f;
    for (int i = threadIdx.x * 4; i < CHW; i += 256 * 4) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + base + i);
        f32x4 gn = {0.f, 0.f, 0.f, 0.f}, gm = gn, gc = gn, de;
        if (g_next) gn = *reinterpret_cast<const f32x4*>(g_next + base + i);
        if (g_mean) gm = *reinterpret_cast<const f32x4*>(g_mean + base + i);
        if (g_control) gc = *reinterpret_cast<const f32x4*>(g_control + base + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            de[e] = coef * (gn[e] + gm[e] + gc[e] + lp_mean * zv[e]);
            acc += gn[e] * zv[e];
            zz += zv[e] * zv[e];
        }
        *reinterpret_cast<f32x4*>(d_eps + base + i) = de;
    }
    acc = wave_sum(acc);
    zz = wave_sum(zz);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc; red[1][threadIdx.x >> 6] = zz; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const float q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        d_sigma[b] = a + lp_sig * (q / (float)CHW - 1.f);
    }
}

// im2col of a 3-channel NCHW fp32 image for the 3x3/s1/p1 stem convs: out[n,y,x,k] (64 wide, bf16),
// k = ci*9 + ky*3 + kx for k < 27, zero otherwise.  Feeds the 1x1 weight-gradient GEMM of the stem.
__global__ void im2col27_kernel(const float* __restrict__ x, bf16* __restrict__ out, int N, int H, int W) {
    const long total = (long)N * H * W * 8;  // 8 pieces of 8 channels per pixel
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int pc = idx & 7;
        long r = idx >> 3;
        const int xx = r % W; r /= W;
        const int yy = r % H;
        const int n = (int)(r / H);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = pc * 8 + e;
            float f = 0.f;
            if (k < 27) {
                const int ci = k / 9, ky = (k % 9) / 3, kx = k % 3;
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) f = x[(((long)n * 3 + ci) * H + iy) * W + ix];
            }
            v[e] = (bf16)f;
        }
        *reinterpret_cast<bf16x8*>(out + idx * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, bf16* __restrict__ out, int N, int C, int HW) {
    const long total = (long)N * C * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = idx % C;
        const long r = idx / C;
        const int px = r % HW;
        const int n = (int)(r / HW);
        out[idx] = (bf16)in[((size_t)n * C + c) * HW + px];
    }
}
__global__ void nhwc_to_nchw_kernel(const bf16* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const long total = (long)N * C * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int px = idx % HW;
        const long r = idx / HW;
        const int c = r % C;
        const int n = (int)(r / C);
        out[idx] = (float)in[((size_t)n * HW + px) * C + c];
    }
}

inline unsigned grid_for(long total, int threads) {
    long b = (total + threads - 1) / threads;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}


// ---------------------------------------------------------------------------------------------
// Output stage of the generate scripts: NCHW fp32 sampler output -> uint8 pixels, the reference's two quantisers with its
// fp32 operation order (every intermediate rounded to fp32 where torch stores one: no fused multiply-add across them):
//   mode 0  generate_cifar10.py:41-42,205-209 / generate_large.py:36-41 through torchvision.utils.save_image:
//           v = (x - (-1)) / 2 ; clamp(0, 1) ; v * 255 ; + 0.5 ; clamp(0, 255) ; truncate
//   mode 1  generate_large.py:43 (the FID / samples_N.npz array): (x + 1) * 127.5 ; clamp(0, 255) ; truncate
// Four pixels per thread: one 16-byte load per channel plane, 12 output bytes (NHWC, C = 3) or 4 bytes per plane (NCHW).
__device__ __forceinline__ uint32_t quant_u8(float x, int mode) {
    float v;
    if (mode == 0) {
        v = __fmul_rn(__fsub_rn(x, -1.f), 0.5f);                // (x - (-1)) / 2: the division by two is exact
        v = fminf(fmaxf(v, 0.f), 1.f);
        v = __fadd_rn(__fmul_rn(v, 255.f), 0.5f);
    } else {
        v = __fmul_rn(__fadd_rn(x, 1.f), 127.5f);
    }
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (uint32_t)v;                                         // truncation, as tensor.to(torch.uint8)
}

__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int N, int C, int HW,
                                                         int mode, int nhwc) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;       // quad of pixels
    const int qpi = HW >> 2;
    if (q >= (long)N * qpi) return;
    const int n = (int)(q / qpi), p0 = (int)(q - (long)n * qpi) * 4;
    if (nhwc && C == 3) {
        uint32_t b[12];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)n * 3 + c) * HW + p0);
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e * 3 + c] = quant_u8(v[e], mode);
        }
        u32x4 w;      // 12 bytes = 3 dwords
        uint32_t* o = reinterpret_cast<uint32_t*>(out + ((size_t)n * HW + p0) * 3);
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = b[4 * d] | (b[4 * d + 1] << 8) | (b[4 * d + 2] << 16) | (b[4 * d + 3] << 24);
        (void)w;
    } else {
        for (int c = 0; c < C; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)n * C + c) * HW + p0);
            if (nhwc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) out[((size_t)n * HW + p0 + e) * C + c] = (uint8_t)quant_u8(v[e], mode);
            } else {
                const uint32_t w = quant_u8(v[0], mode) | (quant_u8(v[1], mode) << 8) | (quant_u8(v[2], mode) << 16) | (quant_u8(v[3], mode) << 24);
                *reinterpret_cast<uint32_t*>(out + ((size_t)n * C + c) * HW + p0) = w;
            }
        }
    }
}

}  // namespace

extern "C" int dxmi_timestep_embedding(const float* t, float* out, int32_t N, int32_t dim, int32_t order,
                                       float max_period, void* stream) {
    DXMI_CHECK_ARG(t && out && N > 0 && dim >= 4, "dxmi_timestep_embedding: bad arguments");
    DXMI_CHECK_ARG(dim % 2 == 0, "dxmi_timestep_embedding: odd dim %d (the zero-padded last column of unet_small.py:25-26 is not implemented)", dim);
    const int total = N * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out,
                       N, dim, order, logf(max_period));
    DXMI_CHECK_LAUNCH("dxmi_timestep_embedding");
    return DXMI_OK;
}

extern "C" int dxmi_var_gather_sched(const int64_t* t, const float* continuous_steps, const float* xmul_tab,
                                     const float* cmul_tab, const float* log_betas_all, float* tau, float* xmul,
                                     float* cmul, float* sigma, int32_t N, int32_t T, void* stream) {
    DXMI_CHECK_ARG(t && continuous_steps && xmul_tab && cmul_tab && log_betas_all && tau && xmul && cmul && sigma,
                   "dxmi_var_gather_sched: null pointer");
    hipLaunchKernelGGL(var_gather_sched_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, t,
                       continuous_steps, xmul_tab, cmul_tab, log_betas_all, tau, xmul, cmul, sigma, N, T);
    DXMI_CHECK_LAUNCH("dxmi_var_gather_sched");
    return DXMI_OK;
}

extern "C" int dxmi_var_step_fwd(const float* x, const float* eps, const float* z, const float* xmul,
                                 const float* cmul, const float* sigma, float* x_next, float* mean, float* control,
                                 float* logp, int32_t N, int32_t CHW, int32_t assoc, void* stream) {
    DXMI_CHECK_ARG(x && eps && z && xmul && cmul && sigma && x_next, "dxmi_var_step_fwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && CHW > 0 && CHW % 4 == 0, "dxmi_var_step_fwd: CHW (%d) must be a multiple of 4", CHW);
    hipLaunchKernelGGL(var_step_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, eps, z, xmul, cmul, sigma,
                       x_next, mean, control, logp, CHW, assoc);
    DXMI_CHECK_LAUNCH("dxmi_var_step_fwd");
    return DXMI_OK;
}

extern "C" int dxmi_pool_act(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pool,
                             int32_t act, void* stream) {
    DXMI_CHECK_ARG(in && out && C % 8 == 0 && (!pool || (H % 2 == 0 && W % 2 == 0)), "dxmi_pool_act: bad arguments");
    const long total = (long)N * (pool ? H / 2 : H) * (pool ? W / 2 : W) * (C / 8);
    hipLaunchKernelGGL(pool_act_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)in,
                       (bf16*)out, N, H, W, C, pool, act);
    DXMI_CHECK_LAUNCH("dxmi_pool_act");
    return DXMI_OK;
}

extern "C" int dxmi_pool_act_bwd(const void* dout, const void* act_out, void* din, int32_t N, int32_t H, int32_t W,
                                 int32_t C, int32_t pool, float slope, void* stream) {
    DXMI_CHECK_ARG(dout && act_out && din && C % 8 == 0 && (!pool || (H % 2 == 0 && W % 2 == 0)), "dxmi_pool_act_bwd: bad arguments");
    const long total = (long)N * H * W * (C / 8);
    hipLaunchKernelGGL(pool_act_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dout,
                       (const bf16*)act_out, (bf16*)din, N, H, W, C, pool, slope);
    DXMI_CHECK_LAUNCH("dxmi_pool_act_bwd");
    return DXMI_OK;
}

extern "C" int dxmi_value_head_bwd(const void* feat, const float* w, const float* dy, void* dfeat, float* s, int32_t N,
                                   int32_t HW, int32_t C, void* stream) {
    DXMI_CHECK_ARG(feat && w && dy && dfeat && s, "dxmi_value_head_bwd: null pointer");
    hipLaunchKernelGGL(value_head_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16*)feat, w, dy, (bf16*)dfeat,
                       s, HW, C);
    DXMI_CHECK_LAUNCH("dxmi_value_head_bwd");
    return DXMI_OK;
}

extern "C" int dxmi_value_head_pgrad(const float* s, const float* w, const float* b, const float* dy, const float* out_w,
                                     float* out, int32_t N, int32_t C, void* stream) {
    DXMI_CHECK_ARG(s && w && b && dy && out && N > 0 && C > 0, "dxmi_value_head_pgrad: bad arguments");
    hipLaunchKernelGGL(value_head_pgrad_kernel, dim3((C + 63) / 64 + 1), dim3(256), 0, (hipStream_t)stream, s, w, b, dy, out_w, out, N, C);
    DXMI_CHECK_LAUNCH("dxmi_value_head_pgrad");
    return DXMI_OK;
}

extern "C" int dxmi_td_gather_cost(const float* traj, const int64_t* state_rows, const int64_t* next_rows, const float* next_dense,
                                   const float* beta, float* out_next, float* out_state, float* cost, int32_t B, int32_t CHW,
                                   int64_t n_src_rows, void* stream) {
    DXMI_CHECK_ARG(traj && state_rows && beta && out_state && cost && (next_rows || next_dense), "dxmi_td_gather_cost: null pointer");
    DXMI_CHECK_ARG(B > 0 && CHW > 0 && CHW % 4 == 0 && n_src_rows > 0, "dxmi_td_gather_cost: B=%d CHW=%d (multiple of 4)", B, CHW);
    hipLaunchKernelGGL(td_gather_cost_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, traj, state_rows, next_rows, next_dense, beta,
                       out_next, out_state, cost, CHW, n_src_rows);
    DXMI_CHECK_LAUNCH("dxmi_td_gather_cost");
    return DXMI_OK;
}

extern "C" int dxmi_td_loss(const float* v, const float* cost, const float* extra, float* grad, float* logs3, int32_t B,
                            void* stream) {
    DXMI_CHECK_ARG(v && cost && grad && logs3 && B > 0, "dxmi_td_loss: bad arguments");
    hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, v, cost, extra, grad, logs3, B);
    DXMI_CHECK_LAUNCH("dxmi_td_loss");
    return DXMI_OK;
}

extern "C" int dxmi_value_head(const void* in, const float* w, const float* b, const float* out_w, const float* out_b,
                               float* out, int32_t N, int32_t HW, int32_t C, void* stream) {
    DXMI_CHECK_ARG(in && w && b && out && (!out_w == !out_b), "dxmi_value_head: null pointer");
    DXMI_CHECK_ARG(C > 0 && C <= 4096, "dxmi_value_head: C=%d (at most 4096 channels)", C);
    hipLaunchKernelGGL(value_head_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, w, b, out_w,
                       out_b, out, HW, C);
    DXMI_CHECK_LAUNCH("dxmi_value_head");
    return DXMI_OK;
}

extern "C" int dxmi_upsample2x(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    DXMI_CHECK_ARG(in && out && C % 8 == 0, "dxmi_upsample2x: bad arguments");
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for((long)N * 4 * H * W * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)in, (bf16*)out, N, H, W, C);
    DXMI_CHECK_LAUNCH("dxmi_upsample2x");
    return DXMI_OK;
}

extern "C" int dxmi_edm_precond(const float* x, const float* sigma, float* x_in, float* t_out, int32_t N, int32_t CHW,
                                float sigma_data, void* stream) {
    DXMI_CHECK_ARG(x && sigma && x_in && t_out && CHW % 4 == 0, "dxmi_edm_precond: bad arguments");
    hipLaunchKernelGGL(edm_precond_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, sigma, x_in, t_out, CHW, sigma_data);
    DXMI_CHECK_LAUNCH("dxmi_edm_precond");
    return DXMI_OK;
}

extern "C" int dxmi_edm_step_fwd(const float* x, const float* model_out, const float* z, const float* sigma,
                                 const float* sigma_down, const float* sigma_up, float* sample, float* mean, int32_t N,
                                 int32_t CHW, float sigma_data, void* stream) {
    DXMI_CHECK_ARG(x && model_out && z && sigma && sigma_down && sigma_up && sample && mean && CHW % 4 == 0,
                   "dxmi_edm_step_fwd: bad arguments");
    hipLaunchKernelGGL(edm_step_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, model_out, z, sigma, sigma_down, sigma_up,
                       sample, mean, CHW, sigma_data);
    DXMI_CHECK_LAUNCH("dxmi_edm_step_fwd");
    return DXMI_OK;
}

extern "C" int dxmi_var_step_bwd(const float* g_next, const float* g_mean, const float* g_control, const float* g_logp,
                                 const float* z, const float* cmul, const float* sigma, float* d_eps, float* d_sigma, int32_t N,
                                 int32_t CHW, void* stream) {
    DXMI_CHECK_ARG(z && cmul && sigma && d_eps && d_sigma, "dxmi_var_step_bwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && CHW > 0 && CHW % 4 == 0, "dxmi_var_step_bwd: CHW (%d) must be a multiple of 4", CHW);
    hipLaunchKernelGGL(step_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, g_next, g_mean, g_control, g_logp, z, cmul, sigma,
                       (const float*)nullptr, 0.f, 0, d_eps, d_sigma, CHW);
    DXMI_CHECK_LAUNCH("dxmi_var_step_bwd");
    return DXMI_OK;
}

extern "C" int dxmi_edm_step_bwd(const float* g_sample, const float* g_mean, const float* z, const float* sigma,
                                 const float* sigma_down, float* d_model_out, float* d_sigma_up, int32_t N, int32_t CHW,
                                 float sigma_data, void* stream) {
    DXMI_CHECK_ARG(z && sigma && sigma_down && d_model_out && d_sigma_up, "dxmi_edm_step_bwd: null pointer");
    DXMI_CHECK_ARG(N > 0 && CHW > 0 && CHW % 4 == 0, "dxmi_edm_step_bwd: CHW (%d) must be a multiple of 4", CHW);
    hipLaunchKernelGGL(step_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, g_sample, g_mean, (const float*)nullptr,
                       (const float*)nullptr, z, (const float*)nullptr, sigma, sigma_down, sigma_data, 1, d_model_out, d_sigma_up, CHW);
    DXMI_CHECK_LAUNCH("dxmi_edm_step_bwd");
    return DXMI_OK;
}

extern "C" int dxmi_quantize_u8(const float* x, void* out, int32_t N, int32_t C, int32_t HW, int32_t mode, int32_t out_nhwc,
                                void* stream) {
    DXMI_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "dxmi_quantize_u8: null pointer / empty shape");
    DXMI_CHECK_ARG(HW % 4 == 0 && (mode == 0 || mode == 1), "dxmi_quantize_u8: HW (%d) must be a multiple of 4, mode 0 or 1", HW);
    hipLaunchKernelGGL(quantize_u8_kernel, dim3(grid_for((long)N * (HW / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, (uint8_t*)out, N, C,
                       HW, mode, out_nhwc);
    DXMI_CHECK_LAUNCH("dxmi_quantize_u8");
    return DXMI_OK;
}

extern "C" int dxmi_im2col27(const float* x, void* out, int32_t N, int32_t H, int32_t W, void* stream) {
    DXMI_CHECK_ARG(x && out, "dxmi_im2col27: null pointer");
    hipLaunchKernelGGL(im2col27_kernel, dim3(grid_for((long)N * H * W * 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)out,
                       N, H, W);
    DXMI_CHECK_LAUNCH("dxmi_im2col27");
    return DXMI_OK;
}

extern "C" int dxmi_nchw_f32_to_nhwc_bf16(const float* in, void* out, int32_t N, int32_t C, int32_t HW, void* stream) {
    DXMI_CHECK_ARG(in && out, "dxmi_nchw_f32_to_nhwc_bf16: null pointer");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)N * C * HW, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       (bf16*)out, N, C, HW);
    DXMI_CHECK_LAUNCH("dxmi_nchw_f32_to_nhwc_bf16");
    return DXMI_OK;
}

extern "C" int dxmi_nhwc_bf16_to_nchw_f32(const void* in, float* out, int32_t N, int32_t C, int32_t HW, void* stream) {
    DXMI_CHECK_ARG(in && out, "dxmi_nhwc_bf16_to_nchw_f32: null pointer");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long)N * C * HW, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)in, out, N, C, HW);
    DXMI_CHECK_LAUNCH("dxmi_nhwc_bf16_to_nchw_f32");
    return DXMI_OK;
}
