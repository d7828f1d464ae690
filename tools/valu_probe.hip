// GPU box probe (round 5): can a mover wave run the GroupNorm(+SiLU) transform of a conv's input (bf16 -> fma -> SiLU -> bf16)
// beside an MFMA wave on the same SIMD, and what does each side pay?  One workgroup per CU: waves 0-3 issue back-to-back
// v_mfma_f32_16x16x32_bf16 (register operands: the matrix pipe's peak), waves 4-7 (one per SIMD) run the transform on 16-byte
// pieces held in registers:
//   mode 0  no VALU waves                  (MFMA alone)
//   mode 1  affine + SiLU, flat out        (2 transcendentals per element)
//   mode 2  affine only, flat out
//   mode 3  affine + SiLU, `work` pieces then s_sleep `nap` (a duty cycle like the conv's: ~2 k of 4.8 k cycles per chunk)
// Printed: cycles per MFMA (16 = pipe full), transform cycles per 16-byte piece and wave.
//      hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_probe tools/valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <bool SILU>
__device__ __forceinline__ u32x4 transform_piece(u32x4 raw, const f32x2 (&A)[4], const f32x2 (&B)[4]) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 x = {__builtin_bit_cast(float, raw[i] << 16), __builtin_bit_cast(float, raw[i] & 0xffff0000u)};
        f32x2 y = x * A[i] + B[i];                               // v_pk_fma_f32
        if (SILU) {                                              // y * rcp(1 + exp2(-y log2 e)): dxmi_silu_fast, packed where the ISA has it
            const f32x2 t = y * f32x2{-1.4426950408889634f, -1.4426950408889634f};
            f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            e = e + f32x2{1.f, 1.f};
            const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
            y = y * r;
        }
        bf16x2 rr = {(bf16)y[0], (bf16)y[1]};
        o[i] = __builtin_bit_cast(unsigned int, rr);
    }
    return o;
}

__global__ __launch_bounds__(512) void probe(int mode, int mfma_iters, int work, int nap, int prio, int lds_ops, unsigned long long* out, float* sink) {
    __shared__ int done;
    __shared__ __attribute__((aligned(16))) char opnd[32768];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave < 4) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (bf16)(0.01f * (lane + e)); b[e] = (bf16)(0.02f * (lane - e)); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (mode >= 10) {            // VALU waves alone: these waves only keep time (~1024 cycles per iteration)
            for (int it = 0; it < mfma_iters; ++it)
                for (int z = 0; z < 2; ++z) __builtin_amdgcn_s_sleep(8);
        } else if (lds_ops) {
            // conv_ws_kernel's half-step pattern: 16 MFMAs on (A[4], B[4]) while the next 4 + 4 operands are read from LDS
            for (int i = threadIdx.x; i < 2048; i += 256) reinterpret_cast<u32x4*>(opnd)[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const char* const ab = opnd + lane * 16;
            bf16x8 A0[4], A1[4], B0[4], B1[4];
            f32x4 ac[4][4];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ac[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < 4; ++i) { A0[i] = *reinterpret_cast<const bf16x8*>(ab + i * 1024); B0[i] = *reinterpret_cast<const bf16x8*>(ab + 4096 + i * 1024); }
            for (int it = 0; it < mfma_iters * 2; ++it) {
#pragma unroll
                for (int hs = 0; hs < 2; ++hs) {
                    const int off = ((it * 2 + hs) & 3) * 8192;
                    if (hs == 0) {
                        for (int i = 0; i < 4; ++i) { A1[i] = *reinterpret_cast<const bf16x8*>(ab + off + i * 1024); B1[i] = *reinterpret_cast<const bf16x8*>(ab + off + 4096 + i * 1024); }
                        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0[j], B0[i], ac[i][j], 0, 0, 0);
                    } else {
                        for (int i = 0; i < 4; ++i) { A0[i] = *reinterpret_cast<const bf16x8*>(ab + off + i * 1024); B0[i] = *reinterpret_cast<const bf16x8*>(ab + off + 4096 + i * 1024); }
                        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[j], B1[i], ac[i][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    for (int z = 0; z < 8; ++z) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[0] += ac[i][j];
        } else
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) atomicAdd(&done, 1);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 1.2345f) sink[0] = s;
        if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
        return;
    }
    if (mode == 0) return;
    if (mode >= 10) mode -= 10;
    if (prio) __builtin_amdgcn_s_setprio(3);
    f32x2 A[4], B[4];
    for (int i = 0; i < 4; ++i) {
        A[i] = f32x2{0.9f + 0.01f * lane, 1.1f - 0.01f * i};
        B[i] = f32x2{0.05f * i, -0.03f * lane};
    }
    u32x4 r[8];
    for (int k = 0; k < 8; ++k) r[k] = u32x4{0x3f803f00u + lane + k, 0xbf803e80u + k, 0x3e003f80u + lane, 0x40003f00u - k};
    volatile int* const dn = &done;
    unsigned long long pieces = 0;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (;;) {
        for (int w = 0; w < work; ++w) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 o = mode == 2 ? transform_piece<false>(r[k], A, B) : transform_piece<true>(r[k], A, B);
                acc ^= o;
                r[k] += u32x4{0x10001u, 0x20003u, 0x10002u, 0x30001u};         // new data every round (nothing to hoist)
            }
        }
        pieces += 8ull * work;
        if (mode == 3) for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(8);
        if (*dn >= 4) break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc[0] == 0x12345u && acc[1] == 7u) sink[1] = 1.f;
    if (lane == 0 && blockIdx.x == 0) {
        out[4 + (wave - 4) * 2] = t1 - t0;
        out[5 + (wave - 4) * 2] = pieces;
    }
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 64);
    const int iters = 20000;        // x 64 MFMAs = 20.5 M cycles at 16 cycles each
    struct { int mode, work, nap, prio, lds; const char* what; } runs[] = {
        {0, 0, 0, 0, 0, "MFMA alone"},
        {1, 4, 0, 0, 0, "SiLU transform flat out"},
        {2, 4, 0, 0, 0, "affine only flat out"},
        {1, 4, 0, 1, 0, "SiLU flat out, s_setprio 3"},
        {2, 4, 0, 1, 0, "affine flat out, s_setprio 3"},
        {3, 1, 3, 1, 0, "SiLU 8 pieces + 3 naps, s_setprio 3"},
        {3, 1, 12, 1, 0, "SiLU 8 pieces + 12 naps, s_setprio 3"},
        {3, 1, 24, 1, 0, "SiLU 8 pieces + 24 naps, s_setprio 3"},
        {0, 0, 0, 0, 1, "MFMA + LDS operand reads alone"},
        {1, 4, 0, 0, 1, "LDS-fed MFMA | SiLU flat out"},
        {1, 4, 0, 1, 1, "LDS-fed MFMA | SiLU flat out, prio 3"},
        {3, 1, 12, 0, 1, "LDS-fed MFMA | SiLU 8 + 12 naps"},
        {3, 1, 12, 1, 1, "LDS-fed MFMA | SiLU 8 + 12 naps, prio 3"},
        {3, 1, 24, 1, 1, "LDS-fed MFMA | SiLU 8 + 24 naps, prio 3"},
        {11, 4, 0, 0, 0, "SiLU transform, MFMA waves asleep"},
        {12, 4, 0, 0, 0, "affine only, MFMA waves asleep"},
    };
    for (auto& rn : runs) {
        hipMemset(out, 0, 16 * 8);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, rn.mode, iters, rn.work, rn.nap, rn.prio, rn.lds, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        const double cyc_mfma = (double)h[0] / (iters * 64.0);
        if (rn.mode < 10) printf("%-38s cycles per MFMA %.2f (pipe %.0f %% busy)", rn.what, cyc_mfma, 1600.0 / cyc_mfma);
        else printf("%-38s", rn.what);
        if (rn.mode) {
            const double cpp = (double)h[4] / (double)h[5];
            printf("  | transform: %.0f cycles per 16-byte piece and wave = %.2f cycles per element, VALU wave active %.0f %% of the MFMA time",
                   cpp, cpp / 8.0, 100.0 * (double)h[4] / (double)h[0]);
        }
        printf("\n");
    }
    return 0;
}
