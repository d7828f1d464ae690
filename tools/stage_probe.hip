// GPU box probe (round 4): what does it cost a MOVER wave to bring 1 KiB from L2 into LDS —
//   mode 0  LDS-DMA             global_load_lds_dwordx4, DEPTH pieces in flight
//   mode 1  register staging    BATCH x global_load_dwordx4 -> s_waitcnt -> BATCH x ds_write_b128, two register sets (the loads of
//                               batch i+1 are in flight while batch i is written)
// with `movers` waves moving and, optionally, four more waves of the workgroup issuing back-to-back MFMAs on the SIMDs the movers
// share (what conv_ws_kernel's movers live with).  One workgroup per CU, L2-resident source (every workgroup streams the same
// 144 KB round and round).      hipcc --offload-arch=gfx950 -O3 -o tools/bin/stage_probe tools/stage_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BATCH>
__global__ __launch_bounds__(768) void probe(const char* __restrict__ src, int region_pieces, int pieces_per_wave, int movers, int mfma_iters,
                                             float* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) *reinterpret_cast<int*>(lds + 159 * 1024) = 0;
    __syncthreads();
    if (wave >= movers) {
        // MFMA waves: 16x16x32 bf16 back to back on register operands
        if (mfma_iters == 0) return;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (lane + e)); b[e] = (__bf16)(0.02f * (lane - e)); }
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // run for as long as the movers do: they raise a flag in LDS when their last piece has landed
        volatile int* const done = reinterpret_cast<volatile int*>(lds + 159 * 1024);
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            if (*done >= movers) { mfma_iters = it + 1; break; }
        }
        if (lane == 0 && blockIdx.x == 0) sink[4 + wave - movers] = (float)mfma_iters;      // 64 MFMAs (1024 cycles) per iteration
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 1.2345f) sink[0] = s;
        return;
    }
    char* const slot = lds + wave * 16 * 1024;            // per-wave LDS window: 16 pieces of 1 KiB, reused
    int* const done = reinterpret_cast<int*>(lds + 159 * 1024);
    const char* const base = src + (size_t)lane * 16;
    int pi = wave;
    if constexpr (MODE == 0) {
        for (int i = 0; i < pieces_per_wave; ++i) {
            __builtin_amdgcn_global_load_lds(GPTR(base + (size_t)(pi % region_pieces) * 1024), LPTR(slot + (i & 15) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            pi += movers;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(done, 1);
    } else {
        u32x4 r0[BATCH], r1[BATCH];
        auto load = [&](u32x4 (&r)[BATCH]) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                r[k] = *reinterpret_cast<const u32x4*>(base + (size_t)(pi % region_pieces) * 1024);
                pi += movers;
            }
        };
        auto store = [&](const u32x4 (&r)[BATCH], int i) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) *reinterpret_cast<u32x4*>(slot + ((i * BATCH + k) & 15) * 1024 + lane * 16) = r[k];
        };
        const int nb = pieces_per_wave / BATCH;
        load(r0);
        for (int i = 0; i < nb; i += 2) {
            load(r1);
            store(r0, i);           // the compiler's vmcnt wait leaves r1's loads in flight
            load(r0);
            store(r1, i + 1);
        }
        if (r0[0].x == 0x12345678u) sink[1] = 1.f;
        if (lane == 0) atomicAdd(done, 1);
    }
}

template <int MODE, int BATCH>
void run(const char* d, float* sink, int movers, int with_mfma, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int ppw = 8192 / movers / 16 * 16;              // ~8 MiB per CU per launch
    auto k = probe<MODE, BATCH>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int threads = (movers + (with_mfma ? 4 : 0)) * 64;
    // MFMA waves run about as long as the movers (tuned by hand: 8 MFMAs = 128 cycles per iteration)
    const int iters = with_mfma ? 1 << 30 : 0;        // until the movers' flag
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 160 * 1024, 0, d, 144, ppw, movers, iters, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 160 * 1024, 0, d, 144, ppw, movers, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)ppw * movers * 1024;
    const double us = ms * 1e3 / reps;
    float h[8] = {0};
    hipMemcpy(h, sink, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s movers %2d %s: %8.1f us  %6.1f GB/s per CU = %5.1f B/clk @2.1GHz, %5.0f clk per KiB and wave", what, movers,
           with_mfma ? "+4 MFMA waves" : "alone        ", us, bytes_cu / us / 1e3, bytes_cu / us / 1e3 / 2.1, us * 2100.0 / ppw);
    if (with_mfma) printf("  | MFMA pipe busy %.2f (at 2.1 GHz)", h[4] * 1024.0 / (us * 2100.0));
    printf("\n");
}

int main() {
    char* d; float* sink;
    hipMalloc(&d, 1 << 24); hipMalloc(&sink, 64);
    hipMemset(d, 1, 1 << 24);
    for (int mf : {0, 1})
        for (int movers : {2, 4, 8}) {
            run<0, 8>(d, sink, movers, mf, "LDS-DMA depth 8");
            run<1, 4>(d, sink, movers, mf, "register staging batch 4");
            run<1, 8>(d, sink, movers, mf, "register staging batch 8");
        }
    return 0;
}
