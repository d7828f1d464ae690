// GPU box probe: how fast ONE CU pulls L2-resident data into LDS by DMA (`global_load_lds`, 1 KiB per wave instruction), as a
// function of the number of issuing waves and the pieces each keeps in flight.  One workgroup per CU; every workgroup streams
// the SAME `region` bytes round and round (L2-resident after the first pass), or its own slice of a large buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_probe tools/dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int DEPTH>
__device__ __forceinline__ void wait_depth() {
    if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
}

// SEG: contiguous bytes per row segment inside one wave instruction (1024: one contiguous KiB; 64: 16 rows x 64 B at pitch 512)
template <int DEPTH, int SEG, bool TOREG>
__global__ __launch_bounds__(1024) void pull(const char* __restrict__ src, size_t region, int pieces_per_wave, int own, int* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const char* base = src + (own ? (size_t)blockIdx.x * region : 0);
    constexpr int LPR = SEG / 16;
    const size_t lane_off = SEG == 1024 ? (size_t)lane * 16 : (size_t)(lane / LPR) * 512 + (lane % LPR) * 16;
    const size_t piece_span = SEG == 1024 ? 1024 : (1024 / SEG) * 512;
    const size_t npieces = region / piece_span;
    char* slot = lds + wave * 32 * 1024 / (nw > 4 ? 2 : 1);      // per-wave LDS window (32 or 16 pieces of 1 KiB, reused)
    int4 acc = {0, 0, 0, 0};
    size_t pi = wave;
    for (int i = 0; i < pieces_per_wave; ++i) {
        const char* g = base + (pi % npieces) * piece_span + lane_off;
        if constexpr (TOREG) {
            const int4 v = *reinterpret_cast<const int4*>(g);
            acc.x += v.x;
        } else {
            __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(slot + (i % (nw > 4 ? 16 : 32)) * 1024), 16, 0, 0);
            wait_depth<DEPTH>();
        }
        pi += nw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 0x12345678) sink[0] = 1;
}

template <int DEPTH, int SEG, bool TOREG>
void run(const char* d, int* sink, int waves, size_t region, int own, const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int ppw = 4096 / waves;                    // 4 MiB per CU per launch
    auto k = pull<DEPTH, SEG, TOREG>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, d, region, ppw, own, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 128 * 1024, 0, d, region, ppw, own, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)ppw * waves * 1024;
    const double us = ms * 1e3 / reps;
    printf("%-34s waves %2d depth %2d seg %4d: %8.1f us  %6.1f GB/s per CU  (%5.2f TB/s chip)\n", what, waves, DEPTH, SEG, us, bytes_cu / us / 1e3,
           bytes_cu * 256 / us / 1e6);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* d; int* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 1, bytes);
    for (int waves : {1, 2, 4, 8, 12}) {
        run<4, 1024, false>(d, sink, waves, 144 << 10, 0, "L2 shared 144 KB, DMA");
        run<8, 1024, false>(d, sink, waves, 144 << 10, 0, "L2 shared 144 KB, DMA");
        run<16, 1024, false>(d, sink, waves, 144 << 10, 0, "L2 shared 144 KB, DMA");
    }
    for (int waves : {4, 8}) {
        run<8, 64, false>(d, sink, waves, 2 << 20, 0, "L2 shared 2 MB, DMA 64-B segs");
        run<8, 1024, false>(d, sink, waves, 2 << 20, 0, "L2 shared 2 MB, DMA");
        run<8, 1024, false>(d, sink, waves, 4 << 20, 1, "own 4 MB slice (HBM), DMA");
        run<8, 1024, true>(d, sink, waves, 144 << 10, 0, "L2 shared 144 KB, to registers");
    }
    return 0;
}
