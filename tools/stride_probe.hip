// GPU box probe: read bandwidth of 16-byte-per-lane loads when a wave instruction covers
//   seg = 1024 contiguous bytes | 16 x 64-byte segments (512-byte pitch) | 8 x 128-byte segments (512-byte pitch)
// (the conv halo staging reads 64-byte segments: 32 channels of one pixel).  hipcc --offload-arch=gfx950 -O3 -o tools/bin/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int SEG>   // bytes of contiguous data per segment within a wave instruction
__global__ __launch_bounds__(256) void rd(const char* __restrict__ src, int* __restrict__ sink, size_t rows, int pitch) {
    // data viewed as rows of `pitch` bytes; a wave instruction reads SEG bytes from each of 1024/SEG consecutive rows
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    constexpr int RPI = 1024 / SEG;          // rows per instruction
    constexpr int LPR = SEG / 16;            // lanes per row
    const int r_in = lane / LPR, c_in = (lane % LPR) * 16;
    v4i acc = {0, 0, 0, 0};
    const int colsteps = pitch / SEG;
    for (size_t rb = wave * RPI; rb < rows; rb += nwaves * RPI) {
#pragma unroll 4
        for (int cs = 0; cs < colsteps; ++cs) {
            const v4i v = *reinterpret_cast<const v4i*>(src + (rb + r_in) * pitch + cs * SEG + c_in);
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = 1;
}

template <int SEG>
void run(const char* name, const char* d, int* sink, size_t bytes, int pitch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t rows = bytes / pitch;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rd<SEG>, dim3(2048), dim3(256), 0, 0, d, sink, rows, pitch);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(rd<SEG>, dim3(2048), dim3(256), 0, 0, d, sink, rows, pitch);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s pitch %4d: %7.1f us  %6.2f TB/s\n", name, pitch, ms * 1e3 / reps, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB: beyond the 256 MiB Infinity Cache
    char* d; int* sink;
    hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(d, 1, bytes);
    for (int pitch : {512, 1024}) {
        run<1024>("1 KiB contiguous / instr", d, sink, bytes, 1024 > pitch ? 1024 : pitch);
        run<128>("8 x 128 B segments / instr", d, sink, bytes, pitch);
        run<64>("16 x 64 B segments / instr", d, sink, bytes, pitch);
    }
    // the same 64-B pattern over a 64 MiB buffer (Infinity-Cache resident after the warm-up launches)
    run<1024>("IC-resident 1 KiB", d, sink, (size_t)64 << 20, 1024);
    run<64>("IC-resident 16 x 64 B", d, sink, (size_t)64 << 20, 512);
    return 0;
}
