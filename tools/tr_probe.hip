// Probe of ds_read_b64_tr_b16 lane semantics on gfx950 (build: hipcc --offload-arch=gfx950 tools/tr_probe.hip -o gpurun_out/tr_probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;  // value = row*64 + col
  __syncthreads();
  const int l = threadIdx.x;
  const int grp = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const short* addr = lds + (grp * 4 + q) * 64 + 4 * p;   // row (4*grp+q), cols 4p..4p+3
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
    printf("\n");
  }
  return 0;
}
