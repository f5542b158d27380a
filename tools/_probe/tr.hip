#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 32];      // [px][ch]
  for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, m = l & 15, g = l >> 4;
  const int k0 = 8 * (g >> 1), r0 = 16 * (g & 1);
  lds_s4* p = (lds_s4*)(lds + (k0 + (m >> 2)) * 32 + r0 + 4 * (m & 3));
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = (8 * (l >> 5) + j) * 32 + (l & 31);
      if (h[l * 4 + j] != want) ++bad;
    }
  printf("tr probe: %d mismatches of 256\n", bad);
  for (int l = 0; l < 64; l += 9) printf("lane %2d: px,ch = (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4]/32, h[l*4]%32, h[l*4+1]/32, h[l*4+1]%32, h[l*4+2]/32, h[l*4+2]%32, h[l*4+3]/32, h[l*4+3]%32);
  return 0;
}
