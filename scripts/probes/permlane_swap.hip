#include <hip/hip_runtime.h>
__global__ void k(float* o, const float* a, const float* b) {
  float x = a[threadIdx.x], y = b[threadIdx.x];
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  auto q = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false);
  o[threadIdx.x] = __uint_as_float(q[0]);
  o[64 + threadIdx.x] = __uint_as_float(q[1]);
}
int main() {
  float *a, *b, *o; float ha[64], hb[64], ho[128];
  for (int i = 0; i < 64; ++i) { ha[i] = i; hb[i] = 100 + i; }
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 512);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, a, b);
  hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; ++r) { for (int i = 0; i < 64; i += 16) printf("[%g..%g] ", ho[r*64+i], ho[r*64+i+15]); printf("\n"); }
  return 0;
}
