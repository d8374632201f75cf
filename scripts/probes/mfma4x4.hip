// Probe: v_mfma_f32_4x4x1_16b_f32 on gfx950 -- operand layout with the cbsz/abid broadcast, issue rate,
// and what it tolerates between issues.
//   hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float acc_t;

// ---- semantics: D[r] at lane l for A = a[l], B = b[l], cbsz/abid given at compile time -----------------
template <int CBSZ, int ABID>
__global__ void sem_kernel(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  acc_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, CBSZ, ABID, 0);
  for (int r = 0; r < 4; ++r) d[r * 64 + l] = c[r];
}

// ---- throughput: NACC independent accumulators, ITER rounds, W waves per SIMD by launch bounds ---------
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, const float* src) {
  acc_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = acc_t{0, 0, 0, 0};
  float a = src[threadIdx.x + 4096 + 64 * (blockIdx.x & 7)], b = src[threadIdx.x + 8192 + 64 * (blockIdx.x & 15)];   // random mantissas
  extern __shared__ float sm[];
  sm[threadIdx.x] = a;
  __syncthreads();
  float extra = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if (MODE == 2) {   // one LDS read per 4 MFMAs feeding B
        if ((i & 3) == 0) b = sm[(threadIdx.x + it + i) & 255];
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
      }
      if (MODE == 3) {   // one global load per 9 MFMAs (L2 resident)
        if ((i % 9) == 0) extra += src[(threadIdx.x + 64 * (it + i)) & 4095];
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
      }
    }
  }
  float s = extra;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int NACC, int MODE>
void run_rate(const char* name, int wgs_per_cu, float* out, const float* src) {
  const int iters = 20000 / NACC * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * wgs_per_cu), block(256);
  const int REP = 40;
  for (int r = 0; r < REP; ++r) hipLaunchKernelGGL((rate_kernel<NACC, MODE>), grid, block, 1024, 0, out, iters, src);
  hipEventRecord(e0);
  for (int r = 0; r < REP; ++r) hipLaunchKernelGGL((rate_kernel<NACC, MODE>), grid, block, 1024, 0, out, iters, src);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= REP;
  const double n_mfma = (double)iters * NACC;                 // per wave
  const double flop_per = (MODE == 1) ? 2048.0 : 512.0;
  const double tf = n_mfma * flop_per * 4 * 256 * wgs_per_cu / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD at 2.4 GHz (upper bound on the clock)
  const double cyc = ms * 1e-3 * 2.4e9 / (n_mfma * wgs_per_cu);
  printf("%-44s nacc=%2d waves/SIMD=%d: %.3f ms  %.1f TF  <=%.1f cyc/MFMA/SIMD @2.4GHz\n", name, NACC, wgs_per_cu, ms, tf, cyc);
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int i = 0; i < 64; ++i) { ha[i] = 1 + i; hb[i] = 100 * (1 + i); }
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  auto check = [&](const char* name, auto kern, int cbsz, int abid) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: D[r][lane 4blk + j] = a[4*src + r] * b[4blk + j], src = (cbsz ? abid : blk)
    int bad = 0;
    for (int r = 0; r < 4; ++r)
      for (int l = 0; l < 64; ++l) {
        int blk = l / 4;
        int src = cbsz == 4 ? abid : (cbsz == 0 ? blk : -1);
        if (src < 0) continue;
        float want = ha[4 * src + r] * hb[l];
        if (hd[r * 64 + l] != want) ++bad;
      }
    printf("semantics %s cbsz=%d abid=%d: mismatches vs hypothesis = %d   (D[0][0..7] = %g %g %g %g %g %g %g %g; D[1][0]=%g D[2][5]=%g)\n",
           name, cbsz, abid, bad, hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[64], hd[128 + 5]);
  };
  check("4x4x1", sem_kernel<0, 0>, 0, 0);
  check("4x4x1", sem_kernel<4, 0>, 4, 0);
  check("4x4x1", sem_kernel<4, 5>, 4, 5);
  check("4x4x1", sem_kernel<4, 15>, 4, 15);

  float *out, *src;
  hipMalloc(&out, 4096); hipMalloc(&src, 65536);
  {
    std::vector<float> hs(16384);
    uint32_t st = 12345;
    for (auto& x : hs) { st = st * 1664525u + 1013904223u; x = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1e-3f; }
    if (getenv("ZERO")) for (auto& x : hs) x = 0.f;
    hipMemcpy(src, hs.data(), 65536, hipMemcpyHostToDevice);
  }
  run_rate<9, 0>("4x4x1_16b cbsz=4, 9 independent acc", 1, out, src);
  run_rate<9, 0>("4x4x1_16b cbsz=4, 9 independent acc", 2, out, src);
  run_rate<36, 0>("4x4x1_16b cbsz=4, 36 independent acc", 1, out, src);
  run_rate<2, 0>("4x4x1_16b cbsz=4, 2 independent acc", 1, out, src);
  run_rate<1, 0>("4x4x1_16b cbsz=4, dependent chain", 1, out, src);
  run_rate<4, 1>("16x16x4, 4 acc", 1, out, src);
  run_rate<4, 1>("16x16x4, 4 acc", 2, out, src);
  run_rate<36, 2>("4x4x1 + 1 ds_read_b32 per 4 MFMA", 1, out, src);
  run_rate<36, 2>("4x4x1 + 1 ds_read_b32 per 4 MFMA", 2, out, src);
  run_rate<36, 3>("4x4x1 + 1 global_load per 9 MFMA", 1, out, src);
  return 0;
}
