// Probe: where an item of rowq_kernel (csrc/rowq.hip) spends its time -- the LAST row of a corner sweep (S = 6^4, every leg 6,
// A[S][v1..v5] -> C[h][S][d1..d5]) with parts of the kernel switched off (RowArgs.pad2_ bits, -DQAMD_RQ_ABLATE) and with
// fewer items (one round of the chip).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=10000000 -DQAMD_RQ_ABLATE -I quimb_amd/csrc \
//         scripts/probes/rowq_probe.hip -o /tmp/rowq_probe && /tmp/rowq_probe
#include "../../quimb_amd/csrc/rowq.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

// calibration: N x 12 independent v_mfma_f32_4x4x1 per wave (nothing else), s_memtime ticks per wave and wall time
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int n, float a, float b) {
  qamdq::acc4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = qamdq::acc4{0, 0, 0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 3, 0);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (threadIdx.x % 64 == 0) { out[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = (float)(t1 - t0); out[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = s; }
}

static void calibrate() {
  float* d;
  hipMalloc(&d, 4096 * 8 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 20000;
  for (int grid : {1, 256, 768, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_rate_kernel, dim3(grid), dim3(256), 0, 0, d, n, 1.0f, 0.0f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    float h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double nm = 48.0 * n;     // MFMAs per wave
    printf("calibration: %4d workgroups of 4 waves: %8.1f us, %10.0f ticks per wave -> %.1f ticks/us, %.2f ticks per MFMA, %.2f ns per MFMA "
           "(4 waves per CU: %.1f TFLOP/s if every CU does this)\n", grid, 1000.0 * ms, h[0], h[0] / (1000.0 * ms), h[0] / nm,
           1e6 * ms / nm, 512.0 * nm * 4 * 256 / (1e-3 * ms) / 1e12 * (grid >= 256 ? grid / 256.0 : 1.0 / 256 * grid) );
  }
  hipFree(d);
}

int main(int argc, char** argv) {
  calibrate();
  const int64_t S = 1296, na = 7776 * S, nc = 7776 * S * 6;
  std::vector<float> hA(na), hW(1296);
  for (auto& x : hA) x = (float)rand() / RAND_MAX;
  for (auto& x : hW) x = (float)rand() / RAND_MAX - 0.3f;
  RowArgs p{};
  { int64_t st = 1; for (int i = 4; i >= 0; --i) { p.sv[i] = st; st *= 6; } }
  { int64_t st = 1; for (int i = 4; i >= 0; --i) { p.sd[i] = st; st *= 6; } p.sh = 7776 * S; }
  p.nS = 1; p.dimS[0] = (uint32_t)S; p.sSa[0] = 7776; p.sSc[0] = 7776;
  for (int c = 0; c < 5; ++c) {
    if (c == 0) { p.ws[c][0] = 36; p.ws[c][1] = 0; p.ws[c][2] = 6; p.ws[c][3] = 1; }
    else { p.ws[c][0] = 216; p.ws[c][1] = 36; p.ws[c][2] = 6; p.ws[c][3] = 1; }
    p.ed[c] = 6;
  }
  p.eh = 6;
  float *dA, *dC, *dW[5];
  hipMalloc(&dA, na * 4); hipMalloc(&dC, nc * 4);
  hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice);
  const void* W[5];
  for (int c = 0; c < 5; ++c) { hipMalloc(&dW[c], 1296 * 4); hipMemcpy(dW[c], hW.data(), 1296 * 4, hipMemcpyHostToDevice); W[c] = dW[c]; }
  {
    int nb = 0;
    const size_t lds = (size_t)(qamdq::D * qamdq::SB + 4 * qamdq::NF * 64 + 220) * sizeof(float);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)qamdq::rowq_kernel<true, true>, 256, lds);
    printf("occupancy: %d workgroups per CU at %zu bytes of LDS\n", nb, lds);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef QAMD_RQ_TIMING
  {
    float* dT;
    const int nw = 768 * 4;
    hipMalloc(&dT, nw * 16 * 4);
    hipMemset(dT, 0, nw * 16 * 4);
    p.items = (uint32_t)(S * 6);
    for (int ab : {59, 4, 63, 512, 1024, 0}) {
      p.pad2_ = ab;
      hipEvent_t t0, t1;
      hipEventCreate(&t0); hipEventCreate(&t1);
      for (int i = 0; i < 2; ++i) qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, dT, nullptr);
      hipEventRecord(t0);
      qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, dT, nullptr);
      hipEventRecord(t1);
      hipEventSynchronize(t1);
      float tms;
      hipEventElapsedTime(&tms, t0, t1);
      std::vector<float> hT(nw * 16);
      hipMemcpy(hT.data(), dT, nw * 16 * 4, hipMemcpyDeviceToHost);
      double whole = 0, wmax = 0;
      for (int b = 0; b < 768 * 4; ++b) { whole += hT[b * 16 + 9]; wmax = hT[b * 16 + 9] > wmax ? hT[b * 16 + 9] : wmax; }
      printf("ablate %3d: %.1f us wall; whole-kernel ticks per wave: mean %.0f, max %.0f -> %.0f ticks/us (max)\n", ab, 1000.0 * tms, whole / (768 * 4), wmax, wmax / (1000.0 * tms));
    }
    p.pad2_ = 0;
    for (int i = 0; i < 3; ++i) qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, dT, nullptr);
    hipDeviceSynchronize();
    hipMemset(dT, 0, nw * 16 * 4);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    hipEventRecord(t0);
    qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, dT, nullptr);
    hipEventRecord(t1);
    hipEventSynchronize(t1);
    float tms;
    hipEventElapsedTime(&tms, t0, t1);
    printf("one instrumented launch: %.1f us\n", 1000.0 * tms);
    std::vector<float> hT(nw * 16);
    hipMemcpy(hT.data(), dT, nw * 16 * 4, hipMemcpyDeviceToHost);
    const char* names[8] = {"site 1 (+ wait for prefetch)", "decode + prefetch issue", "barriers before sites", "LDS reads of sites", "MFMAs + LDS writes", "barrier after last site", "copy-out reads + barrier", "copy-out stores issue"};
    for (int wv = 0; wv < 4; ++wv) {
      double sum[8] = {0};
      for (int b = 0; b < 768; ++b) for (int i = 0; i < 8; ++i) sum[i] += hT[(b * 4 + wv) * 16 + i];
      double tot = 0; for (int i = 0; i < 8; ++i) tot += sum[i];
      printf("wave %d: total %.0f ticks per workgroup (s_memtime)\n", wv, tot / 768);
      for (int i = 0; i < 8; ++i) printf("    %-32s %9.0f  %5.1f %%\n", names[i], sum[i] / 768, 100 * sum[i] / tot);
      double pro = 0, whole = 0, emin = 1e30, emax = 0;
      for (int b = 0; b < 768; ++b) { pro += hT[(b * 4 + wv) * 16 + 8]; whole += hT[(b * 4 + wv) * 16 + 9]; double e = hT[(b * 4 + wv) * 16 + 10]; emin = e < emin ? e : emin; emax = e > emax ? e : emax; }
      printf("    prologue %.0f ticks, whole kernel %.0f ticks per wave; entry stamps spread over %.0f ticks\n", pro / 768, whole / 768, emax - emin);
    }
    return 0;
  }
#endif
  const int abl[] = {0, 512, 1024, 1024 | 512, 128, 128 | 512, 1, 2, 4, 8, 16, 28, 64, 1 | 2 | 8 | 16 | 32, 127 & ~64};   // 512: with the item queue
  for (uint32_t items : {(uint32_t)(S * 6), 768u, 256u}) {
    p.items = items;
    for (int ab : abl) {
      p.pad2_ = (uint32_t)ab;
      for (int i = 0; i < 3; ++i) qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, nullptr, nullptr);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 20;
      for (int i = 0; i < reps; ++i) qamd_rowq_launch(&p, dA, W, dC, nullptr, nullptr, nullptr, nullptr);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("items %5u ablate %3d: %8.1f us per launch\n", items, ab, 1000.0 * ms / reps);
    }
  }
  printf("(bits: 512 item queue instead of static shares, 1024 / 128 wave priorities by (workgroup / 8) %% 3 / (workgroup / 256) %% 3; 1 no result stores, 2 no loads of A, 4 no MFMAs in sites 2-5, 8 no LDS state reads, 16 no LDS state writes, 32 no gather of W, 64 return after site 1)\n");
  return 0;
}
