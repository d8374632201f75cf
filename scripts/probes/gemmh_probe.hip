// gemmh_probe.hip -- standalone check + timing of the split-product join kernel (quimb_amd/csrc/gemmh.hip):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I quimb_amd/csrc scripts/probes/gemmh_probe.hip -o gpurun_out/gemmh_probe
//   gpurun_out/gemmh_probe [M N K ta tb iters]
// 1) small shapes with ragged extents against an fp64 host product, every entry; 2) the given shape: sampled entries against
// fp64, the split passes and the product timed with HIP events, the DOT epilogue against the stored result.
#include "../../quimb_amd/csrc/gemmh.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Dev {
  float *A, *B, *C, *T, *hdr, *slotsA, *slotsB;
  char *PA, *PB, *meanA, *meanB;
  double* partial;
};

static void fill_args(GettArgs& g, SplitArgs& sa, SplitArgs& sb, int M, int N, int K, int ta, int tb) {
  memset(&g, 0, sizeof(g));
  const int BM = 64 * ta, BN = 64 * tb, Kpad = (K + 31) / 32 * 32;
  g.nm = 1; g.dim_m[0] = M; g.sc_m[0] = N;
  g.nn = 1; g.dim_n[0] = N; g.sc_n[0] = 1;
  g.B = 1; g.M = M; g.N = N; g.K = K; g.Kloop = Kpad;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  memset(&sa, 0, sizeof(sa));
  sa.ng = 1; sa.dim[0] = M; sa.stride[0] = 1; sa.sk = M; sa.X = M; sa.Xpad = g.tiles_m * BM; sa.K = K; sa.KG = Kpad / 8;
  sb = sa;
  sb.dim[0] = N; sb.sk = N; sb.X = N; sb.Xpad = g.tiles_n * BN;
}

static double run_case(int M, int N, int K, int ta, int tb, int iters, bool full_check, unsigned seed, int fill, bool centre = true) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(-0.1f, 1.0f), V(-1.f, 1.f);
  std::lognormal_distribution<float> L(0.f, 2.f);
  std::vector<float> hA((size_t)K * M), hB((size_t)K * N), hT((size_t)M * N);
  for (auto& x : hA) x = fill == 0 ? U(rng) : fill == 1 ? V(rng) * L(rng) : 0.f;
  for (auto& x : hB) x = fill == 0 ? U(rng) * 3.7f : fill == 1 ? V(rng) * L(rng) * 1e-3f : 0.f;
  for (auto& x : hT) x = V(rng);
  GettArgs g; SplitArgs sa, sb;
  fill_args(g, sa, sb, M, N, K, ta, tb);
  Dev d;
  const size_t pa = (size_t)qamd_gemmh_image_bytes(sa.Xpad, g.Kloop), pb = (size_t)qamd_gemmh_image_bytes(sb.Xpad, g.Kloop);
  CK(hipMalloc(&d.A, hA.size() * 4)); CK(hipMalloc(&d.B, hB.size() * 4)); CK(hipMalloc(&d.C, hT.size() * 4));
  CK(hipMalloc(&d.T, hT.size() * 4)); CK(hipMalloc(&d.hdr, 64)); CK(hipMalloc(&d.slotsA, 256)); CK(hipMalloc(&d.slotsB, 256));
  CK(hipMalloc(&d.PA, pa)); CK(hipMalloc(&d.PB, pb)); CK(hipMalloc(&d.meanA, qamd_gemmh_mean_bytes(sa.Xpad))); CK(hipMalloc(&d.meanB, qamd_gemmh_mean_bytes(sb.Xpad))); CK(hipMalloc(&d.partial, 8 * g.tiles_m * g.tiles_n));
  CK(hipMemcpy(d.A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d.T, hT.data(), hT.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d.C, 0xff, hT.size() * 4));
  hipEvent_t e0, e1, e2, e3;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  float ms_split = 0, ms_gemm = 0, ms_dot = 0;
  for (int it = 0; it < iters + 2; ++it) {
    CK(hipEventRecord(e0));
    if (qamd_gemmh_absmax_launch(&sa, d.A, d.slotsA, nullptr) || qamd_gemmh_absmax_launch(&sb, d.B, d.slotsB, nullptr)) { printf("absmax launch failed\n"); exit(1); }
    if (qamd_gemmh_split_launch(&sa, d.A, d.slotsA, d.hdr, d.PA, centre ? d.meanA : nullptr, nullptr) || qamd_gemmh_split_launch(&sb, d.B, d.slotsB, d.hdr + 2, d.PB, centre ? d.meanB : nullptr, nullptr)) { printf("split launch failed\n"); exit(1); }
    CK(hipEventRecord(e1));
    if (qamd_gemmh_launch(ta, tb, &g, d.PA, d.PB, d.C, nullptr, nullptr, d.hdr, d.hdr + 2, centre ? d.meanA : nullptr, centre ? d.meanB : nullptr, nullptr, nullptr)) { printf("gemmh launch failed\n"); exit(1); }
    CK(hipEventRecord(e2));
    if (qamd_gemmh_dot_launch(ta, tb, &g, d.PA, d.PB, d.T, d.hdr, d.hdr + 2, centre ? d.meanA : nullptr, centre ? d.meanB : nullptr, d.partial, nullptr)) { printf("dot launch failed\n"); exit(1); }
    CK(hipEventRecord(e3));
    CK(hipDeviceSynchronize());
    if (it >= 2) {
      float a, b, c;
      CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); CK(hipEventElapsedTime(&c, e2, e3));
      ms_split += a; ms_gemm += b; ms_dot += c;
    }
  }
  ms_split /= iters; ms_gemm /= iters; ms_dot /= iters;
  std::vector<float> hC(hT.size());
  std::vector<double> hp(g.tiles_m * g.tiles_n);
  float hdr[4];
  CK(hipMemcpy(hC.data(), d.C, hC.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hp.data(), d.partial, hp.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hdr, d.hdr, 16, hipMemcpyDeviceToHost));
  // reference
  double cmax = 0, worst = 0, worst32 = 0;
  auto ref = [&](int m, int n, float* f32) {
    double s = 0;
    float s32 = 0;
    for (int k = 0; k < K; ++k) { s += (double)hA[(size_t)k * M + m] * hB[(size_t)k * N + n]; s32 = fmaf(hA[(size_t)k * M + m], hB[(size_t)k * N + n], s32); }
    if (f32) *f32 = s32;
    return s;
  };
  std::vector<std::pair<int, int>> pts;
  if (full_check) { for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) pts.push_back({m, n}); }
  else {
    std::uniform_int_distribution<int> Um(0, M - 1), Un(0, N - 1);
    for (int i = 0; i < 4000; ++i) pts.push_back({Um(rng), Un(rng)});
    for (int i = 0; i < 64; ++i) { pts.push_back({M - 1 - i % 7, Un(rng)}); pts.push_back({Um(rng), N - 1 - i % 5}); pts.push_back({(i * 131) % M, (i * 257) % N}); }
    pts.push_back({0, 0}); pts.push_back({M - 1, N - 1}); pts.push_back({0, N - 1}); pts.push_back({M - 1, 0});
  }
  std::vector<double> refs(pts.size());
  std::vector<float> refs32(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) { refs[i] = ref(pts[i].first, pts[i].second, &refs32[i]); cmax = std::max(cmax, std::fabs(refs[i])); }
  int bad = 0;
  double bias = 0, bias32 = 0;      // mean signed relative error: a rounding that is not symmetric shows up here, not in the maximum
  for (size_t i = 0; i < pts.size(); ++i) {
    const double got = hC[(size_t)pts[i].first * N + pts[i].second];
    if (refs[i] != 0) { bias += (got - refs[i]) / std::fabs(refs[i]) / pts.size(); bias32 += ((double)refs32[i] - refs[i]) / std::fabs(refs[i]) / pts.size(); }
    const double e = std::fabs(got - refs[i]) / (cmax > 0 ? cmax : 1), e32 = std::fabs((double)refs32[i] - refs[i]) / (cmax > 0 ? cmax : 1);
    if (!(e < 1e-5)) { if (bad < 5) printf("   MISMATCH at (%d, %d): got %.9g want %.9g\n", pts[i].first, pts[i].second, got, refs[i]); ++bad; }
    worst = std::max(worst, e); worst32 = std::max(worst32, e32);
  }
  // DOT: sum of the stored result times T, in fp64, against the kernel's partial sums
  double dsum = 0, dref = 0;
  for (double x : hp) dsum += x;
  for (size_t i = 0; i < hC.size(); ++i) dref += (double)hC[i] * hT[i];
  const double flop = 2.0 * M * N * K;
  printf("%5d x %5d x %5d tile %dx%d fill %d %s: max-norm err vs fp64 %.2e (an fp32 fmaf chain: %.2e) over %zu entries%s; mean signed rel err %+.2e (fmaf chain %+.2e); dot rel diff %.2e; scales 2^%d 2^%d | split %.3f ms, product %.3f ms = %.1f TFLOP/s fp32-equivalent (%.0f on the f16 pipe), product + dot epilogue %.3f ms\n",
         M, N, K, 64 * ta, 64 * tb, fill, centre ? "centred" : "as is", worst, worst32, pts.size(), bad ? "  ** FAILED **" : "", bias, bias32, std::fabs(dsum - dref) / (std::fabs(dref) + 1e-300),
         (int)std::log2(hdr[0]), (int)std::log2(hdr[2]), ms_split, ms_gemm, flop / ms_gemm * 1e-9, 3 * flop / ms_gemm * 1e-9, ms_dot);
  CK(hipFree(d.A)); CK(hipFree(d.B)); CK(hipFree(d.C)); CK(hipFree(d.T)); CK(hipFree(d.hdr)); CK(hipFree(d.slotsA)); CK(hipFree(d.slotsB));
  CK(hipFree(d.PA)); CK(hipFree(d.PB)); CK(hipFree(d.meanA)); CK(hipFree(d.meanB)); CK(hipFree(d.partial));
  return bad;
}

int main(int argc, char** argv) {
  int bad = 0;
  if (argc < 4) {
    const int tiles[6][2] = {{4, 4}, {3, 4}, {4, 3}, {3, 3}, {2, 4}, {4, 2}};
    for (auto& t : tiles) {
      bad += run_case(300, 520, 200, t[0], t[1], 1, true, 1, 0);
      bad += run_case(516, 260, 72, t[0], t[1], 1, true, 2, 1);
      bad += run_case(1000, 1000, 1024, t[0], t[1], 1, true, 3, 0);
      bad += run_case(300, 520, 200, t[0], t[1], 1, true, 4, 0, false);
      bad += run_case(301, 523, 136, t[0], t[1], 1, true, 5, 0);      // extents that are no multiples of 4: one column per thread in the split pass
    }
    printf(bad ? "FAILED\n" : "all small cases OK\n");
    return bad ? 1 : 0;
  }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const int ta = argc > 4 ? atoi(argv[4]) : 4, tb = argc > 5 ? atoi(argv[5]) : 4, iters = argc > 6 ? atoi(argv[6]) : 5;
  const int fill = argc > 7 ? atoi(argv[7]) : 0;
  const bool centre = argc > 8 ? atoi(argv[8]) != 0 : true;
  bad = run_case(M, N, K, ta, tb, iters, false, 7, fill, centre);
  return bad ? 1 : 0;
}
