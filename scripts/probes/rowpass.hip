// Probe: one ROW of a boundary sweep (five site absorptions of a corner block, quimb tn2d/core.py:1393-1402) as ONE launch.
// The open legs h of earlier rows are spectators of the whole row, and so is the first site's new down leg d1 once
// that site is absorbed: a work item (S = (h...), d1) carries a 6^5-element state through the remaining four sites in LDS
// (two ping-pong images [36 k][216 rest]), every site a [36 x 36] matrix on v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 rowpass.hip -o /tmp/rowpass && /tmp/rowpass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float acc4;
typedef float vec4 __attribute__((ext_vector_type(4), aligned(16)));

struct RowArgs {
  int64_t sv[5];        // input strides of the five up legs v1..v5
  int64_t sd[5], sh;    // output strides of d1..d5 and of the row's new open leg h
  int32_t nS;           // spectator groups
  uint32_t dimS[4];
  int64_t sSa[4], sSc[4];
  int64_t ws[5][4];     // site tensor strides of (up, left, down, right); site 0 has no left leg, site 4's right leg is h
  uint32_t items;       // S values x D
  uint32_t ablate;      // probe: 1 no final stores, 2 no gather of A, 4 no LDS scatter between sites
};

constexpr int D = 6, DD = 36, R = 216, RP = 224, LDW = 48;

__global__ __launch_bounds__(256, 2) void rowpass_kernel(const RowArgs p, const float* __restrict__ A,
                                                         const float* __restrict__ W0, const float* __restrict__ W1,
                                                         const float* __restrict__ W2, const float* __restrict__ W3,
                                                         const float* __restrict__ W4, float* __restrict__ C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ST0 = sm;                      // [36][RP]
  float* ST1 = ST0 + DD * RP;           // [36][RP]
  float* Wl = ST1 + DD * RP;            // [36][LDW]  (+ slack behind it: padded reads of the last state row end here)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
  const uint32_t item = blockIdx.x;
  const uint32_t d1 = item % D;
  uint32_t sidx = item / D;
  int64_t abase = 0, cbase = (int64_t)d1 * p.sd[0];
  for (int g = p.nS - 1; g >= 0; --g) {
    const uint32_t dg = p.dimS[g], q = sidx / dg, r = sidx - q * dg;
    abase += (int64_t)r * p.sSa[g];
    cbase += (int64_t)r * p.sSc[g];
    sidx = q;
  }
  // ---- all global loads of the prologue first: the A elements of site 0 (6 rounds x 6), W0's slice, W1's image pieces -----
  float* Wl2[2] = {Wl, Wl + DD * LDW};
  constexpr int NE = (D * R + 255) / 256;          // rounds of site 0
  constexpr int NWP = (DD * LDW + 255) / 256;      // image pieces per thread and site
  float x[NE][D];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    int e = tid + 256 * i;
    e = e < D * R ? e : D * R - 1;
    const int v2 = e / R, rest = e - v2 * R;
    const int v3 = rest / DD, v4 = (rest / D) % D, v5 = rest % D;
    const float* ap = A + abase + v2 * p.sv[1] + v3 * p.sv[2] + v4 * p.sv[3] + v5 * p.sv[4];
#pragma unroll
    for (int v = 0; v < D; ++v) x[i][v] = (p.ablate & 2) ? 1.0f : ap[v * p.sv[0]];
  }
  float w0[D][D];
#pragma unroll
  for (int v = 0; v < D; ++v)
#pragma unroll
    for (int b = 0; b < D; ++b) w0[v][b] = W0[v * p.ws[0][0] + d1 * p.ws[0][2] + b * p.ws[0][3]];
  float wreg[NWP];
  auto wload = [&](const float* Wc, int c) {
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int e = tid + 256 * i;
      const int k = e / LDW, n = e - k * LDW;
      const bool ok = e < DD * LDW && n < DD;
      const float v = Wc[ok ? (k % D) * p.ws[c][0] + (k / D) * p.ws[c][1] + (n % D) * p.ws[c][2] + (n / D) * p.ws[c][3] : 0];
      wreg[i] = ok ? v : 0.f;
    }
  };
  auto wstore = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int e = tid + 256 * i;
      if (e < DD * LDW) dst[e] = wreg[i];
    }
  };
  wload(W1, 1);
  // ---- site 0: t1[b1][v2..v5] = sum_v1 W0[v1, d1, b1] A[v1, v2..v5]  -> ST0[(b1, v2)][(v3, v4, v5)] ------------------
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + 256 * i;
    if (e < D * R) {
      const int v2 = e / R, rest = e - v2 * R;
#pragma unroll
      for (int b = 0; b < D; ++b) {
        float acc0 = 0.f;
#pragma unroll
        for (int v = 0; v < D; ++v) acc0 += w0[v][b] * x[i][v];
        ST0[(b * D + v2) * RP + rest] = acc0;
      }
    }
  }
  wstore(Wl2[0]);
  // per-lane pieces of the index maps (additive: offset(n) + offset(m))
  const int m0 = 64 * wave + 4 * j;
  int offm_l[4];
  int64_t offm_g[4];
  bool okm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m0 + t;
    okm[t] = m < R;
    const int w = m / DD, r2 = m % DD;
    offm_l[t] = w * RP + r2 * D;                                   // next image: row (b', w), column (r2, d)
    offm_g[t] = (int64_t)(m / DD) * p.sd[1] + (int64_t)((m / D) % D) * p.sd[2] + (int64_t)(m % D) * p.sd[3];   // m = (d2, d3, d4)
  }
  int offn_l[3][4];
  int64_t offn_g[3][4];
  bool okn[3][4];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 16 * nt + 4 * kq + r;
      okn[nt][r] = n < DD;
      offn_l[nt][r] = (n / D) * D * RP + (n % D);                  // n = (b', d)
      offn_g[nt][r] = (int64_t)(n / D) * p.sh + (int64_t)(n % D) * p.sd[4];   // last site: n = (h, d5)
    }
  float* cur = ST0;
  float* nxt = ST1;
#pragma unroll 1
  for (int c = 1; c < 5; ++c) {
    __syncthreads();     // image `cur` and this site's W image complete; the other W image is free
    const float* Wn = c == 1 ? W2 : (c == 2 ? W3 : W4);
    if (c < 4) wload(Wn, c + 1);                      // the next site's image pieces, in flight under the MFMAs
    const float* Wc = Wl2[(c - 1) & 1];
    acc4 acc[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[t][nt] = acc4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const vec4 bv = *reinterpret_cast<const vec4*>(cur + (4 * s + kq) * RP + m0);
      float w[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) w[nt] = Wc[(4 * s + kq) * LDW + 16 * nt + j];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], bv[t], acc[t][nt], 0, 0, 0);
    }
    if (c < 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (okm[t] && okn[nt][r] && !(p.ablate & 4)) nxt[offn_l[nt][r] + offm_l[t]] = acc[t][nt][r];
      wstore(Wl2[c & 1]);
      float* tmp = cur; cur = nxt; nxt = tmp;
    } else {
      float* cp = C + cbase;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (okm[t] && okn[nt][r] && (!(p.ablate & 1) || acc[t][nt][r] == 12345.678f)) cp[offn_g[nt][r] + offm_g[t]] = acc[t][nt][r];
    }
  }
}

static double urand() { return (double)rand() / RAND_MAX - 0.4; }

int main() {
  // row with S = 6^ns spectators.  Layouts (death-ordered, as the executor's): A[v1..v5][S], C[d1..d5][S][h]
  for (int ns = 1; ns <= 3; ++ns) {
    const int64_t S = (int64_t)pow(6, ns);
    const int64_t na = 7776 * S, nc = 7776 * S * 6;
    std::vector<float> hA(na), hW[5];
    for (auto& x : hA) x = (float)urand();
    for (int c = 0; c < 5; ++c) { hW[c].resize(1296); for (auto& x : hW[c]) x = (float)urand(); }
    RowArgs p{};
    // A: [S][v1..v5] (what a fused row hands to the next); C: [d1..d5][S][h] (death-ordered, what follows the last fused row)
    { int64_t st = 1; for (int i = 4; i >= 0; --i) { p.sv[i] = st; st *= 6; } }
    { p.sh = 1; int64_t st = 6 * S; for (int i = 4; i >= 0; --i) { p.sd[i] = st; st *= 6; } }
    p.nS = 1; p.dimS[0] = (uint32_t)S; p.sSa[0] = 7776; p.sSc[0] = 6;
    // site tensors W[up][left][down][right] contiguous (site 0: left has extent 1 -> [up][down][right])
    for (int c = 0; c < 5; ++c) {
      if (c == 0) { p.ws[c][0] = 36; p.ws[c][1] = 0; p.ws[c][2] = 6; p.ws[c][3] = 1; }
      else { p.ws[c][0] = 216; p.ws[c][1] = 36; p.ws[c][2] = 6; p.ws[c][3] = 1; }
    }
    p.items = (uint32_t)(S * 6);
    float *dA, *dC, *dW[5];
    hipMalloc(&dA, na * 4); hipMalloc(&dC, nc * 4);
    hipMemcpy(dA, hA.data(), na * 4, hipMemcpyHostToDevice);
    for (int c = 0; c < 5; ++c) { hipMalloc(&dW[c], 1296 * 4); hipMemcpy(dW[c], hW[c].data(), 1296 * 4, hipMemcpyHostToDevice); }
    hipMemset(dC, 0, nc * 4);
    const size_t lds = (size_t)(2 * DD * RP + 2 * DD * LDW + 64) * 4;
    hipFuncSetAttribute((const void*)rowpass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    auto launch = [&]() { hipLaunchKernelGGL(rowpass_kernel, dim3(p.items), dim3(256), lds, 0, p, dA, dW[0], dW[1], dW[2], dW[3], dW[4], dC); };
    launch();
    hipDeviceSynchronize();
    std::vector<float> hC(nc);
    hipMemcpy(hC.data(), dC, nc * 4, hipMemcpyDeviceToHost);
    // reference for a few S values (fp64)
    double maxerr = 0, maxref = 0;
    for (int64_t s : {(int64_t)0, S / 2, S - 1}) {
      // t[v1..v5] -> site by site
      std::vector<double> t(7776);
      for (int i = 0; i < 7776; ++i) t[i] = hA[(int64_t)s * 7776 + i];
      // state indices: (b, v_c.., d...) generic: keep as map from tuple
      // site 0: t1[d1][b1][v2..v5]
      std::vector<double> cur(6 * 6 * 1296);   // [d1][b1][v2..v5]
      for (int d1 = 0; d1 < 6; ++d1) for (int b = 0; b < 6; ++b) for (int r = 0; r < 1296; ++r) {
        double a = 0; for (int v = 0; v < 6; ++v) a += hW[0][v * 36 + d1 * 6 + b] * t[v * 1296 + r];
        cur[(d1 * 6 + b) * 1296 + r] = a;
      }
      // now cur[dprev (6^c)][b][v_{c+1}][rest_v] ; iterate sites 1..4
      int64_t nd = 6;            // number of accumulated d values
      int64_t nv = 216;          // remaining v's after the one being contracted
      std::vector<double> nx;
      for (int c = 1; c < 5; ++c) {
        // cur[dd (nd)][b (6)][v (6)][rv (nv)]  ->  nx[dd][d (6)][b' (6)][rv]      (site 4: b' = h)
        nx.assign(nd * 36 * nv, 0.0);
        for (int64_t dd = 0; dd < nd; ++dd) for (int d = 0; d < 6; ++d) for (int bp = 0; bp < 6; ++bp) for (int64_t rv = 0; rv < nv; ++rv) {
          double a = 0;
          for (int b = 0; b < 6; ++b) for (int v = 0; v < 6; ++v)
            a += hW[c][v * 216 + b * 36 + d * 6 + bp] * cur[((dd * 6 + b) * 6 + v) * nv + rv];
          nx[((dd * 6 + d) * 6 + bp) * nv + rv] = a;
        }
        cur = nx; nd *= 6; nv /= 6;
      }
      // cur[d1..d5 (7776)][h (6)]
      for (int64_t i = 0; i < 7776; ++i) for (int h = 0; h < 6; ++h) {
        const double ref = cur[i * 6 + h];
        const double got = hC[(i * S + s) * 6 + h];
        maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
      }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("S = 6^%d (%lld items): %.1f us per launch, max err %.3e (max |ref| %.3e), %s\n", ns, (long long)p.items, ms / 20 * 1e3, maxerr, maxref,
           maxerr <= 1e-5 * maxref ? "OK" : "MISMATCH");
    for (uint32_t ab : {1u, 2u, 3u, 4u, 7u}) {
      p.ablate = ab;
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("    ablate %u: %.1f us\n", ab, ms / 20 * 1e3);
    }
    p.ablate = 0;
    hipFree(dA); hipFree(dC); for (int c = 0; c < 5; ++c) hipFree(dW[c]);
  }
  return 0;
}
