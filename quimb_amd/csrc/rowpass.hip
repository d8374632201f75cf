// rowpass.hip -- one ROW of a 2D boundary sweep as ONE launch (gfx950 only, fp32, bond dimension 6, five sites).
//
// quimb absorbs a row of a PEPS-like network into the boundary site by site (quimb/tensor/tn2d/core.py:1393-1402; exact
// mode: every absorption a pairwise contraction of the growing boundary tensor with one site tensor).  For the small rows of
// a corner sweep those five dependent launches are latency, not bandwidth: this kernel runs all five in one.
//
//   T'[S, d1..d5, h] = sum_{v1..v5, b1..b4}  T[S, v1..v5] W1[v1, d1, b1] W2[v2, b1, d2, b2] ... W5[v5, b4, d5, h]
//
// S -- every other index of the boundary tensor -- is a spectator of the whole row, and so is d1 once the first site is
// absorbed: a work item (S, d1) carries a 6^5-element state through the remaining four sites in LDS.  A state image is
// [36 k][216 rest]: k = (bond, next up leg) is what the coming site contracts, rest the other three open positions; a site
// is a [36 x 36] matrix applied to it on v_mfma_f32_16x16x4_f32 (W staged once per site as the MFMA A operand, the image
// read as 16-byte vectors of four consecutive rest values = four B operands), and the result is scattered into the OTHER image
// already in the order the next site wants: row (new bond, next up leg), column (rest', new down leg).  One barrier per
// site; the next site's tensor is fetched under the current site's MFMAs.  The last site's result goes straight to global
// memory at the strides the caller names -- so the row reads and writes the SAME layouts the five separate steps would
// have (any strides on T, the site tensors and T').  Two workgroups share a CU (72 KB of LDS each).
//
// Fused exponent stripping as in chain2*.hip: the result is scaled by 1 / (max|T| max|W1| ... max|W5|) and its own
// absmax recorded; the four intermediates never exist, so they carry no exponent of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamdr {

typedef __attribute__((ext_vector_type(4))) float acc4;
typedef float vec4 __attribute__((ext_vector_type(4), aligned(16)));

constexpr int D = 6, DD = 36, R = 216, RP = 224, LDW = 48;

struct RowPtrs {
  const float* W[5];
  const float* scale_w[5];
};

__device__ __forceinline__ float rp_read_scale(const float* slots) {
  if (!slots) return 1.f;
  float m = 0.f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.f ? m : 1.f;
}

__global__ __launch_bounds__(256, 2) void rowpass_kernel(const RowArgs p, const RowPtrs w, const float* __restrict__ A,
                                                         float* __restrict__ C, const float* __restrict__ scale_a,
                                                         float* __restrict__ absmax_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ST0 = sm;                      // [36][RP]
  float* ST1 = ST0 + DD * RP;           // [36][RP]
  float* Wl = ST1 + DD * RP;            // two site images [36][LDW] (+ slack: the padded reads of a state row's tail end in what follows it --
                                        // finite or not, those columns m >= 216 only ever reach result columns that are not stored)
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
    for (int v = 0; v < D; ++v) x[i][v] = ap[v * p.sv[0]];
  }
  float w0[D][D];
#pragma unroll
  for (int v = 0; v < D; ++v)
#pragma unroll
    for (int b = 0; b < D; ++b) w0[v][b] = w.W[0][v * p.ws[0][0] + d1 * p.ws[0][2] + b * p.ws[0][3]];
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
  wload(w.W[1], 1);
  // ---- site 0: t1[b1][v2..v5] = sum_v1 w.W[0][v1, d1, b1] A[v1, v2..v5]  -> ST0[(b1, v2)][(v3, v4, v5)] ------------------
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
  bool okm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int m = m0 + t;
    okm[t] = m < R;
    const int w = m / DD, r2 = m % DD;
    offm_l[t] = w * RP + r2 * D;                                   // next image: row (b', w), column (r2, d)
  }
  int offn_l[3][4];
  bool okn[3][4];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 16 * nt + 4 * kq + r;
      okn[nt][r] = n < DD;
      offn_l[nt][r] = (n / D) * D * RP + (n % D);                  // n = (b', d)
    }
  float alpha = 1.f / rp_read_scale(scale_a);
#pragma unroll
  for (int c = 0; c < 5; ++c) alpha /= rp_read_scale(w.scale_w[c]);
  float vmax = 0.f;
  float* cur = ST0;
  float* nxt = ST1;
#pragma unroll 1
  for (int c = 1; c < 5; ++c) {
    __syncthreads();     // image `cur` and this site's W image complete; the other W image is free
    if (c < 4) wload(w.W[c + 1], c + 1);                      // the next site's image pieces, in flight under the MFMAs
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
            if (okm[t] && okn[nt][r]) nxt[offn_l[nt][r] + offm_l[t]] = acc[t][nt][r];
      wstore(Wl2[c & 1]);
      float* tmp = cur; cur = nxt; nxt = tmp;
    } else {
      // the last site: n = (h, d5), m = (d2, d3, d4) -> the caller's strides
      float* cp = C + cbase;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = m0 + t;
        const int64_t om = (int64_t)(m / DD) * p.sd[1] + (int64_t)((m / D) % D) * p.sd[2] + (int64_t)(m % D) * p.sd[3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * nt + 4 * kq + r;
            if (okm[t] && okn[nt][r]) {
              const float v = acc[t][nt][r] * alpha;
              cp[om + (int64_t)(n / D) * p.sh + (int64_t)(n % D) * p.sd[4]] = v;
              vmax = fmaxf(vmax, fabsf(v));
            }
          }
      }
    }
  }
  if (absmax_out) {
#pragma unroll
    for (int dl = 32; dl > 0; dl >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, dl, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS), __float_as_uint(vmax));
  }
}

// The FIRST row of a sweep has no boundary tensor yet: its five site tensors (the lattice edge: no up legs) are simply
// multiplied along their bonds,
//   C[d1..d5, h] = sum_{b1..b4} W0[d1, b1] W1[b1, d2, b2] W2[b2, d3, b3] W3[b3, d4, b4] W4[b4, d5, h]
// -- four dependent launches of a few microseconds of work for the pairwise executor, one here.  A thread owns
// (d1, d2, d3, d4): it carries the 6-vector over the open bond through sites 1..3 and writes its 36 results (d5, h).
__global__ __launch_bounds__(256) void rowfirst_kernel(const RowArgs p, const RowPtrs w, float* __restrict__ C,
                                                       float* __restrict__ absmax_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float vmax = 0.f;
  if (e < D * D * D * D) {
    const int d1 = e / (D * D * D), d2 = (e / (D * D)) % D, d3 = (e / D) % D, d4 = e % D;
    const int dd[4] = {d1, d2, d3, d4};
    float t[D], u[D];
#pragma unroll
    for (int b = 0; b < D; ++b) t[b] = w.W[0][d1 * p.ws[0][2] + b * p.ws[0][3]];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
#pragma unroll
      for (int bn = 0; bn < D; ++bn) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += t[b] * w.W[c][b * p.ws[c][1] + dd[c] * p.ws[c][2] + bn * p.ws[c][3]];
        u[bn] = s;
      }
#pragma unroll
      for (int b = 0; b < D; ++b) t[b] = u[b];
    }
    float alpha = 1.f;
#pragma unroll
    for (int c = 0; c < 5; ++c) alpha /= rp_read_scale(w.scale_w[c]);
    float* cp = C + d1 * p.sd[0] + d2 * p.sd[1] + d3 * p.sd[2] + d4 * p.sd[3];
#pragma unroll
    for (int d5 = 0; d5 < D; ++d5)
#pragma unroll
      for (int h = 0; h < D; ++h) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += t[b] * w.W[4][b * p.ws[4][1] + d5 * p.ws[4][2] + h * p.ws[4][3]];
        s *= alpha;
        cp[d5 * p.sd[4] + h * p.sh] = s;
        vmax = fmaxf(vmax, fabsf(s));
      }
  }
  if (absmax_out) {
#pragma unroll
    for (int dl = 32; dl > 0; dl >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, dl, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS), __float_as_uint(vmax));
  }
}

}  // namespace qamdr

using namespace qamdr;

// One row (five sites, D = 6) in one launch.  a->items = (number of S values) * 6; a->nS = -1: the first row (no boundary tensor).  W[c] / scale_w[c]: the five site
// tensors and their absmax slots (NULL = 1); strides in a->ws (elements).
extern "C" int qamd_rowpass_launch(const RowArgs* a, const void* A, const void* const* W, void* C, const void* scale_a,
                                   const void* const* scale_w, void* absmax_out, void* stream) {
  if (!a || a->nS > 4) return -2;
  RowPtrs w;
  for (int c = 0; c < 5; ++c) {
    w.W[c] = (const float*)W[c];
    w.scale_w[c] = scale_w ? (const float*)scale_w[c] : nullptr;
  }
  if (a->nS < 0) {   // the first row of a sweep: no boundary tensor
    QAMD_LAUNCH(rowfirst_kernel, dim3((D * D * D * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a, w, (float*)C,
                (float*)absmax_out);
    return hipGetLastError() == hipSuccess ? 0 : -4;
  }
  if (a->items == 0) return -2;
  const size_t lds = (size_t)(2 * DD * RP + 2 * DD * LDW + 64) * sizeof(float);
  // (per launch, as stream.hip / gemmd.hip do: the attribute belongs to the CURRENT device, and a process-wide flag would
  // leave a second GPU of the process without it -- and be a data race between threads)
  (void)hipFuncSetAttribute((const void*)rowpass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  QAMD_LAUNCH(rowpass_kernel, dim3(a->items), dim3(256), lds, (hipStream_t)stream, *a, w, (const float*)A, (float*)C,
              (const float*)scale_a, (float*)absmax_out);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
