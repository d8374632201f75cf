// gemmh.hip -- the fp32 GEMM-shaped joins on the f16 matrix pipe: split products ("f16x3"), gfx950 only.  OPT-IN
// (qamd_pair_plan.kernel = -7 on input: the k-outer joins, -8: any operand layout; quimb_amd.Options.join_arith = "f16x3" /
// "f16x3-all"): the default stays gemmk.hip on the fp32 MFMA.
//
//   C[m, n] = alpha * sum_k A[k, m] * B[k, n]         (operands fp32 with ANY strides: the split pass re-lays them out; the
//                                                      k-outer joins gemmk.hip takes are the coalesced case it is tuned for)
//
// An fp32 value x, scaled by a power of two so that the operand's largest magnitude sits in [2^14, 2^15), is written as
//   x = h1 + h2 + e,   h1 = fp16(x), h2 = fp16(x - h1),   |e| <= 2^-24 |x|   (two round-to-nearest steps of 11 bits each,
// the residual x - h1 is exact in fp32), for every |x| >= 2^-18 of the operand's maximum; smaller entries lose relative
// but not absolute accuracy (absolute error <= 2^-25 * 2^-15 of the maximum).  Then
//   a * b = a1 b1 + a1 b2 + a2 b1   - dropped: a2 b2 <= 2^-24 |a b| -
// and every one of the three products is EXACT in the fp32 accumulator's input (11 x 11 bits): three passes of
// v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense: 16 x the fp32 MFMA rate) with fp32 accumulation give a result whose error is
// the fp32 accumulation error plus <= 3 * 2^-24 per term -- measured against fp64 (with the centring below) 7e-8 max-norm on
// the 10x10 D=6 joins' shape and fill, where the fp32 MFMA kernel -- an fp32 fma chain -- carries 5e-6 (tests/checks.py
// check_gemmh).  No bf16 (8 bits: six products for the same accuracy), no TF32-like truncation.
//
//  * hmean_kernel / hcentre_kernel / split_kernel, per operand (any strides the fp32 kernels accept): column sums over k
//    (doubles, fixed summation order), the constant to subtract from every column (its mean where the column is coherent
//    and the mean typical of it, else zero: hcentre_kernel), then ONE pass that subtracts, scales (power of two from the
//    tensor's absmax slots) and writes both halves as tile-ready images  P[half][k / 8][x][8 k]  (16 bytes per (k-group, x);
//    x padded with zeros to whole workgroup tiles, K to a multiple of 32): a 1 KiB LDS-DMA piece is 64 consecutive x of
//    one k-group, and a 32x32x16 fragment read (lane = x, 8 consecutive k per half wave) is one conflict-free ds_read_b128.
//    Why centre: the f16 MFMA's accumulate truncates aligned addends ~10 bits below the result's last place -- a bias that
//    an all-positive K = 7776 sum shows as -2e-7 in every entry; centred, the MFMA sums sign-mixed fluctuations and the part
//    carried by the constants, K (c bbar + d (abar - c)), is added exactly (double precision) in the epilogue.
//  * gemmh_kernel: gemmk.hip's recipe on the f16 instruction: 2 x 2 waves, wave tile (32 TA) x (32 TB), operands
//    HBM/L2 -> LDS by LDS-DMA only, a ring of 4 .. 6 stages of 16 k (32 KB each at 256 x 256: whatever fits 160 KB), ONE
//    barrier per stage behind the step's first MFMA, the next step's fragments read right after it, the request for stage
//    t + NS one piece behind each following MFMA, counted vmcnt (never 0 inside the loop): a stage is requested NS - 1
//    steps before it is waited for.  Per 16-k step a wave reads 2 (TA + TB) fragments and issues 3 TA TB MFMAs
//    (a1 b2, a2 b1, a1 b1 -- small terms first).
//  * gemmh8_kernel (even tiles: 256 x 256, 128 x 256, 256 x 128): the same ring with EIGHT waves in two groups that run half a step
//    apart -- while one group issues its step's MFMAs at raised priority, the other reads its fragments and requests a later
//    stage: 7776^3 2.43 -> 2.30 ms, zero operands 1718 -> 1890 TFLOP/s (75 % of the f16 peak).
//  * epilogue as gemmk.hip's: alpha * 2^-(ea + eb), absmax slot, or (DOT) the closing inner product with T.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

#ifndef QAMD_GEMMH_BAND
#define QAMD_GEMMH_BAND 4    // tile rows per band of the tile walk (an XCD's 32 workgroups share BAND A panels and 32 / BAND B panels)
#endif
#ifndef QAMD_GEMMH_W8
#define QAMD_GEMMH_W8 1       // even tiles run the eight-wave, two-group kernel (gemmh8_kernel): +6 % at 7776^3; 0 = gemmh_kernel for every tile
#endif
#define QAMD_GEMMH_NY 32     // partial rows of a column's sums over k (the mean pass)
// An operand's centring block (doubles unless said otherwise), always laid out for two parities of k (period 1 uses parity 0):
//   mean[2][Xpad] | part[3 kinds][NY][2][Xpad] | cused[2][Xpad] (floats)
#define QAMD_GEMMH_PART0(xpad) (2 * (int64_t)(xpad))
#define QAMD_GEMMH_USED0(xpad) ((2 + 6 * QAMD_GEMMH_NY) * (int64_t)(xpad))

namespace qamdh {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float acc16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int64_t hdecomp(uint32_t idx, int n, const uint32_t* dims, const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

// source offset of contraction index k: one stride, or (several K groups) a decomposition
__device__ __forceinline__ int64_t hkoff(const SplitArgs& p, uint32_t k) {
  return p.nk <= 1 ? (int64_t)k * p.sk : hdecomp(k, p.nk, p.dim_k, p.stride_k);
}

// max over a tensor's 64 absmax slots (one per lane, wave reduction)
__device__ __forceinline__ float hread_scale(const float* slots, int lane) {
  static_assert(QAMD_SLOTS == 64, "one slot per lane");
  if (!slots) return 1.f;
  float m = slots[lane];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  return m > 0.f ? m : 1.f;
}

__device__ __forceinline__ void hglds(const char* src, char* dst) {
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// ---- absmax of a strided fp32 operand into 64 slots (only when the caller has none: plain tensordot without exponents) --------
__global__ __launch_bounds__(256) void habsmax_kernel(const SplitArgs p, const float* __restrict__ X, float* __restrict__ slots) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  float m = 0.f;
  if (x < p.X) {
    const int64_t off = hdecomp(x, p.ng, p.dim, p.stride);
    for (uint32_t k = blockIdx.y; k < p.K; k += gridDim.y) m = fmaxf(m, fabsf(X[hkoff(p, k) + off]));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0)
    atomicMax(reinterpret_cast<unsigned int*>(slots) + ((blockIdx.x * 4 + (threadIdx.x >> 6) + blockIdx.y) % QAMD_SLOTS), __float_as_uint(m));
}

// ---- column sums over k, in NY partial rows of doubles: part[y][x] = sum of X[k, x] over the y-th k range, behind the NY rows
// another NY of sum sqrt|X[k, x]| and NY of sum |X[k, x]| (no atomics: the split pass adds a column's partials in a fixed order) --
__global__ __launch_bounds__(256) void hmean_kernel(const SplitArgs p, const float* __restrict__ X, double* __restrict__ part) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Xpad) return;
  const uint32_t pm = p.period == 2 ? 1u : 0u;      // parity mask of k
  double s[2] = {0.0, 0.0};    // the column sums themselves: exact to double precision (they go into the result)
  float r[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};   // the sums the centring RULE looks at: fp32 is plenty (and deterministic)
  if (x < p.X) {
    const float* src = X + hdecomp(x, p.ng, p.dim, p.stride);
    uint32_t per = (p.K + gridDim.y - 1) / gridDim.y;
    per += per & 1;                                   // (even ranges: the parity of k is then the parity of its place in the range)
    const uint32_t k0 = blockIdx.y * per, k1 = (k0 + per < p.K) ? k0 + per : p.K;
    uint32_t k = k0 < p.K ? k0 : p.K;
    for (; k + 8 <= k1; k += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[hkoff(p, k + j)];
      const double se = ((double)v[0] + (double)v[2]) + ((double)v[4] + (double)v[6]), so = ((double)v[1] + (double)v[3]) + ((double)v[5] + (double)v[7]);
      float re = 0.f, ro = 0.f, ae = 0.f, ao = 0.f;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float b0 = fabsf(v[j]), b1 = fabsf(v[j + 1]);
        ae += b0; ao += b1;
        re += __builtin_sqrtf(b0); ro += __builtin_sqrtf(b1);
      }
      if (pm) { s[0] += se; s[1] += so; r[0] += re; r[1] += ro; a1[0] += ae; a1[1] += ao; }
      else { s[0] += se + so; r[0] += re + ro; a1[0] += ae + ao; }
    }
    for (; k < k1; ++k) {
      const float v = src[hkoff(p, k)];
      const uint32_t q = k & pm;
      s[q] += (double)v;
      r[q] += __builtin_sqrtf(fabsf(v));
      a1[q] += fabsf(v);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    part[((int64_t)(0 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = s[q];
    part[((int64_t)(1 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = (double)r[q];
    part[((int64_t)(2 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = (double)a1[q];
  }
}

// ---- the centring constants: one thread per column adds the partials in a fixed order -------------------------------------------
// mean[q][x] = the column's exact mean over the k of parity q (double; period 1: all k, q = 0), cused[q][x] = the fp32
// constant c the split pass subtracts from them: with ANY constants (per parity: one such term each)
//   sum_k a b = sum_k (a - c)(b - d) + K (c bbar + d (abar - c)),
// which the product kernel's epilogue adds back.  c = the mean where the column is COHERENT (|sum x| >= 0.75 sum |x|: only
// same-sign sums carry the accumulate's bias; the sample mean of a sign-mixed column is noise) and the mean is TYPICAL of it;
// where it is carried by outliers (heavy tails: |mean| above 1.5 .. 2 x the power mean (sum sqrt|x| / K)^2, which outliers
// barely move) the constant shrinks to zero -- subtracting such a mean would turn every ordinary entry into a large one of
// the same sign.
__global__ __launch_bounds__(256) void hcentre_kernel(const SplitArgs p, const double* __restrict__ part, int ny,
                                                      double* __restrict__ mean, float* __restrict__ cused) {
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Xpad) return;
  const int np = p.period == 2 ? 2 : 1;
  const double Kp = (double)p.K / np;                                     // k values of one parity (K is even when period = 2)
  for (int q = 0; q < 2; ++q) {
    double abar = 0.0, lam = 0.0;
    if (q < np) {
      double sum = 0.0, rsum = 0.0, asum = 0.0;
      for (int y = 0; y < ny; ++y) {                                      // (fixed order; zeros for the padding columns)
        sum += part[((int64_t)(0 * ny + y) * 2 + q) * p.Xpad + x];
        rsum += part[((int64_t)(1 * ny + y) * 2 + q) * p.Xpad + x];
        asum += part[((int64_t)(2 * ny + y) * 2 + q) * p.Xpad + x];
      }
      abar = sum / Kp;
      const double rh = rsum / Kp, mhalf = rh * rh;
      if (mhalf > 0.0 && asum > 0.0) {
        const double tails = (2.0 - fabs(abar) / mhalf) * 2.0;            // 1 up to |mean| = 1.5 x the power mean, 0 from 2 x
        const double coherent = (fabs(sum) / asum - 0.5) * 4.0;            // 1 from |sum x| = 0.75 sum |x|, 0 below 0.5
        lam = (tails < 0.0 ? 0.0 : (tails > 1.0 ? 1.0 : tails)) * (coherent < 0.0 ? 0.0 : (coherent > 1.0 ? 1.0 : coherent));
      }
    }
    mean[(int64_t)q * p.Xpad + x] = abar;
    cused[(int64_t)q * p.Xpad + x] = (float)(lam * abar);
  }
}

// ---- the split pass ---------------------------------------------------------------------------------------------------------
// hdr[0] = the power of two the operand was multiplied by, hdr[1] = its inverse.  Grid (ceil(Xpad / 256), k-group chunks).
// cused != NULL: the operand is CENTRED first -- the constant cused[x] (hcentre_kernel) is subtracted from every column.
__global__ __launch_bounds__(256) void split_kernel(const SplitArgs p, const float* __restrict__ X, const float* __restrict__ slots,
                                                    float* __restrict__ hdr, h8* __restrict__ P, const float* __restrict__ cused) {
  const int lane = threadIdx.x & 63;
  const float m = hread_scale(slots, lane);
  int q = 0;
  (void)frexpf(m, &q);                                 // m = f 2^q, f in [0.5, 1)  ->  m 2^(15 - q) in [2^14, 2^15)
  if (cused) q += 1;                                   // (|x - c| <= 2 max|x|)
  const bool fin = m > 0.f && m < 1.5e38f;
  const float scale = fin ? ldexpf(1.f, 15 - q) : 1.f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    hdr[0] = scale;
    hdr[1] = fin ? ldexpf(1.f, q - 15) : 1.f;
  }
  const uint32_t x = blockIdx.x * 256 + threadIdx.x;
  if (x >= p.Xpad) return;
  const bool valid = x < p.X;
  const float ahat0 = cused ? cused[x] : 0.f, ahat1 = (cused && p.period == 2) ? cused[(int64_t)p.Xpad + x] : ahat0;   // (k even / odd)
  const float* src = X + (valid ? hdecomp(x, p.ng, p.dim, p.stride) : 0);
  const uint32_t per = (p.KG + gridDim.y - 1) / gridDim.y;
  const uint32_t kg0 = blockIdx.y * per, kg1 = (kg0 + per < p.KG) ? kg0 + per : p.KG;
  h8* P1 = P + x;
  h8* P2 = P + (int64_t)p.KG * p.Xpad + x;
#pragma unroll 2
  for (uint32_t kg = kg0; kg < kg1; ++kg) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t k = 8 * kg + j;
      v[j] = (valid && k < p.K) ? (src[hkoff(p, k)] - ((j & 1) ? ahat1 : ahat0)) * scale : 0.f;
    }
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)v[j];
      a[j] = h;
      b[j] = (_Float16)(v[j] - (float)h);
    }
    P1[(int64_t)kg * p.Xpad] = a;
    P2[(int64_t)kg * p.Xpad] = b;
  }
}

// ---- the same two passes for K-CONTIGUOUS sources (the innermost K group stride-1 in multiples of 8, everything 16-byte aligned:
// row-major A[m, k]): a thread reads the 32 contiguous bytes of one (column, k-group), 16 lanes a 512-byte run of a column --
// and the split pass transposes through LDS, so that its stores are the same 1 KiB runs of consecutive columns --------------
typedef float hf4 __attribute__((ext_vector_type(4)));

// 16 columns per workgroup, 16 lanes along k per column; grid (Xpad / 16, NY)
__global__ __launch_bounds__(256) void hmean_kc_kernel(const SplitArgs p, const float* __restrict__ X, double* __restrict__ part) {
  const uint32_t x = blockIdx.x * 16 + (threadIdx.x >> 4);
  const uint32_t kq = threadIdx.x & 15;
  const bool two = p.period == 2;
  double s[2] = {0.0, 0.0};
  float r[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
  const uint32_t KGv = p.K >> 3;                                  // whole k-groups of the source (K % 8 == 0 here)
  const uint32_t per = (KGv + gridDim.y - 1) / gridDim.y;
  const uint32_t g0 = blockIdx.y * per, g1 = (g0 + per < KGv) ? g0 + per : KGv;
  if (x < p.X) {
    const float* src = X + hdecomp(x, p.ng, p.dim, p.stride);
    for (uint32_t g = g0 + kq; g < g1; g += 16) {
      const hf4* q = reinterpret_cast<const hf4*>(src + hkoff(p, 8 * g));
      const hf4 v0 = q[0], v1 = q[1];
      const double se = ((double)v0[0] + (double)v0[2]) + ((double)v1[0] + (double)v1[2]), so = ((double)v0[1] + (double)v0[3]) + ((double)v1[1] + (double)v1[3]);
      float re = 0.f, ro = 0.f, ae = 0.f, ao = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const float b0 = fabsf(v0[j]), b1 = fabsf(v0[j + 1]), b2 = fabsf(v1[j]), b3 = fabsf(v1[j + 1]);
        ae += b0 + b2; ao += b1 + b3;
        re += __builtin_sqrtf(b0) + __builtin_sqrtf(b2); ro += __builtin_sqrtf(b1) + __builtin_sqrtf(b3);
      }
      if (two) { s[0] += se; s[1] += so; r[0] += re; r[1] += ro; a1[0] += ae; a1[1] += ao; }
      else { s[0] += se + so; r[0] += re + ro; a1[0] += ae + ao; }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {                              // the 16 lanes of a column, fixed order
      s[q] += __shfl_xor(s[q], d, 64);
      r[q] += __shfl_xor(r[q], d, 64);
      a1[q] += __shfl_xor(a1[q], d, 64);
    }
    if (kq == 0 && x < p.Xpad) {
      part[((int64_t)(0 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = s[q];
      part[((int64_t)(1 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = (double)r[q];
      part[((int64_t)(2 * gridDim.y + blockIdx.y) * 2 + q) * p.Xpad + x] = (double)a1[q];
    }
  }
}

// a workgroup: 64 columns x 16 k-groups; grid (Xpad / 64, ceil(KG / 16))
__global__ __launch_bounds__(256) void split_kc_kernel(const SplitArgs p, const float* __restrict__ X, const float* __restrict__ slots,
                                                       float* __restrict__ hdr, h8* __restrict__ P, const float* __restrict__ cused) {
  __shared__ h8 tile[2][16][65];                                   // [half][k-group][column (+1: the transposing accesses)]
  const int lane = threadIdx.x & 63;
  const float m = hread_scale(slots, lane);
  int q = 0;
  (void)frexpf(m, &q);
  if (cused) q += 1;
  const bool fin = m > 0.f && m < 1.5e38f;
  const float scale = fin ? ldexpf(1.f, 15 - q) : 1.f;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
    hdr[0] = scale;
    hdr[1] = fin ? ldexpf(1.f, q - 15) : 1.f;
  }
  const uint32_t x0 = blockIdx.x * 64, g0 = blockIdx.y * 16;
  const uint32_t kq = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  const uint32_t g = g0 + kq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = r0 + 16 * i, x = x0 + row;
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if (x < p.X && 8 * g < p.K) {
      const hf4* src = reinterpret_cast<const hf4*>(X + hdecomp(x, p.ng, p.dim, p.stride) + hkoff(p, 8 * g));
      const hf4 v0 = src[0], v1 = src[1];
      const float c0 = cused ? cused[x] : 0.f, c1 = (cused && p.period == 2) ? cused[(int64_t)p.Xpad + x] : c0;   // (k even / odd)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = ((j < 4 ? v0[j & 3] : v1[j & 3]) - ((j & 1) ? c1 : c0)) * scale;
        const _Float16 h = (_Float16)xv;
        a[j] = h;
        b[j] = (_Float16)(xv - (float)h);
      }
    }
    tile[0][kq][row] = a;
    tile[1][kq][row] = b;
  }
  __syncthreads();
  const uint32_t xc = threadIdx.x & 63, gq = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t gg = gq + 4 * i;
    if (g0 + gg < p.KG && x0 + xc < p.Xpad) {
      P[(int64_t)(g0 + gg) * p.Xpad + x0 + xc] = tile[0][gg][xc];
      P[((int64_t)p.KG + g0 + gg) * p.Xpad + x0 + xc] = tile[1][gg][xc];
    }
  }
}

// ring depth of the product kernel: as many 16-k stages ([2 halves][2 k-groups][BM + BN columns][16 bytes]) as fit 160 KB of
// LDS beside the tile's C-offset tables, at most 6 (the request counter)
__host__ __device__ constexpr int hring_stages(int ta, int tb) {
  const int stage = 64 * 64 * (ta + tb), tables = 8 * 64 * (ta + tb) + 64;
  const int n = (160 * 1024 - tables) / stage;
  return n > 6 ? 6 : n;
}

// ---- the epilogue both product kernels share: acc[i][j][r] = C[row0 + 32 i + (r & 3) + 8 (r >> 2) + 4 kh][col0 + 32 j + l31] of the
// workgroup's tile (row0 / col0: the wave's corner in it; NW: waves per workgroup) ---------------------------------------------------
template <int TA, int TBW, bool DOT, int NW>
__device__ __forceinline__ void hepilogue(const GettArgs& p, acc16 (&acc)[TA][TBW], float* __restrict__ C,
                                          const float* __restrict__ scale_a, const float* __restrict__ scale_b,
                                          const float* __restrict__ hdrA, const float* __restrict__ hdrB,
                                          const double* __restrict__ meanA, const double* __restrict__ meanB,
                                          float* __restrict__ absmax_out, const int64_t* offCm, const int64_t* offCn, uint32_t m0,
                                          uint32_t n0, int row0, int col0, int64_t Mpad, int64_t Npad, int tid, int lane, int wave) {
  const int l31 = lane & 31, kh = lane >> 5;
  // centred operands (meanA / meanB: the columns' means over k as doubles, zeros for the padding; usedA / usedB: the fp32
  // constants ah, bh the split pass subtracted):
  //   sum_k a b = sum_k (a - ah)(b - bh) + K (ah bbar + bh (abar - ah))
  // -- the MFMA sum is over sign-mixed terms of the size of the operands' FLUCTUATIONS, the part carried by the constants is
  // added here in double precision
  const float unscale = hdrA[1] * hdrB[1];
  const bool centred = meanA != nullptr;
  const int np = p.pad_ == 2 ? 2 : 1;                      // centring constants per parity of k (GettArgs.pad_ = the operands' period)
  const double Kp = (double)p.K / np;
  const float* usedA = centred ? reinterpret_cast<const float*>(meanA + QAMD_GEMMH_USED0(Mpad)) : nullptr;
  const float* usedB = centred ? reinterpret_cast<const float*>(meanB + QAMD_GEMMH_USED0(Npad)) : nullptr;
  double bbar[2][TBW], bh[2][TBW];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < TBW; ++j) {
      const int g = n0 + col0 + 32 * j + l31;
      const bool on = centred && q < np;
      bbar[q][j] = on ? meanB[(int64_t)q * Npad + g] * Kp : 0.0;
      bh[q][j] = on ? (double)usedB[(int64_t)q * Npad + g] * Kp : 0.0;
    }
  if constexpr (DOT) {
    // (a tile's epilogue has the CU to itself -- one workgroup per CU --, so T's values are requested a whole sub-tile row
    // block ahead of their use, with clamped addresses and 0 / 1 masks instead of branches: 64 loads in flight per lane)
    __shared__ double dred[NW];
    float dsum = 0.0f;
    double dcorr = 0.0;
    int64_t ocol[TBW];
    float cmask[TBW];
#pragma unroll
    for (int j = 0; j < TBW; ++j) {
      const int nl = col0 + 32 * j + l31;
      ocol[j] = offCn[nl];                               // (columns past N: the table holds column N - 1)
      cmask[j] = (n0 + nl < p.N) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      float tv[16][TBW];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int64_t orow = offCm[ml];                  // (rows past M: row M - 1)
#pragma unroll
        for (int j = 0; j < TBW; ++j) tv[r][j] = C[orow + ocol[j]];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float rmask = (m0 + ml < p.M) ? 1.0f : 0.0f;
        float trow[TBW];
#pragma unroll
        for (int j = 0; j < TBW; ++j) {
          trow[j] = tv[r][j] * (rmask * cmask[j]);
          dsum = __builtin_fmaf(acc[i][j][r], trow[j], dsum);
        }
        if (centred) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (q >= np) continue;
            const double abar = meanA[(int64_t)q * Mpad + m0 + ml], ah = (double)usedA[(int64_t)q * Mpad + m0 + ml], ad = abar - ah;
#pragma unroll
            for (int j = 0; j < TBW; ++j) dcorr += (ah * bbar[q][j] + ad * bh[q][j]) * (double)trow[j];
          }
        }
      }
    }
    double ds = (double)dsum * (double)unscale + dcorr;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ds += __shfl_down(ds, d, 64);
    if (lane == 0) dred[wave] = ds;
    __syncthreads();
    if (tid == 0) {
      double tot = (dred[0] + dred[1]) + (dred[2] + dred[3]);
      if constexpr (NW == 8) tot += (dred[4] + dred[5]) + (dred[6] + dred[7]);
      reinterpret_cast<double*>(absmax_out)[blockIdx.x] = tot;
    }
    return;
  }
  const float strip = 1.0f / (hread_scale(scale_a, lane) * hread_scale(scale_b, lane));
  const float alpha = unscale * strip;
  float vmax = 0.0f;
#pragma unroll
  for (int i = 0; i < TA; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = row0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (m0 + ml >= p.M) continue;
      const int64_t orow = offCm[ml];
      double ah[2] = {0.0, 0.0}, ad[2] = {0.0, 0.0};
      if (centred) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (q >= np) continue;
          const double abar = meanA[(int64_t)q * Mpad + m0 + ml];
          ah[q] = (double)usedA[(int64_t)q * Mpad + m0 + ml];
          ad[q] = abar - ah[q];
        }
      }
#pragma unroll
      for (int j = 0; j < TBW; ++j) {
        const int nl = col0 + 32 * j + l31;
        if (n0 + nl < p.N) {
          float v = acc[i][j][r] * alpha;
          if (centred) v = (float)((double)v + ((ah[0] * bbar[0][j] + ad[0] * bh[0][j]) + (ah[1] * bbar[1][j] + ad[1] * bh[1][j])) * (double)strip);
          C[orow + offCn[nl]] = v;
          vmax = fmaxf(vmax, fabsf(v));
        }
      }
    }
  }
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, d, 64));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(absmax_out + ((blockIdx.x * NW + wave) % QAMD_SLOTS)), __float_as_uint(vmax));
  }
}

// ---- the product --------------------------------------------------------------------------------------------------------------
// p.tiles_m / tiles_n: the tile grid (the images are padded to it), p.Kloop: K rounded up to 32 (the images' k extent),
// p.M / p.N: the valid extents (epilogue), C addressed through p.dim_m / sc_m / dim_n / sc_n as in gemmk.hip.
template <int TA, int TB, bool DOT>
__global__ __launch_bounds__(256, 1) void gemmh_kernel(const GettArgs p, const char* __restrict__ PA, const char* __restrict__ PB,
                                                       float* __restrict__ C, const float* __restrict__ scale_a,
                                                       const float* __restrict__ scale_b, const float* __restrict__ hdrA,
                                                       const float* __restrict__ hdrB, const double* __restrict__ meanA,
                                                       const double* __restrict__ meanB, float* __restrict__ absmax_out) {
  // (behind an operand's Xpad means: the 3 NY partial rows of the mean pass, then the Xpad constants actually subtracted)
  constexpr int BM = 64 * TA, BN = 64 * TB;
  constexpr int SA = 2 * 2 * BM * 16, SB = 2 * 2 * BN * 16, STAGE = SA + SB;   // bytes of a 16-k stage: [half][2 k-groups][x][16]
  constexpr int NS = hring_stages(TA, TB);                                      // ring depth: what fits 160 KB of LDS (4 .. 6)
  constexpr int NP = TA + TB;                                                   // LDS-DMA pieces per wave and stage
  extern __shared__ __attribute__((aligned(16))) char hsmem[];
  char* stages = hsmem;
  int64_t* offCm = reinterpret_cast<int64_t*>(hsmem + NS * STAGE);
  int64_t* offCn = offCm + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile coordinates: each XCD a contiguous run of the tile sequence, bands of 4 tile rows (as gemmk.hip) ----------------
  const uint32_t per_batch = p.tiles_m * p.tiles_n;
  const uint32_t pid = blockIdx.x;
  uint32_t tm, tn;
  {
    const uint32_t xcd = pid & 7, idx = pid >> 3;
    const uint32_t q = per_batch >> 3, r = per_batch & 7;
    const uint32_t s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const uint32_t band = QAMD_GEMMH_BAND * p.tiles_n;
    const uint32_t first_m = (s / band) * QAMD_GEMMH_BAND;
    const uint32_t gsz = (p.tiles_m - first_m) < QAMD_GEMMH_BAND ? (p.tiles_m - first_m) : QAMD_GEMMH_BAND;
    const uint32_t in_band = s % band;
    tm = first_m + in_band % gsz;
    tn = in_band / gsz;
  }
  const uint32_t m0 = tm * BM, n0 = tn * BN;
  for (int i = tid; i < BM + BN; i += 256) {
    if (i < BM) {
      uint32_t g = m0 + i;
      offCm[i] = hdecomp(g < p.M ? g : p.M - 1, p.nm, p.dim_m, p.sc_m);
    } else {
      uint32_t g = n0 + (i - BM);
      offCn[i - BM] = hdecomp(g < p.N ? g : p.N - 1, p.nn, p.dim_n, p.sc_n);
    }
  }

  // ---- LDS-DMA sources: wave w always fetches (half w & 1, k-group w >> 1) of a stage, every 64-row chunk of it -------------
  const int64_t Mpad = (int64_t)p.tiles_m * BM, Npad = (int64_t)p.tiles_n * BN;
  const int64_t KG = p.Kloop >> 3;
  const int wh = wave & 1, wk = wave >> 1;
  const int64_t stepA = 2 * Mpad * 16, stepB = 2 * Npad * 16;
  const char* baseA0 = PA + (((int64_t)wh * KG + wk) * Mpad + m0) * 16;
  const char* baseB0 = PB + (((int64_t)wh * KG + wk) * Npad + n0) * 16;
  const uint32_t lane16 = 16u * lane;
  const int nsteps = (int)(KG >> 1);          // 16-k steps (even: the images' k extent is a multiple of 32)
  const int dstA = ((wh * 2 + wk) * BM) * 16, dstB = SA + ((wh * 2 + wk) * BN) * 16;

  // piece q of a wave's NP for the stage whose sources are srcA_ / srcB_ into ring slot st_: q < TA: A chunk q, else B chunk q - TA
#define QH_PIECE(q_, st_)                                                                                            \
  do {                                                                                                               \
    if ((q_) < TA) hglds(srcA_ + (q_) * 1024 + lane16, stages + (st_) * STAGE + dstA + (q_) * 1024);                 \
    else hglds(srcB_ + ((q_) - TA) * 1024 + lane16, stages + (st_) * STAGE + dstB + ((q_) - TA) * 1024);             \
  } while (0)
  // sources of stage s_ (past the last stage: the last one again -- a slot nobody reads, but the request counts stay exact)
#define QH_SOURCES(s_)                                                                                               \
  const int64_t sc_ = (s_) < nsteps ? (s_) : nsteps - 1;                                                             \
  const char* srcA_ = baseA0 + sc_ * stepA;                                                                          \
  const char* srcB_ = baseB0 + sc_ * stepB

  acc16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int l31 = lane & 31, kh = lane >> 5;
  const int aoff = (kh * BM + wm * (32 * TA) + l31) * 16;
  const int boff = SA + (kh * BN + wn * (32 * TB) + l31) * 16;

  h8 fa[2][2][TA], fb[2][2][TB];     // [buffer][half][sub-tile]
  // fragments of the stage in ring slot st_ into buffer b_
#define QH_FRAGS(b_, st_)                                                                                            \
  do {                                                                                                               \
    const char* As_ = stages + (st_) * STAGE + aoff;                                                                 \
    const char* Bs_ = stages + (st_) * STAGE + boff;                                                                 \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                  \
      _Pragma("unroll") for (int i = 0; i < TA; ++i) fa[b_][h][i] = *reinterpret_cast<const h8*>(As_ + h * (2 * BM * 16) + i * 512); \
      _Pragma("unroll") for (int j = 0; j < TB; ++j) fb[b_][h][j] = *reinterpret_cast<const h8*>(Bs_ + h * (2 * BN * 16) + j * 512); \
    }                                                                                                                \
  } while (0)

  // ---- prologue: the whole ring requested (stages 0 .. NS - 1), stage 0 awaited --------------------------------------------
#pragma unroll
  for (int s0 = 0; s0 < NS; ++s0) {
    QH_SOURCES(s0);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (q < TA) hglds(srcA_ + q * 1024 + lane16, stages + s0 * STAGE + dstA + q * 1024);
      else hglds(srcB_ + (q - TA) * 1024 + lane16, stages + s0 * STAGE + dstB + (q - TA) * 1024);
    }
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * NP) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  QH_FRAGS(0, 0);

  // One 16-k step on fragment buffer b_ (stage t in ring slot st): the first MFMA; then the ONE barrier of the step -- behind
  // it stage t + 1 has landed for every wave (this wave's pieces: all but the NS - 2 younger requests) and every wave holds
  // stage t in registers, so its slot is free --; then the reads of stage t + 1's fragments and, one piece behind each of the
  // following MFMAs, the request for stage t + NS into slot st.  A stage is requested NS - 1 steps (>= 3400 MFMA cycles) before
  // the barrier that waits for it: with the two-stage ring of 32-k stages this kernel started with it was ONE stage, and the
  // matrix pipes sat idle 40 % of the time waiting for the LDS-DMA (profiles/r06_gemmh_pmc.txt).
#define QH_MFMA(i_, j_, ha_, hb_, b_) \
  acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[b_][ha_][i_], fb[b_][hb_][j_], acc[i_][j_], 0, 0, 0)
#define QH_STEP(b_)                                                                                                  \
  do {                                                                                                               \
    const int stn_ = st + 1 >= NS ? 0 : st + 1;                                                                      \
    QH_SOURCES(t + NS);                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    QH_MFMA(0, 0, 0, 1, b_);                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * NP) : "memory");                                  \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");                                                                                   \
    QH_FRAGS((b_) ^ 1, stn_);                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    _Pragma("unroll") for (int pr = 0; pr < 3; ++pr)                                                                 \
      _Pragma("unroll") for (int i = 0; i < TA; ++i) _Pragma("unroll") for (int j = 0; j < TB; ++j) {                \
        const int n_ = (pr * TA + i) * TB + j;                                                                       \
        if (n_ > 0) {                                                                                                \
          if (pr == 0) QH_MFMA(i, j, 0, 1, b_);                                                                      \
          else if (pr == 1) QH_MFMA(i, j, 1, 0, b_);                                                                 \
          else QH_MFMA(i, j, 0, 0, b_);                                                                              \
          if (n_ - 1 < NP) {                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            switch (n_ - 1) {                                                                                        \
              case 0: QH_PIECE(0, st); break;   case 1: QH_PIECE(1, st); break;                                      \
              case 2: QH_PIECE(2, st); break;   case 3: QH_PIECE(3, st); break;                                      \
              case 4: QH_PIECE(4, st); break;   case 5: QH_PIECE(5, st); break;                                      \
              case 6: QH_PIECE(6, st); break;   default: QH_PIECE(7, st); break;                                     \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
          }                                                                                                          \
        }                                                                                                            \
      }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    st = stn_;                                                                                                       \
    ++t;                                                                                                             \
  } while (0)

  int st = 0, t = 0;
  while (t < nsteps) {
    QH_STEP(0);
    QH_STEP(1);
  }
#undef QH_STEP
#undef QH_MFMA
#undef QH_FRAGS
#undef QH_SOURCES
#undef QH_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  hepilogue<TA, TB, DOT, 4>(p, acc, C, scale_a, scale_b, hdrA, hdrB, meanA, meanB, absmax_out, offCm, offCn, m0, n0, wm * (32 * TA),
                            wn * (32 * TB), Mpad, Npad, tid, lane, wave);
}

// ---- the same product with EIGHT waves in two staggered groups (TA, TB even) ---------------------------------------------------------
// 2 x 4 waves, wave tile (32 TA) x (16 TB), two waves per SIMD (<= 256 registers): group 0 = the upper tile half (waves 0..3),
// group 1 = the lower one, running HALF A STEP behind (one extra barrier at its start, one at group 0's end).  A step is two
// phases with a barrier after each: L -- read the step's fragments, request a stage NS - 1 steps ahead -- and C -- the step's
// 3 TA TB / 2 MFMAs at raised priority.  While one group's waves are in C the SIMDs' other waves are in L: the matrix pipe
// always has a wave issuing, and no fragment is double-buffered (its reads have a whole phase to land).
template <int TA, int TB, bool DOT>
__global__ __launch_bounds__(512, 1) void gemmh8_kernel(const GettArgs p, const char* __restrict__ PA, const char* __restrict__ PB,
                                                       float* __restrict__ C, const float* __restrict__ scale_a,
                                                       const float* __restrict__ scale_b, const float* __restrict__ hdrA,
                                                       const float* __restrict__ hdrB, const double* __restrict__ meanA,
                                                       const double* __restrict__ meanB, float* __restrict__ absmax_out) {
  // (behind an operand's Xpad means: the 3 NY partial rows of the mean pass, then the Xpad constants actually subtracted)
  constexpr int BM = 64 * TA, BN = 64 * TB;
  constexpr int SA = 2 * 2 * BM * 16, SB = 2 * 2 * BN * 16, STAGE = SA + SB;   // bytes of a 16-k stage: [half][2 k-groups][x][16]
  constexpr int NS = hring_stages(TA, TB);                                      // ring depth: what fits 160 KB of LDS (4 .. 6)
  constexpr int TBW = TB / 2, NT = 512;                                         // a wave's sub-tile columns (2 x 4 waves), threads
  constexpr int NPA = TA / 2, NPB = TB / 2, NP = NPA + NPB;                     // LDS-DMA pieces per wave and stage
  static_assert(TA % 2 == 0 && TB % 2 == 0, "even tile shapes only");
  extern __shared__ __attribute__((aligned(16))) char hsmem[];
  char* stages = hsmem;
  int64_t* offCm = reinterpret_cast<int64_t*>(hsmem + NS * STAGE);
  int64_t* offCn = offCm + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;           // wm = the wave's GROUP (0: leads, 1: half a step behind)

  // ---- tile coordinates: each XCD a contiguous run of the tile sequence, bands of 4 tile rows (as gemmk.hip) ----------------
  const uint32_t per_batch = p.tiles_m * p.tiles_n;
  const uint32_t pid = blockIdx.x;
  uint32_t tm, tn;
  {
    const uint32_t xcd = pid & 7, idx = pid >> 3;
    const uint32_t q = per_batch >> 3, r = per_batch & 7;
    const uint32_t s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const uint32_t band = QAMD_GEMMH_BAND * p.tiles_n;
    const uint32_t first_m = (s / band) * QAMD_GEMMH_BAND;
    const uint32_t gsz = (p.tiles_m - first_m) < QAMD_GEMMH_BAND ? (p.tiles_m - first_m) : QAMD_GEMMH_BAND;
    const uint32_t in_band = s % band;
    tm = first_m + in_band % gsz;
    tn = in_band / gsz;
  }
  const uint32_t m0 = tm * BM, n0 = tn * BN;
  for (int i = tid; i < BM + BN; i += NT) {
    if (i < BM) {
      uint32_t g = m0 + i;
      offCm[i] = hdecomp(g < p.M ? g : p.M - 1, p.nm, p.dim_m, p.sc_m);
    } else {
      uint32_t g = n0 + (i - BM);
      offCn[i - BM] = hdecomp(g < p.N ? g : p.N - 1, p.nn, p.dim_n, p.sc_n);
    }
  }

  // ---- LDS-DMA sources: wave w always fetches (half w & 1, k-group w >> 1) of a stage, every 64-row chunk of it -------------
  const int64_t Mpad = (int64_t)p.tiles_m * BM, Npad = (int64_t)p.tiles_n * BN;
  const int64_t KG = p.Kloop >> 3;
  const int wh = wave & 1, wk = (wave >> 1) & 1, ws = wave >> 2;   // ws: chunks ws, ws + 2, .. of the (wh, wk) images
  const int64_t stepA = 2 * Mpad * 16, stepB = 2 * Npad * 16;
  const char* baseA0 = PA + (((int64_t)wh * KG + wk) * Mpad + m0) * 16;
  const char* baseB0 = PB + (((int64_t)wh * KG + wk) * Npad + n0) * 16;
  const uint32_t lane16 = 16u * lane;
  const int nsteps = (int)(KG >> 1);          // 16-k steps (even: the images' k extent is a multiple of 32)
  const int dstA = ((wh * 2 + wk) * BM) * 16, dstB = SA + ((wh * 2 + wk) * BN) * 16;

  // piece q of a wave's NP for the stage whose sources are srcA_ / srcB_ into ring slot st_: q < TA: A chunk q, else B chunk q - TA
#define QH_PIECE(q_, st_)                                                                                            \
  do {                                                                                                               \
    if ((q_) < NPA) {                                                                                                \
      const int c_ = ws + 2 * (q_);                                                                                  \
      hglds(srcA_ + c_ * 1024 + lane16, stages + (st_) * STAGE + dstA + c_ * 1024);                                  \
    } else {                                                                                                         \
      const int c_ = ws + 2 * ((q_) - NPA);                                                                          \
      hglds(srcB_ + c_ * 1024 + lane16, stages + (st_) * STAGE + dstB + c_ * 1024);                                  \
    }                                                                                                                \
  } while (0)
  // sources of stage s_ (past the last stage: the last one again -- a slot nobody reads, but the request counts stay exact)
#define QH_SOURCES(s_)                                                                                               \
  const int64_t sc_ = (s_) < nsteps ? (s_) : nsteps - 1;                                                             \
  const char* srcA_ = baseA0 + sc_ * stepA;                                                                          \
  const char* srcB_ = baseB0 + sc_ * stepB

  acc16 acc[TA][TBW];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TBW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int l31 = lane & 31, kh = lane >> 5;
  const int aoff = (kh * BM + wm * (32 * TA) + l31) * 16;
  const int boff = SA + (kh * BN + wn * (32 * TBW) + l31) * 16;

  h8 fa[2][TA], fb[2][TBW];     // [half][sub-tile]: single-buffered
  // ---- prologue: stages 0 .. NS - 2 requested, stage 0 awaited; group 1 then waits half a step ------------------------------
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) {
    QH_SOURCES(s0);
#pragma unroll
    for (int q = 0; q < NP; ++q) QH_PIECE(q, s0);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NP) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wm == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int st = 0, sp = NS - 1;      // ring slots of stage t and of the stage requested in step t (stage t + NS - 1)
  for (int t = 0; t < nsteps; ++t) {
    // ---- L(t): the step's fragments, the request NS - 1 steps ahead (its slot: stage t - 1's, read by both groups by now)
    {
      const char* As_ = stages + st * STAGE + aoff;
      const char* Bs_ = stages + st * STAGE + boff;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TA; ++i) fa[h][i] = *reinterpret_cast<const h8*>(As_ + h * (2 * BM * 16) + i * 512);
#pragma unroll
        for (int j = 0; j < TBW; ++j) fb[h][j] = *reinterpret_cast<const h8*>(Bs_ + h * (2 * BN * 16) + j * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
      QH_SOURCES(t + NS - 1);
#pragma unroll
      for (int q = 0; q < NP; ++q) QH_PIECE(q, sp);
      __builtin_amdgcn_sched_barrier(0);
      // group 1 ends its L at the barrier that precedes group 0's L(t + 1): its pieces of stage t + 1 must have landed
      if (wm == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NP) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- C(t)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TBW; ++j) {
          const int ha = pr == 1 ? 1 : 0, hb = pr == 0 ? 1 : 0;       // a1 b2, a2 b1, a1 b1
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ha][i], fb[hb][j], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    // group 0 ends its C at the same barrier: its pieces of stage t + 1 landed before anybody reads them
    if (wm == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NP) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    st = st + 1 >= NS ? 0 : st + 1;
    sp = sp + 1 >= NS ? 0 : sp + 1;
  }
  if (wm == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#undef QH_SOURCES
#undef QH_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  hepilogue<TA, TBW, DOT, 8>(p, acc, C, scale_a, scale_b, hdrA, hdrB, meanA, meanB, absmax_out, offCm, offCn, m0, n0, wm * (32 * TA),
                             wn * (32 * TBW), Mpad, Npad, tid, lane, wave);
}

template <int TA, int TB, bool DOT>
static int launch_one(const GettArgs& a, const void* PA, const void* PB, void* C, const void* sa, const void* sb, const void* hdrA,
                      const void* hdrB, const void* meanA, const void* meanB, void* amax, hipStream_t st) {
  constexpr int BM = 64 * TA, BN = 64 * TB;
  const size_t lds = (size_t)hring_stages(TA, TB) * 64 * (BM + BN) + (size_t)(BM + BN) * sizeof(int64_t);
  (void)hipFuncSetAttribute((const void*)gemmh_kernel<TA, TB, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = a.tiles_m * a.tiles_n;
  QAMD_LAUNCH((gemmh_kernel<TA, TB, DOT>), dim3(grid), dim3(256), lds, st, a, (const char*)PA, (const char*)PB, (float*)C,
              (const float*)sa, (const float*)sb, (const float*)hdrA, (const float*)hdrB, (const double*)meanA,
              (const double*)meanB, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int TA, int TB, bool DOT>
static int launch_one8(const GettArgs& a, const void* PA, const void* PB, void* C, const void* sa, const void* sb, const void* hdrA,
                       const void* hdrB, const void* meanA, const void* meanB, void* amax, hipStream_t st) {
  constexpr int BM = 64 * TA, BN = 64 * TB;
  const size_t lds = (size_t)hring_stages(TA, TB) * 64 * (BM + BN) + (size_t)(BM + BN) * sizeof(int64_t);
  (void)hipFuncSetAttribute((const void*)gemmh8_kernel<TA, TB, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = a.tiles_m * a.tiles_n;
  QAMD_LAUNCH((gemmh8_kernel<TA, TB, DOT>), dim3(grid), dim3(512), lds, st, a, (const char*)PA, (const char*)PB, (float*)C,
              (const float*)sa, (const float*)sb, (const float*)hdrA, (const float*)hdrB, (const double*)meanA,
              (const double*)meanB, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace qamdh

using namespace qamdh;

#define QAMD_GEMMH_CASES QH_CASE(4, 4) QH_CASE(3, 4) QH_CASE(4, 3) QH_CASE(3, 3) QH_CASE(2, 4) QH_CASE(4, 2)

// bytes of one operand's split images for free extent padded to ``xpad`` and K padded to ``kpad`` (a multiple of 32)
extern "C" int64_t qamd_gemmh_image_bytes(int64_t xpad, int64_t kpad) { return 2 * (kpad / 8) * xpad * 16; }
// bytes of one operand's column means (doubles) + the partial sums they are built from
// (the centring block: QAMD_GEMMH_PART0 / USED0 above)
extern "C" int64_t qamd_gemmh_mean_bytes(int64_t xpad) { return QAMD_GEMMH_USED0(xpad) * 8 + 2 * xpad * 4; }

// absmax of a strided operand into 64 zeroed slots (callers without exponent slots)
extern "C" int qamd_gemmh_absmax_launch(const SplitArgs* a, const void* X, void* slots, void* stream) {
  if (!a || !X || !slots) return -2;
  if (hipMemsetAsync(slots, 0, QAMD_SLOTS * sizeof(float), (hipStream_t)stream) != hipSuccess) return -4;
  const unsigned gy = a->K < 64 ? a->K : 64;
  QAMD_LAUNCH(habsmax_kernel, dim3((a->X + 255) / 256, gy), dim3(256), 0, (hipStream_t)stream, *a, (const float*)X, (float*)slots);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// X (fp32, free bundle a->dim / a->stride, k stride a->sk or the K groups a->dim_k / a->stride_k) -> P[2][a->KG][a->Xpad][8] f16, hdr[0 .. 1] = scale, 1 / scale.
// mean != NULL (qamd_gemmh_mean_bytes(a->Xpad) bytes: a->Xpad means, the partial sums, the constants subtracted): the operand
// is centred (split_kernel); the product kernel is then handed the same pointer.
extern "C" int qamd_gemmh_split_launch(const SplitArgs* a, const void* X, const void* slots, void* hdr, void* P, void* mean,
                                       void* stream) {
  if (!a || !X || !hdr || !P || a->KG == 0 || a->KG % 4 || a->Xpad < a->X || ((uintptr_t)P & 15) || ((uintptr_t)mean & 7)) return -2;
  const unsigned gx = (a->Xpad + 255) / 256;
  double* part = mean ? (double*)mean + QAMD_GEMMH_PART0(a->Xpad) : nullptr;
  float* cused = mean ? reinterpret_cast<float*>((double*)mean + QAMD_GEMMH_USED0(a->Xpad)) : nullptr;
  if (a->period == 2 && a->K % 2) return -2;
  // k-contiguous sources (16-byte loads along k, LDS-transposed stores): the innermost K group stride-1 in multiples of 8,
  // every other stride and the base pointer 16-byte aligned
  bool kc = ((uintptr_t)X & 15) == 0 && a->K % 8 == 0;
  if (a->nk >= 1) {
    kc = kc && a->stride_k[a->nk - 1] == 1 && a->dim_k[a->nk - 1] % 8 == 0;
    for (int g = 0; g + 1 < a->nk; ++g) kc = kc && a->stride_k[g] % 4 == 0;
  } else {
    kc = kc && a->sk == 1;
  }
  for (int g = 0; g < a->ng; ++g) kc = kc && a->stride[g] % 4 == 0;
  if (mean) {
    if (kc) QAMD_LAUNCH(hmean_kc_kernel, dim3((a->Xpad + 15) / 16, QAMD_GEMMH_NY), dim3(256), 0, (hipStream_t)stream, *a, (const float*)X, part);
    else QAMD_LAUNCH(hmean_kernel, dim3(gx, QAMD_GEMMH_NY), dim3(256), 0, (hipStream_t)stream, *a, (const float*)X, part);
    QAMD_LAUNCH(hcentre_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, *a, (const double*)part, QAMD_GEMMH_NY, (double*)mean,
                cused);
    if (hipGetLastError() != hipSuccess) return -4;
  }
  if (kc) {
    QAMD_LAUNCH(split_kc_kernel, dim3((a->Xpad + 63) / 64, (a->KG + 15) / 16), dim3(256), 0, (hipStream_t)stream, *a, (const float*)X,
                (const float*)slots, (float*)hdr, (h8*)P, (const float*)cused);
    return hipGetLastError() == hipSuccess ? 0 : -4;
  }
  unsigned gy = (8192 + gx - 1) / gx;                 // ~8 K workgroups, at least two k-groups each
  if (gy > a->KG / 2) gy = a->KG / 2;
  if (gy < 1) gy = 1;
  QAMD_LAUNCH(split_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, *a, (const float*)X, (const float*)slots, (float*)hdr,
              (h8*)P, (const float*)cused);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// a->tiles_m / tiles_n = ceil(M / 64 ta), ceil(N / 64 tb); a->Kloop = K rounded up to 32 (>= 64); the images padded to both.
// meanA / meanB: both NULL (operands split as they are) or both the pointers the split launches were given (centred operands).
extern "C" int qamd_gemmh_launch(int ta, int tb, const GettArgs* a, const void* PA, const void* PB, void* C, const void* scale_a,
                                 const void* scale_b, const void* hdrA, const void* hdrB, const void* meanA, const void* meanB,
                                 void* absmax_out, void* stream) {
  if (!a || a->Kloop < 64 || a->Kloop % 32 || a->B != 1 || !hdrA || !hdrB || (!meanA) != (!meanB)) return -2;
#define QH_CASE(TA_, TB_) \
  if (QAMD_GEMMH_W8 && ta == TA_ && tb == TB_ && TA_ % 2 == 0 && TB_ % 2 == 0) return launch_one8<(TA_ % 2 ? 4 : TA_), (TB_ % 2 ? 4 : TB_), false>(*a, PA, PB, C, scale_a, scale_b, hdrA, hdrB, meanA, meanB, absmax_out, (hipStream_t)stream); \
  if (ta == TA_ && tb == TB_) return launch_one<TA_, TB_, false>(*a, PA, PB, C, scale_a, scale_b, hdrA, hdrB, meanA, meanB, absmax_out, (hipStream_t)stream);
  QAMD_GEMMH_CASES
#undef QH_CASE
  return -2;
}

// the product consumed by one inner product with T (C's layout): partial[0 .. tiles) one double per workgroup, finished by
// qamd_gemmk_dot_finish (gemmk.hip)
extern "C" int qamd_gemmh_dot_launch(int ta, int tb, const GettArgs* a, const void* PA, const void* PB, const void* T,
                                     const void* hdrA, const void* hdrB, const void* meanA, const void* meanB, void* partial,
                                     void* stream) {
  if (!a || a->Kloop < 64 || a->Kloop % 32 || a->B != 1 || !hdrA || !hdrB || (!meanA) != (!meanB)) return -2;
#define QH_CASE(TA_, TB_)     \
  if (QAMD_GEMMH_W8 && ta == TA_ && tb == TB_ && TA_ % 2 == 0 && TB_ % 2 == 0) return launch_one8<(TA_ % 2 ? 4 : TA_), (TB_ % 2 ? 4 : TB_), true>(*a, PA, PB, const_cast<void*>(T), nullptr, nullptr, hdrA, hdrB, meanA, meanB, partial, (hipStream_t)stream); \
  if (ta == TA_ && tb == TB_) \
    return launch_one<TA_, TB_, true>(*a, PA, PB, const_cast<void*>(T), nullptr, nullptr, hdrA, hdrB, meanA, meanB, partial, (hipStream_t)stream);
  QAMD_GEMMH_CASES
#undef QH_CASE
  return -2;
}
