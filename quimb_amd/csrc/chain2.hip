// chain2.hip -- TWO consecutive "big tensor x small tensor" contractions fused into
// one pass over the big tensor (gfx950 only).
//
//   X[x, y, v, m] = sum_k1      A[k1, v, m] * W1[k1, (x, y)]
//   C[x, n2,  m]  = sum_{y, v}  X[x, y, v, m] * W2[(y, v), n2]
//
// This is a pair of adjacent site absorptions of a 2D boundary sweep
// (quimb/tensor/tn2d/core.py:1402 applied twice): the first contracts (h, v_j) and
// produces (x = v'_j, y = h'); the second contracts y together with the NEXT bond v =
// v_{j+1} that was merely carried along by the first.  Executed step by step, the
// 6^11-element intermediate X is written to and re-read from HBM; here it only ever
// exists in a wave-private LDS tile, so the pair moves the big tensor through HBM
// once (read A, write C) for twice the FLOPs: the arithmetic intensity doubles (9 -> 18
// FLOP/B at D = 6) and the pair becomes balanced between the HBM and MFMA roofs.
//
// Structure (all index sizes = D, compile time):
//  * W1, W2 pre-packed dense by the host ([k1][x*D + y] and [y*D + v][n2]) and staged
//    into LDS once per workgroup.
//  * one wave = one chunk of CH = 16*V values of m at a time, private LDS tile
//    Xt[x][k2 = y*D + v][m]  (D regions of K2*CH elements).
//  * stage 1, for every v: D1_v = W1^T . A[:, v, chunk] on v_mfma_*_16x16x4 with A read
//    from registers that were prefetched one whole chunk ahead (KS1*D vector loads per
//    lane); the accumulators are scattered into Xt.
//  * stage 2, for every x: D2_x = W2^T . Xt[x] with B fragments read from LDS; the result
//    overwrites region x IN PLACE in the layout [n2_out][m][n2_in].
//  * copy-out: for every n2_out the run C[n2_out, chunk, x, n2_in] is contiguous in HBM:
//    coalesced stores straight from the tile.  alpha / absmax as in stream.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct CMfma;
template <> struct CMfma<float> {
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  typedef unsigned int bits_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};
template <> struct CMfma<double> {
  typedef __attribute__((ext_vector_type(4))) double acc_t;
  typedef unsigned long long bits_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};

template <typename T, int V> struct CVec {
  typedef T type __attribute__((ext_vector_type(V), aligned(sizeof(T) * V)));
};
template <typename T> struct CVec<T, 1> { typedef T type; };

template <typename T, int V>
__device__ __forceinline__ void cload(T (&d)[V], uint64_t sbase, uint32_t voff) {
  typedef const __attribute__((address_space(1))) char* gptr_t;
  if constexpr (V == 1) {
    d[0] = *reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<gptr_t>(sbase) + voff);
  } else {
    typedef typename CVec<T, V>::type vt;
    vt v = *reinterpret_cast<const __attribute__((address_space(1))) vt*>(reinterpret_cast<gptr_t>(sbase) + voff);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = v[i];
  }
}

__device__ __forceinline__ void cdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
                                         const int64_t* s2, int64_t* o) {
  int64_t o1 = 0, o2 = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    o1 += (int64_t)r * s1[g];
    o2 += (int64_t)r * s2[g];
    idx = q;
  }
  o[0] = o1;
  o[1] = o2;
}

template <typename T>
__device__ __forceinline__ T cread_scale(const T* slots) {
  if (!slots) return T(1);
  T m = T(0);
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    T v = slots[i];
    m = v > m ? v : m;
  }
  return m > T(0) ? m : T(1);
}

constexpr int cpitch(int n) { return n + ((48 - n % 32) % 32); }

template <typename T, int D, int V>
__global__ __launch_bounds__(256, (V == 1 ? 2 : 1)) void chain2_kernel(const Chain2Args p, const T* __restrict__ A,
                                                         const T* __restrict__ W1p, const T* __restrict__ W2p,
                                                         T* __restrict__ C, const int64_t* __restrict__ offK1,
                                                         const int64_t* __restrict__ offCo,
                                                         const T* __restrict__ scale_a, const T* __restrict__ scale_1,
                                                         const T* __restrict__ scale_2, T* __restrict__ absmax_out) {
  typedef typename CMfma<T>::acc_t acc_t;
  constexpr int K1 = D * D, KS1 = (K1 + 3) / 4, K1P = KS1 * 4;
  constexpr int N1 = D * D, NT1 = (N1 + 15) / 16, LD1 = cpitch(NT1 * 16);
  constexpr int K2 = D * D, KS2 = (K2 + 3) / 4, K2P = KS2 * 4;
  constexpr int N2 = D * D, NT2 = (N2 + 15) / 16, LD2 = cpitch(NT2 * 16);
  constexpr int CH = 16 * V;
  constexpr int RS = K2P * CH;          // elements per x-region of the wave tile (>= K2*CH and >= N2*CH)
  constexpr uint32_t CSTRIDE = 4;
  static_assert(N2 <= K2P, "in-place region too small");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* W1l = reinterpret_cast<T*>(smem);                 // [K1P][LD1]
  T* W2l = W1l + K1P * LD1;                            // [K2P][LD2]
  T* Xall = W2l + K2P * LD2;                           // 4 waves x [D][RS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  for (int e = tid; e < K1P * (NT1 * 16); e += 256) {
    int k = e / (NT1 * 16), n = e - k * (NT1 * 16);
    W1l[k * LD1 + n] = (k < K1 && n < N1) ? W1p[k * N1 + n] : T(0);
  }
  for (int e = tid; e < K2P * (NT2 * 16); e += 256) {
    int k = e / (NT2 * 16), n = e - k * (NT2 * 16);
    W2l[k * LD2 + n] = (k < K2 && n < N2) ? W2p[k * N2 + n] : T(0);
  }
  __syncthreads();
  const T alpha = T(1) / (cread_scale(scale_a) * cread_scale(scale_1) * cread_scale(scale_2));

  const uint32_t blk_first = blockIdx.x * p.chunks_per_block;
  uint32_t c_end = blk_first + p.chunks_per_block;
  if (c_end > p.chunks) c_end = p.chunks;
  const uint32_t c_begin = blk_first + wave;
  if (c_begin >= c_end) return;
  const uint32_t my_chunks = (c_end - c_begin + CSTRIDE - 1) / CSTRIDE;

  int64_t o2[2];
  cdecomp2(c_begin * CH, p.nm, p.dim_m, p.sa_m, p.sc_m, o2);
  uint64_t sbase;
  {
    uint64_t b = (uint64_t)(A + o2[0]);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    sbase = ((uint64_t)hi << 32) | lo;
  }
  int64_t cbase = o2[1];
  const int64_t cstep = (int64_t)(CSTRIDE * CH) * (D * D);   // C stride of m is D*D (block [x][n2_in])
  const uint32_t svb = (uint32_t)(p.sa_v * (int64_t)sizeof(T));

  // per-lane byte offsets of the k1 rows this lane loads (padded rows -> row 0, zero W1 rows)
  uint32_t koff[KS1];
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    int k = 4 * s + kq;
    koff[s] = (uint32_t)(((k < K1 ? offK1[k] : offK1[0]) + V * j) * (int64_t)sizeof(T));
  }

  T* Xt = Xall + (size_t)wave * (D * RS);
  // W fragments for both stages stay in registers for the whole kernel
  T wf1[KS1][NT1], wf2[KS2][NT2];
#pragma unroll
  for (int s = 0; s < KS1; ++s)
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) wf1[s][nt] = W1l[(4 * s + kq) * LD1 + nt * 16 + j];
#pragma unroll
  for (int s = 0; s < KS2; ++s)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) wf2[s][nt] = W2l[(4 * s + kq) * LD2 + nt * 16 + j];
  T areg[D][KS1][V];
  T vmax = T(0);

#ifdef QAMD_CHAIN2_ABLATION   // debugging builds only: runtime ablation bits cost scalar branches
  const uint32_t abl = p.ablate;
#else
  constexpr uint32_t abl = 0;
#endif
  auto issue = [&]() {
    if (abl & 4) return;
    // wave-uniform base per v (SGPR pair) + the 32-bit per-lane row offset: KS1 address
    // VGPRs in total, global_load ... v_off, s[base] addressing
#pragma unroll
    for (int v = 0; v < D; ++v) {
      const uint64_t bv = sbase + (uint64_t)v * svb;
#pragma unroll
      for (int s = 0; s < KS1; ++s) cload<T, V>(areg[v][s], bv, koff[s]);
    }
    sbase += (uint64_t)(CSTRIDE * CH * sizeof(T));
  };

  issue();
  for (uint32_t u = 0; u < my_chunks; ++u) {
    // ---- stage 1: X[x, y, v, chunk] for every v ------------------------------------
#pragma unroll
    for (int v = 0; v < D; ++v) {
      acc_t acc[V][NT1];
#pragma unroll
      for (int t = 0; t < V; ++t)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int t = 0; t < V; ++t) acc[t][nt] = CMfma<T>::run(wf1[s][nt], areg[v][s][t], acc[t][nt]);
      }
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n1 = nt * 16 + CMfma<T>::row(lane, r);
          const bool ok = n1 < N1;
          const int x = ok ? n1 / D : 0, y = ok ? n1 - (n1 / D) * D : 0;
          T* dst = Xt + x * RS + (y * D + v) * CH + V * j;
          if (ok && !(abl & 2)) {
#pragma unroll
            for (int t = 0; t < V; ++t) dst[t] = acc[t][nt][r];
          } else {
            asm volatile("" :: "v"(acc[0][nt][r]));
          }
        }
    }
    // the A registers are free: prefetch the next chunk behind stage 2 and the copy-out
    if (u + 1 < my_chunks) issue();
    __builtin_amdgcn_wave_barrier();

    // ---- stage 2: for every x, in place ---------------------------------------------
    T bx[2][KS2][V];
    auto load_b = [&](T (&dst)[KS2][V], int x) {
      const T* xr = Xt + x * RS + kq * CH + V * j;
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        const bool ok = (K2P == K2) || (4 * s + kq < K2);
#pragma unroll
        for (int t = 0; t < V; ++t) dst[s][t] = ok ? xr[(4 * s) * CH + t] : T(0);
      }
    };
    load_b(bx[0], 0);
#pragma unroll
    for (int x = 0; x < ((abl & 8) ? 0 : D); ++x) {
      if (x + 1 < D) load_b(bx[(x + 1) & 1], x + 1);   // next x's fragments behind this x's MFMAs
      acc_t acc[V][NT2];
#pragma unroll
      for (int t = 0; t < V; ++t)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
          for (int t = 0; t < V; ++t) acc[t][nt] = CMfma<T>::run(wf2[s][nt], bx[x & 1][s][t], acc[t][nt]);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n2 = nt * 16 + CMfma<T>::row(lane, r);
          const bool ok = n2 < N2;
          const int no = ok ? n2 / D : 0, ni = ok ? n2 - (n2 / D) * D : 0;
          T* dst = Xt + x * RS + (no * CH + V * j) * D + ni;
          if (ok) {
#pragma unroll
            for (int t = 0; t < V; ++t) dst[t * D] = acc[t][nt][r] * alpha;
          }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- copy-out: C[n2_out][chunk][x][n2_in], contiguous runs of CH*D*D ---------------
    constexpr int EW = (D % 4 == 0 && sizeof(T) == 4) ? 4 : (D % 2 == 0 ? 2 : 1);   // elements per store
    typedef typename CVec<T, EW>::type evec_t;
    // flattened over (n2_out, run element): exactly TOT/64 stores per lane when 64 | TOT,
    // so the store count is static and the next chunk's loads are waited for by count
    constexpr int RUNV = CH * D * D / EW;          // vectors per n2_out run
    constexpr int TOT = D * RUNV;
    constexpr int NIT = (TOT + 63) / 64;
    int64_t co[D];
#pragma unroll
    for (int no = 0; no < D; ++no) co[no] = offCo[no];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = lane + 64 * it;
      if (TOT % 64 == 0 || q < TOT) {
        const int no = q / RUNV, e = (q - no * RUNV) * EW;
        const int ml = e / (D * D), rem = e - ml * (D * D);
        const int x = rem / D, ni = rem - x * D;
        evec_t val = *reinterpret_cast<const evec_t*>(Xt + x * RS + (no * CH + ml) * D + ni);
        if constexpr (EW == 1) {
          T a = val < T(0) ? -val : val;
          vmax = a > vmax ? a : vmax;
        } else {
#pragma unroll
          for (int i = 0; i < EW; ++i) {
            T a = val[i] < T(0) ? -val[i] : val[i];
            vmax = a > vmax ? a : vmax;
          }
        }
        int64_t cof = co[0];
#pragma unroll
        for (int n = 1; n < D; ++n) cof = (no == n) ? co[n] : cof;
        if (!(abl & 1)) *reinterpret_cast<evec_t*>(C + cbase + cof + e) = val;
      }
    }
    __builtin_amdgcn_wave_barrier();
    cbase += cstep;
  }

  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      T o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0)
      atomicMax(reinterpret_cast<typename CMfma<T>::bits_t*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS),
                CMfma<T>::bits(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <typename T, int D, int V>
static int launch_chain2_dv(const Chain2Args& a, const void* A, const void* W1p, const void* W2p, void* C,
                            const void* offK1, const void* offCo, const void* sa, const void* s1, const void* s2,
                            void* amax, hipStream_t st) {
  constexpr int K1P = ((D * D + 3) / 4) * 4, LD1 = cpitch(((D * D + 15) / 16) * 16);
  constexpr int CH = 16 * V;
  size_t lds = (size_t)(2 * K1P * LD1 + 4 * D * K1P * CH) * sizeof(T);
  if (lds > 160 * 1024) return -2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)chain2_kernel<T, D, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  QAMD_LAUNCH((chain2_kernel<T, D, V>), dim3(a.grid), dim3(256), lds, st, a, (const T*)A, (const T*)W1p,
              (const T*)W2p, (T*)C, (const int64_t*)offK1, (const int64_t*)offCo, (const T*)sa, (const T*)s1,
              (const T*)s2, (T*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T>
static int launch_chain2_t(int D, int V, const Chain2Args& a, const void* A, const void* W1p, const void* W2p,
                           void* C, const void* offK1, const void* offCo, const void* sa, const void* s1,
                           const void* s2, void* amax, hipStream_t st) {
#define QAMD_C2(DD, VV) \
  if (D == DD && V == VV) return launch_chain2_dv<T, DD, VV>(a, A, W1p, W2p, C, offK1, offCo, sa, s1, s2, amax, st)
  QAMD_C2(2, 2); QAMD_C2(3, 2); QAMD_C2(4, 2); QAMD_C2(5, 2); QAMD_C2(6, 2);
  QAMD_C2(2, 1); QAMD_C2(3, 1); QAMD_C2(4, 1); QAMD_C2(5, 1); QAMD_C2(6, 1); QAMD_C2(7, 1);
#undef QAMD_C2
  return -2;
}

// chunk size the kernel uses for (dtype, D); 0 = unsupported
extern "C" int qamd_chain2_chunk(int dtype, int D) {
  // 16-element chunks keep the LDS footprint at two workgroups per CU, which measured faster
  // than 32 (0.94 vs 1.10 ms per 6^9 pair)
  if (dtype == 0) return (D >= 2 && D <= 7) ? 16 : 0;
  if (dtype == 1) return (D >= 2 && D <= 6) ? 16 : 0;
  return 0;
}

extern "C" int qamd_chain2_launch(int dtype, int D, const Chain2Args* a, const void* A, const void* W1p,
                                  const void* W2p, void* C, const void* offK1, const void* offCo,
                                  const void* scale_a, const void* scale_1, const void* scale_2,
                                  void* absmax_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int ch = qamd_chain2_chunk(dtype, D);
  if (!ch) return -2;
  if (dtype == 0)
    return launch_chain2_t<float>(D, ch / 16, *a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
  return launch_chain2_t<double>(D, 1, *a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
}
