// sweep.hip -- streaming "big tensor x small tensor" contraction, aligned fast
// path (the dominant kernel of boundary sweeps / gate application), gfx950 only.
//
//   C[m, n] = alpha * sum_k A[m, k] * W[k, n]
//
// Same mathematics and operand layout as stream.hip, restricted to the case the
// executor's death-ordered layouts produce -- A's stride-1 index is the innermost
// M group and that group is a whole number of 16*V-element chunks -- which lets
// the whole loop be straight-line code the compiler can schedule:
//
//  * V = 16 bytes / sizeof(T) (4 floats, 2 doubles): lane (j = l&15, kq = l>>4)
//    loads one 16-byte vector per k-step: the v_mfma_*_16x16x4 B fragment of V
//    column tiles, 256 contiguous bytes per 16 lanes.
//  * the whole K extent is PS <= 9 k-steps (K <= 36, compile time).  The PS loads of
//    the NEXT chunk are issued as one batch before the current chunk's PS*NT*V
//    MFMAs into the other half of a double-buffered register set.  The loads are
//    inline-asm global_load_dwordx4 so that hipcc does not drain them: one counted
//    `s_waitcnt vmcnt(PS)` per chunk (the PS newest operations are the next chunk's
//    loads), PS KB per wave always in flight, no LDS hop for the big operand.
//  * a workgroup's chunk range never straddles the innermost M group (the host
//    picks the range size as a divisor of the group's chunk count), so chunk base
//    addresses are wave-uniform and advance by a constant: no address arithmetic
//    beyond two 64-bit adds per load.
//  * stores: X = V-wide straight from accumulators (C has the same stride-1 M run);
//    Z = wave-private LDS transpose, then 16-byte stores of contiguous runs
//    (C = [.., m, n_in]).
//  * fused exponent stripping exactly as in stream.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#ifndef QAMD_SWEEP_WAVES
#define QAMD_SWEEP_WAVES 2   // min waves per SIMD the register allocation must allow
#endif
#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct WMfma;
template <> struct WMfma<float> {
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  typedef __attribute__((ext_vector_type(4), aligned(16))) float vec_t;
  typedef unsigned int bits_t;
  static constexpr int V = 4;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};
template <> struct WMfma<double> {
  typedef __attribute__((ext_vector_type(4))) double acc_t;
  typedef __attribute__((ext_vector_type(2), aligned(16))) double vec_t;
  typedef unsigned long long bits_t;
  static constexpr int V = 2;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};

__device__ __forceinline__ void wdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
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

// 16-byte global load the compiler does not count (no s_waitcnt is generated for
// it): wave-uniform 64-bit base in SGPRs + per-lane 32-bit byte offset.
template <typename VT>
__device__ __forceinline__ void gload16(VT& dst, uint32_t voff, uint64_t sbase) {
#ifdef QAMD_ASM_LOADS
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
#else
  typedef const __attribute__((address_space(1))) char* gptr_t;   // global (not flat) address space
#ifndef QAMD_SWEEP_NO_NT  // streamed once: non-temporal (+3-4% measured)
  dst = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) VT*>(reinterpret_cast<gptr_t>(sbase) + voff));
#else
  dst = *reinterpret_cast<const __attribute__((address_space(1))) VT*>(reinterpret_cast<gptr_t>(sbase) + voff);
#endif
#endif
}
// wait until at most N vector-memory operations are outstanding, and tie the
// wait to every buffer register so no consumer can be scheduled above it
template <int N, typename VT, int PS>
__device__ __forceinline__ void gwait(VT (&b)[PS]) {
#ifdef QAMD_ASM_LOADS
  asm volatile("s_waitcnt vmcnt(%c1)" : "+v"(b[0]) : "i"(N));
#pragma unroll
  for (int s = 1; s < PS; ++s) asm volatile("" : "+v"(b[s]));
#endif
}

template <typename T>
__device__ __forceinline__ T wread_scale(const T* slots) {
  if (!slots) return T(1);
  T m = T(0);
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    T v = slots[i];
    m = v > m ? v : m;
  }
  return m > T(0) ? m : T(1);
}

template <typename T, int NT, int PS, bool ZMODE>
__global__ __launch_bounds__(256, QAMD_SWEEP_WAVES) void sweep_kernel(const StreamArgs p, const T* __restrict__ A,
                                                     const T* __restrict__ B, T* __restrict__ C,
                                                     const int64_t* __restrict__ ktab,
                                                     const T* __restrict__ scale_a,
                                                     const T* __restrict__ scale_b,
                                                     T* __restrict__ absmax_out) {
  typedef typename WMfma<T>::acc_t acc_t;
  typedef typename WMfma<T>::vec_t vec_t;
  constexpr int V = WMfma<T>::V;
  constexpr int NPAD = NT * 16;
  constexpr int LDW = NPAD + ((48 - NPAD % 32) % 32);
  constexpr int CH = 16 * V;
  constexpr uint32_t CSTRIDE = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* offCn = reinterpret_cast<int64_t*>(smem);   // [NPAD]
  int64_t* offBn = offCn + NPAD;                        // [NPAD]  (reused as int zoffT[NPAD] in ZMODE)
  int64_t* koffA = offBn + NPAD;                        // [KSP*4] padded to whole panels, -1 = masked
  T* Wl = reinterpret_cast<T*>(koffA + p.Kpad);         // [Kpad][LDW], zero rows beyond K
  T* Zl = Wl + (size_t)p.Kpad * LDW;                    // ZMODE: 4 x [N*CH]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  // ---- stage W and the small tables (once per workgroup) ----------------------
  for (int n = tid; n < NPAD; n += 256) {
    int64_t o[2] = {-1, -1};
    if ((uint32_t)n < p.N) wdecomp2(n, p.nn, p.dim_n, p.sb_n, p.sc_n, o);
    offBn[n] = o[0];
    offCn[n] = o[1];
  }
  for (uint32_t k = tid; k < p.Kpad; k += 256) koffA[k] = (k < p.K) ? ktab[k] : 0;
  __syncthreads();
  {
    const int64_t* ktB = ktab + p.KpadTab;
    const uint32_t total = p.Kpad * NPAD;
    for (uint32_t e = tid; e < total; e += 256) {
      uint32_t k = e / NPAD, n = e - k * NPAD;
      int64_t kb = (k < p.K) ? ktB[k] : -1;
      int64_t nb = offBn[n];
      Wl[k * LDW + n] = (kb >= 0 && nb >= 0) ? B[kb + nb] : T(0);
    }
  }
  __syncthreads();
  int* zoffT = reinterpret_cast<int*>(offBn);
  if constexpr (ZMODE) {
    for (int n = tid; n < NPAD; n += 256) {
      int z = (int)(p.N * CH);  // trash run (any n_in = 0 slot of it)
      if ((uint32_t)n < p.N) {
        uint32_t no = n / p.d_in, ni = n - no * p.d_in;
        z = (int)((no * CH) * p.d_in + ni);
      }
      zoffT[n] = z;
    }
    __syncthreads();
  }
  const T alpha = T(1) / (wread_scale(scale_a) * wread_scale(scale_b));

  // ---- this wave's chunks: first+wave, first+wave+4, ... (range inside one M group)
  const uint32_t blk_first = blockIdx.x * p.chunks_per_wave;   // here: chunks per WORKGROUP
  uint32_t c_end = blk_first + p.chunks_per_wave;
  if (c_end > p.chunks) c_end = p.chunks;
  const uint32_t c_begin = blk_first + wave;
  if (c_begin >= c_end) return;
  const uint32_t my_chunks = (c_end - c_begin + CSTRIDE - 1) / CSTRIDE;

  int64_t o2[2];
  wdecomp2(c_begin * CH, p.nm, p.dim_m, p.sa_m, p.sc_m, o2);
  // wave-uniform byte address of the next chunk to fetch, kept in SGPRs
  uint64_t sbase;
  {
    uint64_t b = (uint64_t)(A + o2[0]);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    sbase = ((uint64_t)hi << 32) | lo;
  }
  int64_t cbase = o2[1];              // C offset of the current chunk's first m
  const int64_t cstep = (int64_t)(CSTRIDE * CH) * p.sc_m_in;

  // per-lane byte offsets of this lane's PS loads, constant over chunks (padded
  // entries point at k = 0: valid memory, multiplied by the zero rows of W)
  uint32_t voff[PS];
#pragma unroll
  for (int s = 0; s < PS; ++s) voff[s] = (uint32_t)((koffA[4 * s + kq] + V * j) * (int64_t)sizeof(T));

  acc_t acc[V][NT];
#pragma unroll
  for (int t = 0; t < V; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};
  T vmax = T(0);
  const T* Wrow = Wl + kq * LDW + j;
  // wave-private transpose tile: N*CH elements + a CH*d_in trash run for the padded rows n >= N
  T* tile = ZMODE ? Zl + (size_t)wave * ((p.N + p.d_in) * CH) : nullptr;
  const uint32_t run_v = ZMODE ? (CH * p.d_in) / V : 1;
  const uint32_t tot_v = ZMODE ? (p.N * CH) / V : 0;

  vec_t bufA[PS], bufB[PS];

  auto issue = [&](vec_t (&dst)[PS]) {
#pragma unroll
    for (int s = 0; s < PS; ++s) gload16(dst[s], voff[s], sbase);
    sbase += (uint64_t)(CSTRIDE * CH * sizeof(T));
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch batch ahead of the MFMA block
  };

  auto compute = [&](vec_t (&cur)[PS]) {
#pragma unroll
    for (int s = 0; s < PS; ++s) {
      T w[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) w[nt] = Wrow[(4 * s) * LDW + nt * 16];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < V; ++t) acc[t][nt] = WMfma<T>::run(w[nt], cur[s][t], acc[t][nt]);
    }
    if constexpr (ZMODE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int z = zoffT[nt * 16 + WMfma<T>::row(lane, r)];   // rows n >= N land in the trash run
#pragma unroll
          for (int t = 0; t < V; ++t) tile[z + (V * j + t) * (int)p.d_in] = acc[t][nt][r] * alpha;
        }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t q = lane; q < tot_v; q += 64) {
        {
          uint32_t no = q / run_v, wv = q - no * run_v;
          vec_t o = *reinterpret_cast<const vec_t*>(tile + (size_t)q * V);
#pragma unroll
          for (int e = 0; e < V; ++e) {
            T a = o[e] < T(0) ? -o[e] : o[e];
            vmax = a > vmax ? a : vmax;
          }
#ifndef QAMD_SWEEP_NO_NT  // streamed once: non-temporal (+3-4% measured)
          __builtin_nontemporal_store(o, reinterpret_cast<vec_t*>(C + cbase + offCn[no * p.d_in] + (int64_t)wv * V));
#else
          *reinterpret_cast<vec_t*>(C + cbase + offCn[no * p.d_in] + (int64_t)wv * V) = o;
#endif
        }
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      T* cp = C + cbase + V * j;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nt * 16 + WMfma<T>::row(lane, r);
          vec_t o;
#pragma unroll
          for (int t = 0; t < V; ++t) {
            o[t] = acc[t][nt][r] * alpha;
            T a = o[t] < T(0) ? -o[t] : o[t];
            vmax = a > vmax ? a : vmax;
          }
          const int64_t on = offCn[n];
          if (on >= 0) *reinterpret_cast<vec_t*>(cp + on) = o;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < V; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};
    cbase += cstep;
  };

  // ---- software pipeline over chunks, double-buffered registers -------------------
  issue(bufA);
  uint32_t u = 0;
  for (; u + 2 <= my_chunks; u += 2) {
    issue(bufB);                 // chunk u+1
    gwait<PS>(bufA);             // the PS newest VMEM ops are bufB's loads -> bufA has landed
    compute(bufA);               // chunk u
    if (u + 2 < my_chunks) {
      issue(bufA);               // chunk u+2
      gwait<PS>(bufB);
    } else {
      gwait<0>(bufB);
    }
    compute(bufB);               // chunk u+1
  }
  if (u < my_chunks) {
    gwait<0>(bufA);
    compute(bufA);
  }

  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      T o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0)
      atomicMax(reinterpret_cast<typename WMfma<T>::bits_t*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS),
                WMfma<T>::bits(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <typename T, int NT, int PS, bool ZMODE>
static int launch_sweep_k(const StreamArgs& a, const void* A, const void* B, void* C, const void* ktab,
                          const void* sa, const void* sb, void* amax, hipStream_t st) {
  constexpr int V = WMfma<T>::V;
  constexpr int NPAD = NT * 16;
  constexpr int LDW = NPAD + ((48 - NPAD % 32) % 32);
  size_t lds = (size_t)(2 * NPAD + a.Kpad) * 8 + (size_t)a.Kpad * LDW * sizeof(T);
  if (ZMODE) lds += (size_t)4 * (a.N + a.d_in) * 16 * V * sizeof(T);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)sweep_kernel<T, NT, PS, ZMODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  QAMD_LAUNCH((sweep_kernel<T, NT, PS, ZMODE>), dim3(a.grid), dim3(256), lds, st, a, (const T*)A, (const T*)B,
              (T*)C, (const int64_t*)ktab, (const T*)sa, (const T*)sb, (T*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T, int NT, bool ZMODE>
static int launch_sweep_p(int PS, const StreamArgs& a, const void* A, const void* B, void* C, const void* ktab,
                          const void* sa, const void* sb, void* amax, hipStream_t st) {
  switch (PS) {
    case 1: return launch_sweep_k<T, NT, 1, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 2: return launch_sweep_k<T, NT, 2, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 3: return launch_sweep_k<T, NT, 3, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 4: return launch_sweep_k<T, NT, 4, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 6: return launch_sweep_k<T, NT, 6, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 9: return launch_sweep_k<T, NT, 9, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    default: return -1;
  }
}

template <typename T, bool ZMODE>
static int launch_sweep_n(int PS, const StreamArgs& a, const void* A, const void* B, void* C, const void* ktab,
                          const void* sa, const void* sb, void* amax, hipStream_t st) {
  switch (a.NT) {
    case 1: return launch_sweep_p<T, 1, ZMODE>(PS, a, A, B, C, ktab, sa, sb, amax, st);
    case 2: return launch_sweep_p<T, 2, ZMODE>(PS, a, A, B, C, ktab, sa, sb, amax, st);
    case 3: return launch_sweep_p<T, 3, ZMODE>(PS, a, A, B, C, ktab, sa, sb, amax, st);
    case 4: return launch_sweep_p<T, 4, ZMODE>(PS, a, A, B, C, ktab, sa, sb, amax, st);
    default: return -1;
  }
}

// PS: k-steps per panel; a->Kpad must be a multiple of 4*PS.
extern "C" int QAMD_SWEEP_ENTRY(int PS, const StreamArgs* a, const void* A, const void* B, void* C,
                                const void* ktab, const void* scale_a, const void* scale_b,
                                void* absmax_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (a->zmode) return launch_sweep_n<QAMD_SWEEP_T, true>(PS, *a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  return launch_sweep_n<QAMD_SWEEP_T, false>(PS, *a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
}
