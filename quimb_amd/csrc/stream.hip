// stream.hip -- "big tensor x small tensor" pairwise contraction, HBM-roofline
// oriented, CDNA4 / gfx950 only.
//
//   C[m, n] = alpha * sum_k A[m, k] * W[k, n]       (M huge, K*N small)
//
// This is THE dominant step shape of tensor-network contraction: a boundary /
// state tensor absorbing a site tensor or a gate (quimb/tensor/tn2d/core.py:1402
// row absorption; Tensor.gate quimb/tensor/tensor_core.py:3152-3159).  The small
// operand W is gathered once per workgroup into LDS; the big operand is streamed
// straight from HBM into MFMA B-operand registers (no LDS hop): the stride-1
// index of A is an M index, so lane (j = l&15, kq = l>>4) loads V consecutive m
// for k = 4s+kq -- exactly the v_mfma_*_16x16x4 B-fragment layout, V tiles at a
// time, 16*V*sizeof(T) contiguous bytes per 16 lanes.  Loads run RING k-steps
// ahead of the MFMAs (across chunk boundaries); waves are independent (no
// workgroup barrier in the stream loop).
//
// Two store paths:
//  * X: C's stride-1 index is the same M group -> D = W^T . A^T has m along lanes,
//    the V accumulator tiles store V-wide vectors straight from registers.
//  * Z: C's stride-1 index is an N group of size d_in directly inside the M run
//    (C[.., m, n_in]) -- what the executor's death-ordered layouts produce.  The
//    wave transposes its 16V x N tile through a private LDS buffer and writes
//    n_out contiguous runs of 16V*d_in elements with 16-byte stores.
//
// Exponent stripping is fused: alpha = 1/(max|A| * max|W|) comes from device
// slots written by the producers' epilogues, and this kernel's own max|C| is
// reduced into its output slots -- no extra pass over the tensor
// (reference semantics: quimb/tensor/tensor_core.py:330-340).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct SMfma;
template <> struct SMfma<float> {
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  typedef unsigned int bits_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ bits_t bits(float v) { return __float_as_uint(v); }
};
template <> struct SMfma<double> {
  typedef __attribute__((ext_vector_type(4))) double acc_t;
  typedef unsigned long long bits_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ bits_t bits(double v) { return (bits_t)__double_as_longlong(v); }
};

template <typename T, int V> struct SVec {
  typedef T type __attribute__((ext_vector_type(V), aligned(sizeof(T) * V)));
};
template <typename T> struct SVec<T, 1> { typedef T type; };

template <typename T, int V>
__device__ __forceinline__ void vload(T (&d)[V], const T* p) {
  if constexpr (V == 1) {
    d[0] = *p;
  } else {
    typename SVec<T, V>::type v = *reinterpret_cast<const typename SVec<T, V>::type*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = v[i];
  }
}
template <typename T, int V>
__device__ __forceinline__ void vstore(T* p, const T (&s)[V]) {
  if constexpr (V == 1) {
    *p = s[0];
  } else {
    typename SVec<T, V>::type v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = s[i];
    *reinterpret_cast<typename SVec<T, V>::type*>(p) = v;
  }
}

__device__ __forceinline__ void sdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
                                         const int64_t* s2, int64_t& o1, int64_t& o2) {
  o1 = 0;
  o2 = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    o1 += (int64_t)r * s1[g];
    o2 += (int64_t)r * s2[g];
    idx = q;
  }
}

// reduce the producer's slots to one scale; slots hold max|x| (0 => treat as 1)
template <typename T>
__device__ __forceinline__ T read_scale(const T* slots) {
  if (!slots) return T(1);
  T m = T(0);
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    T v = slots[i];
    m = v > m ? v : m;
  }
  return m > T(0) ? m : T(1);
}

template <typename T, int V, int NT, int RING, bool ZMODE>
__global__ __launch_bounds__(256) void stream_kernel(const StreamArgs p, const T* __restrict__ A,
                                                      const T* __restrict__ B, T* __restrict__ C,
                                                      const int64_t* __restrict__ ktab,
                                                      const T* __restrict__ scale_a,
                                                      const T* __restrict__ scale_b,
                                                      T* __restrict__ absmax_out) {
  typedef typename SMfma<T>::acc_t acc_t;
  constexpr int NPAD = NT * 16;
  constexpr int LDW = NPAD + ((48 - NPAD % 32) % 32);
  constexpr int CH = 16 * V;             // m per chunk
  constexpr int EV = 16 / sizeof(T);     // elements per 16-byte vector
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* offCn = reinterpret_cast<int64_t*>(smem);        // [NPAD]
  int64_t* offBn = offCn + NPAD;                             // [NPAD]
  int64_t* koffA = offBn + NPAD;                             // [Kpad]
  T* Wl = reinterpret_cast<T*>(koffA + p.Kpad);              // [Kpad][LDW]
  T* Zl = Wl + (size_t)p.Kpad * LDW;                         // ZMODE: 4 x [N*CH] wave-private tiles
  int64_t* gb = reinterpret_cast<int64_t*>(Zl + (size_t)4 * p.N * CH);   // c_break: C offsets of the group starts this
                                                                          // workgroup's rows touch ([p.zb_groups])

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  // ---- stage W and the small offset tables (once per workgroup) ------------
  for (int n = tid; n < NPAD; n += 256) {
    int64_t ob = -1, oc = -1;
    if ((uint32_t)n < p.N) sdecomp2(n, p.nn, p.dim_n, p.sb_n, p.sc_n, ob, oc);
    offBn[n] = ob;
    offCn[n] = oc;
  }
  for (uint32_t k = tid; k < p.Kpad; k += 256) koffA[k] = ktab[k];
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
  if constexpr (ZMODE) {
    int* zt = reinterpret_cast<int*>(offBn);
    for (int n = tid; n < NPAD; n += 256) {
      int z = -1;
      if ((uint32_t)n < p.N) {
        uint32_t no = n / p.d_in, ni = n - no * p.d_in;
        z = (int)((no * CH) * p.d_in + ni);
      }
      zt[n] = z;
    }
    __syncthreads();
  }

  const T alpha = T(1) / (read_scale(scale_a) * read_scale(scale_b));
  // c_break (ZMODE): the C offset of the first row of every innermost-M-group piece this workgroup can meet, tabulated once
  // -- the store path then needs no division or mixed-radix decomposition per chunk
  uint32_t zq_first = 0;
  if constexpr (ZMODE) {
    if (p.c_break) {
      zq_first = (blockIdx.x * (p.chunks_per_wave * 4) * CH) / p.l_in;
      for (uint32_t t = tid; t < p.zb_groups; t += 256) {
        const uint32_t m = (zq_first + t) * p.l_in;
        int64_t oa_, oc = 0;
        if (m < p.M) sdecomp2(m, p.nm, p.dim_m, p.sa_m, p.sc_m, oa_, oc);
        gb[t] = oc;
      }
      __syncthreads();
    }
  }

  // ---- this wave's chunk range ------------------------------------------------
  const uint32_t wglob = blockIdx.x * 4 + wave;
  // the 4 waves of a workgroup interleave over the workgroup's chunk range (wave w
  // takes chunks first+w, first+w+4, ...), so together they touch 4 consecutive
  // chunks -- 4x the contiguous bytes per k-row -- at about the same time
  constexpr uint32_t CSTRIDE = 4;
  const uint32_t blk_first = blockIdx.x * (p.chunks_per_wave * CSTRIDE);
  uint32_t c_end = blk_first + p.chunks_per_wave * CSTRIDE;
  if (c_end > p.chunks) c_end = p.chunks;
  const uint32_t c_begin = blk_first + wave;
  if (c_begin >= c_end) return;
  const uint32_t KS = p.KS;
  const uint32_t my_chunks = (c_end - c_begin + CSTRIDE - 1) / CSTRIDE;
  const uint32_t g_total = my_chunks * KS;
  const bool aligned = p.aligned != 0;
  const uint32_t inner_chunks = p.inner_chunks;  // chunks per innermost M group (aligned mode)

  acc_t acc[V][NT];
#pragma unroll
  for (int t = 0; t < V; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};

  // ---- load cursor (runs RING steps ahead of the compute cursor) ---------------
  uint32_t ld_chunk = c_begin, ld_s = 0, ld_in = aligned ? c_begin % inner_chunks : 0;
  int64_t offA_ld = -1, dummy;
  auto seek_load = [&]() {
    uint32_t m = ld_chunk * CH + (aligned ? 0 : V * j);
    offA_ld = -1;
    if (ld_chunk < c_end && m < p.M) {
      sdecomp2(m, p.nm, p.dim_m, p.sa_m, p.sc_m, offA_ld, dummy);
      if (aligned) offA_ld += V * j;
    }
  };
  seek_load();
  T areg[RING][V];

  auto issue_load = [&](T (&dst)[V]) {
#pragma unroll
    for (int t = 0; t < V; ++t) dst[t] = T(0);
    int64_t ko = koffA[4 * ld_s + kq];
    if (offA_ld >= 0 && ko >= 0) vload<T, V>(dst, A + offA_ld + ko);
    if (++ld_s == KS) {
      ld_s = 0;
      ld_chunk += CSTRIDE;
      ld_in += CSTRIDE;
      if (aligned && ld_in < inner_chunks && ld_chunk < c_end) {
        offA_ld += CSTRIDE * CH;  // same contiguous run
      } else {
        ld_in = aligned ? ld_chunk % inner_chunks : 0;
        seek_load();
      }
    }
  };

#pragma unroll
  for (int u = 0; u < RING; ++u) {
    if ((uint32_t)u < g_total) issue_load(areg[u]);
  }

  // ---- compute / store cursor ----------------------------------------------------
  uint32_t cp_chunk = c_begin, cp_s = 0, cp_in = aligned ? c_begin % inner_chunks : 0;
  int64_t cbase = -1;  // aligned: C offset of the chunk's first m
  if (aligned) sdecomp2(c_begin * CH, p.nm, p.dim_m, p.sa_m, p.sc_m, dummy, cbase);
  T vmax = T(0);
  const T* Wrow = Wl + kq * LDW + j;

  // ZMODE: element offset of accumulator row n inside a wave's LDS tile, or -1.
  // (lives in LDS -- offBn's storage is free once W is staged -- to keep VGPRs low)
  int* zoffT = reinterpret_cast<int*>(offBn);
  T* tile = nullptr;
  if constexpr (ZMODE) tile = Zl + (size_t)wave * (p.N * CH);
  // c_break: (piece index relative to the table, row inside the piece) of the chunk being stored; advanced incrementally
  uint32_t zt_ = 0, zr_ = 0;
  if constexpr (ZMODE) {
    if (p.c_break) {
      const uint32_t m0 = c_begin * CH;
      zt_ = m0 / p.l_in - zq_first;
      zr_ = m0 - (m0 / p.l_in) * p.l_in;
    }
  }

  for (uint32_t g0 = 0; g0 < g_total; g0 += RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const uint32_t g = g0 + u;
      if (g < g_total) {
        T w[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[nt] = Wrow[(4 * cp_s) * LDW + nt * 16];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int t = 0; t < V; ++t) acc[t][nt] = SMfma<T>::run(w[nt], areg[u][t], acc[t][nt]);
        if (g + RING < g_total) issue_load(areg[u]);
        if (++cp_s == KS) {
          cp_s = 0;
          if constexpr (ZMODE) {
            // ---- transpose through LDS, store contiguous runs ------------------
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int z = zoffT[nt * 16 + SMfma<T>::row(lane, r)];
                if (z >= 0) {
#pragma unroll
                  for (int t = 0; t < V; ++t)
                    tile[z + (V * j + t) * (int)p.d_in] = acc[t][nt][r] * alpha;
                }
              }
            const uint32_t run_v = (CH * p.d_in) / EV;          // 16-byte vectors per n_out run
            const uint32_t tot_v = (p.N * CH) / EV;
            if (!p.c_break) {
              __builtin_amdgcn_wave_barrier();
              for (uint32_t q = lane; q < tot_v; q += 64) {
                uint32_t no = q / run_v, wv = q - no * run_v;
                T o[EV];
                vload<T, EV>(o, tile + (size_t)q * EV);
#pragma unroll
                for (int e = 0; e < EV; ++e) {
                  T a = o[e] < T(0) ? -o[e] : o[e];
                  vmax = a > vmax ? a : vmax;
                }
                vstore<T, EV>(C + cbase + offCn[no * p.d_in] + (int64_t)wv * EV, o);
              }
            } else {
              // c_break: the innermost M group of C (length l_in, not a multiple of CH) ends INSIDE this chunk -- the last
              // site of rows 3 / 4 of a corner sweep leaves runs of 36 / 216 open-leg values.  A few pieces, each contiguous
              // in C; their bases come from the workgroup's table of group starts (no division, no decomposition here), a
              // 16-byte vector never crosses a break (host-checked divisibility)
              // pieces of this chunk: piece 0 = rows zr_ .. of group zt_, piece k >= 1 = group zt_ + k from its first row;
              // in ELEMENTS of a run (row x d_in) piece k starts at thr[k - 1]: no division anywhere
              const uint32_t le = p.l_in * p.d_in;                       // elements of a whole piece
              const uint32_t thr0 = (p.l_in - zr_) * p.d_in;            // end of piece 0 (may lie beyond the chunk)
              const int64_t c0 = gb[zt_] + (int64_t)zr_ * p.sc_m_in;
              __builtin_amdgcn_wave_barrier();
              for (uint32_t q = lane; q < tot_v; q += 64) {
                uint32_t no = q / run_v, wv = q - no * run_v;
                const uint32_t e0 = wv * EV;
                T o[EV];
                vload<T, EV>(o, tile + (size_t)q * EV);
#pragma unroll
                for (int e = 0; e < EV; ++e) {
                  T a = o[e] < T(0) ? -o[e] : o[e];
                  vmax = a > vmax ? a : vmax;
                }
                int64_t oc = c0 + (int64_t)e0;
                if (e0 >= thr0) {
                  uint32_t k = 1, rest = e0 - thr0;
                  while (rest >= le) { rest -= le; ++k; }               // (<= CH / l_in + 1 pieces: a handful)
                  oc = gb[zt_ + k] + (int64_t)rest;
                }
                vstore<T, EV>(C + oc + offCn[no * p.d_in], o);
              }
              zr_ += CSTRIDE * CH;                       // this wave's next chunk
              while (zr_ >= p.l_in) { zr_ -= p.l_in; ++zt_; }
            }
            __builtin_amdgcn_wave_barrier();
          } else {
            // ---- V-wide stores straight from the accumulators, lanes along m ------
            int64_t oc = -1;
            if (aligned) {
              oc = cbase + V * j;
            } else {
              uint32_t m = cp_chunk * CH + V * j;
              int64_t oa;
              if (m < p.M) sdecomp2(m, p.nm, p.dim_m, p.sa_m, p.sc_m, oa, oc);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int n = nt * 16 + SMfma<T>::row(lane, r);
                T o[V];
#pragma unroll
                for (int t = 0; t < V; ++t) {
                  o[t] = acc[t][nt][r] * alpha;
                  T a = o[t] < T(0) ? -o[t] : o[t];
                  vmax = a > vmax ? a : vmax;
                }
                const int64_t on = offCn[n];
                if (oc >= 0 && on >= 0) vstore<T, V>(C + oc + on, o);
              }
            }
          }
#pragma unroll
          for (int t = 0; t < V; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = acc_t{0, 0, 0, 0};
          cp_chunk += CSTRIDE;
          cp_in += CSTRIDE;
          if (aligned && !p.c_break && cp_chunk < c_end) {
            if (cp_in < inner_chunks) {
              cbase += (int64_t)(CSTRIDE * CH) * p.sc_m_in;
            } else {
              cp_in = cp_chunk % inner_chunks;
              sdecomp2(cp_chunk * CH, p.nm, p.dim_m, p.sa_m, p.sc_m, dummy, cbase);
            }
          }
        }
      }
    }
  }

  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      T o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0)
      atomicMax(reinterpret_cast<typename SMfma<T>::bits_t*>(absmax_out) + (wglob % QAMD_SLOTS),
                SMfma<T>::bits(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <typename T, int V, int NT, bool ZMODE>
static int launch_stream_vnz(const StreamArgs& a, const void* A, const void* B, void* C, const void* ktab,
                             const void* sa, const void* sb, void* amax, hipStream_t st) {
  constexpr int RING = 8;
  constexpr int NPAD = NT * 16;
  constexpr int LDW = NPAD + ((48 - NPAD % 32) % 32);
  size_t lds = (size_t)(2 * NPAD + a.Kpad) * 8 + (size_t)a.Kpad * LDW * sizeof(T);
  if (ZMODE) lds += (size_t)4 * a.N * 16 * V * sizeof(T) + (size_t)a.zb_groups * sizeof(int64_t);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)stream_kernel<T, V, NT, RING, ZMODE>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  QAMD_LAUNCH((stream_kernel<T, V, NT, RING, ZMODE>), dim3(a.grid), dim3(256), lds, st, a, (const T*)A,
              (const T*)B, (T*)C, (const int64_t*)ktab, (const T*)sa, (const T*)sb, (T*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T, int V, bool ZMODE>
static int launch_stream_vz(const StreamArgs& a, const void* A, const void* B, void* C, const void* ktab,
                            const void* sa, const void* sb, void* amax, hipStream_t st) {
  switch (a.NT) {
    case 1: return launch_stream_vnz<T, V, 1, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 2: return launch_stream_vnz<T, V, 2, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 3: return launch_stream_vnz<T, V, 3, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    case 4: return launch_stream_vnz<T, V, 4, ZMODE>(a, A, B, C, ktab, sa, sb, amax, st);
    default: return -1;
  }
}

extern "C" int qamd_stream_launch(int dtype, int V, const StreamArgs* a, const void* A, const void* B,
                                  void* C, const void* ktab, const void* scale_a, const void* scale_b,
                                  void* absmax_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    if (a->zmode) {
      if (V == 4) return launch_stream_vz<float, 4, true>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
      return -2;
    }
    if (V == 4) return launch_stream_vz<float, 4, false>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
    if (V == 2) return launch_stream_vz<float, 2, false>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
    if (V == 1) return launch_stream_vz<float, 1, false>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  } else if (dtype == 1) {
    if (a->zmode) {
      if (V == 2) return launch_stream_vz<double, 2, true>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
      return -2;
    }
    if (V == 2) return launch_stream_vz<double, 2, false>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
    if (V == 1) return launch_stream_vz<double, 1, false>(*a, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  }
  return -2;
}
