// gett.hip -- pairwise tensor contraction as a GEMM with tensor addressing
// (no operand is ever physically permuted), CDNA4 / gfx950 only.
//
//   C[b, m, n] = sum_k A[b, m, k] * B[b, k, n]
//
// Replaces the reference's per-step  transpose -> reshape -> matmul -> reshape
// lowering (cotengra executor reached from quimb/tensor/contraction.py:285 and
// do("tensordot") at quimb/tensor/tensor_core.py:3793).  Each bundle (b, m, n, k)
// is a list of fused index groups; a tile's row / column element offsets are
// produced by a mixed-radix decomposition into LDS tables, K offsets come from a
// small device table built once per plan.
//
// Structure: 256-thread workgroup (4 wave64), register-prefetched global loads,
// LDS-staged [BK][BX] operand tiles, v_mfma_{f32,f64}_16x16x4 accumulation.
// LDS row pitch is chosen so that the 4 k-rows read by one MFMA fragment fetch
// land on disjoint bank halves (pitch mod 32 == 16 dwords).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

// hipGetLastError() also reports benign stale codes (hipErrorNotReady from an
// event query by the allocator, ...): clear them before each launch so the
// post-launch check only sees this launch.
#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct Mfma;
template <> struct Mfma<float> {
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // D row held by accumulator register r of this lane
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <> struct Mfma<double> {
  typedef __attribute__((ext_vector_type(4))) double acc_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

template <typename T> struct Quad { T v[4]; };

// max over the producer's absmax slots (0 / missing => 1)
template <typename T>
__device__ __forceinline__ T gett_read_scale(const T* slots) {
  if (!slots) return T(1);
  T m = T(0);
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    T v = slots[i];
    m = v > m ? v : m;
  }
  return m > T(0) ? m : T(1);
}
// non-negative IEEE values order like unsigned integers
__device__ __forceinline__ void gett_atomic_max(float* p, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}
__device__ __forceinline__ void gett_atomic_max(double* p, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

__device__ __forceinline__ int64_t decomp_off(uint32_t idx, int n, const uint32_t* dims,
                                              const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d;
    uint32_t r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

__device__ __forceinline__ void decomp_off2(uint32_t idx, int n, const uint32_t* dims,
                                            const int64_t* s1, const int64_t* s2, int64_t& o1,
                                            int64_t& o2) {
  o1 = 0;
  o2 = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d;
    uint32_t r = idx - q * d;
    o1 += (int64_t)r * s1[g];
    o2 += (int64_t)r * s2[g];
    idx = q;
  }
}

// 16-byte / 8-byte packed loads with only the alignment we actually guarantee
template <typename T, int N> struct VecT {
  typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T) * N > 16 ? 16 : sizeof(T) * N)));
};

template <typename T>
__device__ __forceinline__ void load_quad_contig(Quad<T>& q, const T* p, int vec) {
  // 4 consecutive logical elements, memory-contiguous in runs of `vec`
  if (vec == 4) {
    if (sizeof(T) == 4) {
      typename VecT<T, 4>::type v = *reinterpret_cast<const typename VecT<T, 4>::type*>(p);
      q.v[0] = v[0]; q.v[1] = v[1]; q.v[2] = v[2]; q.v[3] = v[3];
    } else {
      typename VecT<T, 2>::type v0 = *reinterpret_cast<const typename VecT<T, 2>::type*>(p);
      typename VecT<T, 2>::type v1 = *reinterpret_cast<const typename VecT<T, 2>::type*>(p + 2);
      q.v[0] = v0[0]; q.v[1] = v0[1]; q.v[2] = v1[0]; q.v[3] = v1[1];
    }
  }
}

// Tile loader.  Tile is [BK][BX] in LDS (pitch LD).  XCONTIG: lanes run along x.
template <typename T, int BX, int BK, int LD, int NQ>
struct TileLoader {
  static constexpr int QUADS = BX * BK / 4;

  // global -> registers
  static __device__ __forceinline__ void load(Quad<T> (&r)[NQ], const T* __restrict__ base,
                                              const int64_t* __restrict__ xoff_lds,
                                              const int64_t* __restrict__ ktab, uint32_t k0,
                                              int kcontig, int vec, int tid) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      int e = tid + qi * 256;
      Quad<T>& q = r[qi];
      q.v[0] = q.v[1] = q.v[2] = q.v[3] = T(0);
      if (QUADS % 256 != 0 && e >= QUADS) continue;
      if (!kcontig) {
        int xq = e % (BX / 4);
        int k = e / (BX / 4);
        int x = xq * 4;
        int64_t ko = ktab[k0 + k];
        if (ko < 0) continue;
        if (vec == 4) {
          int64_t xo = xoff_lds[x];
          if (xo >= 0) load_quad_contig(q, base + xo + ko, 4);
        } else if (vec == 2) {
          int64_t xo0 = xoff_lds[x], xo1 = xoff_lds[x + 2];
          if (xo0 >= 0) {
            typename VecT<T, 2>::type v = *reinterpret_cast<const typename VecT<T, 2>::type*>(base + xo0 + ko);
            q.v[0] = v[0]; q.v[1] = v[1];
          }
          if (xo1 >= 0) {
            typename VecT<T, 2>::type v = *reinterpret_cast<const typename VecT<T, 2>::type*>(base + xo1 + ko);
            q.v[2] = v[0]; q.v[3] = v[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int64_t xo = xoff_lds[x + j];
            if (xo >= 0) q.v[j] = base[xo + ko];
          }
        }
      } else {
        int kq = e % (BK / 4);
        int x = e / (BK / 4);
        int k = kq * 4;
        int64_t xo = xoff_lds[x];
        if (xo < 0) continue;
        if (vec == 4) {
          int64_t ko = ktab[k0 + k];
          if (ko >= 0) load_quad_contig(q, base + xo + ko, 4);
        } else if (vec == 2) {
          int64_t ko0 = ktab[k0 + k], ko1 = ktab[k0 + k + 2];
          if (ko0 >= 0) {
            typename VecT<T, 2>::type v = *reinterpret_cast<const typename VecT<T, 2>::type*>(base + xo + ko0);
            q.v[0] = v[0]; q.v[1] = v[1];
          }
          if (ko1 >= 0) {
            typename VecT<T, 2>::type v = *reinterpret_cast<const typename VecT<T, 2>::type*>(base + xo + ko1);
            q.v[2] = v[0]; q.v[3] = v[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int64_t ko = ktab[k0 + k + j];
            if (ko >= 0) q.v[j] = base[xo + ko];
          }
        }
      }
    }
  }

  // registers -> LDS
  static __device__ __forceinline__ void store(const Quad<T> (&r)[NQ], T* __restrict__ lds,
                                               int kcontig, int tid) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      int e = tid + qi * 256;
      if (QUADS % 256 != 0 && e >= QUADS) continue;
      const Quad<T>& q = r[qi];
      if (!kcontig) {
        int xq = e % (BX / 4);
        int k = e / (BX / 4);
        T* p = lds + k * LD + xq * 4;
        if (sizeof(T) == 4) {
          typename VecT<T, 4>::type v;
          v[0] = q.v[0]; v[1] = q.v[1]; v[2] = q.v[2]; v[3] = q.v[3];
          *reinterpret_cast<typename VecT<T, 4>::type*>(p) = v;
        } else {
          typename VecT<T, 2>::type v0, v1;
          v0[0] = q.v[0]; v0[1] = q.v[1]; v1[0] = q.v[2]; v1[1] = q.v[3];
          *reinterpret_cast<typename VecT<T, 2>::type*>(p) = v0;
          *reinterpret_cast<typename VecT<T, 2>::type*>(p + 2) = v1;
        }
      } else {
        int kq = e % (BK / 4);
        int x = e / (BK / 4);
        T* p = lds + (kq * 4) * LD + x;
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j * LD] = q.v[j];
      }
    }
  }
};

constexpr int lds_pitch(int bx) { return bx + ((48 - bx % 32) % 32); }

template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int BK, bool SWAP>
__global__ __launch_bounds__(256) void gett_kernel(const GettArgs p, const T* __restrict__ A,
                                                    const T* __restrict__ B, T* __restrict__ C,
                                                    const int64_t* __restrict__ ktab,
                                                    const T* __restrict__ scale_a,
                                                    const T* __restrict__ scale_b,
                                                    T* __restrict__ absmax_out) {
  static_assert(WAVES_M * WAVES_N == 4, "256-thread workgroup");
  constexpr int BM = WAVES_M * WM * 16;
  constexpr int BN = WAVES_N * WN * 16;
  constexpr int LDA = lds_pitch(BM);
  constexpr int LDB = lds_pitch(BN);
  constexpr int NQA = (BM * BK / 4 + 255) / 256;
  constexpr int NQB = (BN * BK / 4 + 255) / 256;
  typedef typename Mfma<T>::acc_t acc_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* offAm = reinterpret_cast<int64_t*>(smem);
  int64_t* offCm = offAm + BM;
  int64_t* offBn = offCm + BM;
  int64_t* offCn = offBn + BN;
  // double-buffered operand tiles: [2][BK][LDA] and [2][BK][LDB]
  T* As = reinterpret_cast<T*>(offCn + BN);
  T* Bs = As + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  // ---- tile coordinates ------------------------------------------------
  uint32_t id = blockIdx.x;
  const uint32_t tn = id % p.tiles_n; id /= p.tiles_n;
  const uint32_t tm = id % p.tiles_m; id /= p.tiles_m;
  const uint32_t ks = id % p.split_k;
  const uint32_t bb = id / p.split_k;

  int64_t boffA = 0, boffB = 0, boffC = 0;
  {
    uint32_t idx = bb;
    for (int g = p.nb - 1; g >= 0; --g) {
      uint32_t d = p.dim_b[g];
      uint32_t q = idx / d, r = idx - q * d;
      boffA += (int64_t)r * p.sa_b[g];
      boffB += (int64_t)r * p.sb_b[g];
      boffC += (int64_t)r * p.sc_b[g];
      idx = q;
    }
  }
  boffC += (int64_t)ks * p.slab_stride;

  // ---- row / column offset tables ---------------------------------------
  for (int i = tid; i < BM + BN; i += 256) {
    if (i < BM) {
      uint32_t m = tm * BM + i;
      int64_t oa = -1, oc = -1;
      if (m < p.M) decomp_off2(m, p.nm, p.dim_m, p.sa_m, p.sc_m, oa, oc);
      offAm[i] = oa;
      offCm[i] = oc;
    } else {
      int j = i - BM;
      uint32_t n = tn * BN + j;
      int64_t ob = -1, oc = -1;
      if (n < p.N) decomp_off2(n, p.nn, p.dim_n, p.sb_n, p.sc_n, ob, oc);
      offBn[j] = ob;
      offCn[j] = oc;
    }
  }
  __syncthreads();

  const T* Ab = A + boffA;
  const T* Bb = B + boffB;
  const int64_t* ktA = ktab;
  const int64_t* ktB = ktab + p.Kpad;

  const uint32_t kbeg = ks * p.Kc;
  uint32_t kend = kbeg + p.Kc;
  if (kend > p.Kloop) kend = p.Kloop;
  const int nkt = (int)((kend - kbeg) / BK);

  acc_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  Quad<T> ra[NQA], rb[NQB];
  typedef TileLoader<T, BM, BK, LDA, NQA> LA;
  typedef TileLoader<T, BN, BK, LDB, NQB> LB;

  if (nkt > 0) {
    LA::load(ra, Ab, offAm, ktA, kbeg, p.a_kcontig, p.vec_a, tid);
    LB::load(rb, Bb, offBn, ktB, kbeg, p.b_kcontig, p.vec_b, tid);
    LA::store(ra, As, p.a_kcontig, tid);
    LB::store(rb, Bs, p.b_kcontig, tid);
  }
  __syncthreads();

  const int fr = lane & 15;
  const int fk = lane >> 4;
  const T* Arow = As + fk * LDA + wm * (WM * 16) + fr;
  const T* Brow = Bs + fk * LDB + wn * (WN * 16) + fr;

  // one barrier per k-step: while tile kt is consumed from buffer kt&1, tile kt+1 is
  // fetched into registers and written to the other buffer; the barrier at the end of
  // the step both publishes it and retires every read of the buffer that the step
  // after next will overwrite.
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    const int cur = kt & 1;
    if (more) {
      uint32_t k0 = kbeg + (kt + 1) * BK;
      LA::load(ra, Ab, offAm, ktA, k0, p.a_kcontig, p.vec_a, tid);
      LB::load(rb, Bb, offBn, ktB, k0, p.b_kcontig, p.vec_b, tid);
    }
    const T* Ac = Arow + cur * (BK * LDA);
    const T* Bc = Brow + cur * (BK * LDB);
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      T af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = Ac[k4 * 4 * LDA + i * 16];
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[j] = Bc[k4 * 4 * LDB + j * 16];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (SWAP) acc[i][j] = Mfma<T>::run(bf[j], af[i], acc[i][j]);
          else acc[i][j] = Mfma<T>::run(af[i], bf[j], acc[i][j]);
        }
    }
    if (more) {
      LA::store(ra, As + (cur ^ 1) * (BK * LDA), p.a_kcontig, tid);
      LB::store(rb, Bs + (cur ^ 1) * (BK * LDB), p.b_kcontig, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: direct stores, lanes along C's contiguous bundle -------
  T* Cb = C + boffC;
  // fused exponent stripping: scale by 1/(max|A| max|B|) read from the producers'
  // slots, reduce max|C| into this tensor's slots (split-K defers both to the reduce)
  const T alpha = T(1) / (gett_read_scale(scale_a) * gett_read_scale(scale_b));
  T vmax = T(0);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ml, nl;
        if (SWAP) {
          ml = wm * (WM * 16) + i * 16 + fr;
          nl = wn * (WN * 16) + j * 16 + Mfma<T>::row(lane, r);
        } else {
          ml = wm * (WM * 16) + i * 16 + Mfma<T>::row(lane, r);
          nl = wn * (WN * 16) + j * 16 + fr;
        }
        int64_t om = offCm[ml], on = offCn[nl];
        T v = acc[i][j][r] * alpha;
        if (om >= 0 && on >= 0) {
          Cb[om + on] = v;
          T av = v < T(0) ? -v : v;
          vmax = av > vmax ? av : vmax;
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
    if (lane == 0) gett_atomic_max(absmax_out + ((blockIdx.x * 4 + wave) % QAMD_SLOTS), vmax);
  }
}

// ---- split-K slab reduction ------------------------------------------------
template <typename T>
__global__ void splitk_reduce_kernel(T* __restrict__ C, const T* __restrict__ ws, int64_t n,
                                     int split_k, const T* __restrict__ scale_a,
                                     const T* __restrict__ scale_b, T* __restrict__ absmax_out) {
  const T alpha = T(1) / (gett_read_scale(scale_a) * gett_read_scale(scale_b));
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  T vmax = T(0);
  for (; i < n; i += stride) {
    T s = ws[i];
    for (int k = 1; k < split_k; ++k) s += ws[(int64_t)k * n + i];
    s *= alpha;
    C[i] = s;
    T av = s < T(0) ? -s : s;
    vmax = av > vmax ? av : vmax;
  }
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      T o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if ((threadIdx.x & 63) == 0)
      gett_atomic_max(absmax_out + (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) % QAMD_SLOTS), vmax);
  }
}

// ---- K offset table --------------------------------------------------------
__global__ void build_ktab_kernel(int64_t* __restrict__ ktab, KtabArgs a) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.Kpad) return;
  int64_t oa = -1, ob = -1;
  if (k < a.K) decomp_off2(k, a.nk, a.dim_k, a.sa_k, a.sb_k, oa, ob);
  ktab[k] = oa;
  ktab[a.Kpad + k] = ob;
}

}  // namespace qamd

// ---------------------------------------------------------------------------
// host-side launch table
// ---------------------------------------------------------------------------
using namespace qamd;

template <typename T, int WAVES_M, int WAVES_N, int WM, int WN, int BK>
static int launch_cfg(const GettArgs& a, bool swap, const void* A, const void* B, void* C,
                      const void* ktab, const void* sa, const void* sb, void* amax, hipStream_t st) {
  constexpr int BM = WAVES_M * WM * 16;
  constexpr int BN = WAVES_N * WN * 16;
  size_t lds = (size_t)(2 * BM + 2 * BN) * 8 + (size_t)2 * BK * (lds_pitch(BM) + lds_pitch(BN)) * sizeof(T);
  if (lds > 64 * 1024) {
    if (swap) (void)hipFuncSetAttribute((const void*)gett_kernel<T, WAVES_M, WAVES_N, WM, WN, BK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)gett_kernel<T, WAVES_M, WAVES_N, WM, WN, BK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  uint64_t grid = (uint64_t)a.tiles_m * a.tiles_n * a.split_k * a.B;
  if (grid == 0 || grid > 0x7fffffffull) return -1;
  if (swap)
    QAMD_LAUNCH((gett_kernel<T, WAVES_M, WAVES_N, WM, WN, BK, true>), dim3((uint32_t)grid),
                       dim3(256), lds, st, a, (const T*)A, (const T*)B, (T*)C, (const int64_t*)ktab,
                       (const T*)sa, (const T*)sb, (T*)amax);
  else
    QAMD_LAUNCH((gett_kernel<T, WAVES_M, WAVES_N, WM, WN, BK, false>), dim3((uint32_t)grid),
                       dim3(256), lds, st, a, (const T*)A, (const T*)B, (T*)C, (const int64_t*)ktab,
                       (const T*)sa, (const T*)sb, (T*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T>
static int launch_T(int cfg, const GettArgs& a, bool swap, const void* A, const void* B, void* C,
                    const void* ktab, const void* sa, const void* sb, void* amax, hipStream_t st) {
  switch (cfg) {
    case 0: return launch_cfg<T, 2, 2, 4, 4, 16>(a, swap, A, B, C, ktab, sa, sb, amax, st);  // 128 x 128
    case 1: return launch_cfg<T, 2, 2, 2, 2, 16>(a, swap, A, B, C, ktab, sa, sb, amax, st);  //  64 x  64
    case 2: return launch_cfg<T, 4, 1, 4, 3, 16>(a, swap, A, B, C, ktab, sa, sb, amax, st);  // 256 x  48
    case 3: return launch_cfg<T, 4, 1, 4, 1, 16>(a, swap, A, B, C, ktab, sa, sb, amax, st);  // 256 x  16
    case 4: return launch_cfg<T, 4, 1, 2, 2, 16>(a, swap, A, B, C, ktab, sa, sb, amax, st);  // 128 x  32
    case 5: return launch_cfg<T, 2, 2, 4, 4, 32>(a, swap, A, B, C, ktab, sa, sb, amax, st);  // 128 x 128, k-tile 32
    default: return -1;
  }
}

extern "C" int qamd_gett_launch(int dtype, int cfg, const GettArgs* a, int swap, const void* A,
                                const void* B, void* C, const void* ktab, const void* sa, const void* sb,
                                void* amax, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) return launch_T<float>(cfg, *a, swap != 0, A, B, C, ktab, sa, sb, amax, st);
  if (dtype == 1) return launch_T<double>(cfg, *a, swap != 0, A, B, C, ktab, sa, sb, amax, st);
  return -2;
}

extern "C" void qamd_gett_tile_dims(int cfg, int* bm, int* bn, int* bk) {
  static const int t[6][3] = {{128, 128, 16}, {64, 64, 16}, {256, 48, 16}, {256, 16, 16}, {128, 32, 16}, {128, 128, 32}};
  if (cfg < 0 || cfg > 5) { *bm = *bn = *bk = 0; return; }
  *bm = t[cfg][0]; *bn = t[cfg][1]; *bk = t[cfg][2];
}

extern "C" int qamd_splitk_reduce_launch(int dtype, void* C, const void* ws, int64_t n, int split_k,
                                         const void* sa, const void* sb, void* amax, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == 0)
    QAMD_LAUNCH(splitk_reduce_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, st, (float*)C,
                       (const float*)ws, n, split_k, (const float*)sa, (const float*)sb, (float*)amax);
  else if (dtype == 1)
    QAMD_LAUNCH(splitk_reduce_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, st,
                       (double*)C, (const double*)ws, n, split_k, (const double*)sa, (const double*)sb, (double*)amax);
  else
    return -2;
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int qamd_build_ktab_launch(void* ktab, const KtabArgs* a, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  uint32_t blocks = (a->Kpad + 255) / 256;
  QAMD_LAUNCH(build_ktab_kernel, dim3(blocks), dim3(256), 0, st, (int64_t*)ktab, *a);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
