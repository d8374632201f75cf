// gettf.hip -- the full-tile fast path of the tiled GETT (gett.hip), gfx950 only.
//
//   C[b, m, n] = alpha * sum_k A[b, m, k] * B[b, k, n]        (tensor addressing as in gett.hip)
//
// gett_kernel decides everything at run time inside its k-loop (vector width, which bundle
// is contiguous, edge predicates, 64-bit offsets read back from LDS tables).  For the
// MFMA-bound contractions -- both free bundles and K large -- that control flow, not the
// matrix pipe, sets the pace (80 of 157 TFLOP/s at 4096^3).  This kernel is the same
// algorithm with the decisions made by the host (qamd_pair_plan_finalize picks it when
// every tile is full and both operands take 4-element vector loads):
//
//  * 128 x 128 x 16 tile, 256 threads = 2 x 2 waves of 64 x 64, 16 accumulators per wave on
//    v_mfma_{f32,f64}_16x16x4; AKC / BKC (which bundle of A / B is contiguous) are template
//    parameters, so the loader is straight-line: per k-tile every thread issues 2 + 2
//    16-byte (f32) or 32-byte (f64) loads whose row offsets were hoisted out of the loop.
//  * double-buffered LDS tiles [2][16][144] per operand (pitch = 16 mod 32 dwords: the 4 k-rows
//    of an MFMA fragment land on disjoint bank quarters), one barrier per k-tile, the next
//    tile's global loads in flight behind the current tile's 64 MFMAs per wave.
//  * workgroup -> tile map in bands of 8 row tiles: consecutive workgroup ids (round-robin
//    over the 8 XCDs) get different row tiles of the band, so each XCD keeps ONE A panel in
//    its L2 while the B panels stream past.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct FMfma;
template <> struct FMfma<float> {
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
  static __device__ __forceinline__ void amax(float* p, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
  }
};
template <> struct FMfma<double> {
  typedef __attribute__((ext_vector_type(4))) double acc_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  static __device__ __forceinline__ void amax(double* p, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
  }
};

template <typename T> struct FQuad { T v[4]; };

template <typename T>
__device__ __forceinline__ void fload4(FQuad<T>& q, const T* p) {
  if constexpr (sizeof(T) == 4) {
    typedef T v4 __attribute__((ext_vector_type(4), aligned(16)));
    v4 v = *reinterpret_cast<const v4*>(p);
    q.v[0] = v[0]; q.v[1] = v[1]; q.v[2] = v[2]; q.v[3] = v[3];
  } else {
    typedef T v2 __attribute__((ext_vector_type(2), aligned(16)));
    v2 a = *reinterpret_cast<const v2*>(p), b = *reinterpret_cast<const v2*>(p + 2);
    q.v[0] = a[0]; q.v[1] = a[1]; q.v[2] = b[0]; q.v[3] = b[1];
  }
}
template <typename T>
__device__ __forceinline__ void fstore4(T* p, const FQuad<T>& q) {
  if constexpr (sizeof(T) == 4) {
    typedef T v4 __attribute__((ext_vector_type(4), aligned(16)));
    v4 v; v[0] = q.v[0]; v[1] = q.v[1]; v[2] = q.v[2]; v[3] = q.v[3];
    *reinterpret_cast<v4*>(p) = v;
  } else {
    typedef T v2 __attribute__((ext_vector_type(2), aligned(16)));
    v2 a, b; a[0] = q.v[0]; a[1] = q.v[1]; b[0] = q.v[2]; b[1] = q.v[3];
    *reinterpret_cast<v2*>(p) = a;
    *reinterpret_cast<v2*>(p + 2) = b;
  }
}

__device__ __forceinline__ int64_t fdecomp(uint32_t idx, int n, const uint32_t* dims, const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

template <typename T>
__device__ __forceinline__ T fread_scale(const T* slots) {
  if (!slots) return T(1);
  T m = T(0);
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    T v = slots[i];
    m = v > m ? v : m;
  }
  return m > T(0) ? m : T(1);
}

template <typename T, int V>
__device__ __forceinline__ void floadv(T* dst, const T* p) {
  if constexpr (V * sizeof(T) == 32) {          // 4 doubles: two 16-byte loads
    typedef T v2 __attribute__((ext_vector_type(2), aligned(16)));
    v2 a = *reinterpret_cast<const v2*>(p), b = *reinterpret_cast<const v2*>(p + 2);
    dst[0] = a[0]; dst[1] = a[1]; dst[2] = b[0]; dst[3] = b[1];
  } else {
    typedef T vv __attribute__((ext_vector_type(V), aligned(V * sizeof(T))));
    vv a = *reinterpret_cast<const vv*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) dst[i] = a[i];
  }
}

// One operand's global -> register -> LDS path, in pieces of V (4 or 2) contiguous elements.
// KC = false: lanes run along the free bundle x (the LDS image is written with vector stores);
// KC = true ("k-contiguous"): lanes run along k, 4 lanes cover one row's 16 k values, the LDS
// image is written transposed.
template <typename T, bool KC, int V, int BX, int LD>
struct FLoader {
  static constexpr int NQ = BX * 16 / 1024;   // 4-element groups per thread and k-tile
  static constexpr int XQ = BX / 4;           // groups per k row
  static constexpr int NP = 4 / V;            // pieces per group
  const T* base[KC ? NQ : NP];   // k-contiguous: rows x, x + 64; else the NP pieces of this thread's x group
  int lds_off[NQ];
  int ktoff[NQ];                 // index into the k-offset table relative to the tile's k0

  __device__ __forceinline__ void init(const T* op, uint32_t tile_x0, int nx, const uint32_t* dim_x,
                                       const int64_t* stride_x, int tid) {
    if constexpr (KC) {
      const int kq = tid & 3, x = tid >> 2;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        base[qi] = op + fdecomp(tile_x0 + x + 64 * qi, nx, dim_x, stride_x);
        lds_off[qi] = (4 * kq) * LD + x + 64 * qi;
        ktoff[qi] = 4 * kq;
      }
    } else {
      const int x = 4 * (tid % XQ), k = tid / XQ;
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) base[pc] = op + fdecomp(tile_x0 + x + V * pc, nx, dim_x, stride_x);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        lds_off[qi] = (k + (256 / XQ) * qi) * LD + x;
        ktoff[qi] = k + (256 / XQ) * qi;
      }
    }
  }
  __device__ __forceinline__ void load(FQuad<T> (&r)[NQ], const int64_t* __restrict__ kt, uint32_t k0) const {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) {
        if constexpr (KC) floadv<T, V>(r[qi].v + V * pc, base[qi] + kt[k0 + ktoff[qi] + V * pc]);
        else floadv<T, V>(r[qi].v + V * pc, base[pc] + kt[k0 + ktoff[qi]]);
      }
  }
  __device__ __forceinline__ void store(const FQuad<T> (&r)[NQ], T* __restrict__ tile) const {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      if constexpr (KC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[lds_off[qi] + j * LD] = r[qi].v[j];
      } else {
        fstore4(tile + lds_off[qi], r[qi]);
      }
    }
  }
};

#ifndef QAMD_GF_MINB_F32
#define QAMD_GF_MINB_F32 2
#endif
#ifndef QAMD_GF_MINB_F64
#define QAMD_GF_MINB_F64 2   // 2 workgroups per CU measured 63 vs 44 TFLOP/s at 4096^3 (register cap 256)
#endif
template <typename T, int WN, int VA, int VB, bool AKC, bool BKC, bool SWAP>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? QAMD_GF_MINB_F32 : QAMD_GF_MINB_F64)) void gettf_kernel(const GettArgs p, const T* __restrict__ A,
                                                     const T* __restrict__ B, T* __restrict__ C,
                                                     const int64_t* __restrict__ ktab,
                                                     const T* __restrict__ scale_a, const T* __restrict__ scale_b,
                                                     T* __restrict__ absmax_out) {
  constexpr int BM = 128, BN = 32 * WN, BK = 16, LD = BM + 16, LDB = BN + 16, WM = 4;
  typedef typename FMfma<T>::acc_t acc_t;

  extern __shared__ __attribute__((aligned(16))) char fsmem[];
  int64_t* offCm = reinterpret_cast<int64_t*>(fsmem);
  int64_t* offCn = offCm + BM;
  T* As = reinterpret_cast<T*>(offCn + BN);   // [2][BK][LD]
  T* Bs = As + 2 * BK * LD;                    // [2][BK][LDB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile coordinates: bands of 8 row tiles ---------------------------------------------
  const uint32_t per_batch = p.tiles_m * p.tiles_n;
  const uint32_t slab = blockIdx.x / per_batch;          // (batch, k-split)
  const uint32_t pid = blockIdx.x - slab * per_batch;
  const uint32_t bb = slab / p.split_k, ks = slab - bb * p.split_k;
  const uint32_t band = 8 * p.tiles_n;
  const uint32_t first_m = (pid / band) * 8;
  const uint32_t gsz = (p.tiles_m - first_m) < 8 ? (p.tiles_m - first_m) : 8;
  const uint32_t in_band = pid % band;
  const uint32_t tm = first_m + in_band % gsz;
  const uint32_t tn = in_band / gsz;

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
  boffC += (int64_t)ks * p.slab_stride;   // split-K: partial sums go to slab ks of the workspace

  // C offsets of this tile's rows / columns (epilogue only)
  if (tid < BM) offCm[tid] = fdecomp(tm * BM + tid, p.nm, p.dim_m, p.sc_m);
  else if (tid - BM < BN) offCn[tid - BM] = fdecomp(tn * BN + (tid - BM), p.nn, p.dim_n, p.sc_n);

  FLoader<T, AKC, VA, BM, LD> la;
  FLoader<T, BKC, VB, BN, LDB> lb;
  la.init(A + boffA, tm * BM, p.nm, p.dim_m, p.sa_m, tid);
  lb.init(B + boffB, tn * BN, p.nn, p.dim_n, p.sb_n, tid);
  const int64_t* ktA = ktab;
  const int64_t* ktB = ktab + p.Kpad;
  const uint32_t kbeg = ks * p.Kc;
  const uint32_t kend = (kbeg + p.Kc < p.K) ? kbeg + p.Kc : p.K;
  const int nkt = (int)((kend - kbeg) / BK);

  acc_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  FQuad<T> ra[FLoader<T, AKC, VA, BM, LD>::NQ], rb[FLoader<T, BKC, VB, BN, LDB>::NQ];
  la.load(ra, ktA, kbeg);
  lb.load(rb, ktB, kbeg);
  la.store(ra, As);
  lb.store(rb, Bs);
  __syncthreads();

  const int fr = lane & 15, fk = lane >> 4;
  const T* Arow = As + fk * LD + wm * 64 + fr;
  const T* Brow = Bs + fk * LDB + wn * (WN * 16) + fr;

  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    const int cur = kt & 1;
    if (more) {
      la.load(ra, ktA, kbeg + (uint32_t)(kt + 1) * BK);
      lb.load(rb, ktB, kbeg + (uint32_t)(kt + 1) * BK);
    }
    const T* Ac = Arow + cur * (BK * LD);
    const T* Bc = Brow + cur * (BK * LDB);
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      T af[WM], bf[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) af[i] = Ac[k4 * 4 * LD + i * 16];
#pragma unroll
      for (int j = 0; j < WN; ++j) bf[j] = Bc[k4 * 4 * LDB + j * 16];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          if (SWAP) acc[i][j] = FMfma<T>::run(bf[j], af[i], acc[i][j]);
          else acc[i][j] = FMfma<T>::run(af[i], bf[j], acc[i][j]);
        }
    }
    if (more) {
      la.store(ra, As + (cur ^ 1) * (BK * LD));
      lb.store(rb, Bs + (cur ^ 1) * (BK * LDB));
    }
    __syncthreads();
  }

  // ---- epilogue: lanes (fr) run along C's contiguous bundle -----------------------------------
  T* Cb = C + boffC;
  const T alpha = T(1) / (fread_scale(scale_a) * fread_scale(scale_b));
  T vmax = T(0);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ml, nl;
        if (SWAP) {
          ml = wm * 64 + i * 16 + fr;
          nl = wn * (WN * 16) + j * 16 + FMfma<T>::row(lane, r);
        } else {
          ml = wm * 64 + i * 16 + FMfma<T>::row(lane, r);
          nl = wn * (WN * 16) + j * 16 + fr;
        }
        T v = acc[i][j][r] * alpha;
        Cb[offCm[ml] + offCn[nl]] = v;
        T av = v < T(0) ? -v : v;
        vmax = av > vmax ? av : vmax;
      }
    }
  }
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      T o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0) FMfma<T>::amax(absmax_out + ((blockIdx.x * 4 + wave) % QAMD_SLOTS), vmax);
  }
}

}  // namespace qamd

using namespace qamd;

template <typename T, int WN, int VA, int VB, bool AKC, bool BKC>
static int launch_gettf_ab(const GettArgs& a, int swap, const void* A, const void* B, void* C, const void* ktab,
                           const void* sa, const void* sb, void* amax, hipStream_t st) {
  const size_t lds = (128 + 32 * WN) * sizeof(int64_t) + (size_t)2 * 16 * (144 + 32 * WN + 16) * sizeof(T);
  const unsigned grid = a.tiles_m * a.tiles_n * a.B * a.split_k;
  if (swap) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)gettf_kernel<T, WN, VA, VB, AKC, BKC, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    QAMD_LAUNCH((gettf_kernel<T, WN, VA, VB, AKC, BKC, true>), dim3(grid), dim3(256), lds, st, a, (const T*)A,
                (const T*)B, (T*)C, (const int64_t*)ktab, (const T*)sa, (const T*)sb, (T*)amax);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)gettf_kernel<T, WN, VA, VB, AKC, BKC, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    QAMD_LAUNCH((gettf_kernel<T, WN, VA, VB, AKC, BKC, false>), dim3(grid), dim3(256), lds, st, a, (const T*)A,
                (const T*)B, (T*)C, (const int64_t*)ktab, (const T*)sa, (const T*)sb, (T*)amax);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <typename T, int WN, int VA, int VB>
static int launch_gettf_v(const GettArgs& a, int swap, const void* A, const void* B, void* C, const void* ktab,
                          const void* sa, const void* sb, void* amax, hipStream_t st) {
  if (a.a_kcontig) {
    if (a.b_kcontig) return launch_gettf_ab<T, WN, VA, VB, true, true>(a, swap, A, B, C, ktab, sa, sb, amax, st);
    return launch_gettf_ab<T, WN, VA, VB, true, false>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  }
  if (a.b_kcontig) return launch_gettf_ab<T, WN, VA, VB, false, true>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  return launch_gettf_ab<T, WN, VA, VB, false, false>(a, swap, A, B, C, ktab, sa, sb, amax, st);
}

template <typename T, int WN>
static int launch_gettf_t(const GettArgs& a, int swap, const void* A, const void* B, void* C, const void* ktab,
                          const void* sa, const void* sb, void* amax, hipStream_t st) {
  const bool a4 = a.vec_a >= 4, b4 = a.vec_b >= 4;
  if (a4 && b4) return launch_gettf_v<T, WN, 4, 4>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  if (a4) return launch_gettf_v<T, WN, 4, 2>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  if (b4) return launch_gettf_v<T, WN, 2, 4>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  return launch_gettf_v<T, WN, 2, 2>(a, swap, A, B, C, ktab, sa, sb, amax, st);
}

// Preconditions (checked by the host planner): M % 128 == 0, N % bn == 0 (bn = 128 or 64), K % 16 == 0,
// vec_a, vec_b in {2, 4} with operands aligned to min(16, vec * itemsize) bytes.
extern "C" int qamd_gettf_launch(int dtype, int bn, const GettArgs* a, int swap, const void* A, const void* B, void* C,
                                 const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((bn != 128 && bn != 64) || a->vec_a < 2 || a->vec_b < 2) return -2;
  if (dtype == 0)
    return bn == 128 ? launch_gettf_t<float, 4>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st)
                     : launch_gettf_t<float, 2>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  if (dtype == 1)
    return bn == 128 ? launch_gettf_t<double, 4>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st)
                     : launch_gettf_t<double, 2>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  return -2;
}
