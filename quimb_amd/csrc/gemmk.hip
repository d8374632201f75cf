// gemmk.hip -- MFMA-bound GETT for "k-outer" operands, fp32, gfx950 only.
//
//   C[b, m, n] = alpha * sum_k A[b, k, m] * B[b, k, n]
//
// Both operands carry their free bundle innermost (stride-1, 16-byte vectors) and the contraction
// bundle is ONE fused group further out -- the shape death-ordered executor layouts give every
// GEMM-like join of a contraction tree (two half-network tensors meeting over their shared bonds:
// 6^5 x 6^5 x 6^5 on the 10x10 D=6 lattice).  Extents need not be multiples of anything but 4 (M, N)
// and 8 (K): powers of 6 run on the same straight-line loop as powers of 2.
//
//  * workgroup = 2 x 2 waves, wave tile (32 TA) x (32 TB) on v_mfma_f32_32x32x2_f32 (64 cycles per
//    instruction and SIMD: one wave per SIMD issues the matrix pipe back to back and has ~12 issue
//    slots to spare per MFMA), TA, TB in {2, 3, 4}: workgroup tiles 128 ... 256 on a side, picked by
//    the host for the least quantisation loss on 256 CUs.  The 12- and 9-sub-tile wave tiles run a
//    TWO-stage ring so that two workgroups share a CU (QAMD_GEMMK_CASES below): whatever one wave
//    loses at its barrier, prologue or epilogue, the SIMD's other wave fills.
//  * operands go HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4), never through registers:
//    the stage image [16 k][BM] / [16 k][BN] is lane-linear because the free bundle is contiguous, and
//    a 32x32x2 fragment read (32 consecutive m of one k row per half wave) is conflict-free on it
//    with no padding.  NS-stage ring, ONE barrier per k-tile.  NS = 3: loads two k-tiles ahead, the barrier
//    three k-steps before the tile ends (the tile being waited for was requested ~16 K cycles earlier);
//    NS = 2: one k-tile ahead, the barrier at the tile's last k-step (the request that follows it
//    overwrites the stage this tile was read from).
//  * edge tiles: lanes past M / N re-read the last valid vector (their rows / columns of the tile are
//    never stored, and a product row only ever pollutes itself); K % 16 == 8 runs a half tile last.
//  * TA / TB == 2 or 4: "permuted" fragments -- lane i of sub-tile t holds row TA*i + t, so ONE
//    ds_read_b64 / b128 feeds all sub-tiles and the epilogue stores 8 / 16 contiguous bytes per lane.
//  * workgroup -> tile: each XCD owns a contiguous run of the tile sequence, the sequence walks the
//    tile grid in bands of 4 tile rows, so the 32 workgroups resident on an XCD share 4 A panels and
//    8 B panels in its L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamdk {

typedef __attribute__((ext_vector_type(16))) float acc16;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

__device__ __forceinline__ int64_t kdecomp(uint32_t idx, int n, const uint32_t* dims, const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

__device__ __forceinline__ float kread_scale(const float* slots) {
  if (!slots) return 1.0f;
  float m = 0.0f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.0f ? m : 1.0f;
}

// 16 bytes per lane HBM/L2 -> LDS (LDS-DMA): lane l lands at dst + 16 l bytes, dst wave-uniform.
// (A plain function on purpose: handed a TYPE-DEPENDENT argument inside a template, hipcc 7.2's host pass drops
// the whole kernel's stub without a diagnostic.)
__device__ __forceinline__ void kglds(const char* src, float* dst) {
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// fragment of T sub-tiles for one k row: permuted (one vector read) when T is 2 or 4
template <int T>
__device__ __forceinline__ void kfrag(float (&f)[T], const float* row, int lane31) {
  if constexpr (T == 4) {
    f4 v = *reinterpret_cast<const f4*>(row + 4 * lane31);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
  } else if constexpr (T == 2) {
    f2 v = *reinterpret_cast<const f2*>(row + 2 * lane31);
    f[0] = v[0]; f[1] = v[1];
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) f[t] = row[32 * t + lane31];
  }
}
// local row / column of (sub-tile t, in-tile index i) under the same mapping
template <int T>
__device__ __forceinline__ int kmap(int t, int i) {
  if constexpr (T == 4 || T == 2) return T * i + t;
  else return 32 * t + i;
}

// DOT: the result tile is not stored -- it is multiplied element by element with the tensor `C` points at (same layout as
// the result would have had) and summed: `absmax_out` then receives ONE double per workgroup (the partial inner product,
// unscaled), finished by gemmk_dot_finish_kernel.
template <int TA, int TB, int NS, int MINW, bool DOT = false>
__global__ __launch_bounds__(256, MINW) void gemmk_kernel(const GettArgs p, const float* __restrict__ A,
                                                          const float* __restrict__ B, float* __restrict__ C,
                                                          const float* __restrict__ scale_a,
                                                          const float* __restrict__ scale_b,
                                                          float* __restrict__ absmax_out) {
  constexpr int BM = 64 * TA, BN = 64 * TB, BK = 16;
  constexpr int STAGE = BK * (BM + BN);   // floats per stage: A image [BK][BM], then B image [BK][BN]
  extern __shared__ __attribute__((aligned(16))) char ksmem[];
  float* stages = reinterpret_cast<float*>(ksmem);
  int64_t* offCm = reinterpret_cast<int64_t*>(stages + NS * STAGE);
  int64_t* offCn = offCm + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile coordinates ---------------------------------------------------------------------
  const uint32_t per_batch = p.tiles_m * p.tiles_n;
  const uint32_t bb = blockIdx.x / per_batch;
  const uint32_t pid = blockIdx.x - bb * per_batch;
  uint32_t tm, tn;
  {
    const uint32_t xcd = pid & 7, idx = pid >> 3;
    const uint32_t q = per_batch >> 3, r = per_batch & 7;
    const uint32_t s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const uint32_t band = 4 * p.tiles_n;
    const uint32_t first_m = (s / band) * 4;
    const uint32_t gsz = (p.tiles_m - first_m) < 4 ? (p.tiles_m - first_m) : 4;
    const uint32_t in_band = s % band;
    tm = first_m + in_band % gsz;
    tn = in_band / gsz;
  }
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
  const uint32_t m0 = tm * BM, n0 = tn * BN;
  // C offsets of the tile's rows / columns (epilogue); rows past the edge get a valid dummy
  for (int i = tid; i < BM + BN; i += 256) {
    if (i < BM) {
      uint32_t g = m0 + i;
      offCm[i] = kdecomp(g < p.M ? g : p.M - 1, p.nm, p.dim_m, p.sc_m);
    } else {
      uint32_t g = n0 + (i - BM);
      offCn[i - BM] = kdecomp(g < p.N ? g : p.N - 1, p.nn, p.dim_n, p.sc_n);
    }
  }

  // ---- this lane's LDS-DMA pieces: piece = 1 KiB of a stage image, lane l its l-th 16 bytes ------
  // source = wave-uniform base (advances by 16 k rows per tile, SGPRs) + a 32-bit per-lane byte offset
  uint32_t offA[TA], offB[TB];
#pragma unroll
  for (int q = 0; q < TA; ++q) {
    const int f = 256 * (wave + 4 * q) + 4 * lane;
    const int k = f / BM, m = f - k * BM;
    uint32_t g = m0 + m;
    g = g + 4 <= p.M ? g : p.M - 4;
    offA[q] = (uint32_t)(((int64_t)k * p.sa_k0 + kdecomp(g, p.nm, p.dim_m, p.sa_m)) * 4);
  }
#pragma unroll
  for (int q = 0; q < TB; ++q) {
    const int f = 256 * (wave + 4 * q) + 4 * lane;
    const int k = f / BN, n = f - k * BN;
    uint32_t g = n0 + n;
    g = g + 4 <= p.N ? g : p.N - 4;
    offB[q] = (uint32_t)(((int64_t)k * p.sb_k0 + kdecomp(g, p.nn, p.dim_n, p.sb_n)) * 4);
  }
  const char* baseA = reinterpret_cast<const char*>(A + boffA);
  const char* baseB = reinterpret_cast<const char*>(B + boffB);
  const int64_t stepA = (int64_t)BK * p.sa_k0 * 4, stepB = (int64_t)BK * p.sb_k0 * 4;
  const bool lead_half = (p.K % BK) != 0;   // K % 16 == 8: a leading half tile of 8 k rows
  const int nfull = (int)(p.K / BK);

  // request the next 16 k rows into stage st (every wave: TA + TB pieces)
#define QK_ISSUE(st_)                                                                                   \
  do {                                                                                                  \
    float* sa_ = stages + (st_) * STAGE;                                                                \
    float* sb_ = sa_ + BK * BM;                                                                         \
    _Pragma("unroll") for (int q = 0; q < TA; ++q) kglds(baseA + offA[q], sa_ + 256 * (wave + 4 * q)); \
    _Pragma("unroll") for (int q = 0; q < TB; ++q) kglds(baseB + offB[q], sb_ + 256 * (wave + 4 * q)); \
    baseA += stepA;                                                                                     \
    baseB += stepB;                                                                                     \
  } while (0)

  acc16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int l31 = lane & 31, kh = lane >> 5;
  const int aoff = kh * BM + wm * (32 * TA);
  const int boff = BK * BM + kh * BN + wn * (32 * TB);

  // ---- main loop, software-pipelined over k-steps AND k-tiles --------------------------------------
  // tile t (stage t % NS) is requested two tiles ahead; its ONE barrier sits inside tile t - 1, three
  // k-steps before the end, between two MFMAs: every wave has then finished reading tile t - 2 (whose stage
  // the request for tile t + 1 may overwrite) and has waited for its own pieces of tile t.  The fragments of
  // k-step s + 1 -- of the next tile's first step at the end of a tile -- are read while step s's
  // MFMAs run, so neither LDS latency nor the barrier ever drains the matrix pipe.
  const int ntiles = nfull + (lead_half ? 1 : 0);
  if (lead_half) {
    // the 8 leading k rows: only the pieces that hold rows 0..7 (piece < 2 T), straight into stage 0
    float* sa = stages;
    float* sb = sa + BK * BM;
#pragma unroll
    for (int q = 0; q < TA; ++q)
      if (wave + 4 * q < 2 * TA) kglds(baseA + offA[q], sa + 256 * (wave + 4 * q));
#pragma unroll
    for (int q = 0; q < TB; ++q)
      if (wave + 4 * q < 2 * TB) kglds(baseB + offB[q], sb + 256 * (wave + 4 * q));
    baseA += stepA / 2;
    baseB += stepB / 2;
  } else {
    QK_ISSUE(0);
  }
  // (the launcher guarantees >= 3 tiles: tiles 1 and 2 exist)
  QK_ISSUE(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TA + TB) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  float fa[2][TA], fb[2][TB];
  kfrag<TA>(fa[0], stages + aoff, l31);
  kfrag<TB>(fb[0], stages + boff, l31);

  // one piece of the next request (q < TA: A pieces, else B pieces) into stage st2_
#define QK_PIECE(q_, st2_)                                                                                    \
  do {                                                                                                        \
    if ((q_) < TA) kglds(baseA + offA[(q_) < TA ? (q_) : 0], stages + (st2_) * STAGE + 256 * (wave + 4 * (q_))); \
    else kglds(baseB + offB[(q_) < TA ? 0 : (q_) - TA], stages + (st2_) * STAGE + BK * BM + 256 * (wave + 4 * ((q_) - TA))); \
  } while (0)

#define QK_TILE(NSTEPS_)                                                                                      \
  do {                                                                                                        \
    const float* As_ = stages + st * STAGE + aoff;                                                            \
    const float* Bs_ = stages + st * STAGE + boff;                                                            \
    const int stn_ = st + 1 >= NS ? 0 : st + 1;                                                               \
    const int st2_ = stn_ + 1 >= NS ? 0 : stn_ + 1;                                                           \
    _Pragma("unroll") for (int s = 0; s < NSTEPS_; ++s) {                                                    \
      /* NS == 2: the request for tile t + 2 overwrites THIS tile's stage, so the barrier is the tile's last */ \
      /* step (every wave has read all of it; the reads have returned: lgkmcnt)                             */ \
      const bool sync_ = (s == NSTEPS_ - (NS == 2 ? 1 : 3));                                                  \
      if (sync_) {                                                                                            \
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                         \
        asm volatile("" ::: "memory");                                                                        \
      }                                                                                                       \
      /* one MFMA first: the wait for THIS step's fragments (read a whole step ago) then precedes the    */ \
      /* next step's reads instead of covering them                                                     */ \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][0], fb[s & 1][0], acc[0][0], 0, 0, 0);       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      if (s + 1 < NSTEPS_) {                                                                                  \
        kfrag<TA>(fa[(s + 1) & 1], As_ + 2 * (s + 1) * BM, l31);                                              \
        kfrag<TB>(fb[(s + 1) & 1], Bs_ + 2 * (s + 1) * BN, l31);                                              \
      } else {                                                                                                \
        kfrag<TA>(fa[0], stages + stn_ * STAGE + aoff, l31);                                                  \
        kfrag<TB>(fb[0], stages + stn_ * STAGE + boff, l31);                                                  \
      }                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      _Pragma("unroll") for (int i = 0; i < TA; ++i) _Pragma("unroll") for (int j = 0; j < TB; ++j)          \
          if (i + j > 0) {                                                                                    \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0); \
            /* the request for tile t + 2, one piece behind each of the MFMAs that follow the barrier     */ \
            if (sync_ && i * TB + j - 1 < TA + TB) {                                                          \
              __builtin_amdgcn_sched_barrier(0);                                                              \
              QK_PIECE(i * TB + j - 1, st2_);                                                                 \
              __builtin_amdgcn_sched_barrier(0);                                                              \
            }                                                                                                 \
          }                                                                                                   \
      if (sync_) {                                                                                            \
        /* (2 x 2 sub-tiles: one piece more than there are MFMAs to put them behind) */                     \
        _Pragma("unroll") for (int q = TA * TB - 1; q < TA + TB; ++q) QK_PIECE(q, st2_);                     \
        /* the base never leaves the last tile: past it the request re-reads those rows (a stage nobody reads again) */ \
        baseA += (t + 3 < ntiles) ? stepA : 0;                                                                \
        baseB += (t + 3 < ntiles) ? stepB : 0;                                                                \
      }                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }                                                                                                         \
    st = stn_;                                                                                                \
  } while (0)

  int st = 0;
  int t = 0;
  if (lead_half) {
    QK_TILE(4);
    t = 1;
  }
  for (; t < ntiles; ++t) QK_TILE(8);
#undef QK_TILE
#undef QK_PIECE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue ---------------------------------------------------------------------------------
  float* Cb = C + boffC;
  const bool vecn = p.vec_c >= TB && (TB == 2 || TB == 4);   // TB consecutive n are contiguous and aligned in C
  if constexpr (DOT) {
    __shared__ double dred[4];
    float dsum = 0.0f;
#pragma unroll
    for (int i = 0; i < TA; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm * (32 * TA) + kmap<TA>(i, (r & 3) + 8 * (r >> 2) + 4 * kh);
        if (m0 + ml >= p.M) continue;
        const int64_t orow = offCm[ml];
        if (vecn) {
          const int nl = wn * (32 * TB) + TB * l31;
          if (n0 + nl < p.N) {
            const float* src = Cb + orow + offCn[nl];
            if constexpr (TB == 4) {
              const f4 t = *reinterpret_cast<const f4*>(src);
              dsum += acc[i][0][r] * t[0] + acc[i][1][r] * t[1] + acc[i][2][r] * t[2] + acc[i][3][r] * t[3];
            } else if constexpr (TB == 2) {
              const f2 t = *reinterpret_cast<const f2*>(src);
              dsum += acc[i][0][r] * t[0] + acc[i][1][r] * t[1];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < TB; ++j) {
            const int nl = wn * (32 * TB) + kmap<TB>(j, l31);
            if (n0 + nl < p.N) dsum += acc[i][j][r] * Cb[orow + offCn[nl]];
          }
        }
      }
    }
    double ds = (double)dsum;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ds += __shfl_down(ds, d, 64);
    if (lane == 0) dred[wave] = ds;
    __syncthreads();
    if (tid == 0) reinterpret_cast<double*>(absmax_out)[blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
    return;
  }
  const float alpha = 1.0f / (kread_scale(scale_a) * kread_scale(scale_b));
  float vmax = 0.0f;
#pragma unroll
  for (int i = 0; i < TA; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * (32 * TA) + kmap<TA>(i, (r & 3) + 8 * (r >> 2) + 4 * kh);
      if (m0 + ml >= p.M) continue;
      const int64_t orow = offCm[ml];
      float v[TB];
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        v[j] = acc[i][j][r] * alpha;
        const int nl = wn * (32 * TB) + kmap<TB>(j, l31);
        if (n0 + nl < p.N) {
          const float av = v[j] < 0.0f ? -v[j] : v[j];
          vmax = av > vmax ? av : vmax;
        }
      }
      if (vecn) {
        const int nl = wn * (32 * TB) + TB * l31;
        if (n0 + nl < p.N) {
          float* dst = Cb + orow + offCn[nl];
          if constexpr (TB == 4) { f4 o = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4*>(dst) = o; }
          else if constexpr (TB == 2) { f2 o = {v[0], v[1]}; *reinterpret_cast<f2*>(dst) = o; }
        }
      } else {
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int nl = wn * (32 * TB) + kmap<TB>(j, l31);
          if (n0 + nl < p.N) Cb[orow + offCn[nl]] = v[j];
        }
      }
    }
  }
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      float o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0) atomicMax(reinterpret_cast<unsigned int*>(absmax_out + ((blockIdx.x * 4 + wave) % QAMD_SLOTS)), __float_as_uint(vmax));
  }
}

// the workgroups' partial inner products in a fixed order, scaled like a stored result would have been
__global__ __launch_bounds__(256) void gemmk_dot_finish_kernel(float* __restrict__ out, const double* __restrict__ partial,
                                                               int n, const float* __restrict__ scale_a,
                                                               const float* __restrict__ scale_b,
                                                               const float* __restrict__ scale_t,
                                                               float* __restrict__ absmax_out) {
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int i = tid; i < n; i += 256) s += partial[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    const double z = ((red[0] + red[1]) + (red[2] + red[3])) /
                     ((double)kread_scale(scale_a) * (double)kread_scale(scale_b) * (double)kread_scale(scale_t));
    out[0] = (float)z;
    if (absmax_out) absmax_out[0] = (float)(z < 0.0 ? -z : z);
  }
}

template <int TA, int TB, int NS, int MINW, bool DOT = false>
static int launch_one(const GettArgs& a, const void* A, const void* B, void* C, const void* sa, const void* sb, void* amax,
                      hipStream_t st) {
  constexpr int BM = 64 * TA, BN = 64 * TB;
  const size_t lds = (size_t)NS * 16 * (BM + BN) * sizeof(float) + (size_t)(BM + BN) * sizeof(int64_t);
  if (lds > 64 * 1024)      // (per launch: the attribute belongs to the CURRENT device; a process-wide flag would skip a second GPU)
    (void)hipFuncSetAttribute((const void*)gemmk_kernel<TA, TB, NS, MINW, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  const unsigned grid = a.tiles_m * a.tiles_n * a.B;
  QAMD_LAUNCH((gemmk_kernel<TA, TB, NS, MINW, DOT>), dim3(grid), dim3(256), lds, st, a, (const float*)A, (const float*)B,
              (float*)C, (const float*)sa, (const float*)sb, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace qamdk

using namespace qamdk;

// (tile, ring stages, workgroups per CU).  The 12- and 9-sub-tile wave tiles run a TWO-stage ring so that two workgroups
// share a CU (2 x 61 KB of LDS, 237 registers): the other workgroup's MFMAs fill this one's barrier, prologue and
// epilogue -- 7776^3: 134.7 -> 139.6 TFLOP/s, 8192^3 on 192 x 256: 125.5 -> 129.9 (profiles/r05_gemmk_two_per_cu.txt).
// 4 x 4 (256 accumulator registers) and the 8-sub-tile shapes (no gain measured) keep three stages.
#define QAMD_GEMMK_CASES                                                                \
  QK_CASE(4, 4, 3, 1) QK_CASE(4, 3, 2, 2) QK_CASE(3, 4, 2, 2) QK_CASE(3, 3, 2, 2)        \
  QK_CASE(4, 2, 3, 1) QK_CASE(2, 4, 3, 1) QK_CASE(3, 2, 3, 2) QK_CASE(2, 3, 3, 2) QK_CASE(2, 2, 3, 2)

// ta, tb in {2, 3, 4}: workgroup tile (64 ta) x (64 tb).  a->tiles_m / tiles_n must be ceil(M / 64 ta), ceil(N / 64 tb);
// a->sa_k0 / sb_k0 the strides of the single K group; a->vec_c >= tb enables vector stores along n.
// Preconditions (host planner): fp32, M % 4 == N % 4 == 0, M, N >= 4, K % 8 == 0, 16-byte aligned operands whose free
// bundles are stride-1 in runs of multiples of 4.
extern "C" int qamd_gemmk_launch(int ta, int tb, const GettArgs* a, const void* A, const void* B, void* C,
                                 const void* scale_a, const void* scale_b, void* absmax_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (a->K < 48 || a->K % 8 || a->M < 4 || a->N < 4 || a->M % 4 || a->N % 4) return -2;
#define QK_CASE(TA_, TB_, NS_, MINW_) \
  if (ta == TA_ && tb == TB_) return launch_one<TA_, TB_, NS_, MINW_>(*a, A, B, C, scale_a, scale_b, absmax_out, st);
  QAMD_GEMMK_CASES
#undef QK_CASE
  return -2;
}

// The same product, consumed by ONE inner product with T (laid out as the result would have been) instead of stored:
// partial[0 .. tiles) receives one double per workgroup; qamd_gemmk_dot_finish sums them in a fixed order and applies the
// operands' scales.  Same preconditions as qamd_gemmk_launch; T 16-byte aligned where a->vec_c says so.
extern "C" int qamd_gemmk_dot_launch(int ta, int tb, const GettArgs* a, const void* A, const void* B, const void* T,
                                     void* partial, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (a->K < 48 || a->K % 8 || a->M < 4 || a->N < 4 || a->M % 4 || a->N % 4) return -2;
#define QK_CASE(TA_, TB_, NS_, MINW_) \
  if (ta == TA_ && tb == TB_)         \
    return launch_one<TA_, TB_, NS_, MINW_, true>(*a, A, B, const_cast<void*>(T), nullptr, nullptr, partial, st);
  QAMD_GEMMK_CASES
#undef QK_CASE
  return -2;
}

extern "C" int qamd_gemmk_dot_finish(void* out, const void* partial, int n, const void* scale_a, const void* scale_b,
                                     const void* scale_t, void* absmax_out, void* stream) {
  QAMD_LAUNCH(gemmk_dot_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (float*)out, (const double*)partial, n,
              (const float*)scale_a, (const float*)scale_b, (const float*)scale_t, (float*)absmax_out);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
