// gemmd.hip -- MFMA-bound fp64 GETT on an LDS-DMA ring, gfx950 only.
//
//   C[b, m, n] = alpha * sum_k A[b, m, k] * B[b, k, n]         (tensor addressing as in gett.hip)
//
// The fp64 matrix pipe is slow per instruction (v_mfma_f64_16x16x4_f64: 2048 flop in 64 cycles of a SIMD), so a
// wave that issues MFMAs back to back needs almost nothing else from the CU -- and the round-1 kernel (gettf.hip:
// operands through registers, one barrier per k-tile that every wave drains into, two co-resident workgroups to hide
// it) still stopped at 0.40 of the peak on the effective-Hamiltonian products of a chi = 512 DMRG step
// (quimb/tensor/tensor_core.py:12393-12448 called from quimb/tensor/tn1d/dmrg.py:626-645): 320 tiles of 128 x 128 on
// 512 tile slots.  This kernel is the fp32 k-outer kernel's recipe (gemmk.hip) rebuilt for fp64 and for BOTH operand
// layouts the matvec meets:
//
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 bytes = two doubles per lane), never through
//    registers.  The LDS side of a DMA is lane-linear, the GLOBAL side is any per-lane address -- so the stage image
//    is laid out for the MFMA fragment reads and each lane fetches whatever belongs at its spot:
//      free-contiguous operand ([k][x], x stride-1):  image [16 k][BX] doubles, 16-byte granule g of row k stored at
//        granule g ^ 8 (k & 1): the fragment reads of k rows 4s and 4s + 1 (one ds_read_b64 pass) fall on disjoint
//        bank halves with no padding;
//      k-contiguous operand ([x][k], k stride-1):     a DMA piece is 8 rows x one 128-byte line (16 k) each -- 8 lanes
//        share a line, so the gather stays coalesced -- stored row by row with the 8 granules of row x permuted by
//        slot ^ (x & 7): the fragment reads of 16 consecutive rows meet at most 2 lanes per bank.
//  * 4-stage ring, software-pipelined over k-steps and k-tiles like gemmk.hip: the ONE barrier per k-tile sits before
//    the last k-step of the previous tile, the request for the tile after next goes out one piece per k-step behind
//    that step's first MFMA, and the next step's fragments are always read under the current step's MFMAs.  k offsets are scalar arithmetic on
//    the K groups (no table reads among the requests, no per-lane divisions behind the barrier: both were measured as
//    ~2000 idle matrix-pipe cycles per k-tile).
//  * workgroup = 8 waves (2 x 4), wave tile (16 TA) x (16 TB), TA in {2..5}, TB in {1, 2}: workgroup tiles
//    (32 TA) x (64 TB) = 64 ... 160 by 64 / 128, ONE workgroup per CU = two waves per SIMD (one wave's fragment reads
//    and barrier waits hide behind the other's MFMAs) -- or, for grids of more workgroups than CUs, TWO per CU on a
//    two-stage ring (qamd_gemmd_launch); the host picks the tile for the fewest rounds on 256 CUs
//    (2560 x 2048 -> 160 x 128: 256 tiles, one round) and k slabs for under-filled grids (reduced by gett.hip's
//    splitk_reduce_kernel).
//  * edge tiles re-read the last valid granule / row (their rows and columns of the tile are never stored).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamdd {

typedef __attribute__((ext_vector_type(4))) double acc4;

__device__ __forceinline__ int64_t ddecomp(uint32_t idx, int n, const uint32_t* dims, const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

__device__ __forceinline__ double dread_scale(const double* slots) {
  if (!slots) return 1.0;
  double m = 0.0;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    double v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.0 ? m : 1.0;
}

// 16 bytes per lane HBM/L2 -> LDS (LDS-DMA): lane l lands at dst + 16 l bytes, dst wave-uniform
__device__ __forceinline__ void dglds(const char* src, double* dst) {
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// One operand's share of a stage: the BX / 8 pieces (64 granules each) of its image are dealt to the 8 waves round robin,
// at most NPW per wave.  ``krow`` = the k row (of 16) whose table entry addresses the piece's granule, ``xoff`` = element
// offset of its free-bundle part.
template <int BX, bool KC>
struct DLoader {
  static constexpr int NP = BX / 8;               // pieces per stage
  static constexpr int NPW = (NP + 7) / 8;        // per wave
  int32_t krow[NPW];
  int64_t xoff[NPW];

  // ``sk_in`` = stride of the innermost K group (its size is a multiple of 16: a k-tile never straddles it), so a piece's
  // offset is (wave-uniform offset of the tile's first k row) + (this lane's constant: free-bundle part + krow * sk_in)
  __device__ __forceinline__ void init(uint32_t x0, uint32_t X, int nx, const uint32_t* dim_x, const int64_t* stride_x,
                                       int64_t sk_in, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int pj = wave + 8 * q;                       // (pj >= NP: this wave sits the piece out)
      if constexpr (KC) {                                // piece = rows 8 pj .. 8 pj + 7, one 128-byte line (16 k) each
        const int x = 8 * pj + (lane >> 3);
        const int kp = (lane & 7) ^ (x & 7);             // LDS slot (lane & 7) of row x holds k-pair slot ^ (x & 7)
        uint32_t g = x0 + x;
        g = g < X ? g : X - 1;
        krow[q] = 2 * kp;
        xoff[q] = ddecomp(g, nx, dim_x, stride_x) + 2 * kp * sk_in;
      } else {                                           // image [16 k][BX / 2] granules, odd rows shifted by 8 granules
        const int G = 64 * pj + lane;
        const int k = (G / (BX / 2)) & 15, gs = G % (BX / 2);
        const int gr = gs ^ (8 * (k & 1));
        uint32_t g = x0 + 2 * gr;
        g = g + 2 <= X ? g : X - 2;
        krow[q] = k;
        xoff[q] = ddecomp(g, nx, dim_x, stride_x) + k * sk_in;
      }
    }
  }
};

template <int TA, int TB, bool AKC, bool BKC, bool SWAP, int NS = 4, int MINB = 1>
__global__ __launch_bounds__(512, MINB) void gemmd_kernel(const GettArgs p, const double* __restrict__ A,
                                                       const double* __restrict__ B, double* __restrict__ C,
                                                       const int64_t* __restrict__ ktab,
                                                       const double* __restrict__ scale_a,
                                                       const double* __restrict__ scale_b,
                                                       double* __restrict__ absmax_out) {
  constexpr int BM = 32 * TA, BN = 64 * TB, BK = 16;
  constexpr int STAGE = BK * (BM + BN);   // doubles per stage: A image, then B image
  extern __shared__ __attribute__((aligned(16))) char dsmem[];
  double* stages = reinterpret_cast<double*>(dsmem);
  int64_t* offCm = reinterpret_cast<int64_t*>(stages + NS * STAGE);
  int64_t* offCn = offCm + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile coordinates: (batch, k-split) slab, then an XCD-aware walk of the tile grid in bands of 4 row tiles ------
  const uint32_t per_batch = p.tiles_m * p.tiles_n;
  const uint32_t slab = blockIdx.x / per_batch;
  const uint32_t pid = blockIdx.x - slab * per_batch;
  const uint32_t bb = slab / p.split_k, ks = slab - bb * p.split_k;
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
  boffC += (int64_t)ks * p.slab_stride;   // split-K: partial sums go to slab ks of the workspace
  const uint32_t m0 = tm * BM, n0 = tn * BN;
  for (int i = tid; i < BM + BN; i += 512) {
    if (i < BM) {
      uint32_t g = m0 + i;
      offCm[i] = ddecomp(g < p.M ? g : p.M - 1, p.nm, p.dim_m, p.sc_m);
    } else {
      uint32_t g = n0 + (i - BM);
      offCn[i - BM] = ddecomp(g < p.N ? g : p.N - 1, p.nn, p.dim_n, p.sc_n);
    }
  }

  typedef DLoader<BM, AKC> LA;
  typedef DLoader<BN, BKC> LB;
  LA la;
  LB lb;
  la.init(m0, p.M, p.nm, p.dim_m, p.sa_m, p.sa_k[p.nk - 1], wave, lane);
  lb.init(n0, p.N, p.nn, p.dim_n, p.sb_n, p.sb_k[p.nk - 1], wave, lane);
  const double* Ab = A + boffA;
  const double* Bb = B + boffB;
  const uint32_t kbeg = ks * p.Kc;
  const uint32_t kend = (kbeg + p.Kc < p.K) ? kbeg + p.Kc : p.K;
  const int nt = (int)((kend - kbeg) / BK);

  // EVERY wave issues the same number of requests per tile (the counted waits below rely on it): a wave without a piece
  // in some round fetches the operand's first 16 bytes into a scratch image nobody reads.  The k offset of a piece is
  // computed in registers from the K groups (a table read here would put ordinary loads between the requests, and the
  // request counter is in order)
  double* dummy = reinterpret_cast<double*>(offCn + BN);
  // the request for one k-tile, piece by piece (``piece`` in [0, P): A's pieces first): the loop threads the pieces
  // between the MFMAs of a k-step, the prologue issues them in one go
  const double* At = Ab;
  const double* Bt = Bb;
  uint32_t krem = 0;                       // rows left in the innermost K group from the tile At / Bt point at
  auto tile_base = [&](uint32_t k0) {
    // the tile's first k row in both operands: wave-uniform, scalar arithmetic (k0 is a multiple of 16 and so is the
    // innermost K group: the rows of the tile differ in that group only, which xoff already carries)
    int64_t ka = 0, kb = 0;
    uint32_t idx = k0;
    for (int g = p.nk - 1; g >= 0; --g) {
      const uint32_t d = p.dim_k[g];
      const uint32_t qd = idx / d, r = idx - qd * d;
      if (g == p.nk - 1) krem = d - r;
      ka += (int64_t)r * p.sa_k[g];
      kb += (int64_t)r * p.sb_k[g];
      idx = qd;
    }
    At = Ab + ka;
    Bt = Bb + kb;
  };
  // the NEXT tile (k0 + 16): two additions while it stays inside the innermost K group, the full decomposition at a
  // group boundary
  auto tile_next = [&](uint32_t k0_next) {
    if (krem > BK) {
      krem -= BK;
      At += (int64_t)BK * p.sa_k[p.nk - 1];
      Bt += (int64_t)BK * p.sb_k[p.nk - 1];
    } else {
      tile_base(k0_next);
    }
  };
  auto issue_piece = [&](int st, int piece) {
    double* sa = stages + st * STAGE;
    double* sb = sa + BK * BM;
    if (piece < LA::NPW) {
      const int q = piece;
      const bool mine = wave + 8 * q < LA::NP;
      dglds(reinterpret_cast<const char*>(mine ? At + la.xoff[q] : Ab), mine ? sa + 128 * (wave + 8 * q) : dummy);
    } else {
      const int q = piece - LA::NPW;
      const bool mine = wave + 8 * q < LB::NP;
      dglds(reinterpret_cast<const char*>(mine ? Bt + lb.xoff[q] : Bb), mine ? sb + 128 * (wave + 8 * q) : dummy);
    }
  };
  constexpr int P = LA::NPW + LB::NPW;      // requests per wave and tile
  static_assert((P + 3) / 4 <= TA * TB, "every piece of a request needs an MFMA of its k-step to hide behind");
  static_assert(NS != 2 || P <= TA * TB, "two-stage ring: the whole request goes out behind the last k-step's MFMAs");

  acc4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = acc4{0.0, 0.0, 0.0, 0.0};

  // prologue: tiles 0 and 1 requested in one go; from then on the request for tile t + 2 is issued DURING tile t, a piece
  // per k-step (its stage held tile t - 2, which every wave left before the barrier that opened tile t)
  tile_base(kbeg);
#pragma unroll
  for (int pc = 0; pc < P; ++pc) issue_piece(0, pc);
  if (nt > 1) {
    tile_next(kbeg + BK);
#pragma unroll
    for (int pc = 0; pc < P; ++pc) issue_piece(1, pc);
  }

  // ---- fragment addressing inside a stage (element indices) ---------------------------------------------------
  //  k-contiguous image: row x at x * 16 doubles, k-pair kp = 2 s + (fk >> 1) in slot kp ^ (x & 7), half fk & 1;
  //  free-contiguous image: row k = 4 s + fk at k * BX, column x ^ 16 on odd rows.
  const int fr = lane & 15, fk = lane >> 4;
  int ia[TA], ib[TB], sa_[4], sb_[4];      // per sub-tile base, per k-step addend
#pragma unroll
  for (int i = 0; i < TA; ++i) {
    const int x = wm * (16 * TA) + 16 * i + fr;
    ia[i] = AKC ? x * 16 + (fk & 1) : fk * BM + (x ^ (16 * (fk & 1)));
  }
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    const int x = wn * (16 * TB) + 16 * j + fr;
    ib[j] = BK * BM + (BKC ? x * 16 + (fk & 1) : fk * BN + (x ^ (16 * (fk & 1))));
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    sa_[s] = AKC ? 2 * ((2 * s + (fk >> 1)) ^ (fr & 7)) : 4 * s * BM;
    sb_[s] = BKC ? 2 * ((2 * s + (fk >> 1)) ^ (fr & 7)) : 4 * s * BN;
  }

  // ---- main loop, software-pipelined over k-steps AND k-tiles (gemmk.hip's schedule, 4 stages) ------------------------
  // The barrier that opens tile t + 1 sits INSIDE tile t, before its last k-step: by then every wave has waited for its
  // own pieces of tile t + 1 (requested during tile t - 1) and has finished tile t - 1.  The request for tile t + 2 goes
  // into the stage tile t - 2 left (free since the barrier that opened tile t) one piece per k-step, each behind the
  // first MFMA of its step; the next k-step's fragments (the next tile's first step at the end of a tile) are read while
  // the current step's MFMAs run.  What is left per k-tile besides MFMAs is the barrier skew: requests, scalar offset
  // updates and LDS latency all sit under MFMAs of the same wave, and the two waves of a SIMD no longer do their
  // housekeeping at the same moment (measured before: ~1000 idle pipe cycles per 4096-cycle tile).
  constexpr int Q012 = P - P / 4;            // pieces of a request issued in k-steps 0..2 (piece pc goes to step pc % 4)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");
  if (nt > 1) { /* (tile 1 may still be in flight: only tile 0 is needed) */ }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  double af[2][TA], bf[2][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i) af[0][i] = stages[ia[i] + sa_[0]];
#pragma unroll
  for (int j = 0; j < TB; ++j) bf[0][j] = stages[ib[j] + sb_[0]];
  int st = 0;
  for (int t = 0; t < nt; ++t) {
    const double* S = stages + st * STAGE;
    const int stn = st + 1 >= NS ? 0 : st + 1;
    const int st2 = stn + 1 >= NS ? 0 : stn + 1;
    const bool more = t + 1 < nt, req = t + 2 < nt;
    if (req) tile_next(kbeg + (uint32_t)(t + 2) * BK);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const bool last = s == BK / 4 - 1;
      if (last && more) {
        // this wave's pieces of tile t + 1 (requested during tile t - 1); the pieces of tile t + 2 issued in steps 0..2
        // of this tile may still be in flight
        // (two-stage ring: the request for tile t + 2 goes out BEHIND this barrier, into this tile's stage -- every wave's
        // reads of it must have returned)
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (req) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q012) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (!last) {                      // the next k-step's fragments, read while this step's MFMAs run
#pragma unroll
        for (int i = 0; i < TA; ++i) af[(s + 1) & 1][i] = S[ia[i] + sa_[s + 1]];
#pragma unroll
        for (int j = 0; j < TB; ++j) bf[(s + 1) & 1][j] = S[ib[j] + sb_[s + 1]];
      } else if (more) {                // ... the next TILE's first step (its barrier is behind us)
        const double* Sn = stages + stn * STAGE;
#pragma unroll
        for (int i = 0; i < TA; ++i) af[0][i] = Sn[ia[i] + sa_[0]];
#pragma unroll
        for (int j = 0; j < TB; ++j) bf[0][j] = Sn[ib[j] + sb_[0]];
      }
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
          // this k-step's share of the request for tile t + 2: pieces s, s + 4, ... behind the first MFMAs
          if (NS == 2) {
            if (req && last && i * TB + j < P) issue_piece(st, i * TB + j);
          } else if (req && s + 4 * (i * TB + j) < P) issue_piece(st2, s + 4 * (i * TB + j));
        }
    }
    st = stn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: lanes (fr) run along C's contiguous bundle ---------------------------------------------------
  double* Cb = C + boffC;
  const double alpha = 1.0 / (dread_scale(scale_a) * dread_scale(scale_b));
  double vmax = 0.0;
#pragma unroll
  for (int i = 0; i < TA; ++i) {
#pragma unroll
    for (int j = 0; j < TB; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ml, nl;
        if (SWAP) {
          ml = wm * (16 * TA) + i * 16 + fr;
          nl = wn * (16 * TB) + j * 16 + fk + 4 * r;
        } else {
          ml = wm * (16 * TA) + i * 16 + fk + 4 * r;
          nl = wn * (16 * TB) + j * 16 + fr;
        }
        if (m0 + ml >= p.M || n0 + nl >= p.N) continue;
        const double v = acc[i][j][r] * alpha;
        Cb[offCm[ml] + offCn[nl]] = v;
        const double av = v < 0.0 ? -v : v;
        vmax = av > vmax ? av : vmax;
      }
    }
  }
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      double o = __shfl_down(vmax, d, 64);
      vmax = o > vmax ? o : vmax;
    }
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned long long*>(absmax_out + ((blockIdx.x * 8 + wave) % QAMD_SLOTS)),
                (unsigned long long)__double_as_longlong(vmax));
  }
}

template <int TA, int TB, bool AKC, bool BKC, int NS = 4, int MINB = 1>
static int launch_swap(const GettArgs& a, int swap, const void* A, const void* B, void* C, const void* ktab,
                       const void* sa, const void* sb, void* amax, hipStream_t st) {
  constexpr int BM = 32 * TA, BN = 64 * TB;
  const size_t lds = (size_t)NS * 16 * (BM + BN) * sizeof(double) + (size_t)(BM + BN) * sizeof(int64_t) + 1024;
  const unsigned grid = a.tiles_m * a.tiles_n * a.B * a.split_k;
  if (swap) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)gemmd_kernel<TA, TB, AKC, BKC, true, NS, MINB>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    QAMD_LAUNCH((gemmd_kernel<TA, TB, AKC, BKC, true, NS, MINB>), dim3(grid), dim3(512), lds, st, a, (const double*)A,
                (const double*)B, (double*)C, (const int64_t*)ktab, (const double*)sa, (const double*)sb, (double*)amax);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)gemmd_kernel<TA, TB, AKC, BKC, false, NS, MINB>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    QAMD_LAUNCH((gemmd_kernel<TA, TB, AKC, BKC, false, NS, MINB>), dim3(grid), dim3(512), lds, st, a, (const double*)A,
                (const double*)B, (double*)C, (const int64_t*)ktab, (const double*)sa, (const double*)sb, (double*)amax);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int TA, int TB, int NS = 4, int MINB = 1>
static int launch_layout(const GettArgs& a, int swap, const void* A, const void* B, void* C, const void* ktab,
                         const void* sa, const void* sb, void* amax, hipStream_t st) {
  if (a.a_kcontig) {
    if (a.b_kcontig) return launch_swap<TA, TB, true, true, NS, MINB>(a, swap, A, B, C, ktab, sa, sb, amax, st);
    return launch_swap<TA, TB, true, false, NS, MINB>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  }
  if (a.b_kcontig) return launch_swap<TA, TB, false, true, NS, MINB>(a, swap, A, B, C, ktab, sa, sb, amax, st);
  return launch_swap<TA, TB, false, false, NS, MINB>(a, swap, A, B, C, ktab, sa, sb, amax, st);
}

}  // namespace qamdd

using namespace qamdd;

// ta in {2..5}, tb in {1, 2}: workgroup tile (32 ta) x (64 tb).  a->tiles_m / tiles_n = ceil(M / 32 ta), ceil(N / 64 tb);
// a->Kc a multiple of 16; a->a_kcontig / b_kcontig say which bundle of each operand holds the stride-1 index.
// Preconditions (host planner): fp64, K % 16 == 0, 16-byte aligned operands; a free-contiguous operand's innermost free
// group is stride-1 with an even size and every other stride of it even; a k-contiguous operand's innermost K group is
// stride-1, every other stride of it even; in BOTH cases the innermost K group's size is a multiple of 16 (a k-tile
// never straddles it).
extern "C" int qamd_gemmd_launch(int ta, int tb, const GettArgs* a, int swap, const void* A, const void* B, void* C,
                                 const void* ktab, const void* scale_a, const void* scale_b, void* absmax_out,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (a->K < 16 || a->K % 16 || a->Kc % 16 || a->M < 2 || a->N < 2) return -2;
  // More workgroups than CUs: the two-stage ring (half the LDS) with 4 waves per SIMD allowed, so that TWO workgroups share
  // a CU and one's barrier waits, prologue and epilogue sit under the other's MFMAs -- 4096^3 on 128 x 128 tiles 63.9 -> 67.5
  // TFLOP/s, 96 x 128: 55.1 -> 61.6, 160 x 64: 51.9 -> 59.3 (profiles/r05_gemmd_two_per_cu.txt).  Grids of at most one
  // workgroup per CU (the chi = 512 matvec's products) keep the four-stage ring: nothing to overlap with, deeper prefetch.
  if ((uint64_t)a->tiles_m * a->tiles_n * a->B * a->split_k > 256) {
#define QD2_CASE(TA_, TB_) \
  if (ta == TA_ && tb == TB_) return launch_layout<TA_, TB_, 2, 4>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
    QD2_CASE(4, 2) QD2_CASE(3, 2) QD2_CASE(4, 1) QD2_CASE(5, 1)
#undef QD2_CASE
  }
#define QD_CASE(TA_, TB_) \
  if (ta == TA_ && tb == TB_) return launch_layout<TA_, TB_>(*a, swap, A, B, C, ktab, scale_a, scale_b, absmax_out, st);
  QD_CASE(4, 2) QD_CASE(5, 2) QD_CASE(3, 2) QD_CASE(2, 2) QD_CASE(4, 1) QD_CASE(5, 1) QD_CASE(3, 1) QD_CASE(2, 1)
#undef QD_CASE
  return -2;
}
