// chain2h.hip -- the fused pair of site absorptions on v_mfma_f32_4x4x1_16b_f32 with TWO waves per SIMD
// (gfx950, fp32, D = 6).  Same mathematics and the same instruction as chain2q.hip:
//
//   X[x, y, v, m] = sum_k1      A[k1, v, m] * W1[k1, (x, y)]
//   C[x, n2,  m]  = sum_{y, v}  X[x, y, v, m] * W2[(y, v), n2]          n2 = (no, ni)
//
// chain2q keeps X for 64 m in one wave (216 registers): 471 registers, ONE wave per SIMD, and a lone wave
// issues one instruction per ~4-5 cycles, so every load, LDS access and address computation of a chunk
// (~900 per 3888 MFMAs) displaces MFMA issue (MFMA pipe 67 % busy).  Here a wave owns 32 m:
//
//  * the two 32-lane halves of a register hold TWO DIFFERENT spectators of the same 32 m: in stage 1 two values
//    of v (lanes 0-31: v = 2p, lanes 32-63: v = 2p + 1 -- one load instruction fetches both 128-byte runs), in
//    stage 2 two values of x.  X for all (x, y, v) is then 108 registers, the whole kernel fits 256, and two
//    waves share every SIMD: while one issues loads / LDS traffic / scalar work the other issues MFMAs.
//  * stage 1 leaves register (row (x, y), pair p) = [X(x, y, 2p) | X(x, y, 2p+1)]; stage 2 wants, for the x pair
//    (x0, x1), the operands [X(x0, y, v) | X(x1, y, v)].  ONE v_permlane32_swap of the registers of rows (x0, y) and
//    (x1, y) produces exactly the two operands v = 2p and v = 2p + 1 -- done once per chunk, in place, after
//    stage 1 (54 swaps), no LDS, no extra registers.
//  * everything else as in chain2q: W fragments packed 16 per register (abid picks), A row groups requested
//    through a register ring on ONE scalar base each, the result of a (group of two `no`, x pair) written to
//    a wave-private LDS tile [2 no][32 m][36] and copied out in 4608-byte runs, all of it threaded one
//    instruction per k-step through the MFMAs of the following work.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

typedef __attribute__((ext_vector_type(4))) float h_acc_t;
typedef float h_vec2 __attribute__((ext_vector_type(2), aligned(8)));
typedef const __attribute__((address_space(1))) char* h_gptr_t;

__device__ __forceinline__ float hload(uint64_t sbase, uint32_t voff) {
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<h_gptr_t>(sbase) + voff);
}

__device__ __forceinline__ float hread_scale(const float* slots) {
  if (!slots) return 1.f;
  float m = 0.f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.f ? m : 1.f;
}

__device__ __forceinline__ h_acc_t hmfma(float a, float b, h_acc_t c, int abid) {
  switch (abid) {
#define QAMD_H_CASE(n) case n: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, n, 0);
    QAMD_H_CASE(0) QAMD_H_CASE(1) QAMD_H_CASE(2) QAMD_H_CASE(3) QAMD_H_CASE(4) QAMD_H_CASE(5) QAMD_H_CASE(6) QAMD_H_CASE(7)
    QAMD_H_CASE(8) QAMD_H_CASE(9) QAMD_H_CASE(10) QAMD_H_CASE(11) QAMD_H_CASE(12) QAMD_H_CASE(13) QAMD_H_CASE(14)
#undef QAMD_H_CASE
    default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
  }
}

__device__ __forceinline__ void hdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
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

__device__ __forceinline__ uint64_t huniform64(uint64_t b) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return ((uint64_t)hi << 32) | lo;
}

template <int D, int K1D, int NOD>
__global__ __launch_bounds__(512, 2) void chain2h_kernel(const Chain2Args p, const float* __restrict__ A,
                                                         const float* __restrict__ W1p,
                                                         const float* __restrict__ W2p, float* __restrict__ C,
                                                         const int64_t* __restrict__ offK1,
                                                         const int64_t* __restrict__ offCo,
                                                         const float* __restrict__ scale_a,
                                                         const float* __restrict__ scale_1,
                                                         const float* __restrict__ scale_2,
                                                         float* __restrict__ absmax_out) {
  static_assert(D == 6 || D == 4 || D == 2, "even D: v pairs, x pairs, D*D a whole number of 4-row tiles");
  constexpr int DD = D * D;
  constexpr int K1 = K1D == 2 ? DD : D;
  constexpr int NH = K1D == 2 ? D : 1;
  constexpr int NT1 = DD / 4;                    // stage-1 row tiles (rows = (x, y))
  constexpr int NPV = D / 2;                     // v pairs
  constexpr int NGRP = NPV * NH;                 // row groups (p, h) of D rows (u)
  constexpr int RGRP = (NGRP % 6 == 0 && NGRP > 6) ? 6 : NGRP;   // row groups in the ring
  constexpr int RING = RGRP * D;
  static_assert(NGRP % RGRP == 0, "ring positions must repeat from chunk to chunk");
  constexpr int NO = NOD ? D : 1;
  constexpr int GN = NOD ? ((D % 4 == 0) ? 1 : 2) : 1;
  constexpr int NG = NO / GN;
  constexpr int RG = GN * D;                     // output rows of a group
  constexpr int NT2 = (RG + 3) / 4;
  constexpr int NP = D / 2;                      // x pairs
  constexpr int NSET = ((NG * NP) % 2 == 0) ? 2 : 3;
  static_assert((NG * NP) % NSET == 0, "the rotation must repeat from chunk to chunk");
  constexpr int LASTSET = (NG * NP - 1) % NSET;
  constexpr int TILE = GN * 32 * DD;             // floats of a group's result tile [no][32 m][x][ni]
  constexpr int NW1 = (NT1 * K1 + 15) / 16;
  constexpr int NW2 = (NG * NT2 * DD + 15) / 16;
  constexpr int NWR = RG / 2;                    // 8-byte LDS writes of one (group, x pair)
  static_assert((32 * DD) % 128 == 0, "a `no` run is a whole number of 512-byte wave stores");
  constexpr int IPN = 32 * DD / 128;             // 8-byte-per-lane copy steps per `no`
  constexpr int CPG = GN * IPN;
  constexpr int NCP = CPG + 1;
  static_assert(NWR + NCP <= DD, "a pair's LDS writes and a group's copy-out fit the k-steps of the next pair");

  extern __shared__ __attribute__((aligned(16))) float h_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, ml = lane & 31;

  // ---- site-tensor fragments (as chain2q) ----------------------------------------------------------------
  float w1r[NW1];
#pragma unroll
  for (int R = 0; R < NW1; ++R) {
    const int idx = 16 * R + (lane >> 2), i = lane & 3;
    const int t = idx / K1, k1 = idx - t * K1;
    const int row = 4 * t + i, x = row / D, y = row - x * D;
    const bool ok = t < NT1;
    const int64_t ko = (K1D == 2) ? (k1 / D) * p.w1s[0] + (k1 % D) * p.w1s[1] : k1 * p.w1s[0];
    const float w = W1p[ok ? ko + x * p.w1s[2] + y * p.w1s[3] : 0];
    w1r[R] = ok ? w : 0.f;
  }
  const float alpha = 1.f / (hread_scale(scale_a) * hread_scale(scale_1) * hread_scale(scale_2));
  float w2r[NW2];
#pragma unroll
  for (int R = 0; R < NW2; ++R) {
    const int idx = 16 * R + (lane >> 2), i = lane & 3;
    const int gt = idx / DD, k2 = idx - gt * DD;          // gt = g * NT2 + t ; k2 = y * D + v
    const int g = gt / NT2, t = gt - g * NT2;
    const int rl = 4 * t + i, n2 = g * RG + rl, no = n2 / D, ni = n2 - no * D;
    const int y = k2 / D, v = k2 - y * D;
    const bool ok = gt < NG * NT2 && rl < RG;
    const float w = W2p[ok ? y * p.w2s[0] + v * p.w2s[1] + no * p.w2s[2] + ni * p.w2s[3] : 0];
    w2r[R] = ok ? w * alpha : 0.f;
  }

  // ---- chunks of 32 m: wave gw takes chunks gw, gw + 8 G, ... ----------------------------------------------
  const uint32_t nwaves = 8 * gridDim.x;
  uint32_t c = blockIdx.x * 8 + wave;
  if (c >= p.chunks) return;

  // per-lane byte offsets: the inner k1 index (u), this lane's m and this lane HALF's v (2p or 2p + 1)
  uint32_t uoff[D];
  int64_t sh = 0;
  {
    const int64_t o0 = (int64_t)huniform64((uint64_t)offK1[0]);
#pragma unroll
    for (int u = 0; u < D; ++u)
      uoff[u] = (uint32_t)((offK1[u] - o0 + ml + (int64_t)half * p.sa_v) * (int64_t)sizeof(float));
    if (K1D == 2) sh = (int64_t)huniform64((uint64_t)(offK1[D] - o0));
    A += o0;
  }
  const uint64_t shb = (uint64_t)(sh * (int64_t)sizeof(float));
  const uint64_t svb2 = (uint64_t)(2 * p.sa_v * (int64_t)sizeof(float));
  int64_t co[NO];
#pragma unroll
  for (int no = 0; no < NO; ++no) co[no] = (int64_t)huniform64((uint64_t)offCo[no]);

  float* Tw = h_smem + wave * (2 * TILE);
  float* Tl = Tw + ml * DD + half * D;           // lane part of the result writes: (m, x of the pair)
  uint32_t lane8 = (uint32_t)lane * 8u;          // lane part of the 8-byte copy-out accesses
  uint32_t tsel = 0;

  float ring[RING];
  h_acc_t X[NPV][NT1];                           // [v pair][row tile]: row = 4 t + r; lanes 0-31 <-> v = 2p, 32-63 <-> v = 2p + 1
  h_acc_t acc[NSET][NT2];                        // lanes 0-31 <-> x = 2 xp, 32-63 <-> x = 2 xp + 1
#pragma unroll
  for (int a = 0; a < NSET; ++a)
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc[a][t] = h_acc_t{0, 0, 0, 0};
  float vmax = 0.f;

  auto chunk_bases = [&](uint32_t chunk, uint64_t& abase, int64_t& cbase) {
    int64_t oa, oc;
    hdecomp2(chunk * 32, p.nm, p.dim_m, p.sa_m, p.sc_m, oa, oc);
    abase = huniform64((uint64_t)(A + oa));
    cbase = (int64_t)huniform64((uint64_t)oc);
  };
  // the D row pairs (u = 0 .. D-1) of row group gi = pv * NH + h
  auto load_group = [&](uint64_t abase, int gi, int pos) {
    const int pv = gi / NH, h = gi - pv * NH;
    uint64_t gb = abase + (uint64_t)pv * svb2 + (uint64_t)h * shb;
    asm volatile("" : "+s"(gb));
#pragma unroll
    for (int u = 0; u < D; ++u) {
      asm volatile("" : "+v"(uoff[u]));
      ring[pos + u] = hload(gb, uoff[u]);
    }
  };

  // write w (0 .. NWR-1) of the (group, x pair) held in acc[set]: two result rows of both x of the pair
  auto side_write = [&](int set, int xp, int w, uint32_t tile) {
    const int rl = 2 * w, nol = rl / D, ni = rl - nol * D;
    h_vec2 val;
    val[0] = acc[set][rl / 4][rl & 3];
    val[1] = acc[set][rl / 4][(rl & 3) + 1];
    *reinterpret_cast<h_vec2*>(Tl + tile + nol * (32 * DD) + (2 * xp) * D + ni) = val;
    asm volatile("" ::: "memory");
  };
  h_vec2 stage[2];
  auto side_copy = [&](int g, int s, uint32_t tile, int64_t cb, bool live) {
    if (s > 0 && live) {
      const int q = s - 1, nol = q / IPN, it = q - nol * IPN;
      const h_vec2 val = stage[q & 1];
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(vmax) : "v"(val[0]), "v"(val[1]));
      uint64_t sb = (uint64_t)(C + cb + co[g * GN + nol] + (it & ~7) * 128);
      asm volatile("" : "+s"(sb), "+v"(lane8));
      typedef __attribute__((address_space(1))) char* h_gwptr_t;
      typedef __attribute__((address_space(1))) h_vec2* h_gvptr_t;
      __builtin_nontemporal_store(val, reinterpret_cast<h_gvptr_t>(reinterpret_cast<h_gwptr_t>(sb) + lane8 + (it & 7) * 512));
    }
    if (s < CPG) {
      const int nol = s / IPN, it = s - nol * IPN;
      asm volatile("" ::: "memory");
      stage[s & 1] = *reinterpret_cast<const h_vec2*>(Tw + tile + nol * (32 * DD) + it * 128 + lane * 2);
    }
  };

  uint64_t abase, nbase;
  int64_t cbase, cprev = 0;
  chunk_bases(c, abase, cbase);
#pragma unroll
  for (int gi = 0; gi < RGRP; ++gi) load_group(abase, gi, gi * D);

  bool have_prev = false;
  uint32_t tprev = 0;

  constexpr int LEFT = NWR + NCP;                              // side operations a chunk leaves to the next one
  constexpr int SPR = (LEFT + NGRP * D - 1) / (NGRP * D);      // ... per row of its stage 1 (1 unless rows are few)
  auto stage1_group = [&](int gi) {
    const int pv = gi / NH, h = gi - pv * NH;
#pragma unroll
    for (int uu = 0; uu < D; ++uu) {
      const int u = D - 1 - uu, k1 = h * D + u;
      const float b = ring[(gi % RGRP) * D + u];
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int idx = t * K1 + k1;
        X[pv][t] = hmfma(w1r[idx / 16], b, (h == 0 && uu == 0) ? h_acc_t{0, 0, 0, 0} : X[pv][t], idx % 16);
      }
#pragma unroll
      for (int s = 0; s < SPR; ++s) {
        const int op = (gi * D + uu) * SPR + s;
        if (op < NWR) side_write(LASTSET, NP - 1, op, tprev);
        else if (op < LEFT) side_copy(NG - 1, op - NWR, tprev, cprev, have_prev);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (gi + RGRP < NGRP) load_group(abase, gi + RGRP, (gi % RGRP) * D);
    else load_group(nbase, gi + RGRP - NGRP, (gi % RGRP) * D);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (;;) {
    const uint32_t cn = c + nwaves;
    const bool more = cn < p.chunks;
    int64_t cnext = cbase;
    nbase = abase;
    if (more) chunk_bases(cn, nbase, cnext);

    // ================= stage 1 ====================================================================================
#pragma unroll
    for (int gi = 0; gi < NGRP; ++gi) stage1_group(gi);

    // ---- (v | v+1) halves -> (x0 | x1) halves: rows (x0, y) and (x1, y) of every pair swap their middle halves ----
#pragma unroll
    for (int pv = 0; pv < NPV; ++pv)
#pragma unroll
      for (int xp = 0; xp < NP; ++xp)
#pragma unroll
        for (int y = 0; y < D; ++y) {
          const int r0 = (2 * xp) * D + y, r1 = (2 * xp + 1) * D + y;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[pv][r0 / 4][r0 & 3]),
                                                           __float_as_uint(X[pv][r1 / 4][r1 & 3]), false, false);
          X[pv][r0 / 4][r0 & 3] = __uint_as_float(sw[0]);     // [X(x0, y, 2 pv) | X(x1, y, 2 pv)]
          X[pv][r1 / 4][r1 & 3] = __uint_as_float(sw[1]);     // [X(x0, y, 2 pv + 1) | X(x1, y, 2 pv + 1)]
        }
    __builtin_amdgcn_sched_barrier(0);

    // ================= stage 2: groups of GN `no`, one x pair (the two lane halves) at a time ====================
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const uint32_t tcur = tsel;
      tsel ^= (uint32_t)TILE;
#pragma unroll
      for (int xp = 0; xp < NP; ++xp) {
        const int set = (g * NP + xp) % NSET, pset = (g * NP + xp + NSET - 1) % NSET;
#pragma unroll
        for (int k2 = 0; k2 < DD; ++k2) {
          const int y = k2 / D, v = k2 - y * D;
          const int brow = (2 * xp + (v & 1)) * D + y;
          const float b = X[v / 2][brow / 4][brow & 3];
#pragma unroll
          for (int t = 0; t < NT2; ++t) {
            const int idx = (g * NT2 + t) * DD + k2;
            acc[set][t] = hmfma(w2r[idx / 16], b, k2 == 0 ? h_acc_t{0, 0, 0, 0} : acc[set][t], idx % 16);
          }
          if (xp > 0) {
            if (k2 < NWR) side_write(pset, xp - 1, k2, tcur);
            else if (xp == 1 && g > 0 && k2 - NWR < NCP) side_copy(g - 1, k2 - NWR, tcur ^ (uint32_t)TILE, cbase, true);
          } else if (g > 0) {
            if (k2 < NWR) side_write(pset, NP - 1, k2, tcur ^ (uint32_t)TILE);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    static_assert(NG == 1 || NP >= 2, "a group's copy-out rides on the next group's second x pair");

    have_prev = true;
    tprev = tsel ^ (uint32_t)TILE;
    cprev = cbase;
    if (!more) break;
    c = cn;
    abase = nbase;
    cbase = cnext;
  }

#pragma unroll
  for (int w = 0; w < NWR; ++w) side_write(LASTSET, NP - 1, w, tprev);
#pragma unroll
  for (int s = 0; s < NCP; ++s) side_copy(NG - 1, s, tprev, cprev, true);

  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, d, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * 8 + wave) % QAMD_SLOTS),
                __float_as_uint(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <int D, int K1D, int NOD>
static int launch_chain2h_d(const Chain2Args& a, const void* A, const void* W1p, const void* W2p, void* C,
                            const void* offK1, const void* offCo, const void* sa, const void* s1, const void* s2,
                            void* amax, hipStream_t st) {
  constexpr int GN = NOD ? ((D % 4 == 0) ? 1 : 2) : 1;
  const size_t lds = (size_t)8 * 2 * GN * 32 * D * D * sizeof(float);
  if (lds > 160 * 1024) return -2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)chain2h_kernel<D, K1D, NOD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  QAMD_LAUNCH((chain2h_kernel<D, K1D, NOD>), dim3(a.grid), dim3(512), lds, st, a, (const float*)A, (const float*)W1p,
              (const float*)W2p, (float*)C, (const int64_t*)offK1, (const int64_t*)offCo, (const float*)sa,
              (const float*)s1, (const float*)s2, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int qamd_chain2h_supported(int dtype, int D) { return dtype == 0 && D == 6; }

extern "C" int qamd_chain2h_launch(int D, int k1_single, int no_n2out, const Chain2Args* a, const void* A,
                                   const void* W1p, const void* W2p, void* C, const void* offK1, const void* offCo,
                                   const void* scale_a, const void* scale_1, const void* scale_2, void* absmax_out,
                                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (k1_single && no_n2out) return -2;
  if (D != 6) return -2;
  if (k1_single) return launch_chain2h_d<6, 1, 1>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
  if (no_n2out) return launch_chain2h_d<6, 2, 0>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
  return launch_chain2h_d<6, 2, 1>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
}
