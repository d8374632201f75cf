// chain2q.hip -- the fused pair of site absorptions on v_mfma_f32_4x4x1_16b_f32 (gfx950, fp32, D = 6 / 4 / 2).
//
//   X[x, y, v, m] = sum_k1      A[k1, v, m] * W1[k1, (x, y)]
//   C[x, n2,  m]  = sum_{y, v}  X[x, y, v, m] * W2[(y, v), n2]          n2 = (no, ni)
//
// chain2r.hip runs the same mathematics on 16x16x4 tiles: 36 output rows fill 3 tiles of 16 (25 % of its
// MFMAs multiply padding), a wave's loads are four 64-byte pieces per instruction, and two waves per SIMD
// park each other at VMEM issue.  Here:
//
//  * v_mfma_f32_4x4x1_16b_f32 = 16 independent 4x4 outer products per instruction.  With cbsz = 4 the A operand
//    of ONE block (abid) is broadcast to all 16, so an instruction is
//        out[4 rows][64 columns] += W[4 rows][k] (x) data[k][64 columns]          (8 cycles, 512 FLOP)
//    36 rows = 9 row tiles exactly: no padding.  A whole site tensor lives in 21 VGPRs (16 (tile, k) fragments
//    of 4 values per register; abid, an immediate, picks the fragment).
//  * the 64 lanes ARE 64 consecutive m: every load instruction covers one 256-byte run of one (k1, v) row, and
//    the B operand of k is simply "the register that row was loaded into".  The stage-1 accumulators have the
//    same lane <-> m correspondence, so register r of X tile t IS the stage-2 B operand of contraction row
//    4t + r: no lane shuffles, no LDS between the stages, and any (y, v) order.
//  * one wave per SIMD (about 400 VGPRs): X for all six v (216 registers) stays resident while stage 2 walks the
//    output in groups of two `no` (12 rows = 3 tiles), two x at a time (6 independent accumulator chains).
//  * A rows stream through a ring of 54 registers: row i + 54 is requested the moment row i has been consumed,
//    i.e. 54 loads (13.5 KB) per wave are in flight through the whole of stage 1 and stage 2.
//  * C wants [no][m][x][ni] (death-ordered layouts: the new legs innermost): a group's result goes through a
//    wave-private LDS tile [2 no][64 m][36] (ds_write_b32 with immediate offsets, ds_read_b128 + 16-byte
//    non-temporal stores in 9216-byte runs), double buffered, and both directions are threaded one
//    instruction at a time through the MFMAs of the FOLLOWING work, never issued as a burst.
//
// Per 64-m chunk: 3888 MFMAs (31.1 K cycles) against 216 loads + 54 stores of 256 / 1024 bytes (110 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

typedef __attribute__((ext_vector_type(4))) float q_acc_t;
typedef float q_vec4 __attribute__((ext_vector_type(4), aligned(16)));
typedef const __attribute__((address_space(1))) char* q_gptr_t;

__device__ __forceinline__ float qload(uint64_t sbase, uint32_t voff) {
#ifdef QAMD_C2Q_NT_LOADS   // experiment: every row is read once, by one wave, as whole 256-byte runs
  return __builtin_nontemporal_load(
      reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<q_gptr_t>(sbase) + voff));
#else
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<q_gptr_t>(sbase) + voff);
#endif
}

__device__ __forceinline__ float qread_scale(const float* slots) {
  if (!slots) return 1.f;
  float m = 0.f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.f ? m : 1.f;
}

// out[4 x 64] += Wfrag(block abid of ``a``) (x) b ; abid must reach the builtin as a literal
__device__ __forceinline__ q_acc_t qmfma(float a, float b, q_acc_t c, int abid) {
  switch (abid) {
#define QAMD_Q_CASE(n) case n: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, n, 0);
    QAMD_Q_CASE(0) QAMD_Q_CASE(1) QAMD_Q_CASE(2) QAMD_Q_CASE(3) QAMD_Q_CASE(4) QAMD_Q_CASE(5) QAMD_Q_CASE(6) QAMD_Q_CASE(7)
    QAMD_Q_CASE(8) QAMD_Q_CASE(9) QAMD_Q_CASE(10) QAMD_Q_CASE(11) QAMD_Q_CASE(12) QAMD_Q_CASE(13) QAMD_Q_CASE(14)
#undef QAMD_Q_CASE
    default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
  }
}

__device__ __forceinline__ void qdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
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

__device__ __forceinline__ uint64_t quniform64(uint64_t b) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// K1D: size-D indices in k1 (2 = (h, u) interior site, 1 = row start); NOD: 1 = n2 = (no, ni), 0 = row end (n2 = ni)
template <int D, int K1D, int NOD>
__global__ __launch_bounds__(256, 1) void chain2q_kernel(const Chain2Args p, const float* __restrict__ A,
                                                         const float* __restrict__ W1p,
                                                         const float* __restrict__ W2p, float* __restrict__ C,
                                                         const int64_t* __restrict__ offK1,
                                                         const int64_t* __restrict__ offCo,
                                                         const float* __restrict__ scale_a,
                                                         const float* __restrict__ scale_1,
                                                         const float* __restrict__ scale_2,
                                                         float* __restrict__ absmax_out) {
  static_assert(D == 2 || D == 4 || D == 6, "D*D must be a whole number of 4-row tiles");
  constexpr int DD = D * D;
  constexpr int K1 = K1D == 2 ? DD : D;          // stage-1 contraction length
  constexpr int NH = K1D == 2 ? D : 1;           // values of the outer k1 index
  constexpr int NT1 = DD / 4;                    // stage-1 row tiles (rows = (x, y))
  constexpr int ROWS = D * K1;                   // A rows of one chunk, consumed in the order (v, k1)
#ifdef QAMD_C2Q_RING54
  constexpr int RING = (ROWS > 54 && ROWS % 54 == 0) ? 54 : ((ROWS <= 48) ? ROWS : ROWS / 2);
#else
  constexpr int RING = (ROWS > 108 && ROWS % 108 == 0) ? 108 : ((ROWS <= 48) ? ROWS : ROWS / 2);   // rows in flight
#endif
  static_assert(ROWS % RING == 0 && RING % D == 0, "ring positions must repeat from chunk to chunk, in whole row groups");
  constexpr int NO = NOD ? D : 1;
  constexpr int GN = NOD ? ((D % 4 == 0) ? 1 : 2) : 1;   // `no` values per stage-2 group
  constexpr int NG = NO / GN;                    // groups per chunk
  constexpr int RG = GN * D;                     // output rows of a group
  constexpr int NT2 = (RG + 3) / 4;              // row tiles of a group (row end, D = 6: 2 tiles, 2 padding rows)
  constexpr int NP = D / 2;                      // x pairs
  constexpr int NSET = ((NG * NP) % 2 == 0) ? 2 : 3;   // accumulator sets, rotated pair by pair
  static_assert((NG * NP) % NSET == 0, "the rotation must repeat from chunk to chunk");
  constexpr int LASTSET = (NG * NP - 1) % NSET;  // set holding a chunk's last pair
  constexpr int TILE = GN * 64 * DD;             // floats of a group's result tile [no][m][x][ni]
  constexpr int NW1 = (NT1 * K1 + 15) / 16;      // registers holding W1 fragments
  constexpr int NW2 = (NG * NT2 * DD + 15) / 16; // registers holding W2 fragments
  constexpr int WPP = 2 * RG;                    // LDS writes of one x pair
  constexpr int CPG = GN * (64 * DD / 256);      // copy-out operations (ds_read_b128 + 16-byte store) of a group
  static_assert((64 * DD) % 256 == 0, "a `no` run is a whole number of 1-KB wave stores");

  extern __shared__ __attribute__((aligned(16))) float q_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- site-tensor fragments: register R, lane (b = lane / 4, i = lane % 4) <-> fragment idx = 16 R + b, row 4 t + i
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
  // 1 / (max|A| max|W1| max|W2|) of the fused exponent stripping is folded into the W2 fragments
  const float alpha = 1.f / (qread_scale(scale_a) * qread_scale(scale_1) * qread_scale(scale_2));
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

  // ---- chunks of 64 m: wave gw takes chunks gw, gw + 4 G, ... ----------------------------------------------
  const uint32_t nwaves = 4 * gridDim.x;
  uint32_t c = blockIdx.x * 4 + wave;
  if (c >= p.chunks) return;

  // per-lane byte offsets of the inner k1 index (u) -- the outer one (h) and v go into the scalar base
  uint32_t uoff[D];
  int64_t sh = 0;
  {
    const int64_t o0 = (int64_t)quniform64((uint64_t)offK1[0]);
#pragma unroll
    for (int u = 0; u < D; ++u) uoff[u] = (uint32_t)((offK1[u] - o0 + lane) * (int64_t)sizeof(float));
    if (K1D == 2) sh = (int64_t)quniform64((uint64_t)(offK1[D] - o0));
    A += o0;
  }
  const uint64_t shb = (uint64_t)(sh * (int64_t)sizeof(float));
  const uint64_t svb = (uint64_t)(p.sa_v * (int64_t)sizeof(float));
  int64_t co[NO];
#pragma unroll
  for (int no = 0; no < NO; ++no) co[no] = (int64_t)quniform64((uint64_t)offCo[no]);

  float* Tw = q_smem + wave * (2 * TILE);        // this wave's two result tiles
  float* Tl = Tw + lane * DD;                    // lane part of the result writes
  uint32_t lane16 = (uint32_t)lane * 16u;  // lane part of the 16-byte copy-out accesses
  uint32_t tsel = 0;                             // tile the NEXT group writes (0 / TILE floats)

  float ring[RING];
  q_acc_t X[D][NT1];
  q_acc_t acc[NSET][2][NT2];                     // [set][x in pair][tile]
#pragma unroll
  for (int a = 0; a < NSET; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int t = 0; t < NT2; ++t) acc[a][b][t] = q_acc_t{0, 0, 0, 0};
  float vmax = 0.f;

  auto chunk_bases = [&](uint32_t chunk, uint64_t& abase, int64_t& cbase) {
    int64_t oa, oc;
    qdecomp2(chunk * 64, p.nm, p.dim_m, p.sa_m, p.sc_m, oa, oc);
    abase = quniform64((uint64_t)(A + oa));
    cbase = (int64_t)quniform64((uint64_t)oc);
  };
  // the D rows (u = 0 .. D-1) of row group gi = v * NH + h of the chunk at ``abase`` -> ring positions pos ..
  // (the base is pinned in SGPRs: every load is  global_load v, v_uoff, s[base])
  auto load_group = [&](uint64_t abase, int gi, int pos) {
    const int v = gi / NH, h = gi - v * NH;
    uint64_t gb = abase + (uint64_t)v * svb + (uint64_t)h * shb;
    asm volatile("" : "+s"(gb));
#pragma unroll
    for (int u = 0; u < D; ++u) {
      asm volatile("" : "+v"(uoff[u]));   // keeps the zero-extension next to the load: SGPR base + 32-bit VGPR offset form
      ring[pos + u] = qload(gb, uoff[u]);
    }
  };

  // ---- side work: the LDS writes of a finished x pair and the copy-out of a finished group -----------------
  // write w (0 .. WPP/2-1) of the pair held in acc[set]: two result rows (8 bytes) of one x
  auto side_write = [&](int set, int xp, int w, uint32_t tile) {
    const int xx = (2 * w) / RG, rl = 2 * w - xx * RG;
    const int x = 2 * xp + xx, nol = rl / D, ni = rl - nol * D;
    typedef float q_vec2 __attribute__((ext_vector_type(2), aligned(8)));
    q_vec2 val;
    val[0] = acc[set][xx][rl / 4][rl & 3];
    val[1] = acc[set][xx][rl / 4][(rl & 3) + 1];
    *reinterpret_cast<q_vec2*>(Tl + tile + nol * (64 * DD) + x * D + ni) = val;
    asm volatile("" ::: "memory");   // the copy-out reads these bytes through another vector type: no reordering across
  };
  // copy-out of group g (tile at ``tile``, chunk C offset cb) in CPG + 1 steps: step s stores what step s - 1 read
  q_vec4 stage[2];
  auto side_copy = [&](int g, int s, uint32_t tile, int64_t cb, bool live) {
    constexpr int IPN = CPG / GN;                // 1-KB pieces per `no`
    if (s > 0 && live) {
      const int q = s - 1, nol = q / IPN, it = q - nol * IPN;
      const q_vec4 val = stage[q & 1];
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(vmax) : "v"(val[0]), "v"(val[1]));
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(vmax) : "v"(val[2]), "v"(val[3]));
      uint64_t sb = (uint64_t)(C + cb + co[g * GN + nol] + (it & ~3) * 256);
      asm volatile("" : "+s"(sb), "+v"(lane16));
      typedef __attribute__((address_space(1))) char* q_gwptr_t;
      typedef __attribute__((address_space(1))) q_vec4* q_gvptr_t;
      __builtin_nontemporal_store(val, reinterpret_cast<q_gvptr_t>(reinterpret_cast<q_gwptr_t>(sb) + lane16 + (it & 3) * 1024));
    }
    if (s < CPG) {
      const int nol = s / IPN, it = s - nol * IPN;
      asm volatile("" ::: "memory");
      stage[s & 1] = *reinterpret_cast<const q_vec4*>(Tw + tile + nol * (64 * DD) + it * 256 + lane * 4);
    }
  };
  constexpr int NWR = WPP / 2;                   // 8-byte LDS writes of one x pair
  constexpr int NCP = CPG + 1;                   // steps of one group's copy-out
  static_assert(RG % 2 == 0 && D % 2 == 0, "row pairs (ni even) never straddle a `no`");

  uint64_t abase, nbase;
  int64_t cbase, cprev = 0;
  chunk_bases(c, abase, cbase);
  constexpr int NGRP = ROWS / D, RGRP = RING / D;   // row groups per chunk / in the ring
#pragma unroll
  for (int gi = 0; gi < RGRP; ++gi) load_group(abase, gi, gi * D);

  bool have_prev = false;                        // a previous chunk's last pair / last group is pending (uniform)
  uint32_t tprev = 0;                            // ... in this tile

  // stage 1 of one row group; SIDE: thread the previous chunk's leftovers (one step per row) through it
  auto stage1_group = [&](int gi, bool side) {
    const int v = gi / NH, h = gi - v * NH;
    // the group's loads were issued u = 0 .. D-1: using the LAST one first makes the one wait cover all
#pragma unroll
    for (int uu = 0; uu < D; ++uu) {
      const int u = D - 1 - uu, k1 = h * D + u;
      const float b = ring[(gi % RGRP) * D + u];
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int idx = t * K1 + k1;
        X[v][t] = qmfma(w1r[idx / 16], b, (h == 0 && uu == 0) ? q_acc_t{0, 0, 0, 0} : X[v][t], idx % 16);
      }
      if (side) {
        const int op = gi * D + uu;
        if (op < NWR) side_write(LASTSET, NP - 1, op, tprev);
        else if (op < NWR + NCP) side_copy(NG - 1, op - NWR, tprev, cprev, have_prev);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the registers are free: request the group RING rows ahead (of this chunk or of the next one)
    if (gi + RGRP < NGRP) load_group(abase, gi + RGRP, (gi % RGRP) * D);
    else load_group(nbase, gi + RGRP - NGRP, (gi % RGRP) * D);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int SGRP = (NWR + NCP + D - 1) / D;  // row groups that carry side work
  static_assert(SGRP <= NGRP, "the leftovers fit one stage 1");

#ifdef QAMD_CHAIN2_TIMING   // experiment builds only: s_memtime stamps per phase, written to absmax_out
  uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_amdgcn_s_memtime();
#define QAMD_QSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); uint64_t now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define QAMD_QSTAMP(i) do {} while (0)
#endif
  for (;;) {
    const uint32_t cn = c + nwaves;
    const bool more = cn < p.chunks;
    int64_t cnext = cbase;
    nbase = abase;                               // (the last chunk re-requests its own rows: no branch in the stream)
    if (more) chunk_bases(cn, nbase, cnext);

    QAMD_QSTAMP(0);   // chunk bookkeeping
    // ================= stage 1: X[v] = W1^T . A[:, v]  (+ the previous chunk's leftovers) =====================
    // (on the first chunk the leftovers are dummies: the accumulators hold zeros, the stores are skipped)
#pragma unroll
    for (int gi = 0; gi < SGRP; ++gi) stage1_group(gi, true);
    QAMD_QSTAMP(1);   // stage 1, row groups carrying the leftovers
#pragma unroll
    for (int gi = SGRP; gi < NGRP; ++gi) stage1_group(gi, false);

    QAMD_QSTAMP(2);   // rest of stage 1
    // ================= stage 2: groups of GN `no`, two x at a time =============================================
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
#pragma unroll
          for (int xx = 0; xx < 2; ++xx) {
            const int row = (2 * xp + xx) * D + y;
            const float b = X[v][row / 4][row & 3];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
              const int idx = (g * NT2 + t) * DD + k2;
              acc[set][xx][t] = qmfma(w2r[idx / 16], b, k2 == 0 ? q_acc_t{0, 0, 0, 0} : acc[set][xx][t], idx % 16);
            }
          }
          // side work of this slot: the previous pair's LDS writes first, then the previous group's copy-out
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
    static_assert(NWR <= DD, "a pair's LDS writes fit the k2 slots of the next pair");
    static_assert(NG == 1 || (NP >= 2 && DD - NWR >= NCP), "a group's copy-out fits the slots of the next group's second pair");

    QAMD_QSTAMP(3);   // stage 2
    have_prev = true;
    tprev = tsel ^ (uint32_t)TILE;               // the tile the last group of this chunk wrote
    cprev = cbase;
    if (!more) break;
    c = cn;
    abase = nbase;
    cbase = cnext;
  }

  // ---- drain: the last chunk's last pair and last group -------------------------------------------------------
#pragma unroll
  for (int w = 0; w < NWR; ++w) side_write(LASTSET, NP - 1, w, tprev);
#pragma unroll
  for (int s = 0; s < NCP; ++s) side_copy(NG - 1, s, tprev, cprev, true);

#ifdef QAMD_CHAIN2_TIMING
  QAMD_QSTAMP(4);   // drain
  if (absmax_out && lane == 0) {
    for (int i = 0; i < 8; ++i) absmax_out[(blockIdx.x * 4 + wave) * 8 + i] = (float)tacc[i];
  }
  return;
#endif
  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, d, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS),
                __float_as_uint(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <int D, int K1D, int NOD>
static int launch_chain2q_d(const Chain2Args& a, const void* A, const void* W1p, const void* W2p, void* C,
                            const void* offK1, const void* offCo, const void* sa, const void* s1, const void* s2,
                            void* amax, hipStream_t st) {
  constexpr int GN = NOD ? ((D % 4 == 0) ? 1 : 2) : 1;
  const size_t lds = (size_t)4 * 2 * GN * 64 * D * D * sizeof(float);
  if (lds > 160 * 1024) return -2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)chain2q_kernel<D, K1D, NOD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  QAMD_LAUNCH((chain2q_kernel<D, K1D, NOD>), dim3(a.grid), dim3(256), lds, st, a, (const float*)A, (const float*)W1p,
              (const float*)W2p, (float*)C, (const int64_t*)offK1, (const int64_t*)offCo, (const float*)sa,
              (const float*)s1, (const float*)s2, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// 64-m chunks on the 4x4x1 multi-block MFMA: fp32, even D <= 6
extern "C" int qamd_chain2q_supported(int dtype, int D) { return dtype == 0 && (D == 6 || D == 4); }

extern "C" int qamd_chain2q_launch(int D, int k1_single, int no_n2out, const Chain2Args* a, const void* A,
                                   const void* W1p, const void* W2p, void* C, const void* offK1, const void* offCo,
                                   const void* scale_a, const void* scale_1, const void* scale_2, void* absmax_out,
                                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (k1_single && no_n2out) return -2;
#define QAMD_C2Q(DD)                                                                                                   \
  case DD:                                                                                                             \
    if (k1_single) return launch_chain2q_d<DD, 1, 1>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st); \
    if (no_n2out) return launch_chain2q_d<DD, 2, 0>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);  \
    return launch_chain2q_d<DD, 2, 1>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
  switch (D) {
    QAMD_C2Q(4) QAMD_C2Q(6)
  }
#undef QAMD_C2Q
  return -2;
}
