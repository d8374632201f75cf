// chain2r.hip -- the fused pair of site absorptions (see chain2.hip for the maths) with the
// intermediate kept in REGISTERS (gfx950 only, fp32, D <= 6).
//
//   X[x, y, v, m] = sum_k1      A[k1, v, m] * W1[k1, (x, y)]
//   C[x, n2,  m]  = sum_{y, v}  X[x, y, v, m] * W2[(y, v), n2]          n2 = (no, ni)
//
// chain2_kernel moves X through a wave-private LDS tile (scatter, re-read, in-place
// rewrite, transposing copy-out: ~225 LDS instructions and ~700 address VALU ops per
// 16-m chunk, MFMA pipe 48% busy).  Here the stage-1 accumulators ARE the stage-2 B
// operands, which works because both the order of W1's columns and the order of the
// stage-2 contraction index are free:
//
//  * v_mfma_f32_16x16x4: lane (j, q) [j = lane & 15, q = lane >> 4] of the D registers
//    holds rows 4q + r of column j; as a B operand the same lane supplies k = q of
//    column j.  So register r of a stage-1 tile can be fed straight back as one k-step
//    of stage 2, provided its four lane groups q hold k values of ONE output x.
//  * W1's columns are therefore ordered in "slots": slot = 4*tile + r belongs to one x
//    (slot / SX, SX = ceil(D/4) slots per x) and lane group q holds y = 4*(slot % SX) + q
//    (a zero column when y >= D).  W2's rows are read in the matching order
//    (y = 4*sg + q, v), and W2's columns are ordered the same way for the output
//    (register R <-> no = R / SX, lane group q <-> ni = 4*(R % SX) + q), which makes the
//    64 lanes of every result write hit 64 different LDS banks at D = 6.
//  * the only LDS traffic left is the result tile Ot[no][m][x][ni] -- exactly the order
//    of C in HBM -- written once (ds_write_b32, immediate offsets) and copied out with
//    ds_read_b128 + 16-byte stores whose addresses are SGPR base + lane*16 + immediate.
//
// Cost: slots of the last sub-block of y carry zeros (D = 6: 12 k-steps per x instead
// of 9), i.e. 378 instead of 324 MFMAs per chunk, in exchange for ~3x fewer LDS
// instructions and almost no address arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

typedef __attribute__((ext_vector_type(4))) float r_acc_t;
typedef float r_vec4 __attribute__((ext_vector_type(4), aligned(16)));
struct r_true { static constexpr bool value = true; };
struct r_false { static constexpr bool value = false; };
typedef const __attribute__((address_space(1))) char* r_gptr_t;

__device__ __forceinline__ float rload(uint64_t sbase, uint32_t voff) {
#ifdef QAMD_C2R_NT_LOADS
  return __builtin_nontemporal_load(
      reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<r_gptr_t>(sbase) + voff));
#else
  // plain (cached) loads: a wave uses only 64 B of every 128-B line, the other half belongs to the
  // neighbouring wave of the workgroup -- the line has to survive in L2 until that wave asks for it
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<r_gptr_t>(sbase) + voff);
#endif
}

__device__ __forceinline__ float rread_scale(const float* slots) {
  if (!slots) return 1.f;
  float m = 0.f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.f ? m : 1.f;
}

__device__ __forceinline__ void rdecomp2(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
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

// K1D: number of size-D indices in k1 (2 = (h, u) interior site, 1 = row start: no h yet);
// NOD: 1 = n2 = (no, ni) interior site, 0 = row end: n2 = ni only (no new horizontal bond).
// SC ("super-chunks"): a wave works on PAIRS of adjacent chunks (32 consecutive m).  Every load
// instruction then covers 2 rows x 128 contiguous bytes -- whole cache lines, half as many VMEM
// instructions, no line shared with another wave -- and two lane swaps (v_permlane32_swap +
// v_permlane16_swap) turn each pair of loaded registers into the MFMA B operands of the two chunks.
template <int D, int K1D, int NOD, bool SC>
__global__ __launch_bounds__(256, 2) void chain2r_kernel(const Chain2Args p, const float* __restrict__ A,
                                                         const float* __restrict__ W1p,
                                                         const float* __restrict__ W2p, float* __restrict__ C,
                                                         const int64_t* __restrict__ offK1,
                                                         const int64_t* __restrict__ offCo,
                                                         const float* __restrict__ scale_a,
                                                         const float* __restrict__ scale_1,
                                                         const float* __restrict__ scale_2,
                                                         float* __restrict__ absmax_out) {
  constexpr int N = D * D, K1 = (K1D == 2 ? D * D : D), KS1 = (K1 + 3) / 4;
  constexpr int NO = NOD ? D : 1;
  constexpr int SX = (D + 3) / 4;        // slots (registers) per x / per no
  constexpr int NSLOT = D * SX;          // stage 1: used rows of the permuted W1 column order, in units of 4 lane groups
  constexpr int NT = (NSLOT + 3) / 4;    // stage 1: 16-row MFMA tiles
  constexpr int NSLOT2 = NO * SX;        // stage 2: registers of the permuted W2 column order
  constexpr int NT2 = (NSLOT2 + 3) / 4;  // stage 2: 16-row MFMA tiles
  constexpr int XPT = 4 / SX;            // x values per tile
  constexpr int CH = 16;                 // m values per chunk (one per lane j)
  constexpr int RUN = CH * N;            // contiguous C elements per (no, chunk):  [m][x][ni]
  constexpr int TILE = NO * RUN;         // result tile of one wave
  constexpr uint32_t CSTRIDE = 4;        // the 4 waves of a workgroup interleave chunks
  static_assert(D >= 2 && D <= 8 && (4 % SX) == 0, "slot scheme needs ceil(D/4) in {1, 2}");
  // D % 4 == 2: the last y sub-block fills only lane groups 0, 1.  v_permlane32_swap packs the
  // half-filled registers of two v values into one, so those k-steps are halved (D = 6: 9 instead of
  // 12 k-steps per x, 324 instead of 378 MFMAs per chunk).
  constexpr bool MERGE = (D % 4) == 2;
  constexpr int SXF = MERGE ? SX - 1 : SX;   // sub-blocks consumed unmerged
  constexpr int NPAIR = MERGE ? D / 2 : 0;   // merged k-steps of the last sub-block

  extern __shared__ __attribute__((aligned(16))) float r_smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;

  // ---- W fragments, straight from global into registers (they stay for the whole kernel) ----
  float wf1[KS1][NT];
#pragma unroll
  for (int s = 0; s < KS1; ++s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int g = j >> 2, r = j & 3;
      const int slot = 4 * nt + r, x = slot / SX, y = 4 * (slot % SX) + g, k1 = 4 * s + kq;
      const bool ok = slot < NSLOT && y < D && k1 < K1;
      const int64_t ko = (K1D == 2) ? (k1 / D) * p.w1s[0] + (k1 % D) * p.w1s[1] : k1 * p.w1s[0];
      const float w = W1p[ok ? ko + x * p.w1s[2] + y * p.w1s[3] : 0];
      wf1[s][nt] = ok ? w : 0.f;
    }
  // 1 / (max|A| max|W1| max|W2|) of the fused exponent stripping is folded into the W2 fragments: the
  // stage-2 accumulators go to LDS as they are (no multiply between the MFMAs and the result writes)
  const float alpha = 1.f / (rread_scale(scale_a) * rread_scale(scale_1) * rread_scale(scale_2));
  float wf2[SXF > 0 ? SXF : 1][D][NT2];
#pragma unroll
  for (int sg = 0; sg < SXF; ++sg)
#pragma unroll
    for (int v = 0; v < D; ++v)
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        const int g = j >> 2, r = j & 3;
        const int R = 4 * nt + r, no = R / SX, ni = 4 * (R % SX) + g, y = 4 * sg + kq;
        const bool ok = R < NSLOT2 && ni < D && y < D;
        const float w = W2p[ok ? y * p.w2s[0] + v * p.w2s[1] + no * p.w2s[2] + ni * p.w2s[3] : 0];
        wf2[sg][v][nt] = ok ? w * alpha : 0.f;
      }
  // merged steps: lane groups 0, 1 <-> (y = 4*(SX-1) + q, v = 2p); groups 2, 3 <-> (y = 4*(SX-1) + q - 2, v = 2p + 1)
  float wf2m[NPAIR > 0 ? NPAIR : 1][NT2];
#pragma unroll
  for (int pr = 0; pr < NPAIR; ++pr)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const int g = j >> 2, r = j & 3;
      const int R = 4 * nt + r, no = R / SX, ni = 4 * (R % SX) + g;
      const int y = 4 * (SX - 1) + (kq & 1), v = 2 * pr + (kq >> 1);
      const bool ok = R < NSLOT2 && ni < D && y < D;
      const float w = W2p[ok ? y * p.w2s[0] + v * p.w2s[1] + no * p.w2s[2] + ni * p.w2s[3] : 0];
      wf2m[pr][nt] = ok ? w * alpha : 0.f;
    }

  // work units: chunks of 16 m, or (SC) super-chunks of 32 m; the 4 waves interleave units
  constexpr uint32_t UW = SC ? 2 : 1;
  const uint32_t units = p.chunks / UW, upb = p.chunks_per_block / UW;
  const uint32_t blk_first = blockIdx.x * upb;
  uint32_t c_end = blk_first + upb;
  if (c_end > units) c_end = units;
  const uint32_t c_begin = blk_first + wave;
  if (c_begin >= c_end) return;
  const uint32_t my_chunks = (c_end - c_begin + CSTRIDE - 1) / CSTRIDE;   // units of this wave

  int64_t o2[2];
  rdecomp2(c_begin * (CH * UW), p.nm, p.dim_m, p.sa_m, p.sc_m, o2);
  uint64_t sbase;
  {
    uint64_t b = (uint64_t)(A + o2[0]);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    sbase = ((uint64_t)hi << 32) | lo;
  }
  int64_t cbase = o2[1];
  const int64_t cstep = (int64_t)(CSTRIDE * CH) * N;   // C stride of m is D*D (block [x][ni])
  const uint32_t svb = (uint32_t)(p.sa_v * (int64_t)sizeof(float));

  // per-lane byte offsets of the k1 rows this lane loads (padded rows -> row 0, zero W1 rows)
  uint32_t koff[KS1];
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    int k = 4 * s + kq;
    koff[s] = (uint32_t)(((k < K1 ? offK1[k] : offK1[0]) + j) * (int64_t)sizeof(float));
  }
  // SC: load h of k-step s covers rows 4s + 2h + (lane >> 5), columns m = lane & 31
  uint32_t koff2[SC ? KS1 : 1][2];
  if constexpr (SC) {
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int k = 4 * s + 2 * h + (lane >> 5);
        koff2[s][h] = (uint32_t)(((k < K1 ? offK1[k] : offK1[0]) + (lane & 31)) * (int64_t)sizeof(float));
      }
  }
  int64_t co[NO];
#pragma unroll
  for (int no = 0; no < NO; ++no) co[no] = offCo[no];

  float* Ow = r_smem + wave * TILE;          // this wave's result tile  [no][m][x][ni]
  float* Ol = Ow + j * N + kq;               // lane part of the write address; the rest is immediate
  const bool last_ok = (4 * (SX - 1) + kq) < D;   // lane group valid in the last ni sub-block

  float vmax = 0.f;
#ifdef QAMD_CHAIN2_ABLATION   // debugging builds only (bits: 1 no stores, 4 no loads in the loop)
  const uint32_t abl = p.ablate;
#else
  constexpr uint32_t abl = 0;
#endif

#ifdef QAMD_CHAIN2_TIMING   // experiment builds only: s_memtime stamps per phase, written to absmax_out
  uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_amdgcn_s_memtime();
#define QAMD_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); uint64_t now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define QAMD_STAMP(i) do {} while (0)
#endif

  // copy-out of the tile written by the previous chunk's stage 2: for every no, RUN contiguous
  // elements of C.  It runs inside the NEXT chunk (after its first stage-1 tile), so its stores
  // are never the newest VMEM operations a wait has to look past.
  auto copy_out = [&](int64_t cinc) {
    __builtin_amdgcn_wave_barrier();
    // (a version flattened over (no, element) with a single predicated tail was measured SLOWER: its per-lane
    //  64-bit store addresses cost more than the five extra basic-block boundaries of this form)
#pragma unroll
    for (int no = 0; no < NO; ++no) {
      float* cp = C + cbase + co[no];      // wave-uniform, 16-byte aligned (host contract)
#pragma unroll
      for (int it = 0; it < (RUN + 255) / 256; ++it) {
        const int e = it * 256 + lane * 4;
        if (RUN % 256 == 0 || e < RUN) {
          r_vec4 val = *reinterpret_cast<const r_vec4*>(Ow + no * RUN + e);
          vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3]))));
          if (!(abl & 1)) __builtin_nontemporal_store(val, reinterpret_cast<r_vec4*>(cp + e));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    cbase += cinc;     // C offset of the tile that stage 2 is about to write
  };

  // A registers, double buffered: the loads of chunk u+1 are issued before chunk u is touched
  // and stay in flight for a whole chunk time (the memory-level parallelism of the kernel)
  float bufA[D][KS1], bufB[D][KS1];
  auto issue_v = [&](float (&dst)[D][KS1], uint64_t base, int v) {
    if (abl & 4) return;
#pragma unroll
    for (int s = 0; s < KS1; ++s) dst[v][s] = rload(base + (uint64_t)v * svb, koff[s]);
    __builtin_amdgcn_sched_barrier(0);   // keep this batch where it is written
  };
  auto issue = [&](float (&dst)[D][KS1], uint64_t base) {
#pragma unroll
    for (int v = 0; v < D; ++v) issue_v(dst, base, v);
  };

  // One chunk.  VMEM order per chunk: [18 stores of the previous tile] ... [54 loads of the next
  // chunk]: a wave can have at most 64 vector-memory instructions outstanding, so stores must not
  // be issued right behind a fresh batch of loads (54 + 18 > 64 would park the wave, MFMAs and
  // all, until HBM answers) -- the loads go out one stage-2 block after the stores instead.
  // SC: raw loads of a super-chunk (bufA <- rows 4s, 4s+1 ; bufB <- rows 4s+2, 4s+3, 32 columns each) ...
  auto issue_pair = [&](uint64_t base) {
    if (abl & 4) return;
#pragma unroll
    for (int v = 0; v < D; ++v)
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        bufA[v][s] = rload(base + (uint64_t)v * svb, koff2[SC ? s : 0][0]);
        bufB[v][s] = rload(base + (uint64_t)v * svb, koff2[SC ? s : 0][1]);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  // ... and the two lane swaps that make bufA the B operands of the even chunk, bufB of the odd one
  auto swap_pair = [&]() {
#pragma unroll
    for (int v = 0; v < D; ++v)
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        const auto a32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(bufA[v][s]), __float_as_uint(bufB[v][s]),
                                                          false, false);
        const auto a16 = __builtin_amdgcn_permlane16_swap(a32[0], a32[1], false, false);
        bufA[v][s] = __uint_as_float(a16[0]);
        bufB[v][s] = __uint_as_float(a16[1]);
      }
  };

  // stage-2 results: two accumulator sets (x even / odd) and the LDS write of one x
  r_acc_t accs[2][NT2];
  auto write_x = [&](const r_acc_t (&acc)[NT2], int x) {
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int R = 4 * t + r, no = R / SX, s2 = R % SX;
        if (R < NSLOT2 && (s2 < SX - 1 || D % 4 == 0)) Ol[no * RUN + x * D + 4 * s2] = acc[t][r];
      }
    // last ni sub-block: lane groups with ni >= D hold padding.  For x < D-1 they may write anyway: their
    // target is (x+1, ni - D), which the NEXT x overwrites with real data -- no predicate, no basic-block
    // boundary; only the last x of the chunk is predicated (its spill would land in the next m row).
    if (D % 4 != 0 && (x < D - 1 || last_ok)) {
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int R = 4 * t + r, no = R / SX, s2 = R % SX;
          if (R < NSLOT2 && s2 == SX - 1) Ol[no * RUN + x * D + 4 * s2] = acc[t][r];
        }
    }
  };

  // mode 0: prefetch the next chunk into ``nxt`` in D batches; 1: no prefetch; 2: (SC) issue the next
  // super-chunk's raw loads once this chunk's last stage-1 tile has consumed the registers
  auto chunk = [&](float (&cur)[D][KS1], float (&nxt)[D][KS1], uint64_t nbase, auto copy_prev_tag, int64_t cinc,
                   int mode) {
    constexpr bool copy_prev = decltype(copy_prev_tag)::value;   // compile time: no branch around the copy-out
#ifdef QAMD_C2R_SYNC
    __builtin_amdgcn_s_barrier();   // keep the 4 waves (adjacent 64-B halves of the same lines) in step
#endif
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      // ---- stage 1: one 16-row tile of X for every v; D independent accumulator chains --------
      r_acc_t X[D];
#pragma unroll
      for (int v = 0; v < D; ++v) X[v] = r_acc_t{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < KS1; ++s)
#pragma unroll
        for (int v = 0; v < D; ++v)
          X[v] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][nt], cur[v][s], X[v], 0, 0, 0);
      QAMD_STAMP(nt == 0 ? 0 : (nt == 1 ? 4 : 6));      // stage 1 of tile nt
      if (nt == 0 && copy_prev) copy_out(cinc);
      if (nt == 0) QAMD_STAMP(1);                       // copy-out
      if (NT == 1 && mode == 0) issue(nxt, nbase);
      if (nt == NT - 1 && mode == 2) issue_pair(nbase);
      QAMD_STAMP(2);
      // ---- stage 2 for the x values of this tile: the stage-1 registers are the B operands ----
#pragma unroll
      for (int xl = 0; xl < XPT; ++xl) {
        const int x = nt * XPT + xl;
        if (x < D) {
          // the next chunk's loads go out in D batches of KS1, one in front of every stage-2 block
          // (a 54-load burst parks the wave at VMEM issue while the CU's memory pipe is busy)
          if (NT > 1 && mode == 0) issue_v(nxt, nbase, x);
          // software pipeline over x: the MFMAs of x are issued first, then the result of x-1 (complete long
          // ago: no MFMA -> LDS-write hazard wait) goes to LDS in the shadow of those MFMAs
          r_acc_t (&acc)[NT2] = accs[x & 1];
#pragma unroll
          for (int t = 0; t < NT2; ++t) acc[t] = r_acc_t{0, 0, 0, 0};
#pragma unroll
          for (int sg = 0; sg < SXF; ++sg)
#pragma unroll
            for (int v = 0; v < D; ++v)
#pragma unroll
              for (int t = 0; t < NT2; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf2[sg][v][t], X[v][xl * SX + sg], acc[t], 0, 0, 0);
#pragma unroll
          for (int pr = 0; pr < NPAIR; ++pr) {
            // lanes 0-31 of X[2p] | lanes 0-31 of X[2p+1] (their lanes 32-63 hold the zero rows y >= D)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(X[2 * pr][xl * SX + SX - 1]),
                                                             __float_as_uint(X[2 * pr + 1][xl * SX + SX - 1]), false, false);
            const float xm = __uint_as_float(sw[0]);
#pragma unroll
            for (int t = 0; t < NT2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf2m[pr][t], xm, acc[t], 0, 0, 0);
          }
          if (x > 0) write_x(accs[(x - 1) & 1], x - 1);
        }
      }
      QAMD_STAMP(nt == 0 ? 3 : (nt == 1 ? 5 : 7));      // stage 2 of tile nt
    }
    write_x(accs[(D - 1) & 1], D - 1);                  // the last x of the chunk
  };

  if constexpr (SC) {
    constexpr uint64_t STEP = (uint64_t)(CSTRIDE * 2 * CH * sizeof(float));
    const int64_t cB = (int64_t)CH * N, cA = (int64_t)(CSTRIDE * 2 * CH - CH) * N;   // C steps: even -> odd chunk, odd -> next even
    issue_pair(sbase);
    for (uint32_t u = 0; u < my_chunks; ++u) {
      swap_pair();
      sbase += (u + 1 < my_chunks) ? STEP : 0;           // (the last super-chunk re-loads itself: no branch)
      if (u == 0) chunk(bufA, bufB, sbase, r_false{}, cA, 1);   // even chunk; nothing to flush yet
      else chunk(bufA, bufB, sbase, r_true{}, cA, 1);           // even chunk; flushes the previous odd tile
      chunk(bufB, bufA, sbase, r_true{}, cB, 2);         // odd chunk; flushes the even tile, then prefetches
    }
    copy_out(0);
  } else {
    constexpr uint64_t STEP = (uint64_t)(CSTRIDE * CH * sizeof(float));
    issue(bufA, sbase);
    uint32_t u = 0;
    if (my_chunks >= 2) {                                  // first pair peeled: its first chunk has nothing to flush
      sbase += STEP;
      chunk(bufA, bufB, sbase, r_false{}, cstep, 0);
      sbase += (2 < my_chunks) ? STEP : 0;
      chunk(bufB, bufA, sbase, r_true{}, cstep, 0);
      u = 2;
    }
    for (; u + 2 <= my_chunks; u += 2) {
      sbase += STEP;
      chunk(bufA, bufB, sbase, r_true{}, cstep, 0);        // chunk u   (prefetches u+1)
      sbase += (u + 2 < my_chunks) ? STEP : 0;             // (the last pair re-loads its own chunk: no branch)
      chunk(bufB, bufA, sbase, r_true{}, cstep, 0);        // chunk u+1 (prefetches u+2)
    }
    if (u < my_chunks) {                                   // odd tail (its prefetch re-loads itself)
      if (u == 0) chunk(bufA, bufB, sbase, r_false{}, cstep, 0);
      else chunk(bufA, bufB, sbase, r_true{}, cstep, 0);
    }
    copy_out(0);
  }

#ifdef QAMD_CHAIN2_TIMING
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

template <int D, int K1D, int NOD, bool SC>
static int launch_chain2r_d(const Chain2Args& a, const void* A, const void* W1p, const void* W2p, void* C,
                            const void* offK1, const void* offCo, const void* sa, const void* s1, const void* s2,
                            void* amax, hipStream_t st) {
  size_t lds = (size_t)4 * (NOD ? D : 1) * 16 * D * D * sizeof(float);
  if (lds > 160 * 1024) return -2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)chain2r_kernel<D, K1D, NOD, SC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  QAMD_LAUNCH((chain2r_kernel<D, K1D, NOD, SC>), dim3(a.grid), dim3(256), lds, st, a, (const float*)A, (const float*)W1p,
              (const float*)W2p, (float*)C, (const int64_t*)offK1, (const int64_t*)offCo, (const float*)sa,
              (const float*)s1, (const float*)s2, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// register-resident variant available for (dtype, D)?  It is the only implementation of the
// row-start (k1 = one index) and row-end (n2 = one index) pair shapes.
extern "C" int qamd_chain2r_supported(int dtype, int D) { return dtype == 0 && D >= 2 && D <= 6; }

extern "C" int qamd_chain2r_launch(int D, int k1_single, int no_n2out, const Chain2Args* a, const void* A,
                                   const void* W1p, const void* W2p, void* C, const void* offK1, const void* offCo,
                                   const void* scale_a, const void* scale_1, const void* scale_2, void* absmax_out,
                                   void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (k1_single && no_n2out) return -2;
#define QAMD_C2R(DD)                                                                                                 \
  case DD:                                                                                                           \
    if (a->sc) return -2;   /* the super-chunk variant (measured equal to the default) is no longer instantiated */     \
    if (k1_single) return launch_chain2r_d<DD, 1, 1, false>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st); \
    if (no_n2out) return launch_chain2r_d<DD, 2, 0, false>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);  \
    return launch_chain2r_d<DD, 2, 1, false>(*a, A, W1p, W2p, C, offK1, offCo, scale_a, scale_1, scale_2, absmax_out, st);
  switch (D) {
    QAMD_C2R(2) QAMD_C2R(3) QAMD_C2R(4) QAMD_C2R(5) QAMD_C2R(6)
  }
#undef QAMD_C2R
  return -2;
}
