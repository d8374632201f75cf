// chain3.hip -- THREE consecutive site absorptions of a boundary sweep in one pass (gfx950, fp32, even D <= 6).
//
//   X1[h1, x | b, c, m] = sum_{h, a}  W1[h, a, h1, x]  A[h, a, b, c, m]
//   X2[h2, y | x, c, m] = sum_{h1, b} W2[h1, b, h2, y] X1[h1, x, b, c, m]
//   C [h3, m, x, y, z]  = sum_{h2, c} W3[h2, c, h3, z] X2[h2, y, x, c, m]
//
// The fused PAIR (chain2r.hip) keeps its intermediate in registers: per 16-m chunk a wave holds D^3 x 16
// values.  A third site needs D^4 x 16 values (83 KB at D = 6) and its stage-3 columns (x, y, m) mix what
// different stage-2 columns (x, c, m) produced, so the state has to be EXCHANGED -- here through LDS:
//
//  * one workgroup (NW waves) owns one chunk of 16 m at a time; its state lives in ONE LDS tile
//    T[r1][r2][r3][r0][16 m]  (rows of 16 floats; r0 = the current horizontal bond, fastest, so that the two
//    rows a 32-lane LDS group touches always differ by an odd row count: bank-conflict-free ds_read_b32 /
//    ds_write_b32 in every stage);
//  * every stage is a set of D*D independent "column groups" (16 columns = the 16 m of one value of the two
//    spectator digits); a group reads its D*D contraction rows (KS ds_read_b32 = the MFMA B operands),
//    runs NT x KS v_mfma_f32_16x16x4_f32 against the site tensor's fragments, and writes its D*D result rows
//    back IN PLACE (the rows it read), so stages only need a workgroup barrier between them;
//  * stage 1 never touches LDS for its input: the B operands come straight from HBM in MFMA layout
//    (dword loads, SGPR base + 32-bit lane offset), issued one whole chunk ahead;
//  * the k order (r0 = 2p + (q & 1), rs = 2u + (q >> 1), k-step = (p, u)) and the same decomposition of the
//    output rows make every LDS address  lane part + wave-uniform group base + compile-time immediate;
//    the 36 -> 48 row padding is arranged so that the padding is three whole (dead) registers;
//  * stage 3 writes its rows with the m coordinate XOR-swizzled by the xyz quad, which makes the transposing
//    copy-out (LDS rows [xyz][h3][m] -> HBM runs [h3][m][xyz], 16-byte stores, 1 KB per wave instruction)
//    two-way instead of 32-way bank conflicted;
//  * the copy-out of chunk u overlaps stage 1 of chunk u+1 (whose results wait in registers for the barrier).
//
// Per chunk and CU: 3 x 972 MFMAs (23.3 K cycles on 4 SIMDs) against 166 KB of HBM traffic (14 K cycles at
// the chip's ~11 B/clk/CU streaming rate) and ~6 K cycles of LDS: the kernel is MFMA-bound, which is the point:
// three sites per pass move 2/3 of the bytes the fused pairs move.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

typedef __attribute__((ext_vector_type(4))) float c3_acc_t;
typedef float c3_vec4 __attribute__((ext_vector_type(4), aligned(16)));
typedef const __attribute__((address_space(1))) char* c3_gptr_t;

__device__ __forceinline__ float c3_load(uint64_t sbase, uint32_t voff) {
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<c3_gptr_t>(sbase) + voff);
}

struct c3_tag2 { static constexpr int value = 2; };
struct c3_tag3 { static constexpr int value = 3; };

__device__ __forceinline__ float c3_scale(const float* slots) {
  if (!slots) return 1.f;
  float m = 0.f;
  for (int i = 0; i < QAMD_SLOTS; ++i) {
    float v = slots[i];
    m = v > m ? v : m;
  }
  return m > 0.f ? m : 1.f;
}

// LDS-only workgroup barrier: this wave's LDS traffic is complete, global loads / stores stay in flight
// (__syncthreads() would also drain vmcnt, i.e. the prefetch of the next chunk)
__device__ __forceinline__ void c3_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void c3_decomp(uint32_t idx, int n, const uint32_t* dims, const int64_t* s1,
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

template <int D, int NW>
__global__ __launch_bounds__(NW * 64) void chain3_kernel(const Chain3Args p, const float* __restrict__ A,
                                                         const float* __restrict__ W1p,
                                                         const float* __restrict__ W2p,
                                                         const float* __restrict__ W3p, float* __restrict__ C,
                                                         const int64_t* __restrict__ offK1,
                                                         const int64_t* __restrict__ offCo,
                                                         const float* __restrict__ scale_a,
                                                         const float* __restrict__ scale_1,
                                                         const float* __restrict__ scale_2,
                                                         const float* __restrict__ scale_3,
                                                         float* __restrict__ absmax_out) {
  static_assert(D == 2 || D == 4 || D == 6, "even D only (pairs of bond values per 32-lane LDS group)");
  constexpr int H = D / 2;
  constexpr int KS = H * H;              // k-steps of 4 (K = D*D exactly)
  constexpr int NR = H * H;              // useful result registers per column group (x 4 lane groups = D*D rows)
  constexpr int NT = (NR + 3) / 4;       // 16-row MFMA tiles
  constexpr int NG = D * D;              // column groups per stage
  constexpr int GPW = (NG + NW - 1) / NW;
  constexpr int D2 = D * D, D3 = D * D * D, D4 = D2 * D2;
  constexpr int NTHR = NW * 64;
  constexpr int TILE = D4 * 16;          // floats
  constexpr int WFS = KS * NT * 64;      // floats per stage of fragment cache
  constexpr int QPP = 4 * D3;            // 16-byte quads per h3 plane of the result tile
  constexpr int XQ = D3 / 4;             // quads per m row
  constexpr int WIP = (QPP + 63) / 64;   // wave iterations per plane
  constexpr int NWI = D * WIP;           // wave iterations of one copy-out
  constexpr int CPW = (NWI + NW - 1) / NW;

  extern __shared__ __attribute__((aligned(16))) float c3_smem[];
  float* T = c3_smem;
  float* Wf = c3_smem + TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;

  // ---- site-tensor fragments -> LDS cache, [stage][ks][tile][lane] ------------------------------------
  {
    const float alpha = 1.f / (c3_scale(scale_a) * c3_scale(scale_1) * c3_scale(scale_2) * c3_scale(scale_3));
    for (int idx = tid; idx < 3 * WFS; idx += NTHR) {
      const int st = idx / WFS, rem = idx - st * WFS;
      const int f = rem >> 6, l = rem & 63;
      const int ks = f / NT, t = f - ks * NT;
      const int i = l & 15, kq = l >> 4, qp = i >> 2, r = i & 3;
      const int R = 4 * t + r;
      const int pp = R % H, up = R / H;
      const int r0n = 2 * pp + (qp & 1), rsn = 2 * up + (qp >> 1);
      float w = 0.f;
      if (R < NR) {
        if (st == 0) {
          const int k = 4 * ks + kq;
          w = W1p[(k / D) * p.w1s[0] + (k % D) * p.w1s[1] + r0n * p.w1s[2] + rsn * p.w1s[3]];
        } else {
          const int r0o = 2 * (ks % H) + (kq & 1), rso = 2 * (ks / H) + (kq >> 1);
          const int64_t* ws = st == 1 ? p.w2s : p.w3s;
          const float* Wp = st == 1 ? W2p : W3p;
          w = Wp[r0o * ws[0] + rso * ws[1] + r0n * ws[2] + rsn * ws[3]];
          if (st == 2) w *= alpha;   // 1 / (max|A| max|W1| max|W2| max|W3|) of the fused exponent stripping
        }
      }
      Wf[idx] = w;
    }
  }
  __syncthreads();

  // ---- this workgroup's chunks: positions 2i, 2i+1 (the two 64-byte halves of the same 128-byte lines of A)
  // go to workgroups b, b + 8 -- neighbours in dispatch order on the same XCD, i.e. the same L2 ---------------
  const uint32_t G = gridDim.x;
  uint32_t pos = blockIdx.x;
  if ((G & 15) == 0) {
    const uint32_t xcd = pos & 7, slot = pos >> 3;
    pos = (((slot >> 1) * 8 + xcd) << 1) | (slot & 1);
  }
  if (pos >= p.chunks) return;
  const uint32_t nchunk = (p.chunks - pos + G - 1) / G;

  // per-lane byte offsets of the stage-1 rows this lane loads: k = 4 ks + q
  uint32_t koff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = (uint32_t)((offK1[4 * ks + q] + j) * (int64_t)sizeof(float));
  const uint64_t sb_b = (uint64_t)(p.sa_b * (int64_t)sizeof(float)), sb_c = (uint64_t)(p.sa_c * (int64_t)sizeof(float));

  // LDS lane parts (floats)
  const int L1 = ((q >> 1) * D3 + (q & 1)) * 16 + j;   // stage-1 results:  rows (x = rs_new, h1 = r0_new)
  const int L2 = ((q >> 1) * D2 + (q & 1)) * 16 + j;   // stage 2 in place:  rows (b | y, h1 | h2)
  const int L3 = ((q >> 1) * D + (q & 1)) * 16 + j;    // stage 3 reads:     rows (c, h2)
  const int L3r = ((q >> 1) * D + (q & 1)) * 16;       // stage 3 writes: row part; the m part is swizzled

  float areg[GPW][KS];   // stage-1 B operands of the NEXT stage 1 (loaded one chunk ahead)
  float x1[GPW][NR];     // stage-1 results waiting for the barrier that frees the tile

  auto chunk_bases = [&](uint32_t chunk, uint64_t& abase, int64_t& cbase) {
    int64_t oa, oc;
    c3_decomp(chunk * 16, p.nm, p.dim_m, p.sa_m, p.sc_m, oa, oc);
    const uint64_t b = (uint64_t)(A + oa);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    abase = ((uint64_t)hi << 32) | lo;
    const uint32_t clo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)oc);
    const uint32_t chi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)oc >> 32));
    cbase = (int64_t)(((uint64_t)chi << 32) | clo);
  };

  auto issue_loads = [&](uint64_t abase) {
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int g = wave + NW * i;
      if (g < NG) {
        const uint64_t gb = abase + (uint64_t)(g / D) * sb_b + (uint64_t)(g % D) * sb_c;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) areg[i][ks] = c3_load(gb, koff[ks]);
      }
    }
  };

  auto stage1 = [&]() {
    float wf1[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) wf1[ks][t] = Wf[(ks * NT + t) * 64 + lane];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int g = wave + NW * i;
      if (g < NG) {
        c3_acc_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = c3_acc_t{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[ks][t], areg[i][ks], acc[t], 0, 0, 0);
#pragma unroll
        for (int R = 0; R < NR; ++R) x1[i][R] = acc[R / 4][R % 4];
      }
    }
  };

  auto write_x1 = [&]() {
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int g = wave + NW * i;
      if (g < NG) {
        float* Tl = T + L1 + ((g / D) * D2 + (g % D) * D) * 16;   // group (b, c)
#pragma unroll
        for (int R = 0; R < NR; ++R) Tl[(2 * (R / H) * D3 + 2 * (R % H)) * 16] = x1[i][R];
      }
    }
  };

  // stages 2 and 3: column group -> KS reads, NT*KS MFMAs, NR writes in place
  auto stage23 = [&](auto stage_tag) {
    constexpr int ST = decltype(stage_tag)::value;        // 2 or 3
    constexpr int RS = ST == 2 ? D2 : D;                  // row stride of the contracted / produced spectator digit
    float wf[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[ks][t] = Wf[(ST - 1) * WFS + (ks * NT + t) * 64 + lane];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int g = wave + NW * i;
      if (g < NG) {
        // stage 2: group (x, c) -> rows x*D3 + c*D;  stage 3: group (x, y) -> rows x*D3 + y*D2
        const int gb = (ST == 2 ? (g / D) * D3 + (g % D) * D : (g / D) * D3 + (g % D) * D2) * 16;
        const float* Tr = T + (ST == 2 ? L2 : L3) + gb;
        float bop[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bop[ks] = Tr[(2 * (ks / H) * RS + 2 * (ks % H)) * 16];
        c3_acc_t acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = c3_acc_t{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ks][t], bop[ks], acc[t], 0, 0, 0);
        if constexpr (ST == 2) {
          float* Tw = T + L2 + gb;
#pragma unroll
          for (int R = 0; R < NR; ++R) Tw[(2 * (R / H) * RS + 2 * (R % H)) * 16] = acc[R / 4][R % 4];
        } else {
          // rows (z = 2 up + (q >> 1), h3 = 2 pp + (q & 1)); m swizzled by the xyz quad: xyz = g*D + z
#pragma unroll
          for (int up = 0; up < H; ++up) {
            const int s = ((g * D + 2 * up + (q >> 1)) >> 2) & 15;
            float* Tw = T + gb + L3r + (j ^ s);
#pragma unroll
            for (int pp = 0; pp < H; ++pp) {
              const int R = up * H + pp;
              Tw[(2 * up * RS + 2 * pp) * 16] = acc[R / 4][R % 4];
            }
          }
        }
      }
    }
  };

  float vmax = 0.f;
  // transposing copy-out of the finished tile: plane h3 = one run of 16 * D^3 floats of C
  auto copy_out = [&](int64_t cbase) {
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int wi = wave + NW * i;
      if (wi < NWI) {
        const int h3 = wi / WIP, sub = wi - h3 * WIP;
        const int t = sub * 64 + lane;
        if (QPP % 64 == 0 || t < QPP) {
          const int m = t / XQ, xq = t - m * XQ;
          const float* Tr = T + (4 * xq * D + h3) * 16 + (m ^ (xq & 15));
          c3_vec4 val;
          val[0] = Tr[0];
          val[1] = Tr[D * 16];
          val[2] = Tr[2 * D * 16];
          val[3] = Tr[3 * D * 16];
          vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(val[0]), fabsf(val[1])), fmaxf(fabsf(val[2]), fabsf(val[3]))));
          float* cp = C + cbase + offCo[h3] + m * D3 + 4 * xq;
          __builtin_nontemporal_store(val, reinterpret_cast<c3_vec4*>(cp));
        }
      }
    }
  };


  // ---- prologue: chunk 0 through stage 1 -----------------------------------------------------------------
  uint64_t abase;
  int64_t cbase, cbase_next = 0;
  chunk_bases(pos, abase, cbase);
  issue_loads(abase);
  stage1();
  if (nchunk > 1) {
    chunk_bases(pos + G, abase, cbase_next);
    issue_loads(abase);
  }
  write_x1();
  c3_barrier();

  for (uint32_t u = 0; u < nchunk; ++u) {
    stage23(c3_tag2{});
    c3_barrier();
    stage23(c3_tag3{});
    c3_barrier();
    const bool more = u + 1 < nchunk;
    const int64_t cb = cbase;
    if (more) {
      stage1();                                  // chunk u+1 (its operands were loaded one chunk ago)
      cbase = cbase_next;
      if (u + 2 < nchunk) {
        chunk_bases(pos + (u + 2) * G, abase, cbase_next);
        issue_loads(abase);                      // chunk u+2
      }
    }
    copy_out(cb);                                // chunk u
    c3_barrier();
    if (more) {
      write_x1();
      c3_barrier();
    }
  }

  if (absmax_out) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, d, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * NW + wave) % QAMD_SLOTS),
                __float_as_uint(vmax));
  }
}

}  // namespace qamd

using namespace qamd;

template <int D, int NW>
static int launch_chain3_d(const Chain3Args& a, const void* A, const void* W1p, const void* W2p, const void* W3p, void* C,
                           const void* offK1, const void* offCo, const void* sa, const void* s1, const void* s2,
                           const void* s3, void* amax, hipStream_t st) {
  constexpr int H = D / 2, KS = H * H, NT = (H * H + 3) / 4;
  const size_t lds = ((size_t)D * D * D * D * 16 + 3 * (size_t)KS * NT * 64) * sizeof(float);
  if (lds > 160 * 1024) return -2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)chain3_kernel<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  QAMD_LAUNCH((chain3_kernel<D, NW>), dim3(a.grid), dim3(NW * 64), lds, st, a, (const float*)A, (const float*)W1p,
              (const float*)W2p, (const float*)W3p, (float*)C, (const int64_t*)offK1, (const int64_t*)offCo,
              (const float*)sa, (const float*)s1, (const float*)s2, (const float*)s3, (float*)amax);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int qamd_chain3_supported(int dtype, int D) { return dtype == 0 && (D == 2 || D == 4 || D == 6); }

extern "C" int qamd_chain3_launch(int D, int nw, const Chain3Args* a, const void* A, const void* W1p, const void* W2p,
                                  const void* W3p, void* C, const void* offK1, const void* offCo, const void* scale_a,
                                  const void* scale_1, const void* scale_2, const void* scale_3, void* absmax_out,
                                  void* stream) {
  hipStream_t st = (hipStream_t)stream;
#define QAMD_C3(DD)                                                                                                     \
  case DD:                                                                                                              \
    if (nw == 12) return launch_chain3_d<DD, 12>(*a, A, W1p, W2p, W3p, C, offK1, offCo, scale_a, scale_1, scale_2, scale_3, absmax_out, st); \
    if (nw == 4) return launch_chain3_d<DD, 4>(*a, A, W1p, W2p, W3p, C, offK1, offCo, scale_a, scale_1, scale_2, scale_3, absmax_out, st);   \
    return launch_chain3_d<DD, 8>(*a, A, W1p, W2p, W3p, C, offK1, offCo, scale_a, scale_1, scale_2, scale_3, absmax_out, st);
  switch (D) {
    QAMD_C3(2) QAMD_C3(4) QAMD_C3(6)
  }
#undef QAMD_C3
  return -2;
}
