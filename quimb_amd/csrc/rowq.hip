// rowq.hip -- one ROW of a 2D boundary sweep as ONE launch on v_mfma_f32_4x4x1_16b_f32 (gfx950, fp32, bond dimension 6, five
// sites), for rows of ANY size: the small rows of a corner sweep (dispatch latency) and its last row (6^9 -> 6^10 elements,
// until round 6 three bandwidth-bound passes over 242 MB: 1.25 GB per corner; here 0.28 GB and MFMA-bound).
//
// quimb absorbs a row of a PEPS-like network into the boundary site by site (quimb/tensor/tn2d/core.py:1393-1402; exact
// mode: every absorption a pairwise contraction of the growing boundary tensor with one site tensor):
//
//   T'[S, d1..d5, h] = sum_{v1..v5, b1..b4}  T[S, v1..v5] W1[v1, d1, b1] W2[v2, b1, d2, b2] ... W5[v5, b4, d5, h]
//
// S -- every other index of the boundary tensor -- is a spectator of the whole row, and so is d1 once the first site is
// absorbed: a work item (S, d1) carries a 6^5-element state through the remaining four sites in LDS (31 KB: three to five
// workgroups per CU).  rowpass.hip (round 5) does this on 16x16x4 tiles: 36 rows = 3 tiles of 16, 44 % of its MFMAs multiply
// padding, and the state is scattered between two images.  Here:
//
//  * the state is ONE image  st[bond][x2][x3][x4][x5]  (strides 1296, 216, 36, 6, 1): position c holds the up leg v_c until
//    site c is absorbed and the new down leg d_c afterwards, the bond slot holds b_{c-1} -> b_c (-> h after the last site).
//    Site c contracts the rows (bond, x_c) and writes its result rows (d_c, b_c) back INTO THE SAME ADDRESSES: the 64 lanes
//    of a wave are 64 columns (a value of every other position), a lane reads its 36 rows, multiplies, and overwrites them.
//    No second image, no scatter: the next site merely enumerates the columns differently (additive per-lane / per-row
//    offsets), one barrier per site.
//  * v_mfma_f32_4x4x1_16b_f32 with cbsz = 4 (one block's A operand broadcast to all 16 blocks) is
//        out[4 rows][64 columns] += W[4 rows][k] (x) state[k][64 columns]            (8 cycles, 512 FLOP)
//    36 rows = 9 row tiles exactly; a site tensor is 21 VGPRs of fragments (abid picks one), fetched for site c + 1 under
//    the MFMAs of site c.  324 MFMAs per wave and site against 36 ds_read_b32 + 36 ds_write_b32.
//  * site 1 (no bond yet, K = 6) reads the boundary tensor from global memory in the same lane = column form (256-byte
//    runs when the up legs are innermost) and leaves st[b1][v2..v5]; the last site's image is copied out as 16-byte
//    vectors: 5 KB runs per (h, item) in the layouts the executor chooses for fused rows and for the join's operand.
//  * down legs / the open leg may be SHORTER than 6 (range-sliced cut bonds of a rank's share: quimb_amd/quadrants.py):
//    fewer items (d1), fewer columns for the later sites, row tiles skipped in groups of three.
//  * workgroup -> item is XCD-aware: the (up to six) items of one S sit on the same XCD, so its 31 KB of boundary tensor
//    come from that XCD's L2 after the first read.
//
// Fused exponent stripping as in chain2*.hip: the result is scaled by 1 / (max|T| max|W1| ... max|W5|) (folded into the last
// site's fragments) and its own absmax recorded; the four intermediates never exist, so they carry no exponent of their own.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

#ifdef QAMD_RQ_ABLATE   // scripts/probes/rowq_probe.hip: ablation bits in RowArgs.pad2_
#define RQ_ABL(bit) (p.pad2_ & (bit))
#else
#define RQ_ABL(bit) false
#endif

#ifdef QAMD_RQ_TIMING   // probe builds only: s_memtime stamps per phase, summed per wave, written to absmax_out
#define RQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const uint64_t n_ = __builtin_amdgcn_s_memtime(); tacc[i] += n_ - tlast; tlast = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RQ_STAMP(i) do {} while (0)
#endif

namespace qamdq {

typedef __attribute__((ext_vector_type(4))) float acc4;
typedef float vec4 __attribute__((ext_vector_type(4), aligned(16)));

constexpr int D = 6, DD = 36, SB = 1296, NF = 21;   // NF: registers of 16 fragments each (9 row tiles x 36 k = 324)

struct RowPtrs {
  const float* W[5];
  const float* scale_w[5];
};

// max over a tensor's 64 absmax slots: one slot per lane, a wave reduction (an item is short: a per-thread walk over
// 6 x 64 slots would cost as much as a whole site)
__device__ __forceinline__ float rq_read_scale(const float* slots, int lane) {
  static_assert(QAMD_SLOTS == 64, "one slot per lane");
  if (!slots) return 1.f;
  float m = slots[lane];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  return m > 0.f ? m : 1.f;
}

// out[4 x 64] += Wfrag(block abid of ``a``) (x) b ; abid must reach the builtin as a literal
__device__ __forceinline__ acc4 rq_mfma(float a, float b, acc4 c, int abid) {
  switch (abid) {
#define QAMD_RQ_CASE(n) case n: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, n, 0);
    QAMD_RQ_CASE(0) QAMD_RQ_CASE(1) QAMD_RQ_CASE(2) QAMD_RQ_CASE(3) QAMD_RQ_CASE(4) QAMD_RQ_CASE(5) QAMD_RQ_CASE(6)
    QAMD_RQ_CASE(7) QAMD_RQ_CASE(8) QAMD_RQ_CASE(9) QAMD_RQ_CASE(10) QAMD_RQ_CASE(11) QAMD_RQ_CASE(12) QAMD_RQ_CASE(13)
    QAMD_RQ_CASE(14)
#undef QAMD_RQ_CASE
    default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 15, 0);
  }
}

// ACONT: the up legs v2..v5 are one contiguous run of the boundary tensor (site 1 loads 12-byte vectors).
// FULL: d3..d5 and h of size 6 and the result's (d2..d5) one contiguous, 16-byte aligned run -- the column maps are
// compile-time (d1 only counts the items, d2 is the outermost column digit: a rank's range-sliced share shortens exactly
// these two) and the copy-out moves whole 16-byte vectors; otherwise the extents e[] (d1..d5, h) are runtime and the copy-out
// is element-wise.
// a global load as  global_load_dword v, v_off, s[base:base+1]: the base stays in SGPRs, the per-lane part is 32 bits
typedef const __attribute__((address_space(1))) char* rq_gptr_t;
__device__ __forceinline__ float rq_gload(uint64_t sbase, uint32_t voff) {
  return *reinterpret_cast<const __attribute__((address_space(1))) float*>(reinterpret_cast<rq_gptr_t>(sbase) + voff);
}
__device__ __forceinline__ uint64_t rq_uniform64(uint64_t b) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// LDS accesses of a site as explicit ds_read_b32 / ds_write_b32 with 16-bit immediate offsets off ONE per-lane base: left to
// the compiler they become ds_read2_b32 pairs (8-bit offsets) off ~19 base registers, issued two MFMAs ahead of their use
__device__ __forceinline__ uint32_t rq_lds_addr(const float* q) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)q;
}
#define RQ_DS_READ(dst, addr, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RQ_DS_WRITE(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")
// the registers named here were loaded by RQ_DS_READ: nothing may use them before the count has dropped to n
#define RQ_WAIT7(n, a) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]))
template <int N0, int N>
__device__ __forceinline__ void rq_touch(float* a) {        // (an empty statement that "redefines" a[N0 .. N0 + N): ordering only)
  if constexpr (N >= 12) {
    asm volatile("" : "+v"(a[N0]), "+v"(a[N0 + 1]), "+v"(a[N0 + 2]), "+v"(a[N0 + 3]), "+v"(a[N0 + 4]), "+v"(a[N0 + 5]), "+v"(a[N0 + 6]),
                 "+v"(a[N0 + 7]), "+v"(a[N0 + 8]), "+v"(a[N0 + 9]), "+v"(a[N0 + 10]), "+v"(a[N0 + 11]));
    rq_touch<N0 + 12, N - 12>(a);
  } else if constexpr (N > 0) {
    asm volatile("" : "+v"(a[N0]));
    rq_touch<N0 + 1, N - 1>(a);
  }
}

// LDS-only barrier: __syncthreads() would also wait for this wave's outstanding global stores (the previous item's copy-out,
// which nothing here depends on) and for the prefetch of the next item's boundary tensor
__device__ __forceinline__ void rq_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool FULL, bool ACONT>
__global__ __launch_bounds__(256, 3) void rowq_kernel(const RowArgs p, const RowPtrs w, const float* __restrict__ A,
                                                      float* __restrict__ C, const float* __restrict__ scale_a,
                                                      float* __restrict__ absmax_out, uint32_t* __restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) float st[];       // [6 bond][6][6][6][6], then the four sites' fragments
#ifdef QAMD_RQ_TIMING
  const uint64_t tentry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- this workgroup's items: workgroups are dealt to the XCDs round-robin, XCD x takes the items [x * per, (x + 1) * per)
  // (the up to six items of one S then share that XCD's L2), its workgroups contiguous equal shares of them.  With a queue
  // (opt-in, plan.kernel 3; rows larger than one round of the chip) they draw them one by one from the XCD's counter instead
  // -- the MFMA pipe serves the OLDEST wave first, so workgroups that share a CU advance at very different speeds and static
  // shares end 27 % apart --, the index for the item after next drawn a whole item ahead of its use.
  uint32_t it, it_end, xlo, nx;
  const uint32_t xcd = blockIdx.x & 7;
  {
    const uint32_t per = (p.items + 7) >> 3, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    xlo = xcd * per;
    const uint32_t xhi = (xlo + per < p.items) ? xlo + per : p.items;
    nx = xhi > xlo ? xhi - xlo : 0;
    it = xlo + (uint32_t)(((uint64_t)slot * nx) / nslot);
    it_end = xlo + (uint32_t)(((uint64_t)(slot + 1) * nx) / nslot);
    if (!queue && it >= it_end) return;
  }
  // Experiment (plan.kernel 4 / 5 = RowArgs.pad2_ bits 1024 / 128): distinct wave priorities for the workgroups that share a
  // CU -- they run identical code from the same start, and with equal priority their MFMA phases and their LDS / global
  // phases tend to coincide.  Which workgroups share a CU is the dispatcher's business, though: on the last row alone the two
  // maps below were 257 / 267 us against 287 without in one run and 303 / 249 against 263 in another, and in the step
  // (four corners side by side) nothing at all (profiles/r06_rowq_variants.txt).  Off by default.
  if (p.pad2_ & (1024u | 128u)) {
    const uint32_t pr = (p.pad2_ & 128u) ? (blockIdx.x >> 8) % 3 : (blockIdx.x >> 3) % 3;
    if (pr == 0) __builtin_amdgcn_s_setprio(0);
    else if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(2);
  }
  uint32_t ext[6];                                                  // extents of d1..d5, h
#pragma unroll
  for (int c = 0; c < 5; ++c) ext[c] = (FULL && c > 1) ? D : p.ed[c];    // (d1 only counts the items, d2 is the OUTERMOST column digit of the later sites: both always runtime)
  ext[5] = FULL ? D : p.eh;
  const int blk = lane >> 2, li = lane & 3;

  // ---- once per workgroup: the site-tensor fragments.  Register R, lane (blk, li) <-> fragment idx = 16 R + blk = t * 36 + k,
  // row 4 t + li = (d, b').  Wave c - 1 gathers site c + 1's (W[c], c = 1..4) into LDS in register order -- wl[c - 1][R][lane] --
  // so a site starts with 21 conflict-free ds_read_b32 instead of holding two sites' fragments in registers.
  float* wl = st + D * SB;
  const uint32_t st_addr = rq_lds_addr(st), wl_addr = rq_lds_addr(wl);
  {
    float alpha = 1.f / rq_read_scale(scale_a, lane);
#pragma unroll
    for (int c = 0; c < 5; ++c) alpha /= rq_read_scale(w.scale_w[c], lane);
    const int c = wave + 1;
    const float* Wc = w.W[c];
    const uint32_t s0 = (uint32_t)p.ws[c][0], s1 = (uint32_t)p.ws[c][1], s2 = (uint32_t)p.ws[c][2], s3 = (uint32_t)p.ws[c][3];
    const float scale = c == 4 ? alpha : 1.f;
    float tmp[NF];
#pragma unroll
    for (int R = 0; R < NF; ++R) {
      const int idx = 16 * R + blk;
      const int t = idx / DD, k = idx - t * DD, row = 4 * t + li;
      const int dd = row / D, bp = row - dd * D, b = k / D, v = k - b * D;
      bool ok = t < 9;
      if (!FULL || c == 1) ok = ok && (uint32_t)dd < ext[c] && (c < 4 || (uint32_t)bp < ext[5]);
      const uint32_t off = ok ? v * s0 + b * s1 + dd * s2 + bp * s3 : 0u;
      tmp[R] = RQ_ABL(32) ? 0.5f : Wc[off];
      tmp[R] = ok ? tmp[R] * scale : 0.f;
    }
#pragma unroll
    for (int R = 0; R < NF; ++R) wl[((c - 1) * NF + R) * 64 + lane] = tmp[R];
  }
  // site 1's tensor for every d1 -- w0l[d1][v1][b1], 216 values -- gathered once per workgroup too
  float* w0l = wl + 4 * NF * 64;
  volatile uint32_t* qv = reinterpret_cast<volatile uint32_t*>(w0l + D * DD);   // queue mode: qv[k & 1] = XCD-local index of this workgroup's k-th item
  uint32_t qnext = 0;                                               // (thread 0) the index drawn for the item after next
  const uint32_t nslot = (gridDim.x + 7 - xcd) >> 3;                // workgroups of this XCD
  if (queue && tid == 0) {      // the first two items are dealt, not drawn: 768 workgroups drawing at once queue up at the L2
    qv[0] = blockIdx.x >> 3;
    qv[1] = nslot + (blockIdx.x >> 3);
  }
  if (tid < D * DD) {
    const int dq = tid / DD, v = (tid / D) % D, b = tid % D;
    const bool ok = (uint32_t)dq < ext[0];
    const float x = w.W[0][ok ? v * (uint32_t)p.ws[0][0] + dq * (uint32_t)p.ws[0][2] + b * (uint32_t)p.ws[0][3] : 0u];
    w0l[tid] = ok ? x : 0.f;
  }
  // site 1 is plain FMAs (K = 6, six result rows): thread (v2, v3, v4) owns the six columns v5 = 0..5 -- 216 threads --, so that
  // with the up legs innermost in the boundary tensor (ACONT: what a fused row hands to the next) its 24 bytes per v1 are two
  // 12-byte loads: a third of the load instructions of a lane-per-column form, and a workgroup's loads are issue-bound
  const bool act1 = tid < DD * D;
  uint32_t ao;                                                      // byte offset of this thread's columns at v1 = 0, v5 = 0
  {
    const int t_ = act1 ? tid : 0;
    const int v4 = t_ % D, v3 = (t_ / D) % D, v2 = t_ / DD;
    ao = 4u * (v2 * (uint32_t)p.sv[1] + v3 * (uint32_t)p.sv[2] + v4 * (uint32_t)p.sv[3]);
  }
  const uint32_t sv0 = 4u * (uint32_t)p.sv[0], sv4 = 4u * (uint32_t)p.sv[4];
  auto decode = [&](uint32_t item, uint32_t& d1, int64_t& abase, int64_t& cbase) {
    d1 = item % ext[0];
    uint32_t sidx = item / ext[0];
    abase = 0;
    cbase = (int64_t)d1 * p.sd[0];
    for (int g = p.nS - 1; g >= 0; --g) {
      const uint32_t dg = p.dimS[g], q = sidx / dg, r = sidx - q * dg;
      abase += (int64_t)r * p.sSa[g];
      cbase += (int64_t)r * p.sSc[g];
      sidx = q;
    }
  };
  float an[DD];                                                     // the NEXT item's boundary-tensor elements [v1][v5]
  auto fetch = [&](int64_t abase) {
    const uint64_t gb = rq_uniform64((uint64_t)(A + abase));
    uint64_t gbs[D];
#pragma unroll
    for (int v = 0; v < D; ++v) {
      gbs[v] = gb + (uint64_t)v * sv0;              // the v1 part of the address is uniform: it goes into the SGPR base
      asm volatile("" : "+s"(gbs[v]));
    }
    if (act1) {
#pragma unroll
      for (int v = 0; v < D; ++v) {
        const uint64_t gbv = gbs[v];
        if (ACONT) {
          typedef float vec3 __attribute__((ext_vector_type(3), aligned(4)));
          const __attribute__((address_space(1))) char* q = reinterpret_cast<rq_gptr_t>(gbv) + ao;
          const vec3 lo = *reinterpret_cast<const __attribute__((address_space(1))) vec3*>(q);
          const vec3 hi = *reinterpret_cast<const __attribute__((address_space(1))) vec3*>(q + 12);
          an[v * D + 0] = lo[0]; an[v * D + 1] = lo[1]; an[v * D + 2] = lo[2];
          an[v * D + 3] = hi[0]; an[v * D + 4] = hi[1]; an[v * D + 5] = hi[2];
        } else {
#pragma unroll
          for (int c5 = 0; c5 < D; ++c5) an[v * D + c5] = rq_gload(gbv, ao + c5 * sv4);
        }
        if (RQ_ABL(2)) {
#pragma unroll
          for (int c5 = 0; c5 < D; ++c5) an[v * D + c5] = 1.f;
        }
      }
    }
  };
  uint32_t d1, kitem = 0;                                           // kitem: how many items this workgroup has started
  int64_t abase, cbase;
  __syncthreads();              // the fragment images (site 1 reads w0l first) and the queue words are in place
  if (queue) {
    const uint32_t q0 = qv[0];
    if (q0 >= nx) {                                                 // nothing left for this workgroup (it still counts as finished)
      if (tid == 0) {
        __threadfence();
        if (atomicAdd(queue + 8, 1u) == gridDim.x - 1) {
          for (int x = 0; x < 9; ++x) queue[x] = 0;
        }
      }
      return;
    }
    it = xlo + q0;
  }
  decode(it, d1, abase, cbase);
  fetch(abase);
  float vmax = 0.f;
#ifdef QAMD_RQ_TIMING
  uint64_t tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = __builtin_amdgcn_s_memtime();
  tacc[8] = tlast - tentry;     // prologue
#endif

  for (;;) {
    // (per-lane index maps of the sites and of the copy-out are recomputed per item from an opaque copy of the thread id: left
    // to itself the compiler hoists all of them out of this loop and then spills the prefetched operands to make room)
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    const int lanev = tidv & 63;
    const uint32_t rot = kitem + (blockIdx.x >> 3);
    // ---- site 1: st[b1][v2..v5] = sum_v1 W1[v1, d1, b1] T[S, v1, v2..v5]
    {
      float w0s[DD];                                                // [v1][b1] of this item's d1: 9 broadcast reads
      const vec4* w0p = reinterpret_cast<const vec4*>(w0l + d1 * DD);
#pragma unroll
      for (int i = 0; i < DD / 4; ++i) {
        const vec4 x = w0p[i];
        w0s[4 * i] = x[0]; w0s[4 * i + 1] = x[1]; w0s[4 * i + 2] = x[2]; w0s[4 * i + 3] = x[3];
      }
      if (tidv < DD * D) {
        typedef float vec2 __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
        for (int b = 0; b < D; ++b) {
          float o[D];
#pragma unroll
          for (int c5 = 0; c5 < D; ++c5) {
            float x = 0.f;
#pragma unroll
            for (int v = 0; v < D; ++v) x = __builtin_fmaf(w0s[v * D + b], an[v * D + c5], x);
            o[c5] = x;
          }
          vec2* dst = reinterpret_cast<vec2*>(st + b * SB + D * tidv);
          dst[0] = vec2{o[0], o[1]};
          dst[1] = vec2{o[2], o[3]};
          dst[2] = vec2{o[4], o[5]};
        }
      }
    }
    RQ_STAMP(0);    // site 1 (waits for the prefetched operands)
    if (RQ_ABL(64)) return;
    // the next item's operands are requested now (site 1 has consumed the registers) and arrive under the four sites
    const int64_t ccur = cbase;
    bool more;
    {
      uint32_t nit = it + 1;
      if (queue) {
        const uint32_t qn = qv[(kitem + 1) & 1];
        more = qn < nx;
        nit = xlo + qn;
      } else {
        more = nit < it_end;
      }
      __builtin_amdgcn_sched_barrier(0);      // (the loads below reuse the registers site 1 has just consumed: no hoisting)
      if (more) {
        decode(nit, d1, abase, cbase);
        fetch(abase);
        if (queue && tid == 0) qnext = 2 * nslot + atomicAdd(queue + xcd, 1u);     // ... and the index of the item after next drawn
      }
      it = nit;
      __builtin_amdgcn_sched_barrier(0);
    }
    RQ_STAMP(1);    // decode + issue of the prefetch

    // ---- sites 2 .. 5, in place ---------------------------------------------------------------------------------------
#pragma unroll
    for (int c = 2; c <= 5; ++c) {
      constexpr int SL[4] = {216, 36, 6, 1};                        // strides of the positions 2..5
      const int sc = SL[c - 2];
      rq_barrier();                                                 // the previous site's image is complete
      RQ_STAMP(2);  // barriers before the sites
      if (c == 5 && queue && tid == 0 && more) qv[kitem & 1] = qnext;   // (this item's slot: read for the last time before site 2)
      // this wave's columns: a value of every position but c (d_j before c, v_j after it), position 5 fastest
      uint32_t ncols = 1;
#pragma unroll
      for (int s = 2; s <= 5; ++s)
        if (s != c) ncols *= (s < c ? ext[s - 1] : (uint32_t)D);
      // (fewer than four column tiles -- a sliced d2 halves the columns of sites 3..5 -- would leave the same SIMDs idle
      // in every workgroup of the CU: the tile -> wave map rotates with the item and the workgroup)
      const uint32_t wrot = (wave + rot) & 3u;
      const uint32_t j0 = 64u * wrot + lanev;
      if (64u * wrot < ncols) {
        uint32_t j = j0 < ncols ? j0 : 0, f = 0;
#pragma unroll
        for (int s = 5; s >= 2; --s)
          if (s != c) {
            const uint32_t es = s < c ? ext[s - 1] : (uint32_t)D;
            const uint32_t q = j / es;
            f += (j - q * es) * SL[s - 2];
            j = q;
          }
        // issue order: the fragments of the first three row tiles (7 registers), the lane's 36 state rows, the other 14
        // fragment registers; the first group of MFMAs starts when all but those 14 have arrived
        float cur[NF], bx[DD];
        const uint32_t wa = wl_addr + 4u * lanev, fa = st_addr + 4u * f;
#pragma unroll
        for (int R = 0; R < 7; ++R) RQ_DS_READ(cur[R], wa, ((c - 2) * NF + R) * 256);
#pragma unroll
        for (int k = 0; k < DD; ++k) RQ_DS_READ(bx[k], fa, ((k / D) * SB + (k % D) * sc) * 4);
#pragma unroll
        for (int R = 7; R < NF; ++R) RQ_DS_READ(cur[R], wa, ((c - 2) * NF + R) * 256);
        RQ_WAIT7(14, cur);
        rq_touch<0, DD>(bx);
        RQ_STAMP(3);  // LDS reads of the sites
        if (RQ_ABL(8)) {
#pragma unroll
          for (int k = 0; k < DD; ++k) bx[k] = 1.f;
        }
        // three row tiles at a time (12 result rows = two values of d_c): their results go back into the image at once --
        // this lane's 36 input rows are all in registers by now, so overwriting them is safe -- and only 12 accumulator
        // registers are live beside the next item's prefetched operands
        const uint32_t nrow = D * ext[c - 1];                       // result rows (d_c, b_c), d_c outermost
        const bool wr = j0 < ncols && !RQ_ABL(16);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          if (g == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            rq_touch<7, NF - 7>(cur);
          }
          if ((FULL && c != 2) || 12u * g < nrow) {
            acc4 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = acc4{0, 0, 0, 0};
            if (!RQ_ABL(4)) {
#pragma unroll
              for (int k = 0; k < DD; ++k)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                  const int idx = (3 * g + t) * DD + k;
                  acc[t] = rq_mfma(cur[idx / 16], bx[k], acc[t], idx % 16);
                }
            }
            if (wr) {
#pragma unroll
              for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int row = 4 * (3 * g + t) + r, dd = row / D, bp = row - dd * D;
                  if ((FULL && c != 2) || ((uint32_t)dd < ext[c - 1] && (c < 5 || (uint32_t)bp < ext[5])))
                    RQ_DS_WRITE(fa, acc[t][r], (bp * SB + dd * sc) * 4);
                }
            }
          }
        }
        RQ_STAMP(4);  // MFMAs + LDS writes of the sites
      }
    }
    rq_barrier();
    RQ_STAMP(5);    // barrier after the last site

    // ---- copy-out: st[h][d2..d5] -> C[ccur + h sh + ...]: the image is read into registers, released (barrier), and the
    // stores drain under the next item's first site
    if (FULL) {
      constexpr int NP = (D * (SB / 4) + 255) / 256;                // at most 1944 vectors: 54 e2 per h
      const int vph = 54 * (int)ext[1], NV = D * vph;
      vec4 ov[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int q = tidv + 256 * i;
        const int h = q / vph, r4 = q - h * vph;
        if (q < NV) ov[i] = *reinterpret_cast<const vec4*>(st + h * SB + 4 * r4);
      }
      rq_barrier();
      RQ_STAMP(6);  // copy-out: LDS reads + barrier
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int q = tidv + 256 * i;
        const int h = q / vph, r4 = q - h * vph;
        if (q < NV) {
          const vec4 v = ov[i];
          if (!RQ_ABL(1)) *reinterpret_cast<vec4*>(C + ccur + (int64_t)h * p.sh + 4 * r4) = v;
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
      }
    } else {
      const uint32_t R = ext[1] * ext[2] * ext[3] * ext[4], tot = R * ext[5];
      for (uint32_t q = tidv; q < tot; q += 256) {
        uint32_t h = q / R, r = q - h * R;
        uint32_t f = h * SB;
        int64_t o = ccur + (int64_t)h * p.sh;
#pragma unroll
        for (int s = 5; s >= 2; --s) {
          const uint32_t es = ext[s - 1], qq = r / es, x = r - qq * es;
          constexpr int SL[4] = {216, 36, 6, 1};
          f += x * SL[s - 2];
          o += (int64_t)x * p.sd[s - 1];
          r = qq;
        }
        const float v = st[f];
        C[o] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
      rq_barrier();
    }
    RQ_STAMP(7);    // copy-out: stores issued
    if (!more) break;
    ++kitem;
  }
  if (queue && tid == 0) {      // the last workgroup to finish re-arms the counters for the next launch on this stream
    __threadfence();
    if (atomicAdd(queue + 8, 1u) == gridDim.x - 1) {
      for (int x = 0; x < 9; ++x) queue[x] = 0;
    }
  }
#ifdef QAMD_RQ_TIMING
  tacc[9] = __builtin_amdgcn_s_memtime() - tentry;     // whole kernel, this wave
  tacc[10] = (float)(tentry & 0xffffff);               // (entry stamp, low bits: how far apart the workgroups start)
  if (absmax_out && lane == 0)
    for (int i = 0; i < 16; ++i) absmax_out[(blockIdx.x * 4 + wave) * 16 + i] = (float)tacc[i];
  return;
#endif
  if (absmax_out) {
#pragma unroll
    for (int dl = 32; dl > 0; dl >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, dl, 64));
    if (lane == 0)
      atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + ((blockIdx.x * 4 + wave) % QAMD_SLOTS), __float_as_uint(vmax));
  }
}

}  // namespace qamdq

using namespace qamdq;

#include <map>
#include <mutex>
#include <utility>

// The item queue of a large row (opt-in): nine counters (one per XCD + finished workgroups), re-armed by the launch's last workgroup.
// Launches on ONE stream are ordered, so a (device, stream) pair owns one 64-byte slot for good; a capturing stream gets
// none (the captured node could replay beside an eager launch on the same stream): the kernel then deals equal static shares.
static uint32_t* rq_queue_for(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  struct Pool { uint32_t* base; int used; };
  constexpr int kSlots = 4096, kWords = 16;
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, uint32_t*> slots;
  static std::map<int, Pool> pools;
  std::lock_guard<std::mutex> lock(mu);
  auto it = slots.find({dev, st});
  if (it != slots.end()) return it->second;
  auto pi = pools.find(dev);
  if (pi == pools.end()) {
    uint32_t* base = nullptr;
    if (hipMalloc(&base, (size_t)kSlots * kWords * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(base, 0, (size_t)kSlots * kWords * sizeof(uint32_t)) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    pi = pools.insert({dev, Pool{base, 0}}).first;
  }
  if (pi->second.used >= kSlots) return nullptr;
  uint32_t* q = pi->second.base + (size_t)kWords * pi->second.used++;
  slots[{dev, st}] = q;
  return q;
}

// One row (five sites, bonds and up legs of size 6, new legs of size a->ed[] / a->eh <= 6) in one launch.
// a->items = (number of S values) * ed[0].
extern "C" int qamd_rowq_launch(const RowArgs* a, const void* A, const void* const* W, void* C, const void* scale_a,
                                const void* const* scale_w, void* absmax_out, void* stream) {
  if (!a || a->nS < 0 || a->nS > 4 || a->items == 0) return -2;
  bool full = a->eh == D;
  for (int c = 0; c < 5; ++c) {
    if (a->ed[c] < 1 || a->ed[c] > (uint32_t)D) return -2;
    full = full && (c <= 1 || a->ed[c] == (uint32_t)D);
  }
  if (a->eh < 1 || a->eh > (uint32_t)D) return -2;
  // the vector copy-out: (d2..d5) one contiguous run in C, every run 16-byte aligned
  full = full && a->sd[4] == 1 && a->sd[3] == D && a->sd[2] == DD && a->sd[1] == DD * D && a->sd[0] % 4 == 0 &&
         a->sh % 4 == 0 && ((uintptr_t)C % 16) == 0;
  for (int g = 0; g < a->nS; ++g) full = full && a->sSc[g] % 4 == 0;
  RowPtrs w;
  for (int c = 0; c < 5; ++c) {
    w.W[c] = (const float*)W[c];
    w.scale_w[c] = scale_w ? (const float*)scale_w[c] : nullptr;
  }
  const bool acont = a->sv[4] == 1 && a->sv[3] == D && a->sv[2] == DD && a->sv[1] == DD * D;
  const size_t lds = (size_t)(D * SB + 4 * NF * 64 + D * DD + 4) * sizeof(float);   // the state image + four sites' fragments + site 1's tensor + queue words
  // persistent workgroups: three per CU (LDS), each walks a contiguous share of its XCD's items
  uint32_t grid = 8 * ((a->items + 7) / 8);
  uint32_t* queue = nullptr;
  if (grid > 768) {
    grid = 768;
    // (bit 512 = plan.kernel 3: draw the items from the per-stream queue instead of dealing equal static shares.  The queue
    // evens out the workgroups' finishing times -- the MFMA pipe serves the oldest wave first -- but the step is the same
    // 14.7 ms either way (profiles/r06_rowq_variants.txt), so the default keeps the library free of per-stream state.)
    if (a->pad2_ & 512) queue = rq_queue_for((hipStream_t)stream);
  }
#define QAMD_RQ_GO(F, AC)                                                                                          \
  QAMD_LAUNCH((rowq_kernel<F, AC>), dim3(grid), dim3(256), lds, (hipStream_t)stream, *a, w, (const float*)A, (float*)C, \
              (const float*)scale_a, (float*)absmax_out, queue)
  if (full && acont) QAMD_RQ_GO(true, true);
  else if (full) QAMD_RQ_GO(true, false);
  else if (acont) QAMD_RQ_GO(false, true);
  else QAMD_RQ_GO(false, false);
#undef QAMD_RQ_GO
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
