// microtree.hip -- a whole contraction tree of SMALL tensors executed by one workgroup per instance
// (gfx950 only).
//
// A circuit amplitude (BASELINE config #2: 53 qubits, depth 10) is ~900 pairwise steps on tensors of
// at most 2^9 elements: dispatch-bound -- 23 ms launched step by step, 5.9 ms as a hipGraph (6.5 us
// per node), against a few hundred FLOPs per step.  Here the device walks the tree itself: the plan
// (one qamd_micro_step per pairwise contraction: operand locations, bundle dims and strides) sits in
// device memory, one 256-thread workgroup executes step after step with a workgroup barrier in
// between, intermediates live in a scratch arena that stays in L2, and a GRID of workgroups runs
// independent instances of the same tree (different input tensors: e.g. one bitstring each) side by
// side.  Every step is the same mathematics as gett.hip (C[b,m,n] = sum_k A[b,m,k] B[b,k,n] with
// tensor addressing), evaluated with plain FMAs: at these sizes there is nothing for the matrix
// cores to do.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "micro_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename R, bool CPLX> struct MElem;
template <typename R> struct MElem<R, false> {
  typedef R type;
  static __device__ __forceinline__ R zero() { return R(0); }
  static __device__ __forceinline__ void fma(R& acc, R a, R b) { acc += a * b; }
};
template <typename R> struct MElem<R, true> {
  struct type { R re, im; };
  static __device__ __forceinline__ type zero() { return type{R(0), R(0)}; }
  static __device__ __forceinline__ void fma(type& acc, type a, type b) {
    acc.re += a.re * b.re - a.im * b.im;
    acc.im += a.re * b.im + a.im * b.re;
  }
};

// LDSARENA: the intermediates live in LDS (the host's lifetime-aware allocation fits): a step then depends
// on its predecessor through ~100 ns of LDS instead of a write-to-L2 / read-back round trip.
//
// The plan is fully lowered on the host (qamd_micro_step + etab + ktab): no index arithmetic on the device.
// Software pipeline over steps: while step i computes, the header of step i+1 (wave-uniform -> scalar
// loads), its first KREG k-offset pairs (scalar) and this thread's first address triple are already on
// their way, so the dependent chain per step is  barrier -> operand reads -> FMAs -> result write.
//
// WIDE (fp32 / complex64 trees): the INTERMEDIATES are carried in double precision -- inputs are read as they are, every
// step accumulates in fp64 and stores fp64 into the arena, the root is rounded once on its way out.  A circuit amplitude is
// ~900 chained steps cancelling down to |a| ~ 2^-26: rounded to fp32 after every step they cost 0.7-2.1e-6 relative
// (numpy's own complex64 evaluation of the same tree: 1.4e-6), more than north_star's 1e-6.  Measured on config #2: 3e-8
// instead of 1.75e-6, at 1.35 us per dependent step instead of 0.96 (16-byte LDS elements, fp64 FMA chains): 1.2 ms per
// amplitude instead of 0.86.  Parity comes first; Options.micro_wide = False is the per-step-rounding walk.
template <typename R, bool CPLX, bool LDSARENA, bool WIDE>
__global__ __launch_bounds__(256) void microtree_kernel(const qamd_micro_step* __restrict__ steps, int nsteps,
                                                        const int32_t* __restrict__ etab,
                                                        const int32_t* __restrict__ ktab,
                                                        const void* const* __restrict__ inputs, int ninputs,
                                                        void* __restrict__ arena, int64_t arena_elems,
                                                        void* __restrict__ out, int64_t out_elems) {
  typedef typename MElem<R, CPLX>::type E;                         // element of the inputs and of the result
  typedef typename std::conditional<WIDE, double, R>::type RW;
  typedef typename MElem<RW, CPLX>::type W;                        // element of the intermediates and of the accumulator
  constexpr int KREG = 8;
  extern __shared__ __attribute__((aligned(16))) char mt_lds[];
  const int tid = threadIdx.x;
  const void* const* my_in = inputs + (int64_t)blockIdx.x * ninputs;
  W* my_arena = LDSARENA ? reinterpret_cast<W*>(mt_lds) : reinterpret_cast<W*>(arena) + (int64_t)blockIdx.x * arena_elems;
  E* my_out = reinterpret_cast<E*>(out) + (int64_t)blockIdx.x * out_elems;
  auto widen = [](const E& x) -> W {
    if constexpr (CPLX) return W{(RW)x.re, (RW)x.im};
    else return (RW)x;
  };
  auto narrow = [](const W& x) -> E {
    if constexpr (CPLX) return E{(R)x.re, (R)x.im};
    else return (R)x;
  };

  struct Pre {               // everything of a step that can be fetched before its operands exist
    qamd_micro_step h;
    int32_t ka[KREG], kb[KREG];
    int32_t t0, t1, t2;      // this thread's first address triple
  };
  auto fetch = [&](int si, Pre& p) {
    p.h = steps[si];                                  // wave-uniform address: scalar loads
#pragma unroll
    for (int k = 0; k < KREG; ++k) {
      const bool ok = (uint32_t)k < p.h.K;
      p.ka[k] = ok ? ktab[2 * (p.h.koff + k)] : 0;
      p.kb[k] = ok ? ktab[2 * (p.h.koff + k) + 1] : 0;
    }
    const bool mine = (uint32_t)tid < p.h.total;
    const int32_t* t = etab + 3 * (int64_t)(p.h.eoff + (mine ? tid : 0));
    p.t0 = t[0]; p.t1 = t[1]; p.t2 = t[2];
  };

  Pre cur, nxt;
  fetch(0, cur);
  for (int si = 0; si < nsteps; ++si) {
    if (si + 1 < nsteps) fetch(si + 1, nxt);
    const qamd_micro_step& s = cur.h;
    // operands: an input tensor (E) or an intermediate in the arena (W) -- wave-uniform per step
    const E* Ai = s.a_kind ? nullptr : reinterpret_cast<const E*>(my_in[s.a_ref]);
    const E* Bi = s.b_kind ? nullptr : reinterpret_cast<const E*>(my_in[s.b_ref]);
    const W* Aw = my_arena + (s.a_kind ? s.a_ref : 0);
    const W* Bw = my_arena + (s.b_kind ? s.b_ref : 0);
    auto lda = [&](int32_t o) -> W { return s.a_kind ? Aw[o] : widen(Ai[o]); };
    auto ldb = [&](int32_t o) -> W { return s.b_kind ? Bw[o] : widen(Bi[o]); };
    for (uint32_t e = tid; e < s.total; e += 256) {
      int32_t oa = cur.t0, ob = cur.t1, oc = cur.t2;
      if (e != (uint32_t)tid) {
        const int32_t* t = etab + 3 * (int64_t)(s.eoff + e);
        oa = t[0]; ob = t[1]; oc = t[2];
      }
      W acc = MElem<RW, CPLX>::zero();
      // (prefetching the VALUES of input-kind operands one step ahead was measured slower: it lengthens the
      //  serial load chain of the prefetch itself, which then bounds the step time)
#pragma unroll
      for (int k = 0; k < KREG; ++k)
        if ((uint32_t)k < s.K) MElem<RW, CPLX>::fma(acc, lda(oa + cur.ka[k]), ldb(ob + cur.kb[k]));
      for (uint32_t k = KREG; k < s.K; ++k)
        MElem<RW, CPLX>::fma(acc, lda(oa + ktab[2 * (s.koff + k)]), ldb(ob + ktab[2 * (s.koff + k) + 1]));
      if (s.c_off >= 0) my_arena[s.c_off + oc] = acc;
      else my_out[oc] = narrow(acc);
    }
    if (!LDSARENA) __threadfence_block();
    __syncthreads();
    cur = nxt;
  }
}

}  // namespace qamd

using namespace qamd;

template <typename R, bool CP, bool WIDE>
static int launch_microtree(bool lds, size_t lds_bytes, const qamd_micro_step* steps_dev, int nsteps,
                            const int32_t* etab, const int32_t* ktab, const void* const* inputs_dev, int ninputs, void* arena_dev, int64_t arena_elems,
                            void* out_dev, int64_t out_elems, int ninst, hipStream_t st) {
  if (lds) {
    (void)hipFuncSetAttribute((const void*)microtree_kernel<R, CP, true, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    QAMD_LAUNCH((microtree_kernel<R, CP, true, WIDE>), dim3(ninst), dim3(256), lds_bytes, st, steps_dev, nsteps, etab, ktab,
                inputs_dev, ninputs, arena_dev, arena_elems, out_dev, out_elems);
  } else {
    QAMD_LAUNCH((microtree_kernel<R, CP, false, WIDE>), dim3(ninst), dim3(256), 0, st, steps_dev, nsteps, etab, ktab,
                inputs_dev, ninputs, arena_dev, arena_elems, out_dev, out_elems);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// arena_dev == NULL: the arena (arena_elems elements per instance) is carved out of LDS.  wide (F32 / C64 only): the
// arena's elements are the double-precision counterparts (8 / 16 bytes).
extern "C" int qamd_microtree_launch(int dtype, const qamd_micro_step* steps_dev, int nsteps, const int32_t* etab,
                                     const int32_t* ktab, const void* const* inputs_dev, int ninputs, void* arena_dev,
                                     int64_t arena_elems, void* out_dev, int64_t out_elems, int ninst, int wide,
                                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nsteps <= 0 || ninst <= 0) return -1;
  static const size_t esz[4] = {4, 8, 8, 16};
  if (dtype < 0 || dtype > 3) return -2;
  if (dtype == 1 || dtype == 3) wide = 0;       // already double precision
  const bool lds = arena_dev == nullptr && arena_elems > 0;
  const size_t lds_bytes = lds ? (size_t)arena_elems * esz[dtype] * (wide ? 2 : 1) : 0;
  if (lds_bytes > QAMD_MICRO_LDS_ARENA_BYTES) return -2;
#define QAMD_MT_ARGS lds, lds_bytes, steps_dev, nsteps, etab, ktab, inputs_dev, ninputs, arena_dev, arena_elems, out_dev, out_elems, ninst, st
  switch (dtype) {
    case 0: return wide ? launch_microtree<float, false, true>(QAMD_MT_ARGS) : launch_microtree<float, false, false>(QAMD_MT_ARGS);
    case 1: return launch_microtree<double, false, false>(QAMD_MT_ARGS);
    case 2: return wide ? launch_microtree<float, true, true>(QAMD_MT_ARGS) : launch_microtree<float, true, false>(QAMD_MT_ARGS);
    default: return launch_microtree<double, true, false>(QAMD_MT_ARGS);
  }
#undef QAMD_MT_ARGS
}
