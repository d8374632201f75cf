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

__device__ __forceinline__ void mdecomp3(uint32_t idx, int n, const uint32_t* dims, const int32_t* s0,
                                         const int32_t* s1, const int32_t* s2, int64_t& o0, int64_t& o1, int64_t& o2) {
  for (int g = n - 1; g >= 0; --g) {
    const uint32_t d = dims[g], q = idx / d, r = idx - q * d;
    if (s0) o0 += (int64_t)r * s0[g];
    if (s1) o1 += (int64_t)r * s1[g];
    if (s2) o2 += (int64_t)r * s2[g];
    idx = q;
  }
}

template <typename R, bool CPLX>
__global__ __launch_bounds__(256) void microtree_kernel(const qamd_micro_step* __restrict__ steps, int nsteps,
                                                        const void* const* __restrict__ inputs, int ninputs,
                                                        void* __restrict__ arena, int64_t arena_elems,
                                                        void* __restrict__ out, int64_t out_elems) {
  typedef typename MElem<R, CPLX>::type E;
  __shared__ int32_t koffA[QAMD_MICRO_KMAX], koffB[QAMD_MICRO_KMAX];
  const int tid = threadIdx.x;
  const void* const* my_in = inputs + (int64_t)blockIdx.x * ninputs;
  E* my_arena = reinterpret_cast<E*>(arena) + (int64_t)blockIdx.x * arena_elems;
  E* my_out = reinterpret_cast<E*>(out) + (int64_t)blockIdx.x * out_elems;

  for (int si = 0; si < nsteps; ++si) {
    const qamd_micro_step& s = steps[si];
    const E* A = s.a_kind ? my_arena + s.a_ref : reinterpret_cast<const E*>(my_in[s.a_ref]);
    const E* B = s.b_kind ? my_arena + s.b_ref : reinterpret_cast<const E*>(my_in[s.b_ref]);
    E* C = s.c_off >= 0 ? my_arena + s.c_off : my_out;
    // k offsets of both operands, once per step
    for (uint32_t k = tid; k < s.K; k += 256) {
      int64_t oa = 0, ob = 0, dummy = 0;
      mdecomp3(k, s.nk, s.dim_k, s.sa_k, s.sb_k, nullptr, oa, ob, dummy);
      koffA[k] = (int32_t)oa;
      koffB[k] = (int32_t)ob;
    }
    __syncthreads();
    const uint32_t MN = s.M * s.N, total = s.B * MN;
    for (uint32_t e = tid; e < total; e += 256) {
      const uint32_t b = e / MN, r = e - b * MN, m = r / s.N, n = r - m * s.N;
      int64_t oa = 0, ob = 0, oc = 0;
      mdecomp3(b, s.nb, s.dim_b, s.sa_b, s.sb_b, s.sc_b, oa, ob, oc);
      int64_t dummy = 0;
      mdecomp3(m, s.nm, s.dim_m, s.sa_m, nullptr, s.sc_m, oa, dummy, oc);
      mdecomp3(n, s.nn, s.dim_n, nullptr, s.sb_n, s.sc_n, dummy, ob, oc);
      E acc = MElem<R, CPLX>::zero();
      for (uint32_t k = 0; k < s.K; ++k) MElem<R, CPLX>::fma(acc, A[oa + koffA[k]], B[ob + koffB[k]]);
      C[oc] = acc;
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace qamd

using namespace qamd;

extern "C" int qamd_microtree_launch(int dtype, const qamd_micro_step* steps_dev, int nsteps,
                                     const void* const* inputs_dev, int ninputs, void* arena_dev,
                                     int64_t arena_elems, void* out_dev, int64_t out_elems, int ninst, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nsteps <= 0 || ninst <= 0) return -1;
#define QAMD_MT(R, CP) QAMD_LAUNCH((microtree_kernel<R, CP>), dim3(ninst), dim3(256), 0, st, steps_dev, nsteps, inputs_dev, \
                                   ninputs, arena_dev, arena_elems, out_dev, out_elems)
  switch (dtype) {
    case 0: QAMD_MT(float, false); break;
    case 1: QAMD_MT(double, false); break;
    case 2: QAMD_MT(float, true); break;
    case 3: QAMD_MT(double, true); break;
    default: return -2;
  }
#undef QAMD_MT
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
