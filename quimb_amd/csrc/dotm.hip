// dotm.hip -- "few rows x one long vector": C[s] = sum_k R[s, k] * v[k], S <= 32 rows, K huge,
// both operands contiguous along k (gfx950 only).
//
// The reduction-shaped contractions of a Krylov solver on top of the contraction path
// (norms, Q^H w projections: M*N tiny, K = the whole state vector) are pure HBM streaming:
// S*K + K elements read, S written.  The tiled GETT handled them through 1024-way split-K
// at 0.4-0.6 ms per call; here every workgroup streams a contiguous k range with 16-byte
// loads, keeps S partial sums per lane in registers, reduces them through shuffles + LDS and
// writes ONE slab row of partial sums; the existing deterministic slab reduction
// (splitk_reduce_kernel: fixed summation order, alpha, absmax) finishes the job.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gett_args.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T, int ST>
__global__ __launch_bounds__(256) void dotm_kernel(const DotArgs p, const T* __restrict__ R, const T* __restrict__ v,
                                                   T* __restrict__ slab) {
  constexpr int V = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V), aligned(16)));
  __shared__ T red[4][ST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T acc[ST];
#pragma unroll
  for (int s = 0; s < ST; ++s) acc[s] = T(0);
  // k range of this workgroup, in vectors
  const int64_t nvec = p.K / V;
  const int64_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)blockIdx.x * per;
  int64_t v1 = v0 + per;
  if (v1 > nvec) v1 = nvec;
  for (int64_t i = v0 + tid; i < v1; i += 256) {
    const vec_t x = *reinterpret_cast<const vec_t*>(v + i * V);
    vec_t r[ST];
#pragma unroll
    for (int s = 0; s < ST; ++s)      // all row loads of this step in flight together (ST == S exactly: no predicates)
      r[s] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(R + p.row_off[s] + i * V));
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[s] += r[s][e] * x[e];
  }
  // scalar tail (K % V) handled by the last workgroup's first lane
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    for (int64_t k = nvec * V; k < p.K; ++k)
#pragma unroll
      for (int s = 0; s < ST; ++s) acc[s] += R[p.row_off[s] + k] * v[k];
  }
#pragma unroll
  for (int s = 0; s < ST; ++s) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc[s] += __shfl_down(acc[s], d, 64);
    if (lane == 0) red[wave][s] = acc[s];
  }
  __syncthreads();
  if (tid < ST) slab[(int64_t)blockIdx.x * p.S + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// C[s] = alpha * sum over the slabs, one wave per output: lane l adds slabs l, l + 64, ... in order, then a
// fixed shuffle tree -- deterministic, and 64-way parallel where the generic slab reduction is serial
template <typename T>
__global__ __launch_bounds__(64) void dotm_finish_kernel(T* __restrict__ C, const T* __restrict__ slab, int S, int nslab,
                                                         const T* __restrict__ scale_a, const T* __restrict__ scale_b,
                                                         T* __restrict__ absmax_out) {
  const int s = blockIdx.x, lane = threadIdx.x;
  T acc = T(0);
  for (int i = lane; i < nslab; i += 64) acc += slab[(int64_t)i * S + s];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  if (lane == 0) {
    T ma = T(1), mb = T(1);
    if (scale_a) { T m = T(0); for (int i = 0; i < QAMD_SLOTS; ++i) m = scale_a[i] > m ? scale_a[i] : m; if (m > T(0)) ma = m; }
    if (scale_b) { T m = T(0); for (int i = 0; i < QAMD_SLOTS; ++i) m = scale_b[i] > m ? scale_b[i] : m; if (m > T(0)) mb = m; }
    const T v = acc / (ma * mb);
    C[s] = v;
    if (absmax_out) {
      const T av = v < T(0) ? -v : v;
      if constexpr (sizeof(T) == 4)
        atomicMax(reinterpret_cast<unsigned int*>(absmax_out) + (s % QAMD_SLOTS), __float_as_uint(av));
      else
        atomicMax(reinterpret_cast<unsigned long long*>(absmax_out) + (s % QAMD_SLOTS),
                  (unsigned long long)__double_as_longlong(av));
    }
  }
}

}  // namespace qamd

using namespace qamd;

template <typename T>
static int launch_dotm_t(const DotArgs& a, const void* R, const void* v, void* slab, hipStream_t st) {
#define QAMD_DOT(STV) case STV: QAMD_LAUNCH((dotm_kernel<T, STV>), dim3(a.grid), dim3(256), 0, st, a, (const T*)R, (const T*)v, (T*)slab); break;
  switch (a.S) {   // one instantiation per row count: the row loop is fully unrolled and predicate-free
    QAMD_DOT(1) QAMD_DOT(2) QAMD_DOT(3) QAMD_DOT(4) QAMD_DOT(5) QAMD_DOT(6) QAMD_DOT(7) QAMD_DOT(8)
    QAMD_DOT(9) QAMD_DOT(10) QAMD_DOT(11) QAMD_DOT(12) QAMD_DOT(13) QAMD_DOT(14) QAMD_DOT(15) QAMD_DOT(16)
    QAMD_DOT(17) QAMD_DOT(18) QAMD_DOT(19) QAMD_DOT(20) QAMD_DOT(21) QAMD_DOT(22) QAMD_DOT(23) QAMD_DOT(24)
    QAMD_DOT(25) QAMD_DOT(26) QAMD_DOT(27) QAMD_DOT(28) QAMD_DOT(29) QAMD_DOT(30) QAMD_DOT(31) QAMD_DOT(32)
    default: return -2;
  }
#undef QAMD_DOT
  return hipGetLastError() == hipSuccess ? 0 : -4;
}

// slab: grid x S partial sums (the layout of the split-K workspace with split_k = grid); C: the S results
extern "C" int qamd_dotm_launch(int dtype, const DotArgs* a, const void* R, const void* v, void* slab, void* C,
                                const void* scale_a, const void* scale_b, void* absmax_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dtype == 0) rc = launch_dotm_t<float>(*a, R, v, slab, st);
  else if (dtype == 1) rc = launch_dotm_t<double>(*a, R, v, slab, st);
  else return -2;
  if (rc) return rc;
  if (dtype == 0)
    QAMD_LAUNCH(dotm_finish_kernel<float>, dim3(a->S), dim3(64), 0, st, (float*)C, (const float*)slab, a->S, (int)a->grid,
                (const float*)scale_a, (const float*)scale_b, (float*)absmax_out);
  else
    QAMD_LAUNCH(dotm_finish_kernel<double>, dim3(a->S), dim3(64), 0, st, (double*)C, (const double*)slab, a->S,
                (int)a->grid, (const double*)scale_a, (const double*)scale_b, (double*)absmax_out);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
