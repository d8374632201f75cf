// krylov.hip -- the vector work of one Lanczos step, device-resident (gfx950 only).
//
// The caller of the contraction path in DMRG's local solve (quimb/tensor/tn1d/dmrg.py:626-645 ->
// quimb/linalg/base_linalg.py:80, ARPACK on host vectors in the reference) runs, per matvec, a projection
// h = Q^H w, the update w -= Q^T h, a norm and the scaling that makes w the next basis row.  Composed from
// the general contraction entry points that was ten Python-level operations and two host reads per step --
// 0.29 ms of overhead around a 0.26 ms matvec at chi = 512.  Here it is three launches and no host read:
//
//   qamd_krylov_project : h[i] = <Q_i, w>, i < rows; h_sum (+)= h  (Q read once, partial sums in a FIXED order)
//   qamd_krylov_subtract: w -= sum_i h[i] Q_i, ||w||^2 partials (Q read once, w updated in place)
//   qamd_krylov_extend  : beta = ||w||, q_next = w / beta (zero on breakdown), (alpha, beta) -> device table
//
// All three are HBM streaming (16-byte loads, rows x n + n elements per pass); every reduction is two-stage with
// a fixed summation order, so a solve is reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "program.h"

#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define QAMD_CHECK_LAUNCH() return (hipGetLastError() == hipSuccess ? 0 : -4)

namespace qamd {
namespace krylov {

constexpr int MAX_BLOCKS = 1024;     // partial sums per reduction
constexpr int RG = 8;                // basis rows per workgroup of the projection
constexpr int SUB_ROWS = 64;         // basis rows per launch of the update (coefficients in LDS)

template <typename R> struct Cx { R re, im; };

__device__ __forceinline__ float zero_of(float*) { return 0.f; }
__device__ __forceinline__ double zero_of(double*) { return 0.0; }
template <typename R> __device__ __forceinline__ Cx<R> zero_of(Cx<R>*) { return Cx<R>{R(0), R(0)}; }

__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ double add(double a, double b) { return a + b; }
template <typename R> __device__ __forceinline__ Cx<R> add(Cx<R> a, Cx<R> b) { return Cx<R>{a.re + b.re, a.im + b.im}; }

// acc + conj(q) * w
__device__ __forceinline__ float fma_conj(float q, float w, float acc) { return fmaf(q, w, acc); }
__device__ __forceinline__ double fma_conj(double q, double w, double acc) { return fma(q, w, acc); }
template <typename R> __device__ __forceinline__ Cx<R> fma_conj(Cx<R> q, Cx<R> w, Cx<R> acc) {
  return Cx<R>{acc.re + q.re * w.re + q.im * w.im, acc.im + q.re * w.im - q.im * w.re};
}
// acc - h * q
__device__ __forceinline__ float sub_mul(float acc, float h, float q) { return fmaf(-h, q, acc); }
__device__ __forceinline__ double sub_mul(double acc, double h, double q) { return fma(-h, q, acc); }
template <typename R> __device__ __forceinline__ Cx<R> sub_mul(Cx<R> acc, Cx<R> h, Cx<R> q) {
  return Cx<R>{acc.re - (h.re * q.re - h.im * q.im), acc.im - (h.re * q.im + h.im * q.re)};
}
__device__ __forceinline__ double abs2(float x) { return (double)x * (double)x; }
__device__ __forceinline__ double abs2(double x) { return x * x; }
template <typename R> __device__ __forceinline__ double abs2(Cx<R> x) {
  return (double)x.re * (double)x.re + (double)x.im * (double)x.im;
}
__device__ __forceinline__ float scaled(float x, double s) { return (float)(x * s); }
__device__ __forceinline__ double scaled(double x, double s) { return x * s; }
template <typename R> __device__ __forceinline__ Cx<R> scaled(Cx<R> x, double s) {
  return Cx<R>{(R)(x.re * s), (R)(x.im * s)};
}
__device__ __forceinline__ double real_of(float x) { return x; }
__device__ __forceinline__ double real_of(double x) { return x; }
template <typename R> __device__ __forceinline__ double real_of(Cx<R> x) { return x.re; }

// wavefront shuffle of any 4-byte-multiple type
template <typename T> __device__ __forceinline__ T shfl_down_any(T v, int d) {
  constexpr int W = sizeof(T) / 4;
  union { T t; int w[W]; } in, out;
  in.t = v;
#pragma unroll
  for (int i = 0; i < W; ++i) out.w[i] = __shfl_down(in.w[i], d, 64);
  return out.t;
}

template <typename T, int V> struct alignas(sizeof(T) * V) Pack { T v[V]; };

// ---- h partials: one workgroup = RG rows x a strided set of 256*V-element pieces ---------------------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void project_kernel(T* __restrict__ partial, const T* __restrict__ Q, int64_t ldq,
                                                      int rows, const T* __restrict__ w, int64_t n, int nblk) {
  __shared__ T red[4][RG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.y * RG;
  const int nr = rows - r0 < RG ? rows - r0 : RG;
  T acc[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = zero_of((T*)nullptr);
  const int64_t npack = n / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < npack; i += (int64_t)nblk * 256) {
    const Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(w + i * V);
    Pack<T, V> q[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r)
      if (r < nr) q[r] = *reinterpret_cast<const Pack<T, V>*>(Q + (int64_t)(r0 + r) * ldq + i * V);
#pragma unroll
    for (int r = 0; r < RG; ++r)
      if (r < nr) {
#pragma unroll
        for (int e = 0; e < V; ++e) acc[r] = fma_conj(q[r].v[e], x.v[e], acc[r]);
      }
  }
  if (blockIdx.x == 0 && tid == 0)            // n % V elements at the end
    for (int64_t k = npack * V; k < n; ++k)
      for (int r = 0; r < nr; ++r) acc[r] = fma_conj(Q[(int64_t)(r0 + r) * ldq + k], w[k], acc[r]);
#pragma unroll
  for (int r = 0; r < RG; ++r) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc[r] = add(acc[r], shfl_down_any(acc[r], d));
    if (lane == 0) red[wave][r] = acc[r];
  }
  __syncthreads();
  if (tid < nr)
    partial[(int64_t)(r0 + tid) * nblk + blockIdx.x] = add(add(red[0][tid], red[1][tid]), add(red[2][tid], red[3][tid]));
}

// one workgroup per row: the nblk partial sums in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void project_finish_kernel(T* __restrict__ h, T* __restrict__ hsum,
                                                             const T* __restrict__ partial, int nblk, int accumulate) {
  __shared__ T red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T s = zero_of((T*)nullptr);
  for (int i = tid; i < nblk; i += 256) s = add(s, partial[(int64_t)blockIdx.x * nblk + i]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s = add(s, shfl_down_any(s, d));
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    s = add(add(red[0], red[1]), add(red[2], red[3]));
    h[blockIdx.x] = s;
    if (hsum) hsum[blockIdx.x] = accumulate ? add(hsum[blockIdx.x], s) : s;
  }
}

// ---- w -= sum_i h[i] Q_i (rows <= SUB_ROWS per launch), optionally the partial sums of ||w||^2 ---------------------
template <typename T, int V>
__global__ __launch_bounds__(256) void subtract_kernel(T* __restrict__ w, const T* __restrict__ Q, int64_t ldq, int rows,
                                                       const T* __restrict__ h, int64_t n, double* __restrict__ npart,
                                                       int nblk) {
  __shared__ T hs[SUB_ROWS];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < rows) hs[tid] = h[tid];
  __syncthreads();
  double nrm = 0.0;
  const int64_t npack = n / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < npack; i += (int64_t)nblk * 256) {
    Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(w + i * V);
    int r = 0;
    for (; r + 4 <= rows; r += 4) {           // four row loads in flight
      Pack<T, V> q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const Pack<T, V>*>(Q + (int64_t)(r + u) * ldq + i * V);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) x.v[e] = sub_mul(x.v[e], hs[r + u], q[u].v[e]);
    }
    for (; r < rows; ++r) {
      const Pack<T, V> q = *reinterpret_cast<const Pack<T, V>*>(Q + (int64_t)r * ldq + i * V);
#pragma unroll
      for (int e = 0; e < V; ++e) x.v[e] = sub_mul(x.v[e], hs[r], q.v[e]);
    }
    *reinterpret_cast<Pack<T, V>*>(w + i * V) = x;
#pragma unroll
    for (int e = 0; e < V; ++e) nrm += abs2(x.v[e]);
  }
  if (blockIdx.x == 0 && tid == 0)
    for (int64_t k = npack * V; k < n; ++k) {
      T x = w[k];
      for (int r = 0; r < rows; ++r) x = sub_mul(x, hs[r], Q[(int64_t)r * ldq + k]);
      w[k] = x;
      nrm += abs2(x);
    }
  if (npart == nullptr) return;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) nrm += __shfl_down(nrm, d, 64);
  if (lane == 0) red[wave] = nrm;
  __syncthreads();
  if (tid == 0) npart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- beta = sqrt(sum of the partials); q_next = w / beta, or 0 when the Krylov space is exhausted -----------------
template <typename T, int V>
__global__ __launch_bounds__(256) void extend_kernel(T* __restrict__ qn, const T* __restrict__ w, int64_t n,
                                                     const double* __restrict__ npart, int nparts,
                                                     const T* __restrict__ hj, double* __restrict__ ab, double tiny,
                                                     int nblk) {
  __shared__ double red[4];
  __shared__ double total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int i = tid; i < nparts; i += 256) s += npart[i];      // the same order in every workgroup
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) total = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  const double beta = sqrt(total);
  const double alpha = real_of(hj[0]);
  const double scale = beta > tiny * fmax(fabs(alpha), 1.0) ? 1.0 / beta : 0.0;
  if (blockIdx.x == 0 && tid == 0) {
    ab[0] = alpha;
    ab[1] = beta;
  }
  const int64_t npack = n / V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < npack; i += (int64_t)nblk * 256) {
    Pack<T, V> x = *reinterpret_cast<const Pack<T, V>*>(w + i * V);
#pragma unroll
    for (int e = 0; e < V; ++e) x.v[e] = scaled(x.v[e], scale);
    *reinterpret_cast<Pack<T, V>*>(qn + i * V) = x;
  }
  if (blockIdx.x == 0 && tid == 0)
    for (int64_t k = npack * V; k < n; ++k) qn[k] = scaled(w[k], scale);
}

static inline int blocks_for(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}
static inline int64_t esize(int dtype) { return dtype == 0 ? 4 : (dtype == 3 ? 16 : 8); }
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// workspace: [MAX_BLOCKS doubles of norm partials][rows x MAX_BLOCKS elements of projection partials]
static inline void* proj_partials(void* ws) { return (char*)ws + MAX_BLOCKS * sizeof(double); }

}  // namespace krylov
}  // namespace qamd

using namespace qamd::krylov;

extern "C" int64_t qamd_krylov_workspace_bytes(int32_t rows, int64_t n, int32_t dtype) {
  (void)n;
  if (rows < 0 || dtype < 0 || dtype > 3) return -1;
  return (int64_t)MAX_BLOCKS * sizeof(double) + (int64_t)(rows > 0 ? rows : 1) * MAX_BLOCKS * esize(dtype);
}

// vector width: 16-byte packs when every row start and w are 16-byte aligned, single elements otherwise
#define QAMD_KRYLOV_DISPATCH(WIDE, CALL)                                   \
  switch (dtype) {                                                         \
    case 0: if (WIDE) { CALL(float, 4); } else { CALL(float, 1); } break;  \
    case 1: if (WIDE) { CALL(double, 2); } else { CALL(double, 1); } break;\
    case 2: if (WIDE) { CALL(Cx<float>, 2); } else { CALL(Cx<float>, 1); } break; \
    case 3: CALL(Cx<double>, 1); break;                                    \
    default: return -2;                                                    \
  }

extern "C" int qamd_krylov_project(void* h_dev, void* h_sum_dev, const void* Q, int64_t ldq, int32_t rows, const void* w,
                                   int64_t n, int32_t accumulate, int32_t dtype, void* ws_dev, void* stream) {
  if (qamdp_recording()) return -9;      // a solver step is not part of a contraction's launch program
  if (rows <= 0 || n <= 0) return rows < 0 || n < 0 ? -1 : 0;
  if (!h_dev || !Q || !w || !ws_dev || ldq < n) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = blocks_for(n);
  const bool wide = aligned16(Q) && aligned16(w) && (ldq * esize(dtype)) % 16 == 0;
  void* part = proj_partials(ws_dev);
  dim3 grid(nblk, (rows + RG - 1) / RG);
#define CALL(T, V) QAMD_LAUNCH((project_kernel<T, V>), grid, dim3(256), 0, st, (T*)part, (const T*)Q, ldq, (int)rows, \
                               (const T*)w, n, nblk)
  QAMD_KRYLOV_DISPATCH(wide, CALL)
#undef CALL
  if (hipGetLastError() != hipSuccess) return -4;
  switch (dtype) {
    case 0: QAMD_LAUNCH(project_finish_kernel<float>, dim3(rows), dim3(256), 0, st, (float*)h_dev, (float*)h_sum_dev, (const float*)part, nblk, (int)accumulate); break;
    case 1: QAMD_LAUNCH(project_finish_kernel<double>, dim3(rows), dim3(256), 0, st, (double*)h_dev, (double*)h_sum_dev, (const double*)part, nblk, (int)accumulate); break;
    case 2: QAMD_LAUNCH(project_finish_kernel<Cx<float>>, dim3(rows), dim3(256), 0, st, (Cx<float>*)h_dev, (Cx<float>*)h_sum_dev, (const Cx<float>*)part, nblk, (int)accumulate); break;
    default: QAMD_LAUNCH(project_finish_kernel<Cx<double>>, dim3(rows), dim3(256), 0, st, (Cx<double>*)h_dev, (Cx<double>*)h_sum_dev, (const Cx<double>*)part, nblk, (int)accumulate); break;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_krylov_subtract(void* w, const void* Q, int64_t ldq, int32_t rows, const void* h_dev, int64_t n,
                                    int32_t want_norm, int32_t dtype, void* ws_dev, void* stream) {
  if (qamdp_recording()) return -9;
  if (rows < 0 || n < 0) return -1;
  if (n == 0) return 0;
  if (!w || !ws_dev || (rows > 0 && (!Q || !h_dev)) || ldq < n) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = blocks_for(n);
  const bool wide = aligned16(Q) && aligned16(w) && (ldq * esize(dtype)) % 16 == 0;
  const int64_t es = esize(dtype);
  int r0 = 0;
  do {                                       // SUB_ROWS rows per launch; the norm rides on the last one
    const int nr = rows - r0 < SUB_ROWS ? rows - r0 : SUB_ROWS;
    const bool last = r0 + nr >= rows;
    double* np = (want_norm && last) ? (double*)ws_dev : nullptr;
    const char* Qr = (const char*)Q + (int64_t)r0 * ldq * es;
    const char* hr = (const char*)h_dev + (int64_t)r0 * es;
#define CALL(T, V) QAMD_LAUNCH((subtract_kernel<T, V>), dim3(nblk), dim3(256), 0, st, (T*)w, (const T*)Qr, ldq, nr, \
                               (const T*)hr, n, np, nblk)
    QAMD_KRYLOV_DISPATCH(wide, CALL)
#undef CALL
    if (hipGetLastError() != hipSuccess) return -4;
    r0 += nr;
  } while (r0 < rows);
  return 0;
}

extern "C" int qamd_krylov_extend(void* q_next, const void* w, int64_t n, const void* h_j_dev, double* alpha_beta_dev,
                                  double breakdown_eps, int32_t dtype, const void* ws_dev, void* stream) {
  if (qamdp_recording()) return -9;
  if (n <= 0 || !q_next || !w || !h_j_dev || !alpha_beta_dev || !ws_dev) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = blocks_for(n);
  const bool wide = aligned16(q_next) && aligned16(w);
#define CALL(T, V) QAMD_LAUNCH((extend_kernel<T, V>), dim3(nblk), dim3(256), 0, st, (T*)q_next, (const T*)w, n, \
                               (const double*)ws_dev, nblk, (const T*)h_j_dev, alpha_beta_dev, breakdown_eps, nblk)
  QAMD_KRYLOV_DISPATCH(wide, CALL)
#undef CALL
  QAMD_CHECK_LAUNCH();
}
