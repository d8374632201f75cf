// elementwise.hip -- HBM-bound layout / elementwise kernels for gfx950.
//
//  * tiled N-d permute (index fusion, transpose, isel/take views, diagonals):
//    replaces numpy transpose+reshape copies behind `fuse`
//    (quimb/tensor/array_ops.py:148-182), Tensor.transpose (tensor_core.py:2743)
//    and Tensor.isel (tensor_core.py:2260-2348).
//  * strided sum-reduction, strided binary op, scale/axpby/conj/cast/fill,
//    absmax + exponent stripping (tensor_core.py:330-340 semantics).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "ew_args.h"
#include "program.h"

// hipGetLastError() also reports benign stale codes (hipErrorNotReady from an
// event query by the allocator, ...): clear them before each launch so the
// post-launch check only sees this launch.
#define QAMD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace qamd {

template <typename T> struct CT;  // complex helpers
struct c64 { float re, im; };
struct c128 { double re, im; };

__device__ __forceinline__ int64_t decomp1(uint32_t idx, int n, const uint32_t* dims,
                                           const int64_t* strides) {
  int64_t off = 0;
  for (int g = n - 1; g >= 0; --g) {
    uint32_t d = dims[g];
    uint32_t q = idx / d, r = idx - q * d;
    off += (int64_t)r * strides[g];
    idx = q;
  }
  return off;
}

// ---------------------------------------------------------------------------
// Tiled permute.  Dims are split into three bundles: X (fast in src),
// Y (fast in dst), Z (the rest).  A workgroup moves a TX x TY tile: lanes run
// along x while reading (coalesced in src), along y while writing (coalesced in
// dst), with the transpose done through a padded LDS tile.  If `direct`, X is
// fast in both and the LDS stage is skipped.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void permute_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                       const PermArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int64_t* xs = reinterpret_cast<int64_t*>(smem);  // src offs of x
  int64_t* xd = xs + p.TX;
  int64_t* ys = xd + p.TX;
  int64_t* yd = ys + p.TY;
  T* tile = reinterpret_cast<T*>(yd + p.TY);
  const int pitch = p.TX + 1;
  const int tid = threadIdx.x;

  uint32_t id = blockIdx.x;
  const uint32_t tx = id % p.tiles_x; id /= p.tiles_x;
  const uint32_t ty = id % p.tiles_y; id /= p.tiles_y;
  const uint32_t z = id;

  int64_t zs = p.src_offset, zd = 0;
  {
    uint32_t idx = z;
    for (int g = p.nz - 1; g >= 0; --g) {
      uint32_t d = p.dim_z[g];
      uint32_t q = idx / d, r = idx - q * d;
      zs += (int64_t)r * p.ss_z[g];
      zd += (int64_t)r * p.sd_z[g];
      idx = q;
    }
  }
  // out-of-tile marker: NOT -1 -- source offsets are legitimately negative for reversed (negative-stride) views
  constexpr int64_t kNone = INT64_MIN;
  for (int i = tid; i < p.TX + p.TY; i += 256) {
    if (i < p.TX) {
      uint32_t x = tx * p.TX + i;
      bool ok = x < p.X;
      xs[i] = ok ? decomp1(x, p.nx, p.dim_x, p.ss_x) : kNone;
      xd[i] = ok ? decomp1(x, p.nx, p.dim_x, p.sd_x) : kNone;
    } else {
      int j = i - p.TX;
      uint32_t y = ty * p.TY + j;
      bool ok = y < p.Y;
      ys[j] = ok ? decomp1(y, p.ny, p.dim_y, p.ss_y) : kNone;
      yd[j] = ok ? decomp1(y, p.ny, p.dim_y, p.sd_y) : kNone;
    }
  }
  __syncthreads();

  const int total = p.TX * p.TY;
  if (p.direct) {
    for (int e = tid; e < total; e += 256) {
      int x = e % p.TX, y = e / p.TX;
      int64_t a = xs[x], b = ys[y];
      if (a != kNone && b != kNone) dst[zd + xd[x] + yd[y]] = src[zs + a + b];
    }
    return;
  }
  for (int e = tid; e < total; e += 256) {
    int x = e % p.TX, y = e / p.TX;
    int64_t a = xs[x], b = ys[y];
    if (a != kNone && b != kNone) tile[y * pitch + x] = src[zs + a + b];
  }
  __syncthreads();
  for (int e = tid; e < total; e += 256) {
    int y = e % p.TY, x = e / p.TY;
    int64_t a = xd[x], b = yd[y];
    if (a != kNone && b != kNone) dst[zd + a + b] = tile[y * pitch + x];
  }
}

// ---------------------------------------------------------------------------
// Streaming permute.  The TILE is a set of (parts of) dims -- bundle X, listed in source order -- chosen by
// the host so that it holds a long contiguous run on the source side AND on the destination side
// (plan_permute_stream, api.cpp); everything else is bundle Z, and a workgroup walks a whole chunk of z
// values with the same tile geometry.  Per-element address work is done ONCE per workgroup: every thread
// caches, in registers, the 32-bit source offset / destination offset / LDS slots of its <= 16 elements
// (element e of the tile in SOURCE order when reading, in DESTINATION order when writing: both sides see
// whole contiguous runs whatever the permutation inside the tile is).  Per z a thread issues its loads back
// to back -- the NEXT z's loads go out before this z's tile is written, so up to 2 x 16 KB of reads are in
// flight per workgroup -- scatters them into a skewed LDS tile and writes them out.  ``direct``: the two
// orders coincide inside the tile, no LDS stage.
// ---------------------------------------------------------------------------
template <typename T, int EPT>
__global__ __launch_bounds__(256) void permute_stream_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                              const PermArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem_s[];
  T* tile = reinterpret_cast<T*>(smem_s);
  const int tid = threadIdx.x;
  const uint32_t z0 = blockIdx.x * p.zchunk;
  const uint32_t z1 = min(p.Z, z0 + p.zchunk);
  if (z0 >= z1) return;
  constexpr int32_t kNone = INT32_MIN;
  const int total = (int)p.X;
  // LDS rows = the innermost source group, padded to an odd pitch: lanes that walk ACROSS rows when writing
  // out (any destination order) then hit distinct banks
  const uint32_t inner = p.dim_x[p.nx - 1];
  const uint32_t skew = (inner & 1) ? 0x7fffffffu : inner;

  int32_t so[EPT], dofs[EPT];
  uint32_t slots[EPT];                               // (LDS slot written) | (LDS slot read) << 16
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = tid + 256 * i;
    so[i] = kNone;
    dofs[i] = kNone;
    slots[i] = 0;
    if (e < total) {
      so[i] = (int32_t)decomp1((uint32_t)e, p.nx, p.dim_x, p.ss_x);
      if (p.direct) {
        dofs[i] = (int32_t)decomp1((uint32_t)e, p.nx, p.dim_x, p.sd_x);
      } else {
        // e enumerates the tile in DESTINATION order: digits over the groups sorted by destination stride
        uint32_t idx = (uint32_t)e, pos = 0;
        int64_t off = 0;
        for (int k = 0; k < p.nx; ++k) {
          const int g = p.xorder[k];
          const uint32_t d = p.dim_x[g];
          const uint32_t q = idx / d, r = idx - q * d;
          off += (int64_t)r * p.sd_x[g];
          uint32_t place = 1;                      // place value of group g in the source-order linear index
          for (int h = g + 1; h < p.nx; ++h) place *= p.dim_x[h];
          pos += r * place;
          idx = q;
        }
        dofs[i] = (int32_t)off;
        slots[i] = ((uint32_t)e + (uint32_t)e / skew) | ((pos + pos / skew) << 16);
      }
    }
  }

  auto zbases = [&](uint32_t z, int64_t& zs, int64_t& zd) {
    zs = p.src_offset;
    zd = 0;
    uint32_t idx = z;
    for (int g = p.nz - 1; g >= 0; --g) {
      const uint32_t d = p.dim_z[g];
      const uint32_t q = idx / d, r = idx - q * d;
      zs += (int64_t)r * p.ss_z[g];
      zd += (int64_t)r * p.sd_z[g];
      idx = q;
    }
  };

  T cur[EPT], nxt[EPT];
  int64_t zs, zd;
  zbases(z0, zs, zd);
#pragma unroll
  for (int i = 0; i < EPT; ++i)
    if (so[i] != kNone) cur[i] = src[zs + so[i]];
  for (uint32_t z = z0; z < z1; ++z) {
    const int64_t zd_cur = zd;
    if (z + 1 < z1) {
      zbases(z + 1, zs, zd);
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (so[i] != kNone) nxt[i] = src[zs + so[i]];
    }
    if (p.direct) {
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (dofs[i] != kNone) dst[zd_cur + dofs[i]] = cur[i];
    } else {
      if (z != z0) __syncthreads();                  // the previous tile has been read out
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (so[i] != kNone) tile[slots[i] & 0xffffu] = cur[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < EPT; ++i)
        if (dofs[i] != kNone) dst[zd_cur + dofs[i]] = tile[slots[i] >> 16];
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) cur[i] = nxt[i];
  }
}

// ---------------------------------------------------------------------------
// out[o] = sum_r x[off_o(o) + off_r(r)]   (one wave per output element when the
// reduction is long, one thread otherwise)
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.f; }
template <> __device__ __forceinline__ double zero_of<double>() { return 0.0; }
template <> __device__ __forceinline__ c64 zero_of<c64>() { return c64{0.f, 0.f}; }
template <> __device__ __forceinline__ c128 zero_of<c128>() { return c128{0.0, 0.0}; }
__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ double add(double a, double b) { return a + b; }
__device__ __forceinline__ c64 add(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c128 add(c128 a, c128 b) { return c128{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ float sub(float a, float b) { return a - b; }
__device__ __forceinline__ double sub(double a, double b) { return a - b; }
__device__ __forceinline__ c64 sub(c64 a, c64 b) { return c64{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c128 sub(c128 a, c128 b) { return c128{a.re - b.re, a.im - b.im}; }
// true division (op 3 of the binary kernel): IEEE a / b, not a * (1 / b)
__device__ __forceinline__ float div_(float a, float b) { return a / b; }
__device__ __forceinline__ double div_(double a, double b) { return a / b; }
__device__ __forceinline__ c64 div_(c64 a, c64 b) {
  const float d = b.re * b.re + b.im * b.im;
  return c64{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ __forceinline__ c128 div_(c128 a, c128 b) {
  const double d = b.re * b.re + b.im * b.im;
  return c128{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ double mul(double a, double b) { return a * b; }
__device__ __forceinline__ c64 mul(c64 a, c64 b) {
  return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ c128 mul(c128 a, c128 b) {
  return c128{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ float shfl_down_t(float v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ double shfl_down_t(double v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ c64 shfl_down_t(c64 v, int d) {
  return c64{__shfl_down(v.re, d, 64), __shfl_down(v.im, d, 64)};
}
__device__ __forceinline__ c128 shfl_down_t(c128 v, int d) {
  return c128{__shfl_down(v.re, d, 64), __shfl_down(v.im, d, 64)};
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_sum_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                          const ReduceArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  if (p.wave_per_out) {
    for (int64_t o = wave_global; o < (int64_t)p.n_keep; o += nwaves) {
      int64_t base = decomp1((uint32_t)o, p.nd_keep, p.dim_keep, p.s_keep);
      T acc = zero_of<T>();
      for (uint32_t r = lane; r < p.n_red; r += 64) acc = add(acc, x[base + decomp1(r, p.nd_red, p.dim_red, p.s_red)]);
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) acc = add(acc, shfl_down_t(acc, d));
      if (lane == 0) out[o] = acc;
    }
  } else {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < (int64_t)p.n_keep; i += stride) {
      int64_t base = decomp1((uint32_t)i, p.nd_keep, p.dim_keep, p.s_keep);
      T acc = zero_of<T>();
      for (uint32_t r = 0; r < p.n_red; ++r) acc = add(acc, x[base + decomp1(r, p.nd_red, p.dim_red, p.s_red)]);
      out[i] = acc;
    }
  }
}

// ---------------------------------------------------------------------------
// out[i] = a[offa(i)] op b[offb(i)]
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void binary_kernel(T* __restrict__ out, const T* __restrict__ a,
                                                      const T* __restrict__ b, const BinaryArgs p) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < p.n; i += stride) {
    int64_t oa = 0, ob = 0;
    int64_t idx = i;
    for (int g = p.nd - 1; g >= 0; --g) {
      int64_t d = p.dim[g];
      int64_t q = idx / d, r = idx - q * d;
      oa += r * p.sa[g];
      ob += r * p.sb[g];
      idx = q;
    }
    T va = a[oa], vb = b[ob];
    out[i] = p.op == 0 ? add(va, vb) : (p.op == 1 ? mul(va, vb) : (p.op == 2 ? sub(va, vb) : div_(va, vb)));
  }
}

// ---------------------------------------------------------------------------
// flat elementwise
// ---------------------------------------------------------------------------
template <typename T, typename R>
__global__ void scale_real_kernel(T* __restrict__ x, int64_t n, R f) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) x[i] = x[i] * f;
}
template <typename C, typename R>
__global__ void scale_cplx_kernel(C* __restrict__ x, int64_t n, R fr, R fi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    C v = x[i];
    x[i] = C{v.re * fr - v.im * fi, v.re * fi + v.im * fr};
  }
}
template <typename R>
__global__ void axpby_kernel(R* __restrict__ y, const R* __restrict__ x, int64_t n, R fy, R fx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = y[i] * fy + x[i] * fx;
}
// y * 10^ye + x * 10^xe  ->  y * 10^max(ye, xe), exponents on the device (slice accumulation
// without a host round trip); the exponent itself is advanced by a second one-thread kernel
template <typename R>
__global__ void axpby_exp_kernel(R* __restrict__ y, const R* __restrict__ x, int64_t n,
                                 const double* __restrict__ ye, const double* __restrict__ xe) {
  const double a = ye[0], b = xe[0];
  const double m = a > b ? a : b;
  const R fy = (R)(isfinite(a) ? pow(10.0, a - m) : 0.0), fx = (R)(isfinite(b) ? pow(10.0, b - m) : 0.0);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = y[i] * fy + x[i] * fx;
}
__global__ void max_exp_kernel(double* ye, const double* xe) {
  const double a = ye[0], b = xe[0];
  ye[0] = a > b ? a : b;
}
template <typename C>
__global__ void conj_kernel(C* __restrict__ dst, const C* __restrict__ src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    C v = src[i];
    dst[i] = C{v.re, -v.im};
  }
}
template <typename R>
__global__ void fill_kernel(R* __restrict__ dst, int64_t n, R re, R im, int cplx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (cplx) {
    for (; i < n; i += stride) { dst[2 * i] = re; dst[2 * i + 1] = im; }
  } else {
    for (; i < n; i += stride) dst[i] = re;
  }
}
// cast between {f32,f64,c64,c128}; real->complex sets im=0, complex->real drops im
template <typename D, typename S>
__global__ void cast_kernel(D* __restrict__ dst, const S* __restrict__ src, int64_t n, int dc, int sc) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    S re = sc ? src[2 * i] : src[i];
    S im = sc ? src[2 * i + 1] : S(0);
    if (dc) { dst[2 * i] = (D)re; dst[2 * i + 1] = (D)im; }
    else dst[i] = (D)re;
  }
}

// ---------------------------------------------------------------------------
// absmax / exponent stripping.  |x| >= 0 so the IEEE bit pattern orders like
// an unsigned integer: atomicMax on the bits is an exact, order-independent max.
// ---------------------------------------------------------------------------
template <typename R> struct Bits;
template <> struct Bits<float> {
  typedef unsigned int u;
  static __device__ u to(float v) { return __float_as_uint(v); }
  static __device__ float from(u b) { return __uint_as_float(b); }
};
template <> struct Bits<double> {
  typedef unsigned long long u;
  static __device__ u to(double v) { return (u)__double_as_longlong(v); }
  static __device__ double from(u b) { return __longlong_as_double((long long)b); }
};

template <typename R>
__global__ __launch_bounds__(256) void absmax_kernel(typename Bits<R>::u* __restrict__ scratch,
                                                      const R* __restrict__ x, int64_t n, int cplx) {
  __shared__ R red[4];
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  R m = R(0);
  if (cplx) {
    for (; i < n; i += stride) {
      R re = x[2 * i], im = x[2 * i + 1];
      R a = sqrt(re * re + im * im);
      m = a > m ? a : m;  // NaN-ignoring like a plain compare chain
    }
  } else {
    for (; i < n; i += stride) {
      R a = fabs(x[i]);
      m = a > m ? a : m;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    R o = __shfl_down(m, d, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
    atomicMax(scratch, Bits<R>::to(m));
  }
}

template <typename R>
__global__ void absmax_finish_kernel(double* __restrict__ out, typename Bits<R>::u* scratch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (double)Bits<R>::from(scratch[0]);
    scratch[0] = 0;
  }
}

template <typename R>
__global__ void strip_scale_kernel(R* __restrict__ x, int64_t n_real, const typename Bits<R>::u* scratch) {
  R m = Bits<R>::from(scratch[0]);
  if (!(m > R(0))) return;
  R inv = R(1) / m;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // true division keeps parity with the reference's `x / factor`
  for (; i < n_real; i += stride) x[i] = x[i] / m;
  (void)inv;
}

template <typename R>
__global__ void strip_finish_kernel(double* __restrict__ exponent, typename Bits<R>::u* scratch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    R m = Bits<R>::from(scratch[0]);
    // independent branches of a tree run on different HIP streams and share ONE accumulator: the add is atomic
    if (m > R(0)) atomicAdd(exponent, log10((double)m));
    scratch[0] = 0;
  }
}

// sum_t log10(max over tensor t's slots); one wave per launch is plenty
template <typename R, bool ADD = false>
__global__ void absmax_log10_sum_kernel(const R* __restrict__ slots, int64_t nt, double* __restrict__ out) {
  double s = 0.0;
  for (int64_t t = threadIdx.x; t < nt; t += 64) {
    R m = R(0);
    for (int i = 0; i < 64; ++i) {
      R v = slots[t * 64 + i];
      m = v > m ? v : m;
    }
    if (m > R(0)) s += log10((double)m);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if (threadIdx.x == 0) {
    if constexpr (ADD) atomicAdd(out, s);   // the accumulator also takes the un-fused strips of other lanes
    else out[0] = s;
  }
}

template <typename R>
__global__ void div_by_absmax_kernel(R* __restrict__ x, int64_t n_real, const R* __restrict__ slots) {
  R m = R(0);
  for (int i = 0; i < 64; ++i) {
    R v = slots[i];
    m = v > m ? v : m;
  }
  if (!(m > R(0))) return;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n_real; i += stride) x[i] = x[i] / m;
}

// complex operand -> real 2x2 blocks [[re, im], [-im, re]] so that a REAL contraction
// with K and N doubled performs the complex product (see ops._complex_gett)
template <typename R>
__global__ void complex_expand_kernel(R* __restrict__ dst, const R* __restrict__ src, int64_t n, int conj) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    R re = src[2 * i], im = src[2 * i + 1];
    if (conj) im = -im;
    dst[4 * i + 0] = re;
    dst[4 * i + 1] = im;
    dst[4 * i + 2] = -im;
    dst[4 * i + 3] = re;
  }
}

}  // namespace qamd

using namespace qamd;

static inline uint32_t flat_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (uint32_t)b;
}
#define QAMD_CHECK_LAUNCH() return (hipGetLastError() == hipSuccess ? 0 : -4)

extern "C" int qamd_permute_launch(int esize, void* dst, const void* src, const PermArgs* p, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  uint64_t grid = (uint64_t)p->tiles_x * p->tiles_y * p->Z;
  if (grid == 0 || grid > 0x7fffffffull) return -1;
  size_t lds = (size_t)(2 * p->TX + 2 * p->TY) * 8 + (size_t)(p->TX + 1) * p->TY * esize;
  switch (esize) {
    case 4: QAMD_LAUNCH(permute_kernel<float>, dim3((uint32_t)grid), dim3(256), lds, st, (float*)dst, (const float*)src, *p); break;
    case 8: QAMD_LAUNCH(permute_kernel<double>, dim3((uint32_t)grid), dim3(256), lds, st, (double*)dst, (const double*)src, *p); break;
    case 16: QAMD_LAUNCH(permute_kernel<c128>, dim3((uint32_t)grid), dim3(256), lds, st, (c128*)dst, (const c128*)src, *p); break;
    default: return -2;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_permute_stream_launch(int esize, void* dst, const void* src, const PermArgs* p, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const uint64_t grid = (p->Z + p->zchunk - 1) / p->zchunk;
  if (grid == 0 || grid > 0x7fffffffull) return -1;
  const int total = (int)p->X;
  if (total > 4096) return -2;
  const size_t lds = (size_t)(total + total / std::max<uint32_t>(p->dim_x[p->nx - 1], 1) + 2) * esize;
  if (lds > 64 * 1024) return -2;      // (the planner keeps tiles inside 64 KB; anything else goes to permute_kernel)
#define QAMD_PS(T, E) QAMD_LAUNCH((permute_stream_kernel<T, E>), dim3((uint32_t)grid), dim3(256), lds, st, (T*)dst, (const T*)src, *p)
  const int ept = (total + 255) / 256;
  switch (esize) {
    case 4: if (ept <= 8) QAMD_PS(float, 8); else QAMD_PS(float, 16); break;
    case 8: if (ept <= 8) QAMD_PS(double, 8); else QAMD_PS(double, 16); break;
    case 16: if (ept <= 8) QAMD_PS(c128, 8); else QAMD_PS(c128, 16); break;
    default: return -2;
  }
#undef QAMD_PS
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_reduce_sum_launch(int dtype, void* out, const void* x, const ReduceArgs* p, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int64_t threads = p->wave_per_out ? (int64_t)p->n_keep * 64 : (int64_t)p->n_keep;
  uint32_t grid = flat_grid(threads);
  switch (dtype) {
    case 0: QAMD_LAUNCH(reduce_sum_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)out, (const float*)x, *p); break;
    case 1: QAMD_LAUNCH(reduce_sum_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)out, (const double*)x, *p); break;
    case 2: QAMD_LAUNCH(reduce_sum_kernel<c64>, dim3(grid), dim3(256), 0, st, (c64*)out, (const c64*)x, *p); break;
    case 3: QAMD_LAUNCH(reduce_sum_kernel<c128>, dim3(grid), dim3(256), 0, st, (c128*)out, (const c128*)x, *p); break;
    default: return -2;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_binary_launch(int dtype, void* out, const void* a, const void* b, const BinaryArgs* p, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  uint32_t grid = flat_grid(p->n);
  switch (dtype) {
    case 0: QAMD_LAUNCH(binary_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)out, (const float*)a, (const float*)b, *p); break;
    case 1: QAMD_LAUNCH(binary_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)out, (const double*)a, (const double*)b, *p); break;
    case 2: QAMD_LAUNCH(binary_kernel<c64>, dim3(grid), dim3(256), 0, st, (c64*)out, (const c64*)a, (const c64*)b, *p); break;
    case 3: QAMD_LAUNCH(binary_kernel<c128>, dim3(grid), dim3(256), 0, st, (c128*)out, (const c128*)a, (const c128*)b, *p); break;
    default: return -2;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_scale(void* x, int64_t n, double re, double im, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_SCALE, x, nullptr, nullptr, nullptr, n, dtype, 0, re, im);
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  uint32_t grid = flat_grid(n);
  switch (dtype) {
    case 0: QAMD_LAUNCH((scale_real_kernel<float, float>), dim3(grid), dim3(256), 0, st, (float*)x, n, (float)re); break;
    case 1: QAMD_LAUNCH((scale_real_kernel<double, double>), dim3(grid), dim3(256), 0, st, (double*)x, n, re); break;
    case 2: QAMD_LAUNCH((scale_cplx_kernel<c64, float>), dim3(grid), dim3(256), 0, st, (c64*)x, n, (float)re, (float)im); break;
    case 3: QAMD_LAUNCH((scale_cplx_kernel<c128, double>), dim3(grid), dim3(256), 0, st, (c128*)x, n, re, im); break;
    default: return -2;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_axpby(void* y, const void* x, int64_t n, double fy, double fx, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_AXPBY, y, x, nullptr, nullptr, n, dtype, 0, fy, fx);
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  int64_t nr = (dtype >= 2) ? 2 * n : n;
  uint32_t grid = flat_grid(nr);
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH(axpby_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)y, (const float*)x, nr, (float)fy, (float)fx);
  else if (dtype == 1 || dtype == 3)
    QAMD_LAUNCH(axpby_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)y, (const double*)x, nr, fy, fx);
  else
    return -2;
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_axpby_exp(void* y, const void* x, int64_t n, void* y_exp_dev, const void* x_exp_dev, int32_t dtype,
                              void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_AXPBY_EXP, y, x, y_exp_dev, x_exp_dev, n, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3 || !y_exp_dev || !x_exp_dev) return -2;
  int64_t nr = dtype >= 2 ? 2 * n : n;
  if (nr > 0) {
    uint32_t grid = flat_grid(nr);
    if (dtype == 0 || dtype == 2)
      QAMD_LAUNCH(axpby_exp_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)y, (const float*)x, nr,
                  (const double*)y_exp_dev, (const double*)x_exp_dev);
    else
      QAMD_LAUNCH(axpby_exp_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)y, (const double*)x, nr,
                  (const double*)y_exp_dev, (const double*)x_exp_dev);
  }
  QAMD_LAUNCH(max_exp_kernel, dim3(1), dim3(1), 0, st, (double*)y_exp_dev, (const double*)x_exp_dev);
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_conj(void* dst, const void* src, int64_t n, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_CONJ, dst, src, nullptr, nullptr, n, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  uint32_t grid = flat_grid(n);
  switch (dtype) {
    case 0: if (dst != src) (void)hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, st); break;
    case 1: if (dst != src) (void)hipMemcpyAsync(dst, src, (size_t)n * 8, hipMemcpyDeviceToDevice, st); break;
    case 2: QAMD_LAUNCH(conj_kernel<c64>, dim3(grid), dim3(256), 0, st, (c64*)dst, (const c64*)src, n); break;
    case 3: QAMD_LAUNCH(conj_kernel<c128>, dim3(grid), dim3(256), 0, st, (c128*)dst, (const c128*)src, n); break;
    default: return -2;
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_fill(void* dst, int64_t n, double re, double im, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_FILL, dst, nullptr, nullptr, nullptr, n, dtype, 0, re, im);
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  uint32_t grid = flat_grid(n);
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH(fill_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)dst, n, (float)re, (float)im, dtype == 2);
  else if (dtype == 1 || dtype == 3)
    QAMD_LAUNCH(fill_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)dst, n, re, im, dtype == 3);
  else
    return -2;
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_cast(void* dst, int32_t dd, const void* src, int32_t sd, int64_t n, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_CAST, dst, src, nullptr, nullptr, n, dd, sd, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  if (dd < 0 || dd > 3 || sd < 0 || sd > 3) return -2;
  uint32_t grid = flat_grid(n);
  bool d64 = (dd & 1), s64 = (sd & 1);
  int dc = dd >= 2, sc = sd >= 2;
  if (!d64 && !s64) QAMD_LAUNCH((cast_kernel<float, float>), dim3(grid), dim3(256), 0, st, (float*)dst, (const float*)src, n, dc, sc);
  else if (!d64 && s64) QAMD_LAUNCH((cast_kernel<float, double>), dim3(grid), dim3(256), 0, st, (float*)dst, (const double*)src, n, dc, sc);
  else if (d64 && !s64) QAMD_LAUNCH((cast_kernel<double, float>), dim3(grid), dim3(256), 0, st, (double*)dst, (const float*)src, n, dc, sc);
  else QAMD_LAUNCH((cast_kernel<double, double>), dim3(grid), dim3(256), 0, st, (double*)dst, (const double*)src, n, dc, sc);
  QAMD_CHECK_LAUNCH();
}

// ---- elementwise maths the autoray boundary asks for (abs / sqrt / exp / log / log10) -------
template <typename R>
__device__ __forceinline__ R unary_apply(R v, int op) {
  switch (op) {
    case 0: return fabs(v);
    case 1: return sqrt(v);
    case 2: return exp(v);
    case 3: return log(v);
    default: return log10(v);
  }
}
template <typename R>
__global__ void unary_kernel(R* __restrict__ dst, const R* __restrict__ src, int64_t n, int op, int cplx_abs) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (cplx_abs) {
    for (; i < n; i += stride) {
      R re = src[2 * i], im = src[2 * i + 1];
      dst[i] = sqrt(re * re + im * im);
    }
  } else {
    for (; i < n; i += stride) dst[i] = unary_apply(src[i], op);
  }
}

// max / min over a real array: wave reduction + one compare-and-swap per wave on the value's bits
template <typename R> struct MmBits;
template <> struct MmBits<float> {
  typedef unsigned int u;
  static __device__ __forceinline__ u to(float v) { return __float_as_uint(v); }
  static __device__ __forceinline__ float from(u b) { return __uint_as_float(b); }
};
template <> struct MmBits<double> {
  typedef unsigned long long u;
  static __device__ __forceinline__ u to(double v) { return (u)__double_as_longlong(v); }
  static __device__ __forceinline__ double from(u b) { return __longlong_as_double((long long)b); }
};
template <typename R>
__global__ void minmax_init_kernel(R* out, int want_min) { out[0] = want_min ? R(INFINITY) : R(-INFINITY); }
template <typename R>
__global__ __launch_bounds__(256) void minmax_kernel(R* __restrict__ out, const R* __restrict__ x, int64_t n,
                                                      int want_min) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  R m = want_min ? R(INFINITY) : R(-INFINITY);
  for (; i < n; i += stride) {
    R v = x[i];
    m = want_min ? (v < m ? v : m) : (v > m ? v : m);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    R o = __shfl_down(m, d, 64);
    m = want_min ? (o < m ? o : m) : (o > m ? o : m);
  }
  if ((threadIdx.x & 63) == 0) {
    typedef typename MmBits<R>::u u;
    u* po = reinterpret_cast<u*>(out);
    u old = *po;
    while (true) {
      R cur = MmBits<R>::from(old);
      if (want_min ? !(m < cur) : !(m > cur)) break;
      u prev = atomicCAS(po, old, MmBits<R>::to(m));
      if (prev == old) break;
      old = prev;
    }
  }
}

extern "C" int qamd_unary(void* dst, const void* src, int64_t n, int32_t op, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_UNARY, dst, src, nullptr, nullptr, n, dtype, op, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3 || op < 0 || op > 4) return -2;
  if (dtype >= 2 && op != 0) return -2;   // complex: only abs (-> real magnitudes)
  if (n <= 0) return 0;
  const int cplx = dtype >= 2;
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH(unary_kernel<float>, dim3(flat_grid(n)), dim3(256), 0, st, (float*)dst, (const float*)src, n, op, cplx);
  else
    QAMD_LAUNCH(unary_kernel<double>, dim3(flat_grid(n)), dim3(256), 0, st, (double*)dst, (const double*)src, n, op, cplx);
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_minmax(void* out_dev, const void* x, int64_t n, int32_t want_min, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_MINMAX, out_dev, x, nullptr, nullptr, n, dtype, want_min, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype != 0 && dtype != 1) return -2;
  if (n <= 0) return -1;
  uint32_t grid = flat_grid(n);
  if (grid > 1024) grid = 1024;
  if (dtype == 0) {
    QAMD_LAUNCH(minmax_init_kernel<float>, dim3(1), dim3(1), 0, st, (float*)out_dev, want_min);
    QAMD_LAUNCH(minmax_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)out_dev, (const float*)x, n, want_min);
  } else {
    QAMD_LAUNCH(minmax_init_kernel<double>, dim3(1), dim3(1), 0, st, (double*)out_dev, want_min);
    QAMD_LAUNCH(minmax_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)out_dev, (const double*)x, n, want_min);
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_absmax(void* out_dev, const void* x, int64_t n, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_ABSMAX, out_dev, x, nullptr, nullptr, n, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3) return -2;
  // the 8 bytes after out_dev[0] are used as scratch: out_dev must be >= 16 bytes
  void* scratch = (char*)out_dev + 8;
  (void)hipMemsetAsync(scratch, 0, 8, st);
  uint32_t grid = flat_grid(n);
  if (grid > 1024) grid = 1024;
  int cplx = dtype >= 2;
  if (dtype == 0 || dtype == 2) {
    if (n > 0) QAMD_LAUNCH(absmax_kernel<float>, dim3(grid), dim3(256), 0, st, (unsigned int*)scratch, (const float*)x, n, cplx);
    QAMD_LAUNCH(absmax_finish_kernel<float>, dim3(1), dim3(64), 0, st, (double*)out_dev, (unsigned int*)scratch);
  } else {
    if (n > 0) QAMD_LAUNCH(absmax_kernel<double>, dim3(grid), dim3(256), 0, st, (unsigned long long*)scratch, (const double*)x, n, cplx);
    QAMD_LAUNCH(absmax_finish_kernel<double>, dim3(1), dim3(64), 0, st, (double*)out_dev, (unsigned long long*)scratch);
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_strip_exponent(void* x, int64_t n, int32_t dtype, void* scratch_dev,
                                   void* exponent_dev, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_STRIP, x, scratch_dev, exponent_dev, nullptr, n, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3) return -2;
  if (n <= 0) return 0;
  (void)hipMemsetAsync(scratch_dev, 0, 8, st);
  uint32_t grid = flat_grid(n);
  if (grid > 1024) grid = 1024;
  int cplx = dtype >= 2;
  int64_t nr = cplx ? 2 * n : n;
  if (dtype == 0 || dtype == 2) {
    QAMD_LAUNCH(absmax_kernel<float>, dim3(grid), dim3(256), 0, st, (unsigned int*)scratch_dev, (const float*)x, n, cplx);
    QAMD_LAUNCH(strip_scale_kernel<float>, dim3(flat_grid(nr)), dim3(256), 0, st, (float*)x, nr, (const unsigned int*)scratch_dev);
    QAMD_LAUNCH(strip_finish_kernel<float>, dim3(1), dim3(64), 0, st, (double*)exponent_dev, (unsigned int*)scratch_dev);
  } else {
    QAMD_LAUNCH(absmax_kernel<double>, dim3(grid), dim3(256), 0, st, (unsigned long long*)scratch_dev, (const double*)x, n, cplx);
    QAMD_LAUNCH(strip_scale_kernel<double>, dim3(flat_grid(nr)), dim3(256), 0, st, (double*)x, nr, (const unsigned long long*)scratch_dev);
    QAMD_LAUNCH(strip_finish_kernel<double>, dim3(1), dim3(64), 0, st, (double*)exponent_dev, (unsigned long long*)scratch_dev);
  }
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_absmax_log10_sum(const void* slots, int64_t nt, int32_t dtype, void* out_dev, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_LOG10SUM, slots, out_dev, nullptr, nullptr, nt, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3) return -2;
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH(absmax_log10_sum_kernel<float>, dim3(1), dim3(64), 0, st, (const float*)slots, nt, (double*)out_dev);
  else
    QAMD_LAUNCH(absmax_log10_sum_kernel<double>, dim3(1), dim3(64), 0, st, (const double*)slots, nt, (double*)out_dev);
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_absmax_log10_sum_add(const void* slots, int64_t nt, int32_t dtype, void* acc_dev, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_LOG10SUM_ADD, slots, acc_dev, nullptr, nullptr, nt, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3) return -2;
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH((absmax_log10_sum_kernel<float, true>), dim3(1), dim3(64), 0, st, (const float*)slots, nt, (double*)acc_dev);
  else
    QAMD_LAUNCH((absmax_log10_sum_kernel<double, true>), dim3(1), dim3(64), 0, st, (const double*)slots, nt, (double*)acc_dev);
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_div_by_absmax(void* x, int64_t n, const void* slots, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_DIVABS, x, slots, nullptr, nullptr, n, dtype, 0, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype < 0 || dtype > 3) return -2;
  if (n <= 0) return 0;
  int64_t nr = dtype >= 2 ? 2 * n : n;
  if (dtype == 0 || dtype == 2)
    QAMD_LAUNCH(div_by_absmax_kernel<float>, dim3(flat_grid(nr)), dim3(256), 0, st, (float*)x, nr, (const float*)slots);
  else
    QAMD_LAUNCH(div_by_absmax_kernel<double>, dim3(flat_grid(nr)), dim3(256), 0, st, (double*)x, nr, (const double*)slots);
  QAMD_CHECK_LAUNCH();
}

extern "C" int qamd_complex_expand(void* dst, const void* src, int64_t n, int32_t conj, int32_t dtype, void* stream) {
  if (qamdp_recording()) return qamdp_rec_simple(QP_CEXPAND, dst, src, nullptr, nullptr, n, dtype, conj, 0.0, 0.0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype != 2 && dtype != 3) return -2;
  if (n <= 0) return 0;
  if (dtype == 2)
    QAMD_LAUNCH(complex_expand_kernel<float>, dim3(flat_grid(n)), dim3(256), 0, st, (float*)dst, (const float*)src, n, conj);
  else
    QAMD_LAUNCH(complex_expand_kernel<double>, dim3(flat_grid(n)), dim3(256), 0, st, (double*)dst, (const double*)src, n, conj);
  QAMD_CHECK_LAUNCH();
}
