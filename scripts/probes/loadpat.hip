// Probe: what read bandwidth does the fused-pair load pattern sustain on its own?
// R rows spaced `rs` floats apart; every wave reads a CH-float piece of every row per chunk
// (CH = 16*V floats = 64*V bytes), 4 waves of a workgroup interleave chunks.
//   hipcc --offload-arch=gfx950 -O3 loadpat.hip -o /tmp/loadpat && /tmp/loadpat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int V> struct Vec { typedef float type __attribute__((ext_vector_type(V))); };
template <> struct Vec<1> { typedef float type; };

template <int V, int KS, int DEPTH>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ A, float* __restrict__ out, int64_t rs,
                                                uint32_t chunks, uint32_t cpb) {
  extern __shared__ float smem[];
  typedef typename Vec<V>::type vt;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const uint32_t c0 = blockIdx.x * cpb + wave, c1 = min(chunks, (blockIdx.x + 1) * cpb);
  vt acc = {};
  vt buf[DEPTH][KS];
  const float* base = A + (int64_t)kq * rs + V * j;
  uint32_t c = c0;
  // DEPTH chunks in flight
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
    uint32_t cc = min(c + 4 * d, chunks - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) buf[d][s] = __builtin_nontemporal_load((const vt*)(base + (int64_t)(4 * s) * rs + (int64_t)cc * 16 * V));
  }
  for (; c < c1; c += 4 * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int nd = (d + DEPTH - 1) % DEPTH;
      uint32_t cc = min(c + 4 * (d + DEPTH - 1), chunks - 1);
#pragma unroll
      for (int s = 0; s < KS; ++s) buf[nd][s] = __builtin_nontemporal_load((const vt*)(base + (int64_t)(4 * s) * rs + (int64_t)cc * 16 * V));
#pragma unroll
      for (int s = 0; s < KS; ++s) acc += buf[d][s];
    }
  }
  float r;
  if constexpr (V == 1) r = acc; else { r = 0; for (int i = 0; i < V; ++i) r += acc[i]; }
  if (r == 12345.678f) out[threadIdx.x] = r + smem[threadIdx.x];
}

template <int V, int KS, int DEPTH>
void run(const char* name, const float* A, float* out, int64_t rs, int64_t M, size_t lds) {
  uint32_t chunks = (uint32_t)(M / (16 * V));
  uint32_t cpb = 36;
  while (chunks % cpb) --cpb;
  uint32_t grid = chunks / cpb;
  hipFuncSetAttribute((const void*)probe<V, KS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<V, KS, DEPTH>), dim3(grid), dim3(256), lds, 0, A, out, rs, chunks, cpb);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<V, KS, DEPTH>), dim3(grid), dim3(256), lds, 0, A, out, rs, chunks, cpb);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double bytes = (double)(4 * KS) * M * 4;
  printf("%-34s rows=%3d V=%d depth=%d lds=%3zuK grid=%6u  %.3f ms  %.0f GB/s\n", name, 4 * KS, V, DEPTH, lds / 1024, grid, ms, bytes / ms / 1e6);
}

int main() {
  const int64_t M = 1679616;  // 6^8
  float *A, *out;
  hipMalloc(&A, (size_t)216 * M * 4 + (1 << 20));
  hipMalloc(&out, 4096);
  hipMemset(A, 0, (size_t)216 * M * 4);
  run<1, 54, 2>("216 rows x 64B, 2 chunks in flight", A, out, M, M, 72 * 1024);
  run<2, 54, 2>("216 rows x 128B", A, out, M, M, 72 * 1024);
  run<4, 54, 2>("216 rows x 256B", A, out, M, M, 72 * 1024);
  run<1, 54, 2>("216 rows x 64B, 4 blocks/CU", A, out, M, M, 36 * 1024);
  run<1, 9, 2>("36 rows x 64B (6x longer rows)", A, out, 6 * M, 6 * M, 72 * 1024);
  run<4, 9, 2>("36 rows x 256B (sweep-like)", A, out, 6 * M, 6 * M, 72 * 1024);
  run<1, 9, 4>("36 rows x 64B depth 4", A, out, 6 * M, 6 * M, 72 * 1024);
  run<1, 54, 1>("216 rows x 64B, depth 1", A, out, M, M, 72 * 1024);
  return 0;
}
