// gridbar.hip -- probe: what does a spin barrier over G co-resident workgroups cost on gfx950 (agent-scope fences, per-XCD
// L2s), and do plain loads after it see the other workgroups' stores?  Prices the "walker" design (several dependent
// small contraction steps in ONE launch instead of one ~9 us dependent dispatch each).
//   hipcc --offload-arch=gfx950 -O3 gridbar.hip -o bin/gridbar && bin/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void group_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

// every iteration: each workgroup stores (iter + 1) into its PAYLOAD floats, barrier, then reads the payload of workgroup
// (wg + 1 + iter) % G and counts mismatches
__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, float* buf, int payload, int iters, unsigned* bad) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < payload; i += 256) buf[(size_t)wg * payload + i] = (float)(it + 1);
    group_barrier(counter, (unsigned)(2 * it + 1) * G);
    const int src = (wg + 1 + it) % G;
    for (int i = tid; i < payload; i += 256) nbad += buf[(size_t)src * payload + i] != (float)(it + 1);
    group_barrier(counter, (unsigned)(2 * it + 2) * G);   // nobody overwrites before everybody has read
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  const int iters = 200;
  for (int G : {16, 64, 128, 256, 512}) {
    for (int payload : {256, 16384}) {
      for (int nstreams : {1, 4}) {
        if (nstreams * G > 1024) continue;
        std::vector<hipStream_t> st(nstreams);
        std::vector<unsigned*> ctr(nstreams);
        std::vector<float*> buf(nstreams);
        unsigned* bad;
        hipMalloc(&bad, 4);
        hipMemset(bad, 0, 4);
        for (int s = 0; s < nstreams; ++s) {
          hipStreamCreate(&st[s]);
          hipMalloc(&ctr[s], 4);
          hipMalloc(&buf[s], (size_t)G * payload * 4);
        }
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
          for (int s = 0; s < nstreams; ++s) hipMemsetAsync(ctr[s], 0, 4, st[s]);
          hipDeviceSynchronize();
          hipEvent_t e0, e1;
          hipEventCreate(&e0);
          hipEventCreate(&e1);
          hipEventRecord(e0, st[0]);
          for (int s = 0; s < nstreams; ++s)
            hipLaunchKernelGGL(bar_kernel, dim3(G), dim3(256), 0, st[s], ctr[s], buf[s], payload, iters, bad);
          for (int s = 1; s < nstreams; ++s) {
            hipEvent_t e;
            hipEventCreate(&e);
            hipEventRecord(e, st[s]);
            hipStreamWaitEvent(st[0], e, 0);
          }
          hipEventRecord(e1, st[0]);
          hipDeviceSynchronize();
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        unsigned hb;
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        printf("G=%4d payload=%6d floats/wg streams=%d: %7.2f us per (store, barrier, load, barrier) round  -> %5.2f us per barrier+phase; stale reads: %u\n",
               G, payload, nstreams, best * 1e3f / iters, best * 1e3f / iters / 2, hb);
      }
    }
  }
  return 0;
}
