// Probe: what a grid-wide barrier (one atomic arrival per workgroup + spin on a generation word, device scope) costs inside ONE
// persistent launch, against the number of workgroups -- the alternative to a chain of dependent launches (tools/hw/launch_cost.hip:
// 2.9 us each) for the cross-clip layer chain.  Every round also moves a little data through L2 between the workgroups (a 1-KiB
// row written before the barrier, a neighbour's row read after it) so that the cost includes the release / acquire of real data.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_sync(unsigned* count, unsigned* gen, unsigned nwg, unsigned& my_gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                                   // release this workgroup's stores
    const unsigned target = my_gen + 1;
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  ++my_gen;
  __syncthreads();
}

__global__ __launch_bounds__(512) void k_rounds(unsigned* count, unsigned* gen, float* data, int rounds, unsigned gen0) {
  unsigned my_gen = gen0;
  const unsigned nwg = gridDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x < 256) data[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + r;
    grid_sync(count, gen, nwg, my_gen);
    if (threadIdx.x < 256) acc += __builtin_nontemporal_load(&data[(size_t)((blockIdx.x + 1) % nwg) * 256 + threadIdx.x]);
  }
  if (threadIdx.x < 256) data[(size_t)(nwg + blockIdx.x) * 256 + threadIdx.x] = acc;
}

int main() {
  unsigned* sync; float* data;
  hipMalloc(&sync, 256); hipMemset(sync, 0, 256);
  hipMalloc(&data, 4 << 20); hipMemset(data, 0, 4 << 20);
  unsigned gen0 = 0;
  for (int wgs : {8, 32, 64, 128, 256}) {
    for (int rounds : {1, 101}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds, dim3(wgs), dim3(512), 0, 0, sync, sync + 32, data, rounds, gen0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        gen0 += rounds;
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("%4d workgroups, %4d rounds: %8.2f us per launch", wgs, rounds, best * 1e3);
      if (rounds > 1) printf("   (%.2f us per round incl. 1-KiB exchange)", best * 1e3 / rounds);
      printf("\n");
    }
  }
  float h[256]; hipMemcpy(h, data + 256 * 256, sizeof(h), hipMemcpyDeviceToHost);
  printf("check %g  %s\n", h[0], hipGetErrorString(hipGetLastError()));
  return 0;
}
