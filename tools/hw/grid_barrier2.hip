// Probe (round 4): grid-wide barrier inside ONE persistent launch, three forms, against the number of workgroups:
//   A  round 3's form (tools/hw/grid_barrier.hip): one flat counter, two __threadfence(), acquire-load polling
//   B  the hand-off protocol of the merged q/k/v kernels (MI355X_MICROARCH.md, inter-workgroup visibility): data stored
//      write-through (sc1), every wave drains vmcnt, workgroup barrier, ONE relaxed agent-scope add, relaxed sc1 polling with
//      s_sleep, data read back with sc1 loads -- no fences at all; still one flat counter
//   C  the same protocol, XCD-hierarchical: one counter per XCD (HW_REG_XCC_ID), the XCD's last arriver adds to a top counter,
//      the top's last arriver bumps one generation word per XCD, every workgroup polls its own XCD's word
// Every round moves a 1-KiB row per workgroup (written before the barrier, the neighbour's row read after it).
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier2 grid_barrier2.hip && ./grid_barrier2
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float ld_sc1_f(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sc1_f(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// sync words: [0] flat count, [32] flat generation, [64 + 32 x] XCD x count, [512] top count, [576 + 32 x] XCD x generation
template <int FORM>
__device__ __forceinline__ void grid_sync(unsigned* s, unsigned nwg, unsigned& my_gen, unsigned xcd, unsigned nxcd_wg /* workgroups on my XCD */,
                                          unsigned nxcd /* XCDs in use */) {
  const unsigned target = my_gen + 1;
  if (FORM == 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (__hip_atomic_fetch_add(s, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
        __hip_atomic_store(s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(s + 32, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(s + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its sc1 stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      if (FORM == 1) {
        if (__hip_atomic_fetch_add(s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
          __hip_atomic_store(s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(s + 32, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          while (ld_sc1(s + 32) != target) __builtin_amdgcn_s_sleep(1);
        }
      } else {
        unsigned* xc = s + 64 + 32 * xcd;
        if (__hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd_wg - 1) {
          __hip_atomic_store(xc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (__hip_atomic_fetch_add(s + 512, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nxcd - 1) {
            __hip_atomic_store(s + 512, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned x = 0; x < 8; ++x) __hip_atomic_store(s + 576 + 32 * x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        while (ld_sc1(s + 576 + 32 * xcd) != target) __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  ++my_gen;
}

// census: which XCD every workgroup runs on (the hierarchical form needs the per-XCD workgroup counts)
__global__ void k_census(unsigned* per_xcd) {
  if (threadIdx.x == 0) atomicAdd(per_xcd + xcc_id(), 1u);
}

template <int FORM>
__global__ __launch_bounds__(512) void k_rounds(unsigned* s, float* data, int rounds, unsigned gen0, const unsigned* per_xcd) {
  unsigned my_gen = gen0;
  const unsigned nwg = gridDim.x;
  const unsigned xcd = xcc_id();
  unsigned nx = 0;
  for (int x = 0; x < 8; ++x) nx += per_xcd[x] != 0;
  const unsigned nxw = per_xcd[xcd];
  // every round: workgroup b writes (round, b) into its row, then -- behind the barrier -- must find (round, b + 1) in its
  // neighbour's row: `acc` counts the words that were stale (0 = every hand-off of every round was seen)
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x < 256) {
      const float v = (float)(r * 1024 + (int)blockIdx.x);
      if (FORM == 0) data[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
      else st_sc1_f(&data[(size_t)blockIdx.x * 256 + threadIdx.x], v);
    }
    grid_sync<FORM>(s, nwg, my_gen, xcd, nxw, nx);
    if (threadIdx.x < 256) {
      const unsigned nb = (blockIdx.x + 1) % nwg;
      const float* p = &data[(size_t)nb * 256 + threadIdx.x];
      const float got = FORM == 0 ? __builtin_nontemporal_load(p) : ld_sc1_f(p);
      acc += got != (float)(r * 1024 + (int)nb) ? 1.f : 0.f;
    }
    // (the next round's write of my row must not pass a neighbour that still reads it: a second barrier would double the cost
    //  being measured -- rows are double-buffered by round parity instead)
    data += (r & 1) ? -(ptrdiff_t)(256 * 256) : (ptrdiff_t)(256 * 256);
  }
  if (threadIdx.x < 256) atomicAdd(&s[1000], (unsigned)acc);
}

template <int FORM>
static void run(const char* name, unsigned* sync, float* data, unsigned* census, unsigned& gen0) {
  for (int wgs : {8, 32, 64, 128, 256}) {
    hipMemset(census, 0, 64);
    hipLaunchKernelGGL(k_census, dim3(wgs), dim3(512), 0, 0, census);      // same grid -> same round-robin placement
    hipDeviceSynchronize();
    float t[2];
    int ri = 0;
    for (int rounds : {1, 201}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rounds<FORM>, dim3(wgs), dim3(512), 0, 0, sync, data, rounds, gen0, census);
        hipEventRecord(e1); hipEventSynchronize(e1);
        gen0 += rounds;
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      t[ri++] = best * 1e3f;
    }
    unsigned stale; hipMemcpy(&stale, sync + 1000, sizeof(stale), hipMemcpyDeviceToHost);
    hipMemset(sync + 1000, 0, 4);
    printf("%-28s %4d workgroups: %7.2f us per round incl. 1-KiB exchange (201 rounds: %8.2f us, 1 round: %6.2f us)  stale words: %u\n", name, wgs,
           (t[1] - t[0]) / 200.f, t[1], t[0], stale);
  }
}

int main() {
  unsigned *sync, *census; float* data;
  hipMalloc(&sync, 8192); hipMemset(sync, 0, 8192);
  hipMalloc(&census, 64);
  hipMalloc(&data, 4 << 20); hipMemset(data, 0, 4 << 20);
  unsigned gen0 = 0;
  run<0>("A flat, fences, acquire poll", sync, data, census, gen0);
  hipMemset(sync, 0, 8192); gen0 = 0;
  run<1>("B flat, sc1 data, relaxed", sync, data, census, gen0);
  hipMemset(sync, 0, 8192); gen0 = 0;
  run<2>("C per-XCD hierarchy, sc1", sync, data, census, gen0);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
