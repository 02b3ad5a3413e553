// Experiment (round 5): what does a grid-wide barrier cost on MI355X (256 workgroups x 512 threads, one per CU)?  It prices a persistent
// B = 1 trunk kernel (33 dependent convolutions in one launch) against today's 4.6 us per dependent graph node.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/grid_barrier_cost tools/experiments/grid_barrier_cost.hip (then through gpurun)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(512) void k_cg(float* buf, int iters) {
  cg::grid_group g = cg::this_grid();
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  float v = buf[gid];
  for (int i = 0; i < iters; ++i) {
    buf[(gid + 7919 * (i + 1)) % n] = v + 1.f;      // every phase writes what another workgroup reads next
    g.sync();
    v = buf[gid];
  }
  buf[gid] = v;
}

// hand-written: one arrival counter per phase parity, agent-scope release / acquire
__global__ __launch_bounds__(512) void k_hand(float* buf, unsigned* ctr, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  float v = buf[gid];
  unsigned target = 0;
  for (int i = 0; i < iters; ++i) {
    buf[(gid + 7919 * (i + 1)) % n] = v + 1.f;
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // (the un-scoped builtins are SYSTEM scope: 26 us per phase)
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    v = buf[gid];
  }
  buf[gid] = v;
}

// no cache maintenance at all: the exchanged data moves with write-through stores / cache-bypassing loads (sc0 sc1), the counter with relaxed
// agent-scope atomics; s_waitcnt vmcnt(0) orders a thread's stores before its workgroup's arrival
__device__ __forceinline__ void st_coherent(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float ld_coherent(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__global__ __launch_bounds__(512) void k_hand2(float* buf, unsigned* ctr, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
  float v = buf[gid];
  unsigned target = 0;
  for (int i = 0; i < iters; ++i) {
    st_coherent(buf + (gid + 7919 * (i + 1)) % n, v + 1.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    v = ld_coherent(buf + gid);
  }
  buf[gid] = v;
}

__global__ void k_empty(float* buf) { if (buf == nullptr) buf[0] = 0; }

int main() {
  const int G = 256, T = 512, N = G * T, IT = 200;
  float* buf; unsigned* ctr;
  hipMalloc(&buf, N * sizeof(float)); hipMalloc(&ctr, 64);
  hipMemset(buf, 0, N * sizeof(float)); hipMemset(ctr, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    int iters = IT; void* args[] = {&buf, &iters};
    hipEventRecord(e0);
    hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(G), dim3(T), args, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("cooperative grid.sync(): %s, %.2f us per phase (%d phases)\n", hipGetErrorString(e), ms * 1e3 / IT, IT);
    hipMemset(ctr, 0, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_hand, dim3(G), dim3(T), 0, 0, buf, ctr, IT);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("hand-written barrier:    %.2f us per phase\n", ms * 1e3 / IT);
    hipMemset(ctr, 0, 64); hipMemset(buf, 0, N * sizeof(float));
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_hand2, dim3(G), dim3(T), 0, 0, buf, ctr, IT);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    {
      static float host[256 * 512];
      hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < N; ++i) bad += host[i] != (float)IT;
      printf("write-through data + relaxed counter: %.2f us per phase, %d of %d values wrong\n", ms * 1e3 / IT, bad, N);
    }
    // the alternative: a chain of dependent launches under a graph
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t gr; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < IT; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(T), 0, s, buf);
    hipStreamEndCapture(s, &gr); hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("graph of %d empty dependent launches: %.2f us per node\n", IT, ms * 1e3 / IT);
    hipGraphExecDestroy(ge); hipGraphDestroy(gr); hipStreamDestroy(s);
  }
  float h; hipMemcpy(&h, buf, 4, hipMemcpyDeviceToHost); printf("check %.0f\n", h);
  return 0;
}
