// v_permlane32_swap_b32 on gfx950: what goes where.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  out[threadIdx.x] = a;
  out[64 + threadIdx.x] = b;
}
int main() {
  int* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); int h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("a: lane0 %d lane31 %d lane32 %d lane63 %d\n", h[0], h[31], h[32], h[63]);
  printf("b: lane0 %d lane31 %d lane32 %d lane63 %d\n", h[64], h[95], h[96], h[127]);
  return 0;
}
