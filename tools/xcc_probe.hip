// Which XCD does workgroup b land on?  (speed-only question: validates the `b % 8` affinity assumption)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(x & 0xf);
  }
}
int main() {
  for (int cfg = 0; cfg < 3; ++cfg) {
    dim3 grid = cfg == 0 ? dim3(1024) : cfg == 1 ? dim3(36, 12, 2) : dim3(443, 2);
    int n = grid.x * grid.y * grid.z;
    int* d; hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(probe, grid, dim3(256), 0, 0, d);
    int* h = (int*)malloc(n * 4); hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int ok = 0; for (int i = 0; i < n; ++i) ok += (h[i] == h[i % 8]);
    printf("grid (%d,%d,%d): first 16 xcc:", grid.x, grid.y, grid.z);
    for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
    printf("  | blocks with xcc[b]==xcc[b%%8]: %d / %d\n", ok, n);
    hipFree(d); free(h);
  }
  return 0;
}
