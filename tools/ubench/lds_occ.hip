// how many 256-thread workgroups with a given dynamic LDS size fit on one CU (occupancy API)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(double* o) { extern __shared__ double s[]; s[threadIdx.x] = 1.0; __syncthreads(); o[threadIdx.x] = s[255 - threadIdx.x]; }
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int bytes = 30 * 1024; bytes <= 42 * 1024; bytes += 512) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, bytes);
    printf("%d B (%.1f KB): %d blocks/CU\n", bytes, bytes / 1024.0, nb);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlock %zu\n", p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock);
  return 0;
}
