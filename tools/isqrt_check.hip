#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ int isqrt_floor(int d) {
  int s = (int)sqrtf((float)d);
  s -= (__umul24((unsigned)s, (unsigned)s) > (unsigned)d) ? 1 : 0;
  s += (__umul24((unsigned)(s + 1), (unsigned)(s + 1)) <= (unsigned)d) ? 1 : 0;
  return s;
}
__global__ void k(unsigned long long *bad) {
  // every d in [0, 2^31): thread-strided
  for (long long d = blockIdx.x * (long long)blockDim.x + threadIdx.x; d < (1ll << 31); d += (long long)gridDim.x * blockDim.x) {
    const int s = isqrt_floor((int)d);
    const long long S = s;
    if (!(S * S <= d && (S + 1) * (S + 1) > d)) atomicAdd(bad, 1ull);
  }
}
int main() {
  unsigned long long *b, h = 0;
  hipMalloc(&b, 8); hipMemset(b, 0, 8);
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, b);
  hipMemcpy(&h, b, 8, hipMemcpyDeviceToHost);
  printf("isqrt_floor over [0, 2^31): %llu wrong\n", h);
  return h != 0;
}
