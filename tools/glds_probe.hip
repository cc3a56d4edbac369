// Does global_load_lds_dwordx4 (LDS-DMA, gfx950) accept source addresses that are only 8-byte aligned?
// Each lane copies 16 bytes from src + shift + 16 * lane ... into LDS (wave-uniform base + 16 * lane), the
// workgroup then writes the LDS image out; the host compares.  Prints one line per shift.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* LdsPtr;
typedef const __attribute__((address_space(1))) void* GlobalPtr;

__global__ __launch_bounds__(256) void copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  __shared__ double buf[2048];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pos = j * 512 + wave * 128 + 2 * lane;          // this lane's pair of doubles
    if (pos < n)
      __builtin_amdgcn_global_load_lds((GlobalPtr)(src + pos), (LdsPtr)(buf + j * 512 + wave * 128), 16, 0, 0);
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) dst[i] = buf[i];
}

int main() {
  const int n = 2048;
  double *src, *dst;
  hipMalloc(&src, sizeof(double) * (n + 64));
  hipMalloc(&dst, sizeof(double) * n);
  std::vector<double> h(n + 64), out(n);
  for (int i = 0; i < n + 64; ++i) h[i] = 1000.0 + i;
  hipMemcpy(src, h.data(), sizeof(double) * (n + 64), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    for (int len : {2048, 1500, 130}) {
      hipMemset(dst, 0, sizeof(double) * n);
      hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(256), 0, 0, src + shift, dst, len);
      if (hipDeviceSynchronize() != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
      hipMemcpy(out.data(), dst, sizeof(double) * n, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < len; ++i) bad += out[i] != h[i + shift];
      printf("source offset %d doubles (%s-byte aligned), %d doubles: %d mismatches\n", shift, shift % 2 ? "8" : "16", len, bad);
    }
  }
  return 0;
}
