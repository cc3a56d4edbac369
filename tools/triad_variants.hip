// Stand-alone STREAM-triad / copy variant sweep for gfx950 (not part of the library):
//   hipcc -O3 --offload-arch=gfx950 tools/triad_variants.hip -o build/triad_variants && build/triad_variants
// Prints GB/s (algorithmic: 24 B per element for the triad, 16 B for the copy) of every variant so
// that the library's roofline-denominator kernel (csrc/blas1.hip: triad_kernel) can be set to the
// fastest form.  Buffers: 3 x 2^26 doubles = 1.6 GB (far beyond the 256-MiB Infinity Cache).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef double double2v __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void triad(long n2, double a, const double2v* __restrict__ x,
                                             const double2v* __restrict__ y, double2v* __restrict__ z) {
  const long stride = long(gridDim.x) * 256;
  long i = long(blockIdx.x) * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
    double2v xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      xv[u] = NTL ? __builtin_nontemporal_load(&x[i + u * stride]) : x[i + u * stride];
      yv[u] = NTL ? __builtin_nontemporal_load(&y[i + u * stride]) : y[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      double2v r;
      r.x = fma(a, yv[u].x, xv[u].x);
      r.y = fma(a, yv[u].y, xv[u].y);
      if (NTS) __builtin_nontemporal_store(r, &z[i + u * stride]);
      else z[i + u * stride] = r;
    }
  }
  for (; i < n2; i += stride) {
    double2v xv = x[i], yv = y[i], r;
    r.x = fma(a, yv.x, xv.x);
    r.y = fma(a, yv.y, xv.y);
    z[i] = r;
  }
}

// blocked variant: each workgroup owns a contiguous chunk (UNROLL x 256 double2 per step)
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void triad_blocked(long n2, double a, const double2v* __restrict__ x,
                                                     const double2v* __restrict__ y, double2v* __restrict__ z) {
  const long per = (n2 + gridDim.x - 1) / gridDim.x;
  const long b0 = blockIdx.x * per, b1 = (b0 + per < n2) ? b0 + per : n2;
  long i = b0 + threadIdx.x;
  for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
    double2v xv[UNROLL], yv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      xv[u] = NTL ? __builtin_nontemporal_load(&x[i + u * 256]) : x[i + u * 256];
      yv[u] = NTL ? __builtin_nontemporal_load(&y[i + u * 256]) : y[i + u * 256];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      double2v r;
      r.x = fma(a, yv[u].x, xv[u].x);
      r.y = fma(a, yv[u].y, xv[u].y);
      if (NTS) __builtin_nontemporal_store(r, &z[i + u * 256]);
      else z[i + u * 256] = r;
    }
  }
  for (; i < b1; i += 256) {
    double2v xv = x[i], yv = y[i], r;
    r.x = fma(a, yv.x, xv.x);
    r.y = fma(a, yv.y, xv.y);
    z[i] = r;
  }
}

typedef float float4v __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy16(long n4, const float4v* __restrict__ x, float4v* __restrict__ z) {
  const long stride = long(gridDim.x) * 256;
  for (long i = long(blockIdx.x) * 256 + threadIdx.x; i < n4; i += stride) {
    float4v v = NTL ? __builtin_nontemporal_load(&x[i]) : x[i];
    if (NTS) __builtin_nontemporal_store(v, &z[i]);
    else z[i] = v;
  }
}

template <class F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const long n = 1L << 26;
  double *x, *y, *z;
  CHECK(hipMalloc(&x, n * 8));
  CHECK(hipMalloc(&y, n * 8));
  CHECK(hipMalloc(&z, n * 8));
  CHECK(hipMemset(x, 0, n * 8));
  CHECK(hipMemset(y, 0, n * 8));
  const long n2 = n / 2;
  auto* x2 = reinterpret_cast<const double2v*>(x);
  auto* y2 = reinterpret_cast<const double2v*>(y);
  auto* z2 = reinterpret_cast<double2v*>(z);
  const int reps = 20;
#define RUN(KERN, NAME, GRID)                                                               \
  {                                                                                          \
    double ms = time_ms([&] { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(256), 0, 0, n2, 0.5, x2, y2, z2); }, reps); \
    printf("%-44s grid %6d  %.3f ms  %7.1f GB/s\n", NAME, GRID, ms, 24.0 * n / ms / 1e6);    \
  }
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    RUN((triad<1, false, false>), "triad stride u1 plain", grid);
    RUN((triad<2, false, false>), "triad stride u2 plain", grid);
    RUN((triad<4, false, false>), "triad stride u4 plain", grid);
    RUN((triad<1, false, true>), "triad stride u1 nt-store", grid);
    RUN((triad<4, false, true>), "triad stride u4 nt-store", grid);
    RUN((triad<1, true, true>), "triad stride u1 nt-load nt-store", grid);
    RUN((triad<2, true, true>), "triad stride u2 nt-load nt-store", grid);
    RUN((triad<4, true, true>), "triad stride u4 nt-load nt-store", grid);
    RUN((triad_blocked<4, false, false>), "triad blocked u4 plain", grid);
    RUN((triad_blocked<4, true, true>), "triad blocked u4 nt-load nt-store", grid);
  }
  {  // one double2 per thread, no loop
    const int grid = int((n2 + 255) / 256);
    RUN((triad<1, false, false>), "triad one-shot plain", grid);
    RUN((triad<1, true, true>), "triad one-shot nt", grid);
  }
  const long n4 = n / 2;   // float4 = 16 B
  for (int grid : {2048, 8192, int((n4 + 255) / 256)}) {
    double ms = time_ms([&] { hipLaunchKernelGGL((copy16<false, false>), dim3(grid), dim3(256), 0, 0, n4,
                                                 reinterpret_cast<const float4v*>(x), reinterpret_cast<float4v*>(z)); }, reps);
    printf("%-44s grid %6d  %.3f ms  %7.1f GB/s\n", "copy float4 plain", grid, ms, 16.0 * n / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((copy16<true, true>), dim3(grid), dim3(256), 0, 0, n4,
                                          reinterpret_cast<const float4v*>(x), reinterpret_cast<float4v*>(z)); }, reps);
    printf("%-44s grid %6d  %.3f ms  %7.1f GB/s\n", "copy float4 nt", grid, ms, 16.0 * n / ms / 1e6);
  }
  return 0;
}
