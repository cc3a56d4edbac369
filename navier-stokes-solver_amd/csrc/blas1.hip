// BLAS-1 kernels of the Krylov hot path (fp64, HBM-bound): fill / copy / scal /
// lincomb (<= 4 terms) / deterministic dot / STREAM triad / diagonal scale.
//
// Every kernel is a grid-stride loop over 16-byte (double2) accesses, 256-thread
// workgroups (one wave per SIMD), at most 2048 workgroups (8 per CU) -- the
// element-wise recipe of cdna_hip_programming.md Appendix B.  Unaligned views fall
// back to 8-byte accesses.
#include "nss_common.h"

#include <cstring>

namespace nss {

int g_stream_loads_mode = -1;
int stream_loads_mode() { return g_stream_loads_mode; }


static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof g_error, fmt, ap);
  va_end(ap);
}

Scratch& scratch() {
  static Scratch s;
  if (!s.partials) {
    NSS_HIP(hipMalloc(&s.partials, sizeof(double) * Scratch::kMaxPartials));
    NSS_HIP(hipMalloc(&s.result, sizeof(double) * 8));
    NSS_HIP(hipHostMalloc(&s.host, sizeof(double) * 8, hipHostMallocDefault));
  }
  return s;
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void reciprocal_kernel(int64_t n, const double* __restrict__ x,
                                                             double* __restrict__ y) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) y[i] = 1.0 / x[i];
}

template <bool VEC2>
__global__ __launch_bounds__(kBlock) void fill_kernel(int64_t n, double value, double* __restrict__ x) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (VEC2) {
    const int64_t n2 = n >> 1;
    double2* x2 = reinterpret_cast<double2*>(x);
    for (; i < n2; i += stride) x2[i] = make_double2(value, value);
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = value;
  } else {
    for (; i < n; i += stride) x[i] = value;
  }
}

struct LinArgs {
  const double* x[4];
  double c[4];
};

template <int NT, bool VEC2>
__global__ __launch_bounds__(kBlock) void lincomb_kernel(int64_t n, LinArgs a, double* y) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (VEC2) {
    const int64_t n2 = n >> 1;
    for (; i < n2; i += stride) {
      double2 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) v[t] = reinterpret_cast<const double2*>(a.x[t])[i];
      double2 r = make_double2(a.c[0] * v[0].x, a.c[0] * v[0].y);
#pragma unroll
      for (int t = 1; t < NT; ++t) {
        r.x = fma(a.c[t], v[t].x, r.x);
        r.y = fma(a.c[t], v[t].y, r.y);
      }
      reinterpret_cast<double2*>(y)[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      double r = a.c[0] * a.x[0][n - 1];
#pragma unroll
      for (int t = 1; t < NT; ++t) r = fma(a.c[t], a.x[t][n - 1], r);
      y[n - 1] = r;
    }
  } else {
    for (; i < n; i += stride) {
      double r = a.c[0] * a.x[0][i];
#pragma unroll
      for (int t = 1; t < NT; ++t) r = fma(a.c[t], a.x[t][i], r);
      y[i] = r;
    }
  }
}

// STREAM triad z = x + a y, the roofline denominator.  One-shot launch (one 16-byte access per lane and
// stream, no grid-stride loop) with non-temporal loads and stores: the fastest of the forms swept in
// tools/triad_variants.hip (6.2 TB/s; <= 2048 striding workgroups: 5.5 TB/s).
template <bool VEC2>
__global__ __launch_bounds__(kBlock) void triad_kernel(int64_t n, double a, const double* __restrict__ x,
                                                        const double* __restrict__ y, double* __restrict__ z) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (VEC2) {
    const int64_t n2 = n >> 1;
    if (i < n2) {
      const dbl2v xv = __builtin_nontemporal_load(reinterpret_cast<const dbl2v*>(x) + i);
      const dbl2v yv = __builtin_nontemporal_load(reinterpret_cast<const dbl2v*>(y) + i);
      dbl2v r;
      r.x = fma(a, yv.x, xv.x);
      r.y = fma(a, yv.y, xv.y);
      __builtin_nontemporal_store(r, reinterpret_cast<dbl2v*>(z) + i);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) z[n - 1] = fma(a, y[n - 1], x[n - 1]);
  } else {
    if (i < n) z[i] = fma(a, y[i], x[i]);
  }
}

// y = alpha * d .* x + beta * y
template <bool VEC2, bool BETA>
__global__ __launch_bounds__(kBlock) void diag_kernel(int64_t n, const double* __restrict__ d, double alpha,
                                                       const double* x, double beta, double* y,
                                                       const int32_t* __restrict__ done) {
  if (done != nullptr && *done != 0) return;
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (VEC2) {
    const int64_t n2 = n >> 1;
    for (; i < n2; i += stride) {
      const double2 dv = reinterpret_cast<const double2*>(d)[i];
      const double2 xv = reinterpret_cast<const double2*>(x)[i];
      double2 r = make_double2(alpha * (dv.x * xv.x), alpha * (dv.y * xv.y));
      if (BETA) {
        const double2 yv = reinterpret_cast<const double2*>(y)[i];
        r.x = fma(beta, yv.x, r.x);
        r.y = fma(beta, yv.y, r.y);
      }
      reinterpret_cast<double2*>(y)[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      double r = alpha * (d[n - 1] * x[n - 1]);
      if (BETA) r = fma(beta, y[n - 1], r);
      y[n - 1] = r;
    }
  } else {
    for (; i < n; i += stride) {
      double r = alpha * (d[i] * x[i]);
      if (BETA) r = fma(beta, y[i], r);
      y[i] = r;
    }
  }
}

// donor-cell flux of the explicit convection term: f = adv * avg - 1/2 |adv| * diff
__global__ __launch_bounds__(kBlock) void upwind_flux_kernel(int64_t n, const double* __restrict__ adv,
                                                              const double* __restrict__ avg,
                                                              const double* __restrict__ diff, double* __restrict__ f) {
  const int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) {
    const double a = adv[i];
    f[i] = fma(a, avg[i], -0.5 * (fabs(a) * diff[i]));
  }
}

__global__ __launch_bounds__(kBlock) void gather_kernel(int64_t n, const int32_t* __restrict__ idx,
                                                         const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t stride = int64_t(gridDim.x) * kBlock;
  for (int64_t i = int64_t(blockIdx.x) * kBlock + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}

// y = alpha * d .* x + beta * y; no-op on the device once *done != 0 (done may be NULL)
void diag_apply(int64_t n, const double* d, double alpha, const double* x, double beta, double* y, const int32_t* done,
                hipStream_t st) {
  if (n <= 0) return;
  const int grid = stream_grid(n, kBlock * 4);
  const bool vec = aligned16(d) && aligned16(x) && aligned16(y);
  if (beta == 0.0) {
    if (vec) hipLaunchKernelGGL((diag_kernel<true, false>), dim3(grid), dim3(kBlock), 0, st, n, d, alpha, x, beta, y, done);
    else hipLaunchKernelGGL((diag_kernel<false, false>), dim3(grid), dim3(kBlock), 0, st, n, d, alpha, x, beta, y, done);
  } else {
    if (vec) hipLaunchKernelGGL((diag_kernel<true, true>), dim3(grid), dim3(kBlock), 0, st, n, d, alpha, x, beta, y, done);
    else hipLaunchKernelGGL((diag_kernel<false, true>), dim3(grid), dim3(kBlock), 0, st, n, d, alpha, x, beta, y, done);
  }
  NSS_CHECK_LAUNCH();
}

void gather_launch(int64_t n, const int32_t* idx, const double* src, double* dst, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, st, n, idx, src, dst);
  NSS_CHECK_LAUNCH();
}

// ---- dot: stage 1 = per-workgroup partial sums of up to 4 vector pairs ----------------
struct DotArgs {
  const double* x[4];
  const double* y[4];
  int64_t n[4];
  int32_t first_block[5];  // blocks [first_block[p], first_block[p+1]) work on pair p
  int32_t vec2[4];
};

__global__ __launch_bounds__(kBlock) void dot_partial_kernel(DotArgs a, int npairs, double* __restrict__ partials) {
  __shared__ double lds[kBlock / kWave];
  int p = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (q < npairs && int(blockIdx.x) >= a.first_block[q]) p = q;
  const int nb = a.first_block[p + 1] - a.first_block[p];
  const int b = blockIdx.x - a.first_block[p];
  const double* __restrict__ x = a.x[p];
  const double* __restrict__ y = a.y[p];
  const int64_t n = a.n[p];
  const int64_t stride = int64_t(nb) * kBlock;
  int64_t i = int64_t(b) * kBlock + threadIdx.x;
  double acc0 = 0.0, acc1 = 0.0;
  if (a.vec2[p]) {
    const int64_t n2 = n >> 1;
    for (; i < n2; i += stride) {
      const double2 xv = reinterpret_cast<const double2*>(x)[i];
      const double2 yv = reinterpret_cast<const double2*>(y)[i];
      acc0 = fma(xv.x, yv.x, acc0);
      acc1 = fma(xv.y, yv.y, acc1);
    }
    if ((n & 1) && b == 0 && threadIdx.x == 0) acc0 = fma(x[n - 1], y[n - 1], acc0);
  } else {
    for (; i < n; i += stride) acc0 = fma(x[i], y[i], acc0);
  }
  const double s = block_sum(acc0 + acc1, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// stage 2: one workgroup sums the partials in a fixed order
__global__ __launch_bounds__(kBlock) void dot_final_kernel(int count, const double* __restrict__ partials,
                                                            double* __restrict__ result) {
  __shared__ double lds[kBlock / kWave];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) acc += partials[i];
  const double s = block_sum(acc, lds);
  if (threadIdx.x == 0) result[0] = s;
}

static void launch_dot(int32_t npairs, const int64_t* h_n, const double* const* h_x, const double* const* h_y,
                       double* result_dev, hipStream_t st) {
  NSS_REQUIRE(npairs >= 1 && npairs <= 4, "dot: 1 <= npairs <= 4");
  DotArgs a{};
  int total = 0;
  for (int p = 0; p < npairs; ++p) {
    NSS_REQUIRE(h_n[p] >= 0, "dot: negative length");
    a.x[p] = h_x[p];
    a.y[p] = h_y[p];
    a.n[p] = h_n[p];
    a.vec2[p] = aligned16(h_x[p]) && aligned16(h_y[p]);
    a.first_block[p] = total;
    int g = stream_grid(h_n[p], kBlock * 8);
    if (g > kMaxStreamBlocks / 2) g = kMaxStreamBlocks / 2;
    total += g;
  }
  a.first_block[npairs] = total;
  for (int p = npairs + 1; p < 5; ++p) a.first_block[p] = total;
  Scratch& s = scratch();
  NSS_REQUIRE(total <= Scratch::kMaxPartials, "dot: too many partials");
  hipLaunchKernelGGL(dot_partial_kernel, dim3(total), dim3(kBlock), 0, st, a, npairs, s.partials);
  NSS_CHECK_LAUNCH();
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kBlock), 0, st, total, s.partials, result_dev);
  NSS_CHECK_LAUNCH();
}

}  // namespace nss

using namespace nss;

extern "C" {

int nss_abi_version(void) { return NSS_ABI_VERSION; }

const char* nss_last_error(void) { return g_error; }

int nss_stream_loads_mode(int32_t mode) {
  return guarded([&] {
    NSS_REQUIRE(mode >= -1 && mode <= 1, "stream_loads_mode: -1 (automatic), 0 (never) or 1 (always)");
    nss::g_stream_loads_mode = mode;
  });
}

int nss_device_info(int32_t* cu_count, int64_t* hbm_bytes, int32_t* wavefront_size, char* name, int32_t name_cap) {
  return guarded([&] {
    int dev = 0;
    NSS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    NSS_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
    if (wavefront_size) *wavefront_size = prop.warpSize;
    if (name && name_cap > 0) {
      strncpy(name, prop.gcnArchName, name_cap - 1);
      name[name_cap - 1] = 0;
    }
  });
}

int nss_stream_synchronize(nss_stream_t stream) {
  return guarded([&] { NSS_HIP(hipStreamSynchronize(as_stream(stream))); });
}

int nss_fill_f64(int64_t n, double value, double* x, nss_stream_t stream) {
  return guarded([&] {
    if (n <= 0) return;
    const int grid = stream_grid(n, kBlock * 4);
    if (aligned16(x))
      hipLaunchKernelGGL(fill_kernel<true>, dim3(grid), dim3(kBlock), 0, as_stream(stream), n, value, x);
    else
      hipLaunchKernelGGL(fill_kernel<false>, dim3(grid), dim3(kBlock), 0, as_stream(stream), n, value, x);
    NSS_CHECK_LAUNCH();
  });
}

int nss_reciprocal_f64(int64_t n, const double* x, double* y, nss_stream_t stream) {
  return guarded([&] {
    if (n <= 0) return;
    hipLaunchKernelGGL(reciprocal_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), n, x, y);
    NSS_CHECK_LAUNCH();
  });
}

int nss_lincomb_f64(int64_t n, int32_t nterms, const double* h_coeff, const double* const* h_x, double* y,
                    nss_stream_t stream) {
  return guarded([&] {
    NSS_REQUIRE(nterms >= 1 && nterms <= 4, "lincomb: 1 <= nterms <= 4");
    if (n <= 0) return;
    LinArgs a{};
    bool vec = aligned16(y);
    for (int t = 0; t < nterms; ++t) {
      a.x[t] = h_x[t];
      a.c[t] = h_coeff[t];
      vec = vec && aligned16(h_x[t]);
    }
    const int grid = stream_grid(n, kBlock * 4);
    hipStream_t st = as_stream(stream);
#define NSS_LAUNCH_LC(NT)                                                                      \
  if (vec)                                                                                     \
    hipLaunchKernelGGL((lincomb_kernel<NT, true>), dim3(grid), dim3(kBlock), 0, st, n, a, y);  \
  else                                                                                         \
    hipLaunchKernelGGL((lincomb_kernel<NT, false>), dim3(grid), dim3(kBlock), 0, st, n, a, y);
    switch (nterms) {
      case 1: NSS_LAUNCH_LC(1) break;
      case 2: NSS_LAUNCH_LC(2) break;
      case 3: NSS_LAUNCH_LC(3) break;
      default: NSS_LAUNCH_LC(4) break;
    }
#undef NSS_LAUNCH_LC
    NSS_CHECK_LAUNCH();
  });
}

int nss_copy_f64(int64_t n, const double* x, double* y, nss_stream_t stream) {
  const double one = 1.0;
  const double* xs[1] = {x};
  return nss_lincomb_f64(n, 1, &one, xs, y, stream);
}

int nss_scal_f64(int64_t n, double a, double* x, nss_stream_t stream) {
  const double* xs[1] = {x};
  return nss_lincomb_f64(n, 1, &a, xs, x, stream);
}

int nss_dot_f64(int32_t npairs, const int64_t* h_n, const double* const* h_x, const double* const* h_y,
                double* result_dev, nss_stream_t stream) {
  return guarded([&] { launch_dot(npairs, h_n, h_x, h_y, result_dev, as_stream(stream)); });
}

int nss_dot_host_f64(int32_t npairs, const int64_t* h_n, const double* const* h_x, const double* const* h_y,
                     double* h_result, nss_stream_t stream) {
  return guarded([&] {
    Scratch& s = scratch();
    hipStream_t st = as_stream(stream);
    launch_dot(npairs, h_n, h_x, h_y, s.result, st);
    NSS_HIP(hipMemcpyAsync(s.host, s.result, sizeof(double), hipMemcpyDeviceToHost, st));
    NSS_HIP(hipStreamSynchronize(st));
    *h_result = s.host[0];
  });
}

int nss_gather_f64(int64_t n, const int32_t* idx, const double* src, double* dst, nss_stream_t stream) {
  return guarded([&] {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), n, idx, src,
                       dst);
    NSS_CHECK_LAUNCH();
  });
}

int nss_upwind_flux_f64(int64_t n, const double* adv, const double* avg, const double* diff, double* flux,
                        nss_stream_t stream) {
  return guarded([&] {
    if (n <= 0) return;
    NSS_REQUIRE(adv && avg && diff && flux, "upwind_flux: NULL argument");
    hipLaunchKernelGGL(upwind_flux_kernel, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       as_stream(stream), n, adv, avg, diff, flux);
    NSS_CHECK_LAUNCH();
  });
}

int nss_stream_triad_f64(int64_t n, double a, const double* x, const double* y, double* z, nss_stream_t stream) {
  return guarded([&] {
    if (n <= 0) return;
    if (aligned16(x) && aligned16(y) && aligned16(z))
      hipLaunchKernelGGL(triad_kernel<true>, dim3(unsigned(((n >> 1) + kBlock) / kBlock)), dim3(kBlock), 0,
                         as_stream(stream), n, a, x, y, z);
    else
      hipLaunchKernelGGL(triad_kernel<false>, dim3(unsigned((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                         as_stream(stream), n, a, x, y, z);
    NSS_CHECK_LAUNCH();
  });
}

int nss_diag_apply_f64(int64_t n, const double* d, double alpha, const double* x, double beta, double* y,
                       nss_stream_t stream) {
  return guarded([&] { diag_apply(n, d, alpha, x, beta, y, nullptr, as_stream(stream)); });
}

}  // extern "C"
