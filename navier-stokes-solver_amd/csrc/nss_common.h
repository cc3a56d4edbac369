// Shared host/device helpers of libnsskrylov (gfx950 only: wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <stdexcept>
#include <string>

#include "../../include/nss_krylov.h"

// Non-temporal stores for vectors that are written once and next read by a later kernel: they do not
// allocate in the 4-MiB XCD L2s, which the SpMV kernels need for the gathered x window.  Levels
// (A/B on one box, 1e7 DoF): 1 = t2 / q / z0 / u0 / d0 / w0 (K2 0.183 -> 0.157 ms), 2 = + t4 / t3
// (K2 0.146 ms, iteration +8 % over level 0), 3 = + s0 / t0 (no further change).  Non-temporal stores of the
// block-Jacobi output (scattered 24-byte pieces) and of the plain SpMV result were slower (K1 + block
// Jacobi 0.217 -> 0.255 ms) and are not used.
#ifndef NSS_NT_STORE
#define NSS_NT_STORE 2
#endif
#define NSS_NT_(ptr, v) __builtin_nontemporal_store((v), &(ptr))
#define NSS_PLAIN_(ptr, v) ((ptr) = (v))
#if NSS_NT_STORE >= 1
#define NSS_ST(ptr, v) NSS_NT_(ptr, v)
#else
#define NSS_ST(ptr, v) NSS_PLAIN_(ptr, v)
#endif
#if NSS_NT_STORE >= 2
#define NSS_ST2(ptr, v) NSS_NT_(ptr, v)
#else
#define NSS_ST2(ptr, v) NSS_PLAIN_(ptr, v)
#endif
#if NSS_NT_STORE >= 3
#define NSS_ST3(ptr, v) NSS_NT_(ptr, v)
#else
#define NSS_ST3(ptr, v) NSS_PLAIN_(ptr, v)
#endif

namespace nss {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // 4 waves: one per SIMD of a CU
#ifndef NSS_MAX_STREAM_BLOCKS
#define NSS_MAX_STREAM_BLOCKS 2048
#endif
constexpr int kMaxStreamBlocks = NSS_MAX_STREAM_BLOCKS;  // 256 CUs x 8 resident blocks: grid-stride the rest

void set_error(const char* fmt, ...);

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

[[noreturn]] inline void fail(const char* what, const char* file, int line, hipError_t e) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
  throw Error(buf);
}

#define NSS_HIP(expr)                                            \
  do {                                                           \
    hipError_t nss_e_ = (expr);                                  \
    if (nss_e_ != hipSuccess) ::nss::fail(#expr, __FILE__, __LINE__, nss_e_); \
  } while (0)

#define NSS_CHECK_LAUNCH() NSS_HIP(hipGetLastError())

#define NSS_REQUIRE(cond, msg)                   \
  do {                                           \
    if (!(cond)) throw ::nss::Error(std::string("invalid argument: ") + (msg)); \
  } while (0)

// Run `body` (a lambda) and translate exceptions into the C error convention.
template <class F>
inline int guarded(F&& body) noexcept {
  try {
    body();
    return 0;
  } catch (const std::exception& e) {
    set_error("%s", e.what());
    return 1;
  } catch (...) {
    set_error("unknown C++ exception");
    return 2;
  }
}

inline hipStream_t as_stream(nss_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int stream_grid(int64_t work_items, int items_per_block) {
  int64_t g = (work_items + items_per_block - 1) / items_per_block;
  if (g < 1) g = 1;
  if (g > kMaxStreamBlocks) g = kMaxStreamBlocks;
  return static_cast<int>(g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device-side reductions -------------------------------------------------------
// Sum over the 64 lanes of a wave with a fixed butterfly (deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// Sum over a kBlock-thread workgroup; result valid in every thread.  `lds` holds >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* lds) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // protect lds reuse across consecutive calls
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double t = lds[0];
#pragma unroll
  for (int w = 1; w < kBlock / kWave; ++w) t += lds[w];
  return t;
}

// Library-wide scratch for reductions: partial sums (device) + one pinned host slot.
struct Scratch {
  double* partials = nullptr;   // kMaxPartials doubles
  double* result = nullptr;     // 8 doubles (device)
  double* host = nullptr;       // 8 doubles (pinned)
  static constexpr int kMaxPartials = 4 * kMaxStreamBlocks;
};
Scratch& scratch();

// y = alpha * d .* x + beta * y (blas1.hip); returns at once on the device when *done != 0 (NULL: never)
void diag_apply(int64_t n, const double* d, double alpha, const double* x, double beta, double* y, const int32_t* done,
                hipStream_t st);

}  // namespace nss
