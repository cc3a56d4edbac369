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
// Upper bound of the grid of the element-wise (grid-stride) kernels.  Round 1 capped it at 2048 (8 resident
// workgroups per CU); measured in round 2 (profiles/r02_triad_variants.txt, r02_ab_grid_cap.txt): launches with
// far more, shorter workgroups reach a higher HBM rate on this chip (triad 5.5 -> 6.2 TB/s one-shot; block
// Jacobi + K1 -2 %), so the cap is now high enough that every per-iteration kernel is (nearly) one-shot.
#ifndef NSS_MAX_STREAM_BLOCKS
#define NSS_MAX_STREAM_BLOCKS 65536
#endif
constexpr int kMaxStreamBlocks = NSS_MAX_STREAM_BLOCKS;

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

// 16-byte accesses of two consecutive doubles
typedef double dbl2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store2_nt(double* p, const double2& v) {
  dbl2v t;
  t.x = v.x;
  t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<dbl2v*>(p));
}

// a * b rounded on its own: hipcc (-ffp-contract=fast) fuses a product into a following addition even across
// __dmul_rn / __dadd_rn (seen: one-ulp differences against the kernel that stages its products in LDS); the
// inline instruction cannot be fused
__device__ __forceinline__ double mul_unfused(double a, double b) {
  double p;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
  return p;
}

__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
// streaming form: for operands that are not read again before they are rewritten (kept out of the XCD L2s,
// which the SpMV kernels need for their operand windows)
__device__ __forceinline__ double2 ld2_nt(const double* p) {
  const dbl2v t = __builtin_nontemporal_load(reinterpret_cast<const dbl2v*>(p));
  return double2{t.x, t.y};
}
__device__ __forceinline__ void st2(double* p, const double2& v) { *reinterpret_cast<double2*>(p) = v; }

// Streaming (non-temporal) loads of vector operands that are not read again before they are rewritten pay when
// the vectors do not fit the caches anyway (1e7 DoF: BPCG v2 +11 %, MINRES +8 %) and cost where they do (1e6 DoF,
// 6-MB vectors: -5 %): the element-wise kernels take the choice as a template parameter, made per launch from the
// vector length.  Measured: off is better at 15 MB per vector (-4 % with them), on at 24 MB (+5 %, the 82-non-zero
// operator) and 30 MB (+2 %) (profiles/r02_ab_streaming_loads.txt).
#ifndef NSS_STREAM_LOADS_MIN_BYTES
#define NSS_STREAM_LOADS_MIN_BYTES (20u << 20)
#endif
int stream_loads_mode();                      // -1 automatic, 0 never, 1 always (nss_stream_loads_mode)
inline bool stream_vector_loads(int64_t n) {
  const int mode = stream_loads_mode();
  return mode < 0 ? n * int64_t(sizeof(double)) >= int64_t(NSS_STREAM_LOADS_MIN_BYTES) : mode != 0;
}
template <bool NT>
__device__ __forceinline__ double ld1s(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ double2 ld2s(const double* p) {
  if constexpr (NT) return ld2_nt(p);
  else return ld2(p);
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

// Fixed-order sum of pa[0 .. na) and pb[0 .. nb), evaluated by one 256-thread workgroup; the result
// is valid in every thread.  The tree: 1024 virtual lanes (thread t plays lanes t, t + 256, t + 512,
// t + 768), four strided accumulators per virtual lane, a 64-lane butterfly per virtual wave, the 16
// wave sums added in wave order, pa's total + pb's total.  Used both by the stand-alone sum kernels
// and by the kernels that fold the sum into their prologue, so the two plans give identical bits.
// `lds`: 32 doubles.  ~35 k partials (the 1e7-DoF case) are latency-bound: 9 dependent loads per chain.
// THREADS = 1024 (the stand-alone sum kernel: one virtual lane per thread, shortest dependent chain) gives
// the same bits as THREADS = 256 (inside a kernel's workgroup).
constexpr int kSumLanes = 1024;
struct SumPair { double a, b; };        // the totals of pa and of pb
template <int THREADS = kBlock>
__device__ __forceinline__ SumPair fixed_sums_1024(const double* __restrict__ pa, int na, const double* __restrict__ pb,
                                                   int nb, double* lds) {
  static_assert(THREADS % kWave == 0 && kSumLanes % THREADS == 0, "fixed_sum_1024: bad workgroup size");
  const int tid = threadIdx.x;
  __syncthreads();                      // protect lds reuse
  if (na <= 4 * kSumLanes && nb <= 4 * kSumLanes) {
    // Short sums (the launch-bound small systems): all loads are issued before the first add, so the
    // workgroup waits for ONE memory latency instead of one per loop trip.  Same values, same order as
    // the loops below: with at most four terms per virtual lane the strided loop runs once when all four
    // exist -- (x0 + x1) + (x2 + x3) -- and otherwise the tail loop adds them one after the other --
    // ((x0 + x1) + x2); absent terms are 0.0 and x + 0.0 == x.
    constexpr int kV = kSumLanes / THREADS;
    double xa[kV][4], xb[kV][4];
#pragma unroll
    for (int j = 0; j < kV; ++j) {
      const int v = tid + j * THREADS;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xa[j][q] = v + q * kSumLanes < na ? pa[v + q * kSumLanes] : 0.0;
        xb[j][q] = v + q * kSumLanes < nb ? pb[v + q * kSumLanes] : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < kV; ++j) {
      const int v = tid + j * THREADS;
      const double ra = v + 3 * kSumLanes < na ? (xa[j][0] + xa[j][1]) + (xa[j][2] + xa[j][3])
                                               : (xa[j][0] + xa[j][1]) + xa[j][2];
      const double rb = v + 3 * kSumLanes < nb ? (xb[j][0] + xb[j][1]) + (xb[j][2] + xb[j][3])
                                               : (xb[j][0] + xb[j][1]) + xb[j][2];
      const double sa = wave_sum(ra);
      const double sb = wave_sum(rb);
      if ((tid & (kWave - 1)) == 0) {
        lds[v >> 6] = sa;
        lds[kSumLanes / kWave + (v >> 6)] = sb;
      }
    }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int w = 0; w < kSumLanes / kWave; ++w) {
      ta += lds[w];
      tb += lds[kSumLanes / kWave + w];
    }
    return SumPair{ta, tb};
  }
#pragma unroll
  for (int j = 0; j < kSumLanes / THREADS; ++j) {
    const int v = tid + j * THREADS;    // virtual lane
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = v;
    for (; i + 3 * kSumLanes < na; i += 4 * kSumLanes) {
      a0 += pa[i];
      a1 += pa[i + kSumLanes];
      a2 += pa[i + 2 * kSumLanes];
      a3 += pa[i + 3 * kSumLanes];
    }
    for (; i < na; i += kSumLanes) a0 += pa[i];
    double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    i = v;
    for (; i + 3 * kSumLanes < nb; i += 4 * kSumLanes) {
      b0 += pb[i];
      b1 += pb[i + kSumLanes];
      b2 += pb[i + 2 * kSumLanes];
      b3 += pb[i + 3 * kSumLanes];
    }
    for (; i < nb; i += kSumLanes) b0 += pb[i];
    const double sa = wave_sum((a0 + a1) + (a2 + a3));
    const double sb = wave_sum((b0 + b1) + (b2 + b3));
    if ((tid & (kWave - 1)) == 0) {
      const int vwave = v >> 6;         // virtual wave = real wave + j * THREADS / 64: same 64 lanes
      lds[vwave] = sa;
      lds[kSumLanes / kWave + vwave] = sb;
    }
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
#pragma unroll
  for (int w = 0; w < kSumLanes / kWave; ++w) {
    ta += lds[w];
    tb += lds[kSumLanes / kWave + w];
  }
  return SumPair{ta, tb};
}

template <int THREADS = kBlock>
__device__ __forceinline__ double fixed_sum_1024(const double* __restrict__ pa, int na, const double* __restrict__ pb,
                                                 int nb, double* lds) {
  const SumPair s = fixed_sums_1024<THREADS>(pa, na, pb, nb, lds);
  return s.a + s.b;
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
