// Micro-benchmark behind a design question of csr_stream.h: does the matrix stream of the CSR-stream kernel (per lane
// eight 8-byte value loads + eight 2-byte index loads at a stride of 256 entries) read HBM slower than the same bytes
// taken as four 16-byte + four 4-byte loads of two consecutive entries?  One workgroup of 256 lanes per 2048 entries,
// products to LDS, barrier, 256 partial sums back -- the phase structure and occupancy (16 KiB of LDS) of the real
// kernel without its operand gather.   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/probes/stream_width_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kBlock = 256, kChunk = 2048;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, bool GATHER>   // 0: 8 x (b64 + u16);  1: 4 x (b128 + u32);  2: as 0 with non-temporal loads;  3: as 1 non-temporal
__global__ __launch_bounds__(kBlock) void probe(const double* __restrict__ val, const uint16_t* __restrict__ idx,
                                                 const double* __restrict__ x, double* __restrict__ out, int nblk) {
  __shared__ double prod[kChunk];
  const int tid = threadIdx.x;
  const int b = (blockIdx.x & 7) * ((nblk + 7) / 8) + (blockIdx.x >> 3);
  if (b >= nblk) return;
  const size_t p0 = size_t(b) * kChunk;
  if (MODE == 0 || MODE == 2) {
    double v[8]; uint16_t c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = MODE == 2 ? __builtin_nontemporal_load(val + p0 + tid + j * kBlock) : val[p0 + tid + j * kBlock];
      c[j] = MODE == 2 ? __builtin_nontemporal_load(idx + p0 + tid + j * kBlock) : idx[p0 + tid + j * kBlock];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) prod[tid + j * kBlock] = v[j] * (GATHER ? x[c[j]] : double(c[j]));
  } else {
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    dbl2 v[4]; uint32_t c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const dbl2* pv = reinterpret_cast<const dbl2*>(val + p0) + tid + j * kBlock;
      const uint32_t* pc = reinterpret_cast<const uint32_t*>(idx + p0) + tid + j * kBlock;
      v[j] = MODE == 3 ? __builtin_nontemporal_load(pv) : *pv;
      c[j] = MODE == 3 ? __builtin_nontemporal_load(pc) : *pc;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dbl2 p;
      p.x = v[j].x * (GATHER ? x[c[j] & 0xffff] : double(c[j] & 0xffff));
      p.y = v[j].y * (GATHER ? x[c[j] >> 16] : double(c[j] >> 16));
      reinterpret_cast<dbl2*>(prod)[tid + j * kBlock] = p;
    }
  }
  __syncthreads();
  double s = 0.0;
  for (int j = 0; j < 8; ++j) s += prod[(tid * 8 + j * 257) & (kChunk - 1)];   // "rows" of 8 entries, conflict-free
  out[size_t(b) * kBlock + tid] = s;
}

int main() {
  const int nblk = 70000;                                        // 70000 x 2048 entries: 1.15 GB of values + 0.29 GB of indices
  const size_t n = size_t(nblk) * kChunk;
  double *val, *x, *out; uint16_t* idx;
  CHECK(hipMalloc(&val, n * 8)); CHECK(hipMalloc(&idx, n * 2)); CHECK(hipMalloc(&x, 65536 * 8)); CHECK(hipMalloc(&out, size_t(nblk) * kBlock * 8));
  CHECK(hipMemset(val, 0, n * 8)); CHECK(hipMemset(idx, 0, n * 2)); CHECK(hipMemset(x, 0, 65536 * 8));
  { std::vector<uint16_t> h(n); for (size_t i = 0; i < n; ++i) h[i] = uint16_t((i * 7) & 4095); CHECK(hipMemcpy(idx, h.data(), n * 2, hipMemcpyHostToDevice)); }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = double(n) * 10.0 + double(nblk) * kBlock * 8.0;
  const int grid = ((nblk + 7) / 8) * 8;
  for (int rep = 0; rep < 2; ++rep)
   for (int g = 0; g < 2; ++g)
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int t = 0; t < 6; ++t) {
        CHECK(hipEventRecord(e0));
        if (mode == 0 && g) probe<0, true><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 0 && !g) probe<0, false><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 1 && g) probe<1, true><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 1 && !g) probe<1, false><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 2 && g) probe<2, true><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 2 && !g) probe<2, false><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 3 && g) probe<3, true><<<grid, kBlock>>>(val, idx, x, out, nblk);
        if (mode == 3 && !g) probe<3, false><<<grid, kBlock>>>(val, idx, x, out, nblk);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (t > 0 && ms < best) best = ms;
      }
      const char* names[4] = {"8 x (8 B + 2 B) per lane", "4 x (16 B + 4 B) per lane", "8 x (8 B + 2 B), non-temporal", "4 x (16 B + 4 B), non-temporal"};
      printf("round %d  %s  %-32s %.3f ms  %.2f TB/s\n", rep, g ? "operand gathered (L1 / L2 hits)" : "no operand                     ", names[mode], best, bytes / best / 1e9);
    }
  return 0;
}
