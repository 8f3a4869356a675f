// Which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) give a kernel, and what do two streams on
// complementary masks cost each other?  (profiles/r03_cu_mask_probe.txt)
//
//   1. bit -> (XCD, CU) mapping: a grid of one-wave workgroups records HW_REG_XCC_ID and HW_REG_HW_ID under masks
//      with the first 8 n bits set; the histogram shows whether bit i lands on XCD i % 8 (n CUs on every XCD).
//   2. a store-only kernel (the covariance writer's access pattern: 32 000-byte contiguous pieces) on 25 / 37.5 / 50 /
//      100 % of the CUs: how many CUs does an HBM-write-bound kernel need?
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__global__ void where_kernel(unsigned *out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // keep the workgroup alive a little so the grid spreads over every enabled CU
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc & 0xF;
    out[2 * blockIdx.x + 1] = hw;
  }
}

typedef double d2 __attribute__((ext_vector_type(2)));
// every workgroup stores `pieces` contiguous pieces of 4000 doubles (8 rows of a 500 x 500 covariance)
__global__ __launch_bounds__(256) void store_kernel(double *out, long long npieces) {
  for (long long pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
    double *dst = out + pc * 4000;
    for (int e = threadIdx.x * 2; e < 4000; e += 512) *reinterpret_cast<d2 *>(dst + e) = d2{1.0 + e, 2.0 + pc};
  }
}

// the matrix-core writer's pattern: a 512-thread workgroup owns 16 rows x 500 columns of a sample (64 000 contiguous
// bytes); PIECE = bytes a wave-instruction writes contiguously in one row: 256 (4 rows x 16 lanes, the MFMA accumulator
// layout), 512 (2 rows x 32 lanes) or 1024 (1 row x 64 lanes)
template <int PIECE>
__global__ __launch_bounds__(512) void store_tile_kernel(double *out, long long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = PIECE / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per instruction
  const int lr = lane / LPR, lc = lane % LPR;
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    double *dst = out + t * 8000;          // 16 rows x 500 doubles
    if (PIECE == 1024) {
      // wave w: rows 2w, 2w + 1, four instructions of 1 KiB each per row
      for (int r = 2 * wave; r < 2 * wave + 2; ++r)
        for (int c0 = 0; c0 < 500; c0 += 128) {
          const int g = c0 + 2 * lane;
          if (g < 500) *reinterpret_cast<d2 *>(dst + r * 500 + g) = d2{1.0 + g, 2.0 + t};
        }
    } else {
      // wave w: columns 64 w .. 64 w + 63 of all 16 rows
      for (int cb = 0; cb < 64 * 8 / PIECE; ++cb)
        for (int i = 0; i < 16 / RPI; ++i) {
          const int r = lr + RPI * i, g = 64 * wave + cb * (PIECE / 8) + 2 * lc;
          if (g < 500) *reinterpret_cast<d2 *>(dst + r * 500 + g) = d2{1.0 + g, 2.0 + t};
        }
    }
  }
}

static hipStream_t masked_stream(int first_bit, int last_bit /* exclusive */) {
  unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = first_bit; i < last_bit; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
  return s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const int nwg = 4096;
  unsigned *dout;
  CK(hipMalloc(&dout, sizeof(unsigned) * 2 * nwg));
  std::vector<unsigned> h(2 * nwg);
  for (int n : {1, 8, 24}) {
    hipStream_t s = masked_stream(0, 8 * n);
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, s, dout);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), dout, sizeof(unsigned) * 2 * nwg, hipMemcpyDeviceToHost));
    std::map<unsigned, std::map<unsigned, int>> per;   // xcc -> (se, cu) -> count
    for (int i = 0; i < nwg; ++i) {
      const unsigned hw = h[2 * i + 1];
      const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
      per[h[2 * i]][(se << 8) | (sh << 4) | cu] += 1;
    }
    printf("mask = first %3d bits: ", 8 * n);
    for (auto &x : per) printf("xcc%u:%zu ", x.first, x.second.size());
    printf(" (distinct CUs per XCC)\n");
    CK(hipStreamDestroy(s));
  }
  // 2. store bandwidth against the share of CUs
  const long long npieces = 64LL * 1024;      // 2.1 GB
  double *big;
  CK(hipMalloc(&big, sizeof(double) * 4000 * npieces));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int n : {4, 6, 8, 12, 16, 24, 32}) {
    hipStream_t s = masked_stream(256 - 8 * n, 256);
    for (int wgs_per_cu : {4, 8}) {
      const int grid = 8 * n * wgs_per_cu;
      hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, s, big, npieces);
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(store_kernel, dim3(grid), dim3(256), 0, s, big, npieces);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("store-only, %2d CUs per XCD (%3d CUs), %d workgroups per CU: %.3f ms per 2.1 GB = %.2f TB/s\n", n, 8 * n,
             wgs_per_cu, ms / 3, 8.0 * 4000 * npieces / (ms / 3 * 1e-3) / 1e12);
    }
    CK(hipStreamDestroy(s));
  }
  // 3. the same bytes in the matrix-core writer's tile shape, by contiguous piece per wave-instruction
  const long long ntiles = npieces / 2;
  for (int n : {8, 12, 32}) {
    hipStream_t s = masked_stream(256 - 8 * n, 256);
    for (int piece : {256, 512, 1024}) {
      const int grid = 8 * n * 2;
      auto launch = [&]() {
        if (piece == 256) hipLaunchKernelGGL(store_tile_kernel<256>, dim3(grid), dim3(512), 0, s, big, ntiles);
        else if (piece == 512) hipLaunchKernelGGL(store_tile_kernel<512>, dim3(grid), dim3(512), 0, s, big, ntiles);
        else hipLaunchKernelGGL(store_tile_kernel<1024>, dim3(grid), dim3(512), 0, s, big, ntiles);
      };
      launch();
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 3; ++r) launch();
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("tile stores, %2d CUs per XCD, %4d contiguous bytes per row and wave-instruction: %.3f ms per 2.1 GB = %.2f TB/s\n",
             n, piece, ms / 3, 8.0 * 8000 * ntiles / (ms / 3 * 1e-3) / 1e12);
    }
    CK(hipStreamDestroy(s));
  }
  return 0;
}
