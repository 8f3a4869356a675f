// Store-only bandwidth ceiling of one MI355X: what a pure writer of 16-byte stores sustains, for the
// roofline of the covariance writer (predict_full_rows_kernel).  Not part of the library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
  } while (0)

// every workgroup streams contiguous `chunk`-byte pieces (grid-stride over pieces)
__global__ __launch_bounds__(256) void fill16(d2 *p, size_t n16, size_t piece16, double v) {
  const size_t npieces = (n16 + piece16 - 1) / piece16;
  for (size_t pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
    const size_t base = pc * piece16;
    const size_t end = (base + piece16 < n16) ? base + piece16 : n16;
    for (size_t i = base + threadIdx.x; i < end; i += 256) p[i] = d2{v, v + 1.0};
  }
}

// non-temporal variant
__global__ __launch_bounds__(256) void fill16_nt(d2 *p, size_t n16, size_t piece16, double v) {
  const size_t npieces = (n16 + piece16 - 1) / piece16;
  for (size_t pc = blockIdx.x; pc < npieces; pc += gridDim.x) {
    const size_t base = pc * piece16;
    const size_t end = (base + piece16 < n16) ? base + piece16 : n16;
    for (size_t i = base + threadIdx.x; i < end; i += 256) {
      __builtin_nontemporal_store(d2{v, v + 1.0}, &p[i]);
    }
  }
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB, the size of one 1024-sample covariance batch at F = 500
  d2 *p = nullptr;
  CK(hipMalloc((void **)&p, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t n16 = bytes / 16;
  auto time_it = [&](const char *name, auto launch) -> int {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    return 0;
  };
  time_it("hipMemsetAsync", [&] { (void)hipMemsetAsync(p, 0, bytes, 0); });
  for (int wgs : {256, 512, 1024, 2048, 4096, 16384}) {
    for (size_t piece : {(size_t)4096, (size_t)32768, (size_t)262144}) {
      char name[96];
      snprintf(name, sizeof name, "fill16 grid=%d piece=%zu B", wgs, piece);
      time_it(name, [&] { hipLaunchKernelGGL(fill16, dim3(wgs), dim3(256), 0, 0, p, n16, piece / 16, 1.0); });
    }
  }
  time_it("fill16_nt grid=2048 piece=32768 B",
          [&] { hipLaunchKernelGGL(fill16_nt, dim3(2048), dim3(256), 0, 0, p, n16, (size_t)32768 / 16, 1.0); });
  CK(hipFree(p));
  return 0;
}
