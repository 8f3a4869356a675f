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

// the covariance writer's pattern: workgroup = 8 rows x 500 doubles (32,000 B contiguous) of one 500 x 500 sample,
// thread t stores 16 B at row * 4000 + 16 t (t < 250), i.e. rows start 32 B off the 128-byte lines;
// flat = 1: the same 32,000 bytes written as a flat aligned stream instead
__global__ __launch_bounds__(256) void rows500(double *p, int nsamples, int flat, double v) {
  // grid: (63 row blocks, nsamples / 8); each workgroup walks 8 samples like predict_full_rows_kernel
  const int rb = blockIdx.x;
  for (int ib = 0; ib < 8; ++ib) {
    const size_t sample = (size_t)blockIdx.y * 8 + ib;
    double *base = p + sample * 250000 + (size_t)rb * 4000;
    const int rows = (rb == 62) ? 4 : 8;
    if (flat) {
      for (int e = 2 * threadIdx.x; e < rows * 500; e += 512) *reinterpret_cast<d2 *>(base + e) = d2{v, v + 1.0};
    } else {
      if (threadIdx.x < 250)
        for (int r = 0; r < rows; ++r) *reinterpret_cast<d2 *>(base + r * 500 + 2 * threadIdx.x) = d2{v, v + 1.0};
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
  {
    const int ns = 1024;   // 1024 x 500 x 500 doubles = 2.048e9 B
    auto t2 = [&](const char *name, int flat) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rows500, dim3(63, ns / 8), dim3(256), 0, 0, (double *)p, ns, flat, 1.0);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rows500, dim3(63, ns / 8), dim3(256), 0, 0, (double *)p, ns, flat, 1.0);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 10, 2.048e9 / (ms / 10 * 1e-3) / 1e9);
    };
    t2("rows500 (writer pattern, 32 B-offset rows)", 0);
    t2("rows500 flat (aligned 1 KiB wave stores)", 1);
  }
  CK(hipFree(p));
  return 0;
}
