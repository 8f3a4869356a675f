// Probe: what clock64() counts on gfx950 and how long a v_mfma_f64_16x16x4_f64 occupies a SIMD, on an otherwise idle chip
// (1 workgroup) and on a busy one (1024 workgroups): ticks of clock64 and of wall_clock64 (100 MHz) around a chain of
// independent MFMAs.   usage: clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(long long *out, double *sink, int n) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x + 1.0, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y + 1.0, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x + 2.0, y + 3.0, a3, 0, 0, 0);
  }
  const d4 s = a0 + a1 + a2 + a3;
  const long long c1 = clock64(), w1 = wall_clock64();
  if (s[0] == 123.456) sink[0] = s[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int main() {
  long long *d, h[2];
  double *sink;
  (void)hipMalloc(&d, 16); (void)hipMalloc(&sink, 8);
  for (int blocks : {1, 1, 1024, 1024, 1}) {
    for (int threads : {64, 512}) {
      const int n = 2000;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, sink, n);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%4d workgroups x %3d threads: %d x 4 MFMAs per wave: %lld clock64 ticks (%.1f per MFMA), %lld wall ticks of 10 ns -> clock64 runs at %.0f MHz; one MFMA per wave every %.1f ns\n",
             blocks, threads, n, h[0], (double)h[0] / (4.0 * n), h[1], (double)h[0] / (h[1] * 1e-2), h[1] * 10.0 / (4.0 * n));
    }
  }
  return 0;
}
