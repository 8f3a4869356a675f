// Probe: does a kernel launched with hipExtAnyOrderLaunch start while the kernel in front of it IN THE SAME STREAM is still
// running on this part (hip_ext.h notes the flag as "not supported on GFX9xx" for the module launch)?  A long kernel
// (one workgroup spinning for ~200 us), then a stamp kernel behind it -- with the flag and, as a control, without.
//   usage: anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long *out, unsigned long long ticks) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(unsigned long long *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  unsigned long long *d, h[3];
  CK(hipMalloc((void **)&d, 3 * sizeof(unsigned long long)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int grid : {1, 256})
    for (int flag : {0, 1, 0, 1}) {
      CK(hipMemset(d, 0, sizeof(h)));
      hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(64), 0, st, d, 20000ull);          // 100 MHz ticks: 200 us
      hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d);
      CK(hipGetLastError());
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
      printf("spin grid %3d, second launch %s: spin ran %.1f us; the second kernel started %.1f us after the spin STARTED, "
             "%+.1f us relative to its END -> %s\n", grid, flag ? "hipExtAnyOrderLaunch" : "plain               ",
             (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0,
             h[2] < h[1] ? "CONCURRENT" : "serialised");
    }
  return 0;
}
