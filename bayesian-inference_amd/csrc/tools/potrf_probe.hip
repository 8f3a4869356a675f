// Probe: where the 18 us of one potrf_diag_kernel launch (64 x 64 Cholesky + inverse, the serial step of the blocked
// factorisation, k_fit.hip) go.  Compiles k_fit.hip / k_gemm.hip into the tool with cycle stamps switched on.
//   usage: potrf_probe
#define GPEMU_POTRF_STAMPS 1
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../k_gemm.hip"
#include "../k_fit.hip"

namespace gpemu {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace gpemu

int main() {
  const int64_t Np = 5056;
  double *A, *Dinv;
  int *info;
  hipMalloc(&A, sizeof(double) * Np * Np);
  hipMalloc(&Dinv, sizeof(double) * Np * 64);
  hipMalloc(&info, sizeof(int));
  hipMemset(info, 0, sizeof(int));
  std::vector<double> h((size_t)Np * Np, 0.0);
  for (int64_t i = 0; i < Np; ++i)
    for (int64_t j = 0; j <= i && j > i - 64; ++j) h[i * Np + j] = (i == j) ? 70.0 : 1.0 / (1.0 + (double)(i - j));
  hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nblk = (int)(Np / 64);
  for (int pass = 0; pass < 2; ++pass) {
    hipEventRecord(e0, nullptr);
    for (int jb = 0; jb < nblk; ++jb)
      hipLaunchKernelGGL(gpemu::potrf_diag_kernel, dim3(1, 1), dim3(256), 0, nullptr, A + (int64_t)jb * 64 * Np + (int64_t)jb * 64, Np,
                         Dinv + (int64_t)jb * 64 * 64, 1, jb, info, Np * Np, Np * 64);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d dependent launches: %.2f us each\n", nblk, 1e3 * ms / nblk);
    hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
  }
  {
    // one launch of the panel kernel on the first panel of a fresh matrix: who does what when (us after the first start)
    hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
    int *flags;
    hipMalloc(&flags, sizeof(int) * gpemu::CHOL_FLAGS);
    for (int rows : {79, 39, 11}) {
      hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
      hipMemset(flags, 0, sizeof(int) * gpemu::CHOL_FLAGS);
      hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(gpemu::chol_panel_kernel, dim3(rows, 1), dim3(256), 0, nullptr, A, Np, Dinv, 0, 4, info, flags, 1, Np * Np, Np * 64, 0);
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      static long long ps[128][16];
      hipMemcpyFromSymbol(ps, HIP_SYMBOL(gpemu::g_panel_stamps), sizeof(ps));
      long long t0 = ps[0][0];
      for (int r = 0; r < rows; ++r) t0 = ps[r][0] < t0 ? ps[r][0] : t0;
      printf("panel kernel, %d rows: %.1f us by events\n row  start loaded |  D0 seen, step 0 done | D1 ... | D2 ... | D3 ... | end\n", rows, 1e3 * ms);
      for (int r = 0; r < rows; ++r) {
        if (r >= 6 && r < rows - 3 && r % 16) continue;
        printf("%4d", r);
        for (int k = 0; k <= 10; ++k) printf(" %7.2f", ps[r][k] ? (ps[r][k] - t0) / 100.0 : 0.0);
        printf("\n");
      }
      static long long zero[128][16];
      hipMemcpyToSymbol(HIP_SYMBOL(gpemu::g_panel_stamps), zero, sizeof(zero));
    }
  }
  long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(gpemu::g_potrf_stamps), sizeof(st));
  const char *names[] = {"global load -> LDS", "panel step 0", "panel steps 1-3", "store factor", "16 x 16 inverses", "two merge levels",
                         "store inverse"};
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("shader clock reported %d kHz; clock64 ticks:\n", clk);
  for (int i = 0; i < 7; ++i) printf("  %-22s %8lld ticks\n", names[i], st[i + 1] - st[i]);
  printf("  %-22s %8lld ticks\n", "kernel body", st[7] - st[0]);
  return 0;
}
