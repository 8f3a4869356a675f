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
    for (int64_t j = (i >= 63 ? i - 63 : 0); j <= i; ++j) h[i * Np + j] = (i == j) ? 70.0 : 1.0 / (1.0 + (double)(i - j));
  hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nblk = (int)(Np / 64);
  for (int jb = 0; jb < 3; ++jb) {      // the first blocks one by one: info and the sweep's flags (1 / sqrt of the pivots)
    hipLaunchKernelGGL(gpemu::potrf_diag_kernel, dim3(1, 1), dim3(256), 0, nullptr, A + (int64_t)jb * 64 * Np + (int64_t)jb * 64, Np,
                       Dinv + (int64_t)jb * 64 * 64, 1, jb, info, Np * Np, Np * 64);
    int hinfo = -7;
    hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost);
    double fl[64];
    hipMemcpyFromSymbol(fl, HIP_SYMBOL(gpemu::g_potrf_flags), sizeof(fl));
    double nxt[3];
    hipMemcpy(nxt, A + (int64_t)(jb + 1) * 64 * Np + (int64_t)(jb + 1) * 64, sizeof(nxt), hipMemcpyDeviceToHost);
    printf("block %d: info %d, flags %.6g %.6g %.6g ... %.6g; next block's first row %.6g %.6g %.6g\n", jb, hinfo, fl[0], fl[1], fl[2], fl[63],
           nxt[0], nxt[1], nxt[2]);
  }
  hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
  hipMemset(info, 0, sizeof(int));
  for (int pass = 0; pass < 2; ++pass) {
    hipEventRecord(e0, nullptr);
    for (int jb = 0; jb < nblk; ++jb)
      hipLaunchKernelGGL(gpemu::potrf_diag_kernel, dim3(1, 1), dim3(256), 0, nullptr, A + (int64_t)jb * 64 * Np + (int64_t)jb * 64, Np,
                         Dinv + (int64_t)jb * 64 * 64, 1, jb, info, Np * Np, Np * 64);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int hinfo = -7;
    hipMemcpy(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost);
    printf("%d dependent launches: %.2f us each (info %d)\n", nblk, 1e3 * ms / nblk, hinfo);

    hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
  }
  long long st[16];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(gpemu::g_potrf_stamps), sizeof(st));     // of the last potrf_diag launch (the panel kernel stamps 2 and 5 too)
  {
    // one launch of the panel kernel on the first panel of a fresh matrix: who does what when (us after the first start)
    hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
    int *flags;
    hipMalloc(&flags, sizeof(int) * gpemu::CHOL_FLAGS);
    for (int rows : {79, 39, 11}) {
      hipMemcpy(A, h.data(), sizeof(double) * Np * Np, hipMemcpyHostToDevice);
      hipMemset(flags, 0, sizeof(int) * gpemu::CHOL_FLAGS);
      hipEventRecord(e0, nullptr);
      hipLaunchKernelGGL(gpemu::chol_panel_kernel, dim3(rows, 1), dim3(256), 0, nullptr, A, Np, Dinv, 0, 4, info, flags, 1, Np * Np, Np * 64, 0, 1);
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
        if (r >= 1 && r <= 3)       // a head's last step: Dinv seen -> in LDS -> L formed -> stored -> own block updated -> posted
          printf("   | last step: +%.2f +%.2f +%.2f +%.2f +%.2f", (ps[r][11] - ps[r][2 * r]) / 100.0, (ps[r][12] - ps[r][11]) / 100.0,
                 (ps[r][13] - ps[r][12]) / 100.0, (ps[r][14] - ps[r][13]) / 100.0, (ps[r][2 * r + 1] - ps[r][14]) / 100.0);
        printf("\n");
      }
      static long long zero[128][16];
      hipMemcpyToSymbol(HIP_SYMBOL(gpemu::g_panel_stamps), zero, sizeof(zero));
    }
  }
  const char *names[] = {"global load -> LDS", "panel step 0", "panel steps 1-3", "store factor", "16 x 16 inverses", "two merge levels",
                         "store inverse"};
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("shader clock reported %d kHz; clock64 ticks:\n", clk);
  for (int i = 0; i < 7; ++i) printf("  %-22s %8lld ticks\n", names[i], st[i + 1] - st[i]);
  printf("  %-22s %8lld ticks\n", "kernel body", st[7] - st[0]);
  printf("  %-22s %8lld ticks\n", "step 0: column sweep", st[8] - st[1]);
  printf("  %-22s %8lld ticks\n", "step 0: rank-16 update", st[2] - st[8]);
  printf("  %-22s %8lld %8lld %8lld ticks after the load\n", "step 0: waves 0 1 2 done", st[10] - st[1], st[11] - st[1], st[12] - st[1]);
  printf("  %-22s %8lld ticks\n", "merge level 16", st[9] - st[5]);
  printf("  %-22s %8lld ticks\n", "merge level 32", st[6] - st[9]);
  return 0;
}
