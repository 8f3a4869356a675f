// Probe: the fit side's fp64 GEMM (k_gemm.hip, compiled into this tool) on the shapes the N = 5000 evaluation launches,
// one line each with the rate against the 78.6 TFLOP/s matrix-core peak.  The environment knobs of launch_gemm apply
// (GPEMU_GEMM_SWIZZLE, GPEMU_GEMM_BIG_MIN, ...).      usage: gemm_probe [Np = 5056]
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../k_gemm.hip"

namespace gpemu {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace gpemu

using namespace gpemu;

static double time_ms(const GemmArgs &g, bool ak, bool bk, int batch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch_gemm(g, ak, bk, batch, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(a, nullptr);
  for (int r = 0; r < reps; ++r) launch_gemm(g, ak, bk, batch, nullptr);
  hipEventRecord(b, nullptr);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char **argv) {
  const int64_t Np = argc > 1 ? atoll(argv[1]) : 5056;
  double *A, *B, *C;
  const size_t bytes = sizeof(double) * (size_t)Np * Np;
  hipMalloc(&A, bytes); hipMalloc(&B, bytes); hipMalloc(&C, bytes);
  std::vector<double> h((size_t)Np * Np);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000) - 0.5;
  hipMemcpy(A, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), bytes, hipMemcpyHostToDevice);
  hipMemset(C, 0, bytes);
  auto report = [&](const char *what, const GemmArgs &g, bool ak, bool bk, int batch, double flop) {
    const double ms = time_ms(g, ak, bk, batch, 5);
    printf("%-58s %8.3f ms  %6.2f TFLOP/s  (%.3f of peak)\n", what, ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
  };
  const int n = (int)Np;
  {  // dense square, the four layouts
    GemmArgs g; g.A = A; g.B = B; g.C = C; g.lda = g.ldb = g.ldc = Np; g.M = g.N = n; g.K = n;
    const double flop = 2.0 * n * n * (double)n;
    report("dense  C = A B^T   (A [m][k], B [n][k])", g, false, false, 1, flop);
    report("dense  C = A B     (A [m][k], B [k][n])", g, false, true, 1, flop);
    report("dense  C = A^T B   (A [k][m], B [k][n])", g, true, true, 1, flop);
  }
  {  // Cholesky trailing update of the first panel: rank 256, lower tiles
    GemmArgs g; g.A = A + 256 * Np; g.B = A + 256 * Np; g.C = C; g.lda = g.ldb = g.ldc = Np;
    g.M = g.N = n - 256; g.K = 256; g.alpha = -1.0; g.beta = 1.0; g.lower_only = 1;
    report("syrk   rank-256 update of (Np - 256)^2, lower tiles", g, false, false, 1, (double)g.M * g.M * 256.0);
  }
  {  // the same update with the tile -> XCD ownership of C shifting from launch to launch, as it does in the blocked
     // Cholesky (the trailing matrix shrinks by one panel per update): every tile of C is then read-modify-written by a
     // different XCD than the one whose L2 holds it from the launch before
    hipEvent_t ea, eb;
    hipEventCreate(&ea); hipEventCreate(&eb);
    for (int shift = 0; shift < 2; ++shift) {
      const int reps = 10;
      auto one = [&](int r) {
        const int64_t off = shift ? (int64_t)(r & 3) * 64 : 0;      // start the trailing matrix 0..3 tile rows further down
        GemmArgs g; g.A = A + (256 + off) * Np; g.B = A + (256 + off) * Np; g.C = C + off * Np + off; g.lda = g.ldb = g.ldc = Np;
        g.M = g.N = n - 256 - 256; g.K = 256; g.alpha = -1.0; g.beta = 1.0; g.lower_only = 1;
        launch_gemm(g, false, false, 1, nullptr);
      };
      one(0); one(1); one(2); one(3);
      hipDeviceSynchronize();
      hipEventRecord(ea, nullptr);
      for (int r = 0; r < reps; ++r) one(r);
      hipEventRecord(eb, nullptr);
      hipEventSynchronize(eb);
      float ms = 0;
      hipEventElapsedTime(&ms, ea, eb);
      ms /= reps;
      const double flop = (double)(n - 512) * (n - 512) * 256.0;
      printf("%-58s %8.3f ms  %6.2f TFLOP/s  (%.3f of peak)\n", shift ? "syrk   rank-256, tile ownership shifting every launch" : "syrk   rank-256, same ownership every launch", ms, flop / ms / 1e9, flop / ms / 1e9 / 78.6);
    }
  }
  {  // panel solve and in-panel update
    GemmArgs g; g.A = A + 64 * Np; g.B = B; g.C = C + 64 * Np; g.lda = Np; g.ldb = 64; g.ldc = Np;
    g.M = n - 64; g.N = 64; g.K = 64;
    report("panel  (Np - 64) x 64 x 64", g, false, false, 1, 2.0 * g.M * 64.0 * 64.0);
    g.ldb = Np; g.N = 192; g.beta = 1.0; g.alpha = -1.0;
    report("panel  (Np - 64) x 192 x 64 update", g, false, false, 1, 2.0 * g.M * 192.0 * 64.0);
  }
  {  // triangular inverse, top level: T21 = L21 W11 and W21 = -W22 T21 with b = 2048 (b2 = Np - 2048 - ... here 2048)
    const int b = 2048;
    if (n >= 2 * b) {
      GemmArgs g; g.A = A + (int64_t)b * Np; g.B = B; g.C = C + (int64_t)b * Np; g.lda = g.ldb = g.ldc = Np;
      g.M = b; g.N = b; g.K = b; g.k_from_n = 1;
      report("trtri  T21 = L21 W11, b = 2048 (W11 lower: k >= n)", g, false, true, 1, (double)b * b * b);
      GemmArgs h; h.A = A + (int64_t)b * Np + b; h.B = C + (int64_t)b * Np; h.C = B + (int64_t)b * Np; h.lda = h.ldb = h.ldc = Np;
      h.M = b; h.N = b; h.K = b; h.alpha = -1.0; h.k_to_m = 1;
      report("trtri  W21 = -W22 T21, b = 2048 (W22 lower: k <= m)", h, false, true, 1, (double)b * b * b);
    }
    const int b1 = 256, pairs = n / (2 * b1);
    GemmArgs g; g.A = A + (int64_t)b1 * Np; g.B = B; g.C = C + (int64_t)b1 * Np; g.lda = g.ldb = g.ldc = Np;
    g.strideA = g.strideB = g.strideC = 2 * b1 * Np + 2 * b1;
    g.M = b1; g.N = b1; g.K = b1; g.k_from_n = 1;
    report("trtri  T21 = L21 W11, b = 256, all pairs batched", g, false, true, pairs, (double)pairs * b1 * b1 * b1);
  }
  {  // K^-1 = W^T W
    GemmArgs g; g.A = A; g.B = A; g.C = C; g.lda = g.ldb = g.ldc = Np; g.M = g.N = g.K = n; g.lower_only = 1; g.k_from_m = 1;
    report("WtW    K^-1 = W^T W, lower tiles, k >= m", g, true, true, 1, (double)n * n * n / 3.0);
    g.k_from_m = 0;
    report("WtW    lower tiles, full K range", g, true, true, 1, (double)n * n * n);
    g.lower_only = 0;
    report("WtW    all tiles, full K range", g, true, true, 1, 2.0 * n * n * n);
  }
  hipFree(A); hipFree(B); hipFree(C);
  return 0;
}
