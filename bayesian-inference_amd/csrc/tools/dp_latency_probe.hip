// Probe: dependent-issue latency and single-wave issue interval of the fp64 instructions on the serial path of the
// 64 x 64 diagonal block (k_fit.hip: column sweep).  One wave; clock64 ticks per instruction.
//   usage: dp_latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void probe(double *out, long long *ticks, double seed) {
  double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3;
  long long t[16];
  int k = 0;
  t[k++] = clock64();
  REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
  t[k++] = clock64();                                       // 1: dependent v_fma_f64
  REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
  t[k++] = clock64();                                       // 2: dependent v_mul_f64
  REP64(asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
  t[k++] = clock64();                                       // 3: dependent v_fmac_f64 (VOP2)
  REP64(asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));)
  t[k++] = clock64();                                       // 4: dependent v_fmac_f64_dpp (+ s_nop 1)
  REP64(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));)
  t[k++] = clock64();                                       // 5: dependent v_mov_b64_dpp (+ s_nop 1)
  REP64(asm volatile("s_nop 0\n\tv_rsq_f64 %0, %0" : "+v"(a));)
  t[k++] = clock64();                                       // 6: dependent v_rsq_f64 (+ s_nop 0)
  REP64(asm volatile("s_nop 0\n\tv_rcp_f64 %0, %0" : "+v"(a));)
  t[k++] = clock64();                                       // 7: dependent v_rcp_f64
  REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));)
  t[k++] = clock64();                                       // 8: four independent chains of v_fma_f64 (per group of 4)
  REP64(asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));)
  t[k++] = clock64();                                       // 9: four independent v_fmac_f64_dpp
  float f = (float)a;
  REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(1.0000001f), "v"(0.5f));)
  t[k++] = clock64();                                       // 10: dependent v_fma_f32
  REP64(asm volatile("s_nop 0\n\tv_rsq_f32 %0, %0" : "+v"(f));)
  t[k++] = clock64();                                       // 11: dependent v_rsq_f32
  int sg;
  REP64(asm volatile("v_readlane_b32 %1, %0, 3\n\ts_nop 3\n\tv_mov_b32 %0, %1" : "+v"(f), "=s"(sg));)
  t[k++] = clock64();                                       // 12: v_readlane -> s_nop 3 -> v_mov from the SGPR
  out[threadIdx.x] = a + d0 + d1 + d2 + d3 + f;
  if (threadIdx.x == 0)
    for (int i = 0; i < k; ++i) ticks[i] = t[i];
}

int main() {
  double *out;
  long long *ticks, h[16];
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&ticks, 16 * sizeof(long long));
  const char *names[] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_fmac_f64 (VOP2)", "dependent s_nop 1 + v_fmac_f64_dpp",
                         "dependent s_nop 1 + v_mov_b64_dpp", "dependent s_nop 0 + v_rsq_f64", "dependent s_nop 0 + v_rcp_f64",
                         "4 independent v_fma_f64 (per group)", "4 independent v_fmac_f64_dpp (per group)", "dependent v_fma_f32",
                         "dependent s_nop 0 + v_rsq_f32", "v_readlane + s_nop 3 + v_mov (per round trip)"};
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, out, ticks, 1.5);
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < 12; ++i) printf("%-46s %7.1f ticks\n", names[i], (double)(h[i + 1] - h[i]) / 64.0);
  return 0;
}
