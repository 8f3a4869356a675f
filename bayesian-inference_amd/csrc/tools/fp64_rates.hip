// Micro-benchmark: fp64 throughput of the matrix pipe (v_mfma_f64_16x16x4_f64), of the vector pipe
// (v_fma_f64) and of both together, as a function of waves per SIMD.  Decides where the fp64
// roofline of the triangular GEMM really is on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (8 accumulators); 1: VALU FMA only (16 chains); 2: both interleaved in each wave;
// 3: waves alternate roles by wave id parity
template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(double *out, int iters, double a0, double b0) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = a0 * i;
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = (MODE == 0) || (MODE == 2) || (MODE == 3 && (wave & 1) == 0);
  const bool do_valu = (MODE == 1) || (MODE == 2) || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int iters) {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  int threads = 256 * waves_per_simd;  // one workgroup per CU
  if (threads > 1024) threads = 1024;
  int blocks = prop.multiProcessorCount * (256 * waves_per_simd / threads);
  double *out;
  (void)hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  rate_kernel<MODE><<<blocks, threads>>>(out, iters / 10, 1.0, 1.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  rate_kernel<MODE><<<blocks, threads>>>(out, iters, 1.0000001, 0.9999999);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double nw = (double)blocks * threads / 64;
  double mf = 0, vf = 0;
  if (MODE == 0 || MODE == 2) mf = nw * iters * 8 * 2048.0;
  if (MODE == 3) mf = nw / 2 * iters * 8 * 2048.0;
  if (MODE == 1 || MODE == 2) vf = nw * iters * 8 * 16 * 64 * 2.0;
  if (MODE == 3) vf = nw / 2 * iters * 8 * 16 * 64 * 2.0;
  printf("%-28s waves/SIMD=%d  %8.3f ms  mfma %6.2f TF  valu %6.2f TF  total %6.2f TF\n", name,
         waves_per_simd, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  (void)hipFree(out);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("mfma_f64_16x16x4 only", w, 20000 / w);
    run<1>("v_fma_f64 only", w, 20000 / w);
    run<2>("both, interleaved per wave", w, 10000 / w);
    if (w >= 2) run<3>("both, alternate waves", w, 20000 / w);
  }
  return 0;
}
