// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on every CU (roofline denominator for
// the fp64 triangular GEMM; the local MI355X guide lists no fp64 matrix peak) and a check of the
// operand / accumulator lane maps the kernels rely on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void layout_check(const double *A, const double *B, double *D) {
  // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
double run(int waves_per_simd, int iters) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  double *out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, 10, 1.0, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.0000001, 0.9999999);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * (double)iters * NACC * 2048.0;
  hipFree(out);
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  // layout check with asymmetric integer data
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = i * 3 + k * 7 + 1;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = k * 5 - j * 2 + 3;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  layout_check<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
  printf("layout_check mismatches: %d\n", bad);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device: %s %s CUs=%d clock=%d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  for (int w = 1; w <= 2; ++w) {
    printf("waves/SIMD=%d  acc=1: %.2f TF  acc=4: %.2f TF  acc=8: %.2f TF  acc=16: %.2f TF\n", w,
           run<1>(w, 20000), run<4>(w, 5000), run<8>(w, 2500), run<16>(w, 1250));
  }
  return bad != 0;
}
