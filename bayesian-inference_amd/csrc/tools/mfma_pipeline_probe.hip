// Probe: where does the f64 MFMA rate go when the bare loop (tools/fp64_rates) is dressed up step by
// step into a GEMM main loop?  One 512-thread workgroup per CU (8 waves, 2 per SIMD), each wave
// 2 x 2 MFMA tiles, K step 32 per "tile" iteration (32 MFMAs per wave per iteration).
//   V0 bare MFMA, operands in registers            V1 + operands re-read from LDS each k-step
//   V2 V1 + one __syncthreads per iteration         V3 V2 + 6 ds_write_b128 per thread per iteration
//   V4 V3 + 6 global_load_dwordx4 per thread per iteration (L2-resident source)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(512, 2) void probe(const double *__restrict__ src, double *out, int iters) {
  __shared__ __attribute__((aligned(16))) double sA[2][32][80];
  __shared__ __attribute__((aligned(16))) double sB[2][32][144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, lr = lane & 15, lk = lane >> 4;
  for (int i = tid; i < 2 * 32 * 80; i += 512) (&sA[0][0][0])[i] = 1.0 + 1e-9 * i;
  for (int i = tid; i < 2 * 32 * 144; i += 512) (&sB[0][0][0])[i] = 1.0 - 1e-9 * i;
  __syncthreads();
  d4 acc[2][2];
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = d4{0, 0, 0, 0};
  double a[2][2], b[2][2];
  for (int mi = 0; mi < 2; ++mi) { a[0][mi] = a[1][mi] = 1.0 + lane * 1e-9; b[0][mi] = b[1][mi] = 1.0 - lane * 1e-9; }
  d2 st[6];
  const double *gp = src + (size_t)blockIdx.x * 8192 + tid * 2;
  for (int r = 0; r < 6; ++r) st[r] = d2{1.0, 2.0};
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    d2 nl[6];
    if (V >= 4) {
#pragma unroll
      for (int r = 0; r < 6; ++r) nl[r] = *reinterpret_cast<const d2 *>(gp + r * 1024);
    }
    if (V >= 1) {
      for (int mi = 0; mi < 2; ++mi) a[0][mi] = sA[buf][lk][wm * 32 + mi * 16 + lr];
      for (int ni = 0; ni < 2; ++ni) b[0][ni] = sB[buf][lk][wn * 32 + ni * 16 + lr];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cu = ks & 1, nx = cu ^ 1;
      if (V >= 1 && ks + 1 < 8) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[nx][mi] = sA[buf][(ks + 1) * 4 + lk][wm * 32 + mi * 16 + lr];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[nx][ni] = sB[buf][(ks + 1) * 4 + lk][wn * 32 + ni * 16 + lr];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[V >= 1 ? cu : 0][mi], b[V >= 1 ? cu : 0][ni], acc[mi][ni], 0, 0, 0);
      if (V >= 3 && ks >= 1 && ks <= 6) {
        const int part = ks - 1;
        if (part < 2) *reinterpret_cast<d2 *>(&sA[buf ^ 1][(tid >> 5) + 16 * part][2 * (tid & 31)]) = st[part];
        else *reinterpret_cast<d2 *>(&sB[buf ^ 1][(tid >> 6) + 8 * (part - 2)][2 * (tid & 63)]) = st[part];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V >= 4) {
#pragma unroll
      for (int r = 0; r < 6; ++r) st[r] = nl[r];
    }
    if (V >= 2) __syncthreads();
  }
  double s = 0;
  for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) s += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
  out[blockIdx.x * 512 + tid] = s + st[0][0];
}

template <int V>
void run(const char *name, const double *src, double *out, int ncu) {
  const int iters = 400;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  probe<V><<<ncu, 512>>>(src, out, 40);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    probe<V><<<ncu, 512>>>(src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)ncu * 8 * iters * 32 * 2048.0;
  printf("%-46s %8.3f ms  %6.2f TF  (%.1f cycles per MFMA slot at 2.4 GHz)\n", name, best, flops / best / 1e9,
         best * 1e-3 * 2.4e9 / (iters * 64.0));
}

int main() {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  double *src, *out;
  (void)hipMalloc(&src, sizeof(double) * (size_t)ncu * 8192 + 65536);
  (void)hipMemset(src, 0, sizeof(double) * (size_t)ncu * 8192 + 65536);
  (void)hipMalloc(&out, sizeof(double) * ncu * 512);
  run<0>("V0 bare MFMA (8 waves/CU, 4 acc)", src, out, ncu);
  run<1>("V1 + LDS operand reads", src, out, ncu);
  run<2>("V2 + barrier per 32 MFMAs", src, out, ncu);
  run<3>("V3 + 6 ds_write_b128 per thread interleaved", src, out, ncu);
  run<4>("V4 + 6 global_load_dwordx4 (L2 resident)", src, out, ncu);
  return 0;
}
