// Probe of the producers' residency gate (k_predict.hip: kstar_persist_kernel): many 256-thread workgroups
// (128 VGPRs, 6 KiB LDS) count themselves into their CU by HW_REG_HW_ID / HW_REG_XCC_ID and leave unless they are among the
// first `per_cu` there; the survivors nap for `hold_us`.  While they are resident a second kernel of 256 workgroups with
// the triangular GEMM's footprint (512 threads, 128 VGPRs, 147.5 KiB LDS) is launched from another stream: how long until
// every one of its workgroups has STARTED?  (With at most two producers per CU: at once.)
//   usage: gate_probe [per_cu = 2] [producer workgroups = 768] [hold_us = 300]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void producer(int *cu_cnt, int *keys, unsigned long long *t, int per_cu, unsigned long long hold, double *sink) {
  __shared__ double pad[768];                       // 6 KiB
  __shared__ int keep;
  const int cu_key = ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 255);
  if (threadIdx.x == 0) {
    const int before = atomicAdd(cu_cnt + cu_key, 1);
    keep = before < per_cu;
    if (!keep) atomicAdd(cu_cnt + cu_key, -1);
    keys[blockIdx.x] = keep ? cu_key : -1 - cu_key;
    t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (!keep) return;
  // keep 128 VGPRs alive
  double acc[56];
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = pad[(threadIdx.x + i) % 768];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0000001 + 1e-9;
    __builtin_amdgcn_s_sleep(16);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 56; ++i) s += acc[i];
  if (s == 1.2345e300) *sink = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(cu_cnt + cu_key, -1);
}

__global__ __launch_bounds__(512, 2) void big(unsigned long long *t, int *keys, double *sink) {
  __shared__ double lds[18880];                     // 151040 B
  const int cu_key = ((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 255);
  if (threadIdx.x == 0) { t[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); keys[blockIdx.x] = cu_key; }
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double acc[56];
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = lds[(threadIdx.x + 37 * i) % 18880];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) {          // 20 us
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0000001 + 1e-9;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 56; ++i) s += acc[i];
  if (s == 1.2345e300) *sink = s;
}

int main(int argc, char **argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 2, nprod = argc > 2 ? atoi(argv[2]) : 768;
  const unsigned long long hold = (argc > 3 ? atoll(argv[3]) : 300) * 100ull;
  int *cnt, *keys, *bkeys;
  unsigned long long *tp, *tb;
  double *sink;
  CK(hipMalloc((void **)&cnt, 4096 * sizeof(int)));
  CK(hipMemset(cnt, 0, 4096 * sizeof(int)));
  CK(hipMalloc((void **)&keys, nprod * sizeof(int)));
  CK(hipMalloc((void **)&bkeys, 256 * sizeof(int)));
  CK(hipMalloc((void **)&tp, nprod * sizeof(unsigned long long)));
  CK(hipMalloc((void **)&tb, 256 * sizeof(unsigned long long)));
  CK(hipMalloc((void **)&sink, 8));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(tb, 0, 256 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(producer, dim3(nprod), dim3(256), 0, sb, cnt, keys, tp, per_cu, hold, sink);
    CK(hipStreamSynchronize(sa));
    // the producers are resident by now (they were launched ~20 us ago); the big kernel from the other stream
    for (volatile int spin = 0; spin < 200000; ++spin) {}
    hipLaunchKernelGGL(big, dim3(256), dim3(512), 0, sa, tb, bkeys, sink);
    CK(hipDeviceSynchronize());
    std::vector<int> hk(nprod), hb(256);
    std::vector<unsigned long long> htp(nprod), htb(256);
    CK(hipMemcpy(hk.data(), keys, nprod * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), bkeys, 256 * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(htp.data(), tp, nprod * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(htb.data(), tb, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::map<int, int> surv, all, bigk;
    int ns = 0;
    for (int i = 0; i < nprod; ++i) { const int k = hk[i] >= 0 ? hk[i] : -1 - hk[i]; all[k]++; if (hk[i] >= 0) { surv[k]++; ++ns; } }
    for (int i = 0; i < 256; ++i) bigk[hb[i]]++;
    int mx = 0, mxb = 0;
    for (auto &kv : surv) mx = std::max(mx, kv.second);
    for (auto &kv : bigk) mxb = std::max(mxb, kv.second);
    unsigned long long p0 = ~0ull, b0 = ~0ull, b1 = 0;
    for (auto v : htp) p0 = std::min(p0, v);
    for (auto v : htb) { b0 = std::min(b0, v); b1 = std::max(b1, v); }
    printf("rep %d: %d producers launched, %d kept (at most %d per key), %zu distinct CU keys seen by producers; big kernel: %zu distinct keys "
           "(at most %d workgroups per key), first workgroup started %.1f us after the first producer, the LAST %.1f us after the first\n",
           rep, nprod, ns, mx, all.size(), bigk.size(), mxb, (double)(b0 - p0) / 100.0, (double)(b1 - b0) / 100.0);
    int cnts[4096];
    CK(hipMemcpy(cnts, cnt, sizeof(cnts), hipMemcpyDeviceToHost));
    int left = 0;
    for (int i = 0; i < 4096; ++i) left += cnts[i] != 0;
    if (left) printf("  %d CU counters are not back to zero!\n", left);
  }
  return 0;
}
