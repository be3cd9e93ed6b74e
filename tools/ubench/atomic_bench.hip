// microbenchmark: scattered u32 atomicAdd throughput on MI355X by memory scope.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; return x;
}
__device__ __forceinline__ unsigned int xcc_id() {
  unsigned int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int MODE>
__global__ void k_add(unsigned int* tab, unsigned long long mask, long long n, int stride_words, unsigned long long n_slots) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // skewed distribution: 90 % of accesses to a hot set of 1/256 of the slots
  unsigned long long h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
  unsigned long long slot = (h & 15) < 14 ? ((h >> 8) & (mask >> 8)) * 256 : ((h >> 8) & mask);
  unsigned int* p = tab + slot * stride_words;
  if (MODE == 0) atomicAdd(p, 1u);                                                      // agent (default)
  if (MODE == 1) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 2) __hip_atomic_fetch_add(p + n_slots * stride_words * xcc_id(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 3) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  if (MODE == 4) { unsigned int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (v == 0xffffffffu) p[1] = v; }  // sc1 load only
  if (MODE == 5) { unsigned int v = *(volatile unsigned int*)p; if (v == 0xffffffffu) p[1] = v; }  // plain load only
}

template <int MODE>
double run(unsigned int* tab, unsigned long long slots, long long n, int stride_words, size_t bytes) {
  hipMemset(tab, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_add<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, slots - 1, n, stride_words, slots);
  hipMemset(tab, 0, bytes);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_add<MODE>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tab, slots - 1, n, stride_words, slots);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const long long n = 56000000;
  for (unsigned long long slots : {1ull << 19, 1ull << 23, 1ull << 27}) {
    for (int stride : {1, 8}) {
      size_t bytes = slots * stride * 4 * 8;  // x8 for per-XCD copies
      if (bytes > (8ull << 30)) continue;
      unsigned int* tab; if (hipMalloc(&tab, bytes) != hipSuccess) { printf("alloc fail\n"); continue; }
      double t0 = run<0>(tab, slots, n, stride, bytes);
      double t1 = run<1>(tab, slots, n, stride, bytes);
      // correctness of workgroup-scope adds on a SHARED table: total must be n
      std::vector<unsigned int> h(slots * stride);
      hipMemcpy(h.data(), tab, slots * stride * 4, hipMemcpyDeviceToHost);
      unsigned long long sum1 = 0; for (auto v : h) sum1 += v;
      double t2 = run<2>(tab, slots, n, stride, bytes);
      std::vector<unsigned int> h8(slots * stride * 8);
      hipMemcpy(h8.data(), tab, bytes, hipMemcpyDeviceToHost);
      unsigned long long sum2 = 0; for (auto v : h8) sum2 += v;
      double t3 = run<3>(tab, slots, n, stride, bytes);
      double t4 = run<4>(tab, slots, n, stride, bytes);
      double t5 = run<5>(tab, slots, n, stride, bytes);
      printf("slots=2^%d stride=%dB  agent %.3f ms | workgroup(shared) %.3f ms sum=%llu (%s) | workgroup(per-XCD) %.3f ms sum=%llu (%s) | wavefront %.3f | sc1 load %.3f | plain load %.3f\n",
             __builtin_ctzll(slots), stride * 4, t0, t1, sum1, sum1 == (unsigned long long)n ? "exact" : "LOST", t2, sum2,
             sum2 == (unsigned long long)n ? "exact" : "LOST", t3, t4, t5);
      hipFree(tab);
    }
  }
  return 0;
}
