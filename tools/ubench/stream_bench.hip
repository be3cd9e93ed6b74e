// stream_bench.hip — what a plain streaming read / read+write kernel reaches on this GPU, by launch shape
// (the counting sweeps, mask, pack and filter kernels of the sweep all stream 240 MB arrays at ~2.4 TB/s)
// build: hipcc -O3 --offload-arch=gfx950 stream_bench.hip -o stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
template <int NT, int UNROLL>
__global__ void k_read(const i4* __restrict__ p, long long n4, int* out) {
  long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int acc = 0;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    i4 v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) v[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  }
  for (; i < n4; i += stride) { i4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678) out[0] = acc;
}
template <int UNROLL>
__global__ void k_copy(const i4* __restrict__ p, i4* __restrict__ q, long long n4) {
  long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    i4 v[UNROLL];
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) v[j] = __builtin_nontemporal_load(p + i + j * stride);
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) __builtin_nontemporal_store(v[j], q + i + j * stride);
  }
  for (; i < n4; i += stride) q[i] = p[i];
}
int main() {
  const long long n = 60ll << 20;  // 60 M ints = 240 MB
  int *a, *b, *o;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, 64);
  hipMemset(a, 1, n * 4); hipMemset(b, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch, double bytes) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-44s %7.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  };
  const long long n4 = n / 4;
  for (int blocks : {256, 512, 1024, 2048, 4096, 16384}) {
    for (int threads : {256, 1024}) {
      char nm[96];
      snprintf(nm, 96, "read  nt u4  %5d x %4d", blocks, threads);
      time(nm, [&] { hipLaunchKernelGGL((k_read<1, 4>), dim3(blocks), dim3(threads), 0, 0, (const i4*)a, n4, o); }, n * 4.0);
      snprintf(nm, 96, "read  pl u4  %5d x %4d", blocks, threads);
      time(nm, [&] { hipLaunchKernelGGL((k_read<0, 4>), dim3(blocks), dim3(threads), 0, 0, (const i4*)a, n4, o); }, n * 4.0);
    }
  }
  for (int blocks : {256, 1024, 4096, 16384}) {
    char nm[96];
    snprintf(nm, 96, "read  nt u8  %5d x  256", blocks);
    time(nm, [&] { hipLaunchKernelGGL((k_read<1, 8>), dim3(blocks), dim3(256), 0, 0, (const i4*)a, n4, o); }, n * 4.0);
    snprintf(nm, 96, "copy  nt u4  %5d x  256", blocks);
    time(nm, [&] { hipLaunchKernelGGL((k_copy<4>), dim3(blocks), dim3(256), 0, 0, (const i4*)a, (i4*)b, n4); }, n * 8.0);
  }
  return 0;
}
