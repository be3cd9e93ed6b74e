// microbenchmark: random 16-byte gathers (the table probe of k_nodes_x / k_edges_x) on MI355X:
// what bounds them — requests, or the 128-byte lines they drag from the L2 into the CU?
//   LOC = lanes per line: 1 = every lane its own line, 2/4/8 = that many neighbouring lanes share a 128-byte line
//   MODE 0 plain, 1 nontemporal, 2 sc0 (glc), 3 sc1, 4 sc0 sc1; W = bytes per lane (4, 8, 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; return x;
}

template <int MODE, int W>
__device__ __forceinline__ unsigned long long ld(const uint4* p) {
  if (W == 16) {
    uint4 v;
    if (MODE == 0) v = *p;
    if (MODE == 1) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p)); v = make_uint4(t.x, t.y, t.z, t.w); }
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return (unsigned long long)v.x + v.y + v.z + v.w;
  } else if (W == 8) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    return (unsigned long long)v.x + v.y;
  } else {
    return *reinterpret_cast<const unsigned int*>(p);
  }
}

template <int MODE, int W, int LOC>
__global__ __launch_bounds__(256) void k_gather(const uint4* tab, unsigned long long mask, long long n, unsigned long long* out) {
  long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  const uint4* p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // lanes sharing a line: the line is chosen by (wave-item / LOC), the slot inside it by the lane
    const unsigned long long item = (unsigned long long)(i0 / 4) + (unsigned long long)j * 0x100000000ull;
    const unsigned long long grp = LOC > 1 ? (item / LOC) : item;
    unsigned long long h = mix64(grp * 0x9E3779B97F4A7C15ull + 12345);
    unsigned long long slot = h & mask;
    if (LOC > 1) slot = (slot & ~7ull) | (item % LOC);
    p[j] = tab + slot;
  }
  unsigned long long s = 0;
  if (MODE == 0 || MODE == 1 || W != 16) {
    unsigned long long v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ld<MODE, W>(p[j]);
    s = v[0] + v[1] + v[2] + v[3];
  } else {
    // four loads in flight, one wait
    uint4 v0, v1, v2, v3;
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %4, off sc0\n global_load_dwordx4 %1, %5, off sc0\n global_load_dwordx4 %2, %6, off sc0\n global_load_dwordx4 %3, %7, off sc0\n s_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %4, off sc1\n global_load_dwordx4 %1, %5, off sc1\n global_load_dwordx4 %2, %6, off sc1\n global_load_dwordx4 %3, %7, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n global_load_dwordx4 %1, %5, off sc0 sc1\n global_load_dwordx4 %2, %6, off sc0 sc1\n global_load_dwordx4 %3, %7, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
    s = (unsigned long long)v0.x + v1.y + v2.z + v3.w;
  }
  if (s == 0x12345678ull) out[0] = s;
}

template <int MODE, int W, int LOC>
double run(const uint4* tab, unsigned long long slots, long long n, unsigned long long* out) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<MODE, W, LOC>), dim3(blocks), dim3(256), 0, 0, tab, slots - 1, n, out);
  hipEventRecord(a);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_gather<MODE, W, LOC>), dim3(blocks), dim3(256), 0, 0, tab, slots - 1, n, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const long long n = 56000000;
  unsigned long long* out; hipMalloc(&out, 64);
  for (unsigned long long slots : {1ull << 14, 1ull << 16, 1ull << 18, 1ull << 20, 1ull << 24}) {
    uint4* tab; hipMalloc(&tab, slots * 16); hipMemset(tab, 1, slots * 16);
    printf("table %8.2f MB (56 M gathers, ms):  16B plain %.3f  nt %.3f  sc0 %.3f  sc1 %.3f  sc0sc1 %.3f | 8B %.3f  4B %.3f | 16B with 2 / 4 / 8 lanes per line: %.3f %.3f %.3f\n",
           slots * 16 / 1048576.0, run<0, 16, 1>(tab, slots, n, out), run<1, 16, 1>(tab, slots, n, out), run<2, 16, 1>(tab, slots, n, out),
           run<3, 16, 1>(tab, slots, n, out), run<4, 16, 1>(tab, slots, n, out), run<0, 8, 1>(tab, slots, n, out), run<0, 4, 1>(tab, slots, n, out),
           run<0, 16, 2>(tab, slots, n, out), run<0, 16, 4>(tab, slots, n, out), run<0, 16, 8>(tab, slots, n, out));
    hipFree(tab);
  }
  return 0;
}
