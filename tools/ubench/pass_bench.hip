// microbenchmark: the skeleton of a table pass — stream 4 B per element in, one 16-byte probe per element into a table
// whose hot set (20 k slots taking 97 % of the probes) lies scattered over `slots` 16-byte slots, stream 5 B per element
// out — to learn what such a pass costs on MI355X when nothing else is in it.
//   V 0: int4 in, 4 probes in flight, int4 + u32 out   V 1: same with nontemporal in / out   V 2: no probes
//   V 3: probes only (no streams)   V 4: as V 0 through LDS with one barrier (tile of 1024 per 256 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}

template <int V, int NALU = 0>
__global__ __launch_bounds__(256, 8) void k_pass(const int* __restrict__ in, long long n, const uint4* __restrict__ tab,
                                                 const unsigned int* __restrict__ hot, unsigned int mask,
                                                 int* __restrict__ out, unsigned char* __restrict__ outb) {
  __shared__ int s_in[1024 + 8];
  const long long t = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (t + 4 > n) return;
  i4 x = {1, 2, 3, 4};
  if (V == 4) {
    reinterpret_cast<i4*>(s_in)[threadIdx.x] = *reinterpret_cast<const i4*>(in + t);
    __syncthreads();
    // strided windows as in k_nodes_x
    x.x = s_in[threadIdx.x]; x.y = s_in[threadIdx.x + 256]; x.z = s_in[threadIdx.x + 512]; x.w = s_in[threadIdx.x + 768];
  } else if (V == 1) x = __builtin_nontemporal_load(reinterpret_cast<const i4*>(in + t));
  else if (V != 3) x = *reinterpret_cast<const i4*>(in + t);
  else { x.x = (int)t; x.y = (int)t + 1; x.z = (int)t + 2; x.w = (int)t + 3; }
  if (NALU > 0) {  // dummy dependent integer work between the stream load and the probes (4 chains)
    unsigned int a = (unsigned int)x.x, b = (unsigned int)x.y, c = (unsigned int)x.z, d = (unsigned int)x.w;
#pragma unroll 16
    for (int r = 0; r < NALU / 8; ++r) {
      a = (a ^ (a >> 7)) + b; b = (b ^ (b >> 5)) + c; c = (c ^ (c >> 3)) + d; d = (d ^ (d >> 9)) + a;
    }
    x.x = (int)a; x.y = (int)b; x.z = (int)c; x.w = (int)d;
  }
  unsigned int idx[4];
  const int xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int h = mix32((unsigned int)xs[j] * 0x9E3779B9u + (unsigned int)j);
    // the element's value decides: hot (value % 32 != 0) -> one of the 20 k hot slots
    idx[j] = (h & 31u) ? mix32((h >> 5) % 20000u + 77u) & mask : (h >> 5) & mask;
  }
  uint4 v[4];
  if (V != 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = tab[idx[j]];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = make_uint4(idx[j], 0, 0, 0);
  }
  i4 o = {(int)(v[0].x + v[0].w), (int)(v[1].x + v[1].w), (int)(v[2].x + v[2].w), (int)(v[3].x + v[3].w)};
  const unsigned int ob = (v[0].y & 0xffu) | ((v[1].y & 0xffu) << 8) | ((v[2].y & 0xffu) << 16) | (v[3].y << 24);
  if (V == 3) {
    if (o.x + o.y + o.z + o.w == 0x12345) out[0] = 1;
  } else if (V == 1) {
    __builtin_nontemporal_store(o, reinterpret_cast<i4*>(out + t));
    __builtin_nontemporal_store(ob, reinterpret_cast<unsigned int*>(outb + t));
  } else {
    *reinterpret_cast<i4*>(out + t) = o;
    *reinterpret_cast<unsigned int*>(outb + t) = ob;
  }
}

template <int V, int NALU = 0>
double run(const int* in, long long n, const uint4* tab, const unsigned int* hot, unsigned int mask, int* out, unsigned char* outb) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_pass<V, NALU>), dim3(blocks), dim3(256), 0, 0, in, n, tab, hot, mask, out, outb);
  (void)hipEventRecord(a);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_pass<V, NALU>), dim3(blocks), dim3(256), 0, 0, in, n, tab, hot, mask, out, outb);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const long long n = 60000000;
  int* in; int* out; unsigned char* outb; unsigned int* hot;
  (void)hipMalloc(&in, n * 4 + 64); (void)hipMalloc(&out, n * 4 + 64); (void)hipMalloc(&outb, n + 64); (void)hipMalloc(&hot, 20000 * 4);
  std::vector<int> h(n); unsigned long long s = 88172645463325252ull;
  for (long long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s >> 33); }
  (void)hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  for (unsigned long long slots : {1ull << 18, 1ull << 20, 1ull << 24}) {
    uint4* tab; (void)hipMalloc(&tab, slots * 16); (void)hipMemset(tab, 1, slots * 16);
    std::vector<unsigned int> hh(20000);
    for (auto& v : hh) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (unsigned int)(s >> 20) & (unsigned int)(slots - 1); }
    (void)hipMemcpy(hot, hh.data(), 20000 * 4, hipMemcpyHostToDevice);
    printf("table %7.1f MB, 60 M elements, ms: full %.3f | nontemporal streams %.3f | no probes %.3f | probes only %.3f | via LDS + barrier, strided %.3f\n",
           slots * 16 / 1048576.0, run<0>(in, n, tab, hot, (unsigned)(slots - 1), out, outb), run<1>(in, n, tab, hot, (unsigned)(slots - 1), out, outb),
           run<2>(in, n, tab, hot, (unsigned)(slots - 1), out, outb), run<3>(in, n, tab, hot, (unsigned)(slots - 1), out, outb),
           run<4>(in, n, tab, hot, (unsigned)(slots - 1), out, outb));
    printf("     nontemporal + N dummy VALU instructions per thread: 256: %.3f  512: %.3f  1024: %.3f  2048: %.3f\n",
           run<1, 256>(in, n, tab, hot, (unsigned)(slots - 1), out, outb), run<1, 512>(in, n, tab, hot, (unsigned)(slots - 1), out, outb),
           run<1, 1024>(in, n, tab, hot, (unsigned)(slots - 1), out, outb), run<1, 2048>(in, n, tab, hot, (unsigned)(slots - 1), out, outb));
    (void)hipFree(tab);
  }
  return 0;
}
