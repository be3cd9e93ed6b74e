// amg_scan.hip — exclusive prefix sums in ONE launch (decoupled look-back, hand-written for wave64).
//
// The passes scan short arrays all the time (flags -> positions, counts -> offsets: ~25 scans per cleaning
// sweep over 0.5 - 6 M elements).  A library scan is two launches (state initialisation + scan) through a
// generic dispatch layer; here a scan is one kernel and needs no initialisation launch:
//   * a workgroup takes its tile number from a counter that only ever grows (tile = ticket - the number of
//     tiles all earlier scans used: the host keeps that sum), so tiles start in order and a tile never waits for
//     one that has not started;
//   * a tile publishes {state, epoch, value} as ONE 64-bit word with a relaxed agent-scope store — value and
//     flag travel together, so no fence is needed (a release fence writes the XCD's L2 back) — first its own
//     sum (state 1), then its inclusive prefix (state 2); the status words are never cleared: a word whose
//     epoch is not the current scan's is simply "not there yet" (the buffer is zeroed when the 14-bit epoch wraps);
//   * the first wave of a tile looks back 64 predecessors at a time until it meets an inclusive prefix.
// Values are sums of non-negative counts below 2^48.
#include "amg_device.h"

#define SC_THREADS 256
#define SC_ROWS 16                       // rows of 64 per wave: a wave scans 1024 consecutive elements
#define SC_TILE (SC_THREADS * SC_ROWS)   // 4096 elements per workgroup
#define SC_VAL_MASK ((1ull << 48) - 1ull)

__device__ __forceinline__ unsigned long long sc_word(unsigned int state, unsigned int epoch, unsigned long long v) {
  return ((unsigned long long)state << 62) | ((unsigned long long)(epoch & 0x3fffu) << 48) | (v & SC_VAL_MASK);
}

// What is scanned is what a LOADER makes of element i — an array entry, or something computed on the way that would
// otherwise be a kernel of its own writing an array only the scan reads (the passes launch ~200 kernels per cleaning
// sweep and every launch costs ~5 us however little it does):
//   LoadArr<T>      in[i]
//   LoadFlagWords   the ranking bitmaps of amg_build_x.hip / amg_dist.hip: 32 flag bytes folded into bitmap word i
//                   (stored as a side effect), the value scanned is its popcount
//   LoadBitsPopc    popcount of bitmap word i
//   LoadPairWidth   directed edges of edge class i: a self-loop has one, every other class two
//   LoadArrN<T>, LoadNonzero, LoadByteSet   in[i] / (in[i] != 0) with the terminator built in
// Two scans that do not depend on each other travel as ONE launch (k_exscan<LA, LB>: tiles [0, tiles_a) scan the first
// array, the tiles behind them the second, whose look-back stops at its own first tile).
template <class T>
struct LoadArr {
  const T* in;
  __device__ __forceinline__ unsigned long long operator()(long long i) const { return (unsigned long long)in[i]; }
};
template <class T>
struct LoadArrN {  // in[i] for i < n, 0 from there on (the scan's terminator: no cleared element behind the array)
  const T* in;
  long long n;
  __device__ __forceinline__ unsigned long long operator()(long long i) const { return i < n ? (unsigned long long)in[i] : 0ull; }
};
struct LoadNonzero {  // 1 where in[i] != 0, i < n
  const unsigned int* in;
  long long n;
  __device__ __forceinline__ unsigned long long operator()(long long i) const { return i < n && in[i] != 0u ? 1ull : 0ull; }
};
struct LoadByteSet {  // 1 where byte i is set, i < n
  const unsigned char* in;
  long long n;
  __device__ __forceinline__ unsigned long long operator()(long long i) const { return i < n && in[i] != 0 ? 1ull : 0ull; }
};
struct LoadApplyKill {  // node removal: a marked live node dies; kill[i] is left as "removed now" and counted
  unsigned char* kill;
  unsigned char* alive;
  long long n;
  __device__ __forceinline__ unsigned long long operator()(long long i) const {
    if (i >= n) return 0ull;
    const bool f = kill[i] != 0 && alive[i] != 0;
    if (f) alive[i] = 0;
    kill[i] = f ? 1 : 0;
    return f ? 1ull : 0ull;
  }
};
struct LoadFlagWords {
  const unsigned char* flags;
  unsigned int* bits;
  long long n_words;  // elements from here on count 0 (the scan's terminator: its prefix is the number of set bits)
  __device__ __forceinline__ unsigned long long operator()(long long i) const {
    if (i >= n_words) return 0ull;
    const uint4* p = reinterpret_cast<const uint4*>(flags + 32 * i);
    const uint4 a = p[0], b = p[1];
    auto nib = [](unsigned int x) { return (x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u); };
    const unsigned int w = nib(a.x) | (nib(a.y) << 4) | (nib(a.z) << 8) | (nib(a.w) << 12) | (nib(b.x) << 16) |
                           (nib(b.y) << 20) | (nib(b.z) << 24) | (nib(b.w) << 28);
    bits[i] = w;
    return (unsigned long long)__popc(w);
  }
};
struct LoadBitsPopc {
  const unsigned int* bits;
  long long n_words;
  __device__ __forceinline__ unsigned long long operator()(long long i) const {
    return i < n_words ? (unsigned long long)__popc(bits[i]) : 0ull;
  }
};
struct LoadPairWidth {
  const unsigned long long* pkey;
  long long n_pairs;  // elements from here on count 0 (the scan's terminator)
  __device__ __forceinline__ unsigned long long operator()(long long i) const {
    if (i >= n_pairs) return 0ull;
    const unsigned long long key = pkey[i];
    const unsigned int lo = (unsigned int)((key >> 32) & 0x7fffffffull);
    const unsigned int hi = (unsigned int)(key & 0xffffffffull) - 1u;
    return lo == hi ? 1ull : 2ull;
  }
};

struct LoadNone {
  __device__ __forceinline__ unsigned long long operator()(long long) const { return 0ull; }
};

template <class Load, class LoadB = LoadNone>
__global__ __launch_bounds__(SC_THREADS) void k_exscan(Load load, long long* __restrict__ out_a,
                                                        long long n_a, LoadB load_b, long long* __restrict__ out_b,
                                                        long long n_b, long long tiles_a, unsigned long long* counter,
                                                        unsigned long long ticket_base, unsigned long long* status,
                                                        unsigned int epoch) {
  __shared__ unsigned long long s_wave[SC_THREADS / 64];
  __shared__ unsigned long long s_excl;
  __shared__ unsigned int s_tile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_tile = (unsigned int)(atomicAdd(counter, 1ull) - ticket_base);
  __syncthreads();
  const long long tile = s_tile;
  // the second array's tiles (block-uniform; never taken with LoadNone: tiles_a is then the whole grid)
  const bool second = tile >= tiles_a;
  const long long seg0 = second ? tiles_a : 0;  // first tile of this tile's array: where its look-back ends
  const long long n = second ? n_b : n_a;
  long long* __restrict__ out = second ? out_b : out_a;
  const long long w0 = (tile - seg0) * SC_TILE + (long long)wave * (64 * SC_ROWS);
  // ---- the wave's 1024 elements as 16 coalesced rows; inclusive scan of every row, rows chained
  unsigned long long x[SC_ROWS], inc[SC_ROWS];
#pragma unroll
  for (int r = 0; r < SC_ROWS; ++r) {
    const long long i = w0 + r * 64 + lane;
    x[r] = i < n ? (second ? load_b(i) : load(i)) : 0ull;
  }
  unsigned long long row_off = 0;
#pragma unroll
  for (int r = 0; r < SC_ROWS; ++r) {
    unsigned long long v = x[r];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(v, d, 64);
      if (lane >= d) v += o;
    }
    inc[r] = v + row_off;
    row_off += __shfl(v, 63, 64);
  }
  if (lane == 0) s_wave[wave] = row_off;  // the wave's sum
  __syncthreads();
  unsigned long long wave_excl = 0, tile_sum = 0;
#pragma unroll
  for (int w = 0; w < SC_THREADS / 64; ++w) {
    const unsigned long long s = s_wave[w];
    wave_excl += w < wave ? s : 0ull;
    tile_sum += s;
  }
  // ---- look-back (first wave): sum of everything before this tile
  if (wave == 0) {
    unsigned long long excl = 0;
    if (tile == seg0) {
      if (lane == 0)
        __hip_atomic_store(status + tile, sc_word(2u, epoch, tile_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(status + tile, sc_word(1u, epoch, tile_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (long long look = tile - 1;; look -= 64) {
        const long long idx = look - lane;
        unsigned int state = 2u;  // before the array's first tile: an inclusive prefix of 0
        unsigned long long val = 0;
        if (idx >= seg0) {
          unsigned long long w;
          do {  // the tile at idx has started (tickets are taken in order) and publishes without waiting for anybody
            w = __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((unsigned int)((w >> 48) & 0x3fffu) != (epoch & 0x3fffu) || (w >> 62) == 0ull);
          state = (unsigned int)(w >> 62);
          val = w & SC_VAL_MASK;
        }
        const unsigned long long full = __ballot(state == 2u);
        // lanes up to the nearest inclusive prefix count; nothing further back does
        const int stop = full ? __ffsll((long long)full) - 1 : 63;
        unsigned long long part = lane <= stop ? val : 0ull;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d, 64);
        excl += part;
        if (full) break;
      }
      if (lane == 0)
        __hip_atomic_store(status + tile, sc_word(2u, epoch, excl + tile_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_excl = excl;
  }
  __syncthreads();
  const unsigned long long base = s_excl + wave_excl;
#pragma unroll
  for (int r = 0; r < SC_ROWS; ++r) {
    const long long i = w0 + r * 64 + lane;
    if (i < n) out[i] = (long long)(base + inc[r] - x[r]);
  }
}

template <class Load, class LoadB = LoadNone>
static int exscan(amg_ctx* c, Load load, long long* out, size_t n, LoadB load_b = LoadNone{}, long long* out_b = nullptr,
                  size_t n_b = 0) {
  if (n == 0 && n_b == 0) return AMG_OK;
  const unsigned long long tiles_a = (n + SC_TILE - 1) / SC_TILE;
  const unsigned long long tiles = tiles_a + (n_b + SC_TILE - 1) / SC_TILE;
  // [0] the ticket counter, [8 ...] one status word per tile
  const size_t need = (size_t)(tiles + 8) * sizeof(unsigned long long);
  if (need > c->scan_state.cap || c->scan_epoch >= 0x3fffu) {
    if (need > c->scan_state.cap) {
      AMGCHK(c->scan_state.ensure(need * 2));
      c->scan_tickets = 0;
      HIPCHK(hipMemsetAsync(c->scan_state.p, 0, c->scan_state.cap, c->stream));
    } else {  // the epoch wraps: forget every old status word (the ticket counter keeps counting)
      HIPCHK(hipMemsetAsync(c->scan_state.as<unsigned long long>() + 8, 0,
                            c->scan_state.cap - 8 * sizeof(unsigned long long), c->stream));
    }
    c->scan_epoch = 0;
  }
  const unsigned int epoch = ++c->scan_epoch;
  unsigned long long* st = c->scan_state.as<unsigned long long>();
  hipLaunchKernelGGL((k_exscan<Load, LoadB>), dim3((unsigned int)tiles), dim3(SC_THREADS), 0, c->stream, load, out,
                     (long long)n, load_b, out_b, (long long)n_b, (long long)tiles_a, st, c->scan_tickets, st + 8, epoch);
  c->scan_tickets += tiles;
  return AMG_OK;
}

int prim_exscan_u32_to_i64(amg_ctx* c, const unsigned int* in, long long* out, size_t n) {
  return exscan(c, LoadArr<unsigned int>{in}, out, n);
}

int prim_exscan_i64(amg_ctx* c, const long long* in, long long* out, size_t n) {
  return exscan(c, LoadArr<long long>{in}, out, n);
}

// two independent scans of n + 1 elements each (in[n] counts 0: out[n] = the sum), one launch
int prim_exscan_u32_pair(amg_ctx* c, const unsigned int* in_a, long long* out_a, const unsigned int* in_b, long long* out_b,
                         size_t n) {
  return exscan(c, LoadArrN<unsigned int>{in_a, (long long)n}, out_a, n + 1, LoadArrN<unsigned int>{in_b, (long long)n}, out_b,
                n + 1);
}

int prim_exscan_i64_pair(amg_ctx* c, const long long* in_a, long long* out_a, const long long* in_b, long long* out_b,
                         size_t n) {
  return exscan(c, LoadArrN<long long>{in_a, (long long)n}, out_a, n + 1, LoadArrN<long long>{in_b, (long long)n}, out_b,
                n + 1);
}

// out[i] = set bytes before byte i for i <= n
int prim_exscan_bytes_set(amg_ctx* c, const unsigned char* in, long long* out, size_t n) {
  return exscan(c, LoadByteSet{in, (long long)n}, out, n + 1);
}

// alive[i] = 0 and kill[i] = 1 where kill[i] was set on a live node (kill[i] = 0 elsewhere); out[i] = nodes removed
// before node i for i <= n
int prim_exscan_apply_kill(amg_ctx* c, unsigned char* kill, unsigned char* alive, long long* out, size_t n) {
  return exscan(c, LoadApplyKill{kill, alive, (long long)n}, out, n + 1);
}

// out_keep = exscan(len[i] != 0), out_off = exscan(len[i]) over n + 1 elements, one launch
int prim_exscan_keep_and_len(amg_ctx* c, const unsigned int* len, long long* out_keep, long long* out_off, size_t n) {
  return exscan(c, LoadNonzero{len, (long long)n}, out_keep, n + 1, LoadArrN<unsigned int>{len, (long long)n}, out_off, n + 1);
}

// flags[32 n_words] -> bits[n_words]; out[i] = set bits before word i for i <= n_words (out[n_words] = number of set bits)
int prim_exscan_flag_words(amg_ctx* c, const unsigned char* flags, unsigned int* bits, long long* out, size_t n_words) {
  return exscan(c, LoadFlagWords{flags, bits, (long long)n_words}, out, n_words + 1);
}

int prim_exscan_bits_popc(amg_ctx* c, const unsigned int* bits, long long* out, size_t n_words) {
  return exscan(c, LoadBitsPopc{bits, (long long)n_words}, out, n_words + 1);
}

// out[i] = first directed edge of edge class i, out[n_pairs] = number of directed edges
int prim_exscan_pair_width(amg_ctx* c, const unsigned long long* pkey, long long* out, size_t n_pairs) {
  return exscan(c, LoadPairWidth{pkey, (long long)n_pairs}, out, n_pairs + 1);
}
