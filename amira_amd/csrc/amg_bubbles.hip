// amg_bubbles.hip — the steps of bubble popping that grow with the data, on device ids:
//
//   amg_junction_paths         every path between two junctions of the live graph that has a rival — the set
//                              get_all_paths_between_junctions_in_component collects (reference construct_graph.py:2066-2098)
//                              with new_find_paths_between_nodes (:2292-2342) run once per (start, stop) PAIR of junctions.
//                              The search from a start does not depend on the stop except for where it ends, and a path
//                              without a repeated node ends at the stop the only time it meets it: ONE search per start that
//                              notes every junction it arrives at finds the same paths, J times cheaper.
//   amg_seqs_create            the reads' nucleotide sequences resident in HBM (uploaded once per cleaning run, not once per
//                              sketch call).
//   amg_path_sketch_overlaps   the node sketches (:2148-2158: scaled MinHash, ksize 11, scaled 10, of the stretch of every
//                              read under every occurrence of the node), their unions per path (:1747-1751) and the number of
//                              hashes two paths share (:1775-1786), for every pair the caller lists: windows -> segments ->
//                              k_bs_hash (one wave per segment, straight from the resident bases) -> (path, hash) pairs ->
//                              two stable radix sorts -> one binary search per hash of the first path of a pair.
//
// What decides with these numbers (which path of a pair is the better one, which reads are rewritten) stays host-side
// orchestration over a handful of paths, as in the reference (amira_amd/bubble_popping.py).
#include "amg_kmer.h"

#include <algorithm>

#define NEED_BUILT(c)                                                     \
  do {                                                                    \
    if (!(c)) return amg_fail(AMG_E_ARG, "null ctx");                     \
    if (!(c)->built) return amg_fail(AMG_E_STATE, "amg_build first");     \
    HIPCHK(hipSetDevice((c)->device));                                    \
  } while (0)

static inline unsigned int nblk(long long n, int per) {
  long long b = (n + per - 1) / per;
  return (unsigned int)(b < 1 ? 1 : b);
}

struct BubbleState {
  // amg_junction_paths -> amg_get_junction_paths
  std::vector<int> j_node;
  std::vector<signed char> j_dir;
  std::vector<int> p_start;
  std::vector<long long> p_off;
  std::vector<int> p_node;
  std::vector<signed char> p_dir;
  DevBuf jflag, jpos, jrows, cnt_rec, cnt_int, rec_base, int_base, rec_stop, rec_off, pool_node, pool_dir;
  // amg_path_sketch_overlaps
  DevBuf path_off, path_node, pair_a, pair_b, ncnt, noff, ncur, nlist, segs, out_p, out_h, srt_h, srt_p, srt_i, iota, h2,
      size, pstart, common, row_seq, seg_cnt, seg_base;
};

void bubbles_release(amg_ctx* c) {
  if (!c->bub) return;
  BubbleState* b = c->bub;
  DevBuf* all[] = {&b->jflag, &b->jpos, &b->jrows, &b->cnt_rec, &b->cnt_int, &b->rec_base, &b->int_base, &b->rec_stop,
                   &b->rec_off, &b->pool_node, &b->pool_dir, &b->path_off, &b->path_node, &b->pair_a, &b->pair_b, &b->ncnt,
                   &b->noff, &b->ncur, &b->nlist, &b->segs, &b->out_p, &b->out_h, &b->srt_h, &b->srt_p, &b->srt_i, &b->iota,
                   &b->h2, &b->size, &b->pstart, &b->common, &b->row_seq, &b->seg_cnt, &b->seg_base};
  for (DevBuf* d : all) d->release();
  delete b;
  c->bub = nullptr;
}

static BubbleState* bub_of(amg_ctx* c) {
  if (!c->bub) c->bub = new BubbleState();
  return c->bub;
}

// ------------------------------------------------------------------ junctions (:2252-2265 identify_potential_bubble_starts)
// row 2n (2n + 1) of a live node with more than one live edge in its forward (backward) list; rows ascend = the nodes in
// the reference's insertion order, a node's forward side before its backward side
__global__ void k_bj_flag(GView g, long long rows, unsigned char* __restrict__ flag) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  flag[r] = (g.n_alive[r >> 1] && g.lrows[r].y > 1) ? 1 : 0;
}

__global__ void k_bj_rows(const unsigned char* __restrict__ flag, const long long* __restrict__ pos, long long rows,
                          int* __restrict__ jrows) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows && flag[r]) jrows[pos[r]] = (int)r;
}

// live edges from node a (either list) to node b
__device__ __forceinline__ int bj_edges_between(const GView& g, int a, int b) {
  int n = 0;
  for (int side = 0; side < 2; ++side) {
    const int4 rw = g.lrows[2ll * a + side];
    for (int i = 0; i < rw.y; ++i) n += (g.lent[rw.x + i].x == b) ? 1 : 0;
  }
  return n;
}

#define BJ_MULTI 1ull    // a path ends at a junction over a pair of nodes with more than one edge between them
#define BJ_BUDGET 2ull   // a search was abandoned
#define BJ_STEPS (1ull << 24)

// One WAVE per start junction, the search run cooperatively as in amg_passes.hip (dfs_paths_wave): stack level d lives in
// lane d.  A node entered at depth d >= 1 with L = d + 1 <= distance nodes on the path that is a junction on the side the
// path arrives at is a record {stop junction, the L nodes and directions}; EMIT = false counts records and path nodes,
// EMIT = true writes them behind the start's share of the pools — in search order, which is the order the reference's
// per-stop searches return them in.  Nodes are expanded while L < distance (a longer path ends nowhere: :2305).
template <bool EMIT>
__global__ __launch_bounds__(64) void k_bj_dfs(GView g, const int* __restrict__ jrows, long long J,
                                               const unsigned char* __restrict__ jflag, const long long* __restrict__ jpos,
                                               int distance, long long* __restrict__ cnt_rec, long long* __restrict__ cnt_int,
                                               const long long* __restrict__ rec_base, const long long* __restrict__ int_base,
                                               int* __restrict__ rec_stop, long long* __restrict__ rec_off,
                                               int* __restrict__ pool_node, signed char* __restrict__ pool_dir,
                                               unsigned long long* __restrict__ flags) {
  const long long j = blockIdx.x;
  if (j >= J) return;
  const int lane = threadIdx.x;
  const int row0 = jrows[j];
  int my_node = 0, my_dir = 0, my_cur = 0, my_lim = 0, my_off = 0;
  if (lane == 0) {
    my_node = row0 >> 1;
    my_dir = (row0 & 1) ? -1 : 1;
  }
  int depth = 0;
  bool entering = true;
  long long n_rec = 0, n_int = 0;
  unsigned long long steps = 0, bad = 0;
  while (depth >= 0) {
    const int d = __builtin_amdgcn_readfirstlane(depth);
    if (entering) {
      entering = false;
      if (++steps > BJ_STEPS) {
        bad |= BJ_BUDGET;
        break;
      }
      const int L = d + 1;
      const int cur_node = __builtin_amdgcn_readlane(my_node, d);
      const int cur_dir = __builtin_amdgcn_readlane(my_dir, d);
      if (d >= 1 && L <= distance && (jflag[2ll * cur_node] | jflag[2ll * cur_node + 1])) {
        // the reference asks for THE edge between the last two nodes of every path that ends at a junction node
        // (:2086, :1515-1523) and fails when there are several
        const int prev = __builtin_amdgcn_readlane(my_node, d - 1);
        if (bj_edges_between(g, prev, cur_node) > 1 || bj_edges_between(g, cur_node, prev) > 1) bad |= BJ_MULTI;
        // arriving with direction +1 is arriving through the node's backward side (:1523: -1 x the edge's target direction)
        const long long arow = 2ll * cur_node + (cur_dir == 1 ? 1 : 0);
        if (jflag[arow]) {
          if (EMIT) {
            const long long ri = rec_base[j] + n_rec, io = int_base[j] + n_int;
            if (lane == 0) {
              rec_stop[ri] = (int)jpos[arow];
              rec_off[ri] = io;
            }
            if (lane < L) {
              pool_node[io + lane] = my_node;
              pool_dir[io + lane] = (signed char)my_dir;
            }
          }
          ++n_rec;
          n_int += L;
        }
      }
      if (L >= distance) {
        --depth;
        continue;
      }
      const int4 rw = g.lrows[2ll * cur_node + (cur_dir == 1 ? 0 : 1)];  // uniform address
      if (lane == d) {
        my_cur = 0;
        my_lim = rw.y;
        my_off = rw.x;
      }
    }
    int cur = __builtin_amdgcn_readlane(my_cur, d);
    const int lim = __builtin_amdgcn_readlane(my_lim, d);
    const int row_off = __builtin_amdgcn_readlane(my_off, d);
    bool pushed = false;
    while (cur < lim) {
      const int2 ent = g.lent[row_off + cur];  // uniform address
      ++cur;
      const int t = __builtin_amdgcn_readfirstlane(ent.x);
      const int td = __builtin_amdgcn_readfirstlane(ent.y);
      if (__ballot(lane <= d && my_node == t) != 0ull) continue;  // no node twice on a path (:2327)
      if (lane == d) my_cur = cur;
      if (lane == d + 1) {
        my_node = t;
        my_dir = td;
      }
      ++depth;
      entering = true;
      pushed = true;
      break;
    }
    if (!pushed) --depth;
  }
  if (lane == 0) {
    if (!EMIT) {
      cnt_rec[j] = n_rec;
      cnt_int[j] = n_int;
    }
    if (bad) atomicOr(flags, bad);
  }
}

extern "C" int amg_junction_paths(amg_ctx* c, int32_t max_distance, int64_t* sizes) {
  NEED_BUILT(c);
  if (!sizes) return amg_fail(AMG_E_ARG, "null argument");
  if (max_distance < 2 || max_distance > 64) return amg_fail(AMG_E_ARG, "max_distance must be in [2, 64]");
  BubbleState* b = bub_of(c);
  b->j_node.clear(); b->j_dir.clear(); b->p_start.clear(); b->p_off.assign(1, 0); b->p_node.clear(); b->p_dir.clear();
  sizes[0] = sizes[1] = sizes[2] = sizes[3] = 0;
  const long long D = c->n_nodes, rows = 2 * D;
  if (D == 0) return AMG_OK;
  hipStream_t st = c->stream;
  stages_reset(c);
  AMGCHK(ensure_live_adj(c));
  stage_begin(c, "junction_paths");
  const GView g = make_view(c);
  AMGCHK(b->jflag.ensure((size_t)rows + 64));
  AMGCHK(b->jpos.ensure((size_t)(rows + 2) * sizeof(long long)));
  unsigned long long* flags = c->status.as<unsigned long long>() + ST_MISC;
  {
    ClearList cl;
    cl.add(flags, sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  hipLaunchKernelGGL(k_bj_flag, dim3(nblk(rows, 256)), dim3(256), 0, st, g, rows, b->jflag.as<unsigned char>());
  AMGCHK(prim_exscan_bytes_set(c, b->jflag.as<unsigned char>(), b->jpos.as<long long>(), (size_t)rows));
  long long J = 0;
  {
    FetchList l;
    l.add(b->jpos.as<long long>() + rows);
    AMGCHK(fetch(c, l, reinterpret_cast<unsigned long long*>(&J)));
  }
  if (J == 0) {
    stage_end(c);
    return AMG_OK;
  }
  AMGCHK(b->jrows.ensure((size_t)(J + 1) * sizeof(int)));
  AMGCHK(b->cnt_rec.ensure((size_t)(J + 1) * sizeof(long long)));
  AMGCHK(b->cnt_int.ensure((size_t)(J + 1) * sizeof(long long)));
  AMGCHK(b->rec_base.ensure((size_t)(J + 2) * sizeof(long long)));
  AMGCHK(b->int_base.ensure((size_t)(J + 2) * sizeof(long long)));
  hipLaunchKernelGGL(k_bj_rows, dim3(nblk(rows, 256)), dim3(256), 0, st, b->jflag.as<unsigned char>(), b->jpos.as<long long>(),
                     rows, b->jrows.as<int>());
  hipLaunchKernelGGL(k_bj_dfs<false>, dim3((unsigned int)J), dim3(64), 0, st, g, b->jrows.as<int>(), J,
                     b->jflag.as<unsigned char>(), b->jpos.as<long long>(), (int)max_distance, b->cnt_rec.as<long long>(),
                     b->cnt_int.as<long long>(), (const long long*)nullptr, (const long long*)nullptr, (int*)nullptr,
                     (long long*)nullptr, (int*)nullptr, (signed char*)nullptr, flags);
  AMGCHK(prim_exscan_i64_pair(c, b->cnt_rec.as<long long>(), b->rec_base.as<long long>(), b->cnt_int.as<long long>(),
                              b->int_base.as<long long>(), (size_t)J));
  unsigned long long got[3] = {0, 0, 0};
  {
    FetchList l;
    l.add(b->rec_base.as<long long>() + J);
    l.add(b->int_base.as<long long>() + J);
    l.add(flags);
    AMGCHK(fetch(c, l, got));
  }
  const long long R = (long long)got[0], N = (long long)got[1];
  sizes[3] = (int64_t)got[2];
  std::vector<int> jrows((size_t)J);
  HIPCHK(hipMemcpyAsync(jrows.data(), b->jrows.p, (size_t)J * sizeof(int), hipMemcpyDeviceToHost, st));
  std::vector<long long> rec_base((size_t)J + 1), rec_off((size_t)R + 1);
  std::vector<int> rec_stop((size_t)R), pool_node((size_t)N);
  std::vector<signed char> pool_dir((size_t)N);
  if (R > 0 && !(got[2] & BJ_BUDGET)) {
    AMGCHK(b->rec_stop.ensure((size_t)(R + 1) * sizeof(int)));
    AMGCHK(b->rec_off.ensure((size_t)(R + 1) * sizeof(long long)));
    AMGCHK(b->pool_node.ensure((size_t)(N + 1) * sizeof(int)));
    AMGCHK(b->pool_dir.ensure((size_t)N + 64));
    hipLaunchKernelGGL(k_bj_dfs<true>, dim3((unsigned int)J), dim3(64), 0, st, g, b->jrows.as<int>(), J,
                       b->jflag.as<unsigned char>(), b->jpos.as<long long>(), (int)max_distance, (long long*)nullptr,
                       (long long*)nullptr, b->rec_base.as<long long>(), b->int_base.as<long long>(), b->rec_stop.as<int>(),
                       b->rec_off.as<long long>(), b->pool_node.as<int>(), b->pool_dir.as<signed char>(), flags);
    HIPCHK(hipMemcpyAsync(rec_base.data(), b->rec_base.p, (size_t)(J + 1) * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(rec_off.data(), b->rec_off.p, (size_t)R * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(rec_stop.data(), b->rec_stop.p, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pool_node.data(), b->pool_node.p, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pool_dir.data(), b->pool_dir.p, (size_t)N, hipMemcpyDeviceToHost, st));
  }
  stage_end(c);
  HIPCHK(hipStreamSynchronize(st));
  b->j_node.resize((size_t)J);
  b->j_dir.resize((size_t)J);
  for (long long j = 0; j < J; ++j) {
    b->j_node[(size_t)j] = jrows[(size_t)j] >> 1;
    b->j_dir[(size_t)j] = (jrows[(size_t)j] & 1) ? -1 : 1;
  }
  if (R > 0 && !(got[2] & BJ_BUDGET)) {
    rec_off[(size_t)R] = N;
    // per start: its records by stop (in the order of the junction list, as the reference's inner loop goes), search
    // order kept inside a stop; a stop reached by one path only is no bubble (:2092)
    std::vector<long long> idx;
    for (long long j = 0; j < J; ++j) {
      const long long lo = rec_base[(size_t)j], hi = rec_base[(size_t)j + 1];
      if (hi - lo < 2) continue;
      idx.resize((size_t)(hi - lo));
      for (long long i = lo; i < hi; ++i) idx[(size_t)(i - lo)] = i;
      std::stable_sort(idx.begin(), idx.end(), [&](long long x, long long y) { return rec_stop[(size_t)x] < rec_stop[(size_t)y]; });
      size_t a = 0;
      while (a < idx.size()) {
        size_t e = a + 1;
        while (e < idx.size() && rec_stop[(size_t)idx[e]] == rec_stop[(size_t)idx[a]]) ++e;
        if (e - a > 1)
          for (size_t q = a; q < e; ++q) {
            const long long ri = idx[q];
            b->p_start.push_back((int)j);
            for (long long t = rec_off[(size_t)ri]; t < rec_off[(size_t)ri + 1]; ++t) {
              b->p_node.push_back(pool_node[(size_t)t]);
              b->p_dir.push_back(pool_dir[(size_t)t]);
            }
            b->p_off.push_back((long long)b->p_node.size());
          }
        a = e;
      }
    }
  }
  sizes[0] = J;
  sizes[1] = (int64_t)b->p_start.size();
  sizes[2] = (int64_t)b->p_node.size();
  return AMG_OK;
}

extern "C" int amg_get_junction_paths(amg_ctx* c, int32_t* junction_node, int8_t* junction_dir, int32_t* path_start,
                                      int64_t* path_off, int32_t* path_node, int8_t* path_dir) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->bub) return amg_fail(AMG_E_STATE, "amg_junction_paths first");
  const BubbleState* b = c->bub;
  if (junction_node && !b->j_node.empty()) memcpy(junction_node, b->j_node.data(), b->j_node.size() * sizeof(int));
  if (junction_dir && !b->j_dir.empty()) memcpy(junction_dir, b->j_dir.data(), b->j_dir.size());
  if (path_start && !b->p_start.empty()) memcpy(path_start, b->p_start.data(), b->p_start.size() * sizeof(int));
  if (path_off) memcpy(path_off, b->p_off.data(), b->p_off.size() * sizeof(long long));
  if (path_node && !b->p_node.empty()) memcpy(path_node, b->p_node.data(), b->p_node.size() * sizeof(int));
  if (path_dir && !b->p_dir.empty()) memcpy(path_dir, b->p_dir.data(), b->p_dir.size());
  return AMG_OK;
}

// ------------------------------------------------------------------ the reads' bases, resident
struct amg_seqs {
  int device = 0;
  DevBuf bases, off;
  int64_t n = 0, total = 0;
};

extern "C" int amg_seqs_create(int32_t device, const char* const* seq, const int64_t* len, int64_t n, amg_seqs** out) {
  if (!out || n < 0 || (n > 0 && (!seq || !len))) return amg_fail(AMG_E_ARG, "bad argument");
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  std::vector<long long> off((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    if (len[i] < 0 || (len[i] > 0 && !seq[i])) return amg_fail(AMG_E_ARG, "sequence %lld: bad length or pointer", (long long)i);
    off[(size_t)i + 1] = off[(size_t)i] + len[i];
  }
  amg_seqs* s = new amg_seqs();
  s->device = device;
  s->n = n;
  s->total = off[(size_t)n];
  auto fail = [&](int r) {
    s->bases.release();
    s->off.release();
    delete s;
    return r;
  };
  if (s->bases.ensure((size_t)s->total + 256) != AMG_OK) return fail(AMG_E_NOMEM);
  if (s->off.ensure((size_t)(n + 2) * sizeof(long long)) != AMG_OK) return fail(AMG_E_NOMEM);
  // two pinned buffers take turns: the host fills one while the other is on its way
  const size_t CH = (size_t)32 << 20;
  char* pin[2] = {nullptr, nullptr};
  hipStream_t st = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipHostMalloc(reinterpret_cast<void**>(&pin[i]), CH, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  int which = 0;
  bool in_flight[2] = {false, false};
  size_t fill = 0;
  long long dst = 0;
  auto flush = [&]() -> hipError_t {
    if (fill == 0) return hipSuccess;
    hipError_t r = hipMemcpyAsync(s->bases.as<char>() + dst, pin[which], fill, hipMemcpyHostToDevice, st);
    if (r != hipSuccess) return r;
    r = hipEventRecord(ev[which], st);
    in_flight[which] = true;
    dst += (long long)fill;
    fill = 0;
    which ^= 1;
    if (r == hipSuccess && in_flight[which]) {
      r = hipEventSynchronize(ev[which]);
      in_flight[which] = false;
    }
    return r;
  };
  for (int64_t i = 0; i < n && e == hipSuccess; ++i) {
    size_t done = 0;
    const size_t L = (size_t)len[i];
    while (done < L && e == hipSuccess) {
      const size_t take = std::min(L - done, CH - fill);
      memcpy(pin[which] + fill, seq[i] + done, take);
      fill += take;
      done += take;
      if (fill == CH) e = flush();
    }
  }
  if (e == hipSuccess) e = flush();
  if (e == hipSuccess) e = hipMemcpyAsync(s->off.p, off.data(), (size_t)(n + 1) * sizeof(long long), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  for (int i = 0; i < 2; ++i) {
    if (pin[i]) (void)hipHostFree(pin[i]);
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  }
  if (st) (void)hipStreamDestroy(st);
  if (e != hipSuccess) return fail(amg_fail(AMG_E_HIP, "amg_seqs_create: %s", hipGetErrorString(e)));
  *out = s;
  return AMG_OK;
}

extern "C" int amg_seqs_destroy(amg_seqs* s) {
  if (!s) return AMG_OK;
  (void)hipSetDevice(s->device);
  s->bases.release();
  s->off.release();
  delete s;
  return AMG_OK;
}

// ------------------------------------------------------------------ sketches of paths and what two of them share
struct BsSeg {
  long long src;  // first base in the resident stream
  int len;
  int node;
};
static_assert(sizeof(BsSeg) == 16, "segment record");

#define BS_NEG_POS 1ull  // a gene position below zero: Python's slice would count from the end of the read

__global__ void k_bs_mark(const int* __restrict__ path_node, long long n, unsigned int* __restrict__ ncnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&ncnt[path_node[i]], 1u);
}

// the paths of every node: path p lists node x at some place, x lists p at some place
__global__ void k_bs_fill(const long long* __restrict__ path_off, const int* __restrict__ path_node, long long n_paths,
                          const long long* __restrict__ noff, unsigned int* __restrict__ ncur, int* __restrict__ nlist) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_paths) return;
  for (long long i = path_off[p]; i < path_off[p + 1]; ++i) {
    const int x = path_node[i];
    nlist[noff[x] + atomicAdd(&ncur[x], 1u)] = (int)p;
  }
}

// the windows of the reads that sit on a node of some path: sequence[start of the window's first gene : end of its last
// gene + 1] (:2154-2157) with the clipping of a Python slice.  FILL = false counts them.
template <bool FILL>
__global__ __launch_bounds__(256) void k_bs_segs(const int* __restrict__ tok_node, long long n_tokens,
                                                 const long long* __restrict__ read_off, long long n_reads, int k,
                                                 const long long* __restrict__ gs, const long long* __restrict__ ge,
                                                 const unsigned int* __restrict__ ncnt, const int* __restrict__ row_seq,
                                                 const long long* __restrict__ seq_off, long long n_seqs,
                                                 unsigned long long* __restrict__ ctr, BsSeg* __restrict__ segs,
                                                 unsigned long long* __restrict__ flags) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool mine = false;
  int node = -1;
  if (t < n_tokens) {
    node = tok_node[t];
    mine = node >= 0 && ncnt[node] != 0u;
  }
  const unsigned long long vote = __ballot(mine);
  if (vote == 0ull) return;
  const int lane = threadIdx.x & 63;
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(ctr, (unsigned long long)__popcll(vote));
  base = (unsigned long long)__shfl((long long)base, 0);
  if (!FILL || !mine) return;
  long long lo = 0, hi = n_reads;  // read_off[lo] <= t < read_off[hi]
  while (hi - lo > 1) {
    const long long mid = (lo + hi) >> 1;
    if (read_off[mid] <= t) lo = mid; else hi = mid;
  }
  const long long si = row_seq ? (long long)row_seq[lo] : lo;
  BsSeg s;
  s.node = node;
  s.src = 0;
  s.len = 0;
  if (si >= 0 && si < n_seqs) {
    const long long a = gs[t], b = ge[t + k - 1] + 1;
    if (a < 0 || b < 0) atomicOr(flags, BS_NEG_POS);
    const long long L = seq_off[si + 1] - seq_off[si];
    const long long x = a < L ? a : L, y = b < L ? b : L;
    if (a >= 0 && b >= 0 && y > x) {
      s.src = seq_off[si] + x;
      s.len = (int)(y - x > 0x7fffffffll ? 0x7fffffffll : y - x);
    }
  } else {
    atomicOr(flags, BS_NEG_POS << 1);
  }
  segs[base + (unsigned long long)__popcll(vote & ((1ull << lane) - 1ull))] = s;
}

#define BS_CHUNK 1024
#define BS_MAX_K KM_MAX_K
#define BS_WPB 4

// One WAVE per segment: the segment goes through the wave's slab of LDS a chunk at a time (k - 1 bases of overlap), lane
// i hashes the k-mers that start at i, i + 64, ...; a hash that passes the scaled cut is one (path, hash) pair for every
// path that lists the segment's node.  EMIT = false counts the pairs of every segment (seg_cnt), EMIT = true writes them
// behind the segment's own offset (seg_base = the prefix sums of the counts): no counter is shared between waves — one
// returning atomic per wave and round on a single word was five sixths of this kernel's time.
template <bool EMIT, int NW>  // NW: words of a k-mer, (ksize + 7) / 8 (amg_kmer.h)
__global__ __launch_bounds__(64 * BS_WPB) void k_bs_hash(const BsSeg* __restrict__ segs, long long n_segs,
                                                         const unsigned char* __restrict__ bases, int ksize,
                                                         unsigned long long max_hash, const long long* __restrict__ noff,
                                                         const int* __restrict__ nlist, long long* __restrict__ seg_cnt,
                                                         const long long* __restrict__ seg_base,
                                                         long long cap, unsigned int* __restrict__ out_p,
                                                         unsigned long long* __restrict__ out_h) {
  __shared__ __attribute__((aligned(8))) unsigned char s_b[BS_WPB][BS_CHUNK + BS_MAX_K + 24];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long si = (long long)blockIdx.x * BS_WPB + wv;
  if (si >= n_segs) return;
  const BsSeg sg = segs[si];
  const long long n0 = noff[sg.node];
  const int n_paths = (int)(noff[sg.node + 1] - n0);
  unsigned long long counted = EMIT ? (unsigned long long)seg_base[si] : 0ull;
  unsigned char* sb = s_b[wv];
  for (int c0 = 0; c0 + ksize <= sg.len; c0 += BS_CHUNK) {
    const int have = min(sg.len - c0, BS_CHUNK + ksize - 1);
    for (int i = lane; i < have + 24; i += 64) sb[i] = i < have ? km_stage(bases[sg.src + c0 + i]) : (unsigned char)0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int starts = have - ksize + 1;
    for (int i0 = 0; i0 < starts; i0 += 64) {
      const int i = i0 + lane;
      bool keep = false;
      unsigned long long hv = 0;
      if (i < starts) keep = km_canonical_hash<NW>(sb, i, ksize, &hv) && hv <= max_hash;
      const unsigned long long vote = __ballot(keep);
      if (vote == 0ull) continue;
      const unsigned long long n_keep = (unsigned long long)__popcll(vote);
      const unsigned long long base = counted;
      counted += n_keep * (unsigned long long)n_paths;
      if (!EMIT) continue;
      if (keep) {
        unsigned long long o = base + (unsigned long long)__popcll(vote & ((1ull << lane) - 1ull)) * (unsigned long long)n_paths;
        for (int q = 0; q < n_paths; ++q, ++o)
          if ((long long)o < cap) {
            out_p[o] = (unsigned int)nlist[n0 + q];
            out_h[o] = hv;
          }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (!EMIT && lane == 0) seg_cnt[si] = (long long)counted;
}

__global__ void k_bs_iota(unsigned int* __restrict__ v, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (unsigned int)i;
}

// after the two sorts: pairs in (path, hash) order.  The first pair of a run of equal ones stands for the hash in the
// path's sketch; a path's sketch size is the number of its runs.
__global__ void k_bs_unique(const unsigned int* __restrict__ sp, const unsigned int* __restrict__ si,
                            const unsigned long long* __restrict__ h1, long long n, unsigned long long* __restrict__ h2,
                            unsigned long long* __restrict__ size) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool first = false;
  unsigned int p = 0xffffffffu;
  if (i < n) {
    const unsigned long long h = h1[si[i]];
    h2[i] = h;
    p = sp[i];
    first = i == 0 || sp[i - 1] != p || h1[si[i - 1]] != h;
  }
  // a path's pairs are thousands in a row: nearly every wave sits inside ONE path and adds its count once
  const unsigned int p0 = (unsigned int)__shfl((int)p, 0);
  const unsigned long long firsts = __ballot(first);
  if (__ballot(p != p0 && i < n) == 0ull) {
    if ((threadIdx.x & 63) == 0 && firsts) atomicAdd(&size[p0], (unsigned long long)__popcll(firsts));
  } else if (first) {
    atomicAdd(&size[p], 1ull);
  }
}

__global__ void k_bs_pstart(const unsigned int* __restrict__ sp, long long n, long long n_paths, long long* __restrict__ pstart) {
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_paths) return;
  long long lo = 0, hi = n;  // first i with sp[i] >= p
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)sp[mid] < p) lo = mid + 1; else hi = mid;
  }
  pstart[p] = lo;
}

// |sketch(a) & sketch(b)| for one pair per workgroup: every distinct hash of a is looked up in b's stretch
__global__ __launch_bounds__(256) void k_bs_common(const int* __restrict__ pair_a, const int* __restrict__ pair_b,
                                                   const long long* __restrict__ pstart, const unsigned long long* __restrict__ h2,
                                                   unsigned long long* __restrict__ common) {
  __shared__ unsigned long long s_sum[4];
  const long long q = blockIdx.x;
  const int a = pair_a[q], b = pair_b[q];
  const long long a0 = pstart[a], a1 = pstart[a + 1], b0 = pstart[b], b1 = pstart[b + 1];
  unsigned long long mine = 0;
  for (long long i = a0 + threadIdx.x; i < a1; i += 256) {
    const unsigned long long h = h2[i];
    if (i > a0 && h2[i - 1] == h) continue;
    long long lo = b0, hi = b1;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (h2[mid] < h) lo = mid + 1; else hi = mid;
    }
    if (lo < b1 && h2[lo] == h) ++mine;
  }
  for (int o = 32; o > 0; o >>= 1) mine += (unsigned long long)__shfl_down((long long)mine, o);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) common[q] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

extern "C" int amg_path_sketch_overlaps(amg_ctx* c, const amg_seqs* seqs, const int32_t* row_to_seq, int32_t ksize,
                                        uint64_t scaled, int64_t n_paths, const int64_t* path_off, const int32_t* path_node,
                                        int64_t n_pairs, const int32_t* pair_a, const int32_t* pair_b, int64_t* sketch_size,
                                        int64_t* common) {
  NEED_BUILT(c);
  if (!seqs || n_paths < 0 || n_pairs < 0 || (n_paths > 0 && (!path_off || !sketch_size)) ||
      (n_pairs > 0 && (!pair_a || !pair_b || !common)))
    return amg_fail(AMG_E_ARG, "bad argument");
  if (seqs->device != c->device) return amg_fail(AMG_E_ARG, "the sequences live on another device");
  if (ksize < 1 || ksize > BS_MAX_K) return amg_fail(AMG_E_ARG, "ksize must be in [1, %d]", BS_MAX_K);
  if (scaled == 0) return amg_fail(AMG_E_ARG, "scaled must be >= 1");
  if (!c->have_pos || !c->pos_identity) return amg_fail(AMG_E_STATE, "gene positions of the graph's reads are needed (amg_set_positions)");
  if (n_paths == 0) return AMG_OK;
  const long long D = c->n_nodes, T = c->n_tokens, R = c->n_reads;
  const long long N = path_off[n_paths];
  if (path_off[0] != 0 || N < 0 || (N > 0 && !path_node)) return amg_fail(AMG_E_ARG, "bad path offsets");
  for (int64_t p = 0; p < n_paths; ++p)
    if (path_off[p + 1] < path_off[p]) return amg_fail(AMG_E_ARG, "path offsets not monotone");
  for (long long i = 0; i < N; ++i)
    if (path_node[i] < 0 || path_node[i] >= D) return amg_fail(AMG_E_ARG, "path node %d outside the graph", path_node[i]);
  for (int64_t q = 0; q < n_pairs; ++q)
    if (pair_a[q] < 0 || pair_a[q] >= n_paths || pair_b[q] < 0 || pair_b[q] >= n_paths)
      return amg_fail(AMG_E_ARG, "pair %lld names a path that is not there", (long long)q);
  if (row_to_seq == nullptr && seqs->n < R) return amg_fail(AMG_E_ARG, "fewer sequences than reads");
  // sourmash's cut (amg_minhash.hip)
  unsigned long long max_hash = ~0ull;
  if (scaled > 1) {
    const double qd = 18446744073709551616.0 / (double)scaled;
    max_hash = qd >= 18446744073709551615.0 ? ~0ull : (unsigned long long)qd;
  }
  BubbleState* b = bub_of(c);
  hipStream_t st = c->stream;
  stages_reset(c);
  stage_begin(c, "path_sketches");
  for (int64_t p = 0; p < n_paths; ++p) sketch_size[p] = 0;
  for (int64_t q = 0; q < n_pairs; ++q) common[q] = 0;
  AMGCHK(b->path_off.ensure((size_t)(n_paths + 2) * sizeof(long long)));
  AMGCHK(b->path_node.ensure((size_t)(N + 2) * sizeof(int)));
  AMGCHK(b->pair_a.ensure((size_t)(n_pairs + 2) * sizeof(int)));
  AMGCHK(b->pair_b.ensure((size_t)(n_pairs + 2) * sizeof(int)));
  AMGCHK(b->ncnt.ensure((size_t)(D + 2) * sizeof(unsigned int)));
  AMGCHK(b->ncur.ensure((size_t)(D + 2) * sizeof(unsigned int)));
  AMGCHK(b->noff.ensure((size_t)(D + 2) * sizeof(long long)));
  AMGCHK(b->nlist.ensure((size_t)(N + 2) * sizeof(int)));
  AMGCHK(b->size.ensure((size_t)(n_paths + 2) * sizeof(unsigned long long)));
  AMGCHK(b->pstart.ensure((size_t)(n_paths + 2) * sizeof(long long)));
  AMGCHK(b->common.ensure((size_t)(n_pairs + 2) * sizeof(unsigned long long)));
  HIPCHK(hipMemcpyAsync(b->path_off.p, path_off, (size_t)(n_paths + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
  if (N) HIPCHK(hipMemcpyAsync(b->path_node.p, path_node, (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
  if (n_pairs) {
    HIPCHK(hipMemcpyAsync(b->pair_a.p, pair_a, (size_t)n_pairs * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(b->pair_b.p, pair_b, (size_t)n_pairs * sizeof(int), hipMemcpyHostToDevice, st));
  }
  const int* d_row_seq = nullptr;
  if (row_to_seq) {
    AMGCHK(b->row_seq.ensure((size_t)(R + 2) * sizeof(int)));
    HIPCHK(hipMemcpyAsync(b->row_seq.p, row_to_seq, (size_t)R * sizeof(int), hipMemcpyHostToDevice, st));
    d_row_seq = b->row_seq.as<int>();
  }
  unsigned long long* ctr = c->status.as<unsigned long long>() + ST_COMPACT_A;  // [0] segments, [1] pairs
  unsigned long long* flags = c->status.as<unsigned long long>() + ST_MISC;
  {
    ClearList cl;
    cl.add(b->ncnt.p, (size_t)(D + 2) * sizeof(unsigned int));
    cl.add(b->ncur.p, (size_t)(D + 2) * sizeof(unsigned int));
    cl.add(b->size.p, (size_t)(n_paths + 2) * sizeof(unsigned long long));
    cl.add(ctr, 2 * sizeof(unsigned long long));
    cl.add(flags, sizeof(unsigned long long));
    AMGCHK(clear_many(c, cl));
  }
  auto finish = [&]() -> int {  // sizes and overlaps to the caller
    HIPCHK(hipMemcpyAsync(sketch_size, b->size.p, (size_t)n_paths * sizeof(long long), hipMemcpyDeviceToHost, st));
    stage_end(c);
    HIPCHK(hipStreamSynchronize(st));
    return AMG_OK;
  };
  if (N == 0) return finish();
  hipLaunchKernelGGL(k_bs_mark, dim3(nblk(N, 256)), dim3(256), 0, st, b->path_node.as<int>(), N, b->ncnt.as<unsigned int>());
  AMGCHK(prim_exscan_u32_to_i64(c, b->ncnt.as<unsigned int>(), b->noff.as<long long>(), (size_t)D + 1));  // (ncnt[D] = 0: noff[D] = the sum)
  hipLaunchKernelGGL(k_bs_fill, dim3(nblk(n_paths, 64)), dim3(64), 0, st, b->path_off.as<long long>(), b->path_node.as<int>(),
                     (long long)n_paths, b->noff.as<long long>(), b->ncur.as<unsigned int>(), b->nlist.as<int>());
  const long long* gs = c->gene_start.as<long long>();
  const long long* ge = c->gene_end.as<long long>();
  hipLaunchKernelGGL(k_bs_segs<false>, dim3(nblk(T, 256)), dim3(256), 0, st, c->tok_node.as<int>(), T,
                     c->read_off.as<long long>(), R, (int)c->k, gs, ge, b->ncnt.as<unsigned int>(), d_row_seq,
                     seqs->off.as<long long>(), (long long)seqs->n, ctr, (BsSeg*)nullptr, flags);
  unsigned long long n_segs = 0;
  {
    FetchList l;
    l.add(ctr);
    ClearList after;
    after.add(ctr, sizeof(unsigned long long));
    AMGCHK(fetch(c, l, &n_segs, &after));
  }
  if (n_segs == 0) return finish();
  AMGCHK(b->segs.ensure((size_t)(n_segs + 1) * sizeof(BsSeg)));
  hipLaunchKernelGGL(k_bs_segs<true>, dim3(nblk(T, 256)), dim3(256), 0, st, c->tok_node.as<int>(), T,
                     c->read_off.as<long long>(), R, (int)c->k, gs, ge, b->ncnt.as<unsigned int>(), d_row_seq,
                     seqs->off.as<long long>(), (long long)seqs->n, ctr, b->segs.as<BsSeg>(), flags);
  // pairs per segment -> where every segment's pairs go
  AMGCHK(b->seg_cnt.ensure((size_t)(n_segs + 2) * sizeof(long long)));
  AMGCHK(b->seg_base.ensure((size_t)(n_segs + 2) * sizeof(long long)));
  {
    ClearList cl;
    cl.add(b->seg_cnt.as<long long>() + n_segs, sizeof(long long));
    AMGCHK(clear_many(c, cl));
  }
  const int nw = ((int)ksize + 7) / 8;
  auto count_kernel = nw == 1 ? k_bs_hash<false, 1> : nw == 2 ? k_bs_hash<false, 2> : nw == 3 ? k_bs_hash<false, 3> : k_bs_hash<false, 4>;
  auto emit_kernel = nw == 1 ? k_bs_hash<true, 1> : nw == 2 ? k_bs_hash<true, 2> : nw == 3 ? k_bs_hash<true, 3> : k_bs_hash<true, 4>;
  hipLaunchKernelGGL(count_kernel, dim3(nblk((long long)n_segs, BS_WPB)), dim3(64 * BS_WPB), 0, st, b->segs.as<BsSeg>(),
                     (long long)n_segs, seqs->bases.as<unsigned char>(), (int)ksize, max_hash, b->noff.as<long long>(),
                     b->nlist.as<int>(), b->seg_cnt.as<long long>(), (const long long*)nullptr, 0ll, (unsigned int*)nullptr,
                     (unsigned long long*)nullptr);
  AMGCHK(prim_exscan_i64(c, b->seg_cnt.as<long long>(), b->seg_base.as<long long>(), (size_t)n_segs + 1));  // (seg_cnt[n_segs] = 0)
  unsigned long long got[3] = {0, 0, 0};
  {
    FetchList l;
    l.add(b->seg_base.as<long long>() + n_segs);
    l.add(flags);
    l.add(ctr);
    AMGCHK(fetch(c, l, got));
  }
  if (got[2] != n_segs) {  // (both passes over the windows must have seen the same)
    stage_end(c);
    return amg_fail(AMG_E_OVERFLOW, "path sketches: %llu of %llu segments written", got[2], n_segs);
  }
  if (got[1] & BS_NEG_POS) {
    stage_end(c);
    return amg_fail(AMG_E_ARG, "a gene position below zero");
  }
  if (got[1] & (BS_NEG_POS << 1)) {
    stage_end(c);
    return amg_fail(AMG_E_ARG, "a read without a sequence (row_to_seq)");
  }
  const long long M = (long long)got[0];
  if (M == 0) return finish();
  long long max_pairs = 1ll << 30;  // (40 bytes of buffers per pair; the sorts index pairs with 32 bits)
  if (const char* e = getenv("AMG_TEST_SKETCH_PAIRS")) max_pairs = atoll(e);  // test hook: callers split their paths
  if (M >= max_pairs) {
    stage_end(c);
    return amg_fail(AMG_E_NOMEM, "%lld (path, hash) pairs in one call: split the paths", M);
  }
  AMGCHK(b->out_p.ensure((size_t)(M + 1) * sizeof(unsigned int)));
  AMGCHK(b->out_h.ensure((size_t)(M + 1) * sizeof(unsigned long long)));
  AMGCHK(b->srt_h.ensure((size_t)(M + 1) * sizeof(unsigned long long)));
  AMGCHK(b->srt_p.ensure((size_t)(M + 1) * sizeof(unsigned int)));
  AMGCHK(b->srt_i.ensure((size_t)(M + 1) * sizeof(unsigned int)));
  AMGCHK(b->iota.ensure((size_t)(M + 1) * sizeof(unsigned int)));
  AMGCHK(b->h2.ensure((size_t)(M + 1) * sizeof(unsigned long long)));
  hipLaunchKernelGGL(emit_kernel, dim3(nblk((long long)n_segs, BS_WPB)), dim3(64 * BS_WPB), 0, st, b->segs.as<BsSeg>(),
                     (long long)n_segs, seqs->bases.as<unsigned char>(), (int)ksize, max_hash, b->noff.as<long long>(),
                     b->nlist.as<int>(), (long long*)nullptr, b->seg_base.as<long long>(), M, b->out_p.as<unsigned int>(),
                     b->out_h.as<unsigned long long>());
  // (path, hash) order by two stable sorts: by hash, then by path
  AMGCHK(prim_sort_u64_u32(c, b->out_h.as<unsigned long long>(), b->srt_h.as<unsigned long long>(), b->out_p.as<unsigned int>(),
                           b->srt_p.as<unsigned int>(), (size_t)M, 64));
  hipLaunchKernelGGL(k_bs_iota, dim3(nblk(M, 256)), dim3(256), 0, st, b->iota.as<unsigned int>(), M);
  AMGCHK(prim_sort_u32_u32(c, b->srt_p.as<unsigned int>(), b->out_p.as<unsigned int>(), b->iota.as<unsigned int>(),
                           b->srt_i.as<unsigned int>(), (size_t)M, ilog2_ceil((uint64_t)n_paths + 1) + 1));
  hipLaunchKernelGGL(k_bs_unique, dim3(nblk(M, 256)), dim3(256), 0, st, b->out_p.as<unsigned int>(), b->srt_i.as<unsigned int>(),
                     b->srt_h.as<unsigned long long>(), M, b->h2.as<unsigned long long>(), b->size.as<unsigned long long>());
  hipLaunchKernelGGL(k_bs_pstart, dim3(nblk(n_paths + 1, 256)), dim3(256), 0, st, b->out_p.as<unsigned int>(), M,
                     (long long)n_paths, b->pstart.as<long long>());
  if (n_pairs) {
    hipLaunchKernelGGL(k_bs_common, dim3((unsigned int)n_pairs), dim3(256), 0, st, b->pair_a.as<int>(), b->pair_b.as<int>(),
                       b->pstart.as<long long>(), b->h2.as<unsigned long long>(), b->common.as<unsigned long long>());
    HIPCHK(hipMemcpyAsync(common, b->common.p, (size_t)n_pairs * sizeof(long long), hipMemcpyDeviceToHost, st));
  }
  return finish();
}

// ------------------------------------------------------------------ the alignment of two short gene lists (host)
// needleman_wunsch (construct_graph.py:1433-1480) on interned genes: match 1, mismatch 0, gap -1, borders -index, the
// best of (score, pointer) with the pointers ordered DIAG < LEFT < UP — a tie goes UP, then LEFT.  ops, in alignment
// order: 0 = (x, y), 1 = (x, *), 2 = (*, y); at most n + m of them.  The two paths of a bubble are a few dozen genes:
// this is the host's share of compare_paths (:1566), called once per correction operation.
extern "C" int amg_nw_align(const int32_t* x, int32_t n, const int32_t* y, int32_t m, int8_t* ops, int32_t* n_ops) {
  if (n < 0 || m < 0 || !n_ops || ((n || m) && !ops) || (n && !x) || (m && !y)) return amg_fail(AMG_E_ARG, "bad argument");
  const int W = m + 1;
  std::vector<int> F((size_t)(n + 1) * W);
  std::vector<signed char> P((size_t)(n + 1) * W, 0);
  // F[i + 1][j + 1] = the reference's F[i, j]; its borders are F[i, -1] = -i, F[-1, j] = -j (so F[0, -1] = 0 too)
  F[0] = 0;
  for (int i = 0; i < n; ++i) F[(size_t)(i + 1) * W] = -i;
  for (int j = 0; j < m; ++j) F[(size_t)j + 1] = -j;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const int diag = F[(size_t)i * W + j] + (x[i] == y[j] ? 1 : 0);
      const int left = F[(size_t)i * W + j + 1] - 1;   // F[i - 1, j] - 1: a gene of x against a gap
      const int up = F[(size_t)(i + 1) * W + j] - 1;   // F[i, j - 1] - 1: a gene of y against a gap
      int best = diag;
      signed char ptr = 0;
      if (left >= best) { best = left; ptr = 1; }
      if (up >= best) { best = up; ptr = 2; }
      F[(size_t)(i + 1) * W + j + 1] = best;
      P[(size_t)(i + 1) * W + j + 1] = ptr;
    }
  std::vector<signed char> rev;
  int i = n - 1, j = m - 1;
  while (i >= 0 && j >= 0) {
    const signed char p = P[(size_t)(i + 1) * W + j + 1];
    rev.push_back(p);
    if (p == 0) { --i; --j; } else if (p == 1) { --i; } else { --j; }
  }
  while (i >= 0) { rev.push_back(1); --i; }
  while (j >= 0) { rev.push_back(2); --j; }
  *n_ops = (int32_t)rev.size();
  for (size_t t = 0; t < rev.size(); ++t) ops[t] = rev[rev.size() - 1 - t];
  return AMG_OK;
}
