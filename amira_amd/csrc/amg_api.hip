// amg_api.hip — C ABI plumbing of libamg.so: lifetime, inputs, read-back, timings.
#include <cstdarg>

#include "amg_device.h"

thread_local std::string g_amg_err;

int amg_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_amg_err = buf;
  return code;
}

extern "C" const char* amg_last_error(void) { return g_amg_err.c_str(); }

// ------------------------------------------------------------------ stage timing
void stages_reset(amg_ctx* c) {
  for (auto& s : c->stages) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  c->stages.clear();
}

void stage_begin(amg_ctx* c, const char* name) {
  if (!c->timing) return;
  StageTime s{name, nullptr, nullptr, 0.f};
  (void)hipEventCreate(&s.a);
  (void)hipEventCreate(&s.b);
  (void)hipEventRecord(s.a, c->stream);
  c->stages.push_back(s);
}

void stage_end(amg_ctx* c) {
  if (!c->timing || c->stages.empty()) return;
  (void)hipEventRecord(c->stages.back().b, c->stream);
}

extern "C" int amg_last_timings(amg_ctx* c, const char** names, float* ms, int cap) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  (void)hipStreamSynchronize(c->stream);
  int n = 0;
  for (auto& s : c->stages) {
    if (n >= cap) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) t = -1.f;
    s.ms = t;
    names[n] = s.name;
    ms[n] = t;
    ++n;
  }
  return n;
}

extern "C" int amg_set_timing(amg_ctx* c, int on) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  c->timing = on != 0;
  if (!c->timing) stages_reset(c);
  return AMG_OK;
}

// ------------------------------------------------------------------ clear_many
struct ClearArgs {
  void* p[8];
  unsigned long long words[8];  // 4-byte words per range
  unsigned int fill[8];         // the word each range is filled with
  int n;
};
__global__ __launch_bounds__(256) void k_clear_many(ClearArgs a) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256ull;
  for (int r = 0; r < a.n; ++r) {
    unsigned int* p = reinterpret_cast<unsigned int*>(a.p[r]);
    const unsigned long long w = a.words[r];
    const unsigned int f = a.fill[r];
    // 16-byte stores over the aligned middle, words at the ragged ends
    const unsigned long long head = ((16u - ((unsigned long long)(uintptr_t)p & 15u)) & 15u) >> 2;
    const unsigned long long h = head < w ? head : w;
    const unsigned long long quads = (w - h) >> 2;
    uint4* q = reinterpret_cast<uint4*>(p + h);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < quads; i += stride)
      q[i] = make_uint4(f, f, f, f);
    if (blockIdx.x == 0) {
      for (unsigned long long i = threadIdx.x; i < h; i += 256) p[i] = f;
      for (unsigned long long i = h + quads * 4 + threadIdx.x; i < w; i += 256) p[i] = f;
    }
  }
}

// The mailbox: FETCH_MAX + 1 slots of {value, ticket} in pinned host memory; the host waits until every slot it asked
// for carries the ticket of this read-back: value, system-scope fence, ticket with release semantics.  (Value and
// ticket in ONE 16-byte store without the fence was tried — the ~10 us of idle stream after each read-back looked like
// the L2 write-back of that fence; measured, they are the host's round trip — and removed.)
__global__ void k_fetch(FetchList l, unsigned long long* mail, unsigned long long ticket) {
  const int i = threadIdx.x;
  const int n = l.n > 0 ? l.n : 1;  // an empty list still delivers its ticket (stream_wait)
  if (i < n) {
    const unsigned long long v = i < l.n ? *l.p[i] : 0ull;
    mail[2 * i] = v;
    __threadfence_system();
    __hip_atomic_store(mail + 2 * i + 1, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int fetch(amg_ctx* c, const FetchList& l, unsigned long long* out, const ClearList* filler) {
  if (l.overflow) return amg_fail(AMG_E_ARG, "fetch: more than %d words in one list", FETCH_MAX);
  const bool plain = getenv("AMG_PLAIN_SYNC") != nullptr;  // A/B switch: hipMemcpyAsync + hipStreamSynchronize
  if (plain) {
    for (int i = 0; i < l.n; ++i)
      HIPCHK(hipMemcpyAsync(out + i, l.p[i], sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (filler) AMGCHK(clear_many(c, *filler));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AMG_OK;
  }
  if (!c->mail_host) {
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->mail_host), 2 * (FETCH_MAX + 1) * sizeof(unsigned long long),
                         hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->mail_dev), c->mail_host, 0));
    for (int i = 0; i < 2 * (FETCH_MAX + 1); ++i) c->mail_host[i] = 0;
  }
  const unsigned long long ticket = ++c->mail_ticket;
  hipLaunchKernelGGL(k_fetch, dim3(1), dim3(FETCH_MAX <= 64 ? 64 : 128), 0, c->stream, l, c->mail_dev, ticket);
  if (filler) AMGCHK(clear_many(c, *filler));
  const int n = l.n > 0 ? l.n : 1;
  volatile unsigned long long* m = c->mail_host;
  int have = 0;  // slots 0 .. have - 1 carry the ticket
  for (unsigned long long spins = 0; have < n; ++spins) {
    while (have < n && __atomic_load_n(&m[2 * have + 1], __ATOMIC_ACQUIRE) == ticket) ++have;
    if (have == n) break;
    __builtin_ia32_pause();
    if ((spins & 0xfffffull) == 0xfffffull) {  // ~ms: a launch that failed never delivers the ticket
      const hipError_t e = hipStreamQuery(c->stream);
      if (e != hipSuccess && e != hipErrorNotReady) return amg_fail(AMG_E_HIP, "%s", hipGetErrorString(e));
      if (e == hipSuccess) {
        while (have < n && __atomic_load_n(&m[2 * have + 1], __ATOMIC_ACQUIRE) == ticket) ++have;
        if (have < n) return amg_fail(AMG_E_HIP, "read-back kernel did not run");
      }
    }
  }
  for (int i = 0; i < l.n; ++i) out[i] = m[2 * i];
  return AMG_OK;
}

int stream_wait(amg_ctx* c) {
  FetchList l;
  return fetch(c, l, nullptr);
}

int fetch_status(amg_ctx* c, unsigned long long* out, const ClearList* filler) {
  FetchList l;
  l.add_words(c->status.p, ST_WORDS);
  return fetch(c, l, out, filler);
}

int clear_many(amg_ctx* c, const ClearList& l) {
  if (l.overflow) return amg_fail(AMG_E_ARG, "clear_many: more than %d ranges in one list", CLEAR_MAX);
  if (l.n == 0) return AMG_OK;
  ClearArgs a;
  unsigned long long most = 0;
  a.n = l.n;
  for (int i = 0; i < l.n; ++i) {
    a.p[i] = l.p[i];
    a.words[i] = l.bytes[i] >> 2;
    a.fill[i] = l.fill[i];
    most = a.words[i] > most ? a.words[i] : most;
  }
  unsigned long long blocks = (most / 4 + 256 * 8 - 1) / (256 * 8);  // ~8 quads per thread at the largest range
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_clear_many, dim3((unsigned int)blocks), dim3(256), 0, c->stream, a);
  return AMG_OK;
}

// ------------------------------------------------------------------ lifetime
extern "C" int amg_create(int device, amg_ctx** out) {
  if (!out) return amg_fail(AMG_E_ARG, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return amg_fail(AMG_E_HIP, "no HIP device available (%s): libamg has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return amg_fail(AMG_E_ARG, "device %d out of range [0,%d)", device, n);
  HIPCHK(hipSetDevice(device));
  amg_ctx* c = new amg_ctx();
  c->device = device;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return amg_fail(AMG_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  int r = c->status.ensure(ST_WORDS * sizeof(unsigned long long));
  if (r != AMG_OK) {
    delete c;
    return r;
  }
  const char* t = getenv("AMG_TIMING");
  c->timing = !(t && t[0] == '0');
  *out = c;
  return AMG_OK;
}

extern "C" int amg_destroy(amg_ctx* c) {
  if (!c) return AMG_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  stages_reset(c);
  dist_release(c);
  bubbles_release(c);
  DevBuf* all[] = {&c->tokens,    &c->read_off,  &c->gene_start, &c->gene_end,  &c->read_len,
                   &c->node_tab,  &c->edge_tab,  &c->tok_slot,   &c->tok_node,  &c->tok_dir, &c->tok_pair,
                   &c->node_tokens, &c->node_cov, &c->node_first, &c->node_comp, &c->node_alive,
                   &c->edge_src,  &c->edge_tgt,  &c->edge_sdir,  &c->edge_tdir, &c->edge_cov,
                   &c->edge_alive, &c->adj_off,  &c->adj_edge,   &c->read_fix, &c->ladj_off, &c->ladj, &c->ladj_rows, &c->ladj_pos, &c->ladj_keys, &c->hub_bits, &c->pair_key, &c->pair_first, &c->pair_cnt, &c->dist_a, &c->dist_cnt, &c->dist_first, &c->dist_slot, &c->dist_gtab, &c->dist_lcnt, &c->match_read, &c->match_pos, &c->match_off, &c->match_wave,  &c->c_tokens_buf,
                   &c->c_read_off, &c->c_orig,   &c->c_changed,  &c->c_gstart,  &c->c_gend,
                   &c->c_read_len, &c->status,   &c->sort_tmp, &c->scan_state,   &c->s0, &c->s1, &c->s2, &c->s3,
                   &c->s4, &c->s5, &c->cnt_state, &c->cnt_list, &c->bnd_bits, &c->nw_rec, &c->gap_rec, &c->gm_mask, &c->gm_tab, &c->gm_res, &c->gm_list, &c->gm_q, &c->gm_pool, &c->gm_gene, &c->gm_fail, &c->gm_ctr, &c->nw_big, &c->x_first, &c->x_slot, &c->x_final, &c->x_efirst, &c->x_eslot, &c->x_ecnt, &c->x_ncnt, &c->x_ftag, &c->f_ctrs, &c->x_efinal, &c->x_first_all, &c->pos_off, &c->c_pos_off, &c->pos1_s, &c->pos1_e,
                   &c->c_src, &c->rd_src, &c->alt_tok_node, &c->alt_tok_dir, &c->alt_ntok, &c->alt_ncov, &c->alt_nfirst,
                   &c->alt_nalive, &c->alt_pkey, &c->alt_pfirst, &c->alt_pcnt};
  for (DevBuf* b : all) b->release();
  if (c->mail_host) (void)hipHostFree(c->mail_host);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return AMG_OK;
}

extern "C" int amg_sync(amg_ctx* c) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" void* amg_stream(amg_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ------------------------------------------------------------------ inputs
// on_device: 0 host pointer (copied), 1 device pointer (copied), 2 device pointer BORROWED — no
// copy, the caller keeps the memory alive and unchanged until the next amg_set_* of that array
// or amg_adopt_corrected
static int copy_in(amg_ctx* c, DevBuf& dst, const void* src, size_t bytes, int on_device) {
  if (on_device == 2) {
    dst.borrow(src, bytes);
    return AMG_OK;
  }
  AMGCHK(dst.ensure(bytes + 64));
  if (bytes)
    HIPCHK(hipMemcpyAsync(dst.p, src, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          c->stream));
  return AMG_OK;
}

extern "C" int amg_set_reads(amg_ctx* c, const int32_t* tokens, const int64_t* read_offsets,
                             int64_t n_reads, int32_t two_v, int on_device) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (n_reads < 0 || !read_offsets) return amg_fail(AMG_E_ARG, "bad read_offsets / n_reads");
  if (two_v <= 0 || (two_v & 1)) return amg_fail(AMG_E_ARG, "two_v must be a positive even number");
  HIPCHK(hipSetDevice(c->device));
  int64_t n_tokens = 0;
  if (on_device) {
    HIPCHK(hipMemcpy(&n_tokens, read_offsets + n_reads, sizeof(int64_t), hipMemcpyDeviceToHost));
  } else {
    if (read_offsets[0] != 0) return amg_fail(AMG_E_ARG, "read_offsets[0] must be 0");
    for (int64_t r = 0; r < n_reads; ++r)
      if (read_offsets[r + 1] < read_offsets[r])
        return amg_fail(AMG_E_ARG, "read_offsets not monotone at read %lld", (long long)r);
    n_tokens = read_offsets[n_reads];
  }
  if (n_tokens < 0) return amg_fail(AMG_E_ARG, "read_offsets[n_reads] is negative");
  if (n_tokens > 0 && !tokens) return amg_fail(AMG_E_ARG, "null tokens");
  AMGCHK(copy_in(c, c->tokens, tokens, (size_t)n_tokens * sizeof(int32_t), on_device));
  AMGCHK(copy_in(c, c->read_off, read_offsets, (size_t)(n_reads + 1) * sizeof(int64_t), on_device));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n_reads = n_reads;
  c->n_tokens = n_tokens;
  c->two_v = two_v;
  c->have_pos = c->have_read_len = false;
  c->built = false;
  c->derive_ready = c->dist_candidate = false;
  c->hint_bound = 0;
  c->have_corrected = false;
  c->match_valid = false;
  c->node_hint = 0;
  c->cnt_hint_reset = true;  // what the counting sweeps learned belongs to the previous read set
  return AMG_OK;
}

extern "C" int amg_set_positions(amg_ctx* c, const int64_t* gene_start, const int64_t* gene_end,
                                 const int64_t* read_len, int on_device) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  if (!gene_start || !gene_end) return amg_fail(AMG_E_ARG, "null positions");
  HIPCHK(hipSetDevice(c->device));
  AMGCHK(copy_in(c, c->gene_start, gene_start, (size_t)c->n_tokens * sizeof(int64_t), on_device));
  AMGCHK(copy_in(c, c->gene_end, gene_end, (size_t)c->n_tokens * sizeof(int64_t), on_device));
  c->have_pos = true;
  c->pos_identity = true;
  c->pos0_own = false;
  c->pos_n0 = c->n_tokens;
  c->pos1_used = c->c_pos1_used = 0;
  c->have_corrected = false;
  c->have_read_len = false;
  if (read_len) {
    AMGCHK(copy_in(c, c->read_len, read_len, (size_t)c->n_reads * sizeof(int64_t), on_device));
    c->have_read_len = true;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" int amg_set_read_lengths(amg_ctx* c, const int64_t* read_len, int on_device) {
  if (!c || !read_len) return amg_fail(AMG_E_ARG, "null argument");
  if (c->two_v <= 0) return amg_fail(AMG_E_STATE, "amg_set_reads first");
  HIPCHK(hipSetDevice(c->device));
  AMGCHK(copy_in(c, c->read_len, read_len, (size_t)c->n_reads * sizeof(int64_t), on_device));
  c->have_read_len = true;
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// ------------------------------------------------------------------ counts + read-back
__global__ void k_count_flags(const unsigned char* __restrict__ f, long long n,
                              unsigned long long* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = (i < n && f[i]) ? 1ull : 0ull;
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

static int count_flags(amg_ctx* c, const DevBuf& buf, long long n, int64_t* out) {
  *out = 0;
  if (n <= 0) return AMG_OK;
  unsigned long long* ctr = c->status.as<unsigned long long>() + ST_MISC;
  HIPCHK(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_count_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     buf.as<unsigned char>(), n, ctr);
  unsigned long long h = 0;
  HIPCHK(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *out = (int64_t)h;
  return AMG_OK;
}

extern "C" int amg_sizes(amg_ctx* c, int64_t* n_reads, int64_t* n_tokens) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (n_reads) *n_reads = c->n_reads;
  if (n_tokens) *n_tokens = c->n_tokens;
  return AMG_OK;
}

extern "C" int amg_graph_sizes(amg_ctx* c, int64_t* n_nodes, int64_t* n_edges, int32_t* k) {
  if (!c) return amg_fail(AMG_E_ARG, "null ctx");
  if (!c->built) return amg_fail(AMG_E_STATE, "amg_build first");
  if (n_nodes) *n_nodes = c->n_nodes;
  if (n_edges) *n_edges = c->n_edges;
  if (k) *k = c->k;
  return AMG_OK;
}

extern "C" int amg_counts(amg_ctx* c, amg_counts_t* o) {
  if (!c || !o) return amg_fail(AMG_E_ARG, "null argument");
  memset(o, 0, sizeof(*o));
  o->n_reads = c->n_reads;
  o->n_tokens = c->n_tokens;
  o->two_v = c->two_v;
  if (!c->built) return AMG_OK;
  HIPCHK(hipSetDevice(c->device));
  o->k = c->k;
  o->n_windows = c->n_windows;
  o->n_short_reads = c->n_short;
  o->n_nodes = c->n_nodes;
  o->n_edges = c->n_edges;
  o->n_pairs = c->n_pairs;
  AMGCHK(ensure_components(c));
  o->n_components = c->n_components;
  o->node_table_slots = c->node_slots;
  o->edge_table_slots = c->edge_slots;
  o->exact_keys = c->exact_keys ? (c->x_fp ? 2 : 1) : 0;  // 2: 16-byte slots keyed by verified 94-bit fingerprints
  o->derived = c->derived ? 1 : 0;
  o->build_retries = c->retries;
  AMGCHK(count_flags(c, c->node_alive, c->n_nodes, &o->n_live_nodes));
  AMGCHK(count_flags(c, c->edge_alive, c->n_edges, &o->n_live_edges));
  AMGCHK(count_flags(c, c->read_fix, c->n_reads, &o->n_reads_to_correct));
  return AMG_OK;
}

#define NEED_BUILT(c)                                                     \
  do {                                                                    \
    if (!(c)) return amg_fail(AMG_E_ARG, "null ctx");                     \
    if (!(c)->built) return amg_fail(AMG_E_STATE, "amg_build first");     \
    HIPCHK(hipSetDevice((c)->device));                                    \
  } while (0)

static int d2h(amg_ctx* c, void* dst, const DevBuf& src, size_t bytes) {
  if (!dst || bytes == 0) return AMG_OK;
  HIPCHK(hipMemcpyAsync(dst, src.p, bytes, hipMemcpyDeviceToHost, c->stream));
  return AMG_OK;
}

__global__ void k_first_split(const long long* __restrict__ first, long long n,
                              long long* __restrict__ tok, signed char* __restrict__ dir) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long f = first[i];
  tok[i] = f >> 1;
  dir[i] = (f & 1) ? -1 : 1;
}

extern "C" int amg_get_nodes(amg_ctx* c, int32_t* canon_tokens, uint32_t* coverage,
                             int64_t* first_token, int8_t* first_dir, int32_t* component,
                             uint8_t* alive) {
  NEED_BUILT(c);
  const size_t D = (size_t)c->n_nodes;
  if (component) AMGCHK(ensure_components(c));
  AMGCHK(d2h(c, canon_tokens, c->node_tokens, D * c->k * sizeof(int32_t)));
  AMGCHK(d2h(c, coverage, c->node_cov, D * sizeof(uint32_t)));
  AMGCHK(d2h(c, component, c->node_comp, D * sizeof(int32_t)));
  AMGCHK(d2h(c, alive, c->node_alive, D));
  if ((first_token || first_dir) && D) {
    AMGCHK(c->s1.ensure(D * sizeof(long long)));
    AMGCHK(c->s2.ensure(D));
    hipLaunchKernelGGL(k_first_split, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, c->stream,
                       c->node_first.as<long long>(), (long long)D, c->s1.as<long long>(),
                       c->s2.as<signed char>());
    AMGCHK(d2h(c, first_token, c->s1, D * sizeof(int64_t)));
    AMGCHK(d2h(c, first_dir, c->s2, D));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" int amg_get_edges(amg_ctx* c, int32_t* src, int32_t* tgt, int8_t* sdir, int8_t* tdir,
                             uint32_t* coverage, uint8_t* alive) {
  NEED_BUILT(c);
  const size_t E = (size_t)c->n_edges;
  AMGCHK(d2h(c, src, c->edge_src, E * sizeof(int32_t)));
  AMGCHK(d2h(c, tgt, c->edge_tgt, E * sizeof(int32_t)));
  AMGCHK(d2h(c, sdir, c->edge_sdir, E));
  AMGCHK(d2h(c, tdir, c->edge_tdir, E));
  AMGCHK(d2h(c, coverage, c->edge_cov, E * sizeof(uint32_t)));
  AMGCHK(d2h(c, alive, c->edge_alive, E));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" int amg_get_read_nodes(amg_ctx* c, int32_t* tok_node, int8_t* tok_dir) {
  NEED_BUILT(c);
  AMGCHK(d2h(c, tok_node, c->tok_node, (size_t)c->n_tokens * sizeof(int32_t)));
  AMGCHK(d2h(c, tok_dir, c->tok_dir, (size_t)c->n_tokens));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// the node ids of the windows of a FEW reads, laid end to end (read-path clustering looks at the reads of a gene's
// nodes: tens of thousands of a million — the whole per-window array is 4 bytes per gene of the read set over PCIe)
__global__ __launch_bounds__(256) void k_gather_read_nodes(const int* __restrict__ tok_node,
                                                            const long long* __restrict__ first, const long long* __restrict__ start,
                                                            long long n_rows, int* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const long long a = first[r], o = start[r], n = start[r + 1] - o;
  for (long long i = threadIdx.x & 63; i < n; i += 64) out[o + i] = tok_node[a + i];
}

extern "C" int amg_get_read_nodes_rows(amg_ctx* c, const int64_t* first_token, const int64_t* out_start, int64_t n_rows,
                                       int32_t* node_ids) {
  NEED_BUILT(c);
  if (n_rows < 0 || (n_rows > 0 && (!first_token || !out_start || !node_ids))) return amg_fail(AMG_E_ARG, "bad arguments");
  if (n_rows == 0) return AMG_OK;
  const long long total = out_start[n_rows];
  for (int64_t r = 0; r < n_rows; ++r) {
    const long long n = out_start[r + 1] - out_start[r];
    if (out_start[0] != 0 || n < 0 || first_token[r] < 0 || first_token[r] + n > c->n_tokens)
      return amg_fail(AMG_E_ARG, "amg_get_read_nodes_rows: row %lld lies outside the read set", (long long)r);
  }
  if (total == 0) return AMG_OK;
  hipStream_t st = c->stream;
  AMGCHK(c->s0.ensure((size_t)n_rows * sizeof(long long)));
  AMGCHK(c->s1.ensure((size_t)(n_rows + 1) * sizeof(long long)));
  AMGCHK(c->s2.ensure((size_t)total * sizeof(int)));
  HIPCHK(hipMemcpyAsync(c->s0.p, first_token, (size_t)n_rows * sizeof(long long), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->s1.p, out_start, (size_t)(n_rows + 1) * sizeof(long long), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_gather_read_nodes, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st, c->tok_node.as<int>(),
                     c->s0.as<long long>(), c->s1.as<long long>(), (long long)n_rows, c->s2.as<int>());
  HIPCHK(hipMemcpyAsync(node_ids, c->s2.p, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return AMG_OK;
}

extern "C" int amg_get_node_adj(amg_ctx* c, int64_t* offsets, int32_t* edge_ids) {
  NEED_BUILT(c);
  AMGCHK(ensure_adjacency(c));
  AMGCHK(d2h(c, offsets, c->adj_off, (size_t)(2 * c->n_nodes + 1) * sizeof(int64_t)));
  AMGCHK(d2h(c, edge_ids, c->adj_edge, (size_t)c->n_edges * sizeof(int32_t)));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

extern "C" int amg_get_reads_to_correct(amg_ctx* c, uint8_t* flags) {
  NEED_BUILT(c);
  AMGCHK(d2h(c, flags, c->read_fix, (size_t)c->n_reads));
  HIPCHK(hipStreamSynchronize(c->stream));
  return AMG_OK;
}

// ------------------------------------------------------------------ Node.listOfReads
// (node, read) for every live window; windows are visited in (read, position) order, so a
// STABLE sort by node id leaves each node's reads in first-appearance order with repeats
// adjacent (a read's windows are contiguous) — construct_node.py:64-67.
__global__ void k_read_window_count(const int* __restrict__ tok_node,
                                    const long long* __restrict__ read_off, long long n_reads,
                                    unsigned int* __restrict__ cnt) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  unsigned int n = 0;
  for (long long t = read_off[r]; t < read_off[r + 1]; ++t) n += tok_node[t] >= 0 ? 1u : 0u;
  cnt[r] = n;
}

__global__ void k_read_window_emit(const int* __restrict__ tok_node,
                                   const long long* __restrict__ read_off, long long n_reads,
                                   const long long* __restrict__ base, unsigned int* __restrict__ keys,
                                   unsigned int* __restrict__ vals) {
  long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  long long o = base[r];
  for (long long t = read_off[r]; t < read_off[r + 1]; ++t) {
    int n = tok_node[t];
    if (n >= 0) {
      keys[o] = (unsigned int)n;
      vals[o] = (unsigned int)r;
      ++o;
    }
  }
}

// after the stable sort: keep the first of each run of equal (node, read)
__global__ void k_mark_first_of_run(const unsigned int* __restrict__ keys,
                                    const unsigned int* __restrict__ vals, long long n,
                                    unsigned int* __restrict__ keep,
                                    unsigned int* __restrict__ per_node) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned int kf = (i == 0 || keys[i] != keys[i - 1] || vals[i] != vals[i - 1]) ? 1u : 0u;
  keep[i] = kf;
  if (kf) atomicAdd(&per_node[keys[i]], 1u);
}

__global__ void k_scatter_kept(const unsigned int* __restrict__ vals, const unsigned int* __restrict__ keep,
                               const long long* __restrict__ pos, long long n, int* __restrict__ out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) out[pos[i]] = (int)vals[i];
}

extern "C" int amg_get_node_reads(amg_ctx* c, int64_t* offsets, int32_t* read_idx) {
  NEED_BUILT(c);
  if (!offsets) return amg_fail(AMG_E_ARG, "offsets must not be NULL");
  hipStream_t st = c->stream;
  const long long R = c->n_reads, D = c->n_nodes;
  // live windows per read -> bases
  AMGCHK(c->s0.ensure((size_t)(R + 2) * sizeof(unsigned int)));
  AMGCHK(c->s5.ensure((size_t)(R + 2) * sizeof(long long)));
  HIPCHK(hipMemsetAsync(c->s0.p, 0, (size_t)(R + 2) * sizeof(unsigned int), st));
  if (R > 0)
    hipLaunchKernelGGL(k_read_window_count, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                       c->tok_node.as<int>(), c->read_off.as<long long>(), R, c->s0.as<unsigned int>());
  AMGCHK(prim_exscan_u32_to_i64(c, c->s0.as<unsigned int>(), c->s5.as<long long>(), (size_t)R + 1));
  long long W = 0;
  HIPCHK(hipMemcpyAsync(&W, c->s5.as<long long>() + R, sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  AMGCHK(c->s1.ensure((size_t)(W + 2) * sizeof(unsigned int)));
  AMGCHK(c->s2.ensure((size_t)(W + 2) * sizeof(unsigned int)));
  AMGCHK(c->s3.ensure((size_t)(W + 2) * sizeof(unsigned int)));
  AMGCHK(c->s4.ensure((size_t)(W + 2) * sizeof(unsigned int)));
  if (R > 0)
    hipLaunchKernelGGL(k_read_window_emit, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                       c->tok_node.as<int>(), c->read_off.as<long long>(), R, c->s5.as<long long>(),
                       c->s1.as<unsigned int>(), c->s2.as<unsigned int>());
  AMGCHK(prim_sort_u32_u32(c, c->s1.as<unsigned int>(), c->s3.as<unsigned int>(),
                           c->s2.as<unsigned int>(), c->s4.as<unsigned int>(), (size_t)W,
                           ilog2_ceil((uint64_t)D + 2) + 1));
  // s3 = sorted node ids, s4 = reads; keep flags -> s1, per-node counts -> s0 (D + 1)
  AMGCHK(c->s0.ensure((size_t)(D + 2) * sizeof(unsigned int)));
  HIPCHK(hipMemsetAsync(c->s0.p, 0, (size_t)(D + 2) * sizeof(unsigned int), st));
  if (W > 0)
    hipLaunchKernelGGL(k_mark_first_of_run, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st,
                       c->s3.as<unsigned int>(), c->s4.as<unsigned int>(), W,
                       c->s1.as<unsigned int>(), c->s0.as<unsigned int>());
  AMGCHK(c->s5.ensure((size_t)(D + 2) * sizeof(long long)));
  AMGCHK(prim_exscan_u32_to_i64(c, c->s0.as<unsigned int>(), c->s5.as<long long>(), (size_t)D + 1));
  AMGCHK(d2h(c, offsets, c->s5, (size_t)(D + 1) * sizeof(int64_t)));
  HIPCHK(hipStreamSynchronize(st));
  if (read_idx && W > 0) {
    long long total = offsets[D];
    // positions of kept entries = exclusive scan of keep flags
    AMGCHK(c->s2.ensure((size_t)(W + 2) * sizeof(long long)));
    HIPCHK(hipMemsetAsync(c->s1.as<unsigned int>() + W, 0, sizeof(unsigned int), st));
    AMGCHK(prim_exscan_u32_to_i64(c, c->s1.as<unsigned int>(), c->s2.as<long long>(), (size_t)W + 1));
    AMGCHK(c->s0.ensure((size_t)(total + 2) * sizeof(int)));
    hipLaunchKernelGGL(k_scatter_kept, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, st,
                       c->s4.as<unsigned int>(), c->s1.as<unsigned int>(), c->s2.as<long long>(), W,
                       c->s0.as<int>());
    AMGCHK(d2h(c, read_idx, c->s0, (size_t)total * sizeof(int32_t)));
    HIPCHK(hipStreamSynchronize(st));
  }
  return AMG_OK;
}
