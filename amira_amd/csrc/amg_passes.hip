// amg_passes.hip — coverage filter, node removal, tip clipping, component filter and
// per-read correction on the device (reference construct_graph.py:402-540, 679-861,
// 950-958, 1123-1396).
#include "amg_device.h"

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

// ------------------------------------------------------------------ filter_graph (:523-540)
// list_nodes_to_remove (:496-503): coverage < minNodeCoverage
__global__ void k_filter_nodes(const unsigned int* __restrict__ cov, unsigned char* __restrict__ alive,
                               long long n, unsigned int min_cov) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && alive[i] && cov[i] < min_cov) alive[i] = 0;
}

// list_edges_to_remove (:505-521): coverage < minEdgeCoverage or a doomed endpoint
__global__ void k_filter_edges(const int* __restrict__ src, const int* __restrict__ tgt,
                               const unsigned int* __restrict__ cov,
                               const unsigned char* __restrict__ node_alive,
                               unsigned char* __restrict__ alive, long long n, unsigned int min_cov) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || !alive[e]) return;
  if (cov[e] < min_cov || !node_alive[src[e]] || !node_alive[tgt[e]]) alive[e] = 0;
}

// remove_node_from_reads (:442-461): one wave per read; windows of removed nodes become
// None (-2) and the read joins _readsToCorrect
__global__ __launch_bounds__(256) void k_mask_reads(int* __restrict__ tok_node,
                                                    const long long* __restrict__ read_off,
                                                    long long n_reads,
                                                    const unsigned char* __restrict__ node_alive,
                                                    unsigned char* __restrict__ read_fix) {
  long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_reads) return;
  int lane = threadIdx.x & 63;
  long long a = read_off[r], b = read_off[r + 1];
  bool hit = false;
  for (long long t = a + lane; t < b; t += 64) {
    int n = tok_node[t];
    if (n >= 0 && !node_alive[n]) {
      tok_node[t] = -2;
      hit = true;
    }
  }
  if (__any(hit) && lane == 0) read_fix[r] = 1;
}

static int apply_removals(amg_ctx* c, unsigned int min_edge_cov) {
  hipStream_t st = c->stream;
  if (c->n_edges > 0)
    hipLaunchKernelGGL(k_filter_edges, dim3(nblk(c->n_edges, 256)), dim3(256), 0, st,
                       c->edge_src.as<int>(), c->edge_tgt.as<int>(), c->edge_cov.as<unsigned int>(),
                       c->node_alive.as<unsigned char>(), c->edge_alive.as<unsigned char>(),
                       c->n_edges, min_edge_cov);
  if (c->n_reads > 0)
    hipLaunchKernelGGL(k_mask_reads, dim3(nblk(c->n_reads, 4)), dim3(256), 0, st,
                       c->tok_node.as<int>(), c->read_off.as<long long>(), c->n_reads,
                       c->node_alive.as<unsigned char>(), c->read_fix.as<unsigned char>());
  return AMG_OK;
}

extern "C" int amg_filter(amg_ctx* c, uint32_t min_node_cov, uint32_t min_edge_cov) {
  NEED_BUILT(c);
  stages_reset(c);
  stage_begin(c, "filter");
  if (c->n_nodes > 0)
    hipLaunchKernelGGL(k_filter_nodes, dim3(nblk(c->n_nodes, 256)), dim3(256), 0, c->stream,
                       c->node_cov.as<unsigned int>(), c->node_alive.as<unsigned char>(),
                       c->n_nodes, min_node_cov);
  AMGCHK(apply_removals(c, min_edge_cov));
  stage_end(c);
  c->have_corrected = false;
  return AMG_OK;
}

// ------------------------------------------------------------------ remove_node (:463-484)
__global__ void k_kill_listed(const int* __restrict__ ids, long long n, long long n_nodes,
                              unsigned char* __restrict__ alive) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int v = ids[i];
  if (v >= 0 && v < n_nodes) alive[v] = 0;
}

extern "C" int amg_remove_nodes(amg_ctx* c, const int32_t* node_ids, int64_t n) {
  NEED_BUILT(c);
  if (n < 0 || (n > 0 && !node_ids)) return amg_fail(AMG_E_ARG, "bad node list");
  if (n == 0) return AMG_OK;
  AMGCHK(c->s0.ensure((size_t)n * sizeof(int)));
  HIPCHK(hipMemcpyAsync(c->s0.p, node_ids, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_kill_listed, dim3(nblk(n, 256)), dim3(256), 0, c->stream, c->s0.as<int>(),
                     (long long)n, c->n_nodes, c->node_alive.as<unsigned char>());
  AMGCHK(apply_removals(c, 0));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_corrected = false;
  return AMG_OK;
}

// ------------------------------------------------------------------ not yet implemented
extern "C" int amg_remove_short_linear_paths(amg_ctx* c, int32_t min_length, const uint8_t* protect,
                                             int64_t* n_removed, int32_t* removed_ids) {
  (void)c; (void)min_length; (void)protect; (void)n_removed; (void)removed_ids;
  return amg_fail(AMG_E_STATE, "amg_remove_short_linear_paths: not implemented yet");
}
extern "C" int amg_remove_low_coverage_components(amg_ctx* c, uint32_t m) {
  (void)c; (void)m;
  return amg_fail(AMG_E_STATE, "amg_remove_low_coverage_components: not implemented yet");
}
extern "C" int amg_correct_reads(amg_ctx* c, int64_t* a, int64_t* b) {
  (void)c; (void)a; (void)b;
  return amg_fail(AMG_E_STATE, "amg_correct_reads: not implemented yet");
}
extern "C" int amg_get_corrected(amg_ctx* c, int32_t* tokens, int64_t* read_offsets, int32_t* orig_read,
                                 uint8_t* changed, int64_t* gene_start, int64_t* gene_end) {
  (void)c; (void)tokens; (void)read_offsets; (void)orig_read; (void)changed; (void)gene_start; (void)gene_end;
  return amg_fail(AMG_E_STATE, "amg_get_corrected: not implemented yet");
}
extern "C" int amg_adopt_corrected(amg_ctx* c) {
  (void)c;
  return amg_fail(AMG_E_STATE, "amg_adopt_corrected: not implemented yet");
}
extern "C" int amg_match_patterns(amg_ctx* c, int which, const int32_t* pat, const int64_t* pat_offsets,
                                  int64_t n_pat, int64_t* hit_offsets, int32_t* hit_read, int32_t* hit_pos) {
  (void)c; (void)which; (void)pat; (void)pat_offsets; (void)n_pat; (void)hit_offsets; (void)hit_read; (void)hit_pos;
  return amg_fail(AMG_E_STATE, "amg_match_patterns: not implemented yet");
}
