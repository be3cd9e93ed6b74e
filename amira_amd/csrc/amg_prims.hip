// amg_prims.hip — library primitives (radix sort) behind a narrow interface (prefix sums: amg_scan.hip).
// rocPRIM does the generic sorting of SMALL arrays (distinct nodes, edge
// classes); every gene-mer-sized kernel (window extraction, hashing, table upsert,
// compaction, masking, threading, matching) is hand-written in amg_build.hip /
// amg_passes.hip.  Kept in its own translation unit because it dominates compile time.
#include "amg_internal.h"

#include <rocprim/rocprim.hpp>

// rocPRIM switches to a merge sort (two launches per doubling: ~40 launches of ~6 us) below
// 1 M items; the rebuilds of a cleaning sweep sort 0.5 - 1 M node / edge records three times
// each, where the onesweep radix sort needs 5 - 6 launches.  Keep merge sort for small inputs only.
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                           rocprim::default_config, 32768>;

template <class F>
static int with_temp(amg_ctx* c, F&& call) {
  size_t bytes = 0;
  hipError_t e = call(nullptr, bytes);
  if (e != hipSuccess) return amg_fail(AMG_E_HIP, "rocprim size query: %s", hipGetErrorString(e));
  AMGCHK(c->sort_tmp.ensure(bytes ? bytes : 16));
  e = call(c->sort_tmp.p, bytes);
  if (e != hipSuccess) return amg_fail(AMG_E_HIP, "rocprim run: %s", hipGetErrorString(e));
  return AMG_OK;
}

int prim_sort_u64_u32(amg_ctx* c, const unsigned long long* kin, unsigned long long* kout,
                      const unsigned int* vin, unsigned int* vout, size_t n, int end_bit) {
  if (n == 0) return AMG_OK;
  if (end_bit < 1) end_bit = 1;
  if (end_bit > 64) end_bit = 64;
  return with_temp(c, [&](void* tmp, size_t& bytes) {
    return rocprim::radix_sort_pairs<SortCfg>(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit, c->stream);
  });
}

int prim_sort_u32_u32(amg_ctx* c, const unsigned int* kin, unsigned int* kout,
                      const unsigned int* vin, unsigned int* vout, size_t n, int end_bit) {
  if (n == 0) return AMG_OK;
  if (end_bit < 1) end_bit = 1;
  if (end_bit > 32) end_bit = 32;
  return with_temp(c, [&](void* tmp, size_t& bytes) {
    return rocprim::radix_sort_pairs<SortCfg>(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit, c->stream);
  });
}
