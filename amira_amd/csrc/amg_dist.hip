// amg_dist.hip — multi-GPU read-sharded build (RCCL all-to-all table merge).  Round-1:
// entry points exist so the ABI is complete; the exchange itself is not implemented yet.
#include "amg_internal.h"

extern "C" int amg_dist_unique_id(void* id128) {
  (void)id128;
  return amg_fail(AMG_E_DIST, "amg_dist_*: not implemented in this build");
}
extern "C" int amg_dist_init(amg_ctx* c, const void* id128, int rank, int world) {
  (void)c; (void)id128; (void)rank; (void)world;
  return amg_fail(AMG_E_DIST, "amg_dist_*: not implemented in this build");
}
extern "C" int amg_dist_build(amg_ctx* c, int32_t k, int64_t read_index_base, int64_t token_index_base) {
  (void)c; (void)k; (void)read_index_base; (void)token_index_base;
  return amg_fail(AMG_E_DIST, "amg_dist_*: not implemented in this build");
}
