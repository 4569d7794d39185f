// k_todo.hip — entry points not implemented yet in this round; they fail loudly.
#include "runtime.h"
using namespace dbhip;
#define TODO(name) do { set_error(name ": not implemented yet"); return DBHIP_ERR_UNSUPPORTED; } while (0)
extern "C" {
int32_t dbhip_join_create(int64_t, dbhip_join**) { TODO("dbhip_join_create"); }
int32_t dbhip_join_add_build(dbhip_join*, const uint64_t*, const uint8_t*, int64_t, void*) { TODO("dbhip_join_add_build"); }
int32_t dbhip_join_finalize(dbhip_join*, void*) { TODO("dbhip_join_finalize"); }
int32_t dbhip_join_probe_count(dbhip_join*, const uint64_t*, const uint8_t*, int64_t, uint64_t*, void*) { TODO("dbhip_join_probe_count"); }
int32_t dbhip_join_probe(dbhip_join*, const uint64_t*, const uint8_t*, int64_t, uint32_t*, uint32_t*, int64_t, uint64_t*, void*) { TODO("dbhip_join_probe"); }
int32_t dbhip_join_destroy(dbhip_join*) { TODO("dbhip_join_destroy"); }
int32_t dbhip_sort_perm(const dbhip_col*, const uint8_t*, const uint8_t*, int32_t, int64_t, int64_t, uint32_t*, void*) { TODO("dbhip_sort_perm"); }
int32_t dbhip_vec_distance(int32_t, const float*, int64_t, int32_t, const float*, int32_t, float*, void*) { TODO("dbhip_vec_distance"); }
int32_t dbhip_vec_topk(int32_t, const float*, int64_t, int32_t, const float*, int32_t, int32_t, uint32_t*, float*, void*) { TODO("dbhip_vec_topk"); }
int32_t dbhip_score_u8(int32_t, const uint8_t*, const uint8_t*, int64_t, int32_t, float*, void*) { TODO("dbhip_score_u8"); }
}
