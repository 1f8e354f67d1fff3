// kvarq_amd/csrc/kernels_seeded.hip -- the seed-filter scan kernel (placeholder
// until the fused kernel lands: every sequence goes to the exhaustive path).
#include "kvq_host.h"

struct SeedIndex { int unused; };

SeedIndex *kvq_seed_index_build(kvq_table *t) { (void)t; return nullptr; }
void kvq_seed_index_destroy(SeedIndex *ix) { delete ix; }
int kvq_seeded_launch(kvq_scan *, const KvqParams &, const uint8_t *, int64_t, const uint32_t *, int64_t, int64_t, uint32_t)
{
    kvq_set_error(KVQ_ERR_RUNTIME, "seed-filter kernel not built");
    return KVQ_ERR_RUNTIME;
}
