// placeholder until kernel C lands
#include "common.h"
namespace rattle {
int poa_msa_run(rattle_ctx *, const uint8_t *, const uint64_t *, uint32_t, const uint32_t *, uint32_t, rattle_msa_set **) {
    set_error("poa_msa: not built yet");
    return RATTLE_ERR_STATE;
}
}
