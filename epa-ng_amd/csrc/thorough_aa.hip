// 20-state thorough placement (MFMA path) -- placeholder until the kernel lands.
#include "epa_dev_internal.hpp"

int launch_thorough_aa(epa_ctx* ctx, const epa_pair*, uint64_t, const uint8_t*, const uint32_t*,
                       const uint32_t*, uint32_t, epa_result*, unsigned long long*) {
  return epa_fail(ctx, EPA_ERR_UNSUPPORTED, "thorough placement for 20-state models is not implemented yet");
}
