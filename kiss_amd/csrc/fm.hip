// fm.hip -- FM-index batched query + build kernels (placeholder until the suffix-sort path is parity-green).
#include "kiss_internal.hpp"

extern "C" {

int kiss_hip_fmi_query_batch_dev(kiss_hip_ctx *, const kiss_hip_fmi_view *, const uint8_t *, uint32_t, uint64_t,
                                 uint32_t *, uint32_t *, uint64_t *, uint64_t *, uint32_t *, uint64_t *, uint64_t,
                                 void *)
{
    return KISS_HIP_E_UNSUPPORTED;
}

int kiss_hip_fmi_build_dev(kiss_hip_ctx *, const uint8_t *, uint64_t, const uint32_t *, uint32_t, uint8_t *, uint32_t *,
                           uint8_t *, uint32_t *, uint64_t *, uint32_t *, uint32_t[4], uint32_t *, void *)
{
    return KISS_HIP_E_UNSUPPORTED;
}
}
