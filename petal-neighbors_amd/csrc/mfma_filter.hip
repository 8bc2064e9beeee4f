// mfma_filter.hip -- placeholder until the MFMA filter kernel lands.
#include "pn_internal.h"
namespace pn {
bool mfma_supported(int, size_t) { return false; }
const char *mfma_kernel_name() { return "mfma_filter_kernel"; }
hipError_t launch_mfma_filter_f32(const float *, const float *, size_t, size_t, int, size_t, const float *,
                                  const float *, int, size_t, const MfmaPlan &, const CandBuf &, hipStream_t) {
    return hipErrorNotSupported;
}
}  // namespace pn
