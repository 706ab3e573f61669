// gemm_mfma.hip -- MFMA-tiled dequant-GEMM for batched prefill (placeholder dispatch until the
// tiled kernel lands: returns GPTQ_E_VARIANT so capi.hip falls back to the skinny kernel).
#include "gptq_device.h"
#include "gptq_internal.h"

namespace gptq {
int gemm_dispatch(int bits, bool fused2, const GemvParams &p, hipStream_t s) {
    (void)bits; (void)fused2; (void)p; (void)s;
    return GPTQ_E_VARIANT;
}
}  // namespace gptq
