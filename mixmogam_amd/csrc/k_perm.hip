// k_perm.hip -- EMMAX permutation test GEMM (placeholder until the scan path is parity-green).
#include "mmg_internal.h"
namespace mmg {
int run_perm(mmg_ctx* ctx, const mmg_geno*, int32_t, const double*, int32_t, const double*, const double*, int,
             double*) {
  return set_err(ctx, MMG_E_STATE, "perm kernel not built yet");
}
}  // namespace mmg
