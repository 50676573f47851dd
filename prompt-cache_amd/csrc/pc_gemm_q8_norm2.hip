// pc_gemm_q8's gate|up (SiLU epilogue) launch shapes: fused-RMSNorm source and the quantiser-image source of llama2.py:242 under
// load_in_8bit (templates: pc_gemm_q8.h).  A translation unit of its own so that it compiles next to pc_gemm_q8.hip.
#include "pc_gemm_q8.h"

namespace pcq {
int launch_q8p_silu(const Q8Params& qp, int T, int units, int K, hipStream_t s) { return launch_q8p<EPI_SILU, true, 4>(qp, T, units, K, s); }
}  // namespace pcq
