// pc_gemm_q8's q|k|v (RoPE + append epilogue) launch shapes: fused-RMSNorm source and the quantiser-image source of
// llama2.py:345-347 under load_in_8bit (templates: pc_gemm_q8.h).  A translation unit of its own so that it compiles next to
// pc_gemm_q8.hip (the gate|up twin: pc_gemm_q8_norm2.hip).
#include "pc_gemm_q8.h"

namespace pcq {
int launch_q8p_rope(const Q8Params& qp, int T, int units, int K, hipStream_t s) { return launch_q8p<EPI_ROPE, true, 4>(qp, T, units, K, s); }
}  // namespace pcq
