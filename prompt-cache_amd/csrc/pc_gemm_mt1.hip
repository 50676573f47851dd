// Weight-streaming projections, row regime MT = 1 (1..16 rows... see launch_MT): the launch shapes of this regime whose epilogue
// updates the residual stream or stores (STORE, ADD, GELU); the SiLU / RoPE ones compile next to it in pc_gemm_mt1b.hip
// (templates in pc_gemm_skinny.h; replaces the nn.Linear calls of promptcache/model/llama2.py:345-347, :405, :242, :1050).
#include "pc_gemm_skinny.h"

namespace pcg {
PC_SKINNY_MT_DEFINE_B(launch_skinny_mt1, launch_skinny_mt1_planes, 1)
}  // namespace pcg
