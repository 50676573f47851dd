// Weight-streaming projections, row regime MT = 2 (17..32 rows... see launch_MT): the launch shapes of this regime
// (templates in pc_gemm_skinny.h; replaces the nn.Linear calls of promptcache/model/llama2.py:345-347, :405, :242, :1050).
#include "pc_gemm_skinny.h"

namespace pcg {
PC_SKINNY_MT_DEFINE(launch_skinny_mt2, 2)
}  // namespace pcg
