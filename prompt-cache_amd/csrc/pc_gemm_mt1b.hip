// Weight-streaming projections, row regime MT = 1: the launch shapes with the SiLU (gate|up) and RoPE (q|k|v) epilogues -- a
// translation unit of its own so that it compiles next to pc_gemm_mt1.hip (together they were the longest compile of the library;
// templates in pc_gemm_skinny.h; replaces the nn.Linear calls of promptcache/model/llama2.py:345-347, :242).
#include "pc_gemm_skinny.h"

namespace pcg {
PC_SKINNY_MT_DEFINE_A(launch_skinny_mt1_planes, 1)
}  // namespace pcg
