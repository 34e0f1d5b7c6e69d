// The bf16 step kernel of tb_stepx_bf16_kernels.hip built a third time, carved for THREE workgroups per CU (namespace tb::xb3):
// launches of more than 512 row tiles -- BASELINE.json configs[3], K = 6 futures x 32 scenes x 4 row tiles = 768 -- then run as ONE
// dispatch round instead of one and a half.  Same arithmetic per agent and the same global layouts as tb::xb; what changes is where
// things live: weight units are loaded by the GEMM that consumes them instead of one stage ahead (<= 168 VGPRs; the co-resident
// workgroups hide the latency the prefetch hid), the GRU hidden tiles stay in the rollout workspace (49 KB of LDS).
// tb_device_xdl.hpp (TB_XDL_W3), tb_stepx_kernels.hip (W3).
#define TB_XDL_BF16 1
#define TB_XDL_W3 1
#include "tb_stepx_kernels.hip"
