// The step kernel of tb_stepx_kernels.hip built a second time with ONE bf16 plane per operand (namespace tb::xb):
// "bf16 MFMA inputs, fp32 accumulate" of BASELINE.json configs 4/5.  Same code, same layouts; half the weight bytes and a third
// of the MFMAs of the fp16-pair build, bf16 operand rounding (no fp32-parity claim).
#define TB_XDL_BF16 1
#include "tb_stepx_kernels.hip"
