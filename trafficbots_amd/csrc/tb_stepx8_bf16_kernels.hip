// bf16-operand build of the 8-wave step kernel (namespace tb::xb): see tb_stepx_bf16_kernels.hip
#define TB_XDL_BF16
#include "tb_stepx8_kernels.hip"
