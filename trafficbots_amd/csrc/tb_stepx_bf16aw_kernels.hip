// The bf16 step kernel with ASSIST WAVES (namespace tb::xba): eight-wave workgroups, the second four waves take every other key
// block of the map-attention walks (tb_device_xdl.hpp "Assist waves").  Chosen by tb_api.hip (step_launch) for bf16 launches of one
// workgroup per CU over >= 512 map polylines -- BASELINE configs[4]'s shape.
#define TB_XDL_BF16
#define TB_XDL_AW
#include "tb_stepx_kernels.hip"
