#include "../../include/trafficbots_hip.h"
#include "tb_encode.hpp"
struct tb_ctx;
namespace tb {
int run_encode(struct ::tb_ctx* ctx, const tb_encode_io* io, hipStream_t s) { return 77; }
}
