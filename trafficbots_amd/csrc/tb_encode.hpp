// Weight-offset block of the one-time scene encoders (map encoder, TL / agent-history encoders,
// CVAE personality prior, destination predictor).
#pragma once
#include "tb_device.hpp"

namespace tb {

struct EncMlpW {
    uint32_t w1, b1, w2, b2;  // InputPeEncoder MLP attr -> 32 -> 32, plain row-major
};

struct EncoderW {
    EncMlpW map_enc, tl_enc, agent_enc;
    uint32_t pe_fxy, pe_fyaw;
    XLayerW densetnt[3], map_self, as2pl[3], as2tl[3], inter_prior[3], inter_post[3];
    GruLayerW gru_prior[3], gru_dest[3], gru_post[3];
    uint32_t lat_w1, lat_b1, lat_w2, lat_b2;      // latent_prior_dist.mlp_mean
    uint32_t post_w1, post_b1, post_w2, post_b2;  // latent_post_dist.mlp_mean
    uint32_t post_log_std;                        // latent_post_dist.log_std [16]
    uint32_t dest_w0_map, dest_w0_agent, dest_b0, dest_ln0_g, dest_ln0_b;
    uint32_t dest_w1, dest_b1, dest_ln1_g, dest_ln1_b, dest_w2, dest_b2;
};

}  // namespace tb
