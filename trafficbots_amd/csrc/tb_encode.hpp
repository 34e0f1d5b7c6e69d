// Weight-offset block of the one-time scene encoders (map encoder, TL / agent-history encoders,
// CVAE personality prior, destination predictor).
#pragma once
#include "tb_device.hpp"
#include "tb_rollout.hpp"

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
    // fp16-pair (XDL) packing of the attention blocks, for the encoders' default kernels (tb_encodex_kernels.hip)
    XLayerX densetnt_x[3], map_self_x, as2pl_x[3], as2tl_x[3], inter_prior_x[3], inter_post_x[3];
    GruLayerX gru_prior_x[3], gru_dest_x[3], gru_post_x[3];
    uint32_t dest_w1_x;
    uint32_t dest_w0_map, dest_w0_agent, dest_b0, dest_ln0_g, dest_ln0_b;
    uint32_t dest_w1, dest_b1, dest_ln1_g, dest_ln1_b, dest_w2, dest_b2;
};

// one cross-attention block over 16-row tiles, XDL kernels (K / VT in fragment-major fp16 pairs, same byte size as fp32)
struct XBlockPX {
    const float* W;
    XLayerW L[3];
    XLayerX LX[3];
    int n_layer;
    const float* src;          // [G][n_rows][128]
    const uint8_t* src_valid;  // [G][n_rows]
    float* dst;                // [G][n_rows][128]
    const float* K;            // [G][n_layer] fragment-major K planes
    const float* VT;           // [G][n_layer] fragment-major V planes
    const float* kbias;        // [G][n_pad] additive key mask
    int n_rows, n_pad;
    int eye;                   // MultiAgentTF: self key masked; groups with exactly one valid row pass through
    // k_polyline_fused only (required there): the max over the valid nodes of each polyline (k_pool_nodes); `dst` is not written
    float* pool_out;           // [G][128]
    uint8_t* pool_valid;       // [G]
};

// GRU over time for a 16-agent tile (k_gru_scan / k_gru_scan_x): mode 0 = latent encoder (max over valid steps + DistEncoder mean),
// mode 1 = destination predictor (last valid output + input residual)
struct ScanP {
    const float* W;
    GruLayerW gru[3];
    GruLayerX grux[3];     // XDL kernel only
    uint32_t head_w1, head_b1, head_w2, head_b2;  // mode 0
    int mode, B, S, A;
    const float* x;        // [B][S][A][128]
    const uint8_t* valid;  // [B][S][A]
    float* out_feat;       // mode 1: [B][A][128]
    float* out_mean;       // mode 0: [B][A][16]
    uint8_t* out_valid;    // [B][A]
};

// destination predictor over (agent, polyline) pairs (DestPredictor.forward mode mlp, goal_manager.py:294-333)
struct DestP {
    const float* W;
    uint32_t ln0_g, ln0_b, w1, b1, ln1_g, ln1_b, w2, b2;
    uint32_t w1x;              // XDL packing of w1 (k_dest_pairs_x)
    int B, A, P;
    const float* U;            // [B][P][128]  = W0[:, :128] map_feature + b0
    const float* V;            // [B][A][128]  = W0[:, 128:] agent
    const uint8_t* map_fvalid; // [B][P]
    const int32_t* map_type;   // [B][P]
    const int32_t* agent_type; // [B][A]
    const uint8_t* dist_valid; // [B][A]
    float* logits;             // [B][A][P]
};

namespace xh {
void launch_dest_pairs_x(const DestP& p, hipStream_t s, int lds_pad = 0);
void launch_gru_scan_x(const ScanP& p, int a_pad, hipStream_t s);
hipError_t configure_encodex_kernels();
void launch_range_flag_take_encode(unsigned int* out, hipStream_t s);
void launch_kv_hoist_nx(const float* W, const XLayerW* L, const XLayerX* X, int n_layer, const float* feat, const uint8_t* fvalid, int G,
                        int n_tok, int n_pad, float* K, float* VT, float* kbias, hipStream_t s);
void launch_xblock_x(const XBlockPX& p, int G, hipStream_t s);
void launch_polyline_block_x(const XBlockPX& p, int G, float* K, float* VT, float* kbias, hipStream_t s);
void launch_polyline_fused_x(const XBlockPX& p, int G, hipStream_t s, int eight_waves);  // 0: four waves, 1: eight, 2: eight with merged phases
}  // namespace xh

}  // namespace tb
