// C ABI (include/trafficbots_hip.h): context, weight staging + MFMA-fragment packing, workspace, rollout driver.
#include <hip/hip_runtime.h>

#include <chrono>

#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/trafficbots_hip.h"
#include "tb_internal.hpp"
#include "tb_warm_schedule.inc"  // WS_*: generated from profiles/stage_constants.json (tools/gen_warm_schedule.py)

namespace tb {
size_t step_lds_bytes();
void launch_kv_hoist(const float* W, const XLayerW* L3, const float* feat, const uint8_t* fvalid, int G, int n_tok, int n_pad,
                     float* K, float* VT, float* kbias, hipStream_t s);
void launch_rollout_init(const RolloutP& p, hipStream_t s);
void launch_step(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s);
void launch_rollout_final(const RolloutP& p, float* f_state, uint8_t* f_valid, float* f_hidden, hipStream_t s);
hipError_t configure_rollout_kernels();
#define TB_DECLARE_XDL(NS)                                                                                                       \
    namespace NS {                                                                                                                \
    void launch_step_x(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s);                                              \
    hipError_t configure_stepx_kernel();                                                                                          \
    void launch_fuse_hoist_x(const RolloutP& p, hipStream_t s);                                                                   \
    void launch_step_pre_x(const RolloutP& p, int t0, int n, hipStream_t s);                                                      \
    void launch_inter_pre_x(const RolloutP& p, int t0, int n, hipStream_t s);                                                     \
    void launch_kv_hoist_x(const float* W, const XLayerW* L3, const XLayerX* X3, const float* feat, const uint8_t* fvalid, int G, \
                           int n_tok, int n_pad, float* K, float* VT, float* kbias, int* nkey, hipStream_t s);                    \
    }
TB_DECLARE_XDL(xh)  // fp16 pairs (tb_stepx_kernels.hip)
namespace xh {
void launch_range_flag_take_step(unsigned int* out, hipStream_t s);
}
TB_DECLARE_XDL(xb)  // bf16       (tb_stepx_bf16_kernels.hip)
namespace xb3 {     // bf16, three workgroups per CU (tb_stepx_bf16w3_kernels.hip): launches of more than 512 row tiles
void launch_step_x(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s);
hipError_t configure_stepx_kernel();
void launch_step_pre_x(const RolloutP& p, int t0, int n, hipStream_t s);
void launch_inter_pre_x(const RolloutP& p, int t0, int n, hipStream_t s);
}
namespace xba {     // bf16, eight-wave workgroups with assist waves (tb_stepx_bf16aw_kernels.hip): one workgroup per CU over >= 512 polylines
void launch_step_x(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s);
hipError_t configure_stepx_kernel();
}
int run_encode(struct ::tb_ctx* ctx, const tb_encode_io* io, hipStream_t s);
int run_rule_checks(const tb_rule_io* io, int n_hist, int step_start, uint8_t* raw_ws, hipStream_t s);
int run_encode_posterior(struct ::tb_ctx* ctx, const tb_posterior_io* io, hipStream_t s);
void launch_train_partials(const tb_train_io& io, const float* post_log_std, const float* prior_log_std, hipStream_t s);
hipError_t configure_rule_kernels();
size_t forward_scratch_floats(const tb_forward_io* io);  // tb_forward_kernels.hip
const char* run_forward(const std::map<std::string, const float*>& raw, const tb_forward_io* io, float* scratch, hipStream_t s);
void launch_post_process(const tb_post_io& io, hipStream_t s);
void launch_metric_partials(const tb_metric_io& io, hipStream_t s);
void launch_latent_sample(const tb_latent_sample_io& io, const float* log_std, hipStream_t s);
void launch_dest_sample(const tb_dest_sample_io& io, hipStream_t s);
}  // namespace tb

// ---------------------------------------------------------------------------------------------------
// weight arena builder
// ---------------------------------------------------------------------------------------------------
namespace {

struct Arena {
    std::vector<float> h;
    uint32_t add(const float* src, size_t n) {
        while (h.size() % 64) h.push_back(0.f);
        uint32_t off = (uint32_t)h.size();
        h.insert(h.end(), src, src + n);
        return off;
    }
    uint32_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
};

// W [n_out][k] row-major -> [n_out/16][kp/16][64][4] with lane = kq*16 + m, k = kq*(kp/4) + 4*j + i
std::vector<float> pack_mfma(const float* w, int n_out, int k, int kp) {
    const int n_tiles = n_out / 16, kj = kp / 16;
    std::vector<float> out((size_t)n_tiles * kj * 64 * 4, 0.f);
    for (int t = 0; t < n_tiles; ++t)
        for (int j = 0; j < kj; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int kq = lane >> 4, m = lane & 15;
                    const int kk = kq * (kp / 4) + 4 * j + i;
                    out[(((size_t)t * kj + j) * 64 + lane) * 4 + i] = kk < k ? w[(size_t)(t * 16 + m) * k + kk] : 0.f;
                }
    return out;
}

// fp16-pair packing for the XDL path (tb_device_xdl.hpp): W [n_out][k] row-major, k a multiple of 32 ->
// [n_out/16][k/32][2 planes][64 lanes][8 fp16], lane = kq*16 + row holds W_plane[tile*16 + row][chunk*32 + kq*8 + 0..7];
// w ~ w0 + 2^-11 w1 with w0 = fp16(w), w1 = fp16((w - w0) * 2^11), round-to-nearest-even.  Returned as raw floats (two
// fp16 per float) so that it lives in the same arena.
uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);  // >= 65536 (or inf / nan): inf
    if (x < 0x38800000u) {                                    // < 2^-14: fp16 subnormal (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;           // < 2^-25: rounds to zero
        const int e = (int)(x >> 23);                         // biased fp32 exponent, 102..112
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;    // 24-bit significand
        const int shift = 126 - e;                            // result = mant >> shift, in units of 2^-24; shift in 14..24
        uint32_t r = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13);                   // rebias 127 -> 15, drop 13 mantissa bits
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;   // may carry into the exponent (and to inf): correct
    return (uint16_t)(sign | r);
}
float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float out;
    if (e == 0) {
        out = ldexpf((float)m, -24);
    } else if (e == 31) {
        out = m ? NAN : INFINITY;
    } else {
        const uint32_t x = ((e + 112u) << 23) | (m << 13);
        memcpy(&out, &x, 4);
    }
    return sign ? -out : out;
}
std::vector<float> pack_xdl(const float* w, int n_out, int k) {
    const int n_tiles = n_out / 16, nch = k / 32;
    std::vector<uint16_t> out((size_t)n_tiles * nch * 2 * 512);
    for (int t = 0; t < n_tiles; ++t)
        for (int c = 0; c < nch; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const float v = w[(size_t)(t * 16 + (lane & 15)) * k + c * 32 + (lane >> 4) * 8 + e];
                    const uint16_t h0 = f32_to_f16_rne(v);
                    const float r1 = (v - f16_to_f32(h0)) * 2048.0f;
                    const uint16_t h1 = f32_to_f16_rne(r1);
                    const size_t base = ((size_t)(t * nch + c) * 2) * 512 + (size_t)lane * 8 + e;
                    out[base] = h0;
                    out[base + 512] = h1;
                }
    std::vector<float> f(out.size() / 2);
    memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

struct Stage {
    tb_ctx* ctx;
    bool ok = true;
    // fp16-pair range bookkeeping (xdl_range_ok / ln_range_ok): which kernel family the first out-of-range tensor belongs to
    // (bit 0: step kernels, bit 1: scene encoders) and what it was; never an error by itself -- tb_finalize_weights decides
    unsigned range_hit = 0;
    unsigned range_scope = 3;  // family of the tensors being packed right now
    std::string range_what;
    const std::vector<float>* get(const std::string& name, size_t numel) {
        auto it = ctx->staged.find(name);
        if (it == ctx->staged.end()) {
            if (ok) tb_fail(ctx, "weight '%s' was not loaded", name.c_str());
            ok = false;
            return nullptr;
        }
        if (it->second.size() != numel) {
            if (ok) tb_fail(ctx, "weight '%s': expected %zu elements, got %zu", name.c_str(), numel, it->second.size());
            ok = false;
            return nullptr;
        }
        return &it->second;
    }
};

uint32_t add_plain(Arena& a, Stage& s, const std::string& name, size_t numel) {
    auto v = s.get(name, numel);
    return v ? a.add(*v) : 0;
}

uint32_t add_packed(Arena& a, Stage& s, const std::string& name, int n_out, int k, int kp, int row0 = 0, int rows_total = -1) {
    if (rows_total < 0) rows_total = n_out;
    auto v = s.get(name, (size_t)rows_total * k);
    if (!v) return 0;
    return a.add(pack_mfma(v->data() + (size_t)row0 * k, n_out, k, kp));
}

// single-plane bf16 twin of pack_xdl (operand_precision = 1): [n_out/16][k/32][64 lanes][8 bf16]
std::vector<float> pack_xdl_bf16(const float* w, int n_out, int k) {
    const int n_tiles = n_out / 16, nch = k / 32;
    std::vector<uint16_t> out((size_t)n_tiles * nch * 512);
    for (int t = 0; t < n_tiles; ++t)
        for (int c = 0; c < nch; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const float v = w[(size_t)(t * 16 + (lane & 15)) * k + c * 32 + (lane >> 4) * 8 + e];
                    uint32_t u;
                    memcpy(&u, &v, 4);
                    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
                    out[((size_t)(t * nch + c)) * 512 + (size_t)lane * 8 + e] = (uint16_t)(u >> 16);
                }
    std::vector<float> f(out.size() / 2);
    memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

static bool g_pack_bf16 = false;  // which packing add_xdl emits (set around the two passes of tb_finalize_weights)

// fp16-pair weights share the operand range of tb_device_xdl.hpp: |w| < 65504 (the high plane would be inf).  Checked where the
// planes are made, so that a checkpoint outside the range fails at tb_finalize_weights with the tensor's name instead of producing
// inf / NaN products; the bf16 packing has fp32's range.
bool xdl_range_ok(Stage& s, const std::string& name, const float* w, size_t n) {
    if (g_pack_bf16) return true;
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
    if (m < 65504.0f) return true;
    if (!s.range_hit) {
        char buf[512];
        snprintf(buf, sizeof(buf), "weight '%s': max |w| = %g is outside the fp16-pair range of the fp32-accurate XDL kernels (|w| < 65504)", name.c_str(), (double)m);
        s.range_what = buf;
    }
    s.range_hit |= s.range_scope;
    return false;
}

uint32_t add_xdl(Arena& a, Stage& s, const std::string& name, int n_out, int k, int row0 = 0, int rows_total = -1) {
    if (rows_total < 0) rows_total = n_out;
    auto v = s.get(name, (size_t)rows_total * k);
    if (!v) return 0;
    if (!xdl_range_ok(s, name, v->data() + (size_t)row0 * k, (size_t)n_out * k)) return 0;
    return a.add(g_pack_bf16 ? pack_xdl_bf16(v->data() + (size_t)row0 * k, n_out, k) : pack_xdl(v->data() + (size_t)row0 * k, n_out, k));
}

tb::XLayerX add_xlayer_x(Arena& a, Stage& s, const std::string& p) {
    tb::XLayerX L;
    L.wq = add_xdl(a, s, p + ".attn.in_proj_weight", 128, 128, 0, 384);
    L.wkv = add_xdl(a, s, p + ".attn.in_proj_weight", 256, 128, 128, 384);
    L.wo = add_xdl(a, s, p + ".attn.out_proj_weight", 128, 128);
    L.w1 = add_xdl(a, s, p + ".linear1.weight", 128, 128);
    L.w2 = add_xdl(a, s, p + ".linear2.weight", 128, 128);
    return L;
}

// LayerNorm outputs feed the fp16-pair GEMMs unchecked (tb_device_xdl.hpp): |gamma x^ + beta| <= sqrt(127) max|gamma| + max|beta| must
// stay inside the operand range
void ln_range_ok(Stage& s, const std::string& p) {
    auto g = s.get(p + ".weight", 128);
    auto b = s.get(p + ".bias", 128);
    if (!g || !b) return;
    float mg = 0.f, mb = 0.f;
    for (int i = 0; i < 128; ++i) {
        mg = std::fmax(mg, std::fabs((*g)[i]));
        mb = std::fmax(mb, std::fabs((*b)[i]));
    }
    if (11.27f * mg + mb < 65504.0f) return;
    if (!s.range_hit) {
        char buf[512];
        snprintf(buf, sizeof(buf), "LayerNorm '%s': sqrt(127) max|weight| + max|bias| = %g is outside the fp16-pair operand range (< 65504)", p.c_str(),
                 (double)(11.27f * mg + mb));
        s.range_what = buf;
    }
    s.range_hit |= s.range_scope;
}

tb::XLayerW add_xlayer(Arena& a, Stage& s, const std::string& p) {
    tb::XLayerW L;
    ln_range_ok(s, p + ".norm1");
    ln_range_ok(s, p + ".norm_tgt");
    ln_range_ok(s, p + ".norm2");
    L.ln1_g = add_plain(a, s, p + ".norm1.weight", 128);
    L.ln1_b = add_plain(a, s, p + ".norm1.bias", 128);
    L.lnt_g = add_plain(a, s, p + ".norm_tgt.weight", 128);
    L.lnt_b = add_plain(a, s, p + ".norm_tgt.bias", 128);
    L.ln2_g = add_plain(a, s, p + ".norm2.weight", 128);
    L.ln2_b = add_plain(a, s, p + ".norm2.bias", 128);
    L.wq = add_packed(a, s, p + ".attn.in_proj_weight", 128, 128, 128, 0, 384);
    L.wkv = add_packed(a, s, p + ".attn.in_proj_weight", 256, 128, 128, 128, 384);
    auto bin = s.get(p + ".attn.in_proj_bias", 384);
    if (bin) {
        L.bq = a.add(bin->data(), 128);
        L.bkv = a.add(bin->data() + 128, 256);
    }
    L.wo = add_packed(a, s, p + ".attn.out_proj_weight", 128, 128, 128);
    L.bo = add_plain(a, s, p + ".attn.out_proj_bias", 128);
    L.w1 = add_packed(a, s, p + ".linear1.weight", 128, 128, 128);
    L.b1 = add_plain(a, s, p + ".linear1.bias", 128);
    L.w2 = add_packed(a, s, p + ".linear2.weight", 128, 128, 128);
    L.b2 = add_plain(a, s, p + ".linear2.bias", 128);
    return L;
}

tb::GruLayerW add_gru(Arena& a, Stage& s, const std::string& p, int l) {
    tb::GruLayerW G;
    const std::string sl = std::to_string(l);
    G.wih = add_packed(a, s, p + ".weight_ih_l" + sl, 384, 128, 128);
    G.whh = add_packed(a, s, p + ".weight_hh_l" + sl, 384, 128, 128);
    G.bih = add_plain(a, s, p + ".bias_ih_l" + sl, 384);
    G.bhh = add_plain(a, s, p + ".bias_hh_l" + sl, 384);
    return G;
}

uint32_t add_freqs(Arena& a, Stage& s, const std::string& name, int n_full) {
    auto v = s.get(name, n_full);
    if (!v) return 0;
    std::vector<float> f(n_full / 2);
    for (int i = 0; i < n_full / 2; ++i) f[i] = (*v)[2 * i];  // repeat_interleave(2) table (pos_emb.py:13,43)
    return a.add(f);
}

tb::EncMlpW add_enc(Arena& a, Stage& s, const std::string& p, int attr_dim) {
    tb::EncMlpW e;
    e.w1 = add_plain(a, s, p + ".mlp.fc_layers.0.weight", (size_t)32 * attr_dim);
    e.b1 = add_plain(a, s, p + ".mlp.fc_layers.0.bias", 32);
    e.w2 = add_plain(a, s, p + ".mlp.fc_layers.3.weight", 32 * 32);
    e.b2 = add_plain(a, s, p + ".mlp.fc_layers.3.bias", 32);
    return e;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C" {

void tb_struct_sizes(int32_t out[13]) {
    out[11] = (int32_t)sizeof(tb_latent_sample_io);
    out[12] = (int32_t)sizeof(tb_dest_sample_io);
    out[9] = (int32_t)sizeof(tb_step_override);
    out[10] = (int32_t)sizeof(tb_forward_io);
    out[7] = (int32_t)sizeof(tb_posterior_io);
    out[8] = (int32_t)sizeof(tb_train_io);
    out[0] = (int32_t)sizeof(tb_config);
    out[1] = (int32_t)sizeof(tb_rollout_io);
    out[2] = (int32_t)sizeof(tb_encode_io);
    out[3] = (int32_t)sizeof(tb_rule_io);
    out[4] = (int32_t)sizeof(tb_post_io);
    out[5] = (int32_t)sizeof(tb_metric_io);
    out[6] = (int32_t)sizeof(void*);
}

// Host-side helper of the staging layer (trafficbots_amd/staging.py): bool one-hot rows [n_rows][n_class] -> int32 class index, the
// FIRST set class (what argmax gives on a 0/1 row), -1 where none is set.  Plain host code: numpy has no fast reduction over a short
// last axis (0.6 ms for the three one-hot tensors of a 32-scene batch; this loop: < 0.1 ms) and ctypes releases the GIL around it.
void tb_host_onehot_index(const uint8_t* onehot, int64_t n_rows, int32_t n_class, int32_t* out) {
    for (int64_t r = 0; r < n_rows; ++r) {
        const uint8_t* x = onehot + r * n_class;
        int32_t idx = -1;
        for (int32_t c = n_class - 1; c >= 0; --c) idx = x[c] ? c : idx;
        out[r] = idx;
    }
}

const char* tb_version(void) { return "trafficbots_hip 0.2 (gfx950, fp16-pair XDL MFMA 16x16x32 with fp32 accumulate; fp32 MFMA 16x16x4 kernels selectable)"; }

// tb_config.sw with the debugging environment variables laid over it (read per API call: the tests flip them inside one process).
// The variable of a switch, when set, wins over the configuration field.
extern "C++" TbSw tb_switches_now(const tb_ctx* ctx) {
    const tb_switches& c = ctx->cfg.sw;
    auto env = [](const char* name) -> const char* { return getenv(name); };
    auto off = [&](int32_t field, const char* name) {  // 1 = off; variable "0" = off, anything else = on
        if (const char* e = env(name)) return e[0] == '0' ? 1 : 0;
        return field == 1 ? 1 : 0;
    };
    TbSw s{};
    {
        const char* e = env("TB_STEP_HELPERS");
        if (!e) e = env("TB_GRU_HELPER");  // (the switch's first name)
        // 0: automatic (on, unless another context of the device is launching: two rollouts in flight fill the chip by themselves),
        // 1: off, 2: on regardless
        s.helpers_off = e ? (e[0] == '0' ? 1 : 2) : (c.step_helpers == 1 ? 1 : (c.step_helpers == 2 ? 2 : 0));
    }
    if (const char* e = env("TB_STEP_WARM")) s.warm = e[0] == '0' ? 1 : 2;
    else s.warm = c.step_l2_warmers;
    s.pre_inter_off = off(c.step_pre_inter, "TB_STEP_PRE_INTER");
    s.w3_off = off(c.step_w3, "TB_STEP_W3");
    if (const char* e = env("TB_STEP_AW")) s.aw = e[0] == '0' ? 1 : (e[0] == '2' ? 2 : 0);
    else s.aw = c.step_aw;
    s.lean_off = off(c.step_lean, "TB_STEP_LEAN");
    s.graph_off = off(c.rollout_graph, "TB_ROLLOUT_GRAPH");
    if (const char* e = env("TB_ENCODE_PACK")) s.enc_pack = (e[0] >= '0' && e[0] <= '4') ? e[0] - '0' : 4;
    else s.enc_pack = (c.encode_pack >= 1 && c.encode_pack <= 5) ? c.encode_pack - 1 : 4;
    s.enc_side_off = off(c.encode_side, "TB_ENCODE_SIDE");
    s.enc_dest_side_off = off(c.encode_dest_side, "TB_ENCODE_DEST_SIDE");
    if (const char* e = env("TB_DEST_LDS_PAD")) s.dest_lds_pad = atoi(e);
    else s.dest_lds_pad = c.dest_lds_pad == 0 ? 30000 : (c.dest_lds_pad < 0 ? 0 : c.dest_lds_pad);
    return s;
}

// bf16 launches of more than 512 row tiles (two per CU would need more than one dispatch round) take the three-per-CU carve
// (tb::xb3, same results); tb_switches.step_w3 = 1 keeps them on the two-per-CU carve (development / A-B switch)
static bool w3_launch(const TbSw& sw, size_t n_tiles) { return !sw.w3_off && n_tiles > 512; }

// bf16 launches of one workgroup per CU (129 .. 256 row tiles: too many for helper workgroups, too few for a second dispatch round)
// whose A half walks >= 512 map polylines run eight-wave workgroups: four assist waves take every other key block of the
// map-attention walks (tb::xba; tb_device_xdl.hpp "Assist waves").  The two halves of a walk are merged in a fixed order, so the
// result is deterministic, but it is not the bit pattern of the four-wave kernel (another summation order of the same softmax).
// TB_STEP_AW=0 keeps such launches on the four-wave kernel (development / A-B switch).
// TB_STEP_AW=2 (test switch): such launches of ANY size up to 256 tiles take the assist carve -- the rollout then runs without helper
// workgroups (aw_forced, rollout_setup) -- so that a one-scene case the CPU oracle can follow exercises it.
static bool aw_forced(const TbSw& sw) { return sw.aw == 2; }
static bool aw_launch(const TbSw& sw, const tb::RolloutP& p, int do_a) {
    const size_t n_tiles = (size_t)(p.a_pad / tb::TM) * p.n_inst;
    return sw.aw != 1 && do_a && (n_tiles > 128 || aw_forced(sw)) && n_tiles <= 256 && p.p_pad >= 512;
}

// `rd` supplies what C(t) reads (normally the same struct as `wr`; the batched warm start substitutes its slices), `wr` what
// the launch writes
static void step_launch(const tb_ctx* ctx, const tb::RolloutP& rd, const tb::RolloutP& wr, int t, int do_c, int do_a, hipStream_t s) {
    tb::RolloutP p = rd;
    tb::set_parity(p, t);  // launch t reads the cross-tile buffers of parity t & 1 and writes the other pair (tb_rollout.hpp)
    {
        tb::RolloutP w = wr;
        tb::set_parity(w, t);
        p.valid_w = w.valid_w; p.vbias_w = w.vbias_w; p.kin_w = w.kin_w; p.vtin_w = w.vtin_w; p.x_mid_w = w.x_mid_w;
    }
    const TbSw sw = tb_switches_now(ctx);
    if (ctx->step_kernel == 3 && w3_launch(sw, (size_t)(p.a_pad / tb::TM) * p.n_inst))
        tb::xb3::launch_step_x(p, t, do_c, do_a, s);
    else if (ctx->step_kernel == 3 && aw_launch(sw, p, do_a))
        tb::xba::launch_step_x(p, t, do_c, do_a, s);
    else if (ctx->step_kernel == 3)
        tb::xb::launch_step_x(p, t, do_c, do_a, s);
    else if (ctx->step_kernel == 2)
        tb::xh::launch_step_x(p, t, do_c, do_a, s);
    else
        tb::launch_step(p, t, do_c, do_a, s);
}

// Live contexts of the process (tb_check_status: the fp16-pair range flag is one word per device and kernel family, so the context
// whose check takes it hands the bits to every other context of that device -- ADVICE r04: with two contexts on one device A's check
// used to clear an overflow raised by B, and B got rc 0 for invalid results)
static std::mutex g_ctx_mutex;
static std::vector<tb_ctx*> g_ctx_live;

int tb_create(const tb_config* cfg, tb_ctx** out) {
    if (!cfg || !out) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "trafficbots_hip: no HIP device visible -- this library has no CPU fallback\n");
        return 2;
    }
    tb_ctx* c = new tb_ctx();
    c->cfg = *cfg;
    if (cfg->operand_precision < 0 || cfg->operand_precision > 2) {
        fprintf(stderr, "trafficbots_hip: tb_config.operand_precision must be 0 (fp32-accurate, fp16-pair XDL kernels with automatic fallback), "
                        "1 (bf16 operands) or 2 (exact fp32: fp32 MFMA kernels)\n");
        delete c;
        return 1;
    }
    if (cfg->operand_precision == 1) c->step_kernel = 3;
    if (cfg->operand_precision == 2) {
        c->step_kernel = 0;
        c->encode_kernel = 0;
    }
    if (const char* w = cfg->operand_precision == 2 ? nullptr : getenv("TB_ENCODE_KERNEL")) {  // development switch between the scene-encoder attention kernels
        const std::string k = w;
        if (k == "fp32")
            c->encode_kernel = 0;
        else if (k == "xdl")
            c->encode_kernel = 1;
        else {
            fprintf(stderr, "trafficbots_hip: TB_ENCODE_KERNEL must be fp32 or xdl\n");
            delete c;
            return 1;
        }
    }
    if (const char* w = cfg->operand_precision != 0 ? nullptr : getenv("TB_STEP_KERNEL")) {  // development switch between the fp32-accurate step kernels
        const std::string k = w;
        if (k == "fp32")
            c->step_kernel = 0;
        else if (k == "xdl")
            c->step_kernel = 2;
        else {
            fprintf(stderr, "trafficbots_hip: TB_STEP_KERNEL must be fp32 or xdl\n");
            delete c;
            return 1;
        }
    }
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return 2;
    }
    c->step_kernel0 = c->step_kernel;
    c->encode_kernel0 = c->encode_kernel;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        g_ctx_live.push_back(c);
    }
    *out = c;
    return 0;
}

void tb_destroy(tb_ctx* ctx) {
    if (ctx && ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
    if (ctx && ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        for (size_t i = 0; i < g_ctx_live.size(); ++i)
            if (g_ctx_live[i] == ctx) {
                g_ctx_live.erase(g_ctx_live.begin() + i);
                break;
            }
    }
    if (ctx->enc_fork) (void)hipEventDestroy(ctx->enc_fork);
    if (ctx->enc_join) (void)hipEventDestroy(ctx->enc_join);
    if (ctx->enc_map) (void)hipEventDestroy(ctx->enc_map);
    if (ctx->enc_join2) (void)hipEventDestroy(ctx->enc_join2);
    if (ctx->d_arena) (void)hipFree(ctx->d_arena);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_ws_enc) (void)hipFree(ctx->d_ws_enc);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->d_rule_ws) (void)hipFree(ctx->d_rule_ws);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    for (auto& kv : ctx->warm_tabs)
        if (kv.second.d) (void)hipFree(kv.second.d);
    if (ctx->d_raw) (void)hipFree(ctx->d_raw);
    if (ctx->d_fw) (void)hipFree(ctx->d_fw);
    for (auto e : ctx->ev) (void)hipEventDestroy(e);
    delete ctx;
}

const char* tb_last_error(tb_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int tb_load_weight(tb_ctx* ctx, const char* name, const float* host_data, int64_t numel) {
    if (!ctx || !name || !host_data || numel <= 0) return tb_fail(ctx, "tb_load_weight: bad argument");
    ctx->staged[name] = std::vector<float>(host_data, host_data + numel);
    ctx->finalized = false;
    return 0;
}

int tb_finalize_weights(tb_ctx* ctx, tb_stream stream) {
    if (!ctx) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    TB_HIP(ctx, tb::configure_rollout_kernels());
    TB_HIP(ctx, tb::xh::configure_stepx_kernel());
    TB_HIP(ctx, tb::xb::configure_stepx_kernel());
    TB_HIP(ctx, tb::xb3::configure_stepx_kernel());
    TB_HIP(ctx, tb::xba::configure_stepx_kernel());
    TB_HIP(ctx, tb::configure_rule_kernels());
    TB_HIP(ctx, tb::xh::configure_encodex_kernels());
    Arena a;
    Stage s{ctx};
    tb::PolicyW& pw = ctx->pw;
    for (int i = 0; i < 3; ++i) {
        const std::string si = std::to_string(i);
        pw.as2pl[i] = add_xlayer(a, s, "model.transformer_as2pl.layers." + si);
        pw.as2tl[i] = add_xlayer(a, s, "model.transformer_as2tl.layers." + si);
        pw.inter[i] = add_xlayer(a, s, "model.agent_interaction.transformer.layers." + si);
        pw.gru[i] = add_gru(a, s, "model.agent_temporal.rnn", i);
    }
    {
        tb::EncMlpW e = add_enc(a, s, "model.agent_encoder", 11);
        pw.enc_w1 = e.w1; pw.enc_b1 = e.b1; pw.enc_w2 = e.w2; pw.enc_b2 = e.b2;
    }
    pw.pe_fxy = add_freqs(a, s, "pre_processing.input.pose_pe_agent.pe_xy.freqs", 24);
    pw.pe_fyaw = add_freqs(a, s, "pre_processing.input.pose_pe_agent.pe_yaw.freqs", 48);
    const int gl[3] = {0, 4, 8};
    for (int i = 0; i < 3; ++i) {
        const std::string p = "model.add_goal.mlp_in.fc_layers.";
        pw.goal_in_w[i] = add_packed(a, s, p + std::to_string(gl[i]) + ".weight", 128, 128, 128);
        pw.goal_in_b[i] = add_plain(a, s, p + std::to_string(gl[i]) + ".bias", 128);
        pw.goal_in_g[i] = add_plain(a, s, p + std::to_string(gl[i] + 1) + ".weight", 128);
        pw.goal_in_be[i] = add_plain(a, s, p + std::to_string(gl[i] + 1) + ".bias", 128);
    }
    pw.goal_out_w1 = add_packed(a, s, "model.add_goal.mlp_out.fc_layers.0.weight", 128, 256, 256);
    pw.goal_out_b1 = add_plain(a, s, "model.add_goal.mlp_out.fc_layers.0.bias", 128);
    pw.goal_out_w2 = add_packed(a, s, "model.add_goal.mlp_out.fc_layers.3.weight", 128, 128, 128);
    pw.goal_out_b2 = add_plain(a, s, "model.add_goal.mlp_out.fc_layers.3.bias", 128);
    pw.lat_in_w1 = add_packed(a, s, "model.add_latent.mlp_in.fc_layers.0.weight", 128, 16, 16);
    pw.lat_in_b1 = add_plain(a, s, "model.add_latent.mlp_in.fc_layers.0.bias", 128);
    pw.lat_in_w2 = add_packed(a, s, "model.add_latent.mlp_in.fc_layers.3.weight", 128, 128, 128);
    pw.lat_in_b2 = add_plain(a, s, "model.add_latent.mlp_in.fc_layers.3.bias", 128);
    pw.lat_out_w1 = add_packed(a, s, "model.add_latent.mlp_out.fc_layers.0.weight", 128, 256, 256);
    pw.lat_out_b1 = add_plain(a, s, "model.add_latent.mlp_out.fc_layers.0.bias", 128);
    pw.lat_out_w2 = add_packed(a, s, "model.add_latent.mlp_out.fc_layers.3.weight", 128, 128, 128);
    pw.lat_out_b2 = add_plain(a, s, "model.add_latent.mlp_out.fc_layers.3.bias", 128);
    for (int i = 0; i < 3; ++i) {
        const std::string p = "action_head.mlp_mean." + std::to_string(i) + ".fc_layers.";
        pw.head_w1[i] = add_packed(a, s, p + "0.weight", 128, 128, 128);
        pw.head_b1[i] = add_plain(a, s, p + "0.bias", 128);
        pw.head_w2[i] = add_plain(a, s, p + "2.weight", 2 * 128);
        pw.head_b2[i] = add_plain(a, s, p + "2.bias", 2);
        pw.head_log_std[i] = add_plain(a, s, "action_head.log_std." + std::to_string(i), 2);
    }
    pw.latent_log_std = add_plain(a, s, "model.latent_encoder.latent_prior_dist.log_std", 16);
    for (int i = 0; i < 3; ++i) {
        pw.max_acc[i] = ctx->cfg.max_acc[i];
        pw.max_yaw_rate[i] = ctx->cfg.max_yaw_rate[i];
    }
    pw.dt = ctx->cfg.dt;
    // ---- the same policy Linears once more in the XDL packing of the configured operand precision (fp16 pairs / bf16)
    {
        const bool bf16 = ctx->cfg.operand_precision == 1;
        g_pack_bf16 = bf16;
        tb::PolicyWX& px = bf16 ? ctx->pxb : ctx->px;
        for (int i = 0; i < 3; ++i) {
            const std::string si = std::to_string(i);
            px.as2pl[i] = add_xlayer_x(a, s, "model.transformer_as2pl.layers." + si);
            px.as2tl[i] = add_xlayer_x(a, s, "model.transformer_as2tl.layers." + si);
            px.inter[i] = add_xlayer_x(a, s, "model.agent_interaction.transformer.layers." + si);
            {
                const std::string lp = "model.agent_interaction.transformer.layers." + si;
                auto w = s.get(lp + ".attn.in_proj_weight", 384 * 128);
                auto b = s.get(lp + ".attn.in_proj_bias", 384);
                auto g = s.get(lp + ".norm_tgt.weight", 128);
                auto be = s.get(lp + ".norm_tgt.bias", 128);
                if (w && b && g && be) {
                    std::vector<float> wf(256 * 128), bf(256);
                    for (int r = 0; r < 256; ++r) {
                        double acc = (*b)[128 + r];
                        for (int c = 0; c < 128; ++c) {
                            const float wv = (*w)[(size_t)(128 + r) * 128 + c];
                            wf[(size_t)r * 128 + c] = wv * (*g)[c];
                            acc += (double)wv * (double)(*be)[c];
                        }
                        bf[r] = (float)acc;
                    }
                    xdl_range_ok(s, lp + ".attn.in_proj_weight (K/V rows x norm_tgt.weight)", wf.data(), wf.size());
                    px.inter_kvf[i] = a.add(bf16 ? pack_xdl_bf16(wf.data(), 256, 128) : pack_xdl(wf.data(), 256, 128));
                    px.inter_bkvf[i] = a.add(bf);
                }
            }
            px.gru[i].wih = add_xdl(a, s, "model.agent_temporal.rnn.weight_ih_l" + si, 384, 128);
            px.gru[i].whh = add_xdl(a, s, "model.agent_temporal.rnn.weight_hh_l" + si, 384, 128);
            px.head_w1[i] = add_xdl(a, s, "action_head.mlp_mean." + si + ".fc_layers.0.weight", 128, 128);
        }
        px.goal_out_w1 = add_xdl(a, s, "model.add_goal.mlp_out.fc_layers.0.weight", 128, 256);
        px.goal_out_w2 = add_xdl(a, s, "model.add_goal.mlp_out.fc_layers.3.weight", 128, 128);
        px.lat_out_w1 = add_xdl(a, s, "model.add_latent.mlp_out.fc_layers.0.weight", 128, 256);
        px.lat_out_w2 = add_xdl(a, s, "model.add_latent.mlp_out.fc_layers.3.weight", 128, 128);
        g_pack_bf16 = false;
    }

    // ---- scene-encoder weights
    tb::EncoderW& ew = ctx->ew;
    ew.map_enc = add_enc(a, s, "model.map_encoder.input_pe_encoder", 31);
    ew.tl_enc = add_enc(a, s, "model.tl_encoder", 5);
    ew.agent_enc = add_enc(a, s, "model.agent_encoder", 11);
    ew.pe_fxy = pw.pe_fxy;
    ew.pe_fyaw = pw.pe_fyaw;
    for (int i = 0; i < 3; ++i) {
        const std::string si = std::to_string(i);
        ew.densetnt[i] = add_xlayer(a, s, "model.map_encoder.transformer_densetnt.layers." + si);
        ew.inter_prior[i] = add_xlayer(a, s, "model.latent_encoder.agent_interaction_prior.transformer.layers." + si);
        ew.gru_prior[i] = add_gru(a, s, "model.latent_encoder.agent_temporal_prior.rnn", i);
        ew.inter_post[i] = add_xlayer(a, s, "model.latent_encoder.agent_interaction_post.transformer.layers." + si);
        ew.gru_post[i] = add_gru(a, s, "model.latent_encoder.agent_temporal_post.rnn", i);
        ew.gru_dest[i] = add_gru(a, s, "model.goal_manager.goal_predictor.gru_as.rnn", i);
        ew.as2pl[i] = pw.as2pl[i];
        ew.as2tl[i] = pw.as2tl[i];
    }
    ew.map_self = add_xlayer(a, s, "model.map_encoder.transformer_self_attn.layers.0");
    g_pack_bf16 = false;  // the encoders keep fp32-accurate fp16 pairs in either operand_precision
    for (int i = 0; i < 3; ++i) {
        const std::string si = std::to_string(i);
        ew.densetnt_x[i] = add_xlayer_x(a, s, "model.map_encoder.transformer_densetnt.layers." + si);
        ew.as2pl_x[i] = add_xlayer_x(a, s, "model.transformer_as2pl.layers." + si);
        ew.as2tl_x[i] = add_xlayer_x(a, s, "model.transformer_as2tl.layers." + si);
        ew.inter_prior_x[i] = add_xlayer_x(a, s, "model.latent_encoder.agent_interaction_prior.transformer.layers." + si);
        ew.inter_post_x[i] = add_xlayer_x(a, s, "model.latent_encoder.agent_interaction_post.transformer.layers." + si);
    }
    ew.map_self_x = add_xlayer_x(a, s, "model.map_encoder.transformer_self_attn.layers.0");
    for (int i = 0; i < 3; ++i) {
        const std::string si = std::to_string(i);
        const char* names[3] = {"model.latent_encoder.agent_temporal_prior.rnn", "model.goal_manager.goal_predictor.gru_as.rnn",
                                "model.latent_encoder.agent_temporal_post.rnn"};
        tb::GruLayerX* dst[3] = {ew.gru_prior_x, ew.gru_dest_x, ew.gru_post_x};
        for (int k = 0; k < 3; ++k) {
            dst[k][i].wih = add_xdl(a, s, std::string(names[k]) + ".weight_ih_l" + si, 384, 128);
            dst[k][i].whh = add_xdl(a, s, std::string(names[k]) + ".weight_hh_l" + si, 384, 128);
        }
    }
    ew.lat_w1 = add_packed(a, s, "model.latent_encoder.latent_prior_dist.mlp_mean.fc_layers.0.weight", 128, 128, 128);
    ew.lat_b1 = add_plain(a, s, "model.latent_encoder.latent_prior_dist.mlp_mean.fc_layers.0.bias", 128);
    ew.lat_w2 = add_plain(a, s, "model.latent_encoder.latent_prior_dist.mlp_mean.fc_layers.2.weight", 16 * 128);
    ew.lat_b2 = add_plain(a, s, "model.latent_encoder.latent_prior_dist.mlp_mean.fc_layers.2.bias", 16);
    ew.post_w1 = add_packed(a, s, "model.latent_encoder.latent_post_dist.mlp_mean.fc_layers.0.weight", 128, 128, 128);
    ew.post_b1 = add_plain(a, s, "model.latent_encoder.latent_post_dist.mlp_mean.fc_layers.0.bias", 128);
    ew.post_w2 = add_plain(a, s, "model.latent_encoder.latent_post_dist.mlp_mean.fc_layers.2.weight", 16 * 128);
    ew.post_b2 = add_plain(a, s, "model.latent_encoder.latent_post_dist.mlp_mean.fc_layers.2.bias", 16);
    ew.post_log_std = add_plain(a, s, "model.latent_encoder.latent_post_dist.log_std", 16);
    {
        // dest predictor first layer [128][256] split into the map half (cols 0:128) and the agent half (128:256)
        const std::string p = "model.goal_manager.goal_predictor.mlp.fc_layers.";
        auto w0 = s.get(p + "0.weight", 128 * 256);
        if (w0) {
            std::vector<float> wm(128 * 128), wa(128 * 128);
            for (int o = 0; o < 128; ++o)
                for (int k = 0; k < 128; ++k) {
                    wm[o * 128 + k] = (*w0)[o * 256 + k];
                    wa[o * 128 + k] = (*w0)[o * 256 + 128 + k];
                }
            ew.dest_w0_map = a.add(pack_mfma(wm.data(), 128, 128, 128));
            ew.dest_w0_agent = a.add(pack_mfma(wa.data(), 128, 128, 128));
        }
        ew.dest_b0 = add_plain(a, s, p + "0.bias", 128);
        ew.dest_ln0_g = add_plain(a, s, p + "1.weight", 128);
        ew.dest_ln0_b = add_plain(a, s, p + "1.bias", 128);
        ew.dest_w1 = add_packed(a, s, p + "3.weight", 128, 128, 128);
        ew.dest_w1_x = add_xdl(a, s, p + "3.weight", 128, 128);
        ew.dest_b1 = add_plain(a, s, p + "3.bias", 128);
        ew.dest_ln1_g = add_plain(a, s, p + "4.weight", 128);
        ew.dest_ln1_b = add_plain(a, s, p + "4.bias", 128);
        ew.dest_w2 = add_plain(a, s, p + "6.weight", 128);
        ew.dest_b2 = add_plain(a, s, p + "6.bias", 1);
    }
    if (!s.ok) return 1;
    // ---- fp16-pair range of the loaded tensors: the reference has no range limit (src/models/modules/mlp.py:20-85), so a checkpoint
    // outside it is not refused -- the context falls back to the exact-fp32 twins (fp32 MFMA step kernel k_step, fp32-MFMA encoder
    // blocks), which have fp32's range, and says so (tb_precision_state; once on stderr)
    // (every load starts from the configured kernels: a context that was downgraded -- by an earlier checkpoint or by a run-time
    // overflow under it -- gets the XDL kernels back with in-range weights; an open stepwise rollout belongs to the old weights)
    ctx->step_kernel = ctx->step_kernel0;
    ctx->encode_kernel = ctx->encode_kernel0;
    if (!ctx->warm_tabs.empty()) {  // (the L2 warmers' tables hold arena offsets of the previous weights)
        (void)hipDeviceSynchronize();
        for (auto& kv : ctx->warm_tabs)
            if (kv.second.d) (void)hipFree(kv.second.d);
        ctx->warm_tabs.clear();
    }
    ctx->precision_reason = 0;
    ctx->precision_note.clear();
    ctx->step_active = false;
    if (s.range_hit && ctx->cfg.operand_precision != 2) {
        ctx->encode_kernel = 0;
        if (ctx->cfg.operand_precision == 0) ctx->step_kernel = 0;
        ctx->precision_reason |= 1;
        ctx->precision_note = s.range_what + (ctx->cfg.operand_precision == 0
                                  ? ": this context runs on the exact-fp32 kernels (fp32 MFMA; about 2x slower per step, same results class)"
                                  : ": the scene encoders of this context run on their exact-fp32 kernels (the bf16 step kernels have fp32's range)");
        fprintf(stderr, "trafficbots_hip: %s\n", ctx->precision_note.c_str());
    }

    if (ctx->d_arena) {
        TB_HIP(ctx, hipFree(ctx->d_arena));
        ctx->d_arena = nullptr;
    }
    ctx->arena_floats = a.h.size();
    TB_HIP(ctx, hipMalloc((void**)&ctx->d_arena, a.h.size() * sizeof(float)));
    TB_HIP(ctx, hipMemcpyAsync(ctx->d_arena, a.h.data(), a.h.size() * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    // row-major copies of the policy trunk's tensors for tb_forward (5.7 MB; tb_forward_kernels.hip)
    {
        std::vector<float> rawh;
        std::vector<std::pair<std::string, size_t>> offs;
        for (const auto& kv : ctx->staged) {
            const std::string& k = kv.first;
            if (k.rfind("model.transformer_as2pl.", 0) && k.rfind("model.transformer_as2tl.", 0) && k.rfind("model.agent_interaction.", 0) &&
                k.rfind("model.agent_temporal.", 0) && k.rfind("model.add_goal.", 0) && k.rfind("model.add_latent.", 0))
                continue;
            offs.emplace_back(k, rawh.size());
            rawh.insert(rawh.end(), kv.second.begin(), kv.second.end());
            while (rawh.size() & 3) rawh.push_back(0.f);
        }
        if (ctx->d_raw) TB_HIP(ctx, hipFree(ctx->d_raw));
        ctx->d_raw = nullptr;
        ctx->raw.clear();
        if (!rawh.empty()) {
            TB_HIP(ctx, hipMalloc((void**)&ctx->d_raw, rawh.size() * sizeof(float)));
            TB_HIP(ctx, hipMemcpy(ctx->d_raw, rawh.data(), rawh.size() * sizeof(float), hipMemcpyHostToDevice));
            for (const auto& o : offs) ctx->raw[o.first] = ctx->d_raw + o.second;
        }
    }
    TB_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));  // host staging buffer `a` dies at return
    ctx->staged.clear();
    ctx->finalized = true;
    return 0;
}

extern "C" int tb_forward(tb_ctx* ctx, const tb_forward_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    if (!ctx->finalized) return tb_fail(ctx, "tb_forward: weights not finalized");
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_inst <= 0 || io->n_agent <= 0 || io->n_pl <= 0 || io->n_tl <= 0) return tb_fail(ctx, "tb_forward: empty dimension");
    if (io->n_agent > 4096 || io->n_pl > 8192 || io->n_tl > 8192) return tb_fail(ctx, "tb_forward: more than 8192 keys per attention are not supported");
    if (io->n_inst > 65535) return tb_fail(ctx, "tb_forward: more than 65535 instances per call (grid.y) are not supported");
    const void* need[] = {io->agent_valid, io->agent_feature, io->map_valid, io->map_feature, io->tl_valid, io->tl_feature, io->latent_sample,
                          io->hidden, io->policy_feature};
    for (const void* q : need)
        if (!q) return tb_fail(ctx, "tb_forward: a required buffer pointer is NULL");
    if ((io->goal_feature == nullptr) != (io->goal_valid == nullptr)) return tb_fail(ctx, "tb_forward: goal_feature and goal_valid go together");
    const size_t nf = tb::forward_scratch_floats(io);
    if (nf > ctx->fw_floats) {
        if (ctx->d_fw) {
            TB_HIP(ctx, hipDeviceSynchronize());
            TB_HIP(ctx, hipFree(ctx->d_fw));
            ctx->d_fw = nullptr;
            ctx->fw_floats = 0;
        }
        TB_HIP(ctx, hipMalloc((void**)&ctx->d_fw, nf * sizeof(float)));
        ctx->fw_floats = nf;
    }
    const char* missing = tb::run_forward(ctx->raw, io, ctx->d_fw, (hipStream_t)stream_);
    if (missing) return tb_fail(ctx, "tb_forward: weight tensor %s was not loaded", missing);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

int tb_set_timing(tb_ctx* ctx, int enable) {
    if (!ctx) return 1;
    ctx->timing = enable != 0;
    return 0;
}

int tb_get_timing(tb_ctx* ctx, float* out4) {
    if (!ctx || !out4) return 1;
    out4[0] = out4[1] = out4[2] = 0.f;
    out4[3] = 0.f;
    const int n = ctx->n_timed_steps;  // number of step launches of the last rollout (S + 1)
    if (n < 2) return 0;
    TB_HIP(ctx, hipEventSynchronize(ctx->ev[1 + n]));
    float ms = 0.f;
    TB_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    out4[2] = ms;
    // per launch: 2 = fused C(t)+A(t+1), 1 = one half only (A(1) alone, C(S) alone, the C-only steps of a batched warm start --
    // whose batched A launch is charged to the first of them), 0 = nothing launched
    int n_fused = 0;
    for (int i = 0; i < n; ++i) {
        TB_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[1 + i], ctx->ev[2 + i]));
        if (ctx->launch_kind[i] == 2) {
            out4[0] += ms;
            ++n_fused;
        } else {
            out4[1] += ms;
        }
    }
    out4[3] = (float)n_fused;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
void* tb_ws_guard_take(size_t bytes) {
    static const int mode = [] {
        const char* e = getenv("TB_WS_GUARD");
        return e ? atoi(e) : 0;
    }();
    if (mode != 1 && mode != 2) return nullptr;
    static bool said = false;
    if (!said) {
        said = true;
        fprintf(stderr, "TB_WS_GUARD=%d: every workspace carve is a mapping of its own between unmapped pages (debug mode)\n", mode);
    }
    static size_t gran = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    prop.location.id = dev;
    if (!gran && hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return nullptr;
    const size_t need = bytes ? bytes : 16, n = (need + gran - 1) / gran, total = (n + 2) * gran;
    void* va = nullptr;
    hipMemGenericAllocationHandle_t h;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemAddressReserve(&va, total, gran, nullptr, 0) != hipSuccess || hipMemCreate(&h, n * gran, &prop, 0) != hipSuccess) {
        fprintf(stderr, "TB_WS_GUARD: reserve / create of %zu bytes failed\n", need);
        abort();  // (a debug run that silently fell back to the plain workspace would prove nothing)
    }
    char* mid = static_cast<char*>(va) + gran;
    if (hipMemMap(mid, n * gran, 0, h, 0) != hipSuccess || hipMemSetAccess(mid, n * gran, &acc, 1) != hipSuccess) {
        fprintf(stderr, "TB_WS_GUARD: map of %zu bytes failed\n", need);
        abort();
    }
    (void)hipMemRelease(h);
    (void)hipMemset(mid, 0, n * gran);
    return mode == 2 ? mid : mid + n * gran - ((need + 15) & ~(size_t)15);
}

int tb_ensure_workspace(tb_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return 0;
    if (ctx->d_ws) {
        TB_HIP(ctx, hipDeviceSynchronize());
        TB_HIP(ctx, hipFree(ctx->d_ws));
        ctx->d_ws = nullptr;
        ctx->ws_bytes = 0;
    }
    bytes += bytes / 8;
    TB_HIP(ctx, hipMalloc((void**)&ctx->d_ws, bytes));
    ctx->ws_bytes = bytes;
    return 0;
}

int tb_ensure_workspace_enc(tb_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_enc_bytes) return 0;
    if (ctx->d_ws_enc) {
        TB_HIP(ctx, hipDeviceSynchronize());
        TB_HIP(ctx, hipFree(ctx->d_ws_enc));
        ctx->d_ws_enc = nullptr;
        ctx->ws_enc_bytes = 0;
    }
    bytes += bytes / 8;
    TB_HIP(ctx, hipMalloc((void**)&ctx->d_ws_enc, bytes));
    ctx->ws_enc_bytes = bytes;
    return 0;
}

static void carve_rollout(tb::RolloutP& p, Carver& c, bool with_gh) {
    const size_t B = p.n_scene, N = p.n_inst, NH = p.n_tl_hist;
    p.kpl = c.take<float>(B * 3 * p.p_pad * 128);
    p.vtpl = c.take<float>(B * 3 * 128 * p.p_pad);
    p.kbias_pl = c.take<float>(B * p.p_pad);
    p.ktl = c.take<float>(B * NH * 3 * p.t_pad * 128);
    p.vttl = c.take<float>(B * NH * 3 * 128 * p.t_pad);
    p.kbias_tl = c.take<float>(B * NH * p.t_pad);
    p.nkey_pl = c.take<int>(B);
    p.nkey_tl = c.take<int>(B * NH);
    p.state = c.take<float>(N * p.a_pad * 4);
    p.aux = c.take<float>(N * p.a_pad * 4);
    for (int k = 0; k < 2; ++k) {
        p.valid_b[k] = c.take<uint8_t>(N * p.a_pad);
        p.vbias_b[k] = c.take<float>(N * p.a_pad);
    }
    p.killed = c.take<uint8_t>(N * p.a_pad);
    p.goal_valid = c.take<uint8_t>(N * p.a_pad);
    p.dest_reached = c.take<uint8_t>(N * p.a_pad);
    p.outside = c.take<uint8_t>(N * p.a_pad);
    p.hidden = c.take<float>(3 * N * p.a_pad * 128);
    p.x_mid = c.take<float>(N * p.a_pad * 128);
    p.x_mid_w = p.x_mid;
    if (p.pre_t0 > 0) {  // (pre_t0 doubles as the number of batched warm-start slices while the workspace is carved)
        const size_t n_pre = (size_t)p.pre_t0;
        p.x_mid_pre = c.take<float>(n_pre * N * p.a_pad * 128);
        p.kin_pre = c.take<float>(n_pre * N * 3 * p.a_pad * 128);
        p.vtin_pre = c.take<float>(n_pre * N * 3 * 128 * p.a_pad);
        p.x_int_pre = c.take<float>(n_pre * N * p.a_pad * 128);
        p.vbias_pre = c.take<float>(n_pre * N * p.a_pad);
    }
    for (int k = 0; k < 2; ++k) {
        p.kin_b[k] = c.take<float>(N * 3 * p.a_pad * 128);
        p.vtin_b[k] = c.take<float>(N * 3 * 128 * p.a_pad);
    }
    tb::set_parity(p, 0);
    p.goal_pre = c.take<float>(N * p.a_pad * 128);
    p.lat_pre = c.take<float>(N * p.a_pad * 128);
    p.dest_geo = c.take<float>(N * p.a_pad * 80);
    p.dest_flag = c.take<int>(N * p.a_pad);
    p.prof = c.take<long long>(N * (p.a_pad / 16) * 32);
    p.gh = nullptr;
    p.gh_flag = nullptr;
    p.kv_flag = nullptr;
    if (with_gh) {  // the helper workgroups are on, see rollout_setup
        p.gh = c.take<float>(N * (p.a_pad / 16) * (size_t)18432);
        p.gh_flag = c.take<unsigned int>(N * (p.a_pad / 16) * 3);  // [tiles] GRU flags, then [tiles][2] K/V flags (one memset)
        p.kv_flag = p.gh_flag + N * (p.a_pad / 16);
    }
}

static int rollout_setup(tb_ctx* ctx, const tb_rollout_io* io, tb::RolloutP& p, tb_stream upload_stream) {
    if (!ctx || !io) return 1;
    if (!ctx->finalized) return tb_fail(ctx, "tb_rollout: weights not finalized");
    TB_HIP(ctx, hipSetDevice(ctx->device));  // the context is bound to the device that was current at tb_create
    if (io->n_scene <= 0 || io->k_futures <= 0 || io->n_agent <= 0 || io->n_pl <= 0 || io->n_tl <= 0)
        return tb_fail(ctx, "tb_rollout: empty dimension (B=%d K=%d A=%d P=%d T=%d)", io->n_scene, io->k_futures, io->n_agent,
                       io->n_pl, io->n_tl);
    if (io->n_agent > 256) return tb_fail(ctx, "tb_rollout: n_agent %d > 256 not supported", io->n_agent);
    if (io->n_hist < 1 || io->n_tl_step < 0) return tb_fail(ctx, "tb_rollout: n_hist %d / n_tl_step %d", io->n_hist, io->n_tl_step);
    const int step_start = ctx->cfg.time_step_sim_start;
    if (io->step_end < step_start) return tb_fail(ctx, "tb_rollout: step_end < time_step_sim_start");
    memset(&p, 0, sizeof(p));
    p.W = ctx->d_arena;
    p.pw = ctx->pw;
    p.px = ctx->step_kernel == 3 ? ctx->pxb : ctx->px;
    p.n_scene = io->n_scene;
    p.k_rep = io->k_futures;
    p.n_inst = io->n_scene * io->k_futures;
    p.n_agent = io->n_agent;
    p.a_pad = padk(io->n_agent);
    p.n_pl = io->n_pl;
    p.p_pad = padk(io->n_pl);
    p.n_tl = io->n_tl;
    p.t_pad = padk(io->n_tl);
    p.n_hist = io->n_hist;
    p.n_tl_hist = io->n_tl_step > 0 ? io->n_tl_step : io->n_hist;
    p.latent_log_std = io->latent_posterior ? ctx->ew.post_log_std : ctx->pw.latent_log_std;
    p.step_start = step_start;
    p.n_step_out = io->step_end - step_start + 1;
    p.map_feature = io->map_feature;
    p.tl_feature = io->tl_feature;
    p.hist_valid = io->agent_valid;
    p.hist_state = io->agent_state;
    p.hist_vel = io->agent_vel;
    p.hist_acc = io->agent_acc;
    p.hist_yaw_rate = io->agent_yaw_rate;
    p.tf_mask = io->mask_teacher_forcing;
    p.agent_type = io->agent_type;
    p.agent_size = io->agent_size;
    p.map_boundary = io->map_boundary;
    p.map_valid = io->map_valid;
    p.map_type = io->map_type;
    p.map_pos = io->map_pos;
    p.map_dir = io->map_dir;
    // the personality: given by the caller, or drawn by the prologue into the caller's latent_sample_out (k_rollout_init)
    p.latent_draw = io->latent_sample_out != nullptr;
    p.latent_z = p.latent_draw ? io->latent_sample_out : io->latent_sample;
    p.o_latent_z = io->latent_sample_out;
    p.latent_eps = io->latent_eps;
    p.latent_det = io->latent_deterministic;
    p.action_eps = io->action_eps;
    p.latent_mean = io->latent_mean;
    p.dest = io->dest;
    p.goal_valid0 = io->goal_valid;
    p.preds = io->preds;
    p.o_valid = io->valid;
    p.o_override = io->override_masks;
    p.o_outside = io->outside_map;
    p.o_outside_this = io->outside_map_this_step;
    p.o_dest_reached = io->dest_reached;
    p.o_dest_reached_this = io->dest_reached_this_step;
    p.o_action_logp = io->action_log_probs;
    p.o_latent_logp = io->latent_log_prob;
    p.o_check_state = io->check_state;
    p.o_check_valid = io->check_valid;
    p.tap_step = io->tap_step;
    p.o_action = io->actions;
    p.tap_policy_feature = io->tap_policy_feature;
    p.tap_agent_feature = io->tap_agent_feature;
    const void* required[] = {p.map_feature, io->map_feature_valid, p.tl_feature, io->tl_feature_valid, p.hist_valid, p.hist_state,
                              p.hist_vel, p.hist_acc, p.hist_yaw_rate, p.tf_mask, p.agent_type, p.agent_size, p.map_boundary,
                              p.map_valid, p.map_type, p.map_pos, p.map_dir, p.latent_z, p.latent_mean, p.dest, p.goal_valid0,
                              p.preds, p.o_valid, p.o_override, p.o_outside, p.o_outside_this, p.o_dest_reached,
                              p.o_dest_reached_this, p.o_action_logp, p.o_latent_logp};
    for (const void* q : required)
        if (!q) return tb_fail(ctx, "tb_rollout: a required buffer pointer is NULL");
    // batched warm start (XDL kernels only): A halves of steps step_start .. Wp + 1, Wp = min(W, n_hist - 1, step_end - 1)
    int n_pre = 0;
    if (ctx->step_kernel >= 2 && io->warm_start_steps > 0) {
        const int wp = std::min(std::min(io->warm_start_steps, io->n_hist - 1), io->step_end - 1);
        n_pre = std::max(0, wp - (step_start - 1) + 1);
    }
    p.pre_t0 = n_pre;  // (carve_rollout reads the slice count here)
    // helper workgroups: a launch of at most 128 tiles leaves at least half of the 256 CUs idle (tb_rollout.hpp)
    const TbSw sw = tb_switches_now(ctx);
    p.sw_lean_off = sw.lean_off;
    // another live context of this device launched work within the last 25 ms: its kernels want the CUs a 128-tile launch leaves idle.
    // (The window only has to span the gap between a pipelined neighbour's launches -- the rollouts this matters for, <= 128 tiles,
    // last ~7 ms -- and should not outlive the neighbour by much: with the first choice, 100 ms, ten plain calls after the last
    // pipelined one ran without helpers, 0.5 ms per rollout slower.)
    bool neighbour_active = false;
    if (sw.warm == 0 || sw.helpers_off == 0) {
        const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        for (tb_ctx* o : g_ctx_live)
            if (o != ctx && o->device == ctx->device && o->last_launch_ns && now - o->last_launch_ns < 25000000LL) neighbour_active = true;
    }
    bool with_gh = false;
    {
        // (helpers: tb_switches.step_helpers.  Automatic = on while this context has the device to itself; with a second rollout in
        // flight the two launches' 2 x 128 tiles fill the 256 CUs and 2 x 128 helper workgroups only queue in front of them:
        // wm.pipeline(lanes=2) 5.59 -> 5.35 ms per batch, profiles/r06_experiments.txt item 14.  Same results either way.)
        const bool helpers_on = sw.helpers_off == 2 || (sw.helpers_off == 0 && !neighbour_active);
        with_gh = ctx->step_kernel >= 2 && (size_t)p.n_inst * (p.a_pad / 16) <= 128 && helpers_on;
        if (ctx->step_kernel == 3 && aw_forced(sw) && p.p_pad >= 512) with_gh = false;  // (test switch: the assist carve has no helpers)
    }
    if (!ctx->d_status) {
        TB_HIP(ctx, hipMalloc((void**)&ctx->d_status, 2 * sizeof(unsigned int)));
        TB_HIP(ctx, hipMemset(ctx->d_status, 0, 2 * sizeof(unsigned int)));
    }
    p.sync_err = ctx->d_status + 1;
    if (with_gh && ctx->step_kernel >= 2) {
        // L2 warmers (tb_stepx_kernels.hip): request times of the weight units of a fused launch, cycles since launch start.  The
        // stage durations come from the stage profile of the -DTB_PROFILE build (profiles/stage_constants.json -> tools/
        // gen_warm_schedule.py -> tb_warm_schedule.inc: WS_*; fp16 pairs, headline shape; the key-block slopes extend it to other
        // shapes; bf16 operands: x 0.7).  Only what follows the helpers' own work is listed.  TB_STEP_WARM=0: off.
        // On while this context has the device to itself: the warmers hold two helper workgroups per XCD for the whole launch, +1.4 %
        // for ONE rollout in flight and a 15 % loss when another context's launches want those CUs (bench.py two_batches_in_flight).
        // tb_switches.step_l2_warmers = 0: automatic (off when another live context of this device launched work within the last
        // 25 ms), 1: off, 2: on regardless.
        const bool warm_on = sw.warm == 2 || (sw.warm == 0 && !neighbour_active);
        const long long key = ((long long)p.p_pad << 32) | ((long long)p.a_pad << 8) | ctx->step_kernel;
        if (!warm_on) {
            p.warm_tab = nullptr;
        } else {
            tb_ctx::WarmTab& wt = ctx->warm_tabs[key];
            if (!wt.d) {
                const tb::PolicyWX& x = ctx->step_kernel == 3 ? ctx->pxb : ctx->px;
                const double sc = ctx->step_kernel == 3 ? 0.7 : 1.0;
                const uint32_t gate = ctx->step_kernel == 3 ? 8192u : 16384u;  // floats of one 128 x 128 unit in the arena (bf16: one plane)
                std::vector<int>& tab = wt.host;
                tab.clear();
                auto add = [&](uint32_t off, double t) { tab.push_back((int)off); tab.push_back((int)(t * sc)); };
                const double t_inter = WS_INTER_BASE + WS_INTER_PER_BLOCK * (p.a_pad / 32), t_pl = WS_PL_BASE + WS_PL_PER_BLOCK * (p.p_pad / 32),
                             t_tl = WS_TL_LAYER;
                double t = WS_PROLOGUE + 3 * t_inter;  // the GRU: a unit is requested one unit ahead of its use
                for (int l = 0; l < 3; ++l)
                    for (int g = 0; g < 3; ++g) add(x.gru[l].wih + (uint32_t)g * gate, t + (l * 3 + g - 1) * WS_GRU_UNIT);
                t += WS_GRU;
                add(x.goal_out_w1, t - WS_GRU_UNIT); add(x.goal_out_w2, t); add(x.lat_out_w1, t + 0.5 * WS_ADD_GOAL);
                add(x.lat_out_w2, t + WS_ADD_GOAL);
                t += WS_ADD_GOAL + WS_ADD_LATENT;
                for (int ty = 0; ty < 3; ++ty) add(x.head_w1[ty], t + (ty - 1) * WS_HEAD / 4.0);
                t += WS_HEAD + WS_EPILOGUE_AND_FRONT;  // head, epilogue, the A half's front end
                for (int l = 0; l < 3; ++l, t += t_pl) {
                    add(x.as2pl[l].wq, t - WS_LAYER_WQ_LEAD); add(x.as2pl[l].wo, t + WS_LAYER_WO_AT);
                    add(x.as2pl[l].w1, t + t_pl - WS_LAYER_W1_BEFORE_END); add(x.as2pl[l].w2, t + t_pl - WS_LAYER_W2_BEFORE_END);
                }
                for (int l = 0; l < 3; ++l, t += t_tl) {
                    add(x.as2tl[l].wq, t - WS_LAYER_WQ_LEAD); add(x.as2tl[l].wo, t + WS_LAYER_WO_AT);
                    add(x.as2tl[l].w1, t + t_tl - WS_LAYER_W1_BEFORE_END); add(x.as2tl[l].w2, t + t_tl - WS_LAYER_W2_BEFORE_END);
                }
                add(x.inter_kvf[0], t - WS_LAYER_WQ_LEAD);
                // one table per key, written once (never rewritten: launches in flight / captured graphs keep theirs), uploaded on the
                // caller's stream in front of the launches that read it -- no blocking copy inside tb_rollout
                TB_HIP(ctx, hipMalloc((void**)&wt.d, 128 * sizeof(int)));
                TB_HIP(ctx, hipMemcpyAsync(wt.d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)upload_stream));
                wt.n = (int)tab.size() / 2;
            }
            p.warm_tab = wt.d;
            p.warm_n = wt.n;
        }
    }
    {
        const char* e = getenv("TB_DEBUG_HELPER_DELAY");
        p.dbg_helper_delay = e ? atoi(e) : 0;
    }
    Carver sizing{nullptr};
    carve_rollout(p, sizing, with_gh);
    if (tb_ensure_workspace(ctx, sizing.off + 256)) return 1;
    Carver c{ctx->d_ws};
    carve_rollout(p, c, with_gh);
    p.pre_t0 = 0;
    ctx->n_pre = n_pre;
    ctx->last_prof = p.prof;
    tb_note_launch(ctx);  // (tb_rollout and tb_rollout_begin: this context is launching -- other contexts' L2 warmers stand down)
    return 0;
}

// prologue: hoisted K/V of the map and of every history step's traffic lights; simulator init
static void rollout_prologue(const tb_ctx* ctx, const tb::RolloutP& p, const tb_rollout_io* io, hipStream_t s) {
    if (ctx->step_kernel >= 2) {  // k_step_x reads K / V in XDL operand order (fp16 pairs or bf16)
        const bool bf16k = ctx->step_kernel == 3;
        auto hoist = bf16k ? tb::xb::launch_kv_hoist_x : tb::xh::launch_kv_hoist_x;
        hoist(p.W, p.pw.as2pl, p.px.as2pl, p.map_feature, io->map_feature_valid, p.n_scene, p.n_pl, p.p_pad, p.kpl, p.vtpl, p.kbias_pl,
              p.nkey_pl, s);
        hoist(p.W, p.pw.as2tl, p.px.as2tl, p.tl_feature, io->tl_feature_valid, p.n_scene * p.n_tl_hist, p.n_tl, p.t_pad, p.ktl, p.vttl,
              p.kbias_tl, p.nkey_tl, s);
        tb::launch_rollout_init(p, s);
        // the constant half of add_goal / add_latent's first fusion Linear (reads what k_rollout_init just wrote)
        if (bf16k) tb::xb::launch_fuse_hoist_x(p, s);
        else tb::xh::launch_fuse_hoist_x(p, s);
        if (p.gh_flag) (void)hipMemsetAsync(p.gh_flag, 0, sizeof(unsigned int) * 3 * (size_t)p.n_inst * (p.a_pad / 16), s);
        return;
    }
    tb::launch_kv_hoist(p.W, p.pw.as2pl, p.map_feature, io->map_feature_valid, p.n_scene, p.n_pl, p.p_pad, p.kpl, p.vtpl,
                        p.kbias_pl, s);
    tb::launch_kv_hoist(p.W, p.pw.as2tl, p.tl_feature, io->tl_feature_valid, p.n_scene * p.n_tl_hist, p.n_tl, p.t_pad, p.ktl,
                        p.vttl, p.kbias_tl, s);
    tb::launch_rollout_init(p, s);
}

static int rollout_enqueue(tb_ctx* ctx, const tb_rollout_io* io, const tb::RolloutP& p, hipStream_t s);

// every launch-shaping development switch read below this call (step_launch, rollout_prologue): part of the graph key
// (development-only variables that are not tb_switches fields)
static const char* const kGraphEnv[] = {"TB_DEBUG_HELPER_DELAY", "TB_STEP_KERNEL"};

static void key_append(std::vector<unsigned char>& k, const void* data, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(data);
    k.insert(k.end(), b, b + n);
}

extern "C" int tb_rollout(tb_ctx* ctx, const tb_rollout_io* io, tb_stream stream_) {
    tb::RolloutP p;
    if (rollout_setup(ctx, io, p, stream_)) return 1;
    hipStream_t s = (hipStream_t)stream_;
    ctx->step_active = false;
    // ---- one hipGraph per rollout: every kernel argument of the launch sequence is a function of (p, io, the development switches),
    // so a rollout with the same argument bytes replays the captured graph (bench loops, a serving loop over fixed buffers)
    const TbSw sw = tb_switches_now(ctx);
    const bool use_graph = !sw.graph_off && !ctx->timing && ctx->step_kernel >= 2;
    if (!use_graph) return rollout_enqueue(ctx, io, p, s);
    std::vector<unsigned char> key;
    key.reserve(sizeof(p) + sizeof(*io) + 256);
    key_append(key, &p, sizeof(p));
    key_append(key, io, sizeof(*io));
    if (io->hidden_drop) key_append(key, io->hidden_drop, (size_t)p.n_step_out);  // (host data that shapes the launch sequence)
    key_append(key, &sw, sizeof(sw));  // every launch-shaping switch read below this call (step_launch, rollout_prologue)
    for (const char* name : kGraphEnv) {
        const char* v = getenv(name);
        key_append(key, v ? v : "-", v ? strlen(v) + 1 : 2);
    }
    if (ctx->graph_exec && key == ctx->graph_key) {
        ++ctx->graph_hits;
        TB_HIP(ctx, hipGraphLaunch(ctx->graph_exec, s));
        return 0;
    }
    if (key == ctx->graph_nocapture) return rollout_enqueue(ctx, io, p, s);  // its capture failed once: plain launches from then on
    if (key != ctx->graph_seen) {  // first sight of this argument set: plain launches (a caller with fresh buffers per call never captures)
        ctx->graph_seen = key;
        return rollout_enqueue(ctx, io, p, s);
    }
    // a capture / instantiate failure is not the caller's problem: the error is cleared, the argument set is marked, and the rollout
    // runs as plain launches on the caller's stream -- now and on every later call
    auto fallback = [&]() {
        (void)hipGetLastError();
        ctx->graph_nocapture = key;
        return rollout_enqueue(ctx, io, p, s);
    };
    hipGraph_t graph = nullptr;
    // captured on a private stream (nothing runs there: the launches are only recorded), launched on the caller's -- which may be the
    // legacy default stream, where a capture cannot begin
    if (!ctx->cap_stream && hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        ctx->cap_stream = nullptr;
        return fallback();
    }
    if (hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return fallback();
    const int rc = rollout_enqueue(ctx, io, p, ctx->cap_stream);
    const hipError_t ec = hipStreamEndCapture(ctx->cap_stream, &graph);
    if (rc || ec != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        return fallback();
    }
    if (ctx->graph_exec) {
        (void)hipGraphExecDestroy(ctx->graph_exec);
        ctx->graph_exec = nullptr;
        ctx->graph_key.clear();
    }
    const hipError_t ei = hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !ctx->graph_exec) {
        ctx->graph_exec = nullptr;
        return fallback();
    }
    ctx->graph_key = key;
    ++ctx->graph_captures;
    TB_HIP(ctx, hipGraphLaunch(ctx->graph_exec, s));
    return 0;
}

static bool pre_inter(const tb_ctx* ctx) { return !tb_switches_now(ctx).pre_inter_off; }

extern "C++" void tb_note_launch(tb_ctx* ctx) {
    ctx->last_launch_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int rollout_enqueue(tb_ctx* ctx, const tb_rollout_io* io, const tb::RolloutP& p, hipStream_t s) {
    const int step_start = p.step_start;
    const int n_steps = p.n_step_out;
    const int n_launch = n_steps + 1;
    if (ctx->timing) {
        const size_t need = 2 + (size_t)n_launch;
        while (ctx->ev.size() < need) {
            hipEvent_t e;
            TB_HIP(ctx, hipEventCreate(&e));
            ctx->ev.push_back(e);
        }
        TB_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    }
    rollout_prologue(ctx, p, io, s);
    if (ctx->timing) TB_HIP(ctx, hipEventRecord(ctx->ev[1], s));
    // ---- batched warm start: the A halves of steps t = step_start-1 .. step_start-1+n_pre-1 in one launch (tb_rollout_io.warm_start_steps)
    const int n_pre = ctx->n_pre;
    const int t_pre0 = step_start - 1;
    if (n_pre > 0) {
        if (ctx->step_kernel == 3 && w3_launch(tb_switches_now(ctx), (size_t)(p.a_pad / tb::TM) * p.n_scene * n_pre)) tb::xb3::launch_step_pre_x(p, t_pre0, n_pre, s);
        else if (ctx->step_kernel == 3) tb::xb::launch_step_pre_x(p, t_pre0, n_pre, s);
        else tb::xh::launch_step_pre_x(p, t_pre0, n_pre, s);
        // ... and behind it the interaction blocks of the same steps, once per scene (RolloutP::x_int_pre; TB_STEP_PRE_INTER=0 keeps
        // them in the step-by-step launches: development / A-B switch)
        if (pre_inter(ctx)) {
            if (ctx->step_kernel == 3 && w3_launch(tb_switches_now(ctx), (size_t)(p.a_pad / tb::TM) * p.n_scene * n_pre)) tb::xb3::launch_inter_pre_x(p, t_pre0, n_pre, s);
            else if (ctx->step_kernel == 3) tb::xb::launch_inter_pre_x(p, t_pre0, n_pre, s);
            else tb::xh::launch_inter_pre_x(p, t_pre0, n_pre, s);
        }
    }
    // ---- the sequential loop (waymo_motion.py:269): launch i runs C(start+i-1) then A(start+i); no host sync
    ctx->launch_kind.assign(n_launch, 0);
    for (int i = 0; i < n_launch; ++i) {
        const int t = step_start + i - 1;
        int do_c = i > 0, do_a = i < n_steps;
        tb::RolloutP q = p;
        if (t - t_pre0 < n_pre) do_a = 0;                  // A(t+1) came out of the batched launch
        if (do_c && t - 1 - t_pre0 < n_pre && t - 1 >= t_pre0) {  // C(t) consumes slice t-1-t_pre0 of the batched results
            const size_t z = (size_t)(t - 1 - t_pre0) * p.n_inst * p.a_pad * 128;
            q.x_mid = (pre_inter(ctx) ? p.x_int_pre : p.x_mid_pre) + z;
            q.skip_inter = pre_inter(ctx);
            q.pre_shared = p.k_rep > 1;  // (the slices exist once per scene, in the slot of future 0: no replication)
            q.kin_b[0] = q.kin_b[1] = p.kin_pre + 3 * z;     // (read side only: step_launch picks by parity, the write side is restored below)
            q.vtin_b[0] = q.vtin_b[1] = p.vtin_pre + 3 * z;
        }
        if (do_c || do_a) step_launch(ctx, q, p, t, do_c, do_a, s);
        // train-mode p_drop_hidden with the caller's draws (tb_rollout_io.hidden_drop): the GRU state of every instance is zeroed
        // after step t has been recorded -- the next launch (its helper workgroups included) starts from zeros
        if (do_c && io->hidden_drop && io->hidden_drop[t - step_start])
            TB_HIP(ctx, hipMemsetAsync(p.hidden, 0, sizeof(float) * 3 * (size_t)p.n_inst * p.a_pad * 128, s));
        ctx->launch_kind[i] = (do_c && do_a) ? 2 : ((do_c || do_a) ? 1 : 0);
        if (ctx->timing) TB_HIP(ctx, hipEventRecord(ctx->ev[2 + i], s));
    }
    ctx->n_timed_steps = ctx->timing ? n_launch : 0;
    // ---- final simulator state
    if (io->final_state || io->final_valid || io->final_hidden)
    {
        tb::RolloutP pf = p;
        tb::set_parity(pf, step_start + n_steps);  // validity after the last step = what launch t = step_end wrote
        tb::launch_rollout_final(pf, io->final_state, io->final_valid, io->final_hidden, s);
    }
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- flag-gated traffic-rule checks over a recorded rollout (tb_rules_kernels.hip) ----------------------------------
extern "C" int tb_rule_checks(tb_ctx* ctx, const tb_rule_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_scene <= 0 || io->k_futures <= 0 || io->n_agent <= 0 || io->n_pl <= 0 || io->n_tl <= 0 || io->n_step <= 0)
        return tb_fail(ctx, "tb_rule_checks: empty dimension");
    if (io->n_agent > 256) return tb_fail(ctx, "tb_rule_checks: n_agent %d > 256 not supported", io->n_agent);
    const void* need[] = {io->check_state, io->check_valid, io->agent_type, io->agent_size, io->map_valid, io->map_type, io->map_pos,
                          io->map_dir, io->tl_valid, io->tl_state, io->tl_pos, io->collided, io->collided_this_step,
                          io->run_road_edge, io->run_road_edge_this_step, io->run_red_light, io->run_red_light_this_step,
                          io->passive, io->passive_this_step};
    for (const void* q : need)
        if (!q) return tb_fail(ctx, "tb_rule_checks: a required buffer pointer is NULL");
    const size_t raw_bytes = (size_t)4 * io->n_scene * io->k_futures * io->n_agent * io->n_step;
    if (raw_bytes > ctx->rule_ws_bytes) {
        if (ctx->d_rule_ws) {
            TB_HIP(ctx, hipDeviceSynchronize());
            TB_HIP(ctx, hipFree(ctx->d_rule_ws));
            ctx->d_rule_ws = nullptr;
            ctx->rule_ws_bytes = 0;
        }
        TB_HIP(ctx, hipMalloc((void**)&ctx->d_rule_ws, raw_bytes));
        ctx->rule_ws_bytes = raw_bytes;
    }
    const int n_tl_step = io->n_tl_step > 0 ? io->n_tl_step : ctx->cfg.time_step_current + 1;
    tb::run_rule_checks(io, n_tl_step, ctx->cfg.time_step_sim_start, ctx->d_rule_ws, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- WaymoPostProcessing (tb_post_kernels.hip) ---------------------------------------------------------------------
extern "C" int tb_post_process(tb_ctx* ctx, const tb_post_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_scene <= 0 || io->n_agent <= 0 || io->n_pred <= 0 || io->n_step <= 0 || io->k_pred <= 0)
        return tb_fail(ctx, "tb_post_process: empty dimension");
    if (io->n_pred > 64 || io->k_pred > 16) return tb_fail(ctx, "tb_post_process: n_pred %d > 64 or k_pred %d > 16", io->n_pred, io->k_pred);
    if (io->d_traj < 2 || io->d_traj > 4) return tb_fail(ctx, "tb_post_process: d_traj %d not in 2..4", io->d_traj);
    if ((io->n_mpa != 0 && io->n_mpa != 3) || (io->n_mtr != 0 && io->n_mtr != 3))
        return tb_fail(ctx, "tb_post_process: nms thresholds must be empty or [veh, ped, cyc]");
    const void* need[] = {io->valid, io->scores, io->trajs, io->agent_type, io->waymo_trajs, io->waymo_scores, io->waymo_valid};
    for (const void* q : need)
        if (!q) return tb_fail(ctx, "tb_post_process: a required buffer pointer is NULL");
    tb::launch_post_process(*io, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- ErrorMetrics / TrafficRuleMetrics partial sums (tb_metrics_kernels.hip) -----------------------------------------
extern "C" int tb_metric_partials(tb_ctx* ctx, const tb_metric_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_scene <= 0 || io->n_agent <= 0 || io->k_futures <= 0 || io->n_step <= 0) return tb_fail(ctx, "tb_metric_partials: empty dimension");
    if ((io->gt_valid == nullptr) != (io->gt_states == nullptr)) return tb_fail(ctx, "tb_metric_partials: gt_valid and gt_states go together");
    const void* need[] = {io->pred_valid, io->pred_states, io->override_masks, io->agent_role, io->agent_type, io->outside_map,
                          io->collided, io->run_road_edge, io->run_red_light, io->passive, io->goal_reached, io->dest_reached, io->out};
    for (const void* q : need)
        if (!q) return tb_fail(ctx, "tb_metric_partials: a required buffer pointer is NULL");
    tb::launch_metric_partials(*io, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- the samplers of joint_future_pred (tb_sample_kernels.hip) --------------------------------------------------------
extern "C" int tb_latent_sample(tb_ctx* ctx, const tb_latent_sample_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (!io->log_std && !ctx->finalized) return tb_fail(ctx, "tb_latent_sample: weights not finalized (the log_std vectors are parameters)");
    if (io->n_scene <= 0 || io->n_agent <= 0 || io->k_futures <= 0) return tb_fail(ctx, "tb_latent_sample: empty dimension");
    if (!io->mean) return tb_fail(ctx, "tb_latent_sample: mean is NULL");
    if (!io->sample && !io->log_prob) return tb_fail(ctx, "tb_latent_sample: neither sample nor log_prob requested");
    tb::launch_latent_sample(*io, io->log_std ? io->log_std : ctx->d_arena + (io->posterior ? ctx->ew.post_log_std : ctx->pw.latent_log_std),
                             (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int tb_dest_sample(tb_ctx* ctx, const tb_dest_sample_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_scene <= 0 || io->n_agent <= 0 || io->k_futures <= 0 || io->n_pl <= 0) return tb_fail(ctx, "tb_dest_sample: empty dimension");
    if (!io->dest_logits) return tb_fail(ctx, "tb_dest_sample: dest_logits is NULL");
    if (!io->sample && !io->log_prob && !io->probs) return tb_fail(ctx, "tb_dest_sample: no output requested");
    tb::launch_dest_sample(*io, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- stepwise driving (the reference's stateful WaymoMotion.forward, waymo_motion.py:108-203) ------------------------
extern "C" int tb_rollout_begin(tb_ctx* ctx, const tb_rollout_io* io, tb_stream stream_) {
    tb::RolloutP p;
    if (rollout_setup(ctx, io, p, stream_)) return 1;
    // (tb_rollout-only fields: refused here rather than silently ignored -- ADVICE r03)
    if (io->hidden_drop) return tb_fail(ctx, "tb_rollout_begin: hidden_drop is honoured by tb_rollout only (zero the hidden state between tb_rollout_step calls yourself)");
    hipStream_t s = (hipStream_t)stream_;
    rollout_prologue(ctx, p, io, s);
    step_launch(ctx, p, p, p.step_start - 1, /*do_c=*/0, /*do_a=*/1, s);  // A(sim_start)
    ctx->step_p = p;
    ctx->step_next = p.step_start;
    ctx->step_end = io->step_end;
    ctx->step_active = true;
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int tb_rollout_step_ex(tb_ctx* ctx, const tb_step_override* ov, tb_stream stream_) {
    if (!ctx) return 1;
    if (!ctx->step_active) return tb_fail(ctx, "tb_rollout_step: no rollout in progress (call tb_rollout_begin)");
    if (ctx->step_next > ctx->step_end) return tb_fail(ctx, "tb_rollout_step: step %d is past step_end %d", ctx->step_next, ctx->step_end);
    const int t = ctx->step_next;
    tb::RolloutP p = ctx->step_p;
    if (ov) {
        if ((ov->action == nullptr) != (ov->action_mask == nullptr))
            return tb_fail(ctx, "tb_rollout_step_ex: action and action_mask go together");
        if (!ov->mask && !ov->action_mask) return tb_fail(ctx, "tb_rollout_step_ex: tb_step_override holds neither a state mask nor an action mask");
        if (ov->mask) {
            if (!ov->agent_state || !ov->vel || !ov->acc || !ov->yaw_rate)
                return tb_fail(ctx, "tb_rollout_step_ex: agent_state / vel / acc / yaw_rate must all be given (they are read only where mask is set)");
            p.ovr_mask = ov->mask;
            p.ovr_state = ov->agent_state;
            p.ovr_vel = ov->vel;
            p.ovr_acc = ov->acc;
            p.ovr_yaw_rate = ov->yaw_rate;
            p.ovr_gt_valid = ov->gt_valid;
        } else if (ov->agent_state || ov->vel || ov->acc || ov->yaw_rate || ov->gt_valid) {
            return tb_fail(ctx, "tb_rollout_step_ex: state arrays / gt_valid without tb_step_override.mask");
        }
        p.ovr_action = ov->action;
        p.ovr_action_mask = ov->action_mask;
    }
    step_launch(ctx, p, p, t, /*do_c=*/1, /*do_a=*/t < ctx->step_end, (hipStream_t)stream_);
    tb_note_launch(ctx);
    ctx->step_next = t + 1;
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int tb_rollout_step(tb_ctx* ctx, tb_stream stream_) { return tb_rollout_step_ex(ctx, nullptr, stream_); }

// ---- sticky range flag of the fp16-pair kernels (tb_device_xdl.hpp) -----------------------------------------------
extern "C" int tb_check_status(tb_ctx* ctx, tb_stream stream_) {
    if (!ctx) return 1;
    TB_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream_;
    if (!ctx->d_status) {
        TB_HIP(ctx, hipMalloc((void**)&ctx->d_status, 2 * sizeof(unsigned int)));
        TB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, 2 * sizeof(unsigned int), s));
    }
    TB_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(unsigned int), s));
    tb::xh::launch_range_flag_take_step(ctx->d_status, s);
    tb::xh::launch_range_flag_take_encode(ctx->d_status, s);
    if (!ctx->h_status) TB_HIP(ctx, hipHostMalloc((void**)&ctx->h_status, 2 * sizeof(unsigned int), hipHostMallocDefault));
    TB_HIP(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, s));  // (pinned: truly asynchronous)
    TB_HIP(ctx, hipMemsetAsync(ctx->d_status + 1, 0, sizeof(unsigned int), s));
    TB_HIP(ctx, hipStreamSynchronize(s));
    const unsigned int hw[2] = {ctx->h_status[0], ctx->h_status[1]};
    unsigned int h = hw[0];
    {   // the flag word belongs to the device: what this check took is every context's news, and what another context's check took
        // since this context's last one is this context's news too (conservative: a context may switch kernels for a neighbour's
        // overflow; none returns 0 over invalid results)
        std::lock_guard<std::mutex> lk(g_ctx_mutex);
        // (only contexts whose ACTIVE kernels can raise a bit inherit it -- the step bit goes to contexts on the fp16-pair
        // step kernels, the encoder bit to contexts on the fp16-pair encoders; a bf16 or exact-fp32 neighbour is not re-run, let alone
        // downgraded, for somebody else's overflow)
        auto can_raise = [](const tb_ctx* c) { return (c->step_kernel == 2 ? 1u : 0u) | (c->encode_kernel == 1 ? 2u : 0u); };
        if (h)
            for (tb_ctx* o : g_ctx_live)
                if (o != ctx && o->device == ctx->device) o->range_pending |= h & can_raise(o);
        h = (h | ctx->range_pending) & can_raise(ctx);
        ctx->range_pending = 0;
    }
    if (hw[1])
        return tb_fail(ctx, "step kernel: a tile workgroup gave up waiting for its helper workgroup (the helpers of a launch are expected to be "
                            "dispatched first); results since the last check are invalid -- set TB_STEP_HELPERS=0 and report");
    if (h) {
        // The results of the calls since the last check are invalid, and the context switches ITSELF to the exact-fp32 twins of the
        // kernels that overflowed (fp32's range): the caller re-issues those calls -- the host mirror does it automatically
        // (WaymoMotion(check_range=True)).  Return code 3 tells this case apart from a hard error.
        if ((h & 1u) && ctx->step_kernel == 2) {
            ctx->step_kernel = 0;
            // an open stepwise rollout was begun by the XDL prologue: its hoisted K / V and fusion buffers are in XDL operand order,
            // which the exact kernel cannot read -- the rollout is closed, tb_rollout_step then fails with "call tb_rollout_begin"
            ctx->step_active = false;
        }
        if (h & 2u) ctx->encode_kernel = 0;
        ctx->precision_reason |= 2;
        tb_fail(ctx, "fp16-pair operand range exceeded: a GEMM / attention input of the %s%s%s reached |x| >= 65504 since the last "
                     "check; the results of those calls are invalid.  The context has switched to the exact-fp32 kernels (fp32 MFMA, fp32's range, "
                     "about 2x slower per step): re-issue the calls",
                (h & 1u) ? "step kernels" : "", (h == 3u) ? " and the " : "", (h & 2u) ? "scene encoders" : "");
        ctx->precision_note = ctx->err;
        return 3;
    }
    return 0;
}

extern "C" int tb_rollout_state(tb_ctx* ctx, float* state, uint8_t* valid, float* hidden, tb_stream stream_) {
    if (!ctx) return 1;
    if (!ctx->step_active) return tb_fail(ctx, "tb_rollout_state: no rollout in progress");
    tb::RolloutP pf = ctx->step_p;
    tb::set_parity(pf, ctx->step_next);  // the last launch ran t = step_next - 1 and wrote parity (t + 1) & 1
    tb::launch_rollout_final(pf, state, valid, hidden, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

// which kernels this context runs on, and why (header)
extern "C" int tb_precision_state(tb_ctx* ctx, int32_t* out3) {
    if (!ctx || !out3) return 1;
    out3[0] = ctx->step_kernel == 3 ? 1 : (ctx->step_kernel == 2 ? 0 : 2);
    out3[1] = ctx->encode_kernel == 1 ? 0 : 2;
    out3[2] = ctx->precision_reason;
    return 0;
}
extern "C" const char* tb_precision_note(tb_ctx* ctx) { return ctx ? ctx->precision_note.c_str() : ""; }

// back to the configured kernels after a run-time fallback (header): the arena holds both operand packings of every weight
extern "C" int tb_precision_restore(tb_ctx* ctx, int32_t* out_changed) {
    if (!ctx) return 1;
    if (out_changed) *out_changed = 0;
    if (!(ctx->precision_reason & 2) || (ctx->precision_reason & 1)) return 0;  // (never fell back / the weights themselves are out of range)
    const bool changed = ctx->step_kernel != ctx->step_kernel0 || ctx->encode_kernel != ctx->encode_kernel0;
    ctx->step_kernel = ctx->step_kernel0;
    ctx->encode_kernel = ctx->encode_kernel0;
    ctx->precision_reason &= ~2;
    ctx->precision_note.clear();
    if (changed) ctx->step_active = false;  // (an open stepwise rollout was begun by the other kernels' prologue: operand order differs)
    if (out_changed) *out_changed = changed ? 1 : 0;
    return 0;
}

// development aid (not in the public header): copy the stage time stamps of the last launch of a -DTB_PROFILE build
// hipGraph statistics of the context: out[0] = rollouts captured, out[1] = rollouts replayed from a captured graph
extern "C" int tb_graph_stats(tb_ctx* ctx, int32_t* out2) {
    if (!ctx || !out2) return 1;
    out2[0] = ctx->graph_captures;
    out2[1] = ctx->graph_hits;
    return 0;
}

extern "C" int tb_debug_read_prof(tb_ctx* ctx, long long* host_out, int n_blocks) {
    if (!ctx || !ctx->last_prof) return 1;
    TB_HIP(ctx, hipDeviceSynchronize());
    TB_HIP(ctx, hipMemcpy(host_out, ctx->last_prof, sizeof(long long) * 32 * n_blocks, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int tb_encode_posterior(tb_ctx* ctx, const tb_posterior_io* io, tb_stream stream) {
    if (!ctx || !io) return 1;
    if (!ctx->finalized) return tb_fail(ctx, "tb_encode_posterior: weights not finalized");
    TB_HIP(ctx, hipSetDevice(ctx->device));
    return tb::run_encode_posterior(ctx, io, (hipStream_t)stream);
}

extern "C" int tb_train_partials(tb_ctx* ctx, const tb_train_io* io, tb_stream stream_) {
    if (!ctx || !io) return 1;
    if (!ctx->finalized) return tb_fail(ctx, "tb_train_partials: weights not finalized");
    TB_HIP(ctx, hipSetDevice(ctx->device));
    if (io->n_scene <= 0 || io->n_agent <= 0 || io->n_step <= 0 || io->n_pl <= 0)
        return tb_fail(ctx, "tb_train_partials: empty dimension");
    if (io->crit_pos < 0 || io->crit_pos > 2 || io->crit_rot < 0 || io->crit_rot > 2 || io->crit_spd < 0 || io->crit_spd > 2 ||
        io->angular_type < 0 || io->angular_type > 3)
        return tb_fail(ctx, "tb_train_partials: unknown criterion / angular_type");
    if ((io->gt_valid == nullptr) != (io->gt_states == nullptr)) return tb_fail(ctx, "tb_train_partials: gt_valid / gt_states must come together");
    if ((io->relevant == nullptr) != (io->irrelevant_draw == nullptr))
        return tb_fail(ctx, "tb_train_partials: relevant / irrelevant_draw must come together");
    const void* req[] = {io->pred_valid, io->pred_states, io->override_masks, io->agent_size, io->diffbar_rewards,
                         io->diffbar_rewards_valid, io->out};
    for (const void* q : req)
        if (!q) return tb_fail(ctx, "tb_train_partials: a required buffer pointer is NULL");
    if (io->use_goal && (!io->dest_logits || !io->goal_valid || !io->gt_dest))
        return tb_fail(ctx, "tb_train_partials: use_goal needs dest_logits, goal_valid and gt_dest");
    if (io->use_vae_kl && (!io->post_mean || !io->post_valid || !io->prior_mean || !io->prior_valid))
        return tb_fail(ctx, "tb_train_partials: use_vae_kl needs the posterior and prior personalities");
    tb::launch_train_partials(*io, ctx->d_arena + ctx->ew.post_log_std, ctx->d_arena + ctx->pw.latent_log_std, (hipStream_t)stream_);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int tb_encode_scene(tb_ctx* ctx, const tb_encode_io* io, tb_stream stream) {
    if (!ctx || !io) return 1;
    if (!ctx->finalized) return tb_fail(ctx, "tb_encode_scene: weights not finalized");
    TB_HIP(ctx, hipSetDevice(ctx->device));
    return tb::run_encode(ctx, io, (hipStream_t)stream);
}
