// Internal (not part of the C ABI): context layout shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/trafficbots_hip.h"
#include "tb_encode.hpp"
#include "tb_rollout.hpp"

// The launch-shaping switches as the code reads them: tb_config.sw with the debugging environment variables laid over it
// (tb_switches_now, tb_api.hip).  Resolved once per API call; part of the hipGraph key.
struct TbSw {
    int helpers_off, warm, pre_inter_off, w3_off, aw, lean_off, graph_off, enc_pack, enc_side_off, enc_dest_side_off, dest_lds_pad;
};

struct tb_ctx {
    tb_config cfg;
    int device = 0;  // HIP device the context is bound to
    std::string err;
    std::map<std::string, std::vector<float>> staged;
    float* d_arena = nullptr;
    size_t arena_floats = 0;
    tb::PolicyW pw;
    tb::PolicyWX px;   // fp16-pair packings
    tb::PolicyWX pxb;  // bf16 packings (operand_precision = 1)
    tb::EncoderW ew;
    bool finalized = false;
    int step_kernel = 2;  // 2: k_step_x (fp16-pair XDL MFMA, default)  3: k_step_x, bf16 operands (tb_config.operand_precision = 1)  0: k_step (fp32 MFMA, the A/B twin); TB_STEP_KERNEL overrides
    int step_kernel0 = 2, encode_kernel0 = 1;  // the selection of tb_create (configuration + development switches): what tb_finalize_weights starts from
    int precision_reason = 0;    // bit 0: a loaded tensor is outside the fp16-pair range; bit 1: an activation overflowed at run time
    unsigned int range_pending = 0;  // range-flag bits another context of this device took in ITS check (the flag is one word per device and kernel family): reported by this context's next check
    std::string precision_note;  // the sentence that goes with it (tb_precision_note)
    int encode_kernel = 1;  // 1: XDL attention blocks (tb_encodex_kernels.hip, default)  0: fp32-MFMA blocks; TB_ENCODE_KERNEL overrides
    // workspace
    char* d_ws = nullptr;      // the rollout's carves (tb_rollout / tb_rollout_begin)
    size_t ws_bytes = 0;
    char* d_ws_enc = nullptr;  // the encoders' carves (tb_encode_scene / tb_encode_posterior): a workspace of their own, so that the
    size_t ws_enc_bytes = 0;   // encoders of batch n + 1 may run on another stream beside the rollout of batch n (round 6)
    uint8_t* d_rule_ws = nullptr;  // per-step flags of tb_rule_checks
    size_t rule_ws_bytes = 0;
    long long* last_prof = nullptr;
    unsigned int* d_status = nullptr;  // device word of tb_check_status
    unsigned int* h_status = nullptr;  // its pinned host mirror (hipHostMalloc): the check's read-back is an async copy + one stream synchronise
    // L2 warmers (tb_stepx_kernels.hip): [64][2] {arena offset, request time} of the weight units a launch streams -- one device
    // table per (p_pad, a_pad, step kernel), uploaded once on the caller's stream and never rewritten (launches in flight and
    // captured graphs keep reading theirs: ADVICE r05)
    struct WarmTab { int* d = nullptr; int n = 0; std::vector<int> host; };
    std::map<long long, WarmTab> warm_tabs;
    long long last_launch_ns = 0;      // steady-clock time of this context's last rollout / encode call (the warmers' auto-off)
    // stepwise rollout (tb_rollout_begin / _step / _state)
    tb::RolloutP step_p;
    int step_next = 0, step_end = -1;
    bool step_active = false;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    int n_timed_steps = 0;
    std::vector<uint8_t> launch_kind;  // per step launch of the last tb_rollout: 2 fused, 1 one half, 0 skipped (tb_get_timing)
    int n_pre = 0;                     // batched warm-start slices of the rollout being set up
    // one hipGraph per rollout (tb_rollout): the S + 1 step launches + prologue captured once and replayed while every kernel
    // argument stays the same (same buffers, sizes, switches) -- the launching thread then spends microseconds per rollout instead of
    // ~70 us per launch; TB_ROLLOUT_GRAPH=0 turns it off
    // row-major fp32 copies of the policy-trunk tensors for tb_forward (the un-fused visualisation path), and its scratch
    float* d_raw = nullptr;
    std::map<std::string, const float*> raw;
    float* d_fw = nullptr;
    size_t fw_floats = 0;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t cap_stream = nullptr;  // the context's private stream: the launch sequence is captured on it (the caller's may be the legacy default stream, which cannot capture), and tb_encode_scene forks its side work to it
    // full argument bytes (RolloutP, tb_rollout_io, hidden_drop, the launch-shaping switches) of the captured graph / of the previous
    // call (capture on the second sight) / of an argument set whose capture failed (never tried again): compared byte for byte, a hash
    // alone could replay a graph bound to other buffers
    std::vector<unsigned char> graph_key, graph_seen, graph_nocapture;
    int graph_hits = 0, graph_captures = 0;
    // tb_encode_scene: the agent / traffic-light token encoders and the destination predictor's GRU scan (128 workgroups: half the
    // chip, and independent of the map) run on a side stream beside the map encoder's chip-filling launches; forked from and joined
    // back into the caller's stream with events (TB_ENCODE_SIDE=0: everything on the caller's stream).  2.60 -> 2.40 ms per 32 scenes.
    // (A second fork -- the destination logits beside the personality branch -- gained nothing: 2.42 .. 2.51 ms.)
    hipEvent_t enc_fork = nullptr, enc_join = nullptr, enc_map = nullptr, enc_join2 = nullptr;
};


inline int tb_fail(tb_ctx* ctx, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return 1;
}

#define TB_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return tb_fail(ctx, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Debug facility (TB_WS_GUARD=1 | 2; tests/probes/gpu_guard_pages.py): every carve of the workspace becomes a mapping of its own with
// UNMAPPED address space on both sides (HIP virtual-memory calls), the buffer at the end (1) or at the start (2) of its mapping -- a
// kernel that runs over the end of one internal buffer into the next (invisible inside one allocation) takes a GPU memory fault.
// The mappings live until the process ends; nothing for production.  Returns nullptr when the mode is off.
void* tb_ws_guard_take(size_t bytes);

// bump allocator over the context workspace (pass base = nullptr to size)
struct Carver {
    char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        if (base)
            if (void* g = tb_ws_guard_take(n * sizeof(T))) return reinterpret_cast<T*>(g);
        return p;
    }
};

TbSw tb_switches_now(const tb_ctx* ctx);
void tb_note_launch(tb_ctx* ctx);
int tb_ensure_workspace(tb_ctx* ctx, size_t bytes);
int tb_ensure_workspace_enc(tb_ctx* ctx, size_t bytes);
inline int padk(int x) { return (x + 31) / 32 * 32; }  // key counts / row counts are padded to 32 (tb::KEYPAD)
