// k_step_x8: the fused step launch C(t) + A(t+1) of tb_stepx_kernels.hip on 512-thread workgroups -- 8 waves, TWO per SIMD
// (tb_device_xdl8.hpp).  Same tiling (one workgroup per 16 agents), same global layouts, same per-agent arithmetic; what changes
// is the split of a tile's work over waves: one 16-feature output tile per wave in every Linear, one head per wave PAIR in the
// attention (every second key block each, merged through LDS).  The stages bound by a single wave's dependent instruction chain
// (attention, LayerNorm, epilogues) get a second wave per SIMD to fill their bubbles; the weight-streaming Linears are bound by
// the CU's L2 -> L1 path either way (tools/microtests/attn_loop.hip, gemm_chain.hip).
// The rollout prologue kernels (k_kv_hoist_x, k_fuse_hoist_x, k_pre_replicate) are those of tb_stepx_kernels.hip.
#include "tb_rollout.hpp"
#include "tb_device_xdl8.hpp"
#include "tb_step_common.hpp"

namespace tb {
namespace TB_XNS {

// LDS carve (floats): six fp32 tiles, the Q exchange area (in the slot the 4-wave kernel leaves unused), geometry, LN parameters,
// small state, encoder weights, four plane buffers, the attention merge areas
constexpr int YO_X = 0;
constexpr int YO_H = YO_X + TM * LDT;
constexpr int YO_H1 = YO_H + TM * LDT;
constexpr int YO_H2 = YO_H1 + TM * LDT;
constexpr int YO_GP = YO_H2 + TM * LDT;
constexpr int YO_LP = YO_GP + TM * LDT;
constexpr int YO_XQ = YO_LP + TM * LDT;
constexpr int YO_DG = YO_XQ + XQ8_FLOATS;
constexpr int YO_LN = YO_DG + TM * 80;
constexpr int YO_SMALL = YO_LN + 9 * 768;
constexpr int YO_ENCW = YO_SMALL + SMALL_FLOATS;
constexpr int YO_PL = YO_ENCW + ENCW_FLOATS;  // 4 x [NPL][16][LDP] halfs
constexpr int PLANES_FLOATS8 = PLANES_BYTES / 4;
constexpr int YO_XO = YO_PL + 4 * PLANES_FLOATS8;
constexpr int YO_XS = YO_XO + XO8_FLOATS;
constexpr int STEPX8_LDS_FLOATS = YO_XS + XS8_FLOATS;
static_assert(YO_PL % 4 == 0 && YO_XO % 4 == 0 && YO_XQ % 4 == 0, "16-byte alignment");
static_assert(STEPX8_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ WNext1 q_first8(const float* W, const XLayerW& L, const XLayerX& LX, int wave) { return wnext1(W, LX.wq, W + L.bq, wave); }
__device__ __forceinline__ WNext1 gru_first8(const float* W, const GruLayerW& G, const GruLayerX& GX, int wave) {
    return wnext1(W, GX.wih, W + G.bih, wave);
}

// add_goal / add_latent fusion MLP with the constant half hoisted (see fuse_latent_goal_x): h = relu(W2 relu(W1[:, :128] x + PRE) + b2)
//   PX : planes of x;  P2 : plane buffer for the hidden;  uw : in = the x half of W1, tile `wave` (carries b1)
__device__ __forceinline__ void fuse_latent_goal8(const float* __restrict__ W, uint32_t w2x, uint32_t b2, float* X, xhalf* PX, xhalf* P2,
                                                  const float* PRE, const uint8_t* zvalid, const uint8_t* rowvalid, int tid, WUnit1& uw,
                                                  const WNext1& nxt) {
    const int wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    tile_to_planes8(X, LDT, PX, tid);
    __syncthreads();
    WUnit1 u2;
    const bool zv = zvalid[m] != 0;
    {
        f32x4 acc = uw.b;
        wmma1_pf(acc, uw, PX + m * LDP + kq * 8, PLANE, u2, wnext1(W, w2x, W + b2, wave), lane);
        if (zv) acc += lds4(cptr(const_cast<float*>(PRE), LDT, wave, lane));
        planes_store_c(P2, wave, lane, relu4(acc));
    }
    __syncthreads();
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, P2 + m * LDP + kq * 8, PLANE, uw, nxt, lane);
        const bool rv = rowvalid[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 h = zv ? relu4(acc) : splat(0.f);
        st4(px, rv ? h + lds4(px) : splat(0.f));
    }
    __syncthreads();
}

// PRE = the batched warm start (RolloutP::pre_mode): A half only, inputs from the ground truth, grid.z = steps
template <bool PRE>
__global__ __launch_bounds__(NT8) void k_step_x8(RolloutP p, int t, int do_c, int do_a) {
    if (PRE) {
        do_c = 0;
        do_a = 1;
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + YO_X;
    float* Hs = smem + YO_H;
    float* H1 = smem + YO_H1;
    float* H2 = smem + YO_H2;
    float* GP = smem + YO_GP;
    float* LP = smem + YO_LP;
    float* DG = smem + YO_DG;
    float* LN = smem + YO_LN;
    float* ENCW = smem + YO_ENCW;
    xhalf* PA = reinterpret_cast<xhalf*>(smem + YO_PL);
    xhalf* PB = PA + NPL * PLANE;
    xhalf* PC = PB + NPL * PLANE;
    xhalf* PD = PC + NPL * PLANE;
    const Xch8 xc{smem + YO_XQ, smem + YO_XO, smem + YO_XS};
    const StepSmall sm = step_small(smem + YO_SMALL);
    RowSt* rst = sm.rst;
    float* ubuf = sm.ubuf;
    uint8_t* rowvalid = sm.rowvalid;
    uint8_t* novalid_s = sm.novalid_s;
    uint8_t* gvalid = sm.gvalid;
    int* rtype = sm.rtype;
    int* dflag = sm.dflag;

    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    int n, rt;
    step_tile_map(n, rt);
    if (PRE) t = p.pre_t0 + (int)blockIdx.z;
    const int b = PRE ? n : n / p.k_rep;
    if (PRE) n = b * p.k_rep;
    const int row0 = rt * TM;
    const int n_real = max(0, min(TM, p.n_agent - row0));
    const float* W = p.W;
    const PolicyW& pw = p.pw;
    const PolicyWX& px = p.px;
    const size_t base_row = (size_t)n * p.a_pad + row0;
    const int n_rt = gridDim.x;

    WUnit1 u;
    TB_STAMP(0);
    // ---- launch start: every load of the prologue is issued before the first result is consumed (one cold round trip)
    const uint32_t lnbase[9] = {pw.inter[0].ln1_g, pw.inter[1].ln1_g, pw.inter[2].ln1_g, pw.as2pl[0].ln1_g, pw.as2pl[1].ln1_g,
                                pw.as2pl[2].ln1_g, pw.as2tl[0].ln1_g, pw.as2tl[1].ln1_g, pw.as2tl[2].ln1_g};
    f32x4 lnv[9], rs_st = splat(0.f), rs_ax = splat(0.f);
    int rs_ty = -1;
    uint8_t rs_v = 0, rs_g = 0, vb[4] = {0, 0, 0, 0};
    CInputs<NT8> cin;
    EncWRegs encw;
    TB_SCHED_FENCE();
    wload1(u, do_c ? q_first8(W, pw.inter[0], px.inter[0], wave) : q_first8(W, pw.as2pl[0], px.as2pl[0], wave), lane);
    if (tid < TM) {
        const size_t si = base_row + tid;
        rs_ty = (tid < n_real) ? p.agent_type[(size_t)b * p.n_agent + row0 + tid] : -1;
        if (PRE) {
            if (tid < n_real) {
                const size_t hi = ((size_t)b * p.n_hist + t) * p.n_agent + row0 + tid;
                rs_st = ldg4(p.hist_state + hi * 4);
                rs_ax = f32x4{p.hist_vel[hi * 2], p.hist_vel[hi * 2 + 1], p.hist_acc[hi], p.hist_yaw_rate[hi]};
                rs_v = p.hist_valid[hi];
            }
        } else {
            rs_st = ldg4(p.state + si * 4);
            rs_ax = ldg4(p.aux + si * 4);
            rs_v = p.valid[si];
            rs_g = p.goal_valid[si];
        }
    }
    if (do_c) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i * 64 + lane < p.a_pad) vb[i] = p.valid[(size_t)n * p.a_pad + i * 64 + lane];  // (a_pad <= 256)
        c_inputs_issue<NT8>(p, n, row0, tid, cin);
    }
#pragma unroll
    for (int sl = 0; sl < 9; ++sl)
        if (tid < 192) lnv[sl] = ldg4(W + lnbase[sl] + tid * 4);
    if (do_a && tid < 256) encw_issue(pw, W, tid, encw);
    TB_SCHED_FENCE();
    if (do_a && tid < 256) encw_commit(tid, encw, ENCW);
    if (tid < TM) {
        rtype[tid] = rs_ty;
        rst[tid].st[0] = rs_st.x; rst[tid].st[1] = rs_st.y; rst[tid].st[2] = rs_st.z; rst[tid].st[3] = rs_st.w;
        rst[tid].aux[0] = rs_ax.x; rst[tid].aux[1] = rs_ax.y; rst[tid].aux[2] = rs_ax.z; rst[tid].aux[3] = rs_ax.w;
        rowvalid[tid] = rs_v;
        gvalid[tid] = rs_g;
    }

    if (do_c) {
        // =================================== C(t) ===================================
        int n_valid = 0, hi_valid = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long vm = __ballot(vb[i] != 0);
            n_valid += __popcll(vm);
            if (vm) hi_valid = i * 64 + 64 - __clzll(vm);
        }
        const bool bypass = n_valid == 1;  // agent_interaction.py:61
        const int nk_a = min(p.a_pad, max(32, (hi_valid + 31) & ~31));
        const int ks_a = ((rt * (nk_a >> 5)) / n_rt) << 5;
        c_inputs_commit<NT8>(tid, cin, X, Hs, H1, H2, GP, LP, DG, dflag);
#pragma unroll
        for (int sl = 0; sl < 9; ++sl)
            if (tid < 192) st4(LN + sl * 768 + tid * 4, lnv[sl]);
        if (bypass) wload1(u, gru_first8(W, pw.gru[0], px.gru[0], wave), lane);
        __syncthreads();
        TB_STAMP(1);
        if (!bypass) {
            const float* kvd = p.vbias + (size_t)n * p.a_pad;
            const size_t ls = (size_t)p.a_pad * H;
            const xhalf* K0 = reinterpret_cast<const xhalf*>(p.kin + ((size_t)n * 3) * ls);
            const xhalf* V0 = reinterpret_cast<const xhalf*>(p.vtin + ((size_t)n * 3) * ls);
            xattn_layer8<true, true>(W, pw.inter[0], px.inter[0], X, PA, PB, xc, K0, V0, kvd, nk_a, ks_a, row0, rowvalid, novalid_s, tid, u,
                                     q_first8(W, pw.inter[1], px.inter[1], wave), LN + 0 * 768);
            xattn_layer8<true, true>(W, pw.inter[1], px.inter[1], X, PA, PB, xc, K0 + 2 * ls, V0 + 2 * ls, kvd, nk_a, ks_a, row0, rowvalid, novalid_s,
                                     tid, u, q_first8(W, pw.inter[2], px.inter[2], wave), LN + 1 * 768);
            xattn_layer8<true, true>(W, pw.inter[2], px.inter[2], X, PA, PB, xc, K0 + 4 * ls, V0 + 4 * ls, kvd, nk_a, ks_a, row0, rowvalid, novalid_s,
                                     tid, u, gru_first8(W, pw.gru[0], px.gru[0], wave), LN + 2 * 768);
        }
        TB_STAMP(2);
        // ---- 3-layer GRU, one step.  planes: x0 = PA, h0 = PB, h1 = PD, out0 = PC, h2 -> PB, out1 = PA
        {
            float* hg0 = p.hidden + (((size_t)0 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg1 = p.hidden + (((size_t)1 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg2 = p.hidden + (((size_t)2 * p.n_inst + n) * p.a_pad + row0) * H;
            tile_to_planes8(X, LDT, PA, tid);
            tile_to_planes8(Hs, LDT, PB, tid);
            tile_to_planes8(H1, LDT, PD, tid);
            __syncthreads();
            gru_layer8(W, pw.gru[0], px.gru[0], PA, PB, Hs, PC, nullptr, rowvalid, hg0, TM, tid, u, gru_first8(W, pw.gru[1], px.gru[1], wave));
            tile_to_planes8(H2, LDT, PB, tid);  // (h0's planes are free after the barrier that closed layer 0)
            gru_layer8(W, pw.gru[1], px.gru[1], PC, PD, H1, PA, nullptr, rowvalid, hg1, TM, tid, u, gru_first8(W, pw.gru[2], px.gru[2], wave));
            gru_layer8(W, pw.gru[2], px.gru[2], PA, PB, H2, nullptr, X, rowvalid, hg2, TM, tid, u,
                       wnext1(W, px.goal_out_w1, W + pw.goal_out_b1, wave, 8, 0));
        }
        TB_STAMP(3);
        // ---- add_goal, add_latent (traffic_bots.py:240-241); x planes = PC, hidden = PB
        fuse_latent_goal8(W, px.goal_out_w2, pw.goal_out_b2, X, PC, PB, GP, gvalid, rowvalid, tid, u,
                          wnext1(W, px.lat_out_w1, W + pw.lat_out_b1, wave, 8, 0));
        const int my_ty = (lane < TM && rowvalid[lane]) ? rtype[lane] : -1;
        const bool has0 = __ballot(my_ty == 0) != 0, has1 = __ballot(my_ty == 1) != 0, has2 = __ballot(my_ty == 2) != 0;
        const WNext1 after_head = do_a ? q_first8(W, pw.as2pl[0], px.as2pl[0], wave) : wnext1(W, px.head_w1[0], W + pw.head_b1[0], wave);
        const WNext1 h2 = has2 ? wnext1(W, px.head_w1[2], W + pw.head_b1[2], wave) : after_head;
        const WNext1 h1 = has1 ? wnext1(W, px.head_w1[1], W + pw.head_b1[1], wave) : h2;
        const WNext1 h0 = has0 ? wnext1(W, px.head_w1[0], W + pw.head_b1[0], wave) : h1;
        TB_STAMP(4);
        fuse_latent_goal8(W, px.lat_out_w2, pw.lat_out_b2, X, PC, PB, LP, rowvalid, rowvalid, tid, u, h0);
        TB_STAMP(5);
        if (t == p.tap_step && p.tap_policy_feature)
            store_tile8(p.tap_policy_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

        // ---- action head (action_head.py:69-75): first Linear of every type present, hidden tiles -> Hs / H1 / H2, then ONE
        // reduction stage for the 128 -> 2 Linear of each row's own type (256 threads)
        tile_to_planes8(X, LDT, PA, tid);
        if (tid < 32) ubuf[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const bool present = ty == 0 ? has0 : (ty == 1 ? has1 : has2);
            if (!present) continue;
            WUnit1 uh = u;
            f32x4 acc = uh.b;
            wmma1_pf(acc, uh, PA + m * LDP + kq * 8, PLANE, u, ty == 0 ? h1 : (ty == 1 ? h2 : after_head), lane);
            float* hb = ty == 0 ? Hs : (ty == 1 ? H1 : H2);
            st4(cptr(hb, LDT, wave, lane), relu4(acc));
        }
        __syncthreads();
        if (tid < 256) {
            const int pair = tid >> 3, sub = tid & 7, r = pair >> 1, o = pair & 1;
            const int ty = rtype[r];
            const bool use = ty >= 0 && rowvalid[r];
            const int tyc = ty < 0 ? 0 : ty;
            const float* hb = tyc == 0 ? Hs : (tyc == 1 ? H1 : H2);
            const uint32_t w2o = tyc == 0 ? pw.head_w2[0] : (tyc == 1 ? pw.head_w2[1] : pw.head_w2[2]);
            const uint32_t b2o = tyc == 0 ? pw.head_b2[0] : (tyc == 1 ? pw.head_b2[1] : pw.head_b2[2]);
            const float* w2 = W + w2o + o * H + sub * 16;
            const float* xs = hb + r * LDT + sub * 16;
            float sacc = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x4 a4 = lds4(xs + 4 * k4), w4 = ldg4(w2 + 4 * k4);
                sacc = fmaf(a4.x, w4.x, sacc); sacc = fmaf(a4.y, w4.y, sacc);
                sacc = fmaf(a4.z, w4.z, sacc); sacc = fmaf(a4.w, w4.w, sacc);
            }
            sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0xB1, 0xf, 0xf, true));
            sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x4E, 0xf, 0xf, true));
            sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x141, 0xf, 0xf, true));
            if (sub == 0 && use) ubuf[pair] = sacc + W[b2o + o];
        }
        __syncthreads();
        TB_STAMP(6);
        if (tid < 256) step_epilogue16(p, t, n, b, row0, n_real, tid, sm, DG);
        __syncthreads();
    } else {
#pragma unroll
        for (int sl = 0; sl < 9; ++sl)
            if (tid < 192) st4(LN + sl * 768 + tid * 4, lnv[sl]);
        __syncthreads();
    }
    TB_STAMP(7);
    if (!do_a) return;

    // =================================== A(t+1) ===================================
    const int t1 = t + 1;
    TB_STAMP(30);
    step_encode_inputs_lds<NT8>(p, b, row0, n_real, tid, sm, ENCW, X);
    if (t1 == p.tap_step && p.tap_agent_feature)
        for (int k = 0; k < (PRE ? p.k_rep : 1); ++k)
            store_tile8(p.tap_agent_feature + ((size_t)(n + k) * p.n_agent + row0) * H, X, LDT, n_real, tid);
    TB_STAMP(8);
    const int g_tl = b * p.n_tl_hist + min(t1 - 1, p.n_tl_hist - 1);
    const int nk_t_raw = p.nkey_tl[g_tl];
    const bool tl_empty = nk_t_raw == 0;
    {
        const float* kvd = p.kbias_pl + (size_t)b * p.p_pad;
        const int nk_p = max(32, p.nkey_pl[b]);
        const int ks_p = ((rt * (nk_p >> 5)) / n_rt) << 5;
        const size_t ls = (size_t)p.p_pad * H;
        const xhalf* K0 = reinterpret_cast<const xhalf*>(p.kpl + ((size_t)b * 3) * ls);
        const xhalf* V0 = reinterpret_cast<const xhalf*>(p.vtpl + ((size_t)b * 3) * ls);
        xattn_layer8<true>(W, pw.as2pl[0], px.as2pl[0], X, PA, PB, xc, K0, V0, kvd, nk_p, ks_p, -1, rowvalid, novalid_s, tid, u,
                           q_first8(W, pw.as2pl[1], px.as2pl[1], wave), LN + 3 * 768,
                           p.prof + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32);
        xattn_layer8<true>(W, pw.as2pl[1], px.as2pl[1], X, PA, PB, xc, K0 + 2 * ls, V0 + 2 * ls, kvd, nk_p, ks_p, -1, rowvalid, novalid_s, tid, u,
                           q_first8(W, pw.as2pl[2], px.as2pl[2], wave), LN + 4 * 768);
        xattn_layer8<true>(W, pw.as2pl[2], px.as2pl[2], X, PA, PB, xc, K0 + 4 * ls, V0 + 4 * ls, kvd, nk_p, ks_p, -1, rowvalid, novalid_s, tid, u,
                           q_first8(W, pw.as2tl[0], px.as2tl[0], wave), LN + 5 * 768);
    }
    TB_STAMP(9);
    const WNext1 kv_first = wnext1(W, px.inter_kvf[0], W + px.inter_bkvf[0], wave);
    if (tl_empty) {
        // (the Q unit requested above is dropped; one exposed unit load here keeps the common path free of any select)
        wload1(u, wnext1(W, px.as2tl[0].w1, W + pw.as2tl[0].b1, wave), lane);
        ffn_layer8<true>(W, pw.as2tl[0], px.as2tl[0], X, PA, PB, rowvalid, tid, u, wnext1(W, px.as2tl[1].w1, W + pw.as2tl[1].b1, wave), LN + 6 * 768);
        ffn_layer8<true>(W, pw.as2tl[1], px.as2tl[1], X, PA, PB, rowvalid, tid, u, wnext1(W, px.as2tl[2].w1, W + pw.as2tl[2].b1, wave), LN + 7 * 768);
        ffn_layer8<true>(W, pw.as2tl[2], px.as2tl[2], X, PA, PB, rowvalid, tid, u, kv_first, LN + 8 * 768);
    } else {
        const float* kvd = p.kbias_tl + (size_t)g_tl * p.t_pad;
        const int nk_t = nk_t_raw;
        const int ks_t = ((rt * (nk_t >> 5)) / n_rt) << 5;
        const size_t ls = (size_t)p.t_pad * H;
        const xhalf* K0 = reinterpret_cast<const xhalf*>(p.ktl + ((size_t)g_tl * 3) * ls);
        const xhalf* V0 = reinterpret_cast<const xhalf*>(p.vttl + ((size_t)g_tl * 3) * ls);
        xattn_layer8<true>(W, pw.as2tl[0], px.as2tl[0], X, PA, PB, xc, K0, V0, kvd, nk_t, ks_t, -1, rowvalid, novalid_s, tid, u,
                           q_first8(W, pw.as2tl[1], px.as2tl[1], wave), LN + 6 * 768);
        xattn_layer8<true>(W, pw.as2tl[1], px.as2tl[1], X, PA, PB, xc, K0 + 2 * ls, V0 + 2 * ls, kvd, nk_t, ks_t, -1, rowvalid, novalid_s, tid, u,
                           q_first8(W, pw.as2tl[2], px.as2tl[2], wave), LN + 7 * 768);
        xattn_layer8<true>(W, pw.as2tl[2], px.as2tl[2], X, PA, PB, xc, K0 + 4 * ls, V0 + 4 * ls, kvd, nk_t, ks_t, -1, rowvalid, novalid_s, tid, u,
                           kv_first, LN + 8 * 768);
    }
    TB_STAMP(10);
    const size_t zslice = PRE ? (size_t)blockIdx.z * p.n_inst * p.a_pad * H : 0;
    store_tile8((PRE ? p.x_mid_pre + zslice : p.x_mid_w) + base_row * H, X, LDT, TM, tid);
    {
        const size_t ls = (size_t)p.a_pad * H;
        xhalf* K0 = reinterpret_cast<xhalf*>((PRE ? p.kin_pre + 3 * zslice : p.kin_w) + ((size_t)n * 3) * ls);
        xhalf* V0 = reinterpret_cast<xhalf*>((PRE ? p.vtin_pre + 3 * zslice : p.vtin_w) + ((size_t)n * 3) * ls);
        kv_project_shared8(W, px.inter_kvf, px.inter_bkvf, X, PA, K0, V0, ls, row0, TM, tid, u, kv_first);
    }
    TB_STAMP(11);
}

template __global__ void k_step_x8<false>(RolloutP, int, int, int);
template __global__ void k_step_x8<true>(RolloutP, int, int, int);

// fp16-pair range flag of THIS translation unit (tb_device_xdl.hpp): OR it into *out and clear it (tb_check_status)
#ifndef TB_XDL_BF16
__global__ void k_range_flag_take_step8(unsigned int* out) {
    const unsigned int f = atomicExch(&g_range_flag, 0u);
    if (f) atomicOr(out, 1u);
}
void launch_range_flag_take_step8(unsigned int* out, hipStream_t s) { hipLaunchKernelGGL(k_range_flag_take_step8, dim3(1), dim3(1), 0, s, out); }
#endif

hipError_t configure_stepx8_kernel() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_x8<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(STEPX8_LDS_FLOATS * sizeof(float)));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_x8<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(STEPX8_LDS_FLOATS * sizeof(float)));
}

void launch_step_x8(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_step_x8<false>, grid, dim3(NT8), STEPX8_LDS_FLOATS * sizeof(float), s, p, t, do_c, do_a);
}

void launch_pre_replicate(const RolloutP& p, int n, hipStream_t s);  // (tb_stepx_kernels.hip of the same precision)

// A halves of steps t0 + 1 .. t0 + n from the ground truth of steps t0 .. t0 + n - 1, one launch (RolloutP::pre_mode)
void launch_step_pre_x8(const RolloutP& p0, int t0, int n, hipStream_t s) {
    RolloutP p = p0;
    p.pre_mode = 1;
    p.pre_t0 = t0;
    dim3 grid(p.a_pad / TM, p.n_scene, n);
    hipLaunchKernelGGL(k_step_x8<true>, grid, dim3(NT8), STEPX8_LDS_FLOATS * sizeof(float), s, p, t0, 0, 1);
    if (p.k_rep > 1) launch_pre_replicate(p, n, s);
}

}  // namespace TB_XNS
}  // namespace tb
