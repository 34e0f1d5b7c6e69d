// k_step8: the fused step launch C(t) + A(t+1) of tb_rollout_kernels.hip on 512-thread workgroups (8 waves, two per
// SIMD).  Same tiles, same LDS carve, same global layout and the same per-agent arithmetic as k_step; what changes is
// the split of a tile's work over waves (tb_device8.hpp): one 16-feature output tile per wave in every Linear, one
// attention head per wave pair.  The two waves of a SIMD interleave, so softmax / LayerNorm / barrier bubbles of one
// are filled with MFMAs of the other.
#include "tb_device8.hpp"
#include "tb_step_common.hpp"

namespace tb {

// add_goal / add_latent fusion MLP (add_latent_goal.py:57-77), 8 waves; see fuse_latent_goal
__device__ __forceinline__ void fuse_latent_goal8(const float* __restrict__ W, uint32_t w1, uint32_t w2, uint32_t b2, float* X, float* CAT,
                                                  float* S2, const float* PRE, const uint8_t* zvalid, const uint8_t* rowvalid, int tid,
                                                  WUnit1& uw, const WNext1& nxt) {
    const int wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    {
        const int r = tid >> 5, c4 = (tid & 31) * 4;  // 16 rows x 32 float4 = 512 threads
        st4(CAT + r * LDC + c4, lds4(X + r * LDT + c4));
        st4(CAT + r * LDC + 128 + c4, zvalid[r] ? relu4(lds4(PRE + r * LDT + c4)) : splat(0.f));
    }
    __syncthreads();
    WUnit1 u2;
    {
        f32x4 acc = uw.b;
        const float* xr = CAT + m * LDC + kq * 64;
        wmma1_pf(acc, uw, xr, u2, wnext1(W + w1, nullptr, wave, 16, 8), lane);
        wmma1_pf(acc, u2, xr + 32, uw, wnext1(W + w2, W + b2, wave), lane);
        st4(cptr(S2, LDT, wave, lane), relu4(acc));
    }
    __syncthreads();
    {
        f32x4 acc = uw.b;
        wmma1_pf(acc, uw, S2 + m * LDT + kq * 32, u2, nxt, lane);
        const bool zv = zvalid[m] != 0, rv = rowvalid[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 h = zv ? relu4(acc) : splat(0.f);
        st4(px, rv ? h + lds4(px) : splat(0.f));
        uw = u2;
    }
    __syncthreads();
}

__global__ __launch_bounds__(NTHREADS8) void k_step8(RolloutP p, int t, int do_c, int do_a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + OFF_X;
    float* S1 = smem + OFF_S1;
    float* S2 = smem + OFF_S2;
    float* Hs = smem + OFF_H;
    float* Y = smem + OFF_Y;
    float* CAT = smem + OFF_CAT;
    float* H1 = smem + OFF_H1;
    float* H2 = smem + OFF_H2;
    float* GP = smem + OFF_GP;
    float* LP = smem + OFF_LP;
    float* DG = smem + OFF_DG;
    float* LN = smem + OFF_LN;
    const StepSmall sm = step_small(smem + OFF_SMALL);
    RowSt* rst = sm.rst;
    float* ubuf = sm.ubuf;
    uint8_t* rowvalid = sm.rowvalid;
    uint8_t* novalid_s = sm.novalid_s;
    uint8_t* gvalid = sm.gvalid;
    int* rtype = sm.rtype;
    int* dflag = sm.dflag;
    // attention scratch aliases buffers that are idle while a cross-attention layer runs (Y: GRU only, CAT: fusion only)
    const X8Scratch xs{S1, S2, Y, CAT, CAT + TM * LDT};
    static_assert(TM * LDT + 2 * 4 * 16 * 2 <= TM * LDC, "attention merge scratch must fit the concat tile");

    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    int n, rt;
    step_tile_map(n, rt);
    const int b = n / p.k_rep, row0 = rt * TM;
    const int n_real = max(0, min(TM, p.n_agent - row0));
    const float* W = p.W;
    const PolicyW& pw = p.pw;
    const size_t base_row = (size_t)n * p.a_pad + row0;

    WUnit1 u;
    TB_STAMP(0);
    {
        // LayerNorm parameter blocks -> LDS (slots 0..2 interaction, 3..5 as2pl, 6..8 as2tl)
        const uint32_t base[9] = {pw.inter[0].ln1_g, pw.inter[1].ln1_g, pw.inter[2].ln1_g, pw.as2pl[0].ln1_g, pw.as2pl[1].ln1_g,
                                  pw.as2pl[2].ln1_g, pw.as2tl[0].ln1_g, pw.as2tl[1].ln1_g, pw.as2tl[2].ln1_g};
#pragma unroll
        for (int sl = 0; sl < 9; ++sl)
            if (tid < 192) st4(LN + sl * 768 + tid * 4, ldg4(W + base[sl] + tid * 4));
    }
    if (tid < TM) {
        rtype[tid] = (tid < n_real) ? p.agent_type[(size_t)b * p.n_agent + row0 + tid] : -1;
        const size_t si = base_row + tid;
        const f32x4 st = ldg4(p.state + si * 4), ax = ldg4(p.aux + si * 4);
        rst[tid].st[0] = st.x; rst[tid].st[1] = st.y; rst[tid].st[2] = st.z; rst[tid].st[3] = st.w;
        rst[tid].aux[0] = ax.x; rst[tid].aux[1] = ax.y; rst[tid].aux[2] = ax.z; rst[tid].aux[3] = ax.w;
        rowvalid[tid] = p.valid[si];
        gvalid[tid] = p.goal_valid[si];
    }

    if (do_c) {
        // =================================== C(t) ===================================
        int n_valid = 0;
        for (int i0 = 0; i0 < p.a_pad; i0 += 64)
            n_valid += __popcll(__ballot(i0 + lane < p.a_pad && p.valid[(size_t)n * p.a_pad + i0 + lane] != 0));
        const bool bypass = n_valid == 1;  // agent_interaction.py:61
        wload1(u, bypass ? gru_first8(W, pw.gru[0], wave) : xlayer_first8(W, pw.inter[0], wave), lane);
        step_load_c_inputs<NTHREADS8>(p, n, row0, tid, X, Hs, H1, H2, GP, LP, DG, dflag);
        __syncthreads();
        TB_STAMP(1);
        if (!bypass) {
            const float* kvd = p.vbias + (size_t)n * p.a_pad;
            const size_t ls = (size_t)p.a_pad * H;
            const float* K0 = p.kin + ((size_t)n * 3) * ls;
            const float* V0 = p.vtin + ((size_t)n * 3) * ls;
            xattn_layer8<true>(W, pw.inter[0], X, xs, K0, V0, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                               xlayer_first8(W, pw.inter[1], wave), LN + 0 * 768);
            xattn_layer8<true>(W, pw.inter[1], X, xs, K0 + ls, V0 + ls, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                               xlayer_first8(W, pw.inter[2], wave), LN + 1 * 768);
            xattn_layer8<true>(W, pw.inter[2], X, xs, K0 + 2 * ls, V0 + 2 * ls, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                               gru_first8(W, pw.gru[0], wave), LN + 2 * 768);
        }
        TB_STAMP(2);
        // ---- 3-layer GRU, one step (agent_temporal.py:147-152): X -> Y -> S1 -> X
        {
            float* hg0 = p.hidden + (((size_t)0 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg1 = p.hidden + (((size_t)1 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg2 = p.hidden + (((size_t)2 * p.n_inst + n) * p.a_pad + row0) * H;
            gru_layer8(W, pw.gru[0], X, Hs, Y, rowvalid, hg0, TM, tid, u, gru_first8(W, pw.gru[1], wave));
            gru_layer8(W, pw.gru[1], Y, H1, S1, rowvalid, hg1, TM, tid, u, gru_first8(W, pw.gru[2], wave));
            gru_layer8(W, pw.gru[2], S1, H2, X, rowvalid, hg2, TM, tid, u, wnext1(W + pw.goal_out_w1, W + pw.goal_out_b1, wave, 16, 0));
        }
        TB_STAMP(3);
        // ---- add_goal, add_latent (traffic_bots.py:240-241)
        fuse_latent_goal8(W, pw.goal_out_w1, pw.goal_out_w2, pw.goal_out_b2, X, CAT, S2, GP, gvalid, rowvalid, tid, u,
                          wnext1(W + pw.lat_out_w1, W + pw.lat_out_b1, wave, 16, 0));
        // action-head branches needed by this tile (action_head.py:69-75): one per agent type present
        const int my_ty = (lane < TM && rowvalid[lane]) ? rtype[lane] : -1;
        const bool has0 = __ballot(my_ty == 0) != 0, has1 = __ballot(my_ty == 1) != 0, has2 = __ballot(my_ty == 2) != 0;
        const WNext1 after_head = do_a ? xlayer_first8(W, pw.as2pl[0], wave) : wnext1(W + pw.head_w1[0], W + pw.head_b1[0], wave);
        const WNext1 h2 = has2 ? wnext1(W + pw.head_w1[2], W + pw.head_b1[2], wave) : after_head;
        const WNext1 h1 = has1 ? wnext1(W + pw.head_w1[1], W + pw.head_b1[1], wave) : h2;
        const WNext1 h0 = has0 ? wnext1(W + pw.head_w1[0], W + pw.head_b1[0], wave) : h1;
        TB_STAMP(4);
        fuse_latent_goal8(W, pw.lat_out_w1, pw.lat_out_w2, pw.lat_out_b2, X, CAT, S2, LP, rowvalid, rowvalid, tid, u, h0);
        TB_STAMP(5);
        if (t == p.tap_step && p.tap_policy_feature)
            store_tile8(p.tap_policy_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

        if (tid < 32) ubuf[tid] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const bool present = ty == 0 ? has0 : (ty == 1 ? has1 : has2);
            if (!present) continue;
            WUnit1 uh = u;
            f32x4 acc = uh.b;
            wmma1_pf(acc, uh, X + m * LDT + kq * 32, u, ty == 0 ? h1 : (ty == 1 ? h2 : after_head), lane);
            st4(cptr(S2, LDT, wave, lane), relu4(acc));
            __syncthreads();
            if (tid < 256) {
                // Linear(128 -> 2): 32 (row, output) pairs x 8 lanes, 16 k each, quad + half-row DPP reduction
                const int pair = tid >> 3, sub = tid & 7, r = pair >> 1, o = pair & 1;
                const float* w2 = W + pw.head_w2[ty] + o * H + sub * 16;
                const float* xs_ = S2 + r * LDT + sub * 16;
                float sacc = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 a4 = lds4(xs_ + 4 * k4), w4 = ldg4(w2 + 4 * k4);
                    sacc = fmaf(a4.x, w4.x, sacc); sacc = fmaf(a4.y, w4.y, sacc);
                    sacc = fmaf(a4.z, w4.z, sacc); sacc = fmaf(a4.w, w4.w, sacc);
                }
                sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0xB1, 0xf, 0xf, true));
                sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x4E, 0xf, 0xf, true));
                sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x141, 0xf, 0xf, true));
                if (sub == 0 && rtype[r] == ty && rowvalid[r]) ubuf[pair] = sacc + W[pw.head_b2[ty] + o];
            }
            __syncthreads();
        }
        __syncthreads();
        TB_STAMP(6);
        step_epilogue(p, t, n, b, row0, n_real, tid, sm, DG);
        __syncthreads();
    } else {
        wload1(u, xlayer_first8(W, pw.as2pl[0], wave), lane);
        __syncthreads();
    }
    TB_STAMP(7);
    if (!do_a) return;

    // =================================== A(t+1) ===================================
    const int t1 = t + 1;
    step_encode_inputs<NTHREADS8>(p, t, n, b, row0, n_real, tid, sm, X);
    if (t1 == p.tap_step && p.tap_agent_feature)
        store_tile8(p.tap_agent_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

    TB_STAMP(8);
    // ---- agent -> map polylines (traffic_bots.py:205-211)
    {
        const float* kvd = p.kbias_pl + (size_t)b * p.p_pad;
        const size_t ls = (size_t)p.p_pad * H;
        const float* K0 = p.kpl + ((size_t)b * 3) * ls;
        const float* V0 = p.vtpl + ((size_t)b * 3) * ls;
        xattn_layer8<true>(W, pw.as2pl[0], X, xs, K0, V0, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u,
                           xlayer_first8(W, pw.as2pl[1], wave), LN + 3 * 768,
                           p.prof + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32);
        xattn_layer8<true>(W, pw.as2pl[1], X, xs, K0 + ls, V0 + ls, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u,
                           xlayer_first8(W, pw.as2pl[2], wave), LN + 4 * 768);
        xattn_layer8<true>(W, pw.as2pl[2], X, xs, K0 + 2 * ls, V0 + 2 * ls, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u,
                           xlayer_first8(W, pw.as2tl[0], wave), LN + 5 * 768);
    }
    TB_STAMP(9);
    // ---- agent -> traffic lights of step min(t1-1, n_hist-1) (waymo_motion.py:287, traffic_bots.py:213-219)
    {
        const int g_tl = b * p.n_tl_hist + min(t1 - 1, p.n_tl_hist - 1);
        const float* kvd = p.kbias_tl + (size_t)g_tl * p.t_pad;
        const size_t ls = (size_t)p.t_pad * H;
        const float* K0 = p.ktl + ((size_t)g_tl * 3) * ls;
        const float* V0 = p.vttl + ((size_t)g_tl * 3) * ls;
        xattn_layer8<true>(W, pw.as2tl[0], X, xs, K0, V0, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u,
                           xlayer_first8(W, pw.as2tl[1], wave), LN + 6 * 768);
        xattn_layer8<true>(W, pw.as2tl[1], X, xs, K0 + ls, V0 + ls, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u,
                           xlayer_first8(W, pw.as2tl[2], wave), LN + 7 * 768);
        xattn_layer8<true>(W, pw.as2tl[2], X, xs, K0 + 2 * ls, V0 + 2 * ls, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u,
                           kvproj_first8(W, pw.inter[0], wave), LN + 8 * 768);
    }
    TB_STAMP(10);
    // ---- hand-off to the next launch: x_mid and the interaction K/V of this tile's agents
    store_tile8(p.x_mid_w + base_row * H, X, LDT, TM, tid);
    {
        const size_t ls = (size_t)p.a_pad * H;
        float* K0 = p.kin_w + ((size_t)n * 3) * ls;
        float* V0 = p.vtin_w + ((size_t)n * 3) * ls;
        kv_project_tile8<true>(W, pw.inter[0], X, S1, K0, V0, p.a_pad, row0, TM, tid, u, kvproj_first8(W, pw.inter[1], wave), LN + 0 * 768);
        kv_project_tile8<true>(W, pw.inter[1], X, S1, K0 + ls, V0 + ls, p.a_pad, row0, TM, tid, u, kvproj_first8(W, pw.inter[2], wave),
                               LN + 1 * 768);
        kv_project_tile8<true>(W, pw.inter[2], X, S1, K0 + 2 * ls, V0 + 2 * ls, p.a_pad, row0, TM, tid, u,
                               kvproj_first8(W, pw.inter[2], wave), LN + 2 * 768);
    }
    TB_STAMP(11);
}

hipError_t configure_step8_kernel() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_step8), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(STEP_LDS_FLOATS * sizeof(float)));
}

void launch_step8(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_step8, grid, dim3(NTHREADS8), STEP_LDS_FLOATS * sizeof(float), s, p, t, do_c, do_a);
}

}  // namespace tb
