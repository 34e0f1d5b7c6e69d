// Rollout kernels (gfx950).  The only all-to-all seam of a simulation step is the agent<->agent interaction
// (every agent needs K/V of all agents of its scene); everything else is independent per agent.  So a step is
// cut there and the two halves of CONSECUTIVE steps are fused into one launch:
//
//   k_step(t) :  [C(t)]  3x agent<->agent attn -> 3-layer GRU -> add_goal -> add_latent -> action head ->
//                        unicycle dynamics -> teacher forcing -> rule check / kill / navigator -> buffer writes
//                [A(t+1)] agent attr + pose PE + InputPeEncoder -> 3x agent->map attn -> 3x agent->TL attn
//                        -> x_mid and the three interaction layers' K/V of the tile's agents
//
// One workgroup (4 waves) owns 16 agents of one rollout instance for the whole launch; grid = (a_pad/16, N);
// S+1 launches per rollout (A(1) alone, S-1 fused, C(S) alone), no host synchronisation.
// Weights stream as register units one unit ahead of the MFMAs (tb_device.hpp); the unit chain runs straight
// through all stages of the launch.
#include "tb_step_common.hpp"
#include "tb_sample.hpp"

namespace tb {

// ------------------------------------------------------------------------------------------------
// hoists
// ------------------------------------------------------------------------------------------------
// K/V of all three layers of a cross-attention block for fixed targets (map polylines, TL stop points):
// loop-invariant over the rollout, the reference recomputes them every step
// (transformer.py:189-192, attention.py:81-87).  grid = (n_pad/16, G)
__global__ __launch_bounds__(NTHREADS) void k_kv_hoist(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2,
                                                      const float* __restrict__ feat /*[G][n_tok][128]*/,
                                                      const uint8_t* __restrict__ fvalid /*[G][n_tok]*/, int n_tok, int n_pad,
                                                      float* __restrict__ Kout /*[G][3][n_pad][128]*/,
                                                      float* __restrict__ VTout /*[G][3][128][n_pad]*/,
                                                      float* __restrict__ kbias /*[G][n_pad] 0 valid / -inf*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;
    float* S1 = smem + TM * LDT;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g = blockIdx.y, tok0 = blockIdx.x * TM;
    const int n_real = max(0, min(TM, n_tok - tok0));
    WUnit u;
    wload(u, kvproj_first(W, l0, wave), lane);
    load_tile(T, LDT, feat + ((size_t)g * n_tok + tok0) * H, n_real, tid);
    if (tid < TM)
        kbias[(size_t)g * n_pad + tok0 + tid] = (tid < n_real && fvalid[(size_t)g * n_tok + tok0 + tid]) ? 0.f : -INFINITY;
    __syncthreads();
    kv_project_tile(W, l0, T, S1, Kout + ((size_t)g * 3 + 0) * n_pad * H, VTout + ((size_t)g * 3 + 0) * H * n_pad, n_pad, tok0,
                    n_real, tid, u, kvproj_first(W, l1, wave));
    kv_project_tile(W, l1, T, S1, Kout + ((size_t)g * 3 + 1) * n_pad * H, VTout + ((size_t)g * 3 + 1) * H * n_pad, n_pad, tok0,
                    n_real, tid, u, kvproj_first(W, l2, wave));
    kv_project_tile(W, l2, T, S1, Kout + ((size_t)g * 3 + 2) * n_pad * H, VTout + ((size_t)g * 3 + 2) * H * n_pad, n_pad, tok0,
                    n_real, tid, u, kvproj_first(W, l2, wave));
}

// simulator init (Dynamics.init, dynamics.py:29-48; TrafficBots.init, traffic_bots.py:153-161) +
// goal / latent `mlp_in` hoists (add_latent_goal.py:57) + latent log-prob (distributions.py:11-15).
// grid = (a_pad/16, N)
__global__ __launch_bounds__(NTHREADS) void k_rollout_init(RolloutP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + OFF_X;
    float* S1 = smem + OFF_S1;
    float* S2 = smem + OFF_S2;
    float* Z = smem + OFF_H;  // [16][20]
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int n = blockIdx.y, b = n / p.k_rep, row0 = blockIdx.x * TM;
    const int n_real = max(0, min(TM, p.n_agent - row0));
    const float* W = p.W;
    const PolicyW& pw = p.pw;
    // ---- state init from history frame 0
    if (tid < TM) {
        const int row = row0 + tid;
        const size_t si = (size_t)n * p.a_pad + row;
        f32x4 st = splat(0.f), ax = splat(0.f);
        uint8_t v = 0, gv = 0;
        if (tid < n_real) {
            const size_t hi = ((size_t)b * p.n_hist + 0) * p.n_agent + row;
            st = ldg4(p.hist_state + hi * 4);
            ax = f32x4{p.hist_vel[hi * 2], p.hist_vel[hi * 2 + 1], p.hist_acc[hi], p.hist_yaw_rate[hi]};
            v = p.hist_valid[hi];
            gv = p.goal_valid0[(size_t)n * p.n_agent + row];
        }
        st4(p.state + si * 4, st);
        st4(p.aux + si * 4, ax);
        p.valid[si] = v;  // both parities: the first launch (A half only) reads one and leaves the other to C(step_start)
        p.vbias[si] = v ? 0.f : -INFINITY;
        p.valid_w[si] = v;
        p.vbias_w[si] = v ? 0.f : -INFINITY;
        p.killed[si] = 0;
        p.goal_valid[si] = gv;
        p.dest_reached[si] = 0;
        p.outside[si] = 0;
    }
    // hidden = 0
    for (int l = 0; l < 3; ++l) {
        float* hb = p.hidden + (((size_t)l * p.n_inst + n) * p.a_pad + row0) * H;
        for (int i = tid; i < TM * H / 4; i += NTHREADS) st4(hb + i * 4, splat(0.f));
    }
    // ---- goal feature gather: map_feature[b, dest[n,row]]  (goal_manager.py:121-139)
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        f32x4 v = splat(0.f);
        if (r < n_real) {
            // (an out-of-range destination index -- a caller-supplied goal_sample, file contents -- reads nothing: zero goal feature,
            // far geometry and no destination flag below; the host mirror rejects such indices before the launch)
            const int d = p.dest[(size_t)n * p.n_agent + row0 + r];
            if (d >= 0 && d < p.n_pl) v = ldg4(p.map_feature + ((size_t)b * p.n_pl + d) * H + c4);
        }
        st4(X + r * LDT + c4, v);
    }
    // latent sample tile
    for (int i = tid; i < TM * 16; i += NTHREADS) {
        const int r = i >> 4, c = i & 15;
        float z = 0.f;
        if (r < n_real) {
            const size_t zi = ((size_t)n * p.n_agent + row0 + r) * 16 + c;
            if (p.latent_draw) {  // MyDist.sample (distributions.py:18-38) with the caller's draws: tb_rollout_io.latent_sample_out
                const bool det = p.latent_eps == nullptr || (p.latent_det != nullptr && p.latent_det[(size_t)n * p.n_agent + row0 + r] != 0);
                z = latent_draw(p.latent_mean[((size_t)b * p.n_agent + row0 + r) * 16 + c], det ? 0.f : p.latent_eps[zi],
                                W[p.latent_log_std + c], det);
                p.o_latent_z[zi] = z;
            } else {
                z = p.latent_z[zi];
            }
        }
        Z[r * 20 + c] = z;
    }
    __syncthreads();
    // latent log prob: sum_d -((z-mu)^2)/(2 var) - log(std) - log(sqrt(2 pi))
    if (tid < n_real) {
        const int row = row0 + tid;
        float lp = 0.f;
        for (int d = 0; d < 16; ++d) {
            const float stdv = expf(W[p.latent_log_std + d]);
            const float diff = Z[tid * 20 + d] - p.latent_mean[((size_t)b * p.n_agent + row) * 16 + d];
            lp += -(diff * diff) / (2.f * (stdv * stdv)) - logf(stdv) - 0.9189385332046727f;
        }
        p.o_latent_logp[(size_t)n * p.n_agent + row] = lp;
    }
    // ---- destination polyline geometry per agent, gathered once (traffic_rule_checker.py:85-98): node position,
    // unit direction (dir / |dir|); invalid nodes get a far position and a zero direction so both tests fail
    for (int i = tid; i < TM * 20; i += NTHREADS) {
        const int r = i / 20, k = i % 20;
        f32x4 g = f32x4{1e30f, 1e30f, 0.f, 0.f};
        if (r < n_real) {
            const int d = p.dest[(size_t)n * p.n_agent + row0 + r];
            const bool dok = d >= 0 && d < p.n_pl;
            const size_t nb = ((size_t)b * p.n_pl + (dok ? d : 0)) * 20 + k;
            if (dok && p.map_valid[nb]) {
                const float ddx = p.map_dir[nb * 2], ddy = p.map_dir[nb * 2 + 1];
                const float nrm = sqrtf(fadd_(fmul_(ddx, ddx), fmul_(ddy, ddy)));
                g = f32x4{p.map_pos[nb * 2], p.map_pos[nb * 2 + 1], ddx / nrm, ddy / nrm};
            }
        }
        st4(p.dest_geo + (((size_t)n * p.a_pad + row0 + r) * 20 + k) * 4, g);
    }
    if (tid < TM) {
        int fl = 0;
        if (tid < n_real) {
            const int d = p.dest[(size_t)n * p.n_agent + row0 + tid];
            const int dty = (d >= 0 && d < p.n_pl) ? p.map_type[(size_t)b * p.n_pl + d] : -1;
            fl = (dty >= 0 && dty < 4 ? 1 : 0) | (dty == 4 ? 2 : 0);
        }
        p.dest_flag[(size_t)n * p.a_pad + row0 + tid] = fl;
    }
    // ---- add_goal.mlp_in : 3 x (Linear128 -> LN [-> ReLU]) ; last LN output is stored un-masked, the mask and
    // the trailing ReLU (mlp.py:80-84) are applied per step because goal_valid changes.
    float* cur = X;
    float* nxt = S1;
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        f32x4 acc[2];
        linear128<128>(acc, W + pw.goal_in_w[l], W + pw.goal_in_b[l], cur + m * LDT + kq * 32, wave, lane);
        st4(cptr(S2, LDT, 2 * wave, lane), acc[0]);
        st4(cptr(S2, LDT, 2 * wave + 1, lane), acc[1]);
        __syncthreads();
        layernorm_tile(S2, LDT, nxt, LDT, W + pw.goal_in_g[l], W + pw.goal_in_be[l], tid);
        __syncthreads();
        if (l < 2) {
            for (int i = tid; i < TM * 32; i += NTHREADS) {
                float* q = nxt + (i >> 5) * LDT + (i & 31) * 4;
                st4(q, relu4(lds4(q)));
            }
            __syncthreads();
        }
        float* t = cur; cur = nxt; nxt = t;
    }
    store_tile(p.goal_pre + ((size_t)n * p.a_pad + row0) * H, cur, LDT, TM, tid);
    // ---- add_latent.mlp_in : Linear(16->128) -> ReLU -> Linear(128->128)
    {
        const int tiles[2] = {2 * wave, 2 * wave + 1};
        f32x4 acc[2] = {bias4(W + pw.lat_in_b1, tiles[0], lane), bias4(W + pw.lat_in_b1, tiles[1], lane)};
        gemm_acc<16, 2>(acc, W + pw.lat_in_w1, tiles, Z + m * 20 + kq * 4, lane);
        __syncthreads();
        st4(cptr(S2, LDT, tiles[0], lane), relu4(acc[0]));
        st4(cptr(S2, LDT, tiles[1], lane), relu4(acc[1]));
        __syncthreads();
        f32x4 acc2[2];
        linear128<128>(acc2, W + pw.lat_in_w2, W + pw.lat_in_b2, S2 + m * LDT + kq * 32, wave, lane);
        float* dst = p.lat_pre + ((size_t)n * p.a_pad + row0) * H;
        st4(dst + (size_t)m * H + tiles[0] * 16 + kq * 4, acc2[0]);
        st4(dst + (size_t)m * H + tiles[1] * 16 + kq * 4, acc2[1]);
    }
}

// final simulator state -> caller buffers (Dynamics.agent_state / agent_valid, TrafficBots.hidden)
__global__ void k_rollout_final(RolloutP p, float* __restrict__ f_state, uint8_t* __restrict__ f_valid, float* __restrict__ f_hidden) {
    const int n = blockIdx.x;
    for (int a = threadIdx.x; a < p.n_agent; a += blockDim.x) {
        const size_t si = (size_t)n * p.a_pad + a, di = (size_t)n * p.n_agent + a;
        if (f_state) st4(f_state + di * 4, ldg4(p.state + si * 4));
        if (f_valid) f_valid[di] = p.valid[si];
    }
    if (f_hidden) {
        for (int l = 0; l < 3; ++l)
            for (int i = threadIdx.x; i < p.n_agent * 32; i += blockDim.x) {
                const int a = i >> 5, c4 = (i & 31) * 4;
                st4(f_hidden + (((size_t)l * p.n_inst + n) * p.n_agent + a) * H + c4,
                    ldg4(p.hidden + (((size_t)l * p.n_inst + n) * p.a_pad + a) * H + c4));
            }
    }
}

// ------------------------------------------------------------------------------------------------
// h = relu(W2 relu(W1 [x ; u] + b1) + b2); h = zvalid ? h : 0; x = rowvalid ? h + x : 0   (add_latent_goal.py:57-77)
//   PRE : [16][LDT] LDS copy of mlp_in's (un-masked) output for the tile's agents
//   uw : in = first half (k 0..127) of W1 (carries b1), out = `nxt`
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fuse_latent_goal(const float* __restrict__ W, uint32_t w1, uint32_t w2, uint32_t b2, float* X, float* CAT,
                                                 float* S2, const float* PRE, const uint8_t* zvalid, const uint8_t* rowvalid, int tid,
                                                 WUnit& uw, const WNext& nxt, long long* prof = nullptr) {
    const int wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    // CAT = [x ; relu(mask(pre))]
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        st4(CAT + r * LDC + c4, lds4(X + r * LDT + c4));
        st4(CAT + r * LDC + 128 + c4, zvalid[r] ? relu4(lds4(PRE + r * LDT + c4)) : splat(0.f));
    }
    __syncthreads();
#ifdef TB_PROFILE
    (void)prof;
#endif
    WUnit u2;
    {
        const int ta = 2 * wave, tb_ = 2 * wave + 1;
        f32x4 acc[2] = {uw.b[0], uw.b[1]};
        const float* xr = CAT + m * LDC + kq * 64;
        wmma_pf(acc[0], acc[1], uw, xr, u2, wnext(W + w1, nullptr, ta, tb_, 16, 8), lane);
#ifdef TB_PROFILE
        (void)prof;
#endif
        wmma_pf(acc[0], acc[1], u2, xr + 32, uw, wstd(W + w2, W + b2, wave), lane);
#ifdef TB_PROFILE
        (void)prof;
#endif
        st4(cptr(S2, LDT, ta, lane), relu4(acc[0]));
        st4(cptr(S2, LDT, tb_, lane), relu4(acc[1]));
    }
    __syncthreads();
#ifdef TB_PROFILE
    (void)prof;
#endif
    {
        f32x4 acc[2] = {uw.b[0], uw.b[1]};
        wmma_pf(acc[0], acc[1], uw, S2 + m * LDT + kq * 32, u2, nxt, lane);
        const bool zv = zvalid[m] != 0, rv = rowvalid[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X, LDT, 2 * wave + t, lane);
            const f32x4 h = zv ? relu4(acc[t]) : splat(0.f);
            st4(px, rv ? h + lds4(px) : splat(0.f));
        }
        uw = u2;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_step(RolloutP p, int t, int do_c, int do_a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    kernarg_warm<(int)sizeof(RolloutP) + 12 + 32>();  // (tb_step_common.hpp: one miss of the cold scalar cache instead of a chain)
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

    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    int n, rt;
    step_tile_map(n, rt);
    const int b = n / p.k_rep, row0 = rt * TM;
    const int n_real = max(0, min(TM, p.n_agent - row0));
    const float* W = p.W;
    const PolicyW& pw = p.pw;
    const size_t base_row = (size_t)n * p.a_pad + row0;

    WUnit u;
    TB_STAMP(0);
    {
        // LayerNorm parameter blocks -> LDS (slots 0..2 interaction, 3..5 as2pl, 6..8 as2tl)
        const uint32_t base[9] = {pw.inter[0].ln1_g, pw.inter[1].ln1_g, pw.inter[2].ln1_g, pw.as2pl[0].ln1_g, pw.as2pl[1].ln1_g,
                                  pw.as2pl[2].ln1_g, pw.as2tl[0].ln1_g, pw.as2tl[1].ln1_g, pw.as2tl[2].ln1_g};
#pragma unroll
        for (int sl = 0; sl < 9; ++sl)
            if (tid < 192) st4(LN + sl * 768 + tid * 4, ldg4(W + base[sl] + tid * 4));
    }
    TB_STAMP(12);
    if (tid < TM) {
        rtype[tid] = (tid < n_real) ? p.agent_type[(size_t)b * p.n_agent + row0 + tid] : -1;
        const size_t si = base_row + tid;
        const f32x4 st = ldg4(p.state + si * 4), ax = ldg4(p.aux + si * 4);
        rst[tid].st[0] = st.x; rst[tid].st[1] = st.y; rst[tid].st[2] = st.z; rst[tid].st[3] = st.w;
        rst[tid].aux[0] = ax.x; rst[tid].aux[1] = ax.y; rst[tid].aux[2] = ax.z; rst[tid].aux[3] = ax.w;
        rowvalid[tid] = p.valid[si];
        gvalid[tid] = p.goal_valid[si];
    }
    if (tid == TM) dflag[EPI_POISON_WORD] = 0;  // (read by step_epilogue16; this kernel has no helper hand-off that could raise it)
    TB_STAMP(13);

    if (do_c) {
        // =================================== C(t) ===================================
        // number of valid agents of the instance (agent_interaction.py:61: exactly one -> bypass the block)
        int n_valid = 0;
        for (int i0 = 0; i0 < p.a_pad; i0 += 64)  // every wave counts for itself: no LDS round trip
            n_valid += __popcll(__ballot(i0 + lane < p.a_pad && p.valid[(size_t)n * p.a_pad + i0 + lane] != 0));
        const bool bypass = n_valid == 1;
        TB_STAMP(14);
        wload(u, bypass ? gru_first(W, pw.gru[0], wave) : xlayer_first(W, pw.inter[0], wave), lane);
        // one burst of per-tile inputs for the whole C half (a single exposed global latency)
        step_load_c_inputs<NTHREADS>(p, n, row0, tid, X, Hs, H1, H2, GP, LP, DG, dflag);
        TB_STAMP(15);
        __syncthreads();
        TB_STAMP(1);
        if (!bypass) {
            const float* kvd = p.vbias + (size_t)n * p.a_pad;
            const size_t ls = (size_t)p.a_pad * H;
            const float* K0 = p.kin + ((size_t)n * 3) * ls;
            const float* V0 = p.vtin + ((size_t)n * 3) * ls;
            xattn_layer<true>(W, pw.inter[0], X, S1, S2, K0, V0, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                        xlayer_first(W, pw.inter[1], wave), LN + 0 * 768);
            xattn_layer<true>(W, pw.inter[1], X, S1, S2, K0 + ls, V0 + ls, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                        xlayer_first(W, pw.inter[2], wave), LN + 1 * 768);
            xattn_layer<true>(W, pw.inter[2], X, S1, S2, K0 + 2 * ls, V0 + 2 * ls, kvd, p.a_pad, row0, rowvalid, novalid_s, tid, u,
                        gru_first(W, pw.gru[0], wave), LN + 2 * 768);
        }
        TB_STAMP(2);
        // ---- 3-layer GRU, one step (agent_temporal.py:147-152): X -> Y -> S1 -> X
        {
            float* hg0 = p.hidden + (((size_t)0 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg1 = p.hidden + (((size_t)1 * p.n_inst + n) * p.a_pad + row0) * H;
            float* hg2 = p.hidden + (((size_t)2 * p.n_inst + n) * p.a_pad + row0) * H;
            gru_layer(W, pw.gru[0], X, Hs, Y, rowvalid, hg0, TM, tid, u, gru_first(W, pw.gru[1], wave));
            gru_layer(W, pw.gru[1], Y, H1, S1, rowvalid, hg1, TM, tid, u, gru_first(W, pw.gru[2], wave));
            gru_layer(W, pw.gru[2], S1, H2, X, rowvalid, hg2, TM, tid, u,
                      wnext(W + pw.goal_out_w1, W + pw.goal_out_b1, 2 * wave, 2 * wave + 1, 16, 0));
        }
        TB_STAMP(3);
        // ---- add_goal, add_latent (traffic_bots.py:240-241)
        fuse_latent_goal(W, pw.goal_out_w1, pw.goal_out_w2, pw.goal_out_b2, X, CAT, S2, GP, gvalid, rowvalid, tid, u,
                         wnext(W + pw.lat_out_w1, W + pw.lat_out_b1, 2 * wave, 2 * wave + 1, 16, 0),
                         p.prof + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32);
        // action-head branches needed by this tile (action_head.py:69-75): one per agent type present
        const int my_ty = (lane < TM && rowvalid[lane]) ? rtype[lane] : -1;  // (rowvalid / rtype are stable since C start)
        const bool has0 = __ballot(my_ty == 0) != 0, has1 = __ballot(my_ty == 1) != 0, has2 = __ballot(my_ty == 2) != 0;
        const WNext after_head = do_a ? xlayer_first(W, pw.as2pl[0], wave) : wstd(W + pw.head_w1[0], W + pw.head_b1[0], wave);
        const WNext h2 = has2 ? wstd(W + pw.head_w1[2], W + pw.head_b1[2], wave) : after_head;
        const WNext h1 = has1 ? wstd(W + pw.head_w1[1], W + pw.head_b1[1], wave) : h2;
        const WNext h0 = has0 ? wstd(W + pw.head_w1[0], W + pw.head_b1[0], wave) : h1;
        TB_STAMP(4);
        fuse_latent_goal(W, pw.lat_out_w1, pw.lat_out_w2, pw.lat_out_b2, X, CAT, S2, LP, rowvalid, rowvalid, tid, u, h0);
        TB_STAMP(5);
        if ((t == p.tap_step || p.tap_step == -2) && p.tap_policy_feature)
            store_tile(p.tap_policy_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

        if (tid < 32) ubuf[tid] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const bool present = ty == 0 ? has0 : (ty == 1 ? has1 : has2);
            if (!present) continue;
            WUnit uh = u;
            f32x4 acc[2] = {uh.b[0], uh.b[1]};
            wmma_pf(acc[0], acc[1], uh, X + m * LDT + kq * 32, u, ty == 0 ? h1 : (ty == 1 ? h2 : after_head), lane);
            st4(cptr(S2, LDT, 2 * wave, lane), relu4(acc[0]));
            st4(cptr(S2, LDT, 2 * wave + 1, lane), relu4(acc[1]));
            __syncthreads();
            {
                // Linear(128 -> 2): 32 (row, output) pairs x 8 lanes, 16 k each, quad + half-row DPP reduction
                const int pair = tid >> 3, sub = tid & 7, r = pair >> 1, o = pair & 1;
                const float* w2 = W + pw.head_w2[ty] + o * H + sub * 16;
                const float* xs = S2 + r * LDT + sub * 16;
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
                if (sub == 0 && rtype[r] == ty && rowvalid[r]) ubuf[pair] = sacc + W[pw.head_b2[ty] + o];
            }
            __syncthreads();
        }
        __syncthreads();
        TB_STAMP(6);

        // ---- per-agent epilogue
        // (the 16-lanes-per-agent epilogue of the XDL kernels, loads included: sampled actions -- RolloutP::action_eps -- and the
        // per-call state / action overrides of tb_rollout_step_ex are served by the exact-fp32 kernel as well)
        step_epilogue16<false>(p, t, n, b, row0, n_real, tid, sm, DG);
        __syncthreads();
    } else {
        wload(u, xlayer_first(W, pw.as2pl[0], wave), lane);
        __syncthreads();
    }
    TB_STAMP(7);
    if (!do_a) return;

    // =================================== A(t+1) ===================================
    const int t1 = t + 1;
    step_encode_inputs<NTHREADS>(p, t, n, b, row0, n_real, tid, sm, X);
    if ((t1 == p.tap_step || p.tap_step == -2) && p.tap_agent_feature)
        store_tile(p.tap_agent_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

    TB_STAMP(8);
    // ---- agent -> map polylines (traffic_bots.py:205-211)
    {
        const float* kvd = p.kbias_pl + (size_t)b * p.p_pad;
        const size_t ls = (size_t)p.p_pad * H;
        const float* K0 = p.kpl + ((size_t)b * 3) * ls;
        const float* V0 = p.vtpl + ((size_t)b * 3) * ls;
        xattn_layer<true>(W, pw.as2pl[0], X, S1, S2, K0, V0, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u, xlayer_first(W, pw.as2pl[1], wave),
                    LN + 3 * 768, p.prof + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32);
        xattn_layer<true>(W, pw.as2pl[1], X, S1, S2, K0 + ls, V0 + ls, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u,
                    xlayer_first(W, pw.as2pl[2], wave), LN + 4 * 768);
        xattn_layer<true>(W, pw.as2pl[2], X, S1, S2, K0 + 2 * ls, V0 + 2 * ls, kvd, p.p_pad, -1, rowvalid, novalid_s, tid, u,
                    xlayer_first(W, pw.as2tl[0], wave), LN + 5 * 768);
    }
    TB_STAMP(9);
    // ---- agent -> traffic lights of step min(t1-1, n_hist-1) (waymo_motion.py:287, traffic_bots.py:213-219)
    {
        const int g_tl = b * p.n_tl_hist + min(t1 - 1, p.n_tl_hist - 1);
        const float* kvd = p.kbias_tl + (size_t)g_tl * p.t_pad;
        const size_t ls = (size_t)p.t_pad * H;
        const float* K0 = p.ktl + ((size_t)g_tl * 3) * ls;
        const float* V0 = p.vttl + ((size_t)g_tl * 3) * ls;
        xattn_layer<true>(W, pw.as2tl[0], X, S1, S2, K0, V0, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u, xlayer_first(W, pw.as2tl[1], wave),
                    LN + 6 * 768);
        xattn_layer<true>(W, pw.as2tl[1], X, S1, S2, K0 + ls, V0 + ls, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u,
                    xlayer_first(W, pw.as2tl[2], wave), LN + 7 * 768);
        xattn_layer<true>(W, pw.as2tl[2], X, S1, S2, K0 + 2 * ls, V0 + 2 * ls, kvd, p.t_pad, -1, rowvalid, novalid_s, tid, u,
                    kvproj_first(W, pw.inter[0], wave), LN + 8 * 768);
    }
    TB_STAMP(10);
    // ---- hand-off to the next launch: x_mid and the interaction K/V of this tile's agents (tgt = block input for
    // all three layers, agent_interaction.py:51-52 + transformer.py:82-92)
    store_tile(p.x_mid_w + base_row * H, X, LDT, TM, tid);
    {
        const size_t ls = (size_t)p.a_pad * H;
        float* K0 = p.kin_w + ((size_t)n * 3) * ls;
        float* V0 = p.vtin_w + ((size_t)n * 3) * ls;
        kv_project_tile<true>(W, pw.inter[0], X, S1, K0, V0, p.a_pad, row0, TM, tid, u, kvproj_first(W, pw.inter[1], wave), LN + 0 * 768);
        kv_project_tile<true>(W, pw.inter[1], X, S1, K0 + ls, V0 + ls, p.a_pad, row0, TM, tid, u, kvproj_first(W, pw.inter[2], wave), LN + 1 * 768);
        kv_project_tile<true>(W, pw.inter[2], X, S1, K0 + 2 * ls, V0 + 2 * ls, p.a_pad, row0, TM, tid, u, kvproj_first(W, pw.inter[2], wave),
                        LN + 2 * 768);
    }
    TB_STAMP(11);
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
size_t step_lds_bytes() { return (size_t)STEP_LDS_FLOATS * sizeof(float); }

void launch_kv_hoist(const float* W, const XLayerW* L3, const float* feat, const uint8_t* fvalid, int G, int n_tok, int n_pad,
                     float* K, float* VT, float* kbias, hipStream_t s) {
    dim3 grid(n_pad / TM, G);
    hipLaunchKernelGGL(k_kv_hoist, grid, dim3(NTHREADS), 2 * TM * LDT * sizeof(float), s, W, L3[0], L3[1], L3[2], feat, fvalid,
                       n_tok, n_pad, K, VT, kbias);
}

// the step kernels carve > 64 KiB of dynamic LDS: raise the per-function limit (per device; called from tb_finalize_weights)
hipError_t configure_rollout_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_step), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)step_lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_init), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)step_lds_bytes());
}

void launch_rollout_init(const RolloutP& p, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_rollout_init, grid, dim3(NTHREADS), step_lds_bytes(), s, p);
}

void launch_step(const RolloutP& p, int t, int do_c, int do_a, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_step, grid, dim3(NTHREADS), step_lds_bytes(), s, p, t, do_c, do_a);
}

void launch_rollout_final(const RolloutP& p, float* f_state, uint8_t* f_valid, float* f_hidden, hipStream_t s) {
    hipLaunchKernelGGL(k_rollout_final, dim3(p.n_inst), dim3(256), 0, s, p, f_state, f_valid, f_hidden);
}

}  // namespace tb
