// Rollout kernels (gfx950).  Per simulation step two launches, cut at the only all-to-all seam of the
// step (every agent's interaction layer needs K/V of all agents of its scene):
//   step_a : agent attr + pose PE + InputPeEncoder -> 3x agent->map attn -> 3x agent->TL attn
//            -> x_mid and the 3 interaction layers' K/V of the tile's agents          (rows independent)
//   step_c : 3x agent<->agent attn -> 3-layer GRU -> add_goal -> add_latent -> action head ->
//            unicycle dynamics -> teacher forcing -> rule check / kill / navigator -> buffer writes
// One workgroup (4 waves) owns 16 agents of one rollout instance; grid = (a_pad/16, N).
#include "tb_rollout.hpp"

namespace tb {

// LDS carve for the step kernels (floats)
constexpr int OFF_X = 0;                   // [16][LDT] residual stream
constexpr int OFF_S1 = OFF_X + TM * LDT;   // [16][LDT]
constexpr int OFF_S2 = OFF_S1 + TM * LDT;  // [16][LDT]
constexpr int OFF_H = OFF_S2 + TM * LDT;   // [16][LDT] GRU previous hidden
constexpr int OFF_Y = OFF_H + TM * LDT;    // [16][LDT] GRU out ping
constexpr int OFF_CAT = OFF_Y + TM * LDT;  // [16][LDC] concat tile
constexpr int OFF_SMALL = OFF_CAT + TM * LDC;
constexpr int STEP_LDS_FLOATS = OFF_SMALL + 16 * 16 /*attr*/ + 16 * 32 /*enc hidden*/ + 64 /*u*/ + 64;

__device__ __forceinline__ float fmul_(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd_(float a, float b) { return __fadd_rn(a, b); }

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
                                                      uint8_t* __restrict__ kvalid /*[G][n_pad]*/) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;
    float* S1 = smem + TM * LDT;
    const int tid = threadIdx.x, g = blockIdx.y, tok0 = blockIdx.x * TM;
    const int n_real = min(TM, n_tok - tok0);
    load_tile(T, LDT, feat + ((size_t)g * n_tok + tok0) * H, n_real, tid);
    if (tid < TM) kvalid[(size_t)g * n_pad + tok0 + tid] = (tid < n_real) ? fvalid[(size_t)g * n_tok + tok0 + tid] : 0;
    __syncthreads();
    const XLayerW* Ls[3] = {&l0, &l1, &l2};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        kv_project_tile(W, *Ls[l], T, S1, Kout + ((size_t)g * 3 + l) * n_pad * H, VTout + ((size_t)g * 3 + l) * H * n_pad,
                        n_pad, tok0, n_real, tid);
    }
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
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int n = blockIdx.y, b = n / p.k_rep, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.n_agent - row0);
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
        p.valid[si] = v;
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
            const int d = p.dest[(size_t)n * p.n_agent + row0 + r];
            v = ldg4(p.map_feature + ((size_t)b * p.n_pl + d) * H + c4);
        }
        st4(X + r * LDT + c4, v);
    }
    // latent sample tile
    for (int i = tid; i < TM * 16; i += NTHREADS) {
        const int r = i >> 4, c = i & 15;
        Z[r * 20 + c] = (r < n_real) ? p.latent_z[((size_t)n * p.n_agent + row0 + r) * 16 + c] : 0.f;
    }
    __syncthreads();
    // latent log prob: sum_d -((z-mu)^2)/(2 var) - log(std) - log(sqrt(2 pi))
    if (tid < n_real) {
        const int row = row0 + tid;
        float lp = 0.f;
        for (int d = 0; d < 16; ++d) {
            const float stdv = expf(W[pw.latent_log_std + d]);
            const float diff = Z[tid * 20 + d] - p.latent_mean[((size_t)b * p.n_agent + row) * 16 + d];
            lp += -(diff * diff) / (2.f * (stdv * stdv)) - logf(stdv) - 0.9189385332046727f;
        }
        p.o_latent_logp[(size_t)n * p.n_agent + row] = lp;
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
        __syncthreads();  // goal tile stores above read `cur`; S2 is free again after this barrier
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

// ------------------------------------------------------------------------------------------------
// step kernel A
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_step_a(RolloutP p, int t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + OFF_X;
    float* S1 = smem + OFF_S1;
    float* S2 = smem + OFF_S2;
    float* attr = smem + OFF_SMALL;          // [16][16]
    float* ench = attr + 16 * 16;            // [16][32]
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(ench + 16 * 32);  // [16]
    uint8_t* novalid_s = rowvalid + 16;                              // [16]
    float* pose = reinterpret_cast<float*>(rowvalid + 32);           // [16][4]  x,y,yaw,-

    const int tid = threadIdx.x;
    const int n = blockIdx.y, b = n / p.k_rep, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.n_agent - row0);
    const float* W = p.W;
    const PolicyW& pw = p.pw;

    // ---- agent attributes (sc_input.py:142-165): vel2, spd, yaw_rate, acc, size3, type one-hot3
    if (tid < TM) {
        const int row = row0 + tid;
        const size_t si = (size_t)n * p.a_pad + row;
        const f32x4 st = ldg4(p.state + si * 4);
        const f32x4 ax = ldg4(p.aux + si * 4);
        const uint8_t v = p.valid[si];
        rowvalid[tid] = v;
        float* a = attr + tid * 16;
        int ty = -1;
        f32x4 sz = splat(0.f);
        if (tid < n_real) {
            ty = p.agent_type[(size_t)b * p.n_agent + row];
            const float* s = p.agent_size + ((size_t)b * p.n_agent + row) * 3;
            sz = f32x4{s[0], s[1], s[2], 0.f};
        }
        a[0] = ax.x; a[1] = ax.y; a[2] = st.w; a[3] = ax.w; a[4] = ax.z;
        a[5] = sz.x; a[6] = sz.y; a[7] = sz.z;
        a[8] = ty == 0 ? 1.f : 0.f; a[9] = ty == 1 ? 1.f : 0.f; a[10] = ty == 2 ? 1.f : 0.f;
        pose[tid * 4 + 0] = st.x; pose[tid * 4 + 1] = st.y; pose[tid * 4 + 2] = st.z;
    }
    __syncthreads();
    // ---- pose PE (pose_pe.py:57-62, pos_emb.py:24-25,54-55): 48 sincos per row, 3 per thread
    {
        const int row = tid >> 4, i = tid & 15;
        const float px = pose[row * 4], py = pose[row * 4 + 1], pyaw = pose[row * 4 + 2];
        float* xr = X + row * LDT + 32;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int j = i * 3 + u;
            float arg;
            int c_cos, c_sin;
            if (j < 12) {
                arg = px * W[pw.pe_fxy + j]; c_cos = j; c_sin = 12 + j;
            } else if (j < 24) {
                arg = py * W[pw.pe_fxy + j - 12]; c_cos = 24 + (j - 12); c_sin = 36 + (j - 12);
            } else {
                arg = pyaw * W[pw.pe_fyaw + j - 24]; c_cos = 48 + (j - 24); c_sin = 72 + (j - 24);
            }
            // fp64 sin/cos of the fp32 argument, rounded once: within 0.5 ulp of exact, i.e. as close as
            // possible to whatever libm the reference's host uses (48 per agent, negligible)
            double sv, cv;
            sincos((double)arg, &sv, &cv);
            xr[c_cos] = (float)cv;
            xr[c_sin] = (float)sv;
        }
    }
    // ---- InputPeEncoder MLP 11 -> 32 -> 32 (input_pe_encoder.py:52-54), 2 outputs per thread
    {
        const int row = tid >> 4, o0 = (tid & 15) * 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int o = o0 + u;
            float s = W[pw.enc_b1 + o];
            for (int k = 0; k < 11; ++k) s = fmaf(attr[row * 16 + k], W[pw.enc_w1 + o * 11 + k], s);
            ench[row * 32 + o] = fmaxf(s, 0.f);
        }
    }
    __syncthreads();
    {
        const int row = tid >> 4, o0 = (tid & 15) * 2;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int o = o0 + u;
            float s = W[pw.enc_b2 + o];
            for (int k = 0; k < 32; ++k) s = fmaf(ench[row * 32 + k], W[pw.enc_w2 + o * 32 + k], s);
            X[row * LDT + o] = s;
        }
    }
    __syncthreads();
    // zero invalid rows (input_pe_encoder.py:59)
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5;
        if (!rowvalid[r]) st4(X + r * LDT + (i & 31) * 4, splat(0.f));
    }
    __syncthreads();
    if (t == p.tap_step && p.tap_agent_feature)
        store_tile(p.tap_agent_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

    // ---- agent -> map polylines (traffic_bots.py:205-211)
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        xattn_layer(W, pw.as2pl[l], X, S1, S2, p.kpl + ((size_t)b * 3 + l) * p.p_pad * H,
                    p.vtpl + ((size_t)b * 3 + l) * H * p.p_pad, p.kvalid_pl + (size_t)b * p.p_pad, p.p_pad, -1, rowvalid,
                    novalid_s, tid);
    }
    // ---- agent -> traffic lights of step min(t-1, n_hist-1) (waymo_motion.py:287, traffic_bots.py:213-219)
    const int g_tl = b * p.n_hist + min(t - 1, p.n_hist - 1);
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        xattn_layer(W, pw.as2tl[l], X, S1, S2, p.ktl + ((size_t)g_tl * 3 + l) * p.t_pad * H,
                    p.vttl + ((size_t)g_tl * 3 + l) * H * p.t_pad, p.kvalid_tl + (size_t)g_tl * p.t_pad, p.t_pad, -1, rowvalid,
                    novalid_s, tid);
    }
    // ---- hand-off to step_c: x_mid and the interaction K/V of this tile's agents (tgt = block input for all
    // three layers, agent_interaction.py:51-52 + transformer.py:82-92)
    store_tile(p.x_mid + ((size_t)n * p.a_pad + row0) * H, X, LDT, TM, tid);
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        kv_project_tile(W, pw.inter[l], X, S1, p.kin + ((size_t)n * 3 + l) * p.a_pad * H,
                        p.vtin + ((size_t)n * 3 + l) * H * p.a_pad, p.a_pad, row0, TM, tid);
    }
}

// ------------------------------------------------------------------------------------------------
// step kernel C
// ------------------------------------------------------------------------------------------------
// h = relu(W2 relu(W1 [x ; u] + b1) + b2); h = zvalid ? h : 0; x = rowvalid ? h + x : 0   (add_latent_goal.py:57-77)
__device__ __forceinline__ void fuse_latent_goal(const float* __restrict__ W, uint32_t w1, uint32_t b1, uint32_t w2, uint32_t b2,
                                                 float* X, float* CAT, float* S2, const float* __restrict__ pre_rows /*[16][128] global*/,
                                                 const uint8_t* zvalid, const uint8_t* rowvalid, int tid) {
    const int wave = tid >> 6, lane = tid & 63, kq = lane >> 4, m = lane & 15;
    // CAT = [x ; relu(mask(pre))]
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        st4(CAT + r * LDC + c4, lds4(X + r * LDT + c4));
        const f32x4 u = zvalid[r] ? relu4(ldg4(pre_rows + (size_t)r * H + c4)) : splat(0.f);
        st4(CAT + r * LDC + 128 + c4, u);
    }
    __syncthreads();
    {
        const int tiles[2] = {2 * wave, 2 * wave + 1};
        f32x4 acc[2] = {bias4(W + b1, tiles[0], lane), bias4(W + b1, tiles[1], lane)};
        gemm_acc<256, 2>(acc, W + w1, tiles, CAT + m * LDC + kq * 64, lane);
        st4(cptr(S2, LDT, tiles[0], lane), relu4(acc[0]));
        st4(cptr(S2, LDT, tiles[1], lane), relu4(acc[1]));
    }
    __syncthreads();
    {
        f32x4 acc[2];
        linear128<128>(acc, W + w2, W + b2, S2 + m * LDT + kq * 32, wave, lane);
        const bool zv = zvalid[m] != 0, rv = rowvalid[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X, LDT, 2 * wave + t, lane);
            const f32x4 h = zv ? relu4(acc[t]) : splat(0.f);
            st4(px, rv ? h + lds4(px) : splat(0.f));
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(NTHREADS) void k_step_c(RolloutP p, int t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem + OFF_X;
    float* S1 = smem + OFF_S1;
    float* S2 = smem + OFF_S2;
    float* Hs = smem + OFF_H;
    float* Y = smem + OFF_Y;
    float* CAT = smem + OFF_CAT;
    float* ubuf = smem + OFF_SMALL;  // [16][2] action means
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(ubuf + 32);
    uint8_t* novalid_s = rowvalid + 16;
    uint8_t* gvalid = rowvalid + 32;
    int* rtype = reinterpret_cast<int*>(rowvalid + 48);  // [16]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int n = blockIdx.y, b = n / p.k_rep, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.n_agent - row0);
    const float* W = p.W;
    const PolicyW& pw = p.pw;
    const size_t base_row = (size_t)n * p.a_pad + row0;

    load_tile(X, LDT, p.x_mid + base_row * H, TM, tid);
    if (tid < TM) {
        rowvalid[tid] = p.valid[base_row + tid];
        gvalid[tid] = p.goal_valid[base_row + tid];
        rtype[tid] = (tid < n_real) ? p.agent_type[(size_t)b * p.n_agent + row0 + tid] : -1;
    }
    // number of valid agents of the instance (agent_interaction.py:61: exactly one -> bypass the block)
    const int n_valid = __syncthreads_count(tid < p.a_pad && p.valid[(size_t)n * p.a_pad + tid]);
    if (n_valid != 1) {
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            xattn_layer(W, pw.inter[l], X, S1, S2, p.kin + ((size_t)n * 3 + l) * p.a_pad * H,
                        p.vtin + ((size_t)n * 3 + l) * H * p.a_pad, p.valid + (size_t)n * p.a_pad, p.a_pad, row0, rowvalid,
                        novalid_s, tid);
        }
    }
    // ---- 3-layer GRU, one step (agent_temporal.py:147-152)
    {
        float* in = X;
        float* out = Y;
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            float* hg = p.hidden + (((size_t)l * p.n_inst + n) * p.a_pad + row0) * H;
            load_tile(Hs, LDT, hg, TM, tid);
            __syncthreads();
            gru_layer(W, pw.gru[l], in, Hs, out, rowvalid, hg, TM, tid);
            float* tmp = in; in = out; out = (tmp == X) ? S1 : tmp;  // ping-pong X -> Y -> S1 -> Y
        }
        // result of layer 2 sits in `in`; move to X if needed
        if (in != X) {
            for (int i = tid; i < TM * 32; i += NTHREADS) {
                const int r = i >> 5, c4 = (i & 31) * 4;
                st4(X + r * LDT + c4, lds4(in + r * LDT + c4));
            }
            __syncthreads();
        }
    }
    // ---- add_goal, add_latent (traffic_bots.py:240-241)
    fuse_latent_goal(W, pw.goal_out_w1, pw.goal_out_b1, pw.goal_out_w2, pw.goal_out_b2, X, CAT, S2,
                     p.goal_pre + base_row * H, gvalid, rowvalid, tid);
    fuse_latent_goal(W, pw.lat_out_w1, pw.lat_out_b1, pw.lat_out_w2, pw.lat_out_b2, X, CAT, S2,
                     p.lat_pre + base_row * H, rowvalid, rowvalid, tid);
    if (t == p.tap_step && p.tap_policy_feature)
        store_tile(p.tap_policy_feature + ((size_t)n * p.n_agent + row0) * H, X, LDT, n_real, tid);

    // ---- action head per agent type (action_head.py:69-75); branches without an agent in the tile are skipped
    if (tid < 32) ubuf[tid] = 0.f;
#pragma unroll 1
    for (int ty = 0; ty < 3; ++ty) {
        const int present = __syncthreads_or(tid < TM && rtype[tid] == ty && rowvalid[tid]);
        if (!present) continue;
        f32x4 acc[2];
        linear128<128>(acc, W + pw.head_w1[ty], W + pw.head_b1[ty], X + m * LDT + kq * 32, wave, lane);
        st4(cptr(S2, LDT, 2 * wave, lane), relu4(acc[0]));
        st4(cptr(S2, LDT, 2 * wave + 1, lane), relu4(acc[1]));
        __syncthreads();
        if (tid < 32) {
            const int r = tid >> 1, o = tid & 1;
            if (rtype[r] == ty && rowvalid[r]) {
                float s = W[pw.head_b2[ty] + o];
                const float* w2 = W + pw.head_w2[ty] + o * H;
                for (int k = 0; k < H; ++k) s = fmaf(S2[r * LDT + k], w2[k], s);
                ubuf[tid] = s;
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- per-agent epilogue
    if (tid < n_real) {
        const int row = row0 + tid;
        const size_t si = base_row + tid;
        const int ty = rtype[tid];
        const bool valid_old = rowvalid[tid] != 0;
        const bool have = valid_old && ty >= 0;
        f32x4 st = ldg4(p.state + si * 4);
        // Dynamics.update + MultiPathPP (dynamics.py:74-119,194-228); tanh-bounded action, midpoint unicycle
        float acc_ = 0.f, yr_ = 0.f;
        if (have) {
            acc_ = fmul_(tanhf(ubuf[tid * 2 + 0]), pw.max_acc[ty]);
            yr_ = fmul_(tanhf(ubuf[tid * 2 + 1]), pw.max_yaw_rate[ty]);
        }
        const float half_dt = 0.5f * pw.dt;  // python: 0.5 * self.dt, then cast with the tensor op
        const float v_t = fadd_(st.w, fmul_(half_dt, acc_));
        const float th_t = fadd_(st.z, fmul_(half_dt, yr_));
        float sn, cs;
        sincosf(th_t, &sn, &cs);
        f32x4 pred;
        pred.x = fadd_(st.x, fmul_(pw.dt, fmul_(v_t, cs)));
        pred.y = fadd_(st.y, fmul_(pw.dt, fmul_(v_t, sn)));
        pred.z = fadd_(st.z, fmul_(pw.dt, yr_));
        pred.w = fadd_(st.w, fmul_(pw.dt, acc_));
        if (!have) pred = splat(0.f);
        float alp = 0.f;
        if (valid_old) {
            for (int d = 0; d < 2; ++d) {
                const float ls = (ty >= 0) ? W[pw.head_log_std[ty] + d] : 0.f;
                alp += -logf(expf(ls)) - 0.9189385332046727f;
            }
        }
        // teacher forcing / spawn (dynamics.py:132-149)
        f32x4 cur = pred;
        bool valid = valid_old;
        bool killed = p.killed[si] != 0;
        uint8_t ovr = 0;
        bool gt_valid = false;
        if (t < p.n_hist) {
            const size_t hi = ((size_t)b * p.n_hist + t) * p.n_agent + row;
            ovr = p.tf_mask[hi];
            gt_valid = p.hist_valid[hi] != 0;
            if (ovr && !killed) {
                valid = true;
                cur = ldg4(p.hist_state + hi * 4);
                st4(p.aux + si * 4, f32x4{p.hist_vel[hi * 2], p.hist_vel[hi * 2 + 1], p.hist_acc[hi], p.hist_yaw_rate[hi]});
            }
        }
        // rule checks on the post-override state (traffic_rule_checker.py:101-119,364-410)
        const float* bd = p.map_boundary + (size_t)b * 4;
        const bool out_this = valid && ((cur.x > bd[1]) || (cur.x < bd[0]) || (cur.y > bd[3]) || (cur.y < bd[2]));
        bool outside = (p.outside[si] != 0) || out_this;
        bool dreached = p.dest_reached[si] != 0;
        bool dr_this = false;
        {
            const int d = p.dest[(size_t)n * p.n_agent + row];
            const int dty = p.map_type[(size_t)b * p.n_pl + d];
            const bool is_edge = dty == 4, is_lane = dty >= 0 && dty < 4;
            const float thresh = is_edge ? fmul_(50.f, fadd_(1.f, -0.8f)) : 50.f;
            float hs, hc;
            sincosf(cur.z, &hs, &hc);
            bool pos_r = false, rot_r = false;
            const size_t nb = ((size_t)b * p.n_pl + d) * 20;
            for (int k = 0; k < 20; ++k) {
                if (!p.map_valid[nb + k]) continue;
                const float dx = fadd_(cur.x, -p.map_pos[(nb + k) * 2]), dy = fadd_(cur.y, -p.map_pos[(nb + k) * 2 + 1]);
                const float dist = sqrtf(fadd_(fmul_(dx, dx), fmul_(dy, dy)));
                pos_r |= dist < thresh;
                const float ddx = p.map_dir[(nb + k) * 2], ddy = p.map_dir[(nb + k) * 2 + 1];
                const float nrm = sqrtf(fadd_(fmul_(ddx, ddx), fmul_(ddy, ddy)));
                const float rot = fadd_(fmul_(hc, ddx / nrm), fmul_(hs, ddy / nrm));
                rot_r |= rot > 0.8660254037844387f;
            }
            dr_this = !dreached && valid && ((is_lane && pos_r && rot_r) || (is_edge && pos_r));
            dreached |= dr_this;
        }
        // kill agents that left the map unless ground truth is still valid (dynamics.py:161-167)
        const bool mk = out_this && !gt_valid;
        killed |= mk;
        valid = valid && !mk;
        // navigator (goal_manager.py:155-162)
        const bool gv = (gvalid[tid] != 0) && valid && !dreached;
        // write simulator state
        st4(p.state + si * 4, cur);
        p.valid[si] = valid;
        p.killed[si] = killed;
        p.goal_valid[si] = gv;
        p.dest_reached[si] = dreached;
        p.outside[si] = outside;
        // RolloutBuffer.add (buffer.py:39-70)
        const int s = t - p.step_start;
        const size_t oi = ((size_t)n * p.n_agent + row) * p.n_step_out + s;
        st4(p.preds + oi * 4, pred);
        p.o_valid[oi] = valid_old;
        p.o_override[oi] = ovr;
        p.o_outside[oi] = outside;
        p.o_outside_this[oi] = out_this;
        p.o_dest_reached[oi] = dreached;
        p.o_dest_reached_this[oi] = dr_this;
        p.o_action_logp[oi] = alp;
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
size_t step_lds_bytes() { return (size_t)STEP_LDS_FLOATS * sizeof(float); }

void launch_kv_hoist(const float* W, const XLayerW* L3, const float* feat, const uint8_t* fvalid, int G, int n_tok, int n_pad,
                     float* K, float* VT, uint8_t* kvalid, hipStream_t s) {
    dim3 grid(n_pad / TM, G);
    hipLaunchKernelGGL(k_kv_hoist, grid, dim3(NTHREADS), 2 * TM * LDT * sizeof(float), s, W, L3[0], L3[1], L3[2], feat, fvalid,
                       n_tok, n_pad, K, VT, kvalid);
}

void launch_rollout_init(const RolloutP& p, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_rollout_init, grid, dim3(NTHREADS), step_lds_bytes(), s, p);
}

void launch_step_a(const RolloutP& p, int t, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_step_a, grid, dim3(NTHREADS), step_lds_bytes(), s, p, t);
}

void launch_step_c(const RolloutP& p, int t, hipStream_t s) {
    dim3 grid(p.a_pad / TM, p.n_inst);
    hipLaunchKernelGGL(k_step_c, grid, dim3(NTHREADS), step_lds_bytes(), s, p, t);
}

}  // namespace tb
