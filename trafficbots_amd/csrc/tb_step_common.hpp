// Pieces shared by the step kernels (k_step_x and its fp32-MFMA twin k_step): LDS carve, per-row state,
// the per-agent simulator epilogue and the agent input encoding at the head of the A half.
#pragma once
#include "tb_rollout.hpp"

namespace tb {

// LDS carve for the step kernels (floats)
constexpr int OFF_X = 0;                   // [16][LDT] residual stream
constexpr int OFF_S1 = OFF_X + TM * LDT;   // [16][LDT]
constexpr int OFF_S2 = OFF_S1 + TM * LDT;  // [16][LDT]
constexpr int OFF_H = OFF_S2 + TM * LDT;   // [16][LDT] GRU previous hidden
constexpr int OFF_Y = OFF_H + TM * LDT;    // [16][LDT] GRU out ping
constexpr int OFF_CAT = OFF_Y + TM * LDT;  // [16][LDC] concat tile
constexpr int OFF_H1 = OFF_CAT + TM * LDC; // [16][LDT] GRU hidden layer 1
constexpr int OFF_H2 = OFF_H1 + TM * LDT;  // [16][LDT] GRU hidden layer 2
constexpr int OFF_GP = OFF_H2 + TM * LDT;  // [16][LDT] add_goal.mlp_in output of the tile's agents
constexpr int OFF_LP = OFF_GP + TM * LDT;  // [16][LDT] add_latent.mlp_in output
constexpr int OFF_DG = OFF_LP + TM * LDT;  // [16][80]  destination polyline geometry (20 nodes x px,py,dx,dy)
constexpr int OFF_LN = OFF_DG + TM * 80;   // [9][768] LayerNorm parameter blocks of inter / as2pl / as2tl layers
constexpr int OFF_SMALL = OFF_LN + 9 * 768;
constexpr int SMALL_FLOATS = 16 * 16 /*attr*/ + 16 * 32 /*enc hidden*/ + 16 * 8 /*row state*/ + 32 /*u*/ + 64 /*flags,types*/;
constexpr int STEP_LDS_FLOATS = OFF_SMALL + SMALL_FLOATS;

#ifdef TB_PROFILE
#define TB_STAMP(i)                                                                                        \
    do {                                                                                                   \
        if (threadIdx.x == 0) p.prof[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 32 + (i)] = clock64(); \
    } while (0)
#else
#define TB_STAMP(i) \
    do {            \
    } while (0)
#endif

// Kernel entry (round 5): the step kernels take ~2 KB of arguments, and the compiler loads each field where it is first used -- a
// chain of `s_load; s_waitcnt lgkmcnt(0)` pairs, each one a miss of the (cold) scalar cache, ~600 cycles apiece, seven of them in
// series in front of the first weight request alone.  One scalar load per 64-byte line of the argument segment, all in flight
// together, turns the chain into one miss followed by hits.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    typedef const __attribute__((address_space(4))) unsigned int* kptr_t;
    kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned int acc = 0u;
#pragma unroll
    for (int o = 0; o < BYTES / 4; o += 16) acc |= ka[o];
    asm volatile("" ::"s"(acc));
}
// a zero the compiler cannot see through: `ptr[i + vzero()]` is a VECTOR load (issued with the burst, waited for with it) where
// `ptr[i]` with a wave-uniform i would be a scalar load that stalls the wave on a cold line at its first use
__device__ __forceinline__ int vzero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

__device__ __forceinline__ float fmul_(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd_(float a, float b) { return __fadd_rn(a, b); }

// per-row simulator state carried in LDS between the C and the A half of a launch
struct RowSt {
    float st[4];   // x, y, yaw, spd
    float aux[4];  // vel_x, vel_y, acc, yaw_rate as last teacher-forced
};

// pointers into the SMALL region of the LDS carve
struct StepSmall {
    float* attr;         // [16][16]
    float* ench;         // [16][32]
    RowSt* rst;          // [16]
    float* ubuf;         // [16][2] action means
    uint8_t* rowvalid;   // [16]
    uint8_t* novalid_s;  // [16]
    uint8_t* gvalid;     // [16]
    int* rtype;          // [16]
    int* dflag;          // [16] bit0 lane-type destination, bit1 road-edge destination
};

__device__ __forceinline__ StepSmall step_small(float* small_base) {
    StepSmall s;
    s.attr = small_base;
    s.ench = s.attr + 16 * 16;
    s.rst = reinterpret_cast<RowSt*>(s.ench + 16 * 32);
    s.ubuf = reinterpret_cast<float*>(s.rst + 16);
    s.rowvalid = reinterpret_cast<uint8_t*>(s.ubuf + 32);
    s.novalid_s = s.rowvalid + 16;
    s.gvalid = s.rowvalid + 32;
    s.rtype = reinterpret_cast<int*>(s.rowvalid + 48);
    s.dflag = s.rtype + 16;
    return s;
}

// XCD-aware tile -> workgroup map (speed only): the dispatcher places linear workgroup L on XCD L % 8; give all row
// tiles of an instance the same L % 8 so its map / TL / interaction K,V are fetched into ONE XCD's L2 instead of up to 4
__device__ __forceinline__ void step_tile_map(int& n, int& rt) {
    const int T = gridDim.x, N = gridDim.y, L = blockIdx.y * T + blockIdx.x;
    if ((N & 7) == 0) {
        const int q = L >> 3;
        rt = q % T;
        n = (L & 7) + 8 * (q / T);
    } else {
        rt = blockIdx.x;
        n = blockIdx.y;
    }
}

// ---- the per-tile inputs of the C half (x_mid, three GRU hidden tiles, goal / latent pre-activations, destination
// geometry + flags) as ONE burst: every global load is issued before the first LDS store, so the launch pays one cold
// round trip here instead of one per tile (the compiler otherwise reuses one register quad and waits after each load).
template <int NT>
struct CInputs {
    static constexpr int PER = (TM * 32) / NT;  // float4 per thread per tile: 2 (256 threads) or 1 (512)
    f32x4 v[6][PER], g[2];
    int df;
};

// NSRC = 4: without the goal / latent pre-activation tiles (the LEAN carve of k_step_x reads them where they are used)
template <int NT, int NSRC = 6, bool DG_LDS = true>
__device__ __forceinline__ void c_inputs_issue(const RolloutP& p, int n, int row0, int tid, CInputs<NT>& c) {
    const size_t base_row = (size_t)n * p.a_pad + row0;
    const size_t x_row = p.pre_shared ? (size_t)(n - n % p.k_rep) * p.a_pad + row0 : base_row;  // (RolloutP::pre_shared)
    const float* src[6] = {p.x_mid + x_row * H,
                           p.hidden + (((size_t)0 * p.n_inst + n) * p.a_pad + row0) * H,
                           p.hidden + (((size_t)1 * p.n_inst + n) * p.a_pad + row0) * H,
                           p.hidden + (((size_t)2 * p.n_inst + n) * p.a_pad + row0) * H,
                           p.goal_pre + base_row * H,
                           p.lat_pre + base_row * H};
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int i = 0; i < CInputs<NT>::PER; ++i) {
            const int idx = tid + i * NT;
            c.v[s][i] = ldg4(src[s] + (size_t)(idx >> 5) * H + (idx & 31) * 4);
        }
    // (unconditional loads at clamped indices: a conditional load merges with a constant behind the branch, and the compiler
    // settles that merge with `s_waitcnt vmcnt` + a register copy IN the burst -- a full cold round trip per merge, round 5)
    if (DG_LDS) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = min(tid + i * NT, TM * 20 - 1);
            c.g[i] = ldg4(p.dest_geo + (base_row * 20 + idx) * 4);
        }
    }
    c.df = p.dest_flag[base_row + (tid & (TM - 1))];
}

template <int NT, int NSRC = 6, bool DG_LDS = true>
__device__ __forceinline__ void c_inputs_commit(int tid, const CInputs<NT>& c, float* X, float* Hs, float* H1, float* H2, float* GP,
                                                float* LP, float* DG, int* dflag) {
    float* dst[6] = {X, Hs, H1, H2, GP, LP};
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int i = 0; i < CInputs<NT>::PER; ++i) {
            const int idx = tid + i * NT;
            st4(dst[s] + (idx >> 5) * LDT + (idx & 31) * 4, c.v[s][i]);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NT;
        if (DG_LDS && idx < TM * 20) st4(DG + idx * 4, c.g[i]);
    }
    if (tid < TM) dflag[tid] = c.df;
}

template <int NT>
__device__ __forceinline__ void step_load_c_inputs(const RolloutP& p, int n, int row0, int tid, float* X, float* Hs, float* H1,
                                                   float* H2, float* GP, float* LP, float* DG, int* dflag) {
    CInputs<NT> c;
    TB_SCHED_FENCE();
    c_inputs_issue<NT>(p, n, row0, tid, c);
    TB_SCHED_FENCE();
    c_inputs_commit<NT>(tid, c, X, Hs, H1, H2, GP, LP, DG, dflag);
}

// ---- per-agent simulator epilogue of C(t): dynamics, teacher forcing, rule checks, kill, navigator, buffer writes.
// Thread `tid` < n_real handles row `tid` of the tile; the caller places barriers around it.
__device__ __forceinline__ void step_epilogue(const RolloutP& p, int t, int n, int b, int row0, int n_real, int tid,
                                              const StepSmall& sm, const float* DG) {
    const PolicyW& pw = p.pw;
    const float* W = p.W;
    const size_t base_row = (size_t)n * p.a_pad + row0;
    RowSt* rst = sm.rst; float* ubuf = sm.ubuf; uint8_t* rowvalid = sm.rowvalid; uint8_t* gvalid = sm.gvalid;
    const int* rtype = sm.rtype; const int* dflag = sm.dflag;
    if (tid < n_real) {
        const int row = row0 + tid;
        const size_t si = base_row + tid;
        const int ty = rtype[tid];
        const bool valid_old = rowvalid[tid] != 0;
        const bool have = valid_old && ty >= 0;
        const f32x4 st = f32x4{rst[tid].st[0], rst[tid].st[1], rst[tid].st[2], rst[tid].st[3]};
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
                const f32x4 ax = f32x4{p.hist_vel[hi * 2], p.hist_vel[hi * 2 + 1], p.hist_acc[hi], p.hist_yaw_rate[hi]};
                st4(p.aux + si * 4, ax);
                rst[tid].aux[0] = ax.x; rst[tid].aux[1] = ax.y; rst[tid].aux[2] = ax.z; rst[tid].aux[3] = ax.w;
            }
        }
        // rule checks on the post-override state (traffic_rule_checker.py:101-119,364-410)
        const float* bd = p.map_boundary + (size_t)b * 4;
        const bool out_this = valid && ((cur.x > bd[1]) || (cur.x < bd[0]) || (cur.y > bd[3]) || (cur.y < bd[2]));
        const bool outside = (p.outside[si] != 0) || out_this;
        bool dreached = p.dest_reached[si] != 0;
        bool dr_this = false;
        {
            const bool is_lane = (dflag[tid] & 1) != 0, is_edge = (dflag[tid] & 2) != 0;
            const float thresh = is_edge ? fmul_(50.f, fadd_(1.f, -0.8f)) : 50.f;
            float hs, hc;
            sincosf(cur.z, &hs, &hc);
            bool pos_r = false, rot_r = false;
#pragma unroll 4
            for (int k = 0; k < 20; ++k) {
                const f32x4 g = lds4(DG + (tid * 20 + k) * 4);
                const float dx = fadd_(cur.x, -g.x), dy = fadd_(cur.y, -g.y);
                const float dist = sqrtf(fadd_(fmul_(dx, dx), fmul_(dy, dy)));
                pos_r |= dist < thresh;
                const float rot = fadd_(fmul_(hc, g.z), fmul_(hs, g.w));
                rot_r |= rot > 0.8660254037844387f;
            }
            dr_this = !dreached && valid && ((is_lane && pos_r && rot_r) || (is_edge && pos_r));
            dreached |= dr_this;
        }
        // kill agents that left the map unless ground truth is still valid (dynamics.py:161-167)
        const bool mk = out_this && !gt_valid;
        killed |= mk;
        if (p.o_check_state) {  // what TrafficRuleChecker.check sees (waymo_motion.py:311): post-override, pre-kill
            const size_t ci = ((size_t)n * p.n_agent + row) * p.n_step_out + (t - p.step_start);
            st4(p.o_check_state + ci * 4, cur);
            p.o_check_valid[ci] = valid;
        }
        valid = valid && !mk;
        // navigator (goal_manager.py:155-162)
        const bool gv = (gvalid[tid] != 0) && valid && !dreached;
        // simulator state: global (the next launch's interaction reads `valid` of every agent) + LDS (A half below)
        st4(p.state + si * 4, cur);
        p.valid_w[si] = valid;
        p.vbias_w[si] = valid ? 0.f : -INFINITY;
        p.killed[si] = killed;
        p.goal_valid[si] = gv;
        p.dest_reached[si] = dreached;
        p.outside[si] = outside;
        rst[tid].st[0] = cur.x; rst[tid].st[1] = cur.y; rst[tid].st[2] = cur.z; rst[tid].st[3] = cur.w;
        rowvalid[tid] = valid;
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
        if (p.o_action) {  // vis_dict["action"] (waymo_motion.py:191-194): the physical action applied this step, 0 for invalid agents
            p.o_action[oi * 2 + 0] = acc_;
            p.o_action[oi * 2 + 1] = yr_;
        }
    }
}

// ---- the same epilogue spread over the whole 256-thread workgroup: 16 lanes per agent.  Every lane of a group evaluates the
// agent's dynamics (uniform inside the group, free on a SIMD), the 20 destination nodes are split over the lanes and
// OR-reduced with a ballot, lane 0 of the group writes.  All global loads are issued up front (clamped indices where the
// reference reads nothing), so the stage pays one memory round trip instead of a chain of dependent ones.
// ---- inputs of the epilogue (teacher-forcing source, sticky flags, map boundary, action log-std) and of the A half's attribute
// stage (agent size) that do not depend on the step's own work: requested in the launch prologue's burst by threads 16 .. 48 and
// parked in the LDS regions those stages only write later (attr: the epilogue's per-agent record, ench: sizes + tile constants),
// so neither stage opens with a dependent round trip to memory.  (k_step_x; the fp32-MFMA twin keeps its own loads.)
struct EpiRegs {  // raw load results (wave 0 only; nothing is computed from them before epi_commit)
    f32x4 hst;
    float vel[2], acc, yr, ae[2], ao[2], sz[3], ls[3], bd;
    unsigned int k0, o0, d0, m0, g0, am0;  // bytes
};
constexpr int EPI_FLAG_KILLED = 1, EPI_FLAG_OUTSIDE = 2, EPI_FLAG_DREACHED = 4, EPI_FLAG_OVR = 8, EPI_FLAG_GTV = 16, EPI_FLAG_AOVR = 32;
constexpr int EPI_POISON_WORD = 17;  // StepSmall::dflag[17]: a helper hand-off timed out in this launch (kv_wait_x) -> NaN out
constexpr int EPI_ENCH_SIZE = 0, EPI_ENCH_BD = 48, EPI_ENCH_LS = 52;  // offsets (floats) inside StepSmall::ench

struct TfSource {
    const uint8_t* mask;
    const uint8_t* gtv;
    const float* state;
    const float* vel;
    const float* acc;
    const float* yr;
    bool in_hist;
    size_t hi;
};
// teacher-forcing source of step t: step t of the per-scene history arrays, or the caller's per-instance arrays of a
// tb_rollout_step_ex call (wave-uniform select of base pointers and index)
__device__ __forceinline__ TfSource tf_source(const RolloutP& p, int t, int n, int b, int rowc) {
    TfSource f;
    const bool per_call = p.ovr_mask != nullptr;
    f.in_hist = per_call || t < p.n_hist;
    f.hi = per_call ? (size_t)n * p.n_agent + rowc : ((size_t)b * p.n_hist + min(t, p.n_hist - 1)) * p.n_agent + rowc;
    f.mask = per_call ? p.ovr_mask : p.tf_mask;
    f.gtv = per_call ? (p.ovr_gt_valid ? p.ovr_gt_valid : nullptr) : p.hist_valid;
    f.state = per_call ? p.ovr_state : p.hist_state;
    f.vel = per_call ? p.ovr_vel : p.hist_vel;
    f.acc = per_call ? p.ovr_acc : p.hist_acc;
    f.yr = per_call ? p.ovr_yaw_rate : p.hist_yaw_rate;
    return f;
}

// wave 0 -- lanes 16 .. 31: the epilogue record of agent lane - 16; lanes 32 .. 47: the size of agent lane - 32; lanes 48 .. 53: the six
// action log-stds; lanes 56 .. 59: the map boundary.  EVERY lane of the wave executes EVERY load at a clamped address of its own
// (a = lane & 15), so the burst holds no branch, no merge with a constant and no scalar load from a cold line (a wave-uniform
// address -- the boundary, a log-std -- would become an s_load and stall the wave at its first use); optional arrays that are
// absent are replaced by a buffer that holds the index, their values dropped in epi_commit.
__device__ __forceinline__ void epi_issue(const RolloutP& p, int t, int n, int b, int row0, int tid, bool do_c, EpiRegs& e) {
    const int a = tid & 15, rowc = min(row0 + a, p.n_agent - 1);
    {
        const float* sp = p.agent_size + ((size_t)b * p.n_agent + rowc) * 3;
        e.sz[0] = sp[0]; e.sz[1] = sp[1]; e.sz[2] = sp[2];
    }
    const size_t si = (size_t)n * p.a_pad + row0 + a;
    const TfSource f = tf_source(p, t, n, b, rowc);
    e.hst = ldg4(f.state + f.hi * 4);
    e.vel[0] = f.vel[f.hi * 2]; e.vel[1] = f.vel[f.hi * 2 + 1]; e.acc = f.acc[f.hi]; e.yr = f.yr[f.hi];
    {   // [N,A,S,2] draws; absent: `preds` ([N,A,S,4]) holds the index.  The A-only launch of tb_rollout_begin runs at t = step_start - 1
        // (its record is dropped, do_c = false): without the clamp lane 0 of instance 0 read 8 bytes in FRONT of the array -- a memory
        // fault whenever the array opens an allocation (found in round 5 when a freed workspace moved torch's segments)
#ifdef TB_DBG_OOB_STEP_INDEX  // (test builds only: the bug as it was, to show that tests/probes/gpu_guard_pages.py catches it)
        const float* ap = (p.action_eps ? p.action_eps : p.preds) + (((size_t)n * p.n_agent + rowc) * p.n_step_out + (t - p.step_start)) * 2;
#else
        const float* ap = (p.action_eps ? p.action_eps : p.preds) + (((size_t)n * p.n_agent + rowc) * p.n_step_out + max(t - p.step_start, 0)) * 2;
#endif
        e.ae[0] = ap[0]; e.ae[1] = ap[1];
    }
    const size_t ai = (size_t)n * p.n_agent + rowc;
    {   // per-call action override; absent: `state` ([N,a_pad,4]) holds the index
        const float* ao = (p.ovr_action_mask ? p.ovr_action : p.state) + ai * 2;
        e.ao[0] = ao[0]; e.ao[1] = ao[1];
    }
    // (three loads, one per type, picked by lane in epi_commit: an offset table indexed by lane would be read from the argument
    // segment with a vector load, and the log-std would hang on it)
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) e.ls[ty] = p.W[p.pw.head_log_std[ty] + (tid & 1)];
    e.bd = p.map_boundary[(size_t)b * 4 + (tid & 3)];
    // the byte loads LAST: the compiler moves their zero-extension up to the load -- behind a wait for it, which at the end of the
    // burst costs nothing
    e.k0 = p.killed[si]; e.o0 = p.outside[si]; e.d0 = p.dest_reached[si]; e.m0 = f.mask[f.hi];
    e.g0 = (f.gtv ? f.gtv : f.mask)[f.hi];
    e.am0 = (p.ovr_action_mask ? p.ovr_action_mask : p.valid)[ai];  // (absent: `valid`, [N,a_pad] bytes, holds the index)
}
// the value as the compiler may not look through it: a compare or a mask that is the ONLY use of a loaded value is otherwise moved up
// to the load -- into the burst, behind a wait for it
__device__ __forceinline__ unsigned int pin_v(unsigned int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float pin_f(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ void epi_commit(const RolloutP& p, int n_real, int tid, bool do_c, const EpiRegs& e, const StepSmall& sm) {
    if (do_c && tid >= 16 && tid < 32) {
        float* r = sm.attr + (tid - 16) * 16;
        const bool no_gtv = p.ovr_mask != nullptr && p.ovr_gt_valid == nullptr;  // (tf_source: f.gtv == nullptr)
        const bool has_ae = p.action_eps != nullptr, has_ao = p.ovr_action_mask != nullptr;
        st4(r, e.hst);
        st4(r + 4, f32x4{e.vel[0], e.vel[1], e.acc, e.yr});
        reinterpret_cast<int*>(r)[8] = (pin_v(e.k0) ? EPI_FLAG_KILLED : 0) | (pin_v(e.o0) ? EPI_FLAG_OUTSIDE : 0) |
                                       (pin_v(e.d0) ? EPI_FLAG_DREACHED : 0) | (pin_v(e.m0) ? EPI_FLAG_OVR : 0) |
                                       ((!no_gtv && pin_v(e.g0)) ? EPI_FLAG_GTV : 0) | ((has_ao && pin_v(e.am0)) ? EPI_FLAG_AOVR : 0);
        r[9] = has_ae ? e.ae[0] : 0.f;
        r[10] = has_ae ? e.ae[1] : 0.f;
        r[11] = has_ao ? e.ao[0] : 0.f;
        r[12] = has_ao ? e.ao[1] : 0.f;
    } else if (tid >= 32 && tid < 48) {
        float* r = sm.ench + EPI_ENCH_SIZE + (tid - 32) * 3;
        const bool real = tid - 32 < n_real;  // (rows past n_real: zeros)
        r[0] = real ? e.sz[0] : 0.f; r[1] = real ? e.sz[1] : 0.f; r[2] = real ? e.sz[2] : 0.f;
    } else if (do_c && tid >= 48 && tid < 54) {  // (lane parity = the component the lane loaded)
        // (pinned: a select among the members of one object is otherwise rewritten as an indexed read of the object, which then lives
        // in scratch memory)
        const int ty = (tid - 48) >> 1;
        const float l0 = pin_f(e.ls[0]), l1 = pin_f(e.ls[1]), l2 = pin_f(e.ls[2]);
        sm.ench[EPI_ENCH_LS + (tid - 48)] = ty == 0 ? l0 : (ty == 1 ? l1 : l2);
    } else if (do_c && tid >= 56 && tid < 60) {
        sm.ench[EPI_ENCH_BD + (tid - 56)] = e.bd;
    }
}

// DGG: `DG` points at the tile's destination geometry in the rollout workspace (global) instead of its LDS copy
template <bool PREF = false, bool DGG = false>
__device__ __forceinline__ void step_epilogue16(const RolloutP& p, int t, int n, int b, int row0, int n_real, int tid,
                                                const StepSmall& sm, const float* DG) {
    const PolicyW& pw = p.pw;
    const float* W = p.W;
    RowSt* rst = sm.rst; float* ubuf = sm.ubuf; uint8_t* rowvalid = sm.rowvalid; uint8_t* gvalid = sm.gvalid;
    const int* rtype = sm.rtype; const int* dflag = sm.dflag;
    const int a = tid >> 4, sub = tid & 15, lane = tid & 63;
    const bool real = a < n_real;
    const int row = row0 + a;
    const int rowc = min(row, p.n_agent - 1);
    const size_t si = (size_t)n * p.a_pad + row;
    const TfSource tf = tf_source(p, t, n, b, rowc);
    const bool in_hist = tf.in_hist;
    // ---- loads (PREF: the launch prologue fetched them, epi_issue / epi_commit)
    uint8_t killed0, outside0, dreached0, ovr0, gtv0;
    f32x4 hst, hax, bd;
    float ae0 = 0.f, ae1 = 0.f;  // standard-normal draws of a sampled action (RolloutP::action_eps)
    float ao0 = 0.f, ao1 = 0.f;  // action override in physical units (RolloutP::ovr_action), applied where aovr
    bool aovr = false;
    if (PREF) {
        const float* r = sm.attr + a * 16;
        ae0 = r[9];
        ae1 = r[10];
        ao0 = r[11];
        ao1 = r[12];
        const int fl = reinterpret_cast<const int*>(r)[8];
        aovr = (fl & EPI_FLAG_AOVR) != 0;
        killed0 = (fl & EPI_FLAG_KILLED) != 0; outside0 = (fl & EPI_FLAG_OUTSIDE) != 0; dreached0 = (fl & EPI_FLAG_DREACHED) != 0;
        ovr0 = (fl & EPI_FLAG_OVR) != 0; gtv0 = (fl & EPI_FLAG_GTV) != 0;
        hst = lds4(r);
        hax = lds4(r + 4);
        bd = lds4(sm.ench + EPI_ENCH_BD);
    } else {
        const size_t hi = tf.hi;
        killed0 = p.killed[si]; outside0 = p.outside[si]; dreached0 = p.dest_reached[si];
        ovr0 = tf.mask[hi]; gtv0 = tf.gtv ? tf.gtv[hi] : (uint8_t)0;
        hst = ldg4(tf.state + hi * 4);
        hax = f32x4{tf.vel[hi * 2], tf.vel[hi * 2 + 1], tf.acc[hi], tf.yr[hi]};
        bd = ldg4(p.map_boundary + (size_t)b * 4);
        if (p.action_eps) {
            const float* ap = p.action_eps + (((size_t)n * p.n_agent + rowc) * p.n_step_out + (t - p.step_start)) * 2;
            ae0 = ap[0];
            ae1 = ap[1];
        }
        if (p.ovr_action_mask) {
            const size_t ai = (size_t)n * p.n_agent + rowc;
            aovr = p.ovr_action_mask[ai] != 0;
            ao0 = p.ovr_action[ai * 2];
            ao1 = p.ovr_action[ai * 2 + 1];
        }
    }
    const int ty = rtype[a];
    const bool valid_old = rowvalid[a] != 0;
    const bool have = valid_old && ty >= 0;
    const f32x4 st = f32x4{rst[a].st[0], rst[a].st[1], rst[a].st[2], rst[a].st[3]};
    const float u0 = ubuf[a * 2 + 0], u1 = ubuf[a * 2 + 1];
    const bool gv0 = gvalid[a] != 0;
    const int dfl = dflag[a];
    // ---- Dynamics.update + MultiPathPP (dynamics.py:74-119,194-228); tanh-bounded action, midpoint unicycle
    // action_dist.sample(deterministic) (dynamics.py:77): the mean, or mean + eps * exp(log_std) (Normal.rsample) with the per-type
    // log_std under (type & valid), 0 elsewhere (action_head.py:81-87)
    const bool sampled = p.action_eps != nullptr;
    float lsd[2], us[2] = {u0, u1};
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        lsd[d] = (valid_old && ty >= 0) ? (PREF ? sm.ench[EPI_ENCH_LS + ty * 2 + d] : W[pw.head_log_std[ty] + d]) : 0.f;
        if (sampled) us[d] = fadd_(us[d], fmul_(d == 0 ? ae0 : ae1, expf(lsd[d])));
    }
    float acc_ = 0.f, yr_ = 0.f;
    if (have) {
        acc_ = fmul_(tanhf(us[0]), pw.max_acc[ty]);
        yr_ = fmul_(tanhf(us[1]), pw.max_yaw_rate[ty]);
        // action_override (dynamics.py:96-100): where (mask & agent_valid) the physical action [acc m/s^2, yaw rate rad/s] replaces the
        // policy's; action_log_prob stays that of the policy's own sample
        if (aovr) {
            acc_ = ao0;
            yr_ = ao1;
        }
    }
    const float half_dt = 0.5f * pw.dt;
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
    if (dflag[EPI_POISON_WORD]) pred = splat(__builtin_nanf(""));  // stale interaction K / V were used: fail loudly, not plausibly
    // action_dist.log_prob(sample), masked to 0 for invalid agents (dynamics.py:80; torch Normal.log_prob:
    // -((x - mu)^2) / (2 var) - log(scale) - log(sqrt(2 pi)), summed over the two dims)
    float alp = 0.f;
    if (valid_old) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float sc = expf(lsd[d]);
            if (sampled) {
                const float dv = fadd_(us[d], -(d == 0 ? u0 : u1));
                alp += fadd_(fadd_(-fmul_(dv, dv) / fmul_(2.f, fmul_(sc, sc)), -logf(sc)), -0.9189385332046727f);
            } else {
                alp += -logf(sc) - 0.9189385332046727f;
            }
        }
    }
    // ---- teacher forcing / spawn (dynamics.py:132-149)
    f32x4 cur = pred;
    bool valid = valid_old;
    bool killed = killed0 != 0;
    const uint8_t ovr = in_hist ? ovr0 : 0;
    const bool gt_valid = in_hist && gtv0 != 0;
    const bool forced = ovr && !killed;
    if (forced) {
        valid = true;
        cur = hst;
    }
    // ---- rule checks on the post-override state (traffic_rule_checker.py:101-119,364-410)
    const bool out_this = valid && ((cur.x > bd.y) || (cur.x < bd.x) || (cur.y > bd.w) || (cur.y < bd.z));
    const bool outside = (outside0 != 0) || out_this;
    bool dreached = dreached0 != 0;
    bool dr_this;
    {
        const bool is_lane = (dfl & 1) != 0, is_edge = (dfl & 2) != 0;
        const float thresh = is_edge ? fmul_(50.f, fadd_(1.f, -0.8f)) : 50.f;
        float hs, hc;
        sincosf(cur.z, &hs, &hc);
        bool pos = false, rot = false;
#pragma unroll
        for (int k = sub; k < 20; k += 16) {
            const f32x4 g = DGG ? ldg4(DG + (a * 20 + k) * 4) : lds4(DG + (a * 20 + k) * 4);
            const float dx = fadd_(cur.x, -g.x), dy = fadd_(cur.y, -g.y);
            const float dist = sqrtf(fadd_(fmul_(dx, dx), fmul_(dy, dy)));
            pos |= dist < thresh;
            rot |= fadd_(fmul_(hc, g.z), fmul_(hs, g.w)) > 0.8660254037844387f;
        }
        const int sh = (lane >> 4) * 16;
        const bool pos_r = ((__ballot(pos) >> sh) & 0xFFFFull) != 0;
        const bool rot_r = ((__ballot(rot) >> sh) & 0xFFFFull) != 0;
        dr_this = !dreached && valid && ((is_lane && pos_r && rot_r) || (is_edge && pos_r));
        dreached |= dr_this;
    }
    // kill agents that left the map unless ground truth is still valid (dynamics.py:161-167); navigator (goal_manager.py:155-162)
    const bool mk = out_this && !gt_valid;
    killed |= mk;
    const bool valid_checked = valid;  // what TrafficRuleChecker.check sees (waymo_motion.py:311): post-override, pre-kill
    valid = valid && !mk;
    const bool gv = gv0 && valid && !dreached;
    if (sub == 0 && real) {
        if (p.o_check_state) {
            const size_t ci = ((size_t)n * p.n_agent + row) * p.n_step_out + (t - p.step_start);
            st4(p.o_check_state + ci * 4, cur);
            p.o_check_valid[ci] = valid_checked;
        }
        if (forced) {
            st4(p.aux + si * 4, hax);
            rst[a].aux[0] = hax.x; rst[a].aux[1] = hax.y; rst[a].aux[2] = hax.z; rst[a].aux[3] = hax.w;
        }
        st4(p.state + si * 4, cur);
        p.valid_w[si] = valid;
        p.vbias_w[si] = valid ? 0.f : -INFINITY;
        p.killed[si] = killed;
        p.goal_valid[si] = gv;
        p.dest_reached[si] = dreached;
        p.outside[si] = outside;
        rst[a].st[0] = cur.x; rst[a].st[1] = cur.y; rst[a].st[2] = cur.z; rst[a].st[3] = cur.w;
        rowvalid[a] = valid;
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
        if (p.o_action) {  // vis_dict["action"] (waymo_motion.py:191-194): the physical action applied this step, 0 for invalid agents
            p.o_action[oi * 2 + 0] = acc_;
            p.o_action[oi * 2 + 1] = yr_;
        }
    }
}

// ---- head of A(t+1): agent attributes + pose PE + InputPeEncoder -> X[:, 0:128] (invalid rows zeroed).
// The row/column maps use 256 threads; extra threads of a wider workgroup only take part in the barriers.
template <int NT>
__device__ __forceinline__ void step_encode_inputs(const RolloutP& p, int t, int n, int b, int row0, int n_real, int tid,
                                                   const StepSmall& sm, float* X) {
    (void)t; (void)n;
    const PolicyW& pw = p.pw;
    const float* W = p.W;
    float* attr = sm.attr; float* ench = sm.ench; const RowSt* rst = sm.rst; const uint8_t* rowvalid = sm.rowvalid;
    const int* rtype = sm.rtype;
    const bool act = (NT == 256) || tid < 256;
    // ---- agent attributes (sc_input.py:142-165): vel2, spd, yaw_rate, acc, size3, type one-hot3
    if (tid < TM) {
        const int row = row0 + tid;
        float* a = attr + tid * 16;
        const int ty = rtype[tid];
        f32x4 sz = splat(0.f);
        if (tid < n_real) {
            const float* s = p.agent_size + ((size_t)b * p.n_agent + row) * 3;
            sz = f32x4{s[0], s[1], s[2], 0.f};
        }
        a[0] = rst[tid].aux[0]; a[1] = rst[tid].aux[1]; a[2] = rst[tid].st[3]; a[3] = rst[tid].aux[3]; a[4] = rst[tid].aux[2];
        a[5] = sz.x; a[6] = sz.y; a[7] = sz.z;
        a[8] = ty == 0 ? 1.f : 0.f; a[9] = ty == 1 ? 1.f : 0.f; a[10] = ty == 2 ? 1.f : 0.f;
    }
    __syncthreads();
    // ---- pose PE (pose_pe.py:57-62, pos_emb.py:24-25,54-55): 48 sincos per row, 3 per thread
    if (act) {
        const int row = tid >> 4, i = tid & 15;
        const float px = rst[row].st[0], py = rst[row].st[1], pyaw = rst[row].st[2];
        float* xr = X + row * LDT + 32;
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) {
            const int j = i * 3 + uu;
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
    if (act) {
        const int row = tid >> 4, o0 = (tid & 15) * 2;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int o = o0 + uu;
            float s = W[pw.enc_b1 + o];
            for (int k = 0; k < 11; ++k) s = fmaf(attr[row * 16 + k], W[pw.enc_w1 + o * 11 + k], s);
            ench[row * 32 + o] = fmaxf(s, 0.f);
        }
    }
    __syncthreads();
    if (act) {
        const int row = tid >> 4, o0 = (tid & 15) * 2;
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int o = o0 + uu;
            float s = W[pw.enc_b2 + o];
            for (int k = 0; k < 32; ++k) s = fmaf(ench[row * 32 + k], W[pw.enc_w2 + o * 32 + k], s);
            X[row * LDT + o] = s;
        }
    }
    __syncthreads();
    // zero invalid rows (input_pe_encoder.py:59)
    for (int i = tid; i < TM * 32; i += NT) {
        const int r = i >> 5;
        if (!rowvalid[r]) st4(X + r * LDT + (i & 31) * 4, splat(0.f));
    }
    __syncthreads();
}

// ---- faster head of A(t+1) for k_step_x: the InputPeEncoder weights and the PE frequencies sit in LDS (transposed: lanes of
// a row read consecutive outputs), and sin / cos are an fp64 Cody-Waite reduction + Taylor polynomial (|r| <= pi/4: truncation
// < 1e-11, far below half an fp32 ulp) instead of the library's general-purpose double sincos.  The scalar-per-lane global
// weight reads and the library call made this stage 28 k cycles per launch; same arithmetic for the MLP (fp32 fmaf chain in k
// order) and correctly rounded PE values as before.
constexpr int ENCW_W1T = 0;               // [11][32]
constexpr int ENCW_B1 = ENCW_W1T + 352;   // [32]
constexpr int ENCW_W2T = ENCW_B1 + 32;    // [32][32]
constexpr int ENCW_B2 = ENCW_W2T + 1024;  // [32]
constexpr int ENCW_FXY = ENCW_B2 + 32;    // [12]
constexpr int ENCW_FYAW = ENCW_FXY + 12;  // [24]
constexpr int ENCW_FLOATS = ENCW_FYAW + 24 + 4;

struct EncWRegs {
    f32x4 v[2];
};

// region table of the 369 float4 the encoder needs: (first float4 index, arena offset)
// (compile-time constants in a select chain: a local array indexed by `region` lands in constant MEMORY, and `local` then hangs on a
// dependent load from a cold line in the middle of the launch prologue's burst)
__device__ __forceinline__ const float* encw_src(const PolicyW& pw, const float* W, int q, int& region, int& local) {
    constexpr int first[7] = {0, 88, 96, 352, 360, 363, 369};
    const uint32_t off[6] = {pw.enc_w1, pw.enc_b1, pw.enc_w2, pw.enc_b2, pw.pe_fxy, pw.pe_fyaw};
    region = 0;
    local = q;
    uint32_t o = off[0];
#pragma unroll
    for (int r = 1; r < 6; ++r)
        if (q >= first[r]) {
            region = r;
            local = q - first[r];
            o = off[r];
        }
    return W + o + local * 4;
}

__device__ __forceinline__ void encw_issue(const PolicyW& pw, const float* W, int tid, EncWRegs& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + i * 256;
        int region, local;
        const float* src = encw_src(pw, W, q < 369 ? q : 0, region, local);
        r.v[i] = ldg4(src);
    }
}

__device__ __forceinline__ void encw_commit(int tid, const EncWRegs& r, float* ENCW) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + i * 256;
        if (q >= 369) continue;
        constexpr int first[7] = {0, 88, 96, 352, 360, 363, 369};
        int region = 0, local = q;
#pragma unroll
        for (int rr = 1; rr < 6; ++rr)
            if (q >= first[rr]) {
                region = rr;
                local = q - first[rr];
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = local * 4 + e;
            const float val = r.v[i][e];
            if (region == 0) ENCW[ENCW_W1T + (idx % 11) * 32 + idx / 11] = val;        // W1 [32][11] -> [11][32]
            else if (region == 1) ENCW[ENCW_B1 + idx] = val;
            else if (region == 2) ENCW[ENCW_W2T + (idx & 31) * 32 + (idx >> 5)] = val;  // W2 [32][32] -> transposed
            else if (region == 3) ENCW[ENCW_B2 + idx] = val;
            else if (region == 4) ENCW[ENCW_FXY + idx] = val;
            else ENCW[ENCW_FYAW + idx] = val;
        }
    }
}

// NT = threads of the workgroup: 256, or 512 (k_step_x8) where waves 4-7 only take part in the barriers
template <int NT = 256>
__device__ __forceinline__ void step_encode_inputs_lds(const RolloutP& p, int b, int row0, int n_real, int tid, const StepSmall& sm,
                                                       const float* ENCW, float* X, bool size_in_lds = false) {
    const bool act = NT == 256 || tid < 256;
    float* attr = sm.attr; float* ench = sm.ench; const RowSt* rst = sm.rst; const uint8_t* rowvalid = sm.rowvalid;
    const int* rtype = sm.rtype;
    // ---- agent attributes (sc_input.py:142-165): vel2, spd, yaw_rate, acc, size3, type one-hot3
    if (tid < TM) {
        const int row = row0 + tid;
        float* a = attr + tid * 16;
        const int ty = rtype[tid];
        f32x4 sz = splat(0.f);
        if (size_in_lds) {  // (epi_commit; zero for rows past n_real)
            const float* s = ench + EPI_ENCH_SIZE + tid * 3;
            sz = f32x4{s[0], s[1], s[2], 0.f};
        } else if (tid < n_real) {
            const float* s = p.agent_size + ((size_t)b * p.n_agent + row) * 3;
            sz = f32x4{s[0], s[1], s[2], 0.f};
        }
        a[0] = rst[tid].aux[0]; a[1] = rst[tid].aux[1]; a[2] = rst[tid].st[3]; a[3] = rst[tid].aux[3]; a[4] = rst[tid].aux[2];
        a[5] = sz.x; a[6] = sz.y; a[7] = sz.z;
        a[8] = ty == 0 ? 1.f : 0.f; a[9] = ty == 1 ? 1.f : 0.f; a[10] = ty == 2 ? 1.f : 0.f;
    }
    __syncthreads();
    const int row = (tid >> 4) & 15, i = tid & 15, o0 = i * 2;
    // ---- pose PE (pose_pe.py:57-62, pos_emb.py:24-25,54-55): 48 sincos per row, 3 per thread
    if (act) {
        const float px = rst[row].st[0], py = rst[row].st[1], pyaw = rst[row].st[2];
        float* xr = X + row * LDT + 32;
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) {
            const int j = i * 3 + uu;
            float arg;
            int c_cos, c_sin;
            if (j < 12) {
                arg = px * ENCW[ENCW_FXY + j]; c_cos = j; c_sin = 12 + j;
            } else if (j < 24) {
                arg = py * ENCW[ENCW_FXY + j - 12]; c_cos = 24 + (j - 12); c_sin = 36 + (j - 12);
            } else {
                arg = pyaw * ENCW[ENCW_FYAW + j - 24]; c_cos = 48 + (j - 24); c_sin = 72 + (j - 24);
            }
            float sv, cv;
#ifdef TB_DBG_SINCOSF  // (diagnosis builds only, VERDICT r04 task 4 (d): the library's fp32 sincos instead of the fp64 polynomial)
            sincosf(arg, &sv, &cv);
#else
            sincos_pe(arg, sv, cv);
#endif
            xr[c_cos] = cv;
            xr[c_sin] = sv;
        }
    }
    // ---- InputPeEncoder MLP 11 -> 32 -> 32 (input_pe_encoder.py:52-54), 2 outputs per thread
    if (act) {
        float s0 = ENCW[ENCW_B1 + o0], s1 = ENCW[ENCW_B1 + o0 + 1];
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float a = attr[row * 16 + k];
            s0 = fmaf(a, ENCW[ENCW_W1T + k * 32 + o0], s0);
            s1 = fmaf(a, ENCW[ENCW_W1T + k * 32 + o0 + 1], s1);
        }
        ench[row * 32 + o0] = fmaxf(s0, 0.f);
        ench[row * 32 + o0 + 1] = fmaxf(s1, 0.f);
    }
    __syncthreads();
    if (act) {
        float s0 = ENCW[ENCW_B2 + o0], s1 = ENCW[ENCW_B2 + o0 + 1];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float h = ench[row * 32 + k];
            s0 = fmaf(h, ENCW[ENCW_W2T + k * 32 + o0], s0);
            s1 = fmaf(h, ENCW[ENCW_W2T + k * 32 + o0 + 1], s1);
        }
        const bool rv = rowvalid[row] != 0;  // zero invalid rows (input_pe_encoder.py:59)
        X[row * LDT + o0] = rv ? s0 : 0.f;
        X[row * LDT + o0 + 1] = rv ? s1 : 0.f;
    }
    __syncthreads();
    // the PE part of invalid rows
    for (int q = tid; q < TM * 24; q += NT) {
        const int r = q / 24;
        if (!rowvalid[r]) st4(X + r * LDT + 32 + (q - r * 24) * 4, splat(0.f));
    }
    __syncthreads();
}

}  // namespace tb
