// 8-wave (512-thread, TWO waves per SIMD) variants of the XDL tile stages (tb_device_xdl.hpp), used by k_step_x8.
//
// With 4 waves per workgroup every SIMD holds ONE wave, and the stages that are not bound by the weight stream -- the attention
// loop (softmax VALU + K/V latency), LayerNorm, the epilogues -- run at the speed of one wave's dependent instruction chain.
// tools/microtests/attn_loop.hip: the same attention loop with two waves per head, each taking every second 32-key block,
// finishes a 256-key attention in 0.65x and a 1024-key one in 0.45-0.55x the time (fp16 pairs and bf16 alike), while a chain of
// weight-streaming Linears (tools/microtests/gemm_chain.hip) is 3 % slower with 8 waves x 1 tile than with 4 waves x 2 tiles:
// that stage is bound by the CU's L2 -> L1 path (37-38 B/clk) either way.
//
// Work split of a 16-agent tile over 8 waves:
//   Linear      : wave w computes output tile w (features 16w .. 16w+15); a weight unit is ONE tile x 128 k (8 KiB per wave
//                 as an fp16 pair), requested one unit ahead exactly like the 2-tile units of the 4-wave kernel;
//   attention   : head h = w / 2; its two waves walk the key blocks i = half, half + 2, ... of the (staggered, wrapping) walk,
//                 each with its own online softmax, and merge (o, max, sum) through LDS; wave (h, half) then owns the
//                 normalised output tile 2h + half = w -- the B operand tile of the out-projection it computes next;
//   Q           : the projection gives wave w tile w; the two tiles of a head are exchanged between its waves through LDS;
//   LayerNorm   : 32 threads per row, one float4 each (16-lane DPP reduction + one row swap);
//   everything per agent (epilogue, input encoder) keeps its 256-thread form on waves 0-3.
#pragma once
#include "tb_device_xdl.hpp"

namespace tb {
namespace TB_XNS {

constexpr int NT8 = 512;

// ---------------------------------------------------------------------------------------------
// weight units: 1 output tile x 4 chunks (128 k) x NPL planes = 4 * NPL fragments of 8 halfs per lane (32 / 16 VGPRs) + bias
// ---------------------------------------------------------------------------------------------
struct WUnit1 {
    xh8 w[4][NPL];
    f32x4 b;
};
struct WNext1 {
    const xhalf* wpk;
    const float* bias;  // or nullptr
    int tile;
    int nchunk;  // chunks per output tile of this Linear (K / 32)
    int c0;      // first chunk of this unit
};

__device__ __forceinline__ WNext1 wnext1(const float* arena, uint32_t off, const float* bias, int tile, int nchunk = 4, int c0 = 0) {
    return WNext1{reinterpret_cast<const xhalf*>(arena + off), bias, tile, nchunk, c0};
}
__device__ __forceinline__ const xh8* wfrag1(const WNext1& n, int lane) {
    return reinterpret_cast<const xh8*>(n.wpk + ((size_t)(n.tile * n.nchunk + n.c0) * NPL) * 512 + lane * 8);
}

__device__ __forceinline__ void wload1(WUnit1& u, const WNext1& n, int lane) {
    const xh8* pa = wfrag1(n, lane);
    TB_SCHED_FENCE();
    u.b = n.bias ? ldg4(n.bias + n.tile * 16 + (lane >> 4) * 4) : splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) u.w[c][p] = pa[(c * NPL + p) * 64];
    TB_SCHED_FENCE();
}

// acc += unit . X^T from planes; requests the next unit in the MFMAs' shadow.  bp : this lane's B base = P + m*ld + kq*8
__device__ __forceinline__ void wmma1_pf(f32x4& acc, const WUnit1& u, const xhalf* bp, int plane_stride, WUnit1& un, const WNext1& n, int lane) {
    const xh8* pa = wfrag1(n, lane);
    const float* ba = n.bias ? n.bias + n.tile * 16 + (lane >> 4) * 4 : reinterpret_cast<const float*>(n.wpk);
    TB_SCHED_FENCE();
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    un.b = ldg4(ba);
    f32x4 mid = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) un.w[c][p] = pa[(c * NPL + p) * 64];
        if (NPL == 2) {
            mid = mfma_h(u.w[c][0], x[c][P1], mid);
            mid = mfma_h(u.w[c][P1], x[c][0], mid);
        }
        acc = mfma_h(u.w[c][0], x[c][0], acc);
    }
    // pin the order: the LDS reads + the bias load, then the weight loads spread under the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (NPL == 2) {  // 4 x (2 MFMA, 1 load, 1 MFMA, 1 load)
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    if (NPL == 2) acc += mid * splat(SPLIT_INV);
    if (!n.bias) un.b = splat(0.f);
}

// the same unit without a follow-up request
__device__ __forceinline__ void wmma1(f32x4& acc, const WUnit1& u, const xhalf* bp, int plane_stride) {
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
            mid = mfma_h(u.w[c][0], x[c][P1], mid);
            mid = mfma_h(u.w[c][P1], x[c][0], mid);
        }
        acc = mfma_h(u.w[c][0], x[c][0], acc);
    }
    if (NPL == 2) acc += mid * splat(SPLIT_INV);
}

// ---------------------------------------------------------------------------------------------
// tile <-> planes with 512 threads: 32 threads per row, one float4 each
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_to_planes8(const float* src, int lds_, xhalf* P, int tid) {
    const int row = tid >> 5, c4 = (tid & 31) * 4;
    planes_store4(P, PLANE, LDP, row, c4, lds4(src + row * lds_ + c4));
}

// sum over the 32 lanes that hold one tile row (two adjacent 16-lane DPP rows of the wave)
__device__ __forceinline__ float row32_sum(float v) {
    float lo, hi;
    rows_pair16(row16_sum(v), lo, hi);
    return lo + hi;
}

template <bool PARAMS_IN_LDS = false, bool AFFINE = true>
__device__ __forceinline__ void layernorm_planes8(const float* src, int lds_, xhalf* P, const float* __restrict__ g, const float* __restrict__ b,
                                                  int tid) {
    const int row = tid >> 5, c0 = (tid & 31) * 4;
    const f32x4 a = lds4(src + row * lds_ + c0);
    f32x4 g0 = splat(1.f), b0 = splat(0.f);
    if (AFFINE) {
        if (PARAMS_IN_LDS) {
            g0 = lds4_explicit(g + c0);
            b0 = lds4_explicit(b + c0);
        } else {
            g0 = ldg4(g + c0);
            b0 = ldg4(b + c0);
        }
    }
    const float mean = row32_sum((a.x + a.y) + (a.z + a.w)) * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean);
    const float v = row32_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w));
    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    planes_store4(P, PLANE, LDP, row, c0, AFFINE ? da * splat(rstd) * g0 + b0 : da * splat(rstd));
}

__device__ __forceinline__ void load_tile8(float* dst, int ld, const float* __restrict__ src, int n_real, int tid) {
    const int row = tid >> 5, c4 = (tid & 31) * 4;
    st4(dst + row * ld + c4, row < n_real ? ldg4(src + (size_t)row * H + c4) : splat(0.f));
}
__device__ __forceinline__ void store_tile8(float* __restrict__ dst, const float* src, int ld, int n_real, int tid) {
    const int row = tid >> 5, c4 = (tid & 31) * 4;
    if (row < n_real) st4(dst + (size_t)row * H + c4, lds4(src + row * ld + c4));
}

// ---------------------------------------------------------------------------------------------
// attention, one head per wave PAIR.  Wave (head, half) reduces the key blocks half, half + 2, ... of the walk that starts at
// kstart and wraps; its partial state (unnormalised O as ONE fp32 value per element, running max, running row sum) is merged with
// the partner's through LDS by attention_merge8.
// ---------------------------------------------------------------------------------------------
struct AttnPart8 {
    f32x4 o[2];
    float run_max, run_sum;  // per row (uniform over the four kq lane groups after the row reductions)
};

struct AttnPre8 {
    KFragX k0f;
    VFragX vc;
    AttnPreX whole;  // (short walks: the even wave of a pair walks all blocks with attention_head_x)
};

// key offset of this wave's j-th block
__device__ __forceinline__ int blk8(int kstart, int half, int j, int n_key_pad) { return kwrap(kstart + 32 * (half + 2 * j), n_key_pad); }

__device__ __forceinline__ void attention_prefetch8(AttnPre8& a, const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                    const float* __restrict__ keybias, int n_key_pad, int kstart, int head, int half, int lane) {
    const int kq = lane >> 4;
    const int nblk = n_key_pad >> 5, nb = (nblk - half + 1) >> 1;
    if (nb <= 0) return;  // (wave-uniform) a single block: the odd wave of the pair has nothing to walk
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    const int b0 = blk8(kstart, half, 0, n_key_pad);
    // (only the FIRST block here: eight waves requesting two K blocks and one V block each is a 147 KB burst in front of the
    // barrier -- twice the 4-wave kernel's -- and the waves stall issuing it; the second K block is requested at the head of the walk)
    TB_SCHED_FENCE();
    k_load_x(a.k0f, kbase, bbase, b0);
    v_load_x(a.vc, vbase, b0);
    TB_SCHED_FENCE();
}

// q = the head's two Q^T tiles (features head*32 + tt*16 + 4 kq + r of agent m).  `un` / `nx`: the wave's next weight unit (the
// out-projection tile), requested from inside the loop like in attention_head_x -- or right away when this wave has no block.
template <bool SELFMASK>
__device__ __forceinline__ void attention_half8(const f32x4 (&q)[2], AttnPre8& pre, const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                const float* __restrict__ keybias, int n_key_pad, int kstart, int head, int half, int lane,
                                                int self_key, AttnPart8& part, WUnit1& un, const WNext1& nx) {
    const int kq = lane >> 4;
    const int nblk = n_key_pad >> 5, nb = (nblk - half + 1) >> 1;
    part.o[0] = splat(0.f);
    part.o[1] = splat(0.f);
    part.run_max = -INFINITY;
    part.run_sum = 0.f;
    if (nb <= 0) {
        wload1(un, nx, lane);
        return;
    }
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    xh8 qh, ql;
    split8(q[0], q[1], qh, ql);
    f32x4 oh[2] = {splat(0.f), splat(0.f)}, oc[2] = {splat(0.f), splat(0.f)};
    KFragX kn;
    VFragX vc = pre.vc;
    float run_max = -INFINITY, run_sum = 0.f, new_max, alpha, sv[8];
    TB_SCHED_FENCE();
    k_load_x(kn, kbase, bbase, nb > 1 ? blk8(kstart, half, 1, n_key_pad) : blk8(kstart, half, 0, n_key_pad));
    TB_SCHED_FENCE();
    {
        f32x4 s[2], c[2];
        attn_qk_x(pre.k0f, qh, ql, s, c);
        attn_stats_x<SELFMASK>(s, c, pre.k0f.kb, blk8(kstart, half, 0, n_key_pad) + kq * 4, self_key, run_max, sv, new_max, alpha);
    }
    const int j_issue = nb >= 2 ? nb - 2 : 0;
    for (int j = 0; j < nb; ++j) {
        const int kc = blk8(kstart, half, j, n_key_pad);
        const int kn1 = (j + 1 < nb) ? blk8(kstart, half, j + 1, n_key_pad) : kc;  // clamped re-reads on the tail are harmless
        const int kld = (j + 2 < nb) ? blk8(kstart, half, j + 2, n_key_pad) : kc;
        TB_SCHED_FENCE();
        f32x4 ts[2], tc[2];
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_qk_x(kn, qh, ql, ts, tc);
        in_vgpr(ts[0]); in_vgpr(ts[1]); in_vgpr(tc[0]); in_vgpr(tc[1]);
        const f32x4 nb_[2] = {kn.kb[0], kn.kb[1]};
        TB_SCHED_FENCE();
        k_load_x(kn, kbase, bbase, kld);
        if (j == j_issue) wload1(un, nx, lane);
        float p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = exp2_neg(sv[r] - new_max);
        run_sum = run_sum * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        run_max = new_max;
        xh8 ph, pl;
        split8<false>(f32x4{p[0], p[1], p[2], p[3]}, f32x4{p[4], p[5], p[6], p[7]}, ph, pl);  // (probabilities: in [0, 1])
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            oh[dt] *= splat(alpha);
            oc[dt] *= splat(alpha);
        }
        TB_SCHED_FENCE();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if (NPL == 2) oc[dt] = mfma_h(vc.va[dt][0], pl, oc[dt]);
            oh[dt] = mfma_h(vc.va[dt][0], ph, oh[dt]);
            if (NPL == 2) oc[dt] = mfma_h(vc.va[dt][P1], ph, oc[dt]);
        }
        TB_SCHED_FENCE();
        v_load_x(vc, vbase, kn1);
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_stats_x<SELFMASK>(ts, tc, nb_, kn1 + kq * 4, self_key, run_max, sv, new_max, alpha);  // (unused after the last block)
        TB_SCHED_FENCE();
    }
    part.o[0] = oh[0] + oc[0] * splat(SPLIT_INV);
    part.o[1] = oh[1] + oc[1] * splat(SPLIT_INV);
    part.run_max = run_max;
    part.run_sum = rows_sum(run_sum);
}

// LDS exchange areas of the 8-wave attention (floats): XQ [8 waves][64 lanes][4] Q tiles, XO [8][2 d tiles][64][4] partial outputs,
// XS [8][16 rows][2] (running max, row sum)
constexpr int XQ8_FLOATS = 8 * 256;
constexpr int XO8_FLOATS = 8 * 2 * 256;
constexpr int XS8_FLOATS = 8 * 32;

// publish this wave's partial state; after the workgroup barrier attention_merge8 returns the normalised output TILE `wave`
// (d tile `half` of head `head`) and whether the row had no valid key in either half
__device__ __forceinline__ void attention_publish8(const AttnPart8& part, float* XO, float* XS, int wave, int lane) {
    st4(XO + (wave * 2 + 0) * 256 + lane * 4, part.o[0]);
    st4(XO + (wave * 2 + 1) * 256 + lane * 4, part.o[1]);
    if ((lane >> 4) == 0) {
        XS[wave * 32 + (lane & 15) * 2 + 0] = part.run_max;
        XS[wave * 32 + (lane & 15) * 2 + 1] = part.run_sum;
    }
}
__device__ __forceinline__ bool attention_merge8(const AttnPart8& part, const float* XO, const float* XS, int wave, int lane, f32x4& o) {
    const int half = wave & 1, pw = wave ^ 1, m = lane & 15;
    const f32x4 op = lds4(XO + (pw * 2 + half) * 256 + lane * 4);
    const float mp = XS[pw * 32 + m * 2 + 0], sp = XS[pw * 32 + m * 2 + 1];
    const float new_max = fmaxf(part.run_max, mp);
    const float a_s = exp2_neg(part.run_max - new_max), a_p = exp2_neg(mp - new_max);
    // (a fixed operand order -- even wave's term first -- so that both waves of the pair form the same sum bit for bit)
    const float s_e = half == 0 ? part.run_sum * a_s : sp * a_p, s_o = half == 0 ? sp * a_p : part.run_sum * a_s;
    const float sum = s_e + s_o;
    const bool novalid = !(sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / sum;
    const f32x4 o_e = half == 0 ? part.o[half] * splat(a_s) : op * splat(a_p), o_o = half == 0 ? op * splat(a_p) : part.o[half] * splat(a_s);
    o = (o_e + o_o) * splat(inv);
    return novalid;
}

// K / V accumulators of ONE 16-feature tile of a 16-token tile -> global, fragment-major (tb_device_xdl.hpp).  tile t = wave & 1 of
// head = wave >> 1: the K tile fills halfs t*4 .. t*4+3 of each lane fragment, the V tile the d-tile t fragments.
__device__ __forceinline__ void kv_store_key1(xhalf* __restrict__ Kf, xhalf* __restrict__ Vf, int tok, int head, int t, int lane, const f32x4& ak,
                                              const f32x4& av, bool real) {
    const int kq = lane >> 4;
    const int j = tok & 31;
    xhalf* kblk = Kf + (size_t)(tok >> 5) * KV_BLOCK_HALFS + head * (NPL * 1024);
    xhalf* vblk = Vf + (size_t)(tok >> 5) * KV_BLOCK_HALFS + head * (NPL * 1024);
    const int kt = j >> 4, krow = j & 15;
    const int vq = (j >> 2) & 3, ve = (j >> 4) * 4 + (j & 3);
    xh4 h, l;
    split2(real ? ak : splat(0.f), h, l);
    xhalf* pk = kblk + (kt * 64 + kq * 16 + krow) * 8 + t * 4;
    *reinterpret_cast<xh4*>(pk) = h;
    if (NPL == 2) *reinterpret_cast<xh4*>(pk + 1024) = l;
    split2(real ? av : splat(0.f), h, l);
    xhalf* pv = vblk + (t * 64 + vq * 16 + kq * 4) * 8 + ve;
    pv[0] = h.x; pv[8] = h.y; pv[16] = h.z; pv[24] = h.w;
    if (NPL == 2) {
        pv += 1024;
        pv[0] = l.x; pv[8] = l.y; pv[16] = l.z; pv[24] = l.w;
    }
}

// ---------------------------------------------------------------------------------------------
// One pre-LN cross-attention layer on 8 waves.  u = the Q unit of tile `wave` on entry, `nxt` on exit.
//   X : [16][LDT] fp32 residual stream;  P1, P2 : plane buffers;  XQ / XO / XS : exchange areas (see above)
// ---------------------------------------------------------------------------------------------
struct Xch8 {
    float* xq;
    float* xo;
    float* xs;
};

template <bool LNLDS = false, bool SELFMASK = false>
__device__ __forceinline__ void xattn_layer8(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1, xhalf* P2,
                                             const Xch8& xc, const xhalf* __restrict__ Kmat, const xhalf* __restrict__ VT,
                                             const float* __restrict__ keybias, int n_key_pad, int kstart, int self_key0, const uint8_t* rowvalid,
                                             uint8_t* novalid_s, int tid, WUnit1& u, const WNext1& nxt, const float* lnblk = nullptr, long long* prof = nullptr) {
    if (!LNLDS) lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63, head = wave >> 1, half = wave & 1;
    TB_XSTAMP(16);
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    AttnPre8 apre;
    // Walks of fewer than four key blocks (traffic lights, the interaction of <= 96 agents) are not split: the exchange, the merge
    // and their two barriers cost more than half of such a walk; the even wave of each pair runs attention_head_x over all blocks
    // (the 4-wave kernel's loop) and stores both output tiles of the head, the odd wave only lends its Q tile.
    const bool split = (n_key_pad >> 5) >= 4;  // (workgroup-uniform)
    layernorm_planes8<LNLDS>(X, LDT, P1, lnblk, lnblk + 128, tid);
    if (split) attention_prefetch8(apre, Kmat, VT, keybias, n_key_pad, kstart, head, half, lane);
    else if (half == 0) attention_prefetch_x(apre.whole, Kmat, VT, keybias, n_key_pad, kstart, head, lane);
    __syncthreads();
    TB_XSTAMP(17);
    WUnit1 u2;
    f32x4 qt = u.b;
    wmma1(qt, u, b1, PLANE);
    st4(xc.xq + wave * 256 + lane * 4, qt);
    __syncthreads();
    const f32x4 qp = lds4(xc.xq + (wave ^ 1) * 256 + lane * 4);
    const f32x4 q[2] = {half == 0 ? qt : qp, half == 0 ? qp : qt};
    TB_XSTAMP(18);
    if (split) {
        AttnPart8 part;
        attention_half8<SELFMASK>(q, apre, Kmat, VT, keybias, n_key_pad, kstart, head, half, lane, self_key0 >= 0 ? self_key0 + m : -1, part, u2,
                                  wnext1(W, LX.wo, W + L.bo, wave));
        TB_XSTAMP(19);
        attention_publish8(part, xc.xo, xc.xs, wave, lane);
        __syncthreads();
        f32x4 o;
        const bool novalid = attention_merge8(part, xc.xo, xc.xs, wave, lane, o);
        planes_store_c(P2, wave, lane, o);
        if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;  // (per row, identical in every head)
    } else {
        wload1(u2, wnext1(W, LX.wo, W + L.bo, wave), lane);
        if (half == 0) {
            f32x4 o[2];
            WUnitX dummy;  // (ISSUE = false: attention_head_x requests no weight unit here)
            const bool novalid = attention_head_x<SELFMASK, false>(q, apre.whole, Kmat, VT, keybias, n_key_pad, kstart, head, lane,
                                                                   self_key0 >= 0 ? self_key0 + m : -1, o, dummy, WNextX{});
            planes_store_c(P2, wave, lane, o[0]);
            planes_store_c(P2, wave + 1, lane, o[1]);
            if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
        }
        TB_XSTAMP(19);
    }
    __syncthreads();
    TB_XSTAMP(20);
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, b2, PLANE, u, wnext1(W, LX.w1, W + L.b1, wave), lane);
        const bool nv = novalid_s[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 xo = lds4(px);
        st4(px, nv ? xo : xo + acc);
    }
    __syncthreads();
    TB_XSTAMP(21);
    layernorm_planes8<LNLDS>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    TB_XSTAMP(22);
    {
        f32x4 acc = u.b;
        wmma1_pf(acc, u, b1, PLANE, u2, wnext1(W, LX.w2, W + L.b2, wave), lane);
        planes_store_c(P2, wave, lane, relu4(acc));
    }
    __syncthreads();
    TB_XSTAMP(23);
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, b2, PLANE, u, nxt, lane);
        const bool rv = rowvalid[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 xo = lds4(px);
        st4(px, rv ? xo + acc : splat(0.f));
    }
    __syncthreads();
    TB_XSTAMP(24);
}

// the same layer without its attention half (no valid key at all: SURVEY A.2); u = the FFN1 unit on entry
template <bool LNLDS = false>
__device__ __forceinline__ void ffn_layer8(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1, xhalf* P2,
                                           const uint8_t* rowvalid, int tid, WUnit1& u, const WNext1& nxt, const float* lnblk = nullptr) {
    if (!LNLDS) lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    WUnit1 u2;
    layernorm_planes8<LNLDS>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    {
        f32x4 acc = u.b;
        wmma1_pf(acc, u, b1, PLANE, u2, wnext1(W, LX.w2, W + L.b2, wave), lane);
        planes_store_c(P2, wave, lane, relu4(acc));
    }
    __syncthreads();
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, b2, PLANE, u, nxt, lane);
        const bool rv = rowvalid[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 xo = lds4(px);
        st4(px, rv ? xo + acc : splat(0.f));
    }
    __syncthreads();
}

// K/V of the tile for the three interaction layers from ONE normalisation (norm_tgt folded into kvf / bkvf, PolicyWX).
// u holds the K unit (tile `wave`) of layer 0 on entry and `nxt` on exit; Kmat / VT point at layer 0, layer l at + 2 l ls (halfs).
__device__ __forceinline__ void kv_project_shared8(const float* __restrict__ W, const uint32_t (&kvf)[3], const uint32_t (&bkvf)[3], const float* T,
                                                   xhalf* P1, xhalf* __restrict__ Kmat, xhalf* __restrict__ VT, size_t ls, int tok0, int n_real_rows,
                                                   int tid, WUnit1& u, const WNext1& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    layernorm_planes8<false, false>(T, LDT, P1, nullptr, nullptr, tid);
    __syncthreads();
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    WUnit1 u2;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        f32x4 ak = u.b;
        wmma1_pf(ak, u, b1, PLANE, u2, wnext1(W, kvf[l], W + bkvf[l], 8 + wave), lane);
        f32x4 av = u2.b;
        wmma1_pf(av, u2, b1, PLANE, u, l < 2 ? wnext1(W, kvf[l + 1], W + bkvf[l + 1], wave) : nxt, lane);
        kv_store_key1(Kmat + 2 * l * ls, VT + 2 * l * ls, tok0 + m, wave >> 1, wave & 1, lane, ak, av, m < n_real_rows);
    }
    __syncthreads();
}

// One GRU layer step on 8 waves: wave w computes tile w of the gates r, z, n (agent_temporal.py:147-152)
__device__ __forceinline__ void gru_layer8(const float* __restrict__ W, const GruLayerW& G, const GruLayerX& GX, const xhalf* XinP, const xhalf* HsP,
                                           const float* Hs, xhalf* OutP, float* Out, const uint8_t* rowvalid, float* __restrict__ h_global,
                                           int n_real_rows, int tid, WUnit1& u, const WNext1& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* xr = XinP + m * LDP + kq * 8;
    const xhalf* hr = HsP + m * LDP + kq * 8;
    const float* bih = W + G.bih;
    const float* bhh = W + G.bhh;
    WUnit1 u2;
    f32x4 r = u.b;
    wmma1_pf(r, u, xr, PLANE, u2, wnext1(W, GX.whh, bhh, wave), lane);
    r += u2.b;
    wmma1_pf(r, u2, hr, PLANE, u, wnext1(W, GX.wih, bih, 8 + wave), lane);
    f32x4 z = u.b;
    wmma1_pf(z, u, xr, PLANE, u2, wnext1(W, GX.whh, bhh, 8 + wave), lane);
    z += u2.b;
    wmma1_pf(z, u2, hr, PLANE, u, wnext1(W, GX.wih, bih, 16 + wave), lane);
    f32x4 gin = u.b;
    wmma1_pf(gin, u, xr, PLANE, u2, wnext1(W, GX.whh, bhh, 16 + wave), lane);
    f32x4 ghn = u2.b;
    wmma1_pf(ghn, u2, hr, PLANE, u, nxt, lane);
    const bool rv = rowvalid[m] != 0;
    const f32x4 hold = lds4(Hs + m * LDT + wave * 16 + kq * 4);
    f32x4 hn;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float rg = sigmoidf_(r[q]);
        const float zg = sigmoidf_(z[q]);
        const float ng = tanhf_(gin[q] + rg * ghn[q]);
        hn[q] = rv ? (1.0f - zg) * ng + zg * hold[q] : 0.f;
    }
    if (OutP) planes_store_c(OutP, wave, lane, hn);
    if (Out) st4(cptr(Out, LDT, wave, lane), hn);
    if (m < n_real_rows) st4(h_global + (size_t)m * H + wave * 16 + kq * 4, hn);
    __syncthreads();
}

}  // namespace TB_XNS
}  // namespace tb
