// 8-wave (512-thread, two waves per SIMD) variants of the tile stages, used by k_step8.
//
// With 4 waves per workgroup every SIMD holds ONE wave: the MFMA pipe idles whenever that wave is in a softmax,
// an epilogue, a barrier or an LDS wait (measured: pipe busy 46 % of the time).  Here a workgroup still owns 16
// agents, but wave w computes ONE 16-feature tile (w) of every Linear, and the two waves of a head (2h, 2h+1) split
// the key blocks of the attention between them and merge their partial softmax state through LDS.  The two waves that
// share a SIMD then fill each other's bubbles.
#pragma once
#include "tb_device.hpp"

namespace tb {

constexpr int NTHREADS8 = 512;

struct WUnit1 {
    f32x4 w[8];  // one tile x 8 k-step groups
    f32x4 b;
};

struct WNext1 {
    const float* wpk;
    const float* bias;
    int tile, kj_total, j0;
};

__device__ __forceinline__ WNext1 wnext1(const float* wpk, const float* bias, int tile, int kj_total = 8, int j0 = 0) {
    return WNext1{wpk, bias, tile, kj_total, j0};
}

__device__ __forceinline__ void wload1(WUnit1& u, const WNext1& n, int lane) {
    const float* pa = n.wpk + ((size_t)(n.tile * n.kj_total + n.j0) * 64 + lane) * 4;
    TB_SCHED_FENCE();
    u.b = n.bias ? ldg4(n.bias + n.tile * 16 + (lane >> 4) * 4) : splat(0.f);
#pragma unroll
    for (int j = 0; j < 8; ++j) u.w[j] = ldg4(pa + j * 256);
    TB_SCHED_FENCE();
}

// acc += unit . X^T with the next unit requested in the MFMAs' shadow; even / odd k-groups go to two accumulators
// (a single chain would pay the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32), summed at the end.
__device__ __forceinline__ void wmma1_pf(f32x4& acc, const WUnit1& u, const float* xrow, WUnit1& un, const WNext1& n, int lane) {
    const float* pa = n.wpk + ((size_t)(n.tile * n.kj_total + n.j0) * 64 + lane) * 4;
    const float* ba = n.bias ? n.bias + n.tile * 16 + (lane >> 4) * 4 : n.wpk;
    TB_SCHED_FENCE();
    un.b = ldg4(ba);
#pragma unroll
    for (int j = 0; j < 8; ++j) un.w[j] = ldg4(pa + j * 256);
    f32x4 xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = lds4(xrow + 4 * j);
    f32x4 acc2 = splat(0.f);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        acc = mfma4(u.w[j].x, xv[j].x, acc);
        acc2 = mfma4(u.w[j + 1].x, xv[j + 1].x, acc2);
        acc = mfma4(u.w[j].y, xv[j].y, acc);
        acc2 = mfma4(u.w[j + 1].y, xv[j + 1].y, acc2);
        acc = mfma4(u.w[j].z, xv[j].z, acc);
        acc2 = mfma4(u.w[j + 1].z, xv[j + 1].z, acc2);
        acc = mfma4(u.w[j].w, xv[j].w, acc);
        acc2 = mfma4(u.w[j + 1].w, xv[j + 1].w, acc2);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    acc += acc2;
    if (!n.bias) un.b = splat(0.f);
}

// LayerNorm of a [16][128] LDS tile with 512 threads: 32 threads per row, one float4 each; the 32-lane reduction is a
// 16-lane DPP row reduction plus one row swap (rows {0,1} / {2,3} of the wave hold one tile row each).
template <bool PARAMS_IN_LDS = false>
__device__ __forceinline__ void layernorm_tile8(const float* src, int lds_, float* dst, int ldd, const float* __restrict__ g,
                                                const float* __restrict__ b, int tid) {
    const int row = tid >> 5, c0 = (tid & 31) * 4;
    const f32x4 a = lds4(src + row * lds_ + c0);
    f32x4 g0, b0;
    if (PARAMS_IN_LDS) {
        g0 = lds4_explicit(g + c0);
        b0 = lds4_explicit(b + c0);
    } else {
        g0 = ldg4(g + c0);
        b0 = ldg4(b + c0);
    }
    float lo, hi;
    rows_pair16(row16_sum((a.x + a.y) + (a.z + a.w)), lo, hi);
    const float mean = (lo + hi) * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean);
    rows_pair16(row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w)), lo, hi);
    const float rstd = 1.0f / sqrtf((lo + hi) * (1.0f / 128.0f) + LN_EPS);
    st4(dst + row * ldd + c0, da * splat(rstd) * g0 + b0);
}

__device__ __forceinline__ void load_tile8(float* dst, int ld, const float* __restrict__ src, int n_real, int tid) {
    const int row = tid >> 5, c4 = (tid & 31) * 4;  // 16 rows x 32 float4 = 512
    st4(dst + row * ld + c4, row < n_real ? ldg4(src + (size_t)row * H + c4) : splat(0.f));
}

__device__ __forceinline__ void store_tile8(float* __restrict__ dst, const float* src, int ld, int n_real, int tid) {
    const int row = tid >> 5, c4 = (tid & 31) * 4;
    if (row < n_real) st4(dst + (size_t)row * H + c4, lds4(src + row * ld + c4));
}

// ---------------------------------------------------------------------------------------------
// attention, one head per wave PAIR: wave (head, half) reduces the 32-key blocks half, half+2, half+4, ...
// Partial state (unnormalised O, running max, running sum) is left in registers for the LDS merge.
// ---------------------------------------------------------------------------------------------
struct AttnPart {
    f32x4 o[2];
    float run_max, run_sum;
};

__device__ __forceinline__ void attention_half(const f32x4 (&q)[2], const float* __restrict__ Kmat, const float* __restrict__ VT,
                                               const float* __restrict__ keybias, int n_key_pad, int head, int half, int lane,
                                               int self_key, AttnPart& part) {
    const int kq = lane >> 4, m = lane & 15;
    part.o[0] = splat(0.f);
    part.o[1] = splat(0.f);
    part.run_max = -INFINITY;
    part.run_sum = 0.f;
    const int first = 32 * half;
    if (first >= n_key_pad) return;  // (wave-uniform) a single block: the odd wave has nothing to do
    const float* kbase = Kmat + (size_t)m * H + head * DHEAD + kq * 4;
    const float* vbase = VT + (size_t)(head * DHEAD + m) * n_key_pad + kq * 4;
    const float* bbase = keybias + kq * 4;
    KFrag kn;
    VFrag vc;
    float new_max, alpha, sv[8];
    {
        KFrag k0f;
        TB_SCHED_FENCE();
        k_load(k0f, kbase, bbase, first);
        v_load(vc, vbase, n_key_pad, first);
        k_load(kn, kbase, bbase, first + 64 < n_key_pad ? first + 64 : first);
        TB_SCHED_FENCE();
        f32x4 s0, s1;
        attn_qk(k0f, q, s0, s1);
        in_vgpr(s0);
        in_vgpr(s1);
        attn_stats(s0, s1, k0f.kb, first + kq * 4, self_key, part.run_max, sv, new_max, alpha);
    }
    for (int k0 = first; k0 < n_key_pad; k0 += 64) {
        const bool has_next = k0 + 64 < n_key_pad;
        const int k2 = (k0 + 128 < n_key_pad) ? k0 + 128 : k0;
        const int k1 = has_next ? k0 + 64 : k0;
        TB_SCHED_FENCE();
        f32x4 t0, t1;
        in_vgpr(part.o[0]);
        in_vgpr(part.o[1]);
        attn_qk(kn, q, t0, t1);
        f32x4 nb[2] = {kn.kb[0], kn.kb[1]};
        KFrag k2f;
        VFrag v1f;
        k_load(k2f, kbase, bbase, k2);
        v_load(v1f, vbase, n_key_pad, k1);
        float p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = exp_neg(sv[r] - new_max);
        part.run_sum = part.run_sum * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        part.run_max = new_max;
        part.o[0] *= splat(alpha);
        part.o[1] *= splat(alpha);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            if (g < 11) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        TB_SCHED_FENCE();
        in_vgpr(t0);
        in_vgpr(t1);
        in_vgpr(part.o[0]);
        in_vgpr(part.o[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            part.o[0] = mfma4(vc.va[t][0].x, p[4 * t + 0], part.o[0]);
            part.o[1] = mfma4(vc.va[t][1].x, p[4 * t + 0], part.o[1]);
            part.o[0] = mfma4(vc.va[t][0].y, p[4 * t + 1], part.o[0]);
            part.o[1] = mfma4(vc.va[t][1].y, p[4 * t + 1], part.o[1]);
            part.o[0] = mfma4(vc.va[t][0].z, p[4 * t + 2], part.o[0]);
            part.o[1] = mfma4(vc.va[t][1].z, p[4 * t + 2], part.o[1]);
            part.o[0] = mfma4(vc.va[t][0].w, p[4 * t + 3], part.o[0]);
            part.o[1] = mfma4(vc.va[t][1].w, p[4 * t + 3], part.o[1]);
        }
        attn_stats(t0, t1, nb, k1 + kq * 4, self_key, part.run_max, sv, new_max, alpha);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        TB_SCHED_FENCE();
        kn = k2f;
        vc = v1f;
    }
    part.run_sum = rows_sum(part.run_sum);
}

// LDS scratch of the 8-wave cross-attention layer
struct X8Scratch {
    float* S1;   // [16][LDT] LN output
    float* S2;   // [16][LDT] q exchange, then merged attention output / FFN hidden
    float* PO0;  // [16][LDT] partial O of the even waves
    float* PO1;  // [16][LDT] partial O of the odd waves
    float* PML;  // [2 halves][4 heads][16 agents][2] running max, running sum
};

__device__ __forceinline__ WNext1 xlayer_first8(const float* W, const XLayerW& L, int wave) { return wnext1(W + L.wq, W + L.bq, wave); }
__device__ __forceinline__ WNext1 kvproj_first8(const float* W, const XLayerW& L, int wave) { return wnext1(W + L.wkv, W + L.bkv, wave); }

// One pre-LN cross-attention layer on 8 waves (same arithmetic as xattn_layer; the softmax of a row is merged from two
// partial reductions, which changes only the rounding order).  Ends with a barrier.
template <bool LNLDS = false>
__device__ __forceinline__ void xattn_layer8(const float* __restrict__ W, const XLayerW& L, float* X, const X8Scratch& sc,
                                             const float* __restrict__ Kmat, const float* __restrict__ VT,
                                             const float* __restrict__ keybias, int n_key_pad, int self_key0,
                                             const uint8_t* rowvalid, uint8_t* novalid_s, int tid, WUnit1& u, const WNext1& nxt,
                                             const float* lnblk = nullptr, long long* prof = nullptr) {
#ifdef TB_PROFILE
#define TB_X8STAMP(i) do { if (prof && threadIdx.x == 0) prof[i] = clock64(); } while (0)
#else
#define TB_X8STAMP(i) do { (void)prof; } while (0)
#endif
    if (!LNLDS) lnblk = W + L.ln1_g;
    TB_X8STAMP(16);
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int head = wave >> 1, half = wave & 1;
    layernorm_tile8<LNLDS>(X, LDT, sc.S1, LDT, lnblk, lnblk + 128, tid);
    __syncthreads();
    TB_X8STAMP(17);
    WUnit1 u2;
    // q tile of this wave -> LDS (the pair needs both tiles of its head)
    {
        f32x4 qa = u.b;
        wmma1_pf(qa, u, sc.S1 + m * LDT + kq * 32, u2, wnext1(W + L.wo, W + L.bo, wave), lane);
        st4(cptr(sc.S2, LDT, wave, lane), qa);
    }
    __syncthreads();
    TB_X8STAMP(18);
    AttnPart part;
    {
        f32x4 q[2] = {lds4(cptr(sc.S2, LDT, 2 * head, lane)), lds4(cptr(sc.S2, LDT, 2 * head + 1, lane))};
        attention_half(q, Kmat, VT, keybias, n_key_pad, head, half, lane, self_key0 >= 0 ? self_key0 + m : -1, part);
        TB_X8STAMP(19);
        float* po = half ? sc.PO1 : sc.PO0;
        st4(cptr(po, LDT, 2 * head, lane), part.o[0]);
        st4(cptr(po, LDT, 2 * head + 1, lane), part.o[1]);
        if (kq == 0) {
            float* pml = sc.PML + ((half * 4 + head) * 16 + m) * 2;
            pml[0] = part.run_max;
            pml[1] = part.run_sum;
        }
    }
    __syncthreads();  // (also: every wave has consumed q from S2)
    TB_X8STAMP(20);
    // merge the two partial softmaxes; wave (head, half) finishes d-tile `half` of its head
    {
        const float* p0 = sc.PML + ((0 * 4 + head) * 16 + m) * 2;
        const float* p1 = sc.PML + ((1 * 4 + head) * 16 + m) * 2;
        const float m0 = p0[0], l0 = p0[1], m1 = p1[0], l1 = p1[1];
        const float mm = fmaxf(m0, m1);
        const float a0 = exp_neg(m0 - mm), a1 = exp_neg(m1 - mm);  // (-inf) - (-inf) -> 0, see exp_neg
        const float lsum = l0 * a0 + l1 * a1;
        const bool novalid = !(lsum > 0.f);
        const float inv = novalid ? 0.f : 1.0f / lsum;
        const int tile = 2 * head + half;
        const f32x4 o0 = lds4(cptr(sc.PO0, LDT, tile, lane)), o1 = lds4(cptr(sc.PO1, LDT, tile, lane));
        st4(cptr(sc.S2, LDT, tile, lane), (o0 * splat(a0) + o1 * splat(a1)) * splat(inv));
        if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
    }
    __syncthreads();
    TB_X8STAMP(21);
    // out-proj + residual
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, sc.S2 + m * LDT + kq * 32, u, wnext1(W + L.w1, W + L.b1, wave), lane);
        const bool nv = novalid_s[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 xo = lds4(px);
        st4(px, nv ? xo : xo + acc);
    }
    __syncthreads();
    TB_X8STAMP(22);
    layernorm_tile8<LNLDS>(X, LDT, sc.S1, LDT, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    TB_X8STAMP(23);
    {
        f32x4 acc = u.b;
        wmma1_pf(acc, u, sc.S1 + m * LDT + kq * 32, u2, wnext1(W + L.w2, W + L.b2, wave), lane);
        st4(cptr(sc.S2, LDT, wave, lane), relu4(acc));
    }
    __syncthreads();
    TB_X8STAMP(24);
    {
        f32x4 acc = u2.b;
        wmma1_pf(acc, u2, sc.S2 + m * LDT + kq * 32, u, nxt, lane);
        const bool rv = rowvalid[m] != 0;
        float* px = cptr(X, LDT, wave, lane);
        const f32x4 xo = lds4(px);
        st4(px, rv ? xo + acc : splat(0.f));
    }
    __syncthreads();
    TB_X8STAMP(25);
}

// K/V projection of the tile for one layer on 8 waves: wave w produces K tile w and V tile 8 + w.
template <bool LNLDS = false>
__device__ __forceinline__ void kv_project_tile8(const float* __restrict__ W, const XLayerW& L, const float* T, float* S1,
                                                 float* __restrict__ Kmat, float* __restrict__ VT, int n_key_pad, int tok0,
                                                 int n_real_rows, int tid, WUnit1& u, const WNext1& nxt, const float* lnblk = nullptr) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    if (!LNLDS) lnblk = W + L.ln1_g;
    layernorm_tile8<LNLDS>(T, LDT, S1, LDT, lnblk + 256, lnblk + 384, tid);
    __syncthreads();
    WUnit1 u2;
    const float* xr = S1 + m * LDT + kq * 32;
    f32x4 ak = u.b;
    wmma1_pf(ak, u, xr, u2, wnext1(W + L.wkv, W + L.bkv, 8 + wave), lane);
    f32x4 av = u2.b;
    wmma1_pf(av, u2, xr, u, nxt, lane);
    const bool real = m < n_real_rows;
    st4(Kmat + (size_t)(tok0 + m) * H + wave * 16 + kq * 4, real ? ak : splat(0.f));
    const int f0 = wave * 16 + kq * 4;
    const f32x4 v = real ? av : splat(0.f);
    VT[(size_t)(f0 + 0) * n_key_pad + tok0 + m] = v.x;
    VT[(size_t)(f0 + 1) * n_key_pad + tok0 + m] = v.y;
    VT[(size_t)(f0 + 2) * n_key_pad + tok0 + m] = v.z;
    VT[(size_t)(f0 + 3) * n_key_pad + tok0 + m] = v.w;
    __syncthreads();
}

// One GRU layer step on 8 waves: wave w owns features [16w, 16w+16) of every gate (tiles w, 8+w, 16+w).
__device__ __forceinline__ WNext1 gru_first8(const float* W, const GruLayerW& G, int wave) { return wnext1(W + G.wih, W + G.bih, wave); }

__device__ __forceinline__ void gru_layer8(const float* __restrict__ W, const GruLayerW& G, const float* Xin, const float* Hs, float* Out,
                                           const uint8_t* rowvalid, float* __restrict__ h_global, int n_real_rows, int tid,
                                           WUnit1& u, const WNext1& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int tr = wave, tz = 8 + wave, tn = 16 + wave;
    const float* xr = Xin + m * LDT + kq * 32;
    const float* hr = Hs + m * LDT + kq * 32;
    const float* wih = W + G.wih; const float* whh = W + G.whh; const float* bih = W + G.bih; const float* bhh = W + G.bhh;
    WUnit1 u2;
    f32x4 r = u.b;
    wmma1_pf(r, u, xr, u2, wnext1(whh, bhh, tr), lane);
    r += u2.b;
    wmma1_pf(r, u2, hr, u, wnext1(wih, bih, tz), lane);
    f32x4 z = u.b;
    wmma1_pf(z, u, xr, u2, wnext1(whh, bhh, tz), lane);
    z += u2.b;
    wmma1_pf(z, u2, hr, u, wnext1(wih, bih, tn), lane);
    f32x4 gin = u.b;
    wmma1_pf(gin, u, xr, u2, wnext1(whh, bhh, tn), lane);
    f32x4 ghn = u2.b;
    wmma1_pf(ghn, u2, hr, u, nxt, lane);
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
    st4(cptr(Out, LDT, wave, lane), hn);
    if (m < n_real_rows) st4(h_global + (size_t)m * H + wave * 16 + kq * 4, hn);
    __syncthreads();
}

}  // namespace tb
