// fp32-accurate Linear layers on the XDL matrix pipe from fp16-pair operands (gfx950).
//
// On gfx950 the fp32 MFMA (v_mfma_f32_16x16x4_f32) executes on the same ALUs as the VALU: its 32 cycles are ADDED to
// whatever VALU / LDS / VMEM instructions surround it (tools/microtests/mfma_valu_overlap.hip).  The 16-bit MFMAs run on
// the XDL pipe, issue every ~18 cycles and hide ~2 other instructions each.  A float is written as an fp16 pair,
//     x = x0 + 2^-11 x1,   x0 = fp16(x),  x1 = fp16((x - x0) * 2^11)        (x - x0 and the scaling are exact),
// about 23 significant bits, and a product w.x is taken as  w0.x0 + 2^-11 (w0.x1 + w1.x0)  with both sums accumulated in
// fp32 by v_mfma_f32_16x16x32_f16 (every fp16 product is exact in fp32).  On N(0,1) activations x uniform weights the
// result is closer to the fp64 dot product than the fp32 FMA chain it replaces (tools/microtests/bf16x3_gemm.hip:
// relative rms error 1.2e-7 vs 1.9e-7; a three-plane bf16 split reaches 7e-8 but needs 6 B per weight and is then bound
// by the 64 B/clk/CU L1 fill rate).  24 fp16 MFMAs (~430 cycles) replace the 64 fp32 MFMAs (2048 cycles) of one
// [32 features x 16 agents x 128 k] unit, no longer block the VALU, and move the same 4 B per weight.
// Range: |x| must stay below 65504 (fp16); activations of this model are O(1..100).
//
// Layouts
//   weights  : host-split, [tile of 16 outputs][chunk of 32 k][plane][64 lanes][8 fp16]; lane = kq*16 + row holds
//              W_plane[tile*16 + row][chunk*32 + kq*8 + 0..7]  (A operand of 16x16x32, natural k order)
//   inputs   : LDS "planes" [2][16 agents][LDP fp16]; lane (kq, m) reads 8 consecutive k of agent m (B operand)
//   outputs  : same C layout as the fp32 path (lane (kq, m): features tile*16 + kq*4 + 0..3 of agent m)
#pragma once
#include "tb_device.hpp"

namespace tb {

typedef _Float16 xhalf;
typedef xhalf xh8 __attribute__((ext_vector_type(8)));
typedef xhalf xh4 __attribute__((ext_vector_type(4)));

constexpr int NPL = 2;              // planes per operand
constexpr int LDP = 136;            // fp16 per plane row for 128-wide inputs (272 B: 16 rows x b128 reads hit 64 distinct banks)
constexpr int PLANE = TM * LDP;     // fp16 per plane
constexpr int PLANES_BYTES = NPL * PLANE * 2;  // 8704
constexpr int LDPC = 264;           // row length of the 256-wide concat planes
constexpr int PLANEC = TM * LDPC;

__device__ __forceinline__ f32x4 mfma_h(xh8 a, xh8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;

// fp16 pair of four floats: v = h + 2^-11 l
__device__ __forceinline__ void split2(f32x4 v, xh4& h, xh4& l) {
    h = xh4{(xhalf)v.x, (xhalf)v.y, (xhalf)v.z, (xhalf)v.w};
    const f32x4 r = (v - f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w}) * splat(SPLIT_SCALE);
    l = xh4{(xhalf)r.x, (xhalf)r.y, (xhalf)r.z, (xhalf)r.w};
}

// store four consecutive features of one agent row into the two planes
__device__ __forceinline__ void planes_store4(xhalf* P, int plane_stride, int ld, int row, int col, f32x4 v) {
    xh4 h, l;
    split2(v, h, l);
    xhalf* p = P + row * ld + col;
    *reinterpret_cast<xh4*>(p) = h;
    *reinterpret_cast<xh4*>(p + plane_stride) = l;
}
// C-layout helper: lane (kq, m) owns features tile*16 + kq*4 .. +3 of agent m
__device__ __forceinline__ void planes_store_c(xhalf* P, int tile, int lane, f32x4 v) {
    planes_store4(P, PLANE, LDP, lane & 15, tile * 16 + (lane >> 4) * 4, v);
}

// [16][128] fp32 LDS tile -> planes (256 threads, 2 float4 each)
__device__ __forceinline__ void tile_to_planes(const float* src, int lds_, xhalf* P, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        planes_store4(P, PLANE, LDP, row, c4, lds4(src + row * lds_ + c4));
    }
}

// LayerNorm of a [16][128] fp32 LDS tile written as planes (GEMM input)
template <bool PARAMS_IN_LDS = false>
__device__ __forceinline__ void layernorm_planes(const float* src, int lds_, xhalf* P, const float* __restrict__ g,
                                                 const float* __restrict__ b, int tid) {
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    const f32x4 a = lds4(src + row * lds_ + c0), c = lds4(src + row * lds_ + c0 + 4);
    f32x4 g0, g1, b0, b1;
    if (PARAMS_IN_LDS) {
        g0 = lds4_explicit(g + c0); g1 = lds4_explicit(g + c0 + 4); b0 = lds4_explicit(b + c0); b1 = lds4_explicit(b + c0 + 4);
    } else {
        g0 = ldg4(g + c0); g1 = ldg4(g + c0 + 4); b0 = ldg4(b + c0); b1 = ldg4(b + c0 + 4);
    }
    const float s = row16_sum((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w));
    const float mean = s * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean), dc = c - splat(mean);
    const float v = row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) +
                              (dc.z * dc.z + dc.w * dc.w));
    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    planes_store4(P, PLANE, LDP, row, c0, da * splat(rstd) * g0 + b0);
    planes_store4(P, PLANE, LDP, row, c0 + 4, dc * splat(rstd) * g1 + b1);
}

// ---------------------------------------------------------------------------------------------
// weight units: 2 output tiles x 4 chunks (128 k) x 2 planes = 16 fragments of 8 fp16 per lane (64 VGPRs) + bias
// ---------------------------------------------------------------------------------------------
struct WUnitX {
    xh8 w[2][4][NPL];  // [tile][chunk][plane]
    f32x4 b[2];
};

struct WNextX {
    const xhalf* wpk;   // packed Linear
    const float* bias;   // or nullptr
    int tile_a, tile_b;
    int nchunk;          // chunks per output tile of this Linear (K / 32)
    int c0;              // first chunk of this unit
};

__device__ __forceinline__ WNextX wnextx(const float* arena, uint32_t off, const float* bias, int tile_a, int tile_b, int nchunk = 4,
                                          int c0 = 0) {
    return WNextX{reinterpret_cast<const xhalf*>(arena + off), bias, tile_a, tile_b, nchunk, c0};
}
__device__ __forceinline__ WNextX wstdx(const float* arena, uint32_t off, const float* bias, int wave) {
    return wnextx(arena, off, bias, 2 * wave, 2 * wave + 1);
}

__device__ __forceinline__ const xh8* wfragx(const WNextX& n, int tile, int lane) {
    return reinterpret_cast<const xh8*>(n.wpk + ((size_t)(tile * n.nchunk + n.c0) * NPL) * 512 + lane * 8);
}

__device__ __forceinline__ void wloadx(WUnitX& u, const WNextX& n, int lane) {
    const xh8* pa = wfragx(n, n.tile_a, lane);
    const xh8* pb = wfragx(n, n.tile_b, lane);
    TB_SCHED_FENCE();
    const int bo = (lane >> 4) * 4;
    u.b[0] = n.bias ? ldg4(n.bias + n.tile_a * 16 + bo) : splat(0.f);
    u.b[1] = n.bias ? ldg4(n.bias + n.tile_b * 16 + bo) : splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            u.w[0][c][p] = pa[(c * NPL + p) * 64];
            u.w[1][c][p] = pb[(c * NPL + p) * 64];
        }
    TB_SCHED_FENCE();
}

__device__ __forceinline__ xh8 ldsb8(const xhalf* p) { return *reinterpret_cast<const xh8*>(p); }

// acc_{a,b} += unit . X^T from planes; requests the next unit in the MFMAs' shadow (two VMEM per three MFMAs).
//   bp : this lane's B base = P + m*ld + kq*8 (+ chunk offset of the unit); plane_stride in fp16
// Per chunk and tile three products; the two cross terms go to a second accumulator that is scaled by 2^-11 at the end.
__device__ __forceinline__ void wmmax_pf(f32x4& acc_a, f32x4& acc_b, const WUnitX& u, const xhalf* bp, int plane_stride, WUnitX& un,
                                         const WNextX& n, int lane) {
    const xh8* pa = wfragx(n, n.tile_a, lane);
    const xh8* pb = wfragx(n, n.tile_b, lane);
    const int bo = (lane >> 4) * 4;
    const float* ba = n.bias ? n.bias + n.tile_a * 16 + bo : reinterpret_cast<const float*>(n.wpk);
    const float* bb = n.bias ? n.bias + n.tile_b * 16 + bo : reinterpret_cast<const float*>(n.wpk);
    TB_SCHED_FENCE();
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    un.b[0] = ldg4(ba);
    un.b[1] = ldg4(bb);
    f32x4 mid_a = splat(0.f), mid_b = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            un.w[0][c][p] = pa[(c * NPL + p) * 64];
            un.w[1][c][p] = pb[(c * NPL + p) * 64];
        }
        mid_a = mfma_h(u.w[0][c][0], x[c][1], mid_a);
        mid_b = mfma_h(u.w[1][c][0], x[c][1], mid_b);
        mid_a = mfma_h(u.w[0][c][1], x[c][0], mid_a);
        mid_b = mfma_h(u.w[1][c][1], x[c][0], mid_b);
        acc_a = mfma_h(u.w[0][c][0], x[c][0], acc_a);
        acc_b = mfma_h(u.w[1][c][0], x[c][0], acc_b);
    }
    // pin the order: 8 LDS reads + 2 bias loads, then 8 x (2 MFMA, 1 weight load, 1 MFMA, 1 weight load)
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    acc_a += mid_a * splat(SPLIT_INV);
    acc_b += mid_b * splat(SPLIT_INV);
    if (!n.bias) {
        un.b[0] = splat(0.f);
        un.b[1] = splat(0.f);
    }
}

// offsets (in floats, into the same arena) of the fp16-pair packed Linears of one cross-attention layer / GRU layer
struct XLayerX {
    uint32_t wq, wkv, wo, w1, w2;
};
struct GruLayerX {
    uint32_t wih, whh;
};

__device__ __forceinline__ WNextX xlayer_first_x(const float* W, const XLayerW& L, const XLayerX& LX, int wave) {
    return wstdx(W, LX.wq, W + L.bq, wave);
}
__device__ __forceinline__ WNextX kvproj_first_x(const float* W, const XLayerW& L, const XLayerX& LX, int wave) {
    return wstdx(W, LX.wkv, W + L.bkv, wave);
}
__device__ __forceinline__ WNextX gru_first_x(const float* W, const GruLayerW& G, const GruLayerX& GX, int wave) {
    return wstdx(W, GX.wih, W + G.bih, wave);
}

// ---------------------------------------------------------------------------------------------
// One pre-LN cross-attention layer, GEMMs on XDL, attention (QK / PV) unchanged on the fp32 MFMA.
//   X : [16][LDT] fp32 residual stream (LDS);  P1, P2 : plane buffers (LN output / attention output + FFN hidden)
// ---------------------------------------------------------------------------------------------
template <bool LNLDS = false>
__device__ __forceinline__ void xattn_layer_x(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1,
                                              xhalf* P2, const float* __restrict__ Kmat, const float* __restrict__ VT,
                                              const float* __restrict__ keybias, int n_key_pad, int self_key0, const uint8_t* rowvalid,
                                              uint8_t* novalid_s, int tid, WUnitX& u, const WNextX& nxt, const float* lnblk = nullptr) {
    if (!LNLDS) lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    AttnPre apre;
    attention_prefetch(apre, Kmat, VT, keybias, n_key_pad, wave, lane);
    layernorm_planes<LNLDS>(X, LDT, P1, lnblk, lnblk + 128, tid);
    __syncthreads();
    WUnitX u2;
    f32x4 q[2] = {u.b[0], u.b[1]};
    wmmax_pf(q[0], q[1], u, b1, PLANE, u2, wstdx(W, LX.wo, W + L.bo, wave), lane);
    f32x4 o[2];
    const bool novalid = attention_head(q, apre, Kmat, VT, keybias, n_key_pad, wave, lane, self_key0 >= 0 ? self_key0 + m : -1, o);
    planes_store_c(P2, 2 * wave, lane, o[0]);
    planes_store_c(P2, 2 * wave + 1, lane, o[1]);
    if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
    __syncthreads();
    {
        f32x4 acc[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(acc[0], acc[1], u2, b2, PLANE, u, wstdx(W, LX.w1, W + L.b1, wave), lane);
        const bool nv = novalid_s[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X, LDT, 2 * wave + t, lane);
            const f32x4 xo = lds4(px);
            st4(px, nv ? xo : xo + acc[t]);
        }
    }
    __syncthreads();
    layernorm_planes<LNLDS>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    {
        f32x4 acc[2] = {u.b[0], u.b[1]};
        wmmax_pf(acc[0], acc[1], u, b1, PLANE, u2, wstdx(W, LX.w2, W + L.b2, wave), lane);
        planes_store_c(P2, 2 * wave, lane, relu4(acc[0]));
        planes_store_c(P2, 2 * wave + 1, lane, relu4(acc[1]));
    }
    __syncthreads();
    {
        f32x4 acc[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(acc[0], acc[1], u2, b2, PLANE, u, nxt, lane);
        const bool rv = rowvalid[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X, LDT, 2 * wave + t, lane);
            const f32x4 xo = lds4(px);
            st4(px, rv ? xo + acc[t] : splat(0.f));
        }
    }
    __syncthreads();
}

// K/V projection of the tile's tokens for one layer (LN_tgt -> in_proj rows 128:384), outputs fp32 as the fp32 path
template <bool LNLDS = false>
__device__ __forceinline__ void kv_project_tile_x(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, const float* T,
                                                  xhalf* P1, float* __restrict__ Kmat, float* __restrict__ VT, int n_key_pad, int tok0,
                                                  int n_real_rows, int tid, WUnitX& u, const WNextX& nxt, const float* lnblk = nullptr) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    if (!LNLDS) lnblk = W + L.ln1_g;
    layernorm_planes<LNLDS>(T, LDT, P1, lnblk + 256, lnblk + 384, tid);
    __syncthreads();
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    WUnitX u2;
    f32x4 ak[2] = {u.b[0], u.b[1]};
    wmmax_pf(ak[0], ak[1], u, b1, PLANE, u2, wnextx(W, LX.wkv, W + L.bkv, 8 + 2 * wave, 8 + 2 * wave + 1), lane);
    f32x4 av[2] = {u2.b[0], u2.b[1]};
    wmmax_pf(av[0], av[1], u2, b1, PLANE, u, nxt, lane);
    const bool real = m < n_real_rows;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        st4(Kmat + (size_t)(tok0 + m) * H + (2 * wave + t) * 16 + kq * 4, real ? ak[t] : splat(0.f));
        const int f0 = (2 * wave + t) * 16 + kq * 4;
        const f32x4 v = real ? av[t] : splat(0.f);
        VT[(size_t)(f0 + 0) * n_key_pad + tok0 + m] = v.x;
        VT[(size_t)(f0 + 1) * n_key_pad + tok0 + m] = v.y;
        VT[(size_t)(f0 + 2) * n_key_pad + tok0 + m] = v.z;
        VT[(size_t)(f0 + 3) * n_key_pad + tok0 + m] = v.w;
    }
    __syncthreads();
}

// One GRU layer step: inputs as planes (XinP, HsP) + the fp32 previous hidden (Hs) for the convex update.
//   OutP : planes for the next layer's input, or nullptr;  Out : fp32 LDS tile, or nullptr
__device__ __forceinline__ void gru_layer_x(const float* __restrict__ W, const GruLayerW& G, const GruLayerX& GX, const xhalf* XinP,
                                            const xhalf* HsP, const float* Hs, xhalf* OutP, float* Out, const uint8_t* rowvalid,
                                            float* __restrict__ h_global, int n_real_rows, int tid, WUnitX& u, const WNextX& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int ta = 2 * wave, tb_ = 2 * wave + 1;
    const xhalf* xr = XinP + m * LDP + kq * 8;
    const xhalf* hr = HsP + m * LDP + kq * 8;
    const float* bih = W + G.bih;
    const float* bhh = W + G.bhh;
    WUnitX u2;
    f32x4 r[2] = {u.b[0], u.b[1]};
    wmmax_pf(r[0], r[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, ta, tb_), lane);
    r[0] += u2.b[0];
    r[1] += u2.b[1];
    wmmax_pf(r[0], r[1], u2, hr, PLANE, u, wnextx(W, GX.wih, bih, 8 + ta, 8 + tb_), lane);
    f32x4 z[2] = {u.b[0], u.b[1]};
    wmmax_pf(z[0], z[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, 8 + ta, 8 + tb_), lane);
    z[0] += u2.b[0];
    z[1] += u2.b[1];
    wmmax_pf(z[0], z[1], u2, hr, PLANE, u, wnextx(W, GX.wih, bih, 16 + ta, 16 + tb_), lane);
    f32x4 gin[2] = {u.b[0], u.b[1]};
    wmmax_pf(gin[0], gin[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, 16 + ta, 16 + tb_), lane);
    f32x4 ghn[2] = {u2.b[0], u2.b[1]};
    wmmax_pf(ghn[0], ghn[1], u2, hr, PLANE, u, nxt, lane);
    const bool rv = rowvalid[m] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = 2 * wave + t;
        const f32x4 hold = lds4(Hs + m * LDT + tile * 16 + kq * 4);
        f32x4 hn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rg = sigmoidf_(r[t][q]);
            const float zg = sigmoidf_(z[t][q]);
            const float ng = tanhf_(gin[t][q] + rg * ghn[t][q]);
            hn[q] = rv ? (1.0f - zg) * ng + zg * hold[q] : 0.f;
        }
        if (OutP) planes_store_c(OutP, tile, lane, hn);
        if (Out) st4(cptr(Out, LDT, tile, lane), hn);
        if (m < n_real_rows) st4(h_global + (size_t)m * H + tile * 16 + kq * 4, hn);
    }
    __syncthreads();
}

}  // namespace tb
