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

// Two builds of this header exist (one translation unit each): the default fp16-pair mode above (namespace tb::xh) and, with
// -DTB_XDL_BF16 / tb_stepx_bf16_kernels.hip, a plain bf16 mode (namespace tb::xb): ONE bf16 plane per operand, fp32 accumulate --
// the "bf16 MFMA inputs" configurations of BASELINE.json (configs 4/5); no fp32-parity claim there.
// A third build (-DTB_XDL_BF16 -DTB_XDL_W3 / tb_stepx_bf16w3_kernels.hip, namespace tb::xb3) is the bf16 mode carved for THREE
// workgroups per CU (launches of more than 512 tiles, e.g. K = 6 futures x 32 scenes = 768): a weight unit is not prefetched into
// registers one stage ahead -- `wloadx` only records WHERE the unit is (and fetches its bias), the GEMM that consumes it loads the
// fragments itself -- which takes 32 of the 40 VGPRs of every unit in flight out of the kernel (<= 168 VGPRs = three waves per
// SIMD); the latency the prefetch used to hide is hidden by the two other workgroups of the CU.
#if defined(TB_XDL_W3) && !defined(TB_XDL_BF16)
#error "TB_XDL_W3 is a variant of the bf16 build (the fp16-pair twin was measured and lost: profiles/r03_experiments_not_kept.txt)"
#endif
#if defined(TB_XDL_AW) && (!defined(TB_XDL_BF16) || defined(TB_XDL_W3))
#error "TB_XDL_AW is a variant of the plain bf16 build (one workgroup per CU, 213 VGPRs: two waves per SIMD fit)"
#endif
#ifdef TB_XDL_AW
#define TB_XNS xba
#elif defined(TB_XDL_W3)
#define TB_XNS xb3
#elif defined(TB_XDL_BF16)
#define TB_XNS xb
#else
#define TB_XNS xh
#endif

namespace tb {
namespace TB_XNS {

#ifdef TB_XDL_BF16
typedef __bf16 xhalf;
constexpr int NPL = 1;              // planes per operand
#else
typedef _Float16 xhalf;
constexpr int NPL = 2;              // planes per operand
#endif
typedef xhalf xh8 __attribute__((ext_vector_type(8)));
typedef xhalf xh4 __attribute__((ext_vector_type(4)));
typedef xhalf xh2 __attribute__((ext_vector_type(2)));

#ifndef TB_LDP
#define TB_LDP 136
#endif
constexpr int LDP = TB_LDP;         // fp16 per plane row for 128-wide inputs (272 B; the scene encoders' translation unit: 288 B, see there)
constexpr int PLANE = TM * LDP;     // fp16 per plane
constexpr int PLANES_BYTES = NPL * PLANE * 2;  // 8704
constexpr int LDPC = 264;           // row length of the 256-wide concat planes
constexpr int PLANEC = TM * LDPC;

#ifdef TB_XDL_BF16
__device__ __forceinline__ f32x4 mfma_h(xh8 a, xh8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#else
__device__ __forceinline__ f32x4 mfma_h(xh8 a, xh8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#endif
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;
constexpr int P1 = NPL - 1;  // index of the low plane (aliases plane 0 in the single-plane build, where it is never used)

// Range guard of the fp16-pair mode: an operand with |x| >= 65504 becomes inf in its high plane and NaN in its low one, and a
// NaN does not always survive to the outputs (ReLU and the softmax clamp squash it), so the overflow is FLAGGED: one sticky word
// per translation unit, fetched and cleared by tb_check_status.  Two forms of the check on the GEMM-input stores:
//   * `amax` given (the step kernels): the running max of |x| is kept in ONE register per thread -- two v_max3 per four values,
//     no branch, no memory operation -- and compared once at the end of the launch.  (The first form, a conditional store of the
//     flag at every site, cost 4 % of the fused launch: a possible VMEM store in the instruction stream makes every later
//     `s_waitcnt vmcnt` of the weight prefetch conservative.)
//   * no `amax` (the one-time encoders and hoists): the flag is stored where the overflow happens.
// Unchecked by construction: the softmax probabilities (in [0, 1]), LayerNorm outputs (|x^| <= sqrt(127); gamma / beta are bounded
// when the weights are loaded, tb_finalize_weights), GRU states (|h| <= 1).  The bf16 build has fp32's range.
#ifndef TB_XDL_BF16
static __device__ unsigned int g_range_flag;
#endif
constexpr float XH_MAX = 65504.0f;

// fp16 pair of four floats: v = h + 2^-11 l
// The two forms as argument types: RangeFlag (default: flag at the site) and RangeMax (accumulate; the caller flushes).
struct RangeFlag {};
struct RangeMax {
    float v = 0.f;  // max |x| over the checked operands: one v_max3_f32 with |.| source modifiers per two values (a NaN operand is
                    // skipped by max -- it can only follow an overflow or come in with the data -- the first overflow itself is caught)
};
__device__ __forceinline__ void range_note(const RangeFlag&, const f32x4& v) {
#if !defined(TB_XDL_BF16) && !defined(TB_NO_RANGE_CHECK)
    const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (m >= XH_MAX) g_range_flag = 1u;
#else
    (void)v;
#endif
}
__device__ __forceinline__ void range_note(RangeMax& r, const f32x4& v) {
#if !defined(TB_XDL_BF16) && !defined(TB_NO_RANGE_CHECK)
    r.v = fmaxf(fmaxf(r.v, fabsf(v.x)), fabsf(v.y));
    r.v = fmaxf(fmaxf(r.v, fabsf(v.z)), fabsf(v.w));
#else
    (void)r; (void)v;
#endif
}

template <bool CHECK = true, class R = RangeFlag>
__device__ __forceinline__ void split2(f32x4 v, xh4& h, xh4& l, R&& amax = R{}) {
    if (CHECK) range_note(amax, v);
    h = xh4{(xhalf)v.x, (xhalf)v.y, (xhalf)v.z, (xhalf)v.w};
    if (NPL == 2) {
#ifndef TB_XDL_BF16
        // the rounded values are read back from the PACKED halves (one v_cvt_pk_f16_f32 per pair, then v_cvt_f32_f16 on word 0 / word 1):
        // without the opaque copy the compiler converts every value a second time on its own (v_cvt_f16_f32) for the round trip
        unsigned int w0, w1;
        {
            const xh2 a = xh2{h.x, h.y}, b = xh2{h.z, h.w};
            __builtin_memcpy(&w0, &a, 4);
            __builtin_memcpy(&w1, &b, 4);
            asm("" : "+v"(w0), "+v"(w1));
            xh2 a2, b2;
            __builtin_memcpy(&a2, &w0, 4);
            __builtin_memcpy(&b2, &w1, 4);
            h = xh4{a2.x, a2.y, b2.x, b2.y};
        }
        // l = fp16((v - h) * 2^11) in two mixed-precision fmas per value and nothing else: t = fma(h, -1, v) with h read as fp16 straight
        // from its packed half (exact), then fp16(fma(t, 2^11, 0)) written into the low / high half of the result register
        // (v_fma_mixlo / mixhi_f16: one rounding, as v_cvt_pk_f16_f32 of the exact product had).  Same bits as convert-back, subtract,
        // scale, convert (v_cvt_f32_f16 + v_sub_f32 + v_mul_f32 per value + a v_cvt_pk per pair); the compiler does not form the mixed
        // fmas from the C expression, so they are written out.
        const float sc = SPLIT_SCALE;
        float t0, t1, t2, t3;
        unsigned int l0, l1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(w0), "v"(v.x));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(w0), "v"(v.y));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t2) : "v"(w1), "v"(v.z));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t3) : "v"(w1), "v"(v.w));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l0) : "v"(t0), "s"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l0) : "v"(t1), "s"(sc));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l1) : "v"(t2), "s"(sc));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l1) : "v"(t3), "s"(sc));
        {
            xh2 a3, b3;
            __builtin_memcpy(&a3, &l0, 4);
            __builtin_memcpy(&b3, &l1, 4);
            l = xh4{a3.x, a3.y, b3.x, b3.y};
        }
#else
        const f32x4 r = (v - f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w}) * splat(SPLIT_SCALE);
        l = xh4{(xhalf)r.x, (xhalf)r.y, (xhalf)r.z, (xhalf)r.w};
#endif
    } else {
        l = h;  // (unused)
    }
}

// end of a step launch: raise the sticky flag if any checked operand of this thread left the range
__device__ __forceinline__ void range_flush(const RangeMax& amax) {
#if !defined(TB_XDL_BF16) && !defined(TB_NO_RANGE_CHECK)
    if (amax.v >= XH_MAX) g_range_flag = 1u;
#else
    (void)amax;
#endif
}

// store four consecutive features of one agent row into the two planes
template <bool CHECK = true, class R = RangeFlag>
__device__ __forceinline__ void planes_store4(xhalf* P, int plane_stride, int ld, int row, int col, f32x4 v, R&& amax = R{}) {
    xh4 h, l;
    split2<CHECK>(v, h, l, amax);
    xhalf* p = P + row * ld + col;
    *reinterpret_cast<xh4*>(p) = h;
    if (NPL == 2) *reinterpret_cast<xh4*>(p + plane_stride) = l;
}
// C-layout helper: lane (kq, m) owns features tile*16 + kq*4 .. +3 of agent m
template <bool CHECK = true, class R = RangeFlag>
__device__ __forceinline__ void planes_store_c(xhalf* P, int tile, int lane, f32x4 v, R&& amax = R{}) {
    planes_store4<CHECK>(P, PLANE, LDP, lane & 15, tile * 16 + (lane >> 4) * 4, v, amax);
}

// [16][128] fp32 LDS tile -> planes (256 threads, 2 float4 each)
template <bool CHECK = true, class R = RangeFlag>
__device__ __forceinline__ void tile_to_planes(const float* src, int lds_, xhalf* P, int tid, R&& amax = R{}) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        planes_store4<CHECK>(P, PLANE, LDP, row, c4, lds4(src + row * lds_ + c4), amax);
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
#ifdef TB_DBG_LN_ORDER  // (diagnosis builds only, VERDICT r04 task 4 (d): another association of the two row sums)
    const float s = row16_sum(((((((a.x + a.y) + a.z) + a.w) + c.x) + c.y) + c.z) + c.w);
#else
    const float s = row16_sum((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w));
#endif
    const float mean = s * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean), dc = c - splat(mean);
#ifdef TB_DBG_LN_ORDER
    const float v = row16_sum(((((((da.x * da.x + da.y * da.y) + da.z * da.z) + da.w * da.w) + dc.x * dc.x) + dc.y * dc.y) + dc.z * dc.z) + dc.w * dc.w);
#else
    const float v = row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) +
                              (dc.z * dc.z + dc.w * dc.w));
#endif
    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    planes_store4<false>(P, PLANE, LDP, row, c0, da * splat(rstd) * g0 + b0);  // (bounded by the parameters: checked at load time)
    planes_store4<false>(P, PLANE, LDP, row, c0 + 4, dc * splat(rstd) * g1 + b1);
}

// The same LayerNorm in two steps, for an input that is normalised more than once with different parameters (k_polyline_fused8<true>:
// the block input of LN_tgt is fixed over the layers, and at layer 0 it is LN1's input as well): ln_stats takes the statistics of a
// row segment held in registers (a = columns c0 .. c0 + 3, c = c0 + 4 .. c0 + 7 of row tid >> 4, c0 = (tid & 15) * 8) and leaves the
// centred values and 1 / std; ln_apply scales, shifts, splits and stores.  The arithmetic and its order are those of
// layernorm_planes -- same bits -- and the two 16-lane reductions and the division are paid once.
struct LnSeg {
    f32x4 da, dc;
    float rstd;
};
__device__ __forceinline__ LnSeg ln_stats(const f32x4& a, const f32x4& c) {
    const float s = row16_sum((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w));
    const float mean = s * (1.0f / 128.0f);
    LnSeg r;
    r.da = a - splat(mean);
    r.dc = c - splat(mean);
    const float v = row16_sum((r.da.x * r.da.x + r.da.y * r.da.y) + (r.da.z * r.da.z + r.da.w * r.da.w) + (r.dc.x * r.dc.x + r.dc.y * r.dc.y) +
                              (r.dc.z * r.dc.z + r.dc.w * r.dc.w));
    r.rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    return r;
}
__device__ __forceinline__ void ln_apply(const LnSeg& r, xhalf* P, const float* __restrict__ g, const float* __restrict__ b, int tid) {
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    const f32x4 g0 = ldg4(g + c0), g1 = ldg4(g + c0 + 4), b0 = ldg4(b + c0), b1 = ldg4(b + c0 + 4);
    planes_store4<false>(P, PLANE, LDP, row, c0, r.da * splat(r.rstd) * g0 + b0);
    planes_store4<false>(P, PLANE, LDP, row, c0 + 4, r.dc * splat(r.rstd) * g1 + b1);
}

// The same LayerNorm for N tiles at once (tile t at src + t * src_stride floats, its planes at P + t * p_stride fp16): the N
// independent latency chains (LDS read -> two 16-lane reductions -> rsqrt -> pair split -> LDS write) are written side by side so
// that they overlap in one wave's instruction stream; per tile the arithmetic and its order are those of layernorm_planes.
template <int N>
__device__ __forceinline__ void layernorm_planes_n(const float* src, int src_stride, xhalf* P, int p_stride, const float* __restrict__ g,
                                                   const float* __restrict__ b, int tid) {
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    f32x4 a[N], c[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
        a[t] = lds4(src + t * src_stride + row * LDT + c0);
        c[t] = lds4(src + t * src_stride + row * LDT + c0 + 4);
    }
    const f32x4 g0 = ldg4(g + c0), g1 = ldg4(g + c0 + 4), b0 = ldg4(b + c0), b1 = ldg4(b + c0 + 4);
    float s[N];
#pragma unroll
    for (int t = 0; t < N; ++t) s[t] = (a[t].x + a[t].y) + (a[t].z + a[t].w) + (c[t].x + c[t].y) + (c[t].z + c[t].w);
#pragma unroll
    for (int t = 0; t < N; ++t) s[t] = row16_sum(s[t]);
    f32x4 da[N], dc[N];
    float v[N];
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const float mean = s[t] * (1.0f / 128.0f);
        da[t] = a[t] - splat(mean);
        dc[t] = c[t] - splat(mean);
        v[t] = (da[t].x * da[t].x + da[t].y * da[t].y) + (da[t].z * da[t].z + da[t].w * da[t].w) + (dc[t].x * dc[t].x + dc[t].y * dc[t].y) +
               (dc[t].z * dc[t].z + dc[t].w * dc[t].w);
    }
#pragma unroll
    for (int t = 0; t < N; ++t) v[t] = row16_sum(v[t]);
#pragma unroll
    for (int t = 0; t < N; ++t) {
        const float rstd = 1.0f / sqrtf(v[t] * (1.0f / 128.0f) + LN_EPS);
        planes_store4<false>(P + t * p_stride, PLANE, LDP, row, c0, da[t] * splat(rstd) * g0 + b0);
        planes_store4<false>(P + t * p_stride, PLANE, LDP, row, c0 + 4, dc[t] * splat(rstd) * g1 + b1);
    }
}

// ---------------------------------------------------------------------------------------------
// weight units: 2 output tiles x 4 chunks (128 k) x 2 planes = 16 fragments of 8 fp16 per lane (64 VGPRs) + bias
// ---------------------------------------------------------------------------------------------
struct WNextX {
    const xhalf* wpk;   // packed Linear
    const float* bias;   // or nullptr
    int tile_a, tile_b;
    int nchunk;          // chunks per output tile of this Linear (K / 32)
    int c0;              // first chunk of this unit
};

#ifdef TB_XDL_W3
struct WUnitX {
    WNextX at;           // where the unit's fragments are: loaded by the GEMM that consumes them
    f32x4 b[2];
};
#else
struct WUnitX {
    xh8 w[2][4][NPL];  // [tile][chunk][plane]
    f32x4 b[2];
};
#endif

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

#ifdef TB_XDL_W3
__device__ __forceinline__ void wloadx(WUnitX& u, const WNextX& n, int lane) {
    const int bo = (lane >> 4) * 4;
    u.at = n;
    u.b[0] = n.bias ? ldg4(n.bias + n.tile_a * 16 + bo) : splat(0.f);
    u.b[1] = n.bias ? ldg4(n.bias + n.tile_b * 16 + bo) : splat(0.f);
}
__device__ __forceinline__ xh8 ldsb8(const xhalf* p) { return *reinterpret_cast<const xh8*>(p); }
// acc_{a,b} += unit . X^T: the unit's fragments are requested here, in front of the LDS reads of the B operand
__device__ __forceinline__ void wmmax(f32x4& acc_a, f32x4& acc_b, const WUnitX& u, const xhalf* bp, int plane_stride) {
    const int lane = threadIdx.x & 63;
    const xh8* pa = wfragx(u.at, u.at.tile_a, lane);
    const xh8* pb = wfragx(u.at, u.at.tile_b, lane);
    xh8 wa[4][NPL], wb[4][NPL], x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            wa[c][p] = pa[(c * NPL + p) * 64];
            wb[c][p] = pb[(c * NPL + p) * 64];
        }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid_a = splat(0.f), mid_b = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
            mid_a = mfma_h(wa[c][0], x[c][P1], mid_a);
            mid_b = mfma_h(wb[c][0], x[c][P1], mid_b);
            mid_a = mfma_h(wa[c][P1], x[c][0], mid_a);
            mid_b = mfma_h(wb[c][P1], x[c][0], mid_b);
        }
        acc_a = mfma_h(wa[c][0], x[c][0], acc_a);
        acc_b = mfma_h(wb[c][0], x[c][0], acc_b);
    }
    if (NPL == 2) {
        acc_a += mid_a * splat(SPLIT_INV);
        acc_b += mid_b * splat(SPLIT_INV);
    }
}
__device__ __forceinline__ void wmmax_pf(f32x4& acc_a, f32x4& acc_b, const WUnitX& u, const xhalf* bp, int plane_stride, WUnitX& un,
                                         const WNextX& n, int lane) {
    const WUnitX cur = u;  // (`un` may alias `u`)
    wloadx(un, n, lane);
    wmmax(acc_a, acc_b, cur, bp, plane_stride);
}
#else
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

template <bool SWAP>
__device__ __forceinline__ f32x4 mm_sw(const xh8& w, const xh8& x, const f32x4& acc) {
    return SWAP ? mfma_h(x, w, acc) : mfma_h(w, x, acc);
}

// acc_{a,b} += unit . X^T from planes; requests the next unit in the MFMAs' shadow (two VMEM per three MFMAs).
//   bp : this lane's B base = P + m*ld + kq*8 (+ chunk offset of the unit); plane_stride in fp16
// Per chunk and tile three products; the two cross terms go to a second accumulator that is scaled by 2^-11 at the end.
// SWAP: the MFMA operands change places, so the accumulators hold the TRANSPOSED tile -- lane (kq, m): tokens 4 kq + r of output
// feature tile * 16 + m (the same dot products in the same order: same bits) -- which is the element order of a V fragment
// (k_polyline_fused writes them with one 8-byte store per fragment half instead of four 2-byte scatters); the caller adds the bias.
template <bool SWAP = false>
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
        if (NPL == 2) {
            mid_a = mm_sw<SWAP>(u.w[0][c][0], x[c][P1], mid_a);
            mid_b = mm_sw<SWAP>(u.w[1][c][0], x[c][P1], mid_b);
            mid_a = mm_sw<SWAP>(u.w[0][c][P1], x[c][0], mid_a);
            mid_b = mm_sw<SWAP>(u.w[1][c][P1], x[c][0], mid_b);
        }
        acc_a = mm_sw<SWAP>(u.w[0][c][0], x[c][0], acc_a);
        acc_b = mm_sw<SWAP>(u.w[1][c][0], x[c][0], acc_b);
    }
    // pin the order: the LDS reads + 2 bias loads, then the weight loads spread under the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * NPL, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (NPL == 2) {  // 8 x (2 MFMA, 1 load, 1 MFMA, 1 load)
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    if (NPL == 2) {
        acc_a += mid_a * splat(SPLIT_INV);
        acc_b += mid_b * splat(SPLIT_INV);
    }
    if (!n.bias) {
        un.b[0] = splat(0.f);
        un.b[1] = splat(0.f);
    }
}

// the same unit without a follow-up request (the caller issues the next unit itself, see attention_head_x)
template <bool SWAP = false>
__device__ __forceinline__ void wmmax(f32x4& acc_a, f32x4& acc_b, const WUnitX& u, const xhalf* bp, int plane_stride) {
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid_a = splat(0.f), mid_b = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
            mid_a = mm_sw<SWAP>(u.w[0][c][0], x[c][P1], mid_a);
            mid_b = mm_sw<SWAP>(u.w[1][c][0], x[c][P1], mid_b);
            mid_a = mm_sw<SWAP>(u.w[0][c][P1], x[c][0], mid_a);
            mid_b = mm_sw<SWAP>(u.w[1][c][P1], x[c][0], mid_b);
        }
        acc_a = mm_sw<SWAP>(u.w[0][c][0], x[c][0], acc_a);
        acc_b = mm_sw<SWAP>(u.w[1][c][0], x[c][0], acc_b);
    }
    if (NPL == 2) {
        acc_a += mid_a * splat(SPLIT_INV);
        acc_b += mid_b * splat(SPLIT_INV);
    }
}

#ifndef TB_XDL_W3
// ---------------------------------------------------------------------------------------------
// ONE-tile weight units (round 5, the eight-wave polyline encoder k_polyline_fused8): wave w of eight owns output tile w of a
// 128 -> 128 Linear, so a unit is 1 output tile x 4 chunks x NPL planes = 8 fragments (32 VGPRs with fp16 pairs) + bias.  The
// products of an output element are taken in the order of wmmax (per chunk: the two cross products, then the high product; the cross
// sum scaled once at the end): the same bits as the two-tile units give.
// ---------------------------------------------------------------------------------------------
struct WNext1X {
    const xhalf* wpk;
    const float* bias;   // or nullptr
    int tile, nchunk, c0;
};
struct WUnit1X {
    xh8 w[4][NPL];  // [chunk][plane]
    f32x4 b;
};
__device__ __forceinline__ WNext1X wnext1x(const float* arena, uint32_t off, const float* bias, int tile, int nchunk = 4, int c0 = 0) {
    return WNext1X{reinterpret_cast<const xhalf*>(arena + off), bias, tile, nchunk, c0};
}
__device__ __forceinline__ const xh8* wfrag1x(const WNext1X& n, int lane) {
    return reinterpret_cast<const xh8*>(n.wpk + ((size_t)(n.tile * n.nchunk + n.c0) * NPL) * 512 + lane * 8);
}
__device__ __forceinline__ void wload1x(WUnit1X& u, const WNext1X& n, int lane) {
    const xh8* pa = wfrag1x(n, lane);
    TB_SCHED_FENCE();
    u.b = n.bias ? ldg4(n.bias + n.tile * 16 + (lane >> 4) * 4) : splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) u.w[c][p] = pa[(c * NPL + p) * 64];
    TB_SCHED_FENCE();
}
// acc += unit . X^T for one row tile (bp: this lane's B base, see wmmax_pf)
template <bool SWAP = false>
__device__ __forceinline__ void wmma1x(f32x4& acc, const WUnit1X& u, const xhalf* bp, int plane_stride) {
    xh8 x[4][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) x[c][p] = ldsb8(bp + p * plane_stride + c * 32);
    f32x4 mid = splat(0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
            mid = mm_sw<SWAP>(u.w[c][0], x[c][P1], mid);
            mid = mm_sw<SWAP>(u.w[c][P1], x[c][0], mid);
        }
        acc = mm_sw<SWAP>(u.w[c][0], x[c][0], acc);
    }
    if (NPL == 2) acc += mid * splat(SPLIT_INV);
}
// the same with the request for the next unit issued in front of the MFMAs (a one-tile unit, or -- wmma1x_pf2 -- a two-tile unit);
// with two waves per SIMD the exact interleaving matters less than at one: the loads are pinned in front, the partner wave computes
template <bool SWAP = false>
__device__ __forceinline__ void wmma1x_pf(f32x4& acc, const WUnit1X& u, const xhalf* bp, int plane_stride, WUnit1X& un, const WNext1X& n,
                                          int lane) {  // (`un` must not be `u`)
    wload1x(un, n, lane);
    wmma1x<SWAP>(acc, u, bp, plane_stride);
}
template <bool SWAP = false>
__device__ __forceinline__ void wmma1x_pf2(f32x4& acc, const WUnit1X& u, const xhalf* bp, int plane_stride, WUnitX& un, const WNextX& n,
                                           int lane) {
    wloadx(un, n, lane);
    wmma1x<SWAP>(acc, u, bp, plane_stride);
}
// The same unit applied to THREE row tiles at once (tile t at bp + t * tile_stride), software-pipelined by hand: the operand reads run one
// k chunk (9 MFMAs) ahead of their use, the MFMAs of a chunk go round the three tiles, and the requests for the NEXT unit's fragments
// are spread under the MFMAs -- one every three (one-tile unit) or two (two-tile unit) MFMAs -- instead of standing in front of the
// phase.  Measured reason (profiles/r05_stage_profile_polyline_fused.txt): with eight waves per workgroup each issuing its eight
// 1-KiB requests at the top of a phase, the CU's one vector-memory path (44 B/clk) takes ~1.5 k cycles to accept them all and every
// wave sits in its own request burst before its first operand read; the favoured wave of a SIMD needed 2.65 k cycles for 36 MFMAs.
// Per accumulator the MFMAs are those of wmma1x in the same order: same bits.
__device__ __forceinline__ void wnext_issue(WUnit1X& un, const WNext1X& n, int lane) {
    const xh8* pa = wfrag1x(n, lane);
    un.b = ldg4(n.bias ? n.bias + n.tile * 16 + (lane >> 4) * 4 : reinterpret_cast<const float*>(n.wpk));
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) un.w[c][p] = pa[(c * NPL + p) * 64];
}
__device__ __forceinline__ void wnext_issue(WUnitX& un, const WNextX& n, int lane) {
    const xh8* pa = wfragx(n, n.tile_a, lane);
    const xh8* pb = wfragx(n, n.tile_b, lane);
    const int bo = (lane >> 4) * 4;
    un.b[0] = ldg4(n.bias ? n.bias + n.tile_a * 16 + bo : reinterpret_cast<const float*>(n.wpk));
    un.b[1] = ldg4(n.bias ? n.bias + n.tile_b * 16 + bo : reinterpret_cast<const float*>(n.wpk));
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            un.w[0][c][p] = pa[(c * NPL + p) * 64];
            un.w[1][c][p] = pb[(c * NPL + p) * 64];
        }
}
__device__ __forceinline__ void wnext_fix_bias(WUnit1X& un, const WNext1X& n) {
    if (!n.bias) un.b = splat(0.f);
}
__device__ __forceinline__ void wnext_fix_bias(WUnitX& un, const WNextX& n) {
    if (!n.bias) {
        un.b[0] = splat(0.f);
        un.b[1] = splat(0.f);
    }
}
template <class U>
struct wnext_loads;  // VMEM instructions of a unit request
template <>
struct wnext_loads<WUnit1X> { static constexpr int n = 1 + 4 * NPL; };
template <>
struct wnext_loads<WUnitX> { static constexpr int n = 2 + 8 * NPL; };

template <bool SWAP = false, class UN, class WN>
__device__ __forceinline__ void wmma1x_3(f32x4 (&acc)[3], const WUnit1X& u, const xhalf* bp, int tile_stride, int plane_stride, UN& un,
                                         const WN& next, int lane) {
    TB_SCHED_FENCE();
    xh8 x[4][3][NPL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int p = 0; p < NPL; ++p) x[c][t][p] = ldsb8(bp + t * tile_stride + p * plane_stride + c * 32);
    wnext_issue(un, next, lane);
    f32x4 mid[3] = {splat(0.f), splat(0.f), splat(0.f)};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (NPL == 2) {
#pragma unroll
            for (int t = 0; t < 3; ++t) mid[t] = mm_sw<SWAP>(u.w[c][0], x[c][t][P1], mid[t]);
#pragma unroll
            for (int t = 0; t < 3; ++t) mid[t] = mm_sw<SWAP>(u.w[c][P1], x[c][t][0], mid[t]);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = mm_sw<SWAP>(u.w[c][0], x[c][t][0], acc[t]);
    }
    // pin: the reads of chunks 0 and 1; then per chunk the reads of chunk c + 2 (two chunks of operands in flight or in use: a third
    // spills) and its MFMAs in groups with one request of the next unit behind each group
    static_assert(NPL == 2, "the pinned schedule below is written out for fp16 pairs (6 reads, 9 MFMAs per chunk)");
    constexpr bool TWO = wnext_loads<UN>::n > 12;    // a two-tile unit is requested (18 VMEM instructions; a one-tile unit: 9)
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        if (TWO) {  // 2 MFMA + 1 request, four times, then 1 MFMA + 1 request: 15 requests under chunks 0 .. 2
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        } else {    // 3 MFMA + 1 request, three times: 9 requests under chunks 0 .. 2
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
    }
    if (TWO) {      // the last three requests under chunk 3
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
    }
    TB_SCHED_FENCE();
    wnext_fix_bias(un, next);
    if (NPL == 2) {
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] += mid[t] * splat(SPLIT_INV);
    }
}
#endif  // !TB_XDL_W3
#endif  // TB_XDL_W3

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
// attention on the XDL pipe.  K and V of a group live in global memory as fp16 pairs, FRAGMENT-MAJOR: for every block of
// 32 keys, head and plane the two operand fragments a wave loads are 2 x 1 KiB contiguous (one fully coalesced
// global_load_dwordx4 each; a row-major layout makes every load touch 16 half-used cache lines and runs at ~1/3 the rate):
//   Kf : [key block][head][plane][key tile t][lane = kq*16 + row][8]   key = blk*32 + 16 t + row,
//                                                                       feature = head*32 + (e < 4 ? 4 kq + e : 16 + 4 kq + e - 4)
//   Vf : [key block][head][plane][d tile dt][lane = kq*16 + row][8]    d = head*32 + 16 dt + row,
//                                                                       key = blk*32 + (e < 4 ? 4 kq + e : 16 + 4 kq + e - 4)
// With that element order the Q^T accumulators of the projection (lane (kq, m): features tt*16 + 4 kq + r) and the P^T values
// of the softmax (lane (kq, m): keys t*16 + 4 kq + r) ARE the B operands of v_mfma_f32_16x16x32_f16 -- no data movement, as
// in the fp32 path.  Per 32 keys: 6 MFMAs for S^T = K Q^T (2 key tiles x 3 products) and 6 for O^T += V^T P^T.
// ---------------------------------------------------------------------------------------------
struct KFragX {
    xh8 ka[2][NPL];  // [key tile][plane]
    f32x4 kb[2];     // additive key bias of this lane's 4 keys per tile
};
struct VFragX {
    xh8 va[2][NPL];  // [d tile][plane]
};

constexpr int KV_BLOCK_HALFS = 4 * NPL * 2 * 512;  // fp16 per 32-key block (all heads): 8192

// kfb / vfb : this lane's base = Kf / Vf + head*NPL*1024 + lane*8 ; k0 = first key of the block (multiple of 32)
__device__ __forceinline__ void k_load_x(KFragX& f, const xhalf* __restrict__ kfb, const float* __restrict__ bbase, int k0) {
#if defined(TB_FAKE_KV) || defined(TB_FAKE_K)  // timing experiment only: every block re-reads the first 32 keys -> results are wrong
    k0 = 0;
#endif
    const xhalf* p = kfb + (size_t)(k0 >> 5) * KV_BLOCK_HALFS;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.ka[t][pl] = *reinterpret_cast<const xh8*>(p + (pl * 2 + t) * 512);
        f.kb[t] = ldg4(bbase + k0 + 16 * t);
    }
}
__device__ __forceinline__ void v_load_x(VFragX& f, const xhalf* __restrict__ vfb, int k0) {
#if defined(TB_FAKE_KV) || defined(TB_FAKE_V)
    k0 = 0;
#endif
    const xhalf* p = vfb + (size_t)(k0 >> 5) * KV_BLOCK_HALFS;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.va[dt][pl] = *reinterpret_cast<const xh8*>(p + (pl * 2 + dt) * 512);
}

// eight floats of a lane -> fp16 pair (B operand)
template <bool CHECK = true, class R = RangeFlag>
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, xh8& h, xh8& l, R&& amax = R{}) {
    xh4 h0, l0, h1, l1;
    split2<CHECK>(a, h0, l0, amax);
    split2<CHECK>(b, h1, l1, amax);
    h = xh8{h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    l = xh8{l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
}

// S^T of 32 keys: s = hi products, c = cross products (to be scaled by 2^-11)
__device__ __forceinline__ void attn_qk_x(const KFragX& f, const xh8& qh, const xh8& ql, f32x4 (&s)[2], f32x4 (&c)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        c[t] = splat(0.f);
        if (NPL == 2) c[t] = mfma_h(f.ka[t][0], ql, c[t]);
        s[t] = mfma_h(f.ka[t][0], qh, splat(0.f));
        if (NPL == 2) c[t] = mfma_h(f.ka[t][P1], qh, c[t]);
    }
}

// 2^x for x <= 0 (v_exp_f32: -inf and everything below -126 give an exact 0).  The running max of the online softmax starts at
// RUN_MAX_NONE, a FINITE value below every logit, so that "nothing valid yet" never produces -inf - -inf = NaN: a masked key is
// -inf - finite = -inf -> 0, and the correction factor of a row whose first valid key arrives is 2^(-3e38 - m) = 0 (times an
// accumulator that is still 0).  Same bits as the earlier form, which clamped every argument at -160 instead (one v_max_f32 per key).
constexpr float RUN_MAX_NONE = -3.0e38f;
// (-DTB_DBG_EXPF, diagnosis builds only -- VERDICT r04 task 4 (d): the softmax in natural units with the library's expf, the form the
// reference's softmax has, instead of log2 units + v_exp_f32; tools/gpu_bias_probe.sh)
#ifdef TB_DBG_EXPF
__device__ __forceinline__ float exp2_neg(float x) { return expf(x); }
constexpr float SOFTMAX_UNIT = 1.0f;
#else
__device__ __forceinline__ float exp2_neg(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr float SOFTMAX_UNIT = 1.44269504088896340736f;
#endif

template <bool SELFMASK>
__device__ __forceinline__ void attn_stats_x(const f32x4 (&s)[2], const f32x4 (&c)[2], const f32x4 (&kb)[2], int kb0, int self_key,
                                             float run_max, float (&sv)[8], float& new_max, float& alpha) {
    const float raw[8] = {s[0].x, s[0].y, s[0].z, s[0].w, s[1].x, s[1].y, s[1].z, s[1].w};
    const float crs[8] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w};
    const float bias[8] = {kb[0].x, kb[0].y, kb[0].z, kb[0].w, kb[1].x, kb[1].y, kb[1].z, kb[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        // logits in log2 units: exp(x - max) = 2^(x log2e - max log2e) is then ONE sub + v_exp_f32 per key instead of the
        // seven instructions of the compensated exp_neg.  The exponent carries the rounding of the (logit x constant) product,
        // ~2^-24 |x| <= 1e-6 relative in p -- the same order as the rounding of logit x scale in the reference's own softmax;
        // closed-loop parity is unchanged within its noise (headline golden: 1.6e-4 vs fp32, 8.8e-5 vs fp64).
        // (crs * 2^-11 is exact, so the explicit fma has the bits of multiply-then-add; -ffp-contract=off would not form it)
        const float v = fmaf(NPL == 2 ? fmaf(crs[r], SPLIT_INV, raw[r]) : raw[r], ATTN_SCALE * SOFTMAX_UNIT, bias[r]);
        sv[r] = (SELFMASK && kb0 + 16 * (r >> 2) + (r & 3) == self_key) ? -INFINITY : v;  // eye mask of MultiAgentTF only
    }
    float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    tmax = rows_max(tmax);
    new_max = fmaxf(run_max, tmax);
    alpha = exp2_neg(run_max - new_max);
}

struct AttnPreX {
    KFragX k0f, kn;
    VFragX vc;
};

__device__ __forceinline__ int kwrap(int k, int n_key_pad) { return k >= n_key_pad ? k - n_key_pad : k; }

// kstart: first key block of this workgroup's walk over the keys (a multiple of 32).  The row tiles of one scene start at
// different blocks and wrap around: they all need the same K / V, so each one misses L2 only on the blocks it reaches first
// and finds the others already fetched by its siblings (the softmax is order independent up to rounding).
__device__ __forceinline__ void attention_prefetch_x(AttnPreX& a, const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                     const float* __restrict__ keybias, int n_key_pad, int kstart, int head, int lane) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    TB_SCHED_FENCE();
    k_load_x(a.k0f, kbase, bbase, kstart);
    v_load_x(a.vc, vbase, kstart);
    k_load_x(a.kn, kbase, bbase, n_key_pad > 32 ? kwrap(kstart + 32, n_key_pad) : kstart);
    TB_SCHED_FENCE();
}

// One head over n_key_pad keys with online softmax; q = this wave's Q^T accumulators.  Returns o (normalised) and whether
// the row had no valid key.  Same recurrences as attention_head (tb_device.hpp); the two matrix products are fp16-pair.
template <bool SELFMASK, bool ISSUE = true, class R = RangeFlag>
__device__ __forceinline__ bool attention_head_x(const f32x4 (&q)[2], AttnPreX& pre, const xhalf* __restrict__ Kh,
                                                 const xhalf* __restrict__ Vh, const float* __restrict__ keybias, int n_key_pad,
                                                 int kstart, int head, int lane, int self_key, f32x4 (&o)[2], WUnitX& un,
                                                 const WNextX& nx, long long* prof = nullptr, R&& amax = R{}) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    xh8 qh, ql;
    split8(q[0], q[1], qh, ql, amax);
    f32x4 oh[2] = {splat(0.f), splat(0.f)}, oc[2] = {splat(0.f), splat(0.f)};
    KFragX kn = pre.kn;
    VFragX vc = pre.vc;
    float run_max = RUN_MAX_NONE, run_sum = 0.f, new_max, alpha, sv[8];
    {
        f32x4 s[2], c[2];
        attn_qk_x(pre.k0f, qh, ql, s, c);
        attn_stats_x<SELFMASK>(s, c, pre.k0f.kb, kstart + kq * 4, self_key, run_max, sv, new_max, alpha);
    }
    // The next weight unit (the out-projection) is requested from inside the loop, two blocks before its end: loads
    // return in issue order, so a 16 KB weight request in front of the K / V stream would stall every block behind it.
    const int nblk = n_key_pad >> 5;
    const int i_issue = nblk >= 2 ? nblk - 2 : 0;
    int kc = kstart, k1 = kwrap(kstart + 32, n_key_pad), k2 = kwrap(k1 + 32, n_key_pad);
    for (int i = 0; i < nblk; ++i) {
        const int kn1 = (i + 1 < nblk) ? k1 : kc;  // clamped re-reads on the tail are harmless
        const int kld = (i + 2 < nblk) ? k2 : kc;
        TB_SCHED_FENCE();
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && i < 2) prof[25 + i * 2] = clock64();
#endif
        // QK of the next block (XDL) under the exponentials of this one; its K fragments are consumed here, so the request for
        // the block after it lands in the SAME registers (one K buffer, one V buffer: the loop stays inside the VGPR file)
        f32x4 ts[2], tc[2];
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_qk_x(kn, qh, ql, ts, tc);
        in_vgpr(ts[0]); in_vgpr(ts[1]); in_vgpr(tc[0]); in_vgpr(tc[1]);
        const f32x4 nb[2] = {kn.kb[0], kn.kb[1]};
        TB_SCHED_FENCE();
        k_load_x(kn, kbase, bbase, kld);
        if (ISSUE && i == i_issue) wloadx(un, nx, lane);
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
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && i < 2) prof[26 + i * 2] = clock64();
#endif
        // PV of this block (XDL) under the scale / mask / running max of the next; then its V registers take the next block
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            if (NPL == 2) oc[dt] = mfma_h(vc.va[dt][0], pl, oc[dt]);
            oh[dt] = mfma_h(vc.va[dt][0], ph, oh[dt]);
            if (NPL == 2) oc[dt] = mfma_h(vc.va[dt][P1], ph, oc[dt]);
        }
        TB_SCHED_FENCE();
        v_load_x(vc, vbase, kn1);
        in_vgpr(oh[0]); in_vgpr(oh[1]); in_vgpr(oc[0]); in_vgpr(oc[1]);
        attn_stats_x<SELFMASK>(ts, tc, nb, kn1 + kq * 4, self_key, run_max, sv, new_max, alpha);  // (unused after the last block)
        TB_SCHED_FENCE();
        kc = k1;
        k1 = k2;
        k2 = kwrap(k2 + 32, n_key_pad);
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && i == 1) prof[29] = clock64();
#endif
    }
    run_sum = rows_sum(run_sum);
    const bool novalid = !(run_sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / run_sum;
    o[0] = (oh[0] + oc[0] * splat(SPLIT_INV)) * splat(inv);
    o[1] = (oh[1] + oc[1] * splat(SPLIT_INV)) * splat(inv);
    return novalid;
}

// ---------------------------------------------------------------------------------------------
// Assist waves (round 5; the TB_XDL_AW build, tb_stepx_bf16aw_kernels.hip): a step workgroup of EIGHT waves.  Waves 0-3 run the
// step as before; wave 4 + w sits on the SIMD of wave w and takes every second key block of head w's walk over the map polylines
// (1024 polylines = 32 blocks per layer: 43 % of a stress-shape launch is this walk at one wave per SIMD, bound by the latency of
// its own dependent chain and of two K / V requests in flight).  Two waves per SIMD interleave their chains and double the
// requests in flight.  The split is STATIC (block sequence index even / odd), so the result does not depend on timing; the two
// un-normalised online-softmax states are merged once per layer through LDS.
//   LDS prefix of the AW build (in front of the step carve): 16 words of control + one partial state per (wave, lane)
// Synchronisation: every __syncthreads() of the main waves is a hardware barrier of all eight waves, and the number of barriers
// in front of a map-attention layer depends on the data (interaction bypass, ...), so the main waves COUNT their barriers (thread 0
// adds 1 to an LDS word in front of each: aw_sync, tb_stepx_kernels.hip) and post a command "at barrier number X: layer l" /
// "exit"; an assist wave loops on s_barrier, counts its own, and acts when the posted number is its count.  (A bare "do it now"
// word cannot work: an assist wave that reads it late cannot tell whether it was written in front of the barrier it just left.)
// ---------------------------------------------------------------------------------------------
#ifdef TB_XDL_AW
constexpr int AW_WORDS = 16;                                   // control words (floats of LDS)
constexpr int AW_PART_LD = 12;                                 // floats per lane of a partial state: o[2] (8), max, sum, pad
constexpr int AW_PREFIX = AW_WORDS + 4 * 64 * AW_PART_LD;      // floats in front of the step kernel's carve
// commands to the assist waves: 0 .. 2 = the odd key blocks of map-attention layer l; 255 = leave.  (Splitting the GEMM chains the same
// way -- W_hh h of a GRU layer / the V halves of the interaction K / V projection on the assist waves -- was built, bit-identical,
// and lost 1 %: those chains run at the rate one CU's L1 fills, which a second wave does not raise; profiles/r05_experiments.txt.)
constexpr unsigned AW_OP_EXIT = 255u;
#else
constexpr int AW_PREFIX = 0;
#endif

struct AttnPartX {
    f32x4 o[2];    // un-normalised O^T accumulators, relative to m
    float m, s;    // running max (log2 units; uniform over the four lanes of a row) and this lane's share of the running sum
};

// ---------------------------------------------------------------------------------------------
// The lean key walk (round 5; first built for the bf16 builds, where it is always used; -DTB_LEAN_FP16 = the fp16-pair builds too): the online softmax with a LAZY reference exponent.  With two or three waves per SIMD
// (assist waves, the W3 carve) the walk is bound by VALU issue -- ~75 vector instructions per 32-key block, of which ~30 serve the
// running maximum: two cross-lane reductions with their wait states, the correction factor, the rescale of eight accumulators and
// of the running sum -- although the maximum moves in a handful of blocks only.  Here a row keeps a REFERENCE exponent `ref`
// (uniform over its four lanes) that is moved only when a block holds a logit above ref + 8 (wave-uniform branch on a ballot):
// p = 2^(logit - ref) may then reach 2^8, which bf16 / fp32 hold as well as values below 1, and the normalisation by the running sum
// takes it out again.  The valid keys of a group sit in front of the masked ones (the hoist compacts them), so a walk is a run of
// FULL blocks (no key bias: no loads for it, the scale folds into the exponent, p = 2^(fma(raw, c, -ref))) plus at most one block
// with masked keys, which is taken LAST: the loop over the full blocks is straight-line code (~36 vector instructions per block)
// apart from the reference update, the last two or three blocks go through a general step.  Mathematically the softmax of
// attention_head_x; other roundings and another summation order (no bit-parity claim exists for bf16 operands).
//   A wave's blocks (LeanSeq): nF full blocks at kwrap(k0 + 32 STEP j, R), then the partial block at kP (if kP >= 0).
//   STEP = 1: every block of the group;  2: every other one (assist waves: the main wave takes the even full blocks, the assist
//   wave the odd ones and the partial block).
// ---------------------------------------------------------------------------------------------
constexpr float LEAN_THR = 8.0f;
constexpr float LEAN_SC = ATTN_SCALE * 1.44269504088896340736f;

struct LeanSeq {
    int R;    // keys of the group's full blocks (a multiple of 32)
    int k0;   // first key of this wave's first full block
    int nF;   // number of full blocks this wave takes
    int kP;   // first key of the block with masked keys if this wave takes it, else -1
};
// n_key_pad = the group's keys rounded up to whole blocks (at least one block), n_valid = its valid keys; kstart = where the walk
// of this workgroup starts (any multiple of 32 below n_key_pad; row tiles of an instance start apart).  part: 0 = the whole walk,
// 1 = the main wave's half, 2 = the assist wave's half.
__device__ __forceinline__ LeanSeq lean_seq(int n_key_pad, int n_valid, int kstart, int part) {
    const int nb = n_key_pad >> 5;
    const int nFt = min(n_valid >> 5, nb);
    LeanSeq q;
    q.R = nFt << 5;
    const int ks = kstart < q.R ? kstart : 0;
    const bool hasP = nFt < nb;
    if (part == 0) {
        q.k0 = ks; q.nF = nFt; q.kP = hasP ? q.R : -1;
    } else if (part == 1) {
        q.k0 = ks; q.nF = (nFt + 1) >> 1; q.kP = -1;
    } else {
        q.k0 = kwrap(ks + 32, max(q.R, 32)); q.nF = nFt >> 1; q.kP = hasP ? q.R : -1;
    }
    return q;
}
template <int STEP>
__device__ __forceinline__ int lean_addr(const LeanSeq& q, int j) { return j < q.nF ? kwrap(q.k0 + 32 * STEP * j, q.R) : q.kP; }
__device__ __forceinline__ int lean_count(const LeanSeq& q) { return q.nF + (q.kP >= 0 ? 1 : 0); }

// K fragments of a block / its key bias (read only when the block has masked keys)
__device__ __forceinline__ void k_load_lean_x(KFragX& f, const xhalf* __restrict__ kfb, int k0) {
    const xhalf* p = kfb + (size_t)(k0 >> 5) * KV_BLOCK_HALFS;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) f.ka[t][pl] = *reinterpret_cast<const xh8*>(p + (pl * 2 + t) * 512);
}
__device__ __forceinline__ void kb_load_lean_x(KFragX& f, const float* __restrict__ bbase, int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) f.kb[t] = ldg4(bbase + k0 + 16 * t);
}

__device__ __forceinline__ float max8_x(const float (&r)[8]) {
    return fmaxf(fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3])), fmaxf(fmaxf(r[4], r[5]), fmaxf(r[6], r[7])));
}

// State of a walk.  The logits of a block are kept RELATIVE to the row's reference exponent (v = logit - ref, log2 units): the
// exponentials take them as they are, "a logit more than 2^8 above the reference" is a compare with a constant, and the maximum is
// taken over fma results (on raw MFMA results the compiler would first canonicalise each input: one v_max x, x apiece).
struct LeanState {
    f32x4 oh[2], oc[2];   // O^T accumulators: high products / cross products of the fp16 pairs (oc unused with one plane)
    float nref;      // - reference exponent (uniform over the four lanes of a row)
    float run_sum;   // this lane's share of the sum of p
};

// the reference moves by d >= 0 where a block's largest relative logit `lm` exceeds 8
__device__ __forceinline__ void lean_ref_x(float lm, float (&v)[8], LeanState& st) {
    if (__builtin_amdgcn_ballot_w64(lm > LEAN_THR) != 0ull) {  // (wave-uniform; rows that do not move get d = 0: factors of exactly 1)
        const float d = fmaxf(rows_max(lm), 0.f);
        const float alpha = exp2_neg(-d);
        st.oh[0] *= splat(alpha);
        st.oh[1] *= splat(alpha);
        if (NPL == 2) {
            st.oc[0] *= splat(alpha);
            st.oc[1] *= splat(alpha);
        }
        st.run_sum *= alpha;
        st.nref -= d;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] -= d;
    }
}
// logits of a FULL block (no key bias), relative to the reference
// raw scores of a block: high products + 2^-11 cross products (fp16 pairs)
__device__ __forceinline__ void lean_raw_x(const f32x4 (&s)[2], const f32x4 (&c)[2], float (&raw)[8]) {
    const float a[8] = {s[0].x, s[0].y, s[0].z, s[0].w, s[1].x, s[1].y, s[1].z, s[1].w};
    const float b[8] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = NPL == 2 ? fmaf(b[r], SPLIT_INV, a[r]) : a[r];
}
__device__ __forceinline__ void lean_stats_full_x(const f32x4 (&s)[2], const f32x4 (&c)[2], float (&v)[8], LeanState& st) {
    float raw[8];
    lean_raw_x(s, c, raw);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fmaf(raw[r], LEAN_SC, st.nref);
    lean_ref_x(max8_x(v), v, st);
}
// logits of a block with masked keys
__device__ __forceinline__ void lean_stats_part_x(const f32x4 (&s)[2], const f32x4 (&c)[2], const f32x4 (&kb)[2], float (&v)[8], LeanState& st) {
    float raw[8];
    lean_raw_x(s, c, raw);
    const float bias[8] = {kb[0].x, kb[0].y, kb[0].z, kb[0].w, kb[1].x, kb[1].y, kb[1].z, kb[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fmaf(raw[r], LEAN_SC, bias[r]) + st.nref;
    lean_ref_x(max8_x(v), v, st);
}
// the first block of a walk sets the reference: its row maximum (a finite stand-in when every key of the block is masked: then
// every p of the block is 2^-inf = 0)
__device__ __forceinline__ void lean_stats_first_x(const f32x4 (&s)[2], const f32x4 (&c)[2], const f32x4 (&kb)[2], bool part, float (&v)[8],
                                                   LeanState& st) {
    float raw[8];
    lean_raw_x(s, c, raw);
    if (part) {
        const float bias[8] = {kb[0].x, kb[0].y, kb[0].z, kb[0].w, kb[1].x, kb[1].y, kb[1].z, kb[1].w};
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaf(raw[r], LEAN_SC, bias[r]);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = raw[r] * LEAN_SC;
    }
    const float ref = fmaxf(rows_max(max8_x(v)), RUN_MAX_NONE);
    st.nref = -ref;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] -= ref;
}
// exponentials of the current block -> running sum, P^T as a B operand
__device__ __forceinline__ void lean_exp_x(const float (&v)[8], LeanState& st, xh8& ph, xh8& pl) {
    float p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = exp2_neg(v[r]);
    st.run_sum += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    split8<false>(f32x4{p[0], p[1], p[2], p[3]}, f32x4{p[4], p[5], p[6], p[7]}, ph, pl);  // (p <= 2^8: inside the fp16 range)
}
// O^T += V^T P^T of the current block
__device__ __forceinline__ void lean_pv_x(const VFragX& vc, const xh8& ph, const xh8& pl, LeanState& st) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        if (NPL == 2) st.oc[dt] = mfma_h(vc.va[dt][0], pl, st.oc[dt]);
        st.oh[dt] = mfma_h(vc.va[dt][0], ph, st.oh[dt]);
        if (NPL == 2) st.oc[dt] = mfma_h(vc.va[dt][P1], ph, st.oc[dt]);
    }
}

// One step in the run of full blocks: exponentials and P V of block j (full; `vcur` from the previous step), Q K and statistics of
// block j + 1 (full) -> `vnext`, K of block j + 2 (full) and V of block j + 1 requested.  Straight-line code.
__device__ __forceinline__ void lean_step_full_x(const xh8& qh, const xh8& ql, const xhalf* kbase, const xhalf* vbase, int k_next, int k_nn,
                                                 const float (&vcur)[8], float (&vnext)[8], KFragX& kn, VFragX& vc, LeanState& st) {
    TB_SCHED_FENCE();
    f32x4 ts[2], tc[2];
    in_vgpr(st.oh[0]); in_vgpr(st.oh[1]);
    if (NPL == 2) { in_vgpr(st.oc[0]); in_vgpr(st.oc[1]); }
    attn_qk_x(kn, qh, ql, ts, tc);  // QK of the next block (XDL) under the exponentials of this one
    in_vgpr(ts[0]); in_vgpr(ts[1]);
    if (NPL == 2) { in_vgpr(tc[0]); in_vgpr(tc[1]); }
    TB_SCHED_FENCE();
    k_load_lean_x(kn, kbase, k_nn);
    xh8 ph, pl;
    lean_exp_x(vcur, st, ph, pl);
    TB_SCHED_FENCE();
    lean_pv_x(vc, ph, pl, st);
    TB_SCHED_FENCE();
    v_load_x(vc, vbase, k_next);
    in_vgpr(st.oh[0]); in_vgpr(st.oh[1]);
    if (NPL == 2) { in_vgpr(st.oc[0]); in_vgpr(st.oc[1]); }
    lean_stats_full_x(ts, tc, vnext, st);
    TB_SCHED_FENCE();
}

// The last steps of a walk (at most three): any of "block j has masked keys", "there is a block j + 1 / j + 2", "it has masked keys".
// `v` holds block j's logits on entry and block j + 1's on exit.
template <int STEP>
__device__ __forceinline__ void lean_step_any_x(int j, const LeanSeq& q, const xh8& qh, const xh8& ql, const xhalf* kbase, const xhalf* vbase,
                                                const float* bbase, float (&v)[8], KFragX& kn, VFragX& vc, LeanState& st) {
    const int n_tot = lean_count(q);
    const bool has_next = j + 1 < n_tot, next_part = j + 1 >= q.nF, has_nn = j + 2 < n_tot, nn_part = j + 2 >= q.nF;
    f32x4 ts[2] = {splat(0.f), splat(0.f)}, tc[2] = {splat(0.f), splat(0.f)};
    TB_SCHED_FENCE();
    if (has_next) attn_qk_x(kn, qh, ql, ts, tc);
    const f32x4 nb[2] = {kn.kb[0], kn.kb[1]};
    TB_SCHED_FENCE();
    if (has_nn) {
        k_load_lean_x(kn, kbase, lean_addr<STEP>(q, j + 2));
        if (nn_part) kb_load_lean_x(kn, bbase, lean_addr<STEP>(q, j + 2));
    }
    TB_SCHED_FENCE();
    xh8 ph, pl;
    lean_exp_x(v, st, ph, pl);
    lean_pv_x(vc, ph, pl, st);
    TB_SCHED_FENCE();
    if (has_next) {
        v_load_x(vc, vbase, lean_addr<STEP>(q, j + 1));
        if (next_part) lean_stats_part_x(ts, tc, nb, v, st);
        else lean_stats_full_x(ts, tc, v, st);
    }
    TB_SCHED_FENCE();
}

// prefetch for attention_walk_lean_x: K (+ bias) and V of the wave's first block, K (+ bias) of its second one.  PART 1 = the first
// block's K only, 2 = the rest, 3 = everything (with fp16 pairs the first K goes in front of the LayerNorm, xattn_layer_x)
template <int STEP, int PART = 3>
__device__ __forceinline__ void attention_prefetch_lean_x(AttnPreX& a, const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                          const float* __restrict__ keybias, const LeanSeq& q, int head, int lane) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    const int n_tot = lean_count(q);
    TB_SCHED_FENCE();
    if (n_tot > 0) {
        const int a0 = lean_addr<STEP>(q, 0);
        if (PART & 1) {
            k_load_lean_x(a.k0f, kbase, a0);
            if (q.nF == 0) kb_load_lean_x(a.k0f, bbase, a0);
        }
        if (PART & 2) {
            v_load_x(a.vc, vbase, a0);
            if (n_tot > 1) {
                const int a1 = lean_addr<STEP>(q, 1);
                k_load_lean_x(a.kn, kbase, a1);
                if (q.nF <= 1) kb_load_lean_x(a.kn, bbase, a1);
            }
        }
    }
    TB_SCHED_FENCE();
}

template <int STEP>
__device__ __forceinline__ void attention_walk_lean_x(const xh8& qh, const xh8& ql, AttnPreX& pre, const xhalf* __restrict__ Kh,
                                                      const xhalf* __restrict__ Vh, const float* __restrict__ keybias, const LeanSeq& q, int head,
                                                      int lane, AttnPartX& out) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    LeanState st;
    st.oh[0] = splat(0.f); st.oh[1] = splat(0.f);
    st.oc[0] = splat(0.f); st.oc[1] = splat(0.f);
    st.nref = -RUN_MAX_NONE; st.run_sum = 0.f;
    const int n_tot = lean_count(q);
    if (n_tot > 0) {
        KFragX kn = pre.kn;
        VFragX vc = pre.vc;
        float va[8], vb[8];
        {
            f32x4 s[2], c[2];
            attn_qk_x(pre.k0f, qh, ql, s, c);
            lean_stats_first_x(s, c, pre.k0f.kb, q.nF == 0, va, st);
        }
        // steps whose blocks j, j + 1, j + 2 are all full, two per iteration with the logit arrays swapped (the MFMA results of one
        // block are consumed where they are: no copies across the back edge)
        int j = 0;
        for (; j + 3 < q.nF; j += 2) {
            const int k1 = lean_addr<STEP>(q, j + 1), k2 = lean_addr<STEP>(q, j + 2), k3 = lean_addr<STEP>(q, j + 3);
            lean_step_full_x(qh, ql, kbase, vbase, k1, k2, va, vb, kn, vc, st);
            lean_step_full_x(qh, ql, kbase, vbase, k2, k3, vb, va, kn, vc, st);
        }
        for (; j < n_tot; ++j) lean_step_any_x<STEP>(j, q, qh, ql, kbase, vbase, bbase, va, kn, vc, st);
    }
    out.o[0] = NPL == 2 ? st.oh[0] + st.oc[0] * splat(SPLIT_INV) : st.oh[0];
    out.o[1] = NPL == 2 ? st.oh[1] + st.oc[1] * splat(SPLIT_INV) : st.oh[1];
    out.m = -st.nref;
    out.s = st.run_sum;
}

// The lean walk for key groups that are NOT compacted (the interaction: key slot = agent index, any agent may be invalid, and the
// eye mask of MultiAgentTF hides a row's own key): every block reads its key bias, the lazy reference is the same.  Blocks
// kwrap(kstart + 32 j, n_key_pad), j = 0 .. n_key_pad / 32 - 1; pre = attention_prefetch_x's (K + bias and V of block 0, K + bias of block 1).
template <bool SELFMASK>
__device__ __forceinline__ void lean_stats_m_x(const f32x4 (&s)[2], const f32x4 (&c)[2], const f32x4 (&kb)[2], int kb0, int self_key, bool first,
                                               float (&v)[8], LeanState& st) {
    float raw[8];
    lean_raw_x(s, c, raw);
    const float bias[8] = {kb[0].x, kb[0].y, kb[0].z, kb[0].w, kb[1].x, kb[1].y, kb[1].z, kb[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float l = fmaf(raw[r], LEAN_SC, bias[r]);
        v[r] = (SELFMASK && kb0 + 16 * (r >> 2) + (r & 3) == self_key) ? -INFINITY : l;
    }
    if (first) {
        const float ref = fmaxf(rows_max(max8_x(v)), RUN_MAX_NONE);
        st.nref = -ref;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] -= ref;
    } else {
        // A row that has met no valid key so far (every key of its earlier blocks masked or its own) still carries the stand-in
        // reference: adding +3e38 to real logits would round them away (every valid key of this block would come out with p = 1:
        // uniform weights -- ADVICE r05 high).  Such a row takes its reference from THIS block's unshifted logits, as a first block
        // does; its accumulators and sum are still zero, so nothing is rescaled.  Wave-uniform, rare branch.
        const bool noref = st.nref == -RUN_MAX_NONE;
        if (__builtin_amdgcn_ballot_w64(noref) != 0ull) {
            const float ref = fmaxf(rows_max(max8_x(v)), RUN_MAX_NONE);
            if (noref) st.nref = -ref;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += st.nref;
        lean_ref_x(max8_x(v), v, st);
    }
}
template <bool SELFMASK>
__device__ __forceinline__ void attention_walk_leanm_x(const xh8& qh, const xh8& ql, AttnPreX& pre, const xhalf* __restrict__ Kh,
                                                       const xhalf* __restrict__ Vh, const float* __restrict__ keybias, int n_key_pad, int kstart,
                                                       int head, int lane, int self_key, AttnPartX& out) {
    const int kq = lane >> 4;
    const xhalf* kbase = Kh + head * (NPL * 1024) + lane * 8;
    const xhalf* vbase = Vh + head * (NPL * 1024) + lane * 8;
    const float* bbase = keybias + kq * 4;
    LeanState st;
    st.oh[0] = splat(0.f); st.oh[1] = splat(0.f);
    st.oc[0] = splat(0.f); st.oc[1] = splat(0.f);
    st.nref = -RUN_MAX_NONE; st.run_sum = 0.f;
    KFragX kn = pre.kn;
    VFragX vc = pre.vc;
    float v[8];
    {
        f32x4 s[2], c[2];
        attn_qk_x(pre.k0f, qh, ql, s, c);
        lean_stats_m_x<SELFMASK>(s, c, pre.k0f.kb, kstart + kq * 4, self_key, true, v, st);
    }
    const int nblk = n_key_pad >> 5;
    int k1 = kwrap(kstart + 32, n_key_pad), k2 = kwrap(k1 + 32, n_key_pad);
    for (int i = 0; i < nblk; ++i) {
        const bool has_next = i + 1 < nblk, has_nn = i + 2 < nblk;
        f32x4 ts[2] = {splat(0.f), splat(0.f)}, tc[2] = {splat(0.f), splat(0.f)};
        TB_SCHED_FENCE();
        if (has_next) attn_qk_x(kn, qh, ql, ts, tc);  // QK of the next block under the exponentials of this one
        const f32x4 nb[2] = {kn.kb[0], kn.kb[1]};
        TB_SCHED_FENCE();
        if (has_nn) k_load_x(kn, kbase, bbase, k2);
        xh8 ph, pl;
        lean_exp_x(v, st, ph, pl);
        TB_SCHED_FENCE();
        lean_pv_x(vc, ph, pl, st);
        TB_SCHED_FENCE();
        if (has_next) {
            v_load_x(vc, vbase, k1);
            lean_stats_m_x<SELFMASK>(ts, tc, nb, k1 + kq * 4, self_key, false, v, st);
        }
        TB_SCHED_FENCE();
        k1 = k2;
        k2 = kwrap(k2 + 32, n_key_pad);
    }
    out.o[0] = NPL == 2 ? st.oh[0] + st.oc[0] * splat(SPLIT_INV) : st.oh[0];
    out.o[1] = NPL == 2 ? st.oh[1] + st.oc[1] * splat(SPLIT_INV) : st.oh[1];
    out.m = -st.nref;
    out.s = st.run_sum;
}

// one state -> the normalised attention output
__device__ __forceinline__ bool attention_finish_lean_x(const AttnPartX& a, f32x4 (&o)[2]) {
    const float s = rows_sum(a.s);
    const bool novalid = !(s > 0.f);
    const float inv = novalid ? 0.f : 1.0f / s;
    o[0] = a.o[0] * splat(inv);
    o[1] = a.o[1] * splat(inv);
    return novalid;
}

#ifdef TB_XDL_AW
// two partial states of one row set -> the normalised attention output (merged in the order main, assist)
__device__ __forceinline__ bool attention_merge2_x(const AttnPartX& a, const AttnPartX& b, f32x4 (&o)[2]) {
    const float mx = fmaxf(a.m, b.m);
    const float fa = exp2_neg(a.m - mx), fb = exp2_neg(b.m - mx);
    float s = a.s * fa + b.s * fb;
    s = rows_sum(s);
    const bool novalid = !(s > 0.f);
    const float inv = novalid ? 0.f : 1.0f / s;
    o[0] = (a.o[0] * splat(fa) + b.o[0] * splat(fb)) * splat(inv);
    o[1] = (a.o[1] * splat(fa) + b.o[1] * splat(fb)) * splat(inv);
    return novalid;
}

__device__ __forceinline__ unsigned int* aw_words() {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    return reinterpret_cast<unsigned int*>(smem_all);
}
__device__ __forceinline__ float* aw_part(int wave, int lane) {
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    return smem_all + AW_WORDS + (wave * 64 + lane) * AW_PART_LD;
}

// main waves, thread 0: "at the NEXT barrier: op" (the count of barriers so far is this thread's own word 0)
__device__ __forceinline__ void aw_post(unsigned int op) {
    unsigned int* w = aw_words();
    const unsigned int c = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(w + 2, op, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(w + 1, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif  // TB_XDL_AW

// The same attention when the whole key set is ONE 32-key block (a polyline's 20 nodes, a scene's lit traffic lights): no walk, so no
// second QK for a "next" block, no rescale, no re-loads -- and nothing that pins the schedule, so that two calls for independent
// (query tile, key block) pairs interleave in one instruction stream.  Arithmetic and its order are those of attention_head_x with
// n_key_pad = 32 (the correction factor it multiplies by is an exact 0 x 0): same bits.
template <class R = RangeFlag>
__device__ __forceinline__ bool attention_oneblock_pre_x(const f32x4 (&q)[2], const KFragX& kf, const VFragX& vf, int lane, f32x4 (&o)[2],
                                                         R&& amax = R{});
__device__ __forceinline__ bool attention_oneblock_q_x(const xh8& qh, const xh8& ql, const KFragX& kf, const VFragX& vf, int lane, f32x4 (&o)[2]);
template <class R>
__device__ __forceinline__ bool attention_oneblock_pre_x(const f32x4 (&q)[2], const KFragX& kf, const VFragX& vf, int lane, f32x4 (&o)[2],
                                                         R&& amax) {
    xh8 qh, ql;
    split8(q[0], q[1], qh, ql, amax);
    return attention_oneblock_q_x(qh, ql, kf, vf, lane, o);
}
// (the same with Q already split: a caller that runs several attentions on one Q splits it once)
__device__ __forceinline__ bool attention_oneblock_q_x(const xh8& qh, const xh8& ql, const KFragX& kf, const VFragX& vf, int lane, f32x4 (&o)[2]) {
    const int kq = lane >> 4;
    f32x4 s[2], c[2];
    attn_qk_x(kf, qh, ql, s, c);
    float sv[8], new_max, alpha;
    attn_stats_x<false>(s, c, kf.kb, kq * 4, -1, RUN_MAX_NONE, sv, new_max, alpha);
    float pr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) pr[r] = exp2_neg(sv[r] - new_max);
    float run_sum = 0.f * alpha + (((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7])));
    xh8 ph, pl;
    split8<false>(f32x4{pr[0], pr[1], pr[2], pr[3]}, f32x4{pr[4], pr[5], pr[6], pr[7]}, ph, pl);
    f32x4 oh[2] = {splat(0.f), splat(0.f)}, oc[2] = {splat(0.f), splat(0.f)};
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        if (NPL == 2) oc[dt] = mfma_h(vf.va[dt][0], pl, oc[dt]);
        oh[dt] = mfma_h(vf.va[dt][0], ph, oh[dt]);
        if (NPL == 2) oc[dt] = mfma_h(vf.va[dt][P1], ph, oc[dt]);
    }
    run_sum = rows_sum(run_sum);
    const bool novalid = !(run_sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / run_sum;
    o[0] = (oh[0] + oc[0] * splat(SPLIT_INV)) * splat(inv);
    o[1] = (oh[1] + oc[1] * splat(SPLIT_INV)) * splat(inv);
    return novalid;
}
__device__ __forceinline__ bool attention_oneblock_x(const f32x4 (&q)[2], const xhalf* __restrict__ Kh, const xhalf* __restrict__ Vh,
                                                     const float* __restrict__ keybias, int head, int lane, f32x4 (&o)[2]) {
    KFragX kf;
    VFragX vf;
    k_load_x(kf, Kh + head * (NPL * 1024) + lane * 8, keybias + (lane >> 4) * 4, 0);
    v_load_x(vf, Vh + head * (NPL * 1024) + lane * 8, 0);
    return attention_oneblock_pre_x(q, kf, vf, lane, o);
}

// K / V accumulators of a 16-token tile -> global, fragment-major (see above).  ak / av : this wave's K / V tiles
// (features (2 wave + t)*16 + 4 kq + r of token m); the wave is head `wave`.
// `tok` = the key slot this lane's token goes to (its index, or its rank among the valid tokens when the hoist compacts)
template <class R = RangeFlag>
__device__ __forceinline__ void kv_store_key_x(xhalf* __restrict__ Kf, xhalf* __restrict__ Vf, int tok, int wave, int lane,
                                               const f32x4 (&ak)[2], const f32x4 (&av)[2], bool real, R&& amax = R{}) {
    const int kq = lane >> 4;
    const int j = tok & 31;
    xhalf* kblk = Kf + (size_t)(tok >> 5) * KV_BLOCK_HALFS + wave * (NPL * 1024);
    xhalf* vblk = Vf + (size_t)(tok >> 5) * KV_BLOCK_HALFS + wave * (NPL * 1024);
    const int kt = j >> 4, krow = j & 15;               // key tile / row of this token inside its block
    const int vq = (j >> 2) & 3, ve = (j >> 4) * 4 + (j & 3);  // lane group / element that hold this key in the V fragments
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        xh4 h, l;
        split2(real ? ak[t] : splat(0.f), h, l, amax);
        xhalf* pk = kblk + (kt * 64 + kq * 16 + krow) * 8 + t * 4;  // plane 0, tile kt
        *reinterpret_cast<xh4*>(pk) = h;
        if (NPL == 2) *reinterpret_cast<xh4*>(pk + 1024) = l;
        split2(real ? av[t] : splat(0.f), h, l, amax);
        xhalf* pv = vblk + (t * 64 + vq * 16 + kq * 4) * 8 + ve;    // plane 0, d tile t, rows 4 kq + r
        pv[0] = h.x; pv[8] = h.y; pv[16] = h.z; pv[24] = h.w;
        if (NPL == 2) {
            pv += 1024;
            pv[0] = l.x; pv[8] = l.y; pv[16] = l.z; pv[24] = l.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One pre-LN cross-attention layer, GEMMs and attention on XDL.
//   X : [16][LDT] fp32 residual stream (LDS);  P1, P2 : plane buffers (LN output / attention output + FFN hidden)
// ---------------------------------------------------------------------------------------------
// AW (the TB_XDL_AW build): the partner assist wave of every main wave takes the odd key blocks of this layer (`aw_op` = the layer's
// index for the assist waves, posted for the LayerNorm barrier below)
// COMPACT: the group's valid keys sit in front of the masked ones and `n_valid_keys` is their count (what the hoists of the step
// kernel produce): the attention then takes the lean walk (attention_walk_lean_x) in the bf16 builds (and with -DTB_LEAN_FP16)
template <bool LNLDS = false, bool SELFMASK = false, bool WO_EARLY = true, bool AW = false, bool COMPACT = false, class R = RangeFlag>
__device__ __forceinline__ void xattn_layer_x(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1,
                                              xhalf* P2, const xhalf* __restrict__ Kmat, const xhalf* __restrict__ VT,
                                              const float* __restrict__ keybias, int n_key_pad, int kstart, int self_key0,
                                              const uint8_t* rowvalid, uint8_t* novalid_s, int tid, WUnitX& u, const WNextX& nxt,
                                              const float* lnblk = nullptr, long long* prof = nullptr, R&& amax = R{}, int aw_op = 0,
                                              int n_valid_keys = -1) {
    // n_valid_keys (bf16 builds, attention_walk_lean_x; required there unless SELFMASK): the group's count of valid keys -- they sit in
    // front of the masked ones and n_key_pad is that count rounded up to whole blocks (at least one block)
    if (!LNLDS) lnblk = W + L.ln1_g;
#ifdef TB_XDL_AW
    if (AW && tid == 0) aw_post((unsigned int)aw_op);
#else
    static_assert(!AW, "assist waves: TB_XDL_AW build only");
    (void)aw_op;
#endif
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    TB_XSTAMP(16);
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    AttnPreX apre;
    // The out-projection unit is requested BESIDE the Q projection (wmmax_pf interleaves its 16 requests with the MFMAs, behind the
    // K / V prefetch of the first two key blocks).  Requested from inside the key walk, as at first, the wave sat ~1.5 k cycles in the
    // request's own issue -- the per-wave vector-memory queue is shallow -- in the middle of the attention; requested in front of the
    // LayerNorm it delayed the LayerNorm instead.  fp16 pairs 96.0 -> 92.6 us per fused launch, bf16 68.2 -> 64.6.
    // (WO_EARLY = false keeps the request inside the key walk)
    WUnitX u2;
    // K / V prefetch of the first two key blocks.  With fp16 pairs (16 requests, 16 KB per wave) the first block's K fragments are
    // requested in front of the LayerNorm arithmetic and the rest behind it: the vector-memory queue takes about that many without
    // stalling the wave, so part of the burst runs beside the arithmetic (92.6 -> 92.0 us per fused launch); with one bf16 plane the
    // whole burst is short and stays behind the LayerNorm (in front of it: 64.7 -> 65.2).
#if defined(TB_XDL_BF16) || defined(TB_LEAN_FP16)
    constexpr bool LEANW = COMPACT && !SELFMASK;  // the lean walk for the attention over compacted key groups (map, traffic lights)
    // ... and, with WO_EARLY, the lean walk over un-compacted groups (attention_walk_leanm_x) for the interaction of the step kernel
    constexpr bool LEANM = SELFMASK && WO_EARLY;
#else
    constexpr bool LEANW = false, LEANM = false;
#endif
    if (NPL == 2) {
        if (LEANW) attention_prefetch_lean_x<1, 1>(apre, Kmat, VT, keybias, lean_seq(n_key_pad, n_valid_keys, kstart, 0), wave, lane);
        else {
            TB_SCHED_FENCE();
            k_load_x(apre.k0f, Kmat + wave * (NPL * 1024) + lane * 8, keybias + kq * 4, kstart);
            TB_SCHED_FENCE();
        }
    }
    layernorm_planes<LNLDS>(X, LDT, P1, lnblk, lnblk + 128, tid);
    if (NPL == 2) {
        if (LEANW) attention_prefetch_lean_x<1, 2>(apre, Kmat, VT, keybias, lean_seq(n_key_pad, n_valid_keys, kstart, 0), wave, lane);
        else {
            TB_SCHED_FENCE();
            v_load_x(apre.vc, VT + wave * (NPL * 1024) + lane * 8, kstart);
            k_load_x(apre.kn, Kmat + wave * (NPL * 1024) + lane * 8, keybias + kq * 4, n_key_pad > 32 ? kwrap(kstart + 32, n_key_pad) : kstart);
            TB_SCHED_FENCE();
        }
    } else {
        if (LEANW) attention_prefetch_lean_x<AW ? 2 : 1>(apre, Kmat, VT, keybias, lean_seq(n_key_pad, n_valid_keys, kstart, AW ? 1 : 0), wave, lane);
        else attention_prefetch_x(apre, Kmat, VT, keybias, n_key_pad, kstart, wave, lane);
    }
    __syncthreads();
    TB_XSTAMP(17);
    f32x4 q[2] = {u.b[0], u.b[1]};
    if (WO_EARLY) wmmax_pf(q[0], q[1], u, b1, PLANE, u2, wstdx(W, LX.wo, W + L.bo, wave), lane);
    else wmmax(q[0], q[1], u, b1, PLANE);
    TB_XSTAMP(18);
    f32x4 o[2];
    bool novalid;
    static_assert(!LEANW || WO_EARLY, "the lean walk issues no weight request from inside the walk");
    static_assert(!AW || LEANW, "assist waves split the lean walk");
#ifdef TB_XDL_AW
    if (AW) {
        // this wave: block sequence indices 0, 2, 4, ...; the assist wave on the same SIMD: 1, 3, 5, ... (aw_assist_layer_x)
        xh8 qh, ql;
        split8(q[0], q[1], qh, ql, amax);
        AttnPartX mine, theirs;
        attention_walk_lean_x<2>(qh, ql, apre, Kmat, VT, keybias, lean_seq(n_key_pad, n_valid_keys, kstart, 1), wave, lane, mine);
        TB_XSTAMP(25);
        __syncthreads();  // the assist waves have written their states
        const float* ps = aw_part(wave, lane);
        theirs.o[0] = lds4(ps);
        theirs.o[1] = lds4(ps + 4);
        theirs.m = ps[8];
        theirs.s = ps[9];
        novalid = attention_merge2_x(mine, theirs, o);
    } else
#endif
    if (LEANW) {
        xh8 qh, ql;
        split8(q[0], q[1], qh, ql, amax);
        AttnPartX st;
        attention_walk_lean_x<1>(qh, ql, apre, Kmat, VT, keybias, lean_seq(n_key_pad, n_valid_keys, kstart, 0), wave, lane, st);
        novalid = attention_finish_lean_x(st, o);
    } else if (LEANM) {
        xh8 qh, ql;
        split8(q[0], q[1], qh, ql, amax);
        AttnPartX st;
        attention_walk_leanm_x<SELFMASK>(qh, ql, apre, Kmat, VT, keybias, n_key_pad, kstart, wave, lane, self_key0 >= 0 ? self_key0 + m : -1, st);
        novalid = attention_finish_lean_x(st, o);
    } else
    novalid = attention_head_x<SELFMASK, !WO_EARLY>(q, apre, Kmat, VT, keybias, n_key_pad, kstart, wave, lane,
                                                    self_key0 >= 0 ? self_key0 + m : -1, o, u2, wstdx(W, LX.wo, W + L.bo, wave), prof, amax);
    TB_XSTAMP(19);
    planes_store_c<false>(P2, 2 * wave, lane, o[0]);  // (a convex combination of V, which was checked when it was stored)
    planes_store_c<false>(P2, 2 * wave + 1, lane, o[1]);
    if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
    __syncthreads();
    TB_XSTAMP(20);
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
    TB_XSTAMP(21);
    layernorm_planes<LNLDS>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    TB_XSTAMP(22);
    {
        f32x4 acc[2] = {u.b[0], u.b[1]};
        wmmax_pf(acc[0], acc[1], u, b1, PLANE, u2, wstdx(W, LX.w2, W + L.b2, wave), lane);
        planes_store_c(P2, 2 * wave, lane, relu4(acc[0]), amax);
        planes_store_c(P2, 2 * wave + 1, lane, relu4(acc[1]), amax);
    }
    __syncthreads();
    TB_XSTAMP(23);
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
    TB_XSTAMP(24);
}

#ifdef TB_XDL_AW
// Assist wave `wave` (0..3 = the head / the main wave it shares a SIMD with) for one map-attention layer, entered right behind
// the layer's LayerNorm barrier: the same Q projection from the same planes (the wave's own copy of the weight unit, `u`: loaded
// while it waited), the odd key blocks, its state to LDS.  The caller runs the barrier that publishes the state.
// `apre`: the K / V fragments of the wave's first two blocks, requested while it waited as well (attention_prefetch_lean_x<2>).
__device__ __forceinline__ void aw_assist_layer_x(const WUnitX& u, AttnPreX& apre, const xhalf* P1, const xhalf* __restrict__ Kmat,
                                                  const xhalf* __restrict__ VT, const float* __restrict__ keybias, const LeanSeq& sq, int wave,
                                                  int lane) {
    const int kq = lane >> 4, m = lane & 15;
    f32x4 q[2] = {u.b[0], u.b[1]};
    wmmax(q[0], q[1], u, P1 + m * LDP + kq * 8, PLANE);
    xh8 qh, ql;
    split8<false>(q[0], q[1], qh, ql);
    AttnPartX st;
    attention_walk_lean_x<2>(qh, ql, apre, Kmat, VT, keybias, sq, wave, lane, st);
    float* ps = aw_part(wave, lane);
    st4(ps, st.o[0]);
    st4(ps + 4, st.o[1]);
    ps[8] = st.m;
    ps[9] = st.s;
}
#endif

template <class R = RangeFlag>
__device__ __forceinline__ void kv_store_x(xhalf* __restrict__ Kf, xhalf* __restrict__ Vf, int tok0, int wave, int lane,
                                           const f32x4 (&ak)[2], const f32x4 (&av)[2], bool real, R&& amax = R{}) {
    kv_store_key_x(Kf, Vf, tok0 + (lane & 15), wave, lane, ak, av, real, amax);
}

// The same layer when its target set has no valid key at all (a scene without a lit traffic light): every row is "all keys
// invalid", the reference zeroes such a row's attention output after the out-projection, so x passes the attention
// half unchanged and only x += FFN(LN2(x)) remains.  u = the FFN1 unit on entry.
template <bool LNLDS = false, class R = RangeFlag>
__device__ __forceinline__ void ffn_layer_x(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1, xhalf* P2,
                                            const uint8_t* rowvalid, int tid, WUnitX& u, const WNextX& nxt, const float* lnblk = nullptr,
                                            R&& amax = R{}) {
    if (!LNLDS) lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    WUnitX u2;
    layernorm_planes<LNLDS>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    {
        f32x4 acc[2] = {u.b[0], u.b[1]};
        wmmax_pf(acc[0], acc[1], u, b1, PLANE, u2, wstdx(W, LX.w2, W + L.b2, wave), lane);
        planes_store_c(P2, 2 * wave, lane, relu4(acc[0]), amax);
        planes_store_c(P2, 2 * wave + 1, lane, relu4(acc[1]), amax);
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

// K/V projection of the tile's tokens for one layer (LN_tgt -> in_proj rows 128:384), outputs in XDL operand order.
// slot != nullptr (compacting hoist): token m goes to key slot slot[m] (< 0: not stored) and the slots tok0 + m >= n_valid_keys
// of this tile's own range are zero-filled (masked keys must still hold finite data: 0 x NaN would poison P V).
template <bool LNLDS = false>
__device__ __forceinline__ void kv_project_tile_x(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, const float* T,
                                                  xhalf* P1, xhalf* __restrict__ Kmat, xhalf* __restrict__ VT, int n_key_pad, int tok0,
                                                  int n_real_rows, int tid, WUnitX& u, const WNextX& nxt, const float* lnblk = nullptr,
                                                  const int* slot = nullptr, int n_valid_keys = 0) {
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
    if (slot) {
        const int sl = slot[m];
        if (sl >= 0) kv_store_key_x(Kmat, VT, sl, wave, lane, ak, av, true);
        if (tok0 + m >= n_valid_keys) kv_store_key_x(Kmat, VT, tok0 + m, wave, lane, ak, av, false);
    } else {
        kv_store_x(Kmat, VT, tok0, wave, lane, ak, av, m < n_real_rows);
    }
    __syncthreads();
}

// (x - mean) rstd of a [16][128] fp32 LDS tile as planes: the LayerNorm without its affine (folded into the consumer's weights)
__device__ __forceinline__ void normalize_planes(const float* src, int lds_, xhalf* P, int tid) {
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    const f32x4 a = lds4(src + row * lds_ + c0), c = lds4(src + row * lds_ + c0 + 4);
    const float s = row16_sum((a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w));
    const float mean = s * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean), dc = c - splat(mean);
    const float v = row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) +
                              (dc.z * dc.z + dc.w * dc.w));
    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    planes_store4<false>(P, PLANE, LDP, row, c0, da * splat(rstd));
    planes_store4<false>(P, PLANE, LDP, row, c0 + 4, dc * splat(rstd));
}

// K/V of the tile for the three interaction layers from ONE normalisation (norm_tgt folded into kvf / bkvf, PolicyWX).
// u holds the K unit of layer 0 on entry and `nxt` on exit; Kmat / VT point at layer 0, layer l at + 2 l ls (fp16).
template <class R = RangeFlag>
__device__ __forceinline__ void kv_project_shared_x(const float* __restrict__ W, const uint32_t (&kvf)[3], const uint32_t (&bkvf)[3],
                                                    const float* T, xhalf* P1, xhalf* __restrict__ Kmat, xhalf* __restrict__ VT, size_t ls,
                                                    int tok0, int n_real_rows, int tid, WUnitX& u, const WNextX& nxt, R&& amax = R{},
                                                    int n_layers = 3) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    normalize_planes(T, LDT, P1, tid);
    __syncthreads();
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    WUnitX u2;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l >= n_layers) break;  // (layers 1, 2 are the helper workgroup's when the launch has one: kv_helper_x)
        f32x4 ak[2] = {u.b[0], u.b[1]};
        wmmax_pf(ak[0], ak[1], u, b1, PLANE, u2, wnextx(W, kvf[l], W + bkvf[l], 8 + 2 * wave, 8 + 2 * wave + 1), lane);
        f32x4 av[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(av[0], av[1], u2, b1, PLANE, u, (l < 2 && l + 1 < n_layers) ? wstdx(W, kvf[l + 1], W + bkvf[l + 1], wave) : nxt, lane);
        kv_store_x(Kmat + 2 * l * ls, VT + 2 * l * ls, tok0, wave, lane, ak, av, m < n_real_rows, amax);
    }
    __syncthreads();
}

// The same layer for TWO 16-row tiles of one group (X0 / X1, their plane buffers P1a / P1b and P2a / P2b): every weight unit is
// loaded once and multiplied against both tiles, which halves the weight stream per row -- the encoders' blocks are bound by it.
// The two tiles walk the same keys; their attentions run one after the other (the out-projection unit is requested by the first).
template <bool SELFMASK = false>
__device__ __forceinline__ void xattn_layer_x2(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X0, float* X1,
                                               xhalf* P1a, xhalf* P1b, xhalf* P2a, xhalf* P2b, const xhalf* __restrict__ Kmat,
                                               const xhalf* __restrict__ VT, const float* __restrict__ keybias, int n_key_pad,
                                               int self_key0a, int self_key0b, const uint8_t* rowvalid0, const uint8_t* rowvalid1,
                                               uint8_t* novalid_s0, uint8_t* novalid_s1, int tid, WUnitX& u, const WNextX& nxt,
                                               const xhalf* __restrict__ Kmat1 = nullptr, const xhalf* __restrict__ VT1 = nullptr,
                                               const float* __restrict__ keybias1 = nullptr) {
    // (Kmat1 / VT1 / keybias1: the second tile's own key group -- the head tiles of two polylines share a workgroup, see
    // k_xattn_block_plh; nullptr = both tiles belong to one group and walk the same keys)
    if (!Kmat1) { Kmat1 = Kmat; VT1 = VT; keybias1 = keybias; }
    const float* lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int po = m * LDP + kq * 8;
    AttnPreX apre;
    layernorm_planes<false>(X0, LDT, P1a, lnblk, lnblk + 128, tid);
    layernorm_planes<false>(X1, LDT, P1b, lnblk, lnblk + 128, tid);
    attention_prefetch_x(apre, Kmat, VT, keybias, n_key_pad, 0, wave, lane);
    __syncthreads();
    WUnitX u2;
    f32x4 q0[2] = {u.b[0], u.b[1]}, q1[2] = {u.b[0], u.b[1]};
    wmmax_pf(q0[0], q0[1], u, P1a + po, PLANE, u2, wstdx(W, LX.wo, W + L.bo, wave), lane);  // (the out-projection unit: see xattn_layer_x)
    wmmax(q1[0], q1[1], u, P1b + po, PLANE);
    f32x4 o0[2], o1[2];
    const bool nov0 = attention_head_x<SELFMASK, false>(q0, apre, Kmat, VT, keybias, n_key_pad, 0, wave, lane, self_key0a >= 0 ? self_key0a + m : -1,
                                                        o0, u2, wstdx(W, LX.wo, W + L.bo, wave));
    attention_prefetch_x(apre, Kmat1, VT1, keybias1, n_key_pad, 0, wave, lane);
    const bool nov1 = attention_head_x<SELFMASK, false>(q1, apre, Kmat1, VT1, keybias1, n_key_pad, 0, wave, lane, self_key0b >= 0 ? self_key0b + m : -1,
                                                        o1, u2, wstdx(W, LX.wo, W + L.bo, wave));
    planes_store_c(P2a, 2 * wave, lane, o0[0]);
    planes_store_c(P2a, 2 * wave + 1, lane, o0[1]);
    planes_store_c(P2b, 2 * wave, lane, o1[0]);
    planes_store_c(P2b, 2 * wave + 1, lane, o1[1]);
    if (wave == 0 && kq == 0) {
        novalid_s0[m] = nov0 ? 1 : 0;
        novalid_s1[m] = nov1 ? 1 : 0;
    }
    __syncthreads();
    {
        f32x4 a0[2] = {u2.b[0], u2.b[1]}, a1[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(a0[0], a0[1], u2, P2a + po, PLANE, u, wstdx(W, LX.w1, W + L.b1, wave), lane);
        wmmax(a1[0], a1[1], u2, P2b + po, PLANE);
        const bool nv0 = novalid_s0[m] != 0, nv1 = novalid_s1[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X0, LDT, 2 * wave + t, lane);
            const f32x4 x0 = lds4(px);
            st4(px, nv0 ? x0 : x0 + a0[t]);
            float* py = cptr(X1, LDT, 2 * wave + t, lane);
            const f32x4 x1 = lds4(py);
            st4(py, nv1 ? x1 : x1 + a1[t]);
        }
    }
    __syncthreads();
    layernorm_planes<false>(X0, LDT, P1a, lnblk + 512, lnblk + 640, tid);
    layernorm_planes<false>(X1, LDT, P1b, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    {
        f32x4 a0[2] = {u.b[0], u.b[1]}, a1[2] = {u.b[0], u.b[1]};
        wmmax_pf(a0[0], a0[1], u, P1a + po, PLANE, u2, wstdx(W, LX.w2, W + L.b2, wave), lane);
        wmmax(a1[0], a1[1], u, P1b + po, PLANE);
        planes_store_c(P2a, 2 * wave, lane, relu4(a0[0]));
        planes_store_c(P2a, 2 * wave + 1, lane, relu4(a0[1]));
        planes_store_c(P2b, 2 * wave, lane, relu4(a1[0]));
        planes_store_c(P2b, 2 * wave + 1, lane, relu4(a1[1]));
    }
    __syncthreads();
    {
        f32x4 a0[2] = {u2.b[0], u2.b[1]}, a1[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(a0[0], a0[1], u2, P2a + po, PLANE, u, nxt, lane);
        wmmax(a1[0], a1[1], u2, P2b + po, PLANE);
        const bool rv0 = rowvalid0[m] != 0, rv1 = rowvalid1[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X0, LDT, 2 * wave + t, lane);
            st4(px, rv0 ? lds4(px) + a0[t] : splat(0.f));
            float* py = cptr(X1, LDT, 2 * wave + t, lane);
            st4(py, rv1 ? lds4(py) + a1[t] : splat(0.f));
        }
    }
    __syncthreads();
}

// K/V projection of TWO 16-token tiles of a group for one layer with each unit loaded once
__device__ __forceinline__ void kv_project_tile_x2(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, const float* T0,
                                                   const float* T1, xhalf* P1a, xhalf* P1b, xhalf* __restrict__ Kmat, xhalf* __restrict__ VT,
                                                   int tok0, int n_real0, int n_real1, int tid, WUnitX& u, const WNextX& nxt,
                                                   xhalf* __restrict__ Kmat1 = nullptr, xhalf* __restrict__ VT1 = nullptr) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const float* lnblk = W + L.ln1_g;
    layernorm_planes<false>(T0, LDT, P1a, lnblk + 256, lnblk + 384, tid);
    layernorm_planes<false>(T1, LDT, P1b, lnblk + 256, lnblk + 384, tid);
    __syncthreads();
    const int po = m * LDP + kq * 8;
    WUnitX u2;
    f32x4 k0[2] = {u.b[0], u.b[1]}, k1[2] = {u.b[0], u.b[1]};
    wmmax_pf(k0[0], k0[1], u, P1a + po, PLANE, u2, wnextx(W, LX.wkv, W + L.bkv, 8 + 2 * wave, 8 + 2 * wave + 1), lane);
    wmmax(k1[0], k1[1], u, P1b + po, PLANE);
    f32x4 v0[2] = {u2.b[0], u2.b[1]}, v1[2] = {u2.b[0], u2.b[1]};
    wmmax_pf(v0[0], v0[1], u2, P1a + po, PLANE, u, nxt, lane);
    wmmax(v1[0], v1[1], u2, P1b + po, PLANE);
    kv_store_x(Kmat, VT, tok0, wave, lane, k0, v0, m < n_real0);
    if (Kmat1) kv_store_x(Kmat1, VT1, tok0, wave, lane, k1, v1, m < n_real1);  // (the second tile is another group's: same token range)
    else kv_store_x(Kmat, VT, tok0 + TM, wave, lane, k1, v1, m < n_real1);
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Packed polyline tails (map encoder, 20 nodes per polyline = one 16-row head tile + 4 tail rows): the tail rows of FOUR consecutive
// polylines share one 16-row tile -- row m belongs to group m >> 2 and is its node 16 + (m & 3) -- instead of four tiles with twelve
// padding rows each.  Every row-wise stage (LayerNorm, the Linears) is unchanged; what differs is where a row's keys are:
//   * K / V projection: each lane stores its token into ITS group's key block (slot 16 + (m & 3)) and zero-fills the slots 20 .. 31 of
//     that block (masked keys must hold finite data);
//   * attention: one 32-key walk per group, every row keeps the result of its own group's walk.
// Same arithmetic per row as the padded tiling -> bitwise identical outputs (tests/test_gpu_configs.py).
// `gstride`: fp16 elements between the key blocks of consecutive groups (same layer).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kv_project_tile_xt(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, const float* T,
                                                   xhalf* P1, xhalf* __restrict__ Kmat, xhalf* __restrict__ VT, size_t gstride, int tid,
                                                   WUnitX& u, const WNextX& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const float* lnblk = W + L.ln1_g;
    layernorm_planes<false>(T, LDT, P1, lnblk + 256, lnblk + 384, tid);
    __syncthreads();
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    WUnitX u2;
    f32x4 ak[2] = {u.b[0], u.b[1]};
    wmmax_pf(ak[0], ak[1], u, b1, PLANE, u2, wnextx(W, LX.wkv, W + L.bkv, 8 + 2 * wave, 8 + 2 * wave + 1), lane);
    f32x4 av[2] = {u2.b[0], u2.b[1]};
    wmmax_pf(av[0], av[1], u2, b1, PLANE, u, nxt, lane);
    xhalf* kg = Kmat + (size_t)(m >> 2) * gstride;
    xhalf* vg = VT + (size_t)(m >> 2) * gstride;
    kv_store_key_x(kg, vg, 16 + (m & 3), wave, lane, ak, av, true);
    // zero the key slots 20 .. 31 of the four key blocks with whole granules (this wave's head; lane group kq takes polyline kq):
    // K -- key tile 1, rows 4 .. 15, all four feature quads: 48 granules of 16 B per plane, three per lane;
    // V -- both d tiles, key quads 1 .. 3, elements 4 .. 7 (= keys 16 + 4 quad + e - 4) of d row m: six 8-byte stores per plane
    {
        xhalf* kz = Kmat + (size_t)kq * gstride + wave * (NPL * 1024);
        xhalf* vz = VT + (size_t)kq * gstride + wave * (NPL * 1024);
        const xh8 z8 = {};
        const xh4 z4 = {};
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int idx = m * 3 + c;  // 0 .. 47 -> (row 4 + idx / 4, feature quad idx % 4)
                *reinterpret_cast<xh8*>(kz + pl * 1024 + (64 + (idx & 3) * 16 + 4 + (idx >> 2)) * 8) = z8;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int vq = 1; vq < 4; ++vq) *reinterpret_cast<xh4*>(vz + pl * 1024 + (t * 64 + vq * 16 + m) * 8 + 4) = z4;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void xattn_layer_xt(const float* __restrict__ W, const XLayerW& L, const XLayerX& LX, float* X, xhalf* P1,
                                               xhalf* P2, const xhalf* __restrict__ Kmat, const xhalf* __restrict__ VT,
                                               const float* __restrict__ keybias, size_t gstride, const uint8_t* rowvalid,
                                               uint8_t* novalid_s, int tid, WUnitX& u, const WNextX& nxt) {
    const float* lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    const xhalf* b2 = P2 + m * LDP + kq * 8;
    AttnPreX apre;
    WUnitX u2;
    layernorm_planes<false>(X, LDT, P1, lnblk, lnblk + 128, tid);
    attention_prefetch_x(apre, Kmat, VT, keybias, KEYPAD, 0, wave, lane);
    __syncthreads();
    f32x4 q[2] = {u.b[0], u.b[1]};
    wmmax_pf(q[0], q[1], u, b1, PLANE, u2, wstdx(W, LX.wo, W + L.bo, wave), lane);
    f32x4 osel[2] = {splat(0.f), splat(0.f)};
    bool novsel = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const xhalf* kj = Kmat + (size_t)j * gstride;
        const xhalf* vj = VT + (size_t)j * gstride;
        const float* bj = keybias + j * KEYPAD;
        if (j > 0) attention_prefetch_x(apre, kj, vj, bj, KEYPAD, 0, wave, lane);
        f32x4 o[2];
        const bool nov = attention_head_x<false, false>(q, apre, kj, vj, bj, KEYPAD, 0, wave, lane, -1, o, u2, wstdx(W, LX.wo, W + L.bo, wave));
        if ((m >> 2) == j) {
            osel[0] = o[0];
            osel[1] = o[1];
            novsel = nov;
        }
    }
    planes_store_c<false>(P2, 2 * wave, lane, osel[0]);
    planes_store_c<false>(P2, 2 * wave + 1, lane, osel[1]);
    if (wave == 0 && kq == 0) novalid_s[m] = novsel ? 1 : 0;
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
    layernorm_planes<false>(X, LDT, P1, lnblk + 512, lnblk + 640, tid);
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
        if (OutP) planes_store_c<false>(OutP, tile, lane, hn);  // (|h| <= 1)
        if (Out) st4(cptr(Out, LDT, tile, lane), hn);
        if (m < n_real_rows) st4(h_global + (size_t)m * H + tile * 16 + kq * 4, hn);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// GRU layer step of the STEP kernels with the hidden-side products kept apart:  gh_g = W_hh,g h  for the gates g = r, z, n
// depends only on the previous step's hidden state, i.e. on data that exists when the launch starts, so at small batch sizes a
// HELPER workgroup on an otherwise idle CU computes it while the tile's own workgroup runs the interaction block (gru_hh_helper),
// and hands it over through L2.  Both forms add gh to the input-side accumulator in the same order, so a tile's result does not
// depend on who computed gh (bitwise: tests/test_gpu_parity.py).
//   gh layout (floats) per (instance, row tile): [layer 3][gate 3][wave 4][tile 2][lane 64][4]
// ---------------------------------------------------------------------------------------------
constexpr int GH_TILE_FLOATS = 3 * 3 * 4 * 2 * 64 * 4;  // 18432 = 72 KiB

struct GruGH {
    f32x4 g[3][2];  // W_hh,g h per [gate r, z, n][tile], WITHOUT its bias
    f32x4 b[3][2];  // b_hh,g (added in the order of the single-accumulator form: (b_ih + W_ih x) + b_hh, then the hidden-side products)
};

// hand-off granules: 8-byte relaxed agent-scope atomics on both sides (`global_{load,store}_dwordx2 ... sc1`: L2-served / written
// through, never from a stale L1 line); the flag follows the payload behind `s_waitcnt vmcnt(0)` + a workgroup barrier
// (MI355X_MICROARCH.md, inter-workgroup visibility: "{8-B agent atomics both sides}")
__device__ __forceinline__ void gh_store(float* __restrict__ dst, const f32x4& v) {
    unsigned long long lo, hi;
    const float a[2] = {v.x, v.y}, b[2] = {v.z, v.w};
    __builtin_memcpy(&lo, a, 8);
    __builtin_memcpy(&hi, b, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 gh_load(const float* __restrict__ src) {
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float a[2], b[2];
    __builtin_memcpy(a, &lo, 8);
    __builtin_memcpy(b, &hi, 8);
    return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ void gh_load_layer(GruGH& gh, const float* __restrict__ gh_tile, const float* __restrict__ bhh, int layer, int wave,
                                              int lane) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            gh.g[g][t] = gh_load(gh_tile + ((((size_t)(layer * 3 + g) * 4 + wave) * 2 + t) * 64 + lane) * 4);
            gh.b[g][t] = ldg4(bhh + (g * 8 + 2 * wave + t) * 16 + (lane >> 4) * 4);
        }
}

// the gate arithmetic + stores shared by both forms
// (Hs / hs_ld: the fp32 previous hidden state -- the tile's LDS copy, or with hs_ld = H the rows in the rollout workspace themselves)
__device__ __forceinline__ void gru_finish_x(const f32x4 (&ri)[2], const f32x4 (&zi)[2], const f32x4 (&ni)[2], const GruGH& gh, const float* Hs,
                                             xhalf* OutP, float* Out, const uint8_t* rowvalid, float* __restrict__ h_global, int n_real_rows, int wave,
                                             int lane, int hs_ld = LDT) {
    const int kq = lane >> 4, m = lane & 15;
    const bool rv = rowvalid[m] != 0;
    f32x4 holdv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) holdv[t] = lds4(Hs + m * hs_ld + (2 * wave + t) * 16 + kq * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = 2 * wave + t;
        const f32x4 hold = holdv[t];
        const f32x4 r = (ri[t] + gh.b[0][t]) + gh.g[0][t], z = (zi[t] + gh.b[1][t]) + gh.g[1][t];
        f32x4 hn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rg = sigmoidf_(r[q]);
            const float zg = sigmoidf_(z[q]);
            const float ng = tanhf_(ni[t][q] + rg * (gh.b[2][t][q] + gh.g[2][t][q]));
            hn[q] = rv ? (1.0f - zg) * ng + zg * hold[q] : 0.f;
        }
        if (OutP) planes_store_c<false>(OutP, tile, lane, hn);  // (|h| <= 1)
        if (Out) st4(cptr(Out, LDT, tile, lane), hn);
        if (m < n_real_rows) st4(h_global + (size_t)m * H + tile * 16 + kq * 4, hn);
    }
    __syncthreads();
}

// form 1: the workgroup computes gh itself -- six weight units; u = the W_ih r unit on entry, `nxt` on exit
__device__ __forceinline__ void gru_layer_own_x(const float* __restrict__ W, const GruLayerW& G, const GruLayerX& GX, const xhalf* XinP,
                                                const xhalf* HsP, const float* Hs, xhalf* OutP, float* Out, const uint8_t* rowvalid,
                                                float* __restrict__ h_global, int n_real_rows, int tid, WUnitX& u, const WNextX& nxt,
                                                int hs_ld = LDT) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int ta = 2 * wave, tb_ = 2 * wave + 1;
    const xhalf* xr = XinP + m * LDP + kq * 8;
    const xhalf* hr = HsP + m * LDP + kq * 8;
    const float* bih = W + G.bih;
    const float* bhh = W + G.bhh;
    WUnitX u2;
    GruGH gh;
    f32x4 ri[2] = {u.b[0], u.b[1]};
    wmmax_pf(ri[0], ri[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, ta, tb_), lane);
    gh.b[0][0] = u2.b[0]; gh.b[0][1] = u2.b[1];
    gh.g[0][0] = splat(0.f); gh.g[0][1] = splat(0.f);
    wmmax_pf(gh.g[0][0], gh.g[0][1], u2, hr, PLANE, u, wnextx(W, GX.wih, bih, 8 + ta, 8 + tb_), lane);
    f32x4 zi[2] = {u.b[0], u.b[1]};
    wmmax_pf(zi[0], zi[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, 8 + ta, 8 + tb_), lane);
    gh.b[1][0] = u2.b[0]; gh.b[1][1] = u2.b[1];
    gh.g[1][0] = splat(0.f); gh.g[1][1] = splat(0.f);
    wmmax_pf(gh.g[1][0], gh.g[1][1], u2, hr, PLANE, u, wnextx(W, GX.wih, bih, 16 + ta, 16 + tb_), lane);
    f32x4 ni[2] = {u.b[0], u.b[1]};
    wmmax_pf(ni[0], ni[1], u, xr, PLANE, u2, wnextx(W, GX.whh, bhh, 16 + ta, 16 + tb_), lane);
    gh.b[2][0] = u2.b[0]; gh.b[2][1] = u2.b[1];
    gh.g[2][0] = splat(0.f); gh.g[2][1] = splat(0.f);
    wmmax_pf(gh.g[2][0], gh.g[2][1], u2, hr, PLANE, u, nxt, lane);
    gru_finish_x(ri, zi, ni, gh, Hs, OutP, Out, rowvalid, h_global, n_real_rows, wave, lane, hs_ld);
}

// form 2: gh comes from the helper -- three weight units.  ua = the W_ih r unit on entry; the unit after the layer (`nxt`) lands in ub
__device__ __forceinline__ void gru_layer_gh_x(const float* __restrict__ W, const GruLayerW& G, const GruLayerX& GX, const xhalf* XinP, const float* Hs,
                                               xhalf* OutP, float* Out, const uint8_t* rowvalid, float* __restrict__ h_global, int n_real_rows,
                                               int tid, WUnitX& ua, WUnitX& ub, const WNextX& nxt, const GruGH& gh) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int ta = 2 * wave, tb_ = 2 * wave + 1;
    const xhalf* xr = XinP + m * LDP + kq * 8;
    const float* bih = W + G.bih;
    f32x4 ri[2] = {ua.b[0], ua.b[1]};
    wmmax_pf(ri[0], ri[1], ua, xr, PLANE, ub, wnextx(W, GX.wih, bih, 8 + ta, 8 + tb_), lane);
    f32x4 zi[2] = {ub.b[0], ub.b[1]};
    wmmax_pf(zi[0], zi[1], ub, xr, PLANE, ua, wnextx(W, GX.wih, bih, 16 + ta, 16 + tb_), lane);
    f32x4 ni[2] = {ua.b[0], ua.b[1]};
    wmmax_pf(ni[0], ni[1], ua, xr, PLANE, ub, nxt, lane);
    gru_finish_x(ri, zi, ni, gh, Hs, OutP, Out, rowvalid, h_global, n_real_rows, wave, lane);
}

// the helper workgroup: gh of the three layers from the previous step's hidden state (global, fp32), 9 weight units.
//   smem : 3 plane buffers;  hidden_l = p.hidden + ((l * n_inst + n) * a_pad + row0) * H
__device__ __forceinline__ void gru_hh_helper(const float* __restrict__ W, const GruLayerX (&GX)[3],
                                              const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ h2,
                                              float* __restrict__ gh_tile, unsigned int* __restrict__ flag, unsigned int token, xhalf* smem_planes,
                                              int tid, WUnitX& ua, bool ua_loaded) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int ta = 2 * wave, tb_ = 2 * wave + 1;
    xhalf* HP[3] = {smem_planes, smem_planes + NPL * PLANE, smem_planes + 2 * NPL * PLANE};
    const float* hsrc[3] = {h0, h1, h2};
    WUnitX ub;
    if (!ua_loaded) wloadx(ua, wnextx(W, GX[0].whh, nullptr, ta, tb_), lane);
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NTHREADS;
            const int row = idx >> 5, c4 = (idx & 31) * 4;
            planes_store4<false>(HP[l], PLANE, LDP, row, c4, ldg4(hsrc[l] + (size_t)row * H + c4));  // (|h| <= 1)
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int l = i / 3, g = i % 3;
        const int ln = (i + 1) / 3, gn = (i + 1) % 3;
        WUnitX& cur = (i & 1) ? ub : ua;
        WUnitX& nx_ = (i & 1) ? ua : ub;
        // (the tenth request re-reads the first unit: nobody consumes it)
        const WNextX nn = i < 8 ? wnextx(W, GX[ln].whh, nullptr, gn * 8 + ta, gn * 8 + tb_) : wnextx(W, GX[0].whh, nullptr, ta, tb_);
        f32x4 acc[2] = {splat(0.f), splat(0.f)};
        wmmax_pf(acc[0], acc[1], cur, HP[l] + m * LDP + kq * 8, PLANE, nx_, nn, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) gh_store(gh_tile + ((((size_t)(l * 3 + g) * 4 + wave) * 2 + t) * 64 + lane) * 4, acc[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's payload is at L2 before ...
    __syncthreads();                                  // ... the workgroup ...
    if (tid == 0) __hip_atomic_store(flag, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ... raises the tile's flag
}

// ---------------------------------------------------------------------------------------------
// Interaction K / V of layers 1 and 2 on the helper workgroup.  The K / V of step t's interaction depend only on x_mid(t), which the
// previous launch stored, so the tile's own workgroup projects layer 0 only (the one its siblings need first) at the end of A(t) and
// the helper workgroup of launch t projects layers 1 and 2 from the stored x_mid while the tile workgroups run layer 0.
//   payload: 8-byte relaxed agent-scope atomic stores (written through to L2 / memory), then `s_waitcnt vmcnt(0)` + barrier + flag;
//   the consumers poll the flags of all row tiles of their instance (kv_wait_x), then read K / V with ordinary loads: no workgroup
//   touches those lines earlier in the launch, so neither its L1 nor its L2 can hold an older copy.
// V fragments interleave the keys of two row tiles inside 16 bytes: the helper transposes its tile's half through LDS (wave-private,
// 2 KB per wave) and stores 8-byte granules.  Same arithmetic and the same bytes as kv_project_shared_x (bitwise: the tests).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st8_wt(xhalf* dst, const xh4& v) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class R>
__device__ __forceinline__ void kv_store_wt_x(xhalf* __restrict__ Kf, xhalf* __restrict__ Vf, int tok0, int wave, int lane, const f32x4 (&ak)[2],
                                              const f32x4 (&av)[2], xhalf* vstage, R&& amax) {
    const int kq = lane >> 4, m = lane & 15;
    const int tok = tok0 + m, j = tok & 31;
    xhalf* kblk = Kf + (size_t)(tok >> 5) * KV_BLOCK_HALFS + wave * (NPL * 1024);
    xhalf* vblk = Vf + (size_t)(tok0 >> 5) * KV_BLOCK_HALFS + wave * (NPL * 1024);
    const int kt = j >> 4, krow = j & 15;
    const int vq = (j >> 2) & 3, e = j & 3;
    xhalf* vs = vstage + wave * (NPL * 2 * 64 * 4);  // [plane][d tile][slot 64][4 keys of this row tile]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        xh4 h, l;
        split2(ak[t], h, l, amax);
        xhalf* pk = kblk + (kt * 64 + kq * 16 + krow) * 8 + t * 4;
        st8_wt(pk, h);
        if (NPL == 2) st8_wt(pk + 1024, l);
        split2(av[t], h, l, amax);
        xhalf* ps = vs + ((t * 64) + vq * 16 + kq * 4) * 4 + e;
        ps[0] = h.x; ps[4] = h.y; ps[8] = h.z; ps[12] = h.w;
        if (NPL == 2) {
            ps += 2 * 64 * 4;
            ps[0] = l.x; ps[4] = l.y; ps[8] = l.z; ps[12] = l.w;
        }
    }
    // (one wave: LDS operations complete in issue order; the fence below only stops the compiler from moving the reads up)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int half_off = ((tok0 >> 4) & 1) * 4;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const xh4 v = *reinterpret_cast<const xh4*>(vs + ((pl * 2 + t) * 64 + lane) * 4);
            st8_wt(vblk + pl * 1024 + (t * 64 + lane) * 8 + half_off, v);
        }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// T: [16][LDT] fp32 (LDS), P1: one plane buffer, vstage: 4 x NPL x 2 KB / 2 (LDS).  On exit `u` holds the unit `nxt`.
template <class R>
__device__ __forceinline__ void kv_helper_x(const float* __restrict__ W, const uint32_t (&kvf)[3], const uint32_t (&bkvf)[3],
                                            const float* __restrict__ x_mid_tile, float* T, xhalf* P1, xhalf* vstage, xhalf* __restrict__ Kmat,
                                            xhalf* __restrict__ VT, size_t ls, int tok0, unsigned int* __restrict__ flags, unsigned int token,
                                            int tid, WUnitX& u, const WNextX& nxt, R&& amax) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    wloadx(u, wstdx(W, kvf[1], W + bkvf[1], wave), lane);
    load_tile(T, LDT, x_mid_tile, TM, tid);
    __syncthreads();
    normalize_planes(T, LDT, P1, tid);
    __syncthreads();
    const xhalf* b1 = P1 + m * LDP + kq * 8;
    WUnitX u2;
#pragma unroll
    for (int l = 1; l < 3; ++l) {
        f32x4 ak[2] = {u.b[0], u.b[1]};
        wmmax_pf(ak[0], ak[1], u, b1, PLANE, u2, wnextx(W, kvf[l], W + bkvf[l], 8 + 2 * wave, 8 + 2 * wave + 1), lane);
        f32x4 av[2] = {u2.b[0], u2.b[1]};
        wmmax_pf(av[0], av[1], u2, b1, PLANE, u, l < 2 ? wstdx(W, kvf[2], W + bkvf[2], wave) : nxt, lane);
        kv_store_wt_x(Kmat + 2 * l * ls, VT + 2 * l * ls, tok0, wave, lane, ak, av, vstage, amax);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this also waits for the next unit: the helper is not on the critical path)
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + (l - 1), token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// consumer side: wait until every row tile of the instance has raised its flag of `layer` (1 or 2) for this step.  The helpers are
// dispatched in front of the tile workgroups (blockIdx.z = 0) and wait for nothing, so the wait is bounded; the bound only turns a
// broken assumption into a reported error (RolloutP::sync_err -> tb_check_status) instead of a hang.
// kv_peek_x requests the flags one layer early (the value rides in a register through that layer); kv_wait_x polls only if the
// early value was not the token yet, so on the common path the wait is a compare and a barrier, not a round trip to L2.
__device__ __forceinline__ unsigned int kv_peek_x(const unsigned int* __restrict__ inst_flags, int n_rt, int layer, int tid) {
    return tid < n_rt ? __hip_atomic_load(inst_flags + tid * 2 + (layer - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
// A wait that gives up does NOT let the tile continue silently on stale K / V: it raises the context's sticky word
// (tb_check_status) AND the tile's `poison` word in LDS, which makes the epilogue write NaN as this step's prediction and state -- the
// NaN then spreads through the interaction to the whole instance, so any consumer of the trajectories sees it, checked or not.
__device__ __forceinline__ void kv_wait_x(const unsigned int* __restrict__ inst_flags, int n_rt, int layer, unsigned int token, int tid,
                                          unsigned int* __restrict__ sync_err, unsigned int seen, int* poison) {
    if (tid < n_rt && seen != token) {
        const unsigned int* f = inst_flags + tid * 2 + (layer - 1);
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != token) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 18)) {
                *sync_err = 1u;
                *poison = 1;
                break;
            }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
}

}  // namespace TB_XNS
using namespace TB_XNS;
}  // namespace tb
