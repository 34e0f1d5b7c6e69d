// Device building blocks for the gfx950 rollout kernels (wave64, fp32 MFMA 16x16x4).
//
// Geometry used everywhere: one workgroup = 256 threads = 4 waves (one per SIMD) owns a tile of TM = 16 rows
// (agents / tokens).  Activations of the tile live in LDS as row-major [16][ld] fp32; every Linear
// is computed in TRANSPOSED form  Y^T = W . X^T  with v_mfma_f32_16x16x4_f32:
//     A operand = W   (lane l: W[n0 + (l&15)][k(l>>4, step)])          -- pre-packed, 16 B / lane / 4 steps
//     B operand = X^T (lane l: X[agent = l&15][k(l>>4, step)])        -- ds_read_b128 from the LDS tile
//     D         = Y^T (lane l, reg r: feature n0 + (l>>4)*4 + r, agent l&15)
// so a lane ends up with 4 CONSECUTIVE features of one agent: bias add, activation and the store back
// to the row-major tile are float4 operations, and Q^T / P^T come out of the MFMA exactly in the layout the
// next MFMA wants as its B operand (attention never round-trips through LDS).
// The k index consumed by MFMA step (j, i) in lane group kq = l>>4 is  k = kq*(K/4) + 4*j + i  (any bijection
// works as long as A and B agree; this one makes both operands float4-contiguous).
//
// Latency: with one wave per SIMD nothing hides a load, so weights move as UNITS (2 output tiles x 128 k =
// 16 float4 = 64 VGPRs per lane, 64 MFMAs of work) that are fetched one unit AHEAD of their use: every stage
// function receives its first unit already in registers and leaves the next stage's first unit loading
// (`WNext`), so the ~2k cycles of MFMA of one unit cover the L2 / Infinity-Cache latency of the next.
// Attention K/V tiles are double-buffered the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 128;       // hidden_dim (traffic_bots.yaml:9)
constexpr int NHEAD = 4;     // tf_cfg.n_head
constexpr int DHEAD = 32;
constexpr int TM = 16;       // rows per workgroup tile
constexpr int LDT = 132;     // LDS row stride (floats) of a [16][128] tile (16-B aligned rows)
constexpr int LDC = 260;     // LDS row stride of a [16][256] concat tile
constexpr int NTHREADS = 256;
constexpr int KEYPAD = 32;   // key counts are padded to a multiple of 32 (attention loop is unrolled by two tiles)
constexpr float LN_EPS = 1e-5f;
constexpr float ATTN_SCALE = 0.17677669529663687f;  // 1/sqrt(32)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// wave index inside the workgroup as a wave-UNIFORM (SGPR) value: everything derived from it (weight tile offsets, head
// offsets, K/V bases) then stays in scalar registers instead of per-lane 64-bit VALU address math
__device__ __forceinline__ int wave_of(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// ---------------------------------------------------------------------------------------------
// weight units
// ---------------------------------------------------------------------------------------------
struct WUnit {
    f32x4 w[2][8];  // [tile][k-step group]
    f32x4 b[2];     // bias of the two tiles for this lane's 4 features (zero when the unit continues a reduction)
};

// Where the next unit lives: packed matrix, the two 16-feature tiles, and the k-window (kj_total = K/16,
// j0 = first k-step group of this unit; K = 256 matrices are two units with j0 = 0 and 8).
struct WNext {
    const float* wpk;
    const float* bias;  // bias vector indexed by output feature, or nullptr
    int tile_a, tile_b, kj_total, j0;
};

__device__ __forceinline__ WNext wnext(const float* wpk, const float* bias, int tile_a, int tile_b, int kj_total = 8, int j0 = 0) {
    return WNext{wpk, bias, tile_a, tile_b, kj_total, j0};
}
// the standard Linear(128 -> 128) unit of wave `wave`: tiles {2w, 2w+1}
__device__ __forceinline__ WNext wstd(const float* wpk, const float* bias, int wave) {
    return WNext{wpk, bias, 2 * wave, 2 * wave + 1, 8, 0};
}

// The scheduler must not sink these loads below the MFMAs that follow (it would: their results are needed late),
// nor hoist later loads above them (s_waitcnt vmcnt counts in issue order: anything needed EARLY -- biases --
// has to be requested BEFORE a wload).  sched_barrier(0) pins both sides.
#define TB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void wload(WUnit& u, const WNext& n, int lane) {
    const float* pa_ = n.wpk + ((size_t)(n.tile_a * n.kj_total + n.j0) * 64 + lane) * 4;
    const float* pb_ = n.wpk + ((size_t)(n.tile_b * n.kj_total + n.j0) * 64 + lane) * 4;
#ifdef TB_FAKE_W  // timing experiment only: every unit reads the same 16 KiB (L1-resident) -> results are wrong
    const float* pa = n.wpk + (size_t)lane * 4;
    const float* pb = pa + 64 * 4;
    (void)pa_; (void)pb_;
#else
    const float* pa = pa_;
    const float* pb = pb_;
#endif
    TB_SCHED_FENCE();
    if (n.bias) {
        u.b[0] = ldg4(n.bias + n.tile_a * 16 + (lane >> 4) * 4);
        u.b[1] = ldg4(n.bias + n.tile_b * 16 + (lane >> 4) * 4);
    } else {
        u.b[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        u.b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u.w[0][j] = ldg4(pa + j * 256);
        u.w[1][j] = ldg4(pb + j * 256);
    }
    TB_SCHED_FENCE();
}

// acc_a / acc_b += unit . X^T ; xrow = this lane's pointer to X[agent][kq*(K/4) + 4*j0].  The two accumulators are
// interleaved so consecutive MFMAs never depend on each other (16x16x4 f32: 32-cycle issue, 40-cycle dependent).
__device__ __forceinline__ void wmma(f32x4& acc_a, f32x4& acc_b, const WUnit& u, const float* xrow) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const f32x4 xv = lds4(xrow + 4 * j);
        acc_a = mfma4(u.w[0][j].x, xv.x, acc_a);
        acc_b = mfma4(u.w[1][j].x, xv.x, acc_b);
        acc_a = mfma4(u.w[0][j].y, xv.y, acc_a);
        acc_b = mfma4(u.w[1][j].y, xv.y, acc_b);
        acc_a = mfma4(u.w[0][j].z, xv.z, acc_a);
        acc_b = mfma4(u.w[1][j].z, xv.z, acc_b);
        acc_a = mfma4(u.w[0][j].w, xv.w, acc_a);
        acc_b = mfma4(u.w[1][j].w, xv.w, acc_b);
    }
}

// Same MFMA block with the NEXT unit's 18 loads issued in the shadow of the MFMAs (one VMEM per 4 MFMAs): a
// global_load_dwordx4 costs ~30 issue cycles that would otherwise sit in front of the chain.
__device__ __forceinline__ void wmma_pf(f32x4& acc_a, f32x4& acc_b, const WUnit& u, const float* xrow, WUnit& un, const WNext& n,
                                        int lane) {
    const float* pa = n.wpk + ((size_t)(n.tile_a * n.kj_total + n.j0) * 64 + lane) * 4;
    const float* pb = n.wpk + ((size_t)(n.tile_b * n.kj_total + n.j0) * 64 + lane) * 4;
    const float* ba = n.bias ? n.bias + n.tile_a * 16 + (lane >> 4) * 4 : n.wpk;
    const float* bb = n.bias ? n.bias + n.tile_b * 16 + (lane >> 4) * 4 : n.wpk;
    TB_SCHED_FENCE();
    un.b[0] = ldg4(ba);
    un.b[1] = ldg4(bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        un.w[0][j] = ldg4(pa + j * 256);
        un.w[1][j] = ldg4(pb + j * 256);
    }
    f32x4 xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = lds4(xrow + 4 * j);  // all B operands up front: no LDS wait inside the chain
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc_a = mfma4(u.w[0][j].x, xv[j].x, acc_a);
        acc_b = mfma4(u.w[1][j].x, xv[j].x, acc_b);
        acc_a = mfma4(u.w[0][j].y, xv[j].y, acc_a);
        acc_b = mfma4(u.w[1][j].y, xv[j].y, acc_b);
        acc_a = mfma4(u.w[0][j].z, xv[j].z, acc_a);
        acc_b = mfma4(u.w[1][j].z, xv[j].z, acc_b);
        acc_a = mfma4(u.w[0][j].w, xv[j].w, acc_a);
        acc_b = mfma4(u.w[1][j].w, xv[j].w, acc_b);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // the 8 DS reads first
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // bias loads
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    TB_SCHED_FENCE();
    if (!n.bias) {
        un.b[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        un.b[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// bias for the 4 features a lane holds of tile `tile`
__device__ __forceinline__ f32x4 bias4(const float* __restrict__ b, int tile, int lane) {
    return ldg4(b + tile * 16 + (lane >> 4) * 4);
}

// pointer into a row-major LDS tile for the 4 features this lane holds of tile `tile`
__device__ __forceinline__ float* cptr(float* base, int ld, int tile, int lane) {
    return base + (lane & 15) * ld + tile * 16 + (lane >> 4) * 4;
}

// sin and cos of an fp32 argument (|x| < ~1e5), evaluated in fp64 and rounded once
__device__ __forceinline__ void sincos_pe(float xf, float& s_out, float& c_out) {
    const double x = (double)xf;
    const double n = rint(x * 0.63661977236758134308);  // 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);  // pi/2 in two pieces
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double r2 = r * r;
    double sp = fma(r2, -2.50521083854417187751e-08, 2.75573192239858906526e-06);  // 1/11!, 1/9!
    sp = fma(r2, sp, -1.98412698412698412698e-04);
    sp = fma(r2, sp, 8.33333333333333333333e-03);
    sp = fma(r2, sp, -1.66666666666666666667e-01);
    const double sv = fma(r * r2, sp, r);
    double cp = fma(r2, 2.08767569878680989792e-09, -2.75573192239858906526e-07);  // 1/12!, 1/10!
    cp = fma(r2, cp, 2.48015873015873015873e-05);
    cp = fma(r2, cp, -1.38888888888888888889e-03);
    cp = fma(r2, cp, 4.16666666666666666667e-02);
    cp = fma(r2, cp, -0.5);
    const double cv = fma(r2, cp, 1.0);
    const int qd = (int)n & 3;
    const double ss = (qd & 1) ? cv : sv, cc = (qd & 1) ? sv : cv;
    s_out = (float)((qd & 2) ? -ss : ss);
    c_out = (float)(((qd + 1) & 2) ? -cc : cc);
}

// Un-pipelined Linear 128 -> 128 (used by one-time encoder kernels where latency does not matter).
template <int K>
__device__ __forceinline__ void linear128(f32x4 (&acc)[2], const float* __restrict__ wpk, const float* __restrict__ bias,
                                          const float* xrow, int wave, int lane) {
    static_assert(K == 128, "linear128 is the K = 128 form");
    WUnit u;
    wload(u, wstd(wpk, bias, wave), lane);
    acc[0] = u.b[0];
    acc[1] = u.b[1];
    wmma(acc[0], acc[1], u, xrow);
}

// generic (un-pipelined) accumulate for odd shapes: K multiple of 16, NT tiles
template <int K, int NT>
__device__ __forceinline__ void gemm_acc(f32x4 (&acc)[NT], const float* __restrict__ wpk, const int (&tiles)[NT],
                                         const float* xrow, int lane) {
    constexpr int KJ = K / 16;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        const f32x4 xv = lds4(xrow + 4 * j);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 wv = ldg4(wpk + ((size_t)(tiles[t] * KJ + j) * 64 + lane) * 4);
            acc[t] = mfma4(wv.x, xv.x, acc[t]);
            acc[t] = mfma4(wv.y, xv.y, acc[t]);
            acc[t] = mfma4(wv.z, xv.z, acc[t]);
            acc[t] = mfma4(wv.w, xv.w, acc[t]);
        }
    }
}

// sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); every lane gets the total.  quad xor 1, quad xor 2,
// row_half_mirror, row_mirror -- four VALU+DPP ops instead of four ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm of a [16][128] LDS tile: 16 threads per row, 8 elements each (two float4).
// ---------------------------------------------------------------------------------------------
// explicit LDS load: a generic pointer that may or may not be LDS makes the compiler emit FLAT loads, whose completion can
// only be awaited with vmcnt(0) -- that would drain every prefetched weight / K / V load in flight
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
__device__ __forceinline__ f32x4 lds4_explicit(const float* p) {
    return *reinterpret_cast<lds_f32x4*>((__attribute__((address_space(3))) const char*)p);
}

template <bool PARAMS_IN_LDS = false>
__device__ __forceinline__ void layernorm_tile(const float* src, int lds_, float* dst, int ldd,
                                               const float* __restrict__ g, const float* __restrict__ b, int tid) {
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
    st4(dst + row * ldd + c0, da * splat(rstd) * g0 + b0);
    st4(dst + row * ldd + c0 + 4, dc * splat(rstd) * g1 + b1);
}

// ---------------------------------------------------------------------------------------------
// One attention head (this wave's) over n_key_pad keys (multiple of 32) with online softmax.
//   q[tt][r]  = Q^T[h*32 + tt*16 + kq*4 + r][agent]  (this wave's Q-projection accumulators, bias added)
//   Kmat      = [n_key_pad][128] row-major, VT = [128][n_key_pad] (keys contiguous), both in global (L2)
//   keyvalid  = uint8 [n_key_pad] (0 for padding keys)
//   self_key  = key index that equals THIS LANE's agent (eye mask of MultiAgentTF), or -1
// Returns o[dt][r] = O^T[h*32 + dt*16 + kq*4 + r][agent] (already divided by the softmax sum) and
// whether the agent row had no valid key at all (attention.py:101-107: its output is zeroed after out-proj).
// K/V fragments of key tile t+2 are requested while tile t is being reduced (two register buffers).
// ---------------------------------------------------------------------------------------------
// exp(x) for x <= 0 as one v_exp_f32 with a compensated argument: t = rn(x*log2e), e = x*log2e - t (exact via fma + the
// low word of log2e), exp(x) = 2^t * (1 + e*ln2).  Max relative error 1.2e-7 on [-100, 0] (tools/microtests); no
// denormal / overflow handling is needed for softmax arguments.
__device__ __forceinline__ float exp_neg(float x) {
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    x = fmaxf(x, -110.0f);  // 2^(-110 log2 e) underflows to exactly 0; keeps -inf (masked keys) away from inf - inf
    const float t = x * L2E_HI;
    float e = fmaf(x, L2E_HI, -t);
    e = fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * LN2, r);
}

// all-reduce over the four 16-lane rows of the wave (lanes l, l^16, l^32, l^48) with the gfx950 row swaps:
// permlane16_swap(x, x) = {[x0,x0,x2,x2], [x1,x1,x3,x3]}, permlane32_swap(y, y) = {[y0,y1,y0,y1], [y2,y3,y2,y3]}
// Written as inline asm with both registers as read-write operands: the instruction swaps rows between its two
// registers in place (the clang builtin mis-tracks the second result when both inputs hold the same value).
// hipcc pads nothing inside an asm statement, hence the s_nop's (VALU write -> permlane read, permlane -> VALU).
__device__ __forceinline__ void rows_pair16(float v, float& lo, float& hi) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
__device__ __forceinline__ void rows_pair32(float v, float& lo, float& hi) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo = a;
    hi = b;
}
__device__ __forceinline__ float rows_max(float v) {
    float x, y;
    rows_pair16(v, x, y);
    v = fmaxf(x, y);
    rows_pair32(v, x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float rows_sum(float v) {
    float x, y;
    rows_pair16(v, x, y);
    v = x + y;
    rows_pair32(v, x, y);
    return x + y;
}

// K / V^T fragments of 32 keys (two 16-key tiles) for one head
struct KFrag {
    f32x4 ka[2][2];  // [tile][half of the 32 features]
    f32x4 kb[2];     // additive key bias (0 valid / -inf invalid or padding) of this lane's 4 keys per tile
};
struct VFrag {
    f32x4 va[2][2];
};

__device__ __forceinline__ void k_load(KFrag& f, const float* __restrict__ kbase, const float* __restrict__ bbase, int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f.ka[t][0] = ldg4(kbase + (size_t)(k0 + 16 * t) * H);
        f.ka[t][1] = ldg4(kbase + (size_t)(k0 + 16 * t) * H + 16);
        f.kb[t] = ldg4(bbase + k0 + 16 * t);
    }
}
__device__ __forceinline__ void v_load(VFrag& f, const float* __restrict__ vbase, int n_key_pad, int k0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f.va[t][0] = ldg4(vbase + k0 + 16 * t);
        f.va[t][1] = ldg4(vbase + (size_t)16 * n_key_pad + k0 + 16 * t);
    }
}

// S^T tiles of 32 keys: s0 / s1[r] = logit(key k0 + 16 t + kq*4 + r, agent m), two interleaved MFMA chains
__device__ __forceinline__ void attn_qk(const KFrag& f, const f32x4 (&q)[2], f32x4& s0, f32x4& s1) {
    s0 = splat(0.f);
    s1 = splat(0.f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        s0 = mfma4(f.ka[0][h].x, q[h].x, s0);
        s1 = mfma4(f.ka[1][h].x, q[h].x, s1);
        s0 = mfma4(f.ka[0][h].y, q[h].y, s0);
        s1 = mfma4(f.ka[1][h].y, q[h].y, s1);
        s0 = mfma4(f.ka[0][h].z, q[h].z, s0);
        s1 = mfma4(f.ka[1][h].z, q[h].z, s1);
        s0 = mfma4(f.ka[0][h].w, q[h].w, s0);
        s1 = mfma4(f.ka[1][h].w, q[h].w, s1);
    }
}

// scale + mask the 8 logits of a lane and fold them into the running max: sv = logits/sqrt(d) + bias,
// new_max = max(run_max, max over the 32 keys), alpha = exp(run_max - new_max) (1 when nothing is valid yet)
__device__ __forceinline__ void attn_stats(const f32x4& s0, const f32x4& s1, const f32x4 (&kb)[2], int kb0, int self_key,
                                           float run_max, float (&sv)[8], float& new_max, float& alpha) {
    const float raw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float bias[8] = {kb[0].x, kb[0].y, kb[0].z, kb[0].w, kb[1].x, kb[1].y, kb[1].z, kb[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float v = fmaf(raw[r], ATTN_SCALE, bias[r]);
        sv[r] = (kb0 + 16 * (r >> 2) + (r & 3) == self_key) ? -INFINITY : v;  // self_key = -1 never matches
    }
    float tmax = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    tmax = rows_max(tmax);
    new_max = fmaxf(run_max, tmax);
    // branch-free: (-inf) - (-inf) = NaN is clamped inside exp_neg (v_max drops the NaN) and yields 0, which is
    // harmless because the running sum and the accumulators are still 0 in that case
    alpha = exp_neg(run_max - new_max);
}

// One attention head (this wave's) over n_key_pad keys (multiple of 32) with online softmax.
//   q[tt][r]  = Q^T[h*32 + tt*16 + kq*4 + r][agent]  (this wave's Q-projection accumulators, bias added)
//   Kmat      = [n_key_pad][128] row-major, VT = [128][n_key_pad] (keys contiguous), both in global (L2)
//   keybias   = float [n_key_pad]: 0 for valid keys, -inf for invalid / padding keys
//   self_key  = key index that equals THIS LANE's agent (eye mask of MultiAgentTF), or -1
// Returns o[dt][r] = O^T[h*32 + dt*16 + kq*4 + r][agent] (already divided by the softmax sum) and
// whether the agent row had no valid key at all (attention.py:101-107: its output is zeroed after out-proj).
//
// Software pipeline per 32-key block i (one wave per SIMD issues in order, so VALU only overlaps MFMA when it sits
// BETWEEN MFMAs in program order): phase A = QK MFMAs of block i+1 interleaved with the exponentials of block i,
// phase B = PV MFMAs of block i interleaved with scale / mask / running-max of block i+1; K fragments are requested
// two blocks ahead, V fragments one block ahead.
struct AttnPre {  // first fragments of a head's K / V stream, requested before the Q projection so they land under it
    KFrag k0f, kn;
    VFrag vc;
};

__device__ __forceinline__ void attention_prefetch(AttnPre& a, const float* __restrict__ Kmat, const float* __restrict__ VT,
                                                   const float* __restrict__ keybias, int n_key_pad, int head, int lane) {
    const int kq = lane >> 4, m = lane & 15;
    const float* kbase = Kmat + (size_t)m * H + head * DHEAD + kq * 4;
    const float* vbase = VT + (size_t)(head * DHEAD + m) * n_key_pad + kq * 4;
    const float* bbase = keybias + kq * 4;
    TB_SCHED_FENCE();
    k_load(a.k0f, kbase, bbase, 0);
    v_load(a.vc, vbase, n_key_pad, 0);
    k_load(a.kn, kbase, bbase, n_key_pad > 32 ? 32 : 0);
    TB_SCHED_FENCE();
}

// keep a value in a VGPR (MFMA results otherwise live in AGPRs and every VALU touch costs v_accvgpr moves)
__device__ __forceinline__ void in_vgpr(f32x4& v) { asm("" : "+v"(v)); }

__device__ __forceinline__ bool attention_head(const f32x4 (&q)[2], AttnPre& pre, const float* __restrict__ Kmat,
                                               const float* __restrict__ VT, const float* __restrict__ keybias,
                                               int n_key_pad, int head, int lane, int self_key, f32x4 (&o)[2],
                                               long long* prof = nullptr) {
    const int kq = lane >> 4, m = lane & 15;
    o[0] = splat(0.f);
    o[1] = splat(0.f);
    const float* kbase = Kmat + (size_t)m * H + head * DHEAD + kq * 4;
    const float* vbase = VT + (size_t)(head * DHEAD + m) * n_key_pad + kq * 4;
    const float* bbase = keybias + kq * 4;
    KFrag kn = pre.kn;
    VFrag vc = pre.vc;
    float run_max = -INFINITY, run_sum = 0.f, new_max, alpha, sv[8];
    {
        f32x4 s0, s1;
        attn_qk(pre.k0f, q, s0, s1);
        in_vgpr(s0);
        in_vgpr(s1);
        attn_stats(s0, s1, pre.k0f.kb, kq * 4, self_key, run_max, sv, new_max, alpha);
    }
    for (int k0 = 0; k0 < n_key_pad; k0 += 32) {
        const bool has_next = k0 + 32 < n_key_pad;
        const int k2 = (k0 + 64 < n_key_pad) ? k0 + 64 : k0;   // clamped re-reads on the tail are harmless
        const int k1 = has_next ? k0 + 32 : k0;
        // ---------------- phase A: QK(i+1) || exp(i), issue K(i+2) / V(i+1)
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && k0 < 64) prof[25 + (k0 >> 5) * 2] = clock64();
#endif
        TB_SCHED_FENCE();
        f32x4 t0, t1;
        in_vgpr(o[0]);
        in_vgpr(o[1]);
        attn_qk(kn, q, t0, t1);
        f32x4 nb[2] = {kn.kb[0], kn.kb[1]};
        KFrag k2f;
        VFrag v1f;
        k_load(k2f, kbase, bbase, k2);
        v_load(v1f, vbase, n_key_pad, k1);
        float p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = exp_neg(sv[r] - new_max);  // masked keys: -inf (or NaN) -> clamp -> exactly 0
        run_sum = run_sum * alpha + (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])));
        run_max = new_max;
        o[0] *= splat(alpha);
        o[1] *= splat(alpha);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // up to 5 VALU in its shadow
            if (g < 11) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
        }
        TB_SCHED_FENCE();
        in_vgpr(t0);
        in_vgpr(t1);
        in_vgpr(o[0]);
        in_vgpr(o[1]);
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && k0 < 64) prof[26 + (k0 >> 5) * 2] = clock64();
#endif
        // ---------------- phase B: PV(i) || stats(i+1)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o[0] = mfma4(vc.va[t][0].x, p[4 * t + 0], o[0]);
            o[1] = mfma4(vc.va[t][1].x, p[4 * t + 0], o[1]);
            o[0] = mfma4(vc.va[t][0].y, p[4 * t + 1], o[0]);
            o[1] = mfma4(vc.va[t][1].y, p[4 * t + 1], o[1]);
            o[0] = mfma4(vc.va[t][0].z, p[4 * t + 2], o[0]);
            o[1] = mfma4(vc.va[t][1].z, p[4 * t + 2], o[1]);
            o[0] = mfma4(vc.va[t][0].w, p[4 * t + 3], o[0]);
            o[1] = mfma4(vc.va[t][1].w, p[4 * t + 3], o[1]);
        }
        attn_stats(t0, t1, nb, k1 + kq * 4, self_key, run_max, sv, new_max, alpha);  // (unused after the last block)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        TB_SCHED_FENCE();
        kn = k2f;
        vc = v1f;
#ifdef TB_PROFILE
        if (prof && threadIdx.x == 0 && k0 == 32) prof[29] = clock64();
#endif
    }
    run_sum = rows_sum(run_sum);
    const bool novalid = !(run_sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / run_sum;
    o[0] *= splat(inv);
    o[1] *= splat(inv);
    return novalid;
}

#ifdef TB_PROFILE
#define TB_XSTAMP(i) do { if (prof && threadIdx.x == 0) prof[i] = clock64(); } while (0)
#else
#define TB_XSTAMP(i) do { (void)prof; } while (0)
#endif

// Offsets (in floats) of one pre-LN cross-attention layer inside the weight arena.
struct XLayerW {
    uint32_t ln1_g, ln1_b, lnt_g, lnt_b, ln2_g, ln2_b;
    uint32_t wq, bq;    // in_proj rows 0:128      packed K=128, 8 tiles
    uint32_t wkv, bkv;  // in_proj rows 128:384    packed K=128, 16 tiles (K: 0..7, V: 8..15)
    uint32_t wo, bo, w1, b1, w2, b2;
};

struct GruLayerW {
    uint32_t wih, whh, bih, bhh;  // packed K=128, 24 tiles (r: 0..7, z: 8..15, n: 16..23)
};

// ---------------------------------------------------------------------------------------------
// One pre-LN cross-attention layer (transformer.py:189-239 + attention.py:81-146) on the LDS tile X.
//   X    : [16][LDT] residual stream (in/out)         S1, S2 : [16][LDT] scratch
//   Kmat/VT/keybias : projected keys / values (+ additive 0 / -inf key mask) of the tile's group for THIS layer
//   rowvalid : LDS uint8[16]; invalid rows are zeroed at the end.   novalid_s : LDS uint8[16] scratch.
//   self_key0: key index of row 0 for the eye mask, or -1 for none.
//   u : in = this layer's Wq unit (already requested), out = `nxt` requested.
// All 256 threads must call.  Ends with a barrier.
// ---------------------------------------------------------------------------------------------
template <bool LNLDS = false>
__device__ __forceinline__ void xattn_layer(const float* __restrict__ W, const XLayerW& L, float* X, float* S1, float* S2,
                                            const float* __restrict__ Kmat, const float* __restrict__ VT,
                                            const float* __restrict__ keybias, int n_key_pad, int self_key0,
                                            const uint8_t* rowvalid, uint8_t* novalid_s, int tid, WUnit& u, const WNext& nxt,
                                            const float* lnblk = nullptr, long long* prof = nullptr) {
    // LayerNorm parameters: the six 128-vectors [ln1_g, ln1_b, lnt_g, lnt_b, ln2_g, ln2_b] are contiguous in the arena
    // (tb_api.hip add_xlayer); `lnblk` may point at an LDS copy of that block (no VMEM in front of the K/V prefetch)
    if (!LNLDS) lnblk = W + L.ln1_g;
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    TB_XSTAMP(16);
    AttnPre apre;
    attention_prefetch(apre, Kmat, VT, keybias, n_key_pad, wave, lane);
    // s = LN1(x)
    layernorm_tile<LNLDS>(X, LDT, S1, LDT, lnblk, lnblk + 128, tid);
    __syncthreads();
    TB_XSTAMP(17);
    WUnit u2;
    // q (this wave = head `wave`)
    f32x4 q[2] = {u.b[0], u.b[1]};
    wmma_pf(q[0], q[1], u, S1 + m * LDT + kq * 32, u2, wstd(W + L.wo, W + L.bo, wave), lane);
    TB_XSTAMP(18);
    f32x4 o[2];
    const bool novalid = attention_head(q, apre, Kmat, VT, keybias, n_key_pad, wave, lane,
                                        self_key0 >= 0 ? self_key0 + m : -1, o, prof);
    TB_XSTAMP(19);
    st4(cptr(S2, LDT, 2 * wave, lane), o[0]);
    st4(cptr(S2, LDT, 2 * wave + 1, lane), o[1]);
    if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
    __syncthreads();
    TB_XSTAMP(20);
    // out-proj + residual
    {
        f32x4 acc[2] = {u2.b[0], u2.b[1]};
        wmma_pf(acc[0], acc[1], u2, S2 + m * LDT + kq * 32, u, wstd(W + L.w1, W + L.b1, wave), lane);
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
    // FFN
    layernorm_tile<LNLDS>(X, LDT, S1, LDT, lnblk + 512, lnblk + 640, tid);
    __syncthreads();
    TB_XSTAMP(22);
    {
        f32x4 acc[2] = {u.b[0], u.b[1]};
        wmma_pf(acc[0], acc[1], u, S1 + m * LDT + kq * 32, u2, wstd(W + L.w2, W + L.b2, wave), lane);
        st4(cptr(S2, LDT, 2 * wave, lane), relu4(acc[0]));
        st4(cptr(S2, LDT, 2 * wave + 1, lane), relu4(acc[1]));
    }
    __syncthreads();
    TB_XSTAMP(23);
    {
        f32x4 acc[2] = {u2.b[0], u2.b[1]};
        wmma_pf(acc[0], acc[1], u2, S2 + m * LDT + kq * 32, u, nxt, lane);
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

// first unit of a cross-attention layer / of a K-V projection for wave `wave`
__device__ __forceinline__ WNext xlayer_first(const float* W, const XLayerW& L, int wave) { return wstd(W + L.wq, W + L.bq, wave); }
__device__ __forceinline__ WNext kvproj_first(const float* W, const XLayerW& L, int wave) { return wstd(W + L.wkv, W + L.bkv, wave); }

// ---------------------------------------------------------------------------------------------
// K/V projection of a 16-token tile for one layer: LN_tgt -> in_proj rows 128:384.
//   T : [16][LDT] token features (LDS), S1 scratch.  Writes Kmat rows [tok0, tok0+16) and VT columns.
//   Rows >= n_real_rows (padding tokens) are written as zeros.
// Wave w produces K tiles {2w,2w+1} and V tiles {8+2w, 8+2w+1} (head w).
//   u : in = the K unit (tiles 2w, 2w+1 of wkv), out = `nxt`.   Ends with a barrier.
// ---------------------------------------------------------------------------------------------
template <bool LNLDS = false>
__device__ __forceinline__ void kv_project_tile(const float* __restrict__ W, const XLayerW& L, const float* T, float* S1,
                                                float* __restrict__ Kmat, float* __restrict__ VT, int n_key_pad, int tok0,
                                                int n_real_rows, int tid, WUnit& u, const WNext& nxt, const float* lnblk = nullptr) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    if (!LNLDS) lnblk = W + L.ln1_g;
    layernorm_tile<LNLDS>(T, LDT, S1, LDT, lnblk + 256, lnblk + 384, tid);
    __syncthreads();
    const int tiles[4] = {2 * wave, 2 * wave + 1, 8 + 2 * wave, 8 + 2 * wave + 1};
    WUnit u2;
    f32x4 acc[4];
    acc[0] = u.b[0];
    acc[1] = u.b[1];
    const float* xr = S1 + m * LDT + kq * 32;
    wmma_pf(acc[0], acc[1], u, xr, u2, wnext(W + L.wkv, W + L.bkv, tiles[2], tiles[3]), lane);
    acc[2] = u2.b[0];
    acc[3] = u2.b[1];
    wmma_pf(acc[2], acc[3], u2, xr, u, nxt, lane);
    const bool real = m < n_real_rows;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // K row-major: token tok0+m, features tile*16 + kq*4 .. +3
        st4(Kmat + (size_t)(tok0 + m) * H + tiles[t] * 16 + kq * 4, real ? acc[t] : splat(0.f));
        // V transposed: feature f = (tile-8)*16 + kq*4 + r, token column tok0+m
        const int f0 = (tiles[2 + t] - 8) * 16 + kq * 4;
        const f32x4 v = real ? acc[2 + t] : splat(0.f);
        VT[(size_t)(f0 + 0) * n_key_pad + tok0 + m] = v.x;
        VT[(size_t)(f0 + 1) * n_key_pad + tok0 + m] = v.y;
        VT[(size_t)(f0 + 2) * n_key_pad + tok0 + m] = v.z;
        VT[(size_t)(f0 + 3) * n_key_pad + tok0 + m] = v.w;
    }
    __syncthreads();
}

// sigmoid / tanh from one compensated v_exp_f32 of a non-positive argument and one v_rcp_f32 (absolute error ~1e-7)
__device__ __forceinline__ float sigmoidf_(float x) {
    const float e = exp_neg(-fabsf(x));              // in (0, 1]
    const float r = __builtin_amdgcn_rcpf(1.0f + e);  // 1 / (1 + e^-|x|)
    return x >= 0.f ? r : e * r;
}
__device__ __forceinline__ float tanhf_(float x) {
    const float e = exp_neg(-2.0f * fabsf(x));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

// ---------------------------------------------------------------------------------------------
// One GRU layer step (PyTorch gate order r,z,n; agent_temporal.py:147-152 -> nn.GRU):
//   Xin : [16][LDT] layer input, Hs : [16][LDT] previous hidden of this layer, Out : [16][LDT] new hidden
// Wave w owns features [32w, 32w+32) of every gate; six weight units stream through two register buffers.
//   u : in = W_ih r-unit (tiles 2w, 2w+1), out = `nxt`.
// Caller barriers before (inputs ready) -- ends with a barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ WNext gru_first(const float* W, const GruLayerW& G, int wave) { return wstd(W + G.wih, W + G.bih, wave); }

__device__ __forceinline__ void gru_layer(const float* __restrict__ W, const GruLayerW& G, const float* Xin, const float* Hs,
                                          float* Out, const uint8_t* rowvalid, float* __restrict__ h_global /*[rows][128]*/,
                                          int n_real_rows, int tid, WUnit& u, const WNext& nxt) {
    const int wave = wave_of(tid), lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int tr0 = 2 * wave, tz0 = 8 + 2 * wave, tn0 = 16 + 2 * wave;
    const float* xr = Xin + m * LDT + kq * 32;
    const float* hr = Hs + m * LDT + kq * 32;
    WUnit u2;
    f32x4 r[2], z[2], gin[2], ghn[2];
    const float* wih = W + G.wih; const float* whh = W + G.whh; const float* bih = W + G.bih; const float* bhh = W + G.bhh;
    r[0] = u.b[0]; r[1] = u.b[1];
    wmma_pf(r[0], r[1], u, xr, u2, wnext(whh, bhh, tr0, tr0 + 1), lane);
    r[0] += u2.b[0]; r[1] += u2.b[1];
    wmma_pf(r[0], r[1], u2, hr, u, wnext(wih, bih, tz0, tz0 + 1), lane);
    z[0] = u.b[0]; z[1] = u.b[1];
    wmma_pf(z[0], z[1], u, xr, u2, wnext(whh, bhh, tz0, tz0 + 1), lane);
    z[0] += u2.b[0]; z[1] += u2.b[1];
    wmma_pf(z[0], z[1], u2, hr, u, wnext(wih, bih, tn0, tn0 + 1), lane);
    gin[0] = u.b[0]; gin[1] = u.b[1];
    wmma_pf(gin[0], gin[1], u, xr, u2, wnext(whh, bhh, tn0, tn0 + 1), lane);
    ghn[0] = u2.b[0]; ghn[1] = u2.b[1];
    wmma_pf(ghn[0], ghn[1], u2, hr, u, nxt, lane);
    const bool rv = rowvalid[m] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 hold = lds4(Hs + m * LDT + (2 * wave + t) * 16 + kq * 4);
        f32x4 hn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rg = sigmoidf_(r[t][q]);
            const float zg = sigmoidf_(z[t][q]);
            const float ng = tanhf_(gin[t][q] + rg * ghn[t][q]);
            hn[q] = rv ? (1.0f - zg) * ng + zg * hold[q] : 0.f;
        }
        st4(cptr(Out, LDT, 2 * wave + t, lane), hn);
        if (m < n_real_rows) st4(h_global + (size_t)m * H + (2 * wave + t) * 16 + kq * 4, hn);
    }
    __syncthreads();
}

// cooperative copy of a [rows][128] global block into an LDS tile (rows >= n_real -> zeros)
__device__ __forceinline__ void load_tile(float* dst, int ld, const float* __restrict__ src, int n_real, int tid) {
    // 16 rows x 32 float4 = 512 float4, 2 per thread
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        st4(dst + row * ld + c4, row < n_real ? ldg4(src + (size_t)row * H + c4) : splat(0.f));
    }
}

__device__ __forceinline__ void store_tile(float* __restrict__ dst, const float* src, int ld, int n_real, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        if (row < n_real) st4(dst + (size_t)row * H + c4, lds4(src + row * ld + c4));
    }
}

}  // namespace tb
