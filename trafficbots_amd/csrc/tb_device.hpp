// Device building blocks for the gfx950 rollout kernels (wave64, fp32 MFMA 16x16x4).
//
// Geometry used everywhere: one workgroup = 256 threads = 4 waves owns a tile of TM = 16 rows
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
constexpr float LN_EPS = 1e-5f;
constexpr float ATTN_SCALE = 0.17677669529663687f;  // 1/sqrt(32)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// ---------------------------------------------------------------------------------------------
// acc[t] += Wpk(tile t) . X^T      K = reduction length (multiple of 16), NT = tiles this wave owns.
// wpk: packed weight [n_tiles][K/16][64 lanes][4]; tiles[t] = tile index (16 output features each);
// xrow: THIS LANE's pointer to X[agent][kq*(K/4)] in LDS (caller resolves concat buffers).
// ---------------------------------------------------------------------------------------------
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

// bias for the 4 features a lane holds of tile `tile`
__device__ __forceinline__ f32x4 bias4(const float* __restrict__ b, int tile, int lane) {
    return ldg4(b + tile * 16 + (lane >> 4) * 4);
}

// pointer into a row-major LDS tile for the 4 features this lane holds of tile `tile`
__device__ __forceinline__ float* cptr(float* base, int ld, int tile, int lane) {
    return base + (lane & 15) * ld + tile * 16 + (lane >> 4) * 4;
}

// Standard Linear 128 -> 128 of the 16-row tile: wave w owns tiles {2w, 2w+1}.
template <int K>
__device__ __forceinline__ void linear128(f32x4 (&acc)[2], const float* __restrict__ wpk, const float* __restrict__ bias,
                                          const float* xrow, int wave, int lane) {
    const int tiles[2] = {2 * wave, 2 * wave + 1};
    acc[0] = bias4(bias, tiles[0], lane);
    acc[1] = bias4(bias, tiles[1], lane);
    gemm_acc<K, 2>(acc, wpk, tiles, xrow, lane);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm of a [16][128] LDS tile: 16 threads per row, 8 elements each (two float4).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void layernorm_tile(const float* src, int lds_, float* dst, int ldd,
                                               const float* __restrict__ g, const float* __restrict__ b, int tid) {
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    const f32x4 a = lds4(src + row * lds_ + c0), c = lds4(src + row * lds_ + c0 + 4);
    float s = (a.x + a.y) + (a.z + a.w) + (c.x + c.y) + (c.z + c.w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / 128.0f);
    const f32x4 da = a - splat(mean), dc = c - splat(mean);
    float v = (da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) + (dc.z * dc.z + dc.w * dc.w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    const float rstd = 1.0f / sqrtf(v * (1.0f / 128.0f) + LN_EPS);
    const f32x4 g0 = ldg4(g + c0), g1 = ldg4(g + c0 + 4), b0 = ldg4(b + c0), b1 = ldg4(b + c0 + 4);
    st4(dst + row * ldd + c0, da * splat(rstd) * g0 + b0);
    st4(dst + row * ldd + c0 + 4, dc * splat(rstd) * g1 + b1);
}

// ---------------------------------------------------------------------------------------------
// One attention head (this wave's) over n_key_pad keys with online softmax.
//   q[tt][r]  = Q^T[h*32 + tt*16 + kq*4 + r][agent]  (this wave's Q-projection accumulators, bias added)
//   Kmat      = [n_key_pad][128] row-major, VT = [128][n_key_pad] (keys contiguous), both in global (L2)
//   keyvalid  = uint8 [n_key_pad] (0 for padding keys)
//   self_key  = key index that equals THIS LANE's agent (eye mask of MultiAgentTF), or -1
// Returns o[dt][r] = O^T[h*32 + dt*16 + kq*4 + r][agent] (already divided by the softmax sum) and
// whether the agent row had no valid key at all (attention.py:101-107: its output is zeroed after out-proj).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool attention_head(const f32x4 (&q)[2], const float* __restrict__ Kmat,
                                               const float* __restrict__ VT, const uint8_t* __restrict__ keyvalid,
                                               int n_key_pad, int head, int lane, int self_key, f32x4 (&o)[2]) {
    const int kq = lane >> 4, m = lane & 15;
    float run_max = -INFINITY, run_sum = 0.f;
    o[0] = splat(0.f);
    o[1] = splat(0.f);
    const float* kbase = Kmat + (size_t)m * H + head * DHEAD + kq * 4;
    const float* vbase = VT + (size_t)(head * DHEAD + m) * n_key_pad + kq * 4;
    for (int k0 = 0; k0 < n_key_pad; k0 += 16) {
        const f32x4 ka0 = ldg4(kbase + (size_t)k0 * H), ka1 = ldg4(kbase + (size_t)k0 * H + 16);
        const f32x4 va0 = ldg4(vbase + k0), va1 = ldg4(vbase + (size_t)16 * n_key_pad + k0);
        const uint32_t kv4 = *reinterpret_cast<const uint32_t*>(keyvalid + k0 + kq * 4);
        f32x4 s = splat(0.f);
        s = mfma4(ka0.x, q[0].x, s);
        s = mfma4(ka0.y, q[0].y, s);
        s = mfma4(ka0.z, q[0].z, s);
        s = mfma4(ka0.w, q[0].w, s);
        s = mfma4(ka1.x, q[1].x, s);
        s = mfma4(ka1.y, q[1].y, s);
        s = mfma4(ka1.z, q[1].z, s);
        s = mfma4(ka1.w, q[1].w, s);
        // s[r] = logit(key k0 + kq*4 + r, agent m)
        const int kb = k0 + kq * 4;
        float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = ((kv4 >> (8 * r)) & 0xffu) != 0 && (kb + r) != self_key;
            sv[r] = ok ? sv[r] * ATTN_SCALE : -INFINITY;
        }
        float tmax = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float new_max = fmaxf(run_max, tmax);
        const bool dead = (new_max == -INFINITY);
        const float alpha = dead ? 1.0f : expf(run_max - new_max);
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = dead ? 0.f : expf(sv[r] - new_max);
        run_sum = run_sum * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
        run_max = new_max;
        o[0] *= splat(alpha);
        o[1] *= splat(alpha);
        o[0] = mfma4(va0.x, p[0], o[0]);
        o[0] = mfma4(va0.y, p[1], o[0]);
        o[0] = mfma4(va0.z, p[2], o[0]);
        o[0] = mfma4(va0.w, p[3], o[0]);
        o[1] = mfma4(va1.x, p[0], o[1]);
        o[1] = mfma4(va1.y, p[1], o[1]);
        o[1] = mfma4(va1.z, p[2], o[1]);
        o[1] = mfma4(va1.w, p[3], o[1]);
    }
    run_sum += __shfl_xor(run_sum, 16);
    run_sum += __shfl_xor(run_sum, 32);
    const bool novalid = !(run_sum > 0.f);
    const float inv = novalid ? 0.f : 1.0f / run_sum;
    o[0] *= splat(inv);
    o[1] *= splat(inv);
    return novalid;
}

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
//   Kmat/VT/keyvalid : projected keys / values of the tile's group for THIS layer
//   row_invalid(row) comes from rowvalid[] (LDS uint8[16]); invalid rows are zeroed at the end.
//   novalid_s : LDS uint8[16] scratch.   self_key0: key index of row 0 for the eye mask, or -1 for none.
// All 256 threads must call.  Ends with a barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void xattn_layer(const float* __restrict__ W, const XLayerW& L, float* X, float* S1, float* S2,
                                            const float* __restrict__ Kmat, const float* __restrict__ VT,
                                            const uint8_t* __restrict__ keyvalid, int n_key_pad, int self_key0,
                                            const uint8_t* rowvalid, uint8_t* novalid_s, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    // s = LN1(x)
    layernorm_tile(X, LDT, S1, LDT, W + L.ln1_g, W + L.ln1_b, tid);
    __syncthreads();
    // q (this wave = head `wave`)
    f32x4 q[2];
    linear128<128>(q, W + L.wq, W + L.bq, S1 + m * LDT + kq * 32, wave, lane);
    f32x4 o[2];
    const bool novalid = attention_head(q, Kmat, VT, keyvalid, n_key_pad, wave, lane,
                                        self_key0 >= 0 ? self_key0 + m : -1, o);
    st4(cptr(S2, LDT, 2 * wave, lane), o[0]);
    st4(cptr(S2, LDT, 2 * wave + 1, lane), o[1]);
    if (wave == 0 && kq == 0) novalid_s[m] = novalid ? 1 : 0;
    __syncthreads();
    // out-proj + residual
    {
        f32x4 acc[2];
        linear128<128>(acc, W + L.wo, W + L.bo, S2 + m * LDT + kq * 32, wave, lane);
        const bool nv = novalid_s[m] != 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* px = cptr(X, LDT, 2 * wave + t, lane);
            const f32x4 xo = lds4(px);
            st4(px, nv ? xo : xo + acc[t]);
        }
    }
    __syncthreads();
    // FFN
    layernorm_tile(X, LDT, S1, LDT, W + L.ln2_g, W + L.ln2_b, tid);
    __syncthreads();
    {
        f32x4 acc[2];
        linear128<128>(acc, W + L.w1, W + L.b1, S1 + m * LDT + kq * 32, wave, lane);
        st4(cptr(S2, LDT, 2 * wave, lane), relu4(acc[0]));
        st4(cptr(S2, LDT, 2 * wave + 1, lane), relu4(acc[1]));
    }
    __syncthreads();
    {
        f32x4 acc[2];
        linear128<128>(acc, W + L.w2, W + L.b2, S2 + m * LDT + kq * 32, wave, lane);
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

// ---------------------------------------------------------------------------------------------
// K/V projection of a 16-token tile for one layer: LN_tgt -> in_proj rows 128:384.
//   T : [16][LDT] token features (LDS), S1 scratch.  Writes Kmat rows [tok0, tok0+16) and VT columns.
//   Rows >= n_tok_valid_rows (padding tokens) are written as zeros.
// Wave w produces K tiles {2w,2w+1} and V tiles {8+2w, 8+2w+1} (head w).  Ends with a barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kv_project_tile(const float* __restrict__ W, const XLayerW& L, const float* T, float* S1,
                                                float* __restrict__ Kmat, float* __restrict__ VT, int n_key_pad, int tok0,
                                                int n_real_rows, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    layernorm_tile(T, LDT, S1, LDT, W + L.lnt_g, W + L.lnt_b, tid);
    __syncthreads();
    const int tiles[4] = {2 * wave, 2 * wave + 1, 8 + 2 * wave, 8 + 2 * wave + 1};
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = bias4(W + L.bkv, tiles[t], lane);
    gemm_acc<128, 4>(acc, W + L.wkv, tiles, S1 + m * LDT + kq * 32, lane);
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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// One GRU layer step (PyTorch gate order r,z,n; agent_temporal.py:147-152 -> nn.GRU):
//   Xin : [16][LDT] layer input, Hs : [16][LDT] previous hidden of this layer, Out : [16][LDT] new hidden
// Wave w owns features [32w, 32w+32) of every gate.  Caller barriers before (inputs ready) -- ends with a barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void gru_layer(const float* __restrict__ W, const GruLayerW& G, const float* Xin, const float* Hs,
                                          float* Out, const uint8_t* rowvalid, float* __restrict__ h_global /*[rows][128]*/,
                                          int n_real_rows, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, m = lane & 15;
    const int t_rz[4] = {2 * wave, 2 * wave + 1, 8 + 2 * wave, 8 + 2 * wave + 1};
    const int t_n[2] = {16 + 2 * wave, 16 + 2 * wave + 1};
    f32x4 rz[4], gin[2], ghn[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) rz[t] = bias4(W + G.bih, t_rz[t], lane) + bias4(W + G.bhh, t_rz[t], lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        gin[t] = bias4(W + G.bih, t_n[t], lane);
        ghn[t] = bias4(W + G.bhh, t_n[t], lane);
    }
    const float* xr = Xin + m * LDT + kq * 32;
    const float* hr = Hs + m * LDT + kq * 32;
    gemm_acc<128, 4>(rz, W + G.wih, t_rz, xr, lane);
    gemm_acc<128, 4>(rz, W + G.whh, t_rz, hr, lane);
    gemm_acc<128, 2>(gin, W + G.wih, t_n, xr, lane);
    gemm_acc<128, 2>(ghn, W + G.whh, t_n, hr, lane);
    const bool rv = rowvalid[m] != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const f32x4 hold = lds4(Hs + m * LDT + (2 * wave + t) * 16 + kq * 4);
        f32x4 hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = sigmoidf_(rz[t][r]);
            const float zg = sigmoidf_(rz[2 + t][r]);
            const float ng = tanhf(gin[t][r] + rg * ghn[t][r]);
            hn[r] = rv ? (1.0f - zg) * ng + zg * hold[r] : 0.f;
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
