// One-time scene encoders on the GPU (tb_encode_scene):
//   input features  : SceneCentricInput + InputPeEncoder for agent history, TL stop points and map nodes
//   map encoder     : 3-layer transformer over the 20 nodes of each polyline (tgt = original node features),
//                     masked max-pool, 1-layer self-attention over polylines           (map_encoder.py:72-114)
//   personality prior: as2pl / as2tl / interaction_prior over history steps {0,5,10}, 3-step GRU scan, masked
//                     max over time, MLP -> mean                                         (latent_encoder.py:98-147)
//   destination predictor: 11-step GRU scan + residual, last-valid, pairwise MLP over (agent, polyline)
//                                                                                        (goal_manager.py:229-333)
// All of it is composed from the same 16-row-tile device functions as the rollout step kernels.
#include "tb_internal.hpp"

namespace tb {

void launch_kv_hoist_n(const float* W, const XLayerW* L, int n_layer, const float* feat, const uint8_t* fvalid, int G,
                       int n_tok, int n_pad, float* K, float* VT, float* kbias, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// token feature encoder: attr + pose PE -> MLP(attr,32,32) || PE(96)
// ------------------------------------------------------------------------------------------------
struct TokP {
    const float* W;
    EncMlpW mlp;
    uint32_t pe_fxy, pe_fyaw;
    int kind;     // 0 agent history, 1 traffic light, 2 map node
    int n_tok;    // total tokens
    int per_scene;  // tokens per scene (NH*A, NH*T, P*20)
    int inner;      // A, T or 20
    const uint8_t* valid;
    const float* pos;   // [n_tok][2]
    const float* yaw;   // agent: [n_tok]
    const float* dir;   // tl / map: [n_tok][2]
    // agent attributes
    const float* vel; const float* spd; const float* acc; const float* yaw_rate;
    const int32_t* cls;   // agent: type [B][A]; tl: state [n_tok]; map: type [B][P]
    const float* size;    // [B][A][3]
    float* out;           // [n_tok][128]
    // the reference's own inputs (tb_encode_io.ext_*): attributes [n_tok][attr_dim] and pose PE [n_tok][96] taken as given instead of
    // being assembled / evaluated here (both or neither)
    const float* ext_attr;
    const float* ext_pe;
};

// One workgroup encodes TOK_GROUPS consecutive groups of 16 tokens with the MLP weights (transposed) and the PE frequencies staged
// in LDS once: the scalar weight loads of a 16-token workgroup were an L1-latency chain of ~130 dependent loads.
constexpr int TOK_GROUPS = 16;

template <int DIM>
__device__ __forceinline__ void tok_mlp1(const float* __restrict__ a, const float* __restrict__ w, float& s0, float& s1) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
        const float av = a[k];
        s0 = fmaf(av, w[k * 32], s0);
        s1 = fmaf(av, w[k * 32 + 1], s1);
    }
}

__global__ __launch_bounds__(NTHREADS) void k_encode_tokens(TokP p) {
    // attributes / pose / validity of ALL TOK_GROUPS * 16 tokens of the workgroup, gathered by one thread per token in front of the
    // loop: the per-group gather (16 threads, dependent global loads + the fp64 atan2, then a barrier) was a serial round trip per
    // 16 tokens, sixteen times per workgroup.  Row stride 33: the gather writes one row per thread.
    __shared__ float attr[TOK_GROUPS * TM][33];
    __shared__ __attribute__((aligned(16))) float hid[TM][32];
    __shared__ __attribute__((aligned(16))) float outt[TM][LDT];
    __shared__ float w1t[31 * 32], w2t[32 * 32], b1s[32], b2s[32], fxy[12], fyaw[24];
    __shared__ float pose[TOK_GROUPS * TM][4];
    __shared__ uint8_t rvall[TOK_GROUPS * TM];
    const int tid = threadIdx.x;
    const int attr_dim = p.kind == 0 ? 11 : (p.kind == 1 ? 5 : 31);
    for (int i = tid; i < attr_dim * 32; i += NTHREADS) w1t[i] = p.W[p.mlp.w1 + (i & 31) * attr_dim + (i >> 5)];
    for (int i = tid; i < 32 * 32; i += NTHREADS) w2t[i] = p.W[p.mlp.w2 + (i & 31) * 32 + (i >> 5)];
    if (tid < 32) {
        b1s[tid] = p.W[p.mlp.b1 + tid];
        b2s[tid] = p.W[p.mlp.b2 + tid];
    }
    if (tid < 12) fxy[tid] = p.W[p.pe_fxy + tid];
    if (tid >= 32 && tid < 56) fyaw[tid - 32] = p.W[p.pe_fyaw + tid - 32];
    static_assert(TOK_GROUPS * TM == NTHREADS, "one gathered token per thread");
    {
        const int tk = blockIdx.x * TOK_GROUPS * TM + tid;
        float* a = attr[tid];
        for (int k = 0; k < 32; ++k) a[k] = 0.f;
        float x = 0.f, y = 0.f, yw = 0.f;
        uint8_t v = 0;
        if (tk < p.n_tok && p.ext_attr) {
            v = p.valid[tk];
            for (int k = 0; k < attr_dim; ++k) a[k] = p.ext_attr[(size_t)tk * attr_dim + k];
        } else if (tk < p.n_tok) {
            v = p.valid[tk];
            x = p.pos[(size_t)tk * 2];
            y = p.pos[(size_t)tk * 2 + 1];
            const int b = tk / p.per_scene, r = tk % p.inner;
            if (p.kind == 0) {
                yw = p.yaw[tk];
                const int A = p.inner;
                a[0] = p.vel[(size_t)tk * 2]; a[1] = p.vel[(size_t)tk * 2 + 1]; a[2] = p.spd[tk];
                a[3] = p.yaw_rate[tk]; a[4] = p.acc[tk];
                const float* sz = p.size + ((size_t)b * A + r) * 3;
                a[5] = sz[0]; a[6] = sz[1]; a[7] = sz[2];
                const int ty = p.cls[(size_t)b * A + r];
                if (ty >= 0 && ty < 3) a[8 + ty] = 1.f;
            } else {
                yw = (float)atan2((double)p.dir[(size_t)tk * 2 + 1], (double)p.dir[(size_t)tk * 2]);  // pose_pe.py:61
                if (p.kind == 1) {
                    const int st = p.cls[tk];
                    if (st >= 0 && st < 5) a[st] = 1.f;
                } else {
                    const int pl = tk / 20;  // global polyline index
                    const int ty = p.cls[pl];
                    if (ty >= 0 && ty < 11) a[ty] = 1.f;
                    a[11 + r] = 1.f;  // node one-hot (sc_input.py:127-133)
                }
            }
        }
        pose[tid][0] = x; pose[tid][1] = y; pose[tid][2] = yw;
        rvall[tid] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < TOK_GROUPS; ++g) {
        const int tok0 = (blockIdx.x * TOK_GROUPS + g) * TM;
        if (tok0 >= p.n_tok) break;
        const float (*attrg)[33] = attr + g * TM;
        const uint8_t* rv = rvall + g * TM;
        {
            const int row = tid >> 4, i = tid & 15;
            const float px = pose[g * TM + row][0], py = pose[g * TM + row][1], pyaw = pose[g * TM + row][2];
            float* xr = outt[row] + 32;
            if (p.ext_pe) {  // (wave-uniform) the caller's PE: 96 floats per token, 6 per thread
                const int tk = tok0 + row;
#pragma unroll
                for (int u = 0; u < 6; ++u) xr[i * 6 + u] = tk < p.n_tok ? p.ext_pe[(size_t)tk * 96 + i * 6 + u] : 0.f;
            } else
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int j = i * 3 + u;
                float arg;
                int c_cos, c_sin;
                if (j < 12) {
                    arg = px * fxy[j]; c_cos = j; c_sin = 12 + j;
                } else if (j < 24) {
                    arg = py * fxy[j - 12]; c_cos = 24 + (j - 12); c_sin = 36 + (j - 12);
                } else {
                    arg = pyaw * fyaw[j - 24]; c_cos = 48 + (j - 24); c_sin = 72 + (j - 24);
                }
                // fp64 sin/cos of the fp32 argument, rounded once (sincos_pe: Cody-Waite reduction + Taylor, the step kernel's routine)
                float sv, cv;
                sincos_pe(arg, sv, cv);
                xr[c_cos] = cv;
                xr[c_sin] = sv;
            }
            const int o0 = i * 2;
            float s0 = b1s[o0], s1 = b1s[o0 + 1];
            // (compile-time trip count per token kind: with the run-time bound the loop was not unrolled -- two LDS reads, a wait and two
            // dependent FMAs per iteration, ~3 k cycles per 16-token group; unrolled, the reads are batched in front of the FMA chain.
            // Same products in the same order: same bits)
            if (p.kind == 2) tok_mlp1<31>(attrg[row], w1t + o0, s0, s1);
            else if (p.kind == 0) tok_mlp1<11>(attrg[row], w1t + o0, s0, s1);
            else tok_mlp1<5>(attrg[row], w1t + o0, s0, s1);
            hid[row][o0] = fmaxf(s0, 0.f);
            hid[row][o0 + 1] = fmaxf(s1, 0.f);
        }
        __syncthreads();
        {
            const int row = tid >> 4, o0 = (tid & 15) * 2;
            float s0 = b2s[o0], s1 = b2s[o0 + 1];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const float hv = hid[row][k];
                s0 = fmaf(hv, w2t[k * 32 + o0], s0);
                s1 = fmaf(hv, w2t[k * 32 + o0 + 1], s1);
            }
            outt[row][o0] = s0;
            outt[row][o0 + 1] = s1;
        }
        __syncthreads();
        for (int i = tid; i < TM * 32; i += NTHREADS) {
            const int r = i >> 5, c4 = (i & 31) * 4;
            if (tok0 + r < p.n_tok) st4(p.out + (size_t)(tok0 + r) * H + c4, rv[r] ? lds4(&outt[r][c4]) : splat(0.f));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// generic cross-attention block over groups: n_layer x xattn_layer on 16-row tiles
// ------------------------------------------------------------------------------------------------
struct XBlockP {
    const float* W;
    XLayerW L[3];
    int n_layer;
    const float* src;          // [G][n_rows][128]
    const uint8_t* src_valid;  // [G][n_rows]
    float* dst;                // [G][n_rows][128]
    const float* K;            // [G][n_layer][n_pad][128]
    const float* VT;           // [G][n_layer][128][n_pad]
    const float* kbias;        // [G][n_pad] additive key mask
    int n_rows, n_pad;
    int eye;                   // MultiAgentTF: self key masked; groups with exactly one valid row pass through
};

__global__ __launch_bounds__(NTHREADS) void k_xattn_block(XBlockP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* S1 = X + TM * LDT;
    float* S2 = S1 + TM * LDT;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(S2 + TM * LDT);
    uint8_t* novalid_s = rowvalid + 16;
    const int tid = threadIdx.x, g = blockIdx.y, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.n_rows - row0);
    load_tile(X, LDT, p.src + ((size_t)g * p.n_rows + row0) * H, n_real, tid);
    if (tid < TM) rowvalid[tid] = tid < n_real ? p.src_valid[(size_t)g * p.n_rows + row0 + tid] : 0;
    bool bypass = false;
    if (p.eye) {
        const int cnt = __syncthreads_count(tid < p.n_rows && p.src_valid[(size_t)g * p.n_rows + tid]);
        bypass = cnt == 1;
    } else {
        __syncthreads();
    }
    if (!bypass) {
        const int wave = wave_of(tid), lane = tid & 63;
        WUnit u;
        wload(u, xlayer_first(p.W, p.L[0], wave), lane);
#pragma unroll 1
        for (int l = 0; l < p.n_layer; ++l) {
            const WNext nxt = xlayer_first(p.W, p.L[l + 1 < p.n_layer ? l + 1 : l], wave);
            xattn_layer(p.W, p.L[l], X, S1, S2, p.K + ((size_t)g * p.n_layer + l) * p.n_pad * H,
                        p.VT + ((size_t)g * p.n_layer + l) * H * p.n_pad, p.kbias + (size_t)g * p.n_pad, p.n_pad,
                        p.eye ? row0 : -1, rowvalid, novalid_s, tid, u, nxt);
        }
    }
    store_tile(p.dst + ((size_t)g * p.n_rows + row0) * H, X, LDT, n_real, tid);
}

// K/V hoist for 1..3 layers (same as rollout's k_kv_hoist but with a layer count)
__global__ __launch_bounds__(NTHREADS) void k_kv_hoist_n(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, int n_layer,
                                                        const float* __restrict__ feat, const uint8_t* __restrict__ fvalid, int n_tok,
                                                        int n_pad, float* __restrict__ Kout, float* __restrict__ VTout,
                                                        float* __restrict__ kbias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;
    float* S1 = smem + TM * LDT;
    const int tid = threadIdx.x, g = blockIdx.y, tok0 = blockIdx.x * TM;
    const int n_real = max(0, min(TM, n_tok - tok0));
    load_tile(T, LDT, feat + ((size_t)g * n_tok + tok0) * H, n_real, tid);
    if (tid < TM)
        kbias[(size_t)g * n_pad + tok0 + tid] = (tid < n_real && fvalid[(size_t)g * n_tok + tok0 + tid]) ? 0.f : -INFINITY;
    __syncthreads();
    const XLayerW* Ls[3] = {&l0, &l1, &l2};
    const int wave = wave_of(tid), lane = tid & 63;
    WUnit u;
    wload(u, kvproj_first(W, l0, wave), lane);
#pragma unroll 1
    for (int l = 0; l < n_layer; ++l) {
        const WNext nxt = kvproj_first(W, *Ls[l + 1 < n_layer ? l + 1 : l], wave);
        kv_project_tile(W, *Ls[l], T, S1, Kout + ((size_t)g * n_layer + l) * n_pad * H, VTout + ((size_t)g * n_layer + l) * H * n_pad,
                        n_pad, tok0, n_real, tid, u, nxt);
    }
}

void launch_kv_hoist_n(const float* W, const XLayerW* L, int n_layer, const float* feat, const uint8_t* fvalid, int G, int n_tok,
                       int n_pad, float* K, float* VT, float* kbias, hipStream_t s) {
    dim3 grid(n_pad / TM, G);
    hipLaunchKernelGGL(k_kv_hoist_n, grid, dim3(NTHREADS), 2 * TM * LDT * sizeof(float), s, W, L[0], L[n_layer > 1 ? 1 : 0],
                       L[n_layer > 2 ? 2 : 0], n_layer, feat, fvalid, n_tok, n_pad, K, VT, kbias);
}

// ------------------------------------------------------------------------------------------------
// masked max-pool over the nodes of each polyline (map_encoder.py:95-106): one thread per (polyline, 4 features)
// ------------------------------------------------------------------------------------------------
__global__ void k_pool_nodes(const float* __restrict__ x /*[n_pl][20][128]*/, const uint8_t* __restrict__ v /*[n_pl][20]*/,
                             int n_pl, float* __restrict__ out /*[n_pl][128]*/, uint8_t* __restrict__ out_valid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int pl = idx >> 5, c4 = (idx & 31) * 4;
    if (pl >= n_pl) return;
    f32x4 m = splat(-INFINITY);
    bool any = false;
    for (int k = 0; k < 20; ++k) {
        if (!v[(size_t)pl * 20 + k]) continue;
        any = true;
        const f32x4 a = ldg4(x + ((size_t)pl * 20 + k) * H + c4);
        m = f32x4{fmaxf(m.x, a.x), fmaxf(m.y, a.y), fmaxf(m.z, a.z), fmaxf(m.w, a.w)};
    }
    st4(out + (size_t)pl * H + c4, any ? m : splat(0.f));
    if (c4 == 0) out_valid[pl] = any;
}

// gather history steps {0, stride, 2*stride, ..} : dst[b][s'][r][C] = src[b][s'*stride][r][C]
__global__ void k_gather_steps_f(const float* __restrict__ src, float* __restrict__ dst, int B, int S, int S2, int stride, int RC4) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * S2 * RC4;
    if (idx >= total) return;
    const int c = idx % RC4;
    const int s2 = (idx / RC4) % S2;
    const int b = idx / ((size_t)RC4 * S2);
    st4(dst + idx * 4, ldg4(src + (((size_t)b * S + (size_t)s2 * stride) * RC4 + c) * 4));
}
__global__ void k_gather_steps_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int B, int S, int S2, int stride, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * S2 * R;
    if (idx >= total) return;
    const int r = idx % R;
    const int s2 = (idx / R) % S2;
    const int b = idx / ((size_t)R * S2);
    dst[idx] = src[((size_t)b * S + (size_t)s2 * stride) * R + r];
}

// ------------------------------------------------------------------------------------------------
// GRU scan over S steps for 16-agent tiles (agent_temporal.py:133-146) with fused temporal aggregate
//   mode 0: masked max over time (fill -1e3), then latent mean head  (agent_temporal.py:28-29, latent_encoder.py:194-198)
//   mode 1: + residual, last valid step                              (goal_manager.py:295-300, agent_temporal.py:30-33)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_gru_scan(ScanP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                 // step input
    float* Y0 = X + TM * LDT;
    float* Y1 = Y0 + TM * LDT;
    float* HS = Y1 + TM * LDT;       // 3 hidden tiles [3][16][LDT]
    float* AGG = HS + 3 * TM * LDT;  // aggregate
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(AGG + TM * LDT);
    uint8_t* anyvalid = rowvalid + 16;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int b = blockIdx.y, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.A - row0);
    for (int i = tid; i < 3 * TM * LDT; i += NTHREADS) HS[i] = 0.f;
    for (int i = tid; i < TM * LDT; i += NTHREADS) AGG[i] = p.mode == 0 ? -INFINITY : 0.f;
    if (tid < TM) anyvalid[tid] = 0;
    __syncthreads();
    float* scratch_g = nullptr;
    WUnit u;
    wload(u, gru_first(p.W, p.gru[0], wave), lane);
#pragma unroll 1
    for (int s = 0; s < p.S; ++s) {
        load_tile(X, LDT, p.x + (((size_t)b * p.S + s) * p.A + row0) * H, n_real, tid);
        if (tid < TM) {
            const uint8_t v = tid < n_real ? p.valid[((size_t)b * p.S + s) * p.A + row0 + tid] : 0;
            rowvalid[tid] = v;
            if (v) anyvalid[tid] = 1;
        }
        __syncthreads();
        // three layers; each layer's new hidden replaces HS[l] (copy after the layer's barrier)
        float* in = X;
        float* outs[3] = {Y0, Y1, Y0};
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            float* hs = HS + l * TM * LDT;
            // gru_layer writes hidden to a global pointer too; route that to a dummy by n_real_rows = 0
            gru_layer(p.W, p.gru[l], in, hs, outs[l], rowvalid, scratch_g, 0, tid, u, gru_first(p.W, p.gru[(l + 1) % 3], wave));
            for (int i = tid; i < TM * 32; i += NTHREADS) {
                const int r = i >> 5, c4 = (i & 31) * 4;
                st4(hs + r * LDT + c4, lds4(outs[l] + r * LDT + c4));
            }
            __syncthreads();
            in = outs[l];
        }
        // aggregate (outputs of invalid rows are already zero; hidden reset to zero likewise)
        for (int i = tid; i < TM * 32; i += NTHREADS) {
            const int r = i >> 5, c4 = (i & 31) * 4;
            if (p.mode == 0) {
                // x.masked_fill(~valid, -1e3).amax(1)
                const f32x4 o = rowvalid[r] ? lds4(in + r * LDT + c4) : splat(-1e3f);
                const f32x4 a = lds4(AGG + r * LDT + c4);
                st4(AGG + r * LDT + c4, f32x4{fmaxf(a.x, o.x), fmaxf(a.y, o.y), fmaxf(a.z, o.z), fmaxf(a.w, o.w)});
            } else if (rowvalid[r]) {
                st4(AGG + r * LDT + c4, lds4(in + r * LDT + c4) + lds4(X + r * LDT + c4));
            }
        }
        __syncthreads();
    }
    // rows that were never valid -> 0
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        if (!anyvalid[r]) st4(AGG + r * LDT + c4, splat(0.f));
    }
    __syncthreads();
    if (tid < n_real) p.out_valid[(size_t)b * p.A + row0 + tid] = anyvalid[tid];
    if (p.mode == 1) {
        store_tile(p.out_feat + ((size_t)b * p.A + row0) * H, AGG, LDT, n_real, tid);
        return;
    }
    // latent mean = W2 relu(W1 agg + b1) + b2, masked (latent_encoder.py:168-178; MLP mask, mlp.py:80-82)
    {
        f32x4 acc[2];
        linear128<128>(acc, p.W + p.head_w1, p.W + p.head_b1, AGG + m * LDT + kq * 32, wave, lane);
        st4(cptr(Y0, LDT, 2 * wave, lane), relu4(acc[0]));
        st4(cptr(Y0, LDT, 2 * wave + 1, lane), relu4(acc[1]));
    }
    __syncthreads();
    {
        const int r = tid >> 4, o = tid & 15;
        float s = p.W[p.head_b2 + o];
        const float* w2 = p.W + p.head_w2 + o * H;
        for (int k = 0; k < H; ++k) s = fmaf(Y0[r * LDT + k], w2[k], s);
        if (r < n_real) p.out_mean[((size_t)b * p.A + row0 + r) * 16 + o] = anyvalid[r] ? s : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// rows x Linear(128->128) (+ optional bias): out = x W^T + b        grid = ceil(n_rows/16)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_linear_rows(const float* __restrict__ W, uint32_t w, uint32_t bias, int has_bias,
                                                         const float* __restrict__ x, int n_rows, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float X[TM * LDT];
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int row0 = blockIdx.x * TM, n_real = min(TM, n_rows - row0);
    load_tile(X, LDT, x + (size_t)row0 * H, n_real, tid);
    __syncthreads();
    const int tiles[2] = {2 * wave, 2 * wave + 1};
    f32x4 acc[2];
    acc[0] = has_bias ? bias4(W + bias, tiles[0], lane) : splat(0.f);
    acc[1] = has_bias ? bias4(W + bias, tiles[1], lane) : splat(0.f);
    gemm_acc<128, 2>(acc, W + w, tiles, X + m * LDT + kq * 32, lane);
    if (m < n_real) {
        st4(out + (size_t)(row0 + m) * H + tiles[0] * 16 + kq * 4, acc[0]);
        st4(out + (size_t)(row0 + m) * H + tiles[1] * 16 + kq * 4, acc[1]);
    }
}

// ------------------------------------------------------------------------------------------------
// destination logits: for agent a and 16 polylines  (goal_manager.py:235-244,304-307,329-332)
//   y = relu(LN(U[p] + V[a])) ; y = relu(LN(W1 y + b1)) ; logit = w2.y + b2 ; masks
// grid = (p_tiles, A, B)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool dest_candidate(int mtype, bool mvalid, int atype) {
    // map_type_mask (goal_manager.py:235) and the per-class exclusions (:237-244)
    if (!(mvalid && mtype >= 0 && mtype < 5)) return false;
    if (atype == 0 && mtype == 3) return false;
    if (atype == 1 && mtype < 4) return false;
    if (atype == 2 && mtype < 3) return false;
    return true;
}

__global__ __launch_bounds__(NTHREADS) void k_dest_pairs(DestP p) {
    __shared__ __attribute__((aligned(16))) float X[TM * LDT];
    __shared__ __attribute__((aligned(16))) float Y[TM * LDT];
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int p0 = blockIdx.x * TM, a = blockIdx.y, b = blockIdx.z;
    const int n_real = min(TM, p.P - p0);
    const int atype = p.agent_type[(size_t)b * p.A + a];
    const bool dvalid = p.dist_valid[(size_t)b * p.A + a] != 0;
    // does this agent have any candidate polyline at all? (rows that are all -inf become 0, :331-332)
    bool mine = false;
    for (int q = tid; q < p.P; q += NTHREADS)
        mine |= dest_candidate(p.map_type[(size_t)b * p.P + q], p.map_fvalid[(size_t)b * p.P + q] != 0, atype);
    const bool any_cand = __syncthreads_or(mine);
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        f32x4 v = splat(0.f);
        if (r < n_real) v = ldg4(p.U + ((size_t)b * p.P + p0 + r) * H + c4) + ldg4(p.V + ((size_t)b * p.A + a) * H + c4);
        st4(X + r * LDT + c4, v);
    }
    __syncthreads();
    layernorm_tile(X, LDT, Y, LDT, p.W + p.ln0_g, p.W + p.ln0_b, tid);
    __syncthreads();
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        float* q = Y + (i >> 5) * LDT + (i & 31) * 4;
        st4(q, relu4(lds4(q)));
    }
    __syncthreads();
    {
        f32x4 acc[2];
        linear128<128>(acc, p.W + p.w1, p.W + p.b1, Y + m * LDT + kq * 32, wave, lane);
        st4(cptr(X, LDT, 2 * wave, lane), acc[0]);
        st4(cptr(X, LDT, 2 * wave + 1, lane), acc[1]);
    }
    __syncthreads();
    layernorm_tile(X, LDT, Y, LDT, p.W + p.ln1_g, p.W + p.ln1_b, tid);
    __syncthreads();
    {
        // 16 lanes per row: dot(relu(Y[row]), w2)
        const int row = tid >> 4, c0 = (tid & 15) * 8;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s = fmaf(fmaxf(Y[row * LDT + c0 + k], 0.f), p.W[p.w2 + c0 + k], s);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
        if ((tid & 15) == 0 && row < n_real) {
            const int q = p0 + row;
            float lg = s + p.W[p.b2];
            const bool cand = dest_candidate(p.map_type[(size_t)b * p.P + q], p.map_fvalid[(size_t)b * p.P + q] != 0, atype);
            if (!cand) lg = -INFINITY;
            if (!dvalid || !any_cand) lg = 0.f;
            p.logits[((size_t)b * p.A + a) * p.P + q] = lg;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
static void launch_xblock(const XBlockP& p, int G, hipStream_t s) {
    dim3 grid((p.n_rows + TM - 1) / TM, G);
    hipLaunchKernelGGL(k_xattn_block, grid, dim3(NTHREADS), (3 * TM * LDT + 16) * sizeof(float), s, p);
}

// One attention block (K/V hoist of the targets + the source tiles through the layers) with either kernel family.
// Returns true when the block also did the node pooling asked for with pool_out / pool_valid (the fused polyline kernel).
static bool run_block(bool xdl, const float* W, const XLayerW* L, const XLayerX* LX, int n_layer, const float* tgt, const uint8_t* tgt_valid,
                      int G, int n_tgt, int n_pad, float* K, float* VT, float* kbias, const float* src, const uint8_t* src_valid, float* dst,
                      int n_rows, int eye, hipStream_t s, float* pool_out = nullptr, uint8_t* pool_valid = nullptr, int part = 0, int enc_pack = 4) {
    // part: 0 = K / V hoist + block, 1 = the hoist only, 2 = the block only (the hoist needs the targets alone: it may run on another
    // stream before the block's sources exist; the generic hoist + block pair only)
    if (xdl) {
        XBlockPX x{};
        x.W = W; x.n_layer = n_layer;
        for (int l = 0; l < n_layer; ++l) { x.L[l] = L[l]; x.LX[l] = LX[l]; }
        x.src = src; x.src_valid = src_valid; x.dst = dst; x.K = K; x.VT = VT; x.kbias = kbias; x.n_rows = n_rows; x.n_pad = n_pad; x.eye = eye;
        // the map encoder's polyline block (20 nodes, self-attention inside the polyline) runs on the packed tiling: no padding rows
        const int pmode = enc_pack;  // tb_switches.encode_pack resolved by the caller (tb_switches_now)
        const bool pack = pmode != 0;
        // '4' (default): the fused kernel on eight waves with merged phases (k_polyline_fused8<true>); '3': eight waves, the four-wave
        // kernel's phases (k_polyline_fused8<false>); '2': the fused kernel, K / V in LDS, four waves; '1': packed tiling with the K / V
        // hoist through HBM; '0': padded tiling
        if (pmode >= 2 && pool_out && n_tgt == 20 && n_rows == 20 && n_pad == 32 && tgt == src && tgt_valid == src_valid && !eye && G % 2 == 0) {
            x.pool_out = pool_out; x.pool_valid = pool_valid;
            xh::launch_polyline_fused_x(x, G, s, pmode == 4 ? 2 : (pmode == 3 ? 1 : 0));
            return true;
        }
        if (pack && n_tgt == 20 && n_rows == 20 && n_pad == 32 && tgt == src && tgt_valid == src_valid && !eye && G % 4 == 0) {
            xh::launch_polyline_block_x(x, G, K, VT, kbias, s);
            return false;
        }
        if (part != 2) xh::launch_kv_hoist_nx(W, L, LX, n_layer, tgt, tgt_valid, G, n_tgt, n_pad, K, VT, kbias, s);
        if (part != 1) xh::launch_xblock_x(x, G, s);
        return false;
    }
    if (part != 2) launch_kv_hoist_n(W, L, n_layer, tgt, tgt_valid, G, n_tgt, n_pad, K, VT, kbias, s);
    if (part == 1) return false;
    XBlockP x{};
    x.W = W; x.n_layer = n_layer;
    for (int l = 0; l < n_layer; ++l) x.L[l] = L[l];
    x.src = src; x.src_valid = src_valid; x.dst = dst; x.K = K; x.VT = VT; x.kbias = kbias; x.n_rows = n_rows; x.n_pad = n_pad; x.eye = eye;
    launch_xblock(x, G, s);
    return false;
}

// Scratch of one latent-encoder pass (LatentEncoder.forward, latent_encoder.py:98-147) over S3 down-sampled steps.
struct LatentWs {
    float *kpl, *vtpl, *kvpl, *x0, *x1, *x2, *x3, *tl3, *ktl, *vttl, *kvtl, *kin, *vtin, *kvin;
    uint8_t *v0, *tlv3;
};

static void carve_latent(Carver& c, LatentWs& w, size_t B, size_t S3, size_t A, size_t T, size_t a_pad, size_t p_pad, size_t t_pad) {
    w.kpl = c.take<float>(B * 3 * p_pad * 128);
    w.vtpl = c.take<float>(B * 3 * 128 * p_pad);
    w.kvpl = c.take<float>(B * p_pad);
    w.x0 = c.take<float>(B * S3 * A * 128);
    w.v0 = c.take<uint8_t>(B * S3 * A);
    w.x1 = c.take<float>(B * S3 * A * 128);
    w.x2 = c.take<float>(B * S3 * A * 128);
    w.x3 = c.take<float>(B * S3 * A * 128);
    w.tl3 = c.take<float>(B * S3 * T * 128);
    w.tlv3 = c.take<uint8_t>(B * S3 * T);
    w.ktl = c.take<float>(B * S3 * 3 * t_pad * 128);
    w.vttl = c.take<float>(B * S3 * 3 * 128 * t_pad);
    w.kvtl = c.take<float>(B * S3 * t_pad);
    w.kin = c.take<float>(B * S3 * 3 * a_pad * 128);
    w.vtin = c.take<float>(B * S3 * 3 * 128 * a_pad);
    w.kvin = c.take<float>(B * S3 * a_pad);
}

// Prior (which = 0) or posterior (which = 1) personality over the steps {0, 5, 10, ...} of NS encoded steps: agent -> map,
// agent -> traffic lights (shared as2pl / as2tl weights), interaction, GRU over time, max over valid steps, DistEncoder mean.
// The part of it that needs neither the map feature nor anything computed on the main stream -- the gathers of the down-sampled steps
// and the K / V hoist of the traffic-light targets -- is launch_latent_pre: run_encode puts it on the side stream, under the map encoder.
static void launch_latent_pre(bool xdl, const float* W, const EncoderW& ew, int B, int NS, int A, int T, const float* agent_feature,
                              const uint8_t* agent_valid, const float* tl_feature, const uint8_t* tl_valid, const LatentWs& w, hipStream_t s) {
    const int t_pad = padk(T);
    const int S3 = (NS - 1) / 5 + 1;  // latent_encoder.py:98-103
    const int RC4 = A * 32;
    size_t total = (size_t)B * S3 * RC4;
    hipLaunchKernelGGL(k_gather_steps_f, dim3((total + 255) / 256), dim3(256), 0, s, agent_feature, w.x0, B, NS, S3, 5, RC4);
    total = (size_t)B * S3 * A;
    hipLaunchKernelGGL(k_gather_steps_u8, dim3((total + 255) / 256), dim3(256), 0, s, agent_valid, w.v0, B, NS, S3, 5, A);
    total = (size_t)B * S3 * T * 32;
    hipLaunchKernelGGL(k_gather_steps_f, dim3((total + 255) / 256), dim3(256), 0, s, tl_feature, w.tl3, B, NS, S3, 5, T * 32);
    total = (size_t)B * S3 * T;
    hipLaunchKernelGGL(k_gather_steps_u8, dim3((total + 255) / 256), dim3(256), 0, s, tl_valid, w.tlv3, B, NS, S3, 5, T);
    // K / V of the traffic-light targets of agent -> traffic lights (its block runs in launch_latent_branch)
    run_block(xdl, W, ew.as2tl, ew.as2tl_x, 3, w.tl3, w.tlv3, B * S3, T, t_pad, w.ktl, w.vttl, w.kvtl, w.x1, w.v0, w.x2, A, 0, s, nullptr, nullptr, 1);
}

static void launch_latent_branch(bool xdl, const float* W, const EncoderW& ew, int which, int B, int NS, int A, int P, int T,
                                 const float* map_feature, const uint8_t* map_fvalid, const LatentWs& w,
                                 float* out_mean, uint8_t* out_valid, hipStream_t s) {
    const int a_pad = padk(A), p_pad = padk(P), t_pad = padk(T);
    const int S3 = (NS - 1) / 5 + 1;  // latent_encoder.py:98-103
    const XLayerW* inter = which ? ew.inter_post : ew.inter_prior;
    const GruLayerW* gru = which ? ew.gru_post : ew.gru_prior;
    // agent -> map over the S3*A tokens of each scene (shared as2pl weights)
    run_block(xdl, W, ew.as2pl, ew.as2pl_x, 3, map_feature, map_fvalid, B, P, p_pad, w.kpl, w.vtpl, w.kvpl, w.x0, w.v0, w.x1, S3 * A, 0, s);
    // agent -> traffic lights, per step (K / V hoisted by launch_latent_pre)
    run_block(xdl, W, ew.as2tl, ew.as2tl_x, 3, w.tl3, w.tlv3, B * S3, T, t_pad, w.ktl, w.vttl, w.kvtl, w.x1, w.v0, w.x2, A, 0, s, nullptr, nullptr, 2);
    // interaction (own weights), tgt = block input
    run_block(xdl, W, inter, which ? ew.inter_post_x : ew.inter_prior_x, 3, w.x2, w.v0, B * S3, A, a_pad, w.kin, w.vtin, w.kvin, w.x2, w.v0,
              w.x3, A, 1, s);
    ScanP sp{};
    sp.W = W;
    for (int l = 0; l < 3; ++l) sp.gru[l] = gru[l];
    sp.head_w1 = which ? ew.post_w1 : ew.lat_w1; sp.head_b1 = which ? ew.post_b1 : ew.lat_b1;
    sp.head_w2 = which ? ew.post_w2 : ew.lat_w2; sp.head_b2 = which ? ew.post_b2 : ew.lat_b2;
    sp.mode = 0; sp.B = B; sp.S = S3; sp.A = A; sp.x = w.x3; sp.valid = w.v0; sp.out_mean = out_mean; sp.out_valid = out_valid;
    for (int l = 0; l < 3; ++l) sp.grux[l] = (which ? ew.gru_post_x : ew.gru_prior_x)[l];
    if (xdl) xh::launch_gru_scan_x(sp, a_pad, s);
    else hipLaunchKernelGGL(k_gru_scan, dim3(a_pad / TM, B), dim3(NTHREADS), (8 * TM * LDT + 16) * sizeof(float), s, sp);
}

int run_encode(struct ::tb_ctx* ctx, const tb_encode_io* io, hipStream_t s) {
    const int B = io->n_scene, A = io->n_agent, P = io->n_pl, T = io->n_tl, NH = io->n_hist;
    if (B <= 0 || A <= 0 || P <= 0 || T <= 0) return tb_fail(ctx, "tb_encode_scene: empty dimension");
    if (A > 256) return tb_fail(ctx, "tb_encode_scene: n_agent %d > 256 not supported", A);
    if (NH != ctx->cfg.time_step_current + 1 || (NH - 1) % 5 != 0)
        return tb_fail(ctx, "tb_encode_scene: n_hist %d does not match the config", NH);
    const void* req[] = {io->agent_valid, io->agent_type, io->map_valid, io->map_type, io->tl_valid, io->map_feature,
                         io->map_feature_valid, io->agent_feature, io->tl_feature, io->latent_mean, io->latent_valid,
                         io->dest_logits};
    for (const void* q : req)
        if (!q) return tb_fail(ctx, "tb_encode_scene: a required buffer pointer is NULL");
    // raw scene fields of a token kind are needed unless the caller hands over its attributes and pose PE (ext_*)
    if ((io->ext_agent_attr == nullptr) != (io->ext_agent_pe == nullptr) || (io->ext_map_attr == nullptr) != (io->ext_map_pe == nullptr) ||
        (io->ext_tl_attr == nullptr) != (io->ext_tl_pe == nullptr))
        return tb_fail(ctx, "tb_encode_scene: ext_*_attr and ext_*_pe come in pairs");
    if (!io->ext_agent_attr) {
        const void* raw[] = {io->agent_pos, io->agent_yaw, io->agent_vel, io->agent_spd, io->agent_acc, io->agent_yaw_rate, io->agent_size};
        for (const void* q : raw)
            if (!q) return tb_fail(ctx, "tb_encode_scene: a raw agent field is NULL (and no ext_agent_attr / ext_agent_pe given)");
    }
    if (!io->ext_map_attr && (!io->map_pos || !io->map_dir)) return tb_fail(ctx, "tb_encode_scene: map_pos / map_dir NULL (and no ext_map_*)");
    if (!io->ext_tl_attr && (!io->tl_state || !io->tl_pos || !io->tl_dir)) return tb_fail(ctx, "tb_encode_scene: tl_state / tl_pos / tl_dir NULL (and no ext_tl_*)");
    const float* W = ctx->d_arena;
    const EncoderW& ew = ctx->ew;
    const int a_pad = padk(A), p_pad = padk(P), t_pad = padk(T);
    const int S3 = (NH - 1) / 5 + 1;  // steps {0,5,10} (latent_encoder.py:98-103)
    // map-encoder chunking bounds the per-polyline K/V scratch (3 layers x 32 keys x 128 x 2 x 4 B = 96 KiB / polyline)
    // (the fused polyline kernel keeps K / V in LDS: no scratch, one launch over all polylines)
    tb_note_launch(ctx);
    const TbSw sw = tb_switches_now(ctx);
    const bool fused_pl = ctx->encode_kernel == 1 && sw.enc_pack >= 2 && P % 2 == 0;
    const int scenes_per_chunk = fused_pl ? B : std::max(1, std::min(B, (int)(((size_t)512 << 20) / ((size_t)P * 98304))));
    const size_t kv_scenes = fused_pl ? 0 : (size_t)scenes_per_chunk;

    float *nodef, *nodeo, *kn, *vtn, *kvn, *plf, *kps, *vtps, *kvps, *tgt, *U, *V;
    uint8_t* tgtv;
    LatentWs lws;
    auto carve = [&](Carver& c) {
        nodef = c.take<float>((size_t)B * P * 20 * 128);
        nodeo = c.take<float>(kv_scenes * P * 20 * 128);  // (node features after the block: the fused kernel pools them in place)
        kn = c.take<float>(kv_scenes * P * 3 * 32 * 128);
        vtn = c.take<float>(kv_scenes * P * 3 * 128 * 32);
        kvn = c.take<float>(kv_scenes * P * 32);
        plf = c.take<float>((size_t)B * P * 128);
        kps = c.take<float>((size_t)B * p_pad * 128);
        vtps = c.take<float>((size_t)B * 128 * p_pad);
        kvps = c.take<float>((size_t)B * p_pad);
        carve_latent(c, lws, B, S3, A, T, a_pad, p_pad, t_pad);
        tgt = c.take<float>((size_t)B * A * 128);
        tgtv = c.take<uint8_t>((size_t)B * A);
        U = c.take<float>((size_t)B * P * 128);
        V = c.take<float>((size_t)B * A * 128);
    };
    Carver sz{nullptr};
    carve(sz);
    if (tb_ensure_workspace_enc(ctx, sz.off + 256)) return 1;
    Carver c{ctx->d_ws_enc};
    carve(c);

    // ---- what does not need the map runs beside the map encoder: agent / traffic-light tokens and the destination predictor's GRU
    // scan (128 workgroups) go to a side stream that forks from `s` here and joins it in front of the personality branch
    hipStream_t s2 = s;
    {
        if (!sw.enc_side_off) {
            // (the context's ONE private stream, shared with the rollout's graph capture: a further stream per context shifted the
            // runtime's stream -> hardware-queue assignment so that two rollouts on two caller streams no longer overlapped --
            // bench.py's two_batches_in_flight fell from 597 k to 369 k scene-steps/s)
            if (!ctx->cap_stream) TB_HIP(ctx, hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
            if (!ctx->enc_fork) {
                TB_HIP(ctx, hipEventCreateWithFlags(&ctx->enc_fork, hipEventDisableTiming));
                TB_HIP(ctx, hipEventCreateWithFlags(&ctx->enc_join, hipEventDisableTiming));
                TB_HIP(ctx, hipEventCreateWithFlags(&ctx->enc_map, hipEventDisableTiming));
                TB_HIP(ctx, hipEventCreateWithFlags(&ctx->enc_join2, hipEventDisableTiming));
            }
            s2 = ctx->cap_stream;
            TB_HIP(ctx, hipEventRecord(ctx->enc_fork, s));
            TB_HIP(ctx, hipStreamWaitEvent(s2, ctx->enc_fork, 0));
        }
    }
    auto dest_scan = [&](hipStream_t st) {
        ScanP sp{};
        sp.W = W;
        for (int l = 0; l < 3; ++l) sp.gru[l] = ew.gru_dest[l];
        sp.mode = 1; sp.B = B; sp.S = NH; sp.A = A; sp.x = io->agent_feature; sp.valid = io->agent_valid; sp.out_feat = tgt; sp.out_valid = tgtv;
        for (int l = 0; l < 3; ++l) sp.grux[l] = ew.gru_dest_x[l];
        if (ctx->encode_kernel == 1) xh::launch_gru_scan_x(sp, a_pad, st);
        else hipLaunchKernelGGL(k_gru_scan, dim3(a_pad / TM, B), dim3(NTHREADS), (8 * TM * LDT + 16) * sizeof(float), st, sp);
    };
    // ---- input features (sc_input.py:100-140 + input_pe_encoder.py:52-59); the map tokens first: they head the critical path
    {
        TokP m{};
        m.W = W; m.pe_fxy = ew.pe_fxy; m.pe_fyaw = ew.pe_fyaw;
        m.kind = 2; m.mlp = ew.map_enc; m.n_tok = B * P * 20; m.per_scene = P * 20; m.inner = 20;
        m.valid = io->map_valid; m.pos = io->map_pos; m.dir = io->map_dir; m.cls = io->map_type; m.out = nodef;
        m.ext_attr = io->ext_map_attr; m.ext_pe = io->ext_map_pe;
        hipLaunchKernelGGL(k_encode_tokens, dim3((m.n_tok + TM * TOK_GROUPS - 1) / (TM * TOK_GROUPS)), dim3(NTHREADS), 0, s, m);
        TokP t{};
        t.W = W; t.pe_fxy = ew.pe_fxy; t.pe_fyaw = ew.pe_fyaw;
        t.kind = 0; t.mlp = ew.agent_enc; t.n_tok = B * NH * A; t.per_scene = NH * A; t.inner = A;
        t.valid = io->agent_valid; t.pos = io->agent_pos; t.yaw = io->agent_yaw; t.vel = io->agent_vel; t.spd = io->agent_spd;
        t.acc = io->agent_acc; t.yaw_rate = io->agent_yaw_rate; t.cls = io->agent_type; t.size = io->agent_size;
        t.out = io->agent_feature;
        t.ext_attr = io->ext_agent_attr; t.ext_pe = io->ext_agent_pe;
        hipLaunchKernelGGL(k_encode_tokens, dim3((t.n_tok + TM * TOK_GROUPS - 1) / (TM * TOK_GROUPS)), dim3(NTHREADS), 0, s2, t);
        TokP l{};
        l.W = W; l.pe_fxy = ew.pe_fxy; l.pe_fyaw = ew.pe_fyaw;
        l.kind = 1; l.mlp = ew.tl_enc; l.n_tok = B * NH * T; l.per_scene = NH * T; l.inner = T;
        l.valid = io->tl_valid; l.pos = io->tl_pos; l.dir = io->tl_dir; l.cls = io->tl_state; l.out = io->tl_feature;
        l.ext_attr = io->ext_tl_attr; l.ext_pe = io->ext_tl_pe;
        hipLaunchKernelGGL(k_encode_tokens, dim3((l.n_tok + TM * TOK_GROUPS - 1) / (TM * TOK_GROUPS)), dim3(NTHREADS), 0, s2, l);
        if (s2 != s) {
            dest_scan(s2);
            launch_latent_pre(ctx->encode_kernel == 1, W, ew, B, NH, A, T, io->agent_feature, io->agent_valid, io->tl_feature, io->tl_valid, lws, s2);
            TB_HIP(ctx, hipEventRecord(ctx->enc_join, s2));
        }
    }
    // ---- map encoder
    for (int b0 = 0; b0 < B; b0 += scenes_per_chunk) {
        const int nb = std::min(scenes_per_chunk, B - b0);
        const int G = nb * P;
        const float* src = nodef + (size_t)b0 * P * 20 * 128;
        const uint8_t* sv = io->map_valid + (size_t)b0 * P * 20;
        if (run_block(ctx->encode_kernel == 1, W, ew.densetnt, ew.densetnt_x, 3, src, sv, G, 20, 32, kn, vtn, kvn, src, sv, nodeo, 20, 0, s,
                      plf + (size_t)b0 * P * 128, io->map_feature_valid + (size_t)b0 * P, 0, sw.enc_pack))
            continue;
        const int nthr = G * 32;
        hipLaunchKernelGGL(k_pool_nodes, dim3((nthr + 255) / 256), dim3(256), 0, s, nodeo, sv, G, plf + (size_t)b0 * P * 128,
                           io->map_feature_valid + (size_t)b0 * P);
    }
    {
        run_block(ctx->encode_kernel == 1, W, &ew.map_self, &ew.map_self_x, 1, plf, io->map_feature_valid, B, P, p_pad, kps, vtps, kvps, plf,
                  io->map_feature_valid, io->map_feature, P, 0, s);
    }
    // ---- personality prior on `s`, destination predictor beside it on the side stream: both start from the map feature and share
    // nothing else (the predictor's agent half, the GRU scan over the history, ran on the side stream already)
    auto dest_predictor = [&](hipStream_t st) {
        hipLaunchKernelGGL(k_linear_rows, dim3((B * P + TM - 1) / TM), dim3(NTHREADS), 0, st, W, ew.dest_w0_map, ew.dest_b0, 1,
                           io->map_feature, B * P, U);
        hipLaunchKernelGGL(k_linear_rows, dim3((B * A + TM - 1) / TM), dim3(NTHREADS), 0, st, W, ew.dest_w0_agent, 0u, 0, tgt, B * A, V);
        DestP d{};
        d.W = W; d.ln0_g = ew.dest_ln0_g; d.ln0_b = ew.dest_ln0_b; d.w1 = ew.dest_w1; d.b1 = ew.dest_b1; d.ln1_g = ew.dest_ln1_g;
        d.ln1_b = ew.dest_ln1_b; d.w2 = ew.dest_w2; d.b2 = ew.dest_b2;
        d.B = B; d.A = A; d.P = P; d.U = U; d.V = V; d.map_fvalid = io->map_feature_valid; d.map_type = io->map_type;
        d.agent_type = io->agent_type; d.dist_valid = tgtv; d.logits = io->dest_logits;
        d.w1x = ew.dest_w1_x;
        // (beside the latent branch: TWO workgroups per CU instead of three, see launch_dest_pairs_x -- measured: 30 KB of padding
        // -1 % / -4 % of the encode at the headline / stress shape, 60 KB (one per CU) +3 %: the predictor becomes the long pole;
        // TB_DEST_LDS_PAD = bytes, development switch)
        const int lds_pad = st != s ? sw.dest_lds_pad : 0;
        if (ctx->encode_kernel == 1) xh::launch_dest_pairs_x(d, st, lds_pad);
        else hipLaunchKernelGGL(k_dest_pairs, dim3((P + TM - 1) / TM, A, B), dim3(NTHREADS), 0, st, d);
    };
    const bool dest_side = s2 != s && !sw.enc_dest_side_off;
    if (s2 != s) {
        if (dest_side) {
            TB_HIP(ctx, hipEventRecord(ctx->enc_map, s));
            TB_HIP(ctx, hipStreamWaitEvent(s2, ctx->enc_map, 0));
            dest_predictor(s2);
            TB_HIP(ctx, hipEventRecord(ctx->enc_join2, s2));
        }
        TB_HIP(ctx, hipStreamWaitEvent(s, ctx->enc_join, 0));
    } else {
        launch_latent_pre(ctx->encode_kernel == 1, W, ew, B, NH, A, T, io->agent_feature, io->agent_valid, io->tl_feature, io->tl_valid, lws, s);
    }
    launch_latent_branch(ctx->encode_kernel == 1, W, ew, 0, B, NH, A, P, T, io->map_feature, io->map_feature_valid, lws, io->latent_mean,
                         io->latent_valid, s);
    if (dest_side) {
        TB_HIP(ctx, hipStreamWaitEvent(s, ctx->enc_join2, 0));
    } else {
        if (s2 == s) dest_scan(s);
        dest_predictor(s);
    }
    TB_HIP(ctx, hipGetLastError());
    return 0;
}


// Posterior personality of validation_step / training_step (waymo_motion.py:583,597; latent_encoder.py:119-136): the agent and
// traffic-light tokens of ALL ground-truth steps are encoded (sc_latent.py:150-163,196-217), then the latent branch runs with
// the posterior interaction / GRU / head weights over every 5th step.  The map feature is the one tb_encode_scene produced
// (the reference re-encodes the identical map inputs).
int run_encode_posterior(struct ::tb_ctx* ctx, const tb_posterior_io* io, hipStream_t s) {
    const int B = io->n_scene, A = io->n_agent, P = io->n_pl, T = io->n_tl, NS = io->n_step;
    if (B <= 0 || A <= 0 || P <= 0 || T <= 0) return tb_fail(ctx, "tb_encode_posterior: empty dimension");
    if (A > 256) return tb_fail(ctx, "tb_encode_posterior: n_agent %d > 256 not supported", A);
    if (NS < 1 || (NS - 1) % 5 != 0) return tb_fail(ctx, "tb_encode_posterior: (n_step - 1) %% 5 != 0 (n_step %d)", NS);
    const void* req[] = {io->agent_valid, io->agent_pos, io->agent_yaw, io->agent_vel, io->agent_spd, io->agent_acc,
                         io->agent_yaw_rate, io->agent_type, io->agent_size, io->tl_valid, io->tl_state, io->tl_pos, io->tl_dir,
                         io->map_feature, io->map_feature_valid, io->latent_mean, io->latent_valid};
    for (const void* q : req)
        if (!q) return tb_fail(ctx, "tb_encode_posterior: a required buffer pointer is NULL");
    const float* W = ctx->d_arena;
    const EncoderW& ew = ctx->ew;
    const int a_pad = padk(A), p_pad = padk(P), t_pad = padk(T);
    const int S3 = (NS - 1) / 5 + 1;
    float *af, *tf;
    LatentWs lws;
    auto carve = [&](Carver& c) {
        af = c.take<float>((size_t)B * NS * A * 128);
        tf = c.take<float>((size_t)B * NS * T * 128);
        carve_latent(c, lws, B, S3, A, T, a_pad, p_pad, t_pad);
    };
    Carver sz{nullptr};
    carve(sz);
    if (tb_ensure_workspace_enc(ctx, sz.off + 256)) return 1;
    Carver c{ctx->d_ws_enc};
    carve(c);
    TokP t{};
    t.W = W; t.pe_fxy = ew.pe_fxy; t.pe_fyaw = ew.pe_fyaw;
    t.kind = 0; t.mlp = ew.agent_enc; t.n_tok = B * NS * A; t.per_scene = NS * A; t.inner = A;
    t.valid = io->agent_valid; t.pos = io->agent_pos; t.yaw = io->agent_yaw; t.vel = io->agent_vel; t.spd = io->agent_spd;
    t.acc = io->agent_acc; t.yaw_rate = io->agent_yaw_rate; t.cls = io->agent_type; t.size = io->agent_size;
    t.out = af;
    hipLaunchKernelGGL(k_encode_tokens, dim3((t.n_tok + TM * TOK_GROUPS - 1) / (TM * TOK_GROUPS)), dim3(NTHREADS), 0, s, t);
    TokP l{};
    l.W = W; l.pe_fxy = ew.pe_fxy; l.pe_fyaw = ew.pe_fyaw;
    l.kind = 1; l.mlp = ew.tl_enc; l.n_tok = B * NS * T; l.per_scene = NS * T; l.inner = T;
    l.valid = io->tl_valid; l.pos = io->tl_pos; l.dir = io->tl_dir; l.cls = io->tl_state; l.out = tf;
    hipLaunchKernelGGL(k_encode_tokens, dim3((l.n_tok + TM * TOK_GROUPS - 1) / (TM * TOK_GROUPS)), dim3(NTHREADS), 0, s, l);
    launch_latent_pre(ctx->encode_kernel == 1, W, ew, B, NS, A, T, af, io->agent_valid, tf, io->tl_valid, lws, s);
    launch_latent_branch(ctx->encode_kernel == 1, W, ew, 1, B, NS, A, P, T, io->map_feature, io->map_feature_valid, lws, io->latent_mean,
                         io->latent_valid, s);
    TB_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace tb
