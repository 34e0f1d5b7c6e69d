// Scene-encoder attention blocks on the XDL pipe: the fp16-pair twins of k_xattn_block / k_kv_hoist_n (tb_encode_kernels.hip),
// built from the device code of the step kernel (tb_device_xdl.hpp: fp32-accurate GEMMs as three fp16 MFMAs per product, K / V
// in fragment-major fp16 pairs, log2-domain softmax).  Used by the map encoder (densetnt per polyline, polyline self-attention)
// and by the personality prior / posterior (agent -> map, agent -> traffic lights, interaction).
//   k_kv_hoist_nx    grid (n_pad/16, G): LN_tgt + K/V projection of a 16-token tile for 1..3 layers
//   k_xattn_block_x  grid (ceil(n_rows/16), G): a 16-row tile of the source through the 1..3 layers of one block
#include <hip/hip_runtime.h>

#include "tb_rollout.hpp"
// Plane rows of 288 bytes in THIS translation unit (the step kernels keep 272): a wave's ds_read_b128 is served in four groups of sixteen
// lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} ... (MI355X guide, LDS table), and with 272-byte rows every group of the B-operand reads
// holds a 2-way bank conflict; a row stride of 2 (mod 16) 16-byte blocks puts the kq-even lanes of a group on the even blocks and the
// kq-odd lanes on the odd ones.  Worth nothing while the reads were spread through a phase (profiles/r05_experiments.txt item 19); with
// the polyline encoder's reads issued together at the head of each GEMM phase (item 23) the encoders gain 3 %.  LDS layout only: same bits.
#ifndef TB_LDP
#define TB_LDP 144
#endif
#include "tb_device_xdl.hpp"
#include "tb_encode.hpp"

namespace tb {
namespace TB_XNS {

__global__ __launch_bounds__(NTHREADS) void k_kv_hoist_nx(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, XLayerX x0,
                                                         XLayerX x1, XLayerX x2, int n_layer, const float* __restrict__ feat,
                                                         const uint8_t* __restrict__ fvalid, int n_tok, int n_pad,
                                                         float* __restrict__ Kout, float* __restrict__ VTout, float* __restrict__ kbias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;
    xhalf* P1 = reinterpret_cast<xhalf*>(smem + TM * LDT);
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g = blockIdx.y, tok0 = blockIdx.x * TM;
    const int n_real = max(0, min(TM, n_tok - tok0));
    WUnitX u;
    wloadx(u, kvproj_first_x(W, l0, x0, wave), lane);
    load_tile(T, LDT, feat + ((size_t)g * n_tok + tok0) * H, n_real, tid);
    if (tid < TM)
        kbias[(size_t)g * n_pad + tok0 + tid] = (tid < n_real && fvalid[(size_t)g * n_tok + tok0 + tid]) ? 0.f : -INFINITY;
    __syncthreads();
    const size_t ls = (size_t)n_pad * H;  // floats per (group, layer) = fp16 per plane
    xhalf* K0 = reinterpret_cast<xhalf*>(Kout + ((size_t)g * n_layer) * ls);
    xhalf* V0 = reinterpret_cast<xhalf*>(VTout + ((size_t)g * n_layer) * ls);
    kv_project_tile_x(W, l0, x0, T, P1, K0, V0, n_pad, tok0, n_real, tid, u, kvproj_first_x(W, l1, x1, wave));
    if (n_layer > 1) kv_project_tile_x(W, l1, x1, T, P1, K0 + 2 * ls, V0 + 2 * ls, n_pad, tok0, n_real, tid, u, kvproj_first_x(W, l2, x2, wave));
    if (n_layer > 2) kv_project_tile_x(W, l2, x2, T, P1, K0 + 4 * ls, V0 + 4 * ls, n_pad, tok0, n_real, tid, u, kvproj_first_x(W, l2, x2, wave));
}

__global__ void k_kv_hoist_nx2(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, XLayerX x0, XLayerX x1, XLayerX x2, int n_layer,
                               const float* __restrict__ feat, const uint8_t* __restrict__ fvalid, int n_tok, int n_pad,
                               float* __restrict__ Kout, float* __restrict__ VTout, float* __restrict__ kbias);

void launch_kv_hoist_nx(const float* W, const XLayerW* L, const XLayerX* X, int n_layer, const float* feat, const uint8_t* fvalid, int G,
                        int n_tok, int n_pad, float* K, float* VT, float* kbias, hipStream_t s) {
    dim3 grid(n_pad / TM, G);
    const int i1 = n_layer > 1 ? 1 : 0, i2 = n_layer > 2 ? 2 : 0;
    dim3 grid2(n_pad / (2 * TM), G);
    if (n_tok > TM && (int)grid2.x * G >= 256) {  // (n_pad is a multiple of 32: token tiles pair up)
        hipLaunchKernelGGL(k_kv_hoist_nx2, grid2, dim3(NTHREADS), 2 * TM * LDT * sizeof(float) + 2 * PLANES_BYTES, s, W, L[0], L[i1], L[i2],
                           X[0], X[i1], X[i2], n_layer, feat, fvalid, n_tok, n_pad, K, VT, kbias);
        return;
    }
    hipLaunchKernelGGL(k_kv_hoist_nx, grid, dim3(NTHREADS), TM * LDT * sizeof(float) + PLANES_BYTES, s, W, L[0], L[i1], L[i2], X[0], X[i1],
                       X[i2], n_layer, feat, fvalid, n_tok, n_pad, K, VT, kbias);
}

template <bool EYE>
__global__ __launch_bounds__(NTHREADS) void k_xattn_block_x(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    xhalf* PA = reinterpret_cast<xhalf*>(X + TM * LDT);
    xhalf* PB = PA + NPL * PLANE;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(PB + NPL * PLANE);
    uint8_t* novalid_s = rowvalid + 16;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g = blockIdx.y, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.n_rows - row0);
    WUnitX u;
    wloadx(u, xlayer_first_x(p.W, p.L[0], p.LX[0], wave), lane);
    load_tile(X, LDT, p.src + ((size_t)g * p.n_rows + row0) * H, n_real, tid);
    if (tid < TM) rowvalid[tid] = tid < n_real ? p.src_valid[(size_t)g * p.n_rows + row0 + tid] : 0;
    bool bypass = false;
    if (EYE) {  // MultiAgentTF: a scene with exactly one valid agent passes through (agent_interaction.py)
        const int cnt = __syncthreads_count(tid < p.n_rows && p.src_valid[(size_t)g * p.n_rows + tid]);
        bypass = cnt == 1;
    } else {
        __syncthreads();
    }
    if (!bypass) {
        const size_t ls = (size_t)p.n_pad * H;
        const xhalf* K0 = reinterpret_cast<const xhalf*>(p.K + ((size_t)g * p.n_layer) * ls);
        const xhalf* V0 = reinterpret_cast<const xhalf*>(p.VT + ((size_t)g * p.n_layer) * ls);
        const float* kb = p.kbias + (size_t)g * p.n_pad;
#pragma unroll 1
        for (int l = 0; l < p.n_layer; ++l) {
            const int ln = l + 1 < p.n_layer ? l + 1 : l;
            xattn_layer_x<false, EYE>(p.W, p.L[l], p.LX[l], X, PA, PB, K0 + 2 * l * ls, V0 + 2 * l * ls, kb, p.n_pad, 0, EYE ? row0 : -1,
                                      rowvalid, novalid_s, tid, u, xlayer_first_x(p.W, p.L[ln], p.LX[ln], wave));
        }
    }
    store_tile(p.dst + ((size_t)g * p.n_rows + row0) * H, X, LDT, n_real, tid);
}

// Two row tiles per workgroup (xattn_layer_x2): grid (ceil(n_rows/32), G)
template <bool EYE>
__global__ __launch_bounds__(NTHREADS) void k_xattn_block_x2(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X0 = smem;
    float* X1 = X0 + TM * LDT;
    xhalf* PA0 = reinterpret_cast<xhalf*>(X1 + TM * LDT);
    xhalf* PA1 = PA0 + NPL * PLANE;
    xhalf* PB0 = PA1 + NPL * PLANE;
    xhalf* PB1 = PB0 + NPL * PLANE;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(PB1 + NPL * PLANE);  // [32]
    uint8_t* novalid_s = rowvalid + 32;                                 // [32]
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g = blockIdx.y, row0 = blockIdx.x * 2 * TM;
    const int n_real0 = max(0, min(TM, p.n_rows - row0)), n_real1 = max(0, min(TM, p.n_rows - row0 - TM));
    WUnitX u;
    wloadx(u, xlayer_first_x(p.W, p.L[0], p.LX[0], wave), lane);
    load_tile(X0, LDT, p.src + ((size_t)g * p.n_rows + row0) * H, n_real0, tid);
    load_tile(X1, LDT, p.src + ((size_t)g * p.n_rows + row0 + TM) * H, n_real1, tid);
    if (tid < 2 * TM) rowvalid[tid] = row0 + tid < p.n_rows ? p.src_valid[(size_t)g * p.n_rows + row0 + tid] : 0;
    bool bypass = false;
    if (EYE) {
        const int cnt = __syncthreads_count(tid < p.n_rows && p.src_valid[(size_t)g * p.n_rows + tid]);
        bypass = cnt == 1;
    } else {
        __syncthreads();
    }
    if (!bypass) {
        const size_t ls = (size_t)p.n_pad * H;
        const xhalf* K0 = reinterpret_cast<const xhalf*>(p.K + ((size_t)g * p.n_layer) * ls);
        const xhalf* V0 = reinterpret_cast<const xhalf*>(p.VT + ((size_t)g * p.n_layer) * ls);
        const float* kb = p.kbias + (size_t)g * p.n_pad;
#pragma unroll 1
        for (int l = 0; l < p.n_layer; ++l) {
            const int ln = l + 1 < p.n_layer ? l + 1 : l;
            xattn_layer_x2<EYE>(p.W, p.L[l], p.LX[l], X0, X1, PA0, PA1, PB0, PB1, K0 + 2 * l * ls, V0 + 2 * l * ls, kb, p.n_pad,
                                EYE ? row0 : -1, EYE ? row0 + TM : -1, rowvalid, rowvalid + TM, novalid_s, novalid_s + TM, tid, u,
                                xlayer_first_x(p.W, p.L[ln], p.LX[ln], wave));
        }
    }
    store_tile(p.dst + ((size_t)g * p.n_rows + row0) * H, X0, LDT, n_real0, tid);
    store_tile(p.dst + ((size_t)g * p.n_rows + row0 + TM) * H, X1, LDT, n_real1, tid);
}

// K/V hoist, two token tiles per workgroup: grid (n_pad/32, G)
__global__ __launch_bounds__(NTHREADS) void k_kv_hoist_nx2(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, XLayerX x0,
                                                          XLayerX x1, XLayerX x2, int n_layer, const float* __restrict__ feat,
                                                          const uint8_t* __restrict__ fvalid, int n_tok, int n_pad,
                                                          float* __restrict__ Kout, float* __restrict__ VTout, float* __restrict__ kbias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T0 = smem;
    float* T1 = T0 + TM * LDT;
    xhalf* P1a = reinterpret_cast<xhalf*>(T1 + TM * LDT);
    xhalf* P1b = P1a + NPL * PLANE;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g = blockIdx.y, tok0 = blockIdx.x * 2 * TM;
    const int n_real0 = max(0, min(TM, n_tok - tok0)), n_real1 = max(0, min(TM, n_tok - tok0 - TM));
    WUnitX u;
    wloadx(u, kvproj_first_x(W, l0, x0, wave), lane);
    load_tile(T0, LDT, feat + ((size_t)g * n_tok + tok0) * H, n_real0, tid);
    load_tile(T1, LDT, feat + ((size_t)g * n_tok + tok0 + TM) * H, n_real1, tid);
    if (tid < 2 * TM)
        kbias[(size_t)g * n_pad + tok0 + tid] = (tok0 + tid < n_tok && fvalid[(size_t)g * n_tok + tok0 + tid]) ? 0.f : -INFINITY;
    __syncthreads();
    const size_t ls = (size_t)n_pad * H;
    xhalf* K0 = reinterpret_cast<xhalf*>(Kout + ((size_t)g * n_layer) * ls);
    xhalf* V0 = reinterpret_cast<xhalf*>(VTout + ((size_t)g * n_layer) * ls);
    kv_project_tile_x2(W, l0, x0, T0, T1, P1a, P1b, K0, V0, tok0, n_real0, n_real1, tid, u, kvproj_first_x(W, l1, x1, wave));
    if (n_layer > 1)
        kv_project_tile_x2(W, l1, x1, T0, T1, P1a, P1b, K0 + 2 * ls, V0 + 2 * ls, tok0, n_real0, n_real1, tid, u, kvproj_first_x(W, l2, x2, wave));
    if (n_layer > 2)
        kv_project_tile_x2(W, l2, x2, T0, T1, P1a, P1b, K0 + 4 * ls, V0 + 4 * ls, tok0, n_real0, n_real1, tid, u, kvproj_first_x(W, l2, x2, wave));
}

// ---------------------------------------------------------------------------------------------
// The map encoder's polyline block (20 nodes per polyline, 32 key slots, tgt = src) without its padding rows: per four polylines
// four head tiles (nodes 0 .. 15; two polylines per workgroup share every weight unit) and ONE tail tile (nodes 16 .. 19 of all
// four, tb_device_xdl.hpp "packed polyline tails") instead of eight tiles -- 5/8 of the row work.  Bitwise identical to the padded
// tiling (TB_ENCODE_PACK=0).  G % 4 == 0.
//   k_kv_hoist_plh / k_xattn_block_plh   grid (G / 2): head tiles of polylines 2 i, 2 i + 1
//   k_kv_hoist_plt / k_xattn_block_plt   grid (G / 4): tail tile of polylines 4 i .. 4 i + 3
// ---------------------------------------------------------------------------------------------
constexpr int PL_NODES = 20;

__global__ __launch_bounds__(NTHREADS) void k_kv_hoist_plh(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, XLayerX x0,
                                                          XLayerX x1, XLayerX x2, int n_layer, const float* __restrict__ feat,
                                                          const uint8_t* __restrict__ fvalid, float* __restrict__ Kout,
                                                          float* __restrict__ VTout, float* __restrict__ kbias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T0 = smem;
    float* T1 = T0 + TM * LDT;
    xhalf* P1a = reinterpret_cast<xhalf*>(T1 + TM * LDT);
    xhalf* P1b = P1a + NPL * PLANE;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g0 = blockIdx.x * 2, g1 = g0 + 1;
    WUnitX u;
    wloadx(u, kvproj_first_x(W, l0, x0, wave), lane);
    load_tile(T0, LDT, feat + (size_t)g0 * PL_NODES * H, TM, tid);
    load_tile(T1, LDT, feat + (size_t)g1 * PL_NODES * H, TM, tid);
    if (tid < 2 * TM) {
        const int g = g0 + (tid >> 4), k = tid & 15;
        kbias[(size_t)g * KEYPAD + k] = fvalid[(size_t)g * PL_NODES + k] ? 0.f : -INFINITY;
    }
    __syncthreads();
    const size_t ls = (size_t)KEYPAD * H;
    xhalf* K0 = reinterpret_cast<xhalf*>(Kout + ((size_t)g0 * n_layer) * ls);
    xhalf* V0 = reinterpret_cast<xhalf*>(VTout + ((size_t)g0 * n_layer) * ls);
    xhalf* K1 = reinterpret_cast<xhalf*>(Kout + ((size_t)g1 * n_layer) * ls);
    xhalf* V1 = reinterpret_cast<xhalf*>(VTout + ((size_t)g1 * n_layer) * ls);
    kv_project_tile_x2(W, l0, x0, T0, T1, P1a, P1b, K0, V0, 0, TM, TM, tid, u, kvproj_first_x(W, l1, x1, wave), K1, V1);
    if (n_layer > 1)
        kv_project_tile_x2(W, l1, x1, T0, T1, P1a, P1b, K0 + 2 * ls, V0 + 2 * ls, 0, TM, TM, tid, u, kvproj_first_x(W, l2, x2, wave), K1 + 2 * ls,
                           V1 + 2 * ls);
    if (n_layer > 2)
        kv_project_tile_x2(W, l2, x2, T0, T1, P1a, P1b, K0 + 4 * ls, V0 + 4 * ls, 0, TM, TM, tid, u, kvproj_first_x(W, l2, x2, wave), K1 + 4 * ls,
                           V1 + 4 * ls);
}

// rows of a packed tail tile: row r <-> (polyline gq + (r >> 2), node 16 + (r & 3))
__device__ __forceinline__ void load_tail_tile(float* dst, const float* __restrict__ src, int gq, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        st4(dst + row * LDT + c4, ldg4(src + ((size_t)(gq + (row >> 2)) * PL_NODES + 16 + (row & 3)) * H + c4));
    }
}
__device__ __forceinline__ void store_tail_tile(float* __restrict__ dst, const float* src, int gq, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * NTHREADS;
        const int row = idx >> 5, c4 = (idx & 31) * 4;
        st4(dst + ((size_t)(gq + (row >> 2)) * PL_NODES + 16 + (row & 3)) * H + c4, lds4(src + row * LDT + c4));
    }
}

__global__ __launch_bounds__(NTHREADS) void k_kv_hoist_plt(const float* __restrict__ W, XLayerW l0, XLayerW l1, XLayerW l2, XLayerX x0,
                                                          XLayerX x1, XLayerX x2, int n_layer, const float* __restrict__ feat,
                                                          const uint8_t* __restrict__ fvalid, float* __restrict__ Kout,
                                                          float* __restrict__ VTout, float* __restrict__ kbias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* T = smem;
    xhalf* P1 = reinterpret_cast<xhalf*>(smem + TM * LDT);
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, gq = blockIdx.x * 4;
    WUnitX u;
    wloadx(u, kvproj_first_x(W, l0, x0, wave), lane);
    load_tail_tile(T, feat, gq, tid);
    if (tid < 4 * TM) {  // key slots 16 .. 31 of the four polylines: nodes 16 .. 19, then padding
        const int g = gq + (tid >> 4), k = tid & 15;
        kbias[(size_t)g * KEYPAD + 16 + k] = (k < 4 && fvalid[(size_t)g * PL_NODES + 16 + k]) ? 0.f : -INFINITY;
    }
    __syncthreads();
    const size_t ls = (size_t)KEYPAD * H;
    const size_t gstride = 2 * (size_t)n_layer * ls;
    xhalf* K0 = reinterpret_cast<xhalf*>(Kout + ((size_t)gq * n_layer) * ls);
    xhalf* V0 = reinterpret_cast<xhalf*>(VTout + ((size_t)gq * n_layer) * ls);
    kv_project_tile_xt(W, l0, x0, T, P1, K0, V0, gstride, tid, u, kvproj_first_x(W, l1, x1, wave));
    if (n_layer > 1) kv_project_tile_xt(W, l1, x1, T, P1, K0 + 2 * ls, V0 + 2 * ls, gstride, tid, u, kvproj_first_x(W, l2, x2, wave));
    if (n_layer > 2) kv_project_tile_xt(W, l2, x2, T, P1, K0 + 4 * ls, V0 + 4 * ls, gstride, tid, u, kvproj_first_x(W, l2, x2, wave));
}

__global__ __launch_bounds__(NTHREADS) void k_xattn_block_plh(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X0 = smem;
    float* X1 = X0 + TM * LDT;
    xhalf* PA0 = reinterpret_cast<xhalf*>(X1 + TM * LDT);
    xhalf* PA1 = PA0 + NPL * PLANE;
    xhalf* PB0 = PA1 + NPL * PLANE;
    xhalf* PB1 = PB0 + NPL * PLANE;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(PB1 + NPL * PLANE);  // [32]
    uint8_t* novalid_s = rowvalid + 32;                                 // [32]
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, g0 = blockIdx.x * 2, g1 = g0 + 1;
    WUnitX u;
    wloadx(u, xlayer_first_x(p.W, p.L[0], p.LX[0], wave), lane);
    load_tile(X0, LDT, p.src + (size_t)g0 * PL_NODES * H, TM, tid);
    load_tile(X1, LDT, p.src + (size_t)g1 * PL_NODES * H, TM, tid);
    if (tid < 2 * TM) rowvalid[tid] = p.src_valid[(size_t)(g0 + (tid >> 4)) * PL_NODES + (tid & 15)];
    __syncthreads();
    const size_t ls = (size_t)KEYPAD * H;
    const xhalf* K0 = reinterpret_cast<const xhalf*>(p.K + ((size_t)g0 * p.n_layer) * ls);
    const xhalf* V0 = reinterpret_cast<const xhalf*>(p.VT + ((size_t)g0 * p.n_layer) * ls);
    const xhalf* K1 = reinterpret_cast<const xhalf*>(p.K + ((size_t)g1 * p.n_layer) * ls);
    const xhalf* V1 = reinterpret_cast<const xhalf*>(p.VT + ((size_t)g1 * p.n_layer) * ls);
    const float* kb0 = p.kbias + (size_t)g0 * KEYPAD;
    const float* kb1 = p.kbias + (size_t)g1 * KEYPAD;
#pragma unroll 1
    for (int l = 0; l < p.n_layer; ++l) {
        const int ln = l + 1 < p.n_layer ? l + 1 : l;
        xattn_layer_x2<false>(p.W, p.L[l], p.LX[l], X0, X1, PA0, PA1, PB0, PB1, K0 + 2 * l * ls, V0 + 2 * l * ls, kb0, KEYPAD, -1, -1, rowvalid,
                              rowvalid + TM, novalid_s, novalid_s + TM, tid, u, xlayer_first_x(p.W, p.L[ln], p.LX[ln], wave), K1 + 2 * l * ls,
                              V1 + 2 * l * ls, kb1);
    }
    store_tile(p.dst + (size_t)g0 * PL_NODES * H, X0, LDT, TM, tid);
    store_tile(p.dst + (size_t)g1 * PL_NODES * H, X1, LDT, TM, tid);
}

__global__ __launch_bounds__(NTHREADS) void k_xattn_block_plt(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    xhalf* PA = reinterpret_cast<xhalf*>(X + TM * LDT);
    xhalf* PB = PA + NPL * PLANE;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(PB + NPL * PLANE);
    uint8_t* novalid_s = rowvalid + 16;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, gq = blockIdx.x * 4;
    WUnitX u;
    wloadx(u, xlayer_first_x(p.W, p.L[0], p.LX[0], wave), lane);
    load_tail_tile(X, p.src, gq, tid);
    if (tid < TM) rowvalid[tid] = p.src_valid[(size_t)(gq + (tid >> 2)) * PL_NODES + 16 + (tid & 3)];
    __syncthreads();
    const size_t ls = (size_t)KEYPAD * H;
    const size_t gstride = 2 * (size_t)p.n_layer * ls;
    const xhalf* K0 = reinterpret_cast<const xhalf*>(p.K + ((size_t)gq * p.n_layer) * ls);
    const xhalf* V0 = reinterpret_cast<const xhalf*>(p.VT + ((size_t)gq * p.n_layer) * ls);
    const float* kb = p.kbias + (size_t)gq * KEYPAD;
#pragma unroll 1
    for (int l = 0; l < p.n_layer; ++l) {
        const int ln = l + 1 < p.n_layer ? l + 1 : l;
        xattn_layer_xt(p.W, p.L[l], p.LX[l], X, PA, PB, K0 + 2 * l * ls, V0 + 2 * l * ls, kb, gstride, rowvalid, novalid_s, tid, u,
                       xlayer_first_x(p.W, p.L[ln], p.LX[ln], wave));
    }
    store_tail_tile(p.dst, X, gq, tid);
}

// ---------------------------------------------------------------------------------------------
// Polyline block, FUSED: one workgroup = two polylines = three row tiles (head tile of each + one tile with the four tail nodes of
// both), and the K / V of their 2 x 20 keys never leave the CU -- each layer projects them from the block input (tgt is fixed over the
// layers: transformer_block.py) straight into two 32-key blocks in LDS, in the fragment-major layout the attention reads (the same
// device functions, handed LDS pointers).  Against hoist + block kernels: no 96 KiB per polyline of K / V written to HBM and read
// back (805 MB each way per 32 scenes), every weight unit applied to three tiles.  Same arithmetic per row: bitwise identical.
// LDS: 3 fp32 tiles + 2 x 3 plane sets (the second doubles as the fp32 staging of the block input for LN_tgt) + 64 KB of K / V.
// grid (G / 2); G % 2 == 0.
// ---------------------------------------------------------------------------------------------
constexpr int PLF_LDS_BYTES = 3 * TM * LDT * 4 + 6 * PLANES_BYTES + 4 * KV_BLOCK_HALFS * 2 + 64 * 4 + 128;
static_assert(3 * TM * LDT * 4 <= 3 * PLANES_BYTES || NPL == 1, "the fp32 staging of the block input lives in the second plane sets");

__global__ __launch_bounds__(NTHREADS) void k_polyline_fused(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                                   // [3][16][LDT]
    xhalf* P1 = reinterpret_cast<xhalf*>(X + 3 * TM * LDT);            // [3] plane sets
    xhalf* P2 = P1 + 3 * NPL * PLANE;                                  // [3] plane sets
    // (bf16 build: one plane per set, the staging needs its own room behind the K / V blocks -- see the launch)
    xhalf* KL = P2 + 3 * NPL * PLANE;                                  // K blocks of the two polylines
    xhalf* VL = KL + 2 * KV_BLOCK_HALFS;                               // V blocks
    float* kb = reinterpret_cast<float*>(VL + 2 * KV_BLOCK_HALFS);     // [2][32] additive key mask
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(kb + 64);           // [48]
    uint8_t* novalid_s = rowvalid + 48;                                // [48]
    float* S = NPL == 2 ? reinterpret_cast<float*>(P2) : reinterpret_cast<float*>(novalid_s + 80);  // fp32 staging [3][16][LDT]
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int g0 = blockIdx.x * 2;
    const int po = m * LDP + kq * 8;
    constexpr int PS = NPL * PLANE;  // fp16 per plane set

    // rows of the three tiles: tile 0 / 1 = nodes 0 .. 15 of polyline g0 / g0 + 1, tile 2 rows 0 .. 3 / 4 .. 7 = their nodes 16 .. 19
    auto load3 = [&](float* D) {
        load_tile(D, LDT, p.src + (size_t)g0 * PL_NODES * H, TM, tid);
        load_tile(D + TM * LDT, LDT, p.src + (size_t)(g0 + 1) * PL_NODES * H, TM, tid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NTHREADS;
            const int row = idx >> 5, c4 = (idx & 31) * 4;
            st4(D + 2 * TM * LDT + row * LDT + c4,
                row < 8 ? ldg4(p.src + ((size_t)(g0 + (row >> 2)) * PL_NODES + 16 + (row & 3)) * H + c4) : splat(0.f));
        }
    };
    WUnitX u, u2;
    wloadx(u, kvproj_first_x(p.W, p.L[0], p.LX[0], wave), lane);
    load3(X);
    if (tid < 48) {
        const int t = tid >> 4, r = tid & 15;
        rowvalid[tid] = t < 2 ? p.src_valid[(size_t)(g0 + t) * PL_NODES + r] : (r < 8 ? p.src_valid[(size_t)(g0 + (r >> 2)) * PL_NODES + 16 + (r & 3)] : 0);
    } else if (tid >= 64 && tid < 128) {
        const int pl = (tid - 64) >> 5, k = (tid - 64) & 31;
        kb[tid - 64] = (k < PL_NODES && p.src_valid[(size_t)(g0 + pl) * PL_NODES + k]) ? 0.f : -INFINITY;
    }
    {   // key slots 20 .. 31 stay zero for the whole kernel (masked keys must hold finite data)
        const xh8 z = {};
        xh8* kv8 = reinterpret_cast<xh8*>(KL);
        for (int i = tid; i < 4 * KV_BLOCK_HALFS / 8; i += NTHREADS) kv8[i] = z;
    }
    __syncthreads();

    auto gemm3 = [&](WUnitX& ua, WUnitX& ub, const WNextX& next, const xhalf* P, f32x4 (&a)[3][2]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { a[t][0] = ua.b[0]; a[t][1] = ua.b[1]; }
        wmmax_pf(a[0][0], a[0][1], ua, P + po, PLANE, ub, next, lane);
        wmmax(a[1][0], a[1][1], ua, P + PS + po, PLANE);
        wmmax(a[2][0], a[2][1], ua, P + 2 * PS + po, PLANE);
    };

#pragma unroll 1
    for (int l = 0; l < p.n_layer; ++l) {
        const XLayerW& L = p.L[l];
        const XLayerX& LX = p.LX[l];
        const float* lnblk = p.W + L.ln1_g;
        // ---- K / V of the layer from the block input
        if (l > 0) {  // (layer 0: X still holds the block input)
            load3(S);
            __syncthreads();
        }
        layernorm_planes_n<3>(l > 0 ? S : X, TM * LDT, P1, PS, lnblk + 256, lnblk + 384, tid);
        __syncthreads();
        {
            f32x4 ak[3][2], av[3][2];
            gemm3(u, u2, wnextx(p.W, LX.wkv, p.W + L.bkv, 8 + 2 * wave, 8 + 2 * wave + 1), P1, ak);
            // K: lane (kq, m) = token m, features 4 kq + r of the head's two feature tiles -> its 8-byte half of the key's K granule
            auto k_store = [&](xhalf* kblk, int kt, int krow, const f32x4 (&a)[2]) {
                xhalf* pk = kblk + wave * (NPL * 1024) + (kt * 64 + kq * 16 + krow) * 8;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xh4 h, lo;
                    split2(a[t], h, lo);
                    *reinterpret_cast<xh4*>(pk + t * 4) = h;
                    if (NPL == 2) *reinterpret_cast<xh4*>(pk + t * 4 + 1024) = lo;
                }
            };
            k_store(KL, 0, m, ak[0]);
            k_store(KL + KV_BLOCK_HALFS, 0, m, ak[1]);
            if (m < 8) k_store(KL + (m >> 2) * KV_BLOCK_HALFS, 1, m & 3, ak[2]);
            // V with the MFMA operands swapped: lane (kq, m) holds tokens 4 kq + r of feature m of the head's two d tiles -- the element
            // order of a V fragment half, one 8-byte store each (the token-major accumulators needed four 2-byte scatters per half, with
            // 8-way LDS bank conflicts).  The bias enters as the initial accumulator, as in the other layout: same bits.
            {
                const float* bv = p.W + L.bkv + (8 + 2 * wave) * 16 + m;
                const f32x4 ba = splat(bv[0]), bb = splat(bv[16]);
#pragma unroll
                for (int t = 0; t < 3; ++t) { av[t][0] = ba; av[t][1] = bb; }
                wmmax_pf<true>(av[0][0], av[0][1], u2, P1 + po, PLANE, u, xlayer_first_x(p.W, L, LX, wave), lane);
                wmmax<true>(av[1][0], av[1][1], u2, P1 + PS + po, PLANE);
                wmmax<true>(av[2][0], av[2][1], u2, P1 + 2 * PS + po, PLANE);
                auto v_store = [&](xhalf* vblk, int half, const f32x4 (&a)[2]) {
                    xhalf* pv = vblk + wave * (NPL * 1024) + (kq * 16 + m) * 8 + half * 4;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        xh4 h, lo;
                        split2(a[dt], h, lo);
                        *reinterpret_cast<xh4*>(pv + dt * 512) = h;
                        if (NPL == 2) *reinterpret_cast<xh4*>(pv + dt * 512 + 1024) = lo;
                    }
                };
                v_store(VL, 0, av[0]);
                v_store(VL + KV_BLOCK_HALFS, 0, av[1]);
                // tail tile: lane group kq = 0 / 1 holds nodes 16 .. 19 of polyline 0 / 1 = keys 16 .. 19 = elements 4 .. 7 of key quad 0
                if (kq < 2) {
                    xhalf* pv = VL + kq * KV_BLOCK_HALFS + wave * (NPL * 1024) + m * 8 + 4;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        xh4 h, lo;
                        split2(av[2][dt], h, lo);
                        *reinterpret_cast<xh4*>(pv + dt * 512) = h;
                        if (NPL == 2) *reinterpret_cast<xh4*>(pv + dt * 512 + 1024) = lo;
                    }
                }
            }
        }
        __syncthreads();  // (the projections have read P1: LayerNorm 1 may overwrite it; K / V are in place)
        // ---- the layer on the three tiles
        layernorm_planes_n<3>(X, TM * LDT, P1, PS, lnblk, lnblk + 128, tid);
        __syncthreads();
        {
            f32x4 q[3][2];
            gemm3(u, u2, wstdx(p.W, LX.wo, p.W + L.bo, wave), P1, q);
            // four one-block attentions (tile 0 / polyline 0, tile 1 / polyline 1, the tail tile against both), written without anything
            // that pins the schedule so that their latency chains overlap
            f32x4 o0[2], o1[2], oa[2], ob[2];
            const bool n0 = attention_oneblock_x(q[0], KL, VL, kb, wave, lane, o0);
            const bool n1 = attention_oneblock_x(q[1], KL + KV_BLOCK_HALFS, VL + KV_BLOCK_HALFS, kb + KEYPAD, wave, lane, o1);
            const bool na = attention_oneblock_x(q[2], KL, VL, kb, wave, lane, oa);
            const bool nb = attention_oneblock_x(q[2], KL + KV_BLOCK_HALFS, VL + KV_BLOCK_HALFS, kb + KEYPAD, wave, lane, ob);
            planes_store_c<false>(P2, 2 * wave, lane, o0[0]);
            planes_store_c<false>(P2, 2 * wave + 1, lane, o0[1]);
            planes_store_c<false>(P2 + PS, 2 * wave, lane, o1[0]);
            planes_store_c<false>(P2 + PS, 2 * wave + 1, lane, o1[1]);
            const int grp = m >> 2;  // tail rows 0 .. 3 / 4 .. 7 belong to polyline 0 / 1, rows 8 .. 15 are padding
            const f32x4 z = splat(0.f);
            planes_store_c<false>(P2 + 2 * PS, 2 * wave, lane, grp == 0 ? oa[0] : (grp == 1 ? ob[0] : z));
            planes_store_c<false>(P2 + 2 * PS, 2 * wave + 1, lane, grp == 0 ? oa[1] : (grp == 1 ? ob[1] : z));
            if (wave == 0 && kq == 0) {
                novalid_s[m] = n0 ? 1 : 0;
                novalid_s[TM + m] = n1 ? 1 : 0;
                novalid_s[2 * TM + m] = (grp == 0 ? na : (grp == 1 ? nb : true)) ? 1 : 0;
            }
        }
        __syncthreads();
        {
            f32x4 a[3][2];
            gemm3(u2, u, wstdx(p.W, LX.w1, p.W + L.b1, wave), P2, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool nv = novalid_s[t * TM + m] != 0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float* px = cptr(X + t * TM * LDT, LDT, 2 * wave + c, lane);
                    const f32x4 xo = lds4(px);
                    st4(px, nv ? xo : xo + a[t][c]);
                }
            }
        }
        __syncthreads();
        layernorm_planes_n<3>(X, TM * LDT, P1, PS, lnblk + 512, lnblk + 640, tid);
        __syncthreads();
        {
            f32x4 a[3][2];
            gemm3(u, u2, wstdx(p.W, LX.w2, p.W + L.b2, wave), P1, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                planes_store_c(P2 + t * PS, 2 * wave, lane, relu4(a[t][0]));
                planes_store_c(P2 + t * PS, 2 * wave + 1, lane, relu4(a[t][1]));
            }
        }
        __syncthreads();
        {
            f32x4 a[3][2];
            const int ln = l + 1 < p.n_layer ? l + 1 : l;
            gemm3(u2, u, kvproj_first_x(p.W, p.L[ln], p.LX[ln], wave), P2, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool rv = rowvalid[t * TM + m] != 0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float* px = cptr(X + t * TM * LDT, LDT, 2 * wave + c, lane);
                    st4(px, rv ? lds4(px) + a[t][c] : splat(0.f));
                }
            }
        }
        __syncthreads();
    }
    {   // MapEncoder: max over the valid nodes of the polyline (k_pool_nodes), one (polyline, feature) per thread; the node features
        // themselves have no other consumer and are not written out
        const int pl = tid >> 7, f = tid & 127;
        float mx = -INFINITY;
        bool any = false;
#pragma unroll
        for (int k = 0; k < PL_NODES; ++k) {
            const int t = k < TM ? pl : 2, r = k < TM ? k : pl * 4 + (k - TM);
            if (rowvalid[t * TM + r]) {
                any = true;
                mx = fmaxf(mx, X[(t * TM + r) * LDT + f]);
            }
        }
        p.pool_out[(size_t)(g0 + pl) * H + f] = any ? mx : 0.f;
        if (f == 0) p.pool_valid[g0 + pl] = any;
    }
}

// ---------------------------------------------------------------------------------------------
// k_polyline_fused8 (round 5): the same workgroup -- two polylines, three row tiles, K / V in LDS -- on EIGHT waves, two per SIMD.
// k_polyline_fused is issue-bound at one wave per SIMD (profiles/r03_stage_profile_polyline_fused.txt: six three-tile GEMM phases of
// 72 MFMAs in six dependent chains, three-tile LayerNorms of ~100 VALU instructions per thread and tile, 300 VGPRs), and its 143 KB of
// LDS admit no second workgroup.  Here wave w owns ONE output tile (16 features) of every Linear for the three row tiles -- a unit
// is 32 VGPRs instead of 64, a GEMM phase 36 MFMAs per wave instead of 72, a LayerNorm pass 1.5 tiles per thread instead of 3 -- and
// the second wave of a SIMD computes while the first waits.  The Q projection is the exception: the two waves of a head (2 h, 2 h + 1)
// each take BOTH feature tiles of the head for their own rows (wave 2 h + s: head tile s and the tail tile) with the two-tile unit, so
// that a wave holds the whole Q^T of the attentions it runs (its polyline's head tile and tail rows) and nothing is exchanged; the
// tail tile's Q is computed twice (48 instead of 36 MFMAs in that phase).  Every output element is produced by the same
// instruction sequence as in k_polyline_fused: bit-identical (test_packed_polyline_tiling_is_bitwise_identical).
// ---------------------------------------------------------------------------------------------
constexpr int NT8 = 2 * NTHREADS;

// LayerNorm of the three tiles by 512 threads: tiles 0 / 1 by the two halves of the workgroup, the tail tile by the first half
__device__ __forceinline__ void layernorm_planes_3x8(const float* src, xhalf* P, int p_stride, const float* __restrict__ g,
                                                     const float* __restrict__ b, int tid) {
    const int t = tid >> 8, t16 = tid & 255;
    layernorm_planes<false>(src + t * TM * LDT, LDT, P + t * p_stride, g, b, t16);
    if (tid < NTHREADS) layernorm_planes<false>(src + 2 * TM * LDT, LDT, P + 2 * p_stride, g, b, t16);
}

// MERGE (round 5, second form): LN_tgt and LN1 in ONE phase and the K, V and Q projections in ONE phase.  The block input of LN_tgt
// (fixed over the layers) stays in 16 VGPRs of the thread that normalises it -- no staging copy per layer, no barrier behind it -- and
// LN1 writes the second plane set, so the Q projection no longer waits for the K / V projections to release the first: seven
// barriers per layer instead of ten, two LayerNorm latency chains side by side.  Per element the same instruction sequence: same bits.
template <bool MERGE>
__global__ __launch_bounds__(NT8) void k_polyline_fused8(XBlockPX p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                                   // [3][16][LDT]
    xhalf* P1 = reinterpret_cast<xhalf*>(X + 3 * TM * LDT);            // [3] plane sets
    xhalf* P2 = P1 + 3 * NPL * PLANE;                                  // [3] plane sets
    xhalf* KL = P2 + 3 * NPL * PLANE;                                  // K blocks of the two polylines
    xhalf* VL = KL + 2 * KV_BLOCK_HALFS;                               // V blocks
    float* kb = reinterpret_cast<float*>(VL + 2 * KV_BLOCK_HALFS);     // [2][32] additive key mask
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(kb + 64);           // [48]
    uint8_t* novalid_s = rowvalid + 48;                                // [48]
    float* S = NPL == 2 ? reinterpret_cast<float*>(P2) : reinterpret_cast<float*>(novalid_s + 80);  // fp32 staging [3][16][LDT]
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
#ifdef TB_PROFILE_ENC  // (stage profile: clock64 stamps of thread 0 of one workgroup, printed from the device; tools/gpu_stage_profile_enc.sh)
    __shared__ long long enc_prof[64];
    __shared__ long long enc_wave[16];
    int enc_np = 0;
#define ENC_STAMP() do { if (tid == 0 && enc_np < 64) enc_prof[enc_np++] = clock64(); } while (0)
#else
#define ENC_STAMP() do { } while (0)
#endif
    ENC_STAMP();
    const int head = wave >> 1, half = wave & 1;  // attention head of this wave's output tile / which of the head's two feature tiles
    const int g0 = blockIdx.x * 2;
    const int po = m * LDP + kq * 8;
    constexpr int PS = NPL * PLANE;

    // rows of the three tiles: tile 0 / 1 = nodes 0 .. 15 of polyline g0 / g0 + 1, tile 2 rows 0 .. 3 / 4 .. 7 = their nodes 16 .. 19
    auto load3 = [&](float* D) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + i * NT8;
            const int t = idx >> 9, row = (idx >> 5) & 15, c4 = (idx & 31) * 4;
            f32x4 v;
            if (t < 2) v = ldg4(p.src + ((size_t)(g0 + t) * PL_NODES + row) * H + c4);
            else v = row < 8 ? ldg4(p.src + ((size_t)(g0 + (row >> 2)) * PL_NODES + 16 + (row & 3)) * H + c4) : splat(0.f);
            st4(D + t * TM * LDT + row * LDT + c4, v);
        }
    };
    WUnit1X u, u2;
    WUnitX uq;
    wload1x(u, wnext1x(p.W, p.LX[0].wkv, p.W + p.L[0].bkv, wave), lane);
    // MERGE: this thread's segments of the block input for LN_tgt (layernorm_planes_3x8's assignment: row (tid & 255) >> 4, columns
    // (tid & 15) * 8 .. + 7 of head tile tid >> 8; the first half of the workgroup also of the tail tile, whose rows 8 .. 15 are zeros)
    f32x4 tg_a, tg_c, tt_a = splat(0.f), tt_c = splat(0.f);
    if (MERGE) {
        const int t = tid >> 8, r = (tid & 255) >> 4, c0 = (tid & 15) * 8;
        const float* ph = p.src + ((size_t)(g0 + t) * PL_NODES + r) * H + c0;
        tg_a = ldg4(ph); tg_c = ldg4(ph + 4);
        if (tid < NTHREADS && r < 8) {
            const float* pt = p.src + ((size_t)(g0 + (r >> 2)) * PL_NODES + 16 + (r & 3)) * H + c0;
            tt_a = ldg4(pt); tt_c = ldg4(pt + 4);
        }
    }
    load3(X);
    // the statistics of the block input are taken ONCE: LN_tgt of the three layers and LN1 of layer 0 (X is the block input there,
    // same thread assignment) differ in their parameters only
    LnSeg sg{}, st{};
    if (MERGE) {
        sg = ln_stats(tg_a, tg_c);
        st = ln_stats(tt_a, tt_c);
    }
    if (tid < 48) {
        const int t = tid >> 4, r = tid & 15;
        rowvalid[tid] = t < 2 ? p.src_valid[(size_t)(g0 + t) * PL_NODES + r] : (r < 8 ? p.src_valid[(size_t)(g0 + (r >> 2)) * PL_NODES + 16 + (r & 3)] : 0);
    } else if (tid >= 64 && tid < 128) {
        const int pl = (tid - 64) >> 5, k = (tid - 64) & 31;
        kb[tid - 64] = (k < PL_NODES && p.src_valid[(size_t)(g0 + pl) * PL_NODES + k]) ? 0.f : -INFINITY;
    }
    {   // key slots 20 .. 31 stay zero for the whole kernel (masked keys must hold finite data)
        const xh8 z = {};
        xh8* kv8 = reinterpret_cast<xh8*>(KL);
        for (int i = tid; i < 4 * KV_BLOCK_HALFS / 8; i += NT8) kv8[i] = z;
    }
    __syncthreads();
    ENC_STAMP();  // 1: prologue done
    // the plane set the Q projection reads LN1's output from
    xhalf* const PQ = MERGE ? P2 : P1;

    // this wave's output tile of a Linear over the three row tiles; the next unit is requested first (ua != ub)
    auto gemm3 = [&](const WUnit1X& ua, WUnit1X& ub, const WNext1X& next, const xhalf* P, f32x4 (&a)[3]) {
#pragma unroll
        for (int t = 0; t < 3; ++t) a[t] = ua.b;
        if (MERGE) {  // (hand-pipelined: operand reads a chunk ahead, MFMAs round the three tiles, the next unit's requests under them)
            wmma1x_3(a, ua, P + po, PS, PLANE, ub, next, lane);
            return;
        }
        wmma1x_pf(a[0], ua, P + po, PLANE, ub, next, lane);
        wmma1x(a[1], ua, P + PS + po, PLANE);
        wmma1x(a[2], ua, P + 2 * PS + po, PLANE);
    };

#pragma unroll 1
    for (int l = 0; l < p.n_layer; ++l) {
        const XLayerW& L = p.L[l];
        const XLayerX& LX = p.LX[l];
        const float* lnblk = p.W + L.ln1_g;
        // ---- K / V of the layer from the block input
        if (MERGE) {
            ln_apply(sg, P1 + (tid >> 8) * PS, lnblk + 256, lnblk + 384, tid & 255);
            if (tid < NTHREADS) ln_apply(st, P1 + 2 * PS, lnblk + 256, lnblk + 384, tid);
            if (l == 0) {
                ln_apply(sg, P2 + (tid >> 8) * PS, lnblk, lnblk + 128, tid & 255);
                if (tid < NTHREADS) ln_apply(st, P2 + 2 * PS, lnblk, lnblk + 128, tid);
            } else {
                layernorm_planes_3x8(X, P2, PS, lnblk, lnblk + 128, tid);
            }
        } else {
            if (l > 0) {  // (layer 0: X still holds the block input)
                load3(S);
                __syncthreads();
            }
            layernorm_planes_3x8(l > 0 ? S : X, P1, PS, lnblk + 256, lnblk + 384, tid);
        }
        ENC_STAMP();  // LN_tgt (+ LN1)
        __syncthreads();
        ENC_STAMP();  // barrier
        {
            f32x4 ak[3], av[3];
            gemm3(u, u2, wnext1x(p.W, LX.wkv, p.W + L.bkv, 8 + wave), P1, ak);
            // K: lane (kq, m) = token m, features 4 kq + r of this wave's feature tile -> its 8-byte half of the key's K granule
            auto k_store = [&](xhalf* kblk, int kt, int krow, const f32x4& a) {
                xhalf* pk = kblk + head * (NPL * 1024) + (kt * 64 + kq * 16 + krow) * 8 + half * 4;
                xh4 h, lo;
                split2(a, h, lo);
                *reinterpret_cast<xh4*>(pk) = h;
                if (NPL == 2) *reinterpret_cast<xh4*>(pk + 1024) = lo;
            };
            k_store(KL, 0, m, ak[0]);
            k_store(KL + KV_BLOCK_HALFS, 0, m, ak[1]);
            if (m < 8) k_store(KL + (m >> 2) * KV_BLOCK_HALFS, 1, m & 3, ak[2]);
            ENC_STAMP();  // K projection + stores
            // V with the MFMA operands swapped (k_polyline_fused): lane (kq, m) holds tokens 4 kq + r of feature m of this wave's d tile
            {
                const float* bv = p.W + L.bkv + (8 + wave) * 16 + m;
                const f32x4 ba = splat(bv[0]);
#pragma unroll
                for (int t = 0; t < 3; ++t) av[t] = ba;
                if (MERGE) {
                    wmma1x_3<true>(av, u2, P1 + po, PS, PLANE, uq, xlayer_first_x(p.W, L, LX, head), lane);
                } else {
                    wmma1x_pf2<true>(av[0], u2, P1 + po, PLANE, uq, xlayer_first_x(p.W, L, LX, head), lane);
                    wmma1x<true>(av[1], u2, P1 + PS + po, PLANE);
                    wmma1x<true>(av[2], u2, P1 + 2 * PS + po, PLANE);
                }
                auto v_store = [&](xhalf* pv, const f32x4& a) {
                    xh4 h, lo;
                    split2(a, h, lo);
                    *reinterpret_cast<xh4*>(pv) = h;
                    if (NPL == 2) *reinterpret_cast<xh4*>(pv + 1024) = lo;
                };
                v_store(VL + head * (NPL * 1024) + (kq * 16 + m) * 8 + half * 512, av[0]);
                v_store(VL + KV_BLOCK_HALFS + head * (NPL * 1024) + (kq * 16 + m) * 8 + half * 512, av[1]);
                // tail tile: lane group kq = 0 / 1 holds nodes 16 .. 19 of polyline 0 / 1 = keys 16 .. 19 = elements 4 .. 7 of key quad 0
                if (kq < 2) v_store(VL + kq * KV_BLOCK_HALFS + head * (NPL * 1024) + m * 8 + 4 + half * 512, av[2]);
            }
        }
        ENC_STAMP();  // V projection + stores
        if (!MERGE) {
            __syncthreads();  // (the projections have read P1: LayerNorm 1 may overwrite it; K / V are in place)
            // ---- the layer on the three tiles
            layernorm_planes_3x8(X, P1, PS, lnblk, lnblk + 128, tid);
            __syncthreads();
        }
        {
            // Q of this wave's head for its own rows: head tile `half` (polyline `half`) and the tail tile
            f32x4 qo[2] = {uq.b[0], uq.b[1]}, qt[2] = {uq.b[0], uq.b[1]};
            wload1x(u2, wnext1x(p.W, LX.wo, p.W + L.bo, wave), lane);
            wmmax(qo[0], qo[1], uq, PQ + half * PS + po, PLANE);
            wmmax(qt[0], qt[1], uq, PQ + 2 * PS + po, PLANE);
            ENC_STAMP();  // Q projection
#ifdef TB_PROFILE_ENC
            if (lane == 0 && l == 1) enc_wave[wave] = clock64();  // (per-wave arrival at the barrier in front of the attention, layer 1)
#endif
            if (MERGE) __syncthreads();  // (K / V of every head are in place; every wave has read its Q rows from the second plane set)
            ENC_STAMP();  // barrier
            f32x4 oo[2], ot[2];
            const xhalf* Kp = KL + half * KV_BLOCK_HALFS;
            const xhalf* Vp = VL + half * KV_BLOCK_HALFS;
            const bool no = attention_oneblock_x(qo, Kp, Vp, kb + half * KEYPAD, head, lane, oo);
            const bool nt = attention_oneblock_x(qt, Kp, Vp, kb + half * KEYPAD, head, lane, ot);
            planes_store_c<false>(P2 + half * PS, 2 * head, lane, oo[0]);
            planes_store_c<false>(P2 + half * PS, 2 * head + 1, lane, oo[1]);
            const int grp = m >> 2;  // tail rows 0 .. 3 / 4 .. 7 belong to polyline 0 / 1, rows 8 .. 15 are padding (written by the wave of polyline 0)
            const bool mine = grp == half, pad = half == 0 && grp >= 2;
            if (mine || pad) {
                const f32x4 z = splat(0.f);
                planes_store_c<false>(P2 + 2 * PS, 2 * head, lane, mine ? ot[0] : z);
                planes_store_c<false>(P2 + 2 * PS, 2 * head + 1, lane, mine ? ot[1] : z);
            }
            if (head == 0 && kq == 0) {
                novalid_s[half * TM + m] = no ? 1 : 0;
                if (mine || pad) novalid_s[2 * TM + m] = (mine ? nt : true) ? 1 : 0;
            }
        }
        ENC_STAMP();  // two one-block attentions + plane stores
#ifdef TB_PROFILE_ENC
        if (lane == 0 && l == 1) enc_wave[8 + wave] = clock64();  // (... and at the barrier behind it)
#endif
        __syncthreads();
        ENC_STAMP();  // barrier
        {
            f32x4 a[3];
            gemm3(u2, u, wnext1x(p.W, LX.w1, p.W + L.b1, wave), P2, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool nv = novalid_s[t * TM + m] != 0;
                float* px = cptr(X + t * TM * LDT, LDT, wave, lane);
                const f32x4 xo = lds4(px);
                st4(px, nv ? xo : xo + a[t]);
            }
        }
        ENC_STAMP();  // output projection + residual
        __syncthreads();
        ENC_STAMP();  // barrier
        layernorm_planes_3x8(X, P1, PS, lnblk + 512, lnblk + 640, tid);
        ENC_STAMP();  // LN2
        __syncthreads();
        ENC_STAMP();  // barrier
        {
            f32x4 a[3];
            gemm3(u, u2, wnext1x(p.W, LX.w2, p.W + L.b2, wave), P1, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) planes_store_c(P2 + t * PS, wave, lane, relu4(a[t]));
        }
        ENC_STAMP();  // FFN 1 + ReLU + plane stores
        __syncthreads();
        ENC_STAMP();  // barrier
        {
            f32x4 a[3];
            const int ln = l + 1 < p.n_layer ? l + 1 : l;
            gemm3(u2, u, wnext1x(p.W, p.LX[ln].wkv, p.W + p.L[ln].bkv, wave), P2, a);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bool rv = rowvalid[t * TM + m] != 0;
                float* px = cptr(X + t * TM * LDT, LDT, wave, lane);
                st4(px, rv ? lds4(px) + a[t] : splat(0.f));
            }
        }
        ENC_STAMP();  // FFN 2 + residual
        __syncthreads();
        ENC_STAMP();  // barrier
    }
#ifdef TB_PROFILE_ENC
    if (MERGE && tid == 0 && blockIdx.x == 700) {
        printf("ENCPROF8 entry->prologue %lld\n", enc_prof[1] - enc_prof[0]);
        const char* nm[16] = {"LNtgt+LN1", "bar", "Kproj+store", "Vproj+store", "Qproj", "bar", "attn2+store", "bar", "Oproj+res", "bar", "LN2", "bar", "F1+relu+store", "bar", "F2+res", "bar"};
        for (int l = 0; l < 3; ++l) {
            printf("ENCPROF8 layer %d:", l);
            for (int i = 0; i < 16; ++i) printf(" %s %lld", nm[i], enc_prof[2 + l * 16 + i] - enc_prof[1 + l * 16 + i]);
            printf("\n");
        }
        printf("ENCPROF8 layer 1, arrival of waves 0 .. 7 at the barrier in front of the attention (cycles after the first):");
        long long m0 = enc_wave[0], m1 = enc_wave[8];
        for (int w = 1; w < 8; ++w) { m0 = enc_wave[w] < m0 ? enc_wave[w] : m0; m1 = enc_wave[8 + w] < m1 ? enc_wave[8 + w] : m1; }
        for (int w = 0; w < 8; ++w) printf(" %lld", enc_wave[w] - m0);
        printf("; behind it:");
        for (int w = 0; w < 8; ++w) printf(" %lld", enc_wave[8 + w] - m1);
        printf("\n");
    }
#endif
    if (tid < NTHREADS) {  // MapEncoder: max over the valid nodes of the polyline, one (polyline, feature) per thread (k_polyline_fused)
        const int pl = tid >> 7, f = tid & 127;
        float mx = -INFINITY;
        bool any = false;
#pragma unroll
        for (int k = 0; k < PL_NODES; ++k) {
            const int t = k < TM ? pl : 2, r = k < TM ? k : pl * 4 + (k - TM);
            if (rowvalid[t * TM + r]) {
                any = true;
                mx = fmaxf(mx, X[(t * TM + r) * LDT + f]);
            }
        }
        p.pool_out[(size_t)(g0 + pl) * H + f] = any ? mx : 0.f;
        if (f == 0) p.pool_valid[g0 + pl] = any;
    }
}

void launch_polyline_fused_x(const XBlockPX& p, int G, hipStream_t s, int eight_waves) {
    const size_t lds = PLF_LDS_BYTES + (NPL == 1 ? 3 * TM * LDT * 4 : 0);
    if (eight_waves == 2) hipLaunchKernelGGL(k_polyline_fused8<true>, dim3(G / 2), dim3(NT8), lds, s, p);
    else if (eight_waves) hipLaunchKernelGGL(k_polyline_fused8<false>, dim3(G / 2), dim3(NT8), lds, s, p);
    else hipLaunchKernelGGL(k_polyline_fused, dim3(G / 2), dim3(NTHREADS), lds, s, p);
}

// hoist + block of the polyline encoder on the packed tiling; `p` as for launch_xblock_x (n_rows = 20, n_pad = 32, tgt = src)
void launch_polyline_block_x(const XBlockPX& p, int G, float* K, float* VT, float* kbias, hipStream_t s) {
    const int i1 = p.n_layer > 1 ? 1 : 0, i2 = p.n_layer > 2 ? 2 : 0;
    const size_t lds1 = TM * LDT * sizeof(float) + 2 * PLANES_BYTES + 64, lds2 = 2 * TM * LDT * sizeof(float) + 4 * PLANES_BYTES + 64;
    hipLaunchKernelGGL(k_kv_hoist_plh, dim3(G / 2), dim3(NTHREADS), 2 * TM * LDT * sizeof(float) + 2 * PLANES_BYTES, s, p.W, p.L[0], p.L[i1],
                       p.L[i2], p.LX[0], p.LX[i1], p.LX[i2], p.n_layer, p.src, p.src_valid, K, VT, kbias);
    hipLaunchKernelGGL(k_kv_hoist_plt, dim3(G / 4), dim3(NTHREADS), TM * LDT * sizeof(float) + PLANES_BYTES, s, p.W, p.L[0], p.L[i1], p.L[i2],
                       p.LX[0], p.LX[i1], p.LX[i2], p.n_layer, p.src, p.src_valid, K, VT, kbias);
    hipLaunchKernelGGL(k_xattn_block_plh, dim3(G / 2), dim3(NTHREADS), lds2, s, p);
    hipLaunchKernelGGL(k_xattn_block_plt, dim3(G / 4), dim3(NTHREADS), lds1, s, p);
}

void launch_xblock_x(const XBlockPX& p, int G, hipStream_t s) {
    dim3 grid2((p.n_rows + 2 * TM - 1) / (2 * TM), G);
    if (p.n_rows > TM && (int)grid2.x * G >= 256) {  // two row tiles per workgroup share every weight unit -- when that still fills the chip
        const size_t lds2 = 2 * TM * LDT * sizeof(float) + 4 * PLANES_BYTES + 64;
        if (p.eye) hipLaunchKernelGGL(k_xattn_block_x2<true>, grid2, dim3(NTHREADS), lds2, s, p);
        else hipLaunchKernelGGL(k_xattn_block_x2<false>, grid2, dim3(NTHREADS), lds2, s, p);
        return;
    }
    dim3 grid((p.n_rows + TM - 1) / TM, G);
    const size_t lds = TM * LDT * sizeof(float) + 2 * PLANES_BYTES + 32;
    if (p.eye) hipLaunchKernelGGL(k_xattn_block_x<true>, grid, dim3(NTHREADS), lds, s, p);
    else hipLaunchKernelGGL(k_xattn_block_x<false>, grid, dim3(NTHREADS), lds, s, p);
}

// GRU over the S steps of a 16-agent tile on the XDL pipe (the twin of k_gru_scan): the three hidden states live in LDS as fp32
// tiles (convex update) and are re-split into fp16 pairs every step; plane rotation as in k_step_x's GRU block.  grid (a_pad/16, B)
constexpr int SCANX_LDS_BYTES = 6 * TM * LDT * 4 + 4 * PLANES_BYTES + 64;

__global__ __launch_bounds__(NTHREADS) void k_gru_scan_x(ScanP p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                 // step input
    float* HS = X + TM * LDT;        // 3 hidden tiles [3][16][LDT]
    float* AGG = HS + 3 * TM * LDT;  // aggregate
    float* Y0 = AGG + TM * LDT;      // head scratch
    xhalf* PA = reinterpret_cast<xhalf*>(Y0 + TM * LDT);
    xhalf* PB = PA + NPL * PLANE;
    xhalf* PC = PB + NPL * PLANE;
    xhalf* PD = PC + NPL * PLANE;
    uint8_t* rowvalid = reinterpret_cast<uint8_t*>(PD + NPL * PLANE);
    uint8_t* anyvalid = rowvalid + 16;
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int b = blockIdx.y, row0 = blockIdx.x * TM;
    const int n_real = min(TM, p.A - row0);
    float* H0 = HS, *H1 = HS + TM * LDT, *H2 = HS + 2 * TM * LDT;
    WUnitX u;
    wloadx(u, gru_first_x(p.W, p.gru[0], p.grux[0], wave), lane);
    for (int i = tid; i < 3 * TM * LDT; i += NTHREADS) HS[i] = 0.f;
    for (int i = tid; i < TM * LDT; i += NTHREADS) AGG[i] = p.mode == 0 ? -INFINITY : 0.f;
    if (tid < TM) anyvalid[tid] = 0;
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < p.S; ++s) {
        load_tile(X, LDT, p.x + (((size_t)b * p.S + s) * p.A + row0) * H, n_real, tid);
        if (tid < TM) {
            const uint8_t v = tid < n_real ? p.valid[((size_t)b * p.S + s) * p.A + row0 + tid] : 0;
            rowvalid[tid] = v;
            if (v) anyvalid[tid] = 1;
        }
        __syncthreads();
        tile_to_planes(X, LDT, PA, tid);
        tile_to_planes(H0, LDT, PB, tid);
        tile_to_planes(H1, LDT, PD, tid);
        __syncthreads();
        // (the new hidden of a layer overwrites its fp32 tile in place: every lane reads and writes only its own four elements)
        gru_layer_x(p.W, p.gru[0], p.grux[0], PA, PB, H0, PC, H0, rowvalid, nullptr, 0, tid, u, gru_first_x(p.W, p.gru[1], p.grux[1], wave));
        tile_to_planes(H2, LDT, PB, tid);  // h0's planes are free after the barrier that closed layer 0
        gru_layer_x(p.W, p.gru[1], p.grux[1], PC, PD, H1, PA, H1, rowvalid, nullptr, 0, tid, u, gru_first_x(p.W, p.gru[2], p.grux[2], wave));
        gru_layer_x(p.W, p.gru[2], p.grux[2], PA, PB, H2, nullptr, H2, rowvalid, nullptr, 0, tid, u, gru_first_x(p.W, p.gru[0], p.grux[0], wave));
        // aggregate (outputs of invalid rows are already zero; hidden reset to zero likewise)
        for (int i = tid; i < TM * 32; i += NTHREADS) {
            const int r = i >> 5, c4 = (i & 31) * 4;
            if (p.mode == 0) {  // x.masked_fill(~valid, -1e3).amax(1)
                const f32x4 o = rowvalid[r] ? lds4(H2 + r * LDT + c4) : splat(-1e3f);
                const f32x4 a = lds4(AGG + r * LDT + c4);
                st4(AGG + r * LDT + c4, f32x4{fmaxf(a.x, o.x), fmaxf(a.y, o.y), fmaxf(a.z, o.z), fmaxf(a.w, o.w)});
            } else if (rowvalid[r]) {
                st4(AGG + r * LDT + c4, lds4(H2 + r * LDT + c4) + lds4(X + r * LDT + c4));
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < TM * 32; i += NTHREADS) {  // rows that were never valid -> 0
        const int r = i >> 5, c4 = (i & 31) * 4;
        if (!anyvalid[r]) st4(AGG + r * LDT + c4, splat(0.f));
    }
    __syncthreads();
    if (tid < n_real) p.out_valid[(size_t)b * p.A + row0 + tid] = anyvalid[tid];
    if (p.mode == 1) {
        store_tile(p.out_feat + ((size_t)b * p.A + row0) * H, AGG, LDT, n_real, tid);
        return;
    }
    // latent mean = W2 relu(W1 agg + b1) + b2, masked (latent_encoder.py:168-178): once per tile, the fp32-MFMA Linear
    {
        f32x4 acc[2];
        linear128<128>(acc, p.W + p.head_w1, p.W + p.head_b1, AGG + m * LDT + kq * 32, wave, lane);
        st4(cptr(Y0, LDT, 2 * wave, lane), relu4(acc[0]));
        st4(cptr(Y0, LDT, 2 * wave + 1, lane), relu4(acc[1]));
    }
    __syncthreads();
    {
        const int r = tid >> 4, o = tid & 15;
        float sacc = p.W[p.head_b2 + o];
        const float* w2 = p.W + p.head_w2 + o * H;
        for (int k = 0; k < H; ++k) sacc = fmaf(Y0[r * LDT + k], w2[k], sacc);
        if (r < n_real) p.out_mean[((size_t)b * p.A + row0 + r) * 16 + o] = anyvalid[r] ? sacc : 0.f;
    }
}

// Destination logits of (agent, polyline) pairs, XDL: one workgroup owns a 16-polyline tile of a scene and walks DEST_AGENTS
// agents with the 128x128 Linear resident in registers (the fp32 kernel re-reads that 64 KB unit for every (agent, tile)).
//   logit = w2 . relu(LN1(W1 relu(LN0(U[p] + V[a])) + b1)) + b2, masked by candidate type.      grid (P/16, ceil(A/DEST_AGENTS), B)
constexpr int DEST_AGENTS = 16;

__device__ __forceinline__ bool dest_candidate_x(int mtype, bool mvalid, int atype) {  // goal_manager.py:235-244
    if (!(mvalid && mtype >= 0 && mtype < 5)) return false;
    if (atype == 0 && mtype == 3) return false;
    if (atype == 1 && mtype < 4) return false;
    if (atype == 2 && mtype < 3) return false;
    return true;
}

__global__ __launch_bounds__(NTHREADS) void k_dest_pairs_x(DestP p) {
    __shared__ __attribute__((aligned(16))) float Ut[TM * LDT];
    __shared__ __attribute__((aligned(16))) float X[TM * LDT];
    __shared__ __attribute__((aligned(16))) xhalf P1[NPL * PLANE];
    __shared__ int mtype_s[TM];
    __shared__ uint8_t mvalid_s[TM];
    const int tid = threadIdx.x, wave = wave_of(tid), lane = tid & 63, kq = lane >> 4, m = lane & 15;
    const int p0 = blockIdx.x * TM, b = blockIdx.z, a0 = blockIdx.y * DEST_AGENTS;
    const int n_real = min(TM, p.P - p0);
    WUnitX u;
    wloadx(u, wstdx(p.W, p.w1x, p.W + p.b1, wave), lane);
    // which agent classes have any candidate polyline in this scene (rows that are all -inf become 0, :331-332)
    // (an agent WITHOUT a type -- all-false one-hot, class index -1 -- has no exclusion: every valid lane polyline is a candidate,
    // goal_manager.py:237-244 with agent_type all False; golden `edge_scenes`)
    bool mine[4] = {false, false, false, false};
    for (int q = tid; q < p.P; q += NTHREADS) {
        const int mt = p.map_type[(size_t)b * p.P + q];
        const bool mv = p.map_fvalid[(size_t)b * p.P + q] != 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) mine[c] |= dest_candidate_x(mt, mv, c == 3 ? -1 : c);
    }
    const bool any0 = __syncthreads_or(mine[0]), any1 = __syncthreads_or(mine[1]), any2 = __syncthreads_or(mine[2]),
               anyn = __syncthreads_or(mine[3]);
    for (int i = tid; i < TM * 32; i += NTHREADS) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        st4(Ut + r * LDT + c4, r < n_real ? ldg4(p.U + ((size_t)b * p.P + p0 + r) * H + c4) : splat(0.f));
    }
    if (tid < TM) {
        mtype_s[tid] = tid < n_real ? p.map_type[(size_t)b * p.P + p0 + tid] : -1;
        mvalid_s[tid] = tid < n_real ? p.map_fvalid[(size_t)b * p.P + p0 + tid] : 0;
    }
    const int row = tid >> 4, c0 = (tid & 15) * 8;
    const f32x4 g0a = ldg4(p.W + p.ln0_g + c0), g0b = ldg4(p.W + p.ln0_g + c0 + 4), b0a = ldg4(p.W + p.ln0_b + c0), b0b = ldg4(p.W + p.ln0_b + c0 + 4);
    const f32x4 g1a = ldg4(p.W + p.ln1_g + c0), g1b = ldg4(p.W + p.ln1_g + c0 + 4), b1a = ldg4(p.W + p.ln1_b + c0), b1b = ldg4(p.W + p.ln1_b + c0 + 4);
    const f32x4 w2a = ldg4(p.W + p.w2 + c0), w2b = ldg4(p.W + p.w2 + c0 + 4);
    const float bias2 = p.W[p.b2];
    __syncthreads();
#pragma unroll 1
    for (int a = a0; a < min(a0 + DEST_AGENTS, p.A); ++a) {
        const int atype = p.agent_type[(size_t)b * p.A + a];
        const bool dvalid = p.dist_valid[(size_t)b * p.A + a] != 0;
        const bool any_cand = atype == 0 ? any0 : (atype == 1 ? any1 : (atype == 2 ? any2 : anyn));
        {   // relu(LN0(U + V[a])) -> planes
            const float* v = p.V + ((size_t)b * p.A + a) * H + c0;
            const f32x4 xa = lds4(Ut + row * LDT + c0) + ldg4(v), xc = lds4(Ut + row * LDT + c0 + 4) + ldg4(v + 4);
            const float sm = row16_sum((xa.x + xa.y) + (xa.z + xa.w) + (xc.x + xc.y) + (xc.z + xc.w));
            const float mean = sm * (1.0f / 128.0f);
            const f32x4 da = xa - splat(mean), dc = xc - splat(mean);
            const float var = row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) +
                                        (dc.z * dc.z + dc.w * dc.w));
            const float rstd = 1.0f / sqrtf(var * (1.0f / 128.0f) + LN_EPS);
            planes_store4(P1, PLANE, LDP, row, c0, relu4(da * splat(rstd) * g0a + b0a));
            planes_store4(P1, PLANE, LDP, row, c0 + 4, relu4(dc * splat(rstd) * g0b + b0b));
        }
        __syncthreads();
        {
            f32x4 acc[2] = {u.b[0], u.b[1]};
            wmmax(acc[0], acc[1], u, P1 + m * LDP + kq * 8, PLANE);
            st4(cptr(X, LDT, 2 * wave, lane), acc[0]);
            st4(cptr(X, LDT, 2 * wave + 1, lane), acc[1]);
        }
        __syncthreads();
        {   // w2 . relu(LN1(x)) + b2
            const f32x4 xa = lds4(X + row * LDT + c0), xc = lds4(X + row * LDT + c0 + 4);
            const float sm = row16_sum((xa.x + xa.y) + (xa.z + xa.w) + (xc.x + xc.y) + (xc.z + xc.w));
            const float mean = sm * (1.0f / 128.0f);
            const f32x4 da = xa - splat(mean), dc = xc - splat(mean);
            const float var = row16_sum((da.x * da.x + da.y * da.y) + (da.z * da.z + da.w * da.w) + (dc.x * dc.x + dc.y * dc.y) +
                                        (dc.z * dc.z + dc.w * dc.w));
            const float rstd = 1.0f / sqrtf(var * (1.0f / 128.0f) + LN_EPS);
            const f32x4 ya = relu4(da * splat(rstd) * g1a + b1a), yc = relu4(dc * splat(rstd) * g1b + b1b);
            float s = 0.f;
            s = fmaf(ya.x, w2a.x, s); s = fmaf(ya.y, w2a.y, s); s = fmaf(ya.z, w2a.z, s); s = fmaf(ya.w, w2a.w, s);
            s = fmaf(yc.x, w2b.x, s); s = fmaf(yc.y, w2b.y, s); s = fmaf(yc.z, w2b.z, s); s = fmaf(yc.w, w2b.w, s);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
            if ((tid & 15) == 0 && row < n_real) {
                float lg = s + bias2;
                if (!dest_candidate_x(mtype_s[row], mvalid_s[row] != 0, atype)) lg = -INFINITY;
                if (!dvalid || !any_cand) lg = 0.f;
                p.logits[((size_t)b * p.A + a) * p.P + p0 + row] = lg;
            }
        }
        // (the next agent's planes are written before anyone reads X again; P1 was last read before the barrier above)
    }
}

// `lds_pad`: dynamic LDS the kernel does not use -- it only bounds how many of its workgroups a CU holds (26 KB of static LDS each:
// three by registers).  Beside the latent branch (run_encode: the destination predictor on the side stream) the 32 768-workgroup
// grid otherwise takes three of every SIMD's wave slots and the branch's small dependent kernels run at half speed.
void launch_dest_pairs_x(const DestP& p, hipStream_t s, int lds_pad) {
    dim3 grid((p.P + TM - 1) / TM, (p.A + DEST_AGENTS - 1) / DEST_AGENTS, p.B);
    hipLaunchKernelGGL(k_dest_pairs_x, grid, dim3(NTHREADS), (size_t)lds_pad, s, p);
}

void launch_gru_scan_x(const ScanP& p, int a_pad, hipStream_t s) {
    hipLaunchKernelGGL(k_gru_scan_x, dim3(a_pad / TM, p.B), dim3(NTHREADS), SCANX_LDS_BYTES, s, p);
}


// fp16-pair range flag of THIS translation unit (the scene encoders): OR it into *out (bit 1) and clear it (tb_check_status)
__global__ void k_range_flag_take_encode(unsigned int* out) {
    const unsigned int f = atomicExch(&g_range_flag, 0u);
    if (f) atomicOr(out, 2u);
}
void launch_range_flag_take_encode(unsigned int* out, hipStream_t s) { hipLaunchKernelGGL(k_range_flag_take_encode, dim3(1), dim3(1), 0, s, out); }

hipError_t configure_encodex_kernels() {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_polyline_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             PLF_LDS_BYTES + (NPL == 1 ? 3 * TM * LDT * 4 : 0));
    if (e != hipSuccess) return e;
    const hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_polyline_fused8<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              PLF_LDS_BYTES + (NPL == 1 ? 3 * TM * LDT * 4 : 0));
    if (e8 != hipSuccess) return e8;
    const hipError_t e8m = hipFuncSetAttribute(reinterpret_cast<const void*>(k_polyline_fused8<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               PLF_LDS_BYTES + (NPL == 1 ? 3 * TM * LDT * 4 : 0));
    if (e8m != hipSuccess) return e8m;
    const hipError_t ed = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dest_pairs_x), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (ed != hipSuccess) return ed;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_scan_x), hipFuncAttributeMaxDynamicSharedMemorySize, SCANX_LDS_BYTES);
}

}  // namespace TB_XNS
}  // namespace tb