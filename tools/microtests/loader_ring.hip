// Micro-benchmark for the NEXT structure of the step kernel (profiles/r04_experiments_not_kept.txt): a chain of
// fp16-pair GEMM "units" ([32 outputs per wave x 16 agents x 128 k], 64 KB of weights per unit and workgroup, the geometry of
// tb_device_xdl.hpp::wmmax_pf) whose weights arrive
//   MODE 0  the way they do today: every compute wave requests the NEXT unit's sixteen 1-KiB fragments into registers while it
//           multiplies the current one (register prefetch, one unit ahead);
//   MODE 1  from an LDS ring filled by a FIFTH wave with LDS-DMA (global_load_lds_dwordx4): five 16-KB slots (one slot = one compute
//           wave's quarter of a unit: 80 KB, what is free next to the 79 KB carve of the fp16-pair kernel; an eight-slot ring = two
//           units of run-ahead is measured next to it: what the bf16 build's half-size units would get), no register buffers.
// The point of MODE 1 is the synchronisation, because the step kernel has ~200 workgroup barriers per step on data-dependent paths and
// s_barrier is workgroup-wide: the loader wave does not know the consumers' control flow -- it LOOPS on s_barrier (issue what fits,
// arrive, repeat), so every consumer barrier is matched whichever one it is; a slot is released by the consumer's progress counter
// in LDS, and a slot's arrival is detected by the consumer itself (the last 16 bytes of a quarter are poisoned with an fp16 NaN
// pattern no finite weight contains, and polled), so the loader never waits for its own loads.
// Between two units a "gap" of dependent VALU work stands for the LayerNorm / attention phases during which today's load path idles.
// Prints cycles per unit for gaps of 0 / ~1 k / ~2 k cycles and checks MODE 1's result against MODE 0's bit for bit.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/microtests/loader_ring.hip -o tools/microtests/bin/loader_ring
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int UNIT_BYTES = 65536, QUARTER = 16384, NFRAG = 16;
constexpr int LDP = 136, PLANE = 16 * LDP;  // fp16 per plane row / per plane
constexpr unsigned POISON = 0xFFFFFFFFu;

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ f4 mfma(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// one global_load_lds_dwordx4: lane i copies 16 B from gsrc to LDS byte address lds_dst + 16 i (M0 = destination base; guide 5.7)
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__device__ __forceinline__ float gap_work(float x, int iters) {
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);  // one dependent VALU chain
    return x;
}

// epilogue of a unit: acc (+ cross terms) -> ReLU-ish squash -> fp16 pair -> the OTHER plane buffer, C layout (row m, 4 features)
__device__ __forceinline__ void store_planes(_Float16* P, int tile, int lane, f4 v) {
    const int m = lane & 15, kq = lane >> 4;
    h4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = v[i] * 0.05f;
        x = x > 2.f ? 2.f : (x < -2.f ? -2.f : x);
        h[i] = (_Float16)x;
        l[i] = (_Float16)((x - (float)h[i]) * 2048.0f);
    }
    _Float16* p = P + m * LDP + tile * 16 + kq * 4;
    *reinterpret_cast<h4*>(p) = h;
    *reinterpret_cast<h4*>(p + PLANE) = l;
}

template <int MODE, int NSLOT>
__global__ __launch_bounds__(MODE == 0 ? 256 : 320) void k_chain(const char* __restrict__ W, int n_units, int gap_iters, float* __restrict__ out,
                                                                    long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [ring 5 x 16 KB (MODE 1)] [planes A: 2 x PLANE fp16] [planes B] [ctrl]
    char* ring = smem;
    _Float16* PA = reinterpret_cast<_Float16*>(smem + (MODE == 1 ? NSLOT * QUARTER : 0));
    _Float16* PB = PA + 2 * PLANE;
    volatile int* ctrl = reinterpret_cast<volatile int*>(PB + 2 * PLANE);  // cons[0..3], done
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int m = lane & 15, kq = lane >> 4;
    // activations: something smooth and bounded
    for (int i = tid; i < 2 * PLANE; i += blockDim.x) {
        const int r = (i % PLANE) / LDP, c = i % LDP;
        PA[i] = (_Float16)(i < PLANE ? 0.01f * ((r * 7 + c * 3) % 41 - 20) : 0.003f * ((r + c) % 13 - 6));
        PB[i] = (_Float16)0.f;
    }
    if (tid < 8) ctrl[tid] = 0;
    if (MODE == 1)
        for (int s = tid; s < NSLOT; s += blockDim.x) *reinterpret_cast<volatile u4*>(ring + s * QUARTER + QUARTER - 16) = u4{POISON, POISON, POISON, POISON};
    __syncthreads();
    const long long t0 = clock64();

    if (MODE == 1 && wave == 4) {
        // ---------------- loader wave: issue what fits, arrive at whatever barrier the consumers are at, repeat
        const int total_q = n_units * 4;
        int g = 0;
        for (int tick = 0; tick < (1 << 22); ++tick) {
            while (g < total_q) {  // everything that fits: a consumer may be polling for any of it, and the barrier below cannot
                                   // complete before all of them have their data
                const int need = g - NSLOT;  // the quarter that occupied this slot before
                if (need >= 0 && __builtin_amdgcn_readfirstlane(ctrl[need & 3]) <= (need >> 2)) break;
                const char* src = W + (size_t)(g >> 2) * UNIT_BYTES + (size_t)(g & 3) * QUARTER + lane * 16;
                const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)((g % NSLOT) * QUARTER));  // (wave-uniform by construction: "s" operand)
#pragma unroll
                for (int i = 0; i < NFRAG; ++i) glds16(src + i * 1024, dst + i * 1024);
                ++g;
            }
            __builtin_amdgcn_s_barrier();
            if (__builtin_amdgcn_readfirstlane(ctrl[4])) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---------------- compute waves
    f4 sum = f4{0.f, 0.f, 0.f, 0.f};
    float gapv = 0.25f * lane;
    h8 cur[NFRAG], nxt[NFRAG];
    const char* wq = W + (size_t)wave * QUARTER + lane * 16;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < NFRAG; ++i) cur[i] = *reinterpret_cast<const h8*>(wq + i * 1024);
    }
    _Float16* Pin = PA;
    _Float16* Pout = PB;
    for (int u = 0; u < n_units; ++u) {
        const char* slot = ring + ((4 * u + wave) % NSLOT) * QUARTER;
        if (MODE == 0) {
            if (u + 1 < n_units) {
#pragma unroll
                for (int i = 0; i < NFRAG; ++i) nxt[i] = *reinterpret_cast<const h8*>(wq + (size_t)(u + 1) * UNIT_BYTES + i * 1024);
            }
        } else {
            // the quarter has landed when its last 16 bytes are no longer the poison pattern (LDS-DMA data returns in issue order)
            int spin = 0;
            while (*reinterpret_cast<const volatile unsigned*>(slot + QUARTER - 4) == POISON) {
                __builtin_amdgcn_s_sleep(1);
                if (++spin > (1 << 20)) {  // bounded: a broken protocol shows up as a wrong result and a note, never as a hung GPU
                    ctrl[5] = 1;
                    break;
                }
            }
        }
        f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, crs[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        const _Float16* b = Pin + m * LDP + kq * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const h8 bh = *reinterpret_cast<const h8*>(b + c * 32), bl = *reinterpret_cast<const h8*>(b + PLANE + c * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                h8 ah, al;
                if (MODE == 0) {
                    ah = cur[(t * 4 + c) * 2];
                    al = cur[(t * 4 + c) * 2 + 1];
                } else {
                    ah = *reinterpret_cast<const h8*>(slot + ((t * 4 + c) * 2) * 1024 + lane * 16);
                    al = *reinterpret_cast<const h8*>(slot + ((t * 4 + c) * 2 + 1) * 1024 + lane * 16);
                }
                acc[t] = mfma(ah, bh, acc[t]);
                crs[t] = mfma(ah, bl, crs[t]);
                crs[t] = mfma(al, bh, crs[t]);
            }
        }
        if (MODE == 1) {
            // release the slot: poison first, then the progress counter (same wave, LDS operations complete in order)
            if (lane == 63) *reinterpret_cast<volatile u4*>(const_cast<char*>(slot) + QUARTER - 16) = u4{POISON, POISON, POISON, POISON};
            if (lane == 0) ctrl[wave] = u + 1;
        } else {
#pragma unroll
            for (int i = 0; i < NFRAG; ++i) cur[i] = nxt[i];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f4 v = acc[t] + crs[t] * (1.0f / 2048.0f);
            sum += v;
            store_planes(Pout, 2 * wave + t, lane, v);
        }
        __syncthreads();
        if (gap_iters > 0) {  // a LayerNorm / attention phase: no weight traffic, ends with a barrier like every stage
            gapv = gap_work(gapv, gap_iters);
            __syncthreads();
        }
        _Float16* tmp = Pin;
        Pin = Pout;
        Pout = tmp;
    }
    if (MODE == 1) {
        if (tid == 0) ctrl[4] = 1;
        __syncthreads();  // the loader's last barrier
    }
    const long long t1 = clock64();
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = ctrl[5] ? -1 : t1 - t0;
    float* o = out + ((size_t)blockIdx.x * 256 + tid) * 4;
    o[0] = sum[0] + gapv * 1e-30f;
    o[1] = sum[1];
    o[2] = sum[2];
    o[3] = sum[3];
}

template <int MODE, int NSLOT>
static double run(const char* dW, int n_units, int gap, int n_wg, float* dout, long long* dcyc, std::vector<float>& hout) {
    const size_t lds = (MODE == 1 ? NSLOT * QUARTER : 0) + 4 * PLANE * 2 + 64;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain<MODE, NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_chain<MODE, NSLOT>), dim3(n_wg), dim3(MODE == 0 ? 256 : 320), lds, 0, dW, n_units, gap, dout, dcyc);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> c(n_wg);
    CHECK(hipMemcpy(c.data(), dcyc, n_wg * sizeof(long long), hipMemcpyDeviceToHost));
    hout.resize((size_t)n_wg * 256 * 4);
    CHECK(hipMemcpy(hout.data(), dout, hout.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    if (c[0] < 0) printf("  (MODE %d: a consumer gave up waiting for its slot in %d workgroups)\n", MODE, (int)std::count(c.begin(), c.end(), -1LL));
    // (one time unit of clock64() / s_memtime = one cycle of the 100 MHz reference clock x ... on this part: compare columns, not absolutes)
    return (double)c[n_wg / 2] / n_units;
}

int main(int argc, char** argv) {
    const int n_units = argc > 1 ? atoi(argv[1]) : 64, n_wg = argc > 2 ? atoi(argv[2]) : 128;
    std::vector<_Float16> hW((size_t)n_units * UNIT_BYTES / 2);
    unsigned s = 12345u;
    for (auto& w : hW) {
        s = s * 1664525u + 1013904223u;
        w = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-4f);
    }
    char* dW;
    float* dout;
    long long* dcyc;
    CHECK(hipMalloc((void**)&dW, hW.size() * 2));
    CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&dout, (size_t)n_wg * 256 * 4 * sizeof(float)));
    CHECK(hipMalloc((void**)&dcyc, n_wg * sizeof(long long)));
    printf("%d units of 64 KB (fp16 pairs: 24 MFMA 16x16x32 per wave and unit), %d workgroups, s_memtime ticks per unit (median workgroup)\n", n_units, n_wg);
    printf("%-12s %-34s %-34s %-34s\n", "gap iters", "MODE 0 register prefetch (4 waves)", "MODE 1 loader wave, 5 x 16 KB ring", "MODE 1 loader wave, 8 x 16 KB ring");
    for (int gap : {0, 12, 24, 48, 224}) {  // ~40 ticks per iteration: ~0 / 0.5 k / 1 k / 2 k / 9 k cycles (LayerNorm ~1-2 k, a 256-key walk ~9 k)
        std::vector<float> r0, r1, r2;
        const double c0 = run<0, 5>(dW, n_units, gap, n_wg, dout, dcyc, r0);
        const double c1 = run<1, 5>(dW, n_units, gap, n_wg, dout, dcyc, r1);
        const double c2 = run<1, 8>(dW, n_units, gap, n_wg, dout, dcyc, r2);
        const bool same = memcmp(r0.data(), r1.data(), r0.size() * 4) == 0 && memcmp(r0.data(), r2.data(), r0.size() * 4) == 0;
        printf("%-12d %-34.0f %-34.0f %-34.0f %s\n", gap, c0, c1, c2, same ? "bit-identical" : "RESULTS DIFFER");
    }
    return 0;
}
