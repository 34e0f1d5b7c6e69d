// Micro-benchmark that questions the structural floor of the fused step launch (VERDICT r05 task 5).
//
// Today every 16-agent tile workgroup STREAMS the 67 weight units of the step (4.26 MB as fp16 pairs) through its own CU's vector-load
// path once per step -- 35.8 us at the 42 B/clk four waves sustain (bench.py::structural_floor) -- while at the headline batch 128 of
// the 256 CUs idle.  Those idle CUs have 128 x 160 KB = 20 MB of LDS, five times the weight set.  The WEIGHT-STATIONARY alternative:
// each idle CU keeps ~2 weight units resident in LDS for the whole rollout and SERVES tiles: a tile workgroup hands its [16 x 128]
// activation tile (8 KB) to the server of unit u through L2 (payload, release fence, flag), the server multiplies it by the resident
// unit (the 64 x 3 MFMAs of one fp16-pair unit) and hands the [16 x 128] result back the same way.  67 such hops replace the stream, so
// a hop has to cost well under 35.8 us / 67 = 0.53 us for the exchange to pay (0.8 us is the break-even the verdict names, counting
// the cold start too).
//
// This program measures the hop: `pairs` client workgroups ping-pong with `pairs` server workgroups (client b <-> server b + stride:
// stride = 1 puts partners on DIFFERENT XCDs -- consecutive workgroup ids go round-robin over the 8 XCDs --, stride = 8 on the SAME XCD,
// i.e. one shared L2), payload 8 KB each way, with and without the server's MFMA work, under two protocols:
//   PROTO 0  the library's own hand-off (kv_flag / gh_flag in tb_device_xdl.hpp), made re-usable: the payload moves as 8-byte RELAXED
//            agent-scope atomic stores / loads (written through to and read at the coherence point: no stale L1 / L2 line can answer),
//            `s_waitcnt vmcnt(0)` + barrier, then a relaxed atomic flag; no cache-wide operation anywhere;
//   PROTO 1  ordinary stores / loads bracketed by release / acquire FENCES at agent scope (what the language-level memory model
//            emits: buffer_wbl2 / buffer_inv over the whole cache) -- the naive form, for the record.  Prints microseconds per hop (median / min / max over the pairs) and a
// JSON line for profiles/ws_hop.json.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microtests/ws_hop.hip -o tools/microtests/bin/ws_hop
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int TILE_FLOATS = 16 * 128;  // one [16 x 128] fp32 activation tile: 8 KB
constexpr int NT = 256;                // four waves, as the step kernel's tile workgroups

struct Mail {
    float* req;        // [pairs][TILE_FLOATS]
    float* rep;        // [pairs][TILE_FLOATS]
    unsigned* f_req;   // [pairs * 32] (one flag per 128-byte line)
    unsigned* f_rep;
    long long* ticks;  // [pairs] wall-clock ticks of the client's hop loop (100 MHz)
    float* sink;
    int pairs, stride, hops, mfma_per_hop, proto;
};

__device__ __forceinline__ void put8(float* dst, float a, float b) {
    unsigned long long u;
    const float v[2] = {a, b};
    __builtin_memcpy(&u, v, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void get8(const float* src, float& a, float& b) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float v[2];
    __builtin_memcpy(v, &u, 8);
    a = v[0];
    b = v[1];
}
// one thread's 32 bytes of a tile, by protocol
__device__ __forceinline__ void tile_put(float* base, int tid, const f4& x0, const f4& x1, int proto) {
    if (proto == 0) {
        put8(base + tid * 8, x0[0], x0[1]); put8(base + tid * 8 + 2, x0[2], x0[3]);
        put8(base + tid * 8 + 4, x1[0], x1[1]); put8(base + tid * 8 + 6, x1[2], x1[3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        *reinterpret_cast<f4*>(base + tid * 8) = x0;
        *reinterpret_cast<f4*>(base + tid * 8 + 4) = x1;
        __atomic_thread_fence(__ATOMIC_RELEASE);
    }
}
__device__ __forceinline__ void tile_get(const float* base, int tid, f4& x0, f4& x1, int proto) {
    if (proto == 0) {
        float a[8];
        get8(base + tid * 8, a[0], a[1]); get8(base + tid * 8 + 2, a[2], a[3]);
        get8(base + tid * 8 + 4, a[4], a[5]); get8(base + tid * 8 + 6, a[6], a[7]);
        x0 = f4{a[0], a[1], a[2], a[3]};
        x1 = f4{a[4], a[5], a[6], a[7]};
    } else {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        x0 = *reinterpret_cast<const f4*>(base + tid * 8);
        x1 = *reinterpret_cast<const f4*>(base + tid * 8 + 4);
    }
}

// bounded: a partner that never shows up (it cannot: all 2 x pairs workgroups are resident at once, one or two per CU) ends the
// wait after ~0.1 s instead of hanging the GPU; the timing of such a run is garbage and `gave_up` says so
__device__ unsigned gave_up;
__device__ __forceinline__ void spin_until(const unsigned* flag, unsigned want) {
    for (int i = 0; i < (1 << 21); ++i) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return;
        __builtin_amdgcn_s_sleep(1);
    }
    gave_up = 1u;
}

// role of a workgroup: within a group of 2 * stride consecutive ids the first `stride` are clients, the next `stride` their servers
__global__ __launch_bounds__(NT) void k_ws_hop(Mail m) {
    extern __shared__ _Float16 W[];  // the server's resident unit: 128 x 128 fp16 pairs = 64 KB (two planes of 32 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = blockIdx.x / (2 * m.stride), in = blockIdx.x % (2 * m.stride);
    const bool server = in >= m.stride;
    const int pair = grp * m.stride + (in % m.stride);
    if (pair >= m.pairs) return;
    float* req = m.req + (size_t)pair * TILE_FLOATS;
    float* rep = m.rep + (size_t)pair * TILE_FLOATS;
    unsigned* f_req = m.f_req + (size_t)pair * 32;
    unsigned* f_rep = m.f_rep + (size_t)pair * 32;
    if (server) {
        for (int i = tid; i < 128 * 128 * 2; i += NT) W[i] = (_Float16)(0.001f * (float)((i * 7 + pair) & 63));
        __syncthreads();
        f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int h = 1; h <= m.hops; ++h) {
            if (tid == 0) spin_until(f_req, (unsigned)h);
            __syncthreads();
            f4 x0, x1;
            tile_get(req, tid, x0, x1, m.proto);
            // the unit's products: A operands from the resident weights in LDS, B operand from the tile (stand-in split)
            h8 b;
            for (int i = 0; i < 4; ++i) { b[i] = (_Float16)x0[i]; b[4 + i] = (_Float16)x1[i]; }
            for (int k = 0; k < m.mfma_per_hop / 4; k += 2) {  // (`mfma_per_hop` is the WORKGROUP's count: a unit's 192 MFMAs are 48 per wave)
                const h8 a0 = *reinterpret_cast<const h8*>(W + ((k * 64 + lane) & 4095) * 8);
                const h8 a1 = *reinterpret_cast<const h8*>(W + (((k + 1) * 64 + lane) & 4095) * 8);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, acc[1], 0, 0, 0);
            }
            f4 y0 = x0 + acc[0] * 1e-6f, y1 = x1 + acc[1] * 1e-6f;
            tile_put(rep, tid, y0, y1, m.proto);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(f_rep, (unsigned)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) m.sink[pair] = acc[0][0] + acc[1][0];
    } else {
        f4 x0 = {1.f + tid, 2.f, 3.f, 4.f}, x1 = {5.f, 6.f, 7.f, 8.f};
        __syncthreads();
        const long long t0 = wall_clock64();
        for (int h = 1; h <= m.hops; ++h) {
            tile_put(req, tid, x0, x1, m.proto);
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(f_req, (unsigned)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin_until(f_rep, (unsigned)h);
            }
            __syncthreads();
            tile_get(rep, tid, x0, x1, m.proto);
        }
        const long long t1 = wall_clock64();
        if (tid == 0) {
            m.ticks[pair] = t1 - t0;
            m.sink[m.pairs + pair] = x0[0] + x1[3];
        }
    }
}

int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 128, hops = argc > 2 ? atoi(argv[2]) : 67 * 8;
    Mail m{};
    m.pairs = pairs;
    m.hops = hops;
    CHECK(hipMalloc((void**)&m.req, sizeof(float) * (size_t)pairs * TILE_FLOATS));
    CHECK(hipMalloc((void**)&m.rep, sizeof(float) * (size_t)pairs * TILE_FLOATS));
    CHECK(hipMalloc((void**)&m.f_req, sizeof(unsigned) * (size_t)pairs * 32));
    CHECK(hipMalloc((void**)&m.f_rep, sizeof(unsigned) * (size_t)pairs * 32));
    CHECK(hipMalloc((void**)&m.ticks, sizeof(long long) * pairs));
    CHECK(hipMalloc((void**)&m.sink, sizeof(float) * 2 * pairs));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ws_hop), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    int wall_khz = 100000;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("# ws_hop: %d client + %d server workgroups, %d hops each, payload 8 KB each way, flags release/acquire at agent scope; wall clock %d kHz\n",
           pairs, pairs, hops, wall_khz);
    printf("%-10s %-34s %-10s %10s %10s %10s\n", "protocol", "partners", "mfma/hop/wg", "us/hop med", "min", "max");
    double results[2][3];
    const int strides[2] = {1, 8};
    const int mf[3] = {0, 64, 192};
    for (int proto = 1; proto >= 0; --proto)  // (the library's protocol last: `results` keeps its numbers)
    for (int si = 0; si < 2; ++si)
        for (int mi = 0; mi < 3; ++mi) {
            m.proto = proto;
            m.stride = strides[si];
            m.mfma_per_hop = mf[mi];
            CHECK(hipMemset(m.f_req, 0, sizeof(unsigned) * (size_t)pairs * 32));
            CHECK(hipMemset(m.f_rep, 0, sizeof(unsigned) * (size_t)pairs * 32));
            const int groups = (pairs + m.stride - 1) / m.stride;
            hipLaunchKernelGGL(k_ws_hop, dim3(groups * 2 * m.stride), dim3(NT), 65536, 0, m);
            CHECK(hipDeviceSynchronize());
            unsigned gu = 0;
            CHECK(hipMemcpyFromSymbol(&gu, HIP_SYMBOL(gave_up), sizeof(gu)));
            if (gu) {
                printf("a wait gave up (partner not resident?): timings invalid\n");
                return 2;
            }
            std::vector<long long> t(pairs);
            CHECK(hipMemcpy(t.data(), m.ticks, sizeof(long long) * pairs, hipMemcpyDeviceToHost));
            std::sort(t.begin(), t.end());
            auto us = [&](long long v) { return (double)v / (double)wall_khz * 1e3 / hops; };
            results[si][mi] = us(t[pairs / 2]);
            printf("%-10s %-34s %-10d %10.3f %10.3f %10.3f\n", proto == 0 ? "atomics" : "fences", si == 0 ? "different XCDs (stride 1)" : "same XCD, one L2 (stride 8)", mf[mi], us(t[pairs / 2]),
                   us(t[0]), us(t[pairs - 1]));
        }
    const double stream_us = 35.8, per_hop_budget = stream_us / 67.0;
    printf("\n# budget: the weight stream a hop chain would replace is %.1f us per step = %.3f us per hop over 67 units\n", stream_us, per_hop_budget);
    printf("JSON {\"what\": \"weight-stationary variant of the GEMM chain: hop = tile hands [16x128] fp32 to a CU that keeps the unit in LDS, gets [16x128] back, through L2 with release/acquire flags (tools/microtests/ws_hop.hip)\", "
           "\"us_per_hop_same_xcd_with_unit_mfma\": %.3f, \"us_per_hop_other_xcd_with_unit_mfma\": %.3f, \"us_per_hop_same_xcd_no_compute\": %.3f, "
           "\"hops_per_step\": 67, \"floor_us_same_xcd\": %.1f, \"stream_us_it_replaces\": %.1f, \"break_even_us_per_hop\": %.3f, \"wins\": %s}\n",
           results[1][2], results[0][2], results[1][0], 67.0 * results[1][2], stream_us, per_hop_budget, results[1][2] < per_hop_budget ? "true" : "false");
    return 0;
}
