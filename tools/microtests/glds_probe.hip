// probe: does global_load_lds_dwordx4 land where the loader-ring micro-benchmark expects?  (asm form of the guide vs the builtin;
// LDS destinations below and above 64 KB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int VAR>
__global__ __launch_bounds__(128) void k(const unsigned* src, unsigned* out, unsigned lds_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 40960; i += 128) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;  // 160 KB
    __syncthreads();
    if (wave == 1) {
        const char* g = reinterpret_cast<const char*>(src) + lane * 16;
        if (VAR == 0) glds16_asm(g, __builtin_amdgcn_readfirstlane(lds_off));
        else __builtin_amdgcn_global_load_lds((const u4*)g, (__attribute__((address_space(3))) u4*)(smem + lds_off), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // where did 1 KB land?  report the first LDS dword index that is no longer the fill pattern, and the dword at lds_off + 16 * 5
    if (tid == 0) {
        int first = -1, count = 0;
        for (int i = 0; i < 40960; ++i) {
            const unsigned v = reinterpret_cast<volatile unsigned*>(smem)[i];
            if (v != 0xDEADBEEFu) { if (first < 0) first = i; ++count; }
        }
        out[0] = (unsigned)first; out[1] = (unsigned)count;
        out[2] = reinterpret_cast<volatile unsigned*>(smem)[(lds_off + 16 * 5) / 4];
        out[3] = reinterpret_cast<volatile unsigned*>(smem)[(lds_off + 1020) / 4];
    }
}

int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    hipMalloc((void**)&d, 1024); hipMalloc((void**)&o, 64);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int var = 0; var < 2; ++var)
        for (unsigned off : {0u, 16384u, 65536u, 81920u, 147456u}) {
            hipMemset(o, 0, 64);
            if (var == 0) { hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); hipLaunchKernelGGL(k<0>, dim3(1), dim3(128), 163840, 0, d, o, off); }
            else { hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840); hipLaunchKernelGGL(k<1>, dim3(1), dim3(128), 163840, 0, d, o, off); }
            hipError_t e = hipDeviceSynchronize();
            unsigned r[4];
            hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
            printf("%s dst offset %6u: first changed dword %d (expected %u), %u dwords changed (expected 256), lane-5 dword %u (expected %u), last dword %u (expected 1255)  %s\n",
                   var == 0 ? "asm    " : "builtin", off, (int)r[0], off / 4, r[1], r[2], 1000 + 20, r[3], hipGetErrorString(e));
        }
    return 0;
}
