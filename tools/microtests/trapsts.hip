// does TRAPSTS.EXCP accumulate float exceptions (sticky) on gfx950 without a trap handler?  v_cvt_f16_f32 of 1e6 -> overflow bit
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, _Float16* sink) {
    const unsigned before = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 3);
    const float x = in[threadIdx.x];
    const _Float16 h = (_Float16)x;
    sink[threadIdx.x] = h;
    const unsigned after = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 3);
    const float y = __builtin_amdgcn_exp2f(in[64 + threadIdx.x]);  // exp2(200) -> inf: overflow too
    sink[64 + threadIdx.x] = (_Float16)0.f + (_Float16)(y > 1e30f ? 1.f : 0.f);
    const unsigned after2 = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 3);
    const unsigned mode = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 1);
    if (threadIdx.x == 0) { out[0] = before; out[1] = after; out[2] = after2; out[3] = mode; }
}
int main() {
    float h[128];
    unsigned* out; float* in; _Float16* sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&in, sizeof(h)); (void)hipMalloc(&sink, 512);
    for (int variant = 0; variant < 3; ++variant) {
        for (int i = 0; i < 128; ++i) h[i] = 1.0f;
        if (variant == 1) h[5] = 1e6f;       // fp16 overflow in lane 5
        if (variant == 2) h[64 + 7] = 200.f; // exp2 overflow
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out, sink);
        unsigned o[4];
        (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
        printf("variant %d: TRAPSTS.EXCP before 0x%03x after cvt 0x%03x after exp2 0x%03x  MODE 0x%08x\n", variant, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
