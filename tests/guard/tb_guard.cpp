// Test infrastructure (tests/probes/gpu_guard_pages.py): device buffers with UNMAPPED memory on both sides, made with HIP's virtual
// memory management calls -- reserve (n + 2) granules of address space, back and map the n in the middle.  A kernel that reads or
// writes a byte in front of such a buffer (placed at the start of the mapping) or behind it (placed at its end) takes a GPU memory
// fault instead of touching a neighbour.  Not part of the product library.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>

static size_t g_gran = 0;

static int gran(hipMemAllocationProp* prop) {
    prop->type = hipMemAllocationTypePinned;
    prop->location.type = hipMemLocationTypeDevice;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    prop->location.id = dev;
    if (!g_gran && hipMemGetAllocationGranularity(&g_gran, prop, hipMemAllocationGranularityMinimum) != hipSuccess) return 1;
    return g_gran ? 0 : 1;
}

extern "C" size_t tbg_granularity() {
    hipMemAllocationProp prop = {};
    return gran(&prop) ? 0 : g_gran;
}

// returns the buffer (16-byte aligned; its first byte opens the mapping when at_end == 0, its last 16-byte unit closes it otherwise) or
// NULL; *base / *span describe the reservation for tbg_free
extern "C" void* tbg_alloc(size_t bytes, int at_end, void** base, size_t* span) {
    hipMemAllocationProp prop = {};
    if (gran(&prop)) return nullptr;
    const size_t need = bytes ? bytes : 16;
    const size_t n = (need + g_gran - 1) / g_gran;
    const size_t total = (n + 2) * g_gran;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, g_gran, nullptr, 0) != hipSuccess) return nullptr;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, n * g_gran, &prop, 0) != hipSuccess) {
        (void)hipMemAddressFree(va, total);
        return nullptr;
    }
    char* mid = static_cast<char*>(va) + g_gran;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemMap(mid, n * g_gran, 0, h, 0) != hipSuccess || hipMemSetAccess(mid, n * g_gran, &acc, 1) != hipSuccess) {
        (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, total);
        return nullptr;
    }
    (void)hipMemRelease(h);  // (the mapping keeps the memory alive)
    *base = va;
    *span = total;
    if (!at_end) return mid;
    const size_t padded = (need + 15) & ~(size_t)15;
    return mid + n * g_gran - padded;
}

extern "C" int tbg_free(void* base, size_t span) {
    if (!base || !g_gran) return 1;
    char* mid = static_cast<char*>(base) + g_gran;
    (void)hipMemUnmap(mid, span - 2 * g_gran);
    return hipMemAddressFree(base, span) == hipSuccess ? 0 : 1;
}
