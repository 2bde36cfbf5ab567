// vmm_alloc.hip -- development helper of tools/vmm_probe.py (NOT part of the library): device memory through HIP's virtual
// memory management API (hipMemAddressReserve + hipMemCreate + hipMemMap) with a chosen physical chunk size, to see
// whether the placement classes of the key matrix follow the allocation path.
//   hipcc --offload-arch=gfx950 -shared -fPIC tools/vmm/vmm_alloc.hip -o tools/vmm/libvmm.so
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct VmmBuf {
    void *ptr;
    size_t bytes;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

extern "C" {

int vmm_granularity(int dev, size_t *gmin, size_t *grec)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    if (hipMemGetAllocationGranularity(gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) return -1;
    if (hipMemGetAllocationGranularity(grec, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return -2;
    return 0;
}

// bytes of device memory as ONE virtual range backed by physical chunks of `chunk` bytes each (chunk = 0: one chunk)
int vmm_alloc(int dev, size_t bytes, size_t chunk, size_t align, VmmBuf **out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return -1;
    if (chunk == 0) chunk = bytes;
    chunk = (chunk + gran - 1) / gran * gran;
    bytes = (bytes + chunk - 1) / chunk * chunk;
    VmmBuf *b = new VmmBuf();
    b->bytes = bytes;
    b->ptr = nullptr;
    if (hipMemAddressReserve(&b->ptr, bytes, align, nullptr, 0) != hipSuccess) { delete b; return -2; }
    for (size_t off = 0; off < bytes; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) { fprintf(stderr, "hipMemCreate(%zu): %s\n", chunk, hipGetErrorString(e)); return -3; }
        e = hipMemMap((char *)b->ptr + off, chunk, 0, h, 0);
        if (e != hipSuccess) { fprintf(stderr, "hipMemMap: %s\n", hipGetErrorString(e)); return -4; }
        b->handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipError_t e = hipMemSetAccess(b->ptr, bytes, &acc, 1);
    if (e != hipSuccess) { fprintf(stderr, "hipMemSetAccess: %s\n", hipGetErrorString(e)); return -5; }
    *out = b;
    return 0;
}

void *vmm_ptr(VmmBuf *b) { return b->ptr; }
size_t vmm_bytes(VmmBuf *b) { return b->bytes; }

void vmm_free(VmmBuf *b)
{
    if (!b) return;
    (void)hipMemUnmap(b->ptr, b->bytes);
    for (auto h : b->handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(b->ptr, b->bytes);
    delete b;
}

int plain_alloc(size_t bytes, void **out) { return hipMalloc(out, bytes) == hipSuccess ? 0 : -1; }
void plain_free(void *p) { (void)hipFree(p); }

}  // extern "C"
