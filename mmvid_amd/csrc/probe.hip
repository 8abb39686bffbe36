// Hardware probes (lane layouts of gfx950 instructions whose semantics the kernels rely on).  Not on the
// product path; used by tools/gpu_probe.py to confirm assumptions before a kernel is built on them.
#include "common.h"

namespace {
// which = 0: ds_read_b64_tr_b16.  in: int32[64] per-lane LDS byte offsets.  LDS is filled with u16 value == its
// u16 index.  out: uint16[64][4] what each lane received.
__global__ void probe_tr_b16_kernel(const int* __restrict__ offs, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    typedef __attribute__((address_space(3))) unsigned short lds_u16;
    unsigned addr = (unsigned)(uintptr_t)(lds_u16*)lds + (unsigned)offs[lane];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (unsigned short)(v.x & 0xffff);
    out[lane * 4 + 1] = (unsigned short)(v.x >> 16);
    out[lane * 4 + 2] = (unsigned short)(v.y & 0xffff);
    out[lane * 4 + 3] = (unsigned short)(v.y >> 16);
}
}  // namespace

extern "C" int mmvid_probe(int which, const void* in, void* out, void* stream) {
    MMVID_REQUIRE(which == 0 && in && out, "probe: bad arguments");
    hipLaunchKernelGGL(probe_tr_b16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)in, (unsigned short*)out);
    MMVID_LAUNCH_CHECK("probe");
    return MMVID_OK;
}
