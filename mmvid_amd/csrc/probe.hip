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
// which = 1: buffer_load_dwordx4 ... lds (LDS-DMA through a buffer descriptor).  in: int32[66] = 64 per-lane byte
// voffsets, then soffset, then num_records; the buffer is 4096 bytes of u32 value == its index + 1 (at in + 1024).
// LDS is pre-filled with 0xAAAAAAAA.  out: uint32[256] = the 1 KiB the DMA wrote (lane l -> words 4l..4l+3).
__global__ void probe_buffer_lds_kernel(const int* __restrict__ in, unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const int soff = in[64], nrec = in[65];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(in + 1024), 0, nrec, 0x00020000);
    typedef __attribute__((address_space(3))) void lds_void_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds, 16, in[lane], soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
}  // namespace

extern "C" int mmvid_probe(int which, const void* in, void* out, void* stream) {
    MMVID_REQUIRE((which == 0 || which == 1) && in && out, "probe: bad arguments");
    if (which == 1) {
        hipLaunchKernelGGL(probe_buffer_lds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)in, (unsigned*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    hipLaunchKernelGGL(probe_tr_b16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)in, (unsigned short*)out);
    MMVID_LAUNCH_CHECK("probe");
    return MMVID_OK;
}
