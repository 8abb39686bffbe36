// Hardware probes (lane layouts of gfx950 instructions whose semantics the kernels rely on).  Not on the
// product path; used by tools/gpu_probe.py to confirm assumptions before a kernel is built on them.
#include "common.h"
#include "gemm_core.h"

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
// which = 2: store-pattern bandwidth.  Writes a bf16 [M][N] matrix (values irrelevant) the way a GEMM epilogue would.
//   variant 0: 256x128 tiles, 512 threads, 8 B per lane, 256-B row segments, rows in the epilogue's slab order
//   variant 1: 256x128 tiles, 16 B per lane (16 lanes per 256-B segment), rows in natural order
//   variant 2: as 0 but rows in natural order
//   variant 3: 128x256 tiles, 16 B per lane (512-B segments)
//   variant 4: linear (every block writes one contiguous 64-KiB run)
__global__ __launch_bounds__(512) void probe_store_kernel(int M, int N, int variant, unsigned short* __restrict__ out) {
    const int t = threadIdx.x;
    const uint2 v2 = make_uint2(0x3f803f80u, 0x3f803f80u);
    const uint4 v4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    if (variant == 4) {
        const long base = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 32768;  // elements per block
        for (int k = 0; k < 8; ++k) {
            const long e = base + ((long)k * 512 + t) * 8;
            if (e + 8 <= (long)M * N) *reinterpret_cast<uint4*>(out + e) = v4;
        }
        return;
    }
    if (variant == 3) {
        const int bn0 = blockIdx.x * 256, bm0 = blockIdx.y * 128;
        for (int k = 0; k < 8; ++k) {
            const int idx = t + 512 * k, r = idx >> 5, c = (idx & 31) * 8;
            if (bm0 + r < M && bn0 + c < N) *reinterpret_cast<uint4*>(out + (long)(bm0 + r) * N + bn0 + c) = v4;
        }
        return;
    }
    const int bn0 = blockIdx.x * 128, bm0 = blockIdx.y * 256;
    if (variant == 1) {
        for (int k = 0; k < 8; ++k) {
            const int idx = t + 512 * k, r = idx >> 4, c = (idx & 15) * 8;
            if (bm0 + r < M && bn0 + c < N) *reinterpret_cast<uint4*>(out + (long)(bm0 + r) * N + bn0 + c) = v4;
        }
        return;
    }
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 8; ++k) {
            const int idx = t + 512 * k, r = idx >> 5, c = (idx & 31) * 4;
            const int ml = variant == 0 ? (r >> 5) * 64 + i * 32 + (r & 31) : i * 128 + r;
            if (bm0 + ml < M && bn0 + c < N) *reinterpret_cast<uint2*>(out + (long)(bm0 + ml) * N + bn0 + c) = v2;
        }
}
// which = 3: the DPP / permlane wave reductions of common.h.  in: float[64]; out: float[3][64] = wave_sum_fast, wave_max_fast and
// wave_sum (the ds_bpermute form) as every lane sees them.
__global__ void probe_wave_reduce_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[lane] = wave_sum_fast(v);
    out[64 + lane] = wave_max_fast(v);
    out[128 + lane] = wave_sum(v);
}
// which = 4: LDS read throughput of the attention kernels' access patterns.  One block of 1024 threads (16 waves) on one CU reads a
// [64 rows][128 B] tile 2048 times per wave with pattern in[0]:
//   0  ds_read_b128, lane -> 16 consecutive bytes (the ideal)          1  ds_read_b128, row = lane&31, chunk = (2 ks + h) ^ ((row>>1)&7)
//   2  ds_read_b128, row = lane&31, chunk = (2 ks + h) ^ (row & 7)      3  ds_read_b64_tr_b16 as attn.hip's tr_frag issues it
//   4  ds_read_b64, lane -> 8 consecutive bytes (the ideal for 3)
// out: uint64[4] = shader clocks per wave; the host derives bytes per clock.
__global__ __launch_bounds__(1024) void probe_lds_pattern_kernel(const int* __restrict__ in, unsigned long long* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char tile[2 * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += (int)blockDim.x) reinterpret_cast<int*>(tile)[i] = i;
    __syncthreads();
    const int pat = in[0];
    const int l32 = lane & 31, h = lane >> 5;
    typedef __attribute__((address_space(3))) char lds_c;
    const unsigned base = (unsigned)(uintptr_t)(lds_c*)tile;
    unsigned addr[4];
    for (int ks = 0; ks < 4; ++ks) {
        if (pat == 0) addr[ks] = base + ks * 1024 + lane * 16;
        else if (pat == 1) addr[ks] = base + l32 * 128 + (((2 * ks + h) ^ ((l32 >> 1) & 7)) << 4);
        else if (pat == 2) addr[ks] = base + l32 * 128 + (((2 * ks + h) ^ (l32 & 7)) << 4);
        else if (pat == 4) addr[ks] = base + ks * 512 + lane * 8;
        else {
            const int G = lane >> 4, si = lane & 15;
            const int rowl = 4 * (G >> 1) + (si >> 2), cl = 2 * (G & 1) + ((si & 3) >> 1);
            addr[ks] = base + rowl * 128 + ((cl ^ (rowl >> 1)) << 4) + 8 * (si & 1) + ks * 2048;
        }
    }
    uint4 acc = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < 512; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (pat <= 2) {
                uint4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr[ks]) : "memory");
            } else if (pat == 3) {
                uint2 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr[ks]) : "memory");
            } else {
                uint2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[ks]) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = clock64();
    if (lane == 0 && wave < 4) out[wave] = t1 - t0;
    if (acc.x == 0x12345678u) out[4] = acc.x;
}
// which = 5: the matrix pipe's SUSTAINED ceiling (tools/power_probe.py).  Register-only v_mfma_f32_32x32x16_bf16 chains (4 accumulators, a
// pool of 8 operand fragments cycled so that consecutive MFMAs see different bits), 8 waves per block, no memory traffic at all.
// in (HOST int32[3]) = {iterations of 32 MFMAs per wave, blocks, operand mode: 0 zeros, 1 random bits with bf16 exponents near 1.0}.
__global__ __launch_bounds__(512) void probe_mfma_kernel(int iters, int mode, float* __restrict__ out) {
    const unsigned tid = threadIdx.x + blockIdx.x * 512u;
    if ((mode & 0x100) && threadIdx.x >= 256) return;  // mode bit 8: ONE wave per SIMD (four waves per block) -- can a single wave keep the pipe full?
    bf16x8_t f[8];
    unsigned h = tid * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h = h * 1664525u + 1013904223u;
            // two bf16 per word: random sign + mantissa, exponent 0x3f (|x| in [0.5, 2)) -- the bit statistics of normalised activations
            w[e] = mode ? ((h & 0x80ff80ffu) | 0x3f003f00u) : 0u;
        }
        f[i] = __builtin_bit_cast(bf16x8_t, uint4{w[0], w[1], w[2], w[3]});
    }
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // mode bits 4-7 = R: R fragment reads (ds_read_b128, 1 KiB per wave) per group of 4 MFMAs, from 64 KiB of LDS filled with the same
    // kind of bits -- what a GEMM wave does besides its MFMAs (R = 4: a 2 x 2 register tile, 1 KiB per MFMA; 3: a 4 x 2 tile; 2: 4 x 4).
    // Is the energy of the operand reads a visible part of the power budget?  (tools/power_probe.py)
    __shared__ __attribute__((aligned(16))) unsigned ldsbuf[16384];
    const int R = (mode >> 4) & 15;
    if (R) {
        unsigned g = tid * 747796405u + 2891336453u;
        for (int i = threadIdx.x; i < 16384; i += ((mode & 0x100) ? 256 : 512)) {
            g = g * 1664525u + 1013904223u;
            ldsbuf[i] = (mode & 1) ? ((g & 0x80ff80ffu) | 0x3f003f00u) : 0u;
        }
        __syncthreads();
    }
    const int lane16 = (threadIdx.x & 63) * 4;  // this lane's 16 bytes inside a 1-KiB fragment
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (R) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < R) {
                        const int frag = (it * 29 + u * 4 + r) & 63;  // 64 fragments of 1 KiB
                        f[(u * 4 + r) & 7] = *reinterpret_cast<const bf16x8_t*>(&ldsbuf[frag * 256 + lane16]);
                    }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(u + a) & 7], f[(u + 2 * a + 3) & 7], acc[a], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    if (sum == 1.2345e-30f) out[0] = sum;
}
// which = 6: an LDS-DMA read stream with a known byte count, for calibrating rocprofv3's FETCH_SIZE on this access path (the guide says
// FETCH_SIZE reports HALF the bytes of a wide coalesced read on gfx950, `global_load` and `buffer_load ... lds` alike: profiles/
// r04_pmc_fetch_calibration.txt).  in (HOST int64[2]) = {bytes (multiple of 64 KiB), mode: 0 = buffer_load_dwordx4 ... lds, 1 = global_load_dwordx4
// into registers}; `out` is the buffer that is read (every byte exactly once, 1 KiB per wave-instruction, 256 threads per block).
__global__ __launch_bounds__(256) void probe_stream_kernel(const char* __restrict__ buf, long bytes, int mode, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long per_block = 65536;
    const char* base = buf + (long)blockIdx.x * per_block;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (mode == 0) {
        const mmvid_core::rsrc_t r = mmvid_core::make_rsrc(base, (uint32_t)per_block);
        for (int i = wave; i < 64; i += 4) mmvid_core::blds16(r, (uint32_t)(i * 1024 + lane * 16), 0, lds[wave]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<const uint4*>(lds[wave] + lane * 16);
    } else {
        for (int i = wave; i < 64; i += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + i * 1024 + lane * 16);
            acc.x ^= v.x, acc.y ^= v.y, acc.z ^= v.z, acc.w ^= v.w;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1.f;
    (void)bytes;
}
// which = 7: the dependency chain of a persistent decode kernel.  `blocks` co-resident blocks run `phases` phases; in a phase every block
// POLLS a row of K tagged words written by the other blocks in the previous phase (8 bytes = fp32 payload | 32-bit phase tag, agent-scope
// relaxed atomics: the tag makes the data its own flag, no separate barrier), sums it, and writes its own share of the next row.
// mode 1: a counter barrier (atomic add + poll) followed by plain agent-scope loads of the row instead.  Time / phases = the floor of one
// phase of a megakernel decode step.  in = HOST int32[4]: phases, blocks, K, mode; out = device uint64[2 * K + 64] zeroed by the caller.
// Spins are bounded (a lost block ends the kernel with out[2K + 1] = 1 instead of hanging the device).
__global__ __launch_bounds__(256) void probe_chain_kernel(int phases, int K, int mode, unsigned long long* buf) {
    unsigned long long* rows[2] = {buf, buf + K};
    unsigned int* counter = reinterpret_cast<unsigned int*>(buf + 2 * K);
    unsigned long long* fail = buf + 2 * K + 1;
    __shared__ float red[4];
    const int tid = threadIdx.x, nb = gridDim.x;
    const int per = (K + nb - 1) / nb;  // words a block writes per phase
    float carry = 1.0f;
    for (int p = 1; p <= phases; ++p) {
        unsigned long long* dst = rows[p & 1];
        const unsigned long long* src = rows[(p + 1) & 1];
        // write this block's share (phase 1 needs no input)
        // mode bit 1 (value 2): the words of a block are INTERLEAVED with the other blocks' (word = tid * blocks + block: every 64-byte
        // line has eight writers on eight XCDs) instead of contiguous
        const int widx = (mode & 2) ? tid * nb + (int)blockIdx.x : (int)blockIdx.x * per + tid;
        if (tid < per && widx < K) {
            const unsigned long long word = ((unsigned long long)(unsigned)p << 32) | __float_as_uint(carry * 0.5f + (float)tid);
            __hip_atomic_store(dst + widx, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mode & 1) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(p * nb)) {
                    if (++spins > (1 << 22)) {
                        *fail = 1;
                        break;
                    }
                }
            }
            __syncthreads();
        }
        float s = 0.f;
        if (mode & 4) {  // mode bit 2 (value 4): ONE wave polls the row, 12 words per lane, all requested before any is examined
            if (tid < 64) {
                unsigned long long w[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) w[i] = (i * 64 + tid < K) ? __hip_atomic_load(dst + i * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)(unsigned)p << 32);
                for (int spins = 0;; ++spins) {
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        if ((unsigned)(w[i] >> 32) != (unsigned)p) ok = false, w[i] = __hip_atomic_load(dst + i * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ok) break;
                    if (spins > (1 << 20)) {
                        *fail = 1;
                        break;
                    }
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) s += __uint_as_float((unsigned)w[i]);
            }
        } else
        for (int k = tid; k < K; k += 256) {
            unsigned long long w = __hip_atomic_load(dst + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(mode & 1)) {
                int spins = 0;
                while ((unsigned)(w >> 32) != (unsigned)p) {
                    if (++spins > (1 << 22)) {
                        *fail = 1;
                        break;
                    }
                    w = __hip_atomic_load(dst + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            s += __uint_as_float((unsigned)w);
        }
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        carry = ((red[0] + red[1]) + (red[2] + red[3])) * 1e-6f;
        __syncthreads();
        (void)src;
    }
    if (carry == 1.2345e-30f) buf[2 * K + 2] = 1;
}
}  // namespace

extern "C" int mmvid_probe(int which, const void* in, void* out, void* stream) {
    MMVID_REQUIRE((which >= 0 && which <= 7) && in && out, "probe: bad arguments");
    if (which == 7) {  // in = HOST int32[4]: phases, blocks, K, mode
        const int* a = (const int*)in;
        MMVID_REQUIRE(a[1] > 0 && a[1] <= 256 && a[2] > 0, "probe 7: at most 256 (co-resident) blocks");
        hipLaunchKernelGGL(probe_chain_kernel, dim3(a[1]), dim3(256), 0, (hipStream_t)stream, a[0], a[2], a[3], (unsigned long long*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 6) {  // in = HOST int64[2]: bytes, mode; out = the device buffer that is streamed (its first float is the sink)
        const long long* a = (const long long*)in;
        hipLaunchKernelGGL(probe_stream_kernel, dim3((unsigned)(a[0] / 65536)), dim3(256), 0, (hipStream_t)stream, (const char*)out, (long)a[0],
                           (int)a[1], (float*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 5) {  // in = HOST int32[3]: iterations, blocks, operand mode
        const int* a = (const int*)in;
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(a[1]), dim3(512), 0, (hipStream_t)stream, a[0], a[2], (float*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 4) {
        hipLaunchKernelGGL(probe_lds_pattern_kernel, dim3(1), dim3(((const int*)in == nullptr) ? 256 : 1024), 0, (hipStream_t)stream, (const int*)in, (unsigned long long*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 3) {
        hipLaunchKernelGGL(probe_wave_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)in, (float*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 2) {  // in = HOST int32[3]: M, N, variant
        const int* a = (const int*)in;
        const int M = a[0], N = a[1], variant = a[2];
        dim3 grid = variant == 3 ? dim3(cdiv(N, 256), cdiv(M, 128)) : dim3(cdiv(N, 128), cdiv(M, 256));
        hipLaunchKernelGGL(probe_store_kernel, grid, dim3(512), 0, (hipStream_t)stream, M, N, variant, (unsigned short*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    if (which == 1) {
        hipLaunchKernelGGL(probe_buffer_lds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)in, (unsigned*)out);
        MMVID_LAUNCH_CHECK("probe");
        return MMVID_OK;
    }
    hipLaunchKernelGGL(probe_tr_b16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)in, (unsigned short*)out);
    MMVID_LAUNCH_CHECK("probe");
    return MMVID_OK;
}
