// Optimiser step over flat fp32 buffers (Adam + clip_grad_norm_, train.py:322-325,
// utils_train.py:167-172).  HBM-bound: 28 B read+written per parameter, one launch for the whole model.
//   grad_sqnorm : out[0] += sum g^2   (two-level reduction, fp32 atomics per block)
//   adam_step   : torch.optim.Adam semantics (no amsgrad, weight_decay optional as L2 added to grad):
//                 g *= clip_coef, clip_coef = min(1, max_norm / (sqrt(sqnorm) + 1e-6))   (clip_grad_norm_)
//                 m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//                 p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
//                 also refreshes the bf16 shadow copy the MFMA kernels read.
#include "../../include/mmvid_hip.h"
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, long n,
                                                          float* __restrict__ out) {
    float a = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (long k = i; k < n; ++k) a += g[k] * g[k];
        }
    }
    a = wave_sum(a);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (sh[0] + sh[1]) + (sh[2] + sh[3]));
}

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, max_norm, grad_scale;
};
// "Lazy rows": one table inside the flat buffers (elements [lo, lo + rows * rowlen)) whose rows carry a flag "has ever received
// a gradient".  A row that never did has g = m = v = 0, so Adam (without weight decay) leaves p, m, v unchanged and its g^2 is 0:
// skipping it is EXACT, and saves all 30 B per parameter.  The text embedding is 49,472 x 768 = 30 % of BERT's parameters and
// a step touches at most 3 * B * 64 of its rows (the engine sets the flags from the ids the model logged).
struct LazyRows {
    const unsigned char* flags;  // null: no lazy table
    long lo, hi;                 // element range of the table in the flat buffer
    int rowlen;
};
__device__ __forceinline__ bool lazy_skip(const LazyRows& z, long i) {
    return z.flags && i >= z.lo && i < z.hi && z.flags[(i - z.lo) / z.rowlen] == 0;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   bf16_t* __restrict__ shadow, long n, AdamArgs a,
                                                   const float* __restrict__ sqnorm,
                                                   const float* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev, LazyRows z) {
    if (lr_dev) a.lr = *lr_dev;  // learning rate kept on the device (mmvid_lr_schedule): a captured step follows the schedule
    if (step_dev) {  // step count kept on the device (whole-step graph replay): bias corrections computed here
        const float t = *step_dev;
        a.bc1 = 1.0f - powf(a.beta1, t);
        a.bc2_sqrt = sqrtf(1.0f - powf(a.beta2, t));
    }
    float coef = a.grad_scale;
    if (sqnorm && a.max_norm > 0.f) {
        const float c = a.max_norm / (sqrtf(*sqnorm) * a.grad_scale + 1e-6f);
        coef *= c < 1.f ? c : 1.f;
    }
    const float step = a.lr / a.bc1;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (lazy_skip(z, i)) continue;  // (rowlen % 4 == 0 and lo % 4 == 0: the four elements share a row)
        if (i + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float pe[4] = {pp.x, pp.y, pp.z, pp.w}, ge[4] = {gg.x, gg.y, gg.z, gg.w};
            float me[4] = {mm.x, mm.y, mm.z, mm.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gr = ge[e] * coef + a.weight_decay * pe[e];
                me[e] = a.beta1 * me[e] + (1.f - a.beta1) * gr;
                ve[e] = a.beta2 * ve[e] + (1.f - a.beta2) * gr * gr;
                pe[e] -= step * me[e] / (sqrtf(ve[e]) / a.bc2_sqrt + a.eps);
            }
            *reinterpret_cast<float4*>(p + i) = make_float4(pe[0], pe[1], pe[2], pe[3]);
            *reinterpret_cast<float4*>(m + i) = make_float4(me[0], me[1], me[2], me[3]);
            *reinterpret_cast<float4*>(v + i) = make_float4(ve[0], ve[1], ve[2], ve[3]);
            if (shadow) *reinterpret_cast<uint2*>(shadow + i) = make_uint2(pack_bf2(pe[0], pe[1]), pack_bf2(pe[2], pe[3]));
        } else {
            for (long k = i; k < n; ++k) {
                float gr = g[k] * coef + a.weight_decay * p[k];
                float mk = a.beta1 * m[k] + (1.f - a.beta1) * gr;
                float vk = a.beta2 * v[k] + (1.f - a.beta2) * gr * gr;
                float pk = p[k] - step * mk / (sqrtf(vk) / a.bc2_sqrt + a.eps);
                p[k] = pk, m[k] = mk, v[k] = vk;
                if (shadow) shadow[k] = f2bf(pk);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            *reinterpret_cast<uint2*>(y + i) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        } else {
            for (long k = i; k < n; ++k) y[k] = f2bf(x[k]);
        }
    }
}

static int grid_for(long n) {
    long b = (n + 1023) / 1024;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mmvid_grad_sqnorm(const float* g, int64_t n, float* out_accum, void* stream) {
    MMVID_REQUIRE(g && out_accum && n >= 0, "grad_sqnorm: bad arguments");
    MMVID_REQUIRE(((uintptr_t)g & 15) == 0, "grad_sqnorm: buffer must be 16-byte aligned");
    if (n == 0) return MMVID_OK;
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, (long)n, out_accum);
    MMVID_LAUNCH_CHECK("grad_sqnorm");
    return MMVID_OK;
}

// fixed-order version of grad_sqnorm: per-block partial sums, then one block adds them in index order (deterministic:
// the clip coefficient of a step no longer depends on atomic arrival order)
__global__ __launch_bounds__(256) void grad_sqnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part,
                                                                  LazyRows z) {
    float a = 0.f;
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (lazy_skip(z, i)) continue;
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            a += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (long k = i; k < n; ++k) a += g[k] * g[k];
        }
    }
    a = wave_sum(a);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
    __shared__ float sh[256];
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] += sh[0];
}

static int make_lazy(LazyRows& z, const uint8_t* flags, int64_t lo, int64_t rows, int rowlen, int64_t n) {
    z.flags = nullptr, z.lo = z.hi = 0, z.rowlen = 1;
    if (!flags) return MMVID_OK;
    MMVID_REQUIRE(lo >= 0 && rows >= 0 && rowlen > 0 && rowlen % 4 == 0 && lo % 4 == 0 && lo + rows * rowlen <= n,
                  "lazy rows: the table [%lld, +%lld x %d) must lie inside the buffer and be 4-element aligned", (long long)lo,
                  (long long)rows, rowlen);
    z.flags = flags, z.lo = lo, z.hi = lo + rows * rowlen, z.rowlen = rowlen;
    return MMVID_OK;
}

extern "C" int mmvid_grad_sqnorm_det(const float* g, int64_t n, float* partials, float* out_accum, void* stream) {
    return mmvid_grad_sqnorm_rows(g, n, partials, out_accum, nullptr, 0, 0, 0, stream);
}

extern "C" int mmvid_grad_sqnorm_rows(const float* g, int64_t n, float* partials, float* out_accum, const uint8_t* row_flags,
                                      int64_t table_lo, int64_t table_rows, int rowlen, void* stream) {
    MMVID_REQUIRE(g && partials && out_accum && n >= 0, "grad_sqnorm_det: bad arguments");
    MMVID_REQUIRE(((uintptr_t)g & 15) == 0, "grad_sqnorm_det: buffer must be 16-byte aligned");
    if (n == 0) return MMVID_OK;
    LazyRows z;
    if (int rc = make_lazy(z, row_flags, table_lo, table_rows, rowlen, n)) return rc;
    const int nb = grid_for(n);  // <= 2048: `partials` holds 2048 floats
    hipLaunchKernelGGL(grad_sqnorm_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, (long)n, partials, z);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, out_accum);
    MMVID_LAUNCH_CHECK("grad_sqnorm_det");
    return MMVID_OK;
}

extern "C" int mmvid_adam_step_lr(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                                  const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                                  const float* step_dev, float max_norm, const float* sqnorm, float grad_scale, void* stream);

extern "C" int mmvid_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int step,
                               const float* step_dev, float max_norm, const float* sqnorm, float grad_scale,
                               void* stream) {
    return mmvid_adam_step_lr(p, g, m, v, shadow_bf16, n, lr, nullptr, beta1, beta2, eps, weight_decay, step, step_dev, max_norm,
                              sqnorm, grad_scale, stream);
}

extern "C" int mmvid_adam_step_lr(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                                  const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                                  const float* step_dev, float max_norm, const float* sqnorm, float grad_scale, void* stream) {
    return mmvid_adam_step_rows(p, g, m, v, shadow_bf16, n, lr, lr_dev, beta1, beta2, eps, weight_decay, step, step_dev, max_norm,
                                sqnorm, grad_scale, nullptr, 0, 0, 0, stream);
}

extern "C" int mmvid_adam_step_rows(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                                    const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                                    const float* step_dev, float max_norm, const float* sqnorm, float grad_scale,
                                    const uint8_t* row_flags, int64_t table_lo, int64_t table_rows, int rowlen, void* stream) {
    MMVID_REQUIRE(p && g && m && v && n >= 0 && (step >= 1 || step_dev), "adam_step: bad arguments");
    MMVID_REQUIRE(!row_flags || weight_decay == 0.f, "adam_step: lazy rows are exact only without weight decay");
    LazyRows z;
    if (int rc = make_lazy(z, row_flags, table_lo, table_rows, rowlen, n)) return rc;
    MMVID_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)shadow_bf16 & 7) == 0,
                  "adam_step: buffers must be 16-byte aligned");
    if (n == 0) return MMVID_OK;
    AdamArgs a;
    a.lr = lr, a.beta1 = beta1, a.beta2 = beta2, a.eps = eps, a.weight_decay = weight_decay;
    a.bc1 = 1.0f - powf(beta1, (float)(step >= 1 ? step : 1));
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)(step >= 1 ? step : 1)));
    a.max_norm = max_norm, a.grad_scale = grad_scale;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)shadow_bf16,
                       (long)n, a, sqnorm, step_dev, lr_dev, z);
    MMVID_LAUNCH_CHECK("adam_step");
    return MMVID_OK;
}

extern "C" int mmvid_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
    MMVID_REQUIRE(x && y && n >= 0, "cast_f32_to_bf16: bad arguments");
    MMVID_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0, "cast_f32_to_bf16: alignment");
    if (n == 0) return MMVID_OK;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, (long)n);
    MMVID_LAUNCH_CHECK("cast_f32_to_bf16");
    return MMVID_OK;
}
