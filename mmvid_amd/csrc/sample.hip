// Device-side samplers of the path (gfx950): BERT's mask-predict loop (mmvid_pytorch/dalle_bert.py:514-714) and the
// ART-V token sampler (dalle_artv.py:61-67, 274-281), kept on the GPU end to end -- no .item(), no torch.multinomial.
//
// Sampling rule.  torch.multinomial draws category c with probability p_c by an exponential race (q_c ~ Exp(1),
// argmax p_c / q_c), with or without replacement.  The kernels here take the race variates as INPUT tensors:
//   token draw      tok = first argmin_c  E_c / P_c,   P_c = expf(x_c - max x),  Y = P_tok / sum_c P_c
//   keep selection  keep the k positions with the smallest  E_i / Y_i  (ties: lower index)  -- sampling k of the valid
//                   positions without replacement with weights Y, dalle_bert.py:651-657
// so the host fills E with torch's exponential_() in production and the tests inject the same E into oracle/sampling.py
// and require identical decisions.  E / P is one correctly-rounded fp32 division on both sides.
// All kernels are HBM/latency-bound row kernels: one wave per row, coalesced 4-B/16-B accesses.
#include "../../include/mmvid_hip.h"
#include "common.h"

namespace {

__device__ __forceinline__ float gumbel_from_u(float u) { return -logf(-logf(u + 1e-20f) + 1e-20f); }  // dalle_bert.py:536-538

struct Best {
    float key;
    int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
    return (b.key < a.key || (b.key == a.key && b.idx < a.idx)) ? b : a;
}

// one wave per row
__global__ __launch_bounds__(256) void sample_race_kernel(const float* __restrict__ logits, long ld,
                                                          const float* __restrict__ E, const float* __restrict__ noise_u,
                                                          float temperature, float inv_temp_div, long R, int V,
                                                          long long tok_offset, long long* __restrict__ tok_out,
                                                          float* __restrict__ y_out) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float* x = logits + r * ld;
    const float* e = E + r * (long)V;
    const float* g = noise_u ? noise_u + r * (long)V : nullptr;
    auto xv = [&](int c) -> float {
        float v = x[c] * inv_temp_div;  // ART-V divides the logits by the temperature (dalle_artv.py:275)
        if (g) v += temperature * gumbel_from_u(g[c]);  // BERT adds temperature-scaled Gumbel noise (dalle_bert.py:528)
        return v;
    };
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, xv(c));
    mx = wave_max(mx);
    float s = 0.f;
    Best b = {INFINITY, 0x7fffffff};
    for (int c = lane; c < V; c += 64) {
        const float p = expf(xv(c) - mx);
        s += p;
        const float key = p > 0.f ? e[c] / p : INFINITY;
        b = better(b, Best{key, c});
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best ob;
        ob.key = __shfl_xor(b.key, o, 64);
        ob.idx = __shfl_xor(b.idx, o, 64);
        b = better(b, ob);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const int tok = b.idx < V ? b.idx : 0;
        tok_out[r] = (long long)tok + tok_offset;
        if (y_out) y_out[r] = expf(xv(tok) - mx) / s;
    }
}

// The token draw alone (y_out not wanted: the ART-V sampler), one BLOCK per row: a thread's elements are requested together (the wave-per-row
// loop is a chain of 2 x V / 64 dependent round trips: 13 us per token at V = 1,024, a twentieth of a batch-1 decode step).  The same keys
// E_c / expf(x_c - max) and the same (key, index) order: the token is the wave-per-row kernel's, bit for bit.  E is the row block of draw
// number (*step_dev - step0) when step_dev is given (the variates of a whole sampling loop are drawn at once, outside the captured step).
__global__ __launch_bounds__(256) void sample_race_row_kernel(const float* __restrict__ logits, long ld, const float* __restrict__ E,
                                                              const int* __restrict__ step_dev, int step0, long e_step_stride,
                                                              float inv_temp_div, int V, long long tok_offset,
                                                              long long* __restrict__ tok_out) {
    __shared__ float smx[4];
    __shared__ Best sb[4];
    const long r = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + r * ld;
    const float* e = E + (step_dev ? (long)(*step_dev - step0) * e_step_stride : 0) + r * (long)V;
    float mx = -INFINITY;
    for (int c0 = tid; c0 < V; c0 += 256 * 8) {
        float xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = c0 + 256 * i < V ? x[c0 + 256 * i] * inv_temp_div : -INFINITY;
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, xv[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) smx[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    Best b = {INFINITY, 0x7fffffff};
    for (int c0 = tid; c0 < V; c0 += 256 * 8) {
        float xv[8], ev[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = c0 + 256 * i < V;
            xv[i] = on ? x[c0 + 256 * i] * inv_temp_div : -INFINITY, ev[i] = on ? e[c0 + 256 * i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (c0 + 256 * i >= V) continue;
            const float p = expf(xv[i] - mx);
            b = better(b, Best{p > 0.f ? ev[i] / p : INFINITY, c0 + 256 * i});
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best ob;
        ob.key = __shfl_xor(b.key, o, 64);
        ob.idx = __shfl_xor(b.idx, o, 64);
        b = better(b, ob);
    }
    if (lane == 0) sb[wave] = b;
    __syncthreads();
    if (tid == 0) {
        b = better(better(sb[0], sb[1]), better(sb[2], sb[3]));
        tok_out[r] = (long long)(b.idx < V ? b.idx : 0) + tok_offset;
    }
}

// grid (Bm, b); dynamic LDS: TS floats
__global__ __launch_bounds__(256) void mp_select_keep_kernel(const float* __restrict__ Y, const float* __restrict__ E,
                                                             const unsigned char* __restrict__ preserve, int TS, int Bm,
                                                             int k, unsigned char* __restrict__ mask1) {
    extern __shared__ float keys[];
    __shared__ int cnt[2];
    const int j = blockIdx.x, i = blockIdx.y;
    const float* y = Y + (long)i * TS;
    const float* e = E + ((long)i * Bm + j) * TS;
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    int nv = 0, nnz = 0;
    for (int p = threadIdx.x; p < TS; p += 256) {
        const bool valid = !(preserve && preserve[p]);
        float key = INFINITY;
        if (valid) {
            ++nv;
            if (y[p] > 0.f) ++nnz, key = e[p] / y[p];
        }
        keys[p] = key;
    }
    atomicAdd(&cnt[0], nv), atomicAdd(&cnt[1], nnz);
    __syncthreads();
    // torch.multinomial(Y_valid, k, replacement=False) raises for k <= 0, k > #categories or too few non-zero
    // weights; the reference then samples ONE position instead (dalle_bert.py:653-661)
    const int k_eff = (k >= 1 && k <= cnt[0] && k <= cnt[1]) ? k : 1;
    unsigned char* m = mask1 + ((long)i * Bm + j) * TS;
    for (int p = threadIdx.x; p < TS; p += 256) {
        const bool valid = !(preserve && preserve[p]);
        unsigned char keep = 1;  // preserved positions always stay (dalle_bert.py:664)
        if (valid) {
            const float kp = keys[p];
            int rank = 0;
            for (int q = 0; q < TS; ++q) {
                const float kq = keys[q];
                rank += (kq < kp || (kq == kp && q < p)) ? 1 : 0;
            }
            keep = (rank < k_eff && kp < INFINITY) ? 1 : 0;
        }
        m[p] = keep;
    }
}

// x[seq, l, :] = l < csl ? control_emb[i, l, :] : image_emb[id] + tpos[l - csl],  seq = i*Bm + j,
// id = mask1 ? (mask1[seq][t] ? I_tok[i][t] : MASK) : I_tok[i][t].   One wave per row.
__global__ __launch_bounds__(256) void mp_build_input_kernel(const float* __restrict__ control_emb,
                                                             const float* __restrict__ image_emb, long table_rows,
                                                             const float* __restrict__ tpos,
                                                             const long long* __restrict__ I_tok,
                                                             const unsigned char* __restrict__ mask1, int b, int Bm, int csl,
                                                             int TS, int E, long long MASK, float* __restrict__ out) {
    const long L = csl + TS;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)b * Bm * L) return;
    const int lane = threadIdx.x & 63;
    const long seq = row / L;
    const int l = (int)(row - seq * L);
    const int i = (int)(seq / Bm);
    float4* dst = reinterpret_cast<float4*>(out + row * E);
    if (l < csl) {
        const float4* src = reinterpret_cast<const float4*>(control_emb + ((long)i * csl + l) * E);
        for (int c = lane; c < (E >> 2); c += 64) dst[c] = src[c];
        return;
    }
    const int t = l - csl;
    long long id = I_tok[(long)i * TS + t];
    if (mask1 && !mask1[seq * TS + t]) id = MASK;
    if (id < 0 || id >= table_rows) id = MASK;
    const float4* src = reinterpret_cast<const float4*>(image_emb + id * E);
    const float4* pp = reinterpret_cast<const float4*>(tpos + (long)t * E);
    for (int c = lane; c < (E >> 2); c += 64) {
        const float4 a = src[c], p = pp[c];
        dst[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// One block per video: score the Bm candidates, pick the best, apply the reference's sequential where-chain
// (dalle_bert.py:675-692: candidate j's update starts from candidate j-1's result), dynamic early stop (701-707).
__global__ __launch_bounds__(256) void mp_update_kernel(const unsigned char* __restrict__ mask1, const float* __restrict__ Ynew,
                                                        const long long* __restrict__ Inew, const float* __restrict__ rel_logit,
                                                        const float* __restrict__ vid_logit, int Bm, int TS, int t, int dynamic,
                                                        float* __restrict__ Y, long long* __restrict__ I_tok,
                                                        long long* __restrict__ Imax, float* __restrict__ Smax,
                                                        int* __restrict__ tmax, unsigned char* __restrict__ active,
                                                        float* __restrict__ S_out, int* __restrict__ jmax_out) {
    __shared__ int s_jmax, s_copy;
    const int i = blockIdx.x;
    if (!active[i]) return;  // this video already stopped: its state is frozen
    if (threadIdx.x == 0) {
        float best = -INFINITY;
        int jm = 0;
        for (int j = 0; j < Bm; ++j) {
            const float sr = 1.0f / (1.0f + expf(-rel_logit[(long)i * Bm + j]));
            const float sv = 1.0f / (1.0f + expf(-vid_logit[(long)i * Bm + j]));
            const float s = sr * 0.5f + sv * 0.5f;
            if (S_out) S_out[(long)i * Bm + j] = s;
            if (s > best) best = s, jm = j;  // first maximum, as torch.argmax
        }
        int copy = 1;
        if (dynamic) {
            copy = 0;
            if (best > Smax[i]) Smax[i] = best, tmax[i] = t, copy = 1;
            if (t - tmax[i] >= 5) active[i] = 0;
        }
        s_jmax = jm, s_copy = copy;
        if (jmax_out) jmax_out[i] = jm;
    }
    __syncthreads();
    const int jm = s_jmax;
    for (int p = threadIdx.x; p < TS; p += 256) {
        float y = Y[(long)i * TS + p];
        long long tk = I_tok[(long)i * TS + p];
        for (int j = 0; j <= jm; ++j) {
            const long o = ((long)i * Bm + j) * TS + p;
            if (!mask1[o]) y = Ynew[o], tk = Inew[o];
        }
        Y[(long)i * TS + p] = y;
        I_tok[(long)i * TS + p] = tk;
        if (s_copy) Imax[(long)i * TS + p] = tk;
    }
}

// ---- the two 768 -> 1 heads (to_logits_rel / to_logits_vid = LayerNorm + Linear(dim, 1), dalle_bert.py:418-425) and
// their BCE-with-logits losses (1067-1084, 1107-1123), forward and backward, on R gathered rows of the tower output.
//   z_r   = LN(x[rows[r]]) . w + b
//   loss  = sum_r rw_r * bce(z_r, label_r) / den,   den = den_from ? max(1, sum den_from[0..nden)) : den_const
// One block; R is 2 * batch (a dozen rows), so every reduction is a fixed-order loop: deterministic.
constexpr int HEAD_MAXV = 4;  // float4 per lane -> E <= 1024

__device__ __forceinline__ float head_den(const float* den_from, int nden, float den_const) {
    if (!den_from) return den_const;
    float s = 0.f;
    for (int i = 0; i < nden; ++i) s += den_from[i];
    return fmaxf(1.f, s);
}

__global__ __launch_bounds__(256) void head_bce_fwd_kernel(const float* __restrict__ x, long ldx,
                                                           const long long* __restrict__ rows, int R, int E,
                                                           const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                           float eps, const float* __restrict__ w, const float* __restrict__ b,
                                                           const float* __restrict__ label, const float* __restrict__ rw,
                                                           const float* __restrict__ den_from, int nden, float den_const,
                                                           float* __restrict__ z_out, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float* __restrict__ loss_out) {
    __shared__ float bce[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = E >> 2;
    for (int r = wave; r < R; r += 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + rows[r] * ldx);
        float4 v[HEAD_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAXV; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < nv ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)E + eps);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 g4 = reinterpret_cast<const float4*>(lnw)[c], b4 = reinterpret_cast<const float4*>(lnb)[c];
                const float4 w4 = reinterpret_cast<const float4*>(w)[c];
                d += (((v[i].x - mean) * rstd * g4.x + b4.x) * w4.x + ((v[i].y - mean) * rstd * g4.y + b4.y) * w4.y) +
                     (((v[i].z - mean) * rstd * g4.z + b4.z) * w4.z + ((v[i].w - mean) * rstd * g4.w + b4.w) * w4.w);
            }
        }
        const float z = wave_sum(d) + b[0];
        if (lane == 0) {
            z_out[r] = z, mean_out[r] = mean, rstd_out[r] = rstd;
            if (label) {  // F.binary_cross_entropy_with_logits: max(z,0) - z*y + log1p(exp(-|z|))
                const float l = fmaxf(z, 0.f) - z * label[r] + log1pf(expf(-fabsf(z)));
                bce[r] = l * (rw ? rw[r] : 1.f);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && label && loss_out) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += bce[r];
        loss_out[0] = s / head_den(den_from, nden, den_const);
    }
}

// dz_r = gloss * rw_r / den * (sigmoid(z_r) - label_r); accumulates dw, db, dlnw, dlnb (+=) and adds the LayerNorm
// backward of each row into dx[rows[r]] (rows are distinct).
__global__ __launch_bounds__(256) void head_bce_bwd_kernel(const float* __restrict__ x, long ldx,
                                                           const long long* __restrict__ rows, int R, int E,
                                                           const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                           const float* __restrict__ w, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ label, const float* __restrict__ rw,
                                                           const float* __restrict__ den_from, int nden, float den_const,
                                                           const float* __restrict__ gloss, float* __restrict__ dx, long lddx,
                                                           float* __restrict__ dw, float* __restrict__ db,
                                                           float* __restrict__ dlnw, float* __restrict__ dlnb) {
    __shared__ float red[3][4][HEAD_MAXV * 64 * 4];  // [dw|dlnw|dlnb][wave][E] = 48 KiB
    __shared__ float dbw[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = E >> 2;
    const float scale = gloss[0] / head_den(den_from, nden, den_const);
    float4 aw[HEAD_MAXV], ag[HEAD_MAXV], ab[HEAD_MAXV];
#pragma unroll
    for (int i = 0; i < HEAD_MAXV; ++i) aw[i] = ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float adb = 0.f;
    for (int r = wave; r < R; r += 4) {
        const float dz = scale * (rw ? rw[r] : 1.f) * (1.0f / (1.0f + expf(-z[r])) - label[r]);
        const float mu = mean[r], rs = rstd[r];
        const float4* xr = reinterpret_cast<const float4*>(x + rows[r] * ldx);
        float4 xh[HEAD_MAXV], g[HEAD_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < HEAD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 v = xr[c];
                const float4 g4 = reinterpret_cast<const float4*>(lnw)[c], b4 = reinterpret_cast<const float4*>(lnb)[c];
                const float4 w4 = reinterpret_cast<const float4*>(w)[c];
                xh[i] = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
                const float4 dh = make_float4(dz * w4.x, dz * w4.y, dz * w4.z, dz * w4.w);
                aw[i].x += dz * (xh[i].x * g4.x + b4.x), aw[i].y += dz * (xh[i].y * g4.y + b4.y);
                aw[i].z += dz * (xh[i].z * g4.z + b4.z), aw[i].w += dz * (xh[i].w * g4.w + b4.w);
                ag[i].x += dh.x * xh[i].x, ag[i].y += dh.y * xh[i].y, ag[i].z += dh.z * xh[i].z, ag[i].w += dh.w * xh[i].w;
                ab[i].x += dh.x, ab[i].y += dh.y, ab[i].z += dh.z, ab[i].w += dh.w;
                g[i] = make_float4(dh.x * g4.x, dh.y * g4.y, dh.z * g4.z, dh.w * g4.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            }
        }
        adb += dz;
        const float m1 = wave_sum(s1) / (float)E, m2 = wave_sum(s2) / (float)E;
        float4* dr = reinterpret_cast<float4*>(dx + rows[r] * lddx);
#pragma unroll
        for (int i = 0; i < HEAD_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o = dr[c];
                o.x += rs * (g[i].x - m1 - xh[i].x * m2), o.y += rs * (g[i].y - m1 - xh[i].y * m2);
                o.z += rs * (g[i].z - m1 - xh[i].z * m2), o.w += rs * (g[i].w - m1 - xh[i].w * m2);
                dr[c] = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < HEAD_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            reinterpret_cast<float4*>(red[0][wave])[c] = aw[i];
            reinterpret_cast<float4*>(red[1][wave])[c] = ag[i];
            reinterpret_cast<float4*>(red[2][wave])[c] = ab[i];
        }
    }
    if (lane == 0) dbw[wave] = adb;
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
        if (dw) dw[e] += (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
        if (dlnw) dlnw[e] += (red[1][0][e] + red[1][1][e]) + (red[1][2][e] + red[1][3][e]);
        if (dlnb) dlnb[e] += (red[2][0][e] + red[2][1][e]) + (red[2][2][e] + red[2][3][e]);
    }
    if (threadIdx.x == 0 && db) db[0] += (dbw[0] + dbw[1]) + (dbw[2] + dbw[3]);
}

// ---- BERT training step: every token id of the (up to) three sequences of a step, plus the rows / labels of the MSM
// cross entropy, in ONE elementwise launch (dalle_bert.py:903-973 control ids, 1030-1035 masked targets, 1057 REL swap,
// 1094-1100 VID).  Sequence block s of `ids`: 0 = MSM, then REL negative (control ids of sample (b + B/2) % B:
// swap() of an even batch, 110-114), then VID negative (warped targets).
__global__ __launch_bounds__(256) void bert_build_ids_kernel(const long long* __restrict__ text,
                                                             const long long* __restrict__ text_neg,
                                                             const long long* __restrict__ visual_tok,
                                                             const long long* __restrict__ target,
                                                             const long long* __restrict__ target_warp,
                                                             const unsigned char* __restrict__ mask1, int B, int Ttxt, int Nvis,
                                                             int TS, long long pad_base, long long MASK, int has_rel, int has_vid,
                                                             long long* __restrict__ ids, unsigned char* __restrict__ select_full,
                                                             long long* __restrict__ target_full) {
    const int csl = 1 + Ttxt + Nvis + 2, L = csl + TS;
    const int nseq = 1 + has_rel + has_vid;
    const long total = (long)nseq * B * L;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int l = (int)(idx % L);
    const int sb = (int)(idx / L);
    const int s = sb / B, b = sb - s * B;
    const bool is_rel = has_rel && s == 1;
    const bool is_vid = has_vid && s == 1 + has_rel;
    (void)nseq;
    long long id;
    if (l < csl) {
        if (l == 0) {
            id = 0;  // [REL]
        } else if (l <= Ttxt) {
            const long long* src = (is_rel && text_neg) ? text_neg : text;
            const int bs = (is_rel && !text_neg) ? (b + B / 2) % B : b;
            const long long tk = src[(long)bs * Ttxt + (l - 1)];
            id = tk == 0 ? pad_base + (l - 1) : tk;  // unique pad id per position (917-919)
        } else if (l <= Ttxt + Nvis) {
            const int bs = is_rel ? (b + B / 2) % B : b;
            id = visual_tok ? visual_tok[(long)bs * Nvis + (l - 1 - Ttxt)] : MASK;
        } else {
            id = l - (Ttxt + Nvis);  // [ST1] = 1, [VID] = 2
        }
    } else {
        const int t = l - csl;
        const long long tg = is_vid ? target_warp[(long)b * TS + t] : target[(long)b * TS + t];
        id = mask1[(long)b * TS + t] ? tg : MASK;
    }
    ids[idx] = id;
    if (s == 0) {
        const int t = l - csl;
        select_full[(long)b * L + l] = (l >= csl && !mask1[(long)b * TS + t]) ? 1 : 0;
        target_full[(long)b * L + l] = l >= csl ? target[(long)b * TS + t] : 0;
    }
}

// count of selected rows as a float (the CE denominator), one block
__global__ __launch_bounds__(256) void count_u8_kernel(const unsigned char* __restrict__ sel, long n, float* __restrict__ out) {
    __shared__ int red[256];
    int c = 0;
    for (long i = threadIdx.x; i < n; i += 256) c += sel[i] ? 1 : 0;
    red[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

}  // namespace

extern "C" int mmvid_sample_race(const float* logits, int64_t ld, const float* E, const float* noise_u, float temperature,
                                 float logit_div, int64_t R, int V, int64_t tok_offset, int64_t* tok, float* y,
                                 void* stream) {
    return mmvid_sample_race_at(logits, ld, E, nullptr, 0, 0, noise_u, temperature, logit_div, R, V, tok_offset, tok, y, stream);
}

// The same with the variates of draw number (*step_dev - step0) of a pre-drawn block E [draws][R][V] (e_step_stride = R * V elements);
// step_dev == null: E is the [R][V] block itself.
extern "C" int mmvid_sample_race_at(const float* logits, int64_t ld, const float* E, const int32_t* step_dev, int step0,
                                    int64_t e_step_stride, const float* noise_u, float temperature, float logit_div, int64_t R, int V,
                                    int64_t tok_offset, int64_t* tok, float* y, void* stream) {
    MMVID_REQUIRE(logits && E && tok && R >= 0 && V > 0, "sample_race: bad arguments");
    MMVID_REQUIRE(logit_div > 0.f, "sample_race: logit_div (the softmax temperature divisor) must be > 0");
    if (R == 0) return MMVID_OK;
    if (!y && !noise_u && R <= 1024) {  // the ART-V draw: one block per row
        hipLaunchKernelGGL(sample_race_row_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, E, step_dev, step0,
                           (long)e_step_stride, 1.0f / logit_div, V, (long long)tok_offset, (long long*)tok);
    } else {
        MMVID_REQUIRE(!step_dev, "sample_race_at: a device-side draw index needs the token-only form (no y, no noise, R <= 1024)");
        hipLaunchKernelGGL(sample_race_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, E, noise_u,
                           temperature, 1.0f / logit_div, (long)R, V, (long long)tok_offset, (long long*)tok, y);
    }
    MMVID_LAUNCH_CHECK("sample_race");
    return MMVID_OK;
}

extern "C" int mmvid_mp_select_keep(const float* Y, const float* E, const uint8_t* preserve, int b, int Bm, int TS, int k,
                                    uint8_t* mask1, void* stream) {
    MMVID_REQUIRE(Y && E && mask1 && b > 0 && Bm > 0 && TS > 0 && TS <= 16384, "mp_select_keep: bad arguments");
    hipLaunchKernelGGL(mp_select_keep_kernel, dim3(Bm, b), dim3(256), (size_t)TS * 4, (hipStream_t)stream, Y, E, preserve, TS, Bm,
                       k, mask1);
    MMVID_LAUNCH_CHECK("mp_select_keep");
    return MMVID_OK;
}

extern "C" int mmvid_mp_build_input(const float* control_emb, const float* image_emb, int64_t table_rows, const float* tpos,
                                    const int64_t* I_tok, const uint8_t* mask1, int b, int Bm, int csl, int TS, int E,
                                    int64_t mask_id, float* out, void* stream) {
    MMVID_REQUIRE(control_emb && image_emb && tpos && I_tok && out, "mp_build_input: null pointer");
    MMVID_REQUIRE(E % 4 == 0 && mask_id >= 0 && mask_id < table_rows, "mp_build_input: E=%d mask_id=%ld", E, (long)mask_id);
    const long rows = (long)b * Bm * (csl + TS);
    if (rows == 0) return MMVID_OK;
    hipLaunchKernelGGL(mp_build_input_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, control_emb, image_emb,
                       (long)table_rows, tpos, (const long long*)I_tok, mask1, b, Bm, csl, TS, E, (long long)mask_id, out);
    MMVID_LAUNCH_CHECK("mp_build_input");
    return MMVID_OK;
}

extern "C" int mmvid_mp_update(const uint8_t* mask1, const float* Ynew, const int64_t* Inew, const float* rel_logit,
                               const float* vid_logit, int b, int Bm, int TS, int t, int dynamic, float* Y, int64_t* I_tok,
                               int64_t* Imax, float* Smax, int32_t* tmax, uint8_t* active, float* S_out, int32_t* jmax_out,
                               void* stream) {
    MMVID_REQUIRE(mask1 && Ynew && Inew && rel_logit && vid_logit && Y && I_tok && Imax && Smax && tmax && active,
                  "mp_update: null pointer");
    hipLaunchKernelGGL(mp_update_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, mask1, Ynew, (const long long*)Inew,
                       rel_logit, vid_logit, Bm, TS, t, dynamic, Y, (long long*)I_tok, (long long*)Imax, Smax, tmax, active, S_out,
                       jmax_out);
    MMVID_LAUNCH_CHECK("mp_update");
    return MMVID_OK;
}

extern "C" int mmvid_head_bce_fwd(const float* x, int64_t ldx, const int64_t* rows, int R, int E, const float* ln_w,
                                  const float* ln_b, float eps, const float* w, const float* b, const float* label,
                                  const float* row_weight, const float* den_from, int nden, float den_const, float* z,
                                  float* mean, float* rstd, float* loss, void* stream) {
    MMVID_REQUIRE(x && rows && ln_w && ln_b && w && b && z && mean && rstd, "head_bce_fwd: null pointer");
    MMVID_REQUIRE(R > 0 && R <= 256 && E % 4 == 0 && E <= 64 * 4 * HEAD_MAXV && ldx % 4 == 0, "head_bce_fwd: R=%d E=%d", R, E);
    hipLaunchKernelGGL(head_bce_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (const long long*)rows, R, E,
                       ln_w, ln_b, eps, w, b, label, row_weight, den_from, nden, den_const, z, mean, rstd, loss);
    MMVID_LAUNCH_CHECK("head_bce_fwd");
    return MMVID_OK;
}

extern "C" int mmvid_head_bce_bwd(const float* x, int64_t ldx, const int64_t* rows, int R, int E, const float* ln_w,
                                  const float* ln_b, const float* w, const float* z, const float* mean, const float* rstd,
                                  const float* label, const float* row_weight, const float* den_from, int nden,
                                  float den_const, const float* gloss, float* dx, int64_t lddx, float* dw, float* db,
                                  float* dln_w, float* dln_b, void* stream) {
    MMVID_REQUIRE(x && rows && ln_w && ln_b && w && z && mean && rstd && label && gloss && dx, "head_bce_bwd: null pointer");
    MMVID_REQUIRE(R > 0 && R <= 256 && E % 4 == 0 && E <= 64 * 4 * HEAD_MAXV && ldx % 4 == 0 && lddx % 4 == 0,
                  "head_bce_bwd: R=%d E=%d", R, E);
    hipLaunchKernelGGL(head_bce_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, (const long long*)rows, R, E,
                       ln_w, ln_b, w, z, mean, rstd, label, row_weight, den_from, nden, den_const, gloss, dx, (long)lddx, dw, db,
                       dln_w, dln_b);
    MMVID_LAUNCH_CHECK("head_bce_bwd");
    return MMVID_OK;
}

extern "C" int mmvid_bert_build_ids(const int64_t* text, const int64_t* text_neg, const int64_t* visual_tok,
                                    const int64_t* target, const int64_t* target_warp, const uint8_t* mask1, int B, int Ttxt,
                                    int Nvis, int TS, int64_t pad_base, int64_t mask_id, int has_rel, int has_vid, int64_t* ids,
                                    uint8_t* select_full, int64_t* target_full, float* select_count, void* stream) {
    MMVID_REQUIRE(text && ids && select_full && target_full && (TS == 0 || (target && mask1)), "bert_build_ids: null pointer");
    MMVID_REQUIRE(!has_vid || target_warp, "bert_build_ids: the VID sequence needs target_warp");
    MMVID_REQUIRE(!has_rel || text_neg || B % 2 == 0, "bert_build_ids: REL swapping needs an even batch (dalle_bert.py:1045-1046)");
    const int L = 1 + Ttxt + Nvis + 2 + TS;
    const long total = (long)(1 + (has_rel ? 1 : 0) + (has_vid ? 1 : 0)) * B * L;
    hipLaunchKernelGGL(bert_build_ids_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)text,
                       (const long long*)text_neg, (const long long*)visual_tok, (const long long*)target,
                       (const long long*)target_warp, mask1, B, Ttxt, Nvis, TS, (long long)pad_base, (long long)mask_id,
                       has_rel ? 1 : 0, has_vid ? 1 : 0, (long long*)ids, select_full, (long long*)target_full);
    if (select_count)
        hipLaunchKernelGGL(count_u8_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, select_full, (long)B * L, select_count);
    MMVID_LAUNCH_CHECK("bert_build_ids");
    return MMVID_OK;
}
