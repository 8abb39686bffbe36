// Stochastic front-end of a BERT training step as device code (SURVEY next-row N2): the reference draws these on the
// host, per sample, from four different RNGs (numpy, `random`, torch, torchvision) inside forward():
//   MSM masking strategies        mmvid_pytorch/dalle_bert.py:992-1029  (Bernoulli / full / RandomErasing box / inverse box,
//                                 optional frame preservation 1022-1026)
//   VID negative `warp()`         dalle_bert.py:204-238  (frame from another sample / frame shuffle / colour shift /
//                                 affine warp of one frame, 93-202)
//   visual-token erasing          dalle_bert.py:779-848  (random_erase_codebook, erase_codebook_face)
// Here one counter-based generator (Philox4x32-10) keyed by (seed, step, sample, purpose) makes every decision on the
// device: the captured training step replays without host input, and two ranks with different seeds draw different
// streams.  The reference's streams cannot be reproduced bit for bit (torchvision is unpinned, numpy/python generators
// do not exist on the device); what is restated is each DISTRIBUTION, checked statistically in tests/ against
// oracle/frontend.py.  All kernels are elementwise / tiny: HBM-bound.
#include "../../include/mmvid_hip.h"
#include "common.h"

namespace {

// ---- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") --------------------------------
struct U4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }  // [0, 1)

struct Rng {  // one stream = (seed, step, sample, purpose); draw i of the stream = word (i & 3) of block (i >> 2)
    uint32_t k0, k1, step, sample, purpose;
    __device__ __forceinline__ float uniform(uint32_t i) const {
        const U4 r = philox(i >> 2, purpose, sample, step, k0, k1);
        const uint32_t w = (i & 3) == 0 ? r.x : ((i & 3) == 1 ? r.y : ((i & 3) == 2 ? r.z : r.w));
        return u01(w);
    }
};
__device__ __forceinline__ Rng make_rng(uint64_t seed, const float* step_dev, int sample, int purpose) {
    Rng g;
    g.k0 = (uint32_t)seed, g.k1 = (uint32_t)(seed >> 32);
    g.step = 0u;
    if (step_dev) {  // front-end state: {forward-call counter (fp32), seed low word, seed high word, reserved}
        g.step = (uint32_t)step_dev[0];
        g.k0 ^= __float_as_uint(step_dev[1]), g.k1 ^= __float_as_uint(step_dev[2]);
    }
    g.sample = (uint32_t)sample, g.purpose = (uint32_t)purpose;
    return g;
}
enum { P_STRATEGY = 1, P_BERNOULLI = 2, P_BOX = 3, P_PC = 4, P_WARP = 5, P_PERM = 6, P_ERASE = 7, P_FACE = 8, P_VCOLOR = 9 };

// torchvision.transforms.RandomErasing.get_params (third-party, absent from /root/reference; restated from the
// published semantics): up to 10 attempts of { area*U(scale), exp(U(log ratio)) -> h, w = round(sqrt(..)) ; accept if
// h < H and w < W ; top-left uniform over the valid offsets }.  Returns false when no attempt fits (nothing erased).
__device__ bool erasing_box(const Rng& g, uint32_t base, int H, int W, float s0, float s1, float r0, float r1, int& bi, int& bj,
                            int& bh, int& bw) {
    const float area = (float)(H * W), l0 = logf(r0), l1 = logf(r1);
    for (int a = 0; a < 10; ++a) {
        const float ea = area * (s0 + (s1 - s0) * g.uniform(base + 4 * a));
        const float ar = expf(l0 + (l1 - l0) * g.uniform(base + 4 * a + 1));
        const int h = (int)rintf(sqrtf(ea * ar)), w = (int)rintf(sqrtf(ea / ar));
        if (!(h < H && w < W)) continue;
        bi = (int)(g.uniform(base + 4 * a + 2) * (float)(H - h + 1));
        bj = (int)(g.uniform(base + 4 * a + 3) * (float)(W - w + 1));
        if (bi > H - h) bi = H - h;
        if (bj > W - w) bj = W - w;
        bh = h, bw = w;
        return true;
    }
    return false;
}

// ---- MSM masks: one block per sample.  mask1[b, t*f*f + y*f + x] = 1 where the target token stays VISIBLE.
__global__ __launch_bounds__(256) void msm_mask_kernel(uint64_t seed, const float* __restrict__ step_dev, int T, int f,
                                                       float p1, float p2, float p3, float bern_lo, float bern_hi,
                                                       float pc_prob, unsigned char* __restrict__ mask1,
                                                       float* __restrict__ nfm, int* __restrict__ strategy_out,
                                                       const int* __restrict__ inject,
                                                       const unsigned char* __restrict__ bern_inject) {
    __shared__ int s_strat, s_box[4], s_hasbox;
    __shared__ float s_p;
    __shared__ unsigned char s_keep_frame[64];
    const int b = blockIdx.x, TS = T * f * f;
    if (threadIdx.x == 0 && inject) {  // tests: the decisions of a reference run instead of draws (MSM_INJECT_WORDS per sample:
        const int* d = inject + (long)b * 72;  // strategy, has_box, box i, j, h, w, 2 reserved, keep_frame[64])
        s_strat = d[0], s_hasbox = d[1], s_p = 0.f;
        s_box[0] = d[2], s_box[1] = d[3], s_box[2] = d[4], s_box[3] = d[5];
        for (int t = 0; t < 64; ++t) s_keep_frame[t] = (t < T && d[8 + t]) ? 1 : 0;
        nfm[b] = d[0] == 2 ? 0.f : 1.f;
        if (strategy_out) strategy_out[b] = d[0];
    } else if (threadIdx.x == 0) {
        const Rng g = make_rng(seed, step_dev, b, P_STRATEGY);
        const float u = g.uniform(0);
        const int strat = u < p1 ? 1 : (u < p1 + p2 ? 2 : (u < p1 + p2 + p3 ? 3 : 4));  // np.random.choice([1,2,3,4], p)
        s_strat = strat;
        s_p = bern_lo + (bern_hi - bern_lo) * g.uniform(1);  // np.random.uniform(*msm_bernoulli_prob)
        s_hasbox = 0;
        if (strat >= 3) {
            int bi = 0, bj = 0, bh = 0, bw = 0;
            const Rng gb = make_rng(seed, step_dev, b, P_BOX);
            s_hasbox = erasing_box(gb, 0, f, f, 0.2f, 0.8f, 0.5f, 2.0f, bi, bj, bh, bw) ? 1 : 0;  // dalle_bert.py:290-294
            s_box[0] = bi, s_box[1] = bj, s_box[2] = bh, s_box[3] = bw;
        }
        nfm[b] = strat == 2 ? 0.f : 1.f;
        if (strategy_out) strategy_out[b] = strat;
        for (int t = 0; t < T && t < 64; ++t) s_keep_frame[t] = 0;
        if (pc_prob > 0.f) {  // 1022-1026: with probability pc_prob keep randint(1, T//2) whole frames visible
            const Rng gp = make_rng(seed, step_dev, b, P_PC);
            if (gp.uniform(0) < pc_prob && T >= 2) {
                int cnt = 1 + (int)(gp.uniform(1) * (float)(T / 2));
                if (cnt > T / 2) cnt = T / 2;
                for (int t = 0; t < T && t < 64; ++t) {  // the cnt frames with the smallest draws = a uniform cnt-subset
                    const float ut = gp.uniform(2 + t);
                    int rank = 0;
                    for (int q = 0; q < T; ++q) {
                        const float uq = gp.uniform(2 + q);
                        rank += (uq < ut || (uq == ut && q < t)) ? 1 : 0;
                    }
                    s_keep_frame[t] = rank < cnt ? 1 : 0;
                }
            }
        }
    }
    __syncthreads();
    const int strat = s_strat;
    const Rng gbern = make_rng(seed, step_dev, b, P_BERNOULLI);
    for (int p = threadIdx.x; p < TS; p += 256) {
        const int t = p / (f * f), rem = p - t * f * f, y = rem / f, x = rem - y * f;
        unsigned char m;
        if (strat == 1) {
            m = bern_inject ? bern_inject[(long)b * TS + p] : (gbern.uniform(p) < s_p ? 1 : 0);  // torch.bernoulli(ones * p)
        } else if (strat == 2) {
            m = 0;
        } else {
            const bool inside = s_hasbox && y >= s_box[0] && y < s_box[0] + s_box[2] && x >= s_box[1] && x < s_box[1] + s_box[3];
            m = strat == 3 ? (inside ? 0 : 1) : (inside ? 1 : 0);  // erased box hidden / only the box visible
        }
        if (t < 64 && s_keep_frame[t]) m = 1;
        mask1[(long)b * TS + p] = m;
    }
}

// ---- VID negative: per-sample parameters, then one elementwise pass over the frames
struct WarpParams {  // one per sample
    int mode, j1, src_b, src_t, chan;
    float shift, th[6];
    int perm[32];
};

__global__ void warp_params_kernel(uint64_t seed, const float* __restrict__ step_dev, int B, int T, float p0, float p1, float p2,
                                   WarpParams* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Rng g = make_rng(seed, step_dev, b, P_WARP);
    WarpParams w;
    const float u = g.uniform(0);
    w.mode = u < p0 ? 0 : (u < p0 + p1 ? 1 : (u < p0 + p1 + p2 ? 2 : 3));  // np.random.choice(range(4), p=vid_strategy_prob)
    w.j1 = (int)(g.uniform(1) * (float)T);
    if (w.j1 >= T) w.j1 = T - 1;
    w.src_b = b, w.src_t = w.j1, w.chan = 0, w.shift = 0.f;
    for (int t = 0; t < 32; ++t) w.perm[t] = t;
    for (int e = 0; e < 6; ++e) w.th[e] = 0.f;
    if (w.mode == 0) {  // a frame of another sample (210-217); a batch of one has no "other": keep the frame
        if (B > 1) {
            int o = (int)(g.uniform(2) * (float)(B - 1));
            if (o >= B - 1) o = B - 2;
            w.src_b = o >= b ? o + 1 : o;
        }
        w.src_t = (int)(g.uniform(3) * (float)T);
        if (w.src_t >= T) w.src_t = T - 1;
    } else if (w.mode == 1) {  // a non-identity permutation of the frames (93-108, 218-219)
        const Rng gp = make_rng(seed, step_dev, b, P_PERM);
        for (int attempt = 0; attempt < 16; ++attempt) {
            for (int t = 0; t < T; ++t) w.perm[t] = t;
            for (int t = T - 1; t > 0; --t) {  // Fisher-Yates
                int r = (int)(gp.uniform(attempt * 32 + t) * (float)(t + 1));
                if (r > t) r = t;
                const int tmp = w.perm[t];
                w.perm[t] = w.perm[r], w.perm[r] = tmp;
            }
            bool ident = true;
            for (int t = 0; t < T; ++t) ident = ident && w.perm[t] == t;
            if (!ident || T < 2) break;
        }
    } else if (w.mode == 2) {  // colour shift of one frame (124-135): all channels or one of them
        w.shift = g.uniform(2) - 0.5f;
        w.chan = (int)(g.uniform(3) * 4.0f);
        if (w.chan > 3) w.chan = 3;
    } else {  // affine warp of one frame (168-202 with angle 30, trans 0.1, scale 0.1)
        const float ang = 3.14159265358979323846f * 30.0f / 180.0f;
        const float a = -ang + 2.f * ang * g.uniform(2);
        const float t1 = -0.1f + 0.2f * g.uniform(3), t2 = -0.1f + 0.2f * g.uniform(4);
        const float sc = 0.9f + 0.2f * g.uniform(5);
        w.th[0] = sc * cosf(a), w.th[1] = sc * sinf(-a), w.th[2] = t1;
        w.th[3] = sc * sinf(a), w.th[4] = sc * cosf(a), w.th[5] = t2;
    }
    out[b] = w;
}

__device__ __forceinline__ float reflect_coord(float in, float twice_low, float twice_high) {  // ATen GridSampler.h
    if (twice_low == twice_high) return 0.f;
    const float mn = twice_low * 0.5f, span = (twice_high - twice_low) * 0.5f;
    in = fabsf(in - mn);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    return (flips & 1) ? span - extra + mn : extra + mn;
}

// F.affine_grid(theta, size, align_corners=False) + F.grid_sample(bilinear, padding_mode='reflection', align_corners=False)
__device__ __forceinline__ float affine_sample(const float* __restrict__ img, int H, int W, const float* th, int y, int x) {
    const float xn = (2.f * x + 1.f) / (float)W - 1.f, yn = (2.f * y + 1.f) / (float)H - 1.f;
    const float gx = th[0] * xn + th[1] * yn + th[2], gy = th[3] * xn + th[4] * yn + th[5];
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    ix = fminf(fmaxf(reflect_coord(ix, -1.f, 2.f * W - 1.f), 0.f), (float)(W - 1));
    iy = fminf(fmaxf(reflect_coord(iy, -1.f, 2.f * H - 1.f), 0.f), (float)(H - 1));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float v = 0.f;
    if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) v += img[y0 * W + x0] * wx0 * wy0;
    if (x1 >= 0 && x1 < W && y0 >= 0 && y0 < H) v += img[y0 * W + x1] * wx1 * wy0;
    if (x0 >= 0 && x0 < W && y1 >= 0 && y1 < H) v += img[y1 * W + x0] * wx0 * wy1;
    if (x1 >= 0 && x1 < W && y1 >= 0 && y1 < H) v += img[y1 * W + x1] * wx1 * wy1;
    return v;
}

// x, out: [B, T, C, H, W] fp32.  One thread per output pixel.
__global__ __launch_bounds__(256) void warp_apply_kernel(const float* __restrict__ x, const WarpParams* __restrict__ wp, int B,
                                                         int T, int C, int H, int W, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W, per_b = (long)T * C * hw;
    if (idx >= (long)B * per_b) return;
    const int b = (int)(idx / per_b);
    long r = idx - (long)b * per_b;
    const int t = (int)(r / (C * hw));
    r -= (long)t * C * hw;
    const int c = (int)(r / hw);
    const int pix = (int)(r - (long)c * hw);
    const WarpParams& w = wp[b];
    float v;
    if (w.mode == 1) {
        v = x[(((long)b * T + w.perm[t]) * C + c) * hw + pix];
    } else if (t != w.j1) {
        v = x[idx];
    } else if (w.mode == 0) {
        v = x[(((long)w.src_b * T + w.src_t) * C + c) * hw + pix];
    } else if (w.mode == 2) {
        const float m = (w.chan == 0 || w.chan - 1 == c) ? w.shift : 0.f;
        v = fminf(fmaxf(x[idx] + m, 0.f), 1.f);
    } else {
        v = affine_sample(x + (((long)b * T + t) * C + c) * hw, H, W, w.th, pix / W, pix % W);
    }
    out[idx] = v;
}

// The VID negative differs from the target in ONE frame per sample (modes 0, 2, 3) or is a permutation of its frames
// (mode 1), and the VQGAN encodes frames independently.  So only the frames that are NEW pixels need encoding: one per
// sample (the colour-shifted / affine-warped frame; for modes 0 and 1 the slot is filled with frame j1 and its tokens
// are not used).  new_frames: [B, C, H, W].
__global__ __launch_bounds__(256) void warp_new_frame_kernel(const float* __restrict__ x, const WarpParams* __restrict__ wp, int B,
                                                             int T, int C, int H, int W, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= (long)B * C * hw) return;
    const int b = (int)(idx / (C * hw));
    const long r = idx - (long)b * C * hw;
    const int c = (int)(r / hw);
    const int pix = (int)(r - (long)c * hw);
    const WarpParams& w = wp[b];
    const float* src = x + (((long)b * T + w.j1) * C + c) * hw;
    float v;
    if (w.mode == 2) {
        const float m = (w.chan == 0 || w.chan - 1 == c) ? w.shift : 0.f;
        v = fminf(fmaxf(src[pix] + m, 0.f), 1.f);
    } else if (w.mode == 3) {
        v = affine_sample(src, H, W, w.th, pix / W, pix % W);
    } else {
        v = src[pix];
    }
    out[idx] = v;
}
// tokens of the VID negative from the target's tokens [B, T*n] and the new frames' tokens [B, n]
__global__ __launch_bounds__(256) void warp_tokens_kernel(const long long* __restrict__ tok, const long long* __restrict__ extra,
                                                          const WarpParams* __restrict__ wp, int B, int T, int n,
                                                          long long* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * T * n) return;
    const int b = (int)(idx / ((long)T * n));
    const int r = (int)(idx - (long)b * T * n);
    const int t = r / n, p = r - t * n;
    const WarpParams& w = wp[b];
    long long v;
    if (w.mode == 1)
        v = tok[((long)b * T + w.perm[t]) * n + p];
    else if (t != w.j1)
        v = tok[idx];
    else if (w.mode == 0)
        v = tok[((long)w.src_b * T + w.src_t) * n + p];
    else
        v = extra[(long)b * n + p];
    out[idx] = v;
}

// ---- visual-token erasing on [B, Tv, f, f] int64 token maps
// choice c (drawn once per call, as the reference's single random draw per forward): mode 0 = untouched, 1 = keep only
// the box (everything else -> value), 2 = erase the box.  frame0_full: frame 0 is never touched (face2 / face3 modes).
struct EraseChoice {
    float cum[4];
    int mode[4], box[4][4];  // r0, r1, c0, c1 (half-open)
    int n, frame0_full;
};
__global__ __launch_bounds__(256) void erase_choice_kernel(uint64_t seed, const float* __restrict__ step_dev, EraseChoice ch,
                                                           int B, int Tv, int f, long long value, long long* __restrict__ tok) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * Tv * f * f;
    if (idx >= total) return;
    const Rng g = make_rng(seed, step_dev, 0, P_FACE);
    const float u = g.uniform(0);
    int c = ch.n - 1;
    for (int i = 0; i < ch.n; ++i)
        if (u < ch.cum[i]) {
            c = i;
            break;
        }
    if (ch.mode[c] == 0) return;
    const int x = (int)(idx % f), y = (int)((idx / f) % f), t = (int)((idx / ((long)f * f)) % Tv);
    if (ch.frame0_full && t == 0) return;
    const bool inside = y >= ch.box[c][0] && y < ch.box[c][1] && x >= ch.box[c][2] && x < ch.box[c][3];
    if ((ch.mode[c] == 1 && !inside) || (ch.mode[c] == 2 && inside)) tok[idx] = value;
}

// random_erase_codebook (779-794): per sample RandomErasing(p, scale, ratio, value) on the [Tv, f, f] map (the same box
// on every frame); erase_half: rows f/2.. of every frame instead.
__global__ __launch_bounds__(256) void random_erase_tokens_kernel(uint64_t seed, const float* __restrict__ step_dev, int Tv,
                                                                  int f, float p, float s0, float s1, float r0, float r1,
                                                                  int erase_half, long long value, long long* __restrict__ tok) {
    __shared__ int s_box[4], s_on;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        s_on = 0;
        if (erase_half) {
            s_box[0] = f / 2, s_box[1] = 0, s_box[2] = f - f / 2, s_box[3] = f, s_on = 1;
        } else {
            const Rng g = make_rng(seed, step_dev, b, P_ERASE);
            if (g.uniform(0) < p) {  // torchvision: `if torch.rand(1) < self.p`
                int bi = 0, bj = 0, bh = 0, bw = 0;
                if (erasing_box(g, 4, f, f, s0, s1, r0, r1, bi, bj, bh, bw)) s_box[0] = bi, s_box[1] = bj, s_box[2] = bh, s_box[3] = bw, s_on = 1;
            }
        }
    }
    __syncthreads();
    if (!s_on) return;
    const int n = Tv * f * f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int x = i % f, y = (i / f) % f;
        if (y >= s_box[0] && y < s_box[0] + s_box[2] && x >= s_box[1] && x < s_box[1] + s_box[3]) tok[(long)b * n + i] = value;
    }
}

// visual_aug_mode == 'motion_color' (dalle_bert.py:140-158, 940-943; dalle_artv.py:460-463): with probability p -- ONE draw
// per forward call, `random.random() < 0.9` -- every sample's visual frames first_frame.. get the same per-sample colour
// shift c ~ U(-0.5, 0.5) on all channels or on one of them (random.randint(0, 3)), clamped to [0, 1]; earlier frames and a
// failed gate leave the pixels untouched.  x [B, Tv, C, H, W] fp32, in place.  params_out (optional, tests): per sample
// {gate, shift, chan}.
__global__ __launch_bounds__(256) void visual_color_kernel(uint64_t seed, const float* __restrict__ step_dev, int B, int Tv, int C,
                                                           long hw, float p, int first_frame, float* __restrict__ x,
                                                           float* __restrict__ params_out) {
    const long per_b = (long)Tv * C * hw;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * per_b) return;
    const int b = (int)(idx / per_b);
    const long r = idx - (long)b * per_b;
    const int t = (int)(r / (C * hw)), c = (int)((r - (long)t * C * hw) / hw);
    const Rng gate = make_rng(seed, step_dev, 0, P_VCOLOR);
    const bool on = gate.uniform(0) < p;
    const Rng g = make_rng(seed, step_dev, b + 1, P_VCOLOR);
    const float shift = g.uniform(0) - 0.5f;
    int chan = (int)(g.uniform(1) * 4.0f);
    if (chan > 3) chan = 3;
    if (params_out && r == 0) params_out[3 * b] = on ? 1.f : 0.f, params_out[3 * b + 1] = shift, params_out[3 * b + 2] = (float)chan;
    if (!on || t < first_frame) return;
    const float m = (chan == 0 || chan - 1 == c) ? shift : 0.f;
    x[idx] = fminf(fmaxf(x[idx] + m, 0.f), 1.f);
}

__global__ void counter_add_kernel(float* c, float v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += v;
}

// Learning-rate schedule as a device scalar (utils_train.py:373-385 -> deepspeed WarmupLR, third-party: restated from
// its published semantics).  *step_dev counts finished optimiser steps; the reference calls scheduler.step() after
// every `every`-th iteration (train.py:373-374), so at iteration i the scheduler has been stepped ns = i / every times:
// ns == 0 -> the optimiser's construction lr; else lr = min + (max - min) * gamma(ns - 1),
// gamma(k) = k < warmup ? log(k + 1) / log(warmup) : 1.   kind 0 = constant lr_max.
__global__ void lr_schedule_kernel(const float* __restrict__ step_dev, int kind, float lr_min, float lr_max, int warmup,
                                   int every, float* __restrict__ lr_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lr = lr_max;
    if (kind == 1) {
        const long it = (long)step_dev[0];
        const long ns = it / (every > 0 ? every : 1);
        if (ns > 0) {
            const long k = ns - 1;
            const int wu = warmup < 2 ? 2 : warmup;
            const float gamma = k < wu ? logf((float)(k + 1)) / logf((float)wu) : 1.0f;
            lr = lr_min + (lr_max - lr_min) * gamma;
        }
    }
    lr_out[0] = lr;
}

}  // namespace

extern "C" int mmvid_msm_masks(uint64_t seed, const float* step_dev, int B, int T, int f, const float* strategy_prob,
                               float bern_lo, float bern_hi, float pc_prob, uint8_t* mask1, float* not_fully_masked,
                               int32_t* strategy_out, void* stream) {
    MMVID_REQUIRE(strategy_prob && mask1 && not_fully_masked && B > 0 && T > 0 && T <= 64 && f > 0, "msm_masks: bad arguments");
    hipLaunchKernelGGL(msm_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, seed, step_dev, T, f, strategy_prob[0],
                       strategy_prob[1], strategy_prob[2], bern_lo, bern_hi, pc_prob, mask1, not_fully_masked, strategy_out,
                       (const int*)nullptr, (const unsigned char*)nullptr);
    MMVID_LAUNCH_CHECK("msm_masks");
    return MMVID_OK;
}

extern "C" int mmvid_msm_masks_inject(const int32_t* decisions, const uint8_t* bernoulli, int B, int T, int f, uint8_t* mask1,
                                      float* not_fully_masked, void* stream) {
    MMVID_REQUIRE(decisions && mask1 && not_fully_masked && B > 0 && T > 0 && T <= 64 && f > 0, "msm_masks_inject: bad arguments");
    hipLaunchKernelGGL(msm_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (uint64_t)0, (const float*)nullptr, T, f, 0.f, 0.f,
                       0.f, 0.f, 0.f, 0.f, mask1, not_fully_masked, (int*)nullptr, decisions, bernoulli);
    MMVID_LAUNCH_CHECK("msm_masks_inject");
    return MMVID_OK;
}

extern "C" int mmvid_warp_params_bytes(void) { return (int)sizeof(WarpParams); }

extern "C" int mmvid_vid_warp(uint64_t seed, const float* step_dev, const float* x, int B, int T, int C, int H, int W,
                              const float* strategy_prob, void* params_scratch, int draw_params, float* out, void* stream) {
    MMVID_REQUIRE(x && strategy_prob && params_scratch && out, "vid_warp: null pointer");
    MMVID_REQUIRE(B > 0 && T > 0 && T <= 32, "vid_warp: B=%d T=%d (T <= 32)", B, T);
    hipStream_t s = (hipStream_t)stream;
    if (draw_params)
        hipLaunchKernelGGL(warp_params_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, seed, step_dev, B, T, strategy_prob[0],
                           strategy_prob[1], strategy_prob[2], (WarpParams*)params_scratch);
    const long total = (long)B * T * C * H * W;
    hipLaunchKernelGGL(warp_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, x, (const WarpParams*)params_scratch, B, T, C, H,
                       W, out);
    MMVID_LAUNCH_CHECK("vid_warp");
    return MMVID_OK;
}

extern "C" int mmvid_vid_warp_new_frames(uint64_t seed, const float* step_dev, const float* x, int B, int T, int C, int H, int W,
                                         const float* strategy_prob, void* params_scratch, int draw_params, float* new_frames,
                                         void* stream) {
    MMVID_REQUIRE(x && strategy_prob && params_scratch && new_frames, "vid_warp_new_frames: null pointer");
    MMVID_REQUIRE(B > 0 && T > 0 && T <= 32, "vid_warp_new_frames: B=%d T=%d (T <= 32)", B, T);
    hipStream_t s = (hipStream_t)stream;
    if (draw_params)
        hipLaunchKernelGGL(warp_params_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, seed, step_dev, B, T, strategy_prob[0],
                           strategy_prob[1], strategy_prob[2], (WarpParams*)params_scratch);
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(warp_new_frame_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, x, (const WarpParams*)params_scratch, B, T, C,
                       H, W, new_frames);
    MMVID_LAUNCH_CHECK("vid_warp_new_frames");
    return MMVID_OK;
}

extern "C" int mmvid_vid_warp_tokens(const int64_t* target_tok, const int64_t* new_frame_tok, const void* params, int B, int T,
                                     int n, int64_t* out, void* stream) {
    MMVID_REQUIRE(target_tok && new_frame_tok && params && out && B > 0 && T > 0 && n > 0, "vid_warp_tokens: bad arguments");
    const long total = (long)B * T * n;
    hipLaunchKernelGGL(warp_tokens_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)target_tok,
                       (const long long*)new_frame_tok, (const WarpParams*)params, B, T, n, (long long*)out);
    MMVID_LAUNCH_CHECK("vid_warp_tokens");
    return MMVID_OK;
}

extern "C" int mmvid_erase_tokens_choice(uint64_t seed, const float* step_dev, int nchoice, const float* cumprob,
                                         const int32_t* modes, const int32_t* boxes, int frame0_full, int B, int Tv, int f,
                                         int64_t value, int64_t* tok, void* stream) {
    MMVID_REQUIRE(cumprob && modes && boxes && tok && nchoice >= 1 && nchoice <= 4, "erase_tokens_choice: bad arguments");
    EraseChoice ch = {};
    ch.n = nchoice, ch.frame0_full = frame0_full;
    for (int i = 0; i < nchoice; ++i) {
        ch.cum[i] = cumprob[i], ch.mode[i] = modes[i];
        for (int e = 0; e < 4; ++e) ch.box[i][e] = boxes[i * 4 + e];
    }
    const long total = (long)B * Tv * f * f;
    if (total == 0) return MMVID_OK;
    hipLaunchKernelGGL(erase_choice_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, seed, step_dev, ch, B, Tv, f,
                       (long long)value, (long long*)tok);
    MMVID_LAUNCH_CHECK("erase_tokens_choice");
    return MMVID_OK;
}

extern "C" int mmvid_random_erase_tokens(uint64_t seed, const float* step_dev, int B, int Tv, int f, float p, float scale_lo,
                                         float scale_hi, float ratio_lo, float ratio_hi, int erase_half, int64_t value,
                                         int64_t* tok, void* stream) {
    MMVID_REQUIRE(tok && B > 0 && Tv > 0 && f > 0, "random_erase_tokens: bad arguments");
    hipLaunchKernelGGL(random_erase_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, seed, step_dev, Tv, f, p, scale_lo,
                       scale_hi, ratio_lo, ratio_hi, erase_half, (long long)value, (long long*)tok);
    MMVID_LAUNCH_CHECK("random_erase_tokens");
    return MMVID_OK;
}

extern "C" int mmvid_visual_color_jitter(uint64_t seed, const float* step_dev, float* x, int B, int Tv, int C, int H, int W, float p,
                                         int first_frame, float* params_out, void* stream) {
    MMVID_REQUIRE(x && B > 0 && Tv > 0 && C > 0 && H > 0 && W > 0, "visual_color_jitter: bad arguments");
    const long total = (long)B * Tv * C * H * W;
    hipLaunchKernelGGL(visual_color_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, seed, step_dev, B, Tv, C,
                       (long)H * W, p, first_frame, x, params_out);
    MMVID_LAUNCH_CHECK("visual_color_jitter");
    return MMVID_OK;
}

extern "C" int mmvid_counter_add(float* counter, float value, void* stream) {
    MMVID_REQUIRE(counter, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, value);
    MMVID_LAUNCH_CHECK("counter_add");
    return MMVID_OK;
}

extern "C" int mmvid_lr_schedule(const float* step_dev, int kind, float lr_min, float lr_max, int warmup_steps, int every,
                                 float* lr_out, void* stream) {
    MMVID_REQUIRE(step_dev && lr_out, "lr_schedule: null pointer");
    hipLaunchKernelGGL(lr_schedule_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, kind, lr_min, lr_max, warmup_steps,
                       every, lr_out);
    MMVID_LAUNCH_CHECK("lr_schedule");
    return MMVID_OK;
}
