// Native executor for the VQGAN encoder / decoder op sequence (taming Encoder/Decoder.forward,
// model.py:439-466, 551-582; vae.py:38-56).  The host builds the op list once per input shape
// (mmvid_amd/vae.py) -- convs, GroupNorm+swish, casts, spatial attention, VQ lookup, codebook gather, layout
// kernels, all on offsets into one arena -- and each call is then a single host->native transition that
// enqueues ~190 kernels back to back (the Python-per-op version was host-bound: 12 ms of launch overhead for
// 8.5 ms of GPU work per 48-frame encode).  No allocation, no sync: graph-capturable.
#include "../../include/mmvid_hip.h"
#include "common.h"
#include "graphs.h"

namespace {
inline char* at(void* arena, int64_t off) { return off < 0 ? nullptr : (char*)arena + off; }
}  // namespace

static int vqgan_enqueue(const mmvid_vqgan_op_t* ops, int nops, void* arena, void* stream) {
    for (int i = 0; i < nops; ++i) {
        const mmvid_vqgan_op_t& o = ops[i];
        int rc = 0;
        const bool strict = (o.flags & MMVID_VQFLAG_STRICT) != 0;
        if (strict) {  // fp32-accurate operators (strict.hip): every arena tensor of the plan is fp32
            switch (o.op) {
                case MMVID_VQOP_IMG2NHWC8:
                    rc = mmvid_image_to_nhwc4_f32((const float*)o.ext_in, o.N, o.H, o.W, (float*)at(arena, o.out_f32), stream);
                    break;
                case MMVID_VQOP_CONV:
                    rc = mmvid_conv2d_nhwc_f32(o.mode, (const float*)at(arena, o.in0), o.N, o.H, o.W, o.C, (const float*)o.w,
                                               o.b, o.Cout, (const float*)at(arena, o.in1), (o.flags >> 1) & 1,
                                               (float*)at(arena, o.out_f32), stream);
                    break;
                case MMVID_VQOP_GROUPNORM:
                    rc = mmvid_groupnorm_swish_nhwc_f32((const float*)at(arena, o.in0), o.N, (int64_t)o.H * o.W, o.C,
                                                        (const float*)o.w, o.b, o.eps, o.mode, (float*)at(arena, o.scratch),
                                                        (float*)at(arena, o.out_f32), stream);
                    break;
                case MMVID_VQOP_SPATIAL_ATTN:
                    rc = mmvid_spatial_attention_f32((const float*)at(arena, o.in0), (const float*)at(arena, o.in1),
                                                     (const float*)at(arena, o.in2), o.N, o.H * o.W, o.C, o.eps,
                                                     (float*)at(arena, o.scratch), (float*)at(arena, o.out_f32), stream);
                    break;
                case MMVID_VQOP_GATHER:
                    rc = mmvid_gather_rows((const float*)o.w, o.Cout, (const int64_t*)o.ext_in, (int64_t)o.N * o.H * o.W, o.C,
                                           (float*)at(arena, o.out_f32), nullptr, stream);
                    break;
                case MMVID_VQOP_EXT_CAST:
                    if (hipMemcpyAsync(at(arena, o.out_f32), o.ext_in, (size_t)o.N * o.H * o.W * o.C * 4, hipMemcpyDeviceToDevice,
                                       (hipStream_t)stream) != hipSuccess) {
                        mmvid_set_error("vqgan_run: memcpy failed");
                        rc = MMVID_ERR_HIP;
                    }
                    break;
                default:
                    mmvid_set_error("vqgan_run: op %d at %d has no strict form", o.op, i);
                    return MMVID_ERR_ARG;
            }
            if (rc) return rc;
            continue;
        }
        if (o.flags & MMVID_VQFLAG_SPLIT) {  // the bf16-pair operator (conv.hip: mmvid_conv2d_nhwc_split3)
            switch (o.op) {
                case MMVID_VQOP_IMG2NHWC8:
                    rc = mmvid_image_to_nhwc8_split((const float*)o.ext_in, o.N, o.H, o.W, at(arena, o.out_bf16), stream);
                    break;
                case MMVID_VQOP_CONV:
                    if ((o.flags & 8) && (o.flags & MMVID_VQFLAG_F16))  // one fp16 product (in0 = an fp16 tensor written by the GroupNorm below)
                        rc = mmvid_conv3x3_strip_nhwc_f16(at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout, (const float*)at(arena, o.in1),
                                                          (float*)at(arena, o.out_f32),
                                                          (o.flags & 4) ? (float*)at(arena, o.scratch) + (int64_t)o.N * o.Cout * 2 : nullptr,
                                                          at(arena, o.out_bf16), stream);
                    else if (o.flags & 8)
                        rc = mmvid_conv3x3_strip_nhwc_split3(at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout,
                                                             (const float*)at(arena, o.in1), (float*)at(arena, o.out_f32),
                                                             (o.flags & 4) ? (float*)at(arena, o.scratch) + (int64_t)o.N * o.Cout * 2 : nullptr,
                                                             at(arena, o.out_bf16), stream);
                    else
                        rc = mmvid_conv2d_nhwc_split3(o.mode, at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout,
                                                      (const float*)at(arena, o.in1), (o.flags >> 1) & 1, (float*)at(arena, o.out_f32),
                                                      (o.flags & 4) ? (float*)at(arena, o.scratch) + (int64_t)o.N * o.Cout * 2 : nullptr,
                                                      (o.flags & 32) ? 4 : 1, (o.flags & 32) ? (float*)at(arena, o.scratch) : nullptr,
                                                      stream);
                    break;
                case MMVID_VQOP_GROUPNORM:
                    rc = ((o.flags & MMVID_VQFLAG_F16) ? mmvid_groupnorm_swish_nhwc_f16out : mmvid_groupnorm_swish_nhwc_split)(
                        (const float*)at(arena, o.in0), o.N, (int64_t)o.H * o.W, o.C, (const float*)o.w, o.b, o.eps, o.mode,
                        (float*)at(arena, o.scratch), (o.flags & 2) ? (o.H * o.W) / ((o.flags & 8) ? 64 : 128) : 0, at(arena, o.out_bf16), stream);
                    break;
                case MMVID_VQOP_CAST:
                    rc = mmvid_split_f32_bf16x2((const float*)at(arena, o.in0), (int64_t)o.N * o.H * o.W * o.C, at(arena, o.out_bf16),
                                                stream);
                    break;
                default:
                    mmvid_set_error("vqgan_run: op %d at %d has no split form", o.op, i);
                    return MMVID_ERR_ARG;
            }
            if (rc) return rc;
            continue;
        }
        switch (o.op) {
            case MMVID_VQOP_IMG2NHWC8:
                rc = mmvid_image_to_nhwc8((const float*)o.ext_in, o.N, o.H, o.W, at(arena, o.out_bf16), stream);
                break;
            case MMVID_VQOP_CONV:
                if (o.flags & 8) {  // 3x3 stride-1 convolution in strip form (the planner checked the geometry)
                    rc = mmvid_conv3x3_strip_nhwc(at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout,
                                                  (o.flags & 1) ? nullptr : at(arena, o.in1),
                                                  (o.flags & 1) ? (const float*)at(arena, o.in1) : nullptr, at(arena, o.out_bf16),
                                                  (float*)at(arena, o.out_f32),
                                                  (o.flags & 4) ? (float*)at(arena, o.scratch) + (int64_t)o.N * o.Cout * 2 : nullptr,
                                                  stream);
                    break;
                }
                if (o.flags & 32) {  // deep layer on a small map: split-K by 4, fixed-order reduce (workspace at scratch)
                    rc = mmvid_conv2d_nhwc_splitk(o.mode, at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout,
                                                  (o.flags & 1) ? nullptr : at(arena, o.in1),
                                                  (o.flags & 1) ? (const float*)at(arena, o.in1) : nullptr, (o.flags >> 1) & 1,
                                                  at(arena, o.out_bf16), (float*)at(arena, o.out_f32), nullptr, 4,
                                                  (float*)at(arena, o.scratch), stream);
                    break;
                }
                rc = mmvid_conv2d_nhwc(o.mode, at(arena, o.in0), o.N, o.H, o.W, o.C, o.w, o.b, o.Cout,
                                       (o.flags & 1) ? nullptr : at(arena, o.in1),
                                       (o.flags & 1) ? (const float*)at(arena, o.in1) : nullptr, (o.flags >> 1) & 1,
                                       at(arena, o.out_bf16), (float*)at(arena, o.out_f32),
                                       (o.flags & 4) ? (float*)at(arena, o.scratch) + (int64_t)o.N * o.Cout * 2 : nullptr,
                                       stream);
                break;
            case MMVID_VQOP_GROUPNORM:
                rc = mmvid_groupnorm_swish_nhwc(at(arena, o.in0), (o.flags & 1) ? 0 : 1, o.N, (int64_t)o.H * o.W, o.C,
                                                (const float*)o.w, o.b, o.eps, o.mode, (float*)at(arena, o.scratch),
                                                (o.flags & 2) ? (o.H * o.W) / ((o.flags & 8) ? 64 : 128) : 0, at(arena, o.out_bf16),
                                                (float*)at(arena, o.out_f32), stream);
                break;
            case MMVID_VQOP_CAST:
                rc = mmvid_cast_f32_to_bf16((const float*)at(arena, o.in0), at(arena, o.out_bf16),
                                            (int64_t)o.N * o.H * o.W * o.C, stream);
                break;
            case MMVID_VQOP_SPATIAL_ATTN:
                rc = mmvid_spatial_attention_ld(at(arena, o.in0), at(arena, o.in1), at(arena, o.in2), o.pad > 0 ? o.pad : o.C, o.N,
                                                o.H * o.W, o.C, o.eps, (float*)at(arena, o.scratch), at(arena, o.out_bf16), stream);
                break;
            case MMVID_VQOP_VQ_ARGMIN:
                rc = mmvid_vq_argmin_l2((const float*)at(arena, o.in0), (const float*)o.w, o.b, (int64_t)o.N * o.H * o.W,
                                        o.Cout, o.C, (int64_t*)o.ext_out, nullptr, stream);
                break;
            case MMVID_VQOP_GATHER:
                rc = mmvid_gather_rows((const float*)o.w, o.Cout, (const int64_t*)o.ext_in, (int64_t)o.N * o.H * o.W, o.C,
                                       nullptr, at(arena, o.out_bf16), stream);
                break;
            case MMVID_VQOP_EXT_CAST:
                rc = mmvid_cast_f32_to_bf16((const float*)o.ext_in, at(arena, o.out_bf16), (int64_t)o.N * o.H * o.W * o.C, stream);
                break;
            case MMVID_VQOP_NHWC2NCHW:
                rc = mmvid_nhwc_to_nchw_f32((const float*)at(arena, o.in0), o.N, o.H, o.W, o.C, o.Cout, (float*)o.ext_out,
                                            stream);
                break;
            default:
                mmvid_set_error("vqgan_run: unknown op %d at %d", o.op, i);
                return MMVID_ERR_ARG;
        }
        if (rc) return rc;
    }
    return MMVID_OK;
}

extern "C" int mmvid_vqgan_run(const mmvid_vqgan_op_t* ops, int nops, void* arena, void* stream) {
    MMVID_REQUIRE(ops && arena && nops >= 0, "vqgan_run: bad arguments");
    uint64_t key = mmvid_hash_bytes(ops, sizeof(mmvid_vqgan_op_t) * (size_t)nops, 0xcbf29ce484222325ull ^ 1);
    key = mmvid_hash_ptr(arena, mmvid_hash_ptr(stream, key));
    return mmvid_run_cached(key, (hipStream_t)stream, [=](hipStream_t s) { return vqgan_enqueue(ops, nops, arena, (void*)s); });
}
