// Optional per-launch timing with HIP events on the launching stream (used by bench.py for the roofline
// line; off by default: zero cost).  Classes are kernel families; flops are the ALGORITHMIC flops of the launch.
#pragma once
#include <hip/hip_runtime.h>

enum MmvidProfClass { PROF_GEMM_NT = 0, PROF_GEMM_NN = 1, PROF_GEMM_TN = 2, PROF_CONV = 3, PROF_ATTN_FWD = 4, PROF_ATTN_BWD = 5, PROF_NCLASS = 6 };

struct MmvidProfScope {
    int slot;
    hipStream_t s;
    MmvidProfScope(int cls, double flops, hipStream_t stream);
    ~MmvidProfScope();
};
