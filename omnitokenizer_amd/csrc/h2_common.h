// Shared by the fp16-split kernels (gemm_h2.hip, attn_h2.hip): vector types and the power-of-two operand scale.
#pragma once
#include "common.h"

#include <cmath>

namespace omnitok {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// power-of-two scale s = 2^-e with bound * s in (2^14, 2^15]  (bound <= 0 or non-finite: s = 1)
__host__ __device__ __forceinline__ float h2_scale_of_bound(float bound) {
    if (!(bound > 0.0f) || !(bound < 3.0e38f)) return 1.0f;
    int x;
    (void)frexpf(bound, &x);  // bound = m * 2^x, m in [0.5, 1)  ->  bound <= 2^x
    int e = x - 15;
    if (e > 100) e = 100;
    if (e < -100) e = -100;
    return ldexpf(1.0f, -e);
}

}  // namespace omnitok
