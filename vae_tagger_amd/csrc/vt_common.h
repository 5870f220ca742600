// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VT_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define VT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float vt_silu(float x) {
    // x * sigmoid(x); __expf -> v_exp_f32, division -> v_rcp_f32 (error ~1 ulp, far below bf16 output)
    return x * __frcp_rn(1.0f + __expf(-x));
}

__device__ __forceinline__ float vt_sigmoid_accurate(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs
// (blocks b and b+8 share an L2).  Give each XCD a contiguous run of logical tiles so tiles that
// share operand panels hit the same L2.  Speed only; any placement is correct.
__device__ __forceinline__ int vt_xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
