// device_math.hpp — small float3 / half helpers for the gfx950 kernels.
//
// Arithmetic contract: every file including this header is compiled
// with -ffp-contract=off; a fused multiply-add happens exactly where fmaf() is written.
// Division and sqrtf are the correctly rounded HIP defaults.  This makes the TSDF kernels
// bit-reproducible against the IEEE restatement used by the parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfa {

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
// x*x' + y*y' + z*z' with the two trailing products fused (kfusion dot(), temp_utils.hpp:32-34)
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

// 3x3 row-major matrix + translation, passed by value as kernel arguments (SGPRs)
struct Mat3 {
    float m[9];
};
struct Aff3 {
    float m[9];
    float t[3];
};
__device__ __forceinline__ f3 mul(const Mat3& R, f3 v) {
    return mk3(dot(mk3(R.m[0], R.m[1], R.m[2]), v), dot(mk3(R.m[3], R.m[4], R.m[5]), v),
               dot(mk3(R.m[6], R.m[7], R.m[8]), v));
}
__device__ __forceinline__ f3 mulR(const Aff3& A, f3 v) {
    return mk3(dot(mk3(A.m[0], A.m[1], A.m[2]), v), dot(mk3(A.m[3], A.m[4], A.m[5]), v),
               dot(mk3(A.m[6], A.m[7], A.m[8]), v));
}
// kfusion normalized(): v * rsqrt(dot(v,v)), with the correctly rounded reciprocal square root
__device__ __forceinline__ f3 normalized(f3 v) { return v * (1.0f / sqrtf(dot(v, v))); }

// half <-> float: v_cvt_f16_f32 (round-to-nearest-even, f16 subnormals kept) / v_cvt_f32_f16
// The empty asm makes the float opaque to instruction selection: without it the backend folds
// a preceding fp32 multiply into v_fma_mixlo_f16, which rounds the exact product ONCE to half
// instead of fp32-then-half (observed on compute_dists; differs from __float2half_rn(a*b) in
// ~1e-4 of the pixels).
__device__ __forceinline__ uint32_t float_to_half_bits(float f) {
    asm("" : "+v"(f));
    _Float16 h = (_Float16)f;
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t bits) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffffu));
}

// TSDF voxel packing (kfusion pack_tsdf / unpack_tsdf): low half = tsdf (fp16), high = weight
__device__ __forceinline__ uint32_t pack_tsdf(float tsdf, int weight) {
    return float_to_half_bits(tsdf) | ((uint32_t)weight << 16);
}
__device__ __forceinline__ float unpack_tsdf(uint32_t v) { return half_bits_to_float(v); }

// 64-lane wave total by DPP (no LDS traffic, ~6 dependent VALU ops): Hillis-Steele inclusive scan
// inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 carry the row
// totals across rows; lane 63 ends up with the wave total, returned wave-uniformly.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_total(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive prefix sum over the 64 lanes, integers: the same six DPP steps (no LDS crossbar: a __shfl_up chain costs
// ~60 clocks per step, a DPP add ~8)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_addi(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_addi<0x111, 0xf>(v);  // row_shr:1
    v = dpp_addi<0x112, 0xf>(v);  // row_shr:2
    v = dpp_addi<0x114, 0xf>(v);  // row_shr:4
    v = dpp_addi<0x118, 0xf>(v);  // row_shr:8
    v = dpp_addi<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_addi<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// 64-lane wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;  // valid in every lane
}
__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace dfa
