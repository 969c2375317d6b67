// gfx950 MFMA / LDS-DMA helpers shared by the GEMM and attention kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ec {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 16-bit element traits: DT = EC_F16 (0) or EC_BF16 (1)
template <int DT> struct T16;
template <> struct T16<0> {
    typedef _Float16 elem;
    typedef f16x8 v8;
    typedef f16x4 v4;
};
template <> struct T16<1> {
    typedef __bf16 elem;
    typedef bf16x8 v8;
    typedef bf16x4 v4;
};

// D[16x16] += A[16x32] * B[32x16]; lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15],
// D[(l>>4)*4+r][l&15].
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ _Float16 to16(float x, _Float16) { return (_Float16)x; }
__device__ __forceinline__ __bf16 to16(float x, __bf16) { return (__bf16)x; }

// 16-byte global -> LDS DMA (no VGPR destination).  lds must be wave-uniform: the
// hardware writes lane i's 16 bytes at lds + 16*i.
__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)g,
        (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// Reductions over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the
// gfx950 row-swap VALU ops instead of ds_bpermute round trips through LDS:
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the
// second, v_permlane32_swap the upper half of the first with the lower half of the second.
// (Written as inline asm: hipcc folds max / add over the two results of the swap builtins
// to one operand.  s_nop 1 covers the VALU-write -> v_permlane-read hazard.)
__device__ __forceinline__ float xor_max(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32 %0, %0, %1\n\t"
                 "v_mov_b32 %1, %0\n\t"
                 "s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"
                 "s_nop 1\n\t"
                 "v_max_f32 %0, %0, %1"
                 : "+v"(a), "+v"(b));
    return a;
}
__device__ __forceinline__ float xor_sum(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32 %0, %0, %1\n\t"
                 "v_mov_b32 %1, %0\n\t"
                 "s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32 %0, %0, %1"
                 : "+v"(a), "+v"(b));
    return a;
}

}  // namespace ec
