// Issue rate of the gfx950 matrix instructions the lo products could run on (round 6): one wave per SIMD,
// 16 independent accumulators, back-to-back issue; TFLOP/s over the whole chip and the ratio to v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_f8.hip -o /tmp/probe_mfma_f8 && /tmp/probe_mfma_f8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef long i64;

template <int MODE> __global__ __launch_bounds__(256) void probe(const int *src, float *out, int iters)
{
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) a[i][e] = src[(threadIdx.x * 4 + i) * 8 + e], b[i][e] = src[8192 + (threadIdx.x * 4 + i) * 8 + e];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if constexpr (MODE == 0) {          // f16 16x16x32
                    const f16x8 x = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[i], a[i], 0, 1, 2, 3));
                    const f16x8 y = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[j], b[j], 0, 1, 2, 3));
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[i * 4 + j], 0, 0, 0);
                } else if constexpr (MODE == 1) {   // scaled e4m3 x e4m3, 16x16x128
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else if constexpr (MODE == 2) {   // scaled e2m3 x e2m3 (FP6), 16x16x128
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else if constexpr (MODE == 3) {   // scaled fp4 x fp4
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else if constexpr (MODE == 4) {   // e4m3 x e2m3 (mixed)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 4 + j], 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                } else if constexpr (MODE == 5) {   // plain fp8 16x16x32
                    const i64 x = ((i64)a[i][1] << 32) | (unsigned)a[i][0], y = ((i64)b[j][1] << 32) | (unsigned)b[j][0];
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(x, y, acc[i * 4 + j], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    int *src;
    float *out;
    const int iters = 20000, blocks = 256;
    hipMalloc(&src, 16384 * 4 * 2);
    hipMalloc(&out, blocks * 256 * 4);
    std::vector<int> h(16384 * 2);
    const char *names[6] = {"v_mfma_f32_16x16x32_f16", "v_mfma_scale_f32_16x16x128_f8f6f4 e4m3 x e4m3", "... e2m3 x e2m3 (FP6)", "... fp4 x fp4",
                            "... e4m3 x e2m3", "v_mfma_f32_16x16x32_fp8_fp8"};
    const double kk[6] = {32, 128, 128, 128, 128, 32};
    for (int data = 0; data < 2; data++) {
        for (auto &v : h) v = data ? (rand() & 0x3f3f3f3f) | 0x20202020 : 0;      // random: small finite e4m3 / f16 values
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        double base = 0;
        for (int m = 0; m < 6; m++) {
            void (*k[6])(const int *, float *, int) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>};
            hipEvent_t e0, e1;
            hipEventCreate(&e0), hipEventCreate(&e1);
            k[m]<<<blocks, 256>>>(src, out, 200);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k[m]<<<blocks, 256>>>(src, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double tf = 2.0 * 16 * 16 * kk[m] * 16 * 4 * blocks * iters / ms / 1e9;
            if (m == 0) base = tf;
            printf("%s operands  %-48s: %8.3f ms  %7.0f TFLOP/s  %.2f x f16\n", data ? "random" : "zero  ", names[m], ms, tf, tf / base);
        }
    }
    return 0;
}
