#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"p16 ret0", "p16 ret1", "p32 ret0", "p32 ret1"};
    for (int a = 0; a < 4; a++) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[a * 64 + i]); printf("\n"); }
    return 0;
}
