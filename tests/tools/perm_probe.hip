#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__global__ void k(unsigned* out) {
    unsigned v = threadIdx.x;
    u2 t = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    u2 s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[threadIdx.x] = t[0]; out[64 + threadIdx.x] = t[1]; out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"p32[0]", "p32[1]", "p16[0]", "p16[1]"};
    for (int a = 0; a < 4; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %u", h[a * 64 + i]); printf("\n"); }
    return 0;
}
