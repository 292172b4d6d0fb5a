// ds_read_b64_tr_b16 semantics probe (gfx950): every lane supplies the address of its own 8-byte chunk (chunk l = halves 4l..4l+3
// of an LDS array holding lds[i] = i); prints, per lane and returned element, the SOURCE (lane, element) the value came from.
// Build: hipcc -O2 --offload-arch=gfx950 tools/probes/tr16_probe.hip -o tools/probes/tr16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
    __shared__ __attribute__((aligned(16))) __fp16 lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = (__fp16)(float)i;
    __syncthreads();
    h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
    float* d;
    hipMalloc(&d, 256 * sizeof(float));
    k<<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (l%2d,e%d)", (int)h[l * 4 + j] / 4, (int)h[l * 4 + j] % 4);
        printf("\n");
    }
    return 0;
}
