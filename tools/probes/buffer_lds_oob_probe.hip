// Does `buffer_load_dwordx4 ... offen lds` write ZEROS into LDS for lanes whose offset is out of range, and is the SGPR offset part of the
// range check?  (The conv kernels want: per-lane validity by an out-of-range voffset, the tap offset in soffset.)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/buffer_lds_oob_probe.hip -o tools/probes/buffer_lds_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* x, float* y, unsigned num_records, int soff) {
    __shared__ float sm[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) sm[i] = 123.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, num_records, 0x00020000);
    int voff = (threadIdx.x & 1) ? -1 : (int)threadIdx.x * 16;   // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)sm, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) y[i] = sm[i];
}
int main() {
    float *x, *y, h[4096], o[256];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMalloc(&x, sizeof(h)); hipMalloc(&y, sizeof(o));
    hipMemcpy(x, h, sizeof(h), hipMemcpyHostToDevice);
    struct { unsigned nr; int soff; const float* base; const char* what; } cases[] = {
        {0xFFFFFFFFu, 0, x, "num_records max, soffset 0"},
        {0xFFFFFFFFu, 4096, x, "num_records max, soffset 4096 (1024 floats)"},
        {1024u, 4096, x, "num_records 1024 B, soffset 4096: in range only if soffset is NOT checked"},
        {0xFFFFFFFFu, 4096, x + 1024, "base advanced by 1024 floats, then lanes read base[-...]? no: same as case 2 shifted"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c.base, y, c.nr, c.soff);
        hipMemcpy(o, y, sizeof(o), hipMemcpyDeviceToHost);
        printf("%s\n  lane0: %g %g %g %g | lane1 (oob): %g %g %g %g | lane2: %g %g | lane3 (oob): %g | lane 62: %g  lane 63 (oob): %g\n", c.what, o[0], o[1], o[2],
               o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[12], o[248], o[252]);
    }
    return 0;
}
