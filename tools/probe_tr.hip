// Hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (used to decide the LDS layouts of the v2 kernels).
// Fills LDS with element index, every lane passes its own 8-byte-aligned address, prints what each lane received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane l addresses 4 contiguous elements at element offset 4*l (dense 64x4 = 256 elements)
    // mode 1: "4x16 row-major block per 16-lane group, row stride 16": lane i of group g -> row (i>>2), col 4*(i&3)
    // mode 2: row stride 64 elements (rows far apart): lane i of group g -> row (i>>2) of 4, col 4*(i&3) + 16*g
    int eoff;
    if (mode == 0) eoff = 4 * l;
    else if (mode == 1) eoff = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + 4 * (l & 3);
    else eoff = ((l & 15) >> 2) * 64 + 4 * (l & 3) + 16 * (l >> 4);
    const unsigned addr = (unsigned)(uintptr_t)(&lds[0]) + 2u * eoff;
    uint64_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
    const unsigned r0 = (unsigned)r, r1 = (unsigned)(r >> 32);
    out[l * 4 + 0] = r0 & 0xffff; out[l * 4 + 1] = r0 >> 16; out[l * 4 + 2] = r1 & 0xffff; out[l * 4 + 3] = r1 >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
