#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (uint32_t)(2 * i) | ((uint32_t)(2 * i + 1) << 16);
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    uint32_t aA = base + (threadIdx.x & 63) * 20, aB = base + ((threadIdx.x * 7) & 63) * 20 + 4;
    uint32_t r0, r1;
    asm volatile("ds_read_u16_d16 %0, %2\n\tds_read_u16_d16_hi %0, %3\n\t"
                 "ds_read_u16_d16 %1, %2 offset:2\n\tds_read_u16_d16_hi %1, %3 offset:2\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1) : "v"(aA), "v"(aB) : "memory");
    out[threadIdx.x * 4 + 0] = r0;
    out[threadIdx.x * 4 + 1] = r1;
    const uint16_t* h = (const uint16_t*)lds;
    out[threadIdx.x * 4 + 2] = h[(aA - base) / 2] | ((uint32_t)h[(aB - base) / 2] << 16);
    out[threadIdx.x * 4 + 3] = h[(aA - base) / 2 + 1] | ((uint32_t)h[(aB - base) / 2 + 1] << 16);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) if (h[i*4] != h[i*4+2] || h[i*4+1] != h[i*4+3]) { if (bad < 5) printf("lane %d: %08x %08x vs %08x %08x\n", i, h[i*4], h[i*4+1], h[i*4+2], h[i*4+3]); bad++; }
    printf("bad %d\n", bad);
}
