// Round 5: what does it cost to free and allocate a multi-GB workspace buffer per search? (tools/scratch, GPU box)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
    hipFree(0);
    for (size_t gb : {1, 4, 15}) {
        for (int rep = 0; rep < 4; ++rep) {
            void* p = nullptr;
            auto t0 = std::chrono::steady_clock::now();
            if (hipMalloc(&p, gb << 30) != hipSuccess) { printf("malloc failed\n"); return 1; }
            auto t1 = std::chrono::steady_clock::now();
            hipMemsetAsync(p, 0, 1 << 20, 0);
            hipDeviceSynchronize();
            auto t2 = std::chrono::steady_clock::now();
            hipFree(p);
            auto t3 = std::chrono::steady_clock::now();
            printf("%zu GB: hipMalloc %.3f ms, hipFree %.3f ms\n", gb, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                   std::chrono::duration<double, std::milli>(t3 - t2).count());
        }
    }
    return 0;
}
