// Mixed-stream VALU issue microbenchmark for gfx950 (scratch tool, not part of the product).
// Emulates one Smith-Waterman column of R cell pairs with H[R], E[R] in registers, for the
// candidate instruction mixes of the headline kernel, and prints SIMD cycles per cell pair.
// Also prints what v_pk_maximum3_f16 returns on NaN / inf / denormal bit patterns.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/ubench_mix && tools/ubench_mix [GHz] [waves/SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int R = 56;
constexpr int COLS = 4000;

// MIX 0: today's cell: 4 v_pk_add_f16 + 3 v_pk_maximum3_f16 (+ 0.5 for the running best)
// MIX 1: 4 v_add_u32 + 3.5 max3   (biased integer halves, unshifted)
// MIX 2: 3 v_add_u32 + 3.5 max3   (column-shifted: E extension free)
// MIX 3: MIX 2 + 1 v_add_u32      (row key for end locations)
// MIX 4: 3 v_pk_add_u16 + 3.5 max3 (is it the packed add or the half-float add that is slow?)
template <int MIX>
__global__ __launch_bounds__(256) void k_mix(unsigned* out, unsigned seed) {
    unsigned H[R], E[R];
    for (int r = 0; r < R; ++r) { H[r] = 0x08000800u + threadIdx.x + r; E[r] = 0x08000800u; }
    unsigned sc = 0x00010002u + seed, open2 = 0x00030003u, ext2 = 0x00010001u, fl = 0x08000800u;
    unsigned best = 0, held = 0;
    for (int j = 0; j < COLS; ++j) {
        unsigned f = fl, dsum = fl, h, hmo, t;
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(fl) : "v"(ext2));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned dnext;
            if (MIX == 0) {
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(dnext) : "v"(H[r]), "v"(sc));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(h) : "v"(dsum), "v"(E[r]), "v"(f));
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(hmo) : "v"(h), "v"(open2));
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(E[r]), "v"(ext2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(E[r]) : "v"(t), "v"(hmo), "v"(fl));
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(f), "v"(ext2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(f) : "v"(t), "v"(hmo), "v"(fl));
            } else if (MIX == 1) {
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(dnext) : "v"(H[r]), "v"(sc));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(h) : "v"(dsum), "v"(E[r]), "v"(f));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(hmo) : "v"(h), "v"(open2));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(E[r]), "v"(ext2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(E[r]) : "v"(t), "v"(hmo), "v"(fl));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(t) : "v"(f), "v"(ext2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(f) : "v"(t), "v"(hmo), "v"(fl));
            } else if (MIX == 2 || MIX == 3) {
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(dnext) : "v"(H[r]), "v"(sc));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(h) : "v"(dsum), "v"(E[r]), "v"(f));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(hmo) : "v"(h), "v"(open2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(E[r]) : "v"(E[r]), "v"(hmo), "v"(fl));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(t) : "v"(f), "v"(hmo), "v"(fl));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(f) : "v"(t), "v"(ext2));
                if (MIX == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(h) : "v"(ext2));
            } else {
                asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(dnext) : "v"(H[r]), "v"(sc));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(h) : "v"(dsum), "v"(E[r]), "v"(f));
                asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(hmo) : "v"(h), "v"(open2));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(E[r]) : "v"(E[r]), "v"(hmo), "v"(fl));
                asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(t) : "v"(f), "v"(hmo), "v"(fl));
                asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(f) : "v"(t), "v"(ext2));
            }
            if (r & 1) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(best) : "v"(held), "v"(h));
            else held = h;
            H[r] = h;
            dsum = dnext;
        }
    }
    unsigned x = best;
    for (int r = 0; r < R; ++r) x ^= H[r] ^ E[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

__global__ void k_semantics(const unsigned* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) {
        unsigned a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2], r;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        out[i] = r;
    }
}

template <int MIX>
static void run(const char* name, double perPair, double ghz, int wavesPerSimd, int cus, unsigned* out) {
    const int blocks = cus * wavesPerSimd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mix<MIX>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix<MIX>, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double simdCycles = best * 1e-3 * ghz * 1e9;
    const double perCellPair = simdCycles / ((double)COLS * R * wavesPerSimd);
    printf("%-44s %8.3f ms  %6.2f cycles per cell pair (%.1f instr) = %.2f cycles/instr -> %.2f TCUPS at 1024 SIMDs\n",
           name, best, perCellPair, perPair, perCellPair / perPair, 1024.0 * ghz * 128.0 / perCellPair / 1000.0);
}

int main(int argc, char** argv) {
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int wavesPerSimd = argc > 2 ? atoi(argv[2]) : 3;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    unsigned* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    printf("CUs %d, %d waves/SIMD, clock assumed %.2f GHz, R = %d\n", cus, wavesPerSimd, ghz, R);
    run<0>("4 pk_add_f16 + 3.5 pk_maximum3_f16 (today)", 7.5, ghz, wavesPerSimd, cus, out);
    run<1>("4 add/sub_u32 + 3.5 pk_maximum3_f16", 7.5, ghz, wavesPerSimd, cus, out);
    run<2>("3 add/sub_u32 + 3.5 pk_maximum3_f16 (shifted)", 6.5, ghz, wavesPerSimd, cus, out);
    run<3>("4 add/sub_u32 + 3.5 pk_maximum3_f16 (+row key)", 7.5, ghz, wavesPerSimd, cus, out);
    run<4>("3 pk_add/sub_u16 + 3.5 pk_maximum3_f16", 6.5, ghz, wavesPerSimd, cus, out);

    // semantics of v_pk_maximum3_f16 on the bit patterns the integer-halves scheme can meet
    std::vector<unsigned> in = {
        0x7C057C05u, 0x30003000u, 0x40004000u,   // NaN pattern first
        0x30003000u, 0x7C057C05u, 0x40004000u,   // NaN second
        0x30003000u, 0x40004000u, 0x7FFF7E01u,   // NaN third
        0x7C007C00u, 0x30003000u, 0x7BFF7BFFu,   // +inf
        0x00010002u, 0x00030001u, 0x00000000u,   // denormals
        0x03FF0400u, 0x04000001u, 0x00050005u,   // denormal / smallest normal
        0x08010802u, 0x08020801u, 0x08000800u,   // ordinary
        0x7BFF0400u, 0x04007BFFu, 0x08000800u,
    };
    const int n = (int)in.size() / 3;
    unsigned *din, *dout; CHECK(hipMalloc(&din, in.size() * 4)); CHECK(hipMalloc(&dout, n * 4));
    CHECK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, din, dout, n);
    std::vector<unsigned> res(n); CHECK(hipMemcpy(res.data(), dout, n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i)
        printf("maximum3(%08x, %08x, %08x) = %08x\n", in[3 * i], in[3 * i + 1], in[3 * i + 2], res[i]);
    return 0;
}
