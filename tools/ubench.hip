// VALU issue-rate microbenchmark for gfx950 (scratch tool, not part of the product).
// Prints SIMD cycles per wave64 instruction, assuming the clock given on argv[1] (GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 20000;
constexpr int UNROLL = 32;   // instructions per loop body (8 chains x 4)

#define BODY8(OP) \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define DEFINE(NAME, ASM3)                                                                 \
__global__ void k_##NAME(unsigned* out, unsigned seed) {                                   \
    unsigned a[8], b = seed + threadIdx.x, c = seed * 3 + 1;                               \
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 77 + threadIdx.x;                        \
    for (int it = 0; it < ITERS; ++it) {                                                   \
        _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i)                                  \
                asm volatile(ASM3 : "+v"(a[i]) : "v"(b), "v"(c));                          \
        }                                                                                  \
    }                                                                                      \
    unsigned r = 0;                                                                        \
    for (int i = 0; i < 8; ++i) r ^= a[i];                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                        \
}

DEFINE(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEFINE(pk_add_i16_clamp, "v_pk_add_i16 %0, %0, %1 clamp")
DEFINE(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
DEFINE(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
DEFINE(perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEFINE(max_i32, "v_max_i32 %0, %0, %1")
DEFINE(max3_i32, "v_max3_i32 %0, %0, %1, %2")
DEFINE(add_u32, "v_add_u32 %0, %0, %1")
DEFINE(sub_u32_clamp, "v_sub_u32_e64 %0, %0, %1 clamp")
DEFINE(add3_u32, "v_add3_u32 %0, %0, %1, %2")
DEFINE(pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
DEFINE(pk_add_f16, "v_pk_add_f16 %0, %0, %1")
DEFINE(pk_max_f16, "v_pk_max_f16 %0, %0, %1")
DEFINE(max3_i16, "v_max3_i16 %0, %0, %1, %2")
DEFINE(max_i16, "v_max_i16 %0, %0, %1")
DEFINE(max_u16_sdwa, "v_max_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
DEFINE(dot4_i32_i8, "v_dot4_i32_i8 %0, %0, %1, %2")
DEFINE(mov_b32, "v_mov_b32 %0, %1")
DEFINE(lshl_or_b32, "v_lshl_or_b32 %0, %0, 16, %1")
DEFINE(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
DEFINE(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEFINE(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEFINE(mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFINE(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
DEFINE(mix_pk_and_i32, "v_pk_max_i16 %0, %0, %1\n\tv_max_i32 %0, %0, %2")
DEFINE(med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEFINE(maximum3_f32, "v_maximum3_f32 %0, %0, %1, %2")
DEFINE(pk_max_i16_sgpr, "v_pk_max_i16 %0, %0, s4")
DEFINE(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE(min_u32, "v_min_u32 %0, %0, %1")
DEFINE(max_u32, "v_max_u32 %0, %0, %1")
DEFINE(sub_u32, "v_sub_u32 %0, %0, %1")
DEFINE(and_b32, "v_and_b32 %0, %0, %1")
DEFINE(xor_b32, "v_xor_b32 %0, %0, %1")
DEFINE(lshlrev_b32, "v_lshlrev_b32 %0, 1, %0")
DEFINE(max_f32, "v_max_f32 %0, %0, %1")
DEFINE(add_f32, "v_add_f32 %0, %0, %1")
DEFINE(max_u16, "v_max_u16 %0, %0, %1")
DEFINE(pk_min_i16, "v_pk_min_i16 %0, %0, %1")
DEFINE(add_u16, "v_add_u16 %0, %0, %1")
DEFINE(fmac_f32, "v_fmac_f32 %0, %1, %2")
DEFINE(add_u32_e64, "v_add_u32_e64 %0, %0, %1")
DEFINE(max_i32_e64, "v_max_i32_e64 %0, %0, %1")
DEFINE(sat_pk_sdwa_add_u8, "v_add_u16_sdwa %0, %0, %1 clamp dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0")


// ---- mixed streams: does a full-rate op stay full-rate next to half-rate ones? ----
#define DEFINE2(NAME, ASMA, ASMB, NA, NB)                                                  \
__global__ void k_##NAME(unsigned* out, unsigned seed) {                                   \
    unsigned a[8], d[8], b = seed + threadIdx.x, c = seed * 3 + 1;                         \
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 77 + threadIdx.x; d[i] = a[i] * 3; }   \
    for (int it = 0; it < ITERS; ++it) {                                                   \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                \
                if (i % ((NA) + (NB)) < (NA)) asm volatile(ASMA : "+v"(a[i]) : "v"(b), "v"(c)); \
                else asm volatile(ASMB : "+v"(d[i]) : "v"(b), "v"(c));                     \
            }                                                                              \
        }                                                                                  \
    }                                                                                      \
    unsigned r = 0;                                                                        \
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ d[i];                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                        \
}
DEFINE2(alt_add_max3, "v_add_u32 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %2", 1, 1)
DEFINE2(alt_add_add, "v_add_u32 %0, %0, %1", "v_sub_u32 %0, %0, %1", 1, 1)
DEFINE2(alt_3add_1max3, "v_add_u32 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %2", 3, 1)
DEFINE2(alt_1add_3max3, "v_add_u32 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %2", 1, 3)
DEFINE2(alt_add_pkmax, "v_add_u32 %0, %0, %1", "v_pk_max_i16 %0, %0, %1", 1, 1)
DEFINE2(alt_add_max3_2src, "v_add_u32 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %1", 1, 1)
DEFINE2(alt_maxu16_max3, "v_max_u16 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %2", 1, 1)
DEFINE2(alt_pkaddu16_max3, "v_pk_add_u16 %0, %0, %1", "v_pk_maximum3_f16 %0, %0, %1, %2", 1, 1)

struct Entry { const char* name; void (*fn)(unsigned*, unsigned); int perAsm; };
#define E(NAME, N) {#NAME, k_##NAME, N}

int main(int argc, char** argv) {
    double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    int wavesPerSimd = argc > 2 ? atoi(argv[2]) : 4;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    int blocks = cus * wavesPerSimd;     // 256 threads = 4 waves = one wave per SIMD per block
    unsigned* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    std::vector<Entry> es = { E(pk_max_i16,1), E(pk_add_i16_clamp,1), E(pk_sub_u16_clamp,1), E(pk_add_u16,1), E(perm_b32,1),
        E(max_i32,1), E(max3_i32,1), E(add_u32,1), E(sub_u32_clamp,1), E(add3_u32,1), E(pk_maximum3_f16,1), E(pk_add_f16,1),
        E(pk_max_f16,1), E(max3_i16,1), E(max_i16,1), E(max_u16_sdwa,1), E(dot4_i32_i8,1), E(mov_b32,1), E(lshl_or_b32,1),
        E(and_or_b32,1), E(fma_f32,1), E(pk_fma_f16,1), E(mad_i32_i24,1), E(pk_mad_i16,1), E(mix_pk_and_i32,2), E(med3_i32,1),
        E(maximum3_f32,1), E(pk_max_i16_sgpr,1), E(cndmask,1), E(min_u32,1), E(max_u32,1), E(sub_u32,1), E(and_b32,1), E(xor_b32,1), E(lshlrev_b32,1), E(max_f32,1), E(add_f32,1), E(max_u16,1), E(pk_min_i16,1), E(add_u16,1), E(fmac_f32,1), E(add_u32_e64,1), E(max_i32_e64,1), E(sat_pk_sdwa_add_u8,1), E(alt_add_max3,1), E(alt_add_add,1), E(alt_3add_1max3,1), E(alt_1add_3max3,1), E(alt_add_pkmax,1), E(alt_add_max3_2src,1), E(alt_maxu16_max3,1), E(alt_pkaddu16_max3,1) };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_add_u32, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    printf("CUs %d, %d waves/SIMD, clock assumed %.2f GHz\n", cus, wavesPerSimd, ghz);
    for (auto& e : es) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1u);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double instrPerWave = (double)ITERS * UNROLL * e.perAsm;
        double simdCycles = best * 1e-3 * ghz * 1e9;            // cycles elapsed on every SIMD
        double perInstr = simdCycles / (instrPerWave * wavesPerSimd);
        printf("%-22s %8.3f ms  %.2f cycles/instr/SIMD\n", e.name, best, perInstr);
    }
    return 0;
}
