// Inter-sequence Smith-Waterman score kernel for gfx950.
//
// Replaces the SIMD inner loop of opalSearchDatabase (declared
// src/pyopal/opal.pxd:38-52; the SSE/AVX2 body lives in the absent
// vendor/opal) for mode = OPAL_MODE_SW, searchType = OPAL_SEARCH_SCORE.
//
// Mapping. SWIPE's "one SIMD lane = one database sequence" is widened to the
// 64-lane wavefront, and every lane carries TWO targets in the halves of a
// 32-bit VGPR (v_pk_*_i16 / v_pk_*_u16), so a wavefront advances 128 targets by
// one database column per pass over the query rows. The query rows of a strip
// (R <= 64) live in registers as H[R], E[R]; the substitution scores of the
// strip ("query profile", one row per residue symbol) live in LDS and are
// fetched with one ds_read_b128 per 8 rows per target. Longer queries are
// strip-mined; the last row of a strip is carried to the next strip through
// HBM in [column][lane] order (512 B per wavefront-column, coalesced).
//
// Per cell pair (two targets): 9 packed VALU ops + 1 v_perm_b32:
//   h   = sat_add_i16(Hdiag, s)        e/f are kept >= 0, so no explicit
//   h   = max(h, E[r]); h = max(h, f)  max(h, 0) is needed (SW floor)
//   best= max(best, h)
//   hmo = sat_sub_u16(h, open)
//   E[r]= max(sat_sub_u16(E[r], ext), hmo)
//   f   = max(sat_sub_u16(f, ext), hmo)
// A lane whose best reaches 32767 is flagged and recomputed at 32 bit by the
// intra-sequence kernel (the 8/16/32-bit ladder of the reference collapses to
// 16/32 here: gfx950 has no packed 8-bit max).
#include "common.h"

namespace miopal {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t pk_add_sat_i16(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}

// Arithmetic flavours of the cell update. Both keep E and F >= 0 so that the SW
// floor costs nothing.
//   ArithI16: saturating packed int16 (exact below 32767).
//   ArithF16: packed IEEE half (integers are exact below 2048) using gfx950's
//             v_pk_maximum3_f16, which folds two of the three max operations of a
//             cell into one instruction: 7.5 packed ops per cell pair instead of 9.
//             Lanes whose best reaches 2048 are flagged and recomputed wider.
// (ds_read_u16_d16 / _d16_hi cannot assemble the {A, B} score pair for free: with
// SRAM-ECC registers gfx950 d16 loads overwrite the whole VGPR; measured.)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t pk_add_f16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
static __device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    f16x2 r = __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
        __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(uint32_t, r);
}

struct ArithI16 {
    uint32_t open2, ext2;
    __device__ __forceinline__ ArithI16(int open, int ext)
        : open2((uint32_t)open * 0x00010001u), ext2((uint32_t)ext * 0x00010001u) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_sat_i16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const {
        return pk_max_i16(pk_max_i16(d, e), f);
    }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        (void)held; (void)r;
        best = pk_max_i16(best, h);
    }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_sub_sat_u16(h, open2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const {
        return pk_max_i16(pk_sub_sat_u16(x, ext2), hmo);
    }
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)half; }
    static constexpr int kLimit = 0x7fff;
};

struct ArithF16 {
    uint32_t negOpen2, negExt2;
    static __device__ __forceinline__ uint32_t pack(int v) {
        const _Float16 h = (_Float16)(float)v;
        return (uint32_t)__builtin_bit_cast(unsigned short, h) * 0x00010001u;
    }
    __device__ __forceinline__ ArithF16(int open, int ext)
        : negOpen2(pack(-min(open, 2048))), negExt2(pack(-min(ext, 2048))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_f16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max3_f16(d, e, f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        if (r & 1) best = pk_max3_f16(best, held, h);
        else held = h;
    }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_add_f16(h, negOpen2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const {
        return pk_max3_f16(pk_add_f16(x, negExt2), hmo, 0u);
    }
    static __device__ __forceinline__ int toInt(uint32_t half) {
        return (int)(float)__builtin_bit_cast(_Float16, (unsigned short)half);
    }
    static constexpr int kLimit = 2048;
};

// 16-byte slots per profile row in LDS: odd, so that the 16 lanes of a
// ds_read_b128 lane group that hold different symbols land on different slots.
template <int R>
struct ProfileLayout {
    static constexpr int kSlots = (R / 8) | 1;
};

constexpr int kWavesPerBlock = 4;

template <int R, bool MULTI, typename Arith>
__global__ __launch_bounds__(kWavesPerBlock * kLanes) void interseq_sw_score(InterseqArgs a) {
    constexpr int SLOTS = ProfileLayout<R>::kSlots;
    __shared__ uint4 lds[kWavesPerBlock][(kMaxAlphabet + 1) * SLOTS];

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * kWavesPerBlock + wave;
    if (g >= a.nGroups) return;  // wave-uniform; no block-level barrier below

    uint4* prof = lds[wave];
    const uint2* pack = a.pack + a.groupOff[g];
    const int nChunks = a.groupChunks[g];
    const Arith ar(a.gapOpen, a.gapExt);
    const uint4* gprof = reinterpret_cast<const uint4*>(a.profile);
    const int rowSlotsGlobal = a.qPad / 8;

    uint32_t best = 0, held = 0;
    uint32_t H[R], E[R];

    const int nStrips = MULTI ? a.nStrips : 1;
    for (int s = 0; s < nStrips; ++s) {
        // stage this strip's slice of the query profile into the wave's LDS region
        for (int idx = lane; idx < a.nSymbols * (R / 8); idx += kLanes) {
            const int t = idx / (R / 8), k = idx - t * (R / 8);
            prof[t * SLOTS + k] = gprof[t * rowSlotsGlobal + s * (R / 8) + k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int r = 0; r < R; ++r) {
            H[r] = 0;
            E[r] = 0;
        }

        const uint2* bin = nullptr;
        uint2* bout = nullptr;
        if (MULTI) {
            bin = a.boundary[(s + 1) & 1] + a.boundaryOff[g];
            bout = a.boundary[s & 1] + a.boundaryOff[g];
        }
        const bool hasIn = MULTI && s > 0;
        const bool hasOut = MULTI && s + 1 < nStrips;

        uint32_t hdiagTop = 0;  // H of the row above the strip, previous column
        uint2 cur = pack[lane];
        uint2 b0 = {0, 0}, b1 = {0, 0}, b2 = {0, 0}, b3 = {0, 0};
        if (hasIn) {
            b0 = bin[0 * kLanes + lane];
            b1 = bin[1 * kLanes + lane];
            b2 = bin[2 * kLanes + lane];
            b3 = bin[3 * kLanes + lane];
        }
        for (int c = 0; c < nChunks; ++c) {
            uint2 nxt = {0, 0};
            uint2 n0 = {0, 0}, n1 = {0, 0}, n2 = {0, 0}, n3 = {0, 0};
            if (c + 1 < nChunks) {
                nxt = pack[(size_t)(c + 1) * kLanes + lane];
                if (hasIn) {
                    const uint2* p = bin + (size_t)(c + 1) * 4 * kLanes + lane;
                    n0 = p[0 * kLanes];
                    n1 = p[1 * kLanes];
                    n2 = p[2 * kLanes];
                    n3 = p[3 * kLanes];
                }
            }
            uint32_t ra = cur.x, rb = cur.y;
#pragma unroll 1
            for (int cc = 0; cc < 4; ++cc) {
                const uint32_t tA = ra & 0xffu, tB = rb & 0xffu;
                ra >>= 8;
                rb >>= 8;
                const uint4* pa = prof + tA * SLOTS;
                const uint4* pb = prof + tB * SLOTS;
                uint32_t diag = hdiagTop;
                uint32_t f = 0;
                if (MULTI) {
                    hdiagTop = b0.x;  // H[-1][j] becomes the diagonal of the next column
                    f = b0.y;
                }
                // Profile rows are fetched one 8-row block ahead of their use; the
                // compiler barrier keeps hipcc from hoisting every ds_read_b128 to
                // the top of the column (which costs ~60 VGPRs and a wave of occupancy).
                constexpr int NB = R / 8;
                uint4 va[NB], vb[NB];
                va[0] = pa[0];
                vb[0] = pb[0];
                // {A's score, B's score} for query row r
                auto score = [&](int r) -> uint32_t {
                    const uint4 x = va[r >> 3], y = vb[r >> 3];
                    const int k = (r & 7) >> 1;
                    const uint32_t wa = k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                    const uint32_t wb = k == 0 ? y.x : k == 1 ? y.y : k == 2 ? y.z : y.w;
                    return __builtin_amdgcn_perm(wb, wa, (r & 1) ? 0x07060302u : 0x05040100u);
                };
                uint32_t dsum = ar.addScore(diag, score(0));
#pragma unroll
                for (int r8 = 0; r8 < NB; ++r8) {
                    if (r8 + 1 < NB) {
                        va[r8 + 1] = pa[r8 + 1];
                        vb[r8 + 1] = pb[r8 + 1];
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = r8 * 8 + k;
                        // consume the old H[r] (diagonal of row r+1) before H[r] is rewritten
                        uint32_t dnext = 0;
                        if (r + 1 < R) dnext = ar.addScore(H[r], score(r + 1));
                        const uint32_t h = ar.hmax(dsum, E[r], f);
                        ar.track(best, held, h, r);
                        const uint32_t hmo = ar.afterOpen(h);
                        E[r] = ar.gap(E[r], hmo);
                        f = ar.gap(f, hmo);
                        H[r] = h;
                        dsum = dnext;
                    }
                }
                if (hasOut) {
                    bout[((size_t)c * 4 + cc) * kLanes + lane] = make_uint2(H[R - 1], f);
                }
                if (MULTI) {
                    b0 = b1;
                    b1 = b2;
                    b2 = b3;
                }
            }
            cur = nxt;
            if (MULTI) {
                b0 = n0;
                b1 = n1;
                b2 = n2;
                b3 = n3;
            }
        }
        if (MULTI) {
            // the next strip of this wave reads what this strip wrote
            __threadfence();
        }
    }

    const int lo = Arith::toInt(best & 0xffffu), hi = Arith::toInt(best >> 16);
    const size_t base = (size_t)g * kGroupTargets;
    a.score[base + lane] = lo;
    a.score[base + kLanes + lane] = hi;
    a.overflow[base + lane] = lo >= Arith::kLimit;
    a.overflow[base + kLanes + lane] = hi >= Arith::kLimit;
}

template <int R, typename Arith>
static hipError_t launchR(const InterseqArgs& a, hipStream_t stream) {
    const int blocks = (a.nGroups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (a.nStrips > 1)
        hipLaunchKernelGGL((interseq_sw_score<R, true, Arith>), dim3(blocks), dim3(kWavesPerBlock * kLanes), 0, stream, a);
    else
        hipLaunchKernelGGL((interseq_sw_score<R, false, Arith>), dim3(blocks), dim3(kWavesPerBlock * kLanes), 0, stream, a);
    return hipGetLastError();
}

template <typename Arith>
static hipError_t launchArith(const InterseqArgs& a, int rowsPerStrip, hipStream_t stream) {
    switch (rowsPerStrip) {
        case 8: return launchR<8, Arith>(a, stream);
        case 16: return launchR<16, Arith>(a, stream);
        case 24: return launchR<24, Arith>(a, stream);
        case 32: return launchR<32, Arith>(a, stream);
        case 40: return launchR<40, Arith>(a, stream);
        case 48: return launchR<48, Arith>(a, stream);
        case 56: return launchR<56, Arith>(a, stream);
        case 64: return launchR<64, Arith>(a, stream);
    }
    return hipErrorInvalidValue;
}

// Rows per strip are a multiple of 8 (one ds_read_b128 = 8 16-bit scores).
// halfFloat selects the f16 arithmetic (profile must then hold f16 bit patterns).
hipError_t launchInterseqSwScore(const InterseqArgs& a, int rowsPerStrip, bool halfFloat, hipStream_t stream) {
    if (a.nGroups <= 0) return hipSuccess;
    return halfFloat ? launchArith<ArithF16>(a, rowsPerStrip, stream)
                     : launchArith<ArithI16>(a, rowsPerStrip, stream);
}

}  // namespace miopal
