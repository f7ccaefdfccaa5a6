// Inter-sequence DP kernel for gfx950: scores of one query against 128 targets
// per wavefront, for all four alignment modes. Included by the interseq_*.hip
// translation units, one per arithmetic flavour (keeps hipcc builds parallel).
//
// Replaces the SIMD inner loop of opalSearchDatabase (declared
// src/pyopal/opal.pxd:38-52; the SSE/AVX2 body lives in the absent
// vendor/opal) for searchType = OPAL_SEARCH_SCORE.
//
// Mapping. SWIPE's "one SIMD lane = one database sequence" is widened to the
// 64-lane wavefront, and every lane carries TWO targets in the halves of a
// 32-bit VGPR (packed 16-bit arithmetic), so a wavefront advances 128 targets
// by one database column per pass over the query rows. The query rows of a
// strip (R <= 64) live in registers as H[R], E[R]; the substitution scores of
// the strip ("query profile", one row per residue symbol) live in LDS and are
// fetched with one ds_read_b128 per 8 rows per target.
//
// Long queries are cut into strips of R rows. The W wavefronts of a workgroup
// work on the SAME 128 targets, wavefront w on strip w of the current round,
// one 4-column chunk behind wavefront w-1 (a software pipeline along the
// anti-diagonal of (strip, chunk)); the last row of a strip reaches the next
// strip through a 2 KB LDS buffer, and only the hop from strip W-1 to strip 0 of
// the next round goes through HBM, in [column][lane] order (512 B per
// wavefront-column, coalesced). With W = 1 this degenerates to one wavefront per
// group walking its strips in turn.
//
// Cell update, per pair of targets:
//   h    = max(Hdiag + s, E[r], F)
//   hmo  = h - open
//   E[r] = max(E[r] - ext, hmo)        (gap consuming target residues)
//   F    = max(F - ext, hmo)           (gap consuming query residues)
// Arithmetic flavours (struct Arith*):
//   ArithSwF16  Smith-Waterman, packed half floats (integers exact below 2048);
//               E and F are floored at 0, so H needs no floor of its own, and
//               v_pk_maximum3_f16 folds two max per cell: 7.5 ops + 1 v_perm.
//   ArithSwI16  Smith-Waterman, saturating int16 with unsigned-saturating
//               subtraction for the same floor: 9 ops + 1 v_perm.
//   ArithI16    signed saturating int16 for NW / HW / OV and for the anchored
//               reverse pass: 8 ops + 1 v_perm (+1 when every cell is a candidate).
// A lane that reaches the limit of its flavour is flagged; the host recomputes
// it with the next rung (int16, then the int32 intra-sequence kernel): the
// reference's 8/16/32-bit ladder (src/pyopal/lib.pyx:1283-1289) on hardware that
// has no packed 8-bit max. (ds_read_u16_d16[_hi] cannot build the {A, B} score
// pair for free: with SRAM-ECC registers gfx950 d16 loads overwrite the whole
// VGPR; measured.)
#pragma once
#include <algorithm>

#include <type_traits>

#include "common.h"

namespace miopal {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ uint32_t pk_add_sat_i16(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t pk_sub_sat_i16(uint32_t a, uint32_t b) {
    s16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b));
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
static __device__ __forceinline__ uint32_t pk_add_f16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b));
}
static __device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    f16x2 r = __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
        __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t pk_max2_f16(uint32_t a, uint32_t b) {
    f16x2 r = __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t dup16(int v) { return ((uint32_t)v & 0xffffu) * 0x00010001u; }

// ---- arithmetic flavours --------------------------------------------------------
struct ArithSwI16 {
    static constexpr bool kFloor = true;    // Smith-Waterman
    static constexpr bool kDiag = false;
    static constexpr int kLimit = 0x7fff;   // a best at or above this may have clipped
    static constexpr bool kColShift = false;    // values of column j carry j * ext (ArithSwU16)
    static constexpr bool kStoresOpen = false;  // H[r] keeps h (true: h - (open - ext), ArithU16Diag)
    static constexpr bool kWeakPad = false;     // padding scores cannot win a max on their own
    static __device__ __forceinline__ uint32_t answerLowest() { return lowest(); }
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t) const { return v; }
    uint32_t open2, ext2;
    __device__ __forceinline__ ArithSwI16(int open, int ext) : open2(dup16(min(open, 32767))), ext2(dup16(min(ext, 32767))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_sat_i16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max_i16(pk_max_i16(d, e), f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        (void)held; (void)r;
        best = pk_max_i16(best, h);
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const { return pk_max_i16(a, b); }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_sub_sat_u16(h, open2); }
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return pk_sub_sat_u16(h, open2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max_i16(pk_sub_sat_u16(x, ext2), hmo); }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return dup16(max(v, 0)); }
    static __device__ __forceinline__ uint32_t lowest() { return 0u; }
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)(short)half; }
};

struct ArithSwF16 {
    static constexpr bool kFloor = true;
    static constexpr bool kDiag = false;
    static constexpr int kLimit = 2048;
    static constexpr bool kColShift = false;    // values of column j carry j * ext (ArithSwU16)
    static constexpr bool kStoresOpen = false;  // H[r] keeps h (true: h - (open - ext), ArithU16Diag)
    static constexpr bool kWeakPad = false;     // padding scores cannot win a max on their own
    static __device__ __forceinline__ uint32_t answerLowest() { return lowest(); }
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t) const { return v; }
    uint32_t negOpen2, negExt2;
    static __device__ __forceinline__ uint32_t pack(int v) {
        const _Float16 h = (_Float16)(float)v;
        return (uint32_t)__builtin_bit_cast(unsigned short, h) * 0x00010001u;
    }
    __device__ __forceinline__ ArithSwF16(int open, int ext) : negOpen2(pack(-min(open, 2048))), negExt2(pack(-min(ext, 2048))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_f16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max3_f16(d, e, f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        if (r & 1) best = pk_max3_f16(best, held, h);
        else held = h;
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const { return pk_max3_f16(a, b, b); }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_max3_f16(pk_add_f16(h, negOpen2), 0u, 0u); }
    // inside a cell the floor comes with the max3 of gap(), so the open step needs none
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return pk_add_f16(h, negOpen2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max3_f16(pk_add_f16(x, negExt2), hmo, 0u); }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return pack(max(v, 0)); }
    static __device__ __forceinline__ uint32_t lowest() { return 0u; }
    static __device__ __forceinline__ int toInt(uint32_t half) {
        return (int)(float)__builtin_bit_cast(_Float16, (unsigned short)half);
    }
};

struct ArithI16 {
    static constexpr bool kFloor = false;   // NW / HW / OV / anchored reverse pass
    static constexpr bool kDiag = false;
    static constexpr int kLimit = 0x7fff;   // ranges are checked statically by the host
    static constexpr bool kColShift = false;    // values of column j carry j * ext (ArithSwU16)
    static constexpr bool kStoresOpen = false;  // H[r] keeps h (true: h - (open - ext), ArithU16Diag)
    static constexpr bool kWeakPad = false;     // padding scores cannot win a max on their own
    static __device__ __forceinline__ uint32_t answerLowest() { return lowest(); }
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t) const { return v; }
    uint32_t open2, ext2;
    __device__ __forceinline__ ArithI16(int open, int ext) : open2(dup16(min(open, 32767))), ext2(dup16(min(ext, 32767))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_sat_i16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max_i16(pk_max_i16(d, e), f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        (void)held; (void)r;
        best = pk_max_i16(best, h);
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const { return pk_max_i16(a, b); }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_sub_sat_i16(h, open2); }
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return pk_sub_sat_i16(h, open2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max_i16(pk_sub_sat_i16(x, ext2), hmo); }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return dup16(max(v, -32768)); }
    static __device__ __forceinline__ uint32_t lowest() { return 0x80008000u; }  // acts as -infinity
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)(short)half; }
};

// Signed int16 on anti-diagonally shifted values: every cell holds X + (i + j) * ext.
// Extending a gap then costs nothing (the shift of the next cell absorbs the ext), both
// gap kinds open with the same h + (ext - open), and the diagonal step's 2 * ext is
// folded into the query profile: 6 ops per cell pair instead of 8. Answers are shifted
// back where they are read. Needs (Q + L) * ext of extra head-room (checked by the host).
struct ArithI16Diag {
    static constexpr bool kFloor = false;
    static constexpr bool kDiag = true;
    static constexpr int kLimit = 0x7fff;
    static constexpr bool kColShift = false;    // values of column j carry j * ext (ArithSwU16)
    static constexpr bool kStoresOpen = false;  // H[r] keeps h (true: h - (open - ext), ArithU16Diag)
    static constexpr bool kWeakPad = false;     // padding scores cannot win a max on their own
    static __device__ __forceinline__ uint32_t answerLowest() { return lowest(); }
    // packed true values of stored cells whose shift is `shift2`
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t shift2) const { return pk_sub_sat_i16(v, shift2); }
    uint32_t extMinusOpen2;
    __device__ __forceinline__ ArithI16Diag(int open, int ext) : extMinusOpen2(dup16(max(ext - open, -32768))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return pk_add_sat_i16(h, s); }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max_i16(pk_max_i16(d, e), f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        (void)held; (void)r;
        best = pk_max_i16(best, h);
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const { return pk_max_i16(a, b); }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return pk_add_sat_i16(h, extMinusOpen2); }
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return pk_add_sat_i16(h, extMinusOpen2); }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max_i16(x, hmo); }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return dup16(max(v, -32768)); }
    static __device__ __forceinline__ uint32_t lowest() { return 0x80008000u; }
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)(short)half; }
};

// NW / HW / OV on anti-diagonally shifted values like ArithI16Diag, as unsigned patterns compared as
// half floats (see "biased integer halves" further down; same idea in the general kernel, round 2).
// Plain form of a value x of cell (i, j): P(x) = kU16Zero + x + (i + j) * ext. H[r] keeps the STORED
// form S(h) = P(h) - c, c = open - ext >= 0, which is at once the value both gap kinds open with, so:
//     d = S(Hd) + s''          s'' = s + 2 ext + c >= 0 (host: open + ext >= -min S): a plain integer add
//     h = max3(d, E, F)         one v_pk_maximum3_f16
//     hmo = h - c               integer subtract; also the next column's stored H
//     E = max(E, hmo), F = max(F, hmo)
// 2 integer adds + 3 half-float max + 1 v_perm per cell pair, against 2 saturating packed adds + 4
// packed max + 1 v_perm. Padding (symbol and rows) scores c after the shift, so that chains of padding
// cells never sink below lowest() + c and `h - c` can neither borrow from the other half nor leave the
// normal half floats; it can therefore reach real
// magnitudes, and the places that read answers mask rows and columns beyond the pair explicitly
// (kWeakPad). The host checks the static range (every pattern within [0, 0x7BFF]); answers are turned
// into packed true int16 values where they are read (toTrue).
constexpr int kU16Zero = 0x1000;
static_assert(kU16Zero == kUnsignedDiagZero, "common.h mirrors this");
struct ArithU16Diag {
    static constexpr bool kFloor = false;
    static constexpr bool kDiag = true;
    static constexpr int kLimit = 0x7fff;
    static constexpr bool kColShift = false;    // values of column j carry j * ext (ArithSwU16)
    static constexpr bool kStoresOpen = true;
    static constexpr bool kWeakPad = true;
    int c;
    uint32_t c2, trueBias2;
    __device__ __forceinline__ ArithU16Diag(int open, int ext)
        : c(open - ext), c2((uint32_t)(open - ext) * 0x00010001u), trueBias2(dup16(kU16Zero - (open - ext))) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return h + s; }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max3_f16(d, e, f); }
    __device__ __forceinline__ void track(uint32_t& best, uint32_t& held, uint32_t h, int r) const {
        (void)held; (void)r;
        best = pk_max3_f16(best, h, h);
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const { return pk_max_i16(a, b); }  // answers: true values
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return h; }   // stored form = opened plain form
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return h - c2; }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max3_f16(x, hmo, hmo); }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return (uint32_t)(kU16Zero + v - c) * 0x00010001u; }
    // ("minus infinity" of the cells: the smallest NORMAL half-float pattern. Patterns below 0x0400 are
    // denormals: they compare correctly, but a database with ragged groups - a third of its cells
    // padding, all of them down there - ran 20 % slower with 0 here.)
    static __device__ __forceinline__ uint32_t lowest() { return 0x04000400u; }
    static __device__ __forceinline__ uint32_t answerLowest() { return 0x80008000u; }
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)(short)half; }
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t shift2) const {
        const u16x2 r = __builtin_bit_cast(u16x2, v) - (__builtin_bit_cast(u16x2, shift2) + __builtin_bit_cast(u16x2, trueBias2));
        return __builtin_bit_cast(uint32_t, r);
    }
};

// Smith-Waterman in the general kernel on the representation of the pair-table kernel ("biased
// integer halves" further down), round 2: unsigned patterns compared as half floats, values of
// column j shifted by j * ext. Plain form of a value x of column j: P(x) = kSwU16Zero + x + j ext;
// H[r] keeps the stored form S(h) = P(h) - K with K >= 0 chosen by the host so that the profile
// entries s'' = s + ext + K are not negative (v_perm builds the pair of scores from 16-bit halves:
// a plain integer add over both halves needs them non-negative):
//     d = S(Hd) + s''                     plain form, on this column's scale
//     h = max3(d, E, F)
//     S(h) = h - K;  hmo = h - (open - ext)
//     E = max3(E, hmo, zero');  F = max3(F, hmo, zero') - ext        zero' = the next column's zero
// 4 integer adds + 3.5 max3 + 1 v_perm against 4 v_pk_add_f16 + 3.5 max3 + 1 v_perm, and an exact
// range of a.biasedLimit (tens of thousands minus ext x the longest packed target) instead of 2048:
// a 300-residue query's strong hits no longer send the whole view to the int16 rung. The all-cells
// maximum is folded per column and unshifted there (an integer max of true values, sticky for the
// inf / NaN patterns of a lane that left the range). No rebasing: the host only picks this flavour
// when zero + ext x (longest packed target) leaves a range worth having. Scores only (the end-location
// form keeps ArithSwF16: its row scans compare packed values of different columns).
constexpr int kSwU16Zero = 0x1000;
static_assert(kSwU16Zero == kSwShiftZero, "common.h mirrors this");
struct ArithSwU16 {
    static constexpr bool kFloor = true;
    static constexpr bool kDiag = false;
    static constexpr int kLimit = 0x7fff;       // (unused: the limit of this flavour is a.biasedLimit)
    static constexpr bool kColShift = true;
    static constexpr bool kStoresOpen = false;
    static constexpr bool kWeakPad = false;
    uint32_t c2, k2, ext2;
    __device__ __forceinline__ ArithSwU16(int open, int ext, int K = 0)
        : c2((uint32_t)(open - ext) * 0x00010001u), k2((uint32_t)K * 0x00010001u), ext2((uint32_t)ext * 0x00010001u) {}
    __device__ __forceinline__ uint32_t addScore(uint32_t h, uint32_t s) const { return h + s; }
    __device__ __forceinline__ uint32_t hmax(uint32_t d, uint32_t e, uint32_t f) const { return pk_max3_f16(d, e, f); }
    __device__ __forceinline__ void track(uint32_t& cm, uint32_t& held, uint32_t h, int r) const {
        if (r & 1) cm = pk_max3_f16(cm, held, h);
        else held = h;
    }
    __device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) const {   // answers: true values
        const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
        return __builtin_bit_cast(uint32_t, r);
    }
    __device__ __forceinline__ uint32_t afterOpen(uint32_t h) const { return h; }   // (borders are set up by hand)
    __device__ __forceinline__ uint32_t cellOpen(uint32_t h) const { return h - c2; }
    __device__ __forceinline__ uint32_t gap(uint32_t x, uint32_t hmo) const { return pk_max3_f16(x, hmo, hmo); }
    __device__ __forceinline__ uint32_t stored(uint32_t h) const { return h - k2; }
    __device__ __forceinline__ uint32_t fromInt(int v) const { return (uint32_t)v * 0x00010001u; }
    static __device__ __forceinline__ uint32_t lowest() { return 0u; }             // of the answers: true values
    static __device__ __forceinline__ uint32_t answerLowest() { return 0u; }
    static __device__ __forceinline__ int toInt(uint32_t half) { return (int)(half & 0xffffu); }
    __device__ __forceinline__ uint32_t toTrue(uint32_t v, uint32_t) const { return v; }
};

// 16-byte slots per profile row in LDS: odd, so that the 16 lanes of a
// ds_read_b128 lane group that hold different symbols land on different slots.
template <int R>
struct ProfileLayout {
    static constexpr int kSlots = (R / 8) | 1;
};

// take the halves of `v` named by the 0 / 0xffff masks in `m`, keep `dst` elsewhere
static __device__ __forceinline__ uint32_t selectHalves(uint32_t dst, uint32_t v, uint32_t m) {
    return (dst & ~m) | (v & m);
}

template <typename Arith>
static __device__ __forceinline__ Arith makeArith(const InterseqArgs& a) {
    if constexpr (Arith::kColShift) return Arith(a.gapOpen, a.gapExt, a.scoreBias);
    else return Arith(a.gapOpen, a.gapExt);
}

// MULTI = false: the query fits one strip (no boundary traffic, no rounds).
// LOC = true: also report where the answer was found (OPAL_SEARCH_SCORE_END): the
// first maximum when candidates are visited target column by target column and,
// inside a column, query row by query row (oracle/opal_oracle.c).
template <int R, typename Arith, int W, bool TRACK_ALL, bool MULTI, bool LOC, bool UNITS = false>
__global__ __launch_bounds__(W * kLanes)
void interseq_kernel(InterseqArgs a) {
    static_assert(MULTI || W == 1, "a single strip needs a single wavefront");
    static_assert(!UNITS || (MULTI && !LOC), "unit mode: scores of multi-strip queries");
    constexpr int kLowInt = Arith::kFloor ? 0 : INT32_MIN;
    static_assert(!(Arith::kDiag && TRACK_ALL), "the shifted flavour has no all-cells maximum");
    constexpr bool kRegions = !Arith::kFloor;  // Smith-Waterman flavours only know the all-cells maximum
    constexpr int SLOTS = ProfileLayout<R>::kSlots;
    constexpr int NB = R / 8;
    constexpr int WB = W > 1 ? W : 1;
    __shared__ uint4 ldsProf[W][(kMaxAlphabet + 1) * SLOTS];
    __shared__ uint2 ldsBnd[WB][2][W > 1 ? 4 * kLanes : 1];
    __shared__ uint32_t ldsOut[WB][W > 1 ? kLanes : 1];
    __shared__ int ldsLoc[(LOC && W > 1) ? W : 1][(LOC && W > 1) ? 10 : 1][(LOC && W > 1) ? kLanes : 1];

    const int wave = W > 1 ? (int)(threadIdx.x >> 6) : 0;
    const int lane = threadIdx.x & 63;
    const int nStrips = MULTI ? a.nStrips : 1;
    const int nRounds = (nStrips + W - 1) / W;
    // Unit mode (scores only, several rounds of strips per group): a workgroup does not own a group
    // for all its rounds but takes (group, round) units from a counter, rounds in order. The rounds
    // of a group already hand their last row over through HBM; a group of a 2000-residue query is
    // 4 rounds, and 782 such groups on 256 resident workgroups are 3.05 workgroup-lifetimes, i.e. 4:
    // with units of one round the last quarter of the launch is no longer three quarters idle.
    // The few answers a wavefront carries from round to round travel through a.unitPartial.
    __shared__ int ldsUnit;
    // (a template parameter: the extra scalars it keeps alive cost the classic form 5 % when both
    // shared one instantiation)
    constexpr bool unitMode = UNITS;
    int unitRound = 0;
next_unit:
    int groupIndex = blockIdx.x;
    if (unitMode) {
        if (threadIdx.x == 0) ldsUnit = atomicAdd(a.unitCounter, 1);
        __syncthreads();
        const int u = ldsUnit;
        __syncthreads();
        if (u >= a.nGroups * nRounds) return;
        groupIndex = u % a.nGroups;     // rounds in order: round r of every group before round r + 1 of any
        unitRound = u / a.nGroups;
        if (unitRound > 0) {
            // the round before this one, done by whoever took that unit (earlier than this one)
            if (threadIdx.x == 0)
                while (__hip_atomic_load(a.unitFlags + groupIndex, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < unitRound)
                    __builtin_amdgcn_s_sleep(4);
            __syncthreads();
        }
    }
    const int g = groupIndex + a.groupBase;  // one group of 128 targets per workgroup (and unit)

    uint4* prof = ldsProf[wave];
    const uint2* pack = a.pack + a.groupOff[g];
    // (a leading group whose longest targets were handed to the int32 kernel stops at the longest that stays)
    const int nChunks = groupIndex < a.capGroups ? min(a.groupChunks[g], a.capChunks) : a.groupChunks[g];
    if (nChunks > a.priorityChunks) __builtin_amdgcn_s_setprio(3);  // long group: critical path
    else if (unitMode) __builtin_amdgcn_s_setprio(0);
    const Arith ar = makeArith<Arith>(a);
    const uint4* gprof = reinterpret_cast<const uint4*>(a.profile);
    const int rowSlotsGlobal = a.qPad / 8;
    const int Q = a.qLen;
    const bool topGap = a.topGap, leftGap = a.leftGap;
    const int region = kRegions ? a.region : (int)kAllCells;
    const int open = a.gapOpen, ext = a.gapExt;
    // value of a border cell k residues into a border (the other index is -1); the shifted
    // flavour stores X + (i + j) * ext (the other index is -1 here)
    auto border = [&](bool gap, int k) -> int {
        const int v = gap ? borderGap(k, open, ext) : 0;
        return Arith::kDiag ? v + (k - 1) * ext : v;
    };
    // shift a stored value of cell (i, j) back to its true value
    auto unshift = [&](uint32_t v, int i, int j) -> uint32_t {
        if (Arith::kDiag) return ar.toTrue(v, dup16((i + j) * ext));
        return v;
    };

    // per-lane target lengths (NW / OV take answers at each target's own last column)
    const size_t base = (size_t)g * kGroupTargets;
    int lenA = 0, lenB = 0;
    if (LOC || (kRegions && (!TRACK_ALL || region != kAllCells))) {
        lenA = a.lens[base + lane];
        lenB = a.lens[base + kLanes + lane];
    }

    uint32_t best = Arith::lowest(), held = Arith::lowest();  // maximum over all cells (TRACK_ALL)
    uint32_t ans = Arith::answerLowest();                      // answer of the other regions (true values)
    uint32_t H[R], E[R];
    // LOC state, one set per packed half (A = low, B = high): running best score with
    // its column / row, the same for the strip in flight (all-cells region), and the
    // last-column candidate of OV
    int runA = kLowInt, runB = kLowInt, colA = -1, colB = -1, rowA = -1, rowB = -1;
    int scolA = -1, scolB = -1, srowA = -1, srowB = -1;
    int cbA = kLowInt, cbB = kLowInt, crowA = -1, crowB = -1;
    uint32_t bestPrev = Arith::lowest();

    const int lastStrip = nStrips - 1;
    const int rl = Q - 1 - lastStrip * R;  // row of the last query residue inside the last strip
    const int nSteps = nChunks + (W - 1);
    if (unitMode && unitRound > 0) {
        // what this wavefront's strips of the earlier rounds found
        const uint2 p = a.unitPartial[((size_t)groupIndex * W + wave) * kLanes + lane];
        best = p.x;
        ans = p.y;
    }

    for (int rho = unitMode ? unitRound : 0; rho < (unitMode ? unitRound + 1 : nRounds); ++rho) {
        const int s = rho * W + wave;
        const bool active = s < nStrips;
        const bool isLast = s == lastStrip;
        const bool inGlobal = MULTI && active && s > 0 && wave == 0;  // boundary of the previous round, from HBM
        const bool inLds = MULTI && active && s > 0 && wave > 0;      // boundary of the wavefront above, from LDS
        const bool outGlobal = MULTI && active && s + 1 < nStrips && wave == W - 1;
        const bool outLds = MULTI && active && s + 1 < nStrips && wave < W - 1;
        const uint2* bin = nullptr;
        uint2* bout = nullptr;
        if (MULTI && nRounds > 1) {
            bin = a.boundary[(rho + 1) & 1] + a.boundaryOff[g];
            bout = a.boundary[rho & 1] + a.boundaryOff[g];
        }

        uint32_t hdiagTop = Arith::lowest();
        // column-shifted flavour: the zero of the last column done (column -1 at a round's start)
        uint32_t fl = 0u;
        if (active) {
            // stage this strip's slice of the query profile into the wavefront's LDS region
            for (int idx = lane; idx < a.nSymbols * (R / 8); idx += kLanes) {
                const int t = idx / (R / 8), k = idx - t * (R / 8);
                prof[t * SLOTS + k] = gprof[t * rowSlotsGlobal + s * (R / 8) + k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // column -1: left border H[i][-1] and the E it induces in column 0
            const int i0 = s * R;
            if constexpr (Arith::kColShift) {
                // H[r][-1] = 0 on column -1's scale (stored form), E[r][0] = 0 on column 0's
                fl = ar.fromInt(kSwU16Zero) - ar.ext2;
                const uint32_t h0 = ar.stored(fl), e0 = fl + ar.ext2;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    H[r] = h0;
                    E[r] = e0;
                }
                hdiagTop = h0;
            } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t hl;
                if (Arith::kFloor) hl = 0u;
                else if (i0 + r >= Q) hl = Arith::lowest();
                else hl = ar.fromInt(border(leftGap, i0 + r));
                H[r] = hl;
                E[r] = ar.afterOpen(hl);
            }
            // H[i0-1][-1]: diagonal of the strip's first row in column 0
            if (Arith::kFloor) hdiagTop = 0u;
            else hdiagTop = s == 0 ? ar.fromInt(Arith::kDiag ? -2 * ext : 0) : ar.fromInt(border(leftGap, i0 - 1));
            }
        }

        uint2 cur = {0, 0}, nxt = {0, 0};
        uint2 n0 = {0, 0}, n1 = {0, 0}, n2 = {0, 0}, n3 = {0, 0};
        if (active) {
            cur = pack[lane];
            if (inGlobal) {
                n0 = bin[0 * kLanes + lane];
                n1 = bin[1 * kLanes + lane];
                n2 = bin[2 * kLanes + lane];
                n3 = bin[3 * kLanes + lane];
            }
        }

        for (int tau = 0; tau < nSteps; ++tau) {
            const int c = tau - wave;
            if (active && c >= 0 && c < nChunks) {
                uint2 b0 = n0, b1 = n1, b2 = n2, b3 = n3;
                if (W > 1 && inLds) {
                    const uint2* p = ldsBnd[wave - 1][c & 1] + lane;
                    b0 = p[0 * kLanes];
                    b1 = p[1 * kLanes];
                    b2 = p[2 * kLanes];
                    b3 = p[3 * kLanes];
                }
                if (c + 1 < nChunks) {
                    nxt = pack[(size_t)(c + 1) * kLanes + lane];
                    if (MULTI && inGlobal) {
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
                    const int j = c * 4 + cc;
                    const uint32_t tA = ra & 0xffu, tB = rb & 0xffu;
                    ra >>= 8;
                    rb >>= 8;
                    const uint4* pa = prof + tA * SLOTS;
                    const uint4* pb = prof + tB * SLOTS;
                    // row above the strip: H[i0-1][j-1] (diagonal) and the F entering row i0
                    uint32_t diag = hdiagTop;
                    uint32_t f;
                    uint32_t fl1 = 0u, cm = 0u, cheld = 0u;   // column-shifted flavour only
                    if constexpr (Arith::kColShift) {
                        fl += ar.ext2;                    // this column's zero
                        fl1 = fl + ar.ext2;               // the next column's
                        asm volatile("" : "+v"(fl1));     // (a VGPR: max3 with a scalar operand issues slower)
                        cm = cheld = fl;
                    }
                    if (!MULTI || s == 0) {
                        uint32_t topH = Arith::kFloor ? 0u : ar.fromInt(border(topGap, j));  // H[-1][j]
                        if constexpr (Arith::kColShift) topH = ar.stored(fl);                 // 0 on this column's scale
                        hdiagTop = topH;
                        f = ar.afterOpen(topH);
                        if constexpr (Arith::kColShift) f = fl;                               // F entering row 0: the floor
                    } else {
                        hdiagTop = b0.x;
                        f = b0.y;
                    }
                    // Profile rows are fetched one 8-row block ahead of their use; the compiler
                    // barrier at the end of each block keeps hipcc from hoisting every
                    // ds_read_b128 and v_perm to the top of the column (~60 VGPRs, one wave
                    // of occupancy).
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
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int r = r8 * 8 + k;
                            // consume the old H[r] (diagonal of row r+1) before H[r] is rewritten
                            uint32_t dnext = 0;
                            if (r + 1 < R) dnext = ar.addScore(H[r], score(r + 1));
                            const uint32_t h = ar.hmax(dsum, E[r], f);
                            const uint32_t hmo = ar.cellOpen(h);
                            if constexpr (Arith::kColShift) {
                                ar.track(cm, cheld, h, r);                     // this column's maximum
                                E[r] = pk_max3_f16(E[r], hmo, fl1);
                                asm volatile("" : "+v"(E[r]));
                                f = pk_max3_f16(f, hmo, fl1) - ar.ext2;
                                H[r] = ar.stored(h);
                            } else {
                            if (TRACK_ALL) ar.track(best, held, h, r);
                            E[r] = ar.gap(E[r], hmo);
                            // (at its row: E is off the critical path, and hipcc defers the R updates to
                            // the end of the block, each holding on to its hmo)
                            asm volatile("" : "+v"(E[r]));
                            f = ar.gap(f, hmo);
                            H[r] = Arith::kStoresOpen ? hmo : h;
                            }
                            dsum = dnext;
                        }
                        // pins the schedule: the block's cells are finished before the loads
                        // (and v_perm) of the block after next may start
                        asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
                    }
                    if constexpr (Arith::kColShift) {
                        // true values of this column's maximum: an integer max (sticky for inf / NaN patterns)
                        if (R & 1) cm = pk_max3_f16(cm, cheld, cheld);
                        best = ar.max2(best, cm - fl);
                    }
                    if (outGlobal) bout[((size_t)c * 4 + cc) * kLanes + lane] = make_uint2(H[R - 1], f);
                    if (W > 1 && outLds) ldsBnd[wave][c & 1][cc * kLanes + lane] = make_uint2(H[R - 1], f);
                    if (MULTI) {
                        b0 = b1;
                        b1 = b2;
                        b2 = b3;
                    }

                    if constexpr (LOC && TRACK_ALL) {
                        // a half of `best` that changed in this column holds a new strict
                        // maximum: its row is the first row of the column with that value
                        const uint32_t ch = best ^ bestPrev;
                        if (region == kAllCells && __builtin_amdgcn_ballot_w64(ch != 0) != 0) {
                            const uint32_t nbA = best & 0xffffu, nbB = best >> 16;
                            int ra = 0, rb = 0;
#pragma unroll
                            for (int r = R - 1; r >= 0; --r) {
                                if ((H[r] & 0xffffu) == nbA) ra = r;
                                if ((H[r] >> 16) == nbB) rb = r;
                            }
                            if (ch & 0xffffu) {
                                scolA = j;
                                srowA = s * R + ra;
                            }
                            if (ch >> 16) {
                                scolB = j;
                                srowB = s * R + rb;
                            }
                            bestPrev = best;
                        }
                    }
                    if constexpr (LOC && kRegions) {
                        if (region != kAllCells) {
                            const uint32_t lastMask = ((j == lenA - 1) ? 0x0000ffffu : 0u) |
                                                      ((j == lenB - 1) ? 0xffff0000u : 0u);
                            if (isLast) {
                                uint32_t hq = H[0];
#pragma unroll
                                for (int r = 1; r < R; ++r)
                                    if (r == rl) hq = H[r];
                                hq = unshift(hq, Q - 1, j);
                                const int qA = Arith::toInt(hq & 0xffffu), qB = Arith::toInt(hq >> 16);
                                if (region == kLastCell) {
                                    if (j == lenA - 1) { runA = qA; colA = j; }
                                    if (j == lenB - 1) { runB = qB; colB = j; }
                                } else {
                                    // last row: HW scans columns < len, OV columns < len - 1 (its
                                    // last column is scanned row by row below)
                                    const int cut = region == kLastRowCol ? 1 : 0;
                                    if (j < lenA - cut && qA > runA) { runA = qA; colA = j; }
                                    if (j < lenB - cut && qB > runB) { runB = qB; colB = j; }
                                }
                            }
                            if (region == kLastRowCol && __builtin_amdgcn_ballot_w64(lastMask != 0) != 0) {
                                int mA = INT32_MIN, mB = INT32_MIN, ra = 0, rb = 0;
                                uint32_t off = Arith::kDiag ? dup16((s * R + R - 1 + j) * ext) : 0u;
                                const uint32_t step = Arith::kDiag ? dup16(ext) : 0u;
#pragma unroll
                                for (int r = R - 1; r >= 0; --r) {
                                    const uint32_t offR = off;
                                    if (Arith::kDiag) {
                                        off -= step;  // running shift of the row above
                                        asm volatile("" : "+v"(off));
                                    }
                                    if (s * R + r < Q) {
                                        const uint32_t hv = Arith::kDiag ? ar.toTrue(H[r], offR) : H[r];
                                        const int hA = Arith::toInt(hv & 0xffffu), hB = Arith::toInt(hv >> 16);
                                        if (hA >= mA) { mA = hA; ra = r; }
                                        if (hB >= mB) { mB = hB; rb = r; }
                                    }
                                }
                                if ((lastMask & 0xffffu) && mA > cbA) { cbA = mA; crowA = s * R + ra; }
                                if ((lastMask >> 16) && mB > cbB) { cbB = mB; crowB = s * R + rb; }
                            }
                        }
                    }
                    // ---- answers on the last query row / each target's last column ----
                    if (!LOC && kRegions && (!TRACK_ALL || region != kAllCells)) {
                        const uint32_t lastMask = ((j == lenA - 1) ? 0x0000ffffu : 0u) |
                                                  ((j == lenB - 1) ? 0xffff0000u : 0u);
                        if (isLast) {
                            uint32_t hq = H[0];
#pragma unroll
                            for (int r = 1; r < R; ++r)
                                if (r == rl) hq = H[r];
                            hq = unshift(hq, Q - 1, j);
                            if (region == kLastCell) {
                                ans = selectHalves(ans, hq, lastMask);
                            } else if (Arith::kWeakPad) {
                                // last row, columns of the target only
                                const uint32_t inside = ((j < lenA) ? 0x0000ffffu : 0u) | ((j < lenB) ? 0xffff0000u : 0u);
                                ans = selectHalves(ans, ar.max2(ans, hq), inside);
                            } else {
                                ans = ar.max2(ans, hq);  // last row; padded columns never exceed real ones
                            }
                        }
                        if (region == kLastRowCol && __builtin_amdgcn_ballot_w64(lastMask != 0) != 0) {
                            // some lane is on its target's last column: maximum over this strip's rows
                            // (the shift of row r is kept as a running packed value: one add per row
                            // here instead of R live offsets)
                            uint32_t off = Arith::kDiag ? dup16((s * R + j) * ext) : 0u;
                            const uint32_t step = Arith::kDiag ? dup16(ext) : 0u;
                            uint32_t cm = Arith::kDiag ? ar.toTrue(H[0], off) : H[0];
#pragma unroll
                            for (int r = 1; r < R; ++r) {
                                if (Arith::kDiag) {
                                    off += step;  // halves stay below 32000: no carry between them
                                    asm volatile("" : "+v"(off));
                                    // (rows beyond the query only matter when padding can score: kWeakPad)
                                    if (!Arith::kWeakPad || s * R + r < Q) cm = ar.max2(cm, ar.toTrue(H[r], off));
                                } else {
                                    cm = ar.max2(cm, H[r]);
                                }
                            }
                            ans = selectHalves(ans, ar.max2(ans, cm), lastMask);
                        }
                    }
                }
                cur = nxt;
            }
            if (W > 1) __syncthreads();
        }
        if constexpr (LOC && TRACK_ALL) {
            if (region == kAllCells && active) {
                // fold this strip's first maximum into the running one: higher score wins,
                // then the smaller column (rows of later strips are larger)
                const int sA = Arith::toInt(best & 0xffffu), sB = Arith::toInt(best >> 16);
                if (scolA >= 0 && (sA > runA || (sA == runA && scolA < colA))) { runA = sA; colA = scolA; rowA = srowA; }
                if (scolB >= 0 && (sB > runB || (sB == runB && scolB < colB))) { runB = sB; colB = scolB; rowB = srowB; }
                best = Arith::lowest();
                held = Arith::lowest();
                bestPrev = Arith::lowest();
                scolA = scolB = srowA = srowB = -1;
            }
        }
        if (MULTI && nRounds > 1) {
            // the next round's first strip reads what this round's last strip wrote to HBM
            __threadfence();
            if (W > 1) __syncthreads();
        }
    }

    // ---- combine the wavefronts' partial answers ---------------------------------
    if constexpr (LOC) {
        if (W > 1) {
            ldsLoc[wave][0][lane] = runA;
            ldsLoc[wave][1][lane] = runB;
            ldsLoc[wave][2][lane] = colA;
            ldsLoc[wave][3][lane] = colB;
            ldsLoc[wave][4][lane] = rowA;
            ldsLoc[wave][5][lane] = rowB;
            ldsLoc[wave][6][lane] = cbA;
            ldsLoc[wave][7][lane] = cbB;
            ldsLoc[wave][8][lane] = crowA;
            ldsLoc[wave][9][lane] = crowB;
            __syncthreads();
            if (wave != 0) return;
            for (int w = 1; w < W; ++w) {
                const int oA = ldsLoc[w][0][lane], oB = ldsLoc[w][1][lane];
                const int ocA = ldsLoc[w][2][lane], ocB = ldsLoc[w][3][lane];
                const int orA = ldsLoc[w][4][lane], orB = ldsLoc[w][5][lane];
                // same order as the scan: score, then column, then row
                if (ocA >= 0 && (colA < 0 || oA > runA || (oA == runA && (ocA < colA || (ocA == colA && orA < rowA))))) {
                    runA = oA; colA = ocA; rowA = orA;
                }
                if (ocB >= 0 && (colB < 0 || oB > runB || (oB == runB && (ocB < colB || (ocB == colB && orB < rowB))))) {
                    runB = oB; colB = ocB; rowB = orB;
                }
                const int pA = ldsLoc[w][6][lane], pB = ldsLoc[w][7][lane];
                const int prA = ldsLoc[w][8][lane], prB = ldsLoc[w][9][lane];
                if (prA >= 0 && (crowA < 0 || pA > cbA || (pA == cbA && prA < crowA))) { cbA = pA; crowA = prA; }
                if (prB >= 0 && (crowB < 0 || pB > cbB || (pB == cbB && prB < crowB))) { cbB = pB; crowB = prB; }
            }
        }
        if (region != kAllCells) {
            // answers of the last-row regions sit on query row Q - 1
            rowA = colA >= 0 ? Q - 1 : -1;
            rowB = colB >= 0 ? Q - 1 : -1;
        }
        if (region == kLastRowCol) {
            // OV: the last column is scanned after the last row, strictly greater wins
            if (crowA >= 0 && (colA < 0 || cbA > runA)) { runA = cbA; rowA = crowA; colA = lenA - 1; }
            if (crowB >= 0 && (colB < 0 || cbB > runB)) { runB = cbB; rowB = crowB; colB = lenB - 1; }
        }
        a.score[base + lane] = runA;
        a.score[base + kLanes + lane] = runB;
        a.endI[base + lane] = rowA;
        a.endI[base + kLanes + lane] = rowB;
        a.endJ[base + lane] = colA;
        a.endJ[base + kLanes + lane] = colB;
        if (a.overflow) {
            a.overflow[base + lane] = runA >= Arith::kLimit;
            a.overflow[base + kLanes + lane] = runB >= Arith::kLimit;
        }
        return;
    }
    if (unitMode && unitRound + 1 < nRounds) {
        // not the group's last round: leave the partial answers and the round's boundary row (HBM,
        // fenced above) to whoever takes the next round of this group
        a.unitPartial[((size_t)groupIndex * W + wave) * kLanes + lane] = make_uint2(best, ans);
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(a.unitFlags + groupIndex, unitRound + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        goto next_unit;
    }
    {
        uint32_t res = (TRACK_ALL && region == kAllCells) ? best : ans;
        if (W > 1) {
            ldsOut[wave][lane] = res;
            __syncthreads();
            if (wave == 0)
                for (int w = 1; w < W; ++w) res = ar.max2(res, ldsOut[w][lane]);
        }
        if (wave == 0) {
            const int lo = Arith::toInt(res & 0xffffu), hi = Arith::toInt(res >> 16);
            a.score[base + lane] = lo;
            a.score[base + kLanes + lane] = hi;
            if (a.overflow) {
                const int limit = Arith::kColShift ? a.biasedLimit : Arith::kLimit;
                a.overflow[base + lane] = lo >= limit;
                a.overflow[base + kLanes + lane] = hi >= limit;
            }
        }
    }
    if (unitMode) goto next_unit;   // (the barriers of the hand-out keep ldsOut safe)
}

template <int R, typename Arith, bool TRACK_ALL, bool LOC>
static hipError_t launchW(const InterseqArgs& a, int waves, hipStream_t stream) {
    const dim3 grid(a.nGroups);
    if (a.nStrips == 1) {
        hipLaunchKernelGGL((interseq_kernel<R, Arith, 1, TRACK_ALL, false, LOC>), grid, dim3(kLanes), 0, stream, a);
        return hipGetLastError();
    }
    if constexpr (!LOC) {
        if (a.unitCounter != nullptr) {
            switch (waves) {
                case 1: hipLaunchKernelGGL((interseq_kernel<R, Arith, 1, TRACK_ALL, true, false, true>), grid, dim3(1 * kLanes), 0, stream, a); break;
                case 2: hipLaunchKernelGGL((interseq_kernel<R, Arith, 2, TRACK_ALL, true, false, true>), grid, dim3(2 * kLanes), 0, stream, a); break;
                case 4: hipLaunchKernelGGL((interseq_kernel<R, Arith, 4, TRACK_ALL, true, false, true>), grid, dim3(4 * kLanes), 0, stream, a); break;
                case 8: hipLaunchKernelGGL((interseq_kernel<R, Arith, 8, TRACK_ALL, true, false, true>), grid, dim3(8 * kLanes), 0, stream, a); break;
                default: return hipErrorInvalidValue;
            }
            return hipGetLastError();
        }
    }
    switch (waves) {
        case 1: hipLaunchKernelGGL((interseq_kernel<R, Arith, 1, TRACK_ALL, true, LOC>), grid, dim3(1 * kLanes), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((interseq_kernel<R, Arith, 2, TRACK_ALL, true, LOC>), grid, dim3(2 * kLanes), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((interseq_kernel<R, Arith, 4, TRACK_ALL, true, LOC>), grid, dim3(4 * kLanes), 0, stream, a); break;
        case 8: hipLaunchKernelGGL((interseq_kernel<R, Arith, 8, TRACK_ALL, true, LOC>), grid, dim3(8 * kLanes), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Rows per strip are a multiple of 8 (one ds_read_b128 = 8 16-bit scores);
// `waves` (1, 2, 4 or 8) is the number of strips of a group in flight.
template <typename Arith, bool TRACK_ALL, bool LOC>
static hipError_t launchFlavour(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    switch (rowsPerStrip) {
        case 8: return launchW<8, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 16: return launchW<16, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 24: return launchW<24, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 32: return launchW<32, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 40: return launchW<40, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 48: return launchW<48, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 56: return launchW<56, Arith, TRACK_ALL, LOC>(a, waves, stream);
        case 64: return launchW<64, Arith, TRACK_ALL, LOC>(a, waves, stream);
    }
    return hipErrorInvalidValue;
}


// ---- single-strip Smith-Waterman with a pair-indexed profile --------------------
// The {A's score, B's score} pair of a query row depends on the query row and on the
// two residues (tA, tB) only, so for a query that fits one strip the whole table
//     pairs[tA * nSymbols + tB][r] = { S[q_r][tA], S[q_r][tB] }
// fits the CU's 160 KB of LDS (625 rows x 60 dwords for the 24-letter protein
// alphabet and R = 56). One ds_read_b128 then delivers four ready-made operands and
// the v_perm_b32 per cell pair disappears (about 11 % of the VALU work). The table
// is built once per workgroup, so workgroups are persistent: kPairWaves wavefronts
// each pull groups from a shared counter until none is left.
constexpr int kPairWaves = 12;  // 3 per SIMD at <= 168 VGPRs

template <int R>
struct PairLayout {
    static constexpr int kRowSlots = ((R + 3) / 4) | 1;  // 16-byte slots per row, odd: rows start on different slots
    static __host__ __device__ constexpr size_t bytes(int nSymbols) {
        return (size_t)nSymbols * nSymbols * kRowSlots * 16;
    }
};

template <int R, typename Arith>
__global__ __launch_bounds__(kPairWaves * kLanes) void interseq_pair_kernel(InterseqArgs a) {
    constexpr int SLOTS = PairLayout<R>::kRowSlots;
    constexpr int NB4 = R / 4;
    extern __shared__ uint4 pairs[];

    const int lane = threadIdx.x & 63;
    const int nSym = a.nSymbols;
    const Arith ar(a.gapOpen, a.gapExt);

    // build the table: one thread per (pair row, 2 query rows)
    {
        const uint32_t* gw = reinterpret_cast<const uint32_t*>(a.profile);
        uint32_t* pw = reinterpret_cast<uint32_t*>(pairs);
        const int rowWords = a.qPad / 2;
        const int total = nSym * nSym * (R / 2);
        for (int idx = threadIdx.x; idx < total; idx += kPairWaves * kLanes) {
            const int row = idx / (R / 2), k = idx - row * (R / 2);
            const int tA = row / nSym, tB = row - tA * nSym;
            const uint32_t wa = gw[tA * rowWords + k], wb = gw[tB * rowWords + k];
            pw[row * (SLOTS * 4) + 2 * k] = __builtin_amdgcn_perm(wb, wa, 0x05040100u);
            pw[row * (SLOTS * 4) + 2 * k + 1] = __builtin_amdgcn_perm(wb, wa, 0x07060302u);
        }
    }
    __syncthreads();

    // Hand-out of groups (sorted longest first). First round: static and striped, so that
    // every CU, and every SIMD inside it (wavefronts w, w+4, w+8 share one), starts with
    // a mix of long and short groups: wavefront w takes tier kTier[w % 4][w / 4] of
    // gridDim.x groups each, in snake order across workgroups. Later rounds: whoever
    // finishes pulls the next group from a shared counter, so wavefronts finish together.
    const int wave = threadIdx.x >> 6;
    constexpr int kTier[4][3] = {{0, 6, 11}, {1, 7, 8}, {2, 5, 9}, {3, 4, 10}};
    const int tier = kTier[wave & 3][wave >> 2];
    const int firstDynamic = kPairWaves * gridDim.x;
    bool firstRound = true;
    for (;;) {
        int g;
        if (firstRound) {
            g = tier * gridDim.x + ((tier & 1) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x);
            firstRound = false;
            if (g >= a.nGroups) continue;
            g += a.groupBase;
        } else {
            g = 0;
            if (lane == 0) g = atomicAdd(a.workCounter, 1);
            g = __builtin_amdgcn_readfirstlane(g) + firstDynamic;
            if (g >= a.nGroups) break;
            g += a.groupBase;
        }
        const uint2* pack = a.pack + a.groupOff[g];
        const int nChunks = a.groupChunks[g];
        // a group much longer than a wavefront's balanced share is on the critical path:
        // let it win the SIMD's issue arbitration against its co-resident wavefronts
        if (nChunks > a.priorityChunks) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        uint32_t best = 0u, held = 0u;
        uint32_t H[R], E[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            H[r] = 0u;
            E[r] = 0u;
        }
        uint2 cur = pack[lane];
        for (int c = 0; c < nChunks; ++c) {
            uint2 nxt = {0, 0};
            if (c + 1 < nChunks) nxt = pack[(size_t)(c + 1) * kLanes + lane];
            uint32_t ra = cur.x, rb = cur.y;
#pragma unroll 1
            for (int cc = 0; cc < 4; ++cc) {
                const uint32_t tA = ra & 0xffu, tB = rb & 0xffu;
                ra >>= 8;
                rb >>= 8;
                const uint4* prow = pairs + (tA * nSym + tB) * SLOTS;
                uint32_t f = 0u;
                uint4 v[NB4];
                v[0] = prow[0];
                auto score = [&](int r) -> uint32_t {
                    const uint4 x = v[r >> 2];
                    const int k = r & 3;
                    return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                };
                uint32_t dsum = ar.addScore(0u, score(0));
#pragma unroll
                for (int r4 = 0; r4 < NB4; ++r4) {
                    if (r4 + 1 < NB4) v[r4 + 1] = prow[r4 + 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = r4 * 4 + k;
                        uint32_t dnext = 0;
                        if (r + 1 < R) dnext = ar.addScore(H[r], score(r + 1));
                        const uint32_t h = ar.hmax(dsum, E[r], f);
                        ar.track(best, held, h, r);
                        const uint32_t hmo = ar.cellOpen(h);
                        E[r] = ar.gap(E[r], hmo);
                        f = ar.gap(f, hmo);
                        H[r] = h;
                        dsum = dnext;
                    }
                    if (r4 & 1) asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
                }
            }
            cur = nxt;
        }
        const int lo = Arith::toInt(best & 0xffffu), hi = Arith::toInt(best >> 16);
        const size_t base = (size_t)g * kGroupTargets;
        a.score[base + lane] = lo;
        a.score[base + kLanes + lane] = hi;
        if (a.overflow) {
            a.overflow[base + lane] = lo >= Arith::kLimit;
            a.overflow[base + kLanes + lane] = hi >= Arith::kLimit;
        }
    }
}

// per-device: hipFuncSetAttribute applies to the current device only
static inline bool firstUseOnThisDevice(uint64_t* seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const uint64_t bit = 1ull << dev;
    const uint64_t old = __atomic_fetch_or(seen, bit, __ATOMIC_RELAXED);
    return !(old & bit);
}

template <int R, typename Arith>
static hipError_t launchPairR(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    const size_t lds = PairLayout<R>::bytes(a.nSymbols);
    static uint64_t configured = 0;  // one bit per device: the attribute belongs to the device
    if (firstUseOnThisDevice(&configured)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&interseq_pair_kernel<R, Arith>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    // (every CU takes part even when the groups are fewer than its wavefronts: the first round hands
    // out group g to wavefront tier g / blocks of workgroup g % blocks, and a wavefront alone on its
    // SIMD sweeps its group three times faster than one of three)
    const int blocks = std::max(1, std::min(computeUnits, a.nGroups));
    hipLaunchKernelGGL((interseq_pair_kernel<R, Arith>), dim3(blocks), dim3(kPairWaves * kLanes), lds, stream, a);
    return hipGetLastError();
}

// Smith-Waterman, one strip, pair-indexed profile. Returns hipErrorInvalidValue when
// the table does not fit LDS (the caller then uses the v_perm variant).
template <typename Arith>
static hipError_t launchPairFlavour(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    if (a.nStrips != 1) return hipErrorInvalidValue;
    switch (rowsPerStrip) {
        case 8: return launchPairR<8, Arith>(a, computeUnits, stream);
        case 16: return launchPairR<16, Arith>(a, computeUnits, stream);
        case 24: return launchPairR<24, Arith>(a, computeUnits, stream);
        case 32: return launchPairR<32, Arith>(a, computeUnits, stream);
        case 40: return launchPairR<40, Arith>(a, computeUnits, stream);
        case 48: return launchPairR<48, Arith>(a, computeUnits, stream);
        case 56: return launchPairR<56, Arith>(a, computeUnits, stream);
        case 64: return launchPairR<64, Arith>(a, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

// ---- single-strip Smith-Waterman on biased integer halves, column-shifted ---------
// Third arithmetic for the pair-table kernel. A half holds the unsigned integer
//     kBiasedZero + x + shift(j),      shift(j) = (j - j0) * ext,
// for a true value x of column j, and is COMPARED as a half float: between 0x0400 (smallest
// normal) and 0x7BFF (largest finite) the order of the bit patterns is the order of the
// numbers, so v_pk_maximum3_f16 still folds two max per cell while every addition is a plain
// 32-bit integer add over both halves at once: the pair table holds
// (sB' << 16) + sA' as ONE signed integer, so the borrow of a negative low score is undone by
// the carry of the sum as long as each half stays in [0, 65535]. The column shift makes
// extending E free (the next column's shift absorbs the ext) and lets E and F open with the same
// h - (open - ext); the Smith-Waterman floor becomes the column's own zero `fl`:
//     h    = max3(Hdiag + s', E, F)                s' = s + ext (one column to the right)
//     hmo  = h - (open - ext)
//     E    = max3(E, hmo, fl + ext)                (already on the next column's scale)
//     F    = max3(F, hmo, fl + ext) - ext
// 3 integer adds + 3.5 max3 per cell pair instead of 4 packed half adds + 3.5 max3, and the exact
// range grows from 2048 to kBiasedLimit. Measured mix: tools/ubench_mix.hip. Every kBiasedMaxShift
// of accumulated shift the state is rebased (112 subtractions). A half that reaches 0x7C00
// (inf / NaN patterns) makes the column maximum inf / NaN (maximum3 propagates NaN, payloads stay
// below 0x8000), the running best is an INTEGER max of (column maximum - fl) and therefore ends
// at or above kBiasedLimit: the lane is flagged and redone by the next rung.
constexpr int kBiasedZero = 0x0800;      // pattern of a true 0 at shift 0; 0x0400 of room below
constexpr int kBiasedGuard = 0x0400;     // how far a value may dip below the column's zero
constexpr int kBiasedMaxShift = 4096;
constexpr int kBiasedLimit = 0x7C00 - kBiasedZero - kBiasedMaxShift;  // 25600
constexpr int kBiasedPadScore = -kBiasedGuard;  // padding symbol / padding rows (true value)
static_assert(kBiasedLimit == kBiasedScoreLimit && kBiasedPadScore == kBiasedPad, "common.h mirrors these");
static_assert(kBiasedMaxMagnitude <= kBiasedGuard && 5 * kBiasedMaxExt <= kBiasedMaxShift, "guard band");

// End locations (LOC): every value is scaled by 2^kBits (kBits = 4, 5 or 6: enough for the strip's
// rows), so the low bits of a cell are free, and the key  h + (2^kBits - 1 - row)  is folded instead
// of h: the column maximum then carries the FIRST row that holds it, for one more integer add per
// cell pair, and the running best (an integer max against best | rowmask, i.e. strictly greater
// scores only) carries the row of the first maximum in column-major order. No row scan. The exact
// range shrinks to kLocLimit(kBits) - 384 for 64 rows, above what a 64-residue query can score with
// the usual protein matrices short of near-identity; flagged lanes take the next rung as always.
constexpr int kLocZero = 0x0C00;         // 0x0800 of room below: scores down to -32 at 6 bits
constexpr int kLocGuard = 0x0800;
static_assert(kLocGuard == kLocGuardBand && kBiasedMaxShift == kLocMaxShift && kLocZero == kLocZeroPattern, "common.h mirrors these");
__host__ __device__ constexpr int locRowBits(int rows) { return rows <= 16 ? 4 : rows <= 32 ? 5 : 6; }
__host__ __device__ constexpr int locLimit(int bits) { return (0x7C00 - kLocZero - kBiasedMaxShift) >> bits; }

static __device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// the same signed value in both halves, as one integer (see above)
// (unsigned arithmetic: the product wraps modulo 2^32 for negative or large v, no signed overflow)
static __device__ __forceinline__ uint32_t both(int v) { return (uint32_t)v * 0x00010001u; }

// Pacing of the wavefronts that share a SIMD (round 3). The twelve wavefronts of a unit sweep groups of
// (nearly) equal length and meet at the unit's barrier, but a SIMD serves its OLDEST ready wavefront
// first: of three with equal work the first is done after little more than half the unit's time, the
// last sweeps the rest of its group alone on a SIMD it cannot fill, and meanwhile the others wait at the
// barrier (PMC of the round-2 kernel on BASELINE configs[3]: wavefronts parked 47 % of their life,
// 15 % in the one-strip kernel whose wavefronts fetch new groups on their own). Each wavefront
// therefore publishes the chunk it is on in LDS, per hardware SIMD, and the one furthest behind runs at
// a higher priority: the three advance together and reach the barrier together.
#ifndef MIOPAL_STRIP_PACE
#define MIOPAL_STRIP_PACE 1
#endif

// Diagnostic builds of the strips kernels (tools/ab_build.sh; never the product build):
//   -DMIOPAL_STRIP_TIMING=1   every wavefront adds up s_memtime deltas around its wait sites and leaves them in
//                             InterseqArgs::stripTiming: [0] taking a unit (counter, the two barriers, the table
//                             when the strip changes), [1] polling the strip above, [2] the s_waitcnt before
//                             progress is published, [3] the sweeps (polls and publishes included), [4] the
//                             wavefront's life, [5] wavefronts (s_memtime comes back through lgkmcnt, so the LDS
//                             reads in flight are waited for at every mark: the build is slower than the product's)
//   -DMIOPAL_ABL_NO_POLL / _NO_ROW_LOADS / _NO_ROW_STORES / _NO_PUBLISH_WAIT
//                             ablations: the same instruction stream without one of the hand-over's parts (wrong
//                             results, honest timing: what that part costs)
#ifndef MIOPAL_STRIP_TIMING
#define MIOPAL_STRIP_TIMING 0
#endif
// Boundary rows between strips travel 16 bytes a lane (round 4): {H, F} of TWO neighbouring columns per store /
// load, [column pair][lane] - the same bytes as before, half the memory instructions. An 8-byte sc1 store is one
// fabric write per lane at 2.7 x the time per byte of a 16-byte one, an 8-byte sc1 load runs at 0.54-0.70 x the
// 16-byte rate (MI355X_MICROARCH.md, "stores of each flavour"); the ablation builds below showed each direction of
// the hand-over costing a cfg4 search 9 % (profiles/r04_strips_ablation.txt). buffer_load / buffer_store
// dwordx4 with sc1 (the atomics of common.h stop at 8 bytes); the compiler tracks them in vmcnt like any load.
typedef unsigned int StripU4 __attribute__((ext_vector_type(4)));
struct StripRows {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ explicit StripRows(void* base)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000)) {}
    __device__ __forceinline__ StripU4 load(int pair, int lane) const {
#ifdef MIOPAL_ABL_NO_ROW_LOADS
        return StripU4{0x0a000a00u + (uint32_t)pair, 0x08000800u, 0x0a000a00u, 0x08000800u};
#else
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (pair * kLanes + lane) * 16, 0, 16 /* sc1 */);
#endif
    }
    __device__ __forceinline__ void store(int pair, int lane, StripU4 v) const {
#ifndef MIOPAL_ABL_NO_ROW_STORES
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (pair * kLanes + lane) * 16, 0, 16 /* sc1 */);
#else
        asm volatile("" ::"v"(v));
#endif
    }
};
struct StripTimer {
#if MIOPAL_STRIP_TIMING
    unsigned long long t[4] = {0, 0, 0, 0}, mark[4] = {0, 0, 0, 0}, born = __builtin_amdgcn_s_memtime();
    __device__ __forceinline__ void start(int k) { mark[k] = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void stop(int k) { t[k] += __builtin_amdgcn_s_memtime() - mark[k]; }
    __device__ __forceinline__ void flush(unsigned long long* out, int lane) {
        if (!out || lane != 0) return;
        for (int k = 0; k < 4; ++k) atomicAdd(out + k, t[k]);
        atomicAdd(out + 4, __builtin_amdgcn_s_memtime() - born);
        atomicAdd(out + 5, 1ull);
    }
#else
    __device__ __forceinline__ void start(int) {}
    __device__ __forceinline__ void stop(int) {}
    __device__ __forceinline__ void flush(unsigned long long*, int) {}
#endif
};
constexpr int kPaceInts = 20;   // 4 SIMDs x 4 slots of progress + 4 slot counters
struct SimdPace {
    int* mine = nullptr;        // this wavefront's progress word
    const int4* simdRow = nullptr;
    // (once per kernel; `lds` = kPaceInts ints, every thread of the workgroup calls this)
    __device__ __forceinline__ void init(int* lds, int lane) {
        if (threadIdx.x < kPaceInts) lds[threadIdx.x] = threadIdx.x < 16 ? INT32_MAX : 0;
        __syncthreads();
        // HW_REG_HW_ID (4), SIMD_ID = bits 5:4
        const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | ((2 - 1) << 11)) & 3;
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&lds[16 + simd], 1);
        slot = __builtin_amdgcn_readfirstlane(slot) & 3;
        mine = lds + simd * 4 + slot;
        simdRow = reinterpret_cast<const int4*>(lds + simd * 4);
    }
    __device__ __forceinline__ void begin(int lane) const {
        if (MIOPAL_STRIP_PACE && lane == 0) *mine = 0;
    }
    __device__ __forceinline__ void end(int lane) const {
        if (MIOPAL_STRIP_PACE && lane == 0) *mine = INT32_MAX;   // (at the barrier: nobody waits for this one)
    }
    // at the top of chunk c; `fixed`: the wavefront keeps the priority it has (a long group)
    __device__ __forceinline__ void step(int c, int lane, bool fixed) const {
        if (!MIOPAL_STRIP_PACE || fixed) return;
        if (lane == 0) *mine = c;
        const int4 p = *simdRow;
        const int behind = __builtin_amdgcn_readfirstlane(min(min(p.x, p.y), min(p.z, p.w)));
        if (c <= behind) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
    }
};

// The same pacing in the one-strip kernels, by what is left of each wavefront's current group: the last
// groups of a SIMD end together instead of one after the other (round 3; 393k / 500k targets x 300 at
// Q = 53: 0.634 -> 0.574 / 0.816 -> 0.742 ms, nothing lost at other sizes: profiles/r03_headline_pace_ab.txt)
#ifndef MIOPAL_HEADLINE_PACE
#define MIOPAL_HEADLINE_PACE 1
#endif

template <int R, bool LOC>
__global__ __launch_bounds__(kPairWaves * kLanes) void interseq_pair_biased_kernel(InterseqArgs a) {
    constexpr int SLOTS = PairLayout<R>::kRowSlots;
    constexpr int NB4 = (R + 3) / 4;   // R need not be a multiple of 4: the last read is partly used
    constexpr int kBits = LOC ? locRowBits(R) : 0;
    constexpr int kRowMask = (1 << kBits) - 1;
    extern __shared__ uint4 pairs[];

    const int lane = threadIdx.x & 63;
    const int nSym = a.nSymbols;
    const int ext = a.gapExt << kBits;   // pattern units
    const uint32_t ext2 = both(ext), openMinusExt2 = both((a.gapOpen << kBits) - ext);
    const uint32_t zero2 = both(LOC ? kLocZero : kBiasedZero);

    // build the table: one thread per (pair row, query row); profile = true scores as int16
    {
        const int16_t* gp = a.profile;
        uint32_t* pw = reinterpret_cast<uint32_t*>(pairs);
        const int total = nSym * nSym * R;
        for (int idx = threadIdx.x; idx < total; idx += kPairWaves * kLanes) {
            const int row = idx / R, r = idx - row * R;
            const int tA = row / nSym, tB = row - tA * nSym;
            // (padding symbol / padding rows: as far below the column's zero as the guard band allows)
            constexpr int kPadPattern = LOC ? -kLocGuard : kBiasedPadScore;
            const int vA = gp[tA * a.qPad + r], vB = gp[tB * a.qPad + r];
            const int sA = (vA == kBiasedPadScore ? kPadPattern : vA << kBits) + ext;
            const int sB = (vB == kBiasedPadScore ? kPadPattern : vB << kBits) + ext;
            pw[row * (SLOTS * 4) + r] = (uint32_t)(sB * 65536 + sA);
        }
    }
    // groups handed to the wavefronts of each SIMD (end game below): 16 bytes behind the table
    int* simdTaken = reinterpret_cast<int*>(pairs + nSym * nSym * SLOTS);
    if (threadIdx.x < 4) simdTaken[threadIdx.x] = 0;
    __syncthreads();
    // (the wavefronts of a SIMD paced by what is left of their groups: MIOPAL_HEADLINE_PACE above)
    SimdPace pace;
    if (MIOPAL_HEADLINE_PACE) pace.init(simdTaken + 4, lane);

    // (wave-uniform by construction; said so, or everything derived from it would sit in VGPRs)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int kTier[4][3] = {{0, 6, 11}, {1, 7, 8}, {2, 5, 9}, {3, 4, 10}};
    const int tier = kTier[wave & 3][wave >> 2];
    const int firstDynamic = kPairWaves * gridDim.x;
    // HW_REG_HW_ID (4), SIMD_ID = bits 5:4
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | ((2 - 1) << 11)) & 3;
    bool firstRound = true;
    for (;;) {
        int g;
        if (firstRound) {
            g = tier * gridDim.x + ((tier & 1) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x);
            firstRound = false;
            if (lane == 0 && a.tailThrottle > 0) atomicAdd(&simdTaken[simd], 1);
            if (g >= a.nGroups) continue;
            g += a.groupBase;
        } else {
            g = 0;
            if (lane == 0) {
                // Groups of similar length (a.tailThrottle = groups per SIMD, rounded up; else 0): a
                // SIMD that takes one group more than its share works a whole extra round while its
                // neighbours idle, so every SIMD stops at its share - counted in LDS per hardware
                // SIMD - and the last round runs two wavefronts (or one) per SIMD, each faster for it.
                // Shares add up to at least the number of groups, and a SIMD below its share keeps
                // asking, so every group is taken.
                bool take = true;
                if (a.tailThrottle > 0) take = atomicAdd(&simdTaken[simd], 1) < a.tailThrottle;
                g = take ? atomicAdd(a.workCounter, 1) : INT32_MAX - firstDynamic;
            }
            g = __builtin_amdgcn_readfirstlane(g) + firstDynamic;
            if (g >= a.nGroups) break;
            g += a.groupBase;
        }
        const uint2* pack = a.pack + a.groupOff[g];
        const int nChunks = a.groupChunks[g];
        if (nChunks > a.priorityChunks) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        uint32_t best = 0u;                 // true values (LOC: keys), integer order
        int colA = -1, colB = -1;           // LOC: column of the first maximum of each half
        uint32_t fl = zero2 - ext2;         // zero of column -1
        int shift = -ext;                   // fl = zero2 + both(shift); wave-uniform
        uint32_t H[R], E[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            H[r] = fl;                      // H[r][-1] = 0 on column -1's scale
            E[r] = zero2;                   // E[r][0]  = 0 on column 0's scale
        }
        uint2 cur = pack[lane];
        // LDS row of a residue pair (24-bit multiplies: v_mul_lo_u32 is a quarter-rate instruction)
        auto rowOf = [&](uint32_t tA, uint32_t tB) -> const uint4* {
            const uint32_t rowIdx = __umul24(tA, (uint32_t)nSym) + tB;
            return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pairs) +
                                                  __umul24(rowIdx, (uint32_t)(SLOTS * 16)));
        };
        // The first two 4-row blocks of a column are fetched while the column before it is still
        // being computed, the others two blocks ahead of their use: an LDS read waits behind the
        // other eleven wavefronts' (bank-conflicted) reads, and a column that starts by waiting
        // for its first scores leaves the SIMD to two wavefronts.
#ifndef MIOPAL_PAIR_AHEAD
#define MIOPAL_PAIR_AHEAD 3
#endif
#ifndef MIOPAL_PAIR_XCOL
#define MIOPAL_PAIR_XCOL 1
#endif
        // (three blocks ahead is the best of 1..3 on cfg2; 64 rows leave registers for two)
        constexpr int kWant = LOC ? (R > 62 ? 1 : R > 58 ? 2 : MIOPAL_PAIR_AHEAD) : (R > 60 ? 2 : MIOPAL_PAIR_AHEAD);
        constexpr int kAhead = NB4 > kWant ? kWant : 1;
        constexpr bool kAcross = MIOPAL_PAIR_XCOL != 0;
        const uint4* prowNext = rowOf(cur.x & 0xffu, cur.y & 0xffu);
        uint4 vn[kAhead];
        if (kAcross) {
#pragma unroll
            for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
        }
        for (int c = 0; c < nChunks; ++c) {
            uint2 nxt = {0, 0};
            if (c + 1 < nChunks) nxt = pack[(size_t)(c + 1) * kLanes + lane];
            if (MIOPAL_HEADLINE_PACE) pace.step(c - nChunks, lane, nChunks > a.priorityChunks);
            uint32_t ra = cur.x, rb = cur.y;
#pragma unroll 1
            for (int cc = 0; cc < 4; ++cc) {
                const uint4* prow = prowNext;
                uint4 v[NB4];
#pragma unroll
                for (int k = 0; k < kAhead; ++k) v[k] = kAcross ? vn[k] : prow[k];
                // the column after this one (past the group's end: row 0, never used)
                ra = cc < 3 ? ra >> 8 : nxt.x;
                rb = cc < 3 ? rb >> 8 : nxt.y;
                prowNext = rowOf(ra & 0xffu, rb & 0xffu);
                auto score = [&](int r) -> uint32_t {
                    const uint4 x = v[r >> 2];
                    const int k = r & 3;
                    return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                };
                uint32_t dsum = fl + score(0);      // H[-1][j-1] = 0 on the previous column's scale
                fl += ext2;                         // this column's zero
                uint32_t fl1 = fl + ext2;           // the next column's
                // (wave-uniform, so hipcc would keep it in an SGPR: v_pk_maximum3_f16 with a scalar
                // operand issues ~10 % slower, profiles/r01_valu_issue_rates.txt)
                asm volatile("" : "+v"(fl1));
                uint32_t f = fl, cm = fl, held = fl;
#pragma unroll
                for (int r4 = 0; r4 < NB4; ++r4) {
                    if (r4 + kAhead < NB4) v[r4 + kAhead] = prow[r4 + kAhead];
                    if (kAcross && r4 == (NB4 > 3 ? NB4 - 3 : 0)) {
#pragma unroll
                        for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = r4 * 4 + k;
                        if (r >= R) continue;
                        uint32_t dnext = 0;
                        if (r + 1 < R) dnext = H[r] + score(r + 1);
                        const uint32_t h = pk_max3_f16(dsum, E[r], f);
                        // what the column maximum folds: h, or h with the row in its free low bits
                        const uint32_t hk = LOC ? h + both(kRowMask - r) : h;
                        if (r & 1) cm = pk_max3_f16(cm, held, hk);
                        else held = hk;
                        const uint32_t hmo = h - openMinusExt2;
                        E[r] = pk_max3_f16(E[r], hmo, fl1);
                        if (r + 1 < R) f = pk_max3_f16(f, hmo, fl1) - ext2;
                        H[r] = h;
                        dsum = dnext;
                    }
                    // pins the schedule: hipcc would otherwise hoist every ds_read to the column's top
                    asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
                }
                if (R & 1) cm = pk_max3_f16(cm, held, held);
                if constexpr (LOC) {
                    // strictly greater scores only: compare against best with its row bits all set
                    const uint32_t cand = cm - fl, thr = best | both(kRowMask);
                    const uint32_t ch = pk_max_u16(thr, cand) ^ thr;   // nonzero half: new maximum
                    const int j = c * 4 + cc;
                    if (ch & 0xffffu) {
                        best = (best & 0xffff0000u) | (cand & 0xffffu);
                        colA = j;
                    }
                    if (ch >> 16) {
                        best = (best & 0xffffu) | (cand & 0xffff0000u);
                        colB = j;
                    }
                } else {
                    best = pk_max_u16(best, cm - fl);
                }
            }
            cur = nxt;
            shift += 4 * ext;
            if (shift + 4 * ext > kBiasedMaxShift) {
                // rebase: H is on the last column's scale, E on the next one's, both relative to fl
                const uint32_t d = both(shift);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    H[r] -= d;
                    E[r] -= d;
                }
                fl -= d;
                shift = 0;
            }
        }
        const int lo = (int)(best & 0xffffu) >> kBits, hi = (int)(best >> 16) >> kBits;
        const size_t base = (size_t)g * kGroupTargets;
        if (a.directOut) {
            // database order at once (the host has made sure that no lane can leave its range and that
            // nothing else writes these results): no view-order array, no scatter kernel
            const int posA = (int)base + lane, posB = posA + kLanes;
            if (posA < a.directN) {
                const int id = a.directIds[posA];
                a.directOut[id] = lo;
                if constexpr (LOC) {
                    a.directEndI[id] = colA < 0 ? -1 : kRowMask - (int)(best & kRowMask);
                    a.directEndJ[id] = colA;
                }
            }
            if (posB < a.directN) {
                const int id = a.directIds[posB];
                a.directOut[id] = hi;
                if constexpr (LOC) {
                    a.directEndI[id] = colB < 0 ? -1 : kRowMask - (int)((best >> 16) & kRowMask);
                    a.directEndJ[id] = colB;
                }
            }
            continue;
        }
        a.score[base + lane] = lo;
        a.score[base + kLanes + lane] = hi;
        if constexpr (LOC) {
            a.endI[base + lane] = colA < 0 ? -1 : kRowMask - (int)(best & kRowMask);
            a.endI[base + kLanes + lane] = colB < 0 ? -1 : kRowMask - (int)((best >> 16) & kRowMask);
            a.endJ[base + lane] = colA;
            a.endJ[base + kLanes + lane] = colB;
        }
        if (a.overflow) {
            // (at most locLimit(kBits) / kBiasedLimit; lower when single steps are large, host.hip)
            a.overflow[base + lane] = lo >= a.biasedLimit;
            a.overflow[base + kLanes + lane] = hi >= a.biasedLimit;
        }
    }
    if (MIOPAL_HEADLINE_PACE && lane == 0) *pace.mine = INT32_MAX;   // (done: nobody waits for this one)
}

template <int R, bool LOC>
static hipError_t launchPairBiasedR(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    const size_t lds = PairLayout<R>::bytes(a.nSymbols) + 16 + kPaceInts * sizeof(int);  // table + per-SIMD counters (+ pacing)
    static uint64_t configured = 0;  // one bit per device; setting the attribute twice is harmless
    if (firstUseOnThisDevice(&configured)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&interseq_pair_biased_kernel<R, LOC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    // (every CU takes part even when the groups are fewer than its wavefronts: the first round hands
    // out group g to wavefront tier g / blocks of workgroup g % blocks, and a wavefront alone on its
    // SIMD sweeps its group three times faster than one of three)
    const int blocks = std::max(1, std::min(computeUnits, a.nGroups));
    hipLaunchKernelGGL((interseq_pair_biased_kernel<R, LOC>), dim3(blocks), dim3(kPairWaves * kLanes), lds, stream, a);
    return hipGetLastError();
}

template <int kLo, bool LOC>
static hipError_t launchPairBiased(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    if (a.nStrips != 1) return hipErrorInvalidValue;
    // rows: any even number in [kLo, kLo + 14]; the translation units interseq_swb16_{a,b,c,d}.hip
    // share the 32 instantiations
    switch (rowsPerStrip - kLo) {
        case 0: return launchPairBiasedR<kLo, LOC>(a, computeUnits, stream);
        case 2: return launchPairBiasedR<kLo + 2, LOC>(a, computeUnits, stream);
        case 4: return launchPairBiasedR<kLo + 4, LOC>(a, computeUnits, stream);
        case 6: return launchPairBiasedR<kLo + 6, LOC>(a, computeUnits, stream);
        case 8: return launchPairBiasedR<kLo + 8, LOC>(a, computeUnits, stream);
        case 10: return launchPairBiasedR<kLo + 10, LOC>(a, computeUnits, stream);
        case 12: return launchPairBiasedR<kLo + 12, LOC>(a, computeUnits, stream);
        case 14: return launchPairBiasedR<kLo + 14, LOC>(a, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

// ---- Smith-Waterman scores of SEVERAL strips on the pair table (round 2) -----------
// The biased-halves kernel above for queries of more than one strip. A pair table holds one strip
// of the query and fills the CU's LDS, so all twelve wavefronts of a workgroup work on the same
// strip: a workgroup takes units (batch of 12 groups, strip) from a counter, strip-major, rebuilds
// the table when the strip changes (150 KB of LDS writes, a few microseconds) and each wavefront
// sweeps its group through that strip. The last row of a strip reaches the strip below through HBM: per column and
// lane the pair (H, F) as patterns on that column's scale - both strips rebase their column shift at
// the same chunks, so a pattern means the same in either - 512 bytes per column and wavefront,
// written once and read once. The wavefront of (group, s) follows the wavefront of (group, s - 1) -
// which may still be running in another workgroup when the batches are few - two chunks behind: the producer publishes the
// number of finished chunks (release), the consumer acquires it before it fetches rows it has not
// seen published. Every taken unit's producer was taken before it and strip 0 waits for nobody, so
// the chain always moves; a consumer that nevertheless waits longer than about a second poisons its
// own progress counter and flags its lanes, and the next rung recomputes them.
// A group's maximum is the maximum over its strips: atomicMax into the (zeroed) view scores.
#ifndef MIOPAL_STRIP_ROWS_AHEAD
#define MIOPAL_STRIP_ROWS_AHEAD 2
#endif
#ifndef MIOPAL_STRIP_SLEEP
#define MIOPAL_STRIP_SLEEP 8
#endif
#ifndef MIOPAL_STRIP_SLACK
#define MIOPAL_STRIP_SLACK 0
#endif
// progress is published every (mask + 1) chunks and at the end of the sweep (round 3: every 8 chunks
// instead of 4, and the counter fetched a chunk before it is needed, change nothing measurable - the
// hand-over is not what the wavefronts wait for: profiles/r03_handover_experiments.txt)
#ifndef MIOPAL_STRIP_PUBLISH_MASK
#define MIOPAL_STRIP_PUBLISH_MASK 3
#endif
constexpr int kStripPoison = 1 << 30;
constexpr int kStripSpinCap = 1 << 21;    // x s_sleep 8 (512 cycles): about half a second

// KNOWN (round 3, second pass of an `end` search whose scores are beyond the row keys' range): the optimum
// of every lane half is known (a.known, view order, from the scores-only pass of this very kernel); the
// sweep is the same, but instead of a running maximum every column's maximum is compared with the optimum,
// and the first time they are equal - once per lane half and strip at most - a scan of the column's rows
// finds the first row that holds it. The (score, column, row) keys are the ones of LOC.
template <int R, bool LOC, bool KNOWN = false>
__global__ __launch_bounds__(kPairWaves * kLanes) void interseq_pair_strips_kernel(InterseqArgs a) {
    static_assert(!(LOC && KNOWN), "row keys or a known optimum");
    constexpr int SLOTS = PairLayout<R>::kRowSlots;
    constexpr int NB4 = (R + 3) / 4;
    // end locations (LOC): values scaled by 2^kBits, the row inside the strip in the low bits of what
    // the column maximum folds - the one-strip kernel's scheme, per strip
    constexpr int kBits = LOC ? locRowBits(R) : 0;
    constexpr int kRowMask = (1 << kBits) - 1;
    extern __shared__ uint4 pairs[];

    const int lane = threadIdx.x & 63;
    const int nSym = a.nSymbols;
    const int ext = a.gapExt << kBits;   // pattern units
    const uint32_t ext2 = both(ext), openMinusExt2 = both((a.gapOpen << kBits) - ext);
    const uint32_t zero2 = both(LOC ? kLocZero : kBiasedZero);
    const int nStrips = a.nStrips;
    const int perBatch = a.batchGroups;   // 12, or fewer when the groups are few (more CUs, faster wavefronts)
    const int nBatches = (a.nGroups + perBatch - 1) / perBatch;
    int* ctl = reinterpret_cast<int*>(pairs + nSym * nSym * SLOTS);   // 16 bytes behind the table
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tableStrip = -1;
    const int spinCap = a.stripSpinCap > 0 ? a.stripSpinCap : kStripSpinCap;
    SimdPace pace;
    pace.init(ctl + 4, lane);
    StripTimer timer;

    for (;;) {
        timer.start(0);
        if (threadIdx.x == 0) {
            int next = atomicAdd(a.unitCounter, 1);
            // Too many lanes have left the exact range: the host will redo the whole view on the next
            // rung whatever this launch still computes (end locations of long queries against long
            // targets under cheap gaps: scores in the thousands, a range of 384) - stop here.
            if (a.stripAbort && __hip_atomic_load(a.stripAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.stripAbortAt) {
                if (next < nBatches * nStrips) atomicAdd(a.stripGaveUp, a.stripAbortAt);
                next = INT32_MAX;
            }
            ctl[0] = next;
        }
        __syncthreads();   // (and: every wavefront has left the table of the unit before)
        const int u = __builtin_amdgcn_readfirstlane(ctl[0]);
        if (u >= nBatches * nStrips) {
            timer.stop(0);
            break;
        }
        // strip-major: strip s of every batch before strip s + 1 of any. The wavefront above is then
        // (number of batches) units ahead - usually finished, never just started: with batch-major
        // order every unit began by waiting for the unit taken a moment before it to get two chunks
        // ahead, and a chain of 20 strips idled a fifth of the time - and a workgroup keeps its table
        // until the strip changes.
        const int s = u / nBatches, b = u - s * nBatches;
        if (s != tableStrip) {
            const int16_t* gp = a.profile + s * R;
            uint32_t* pw = reinterpret_cast<uint32_t*>(pairs);
            const int total = nSym * nSym * R;
            for (int idx = threadIdx.x; idx < total; idx += kPairWaves * kLanes) {
                const int row = idx / R, r = idx - row * R;
                const int tA = row / nSym, tB = row - tA * nSym;
                constexpr int kPadPattern = LOC ? -kLocGuard : kBiasedPadScore;
                const int vA = gp[tA * a.qPad + r], vB = gp[tB * a.qPad + r];
                const int sA = (vA == kBiasedPadScore ? kPadPattern : vA << kBits) + ext;
                const int sB = (vB == kBiasedPadScore ? kPadPattern : vB << kBits) + ext;
                pw[row * (SLOTS * 4) + r] = (uint32_t)(sB * 65536 + sA);
            }
            tableStrip = s;
        }
        __syncthreads();   // table ready; ctl[0] read by everybody
        timer.stop(0);
        const int gIdx = b * perBatch + wave;
        if (wave >= perBatch || gIdx >= a.nGroups) continue;
        const int g = gIdx + a.groupBase;
        const uint2* pack = a.pack + a.groupOff[g];
        // (a leading group whose longest targets were handed to the int32 kernel stops at the longest that stays)
        const int nChunks = gIdx < a.capGroups ? min(a.groupChunks[g], a.capChunks) : a.groupChunks[g];
        // (a long group is the launch's critical path: it wins the SIMD's issue arbitration)
        const bool longGroup = nChunks > a.priorityChunks;
        if (longGroup) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        const bool fromAbove = s > 0, toBelow = s + 1 < nStrips;
        // boundary rows: [column][lane] (H, F) of the strip's last row
        // (wave-uniform bases and 32-bit element offsets: scalar base + one offset register per access)
        // The producer and the consumer of a row usually sit in different XCDs, whose L2 caches do not
        // see each other: the rows travel as agent-scope relaxed atomics (write-through stores, loads
        // that bypass the local L2), ordered against the progress counter by waiting for the memory
        // operations themselves - release / acquire FENCES at agent scope write back and invalidate
        // the whole L2 each time, 80 us per chunk when every wavefront of the chip does it.
        const StripRows rowsIn(a.boundary[(s + 1) & 1] + a.boundaryOff[g]), rowsOut(a.boundary[s & 1] + a.boundaryOff[g]);
        int* progOut = a.unitFlags + (size_t)gIdx * nStrips + s;
        const int* progIn = progOut - 1;
        const int lastCol = nChunks * 4 - 1;
        // chunks the strip above must be ahead: the rows of columns up to j + (rows ahead) are fetched
        constexpr int kLag = 1 + (MIOPAL_STRIP_ROWS_AHEAD + 3) / 4;
        int avail = fromAbove ? 0 : nChunks;   // chunks of the strip above known to be published
        // (test hook, host.hip miopalTestInjectFault: this unit behaves like one that died - no sweep, no
        // progress published, its lanes flagged - so that the strip below runs into its time-out)
        const bool injected = u + 1 == a.faultUnit1;
        bool dead = injected;
        auto waitFor = [&](int need) {
#ifdef MIOPAL_ABL_NO_POLL
            avail = nChunks;
#endif
            if (avail >= need) return;
            timer.start(1);
            // (wave-uniform: every lane reads the same counter)
            int spins = 0;
            while (avail < need) {
                avail = stripPoll(progIn);
                if (avail >= need) break;
                __builtin_amdgcn_s_sleep(MIOPAL_STRIP_SLEEP);
                if (++spins > spinCap) avail = kStripPoison;
                // (the unit above may never be taken once the launch has given up)
                if (a.stripAbort && __hip_atomic_load(a.stripAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.stripAbortAt)
                    avail = kStripPoison;
            }
            timer.stop(1);
            if (avail >= kStripPoison) dead = true;
        };
        if (!dead) waitFor(min(kLag + MIOPAL_STRIP_SLACK, nChunks));

        uint32_t best = 0u;                 // true values (LOC: keys), integer order
        int colA = -1, colB = -1;           // LOC: column of the first maximum of each half in this strip
        // KNOWN: the optimum of both halves as one packed value; a half without a positive optimum, or whose
        // optimum is beyond the exact range (redone by the next rung), has nothing to find
        uint32_t target2 = 0u;
        bool foundA = true, foundB = true;
        if constexpr (KNOWN) {
            const size_t kbase = (size_t)g * kGroupTargets;
            const int kA = a.known[kbase + lane], kB = a.known[kbase + kLanes + lane];
            foundA = kA <= 0 || kA >= a.biasedLimit;
            foundB = kB <= 0 || kB >= a.biasedLimit;
            target2 = ((uint32_t)(foundB ? 0xffff : kB) << 16) | (uint32_t)(foundA ? 0xffff : kA);
        }
        // the sweep, compiled for the three kinds of strip (first / inner / last): no selects between
        // border and row above, no stores from the last strip
        auto sweep = [&](auto fromAboveC, auto toBelowC) {
            constexpr bool kFromAbove = decltype(fromAboveC)::value, kToBelow = decltype(toBelowC)::value;
            uint32_t fl = zero2 - ext2;         // zero of column -1
            int shift = -ext;                   // fl = zero2 + both(shift); wave-uniform
            uint32_t H[R], E[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                H[r] = fl;
                E[r] = zero2;
            }
            // row above: column j in hand, the next kRowsAhead columns on their way (the loads go to
            // memory, past the L2: a microsecond or two, a column takes about one)
            // (two columns per load: the pair in use and the next one on its way - it is asked for at the pair's
            // first column and first read two columns later)
            StripU4 pq0 = {fl, zero2, fl, zero2}, pq1 = pq0;
            uint32_t keepH = 0, keepF = 0;      // the even column's {H, F} waiting for its neighbour's
            uint32_t hbPrev = fl;               // H of the row above at column j - 1 (left border: 0)
            if constexpr (kFromAbove) pq0 = rowsIn.load(0, lane);
            uint2 cur = pack[lane];
            auto rowOf = [&](uint32_t tA, uint32_t tB) -> const uint4* {
                const uint32_t rowIdx = __umul24(tA, (uint32_t)nSym) + tB;
                return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pairs) +
                                                      __umul24(rowIdx, (uint32_t)(SLOTS * 16)));
            };
            // (the second pass of an `end` search has a row scan inside the column loop: it gives the
            // registers of two prefetched blocks of pair-table rows for it)
            // (round 4: the boundary rows in 16-byte pairs cost four registers: tall strips give a prefetched block for them)
            constexpr int kWant = KNOWN ? 1 : R > (LOC ? 38 : 44) ? 2 : MIOPAL_PAIR_AHEAD;
            constexpr int kAhead = NB4 > kWant ? kWant : 1;
            const uint4* prowNext = rowOf(cur.x & 0xffu, cur.y & 0xffu);
            uint4 vn[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
            for (int c = 0; c < nChunks && !dead; ++c) {
                uint2 nxt = {0, 0};
                if (c + 1 < nChunks) nxt = pack[(size_t)(c + 1) * kLanes + lane];
                pace.step(c, lane, longGroup);
                if constexpr (kFromAbove) {
                    waitFor(min(c + kLag, nChunks));
                    // (a unit that has given up computes nothing more: what it would publish from rows that
                    // never arrived could be taken for real by a strip below that needs no further poll)
                    if (dead) break;
                }
                uint32_t ra = cur.x, rb = cur.y;
                // (the two columns of a pair as two copies of the body: which half of the pair a column reads and
                // whether it stores are compile-time then - as run-time selects they cost the kernel 24 registers)
                auto column = [&](auto oddC, int cc) {
                    constexpr bool odd = decltype(oddC)::value;
                    const int j = c * 4 + cc;
                    const uint4* prow = prowNext;
                    uint4 v[NB4];
#pragma unroll
                    for (int k = 0; k < kAhead; ++k) v[k] = vn[k];
                    if constexpr (kFromAbove && !odd) pq1 = rowsIn.load(min(j / 2 + 1, lastCol / 2), lane);
                    const uint32_t hAbove = odd ? pq0.z : pq0.x, fAbove = odd ? pq0.w : pq0.y;
                    ra = cc < 3 ? ra >> 8 : nxt.x;
                    rb = cc < 3 ? rb >> 8 : nxt.y;
                    prowNext = rowOf(ra & 0xffu, rb & 0xffu);
                    auto score = [&](int r) -> uint32_t {
                        const uint4 x = v[r >> 2];
                        const int k = r & 3;
                        return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                    };
                    // the row above at column j - 1, one column to the right; strip 0: the border's 0
                    uint32_t dsum = (kFromAbove ? hbPrev : fl) + score(0);
                    fl += ext2;                         // this column's zero
                    uint32_t fl1 = fl + ext2;           // the next column's
                    asm volatile("" : "+v"(fl1));
                    uint32_t f = kFromAbove ? fAbove : fl, cm = fl, held = fl;
#pragma unroll
                    for (int r4 = 0; r4 < NB4; ++r4) {
                        if (r4 + kAhead < NB4) v[r4 + kAhead] = prow[r4 + kAhead];
                        if (r4 == (NB4 > 3 ? NB4 - 3 : 0)) {
#pragma unroll
                            for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = r4 * 4 + k;
                            if (r >= R) continue;
                            uint32_t dnext = 0;
                            if (r + 1 < R) dnext = H[r] + score(r + 1);
                            const uint32_t h = pk_max3_f16(dsum, E[r], f);
                            const uint32_t hk = LOC ? h + both(kRowMask - r) : h;
                            if (r & 1) cm = pk_max3_f16(cm, held, hk);
                            else held = hk;
                            const uint32_t hmo = h - openMinusExt2;
                            E[r] = pk_max3_f16(E[r], hmo, fl1);
                            f = pk_max3_f16(f, hmo, fl1) - ext2;   // (after the last row: what the strip below starts from)
                            H[r] = h;
                            dsum = dnext;
                        }
                        asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
                    }
                    if (R & 1) cm = pk_max3_f16(cm, held, held);
                    if constexpr (LOC) {
                        // strictly greater scores only: compare against best with its row bits all set
                        const uint32_t cand = cm - fl, thr = best | both(kRowMask);
                        const uint32_t ch = pk_max_u16(thr, cand) ^ thr;   // nonzero half: new maximum
                        if (ch & 0xffffu) {
                            best = (best & 0xffff0000u) | (cand & 0xffffu);
                            colA = j;
                        }
                        if (ch >> 16) {
                            best = (best & 0xffffu) | (cand & 0xffff0000u);
                            colB = j;
                        }
                    } else if constexpr (KNOWN) {
                        const uint32_t x = (cm - fl) ^ target2;
                        const bool hitA = !foundA && (x & 0xffffu) == 0, hitB = !foundB && (x >> 16) == 0;
                        if (__builtin_amdgcn_ballot_w64(hitA || hitB) != 0) {
                            // the first row of this column that holds the optimum of its half
                            int ia = 0, ib = 0;
#pragma unroll
                            for (int r = R - 1; r >= 0; --r) {
                                const uint32_t d = (H[r] - fl) ^ target2;
                                if ((d & 0xffffu) == 0) ia = r;
                                if ((d >> 16) == 0) ib = r;
                                asm volatile("" : "+v"(ia), "+v"(ib));
                            }
                            // (only remembered here: the keys leave after the sweep, when the DP state is dead -
                            // anything more inside the column loop costs registers the loop does not have)
                            if (hitA) {
                                colA = (j << 6) | ia;
                                foundA = true;
                            }
                            if (hitB) {
                                colB = (j << 6) | ib;
                                foundB = true;
                            }
                        }
                    } else {
                        best = pk_max_u16(best, cm - fl);
                    }
                    // (compiled per kind of strip: a run-time branch here costs 27 registers)
                    if constexpr (kToBelow) {
                        if constexpr (odd) {
                            rowsOut.store(j / 2, lane, StripU4{keepH, keepF, H[R - 1], f});
                        } else {
                            keepH = H[R - 1];
                            keepF = f;
                        }
                    }
                    hbPrev = hAbove;
                    if constexpr (odd) pq0 = pq1;
                };
#pragma unroll 1
                for (int cp = 0; cp < 4; cp += 2) {
                    column(std::false_type{}, cp);
                    column(std::true_type{}, cp + 1);
                }
                cur = nxt;
                shift += 4 * ext;
                if (shift + 4 * ext > kBiasedMaxShift) {
                    // rebase (the strip above did the same after this chunk: the rows it wrote from
                    // the next chunk on are already on the new scale, the one kept in hbPrev is not)
                    const uint32_t d = both(shift);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        H[r] -= d;
                        E[r] -= d;
                    }
                    hbPrev -= d;
                    fl -= d;
                    shift = 0;
                }
                // (every fourth chunk and the last: the wait below also waits for the loads in flight)
                if (kToBelow && (((c + 1) & MIOPAL_STRIP_PUBLISH_MASK) == 0 || c + 1 == nChunks)) {
                    // every row store of the chunks has COMPLETED before the counter moves (stripPublish, common.h)
                    timer.start(2);
                    stripPublish(progOut, c + 1, lane);
                    timer.stop(2);
                }
            }
        };
        if (!dead) {
            pace.begin(lane);
            timer.start(3);
            if (!fromAbove) sweep(std::false_type{}, std::true_type{});
            else if (toBelow) sweep(std::true_type{}, std::true_type{});
            else sweep(std::true_type{}, std::false_type{});
            timer.stop(3);
            pace.end(lane);
        }
        const size_t base = (size_t)g * kGroupTargets;
        if (dead) {
            // the strip above never got here: leave the answer to the next rung, tell the strip below
            if (toBelow && lane == 0 && !injected) __hip_atomic_store(progOut, kStripPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.overflow) a.overflow[base + lane] = a.overflow[base + kLanes + lane] = 1;
            continue;
        }
        if constexpr (KNOWN) {
            // (score, column, row) of the first cell of this strip that holds the optimum; scores and flags are
            // the first pass's
            auto key = [&](uint32_t score, int where) -> unsigned long long {
                return ((unsigned long long)score << 40) | ((unsigned long long)(0xFFFFFu - (unsigned)(where >> 6)) << 20) |
                       (unsigned long long)(0xFFFFFu - (unsigned)(s * R + (where & 63)));
            };
            if (colA >= 0) atomicMax(a.stripKeys + base + lane, key(target2 & 0xffffu, colA));
            if (colB >= 0) atomicMax(a.stripKeys + base + kLanes + lane, key(target2 >> 16, colB));
            continue;
        }
        const int lo = (int)(best & 0xffffu) >> kBits, hi = (int)(best >> 16) >> kBits;
        if constexpr (LOC) {
            // first maximum in column-major order over the strips: highest score, then smallest column,
            // then smallest row, as one 64-bit key (a strip without a positive cell contributes nothing)
            auto key = [&](int score, int col, uint32_t half) -> unsigned long long {
                const int row = s * R + (kRowMask - (int)(half & kRowMask));
                return ((unsigned long long)(unsigned)score << 40) | ((unsigned long long)(0xFFFFFu - (unsigned)col) << 20) |
                       (unsigned long long)(0xFFFFFu - (unsigned)row);
            };
            if (colA >= 0) atomicMax(a.stripKeys + base + lane, key(lo, colA, best));
            if (colB >= 0) atomicMax(a.stripKeys + base + kLanes + lane, key(hi, colB, best >> 16));
        } else {
            atomicMax(a.score + base + lane, lo);
            atomicMax(a.score + base + kLanes + lane, hi);
        }
        if (a.overflow) {
            // (first flags only are counted: a lane that left the range in this strip usually does in the next)
            const bool fa = lo >= a.biasedLimit && a.overflow[base + lane] == 0;
            const bool fb = hi >= a.biasedLimit && a.overflow[base + kLanes + lane] == 0;
            if (fa) a.overflow[base + lane] = 1;
            if (fb) a.overflow[base + kLanes + lane] = 1;
            if (a.stripAbort) {
                const int fresh = __popcll(__builtin_amdgcn_ballot_w64(fa)) + __popcll(__builtin_amdgcn_ballot_w64(fb));
                if (fresh > 0 && lane == 0) atomicAdd(a.stripAbort, fresh);
            }
        }
    }
    timer.flush(a.stripTiming, lane);
}

template <int R, bool LOC, bool KNOWN = false>
static hipError_t launchPairStripsR(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    const size_t lds = PairLayout<R>::bytes(a.nSymbols) + 16 + kPaceInts * sizeof(int);  // table + the unit in flight + pacing
    static uint64_t configured = 0;  // one bit per device
    if (firstUseOnThisDevice(&configured)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&interseq_pair_strips_kernel<R, LOC, KNOWN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    const int nBatches = (a.nGroups + a.batchGroups - 1) / a.batchGroups;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(computeUnits, (int64_t)nBatches * a.nStrips));
    hipLaunchKernelGGL((interseq_pair_strips_kernel<R, LOC, KNOWN>), dim3(blocks), dim3(kPairWaves * kLanes), lds, stream, a);
    return hipGetLastError();
}

constexpr int kStripsMaxRows = 52;      // taller strips spill (the boundary rows cost 9 registers)
constexpr int kStripsMaxRowsLoc = 48;   // with end locations
constexpr int kStripsMaxRowsKnown = 40; // second pass of an `end` search (a row scan inside the column loop)
static_assert(kStripsMaxRows == kPairStripsMaxRows && kStripsMaxRowsLoc == kPairStripsMaxRowsLoc &&
              kStripsMaxRowsKnown == kPairStripsMaxRowsKnown, "common.h mirrors these");
template <int kLo, int kStep, bool LOC, bool KNOWN>
static hipError_t launchPairStripsCase(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    if constexpr (kLo + kStep <= (KNOWN ? kStripsMaxRowsKnown : LOC ? kStripsMaxRowsLoc : kStripsMaxRows))
        return launchPairStripsR<kLo + kStep, LOC, KNOWN>(a, computeUnits, stream);
    else return hipErrorInvalidValue;
}
template <int kLo, bool LOC, bool KNOWN = false>
static hipError_t launchPairStrips(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    if (a.nStrips < 2 || !a.unitCounter || !a.unitFlags || !a.boundary[0] || !a.boundary[1] || a.batchGroups < 1 ||
        a.batchGroups > kPairWaves || ((LOC || KNOWN) && !a.stripKeys) || (KNOWN && !a.known))
        return hipErrorInvalidValue;
    switch (rowsPerStrip - kLo) {
        case 0: return launchPairStripsCase<kLo, 0, LOC, KNOWN>(a, computeUnits, stream);
        case 2: return launchPairStripsCase<kLo, 2, LOC, KNOWN>(a, computeUnits, stream);
        case 4: return launchPairStripsCase<kLo, 4, LOC, KNOWN>(a, computeUnits, stream);
        case 6: return launchPairStripsCase<kLo, 6, LOC, KNOWN>(a, computeUnits, stream);
        case 8: return launchPairStripsCase<kLo, 8, LOC, KNOWN>(a, computeUnits, stream);
        case 10: return launchPairStripsCase<kLo, 10, LOC, KNOWN>(a, computeUnits, stream);
        case 12: return launchPairStripsCase<kLo, 12, LOC, KNOWN>(a, computeUnits, stream);
        case 14: return launchPairStripsCase<kLo, 14, LOC, KNOWN>(a, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

// ---- one-strip NW / HW / OV on biased integer halves, column-shifted --------------
// The pair-table kernel for the modes without a floor (round 2). Same representation as above - a
// half holds zero + x + sigma(j) and is compared as a half float, additions are 32-bit integer adds,
// values of column j carry sigma(j) = j * ext so that extending E is free - with what the other
// modes need:
//  * borders: H[-1][j], H[i][-1] = 0 or the gap of j + 1 / i + 1 residues (borderGap), entered per
//    column as wave-uniform patterns;
//  * no floor: E = max(E, hmo), F = max(F, hmo) - ext; 3 integer adds + 3 max per cell pair (the
//    general kernel's shifted int16 flavour: 6 packed operations + 1 v_perm);
//  * the answer of a lane half is read at its own last column / on the last query row, turned into a
//    true 32-bit value x = pattern - (zero + sigma(j)): the lanes never leave their range because of
//    a target's length (NW scores of 35000-residue targets are far below -32768, the pattern is not),
//    so nothing is flagged and no target is redone at 32 bit. The host checks the static bounds
//    (zero covers 2 open + (Q + 2) ext + |min S|, Q (max S + ext) + zero + kBiasedMaxShift < 0x7C00);
//  * with a free top border (HW, OV) x is bounded and sigma is rebased like the Smith-Waterman
//    kernel's; with a penalised one (NW) x + j ext is bounded and sigma just grows (as an int).
// End locations (optional): the scan rules of oracle/opal_oracle.c on those per-column values.
// Twelve wavefronts per workgroup (three per SIMD, 168 VGPRs) up to 54 rows, eight (256 VGPRs) beyond:
// beside H[R], E[R] this kernel keeps per-lane answers, lengths and locations.
__host__ __device__ constexpr int globalWaves(int rows) { return rows <= 54 ? 12 : 8; }

template <int R>
__global__ __launch_bounds__(globalWaves(R) * kLanes) void interseq_pair_global_kernel(InterseqArgs a) {
    constexpr int kGlobalWaves = globalWaves(R);
    constexpr int SLOTS = PairLayout<R>::kRowSlots;
    constexpr int NB4 = (R + 3) / 4;
    extern __shared__ uint4 pairs[];

    const int lane = threadIdx.x & 63;
    const int nSym = a.nSymbols;
    const int ext = a.gapExt, open = a.gapOpen, Q = a.qLen;
    const int zero = a.biasedZero;
    const bool topGap = a.topGap, leftGap = a.leftGap;
    const int region = a.region;
    const bool locate = a.endI != nullptr;
    const uint32_t openMinusExt2 = both(open - ext);

    // table: s'' = s + 2 ext + c = s + ext + open of both targets as one integer (round 3: every cell (r, j) is
    // on the scale zero + sigma(j) + r ext, H kept c = open - ext below its plain form: 2 integer adds + 3
    // max per cell pair, see interseq_pair_global_strips_kernel); padding symbol / rows add c
    {
        const int16_t* gp = a.profile;
        uint32_t* pw = reinterpret_cast<uint32_t*>(pairs);
        const int total = nSym * nSym * R;
        for (int idx = threadIdx.x; idx < total; idx += kGlobalWaves * kLanes) {
            const int row = idx / R, r = idx - row * R;
            const int tA = row / nSym, tB = row - tA * nSym;
            const int vA = gp[tA * a.qPad + r], vB = gp[tB * a.qPad + r];
            const int sA = vA == kBiasedPadScore ? open - ext : vA + ext + open;
            const int sB = vB == kBiasedPadScore ? open - ext : vB + ext + open;
            pw[row * (SLOTS * 4) + r] = (uint32_t)(sB * 65536 + sA);
        }
    }
    int* simdTaken = reinterpret_cast<int*>(pairs + nSym * nSym * SLOTS);
    if (threadIdx.x < 4) simdTaken[threadIdx.x] = 0;
    __syncthreads();
    SimdPace pace;
    if (MIOPAL_HEADLINE_PACE) pace.init(simdTaken + 4, lane);

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int kTier12[4][3] = {{0, 6, 11}, {1, 7, 8}, {2, 5, 9}, {3, 4, 10}};
    constexpr int kTier8[4][2] = {{0, 7}, {1, 6}, {2, 5}, {3, 4}};
    const int tier = kGlobalWaves == 12 ? kTier12[wave & 3][wave >> 2] : kTier8[wave & 3][(wave >> 2) & 1];
    const int firstDynamic = kGlobalWaves * gridDim.x;
    const int simd = (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | ((2 - 1) << 11)) & 3;
    bool firstRound = true;
    for (;;) {
        int g;
        if (firstRound) {
            g = tier * gridDim.x + ((tier & 1) ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x);
            firstRound = false;
            if (lane == 0 && a.tailThrottle > 0) atomicAdd(&simdTaken[simd], 1);
            if (g >= a.nGroups) continue;
            g += a.groupBase;
        } else {
            g = 0;
            if (lane == 0) {
                bool take = true;
                if (a.tailThrottle > 0) take = atomicAdd(&simdTaken[simd], 1) < a.tailThrottle;
                g = take ? atomicAdd(a.workCounter, 1) : INT32_MAX - firstDynamic;
            }
            g = __builtin_amdgcn_readfirstlane(g) + firstDynamic;
            if (g >= a.nGroups) break;
            g += a.groupBase;
        }
        const uint2* pack = a.pack + a.groupOff[g];
        const int nChunks = a.groupChunks[g];
        if (nChunks > a.priorityChunks) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        const size_t base = (size_t)g * kGroupTargets;
        const int lenA = a.lens[base + lane], lenB = a.lens[base + kLanes + lane];

        // answers (true values) and where they were found
        int runA = INT32_MIN, runB = INT32_MIN, colA = -1, colB = -1, rowA = -1, rowB = -1;
        int cbA = INT32_MIN, cbB = INT32_MIN, crowA = -1, crowB = -1;   // OV: best of the last column

        int sigma = zero - ext;             // zero + sigma(j) of the last column done (column -1 here)
        int shift = -ext;                   // part of sigma accumulated since the last rebase
        uint32_t H[R], E[R];
        {
            // (opaque to the optimiser: the 2 R border values are the same for every group, and hoisting
            // them out of the group loop would cost 2 R registers for the whole kernel)
            int zeroHere = zero, openHere = open, extHere = ext;
            asm volatile("" : "+s"(zeroHere), "+s"(openHere), "+s"(extHere));
            // H[r][-1]: one gap of r + 1 residues or r + 1 one-residue gaps (borderGap), as running sums
            int one = openHere, many = openHere, rowShift = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int left = (leftGap ? -min(one, many) : 0) + rowShift;
                H[r] = both(zeroHere - extHere + left - (openHere - extHere));   // stored form, scale of (r, -1)
                E[r] = both(zeroHere + left - openHere);                  // E[r][0], plain form, scale of (r, 0)
                one += extHere;
                many += openHere;
                rowShift += extHere;
            }
        }
        uint2 cur = pack[lane];
        auto rowOf = [&](uint32_t tA, uint32_t tB) -> const uint4* {
            const uint32_t rowIdx = __umul24(tA, (uint32_t)nSym) + tB;
            return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pairs) +
                                                  __umul24(rowIdx, (uint32_t)(SLOTS * 16)));
        };
        constexpr int kWant = 3;
        constexpr int kAhead = NB4 > kWant ? kWant : 1;
        const uint4* prowNext = rowOf(cur.x & 0xffu, cur.y & 0xffu);
        uint4 vn[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
        for (int c = 0; c < nChunks; ++c) {
            uint2 nxt = {0, 0};
            if (c + 1 < nChunks) nxt = pack[(size_t)(c + 1) * kLanes + lane];
            if (MIOPAL_HEADLINE_PACE) pace.step(c - nChunks, lane, nChunks > a.priorityChunks);
            uint32_t ra = cur.x, rb = cur.y;
#pragma unroll 1
            for (int cc = 0; cc < 4; ++cc) {
                const int j = c * 4 + cc;
                const uint4* prow = prowNext;
                uint4 v[NB4];
#pragma unroll
                for (int k = 0; k < kAhead; ++k) v[k] = vn[k];
                ra = cc < 3 ? ra >> 8 : nxt.x;
                rb = cc < 3 ? rb >> 8 : nxt.y;
                prowNext = rowOf(ra & 0xffu, rb & 0xffu);
                auto score = [&](int r) -> uint32_t {
                    const uint4 x = v[r >> 2];
                    const int k = r & 3;
                    return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                };
                // row above the strip: H[-1][j-1] on the previous column's scale, H[-1][j] on this one's
                const int topPrev = (j == 0 || !topGap) ? 0 : borderGap(j - 1, open, ext);
                const int topHere = topGap ? borderGap(j, open, ext) : 0;
                // (H[-1][j-1] in stored form on the scale of (-1, j - 1))
                uint32_t dsum = both(sigma + topPrev - open) + score(0);
                sigma += ext;
                uint32_t f = both(sigma + topHere - open);      // F entering row 0
                asm volatile("" : "+v"(f));
#pragma unroll
                for (int r4 = 0; r4 < NB4; ++r4) {
                    if (r4 + kAhead < NB4) v[r4 + kAhead] = prow[r4 + kAhead];
                    if (r4 == (NB4 > 3 ? NB4 - 3 : 0)) {
#pragma unroll
                        for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int r = r4 * 4 + k;
                        if (r >= R) continue;
                        uint32_t dnext = 0;
                        if (r + 1 < R) dnext = H[r] + score(r + 1);
                        const uint32_t h = pk_max3_f16(dsum, E[r], f);
                        const uint32_t hmo = h - openMinusExt2;
                        E[r] = pk_max2_f16(E[r], hmo);
                        // (here, not whenever the scheduler likes: E is off the critical path, and the R
                        // deferred updates would each hold on to their hmo)
                        asm volatile("" : "+v"(E[r]));
                        if (r + 1 < R) f = pk_max2_f16(f, hmo);
                        H[r] = hmo;
                        dsum = dnext;
                    }
                    // pins the schedule: no ds_read and none of the NEXT blocks' diagonal sums (they only
                    // need last column's H, all of it available at the column's top: hoisting them costs
                    // R registers) may start before this block's cells are done
                    asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
#pragma unroll
                    for (int k = 1; k <= 4; ++k)
                        if (r4 * 4 + 3 + k < R) asm volatile("" : "+v"(H[r4 * 4 + 3 + k]));
                }
                // ---- answers: the last query row is row R - 1 or R - 2 (R = Q rounded up to even)
                const uint32_t hq = (Q & 1) ? H[R >= 2 ? R - 2 : 0] : H[R - 1];
                // (stored form on the scale of (Q - 1, j): true value = pattern - sigma - (Q - 1) ext + c)
                const int back = sigma + (Q - 1) * ext - (open - ext);
                const int qA = (int)(hq & 0xffffu) - back, qB = (int)(hq >> 16) - back;
                if (region == kLastCell) {
                    if (j == lenA - 1) { runA = qA; colA = j; }
                    if (j == lenB - 1) { runB = qB; colB = j; }
                } else {
                    // last row: HW scans columns < len, OV columns < len - 1 (its last column is
                    // scanned row by row below)
                    const int cut = region == kLastRowCol ? 1 : 0;
                    if (j < lenA - cut && qA > runA) { runA = qA; colA = j; }
                    if (j < lenB - cut && qB > runB) { runB = qB; colB = j; }
                    const bool lastA = j == lenA - 1, lastB = j == lenB - 1;
                    if (region == kLastRowCol && __builtin_amdgcn_ballot_w64(lastA || lastB) != 0) {
                        // some lane is on its target's last column: first maximum over the query rows
                        int mA = INT32_MIN, mB = INT32_MIN, ia = 0, ib = 0;
                        int off = (R - 1) * ext;   // row r is on the scale of column j plus r ext
                        asm volatile("" : "+v"(off));
#pragma unroll
                        for (int r = R - 1; r >= 0; --r) {
                            // (rows beyond the query: only row R - 1, when Q is odd; one test, not one
                            // per row - per-row predicates would be hoisted into a hundred SGPRs)
                            if (!(r == R - 1 && (Q & 1))) {
                                const int hA = (int)(H[r] & 0xffffu) - off, hB = (int)(H[r] >> 16) - off;
                                if (hA >= mA) { mA = hA; ia = r; }
                                if (hB >= mB) { mB = hB; ib = r; }
                            }
                            off -= ext;
                            // (row by row: the scheduler would otherwise unpack all 2 R halves up front)
                            asm volatile("" : "+v"(mA), "+v"(mB), "+v"(off));
                        }
                        const int back = sigma - (open - ext);
                        if (lastA) { cbA = mA - back; crowA = ia; }
                        if (lastB) { cbB = mB - back; crowB = ib; }
                    }
                }
            }
            cur = nxt;
            shift += 4 * ext;
            if (!topGap && shift + 4 * ext > kBiasedMaxShift) {
                const uint32_t d = both(shift);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    H[r] -= d;
                    E[r] -= d;
                }
                sigma -= shift;
                shift = 0;
            }
        }
        if (region != kLastCell || true) {
            rowA = colA >= 0 ? Q - 1 : -1;
            rowB = colB >= 0 ? Q - 1 : -1;
        }
        if (region == kLastRowCol) {
            // OV: the last column is scanned after the last row's earlier columns, strictly greater wins
            if (crowA >= 0 && (colA < 0 || cbA > runA)) { runA = cbA; rowA = crowA; colA = lenA - 1; }
            if (crowB >= 0 && (colB < 0 || cbB > runB)) { runB = cbB; rowB = crowB; colB = lenB - 1; }
        }
        if (a.directOut) {
            // database order at once (nothing else writes these results: see the Smith-Waterman kernel)
            const int posA = (int)base + lane, posB = posA + kLanes;
            if (posA < a.directN) {
                const int id = a.directIds[posA];
                a.directOut[id] = runA;
                if (locate) {
                    a.directEndI[id] = rowA;
                    a.directEndJ[id] = colA;
                }
            }
            if (posB < a.directN) {
                const int id = a.directIds[posB];
                a.directOut[id] = runB;
                if (locate) {
                    a.directEndI[id] = rowB;
                    a.directEndJ[id] = colB;
                }
            }
            continue;
        }
        a.score[base + lane] = runA;
        a.score[base + kLanes + lane] = runB;
        if (locate) {
            a.endI[base + lane] = rowA;
            a.endI[base + kLanes + lane] = rowB;
            a.endJ[base + lane] = colA;
            a.endJ[base + kLanes + lane] = colB;
        }
        if (a.overflow) {
            a.overflow[base + lane] = 0;
            a.overflow[base + kLanes + lane] = 0;
        }
    }
    if (MIOPAL_HEADLINE_PACE && lane == 0) *pace.mine = INT32_MAX;   // (done: nobody waits for this one)
}

template <int R>
static hipError_t launchPairGlobalR(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    const size_t lds = PairLayout<R>::bytes(a.nSymbols) + 16 + kPaceInts * sizeof(int);
    static uint64_t configured = 0;
    if (firstUseOnThisDevice(&configured)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&interseq_pair_global_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    constexpr int kGlobalWaves = globalWaves(R);
    const int blocks = std::max(1, std::min(computeUnits, a.nGroups));   // (see launchPairBiasedR)
    hipLaunchKernelGGL((interseq_pair_global_kernel<R>), dim3(blocks), dim3(kGlobalWaves * kLanes), lds, stream, a);
    return hipGetLastError();
}

template <int kLo>
static hipError_t launchPairGlobal(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    if (a.nStrips != 1) return hipErrorInvalidValue;
    switch (rowsPerStrip - kLo) {
        case 0: return launchPairGlobalR<kLo>(a, computeUnits, stream);
        case 2: return launchPairGlobalR<kLo + 2>(a, computeUnits, stream);
        case 4: return launchPairGlobalR<kLo + 4>(a, computeUnits, stream);
        case 6: return launchPairGlobalR<kLo + 6>(a, computeUnits, stream);
        case 8: return launchPairGlobalR<kLo + 8>(a, computeUnits, stream);
        case 10: return launchPairGlobalR<kLo + 10>(a, computeUnits, stream);
        case 12: return launchPairGlobalR<kLo + 12>(a, computeUnits, stream);
        case 14: return launchPairGlobalR<kLo + 14>(a, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

// H[row] for a wave-uniform, run-time `row` in [LO, HI]: a binary tree of scalar branches (the opaque
// move in the leaves keeps the compiler from turning the tree into HI - LO selects on hoisted predicates)
template <int LO, int HI, int N>
static __device__ __forceinline__ uint32_t pickRow(const uint32_t (&H)[N], int row) {
    if constexpr (LO >= HI) {
        uint32_t v = H[LO];
        asm volatile("" : "+v"(v));
        return v;
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (row <= MID) return pickRow<LO, MID>(H, row);
        return pickRow<MID + 1, HI>(H, row);
    }
}

// ---- NW / HW / OV of SEVERAL strips on the pair table (round 3) ---------------------
// The one-strip kernel's cell above (3 integer adds + 3 max per cell pair, no v_perm) in the unit scheme
// of interseq_pair_strips_kernel: a workgroup takes (batch of up to 12 groups, strip) units from a
// counter, strip-major, rebuilds its pair table when the strip changes, and the last row of a strip
// reaches the strip below through HBM as relaxed agent-scope atomics behind a progress counter
// (stripPublish / stripPoll, common.h). What the modes without a floor need on top:
//  * borders: the left border H[i][-1] of the strip's own rows as running sums, the top border only in
//    strip 0 (the other strips start from the row above), both as wave-uniform patterns;
//  * the scale: a half holds zero + x + sigma(j). With a free top border (HW, OV) x is bounded by the
//    QUERY (not the strip) and sigma is rebased at the same chunks in every strip, so a pattern means
//    the same on both sides of a boundary; with a penalised one (NW) x + j ext is bounded and sigma
//    just grows. The host checks the static range for the whole query (host.hip, globalStrips);
//  * answers: only the last strip holds the last query row (NW: its value at each lane's own last
//    column; HW / OV: maximum over the columns), and OV also takes the maximum over the rows of each
//    target's last column, strip by strip.
//    Scores only (LOC = false): the last strip stores the last-row answer (NW, HW) or every strip folds
//    its candidates into the view scores with atomicMax (OV; the host fills them with INT32_MIN).
//    With end locations (LOC = true) every candidate becomes a 64-bit key
//      (score + 2^31) << 32 | last-row candidate << 31 | 0x7FFFFFFF - (column | row)
//    merged with atomicMax: highest score; at equal scores the last row before the last column (the scan
//    order of oracle/opal_oracle.c), then the smallest column / row. decode_global_keys_kernel
//    (pack.hip) turns the keys into view-order scores and end locations. The row and column
//    bookkeeping costs registers: strips of at most 48 rows (52 without);
//  * nothing leaves its range, so nothing is flagged - except the lanes of a unit whose strip above
//    never arrived (time-out escape, as in the Smith-Waterman kernel): they are flagged for the int32
//    kernel and the strip below is told (poison).
template <int R, bool LOC>
__global__ __launch_bounds__(kPairWaves * kLanes) void interseq_pair_global_strips_kernel(InterseqArgs a) {
    constexpr int SLOTS = PairLayout<R>::kRowSlots;
    constexpr int NB4 = (R + 3) / 4;
    extern __shared__ uint4 pairs[];

    const int lane = threadIdx.x & 63;
    const int nSym = a.nSymbols;
    const int ext = a.gapExt, open = a.gapOpen, Q = a.qLen;
    const int zero = a.biasedZero;
    const bool topGap = a.topGap, leftGap = a.leftGap;
    const int region = a.region;
    const uint32_t ext2 = both(ext), openMinusExt2 = both(open - ext);
    const int nStrips = a.nStrips;
    const int perBatch = a.batchGroups;
    const int nBatches = (a.nGroups + perBatch - 1) / perBatch;
    int* ctl = reinterpret_cast<int*>(pairs + nSym * nSym * SLOTS);   // 16 bytes behind the table
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tableStrip = -1;
    const int spinCap = a.stripSpinCap > 0 ? a.stripSpinCap : kStripSpinCap;
    const int rl = Q - 1 - (nStrips - 1) * R;   // row of the last query residue inside the last strip
    SimdPace pace;
    pace.init(ctl + 4, lane);
    StripTimer timer;

    for (;;) {
        timer.start(0);
        if (threadIdx.x == 0) ctl[0] = atomicAdd(a.unitCounter, 1);
        __syncthreads();   // (and: every wavefront has left the table of the unit before)
        const int u = __builtin_amdgcn_readfirstlane(ctl[0]);
        if (u >= nBatches * nStrips) {
            timer.stop(0);
            break;
        }
        const int s = u / nBatches, b = u - s * nBatches;   // strip-major (see interseq_pair_strips_kernel)
        if (s != tableStrip) {
            // s'' = s + 2 ext + c = s + ext + open of both targets as one integer (the diagonal step crosses
            // two anti-diagonals, and H is kept in its stored form, c = open - ext below its plain form);
            // padding symbol / rows add c: the pattern of a padding cell is the one of its diagonal neighbour
            const int16_t* gp = a.profile + s * R;
            uint32_t* pw = reinterpret_cast<uint32_t*>(pairs);
            const int total = nSym * nSym * R;
            for (int idx = threadIdx.x; idx < total; idx += kPairWaves * kLanes) {
                const int row = idx / R, r = idx - row * R;
                const int tA = row / nSym, tB = row - tA * nSym;
                const int vA = gp[tA * a.qPad + r], vB = gp[tB * a.qPad + r];
                const int sA = vA == kBiasedPadScore ? open - ext : vA + ext + open;
                const int sB = vB == kBiasedPadScore ? open - ext : vB + ext + open;
                pw[row * (SLOTS * 4) + r] = (uint32_t)(sB * 65536 + sA);
            }
            tableStrip = s;
        }
        __syncthreads();   // table ready; ctl[0] read by everybody
        timer.stop(0);
        const int gIdx = b * perBatch + wave;
        if (wave >= perBatch || gIdx >= a.nGroups) continue;
        const int g = gIdx + a.groupBase;
        const uint2* pack = a.pack + a.groupOff[g];
        const int nChunks = gIdx < a.capGroups ? min(a.groupChunks[g], a.capChunks) : a.groupChunks[g];
        const bool longGroup = nChunks > a.priorityChunks;
        if (longGroup) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
        const bool fromAbove = s > 0, toBelow = s + 1 < nStrips;
        const StripRows rowsIn(a.boundary[(s + 1) & 1] + a.boundaryOff[g]), rowsOut(a.boundary[s & 1] + a.boundaryOff[g]);
        int* progOut = a.unitFlags + (size_t)gIdx * nStrips + s;
        const int* progIn = progOut - 1;
        const int lastCol = nChunks * 4 - 1;
        constexpr int kLag = 1 + (MIOPAL_STRIP_ROWS_AHEAD + 3) / 4;
        int avail = fromAbove ? 0 : nChunks;   // chunks of the strip above known to be published
        // (test hook, host.hip miopalTestInjectFault: this unit behaves like one that died - no sweep, no
        // progress published, its lanes flagged - so that the strip below runs into its time-out)
        const bool injected = u + 1 == a.faultUnit1;
        bool dead = injected;
        auto waitFor = [&](int need) {
#ifdef MIOPAL_ABL_NO_POLL
            avail = nChunks;
#endif
            if (avail >= need) return;
            timer.start(1);
            int spins = 0;
            while (avail < need) {
                avail = stripPoll(progIn);
                if (avail >= need) break;
                __builtin_amdgcn_s_sleep(MIOPAL_STRIP_SLEEP);
                if (++spins > spinCap) avail = kStripPoison;
            }
            timer.stop(1);
            if (avail >= kStripPoison) dead = true;
        };
        if (!dead) waitFor(min(kLag + MIOPAL_STRIP_SLACK, nChunks));

        const size_t base = (size_t)g * kGroupTargets;
        const int lenA = a.lens[base + lane], lenB = a.lens[base + kLanes + lane];
        // candidates (true values) of this unit: last query row (last strip), last column (OV, any strip);
        // their columns / rows only with LOC
        int runA = INT32_MIN, runB = INT32_MIN, colA = -1, colB = -1;
        int cbA = INT32_MIN, cbB = INT32_MIN, crowA = -1, crowB = -1;

        // the sweep, compiled for the three kinds of strip (first / inner / last)
        auto sweep = [&](auto fromAboveC, auto toBelowC) {
            constexpr bool kFromAbove = decltype(fromAboveC)::value, kToBelow = decltype(toBelowC)::value;
            int sigma = zero - ext;             // zero + sigma(j) of the last column done (column -1 here)
            int shift = -ext;                   // part of sigma accumulated since the last rebase
            uint32_t H[R], E[R];
            {
                // H[i][-1], i = s R + r: one gap of i + 1 residues or i + 1 one-residue gaps (borderGap), as
                // running sums in VECTOR registers (wave-uniform values: left to itself the compiler keeps
                // all 2 R of them in scalar registers first, and spills hundreds)
                const int i0 = s * R;
                int one = open + i0 * ext, many = (i0 + 1) * open;
                int rowShift = 0;                                      // r ext: the anti-diagonal part of the scale
                asm volatile("" : "+v"(one), "+v"(many), "+v"(rowShift));
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int left = (leftGap ? -min(one, many) : 0) + rowShift;
                    H[r] = both(zero - ext + left - (open - ext));    // stored form, on the scale of (r, -1)
                    E[r] = both(zero + left - open);                  // E[i][0], plain form on the scale of (r, 0)
                    one += ext;
                    many += open;
                    rowShift += ext;
                }
            }
            StripU4 pq0 = {0u, 0u, 0u, 0u}, pq1 = pq0;   // the pair of columns in use, the next one on its way
            uint32_t keepH = 0, keepF = 0;
            // H of the row above at column j - 1, on that column's scale: the left border of row s R - 1
            uint32_t hbPrev = 0u;
            if constexpr (kFromAbove) {
                // (stored form on the scale of (-1, -1): one row above the strip's row 0)
                hbPrev = both(zero - ext + (leftGap ? borderGap(s * R - 1, open, ext) : 0) - ext - (open - ext));
                pq0 = rowsIn.load(0, lane);
            }
            uint2 cur = pack[lane];
            auto rowOf = [&](uint32_t tA, uint32_t tB) -> const uint4* {
                const uint32_t rowIdx = __umul24(tA, (uint32_t)nSym) + tB;
                return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(pairs) +
                                                      __umul24(rowIdx, (uint32_t)(SLOTS * 16)));
            };
            constexpr int kWant = R > 50 ? 1 : R > (LOC ? 38 : 44) ? 2 : MIOPAL_PAIR_AHEAD;
            constexpr int kAhead = NB4 > kWant ? kWant : 1;
            const uint4* prowNext = rowOf(cur.x & 0xffu, cur.y & 0xffu);
            uint4 vn[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
            // top border of the matrix (strip 0), as running values: H[-1][j - 1] and H[-1][j]
            int topPrev = 0, topHere = topGap ? borderGap(0, open, ext) : 0;
            for (int c = 0; c < nChunks && !dead; ++c) {
                uint2 nxt = {0, 0};
                if (c + 1 < nChunks) nxt = pack[(size_t)(c + 1) * kLanes + lane];
                pace.step(c, lane, longGroup);
                if constexpr (kFromAbove) {
                    waitFor(min(c + kLag, nChunks));
                    // (a unit that has given up computes nothing more: what it would publish from rows that
                    // never arrived could be taken for real by a strip below that needs no further poll)
                    if (dead) break;
                }
                uint32_t ra = cur.x, rb = cur.y;
                // (the two columns of a pair as two copies of the body: see interseq_pair_strips_kernel)
                auto column = [&](auto oddC, int cc) {
                    constexpr bool odd = decltype(oddC)::value;
                    const int j = c * 4 + cc;
                    const uint4* prow = prowNext;
                    uint4 v[NB4];
#pragma unroll
                    for (int k = 0; k < kAhead; ++k) v[k] = vn[k];
                    if constexpr (kFromAbove && !odd) pq1 = rowsIn.load(min(j / 2 + 1, lastCol / 2), lane);
                    const uint32_t hAbove = odd ? pq0.z : pq0.x, fAbove = odd ? pq0.w : pq0.y;
                    ra = cc < 3 ? ra >> 8 : nxt.x;
                    rb = cc < 3 ? rb >> 8 : nxt.y;
                    prowNext = rowOf(ra & 0xffu, rb & 0xffu);
                    auto score = [&](int r) -> uint32_t {
                        const uint4 x = v[r >> 2];
                        const int k = r & 3;
                        return k == 0 ? x.x : k == 1 ? x.y : k == 2 ? x.z : x.w;
                    };
                    uint32_t dsum, f;
                    if constexpr (kFromAbove) {
                        dsum = hbPrev + score(0);       // the row above at column j - 1, one column to the right
                        sigma += ext;
                        f = fAbove;                     // F entering the strip's first row, on this column's scale
                    } else {
                        // row above the matrix: H[-1][j-1] in stored form on the scale of (-1, j - 1), the F that
                        // H[-1][j] opens in plain form on the scale of (0, j)
                        dsum = both(sigma + topPrev - open) + score(0);
                        sigma += ext;
                        f = both(sigma + topHere - open);
                        asm volatile("" : "+v"(f));
                        topPrev = topHere;
                        if (topGap) topHere = borderGap(j + 1, open, ext);
                    }
#pragma unroll
                    for (int r4 = 0; r4 < NB4; ++r4) {
                        if (r4 + kAhead < NB4) v[r4 + kAhead] = prow[r4 + kAhead];
                        if (r4 == (NB4 > 3 ? NB4 - 3 : 0)) {
#pragma unroll
                            for (int k = 0; k < kAhead; ++k) vn[k] = prowNext[k];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = r4 * 4 + k;
                            if (r >= R) continue;
                            uint32_t dnext = 0;
                            if (r + 1 < R) dnext = H[r] + score(r + 1);
                            // 2 integer adds + 1 max3 + 2 max: every cell (r, j) is on the scale zero + sigma(j) +
                            // r ext, so extending either gap costs nothing, both open with hmo = h - c, and hmo is
                            // at once the stored form of h that the next column's diagonal sum starts from
                            const uint32_t h = pk_max3_f16(dsum, E[r], f);
                            const uint32_t hmo = h - openMinusExt2;
                            E[r] = pk_max2_f16(E[r], hmo);
                            asm volatile("" : "+v"(E[r]));
                            // (after the last row: what the strip below starts from)
                            if (kToBelow || r + 1 < R) f = pk_max2_f16(f, hmo);
                            H[r] = hmo;
                            dsum = dnext;
                        }
                        asm volatile("" : "+v"(f), "+v"(dsum)::"memory");
#pragma unroll
                        for (int k = 1; k <= 4; ++k)
                            if (r4 * 4 + 3 + k < R) asm volatile("" : "+v"(H[r4 * 4 + 3 + k]));
                    }
                    if constexpr (kToBelow) {
                        // (row R of this strip is row 0 of the strip below, whose scale starts again at r = 0)
                        const uint32_t down = both(R * ext);
                        if constexpr (odd) {
                            rowsOut.store(j / 2, lane, StripU4{keepH, keepF, H[R - 1] - down, f - down});
                        } else {
                            keepH = H[R - 1] - down;
                            keepF = f - down;
                        }
                    }
                    if constexpr (kFromAbove) {
                        hbPrev = hAbove;
                        if constexpr (odd) pq0 = pq1;
                    }
                    // ---- answers
                    if constexpr (!kToBelow) {
                        // the last query row: row rl of this strip, the same for the whole launch - usually
                        // the strip's last (the host prefers strip heights that divide the query). Otherwise a
                        // tree of wave-uniform BRANCHES finds it (selects would be R hoisted scalar predicates).
                        uint32_t hq = H[R - 1];
                        if (rl != R - 1) hq = pickRow<0, R - 2>(H, rl);
                        // (stored form on the scale of (rl, j): true value = pattern - sigma - rl ext + c)
                        const int back = sigma + rl * ext - (open - ext);
                        const int qA = (int)(hq & 0xffffu) - back, qB = (int)(hq >> 16) - back;
                        if (region == kLastCell) {
                            if (j == lenA - 1) { runA = qA; if (LOC) colA = j; }
                            if (j == lenB - 1) { runB = qB; if (LOC) colB = j; }
                        } else {
                            // HW scans columns < len, OV columns < len - 1 (its last column is scanned row by row)
                            const int cut = region == kLastRowCol ? 1 : 0;
                            if (j < lenA - cut && qA > runA) { runA = qA; if (LOC) colA = j; }
                            if (j < lenB - cut && qB > runB) { runB = qB; if (LOC) colB = j; }
                        }
                    }
                    if (region == kLastRowCol) {
                        const bool lastA = j == lenA - 1, lastB = j == lenB - 1;
                        if (__builtin_amdgcn_ballot_w64(lastA || lastB) != 0) {
                            // some lane is on its target's last column: maximum over this strip's query rows. The
                            // patterns of one column share a scale, so the packed maximum is a chain of max3 on
                            // both halves at once. (Rows beyond the query exist in the last strip only; their
                            // bound is made opaque HERE so that the row predicates are not hoisted out of the
                            // column loop into scalar register pairs.)
                            int rows = kToBelow ? R : rl + 1;
                            if (!kToBelow) asm volatile("" : "+v"(rows));
                            // (row r is on the scale of column j plus r ext: taken off with a running offset)
                            uint32_t off = 0u;
                            auto rowValue = [&](int r) -> uint32_t {
                                uint32_t v = H[r];
                                if (!kToBelow && r > 0) v = r < rows ? H[r] : H[0] + off;
                                return v - off;
                            };
                            uint32_t m2 = H[0];
#pragma unroll
                            for (int r = 1; r < R; ++r) {
                                off += ext2;
                                asm volatile("" : "+v"(off));
                                m2 = pk_max2_f16(m2, rowValue(r));
                            }
                            const int mA = (int)(m2 & 0xffffu), mB = (int)(m2 >> 16);
                            const int back = sigma - (open - ext);
                            if (lastA) cbA = mA - back;
                            if (lastB) cbB = mB - back;
                            if constexpr (LOC) {
                                // first row that holds the maximum of its half
                                int ia = 0, ib = 0;
#pragma unroll
                                for (int r = R - 1; r >= 0; --r) {
                                    const uint32_t d = rowValue(r) ^ m2;
                                    if ((d & 0xffffu) == 0) ia = r;
                                    if ((d >> 16) == 0) ib = r;
                                    asm volatile("" : "+v"(ia), "+v"(ib), "+v"(off));
                                    off -= ext2;
                                }
                                if (lastA) crowA = s * R + ia;
                                if (lastB) crowB = s * R + ib;
                            }
                        }
                    }
                };
#pragma unroll 1
                for (int cp = 0; cp < 4; cp += 2) {
                    column(std::false_type{}, cp);
                    column(std::true_type{}, cp + 1);
                }
                cur = nxt;
                shift += 4 * ext;
                if (!topGap && shift + 4 * ext > kBiasedMaxShift) {
                    // rebase (every strip does it after the same chunks: the rows the strip above wrote from
                    // the next chunk on are already on the new scale, the one kept in hbPrev is not)
                    const uint32_t d = both(shift);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        H[r] -= d;
                        E[r] -= d;
                    }
                    if constexpr (kFromAbove) hbPrev -= d;
                    sigma -= shift;
                    shift = 0;
                }
                if (kToBelow && (((c + 1) & MIOPAL_STRIP_PUBLISH_MASK) == 0 || c + 1 == nChunks)) {
                    timer.start(2);
                    stripPublish(progOut, c + 1, lane);
                    timer.stop(2);
                }
            }
        };
        if (!dead) {
            pace.begin(lane);
            timer.start(3);
            if (!fromAbove && toBelow) sweep(std::false_type{}, std::true_type{});
            else if (fromAbove && toBelow) sweep(std::true_type{}, std::true_type{});
            else if (fromAbove) sweep(std::true_type{}, std::false_type{});
            // (a single strip is the one-strip kernel's business: launchPairGlobalStrips refuses it)
            timer.stop(3);
            pace.end(lane);
        }
        if (dead) {
            // the strip above never got here: leave the answers to the int32 kernel, tell the strip below
            if (toBelow && lane == 0 && !injected) __hip_atomic_store(progOut, kStripPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.overflow) a.overflow[base + lane] = a.overflow[base + kLanes + lane] = 1;
            continue;
        }
        if constexpr (LOC) {
            auto key = [](int score, bool onLastRow, int index) -> unsigned long long {
                return ((unsigned long long)((uint32_t)score ^ 0x80000000u) << 32) | (onLastRow ? 0x80000000ull : 0ull) |
                       (unsigned long long)(0x7FFFFFFFu - (uint32_t)index);
            };
            if (!toBelow) {
                if (colA >= 0) atomicMax(a.stripKeys + base + lane, key(runA, true, colA));
                if (colB >= 0) atomicMax(a.stripKeys + base + kLanes + lane, key(runB, true, colB));
            }
            if (region == kLastRowCol) {
                if (crowA >= 0) atomicMax(a.stripKeys + base + lane, key(cbA, false, crowA));
                if (crowB >= 0) atomicMax(a.stripKeys + base + kLanes + lane, key(cbB, false, crowB));
            }
        } else if (region == kLastRowCol) {
            // OV: every strip's last-column maximum and the last strip's last-row maximum (INT32_MIN: none)
            atomicMax(a.score + base + lane, max(runA, cbA));
            atomicMax(a.score + base + kLanes + lane, max(runB, cbB));
        } else if (!toBelow) {
            a.score[base + lane] = runA;
            a.score[base + kLanes + lane] = runB;
        }
    }
    timer.flush(a.stripTiming, lane);
}

template <int R, bool LOC>
static hipError_t launchPairGlobalStripsR(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    const size_t lds = PairLayout<R>::bytes(a.nSymbols) + 16 + kPaceInts * sizeof(int);  // table + the unit in flight + pacing
    static uint64_t configured = 0;  // one bit per device
    if (firstUseOnThisDevice(&configured)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&interseq_pair_global_strips_kernel<R, LOC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    const int nBatches = (a.nGroups + a.batchGroups - 1) / a.batchGroups;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(computeUnits, (int64_t)nBatches * a.nStrips));
    hipLaunchKernelGGL((interseq_pair_global_strips_kernel<R, LOC>), dim3(blocks), dim3(kPairWaves * kLanes), lds, stream, a);
    return hipGetLastError();
}

template <int kLo, int kStep, bool LOC>
static hipError_t launchPairGlobalStripsCase(const InterseqArgs& a, int computeUnits, hipStream_t stream) {
    if constexpr (kLo + kStep <= (LOC ? kStripsMaxRowsLoc : kStripsMaxRows))
        return launchPairGlobalStripsR<kLo + kStep, LOC>(a, computeUnits, stream);
    else return hipErrorInvalidValue;
}
template <int kLo, bool LOC>
static hipError_t launchPairGlobalStrips(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    if (a.nStrips < 2 || !a.unitCounter || !a.unitFlags || !a.boundary[0] || !a.boundary[1] || a.batchGroups < 1 ||
        a.batchGroups > kPairWaves || (LOC && !a.stripKeys) || a.region == kAllCells)
        return hipErrorInvalidValue;
    switch (rowsPerStrip - kLo) {
        case 0: return launchPairGlobalStripsCase<kLo, 0, LOC>(a, computeUnits, stream);
        case 2: return launchPairGlobalStripsCase<kLo, 2, LOC>(a, computeUnits, stream);
        case 4: return launchPairGlobalStripsCase<kLo, 4, LOC>(a, computeUnits, stream);
        case 6: return launchPairGlobalStripsCase<kLo, 6, LOC>(a, computeUnits, stream);
        case 8: return launchPairGlobalStripsCase<kLo, 8, LOC>(a, computeUnits, stream);
        case 10: return launchPairGlobalStripsCase<kLo, 10, LOC>(a, computeUnits, stream);
        case 12: return launchPairGlobalStripsCase<kLo, 12, LOC>(a, computeUnits, stream);
        case 14: return launchPairGlobalStripsCase<kLo, 14, LOC>(a, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

// Bytes of LDS the pair table needs for this strip height (host-side sizing).
static inline size_t pairTableBytes(int rowsPerStrip, int nSymbols) {
    return (size_t)nSymbols * nSymbols * (size_t)(((rowsPerStrip + 3) / 4) | 1) * 16;
}

}  // namespace miopal
