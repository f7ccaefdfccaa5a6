// Two pairs per lane: the later passes of a `full` search on packed 16-bit halves (round 5).
// src/pyopal/opal.pxd:17-19 (OPAL_SEARCH_ALIGNMENT); what the reference does with the result:
// src/pyopal/platform/pyx.in:95-99, src/pyopal/lib.pyx:999-1037.
//
// perpair.hip gives every (query window, target window) problem of the traceback a lane of its own and
// computes it at 32 bit: 15 VALU instructions per cell for the direction pass, 8 for the start-cell scan,
// against 3.5 in the search kernels. Here a lane carries TWO unrelated pairs, one in each 16-bit half of its
// registers - neither a shared query window nor a shared length is needed: each half reads its own residues
// from the linear database and its own bytes of the query profile in LDS (one v_perm_b32 per row puts the two
// scores side by side).
//
//   perpair_packed_trace_kernel         directions of the [start..end] rectangles as bit planes (every mode)
//   perpair_packed_scan_kernel          start cells of queries of one strip (SW; HW: the answer in the query's last row only)
//   perpair_packed_scan_strips_kernel   ... of several strips (perpair_packed_strips.inc)
//
// Values are unsigned patterns compared as half floats (interseq_impl.h, "biased integer halves"): between
// 0x0400 and 0x7BFF the order of the bit patterns is the order of the numbers, so v_pk_maximum3_f16 folds two
// max per cell while additions are plain 32-bit adds over both halves. Cells sit on the anti-diagonal scale
// X'' = X + (i + j) ext (ArithU16Diag): both gap kinds extend for free and open with the same h - (open - ext),
// and the stored H is that very number:
//     d = HS(diag) + s''        s'' = S + open + ext (+ bias), an unsigned byte of the profile
//     h = max3(d, E, F)
//     hmo = h - c               c = open - ext; the next column's / row's stored H and the opener of both gaps
//     E = max(E, hmo)   F = max(F, hmo)
// With open >= ext every border value is the constant Z - 2 open on this scale.
//
// Direction flags without a packed compare. The four flags of a cell (came from the diagonal / from E / E was
// opened / F was opened; same comparisons and tie-breaks as perpair_kernel) are all of the form "x - y == 0" or
// "x - y >= c" for 0 <= x - y < 256 (bounded by the scoring scheme, see packedTraceFits). For t = 0x8000 - (x - y)
// per half, bit 15 says "equal" and bits 8..14 all say "different": ONE 32-bit add per flag and cell pair (on
// 0x8000 - h, shared by the four) and ONE v_bfi_b32 that drops the bit of row r into bit 15 - (r mod 8) of an
// accumulator of eight rows - 10 full-rate adds + 4 v_bfi per cell PAIR where the 32-bit kernel needs
// 4 x (v_sub + v_alignbit) per cell. Four accumulators are folded into the dword of 32 rows the walk reads by
// three v_perm_b32.
//
// Model and tie-breaks: oracle/opal_oracle.c (rules 5-7).
#include "common.h"
#include "tuning.h"

#include <type_traits>

namespace miopal {

namespace {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pkMax3(uint32_t a, uint32_t b, uint32_t c) {
    f16x2 r = __builtin_elementwise_maximum(
        __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
        __builtin_bit_cast(f16x2, c));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t pkMax(uint32_t a, uint32_t b) {
    f16x2 r = __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
static __device__ __forceinline__ uint32_t both(int v) { return (uint32_t)v * 0x00010001u; }
// (mask & a) | (~mask & b): one v_bfi_b32 (left to itself hipcc makes v_and + a share of a v_or3 of it)
static __device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
#ifndef MIOPAL_PACKED_NO_BFI
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
#else
    return (a & mask) | (b & ~mask);
#endif
}

// ... with a mask of the lane's own (the select trees of the OV scan)
static __device__ __forceinline__ uint32_t bfiLane(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
// rows[] (a power-of-two-sized view: groups of eight, up to eight groups): the entry of row r of either half, r in
// bits 0-5 / 16-21 of rs2 - a tree of selects over both halves at once, one v_bfi_b32 a node
template <int ROWS>
static __device__ __forceinline__ uint32_t pickRowOfEither(const uint32_t (&rows)[ROWS], uint32_t rs2) {
    static_assert(ROWS % 8 == 0 && ROWS <= 64, "groups of eight rows");
    auto maskOf = [&](int bit) -> uint32_t {
        const uint32_t x = (rs2 >> bit) & 0x00010001u;
        return (x << 16) - x;   // 0xffff in the half whose bit is set
    };
    uint32_t t[32];
    const uint32_t m0 = maskOf(0);
#pragma unroll
    for (int k = 0; k < ROWS / 2; ++k) t[k] = bfiLane(m0, rows[2 * k + 1], rows[2 * k]);
    const uint32_t m1 = maskOf(1);
#pragma unroll
    for (int k = 0; k < ROWS / 4; ++k) t[k] = bfiLane(m1, t[2 * k + 1], t[2 * k]);
    const uint32_t m2 = maskOf(2);
#pragma unroll
    for (int k = 0; k < ROWS / 8; ++k) t[k] = bfiLane(m2, t[2 * k + 1], t[2 * k]);
    constexpr int G = ROWS / 8;   // one entry per group of eight now
    if (G == 1) return t[0];
    // (groups that are not there are never asked for: any entry stands in for them)
    const uint32_t m3 = maskOf(3);
    uint32_t u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = bfiLane(m3, t[2 * k + 1 < G ? 2 * k + 1 : G - 1], t[2 * k < G ? 2 * k : G - 1]);
    if (G <= 2) return u[0];
    const uint32_t m4 = maskOf(4);
    const uint32_t v0 = bfiLane(m4, u[1], u[0]), v1 = bfiLane(m4, u[3], u[2]);
    if (G <= 4) return v0;
    return bfiLane(maskOf(5), v1, v0);
}

// Workgroups of four wavefronts, two per CU (two wavefronts per SIMD), when their LDS fits twice - a workgroup's
// slot is free again when its LAST wavefront is done, and eight sorted wavefronts differ by more than four - else
// of eight.
constexpr int kPkMaxWaves = 8;
constexpr int kPkStageBytes = 16 * 1024;      // per wavefront: four columns x four lines x 64 lanes x 16 bytes
constexpr int kPkZero = kPackedZero;          // pattern of the value 0
constexpr uint32_t kPkC = 0x80008000u;

// the lane's residues, four columns per load (perpair_profile_kernel): unconditional loads at addresses
// clamped into the database, put in place when they are used
struct ResidueStream {
    const uint8_t* tptr;
    const uint8_t* dbLo;
    const uint8_t* dbHi;
    uint32_t padWord;
    int L;
    __device__ __forceinline__ uint32_t fetchRaw(int j0) const {
        const uint8_t* at = tptr + j0;
        at = at < dbLo ? dbLo : at;
        at = at > dbHi ? dbHi : at;
        uint32_t w;
        __builtin_memcpy(&w, at, 4);
        return w;
    }
    __device__ __forceinline__ uint32_t inPlace(uint32_t raw, int j0) const {
        const uint8_t* at = tptr + j0;
        const int64_t below = dbLo - at, above = at - dbHi;   // > 0: the clamp moved the load by this many bytes
        uint32_t w = raw;
        if (below > 0) w = below >= 4 ? 0u : raw << (8 * (int)below);
        if (above > 0) w = above >= 4 ? 0u : raw >> (8 * (int)above);
        const int valid = L - j0;   // columns of the four that exist
        const uint32_t keep = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : (1u << (8 * valid)) - 1u;
        return (w & keep) | (padWord & ~keep);
    }
};

// four accumulators of eight rows each (bits 15..8 = pair A's rows, 31..24 = pair B's) -> the dwords of 32 rows
// of either pair, row 0 in bit 31
static __device__ __forceinline__ void foldPlanes(const uint32_t acc[4], uint32_t& outA, uint32_t& outB) {
    const uint32_t t01 = __builtin_amdgcn_perm(acc[0], acc[1], 0x07030501u);
    const uint32_t t23 = __builtin_amdgcn_perm(acc[2], acc[3], 0x07030501u);
    outA = __builtin_amdgcn_perm(t01, t23, 0x05040100u);
    outB = __builtin_amdgcn_perm(t01, t23, 0x07060302u);
}

// MULTI: queries of several strips (the last row of a strip travels to the next through a.boundary); a one-strip
// query's kernel holds neither those rows nor the registers they are fetched into two columns ahead.
template <bool BIASED, int kPkWaves, bool MULTI>
__global__ __launch_bounds__(kPkWaves * kLanes, 2) void perpair_packed_trace_kernel(PerPairArgs a) {
    constexpr int kPkBlock = kPkWaves * kLanes;
    extern __shared__ __attribute__((aligned(16))) uint8_t pkLds[];
    uint4* const stageAll = reinterpret_cast<uint4*>(pkLds);
    uint8_t* const prof = pkLds + kPkWaves * kPkStageBytes;
    const int A = a.alphabet;
    const int Qtot = a.queryLength;
    const int pstride = a.profileStride;   // bytes per residue row, a multiple of 4, >= Qtot + 64 + 8
    const int open = a.gapOpen, ext = a.gapExt, c = open - ext;
    for (int idx = threadIdx.x; idx < (A + 1) * pstride; idx += kPkBlock) {
        const int t = idx / pstride, y = idx - t * pstride;
        int v = 0;   // padding symbol and rows: the lowest score
        if (t < A && y < Qtot) v = a.matrix[(int)a.query[y] * A + t] + open + ext + a.packedBias;
        prof[idx] = (uint8_t)v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = blockIdx.x * kPkWaves + wave;
    const int idxA = W * 128 + lane, idxB = idxA + kLanes;
    // (hybrid direction pass: the leading wavefronts' jobs - of 64 - are outliers done by intraseq_kernel)
    // (one launch over several batches' lists: a count of leading wavefronts per batch, results relative to the batch)
    int outBaseA = 0, outBaseB = 0;
    bool activeA = idxA < a.nJobs, activeB = idxB < a.nJobs;
    {
        int batchA = 0, batchB = 0, inA = idxA >> 6, inB = idxB >> 6;   // wave-uniform all of them
        if (a.outBatch > 0) {
            batchA = idxA / a.outBatch;
            batchB = idxB / a.outBatch;
            outBaseA = batchA * a.outBatch;
            outBaseB = batchB * a.outBatch;
            inA -= outBaseA >> 6;
            inB -= outBaseB >> 6;
        }
        if (a.skipWaves != nullptr) {
            if (activeA && inA < a.skipWaves[batchA]) activeA = false;
            if (activeB && inB < a.skipWaves[batchB]) activeB = false;
        }
    }
    if (__builtin_amdgcn_ballot_w64(activeA || activeB) == 0) return;
    PairJob jobA{}, jobB{};
    if (activeA) jobA = a.jobs[idxA];
    if (activeB) jobB = a.jobs[idxB];
    const int QA = jobA.qLen, LA = jobA.tLen, QB = jobB.qLen, LB = jobB.tLen;

    int maxQ = max(QA, QB), maxL = max(LA, LB);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        maxQ = max(maxQ, __shfl_xor(maxQ, off));
        maxL = max(maxL, __shfl_xor(maxL, off));
    }
    maxQ = __builtin_amdgcn_readfirstlane(maxQ);
    maxL = __builtin_amdgcn_readfirstlane(maxL);
    const int nStrips = MULTI ? (maxQ + kLanes - 1) / kLanes : 1;   // (launchPerPairPackedTrace: one strip unless MULTI)

    const uint32_t c2 = both(c), kc2 = kPkC + both(c - 1);
    const uint32_t negBias2 = 0u - both(a.packedBias);
    const uint32_t top2 = both(kPkZero - 2 * open);       // every border cell in stored form (open >= ext)
    int bestA = INT32_MIN, bestB = INT32_MIN;

    ResidueStream rsA{a.residues + jobA.tOff, a.residues, a.residues + a.residueCount - 4, (uint32_t)A * 0x01010101u, LA};
    ResidueStream rsB{a.residues + jobB.tOff, a.residues, a.residues + a.residueCount - 4, (uint32_t)A * 0x01010101u, LB};
    // strip boundaries of the wavefront: (stored H, F before it met that H) of the strip's last row, per column
    uint2* bnd = MULTI && a.boundary ? reinterpret_cast<uint2*>(a.boundary) + (int64_t)W * a.boundaryStride * kLanes + lane : nullptr;
    // lines of the two pairs: [pair / 64][strip][column / 4][rows 0-31 | 32-63][pair % 64][column % 4][plane]
    uint8_t* const waveDirsA = a.dirs + (int64_t)(W * 2) * a.dirWaveStride;        // (wave-uniform: the lines of pair 0 of either half)
    uint8_t* const waveDirsB = a.dirs + (int64_t)(W * 2 + 1) * a.dirWaveStride;
    uint4* const stageWave = stageAll + wave * (kPkStageBytes / 16);      // [column % 4][line][pair of the half]
    uint4* const stage = stageWave + lane;
    // pairs of either half whose lines are written (none of a half left to intraseq_kernel, or beyond the list)
    const int liveA = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64(activeA)));
    const int liveB = __builtin_amdgcn_readfirstlane(__builtin_popcountll(__builtin_amdgcn_ballot_w64(activeB)));

    // deposit masks: row r of a group of eight -> bit 15 - r of either half
    // (wave-uniform constants: the compiler keeps them in SGPRs, one per v_bfi)
    auto K = [](int r) -> uint32_t { return (0x8000u >> (r & 7)) * 0x00010001u; };

    for (int s = 0; s < nStrips; ++s) {
        const int row0 = s * kLanes;
        const int rowsHere = min(maxQ - row0, kLanes);  // wave-uniform
        // One strip of GROUPS x 8 rows. GROUPS is a compile-time constant (the switch below): with a run-time
        // row count the unrolled rows need a way out every eight rows, and what is live where the ways out meet
        // cost the kernel a hundred spilled registers.
        auto sweepStrip = [&](auto groupsTag) {
            constexpr int GROUPS = decltype(groupsTag)::value, ROWS = GROUPS * 8;
            const bool toNext = MULTI && s + 1 < nStrips;
            // the halves' first rows along the profile; rows beyond the query read the pad bytes behind it
            const int ysA = min(jobA.qOff + row0, Qtot), ysB = min(jobB.qOff + row0, Qtot);
            const uint32_t shiftA = (uint32_t)ysA & 3u, shiftB = (uint32_t)ysB & 3u;
            const int yAlA = ysA & ~3, yAlB = ysB & ~3;
            uint32_t HS[ROWS], E[ROWS];
#pragma unroll
            for (int i = 0; i < ROWS; ++i) HS[i] = E[i] = top2;   // column -1
            uint8_t* const waveStripA = waveDirsA + (int64_t)s * a.dirStripColumns * (kLanes / 2 * kLanes);
            uint8_t* const waveStripB = waveDirsB + (int64_t)s * a.dirStripColumns * (kLanes / 2 * kLanes);
            // the cell above-left of the strip's first cell: the origin (value 0 two steps up the scale) or a border cell
            uint32_t aboveHsPrev = s == 0 ? both(kPkZero - 2 * ext - c) : top2;
            const int lastLocalA = QA - 1 - row0, lastLocalB = QB - 1 - row0;

            // the halves' residues, four columns per load; the column's own pair is picked a column ahead so that
            // the first profile dwords of column j + 1 can be fetched while column j is computed
            uint32_t wcurA = rsA.inPlace(rsA.fetchRaw(0), 0), rawNextA = rsA.fetchRaw(4);
            uint32_t wcurB = rsB.inPlace(rsB.fetchRaw(0), 0), rawNextB = rsB.fetchRaw(4);
            uint2 aboveNext = make_uint2(top2, top2), aboveNext2 = aboveNext;
            // (two columns ahead: one ahead the direction kernel of Q = 300 took 7.06 ms per launch, two 6.56)
            if (MULTI && s > 0) {
                aboveNext = bnd[0];
                aboveNext2 = bnd[(int64_t)min(1, maxL - 1) * kLanes];
            }
            const int sweep = (maxL + 3) & ~3;
            // "E was opened" of column j is known when column j - 1 is done: the planes of the column before wait
            // here (column 0 opens from the border)
            uint32_t prevO[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // [A rows 0-31, A 32-63, B 0-31, B 32-63]
            auto rowOf = [&](uint32_t t, int yAl) { return reinterpret_cast<const uint32_t*>(prof + t * pstride + yAl); };
            const uint32_t* prowA = rowOf(wcurA & 0xffu, yAlA);
            const uint32_t* prowB = rowOf(wcurB & 0xffu, yAlB);
            // An LDS read queues behind the other wavefronts' reads: the dwords of a column are fetched three blocks of
            // four rows ahead of their use, the first three while the column before is still being computed
            // (with a read per block, used at once: SQ_WAIT_ANY 39 % of the wavefronts' cycles)
            uint32_t pA0 = prowA[0], pA1 = prowA[1], pA2 = prowA[2];
            uint32_t pB0 = prowB[0], pB1 = prowB[1], pB2 = prowB[2];
            for (int j = 0; j < sweep; ++j) {
                // the residues of the next column
                uint32_t tnA, tnB;
                if ((j & 3) == 3) {
                    wcurA = rsA.inPlace(rawNextA, j + 1);
                    rawNextA = rsA.fetchRaw(j + 5);
                    wcurB = rsB.inPlace(rawNextB, j + 1);
                    rawNextB = rsB.fetchRaw(j + 5);
                    tnA = wcurA & 0xffu;
                    tnB = wcurB & 0xffu;
                } else {
                    tnA = (wcurA >> (8 * ((j & 3) + 1))) & 0xffu;
                    tnB = (wcurB >> (8 * ((j & 3) + 1))) & 0xffu;
                }
                const uint32_t* const prowNextA = rowOf(tnA, yAlA);
                const uint32_t* const prowNextB = rowOf(tnB, yAlB);
                uint32_t hsUp = top2, fOldUp = top2;
                if (MULTI && s > 0) {
                    const uint2 above = aboveNext;
                    aboveNext = aboveNext2;
                    aboveNext2 = bnd[(int64_t)min(j + 2, maxL - 1) * kLanes];   // (unconditional: no copy behind the load)
                    hsUp = above.x;
                    fOldUp = above.y;
                }
                uint32_t hsDiag = aboveHsPrev;
                aboveHsPrev = hsUp;
                // F of the first row, and whether it was opened: not opened <=> fOld > hsUp <=> bit 15 of fOld + 0x7fff - hsUp
                uint32_t F = pkMax(fOldUp, hsUp);
                uint32_t fOldLast = fOldUp;
                uint32_t accD[4], accE[4], accO[4], accF[5];   // of the half strip at hand
                accF[0] = (fOldUp + (kPkC - 0x00010001u) - hsUp) & K(0);
                uint32_t wloA = pA0, wloB = pB0, n1A = pA1, n1B = pB1, n2A = pA2, n2B = pB2, fourA = 0, fourB = 0;
                constexpr int kLastDword = ROWS / 4;                       // dwords 0 .. ROWS / 4 of the rows are read
                constexpr int kNextAt = ROWS >= 12 ? ROWS / 4 - 3 : 0;     // block at which the next column's are
#pragma unroll
                for (int i = 0; i < ROWS; ++i) {
                    if ((i & 3) == 0) {
                        const int blk = i >> 2;
                        const uint32_t whiA = n1A, whiB = n1B;
                        fourA = __builtin_amdgcn_alignbyte(whiA, wloA, shiftA);
                        fourB = __builtin_amdgcn_alignbyte(whiB, wloB, shiftB);
                        wloA = whiA;
                        wloB = whiB;
                        n1A = n2A;
                        n1B = n2B;
                        if (blk + 3 <= kLastDword) {
                            n2A = prowA[blk + 3];
                            n2B = prowB[blk + 3];
                        }
                        if (blk == kNextAt) {
                            pA0 = prowNextA[0]; pA1 = prowNextA[1]; pA2 = prowNextA[2];
                            pB0 = prowNextB[0]; pB1 = prowNextB[1]; pB2 = prowNextB[2];
                        }
                    }
                    // {0, B's byte, 0, A's byte}
                    const uint32_t ub2 = __builtin_amdgcn_perm(fourB, fourA, 0x0c040c00u + (uint32_t)(i & 3) * 0x00010001u);
                    uint32_t d = hsDiag + ub2;
                    if (BIASED) d += negBias2;
                    const uint32_t eOld = E[i];
                    const uint32_t h = pkMax3(d, eOld, F);
                    const uint32_t hC = kPkC - h;           // 0x8000 - h per half
                    const uint32_t hC2 = kc2 - h;           // 0x8000 + c - 1 - h
                    const int g = (i >> 3) & 3;             // group of eight rows inside the half strip
                    if ((i & 7) == 0) {
                        accD[g] = (d + hC) & K(i);
                        accE[g] = (eOld + hC) & K(i);
                        accO[g] = (eOld + hC2) & K(i);
                    } else {
                        accD[g] = bfi(K(i), d + hC, accD[g]);
                        accE[g] = bfi(K(i), eOld + hC, accE[g]);
                        accO[g] = bfi(K(i), eOld + hC2, accO[g]);
                    }
                    // "F was opened" belongs to the row below
                    if ((i & 7) == 7) accF[g + 1] = (F + hC2) & K(i + 1);
                    else accF[g] = bfi(K(i + 1), F + hC2, accF[g]);
                    const uint32_t hmo = h - c2;
                    E[i] = pkMax(eOld, hmo);
                    fOldLast = F;
                    F = pkMax(F, hmo);
                    hsDiag = HS[i];
                    HS[i] = hmo;
                    // pins the schedule: hipcc would otherwise hoist every ds_read of the column to its top
                    if ((i & 3) == 3) asm volatile("" : "+v"(F), "+v"(hsDiag)::"memory");
                    if ((i & 31) == 31 || i == ROWS - 1) {
                        // A half strip is complete: fold its planes and put the column's 16 bytes of either pair into
                        // the wavefront's staging lines (bits of rows the half strip does not hold are never read).
                        // Polarity: "equal" planes are true in bit 15 of their group and inverted below it, "opened"
                        // planes the other way round.
                        const int hs = i >> 5, held = ((i & 31) >> 3) + 1;
                        uint32_t aD[4], aE[4], aO[4], aF[4];
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            aD[x] = x < held ? accD[x] : 0u;
                            aE[x] = x < held ? accE[x] : 0u;
                            aO[x] = x < held ? accO[x] : 0u;
                            aF[x] = x < held ? accF[x] : 0u;
                        }
                        uint32_t dA, dB, eA, eB, oA, oB, fA, fB;
                        foldPlanes(aD, dA, dB);
                        foldPlanes(aE, eA, eB);
                        foldPlanes(aO, oA, oB);
                        foldPlanes(aF, fA, fB);
                        stage[((j & 3) * 4 + hs) * kLanes] = make_uint4(dA ^ 0x7f7f7f7fu, eA ^ 0x7f7f7f7fu, prevO[hs], fA ^ 0x80808080u);
                        stage[((j & 3) * 4 + 2 + hs) * kLanes] = make_uint4(dB ^ 0x7f7f7f7fu, eB ^ 0x7f7f7f7fu, prevO[2 + hs], fB ^ 0x80808080u);
                        prevO[hs] = oA ^ 0x80808080u;
                        prevO[2 + hs] = oB ^ 0x80808080u;
                        accF[0] = accF[4];   // (row 32's flag came with row 31)
                    }
                }
                prowA = prowNextA;
                prowB = prowNextB;
                if (ROWS == kLanes && toNext && j < maxL) bnd[(int64_t)j * kLanes] = make_uint2(HS[ROWS - 1], fOldLast);
                if ((j & 3) == 3) {
                    // Block of four columns: the staged lines leave for HBM. Every store instruction writes 1 KB in one
                    // piece - lane l the 16 bytes of column l % 4 of pair 16 k + l / 4 - i.e. whole 128-byte lines. (A lane
                    // writing its own line, 16 bytes an instruction, touched 64 lines per instruction, a quarter of a
                    // half each: the L2 fetched every line it was handed a piece of - FETCH_SIZE as large as WRITE_SIZE.)
                    const int64_t blockOff = (int64_t)(j >> 2) * (2 * kLanes * 64);
                    const int x = lane & 3, sub = lane >> 2;
#pragma unroll
                    for (int hs = 0; hs < (ROWS > 32 ? 2 : 1); ++hs) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int pr = 16 * k + sub;   // pair of its half of the wavefront
                            const uint4 va = stageWave[(x * 4 + hs) * kLanes + pr];
                            const uint4 vb = stageWave[(x * 4 + 2 + hs) * kLanes + pr];
                            if (pr < liveA) *reinterpret_cast<uint4*>(waveStripA + blockOff + hs * (kLanes * 64) + pr * 64 + x * 16) = va;
                            if (pr < liveB) *reinterpret_cast<uint4*>(waveStripB + blockOff + hs * (kLanes * 64) + pr * 64 + x * 16) = vb;
                        }
                        asm volatile("" ::: "memory");
                    }
                }
                // score of a window = its last cell; lanes of a wavefront (sorted by length) finish in a handful of
                // columns, so the row select runs rarely
                const bool mineA = j == LA - 1 && lastLocalA >= 0 && lastLocalA < ROWS;
                const bool mineB = j == LB - 1 && lastLocalB >= 0 && lastLocalB < ROWS;
                if (__builtin_amdgcn_ballot_w64(mineA || mineB) != 0) {
                    uint32_t vA = 0, vB = 0;
                    // (opaque here: the comparisons would otherwise be hoisted out of the column loop as lane masks
                    // kept - and spilled - across it)
                    int llA = lastLocalA, llB = lastLocalB;
                    asm volatile("" : "+v"(llA), "+v"(llB));
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) {
                        vA = (i == llA) ? HS[i] : vA;
                        vB = (i == llB) ? HS[i] : vB;
                    }
                    if (mineA) bestA = (int)(vA & 0xffffu) - kPkZero + c - (QA - 1 + j) * ext;
                    if (mineB) bestB = (int)(vB >> 16) - kPkZero + c - (QB - 1 + j) * ext;
                }
            }
        };
        using std::integral_constant;
        switch ((rowsHere + 7) >> 3) {   // wave-uniform
            case 1: sweepStrip(integral_constant<int, 1>{}); break;
            case 2: sweepStrip(integral_constant<int, 2>{}); break;
            case 3: sweepStrip(integral_constant<int, 3>{}); break;
            case 4: sweepStrip(integral_constant<int, 4>{}); break;
            case 5: sweepStrip(integral_constant<int, 5>{}); break;
            case 6: sweepStrip(integral_constant<int, 6>{}); break;
            case 7: sweepStrip(integral_constant<int, 7>{}); break;
            default: sweepStrip(integral_constant<int, 8>{}); break;
        }
    }

    auto finish = [&](bool active, const PairJob& job, int best, int outBase) {
        if (!active) return;
        if (!(job.qLen > 0 && job.tLen > 0)) {
            // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
            best = 0;
            if (job.qLen > 0) best = borderGap(job.qLen - 1, open, ext);
            if (job.tLen > 0) best = borderGap(job.tLen - 1, open, ext);
        }
        a.score[outBase + job.out] = best;
    };
    finish(activeA, jobA, bestA, outBaseA);
    finish(activeB, jobB, bestB, outBaseB);
}


// ---- the start-cell scan with two pairs per lane -------------------------------------------------------
// The reversed-prefix scans (perpair.hip: perpair_scan_refill_kernel, perpair_profile_kernel<kAllCells>) on the
// same halves, for queries of ONE strip (several: perpair_packed_strips.inc, the lanes in step). Persistent
// wavefronts; every HALF of a lane runs its own schedule: its own pair, its own column - a half that is done with
// its pair takes the next pair of the list (one atomic per wavefront and refill), at multiples of four columns so
// that the four-residue loads stay in step. What this costs: 1M pairs on 3072 x 128 half-slots are 2.5 pairs per
// slot, and when the list runs out every wavefront finishes its own stragglers - the halves are busy 70 % of a
// wavefront's life (tools/r05_scan_schedule_sim.py; no order known beforehand helps: the window's length is what
// the scan finds out).
//
// Cells on the column scale X' = X + j ext, everything times 8: the three low bits of a value are free and the
// column's maximum is folded over KEYS, value + 7 - (group of eight rows), so that the maximum itself says which
// group holds its first row - the known optimum of the forward pass is met once per pair and strip, but with 128
// pairs per wavefront that is nearly every column, and finding the row by comparison cost a third of a column:
//     d = HS(diag) + s'        s' = 8 (S + open + bias), an unsigned byte of the profile (S + open + bias <= 31)
//     e = max(E, HS)           (extending is free on this scale)
//     f = max(F, HS above) - 8 ext
//     h = max3(d, e, f);  HS = h - 8 (open - ext)
// No cell of an anchored scan exceeds the optimum, so "the column's maximum equals the optimum" is the hit, and the
// first column-major cell that holds it is the start cell (oracle/opal_oracle.c, rule 6).
constexpr int kScanBlock = 256;
constexpr int kScanWaves = kScanBlock / kLanes;

struct ScanHalf {
    int Q, need, out, j, y0, strip, bcol, brow;
    uint32_t wcur, rawNext;
    const uint8_t* tptr;
};

// OV (the answer in the pair's OWN last row - the reversed prefix ends where the forward pass ended - or anywhere in its
// last column): an instantiation of its own, its extra work (a select tree per column) is nobody else's.
// REGION: 0 = any cell answers (SW), 1 = the query's last row (HW), 2 = OV; Smith-Waterman's instantiation holds none of the
// others' registers (167 of them: three wavefronts per SIMD; with the last row's cell kept as well, 175 and two).
template <int GROUPS, bool BIASED, int REGION>
__global__ __launch_bounds__(kScanBlock, (GROUPS <= 7 && REGION == 0) ? 3 : 2) void perpair_packed_scan_kernel(PerPairArgs a) {
    constexpr bool OV = REGION == 2;
    constexpr int ROWS = GROUPS * 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t pkLds[];
    uint8_t* const prof = pkLds;
    const int A = a.alphabet;
    const int Qtot = a.queryLength;
    const int pstride = a.profileStride;
    const int open = a.gapOpen, ext = a.gapExt;
    const int Z = a.packedZero;   // pattern of the value 0 on column 0: a multiple of 8 that leaves room for the last row's border
    for (int idx = threadIdx.x; idx < (A + 1) * pstride; idx += kScanBlock) {
        const int t = idx / pstride, y = idx - t * pstride;
        int v = 0;
        if (t < A && y < Qtot) v = 8 * (a.matrix[(int)a.query[Qtot - 1 - y] * A + t] + open + a.packedBias);
        prof[idx] = (uint8_t)v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t ext2 = both(8 * ext), c2 = both(8 * (open - ext));
    const uint32_t negBias2 = 0u - both(8 * a.packedBias);
    const int topPat = Z - 8 * (2 * open - ext);       // row -1 in stored form, any column but -1
    const uint32_t top2 = both(topPat);
    const uint32_t padWord = (uint32_t)A * 0x01010101u;
    const uint8_t* const dbLo = a.residues;
    const uint8_t* const dbHi = a.residues + a.residueCount - 4;

    ScanHalf hA{}, hB{};
    bool busyA = false, busyB = false;
    hA.tptr = hB.tptr = a.residues;
    hA.wcur = hB.wcur = hA.rawNext = hB.rawNext = padWord;
    hA.bcol = hB.bcol = hA.brow = hB.brow = -1;
    uint32_t tgt2 = 0xffffffffu;      // pattern of the optimum on the halves' current columns (idle: never met)
    uint32_t incr2 = 0;               // what a column adds to it (busy halves)
    uint32_t aboveHsPrev = top2;
    uint32_t yAl2 = 0, shiftA = 0, shiftB = 0;   // (yAl2: the halves' aligned profile offsets, 16 bits each)
    uint32_t HS[ROWS], E[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) HS[i] = E[i] = top2;
    bool exhausted = false;   // wave-uniform: no pair left to take
    constexpr bool lastRowOnly = REGION == 1;
    const int lastRowLocal = lastRowOnly ? Qtot - 1 : 0;      // one strip: the query's last row (every HW prefix has them all)
    uint32_t rs2 = 0;                                         // OV: the halves' own last rows, 16 bits each

    auto fetchRaw = [&](const ScanHalf& h, int j0) -> uint32_t {
        const uint8_t* at = h.tptr - j0 - 3;
        at = at < dbLo ? dbLo : at;
        at = at > dbHi ? dbHi : at;
        uint32_t w;
        __builtin_memcpy(&w, at, 4);
        return w;
    };
    auto inPlace = [&](const ScanHalf& h, int L, uint32_t raw, int j0) -> uint32_t {
        const uint8_t* at = h.tptr - j0 - 3;
        const int64_t below = dbLo - at, above = at - dbHi;
        uint32_t w = raw;
        if (below > 0) w = below >= 4 ? 0u : raw << (8 * (int)below);
        if (above > 0) w = above >= 4 ? 0u : raw >> (8 * (int)above);
        w = __builtin_bswap32(w);
        const int valid = L - j0;
        const uint32_t keep = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : (1u << (8 * valid)) - 1u;
        return (w & keep) | (padWord & ~keep);
    };
    // (the pair's target length: the prefix ends where the job says; `need` shrinks, the length does not)
    int LA = 0, LB = 0;
    int stopA = 0, stopB = 0;

    int windowMax = 0, rowsMax = 0;   // (with start cells: the longest target window and tallest query window seen)

    for (int w = 0;; ++w) {
        if ((w & 3) == 0) {
            // ---- service: halves that go on to their pair's next strip, halves that take a new pair
            const bool wantA = !busyA, wantB = !busyB;
            const uint64_t maskA = __builtin_amdgcn_ballot_w64(wantA), maskB = __builtin_amdgcn_ballot_w64(wantB);
            const int idle = __builtin_popcountll(maskA) + __builtin_popcountll(maskB);
            if (exhausted && idle == 2 * kLanes) break;
            const bool refill = !exhausted && idle > 0 && (idle >= a.refillLanes || w == 0);
            if (refill) {
                bool startA = false, startB = false;
                {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(a.jobCounter, idle);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base + idle >= a.nJobs) exhausted = true;
                    const int rankA = __builtin_amdgcn_mbcnt_hi((uint32_t)(maskA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)maskA, 0));
                    const int rankB = __builtin_popcountll(maskA) +
                                      __builtin_amdgcn_mbcnt_hi((uint32_t)(maskB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)maskB, 0));
                    auto take = [&](bool want, int k, ScanHalf& h, int& L, int& stop, bool& start) {
                        if (!want || k >= a.nJobs) return;
                        PairJob job{};
                        if (a.jobs != nullptr) {
                            job = a.jobs[k];
                        } else {
                            // the reversed prefixes anchored on the end cell, straight from the end pass's arrays
                            const int src = a.order ? a.order[k] : k;
                            const int qe = a.fwdEndQ[src], te = a.fwdEndT[src];
                            job.out = src;
                            job.stop = a.fwdScore[src];
                            if (qe >= 0 && te >= 0) {
                                job.tOff = a.fwdOffsets[src] + te;
                                job.tLen = te + 1;
                                job.qOff = qe;
                                job.qLen = qe + 1;
                            }
                        }
                        h.Q = job.qLen;
                        L = job.tLen;
                        h.need = L;
                        h.out = job.out;
                        stop = job.stop;
                        h.tptr = a.residues + job.tOff;
                        h.y0 = Qtot - 1 - job.qOff;
                        h.strip = -1;
                        h.bcol = h.brow = -1;
                        if (job.qLen > 0 && job.tLen > 0) {
                            start = true;
                        } else if (a.startQ != nullptr) {
                            a.startQ[job.out] = -1;   // (no end cell: no alignment)
                            a.startT[job.out] = -1;
                        } else {
                            // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
                            int v = 0;
                            if (job.qLen > 0) v = borderGap(job.qLen - 1, open, ext);
                            if (job.tLen > 0) v = borderGap(job.tLen - 1, open, ext);
                            a.score[job.out] = v;
                            if (a.endI) a.endI[job.out] = -1;
                            if (a.endJ) a.endJ[job.out] = -1;
                        }
                    };
                    take(wantA, base + rankA, hA, LA, stopA, startA);
                    take(wantB, base + rankB, hB, LB, stopB, startB);
                }
                if (__builtin_amdgcn_ballot_w64(startA || startB) != 0) {
                    // a half starts a strip: column 0 of it, its rows on the border, its own optimum to meet
                    auto begin = [&](bool start, ScanHalf& h) {
                        if (!start) return;
                        h.strip += 1;
                        h.j = 0;
                        h.rawNext = fetchRaw(h, 0);
                    };
                    begin(startA, hA);
                    begin(startB, hB);
                    const uint32_t m2 = (startA ? 0x0000ffffu : 0u) | (startB ? 0xffff0000u : 0u);
                    const int rowA0 = hA.strip * kLanes, rowB0 = hB.strip * kLanes;
                    // rows: H[i][-1] = -(open + i ext) one column to the left of column 0: stored form Z - 8 (2 open + i ext)
                    const uint32_t baseRows = ((uint32_t)(Z - 8 * (2 * open + rowA0 * ext)) & 0xffffu) |
                                              ((uint32_t)(Z - 8 * (2 * open + rowB0 * ext)) << 16);
#pragma unroll
                    for (int i = 0; i < ROWS; ++i) {
                        const uint32_t v = baseRows - both(8 * i * ext);
                        HS[i] = (v & m2) | (HS[i] & ~m2);
                        E[i] = (v & m2) | (E[i] & ~m2);
                    }
                    // the cell above-left of the strip: the origin (0 on column -1's scale) or a border cell
                    const int diagA = rowA0 == 0 ? Z - 8 * open : Z - 8 * (2 * open + (rowA0 - 1) * ext);
                    const int diagB = rowB0 == 0 ? Z - 8 * open : Z - 8 * (2 * open + (rowB0 - 1) * ext);
                    const uint32_t diag2 = ((uint32_t)diagA & 0xffffu) | ((uint32_t)diagB << 16);
                    aboveHsPrev = (diag2 & m2) | (aboveHsPrev & ~m2);
                    const uint32_t t2 = ((uint32_t)(Z + 8 * stopA) & 0xffffu) | ((uint32_t)(Z + 8 * stopB) << 16);
                    tgt2 = (t2 & m2) | (tgt2 & ~m2);
                    const int ysA = min(hA.y0 + rowA0, Qtot), ysB = min(hB.y0 + rowB0, Qtot);
                    if (startA) {
                        shiftA = (uint32_t)ysA & 3u;
                        yAl2 = (yAl2 & 0xffff0000u) | (uint32_t)(ysA & ~3);
                        if (OV) rs2 = (rs2 & 0xffff0000u) | (uint32_t)(hA.Q - 1);
                        busyA = true;
                    }
                    if (startB) {
                        shiftB = (uint32_t)ysB & 3u;
                        yAl2 = (yAl2 & 0x0000ffffu) | ((uint32_t)(ysB & ~3) << 16);
                        if (OV) rs2 = (rs2 & 0x0000ffffu) | ((uint32_t)(hB.Q - 1) << 16);
                        busyB = true;
                    }
                    incr2 = (busyA ? (uint32_t)(8 * ext) : 0u) | (busyB ? (uint32_t)(8 * ext) << 16 : 0u);
                }
                if (exhausted && __builtin_amdgcn_ballot_w64(busyA || busyB) == 0) break;
            }
            hA.wcur = inPlace(hA, LA, hA.rawNext, hA.j);
            hA.rawNext = fetchRaw(hA, hA.j + 4);
            hB.wcur = inPlace(hB, LB, hB.rawNext, hB.j);
            hB.rawNext = fetchRaw(hB, hB.j + 4);
        }
        const uint32_t tA = (hA.wcur >> (8 * (w & 3))) & 0xffu, tB = (hB.wcur >> (8 * (w & 3))) & 0xffu;
        const uint32_t* prowA = reinterpret_cast<const uint32_t*>(prof + tA * pstride + (yAl2 & 0xffffu));
        const uint32_t* prowB = reinterpret_cast<const uint32_t*>(prof + tB * pstride + (yAl2 >> 16));
        uint32_t hsUp = top2, F = top2;
        uint32_t hsDiag = aboveHsPrev;
        aboveHsPrev = hsUp;
        uint32_t gm[GROUPS];
        uint32_t held = 0;
        uint32_t hLast = 0;   // (scanLastRow: the cell of the query's last row - in the strip's last group of eight)
        uint32_t wloA = prowA[0], wloB = prowB[0], n1A = prowA[1], n1B = prowB[1], fourA = 0, fourB = 0;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if ((i & 3) == 0) {
                const int blk = i >> 2;
                const uint32_t whiA = n1A, whiB = n1B;
                fourA = __builtin_amdgcn_alignbyte(whiA, wloA, shiftA);
                fourB = __builtin_amdgcn_alignbyte(whiB, wloB, shiftB);
                wloA = whiA;
                wloB = whiB;
                if (blk + 2 <= ROWS / 4) {
                    n1A = prowA[blk + 2];
                    n1B = prowB[blk + 2];
                }
            }
            const uint32_t ub2 = __builtin_amdgcn_perm(fourB, fourA, 0x0c040c00u + (uint32_t)(i & 3) * 0x00010001u);
            uint32_t d = hsDiag + ub2;
            if (BIASED) d += negBias2;
            const uint32_t e = pkMax(E[i], HS[i]);
            const uint32_t f = pkMax(F, hsUp) - ext2;
            const uint32_t h = pkMax3(d, e, f);
            if (lastRowOnly && i >= ROWS - 8) hLast = i == lastRowLocal ? h : hLast;   // (wave-uniform)
            // the group's maximum, two rows a step
            const int r = i & 7, g = i >> 3;
            if (r == 1) gm[g] = pkMax(held, h);
            else if (r & 1) gm[g] = pkMax3(gm[g], held, h);
            else held = h;
            const uint32_t hs = h - c2;
            hsDiag = HS[i];
            HS[i] = hs;
            E[i] = e;
            F = f;
            hsUp = hs;
            if ((i & 3) == 3) asm volatile("" : "+v"(F), "+v"(hsDiag)::"memory");
        }
        // the column's maximum over keys: value + 7 - group, i.e. the first group that holds it
        uint32_t cmk = gm[0] + both(7);
#pragma unroll
        for (int g = 1; g < GROUPS; g += 2) {
            if (g + 1 < GROUPS) cmk = pkMax3(cmk, gm[g] + both(7 - g), gm[g + 1] + both(6 - g));
            else cmk = pkMax(cmk, gm[g] + both(7 - g));
        }
        // (HW, scanLastRow: only the cell of the query's last row answers, and cells above it may well be larger)
        uint32_t x = lastRowOnly ? hLast ^ tgt2 : (cmk & 0xfff8fff8u) ^ tgt2;
        // OV: a half's last column answers in every row (no cell of the answer region exceeds the optimum, the rows beyond
        // the pair's query are smaller still: the test above); every other column only in the pair's own last row
        const bool lastColA = hA.j + 1 >= hA.need, lastColB = hB.j + 1 >= hB.need;
        if (OV) {
            const uint32_t own = pickRowOfEither<ROWS>(HS, rs2) ^ (tgt2 - c2);   // (HS holds h - c2)
            x = (lastColA ? x & 0xffffu : own & 0xffffu) | (lastColB ? x & 0xffff0000u : own & 0xffff0000u);
        }
        const bool hitA = busyA && (x & 0xffffu) == 0, hitB = busyB && (x >> 16) == 0;
        if (__builtin_amdgcn_ballot_w64(hitA || hitB) != 0) {
            const int gstarA = 7 - (int)(cmk & 7u), gstarB = 7 - (int)((cmk >> 16) & 7u);
            int rowA = OV ? (int)(rs2 & 0xffffu) : lastRowLocal, rowB = OV ? (int)(rs2 >> 16) : lastRowLocal;
#pragma unroll
            for (int g = 0; g < GROUPS && !lastRowOnly; ++g) {
                const bool inA = hitA && gstarA == g && (!OV || lastColA), inB = hitB && gstarB == g && (!OV || lastColB);
                if (__builtin_amdgcn_ballot_w64(inA || inB) == 0) continue;
                // the first row of the group that holds the group's maximum: keys again, value + 7 - row
                uint32_t m = pkMax(HS[8 * g] + both(7), HS[8 * g + 1] + both(6));
                m = pkMax3(m, HS[8 * g + 2] + both(5), HS[8 * g + 3] + both(4));
                m = pkMax3(m, HS[8 * g + 4] + both(3), HS[8 * g + 5] + both(2));
                m = pkMax3(m, HS[8 * g + 6] + both(1), HS[8 * g + 7]);
                if (inA) rowA = 8 * g + 7 - (int)(m & 7u);
                if (inB) rowB = 8 * g + 7 - (int)((m >> 16) & 7u);
            }
            // of the strips' hits the smallest column wins, then the smallest row: an earlier strip's at the same column
            if (hitA && (hA.bcol < 0 || hA.j < hA.bcol)) {
                hA.bcol = hA.j;
                hA.brow = hA.strip * kLanes + rowA;
            }
            if (hitB && (hB.bcol < 0 || hB.j < hB.bcol)) {
                hB.bcol = hB.j;
                hB.brow = hB.strip * kLanes + rowB;
            }
        }
        // ---- a half whose strip ends here: the optimum met, or no column left that could still matter
        const bool endA = busyA && (hitA || hA.j + 1 >= hA.need), endB = busyB && (hitB || hB.j + 1 >= hB.need);
        if (__builtin_amdgcn_ballot_w64(endA || endB) != 0) {
            auto finish = [&](bool end, ScanHalf& h, bool& busy, int stop, int L) {
                if (!end) return;
                busy = false;
                if (a.startQ != nullptr) {
                    // the start cell (what start_cells_kernel makes of the reverse pass: intraseq.hip); a scan that never
                    // met the optimum of the forward pass is reported, its window is the whole prefix
                    const bool found = h.bcol >= 0;
                    // (HW: the whole query in one gap before the target's first aligned residue is a border cell of the
                    // reversed problem, which no scan computes: start_cells_kernel, oracle/opal_oracle.c)
                    const bool border = !found && (lastRowOnly || OV) && stop == borderGap(h.Q - 1, open, ext);
                    // (OV: ... or the whole target prefix in one gap: start_cells_kernel asks in this order)
                    const bool borderT = OV && !found && !border && stop == borderGap(L - 1, open, ext);
                    if (!found && !border && !borderT) atomicExch(a.startChecks, h.out + 1);
                    const int sq = found ? h.Q - 1 - h.brow : borderT ? h.Q : 0, st = found ? L - 1 - h.bcol : border ? L : 0;
                    a.startQ[h.out] = sq;
                    a.startT[h.out] = st;
                    windowMax = max(windowMax, L - st);
                    rowsMax = max(rowsMax, h.Q - sq);
                    return;
                }
                a.score[h.out] = h.bcol >= 0 ? stop : INT32_MIN;
                if (a.endI) a.endI[h.out] = h.bcol >= 0 ? h.brow : -1;
                if (a.endJ) a.endJ[h.out] = h.bcol;
            };
            finish(endA, hA, busyA, stopA, LA);
            finish(endB, hB, busyB, stopB, LB);
            const uint32_t idle2 = (busyA ? 0u : 0x0000ffffu) | (busyB ? 0u : 0xffff0000u);
            tgt2 |= idle2;   // (a pattern no value reaches)
            incr2 = (busyA ? (uint32_t)(8 * ext) : 0u) | (busyB ? (uint32_t)(8 * ext) << 16 : 0u);
        }
        hA.j += busyA ? 1 : 0;   // (an idle half stays where it is: its column indexes the line it reads)
        hB.j += busyB ? 1 : 0;
        tgt2 += incr2;
    }
    if (a.startQ != nullptr) {
        // the slots of the traceback are sized by these two: one pair of atomics per wavefront
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            windowMax = max(windowMax, __shfl_xor(windowMax, off));
            rowsMax = max(rowsMax, __shfl_xor(rowsMax, off));
        }
        if (lane == 0 && windowMax > 0) {
            atomicMax(a.startChecks + 1, windowMax);
            atomicMax(a.startChecks + 2, rowsMax);
        }
    }
}

#include "perpair_packed_strips.inc"

}  // namespace

// Does the packed direction pass apply? (host side: host_full.inc)
//   open >= ext >= 0                                   borders are constants on the anti-diagonal scale
//   S + open + ext + bias in [0, 255]                  the profile holds unsigned bytes
//   maxS+ + open + max(open, -minS) <= 255             the flags' differences stay below 256 (see the proof in DESIGN.md)
//   Z - 4 open - ext - bias >= 0x0400, Z + best + (rows + columns) ext + slack <= 0x7BFF   normal half floats
// best: an upper bound of any cell of any window (min(rows, columns) x maxS+, or the query's own best).
bool packedTraceFits(int queryLength, int alphabet, int open, int ext, int maxScore, int minScore, int64_t rows,
                     int64_t columns, int64_t best, int* bias, int* stride, size_t* ldsBytes) {
    if (!(open >= ext && ext >= 0)) return false;
    const int b = std::max(0, -(minScore + open + ext));
    const int maxPos = std::max(maxScore, 0);
    if (maxScore + open + ext + b > 255) return false;
    if (maxPos + open + std::max(open, -minScore) > 255) return false;
    if (kPackedZero - 4 * (int64_t)open - ext - b < 0x0400) return false;
    const int64_t slack = 2 * ((int64_t)maxPos + open + ext) + b;
    if (kPackedZero + std::max<int64_t>(best, 0) + (rows + columns) * ext + slack > 0x7BFF) return false;
    const int pstride = perPairProfileStride(queryLength);
    const size_t profile = (size_t)(alphabet + 1) * pstride + 16;
    // (two workgroups of four wavefronts per CU when they fit its 160 KB, one of eight otherwise)
    size_t bytes = 4 * (size_t)kPkStageBytes + profile;
    if (2 * bytes > 156 * 1024) bytes = (size_t)kPkMaxWaves * kPkStageBytes + profile;
    if (bytes > 160 * 1024) return false;
    *bias = b;
    *stride = pstride;
    *ldsBytes = bytes;
    return true;
}

static inline bool firstUseHere(uint64_t* seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const uint64_t bit = 1ull << dev;
    const uint64_t old = __atomic_fetch_or(seen, bit, __ATOMIC_RELAXED);
    return !(old & bit);
}

template <bool BIASED, int WAVES, bool MULTI>
static hipError_t launchPackedTraceAs(const PerPairArgs& a, size_t ldsBytes, hipStream_t stream) {
    static uint64_t configured = 0;   // one bit per device: the attribute belongs to the device
    if (firstUseHere(&configured)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&perpair_packed_trace_kernel<BIASED, WAVES, MULTI>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            __atomic_fetch_and(&configured, ~(1ull << dev), __ATOMIC_RELAXED);
            return e;
        }
    }
    const int waves = (a.nJobs + 2 * kLanes - 1) / (2 * kLanes);
    hipLaunchKernelGGL((perpair_packed_trace_kernel<BIASED, WAVES, MULTI>), dim3((waves + WAVES - 1) / WAVES), dim3(WAVES * kLanes),
                       ldsBytes, stream, a);
    return hipGetLastError();
}

hipError_t launchPerPairPackedTrace(const PerPairArgs& a, size_t ldsBytes, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    if (a.profileStride <= 0 || a.dirs == nullptr || (a.dirStripColumns & 3)) return hipErrorInvalidValue;
    const size_t profile = (size_t)(a.alphabet + 1) * a.profileStride + 16;
    const bool four = ldsBytes == 4 * (size_t)kPkStageBytes + profile;
    // (host_full.inc: the rows between strips are there exactly when a window has more than 64 rows)
    const bool multi = a.boundary != nullptr;
    if (multi && a.boundaryStride <= 0) return hipErrorInvalidValue;
    const int which = (a.packedBias > 0 ? 4 : 0) + (four ? 2 : 0) + (multi ? 1 : 0);
    switch (which) {
        case 7: return launchPackedTraceAs<true, 4, true>(a, ldsBytes, stream);
        case 6: return launchPackedTraceAs<true, 4, false>(a, ldsBytes, stream);
        case 5: return launchPackedTraceAs<true, kPkMaxWaves, true>(a, ldsBytes, stream);
        case 4: return launchPackedTraceAs<true, kPkMaxWaves, false>(a, ldsBytes, stream);
        case 3: return launchPackedTraceAs<false, 4, true>(a, ldsBytes, stream);
        case 2: return launchPackedTraceAs<false, 4, false>(a, ldsBytes, stream);
        case 1: return launchPackedTraceAs<false, kPkMaxWaves, true>(a, ldsBytes, stream);
        default: return launchPackedTraceAs<false, kPkMaxWaves, false>(a, ldsBytes, stream);
    }
}


// Does the packed start-cell scan apply? (Smith-Waterman prefixes with a known optimum; host_full.inc)
//   open >= ext >= 0, S + open + bias in [0, 31] (times 8 an unsigned byte)
//   Z = 0x0400 + 8 (2 open + (Q + 64) ext + bias) and Z + 8 (best + longest ext) + slack <= 0x7BFF: normal half floats
bool packedScanFits(int queryLength, int alphabet, int open, int ext, int maxScore, int minScore, int64_t longest,
                    int64_t best, int* bias, int* zero, int* stride, size_t* ldsBytes) {
    if (!(open >= ext && ext >= 0)) return false;
    const int b = std::max(0, -(minScore + open));
    if (maxScore + open + b > 31) return false;
    // the zero: the last row's border, a bias below it, still a normal half float
    const int64_t z = 0x0400 + 8 * (2 * (int64_t)open + ((int64_t)queryLength + kLanes) * ext + b);
    const int64_t slack = 8 * (2 * ((int64_t)std::max(maxScore, 0) + open + ext) + b) + 8;
    if (z + 8 * (std::max<int64_t>(best, 0) + longest * ext) + slack > 0x7BFF) return false;
    const int pstride = perPairProfileStride(queryLength);
    const size_t bytes = (size_t)(alphabet + 1) * pstride + 16;
    if (bytes > 64 * 1024 || pstride > 0xfff0) return false;
    *bias = b;
    *zero = (int)z;
    *stride = pstride;
    *ldsBytes = bytes;
    return true;
}

template <int GROUPS>
static hipError_t launchPackedScanAs(const PerPairArgs& a, size_t ldsBytes, int blocks, hipStream_t stream) {
    const dim3 grid(blocks), block(kScanBlock);
    const bool biased = a.packedBias > 0;
    switch (a.scanLastRow) {
        case 2:
            if (biased) hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, true, 2>), grid, block, ldsBytes, stream, a);
            else hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, false, 2>), grid, block, ldsBytes, stream, a);
            break;
        case 1:
            if (biased) hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, true, 1>), grid, block, ldsBytes, stream, a);
            else hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, false, 1>), grid, block, ldsBytes, stream, a);
            break;
        default:
            if (biased) hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, true, 0>), grid, block, ldsBytes, stream, a);
            else hipLaunchKernelGGL((perpair_packed_scan_kernel<GROUPS, false, 0>), grid, block, ldsBytes, stream, a);
    }
    return hipGetLastError();
}

hipError_t launchPerPairPackedScan(const PerPairArgs& a, size_t ldsBytes, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    if (a.profileStride <= 0 || !a.reversed) return hipErrorInvalidValue;
    const int waves = (a.nJobs + 2 * kLanes - 1) / (2 * kLanes);
    if (a.queryLength > kLanes) {
        // several strips: a wavefront per 128 jobs of the sorted list, the rows between strips at a.boundary
        // ([wavefront][column][lane] x 8 bytes, a.boundaryStride columns per wavefront)
        if (a.boundary == nullptr || a.boundaryStride <= 0 || a.jobs == nullptr) return hipErrorInvalidValue;
        const dim3 grid((waves + kScanWaves - 1) / kScanWaves), block(kScanBlock);
        if (a.scanLastRow == 2) {
            if (a.packedBias > 0) hipLaunchKernelGGL((perpair_packed_scan_strips_kernel<true, true>), grid, block, ldsBytes, stream, a);
            else hipLaunchKernelGGL((perpair_packed_scan_strips_kernel<false, true>), grid, block, ldsBytes, stream, a);
        } else if (a.packedBias > 0) {
            hipLaunchKernelGGL((perpair_packed_scan_strips_kernel<true, false>), grid, block, ldsBytes, stream, a);
        } else {
            hipLaunchKernelGGL((perpair_packed_scan_strips_kernel<false, false>), grid, block, ldsBytes, stream, a);
        }
        return hipGetLastError();
    }
    if (a.jobCounter == nullptr || a.computeUnits <= 0) return hipErrorInvalidValue;
    if (a.jobs == nullptr && (!a.fwdScore || !a.fwdEndQ || !a.fwdEndT || !a.fwdOffsets)) return hipErrorInvalidValue;
    if (a.startQ != nullptr && (!a.startT || !a.startChecks)) return hipErrorInvalidValue;
    PerPairArgs b = a;
    if (b.refillLanes <= 0) b.refillLanes = 24;
    if (const char* e = tuned(Tune::SCAN_REFILL_LANES)) b.refillLanes = std::min(128, std::max(1, atoi(e)));   // (experiments)
    // persistent wavefronts: two per SIMD, three when the strip has up to 56 rows (167 registers)
    int perCu = a.queryLength <= 56 && a.scanLastRow == 0 ? 3 : 2;   // (HW, OV: their extra registers leave room for two)
    if (const char* e = tuned(Tune::SCAN_BLOCKS_PER_CU)) perCu = std::max(1, std::min(atoi(e), 8));   // (experiments)
    const int blocks = std::min((waves + kScanWaves - 1) / kScanWaves, a.computeUnits * perCu);
    switch ((a.queryLength + 7) / 8) {
        case 1: return launchPackedScanAs<1>(b, ldsBytes, blocks, stream);
        case 2: return launchPackedScanAs<2>(b, ldsBytes, blocks, stream);
        case 3: return launchPackedScanAs<3>(b, ldsBytes, blocks, stream);
        case 4: return launchPackedScanAs<4>(b, ldsBytes, blocks, stream);
        case 5: return launchPackedScanAs<5>(b, ldsBytes, blocks, stream);
        case 6: return launchPackedScanAs<6>(b, ldsBytes, blocks, stream);
        case 7: return launchPackedScanAs<7>(b, ldsBytes, blocks, stream);
        default: return launchPackedScanAs<8>(b, ldsBytes, blocks, stream);
    }
}

}  // namespace miopal
