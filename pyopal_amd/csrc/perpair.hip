// Lane-per-pair kernels for gfx950: the second and third pass of a `full` search
// (src/pyopal/opal.pxd:17-19, OPAL_SEARCH_ALIGNMENT). Three kernels, same model, same results:
//   perpair_kernel<MODE>           matrix rows in LDS; every region of the start-cell scan, any matrix / gaps
//   perpair_profile_kernel<MODE>   query profile in LDS, columns on a sliding scale, bit planes (round 3): the
//                                  scans and the directions of every mode when score + open fits a byte
//   perpair_scan_refill_kernel     the Smith-Waterman scan of a one-strip query by persistent wavefronts whose
//                                  lanes take the next pair when they are done
// The first one is described here, the others where they start.
//
// After the score/end pass every target has its own small problem anchored on its end cell:
// the reversed prefixes q[endQ..0] x t[endT..0] (start-location scan) and then the rectangle
// [start..end] (traceback directions). These problems share neither a query window nor a
// length, so the wavefront-per-pair anti-diagonal kernel (intraseq.hip) spends most of its
// steps filling and draining 64 lanes for ~50 columns. Here one LANE owns one pair and sweeps
// it column by column at 32 bit, like the inter-sequence kernel but with a private query
// window: H[64] / E[64] of the previous column live in VGPRs, the lane's query residues are
// pre-scaled LDS row offsets packed two per VGPR, the substitution matrix sits in LDS with an
// extra pad row and column. A wavefront runs until its longest pair is done. Query windows of
// more than 64 rows are swept strip by strip; the last row of a strip reaches the next strip
// through HBM, 512 bytes per wavefront-column.
//
// Rows beyond the lane's query window and columns beyond its target read the pad row/column
// (a large negative score): every value computed there is bounded by a valid cell that comes
// earlier in the column-major scan, so the strictly-greater candidate rule needs no masks
// (region "all cells"); the other regions test the row/column explicitly.
//
// Model and tie-breaks: oracle/opal_oracle.c (rules 5-7).
#include "common.h"
#include "tuning.h"

namespace miopal {

namespace {

constexpr int kNegInf = INT32_MIN / 4;
constexpr int kPadScore = -(1 << 28);
constexpr int kStride = kMaxAlphabet + 1;  // matrix rows in LDS, in ints (pad column included)
constexpr int kBlock = 256;

constexpr int kQueryLds = 4096;  // longest query the kernel stages in LDS

template <int MODE>  // kAllCells / kLastRow / kLastRowCol: start-location scan; kPerPairTrace: directions
__global__ __launch_bounds__(kBlock) void perpair_kernel(PerPairArgs a) {
    __shared__ int smat[kStride * kStride];
    __shared__ uint8_t qlds[kQueryLds];
    const int A = a.alphabet;
    for (int idx = threadIdx.x; idx < kStride * kStride; idx += kBlock) {
        const int q = idx / kStride, t = idx % kStride;
        // `open` is folded into the scores: the columns keep H - open (see the cell update)
        smat[idx] = (q < A && t < A) ? a.matrix[q * A + t] + a.gapOpen : kPadScore;
    }
    for (int x = threadIdx.x; x < a.queryLength; x += kBlock) qlds[x] = a.query[x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    // (hybrid direction pass: the leading wavefronts' jobs are outliers done by intraseq_kernel)
    const bool active = idx < a.nJobs && !(a.skipWaves != nullptr && (idx >> 6) < *a.skipWaves);
    PairJob job{};
    if (active) job = a.jobs[idx];
    const int Q = job.qLen, L = job.tLen;
    const int open = a.gapOpen, ext = a.gapExt;

    int maxQ = Q, maxL = L;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        maxQ = max(maxQ, __shfl_xor(maxQ, off));
        maxL = max(maxL, __shfl_xor(maxL, off));
    }
    maxQ = __builtin_amdgcn_readfirstlane(maxQ);
    maxL = __builtin_amdgcn_readfirstlane(maxL);
    const int nStrips = (maxQ + kLanes - 1) / kLanes;  // of the wavefront's tallest window

    // running answer over the strips: (score, column, row) of the first maximum
    int best = INT32_MIN, brow = -1, bcol = -1;
    // The known optimum ends the scan: the first cell that holds it is the first maximum. With several
    // strips the columns a lane still needs shrink instead: once a strip has met the optimum in column c
    // the strips below can only beat it in a smaller column (their rows are larger), so they - and the
    // rows handed down to them - stop at c.
    const bool stopOn = (job.rules & kRuleStop) != 0;
    const int stopScore = job.stop;
    int need = L;   // columns this lane still has to sweep
    const uint8_t* tptr = a.residues + job.tOff;
    const int64_t tStep = job.tStep;
    // strip boundaries of the wavefront: (H - open, F) of the strip's last row, per column
    int2* bnd = a.boundary ? a.boundary + (int64_t)(idx >> 6) * a.boundaryStride * kLanes + lane : nullptr;
    uint8_t* dirs = nullptr;
    if (MODE == kPerPairTrace) dirs = a.dirs + (int64_t)(idx >> 6) * a.dirWaveStride + lane;

    for (int s = 0; s < nStrips; ++s) {
        const int row0 = s * kLanes;
        const int rowsHere = min(maxQ - row0, kLanes);  // wave-uniform
        const bool toNext = s + 1 < nStrips;
        // LDS byte offsets of the lane's query rows, two per register; pad row beyond the window
        uint32_t qo[kLanes / 2];
#pragma unroll
        for (int i = 0; i < kLanes; i += 2) {
            const int q0 = row0 + i < Q ? qlds[job.qOff + (row0 + i) * job.qStep] : A;
            const int q1 = row0 + i + 1 < Q ? qlds[job.qOff + (row0 + i + 1) * job.qStep] : A;
            qo[i >> 1] = (uint32_t)(q0 * kStride * 4) | ((uint32_t)(q1 * kStride * 4) << 16);
        }
        // Previous column, kept as HM = H - open: the same number opens a gap to the right (E of
        // the next column) and downwards (F of the next row), and the diagonal gets `open` back
        // from the LDS scores.
        int HM[kLanes], E[kLanes];
#pragma unroll
        for (int i = 0; i < kLanes; ++i) {
            HM[i] = row0 + i < Q ? borderGap(row0 + i, open, ext) - open : kNegInf;  // column -1
            E[i] = kNegInf;
        }
        // first maximum of this strip (column-major inside the strip)
        int sbest = INT32_MIN, srow = -1, scol = -1;
        uint8_t* dcol = MODE == kPerPairTrace ? dirs + (int64_t)s * a.dirStripColumns * (kLanes / 2 * kLanes) : nullptr;
        // row above the strip at column j - 1 (diagonal of the strip's first row)
        int aboveHmPrev = (s == 0 ? 0 : borderGap(row0 - 1, open, ext)) - open;

        int tcolNext = (L > 0 ? (int)tptr[0] : A) * 4;
        int maxNeed = MODE == kPerPairTrace ? L : need;   // wave-uniform bound of this strip's sweep
        if (MODE != kPerPairTrace) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) maxNeed = max(maxNeed, __shfl_xor(maxNeed, off));
            maxNeed = __builtin_amdgcn_readfirstlane(maxNeed);
        } else {
            maxNeed = maxL;
        }
        for (int j = 0; j < maxNeed; ++j) {
            const int tcol = tcolNext;
            {
                int t = A;
                if (j + 1 < L) t = tptr[(int64_t)(j + 1) * tStep];
                tcolNext = t * 4;
            }
            const char* mcol = (const char*)smat + tcol;
            int hmUp, fUp;
            if (s == 0) {
                hmUp = borderGap(j, open, ext) - open;
                fUp = kNegInf;
            } else {
                const int2 above = bnd[(int64_t)j * kLanes];
                hmUp = above.x;
                fUp = above.y;
            }
            int hmDiag = aboveHmPrev;
            aboveHmPrev = hmUp;
            const bool colOk = j < L, lastCol = j == L - 1;
            const int bestBefore = sbest;
            int codeEven = 0;
#pragma unroll
            for (int i = 0; i < kLanes; ++i) {
                if ((i & 7) == 0 && i >= rowsHere) break;  // wave-uniform
                const uint32_t off = (i & 1) ? (qo[i >> 1] >> 16) : (qo[i >> 1] & 0xffffu);
                const int sc = *(const int*)(mcol + off);
                const int eOpen = HM[i], eExt = E[i] - ext;
                const int fOpen = hmUp, fExt = fUp - ext;
                const int e = max(eOpen, eExt);
                const int f = max(fOpen, fExt);
                const int d = hmDiag + sc;
                const int h = max(d, max(e, f));
                if (MODE == kPerPairTrace) {
                    // same code as intraseq_kernel<true>: diag > E (target gap) > F (query gap);
                    // inside a gap, closing it is preferred to extending it. Four bits per cell:
                    // rows i, i + 1 share a byte, [column][row pair][lane] (64-byte wave stores)
                    const int which = (h == d) ? 0 : (h == e) ? 1 : 2;
                    const int code = which | (e == eOpen ? 4 : 0) | (f == fOpen ? 8 : 0);
                    if (i & 1) dcol[((int64_t)j * (kLanes / 2) + (i >> 1)) * kLanes] = (uint8_t)(codeEven | (code << 4));
                    else codeEven = code;
                } else {
                    bool cand = true;  // kAllCells: pad rows / columns never beat a valid cell
                    if (MODE == kLastRow) cand = colOk && row0 + i == Q - 1;
                    if (MODE == kLastRowCol) cand = colOk && (row0 + i == Q - 1 || (lastCol && row0 + i < Q));
                    const bool take = cand && h > sbest;
                    sbest = take ? h : sbest;
                    srow = take ? i : srow;
                }
                const int hm = h - open;
                hmDiag = HM[i];
                HM[i] = hm;
                E[i] = e;
                hmUp = hm;
                fUp = f;
            }
            // (after a full strip hmUp / fUp are those of its last row)
            if (toNext) bnd[(int64_t)j * kLanes] = make_int2(hmUp, fUp);
            if (MODE == kPerPairTrace) {
                // score of the window = its last cell; lanes of a wavefront (sorted by length)
                // finish in a handful of columns, so the row select runs rarely
                const bool mine = lastCol && Q > row0 && Q <= row0 + kLanes;  // the lane's last strip
                if (__builtin_amdgcn_ballot_w64(mine) != 0) {
                    int v = 0;
#pragma unroll
                    for (int i = 0; i < kLanes; ++i) v = (row0 + i == Q - 1) ? HM[i] : v;
                    if (mine) best = v + open;
                }
            } else {
                scol = sbest != bestBefore ? j : scol;  // candidates only ever raise `sbest`
                // the optimum of the forward pass is the first maximum of this scan: a lane that
                // met it is finished; the wavefront leaves when no lane has work left
                // (also with a strip below: it needs no column beyond the ones every lane still needed here)
                const bool more = j + 1 < need && !(stopOn && sbest == stopScore);
                if (__builtin_amdgcn_ballot_w64(more) == 0) break;
            }
        }
        if (MODE != kPerPairTrace && Q > row0 && scol >= 0) {
            // fold the strip in: higher score, then smaller column (rows of later strips are larger)
            if (sbest > best || (sbest == best && scol < bcol)) {
                best = sbest;
                brow = row0 + srow;
                bcol = scol;
            }
        }
        if (MODE != kPerPairTrace && stopOn && best == stopScore) need = min(need, bcol + 1);
    }

    if (active) {
        int bi = -1, bj = -1;
        if (Q > 0 && L > 0) {
            bi = brow;
            bj = bcol;
        } else {
            // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
            best = 0;
            if (Q > 0) best = borderGap(Q - 1, open, ext);
            if (L > 0) best = borderGap(L - 1, open, ext);
        }
        a.score[job.out] = best;
        if (MODE != kPerPairTrace && a.endI) a.endI[job.out] = bi;
        if (MODE != kPerPairTrace && a.endJ) a.endJ[job.out] = bj;
    }
}

// ---- the same two passes with a query profile and columns on a sliding scale -----------------
// What the kernel above spends per cell: three instructions on the substitution score (row offset,
// address, LDS read), two per gap state, three on the candidate test. Here:
//   * the scores come from a QUERY PROFILE in LDS, prof[t][y] = matrix[query(y)][t] + open as a
//     signed byte, y running along the lane's walk of the query (reversed for the start-cell scan):
//     one aligned dword read + one v_alignbyte give four rows of the lane, and the byte is sign-
//     extended inside the add (SDWA);
//   * column j is kept on the scale X' = X + j * ext: extending a horizontal gap costs nothing
//     (E' = max(E', HM)), the vertical one subtracts once (F' = max(F', HM above) - ext), and
//     HM = H' - (open - ext) opens both;
//   * the scan keeps the column's maximum only; rows are compared when a lane meets its known
//     optimum (once per lane), or - without one - when a column beats the running maximum;
//   * the direction pass leaves four BIT PLANES per cell (came from the diagonal / from E / E was
//     opened / F was opened), each shifted in with a subtraction + v_alignbit, stored as dwords of 32
//     rows: [pair / 64][strip][column / 4][rows 0-31 | 32-63][pair % 64][plane][column % 4] - the flags of
//     four columns of a lane's half strip are one 64-byte line (round 4; walk_planes_kernel).
// Same model and tie-breaks as perpair_kernel (the flags are the same comparisons).
constexpr int kProfilePad = -128;

// One flag of a cell shifted into its plane: `larger` is the maximum the flag asks about, `part` the
// operand it may equal, so part - larger is negative exactly when they differ, and v_alignbit shifts
// that sign bit in: two full-rate VALU instructions per flag and no SGPR in between (a compare into an
// SGPR pair + add-with-carry took ~15 cycles each on gfx950, PMC: 8.4 cycles per VALU instruction for
// the whole kernel). The planes collect 1 = differs; they are inverted when stored.
__device__ __forceinline__ uint32_t shiftInDiffers(uint32_t plane, int larger, int part) {
    return __builtin_amdgcn_alignbit(plane, (uint32_t)(part - larger), 31);
}

__device__ __forceinline__ int pick2(bool second, int a, int b) {
    asm volatile("" : "+v"(a), "+v"(b));
    return second ? b : a;
}

template <int MODE>  // kAllCells / kLastRow / kLastRowCol (start-cell scan) or kPerPairTrace (directions)
__global__ __launch_bounds__(kBlock) void perpair_profile_kernel(PerPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) int8_t prof[];
    const int A = a.alphabet;
    const int Qtot = a.queryLength;
    const int pstride = a.profileStride;   // bytes per residue row, a multiple of 4, >= Qtot + 64 + 8
    for (int idx = threadIdx.x; idx < (A + 1) * pstride; idx += kBlock) {
        const int t = idx / pstride, y = idx - t * pstride;
        int v = kProfilePad;
        if (t < A && y < Qtot) v = a.matrix[(int)a.query[a.reversed ? Qtot - 1 - y : y] * A + t] + a.gapOpen;
        prof[idx] = (int8_t)v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const bool active = idx < a.nJobs && !(a.skipWaves != nullptr && (idx >> 6) < *a.skipWaves);
    PairJob job{};
    if (active) job = a.jobs[idx];
    const int Q = job.qLen, L = job.tLen;
    const int open = a.gapOpen, ext = a.gapExt, c = open - ext;

    int maxQ = Q, maxL = L;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        maxQ = max(maxQ, __shfl_xor(maxQ, off));
        maxL = max(maxL, __shfl_xor(maxL, off));
    }
    maxQ = __builtin_amdgcn_readfirstlane(maxQ);
    maxL = __builtin_amdgcn_readfirstlane(maxL);
    const int nStrips = (maxQ + kLanes - 1) / kLanes;

    int best = INT32_MIN, brow = -1, bcol = -1;
    const bool stopOn = (job.rules & kRuleStop) != 0;
    const int stopScore = job.stop;
    int need = L;
    const uint8_t* tptr = a.residues + job.tOff;
    const int64_t tStep = job.tStep;
    int2* bnd = a.boundary ? a.boundary + (int64_t)(idx >> 6) * a.boundaryStride * kLanes + lane : nullptr;
    uint8_t* dirs = nullptr;
    // (lanes side by side, one dword each: the walk reads one or two planes per step, and the pairs of a
    // wavefront - sorted by length, walking back from similar cells - share its cache lines)
    // (round 4: per block of FOUR columns and half strip, one 64-byte line per lane: [plane][column] dwords - every
    // flag of the cells a path crosses in those columns comes with one line, which the walk fetches once)
    if (MODE == kPerPairTrace) dirs = a.dirs + (int64_t)(idx >> 6) * a.dirWaveStride + lane * 64;
    // the lane's first row along the profile
    const int yBase = a.reversed ? Qtot - 1 - job.qOff : job.qOff;

    for (int s = 0; s < nStrips; ++s) {
        const int row0 = s * kLanes;
        const int rowsHere = min(maxQ - row0, kLanes);  // wave-uniform
        const bool toNext = s + 1 < nStrips;
        // rows beyond the query read the pad bytes behind it (64 of them: a whole strip may)
        const int ys = min(yBase + row0, Qtot);
        const uint32_t shift = (uint32_t)ys & 3u;
        const int yAligned = ys & ~3;
        int HM[kLanes], E[kLanes];
#pragma unroll
        for (int i = 0; i < kLanes; ++i) {
            HM[i] = row0 + i < Q ? borderGap(row0 + i, open, ext) - open : kNegInf;  // column -1 (on its own scale)
            E[i] = kNegInf;
        }
        int sbest = INT32_MIN, srow = -1, scol = -1;
        // regions "last row" / "last row or column" (HW / OV): the candidates of a column are one row of the lane's
        // own - picked out of the 64 registers by the bits of its index, 63 selects a column - and, in the lane's
        // last column, every row of its window
        const int lastLocal = Q - 1 - row0;
        const bool lastHere = lastLocal >= 0 && lastLocal < kLanes;
        uint8_t* dcol = MODE == kPerPairTrace ? dirs + (int64_t)s * a.dirStripColumns * (kLanes / 2 * kLanes) : nullptr;
        int aboveHmPrev = (s == 0 ? 0 : borderGap(row0 - 1, open, ext)) - open;
        // bits of the rows a half strip really holds sit at the top of its planes
        const int rows8 = (rowsHere + 7) & ~7;
        const int pad0 = 32 - min(rows8, 32), pad1 = 32 - max(rows8 - 32, 0);

        // The lane's target residues, four columns per load and four columns ahead (a wavefront holds two
        // columns' worth of work per SIMD at most: a byte per column fetched one column ahead was waited for).
        // The load is unconditional - its address clamped into the database - and what it returned is only
        // put in place (column j0 in the lowest byte, the pad residue beyond the target) when it is needed.
        const uint32_t padWord = (uint32_t)A * 0x01010101u;
        const uint8_t* const dbLo = a.residues;
        const uint8_t* const dbHi = a.residues + a.residueCount - 4;
        auto wanted = [&](int j0) { return a.reversed ? tptr - j0 - 3 : tptr + j0; };
        auto fetchRaw = [&](int j0) -> uint32_t {
            const uint8_t* at = wanted(j0);
            at = at < dbLo ? dbLo : at;
            at = at > dbHi ? dbHi : at;
            uint32_t w;
            __builtin_memcpy(&w, at, 4);
            return w;
        };
        auto inPlace = [&](uint32_t raw, int j0) -> uint32_t {
            const uint8_t* at = wanted(j0);
            const int64_t below = dbLo - at, above = at - dbHi;   // > 0: the clamp moved the load by this many bytes
            uint32_t w = raw;
            if (below > 0) w = below >= 4 ? 0u : raw << (8 * (int)below);
            if (above > 0) w = above >= 4 ? 0u : raw >> (8 * (int)above);
            if (a.reversed) w = __builtin_bswap32(w);
            const int valid = L - j0;   // columns of the four that exist
            const uint32_t keep = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : (1u << (8 * valid)) - 1u;
            return (w & keep) | (padWord & ~keep);
        };
        uint32_t wcur = padWord, rawNext = fetchRaw(0);
        int2 aboveNext = make_int2(0, 0);
        if (s > 0) aboveNext = bnd[0];
        int maxNeed = need;
        if (MODE != kPerPairTrace) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) maxNeed = max(maxNeed, __shfl_xor(maxNeed, off));
            maxNeed = __builtin_amdgcn_readfirstlane(maxNeed);
        } else {
            maxNeed = maxL;
        }
        // Direction pass: the planes of four columns leave together, 16 bytes per lane and plane (the three
        // columns before the current one wait in registers; the sweep runs to a multiple of four columns -
        // pad columns beyond every target, whose bits nobody reads).
        const int sweep = MODE == kPerPairTrace ? (maxNeed + 3) & ~3 : maxNeed;
        uint32_t held[3][8];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 8; ++y) held[x][y] = 0;
        int scale = 0;   // j * ext
        for (int j = 0; j < sweep; ++j, scale += ext) {
            if ((j & 3) == 0) {
                wcur = inPlace(rawNext, j);
                rawNext = fetchRaw(j + 4);
            }
            const int t = (int)((wcur >> (8 * (j & 3))) & 0xffu);
            const uint32_t* prow = (const uint32_t*)(prof + t * pstride + yAligned);
            int hmUp, fUp;
            if (s == 0) {
                hmUp = borderGap(j, open, ext) + scale - c;
                fUp = kNegInf;
            } else {
                // (the row above, fetched one column ahead)
                const int2 above = aboveNext;
                aboveNext = bnd[(int64_t)min(j + 1, maxNeed - 1) * kLanes];   // (unconditional: no copy behind the load)
                hmUp = above.x;
                fUp = above.y;
            }
            int hmDiag = aboveHmPrev;
            aboveHmPrev = hmUp;
            int cm = INT32_MIN;
            uint32_t pD[2] = {0, 0}, pE[2] = {0, 0}, pO[2] = {0, 0}, pF[2] = {0, 0};
            uint32_t wlo = prow[0], four = 0;
            // (opaque per column: the eight comparisons below stay scalar compares here instead of eight
            // lane masks kept - and spilled - across the loop)
            int groups = rows8 >> 3;
            asm volatile("" : "+s"(groups));
#pragma unroll
            for (int i = 0; i < kLanes; ++i) {
                if ((i & 7) == 0 && (i >> 3) >= groups) break;  // wave-uniform
                if ((i & 3) == 0) {
                    const uint32_t whi = prow[(i >> 2) + 1];
                    four = __builtin_amdgcn_alignbyte(whi, wlo, shift);
                    wlo = whi;
                }
                const int sc = (int)(int8_t)(four >> (8 * (i & 3)));
                const int d = hmDiag + sc;
                const int e = max(E[i], HM[i]);
                const int fm = max(fUp, hmUp);
                const int f = fm - ext;
                const int h = max(d, max(e, f));
                if (MODE == kPerPairTrace) {
                    // (four accumulators: two flags chained into one - two bits a row, one dword a step for the
                    // walk - cost the kernel a third more time than the walk gained)
                    const int k = i >> 5;
                    pD[k] = shiftInDiffers(pD[k], h, d);
                    pE[k] = shiftInDiffers(pE[k], h, e);
                    pO[k] = shiftInDiffers(pO[k], e, HM[i]);
                    pF[k] = shiftInDiffers(pF[k], fm, hmUp);
                } else if (MODE != kLastRow) {
                    cm = max(cm, h);
                }
                const int hm = h - c;
                hmDiag = HM[i];
                HM[i] = hm;
                E[i] = e;
                hmUp = hm;
                fUp = f;
            }
            if (toNext && j < maxNeed) bnd[(int64_t)j * kLanes] = make_int2(hmUp, fUp);
            if (MODE == kPerPairTrace) {
                // (the bits of the rows a half strip really holds sit at the top of its planes)
                uint32_t now[8] = {~pD[0] << pad0, ~pE[0] << pad0, ~pO[0] << pad0, ~pF[0] << pad0,
                                   ~pD[1] << pad1, ~pE[1] << pad1, ~pO[1] << pad1, ~pF[1] << pad1};
                if ((j & 3) == 3) {   // wave-uniform
                    // block of four columns: two tiles (rows 0-31 | 32-63) of 64 lanes x 64 bytes; the four stores of
                    // a lane fill its line - whole lines leave for HBM
                    uint8_t* at = dcol + (int64_t)(j >> 2) * (2 * kLanes * 64);
#pragma unroll
                    for (int y = 0; y < 8; ++y) {
                        if (y >= 4 && rows8 <= 32) break;   // wave-uniform
                        *(uint4*)(at + (y >> 2) * (kLanes * 64) + (y & 3) * 16) = make_uint4(held[0][y], held[1][y], held[2][y], now[y]);
                    }
                } else {
#pragma unroll
                    for (int y = 0; y < 8; ++y) {
                        held[0][y] = held[1][y];
                        held[1][y] = held[2][y];
                        held[2][y] = now[y];
                    }
                }
                const bool mine = j == L - 1 && Q > row0 && Q <= row0 + kLanes;  // the lane's last strip
                if (__builtin_amdgcn_ballot_w64(mine) != 0) {
                    int v = 0;
#pragma unroll
                    for (int i = 0; i < kLanes; ++i) v = (row0 + i == Q - 1) ? HM[i] : v;
                    if (mine) best = v + c - scale;
                }
            } else {
                // pad rows and columns are bounded by a valid cell that comes earlier in the column-major
                // scan (perpair_kernel), so the column's maximum may include them
                int top = MODE == kLastRow ? INT32_MIN : cm - scale;   // ("last row" keeps no column maximum)
                bool hit = top > sbest && (!stopOn || top >= stopScore);
                if (MODE == kLastRow || MODE == kLastRowCol) {
                    // H of the lane's last query row in this column
                    // (every level spelled out: a loop over the levels left the array in scratch memory)
                    const bool b0 = lastLocal & 1, b1 = lastLocal & 2, b2 = lastLocal & 4, b3 = lastLocal & 8,
                               b4 = lastLocal & 16, b5 = lastLocal & 32;
                    int p32[32], p16[16], p8[8], p4[4], p2[2];
                    // (pick2 hides the operands: "one of two neighbouring array elements" otherwise becomes a load at a
                    // computed address, and the 64 registers an array in scratch memory)
#pragma unroll
                    for (int k = 0; k < 32; ++k) p32[k] = pick2(b0, HM[2 * k], HM[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 16; ++k) p16[k] = pick2(b1, p32[2 * k], p32[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) p8[k] = pick2(b2, p16[2 * k], p16[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) p4[k] = pick2(b3, p8[2 * k], p8[2 * k + 1]);
#pragma unroll
                    for (int k = 0; k < 2; ++k) p2[k] = pick2(b4, p4[2 * k], p4[2 * k + 1]);
                    const int topLast = pick2(b5, p2[0], p2[1]) + c - scale;
                    const bool lastRowHit = lastHere && j < L && topLast > sbest && (!stopOn || topLast >= stopScore);
                    // the lane's last column: every row of its window is a candidate, in row order - the column's first
                    // maximum (rows beyond the window stay below the one above them); elsewhere the last row alone
                    const bool wholeColumn = MODE == kLastRowCol && j == L - 1 && row0 < Q;
                    hit = wholeColumn && hit;
                    if (lastRowHit && !hit) {
                        sbest = topLast;
                        srow = lastLocal;
                        scol = j;
                    }
                }
                if (__builtin_amdgcn_ballot_w64(hit) != 0) {
                    // the first row that holds the maximum: the first group of eight rows that does, then its rows -
                    // only the groups a hit lane points at are looked at (perpair_scan_refill_kernel)
                    const int want = cm - c;
                    int gstar = -1;
#pragma unroll
                    for (int g = kLanes / 8 - 1; g >= 0; --g) {
                        if (8 * g >= rows8) continue;
                        int m = HM[8 * g];
#pragma unroll
                        for (int r = 1; r < 8; ++r) m = max(m, HM[8 * g + r]);
                        gstar = m == want ? g : gstar;
                    }
                    int first = -1;
#pragma unroll
                    for (int g = 0; g < kLanes / 8; ++g) {
                        if (__builtin_amdgcn_ballot_w64(hit && gstar == g) == 0) continue;
#pragma unroll
                        for (int r = 7; r >= 0; --r) first = (gstar == g && HM[8 * g + r] == want) ? 8 * g + r : first;
                    }
                    if (hit) {
                        sbest = top;
                        srow = first;
                        scol = j;
                    }
                }
                const bool more = j + 1 < need && !(stopOn && sbest == stopScore);
                if (__builtin_amdgcn_ballot_w64(more) == 0) break;
            }
        }
        if (MODE != kPerPairTrace && Q > row0 && scol >= 0) {
            if (sbest > best || (sbest == best && scol < bcol)) {
                best = sbest;
                brow = row0 + srow;
                bcol = scol;
            }
        }
        if (MODE != kPerPairTrace && stopOn && best == stopScore) need = min(need, bcol + 1);
    }

    if (active) {
        int bi = -1, bj = -1;
        if (Q > 0 && L > 0) {
            bi = brow;
            bj = bcol;
        } else {
            best = 0;
            if (Q > 0) best = borderGap(Q - 1, open, ext);
            if (L > 0) best = borderGap(L - 1, open, ext);
        }
        a.score[job.out] = best;
        if (MODE != kPerPairTrace && a.endI) a.endI[job.out] = bi;
        if (MODE != kPerPairTrace && a.endJ) a.endJ[job.out] = bj;
    }
}

// ---- the start-cell scan of one-strip queries with lanes that take the next pair when they are done ------
// perpair_profile_kernel<kAllCells> runs a wavefront until its longest lane is done: the reversed-prefix scans of
// cfg3 (53-aa query, gap 3/1) need 60 columns on average and 95 for the longest of 64 - a third of the lane
// columns idle. Here a wavefront is persistent: every fourth column it counts the lanes that have met their
// optimum, and once kRefillLanes of them are idle they take the next pairs of the list (one atomic per
// wavefront and refill), start over at column 0 of their own pair and run beside the lanes still at work. A lane's
// column is its own (border, scale and target position follow it); refills happen on multiples of four columns,
// so the four-residue loads stay in step. Same cell, same candidates, same results as the kernel above.
constexpr int kRefillLanes = 12;

template <int MODE>  // kAllCells, kLastRow or kLastRowCol
__global__ __launch_bounds__(kBlock) void perpair_scan_refill_kernel(PerPairArgs a) {
    extern __shared__ __attribute__((aligned(16))) int8_t prof[];
    const int A = a.alphabet;
    const int Qtot = a.queryLength;   // <= 64: one strip
    const int pstride = a.profileStride;
    for (int idx = threadIdx.x; idx < (A + 1) * pstride; idx += kBlock) {
        const int t = idx / pstride, y = idx - t * pstride;
        int v = kProfilePad;
        if (t < A && y < Qtot) v = a.matrix[(int)a.query[a.reversed ? Qtot - 1 - y : y] * A + t] + a.gapOpen;
        prof[idx] = (int8_t)v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int open = a.gapOpen, ext = a.gapExt, c = open - ext;
    const int rows8 = (min(Qtot, kLanes) + 7) & ~7;
    const uint32_t padWord = (uint32_t)A * 0x01010101u;
    const uint8_t* const dbLo = a.residues;
    const uint8_t* const dbHi = a.residues + a.residueCount - 4;

    // the lane's pair
    bool busy = false;
    int Q = 0, L = 0, out = 0, stopScore = 0;
    bool stopOn = false;
    const uint8_t* tptr = a.residues;
    int yAligned = Qtot & ~3;
    uint32_t shift = (uint32_t)Qtot & 3u;
    int j = 0, scale = 0;
    int HM[kLanes], E[kLanes];
#pragma unroll
    for (int i = 0; i < kLanes; ++i) HM[i] = E[i] = kNegInf;
    int aboveHmPrev = -open;
    int sbest = INT32_MIN, srow = -1, scol = -1;
    uint32_t wcur = padWord, rawNext = padWord;
    bool exhausted = false;   // wave-uniform: no pair left to take

    auto wanted = [&](int j0) { return a.reversed ? tptr - j0 - 3 : tptr + j0; };
    auto fetchRaw = [&](int j0) -> uint32_t {
        const uint8_t* at = wanted(j0);
        at = at < dbLo ? dbLo : at;
        at = at > dbHi ? dbHi : at;
        uint32_t w;
        __builtin_memcpy(&w, at, 4);
        return w;
    };
    auto inPlace = [&](uint32_t raw, int j0) -> uint32_t {
        const uint8_t* at = wanted(j0);
        const int64_t below = dbLo - at, above = at - dbHi;
        uint32_t w = raw;
        if (below > 0) w = below >= 4 ? 0u : raw << (8 * (int)below);
        if (above > 0) w = above >= 4 ? 0u : raw >> (8 * (int)above);
        if (a.reversed) w = __builtin_bswap32(w);
        const int valid = L - j0;
        const uint32_t keep = valid >= 4 ? 0xffffffffu : valid <= 0 ? 0u : (1u << (8 * valid)) - 1u;
        return (w & keep) | (padWord & ~keep);
    };

    for (int w = 0;; ++w) {
        if ((w & 3) == 0) {
            const uint64_t idleMask = __builtin_amdgcn_ballot_w64(!busy);
            const int idle = __builtin_popcountll(idleMask);
            if (exhausted && idle == kLanes) break;
            if (!exhausted && (idle >= a.refillLanes || w == 0)) {
                int base = 0;
                if (lane == 0) base = atomicAdd(a.jobCounter, idle);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + idle >= a.nJobs) exhausted = true;
                if (!busy) {
                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idleMask, 0));
                    const int k = base + rank;
                    if (k < a.nJobs) {
                        const PairJob job = a.jobs[k];
                        Q = job.qLen;
                        L = job.tLen;
                        out = job.out;
                        stopOn = (job.rules & kRuleStop) != 0;
                        stopScore = job.stop;
                        tptr = a.residues + job.tOff;
                        const int ys = min(a.reversed ? Qtot - 1 - job.qOff : job.qOff, Qtot);
                        shift = (uint32_t)ys & 3u;
                        yAligned = ys & ~3;
                        j = 0;
                        scale = 0;
                        sbest = INT32_MIN;
                        srow = scol = -1;
                        aboveHmPrev = -open;
#pragma unroll
                        for (int i = 0; i < kLanes; ++i) {
                            HM[i] = i < Q ? borderGap(i, open, ext) - open : kNegInf;
                            E[i] = kNegInf;
                        }
                        rawNext = fetchRaw(0);
                        busy = Q > 0 && L > 0;
                        if (!busy) {
                            // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
                            int v = 0;
                            if (Q > 0) v = borderGap(Q - 1, open, ext);
                            if (L > 0) v = borderGap(L - 1, open, ext);
                            a.score[out] = v;
                            if (a.endI) a.endI[out] = -1;
                            if (a.endJ) a.endJ[out] = -1;
                        }
                    }
                }
                if (exhausted && __builtin_amdgcn_ballot_w64(busy) == 0) break;
            }
            wcur = inPlace(rawNext, j);
            rawNext = fetchRaw(j + 4);
        }
        const int t = (int)((wcur >> (8 * (w & 3))) & 0xffu);
        const uint32_t* prow = (const uint32_t*)(prof + t * pstride + yAligned);
        int hmUp = borderGap(j, open, ext) + scale - c;
        int fUp = kNegInf;
        int hmDiag = aboveHmPrev;
        aboveHmPrev = hmUp;
        int cm = INT32_MIN;
        uint32_t wlo = prow[0], four = 0;
        int groups = rows8 >> 3;
        asm volatile("" : "+s"(groups));
#pragma unroll
        for (int i = 0; i < kLanes; ++i) {
            if ((i & 7) == 0 && (i >> 3) >= groups) break;  // wave-uniform
            if ((i & 3) == 0) {
                const uint32_t whi = prow[(i >> 2) + 1];
                four = __builtin_amdgcn_alignbyte(whi, wlo, shift);
                wlo = whi;
            }
            const int sc = (int)(int8_t)(four >> (8 * (i & 3)));
            const int d = hmDiag + sc;
            const int e = max(E[i], HM[i]);
            const int fm = max(fUp, hmUp);
            const int f = fm - ext;
            const int h = max(d, max(e, f));
            if (MODE != kLastRow) cm = max(cm, h);
            const int hm = h - c;
            hmDiag = HM[i];
            HM[i] = hm;
            E[i] = e;
            hmUp = hm;
            fUp = f;
        }
        const int top = MODE == kLastRow ? INT32_MIN : cm - scale;
        bool hit = busy && top > sbest && (!stopOn || top >= stopScore);
        if (MODE == kLastRow || MODE == kLastRowCol) {
            // HW / OV: the lane's last query row (perpair_profile_kernel), every row in its last column
            const int lastLocal = Q - 1;
            const bool b0 = lastLocal & 1, b1 = lastLocal & 2, b2 = lastLocal & 4, b3 = lastLocal & 8,
                       b4 = lastLocal & 16, b5 = lastLocal & 32;
            int p32[32], p16[16], p8[8], p4[4], p2[2];
#pragma unroll
            for (int k = 0; k < 32; ++k) p32[k] = pick2(b0, HM[2 * k], HM[2 * k + 1]);
#pragma unroll
            for (int k = 0; k < 16; ++k) p16[k] = pick2(b1, p32[2 * k], p32[2 * k + 1]);
#pragma unroll
            for (int k = 0; k < 8; ++k) p8[k] = pick2(b2, p16[2 * k], p16[2 * k + 1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) p4[k] = pick2(b3, p8[2 * k], p8[2 * k + 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) p2[k] = pick2(b4, p4[2 * k], p4[2 * k + 1]);
            const int topLast = pick2(b5, p2[0], p2[1]) + c - scale;
            const bool lastRowHit = busy && j < L && topLast > sbest && (!stopOn || topLast >= stopScore);
            const bool wholeColumn = MODE == kLastRowCol && j == L - 1;
            hit = wholeColumn && hit;
            if (lastRowHit && !hit) {
                sbest = topLast;
                srow = lastLocal;
                scol = j;
            }
        }
        if (__builtin_amdgcn_ballot_w64(hit) != 0) {
            // the first row that holds the maximum: the first group of eight rows that does (their maxima are taken
            // here, from the column just written - kept across the column they cost 50 registers and a wavefront per
            // SIMD), then its rows - only the groups a hit lane points at are looked at (comparing all 64 rows was a
            // quarter of this kernel's instructions)
            const int want = cm - c;
            int gstar = -1;
#pragma unroll
            for (int g = kLanes / 8 - 1; g >= 0; --g) {
                if (8 * g >= rows8) continue;
                int m = HM[8 * g];
#pragma unroll
                for (int r = 1; r < 8; ++r) m = max(m, HM[8 * g + r]);
                gstar = m == want ? g : gstar;
            }
            int first = -1;
#pragma unroll
            for (int g = 0; g < kLanes / 8; ++g) {
                if (__builtin_amdgcn_ballot_w64(hit && gstar == g) == 0) continue;
#pragma unroll
                for (int r = 7; r >= 0; --r) first = (gstar == g && HM[8 * g + r] == want) ? 8 * g + r : first;
            }
            if (hit) {
                sbest = top;
                srow = first;
                scol = j;
            }
        }
        const bool more = j + 1 < L && !(stopOn && sbest == stopScore);
        const bool done = busy && !more;
        if (__builtin_amdgcn_ballot_w64(done) != 0) {
            if (done) {
                a.score[out] = sbest;
                if (a.endI) a.endI[out] = scol >= 0 ? srow : -1;
                if (a.endJ) a.endJ[out] = scol;
                busy = false;
            }
        }
        ++j;
        scale += ext;
    }
}

}  // namespace

// LDS of the profile kernel for a query of `queryLength` residues (0: the profile does not apply)
size_t perPairProfileBytes(int queryLength, int alphabet, int* stride) {
    // (an odd number of dwords: with a multiple of 128 bytes - a 53-residue query's 128 - every residue's row starts
    // in the same LDS bank and the lanes of a read, which hold different residues at the same row, queue up behind one
    // another: SQ_LDS_BANK_CONFLICT 88 % of the LDS cycles, the LDS busy half the kernel's time)
    const int pstride = perPairProfileStride(queryLength);
    *stride = pstride;
    const size_t bytes = (size_t)(alphabet + 1) * pstride + 16;
    return bytes <= 64 * 1024 ? bytes : 0;
}

hipError_t launchPerPair(const PerPairArgs& a, int mode, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    const dim3 grid((a.nJobs + kBlock - 1) / kBlock), block(kBlock);
    if (a.profileStride > 0 && (mode == kAllCells || mode == kPerPairTrace || mode == kLastRow || mode == kLastRowCol)) {
        const size_t lds = (size_t)(a.alphabet + 1) * a.profileStride + 16;
        if (mode != kPerPairTrace && a.jobCounter != nullptr && a.queryLength <= kLanes && a.computeUnits > 0) {
            // persistent wavefronts, three per SIMD (163 VGPRs; the HW / OV regions' row select takes 190: two)
            int perCu = mode == kAllCells ? 3 : 2;
            if (const char* e = tuned(Tune::SCAN_BLOCKS_PER_CU)) perCu = std::max(1, atoi(e));   // (experiments)
            PerPairArgs b = a;
            b.refillLanes = kRefillLanes;
            if (const char* e = tuned(Tune::SCAN_REFILL_LANES)) b.refillLanes = std::min(64, std::max(1, atoi(e)));
            const int blocks = (int)std::min<int64_t>(((int64_t)a.nJobs + kBlock - 1) / kBlock, (int64_t)a.computeUnits * perCu);
            if (mode == kAllCells) hipLaunchKernelGGL(perpair_scan_refill_kernel<kAllCells>, dim3(blocks), block, lds, stream, b);
            else if (mode == kLastRow) hipLaunchKernelGGL(perpair_scan_refill_kernel<kLastRow>, dim3(blocks), block, lds, stream, b);
            else hipLaunchKernelGGL(perpair_scan_refill_kernel<kLastRowCol>, dim3(blocks), block, lds, stream, b);
            return hipGetLastError();
        }
        if (mode == kAllCells) hipLaunchKernelGGL((perpair_profile_kernel<kAllCells>), grid, block, lds, stream, a);
        else if (mode == kLastRow) hipLaunchKernelGGL((perpair_profile_kernel<kLastRow>), grid, block, lds, stream, a);
        else if (mode == kLastRowCol) hipLaunchKernelGGL((perpair_profile_kernel<kLastRowCol>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((perpair_profile_kernel<kPerPairTrace>), grid, block, lds, stream, a);
        return hipGetLastError();
    }
    switch (mode) {
        case kAllCells: hipLaunchKernelGGL((perpair_kernel<kAllCells>), grid, block, 0, stream, a); break;
        case kLastRow: hipLaunchKernelGGL((perpair_kernel<kLastRow>), grid, block, 0, stream, a); break;
        case kLastRowCol: hipLaunchKernelGGL((perpair_kernel<kLastRowCol>), grid, block, 0, stream, a); break;
        case kPerPairTrace: hipLaunchKernelGGL((perpair_kernel<kPerPairTrace>), grid, block, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace miopal
