// Intra-sequence kernels for gfx950: one wavefront aligns one (query, target)
// pair at 32 bit, sweeping anti-diagonals of a 64-row strip (lane = query row).
//
// This is the exact, all-modes path behind opalSearchDatabase
// (src/pyopal/opal.pxd:38-52): NW / HW / OV / SW scores with end locations,
// the reversed-prefix pass that finds start locations, and the direction
// matrix for the traceback. It also recomputes targets whose packed 16-bit
// lanes saturated in the inter-sequence kernel (the 32-bit rung of the
// reference's 8/16/32-bit overflow ladder, src/pyopal/lib.pyx:1283-1289) and
// carries targets too long for one-lane-per-target scheduling.
//
// Model and tie-breaks: oracle/opal_oracle.c.
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace miopal {

constexpr int kNegInf = INT32_MIN / 4;
constexpr int kJobsPerBlock = 4;
constexpr int kMatStride = kMaxAlphabet + 1;  // odd stride: rows fall on different LDS banks

// better(a, b): does candidate a replace b in the column-major, strictly-greater scan?
static __device__ __forceinline__ bool better(int sa, int ja, int ia, int sb, int jb, int ib) {
    if (sa != sb) return sa > sb;
    if (ja != jb) return ja < jb;
    return ia < ib;
}

template <bool TRACE>
__global__ __launch_bounds__(kJobsPerBlock * kLanes) void intraseq_kernel(IntraseqArgs a) {
    __shared__ int smat[kMaxAlphabet * kMatStride];
    const int A = a.alphabet;
    for (int idx = threadIdx.x; idx < A * A; idx += blockDim.x)
        smat[(idx / A) * kMatStride + (idx % A)] = a.matrix[idx];
    __syncthreads();

    // (readfirstlane tells the compiler what it cannot see: the wavefront index, and with it the
    // job and every loop bound derived from it, is uniform - scalar loads, scalar loop control)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int jobIdx = blockIdx.x * kJobsPerBlock + wave;
    if (jobIdx >= a.nJobs) return;  // wave-uniform, after the only barrier
    const bool headOnly = a.headWaves != nullptr;
    if (headOnly && jobIdx >= *a.headWaves * kLanes) return;

    PairJob job = a.jobs[jobIdx];
    if (headOnly) {
        job.dirOff = (int64_t)jobIdx * a.headDirStride;
        job.wsOff = (int64_t)jobIdx * a.headWsStride;
    }
    const int Q = job.qLen, L = job.tLen;
    // a pair is a chain of L + 63 dependent steps: beside the packed kernel (side stream)
    // it should win the SIMD's issue arbitration, it needs few slots
    if (a.raisePriority) __builtin_amdgcn_s_setprio(3);
    const bool topGap = job.rules & 1, leftGap = job.rules & 2, floor0 = job.rules & 4;
    const int hFloor = floor0 ? 0 : INT32_MIN;   // Smith-Waterman floor as a max that is always there
    const int region = (job.rules >> 4) & 3;
    const int open = a.gapOpen, ext = a.gapExt;

    int best = floor0 ? 0 : INT32_MIN;
    int bi = -1, bj = -1;

    if (Q > 0 && L > 0) {
        const int nStrips = (Q + kLanes - 1) / kLanes;
        const int nSteps = L + kLanes - 1;  // layout of the direction bytes (walk_kernel)
        const bool stopEnabled = (job.rules & kRuleStop) && nStrips == 1;
        const int stopScore = job.stop;
        const uint8_t* tptr = a.residues + job.tOff;
        const uint8_t* qptr = a.query + job.qOff;
        for (int s = 0; s < nStrips; ++s) {
            const int i = s * kLanes + lane;
            const bool rowActive = i < Q;
            const int qres = rowActive ? qptr[(int64_t)i * job.qStep] : 0;
            const int* srow = smat + qres * kMatStride;
            // Lane state. hLeft = H[i][j-1] doubles as the value handed to the row below (it is read
            // by wave_shr before this step overwrites it, i.e. as H[i][j-1] = H of the row above at
            // the reader's column); before a lane starts it holds the left border H[i][-1], which
            // is exactly the diagonal / upper value the row below needs for its first column.
            int hLeft = leftGap ? borderGap(i, open, ext) : 0;  // H[i][-1]
            int eLeft = kNegInf;
            int hDiag = (i == 0) ? 0 : (leftGap ? borderGap(i - 1, open, ext) : 0);  // H[i-1][-1]
            int fCur = kNegInf;
            const int2* bin = a.boundary[(s + 1) & 1] + job.wsOff;
            int2* bout = a.boundary[s & 1] + job.wsOff;
            const bool lastStrip = s + 1 == nStrips;
            uint8_t* dirs = TRACE ? a.dirs + job.dirOff + (size_t)s * nSteps * kLanes : nullptr;

            // per-lane candidate rules, hoisted out of the step loop
            const bool rowIsLast = i == Q - 1;
            const bool candAlways = rowActive && (region == kAllCells || (rowIsLast && region != kLastCell));
            const bool candOnLastCol = rowActive && (region == kLastRowCol || (region == kLastCell && rowIsLast));
            const bool writer = lane == kLanes - 1 && !lastStrip;

            // A step is a chain of dependent operations (a pair is L + 63 steps long, however
            // many wavefronts the chip has), so everything that does not depend on the DP values
            // is taken off that chain:
            //  * target residues travel down the lanes in a shift register that runs ONE step
            //    ahead of the DP, so the substitution score of step k + 1 is fetched from LDS
            //    while step k is computed;
            //  * the values entering lane 0 (next residue, row above the strip) sit in 64-entry
            //    lane buffers that are rotated by one lane per step (wave_rol:1); wave_shr:1
            //    keeps its `old` operand in lane 0, so feeding lane 0 costs no select;
            //  * the top border of the first strip is kept as two running sums (one gap of k + 1
            //    residues, k + 1 one-residue gaps: common.h borderGap) instead of being
            //    multiplied out at every step.
            constexpr int kShr1 = 0x138, kRol1 = 0x134;
            int tres = lane == 0 ? (int)tptr[0] : 0;  // residue of step 0 (lane 0 only)
            int scCur = srow[tres];
            int tbuf = 0, bH = 0, bF = kNegInf;
            int topOne = open, topMany = open;  // cost of the border cell of step k, either way

            // the last strip has min(Q - 64 s, 64) rows: its sweep is that much shorter
            const int rows = min(Q - s * kLanes, kLanes);
            int kLimit = L + rows - 1;  // wave-uniform; shrinks once the known optimum is met
            for (int k0 = 0; k0 < kLimit; k0 += kLanes) {
                {
                    const int kt = k0 + 1 + lane;  // residues of steps k0+1 .. k0+64
                    tbuf = kt < L ? tptr[(int64_t)kt * job.tStep] : 0;
                    if (s > 0) {
                        const int kb = k0 + lane;  // row above the strip, columns k0 .. k0+63
                        const int2 b = kb < L ? bin[kb] : make_int2(0, kNegInf);
                        bH = b.x;
                        bF = b.y;
                    }
                }
                const int kEnd = min(k0 + kLanes, kLimit);
                // The step is a chain of dependent operations on ONE wavefront: every taken branch is a
                // bubble nothing else fills. The loop is compiled per kind of strip (first: border sums;
                // later: the row above from the lane buffers) and with / without the stop rule, the floor
                // is a max with 0 or INT32_MIN, the candidate test is mask arithmetic.
                auto steps = [&](auto firstC, auto stopC) {
                    constexpr bool kFirst = decltype(firstC)::value, kStop = decltype(stopC)::value;
                    for (int k = k0; k < kEnd; ++k) {
                        // residue stage of step k + 1
                        const int tnext = __builtin_amdgcn_update_dpp(tbuf, tres, kShr1, 0xf, 0xf, false);
                        tbuf = __builtin_amdgcn_update_dpp(tbuf, tbuf, kRol1, 0xf, 0xf, false);
                        const int scNext = srow[tnext];
                        tres = tnext;

                        // DP stage of step k
                        int hTop = bH, fTop = bF;
                        if constexpr (kFirst) {
                            hTop = topGap ? -min(topOne, topMany) : 0;
                            fTop = kNegInf;
                            topOne += ext;
                            topMany += open;
                        } else {
                            bH = __builtin_amdgcn_update_dpp(bH, bH, kRol1, 0xf, 0xf, false);
                            bF = __builtin_amdgcn_update_dpp(bF, bF, kRol1, 0xf, 0xf, false);
                        }
                        const int hUp = __builtin_amdgcn_update_dpp(hTop, hLeft, kShr1, 0xf, 0xf, false);
                        const int fUp = __builtin_amdgcn_update_dpp(fTop, fCur, kShr1, 0xf, 0xf, false);
                        const int j = k - lane;
                        const int eOpen = hLeft - open, eExt = eLeft - ext;
                        const int fOpen = hUp - open, fExt = fUp - ext;
                        const int e = max(eOpen, eExt);
                        const int f = max(fOpen, fExt);
                        const int d = hDiag + scCur;
                        const int h = max(max(d, hFloor), max(e, f));
                        if (TRACE) {
                            // priority diag > E (target gap) > F (query gap); inside a gap,
                            // closing it (back to H) is preferred to extending it
                            const int which = (h == d) ? 0 : (h == e) ? 1 : 2;
                            dirs[(size_t)k * kLanes + lane] =
                                (uint8_t)(which | (e == eOpen ? 4 : 0) | (f == fOpen ? 8 : 0));
                        }
                        // a lane that has not reached its first column keeps its borders; past its
                        // last column nothing reads its state any more
                        const bool started = j >= 0;
                        hDiag = hUp;
                        hLeft = started ? h : hLeft;
                        eLeft = started ? e : eLeft;
                        fCur = started ? f : fCur;
                        // candidates: inside a strip a lane's columns come in order, but the lane's row
                        // of a later strip may tie with an earlier strip's best at a smaller column;
                        // ties between lanes are settled at the end
                        const bool inside = (unsigned)j < (unsigned)L;
                        const bool cand = candAlways | (candOnLastCol & (j == L - 1));
                        const bool improves = (h > best) | ((h == best) & (j < bj));
                        const bool take = inside & cand & improves;
                        best = take ? h : best;
                        bi = take ? i : bi;
                        bj = take ? j : bj;
                        if (writer & inside) bout[j] = make_int2(h, f);
                        scCur = scNext;
                        if constexpr (kStop) {
                            // First maximum of the column-major scan = the first column holding the
                            // (known) optimum: once a lane meets it in column c, only the steps that
                            // complete columns <= c can still change the answer.
                            const bool hit = take & (h == stopScore);
                            if (__builtin_amdgcn_ballot_w64(hit)) {
                                int c = hit ? j : INT32_MAX;
#pragma unroll
                                for (int off = 32; off > 0; off >>= 1) c = min(c, __shfl_xor(c, off));
                                kLimit = min(kLimit, __builtin_amdgcn_readfirstlane(c) + rows);
                            }
                            if (k + 1 >= kLimit) break;
                        }
                    }
                };
                if (s == 0) {
                    if (stopEnabled) steps(std::true_type{}, std::true_type{});
                    else steps(std::true_type{}, std::false_type{});
                } else {
                    steps(std::false_type{}, std::false_type{});   // (the stop rule is for pairs of one strip)
                }
            }
            if (!lastStrip) __threadfence();
        }
        // wave reduction: highest score, then smallest column, then smallest row
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int os = __shfl_xor(best, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
            const bool mineEmpty = bi < 0, otherEmpty = oi < 0;
            bool take;
            if (otherEmpty) take = false;
            else if (mineEmpty) take = !floor0 || os > best;
            else take = better(os, oj, oi, best, bj, bi);
            if (take) {
                best = os;
                bi = oi;
                bj = oj;
            }
        }
    } else if (!floor0) {
        // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
        best = 0;
        if (Q > 0) best = leftGap ? borderGap(Q - 1, open, ext) : 0;
        if (L > 0) best = topGap ? borderGap(L - 1, open, ext) : 0;
    }
    if (lane == 0) {
        a.score[job.out] = best;
        if (a.endI) a.endI[job.out] = bi;
        if (a.endJ) a.endJ[job.out] = bj;
    }
}

// ---- pairs of one strip, two columns a step (round 4) ------------------------------------------
// intraseq_kernel advances one anti-diagonal a step: ~33 instructions on ONE wavefront, a chain of L + Q steps of
// ~350 cycles (0.15 us) that nothing overlaps - the 8000-residue target of a log-normal database is 1.2 ms on the
// side stream, whatever the packed launch beside it does, and the pace of NW / OV searches of such databases
// (profiles/r04_skewed_skip_shares.txt). A step is bound by instruction issue, not by the depth of its
// dependences: the shuffles that hand a row's values to the row below, the residue stage, the loop - paid per
// step whatever the step computes. Here a lane computes TWO columns of its row per step (lane i, step k: columns
// 2 (k - i) and 2 (k - i) + 1; what the row above computed one step earlier), so a pair is a chain of L / 2 + Q
// steps of ~1.4 x the work: scores (and, with LOC, end locations: the same first maximum in column-major order)
// of forward or reversed jobs of one strip, every border rule and answer region, no stop rule, no direction
// bytes - the side jobs of one-strip searches and small searches.
template <bool LOC>
__global__ __launch_bounds__(kJobsPerBlock * kLanes) void intraseq_wide_kernel(IntraseqArgs a) {
    __shared__ int smat[kMaxAlphabet * kMatStride];
    const int A = a.alphabet;
    for (int idx = threadIdx.x; idx < A * A; idx += blockDim.x)
        smat[(idx / A) * kMatStride + (idx % A)] = a.matrix[idx];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int jobIdx = blockIdx.x * kJobsPerBlock + wave;
    if (jobIdx >= a.nJobs) return;  // wave-uniform, after the only barrier
    const PairJob job = a.jobs[jobIdx];
    const int Q = job.qLen, L = job.tLen;   // Q <= 64 (launchIntraseq)
    if (a.raisePriority) __builtin_amdgcn_s_setprio(3);
    const bool topGap = job.rules & 1, leftGap = job.rules & 2, floor0 = job.rules & 4;
    const int hFloor = floor0 ? 0 : INT32_MIN;
    const int region = (job.rules >> 4) & 3;
    const int open = a.gapOpen, ext = a.gapExt;
    int best = floor0 ? 0 : INT32_MIN;
    int bi = -1, bj = -1;
    if (Q > 0 && L > 0) {
        const uint8_t* tptr = a.residues + job.tOff;
        const int i = lane;
        const bool rowActive = i < Q;
        const int qres = rowActive ? a.query[job.qOff + (int64_t)i * job.qStep] : 0;
        const int* srow = smat + qres * kMatStride;
        int hLeft = leftGap ? borderGap(i, open, ext) : 0;                               // H[i][-1]
        int eLeft = kNegInf;
        int hDiag = (i == 0) ? 0 : (leftGap ? borderGap(i - 1, open, ext) : 0);          // H[i-1][-1]
        int hA = 0, fA = kNegInf, hB = 0, fB = kNegInf;   // H, F of the step's two columns, as the row below reads them
        const bool rowIsLast = i == Q - 1;
        const bool candAlways = rowActive && (region == kAllCells || (rowIsLast && region != kLastCell));
        const bool candOnLastCol = rowActive && (region == kLastRowCol || (region == kLastCell && rowIsLast));
        constexpr int kShr1 = 0x138, kRol1 = 0x134;
        auto pairAt = [&](int p) -> int {   // residues of columns 2 p and 2 p + 1, one byte each (0 beyond the target)
            const int j = 2 * p;
            const int r0 = j < L ? (int)tptr[(int64_t)j * job.tStep] : 0;
            const int r1 = j + 1 < L ? (int)tptr[(int64_t)(j + 1) * job.tStep] : 0;
            return r0 | (r1 << 8);
        };
        int tres = lane == 0 ? pairAt(0) : 0;   // the residue pair of step 0 (lane 0 only)
        int scA = srow[tres & 0xff], scB = srow[(tres >> 8) & 0xff];
        int tbuf = 0;
        // top border of the two columns of step k, either way of paying for it (common.h borderGap)
        int topOne = open, topMany = open;
        const int nSteps = (L + 1) / 2 + Q - 1;   // wave-uniform
        auto candidate = [&](int h, int j) {
            const bool inside = (unsigned)j < (unsigned)L;
            const bool cand = candAlways | (candOnLastCol & (j == L - 1));
            if constexpr (LOC) {
                const bool improves = (h > best) | ((h == best) & (j < bj));
                const bool take = inside & cand & improves;
                best = take ? h : best;
                bi = take ? i : bi;
                bj = take ? j : bj;
            } else {
                best = max(best, (inside & cand) ? h : INT32_MIN);
            }
        };
        for (int k0 = 0; k0 < nSteps; k0 += kLanes) {
            tbuf = pairAt(k0 + 1 + lane);   // residue pairs of steps k0 + 1 .. k0 + 64
            const int kEnd = min(k0 + kLanes, nSteps);
            for (int k = k0; k < kEnd; ++k) {
                // residue stage of step k + 1
                const int tnext = __builtin_amdgcn_update_dpp(tbuf, tres, kShr1, 0xf, 0xf, false);
                tbuf = __builtin_amdgcn_update_dpp(tbuf, tbuf, kRol1, 0xf, 0xf, false);
                const int scNextA = srow[tnext & 0xff], scNextB = srow[(tnext >> 8) & 0xff];
                tres = tnext;
                // the row above (lane 0: the top border of columns 2 k and 2 k + 1)
                const int hTop0 = topGap ? -min(topOne, topMany) : 0;
                const int hTop1 = topGap ? -min(topOne + ext, topMany + open) : 0;
                topOne += 2 * ext;
                topMany += 2 * open;
                const int hUp0 = __builtin_amdgcn_update_dpp(hTop0, hA, kShr1, 0xf, 0xf, false);
                const int fUp0 = __builtin_amdgcn_update_dpp(kNegInf, fA, kShr1, 0xf, 0xf, false);
                const int hUp1 = __builtin_amdgcn_update_dpp(hTop1, hB, kShr1, 0xf, 0xf, false);
                const int fUp1 = __builtin_amdgcn_update_dpp(kNegInf, fB, kShr1, 0xf, 0xf, false);
                const int j0 = 2 * (k - lane);
                // first column of the step
                const int e0 = max(hLeft - open, eLeft - ext);
                const int f0 = max(hUp0 - open, fUp0 - ext);
                const int h0 = max(max(hDiag + scA, hFloor), max(e0, f0));
                // ... and the second, one cell to the right
                const int e1 = max(h0 - open, e0 - ext);
                const int f1 = max(hUp1 - open, fUp1 - ext);
                const int h1 = max(max(hUp0 + scB, hFloor), max(e1, f1));
                hA = h0; fA = f0; hB = h1; fB = f1;
                // a lane that has not reached its first column keeps its borders
                const bool started = j0 >= 0;
                hLeft = started ? h1 : hLeft;
                eLeft = started ? e1 : eLeft;
                hDiag = started ? hUp1 : hDiag;
                candidate(h0, j0);
                candidate(h1, j0 + 1);
                scA = scNextA;
                scB = scNextB;
            }
        }
        // wave reduction: highest score, then smallest column, then smallest row
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int os = __shfl_xor(best, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
            if constexpr (LOC) {
                const bool mineEmpty = bi < 0, otherEmpty = oi < 0;
                bool take;
                if (otherEmpty) take = false;
                else if (mineEmpty) take = !floor0 || os > best;
                else take = better(os, oj, oi, best, bj, bi);
                if (take) {
                    best = os;
                    bi = oi;
                    bj = oj;
                }
            } else {
                best = max(best, os);
            }
        }
    } else if (!floor0) {
        // degenerate pair: closed forms of the border (oracle/opal_oracle.c, dp_pass)
        best = 0;
        if (Q > 0) best = leftGap ? borderGap(Q - 1, open, ext) : 0;
        if (L > 0) best = topGap ? borderGap(L - 1, open, ext) : 0;
    }
    if (lane == 0) {
        a.score[job.out] = best;
        if (LOC && a.endI) a.endI[job.out] = bi;
        if (LOC && a.endJ) a.endJ[job.out] = bj;
    }
}

// ---- long pairs, strip-parallel --------------------------------------------------------------
// intraseq_kernel sweeps the strips of a pair one after the other: a chain of strips x (L + 63)
// dependent steps of ~600 cycles each, however many wavefronts the chip has. The 35 000-residue
// target of BASELINE configs[3] against the 2000-residue query is 32 strips: 280 ms for one pair,
// five times the rest of the search. Here a wavefront owns ONE (pair, strip) unit, taken from a
// counter pair-major, and the strips of a pair run side by side: strip s follows strip s - 1 through
// the boundary row in HBM, block of 64 columns by block, behind a progress counter - the scheme of
// interseq_pair_strips_kernel (relaxed agent-scope atomics for the rows, which cross XCDs; every
// taken unit's producer was taken before it, strip 0 waits for nobody). The chain becomes L + 63
// steps plus two blocks of lag per strip. Each unit leaves (score, row, column) of its strip;
// merge_strip_partials_kernel folds them with the tie-breaks of the wave reduction. Scores and end
// locations only (no direction bytes), pairs of one common number of strips.
constexpr int kStripWaitCap = 1 << 22;   // x s_sleep 4: about two seconds, then the unit gives up (a.error)

// (launched as workgroups of 4 wavefronts, or of 16 beside a persistent packed launch: see launchIntraseqStrips)
__global__ __launch_bounds__(1024) void intraseq_strips_kernel(IntraseqArgs a) {
    __shared__ int smat[kMaxAlphabet * kMatStride];
    const int A = a.alphabet;
    for (int idx = threadIdx.x; idx < A * A; idx += blockDim.x)
        smat[(idx / A) * kMatStride + (idx % A)] = a.matrix[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int unit = 0;
    if (lane == 0) unit = atomicAdd(a.stripCounter, 1);
    unit = __builtin_amdgcn_readfirstlane(unit);
    const int nStrips = a.nStrips;
    if (unit >= a.nJobs * nStrips) return;
    const int jobIdx = unit / nStrips, s = unit - jobIdx * nStrips;
    const PairJob job = a.jobs[jobIdx];
    const int Q = job.qLen, L = job.tLen;
    int4* partial = a.stripPartial + unit;
    int* progOut = a.stripProgress + unit;
    const int* progIn = progOut - 1;
    if (s * kLanes >= Q || L <= 0) {
        // (a pair with fewer strips than the launch's, or an empty target: nothing in this strip)
        if (lane == 0) {
            *partial = make_int4(0, -1, -1, 0);
            __hip_atomic_store(progOut, L > 0 ? L : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (unit + 1 == a.faultUnit1) {
        // (test hook, host.hip miopalTestInjectFault: this unit publishes nothing, the strip below it
        // runs into its time-out)
        if (lane == 0) *partial = make_int4(0, -1, -1, 0);
        return;
    }
    const int waitCap = a.stripWaitCap > 0 ? a.stripWaitCap : kStripWaitCap;
    if (a.raisePriority) __builtin_amdgcn_s_setprio(3);
    const bool topGap = job.rules & 1, leftGap = job.rules & 2, floor0 = job.rules & 4;
    const int hFloor = floor0 ? 0 : INT32_MIN;
    const int region = (job.rules >> 4) & 3;
    const int open = a.gapOpen, ext = a.gapExt;
    const uint8_t* tptr = a.residues + job.tOff;
    const uint8_t* qptr = a.query + job.qOff;
    const int jobStrips = (Q + kLanes - 1) / kLanes;
    const int i = s * kLanes + lane;
    const bool rowActive = i < Q;
    const int qres = rowActive ? qptr[(int64_t)i * job.qStep] : 0;
    const int* srow = smat + qres * kMatStride;
    int hLeft = leftGap ? borderGap(i, open, ext) : 0;  // H[i][-1]
    int eLeft = kNegInf;
    int hDiag = (i == 0) ? 0 : (leftGap ? borderGap(i - 1, open, ext) : 0);  // H[i-1][-1]
    int fCur = kNegInf;
    unsigned long long* bin = reinterpret_cast<unsigned long long*>(a.boundary[(s + 1) & 1] + job.wsOff);
    unsigned long long* bout = reinterpret_cast<unsigned long long*>(a.boundary[s & 1] + job.wsOff);
    const bool lastStrip = s + 1 == jobStrips;
    const bool rowIsLast = i == Q - 1;
    const bool candAlways = rowActive && (region == kAllCells || (rowIsLast && region != kLastCell));
    const bool candOnLastCol = rowActive && (region == kLastRowCol || (region == kLastCell && rowIsLast));
    const bool writer = lane == kLanes - 1 && !lastStrip;
    int best = floor0 ? 0 : INT32_MIN, bi = -1, bj = -1;

    constexpr int kShr1 = 0x138, kRol1 = 0x134;
    int tres = lane == 0 ? (int)tptr[0] : 0;
    int scCur = srow[tres];
    int tbuf = 0, bH = 0, bF = kNegInf;
    int topOne = open, topMany = open;
    const int rows = min(Q - s * kLanes, kLanes);
    const int kLimit = L + rows - 1;
    int avail = s == 0 ? L : 0;   // columns of the row above known to be published
    bool dead = false;
    for (int k0 = 0; k0 < kLimit && !dead; k0 += kLanes) {
        {
            const int kt = k0 + 1 + lane;
            tbuf = kt < L ? tptr[(int64_t)kt * job.tStep] : 0;
            if (s > 0) {
                const int need = min(k0 + kLanes, L);
                int spins = 0;
                while (avail < need) {
                    avail = stripPoll(progIn);
                    if (avail >= need) break;
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > waitCap) {
                        dead = true;
                        break;
                    }
                }
                if (dead) break;
                const int kb = k0 + lane;
                if (kb < L) {
                    const unsigned long long v = __hip_atomic_load(bin + kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bH = (int)(uint32_t)v;
                    bF = (int)(uint32_t)(v >> 32);
                } else {
                    bH = 0;
                    bF = kNegInf;
                }
            }
        }
        const int kEnd = min(k0 + kLanes, kLimit);
        // (compiled per kind of strip, mask arithmetic for the candidates: see intraseq_kernel)
        auto steps = [&](auto firstC) {
            constexpr bool kFirst = decltype(firstC)::value;
            for (int k = k0; k < kEnd; ++k) {
                const int tnext = __builtin_amdgcn_update_dpp(tbuf, tres, kShr1, 0xf, 0xf, false);
                tbuf = __builtin_amdgcn_update_dpp(tbuf, tbuf, kRol1, 0xf, 0xf, false);
                const int scNext = srow[tnext];
                tres = tnext;
                int hTop = bH, fTop = bF;
                if constexpr (kFirst) {
                    hTop = topGap ? -min(topOne, topMany) : 0;
                    fTop = kNegInf;
                    topOne += ext;
                    topMany += open;
                } else {
                    bH = __builtin_amdgcn_update_dpp(bH, bH, kRol1, 0xf, 0xf, false);
                    bF = __builtin_amdgcn_update_dpp(bF, bF, kRol1, 0xf, 0xf, false);
                }
                const int hUp = __builtin_amdgcn_update_dpp(hTop, hLeft, kShr1, 0xf, 0xf, false);
                const int fUp = __builtin_amdgcn_update_dpp(fTop, fCur, kShr1, 0xf, 0xf, false);
                const int j = k - lane;
                const int eOpen = hLeft - open, eExt = eLeft - ext;
                const int fOpen = hUp - open, fExt = fUp - ext;
                const int e = max(eOpen, eExt);
                const int f = max(fOpen, fExt);
                const int d = hDiag + scCur;
                const int h = max(max(d, hFloor), max(e, f));
                const bool started = j >= 0;
                hDiag = hUp;
                hLeft = started ? h : hLeft;
                eLeft = started ? e : eLeft;
                fCur = started ? f : fCur;
                const bool inside = (unsigned)j < (unsigned)L;
                const bool cand = candAlways | (candOnLastCol & (j == L - 1));
                const bool improves = (h > best) | ((h == best) & (j < bj));
                const bool take = inside & cand & improves;
                best = take ? h : best;
                bi = take ? i : bi;
                bj = take ? j : bj;
                if (writer & inside)
                    __hip_atomic_store(bout + j, ((unsigned long long)(uint32_t)f << 32) | (uint32_t)h, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                scCur = scNext;
            }
        };
        if (s == 0) steps(std::true_type{});
        else steps(std::false_type{});
        if (!lastStrip) {
            // columns written so far by lane 63: j = k - 63 for the steps done
            // (every row store acknowledged before the counter moves: stripPublish, common.h)
            const int done = min(max(kEnd - (kLanes - 1), 0), L);
            stripPublish(progOut, done, lane);
        }
    }
    if (dead) {
        // (never seen: the strip above was taken before this one and waits for nobody below it)
        if (lane == 0) {
            atomicAdd(a.error, 1);
            *partial = make_int4(0, -1, -1, 0);
            __hip_atomic_store(progOut, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // wave reduction: highest score, then smallest column, then smallest row
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int os = __shfl_xor(best, off), oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
        const bool mineEmpty = bi < 0, otherEmpty = oi < 0;
        bool take;
        if (otherEmpty) take = false;
        else if (mineEmpty) take = !floor0 || os > best;
        else take = better(os, oj, oi, best, bj, bi);
        if (take) {
            best = os;
            bi = oi;
            bj = oj;
        }
    }
    if (lane == 0) *partial = make_int4(best, bi, bj, 0);
}

// One thread per pair: the strips' answers folded in strip order with the rule of the wave reduction.
__global__ void merge_strip_partials_kernel(IntraseqArgs a) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.nJobs) return;
    const PairJob job = a.jobs[k];
    const bool floor0 = job.rules & 4;
    int best = floor0 ? 0 : INT32_MIN, bi = -1, bj = -1;
    const bool degenerate = job.qLen <= 0 || job.tLen <= 0;
    if (degenerate && !floor0) {
        // closed forms of the border (oracle/opal_oracle.c, dp_pass), as in intraseq_kernel
        const bool topGap = job.rules & 1, leftGap = job.rules & 2;
        best = 0;
        if (job.qLen > 0) best = leftGap ? borderGap(job.qLen - 1, a.gapOpen, a.gapExt) : 0;
        if (job.tLen > 0) best = topGap ? borderGap(job.tLen - 1, a.gapOpen, a.gapExt) : 0;
    }
    for (int s = 0; s < a.nStrips && !degenerate; ++s) {
        const int4 p = a.stripPartial[(size_t)k * a.nStrips + s];
        const bool mineEmpty = bi < 0, otherEmpty = p.y < 0;
        bool take;
        if (otherEmpty) take = false;
        else if (mineEmpty) take = !floor0 || p.x > best;
        else take = better(p.x, p.z, p.y, best, bj, bi);
        if (take) {
            best = p.x;
            bi = p.y;
            bj = p.z;
        }
    }
    a.score[job.out] = best;
    if (a.endI) a.endI[job.out] = bi;
    if (a.endJ) a.endJ[job.out] = bj;
}

// One thread per pair: walk the direction bytes back from the end cell. Every memory access of
// a step is divergent (64 lanes, 64 lines), so the step is kept to ONE such access, the
// direction byte: the query sits in LDS, the target residues are fetched four at a time, and
// with fixed slots the operations leave as whole dwords.
constexpr int kWalkQueryLds = 4096;

__global__ __launch_bounds__(64) void walk_kernel(WalkArgs a) {
    __shared__ uint8_t qlds[kWalkQueryLds];
    const bool queryInLds = a.queryLength <= kWalkQueryLds;
    if (queryInLds)
        for (int x = threadIdx.x; x < a.queryLength; x += 64) qlds[x] = a.query[x];
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.nJobs) return;
    const PairJob job = a.jobs[idx];
    const int n = job.qLen, m = job.tLen;
    const int nSteps = m + kLanes - 1;
    // hybrid direction pass: the head of the sorted list was done by intraseq_kernel
    const bool head = a.headWaves != nullptr && idx < *a.headWaves * kLanes;
    const bool laneMajor = a.dirWaveStride > 0 && !head;
    const uint8_t* dirs = head ? a.headDirs + (int64_t)idx * a.headDirStride
                        : laneMajor ? a.dirs + (int64_t)(idx >> 6) * a.dirWaveStride + (idx & 63)
                                    : a.dirs + job.dirOff;
    // (perpair_profile_kernel's bit planes are walked by walk_planes_kernel below; this kernel then only takes the
    // wavefronts at the head of a hybrid batch, whose directions have intraseq_kernel's layout)
    if (laneMajor && a.dirPlanes) return;
    const uint8_t* q = a.query + job.qOff;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(a.residues);  // hipMalloc'ed: aligned
    const int slot = a.slotByOut ? job.out : idx;
    uint8_t* ops = a.opsOff ? a.ops + a.opsOff[slot] : a.ops + (int64_t)slot * a.opsSlot;
    int64_t pos = a.opsOff ? a.opsOff[slot + 1] - a.opsOff[slot] : a.opsSlot;
    const bool dwords = !a.opsOff && (a.opsSlot & 3) == 0;  // slots start and end on dword boundaries
    uint32_t acc = 0, tword = 0;
    int64_t twordAt = -1;
    int i = n - 1, j = m - 1, state = 0, len = 0;
    auto emit = [&](uint32_t op) {
        ++len;
        --pos;
        if (dwords) {
            acc = (acc << 8) | op;  // the newest operation has the lowest address
            if ((pos & 3) == 0) *reinterpret_cast<uint32_t*>(ops + pos) = acc;
        } else {
            ops[pos] = (uint8_t)op;
        }
    };
    while (i >= 0 || j >= 0) {
        if (i < 0) {
            for (; j >= 0; --j) emit(2);
            break;
        }
        if (j < 0) {
            for (; i >= 0; --i) emit(1);
            break;
        }
        const int l = i & 63;
        uint8_t d;
        {
            // perpair_kernel's layout holds two rows per byte ([strip][column][row pair][lane])
            d = laneMajor
                ? (uint8_t)((dirs[(((int64_t)(i >> 6) * a.dirStripColumns + j) * (kLanes / 2) + (l >> 1)) * kLanes] >> ((l & 1) * 4)) & 0xf)
                : dirs[((size_t)(i >> 6) * nSteps + (j + l)) * kLanes + l];
        }
        if (state == 0) {
            const int c = d & 3;
            if (c == 0) {
                const int64_t at = job.tOff + j;  // forward jobs only (tStep = 1)
                if ((at >> 2) != twordAt) {
                    twordAt = at >> 2;
                    tword = words[twordAt];
                }
                const uint32_t tr = (tword >> ((at & 3) * 8)) & 0xffu;
                const uint32_t qr = queryInLds ? qlds[job.qOff + i] : q[i];
                emit(qr == tr ? 0 : 3);
                --i; --j;
            } else {
                state = c;
            }
        } else if (state == 1) {
            emit(2);
            if (d & 4) state = 0;
            --j;
        } else {
            emit(1);
            if (d & 8) state = 0;
            --i;
        }
    }
    if (dwords && (pos & 3)) {
        // the lowest dword is only partly filled: its top bytes go out one by one
        const int fill = 4 - (int)(pos & 3);
        for (int x = 0; x < fill; ++x) ops[pos + x] = (uint8_t)(acc >> (8 * x));
    }
    a.opsLen[slot] = len;
}

// The same walk over perpair_profile_kernel's bit planes (round 4). The planes come in 64-byte lines: the four
// flags (came from the diagonal / from E / E was opened / F was opened) of FOUR columns x 32 rows of one pair -
// [pair / 64][strip][column / 4][rows 0-31 | 32-63][pair % 64][plane][column % 4] dwords. A lane's path needs the
// lines one after the other, each after a round trip to HBM; what round 3 did - a load per step, a different line
// every step, one dword used of each - cost a memory latency per operation and moved 50 x the bytes the walk
// needs (profiles/r03_pmc_cfg3full_walk.json). Here the wavefront alternates between two phases: every lane
// that is not finished fetches the line its path is on (and the eight target residues around it) into LDS - ONE
// round trip for the whole wavefront - then all lanes step through what they hold, a few cycles a step, until
// none can go on. A lane that leaves its line early waits for the others; the wavefront pays a latency per line of
// its longest path, not per step of it.
//
// Round 5: the steps themselves. Round 4's loop branched on the kind of step (border / diagonal / probe / either gap,
// each with its own copy of the 16-byte flush): with 64 lanes on 64 different paths the wavefront ran EVERY arm, ~165
// VALU instructions and ~45 branches per step (profiles/r05a_pmc_cfg3full_walk_planes_kernel.json: 7.0e7 instructions
// per 250k pairs), and at four wavefronts per SIMD that made the walk as much VALU- as latency-bound. Now a step is one
// straight block under one exec mask: the four flags of the cell come with ONE ds_read_b128 (COLMAJOR lines hold a
// column's four planes side by side), "off the diagonal: E or F?" is decided in the step that finds it out (no probe
// step), everything is a select, the operations collect in a 128-bit shift register (four v_alignbit per step) and
// leave 16 bytes at a time.
template <bool COLMAJOR>
__global__ __launch_bounds__(64) void walk_planes_kernel(WalkArgs a) {
    // (the query in dynamic LDS, sized by its length: a fixed 4 KB of it cost the CU wavefronts it has registers for)
    extern __shared__ __attribute__((aligned(16))) uint8_t qlds[];
    // COLMAJOR: [column % 4][lane] x the four planes; else [plane * 4 + column % 4][lane] dwords (no two lanes share a bank)
    // TWO lines per lane: the one the path is on and its left neighbour (the next one of a path that keeps to its half
    // strip) - a wavefront pays a round trip per line of its longest path, and with the steps cheap that latency is the walk
    __shared__ uint4 lineLds[8 * 64];
    for (int x = threadIdx.x; x < a.queryLength; x += 64) qlds[x] = a.query[x];   // (launchWalk: the query fits)
    __syncthreads();
    const int lane = threadIdx.x;
    const int idx = blockIdx.x * 64 + lane;
    // hybrid direction pass: the wavefronts at the head of the sorted list were done by intraseq_kernel (walk_kernel)
    if (a.headWaves != nullptr && (int)blockIdx.x < *a.headWaves) return;
    const bool live = idx < a.nJobs;
    PairJob job{};
    if (live) job = a.jobs[idx];
    const int n = job.qLen, m = job.tLen;
    const uint32_t* planeBase = reinterpret_cast<const uint32_t*>(a.dirs + (int64_t)(idx >> 6) * a.dirWaveStride) + lane * 16;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(a.residues);  // hipMalloc'ed: aligned
    const int slot = a.slotByOut ? job.out : idx;
    uint8_t* ops = a.ops + (int64_t)slot * a.opsSlot;     // (launchWalk: fixed slots of whole 16-byte pieces, < 2 GB)
    int pos = (int)a.opsSlot;
    // the last sixteen operations, the newest in the lowest byte of w0: what ops[pos .. pos + 16) holds or will hold
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    int i = n - 1, j = m - 1, state = 0, len = 0;   // state: 0 = H, 1 = in a gap of the query (E), 2 = of the target (F)
    // the lines in LDS cover rows iLo .. iLo + 31 of columns jBase .. jBase + 7, the residues columns resLo .. resLo + 15
    // (resLo <= max(jBase, 0)); a path only moves up and to the left, so "still on the lines" is i >= iLo && j >= jBase
    int iLo = INT32_MAX, jBase = INT32_MAX, resLo = INT32_MAX;   // (jBase: column of slot 0, four below the line's own)
    uint32_t resBase = 0;     // dword of the residue block's first column, counted from the target's first dword
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;   // the target residues of sixteen columns (five dwords: any alignment)
    const uint32_t t3 = (uint32_t)(job.tOff & 3);
    const int qBase = job.qOff;
    const uint32_t* const lineWords = reinterpret_cast<const uint32_t*>(lineLds);
    const int stripTiles = (int)((a.dirStripColumns >> 2) * 2);   // tiles (column block, half) per strip (launchWalk: fits)
    for (;;) {
        // ---- step while any lane holds what its next step reads
        for (;;) {
            const bool inside = (i | j) >= 0;
            const bool walking = live && (i >= 0 || j >= 0);
            const bool can = walking && (!inside || (i >= iLo && j >= jBase));
            if (__builtin_amdgcn_ballot_w64(can) == 0) break;
            if (can) {
                const uint32_t ic = (uint32_t)max(i, 0), jc = (uint32_t)max(j, 0);
                uint32_t f0, f1, f2, f3;
                const uint32_t slot = (uint32_t)((int)jc - jBase) & 7u;
                if (COLMAJOR) {
                    const uint4 p = lineLds[slot * 64 + lane];
                    f0 = p.x; f1 = p.y; f2 = p.z; f3 = p.w;
                } else {
                    // plane P of line L (0: the left one): slot L * 4 + P, component column % 4
                    const uint32_t* at = lineWords + ((slot & 4u) * 64 + lane) * 4 + (slot & 3u);
                    f0 = at[0]; f1 = at[64 * 4]; f2 = at[2 * 64 * 4]; f3 = at[3 * 64 * 4];
                }
                const uint32_t sh = 31u - (ic & 31u);
                const bool fromDiag = (f0 >> sh) & 1u, fromE = (f1 >> sh) & 1u;
                const bool openedE = (f2 >> sh) & 1u, openedF = (f3 >> sh) & 1u;
                // residue of column j: byte (tOff + j) of the database, out of the five dwords held
                const uint32_t rel = ((t3 + jc) >> 2) - resBase;
                const uint32_t rw = rel == 0 ? r0 : rel == 1 ? r1 : rel == 2 ? r2 : rel == 3 ? r3 : r4;
                const uint32_t tr = (rw >> (((t3 + jc) & 3u) * 8u)) & 0xffu;
                const uint32_t qr = qlds[qBase + (int)ic];
                const bool diag = inside && state == 0 && fromDiag;
                // the gap this step is in: the one it was in, the one the cell says it came from, or - on a border -
                // the only one left (i < 0: the rest of the target against nothing)
                int gap = state == 0 ? (fromE ? 1 : 2) : state;
                gap = inside ? gap : (i < 0 ? 1 : 2);
                const bool inE = gap == 1;
                const uint32_t op = diag ? (qr == tr ? 0u : 3u) : (inE ? 2u : 1u);
                const bool opened = inE ? openedE : openedF;
                state = (diag || (inside && opened)) ? 0 : gap;
                i -= (diag || !inE) ? 1 : 0;
                j -= (diag || inE) ? 1 : 0;
                ++len;
                --pos;
                w3 = __builtin_amdgcn_alignbit(w3, w2, 24);
                w2 = __builtin_amdgcn_alignbit(w2, w1, 24);
                w1 = __builtin_amdgcn_alignbit(w1, w0, 24);
                w0 = (w0 << 8) | op;
                if ((pos & 15) == 0) *reinterpret_cast<uint4*>(ops + pos) = make_uint4(w0, w1, w2, w3);
            }
        }
        // ---- one round trip: every unfinished lane fetches the line of its current cell (and, every fourth
        // time or so, the residues of the next sixteen columns)
        const bool walking = live && i >= 0 && j >= 0;
        if (__builtin_amdgcn_ballot_w64(walking) == 0) break;
        if (walking) {
            // (every walking lane is off its lines here. The residues held always reach down to the lines' first column:
            // sixteen columns that end with the line's own four)
            const bool newLine = true, newRes = max((j & ~3) - 4, 0) < resLo;
            uint4 p0 = make_uint4(0, 0, 0, 0), p1 = p0, p2 = p0, p3 = p0, l0 = p0, l1 = p0, l2 = p0, l3 = p0;
            if (newLine) {
                const int tile = (int)__umul24((uint32_t)(i >> 6), (uint32_t)stripTiles) + (j >> 2) * 2 + ((i >> 5) & 1);
                const uint4* src = reinterpret_cast<const uint4*>(planeBase + (int64_t)tile * (kLanes * 16));
                p0 = src[0]; p1 = src[1]; p2 = src[2]; p3 = src[3];
                if (j >= 4) {
                    // the same rows of the four columns to the left: two tiles back (tiles alternate between the halves)
                    const uint4* left = src - 2 * (kLanes * 4);
                    l0 = left[0]; l1 = left[1]; l2 = left[2]; l3 = left[3];
                }
            }
            if (newRes) {
                resLo = max((j & ~3) - 12, 0);
                const int64_t first = (job.tOff + resLo) >> 2;
                r0 = words[first]; r1 = words[first + 1]; r2 = words[first + 2]; r3 = words[first + 3]; r4 = words[first + 4];
                resBase = (t3 + (uint32_t)resLo) >> 2;
            }
            if (newLine) {
                iLo = i & ~31;
                jBase = (j & ~3) - 4;
                // (either order of a line's sixteen dwords: four of them side by side per slot)
                lineLds[0 * 64 + lane] = l0; lineLds[1 * 64 + lane] = l1; lineLds[2 * 64 + lane] = l2; lineLds[3 * 64 + lane] = l3;
                lineLds[4 * 64 + lane] = p0; lineLds[5 * 64 + lane] = p1; lineLds[6 * 64 + lane] = p2; lineLds[7 * 64 + lane] = p3;
            }
        }
    }
    if (live) {
        // what has not left yet: the operations from pos up to the next 16-byte boundary, the lowest bytes of the register
        const int rest = (16 - (pos & 15)) & 15;
        for (int x = 0; x < rest; ++x) {
            const uint32_t w = x < 4 ? w0 : x < 8 ? w1 : x < 12 ? w2 : w3;
            ops[pos + x] = (uint8_t)(w >> (8 * (x & 3)));
        }
        a.opsLen[slot] = len;
    }
}

// Jobs of the start-location pass, built where the end locations already are (HBM):
// reversed prefixes q[0..endQ], t[0..endT] anchored on the end cell. Targets without an
// end cell get an empty job (its outputs are ignored by the host).
__global__ void reverse_jobs_kernel(int n, const int32_t* score, const int32_t* endQ, const int32_t* endT,
                                    const int64_t* offsets, int rules, int64_t wsStride, PairJob* jobs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int qe = endQ[k], te = endT[k];
    PairJob j{};
    j.out = k;
    j.rules = rules;
    j.tStep = -1;
    j.qStep = -1;
    j.wsOff = (int64_t)k * wsStride;  // strip-boundary workspace (queries of more than 64 rows)
    if (score) {
        // the reversed problem has the same optimum as the forward one: stop at its first column
        j.rules |= kRuleStop;
        j.stop = score[k];
    }
    if (qe >= 0 && te >= 0) {
        j.tOff = offsets[k] + te;
        j.tLen = te + 1;
        j.qOff = qe;
        j.qLen = qe + 1;
    }
    jobs[k] = j;
}

// ---- full-mode pipeline kept in HBM (queries that fit one 64-row strip) --------------
// Start cells from the reversed-prefix pass, including the degenerate optimum of HW / OV
// (oracle/opal_oracle.c): one gap over one sequence only = a border cell of the reversed
// problem. Slots whose reverse score disagrees raise *mismatch.
__global__ void start_cells_kernel(int n, int mode, int open, int ext, const int32_t* score,
                                   const int32_t* endQ, const int32_t* endT, const int32_t* rScore,
                                   const int32_t* rI, const int32_t* rJ, int32_t* startQ,
                                   int32_t* startT, int* mismatch) {
    // mismatch[0]: 1 + index of a slot whose reverse pass disagrees; mismatch[1], [2]: longest
    // target window and tallest query window of the slice (they size the direction and
    // operation slots of the traceback)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    int window = 0, rows = 0;
    const bool live = k < n;
    const int qe = live ? endQ[k] : -1, te = live ? endT[k] : -1;
    int sq = -1, st = -1;
    if (qe >= 0 && te >= 0) {
        if (mode == 0 /* NW */) {
            sq = 0;
            st = 0;
        } else {
            int ri = rI[k], rj = rJ[k];
            if (rScore[k] != score[k] || ri < 0 || rj < 0) {
                if (mode != 3 /* SW */ && score[k] == borderGap(qe, open, ext)) {
                    ri = qe;
                    rj = -1;
                } else if (mode == 2 /* OV */ && score[k] == borderGap(te, open, ext)) {
                    ri = -1;
                    rj = te;
                } else {
                    atomicExch(mismatch, k + 1);
                    ri = qe;
                    rj = te;
                }
            }
            sq = qe - ri;
            st = te - rj;
        }
        window = te - st + 1;
        rows = qe - sq + 1;
    }
    if (live) {
        startQ[k] = sq;
        startT[k] = st;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        window = max(window, __shfl_xor(window, off));
        rows = max(rows, __shfl_xor(rows, off));
    }
    // one pair of atomics per workgroup, results not waited for (a volatile read of the running maxima by every
    // wavefront first - to skip the atomic - made 31k uncached reads of one line: 0.16 ms for 1M targets)
    __shared__ int blockWindow, blockRows;
    if (threadIdx.x == 0) blockWindow = blockRows = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&blockWindow, window);
        atomicMax(&blockRows, rows);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockWindow > 0) atomicMax(&mismatch[1], blockWindow);
        if (blockRows > 0) atomicMax(&mismatch[2], blockRows);
    }
}

// Traceback jobs on the [start..end] rectangles; job k owns direction slot k.
__global__ void trace_jobs_kernel(int n, int rules, const int32_t* startQ, const int32_t* startT,
                                  const int32_t* endQ, const int32_t* endT, const int64_t* offsets,
                                  int64_t dirStride, int64_t wsStride, PairJob* jobs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    PairJob j{};
    j.out = k;
    j.rules = rules;
    j.tStep = 1;
    j.qStep = 1;
    j.dirOff = (int64_t)k * dirStride;
    j.wsOff = (int64_t)k * wsStride;  // strip-boundary workspace (windows of more than 64 rows)
    if (endQ[k] >= 0 && endT[k] >= 0) {
        j.tOff = offsets[k] + startT[k];
        j.tLen = endT[k] - startT[k] + 1;
        j.qOff = startQ[k];
        j.qLen = endQ[k] - startQ[k] + 1;
    }
    jobs[k] = j;
}

// Pack the operations (written from the back of fixed-size slots) into one buffer, in slice
// order. The destination of slot k is *base + (sum of lens[0..k)): block sums first, then each
// block adds up the sums before it and scans its own 256 lengths in LDS. *next receives the
// running total for the following batch - the host never has to see the lengths in between.
constexpr int kGatherBlock = 256;

__global__ __launch_bounds__(kGatherBlock) void ops_block_sums_kernel(int n, const int32_t* lens,
                                                                      int64_t* blockSums) {
    __shared__ int red[kGatherBlock];
    const int t = threadIdx.x, k = blockIdx.x * kGatherBlock + t;
    red[t] = k < n ? lens[k] : 0;
    __syncthreads();
    for (int off = kGatherBlock / 2; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    if (t == 0) blockSums[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(kGatherBlock) void gather_ops_kernel(int n, const uint8_t* slots, int64_t slotBytes,
                                                                  const int32_t* lens, const int64_t* blockSums,
                                                                  const int64_t* base, int64_t* next,
                                                                  uint8_t* out) {
    __shared__ int64_t red[kGatherBlock];
    __shared__ int scan[kGatherBlock];
    const int t = threadIdx.x, k = blockIdx.x * kGatherBlock + t;
    int64_t before = 0;
    for (int b = t; b < (int)blockIdx.x; b += kGatherBlock) before += blockSums[b];
    red[t] = before;
    const int len = k < n ? lens[k] : 0;
    scan[t] = len;
    __syncthreads();
    for (int off = kGatherBlock / 2; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    for (int off = 1; off < kGatherBlock; off <<= 1) {
        const int v = t >= off ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    // the wavefront copies its 64 alignments one after the other, 64 bytes per step: both
    // sides of the copy are contiguous runs
    const int64_t dstOff = *base + red[0] + scan[t] - len;
    if (k == n - 1) *next = dstOff + len;
    const int64_t srcOff = (int64_t)(k + 1) * slotBytes - len;
    const int lane = t & 63;
#pragma unroll 1
    for (int p = 0; p < 64; ++p) {
        const int plen = __shfl(len, p);
        if (plen == 0) continue;  // wave-uniform
        const uint8_t* src = slots + __shfl(srcOff, p);
        uint8_t* dst = out + __shfl(dstOff, p);
        for (int i = lane; i < plen; i += 64) dst[i] = src[i];
    }
}

// ---- counting sort of pair jobs by target-window length, longest first ---------------
// perpair_kernel runs a wavefront until its longest pair is done: 64 neighbours of similar
// length waste few columns, and the longest wavefronts start first.
constexpr int kSortBlock = 1024;
constexpr int kSortMaxBins = 8192;  // 32 KB of LDS

// (keys are lengths >> shift, so that windows of any length sort with at most 8192 bins)
// Round 5: a second, minor key - the window's ROWS in groups of eight. A direction wavefront sweeps the rows of its
// tallest window times the columns of its longest for all of its pairs; by length alone 82 % (cfg3) to 59 % (short
// alignments) of the swept cells belong to a pair, with the rows as a minor key 89 % to 76 %
// (profiles/r05_sweep_efficiency.txt). SortKey{shift, G, rowShift}: key = (tLen >> shift) * G + rows-of-eight >> rowShift;
// G = 1 is the sort by length alone. maxKey is the largest key.
struct SortKey {
    int shift, groups, rowShift;
    __device__ __forceinline__ int of(const PairJob& j, int maxKey) const {
        const int lenKey = j.tLen >> shift;
        if (groups == 1) return min(lenKey, maxKey);
        const int rows = min(groups - 1, ((j.qLen + 7) >> 3) >> rowShift);
        return min(lenKey * groups + rows, maxKey);
    }
};

__global__ __launch_bounds__(kSortBlock) void job_length_histogram_kernel(const PairJob* jobs, int n, int maxLen,
                                                                          SortKey sk, int* bins) {
    extern __shared__ int local[];
    for (int b = threadIdx.x; b <= maxLen; b += kSortBlock) local[b] = 0;
    __syncthreads();
    for (int k = blockIdx.x * kSortBlock + threadIdx.x; k < n; k += gridDim.x * kSortBlock)
        atomicAdd(&local[sk.of(jobs[k], maxLen)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b <= maxLen; b += kSortBlock)
        if (local[b]) atomicAdd(&bins[b], local[b]);
}

// bins[b] <- first position of length b in the sorted order (lengths descending); one block
__global__ __launch_bounds__(kSortBlock) void job_length_offsets_kernel(int maxLen, int* bins, int n, SortKey sk,
                                                                        int* headWaves, int maxHeadWaves) {
    __shared__ int partial[kSortBlock];
    const int nBins = maxLen + 1;
    const int per = (nBins + kSortBlock - 1) / kSortBlock;
    // thread t owns the bins of rank [t * per, (t + 1) * per) counted from the longest
    int sum = 0;
    for (int r = threadIdx.x * per; r < min(nBins, (threadIdx.x + 1) * per); ++r) sum += bins[maxLen - r];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kSortBlock; off <<= 1) {
        const int v = (int)threadIdx.x >= off ? partial[threadIdx.x - off] : 0;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    int run = partial[threadIdx.x] - sum;
    // bins[b] becomes the number of jobs with a key above b (non-increasing in b). The 90th percentile,
    // the smallest key with at most 10 % of the jobs above it, is found on the way: every thread
    // looks at the bins it rewrites (one thread walking the bins through global memory took 0.3 ms).
    __shared__ int p90Shared;
    if (threadIdx.x == 0) p90Shared = maxLen;
    __syncthreads();
    int mine = maxLen;
    for (int r = threadIdx.x * per; r < min(nBins, (threadIdx.x + 1) * per); ++r) {
        const int b = maxLen - r;
        const int c = bins[b];
        bins[b] = run;
        if (run <= n / 10) mine = min(mine, b);
        run += c;
    }
    if (headWaves != nullptr) {
        // Outliers: keys above twice the 90th percentile (and above 64 residues).
        // p90 = the smallest key b with bins[b] <= n / 10, i.e. at most 10 % of the jobs above it
        if (mine < maxLen) atomicMin(&p90Shared, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            // (in length keys: the rows are the minor key)
            const int p90 = p90Shared / sk.groups;
            const int floorKey = 64 >> sk.shift;
            const int longKey = (max(2 * p90, floorKey) + 1) * sk.groups;      // outliers have key >= longKey
            int count = 0;
            if (longKey <= maxLen) count = longKey >= 1 ? bins[longKey - 1] : n;
            *headWaves = min((count + kLanes - 1) / kLanes, maxHeadWaves);
        }
    }
}

// Scatter: a block ranks its jobs per length in LDS and reserves one range per (block, length)
// with a single global atomic, instead of one contended atomic per job. The 56-byte jobs are then copied
// dword by dword with neighbouring lanes on neighbouring dwords of the same job (a thread writing its own
// job made 64 separate partial-line writes of every store instruction: 0.2 ms per 250k jobs).
constexpr int kJobWords = sizeof(PairJob) / 4;
static_assert(sizeof(PairJob) % 4 == 0, "PairJob is copied as dwords");

__global__ __launch_bounds__(kSortBlock) void job_length_scatter_kernel(const PairJob* jobs, int n, int maxLen,
                                                                        SortKey sk, int* bins, PairJob* sorted) {
    extern __shared__ int local[];
    __shared__ int dest[kSortBlock];
    for (int b = threadIdx.x; b <= maxLen; b += kSortBlock) local[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortBlock;
    const int64_t k = base + threadIdx.x;
    int key = 0, rank = 0;
    if (k < n) {
        key = sk.of(jobs[k], maxLen);
        rank = atomicAdd(&local[key], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= maxLen; b += kSortBlock) {
        const int c = local[b];
        if (c) local[b] = atomicAdd(&bins[b], c);
    }
    __syncthreads();
    dest[threadIdx.x] = local[key] + rank;
    __syncthreads();
    const int here = (int)min<int64_t>(kSortBlock, n - base);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(jobs + base);
    uint32_t* dst = reinterpret_cast<uint32_t*>(sorted);
    for (int t = threadIdx.x; t < here * kJobWords; t += kSortBlock) {
        const int job = t / kJobWords, word = t - job * kJobWords;
        dst[(int64_t)dest[job] * kJobWords + word] = src[t];
    }
}

// ---- the same sort for bare indices (round 5): order[] = 0 .. n - 1 by keys[] descending ---------------------
// The persistent start-cell scan takes its pairs in this order, longest PREFIXES first: what it cannot know is how far
// a scan runs, but it never runs beyond its prefix - with the short prefixes last, the stragglers every wavefront
// finishes alone when the list has run out are short ones (simulated on cfg3's own windows: the last wavefront done at
// column 212 instead of 264, tools/r05_scan_schedule_sim.py).
__global__ __launch_bounds__(kSortBlock) void key_histogram_kernel(const int32_t* keys, int n, int maxKey, int shift, int* bins) {
    extern __shared__ int local[];
    for (int b = threadIdx.x; b <= maxKey; b += kSortBlock) local[b] = 0;
    __syncthreads();
    for (int k = blockIdx.x * kSortBlock + threadIdx.x; k < n; k += gridDim.x * kSortBlock)
        atomicAdd(&local[min(max(keys[k], 0) >> shift, maxKey)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b <= maxKey; b += kSortBlock)
        if (local[b]) atomicAdd(&bins[b], local[b]);
}

__global__ __launch_bounds__(kSortBlock) void key_scatter_kernel(const int32_t* keys, int n, int maxKey, int shift, int* bins,
                                                                 int* order) {
    extern __shared__ int local[];
    for (int b = threadIdx.x; b <= maxKey; b += kSortBlock) local[b] = 0;
    __syncthreads();
    const int64_t k = (int64_t)blockIdx.x * kSortBlock + threadIdx.x;
    int key = 0, rank = 0;
    if (k < n) {
        key = min(max(keys[k], 0) >> shift, maxKey);
        rank = atomicAdd(&local[key], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= maxKey; b += kSortBlock) {
        const int c = local[b];
        if (c) local[b] = atomicAdd(&bins[b], c);
    }
    __syncthreads();
    if (k < n) order[local[key] + rank] = (int)k;
}

hipError_t launchSortIndicesByKey(const int32_t* keys, int n, int maxValue, int* bins, int* order, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    SortKey sk{0, 1, 0};
    while ((maxValue >> sk.shift) >= kSortMaxBins) ++sk.shift;
    const int maxKey = std::max(maxValue, 0) >> sk.shift;
    hipError_t e = hipMemsetAsync(bins, 0, (size_t)(maxKey + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int blocks = std::min((n + kSortBlock - 1) / kSortBlock, 1024);
    hipLaunchKernelGGL(key_histogram_kernel, dim3(blocks), dim3(kSortBlock), (size_t)(maxKey + 1) * sizeof(int), stream, keys, n,
                       maxKey, sk.shift, bins);
    hipLaunchKernelGGL(job_length_offsets_kernel, dim3(1), dim3(kSortBlock), 0, stream, maxKey, bins, n, sk, (int*)nullptr, 0);
    hipLaunchKernelGGL(key_scatter_kernel, dim3((n + kSortBlock - 1) / kSortBlock), dim3(kSortBlock),
                       (size_t)(maxKey + 1) * sizeof(int), stream, keys, n, maxKey, sk.shift, bins, order);
    return hipGetLastError();
}

hipError_t launchSortJobsByLength(const PairJob* jobs, int n, int maxLen, int* bins, PairJob* sorted,
                                  hipStream_t stream, int* headWaves, int maxHeadWaves, int queryRows) {
    if (n <= 0) return hipSuccess;
    SortKey sk{0, 1, 0};
    if (queryRows > 8) {
        // rows of eight as the minor key; lengths in steps of four (the columns a wavefront sweeps are)
        sk.groups = (queryRows + 7) / 8 + 1;
        while (sk.groups > 64) { sk.groups = (sk.groups + 1) / 2; ++sk.rowShift; }
        sk.shift = 2;
    }
    while (((int64_t)(maxLen >> sk.shift) + 1) * sk.groups > kSortMaxBins) ++sk.shift;
    const int maxKey = (maxLen >> sk.shift) * sk.groups + sk.groups - 1;
    hipError_t e = hipMemsetAsync(bins, 0, (size_t)(maxKey + 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int blocks = std::min((n + kSortBlock - 1) / kSortBlock, 1024);
    hipLaunchKernelGGL(job_length_histogram_kernel, dim3(blocks), dim3(kSortBlock),
                       (size_t)(maxKey + 1) * sizeof(int), stream, jobs, n, maxKey, sk, bins);
    hipLaunchKernelGGL(job_length_offsets_kernel, dim3(1), dim3(kSortBlock), 0, stream, maxKey, bins, n, sk,
                       headWaves, maxHeadWaves);
    hipLaunchKernelGGL(job_length_scatter_kernel, dim3((n + kSortBlock - 1) / kSortBlock), dim3(kSortBlock),
                       (size_t)(maxKey + 1) * sizeof(int), stream, jobs, n, maxKey, sk, bins, sorted);
    return hipGetLastError();
}

hipError_t launchStartCells(int n, int mode, int open, int ext, const int32_t* score, const int32_t* endQ,
                            const int32_t* endT, const int32_t* rScore, const int32_t* rI, const int32_t* rJ,
                            int32_t* startQ, int32_t* startT, int* mismatch, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(start_cells_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, mode, open, ext, score,
                       endQ, endT, rScore, rI, rJ, startQ, startT, mismatch);
    return hipGetLastError();
}

hipError_t launchTraceJobs(int n, int rules, const int32_t* startQ, const int32_t* startT, const int32_t* endQ,
                           const int32_t* endT, const int64_t* offsets, int64_t dirStride, int64_t wsStride,
                           PairJob* jobs, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(trace_jobs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, rules, startQ, startT,
                       endQ, endT, offsets, dirStride, wsStride, jobs);
    return hipGetLastError();
}

hipError_t launchGatherOps(int n, const uint8_t* slots, int64_t slotBytes, const int32_t* lens,
                           int64_t* blockSums, const int64_t* base, int64_t* next, uint8_t* out,
                           hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int blocks = (n + kGatherBlock - 1) / kGatherBlock;
    hipLaunchKernelGGL(ops_block_sums_kernel, dim3(blocks), dim3(kGatherBlock), 0, stream, n, lens, blockSums);
    hipLaunchKernelGGL(gather_ops_kernel, dim3(blocks), dim3(kGatherBlock), 0, stream, n, slots, slotBytes, lens,
                       blockSums, base, next, out);
    return hipGetLastError();
}

hipError_t launchReverseJobs(int n, const int32_t* score, const int32_t* endQ, const int32_t* endT,
                             const int64_t* offsets, int rules, int64_t wsStride, PairJob* jobs,
                             hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(reverse_jobs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, score, endQ, endT,
                       offsets, rules, wsStride, jobs);
    return hipGetLastError();
}

hipError_t launchIntraseq(const IntraseqArgs& a, bool trace, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    const int blocks = (a.nJobs + kJobsPerBlock - 1) / kJobsPerBlock;
    if (a.wide && !trace && !a.headWaves) {
        // (the host has checked its jobs: one strip each, no stop rule)
        if (a.endI || a.endJ)
            hipLaunchKernelGGL((intraseq_wide_kernel<true>), dim3(blocks), dim3(kJobsPerBlock * kLanes), 0, stream, a);
        else
            hipLaunchKernelGGL((intraseq_wide_kernel<false>), dim3(blocks), dim3(kJobsPerBlock * kLanes), 0, stream, a);
        return hipGetLastError();
    }
    if (trace)
        hipLaunchKernelGGL((intraseq_kernel<true>), dim3(blocks), dim3(kJobsPerBlock * kLanes), 0, stream, a);
    else
        hipLaunchKernelGGL((intraseq_kernel<false>), dim3(blocks), dim3(kJobsPerBlock * kLanes), 0, stream, a);
    return hipGetLastError();
}

hipError_t launchIntraseqStrips(const IntraseqArgs& a, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    if (a.nStrips < 2 || !a.stripCounter || !a.stripProgress || !a.stripPartial || !a.error || !a.boundary[0] || !a.boundary[1])
        return hipErrorInvalidValue;
    const int64_t units = (int64_t)a.nJobs * a.nStrips;
    if (units > INT32_MAX / 2) return hipErrorInvalidValue;
    // Beside a persistent packed launch (one workgroup per CU holding every register of it) the units come
    // as workgroups of 16 wavefronts: a packed workgroup cannot start on a CU that hosts even one of these
    // wavefronts, and 1056 of them in workgroups of 4 sit on every CU of the chip for the length of the
    // longest chain (cfg4 with its tail: the packed launch started 7 ms late); in workgroups of 16 they
    // occupy a quarter of the CUs, the packed workgroups of the others start at once and take the units.
    const int perBlock = a.fatBlocks ? 16 : kJobsPerBlock;
    const int blocks = (int)((units + perBlock - 1) / perBlock);
    hipLaunchKernelGGL(intraseq_strips_kernel, dim3(blocks), dim3(perBlock * kLanes), 0, stream, a);
    hipLaunchKernelGGL(merge_strip_partials_kernel, dim3((a.nJobs + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launchWalk(const WalkArgs& a, hipStream_t stream) {
    if (a.nJobs <= 0) return hipSuccess;
    if (a.dirPlanes) {
        // (what the query-profile form of the direction pass guarantees: host_full.inc)
        if (a.dirWaveStride <= 0 || (a.dirStripColumns & 3) || a.opsOff || (a.opsSlot & 15) || a.queryLength > kWalkQueryLds)
            return hipErrorInvalidValue;
        // (tiles and slot offsets as 32-bit numbers in the kernel)
        if (a.opsSlot >= (1ll << 31) || (a.dirStripColumns >> 1) * ((a.queryLength + 63) / 64 + 1) >= (1ll << 31) ||
            (a.dirStripColumns >> 1) >= (1 << 24))
            return hipErrorInvalidValue;
        const dim3 grid((a.nJobs + 63) / 64), block(64);
        const size_t lds = (size_t)((a.queryLength + 15) & ~15);
        if (a.dirColumnMajor) hipLaunchKernelGGL(walk_planes_kernel<true>, grid, block, lds, stream, a);
        else hipLaunchKernelGGL(walk_planes_kernel<false>, grid, block, lds, stream, a);
        if (!a.headWaves) return hipGetLastError();
    }
    hipLaunchKernelGGL(walk_kernel, dim3((a.nJobs + 63) / 64), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace miopal
