// Shared declarations between the host scheduler and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miopal {

constexpr int kLanes = 64;
constexpr int kGroupTargets = 128;  // one wavefront owns 64 lanes x 2 packed 16-bit targets
constexpr int kMaxAlphabet = 32;    // src/pyopal/lib.pxd:28-32
constexpr int kMaxStripRows = 64;   // query rows held in registers per pass

// Border / answer rules of one DP pass (see oracle/opal_oracle.c for the model).
enum Region : int { kLastCell = 0, kLastRow = 1, kLastRowCol = 2, kAllCells = 3 };
struct DpRules {
    int topGap;   // H[-1][j] = -(open + j*ext) instead of 0
    int leftGap;  // H[i][-1] = -(open + i*ext) instead of 0
    int floor0;   // Smith-Waterman floor
    int region;   // Region
};

// ---- inter-sequence kernel (one lane = two targets, packed int16) ---------
struct InterseqArgs {
    const uint2* pack;         // [group][chunk][lane] -> {4 residues of A, 4 residues of B}
    const int64_t* groupOff;   // first uint2 of each group
    const int* groupChunks;    // 4-column chunks per group
    int nGroups;               // groups to process, starting at groupBase
    int groupBase;             // leading (longest) groups skipped by this launch
    const int16_t* profile;    // [A+1][Qpad] substitution scores, row A = padding symbol
    int nSymbols;              // A + 1
    int qPad;                  // nStrips * R
    int nStrips;
    int qLen;                  // true query length (rows beyond it are padding)
    int gapOpen, gapExt;
    int topGap, leftGap;       // border rules (DpRules)
    int region;                // Region: where the answer is taken
    const int32_t* lens;       // [nGroups * 128] target lengths, packed-view order (0 = absent)
    int32_t* score;            // [nGroups * 128], packed-view order
    int32_t* endI;             // LOC kernels: query coordinate of the answer (-1 = none)
    int32_t* endJ;             // LOC kernels: target coordinate
    uint8_t* overflow;         // [nGroups * 128], 1 = lane reached the flavour's limit (may be null)
    int priorityChunks;        // groups with more chunks than this raise their wave priority
    int* workCounter;          // zeroed before launch: next group to hand out (persistent kernels)
    // unit mode of the general kernel (scores, several rounds of strips; null: a workgroup per group)
    int* unitCounter;          // zeroed before the launch: next (round, group) unit
    int* unitFlags;            // [nGroups], zeroed: rounds of the group that are complete
    uint2* unitPartial;        // [nGroups][W][64]: {all-cells best, region answer} carried between rounds
    int scoreBias;             // ArithSwU16: K, added to every profile entry and taken off the stored H
    int biasedLimit;           // biased flavours: a best at or above this (true score) is flagged
    int biasedZero;            // global biased kernel: pattern of a true 0 at shift 0 (covers the values below 0)
    int capGroups, capChunks;  // general kernel: the first capGroups groups of the launch sweep at most capChunks chunks
                               // (their longest targets are computed elsewhere: host.hip, lanes that do not fit)
    int tailThrottle;          // > 0: groups are of similar length; groups per SIMD, rounded up (interseq_impl.h)
    unsigned long long* stripKeys;   // strips kernel with end locations: (score, column, row) keys, view order, zeroed
    int* stripAbort;           // strips kernel: lanes flagged so far (first flags only); at stripAbortAt the launch gives up:
    int stripAbortAt;          //   the view is redone by the next rung anyway (it adds the same to *stripGaveUp)
    int* stripGaveUp;          //   = the overflow counter the host reads after the scatter
    int batchGroups;           // strips kernel: groups a workgroup sweeps side by side (1..12; fewer when the groups are few)
    const int32_t* known;      // strips kernel, second pass of an `end` search: every lane half's optimum (view order)
    // one-strip biased Smith-Waterman kernel, scores only: results straight into database order (no view-order
    // array, no scatter kernel); directOut may be pinned host memory (miopalSearch: no D2H copy either)
    int32_t* directOut;        // already offset by - sliceStart: entry directIds[view position]; null: a.score
    int32_t* directEndI;       // with end locations: the same for the query / target coordinate of the answer
    int32_t* directEndJ;
    const int32_t* directIds;  // view position -> database index
    int directN;               // view positions that hold a target
    int stripSpinCap;          // strips kernels: polls (x s_sleep) before a unit gives up on the strip above; 0 = the default
    int faultUnit1;            // strips kernels, test hook: unit (this - 1) behaves as if it had died; 0 = none
    unsigned long long* stripTiming;   // diagnostic builds (-DMIOPAL_STRIP_TIMING=1, interseq_impl.h): six counters; else null
    uint2* boundary[2];        // ping-pong strip boundaries, same indexing as pack*4
    const int64_t* boundaryOff;
};

// ---- intra-sequence kernel (one wavefront = one pair, int32) --------------
struct PairJob {
    int64_t tOff;    // index of the first residue visited in the linear database
    int32_t tLen;
    int32_t tStep;   // +1 forward, -1 reversed prefix
    int32_t qOff;    // index of the first query residue visited
    int32_t qLen;
    int32_t qStep;
    int32_t rules;   // bit0 topGap, bit1 leftGap, bit2 floor0, bits 4..5 region, bit6 stop
    int64_t wsOff;   // int2 elements into the strip-boundary workspace (per buffer)
    int64_t dirOff;  // bytes into the direction workspace (trace only)
    int32_t out;     // result slot
    int32_t stop;    // with rules bit6 (single-strip pairs): the optimum is known to be `stop`,
                     // the scan ends with the first column that reaches it
};
constexpr int kRuleStop = 0x40;

struct IntraseqArgs {
    const PairJob* jobs;
    int nJobs;
    const uint8_t* residues;  // linear database
    const uint8_t* query;
    const int32_t* matrix;    // [A][A]
    int alphabet;
    int gapOpen, gapExt;
    int2* boundary[2];
    uint8_t* dirs;            // direction bytes (trace only)
    int32_t* score;
    int32_t* endI;            // query coordinate of the best cell (pass coordinates)
    int32_t* endJ;            // target coordinate
    int raisePriority;        // run the wavefronts at s_setprio 3 (side-stream launches)
    // Hybrid direction pass: the job list is sorted longest first and this kernel takes only its
    // head, the first *headWaves x 64 jobs (the rest belongs to perpair_kernel); their direction
    // bytes and strip boundaries are then addressed by position, not by job.dirOff / job.wsOff.
    const int* headWaves;     // null: every job, addressed through the job itself
    int64_t headDirStride;    // direction bytes per job
    int64_t headWsStride;     // boundary columns per job
    // intraseq_strips_kernel (long pairs, one wavefront per (pair, strip) unit):
    int nStrips;              // strips of every pair of the launch
    int* stripCounter;        // zeroed: next unit
    int* stripProgress;       // zeroed, [nJobs x nStrips]: columns of the strip's last row published
    int4* stripPartial;       // [nJobs x nStrips]: (score, row, column) of the strip
    int* error;               // incremented by a unit that gave up waiting (never seen)
    int fatBlocks;            // workgroups of 16 wavefronts (beside a persistent packed launch) instead of 4
    int stripWaitCap;         // polls before a unit gives up on the strip above; 0 = the default
    int faultUnit1;           // test hook: unit (this - 1) publishes nothing; 0 = none
    int wide;                 // every job is a pair of one strip without the stop rule: intraseq_wide_kernel (two columns a step)
};

struct WalkArgs {
    const PairJob* jobs;
    int nJobs;
    const uint8_t* residues;
    const uint8_t* query;
    const uint8_t* dirs;
    uint8_t* ops;             // per job: buffer of qLen + tLen bytes, filled from the back
    const int64_t* opsOff;    // [nJobs + 1], or null: job k owns the fixed slot k * opsSlot
    int64_t opsSlot;
    int32_t* opsLen;
    // direction layout: 0 = intraseq_kernel (anti-diagonal major, at job.dirOff),
    // > 0 = perpair_kernel ([strip][j][i / 2][lane], 4 bits per cell, per 64 consecutive jobs, this many bytes apart)
    int64_t dirWaveStride;
    int64_t dirStripColumns;  // columns per strip in the perpair_kernel layout
    // hybrid direction pass: jobs at positions below *headWaves x 64 have intraseq_kernel's layout
    // in `headDirs`, headDirStride bytes per job
    const int* headWaves;
    const uint8_t* headDirs;
    int64_t headDirStride;
    int slotByOut;            // ops slot / opsLen entry = job.out instead of the job's position
    int queryLength;          // whole query (staged in LDS when it fits)
    int dirPlanes;            // lane-major directions are perpair_profile_kernel's bit planes
    int dirColumnMajor;       // ... in lines of [column % 4][plane] dwords (perpair_packed.hip) instead of [plane][column % 4]
};

// perpair_kernel: one lane per (query window, target window) pair of a one-strip query
constexpr int kPerPairTrace = 4;  // beside the Region values: write direction bytes
struct PerPairArgs {
    const PairJob* jobs;      // borders penalised on both sides, no floor (rules bits 0..2 ignored)
    int nJobs;
    const uint8_t* residues;
    const uint8_t* query;
    int queryLength;          // <= 4096 (staged in LDS)
    const int* matrix;
    int alphabet;
    int gapOpen, gapExt;
    int32_t* score;           // by job.out (trace: score of the last cell of the window)
    int32_t* endI;
    int32_t* endJ;
    uint8_t* dirs;            // trace: [job / 64][strip][j][i / 2][job % 64], two rows (4 bits each) per byte
    int64_t dirWaveStride;    // bytes per 64 consecutive jobs (>= strips * dirStripColumns * 2048)
    int64_t dirStripColumns;  // columns reserved per strip (>= longest target window)
    int2* boundary;           // query windows of more than 64 rows: [job / 64][column][job % 64]
    int64_t boundaryStride;   // columns per 64 consecutive jobs (>= longest target window)
    const int* skipWaves;     // hybrid direction pass: the first *skipWaves wavefronts' jobs belong to
                              // intraseq_kernel (null: none)
    // > 0: perpair_profile_kernel (every region of the scan but kLastCell, and kPerPairTrace): bytes per residue row of its
    // query profile in LDS (perPairProfileBytes); every job walks the query the same way (`reversed`), and
    // the directions leave as bit planes: [job / 64][strip][j / 4][rows 0-31 | 32-63][job % 64][plane][j % 4] dwords
    int profileStride;
    int reversed;
    int scanLastRow;          // packed start-cell scans: 0 = any cell answers (SW), 1 = the query's last row (HW), 2 = the pair's own last row or its last column (OV)
    int64_t residueCount;     // bytes at `residues` (the profile kernel reads them four at a time, clamped; >= 4)
    // non-null (with computeUnits): the start-cell scan of a one-strip query by persistent wavefronts whose lanes take
    // the next job when they are done (perpair_scan_refill_kernel); the counter is zero at launch
    int* jobCounter;
    int computeUnits;
    int refillLanes;          // idle lanes of a wavefront that trigger a refill (set by launchPerPair)
    // perpair_packed.hip (two pairs per lane on 16-bit halves): added to every profile entry so that it is an unsigned byte
    int packedBias;
    // perpair_packed.hip, one launch over the sorted lists of several batches (host_full.inc): jobs per batch (a multiple
    // of 64; 0: the list is one batch). job.out is relative to its batch, and skipWaves holds one count per batch.
    int outBatch;
    // perpair_packed_scan_kernel / perpair_packed_scan_strips_kernel (the latter's rows between strips: `boundary`,
    // [job / 128][column][lane] x 8 bytes)
    int packedZero;           // pattern of the value 0 (packedScanFits)
    // The packed scans of a Smith-Waterman search write the START CELLS themselves (no reverse-pass arrays, no
    // start_cells_kernel behind them): startQ / startT by job.out, startChecks = {1 + index of a pair whose scan never met
    // its optimum, longest target window, tallest query window} (zeroed by the host). And the one-strip scan builds its
    // jobs from the end pass's arrays (slice order; jobs == nullptr): no job list in HBM.
    int32_t* startQ;
    int32_t* startT;
    int* startChecks;
    const int32_t* fwdScore;
    const int32_t* fwdEndQ;
    const int32_t* fwdEndT;
    const int64_t* fwdOffsets;
    const int* order;         // ... taken in this order (null: 0, 1, 2, ...): the persistent scan, longest prefixes first
};
hipError_t launchPerPair(const PerPairArgs& a, int mode, hipStream_t stream);
// perpair_packed.hip: the direction pass with two pairs per lane. packedTraceFits says whether it applies (rows /
// columns: the tallest / longest window of the batch, strips and blocks of four rounded up; best: an upper bound of
// any cell) and returns the bias, the profile stride and the LDS bytes to launch with. The planes leave in lines of
// [column % 4][plane] dwords (WalkArgs::dirColumnMajor).
constexpr int kPackedZero = 0x0800;
bool packedTraceFits(int queryLength, int alphabet, int open, int ext, int maxScore, int minScore, int64_t rows,
                     int64_t columns, int64_t best, int* bias, int* stride, size_t* ldsBytes);
hipError_t launchPerPairPackedTrace(const PerPairArgs& a, size_t ldsBytes, hipStream_t stream);
// the start-cell scan of Smith-Waterman searches with two pairs per lane (persistent wavefronts, a.jobCounter zeroed):
// longest = the longest prefix, best = an upper bound of the optimum
bool packedScanFits(int queryLength, int alphabet, int open, int ext, int maxScore, int minScore, int64_t longest,
                    int64_t best, int* bias, int* zero, int* stride, size_t* ldsBytes);
hipError_t launchPerPairPackedScan(const PerPairArgs& a, size_t ldsBytes, hipStream_t stream);
// bytes per residue row of the lane-per-pair kernels' query profile: >= queryLength + 64 + 8, an odd number of dwords
inline int perPairProfileStride(int queryLength) {
    const int dwords = (queryLength + kLanes + 8 + 3) / 4;
    return (dwords | 1) * 4;
}
// LDS bytes of the profile kernel for this query (0: too long for it), and the stride to pass
size_t perPairProfileBytes(int queryLength, int alphabet, int* stride);

struct PackArgs {
    const uint8_t* residues;
    const int64_t* offsets;   // [N + 1] into residues
    const int32_t* ids;       // view position -> database index
    const int32_t* segStart;  // view position -> first residue of its segment (null: whole targets)
    const int32_t* lens;      // view position -> residues of the segment (with segStart)
    int nTargets;             // (virtual) targets in the view
    const int64_t* groupOff;
    const int* groupChunks;
    const int64_t* chunkPrefix;  // [nGroups + 1] running number of chunks
    int nGroups;
    int padSymbol;
    uint2* pack;
};

// ---- launchers (defined next to their kernels) -----------------------------
enum InterseqFlavour : int {
    kSwHalf = 0,              // Smith-Waterman, packed half floats (exact below 2048)
    kSwInt16 = 1,             // Smith-Waterman, saturating int16
    kSignedInt16 = 2,         // NW / HW / OV, signed saturating int16
    kSignedInt16AllCells = 3, // anchored reverse pass: signed, every cell is a candidate
    kSignedInt16Diag = 4,     // NW / HW / OV on anti-diagonally shifted values (6 ops per cell pair)
    kUnsignedDiag = 5,        // the same on unsigned patterns compared as half floats (5 cheaper ops, interseq_impl.h)
    kSwShifted = 6            // Smith-Waterman scores on column-shifted unsigned patterns (ArithSwU16)
};
constexpr int kSwShiftZero = 0x1000;       // = kSwU16Zero
constexpr int kUnsignedDiagZero = 0x1000;  // = kU16Zero
hipError_t launchInterseq(const InterseqArgs& a, int rowsPerStrip, int waves, InterseqFlavour flavour,
                          bool locate, hipStream_t stream);
// pair-indexed LDS profile (single strip, Smith-Waterman); false = table does not fit LDS
bool interseqPairFits(int rowsPerStrip, int nSymbols);
enum PairFlavour : int {
    kPairSwInt16 = 0,   // saturating int16
    kPairSwHalf = 1,    // packed half floats, exact below 2048
    kPairSwBiased = 2,  // biased integer halves compared as half floats, column-shifted (interseq_impl.h)
    kPairGlobalBiased = 3, // NW / HW / OV on the same representation (scores, optional end locations)
    kPairSwStrips = 4,     // Smith-Waterman scores of several strips on biased halves (units of (batch, strip))
    kPairGlobalStrips = 5  // NW / HW / OV of several strips on the same units (scores; end locations through keys)
};
// limits of the biased flavour (host-side range checks; the kernel's constants are in interseq_impl.h)
constexpr int kPairStripsMaxRows = 52, kPairStripsMaxRowsLoc = 48;   // tallest strips of the multi-strip pair-table kernel
constexpr int kPairStripsMaxRowsKnown = 40;   // ... of an `end` search in two sweeps (scores, then the cell that holds them)
constexpr int kBiasedScoreLimit = 25600;   // = kBiasedLimit: a best at or above it is recomputed
constexpr int kBiasedMaxMagnitude = 1024;  // |score|, open - ext, ext - open
constexpr int kBiasedMaxExt = 512;
constexpr int kBiasedMaxStepUp = 0x1000;   // (score + ext) << bits, (ext - open) << bits; above 0x0400 the limit shrinks
constexpr int kBiasedPad = -1024;          // = kBiasedPadScore: padding symbol / rows in the profile
// the biased flavour with end locations (scaled by 2^bits, bits = 4 / 5 / 6 for <= 16 / 32 / 64 rows)
constexpr int kLocZeroPattern = 0x0C00;
constexpr int kLocGuardBand = 0x0800;      // |score| << bits, (open - ext) << bits, (ext - open) << bits
constexpr int kLocMaxShift = 4096;         // 5 * (ext << bits) must fit
inline int locRowBitsHost(int rows) { return rows <= 16 ? 4 : rows <= 32 ? 5 : 6; }
hipError_t launchInterseqPair(const InterseqArgs& a, int rowsPerStrip, PairFlavour flavour, int computeUnits,
                              hipStream_t stream, bool locate = false);
hipError_t launchInterseqPairSwBiasedA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedC(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedD(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwStripsA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwStripsB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwStripsLocA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwStripsLocB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
// second pass of an `end` search: the first cell that holds each target's known optimum (a.known)
hipError_t launchInterseqPairSwStripsKnownA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalStripsA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalStripsB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalStripsLocA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalStripsLocB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
// keys of the multi-strip NW / HW / OV kernel -> view-order scores and end locations (endI / endJ may be null)
hipError_t launchDecodeGlobalKeys(const unsigned long long* keys, const int32_t* lens, int n, int queryLength,
                                  int32_t* score, int32_t* endI, int32_t* endJ, hipStream_t stream);
// (score, column, row) keys of the strips kernel -> view-order scores and end locations
hipError_t launchDecodeStripKeys(const unsigned long long* keys, int n, int32_t* score, int32_t* endI, int32_t* endJ, hipStream_t stream);
hipError_t launchInterseqPairGlobalA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalC(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairGlobalD(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedLocA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedLocB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedLocC(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwBiasedLocD(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwHalf(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream);
hipError_t launchInterseqPairSwInt16(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream);
hipError_t launchInterseqSwHalf(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSwHalfLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSwInt16(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSwInt16Loc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSigned(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSignedLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSignedAll(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSignedDiag(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSignedDiagLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSwShifted(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqUnsignedDiag(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqUnsignedDiagLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchInterseqSignedAllLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream);
hipError_t launchIntraseq(const IntraseqArgs& a, bool trace, hipStream_t stream);
hipError_t launchIntraseqStrips(const IntraseqArgs& a, hipStream_t stream);
hipError_t launchWalk(const WalkArgs& a, hipStream_t stream);
hipError_t launchStartCells(int n, int mode, int open, int ext, const int32_t* score, const int32_t* endQ,
                            const int32_t* endT, const int32_t* rScore, const int32_t* rI, const int32_t* rJ,
                            int32_t* startQ, int32_t* startT, int* mismatch, hipStream_t stream);
hipError_t launchTraceJobs(int n, int rules, const int32_t* startQ, const int32_t* startT, const int32_t* endQ,
                           const int32_t* endT, const int64_t* offsets, int64_t dirStride, int64_t wsStride,
                           PairJob* jobs, hipStream_t stream);
// Counting sort by tLen, longest first (lengths are coarsened so that at most 8192 bins are needed).
// queryRows > 8: the window's rows in groups of eight as a minor key, tallest first (what a direction wavefront
// sweeps is its tallest window times its longest). bins: 8192 ints.
// headWaves (optional): receives how many leading wavefronts of 64 sorted jobs are outliers - more
// than twice as long as the 90th percentile - capped at maxHeadWaves: a lane-per-pair wavefront
// lasts as long as its longest lane, those few go to the wavefront-per-pair kernel instead.
hipError_t launchSortJobsByLength(const PairJob* jobs, int n, int maxLen, int* bins, PairJob* sorted,
                                  hipStream_t stream, int* headWaves = nullptr, int maxHeadWaves = 0,
                                  int queryRows = 0);
// order[] = the indices 0 .. n - 1 by keys[] descending (negative keys count as 0); bins: 8192 ints
hipError_t launchSortIndicesByKey(const int32_t* keys, int n, int maxValue, int* bins, int* order, hipStream_t stream);
// blockSums: (n + 255) / 256 entries of scratch; *base = bytes already in `out`, *next = *base + this batch
hipError_t launchGatherOps(int n, const uint8_t* slots, int64_t slotBytes, const int32_t* lens,
                           int64_t* blockSums, const int64_t* base, int64_t* next, uint8_t* out,
                           hipStream_t stream);
hipError_t launchReverseJobs(int n, const int32_t* score, const int32_t* endQ, const int32_t* endT,
                             const int64_t* offsets, int rules, int64_t wsStride, PairJob* jobs,
                             hipStream_t stream);
hipError_t launchPack(const PackArgs& a, int64_t totalChunks, hipStream_t stream);
// segmented views with end locations (pack.hip): keyed atomicMax per window, then unpack
hipError_t launchScatterKeyed(const int32_t* viewScore, const int32_t* viewEndI, const int32_t* viewEndJ,
                              const uint8_t* viewOverflow, const int32_t* ids, const int32_t* segStart,
                              int nTargets, int64_t sliceStart, unsigned long long* keys, int32_t* overflowCount,
                              hipStream_t stream, int scoreBias = 0);
hipError_t launchDecodeKeys(const unsigned long long* keys, int n, int32_t* score, int32_t* endI, int32_t* endJ,
                            hipStream_t stream, int scoreBias = 0);
hipError_t launchFillInt32(int32_t* out, int n, int32_t value, hipStream_t stream);
// subset of a resident database: dst[dstOff[k] ...] = src[srcStart[k] ...] for every sequence k (dstOff has n + 1 entries)
// device -> pinned host copy by a small kernel of our own (see pack.hip)
hipError_t launchCopyOut(const void* src, void* dst, int64_t bytes, hipStream_t stream);
// units of 64 one-byte operations of `src` -> 16 bytes each of `dst`, two bits per operation (both 16-byte aligned)
hipError_t launchCopyOutPacked(const void* src, void* dst, int64_t firstUnit, int64_t lastUnit, hipStream_t stream);
hipError_t launchGatherSequences(const uint8_t* src, const int64_t* srcStart, const int64_t* dstOff, int64_t n,
                                 uint8_t* dst, hipStream_t stream);
// takeMax: several view positions (segments) may belong to one target; `out` starts at 0
hipError_t launchScatter(const int32_t* viewScore, const uint8_t* viewOverflow, const int32_t* ids,
                         int nTargets, int64_t sliceStart, int32_t* out, int32_t* overflowCount,
                         bool takeMax, hipStream_t stream);
hipError_t launchScatterEnds(const int32_t* viewEndI, const int32_t* viewEndJ, const int32_t* ids,
                             int nTargets, int64_t sliceStart, int32_t* outI, int32_t* outJ,
                             hipStream_t stream);

// Value of a penalised border cell k (0-based) residues into the border: one gap of k + 1
// residues or, when opening is cheaper than extending, k + 1 one-residue gaps (what the
// recurrence does inside the matrix). Same definition as oracle/opal_oracle.c.
__host__ __device__ inline int borderGap(int k, int open, int ext) {
    const long long one = open + (long long)k * ext, many = (long long)(k + 1) * open;
    return -(int)(one < many ? one : many);
}

// ---- hand-over of boundary rows between strips that run in different workgroups (often in
// different XCDs, whose L2 caches do not see each other). Used by interseq_pair_strips_kernel,
// interseq_pair_global_strips_kernel (interseq_impl.h) and intraseq_strips_kernel (intraseq.hip); the
// one place where the pair "fence + s_waitcnt" lives, so that the kernels cannot diverge again.
//
// The scheme sits OUTSIDE the LLVM AMDGPU memory model (relaxed atomics carry no ordering there); it
// relies on what the gfx950 ISA does with them (MI355X_MICROARCH.md, "Valid forms"):
//  * the rows and the progress counter are relaxed AGENT-scope atomics of 4 or 8 bytes: they compile
//    to global_store / global_load with sc1 - a store that is written through to memory (acknowledged,
//    i.e. counted down from vmcnt, only once it is past the local L2) and a load that misses the
//    local L2;
//  * producer: every wavefront that stored rows waits for its OWN stores to be acknowledged
//    (s_waitcnt vmcnt(0)) before it moves the counter. The workgroup-scope release fence in front
//    only keeps the compiler from sinking the stores below; it orders, it does not wait - without the
//    s_waitcnt a counter was seen before its rows under load (one wrong score in twenty cfg4 runs);
//  * consumer: the wavefront that polls loads rows only after its poll matched. The poll's value goes
//    through readfirstlane into a scalar the loop branches on, so the row loads cannot be issued
//    before the poll's data has returned; memory instructions of one wavefront issue in order, and the
//    workgroup-scope acquire fence keeps the compiler from hoisting them. No other wavefront reads
//    rows on the strength of somebody else's poll.
// A stale row would be a silently wrong score (the time-out path only covers a strip that never
// arrives): tests/test_gpu_biased.py and tests/test_gpu_fullsize.py keep one strips-against-general
// comparison each in the GPU tier.
static __device__ __forceinline__ void stripPublish(int* counter, int value, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#ifndef MIOPAL_ABL_NO_PUBLISH_WAIT   // (ablation builds only: wrong, timing)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    if (lane == 0) __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform value of the producer's counter; rows may be loaded once it is large enough
static __device__ __forceinline__ int stripPoll(const int* counter) {
    const int v = __hip_atomic_load(const_cast<int*>(counter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the rows are fetched after the counter)
    return __builtin_amdgcn_readfirstlane(v);
}

inline int packRules(const DpRules& r) {
    return (r.topGap ? 1 : 0) | (r.leftGap ? 2 : 0) | (r.floor0 ? 4 : 0) | (r.region << 4);
}

}  // namespace miopal
