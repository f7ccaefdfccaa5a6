// Tuning switches of the library: read from the environment ONCE (first use), kept in a process-wide
// table of atomics, changed afterwards only through miopalSetTuning (include/miopal.h). Nothing on the
// search path calls getenv: the C ABI is called without the GIL from many threads
// (src/pyopal/lib.pyx:1364) and getenv racing another thread's putenv is a data race in the C library.
//
// Every switch is MIOPAL_<NAME> in the environment. Switches are diagnostics and A/B levers; the two
// product knobs among them (CUs kept out of the persistent launch for a collective, the routing of small
// searches) are also per-handle options (miopalDbSetOption), which take precedence.
#pragma once
#include <atomic>

namespace miopal {

#define MIOPAL_TUNING_SWITCHES(X)                                                                          \
    X(ALWAYS_SKIP) X(BATCH_GROUPS) X(DEVICE) X(FIXED_DIRECT_LIMIT) X(FORCE_LANE_PER_PAIR) X(HOST_THREADS)  \
    X(HOST_TRACEBACK) X(LOOSE_REACH) X(NO_BIASED) X(NO_CALLER_PINNED) X(NO_DEFERRED_RESULTS)               \
    X(NO_DIAG_SHIFT) X(NO_DIRECT_SCATTER) X(NO_GLOBAL_STRIPS) X(NO_HOST_SCATTER) X(NO_HUGEPAGE)            \
    X(NO_HYBRID_TRACE) X(NO_OPS_OVERLAP) X(NO_PAIR_STRIPS) X(NO_PAIR_STRIP_UNITS) X(NO_PAIR_TABLE)         \
    X(NO_PERPAIR) X(NO_PERPAIR_PROFILE) X(NO_PRIORITY) X(NO_SCAN_REFILL) X(NO_SEGMENTS) X(NO_SIDE_STREAM)  \
    X(NO_SKIM) X(NO_SMALL_SEARCH) X(NO_SW_SHIFT) X(NO_TWO_PASS_ENDS) X(NO_UNSIGNED_DIAG)                   \
    X(NO_VIEW_PREFETCH) X(PACKED_FIRST) X(PAIR_STRIPS) X(PHASE_TIMING) X(RESERVE_CUS) X(RUNTIME_COPY)      \
    X(SCAN_BLOCKS_PER_CU) X(SCAN_REFILL_LANES) X(SHORT_STRIDE) X(SMALL_STEPS) X(SPARE_HANDLE_MB)           \
    X(STRIPS) X(STRIPS_RESERVE) X(TAIL_THROTTLE) X(THIN_SIDE) X(TWO_PASS_ENDS) X(UNITS)                    \
    X(UPLOAD_PIECE_KB) X(UPLOAD_STREAMS) X(UPLOAD_THREADS) X(VERBOSE) X(VIEW_CACHE_MB)                     \
    X(WINDOWS_WHENEVER_POSSIBLE) X(TEST_REFUSE_PAIR_LAUNCH) X(NO_PACKED_OPS) X(STRIP_TIMING) X(SKIP_SHARES) X(SEARCH_UNDER_UPLOAD) X(NO_WIDE_PAIRS) X(NO_SIDE_COPIES) X(NO_EARLY_HOST_SHARE) X(NO_UNPACK_CREW) X(NO_STREAM_STORES) X(NO_PACKED_TRACE) X(NO_PACKED_SCAN) X(NO_ONE_LAUNCH) X(ONE_LAUNCH_GROUP) X(NO_ASYNC_SHARES) X(NO_LATE_RESULT_COPIES) X(PARKED_WORKSPACE_MB) X(NO_SORT_BY_ROWS) X(NO_JOBS_AHEAD) X(NO_SCAN_ORDER) X(NO_PACKED_HW_SCAN)

enum class Tune : int {
#define X(name) name,
    MIOPAL_TUNING_SWITCHES(X)
#undef X
    kCount
};

// The value of a switch (a NUL-terminated string that lives for the rest of the process), or null when it
// is not set - what getenv("MIOPAL_<NAME>") returned when the table was filled, unless changed since.
const char* tuned(Tune key);

}  // namespace miopal
