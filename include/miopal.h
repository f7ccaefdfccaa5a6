/*
 * miopal.h -- resident-database extension of the Opal C ABI for MI355X.
 *
 * `opalSearchDatabase` (opal.h; src/pyopal/opal.pxd:38-52) receives N host
 * pointers on every call (the layout of pyopal's Database,
 * src/pyopal/lib.pxd:95-98), which forces a pack + PCIe upload per query. The
 * entry points below are what a GPU-aware binding would call instead: the
 * database is packed and uploaded once, searches run against the device
 * mirror. They replace, one for one:
 *
 *   miopalDbCreate          <- the (sequences, lengths, size) triple that
 *                              BaseDatabase.get_sequences/get_lengths/get_size
 *                              hand to the plugin (src/pyopal/lib.pxd:90-92,
 *                              src/pyopal/platform/pyx.in:54-59)
 *   miopalSearch            <- opalSearchDatabase(...) at
 *                              src/pyopal/platform/pyx.in:77-91, including the
 *                              [start, end) slicing done there by pointer
 *                              offset (pyx.in:80-82)
 *   miopalSearchDeviceScores<- same call for searchType = OPAL_SEARCH_SCORE
 *                              with results left in HBM (multi-GPU shard
 *                              driver, src/pyopal/_align.py:150-170)
 *
 * All functions return 0 or an OPAL_ERR_* / MIOPAL_ERR_* code (opal.h), never
 * throw, never take the Python GIL, and may be called concurrently from many
 * threads on one handle (the reference's callers do exactly that while
 * holding the database read lock: src/pyopal/lib.pyx:1364,
 * src/pyopal/_align.py:150-170). No torch types cross this boundary.
 */
#ifndef MIOPAL_H
#define MIOPAL_H

#include <stdint.h>

#include "opal.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct MiopalDb MiopalDb;

/* Number of usable gfx950 devices (0 when there is none or HIP fails). */
int miopalDeviceCount(void);

/* Human-readable description of the last error raised on this thread. */
const char* miopalLastError(void);

/*
 * Build a device-resident database from N host sequences of ordinals
 * (< alphabetLength <= 32). `device` is the HIP device ordinal. The host
 * buffers are not referenced after the call returns.
 */
int miopalDbCreate(MiopalDb** out, const unsigned char* const* sequences, const int* lengths,
                   int64_t count, int alphabetLength, int device);

/* Same, from one concatenated residue buffer and offsets[count + 1]. */
int miopalDbCreateFlat(MiopalDb** out, const unsigned char* residues, const int64_t* offsets,
                       int64_t count, int alphabetLength, int device);

/*
 * A new handle holding the targets `indices[0 .. count)` of `parent` (any order, repeats allowed), on
 * the parent's device: the residues are gathered from the parent's resident copy on the device, nothing
 * is uploaded. What Database.mask / Database.extract need (src/pyopal/lib.pyx:694-778: subsets that
 * share the parent's sequence buffers) - here the subset owns a device-side copy of its residues, so
 * either handle may be destroyed first. The parent must not be destroyed during the call.
 */
int miopalDbCreateSubset(MiopalDb** out, const MiopalDb* parent, const int64_t* indices, int64_t count);

void miopalDbDestroy(MiopalDb* db);

/* opalSearchDatabase (include/opal.h) receives the whole database on every call; the library keeps
 * the handles such calls built (device memory, pinned bounce buffers, streams: at most 4 handles of at
 * most MIOPAL_SPARE_HANDLE_MB, default 4096, MiB of device memory each) and refills them on the next
 * call. This releases everything kept that way. No reference counterpart.
 */
void miopalReleaseCaches(void);

/*
 * Searches keep their device workspaces (strip boundaries, direction bits and operations of `full`, staging)
 * parked on the handle for the next search of whatever thread comes first - up to 64 GB per handle, beyond which a
 * returning workspace is freed. A process that has run a burst of large `full` searches from many threads and
 * wants the memory back calls this: every idle workspace of the handle is released (running searches keep
 * theirs). Returns the device bytes released. No reference counterpart.
 */
int64_t miopalDbReleaseWorkspaces(MiopalDb* db);

int64_t miopalDbCount(const MiopalDb* db);
int64_t miopalDbTotalLength(const MiopalDb* db);
/* Bytes of HBM held by the mirror (linear copy + packed views). */
int64_t miopalDbDeviceBytes(const MiopalDb* db);

/*
 * One query against targets [start, end) of the database.
 * Host output arrays have end - start entries and may be NULL when the
 * search type does not produce them:
 *   score                                  all search types
 *   endTarget, endQuery                    SCORE_END and ALIGNMENT
 *   startTarget, startQuery                ALIGNMENT
 *   alignment[k] (malloc'ed, caller frees), alignmentLength[k]   ALIGNMENT
 */
int miopalSearch(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                 int gapExt, const int* scoreMatrix, int alphabetLength, int searchType,
                 int mode, int64_t start, int64_t end, int* score, int* endTarget,
                 int* endQuery, int* startTarget, int* startQuery, unsigned char** alignment,
                 int* alignmentLength);

/*
 * Same search with the alignments returned as ONE malloc'ed buffer
 * (*operations, caller frees) and operationOffsets[end - start + 1]: target k's
 * operations are (*operations)[operationOffsets[k] .. operationOffsets[k+1]).
 * This is the bulk form of the per-result `alignment` pointers of
 * OpalSearchResult (src/pyopal/opal.pxd:24-32): one allocation instead of N.
 */
int miopalSearchFlat(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                     int gapExt, const int* scoreMatrix, int alphabetLength, int searchType,
                     int mode, int64_t start, int64_t end, int* score, int* endTarget,
                     int* endQuery, int* startTarget, int* startQuery,
                     unsigned char** operations, int64_t* operationOffsets);

/*
 * miopalSearchFlat with the operations written into a buffer the caller lends.
 * In: *operations is NULL, or a malloc'ed buffer of *operationsCapacity bytes
 * (the one an earlier call returned, its contents no longer needed). Out:
 * *operations holds the operations and *operationsCapacity its size in bytes.
 * "Large enough" is judged against the search's WORST case - targets x (query
 * length + longest alignment window, rounded up to 16), not against the
 * operations it ends up producing; a buffer this library returned carries a
 * quarter of headroom for that reason.
 * When the lent buffer is large enough the SAME pointer comes back, written in
 * place: its pages are resident already, where a fresh buffer of a million
 * alignments is 67-360 MB of first-touch page faults per search (and as much to
 * unmap when it is freed). When it is too small the call returns a buffer of its
 * own and leaves the lent one untouched - still the caller's, to free or to keep.
 * On error both values are as they were. Same per-result `alignment` semantics
 * as above (src/pyopal/opal.pxd:24-32).
 */
int miopalSearchFlatInto(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                         int gapExt, const int* scoreMatrix, int alphabetLength, int searchType,
                         int mode, int64_t start, int64_t end, int* score, int* endTarget,
                         int* endQuery, int* startTarget, int* startQuery,
                         unsigned char** operations, int64_t* operationsCapacity,
                         int64_t* operationOffsets);

/*
 * Score-only search whose int32 results stay in HBM: `deviceScores` is a
 * device pointer with end - start entries (database order), `stream` a
 * hipStream_t (NULL = the null stream). The call only enqueues work; the
 * scores are valid once `stream` has drained. Targets whose 16-bit lanes
 * saturate are recomputed at 32 bit inside the same stream. (Two kinds of
 * search synchronise before they return: one whose lanes may leave the
 * exact range - a 4-byte count comes back - and one that sent long pairs of
 * a multi-strip query to the wavefront-per-pair kernel's strip units, whose
 * give-up counter is read so that a partial answer cannot pass for a score.)
 */
int miopalSearchDeviceScores(MiopalDb* db, const unsigned char* query, int queryLength,
                             int gapOpen, int gapExt, const int* scoreMatrix,
                             int alphabetLength, int mode, int64_t start, int64_t end,
                             int* deviceScores, void* stream);

/*
 * Timing of the dominant kernel of the most recent search on this handle,
 * measured with HIP events on the stream the kernel ran on. Returns the
 * number of launches timed; *ms receives their total duration.
 * Enabled with miopalSetProfiling(db, 1) (adds two event records per launch).
 */
void miopalSetProfiling(MiopalDb* db, int enabled);
int miopalLastKernelTime(MiopalDb* db, float* ms);

/*
 * How the score pass of the calling thread's most recent search was scheduled
 * (diagnostics for tests and benchmarks; no reference counterpart):
 *   counts[0] targets computed by the wavefront-per-pair (int32) kernel
 *   counts[1] kernel of the lane-per-target pass, low four bits: 0 none, 1 general (v_perm profile), 2 / 3 / 4
 *             pair table with int16 / half-float / biased-integer lanes (Smith-Waterman), 5 pair table for
 *             NW / HW / OV, 6 pair table strip by strip (Smith-Waterman scores of several strips), 7 the same for
 *             NW / HW / OV; + 16 when the
 *             pair-table launch was refused and the general kernel ran instead;
 *             + 32 x the general kernel's lane arithmetic (0 half floats, 1 int16, 2 signed int16, 4 / 5
 *             anti-diagonally shifted signed / unsigned, 6 column-shifted unsigned Smith-Waterman)
 *   counts[2] groups given to the main lane-per-target kernel
 *   counts[3] targets recomputed because a 16-bit lane left its exact range
 */
void miopalLastRouting(int64_t counts[4]);

/*
 * How the passes BEHIND the score / end pass of the calling thread's most recent OPAL_SEARCH_ALIGNMENT
 * search ran (diagnostics for tests; no reference counterpart). Bits:
 *   1  start cells: one lane per pair          2  ... in the query-profile form (perpair_profile_kernel)
 *   4  directions: one lane per pair           8  ... in the query-profile form
 *  16  operations copied to the host batch by batch beside the next batch
 *  32  start cells by persistent wavefronts whose lanes take the next pair when they are done
 *      (perpair_scan_refill_kernel: queries of one 64-row strip)
 *  64  directions: two pairs per lane on 16-bit halves (perpair_packed.hip)
 * 128  start cells: two pairs per lane on 16-bit halves
 * 0: no such search yet, or one whose traceback batches were built on the host.
 */
int miopalLastFullRouting(void);

/* Result-struct form, identical in shape to opalSearchDatabase but against
 * the resident mirror (what the platform plugin calls). */
int miopalSearchResults(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                        int gapExt, const int* scoreMatrix, int alphabetLength,
                        OpalSearchResult* results[], int searchType, int mode,
                        int overflowMethod, int64_t start, int64_t end);

/*
 * Host-side self tests of the scheduler that need no device (test hook; no reference counterpart).
 *   which = 1  the packed-view cache survives a builder that throws: the placeholder is dropped, the
 *              call returns MIOPAL_ERR_INTERNAL, waiters are woken and the same slice can be built
 *              afterwards (a leaked placeholder would block every later search of the slice).
 * Returns 0 when the property holds, a positive step number otherwise.
 */
int miopalSelfTest(int which);

/*
 * Fault injection for the time-out escapes of the strip hand-over (test hook; no reference
 * counterpart; an argument of the NEXT search of the calling thread, not an environment switch):
 *   kind 1  unit `unit` of the pair-table strips kernels (Smith-Waterman or NW / HW / OV of several
 *           strips) publishes nothing: the units below it give up after `spinCap` polls, flag their
 *           lanes, and the int32 kernel returns the right scores;
 *   kind 2  unit `unit` of the wavefront-per-pair strips kernel publishes nothing: the search returns
 *           MIOPAL_ERR_INTERNAL and the handle stays usable.
 * spinCap = 0 keeps the kernels' own patience (about a second).
 */
void miopalTestInjectFault(int kind, int unit, int spinCap);

/*
 * Tuning switches (diagnostics and A/B levers; pyopal_amd/csrc/tuning.h lists them). Each is MIOPAL_<NAME>
 * in the environment, which the library reads ONCE, at its first use of any switch; afterwards a switch
 * only changes through miopalSetTuning. Nothing on the search path calls getenv: the searches run without
 * the GIL on many threads (src/pyopal/lib.pyx:1364) and getenv racing another thread's putenv is a data
 * race in the C library. `name` with or without the MIOPAL_ prefix; value NULL = unset. The change is seen
 * by searches that start after the call. No reference counterpart.
 */
int miopalSetTuning(const char* name, const char* value);
const char* miopalGetTuning(const char* name);   /* NULL: unset or unknown */

/*
 * Per-handle options: the two product knobs among the switches, as arguments instead of process-wide
 * environment (value -1 = back to the process default):
 *   "reserve_cus"   compute units kept out of the persistent search launches, so that the kernels of a
 *                   collective that gathers the previous search's scores (one process per GPU, RCCL) find
 *                   room beside the search (bench.py, multi-GPU ranks: 8); default MIOPAL_RESERVE_CUS / none
 *   "small_search"  1: searches of few targets may take the wavefront-per-pair kernel (shorter start-up),
 *                   0: they stay on the lane-per-target kernels; default 1 unless MIOPAL_NO_SMALL_SEARCH
 */
int miopalDbSetOption(MiopalDb* db, const char* name, int64_t value);

/*
 * Test hook: `count` > 0 makes miopalDeviceCount() report that many devices, ordinals 0 .. count - 1 mapped
 * round-robin onto the physical gfx950 devices, so that the multi-device path of a one-process caller
 * (pyopal_amd.align: a handle, mirror and stream set per ordinal, hipSetDevice per call,
 * src/pyopal/_align.py:150-170) runs on a box with one GPU. 0 restores the physical devices.
 */
int miopalTestSetLogicalDevices(int count);

#ifdef __cplusplus
}
#endif
#endif /* MIOPAL_H */
