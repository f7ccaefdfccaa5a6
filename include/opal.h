/*
 * opal.h -- drop-in C ABI of the Opal database-search entry points, served by
 * hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * This header re-states, declaration for declaration, what the reference binds
 * through Cython in `src/pyopal/opal.pxd:1-67` (the real `opal.h` lives in the
 * un-vendored submodule `vendor/opal` and is absent from the reference tree).
 * A build of pyopal that puts this header first on its include path and links
 * `libmiopal.so` instead of compiling `vendor/opal/src/opal.cpp` gets the GPU
 * path behind the unchanged `opal.opalSearchDatabase(...)` call in
 * `src/pyopal/platform/pyx.in:76-91`.
 *
 * Numeric values of the constants: only the OPAL_ALIGN_* order is observable
 * through the reference's Python API (`src/pyopal/lib.pyx:97-102`,
 * `lib.pyx:991-996`, `lib.pyx:1017-1026`); the others follow upstream Opal's
 * published header.
 */
#ifndef MIOPAL_OPAL_H
#define MIOPAL_OPAL_H

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes (src/pyopal/opal.pxd:3-5; mapped to exceptions at
 * src/pyopal/platform/pyx.in:102-107). */
#define OPAL_ERR_OVERFLOW 1
#define OPAL_ERR_NO_SIMD_SUPPORT 2 /* here: "no usable gfx950 device" */
#define OPAL_ERR_INVALID_MODE 3
/* Extension: any HIP runtime failure. Surfaces through the generic branch
 * `RuntimeError(f"failed to align to Opal database (code={retcode})")`. */
#define MIOPAL_ERR_HIP 100
#define MIOPAL_ERR_BAD_ARGUMENT 101
#define MIOPAL_ERR_INTERNAL 102

/* Alignment modes (src/pyopal/opal.pxd:7-10). */
#define OPAL_MODE_NW 0
#define OPAL_MODE_HW 1
#define OPAL_MODE_OV 2
#define OPAL_MODE_SW 3

/* Overflow strategies (src/pyopal/opal.pxd:12-13). Accepted and ignored:
 * the GPU path picks the narrowest exact lane width per target, so results
 * are identical for both values. */
#define OPAL_OVERFLOW_SIMPLE 0
#define OPAL_OVERFLOW_BUCKETS 1

/* Search types (src/pyopal/opal.pxd:15-17). */
#define OPAL_SEARCH_SCORE 0
#define OPAL_SEARCH_SCORE_END 1
#define OPAL_SEARCH_ALIGNMENT 2

/* Alignment operations (src/pyopal/opal.pxd:19-22). */
#define OPAL_ALIGN_MATCH 0    /* 'M' */
#define OPAL_ALIGN_DEL 1      /* query residue against a gap    */
#define OPAL_ALIGN_INS 2      /* target residue against a gap   */
#define OPAL_ALIGN_MISMATCH 3 /* 'X' */

/* src/pyopal/opal.pxd:24-32. Locations are 0-based and inclusive; -1 = unset.
 * `alignment` is allocated with malloc() by the callee and owned by the caller
 * afterwards (pyopal releases it in ScoreResult.__dealloc__, lib.pyx:797-798). */
struct OpalSearchResult {
    int scoreSet;
    int score;
    int endLocationTarget;
    int endLocationQuery;
    int startLocationTarget;
    int startLocationQuery;
    unsigned char* alignment;
    int alignmentLength;
};
typedef struct OpalSearchResult OpalSearchResult;

/* src/pyopal/opal.pxd:34-36 */
void opalInitSearchResult(OpalSearchResult* result);
int opalSearchResultIsEmpty(const OpalSearchResult result);
void opalSearchResultSetScore(OpalSearchResult* result, int score);

/* src/pyopal/opal.pxd:38-52. One query against dbLength targets.
 * Packs and uploads the targets on every call; callers that reuse a database
 * should hold a MiopalDb (miopal.h) instead. Thread-safe and re-entrant. */
int opalSearchDatabase(
    unsigned char query[], int queryLength,
    unsigned char* db[], int dbLength, int dbSeqLengths[],
    int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
    OpalSearchResult* results[],
    const int searchType, int mode, int overflowMethod);

/* src/pyopal/opal.pxd:54-65. SW, score only. */
int opalSearchDatabaseCharSW(
    unsigned char query[], int queryLength,
    unsigned char** db, int dbLength, int dbSeqLengths[],
    int gapOpen, int gapExt, int* scoreMatrix, int alphabetLength,
    OpalSearchResult* results[]);

#ifdef __cplusplus
}
#endif
#endif /* MIOPAL_OPAL_H */
