/*
 * opal_oracle.c -- TEST INFRASTRUCTURE. Scalar CPU restatement of the
 * behaviour of `opalSearchDatabase` (declared `src/pyopal/opal.pxd:38-52`,
 * called at `src/pyopal/platform/pyx.in:76-91`).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's library. The product path (libmiopal.so) never does.
 *
 * Parity status. The arithmetic of the reference lives in `vendor/opal`
 * (git submodule `https://github.com/althonos/opal`, .gitmodules:1-3; pinned
 * commit not recorded in the tree, release context pyopal 0.7.3,
 * pyproject.toml:7) which is ABSENT from /root/reference, so the reference
 * cannot be built or run here. This restatement follows the documented
 * semantics (src/pyopal/lib.pyx:1269-1318, src/pyopal/_align.py:42-98) and is
 * pinned by every known-answer vector the reference's tests hold for this
 * path (tests/golden/reference_vectors.json, checked by
 * tests/test_oracle_golden.py):
 *   - src/pyopal/tests/test_aligner.py:42-79   NW  44, end (5,7), start (0,0)
 *   - src/pyopal/tests/test_aligner.py:93-131  SW  47, end (5,7), start (0,1)
 *   - src/pyopal/tests/test_align.py:9-37      NW full on 4 targets
 *   - src/pyopal/lib.pyx:1006-1010             CIGAR 1D5M1D1M
 *   - src/pyopal/lib.pyx:1076-1082             coverage 1.0 / 0.875
 *   - src/pyopal/_align.py:106-111             SW gap_open=2: 41/31/23
 * Everything those vectors do not reach (HW/OV values, tie-breaks between
 * equal-scoring end cells / start cells / traceback moves other than
 * "diagonal first") is a documented choice of this file: PARITY UNPINNED
 * beyond the vectors above.
 *
 * Model:
 *   E[i][j] = max(E[i][j-1] - ext, H[i][j-1] - open)   gap consuming target
 *   F[i][j] = max(F[i-1][j] - ext, H[i-1][j] - open)   gap consuming query
 *   H[i][j] = max(H[i-1][j-1] + S[q_i][t_j], E[i][j], F[i][j])   (SW: and 0)
 * a gap of length n costs open + (n-1)*ext (pinned by _align.py:106-111).
 * Borders:  NW  H[-1][j] = -(open + j*ext), H[i][-1] = -(open + i*ext)
 *               (-(k+1)*open when open < ext: k+1 one-residue gaps are cheaper)
 *           HW  H[-1][j] = 0,               H[i][-1] = -(open + i*ext)
 *           OV  both 0;  SW both 0 and H floored at 0.
 * Answer:   NW last cell; HW max of last query row; OV max of last row and
 *           last column; SW max of all cells.
 * End location: candidates are scanned target column by target column, query
 *           row by query row inside a column; a candidate replaces the best
 *           only when strictly greater. SW starts from best = 0 with no
 *           location (an all-non-positive matrix gives score 0, locations -1
 *           and an empty alignment).
 * Start location (ALIGNMENT, not NW): the same scan applied to the reversed
 *           prefixes q[0..qe], t[0..te] under NW borders (the alignment is
 *           anchored on the end cell); region all cells for SW, last row for
 *           HW, last row and last column for OV.
 *           When every substitution is worse than gapping, the HW / OV optimum can
 *           be a single gap over one sequence only; the other sequence then has an
 *           empty span (start = end + 1) and the alignment is all DEL or all INS.
 * Traceback: global alignment of q[qs..qe] with t[ts..te], walked back from
 *           the end cell; on ties diagonal first (pinned, lib.pyx:1006-1010),
 *           then the target-consuming gap (INS), then the query-consuming gap
 *           (DEL); inside a gap "close the gap" is preferred to "extend".
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/opal.h"

#define NEG_INF (INT_MIN / 4)

static inline int imax(int a, int b) { return a > b ? a : b; }

typedef struct {
    int top_gap;  /* H[-1][j] penalised */
    int left_gap; /* H[i][-1] penalised */
    int floor0;   /* SW */
    int region;   /* 0 last cell, 1 last row, 2 last row + last column, 3 all */
} dp_rules;

static int rules_for_mode(int mode, dp_rules* r) {
    switch (mode) {
        case OPAL_MODE_NW: *r = (dp_rules){1, 1, 0, 0}; return 0;
        case OPAL_MODE_HW: *r = (dp_rules){0, 1, 0, 1}; return 0;
        case OPAL_MODE_OV: *r = (dp_rules){0, 0, 0, 2}; return 0;
        case OPAL_MODE_SW: *r = (dp_rules){0, 0, 1, 3}; return 0;
    }
    return OPAL_ERR_INVALID_MODE;
}

/* Sequence accessors that can walk a buffer backwards (reverse pass). */
typedef struct {
    const unsigned char* p;
    int step; /* +1 or -1 */
} seqview;
static inline int sv(const seqview* s, int k) { return s->p[(long)k * s->step]; }

static inline int border(int gap, int k, int open, int ext) {
    /* value of a border cell that is k (0-based) residues into a border gap: one gap of
     * k + 1 residues, or, when opening is cheaper than extending (open < ext), k + 1
     * gaps of one residue, which is what the recurrence itself does inside the matrix */
    if (!gap) return 0;
    long one = open + (long)k * ext, many = (long)(k + 1) * open;
    return -(int)(one < many ? one : many);
}

/*
 * One DP pass over Q x L cells with rolling columns. Returns the best score
 * and its location under `r`. Overflow of the int range is tracked in 64 bit
 * and reported through *ovf.
 */
static void dp_pass(const seqview* q, int Q, const seqview* t, int L, int open, int ext,
                    const int* S, int A, const dp_rules* r, int* best_out, int* bi_out,
                    int* bj_out, int* ovf) {
    int64_t best = r->floor0 ? 0 : INT64_MIN;
    int bi = -1, bj = -1;
    if (Q <= 0 || L <= 0) {
        /* degenerate: closed forms */
        int score = 0;
        if (!r->floor0) {
            if (Q > 0 && L <= 0) score = border(r->left_gap, Q - 1, open, ext);
            if (L > 0 && Q <= 0) score = border(r->top_gap, L - 1, open, ext);
        }
        *best_out = score;
        *bi_out = -1;
        *bj_out = -1;
        return;
    }
    int64_t* Hc = (int64_t*)malloc(sizeof(int64_t) * (size_t)Q);
    int64_t* Ec = (int64_t*)malloc(sizeof(int64_t) * (size_t)Q);
    for (int i = 0; i < Q; i++) {
        Hc[i] = border(r->left_gap, i, open, ext); /* column -1 */
        Ec[i] = INT64_MIN / 4;
    }
    for (int j = 0; j < L; j++) {
        int tj = sv(t, j);
        int64_t hdiag = (j == 0) ? 0 : border(r->top_gap, j - 1, open, ext); /* H[-1][j-1] */
        int64_t hup = border(r->top_gap, j, open, ext);                      /* H[-1][j]   */
        int64_t f = INT64_MIN / 4;
        for (int i = 0; i < Q; i++) {
            int64_t e = Ec[i] - ext;
            if (Hc[i] - open > e) e = Hc[i] - open;
            f = f - ext;
            if (hup - open > f) f = hup - open;
            int64_t h = hdiag + S[sv(q, i) * A + tj];
            if (e > h) h = e;
            if (f > h) h = f;
            if (r->floor0 && h < 0) h = 0;
            hdiag = Hc[i];
            Hc[i] = h;
            Ec[i] = e;
            hup = h;
            int cand;
            switch (r->region) {
                case 0: cand = (i == Q - 1 && j == L - 1); break;
                case 1: cand = (i == Q - 1); break;
                case 2: cand = (i == Q - 1 || j == L - 1); break;
                default: cand = 1;
            }
            if (cand && h > best) {
                best = h;
                bi = i;
                bj = j;
            }
            if (h > INT_MAX || h < NEG_INF) *ovf = 1;
        }
    }
    free(Hc);
    free(Ec);
    if (best > INT_MAX || best < INT_MIN) *ovf = 1;
    *best_out = (int)best;
    *bi_out = bi;
    *bj_out = bj;
}

/* Global alignment of q[0..n-1] with t[0..m-1] and traceback. */
static int traceback(const unsigned char* q, int n, const unsigned char* t, int m, int open,
                     int ext, const int* S, int A, int expect, unsigned char** aln_out,
                     int* len_out) {
    size_t W = (size_t)m + 1;
    size_t cells = ((size_t)n + 1) * W;
    int* H = (int*)malloc(sizeof(int) * cells);
    int* E = (int*)malloc(sizeof(int) * cells);
    int* F = (int*)malloc(sizeof(int) * cells);
    unsigned char* ops = (unsigned char*)malloc((size_t)n + (size_t)m + 1);
    if (!H || !E || !F || !ops) {
        free(H); free(E); free(F); free(ops);
        return MIOPAL_ERR_INTERNAL;
    }
#define AT(M, i, j) M[(size_t)(i) * W + (size_t)(j)]
    AT(H, 0, 0) = 0;
    AT(E, 0, 0) = AT(F, 0, 0) = NEG_INF;
    for (int j = 1; j <= m; j++) {
        AT(H, 0, j) = border(1, j - 1, open, ext);
        AT(E, 0, j) = AT(H, 0, j);
        AT(F, 0, j) = NEG_INF;
    }
    for (int i = 1; i <= n; i++) {
        AT(H, i, 0) = border(1, i - 1, open, ext);
        AT(F, i, 0) = AT(H, i, 0);
        AT(E, i, 0) = NEG_INF;
        for (int j = 1; j <= m; j++) {
            int e = imax(AT(E, i, j - 1) - ext, AT(H, i, j - 1) - open);
            int f = imax(AT(F, i - 1, j) - ext, AT(H, i - 1, j) - open);
            int h = AT(H, i - 1, j - 1) + S[q[i - 1] * A + t[j - 1]];
            AT(E, i, j) = e;
            AT(F, i, j) = f;
            AT(H, i, j) = imax(h, imax(e, f));
        }
    }
    int rc = 0;
    if (AT(H, n, m) != expect) rc = MIOPAL_ERR_INTERNAL;
    int len = 0, i = n, j = m, state = 0; /* 0 = H, 1 = E, 2 = F */
    while (i > 0 || j > 0) {
        if (i == 0) { ops[len++] = OPAL_ALIGN_INS; j--; continue; }
        if (j == 0) { ops[len++] = OPAL_ALIGN_DEL; i--; continue; }
        if (state == 0) {
            int h = AT(H, i, j);
            if (h == AT(H, i - 1, j - 1) + S[q[i - 1] * A + t[j - 1]]) {
                ops[len++] = (q[i - 1] == t[j - 1]) ? OPAL_ALIGN_MATCH : OPAL_ALIGN_MISMATCH;
                i--; j--;
            } else if (h == AT(E, i, j)) {
                state = 1;
            } else {
                state = 2;
            }
        } else if (state == 1) {
            ops[len++] = OPAL_ALIGN_INS;
            if (AT(E, i, j) == AT(H, i, j - 1) - open) state = 0;
            j--;
        } else {
            ops[len++] = OPAL_ALIGN_DEL;
            if (AT(F, i, j) == AT(H, i - 1, j) - open) state = 0;
            i--;
        }
    }
#undef AT
    for (int a = 0, b = len - 1; a < b; a++, b--) {
        unsigned char tmp = ops[a]; ops[a] = ops[b]; ops[b] = tmp;
    }
    free(H); free(E); free(F);
    *aln_out = ops;
    *len_out = len;
    return rc;
}

void oracleInitSearchResult(OpalSearchResult* r) {
    r->scoreSet = 0;
    r->score = 0;
    r->endLocationTarget = r->endLocationQuery = -1;
    r->startLocationTarget = r->startLocationQuery = -1;
    r->alignment = NULL;
    r->alignmentLength = 0;
}

/* One query against one target. */
int oracleAlignPair(const unsigned char* query, int Q, const unsigned char* target, int L,
                    int open, int ext, const int* S, int A, int searchType, int mode,
                    OpalSearchResult* res) {
    dp_rules r;
    int rc = rules_for_mode(mode, &r);
    if (rc) return rc;
    if (searchType < OPAL_SEARCH_SCORE || searchType > OPAL_SEARCH_ALIGNMENT)
        return OPAL_ERR_INVALID_MODE;
    seqview qv = {query, 1}, tv = {target, 1};
    int score, qe, te, ovf = 0;
    dp_pass(&qv, Q, &tv, L, open, ext, S, A, &r, &score, &qe, &te, &ovf);
    if (ovf) return OPAL_ERR_OVERFLOW;
    res->scoreSet = 1;
    res->score = score;
    if (searchType == OPAL_SEARCH_SCORE) return 0;
    res->endLocationQuery = qe;
    res->endLocationTarget = te;
    if (searchType == OPAL_SEARCH_SCORE_END) return 0;
    if (qe < 0 || te < 0) return 0; /* empty alignment */
    int qs = 0, ts = 0;
    if (mode != OPAL_MODE_NW) {
        dp_rules rr = {1, 1, 0, r.region};
        seqview rq = {query + qe, -1}, rt = {target + te, -1};
        int rs, ri, rj;
        dp_pass(&rq, qe + 1, &rt, te + 1, open, ext, S, A, &rr, &rs, &ri, &rj, &ovf);
        if (ovf) return OPAL_ERR_OVERFLOW;
        if (rs != score) {
            /* Degenerate optimum: the best HW / OV "alignment" is one gap that consumes
             * residues of one sequence only (every substitution is worse than gapping).
             * Anchored on the end cell that is a border cell of the reversed problem:
             * the whole query piece gapped with no target residue (border of the last
             * row), or, for OV, the whole target piece gapped with no query residue
             * (border of the last column). The empty side gets start = end + 1. */
            if (mode != OPAL_MODE_SW && score == border(1, qe, open, ext)) {
                ri = qe;
                rj = -1;
            } else if (mode == OPAL_MODE_OV && score == border(1, te, open, ext)) {
                ri = -1;
                rj = te;
            } else {
                return MIOPAL_ERR_INTERNAL;
            }
        }
        qs = qe - ri;
        ts = te - rj;
    }
    res->startLocationQuery = qs;
    res->startLocationTarget = ts;
    return traceback(query + qs, qe - qs + 1, target + ts, te - ts + 1, open, ext, S, A, score,
                     &res->alignment, &res->alignmentLength);
}

/* Same shape as opalSearchDatabase (src/pyopal/opal.pxd:38-52). */
int oracleSearchDatabase(unsigned char query[], int queryLength, unsigned char* db[],
                         int dbLength, int dbSeqLengths[], int gapOpen, int gapExt,
                         int* scoreMatrix, int alphabetLength, OpalSearchResult* results[],
                         const int searchType, int mode, int overflowMethod) {
    (void)overflowMethod;
    for (int k = 0; k < dbLength; k++) {
        int rc = oracleAlignPair(query, queryLength, db[k], dbSeqLengths[k], gapOpen, gapExt,
                                 scoreMatrix, alphabetLength, searchType, mode, results[k]);
        if (rc) return rc;
    }
    return 0;
}

/*
 * Flat-batch form used by the tests and by bench.py's cpu_baseline leg:
 * residues concatenated, offsets[n+1]; outputs are plain int arrays (any of
 * them may be NULL). Alignments are returned concatenated in aln (capacity
 * aln_cap bytes) with aln_off[n+1]. Returns 0 or an error code.
 */
int oracleSearchFlat(const unsigned char* query, int Q, const unsigned char* residues,
                     const int64_t* offsets, int n, int open, int ext, const int* S, int A,
                     int searchType, int mode, int* score, int* end_t, int* end_q,
                     int* start_t, int* start_q, unsigned char* aln, int64_t aln_cap,
                     int64_t* aln_off) {
    int64_t used = 0;
    if (aln_off) aln_off[0] = 0;
    for (int k = 0; k < n; k++) {
        OpalSearchResult r;
        oracleInitSearchResult(&r);
        int L = (int)(offsets[k + 1] - offsets[k]);
        int rc = oracleAlignPair(query, Q, residues + offsets[k], L, open, ext, S, A, searchType,
                                 mode, &r);
        if (rc) { free(r.alignment); return rc; }
        if (score) score[k] = r.score;
        if (end_t) end_t[k] = r.endLocationTarget;
        if (end_q) end_q[k] = r.endLocationQuery;
        if (start_t) start_t[k] = r.startLocationTarget;
        if (start_q) start_q[k] = r.startLocationQuery;
        if (aln_off) {
            if (aln && used + r.alignmentLength <= aln_cap && r.alignmentLength > 0)
                memcpy(aln + used, r.alignment, (size_t)r.alignmentLength);
            else if (r.alignmentLength > 0 && aln) { free(r.alignment); return MIOPAL_ERR_BAD_ARGUMENT; }
            used += r.alignmentLength;
            aln_off[k + 1] = used;
        }
        free(r.alignment);
    }
    return 0;
}
