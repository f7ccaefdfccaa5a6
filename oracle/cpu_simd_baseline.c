/*
 * cpu_simd_baseline.c -- TEST / BENCH INFRASTRUCTURE. This repository's own
 * AVX2 inter-sequence Smith-Waterman (score only), used as the CPU baseline
 * leg of bench.py ("cpu_baseline.kind" = "port") and checked against
 * opal_oracle.c by tests/test_cpu_baseline.py. NOT Opal: the reference's AVX2
 * code (vendor/opal, absent from /root/reference) cannot be built here; this
 * file implements the scheme the reference documents for it
 * (README.md:26-28, src/pyopal/lib.pyx:1283-1289): one SIMD lane per database
 * sequence (SWIPE), 8-bit lanes first (32 per AVX2 register), targets whose
 * lanes saturate recomputed with 16-bit lanes, then with the 32-bit scalar
 * recurrence ("simple" overflow strategy). Threads take batches of 32 targets
 * like the chunked ThreadPool of src/pyopal/_align.py:150-170.
 *
 * The product path never loads this file.
 */
#include <immintrin.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES8 32
#define LANES16 16
#define MAXA 32

typedef struct {
    int64_t n;
    int alphabet;
    const unsigned char* residues; /* borrowed */
    const int64_t* offsets;        /* borrowed */
    int32_t* order;                /* targets sorted by length, longest first */
    int64_t nBatches;              /* batches of 32 consecutive sorted targets */
    int64_t* batchOff;             /* byte offset of each batch in cols */
    int32_t* batchLen;             /* columns per batch (longest member) */
    unsigned char* cols;           /* [batch][column][32 lanes], pad = alphabet */
} CpuDb;

static const CpuDb* g_sort_db;
static int cmp_len_desc(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    const int64_t lx = g_sort_db->offsets[x + 1] - g_sort_db->offsets[x];
    const int64_t ly = g_sort_db->offsets[y + 1] - g_sort_db->offsets[y];
    if (lx != ly) return lx > ly ? -1 : 1;
    return x < y ? -1 : (x > y);
}

void cpuSimdFree(CpuDb* db) {
    if (!db) return;
    free(db->order); free(db->batchOff); free(db->batchLen); free(db->cols); free(db);
}

/* Transposed copy of the database (the CPU analogue of the GPU pack; untimed). */
CpuDb* cpuSimdPrepare(const unsigned char* residues, const int64_t* offsets, int64_t n, int alphabet) {
    CpuDb* db = (CpuDb*)calloc(1, sizeof(CpuDb));
    db->n = n; db->alphabet = alphabet; db->residues = residues; db->offsets = offsets;
    db->order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t k = 0; k < n; k++) db->order[k] = (int32_t)k;
    g_sort_db = db;
    qsort(db->order, (size_t)n, sizeof(int32_t), cmp_len_desc);
    db->nBatches = (n + LANES8 - 1) / LANES8;
    db->batchOff = (int64_t*)malloc(sizeof(int64_t) * (size_t)(db->nBatches + 1));
    db->batchLen = (int32_t*)malloc(sizeof(int32_t) * (size_t)(db->nBatches + 1));
    int64_t total = 0;
    for (int64_t b = 0; b < db->nBatches; b++) {
        const int32_t id = db->order[b * LANES8];
        const int32_t len = (int32_t)(offsets[id + 1] - offsets[id]);
        db->batchOff[b] = total; db->batchLen[b] = len;
        total += (int64_t)len * LANES8;
    }
    db->batchOff[db->nBatches] = total;
    db->cols = (unsigned char*)aligned_alloc(64, (size_t)((total + 63) / 64 * 64 + 64));
    memset(db->cols, alphabet, (size_t)total);
    for (int64_t b = 0; b < db->nBatches; b++) {
        unsigned char* dst = db->cols + db->batchOff[b];
        for (int l = 0; l < LANES8; l++) {
            const int64_t k = b * LANES8 + l;
            if (k >= n) break;
            const int32_t id = db->order[k];
            const unsigned char* src = residues + offsets[id];
            const int64_t len = offsets[id + 1] - offsets[id];
            for (int64_t j = 0; j < len; j++) dst[j * LANES8 + l] = src[j];
        }
    }
    return db;
}

/* scalar 32-bit SW score (last rung of the ladder) */
static int sw_scalar(const unsigned char* q, int Q, const unsigned char* t, int64_t L, int open,
                     int ext, const int* S, int A) {
    int* H = (int*)calloc((size_t)Q, sizeof(int));
    int* E = (int*)calloc((size_t)Q, sizeof(int));
    int best = 0;
    for (int64_t j = 0; j < L; j++) {
        int diag = 0, f = 0, hup = 0;
        for (int i = 0; i < Q; i++) {
            int e = E[i] - ext;
            if (H[i] - open > e) e = H[i] - open;
            if (e < 0) e = 0;
            f = f - ext;
            if (hup - open > f) f = hup - open;
            if (f < 0) f = 0;
            int h = diag + S[q[i] * A + t[j]];
            if (e > h) h = e;
            if (f > h) h = f;
            if (h < 0) h = 0;
            diag = H[i];
            H[i] = h;
            E[i] = e;
            hup = h;
            if (h > best) best = h;
        }
    }
    free(H); free(E);
    return best;
}

/* 8-bit lanes, biased unsigned arithmetic. Returns per-lane best (255 = saturated). */
static void batch8(const CpuDb* db, int64_t b, const unsigned char* q, int Q, int open, int ext,
                   const __m256i* tabLo, const __m256i* tabHi, int bias, __m256i* H, __m256i* E,
                   __m256i* P, unsigned char* out) {
    const __m256i vbias = _mm256_set1_epi8((char)bias);
    const __m256i vopen = _mm256_set1_epi8((char)(open > 255 ? 255 : open));
    const __m256i vext = _mm256_set1_epi8((char)(ext > 255 ? 255 : ext));
    const __m256i zero = _mm256_setzero_si256();
    const __m256i v15 = _mm256_set1_epi8(15);
    for (int i = 0; i < Q; i++) { H[i] = zero; E[i] = zero; }
    __m256i best = zero;
    const unsigned char* col = db->cols + db->batchOff[b];
    const int len = db->batchLen[b];
    const int A = db->alphabet;
    for (int j = 0; j < len; j++, col += LANES8) {
        const __m256i t = _mm256_load_si256((const __m256i*)col);
        /* column profile P[a][lane] = S[a][t_lane] + bias, by two 16-entry byte shuffles */
        const __m256i isHi = _mm256_cmpgt_epi8(t, v15);
        for (int a = 0; a < A; a++) {
            const __m256i lo = _mm256_shuffle_epi8(tabLo[a], t);
            const __m256i hi = _mm256_shuffle_epi8(tabHi[a], _mm256_and_si256(t, v15));
            P[a] = _mm256_blendv_epi8(lo, hi, isHi);
        }
        __m256i diag = zero, f = zero;
        for (int i = 0; i < Q; i++) {
            __m256i h = _mm256_subs_epu8(_mm256_adds_epu8(diag, P[q[i]]), vbias);
            h = _mm256_max_epu8(h, E[i]);
            h = _mm256_max_epu8(h, f);
            best = _mm256_max_epu8(best, h);
            const __m256i hmo = _mm256_subs_epu8(h, vopen);
            E[i] = _mm256_max_epu8(_mm256_subs_epu8(E[i], vext), hmo);
            f = _mm256_max_epu8(_mm256_subs_epu8(f, vext), hmo);
            diag = H[i];
            H[i] = h;
        }
    }
    _mm256_storeu_si256((__m256i*)out, best);
}

/* 16-bit signed lanes for 16 targets (half of an 8-bit batch). */
static void batch16(const CpuDb* db, int64_t b, int half, const unsigned char* q, int Q, int open,
                    int ext, const int* S, __m256i* H, __m256i* E, short* out) {
    const __m256i vopen = _mm256_set1_epi16((short)(open > 32767 ? 32767 : open));
    const __m256i vext = _mm256_set1_epi16((short)(ext > 32767 ? 32767 : ext));
    const __m256i zero = _mm256_setzero_si256();
    for (int i = 0; i < Q; i++) { H[i] = zero; E[i] = zero; }
    __m256i best = zero;
    const unsigned char* col = db->cols + db->batchOff[b] + half * LANES16;
    const int len = db->batchLen[b];
    const int A = db->alphabet;
    short sc[LANES16] __attribute__((aligned(32)));
    for (int j = 0; j < len; j++, col += LANES8) {
        __m256i diag = zero, f = zero;
        for (int i = 0; i < Q; i++) {
            const int* row = S + q[i] * A;
            for (int l = 0; l < LANES16; l++) sc[l] = col[l] < A ? (short)row[col[l]] : (short)-32768;
            __m256i h = _mm256_adds_epi16(diag, _mm256_load_si256((const __m256i*)sc));
            h = _mm256_max_epi16(h, E[i]);
            h = _mm256_max_epi16(h, f);
            best = _mm256_max_epi16(best, h);
            const __m256i hmo = _mm256_subs_epu16(h, vopen);
            E[i] = _mm256_max_epi16(_mm256_subs_epu16(E[i], vext), hmo);
            f = _mm256_max_epi16(_mm256_subs_epu16(f, vext), hmo);
            diag = H[i];
            H[i] = h;
        }
    }
    _mm256_storeu_si256((__m256i*)out, best);
}

/*
 * SW score of one query against every target. scores[] is in database order.
 * Returns 0, or -1 when the matrix does not fit the 8-bit bias scheme.
 */
int cpuSimdSearchSW(const CpuDb* db, const unsigned char* q, int Q, int open, int ext,
                    const int* S, int A, int* scores, int threads) {
    if (A != db->alphabet || A >= MAXA || Q <= 0) return -1; /* symbol A is the padding code */
    int minS = 0, maxS = 0;
    for (int k = 0; k < A * A; k++) { if (S[k] < minS) minS = S[k]; if (S[k] > maxS) maxS = S[k]; }
    const int bias = -minS;
    if (bias + maxS > 127 || open < 0 || ext < 0) return -1;
    /* byte tables: row a, entries 0..15 and 16..31 (+ padding symbol = A -> 0 = "-bias") */
    __m256i tabLo[MAXA], tabHi[MAXA];
    for (int a = 0; a < A; a++) {
        unsigned char lo[16], hi[16];
        for (int t = 0; t < 16; t++) {
            lo[t] = (unsigned char)(t < A ? S[a * A + t] + bias : 0);
            hi[t] = (unsigned char)(t + 16 < A ? S[a * A + t + 16] + bias : 0);
        }
        tabLo[a] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)lo));
        tabHi[a] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)hi));
    }
    const int limit8 = 255 - bias - maxS; /* below this no lane ever clipped */
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel
    {
        __m256i* H = (__m256i*)aligned_alloc(32, sizeof(__m256i) * (size_t)Q);
        __m256i* E = (__m256i*)aligned_alloc(32, sizeof(__m256i) * (size_t)Q);
        __m256i P[MAXA];
        unsigned char out8[LANES8];
        short out16[LANES16];
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < db->nBatches; b++) {
            batch8(db, b, q, Q, open, ext, tabLo, tabHi, bias, H, E, P, out8);
            int redo[2] = {0, 0};
            for (int l = 0; l < LANES8; l++) {
                const int64_t k = b * LANES8 + l;
                if (k >= db->n) break;
                if (out8[l] >= limit8) redo[l / LANES16] = 1;
                else scores[db->order[k]] = out8[l];
            }
            for (int half = 0; half < 2; half++) {
                if (!redo[half]) continue;
                batch16(db, b, half, q, Q, open, ext, S, H, E, out16);
                for (int l = 0; l < LANES16; l++) {
                    const int64_t k = b * LANES8 + half * LANES16 + l;
                    if (k >= db->n) break;
                    const int32_t id = db->order[k];
                    if (out8[half * LANES16 + l] < limit8)
                        continue; /* the 8-bit result was exact */
                    if (out16[l] >= 32767)
                        scores[id] = sw_scalar(q, Q, db->residues + db->offsets[id],
                                               db->offsets[id + 1] - db->offsets[id], open, ext, S, A);
                    else if (out8[half * LANES16 + l] >= limit8)
                        scores[id] = out16[l];
                }
            }
        }
        free(H); free(E);
    }
    return 0;
}

int cpuSimdThreads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
