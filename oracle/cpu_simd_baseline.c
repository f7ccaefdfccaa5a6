/*
 * cpu_simd_baseline.c -- TEST / BENCH INFRASTRUCTURE. This repository's own
 * AVX2 inter-sequence Smith-Waterman, and (second half of the file) the NW / HW / OV modes
 * on signed 16-bit lanes with a 64-bit scalar last rung; score only. The Smith-Waterman part is the CPU baseline
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

/* 16-bit signed lanes for 16 targets (half of an 8-bit batch); column profile from the same byte
 * tables as the 8-bit pass, widened per query row. */
static void batch16(const CpuDb* db, int64_t b, int half, const unsigned char* q, int Q, int open,
                    int ext, const __m256i* tabLo, const __m256i* tabHi, int bias, __m256i* H, __m256i* E,
                    __m256i* P, short* out) {
    const __m256i vopen = _mm256_set1_epi16((short)(open > 32767 ? 32767 : open));
    const __m256i vext = _mm256_set1_epi16((short)(ext > 32767 ? 32767 : ext));
    const __m256i vbias = _mm256_set1_epi16((short)bias);
    const __m256i zero = _mm256_setzero_si256();
    const __m256i v15 = _mm256_set1_epi8(15);
    for (int i = 0; i < Q; i++) { H[i] = zero; E[i] = zero; }
    __m256i best = zero;
    const unsigned char* col = db->cols + db->batchOff[b] + half * LANES16;
    const int len = db->batchLen[b];
    const int A = db->alphabet;
    for (int j = 0; j < len; j++, col += LANES8) {
        const __m256i t = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)col));
        const __m256i isHi = _mm256_cmpgt_epi8(t, v15);
        for (int a = 0; a < A; a++) {
            const __m256i lo = _mm256_shuffle_epi8(tabLo[a], t);
            const __m256i hi = _mm256_shuffle_epi8(tabHi[a], _mm256_and_si256(t, v15));
            P[a] = _mm256_blendv_epi8(lo, hi, isHi);
        }
        __m256i diag = zero, f = zero;
        for (int i = 0; i < Q; i++) {
            /* (the padding symbol scores -bias <= 0: it cannot raise a Smith-Waterman maximum) */
            const __m256i s = _mm256_sub_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(P[q[i]])), vbias);
            __m256i h = _mm256_adds_epi16(diag, s);
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
                batch16(db, b, half, q, Q, open, ext, tabLo, tabHi, bias, H, E, P, out16);
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

/* ---- NW / HW / OV: signed 16-bit lanes, then the 64-bit scalar recurrence ------------------ */

static inline long long border_value(int gap, long long k, int open, int ext) {
    /* same definition as opal_oracle.c: one gap of k + 1 residues or k + 1 one-residue gaps */
    if (!gap) return 0;
    const long long one = open + k * ext, many = (k + 1) * (long long)open;
    return -(one < many ? one : many);
}

/* scalar 64-bit pass (last rung): the model of opal_oracle.c's dp_pass, score only */
static long long global_scalar(const unsigned char* q, int Q, const unsigned char* t, int64_t L, int open,
                               int ext, const int* S, int A, int topGap, int leftGap, int region) {
    if (Q <= 0 || L <= 0) {
        if (Q > 0 && L <= 0) return border_value(leftGap, Q - 1, open, ext);
        if (L > 0 && Q <= 0) return border_value(topGap, L - 1, open, ext);
        return 0;
    }
    long long* H = (long long*)malloc(sizeof(long long) * (size_t)Q);
    long long* E = (long long*)malloc(sizeof(long long) * (size_t)Q);
    for (int i = 0; i < Q; i++) { H[i] = border_value(leftGap, i, open, ext); E[i] = LLONG_MIN / 4; }
    long long best = LLONG_MIN;
    for (int64_t j = 0; j < L; j++) {
        long long diag = j == 0 ? 0 : border_value(topGap, j - 1, open, ext);
        long long hup = border_value(topGap, j, open, ext), f = LLONG_MIN / 4;
        for (int i = 0; i < Q; i++) {
            long long e = E[i] - ext;
            if (H[i] - open > e) e = H[i] - open;
            f -= ext;
            if (hup - open > f) f = hup - open;
            long long h = diag + S[q[i] * A + t[j]];
            if (e > h) h = e;
            if (f > h) h = f;
            diag = H[i]; H[i] = h; E[i] = e; hup = h;
            const int cand = region == 0 ? (i == Q - 1 && j == L - 1)
                           : region == 1 ? (i == Q - 1) : (i == Q - 1 || j == L - 1);
            if (cand && h > best) best = h;
        }
    }
    free(H); free(E);
    return best;
}

/* 16 targets of batch b (half = 0 / 1) in signed saturating 16-bit lanes. lens[l] = columns of
 * lane l (0 = absent). out[l] = answer of lane l under `region`. */
static void global16(const CpuDb* db, int64_t b, int half, const unsigned char* q, int Q, int open, int ext,
                     const __m256i* tabLo, const __m256i* tabHi, int bias, int topGap, int leftGap,
                     int region, const int* lens, __m256i* H, __m256i* E, __m256i* P, short* out) {
    const __m256i vopen = _mm256_set1_epi16((short)open), vext = _mm256_set1_epi16((short)ext);
    const __m256i vbias = _mm256_set1_epi16((short)bias);
    const __m256i neg = _mm256_set1_epi16(-32768);
    const __m256i v15 = _mm256_set1_epi8(15);
    short lenS[LANES16] __attribute__((aligned(32)));
    int longest = 0;
    for (int l = 0; l < LANES16; l++) { lenS[l] = (short)lens[l]; if (lens[l] > longest) longest = lens[l]; }
    const __m256i vlen = _mm256_load_si256((const __m256i*)lenS);
    for (int i = 0; i < Q; i++) {
        H[i] = _mm256_set1_epi16((short)border_value(leftGap, i, open, ext));
        E[i] = neg;
    }
    __m256i ans = neg;
    const unsigned char* col = db->cols + db->batchOff[b] + half * LANES16;
    const int A = db->alphabet;
    for (int j = 0; j < longest; j++, col += LANES8) {
        /* 16 residues -> column profile: P[a] = S[a][t_lane] + bias as bytes (padding symbol -> 0) */
        const __m128i t16 = _mm_loadu_si128((const __m128i*)col);
        const __m256i t = _mm256_broadcastsi128_si256(t16);
        const __m256i isHi = _mm256_cmpgt_epi8(t, v15);
        for (int a = 0; a < A; a++) {
            const __m256i lo = _mm256_shuffle_epi8(tabLo[a], t);
            const __m256i hi = _mm256_shuffle_epi8(tabHi[a], _mm256_and_si256(t, v15));
            P[a] = _mm256_blendv_epi8(lo, hi, isHi);
        }
        __m256i diag = _mm256_set1_epi16((short)(j == 0 ? 0 : border_value(topGap, j - 1, open, ext)));
        __m256i hup = _mm256_set1_epi16((short)border_value(topGap, j, open, ext));
        __m256i f = neg;
        for (int i = 0; i < Q; i++) {
            const __m256i s = _mm256_sub_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(P[q[i]])), vbias);
            __m256i e = _mm256_max_epi16(_mm256_subs_epi16(E[i], vext), _mm256_subs_epi16(H[i], vopen));
            f = _mm256_max_epi16(_mm256_subs_epi16(f, vext), _mm256_subs_epi16(hup, vopen));
            __m256i h = _mm256_adds_epi16(diag, s);
            h = _mm256_max_epi16(h, e);
            h = _mm256_max_epi16(h, f);
            diag = H[i];
            H[i] = h;
            E[i] = e;
            hup = h;
        }
        const __m256i vj = _mm256_set1_epi16((short)j);
        const __m256i last = _mm256_cmpeq_epi16(_mm256_add_epi16(vj, _mm256_set1_epi16(1)), vlen);  /* j == len - 1 */
        if (region == 0) {
            ans = _mm256_blendv_epi8(ans, H[Q - 1], last);
        } else {
            const __m256i inside = _mm256_cmpgt_epi16(vlen, vj);                                      /* j < len */
            ans = _mm256_blendv_epi8(ans, _mm256_max_epi16(ans, H[Q - 1]), inside);
            if (region == 2 && !_mm256_testz_si256(last, last)) {
                __m256i cm = H[0];
                for (int i = 1; i < Q; i++) cm = _mm256_max_epi16(cm, H[i]);
                ans = _mm256_blendv_epi8(ans, _mm256_max_epi16(ans, cm), last);
            }
        }
    }
    _mm256_storeu_si256((__m256i*)out, ans);
}

/*
 * NW (mode 0) / HW (1) / OV (2) score of one query against every target; scores[] in database
 * order. Whether a target fits 16-bit lanes is known from its length (every H, E, F lies above
 * -(3 open + (Q + L) ext) and H below min(Q, L) max(S)); the others take the 64-bit scalar pass.
 * Returns 0, or -1 when the matrix does not fit the byte tables.
 */
int cpuSimdSearchGlobal(const CpuDb* db, const unsigned char* q, int Q, int open, int ext, const int* S,
                        int A, int mode, int* scores, int threads) {
    if (A != db->alphabet || A >= MAXA || Q <= 0 || mode < 0 || mode > 2) return -1;
    int minS = 0, maxS = 0;
    for (int k = 0; k < A * A; k++) { if (S[k] < minS) minS = S[k]; if (S[k] > maxS) maxS = S[k]; }
    const int bias = -minS;
    if (bias + maxS > 255 || open < 0 || ext < 0 || open > 30000 || ext > 30000) return -1;
    const int topGap = mode == 0, leftGap = mode != 2, region = mode;
    __m256i tabLo[MAXA], tabHi[MAXA];
    for (int a = 0; a < A; a++) {
        unsigned char lo[16], hi[16];
        for (int t = 0; t < 16; t++) {
            lo[t] = (unsigned char)(t < A ? S[a * A + t] + bias : 0);       /* padding symbol A -> -bias */
            hi[t] = (unsigned char)(t + 16 < A ? S[a * A + t + 16] + bias : 0);
        }
        tabLo[a] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)lo));
        tabHi[a] = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i*)hi));
    }
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
        short out16[LANES16];
#pragma omp for schedule(dynamic, 1)
        for (int64_t hb = 0; hb < 2 * db->nBatches; hb++) {
            const int64_t b = hb >> 1;
            const int half = (int)(hb & 1);
            int lens[LANES16], fits[LANES16], any = 0;
            for (int l = 0; l < LANES16; l++) {
                const int64_t k = b * LANES8 + half * LANES16 + l;
                lens[l] = 0; fits[l] = 0;
                if (k >= db->n) continue;
                const int32_t id = db->order[k];
                const int64_t L = db->offsets[id + 1] - db->offsets[id];
                const int64_t low = 3 * (int64_t)open + (Q + L) * (int64_t)ext - minS;
                const int64_t high = (Q < L ? Q : L) * (int64_t)(maxS > 0 ? maxS : 0);
                if (L > 0 && L < 32000 && low < 32000 && high < 32000) { fits[l] = 1; lens[l] = (int)L; any = 1; }
            }
            if (any) global16(db, b, half, q, Q, open, ext, tabLo, tabHi, bias, topGap, leftGap, region, lens, H, E, P, out16);
            for (int l = 0; l < LANES16; l++) {
                const int64_t k = b * LANES8 + half * LANES16 + l;
                if (k >= db->n) break;
                const int32_t id = db->order[k];
                if (fits[l]) scores[id] = out16[l];
                else scores[id] = (int)global_scalar(q, Q, db->residues + db->offsets[id],
                                                     db->offsets[id + 1] - db->offsets[id], open, ext, S, A,
                                                     topGap, leftGap, region);
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
