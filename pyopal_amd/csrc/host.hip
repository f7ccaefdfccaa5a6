// Host-side scheduler behind the C ABI of include/opal.h and include/miopal.h.
//
// Plays the role of the scalar driver code of opalSearchDatabase (declared
// src/pyopal/opal.pxd:38-52, called src/pyopal/platform/pyx.in:76-91): argument
// checks, choice of lane width per target (the reference's 8/16/32-bit ladder,
// src/pyopal/lib.pyx:1283-1289, becomes packed-16 / 32 here), the score pass,
// the reversed-prefix pass for start locations and the traceback, all against
// a device-resident database.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <exception>
#include <ctime>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <vector>

#include "../../include/miopal.h"
#include "common.h"
#include "tuning.h"

using namespace miopal;

// ---------------------------------------------------------------------------
// tuning switches (tuning.h): the environment is read here, once, and nowhere else
// ---------------------------------------------------------------------------
namespace miopal {
namespace {
const char* const kTuningNames[] = {
#define X(name) "MIOPAL_" #name,
    MIOPAL_TUNING_SWITCHES(X)
#undef X
};
struct Tuning {
    std::atomic<const char*> value[(int)Tune::kCount];
    Tuning() { fromEnv(); }
    // (copies: a later putenv of the process may not pull a string from under a search)
    void fromEnv() {
        for (int k = 0; k < (int)Tune::kCount; ++k) {
            const char* v = getenv(kTuningNames[k]);
            value[k].store(v ? strdup(v) : nullptr, std::memory_order_release);
        }
    }
    static Tuning& table() {
        static Tuning t;   // (thread-safe initialisation)
        return t;
    }
    static int find(const char* name) {
        if (!name) return -1;
        if (strncmp(name, "MIOPAL_", 7) == 0) name += 7;
        for (int k = 0; k < (int)Tune::kCount; ++k)
            if (strcmp(kTuningNames[k] + 7, name) == 0) return k;
        return -1;
    }
};
}  // namespace
const char* tuned(Tune key) { return Tuning::table().value[(int)key].load(std::memory_order_acquire); }
}  // namespace miopal

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_lastError;
thread_local int64_t g_lastRouting[4] = {0, 0, 0, 0};  // miopalLastRouting
thread_local int g_lastFullRouting = 0;                // miopalLastFullRouting
thread_local int g_fault[3] = {0, 0, 0};               // miopalTestInjectFault: kind, unit, spin cap
// miopalTestSetLogicalDevices: > 0 = that many device ordinals, mapped round-robin onto the physical
// devices (the multi-device code of a one-process caller - a handle, a stream set, a mirror per
// ordinal, hipSetDevice per call - on a box with one GPU)
std::atomic<int> g_logicalDevices{0};

// HIP ordinals of the gfx950 devices, in order: the only ones the kernels were built for. Device ordinals of this
// library index THIS list (on a box with nothing but MI355X: the identity).
const std::vector<int>& usableDevices() {
    static const std::vector<int> list = [] {
        std::vector<int> v;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) return v;
        for (int d = 0; d < n; ++d) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) v.push_back(d);
        }
        return v;
    }();
    return list;
}
int physicalDeviceCount() { return (int)usableDevices().size(); }

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_lastError = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(MIOPAL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                               \
    } while (0)

#define RC_TRY(expr)          \
    do {                      \
        int _rc = (expr);     \
        if (_rc) return _rc;  \
    } while (0)

// The two events around a timed launch: destroyed unless they were handed to the workspace's list
// (the score pass has early returns between the first record and the hand-over).
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    EventPair() = default;
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    std::pair<hipEvent_t, hipEvent_t> release() {
        std::pair<hipEvent_t, hipEvent_t> p(e0, e1);
        e0 = e1 = nullptr;
        return p;
    }
};

struct PhaseTimer {
    bool on;
    double t0;
    static double now() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec + 1e-9 * ts.tv_nsec;
    }
    PhaseTimer() : on(tuned(Tune::PHASE_TIMING) != nullptr), t0(now()) {}
    void mark(const char* what) {
        if (!on) return;
        const double t = now();
        fprintf(stderr, "[miopal] %-28s %8.3f ms\n", what, (t - t0) * 1e3);
        t0 = t;
    }
};

// Streams that only carry uploads (database pieces, view construction) are created at the LOWEST
// priority: the runtime multiplexes streams onto four hardware queues per priority level, and every
// stream left at the default level makes it likelier that two workspaces' search streams share one.
inline hipError_t createUploadStream(hipStream_t* s) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
        (void)hipGetLastError();
        least = greatest = 0;
    }
    if (least > 0) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

constexpr int kLongTarget = 8192;          // longer targets always take the intra-sequence path
constexpr int64_t kDirBudgetOneLaunch = 8ll << 30;   // directions of every batch of a `full` search at once (host_full.inc)
constexpr int64_t kDirBudget = 2ll << 30;  // direction workspace: 2 x this per device-resident traceback batch, 1 x per host-built batch
constexpr int64_t kInt32Safe = 1ll << 29;
constexpr int kMaxDirectRecompute = 2048;  // lanes that left their range and are sent straight to int32: at least this many (see directLimit)
// A lane that owns a whole target walks its columns one after the other (about 0.8 us per
// column of 56 rows): whatever the number of targets, the lane-per-target kernels need
// (longest target) x that. Few targets of a one-strip query are done sooner by the
// wavefront-per-pair kernel, whose anti-diagonal step is ~10x shorter and which still has a
// wavefront for every pair at this count.
constexpr int64_t kSmallSearch = 4096;
constexpr size_t kParkedWorkspaceFloor = 4ull << 30;   // idle per-handle workspaces kept: this, or four times the database (host_types.inc)
constexpr size_t kMaxCachedViews = 64;                 // packed views per handle (beside the byte budget)

// malloc-backed byte buffer: grows without zero-filling, and its storage can be handed to the
// caller of the C ABI (who frees it with free()).
struct HostBytes {
    uint8_t* data = nullptr;
    size_t size = 0, cap = 0;
    // `data` is a buffer the caller of the C ABI lent for this search (miopalSearchFlatInto): written in
    // place when it is large enough, never freed or moved here - a search that needs more leaves it alone
    // and returns a buffer of its own
    bool lent = false;
    HostBytes() = default;
    HostBytes(const HostBytes&) = delete;
    HostBytes& operator=(const HostBytes&) = delete;
    ~HostBytes() {
        if (!lent) free(data);
    }
    void lend(uint8_t* buffer, size_t capacity) {
        if (!buffer || capacity == 0) return;
        data = buffer;
        cap = capacity;
        size = 0;
        lent = true;
    }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        want = std::max(want, cap + cap / 2);
        uint8_t* keep = nullptr;   // a lent buffer that is too small: its contents move, it stays the caller's
        if (lent) {
            keep = data;
            data = nullptr;
            lent = false;
            want += want / 4;   // (the next search of this kind fits the buffer it gets back)
        } else if (!data && want >= (8u << 20)) {
            // (a first large buffer too: lent back, it is judged against the NEXT search's worst case - a query of the
            // same length whose longest window rounds sixteen columns up must not find it too small)
            want += want / 4;
        }
        const bool hugePages = !tuned(Tune::NO_HUGEPAGE);
        if (!data && want >= (8u << 20) && hugePages) {
            // a large result buffer is written once, front to back: ask for huge pages so that
            // first touch costs tens of faults instead of tens of thousands
            void* p = nullptr;
            if (posix_memalign(&p, 2u << 20, want) == 0) {
                madvise(p, want, MADV_HUGEPAGE);
                data = (uint8_t*)p;
                if (keep && size) memcpy(data, keep, size);
                cap = want;
                return true;
            }
        }
        uint8_t* p = (uint8_t*)realloc(data, want);
        if (!p) {
            if (keep) {   // (as before the call)
                data = keep;
                lent = true;
            }
            return false;
        }
        if (keep && size) memcpy(p, keep, size);
        data = p;
        cap = want;
        return true;
    }
    bool resize(size_t n) {
        if (!reserve(std::max<size_t>(n, 1))) return false;
        size = n;
        return true;
    }
    bool append(const uint8_t* src, size_t n) {
        if (!reserve(size + n)) return false;
        memcpy(data + size, src, n);
        size += n;
        return true;
    }
    // the storage leaves for the caller (a lent buffer goes back the same way); `capacity`: its size in bytes
    uint8_t* release(size_t* capacity = nullptr) {
        uint8_t* p = data;
        if (capacity) *capacity = cap;
        data = nullptr;
        size = cap = 0;
        lent = false;
        return p;
    }
};

#include "host_workspace.inc"
#include "host_types.inc"
#include "host_upload.inc"
#include "host_view.inc"
#include "host_search.inc"
#include "host_handle.inc"
// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int miopalDeviceCount(void) {
    const int physical = physicalDeviceCount();
    const int logical = g_logicalDevices.load(std::memory_order_relaxed);
    return (logical > 0 && physical > 0) ? logical : physical;
}

int miopalTestSetLogicalDevices(int count) {
    if (count < 0 || count > 64) return fail(MIOPAL_ERR_BAD_ARGUMENT, "logical device count %d not in 0..64", count);
    g_logicalDevices.store(count, std::memory_order_relaxed);
    return 0;
}

int miopalSetTuning(const char* name, const char* value) {
    const int k = Tuning::find(name);
    if (k < 0) return fail(MIOPAL_ERR_BAD_ARGUMENT, "unknown tuning switch %s", name ? name : "(null)");
    // (the old string is left alone: a search on another thread may be reading it)
    Tuning::table().value[k].store(value ? strdup(value) : nullptr, std::memory_order_release);
    return 0;
}

const char* miopalGetTuning(const char* name) {
    const int k = Tuning::find(name);
    return k < 0 ? nullptr : tuned((Tune)k);
}

int miopalDbSetOption(MiopalDb* db, const char* name, int64_t value) {
    if (!db || !name) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null handle / option name");
    if (strcmp(name, "reserve_cus") == 0) {
        if (value < -1 || value > 4096) return fail(MIOPAL_ERR_BAD_ARGUMENT, "reserve_cus %lld out of range", (long long)value);
        db->optReserveCus.store((int)value, std::memory_order_relaxed);
        return 0;
    }
    if (strcmp(name, "small_search") == 0) {
        if (value < -1 || value > 1) return fail(MIOPAL_ERR_BAD_ARGUMENT, "small_search takes -1 (default), 0 or 1");
        db->optSmallSearch.store((int)value, std::memory_order_relaxed);
        return 0;
    }
    return fail(MIOPAL_ERR_BAD_ARGUMENT, "unknown handle option %s", name);
}

const char* miopalLastError(void) { return g_lastError.c_str(); }

int miopalDbCreate(MiopalDb** out, const unsigned char* const* sequences, const int* lengths,
                   int64_t count, int alphabetLength, int device) {
    return guarded([&]() -> int {
    if (!out || count < 0 || (count > 0 && (!sequences || !lengths)))
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad arguments to miopalDbCreate");
    if (alphabetLength <= 0 || alphabetLength > kMaxAlphabet)
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "alphabet length %d not in 1..32", alphabetLength);
    if (count >= INT32_MAX) return fail(MIOPAL_ERR_BAD_ARGUMENT, "too many sequences");
    std::vector<int64_t> offsets((size_t)count + 1, 0);
    for (int64_t k = 0; k < count; ++k) {
        if (lengths[k] < 0) return fail(MIOPAL_ERR_BAD_ARGUMENT, "negative sequence length");
        offsets[(size_t)k + 1] = offsets[(size_t)k] + lengths[k];
    }
    for (int64_t k = 0; k < count; ++k)
        if (lengths[k] > 0 && !sequences[k]) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null sequence %lld", (long long)k);
    ResidueSource src;
    src.sequences = sequences;
    return createCommon(out, src, std::move(offsets), count, alphabetLength, device);
    });
}

int miopalDbCreateFlat(MiopalDb** out, const unsigned char* residues, const int64_t* offsets,
                       int64_t count, int alphabetLength, int device) {
    return guarded([&]() -> int {
    if (!out || count < 0 || !offsets) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad arguments to miopalDbCreateFlat");
    if (alphabetLength <= 0 || alphabetLength > kMaxAlphabet)
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "alphabet length %d not in 1..32", alphabetLength);
    if (count >= INT32_MAX) return fail(MIOPAL_ERR_BAD_ARGUMENT, "too many sequences");
    std::vector<int64_t> off(offsets, offsets + count + 1);
    if (off[0] != 0) return fail(MIOPAL_ERR_BAD_ARGUMENT, "offsets must start at 0");
    for (int64_t k = 0; k < count; ++k)
        if (off[(size_t)k + 1] < off[(size_t)k] || off[(size_t)k + 1] - off[(size_t)k] > INT32_MAX)
            return fail(MIOPAL_ERR_BAD_ARGUMENT, "offsets must be non-decreasing");
    if (off[(size_t)count] > 0 && !residues) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null residues");
    ResidueSource src;
    src.flat = residues;
    return createCommon(out, src, std::move(off), count, alphabetLength, device);
    });
}

int miopalDbCreateSubset(MiopalDb** out, const MiopalDb* parent, const int64_t* indices, int64_t count) {
    return guarded([&]() -> int {
    if (!out) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null output handle");
    *out = nullptr;
    if (!parent) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null parent handle");
    if (count < 0 || (count > 0 && !indices) || count >= INT32_MAX) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad subset");
    std::vector<int64_t> offsets((size_t)count + 1, 0), srcStart((size_t)std::max<int64_t>(count, 1), 0);
    for (int64_t k = 0; k < count; ++k) {
        const int64_t id = indices[k];
        if (id < 0 || id >= parent->count) return fail(MIOPAL_ERR_BAD_ARGUMENT, "subset index %lld outside the database", (long long)id);
        srcStart[(size_t)k] = parent->offsets[(size_t)id];
        offsets[(size_t)k + 1] = offsets[(size_t)k] + (parent->offsets[(size_t)id + 1] - parent->offsets[(size_t)id]);
    }
    std::unique_ptr<MiopalDb> db;
    RC_TRY(newHandle(&db, parent->device));
    db->alphabet = parent->alphabet;
    db->count = count;
    db->total = offsets[(size_t)count];
    for (int64_t k = 0; k < count; ++k) db->maxLen = std::max(db->maxLen, offsets[(size_t)k + 1] - offsets[(size_t)k]);
    const size_t wantRes = (size_t)db->total + 64, wantOff = (size_t)(count + 1) * sizeof(int64_t);
    HIP_TRY(hipMalloc(&db->d_residues, wantRes));
    db->residueCap = wantRes;
    HIP_TRY(hipMalloc(&db->d_offsets, wantOff));
    db->offsetsCap = wantOff;
    RC_TRY(uploadOnce(db->device, db->d_offsets, offsets.data(), wantOff));
    if (count > 0) {
        // the residues never leave the device: gathered from the parent's resident copy
        // (on a stream of its own: the null stream would serialise against every blocking stream of the process)
        struct Scratch {
            int64_t* d_src = nullptr;
            hipStream_t stream = nullptr;
            ~Scratch() {
                if (stream) (void)hipStreamDestroy(stream);
                if (d_src) (void)hipFree(d_src);
            }
        } scratch;
        HIP_TRY(hipMalloc(&scratch.d_src, (size_t)count * sizeof(int64_t)));
        HIP_TRY(createUploadStream(&scratch.stream));
        RC_TRY(uploadOnce(db->device, scratch.d_src, srcStart.data(), (size_t)count * sizeof(int64_t)));
        hipError_t e = launchGatherSequences(parent->d_residues, scratch.d_src, db->d_offsets, count, db->d_residues, scratch.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(scratch.stream);
        if (e != hipSuccess) return fail(MIOPAL_ERR_HIP, "gathering the subset failed: %s", hipGetErrorString(e));
    }
    db->offsets = std::move(offsets);
    *out = db.release();
    return 0;
    });
}

void miopalDbDestroy(MiopalDb* db) { delete db; }

void miopalReleaseCaches(void) {
    std::vector<std::unique_ptr<MiopalDb>> handles;
    {
        SpareHandles& sp = spareHandles();
        std::lock_guard<std::mutex> g(sp.m);
        handles.swap(sp.idle);
    }
    handles.clear();
    std::vector<std::unique_ptr<UploadKit>> kits;
    {
        StagingPool& pool = stagingPool();
        std::lock_guard<std::mutex> g(pool.m);
        kits.swap(pool.free);
        pool.freeStreams.clear();
    }
    kits.clear();
    // (idle workspaces of every live handle)
    std::vector<MiopalDb*> none;
    MiopalDb::liveHandles(nullptr, 0, &none);
}

static int64_t releaseIdleWorkspaces(MiopalDb* db);


void MiopalDb::liveHandles(MiopalDb* db, int what, std::vector<MiopalDb*>* out) {
    // (never destroyed: handles parked in the spare pool are destroyed at exit, after function-local statics of
    // this kind would have been)
    static std::mutex& m = *new std::mutex;
    static std::vector<MiopalDb*>& live = *new std::vector<MiopalDb*>;
    std::lock_guard<std::mutex> g(m);
    if (what > 0) live.push_back(db);
    else if (what < 0) live.erase(std::remove(live.begin(), live.end(), db), live.end());
    else if (out) {
        // (under the lock: a handle cannot be destroyed while its idle workspaces are released)
        for (MiopalDb* h : live) (void)releaseIdleWorkspaces(h);
    }
}

static int64_t releaseIdleWorkspaces(MiopalDb* db) {
    std::vector<std::unique_ptr<Workspace>> idle;
    {
        std::lock_guard<std::mutex> g(db->wsMutex);
        idle.swap(db->ownedFree);
    }
    int64_t bytes = 0;
    for (const auto& w : idle) bytes += (int64_t)w->bytes();
    {
        // (the kernel timings of the last profiled search live in its workspace: miopalLastKernelTime)
        std::lock_guard<std::mutex> g(db->timingMutex);
        for (const auto& w : idle)
            if (db->lastTimed == w.get()) db->lastTimed = nullptr;
    }
    (void)hipSetDevice(db->device);
    idle.clear();   // (workspaces in use by running searches are not in the list: they come back and are kept)
    return bytes;
}

int64_t miopalDbReleaseWorkspaces(MiopalDb* db) {
    if (!db) return 0;
    return releaseIdleWorkspaces(db);
}

int64_t miopalDbCount(const MiopalDb* db) { return db ? db->count : 0; }
int64_t miopalDbTotalLength(const MiopalDb* db) { return db ? db->total : 0; }
int64_t miopalDbDeviceBytes(const MiopalDb* db) {
    if (!db) return 0;
    MiopalDb* m = const_cast<MiopalDb*>(db);
    int64_t t = db->total + 64 + (db->count + 1) * 8;
    {
        std::lock_guard<std::mutex> g(m->viewMutex);
        for (auto& v : m->views) t += v.view ? (int64_t)v.view->deviceBytes : 0;
    }
    return t;
}

void miopalSetProfiling(MiopalDb* db, int enabled) {
    if (db) db->profiling.store(enabled ? 1 : 0);
}

void miopalTestInjectFault(int kind, int unit, int spinCap) {
    g_fault[0] = kind;
    g_fault[1] = unit;
    g_fault[2] = spinCap;
}

int miopalSelfTest(int which) {
    return guarded([&]() -> int {
        if (which == 2) {
            // the operations' way back from two bits each (unpack::, host_workspace.inc): every code path of this CPU -
            // bit by bit, the table, pdep, AVX-512 VBMI, several threads, the crew started beforehand - against the
            // definition, on ranges that start and end anywhere
            const int64_t n = (9 << 20) + 37;
            std::vector<uint8_t> ops((size_t)n), packed((size_t)(n + 3) / 4 + 64, 0), out((size_t)n + 64);
            uint32_t x = 12345;
            for (int64_t p = 0; p < n; ++p) {
                x = x * 1664525u + 1013904223u;
                ops[(size_t)p] = (uint8_t)(x >> 30);
                packed[(size_t)(p >> 2)] |= (uint8_t)(ops[(size_t)p] << (2 * (p & 3)));
            }
            auto same = [&](int64_t from, int64_t to) {
                for (int64_t p = from; p < to; ++p)
                    if (out[(size_t)p] != ops[(size_t)p]) return false;
                return true;
            };
            const int64_t cuts[][2] = {{0, n}, {1, n - 1}, {5, 77}, {63, 64}, {64, 64}, {7, 8 << 20}, {4097, n}, {n - 3, n}};
            for (const auto& c : cuts) {
                std::fill(out.begin(), out.end(), 9);
                unpack::ops(out.data(), packed.data(), c[0], c[1], 8);
                if (!same(c[0], c[1])) return 21;
                if (c[0] > 0 && out[(size_t)c[0] - 1] != 9) return 22;   // (nothing written outside the range)
                if (out[(size_t)c[1]] != 9) return 23;
                std::fill(out.begin(), out.end(), 9);
                unpack::rangePlain(out.data(), packed.data(), c[0], c[1]);
                if (!same(c[0], c[1]) || out[(size_t)c[1]] != 9) return 24;
                std::fill(out.begin(), out.end(), 9);
                {
                    unpack::Crew crew(7);
                    crew.run(out.data(), packed.data(), c[0], c[1]);
                }
                if (!same(c[0], c[1])) return 100 + (int)(&c - &cuts[0]);   // (100 + the range's number: the crew left operations out)
                if (out[(size_t)c[1]] != 9 || (c[0] > 0 && out[(size_t)c[0] - 1] != 9)) return 200 + (int)(&c - &cuts[0]);
            }
            { unpack::Crew unused(3); }   // (a crew that is never run goes away quietly)
            return 0;
        }
        if (which != 1) return -1;
        // (no device call on the way: the handle is never filled, the builders are injected)
        std::unique_ptr<MiopalDb> db(new MiopalDb());
        std::shared_ptr<View> got;
        int calls = 0;
        int rc = getViewWith(db.get(), 0, 10, 0, &got, [&](std::shared_ptr<View>*) -> int {
            ++calls;
            throw std::bad_alloc();
        });
        if (rc != MIOPAL_ERR_INTERNAL) return 1;
        if (calls != 2) return 2;                  // built, views evicted, built once more
        if (!db->views.empty()) return 3;          // no placeholder left behind
        // a thread that waits for the very slice while its builder fails must come back too
        std::atomic<int> stage{0};
        int rcWaiter = -1;
        std::thread first([&] {
            std::shared_ptr<View> v;
            (void)getViewWith(db.get(), 0, 10, 0, &v, [&](std::shared_ptr<View>*) -> int {
                // (called twice: a failed build is retried once after the idle views were dropped)
                int expected = 0;
                stage.compare_exchange_strong(expected, 1);
                for (int spins = 0; stage.load() < 2 && spins < 5000; ++spins)
                    std::this_thread::sleep_for(std::chrono::milliseconds(1));
                std::this_thread::sleep_for(std::chrono::milliseconds(20));   // the waiter is in wait() by now
                throw std::runtime_error("injected");
            });
        });
        for (int spins = 0; stage.load() < 1 && spins < 5000; ++spins) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        std::thread waiter([&] {
            std::shared_ptr<View> v;
            stage.store(2);
            rcWaiter = getViewWith(db.get(), 0, 10, 0, &v, [&](std::shared_ptr<View>* out) -> int {
                out->reset(new View());
                return 0;
            });
        });
        first.join();
        waiter.join();
        if (rcWaiter != 0) return 4;
        if (db->views.size() != 1 || db->views.front().building) return 5;
        db->views.clear();
        return 0;
    });
}

void miopalLastRouting(int64_t counts[4]) {
    if (!counts) return;
    for (int k = 0; k < 4; ++k) counts[k] = g_lastRouting[k];
}

int miopalLastFullRouting(void) { return g_lastFullRouting; }

int miopalLastKernelTime(MiopalDb* db, float* ms) {
    if (!db || !ms) return 0;
    std::lock_guard<std::mutex> g(db->timingMutex);
    *ms = 0.f;
    Workspace* ws = db->lastTimed;
    if (!ws) return 0;
    int n = 0;
    for (auto& ev : ws->timings) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess &&
            hipEventElapsedTime(&t, ev.first, ev.second) == hipSuccess) {
            *ms += t;
            ++n;
        }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ws->timings.clear();
    return n;
}

int miopalSearchDeviceScores(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                             int gapExt, const int* scoreMatrix, int alphabetLength, int mode,
                             int64_t start, int64_t end, int* deviceScores, void* stream) {
    return guarded([&]() -> int {
    RC_TRY(validate(db, query, queryLength, scoreMatrix, alphabetLength, OPAL_SEARCH_SCORE, mode, start, end));
    if (end == start) return 0;
    if (!deviceScores) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null device score buffer");
    HIP_TRY(hipSetDevice(db->device));
    WorkspaceLease lease(db);
    RC_TRY(lease.acquireExternal((hipStream_t)stream));
    Search s{db, lease.ws, (hipStream_t)stream, query, queryLength, gapOpen, gapExt, alphabetLength,
             OPAL_SEARCH_SCORE, mode, scoreMatrix, start, end, end - start};
    RC_TRY(s.prepare());
    RC_TRY(s.scorePass((int32_t*)deviceScores, nullptr, nullptr));
    if (s.d_stripError) {
        // Long pairs went through the (pair, strip) units of the int32 kernel, whose units can give up
        // waiting (never seen): the only searches of this entry point that synchronise, so that a
        // partial answer cannot pass for a score.
        RC_TRY(lease.ws->stageDownload(&s.stripErrorHost, s.d_stripError, sizeof(int)));
        RC_TRY(lease.ws->finishDownloads());
        RC_TRY(s.checkStripError());
    }
    return 0;
    });
}

#include "host_full.inc"
int miopalSearch(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen, int gapExt,
                 const int* scoreMatrix, int alphabetLength, int searchType, int mode, int64_t start,
                 int64_t end, int* score, int* endTarget, int* endQuery, int* startTarget,
                 int* startQuery, unsigned char** alignment, int* alignmentLength) {
    return guarded([&]() -> int {
    return searchImpl(db, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, searchType, mode,
                      start, end, score, endTarget, endQuery, startTarget, startQuery, alignment,
                      alignmentLength, nullptr, nullptr);
    });
}

int miopalSearchFlat(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen, int gapExt,
                     const int* scoreMatrix, int alphabetLength, int searchType, int mode, int64_t start,
                     int64_t end, int* score, int* endTarget, int* endQuery, int* startTarget,
                     int* startQuery, unsigned char** operations, int64_t* operationOffsets) {
    return guarded([&]() -> int {
    HostBytes ops;
    const bool full = searchType == OPAL_SEARCH_ALIGNMENT;
    if (full && !operations) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null alignment outputs");
    if (operations) *operations = nullptr;
    RC_TRY(searchImpl(db, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, searchType, mode,
                      start, end, score, endTarget, endQuery, startTarget, startQuery, nullptr, nullptr,
                      full ? &ops : nullptr, full ? operationOffsets : nullptr));
    if (full && end > start) {
        if (!ops.data && !ops.resize(0)) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
        *operations = ops.release();
    }
    return 0;
    });
}

int miopalSearchFlatInto(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen, int gapExt,
                         const int* scoreMatrix, int alphabetLength, int searchType, int mode, int64_t start,
                         int64_t end, int* score, int* endTarget, int* endQuery, int* startTarget,
                         int* startQuery, unsigned char** operations, int64_t* operationsCapacity,
                         int64_t* operationOffsets) {
    return guarded([&]() -> int {
    HostBytes ops;
    const bool full = searchType == OPAL_SEARCH_ALIGNMENT;
    if (full && (!operations || !operationsCapacity)) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null alignment outputs");
    if (full && *operations && *operationsCapacity > 0) ops.lend(*operations, (size_t)*operationsCapacity);
    RC_TRY(searchImpl(db, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, searchType, mode,
                      start, end, score, endTarget, endQuery, startTarget, startQuery, nullptr, nullptr,
                      full ? &ops : nullptr, full ? operationOffsets : nullptr));
    if (full && end > start) {
        if (!ops.data && !ops.resize(0)) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
        size_t capacity = 0;
        *operations = ops.release(&capacity);
        *operationsCapacity = (int64_t)capacity;
    }
    return 0;
    });
}

int miopalSearchResults(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen,
                        int gapExt, const int* scoreMatrix, int alphabetLength,
                        OpalSearchResult* results[], int searchType, int mode, int overflowMethod,
                        int64_t start, int64_t end) {
    return guarded([&]() -> int {
    (void)overflowMethod;
    if (!db) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null database handle");
    if (end > db->count) end = db->count;
    const int64_t n = end - start;
    if (n <= 0) return n < 0 ? fail(MIOPAL_ERR_BAD_ARGUMENT, "bad slice") : 0;
    if (!results) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null results");
    // (not a vector: a million zeros written first cost as much as the copy that overwrites them)
    std::unique_ptr<int[]> scoreOwner(new int[(size_t)n]);
    int* const score = scoreOwner.get();
    std::vector<int> et, eq, st, sq, alen;
    std::vector<unsigned char*> aln;
    if (searchType >= OPAL_SEARCH_SCORE_END) { et.resize((size_t)n); eq.resize((size_t)n); }
    if (searchType == OPAL_SEARCH_ALIGNMENT) { st.resize((size_t)n); sq.resize((size_t)n); alen.resize((size_t)n); aln.resize((size_t)n, nullptr); }
    int rc = miopalSearch(db, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, searchType,
                          mode, start, end, score, et.empty() ? nullptr : et.data(),
                          eq.empty() ? nullptr : eq.data(), st.empty() ? nullptr : st.data(),
                          sq.empty() ? nullptr : sq.data(), aln.empty() ? nullptr : aln.data(),
                          alen.empty() ? nullptr : alen.data());
    if (rc) {
        for (auto p : aln) free(p);
        return rc;
    }
    PhaseTimer pt;
    // (a million scattered 40-byte records behind a million pointers: memory latency, more threads help)
    const int nSlices = hostThreads((size_t)n, 65536, 8);
    parallelSlices(nSlices, [&](int t) {
        for (int64_t k = n * t / nSlices; k < n * (t + 1) / nSlices; ++k) {
            OpalSearchResult* r = results[k];
            r->scoreSet = 1;
            r->score = score[(size_t)k];
            if (searchType >= OPAL_SEARCH_SCORE_END) {
                r->endLocationTarget = et[(size_t)k];
                r->endLocationQuery = eq[(size_t)k];
            }
            if (searchType == OPAL_SEARCH_ALIGNMENT) {
                r->startLocationTarget = st[(size_t)k];
                r->startLocationQuery = sq[(size_t)k];
                r->alignment = alen[(size_t)k] > 0 ? aln[(size_t)k] : nullptr;
                if (alen[(size_t)k] == 0) free(aln[(size_t)k]);
                r->alignmentLength = alen[(size_t)k];
            }
        }
    });
    pt.mark("result structs");
    return 0;
    });
}

// ---- opal.h ---------------------------------------------------------------
void opalInitSearchResult(OpalSearchResult* r) {
    r->scoreSet = 0;
    r->score = 0;
    r->endLocationTarget = r->endLocationQuery = -1;
    r->startLocationTarget = r->startLocationQuery = -1;
    r->alignment = nullptr;
    r->alignmentLength = 0;
}

int opalSearchResultIsEmpty(const OpalSearchResult r) { return !r.scoreSet; }

void opalSearchResultSetScore(OpalSearchResult* r, int score) {
    r->scoreSet = 1;
    r->score = score;
}

int opalSearchDatabase(unsigned char query[], int queryLength, unsigned char* db[], int dbLength,
                       int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix,
                       int alphabetLength, OpalSearchResult* results[], const int searchType, int mode,
                       int overflowMethod) {
    return guarded([&]() -> int {
    if (dbLength <= 0) return 0;
    int device = 0;
    if (const char* env = tuned(Tune::DEVICE)) device = atoi(env);
    const bool verbose = tuned(Tune::VERBOSE) != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!db || !dbSeqLengths) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad arguments to opalSearchDatabase");
    if (alphabetLength <= 0 || alphabetLength > kMaxAlphabet)
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "alphabet length %d not in 1..32", alphabetLength);
    // offsets = prefix sums of the lengths, in slices worked on side by side (a million targets: 12 MB read)
    std::vector<int64_t> offsets((size_t)dbLength + 1);
    {
        const int nSlices = hostThreads((size_t)dbLength, 65536);
        std::vector<int64_t> sliceSum((size_t)nSlices + 1, 0);
        std::vector<int> sliceBad((size_t)nSlices, 0);
        auto lo = [&](int t) { return (int)((int64_t)dbLength * t / nSlices); };
        parallelSlices(nSlices, [&](int t) {
            int64_t sum = 0;
            for (int k = lo(t); k < lo(t + 1); ++k) {
                if (dbSeqLengths[k] < 0) { sliceBad[(size_t)t] = 1; return; }
                if (dbSeqLengths[k] > 0 && !db[k]) { sliceBad[(size_t)t] = 2; return; }
                sum += dbSeqLengths[k];
            }
            sliceSum[(size_t)t + 1] = sum;
        });
        for (int t = 0; t < nSlices; ++t) {
            if (sliceBad[(size_t)t] == 1) return fail(MIOPAL_ERR_BAD_ARGUMENT, "negative sequence length");
            if (sliceBad[(size_t)t] == 2) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null sequence in the database");
            sliceSum[(size_t)t + 1] += sliceSum[(size_t)t];
        }
        parallelSlices(nSlices, [&](int t) {
            int64_t at = sliceSum[(size_t)t];
            for (int k = lo(t); k < lo(t + 1); ++k) {
                offsets[(size_t)k] = at;
                at += dbSeqLengths[k];
            }
        });
        offsets[(size_t)dbLength] = sliceSum[(size_t)nSlices];
    }
    // a handle of an earlier call, or a new one
    std::unique_ptr<MiopalDb> h;
    {
        SpareHandles& sp = spareHandles();
        std::lock_guard<std::mutex> g(sp.m);
        for (size_t k = sp.idle.size(); k-- > 0;)
            if (sp.idle[k]->device == device) {
                h = std::move(sp.idle[k]);
                sp.idle.erase(sp.idle.begin() + (long)k);
                break;
            }
    }
    if (!h) RC_TRY(newHandle(&h, device));
    ResidueSource src;
    src.sequences = (const unsigned char* const*)db;
    // Round 4, opt-in (MIOPAL_SEARCH_UNDER_UPLOAD=1): the search UNDER the upload. The lengths are all a packed
    // view's lists need, so the database is cut into a few contiguous segments (by residues: 30 / 30 / 25 / 15 %,
    // the last one short); a segment is packed and searched, and its result structs are written, as soon as its
    // residues are on the device, while the later segments still cross PCIe. Measured (1M x 300, profiles/
    // r04_search_under_upload.txt): 10.3-10.4 ms per call against 10.5-10.6 with the search behind the upload on one
    // box, 11.2-11.9 against 11.2-12.2 on another - the host-to-device copies slow down beside the search kernels by
    // what the overlap saves (upload 7.3 -> 9.2 ms), whatever the number of host threads or CUs left free. Not the
    // default for that reason; kept, with its test, because the arithmetic is right for a host whose copies do not
    // share the device with the kernels. Large databases of short targets only (those whose view lists are built
    // ahead: fillHandle); targets are independent, so the results are those of one search of the whole database
    // (src/pyopal/_align.py:150-170 cuts the same way over its threads).
    int64_t longest = 0;
    if (tuned(Tune::SEARCH_UNDER_UPLOAD)) {
        const int nSlices = hostThreads((size_t)dbLength, 65536);
        std::vector<int> sliceMax((size_t)nSlices, 0);
        parallelSlices(nSlices, [&](int t) {
            int m = 0;
            for (int k = (int)((int64_t)dbLength * t / nSlices); k < (int)((int64_t)dbLength * (t + 1) / nSlices); ++k) m = std::max(m, dbSeqLengths[k]);
            sliceMax[(size_t)t] = m;
        });
        for (int m : sliceMax) longest = std::max<int64_t>(longest, m);
    }
    const int64_t totalResidues = offsets[(size_t)dbLength];
    const bool under = tuned(Tune::SEARCH_UNDER_UPLOAD) && dbLength >= 262144 && totalResidues >= (96ll << 20) &&
                       longest <= 256 + 128 && queryLength > 0 && !tuned(Tune::NO_VIEW_PREFETCH);
    int rc = 0;
    auto t1 = t0;
    if (under) {
        std::vector<int64_t> bounds{0};
        for (double share : {0.30, 0.60, 0.85}) {
            const int64_t goal = (int64_t)(share * (double)totalResidues);
            int64_t k = std::upper_bound(offsets.begin(), offsets.end(), goal) - offsets.begin() - 1;
            k = std::min<int64_t>(std::max<int64_t>(k, bounds.back()), dbLength);
            if (k > bounds.back()) bounds.push_back(k);
        }
        if (bounds.back() < dbLength) bounds.push_back(dbLength);
        std::vector<int64_t> segEndByte;
        for (size_t k = 1; k < bounds.size(); ++k) segEndByte.push_back(offsets[(size_t)bounds[k]]);
        UploadProgress progress;
        bool ready = false;
        int fillRc = 0;
        std::string fillError;
        MiopalDb* const hd = h.get();
        std::thread filler([&, hd] {
            // (errors are per thread: this one's message is handed to the caller's)
            fillRc = guarded([&] { return fillHandle(hd, src, std::move(offsets), dbLength, alphabetLength, true, &bounds, &progress, &ready); });
            if (fillRc) fillError = g_lastError;
            progress.finish();
        });
        struct Join {
            std::thread& t;
            ~Join() { if (t.joinable()) t.join(); }
        } joinFiller{filler};
        {
            std::unique_lock<std::mutex> g(progress.m);
            progress.cv.wait(g, [&] { return ready || progress.finished; });
        }
        for (size_t k = 0; k + 1 < bounds.size() && rc == 0; ++k) {
            if (!progress.waitFor((size_t)segEndByte[k])) break;   // (the upload ended early: its error is reported below)
            if (k == 0) (void)hipSetDevice(h->device);
            rc = miopalSearchResults(hd, query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength,
                                     results + bounds[k], searchType, mode, overflowMethod, bounds[k], bounds[k + 1]);
        }
        filler.join();
        t1 = std::chrono::steady_clock::now();
        if (fillRc) {
            g_lastError = fillError;
            rc = fillRc;
        }
    } else {
        rc = fillHandle(h.get(), src, std::move(offsets), dbLength, alphabetLength, /*prefetchView=*/true);
        t1 = std::chrono::steady_clock::now();
        if (rc == 0)
            rc = miopalSearchResults(h.get(), query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, results,
                                     searchType, mode, overflowMethod, 0, dbLength);
    }
    const auto t2 = std::chrono::steady_clock::now();
    {
        // view lists built ahead that no search of this call asked for (another overlap, a search that built its own
        // before the list was ready, a loop that stopped on an error) would otherwise keep their device blocks, and
        // stay out of handleDeviceBytes, until the handle is filled again
        std::lock_guard<std::mutex> g(h->viewMutex);
        h->prefetched.clear();
    }
    if (rc == 0) {
        int64_t keepMb = 4096;
        if (const char* env = tuned(Tune::SPARE_HANDLE_MB)) keepMb = atoll(env);
        if (handleDeviceBytes(h.get()) <= (keepMb << 20)) {
            SpareHandles& sp = spareHandles();
            std::lock_guard<std::mutex> g(sp.m);
            if (sp.idle.size() < kSpareHandles) sp.idle.emplace_back(std::move(h));
        }
    }
    if (h) {
        (void)hipSetDevice(h->device);
        h.reset();
    }
    if (verbose) {
        const auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "miopal: opalSearchDatabase: upload %.1f ms, search + results %.1f ms, release %.1f ms\n",
                ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
    }
    return rc;
    });
}

int opalSearchDatabaseCharSW(unsigned char query[], int queryLength, unsigned char** db, int dbLength,
                             int dbSeqLengths[], int gapOpen, int gapExt, int* scoreMatrix,
                             int alphabetLength, OpalSearchResult* results[]) {
    return opalSearchDatabase(query, queryLength, db, dbLength, dbSeqLengths, gapOpen, gapExt, scoreMatrix,
                              alphabetLength, results, OPAL_SEARCH_SCORE, OPAL_MODE_SW, OPAL_OVERFLOW_BUCKETS);
}

}  // extern "C"
