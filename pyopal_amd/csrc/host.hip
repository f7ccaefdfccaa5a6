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
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

using namespace miopal;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_lastError;
thread_local int64_t g_lastRouting[4] = {0, 0, 0, 0};  // miopalLastRouting
thread_local int g_fault[3] = {0, 0, 0};               // miopalTestInjectFault: kind, unit, spin cap

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
    PhaseTimer() : on(getenv("MIOPAL_PHASE_TIMING") != nullptr), t0(now()) {}
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
constexpr int64_t kDirBudget = 2ll << 30;  // direction workspace: 2 x this per device-resident traceback batch, 1 x per host-built batch
constexpr int64_t kInt32Safe = 1ll << 29;
constexpr int kMaxDirectRecompute = 2048;  // lanes that left their range and are sent straight to int32: at least this many (see directLimit)
// A lane that owns a whole target walks its columns one after the other (about 0.8 us per
// column of 56 rows): whatever the number of targets, the lane-per-target kernels need
// (longest target) x that. Few targets of a one-strip query are done sooner by the
// wavefront-per-pair kernel, whose anti-diagonal step is ~10x shorter and which still has a
// wavefront for every pair at this count.
constexpr int64_t kSmallSearch = 4096;
constexpr size_t kParkedWorkspaceBytes = 64ull << 30;  // idle per-handle workspaces kept at most
constexpr size_t kMaxCachedViews = 64;                 // packed views per handle (beside the byte budget)

// malloc-backed byte buffer: grows without zero-filling, and its storage can be handed to the
// caller of the C ABI (who frees it with free()).
struct HostBytes {
    uint8_t* data = nullptr;
    size_t size = 0, cap = 0;
    HostBytes() = default;
    HostBytes(const HostBytes&) = delete;
    HostBytes& operator=(const HostBytes&) = delete;
    ~HostBytes() { free(data); }
    bool reserve(size_t want) {
        if (want <= cap) return true;
        want = std::max(want, cap + cap / 2);
        static const bool hugePages = !getenv("MIOPAL_NO_HUGEPAGE");
        if (!data && want >= (8u << 20) && hugePages) {
            // a large result buffer is written once, front to back: ask for huge pages so that
            // first touch costs tens of faults instead of tens of thousands
            void* p = nullptr;
            if (posix_memalign(&p, 2u << 20, want) == 0) {
                madvise(p, want, MADV_HUGEPAGE);
                data = (uint8_t*)p;
                cap = want;
                return true;
            }
        }
        uint8_t* p = (uint8_t*)realloc(data, want);
        if (!p) return false;
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
    uint8_t* release() {
        uint8_t* p = data;
        data = nullptr;
        size = cap = 0;
        return p;
    }
};

// ---------------------------------------------------------------------------
// per-stream workspace: growable device buffers reused in stream order
// ---------------------------------------------------------------------------
enum Slot {
    kQuery, kMatrix, kProfile, kViewScore, kViewOvf, kCounter, kBoundary0, kBoundary1,
    kScore, kEndI, kEndJ, kJobs, kPairB0, kPairB1, kAuxJobs, kAuxPairB0, kAuxPairB1, kRScore, kRI, kRJ, kDirs, kOps, kOpsOff,
    kOpsLen, kOvfHost, kWorkCounter, kViewEndI, kViewEndJ, kStartQ, kStartT, kMismatch, kCompactOps, kTraceScore, kOpsTotals, kSortBins, kSortedJobs, kKeys, kHeadWaves, kHeadDirs, kUnitState, kUnitPartial, kStripKeys,
    kPairStripState, kPairStripPartial, kPairStripSpare, kAuxPairStripState, kAuxPairStripPartial, kPairStripError, kSlots
};

struct Workspace {
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    void* buf[kSlots] = {};
    size_t cap[kSlots] = {};
    std::mutex busy;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timings;  // dominant-kernel launches
    // side stream for the intra-sequence recompute that runs beside the inter-sequence kernel
    hipStream_t aux = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;

    // Results go to the host through a pinned staging buffer: copying straight into the
    // caller's pageable arrays makes the runtime pin and unpin those pages on every call
    // (tens of milliseconds on fresh memory), a pinned bounce buffer does not.
    void* pinned = nullptr;
    size_t pinnedCap = 0, pinnedUsed = 0;
    uint64_t stagingGeneration = 0;   // moves whenever the staging buffer is drained (its contents are void then)
    struct Pending { void* dst; size_t off, bytes; };
    std::vector<Pending> pending;
    std::vector<int32_t> hostScratchA, hostScratchB;  // per-target host arrays of a full search

    // Large results land in pages the caller has not touched yet; a few threads fault them in
    // side by side.
    static void copyOut(void* dst, const char* src, size_t bytes) {
        constexpr size_t kPiece = 4u << 20;
        if (bytes < 2 * kPiece) {
            memcpy(dst, src, bytes);
            return;
        }
        const int nThreads = (int)std::min<size_t>(4, bytes / kPiece);
        const size_t share = ((bytes / nThreads) + 4095) & ~(size_t)4095;
        std::vector<std::thread> pool;
        for (int t = 1; t < nThreads; ++t) {
            const size_t lo = share * t, hi = std::min(bytes, lo + share);
            if (lo >= hi) continue;
            try {
                pool.emplace_back([=] { memcpy((char*)dst + lo, src + lo, hi - lo); });
            } catch (const std::exception&) {   // no thread to be had: this one copies the share
                memcpy((char*)dst + lo, src + lo, hi - lo);
            }
        }
        memcpy(dst, src, std::min(bytes, share));
        for (auto& th : pool) th.join();
    }

    // Every per-search transfer goes through this pinned buffer, in both directions. Copying
    // from or to the caller's pageable memory makes the runtime pin those pages for the
    // transfer; when the caller later frees or trims that memory (a result buffer, a heap
    // that shrinks) the driver has to quiesce the process's queues to drop the mapping -
    // measured as a 20-30 ms stall of the NEXT search.
    int finishDownloads() {
        if (pending.empty() && pinnedUsed == 0) return 0;
        HIP_TRY(hipStreamSynchronize(stream));
        if (aux) HIP_TRY(hipStreamSynchronize(aux));
        for (const Pending& p : pending) copyOut(p.dst, (const char*)pinned + p.off, p.bytes);
        pending.clear();
        pinnedUsed = 0;
        ++stagingGeneration;
        return 0;
    }
    // A search that fails after stageDownload() leaves entries that point at the caller's (or the
    // failed call's local) memory: they must never be copied out by a later search.
    void abandonDownloads() {
        if (pending.empty()) return;
        (void)hipStreamSynchronize(stream);
        if (aux) (void)hipStreamSynchronize(aux);
        pending.clear();
        pinnedUsed = 0;
        ++stagingGeneration;
    }
    // room for `bytes` more in the staging buffer (drains it, and grows it, when needed)
    int reserveStaging(size_t aligned) {
        if (pinnedUsed + aligned <= pinnedCap) return 0;
        RC_TRY(finishDownloads());
        if (aligned > pinnedCap) {
            if (pinned) HIP_TRY(hipHostFree(pinned));
            pinned = nullptr;
            pinnedCap = 0;
            const size_t want = aligned + aligned / 4 + (1u << 20);
            HIP_TRY(hipHostMalloc(&pinned, want, hipHostMallocDefault));
            pinnedCap = want;
        }
        return 0;
    }
    int stageDownload(void* dst, const void* deviceSrc, size_t bytes) {
        if (bytes == 0) return 0;
        const size_t aligned = (bytes + 255) & ~(size_t)255;
        RC_TRY(reserveStaging(aligned));
        HIP_TRY(hipMemcpyAsync((char*)pinned + pinnedUsed, deviceSrc, bytes, hipMemcpyDeviceToHost, stream));
        pending.push_back({dst, pinnedUsed, bytes});
        pinnedUsed += aligned;
        return 0;
    }
    // host -> device on `on` (the workspace's stream or its side stream); `src` may be reused
    // as soon as the call returns
    int stageUpload(void* deviceDst, const void* src, size_t bytes, hipStream_t on) {
        if (bytes == 0) return 0;
        const size_t aligned = (bytes + 255) & ~(size_t)255;
        RC_TRY(reserveStaging(aligned));
        memcpy((char*)pinned + pinnedUsed, src, bytes);
        HIP_TRY(hipMemcpyAsync(deviceDst, (const char*)pinned + pinnedUsed, bytes, hipMemcpyHostToDevice, on));
        pinnedUsed += aligned;
        return 0;
    }

    int ensureAux() {
        if (aux) return 0;
        // A stream of its own PRIORITY: the runtime multiplexes the streams of a process onto four
        // hardware queues per priority level, and two streams that share a queue run in order. With
        // a handful of streams alive (workspaces, view construction, the upload pool) the side stream
        // and its workspace's main stream sometimes landed on one queue: the kernel that should run
        // BESIDE the packed one ran after it (log-normal database, NW at Q = 150: 7.0 instead of 4.0 ms).
        // The work on this stream is the launch's critical path anyway.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
            (void)hipGetLastError();
            least = greatest = 0;
        }
        if (greatest != least) HIP_TRY(hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, greatest));
        else HIP_TRY(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&evJoin, hipEventDisableTiming));
        return 0;
    }

    int get(int slot, size_t bytes, void** out) {
        if (bytes == 0) bytes = 16;
        if (cap[slot] < bytes) {
            if (buf[slot]) {
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipFree(buf[slot]));
                buf[slot] = nullptr;
                cap[slot] = 0;
            }
            size_t want = bytes + bytes / 8 + 256;
            HIP_TRY(hipMalloc(&buf[slot], want));
            cap[slot] = want;
            if (want > (512u << 20) && getenv("MIOPAL_VERBOSE"))
                fprintf(stderr, "miopal: workspace slot %d grows to %zu MiB\n", slot, want >> 20);
        }
        *out = buf[slot];
        return 0;
    }
    // Like get(), but an allocation the device cannot serve is not an error: returns false
    // (the caller asks for less). The slack of get() is left out: these are the big buffers.
    bool tryGet(int slot, size_t bytes, void** out) {
        if (cap[slot] >= bytes) {
            *out = buf[slot];
            return true;
        }
        if (buf[slot]) {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(buf[slot]);
            buf[slot] = nullptr;
            cap[slot] = 0;
        }
        if (hipMalloc(&buf[slot], bytes) != hipSuccess) {
            (void)hipGetLastError();  // out of memory is handled by the caller
            buf[slot] = nullptr;
            return false;
        }
        cap[slot] = bytes;
        if (bytes > (512u << 20) && getenv("MIOPAL_VERBOSE"))
            fprintf(stderr, "miopal: workspace slot %d grows to %zu MiB\n", slot, bytes >> 20);
        *out = buf[slot];
        return true;
    }
    size_t bytes() const {
        size_t t = 0;
        for (size_t c : cap) t += c;
        return t;
    }
    ~Workspace() {
        for (auto& ev : timings) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        for (void* p : buf)
            if (p) (void)hipFree(p);
        if (pinned) (void)hipHostFree(pinned);
        if (evFork) (void)hipEventDestroy(evFork);
        if (evJoin) (void)hipEventDestroy(evJoin);
        if (aux) (void)hipStreamDestroy(aux);
        if (ownsStream && stream) (void)hipStreamDestroy(stream);
    }
};

// ---------------------------------------------------------------------------
// packed view of a database slice
// ---------------------------------------------------------------------------
struct View {
    int64_t start = 0, end = 0;
    int overlap = 0;                 // > 0: long targets are cut into windows overlapping by this much
    int stride = 0;                  // ... starting every `stride` residues
    int32_t* d_segStart = nullptr;   // view position -> first residue of its window (segmented views)
    int nPacked = 0;                 // targets in the packed groups
    int nGroups = 0;
    int maxPackedLen = 0;
    std::vector<int> groupChunksHost;  // 4-column chunks per group (longest group first)
    int64_t totalChunks = 0;
    std::vector<int32_t> ids;        // view position -> database index (packed part)
    std::vector<int32_t> longIds;    // targets always handled by the intra-sequence kernel
    int32_t* d_ids = nullptr;
    int32_t* d_lens = nullptr;       // target lengths in view order, padded to whole groups
    uint2* d_pack = nullptr;
    int64_t* d_groupOff = nullptr;
    int* d_groupChunks = nullptr;
    int64_t* d_boundaryOff = nullptr;
    size_t deviceBytes = 0;
    void* d_meta = nullptr;          // one allocation behind d_ids .. d_boundaryOff (and the pack kernel's prefix)
    bool packPending = false;        // lists built and uploaded, residues not packed yet (prefetched view)
    PackArgs pendingPack{};
    size_t packCap = 0, metaCap = 0; // sizes of the two allocations (a refilled handle re-uses them)
    ~View() {
        if (d_pack) (void)hipFree(d_pack);
        if (d_meta) (void)hipFree(d_meta);
    }
};

}  // namespace

struct MiopalDb {
    int device = 0;
    int alphabet = 0;
    int64_t count = 0;
    int64_t total = 0;
    int64_t maxLen = 0;            // longest sequence (range checks)
    int computeUnits = 256;
    std::vector<int64_t> offsets;  // host copy, [count + 1]
    uint8_t* d_residues = nullptr;
    int64_t* d_offsets = nullptr;
    size_t residueCap = 0, offsetsCap = 0;   // allocated bytes (a handle may be refilled: opalSearchDatabase)
    // device blocks of dropped views, kept for the views of the next filling (at most kSpareBlocks)
    std::vector<std::pair<void*, size_t>> spareBlocks;

    struct ViewSlot {
        int64_t start, end;
        int overlap, stride;
        bool building;                 // placeholder: the view is being built outside the lock
        std::shared_ptr<View> view;
    };
    // view lists built while the residues were still on their way (opalSearchDatabase: fillHandle)
    std::shared_ptr<View> prefetched;
    std::mutex viewMutex;
    std::condition_variable viewReady;
    std::list<ViewSlot> views;         // most recent first
    size_t viewBudgetBytes = (size_t)64 << 30;   // set from the device's memory at creation

    // pinned staging buffers + streams for view construction, re-used (hipHostMalloc and
    // hipStreamCreate are slow and serialise between threads)
    struct UploadChannel {
        void* pinned = nullptr;
        size_t cap = 0;
        hipStream_t stream = nullptr;
        ~UploadChannel() {
            if (pinned) (void)hipHostFree(pinned);
            if (stream) (void)hipStreamDestroy(stream);
        }
    };
    std::mutex uploadMutex;
    std::vector<std::unique_ptr<UploadChannel>> uploadFree;

    std::mutex wsMutex;
    std::vector<std::unique_ptr<Workspace>> ownedFree;              // internal streams, idle
    std::map<hipStream_t, std::unique_ptr<Workspace>> external;     // caller streams

    std::atomic<int> profiling{0};
    std::mutex timingMutex;
    Workspace* lastTimed = nullptr;

    ~MiopalDb() {
        (void)hipSetDevice(device);
        views.clear();
        uploadFree.clear();
        ownedFree.clear();
        external.clear();
        for (auto& b : spareBlocks) (void)hipFree(b.first);
        if (d_residues) (void)hipFree(d_residues);
        if (d_offsets) (void)hipFree(d_offsets);
    }
};

namespace {

struct WorkspaceLease {
    MiopalDb* db;
    Workspace* ws = nullptr;
    bool owned = false;
    std::unique_lock<std::mutex> lock;
    explicit WorkspaceLease(MiopalDb* d) : db(d) {}
    int acquireInternal() {
        std::unique_ptr<Workspace> w;
        {
            std::lock_guard<std::mutex> g(db->wsMutex);
            if (!db->ownedFree.empty()) {
                w = std::move(db->ownedFree.back());
                db->ownedFree.pop_back();
            }
        }
        if (!w) {
            w.reset(new Workspace());
            HIP_TRY(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
            w->ownsStream = true;
        }
        ws = w.release();
        owned = true;
        return 0;
    }
    int acquireExternal(hipStream_t s) {
        {
            std::lock_guard<std::mutex> g(db->wsMutex);
            auto& slot = db->external[s];
            if (!slot) {
                slot.reset(new Workspace());
                slot->stream = s;
            }
            ws = slot.get();
        }
        lock = std::unique_lock<std::mutex>(ws->busy);
        return 0;
    }
    ~WorkspaceLease() {
        // downloads still pending here belong to a search that returned an error
        if (ws) ws->abandonDownloads();
        if (owned && ws) {
            // Idle workspaces keep their buffers for the next search on whatever thread comes
            // first, but not without bound: many threads that each ran one `full` search would
            // otherwise park tens of GB apiece (the reference's thread pools default to one
            // thread per CPU core). Beyond 64 GB of parked buffers this one is released.
            std::unique_ptr<Workspace> mine(ws);
            std::lock_guard<std::mutex> g(db->wsMutex);
            size_t parked = mine->bytes();
            for (const auto& w : db->ownedFree) parked += w->bytes();
            if (parked <= kParkedWorkspaceBytes || db->ownedFree.empty()) db->ownedFree.emplace_back(std::move(mine));
        }
    }
};

// Host -> device copies of database construction go through pinned bounce pieces owned by the
// library, so that the runtime never pins (and keeps a mapping of) the caller's or the C library's
// pageable memory: see Workspace::finishDownloads. The pieces are filled by a few host threads -
// each with its own two pieces and its own stream, taking piece numbers from a counter - while
// earlier pieces are on their way: gathering a million sequences from a million pointers
// (the hand-off of opalSearchDatabase, src/pyopal/opal.pxd:38-52), checking every residue and
// crossing PCIe overlap. Pieces are kept for the next database of the process (pinning memory
// costs more than copying through it), up to kStagingKept of them.
constexpr size_t kStagingPiece = (size_t)8 << 20;
constexpr size_t kStagingKept = 8;   // kits: two pieces and two events each
constexpr int kUploadStreams = 2;    // copies of all filling threads share these (see streamedUpload)

struct UploadKit {
    int device = -1;
    void* piece[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    ~UploadKit() {
        for (hipEvent_t e : done)
            if (e) (void)hipEventDestroy(e);
        for (void* p : piece)
            if (p) (void)hipHostFree(p);
    }
    bool second() {
        if (piece[1]) return true;
        if (hipHostMalloc(&piece[1], kStagingPiece, hipHostMallocPortable) != hipSuccess) (void)hipGetLastError();
        return piece[1] != nullptr;
    }
};

struct StagingPool {
    std::mutex m;
    std::vector<std::unique_ptr<UploadKit>> free;
    // (the calling thread has made `device` current)
    std::unique_ptr<UploadKit> take(int device, int* why) {
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t k = free.size(); k-- > 0;)
                if (free[k]->device == device) {
                    std::unique_ptr<UploadKit> kit = std::move(free[k]);
                    free.erase(free.begin() + (long)k);
                    return kit;
                }
        }
        std::unique_ptr<UploadKit> kit(new UploadKit());
        kit->device = device;
        if (hipHostMalloc(&kit->piece[0], kStagingPiece, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            *why = 2;
            return nullptr;
        }
        if (hipEventCreateWithFlags(&kit->done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&kit->done[1], hipEventDisableTiming) != hipSuccess) {
            *why = 1;
            return nullptr;
        }
        return kit;
    }
    void give(std::unique_ptr<UploadKit> kit) {
        if (!kit) return;
        std::lock_guard<std::mutex> g(m);
        if (free.size() < kStagingKept) free.emplace_back(std::move(kit));
    }
    // the copy streams of one upload (idle ones of the device, or new ones)
    struct StreamSet {
        int device = -1;
        hipStream_t s[kUploadStreams] = {};
        ~StreamSet() {
            for (hipStream_t x : s)
                if (x) (void)hipStreamDestroy(x);
        }
    };
    std::vector<std::unique_ptr<StreamSet>> freeStreams;
    std::unique_ptr<StreamSet> takeStreams(int device) {
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t k = freeStreams.size(); k-- > 0;)
                if (freeStreams[k]->device == device) {
                    std::unique_ptr<StreamSet> set = std::move(freeStreams[k]);
                    freeStreams.erase(freeStreams.begin() + (long)k);
                    return set;
                }
        }
        std::unique_ptr<StreamSet> set(new StreamSet());
        set->device = device;
        for (hipStream_t& x : set->s)
            if (createUploadStream(&x) != hipSuccess) return nullptr;
        return set;
    }
    void giveStreams(std::unique_ptr<StreamSet> set) {
        if (!set) return;
        std::lock_guard<std::mutex> g(m);
        if (freeStreams.size() < 4) freeStreams.emplace_back(std::move(set));
    }
};
StagingPool& stagingPool() {
    static StagingPool* pool = new StagingPool();   // never destroyed: the runtime may be gone at exit
    return *pool;
}

// dst[0..n) = src[0..n), and the largest byte seen folded into `top` (16 lanes) / `topTail`
typedef unsigned char Bytes16 __attribute__((vector_size(16)));
inline void copyWithMax(unsigned char* dst, const unsigned char* src, size_t n, Bytes16& top, unsigned& topTail) {
    if (n < 16) {
        for (size_t i = 0; i < n; ++i) {
            dst[i] = src[i];
            topTail = std::max<unsigned>(topTail, src[i]);
        }
        return;
    }
    Bytes16 t = top;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        Bytes16 a, b, c, d;
        memcpy(&a, src + i, 16);
        memcpy(&b, src + i + 16, 16);
        memcpy(&c, src + i + 32, 16);
        memcpy(&d, src + i + 48, 16);
        memcpy(dst + i, &a, 16);
        memcpy(dst + i + 16, &b, 16);
        memcpy(dst + i + 32, &c, 16);
        memcpy(dst + i + 48, &d, 16);
        t = __builtin_elementwise_max(t, __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d)));
    }
    for (; i + 16 <= n; i += 16) {
        Bytes16 a;
        memcpy(&a, src + i, 16);
        memcpy(dst + i, &a, 16);
        t = __builtin_elementwise_max(t, a);
    }
    if (i < n) {   // the last, overlapping 16 bytes
        Bytes16 a;
        memcpy(&a, src + n - 16, 16);
        memcpy(dst + n - 16, &a, 16);
        t = __builtin_elementwise_max(t, a);
    }
    top = t;
}
inline unsigned largestByte(const Bytes16& top, unsigned topTail) {
    unsigned m = topTail;
    for (int i = 0; i < 16; ++i) m = std::max<unsigned>(m, top[i]);
    return m;
}

int uploadThreads(size_t bytes) {
    // (one stream each: beyond the runtime's four hardware queues a stream's first copy waits for the others)
    int t = (int)std::min<size_t>(4, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* env = getenv("MIOPAL_UPLOAD_THREADS")) t = std::max(1, std::min(64, atoi(env)));
    const size_t pieces = (bytes + kStagingPiece - 1) / kStagingPiece;
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)t, pieces));
}

// fill(dst, offset, n): writes bytes [offset, offset + n) of the source into dst, returns 0 or a
// positive code of its own (handed back through fillCode); called from several threads, on
// disjoint ranges.
template <class Fill>
int streamedUpload(int device, void* deviceDst, size_t bytes, const Fill& fill, int* fillCode = nullptr) {
    if (fillCode) *fillCode = 0;
    if (bytes == 0) return 0;
    const size_t pieces = (bytes + kStagingPiece - 1) / kStagingPiece;
    const int nThreads = uploadThreads(bytes);
    const bool timed = getenv("MIOPAL_PHASE_TIMING") != nullptr && bytes > (64u << 20);
    // All copies go down a couple of streams shared by the filling threads: the runtime multiplexes
    // streams onto four hardware queues, and with a stream per thread (beside those of the searches)
    // the first copy of an unlucky stream was seen to wait 14 ms for the others to finish.
    if (hipSetDevice(device) != hipSuccess) return fail(MIOPAL_ERR_HIP, "hipSetDevice(%d) failed", device);
    std::unique_ptr<StagingPool::StreamSet> streams = stagingPool().takeStreams(device);
    if (!streams) return fail(MIOPAL_ERR_HIP, "cannot create the upload streams");
    int nStreams = kUploadStreams;
    if (const char* env = getenv("MIOPAL_UPLOAD_STREAMS")) nStreams = std::max(1, std::min(kUploadStreams, atoi(env)));
    std::atomic<int> workers{0};
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};     // 1: HIP, 2: out of pinned memory, otherwise the fill's code << 2
    auto work = [&]() {
        const double tStart = timed ? PhaseTimer::now() : 0;
        double tWait = 0, tFill = 0, tCall = 0;
        if (hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
        int why = 0;
        std::unique_ptr<UploadKit> kit = stagingPool().take(device, &why);
        if (!kit) { failed.store(why); return; }
        const double tSetup = timed ? PhaseTimer::now() : 0;
        hipStream_t stream = streams->s[workers.fetch_add(1) % nStreams];
        int inFlight = 0;    // bit b: piece b has a copy under way
        for (int k = 0; failed.load(std::memory_order_relaxed) == 0; ++k) {
            const size_t p = next.fetch_add(1);
            if (p >= pieces) break;
            const int b = k & 1;
            if (b == 1 && !kit->second()) { failed.store(2); break; }
            const double ta = timed ? PhaseTimer::now() : 0;
            if (k >= 2 && hipEventSynchronize(kit->done[b]) != hipSuccess) { failed.store(1); break; }
            const double tb = timed ? PhaseTimer::now() : 0;
            const size_t off = p * kStagingPiece, n = std::min(kStagingPiece, bytes - off);
            if (const int rc = fill((unsigned char*)kit->piece[b], off, n)) { failed.store(rc << 2); break; }
            const double tc = timed ? PhaseTimer::now() : 0;
            if (hipMemcpyAsync((char*)deviceDst + off, kit->piece[b], n, hipMemcpyHostToDevice, stream) != hipSuccess ||
                hipEventRecord(kit->done[b], stream) != hipSuccess) {
                failed.store(1);
                break;
            }
            inFlight |= 1 << b;
            if (timed) {
                const double td = PhaseTimer::now();
                tWait += tb - ta; tFill += tc - tb; tCall += td - tc;
            }
        }
        const double te = timed ? PhaseTimer::now() : 0;
        // (the pieces go back to the pool only when their copies have left them)
        for (int b = 0; b < 2; ++b)
            if ((inFlight >> b & 1) && hipEventSynchronize(kit->done[b]) != hipSuccess) {
                failed.store(1);
                (void)hipStreamSynchronize(stream);
            }
        if (timed)
            fprintf(stderr, "[miopal]   upload thread: setup %.2f wait %.2f fill %.2f enqueue %.2f drain %.2f ms\n",
                    (tSetup - tStart) * 1e3, tWait * 1e3, tFill * 1e3, tCall * 1e3, (PhaseTimer::now() - te) * 1e3);
        stagingPool().give(std::move(kit));
    };
    if (nThreads == 1) {
        work();
    } else {
        // (a thread that cannot be started is no error: the pieces are handed out by a counter, whoever
        // runs takes them - in the last resort this thread alone)
        std::vector<std::thread> pool;
        pool.reserve((size_t)nThreads);
        for (int t = 1; t < nThreads; ++t) {
            try {
                pool.emplace_back(work);
            } catch (const std::exception&) {
                break;
            }
        }
        work();
        for (auto& t : pool) t.join();
    }
    for (int k = 0; k < nStreams; ++k)
        if (hipStreamSynchronize(streams->s[k]) != hipSuccess) failed.store(1);
    if (failed.load() != 1) stagingPool().giveStreams(std::move(streams));
    const int f = failed.load();
    if (f == 1) return fail(MIOPAL_ERR_HIP, "host to device copy failed: %s", hipGetErrorString(hipGetLastError()));
    if (f == 2) return fail(MIOPAL_ERR_HIP, "cannot allocate the upload bounce buffer");
    if (fillCode) *fillCode = f >> 2;   // (worker threads cannot leave a message: the caller words it)
    return 0;
}

int uploadOnce(int device, void* deviceDst, const void* src, size_t bytes) {
    return streamedUpload(device, deviceDst, bytes, [src](unsigned char* dst, size_t off, size_t n) {
        memcpy(dst, (const char*)src + off, n);
        return 0;
    });
}

// A staging channel of the handle for the duration of one view construction.
struct UploadLease {
    MiopalDb* db;
    std::unique_ptr<MiopalDb::UploadChannel> ch;
    explicit UploadLease(MiopalDb* d) : db(d) {}
    int acquire(size_t bytes) {
        {
            std::lock_guard<std::mutex> g(db->uploadMutex);
            if (!db->uploadFree.empty()) {
                ch = std::move(db->uploadFree.back());
                db->uploadFree.pop_back();
            }
        }
        if (!ch) {
            ch.reset(new MiopalDb::UploadChannel());
            HIP_TRY(createUploadStream(&ch->stream));
        }
        if (ch->cap < bytes) {
            if (ch->pinned) HIP_TRY(hipHostFree(ch->pinned));
            ch->pinned = nullptr;
            ch->cap = 0;
            const size_t want = bytes + bytes / 4 + (1u << 16);
            HIP_TRY(hipHostMalloc(&ch->pinned, want, hipHostMallocDefault));
            ch->cap = want;
        }
        return 0;
    }
    ~UploadLease() {
        if (!ch) return;
        std::lock_guard<std::mutex> g(db->uploadMutex);
        if (db->uploadFree.size() < 32) db->uploadFree.emplace_back(std::move(ch));
    }
};

int dbLen(const MiopalDb* db, int64_t id) { return (int)(db->offsets[id + 1] - db->offsets[id]); }

// ---- view construction -------------------------------------------------------
// Window stride of a segmented view: with overlap O (>= the longest span a local alignment of the
// query can have in the target) every alignment lies inside one window [k S, k S + S + O).
int segmentStride(int overlap) { return std::max(256, (overlap * 3 / 5 + 63) / 64 * 64); }

// Host loops over a million targets (view lists, result structs) are cut into slices worked on by
// a few threads; MIOPAL_HOST_THREADS overrides the count (default: up to 4).
int hostThreads(size_t items, size_t perThread, size_t atMost = 4) {
    int t = (int)std::min<size_t>(atMost, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* env = getenv("MIOPAL_HOST_THREADS")) t = std::max(1, std::min(64, atoi(env)));
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)t, items / std::max<size_t>(perThread, 1)));
}
template <class Body>
void parallelSlices(int nSlices, const Body& body) {
    if (nSlices <= 1) {
        body(0);
        return;
    }
    // (what a slice throws - std::bad_alloc - is carried back to the caller once every thread has ended)
    std::vector<std::exception_ptr> thrown((size_t)nSlices);
    auto run = [&body, &thrown](int t) {
        try {
            body(t);
        } catch (...) {
            thrown[(size_t)t] = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)nSlices - 1);
    int started = 1;   // slices [1, started) have a thread of their own
    for (; started < nSlices; ++started) {
        try {
            pool.emplace_back(run, started);
        } catch (const std::exception&) {
            break;
        }
    }
    run(0);
    for (int t = started; t < nSlices; ++t) run(t);   // (no thread to be had for these)
    for (auto& th : pool) th.join();
    for (const auto& e : thrown)
        if (e) std::rethrow_exception(e);
}

// A device block for a view: one left behind by the views of the handle's previous filling when it
// fits without wasting more than half of itself, a fresh allocation otherwise.
constexpr size_t kSpareBlocks = 4;
int viewBlock(MiopalDb* db, size_t bytes, void** out, size_t* cap) {
    {
        std::lock_guard<std::mutex> g(db->viewMutex);
        auto best = db->spareBlocks.end();
        for (auto it = db->spareBlocks.begin(); it != db->spareBlocks.end(); ++it)
            if (it->second >= bytes && it->second / 2 <= bytes + 4096 && (best == db->spareBlocks.end() || it->second < best->second))
                best = it;
        if (best != db->spareBlocks.end()) {
            *out = best->first;
            *cap = best->second;
            db->spareBlocks.erase(best);
            return 0;
        }
    }
    HIP_TRY(hipMalloc(out, bytes));
    *cap = bytes;
    return 0;
}

int finishView(MiopalDb* db, View* v);

// packNow = false: everything but the pack kernel (which reads the residues on the device): the lists,
// the device blocks and the upload of the small arrays - what can be done while the residues are still
// crossing PCIe; finishView() packs.
int buildView(MiopalDb* db, int64_t start, int64_t end, int overlap, std::shared_ptr<View>* out, bool packNow = true,
              int strideWanted = -1) {
    if (packNow) {
        // (lists of this very slice built ahead, beside the upload of the residues: only the packing is left)
        std::shared_ptr<View> ready;
        {
            std::lock_guard<std::mutex> g(db->viewMutex);
            if (db->prefetched && db->prefetched->start == start && db->prefetched->end == end &&
                db->prefetched->overlap == overlap)
                ready = std::move(db->prefetched);
            db->prefetched.reset();
        }
        if (ready) {
            RC_TRY(finishView(db, ready.get()));
            *out = ready;
            return 0;
        }
    }
    PhaseTimer pt;
    auto v = std::make_shared<View>();
    v->start = start;
    v->end = end;
    v->overlap = overlap;
    // The view's entries (targets, or windows of long targets), longest first: the heaviest
    // wavefronts are dispatched first. A stable counting sort by length (windows of one target stay
    // in order), in slices of the target range worked on side by side: count, place, scatter. For
    // a million targets the lists used to cost more than the search itself.
    const int64_t nT = end - start;
    const int stride = overlap > 0 ? (strideWanted > 0 ? strideWanted : segmentStride(overlap)) : 0, window = stride + overlap;
    v->stride = stride;
    const int nSlices = hostThreads((size_t)nT, 65536);
    auto sliceLo = [&](int t) { return start + nT * t / nSlices; };
    auto windowsOf = [&](int L) {   // windows until the tail is inside the overlap of the previous one
        if (L <= window) return 1;
        return 1 + (L - overlap - 1) / stride;
    };
    std::vector<int64_t> sliceEntries((size_t)nSlices + 1, 0);
    std::vector<int> sliceLongest((size_t)nSlices, 0);
    std::vector<std::vector<int32_t>> sliceLong((size_t)nSlices);
    parallelSlices(nSlices, [&](int t) {
        int64_t entries = 0;
        int longest = 0;
        for (int64_t k = sliceLo(t); k < sliceLo(t + 1); ++k) {
            const int L = dbLen(db, k);
            if (overlap > 0) {
                entries += windowsOf(L);
                longest = std::max(longest, std::min(L, window));
            } else if (L > kLongTarget) {
                sliceLong[(size_t)t].push_back((int32_t)k);
            } else {
                ++entries;
                longest = std::max(longest, L);
            }
        }
        sliceEntries[(size_t)t + 1] = entries;
        sliceLongest[(size_t)t] = longest;
    });
    pt.mark("view: count");
    int longest = 0;
    for (int t = 0; t < nSlices; ++t) {
        sliceEntries[(size_t)t + 1] += sliceEntries[(size_t)t];
        longest = std::max(longest, sliceLongest[(size_t)t]);
        v->longIds.insert(v->longIds.end(), sliceLong[(size_t)t].begin(), sliceLong[(size_t)t].end());
    }
    const size_t nEntries = (size_t)sliceEntries[(size_t)nSlices];
    if (nEntries >= (size_t)INT32_MAX) return fail(MIOPAL_ERR_BAD_ARGUMENT, "too many windows in one view");
    const size_t nBins = (size_t)longest + 1;
    // place[t][b]: first sorted position of slice t's entries of length longest - b
    std::vector<int64_t> place((size_t)nSlices * nBins, 0);
    parallelSlices(nSlices, [&](int t) {
        int64_t* mine = place.data() + (size_t)t * nBins;
        for (int64_t k = sliceLo(t); k < sliceLo(t + 1); ++k) {
            const int L = dbLen(db, k);
            if (overlap > 0) {
                if (L <= window) {
                    ++mine[longest - L];
                } else {
                    const int w = windowsOf(L);
                    mine[longest - window] += w - 1;
                    ++mine[longest - std::min(window, L - (w - 1) * stride)];
                }
            } else if (L <= kLongTarget) {
                ++mine[longest - L];
            }
        }
    });
    pt.mark("view: histogram");
    {
        int64_t running = 0;
        for (size_t bin = 0; bin < nBins; ++bin)
            for (int t = 0; t < nSlices; ++t) {
                int64_t& c = place[(size_t)t * nBins + bin];
                const int64_t n = c;
                c = running;
                running += n;
            }
    }
    pt.mark("view: places");
    const int nGroupsAll = (int)((nEntries + kGroupTargets - 1) / kGroupTargets);
    std::vector<int32_t> ids(nEntries), segStart(overlap > 0 ? nEntries : 0);
    std::vector<int32_t> vlen((size_t)nGroupsAll * kGroupTargets);   // padded to whole groups with zeros
    parallelSlices(nSlices, [&](int t) {
        int64_t* mine = place.data() + (size_t)t * nBins;
        for (int64_t k = sliceLo(t); k < sliceLo(t + 1); ++k) {
            const int L = dbLen(db, k);
            if (overlap > 0) {
                const int w = windowsOf(L);
                for (int x = 0; x < w; ++x) {
                    const int s0 = x * stride, len = w == 1 ? L : std::min(window, L - s0);
                    const size_t at = (size_t)mine[longest - len]++;
                    ids[at] = (int32_t)k;
                    segStart[at] = s0;
                    vlen[at] = len;
                }
            } else if (L <= kLongTarget) {
                const size_t at = (size_t)mine[longest - L]++;
                ids[at] = (int32_t)k;
                vlen[at] = L;
            }
        }
    });
    pt.mark("view: scatter");
    v->nPacked = (int)ids.size();
    v->nGroups = (v->nPacked + kGroupTargets - 1) / kGroupTargets;
    std::vector<int64_t> groupOff(v->nGroups + 1, 0), chunkPrefix(v->nGroups + 1, 0), boundaryOff(v->nGroups + 1, 0);
    std::vector<int> groupChunks(std::max(v->nGroups, 1), 0);
    for (int g = 0; g < v->nGroups; ++g) {
        const int maxLen = vlen[(size_t)g * kGroupTargets];  // sorted: first is longest
        v->maxPackedLen = std::max(v->maxPackedLen, maxLen);
        const int chunks = std::max(1, (maxLen + 3) / 4);
        groupChunks[g] = chunks;
        chunkPrefix[g + 1] = chunkPrefix[g] + chunks;
        groupOff[g + 1] = groupOff[g] + (int64_t)chunks * kLanes;
        boundaryOff[g + 1] = boundaryOff[g] + (int64_t)chunks * 4 * kLanes;
    }
    v->totalChunks = chunkPrefix[v->nGroups];
    v->groupChunksHost.assign(groupChunks.begin(), groupChunks.begin() + v->nGroups);
    if (v->nGroups > 0) {
        // Two device allocations and one staged upload per view: the small per-target and per-group
        // arrays share one blob (hipMalloc, hipHostMalloc and stream creation serialise in the
        // runtime, and thread-chunked callers build their slices' views side by side).
        const size_t packBytes = (size_t)groupOff[v->nGroups] * sizeof(uint2);
        struct Part { const void* src; size_t bytes, at; };
        Part parts[] = {
            {ids.data(), ids.size() * sizeof(int32_t), 0},
            {segStart.data(), overlap > 0 ? segStart.size() * sizeof(int32_t) : 0, 0},
            {vlen.data(), vlen.size() * sizeof(int32_t), 0},
            {groupOff.data(), groupOff.size() * sizeof(int64_t), 0},
            {groupChunks.data(), groupChunks.size() * sizeof(int), 0},
            {boundaryOff.data(), boundaryOff.size() * sizeof(int64_t), 0},
            {chunkPrefix.data(), chunkPrefix.size() * sizeof(int64_t), 0},
        };
        size_t metaBytes = 0;
        for (Part& p : parts) {
            p.at = metaBytes;
            metaBytes += (p.bytes + 255) & ~(size_t)255;
        }
        pt.mark("view: host lists");
        RC_TRY(viewBlock(db, std::max<size_t>(metaBytes, 256), &v->d_meta, &v->metaCap));
        RC_TRY(viewBlock(db, packBytes, (void**)&v->d_pack, &v->packCap));
        pt.mark("view: device allocation");
        char* meta = (char*)v->d_meta;
        v->d_ids = (int32_t*)(meta + parts[0].at);
        v->d_segStart = overlap > 0 ? (int32_t*)(meta + parts[1].at) : nullptr;
        v->d_lens = (int32_t*)(meta + parts[2].at);
        v->d_groupOff = (int64_t*)(meta + parts[3].at);
        v->d_groupChunks = (int*)(meta + parts[4].at);
        v->d_boundaryOff = (int64_t*)(meta + parts[5].at);
        UploadLease up(db);
        RC_TRY(up.acquire(metaBytes));
        for (const Part& p : parts)
            if (p.bytes) memcpy((char*)up.ch->pinned + p.at, p.src, p.bytes);
        HIP_TRY(hipMemcpyAsync(meta, up.ch->pinned, metaBytes, hipMemcpyHostToDevice, up.ch->stream));
        PackArgs pa{};
        pa.residues = db->d_residues;
        pa.offsets = db->d_offsets;
        pa.ids = v->d_ids;
        pa.segStart = v->d_segStart;
        pa.lens = v->d_lens;
        pa.nTargets = v->nPacked;
        pa.groupOff = v->d_groupOff;
        pa.groupChunks = v->d_groupChunks;
        pa.chunkPrefix = (const int64_t*)(meta + parts[6].at);
        pa.nGroups = v->nGroups;
        pa.padSymbol = db->alphabet;
        pa.pack = v->d_pack;
        v->deviceBytes = packBytes + metaBytes;
        if (!packNow) {
            HIP_TRY(hipStreamSynchronize(up.ch->stream));   // (the bounce buffer goes back to the pool)
            v->pendingPack = pa;
            v->packPending = true;
            pt.mark("view: upload (pack later)");
        } else {
            // (the channel's own stream, and a wait for that stream only: other threads' searches go on)
            HIP_TRY(launchPack(pa, v->totalChunks, up.ch->stream));
            HIP_TRY(hipStreamSynchronize(up.ch->stream));
            pt.mark("view: upload + pack");
        }
    }
    v->ids = std::move(ids);
    *out = v;
    return 0;
}

int finishView(MiopalDb* db, View* v) {
    if (!v->packPending) return 0;
    PhaseTimer pt;
    UploadLease up(db);
    RC_TRY(up.acquire(256));
    HIP_TRY(launchPack(v->pendingPack, v->totalChunks, up.ch->stream));
    HIP_TRY(hipStreamSynchronize(up.ch->stream));
    v->packPending = false;
    pt.mark("view: pack (lists built beside the upload)");
    return 0;
}

// Cache of packed views, most recent first. A view is about as large as the slice it packs, so
// the cache is bounded by bytes (a share of the device's memory) as well as by count; thread-chunked
// callers (src/pyopal/_align.py:150-170) re-use their slice on every query, queries of different
// lengths add segmented views of the same slice. A view is built OUTSIDE the lock - other threads
// keep searching their own views meanwhile - behind a placeholder that threads wanting the same
// view wait on. Views in use are kept alive by their shared_ptr whatever the cache drops.
void evictViews(MiopalDb* db, size_t budget, size_t keepCount) {
    size_t total = 0, n = 0;
    for (const auto& s : db->views) {
        total += s.view ? s.view->deviceBytes : 0;
        ++n;
    }
    for (auto it = db->views.end(); it != db->views.begin() && (total > budget || n > keepCount);) {
        --it;
        if (it->building || it == db->views.begin()) continue;   // never the newest, never one being built
        total -= it->view ? it->view->deviceBytes : 0;
        --n;
        it = db->views.erase(it);
    }
}

// `build` fills the view (buildView; the self test of the guard injects its own). Whatever it does -
// return an error, or throw (its vectors have a million entries and parallelSlices rethrows what its
// worker threads threw) - the placeholder is gone and its waiters are woken when this returns: a
// placeholder left behind would block every later search of the same slice forever.
template <class Build>
int getViewWith(MiopalDb* db, int64_t start, int64_t end, int overlap, std::shared_ptr<View>* out, const Build& build,
                int stride = 0) {
    std::unique_lock<std::mutex> lk(db->viewMutex);
    for (;;) {
        auto it = db->views.begin();
        for (; it != db->views.end(); ++it)
            if (it->start == start && it->end == end && it->overlap == overlap && it->stride == stride) break;
        if (it == db->views.end()) break;
        if (!it->building) {
            *out = it->view;
            db->views.splice(db->views.begin(), db->views, it);
            return 0;
        }
        db->viewReady.wait(lk);   // somebody is building this very view
    }
    db->views.push_front(MiopalDb::ViewSlot{start, end, overlap, stride, true, nullptr});
    lk.unlock();
    std::shared_ptr<View> v;
    auto guardedBuild = [&]() -> int {
        try {
            return build(&v);
        } catch (const std::bad_alloc&) {
            return fail(MIOPAL_ERR_INTERNAL, "out of host memory while building a packed view");
        } catch (const std::exception& e) {
            return fail(MIOPAL_ERR_INTERNAL, "building a packed view failed: %s", e.what());
        } catch (...) {
            return fail(MIOPAL_ERR_INTERNAL, "building a packed view failed");
        }
    };
    int rc = guardedBuild();
    if (rc != 0) {
        // most likely out of device memory: drop every idle view and try once more
        v.reset();
        lk.lock();
        evictViews(db, 0, 1);
        lk.unlock();
        (void)hipGetLastError();
        rc = guardedBuild();
    }
    lk.lock();
    auto mine = db->views.begin();
    for (; mine != db->views.end(); ++mine)
        if (mine->building && mine->start == start && mine->end == end && mine->overlap == overlap && mine->stride == stride) break;
    if (rc != 0) {
        if (mine != db->views.end()) db->views.erase(mine);
        db->viewReady.notify_all();
        return rc;
    }
    if (mine == db->views.end()) {   // (cannot happen: placeholders are only removed by their builder)
        db->views.push_front(MiopalDb::ViewSlot{start, end, overlap, stride, false, v});
    } else {
        mine->view = v;
        mine->building = false;
        db->views.splice(db->views.begin(), db->views, mine);
    }
    evictViews(db, db->viewBudgetBytes, kMaxCachedViews);
    db->viewReady.notify_all();
    *out = v;
    return 0;
}

int getView(MiopalDb* db, int64_t start, int64_t end, int overlap, std::shared_ptr<View>* out, int stride = 0) {
    return getViewWith(db, start, end, overlap, out,
                       [&](std::shared_ptr<View>* v) { return buildView(db, start, end, overlap, v, true, stride); }, stride);
}

// ---- one search ----------------------------------------------------------------
struct Search {
    MiopalDb* db;
    Workspace* ws;
    hipStream_t stream;
    const unsigned char* query;
    int Q, open, ext, A, searchType, mode;
    const int* matrix;
    int64_t start, end;
    int64_t n;
    int maxScore = 0, minScore = 0;
    int64_t balancedChunks = 0;
    bool globalPairRefused = false;   // the pair-table launch for NW / HW / OV failed on this device
    bool pairStripsRefused = false;   // the same for the multi-strip Smith-Waterman kernel
    bool globalStripsRefused = false; // ... and for the multi-strip NW / HW / OV kernel
    int* d_stripError = nullptr;      // units of intraseq_strips_kernel that gave up waiting (never seen)
    int stripErrorHost = 0;
    bool stripsEndsDeclined = false;  // ... with end locations: a probe of the longest groups left its range of 384
    // Smith-Waterman end locations of several strips whose scores are beyond the row keys' range: two sweeps of
    // the strips kernel - scores, then the first cell that holds each target's score (round 3)
    bool twoPassEnds = false;
    // miopalSearch: a pinned, device-visible buffer the scores may be written to directly (the one-strip
    // Smith-Waterman fast path scatters into it from the kernel); wroteHost says that it was used
    int32_t* hostScoreOut = nullptr;
    uint64_t hostScoreGeneration = 0;   // of the staging buffer when hostScoreOut was reserved in it
    bool hostScoreIsCallers = false;    // hostScoreOut is the caller's own pinned array
    bool wroteHost = false;
    bool besidePersistent = false;    // the side jobs of this pass run beside a strips kernel (persistent, one workgroup per CU)
    // a score pass that starts over (refused launch, declined probe) has already put its side jobs on
    // the side stream: they are not enqueued twice, and the join still waits for them
    // test hook (miopalTestInjectFault): kind 1 = a unit of the pair-table strips kernels, 2 = a unit of
    // intraseq_strips_kernel publishes nothing; the units below it wait at most faultSpinCap polls
    int faultKind = 0, faultUnit = 0, faultSpinCap = 0;
    std::vector<int32_t> sideDone;    // result slots computed on the side stream in this search, sorted
    bool sideForked = false;

    uint8_t* d_query = nullptr;
    int32_t* d_matrix = nullptr;

    int rulesFor(int m, DpRules* r) const {
        switch (m) {
            case OPAL_MODE_NW: *r = {1, 1, 0, kLastCell}; return 0;
            case OPAL_MODE_HW: *r = {0, 1, 0, kLastRow}; return 0;
            case OPAL_MODE_OV: *r = {0, 0, 0, kLastRowCol}; return 0;
            case OPAL_MODE_SW: *r = {0, 0, 1, kAllCells}; return 0;
        }
        return fail(OPAL_ERR_INVALID_MODE, "invalid alignment mode %d", m);
    }

    int prepare() {
        if (g_fault[0] != 0) {   // one search only
            faultKind = g_fault[0];
            faultUnit = g_fault[1];
            faultSpinCap = g_fault[2];
            g_fault[0] = 0;
        }
        maxScore = *std::max_element(matrix, matrix + A * A);
        minScore = *std::min_element(matrix, matrix + A * A);
        return 0;
    }

    // query + matrix are only needed by the intra-sequence / traceback kernels
    int ensurePairInputs() {
        if (d_query) return 0;
        void* p;
        RC_TRY(ws->get(kQuery, (size_t)std::max(Q, 1), &p));
        d_query = (uint8_t*)p;
        RC_TRY(ws->get(kMatrix, (size_t)A * A * sizeof(int32_t), &p));
        d_matrix = (int32_t*)p;
        RC_TRY(ws->stageUpload(d_query, query, (size_t)Q, stream));
        RC_TRY(ws->stageUpload(d_matrix, matrix, (size_t)A * A * sizeof(int32_t), stream));
        return 0;
    }

    // Conservative range check for the 32-bit kernels (the reference returns
    // OPAL_ERR_OVERFLOW when its widest lanes overflow, pyx.in:104-105).
    int checkInt32(int64_t maxLen) const {
        const int64_t mag = std::max<int64_t>(std::llabs((long long)maxScore), std::llabs((long long)minScore));
        const int64_t bound = 2 * (int64_t)std::llabs((long long)open) +
                              ((int64_t)Q + maxLen) * std::llabs((long long)ext) +
                              std::min<int64_t>(Q, maxLen) * mag + mag;
        if (bound >= kInt32Safe)
            return fail(OPAL_ERR_OVERFLOW, "scores may exceed the 32-bit range (bound %lld)", (long long)bound);
        return 0;
    }

    bool interseqUsable() const {
        if (Q <= 0) return false;
        if (open < 0 || ext < 0) return false;
        if (maxScore > 16383 || minScore < -16383) return false;
        return true;
    }

    // Runs the intra-sequence kernel over `jobs`; results land in the given device arrays.
    int runPairs(std::vector<PairJob>& jobs, bool trace, int32_t* d_score, int32_t* d_endI,
                 int32_t* d_endJ, uint8_t* d_dirs, hipStream_t on = nullptr, int slotBase = 0) {
        if (jobs.empty()) return 0;
        if (!on) on = stream;
        RC_TRY(ensurePairInputs());
        int64_t wsElems = 0;
        for (auto& j : jobs) {
            j.wsOff = wsElems;
            if (j.qLen > kLanes) wsElems += j.tLen;
        }
        void *pj, *b0, *b1;
        // the side stream has its own job / boundary buffers (slotBase = kAuxJobs - kJobs)
        RC_TRY(ws->get(kJobs + slotBase, jobs.size() * sizeof(PairJob), &pj));
        RC_TRY(ws->get(kPairB0 + slotBase, (size_t)wsElems * sizeof(int2), &b0));
        RC_TRY(ws->get(kPairB1 + slotBase, (size_t)wsElems * sizeof(int2), &b1));
        RC_TRY(ws->stageUpload(pj, jobs.data(), jobs.size() * sizeof(PairJob), on));
        IntraseqArgs a{};
        a.jobs = (const PairJob*)pj;
        a.nJobs = (int)jobs.size();
        a.residues = db->d_residues;
        a.query = d_query;
        a.matrix = d_matrix;
        a.alphabet = A;
        a.gapOpen = open;
        a.gapExt = ext;
        a.boundary[0] = (int2*)b0;
        a.boundary[1] = (int2*)b1;
        a.dirs = d_dirs;
        a.score = d_score;
        a.endI = d_endI;
        a.endJ = d_endJ;
        a.raisePriority = (on != stream && !getenv("MIOPAL_NO_PRIORITY")) ? 1 : 0;
        // Pairs of several strips, scores / end locations: one wavefront per (pair, strip), the strips of
        // a pair side by side (intraseq.hip) - a pair is then a chain of L + 63 steps, not strips x that.
        // Worth it while the pairs are few against the chip (a chain is what is waited for); thousands of
        // pairs fill the chip either way.
        bool uniform = !trace && !jobs.empty() && jobs[0].qLen > kLanes && !getenv("MIOPAL_NO_PAIR_STRIP_UNITS");
        for (const auto& j : jobs)
            if (j.qLen != jobs[0].qLen) uniform = false;
        const int jobStrips = uniform ? (jobs[0].qLen + kLanes - 1) / kLanes : 1;
        if (uniform && jobStrips >= 2 && (int64_t)jobs.size() * jobStrips <= (1 << 20) &&
            (int64_t)jobs.size() <= 16 * (int64_t)db->computeUnits) {
            void *st, *pt;
            const size_t ints = jobs.size() * (size_t)jobStrips + 1;
            RC_TRY(ws->get(kPairStripState + slotBase, ints * sizeof(int), &st));
            RC_TRY(ws->get(kPairStripPartial + slotBase, (ints - 1) * sizeof(int4), &pt));
            HIP_TRY(hipMemsetAsync(st, 0, ints * sizeof(int), on));
            if (!d_stripError) {
                // The counter only ever moves when a unit gives up, and a search that sees it moved
                // fails and zeroes it again (checkStripError): it is zeroed ONCE, synchronously, when the
                // workspace allocates it - no memset on whichever of the two streams asks first, which
                // launches on the other stream would not be ordered after.
                const bool fresh = ws->cap[kPairStripError] == 0;
                void* pe;
                RC_TRY(ws->get(kPairStripError, sizeof(int), &pe));
                if (fresh) HIP_TRY(hipMemset(pe, 0, sizeof(int)));
                d_stripError = (int*)pe;
            }
            a.nStrips = jobStrips;
            a.stripCounter = (int*)st;
            a.stripProgress = (int*)st + 1;
            a.stripPartial = (int4*)pt;
            a.error = d_stripError;
            a.stripWaitCap = faultSpinCap;
            a.faultUnit1 = faultKind == 2 ? faultUnit + 1 : 0;
            a.fatBlocks = besidePersistent && !getenv("MIOPAL_THIN_SIDE") ? 1 : 0;
            HIP_TRY(launchIntraseqStrips(a, on));
            return 0;
        }
        HIP_TRY(launchIntraseq(a, trace, on));
        return 0;
    }

    // Same with a job list that already sits in HBM. wsStride > 0: the jobs' query pieces may have
    // more than 64 rows, job k owns wsStride strip-boundary columns at wsOff = k * wsStride.
    // headWaves: only the first *headWaves x 64 jobs (hybrid direction pass), addressed by position.
    int runDeviceJobs(const PairJob* d_jobs, int nJobs, int32_t* d_score, int32_t* d_endI, int32_t* d_endJ,
                      bool trace = false, uint8_t* d_dirs = nullptr, int64_t wsStride = 0,
                      const int* headWaves = nullptr, int64_t headDirStride = 0) {
        if (nJobs <= 0) return 0;
        RC_TRY(ensurePairInputs());
        IntraseqArgs a{};
        a.headWaves = headWaves;
        a.headDirStride = headDirStride;
        a.headWsStride = wsStride;
        if (wsStride > 0) {
            void *b0, *b1;
            RC_TRY(ws->get(kPairB0, (size_t)nJobs * wsStride * sizeof(int2), &b0));
            RC_TRY(ws->get(kPairB1, (size_t)nJobs * wsStride * sizeof(int2), &b1));
            a.boundary[0] = (int2*)b0;
            a.boundary[1] = (int2*)b1;
        }
        a.jobs = d_jobs;
        a.nJobs = nJobs;
        a.residues = db->d_residues;
        a.query = d_query;
        a.matrix = d_matrix;
        a.alphabet = A;
        a.gapOpen = open;
        a.gapExt = ext;
        a.score = d_score;
        a.endI = d_endI;
        a.endJ = d_endJ;
        a.dirs = d_dirs;
        HIP_TRY(launchIntraseq(a, trace, stream));
        return 0;
    }

    // After the streams have drained: did a (pair, strip) unit of intraseq_strips_kernel give up?
    // (both entry points that can route pairs there call it; never seen outside the fault-injection test)
    int checkStripError() {
        if (stripErrorHost == 0) return 0;
        const int seen = stripErrorHost;
        stripErrorHost = 0;
        HIP_TRY(hipMemset(d_stripError, 0, sizeof(int)));
        return fail(MIOPAL_ERR_INTERNAL, "%d (pair, strip) units of the wavefront-per-pair kernel gave up waiting for the strip above", seen);
    }

    PairJob forwardJob(int64_t id, int rules) const {
        PairJob j{};
        j.tOff = db->offsets[id];
        j.tLen = dbLen(db, id);
        j.tStep = 1;
        j.qOff = 0;
        j.qLen = Q;
        j.qStep = 1;
        j.rules = rules;
        j.out = (int32_t)(id - start);
        return j;
    }

    // Score pass (all search types). d_score/d_endI/d_endJ are in database order.
    int scorePass(int32_t* d_score, int32_t* d_endI, int32_t* d_endJ) {
        return scorePassImpl(d_score, d_endI, d_endJ, true);
    }

    int scorePassImpl(int32_t* d_score, int32_t* d_endI, int32_t* d_endJ, bool useHalf) {
        DpRules r;
        RC_TRY(rulesFor(mode, &r));
        const int rules = packRules(r);
        std::vector<PairJob> jobs;
        RC_TRY(checkInt32(db->maxLen));

        g_lastRouting[0] = g_lastRouting[1] = g_lastRouting[2] = g_lastRouting[3] = 0;
        if (!interseqUsable() || (Q <= kLanes && n <= kSmallSearch && !getenv("MIOPAL_NO_SMALL_SEARCH"))) {
            g_lastRouting[0] = n;
            jobs.reserve((size_t)n);
            for (int64_t k = start; k < end; ++k) jobs.push_back(forwardJob(k, rules));
            return runPairs(jobs, false, d_score, d_endI, d_endJ, nullptr);
        }

        PhaseTimer spt;
        // Smith-Waterman: long targets can be searched as overlapping windows. A
        // local alignment with a positive score has at most Q aligned pairs and, each gap
        // column costing at least min(open, ext), at most Q * max(S) / min(open, ext) gap
        // columns, so it spans at most that many target columns (`reach`): with windows that
        // overlap by `reach` every alignment lies inside one of them, and the maximum over a
        // target's windows is its score. No window is longer than stride + overlap, so no
        // target has to leave the packed kernel for the 14x dearer wavefront-per-pair kernel.
        int overlap = 0;
        if (mode == OPAL_MODE_SW && std::min(open, ext) > 0 && maxScore > 0 && (int64_t)Q * maxScore < (1 << 23) &&
            Q < 65536 && db->maxLen < (1 << 24) && !getenv("MIOPAL_NO_SEGMENTS")) {
            const int64_t reach = Q + (int64_t)Q * maxScore / std::min(open, ext) + 1;
            // (steps of 128: queries of similar length share one cached view, and the window - the
            // critical path of a group of long targets - stays close to what the query needs)
            const int64_t rounded = (reach + 127) / 128 * 128;
            if (rounded <= 2048 && db->maxLen > segmentStride((int)rounded) + rounded) overlap = (int)rounded;
        }
        // HW (the whole query, free ends in the target) can be cut the same way. Its optimum is at least
        // -(open + (Q - 1) ext) (the query gapped against nothing) and at most Q max(S) minus what its
        // gap columns in the target cost, so it has at most (Q max(S) + open + (Q - 1) ext) / min(open, ext)
        // of them and spans Q + that many columns; a window's penalised left border stands for real
        // alignments (the query's head gapped from the free top border), so no window exceeds the whole
        // target and the one that holds the optimal alignment reaches it. NW spans the whole target by
        // definition; OV's last-column candidates only exist in a target's last window: both stay whole.
        int keyBias = 0;
        if (mode == OPAL_MODE_HW && std::min(open, ext) > 0 && Q < 4096 && db->maxLen < (1 << 24) &&
            !getenv("MIOPAL_NO_SEGMENTS")) {
            const int64_t gaps = ((int64_t)Q * std::max(maxScore, 0) + open + ((int64_t)Q - 1) * ext) / std::min(open, ext);
            const int64_t rounded = (Q + gaps + 1 + 127) / 128 * 128;
            const int64_t lowest = 2 * (int64_t)open + ((int64_t)Q + rounded + 8) * ext + (int64_t)Q * std::max(0, -minScore);
            if (rounded <= 2048 && db->maxLen > segmentStride((int)rounded) + rounded && lowest < (1 << 21)) {
                overlap = (int)rounded;
                keyBias = 1 << 22;   // scores above -2^22 in the key's 24-bit score field
            }
        }
        // Window stride. The overlap is what correctness needs; the stride only trades the length of a window
        // (a group of long targets is a chain of stride + overlap columns) against the work done twice (every
        // window repeats `overlap` columns: at the shortest stride, 0.6 overlap, a long target costs 2.7 times
        // its cells). When long targets are the bulk of the database rather than its tail - a tenth of 2M
        // targets thirty times as long as the rest: 5.3 TCUPS for Smith-Waterman at Q = 53 where NW, which
        // cannot cut them, ran at 9.6 - a window may be as long as 1.5 balanced shares of a wavefront slot.
        int stride = overlap > 0 ? segmentStride(overlap) : 0;
        if (overlap > 0 && !getenv("MIOPAL_SHORT_STRIDE")) {
            const double share = db->count > 0 ? (double)n / (double)db->count : 1.0;
            const double balancedColumns = (double)db->total * share / ((double)kGroupTargets * 12.0 * db->computeUnits);
            const int64_t want = (int64_t)(1.5 * balancedColumns) - overlap;
            if (want > stride) stride = (int)std::min<int64_t>((want + 255) / 256 * 256, 8192);
        }
        std::shared_ptr<View> view;
        RC_TRY(getView(db, start, end, overlap, &view, stride));
        spt.mark("    view lookup");
        // not handled by the packed kernel: can be recomputed beside it
        std::vector<PairJob> sideJobs;
        for (int32_t id : view->longIds) sideJobs.push_back(forwardJob(id, rules));
        // windows of one target are merged with atomicMax: start from 0 (Smith-Waterman scores
        // are never negative); whole-target results of the int32 kernel are plain stores of the
        // final value, in either order the maximum is that value
        const bool keyed = overlap > 0 && searchType != OPAL_SEARCH_SCORE;  // with end locations
        void* keys = nullptr;
        if (keyed) {
            RC_TRY(ws->get(kKeys, (size_t)n * sizeof(unsigned long long), &keys));
            HIP_TRY(hipMemsetAsync(keys, 0, (size_t)n * sizeof(unsigned long long), stream));
        } else if (overlap > 0) {
            // (Smith-Waterman scores start from 0, HW scores from "minus infinity")
            if (keyBias) HIP_TRY(launchFillInt32(d_score, (int)n, INT32_MIN, stream));
            else HIP_TRY(hipMemsetAsync(d_score, 0, (size_t)n * sizeof(int32_t), stream));
        }
        // (a target whose windows are neighbours in the view is queued once)
        auto queueWhole = [&](std::vector<PairJob>& list, int32_t id) {
            if (list.empty() || list.back().out != (int32_t)(id - start)) list.push_back(forwardJob(id, rules));
        };

        // Strips and wavefronts per workgroup for queries of more than 64 rows. A workgroup of W
        // wavefronts pipelines W strips of one group; a round with fewer strips than W leaves
        // wavefronts (and their registers and LDS) idle, so the number of strips is a multiple
        // of W, at the price of shorter strips (each strip pays ~6 rows' worth of per-column
        // overhead). Measured on 500k x 300 (tools history): Q=300 as 5 strips / 4 wavefronts
        // 10.1 ms, as 8 / 4 6.6 ms; Q=150 as 3 / 2 4.7 ms, as 4 / 4 3.3 ms. W = 4 is the cheapest
        // pipeline per cell, W = 8 pays off when there are too few groups to fill the chip
        // (Q=2000 vs 100k x 2000: 782 groups), W = 1 (strips in turn through HBM) only with
        // thousands of groups.
        int nStrips = std::max(1, (Q + kMaxStripRows - 1) / kMaxStripRows);
        int waves = nStrips >= 8 ? 8 : nStrips >= 4 ? 4 : nStrips >= 2 ? 2 : 1;
        if (Q > kMaxStripRows) {
            const double need = 24.0 * db->computeUnits;  // wavefronts that fill the chip
            const double groups = std::max(1, view->nGroups);
            double bestCost = 0;
            for (int w : {4, 8, 2, 1}) {
                const int strips = w * ((Q + kMaxStripRows * w - 1) / (kMaxStripRows * w));
                const int rows = ((Q + strips - 1) / strips + 7) / 8 * 8;
                // (strips are whole multiples of 8 rows; the last one must still hold row Q - 1)
                if ((strips - 1) * rows >= Q) continue;
                const double pipeline = w == 4 ? 1.0 : w == 8 ? 1.12 : w == 2 ? 1.06 : 1.08;
                double cost = (double)strips * (rows + 6) * pipeline;
                const double parallel = groups * w;
                if (parallel < need) cost *= need / parallel;
                if (bestCost == 0 || cost < bestCost) {
                    bestCost = cost;
                    nStrips = strips;
                    waves = w;
                }
            }
        }
        if (const char* o = getenv("MIOPAL_STRIPS")) {  // experiments: "<strips>,<wavefronts>"
            int a = 0, b = 0;
            const int least = std::max(1, (Q + kMaxStripRows - 1) / kMaxStripRows);
            if (sscanf(o, "%d,%d", &a, &b) == 2 && a >= least && (b == 1 || b == 2 || b == 4 || b == 8) &&
                (a - 1) * (((Q + a - 1) / a + 7) / 8 * 8) < Q) {
                nStrips = a;
                waves = b;
            }
        }
        // Smith-Waterman scores of more rows than one pair table holds: the pair-table kernel strip by
        // strip (interseq_pair_strips_kernel: units of (batch of 12 groups, strip), boundary rows
        // through HBM), when the scores and gap costs fit the biased halves' guard band. Strips of at
        // most 52 rows (the kernel's register budget), all of the same even height.
        int stripRows = 0;
        {
            const char* noPair = getenv("MIOPAL_NO_PAIR_TABLE");
            const int64_t up = std::max<int64_t>((int64_t)maxScore + ext, (int64_t)ext - open);
            const int64_t down = std::max<int64_t>(-((int64_t)minScore + ext), (int64_t)open - ext);
            // (with end locations every value is scaled by 2^bits: the row inside a strip of 32 .. 48 rows)
            const bool wantEnds = searchType != OPAL_SEARCH_SCORE;
            // (two sweeps: unscaled values like a score search, strips of at most 40 rows; MIOPAL_TWO_PASS_ENDS=1
            // asks for them whatever the scores - tests)
            if (wantEnds && mode == OPAL_MODE_SW && getenv("MIOPAL_TWO_PASS_ENDS")) twoPassEnds = true;
            const bool rowKeyEnds = wantEnds && !twoPassEnds;
            auto band = [&](int rowsP) {
                const int sbits = rowKeyEnds ? locRowBitsHost(rowsP) : 0;
                return (up << sbits) <= kBiasedMaxStepUp && (down << sbits) <= (rowKeyEnds ? kLocGuardBand : kBiasedMaxMagnitude) &&
                       5 * ((int64_t)ext << sbits) <= kLocMaxShift && minScore > kBiasedPad;
            };
            const int single = std::max(2, (Q + 1) / 2 * 2);
            const bool oneStrip = Q <= kLanes && interseqPairFits(single, A + 1);
            int maxRows = !wantEnds ? kPairStripsMaxRows : twoPassEnds ? kPairStripsMaxRowsKnown : kPairStripsMaxRowsLoc;
            while (maxRows >= 32 && !interseqPairFits(maxRows, A + 1)) maxRows -= 2;
            if (mode == OPAL_MODE_SW && Q < (1 << 20) && useHalf && !oneStrip && maxRows >= 32 &&
                !pairStripsRefused && !(wantEnds && stripsEndsDeclined && !twoPassEnds) && !(noPair && noPair[0] == '1') && !getenv("MIOPAL_NO_BIASED") &&
                !getenv("MIOPAL_NO_PAIR_STRIPS") && !getenv("MIOPAL_STRIPS")) {
                const int ns = std::max(2, (Q + maxRows - 1) / maxRows);
                const int rowsP = ((Q + ns - 1) / ns + 1) / 2 * 2;
                // Worth it when there are units enough to keep every CU on one strip for a while, or strips
                // enough that the general kernel pays many rounds: measured on 2k .. 1M x 300, the strips
                // kernel wins from about 2.5 units per CU on (+9 .. +24 %) and from 20 strips on at any
                // size, and loses below (50k x 300 at Q = 300, 0.8 units per CU: 1.17 against 0.85 ms)
                const int64_t units = (int64_t)((view->nGroups + 11) / 12) * ns;
                // A wavefront sweeps a strip of ~50 rows at about a microsecond per column, whoever shares
                // its CU (a batch is twelve neighbours of the length-sorted view: all long, or all short),
                // where the general kernel pipelines a group's strips over the wavefronts of a workgroup
                // at a third of that. The longest group must therefore be short against the launch, or
                // it IS the launch: log-normal lengths, 500k targets at Q = 150, windows of 3072 columns:
                // 5.3 ms against the general kernel's 2.9 ms.
                int64_t totalChunks = 0;
                for (int c : view->groupChunksHost) totalChunks += c;
                const int64_t balanced = totalChunks * ns / ((int64_t)db->computeUnits * 12);
                // (Round 3, with the wavefronts of a SIMD paced: measured again over 20k .. 2M targets, uniform,
                // log-normal and bimodal lengths - tools/quick_routing_ab.py, profiles/r03_routing_ab.txt. The
                // strips kernel wins from 1.5 units per CU on; its longest group may be as long as 1.2 balanced
                // shares of a wavefront slot - the chain of that group is then about the whole launch - and with
                // 16 strips or more the general kernel's rounds cost more than any chain.)
                const bool hidden = view->nGroups > 0 && 5 * (int64_t)view->groupChunksHost[0] <= 6 * balanced;
                // (with 16 strips or more the strips kernel whatever the longest group: log-normal lengths, 20k
                // targets at Q = 2000: 3.2 against the general kernel's 1.1 TCUPS; the one shape that loses - 20k
                // targets, a tenth of them thirty times as long as the rest - loses 15 %: profiles/r03c_routing_table.txt)
                const bool enough = (ns >= 16 || (2 * units >= 3 * (int64_t)db->computeUnits && hidden)) || getenv("MIOPAL_PAIR_STRIPS");
                if (enough && rowsP >= 32 && rowsP <= maxRows && band(rowsP) && (int64_t)(ns - 1) * rowsP < Q && ns <= 4096) {
                    stripRows = rowsP;
                    nStrips = ns;
                    waves = 1;   // (strips of a group in turn, each in its own unit)
                }
            }
        }
        // NW / HW / OV of more rows than one pair table holds: the same units with the one-strip global
        // kernel's cell (interseq_pair_global_strips_kernel, round 3: 3 integer adds + 3 max per cell pair
        // and no v_perm, against 5 + 1 v_perm and a barrier per chunk on the general kernel's cheapest
        // lanes). The true values around a pattern's zero are bounded by the QUERY (all of it: the strips
        // share one scale), not by the targets' lengths, so no target is redone at 32 bit; the bounds are
        // static:   below zero: 3 open + (Q + 4) ext + |min S|
        //           above: NW Q (max S + ext), HW / OV Q max S + the rebase shift.
        // A 2000-residue query under BLOSUM62 3 / 1 (BASELINE configs[3]) fits; from about 2200 (HW / OV)
        // and 2350 residues (NW) on the general kernel takes over.
        bool globalStrips = false;
        if (mode != OPAL_MODE_SW && useHalf && !globalStripsRefused && !getenv("MIOPAL_NO_BIASED") &&
            !getenv("MIOPAL_NO_GLOBAL_STRIPS") && !getenv("MIOPAL_STRIPS")) {
            const char* noPair = getenv("MIOPAL_NO_PAIR_TABLE");
            const int single = std::max(2, (Q + 1) / 2 * 2);
            const bool oneStrip = Q <= kLanes && interseqPairFits(single, A + 1);
            // (with end locations the row / column bookkeeping costs registers: shorter strips)
            int maxRows = searchType != OPAL_SEARCH_SCORE ? kPairStripsMaxRowsLoc : kPairStripsMaxRows;
            while (maxRows >= 32 && !interseqPairFits(maxRows, A + 1)) maxRows -= 2;
            const int64_t pos = std::max(maxScore, 0);
            const int64_t zeroG = 0x0400 + 3 * (int64_t)open + ((int64_t)Q + 4) * ext + std::max(0, -minScore);
            // (the cells of a strip are on anti-diagonally shifted scales - row r carries r ext more - and H is
            // kept open - ext below its plain form: one strip's rows and an opening more on either side; the
            // kernel's zero is the one-strip kernel's, which has this room below it: 3 open cover 2)
            const int64_t above = (r.topGap ? (int64_t)Q * (pos + ext) : (int64_t)Q * pos + kLocMaxShift) +
                                  ((int64_t)kPairStripsMaxRows + 4) * ext + open;
            const bool inRange = zeroG + above + 5 * (int64_t)ext + pos < 0x7C00 && 5 * (int64_t)ext <= kLocMaxShift &&
                                 minScore > kBiasedPad && (r.topGap ? open >= ext : true);
            if (!oneStrip && maxRows >= 32 && Q > 32 && inRange && !(noPair && noPair[0] == '1')) {
                // strip height: even, at most maxRows; every strip costs about four rows' worth of per-column
                // work on top of its cells (row above in, last row out, answers), and a last query row that
                // is the strip's last row is read without a select (Q = 2000: 40 strips of 50 rows)
                int bestRows = 0, bestNs = 0;
                int64_t bestCost = 0;
                for (int rowsP = maxRows; rowsP >= 32; rowsP -= 2) {
                    const int ns = (Q + rowsP - 1) / rowsP;
                    if (ns < 2 || (int64_t)(ns - 1) * rowsP >= Q) continue;
                    const int64_t cost = (int64_t)ns * (rowsP + 4) + ((int64_t)ns * rowsP - Q > 1 ? rowsP / 8 : 0);
                    if (bestRows == 0 || cost < bestCost) {
                        bestRows = rowsP;
                        bestNs = ns;
                        bestCost = cost;
                    }
                }
                if (bestRows > 0 && bestNs <= 4096) {
                    const int64_t units = (int64_t)((view->nGroups + 11) / 12) * bestNs;
                    int64_t totalChunks = 0;
                    for (int c : view->groupChunksHost) totalChunks += c;
                    const int64_t balanced = totalChunks * bestNs / ((int64_t)db->computeUnits * 12);
                    // (the general kernel's lanes are a third slower for these modes than this kernel's: a longest
                    // group of up to 1.5 balanced shares still pays, and so does one unit per CU)
                    const bool hidden = view->nGroups > 0 && 2 * (int64_t)view->groupChunksHost[0] <= 3 * balanced;
                    // (the same two limits as the Smith-Waterman strips kernel: units enough to keep every
                    // CU on one strip for a while, the longest group short against the launch)
                    const bool enough = (bestNs >= 16 || (units >= (int64_t)db->computeUnits && hidden)) || getenv("MIOPAL_PAIR_STRIPS");
                    if (enough) {
                        stripRows = bestRows;
                        nStrips = bestNs;
                        waves = 1;
                        globalStrips = true;
                    }
                }
            }
        }
        // A group keeps its wavefronts busy for (columns of its longest target) x (rounds of strips).
        // Groups far above the balanced share of a workgroup slot would stretch the kernel to
        // their own length (one lane per target cannot split a target), so the leading
        // (longest) groups are skipped and their targets go to the intra-sequence kernel,
        // which spreads each pair over 64 lanes.
        int firstGroup = 0;
        if (view->nGroups > 0) {
            int64_t total = 0;
            for (int c : view->groupChunksHost) total += c;
            // (the strips kernel: 12 wavefronts per CU, a group's strips side by side in different ones)
            const int64_t slots = stripRows ? std::max<int64_t>(1, (int64_t)db->computeUnits * 12 / nStrips)
                                            : (int64_t)db->computeUnits * std::max(1, 12 / waves);
            // (2.5 x the balanced share; measured again in round 2 on the log-normal database, NW at
            // Q = 150: keeping the longest group in the packed kernel costs 5.6 ms against 4.0)
            int64_t limit = std::max<int64_t>(5 * (total / std::max<int64_t>(slots, 1)) / 2, 128);
            // windows of a segmented view are as short as a group of long targets can get
            if (overlap > 0) limit = std::max<int64_t>(limit, (stride + overlap + 3) / 4);
            while (firstGroup < view->nGroups && view->groupChunksHost[firstGroup] > limit) ++firstGroup;
            // ... unless the groups above the limit ARE the search (round 3: a tenth of 500k targets thirty
            // times as long as the rest - 391 groups of 3000 columns hold three quarters of the cells, all of
            // them beyond 2.5 balanced shares: 50 560 targets on the int32 kernel, 7.7 ms for NW at Q = 53,
            // 0.5 TCUPS). Rough costs of both ways: a wavefront sweeps a 4-column chunk of ~54 rows in about
            // 7 us beside two others on its SIMD and 2.5 times faster alone - which is how the long groups at
            // the head of the length-sorted hand-out end, the short ones long done; the int32 kernel fills
            // about 1e12 cells a second beside the packed launch.
            if (firstGroup > 0 && !getenv("MIOPAL_ALWAYS_SKIP")) {
                const int rowsNow = stripRows ? stripRows : std::min(Q, kMaxStripRows);
                const double rounds = stripRows ? 1.0 : (double)((nStrips + waves - 1) / waves);
                // (the general kernel of several strips: a workgroup pipelines a group's strips with a barrier
                // per chunk step, about 7.5 us per step and round whoever else is on the CU - no faster alone)
                const bool pipelined = !stripRows && nStrips > 1;
                const double tau = pipelined ? 7.5e-6 * rounds : 7e-6 * rowsNow / 54.0;
                const double alone = pipelined ? 1.0 : 2.5;
                const double balancedAll = (double)total / (double)std::max<int64_t>(slots, 1) * (stripRows ? nStrips : 1);
                int64_t skippedChunks = 0;
                for (int g = 0; g < firstGroup; ++g) skippedChunks += view->groupChunksHost[g];
                const double balancedRest = (double)(total - skippedChunks) / (double)std::max<int64_t>(slots, 1) * (stripRows ? nStrips : 1);
                const double keep = std::max(balancedAll * tau, view->groupChunksHost[0] * tau / alone);
                const double nextLongest = firstGroup < view->nGroups ? view->groupChunksHost[firstGroup] : 0;
                const double side = (double)skippedChunks * 4.0 * kGroupTargets * (double)Q / 1e12;
                const double skip = std::max({balancedRest * tau, nextLongest * tau / alone, side});
                if (skip >= keep) firstGroup = 0;
            }
            const int skipped = std::min(firstGroup * kGroupTargets, view->nPacked);
            for (int k = 0; k < skipped; ++k) queueWhole(sideJobs, view->ids[k]);
            balancedChunks = total / std::max<int64_t>(slots, 1);
        }
        const int firstPos = std::min(firstGroup * kGroupTargets, view->nPacked);
        g_lastRouting[0] = (int64_t)sideJobs.size();
        g_lastRouting[2] = view->nGroups - firstGroup;

        if (view->nGroups > firstGroup) {
            const bool pairStrips = stripRows > 0;
            const int rows = pairStrips ? stripRows : (((Q + nStrips - 1) / nStrips) + 7) / 8 * 8;
            const int qPad = nStrips * rows;
            const int nSym = A + 1;
            // Lane arithmetic. Smith-Waterman: packed half floats are exact for integers
            // below 2048 and cost fewer instructions per cell (interseq_impl.h); saturating
            // int16 is the second rung, the int32 intra-sequence kernel the last. The other
            // modes use signed int16 lanes; whether a target fits is known from its length:
            //   every true H, E, F >= -(3*open + (Q + L)*ext)   and   H <= min(Q, L)*maxScore
            const bool sw = mode == OPAL_MODE_SW;
            const bool locate = searchType != OPAL_SEARCH_SCORE;  // end locations wanted
            int packedSkip = firstPos;   // first view position whose packed result is scattered
            int capGroups = 0, capChunks = 0;
            // Smith-Waterman with several strips and no windows (long queries): a few targets far longer
            // than the rest of the first group set that group's - on the strips kernel the launch's - length
            // (cfg4 with its tail: 4000 .. 8000 residues among 2000-residue targets, 56 instead of 47 ms).
            // With a pair's strips side by side the int32 kernel takes them in a few milliseconds beside the
            // packed launch: up to 64 leading targets more than a quarter longer than the longest of the next
            // group go there, and the first group stops at the longest target that stays.
            if ((sw || globalStrips) && overlap == 0 && Q > kLanes && view->nPacked - firstPos > 2 * kGroupTargets &&
                !getenv("MIOPAL_NO_SIDE_STREAM") && !getenv("MIOPAL_NO_SKIM")) {
                const int ref = dbLen(db, view->ids[firstPos + kGroupTargets]);
                int k = 0;
                while (k < 64 && (int64_t)dbLen(db, view->ids[firstPos + k]) * 4 > (int64_t)ref * 5 + 1024) ++k;
                if (k > 0) {
                    for (int x = 0; x < k; ++x) sideJobs.push_back(forwardJob(view->ids[firstPos + x], rules));
                    g_lastRouting[0] = (int64_t)sideJobs.size();
                    packedSkip = firstPos + k;
                    capGroups = 1;
                    capChunks = std::max(1, (dbLen(db, view->ids[firstPos + k]) + 3) / 4);
                }
            }
            // one strip + Smith-Waterman scores: the pair-indexed LDS profile saves the v_perm per cell
            const char* noPair = getenv("MIOPAL_NO_PAIR_TABLE");
            // first rung of the pair-table kernel: biased integer halves (exact below 25600,
            // interseq_impl.h) when the scores and gap costs leave its guard band alone: a step up
            // (score + ext, or ext - open) of at most 0x0400 so that a finite half cannot jump over
            // the NaN patterns, a step down (score + ext, open - ext) within the room below zero.
            // With end locations every value is scaled by 2^bits (row keys in the low bits).
            const int pairRows = std::max(2, (Q + 1) / 2 * 2);
            // (row keys in the low bits of every value - unless the end locations come from a second sweep)
            const bool twoPass = pairStrips && sw && locate && twoPassEnds;
            const bool rowKeys = locate && !twoPass;
            const int bits = rowKeys ? locRowBitsHost(pairStrips ? stripRows : pairRows) : 0;
            const int64_t up = std::max<int64_t>((int64_t)maxScore + ext, (int64_t)ext - open);
            const int64_t down = std::max<int64_t>(-((int64_t)minScore + ext), (int64_t)open - ext);
            // (A step up of more than 0x0400 could carry a finite half past the NaN patterns, 0x7C00 to
            // 0x7FFF, into the negative ones, where the max would drop it: the limit is then lowered
            // by the excess, so that the cell it would jump from is itself flagged.)
            const bool biasedFits = nStrips == 1 && useHalf && !getenv("MIOPAL_NO_BIASED") &&
                                    (up << bits) <= kBiasedMaxStepUp &&
                                    (down << bits) <= (rowKeys ? kLocGuardBand : kBiasedMaxMagnitude) &&
                                    5 * ((int64_t)ext << bits) <= kLocMaxShift && minScore > kBiasedPad;
            const int biasedLimit =
                (int)(((rowKeys ? 0x7C00 - kLocZeroPattern - kLocMaxShift : kBiasedScoreLimit) -
                       std::max<int64_t>(0, (up << bits) - 0x0400)) >> bits);
            const bool usePair = sw && nStrips == 1 && !(noPair && noPair[0] == '1') &&
                                 interseqPairFits(biasedFits ? pairRows : rows, nSym) && (!locate || biasedFits);
            const bool biased = usePair && biasedFits;
            // Smith-Waterman scores in the general kernel (several strips, or a pair table that does not
            // fit LDS): column-shifted unsigned patterns (ArithSwU16) when the longest packed target
            // leaves a range worth having - zero + ext x columns + score below 0x7C00.
            int swBias = 0, swLimit = 0;
            bool swShifted = false;
            if (sw && !locate && !usePair && !pairStrips && useHalf && !getenv("MIOPAL_NO_SW_SHIFT")) {
                swBias = std::max(0, -(minScore + ext));                       // profile entries s + ext + K >= 0
                const int64_t stepUp = std::max<int64_t>((int64_t)maxScore + ext, (int64_t)ext - open);
                const int64_t lim = 0x7C00 - kSwShiftZero - (int64_t)ext * (view->maxPackedLen + 8) -
                                    std::max<int64_t>(0, stepUp - 0x0400);
                if (lim >= 4096 && swBias + ext <= 0x0800 && open - ext <= 0x0800 && stepUp <= 0x1000 &&
                    (int64_t)maxScore + ext + swBias < 0x4000) {
                    swShifted = true;
                    swLimit = (int)lim;
                }
            }
            // Half floats turn a sum above 65504 into +inf, and inf + (-inf padding) into NaN, which
            // the flag `best >= 2048` would miss (NaN converts to 0): only matrices whose best
            // possible score stays finite take the half-float rung.
            const bool halfFloat = sw && useHalf && !biased && !swShifted && !pairStrips && maxScore <= 1024 && minScore >= -1024 &&
                                   (int64_t)std::min<int64_t>(Q, db->maxLen) * std::max(maxScore, 0) < 60000;
            InterseqFlavour flavour = sw ? (swShifted ? kSwShifted : halfFloat ? kSwHalf : kSwInt16) : kSignedInt16;
            int profileShift = swShifted ? ext + swBias : 0;
            // One-strip NW / HW / OV: the pair-table kernel on biased integer halves (interseq_impl.h).
            // The true values around a pattern's zero are bounded by the query, not by the targets'
            // lengths, so no target is redone at 32 bit; the bounds are static:
            //   below zero: 3 open + (Q + 4) ext + |min S|,   above: Q (max S + ext) + the rebase shift
            const int64_t globalZero = 0x0400 + 3 * (int64_t)open + ((int64_t)Q + 4) * ext + std::max(0, -minScore);
            const bool globalPair =
                !sw && nStrips == 1 && !globalPairRefused && !(noPair && noPair[0] == '1') && !getenv("MIOPAL_NO_BIASED") &&
                interseqPairFits(pairRows, nSym) && minScore > kBiasedPad && (r.topGap ? open >= ext : true) &&
                5 * (int64_t)ext <= kLocMaxShift &&
                // (+ the strip's rows and an opening: the cells are on anti-diagonally shifted scales, round 3)
                globalZero + (int64_t)Q * (std::max(maxScore, 0) + ext) + kLocMaxShift + 5 * (int64_t)ext +
                        std::max(maxScore, 0) + ((int64_t)Q + 4) * ext + open < 0x7C00;
            if (globalPair || globalStrips) {
                // only empty targets (closed forms of the border) are left to the int32 kernel
                for (int e = view->nPacked - 1; e >= firstPos && dbLen(db, view->ids[e]) == 0; --e)
                    jobs.push_back(forwardJob(view->ids[e], rules));
            } else if (!sw) {
                const int64_t pos = std::max(maxScore, 0);
                auto fitsPlain = [&](int64_t L) {
                    return L > 0 && 3 * (int64_t)open + (Q + L) * ext < 32000 && std::min<int64_t>(Q, L) * pos < 32000;
                };
                // the shifted flavour stores X + (i + j) * ext: (Q + L) * ext more head-room
                auto fitsDiag = [&](int64_t L) {
                    return L > 0 && 3 * (int64_t)open + (Q + L + 2) * ext < 32000 &&
                           std::min<int64_t>(Q, L) * pos + (Q + L) * ext < 32000;
                };
                // longest packed target that the plain flavour can take (view order: longest first)
                int firstFit = firstPos;
                while (firstFit < view->nPacked && !fitsPlain(dbLen(db, view->ids[firstFit]))) ++firstFit;
                // The shifted flavours need more head-room, i.e. shorter targets. A few targets too long
                // for them would put the WHOLE view on the plain int16 lanes (8 operations per cell pair
                // instead of 5: cfg4 with the reference's 1000 .. 35000 tail, NW: 78 ms instead of 48); the
                // int32 kernel takes a long pair in a few milliseconds now (one wavefront per strip), so up
                // to 1024 of the longest targets are handed to it when that buys the view a cheaper flavour.
                auto firstThat = [&](int from, auto&& fits) {
                    int p = from;
                    while (p < view->nPacked && p - from <= 1024 && !fits(dbLen(db, view->ids[p]))) ++p;
                    return (p < view->nPacked && p - from <= 1024 && dbLen(db, view->ids[p]) > 0) ? p : -1;
                };
                const int firstDiag = getenv("MIOPAL_NO_DIAG_SHIFT") ? -1 : firstThat(firstFit, fitsDiag);
                if (firstDiag >= 0) {
                    firstFit = firstDiag;
                    flavour = kSignedInt16Diag;
                    profileShift = 2 * ext;
                    // The same shift on unsigned patterns compared as half floats (ArithU16Diag: integer
                    // adds, one max3 for h): scores after the shift must not be negative
                    // (s + ext + open >= 0), open >= ext, and every pattern
                    // zero + x + (i + j) ext within [0, 0x7BFF] for the longest target that stays packed.
                    const int64_t c = (int64_t)open - ext;
                    const int64_t below = 3 * (int64_t)open + 2 * (int64_t)ext + std::max(0, -minScore) + c;
                    auto fitsUnsigned = [&](int64_t L) {
                        return L > 0 && kUnsignedDiagZero + std::min<int64_t>(Q, L) * pos + (Q + L + 2) * (int64_t)ext + pos +
                                                2 * (int64_t)ext + c < 0x7C00;
                    };
                    if (c >= 0 && (int64_t)minScore + ext + open >= 0 &&
                        0x0400 + below + 64 <= kUnsignedDiagZero &&   // real cells stay above the padding cells' floor
                        !getenv("MIOPAL_NO_UNSIGNED_DIAG")) {
                        const int firstU = firstThat(firstFit, fitsUnsigned);
                        if (firstU >= 0) {
                            firstFit = firstU;
                            flavour = kUnsignedDiag;
                            profileShift = 2 * ext + (int)c;
                        }
                    }
                }
                // The targets that do not fit form a prefix of the view: they go to the int32 kernel BESIDE
                // the packed launch, and the scatter starts behind them (their lanes of the packed kernel hold
                // nothing). Empty targets (closed forms of the border) form a suffix, redone after the scatter.
                // (windows of a segmented view are merged by key afterwards: there the whole targets are redone after)
                if (overlap > 0) {
                    for (int k = firstPos; k < firstFit; ++k) queueWhole(jobs, view->ids[k]);
                } else {
                    for (int k = firstPos; k < firstFit; ++k) sideJobs.push_back(forwardJob(view->ids[k], rules));
                    packedSkip = firstFit;
                    g_lastRouting[0] = (int64_t)sideJobs.size();
                    // ... and the groups they sit in sweep no further than the longest target that stays (cfg4
                    // with its tail: five targets of 4000 .. 8000 residues kept the first group, and with it
                    // the launch, at 8000 columns: 57 instead of 48 ms)
                    if (firstFit > firstPos && firstFit < view->nPacked) {
                        capGroups = (firstFit - firstPos + kGroupTargets - 1) / kGroupTargets;
                        capChunks = std::max(1, (dbLen(db, view->ids[firstFit]) + 3) / 4);
                    }
                }
                for (int e = view->nPacked - 1; e >= firstFit && dbLen(db, view->ids[e]) == 0; --e)
                    jobs.push_back(forwardJob(view->ids[e], rules));
            }
            // query profile: profile[t][i] = S[q_i][t]; padding symbol and padding rows can
            // never win a max: -32768 (int16) or -inf (half)
            auto enc = [&](int v) -> int16_t {
                if (!halfFloat) return (int16_t)v;
                const _Float16 h = (_Float16)(float)v;
                int16_t bits;
                memcpy(&bits, &h, sizeof bits);
                return bits;
            };
            // (the unsigned shifted flavour: padding scores open - ext after the shift, see ArithU16Diag)
            const int16_t padValue = (biased || globalPair || pairStrips) ? (int16_t)kBiasedPad
                                     : swShifted ? (int16_t)0   // s + ext + K = 0: a true score of -(ext + K) <= 0
                                     : flavour == kUnsignedDiag ? (int16_t)(open - ext)
                                     : halfFloat ? (int16_t)0xFC00 : (int16_t)-32768;
            std::vector<int16_t> prof((size_t)nSym * qPad, padValue);
            for (int t = 0; t < A; ++t)
                for (int i = 0; i < Q; ++i) prof[(size_t)t * qPad + i] = enc(matrix[query[i] * A + t] + profileShift);
            // targets kept out of the packed view (too long for one lane each) are computed by the
            // int32 kernel on a side stream BESIDE the packed kernel; packed targets that need
            // the int32 kernel are redone after it, because both write the same result slots
            bool forked = sideForked;
            bool directScatter = false;   // the packed kernel wrote database order itself
            if (!sideDone.empty()) {
                sideJobs.erase(std::remove_if(sideJobs.begin(), sideJobs.end(), [&](const PairJob& j) {
                                   return std::binary_search(sideDone.begin(), sideDone.end(), j.out);
                               }), sideJobs.end());
            }
            // The side kernel is handed to its stream BEFORE the packed launch: its wavefronts are dispatched
            // first, over the whole chip, and the persistent packed workgroups (one per CU, every register of
            // it) start on a CU when its side wavefronts are done. (Handed over after the packed launch the
            // side kernel only finds the CUs the launch left out: cfg4 with its tail, 8 CUs: 54 against 44 ms,
            // MIOPAL_PACKED_FIRST.) Beside a strips kernel the side units come in workgroups of 16 wavefronts,
            // so that they hold few CUs (launchIntraseqStrips), and the packed launch takes EVERY CU: the
            // workgroups that start late simply take fewer units.
            bool sidePending = false;
            auto enqueueSide = [&]() -> int {
                RC_TRY(runPairs(sideJobs, false, d_score, d_endI, d_endJ, nullptr, ws->aux, kAuxJobs - kJobs));
                HIP_TRY(hipEventRecord(ws->evJoin, ws->aux));
                for (const PairJob& j : sideJobs) sideDone.push_back(j.out);
                std::sort(sideDone.begin(), sideDone.end());
                sideJobs.clear();
                sidePending = false;
                sideForked = true;   // (a score pass that starts over still joins what is on the side stream)
                spt.mark("    side jobs enqueued");
                return 0;
            };
            if (!sideJobs.empty() && !getenv("MIOPAL_NO_SIDE_STREAM")) {
                RC_TRY(ws->ensureAux());
                RC_TRY(ensurePairInputs());
                HIP_TRY(hipEventRecord(ws->evFork, stream));
                HIP_TRY(hipStreamWaitEvent(ws->aux, ws->evFork, 0));
                forked = true;
                sidePending = true;
                besidePersistent = pairStrips;
                if (!getenv("MIOPAL_PACKED_FIRST")) RC_TRY(enqueueSide());
            }
            void *pp, *vs, *vo, *ct;
            RC_TRY(ws->get(kProfile, prof.size() * sizeof(int16_t), &pp));
            RC_TRY(ws->get(kViewScore, (size_t)view->nGroups * kGroupTargets * sizeof(int32_t), &vs));
            RC_TRY(ws->get(kViewOvf, (size_t)view->nGroups * kGroupTargets, &vo));
            RC_TRY(ws->get(kCounter, sizeof(int32_t), &ct));
            RC_TRY(ws->stageUpload(pp, prof.data(), prof.size() * sizeof(prof[0]), stream));
            // lanes can only leave the exact range when min(Q, L) * maxScore reaches the limit
            // (also bounded by the query itself: every residue is aligned at most once, at best with
            // its most favourable partner - 280 for the 53-aa README query under BLOSUM62, where
            // Q * max(S) says 583)
            int64_t queryBest = 0;
            for (int i = 0; i < Q; ++i) {
                int rowMax = 0;
                for (int t = 0; t < A; ++t) rowMax = std::max(rowMax, matrix[query[i] * A + t]);
                queryBest += rowMax;
            }
            const int64_t reach = std::min<int64_t>((int64_t)std::min(Q, view->maxPackedLen) * std::max(maxScore, 0), queryBest);
            const int64_t limit = (biased || pairStrips) ? biasedLimit : swShifted ? swLimit : halfFloat ? 2048 : 32767;
            // (the multi-strip NW / HW / OV kernel flags nothing for its range; the count brings back the
            // lanes of units that gave up on the strip above - never seen outside the fault-injection test)
            // (the strips kernels: always - the count also brings back the lanes of a unit that gave up on
            // the strip above it; searches of several strips take milliseconds, the 4-byte download is free)
            const bool mayOverflow = sw ? (reach >= limit || pairStrips) : globalStrips;
            // How many flagged lanes are redone one by one before the whole view takes the next rung: the int32
            // kernel fills about 1e12 cells a second, the next rung 5e12 .. 8e12 over the WHOLE view - an eighth
            // of the view's targets costs the same either way (round 3; it was 2048 whatever the size: 0.3 % of
            // 1M x 300 beyond the row keys' 384 at Q = 1000 sent the other 99.7 % through a second launch)
            const int64_t directLimit = getenv("MIOPAL_FIXED_DIRECT_LIMIT") ? kMaxDirectRecompute
                                        : std::max<int64_t>(kMaxDirectRecompute, (view->nPacked - packedSkip) / 8);
            if (mayOverflow) HIP_TRY(hipMemsetAsync(ct, 0, sizeof(int32_t), stream));
            InterseqArgs ia{};
            ia.pack = view->d_pack;
            ia.groupOff = view->d_groupOff;
            ia.groupChunks = view->d_groupChunks;
            ia.nGroups = view->nGroups - firstGroup;
            ia.groupBase = firstGroup;
            ia.profile = (const int16_t*)pp;
            ia.nSymbols = nSym;
            ia.qPad = qPad;
            ia.nStrips = nStrips;
            ia.qLen = Q;
            ia.gapOpen = std::min(open, 32767);
            ia.gapExt = std::min(ext, 32767);
            ia.topGap = r.topGap;
            ia.leftGap = r.leftGap;
            ia.region = r.region;
            ia.lens = view->d_lens;
            ia.score = (int32_t*)vs;
            if (locate) {
                void *vi, *vj;
                RC_TRY(ws->get(kViewEndI, (size_t)view->nGroups * kGroupTargets * sizeof(int32_t), &vi));
                RC_TRY(ws->get(kViewEndJ, (size_t)view->nGroups * kGroupTargets * sizeof(int32_t), &vj));
                ia.endI = (int32_t*)vi;
                ia.endJ = (int32_t*)vj;
            }
            ia.overflow = (sw || globalStrips) ? (uint8_t*)vo : nullptr;
            ia.stripSpinCap = faultSpinCap;
            ia.faultUnit1 = faultKind == 1 ? faultUnit + 1 : 0;
            ia.biasedLimit = swShifted ? swLimit : biasedLimit;
            ia.scoreBias = swBias;
            ia.biasedZero = (int)globalZero;
            ia.boundaryOff = view->d_boundaryOff;
            ia.capGroups = capGroups;
            ia.capChunks = capChunks;
            ia.priorityChunks = getenv("MIOPAL_NO_PRIORITY") ? INT32_MAX : (int)std::min<int64_t>(std::max<int64_t>(balancedChunks, 16), INT32_MAX);
            if ((nStrips + waves - 1) / waves > 1) {
                void *b0, *b1;
                const size_t bytes = (size_t)view->totalChunks * 4 * kLanes * sizeof(uint2);
                RC_TRY(ws->get(kBoundary0, bytes, &b0));
                RC_TRY(ws->get(kBoundary1, bytes, &b1));
                ia.boundary[0] = (uint2*)b0;
                ia.boundary[1] = (uint2*)b1;
            }
            // several rounds of strips per group, scores only: (group, round) units instead of one
            // workgroup per group (interseq_impl.h, "unit mode")
            // Only when the groups are few for the chip (under six workgroup-lifetimes): the rounds of
            // a group are then far apart in time and its boundary rows come back from HBM, not from
            // the caches (500k x 300 at Q = 300, 7.6 lifetimes: 6.6 ms classic, 7.0 ms in units;
            // 100k x 2000 at Q = 2000, 3.05 lifetimes: 66 ms classic, 55 ms in units).
            const int64_t unitSlots = (int64_t)db->computeUnits * std::max(1, 8 / waves);
            const char* um = getenv("MIOPAL_UNITS");
            const bool wantUnits = um ? um[0] == '1' : (int64_t)(view->nGroups - firstGroup) < 6 * unitSlots;
            if (pairStrips) {
                // unit counter + chunks published per (group, strip); scores and flags start from zero
                // (a group's answer is the maximum over its strips' units)
                void* us;
                const size_t ints = (size_t)ia.nGroups * nStrips + 2;
                RC_TRY(ws->get(kUnitState, ints * sizeof(int), &us));
                HIP_TRY(hipMemsetAsync(us, 0, ints * sizeof(int), stream));
                if (mayOverflow && sw) {
                    // more flagged lanes than are redone one by one: the launch stops, the view takes the next rung
                    ia.stripAbort = (int*)us + ints - 1;
                    ia.stripAbortAt = (int)std::min<int64_t>(2 * directLimit, INT32_MAX / 2);
                    ia.stripGaveUp = (int*)ct;
                }
                // (the scores-only form of the NW / HW / OV kernel folds OV's candidates into the view scores,
                // and what a unit that gave up leaves behind is "minus infinity" in every mode)
                if (globalStrips) HIP_TRY(launchFillInt32((int32_t*)vs, view->nGroups * kGroupTargets, INT32_MIN, stream));
                else HIP_TRY(hipMemsetAsync(vs, 0, (size_t)view->nGroups * kGroupTargets * sizeof(int32_t), stream));
                HIP_TRY(hipMemsetAsync(vo, 0, (size_t)view->nGroups * kGroupTargets, stream));
                ia.unitCounter = (int*)us;
                ia.unitFlags = (int*)us + 1;
                if (locate) {
                    void* sk;
                    RC_TRY(ws->get(kStripKeys, (size_t)view->nGroups * kGroupTargets * sizeof(unsigned long long), &sk));
                    HIP_TRY(hipMemsetAsync(sk, 0, (size_t)view->nGroups * kGroupTargets * sizeof(unsigned long long), stream));
                    ia.stripKeys = (unsigned long long*)sk;
                }
            } else if ((nStrips + waves - 1) / waves > 1 && !locate && !usePair && !globalPair && wantUnits) {
                void *us, *up;
                const size_t ints = (size_t)ia.nGroups + 1;
                RC_TRY(ws->get(kUnitState, ints * sizeof(int), &us));
                RC_TRY(ws->get(kUnitPartial, (size_t)ia.nGroups * waves * kLanes * sizeof(uint2), &up));
                HIP_TRY(hipMemsetAsync(us, 0, ints * sizeof(int), stream));
                ia.unitCounter = (int*)us;
                ia.unitFlags = (int*)us + 1;
                ia.unitPartial = (uint2*)up;
            }
            const bool timed = db->profiling.load() != 0;
            EventPair ev;
            if (timed) {
                HIP_TRY(hipEventCreate(&ev.e0));
                HIP_TRY(hipEventCreate(&ev.e1));
                HIP_TRY(hipEventRecord(ev.e0, stream));
            }
            g_lastRouting[1] = 1 + 32 * (int)flavour;  // general kernel and its lane arithmetic
            if (pairStrips) {
                int pairUnits = db->computeUnits;
                if (const char* r = getenv("MIOPAL_RESERVE_CUS"))
                    pairUnits = std::max(1, pairUnits - std::max(0, atoi(r)));
                else if (forked && getenv("MIOPAL_STRIPS_RESERVE"))
                    // (round 2 kept CUs out of the launch for the side kernel, one per 256 pairs, at least 8;
                    // the units are taken dynamically, so a workgroup whose CU is busy with side wavefronts
                    // at first just starts later and takes fewer)
                    pairUnits = std::max(1, pairUnits - (int)std::min<int64_t>(pairUnits / 4, std::max<int64_t>(8, (g_lastRouting[0] + 255) / 256)));
                const PairFlavour stripsFlavour = globalStrips ? kPairGlobalStrips : kPairSwStrips;
                g_lastRouting[1] = 2 + (int)stripsFlavour;
                // few (group, strip) units: fewer groups per workgroup, so that every CU gets a unit and a
                // wavefront shares its SIMD with fewer others
                ia.batchGroups = (int)std::max<int64_t>(1, std::min<int64_t>(12, (int64_t)ia.nGroups * nStrips / std::max(1, pairUnits)));
                if (const char* bg = getenv("MIOPAL_BATCH_GROUPS"))   // experiments
                    ia.batchGroups = std::max(1, std::min(12, atoi(bg)));
                hipError_t pe = hipSuccess;
                // (random pairs only get there in the linear regime of the scoring system, and then score
                // about half a unit per aligned residue: nothing to probe for under ~500 residues)
                if (sw && rowKeys && mayOverflow && std::min(Q, view->maxPackedLen) >= 512 && !getenv("MIOPAL_PAIR_STRIPS")) {
                    // With end locations a lane is exact below 384 (768 for strips of 32 rows). Scores of
                    // long queries against long targets under cheap gaps are in the thousands - every lane
                    // would be redone, and a launch that gives up half-way has cost half its time. The
                    // twelve longest groups (the highest scores) go through the scores-only kernel first:
                    // 1536 targets, a unit's time; when half of them are beyond the range the general
                    // kernel takes the search.
                    InterseqArgs probe = ia;
                    probe.nGroups = std::min(12, ia.nGroups);
                    probe.batchGroups = 1;
                    probe.overflow = nullptr;
                    probe.stripKeys = nullptr;
                    probe.stripAbort = nullptr;
                    pe = launchInterseqPair(probe, rows, kPairSwStrips, pairUnits, stream, false);
                    if (pe == hipSuccess) {
                        const int lanes = probe.nGroups * kGroupTargets;
                        std::vector<int32_t> seen((size_t)lanes);
                        RC_TRY(ws->stageDownload(seen.data(), ia.score + (size_t)firstGroup * kGroupTargets, (size_t)lanes * sizeof(int32_t)));
                        RC_TRY(ws->finishDownloads());
                        int beyond = 0;
                        for (int32_t v : seen) beyond += v >= biasedLimit;
                        if (2 * beyond >= lanes) {
                            // (round 3: two sweeps of this kernel instead of the general kernel's row scans;
                            // MIOPAL_NO_TWO_PASS_ENDS restores round 2)
                            stripsEndsDeclined = true;
                            twoPassEnds = !getenv("MIOPAL_NO_TWO_PASS_ENDS");
                            return scorePassImpl(d_score, d_endI, d_endJ, useHalf);
                        }
                        // (the probe's units and scores are wiped: the real launch starts from zero)
                        HIP_TRY(hipMemsetAsync(ia.unitCounter, 0, ((size_t)ia.nGroups * nStrips + 2) * sizeof(int), stream));
                        HIP_TRY(hipMemsetAsync(ia.score, 0, (size_t)view->nGroups * kGroupTargets * sizeof(int32_t), stream));
                    }
                }
                if (pe == hipSuccess) pe = launchInterseqPair(ia, rows, stripsFlavour, pairUnits, stream, rowKeys);
                if (pe == hipSuccess && sidePending) RC_TRY(enqueueSide());
                if (pe == hipSuccess && twoPass) {
                    // second sweep: the first cell (column-major) that holds each target's score, as keys
                    InterseqArgs second = ia;
                    second.known = ia.score;
                    second.overflow = nullptr;
                    second.stripAbort = nullptr;
                    second.faultUnit1 = 0;
                    HIP_TRY(hipMemsetAsync(ia.unitCounter, 0, ((size_t)ia.nGroups * nStrips + 1) * sizeof(int), stream));
                    pe = launchInterseqPair(second, rows, stripsFlavour, pairUnits, stream, false);
                }
                if (pe != hipSuccess) {
                    // (e.g. the runtime refuses 150 KB of dynamic LDS: start over on the general kernel)
                    (void)hipGetLastError();
                    (globalStrips ? globalStripsRefused : pairStripsRefused) = true;
                    if (getenv("MIOPAL_VERBOSE"))
                        fprintf(stderr, "miopal: multi-strip pair-table kernel refused (%s), using the general kernel\n",
                                hipGetErrorString(pe));
                    return scorePassImpl(d_score, d_endI, d_endJ, useHalf);
                }
                // (score, column, row) keys merged over the strips -> view-order scores and end locations
                if (globalStrips && locate)
                    HIP_TRY(launchDecodeGlobalKeys(ia.stripKeys, view->d_lens, view->nGroups * kGroupTargets, Q, ia.score,
                                                   ia.endI, ia.endJ, stream));
                else if (locate)
                    HIP_TRY(launchDecodeStripKeys(ia.stripKeys, view->nGroups * kGroupTargets, ia.score, ia.endI, ia.endJ, stream));
            } else if (usePair || globalPair) {
                void* wc;
                RC_TRY(ws->get(kWorkCounter, sizeof(int), &wc));
                HIP_TRY(hipMemsetAsync(wc, 0, sizeof(int), stream));
                ia.workCounter = (int*)wc;
                // The persistent workgroups fill every CU (LDS and registers): a kernel of another
                // stream - the collective that gathers the previous search's scores - would wait for
                // them to leave. MIOPAL_RESERVE_CUS keeps a few CUs out of the launch for it.
                int pairUnits = db->computeUnits;
                if (const char* r = getenv("MIOPAL_RESERVE_CUS"))
                    pairUnits = std::max(1, pairUnits - std::max(0, atoi(r)));
                else if (forked)
                    // The persistent workgroups hold their CU's registers for the whole launch: the
                    // wavefront-per-pair kernel on the side stream only finds room as they leave, i.e. it
                    // runs AFTER the packed kernel (log-normal lengths, NW at Q = 53, 6400 pairs on the side:
                    // 2.07 ms; with CUs kept out of the persistent launch 1.66 ms). One CU per 256 pairs.
                    pairUnits = std::max(1, pairUnits - (int)std::min<int64_t>(pairUnits / 4, std::max<int64_t>(8, (g_lastRouting[0] + 255) / 256)));
                // groups of similar length: every SIMD takes the same share of them (interseq_impl.h)
                {
                    const int blocks = std::max(1, std::min(pairUnits, ia.nGroups));
                    const int longest = view->groupChunksHost[firstGroup];
                    const int shortest = view->groupChunksHost[view->nGroups - 1];
                    const char* tt = getenv("MIOPAL_TAIL_THROTTLE");
                    const bool uniform = (int64_t)shortest * 5 >= (int64_t)longest * 4;
                    ia.tailThrottle = (tt ? tt[0] == '1' : uniform) ? (ia.nGroups + blocks * 4 - 1) / (blocks * 4) : 0;
                }
                // Headline fast path: one strip, Smith-Waterman scores, no lane can leave its range, nothing
                // else writes the results (no side jobs, no skipped groups, no windows): the kernel writes
                // database order itself - into the caller's device buffer, or for miopalSearch into the
                // pinned host buffer the results leave from (no scatter kernel, no device-to-host copy).
                if (biased && !locate && !mayOverflow && overlap == 0 && !forked && sideJobs.empty() && jobs.empty() &&
                    firstGroup == 0 && packedSkip == 0 && !getenv("MIOPAL_NO_DIRECT_SCATTER")) {
                    int32_t* target = d_score;
                    if (hostScoreOut && (hostScoreIsCallers || hostScoreGeneration == ws->stagingGeneration) &&
                        !getenv("MIOPAL_NO_HOST_SCATTER")) {
                        target = hostScoreOut;
                        wroteHost = true;
                    }
                    ia.directOut = target - start;
                    ia.directIds = view->d_ids;
                    ia.directN = view->nPacked;
                    ia.overflow = nullptr;
                    directScatter = true;
                }
                const PairFlavour pf = globalPair ? kPairGlobalBiased : biased ? kPairSwBiased : halfFloat ? kPairSwHalf : kPairSwInt16;
                g_lastRouting[1] = 2 + (int)pf;
                // the biased kernel exists for every even number of rows: no padding rows to 8
                const hipError_t pe = launchInterseqPair(ia, (biased || globalPair) ? pairRows : rows, pf, pairUnits, stream, locate);
                if (pe != hipSuccess) {
                    // e.g. the runtime refuses 150 KB of dynamic LDS: use the v_perm variant (the
                    // biased profile is a plain int16 profile whose padding score, -1024, cannot raise
                    // a Smith-Waterman maximum either)
                    (void)hipGetLastError();
                    if (globalPair) {
                        // (its profile and its routing of long targets do not suit the general kernel:
                        // start over without it)
                        globalPairRefused = true;
                        return scorePassImpl(d_score, d_endI, d_endJ, useHalf);
                    }
                    g_lastRouting[1] |= 16;
                    if (getenv("MIOPAL_VERBOSE"))
                        fprintf(stderr, "miopal: pair-table kernel refused (%s), using the general kernel\n",
                                hipGetErrorString(pe));
                    if (directScatter) {   // (the general kernel writes view order)
                        directScatter = wroteHost = false;
                        ia.directOut = nullptr;
                        ia.overflow = (uint8_t*)vo;
                    }
                    HIP_TRY(launchInterseq(ia, rows, waves, flavour, locate, stream));
                }
            } else {
                HIP_TRY(launchInterseq(ia, rows, waves, flavour, locate, stream));
            }
            if (timed) {
                HIP_TRY(hipEventRecord(ev.e1, stream));
                std::lock_guard<std::mutex> g(db->timingMutex);
                ws->timings.emplace_back(ev.release());
                db->lastTimed = ws;
            }
            const int nScatter = view->nPacked - packedSkip;
            if (keyed) {
                // nothing else writes the results of a segmented search before this point: no
                // target is kept out of the view, no group is skipped (limit >= one window)
                if (firstPos != 0 || !sideJobs.empty() || forked)
                    return fail(MIOPAL_ERR_INTERNAL, "segmented view with side jobs");
                HIP_TRY(launchScatterKeyed(ia.score, ia.endI, ia.endJ, (const uint8_t*)vo, view->d_ids,
                                           view->d_segStart, nScatter, start, (unsigned long long*)keys,
                                           mayOverflow ? (int32_t*)ct : nullptr, stream, keyBias));
                HIP_TRY(launchDecodeKeys((const unsigned long long*)keys, (int)n, d_score, d_endI, d_endJ, stream, keyBias));
            } else if (!directScatter) {
                HIP_TRY(launchScatter(ia.score + packedSkip, (const uint8_t*)vo + packedSkip, view->d_ids + packedSkip,
                                      nScatter, start, d_score, mayOverflow ? (int32_t*)ct : nullptr, overlap > 0,
                                      stream));
                if (locate)
                    HIP_TRY(launchScatterEnds(ia.endI + packedSkip, ia.endJ + packedSkip, view->d_ids + packedSkip,
                                              nScatter, start, d_endI, d_endJ, stream));
            }
            if (forked) HIP_TRY(hipStreamWaitEvent(stream, ws->evJoin, 0));
            spt.mark("    packed kernel enqueued");
            if (spt.on) {
                HIP_TRY(hipStreamSynchronize(stream));
                spt.mark("    packed + side kernels done");
            }
            if (mayOverflow) {
                int32_t count = 0;
                RC_TRY(ws->stageDownload(&count, ct, sizeof(int32_t)));
                RC_TRY(ws->finishDownloads());
                if (pairStrips && sw && rowKeys && count > directLimit && !twoPassEnds && !getenv("MIOPAL_NO_TWO_PASS_ENDS")) {
                    // many lanes left the row keys' range (384 .. 768): the scores' own range is 25600 -
                    // two sweeps of the strips kernel before the int16 rung
                    twoPassEnds = true;
                    return scorePassImpl(d_score, d_endI, d_endJ, useHalf);
                }
                if ((halfFloat || biased || swShifted || pairStrips) && count > directLimit) {
                    // many targets left the half-float range: second rung, int16 lanes,
                    // over the whole view (its results overwrite the first pass)
                    return scorePassImpl(d_score, d_endI, d_endJ, false);
                }
                if (count > 0) {
                    std::vector<uint8_t> flags((size_t)view->nPacked);
                    RC_TRY(ws->stageDownload(flags.data(), vo, flags.size()));
                    RC_TRY(ws->finishDownloads());
                    for (int k = packedSkip; k < view->nPacked; ++k)
                        if (flags[k]) queueWhole(jobs, view->ids[k]);
                    g_lastRouting[3] = count;
                }
            }
        }
        jobs.insert(jobs.end(), sideJobs.begin(), sideJobs.end());
        return runPairs(jobs, false, d_score, d_endI, d_endJ, nullptr);
    }
};

int validate(const MiopalDb* db, const unsigned char* query, int Q, const int* matrix, int A,
             int searchType, int mode, int64_t start, int64_t end) {
    if (!db) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null database handle");
    if (mode < OPAL_MODE_NW || mode > OPAL_MODE_SW) return fail(OPAL_ERR_INVALID_MODE, "invalid alignment mode %d", mode);
    if (searchType < OPAL_SEARCH_SCORE || searchType > OPAL_SEARCH_ALIGNMENT)
        return fail(OPAL_ERR_INVALID_MODE, "invalid search type %d", searchType);
    if (Q < 0 || (Q > 0 && !query)) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad query");
    if (!matrix || A <= 0 || A > kMaxAlphabet) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad score matrix / alphabet length %d", A);
    if (A != db->alphabet) return fail(MIOPAL_ERR_BAD_ARGUMENT, "alphabet length %d differs from the database's %d", A, db->alphabet);
    for (int i = 0; i < Q; ++i)
        if (query[i] >= A) return fail(MIOPAL_ERR_BAD_ARGUMENT, "query residue %d out of range at %d", query[i], i);
    if (start < 0 || end < start || end > db->count) return fail(MIOPAL_ERR_BAD_ARGUMENT, "bad slice [%lld, %lld)", (long long)start, (long long)end);
    return 0;
}

// The database's residues as the caller holds them: one flat array, or one pointer per sequence
// (opalSearchDatabase's `unsigned char* db[]`).
struct ResidueSource {
    const unsigned char* flat = nullptr;
    const unsigned char* const* sequences = nullptr;
};

int newHandle(std::unique_ptr<MiopalDb>* out, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(OPAL_ERR_NO_SIMD_SUPPORT, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MIOPAL_ERR_BAD_ARGUMENT, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<MiopalDb> db(new MiopalDb());
    db->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        db->computeUnits = prop.multiProcessorCount;
    // cached packed views may take a third of the device's memory (MIOPAL_VIEW_CACHE_MB overrides)
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB > 0) db->viewBudgetBytes = totalB / 3;
    if (const char* mb = getenv("MIOPAL_VIEW_CACHE_MB")) db->viewBudgetBytes = (size_t)std::max(0, atoi(mb)) << 20;
    *out = std::move(db);
    return 0;
}

// (Re)fills a handle nobody else is using: the sequences replace whatever it held; device
// allocations that are large enough stay, the blocks of its cached views are kept for the new views.
int fillHandle(MiopalDb* db, const ResidueSource& src, std::vector<int64_t>&& offsets, int64_t count,
               int alphabetLength, bool prefetchView = false) {
    const int device = db->device;
    HIP_TRY(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> g(db->viewMutex);
        for (auto& slot : db->views) {
            View* v = slot.view.get();
            if (!v || slot.view.use_count() != 1) continue;
            for (auto blk : {std::make_pair((void**)&v->d_pack, v->packCap), std::make_pair(&v->d_meta, v->metaCap)})
                if (*blk.first && db->spareBlocks.size() < kSpareBlocks) {
                    db->spareBlocks.emplace_back(*blk.first, blk.second);
                    *blk.first = nullptr;
                }
        }
        db->views.clear();
        db->prefetched.reset();
    }
    db->alphabet = alphabetLength;
    db->count = count;
    db->offsets = std::move(offsets);
    const std::vector<int64_t>& off = db->offsets;
    db->total = off[(size_t)count];
    db->maxLen = 0;
    {
        const int nSlices = hostThreads((size_t)count, 65536);
        std::vector<int64_t> longest((size_t)nSlices, 0);
        parallelSlices(nSlices, [&](int t) {
            int64_t m = 0;
            for (int64_t k = count * t / nSlices; k < count * (t + 1) / nSlices; ++k) m = std::max(m, off[(size_t)k + 1] - off[(size_t)k]);
            longest[(size_t)t] = m;
        });
        for (int64_t m : longest) db->maxLen = std::max(db->maxLen, m);
    }
    const size_t wantRes = (size_t)db->total + 64, wantOff = (size_t)(count + 1) * sizeof(int64_t);
    if (db->residueCap < wantRes || db->residueCap / 4 > wantRes + (1u << 20)) {
        if (db->d_residues) HIP_TRY(hipFree(db->d_residues));
        db->d_residues = nullptr;
        db->residueCap = 0;
        HIP_TRY(hipMalloc(&db->d_residues, wantRes));
        db->residueCap = wantRes;
    }
    if (db->offsetsCap < wantOff || db->offsetsCap / 4 > wantOff + (1u << 20)) {
        if (db->d_offsets) HIP_TRY(hipFree(db->d_offsets));
        db->d_offsets = nullptr;
        db->offsetsCap = 0;
        HIP_TRY(hipMalloc(&db->d_offsets, wantOff));
        db->offsetsCap = wantOff;
    }
    // opalSearchDatabase hands the whole database over on every call and searches all of it at once: the
    // lists of that search's packed view only need the lengths, so they are built (and their small arrays
    // uploaded) by another thread WHILE the residues cross PCIe; the search then only packs. Only for
    // databases whose targets are shorter than any window of a segmented view (the search would ask for
    // another view otherwise). A failure here is no error: the search builds its view itself.
    std::thread viewThread;
    if (prefetchView && count >= 65536 && db->maxLen <= 256 + 128 && !getenv("MIOPAL_NO_VIEW_PREFETCH")) {
        try {
            viewThread = std::thread([db, count, device] {
                if (hipSetDevice(device) != hipSuccess) return;
                std::shared_ptr<View> v;
                try {
                    if (buildView(db, 0, count, 0, &v, false) != 0) {
                        (void)hipGetLastError();
                        return;
                    }
                } catch (...) {
                    return;
                }
                std::lock_guard<std::mutex> g(db->viewMutex);
                db->prefetched = std::move(v);
            });
        } catch (const std::exception&) {
        }
    }
    struct Joiner {
        std::thread& t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{viewThread};
    // residues: gathered (or copied) piece by piece into the bounce pieces and checked there
    const unsigned limit = (unsigned)alphabetLength;
    int bad = 0;
    RC_TRY(streamedUpload(device, db->d_residues, (size_t)db->total,
        [&](unsigned char* dst, size_t at, size_t n) {
            Bytes16 top = {};
            unsigned topTail = 0;
            if (src.flat) {
                copyWithMax(dst, src.flat + at, n, top, topTail);
            } else {
                // first sequence that ends beyond `at`
                size_t k = (size_t)(std::upper_bound(off.begin(), off.begin() + count + 1, (int64_t)at) - off.begin()) - 1;
                size_t pos = at;
                const size_t stop = at + n;
                while (pos < stop) {
                    const size_t seqEnd = (size_t)off[k + 1];
                    if (seqEnd <= pos) { ++k; continue; }   // (empty sequences)
                    const size_t m = std::min(seqEnd, stop) - pos;
                    copyWithMax(dst + (pos - at), src.sequences[k] + (pos - (size_t)off[k]), m, top, topTail);
                    pos += m;
                }
            }
            const unsigned largest = largestByte(top, topTail);
            return largest >= limit ? (int)largest + 1 : 0;
        }, &bad));
    if (bad) return fail(MIOPAL_ERR_BAD_ARGUMENT, "residue %d out of range for alphabet %d", bad - 1, alphabetLength);
    RC_TRY(uploadOnce(device, db->d_offsets, off.data(), wantOff));
    return 0;
}

int createCommon(MiopalDb** out, const ResidueSource& src, std::vector<int64_t>&& offsets,
                 int64_t count, int alphabetLength, int device) {
    std::unique_ptr<MiopalDb> db;
    RC_TRY(newHandle(&db, device));
    RC_TRY(fillHandle(db.get(), src, std::move(offsets), count, alphabetLength));
    *out = db.release();
    return 0;
}

// Handles of finished opalSearchDatabase calls, kept (with their device memory, bounce buffers and
// streams) for the next call of the process: the reference's entry point hands the whole database
// over on every call, and allocating and releasing a handle's resources costs as much as the search.
// Bounded by count and by the device memory a parked handle may hold (MIOPAL_SPARE_HANDLE_MB,
// default 4096; 0 keeps none); miopalReleaseCaches() drops them.
constexpr size_t kSpareHandles = 4;
struct SpareHandles {
    std::mutex m;
    std::vector<std::unique_ptr<MiopalDb>> idle;
};
SpareHandles& spareHandles() {
    static SpareHandles* s = new SpareHandles();   // never destroyed: the runtime may be gone at exit
    return *s;
}

int64_t handleDeviceBytes(MiopalDb* db) {
    int64_t t = (int64_t)(db->residueCap + db->offsetsCap);
    {
        std::lock_guard<std::mutex> g(db->viewMutex);
        for (auto& v : db->views) t += v.view ? (int64_t)(v.view->packCap + v.view->metaCap) : 0;
        for (auto& b : db->spareBlocks) t += (int64_t)b.second;
    }
    std::lock_guard<std::mutex> g(db->wsMutex);
    for (auto& w : db->ownedFree) t += (int64_t)w->bytes();
    return t;
}


// No C++ exception leaves the C ABI (the callers are C, Cython with the GIL released, ctypes): whatever
// the host code throws - in practice std::bad_alloc from a container of a million entries - becomes
// MIOPAL_ERR_INTERNAL with the reason in miopalLastError().
template <class F>
int guarded(F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
    } catch (const std::exception& e) {
        return fail(MIOPAL_ERR_INTERNAL, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(MIOPAL_ERR_INTERNAL, "unexpected exception");
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int miopalDeviceCount(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
    return usable;
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
        int64_t* d_src = nullptr;
        HIP_TRY(hipMalloc(&d_src, (size_t)count * sizeof(int64_t)));
        int rc = uploadOnce(db->device, d_src, srcStart.data(), (size_t)count * sizeof(int64_t));
        hipError_t e = hipSuccess;
        if (rc == 0) {
            e = launchGatherSequences(parent->d_residues, d_src, db->d_offsets, count, db->d_residues, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        }
        (void)hipFree(d_src);
        if (rc) return rc;
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

static int searchImpl(MiopalDb* db, const unsigned char* query, int queryLength, int gapOpen, int gapExt,
                      const int* scoreMatrix, int alphabetLength, int searchType, int mode, int64_t start,
                      int64_t end, int* score, int* endTarget, int* endQuery, int* startTarget,
                      int* startQuery, unsigned char** alignment, int* alignmentLength,
                      HostBytes* flatOps, int64_t* flatOff) {
    RC_TRY(validate(db, query, queryLength, scoreMatrix, alphabetLength, searchType, mode, start, end));
    const int64_t n = end - start;
    if (n == 0) return 0;
    if (!score) return fail(MIOPAL_ERR_BAD_ARGUMENT, "null score output");
    if (searchType >= OPAL_SEARCH_SCORE_END && (!endTarget || !endQuery))
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "null end-location outputs");
    const bool flat = flatOps != nullptr;
    if (searchType == OPAL_SEARCH_ALIGNMENT &&
        (!startTarget || !startQuery || (flat ? !flatOff : (!alignment || !alignmentLength))))
        return fail(MIOPAL_ERR_BAD_ARGUMENT, "null alignment outputs");
    HIP_TRY(hipSetDevice(db->device));
    PhaseTimer pt;
    WorkspaceLease lease(db);
    RC_TRY(lease.acquireInternal());
    Workspace* ws = lease.ws;
    hipStream_t stream = ws->stream;
    Search s{db, ws, stream, query, queryLength, gapOpen, gapExt, alphabetLength, searchType, mode,
             scoreMatrix, start, end, n};
    RC_TRY(s.prepare());
    pt.mark("workspace + query");

    void *ps, *pi = nullptr, *pj = nullptr;
    RC_TRY(ws->get(kScore, (size_t)n * sizeof(int32_t), &ps));
    const bool wantEnd = searchType >= OPAL_SEARCH_SCORE_END;
    if (wantEnd) {
        RC_TRY(ws->get(kEndI, (size_t)n * sizeof(int32_t), &pi));
        RC_TRY(ws->get(kEndJ, (size_t)n * sizeof(int32_t), &pj));
    }
    // scores of a plain score search may leave the kernel straight for the pinned staging buffer
    size_t hostOff = 0;
    const size_t hostBytes = ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;
    bool callerPinned = false;
    if (searchType == OPAL_SEARCH_SCORE) {
        // A caller whose result array is itself pinned, device-visible host memory (hipHostMalloc,
        // hipHostRegister, a pinned torch tensor) gets the scores written into it by the kernel: no
        // bounce buffer, no copy on the host.
        hipPointerAttribute_t attr;
        if (!getenv("MIOPAL_NO_CALLER_PINNED") && hipPointerGetAttributes(&attr, score) == hipSuccess &&
            attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr) {
            s.hostScoreOut = (int32_t*)attr.devicePointer;
            s.hostScoreIsCallers = true;
            callerPinned = true;
        } else {
            (void)hipGetLastError();   // (an ordinary pointer is "invalid value" to the runtime)
            // (room for the pass's own small uploads behind it, so that nothing drains or reallocates the
            // buffer before the kernel is launched; if it happens all the same the generation tells)
            RC_TRY(ws->reserveStaging(hostBytes + (1u << 20)));
            hostOff = ws->pinnedUsed;
            ws->pinnedUsed += hostBytes;
            s.hostScoreOut = (int32_t*)((char*)ws->pinned + hostOff);
            s.hostScoreGeneration = ws->stagingGeneration;
        }
    }
    RC_TRY(s.scorePass((int32_t*)ps, (int32_t*)pi, (int32_t*)pj));
    if (s.wroteHost && callerPinned) {
        // (nothing to copy: visible to the host once the stream has drained, below)
    } else if (s.wroteHost) {
        if (s.hostScoreGeneration != ws->stagingGeneration)
            return fail(MIOPAL_ERR_INTERNAL, "the staging buffer was drained under a kernel that writes to it");
        ws->pending.push_back({score, hostOff, (size_t)n * sizeof(int32_t)});
    } else {
        RC_TRY(ws->stageDownload(score, ps, (size_t)n * sizeof(int)));
    }
    if (wantEnd) {
        RC_TRY(ws->stageDownload(endQuery, pi, (size_t)n * sizeof(int)));
        RC_TRY(ws->stageDownload(endTarget, pj, (size_t)n * sizeof(int)));
    }
    if (s.d_stripError) RC_TRY(ws->stageDownload(&s.stripErrorHost, s.d_stripError, sizeof(int)));
    RC_TRY(ws->finishDownloads());
    HIP_TRY(hipStreamSynchronize(stream));
    RC_TRY(s.checkStripError());
    pt.mark("score/end pass + D2H");
    if (searchType != OPAL_SEARCH_ALIGNMENT) return 0;

    DpRules fr;
    RC_TRY(s.rulesFor(mode, &fr));

    // ---- queries of one strip: the rest of the pipeline stays in HBM ---------------
    // (start cells, traceback jobs, direction bytes and operations are produced and
    // consumed on the device; the host only prefix-sums the alignment lengths)
    {
        const bool oneStrip = queryLength <= kLanes;
        bool deviceFull = queryLength > 0 && db->maxLen > 0 && !getenv("MIOPAL_HOST_TRACEBACK");
        if (deviceFull) {
            HostBytes localOps;
            std::vector<int64_t> localOff;
            HostBytes* outOps = flatOps;
            int64_t* outOff = flatOff;
            if (!flat) {
                localOff.assign((size_t)n + 1, 0);
                outOps = &localOps;
                outOff = localOff.data();
            }
            void *rs = nullptr, *ri = nullptr, *rj = nullptr, *pjobs, *psq, *pst, *pmis, *plen, *pcompact, *pts;
            // one lane per pair (perpair.hip) instead of one wavefront per pair (intraseq.hip)
            // (it stages the query in LDS: up to 4096 residues; longer ones keep the wavefront-per-pair kernel)
            const bool lanePerPair = queryLength <= 4096 && !getenv("MIOPAL_NO_PERPAIR") &&
                                     (n > kSmallSearch || getenv("MIOPAL_NO_SMALL_SEARCH"));
            RC_TRY(s.ensurePairInputs());
            PerPairArgs perPair{};
            perPair.residues = db->d_residues;
            perPair.query = s.d_query;
            perPair.queryLength = queryLength;
            perPair.matrix = s.d_matrix;
            perPair.alphabet = alphabetLength;
            perPair.gapOpen = gapOpen;
            perPair.gapExt = gapExt;
            RC_TRY(ws->get(kJobs, (size_t)n * sizeof(PairJob), &pjobs));
            RC_TRY(ws->get(kStartQ, (size_t)n * sizeof(int32_t), &psq));
            RC_TRY(ws->get(kStartT, (size_t)n * sizeof(int32_t), &pst));
            RC_TRY(ws->get(kMismatch, 3 * sizeof(int), &pmis));
            RC_TRY(ws->get(kOpsLen, (size_t)n * sizeof(int32_t), &plen));
            RC_TRY(ws->get(kRScore, (size_t)n * sizeof(int32_t), &rs));
            HIP_TRY(hipMemsetAsync(pmis, 0, 3 * sizeof(int), stream));
            if (mode != OPAL_MODE_NW) {
                RC_TRY(ws->get(kRI, (size_t)n * sizeof(int32_t), &ri));
                RC_TRY(ws->get(kRJ, (size_t)n * sizeof(int32_t), &rj));
                const DpRules rr{1, 1, 0, fr.region};
                // Which kernel scans the reversed prefixes. One strip: lane per pair (it stops at the
                // first column that holds the optimum). More strips: no early stop, every job sweeps
                // its whole prefix; a lane walks it alone (14 instructions per cell, but a wavefront
                // lasts as long as its longest lane), a wavefront per pair spends 64 lanes on it
                // (~50 instructions per 64-row column step). Rough cost of either, in ms, from the
                // mean and the longest target of the database:
                bool scanLanePerPair = lanePerPair;
                if (lanePerPair && !oneStrip) {
                    const double strips = (double)((queryLength + kLanes - 1) / kLanes);
                    const double meanLen = (double)db->total / (double)std::max<int64_t>(db->count, 1);
                    const double perLane = std::max((double)n * strips * meanLen * (14 * 4.5) / (1024 * 2.4e6),
                                                    strips * (double)db->maxLen * (64 * 14 * 4.5) / 2.4e6);
                    const double perWave = std::max((double)n * strips * (meanLen + 63) * 225.0 / (1024 * 2.4e6),
                                                    strips * (double)(db->maxLen + 63) * 250.0 / 2.4e6);
                    scanLanePerPair = perLane <= perWave;
                }
                if (scanLanePerPair) {
                    // chunks of whole wavefronts whose strip boundaries (8 B per column and pair) fit 4 GB
                    const int64_t chunk =
                        oneStrip ? n : std::max<int64_t>(kLanes, (4ll << 30) / (8 * db->maxLen) / kLanes * kLanes);
                    void *pb = nullptr, *pbins = nullptr, *psorted = nullptr;
                    if (!oneStrip) {
                        const int64_t most = std::min(chunk, n);
                        RC_TRY(ws->get(kPairB0, (size_t)((most + kLanes - 1) / kLanes * kLanes) * db->maxLen * sizeof(int2), &pb));
                        // prefixes of similar length share a wavefront (results stay addressed by job.out)
                        RC_TRY(ws->get(kSortBins, (size_t)(std::min<int64_t>(db->maxLen, 8191) + 1) * sizeof(int), &pbins));
                        RC_TRY(ws->get(kSortedJobs, (size_t)most * sizeof(PairJob), &psorted));
                    }
                    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                        const int nc = (int)std::min<int64_t>(chunk, n - c0);
                        const PairJob* jobs = (PairJob*)pjobs + c0;
                        HIP_TRY(launchReverseJobs(nc, (const int32_t*)ps + c0, (const int32_t*)pi + c0,
                                                  (const int32_t*)pj + c0, db->d_offsets + start + c0,
                                                  packRules(rr), 0, (PairJob*)pjobs + c0, stream));
                        if (!oneStrip) {
                            HIP_TRY(launchSortJobsByLength(jobs, nc, (int)db->maxLen, (int*)pbins, (PairJob*)psorted, stream));
                            jobs = (const PairJob*)psorted;
                        }
                        PerPairArgs pa = perPair;
                        pa.jobs = jobs;
                        pa.nJobs = nc;
                        pa.score = (int32_t*)rs + c0;
                        pa.endI = (int32_t*)ri + c0;
                        pa.endJ = (int32_t*)rj + c0;
                        pa.boundary = (int2*)pb;
                        pa.boundaryStride = db->maxLen;
                        HIP_TRY(launchPerPair(pa, fr.region, stream));
                    }
                } else {
                    // chunks of targets whose strip boundaries (16 B per column and pair) fit 4 GB
                    const int64_t wsStride = oneStrip ? 0 : db->maxLen;
                    const int64_t chunk = oneStrip ? n : std::max<int64_t>(1, (4ll << 30) / (16 * wsStride));
                    for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                        const int nc = (int)std::min<int64_t>(chunk, n - c0);
                        PairJob* jobs = (PairJob*)pjobs + c0;
                        HIP_TRY(launchReverseJobs(nc, (const int32_t*)ps + c0, (const int32_t*)pi + c0,
                                                  (const int32_t*)pj + c0, db->d_offsets + start + c0,
                                                  packRules(rr), wsStride, jobs, stream));
                        RC_TRY(s.runDeviceJobs(jobs, nc, (int32_t*)rs + c0, (int32_t*)ri + c0, (int32_t*)rj + c0,
                                               false, nullptr, wsStride));
                    }
                }
            }
            HIP_TRY(launchStartCells((int)n, mode, gapOpen, gapExt, (const int32_t*)ps, (const int32_t*)pi,
                                     (const int32_t*)pj, (const int32_t*)rs, (const int32_t*)ri, (const int32_t*)rj,
                                     (int32_t*)psq, (int32_t*)pst, (int*)pmis, stream));
            pt.mark("start cells (enqueued)");
            // The slots of the traceback are sized by the longest target window of the slice
            // (local alignments are short whatever the targets' lengths): one small D2H + sync.
            int checks[3] = {0, 0, 0};
            RC_TRY(ws->stageDownload(checks, pmis, sizeof checks));
            RC_TRY(ws->finishDownloads());
            if (checks[0])
                return fail(MIOPAL_ERR_INTERNAL, "reverse pass disagrees with the forward score for target %lld",
                            (long long)(start + checks[0] - 1));
            const int64_t maxWindow = std::max(checks[1], 1);
            const int64_t windowStrips = (std::max(checks[2], 1) + kLanes - 1) / kLanes;  // tallest query window
            // direction bytes of one pair (either layout: anti-diagonals of 64 lanes per strip, or
            // 64 rows per column)
            const int64_t slotDir = windowStrips * (maxWindow + kLanes - 1) * kLanes;
            const int64_t slotOps = (queryLength + maxWindow + 3) & ~(int64_t)3;  // operations of one pair
            const bool fits = n * slotOps <= (16ll << 30);  // else: host-built batches below
            if (fits) {
                RC_TRY(ws->get(kCompactOps, (size_t)(n * slotOps), &pcompact));
                // traceback in batches of whole direction slots and whole wavefronts of 64 pairs: at most
                // 4 GB of directions per batch (the lane-per-pair layout packs two cells per byte)
                auto batchFor = [&](int64_t slotBytes) {
                    return std::max<int64_t>(kLanes, std::min<int64_t>(n + kLanes - 1, kDirBudget * 2 / slotBytes) / kLanes * kLanes);
                };
                const int64_t batchLane = batchFor(std::max<int64_t>(slotDir / 2, 1)), batchWave = batchFor(slotDir);
                int64_t batch = batchLane;
                const int64_t opsCap = n * slotOps;
                const bool overlapOps = opsCap <= (512ll << 20) && !getenv("MIOPAL_NO_OPS_OVERLAP");
                void *pd, *pslots, *pbins = nullptr, *psorted = nullptr;
                // Direction pass: one lane per pair needs ~64 x fewer instructions per cell but a
                // lane walks its whole window alone (strips x columns x 64 rows, ~0.4 us per strip
                // column); with few pairs per batch or huge windows (global alignments of long
                // targets) a wavefront per pair is done sooner. Rough cost of either, in ms:
                bool traceLanePerPair = lanePerPair;
                if (lanePerPair) {
                    const double cells = (double)windowStrips * (double)(maxWindow + kLanes - 1);
                    const double pairsL = (double)std::min<int64_t>(batchLane, n), pairsW = (double)std::min<int64_t>(batchWave, n);
                    const double perLane = std::ceil((double)n / (double)batchLane) * std::ceil(pairsL / kLanes / 2048.0) *
                                           cells * (64 * 21 * 4.5) / 2.4e6;
                    const double perWave = std::ceil((double)n / (double)batchWave) *
                                           std::max(pairsW * cells * 225.0 / (1024 * 2.4e6), cells * 250.0 / 2.4e6);
                    traceLanePerPair = perLane <= perWave;
                }
                batch = traceLanePerPair ? batchLane : batchWave;
                if (overlapOps && n >= 4 * 65536)
                    batch = std::min(batch, std::max<int64_t>(65536, (n / 4 + kLanes - 1) / kLanes * kLanes));
                if (pt.on)
                    fprintf(stderr, "[miopal]   traceback: longest window %lld columns, %lld strip(s) of rows, %lld pairs per batch, %s per pair\n",
                            (long long)maxWindow, (long long)windowStrips, (long long)batch,
                            traceLanePerPair ? "lane" : "wavefront");
                const bool sortJobs = traceLanePerPair;
                // hybrid: up to 2 GB of direction bytes for the outliers a wavefront-per-pair pass takes
                int64_t maxHead = 0;
                void *phead = nullptr, *pheadDirs = nullptr;
                if (sortJobs && !getenv("MIOPAL_NO_HYBRID_TRACE")) {
                    maxHead = std::min<int64_t>((1ll << 30) / slotDir / kLanes, (batch + kLanes - 1) / kLanes);
                    if (maxHead > 0) {
                        RC_TRY(ws->get(kHeadWaves, sizeof(int), &phead));
                        if (!ws->tryGet(kHeadDirs, (size_t)(maxHead * kLanes * slotDir), &pheadDirs)) maxHead = 0;
                    }
                }
                if (sortJobs) {
                    RC_TRY(ws->get(kSortBins, (size_t)(std::min<int64_t>(maxWindow, 8191) + 1) * sizeof(int), &pbins));
                    RC_TRY(ws->get(kSortedJobs, (size_t)batch * sizeof(PairJob), &psorted));
                }
                // the direction workspace is the one allocation that can be refused on a GPU shared
                // with other work: halve the batch until it fits
                // (the lane-per-pair layout packs two rows per byte: half the bytes per pair)
                const int64_t slotDirUsed = traceLanePerPair ? slotDir / 2 : slotDir;
                while (!ws->tryGet(kDirs, (size_t)(batch * slotDirUsed), &pd)) {
                    if (batch <= kLanes) return fail(MIOPAL_ERR_HIP, "out of device memory for the traceback workspace");
                    batch = std::max<int64_t>(kLanes, batch / 2 / kLanes * kLanes);
                }
                RC_TRY(ws->get(kOps, (size_t)(batch * slotOps), &pslots));
                RC_TRY(ws->get(kTraceScore, (size_t)n * sizeof(int32_t), &pts));
                // Operations leave for the host batch by batch, on the side stream, while the next batch
                // is computed: at least four batches when there are enough pairs to fill the chip with
                // each (the copy of the last batch is all that is not hidden). They are staged in
                // pinned memory, contiguously, and copied out once the total is known.
                const int64_t nBatches = (n + batch - 1) / batch;
                struct Events {
                    std::vector<hipEvent_t> ev;
                    ~Events() {
                        for (hipEvent_t e : ev)
                            if (e) (void)hipEventDestroy(e);
                    }
                } batchDone;
                size_t opsBase = 0;
                int64_t opsFetched = 0;
                int64_t* fetchedTotal = nullptr;   // pinned: running total read back per batch
                if (overlapOps) {
                    RC_TRY(ws->ensureAux());
                    batchDone.ev.assign((size_t)nBatches, nullptr);
                    for (auto& e : batchDone.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    // room for the operations, the running totals and the small per-target arrays that
                    // follow, reserved in one go (a later drain would reset the staging buffer)
                    const size_t head = 256 + (((size_t)nBatches * 8 + 255) & ~(size_t)255);
                    RC_TRY(ws->reserveStaging(head + (size_t)opsCap + 256 + (size_t)n * 20 + 4096));
                    fetchedTotal = (int64_t*)((char*)ws->pinned + ws->pinnedUsed);
                    opsBase = ws->pinnedUsed + head;
                    ws->pinnedUsed = opsBase + (((size_t)opsCap + 255) & ~(size_t)255);
                }
                void *pblock, *ptotals;
                RC_TRY(ws->get(kOpsOff, (size_t)((batch + 255) / 256) * sizeof(int64_t), &pblock));
                RC_TRY(ws->get(kOpsTotals, (size_t)(nBatches + 1) * sizeof(int64_t), &ptotals));
                HIP_TRY(hipMemsetAsync(ptotals, 0, sizeof(int64_t), stream));
                // batch b's compacted operations: wait for its gather on the side stream, read the
                // running total, start the copy of its share
                auto fetchOps = [&](int64_t b) -> int {
                    HIP_TRY(hipStreamWaitEvent(ws->aux, batchDone.ev[(size_t)b], 0));
                    HIP_TRY(hipMemcpyAsync(fetchedTotal + b, (const int64_t*)ptotals + b + 1, sizeof(int64_t),
                                           hipMemcpyDeviceToHost, ws->aux));
                    HIP_TRY(hipStreamSynchronize(ws->aux));
                    const int64_t upTo = fetchedTotal[b];
                    if (upTo < opsFetched || upTo > opsCap) return fail(MIOPAL_ERR_INTERNAL, "bad operation count");
                    if (upTo > opsFetched)
                        HIP_TRY(hipMemcpyAsync((char*)ws->pinned + opsBase + opsFetched, (const char*)pcompact + opsFetched,
                                               (size_t)(upTo - opsFetched), hipMemcpyDeviceToHost, ws->aux));
                    opsFetched = upTo;
                    return 0;
                };
                for (int64_t b0 = 0, b = 0; b0 < n; b0 += batch, ++b) {
                    const int nb = (int)std::min<int64_t>(batch, n - b0);
                    PairJob* jobs = (PairJob*)pjobs + b0;
                    HIP_TRY(launchTraceJobs(nb, packRules(DpRules{1, 1, 0, kLastCell}), (const int32_t*)psq + b0,
                                            (const int32_t*)pst + b0, (const int32_t*)pi + b0, (const int32_t*)pj + b0,
                                            db->d_offsets + start + b0, slotDir, windowStrips > 1 ? maxWindow : 0, jobs,
                                            stream));
                    // job.out is relative to the batch: offset the score pointer
                    WalkArgs wa{};
                    if (traceLanePerPair) {
                        // neighbours of similar length share a wavefront; results stay addressed by job.out
                        if (sortJobs) {
                            HIP_TRY(launchSortJobsByLength(jobs, nb, (int)maxWindow, (int*)pbins, (PairJob*)psorted,
                                                           stream, (int*)phead, (int)maxHead));
                            jobs = (PairJob*)psorted;
                            wa.slotByOut = 1;
                        }
                        PerPairArgs pa = perPair;
                        if (maxHead > 0) {
                            // outliers at the head of the sorted list: one wavefront per pair
                            const int nh = (int)std::min<int64_t>(nb, maxHead * kLanes);
                            RC_TRY(s.runDeviceJobs(jobs, nh, (int32_t*)pts + b0, nullptr, nullptr, true,
                                                   (uint8_t*)pheadDirs, windowStrips > 1 ? maxWindow : 0,
                                                   (const int*)phead, slotDir));
                            pa.skipWaves = (const int*)phead;
                            wa.headWaves = (const int*)phead;
                            wa.headDirs = (const uint8_t*)pheadDirs;
                            wa.headDirStride = slotDir;
                        }
                        pa.jobs = jobs;
                        pa.nJobs = nb;
                        pa.dirs = (uint8_t*)pd;
                        pa.score = (int32_t*)pts + b0;  // job.out is relative to the batch
                        pa.dirWaveStride = slotDirUsed * kLanes;
                        pa.dirStripColumns = maxWindow + kLanes - 1;
                        if (windowStrips > 1) {
                            void* pb;
                            RC_TRY(ws->get(kPairB0, (size_t)((nb + kLanes - 1) / kLanes * kLanes) * maxWindow *
                                                        sizeof(int2), &pb));
                            pa.boundary = (int2*)pb;
                            pa.boundaryStride = maxWindow;
                        }
                        HIP_TRY(launchPerPair(pa, kPerPairTrace, stream));
                        wa.dirWaveStride = pa.dirWaveStride;
                        wa.dirStripColumns = pa.dirStripColumns;
                    } else {
                        RC_TRY(s.runDeviceJobs(jobs, nb, (int32_t*)pts + b0, nullptr, nullptr, true, (uint8_t*)pd,
                                               windowStrips > 1 ? maxWindow : 0));
                    }
                    wa.jobs = jobs;
                    wa.nJobs = nb;
                    wa.residues = db->d_residues;
                    wa.query = s.d_query;
                    wa.dirs = (const uint8_t*)pd;
                    wa.ops = (uint8_t*)pslots;
                    wa.opsOff = nullptr;
                    wa.opsSlot = slotOps;
                    wa.queryLength = queryLength;
                    wa.opsLen = (int32_t*)plen + b0;
                    HIP_TRY(launchWalk(wa, stream));
                    HIP_TRY(launchGatherOps(nb, (const uint8_t*)pslots, slotOps, (const int32_t*)plen + b0,
                                            (int64_t*)pblock, (const int64_t*)ptotals + b, (int64_t*)ptotals + b + 1,
                                            (uint8_t*)pcompact, stream));
                    if (overlapOps) {
                        HIP_TRY(hipEventRecord(batchDone.ev[(size_t)b], stream));
                        if (b > 0) RC_TRY(fetchOps(b - 1));   // batch b is queued behind it: the GPU stays busy
                    }
                }
                if (overlapOps) RC_TRY(fetchOps(nBatches - 1));
                pt.mark("traceback batches (enqueued)");
                // results back to the host: the small arrays first (they carry the total size),
                // the operations while the host turns lengths into offsets
                int64_t total = 0;
                // (host scratch kept on the workspace: no malloc / free of megabytes per search)
                if (ws->hostScratchA.size() < (size_t)n) ws->hostScratchA.resize((size_t)n);
                if (ws->hostScratchB.size() < (size_t)n) ws->hostScratchB.resize((size_t)n);
                int32_t* const tscore = ws->hostScratchA.data();
                int32_t* const lens = ws->hostScratchB.data();
                RC_TRY(ws->stageDownload(&total, (const int64_t*)ptotals + nBatches, sizeof(int64_t)));
                RC_TRY(ws->stageDownload(lens, plen, (size_t)n * sizeof(int32_t)));
                RC_TRY(ws->stageDownload(startQuery, psq, (size_t)n * sizeof(int32_t)));
                RC_TRY(ws->stageDownload(startTarget, pst, (size_t)n * sizeof(int32_t)));
                RC_TRY(ws->stageDownload(tscore, pts, (size_t)n * sizeof(int32_t)));
                RC_TRY(ws->finishDownloads());
                pt.mark("device pipeline + small D2H");
                if (total < 0 || total > n * slotOps) return fail(MIOPAL_ERR_INTERNAL, "bad operation count");
                if (!outOps->resize((size_t)total)) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
                std::thread copier;
                if (overlapOps) {
                    // already in pinned memory (finishDownloads above waited for the side stream too): copied
                    // out by helper threads while this one turns lengths into offsets and checks the scores
                    if (opsFetched != total) return fail(MIOPAL_ERR_INTERNAL, "operation count changed");
                    uint8_t* const dst = outOps->data;
                    const char* const src = (const char*)ws->pinned + opsBase;
                    try {
                        copier = std::thread([dst, src, total] { Workspace::copyOut(dst, src, (size_t)total); });
                    } catch (const std::exception&) {
                        Workspace::copyOut(dst, src, (size_t)total);
                    }
                } else {
                    RC_TRY(ws->stageDownload(outOps->data, pcompact, (size_t)total));
                }
                struct Joiner {
                    std::thread& t;
                    ~Joiner() { if (t.joinable()) t.join(); }
                } joinCopier{copier};
                outOff[0] = 0;
                int64_t wrongAt = -1;
                for (int64_t k = 0; k < n; ++k) {
                    outOff[k + 1] = outOff[k] + lens[(size_t)k];
                    if (endQuery[k] >= 0 && endTarget[k] >= 0 && tscore[(size_t)k] != score[k] && wrongAt < 0) wrongAt = k;
                }
                if (outOff[n] != total) return fail(MIOPAL_ERR_INTERNAL, "operation offsets disagree with the device");
                if (copier.joinable()) copier.join();
                RC_TRY(ws->finishDownloads());
                pt.mark("operations D2H");
                if (wrongAt >= 0)
                    return fail(MIOPAL_ERR_INTERNAL, "traceback score %d differs from search score %d for target %lld",
                                tscore[(size_t)wrongAt], score[wrongAt], (long long)(start + wrongAt));
                if (!flat) {
                    for (int64_t k = 0; k < n; ++k) {
                        const int64_t len = outOff[k + 1] - outOff[k];
                        alignment[k] = nullptr;
                        alignmentLength[k] = (int)len;
                        if (endQuery[k] < 0 || endTarget[k] < 0) continue;
                        unsigned char* buf = (unsigned char*)malloc((size_t)std::max<int64_t>(len, 1));
                        if (!buf) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
                        memcpy(buf, outOps->data + outOff[k], (size_t)len);
                        alignment[k] = buf;
                    }
                }
                pt.mark("host copy-out");
                return 0;
            }
        }
    }

    // ---- start locations: reversed prefixes anchored on the end cell ----------
    for (int64_t k = 0; k < n; ++k) {
        startQuery[k] = startTarget[k] = -1;
        if (flat) {
            flatOff[k + 1] = 0;  // lengths first, prefix-summed at the end
        } else {
            alignment[k] = nullptr;
            alignmentLength[k] = 0;
        }
    }
    if (flat) {
        flatOff[0] = 0;
        if (!flatOps->reserve((size_t)n * (size_t)std::min<int64_t>(queryLength + 16, 4096)))
            return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
    }
    std::vector<int64_t> live;  // slots with a non-empty alignment
    for (int64_t k = 0; k < n; ++k)
        if (endQuery[k] >= 0 && endTarget[k] >= 0) live.push_back(k);
    if (mode == OPAL_MODE_NW) {
        for (int64_t k : live) startQuery[k] = startTarget[k] = 0;
    } else if (!live.empty()) {
        const DpRules rr{1, 1, 0, fr.region};
        void *rs, *ri, *rj;
        const bool onDevice = queryLength <= kLanes;  // no strip workspace: jobs are built in HBM
        const size_t nOut = onDevice ? (size_t)n : live.size();
        RC_TRY(ws->get(kRScore, nOut * sizeof(int32_t), &rs));
        RC_TRY(ws->get(kRI, nOut * sizeof(int32_t), &ri));
        RC_TRY(ws->get(kRJ, nOut * sizeof(int32_t), &rj));
        if (onDevice) {
            void* pjobs;
            RC_TRY(ws->get(kJobs, (size_t)n * sizeof(PairJob), &pjobs));
            // pi / pj still hold the end locations of the forward pass (slice order)
            HIP_TRY(launchReverseJobs((int)n, (const int32_t*)ps, (const int32_t*)pi, (const int32_t*)pj,
                                      db->d_offsets + start, packRules(rr), 0, (PairJob*)pjobs, stream));
            RC_TRY(s.runDeviceJobs((const PairJob*)pjobs, (int)n, (int32_t*)rs, (int32_t*)ri, (int32_t*)rj));
        } else {
            std::vector<PairJob> jobs(live.size());
            for (size_t x = 0; x < live.size(); ++x) {
                const int64_t k = live[x];
                PairJob& j = jobs[x];
                j = PairJob{};
                j.tOff = db->offsets[(size_t)(start + k)] + endTarget[k];
                j.tLen = endTarget[k] + 1;
                j.tStep = -1;
                j.qOff = endQuery[k];
                j.qLen = endQuery[k] + 1;
                j.qStep = -1;
                j.rules = packRules(rr);
                j.out = (int32_t)x;
            }
            RC_TRY(s.runPairs(jobs, false, (int32_t*)rs, (int32_t*)ri, (int32_t*)rj, nullptr));
        }
        std::unique_ptr<int32_t[]> hs(new int32_t[nOut]), hi(new int32_t[nOut]), hj(new int32_t[nOut]);
        RC_TRY(ws->stageDownload(hs.get(), rs, nOut * sizeof(int32_t)));
        RC_TRY(ws->stageDownload(hi.get(), ri, nOut * sizeof(int32_t)));
        RC_TRY(ws->stageDownload(hj.get(), rj, nOut * sizeof(int32_t)));
        RC_TRY(ws->finishDownloads());
        for (size_t x = 0; x < live.size(); ++x) {
            const int64_t k = live[x];
            const size_t o = onDevice ? (size_t)k : x;
            int ri = hi[o], rj = hj[o];
            if (hs[o] != score[k] || ri < 0 || rj < 0) {
                // degenerate optimum (oracle/opal_oracle.c): one gap over one sequence only,
                // i.e. a border cell of the reversed problem; the other sequence's span is empty
                if (mode != OPAL_MODE_SW && score[k] == borderGap(endQuery[k], gapOpen, gapExt)) {
                    ri = endQuery[k];
                    rj = -1;
                } else if (mode == OPAL_MODE_OV && score[k] == borderGap(endTarget[k], gapOpen, gapExt)) {
                    ri = -1;
                    rj = endTarget[k];
                } else {
                    return fail(MIOPAL_ERR_INTERNAL, "reverse pass disagrees with the forward score for target %lld",
                                (long long)(start + k));
                }
            }
            startQuery[k] = endQuery[k] - ri;
            startTarget[k] = endTarget[k] - rj;
        }
    }

    pt.mark("start-location pass");
    // ---- traceback on [start..end] rectangles, batched by direction workspace --
    size_t pos = 0;
    while (pos < live.size()) {
        std::vector<PairJob> jobs;
        std::vector<int64_t> opsOff(1, 0);
        int64_t dirBytes = 0;
        size_t first = pos;
        while (pos < live.size()) {
            const int64_t k = live[pos];
            const int qn = endQuery[k] - startQuery[k] + 1, tn = endTarget[k] - startTarget[k] + 1;
            const int64_t need = (int64_t)((qn + kLanes - 1) / kLanes) * (tn + kLanes - 1) * kLanes;
            if (!jobs.empty() && dirBytes + need > kDirBudget) break;
            PairJob j{};
            j.tOff = db->offsets[(size_t)(start + k)] + startTarget[k];
            j.tLen = tn;
            j.tStep = 1;
            j.qOff = startQuery[k];
            j.qLen = qn;
            j.qStep = 1;
            j.rules = packRules(DpRules{1, 1, 0, kLastCell});
            j.dirOff = dirBytes;
            j.out = (int32_t)jobs.size();
            jobs.push_back(j);
            dirBytes += need;
            opsOff.push_back(opsOff.back() + qn + tn);
            ++pos;
        }
        void *pd, *po, *poff, *plen, *pscore;
        RC_TRY(ws->get(kDirs, (size_t)dirBytes, &pd));
        RC_TRY(ws->get(kOps, (size_t)opsOff.back(), &po));
        RC_TRY(ws->get(kOpsOff, opsOff.size() * sizeof(int64_t), &poff));
        RC_TRY(ws->get(kOpsLen, jobs.size() * sizeof(int32_t), &plen));
        RC_TRY(ws->get(kRScore, jobs.size() * sizeof(int32_t), &pscore));
        RC_TRY(ws->stageUpload(poff, opsOff.data(), opsOff.size() * sizeof(int64_t), stream));
        RC_TRY(s.runPairs(jobs, true, (int32_t*)pscore, nullptr, nullptr, (uint8_t*)pd));
        WalkArgs wa{};
        void* pjobs;
        RC_TRY(ws->get(kJobs, jobs.size() * sizeof(PairJob), &pjobs));
        wa.jobs = (const PairJob*)pjobs;
        wa.nJobs = (int)jobs.size();
        wa.residues = db->d_residues;
        wa.query = s.d_query;
        wa.dirs = (const uint8_t*)pd;
        wa.ops = (uint8_t*)po;
        wa.opsOff = (const int64_t*)poff;
        wa.opsLen = (int32_t*)plen;
        wa.queryLength = queryLength;
        HIP_TRY(launchWalk(wa, stream));
        if (pt.on) { HIP_TRY(hipStreamSynchronize(stream)); pt.mark("  trace + walk kernels"); }
        const size_t opsBytes = (size_t)opsOff.back();
        std::unique_ptr<uint8_t[]> ops(new uint8_t[std::max<size_t>(opsBytes, 1)]);
        std::unique_ptr<int32_t[]> lens(new int32_t[jobs.size()]), tscore(new int32_t[jobs.size()]);
        RC_TRY(ws->stageDownload(ops.get(), po, opsBytes));
        RC_TRY(ws->stageDownload(lens.get(), plen, jobs.size() * sizeof(int32_t)));
        RC_TRY(ws->stageDownload(tscore.get(), pscore, jobs.size() * sizeof(int32_t)));
        RC_TRY(ws->finishDownloads());
        pt.mark("  ops D2H");
        for (size_t x = 0; x < jobs.size(); ++x) {
            const int64_t k = live[first + x];
            if (tscore[x] != score[k])
                return fail(MIOPAL_ERR_INTERNAL, "traceback score %d differs from search score %d for target %lld",
                            tscore[x], score[k], (long long)(start + k));
            const int len = lens[x];
            const uint8_t* src = ops.get() + opsOff[x + 1] - len;
            if (flat) {
                // jobs are visited in increasing target order, so appending keeps slice order
                if (!flatOps->append(src, (size_t)len)) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
                flatOff[k + 1] = len;
            } else {
                unsigned char* buf = (unsigned char*)malloc((size_t)std::max(len, 1));
                if (!buf) return fail(MIOPAL_ERR_INTERNAL, "out of host memory");
                memcpy(buf, src, (size_t)len);
                alignment[k] = buf;
                alignmentLength[k] = len;
            }
        }
        pt.mark("  host copy-out");
    }
    if (flat)
        for (int64_t k = 0; k < n; ++k) flatOff[k + 1] += flatOff[k];
    return 0;
}

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
    if (const char* env = getenv("MIOPAL_DEVICE")) device = atoi(env);
    const bool verbose = getenv("MIOPAL_VERBOSE") != nullptr;
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
    int rc = fillHandle(h.get(), src, std::move(offsets), dbLength, alphabetLength, /*prefetchView=*/true);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc == 0)
        rc = miopalSearchResults(h.get(), query, queryLength, gapOpen, gapExt, scoreMatrix, alphabetLength, results,
                                 searchType, mode, overflowMethod, 0, dbLength);
    const auto t2 = std::chrono::steady_clock::now();
    if (rc == 0) {
        int64_t keepMb = 4096;
        if (const char* env = getenv("MIOPAL_SPARE_HANDLE_MB")) keepMb = atoll(env);
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
