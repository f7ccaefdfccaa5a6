// Layout kernels: build the lane-interleaved view of the database that the
// inter-sequence kernel streams, and put its results back in database order.
//
// The reference hands opalSearchDatabase N separately allocated host buffers
// (src/pyopal/lib.pxd:95-98, src/pyopal/platform/pyx.in:54-59). On the device
// the database is one linear residue array plus offsets (kept for the
// intra-sequence / traceback kernels) and, per searched slice, a packed view:
// targets sorted by length, 128 per group, stored so that a wavefront reads one
// contiguous 512-byte line per 4 database columns:
//     pack[groupOff[g] + chunk * 64 + lane] = { 4 residues of target A(lane),
//                                               4 residues of target B(lane) }
// with A(lane) = view position g*128 + lane and B(lane) = g*128 + 64 + lane.
// Positions past a target's end hold the padding symbol (alphabetLength), whose
// profile row is -32768, so padded cells can never raise a score.
#include "common.h"

namespace miopal {


static __device__ __forceinline__ uint32_t gather4(const uint8_t* base, int64_t len, int64_t col,
                                                   uint32_t pad) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t c = col + b;
        const uint32_t v = (base != nullptr && c < len) ? base[c] : pad;
        w |= v << (8 * b);
    }
    return w;
}

__global__ void pack_kernel(PackArgs a) {
    // one thread per (chunk, lane) element; chunks are numbered across groups
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t totalChunks = a.chunkPrefix[a.nGroups];
    if (e >= totalChunks * kLanes) return;
    const int64_t chunkGlobal = e / kLanes;
    const int lane = (int)(e % kLanes);
    // binary search the group of this chunk
    int lo = 0, hi = a.nGroups - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.chunkPrefix[mid] <= chunkGlobal) lo = mid; else hi = mid - 1;
    }
    const int g = lo;
    const int64_t chunk = chunkGlobal - a.chunkPrefix[g];
    const int posA = g * kGroupTargets + lane, posB = posA + kLanes;
    const uint8_t* pa = nullptr;
    const uint8_t* pb = nullptr;
    int64_t la = 0, lb = 0;
    if (posA < a.nTargets) {
        const int id = a.ids[posA];
        pa = a.residues + a.offsets[id];
        la = a.offsets[id + 1] - a.offsets[id];
        if (a.segStart) {  // a window of a long target
            pa += a.segStart[posA];
            la = a.lens[posA];
        }
    }
    if (posB < a.nTargets) {
        const int id = a.ids[posB];
        pb = a.residues + a.offsets[id];
        lb = a.offsets[id + 1] - a.offsets[id];
        if (a.segStart) {
            pb += a.segStart[posB];
            lb = a.lens[posB];
        }
    }
    const uint32_t pad = (uint32_t)a.padSymbol;
    a.pack[a.groupOff[g] + chunk * kLanes + lane] =
        make_uint2(gather4(pa, la, chunk * 4, pad), gather4(pb, lb, chunk * 4, pad));
}

// view order -> database order (relative to the slice start); counts saturated lanes
template <bool TAKE_MAX>
__global__ void scatter_kernel(const int32_t* viewScore, const uint8_t* viewOverflow,
                               const int32_t* ids, int nTargets, int64_t sliceStart,
                               int32_t* out, int32_t* overflowCount) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nTargets) return;
    // segmented views: a target's score is the maximum over its overlapping windows
    if (TAKE_MAX) atomicMax(&out[ids[k] - sliceStart], viewScore[k]);
    else out[ids[k] - sliceStart] = viewScore[k];
    if (overflowCount != nullptr && viewOverflow[k]) atomicAdd(overflowCount, 1);
}

__global__ void scatter_ends_kernel(const int32_t* viewEndI, const int32_t* viewEndJ, const int32_t* ids,
                                    int nTargets, int64_t sliceStart, int32_t* outI, int32_t* outJ) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nTargets) return;
    const int64_t slot = ids[k] - sliceStart;
    outI[slot] = viewEndI[k];
    outJ[slot] = viewEndJ[k];
}

// Segmented views with end locations: the end cell of a target is the first maximum of the
// column-major scan over the whole target = highest score, then smallest end column, then
// smallest row among its windows' own first maxima (a window holding an optimal alignment
// reports exactly the target's end cell, no window reports a cell that is not an optimum of
// the whole target). One 64-bit atomicMax per window on
//     score << 40 | (2^24 - 1 - column) << 16 | (2^16 - 1 - row)
// into zero-initialised keys, then a pass that unpacks them.
// scoreBias = 0: Smith-Waterman (scores below 1 have no location and leave the key 0); scoreBias > 0:
// HW, whose scores can be negative - every window with an end column takes part, the score field
// holds score + scoreBias > 0.
__global__ void scatter_keyed_kernel(const int32_t* viewScore, const int32_t* viewEndI, const int32_t* viewEndJ,
                                     const uint8_t* viewOverflow, const int32_t* ids, const int32_t* segStart,
                                     int nTargets, int64_t sliceStart, unsigned long long* keys,
                                     int32_t* overflowCount, int scoreBias) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nTargets) return;
    if (overflowCount != nullptr && viewOverflow[k]) atomicAdd(overflowCount, 1);
    const int score = viewScore[k] + scoreBias;
    if (score <= 0 || viewEndJ[k] < 0) return;
    const unsigned long long col = (unsigned long long)(segStart[k] + viewEndJ[k]);
    const unsigned long long row = (unsigned long long)viewEndI[k];
    const unsigned long long key =
        ((unsigned long long)score << 40) | ((0xFFFFFFull - col) << 16) | (0xFFFFull - row);
    atomicMax(&keys[ids[k] - sliceStart], key);
}

__global__ void decode_keys_kernel(const unsigned long long* keys, int n, int32_t* score, int32_t* endI,
                                   int32_t* endJ, int scoreBias) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned long long key = keys[k];
    const int s = (int)(key >> 40);
    // (an untouched key: no window reported anything - Smith-Waterman score 0; with a bias the target
    // is one the int32 kernel computes afterwards, e.g. an empty one)
    score[k] = key != 0 ? s - scoreBias : 0;
    endI[k] = key != 0 ? (int)(0xFFFFull - (key & 0xFFFFull)) : -1;
    endJ[k] = key != 0 ? (int)(0xFFFFFFull - ((key >> 16) & 0xFFFFFFull)) : -1;
}

// keys of the multi-strip pair-table kernel (interseq_impl.h): score << 40 | (0xFFFFF - column) << 20 |
// (0xFFFFF - row), merged over a group's strips with atomicMax; an untouched key is a score of 0
__global__ void decode_strip_keys_kernel(const unsigned long long* keys, int n, int32_t* score, int32_t* endI,
                                         int32_t* endJ) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned long long key = keys[k];
    score[k] = (int)(key >> 40);
    endI[k] = key != 0 ? (int)(0xFFFFFull - (key & 0xFFFFFull)) : -1;
    endJ[k] = key != 0 ? (int)(0xFFFFFull - ((key >> 20) & 0xFFFFFull)) : -1;
}

// keys of the multi-strip NW / HW / OV pair-table kernel (interseq_impl.h):
// (score + 2^31) << 32 | last-row candidate << 31 | 0x7FFFFFFF - (column | row); an untouched key belongs
// to an absent or empty target, or to lanes the int32 kernel recomputes (endI / endJ may be null)
__global__ void decode_global_keys_kernel(const unsigned long long* keys, const int32_t* lens, int n, int queryLength,
                                          int32_t* score, int32_t* endI, int32_t* endJ) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const unsigned long long key = keys[k];
    const uint32_t low = (uint32_t)key;
    const int index = (int)(0x7FFFFFFFu - (low & 0x7FFFFFFFu));
    const bool onLastRow = (low >> 31) != 0;
    score[k] = key != 0 ? (int)((uint32_t)(key >> 32) ^ 0x80000000u) : INT32_MIN;
    if (endI) endI[k] = key == 0 ? -1 : onLastRow ? queryLength - 1 : index;
    if (endJ) endJ[k] = key == 0 ? -1 : onLastRow ? index : lens[k] - 1;
}

// subset of a resident database (miopalDbCreateSubset): sequence k of the new linear database is the
// parent's residues [srcStart[k], srcStart[k] + length), one wavefront per sequence
__global__ __launch_bounds__(256) void gather_sequences_kernel(const uint8_t* src, const int64_t* srcStart,
                                                               const int64_t* dstOff, int64_t n, uint8_t* dst) {
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t from = srcStart[k], to = dstOff[k], len = dstOff[k + 1] - to;
    for (int64_t p = lane; p < len; p += 64) dst[to + p] = src[from + p];
}

// Device memory to pinned host memory beside a running search: the runtime's copy on this pool is a
// shader kernel that fills the chip (a direction kernel of 0.49 ms lasted 0.78 ms with a 0.34-ms copy
// beside it: its wavefronts take the registers of one of the two wavefronts a SIMD could hold); this one
// keeps to 64 small workgroups - PCIe needs a few hundred KB in flight, not the chip. src and dst are
// equally aligned (mod 16).
__global__ __launch_bounds__(256) void copy_out_kernel(const uint8_t* src, uint8_t* dst, int64_t bytes) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, all = (int64_t)gridDim.x * blockDim.x;
    const int64_t head = std::min<int64_t>(bytes, (16 - (int64_t)(reinterpret_cast<uintptr_t>(dst) & 15)) & 15);
    const int64_t body = (bytes - head) / 16;
    const uint4* s16 = reinterpret_cast<const uint4*>(src + head);
    uint4* d16 = reinterpret_cast<uint4*>(dst + head);
    for (int64_t k = tid; k < body; k += all) d16[k] = s16[k];
    const int64_t tail0 = head + body * 16;
    if (tid < head) dst[tid] = src[tid];
    if (tid < bytes - tail0) dst[tail0 + tid] = src[tail0 + tid];
}

hipError_t launchCopyOut(const void* src, void* dst, int64_t bytes, hipStream_t stream) {
    if (bytes <= 0) return hipSuccess;
    if (((reinterpret_cast<uintptr_t>(src) ^ reinterpret_cast<uintptr_t>(dst)) & 15) != 0)
        return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, stream);
    hipLaunchKernelGGL(copy_out_kernel, dim3(64), dim3(256), 0, stream, (const uint8_t*)src, (uint8_t*)dst, bytes);
    return hipGetLastError();
}

// The alignment operations on their way to the host, two bits each (codes 0 .. 3, include/opal.h): what a
// device-to-host copy costs the kernels beside it goes with its BYTES (profiles/r03_full_copy_overlap.txt:
// two thirds of its own duration, whoever issues it and with however few workgroups), so the operations cross
// PCIe packed - a quarter of the bytes - and the host unpacks them into the caller's buffer where it used to
// copy them (host_full.inc, unpackOps). Unit k of 64 operations (bytes [64 k, 64 k + 64) of the compacted
// stream) becomes the 16 bytes [16 k, 16 k + 16) of `dst`, operation p in bits 2 (p % 4) of byte p / 4.
__global__ __launch_bounds__(256) void copy_out_packed_kernel(const uint4* src, uint4* dst, int64_t firstUnit,
                                                              int64_t lastUnit) {
    const int64_t all = (int64_t)gridDim.x * blockDim.x;
    auto pack4 = [](uint32_t v) -> uint32_t { return ((v & 0x03030303u) * 0x01041040u) >> 24; };
    auto pack16 = [&](uint4 v) -> uint32_t { return pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24); };
    for (int64_t k = firstUnit + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < lastUnit; k += all)
        dst[k] = make_uint4(pack16(src[4 * k]), pack16(src[4 * k + 1]), pack16(src[4 * k + 2]), pack16(src[4 * k + 3]));
}

hipError_t launchCopyOutPacked(const void* src, void* dst, int64_t firstUnit, int64_t lastUnit, hipStream_t stream) {
    if (lastUnit <= firstUnit) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_out_packed_kernel, dim3(64), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, firstUnit,
                       lastUnit);
    return hipGetLastError();
}

__global__ void fill_int32_kernel(int32_t* out, int n, int32_t value) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = value;
}

hipError_t launchScatterKeyed(const int32_t* viewScore, const int32_t* viewEndI, const int32_t* viewEndJ,
                              const uint8_t* viewOverflow, const int32_t* ids, const int32_t* segStart,
                              int nTargets, int64_t sliceStart, unsigned long long* keys, int32_t* overflowCount,
                              hipStream_t stream, int scoreBias) {
    if (nTargets <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_keyed_kernel, dim3((nTargets + 255) / 256), dim3(256), 0, stream, viewScore,
                       viewEndI, viewEndJ, viewOverflow, ids, segStart, nTargets, sliceStart, keys, overflowCount,
                       scoreBias);
    return hipGetLastError();
}

hipError_t launchDecodeKeys(const unsigned long long* keys, int n, int32_t* score, int32_t* endI, int32_t* endJ,
                            hipStream_t stream, int scoreBias) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n, score, endI, endJ,
                       scoreBias);
    return hipGetLastError();
}

hipError_t launchDecodeStripKeys(const unsigned long long* keys, int n, int32_t* score, int32_t* endI, int32_t* endJ,
                                 hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_strip_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, n, score, endI, endJ);
    return hipGetLastError();
}

hipError_t launchDecodeGlobalKeys(const unsigned long long* keys, const int32_t* lens, int n, int queryLength,
                                  int32_t* score, int32_t* endI, int32_t* endJ, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_global_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, keys, lens, n, queryLength,
                       score, endI, endJ);
    return hipGetLastError();
}

hipError_t launchGatherSequences(const uint8_t* src, const int64_t* srcStart, const int64_t* dstOff, int64_t n,
                                 uint8_t* dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    if ((n + 3) / 4 > INT32_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_sequences_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, src, srcStart, dstOff, n, dst);
    return hipGetLastError();
}

hipError_t launchFillInt32(int32_t* out, int n, int32_t value, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_int32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, out, n, value);
    return hipGetLastError();
}

hipError_t launchScatterEnds(const int32_t* viewEndI, const int32_t* viewEndJ, const int32_t* ids,
                             int nTargets, int64_t sliceStart, int32_t* outI, int32_t* outJ,
                             hipStream_t stream) {
    if (nTargets <= 0) return hipSuccess;
    const int threads = 256;
    hipLaunchKernelGGL(scatter_ends_kernel, dim3((nTargets + threads - 1) / threads), dim3(threads), 0, stream,
                       viewEndI, viewEndJ, ids, nTargets, sliceStart, outI, outJ);
    return hipGetLastError();
}

hipError_t launchPack(const PackArgs& a, int64_t totalChunks, hipStream_t stream) {
    const int64_t n = totalChunks * kLanes;
    if (n <= 0) return hipSuccess;
    const int threads = 256;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launchScatter(const int32_t* viewScore, const uint8_t* viewOverflow, const int32_t* ids,
                         int nTargets, int64_t sliceStart, int32_t* out, int32_t* overflowCount,
                         bool takeMax, hipStream_t stream) {
    if (nTargets <= 0) return hipSuccess;
    const int threads = 256;
    const dim3 grid((nTargets + threads - 1) / threads);
    if (takeMax)
        hipLaunchKernelGGL(scatter_kernel<true>, grid, dim3(threads), 0, stream, viewScore, viewOverflow, ids,
                           nTargets, sliceStart, out, overflowCount);
    else
        hipLaunchKernelGGL(scatter_kernel<false>, grid, dim3(threads), 0, stream, viewScore, viewOverflow, ids,
                           nTargets, sliceStart, out, overflowCount);
    return hipGetLastError();
}

}  // namespace miopal
