// Dispatcher of the inter-sequence kernel flavours (kernels: interseq_impl.h).
#include "common.h"

namespace miopal {

hipError_t launchInterseq(const InterseqArgs& a, int rowsPerStrip, int waves, InterseqFlavour flavour,
                          bool locate, hipStream_t stream) {
    if (a.nGroups <= 0) return hipSuccess;
    if (locate) {
        switch (flavour) {
            case kSwHalf: return launchInterseqSwHalfLoc(a, rowsPerStrip, waves, stream);
            case kSwInt16: return launchInterseqSwInt16Loc(a, rowsPerStrip, waves, stream);
            case kSignedInt16: return launchInterseqSignedLoc(a, rowsPerStrip, waves, stream);
            case kSignedInt16AllCells: return launchInterseqSignedAllLoc(a, rowsPerStrip, waves, stream);
            case kSignedInt16Diag: return launchInterseqSignedDiagLoc(a, rowsPerStrip, waves, stream);
            case kUnsignedDiag: return launchInterseqUnsignedDiagLoc(a, rowsPerStrip, waves, stream);
            case kSwShifted: return hipErrorInvalidValue;  // scores only
        }
        return hipErrorInvalidValue;
    }
    switch (flavour) {
        case kSwHalf: return launchInterseqSwHalf(a, rowsPerStrip, waves, stream);
        case kSwInt16: return launchInterseqSwInt16(a, rowsPerStrip, waves, stream);
        case kSignedInt16: return launchInterseqSigned(a, rowsPerStrip, waves, stream);
        case kSignedInt16AllCells: return launchInterseqSignedAll(a, rowsPerStrip, waves, stream);
        case kSignedInt16Diag: return launchInterseqSignedDiag(a, rowsPerStrip, waves, stream);
        case kUnsignedDiag: return launchInterseqUnsignedDiag(a, rowsPerStrip, waves, stream);
        case kSwShifted: return launchInterseqSwShifted(a, rowsPerStrip, waves, stream);
    }
    return hipErrorInvalidValue;
}

bool interseqPairFits(int rowsPerStrip, int nSymbols) {
    const size_t bytes = (size_t)nSymbols * nSymbols * (size_t)(((rowsPerStrip + 3) / 4) | 1) * 16;
    return bytes <= 158 * 1024;  // leaves the runtime a little of the 160 KB
}

hipError_t launchInterseqPair(const InterseqArgs& a, int rowsPerStrip, PairFlavour flavour, int computeUnits,
                              hipStream_t stream, bool locate) {
    if (a.nGroups <= 0) return hipSuccess;
    if (locate && flavour != kPairSwBiased && flavour != kPairGlobalBiased && flavour != kPairSwStrips && flavour != kPairGlobalStrips)
        return hipErrorInvalidValue;
    if (a.nStrips != 1 && flavour != kPairSwStrips && flavour != kPairGlobalStrips) return hipErrorInvalidValue;
    switch (flavour) {
        case kPairGlobalBiased:
            // (end locations are a run-time option of this kernel: a.endI != nullptr)
            if (rowsPerStrip < 2 || rowsPerStrip > 64 || (rowsPerStrip & 1)) return hipErrorInvalidValue;
            if (rowsPerStrip < 18) return launchInterseqPairGlobalA(a, rowsPerStrip, computeUnits, stream);
            if (rowsPerStrip < 34) return launchInterseqPairGlobalB(a, rowsPerStrip, computeUnits, stream);
            if (rowsPerStrip < 50) return launchInterseqPairGlobalC(a, rowsPerStrip, computeUnits, stream);
            return launchInterseqPairGlobalD(a, rowsPerStrip, computeUnits, stream);
        case kPairSwBiased:
            // any even number of rows
            if (rowsPerStrip < 2 || rowsPerStrip > 64 || (rowsPerStrip & 1)) return hipErrorInvalidValue;
            if (locate) {
                if (rowsPerStrip < 18) return launchInterseqPairSwBiasedLocA(a, rowsPerStrip, computeUnits, stream);
                if (rowsPerStrip < 34) return launchInterseqPairSwBiasedLocB(a, rowsPerStrip, computeUnits, stream);
                if (rowsPerStrip < 50) return launchInterseqPairSwBiasedLocC(a, rowsPerStrip, computeUnits, stream);
                return launchInterseqPairSwBiasedLocD(a, rowsPerStrip, computeUnits, stream);
            }
            if (rowsPerStrip < 18) return launchInterseqPairSwBiasedA(a, rowsPerStrip, computeUnits, stream);
            if (rowsPerStrip < 34) return launchInterseqPairSwBiasedB(a, rowsPerStrip, computeUnits, stream);
            if (rowsPerStrip < 50) return launchInterseqPairSwBiasedC(a, rowsPerStrip, computeUnits, stream);
            return launchInterseqPairSwBiasedD(a, rowsPerStrip, computeUnits, stream);
        case kPairSwStrips:
            if (rowsPerStrip < 32 || rowsPerStrip > (locate ? kPairStripsMaxRowsLoc : kPairStripsMaxRows) || (rowsPerStrip & 1))
                return hipErrorInvalidValue;
            if (a.known) {
                // (second pass of an `end` search: the known optimum is looked for, no row keys)
                if (locate || rowsPerStrip > kPairStripsMaxRowsKnown) return hipErrorInvalidValue;
                return launchInterseqPairSwStripsKnownA(a, rowsPerStrip, computeUnits, stream);
            }
            if (locate) {
                if (rowsPerStrip < 48) return launchInterseqPairSwStripsLocA(a, rowsPerStrip, computeUnits, stream);
                return launchInterseqPairSwStripsLocB(a, rowsPerStrip, computeUnits, stream);
            }
            if (rowsPerStrip < 48) return launchInterseqPairSwStripsA(a, rowsPerStrip, computeUnits, stream);
            return launchInterseqPairSwStripsB(a, rowsPerStrip, computeUnits, stream);
        case kPairGlobalStrips:
            if (rowsPerStrip < 32 || rowsPerStrip > (locate ? kPairStripsMaxRowsLoc : kPairStripsMaxRows) || (rowsPerStrip & 1))
                return hipErrorInvalidValue;
            if (locate) {
                // (end locations leave as keys: decode_global_keys_kernel)
                if (rowsPerStrip < 48) return launchInterseqPairGlobalStripsLocA(a, rowsPerStrip, computeUnits, stream);
                return launchInterseqPairGlobalStripsLocB(a, rowsPerStrip, computeUnits, stream);
            }
            if (rowsPerStrip < 48) return launchInterseqPairGlobalStripsA(a, rowsPerStrip, computeUnits, stream);
            return launchInterseqPairGlobalStripsB(a, rowsPerStrip, computeUnits, stream);
        case kPairSwHalf: return launchInterseqPairSwHalf(a, rowsPerStrip, computeUnits, stream);
        case kPairSwInt16: return launchInterseqPairSwInt16(a, rowsPerStrip, computeUnits, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace miopal
