// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): NW / HW / OV of several
// strips on biased integer halves with the pair-indexed LDS profile, with end locations; strips of 48 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairGlobalStripsLocB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairGlobalStrips<48, true>(a, rows, computeUnits, stream);
}

}  // namespace miopal
