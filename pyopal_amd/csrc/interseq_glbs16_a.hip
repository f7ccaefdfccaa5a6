// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): NW / HW / OV of several
// strips on biased integer halves with the pair-indexed LDS profile; strips of 32..46 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairGlobalStripsA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairGlobalStrips<32, false>(a, rows, computeUnits, stream);
}

}  // namespace miopal
