// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman scores
// with end locations of several strips on biased integer halves with the pair-indexed LDS profile;
// strips of 32..46 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwStripsLocA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairStrips<32, true>(a, rows, computeUnits, stream);
}

}  // namespace miopal
