// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman scores
// with end locations of several strips on biased integer halves with the pair-indexed LDS profile;
// strips of 48 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwStripsLocB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairStrips<48, true>(a, rows, computeUnits, stream);
}

}  // namespace miopal
