// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman on
// biased integer halves, column-shifted, with the pair-indexed LDS profile.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwBiased(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    return launchPairBiased<0>(a, rowsPerStrip, computeUnits, stream);
}

}  // namespace miopal
