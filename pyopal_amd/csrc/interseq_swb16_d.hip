// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman on
// biased integer halves, column-shifted, pair-indexed LDS profile; strips of 50..64 rows, scores only.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwBiasedD(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairBiased<50, false>(a, rows, computeUnits, stream);
}

}  // namespace miopal
