// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman on
// biased integer halves, column-shifted, pair-indexed LDS profile; strips of 2..16 rows, scores only.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwBiasedA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairBiased<2, false>(a, rows, computeUnits, stream);
}

}  // namespace miopal
