// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman on
// biased integer halves, column-shifted, pair-indexed LDS profile; strips of 18..32 rows, scores only.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwBiasedB(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairBiased<18, false>(a, rows, computeUnits, stream);
}

}  // namespace miopal
