// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman on
// biased integer halves, column-shifted, pair-indexed LDS profile; strips of 34..48 rows, with end
// locations (row keys in the low bits of every value).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwBiasedLocC(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairBiased<34, true>(a, rows, computeUnits, stream);
}

}  // namespace miopal
