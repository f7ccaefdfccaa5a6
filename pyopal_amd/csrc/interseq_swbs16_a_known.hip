// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): second pass of a Smith-Waterman
// `end` search of several strips whose scores are beyond the row keys' range - the first cell that holds each
// target's known optimum; strips of 32..40 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairSwStripsKnownA(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairStrips<32, false, true>(a, rows, computeUnits, stream);
}

}  // namespace miopal
