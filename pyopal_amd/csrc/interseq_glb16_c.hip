// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): NW / HW / OV on biased
// integer halves, column-shifted, pair-indexed LDS profile; strips of 34..48 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairGlobalC(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairGlobal<34>(a, rows, computeUnits, stream);
}

}  // namespace miopal
