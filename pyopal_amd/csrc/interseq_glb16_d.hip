// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): NW / HW / OV on biased
// integer halves, column-shifted, pair-indexed LDS profile; strips of 50..64 rows.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqPairGlobalD(const InterseqArgs& a, int rows, int computeUnits, hipStream_t stream) {
    return launchPairGlobal<50>(a, rows, computeUnits, stream);
}

}  // namespace miopal
