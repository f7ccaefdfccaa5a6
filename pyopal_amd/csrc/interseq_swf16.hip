// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqSwHalf(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithSwF16, true, false>(a, rowsPerStrip, waves, stream);
}

hipError_t launchInterseqPairSwHalf(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    return launchPairFlavour<ArithSwF16>(a, rowsPerStrip, computeUnits, stream);
}

}  // namespace miopal
