// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqSwInt16(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithSwI16, true, false>(a, rowsPerStrip, waves, stream);
}

hipError_t launchInterseqPairSwInt16(const InterseqArgs& a, int rowsPerStrip, int computeUnits, hipStream_t stream) {
    return launchPairFlavour<ArithSwI16>(a, rowsPerStrip, computeUnits, stream);
}

}  // namespace miopal
