// One arithmetic flavour of the inter-sequence kernel, with end locations (interseq_impl.h).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqSwHalfLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithSwF16, true, true>(a, rowsPerStrip, waves, stream);
}

}  // namespace miopal
