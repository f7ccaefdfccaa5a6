// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqSignedDiagLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithI16Diag, false, true>(a, rowsPerStrip, waves, stream);
}

}  // namespace miopal
