// Dispatcher of the inter-sequence kernel flavours (kernels: interseq_impl.h).
#include "common.h"

namespace miopal {

hipError_t launchInterseq(const InterseqArgs& a, int rowsPerStrip, int waves, InterseqFlavour flavour,
                          hipStream_t stream) {
    if (a.nGroups <= 0) return hipSuccess;
    switch (flavour) {
        case kSwHalf: return launchInterseqSwHalf(a, rowsPerStrip, waves, stream);
        case kSwInt16: return launchInterseqSwInt16(a, rowsPerStrip, waves, stream);
        case kSignedInt16: return launchInterseqSigned(a, rowsPerStrip, waves, stream);
        case kSignedInt16AllCells: return launchInterseqSignedAll(a, rowsPerStrip, waves, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace miopal
