// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): NW / HW / OV on unsigned
// anti-diagonally shifted patterns compared as half floats (ArithU16Diag).
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqUnsignedDiagLoc(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithU16Diag, false, true>(a, rowsPerStrip, waves, stream);
}

}  // namespace miopal
