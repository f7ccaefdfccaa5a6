// One arithmetic flavour of the inter-sequence kernel (see interseq_impl.h): Smith-Waterman scores on
// column-shifted unsigned patterns compared as half floats (ArithSwU16), any number of strips.
#include "interseq_impl.h"

namespace miopal {

hipError_t launchInterseqSwShifted(const InterseqArgs& a, int rowsPerStrip, int waves, hipStream_t stream) {
    return launchFlavour<ArithSwU16, true, false>(a, rowsPerStrip, waves, stream);
}

}  // namespace miopal
