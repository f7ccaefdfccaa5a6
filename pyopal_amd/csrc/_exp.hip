#include "interseq_impl.h"
namespace miopal {
hipError_t expLaunch(const InterseqArgs& a, int cu, hipStream_t s) { return EXP_FN<EXP_R, EXP_LOC>(a, cu, s); }
}
