// k_blur3_dma<double, ..., EPI 3 / 4>: the two halves of a Lanczos step inside the one-pass
// blur (nsol_blur3_dma.hpp), in a translation unit of their own: the instantiations of
// that kernel are most of the library's compile time
#define NSOL_BLUR3_DMA_IMPL
#include "nsol_blur3_dma.hpp"

namespace nsol_blur3 {
NSOL_B3L_DEF(double)
}  // namespace nsol_blur3
