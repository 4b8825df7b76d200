// LDS-tiled f64 convolution for gfx950 — placeholder until the tiled kernel lands: reports
// "unsupported" so that every product runs through the reference-order kernel.
#include "gft_kernels.hpp"

namespace gft {
bool conv_tiled_f64(hipStream_t, const double*, const double*, double*, const ConvArgs&, void*, size_t,
                    size_t* ws_needed) {
    if (ws_needed) *ws_needed = 0;
    return false;
}
}  // namespace gft
