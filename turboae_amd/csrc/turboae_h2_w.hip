// fp16-split CNN kernels (turboae_h2_impl.hpp) instantiated for the channel widths other than 100: 124 (widths 101..124), 64, 32.
#include "turboae_h2_impl.hpp"

namespace tae {

template hipError_t launch_fused_h_u<124>(bool, const FusedParams&, int, hipStream_t);
template hipError_t launch_fused_h_u<64>(bool, const FusedParams&, int, hipStream_t);
template hipError_t launch_fused_h_u<32>(bool, const FusedParams&, int, hipStream_t);
template hipError_t launch_seg_h_u<124>(const SegParams&, int, hipStream_t);
template hipError_t launch_seg_h_u<64>(const SegParams&, int, hipStream_t);
template hipError_t launch_seg_h_u<32>(const SegParams&, int, hipStream_t);

}  // namespace tae
