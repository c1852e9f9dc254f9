// fp16-split CNN kernels: the 100-wide instantiations (BASELINE's width) and the host-side entry points; the kernels themselves live in
// turboae_h2_impl.hpp, the other widths (124, 64, 32) are instantiated in turboae_h2_w.hip.
#include "turboae_h2_impl.hpp"

namespace tae {

template hipError_t launch_fused_h_u<100>(bool, const FusedParams&, int, hipStream_t);
template hipError_t launch_seg_h_u<100>(const SegParams&, int, hipStream_t);
extern template hipError_t launch_fused_h_u<124>(bool, const FusedParams&, int, hipStream_t);
extern template hipError_t launch_fused_h_u<64>(bool, const FusedParams&, int, hipStream_t);
extern template hipError_t launch_fused_h_u<32>(bool, const FusedParams&, int, hipStream_t);
extern template hipError_t launch_seg_h_u<124>(const SegParams&, int, hipStream_t);
extern template hipError_t launch_seg_h_u<64>(const SegParams&, int, hipStream_t);
extern template hipError_t launch_seg_h_u<32>(const SegParams&, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
hipError_t launch_fused_h(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 124: return launch_fused_h_u<124>(decoder, P, grid, st);
        case 100: return launch_fused_h_u<100>(decoder, P, grid, st);
        case 64: return launch_fused_h_u<64>(decoder, P, grid, st);
        case 32: return launch_fused_h_u<32>(decoder, P, grid, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_seg_h(int U, const SegParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 124: return launch_seg_h_u<124>(P, grid, st);
        case 100: return launch_seg_h_u<100>(P, grid, st);
        case 64: return launch_seg_h_u<64>(P, grid, st);
        case 32: return launch_seg_h_u<32>(P, grid, st);
        default: return hipErrorInvalidValue;
    }
}

int seg_lds_bytes_h(int U, int T, int n_layer, int taps) {
    const int range_layers = n_layer;
    const int pad = taps / 2, rows = T + 2 * pad * n_layer + 3 + 2 * pad;
    size_t b = 2 * (size_t)(rows + 2) * U * 2 + 2 * (size_t)(rows + 1 + kXSlack) * kXRowB + (size_t)kHeadSlots * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kHeadSlots * 8 * 4 + kRangeHeaderB + (size_t)range_layers * kRangeLayerB;     // head-combine scratch + range bookkeeping
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

int seg_lds_bytes_h_dense(int U, int T, int n_layer) {
    const int range_layers = n_layer;
    const int rows = T + 4 * n_layer + 3 + 4;
    const int npanel = n_layer > 1 ? n_layer - 1 : 1;
    size_t b = (size_t)npanel * 2 * (size_t)(rows + 2) * U * 2 + 2 * (size_t)(rows + 1 + kXSlack) * kXRowB + (size_t)kHeadSlots * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kHeadSlots * 8 * 4 + kRangeHeaderB + (size_t)range_layers * kRangeLayerB;     // head-combine scratch + range bookkeeping
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

int fused_lds_bytes_h(int U, int L, int nb, int taps, int range_layers) {
    const int pad = taps / 2, rows = nb * (L + pad) + pad;
    size_t b = 2 * (size_t)(rows + 2) * U * 2 + 4 * (size_t)(rows + 1 + kXSlack) * kXRowB + 2 * (size_t)L * 4 + (size_t)kHeadSlots * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kHeadSlots * 8 * 4 + kRangeHeaderB + (size_t)range_layers * kRangeLayerB;     // head-combine scratch + range bookkeeping
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

}  // namespace tae
