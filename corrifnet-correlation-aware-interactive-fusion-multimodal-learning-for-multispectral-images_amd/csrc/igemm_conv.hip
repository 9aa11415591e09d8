// The implicit-convolution half of the gemm_fwd family (GEMM = false: forward 1x3x3 / 3x3x3 convolutions and their data gradients on the
// float4 loader), instantiated in its own translation unit so that it compiles in parallel with the plain-GEMM half (igemm.hip).
#include "igemm_fwd.h"

#define CORRIF_CONV_VARIANT(BM, BN, WM, WN) \
    template int launch_variant<BM, BN, WM, WN, 4, false, 0>(GemmArgs&, int, hipStream_t, bool, size_t*); \
    template int launch_variant<BM, BN, WM, WN, 4, false, 1>(GemmArgs&, int, hipStream_t, bool, size_t*);
CORRIF_CONV_VARIANT(128, 128, 2, 2)
CORRIF_CONV_VARIANT(128, 64, 2, 2)
CORRIF_CONV_VARIANT(64, 64, 2, 2)
CORRIF_CONV_VARIANT(256, 32, 4, 1)
#undef CORRIF_CONV_VARIANT
