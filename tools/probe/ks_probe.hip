// register-usage probe: single instantiations of the product GEMM kernel
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c tools/probe/ks_probe.hip -o /dev/null -Rpass-analysis=kernel-resource-usage
#define IGEMM_PROBE_ONLY
#include "../../corrifnet-correlation-aware-interactive-fusion-multimodal-learning-for-multispectral-images_amd/csrc/igemm.hip"
template __global__ void gemm_fwd_kernel<128, 128, 2, 2, 4, false, 0, true>(GemmArgs);
template __global__ void gemm_fwd_kernel<128, 128, 2, 2, 4, false, 0, false>(GemmArgs);
template __global__ void gemm_fwd_kernel<128, 128, 2, 2, 4, true, 1, true>(GemmArgs);
