// register-usage probe: single instantiations of the product GEMM kernels
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -c tools/probe/ks_probe.hip -o /tmp/p.co -Rpass-analysis=kernel-resource-usage
#define IGEMM_PROBE_ONLY
#include "../../corrifnet-correlation-aware-interactive-fusion-multimodal-learning-for-multispectral-images_amd/csrc/igemm_wgrad.hip"
template __global__ void gemm_fwd_kernel<128, 128, 2, 2, 4, false, 0, 2>(GemmArgs);
template __global__ void gemm_fwd_kernel<128, 128, 2, 2, 4, true, 1, 2>(GemmArgs);
template __global__ void gemm_fwd_kernel<128, 64, 2, 2, 4, false, 0, 2>(GemmArgs);
template __global__ void gemm_fwd_kernel<64, 64, 2, 2, 4, true, 1, 2>(GemmArgs);
template __global__ void gemm_sk_kernel<128, 128, 2, 2, 4, false, 0, 1>(GemmArgs);
template __global__ void wgrad_split_kernel<64, 128, 2, 2, false>(WgradArgs);
template __global__ void wgrad_split_kernel<128, 128, 2, 2, false>(WgradArgs);
