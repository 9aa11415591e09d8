// One wave samples the shader cycle counter against the 100 MHz wall counter once per millisecond:
// the ratio is the shader clock the chip is actually running at while other streams are busy.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void clock_probe_kernel(uint64_t* out, int nsamples, int ticks_per_sample) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < nsamples; ++i) {
        uint64_t w0 = wall_clock64();
        out[2 * i] = w0;
        out[2 * i + 1] = clock64();
        for (int spin = 0; spin < 200000; ++spin) {          // bounded: at most ~200000 sleeps per sample
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - w0 >= (uint64_t)ticks_per_sample) break;
        }
    }
}
extern "C" int clock_probe(uint64_t* out, int nsamples, int ticks_per_sample, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, nsamples, ticks_per_sample);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
