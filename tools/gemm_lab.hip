// gemm_lab: ablation / variant bench of the fp32-MFMA GEMM main loop (diagnostic tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o gpurun_out/gemm_lab && gpurun_out/gemm_lab
// C[M,N] = A[M,K] . B[N,K]^T, 128x128 tile, 4 waves (2x2 of 64x64), v_mfma_f32_32x32x2_f32 - the product kernel's structure
// (csrc/igemm.hip gemm_fwd_kernel<128,128,2,2,4,true,0>) with pieces removed or changed, to see where the MFMA pipe idles.
// Ablated variants compute garbage on purpose; only V0/V4/V6/V7/V8 are checked against a reference element.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    uint32_t q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

enum { V_BASE = 0, V_NOGLOBAL = 1, V_NOSTORE = 2, V_MFMAONLY = 3, V_BK64 = 4, V_PRIVATE = 6, V_DBUF = 7 };

// ---------------------------------------------------------------------------------------------------------------
// shared-tile kernel (V0..V4, V7).  BKT = K depth per LDS tile.
template <int V, int BKT, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int M, int N, int K) {
    constexpr int BM = 128, BN = 128, PITCH = BKT + 4;
    constexpr int NBUF = (V == V_DBUF) ? 2 : 1;
    constexpr int CPR = BKT / 4;                    // float4 chunks per row
    constexpr int RPT = 256 / CPR;                  // rows covered per pass
    constexpr int AI = BM / RPT;                    // chunks per thread per operand
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tiles_n = N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kc = tid % CPR, ar = tid / CPR;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[AI], rb[AI];
    const int nk = K / BKT;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar + RPT * i) * K + kt * BKT + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar + RPT * i) * K + kt * BKT + kc * 4);
        }
    };
    auto store_tile = [&](int buf) {
        float* As = lds + buf * (BM + BN) * PITCH;
        float* Bs = As + BM * PITCH;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            *reinterpret_cast<f32x4*>(&As[(ar + RPT * i) * PITCH + kc * 4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(ar + RPT * i) * PITCH + kc * 4]) = rb[i];
        }
    };
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    auto compute = [&](int buf) {
        const float* As = lds + buf * (BM + BN) * PITCH;
        const float* Bs = As + BM * PITCH;
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * 2 + i) * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * 2 + j) * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (V == V_MFMAONLY) {
        float a0 = A[tid], b0 = B[tid];
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int s = 0; s < BKT / 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i][j], 0, 0, 0);
            asm volatile("" : "+v"(a0), "+v"(b0));
        }
    } else if constexpr (V == V_DBUF) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            compute(kt & 1);
            if (kt + 1 < nk) store_tile((kt + 1) & 1);     // other buffer: last read one iteration ago, fenced by the barrier below
            __syncthreads();
        }
    } else {
        load_tile(0);
        if (V == V_NOSTORE) { store_tile(0); __syncthreads(); }
        for (int kt = 0; kt < nk; ++kt) {
            if (V != V_NOSTORE) { store_tile(0); __syncthreads(); }
            if (V != V_NOGLOBAL && kt + 1 < nk) load_tile(kt + 1);
            if (V == V_NOGLOBAL) { asm volatile("" : "+v"(ra[0]), "+v"(rb[0])); }
            compute(0);
            if (V != V_NOSTORE) __syncthreads();
        }
    }
    // simple epilogue (scalar stores)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// V6: every wave stages ITS OWN 64 rows of A and 64 rows of B in a private LDS region: no workgroup barrier at all
// (LDS operations of one wave complete in order).  2x the global->LDS traffic of the shared tile (each panel is staged by
// the two waves that use it), which the f32 MFMA rate can afford: 16 KB per wave per 4096 matrix cycles.
template <int BKT, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_private_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                 float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 128, BN = 128, PITCH = BKT + 4;
    constexpr int CPR = BKT / 4;                    // float4 chunks per row
    constexpr int RPP = 64 / CPR;                   // rows per pass of one wave
    constexpr int NI = 64 / RPP;                    // passes per operand
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tiles_n = N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM + wm * 64, n0 = (tile % tiles_n) * BN + wn * 64;
    float* As = lds + wave * 128 * PITCH;
    float* Bs = As + 64 * PITCH;
    const int kc = lane % CPR, ar = lane / CPR;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[NI], rb[NI];
    const int nk = K / BKT;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar + RPP * i) * K + kt * BKT + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar + RPP * i) * K + kt * BKT + kc * 4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            *reinterpret_cast<f32x4*>(&As[(ar + RPP * i) * PITCH + kc * 4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(ar + RPP * i) * PITCH + kc * 4]) = rb[i];
        }
    };
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[(i * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[(j * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
    }
}

__global__ void fill_kernel(float* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xFFFF) - 32768) * (1.0f / 32768.0f);
    }
}

struct Shape { int M, N, K; };

template <typename F>
static float time_it(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

static double ref_elem(const std::vector<float>& A, const std::vector<float>& B, int K, int r, int c) {
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)A[(int64_t)r * K + k] * B[(int64_t)c * K + k];
    return s;
}

int main() {
    Shape shapes[] = {{4096, 4096, 4096}, {25088, 256, 2304}, {6272, 512, 4608}, {100352, 128, 1152}, {25088, 1024, 256}};
    for (auto& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        float *A, *B, *C;
        CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        fill_kernel<<<2048, 256>>>(A, (int64_t)M * K, 1u);
        fill_kernel<<<2048, 256>>>(B, (int64_t)N * K, 7u);
        CK(hipDeviceSynchronize());
        std::vector<float> hA((size_t)64 * K), hB((size_t)N * K);
        CK(hipMemcpy(hA.data(), A, (size_t)64 * K * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hB.data(), B, (size_t)N * K * 4, hipMemcpyDeviceToHost));
        const double ref = ref_elem(hA, hB, K, 37, 101);
        const unsigned tiles = (unsigned)(M / 128) * (N / 128);
        const double fl = 2.0 * M * N * K;
        printf("shape M %d N %d K %d  tiles %u\n", M, N, K, tiles);
        auto report = [&](const char* name, float ms, bool check) {
            float got = 0;
            CK(hipMemcpy(&got, C + (int64_t)37 * N + 101, 4, hipMemcpyDeviceToHost));
            printf("  %-44s %8.3f ms  %6.1f TF/s  %s\n", name, ms, fl / ms / 1e9, check ? (fabs(got - ref) < 1e-3 * (fabs(ref) + 1) ? "ok" : "WRONG") : "-");
            fflush(stdout);
        };
#define RUN(NAME, KERNEL, LDSB, CHECK)                                                                   \
    {                                                                                                    \
        CK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
        CK(hipMemset(C, 0, (size_t)M * N * 4));                                                          \
        float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(tiles), dim3(256), LDSB, 0, A, B, C, M, N, K); }, 10); \
        CK(hipGetLastError());                                                                           \
        report(NAME, ms, CHECK);                                                                         \
    }
        RUN("V0 baseline BK32 (2 barriers / tile)", (gemm_kernel<V_BASE, 32, 1>), 256 * 36 * 4, true);
        RUN("V0 baseline, launch_bounds(256,2)", (gemm_kernel<V_BASE, 32, 2>), 256 * 36 * 4, true);
        RUN("V0 baseline, launch_bounds(256,3)", (gemm_kernel<V_BASE, 32, 3>), 256 * 36 * 4, true);
        RUN("V1 no global loads in the loop", (gemm_kernel<V_NOGLOBAL, 32, 1>), 256 * 36 * 4, false);
        RUN("V2 no LDS stores / barriers in the loop", (gemm_kernel<V_NOSTORE, 32, 1>), 256 * 36 * 4, false);
        RUN("V3 MFMA only", (gemm_kernel<V_MFMAONLY, 32, 1>), 256 * 36 * 4, false);
        RUN("V4 BK64", (gemm_kernel<V_BK64, 64, 1>), 256 * 68 * 4, true);
        RUN("V4 BK64, launch_bounds(256,2)", (gemm_kernel<V_BK64, 64, 2>), 256 * 68 * 4, true);
        RUN("V7 double-buffered LDS, 1 barrier / tile", (gemm_kernel<V_DBUF, 32, 1>), 2 * 256 * 36 * 4, true);
        RUN("V7 double-buffered, launch_bounds(256,2)", (gemm_kernel<V_DBUF, 32, 2>), 2 * 256 * 36 * 4, true);
        RUN("V7 double-buffered BK64", (gemm_kernel<V_DBUF, 64, 1>), 2 * 256 * 68 * 4, true);
        RUN("V6 wave-private staging BK32, no barrier", (gemm_private_kernel<32, 1>), 4 * 128 * 36 * 4, true);
        RUN("V6 wave-private BK32, launch_bounds(256,2)", (gemm_private_kernel<32, 2>), 4 * 128 * 36 * 4, true);
        RUN("V6 wave-private staging BK64", (gemm_private_kernel<64, 1>), 4 * 128 * 68 * 4, true);
        RUN("V6 wave-private BK16", (gemm_private_kernel<16, 1>), 4 * 128 * 20 * 4, true);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    return 0;
}
