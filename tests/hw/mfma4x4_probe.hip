// Probe of v_mfma_f32_4x4x1_16b_f32 operand maps and the CBSZ/ABID A-broadcast on gfx950 (exact integer data).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void probe(const float* a, const float* b, float* d) {
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = acc[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }       // A[b][i] = 1 + 4b + i ; B[b][j] = 100 + 4b + j
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    int bad0 = 0, bad1 = 0, bad2 = 0;
    probe<0, 0><<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {      // expect D[vgpr r][lane 4b+j] = A[b][r] * B[b][j]
        int bb = l >> 2; float e = (1 + 4 * bb + r) * (100.0f + l);
        if (hd[r * 64 + l] != e) ++bad0;
    }
    probe<4, 5><<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {      // expect A taken from block 5 for every block
        float e = (1 + 4 * 5 + r) * (100.0f + l);
        if (hd[r * 64 + l] != e) ++bad1;
    }
    probe<4, 15><<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        float e = (1 + 4 * 15 + r) * (100.0f + l);
        if (hd[r * 64 + l] != e) ++bad2;
    }
    printf("plain map mismatches %d ; cbsz=4 abid=5 mismatches %d ; cbsz=4 abid=15 mismatches %d\n", bad0, bad1, bad2);
    if (bad1) { printf("sample cbsz4/abid15 lane0: %g %g %g %g  lane 63: %g\n", hd[0], hd[64], hd[128], hd[192], hd[63]); }
    return (bad0 || bad1 || bad2) ? 1 : 0;
}
