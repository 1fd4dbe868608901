// Probe: are v_mfma_f32_32x32x2_f32, 16x16x4 and 4x4x1 bit-identical to a sequential fmaf chain over k ascending?
// Measured on MI355X (gfx950): yes, 0 differing outputs for all three at K = 336.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f32_order.hip -o gpurun_out/mfma_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma32(const float* A, const float* B, float* C, int K) {      // A [32][K], B [K][32], C [32][32]
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + lhi], B[(k + lhi) * 32 + l31], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi; C[row * 32 + l31] = acc[r]; }
}
// 4x4x1, 16 blocks: lane = 4 * block + i: A operand a[i][block], B operand b[block][j = i] -> D[block][i'][j]: vgpr r = row i', lane = 4 block + j
__global__ void k_mfma4(const float* A, const float* B, float* C, int K) {       // rows 0..3 of A against all 32 columns of B (blocks 0..7), C4 [4][32]
    const int lane = threadIdx.x, blk = lane >> 2, i = lane & 3;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
        const float a = A[i * K + k];                                   // row i (same for every block)
        const float b = blk < 8 ? B[k * 32 + blk * 4 + i] : 0.0f;       // column 4 blk + i
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    }
    if (blk < 8) for (int r = 0; r < 4; ++r) C[r * 32 + blk * 4 + i] = acc[r];
}
// 16x16x4: A lane (row l & 15, k = l >> 4), B lane (k = l >> 4, col l & 15), D reg r: row 4 (l >> 4) + r, col l & 15
__global__ void k_mfma16(const float* A, const float* B, float* C, int K) {      // rows 0..15 of A, columns 0..15 of B -> C16 [16][16]
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[l15 * K + k + lq], B[(k + lq) * 32 + l15], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * lq + r) * 16 + l15] = acc[r];
}
__global__ void k_fma(const float* A, const float* B, float* C, int K, int mode) {
    const int row = threadIdx.x / 32, col = threadIdx.x % 32;
    for (int rr = row; rr < 32; rr += blockDim.x / 32) {
        float acc = 0.0f;
        if (mode == 0) for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[rr * K + k], B[k * 32 + col], acc);
        else if (mode == 1) for (int k = 0; k < K; ++k) acc = acc + A[rr * K + k] * B[k * 32 + col];      // -ffp-contract=off: mul, then add
        else for (int k = 0; k < K; k += 2) { float p = __builtin_fmaf(A[rr * K + k + 1], B[(k + 1) * 32 + col], A[rr * K + k] * B[k * 32 + col]); acc = acc + p; }
        C[rr * 32 + col] = acc;
    }
}
int main() {
    const int K = 336;
    std::vector<float> A(32 * K), B(K * 32), C0(1024), C1(1024), C2(1024), C3(1024), C4(128);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.0f - 0.3f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); hipMemcpy(C0.data(), dC, 4096, hipMemcpyDeviceToHost);
    for (int m = 0; m < 3; ++m) {
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(256), 0, 0, dA, dB, dC, K, m);
        hipMemcpy((m == 0 ? C1 : m == 1 ? C2 : C3).data(), dC, 4096, hipMemcpyDeviceToHost);
    }
    hipLaunchKernelGGL(k_mfma4, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); hipMemcpy(C4.data(), dC, 512, hipMemcpyDeviceToHost);
    std::vector<float> C16(256);
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K); hipMemcpy(C16.data(), dC, 1024, hipMemcpyDeviceToHost);
    int d6 = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) d6 += memcmp(&C16[r * 16 + c], &C1[r * 32 + c], 4) != 0;
    printf("mfma16x16x4 vs fmaf-chain: %d differ (of 256)\n", d6);
    int d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
    for (int i = 0; i < 1024; ++i) { d1 += memcmp(&C0[i], &C1[i], 4) != 0; d2 += memcmp(&C0[i], &C2[i], 4) != 0; d3 += memcmp(&C0[i], &C3[i], 4) != 0; }
    for (int i = 0; i < 128; ++i) { d4 += memcmp(&C4[i], &C0[i], 4) != 0; d5 += memcmp(&C4[i], &C1[i], 4) != 0; }
    printf("mfma32x32x2 vs fmaf-chain: %d differ; vs mul+add chain: %d; vs pairwise(fma(a1 b1, a0 b0)) + acc: %d   (of 1024)\n", d1, d2, d3);
    printf("mfma4x4x1 vs mfma32x32x2: %d differ; vs fmaf-chain: %d   (of 128)\n", d4, d5);
    printf("sample: %.9g %.9g %.9g %.9g %.9g\n", C0[5], C1[5], C2[5], C3[5], C4[5]);
    return 0;
}
