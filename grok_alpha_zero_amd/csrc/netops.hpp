// netops.hpp — the small layers around the MFMA trunk convolution: stem convolutions on the int8 input planes, direct
// convolutions with few channels (head convs, TicTacToe's 64-filter blocks), per-channel BN+ReLU, dense layers, softmax.
// These are HBM/latency-bound or tiny; they are plain coalesced VALU kernels (fp32 accumulate).
// Layers restated: Gomoku/Build_Model.py:21-86, TicTacToe/Build_Model.py:17-52, Connect4/Build_Model.py:22-75.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "conv3x3.hpp"

namespace gaz {

enum { NACT_NONE = 0, NACT_RELU = 1, NACT_GELU = 2, NACT_TANH = 3 };

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == NACT_RELU) return fmaxf(v, 0.0f);
    if (act == NACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == NACT_TANH) return tanhf(v);
    return v;
}

// ---- stem: KxK conv on the int8 planes [M][CIN] -> act(acc*scale+shift) bf16 [M][COUT]; optional out2 = relu(out1*sB+tB)
struct StemGenArgs {
    const int8_t* in; const float* w;          // w [K*K][COUT][CIN]
    const float* scale; const float* shift; const float* scaleB; const float* shiftB;
    bf16_t* out1; bf16_t* out2;
    int M, H, W, CIN, COUT, K, act;
};
__global__ __launch_bounds__(256) void k_stem_generic(StemGenArgs a) {
    const int groups = a.COUT / 8;                                  // 8 output channels per thread
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long gr = item / groups; const int ch0 = (int)(item % groups) * 8;
    if (gr >= a.M) return;
    const int HW = a.H * a.W, cell = (int)(gr % HW), y = cell / a.W, x = cell % a.W, r = a.K / 2;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ky = 0; ky < a.K; ++ky)
        for (int kx = 0; kx < a.K; ++kx) {
            const int yy = y + ky - r, xx = x + kx - r;
            if ((unsigned)yy >= (unsigned)a.H || (unsigned)xx >= (unsigned)a.W) continue;
            const int8_t* px = a.in + (gr + (long)(ky - r) * a.W + (kx - r)) * a.CIN;
            const float* wt = a.w + ((size_t)(ky * a.K + kx) * a.COUT + ch0) * a.CIN;
            for (int c = 0; c < a.CIN; ++c) {
                const float xv = (float)px[c];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += xv * wt[j * a.CIN + c];
            }
        }
    float v[8], w2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = apply_act(acc[j] * a.scale[ch0 + j] + a.shift[ch0 + j], a.act);
        w2[j] = a.out2 ? fmaxf(v[j] * a.scaleB[ch0 + j] + a.shiftB[ch0 + j], 0.0f) : 0.0f;
    }
    const size_t o = (size_t)gr * a.COUT + ch0;
    *reinterpret_cast<uint4*>(a.out1 + o) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    if (a.out2)
        *reinterpret_cast<uint4*>(a.out2 + o) = make_uint4(pack_bf16(w2[0], w2[1]), pack_bf16(w2[2], w2[3]), pack_bf16(w2[4], w2[5]), pack_bf16(w2[6], w2[7]));
}

// ---- head convolution 32 -> COUT (8 | 4) with a flat fp32 output: one thread per cell computes all COUT outputs, the weights
// sit in LDS as [tap][cin][cout] (every lane reads the same address: broadcast), the 32 input channels of a tap arrive as
// four 16-byte loads.  out[b][cell * COUT + co] = act((acc + bias[co]) * fs[f] + ft[f])  (Gomoku p.c2 3x3 / v.c2 1x1).
struct ConvSmallArgs { const bf16_t* in; const float* w; const float* bias; float* flat; const float* fs; const float* ft; int act; int M, H, W; };
template <int COUT, int K>
__global__ __launch_bounds__(256) void k_conv_small(ConvSmallArgs a) {
    constexpr int CIN = 32, T = K * K;
    __shared__ float wl[T * CIN * COUT];
    for (int i = threadIdx.x; i < T * CIN * COUT; i += 256) {        // [tap][cout][cin] -> [tap][cin][cout]
        const int tap = i / (CIN * COUT), rem = i % (CIN * COUT), co = rem / CIN, c = rem % CIN;
        wl[(tap * CIN + c) * COUT + co] = a.w[i];
    }
    __syncthreads();
    const long gr = (long)blockIdx.x * 256 + threadIdx.x;
    if (gr >= a.M) return;
    const int HW = a.H * a.W, cell = (int)((unsigned long)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W, r = K / 2;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
        const int dy = tap / K - r, dx = tap % K - r;
        if ((unsigned)(y + dy) >= (unsigned)a.H || (unsigned)(x + dx) >= (unsigned)a.W) continue;
        const uint4* px = reinterpret_cast<const uint4*>(a.in + (gr + (long)dy * a.W + dx) * CIN);
        uint4 pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[q] = px[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned pw[4] = {pv[q].x, pv[q].y, pv[q].z, pv[q].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = (j & 1) ? __uint_as_float(pw[j >> 1] & 0xFFFF0000u) : __uint_as_float(pw[j >> 1] << 16);
                const float* wr = &wl[(tap * CIN + q * 8 + j) * COUT];
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] += xv * wr[co];
            }
        }
    }
    const long b = gr / HW;
    float* o = a.flat + (size_t)b * HW * COUT + (size_t)cell * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = acc[co] + (a.bias ? a.bias[co] : 0.0f);
        const int f = cell * COUT + co;
        if (a.fs) v = v * a.fs[f] + a.ft[f];
        o[co] = apply_act(v, a.act);
    }
}

// ---- direct convolution for layers with few channels: in bf16 [M][CIN], w f32 [K*K][COUT][CIN], one thread per (cell, cout)
struct ConvDirectArgs {
    const bf16_t* in; const float* w;
    const float* scaleA; const float* shiftA;   // per cout (null: 1 / 0)
    const bf16_t* res;                          // [M][COUT] residual or null (may alias out1)
    bf16_t* out1; int act1;                     // bf16 [M][COUT] (null in flat mode)
    const float* scaleB; const float* shiftB; bf16_t* out2;
    float* flat; const float* fs; const float* ft; int flat_act;    // flat mode: f32 [B][HW*COUT] = act(v * fs[f] + ft[f]) (fs null: identity)
    int M, H, W, CIN, COUT, K;
};
__global__ __launch_bounds__(256) void k_conv_direct(ConvDirectArgs a) {
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long gr = item / a.COUT; const int co = (int)(item % a.COUT);
    if (gr >= a.M) return;
    const int HW = a.H * a.W, cell = (int)(gr % HW), y = cell / a.W, x = cell % a.W, r = a.K / 2;
    float acc = 0.0f;
    for (int ky = 0; ky < a.K; ++ky)
        for (int kx = 0; kx < a.K; ++kx) {
            const int yy = y + ky - r, xx = x + kx - r;
            if ((unsigned)yy >= (unsigned)a.H || (unsigned)xx >= (unsigned)a.W) continue;
            const bf16_t* px = a.in + (gr + (long)(ky - r) * a.W + (kx - r)) * a.CIN;
            const float* wt = a.w + ((size_t)(ky * a.K + kx) * a.COUT + co) * a.CIN;
            for (int c = 0; c < a.CIN; c += 8) {                     // CIN is a multiple of 8
                const uint4 pv = *reinterpret_cast<const uint4*>(px + c);
                const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc += __uint_as_float(pw[j] << 16) * wt[c + 2 * j];
                    acc += __uint_as_float(pw[j] & 0xFFFF0000u) * wt[c + 2 * j + 1];
                }
            }
        }
    float v = acc * (a.scaleA ? a.scaleA[co] : 1.0f) + (a.shiftA ? a.shiftA[co] : 0.0f);
    if (a.flat) {
        const long b = gr / HW; const int f = cell * a.COUT + co;
        if (a.fs) v = v * a.fs[f] + a.ft[f];
        a.flat[(size_t)b * HW * a.COUT + f] = apply_act(v, a.flat_act);
        return;
    }
    const size_t o = (size_t)gr * a.COUT + co;
    if (a.res) v += bf2f(a.res[o]);
    v = apply_act(v, a.act1);
    a.out1[o] = f2bf(v);
    if (a.out2) a.out2[o] = f2bf(fmaxf(v * a.scaleB[co] + a.shiftB[co], 0.0f));
}

// ---- per-channel BN + ReLU on a bf16 NHWC tensor: out = relu(in * s[c] + t[c])
__global__ __launch_bounds__(256) void k_affine_relu(const bf16_t* in, const float* s, const float* t, bf16_t* out, long n8, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;            // one 8-channel group per thread
    if (i >= n8) return;
    const int c0 = (int)((i * 8) % C);
    const uint4 pv = *reinterpret_cast<const uint4*>(in + i * 8);
    const unsigned pw[4] = {pv.x, pv.y, pv.z, pv.w};
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(pw[j] << 16); v[2 * j + 1] = __uint_as_float(pw[j] & 0xFFFF0000u); }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * s[c0 + j] + t[c0 + j], 0.0f);
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// ---- dense fp32: out[b][n] = act((sum_k in[b][k] * w[k][n]) * s[n] + t[n]); block = 128 threads, 8 rows at a time
__global__ __launch_bounds__(128) void k_dense(const float* in, const float* w, const float* scale, const float* shift, float* out,
                                               int B, int K, int N, int act) {
    extern __shared__ float fl[];                                  // [8][K]
    const int b0 = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * K; i += 128) { const int p = i / K, k = i % K; fl[i] = (b0 + p < B) ? in[(size_t)(b0 + p) * K + k] : 0.0f; }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 128) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < K; ++k) {
            const float wv = w[(size_t)k * N + n];
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] += fl[p * K + k] * wv;
        }
        const float s = scale ? scale[n] : 1.0f, t = shift ? shift[n] : 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) if (b0 + p < B) out[(size_t)(b0 + p) * N + n] = apply_act(acc[p] * s + t, act);
    }
}

// ---- row softmax (A <= 256), one wave per row; logits = 1 copies the raw values (Gumbel policy head)
__global__ __launch_bounds__(64) void k_softmax_rows(const float* in, float* out, int B, int A, int logits) {
    const int b = blockIdx.x, l = threadIdx.x;
    if (b >= B) return;
    float v[4]; float mx = -3.0e38f;
    for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; v[j] = i < A ? in[(size_t)b * A + i] : -3.0e38f; mx = fmaxf(mx, v[j]); }
    if (logits == 1) { for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; if (i < A) out[(size_t)b * A + i] = v[j]; } return; }
    if (logits == 2) {                                              // Stablemax layer (Net/Stablemax.py:8-12)
        float sx[4], ssum = 0.0f;
        for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; sx[j] = i < A ? (v[j] >= 0.0f ? v[j] + 1.0f : 1.0f / (1.0f - v[j])) : 0.0f; ssum += sx[j]; }
        for (int m = 32; m >= 1; m >>= 1) ssum += __shfl_xor(ssum, m, 64);
        for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; if (i < A) out[(size_t)b * A + i] = sx[j] / ssum; }
        return;
    }
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float e[4], sum = 0.0f;
    for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; e[j] = i < A ? expf(v[j] - mx) : 0.0f; sum += e[j]; }
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
    for (int j = 0; j < 4; ++j) { const int i = l + 64 * j; if (i < A) out[(size_t)b * A + i] = e[j] / sum; }
}

}  // namespace gaz
