// resblock.hpp — one whole pre-activation residual block (Net/ResNet/ResNet_Block.py:27-41) as ONE kernel:
//     x_out = Conv3x3_2( relu(bn2( Conv3x3_1( relu(bn1(x)) ) )) ) + x
// Fusing the two convolutions removes the block's intermediate tensors from HBM: the unfused pair moves
// 44 (in) + 44 (h) | 44 (h) + 44 (res) + 44 (x) + 44 (pre-activated copy) MB per block at B = 4096; the fused kernel reads x
// once (+ a re-read of the L2-hot centre rows for the residual) and writes x_out once.
//
// One 512-thread workgroup produces 256 - 2(W+1) output rows (240 for Connect4) of all 128 channels:
//   1. LDS-DMA the raw rows [m0 - 2h, m0 + 256) (h = W + 1) of x into the swizzled image, weights of conv1 tap 0 into slice 0;
//   2. transform the image in place: relu(x * s1 + t1)   (pre-activation BN cannot fold into a conv: the residual needs raw x);
//   3. conv1 as in conv3x3.hpp over the 256 rows [m0 - h, m0 + 256 - h)  (halo rows are recomputed: +6.7 % MFMA work);
//   4. its accumulators -> relu(acc * s2 + t2) -> bf16 -> the SAME LDS region, now the image of h (the x image is dead);
//   5. conv2 over rows [m0, m0 + 256) reading that image (only the first 256 - 2h rows are valid outputs);
//   6. epilogue through an fp32 LDS tile: + bias + residual x (global, L2-hot) -> x_out (a different buffer: neighbouring
//      workgroups still read their halo rows of x).
// The 18 weight slices (9 taps x 2 convs) stream through the two LDS slice buffers as one sequence.
#pragma once
#include "conv3x3.hpp"

namespace gaz {

struct ResBlockArgs {
    const bf16_t* xin; bf16_t* xout;              // [M][128]
    const bf16_t* w1; const bf16_t* w2;           // fragment order, see arrange_conv_weights()
    const float* s1; const float* t1;             // bn1 (applied with ReLU to the input image)
    const float* s2; const float* t2;             // bn2 folded with conv1's bias (ReLU)
    const float* b2;                              // conv2 bias
    int M, H, W;
};

constexpr int RB_ROWS = 256, RB_THREADS = 512;

__global__ __launch_bounds__(RB_THREADS, 2) void k_resblock(ResBlockArgs a) {
    constexpr int CIN = 128, BN = 128, SLOTS = 16, WN = 2, TM = 2, TN = 2;
    constexpr int AROWS = CONV_AROWS_256, ZROW = AROWS - 1, BSL = BN * SLOTS;
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    uint4* Bs = lds + AROWS * SLOTS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lhi = lane >> 5;
    const int h = a.W + 1, HW = a.H * a.W, bmo = RB_ROWS - 2 * h;
    const long m0 = (long)blockIdx.x * bmo;
    const uint4* in4 = reinterpret_cast<const uint4*>(a.xin);

    // ---- 1. image of x: image row q <-> global row m0 - 2h + q, q in [0, 256 + 2h)
    const int n_aslots = (RB_ROWS + 2 * h) * SLOTS;
    for (int base = wave * 64; base < n_aslots; base += RB_THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 - 2 * h + lr;
        gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
    }
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);
    {
        const uint4* w4 = reinterpret_cast<const uint4*>(a.w1);
        for (int base = wave * 64; base < BSL; base += RB_THREADS)
            __builtin_amdgcn_global_load_lds((const void*)(w4 + base + lane), (lds_ptr_t)(Bs + base), 16, 0, 0);
    }
    // per-lane geometry: conv1 row j <-> global m0 - h + j ; conv2 row i <-> global m0 + i
    int lrow[TM]; unsigned vmask1[TM], vmask2[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wm * TM + tm) * 32 + l31;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const long gr = m0 + lrow[tm] - (which == 0 ? h : 0);
            unsigned m = 0;
            if (gr >= 0 && gr < a.M) {
                const int cell = (int)(gr % HW), y = cell / a.W, x = cell % a.W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    m |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << t;
                }
            }
            if (which == 0) vmask1[tm] = m; else vmask2[tm] = m;
        }
    }
    int bbase[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bbase[tn] = lhi * BN + (wn * TN + tn) * 32 + l31;

    __syncthreads();                                // x image + slice 0 landed

    // ---- 2. in-place pre-activation of the image: relu(x * s1 + t1)
    for (int i = tid; i < n_aslots; i += RB_THREADS) {
        const int lr = i / SLOTS, sp = i % SLOTS, ch0 = (sp ^ (lr & 15)) * 8;
        uint4 v = As[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = fmaxf(__uint_as_float(w[j] << 16) * a.s1[ch0 + 2 * j] + a.t1[ch0 + 2 * j], 0.0f);
            const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * a.s1[ch0 + 2 * j + 1] + a.t1[ch0 + 2 * j + 1], 0.0f);
            w[j] = pack_bf16(lo, hi);
        }
        As[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();

    f32x16 acc[TM][TN];
    for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
        for (int tap = 0; tap < 9; ++tap) {
            const int sl = conv * 9 + tap;
            const uint4* Bc = Bs + (sl & 1) * BSL;
            if (sl + 1 < 18) {                      // DMA of the next weight slice (conv1 taps, then conv2 taps)
                uint4* Bn = Bs + ((sl + 1) & 1) * BSL;
                const uint4* wsrc = reinterpret_cast<const uint4*>(sl + 1 < 9 ? a.w1 : a.w2) + (size_t)((sl + 1) % 9) * BSL;
                for (int base = wave * 64; base < BSL; base += RB_THREADS)
                    __builtin_amdgcn_global_load_lds((const void*)(wsrc + base + lane), (lds_ptr_t)(Bn + base), 16, 0, 0);
            }
            const int off = (tap / 3 - 1) * a.W + (tap % 3 - 1);
            int abase[TM], axor[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bool ok = ((conv == 0 ? vmask1[tm] : vmask2[tm]) >> tap) & 1u;
                const int ar = ok ? lrow[tm] + h + off : ZROW;       // both images: operand row = own row + h + tap offset
                abase[tm] = ar * SLOTS; axor[tm] = ar & 15;
            }
            uint4 afr[2][TM], bfr[2][TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + (lhi ^ axor[tm])];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bfr[0][tn] = Bc[bbase[tn]];
#pragma unroll
            for (int ks = 0; ks < CIN / 16; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < CIN / 16) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks + 1) * 2 + lhi) ^ axor[tm])];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bfr[nxt][tn] = Bc[bbase[tn] + (ks + 1) * 2 * BN];
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const bf16x8 af = *reinterpret_cast<bf16x8*>(&afr[cur][tm]);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<bf16x8*>(&bfr[cur][tn]), acc[tm][tn], 0, 0, 0);
                }
            }
            __syncthreads();                        // next slice landed; everyone is done with this slice (and, at tap 8, with the image)
        }
        if (conv == 0) {
            // ---- 4. h = relu(acc * s2 + t2) as bf16 into the image region (row j of h at image row j, same swizzle)
            bf16_t* Hs = reinterpret_cast<bf16_t*>(As);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = (wn * TN + tn) * 32 + l31;
                    const float s2 = a.s2[col], t2 = a.t2[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float v = fmaxf(acc[tm][tn][r] * s2 + t2, 0.0f);
                        Hs[row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))] = (bf16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
                }
            __syncthreads();
        }
    }

    // ---- 6. epilogue: fp32 tile, + bias + residual, rows [m0, m0 + bmo)
    constexpr int CT = BN + 4;
    float* Ct = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (wn * TN + tn) * 32 + l31;
            const float tA = a.b2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                Ct[row * CT + col] = acc[tm][tn][r] + tA;
            }
        }
    __syncthreads();
    const int chunk = tid % 16, r0 = tid / 16;
    for (int row = r0; row < bmo; row += RB_THREADS / 16) {
        const long gr = m0 + row;
        if (gr >= a.M) break;
        const float4 c0 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8 + 4]);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const size_t o = (size_t)gr * BN + chunk * 8;
        const uint4 rv = *reinterpret_cast<const uint4*>(a.xin + o);
        const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(rw[j] << 16); v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u); }
        *reinterpret_cast<uint4*>(a.xout + o) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    }
}

}  // namespace gaz
