// resnet.hip — the ResNet policy/value evaluator as hand-written gfx950 kernels.
//
// Network (Connect4/Build_Model.py:10-88 + Net/ResNet/ResNet_Block.py:5-41; fp32 PyTorch restatement in
// grok_alpha_zero_amd/net.py): stem Conv3x3(4->128)+BN+GELU, N pre-activation residual blocks of two
// Conv3x3(128->128), policy/value heads (Conv3x3 128->8 each, flatten, BN, ReLU, Dense128, BN, ReLU, Dense64,
// Dense7|1, softmax|tanh).  One forward pass evaluates row g of the wave batch for every game.
//
// Trunk convolution = implicit GEMM on the matrix cores: M = B*H*W flattened board cells, N = Cout,
// K = 9*Cin, bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16).  A workgroup owns 128 consecutive
// cells; it stages those rows plus a (W+1)-row halo of the NHWC activation ONCE in LDS (XOR-swizzled
// 16-byte slots, conflict-free ds_read_b128) and serves all nine taps from that image: the tap (dy,dx)
// operand of cell r is image row r + dy*W + dx, zeroed by a per-lane validity bit at board edges.  The
// per-tap weight slice [Cout][Cin] streams through LDS with the next slice prefetched into registers
// during the MFMA phase.  Pre-activation BN cannot fold into the previous conv (the residual needs the
// un-normalised tensor), so it is fused as a dual-output epilogue: conv2 writes the raw stream x and
// relu(bn1_next(x)) in one pass; conv1's epilogue applies bn2 + ReLU; the stem applies BN + GELU.
#include <math.h>
#include <map>
#include <string>
#include <vector>
#include <algorithm>
#include "evaluator.hpp"
#include "conv3x3.hpp"
#include "netops.hpp"
#include "resblock.hpp"
#include "trunk.hpp"
#include "tile_perm.hpp"
#include "puct_core.hpp"      // the fused tree + trunk launch at the end of this file steps the games with the tree kernel's device code
#include "gumbel_core.hpp"

namespace gaz {

static bf16_t f2bf_host(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// k_resblock3 tiling: tile height 64 * TM, either with a halo (valid rows 64 TM - 2 (W + 1), conv1 recomputes the halo) or
// board-aligned (k = floor(64 TM / HW) whole boards per tile, no halo).  Cost = rounds of (2 workgroups x n_cus) x TM; ties go
// to the halo tiling (more valid rows per tile), then to the taller tile.
struct Rb3Plan { int tm, halo, tile_rows; };
static Rb3Plan rb3_plan(int M, int H, int W, int n_cus) {
    Rb3Plan best{3, W + 1, 64 * 3 - 2 * (W + 1)}; double best_cost = 1e30;
    for (int aligned = 0; aligned < 2; ++aligned)
        for (int c = 2; c <= 4; ++c) {
            const int rows = aligned ? (64 * c / (H * W)) * (H * W) : 64 * c - 2 * (W + 1);
            if (rows <= 0) continue;
            const long nt = (M + rows - 1) / rows;
            const double cost = (double)((nt + 2 * n_cus - 1) / (2 * n_cus)) * c;
            if (cost < best_cost || (cost == best_cost && !aligned)) { best_cost = cost; best = Rb3Plan{c, aligned ? 0 : W + 1, rows}; }
        }
    return best;
}
static void rb3_launch(hipStream_t s, ResBlockArgs r, const Rb3Plan& p, int ring) {
    r.halo = p.halo; r.tile_rows = p.tile_rows;
    const int nwg = (r.M + p.tile_rows - 1) / p.tile_rows, tm = p.tm;
    if (tm == 2 && ring == 4) hipLaunchKernelGGL((k_resblock3<2, 4>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<2>(), s, r);
    else if (tm == 2) hipLaunchKernelGGL((k_resblock3<2, 8>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<2>(), s, r);
    else if (tm == 3 && ring == 4) hipLaunchKernelGGL((k_resblock3<3, 4>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<3>(), s, r);
    else if (tm == 3) hipLaunchKernelGGL((k_resblock3<3, 8>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<3>(), s, r);
    else hipLaunchKernelGGL((k_resblock3<4, 4>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<4>(), s, r);
}

// ---- stem: Conv3x3(C_in = 4, int8 planes) -> 128, BN, exact GELU; second output relu(bn1_0(x))
struct StemArgs {
    const int8_t* in;        // [B][HW][4]
    const float* w;          // [9][128][4]
    const float* scale; const float* shift;
    const float* scaleB; const float* shiftB;
    bf16_t* out1; bf16_t* out2;
    int M, H, W;
};

__global__ __launch_bounds__(256) void k_stem(StemArgs a) {
    __shared__ float wl[36][128];
    const int tid = threadIdx.x;
    for (int i = tid; i < 36 * 128; i += 256) {
        const int k = i / 128, ch = i % 128, tap = k / 4, c = k % 4;
        wl[k][ch] = a.w[(tap * 128 + ch) * 4 + c];
    }
    __syncthreads();
    const int chg = tid & 15, rsub = tid >> 4, HW = a.H * a.W;
    const int ch0 = chg * 8;
    float s[8], t[8], sb[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = a.scale[ch0 + j]; t[j] = a.shift[ch0 + j]; sb[j] = a.scaleB[ch0 + j]; tb[j] = a.shiftB[ch0 + j]; }
    const int* in32 = reinterpret_cast<const int*>(a.in);
    for (int pass = 0; pass < 4; ++pass) {
        const long gr = (long)blockIdx.x * 64 + pass * 16 + rsub;
        if (gr >= a.M) continue;
        const int cell = (int)(gr % HW), y = cell / a.W, x = cell % a.W;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(y + dy) >= (unsigned)a.H || (unsigned)(x + dx) >= (unsigned)a.W) continue;
            const int packed = in32[gr + dy * a.W + dx];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xv = (float)(int8_t)((packed >> (8 * c)) & 0xFF);
                const float* wr = &wl[tap * 4 + c][ch0];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += xv * wr[j];
            }
        }
        bf16_t o1[8], o2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j] * s[j] + t[j];
            v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            o1[j] = f2bf(v); o2[j] = f2bf(fmaxf(v * sb[j] + tb[j], 0.0f));
        }
        *reinterpret_cast<uint4*>(a.out1 + (size_t)gr * 128 + ch0) = *reinterpret_cast<uint4*>(o1);
        if (a.out2) *reinterpret_cast<uint4*>(a.out2 + (size_t)gr * 128 + ch0) = *reinterpret_cast<uint4*>(o2);
    }
}

// ---- stem on the matrix cores.  D[channel][cell] = W'[channel][k] * X[k][cell], k = tap * 4 + plane (36, padded to 48).
// The int8 planes are exact in bf16; the fp32 weights (BN scale folded in) are split into bf16 hi + lo halves that
// share the activation operand (6 k-steps), which keeps ~16 mantissa bits.  One wave = one 32-cell tile x all 128
// channels; each lane ends up with 4 consecutive channels of its own cell per accumulator quad -> 8-byte stores.
// GELU is x * Phi(x) with Phi from the Abramowitz-Stegun 7.1.26 erfc polynomial (|err| < 8e-8, output is bf16).
struct StemMArgs {
    const int8_t* in; const uint4* wfrag; const float* shift; bf16_t* out;
    const float* scaleB; const float* shiftB; bf16_t* out2;      // OUT2: second output relu(out * scaleB + shiftB)
    int M, H, W, tiles_per_wave;
};

// CIN input planes (2: Gomoku, 4: Connect4), COUT channels, k = tap * CIN + plane padded to KST k-steps of 16; operand
// k-steps [0, KST) carry the hi halves of the weights, [KST, 2 KST) the lo halves, against the same activations.
template <int CIN> constexpr int stem_ksteps() { return (9 * CIN + 15) / 16; }
template <int CIN, int COUT, bool GELU, bool OUT2>
__global__ __launch_bounds__(256) void k_stem_mfma(StemMArgs a) {
    constexpr int KST = stem_ksteps<CIN>(), TPL = 8 / CIN, NT = COUT / 32;
    __shared__ uint4 wl[2 * KST * 2 * COUT];       // [k-step (KST hi + KST lo)][k-half][channel] x 8 bf16
    __shared__ float sh[COUT]; __shared__ float sBl[OUT2 ? COUT : 1]; __shared__ float tBl[OUT2 ? COUT : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    {
        constexpr int NW = 2 * KST * 2 * COUT / 256;
        uint4 wv[NW];
#pragma unroll
        for (int c = 0; c < NW; ++c) wv[c] = a.wfrag[tid + 256 * c];
#pragma unroll
        for (int c = 0; c < NW; ++c) wl[tid + 256 * c] = wv[c];
    }
    for (int i = tid; i < COUT; i += 256) { sh[i] = a.shift[i]; if (OUT2) { sBl[i] = a.scaleB[i]; tBl[i] = a.shiftB[i]; } }
    __syncthreads();
    const int HW = a.H * a.W;
    for (int t = 0; t < a.tiles_per_wave; ++t) {
        const long tile = ((long)blockIdx.x * 4 + wave) * a.tiles_per_wave + t;
        if (tile * 32 >= a.M) break;               // wave-uniform
        const long gr = tile * 32 + l31;
        const bool rok = gr < a.M;
        const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
        uint4 bfr[KST];
#pragma unroll
        for (int ks = 0; ks < KST; ++ks) {
            int pl[8];                             // the 8 activations of this lane: taps ks * 16 / CIN + lhi * TPL + 0.., CIN planes each
#pragma unroll
            for (int h = 0; h < TPL; ++h) {
                const int tap = ks * (16 / CIN) + lhi * TPL + h;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                int packed = 0;
                if (rok && tap < 9 && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) {
                    if (CIN == 4) packed = reinterpret_cast<const int*>(a.in)[gr + dy * a.W + dx];
                    else packed = reinterpret_cast<const unsigned short*>(a.in)[gr + dy * a.W + dx];
                }
#pragma unroll
                for (int c = 0; c < CIN; ++c) pl[h * CIN + c] = (int)(int8_t)((packed >> (8 * c)) & 0xFF);
            }
            bfr[ks] = make_uint4(s8x2_to_bf16x2(pl[0], pl[1]), s8x2_to_bf16x2(pl[2], pl[3]), s8x2_to_bf16x2(pl[4], pl[5]), s8x2_to_bf16x2(pl[6], pl[7]));
        }
        f32x16 acc[NT];
#pragma unroll
        for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][r] = 0.0f;
#pragma unroll
        for (int ks2 = 0; ks2 < 2 * KST; ++ks2) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(&bfr[ks2 % KST]);
#pragma unroll
            for (int tm = 0; tm < NT; ++tm) {
                const uint4 av = wl[(ks2 * 2 + lhi) * COUT + tm * 32 + l31];
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&av), bf, acc[tm], 0, 0, 0);
            }
        }
        if (!rok) continue;
        // D rows (channels) of lane: (r & 3) + 8 * (r >> 2) + 4 * lhi, column (cell) = l31
#pragma unroll
        for (int tm = 0; tm < NT; ++tm)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch0 = tm * 32 + rg * 8 + lhi * 4;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = acc[tm][rg * 4 + q] + sh[ch0 + q];
                    v[q] = GELU ? gelu_as(u) : fmaxf(u, 0.0f);
                }
                *reinterpret_cast<uint2*>(a.out + (size_t)gr * COUT + ch0) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
                if (OUT2) {
                    float w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = fmaxf(v[q] * sBl[ch0 + q] + tBl[ch0 + q], 0.0f);
                    *reinterpret_cast<uint2*>(a.out2 + (size_t)gr * COUT + ch0) = make_uint2(pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]));
                }
            }
    }
}

// host: [9][COUT][CIN] fp32 stem weights (BN scale folded in) -> the operand layout above
static std::vector<bf16_t> stem_fragments(const float* w, const float* sc, int CIN, int COUT) {
    const int KST = (9 * CIN + 15) / 16;
    std::vector<bf16_t> h((size_t)2 * KST * 2 * COUT * 8, 0);
    for (int ks2 = 0; ks2 < 2 * KST; ++ks2) for (int half = 0; half < 2; ++half) for (int ch = 0; ch < COUT; ++ch) for (int j = 0; j < 8; ++j) {
        const int k = (ks2 % KST) * 16 + half * 8 + j;
        if (k >= 9 * CIN) continue;
        const float wv = w[((k / CIN) * COUT + ch) * CIN + (k % CIN)] * sc[ch];
        const bf16_t hi = f2bf_host(wv);
        unsigned hu = (unsigned)hi << 16; float hf; memcpy(&hf, &hu, 4);
        h[(((size_t)ks2 * 2 + half) * COUT + ch) * 8 + j] = ks2 < KST ? hi : f2bf_host(wv - hf);
    }
    return h;
}

// ---- Dense(K -> N) + optional per-output scale / shift + activation on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
// Block = 32 positions x 128 outputs (4 waves x 32 columns), blockIdx.y = 128-output group; K in chunks staged in LDS.
struct DenseMArgs { const float* in; const float* w; const float* scale; const float* shift; float* out; int B, K, N, act; };
constexpr int DM_KC = 448;                         // 32 x 449 floats = 57 KB of LDS per chunk
__global__ __launch_bounds__(256) void k_dense_mfma(DenseMArgs a) {
    extern __shared__ float fl[];                  // [32][DM_KC + 1]
    constexpr int G = 8, D = 4, FP = DM_KC + 1;
    const int b0 = blockIdx.x * 32, K = a.K, N = a.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int n = blockIdx.y * 128 + wave * 32 + l31, nc = n < N ? n : N - 1;          // clamped column: loads stay in bounds
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += DM_KC) {
        const int kc = K - k0 < DM_KC ? K - k0 : DM_KC, NG = (kc + 2 * G - 1) / (2 * G);
        // round 2: the chunk's first weight groups and ALL of a wave's staging loads (8 rows x 7) are in flight before the first LDS
        // store — the old per-row load -> store loop cost one L2 round trip per row and chunk (40 in a row for K = 1800)
        float bq[D][G];
        auto ldg = [&](int g, float* dst) {
#pragma unroll
            for (int j = 0; j < G; ++j) { const int k = min(g * 2 * G + 2 * j + lhi, kc - 1); dst[j] = a.w[(size_t)(k0 + k) * N + nc]; }
        };
#pragma unroll
        for (int d = 0; d < D - 1; ++d) ldg(d, bq[d]);
        {
            float v[8][7];
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) {
                const size_t rb = (size_t)min(b0 + wv + 4 * pi, a.B - 1) * K + k0;
#pragma unroll
                for (int c = 0; c < 7; ++c) v[pi][c] = a.in[rb + min(ln + 64 * c, kc - 1)];
            }
            __syncthreads();                       // the previous chunk's readers are done
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) {
                const int p = wv + 4 * pi;
                const bool rok = b0 + p < a.B;
#pragma unroll
                for (int c = 0; c < 7; ++c) { const int k = ln + 64 * c; if (k < NG * 2 * G) fl[p * FP + k] = (rok && k < kc) ? v[pi][c] : 0.0f; }
            }
        }
        __syncthreads();
        for (int g0 = 0; g0 < NG; g0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int g = g0 + d;
                ldg(g + D - 1, bq[(d + D - 1) % D]);
                if (g < NG) {                       // wave-uniform
                    float av[G];
#pragma unroll
                    for (int j = 0; j < G; ++j) av[j] = fl[l31 * FP + g * 2 * G + 2 * j + lhi];
#pragma unroll
                    for (int j = 0; j < G; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[d][j], acc, 0, 0, 0);
                }
            }
        }
    }
    if (n >= N) return;
    const float s = a.scale ? a.scale[n] : 1.0f, t = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (b0 + row < a.B) a.out[(size_t)(b0 + row) * N + n] = apply_act(acc[r] * s + t, a.act);
    }
}

// ---- heads: Dense(F -> 128) + folded BN + ReLU, fp32.  Block = 128 threads (one output each), 8 positions at a time.
__global__ __launch_bounds__(128) void k_dense1(const float* feat, const float* w, const float* scale, const float* shift,
                                                float* out, int B, int F) {
    extern __shared__ float fl[];                  // [8][F]
    const int j = threadIdx.x;
    const int b0 = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * F; i += 128) {
        const int p = i / F, k = i % F;
        fl[i] = (b0 + p < B) ? feat[(size_t)(b0 + p) * F + k] : 0.0f;
    }
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < F; ++k) {
        const float wv = w[(size_t)k * 128 + j];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] += fl[p * F + k] * wv;
    }
    const float s = scale[j], t = shift[j];
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if (b0 + p < B) out[(size_t)(b0 + p) * 128 + j] = fmaxf(acc[p] * s + t, 0.0f);
}

// ---- heads: the same Dense(F -> 128) + folded BN + ReLU for BOTH heads on the matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32): block = 32 positions x 128 outputs, 4 waves x 32 columns; blockIdx.y = head.
struct Dense1Args { const float* feat[2]; const float* w[2]; const float* scale[2]; const float* shift[2]; float* out[2]; int B, F; };
// Round 3: split-K inside the workgroup.  The kernel is a chain of L2 round trips (168 weight loads per lane with five groups in flight; 5.6 us of
// MFMA in 16.7 us at one wave per SIMD), so eight waves take half the k-groups each — waves 0-3 the first half, waves 4-7 the second — and the
// two partial sums meet in LDS: out = relu((first + second) * s + t).  One summation order whatever the batch (rows stay batch-independent).
constexpr int D1_THREADS = 512;
__global__ __launch_bounds__(D1_THREADS) void k_dense1_mfma(Dense1Args a) {
    extern __shared__ float fl[];                  // [32][FP], FP = F rounded up to 16, + 1 (zero-padded columns); then [4][16][64] partial sums
    constexpr int G = 8, D = 6;                    // 16 k per group, six-deep register ring of weight groups
    const int head = blockIdx.y, b0 = blockIdx.x * 32, F = a.F, NG = (F + 2 * G - 1) / (2 * G), FP = NG * 2 * G + 1;
    const float* feat = a.feat[head]; const float* w = a.w[head];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5, wq = wave & 3, half = wave >> 2, n = wq * 32 + l31;
    const int NGH = (NG + 1) / 2, gA = half * NGH, gB = half ? NG : NGH;          // this wave's k-groups [gA, gB)
    // A[i = l31][k + lhi], B[k + lhi][j = l31].  The weights come from L2 (~1 us away under load): the next five groups are in
    // flight while a group multiplies, and the first five are requested BEFORE the features are staged (round 2: the kernel is a chain
    // of round trips with one wave per SIMD, nothing else hides them).  Loads past the last row are clamped (their A columns are zero).
    float bq[D][G];
    auto ldg = [&](int g, float* dst) {
#pragma unroll
        for (int j = 0; j < G; ++j) { const int k = min(g * 2 * G + 2 * j + lhi, F - 1); dst[j] = w[(size_t)k * 128 + n]; }
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) ldg(gA + d, bq[d]);
    // features -> LDS: one wave per position row, coalesced; loads are unconditional (clamped).  ALL of a wave's loads (4 rows x 6) are
    // issued before its first LDS store: one round trip for the staging instead of one per row.
    {
        float v[4][6];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int p = wave + 8 * pi;
            const size_t rb = (size_t)min(b0 + p, a.B - 1) * F;
#pragma unroll
            for (int c = 0; c < 6; ++c) v[pi][c] = feat[rb + min(lane + 64 * c, F - 1)];
        }
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int p = wave + 8 * pi;
            const bool rok = b0 + p < a.B;
#pragma unroll
            for (int c = 0; c < 6; ++c) { const int k = lane + 64 * c; if (k < FP) fl[p * FP + k] = (rok && k < F) ? v[pi][c] : 0.0f; }
        }
        for (int p = wave; p < 32; p += 8)           // F > 384 (not the Connect4 net): the rest of the row, the old way
            for (int k = lane + 64 * 6; k < FP; k += 64) fl[p * FP + k] = (b0 + p < a.B && k < F) ? feat[(size_t)min(b0 + p, a.B - 1) * F + k] : 0.0f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int g0 = gA; g0 < gB; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int g = g0 + d;
            ldg(g + D - 1, bq[(d + D - 1) % D]);
            if (g < gB) {                           // wave-uniform
                float av[G];
#pragma unroll
                for (int j = 0; j < G; ++j) av[j] = fl[l31 * FP + g * 2 * G + 2 * j + lhi];
#pragma unroll
                for (int j = 0; j < G; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bq[d][j], acc, 0, 0, 0);
            }
        }
    }
    float* part = fl + 32 * FP;                    // [wq][r][lane]: the second half's sums
    if (half) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wq * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (half) return;
    const float s = a.scale[head][n], t = a.shift[head][n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (b0 + row < a.B) a.out[head][(size_t)(b0 + row) * 128 + n] = fmaxf((acc[r] + part[(wq * 16 + r) * 64 + lane]) * s + t, 0.0f);
    }
}

// ---- heads tail: Dense(128 -> 64) -> Dense(64 -> A | 1) -> softmax | tanh.  One wave per position.
struct TailArgs {
    const float* p_d1; const float* v_d1;          // [B][128]
    const float* p_w2; const float* p_b2; const float* p_w3; const float* p_b3;
    const float* v_w2; const float* v_b2; const float* v_w3; const float* v_b3;
    float* policy; float* value; int B, A, logits;
};
// 16 positions per block (8 threads each), blockIdx.y = head; Dense(128 -> 64) weights staged in LDS once per block.
// Summation order per output is k ascending starting from the bias, independent of the batch composition.
__global__ __launch_bounds__(128) void k_tail(TailArgs a) {
    __shared__ float w2l[128 * 64]; __shared__ float xl[16][128]; __shared__ float hl[16][64]; __shared__ float lg[16][64];
    __shared__ float w3l[64 * 8];                  // Dense(64 -> nout) weights when nout <= 8 (Connect4: 7 | 1)
    const int head = blockIdx.y, tid = threadIdx.x, p = tid >> 3, q = tid & 7, b0 = blockIdx.x * 16;
    const float* d1 = head == 0 ? a.p_d1 : a.v_d1;
    const float* w2 = head == 0 ? a.p_w2 : a.v_w2; const float* b2 = head == 0 ? a.p_b2 : a.v_b2;
    const float* w3 = head == 0 ? a.p_w3 : a.v_w3; const float* b3 = head == 0 ? a.p_b3 : a.v_b3;
    const int nout = head == 0 ? a.A : 1;
    {   // 16 + 4 float4 per thread, all loads issued before the first LDS store
        float4 wv[16], xv[4];
#pragma unroll
        for (int c = 0; c < 16; ++c) wv[c] = reinterpret_cast<const float4*>(w2)[tid + 128 * c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = tid + 128 * c, pp = min(b0 + i / 32, a.B - 1);
            xv[c] = reinterpret_cast<const float4*>(d1 + (size_t)pp * 128)[i % 32];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) reinterpret_cast<float4*>(w2l)[tid + 128 * c] = wv[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) reinterpret_cast<float4*>(&xl[0][0])[tid + 128 * c] = xv[c];
        if (nout <= 8) {
            float w3v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) w3v[c] = w3[min(tid + 128 * c, 64 * nout - 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) w3l[tid + 128 * c] = w3v[c];
        }
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = b2[q * 8 + j];
    for (int k = 0; k < 128; ++k) {
        const float xv = xl[p][k];
        const float4 wa = *reinterpret_cast<const float4*>(&w2l[k * 64 + q * 8]), wb = *reinterpret_cast<const float4*>(&w2l[k * 64 + q * 8 + 4]);
        acc[0] += xv * wa.x; acc[1] += xv * wa.y; acc[2] += xv * wa.z; acc[3] += xv * wa.w;
        acc[4] += xv * wb.x; acc[5] += xv * wb.y; acc[6] += xv * wb.z; acc[7] += xv * wb.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) hl[p][q * 8 + j] = acc[j];
    __syncthreads();
    for (int o = q; o < nout; o += 8) {
        float z = b3[o];
        if (nout <= 8) { for (int k = 0; k < 64; ++k) z += hl[p][k] * w3l[k * nout + o]; }
        else { for (int k = 0; k < 64; ++k) z += hl[p][k] * w3[k * nout + o]; }
        lg[p][o] = z;
    }
    __syncthreads();
    if (b0 + p >= a.B) return;
    const int b = b0 + p;
    if (head == 0) {
        for (int o = q; o < nout; o += 8) {
            if (a.logits == 1) a.policy[(size_t)b * a.A + o] = lg[p][o];
            else if (a.logits == 2) {               // Stablemax layer (Net/Stablemax.py:8-12): s(x) = x + 1 | 1 / (1 - x), normalised
                float sum = 0.f;
                for (int k = 0; k < nout; ++k) { const float x = lg[p][k]; sum += x >= 0.0f ? x + 1.0f : 1.0f / (1.0f - x); }
                const float x = lg[p][o];
                a.policy[(size_t)b * a.A + o] = (x >= 0.0f ? x + 1.0f : 1.0f / (1.0f - x)) / sum;
            } else {
                float mx = lg[p][0];
                for (int k = 1; k < nout; ++k) mx = fmaxf(mx, lg[p][k]);
                float sum = 0.f;
                for (int k = 0; k < nout; ++k) sum += expf(lg[p][k] - mx);
                a.policy[(size_t)b * a.A + o] = expf(lg[p][o] - mx) / sum;
            }
        }
    } else if (q == 0) {
        a.value[b] = tanhf(lg[p][0]);
    }
}

// ------------------------------------------------------------------------------------------ host
struct ResNetEvaluator : Evaluator {
    int H, W, C, A, HW, blocks, filters, nmax, logits;
    std::map<std::string, float*> f32;              // device fp32 tensors by name
    std::map<std::string, bf16_t*> b16;             // device bf16 conv weights by name
    bf16_t *X = nullptr, *Aa = nullptr, *Hh = nullptr, *X2 = nullptr;
    bool fused = true;
    int stamp_calls = 0, n_cus = 256;
    bf16_t* stem_frag = nullptr;
    bool trunk_m16 = true;                          // ... on v_mfma_f32_16x16x32_bf16 (higher clock held); GAZ_TRUNK_M16=0 -> 32x32x16 (bit-identical to k_resblock3)
    bool trunk_mix = true;                          // ... with the last partial round in cheaper 96-row tiles (k_trunk_mix); GAZ_TRUNK_MIX=0 -> one tile shape
    bool trunk_whole = true;                        // ... including the stem and the heads' first convolution; GAZ_TRUNK_WHOLE=0 -> k_stem_mfma / k_conv_heads
    bool trunk = true;                              // k_trunk: every block in one kernel (trunk.hpp); GAZ_TRUNK=0 -> one k_resblock3 per block
    bf16_t* trunk_w = nullptr; float* trunk_prm = nullptr;
    std::string dom_label;
    uint8_t *perm_big = nullptr, *perm_small = nullptr;    // TrunkArgs::perm for the 128-row and the 96-row tile (tile_perm.hpp tile_layout; GAZ_TILE_PERM=0: none)
    unsigned boff_big = 0, boff_small = 0; int perm_clashes = 0;      // TrunkArgs::boff; residue clashes the layout could not avoid (0: no LDS conflict in any fragment read)
    float *pfeat = nullptr, *vfeat = nullptr, *pd1 = nullptr, *vd1 = nullptr;
    std::vector<void*> allocs;
    bool loaded = false;
    std::vector<hipEvent_t> tev;                    // pairs around the trunk conv chain
    int64_t n_trunk_launches = 0;

    ~ResNetEvaluator() override { for (void* p : allocs) hipFree(p); for (auto e : tev) hipEventDestroy(e); }

    template <class T> T* dalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr; allocs.push_back(p); return (T*)p; }

    int load(const gaz_tensor* t, int n, hipStream_t s, std::string* err) override {
        std::map<std::string, const gaz_tensor*> by;
        for (int i = 0; i < n; ++i) by[t[i].name] = &t[i];
        auto need = [&](const std::string& name, int64_t numel) -> const gaz_tensor* {
            auto it = by.find(name);
            if (it == by.end()) { *err = "missing tensor " + name; return nullptr; }
            if (it->second->numel != numel) { *err = "tensor " + name + " has " + std::to_string(it->second->numel) + " elements, expected " + std::to_string(numel); return nullptr; }
            return it->second;
        };
        auto up_f32 = [&](const std::string& name, int64_t numel) -> bool {
            const gaz_tensor* g = need(name, numel); if (!g) return false;
            float* d = dalloc<float>(numel); if (!d) { *err = "hipMalloc"; return false; }
            hipMemcpy(d, g->data, numel * 4, hipMemcpyHostToDevice); f32[name] = d; return true;
        };
        // a block's conv1 / conv2 weights share one allocation (conv2 right behind conv1): k_resblock3 walks them as 18 slices
        bool pair_first = false; bf16_t* pair_next = nullptr;
        auto up_b16 = [&](const std::string& name, int64_t numel) -> bool {      // conv weights [9][cout][cin] -> fragment order
            const gaz_tensor* g = need(name, numel); if (!g) return false;
            std::vector<bf16_t> h(numel);
            const int cin = 128, cout = (int)(numel / (9 * cin));
            arrange_conv_weights(g->data, cout, cin, h.data(), f2bf_host);
            bf16_t* d = pair_next ? pair_next : dalloc<bf16_t>(numel * (pair_first ? 2 : 1)); if (!d) { *err = "hipMalloc"; return false; }
            pair_next = pair_first ? d + numel : nullptr; pair_first = false;
            hipMemcpy(d, h.data(), numel * 2, hipMemcpyHostToDevice); b16[name] = d; return true;
        };
        const int Fc = filters, F = HW * 8;
        if (!up_f32("stem.w", 9 * 128 * C) || !up_f32("stem.scale", 128) || !up_f32("stem.shift", 128)) return 1;
        {   // MFMA stem operand, BN scale folded into the weights
            if (C != 4) { *err = "stem expects 4 input planes"; return 1; }
            const std::vector<bf16_t> h = stem_fragments(by["stem.w"]->data, by["stem.scale"]->data, 4, 128);
            stem_frag = dalloc<bf16_t>(h.size()); if (!stem_frag) { *err = "hipMalloc"; return 1; }
            hipMemcpy(stem_frag, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        }
        for (int i = 0; i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i);
            pair_first = true;
            if (!up_f32(b + ".bn1.scale", Fc) || !up_f32(b + ".bn1.shift", Fc) || !up_b16(b + ".conv1.w", 9LL * Fc * Fc) ||
                !up_f32(b + ".conv1.scale", Fc) || !up_f32(b + ".conv1.shift", Fc) || !up_b16(b + ".conv2.w", 9LL * Fc * Fc) ||
                !up_f32(b + ".conv2.bias", Fc)) return 1;
        }
        if (blocks > 0) {   // k_trunk operands: the slices of all blocks as one array, the per-block parameters as [block][5][128]
            const size_t WB = 18 * (size_t)Fc * Fc;
            trunk_w = dalloc<bf16_t>(blocks * WB); trunk_prm = dalloc<float>((size_t)blocks * TR_PRM);
            if (!trunk_w || !trunk_prm) { *err = "hipMalloc"; return 1; }
            for (int i = 0; i < blocks; ++i) {
                const std::string b = "block" + std::to_string(i);
                hipMemcpy(trunk_w + i * WB, b16[b + ".conv1.w"], WB * 2, hipMemcpyDeviceToDevice);     // conv2 sits right behind conv1
                const char* names[5] = {".bn1.scale", ".bn1.shift", ".conv1.scale", ".conv1.shift", ".conv2.bias"};
                for (int k = 0; k < 5; ++k) hipMemcpy(trunk_prm + ((size_t)i * 5 + k) * 128, f32[b + names[k]], 128 * 4, hipMemcpyDeviceToDevice);
            }
        }
        if (!(getenv("GAZ_TILE_PERM") && atoi(getenv("GAZ_TILE_PERM")) == 0) && !perm_big && 96 / HW >= 1) {
            // the kernels' static sit-out masks (trunk.hpp conv9): 128-row tile = two wave rows (y = 0 | x = 0) and (y = H - 1 | x = W - 1); 96-row
            // tile = one wave row (y = 0 | y = H - 1 | x = 0).  Both permutations must deliver exactly that, or neither is used.
            unsigned want_big[8] = {0x007u, 0x049u, 0, 0, 0x1C0u, 0x124u, 0, 0}, want_small[6] = {0x007u, 0x1C0u, 0x049u, 0, 0, 0};
            const TileLayout lb = tile_layout(H, W, 128 / HW, 128, 64, 2, want_big), ls = tile_layout(H, W, 96 / HW, 96, 96, 3, want_small);
            const bool ok = !lb.perm.empty() && !ls.perm.empty() && lb.boff.size() <= 3 && ls.boff.size() <= 3;
            if (ok && (perm_big = dalloc<uint8_t>(lb.perm.size())) && (perm_small = dalloc<uint8_t>(ls.perm.size()))) {
                hipMemcpy(perm_big, lb.perm.data(), lb.perm.size(), hipMemcpyHostToDevice); hipMemcpy(perm_small, ls.perm.data(), ls.perm.size(), hipMemcpyHostToDevice);
                boff_big = boff_small = 0;
                for (size_t b = 0; b < lb.boff.size(); ++b) boff_big |= (unsigned)lb.boff[b] << (8 * b);
                for (size_t b = 0; b < ls.boff.size(); ++b) boff_small |= (unsigned)ls.boff[b] << (8 * b);
                perm_clashes = lb.clashes + ls.clashes;
            } else { perm_big = perm_small = nullptr; }
        }
        if (!up_b16("heads.conv.w", 9LL * 32 * Fc) || !up_f32("heads.conv.bias", 32)) return 1;
        for (const char* pre : {"p", "v"}) {
            const std::string p = pre; const int nout = p == "p" ? A : 1;
            if (!up_f32(p + ".bn0.scale", F) || !up_f32(p + ".bn0.shift", F) || !up_f32(p + ".d1.w", (int64_t)F * 128) ||
                !up_f32(p + ".d1.scale", 128) || !up_f32(p + ".d1.shift", 128) || !up_f32(p + ".d2.w", 128 * 64) ||
                !up_f32(p + ".d2.bias", 64) || !up_f32(p + ".d3.w", 64 * nout) || !up_f32(p + ".d3.bias", nout)) return 1;
        }
        hipStreamSynchronize(s);
        loaded = true;
        return 0;
    }

    void conv_trunk(hipStream_t s, const bf16_t* in, const bf16_t* w, const float* sA, const float* tA, const bf16_t* res,
                    bf16_t* out1, int act1, const float* sB, const float* tB, bf16_t* out2, int M) {
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.wgt = w; a.scaleA = sA; a.shiftA = tA; a.res = res; a.out1 = out1; a.act1 = act1;
        a.scaleB = sB; a.shiftB = tB; a.out2 = out2; a.M = M; a.H = H; a.W = W;
        static const int variant = getenv("GAZ_CONV_VARIANT") ? atoi(getenv("GAZ_CONV_VARIANT")) : 0;
        if (variant == 0) {          // 256 rows / 512 threads / whole-tap slices, one workgroup per CU
            const size_t lds = conv_lds_bytes<128, 128, 256, 1>();
            hipLaunchKernelGGL((k_conv3x3<128, 128, 256, 4, 2, 2, 2, 1, 1, 0>), dim3((M + 255) / 256), dim3(512), lds, s, a);
        } else {                     // 128 rows / 256 threads / half-tap slices, two workgroups per CU overlap their phases
            const size_t lds = conv_lds_bytes<128, 128, 128, 2>();
            hipLaunchKernelGGL((k_conv3x3<128, 128, 128, 2, 2, 2, 2, 2, 2, 0>), dim3((M + 127) / 128), dim3(256), lds, s, a);
        }
        n_trunk_launches++;
    }

    bool supports_row_base() const override { return true; }
    bool head_features(const float** p, const float** v, int* pr, int* vr) override { *p = pfeat; *v = vfeat; *pr = HW * 8; *vr = HW * 8; return true; }
    bool supports_split() const override { return fused && trunk && trunk_whole && blocks > 0 && HW <= 128; }
    // kernel arguments + grid of the whole-trunk launch for rows [p0, p0 + n) (k_trunk_mix / k_trunk with stem and heads inside)
    bool make_trunk_plan(const int8_t* in, int n, int p0, TrunkLaunchPlan& P) {
        if (!loaded || !supports_split()) return false;
        const int M = n * HW;
        TrunkArgs& r = P.args; memset(&r, 0, sizeof(r));
        r.xin = this->X + (size_t)p0 * HW * 128; r.xout = this->X2 + (size_t)p0 * HW * 128; r.w = trunk_w; r.prm = trunk_prm; r.M = M; r.H = H; r.W = W; r.nblocks = blocks;
        r.tile_rows = (128 / HW) * HW; r.stamps = nullptr; r.perm = perm_big; r.perm_small = perm_small; r.boff = boff_big; r.boff_small = boff_small;
        r.planes = in; r.stem_frag = reinterpret_cast<const uint4*>(stem_frag); r.stem_shift = f32["stem.shift"];
        r.hw = b16["heads.conv.w"]; r.hbias = f32["heads.conv.bias"]; r.p_fs = f32["p.bn0.scale"]; r.p_ft = f32["p.bn0.shift"];
        r.v_fs = f32["v.bn0.scale"]; r.v_ft = f32["v.bn0.shift"];
        r.p_feat = this->pfeat + (size_t)p0 * HW * 8; r.v_feat = this->vfeat + (size_t)p0 * HW * 8;
        int nwg = (M + r.tile_rows - 1) / r.tile_rows;
        bool mix = false;
        if (trunk_mix && 96 / HW >= 1 && 96 / HW < 128 / HW) {    // whole rounds of 128-row tiles, the rest in 96-row tiles (k_trunk_mix)
            const int slots = 2 * n_cus, bb = 128 / HW, sb = 96 / HW;
            int nb = (n / (bb * slots)) * slots, ns = (n - nb * bb + sb - 1) / sb;
            const double cost_mix = nb / slots + 0.78 * ((ns + slots - 1) / slots), cost_big = (nwg + slots - 1) / slots;
            // one of several game groups: the other groups' tiles fill a ragged last round, so every board that fits one goes into a 3-board tile
            // (46 us per board and slot against 51 in a 2-board tile) and only the remainder into 2-board tiles
            if (shared_chip && perm_big && perm_small) { nb = n / bb; ns = (n - nb * bb + sb - 1) / sb; }
            if (cost_mix < cost_big || (shared_chip && perm_big && perm_small)) { mix = true; r.n_big = nb; r.small_rows = sb * HW; nwg = nb + ns; }
        }
        P.nwg = nwg; P.mix = mix; P.lds_bytes = (unsigned)trunk_lds_bytes(128);
        return true;
    }
    TrunkLaunchPlan fused_plan;
    const void* trunk_plan(const int8_t* in, int n, int p0, const FuseHandoff& h) override {
        if (!trunk_m16 || !make_trunk_plan(in, n, p0, fused_plan)) return nullptr;
        TrunkArgs& r = fused_plan.args;
        r.ready = h.ready; r.epoch = h.epoch; r.eval_done = h.eval_done; r.fuse_fault = h.fuse_fault; r.spin_ticks = h.spin_ticks; r.test_fault_mod = h.test_fault_mod; r.stamps = h.stamps;
        r.queue = (perm_big && perm_small && fused_plan.mix) ? h.queue : nullptr;       // the completion queue needs the variant that maps image rows to games itself
        return &fused_plan;
    }
    bool plan_uses_queue(const void* plan) const override { return plan && static_cast<const TrunkLaunchPlan*>(plan)->args.queue != nullptr; }
    bool shared_chip = false;
    void set_shared_chip(bool on) override { static const bool off = getenv("GAZ_GROUP_BIG_TILES") && atoi(getenv("GAZ_GROUP_BIG_TILES")) == 0; shared_chip = on && !off; }
    int round_rows() const override { return 2 * n_cus * (128 / HW); }
    void forward(hipStream_t s, const int8_t* in, float* policy, float* value, int n, bool timing, int p0 = 0) override {
        forward_trunk(s, in, n, timing, p0);
        forward_heads(s, policy, value, n, p0);
    }
    // stem + every residual block (+ the heads' first convolution): planes of rows [p0, p0 + n) -> pfeat / vfeat rows [p0, p0 + n)
    void forward_trunk(hipStream_t s, const int8_t* in, int n, bool timing, int p0) override {
        if (!loaded) return;                        // engine_create without weights: outputs stay as they are
        const int M = n * HW;
        // activation / feature buffers of rows [p0, p0 + n)
        bf16_t* const X = this->X + (size_t)p0 * HW * 128; bf16_t* const X2 = this->X2 + (size_t)p0 * HW * 128;
        bf16_t* const Aa = this->Aa + (size_t)p0 * HW * 128; bf16_t* const Hh = this->Hh + (size_t)p0 * HW * 128;
        float* const pfeat = this->pfeat + (size_t)p0 * HW * 8; float* const vfeat = this->vfeat + (size_t)p0 * HW * 8;
        StemArgs st; st.in = in; st.w = f32["stem.w"]; st.scale = f32["stem.scale"]; st.shift = f32["stem.shift"];
        st.scaleB = blocks ? f32["block0.bn1.scale"] : f32["stem.scale"]; st.shiftB = blocks ? f32["block0.bn1.shift"] : f32["stem.shift"];
        st.out1 = X; st.out2 = fused ? nullptr : Aa; st.M = M; st.H = H; st.W = W;    // the fused blocks pre-activate on load
        const bool use_trunk = fused && trunk && blocks > 0 && HW <= 128;
        const bool whole = use_trunk && trunk_whole;     // stem and heads' first convolution inside k_trunk as well
        if (whole) {
        } else if (fused) {
            StemMArgs sm; memset(&sm, 0, sizeof(sm)); sm.in = in; sm.wfrag = reinterpret_cast<const uint4*>(stem_frag); sm.shift = f32["stem.shift"]; sm.out = X;
            sm.M = M; sm.H = H; sm.W = W;
            const int tiles = (M + 31) / 32;
            sm.tiles_per_wave = std::max(1, (tiles + 1343) / 2688);     // ~2.7k waves: ten per CU, each amortising the 24-KB weight stage
            hipLaunchKernelGGL((k_stem_mfma<4, 128, true, false>), dim3((tiles + 4 * sm.tiles_per_wave - 1) / (4 * sm.tiles_per_wave)), dim3(256), 0, s, sm);
        } else {
            hipLaunchKernelGGL(k_stem, dim3((M + 63) / 64), dim3(256), 0, s, st);
        }
        hipEvent_t e0 = 0, e1 = 0;
        if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); tev.push_back(e0); tev.push_back(e1); hipEventRecord(e0, s); }
        bf16_t* cur = X;
        if (use_trunk) {                            // every block in one kernel, k whole boards per workgroup (trunk.hpp)
            TrunkArgs r; memset(&r, 0, sizeof(r)); r.xin = X; r.xout = X2; r.w = trunk_w; r.prm = trunk_prm; r.M = M; r.H = H; r.W = W; r.nblocks = blocks;
            r.tile_rows = (128 / HW) * HW; r.stamps = nullptr; r.perm = perm_big; r.perm_small = perm_small; r.boff = boff_big; r.boff_small = boff_small;
            r.planes = in; r.stem_frag = reinterpret_cast<const uint4*>(stem_frag); r.stem_shift = f32["stem.shift"];
            r.hw = b16["heads.conv.w"]; r.hbias = f32["heads.conv.bias"]; r.p_fs = f32["p.bn0.scale"]; r.p_ft = f32["p.bn0.shift"];
            r.v_fs = f32["v.bn0.scale"]; r.v_ft = f32["v.bn0.shift"]; r.p_feat = pfeat; r.v_feat = vfeat;
            int nwg = (M + r.tile_rows - 1) / r.tile_rows;
            bool mix = false;
            if (whole && trunk_mix && 96 / HW >= 1 && 96 / HW < 128 / HW) {    // whole rounds of 128-row tiles, the rest in 96-row tiles (k_trunk_mix)
                const int slots = 2 * n_cus, bb = 128 / HW, sb = 96 / HW;
                const int nb = (n / (bb * slots)) * slots, ns = (n - nb * bb + sb - 1) / sb;
                const double cost_mix = nb / slots + 0.78 * ((ns + slots - 1) / slots), cost_big = (nwg + slots - 1) / slots;
                if (cost_mix < cost_big) { mix = true; r.n_big = nb; r.small_rows = sb * HW; nwg = nb + ns; }
            }
            static const char* stamp_path = getenv("GAZ_TRUNK_STAMPS");    // diagnostic: phase stamps of the third launch -> file
            const bool stamp = stamp_path && ++stamp_calls == 3;
            if (stamp) { hipMalloc((void**)&r.stamps, (size_t)nwg * 128 * 8); hipMemsetAsync(r.stamps, 0, (size_t)nwg * 128 * 8, s); }
            if (mix && trunk_m16 && r.perm && r.perm_small) hipLaunchKernelGGL((k_trunk_mix<8, 2, true, true, true, true>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            else if (mix && trunk_m16) hipLaunchKernelGGL((k_trunk_mix<8, 2, true, true, true>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            else if (whole && trunk_m16) hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, true, true, false, true>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            else if (mix) hipLaunchKernelGGL((k_trunk_mix<8, 2, true, true>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            else if (whole) hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, true, true>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            else hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, false, false>), dim3(nwg), dim3(TR_THREADS), trunk_lds_bytes(128), s, r);
            if (stamp) {
                std::vector<unsigned long long> hst((size_t)nwg * 128);
                hipStreamSynchronize(s);
                hipMemcpy(hst.data(), r.stamps, hst.size() * 8, hipMemcpyDeviceToHost); hipFree(r.stamps);
                if (FILE* f = fopen(stamp_path, "wb")) { fwrite(hst.data(), 8, hst.size(), f); fclose(f); }
            }
            cur = X2;
        }
        for (int i = 0; fused && !use_trunk && i < blocks; ++i) {             // one kernel per residual block (resblock.hpp)
            const std::string b = "block" + std::to_string(i);
            ResBlockArgs r; r.xin = cur; r.xout = cur == X ? X2 : X; r.w1 = b16[b + ".conv1.w"]; r.w2 = b16[b + ".conv2.w"];
            r.s1 = f32[b + ".bn1.scale"]; r.t1 = f32[b + ".bn1.shift"]; r.s2 = f32[b + ".conv1.scale"]; r.t2 = f32[b + ".conv1.shift"];
            r.b2 = f32[b + ".conv2.bias"]; r.M = M; r.H = H; r.W = W; r.stamps = nullptr;
            static const int rbv = getenv("GAZ_RB") ? atoi(getenv("GAZ_RB")) : 3;
            static const int rb_tm = getenv("GAZ_RB_TM") ? atoi(getenv("GAZ_RB_TM")) : 0;
            static const int rb_ring = getenv("GAZ_RB_RING") ? atoi(getenv("GAZ_RB_RING")) : 8;
            Rb3Plan plan = rb3_plan(M, H, W, n_cus);
            if (rb_tm) plan = Rb3Plan{rb_tm, W + 1, 64 * rb_tm - 2 * (W + 1)};
            const int bmo = rbv == 3 ? plan.tile_rows : RB_ROWS - 2 * (W + 1);
            const size_t lds = conv_lds_bytes<128, 128, 256, 1>();
            const int nwg = (M + bmo - 1) / bmo;
            static const char* stamp_path = getenv("GAZ_RB_STAMPS");       // diagnostic: phase stamps of one launch -> file
            const bool stamp = stamp_path && i == 1 && ++stamp_calls == 3;
            if (stamp) { hipMalloc((void**)&r.stamps, (size_t)nwg * RB_STAMPS * 8); hipMemsetAsync(r.stamps, 0, (size_t)nwg * RB_STAMPS * 8, s); }
            if (rbv == 3) rb3_launch(s, r, plan, rb_ring);
            else hipLaunchKernelGGL(k_resblock, dim3(nwg), dim3(RB_THREADS), lds, s, r);
            if (stamp) {
                std::vector<unsigned long long> hst((size_t)nwg * RB_STAMPS);
                hipStreamSynchronize(s);
                hipMemcpy(hst.data(), r.stamps, hst.size() * 8, hipMemcpyDeviceToHost); hipFree(r.stamps);
                if (FILE* f = fopen(stamp_path, "wb")) { fwrite(hst.data(), 8, hst.size(), f); fclose(f); }
            }
            cur = r.xout;
        }
        for (int i = 0; !fused && i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i), nb = "block" + std::to_string(i + 1);
            conv_trunk(s, Aa, b16[b + ".conv1.w"], f32[b + ".conv1.scale"], f32[b + ".conv1.shift"], nullptr, Hh, ACT_RELU,
                       nullptr, nullptr, nullptr, M);
            const bool last = i + 1 == blocks;
            conv_trunk(s, Hh, b16[b + ".conv2.w"], nullptr, f32[b + ".conv2.bias"], X, X, ACT_NONE,
                       last ? nullptr : f32[nb + ".bn1.scale"], last ? nullptr : f32[nb + ".bn1.shift"], last ? nullptr : Aa, M);
        }
        if (timing) hipEventRecord(e1, s);
        if (!whole) {   // first conv of both heads + flat BN + ReLU (k_conv_heads)
            HeadsConvArgs hc; hc.in = cur; hc.wgt = b16["heads.conv.w"]; hc.bias = f32["heads.conv.bias"];
            hc.p_fs = f32["p.bn0.scale"]; hc.p_ft = f32["p.bn0.shift"]; hc.v_fs = f32["v.bn0.scale"]; hc.v_ft = f32["v.bn0.shift"];
            hc.p_feat = pfeat; hc.v_feat = vfeat; hc.M = M; hc.H = H; hc.W = W;
            hipLaunchKernelGGL(k_conv_heads, dim3((M + HC_ROWS - 1) / HC_ROWS), dim3(RB3_THREADS), hc_lds_bytes(), s, hc);
        }
    }
    // Dense-1 (both heads) + tail: pfeat / vfeat rows [p0, p0 + n) -> policy / value
    void forward_heads(hipStream_t s, float* policy, float* value, int n, int p0) override {
        if (!loaded) return;
        float* const pfeat = this->pfeat + (size_t)p0 * HW * 8; float* const vfeat = this->vfeat + (size_t)p0 * HW * 8;
        float* const pd1 = this->pd1 + (size_t)p0 * 128; float* const vd1 = this->vd1 + (size_t)p0 * 128;
        const int F = HW * 8;
        Dense1Args d; d.B = n; d.F = F;
        d.feat[0] = pfeat; d.w[0] = f32["p.d1.w"]; d.scale[0] = f32["p.d1.scale"]; d.shift[0] = f32["p.d1.shift"]; d.out[0] = pd1;
        d.feat[1] = vfeat; d.w[1] = f32["v.d1.w"]; d.scale[1] = f32["v.d1.scale"]; d.shift[1] = f32["v.d1.shift"]; d.out[1] = vd1;
        hipLaunchKernelGGL(k_dense1_mfma, dim3((n + 31) / 32, 2), dim3(D1_THREADS), (size_t)32 * ((F + 15) / 16 * 16 + 1) * 4 + 4 * 16 * 64 * 4, s, d);
        TailArgs t; t.p_d1 = pd1; t.v_d1 = vd1; t.p_w2 = f32["p.d2.w"]; t.p_b2 = f32["p.d2.bias"]; t.p_w3 = f32["p.d3.w"];
        t.p_b3 = f32["p.d3.bias"]; t.v_w2 = f32["v.d2.w"]; t.v_b2 = f32["v.d2.bias"]; t.v_w3 = f32["v.d3.w"]; t.v_b3 = f32["v.d3.bias"];
        t.policy = policy; t.value = value; t.B = n; t.A = A; t.logits = logits;
        hipLaunchKernelGGL(k_tail, dim3((n + 15) / 16, 2), dim3(128), 0, s, t);
    }

    bool ready() const override { return loaded; }
    void timing_reset() override { for (auto e : tev) hipEventDestroy(e); tev.clear(); n_trunk_launches = 0; }
    void timing_get(double* ms, int64_t* launches) override {
        double t = 0;
        for (size_t i = 0; i + 1 < tev.size(); i += 2) { float a = 0; hipEventElapsedTime(&a, tev[i], tev[i + 1]); t += a; }
        *ms = t; *launches = fused && trunk ? (int64_t)(tev.size() / 2) : (int64_t)(tev.size() / 2) * (fused ? 1 : 2) * blocks;
    }
    const char* dominant_kernel(int n, double* flops) override {
        const double conv = 2.0 * (double)n * HW * 128.0 * 1152.0;
        *flops = fused ? 2 * conv : conv;
        if (fused && trunk) {
            *flops = 2 * conv * blocks;
            if (trunk_whole && trunk_m16 && trunk_mix && perm_big && perm_small) {
                // MFMA rows x taps actually issued / counted (n HW rows x 9 taps): 128-row tiles run 30 of 36 tile-taps per wave, 96-row tiles 45 of 54
                const int slots = 2 * n_cus, bb = 128 / HW, sb = 96 / HW, nb = (n / (bb * slots)) * slots, ns = (n - nb * bb + sb - 1) / sb;
                const int nwg = (n + bb - 1) / bb;
                const bool mix = nb / slots + 0.78 * ((ns + slots - 1) / slots) < (double)((nwg + slots - 1) / slots);      // make_trunk_plan's choice
                const double share = n > 0 && mix ? (nb * 128.0 * 30.0 / 36.0 + ns * 96.0 * 45.0 / 54.0) / ((double)n * HW) : 1.0;
                char buf[640];
                snprintf(buf, sizeof(buf), "k_trunk_mix (stem + the whole residual trunk + heads' first conv in one launch: blocks x two 3x3 convs 128->128 counted, "
                         "implicit GEMM on v_mfma_f32_16x16x32_bf16, activations resident in LDS; the cells of each board edge form whole MFMA tiles that sit out "
                         "the taps on which they read only zero padding: MFMA work issued = %.3f of the counted FLOPs)", share);
                dom_label = buf;
                if (mix) return dom_label.c_str();
            }
            if (trunk_whole) return trunk_m16 ? "k_trunk_mix (stem + the whole residual trunk + heads' first conv in one launch: blocks x two 3x3 convs 128->128 counted, implicit GEMM on v_mfma_f32_16x16x32_bf16, activations resident in LDS)"
                                              : "k_trunk_mix (stem + the whole residual trunk + heads' first conv in one launch: blocks x two 3x3 convs 128->128 counted, implicit GEMM on v_mfma_f32_32x32x16_bf16, activations resident in LDS)";
            return trunk_m16 ? "k_trunk (the whole residual trunk: blocks x two 3x3 convs 128->128, implicit GEMM on v_mfma_f32_16x16x32_bf16, activations resident in LDS)"
                             : "k_trunk (the whole residual trunk: blocks x two 3x3 convs 128->128, implicit GEMM on v_mfma_f32_32x32x16_bf16, activations resident in LDS)";
        }
        return fused ? "k_resblock3 (whole residual block: two 3x3 convs 128->128, implicit GEMM on MFMA 32x32x16 bf16)"
                     : "k_conv3x3<128,128> (trunk 3x3 conv, implicit GEMM on MFMA 32x32x16 bf16)";
    }
};

// ------------------------------------------------------------------------------------------ Gomoku / TicTacToe
// Same trunk kernel (k_conv3x3) where the layer is 128-wide; the 256-channel first block of the Gomoku net uses the
// CIN = 256 / 128-row instantiation (3x3 conv1 and the 1x1 residual projection); narrow layers go through netops.hpp.
struct GenericEvaluator : Evaluator {
    int H, W, C, A, HW, blocks, F, nmax, logits; bool gomoku;
    std::map<std::string, float*> f32; std::map<std::string, bf16_t*> b16; std::map<std::string, const gaz_tensor*> by;
    std::vector<void*> allocs; bool loaded = false; std::string lerr;
    bf16_t *X0 = nullptr, *A0 = nullptr, *X = nullptr, *Aa = nullptr, *Hh = nullptr, *Va = nullptr, *PH = nullptr, *VH = nullptr;
    float *pfeat = nullptr, *vfeat = nullptr, *pd1 = nullptr, *vd1 = nullptr, *pd2 = nullptr, *vd2 = nullptr, *plog = nullptr;
    std::vector<hipEvent_t> tev; int trunk_convs = 0; bool one_launch = false;   // one_launch: the bracket holds ONE kernel (the whole trunk, block 0 included)
    bool fused = true, block0_fused = false, block0_inplace = false; int n_cus = 256, fused_blocks = 0;
    hipStream_t side_stream = 0; hipEvent_t ev_fork = 0, ev_join = 0;     // value head next to the policy head (Gomoku)   // Gomoku: k_block0 + k_resblock3 (one kernel per block)
    bool trunk = true, trunk_m16 = true; bf16_t* trunk_w = nullptr; float* trunk_prm = nullptr;      // ... or k_block0 + ONE k_trunk launch for blocks 1.. (GAZ_TRUNK=0: per block)
    // ... or block 0 INSIDE that launch (trunk.hpp B0; GAZ_BLOCK0_IN_TRUNK=0: k_block0 ahead of it): block 0's 29 slices + the blocks' 18 each
    bool block0_in_trunk = true; bf16_t* trunk_w0 = nullptr; float* trunk_prm0 = nullptr;
    bool stem_in_trunk_ok = true;
    bf16_t* stem_frag = nullptr;

    ~GenericEvaluator() override {
        if (side_stream) { hipStreamSynchronize(side_stream); hipEventDestroy(ev_fork); hipEventDestroy(ev_join); hipStreamDestroy(side_stream); }
        for (void* p : allocs) hipFree(p); for (auto e : tev) hipEventDestroy(e);
    }
    template <class T> T* dalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, (n + 64) * sizeof(T)) != hipSuccess) return nullptr; allocs.push_back(p); return (T*)p; }
    bool ready() const override { return loaded; }

    const gaz_tensor* need(const std::string& name, int64_t numel) {
        auto it = by.find(name);
        if (it == by.end()) { lerr = "missing tensor " + name; return nullptr; }
        if (it->second->numel != numel) { lerr = "tensor " + name + ": " + std::to_string(it->second->numel) + " elements, expected " + std::to_string(numel); return nullptr; }
        return it->second;
    }
    bool up(const std::string& name, int64_t numel) {
        const gaz_tensor* g = need(name, numel); if (!g) return false;
        float* d = dalloc<float>(numel); if (!d) { lerr = "hipMalloc"; return false; }
        hipMemcpy(d, g->data, numel * 4, hipMemcpyHostToDevice); f32[name] = d; return true;
    }
    bool up_mfma(const std::string& name, int cout, int cin, int ntaps) {
        const int64_t numel = (int64_t)ntaps * cout * cin;
        const gaz_tensor* g = need(name, numel); if (!g) return false;
        std::vector<bf16_t> h(numel);
        arrange_conv_weights(g->data, cout, cin, h.data(), f2bf_host, ntaps);
        bf16_t* d = dalloc<bf16_t>(numel); if (!d) { lerr = "hipMalloc"; return false; }
        hipMemcpy(d, h.data(), numel * 2, hipMemcpyHostToDevice); b16[name] = d; return true;
    }

    // the 29 K = 128 weight slices of k_block0: conv1 low / high input half (9 taps each), conv2 (9), projection low / high
    bool up_block0(const std::string& b) {
        const gaz_tensor* c1 = need(b + ".conv1.w", 9LL * 128 * 256); const gaz_tensor* c2 = need(b + ".conv2.w", 9LL * 128 * 128);
        const gaz_tensor* pj = need(b + ".proj.w", 128LL * 256); if (!c1 || !c2 || !pj) return false;
        const size_t SL = (size_t)128 * 128;                           // bf16 elements per slice
        std::vector<bf16_t> full1(9 * 2 * SL), fullp(2 * SL), h(29 * SL);
        arrange_conv_weights(c1->data, 128, 256, full1.data(), f2bf_host, 9);     // [tap][16 k-steps][2][128][8]: k-steps 0-7 = channels 0-127
        arrange_conv_weights(pj->data, 128, 256, fullp.data(), f2bf_host, 1);
        for (int half = 0; half < 2; ++half)
            for (int tap = 0; tap < 9; ++tap)
                memcpy(&h[((size_t)half * 9 + tap) * SL], &full1[((size_t)tap * 2 + half) * SL], SL * sizeof(bf16_t));
        arrange_conv_weights(c2->data, 128, 128, &h[18 * SL], f2bf_host, 9);
        memcpy(&h[27 * SL], &fullp[0], 2 * SL * sizeof(bf16_t));
        bf16_t* d = dalloc<bf16_t>(h.size()); if (!d) { lerr = "hipMalloc"; return false; }
        hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice); b16[b + ".w29"] = d; return true;
    }

    // conv1 / conv2 of a 128 -> 128 block in ONE allocation (conv2 right behind conv1): k_resblock3 walks them as 18 slices
    bool up_mfma_pair(const std::string& n1, const std::string& n2, int F_) {
        const int64_t numel = 9LL * F_ * F_;
        const gaz_tensor* g1 = need(n1, numel); const gaz_tensor* g2 = need(n2, numel); if (!g1 || !g2) return false;
        std::vector<bf16_t> h(2 * numel);
        arrange_conv_weights(g1->data, F_, F_, h.data(), f2bf_host, 9);
        arrange_conv_weights(g2->data, F_, F_, h.data() + numel, f2bf_host, 9);
        bf16_t* d = dalloc<bf16_t>(2 * numel); if (!d) { lerr = "hipMalloc"; return false; }
        hipMemcpy(d, h.data(), 2 * numel * 2, hipMemcpyHostToDevice); b16[n1] = d; b16[n2] = d + numel; return true;
    }

    int load(const gaz_tensor* t, int n, hipStream_t s, std::string* err) override {
        by.clear(); for (int i = 0; i < n; ++i) by[t[i].name] = &t[i];
        const int SC = gomoku ? 256 : 128, K = gomoku ? 3 : 5;
        bool ok = up("stem.w", (int64_t)K * K * SC * C) && up("stem.scale", SC) && up("stem.shift", SC);
        if (ok && gomoku) {                         // MFMA stem operand (2 planes x 9 taps -> 256 channels), BN scale folded in
            const std::vector<bf16_t> h = stem_fragments(by["stem.w"]->data, by["stem.scale"]->data, 2, 256);
            stem_frag = dalloc<bf16_t>(h.size()); ok = stem_frag != nullptr;
            if (ok) hipMemcpy(stem_frag, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        }
        for (int i = 0; ok && i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i); const int cin = i == 0 ? SC : F;
            ok = up(b + ".bn1.scale", cin) && up(b + ".bn1.shift", cin) && up(b + ".conv1.scale", F) && up(b + ".conv1.shift", F) && up(b + ".conv2.bias", F);
            if (!ok) break;
            if (gomoku && cin == F) ok = up_mfma_pair(b + ".conv1.w", b + ".conv2.w", F);
            else if (gomoku) ok = up_mfma(b + ".conv1.w", F, cin, 9) && up_mfma(b + ".conv2.w", F, F, 9);
            else ok = up(b + ".conv1.w", 9LL * F * cin) && up(b + ".conv2.w", 9LL * F * F);
            if (ok && cin != F) ok = (gomoku ? up_mfma(b + ".proj.w", F, cin, 1) : up(b + ".proj.w", (int64_t)F * cin)) && up(b + ".proj.bias", F);
            if (ok && gomoku && cin != F && F == 128 && cin == 256) ok = up_block0(b);
        }
        if (ok && gomoku) {
            ok = up("p.bn0.scale", F) && up("p.bn0.shift", F) && up("v.bn0.scale", F) && up("v.bn0.shift", F) &&
                 up_mfma("p.c1.w", 32, F, 9) && up("p.c1.scale", 32) && up("p.c1.shift", 32) && up_mfma("v.c1.w", 32, F, 9) && up("v.c1.scale", 32) && up("v.c1.shift", 32) &&
                 up("p.c2.w", 9 * 8 * 32) && up("p.c2.bias", 8) && up("p.bn2.scale", HW * 8) && up("p.bn2.shift", HW * 8) &&
                 up("v.c2.w", 4 * 32) && up("v.c2.bias", 4) && up("v.bn2.scale", HW * 4) && up("v.bn2.shift", HW * 4) &&
                 up("p.d1.w", (int64_t)HW * 8 * 512) && up("p.d1.scale", 512) && up("p.d1.shift", 512) && up("p.d2.w", 512LL * A) && up("p.d2.bias", A) &&
                 up("v.d1.w", (int64_t)HW * 4 * 256) && up("v.d1.scale", 256) && up("v.d1.shift", 256) && up("v.d2.w", 256 * 128) && up("v.d2.scale", 128) && up("v.d2.shift", 128) &&
                 up("v.d3.w", 128) && up("v.d3.bias", 1);
        } else if (ok) {
            ok = up("p.c.w", 8 * F) && up("p.c.scale", 8) && up("p.c.shift", 8) && up("v.c.w", 4 * F) && up("v.c.scale", 4) && up("v.c.shift", 4) &&
                 up("p.d1.w", HW * 8 * 128) && up("p.d1.bias", 128) && up("p.d2.w", 128 * 64) && up("p.d2.bias", 64) && up("p.d3.w", 64 * A) && up("p.d3.bias", A) &&
                 up("v.d1.w", HW * 4 * 128) && up("v.d1.bias", 128) && up("v.d2.w", 128 * 64) && up("v.d2.bias", 64) && up("v.d3.w", 64) && up("v.d3.bias", 1);
        }
        if (ok && gomoku && blocks > 1 && F == 128) {     // k_trunk operands of blocks 1..: slices as one array, parameters as [block][5][128]
            const size_t WB = 18 * (size_t)F * F; const int nb = blocks - 1;
            trunk_w = dalloc<bf16_t>(nb * WB); trunk_prm = dalloc<float>((size_t)nb * TR_PRM);
            ok = trunk_w && trunk_prm; if (!ok) lerr = "hipMalloc";
            for (int i = 1; ok && i < blocks; ++i) {
                const std::string b = "block" + std::to_string(i);
                hipMemcpy(trunk_w + (i - 1) * WB, b16[b + ".conv1.w"], WB * 2, hipMemcpyDeviceToDevice);       // conv2 sits right behind conv1 (up_mfma_pair)
                const char* names[5] = {".bn1.scale", ".bn1.shift", ".conv1.scale", ".conv1.shift", ".conv2.bias"};
                for (int k = 0; k < 5; ++k) hipMemcpy(trunk_prm + ((size_t)(i - 1) * 5 + k) * 128, f32[b + names[k]], 128 * 4, hipMemcpyDeviceToDevice);
            }
            if (ok && b16.count("block0.w29")) {     // block 0 inside the launch: its slices ahead of the blocks', its parameters as TrunkArgs::prm0
                const size_t S0 = 29 * (size_t)F * F;
                trunk_w0 = dalloc<bf16_t>(S0 + nb * WB); trunk_prm0 = dalloc<float>(1024);
                ok = trunk_w0 && trunk_prm0; if (!ok) lerr = "hipMalloc";
                if (ok) {
                    hipMemcpy(trunk_w0, b16["block0.w29"], S0 * 2, hipMemcpyDeviceToDevice);
                    hipMemcpy(trunk_w0 + S0, trunk_w, nb * WB * 2, hipMemcpyDeviceToDevice);
                    std::vector<float> h(1024, 0.0f);
                    const float* b2 = by["block0.conv2.bias"]->data; const float* bp = by["block0.proj.bias"]->data;
                    memcpy(&h[0], by["block0.bn1.scale"]->data, 256 * 4); memcpy(&h[256], by["block0.bn1.shift"]->data, 256 * 4);
                    memcpy(&h[640], by["block0.conv1.scale"]->data, 128 * 4); memcpy(&h[768], by["block0.conv1.shift"]->data, 128 * 4);
                    for (int c = 0; c < 128; ++c) h[896 + c] = b2[c] + bp[c];
                    hipMemcpy(trunk_prm0, h.data(), 1024 * 4, hipMemcpyHostToDevice);
                }
            }
        }
        hipStreamSynchronize(s);
        if (!ok) { *err = lerr; return 1; }
        loaded = true; return 0;
    }

    template <int CIN, int NT> void conv_mfma(hipStream_t s, const bf16_t* in, const bf16_t* w, const float* sA, const float* tA, const bf16_t* res,
                                               bf16_t* out1, int act1, const float* sB, const float* tB, bf16_t* out2, int M) {
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.wgt = w; a.scaleA = sA; a.shiftA = tA; a.res = res; a.out1 = out1; a.act1 = act1; a.scaleB = sB; a.shiftB = tB; a.out2 = out2;
        a.M = M; a.H = H; a.W = W;
        if (CIN == 128) {
            const size_t lds = conv_lds_bytes<128, 128, 256, 1>();
            hipLaunchKernelGGL((k_conv3x3<128, 128, 256, 4, 2, 2, 2, 1, 1, 0, NT>), dim3((M + 255) / 256), dim3(512), lds, s, a);
        } else {
            const size_t lds = conv_lds_bytes<256, 128, 128, 4>();
            hipLaunchKernelGGL((k_conv3x3<256, 128, 128, 2, 2, 2, 2, 4, 1, 0, NT>), dim3((M + 127) / 128), dim3(256), lds, s, a);
        }
    }
    void conv_mfma32(hipStream_t s, const bf16_t* in, const bf16_t* w, const float* sA, const float* tA, bf16_t* out1, int M) {
        ConvArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.wgt = w; a.scaleA = sA; a.shiftA = tA; a.out1 = out1; a.act1 = ACT_RELU; a.M = M; a.H = H; a.W = W;
        const size_t lds = conv_lds_bytes<128, 32, 256, 1>();
        hipLaunchKernelGGL((k_conv3x3<128, 32, 256, 8, 1, 1, 1, 1, 1, 0, 9>), dim3((M + 255) / 256), dim3(512), lds, s, a);
    }
    void conv_direct(hipStream_t s, const bf16_t* in, const float* w, int cin, int cout, int K, const float* sA, const float* tA, const bf16_t* res,
                     bf16_t* out1, int act1, const float* sB, const float* tB, bf16_t* out2, float* flat, const float* fs, const float* ft,
                     int flat_act, int M) {
        ConvDirectArgs a; memset(&a, 0, sizeof(a));
        a.in = in; a.w = w; a.scaleA = sA; a.shiftA = tA; a.res = res; a.out1 = out1; a.act1 = act1; a.scaleB = sB; a.shiftB = tB; a.out2 = out2;
        a.flat = flat; a.fs = fs; a.ft = ft; a.flat_act = flat_act; a.M = M; a.H = H; a.W = W; a.CIN = cin; a.COUT = cout; a.K = K;
        const long items = (long)M * cout;
        hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, a);
    }
    void dense(hipStream_t s, const float* in, const std::string& w, const float* sc, const float* sh, float* out, int n, int K, int N, int act) {
        if (gomoku && N >= 32) {                    // fp32 MFMA path
            DenseMArgs d; d.in = in; d.w = f32[w]; d.scale = sc; d.shift = sh; d.out = out; d.B = n; d.K = K; d.N = N; d.act = act;
            hipLaunchKernelGGL(k_dense_mfma, dim3((n + 31) / 32, (N + 127) / 128), dim3(256), (size_t)32 * (DM_KC + 1) * 4, s, d);
            return;
        }
        hipLaunchKernelGGL(k_dense, dim3((n + 7) / 8), dim3(128), (size_t)8 * K * 4, s, in, f32[w], sc, sh, out, n, K, N, act);
    }
    float* g(const std::string& k) { return f32[k]; }
    bool head_features(const float** p, const float** v, int* pr, int* vr) override { *p = pfeat; *v = vfeat; *pr = HW * 8; *vr = HW * 4; return true; }

    // ---- the forward pass in two parts (engine.hip: the fused tree + trunk launch replaces forward_trunk by its own launch of the same kernel)
    bf16_t* trunk_out = nullptr;                    // raw trunk activation the heads read (set by forward_trunk / the fused launch's plan)
    bool fuse_heads = false;
    void forward(hipStream_t s, const int8_t* in, float* policy, float* value, int n, bool timing, int = 0) override {
        forward_trunk(s, in, n, timing, 0);
        forward_heads(s, policy, value, n, 0);
    }
    bool fusable() const { return gomoku && fused && blocks > 1 && block0_in_trunk && trunk && trunk_m16 && trunk_w0 && HW <= 256; }
    bool supports_split() const override { return loaded && fusable(); }
    TrunkLaunchPlan fused_plan;
    // the Gomoku trunk launch with the stem inside (trunk.hpp S0): it reads nothing but the int8 planes, so it can share a launch with the tree step
    const void* trunk_plan(const int8_t* in, int n, int p0, const FuseHandoff& h) override {
        if (!supports_split() || p0 != 0) return nullptr;
        TrunkArgs& t = fused_plan.args; memset(&t, 0, sizeof(t));
        t.prm0 = trunk_prm0; t.xout = Hh; t.w = trunk_w0; t.prm = trunk_prm; t.M = n * HW; t.H = H; t.W = W; t.nblocks = blocks - 1; t.tile_rows = HW;
        t.planes = in; t.stem_frag = reinterpret_cast<const uint4*>(stem_frag); t.stem_shift = g("stem.shift");
        t.ready = h.ready; t.epoch = h.epoch; t.eval_done = h.eval_done; t.fuse_fault = h.fuse_fault; t.spin_ticks = h.spin_ticks; t.test_fault_mod = h.test_fault_mod;
        t.stamps = h.stamps; t.queue = h.queue;
        fused_plan.nwg = n; fused_plan.mix = 0; fused_plan.lds_bytes = (unsigned)trunk_lds_bytes_s0();
        trunk_out = Hh; fuse_heads = true;
        return &fused_plan;
    }
    bool plan_uses_queue(const void* plan) const override { return plan && static_cast<const TrunkLaunchPlan*>(plan)->args.queue != nullptr; }
    void forward_trunk(hipStream_t s, const int8_t* in, int n, bool timing, int) override {
        if (!loaded) return;
        const int M = n * HW, SC = gomoku ? 256 : 128;
        StemGenArgs st; memset(&st, 0, sizeof(st));
        st.in = in; st.w = g("stem.w"); st.scale = g("stem.scale"); st.shift = g("stem.shift"); st.scaleB = g("block0.bn1.scale"); st.shiftB = g("block0.bn1.shift");
        st.out1 = X0; st.out2 = A0; st.M = M; st.H = H; st.W = W; st.CIN = C; st.COUT = SC; st.K = gomoku ? 3 : 5; st.act = gomoku ? NACT_RELU : NACT_GELU;
        const bool fuse = gomoku && fused && blocks > 1;
        // round 3: the stem inside the one trunk launch too (trunk.hpp S0; GAZ_STEM_IN_TRUNK=1): no stem kernel, no 256-channel stem tensor,
        // bit-identical outputs (tests/test_evaluator_gpu.py) — and measured SLOWER on its own: with one 512-thread workgroup per CU nothing
        // overlaps the four half-image stem passes (trunk launch 2298 -> 2444 us per 2048 positions against the 86 us of k_stem_mfma it
        // replaces: evaluator pass 2702 -> 2762 us, gpurun_out r03 gmk1 / gmk2).  Kept as an opt-in: it is what lets the launch depend on
        // nothing but the int8 planes, i.e. what a fused tree + trunk launch for Gomoku needs.
        const bool stem_in_trunk = fuse && stem_in_trunk_ok && block0_in_trunk && trunk && trunk_m16 && trunk_w0 && HW <= 256;
        if (stem_in_trunk) {
        } else if (gomoku) {
            StemMArgs sm; memset(&sm, 0, sizeof(sm)); sm.in = in; sm.wfrag = reinterpret_cast<const uint4*>(stem_frag); sm.shift = g("stem.shift");
            sm.out = X0; sm.scaleB = g("block0.bn1.scale"); sm.shiftB = g("block0.bn1.shift"); sm.out2 = A0; sm.M = M; sm.H = H; sm.W = W;
            const int tiles = (M + 31) / 32;
            sm.tiles_per_wave = std::max(1, (tiles + 1343) / 2688);
            // round 2: with the fused first block (k_block0) the stem writes only the raw tensor — block 0 pre-activates its operand in
            // LDS — which halves this HBM-bound kernel's writes (236 MB less per forward at 2048 positions).  GAZ_STEM_A0=1: the old way
            static const bool stem_a0 = getenv("GAZ_STEM_A0") && atoi(getenv("GAZ_STEM_A0")) != 0;
            block0_inplace = gomoku && fused && blocks > 1 && b16.count("block0.w29") && !stem_a0;
            if (block0_inplace) hipLaunchKernelGGL((k_stem_mfma<2, 256, false, false>), dim3((tiles + 4 * sm.tiles_per_wave - 1) / (4 * sm.tiles_per_wave)), dim3(256), 0, s, sm);
            else hipLaunchKernelGGL((k_stem_mfma<2, 256, false, true>), dim3((tiles + 4 * sm.tiles_per_wave - 1) / (4 * sm.tiles_per_wave)), dim3(256), 0, s, sm);
        } else {
            hipLaunchKernelGGL(k_stem_generic, dim3((unsigned)(((long)M * (SC / 8) + 255) / 256)), dim3(256), 0, s, st);
        }
        hipEvent_t e0 = 0, e1 = 0;
        if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); tev.push_back(e0); tev.push_back(e1); hipEventRecord(e0, s); }
        trunk_convs = 0; fused_blocks = 0; one_launch = false;
        bf16_t* cur = X;                            // raw trunk activation after the last block
        for (int i = 0; i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i), nb = "block" + std::to_string(i + 1);
            const bool first = i == 0, last = i + 1 == blocks;
            if (fuse && first && block0_in_trunk && trunk && trunk_m16 && trunk_w0 && HW <= 256) {      // the whole trunk, block 0 included, in one launch
                TrunkArgs t; memset(&t, 0, sizeof(t));
                t.x0 = X0; t.prm0 = trunk_prm0; t.xout = Hh; t.w = trunk_w0; t.prm = trunk_prm; t.M = M; t.H = H; t.W = W; t.nblocks = blocks - 1; t.tile_rows = HW;
                t.planes = in; t.stem_frag = reinterpret_cast<const uint4*>(stem_frag); t.stem_shift = g("stem.shift");
                if (stem_in_trunk) hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, false, false, false, true, 8, true, 0, true>), dim3(n), dim3(512), trunk_lds_bytes_s0(), s, t);
                else hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, false, false, false, true, 8, true>), dim3(n), dim3(512), trunk_lds_bytes(256), s, t);
                cur = Hh; fused_blocks += blocks; block0_fused = true; one_launch = true;
                break;
            }
            if (fuse && first && b16.count(b + ".w29")) {          // the 256 -> 128 block with its projection, one kernel (k_block0)
                Block0Args r; memset(&r, 0, sizeof(r));
                r.a0 = block0_inplace ? nullptr : A0; r.x0 = X0; r.xout = X; r.w = b16[b + ".w29"]; r.s2 = g(b + ".conv1.scale"); r.t2 = g(b + ".conv1.shift");
                r.s1 = g(b + ".bn1.scale"); r.t1 = g(b + ".bn1.shift");
                r.b2 = g(b + ".conv2.bias"); r.bp = g(b + ".proj.bias"); r.M = M; r.H = H; r.W = W;
                const Rb3Plan plan = rb3_plan(M, H, W, n_cus);
                r.halo = plan.halo; r.tile_rows = plan.tile_rows;
                const int nwg = (M + plan.tile_rows - 1) / plan.tile_rows;
                if (plan.tm == 2) hipLaunchKernelGGL((k_block0<2, 8>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<2>(), s, r);
                else if (plan.tm == 3) hipLaunchKernelGGL((k_block0<3, 8>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<3>(), s, r);
                else hipLaunchKernelGGL((k_block0<4, 4>), dim3(nwg), dim3(RB3_THREADS), rb3_lds_bytes<4>(), s, r);
                cur = X; fused_blocks++; block0_fused = true;
                if (trunk && trunk_w && HW <= 256) {   // blocks 1.. in one launch: one board per workgroup, residual stream through L2 (trunk.hpp, RESG)
                    TrunkArgs t; memset(&t, 0, sizeof(t));
                    t.xin = X; t.xout = Hh; t.w = trunk_w; t.prm = trunk_prm; t.M = M; t.H = H; t.W = W; t.nblocks = blocks - 1; t.tile_rows = HW;
                    // default: 8 waves (WM = 4 x WN = 2, the Connect4 wave tile on 16x16x32), one workgroup per CU, x and operand images in
                    // LDS; GAZ_TRUNK_M16=0: 4 waves of 128 cells x 64 channels on 32x32x16, two workgroups per CU, residual stream through L2
                    if (trunk_m16) hipLaunchKernelGGL((k_trunk<2, 2, 8, 2, false, false, false, true, 8>), dim3(n), dim3(512), trunk_lds_bytes(256), s, t);
                    else hipLaunchKernelGGL((k_trunk<4, 2, 4, 2, false, false, true>), dim3(n), dim3(TR_THREADS), trunk_lds_bytes(256, true), s, t);
                    cur = Hh; fused_blocks += blocks - 1;
                    break;
                }
                continue;
            }
            if (fuse && !first) {                   // whole block in one kernel: raw x in, raw x out (pre-activation on load)
                ResBlockArgs r; memset(&r, 0, sizeof(r));
                r.xin = cur; r.xout = cur == X ? Hh : X; r.w1 = b16[b + ".conv1.w"]; r.w2 = b16[b + ".conv2.w"];
                r.s1 = g(b + ".bn1.scale"); r.t1 = g(b + ".bn1.shift"); r.s2 = g(b + ".conv1.scale"); r.t2 = g(b + ".conv1.shift");
                r.b2 = g(b + ".conv2.bias"); r.M = M; r.H = H; r.W = W;
                const Rb3Plan plan = rb3_plan(M, H, W, n_cus);
                rb3_launch(s, r, plan, plan.tm == 4 ? 4 : 8);
                cur = r.xout; fused_blocks++;
                continue;                           // the heads pre-activate the raw output themselves (k_conv_head32)
            }
            const bf16_t* ain = first ? A0 : Aa;
            const float* sB = last ? (gomoku ? g("p.bn0.scale") : nullptr) : (fuse ? nullptr : g(nb + ".bn1.scale"));
            const float* tB = last ? (gomoku ? g("p.bn0.shift") : nullptr) : (fuse ? nullptr : g(nb + ".bn1.shift"));
            bf16_t* o2 = sB ? Aa : nullptr;
            if (gomoku) {
                if (first) {
                    conv_mfma<256, 9>(s, ain, b16[b + ".conv1.w"], g(b + ".conv1.scale"), g(b + ".conv1.shift"), nullptr, Hh, ACT_RELU, nullptr, nullptr, nullptr, M);
                    conv_mfma<256, 1>(s, X0, b16[b + ".proj.w"], nullptr, g(b + ".proj.bias"), nullptr, X, ACT_NONE, nullptr, nullptr, nullptr, M);
                } else {
                    conv_mfma<128, 9>(s, ain, b16[b + ".conv1.w"], g(b + ".conv1.scale"), g(b + ".conv1.shift"), nullptr, Hh, ACT_RELU, nullptr, nullptr, nullptr, M);
                    trunk_convs++;
                }
                conv_mfma<128, 9>(s, Hh, b16[b + ".conv2.w"], nullptr, g(b + ".conv2.bias"), X, X, ACT_NONE, sB, tB, o2, M);
                trunk_convs++;
            } else {
                const int cin = first ? SC : F;
                conv_direct(s, ain, g(b + ".conv1.w"), cin, F, 3, g(b + ".conv1.scale"), g(b + ".conv1.shift"), nullptr, Hh, NACT_RELU, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, 0, M);
                if (first) conv_direct(s, X0, g(b + ".proj.w"), cin, F, 1, nullptr, g(b + ".proj.bias"), nullptr, X, NACT_NONE, nullptr, nullptr, nullptr,
                                       nullptr, nullptr, nullptr, 0, M);
                conv_direct(s, Hh, g(b + ".conv2.w"), F, F, 3, nullptr, g(b + ".conv2.bias"), X, X, NACT_NONE, sB, tB, o2, nullptr, nullptr, nullptr, 0, M);
            }
        }
        if (timing) hipEventRecord(e1, s);
        trunk_out = cur; fuse_heads = fuse;
    }
    // (measured and dropped, round 3: the heads on high-priority streams of their own, forked from / joined to the caller's stream — with two game
    // groups 2160 instead of 2312 positions/s: the priority kernels break into the other group's trunk rounds, and two more event hops per wave)
    void forward_heads(hipStream_t s, float* policy, float* value, int n, int) override {
        if (!loaded) return;
        const int M = n * HW;
        bf16_t* const cur = trunk_out; const bool fuse = fuse_heads;
        if (gomoku) {
            if (fuse) {                             // both heads' BN + ReLU + Conv3x3 128 -> 32 + BN + ReLU from the raw trunk output
                Head32Args hh; memset(&hh, 0, sizeof(hh));
                hh.in = cur; hh.wgt[0] = b16["p.c1.w"]; hh.wgt[1] = b16["v.c1.w"];
                hh.s0[0] = g("p.bn0.scale"); hh.t0[0] = g("p.bn0.shift"); hh.s0[1] = g("v.bn0.scale"); hh.t0[1] = g("v.bn0.shift");
                hh.s1[0] = g("p.c1.scale"); hh.t1[0] = g("p.c1.shift"); hh.s1[1] = g("v.c1.scale"); hh.t1[1] = g("v.c1.shift");
                hh.out[0] = PH; hh.out[1] = VH; hh.M = M; hh.H = H; hh.W = W;
                hipLaunchKernelGGL(k_conv_head32, dim3((M + HC_ROWS - 1) / HC_ROWS, 2), dim3(RB3_THREADS), hc_lds_bytes(), s, hh);
            } else {
                hipLaunchKernelGGL(k_affine_relu, dim3((unsigned)(((long)M * F / 8 + 255) / 256)), dim3(256), 0, s, cur, g("v.bn0.scale"), g("v.bn0.shift"), Va, (long)M * F / 8, F);
                conv_mfma32(s, Aa, b16["p.c1.w"], g("p.c1.scale"), g("p.c1.shift"), PH, M);
                conv_mfma32(s, Va, b16["v.c1.w"], g("v.c1.scale"), g("v.c1.shift"), VH, M);
            }
            // the two heads are independent chains of small kernels (policy ~0.22 ms, value ~0.08 ms per 2048 positions): the value head
            // runs on a side stream next to the policy head and joins before the softmax (round 2; GAZ_HEADS_2STREAMS=0 -> one stream)
            static const bool two = !(getenv("GAZ_HEADS_2STREAMS") && atoi(getenv("GAZ_HEADS_2STREAMS")) == 0);
            hipStream_t sv = s;
            if (two) {
                if (!side_stream) { hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking); hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming); hipEventCreateWithFlags(&ev_join, hipEventDisableTiming); }
                hipEventRecord(ev_fork, s); hipStreamWaitEvent(side_stream, ev_fork, 0); sv = side_stream;
            }
            {
                ConvSmallArgs c; c.in = PH; c.w = g("p.c2.w"); c.bias = g("p.c2.bias"); c.flat = pfeat; c.fs = g("p.bn2.scale"); c.ft = g("p.bn2.shift");
                c.act = NACT_RELU; c.M = M; c.H = H; c.W = W;
                hipLaunchKernelGGL((k_conv_small<8, 3>), dim3((M + 255) / 256), dim3(256), 0, s, c);
                c.in = VH; c.w = g("v.c2.w"); c.bias = g("v.c2.bias"); c.flat = vfeat; c.fs = g("v.bn2.scale"); c.ft = g("v.bn2.shift");
                hipLaunchKernelGGL((k_conv_small<4, 1>), dim3((M + 255) / 256), dim3(256), 0, sv, c);
            }
            dense(sv, vfeat, "v.d1.w", g("v.d1.scale"), g("v.d1.shift"), vd1, n, HW * 4, 256, NACT_RELU);
            dense(sv, vd1, "v.d2.w", g("v.d2.scale"), g("v.d2.shift"), vd2, n, 256, 128, NACT_RELU);
            dense(sv, vd2, "v.d3.w", nullptr, g("v.d3.bias"), value, n, 128, 1, NACT_TANH);
            if (two) hipEventRecord(ev_join, side_stream);
            dense(s, pfeat, "p.d1.w", g("p.d1.scale"), g("p.d1.shift"), pd1, n, HW * 8, 512, NACT_RELU);
            dense(s, pd1, "p.d2.w", nullptr, g("p.d2.bias"), plog, n, 512, A, NACT_NONE);
            if (two) hipStreamWaitEvent(s, ev_join, 0);
        } else {
            conv_direct(s, X, g("p.c.w"), F, 8, 1, g("p.c.scale"), g("p.c.shift"), nullptr, nullptr, 0, nullptr, nullptr, nullptr, pfeat, nullptr, nullptr, NACT_NONE, M);
            conv_direct(s, X, g("v.c.w"), F, 4, 1, g("v.c.scale"), g("v.c.shift"), nullptr, nullptr, 0, nullptr, nullptr, nullptr, vfeat, nullptr, nullptr, NACT_NONE, M);
            dense(s, pfeat, "p.d1.w", nullptr, g("p.d1.bias"), pd1, n, HW * 8, 128, NACT_RELU);
            dense(s, pd1, "p.d2.w", nullptr, g("p.d2.bias"), pd2, n, 128, 64, NACT_NONE);
            dense(s, pd2, "p.d3.w", nullptr, g("p.d3.bias"), plog, n, 64, A, NACT_NONE);
            dense(s, vfeat, "v.d1.w", nullptr, g("v.d1.bias"), vd1, n, HW * 4, 128, NACT_NONE);
            dense(s, vd1, "v.d2.w", nullptr, g("v.d2.bias"), vd2, n, 128, 64, NACT_RELU);
            dense(s, vd2, "v.d3.w", nullptr, g("v.d3.bias"), value, n, 64, 1, NACT_TANH);
        }
        hipLaunchKernelGGL(k_softmax_rows, dim3(n), dim3(64), 0, s, plog, policy, n, A, logits);
    }
    void timing_reset() override { for (auto e : tev) hipEventDestroy(e); tev.clear(); }
    void timing_get(double* ms, int64_t* launches) override {
        double t = 0;
        for (size_t i = 0; i + 1 < tev.size(); i += 2) { float a = 0; hipEventElapsedTime(&a, tev[i], tev[i + 1]); t += a; }
        // launches of the dominant kernel inside the bracket: fused blocks (k_resblock3) or 128 -> 128 convs; block 0 of the
        // Gomoku net (256 -> 128 + projection) rides in the same bracket and is counted as one more launch-equivalent
        *ms = t; *launches = (int64_t)(tev.size() / 2) * (one_launch ? 1 : (fused_blocks > 0 ? fused_blocks + (block0_fused ? 0 : 1) : (trunk_convs > 0 ? trunk_convs : 1)));
    }
    const char* dominant_kernel(int n, double* flops) override {
        const double conv = 2.0 * (double)n * HW * 128.0 * 1152.0;
        if (!gomoku) { *flops = 0; return ""; }
        const bool fz = fused && blocks > 1;
        *flops = fz ? 2 * conv : conv;
        if (fz && trunk && trunk_w && trunk_m16 && block0_in_trunk && trunk_w0 && HW <= 256) {
            // priced per LAUNCH: the one kernel runs block 0 (3x3 256->128 = two 128-channel convolutions, 3x3 128->128, the 1x1
            // projection 256->128) and blocks - 1 regular blocks of two 3x3 convolutions each
            *flops = (2.0 * (blocks - 1) + 3.0) * conv + 2.0 * (double)n * HW * 256.0 * 128.0;
            return "k_trunk<8 waves, B0> (the whole Gomoku trunk in ONE launch per forward: block 0 with its 256-channel input and 1x1 projection + the other residual blocks, "
                   "every 3x3 / 1x1 convolution of them counted; implicit GEMM on v_mfma_f32_16x16x32_bf16, activations resident in LDS)";
        }
        if (fz && trunk && trunk_w) return trunk_m16 ? "k_trunk<RESG, 8 waves> (blocks 1.. in one launch, priced per residual block: two 3x3 convs 128->128, implicit GEMM on v_mfma_f32_16x16x32_bf16)"
                                                     : "k_trunk<4,2,RESG> (blocks 1.. in one launch, priced per residual block: two 3x3 convs 128->128, implicit GEMM on v_mfma_f32_32x32x16_bf16)";
        return fz ? "k_resblock3 (whole residual block: two 3x3 convs 128->128, implicit GEMM on MFMA 32x32x16 bf16)"
                  : "k_conv3x3<128,128> (trunk 3x3 conv, implicit GEMM on MFMA 32x32x16 bf16)";
    }
};

static Evaluator* make_generic_evaluator(const gaz_engine_config& cfg, int H, int W, int C, int A, bool gomoku, std::string* err) {
    if (gomoku && cfg.net_filters != 128) { *err = "Gomoku net: net_filters must be 128"; return nullptr; }
    if (!gomoku && cfg.net_filters != 64 && cfg.net_filters != 0 && cfg.net_filters != 128) { *err = "TicTacToe net uses 64 filters"; return nullptr; }
    if (cfg.net_blocks < 1 || cfg.net_blocks > 64) { *err = "net_blocks out of range"; return nullptr; }
    GenericEvaluator* e = new GenericEvaluator();
    e->H = H; e->W = W; e->C = C; e->A = A; e->HW = H * W; e->blocks = cfg.net_blocks; e->F = gomoku ? 128 : 64; e->nmax = cfg.n_games;
    e->logits = cfg.policy_is_logits; e->gomoku = gomoku;
    e->fused = !(getenv("GAZ_FUSED") && atoi(getenv("GAZ_FUSED")) == 0);
    e->trunk = !(getenv("GAZ_TRUNK") && atoi(getenv("GAZ_TRUNK")) == 0);
    e->trunk_m16 = !(getenv("GAZ_TRUNK_M16") && atoi(getenv("GAZ_TRUNK_M16")) == 0);
    e->block0_in_trunk = !(getenv("GAZ_BLOCK0_IN_TRUNK") && atoi(getenv("GAZ_BLOCK0_IN_TRUNK")) == 0);
    e->stem_in_trunk_ok = getenv("GAZ_STEM_IN_TRUNK") && atoi(getenv("GAZ_STEM_IN_TRUNK")) != 0;
    { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev); if (hipGetDeviceProperties(&pr, dev) == hipSuccess) e->n_cus = pr.multiProcessorCount; }
    const size_t M = (size_t)cfg.n_games * e->HW, SC = gomoku ? 256 : 128, F = e->F, n = cfg.n_games;
    e->X0 = e->dalloc<bf16_t>(M * SC); e->A0 = e->dalloc<bf16_t>(M * SC); e->X = e->dalloc<bf16_t>(M * F); e->Aa = e->dalloc<bf16_t>(M * F);
    e->Hh = e->dalloc<bf16_t>(M * F); e->Va = e->dalloc<bf16_t>(M * F); e->PH = e->dalloc<bf16_t>(M * 32); e->VH = e->dalloc<bf16_t>(M * 32);
    e->pfeat = e->dalloc<float>(n * e->HW * 8); e->vfeat = e->dalloc<float>(n * e->HW * 8); e->pd1 = e->dalloc<float>(n * 512); e->vd1 = e->dalloc<float>(n * 256);
    e->pd2 = e->dalloc<float>(n * 128); e->vd2 = e->dalloc<float>(n * 128); e->plog = e->dalloc<float>(n * 256);
    if (!e->X0 || !e->A0 || !e->X || !e->Aa || !e->Hh || !e->Va || !e->PH || !e->VH || !e->pfeat || !e->vfeat || !e->pd1 || !e->vd1 || !e->pd2 || !e->vd2 || !e->plog) {
        *err = "hipMalloc failed"; delete e; return nullptr;
    }
    hipFuncSetAttribute((const void*)(k_block0<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb3_lds_bytes<4>());
    hipFuncSetAttribute((const void*)k_conv_head32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hc_lds_bytes());
    hipFuncSetAttribute((const void*)(k_trunk<4, 2, 4, 2, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(256, true));
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, false, false, false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(256));
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, false, false, false, true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(256));
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, false, false, false, true, 8, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes_s0());
    hipFuncSetAttribute((const void*)(k_resblock3<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb3_lds_bytes<4>());
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 128, 256, 4, 2, 2, 2, 1, 1, 0, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<256, 128, 128, 2, 2, 2, 2, 4, 1, 0, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<256, 128, 128, 2, 2, 2, 2, 4, 1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 128, 256, 4, 2, 2, 2, 1, 1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 32, 256, 8, 1, 1, 1, 1, 1, 0, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_dense, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    return e;
}

Evaluator* make_resnet_evaluator(const gaz_engine_config& cfg, int H, int W, int C, int A, std::string* err) {
    if (H == 15 && W == 15) return make_generic_evaluator(cfg, H, W, C, A, true, err);
    if (H == 3 && W == 3) return make_generic_evaluator(cfg, H, W, C, A, false, err);
    if (!(H == 6 && W == 7 && C == 4)) { *err = "no network is defined for this board"; return nullptr; }
    if (cfg.net_filters != 128) { *err = "net_filters must be 128"; return nullptr; }
    if (cfg.net_blocks < 1 || cfg.net_blocks > 64) { *err = "net_blocks out of range"; return nullptr; }
    ResNetEvaluator* e = new ResNetEvaluator();
    e->H = H; e->W = W; e->C = C; e->A = A; e->HW = H * W; e->blocks = cfg.net_blocks; e->filters = 128; e->nmax = cfg.n_games;
    e->logits = cfg.policy_is_logits;
    const size_t M = (size_t)cfg.n_games * e->HW;
    e->X2 = e->dalloc<bf16_t>(M * 128 + 1024); e->fused = !(getenv("GAZ_FUSED") && atoi(getenv("GAZ_FUSED")) == 0);
    e->trunk = !(getenv("GAZ_TRUNK") && atoi(getenv("GAZ_TRUNK")) == 0);
    e->trunk_whole = !(getenv("GAZ_TRUNK_WHOLE") && atoi(getenv("GAZ_TRUNK_WHOLE")) == 0);
    e->trunk_mix = !(getenv("GAZ_TRUNK_MIX") && atoi(getenv("GAZ_TRUNK_MIX")) == 0);
    e->trunk_m16 = !(getenv("GAZ_TRUNK_M16") && atoi(getenv("GAZ_TRUNK_M16")) == 0);
    { hipDeviceProp_t pr; int dev = 0; hipGetDevice(&dev); if (hipGetDeviceProperties(&pr, dev) == hipSuccess) e->n_cus = pr.multiProcessorCount; }
    e->X = e->dalloc<bf16_t>(M * 128 + 1024); e->Aa = e->dalloc<bf16_t>(M * 128 + 1024); e->Hh = e->dalloc<bf16_t>(M * 128 + 1024);
    e->pfeat = e->dalloc<float>((size_t)cfg.n_games * e->HW * 8); e->vfeat = e->dalloc<float>((size_t)cfg.n_games * e->HW * 8);
    e->pd1 = e->dalloc<float>((size_t)cfg.n_games * 128); e->vd1 = e->dalloc<float>((size_t)cfg.n_games * 128);
    if (!e->X || !e->Aa || !e->Hh || !e->pfeat || !e->vfeat || !e->pd1 || !e->vd1) { *err = "hipMalloc failed"; delete e; return nullptr; }
    // dynamic LDS above 64 KB needs the attribute
    hipFuncSetAttribute((const void*)k_resblock, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_resblock3<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb3_lds_bytes<4>());
    hipFuncSetAttribute((const void*)k_conv_heads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hc_lds_bytes());
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_trunk_mix<8, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_trunk_mix<8, 2, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_trunk_mix<8, 2, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_trunk<2, 2, 8, 2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes(128));
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 128, 256, 4, 2, 2, 2, 1, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 128, 128, 2, 2, 2, 2, 2, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

}  // namespace gaz

// =====================================================================================================================
// Fused tree + trunk launch — ONE launch per simulation wave for the headline path: the PUCT tree step of every Connect4 game AND the trunk kernel
// (stem + all residual blocks + the heads' first convolution) of the leaf rows those games write.
//
// Why: the tree kernel's duration is its slowest games (a dependent chain of HBM round trips per simulation: ~27 us for the mean
// game, 65 us for the launch), and while it runs the matrix cores idle; launched as separate kernels the trunk cannot start before
// the last game is done.  Here the first blocks of the grid are tree blocks (4 waves x four 16-lane teams = 16 games each) and the
// rest are trunk workgroups; workgroups are dispatched in index order, so every tree block is resident before any trunk workgroup
// exists, and a trunk workgroup only waits (sleeping) for the done flags of ITS three boards: the trunk starts on the games that are
// ready while the slow ones finish.  A tree block shares its CU with one trunk workgroup; when it exits, the second one moves in.
// The hand-over of a leaf row from a team to a workgroup on another CU / XCD inside a running kernel goes through system-scope
// stores and loads (puct_core.hpp store_coherent / publish_done, trunk.hpp TrunkArgs::ready) — no cache flush, no atomics on the tree.
// Results are bit-identical to separate launches (same device functions; tests/test_evaluator_gpu.py).
namespace gaz {

typedef TeamGame<GAME_C4> GP4;

// The tree role inlined or as a function call (-DGAZ_TREE_ROLE_ATTR='__attribute__((noinline))').  Inlined, its scalar-register pressure (337
// spilled SGPRs, parked in lanes of 6 VGPRs that are then reserved for the whole kernel) costs the trunk role of the edge-tile variant a few
// spilled VGPRs (256 + 7, none inside the tap loop: +1.4 % on a workgroup's compute time); as a call the trunk role is spill-free but the TREE
// role — the launch's critical path — slows down by a quarter (callee-saved registers and spills go through scratch memory: tree waves 53 -> 66
// us at the median, fused launch 509 -> 541 us, same box, tools/ab_fused.sh, gpurun_out r03 ab1).  Measured, inlined wins by 4 %: the default.
#ifndef GAZ_TREE_ROLE_ATTR
#define GAZ_TREE_ROLE_ATTR __forceinline__
#endif
template <class GP, class LOCAL, bool GUMBEL, int THREADS = TR_THREADS, size_t LDS_BYTES = trunk_lds_bytes(96)>
__device__ GAZ_TREE_ROLE_ATTR void tree_role(const DevParams<GP>& E, int g0, int g1, uint4* lds) {
    constexpr int PER = WAVE / GP::TEAM, NT = (THREADS / WAVE) * PER;
    // (measured and dropped: s_setprio 3 for the tree waves — no change, 60.8 k vs 60.8 k / 339.7 k vs 339.6 k: they wait on memory round trips, not on issue slots)
    Scratch<GP>* S = reinterpret_cast<Scratch<GP>*>(lds);
    LOCAL* L = reinterpret_cast<LOCAL*>(S + NT);
    uint32_t* rank = reinterpret_cast<uint32_t*>(L + NT);           // completion queue: how many games of this block have finished (DevParams::done_queue)
    static_assert(NT * (sizeof(Scratch<GP>) + sizeof(LOCAL)) + 16 <= LDS_BYTES, "tree scratch must fit the trunk's LDS allocation");
    if (E.done_queue) {
        if (threadIdx.x == 0) *rank = 0;
        __syncthreads();
    }
    // A tree block steps queue_gpb games in rounds of NT (completion queue only; otherwise one round): fewer, longer-lived tree blocks hold
    // fewer of the launch's slots, so more trunk workgroups are resident from the start — and with the queue they find finished games at once.
    const int w = threadIdx.x >> 6, t = team_in_wave<GP>(), i = w * PER + t;
    const int per_block = E.done_queue ? E.queue_gpb : NT;
    for (int r = 0; r * NT < per_block; ++r) {
        const int g = g0 + (int)blockIdx.x * per_block + r * NT + w * PER + t;
        if (g < g1) {
            if constexpr (GUMBEL) g_game_step<GP>(E, g, S[i], L[i], rank, (int)blockIdx.x);
            else game_step<GP>(E, g, S[i], L[i], rank, (int)blockIdx.x);
        }
    }
}

template <bool MIX, bool SKIP, class GP, class LOCAL, bool GUMBEL>
__device__ __forceinline__ void wave_trunk_body(const DevParams<GP>& E, int g0, int g1, int n_tree_blocks, const TrunkArgs& a) {
    if ((int)blockIdx.x < n_tree_blocks) {
        // ---- tree role: wave w of the block steps games [(4 b + w) PER, +PER); its scratch lives in the launch's dynamic LDS
        extern __shared__ uint4 lds[];
        const DevParams<GP> El = E;                 // the callee takes an address: copy the kernel argument to the stack HERE, not at kernel entry for every workgroup
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(size_t)blockIdx.x * 128 + (threadIdx.x >> 6)] = wall_clock64();            // GAZ_FUSED_STAMPS: wave w starts at [w] ...
        tree_role<GP, LOCAL, GUMBEL>(El, g0, g1, lds);
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(size_t)blockIdx.x * 128 + 4 + (threadIdx.x >> 6)] = wall_clock64();        // ... and ends at [4 + w]
        return;
    }
    // ---- trunk role: exactly k_trunk_mix / k_trunk of trunk.hpp on workgroup index bid
    const int bid = (int)blockIdx.x - n_tree_blocks;
    if (MIX && bid >= a.n_big) trunk_tile<3, 4, 8, true, true, false, true, 4, false, SKIP ? 2 : 0>(a, (long)a.n_big * a.tile_rows + (long)(bid - a.n_big) * a.small_rows, a.small_rows, SKIP ? a.perm_small : nullptr, a.boff_small);
    else trunk_tile<2, 2, 8, true, true, false, true, 4, false, SKIP ? 1 : 0>(a, (long)bid * a.tile_rows, a.tile_rows, SKIP ? a.perm : nullptr, a.boff);
}

template <bool MIX, bool SKIP> __global__ __launch_bounds__(TR_THREADS, 2) void k_wave_trunk(DevParams<GP4> E, int g0, int g1, int n_tree_blocks, TrunkArgs a) {
    wave_trunk_body<MIX, SKIP, GP4, PuctLocal<GP4>, false>(E, g0, g1, n_tree_blocks, a);
}

// Gomoku (BASELINE configs[3]): one game per wavefront, eight per 512-thread tree block and round; the trunk role is the 8-wave launch with block 0
// AND the 256-channel stem inside (trunk.hpp S0), one board = one queue entry per workgroup, one workgroup per CU.  The leaf rows of a 15 x 15 x 2
// board are 450 bytes — not dword-aligned from game to game — so this hand-over uses the release / acquire form (MI355X_MICROARCH.md, Valid
// forms): the game's wave writes its row with plain stores, waits for them, releases at agent scope (puct_core.hpp publish_done,
// DevParams::handoff_release) and then writes its queue entry; the workgroup's polling lane acquires before the barrier, the planes are
// read with plain loads behind it.  A tree step here is hundreds of microseconds, the two fences a few.
typedef Game<GAME_GMK> GGMK;
__global__ __launch_bounds__(512, 1) void k_wave_trunk_gmk(DevParams<GGMK> E, int g0, int g1, int n_tree_blocks, TrunkArgs a) {
    if ((int)blockIdx.x < n_tree_blocks) {
        extern __shared__ uint4 lds[];
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(size_t)blockIdx.x * 128 + (threadIdx.x >> 6)] = wall_clock64();
        tree_role<GGMK, PuctLocal<GGMK>, false, 512, trunk_lds_bytes_s0()>(E, g0, g1, lds);
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(size_t)blockIdx.x * 128 + 8 + (threadIdx.x >> 6)] = wall_clock64();
        return;
    }
    const int bid = (int)blockIdx.x - n_tree_blocks;
    trunk_tile<2, 2, 8, false, false, false, true, 8, true, 0, true>(a, (long)bid * a.tile_rows, a.tile_rows, nullptr, 0);
}

bool launch_wave_trunk_gmk(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan) {
    if (!plan || !dev_params) return false;
    const TrunkLaunchPlan& P = *static_cast<const TrunkLaunchPlan*>(plan);
    const DevParams<GGMK>& E = *static_cast<const DevParams<GGMK>*>(dev_params);
    const int per_block = E.done_queue ? E.queue_gpb : 8;
    const int n_tree = (g1 - g0 + per_block - 1) / per_block;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)k_wave_trunk_gmk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trunk_lds_bytes_s0()); attr = true; }
    hipLaunchKernelGGL(k_wave_trunk_gmk, dim3(n_tree + P.nwg), dim3(512), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    return true;
}

// The Gumbel search (MCTS_Gumbel.py:562-679) in the same launch shape.  TEAMS: four games per wavefront (16 games per tree block, as the PUCT
// launch) or one game per wavefront (4 per tree block: less lock-step, but four times the tree blocks ahead of the trunk workgroups).
template <bool MIX, bool SKIP, bool TEAMS> __global__ __launch_bounds__(TR_THREADS, 2) void k_wave_trunk_gumbel(DevParams<Game<GAME_C4>> E, int g0, int g1, int n_tree_blocks, TrunkArgs a) {
    if constexpr (TEAMS) wave_trunk_body<MIX, SKIP, GP4, GumbelLocal<GP4>, true>(*reinterpret_cast<const DevParams<GP4>*>(&E), g0, g1, n_tree_blocks, a);
    else wave_trunk_body<MIX, SKIP, Game<GAME_C4>, GumbelLocal<Game<GAME_C4>>, true>(E, g0, g1, n_tree_blocks, a);
}

bool launch_wave_trunk_c4_gumbel(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan) {
    if (!plan || !dev_params) return false;
    const TrunkLaunchPlan& P = *static_cast<const TrunkLaunchPlan*>(plan);
    const DevParams<Game<GAME_C4>>& E = *static_cast<const DevParams<Game<GAME_C4>>*>(dev_params);
    static const bool teams = !(getenv("GAZ_FUSE_GUMBEL_TEAMS") && atoi(getenv("GAZ_FUSE_GUMBEL_TEAMS")) == 0);
    const int per_block = E.done_queue ? E.queue_gpb : (TR_THREADS / WAVE) * (teams ? WAVE / GP4::TEAM : 1);
    const int n_tree = (g1 - g0 + per_block - 1) / per_block;
    const bool skip = P.args.perm && P.args.perm_small;
    if (!P.mix || !skip) return false;              // only the headline trunk variant is built for this launch
    if (teams) hipLaunchKernelGGL((k_wave_trunk_gumbel<true, true, true>), dim3(n_tree + P.nwg), dim3(TR_THREADS), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    else hipLaunchKernelGGL((k_wave_trunk_gumbel<true, true, false>), dim3(n_tree + P.nwg), dim3(TR_THREADS), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    return true;
}

bool launch_wave_trunk_c4(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan) {
    if (!plan || !dev_params) return false;
    const TrunkLaunchPlan& P = *static_cast<const TrunkLaunchPlan*>(plan);
    const DevParams<GP4>& E = *static_cast<const DevParams<GP4>*>(dev_params);
    constexpr int GAMES_PER_BLOCK = (TR_THREADS / WAVE) * (WAVE / GP4::TEAM);
    const int per_block = E.done_queue ? E.queue_gpb : GAMES_PER_BLOCK;
    const int n_tree = (g1 - g0 + per_block - 1) / per_block;
    const bool skip = P.args.perm && P.args.perm_small;       // both tile shapes carry an edge-tile permutation (checked by the evaluator)
    if (P.mix && skip) hipLaunchKernelGGL((k_wave_trunk<true, true>), dim3(n_tree + P.nwg), dim3(TR_THREADS), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    else if (P.mix) hipLaunchKernelGGL((k_wave_trunk<true, false>), dim3(n_tree + P.nwg), dim3(TR_THREADS), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    else hipLaunchKernelGGL((k_wave_trunk<false, false>), dim3(n_tree + P.nwg), dim3(TR_THREADS), P.lds_bytes, s, E, g0, g1, n_tree, P.args);
    return true;
}

}  // namespace gaz
