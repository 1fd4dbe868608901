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
    int halo, tile_rows;                          // k_resblock3: W + 1 and 64 TM - 2 (W + 1); or 0 and k * H * W when a tile is k whole boards
    unsigned long long* stamps;                   // diagnostic: [workgroup][RB_STAMPS]: 32 wall-clock ticks (100 MHz) + 32 shader-clock counts of wave 0, or null
};

constexpr int RB_ROWS = 256, RB_THREADS = 512, RB_STAMPS = 64;
#define RB_STAMP(i) do { if (a.stamps && tid == 0) { a.stamps[(size_t)blockIdx.x * RB_STAMPS + (i)] = wall_clock64(); a.stamps[(size_t)blockIdx.x * RB_STAMPS + 32 + (i)] = clock64(); } } while (0)

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
    RB_STAMP(0);

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
    // per-thread parameters, fetched while the DMA is in flight (a load inside a later loop costs an L2 round trip there).
    // Transform: slot i = tid + 512 * it keeps sp and (lr & 15), so one thread always works on the same 8 channels.
    const int tch0 = ((tid % SLOTS) ^ ((tid / SLOTS) & 15)) * 8;
    float ps1[8], pt1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ps1[j] = a.s1[tch0 + j]; pt1[j] = a.t1[tch0 + j]; }
    float ps2[TN], pt2[TN], pb2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { const int col = (wn * TN + tn) * 32 + l31; ps2[tn] = a.s2[col]; pt2[tn] = a.t2[col]; pb2[tn] = a.b2[col]; }

    __syncthreads();                                // x image + slice 0 landed
    RB_STAMP(1);

    // ---- 2. in-place pre-activation of the image: relu(x * s1 + t1)
    for (int i = tid; i < n_aslots; i += RB_THREADS) {
        uint4 v = As[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = fmaxf(__uint_as_float(w[j] << 16) * ps1[2 * j] + pt1[2 * j], 0.0f);
            const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * ps1[2 * j + 1] + pt1[2 * j + 1], 0.0f);
            w[j] = pack_bf16(lo, hi);
        }
        As[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    RB_STAMP(2);

    f32x16 acc[TM][TN];
    constexpr int EPI_IT = RB_ROWS / (RB_THREADS / 16);
    uint4 resv[EPI_IT];
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
            RB_STAMP(3 + sl);
        }
        if (conv == 0) {
            // ---- 4. h = relu(acc * s2 + t2) as bf16 into the image region (row j of h at image row j, same swizzle)
            bf16_t* Hs = reinterpret_cast<bf16_t*>(As);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = (wn * TN + tn) * 32 + l31;
                    const float s2 = ps2[tn], t2 = pt2[tn];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float v = fmaxf(acc[tm][tn][r] * s2 + t2, 0.0f);
                        Hs[row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))] = (bf16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
                }
            __syncthreads();
            RB_STAMP(21);
            // residual rows of this thread's epilogue slots: issued now, consumed after conv2 (L2 round trips hidden)
#pragma unroll
            for (int it = 0; it < EPI_IT; ++it) {
                long gr = m0 + tid / 16 + it * (RB_THREADS / 16);
                gr = gr < a.M ? gr : (long)a.M - 1;
                resv[it] = *reinterpret_cast<const uint4*>(a.xin + (size_t)gr * BN + (tid % 16) * 8);
            }
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
            const float tA = pb2[tn];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                Ct[row * CT + col] = acc[tm][tn][r] + tA;
            }
        }
    __syncthreads();
    RB_STAMP(22);
    const int chunk = tid % 16, r0 = tid / 16;
#pragma unroll
    for (int it = 0; it < EPI_IT; ++it) {
        const int row = r0 + it * (RB_THREADS / 16);
        const long gr = m0 + row;
        if (row >= bmo || gr >= a.M) break;
        const float4 c0 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8 + 4]);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const size_t o = (size_t)gr * BN + chunk * 8;
        const unsigned rw[4] = {resv[it].x, resv[it].y, resv[it].z, resv[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(rw[j] << 16); v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u); }
        *reinterpret_cast<uint4*>(a.xout + o) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
    }
    RB_STAMP(23);
}

// ---- k_resblock3: the same block with the weights OUT of LDS and barrier-free taps, in 256-thread workgroups, TWO per CU.
// Stamps of k_resblock (tools/rb_stamps.py): a tap took 3460 cycles against 2048 of pure MFMA — ~900 went to issuing the
// weight LDS-DMA (all eight waves at once, right after the barrier), ~500 to the per-tap barrier — and 37 % of a workgroup's
// life (image DMA wait, transform, h write, epilogue) left the matrix cores idle because one 8-wave workgroup owned the CU.
// Here each wave loads ITS B fragments straight from L2 into registers, the image is read-only during a conv (no barrier in
// the tap loop), and a workgroup is 4 waves (one per SIMD), each 128 rows x 64
// channels (TM = 4, TN = 2: 8 MFMAs per 4 LDS fragment reads); LDS is the 74-KB image only, so two workgroups share a CU
// and one's serial phases hide behind the other's MFMAs.  B fragments: a four-k-step register ring fed from L2.
// TM picks the tile height (ROWS = 64 * TM image rows computed, ROWS - 2(W+1) of them valid outputs): the host takes the TM
// whose tile count fills whole rounds of 2 workgroups x 256 CUs best (Connect4, 4096 games: TM = 3 -> 978 tiles = 1.91 rounds).
constexpr int RB3_THREADS = 256;
template <int TM> constexpr int rb3_rows() { return 64 * TM; }
template <int TM> constexpr size_t rb3_lds_bytes() { return (size_t)(rb3_rows<TM>() + 2 * CONV_HALO_MAX + 1) * 256; }
template <int TM, int RING>
__global__ __launch_bounds__(RB3_THREADS, 2) void k_resblock3(ResBlockArgs a) {
    constexpr int BN = 128, SLOTS = 16, TN = 2, KS = 8, ROWS = rb3_rows<TM>();
    constexpr int AROWS = ROWS + 2 * CONV_HALO_MAX + 1, ZROW = AROWS - 1, BSL = BN * SLOTS;
    extern __shared__ uint4 lds[];
    uint4* As = lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    // halo mode: h = W + 1 rows of context on both sides, conv1 recomputes them; board-aligned mode (tile = whole boards): every
    // in-board tap stays inside the tile and the out-of-board ones are masked, so no halo is loaded or recomputed at all (h = 0)
    const int h = a.halo, HW = a.H * a.W, bmo = a.tile_rows;
    const long m0 = (long)blockIdx.x * bmo;
    const uint4* in4 = reinterpret_cast<const uint4*>(a.xin);
    const int col0 = wn * 64 + l31;                 // + 32 * tn
    RB_STAMP(0);

    // ---- 1. image of x: image row q <-> global row m0 - 2h + q
    const int n_aslots = (ROWS + 2 * h) * SLOTS;
    for (int base = wave * 64; base < n_aslots; base += RB3_THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 - 2 * h + lr;
        gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
    }
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);

    // B ring: fragment (global k-step g = sl * 8 + ks, tn) lives in bfr[g % RING][tn]
    // Buffer loads: one VGPR (the lane part of the address) serves every fragment, the slice / k-step part is scalar.
    // The host puts conv2's weights right behind conv1's, so the 18 slices are one array.
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, 0, 18 * BSL * 16, 0x00020000);
    const int bvo = (lhi * BN + col0) * 16;
    auto ldb = [&](int slice, int ks, int tn) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo, ((slice * BSL) + ks * 2 * BN + tn * 32) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 bfr[RING][TN];
#pragma unroll
    for (int g = 0; g < RING; ++g)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[g][tn] = ldb(0, g, tn);
    const int tch0 = ((tid % SLOTS) ^ ((tid / SLOTS) & 15)) * 8;
    float ps1[8], pt1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ps1[j] = a.s1[tch0 + j]; pt1[j] = a.t1[tch0 + j]; }
    float ps2[TN], pt2[TN], pb2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { ps2[tn] = a.s2[col0 + tn * 32]; pt2[tn] = a.t2[col0 + tn * 32]; pb2[tn] = a.b2[col0 + tn * 32]; }

    int lrow[TM]; unsigned vmask[TM];               // bits 0-8: conv1 taps, 9-17: conv2 taps
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wm * TM + tm) * 32 + l31;
        unsigned mm = 0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const long gr = m0 + lrow[tm] - (which == 0 ? h : 0);
            if (gr >= 0 && gr < a.M) {
                const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << (which * 9 + t);
                }
            }
        }
        vmask[tm] = mm;
    }
    __syncthreads();                                // x image landed
    RB_STAMP(1);

    // ---- 2. in-place pre-activation of the image: relu(x * s1 + t1)
    for (int i = tid; i < n_aslots; i += RB3_THREADS) {
        uint4 v = As[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = fmaxf(__uint_as_float(w[j] << 16) * ps1[2 * j] + pt1[2 * j], 0.0f);
            const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * ps1[2 * j + 1] + pt1[2 * j + 1], 0.0f);
            w[j] = pack_bf16(lo, hi);
        }
        As[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    RB_STAMP(2);

    // The tap loops are real loops (code ~10 KB): fully unrolled, the 18 taps are ~60 KB of straight-line code that every
    // wave streams through the instruction cache exactly once.  Ring slot = ks % RING is then tap-independent.
    static_assert(KS % RING == 0, "ring slot must not depend on the tap");
    f32x16 acc[TM][TN];
#pragma unroll                                      // two copies of the tap loop: keeps the h-write address math out of any loop
    for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
        // Image fragments: NB register buffers, k-step ks in afr[ks % NB], fetched NB - 1 k-steps ahead and across the tap boundary;
        // __builtin_amdgcn_sched_barrier pins each k-step's reads, MFMAs and weight loads where they are written (left alone, the
        // scheduler sinks every ds_read to just before its MFMA and all the ring's loads to the end of the tap: see trunk.hpp).
        constexpr int NB = TM >= 4 ? 2 : 4, PD = NB - 1;
        const char* Ab = reinterpret_cast<const char*>(As);
        int pb[TM], pbn[TM];                        // byte address of k-slot 0 of this lane's image row: row * 256 | ((lhi ^ row & 15) << 4)
        auto tap_rows = [&](int tap, int (&o)[TM]) {
            const int ty = tap / 3, off = (ty - 1) * a.W + (tap - ty * 3 - 1);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bool ok = (vmask[tm] >> (conv * 9 + tap)) & 1u;
                const int ar = ok ? lrow[tm] + h + off : ZROW;
                o[tm] = ar * 256 + ((lhi ^ (ar & 15)) << 4);
            }
        };
        uint4 afr[NB][TM];
        tap_rows(0, pb);
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) afr[d][tm] = *reinterpret_cast<const uint4*>(Ab + (pb[tm] ^ (d * 32)));
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int sl = conv * 9 + tap;
            const int nsl = sl + 1 < 18 ? sl + 1 : sl;
            tap_rows(tap < 8 ? tap + 1 : 8, pbn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    afr[(ks + PD) % NB][tm] = *reinterpret_cast<const uint4*>(Ab + ((ks + PD < KS ? pb[tm] : pbn[tm]) ^ (((ks + PD) % KS) * 32)));
                __builtin_amdgcn_sched_barrier(0);          // reads first: moved behind the MFMAs they lose a k-step of their lead
                bf16x8 bf[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<bf16x8*>(&bfr[ks % RING][tn]);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&afr[ks % NB][tm]), bf[tn], acc[tm][tn], 0, 0, 0);
                // refill this ring slot with k-step ks + RING (of this tap, or of the next one)
                if (ks + RING < KS) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(sl, ks + RING, tn);
                } else {                            // the last tap re-reads its own slice: harmless, keeps the loop branch-free
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(nsl, ks + RING - KS, tn);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) pb[tm] = pbn[tm];
            RB_STAMP(3 + sl);
        }
        if (conv == 0) {
            __syncthreads();                        // every wave is done with the x image
            // ---- 4. h = relu(acc * s2 + t2) as bf16 into the image region (row j of h at image row j, same swizzle)
            bf16_t* Hs = reinterpret_cast<bf16_t*>(As);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const int col = col0 + tn * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float v = fmaxf(acc[tm][tn][r] * ps2[tn] + pt2[tn], 0.0f);
                        Hs[row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))] = (bf16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
                    }
                }
            __syncthreads();
            RB_STAMP(21);
        }
    }
    // residual rows of all this thread's epilogue slots: one L2 round trip, hidden behind the barrier and the tile writes
    // (the B ring and the A fragments are dead by now, so the registers are there)
    uint4 resv[4 * TM];
#pragma unroll
    for (int q = 0; q < 4 * TM; ++q) {
        long gr = m0 + (q / (2 * TM)) * (ROWS / 2) + tid / 16 + (q % (2 * TM)) * (RB3_THREADS / 16);
        gr = gr < a.M ? gr : (long)a.M - 1;
        resv[q] = *reinterpret_cast<const uint4*>(a.xin + (size_t)gr * BN + (tid % 16) * 8);
    }
    __syncthreads();                                // every wave is done with the h image

    // ---- 6. epilogue, half the rows at a time through an fp32 tile over the image region: + bias + residual, rows [m0, m0 + bmo)
    constexpr int CT = BN + 4, HR = ROWS / 2;
    float* Ct = reinterpret_cast<float*>(lds);
    static_assert((size_t)HR * CT * 4 <= (size_t)AROWS * SLOTS * 16, "epilogue tile must fit in the image region");
    const int chunk = tid % 16, r0 = tid / 16;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        Ct[row * CT + col0 + tn * 32] = acc[tm][tn][r] + pb2[tn];
                    }
        }
        __syncthreads();
        if (pass == 0) RB_STAMP(22);
#pragma unroll
        for (int q = 0; q < 2 * TM; ++q) {          // 2 * TM rows per thread and pass
            const int rl = r0 + q * (RB3_THREADS / 16), row = pass * HR + rl;
            const long gr = m0 + row;
            if (row < bmo && gr < a.M) {
                const float4 c0 = *reinterpret_cast<const float4*>(&Ct[rl * CT + chunk * 8]);
                const float4 c1 = *reinterpret_cast<const float4*>(&Ct[rl * CT + chunk * 8 + 4]);
                float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const size_t o = (size_t)gr * BN + chunk * 8;
                const uint4 rq = resv[pass * 2 * TM + q];
                const unsigned rw[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] += __uint_as_float(rw[j] << 16); v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u); }
                *reinterpret_cast<uint4*>(a.xout + o) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            }
        }
        if (pass == 0) __syncthreads();
    }
    RB_STAMP(23);
}

// ---- k_conv_heads: the first convolution of both heads (Connect4/Build_Model.py:41,62: two 3x3 convs 128 -> 8, fused into one
// 128 -> 16 GEMM, padded to a 32-column MFMA tile) in the style of k_resblock3: image by LDS-DMA, weights from L2 into a register
// ring, no barrier in the tap loop, two 256-thread workgroups per CU.  A wave owns 64 rows x the 32 columns; the epilogue applies
// the flat-feature BN + ReLU of each head and writes fp32 [B][HW * 8] per head (EPI = 1 of k_conv3x3, which this replaces: that
// kernel spent 40 us per forward in per-tap barriers and weight DMA for 6 us of MFMA work).
struct HeadsConvArgs {
    const bf16_t* in; const bf16_t* wgt;          // [M][128]; fragment order [9][8 k-steps][2][32 cout][8]
    const float* bias;                            // [32]
    const float* p_fs; const float* p_ft; const float* v_fs; const float* v_ft;   // [HW * 8] flat BN scale / shift
    float* p_feat; float* v_feat;                 // [B][HW * 8]
    int M, H, W;
};
constexpr int HC_ROWS = 256;
constexpr size_t hc_lds_bytes() { return (size_t)(HC_ROWS + 2 * CONV_HALO_MAX + 1) * 256; }

__global__ __launch_bounds__(RB3_THREADS, 2) void k_conv_heads(HeadsConvArgs a) {
    constexpr int SLOTS = 16, TM = 2, KS = 8, BN = 32, BSL = BN * SLOTS;
    constexpr int AROWS = HC_ROWS + 2 * CONV_HALO_MAX + 1, ZROW = AROWS - 1;
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int h = a.W + 1, HW = a.H * a.W;
    const long m0 = (long)blockIdx.x * HC_ROWS;
    const uint4* in4 = reinterpret_cast<const uint4*>(a.in);

    // image row q <-> global row m0 - h + q
    const int n_aslots = (HC_ROWS + 2 * h) * SLOTS;
    for (int base = wave * 64; base < n_aslots; base += RB3_THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 - h + lr;
        gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
    }
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);

    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, 9 * BSL * 16, 0x00020000);
    const int bvo = (lhi * BN + l31) * 16;
    auto ldb = [&](int tap, int ks) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo, (tap * BSL + ks * 2 * BN) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    // Only two MFMAs separate consecutive k-steps here, so the ring is three taps (24 k-steps, ~1500 MFMA cycles) deep.
    constexpr int RING = 24;
    uint4 bfr[RING];
#pragma unroll
    for (int g = 0; g < RING; ++g) bfr[g] = ldb(g / KS, g % KS);

    int lrow[TM]; unsigned vmask[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wave * TM + tm) * 32 + l31;
        const long gr = m0 + lrow[tm];
        unsigned mm = 0;
        if (gr < a.M) {
            const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << t;
            }
        }
        vmask[tm] = mm;
    }
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.0f;
    __syncthreads();                                // image landed (the only barrier)

#pragma unroll 1
    for (int t3 = 0; t3 < 3; ++t3) {                // three taps per iteration: ring slot = k-step index within the triple
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int tap = t3 * 3 + tt;
            const int off = (t3 - 1) * a.W + (tt - 1);
            int abase[TM], axor[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bool ok = (vmask[tm] >> tap) & 1u;
                const int ar = ok ? lrow[tm] + h + off : ZROW;
                abase[tm] = ar * SLOTS; axor[tm] = ar & 15;
            }
            uint4 afr[2][TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + (lhi ^ axor[tm])];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < KS) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks + 1) * 2 + lhi) ^ axor[tm])];
                }
                const bf16x8 bf = *reinterpret_cast<bf16x8*>(&bfr[tt * KS + ks]);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&afr[cur][tm]), bf, acc[tm], 0, 0, 0);
                const int ntap = tap + 3 < 9 ? tap + 3 : tap;           // the last three taps re-read themselves
                bfr[tt * KS + ks] = ldb(ntap, ks);
            }
        }
    }
    // columns 0-7 policy conv, 8-15 value conv, the rest padding; C/D layout: col = l31, row = (r & 3) + 8 (r >> 2) + 4 lhi
    if (l31 >= 16) return;
    const float tA = a.bias[l31];
    const float* __restrict__ fs = l31 < 8 ? a.p_fs : a.v_fs; const float* __restrict__ ft = l31 < 8 ? a.p_ft : a.v_ft;
    float* __restrict__ feat = l31 < 8 ? a.p_feat : a.v_feat;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // all 32 BN parameters first, then the stores: a load behind a store through an unrelated pointer is not hoisted, and
        // the loop would pay one L2 round trip per output
        float sc[16], sh[16]; size_t o[16]; bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wave * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const long gr = m0 + row;
            ok[r] = gr < a.M;
            const unsigned g32 = ok[r] ? (unsigned)gr : 0u;
            const unsigned b = g32 / (unsigned)HW; const int cell = (int)(g32 - b * (unsigned)HW), f = cell * 8 + (l31 & 7);
            sc[r] = fs[f]; sh[r] = ft[f]; o[r] = (size_t)b * (HW * 8) + f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) feat[o[r]] = fmaxf((acc[tm][r] + tA) * sc[r] + sh[r], 0.0f);
    }
}

// ---- k_block0: the FIRST residual block of the Gomoku network (Gomoku/Build_Model.py: 256 stem channels -> num_filters, so the
// block has a 1x1 projection on its skip path, Net/ResNet/ResNet_Block.py:21-33) in the k_resblock3 structure:
//     x_out = Conv3x3_2( relu(bn2( Conv3x3_1(a0) )) ) + Conv1x1_proj(x0)      a0 = relu(bn1(x0)) comes from the stem kernel
// The 256-channel operands pass through the 128-channel LDS image in two halves: 29 weight slices of K = 128 in one array —
// conv1 low half (9 taps), conv1 high half (9), conv2 (9), projection low / high (centre tap only) — walked by the same
// register ring; the image is reloaded by LDS-DMA between the phases (5 loads instead of 1, the price of not writing
// h = conv1(...) and proj(x0) to HBM and back: 873 us for three kernels before).
struct Block0Args {
    const bf16_t* a0; const bf16_t* x0;           // [M][256] pre-activated / raw stem output.  a0 == null (round 2): the stem writes only x0 and
    const float* s1; const float* t1;             // conv1's operand relu(x0 * s1 + t1) is made IN LDS after the image landed ([256] each)
    bf16_t* xout;                                 // [M][128]
    const bf16_t* w;                              // 29 slices [8 k-steps][2][128 cout][8] (fragment order)
    const float* s2; const float* t2;             // bn2 folded with conv1's bias (ReLU)
    const float* b2; const float* bp;             // conv2 bias, projection bias
    int M, H, W, halo, tile_rows;
};

template <int TM, int RING>
__global__ __launch_bounds__(RB3_THREADS, 2) void k_block0(Block0Args a) {
    constexpr int BN = 128, SLOTS = 16, TN = 2, KS = 8, ROWS = rb3_rows<TM>(), NSL = 29;
    constexpr int AROWS = ROWS + 2 * CONV_HALO_MAX + 1, ZROW = AROWS - 1, BSL = BN * SLOTS;
    static_assert(KS % RING == 0, "ring slot must not depend on the slice");
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhi = lane >> 5;
    const int h = a.halo, HW = a.H * a.W, bmo = a.tile_rows;
    const long m0 = (long)blockIdx.x * bmo;
    const int col0 = wn * 64 + l31;

    // image row q <-> global row row0 + q of channels [128 half, 128 half + 128) of a 256-channel tensor
    auto load_image = [&](const bf16_t* src, int half, long row0, int nrows) {
        const uint4* in4 = reinterpret_cast<const uint4*>(src);
        for (int base = wave * 64; base < nrows * SLOTS; base += RB3_THREADS) {
            const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
            long gr = row0 + lr;
            gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);
            __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * (2 * SLOTS) + half * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
        }
    };
    // in-place pre-activation of the image just loaded from the RAW stem output: slot i of image row lr holds channels
    // 128 half + 8 (sp ^ (lr & 15)) ..+7 (the swizzle went through the DMA's source address)
    auto preactivate = [&](int half, int nrows) {
        for (int i = tid; i < nrows * SLOTS; i += RB3_THREADS) {
            const int lr = i / SLOTS, sp = i % SLOTS, c0 = half * 128 + ((sp ^ (lr & 15)) << 3);
            const float4 sa = *reinterpret_cast<const float4*>(a.s1 + c0), sb = *reinterpret_cast<const float4*>(a.s1 + c0 + 4);
            const float4 ta = *reinterpret_cast<const float4*>(a.t1 + c0), tb = *reinterpret_cast<const float4*>(a.t1 + c0 + 4);
            const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w}, sh[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
            const uint4 v = As[i];
            unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = fmaxf(__uint_as_float(w[j] << 16) * sc[2 * j] + sh[2 * j], 0.0f);
                const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * sc[2 * j + 1] + sh[2 * j + 1], 0.0f);
                w[j] = pack_bf16(lo, hi);
            }
            As[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    const bool inplace = a.a0 == nullptr;
    load_image(inplace ? a.x0 : a.a0, 0, m0 - 2 * h, ROWS + 2 * h);
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);

    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, NSL * BSL * 16, 0x00020000);
    const int bvo = (lhi * BN + col0) * 16;
    auto ldb = [&](int slice, int ks, int tn) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo, ((slice * BSL) + ks * 2 * BN + tn * 32) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 bfr[RING][TN];
#pragma unroll
    for (int g = 0; g < RING; ++g)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[g][tn] = ldb(0, g, tn);
    float ps2[TN], pt2[TN], pb[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) { ps2[tn] = a.s2[col0 + tn * 32]; pt2[tn] = a.t2[col0 + tn * 32]; pb[tn] = a.b2[col0 + tn * 32] + a.bp[col0 + tn * 32]; }

    int lrow[TM]; unsigned vmask[TM];               // bits 0-8: conv1 taps, 9-17: conv2 taps, 18: the row itself (projection)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wm * TM + tm) * 32 + l31;
        unsigned mm = 0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const long gr = m0 + lrow[tm] - (which == 0 ? h : 0);
            if (gr >= 0 && gr < a.M) {
                const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int dy = t / 3 - 1, dx = t % 3 - 1;
                    mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << (which * 9 + t);
                }
                if (which == 1) mm |= 1u << 18;
            }
        }
        vmask[tm] = mm;
    }

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
    };
    // slices [s0, s0 + n): slice s uses tap offset taps9 ? (s - s0) : centre, validity bit mbit0 (+ tap index when taps9)
    auto run_slices = [&](int s0, int n, bool taps9, int mbit0) {
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            const int sl = s0 + i, nsl = sl + 1 < NSL ? sl + 1 : sl;
            const int tap = taps9 ? i : 4, ty = tap / 3, off = (ty - 1) * a.W + (tap - ty * 3 - 1);
            const int mbit = taps9 ? mbit0 + i : mbit0;
            int abase[TM], axor[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bool ok = (vmask[tm] >> mbit) & 1u;
                const int ar = ok ? lrow[tm] + h + off : ZROW;
                abase[tm] = ar * SLOTS; axor[tm] = ar & 15;
            }
            uint4 afr[2][TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + (lhi ^ axor[tm])];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < KS) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks + 1) * 2 + lhi) ^ axor[tm])];
                }
                bf16x8 bf[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<bf16x8*>(&bfr[ks % RING][tn]);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&afr[cur][tm]), bf[tn], acc[tm][tn], 0, 0, 0);
                if (ks + RING < KS) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(sl, ks + RING, tn);
                } else {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(nsl, ks + RING - KS, tn);
                }
            }
        }
    };

    zero_acc();
    __syncthreads();                                // a0 / x0 low half landed
    if (inplace) { preactivate(0, ROWS + 2 * h); __syncthreads(); }
    run_slices(0, 9, true, 0);                      // conv1, input channels 0-127
    __syncthreads();                                // every wave is done with this image
    load_image(inplace ? a.x0 : a.a0, 1, m0 - 2 * h, ROWS + 2 * h);
    __syncthreads();
    if (inplace) { preactivate(1, ROWS + 2 * h); __syncthreads(); }
    run_slices(9, 9, true, 0);                      // conv1, input channels 128-255
    __syncthreads();
    {   // h = relu(acc * s2 + t2) as bf16 into the image region (row j of h at image row j, same swizzle)
        bf16_t* Hs = reinterpret_cast<bf16_t*>(As);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = col0 + tn * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float v = fmaxf(acc[tm][tn][r] * ps2[tn] + pt2[tn], 0.0f);
                    Hs[row * 128 + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7))] = (bf16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
                }
            }
    }
    __syncthreads();
    zero_acc();
    run_slices(18, 9, true, 9);                     // conv2 over h
    __syncthreads();
    load_image(a.x0, 0, m0 - h, ROWS + h);          // image row q <-> global m0 - h + q, like the h image
    __syncthreads();
    run_slices(27, 1, false, 18);                   // + projection of x0, channels 0-127
    __syncthreads();
    load_image(a.x0, 1, m0 - h, ROWS + h);
    __syncthreads();
    run_slices(28, 1, false, 18);                   // + projection of x0, channels 128-255
    __syncthreads();

    // epilogue, half the rows at a time through an fp32 tile over the image region: + conv2 bias + projection bias
    constexpr int CT = BN + 4, HR = ROWS / 2;
    float* Ct = reinterpret_cast<float*>(lds);
    static_assert((size_t)HR * CT * 4 <= (size_t)AROWS * SLOTS * 16, "epilogue tile must fit in the image region");
    const int chunk = tid % 16, r0 = tid / 16;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        Ct[row * CT + col0 + tn * 32] = acc[tm][tn][r] + pb[tn];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2 * TM; ++q) {
            const int rl = r0 + q * (RB3_THREADS / 16), row = pass * HR + rl;
            const long gr = m0 + row;
            if (row < bmo && gr < a.M) {
                const float4 c0 = *reinterpret_cast<const float4*>(&Ct[rl * CT + chunk * 8]);
                const float4 c1 = *reinterpret_cast<const float4*>(&Ct[rl * CT + chunk * 8 + 4]);
                *reinterpret_cast<uint4*>(a.xout + (size_t)gr * BN + chunk * 8) =
                    make_uint4(pack_bf16(c0.x, c0.y), pack_bf16(c0.z, c0.w), pack_bf16(c1.x, c1.y), pack_bf16(c1.z, c1.w));
            }
        }
        if (pass == 0) __syncthreads();
    }
}

// ---- k_conv_head32: first convolution of a Gomoku head (Gomoku/Build_Model.py: BN, ReLU, Conv3x3 128 -> 32, BN, ReLU) for both
// heads in one launch (blockIdx.y = head).  Structure of k_conv_heads plus the in-place pre-activation of k_resblock3: the raw
// trunk output is loaded once per head and transformed in LDS with that head's BN, which removes the two k_affine_relu passes
// (a 118-MB read and write each) the separate kernels needed.  Output bf16 [M][32] = relu(acc * scale + shift).
struct Head32Args {
    const bf16_t* in;                             // [M][128] raw trunk output
    const bf16_t* wgt[2];                         // fragment order [9][8][2][32][8]
    const float* s0[2]; const float* t0[2];       // head BN on the input (with ReLU)
    const float* s1[2]; const float* t1[2];       // BN after the conv folded with its bias (with ReLU)
    bf16_t* out[2];                               // [M][32]
    int M, H, W;
};

__global__ __launch_bounds__(RB3_THREADS, 2) void k_conv_head32(Head32Args a) {
    constexpr int SLOTS = 16, TM = 2, KS = 8, BN = 32, BSL = BN * SLOTS, RING = 24;
    constexpr int AROWS = HC_ROWS + 2 * CONV_HALO_MAX + 1, ZROW = AROWS - 1;
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    const int head = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int h = a.W + 1, HW = a.H * a.W;
    const long m0 = (long)blockIdx.x * HC_ROWS;
    const uint4* in4 = reinterpret_cast<const uint4*>(a.in);

    const int n_aslots = (HC_ROWS + 2 * h) * SLOTS;
    for (int base = wave * 64; base < n_aslots; base += RB3_THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 - h + lr;
        gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
    }
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);

    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt[head], 0, 9 * BSL * 16, 0x00020000);
    const int bvo = (lhi * BN + l31) * 16;
    auto ldb = [&](int tap, int ks) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo, (tap * BSL + ks * 2 * BN) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 bfr[RING];
#pragma unroll
    for (int g = 0; g < RING; ++g) bfr[g] = ldb(g / KS, g % KS);
    const int tch0 = ((tid % SLOTS) ^ ((tid / SLOTS) & 15)) * 8;
    float ps0[8], pt0[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ps0[j] = a.s0[head][tch0 + j]; pt0[j] = a.t0[head][tch0 + j]; }
    const float ps1 = a.s1[head][l31], pt1 = a.t1[head][l31];

    int lrow[TM]; unsigned vmask[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wave * TM + tm) * 32 + l31;
        const long gr = m0 + lrow[tm];
        unsigned mm = 0;
        if (gr < a.M) {
            const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << t;
            }
        }
        vmask[tm] = mm;
    }
    f32x16 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][r] = 0.0f;
    __syncthreads();                                // image landed
    for (int i = tid; i < n_aslots; i += RB3_THREADS) {        // in-place pre-activation with this head's BN
        uint4 v = As[i];
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = fmaxf(__uint_as_float(w[j] << 16) * ps0[2 * j] + pt0[2 * j], 0.0f);
            const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * ps0[2 * j + 1] + pt0[2 * j + 1], 0.0f);
            w[j] = pack_bf16(lo, hi);
        }
        As[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();

#pragma unroll 1
    for (int t3 = 0; t3 < 3; ++t3) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            const int tap = t3 * 3 + tt;
            const int off = (t3 - 1) * a.W + (tt - 1);
            int abase[TM], axor[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const bool ok = (vmask[tm] >> tap) & 1u;
                const int ar = ok ? lrow[tm] + h + off : ZROW;
                abase[tm] = ar * SLOTS; axor[tm] = ar & 15;
            }
            uint4 afr[2][TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + (lhi ^ axor[tm])];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < KS) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks + 1) * 2 + lhi) ^ axor[tm])];
                }
                const bf16x8 bf = *reinterpret_cast<bf16x8*>(&bfr[tt * KS + ks]);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&afr[cur][tm]), bf, acc[tm], 0, 0, 0);
                const int ntap = tap + 3 < 9 ? tap + 3 : tap;
                bfr[tt * KS + ks] = ldb(ntap, ks);
            }
        }
    }
    // C/D layout: col = l31 (channel), row = (r & 3) + 8 (r >> 2) + 4 lhi: 64 contiguous bytes per row and half-wave
    bf16_t* out = a.out[head];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wave * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const long gr = m0 + row;
            if (gr < a.M) out[(size_t)gr * 32 + l31] = (bf16_t)(pack_bf16(fmaxf(acc[tm][r] * ps1 + pt1, 0.0f), 0.0f) & 0xFFFFu);
        }
}

}  // namespace gaz
