// conv3x3.hpp — 3x3 "same" convolution on NHWC bf16 board tensors as an implicit GEMM on the gfx950 matrix
// cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Replaces the Conv2D layers of the reference network
// (Net/ResNet/ResNet_Block.py:12-20, Connect4/Build_Model.py:41,62) as executed by ONNX Runtime.
//
// GEMM view: M = B*H*W flattened board cells, N = Cout, K = 9*Cin.  One 512-thread workgroup (8 waves, two
// per SIMD) owns 256 consecutive cells x all Cout:
//   * the 256 rows plus a (W+1)-row halo of the activation are brought into LDS ONCE by LDS-DMA
//     (global_load_lds_dwordx4) and serve all nine taps: the tap (dy,dx) operand of cell r is image row
//     r + dy*W + dx, zeroed by a per-lane validity bit at board edges (rows of neighbouring boards that the
//     halo drags in are never selected, so nothing needs to be zero-filled);
//   * the per-tap weight slice [Cout][Cin] (32 KB) is double-buffered in LDS: the DMA of tap t+1 is in flight
//     while tap t multiplies, one barrier per tap;
//   * the activation image is XOR-swizzled in 16-byte slots (slot ^= row & 15) by permuting the per-lane SOURCE
//     address of the DMA (the LDS side of a DMA is lane-linear); the weights are pre-arranged on the host in
//     fragment order [tap][k-step][k-half][cout][8] so their DMA is a straight copy and every fragment read is
//     base + compile-time offset; both give conflict-free ds_read_b128; masked rows read a zero row instead of
//     being zeroed in registers; fragments are register double-buffered across k-steps;
//   * epilogue: accumulators -> fp32 LDS tile -> per-thread 16-byte channel groups: folded BN scale/shift,
//     residual add, activation, and (for the pre-activation blocks) a second output relu(bn_next(x)) — whole
//     256-byte rows per 16 lanes, dwordx4 stores.
// Algorithmic cost of one launch (Connect4 trunk, B = 4096): 2*M*N*K = 2*172032*128*1152 = 50.7 GFLOP.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gaz {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

enum { ACT_NONE = 0, ACT_RELU = 1 };

struct ConvArgs {
    const bf16_t* in;        // [M][CIN] NHWC rows
    const bf16_t* wgt;       // [9][CIN/16][2][BN][8]: fragment order (k = ks*16 + half*8 + j), see arrange_conv_weights()
    const float* scaleA;     // [BN] or null (=1)
    const float* shiftA;     // [BN] or null (=0)
    const bf16_t* res;       // [M][BN] residual or null (may alias out1)
    bf16_t* out1;            // [M][BN]: act1(acc*scaleA + shiftA + res)
    int act1;
    const float* scaleB;     // second output: relu(out1*scaleB + shiftB); out2 null = none
    const float* shiftB;
    bf16_t* out2;
    int M, H, W;
};

constexpr int CONV_HALO_MAX = 16;           // W + 1 <= 16
constexpr int CONV_AROWS_256 = 256 + 2 * CONV_HALO_MAX + 1;
// LDS bytes of one workgroup: activation image (BM + halo rows + one zero row) + two weight slices of CIN/KSPLIT channels
template <int CIN, int BN, int BM, int KSPLIT> constexpr size_t conv_lds_bytes() {
    return (size_t)((BM + 2 * CONV_HALO_MAX + 1) * (CIN / 8) + 2 * BN * (CIN / 8) / KSPLIT) * 16;
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {     // v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return *reinterpret_cast<unsigned*>(&v);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// BM rows per workgroup, WM x WN waves each owning TM x TN MFMA tiles of 32x32; the weight stream is cut into
// 9 * KSPLIT slices of CIN / KSPLIT input channels.  OCC = workgroups per CU the LDS budget is sized for.
template <int CIN, int BN, int BM, int WM, int WN, int TM, int TN, int KSPLIT, int OCC, int EPI, int NTAPS = 9>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * OCC + 3) / 4) void k_conv3x3(ConvArgs a) {
    constexpr int CONV_BM = BM, CONV_THREADS = WM * WN * 64;
    static_assert(WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    constexpr int SLOTS = CIN / 8;                 // 16-byte slots per row (16 for CIN = 128)
    static_assert(SLOTS == 16 || SLOTS == 32, "swizzle: XOR of the low four slot bits with row & 15");
    static_assert(NTAPS == 9 || NTAPS == 1, "3x3 or 1x1 (centre tap only)");
    constexpr int AROWS = BM + 2 * CONV_HALO_MAX + 1, ZROW = AROWS - 1;
    constexpr int KSS = CIN / 16 / KSPLIT;         // k-steps per weight slice
    constexpr int BSL = BN * SLOTS / KSPLIT;       // 16-byte units per weight slice
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    uint4* Bs = lds + AROWS * SLOTS;               // two slices of BSL slots

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const long m0 = (long)blockIdx.x * CONV_BM;
    const int halo = a.W + 1, HW = a.H * a.W;
    const uint4* in4 = reinterpret_cast<const uint4*>(a.in);
    const uint4* w4 = reinterpret_cast<const uint4*>(a.wgt);

    // ---- LDS-DMA of the activation image.  Slot index i = row * 16 + s' is lane-linear; it receives source
    // slot s = s' ^ (row & 15) of that row, so a later read of logical slot s at s ^ (row & 15) finds it.
    const int n_aslots = (CONV_BM + 2 * halo) * SLOTS;
    for (int base = wave * 64; base < n_aslots; base += CONV_THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 - halo + lr;
        gr = gr < 0 ? 0 : (gr >= a.M ? (long)a.M - 1 : gr);     // rows outside the tensor are never selected
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + (sp ^ (lr & 15))), (lds_ptr_t)(As + base), 16, 0, 0);
    }
    if (tid < SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);
    // weight slice of tap 0 -> Bs[0] (already in fragment order: straight copy)
    for (int base = wave * 64; base < BSL; base += CONV_THREADS)
        __builtin_amdgcn_global_load_lds((const void*)(w4 + base + lane), (lds_ptr_t)(Bs + base), 16, 0, 0);

    // ---- per-lane geometry of the TM row tiles this wave owns
    int lrow[TM]; unsigned vmask[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wm * TM + tm) * 32 + l31;
        const long gr = m0 + lrow[tm];
        const int cell = (int)((unsigned)gr % (unsigned)HW), y = cell / a.W, x = cell % a.W;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = gr < a.M && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
            m |= (ok ? 1u : 0u) << t;
        }
        vmask[tm] = m;
    }
    int bbase[TN];                                  // unit index of (ks = 0, half = lhi, n) in a slice
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bbase[tn] = lhi * BN + (wn * TN + tn) * 32 + l31;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    __syncthreads();                                // image + slice 0 have landed (the barrier drains the DMA queue)

    for (int sl = 0; sl < NTAPS * KSPLIT; ++sl) {
        const int tap = NTAPS == 9 ? sl / KSPLIT : 4, ks0 = (sl % KSPLIT) * KSS;
        const uint4* Bc = Bs + (sl & 1) * BSL;
        if (sl + 1 < NTAPS * KSPLIT ) {  // DMA of the next slice into the other buffer, in flight during the MFMAs
            uint4* Bn = Bs + ((sl + 1) & 1) * BSL;
            const uint4* wsrc = w4 + (size_t)(sl + 1) * BSL;
            for (int base = wave * 64; base < BSL; base += CONV_THREADS)
                __builtin_amdgcn_global_load_lds((const void*)(wsrc + base + lane), (lds_ptr_t)(Bn + base), 16, 0, 0);
        }
        const int off = (tap / 3 - 1) * a.W + (tap % 3 - 1);
        int abase[TM], axor[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const bool ok = (vmask[tm] >> tap) & 1u;
            const int ar = ok ? lrow[tm] + halo + off : ZROW;       // masked taps read the zero row
            abase[tm] = ar * SLOTS; axor[tm] = ar & 15;
        }
        // software pipeline over the k-steps: fragments of step ks+1 are in flight while step ks multiplies
        uint4 afr[2][TM], bfr[2][TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + ((ks0 * 2 + lhi) ^ axor[tm])];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[0][tn] = Bc[bbase[tn]];
#pragma unroll
        for (int ks = 0; ks < KSS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < KSS) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks0 + ks + 1) * 2 + lhi) ^ axor[tm])];
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
        __syncthreads();                            // slice tap+1 landed; everyone is done reading slice tap
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    static_assert(EPI == 0, "one epilogue");
    {
        constexpr int CT = BN + 4;
        float* Ct = reinterpret_cast<float*>(lds);  // [BM][BN + 4] fp32 over the (now idle) image + slices
        static_assert((size_t)BM * CT * 4 <= conv_lds_bytes<CIN, BN, BM, KSPLIT>(), "epilogue tile must fit");
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = (wn * TN + tn) * 32 + l31;
                const float sA = a.scaleA ? a.scaleA[col] : 1.0f, tA = a.shiftA ? a.shiftA[col] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    Ct[row * CT + col] = acc[tm][tn][r] * sA + tA;
                }
            }
        __syncthreads();
        constexpr int CHUNKS = BN / 8;              // 16-byte (8 x bf16) groups per row
        const int chunk = tid % CHUNKS, r0 = tid / CHUNKS;
        float sB[8], tB[8];
        if (a.out2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sB[j] = a.scaleB[chunk * 8 + j]; tB[j] = a.shiftB[chunk * 8 + j]; }
        }
        for (int row = r0; row < CONV_BM; row += CONV_THREADS / CHUNKS) {
            const long gr = m0 + row;
            if (gr >= a.M) break;
            const float4 c0 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8]);
            const float4 c1 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8 + 4]);
            float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const size_t o = (size_t)gr * BN + chunk * 8;
            if (a.res) {
                const uint4 rv = *reinterpret_cast<const uint4*>(a.res + o);
                const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] += __uint_as_float(rw[j] << 16);
                    v[2 * j + 1] += __uint_as_float(rw[j] & 0xFFFF0000u);
                }
            }
            if (a.act1 == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
            }
            *reinterpret_cast<uint4*>(a.out1 + o) =
                make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
            if (a.out2) {
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = fmaxf(v[j] * sB[j] + tB[j], 0.0f);
                *reinterpret_cast<uint4*>(a.out2 + o) =
                    make_uint4(pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]), pack_bf16(w[4], w[5]), pack_bf16(w[6], w[7]));
            }
        }
    }
}

// Host: [9][cout][cin] (export order) -> [9][cin/16][2][cout][8] (MFMA B-fragment order).
inline void arrange_conv_weights(const float* src, int cout, int cin, bf16_t* dst, bf16_t (*cvt)(float), int ntaps = 9) {
    for (int tap = 0; tap < ntaps; ++tap)
        for (int n = 0; n < cout; ++n)
            for (int k = 0; k < cin; ++k) {
                const int ks = k / 16, half = (k % 16) / 8, j = k % 8;
                dst[((((size_t)tap * (cin / 16) + ks) * 2 + half) * cout + n) * 8 + j] = cvt(src[((size_t)tap * cout + n) * cin + k]);
            }
}

}  // namespace gaz
