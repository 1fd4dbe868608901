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
#include "evaluator.hpp"

namespace gaz {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static bf16_t f2bf_host(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct ConvArgs {
    const bf16_t* in;        // [M][CIN] NHWC rows
    const bf16_t* wgt;       // [9][BN][CIN]
    const float* scaleA;     // [BN] or null (=1)
    const float* shiftA;     // [BN]
    const bf16_t* res;       // [M][BN] residual or null
    bf16_t* out1;            // [M][BN]: act1(acc*scaleA + shiftA + res)
    int act1;
    const float* scaleB;     // second output: relu(out1*scaleB + shiftB), null = none
    const float* shiftB;
    bf16_t* out2;
    // heads epilogue (EPI == 1): features relu((acc + shiftA[c]) * fs[cell*8+c] + ft[cell*8+c]) -> f32 [B][HW*8]
    const float* p_fs; const float* p_ft; const float* v_fs; const float* v_ft;
    float* p_feat; float* v_feat;
    int M, H, W;
};

constexpr int CONV_BM = 128;
constexpr int CONV_HALO_MAX = 16;

template <int CIN, int BN, int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__(256, 2) void k_conv3x3(ConvArgs a) {
    static_assert(WM * WN == 4 && WM * TM * 32 == CONV_BM && WN * TN * 32 == BN, "tile shape");
    constexpr int SLOTS = CIN / 8;                 // 16-byte slots per row
    constexpr int AROWS = CONV_BM + 2 * CONV_HALO_MAX;
    constexpr int NB = BN * SLOTS / 256;           // weight uint4 per thread per tap
    extern __shared__ uint4 lds[];
    uint4* As = lds;
    uint4* Bs = lds + AROWS * SLOTS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const long m0 = (long)blockIdx.x * CONV_BM;
    const int halo = a.W + 1, HW = a.H * a.W;

    // ---- stage the activation image: rows [m0 - halo, m0 + BM + halo)
    const uint4* in4 = reinterpret_cast<const uint4*>(a.in);
    for (int idx = tid; idx < (CONV_BM + 2 * halo) * SLOTS; idx += 256) {
        const int lr = idx / SLOTS, s = idx % SLOTS;
        const long gr = m0 - halo + lr;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gr >= 0 && gr < a.M) v = in4[gr * SLOTS + s];
        As[lr * SLOTS + (s ^ (lr & 15))] = v;
    }
    // ---- per-lane geometry of the TM row tiles this wave owns
    int lrow[TM]; unsigned vmask[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        lrow[tm] = (wm * TM + tm) * 32 + l31;
        const long gr = m0 + lrow[tm];
        const int cell = (int)(gr % HW), y = cell / a.W, x = cell % a.W;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const bool ok = gr < a.M && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
            m |= (ok ? 1u : 0u) << t;
        }
        vmask[tm] = m;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;

    const uint4* w4 = reinterpret_cast<const uint4*>(a.wgt);
    uint4 breg[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) breg[j] = w4[(size_t)(tid + j * 256)];

    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();                            // everyone is done with the previous weight slice (and As is staged)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + j * 256, n = idx / SLOTS, s = idx % SLOTS;
            Bs[n * SLOTS + (s ^ (n & 15))] = breg[j];
        }
        __syncthreads();
        if (tap + 1 < 9) {
#pragma unroll
            for (int j = 0; j < NB; ++j) breg[j] = w4[(size_t)(tap + 1) * (BN * SLOTS) + tid + j * 256];
        }
        const int off = (tap / 3 - 1) * a.W + (tap % 3 - 1);
        // fragment addresses of this tap (the XOR swizzle only touches the slot index, so ks adds in XOR space)
        int abase[TM], axor[TM], bbase[TN], bxor[TN]; bool aval[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int ar = lrow[tm] + halo + off;
            abase[tm] = ar * SLOTS; axor[tm] = ar & 15; aval[tm] = (vmask[tm] >> tap) & 1u;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = (wn * TN + tn) * 32 + l31;
            bbase[tn] = n * SLOTS; bxor[tn] = n & 15;
        }
        // software pipeline over the k-steps: fragments of step ks+1 are in flight while step ks multiplies
        uint4 afr[2][TM], bfr[2][TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) afr[0][tm] = As[abase[tm] + (lhi ^ axor[tm])];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[0][tn] = Bs[bbase[tn] + (lhi ^ bxor[tn])];
#pragma unroll
        for (int ks = 0; ks < CIN / 16; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < CIN / 16) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) afr[nxt][tm] = As[abase[tm] + (((ks + 1) * 2 + lhi) ^ axor[tm])];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bfr[nxt][tn] = Bs[bbase[tn] + (((ks + 1) * 2 + lhi) ^ bxor[tn])];
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                uint4 v = afr[cur][tm];
                if (!aval[tm]) v = make_uint4(0, 0, 0, 0);
                const bf16x8 af = *reinterpret_cast<bf16x8*>(&v);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<bf16x8*>(&bfr[cur][tn]), acc[tm][tn], 0, 0, 0);
            }
        }
    }

    if (EPI == 0) {
        // ---- epilogue through LDS: accumulators -> fp32 tile [128][BN + 4], then whole 16-byte channel groups per
        // thread: residual load, activation, both outputs as dwordx4 stores (full 256-B rows per 16 lanes).
        constexpr int CT = BN + 4;
        float* Ct = reinterpret_cast<float*>(lds);
        __syncthreads();                            // all waves are done reading As / Bs
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
        for (int row = r0; row < CONV_BM; row += 256 / CHUNKS) {
            const long gr = m0 + row;
            if (gr >= a.M) break;
            const float4 c0 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8]);
            const float4 c1 = *reinterpret_cast<const float4*>(&Ct[row * CT + chunk * 8 + 4]);
            float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const size_t o = (size_t)gr * BN + chunk * 8;
            if (a.res) {
                const uint4 rv = *reinterpret_cast<const uint4*>(a.res + o);
                const bf16_t* rb = reinterpret_cast<const bf16_t*>(&rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += bf2f(rb[j]);
            }
            if (a.act1 == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
            }
            bf16_t o1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o1[j] = f2bf(v[j]);
            *reinterpret_cast<uint4*>(a.out1 + o) = *reinterpret_cast<const uint4*>(o1);
            if (a.out2) {
                bf16_t o2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o2[j] = f2bf(fmaxf(v[j] * sB[j] + tB[j], 0.0f));
                *reinterpret_cast<uint4*>(a.out2 + o) = *reinterpret_cast<const uint4*>(o2);
            }
        }
        return;
    }

    // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = (wn * TN + tn) * 32 + l31;
            const float sA = a.scaleA ? a.scaleA[col] : 1.0f, tA = a.shiftA ? a.shiftA[col] : 0.0f;
            float sB = 0.f, tB = 0.f;
            if (EPI == 0 && a.out2) { sB = a.scaleB[col]; tB = a.shiftB[col]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const long gr = m0 + row;
                if (gr >= a.M) continue;
                float v = acc[tm][tn][r] * sA + tA;
                if (EPI == 0) {
                    const size_t o = (size_t)gr * BN + col;
                    if (a.res) v += bf2f(a.res[o]);
                    if (a.act1 == ACT_RELU) v = fmaxf(v, 0.0f);
                    a.out1[o] = f2bf(v);
                    if (a.out2) a.out2[o] = f2bf(fmaxf(v * sB + tB, 0.0f));
                } else {
                    if (col < 16) {     // 0-7 policy conv channels, 8-15 value conv channels
                        const long b = gr / HW; const int cell = (int)(gr % HW), c = col & 7;
                        const int f = cell * 8 + c;
                        if (col < 8) a.p_feat[(size_t)b * (HW * 8) + f] = fmaxf(v * a.p_fs[f] + a.p_ft[f], 0.0f);
                        else a.v_feat[(size_t)b * (HW * 8) + f] = fmaxf(v * a.v_fs[f] + a.v_ft[f], 0.0f);
                    }
                }
            }
        }
    }
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
        *reinterpret_cast<uint4*>(a.out2 + (size_t)gr * 128 + ch0) = *reinterpret_cast<uint4*>(o2);
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

// ---- heads tail: Dense(128 -> 64) -> Dense(64 -> A | 1) -> softmax | tanh.  One wave per position.
struct TailArgs {
    const float* p_d1; const float* v_d1;          // [B][128]
    const float* p_w2; const float* p_b2; const float* p_w3; const float* p_b3;
    const float* v_w2; const float* v_b2; const float* v_w3; const float* v_b3;
    float* policy; float* value; int B, A, logits;
};
__global__ __launch_bounds__(64) void k_tail(TailArgs a) {
    __shared__ float x[128]; __shared__ float h[64]; __shared__ float lg[64];
    const int b = blockIdx.x, l = threadIdx.x;
    for (int head = 0; head < 2; ++head) {
        const float* d1 = head == 0 ? a.p_d1 : a.v_d1;
        const float* w2 = head == 0 ? a.p_w2 : a.v_w2; const float* b2 = head == 0 ? a.p_b2 : a.v_b2;
        const float* w3 = head == 0 ? a.p_w3 : a.v_w3; const float* b3 = head == 0 ? a.p_b3 : a.v_b3;
        const int nout = head == 0 ? a.A : 1;
        x[l] = d1[(size_t)b * 128 + l]; x[l + 64] = d1[(size_t)b * 128 + 64 + l];
        __syncthreads();
        float acc = b2[l];
        for (int k = 0; k < 128; ++k) acc += x[k] * w2[k * 64 + l];
        h[l] = acc;
        __syncthreads();
        if (l < nout) {
            float z = b3[l];
            for (int k = 0; k < 64; ++k) z += h[k] * w3[k * nout + l];
            lg[l] = z;
        }
        __syncthreads();
        if (head == 0) {
            if (l < nout) {
                if (a.logits) a.policy[(size_t)b * a.A + l] = lg[l];
                else {
                    float mx = lg[0];
                    for (int k = 1; k < nout; ++k) mx = fmaxf(mx, lg[k]);
                    float sum = 0.f;
                    for (int k = 0; k < nout; ++k) sum += expf(lg[k] - mx);
                    a.policy[(size_t)b * a.A + l] = expf(lg[l] - mx) / sum;
                }
            }
        } else if (l == 0) {
            a.value[b] = tanhf(lg[0]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ host
struct ResNetEvaluator : Evaluator {
    int H, W, C, A, HW, blocks, filters, nmax, logits;
    std::map<std::string, float*> f32;              // device fp32 tensors by name
    std::map<std::string, bf16_t*> b16;             // device bf16 conv weights by name
    bf16_t *X = nullptr, *Aa = nullptr, *Hh = nullptr;
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
        auto up_b16 = [&](const std::string& name, int64_t numel) -> bool {
            const gaz_tensor* g = need(name, numel); if (!g) return false;
            std::vector<bf16_t> h(numel);
            for (int64_t i = 0; i < numel; ++i) h[i] = f2bf_host(g->data[i]);
            bf16_t* d = dalloc<bf16_t>(numel); if (!d) { *err = "hipMalloc"; return false; }
            hipMemcpy(d, h.data(), numel * 2, hipMemcpyHostToDevice); b16[name] = d; return true;
        };
        const int Fc = filters, F = HW * 8;
        if (!up_f32("stem.w", 9 * 128 * C) || !up_f32("stem.scale", 128) || !up_f32("stem.shift", 128)) return 1;
        for (int i = 0; i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i);
            if (!up_f32(b + ".bn1.scale", Fc) || !up_f32(b + ".bn1.shift", Fc) || !up_b16(b + ".conv1.w", 9LL * Fc * Fc) ||
                !up_f32(b + ".conv1.scale", Fc) || !up_f32(b + ".conv1.shift", Fc) || !up_b16(b + ".conv2.w", 9LL * Fc * Fc) ||
                !up_f32(b + ".conv2.bias", Fc)) return 1;
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
        const size_t lds = (size_t)((CONV_BM + 2 * CONV_HALO_MAX) * 16 + 128 * 16) * 16;
        hipLaunchKernelGGL((k_conv3x3<128, 128, 2, 2, 2, 2, 0>), dim3((M + CONV_BM - 1) / CONV_BM), dim3(256), lds, s, a);
        n_trunk_launches++;
    }

    void forward(hipStream_t s, const int8_t* in, float* policy, float* value, int n, bool timing) override {
        if (!loaded) return;                        // engine_create without weights: outputs stay as they are
        const int M = n * HW;
        StemArgs st; st.in = in; st.w = f32["stem.w"]; st.scale = f32["stem.scale"]; st.shift = f32["stem.shift"];
        st.scaleB = blocks ? f32["block0.bn1.scale"] : f32["stem.scale"]; st.shiftB = blocks ? f32["block0.bn1.shift"] : f32["stem.shift"];
        st.out1 = X; st.out2 = Aa; st.M = M; st.H = H; st.W = W;
        hipLaunchKernelGGL(k_stem, dim3((M + 63) / 64), dim3(256), 0, s, st);
        hipEvent_t e0 = 0, e1 = 0;
        if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); tev.push_back(e0); tev.push_back(e1); hipEventRecord(e0, s); }
        for (int i = 0; i < blocks; ++i) {
            const std::string b = "block" + std::to_string(i), nb = "block" + std::to_string(i + 1);
            conv_trunk(s, Aa, b16[b + ".conv1.w"], f32[b + ".conv1.scale"], f32[b + ".conv1.shift"], nullptr, Hh, ACT_RELU,
                       nullptr, nullptr, nullptr, M);
            const bool last = i + 1 == blocks;
            conv_trunk(s, Hh, b16[b + ".conv2.w"], nullptr, f32[b + ".conv2.bias"], X, X, ACT_NONE,
                       last ? nullptr : f32[nb + ".bn1.scale"], last ? nullptr : f32[nb + ".bn1.shift"], last ? nullptr : Aa, M);
        }
        if (timing) hipEventRecord(e1, s);
        {
            ConvArgs a; memset(&a, 0, sizeof(a));
            a.in = X; a.wgt = b16["heads.conv.w"]; a.shiftA = f32["heads.conv.bias"]; a.M = M; a.H = H; a.W = W;
            a.p_fs = f32["p.bn0.scale"]; a.p_ft = f32["p.bn0.shift"]; a.v_fs = f32["v.bn0.scale"]; a.v_ft = f32["v.bn0.shift"];
            a.p_feat = pfeat; a.v_feat = vfeat;
            const size_t lds = (size_t)((CONV_BM + 2 * CONV_HALO_MAX) * 16 + 32 * 16) * 16;
            hipLaunchKernelGGL((k_conv3x3<128, 32, 4, 1, 1, 1, 1>), dim3((M + CONV_BM - 1) / CONV_BM), dim3(256), lds, s, a);
        }
        const int F = HW * 8;
        hipLaunchKernelGGL(k_dense1, dim3((n + 7) / 8), dim3(128), (size_t)8 * F * 4, s, pfeat, f32["p.d1.w"], f32["p.d1.scale"],
                           f32["p.d1.shift"], pd1, n, F);
        hipLaunchKernelGGL(k_dense1, dim3((n + 7) / 8), dim3(128), (size_t)8 * F * 4, s, vfeat, f32["v.d1.w"], f32["v.d1.scale"],
                           f32["v.d1.shift"], vd1, n, F);
        TailArgs t; t.p_d1 = pd1; t.v_d1 = vd1; t.p_w2 = f32["p.d2.w"]; t.p_b2 = f32["p.d2.bias"]; t.p_w3 = f32["p.d3.w"];
        t.p_b3 = f32["p.d3.bias"]; t.v_w2 = f32["v.d2.w"]; t.v_b2 = f32["v.d2.bias"]; t.v_w3 = f32["v.d3.w"]; t.v_b3 = f32["v.d3.bias"];
        t.policy = policy; t.value = value; t.B = n; t.A = A; t.logits = logits;
        hipLaunchKernelGGL(k_tail, dim3(n), dim3(64), 0, s, t);
    }

    bool ready() const override { return loaded; }
    void timing_reset() override { for (auto e : tev) hipEventDestroy(e); tev.clear(); n_trunk_launches = 0; }
    void timing_get(double* ms, int64_t* launches) override {
        double t = 0;
        for (size_t i = 0; i + 1 < tev.size(); i += 2) { float a = 0; hipEventElapsedTime(&a, tev[i], tev[i + 1]); t += a; }
        *ms = t; *launches = (int64_t)(tev.size() / 2) * 2 * blocks;
    }
};

Evaluator* make_resnet_evaluator(const gaz_engine_config& cfg, int H, int W, int C, int A, std::string* err) {
    if (!(H == 6 && W == 7 && C == 4)) { *err = "the HIP ResNet evaluator is built for the Connect4 network in this version"; return nullptr; }
    if (cfg.net_filters != 128) { *err = "net_filters must be 128"; return nullptr; }
    if (cfg.net_blocks < 1 || cfg.net_blocks > 64) { *err = "net_blocks out of range"; return nullptr; }
    ResNetEvaluator* e = new ResNetEvaluator();
    e->H = H; e->W = W; e->C = C; e->A = A; e->HW = H * W; e->blocks = cfg.net_blocks; e->filters = 128; e->nmax = cfg.n_games;
    e->logits = cfg.policy_is_logits;
    const size_t M = (size_t)cfg.n_games * e->HW;
    e->X = e->dalloc<bf16_t>(M * 128 + 1024); e->Aa = e->dalloc<bf16_t>(M * 128 + 1024); e->Hh = e->dalloc<bf16_t>(M * 128 + 1024);
    e->pfeat = e->dalloc<float>((size_t)cfg.n_games * e->HW * 8); e->vfeat = e->dalloc<float>((size_t)cfg.n_games * e->HW * 8);
    e->pd1 = e->dalloc<float>((size_t)cfg.n_games * 128); e->vd1 = e->dalloc<float>((size_t)cfg.n_games * 128);
    if (!e->X || !e->Aa || !e->Hh || !e->pfeat || !e->vfeat || !e->pd1 || !e->vd1) { *err = "hipMalloc failed"; delete e; return nullptr; }
    // dynamic LDS above 64 KB needs the attribute
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 128, 2, 2, 2, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)(k_conv3x3<128, 32, 4, 1, 1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e;
}

}  // namespace gaz
