// trunk.hpp — the WHOLE residual trunk (every 128 -> 128 pre-activation block, Net/ResNet/ResNet_Block.py:27-41 as stacked by
// Connect4/Build_Model.py:25-29) as ONE kernel.  A convolution never looks across a board edge, so a tile made of whole boards
// depends on nothing outside itself through ALL the blocks: a workgroup takes k boards, keeps their activations in LDS from the
// stem output to the heads' input, and walks the blocks x 18 weight slices as one stream.  Against one k_resblock3 launch per
// block this removes, per block: the image load and its wait, the pre-activation pass, the halo recomputation (Connect4:
// 192 MFMA rows for 160 useful -> 128 for 126), the HBM round trip of x and the ramp / tail of a kernel boundary.
//
// Workgroup = 256 threads = 4 waves, wave (wm, wn) owns 32 TM board cells x 64 channels.  LDS: two swizzled bf16 images of
// 64 TM rows x 128 channels — Xs, the raw residual stream x, and As, the operand of the running convolution
// (relu(bn1(x)), then h = relu(bn2(conv1))) — one zero row for the board-edge taps, and two parameter sets (this block's
// and the next one's).  74.8 KB at TM = 2: two workgroups per CU, one's barriers hide behind the other's MFMAs.
//
// The MFMA operands are SWAPPED against k_resblock3 (weights as A, cells as B): a lane's 16 accumulators are then 4 groups of
// 4 consecutive CHANNELS of one cell, so h and x go back into the images as 8-byte LDS writes (4 per 32 x 32 tile instead of
// 16 two-byte ones) and the epilogue needs no fp32 staging tile: residual read, + bias, new x, next block's pre-activation,
// all at the lane's own image addresses.
// Numerics are those of the per-block path bit for bit (x is rounded to bf16 between blocks there too, and the arithmetic
// is written in the same order), which tests/test_evaluator_gpu.py checks.
#pragma once
#include "resblock.hpp"

namespace gaz {

struct TrunkArgs {
    const bf16_t* xin; bf16_t* xout;              // [M][128] raw stream: stem output in, heads' input out
    const bf16_t* w;                              // [nblocks][18 slices][8 k-steps][2][128][8], fragment order (arrange_conv_weights)
    const float* prm;                             // [nblocks][5][128]: bn1 scale, bn1 shift, conv1 scale, conv1 shift (bn2 folded), conv2 bias
    int M, H, W, nblocks;
    int tile_rows;                                // k * H * W, k whole boards, <= the workgroup's MFMA rows
    int n_big, small_rows;                        // k_trunk_mix: workgroups [0, n_big) take tile_rows each, the others small_rows
    unsigned long long* stamps;                   // diagnostic (GAZ_TRUNK_STAMPS, tools/trunk_stamps.py): [workgroup][128] or null
    // STEM: the stem convolution (k_stem_mfma's operands) computed straight into the images instead of reading xin
    const int8_t* planes; const uint4* stem_frag; const float* stem_shift;        // [M][4] int8; [6 k-steps][2][128] x 8 bf16 (hi | lo); [128]
    // fused tree + trunk launch (resnet.hip k_wave_trunk): board b's planes are valid once ready[b] has reached `epoch` (written by the tree team
    // of game b in this very launch); null = the planes were complete before the launch
    const unsigned* ready; unsigned epoch;
    // ... and the wait is BOUNDED: HIP promises no dispatch order, so a tree block of this launch may not be resident yet (or ever, while every
    // slot is held by a waiting trunk workgroup).  After spin_ticks of the 100-MHz wall clock — or at once when an earlier workgroup has already
    // given up (*fuse_fault != 0) — the workgroup leaves its boards unevaluated and counts itself in *fuse_fault; a workgroup that does take
    // its boards on writes `epoch` to eval_done[board], so the games of the others re-request (DevParams::eval_done) and the host falls back to
    // separate launches (engine.hip poll_fuse_fault).
    unsigned* eval_done; int* fuse_fault; unsigned spin_ticks;
    // completion queue (round 3; puct_core.hpp DevParams::done_queue): with queue != null board b of the batch is not game b but ENTRY b of the
    // launch's queue — whichever game finished b-th in rank-major order — so the first workgroups run on the games that are ready first.  The
    // workgroup waits (bounded, as above) for its entries' tags to reach `epoch`, then reads its planes from and writes its head features to
    // the rows of those games.  SKIPSET variants only (they map image rows to global rows themselves, see boff).
    const unsigned long long* queue;
    unsigned test_fault_mod;                        // test hook (gaz_engine_debug_fused_fault): workgroups with index % mod == 1 behave as if their wait had timed out
    // B0 (Gomoku, round 2): the FIRST block of the network — 256 stem channels -> 128 with a 1x1 projection on the skip path
    // (Net/ResNet/ResNet_Block.py:21-33) — runs inside this launch too, ahead of blocks 1..: x0 = raw stem output [M][256];
    // w then starts with block 0's 29 slices (conv1 channels 0-127 | 128-255, conv2, projection low | high); prm0 [1024] =
    // bn1 scale [256] | bn1 shift [256] | pad [128] | conv1 scale [128] | conv1 shift [128] | conv2 bias + projection bias [128]
    const bf16_t* x0; const float* prm0;
    // 16x16x32 build: which image row (cell of the tile) each MFMA row of the workgroup computes — [MFMA row] -> image row, a bijection, or
    // null = identity.  The host (tile_perm, resnet.hip) gathers the cells of a board edge into whole 16-row MFMA tiles: such a tile reads
    // nothing but zero padding on the three taps that look across its edge, and the kernel variants built with SKIPSET != 0 leave its MFMAs
    // out there (conv_taps_static; the host launches them only with permutations that deliver exactly the masks they assume).
    const uint8_t* perm; const uint8_t* perm_small;     // tile_rows-shaped tiles; small_rows-shaped tiles (k_trunk_mix)
    // ... and where the tile's boards sit in the LDS images: byte b = image row of cell 0 of board b (tile_perm.hpp TileLayout::boff; natural =
    // b * H * W).  Shifting a board by a row or two changes nothing for the convolution (a board's cells stay contiguous, neighbours are
    // row +- 1, +- W) but lets every permuted MFMA tile hold exactly two rows of each residue mod 8, the swizzle's conflict-free condition for the
    // fragment reads of EVERY tap (round 3: SQ_LDS_BANK_CONFLICT back from 35.6 M to the natural order's level).  Used by the SKIPSET variants only.
    unsigned boff, boff_small;
    // HEADS: the first convolution of both heads (k_conv_heads' operands) from the final image instead of writing xout
    const bf16_t* hw; const float* hbias;                                         // [9][8 k-steps][2][32][8]; [32]
    const float* p_fs; const float* p_ft; const float* v_fs; const float* v_ft; float* p_feat; float* v_feat;   // [HW * 8] flat BN; [B][HW * 8]
};

// what ResNetEvaluator::forward_trunk launches, as data: used by the fused tree + trunk launch (resnet.hip k_wave_trunk)
struct TrunkLaunchPlan { TrunkArgs args; int nwg; int mix; unsigned lds_bytes; };

// GELU = x * Phi(x), Phi from the Abramowitz-Stegun 7.1.26 erfc polynomial (|err| < 8e-8 on Phi; as bf16 the result differs from the erf form
// in 0.14 % of the elements of a normal sample, by one ulp).  ONE operation sequence in two forms — explicit fused multiply-adds (the build runs with
// -ffp-contract=off), v_exp_f32 and v_rcp_f32 — so that the scalar form (k_stem_mfma) and the packed form (two values per v_pk_* instruction: the
// in-kernel stem's epilogue, round 3: 17.7 k cycles of a workgroup's 264 k were this function) give the same bits.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float gelu_as(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    p = __builtin_fmaf(t, p, 1.421413741f); p = __builtin_fmaf(t, p, -0.284496736f); p = __builtin_fmaf(t, p, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f((x * x) * -1.44269504088896340736f);      // exp(-x^2)
    p = (0.5f * p) * e;                             // 0.5 * erfc(x)
    return v * (v < 0.0f ? p : 1.0f - p);
}
__device__ __forceinline__ f32x2_t gelu_as2(f32x2_t v) {
    const f32x2_t av = {fabsf(v.x), fabsf(v.y)};
    const f32x2_t x = av * 0.70710678118654752440f;
    const f32x2_t d = __builtin_elementwise_fma((f32x2_t)(0.3275911f), x, (f32x2_t)(1.0f));
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2_t p = __builtin_elementwise_fma(t, (f32x2_t)(1.061405429f), (f32x2_t)(-1.453152027f));
    p = __builtin_elementwise_fma(t, p, (f32x2_t)(1.421413741f)); p = __builtin_elementwise_fma(t, p, (f32x2_t)(-0.284496736f)); p = __builtin_elementwise_fma(t, p, (f32x2_t)(0.254829592f));
    p = p * t;
    const f32x2_t q = (x * x) * -1.44269504088896340736f;
    const f32x2_t e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    p = (p * 0.5f) * e;
    const f32x2_t one_m = (f32x2_t)(1.0f) - p;
    const f32x2_t ph = {v.x < 0.0f ? p.x : one_m.x, v.y < 0.0f ? p.y : one_m.y};
    return v * ph;
}
__device__ __forceinline__ unsigned s8x2_to_bf16x2(int lo, int hi) {      // two small integers -> packed bf16 (exact)
    return (__float_as_uint((float)lo) >> 16) | (__float_as_uint((float)hi) & 0xFFFF0000u);
}
// stamp i of wave 0: wall clock (100 MHz) at [i], shader clock at [64 + i].  0 start, 1 image landed, 2 block 0's operand ready,
// 3 + 6 b + {0 conv1 taps, 1 barrier, 2 h written + barrier, 3 conv2 taps, 4 barrier, 5 epilogue + barrier} for b < 10, 63 end
#define TR_STAMP(i) do { if (a.stamps && tid == 0) { a.stamps[(size_t)blockIdx.x * 128 + (i)] = wall_clock64(); a.stamps[(size_t)blockIdx.x * 128 + 64 + (i)] = clock64(); } } while (0)
constexpr int TR_THREADS = 256, TR_PRM = 5 * 128;
constexpr int TR_ZROWS = 16;                        // zero rows behind the operand image, see trunk_tile
constexpr int TR_XTRA = 64;                          // bytes behind the parameter sets: the "gave up" word and the tile's game indices (completion queue)
constexpr size_t trunk_lds_bytes(int rows, bool resg = false) { return (size_t)((resg ? 1 : 2) * rows + TR_ZROWS) * 256 + 2 * TR_PRM * 4 + TR_XTRA; }

// Image swizzle: 16-byte slot of k-group g (channels 8 g ..+7) in image row `row`.
//  * 32x32x16 build: g ^ (row & 15).  A ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
//    and the same + 32 (MI355X_MICROARCH.md, LDS) — which with lane = (row l & 31, k-half l >> 5) are 16 rows of ONE k-group: 16 slots.
//  * 16x16x32 build: lane = (row l & 15, k-group l >> 4), so a hardware group holds 8 rows of k-group 2 h and the 8 other rows of
//    k-group 2 h + 1.  With the XOR swizzle the two halves collide whenever the tile's first row is odd, i.e. on the four taps with
//    an odd offset: 8 LDS cycles per read instead of 4 (SQ_LDS_BANK_CONFLICT per launch, with the zero block below in both builds:
//    28.9 M conflict cycles with the XOR swizzle, 9.6 M with this one; 46-49 M before either fix).  Here the k-group's
//    low bit picks the half of the row and the 3 remaining bits are XORed with row & 7: any 8 consecutive rows spread over the 8 slots
//    of a half, whatever the first row — conflict-free for every tap.
template <bool M16> __device__ __forceinline__ int swz_slot(int g, int row) { return M16 ? (((g & 1) << 3) | ((g >> 1) ^ (row & 7))) : (g ^ (row & 15)); }
template <bool M16> __device__ __forceinline__ int swz_inv(int sp, int row) { return M16 ? ((((sp & 7) ^ (row & 7)) << 1) | (sp >> 3)) : (sp ^ (row & 15)); }

// TM = 32-cell tiles per wave, WN = waves across the 128 channels (2: wave = 32 TM cells x 64 channels, two waves down the cells;
// 4: wave = 32 TM cells x 32 channels, every wave all the cells of the tile — the workgroup then pulls each weight fragment from
// L2 once instead of twice, for twice the LDS fragment reads per MFMA)
// RESG: no x image in LDS — the residual stream goes through global memory (xin for the first block, xout afterwards; each lane
// re-reads exactly the 8-byte groups it wrote one block earlier), which halves the LDS footprint: the 256-row tile of a Gomoku
// board (TM = 4: wave = 128 cells x 64 channels, half the weight bytes per MFMA of the TM = 2 shape) still fits twice on a CU.
// NW = waves per workgroup (4; 8 for the 256-row Gomoku tile: WM = 4 waves down the cells, one workgroup per CU with both images in LDS).
// S0 (round 3, with B0): the 256-channel STEM of the Gomoku network (Gomoku/Build_Model.py:21-24: Conv3x3 2 -> 256, BN, ReLU; k_stem_mfma's
// arithmetic) is computed inside this launch as well, half by half, straight into the operand image — no stem kernel, no 236-MB stem tensor
// written and read back four times, and the launch depends on nothing but the int8 planes: it can share a launch with the tree step.
template <int TM, int WN, int RING, bool STEM, bool HEADS, bool RESG = false, bool M16 = false, int NW = 4, bool B0 = false, int SKIPSET = 0, bool S0 = false>
__device__ __forceinline__ void trunk_tile(const TrunkArgs& a, const long m0_, const int tile_rows, const uint8_t* perm = nullptr, const unsigned boffp = 0) {
    static_assert(!S0 || (B0 && TM == 2 && WN == 2), "in-kernel 256-channel stem: block-0 variant, 64 x 64 wave tile");
    long m0 = m0_;                                  // S0 + completion queue: replaced by the queue entry's game below
    static_assert(!B0 || (M16 && !STEM && !HEADS && !RESG), "block 0 inside the launch: 16x16x32 build with both images in LDS");
    static_assert(SKIPSET == 0 || (STEM && HEADS && M16), "board offsets in the image: only where the kernel itself maps image rows to global rows");
    constexpr int SL0 = B0 ? 29 : 0;                // weight slices of block 0 ahead of the regular blocks' 18 each
    constexpr int BN = 128, SLOTS = 16, WM = NW / WN, TN = 4 / WN, KS = 8, ROWS = 32 * TM * WM, ZROW = ROWS, BSL = BN * SLOTS, THREADS = 64 * NW;
    static_assert(NW == 4 || (!STEM && !HEADS), "stem / heads phases assume four waves");
    static_assert(KS % RING == 0, "ring slot must not depend on the tap");
    extern __shared__ uint4 lds[];
    uint4* As = lds;                              // operand image, rows [0, ROWS) + the zero row
    // Board-edge taps read zeros.  ONE zero row would make every masked lane of a ds_read_b128 lane group hit the same bank as
    // some unmasked lane of the group (a fifth of all (cell, tap) pairs of a 6 x 7 board is masked: SQ_LDS_BANK_CONFLICT showed as
    // many conflict cycles as useful ones); sixteen zero rows let a masked lane read row (its row & 15) of the zero block, i.e. the
    // banks its own row would have used, so the group's access pattern stays the conflict-free one.
    uint4* Xs = lds + (ROWS + TR_ZROWS) * SLOTS;  // raw stream (not with RESG)
    float* Ps = reinterpret_cast<float*>(Xs + (RESG ? 0 : ROWS * SLOTS));      // [2][5][128]
    volatile int* Xt = reinterpret_cast<volatile int*>(Ps + 2 * TR_PRM);       // [TR_XTRA / 4]: [0] gave up, [1..3] game index of the tile's boards (completion queue)
    static_assert(!RESG || (!STEM && !HEADS), "the stem / heads phases work on the x image");
    constexpr int NB = TM >= 4 ? 2 : 4, PD = NB - 1;   // operand-fragment buffers and prefetch distance in k-steps
    char* Ab = reinterpret_cast<char*>(As);
    char* Xb = reinterpret_cast<char*>(Xs);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, lhi = lane >> 5;
    const int HW = a.H * a.W;
    // Divisions by the board's width / size are multiplications by a 16-bit reciprocal, formed ONCE (exact for n < 65536 / d: cells and image rows
    // are < 272).  Round 3: as plain `/` and `%` on run-time values the prologue, the stem and the heads held ~40 64-bit scalar divisions (m0 / HW, one
    // per locate() call: ~200 scalar instructions each) and ~45 32-bit vector ones — a third of the stem phase's instructions.
    // The three values live in LDS behind the parameter sets (Xt[4..6], written by thread 0 in front of the first barrier) and are read where they
    // are used: held in scalar registers across the block loop they cost the edge-tile variants spilled registers (scratch traffic: 67 -> 82 MB
    // of HBM traffic per launch by counter).
    auto cell_yx = [&](const int cell, int& y, int& x) { y = (int)(((unsigned)cell * (unsigned)Xt[4]) >> 16); x = cell - y * a.W; };
    if (tid == 0) {
        Xt[4] = (int)((65536u + (unsigned)a.W - 1u) / (unsigned)a.W); Xt[5] = (int)((65536u + (unsigned)HW - 1u) / (unsigned)HW);
        Xt[6] = (int)((unsigned)m0_ / (unsigned)HW);            // first board of the tile (natural order; the completion queue names games instead)
    }
    // image row -> (cell of its board, global row); false for padding rows and rows beyond the batch.  Natural layout: image row q <-> global row
    // m0 + q; SKIPSET variants: board b of the tile starts at image row byte b of boffp (TrunkArgs::boff)
    auto locate = [&](const int row, int& cell, long& grow, const unsigned bo = 0xFFFFFFFFu) -> bool {      // bo: see the heads phase
        const unsigned boffq = bo == 0xFFFFFFFFu ? boffp : bo;
        if constexpr (SKIPSET == 0) {
            cell = row - (int)(((unsigned)row * (unsigned)Xt[5]) >> 16) * HW; grow = m0 + row;
            return row < tile_rows && grow < a.M;
        } else {
            bool ok = false; cell = 0; grow = 0;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int o = (int)((boffq >> (8 * b)) & 0xFFu);
                if (b * HW < tile_rows && row >= o && row < o + HW) {
                    cell = row - o;
                    // completion queue: board b of the tile is the game its queue entry names (Xt[1 + b], written before the barrier below; -1 = no entry)
                    const long gb = a.queue ? (long)Xt[1 + b] : (long)(Xt[6] + b);
                    ok = gb >= 0; grow = gb * HW + cell;
                }
            }
            return ok && grow < a.M;
        }
    };
    const uint4* in4 = reinterpret_cast<const uint4*>(a.xin);
    const int last_slice = SL0 + a.nblocks * 18 - 1;
    TR_STAMP(0);

    constexpr int n_slots = ROWS * SLOTS;
    // B0: image row q <-> global row m0 + q of channels [128 half, 128 half + 128) of the 256-channel stem output, into the OPERAND image
    auto load_x0_half = [&](int half) {
        const uint4* x04 = reinterpret_cast<const uint4*>(a.x0);
        for (int base = wave * 64; base < n_slots; base += THREADS) {
            const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
            long gr = m0 + lr;
            gr = gr >= a.M ? (long)a.M - 1 : gr;
            __builtin_amdgcn_global_load_lds((const void*)(x04 + gr * (2 * SLOTS) + half * SLOTS + swz_inv<M16>(sp, lr)), (lds_ptr_t)(As + base), 16, 0, 0);
        }
    };
    if (B0 && !S0) load_x0_half(0);
    if (!STEM && !B0) {
    // ---- raw rows of the tile -> Xs by LDS-DMA, swizzled through the source address (image row q <-> global row m0 + q)
    for (int base = wave * 64; base < n_slots; base += THREADS) {
        const int i = base + lane, lr = i / SLOTS, sp = i % SLOTS;
        long gr = m0 + lr;
        gr = gr >= a.M ? (long)a.M - 1 : gr;
        __builtin_amdgcn_global_load_lds((const void*)(in4 + gr * SLOTS + swz_inv<M16>(sp, lr)), (lds_ptr_t)((RESG ? As : Xs) + base), 16, 0, 0);
    }
    }
    if (tid < TR_ZROWS * SLOTS) As[ZROW * SLOTS + tid] = make_uint4(0, 0, 0, 0);
    const float4* prm4 = reinterpret_cast<const float4*>(a.prm);
    float4* Ps4 = reinterpret_cast<float4*>(Ps);
    if (B0) { for (int i = tid; i < 256; i += THREADS) Ps4[i] = reinterpret_cast<const float4*>(a.prm0)[i]; }   // block 0's parameters fill both sets
    else if (tid < TR_PRM / 4) Ps4[tid] = prm4[tid];
    if (STEM && tid < 32) Ps4[TR_PRM / 4 + tid] = reinterpret_cast<const float4*>(a.stem_shift)[tid];      // the idle parameter set holds the stem's shift

    // B ring as in k_resblock3: fragment of global k-step g = slice * 8 + ks in bfr[g % RING]; the slices of ALL blocks are one array
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (last_slice + 1) * BSL * 16, 0x00020000);
    const int col0 = wn * (32 * TN) + l31;          // + 32 * tn: the channel this lane feeds into the weight operand
    const int bvo = (lhi * BN + col0) * 16;
    auto ldb = [&](int slice, int ks, int tn) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo, ((slice * BSL) + ks * 2 * BN + tn * 32) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 bfr[RING][TN];
    if (!M16)
#pragma unroll
    for (int g = 0; g < RING; ++g)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bfr[g][tn] = ldb(0, g, tn);

    // per-lane geometry: cell lrow[tm] of the tile is this lane's column of the cell operand (its 9 tap-validity bits: behind the barrier below)
    int lrow[TM]; unsigned vmask[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) lrow[tm] = (wm * TM + tm) * 32 + l31;
    // byte offset of this lane's 8-byte group j (channels 8 cslot + 4 lhi ..+3, cslot = (wn TN + tn) 4 + j) of cell lrow[tm] in either image
    auto img_off = [&](int tm, int tn, int j) -> int {
        const int row = lrow[tm], cslot = (wn * TN + tn) * 4 + j;
        return row * 256 + (swz_slot<M16>(cslot, row) << 4) + lhi * 8;
    };
    // fused launch: "this workgroup's wait ran out" (Xt[0]), workgroup-uniform after the barrier
    if ((STEM || S0) && (a.ready || a.queue)) {     // wait for the tree teams of this tile's boards (see TrunkArgs::ready / queue), for a bounded time
        if (tid == 0) Xt[0] = 0;                    // tid 0 and the pollers (tid < boards per tile <= 3) are lanes of wave 0: LDS accesses of one wave are in order
        const long b = (long)((unsigned)m0_ / (unsigned)HW) + tid;          // board of the batch = done flag index, or queue entry (in front of the barrier: not from Xt[6])
        if (tid < tile_rows / HW) {
            int game = -1;
            if (b * HW < a.M) {
                bool fail = a.test_fault_mod && blockIdx.x % a.test_fault_mod == 1;
                auto arrived = [&]() -> bool {
                    if ((SKIPSET != 0 || S0) && a.queue) {
                        const unsigned long long v = __hip_atomic_load(a.queue + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        game = (int)(unsigned)v;
                        return (unsigned)(v >> 32) == a.epoch;
                    }
                    game = (int)b;
                    return (int)(__hip_atomic_load(a.ready + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - a.epoch) >= 0;
                };
                if (!fail && !arrived()) {
                    const unsigned long long t0 = wall_clock64();
                    fail = a.fuse_fault && __hip_atomic_load(a.fuse_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    while (!fail && !arrived()) {
                        __builtin_amdgcn_s_sleep(16);
                        fail = wall_clock64() - t0 > (unsigned long long)a.spin_ticks;
                    }
                }
                if (fail) { Xt[0] = 1; game = -1; }
                else if (S0) {                      // the row was written with plain stores and released (puct_core.hpp publish_done): acquire, so that
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // this CU's L1 holds no older copy; the barrier below holds the other waves back
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            Xt[1 + tid] = game;
        }
        asm volatile("" ::: "memory");              // no plane load may be hoisted above the poll
    }
    __syncthreads();                                // Xs, block 0's parameters and the zero row landed
    if ((STEM || S0) && (a.ready || a.queue)) {
        if (__builtin_amdgcn_readfirstlane(Xt[0])) {     // (scalar branch) leave the tile out: no board of it is marked, their games re-request (DevParams::eval_done)
            if (tid == 0 && a.fuse_fault) atomicAdd(a.fuse_fault, 1);
            return;
        }
        if (tid < tile_rows / HW && Xt[1 + tid] >= 0 && a.eval_done) a.eval_done[Xt[1 + tid]] = a.epoch;      // this launch evaluates these games (read by the NEXT launch's tree step)
        if (S0 && a.queue) m0 = (long)__builtin_amdgcn_readfirstlane(Xt[1]) * HW;      // one board per tile: the queue entry's game IS the tile's global row base
    }
    // (behind the barrier: with the completion queue, which games — hence which rows are valid — is only known now)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        unsigned mm = 0;
        int cell; long grow_;
        if (locate(lrow[tm], cell, grow_)) {
            int y, x; cell_yx(cell, y, x);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << t;
            }
        }
        vmask[tm] = mm;
    }
    TR_STAMP(1);

    f32x16 acc[TM][TN];
    if (STEM) {
        // ---- stem (Connect4/Build_Model.py:22-24: Conv3x3 4 -> 128, BN, GELU) as in k_stem_mfma: k = tap * 4 + plane padded to 48, the
        // fp32 weights (BN scale folded in) as bf16 hi + lo halves against the same exact int8 activations; x and block 0's operand
        // go straight into the images
        const int* in32 = reinterpret_cast<const int*>(a.planes);
        uint4 sw[6][TN];
#pragma unroll
        for (int ks2 = 0; ks2 < 6; ++ks2)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) sw[ks2][tn] = a.stem_frag[(ks2 * 2 + lhi) * 128 + col0 + tn * 32];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int cell_; long gr;
            const bool rok = locate(lrow[tm], cell_, gr);
            uint4 cf[3];
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                int pl[8];                          // this lane's 8 activations: taps 4 ks + 2 lhi + {0, 1}, 4 planes each
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tap = ks * 4 + lhi * 2 + h;
                    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                    int packed = 0;
                    if (rok && tap < 9 && ((vmask[tm] >> tap) & 1u))   // fused launch: the row was written during this launch -> read it where it was written to
                        packed = (a.ready || a.queue) ? __hip_atomic_load(in32 + (gr + dy * a.W + dx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : in32[gr + dy * a.W + dx];
#pragma unroll
                    for (int c = 0; c < 4; ++c) pl[h * 4 + c] = (int)(int8_t)((packed >> (8 * c)) & 0xFF);
                }
                cf[ks] = make_uint4(s8x2_to_bf16x2(pl[0], pl[1]), s8x2_to_bf16x2(pl[2], pl[3]), s8x2_to_bf16x2(pl[4], pl[5]), s8x2_to_bf16x2(pl[6], pl[7]));
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
#pragma unroll
                for (int ks2 = 0; ks2 < 6; ++ks2)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&sw[ks2][tn]), *reinterpret_cast<const bf16x8*>(&cf[ks2 % 3]), acc[tm][tn], 0, 0, 0);
            }
        }
        const float* SH = Ps + TR_PRM;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = ((wn * TN + tn) * 4 + j) * 8 + 4 * lhi, o = img_off(tm, tn, j);
                    const float4 sh = *reinterpret_cast<const float4*>(&SH[c0]);
                    const float4 s = *reinterpret_cast<const float4*>(&Ps[c0]);
                    const float4 t = *reinterpret_cast<const float4*>(&Ps[128 + c0]);
                    const f32x2_t g01 = gelu_as2(f32x2_t{acc[tm][tn][4 * j + 0] + sh.x, acc[tm][tn][4 * j + 1] + sh.y});
                    const f32x2_t g23 = gelu_as2(f32x2_t{acc[tm][tn][4 * j + 2] + sh.z, acc[tm][tn][4 * j + 3] + sh.w});
                    const float v0 = g01.x, v1 = g01.y, v2 = g23.x, v3 = g23.y;
                    const uint2 xn = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                    *reinterpret_cast<uint2*>(Xb + o) = xn;
                    const float a0 = fmaxf(__uint_as_float(xn.x << 16) * s.x + t.x, 0.0f), a1 = fmaxf(__uint_as_float(xn.x & 0xFFFF0000u) * s.y + t.y, 0.0f);
                    const float a2 = fmaxf(__uint_as_float(xn.y << 16) * s.z + t.z, 0.0f), a3 = fmaxf(__uint_as_float(xn.y & 0xFFFF0000u) * s.w + t.w, 0.0f);
                    *reinterpret_cast<uint2*>(Ab + o) = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
                }
    } else if (B0) {
        // the operand of conv1 is made further down (preact_half), half by half
    } else {
    // ---- block 0's operand: As = relu(x * s1 + t1)
    {
        const int tch0 = swz_inv<M16>(tid % SLOTS, tid / SLOTS) * 8;      // slot i = tid + THREADS it keeps sp and (lr & 15)
        float ps1[8], pt1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ps1[j] = Ps[tch0 + j]; pt1[j] = Ps[128 + tch0 + j]; }
        for (int i = tid; i < n_slots; i += THREADS) {
            const uint4 v = RESG ? As[i] : Xs[i];
            unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = fmaxf(__uint_as_float(w[j] << 16) * ps1[2 * j] + pt1[2 * j], 0.0f);
                const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * ps1[2 * j + 1] + pt1[2 * j + 1], 0.0f);
                w[j] = pack_bf16(lo, hi);
            }
            As[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    }
    __syncthreads();
    TR_STAMP(2);

    if constexpr (M16) {
    // ---- the blocks on v_mfma_f32_16x16x32_bf16.  Same FLOPs, same operand bytes and the same cycles per FLOP as 32x32x16, but the chip
    // holds a higher clock on this shape (MI355X_MICROARCH.md 'DVFS give-back' item 7): measured here on a timing build with identical
    // operand traffic, 1.70 -> 1.95 GHz in the kernel and -10 % kernel time at +5.6 % cycles.  Wave tile = NC 16-cell tiles x NCH
    // 16-channel tiles; lane (l15, lq): operand row / column l15, k-group lq of the 32-wide k-step; D: cell l15, channels 4 lq ..+3, so
    // the images still take 8-byte groups.  One accumulation order per output element whatever the tile shape.
    static_assert(!M16 || !RESG, "16x16x32 path keeps x in LDS");
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int NC = 2 * TM, NCH = 2 * TN, KS32 = 4;
    const int l15 = lane & 15, lq = lane >> 4;
    const int bvo16 = (lq * BN + wn * (32 * TN) + l15) * 16;
    auto ldw = [&](int slice, int ks32, int ct) -> uint4 {
        const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(wrs, bvo16, ((slice * BSL) + ks32 * 4 * BN + ct * 16) * 16, 0);
        return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
    };
    uint4 wfr[KS32][NCH];                           // weight ring: one tap (four k-steps) ahead
#pragma unroll
    for (int g = 0; g < KS32; ++g)
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) wfr[g][ct] = ldw(0, g, ct);
    // crow (image row, < 256) and the nine tap-validity bits share one register per MFMA tile: they are live across the whole block loop, and the
    // edge-tile variants have no register to spare (round 3: packing them is what keeps the queue / board-offset geometry from spilling)
    unsigned cgeo[NC];
    auto crow_of = [&](int t) -> int { return (int)(cgeo[t] & 0xFFu); };
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const int mrow = wm * TM * 32 + t * 16 + l15;
        const int crow_t = perm ? (int)perm[mrow] : mrow;
        unsigned mm = 0;
        int cell; long grow_;
        if (locate(crow_t, cell, grow_)) {
            int y, x; cell_yx(cell, y, x);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int dy = q / 3 - 1, dx = q % 3 - 1;
                mm |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << q;
            }
        }
        cgeo[t] = (unsigned)crow_t | (mm << 8);
    }
    auto off16 = [&](int ct, int t) -> int {        // this lane's 8-byte group (channels 16 ct + 4 lq ..+3 of the wave's slab) of cell crow[t]
        const int row = crow_of(t), cslot = wn * TN * 4 + ct * 2 + (lq >> 1);
        return row * 256 + (swz_slot<true>(cslot, row) << 4) + (lq & 1) * 8;
    };
    f32x4 acc16[NCH][NC];
    // fragment reads by absolute LDS address: one v_xor per read (base ^ k-step bits; the image starts on a 256-byte boundary), where
    // pointer arithmetic on the shared array costs an extra add of its link-time base per read — issue slots the MFMAs' shadows lack
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4_t;
    const unsigned ldsb = (unsigned)(size_t)(lds_ptr_t)As;
    if (ldsb & 255u) __builtin_trap();
    int pb[NC], pbn[NC];                            // byte address of k-group lq of k-step 0 of this lane's operand row
    auto tap_rows = [&](int tap, int (&o)[NC], int zz = 0) {       // zz: an opaque 0 (conv_taps_static)
        const int ty = tap / 3, off = (ty - 1) * a.W + (tap - ty * 3 - 1);
#pragma unroll
        for (int t = 0; t < NC; ++t) {
            const bool ok = (cgeo[t] >> (8 + tap)) & 1u;
            const int ar = ok ? crow_of(t) + off : ZROW + ((crow_of(t) + off) & 15);
            o[t] = ((int)ldsb | zz) + ar * 256 + (swz_slot<true>(lq, ar) << 4);   // k-step ks: ^ (ks << 5), see below
        }
    };
    u32x4_t cfr[2][NC];                             // cell fragments, one k-step (256 MFMA cycles) ahead, across the tap boundary
    auto zero_acc16 = [&]() {
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct)
#pragma unroll
            for (int t = 0; t < NC; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[ct][t][r] = 0.0f;
    };
    // taps [tap0, tap0 + ntaps) of the 3x3 stencil over the operand image, weight slice sl0 + i for the i-th of them; accumulates
    auto conv_taps = [&](const int sl0, const int tap0, const int ntaps) {
        tap_rows(tap0, pb);
#pragma unroll
        for (int t = 0; t < NC; ++t) cfr[0][t] = *(lds_u4_t)(unsigned)pb[t];
#pragma unroll 1
        for (int i = 0; i < ntaps; ++i) {
            const int sl = sl0 + i;
            const int nsl = sl < last_slice ? sl + 1 : sl;
            tap_rows(i + 1 < ntaps ? tap0 + i + 1 : tap0 + i, pbn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
#pragma unroll
                for (int ct = 0; ct < NCH / 2; ++ct)
#pragma unroll
                    for (int t = 0; t < NC; ++t)
                        acc16[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wfr[ks][ct]), *reinterpret_cast<bf16x8*>(&cfr[ks & 1][t]), acc16[ct][t], 0, 0, 0);
                // the next k-step's reads in the middle of this one's MFMAs: this k-step's fragments are still live, so they cannot
                // be allocated over them (behind the last MFMA the scheduler does exactly that: prefetch distance zero), and they
                // issue in the shadow of the MFMAs above.  (Round 3: in FRONT of the k-step's MFMAs — a whole k-step of distance — the kernel
                // alone is unchanged, 431 us, and the wave of two game groups gets slower, 462 vs 455 us: not kept)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NC; ++t)
                    cfr[(ks + 1) & 1][t] = *(lds_u4_t)(unsigned)((ks + 1 < KS32 ? pb[t] : pbn[t]) ^ (((ks + 1) % KS32) << 5));      // slot of k-group 4 ks + lq = slot of lq ^ (2 ks)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = NCH / 2; ct < NCH; ++ct)
#pragma unroll
                    for (int t = 0; t < NC; ++t)
                        acc16[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wfr[ks][ct]), *reinterpret_cast<bf16x8*>(&cfr[ks & 1][t]), acc16[ct][t], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct) wfr[ks][ct] = ldw(nsl, ks, ct);    // the next slice's k-step ks; the very last slice re-reads itself
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NC; ++t) pb[t] = pbn[t];
        }
    };
    // The same nine taps with the taps unrolled and, per tap, the MFMA tiles that sit it out known at compile time (M0 / M1: bit q = MFMA
    // tile 0 / 1 of this wave reads only zero padding on tap q, see TrunkArgs::perm): straight-line code, no MFMAs and no fragment reads
    // for a tile on the taps it sits out.  (Run-time branches around the MFMAs cost more than the skipped MFMAs give: 524 vs 490 us.)
    auto conv_taps_static = [&](auto m0c, auto m1c, auto m2c, const int sl0) {
        constexpr unsigned M0 = decltype(m0c)::value, M1 = decltype(m1c)::value, M2 = decltype(m2c)::value;
        // with the taps unrolled the nine sets of fragment addresses are invariants of the block loop, and hoisted out of it they cost 36
        // registers the tap loop does not have (191 spilled): an opaque zero ties them to this call
        int zz = 0;
        asm volatile("" : "+v"(zz));
        tap_rows(0, pb, zz);
#pragma unroll
        for (int t = 0; t < NC; ++t) cfr[0][t] = *(lds_u4_t)(unsigned)pb[t];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sl = sl0 + tap;
            const int nsl = sl < last_slice ? sl + 1 : sl;
            tap_rows(tap < 8 ? tap + 1 : 8, pbn, zz);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
#pragma unroll
                for (int ct = 0; ct < NCH / 2; ++ct)
#pragma unroll
                    for (int t = 0; t < NC; ++t)
                        if (!(t == 0 && ((M0 >> tap) & 1u)) && !(t == 1 && ((M1 >> tap) & 1u)) && !(t == 2 && ((M2 >> tap) & 1u)))
                            acc16[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wfr[ks][ct]), *reinterpret_cast<bf16x8*>(&cfr[ks & 1][t]), acc16[ct][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NC; ++t) {
                    const int ntap = ks + 1 < KS32 ? tap : (tap < 8 ? tap + 1 : 8);       // the tap the fragment is for
                    if (!(t == 0 && ((M0 >> ntap) & 1u)) && !(t == 1 && ((M1 >> ntap) & 1u)) && !(t == 2 && ((M2 >> ntap) & 1u)))
                        cfr[(ks + 1) & 1][t] = *(lds_u4_t)(unsigned)((ks + 1 < KS32 ? pb[t] : pbn[t]) ^ (((ks + 1) % KS32) << 5));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = NCH / 2; ct < NCH; ++ct)
#pragma unroll
                    for (int t = 0; t < NC; ++t)
                        if (!(t == 0 && ((M0 >> tap) & 1u)) && !(t == 1 && ((M1 >> tap) & 1u)) && !(t == 2 && ((M2 >> tap) & 1u)))
                            acc16[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wfr[ks][ct]), *reinterpret_cast<bf16x8*>(&cfr[ks & 1][t]), acc16[ct][t], 0, 0, 0);
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct) wfr[ks][ct] = ldw(nsl, ks, ct);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NC; ++t) pb[t] = pbn[t];
        }
    };
    if constexpr (B0) {
        // ---- block 0 (256 -> 128 with a projection): x_out = conv2(relu(bn2(conv1(relu(bn1(x0)))))) + proj(x0) + (b2 + bp).  The 256-channel
        // operands pass through the 128-channel operand image in halves; conv2 and the projection share ONE accumulator.
        auto preact_half = [&](int half) {          // in place: slot sp of row lr holds channels 128 half + 8 swz_inv(sp, lr) ..+7
            for (int i = tid; i < n_slots; i += THREADS) {
                const int lr = i / SLOTS, sp = i % SLOTS, c0 = half * 128 + swz_inv<true>(sp, lr) * 8;
                const uint4 v = As[i];
                unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = fmaxf(__uint_as_float(w[j] << 16) * Ps[c0 + 2 * j] + Ps[256 + c0 + 2 * j], 0.0f);
                    const float hi = fmaxf(__uint_as_float(w[j] & 0xFFFF0000u) * Ps[c0 + 2 * j + 1] + Ps[256 + c0 + 2 * j + 1], 0.0f);
                    w[j] = pack_bf16(lo, hi);
                }
                As[i] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        };
        // S0: this lane's stem activations (taps x 2 planes of cells lrow[0], lrow[1]; exact small integers as bf16) are the same for all four
        // half-image passes and for both waves of a wave row: made once, parked in LDS behind the parameter sets.  K = 18 padded to two k-steps
        // of 16: k = tap * 2 + plane, lane half lhi holds k-groups of 8 (k_stem_mfma's operand layout, CIN = 2).
        uint4* Cf = reinterpret_cast<uint4*>(reinterpret_cast<char*>(const_cast<int*>(Xt)) + TR_XTRA);    // [wave row][tm][k-step][lane]
        if constexpr (S0) {
            const unsigned short* in16 = reinterpret_cast<const unsigned short*>(a.planes);
            if (wn == 0) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    int cell_; long gr;
                    const bool rok = locate(lrow[tm], cell_, gr);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        int pl[8];
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const int tap = ks * 8 + lhi * 4 + h;
                            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                            int packed = 0;
                            if (rok && tap < 9 && ((vmask[tm] >> tap) & 1u))
                                packed = (int)in16[gr + dy * a.W + dx];       // (fused launch: behind the poller's acquire + the barrier)
                            pl[h * 2] = (int)(int8_t)(packed & 0xFF); pl[h * 2 + 1] = (int)(int8_t)((packed >> 8) & 0xFF);
                        }
                        Cf[((wm * TM + tm) * 2 + ks) * 64 + lane] = make_uint4(s8x2_to_bf16x2(pl[0], pl[1]), s8x2_to_bf16x2(pl[2], pl[3]), s8x2_to_bf16x2(pl[4], pl[5]), s8x2_to_bf16x2(pl[6], pl[7]));
                    }
                }
            }
            __syncthreads();
        }
        // one 128-channel half of the stem output over the operand image: raw (the projection's operand) or pre-activated with block 0's bn1
        // (conv1's operand) — bit for bit what k_stem_mfma writes and load_x0_half / preact_half then bring in
        auto stem_half = [&](const int half, const bool preact) {
            uint4 sw[4][TN];                        // weight fragments: k-steps hi0, hi1, lo0, lo1 (stem_fragments: [k-step][k-half][256 channels])
#pragma unroll
            for (int ks2 = 0; ks2 < 4; ++ks2)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) sw[ks2][tn] = a.stem_frag[(ks2 * 2 + lhi) * 256 + half * 128 + wn * (32 * TN) + tn * 32 + l31];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const uint4 c0 = Cf[((wm * TM + tm) * 2 + 0) * 64 + lane], c1 = Cf[((wm * TM + tm) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    f32x16 sacc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
                    for (int ks2 = 0; ks2 < 4; ++ks2)
                        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&sw[ks2][tn]), *reinterpret_cast<const bf16x8*>((ks2 & 1) ? &c1 : &c0), sacc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = half * 128 + ((wn * TN + tn) * 4 + j) * 8 + 4 * lhi;       // channel of the 256
                        const float4 sh = *reinterpret_cast<const float4*>(a.stem_shift + c);
                        const float v0 = fmaxf(sacc[4 * j + 0] + sh.x, 0.0f), v1 = fmaxf(sacc[4 * j + 1] + sh.y, 0.0f);
                        const float v2 = fmaxf(sacc[4 * j + 2] + sh.z, 0.0f), v3 = fmaxf(sacc[4 * j + 3] + sh.w, 0.0f);
                        uint2 xn = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                        if (preact) {
                            const float4 sc = *reinterpret_cast<const float4*>(&Ps[c]);
                            const float4 t4 = *reinterpret_cast<const float4*>(&Ps[256 + c]);
                            const float a0 = fmaxf(__uint_as_float(xn.x << 16) * sc.x + t4.x, 0.0f), a1 = fmaxf(__uint_as_float(xn.x & 0xFFFF0000u) * sc.y + t4.y, 0.0f);
                            const float a2 = fmaxf(__uint_as_float(xn.y << 16) * sc.z + t4.z, 0.0f), a3 = fmaxf(__uint_as_float(xn.y & 0xFFFF0000u) * sc.w + t4.w, 0.0f);
                            xn = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
                        }
                        *reinterpret_cast<uint2*>(Ab + img_off(tm, tn, j)) = xn;
                    }
                }
            }
        };
        // five passes over the operand image, ONE copy of the tap loop: conv1 over the pre-activated low / high input half, conv2 over
        // h, the skip path's 1x1 projection (centre tap) over the raw low / high half
#pragma unroll 1
        for (int ph = 0; ph < 5; ++ph) {
            if (ph > 0) __syncthreads();            // every wave is done with the previous image
            if (S0 && ph != 2) { stem_half(ph == 0 || ph == 3 ? 0 : 1, ph < 2); }
            if (ph == 2) {                          // block 0's bn1 vectors (parameter set 0) are free: the first regular block's parameters
                if (tid < TR_PRM / 4) Ps4[tid] = prm4[tid];
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct) {  // h = relu(acc * s2 + t2) as bf16 over the operand image
                    const int c0 = wn * TN * 32 + ct * 16 + 4 * lq;
                    const float4 sc = *reinterpret_cast<const float4*>(&Ps[640 + c0]);
                    const float4 t4 = *reinterpret_cast<const float4*>(&Ps[768 + c0]);
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        const float v0 = fmaxf(acc16[ct][t][0] * sc.x + t4.x, 0.0f), v1 = fmaxf(acc16[ct][t][1] * sc.y + t4.y, 0.0f);
                        const float v2 = fmaxf(acc16[ct][t][2] * sc.z + t4.z, 0.0f), v3 = fmaxf(acc16[ct][t][3] * sc.w + t4.w, 0.0f);
                        *reinterpret_cast<uint2*>(Ab + off16(ct, t)) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                    }
                }
            } else if (ph > 0 && !S0) {
                load_x0_half(ph == 3 ? 0 : 1);      // ph 1: high half; ph 3 / 4: the raw stem output again, low / high
                __syncthreads();
            }
            if (ph < 2 && !S0) { preact_half(ph); }
            __syncthreads();
            if (ph == 0 || ph == 2) zero_acc16();
            conv_taps(ph < 3 ? 9 * ph : 24 + ph, ph < 3 ? 0 : 4, ph < 3 ? 9 : 1);
        }
        __syncthreads();                            // every wave is done with the image
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) {          // x = bf16(acc + (b2 + bp)) -> x image; the next block's operand relu(x s1 + t1) -> operand image
            const int c0 = wn * TN * 32 + ct * 16 + 4 * lq;
            const float4 b = *reinterpret_cast<const float4*>(&Ps[896 + c0]);
            const float4 sc = *reinterpret_cast<const float4*>(&Ps[c0]);
            const float4 t4 = *reinterpret_cast<const float4*>(&Ps[128 + c0]);
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const int o = off16(ct, t);
                const uint2 xn = make_uint2(pack_bf16(acc16[ct][t][0] + b.x, acc16[ct][t][1] + b.y), pack_bf16(acc16[ct][t][2] + b.z, acc16[ct][t][3] + b.w));
                *reinterpret_cast<uint2*>(Xb + o) = xn;
                const float a0 = fmaxf(__uint_as_float(xn.x << 16) * sc.x + t4.x, 0.0f), a1 = fmaxf(__uint_as_float(xn.x & 0xFFFF0000u) * sc.y + t4.y, 0.0f);
                const float a2 = fmaxf(__uint_as_float(xn.y << 16) * sc.z + t4.z, 0.0f), a3 = fmaxf(__uint_as_float(xn.y & 0xFFFF0000u) * sc.w + t4.w, 0.0f);
                *reinterpret_cast<uint2*>(Ab + o) = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
            }
        }
        __syncthreads();
    }
    // The block loop, once per set of sit-out masks: a wave row with a row permutation runs the instance whose tap loop is straight-line code
    // for ITS edge tiles (m0c / m1c: TrunkArgs::perm, conv_taps_static); 0 / 0 = the plain loop.  The instances are whole-wave alternatives
    // (every wave of the workgroup passes the same barriers in the same order), so no branch sits inside a loop: two tap-loop variants in
    // an if / else INSIDE the block loop cost 175 spilled registers.
    auto block_loop = [&](auto m0c, auto m1c, auto m2c) {
    constexpr bool STATIC_TAPS = decltype(m0c)::value != 0 || decltype(m1c)::value != 0 || decltype(m2c)::value != 0;
#pragma unroll 1
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const float* P = Ps + (blk & 1) * TR_PRM;
        const float* Pn = Ps + ((blk + 1) & 1) * TR_PRM;
        const bool more = blk + 1 < a.nblocks;
        float4 pnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (more && tid < TR_PRM / 4) pnext = prm4[(blk + 1) * (TR_PRM / 4) + tid];
#pragma unroll
        for (int conv = 0; conv < 2; ++conv) {
            zero_acc16();
            if constexpr (STATIC_TAPS) conv_taps_static(m0c, m1c, m2c, SL0 + blk * 18 + conv * 9);
            else conv_taps(SL0 + blk * 18 + conv * 9, 0, 9);
            if (blk < 10) TR_STAMP(3 + 6 * blk + 3 * conv);
            if (conv == 0) {
                if (more && tid < TR_PRM / 4) Ps4[((blk + 1) & 1) * (TR_PRM / 4) + tid] = pnext;
                __syncthreads();                    // every wave is done with the operand image
                if (blk < 10) TR_STAMP(4 + 6 * blk);
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct) {
                    const int c0 = wn * TN * 32 + ct * 16 + 4 * lq;
                    const float4 s = *reinterpret_cast<const float4*>(&P[2 * 128 + c0]);
                    const float4 t4 = *reinterpret_cast<const float4*>(&P[3 * 128 + c0]);
#pragma unroll
                    for (int t = 0; t < NC; ++t) {
                        const float v0 = fmaxf(acc16[ct][t][0] * s.x + t4.x, 0.0f), v1 = fmaxf(acc16[ct][t][1] * s.y + t4.y, 0.0f);
                        const float v2 = fmaxf(acc16[ct][t][2] * s.z + t4.z, 0.0f), v3 = fmaxf(acc16[ct][t][3] * s.w + t4.w, 0.0f);
                        *reinterpret_cast<uint2*>(Ab + off16(ct, t)) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                    }
                }
                __syncthreads();
                if (blk < 10) TR_STAMP(5 + 6 * blk);
            }
        }
        __syncthreads();                            // every wave is done with the h image
        if (blk < 10) TR_STAMP(7 + 6 * blk);
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) {
            const int c0 = wn * TN * 32 + ct * 16 + 4 * lq;
            const float4 b = *reinterpret_cast<const float4*>(&P[4 * 128 + c0]);
            const float4 s = *reinterpret_cast<const float4*>(&Pn[c0]);
            const float4 t4 = *reinterpret_cast<const float4*>(&Pn[128 + c0]);
#pragma unroll
            for (int t = 0; t < NC; ++t) {
                const int o = off16(ct, t);
                const uint2 xo = *reinterpret_cast<const uint2*>(Xb + o);
                const float v0 = (acc16[ct][t][0] + b.x) + __uint_as_float(xo.x << 16);
                const float v1 = (acc16[ct][t][1] + b.y) + __uint_as_float(xo.x & 0xFFFF0000u);
                const float v2 = (acc16[ct][t][2] + b.z) + __uint_as_float(xo.y << 16);
                const float v3 = (acc16[ct][t][3] + b.w) + __uint_as_float(xo.y & 0xFFFF0000u);
                const uint2 xn = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                *reinterpret_cast<uint2*>(Xb + o) = xn;
                if (more) {
                    const float a0 = fmaxf(__uint_as_float(xn.x << 16) * s.x + t4.x, 0.0f), a1 = fmaxf(__uint_as_float(xn.x & 0xFFFF0000u) * s.y + t4.y, 0.0f);
                    const float a2 = fmaxf(__uint_as_float(xn.y << 16) * s.z + t4.z, 0.0f), a3 = fmaxf(__uint_as_float(xn.y & 0xFFFF0000u) * s.w + t4.w, 0.0f);
                    *reinterpret_cast<uint2*>(Ab + o) = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
                }
            }
        }
        __syncthreads();
        if (blk < 10) TR_STAMP(8 + 6 * blk);
    }
    };
    typedef std::integral_constant<unsigned, 0u> no_mask;
    if constexpr (SKIPSET == 1) {                   // two wave rows: (y = 0 edge, x = 0 edge) | (y = H - 1 edge, x = W - 1 edge)
        if (wm == 0) block_loop(std::integral_constant<unsigned, 0x007u>{}, std::integral_constant<unsigned, 0x049u>{}, no_mask{});
        else block_loop(std::integral_constant<unsigned, 0x1C0u>{}, std::integral_constant<unsigned, 0x124u>{}, no_mask{});
    } else if constexpr (SKIPSET == 2) {              // one wave row: y = 0 edge, y = H - 1 edge, x = 0 edge (the padding rows fill it up)
        block_loop(std::integral_constant<unsigned, 0x007u>{}, std::integral_constant<unsigned, 0x1C0u>{}, std::integral_constant<unsigned, 0x049u>{});
    } else {                                        // (a Gomoku board in 256 rows has one tile per edge = four wave-row roles: four instances
                                                    // of this loop in one kernel spill 139 - 159 registers, measured 1458 vs 1840 positions/s: not built)
        block_loop(no_mask{}, no_mask{}, no_mask{});
    }
    } else {
#pragma unroll 1
    for (int blk = 0; blk < a.nblocks; ++blk) {
        const float* P = Ps + (blk & 1) * TR_PRM;
        const float* Pn = Ps + ((blk + 1) & 1) * TR_PRM;
        const bool more = blk + 1 < a.nblocks;
        float4 pnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (more && tid < TR_PRM / 4) pnext = prm4[(blk + 1) * (TR_PRM / 4) + tid];      // lands behind conv1
#pragma unroll                                      // two copies of the tap loop: keeps the h-write address math out of any loop
        for (int conv = 0; conv < 2; ++conv) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.0f;
            // Cell-operand fragments: NB register buffers (four; two at TM = 4, where a k-step is 256 MFMA cycles), k-step ks in
            // afr[ks % NB], fetched NB - 1 k-steps ahead and across the
            // tap boundary (the last three k-steps of a tap fetch the next tap's first three), so no MFMA waits on an LDS
            // round trip.  __builtin_amdgcn_sched_barrier pins each k-step's reads, MFMAs and weight loads where they are
            // written: left alone, the scheduler sinks every ds_read to just before its MFMA and the ring's loads to the end of
            // the tap (both latencies fully exposed: a lone wave ran the taps at 50 % of the MFMA rate).
            int pb[TM], pbn[TM];                    // byte address of k-slot 0 of this lane's operand row: row * 256 | ((lhi ^ row & 15) << 4)
            auto tap_rows = [&](int tap, int (&o)[TM]) {
                const int ty = tap / 3, off = (ty - 1) * a.W + (tap - ty * 3 - 1);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const bool ok = (vmask[tm] >> tap) & 1u;
                    const int ar = ok ? lrow[tm] + off : ZROW + ((lrow[tm] + off) & 15);
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
                const int sl = blk * 18 + conv * 9 + tap;
                const int nsl = sl < last_slice ? sl + 1 : sl;
                tap_rows(tap < 8 ? tap + 1 : 8, pbn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        afr[(ks + PD) % NB][tm] = *reinterpret_cast<const uint4*>(Ab + ((ks + PD < KS ? pb[tm] : pbn[tm]) ^ (((ks + PD) % KS) * 32)));
                    __builtin_amdgcn_sched_barrier(0);      // reads first (see the 16x16x32 loop)
                    bf16x8 bf[TN];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<bf16x8*>(&bfr[ks % RING][tn]);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)     // D[channel][cell]: weights are the A operand
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[tn], *reinterpret_cast<bf16x8*>(&afr[ks % NB][tm]), acc[tm][tn], 0, 0, 0);
                    if (ks + RING < KS) {
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(sl, ks + RING, tn);
                    } else {                        // next slice (the next conv's or the next block's); the very last one re-reads itself
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) bfr[ks % RING][tn] = ldb(nsl, ks + RING - KS, tn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) pb[tm] = pbn[tm];
            }
            if (blk < 10) TR_STAMP(3 + 6 * blk + 3 * conv);
            if (conv == 0) {
                if (more && tid < TR_PRM / 4) Ps4[((blk + 1) & 1) * (TR_PRM / 4) + tid] = pnext;
                __syncthreads();                    // every wave is done with the operand image
                if (blk < 10) TR_STAMP(4 + 6 * blk);
                // ---- h = relu(acc * s2 + t2) as bf16 over the operand image
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c0 = ((wn * TN + tn) * 4 + j) * 8 + 4 * lhi;
                            const float4 s = *reinterpret_cast<const float4*>(&P[2 * 128 + c0]);
                            const float4 t = *reinterpret_cast<const float4*>(&P[3 * 128 + c0]);
                            const float v0 = fmaxf(acc[tm][tn][4 * j + 0] * s.x + t.x, 0.0f), v1 = fmaxf(acc[tm][tn][4 * j + 1] * s.y + t.y, 0.0f);
                            const float v2 = fmaxf(acc[tm][tn][4 * j + 2] * s.z + t.z, 0.0f), v3 = fmaxf(acc[tm][tn][4 * j + 3] * s.w + t.w, 0.0f);
                            *reinterpret_cast<uint2*>(Ab + img_off(tm, tn, j)) = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                        }
                __syncthreads();
                if (blk < 10) TR_STAMP(5 + 6 * blk);
            }
        }
        // RESG: the residual groups of the first cell tile are on their way while the barrier waits
        const bf16_t* rsrc = blk == 0 ? a.xin : a.xout;
        uint2 rres[TN * 4];                         // one cell tile's groups at a time (a second buffer spills at TM = 4)
        auto res_load = [&](int tm, uint2 (&o)[TN * 4]) {
            const long gr = m0 + lrow[tm];
            const bool rok = lrow[tm] < tile_rows && gr < a.M;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = ((wn * TN + tn) * 4 + j) * 8 + 4 * lhi;
                    o[tn * 4 + j] = rok ? *reinterpret_cast<const uint2*>(rsrc + (size_t)gr * BN + c0) : make_uint2(0u, 0u);
                }
        };
        if (RESG) res_load(0, rres);
        __syncthreads();                            // every wave is done with the h image
        if (blk < 10) TR_STAMP(7 + 6 * blk);
        // ---- x = bf16(acc + bias + x) in place; the next block's operand relu(x * s1' + t1') over the operand image
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const long grow = m0 + lrow[tm];
            const bool rowok = lrow[tm] < tile_rows && grow < a.M;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = ((wn * TN + tn) * 4 + j) * 8 + 4 * lhi, o = img_off(tm, tn, j);
                    const float4 b = *reinterpret_cast<const float4*>(&P[4 * 128 + c0]);
                    const uint2 xo = RESG ? rres[tn * 4 + j] : *reinterpret_cast<const uint2*>(Xb + o);
                    const float v0 = (acc[tm][tn][4 * j + 0] + b.x) + __uint_as_float(xo.x << 16);
                    const float v1 = (acc[tm][tn][4 * j + 1] + b.y) + __uint_as_float(xo.x & 0xFFFF0000u);
                    const float v2 = (acc[tm][tn][4 * j + 2] + b.z) + __uint_as_float(xo.y << 16);
                    const float v3 = (acc[tm][tn][4 * j + 3] + b.w) + __uint_as_float(xo.y & 0xFFFF0000u);
                    const uint2 xn = make_uint2(pack_bf16(v0, v1), pack_bf16(v2, v3));
                    if (!RESG) *reinterpret_cast<uint2*>(Xb + o) = xn;
                    else if (rowok) *reinterpret_cast<uint2*>(a.xout + (size_t)grow * BN + c0) = xn;
                    if (more) {
                        const float4 s = *reinterpret_cast<const float4*>(&Pn[c0]);
                        const float4 t = *reinterpret_cast<const float4*>(&Pn[128 + c0]);
                        const float a0 = fmaxf(__uint_as_float(xn.x << 16) * s.x + t.x, 0.0f), a1 = fmaxf(__uint_as_float(xn.x & 0xFFFF0000u) * s.y + t.y, 0.0f);
                        const float a2 = fmaxf(__uint_as_float(xn.y << 16) * s.z + t.z, 0.0f), a3 = fmaxf(__uint_as_float(xn.y & 0xFFFF0000u) * s.w + t.w, 0.0f);
                        *reinterpret_cast<uint2*>(Ab + o) = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
                    }
                }
            if (RESG && tm + 1 < TM) res_load(tm + 1, rres);
        }
        __syncthreads();
        if (blk < 10) TR_STAMP(8 + 6 * blk);
    }
    }

    if (HEADS) {
        // ---- first convolution of both heads (Connect4/Build_Model.py:41,62: two Conv3x3 128 -> 8 as one 128 -> 16 GEMM padded to a
        // 32-channel MFMA tile) + each head's flat BN + ReLU, as k_conv_heads computes them, from the raw x image: wave w takes cells
        // [32 w, 32 w + 32) x the 32 channels.  72 MFMAs per wave with one MFMA per k-step, so the weight ring is three taps deep.
        constexpr int HBSL = 32 * SLOTS, HRING = 24;
        // the lane geometry of this phase is recomputed from an opaque copy of the thread index: taken from the values set up before the
        // block loop it stays live across the loop, which the static tap loops cannot afford (11 spilled registers, 40 MB of scratch traffic
        // per launch)
        int tidh = tid;
        asm volatile("" : "+v"(tidh));
        const int l31h = tidh & 31, lhih = (tidh >> 5) & 1, waveh = tidh >> 6;
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.hw, 0, 9 * HBSL * 16, 0x00020000);
        const int hvo = (lhih * 32 + l31h) * 16;
        auto ldh = [&](int tap, int ks) -> uint4 {
            const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(hrs, hvo, (tap * HBSL + ks * 2 * 32) * 16, 0);
            return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
        };
        uint4 hfr[HRING];
#pragma unroll
        for (int g = 0; g < HRING; ++g) hfr[g] = ldh(g / KS, g % KS);
        const int hrow = waveh * 32 + l31h;         // cell tile `wave` of the workgroup
        unsigned hmask = 0;
        // (an opaque copy of the packed board offsets: decoded from the SAME value as before the block loop, the three offsets stay live across it —
        // in scalar registers the edge-tile variants do not have: they were spilled to scratch memory, 40 MB of HBM traffic per launch)
        unsigned boh = boffp & 0x00FFFFFFu;
        asm volatile("" : "+s"(boh));
        int hcell; long hgr;
        const bool hok = locate(hrow, hcell, hgr, boh);
        if (hok) {
            int y, x; cell_yx(hcell, y, x);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                hmask |= (((unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W) ? 1u : 0u) << t;
            }
        }
        // the epilogue's per-position parameters (flat BN of both heads: L2 round trips) are requested here, in front of the taps, not behind them
        float4 hbi[2], hsc[2], hsh[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { hbi[j] = hsc[j] = hsh[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
        if (hok) {
            const int f = hcell * 8 + 4 * lhih;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                hbi[j] = *reinterpret_cast<const float4*>(a.hbias + 8 * j + 4 * lhih);
                hsc[j] = *reinterpret_cast<const float4*>((j ? a.v_fs : a.p_fs) + f);
                hsh[j] = *reinterpret_cast<const float4*>((j ? a.v_ft : a.p_ft) + f);
            }
        }
        if (waveh * 32 < ROWS) {                    // wave-uniform (WN = 4, TM = 3: the fourth wave has no cell tile)
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.0f;
#pragma unroll 1
        for (int t3 = 0; t3 < 3; ++t3) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const int tap = t3 * 3 + tt, off = (t3 - 1) * a.W + (tt - 1);
                const bool ok = (hmask >> tap) & 1u;
                const int ar = hrow + off;
                // k-group 2 ks + lhi of row ar: 32x32x16 layout slot (lhi ^ ar & 15) ^ 2 ks; 16x16x32 layout slot (lhi << 3 | ar & 7) ^ ks
                const int pbh = (ok ? ROWS + TR_ZROWS + ar : ZROW + (ar & 15)) * 256 + (swz_slot<M16>(lhih, ar) << 4);
                uint4 hf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) hf[ks] = *reinterpret_cast<const uint4*>(Ab + (pbh ^ (ks * (M16 ? 16 : 32))));
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&hfr[tt * KS + ks]), *reinterpret_cast<bf16x8*>(&hf[ks]), hacc, 0, 0, 0);
                    const int ntap = tap + 3 < 9 ? tap + 3 : tap;
                    hfr[tt * KS + ks] = ldh(ntap, ks);
                }
            }
        }
        // lane: cell hrow, channels 8 j + 4 lhi + q: j = 0 policy head, j = 1 value head, j = 2, 3 padding
        const long gr = hgr;
        if (hok) {
            // (flat feature index of the board = hcell 8 + 4 lhi; the output row of board b starts at b HW 8, i.e. the element is gr 8 + 4 lhi)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 bi = hbi[j], sc = hsc[j], sh = hsh[j];
                float4 o;
                o.x = fmaxf((hacc[4 * j + 0] + bi.x) * sc.x + sh.x, 0.0f); o.y = fmaxf((hacc[4 * j + 1] + bi.y) * sc.y + sh.y, 0.0f);
                o.z = fmaxf((hacc[4 * j + 2] + bi.z) * sc.z + sh.z, 0.0f); o.w = fmaxf((hacc[4 * j + 3] + bi.w) * sc.w + sh.w, 0.0f);
                *reinterpret_cast<float4*>((j ? a.v_feat : a.p_feat) + (size_t)gr * 8 + 4 * lhih) = o;
            }
        }
        }
    } else if (!RESG) {
    // ---- the tile's rows of x -> global, whole 256-byte rows
    uint4* out4 = reinterpret_cast<uint4*>(a.xout);
    for (int i = tid; i < tile_rows * SLOTS; i += THREADS) {
        const int lr = i / SLOTS, sp = i % SLOTS;
        const long gr = m0 + lr;
        if (gr < a.M) out4[gr * SLOTS + sp] = Xs[lr * SLOTS + swz_slot<M16>(sp, lr)];
    }
    }
    TR_STAMP(63);
}

template <int TM, int WN, int RING, int OCC, bool STEM, bool HEADS, bool RESG = false, bool M16 = false, int NW = 4, bool B0 = false, int SKIPSET = 0, bool S0 = false>
__global__ __launch_bounds__(64 * NW, OCC) void k_trunk(TrunkArgs a) {
    trunk_tile<TM, WN, RING, STEM, HEADS, RESG, M16, NW, B0, SKIPSET, S0>(a, (long)blockIdx.x * a.tile_rows, a.tile_rows, SKIPSET ? a.perm : nullptr, a.boff);
}
constexpr size_t trunk_lds_bytes_s0() { return trunk_lds_bytes(256) + 4 * 2 * 2 * 64 * 16; }      // + the parked stem activations (Cf)

// Two tile shapes in one launch.  Workgroups are dispatched in index order and a CU holds two, so the batch is processed in rounds of
// 2 x CUs tiles; 4096 Connect4 boards in 3-board tiles are 1366 tiles = 2.67 rounds, the third one two thirds full and as long as
// the others.  Here the first n_big workgroups (whole rounds) take 3 boards in the 128-row shape and the rest 2 boards in a 96-row
// shape (TM = 3, WN = 4: every wave all 96 cells x 32 channels) that issues three quarters of the MFMAs: 1024 + 512 tiles = three full
// rounds, the last one cheaper.
template <int RING, int OCC, bool STEM, bool HEADS, bool M16 = false, bool SKIP = false>
__global__ __launch_bounds__(TR_THREADS, OCC) void k_trunk_mix(TrunkArgs a) {
    if ((int)blockIdx.x < a.n_big) trunk_tile<2, 2, RING, STEM, HEADS, false, M16, 4, false, SKIP ? 1 : 0>(a, (long)blockIdx.x * a.tile_rows, a.tile_rows, SKIP ? a.perm : nullptr, a.boff);
    else trunk_tile<3, 4, RING, STEM, HEADS, false, M16, 4, false, SKIP ? 2 : 0>(a, (long)a.n_big * a.tile_rows + (long)((int)blockIdx.x - a.n_big) * a.small_rows, a.small_rows, SKIP ? a.perm_small : nullptr, a.boff_small);
}

}  // namespace gaz
