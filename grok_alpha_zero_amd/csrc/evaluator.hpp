// evaluator.hpp — the batched policy/value evaluator behind the wave kernel.
//
// Replaces the reference's session.run round trip (MCTS.py:224-235 -> Client_Server.py:28-55 ->
// Client_Server.Server.start :162-217 -> onnxruntime): the wave kernel has already written every game's
// encoded leaf state as row g of `in` (int8 [n][H*W*C]); forward() fills policy f32 [n][A] and value f32 [n]
// on the same stream.  Rows are independent: a row's outputs never depend on which other rows are in the
// batch (required for bit-reproducible search).
#pragma once
#include <string>
#include "../../include/gaz_engine.h"
#include "rt.hpp"

namespace gaz {

// what a fused tree + trunk launch hands from the tree teams to the trunk workgroups (trunk.hpp TrunkArgs::ready ... test_fault_mod)
struct FuseHandoff { const unsigned* ready; unsigned epoch; unsigned* eval_done; int* fuse_fault; unsigned spin_ticks; unsigned test_fault_mod;
                     unsigned long long* stamps;         // stamps: diagnostic (GAZ_FUSED_STAMPS), [block][128] or null
                     const unsigned long long* queue; }; // completion queue (DevParams::done_queue) or null: ready[] flags per board

struct Evaluator {
    virtual ~Evaluator() {}
    virtual int load(const gaz_tensor* t, int n, hipStream_t s, std::string* err) { (void)t; (void)n; (void)s; (void)err; return 0; }
    // p0: first row of the evaluator's internal activation buffers this call may use (rows [p0, p0 + n)); two calls on
    // disjoint row ranges may be in flight on different streams (the engine pipelines two halves of the games)
    virtual void forward(hipStream_t s, const int8_t* in, float* policy, float* value, int n, bool timing, int p0 = 0) = 0;
    virtual bool supports_row_base() const { return false; }
    // group pipeline (engine.hip run_waves_pipelined): the forward pass in two parts that may run on different streams — the trunk
    // (input planes of rows [p0, p0 + n) -> head features in the evaluator's own buffers) and the heads (features -> policy / value)
    virtual bool supports_split() const { return false; }
    virtual int round_rows() const { return 0; }                  // positions one full round of the trunk kernel's tiles covers
    virtual void forward_trunk(hipStream_t s, const int8_t* in, int n, bool timing, int p0) { (void)s; (void)in; (void)n; (void)timing; (void)p0; }
    virtual void forward_heads(hipStream_t s, float* policy, float* value, int n, int p0) { (void)s; (void)policy; (void)value; (void)n; (void)p0; }
    // fused tree + trunk launch (resnet.hip k_wave_trunk): the trunk part as a launch PLAN (kernel arguments + grid) instead of a launch; the
    // planes of board b are valid once ready[b] == epoch (FuseHandoff, below).  Null = this evaluator / configuration cannot be fused.
    virtual const void* trunk_plan(const int8_t* in, int n, int p0, const struct FuseHandoff& h) { (void)in; (void)n; (void)p0; (void)h; return nullptr; }
    virtual bool plan_uses_queue(const void* plan) const { (void)plan; return false; }      // did trunk_plan take FuseHandoff::queue on?
    // this evaluator serves ONE of several game groups whose launches share the chip (engine.hip GroupEngine): tile rounds of a single launch
    // need not come out even — another group's tiles fill the slots — so the plan may take the more efficient tile shape throughout
    virtual void set_shared_chip(bool on) { (void)on; }
    virtual bool ready() const { return true; }
    virtual void timing_reset() {}
    virtual void timing_get(double* ms, int64_t* launches) { *ms = 0; *launches = 0; }
    // the kernel bench.py prices against the roofline: name and algorithmic FLOPs of one launch at batch n
    virtual const char* dominant_kernel(int n, double* flops) { (void)n; *flops = 0; return ""; }
    // diagnostics (numerics tests): device pointers of the flat head features the last forward() wrote, f32 [n][row_floats] each
    virtual bool head_features(const float** p_feat, const float** v_feat, int* p_row_floats, int* v_row_floats) { (void)p_feat; (void)v_feat; (void)p_row_floats; (void)v_row_floats; return false; }
};

// synthetic evaluator for parity tests: outputs are exact float32 functions of a hash of the input row
//   policy[a] = ((h_a >> 8) + 1) * 2^-24,  value = (h_v >> 8) * 2^-23 - 1,  h = fmix32(FNV-1a(row, salt ^ x))
GAZ_DEV uint32_t hash_row(const int8_t* s, int n, uint32_t seed) {
    uint32_t h = 2166136261u ^ seed;
    for (int i = 0; i < n; ++i) { h ^= (uint8_t)s[i]; h *= 16777619u; }
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

template <int UNUSED> GAZ_KERNEL_WIDE k_hash_eval(const int8_t* in, float* policy, float* value, int n, int row_bytes, int A, uint32_t salt) {
#ifdef GAZ_HOST_EMU
    const int first = block_id(), step = 1 << 30;
#else
    const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
#endif
    for (int idx = first; idx < n * (A + 1); idx += step) {
        const int g = idx / (A + 1), a = idx % (A + 1);
        const int8_t* row = in + (size_t)g * row_bytes;
        if (a < A) {
            uint32_t h = hash_row(row, row_bytes, salt ^ ((uint32_t)(a + 1) * 0x9E3779B1u));
            policy[(size_t)g * A + a] = (float)((h >> 8) + 1u) * (1.0f / 16777216.0f);
        } else {
            uint32_t h = hash_row(row, row_bytes, salt ^ 0x51ED270Bu);
            value[g] = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        }
    }
}

struct HashEvaluator : Evaluator {
    int row_bytes, A; uint32_t salt;
    HashEvaluator(int rb, int a, uint32_t s) : row_bytes(rb), A(a), salt(s) {}
    void forward(hipStream_t s, const int8_t* in, float* policy, float* value, int n, bool, int = 0) override {
        const int total = n * (A + 1);
#ifdef GAZ_HOST_EMU
        GAZ_LAUNCH(k_hash_eval<0>, total, 1, s, in, policy, value, n, row_bytes, A, salt);
#else
        GAZ_LAUNCH(k_hash_eval<0>, (total + 255) / 256, 256, s, in, policy, value, n, row_bytes, A, salt);
#endif
    }
};

Evaluator* make_resnet_evaluator(const gaz_engine_config& cfg, int H, int W, int C, int A, std::string* err);
// resnet.hip k_wave_trunk: ONE launch = the PUCT tree step of Connect4 games [g0, g1) (16-lane teams) + the trunk kernel of their leaf rows.
// dev_params: DevParams<TeamGame<GAME_C4>> by value; plan: what Evaluator::trunk_plan returned.  false = not launched.
bool launch_wave_trunk_c4(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan);
// the same for the Gumbel search (MCTS_Gumbel.py:562-679) of Connect4: one wavefront per game, four games per tree block.  dev_params: DevParams<Game<GAME_C4>>
bool launch_wave_trunk_c4_gumbel(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan);
// the same for Gomoku's PUCT search: one game per wavefront, eight per tree block and round.  dev_params: DevParams<Game<GAME_GMK>>
bool launch_wave_trunk_gmk(hipStream_t s, const void* dev_params, int g0, int g1, const void* plan);

inline Evaluator* make_evaluator(const gaz_engine_config& cfg, int H, int W, int C, int A, std::string* err) {
    if (cfg.evaluator == GAZ_EVAL_HASH) return new HashEvaluator(H * W * C, A, cfg.hash_salt);
    if (cfg.evaluator == GAZ_EVAL_RESNET) return make_resnet_evaluator(cfg, H, W, C, A, err);
    if (cfg.evaluator == GAZ_EVAL_EXTERNAL) return nullptr;
    *err = "unknown evaluator id";
    return nullptr;
}

}  // namespace gaz
