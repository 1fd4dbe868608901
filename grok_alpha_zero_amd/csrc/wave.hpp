// wave.hpp — wavefront primitives for the tree kernels (gfx950, wave64).
//
// A TEAM of lanes owns one game (the whole wavefront, or a 16-lane row: four games per wavefront, see "Teams" below): control flow
// is uniform within the team, lanes fan out over the children of a node (PUCT scoring, team argmax), over candidate actions
// (terminal probe), over gamma variates (Dirichlet noise) and over path levels (backup).  All cross-lane traffic is __shfl / __ballot (DPP /
// ds_bpermute, no LDS round trip) plus a small per-wave LDS scratch.
//
// GAZ_HOST_EMU builds the very same device functions for the CPU with a wave of ONE lane (every
// `for (i = lane; i < n; i += WAVE)` loop degenerates to a serial loop).  That build exists only for the
// CPU test-suite (tests/emu), so the device logic is exercised by `pytest -m "not gpu"`; the product
// library is always the hipcc build and never falls back to it.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef GAZ_HOST_EMU
#include <math.h>
#define GAZ_DEV inline
#define GAZ_HD inline
#define GAZ_KERNEL inline void
#define GAZ_KERNEL_TEAMS inline void
#define GAZ_KERNEL_WIDE inline void
#define GAZ_SHARED static thread_local
struct char4 { signed char x, y, z, w; };
struct uint4 { unsigned int x, y, z, w; };
namespace gaz {
constexpr int WAVE = 1;
inline thread_local int emu_block_id = 0;
inline int lane_id() { return 0; }
inline int block_id() { return emu_block_id; }
inline uint64_t ballot(bool p) { return p ? 1ull : 0ull; }
template <class T> inline T shfl(T v, int) { return v; }
template <class T> inline T shfl_xor(T v, int) { return v; }
inline void wave_sync() {}
inline double dsqrt(double x) { return __builtin_sqrt(x); }
inline int popcll(uint64_t x) { return __builtin_popcountll(x); }
inline int ffsll0(uint64_t x) { return __builtin_ctzll(x); }   // index of lowest set bit
inline int flsll0(uint64_t x) { return 63 - __builtin_clzll(x); }
template <class T> inline T atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomic_exch(T* p, T v) { T o = *p; *p = v; return o; }
}  // namespace gaz
#include <time.h>
inline long long wall_clock64() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 100000000ll + t.tv_nsec / 10; }   // 100 MHz, as the device's
#else
#include <hip/hip_runtime.h>
#define GAZ_DEV __device__ __forceinline__
#define GAZ_HD __host__ __device__ __forceinline__
// one wavefront per workgroup.  GAZ_TREE_WPE pins the waves per SIMD the register allocator aims for (4 -> 128 VGPRs with
// spills to scratch, 3 -> 168, 2 -> no spills); see DESIGN.md for the measured choice.
#ifndef GAZ_TREE_WPE
#define GAZ_TREE_WPE 3                             // measured: 0.074 / 0.080 / 0.083 ms per PUCT wave at 3 / 2 / 4
#endif
#define GAZ_KERNEL __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GAZ_TREE_WPE, GAZ_TREE_WPE))) void
// four games per wavefront (16-lane teams): one wave per SIMD covers 4096 games, so the register budget is generous
#define GAZ_KERNEL_TEAMS __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void
#define GAZ_KERNEL_WIDE __global__ void               // kernels launched with more than one wavefront per workgroup
#define GAZ_SHARED __shared__
namespace gaz {
constexpr int WAVE = 64;
GAZ_DEV int lane_id() { return threadIdx.x & 63; }
GAZ_DEV int block_id() { return blockIdx.x; }
GAZ_DEV uint64_t ballot(bool p) { return __ballot(p); }
template <class T> GAZ_DEV T shfl(T v, int src) { return __shfl(v, src, 64); }
template <class T> GAZ_DEV T shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
// One wave per workgroup: lanes run in lockstep and the wave's LDS / vector-memory operations are performed in program
// order, so cross-lane hand-offs through LDS or global memory need only a WAVEFRONT-scope fence (no instruction on
// gfx950: no s_waitcnt vmcnt(0) drain, no s_barrier) plus a scheduling barrier for the compiler.  A workgroup-scope
// __syncthreads() here made every hand-off wait for all outstanding stores (a full HBM round trip each).
GAZ_DEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
GAZ_DEV double dsqrt(double x) { return __dsqrt_rn(x); }
GAZ_DEV int popcll(uint64_t x) { return __popcll(x); }
GAZ_DEV int ffsll0(uint64_t x) { return __ffsll((unsigned long long)x) - 1; }
GAZ_DEV int flsll0(uint64_t x) { return 63 - __clzll((long long)x); }
template <class T> GAZ_DEV T atomic_add(T* p, T v) { return atomicAdd(p, v); }
template <class T> GAZ_DEV T atomic_max(T* p, T v) { return atomicMax(p, v); }
template <class T> GAZ_DEV T atomic_exch(T* p, T v) { return atomicExch(p, v); }
}  // namespace gaz
#endif

namespace gaz {
// ---------------------------------------------------------------------------------------------------------------------------------
// Teams: a game is owned by a TEAM of G::TEAM consecutive lanes — the whole wavefront (64) for Gomoku's 225-wide nodes, a 16-lane
// DPP row for Connect4 / TicTacToe (<= 9 children, <= 42 cells), so that one wavefront steps FOUR games.  The tree kernel is bound
// by vector-instruction issue (a wave64 instruction costs four cycles whatever the number of live lanes, and a Connect4 node keeps
// at most 7 of 64 busy): four games per instruction stream cut the instructions issued per game.  Control flow is uniform within a
// team and may diverge between the teams of a wave (the hardware masks lanes); all cross-lane traffic of a game stays inside its row:
// ballots are masked to the row, butterflies use xor < TEAM, "uniform" values are per-team VGPRs (read from one address by all lanes
// of the team, or brought to every lane by a full butterfly), never wave scalars.
#ifdef GAZ_HOST_EMU
#define GAZ_TEAM(n) 1
#else
#define GAZ_TEAM(n) (n)
#endif
template <class G> GAZ_DEV int tlane() { return lane_id() & (G::TEAM - 1); }            // lane within the team
template <class G> GAZ_DEV int team_in_wave() { return lane_id() / G::TEAM; }            // which of the wave's teams
template <class G> GAZ_DEV uint64_t tballot(bool p) {                                    // ballot over the team's lanes, bit i = team lane i
    const uint64_t m = ballot(p);
    if (G::TEAM >= 64) return m;
    return (m >> (team_in_wave<G>() * G::TEAM)) & ((1ull << (G::TEAM & 63)) - 1ull);
}
template <class G, class T> GAZ_DEV T tshfl(T v, int src) { return shfl(v, team_in_wave<G>() * G::TEAM + src); }   // value of team lane `src`
// a value every lane of the team already holds (read from one address) — kept in an SGPR when the team is the whole wave
template <class G, class T> GAZ_DEV T tuni(T v) {
#ifdef GAZ_HOST_EMU
    return v;
#else
    if (G::TEAM >= 64) return (T)__builtin_amdgcn_readfirstlane((int)v);
    return v;
#endif
}
// team argmax over (score, index): largest score wins, ties -> LOWEST index; lanes holding no candidate pass idx = INT32_MAX.
// Full butterfly: EVERY lane of the team ends up with the result.
template <class G> GAZ_DEV void team_argmax(double& score, int& idx) {
#ifndef GAZ_HOST_EMU
#pragma unroll
    for (int m = G::TEAM / 2; m >= 1; m >>= 1) {
        double os = shfl_xor(score, m);
        int oi = shfl_xor(idx, m);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || os > score || (os == score && oi < idx));
        if (take) { score = os; idx = oi; }
    }
#endif
}
template <class G> GAZ_DEV void team_argmax_u32(uint32_t& v, int& idx) {
#ifndef GAZ_HOST_EMU
#pragma unroll
    for (int m = G::TEAM / 2; m >= 1; m >>= 1) {
        uint32_t ov = shfl_xor(v, m);
        int oi = shfl_xor(idx, m);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || ov > v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
#endif
}
template <class G> GAZ_DEV uint64_t team_sum_u64(uint64_t v) {                            // wrap-around add: order-independent
#ifndef GAZ_HOST_EMU
#pragma unroll
    for (int m = G::TEAM / 2; m >= 1; m >>= 1) v += shfl_xor(v, m);
#endif
    return v;
}

// wave argmax over (score, index): largest score wins, ties -> LOWEST index (np.argmax semantics,
// MCTS.py:191).  Lanes holding no candidate pass idx = INT32_MAX.
// `span`: candidates live in lanes [0, span) only (wave-uniform; span = 64 reduces the whole wave).  Lane 0 ends up with the
// result either way; the butterfly below span leaves it in every lane < span, callers broadcast with uni().
GAZ_DEV void wave_argmax(double& score, int& idx, int span = 64) {
#ifndef GAZ_HOST_EMU
    int m0 = 32;
    if (span <= 8) m0 = 4; else if (span <= 16) m0 = 8; else if (span <= 32) m0 = 16;
    for (int m = m0; m >= 1; m >>= 1) {
        double os = shfl_xor(score, m);
        int oi = shfl_xor(idx, m);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || os > score || (os == score && oi < idx));
        if (take) { score = os; idx = oi; }
    }
#endif
}
// wave max of uint32 with lowest index on ties (np.argmax over child_visits, MCTS.py:604)
GAZ_DEV void wave_argmax_u32(uint32_t& v, int& idx) {
#ifndef GAZ_HOST_EMU
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        uint32_t ov = shfl_xor(v, m);
        int oi = shfl_xor(idx, m);
        bool take = (oi != 0x7fffffff) && (idx == 0x7fffffff || ov > v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
#endif
}
// wave sum of uint64 (wrap-around add: order-independent)
GAZ_DEV uint64_t wave_sum_u64(uint64_t v) {
#ifndef GAZ_HOST_EMU
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
#endif
    return v;
}
}  // namespace gaz
