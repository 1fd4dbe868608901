// tree.hpp — HBM layout of the search state.
//
// Reference objects replaced (MCTS.py:20-72): Node / Root python objects with per-node numpy arrays
// child_visits u32[A], child_values f32[A], child_prob_priors f32[A], a children list and a deque of
// un-popped legal actions in descending-prior order.
//
// Here: one flat arena per (game, tree); a node is a fixed-size record holding a 32-byte header, then
// structure-of-arrays blocks over its children — P f32[APAD] | N u32[APAD] | W f32[APAD] |
// child i32[APAD] | action u8[APAD] — then the node's int8 board.  Children are stored in descending-prior
// order, so "pop the next legal action" (MCTS.py:437) is slot n_children and the reference's invariant
// best_index <= len(children) (MCTS.py:217) holds by construction.  A select step touches one record
// (Connect4: 224 B = two 128-B lines; lanes read N/W/P of consecutive children = coalesced 32-B rows).
// Terminal children (MCTS.py:403-426) need no record: they are encoded in the parent's child[] word.
#pragma once
#include "games.hpp"

namespace gaz {

enum : int32_t { CHILD_NONE = -1, CHILD_LEAF_DRAW = -2, CHILD_LEAF_WIN = -3 };
enum : uint8_t { NF_TERMINAL_PARENT = 1 };

struct NodeHdr {          // 32 bytes
    int32_t parent;       // node index in this tree's arena, -1 for a root created by create_expand_root
    int16_t slot;         // child_id (MCTS.py:32)
    uint8_t n_actions;    // len(child_visits)
    uint8_t n_children;   // len(children): expanded so far
    int8_t player;        // current_player (who just moved into this position)
    uint8_t flags;
    uint16_t n_hist;      // len(action_history) of this position
    uint8_t hist3[3];     // last three actions of the path, newest first (Connect4 input planes)
    uint8_t action;       // action_history[-1]
    uint32_t pad[4];
};
static_assert(sizeof(NodeHdr) == 32, "NodeHdr must be 32 bytes");

template <class G> struct NodeLayout {
    // P first: a backup updates N[s] and W[s] of every node on the path, and with N | W behind the 32-byte header + P block the two
    // words of a Connect4 node share one 64-byte sector (bytes 64..127) instead of straddling two
    static constexpr int OFF_P = 32;
    static constexpr int OFF_N = OFF_P + 4 * G::APAD;
    static constexpr int OFF_W = OFF_N + 4 * G::APAD;
    static constexpr int OFF_CHILD = OFF_W + 4 * G::APAD;
    static constexpr int OFF_ACT = OFF_CHILD + 4 * G::APAD;
    static constexpr int OFF_BOARD = OFF_ACT + G::APAD;
    static constexpr int SIZE = (OFF_BOARD + G::BPAD + 31) / 32 * 32;
};

template <class G> struct NodeRef {
    uint8_t* p;
    GAZ_DEV NodeHdr* hdr() const { return reinterpret_cast<NodeHdr*>(p); }
    GAZ_DEV uint32_t* N() const { return reinterpret_cast<uint32_t*>(p + NodeLayout<G>::OFF_N); }
    GAZ_DEV float* W() const { return reinterpret_cast<float*>(p + NodeLayout<G>::OFF_W); }
    GAZ_DEV float* P() const { return reinterpret_cast<float*>(p + NodeLayout<G>::OFF_P); }
    GAZ_DEV int32_t* child() const { return reinterpret_cast<int32_t*>(p + NodeLayout<G>::OFF_CHILD); }
    GAZ_DEV uint8_t* act() const { return p + NodeLayout<G>::OFF_ACT; }
    GAZ_DEV int8_t* board() const { return reinterpret_cast<int8_t*>(p + NodeLayout<G>::OFF_BOARD); }
};

struct TreeState {        // 32 bytes, one per (game, tree)
    int32_t root;         // node index of the current root, -1 = no tree yet
    uint32_t n_nodes;     // bump allocator
    uint64_t root_visits; // Root.visits (MCTS.py:70, carried over on re-root MCTS.py:654)
    uint32_t event;       // RNG event counter of this tree's stream
    uint32_t half;        // which half of the (double-buffered) arena holds the tree (re-root compaction)
    uint32_t pad[2];
};

// per-game phases of the self-play state machine (Self_Play.play, Self_Play.py:71-157)
enum : int32_t {
    PH_NEW_GAME = 0, PH_ROOT = 1, PH_MOVE_BEGIN = 2, PH_SIMS = 3, PH_MOVE_END = 4, PH_WAIT_HOST = 5,
    PH_APPLY = 6, PH_RING_WAIT = 7, PH_HALT = 8,
    PH_IDLE = 9            // sync + single-tree hosts (mcts.py): position set / move applied, waiting for run() or the next move
};
enum : int32_t { PEND_NONE = 0, PEND_ROOT = 1, PEND_EXPAND = 2 };

template <class G> struct GameState {
    int8_t board[G::BPAD];
    uint8_t hist[G::TPAD];
    int32_t n_hist;
    int32_t next_player;
    uint32_t game_seq;
    int32_t phase;
    int32_t roots_todo;        // bitmask of trees that need create_expand_root
    int32_t runner;            // tree running the current move
    int32_t sims_done, iter_limit, fully_visited;
    int32_t tau_on[2];         // tau = 1.0 (1) or 0 (0) per tree (Self_Play.py:86-95)
    int32_t pend_kind, pend_tree, pend_parent, pend_slot, pend_node, pend_depth;
    int32_t chosen;            // action sampled at MOVE_END
    int32_t host_move;         // sync mode: host override, -1 = use chosen
    int32_t winner;
    uint32_t move_evals;       // evaluator calls during the current run()
    uint64_t n_evals, n_sims, n_plies;  // lifetime counters of this slot (n_plies = positions played)
    uint64_t n_hits;           // evaluations answered by the on-device evaluation cache (included in n_evals)
    uint64_t move_t0;          // wall clock at MOVE_BEGIN (DevParams::move_time_ticks)
    uint32_t slot_id;          // the game's GLOBAL slot (slot_offset + the physical slot it started in): RNG streams and the record
    uint32_t pad_;             // header carry this, not the physical index, so gaz_engine_repack may move a running game
};

// per-game record of the game in progress (and of finished games in the ring); see engine.hip for the
// host-side view.  Layout in bytes, all arrays [MAXT] major:
template <class G> struct RecLayout {
    static constexpr int T = G::MAXT;
    static constexpr int OFF_HDR = 0;                       // int32 T, winner, slot, seq
    static constexpr int OFF_ACT = 16;                      // u8 [TPAD]
    static constexpr int OFF_Q = OFF_ACT + G::TPAD;         // f32 [T]
    static constexpr int OFF_RV = OFF_Q + 4 * T;            // u32 [T] root.visits
    static constexpr int OFF_EV = OFF_RV + 4 * T;           // u32 [T] evaluator calls of the move
    static constexpr int OFF_POL = OFF_EV + 4 * T;          // f32 [T][A] improved policy N/sum(N)
    static constexpr int OFF_N = OFF_POL + 4 * T * G::A;    // u32 [T][A]
    static constexpr int OFF_W = OFF_N + 4 * T * G::A;      // f32 [T][A]
    static constexpr int OFF_P = OFF_W + 4 * T * G::A;      // f32 [T][A]
    static constexpr int SIZE = (OFF_P + 4 * T * G::A + 15) / 16 * 16;
};

}  // namespace gaz
