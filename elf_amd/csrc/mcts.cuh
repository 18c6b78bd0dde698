// Device-resident MCTS for gfx950 (CDNA4): one search tree per game, one wave64 per game (select / backup)
// or per leaf (feature write, expansion).  Replaces, for the self-play hot path, the reference's CPU search
//   src_cpp/elf/ai/tree_search/tree_search_node.h   NodeT: UCT :361-397, findMove :205-231, addVirtualLoss :233-251,
//                                                   updateEdgeStats :253-278, followEdge :280-302, setEvaluation :176-203,
//                                                   enhanceExploration :132-155; SearchTreeT::treeAdvance :420-436
//   src_cpp/elf/ai/tree_search/tree_search_base.h   EdgeInfo::getScore :132-157, MCTSResultT::addActions :237-294
//   src_cpp/elf/ai/tree_search/tree_search.h        single_rollout :264-322, batch_rollouts :200-262
//   src_cpp/elfgames/go/mcts/mcts.h                 MCTSActor: pre_evaluate :185-207, get_extractor :175-183,
//                                                   post_nn_result :209-230, remove_pass_if_dangerous :232-242, pi2response :256-332
// Semantics reproduced bit-for-bit (single search thread): edge iteration order of the
// reference's std::unordered_map (edges are STORED in that order, so "first in iteration order" is "lowest
// index"), std::sort order of the priors, the float/double promotion sequence of the PUCT score, the
// sequential fp32 sums (FPU running mean, prior normalisation), virtual loss, duplicate-leaf handling.
// One documented canonicalisation: leaves of a batch are backed up in first-occurrence order (the reference
// iterates an unordered_map keyed by heap addresses, tree_search.h:216,245).
// num_threads = T > 1 (TreeSearchT's thread pool, tree_search.h:345-368): one step runs the batch_rollouts of T search
// threads back to back on the shared tree -- thread t's K descents see the virtual losses of threads < t, leaves are
// de-duplicated per thread (traj_counts is per batch_rollouts call), evaluation and backup follow in thread order.  That is
// one of the interleavings the reference's racing threads can produce; T x num_rollouts_per_thread rollouts per move.
//
// HBM layout (sized for 288 GB; round 5: 5.9 KB instead of 12.5 KB per node; round 6: ONE pool per context, shared by its games --
// the reference allocates nodes from the heap (tree_search_node.h:420-467) and a game whose kept subtree is large simply takes more; G fixed
// worst-case pools made every game pay for the worst one).  Per context TWO pools of fixed-size node records:
//   small  [64 B header][compact board 2624 B][NE x f32 prior][NE x u16 coord][NE x u16 orig][16 x 16 B touched-edge stats]   5888 B at 19x19
//   big    the same with NE = 368 touched-edge entries                                                                        11520 B
// Per edge a node keeps {prior, coord, orig} (8 B) for every legal move, and {reward, visits, virtual loss, child id} (16 B) only for
// the edges that have been FOLLOWED at least once ("touched").  Every node but the root hangs on exactly one touched edge, so a tree of n
// nodes has n - 1 touched edges in all: almost every node has none or a few.  A node is born in the small pool (room for 16 touched
// edges); when its 17th edge is followed it MOVES to the big pool (k_mcts_select, promote: copy, patch the parent's child id and the
// 16 children's parent ids).  Promotion fires on the (TCS+1)-th followed edge, so a big node has >= TCS + 1 = 17 children that are nodes
// themselves: Cb = Cs / TCS + 1 big records can never run out before the small pool does.  Node id < Cs: small record id; else big
// record id - Cs.  Node ids are CONTEXT-wide: a free stack per class with an atomic top (a wave pops the ids of a whole step with one
// atomic at the head of k_mcts_select into its game's stash -- off the descents' chain -- and the tree sweep of a move boundary
// pushes dead ids back); per id a parent (-2 free, -1 root) and the owning game.
// The board of a node is the slot without its Bloom words (CBoard, go_board.cuh): the filter a forward needs is rebuilt in LDS from the
// game board's own Bloom words (one copy per launch) plus the hashes of the positions on the descent's path (the node header carries
// its position's hash, so the descent collects them on its way down).
// The edge arrays are kept in SCORING order: first the edges that have been followed at least once, sorted by their index in the
// reference's unordered_map iteration order (`orig`), then the never-followed ones by descending prior.  A never-followed edge
// has N = 0, vl = 0, so its PUCT score is a monotone function of its prior: select scores the followed edges and the head of the
// prior-sorted run -- ONE coalesced round of 64 entries for almost every node instead of all 362 edges -- and still returns the
// reference's arg-max (ties -> lowest `orig`), see k_mcts_select.  Every order-dependent rule of the reference (PUCT ties,
// Dirichlet eta_i <-> i-th edge, most-visited ties, MCTSPolicy order) goes through `orig`.  An entry moves only when an edge is
// followed for the first time (it joins the sorted prefix; the entries it passes shift up by one and their child nodes'
// `parent_edge` follows), so a child's `parent_edge` is always the current position of its edge.
// "allocateState" (tree_search.h:174-190) is: 16-B/lane coalesced load of the parent's compact board -> Board::forward in LDS -> coalesced store.
#pragma once
#include "go_board.cuh"
#include "stl_emul.h"

namespace elfgo {

enum { NS_NOT_VISITED = 0, NS_EVAL_REQUESTED = 1, NS_VISITED = 2 };   // NodeT::VisitType
enum { LK_NN = 0, LK_TERMINAL = 1, LK_REVISIT = 2, LK_STATE_PENDING = 3 };   // PENDING: between k_mcts_select and k_mcts_leafstate
enum { MCTS_ERR_POOL = 1, MCTS_ERR_ROOT_HASH = 2, MCTS_ERR_FORWARD = 4, MCTS_ERR_RNG = 8, MCTS_ERR_VERSION = 16 };
constexpr int MCTS_KMAX = 1024;  // max rollouts per step = num_threads x rollouts_per_batch: the stride of the per-game leaf / row tables in HBM.
                                 // The leaf table of a step in LDS (k_mcts_select) is sized by the launch: 20 B per rollout of the step
constexpr int MCTS_REFILL = 8;   // steps' worth of node ids a game's stash is topped up to (k_mcts_select); stash capacity = (MCTS_REFILL + 1) x KTA
enum { GM_IDLE = 0, GM_SEARCH = 1, GM_POLICY_ONLY = 2 };   // per-game mask byte (elfmcts_set_game_mask)

struct NodeHdr {          // 64 B
  int parent;             // node id, -1 for the root
  int parent_edge;        // index of the edge in the parent's arrays
  int n_edges;            // stateActions_.size()
  int num_visits;         // numVisits_
  float V;                // V_
  float unsigned_mean_q;  // unsignedMeanQ_
  float unsigned_parent_q;
  int status;             // NS_*
  int flip;               // flipQSign_
  int has_state;          // stateType_ == NODE_STATE_SET
  int n_touched;          // edges followed at least once = length of the orig-sorted prefix of the edge arrays
  int pad0;
  u32 hash_lo, hash_hi;   // Zobrist hash of the node's position (words 12, 13): the descent feeds them to the superko filter
  int pad1[2];
};
static_assert(sizeof(NodeHdr) == 64, "NodeHdr must be 64 bytes");

struct TStat {            // statistics of a FOLLOWED edge (EdgeInfo, tree_search_base.h:102-124, minus the prior)
  float reward;           // black-positive sum
  int visits;             // num_visits
  float vloss;            // virtual_loss
  int child;              // node id of the child (always >= 0: an edge is touched when its child is created)
};
static_assert(sizeof(TStat) == 16, "TStat is one dwordx4");
// "no statistics": what a never-followed edge stands for (field by field: a braced constant would be copied from memory through scratch)
__device__ __forceinline__ TStat tst_none() { TStat t; t.reward = 0.0f; t.visits = 0; t.vloss = 0.0f; t.child = -1; return t; }
__device__ __forceinline__ TStat tst_make(float reward, int visits, float vloss, int child) {
  TStat t; t.reward = reward; t.visits = visits; t.vloss = vloss; t.child = child; return t;
}

// layout of a node record (both classes); a record is addressed as bytes
template <int N>
struct NodeL {
  static constexpr int NE = (N * N + 1 + 15) & ~15;
  static constexpr int TCS = 16;                                   // touched-edge capacity of a small record
  static constexpr int OFF_BOARD = 64;
  static constexpr int OFF_PRIOR = OFF_BOARD + (int)sizeof(CBoard<N>);
  static constexpr int OFF_COORD = OFF_PRIOR + NE * 4;
  static constexpr int OFF_ORIG = OFF_COORD + NE * 2;
  static constexpr int OFF_TST = OFF_ORIG + NE * 2;
  static constexpr int SMALL = OFF_TST + TCS * 16;
  static constexpr int BIG = OFF_TST + NE * 16;
  static_assert(OFF_PRIOR % 16 == 0 && OFF_COORD % 16 == 0 && OFF_ORIG % 16 == 0 && OFF_TST % 16 == 0, "16-B aligned arrays");
  static_assert(SMALL % 128 == 0 && BIG % 128 == 0, "records are 128-B granular");
};
static_assert(NodeL<19>::SMALL == 5888 && NodeL<19>::BIG == 11520, "19x19 node records");
static_assert(NodeL<9>::SMALL == 1920 && NodeL<9>::BIG == 3200, "9x9 node records");

template <int N>
struct NodeRef {
  using L = NodeL<N>;
  char* p;
  __device__ __forceinline__ NodeHdr& h() const { return *reinterpret_cast<NodeHdr*>(p); }
  __device__ __forceinline__ CBoard<N>& board() const { return *reinterpret_cast<CBoard<N>*>(p + L::OFF_BOARD); }
  __device__ __forceinline__ float* prior() const { return reinterpret_cast<float*>(p + L::OFF_PRIOR); }
  __device__ __forceinline__ u16* coord() const { return reinterpret_cast<u16*>(p + L::OFF_COORD); }
  __device__ __forceinline__ u16* orig() const { return reinterpret_cast<u16*>(p + L::OFF_ORIG); }
  __device__ __forceinline__ TStat* tst() const { return reinterpret_cast<TStat*>(p + L::OFF_TST); }
};

struct TreeCfg {          // TSOptions / SearchAlgoOptions (tree_search_options.h:23-229) + MCTSActorParams (go/mcts/mcts.h:17-37)
  int rollouts_per_batch;
  int virtual_loss;
  int use_prior;
  int unexplored_q_zero;
  int root_unexplored_q_zero;
  float c_puct;
  float komi;
  int ply_pass_enabled;
  int remove_pass_if_dangerous;
  int rotation_flip;
  int num_threads;        // TSOptions.num_threads (search threads per game, emulated in sequence)
  long long required_version;   // MCTSActorParams.required_version (< 0: no check)
};

struct GameState {        // 64 B per game
  int root;
  int free_top;           // number of small ids in this game's stash (popped from the context's free stack, not yet used)
  int err;
  int rng_pos;            // D4 draws consumed from d4buf this move
  int n_unique;           // leaves of the current batch
  int n_nn;               // ... of which need the net
  int row_base;
  int rollouts_done;
  long long node_visits;  // sum over rollouts of the number of visited (selected-at) nodes: mean depth = node_visits / rollouts
  int live;               // node ids this game holds (its tree; the stash is not counted)
  int promotions;         // small -> big moves so far (statistics)
  int live_peak;          // maximum of `live` since the last elfmcts_pool_info(reset) (statistics)
  int tie_hint;           // k_mcts_expand: the last reply row of this game held equal priors among its valid candidates (a HINT: which of two
                          // exact paths a row tries first; written without ordering by the rows of a step)
  int pad[2];
};
static_assert(sizeof(GameState) == 64, "GameState must be 64 bytes");

struct LeafRec {          // 32 B
  int node;
  int count;
  int kind;
  int d4;
  float value;
  int nn_index;
  int depth;              // edges between the root and this leaf (length of the trajectory)
  int thread;             // the search thread (TSOptions.num_threads) whose batch_rollouts found the leaf first
};
constexpr int MCTS_PATH_LV = 64;   // levels of a descent whose position hashes select hands to k_mcts_leafstate (deeper: exact checks only)

struct RowRec { int game, node, d4, pad; };

struct PoolTops {         // 64 B, one per context
  int small;              // ids on the context's small free stack (entries [0, small) of gstack)
  int big;                // ... big free stack
  int err;                // MCTS_ERR_* raised by the context-wide kernels (sweep)
  int pad[13];
};

template <int N>
struct TreePool {
  using L = NodeL<N>;
  char* small;            // [Cs] records of L::SMALL bytes, shared by the context's games
  char* big;              // [Cb] records of L::BIG bytes
  int* gstack;            // [Cs] free small record ids, [0, tops->small) valid
  int* gstack_big;        // [Cb] free big record ids (node id - Cs)
  PoolTops* tops;
  int* free_stack;        // [G][SC] the games' stashes of small ids (GameState.free_top of them valid)
  int* parent_of;         // [C] dense copy of NodeHdr.parent for the tree sweeps: -2 free slot, -1 root, else parent id
  int* owner;             // [C] game that holds the id (valid while parent_of != -2)
  unsigned char* keep;    // [C] scratch of treeAdvance (reachable from the next root)
  int* adv;               // [G] scratch of treeAdvance / clear: -2 game untouched, -1 free everything + fresh root, else id of the next root
  int* live_tmp;          // [G] scratch of treeAdvance: nodes kept
  GameState* gs;          // [G]
  LeafRec* leaves;        // [G][MCTS_KMAX]
  unsigned char* d4buf;   // [G][NT][W / NT]  pre-drawn rng() % 8 of each search thread's MCTSActor mt19937 (go/mcts/mcts.h:175-183)
  int* rng_pos_t;         // [G][NT] draws each search thread's actor has consumed this move
  u32* pathbuf;           // [G][KTA][MCTS_PATH_LV][2] hash (lo, hi) of the nodes on each unique leaf's path, root first (k_mcts_select -> k_mcts_leafstate)
  int KTA;                // leaves per game and step the path buffer was laid out for (num_threads x rollouts_per_batch at creation)
  const double* sqrt_tab; // [sqrt_n] host libm sqrt((double)k): the reference's std::sqrt(int) (tree_search_base.h:153)
  const unsigned char* mask;   // [G] or nullptr (every game GM_SEARCH): which games the per-game launches act on (GM_*)
  const long long* req_ver;    // [G] or nullptr (TreeCfg.required_version for every game): MCTSActorParams.required_version per game
  int sqrt_n;
  int Cs, Cb, C;          // small records, big records, node ids (Cs + Cb) of the CONTEXT
  int SC;                 // stash capacity per game ((MCTS_REFILL + 1) x KTA: the top-up + the small records a step's promotions return)
  int W, G;
  int NT;                 // search threads the D4 windows were laid out for (TSOptions.num_threads when the pool was created)
  __device__ __forceinline__ int game_mode(int g) const { return mask ? (int)mask[g] : (int)GM_SEARCH; }
};

// the records of the context (node ids are context-wide; `g` is kept in the signature for the call sites' sake)
template <int N>
struct GameNodes {
  using L = NodeL<N>;
  char* sm;
  char* bg;
  int Cs;
  __device__ __forceinline__ GameNodes(const TreePool<N>& tp, int) : sm(tp.small), bg(tp.big), Cs(tp.Cs) {}
  __device__ __forceinline__ bool is_big(int id) const { return id >= Cs; }
  __device__ __forceinline__ NodeRef<N> operator[](int id) const {
    return NodeRef<N>{id < Cs ? sm + (size_t)id * L::SMALL : bg + (size_t)(id - Cs) * L::BIG};
  }
  __device__ __forceinline__ int cap(int id) const { return id < Cs ? L::TCS : L::NE; }   // touched-edge capacity
};

__device__ __forceinline__ float rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ void mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
// monotone float -> u32 key (ascending with the value) and back
__device__ __forceinline__ u32 f2ukey(float p) { const u32 b = __float_as_uint(p); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ukey2f(u32 k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// wave-uniform copy of a node header
struct HdrU {
  int parent, parent_edge, n_edges, num_visits, status, flip, has_state, n_touched;
  float V, umq, upq;
  __device__ __forceinline__ void load(const NodeHdr* h, int lane) {
    set(lane < 16 ? reinterpret_cast<const int*>(h)[lane] : 0);
  }
  __device__ __forceinline__ void set(int w) {
    parent = rl(w, 0); parent_edge = rl(w, 1); n_edges = rl(w, 2); num_visits = rl(w, 3);
    V = __int_as_float(rl(w, 4)); umq = __int_as_float(rl(w, 5)); upq = __int_as_float(rl(w, 6));
    status = rl(w, 7); flip = rl(w, 8); has_state = rl(w, 9); n_touched = rl(w, 10);
  }
};

// Superko records of a TREE node: the reference copies the whole record map into every
// GoState (go_state.h:117-124); here the records of a node are its ancestors' positions (parent-linked
// chain up to the root) followed by the game's own records.  Same hit rule: hash, then full image.
template <int N>
struct TreeSK {
  using G = Geo<N>;
  GameNodes<N> nodes;
  int from;            // node whose state is being forwarded (the new child's parent)
  int move_out;        // the move being played from `from`
  GameSK<N> game;      // records of the game board the root was copied from
  int root_sk_len;     // records that precede the root position
  __device__ __forceinline__ void record(int, u64, u64, u64, int) const {}
  __device__ __forceinline__ bool exact_hit(int, u64 hash, u64 Bw, u64 Ww, int lane) const {
    int a = from, mv = move_out;
    bool hit = false;
    for (;;) {
      const NodeRef<N> nd = nodes[a];
      if (mv != M_PASS) {
        const u64 h = nd.board().h.hash;
        if (rfl((int)(h == hash))) {
          const int cnt = nd.board().h.hist_cnt;
          const int newest = (cnt + HIST - 1) & (HIST - 1);
          bool same = true;
          if (lane < G::R) {
            const u64 b = cnt ? nd.board().hist[newest][0][lane] : 0ull, w = cnt ? nd.board().hist[newest][1][lane] : 0ull;
            same = b == Bw && w == Ww;
          }
          if (__all(same)) hit = true;
        }
      }
      const int p = rfl(nd.h().parent);
      if (p < 0) break;
      mv = rfl((int)nodes[p].coord()[rfl(nd.h().parent_edge)]);
      a = p;
    }
    if (hit) return true;
    return game.exact_hit(root_sk_len, hash, Bw, Ww, lane);
  }
};

// ------------------------------------------------------------------------------------------------
// tree bookkeeping
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void node_init(const TreePool<N>& tp, int g, int id, int parent, int parent_edge, float parent_q, int lane) {
  const NodeRef<N> nd = GameNodes<N>(tp, g)[id];
  if (lane < 16) {
    int v = 0;
    if (lane == 0) v = parent;
    else if (lane == 1) v = parent_edge;
    else if (lane == 5 || lane == 6) v = __float_as_int(parent_q);   // NodeT ctor: unsignedMeanQ_ = unsignedParentQ_ (:99-103)
    reinterpret_cast<int*>(&nd.h())[lane] = v;
  }
  if (lane == 0) { tp.parent_of[id] = parent; tp.owner[id] = g; }
}

// Pops up to `want` ids from the context's small free stack into game g's stash (wave-uniform; one atomic).  Only pops run concurrently
// (pushes happen in the sweep kernels of a move boundary, ordered by the stream), so a pop that overshoots an almost empty stack gives
// the shortfall back and takes what was there.  Returns the new stash count.
template <int N>
__device__ __forceinline__ int stash_refill(const TreePool<N>& tp, int g, int have, int want, int lane) {
  int* fs = tp.free_stack + (size_t)g * tp.SC;
  int old = 0;
  if (lane == 0) old = atomicSub(&tp.tops->small, want);
  old = rfl(old);
  const int lo = old - want > 0 ? old - want : 0;
  const int got = old > lo ? old - lo : 0;
  if (got < want && lane == 0) atomicAdd(&tp.tops->small, want - got);
  for (int i = lane; i < got; i += 64) fs[have + i] = tp.gstack[lo + i];
  return have + got;
}
// one big record id (or -1): promotions are rare, the atomic sits on their path only
template <int N>
__device__ __forceinline__ int pop_big(const TreePool<N>& tp, int lane) {
  int id = -1;
  if (lane == 0) {
    const int old = atomicSub(&tp.tops->big, 1);
    if (old > 0) id = tp.Cs + tp.gstack_big[old - 1];
    else atomicAdd(&tp.tops->big, 1);
  }
  return rfl(id);
}

// lanes below this one whose bit is set in m
__device__ __forceinline__ int mbcnt64(u64 m) { return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
// the lane mask of a predicate as it sits in the scalar registers (HIP's __ballot goes through a 0 / 1 vector first)
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// a lane mask from the scalar registers as a per-lane predicate (no instruction: the mask IS the condition)
__device__ __forceinline__ bool lane_of(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// wave-wide maximum of a u32 on the DPP network (no LDS crossbar round trips): butterflies inside each 16-lane row
// (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8), then row_bcast:15 into rows 1/3 and row_bcast:31 into rows 2/3;
// lane 63 ends up with the maximum over all 64 lanes.  EXEC must be full.
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#define ELF_DPP_MAX(ctrl, rmask)                                                                       \
  {                                                                                                    \
    const u32 o = (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xf, false);           \
    v = o > v ? o : v;                                                                                 \
  }
  ELF_DPP_MAX(0xB1, 0xf)
  ELF_DPP_MAX(0x4E, 0xf)
  ELF_DPP_MAX(0x124, 0xf)
  ELF_DPP_MAX(0x128, 0xf)
  ELF_DPP_MAX(0x142, 0xa)
  ELF_DPP_MAX(0x143, 0xc)
#undef ELF_DPP_MAX
  return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// Every id of the context free (creation): the stacks hand out 0, 1, 2, ...
template <int N>
__global__ __launch_bounds__(256) void k_mcts_pool_init(TreePool<N> tp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < tp.Cs) tp.gstack[i] = tp.Cs - 1 - i;
  if (i < tp.Cb) tp.gstack_big[i] = tp.Cb - 1 - i;
  if (i < tp.C) { tp.parent_of[i] = -2; tp.owner[i] = -1; tp.keep[i] = 0; }
  if (i < tp.G) { tp.adv[i] = -1; tp.live_tmp[i] = 0; }
  if (i == 0) { tp.tops->small = tp.Cs; tp.tops->big = tp.Cb; tp.tops->err = 0; }
}

// SearchTreeT::clear (:411-416) for the listed games (games == nullptr: all n = G of them): adv = "free everything, fresh root"; the
// sweep + re-root launches that follow (elfmcts_clear) do the work.  Search counters are reset here.
template <int N>
__global__ __launch_bounds__(256) void k_mcts_clear_mark(TreePool<N> tp, const int32_t* games, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = games ? games[i] : i;
  tp.adv[g] = -1;
  tp.live_tmp[g] = 0;
  GameState& s = tp.gs[g];
  s.err = 0; s.rng_pos = 0; s.n_unique = 0; s.n_nn = 0; s.row_base = 0; s.rollouts_done = 0;
  for (int t = 0; t < tp.NT; ++t) tp.rng_pos_t[(size_t)g * tp.NT + t] = 0;
  // node_visits / promotions are lifetime counters (statistics): not reset with the tree
}
// TreeSearchT::setRootNodeState (tree_search.h:478-493): give the root a copy of the game's state if it
// has none; otherwise check hash equality (StateTrait::equals, go/mcts/ai.h:40-42).
template <int N, class PoolT>
__global__ __launch_bounds__(64) void k_mcts_set_root(TreePool<N> tp, PoolT pool, const int32_t* board_ids) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (rfl(tp.game_mode(g)) == GM_IDLE) return;
  const GameNodes<N> nodes(tp, g);
  GameState& s = tp.gs[g];
  const int root = rfl(s.root);
  const Slot<N>* src = &pool.slots[board_ids ? board_ids[g] : g];
  const NodeRef<N> r = nodes[root];
  if (rfl(r.h().has_state) == 0) {
    const uint4* sp = reinterpret_cast<const uint4*>(src);          // the slot's prefix is the compact board
    uint4* dp = reinterpret_cast<uint4*>(&r.board());
    for (int j = lane; j < (int)(sizeof(CBoard<N>) / 16); j += 64) dp[j] = sp[j];
    if (lane == 0) {
      const u64 hh = src->h.hash;
      r.h().has_state = 1; r.h().hash_lo = (u32)hh; r.h().hash_hi = (u32)(hh >> 32);
    }
  } else if (lane == 0 && r.board().h.hash != src->h.hash) {
    s.err |= MCTS_ERR_ROOT_HASH;
  }
  if (lane == 0) { s.rng_pos = 0; s.rollouts_done = 0; }
  for (int t = lane; t < tp.NT; t += 64) tp.rng_pos_t[(size_t)g * tp.NT + t] = 0;   // a fresh window per search thread and move
}

// The move of a node from its small record to a big one (k_mcts_select, the 17th followed edge): copy the record, give the 16
// children their parent's new id.  Kept out of line: inlined, its temporaries raised the select kernel's register count by 19 (one
// wave per SIMD less), and it runs once per 17-child node.
template <int N>
__device__ __attribute__((noinline)) void promote_record(GameNodes<N> nodes, NodeRef<N> src, NodeRef<N> dst, int* po, int bid) {
  using NL = NodeL<N>;
  const int lane = threadIdx.x & 63;
  const uint4* sp = reinterpret_cast<const uint4*>(src.p);
  uint4* dp = reinterpret_cast<uint4*>(dst.p);
  mem_sync();                              // lane 0's store of the node's running mean (this level) precedes the other lanes' reads
#pragma unroll 1
  for (int q = lane; q < NL::SMALL / 16; q += 64) dp[q] = sp[q];
  const int cch = lane < NL::TCS ? src.tst()[lane].child : 0;
  if (lane < NL::TCS) po[cch] = bid;
#pragma unroll 1
  for (int k = 0; k < NL::TCS; ++k) {      // scalar addresses: no per-lane record arithmetic on this rare path
    const NodeRef<N> cr = nodes[rl(cch, k)];
    if (lane == 0) cr.h().parent = bid;
  }
}

// ------------------------------------------------------------------------------------------------
// select: num_threads x rollouts_per_batch sequential descents per game (TreeSearchSingleThreadT::batch_rollouts, first
// half, once per search thread)
//
// findMove/UCT over a node WITHOUT touching all of its edges.  The reference scores every edge and keeps the first maximum in
// iteration order.  An edge that was never followed has N = 0, vl = 0, reward = 0: its Q is the node's first-play urgency and
// its U = (float)((double)(prior / 1) * sqrt) * c_puct is a monotone non-decreasing function of the prior (every rounding step
// is monotone, c_puct >= 0), so among the never-followed edges the maxima are the HEAD of the list sorted by descending prior --
// exactly the run whose score equals the score of its first element.  `perm` keeps [followed edges by ascending index | the
// rest by descending prior]; one round of 64 lanes scores the followed edges plus the head of that run, further rounds are
// read only while followed edges remain or the run of equal scores continues (uniform priors).  The maximum is taken over
// (score, lowest index), which is the reference's result; the FPU running mean sums the followed edges in index order, which is
// lane order.
// ------------------------------------------------------------------------------------------------
// phase markers for cycle attribution (tools/select_phases.sh builds with -DELF_PROFILE_SELECT; compiled out of the library)
#ifdef ELF_PROFILE_SELECT
__device__ unsigned long long g_select_phase[4096][8];   // per block id: no atomics, summed on the host
__device__ unsigned long long g_select_step[1024][4];    // per step (rollouts_done / KT, mod 1024): sum, max of the waves' ticks, waves, max visited nodes
#define SEL_PHASE(k) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); sel_acc[k] += _t - sel_t; sel_t = _t; } while (0)
#else
#define SEL_PHASE(k)
#endif

__device__ __attribute__((noinline)) double sqrt_beyond_table(int v) { return sqrt((double)v); }

// A/B switches (tools/gpu_r4_f.sh, gpu_r4_g.sh); measured: profiles/r04f_select_ab.txt, r04g_expand_ab.txt
#ifndef ELF_EXP_TIE_PROBE
#define ELF_EXP_TIE_PROBE 1  // k_mcts_expand looks for one pair of equal priors before it sorts, in games whose last row had one
#endif
#ifndef ELF_EXP_BLOCKED
#define ELF_EXP_BLOCKED 1    // k_mcts_expand sorts with the slots blocked 8 per lane (bitonic_sort512_blocked)
#endif
#ifndef ELF_SEL_D4PRE
#define ELF_SEL_D4PRE 1      // the next 64 pre-drawn D4 codes are read once per launch (lane l = the l-th draw), not one dependent load per net leaf
#endif

template <int N, class PoolT>
__global__ __launch_bounds__(64) void k_mcts_select(TreePool<N> tp, PoolT pool, const int32_t* board_ids, TreeCfg cfg) {
  using NL = NodeL<N>;
  __shared__ __attribute__((aligned(16))) float uqs[64 + 8];   // unsigned child Qs of one round's visited edges, compacted
  // the unique leaves of this step (unique per search thread), in first-occurrence order: five arrays of KTP entries in the launch's
  // dynamic LDS (KTP = num_threads x rollouts_per_batch rounded up to 64; elfmcts_select passes 20 x KTP bytes)
  extern __shared__ int lf_dyn[];
  const int KTP = (cfg.rollouts_per_batch * cfg.num_threads + 63) & ~63;
  int* const lf_node = lf_dyn;
  int* const lf_count = lf_dyn + KTP;
  int* const lf_meta = lf_dyn + 2 * KTP;
  int* const lf_nn = lf_dyn + 3 * KTP;
  float* const lf_value = reinterpret_cast<float*>(lf_dyn + 4 * KTP);
  const int g = blockIdx.x, lane = threadIdx.x;
  const int mode = rfl(tp.game_mode(g));
  if (mode == GM_IDLE) {                     // this game does not search in this step: no leaves, no rows
    if (lane == 0) { tp.gs[g].n_unique = 0; tp.gs[g].n_nn = 0; }
    return;
  }
  const u64 lt_mask = (1ull << lane) - 1ull;
#ifdef ELF_PROFILE_SELECT
  unsigned long long sel_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sel_t = __builtin_amdgcn_s_memtime();
#endif
  const GameNodes<N> nodes(tp, g);
  int* fs = tp.free_stack + (size_t)g * tp.SC;
  int* po = tp.parent_of;
  GameState& gs = tp.gs[g];
  int root = rfl(gs.root);
  int free_top = rfl(gs.free_top), err = 0, promotions = 0, created = 0;
  // The descent creates nodes (id, header, edge bookkeeping) but not their STATES: "allocateState" (tree_search.h:174-190: copy of the
  // parent's state + forward) of every new leaf, the terminal test and the D4 draws happen after this kernel, one wave per leaf
  // (k_mcts_leafstate, k_mcts_leafindex) -- a new leaf has no edges, so nothing in this step descends through it or needs its board.
  // What the state kernel needs from the descent is the path's position hashes (the superko filter): they arrive with every node
  // header (words 12, 13) and go to the leaf's row of the path buffer on the way down.
  u32* const pathrow0 = tp.pathbuf + (size_t)g * tp.KTA * (2 * MCTS_PATH_LV);
  int n_unique = 0, thread_start = 0, cur_t = -1;
  const float vl_f = (float)cfg.virtual_loss;
  int visited_nodes = 0;
  // GM_POLICY_ONLY = TreeSearchT::runPolicyOnly (tree_search.h:385-407): the root is evaluated if it has not been yet, nothing else
  const bool root_only = mode == GM_POLICY_ONLY;
  const int K = root_only ? 1 : cfg.rollouts_per_batch, KT = root_only ? 1 : K * cfg.num_threads;
  // The ids this step can need (one per descent) come from the game's stash; when it holds fewer, it is topped up from the context's
  // free stack with ONE atomic, before the descents -- to MCTS_REFILL steps' worth, so that a game pops once in ~MCTS_REFILL steps, and
  // the games' first pops ask for different amounts so that they do not all come back in the same step (every wave of the launch
  // starts at once: one same-address atomic per game and step serialised for ~85 us at 4608 games, measured).
  if (!root_only && free_top < KT) {
    // ... but never more than a quarter of the game's nominal share of the pool: a wide step (num_threads x rollouts per batch up to
    // 1024) on a small pool must not let the first games' stashes starve the others
    int most = (tp.Cs / tp.G) / (4 * KT);
    most = most < 1 ? 1 : most > MCTS_REFILL ? MCTS_REFILL : most;
    int steps_worth = rfl((int)(gs.rollouts_done == 0 && gs.node_visits == 0)) ? 1 + g % MCTS_REFILL : MCTS_REFILL;
    steps_worth = steps_worth > most ? most : steps_worth;
    free_top = stash_refill(tp, g, free_top, steps_worth * KT - free_top, lane);
    mem_sync();
  }

  for (int j = 0, jk = 0; j < KT; ++j, ++jk) {
    if (jk == K) jk = 0;
    if (jk == 0) { thread_start = n_unique; ++cur_t; }   // traj_counts is per batch_rollouts call, i.e. per search thread
    int node = root, depth = 0;
    HdrU h;
    // the one memory round trip of a level: the header and the first 64 entries of the scoring order, requested together
    // (speculatively: a leaf's edge arrays are never used), all coalesced.  A small record holds statistics for 16 entries only.
    int hw;
    float pr0;
    TStat ts0;
    u32 cd0, og0;
    auto request = [&](int nid) {
      const NodeRef<N> q = nodes[nid];
      hw = lane < 16 ? reinterpret_cast<const int*>(q.p)[lane] : 0;
      pr0 = q.prior()[lane];
      cd0 = q.coord()[lane];
      og0 = q.orig()[lane];
      ts0 = tst_none();
      if (nodes.is_big(nid) || lane < NL::TCS) ts0 = q.tst()[lane];
    };
    request(node);
    for (;;) {                   // single_rollout, tree_search.h:264-322
      h.set(hw);
      // the position hash of every node on the path (header words 12, 13) -> the path row of the leaf this descent will end in (slot
      // n_unique: a descent that ends in a duplicate leaf leaves its row to the next one)
      if (depth < MCTS_PATH_LV && (lane >> 1) == 6) pathrow0[(size_t)n_unique * (2 * MCTS_PATH_LV) + 2 * depth + lane - 12] = (u32)hw;
      SEL_PHASE(0);   // header + scoring order arrive
      if (h.status != NS_VISITED || h.n_edges == 0 || root_only) break;
      const NodeRef<N> nd = nodes[node];
      // ---- findMove :205-231 + UCT :361-397 + EdgeInfo::getScore (tree_search_base.h:132-157)
      float umq = h.umq;
      if (cfg.unexplored_q_zero || (cfg.root_unexplored_q_zero && depth == 0)) umq = 0.0f;
      const float fpu = h.flip ? -umq : umq;
      const int all_visits = h.num_visits + 1;
      // sqrt(N + 1) from the host-libm table through the SCALAR cache (a vector load would queue behind the edge entries in vmcnt).
      // A constant-address-space load with a wave-uniform index: the compiler emits s_load_dwordx2 and tracks its lgkmcnt itself.
      const bool sq_tab = all_visits < tp.sqrt_n;
      typedef const __attribute__((address_space(4))) u64 cu64;
      const u64 sq_bits = reinterpret_cast<cu64*>(reinterpret_cast<uintptr_t>(tp.sqrt_tab))[rfl(sq_tab ? all_visits : 0)];
      const int ne = h.n_edges, nt = h.n_touched;
      // beyond the table (a node with >= 2^17 visits: not reached by any configured search) the square root is computed in double
      // precision by a call that is kept out of line, so that the common path does not carry its ~12 FP64 instructions
      double sq = __longlong_as_double((long long)sq_bits);
      if (__builtin_expect(!sq_tab, 0)) sq = sqrt_beyond_table(all_visits);
      u32 best_key = 0, unt_key0 = 0;
      int best_e = 0x7FFFFFFF, best_pos = 0, best_child = -1, best_mv = 0;
      float best_vl = 0.0f, best_prior_v = 0.0f, tq = 0.0f;
      int tv = 0;
      float uq_r = 0.0f;
      u64 vm_r = 0;                                          // the round's unsigned child Qs and which lanes' count
      // BestAction::addAction :333-347: sequential fp32 sum of unsigned_q over edges that are not first visits, in edge order.  The sum
      // of a round is taken AFTER the round's arg-max (and, for the last round, after the chosen child's record has been requested):
      // nothing on the way to the next level waits for it.
      auto fpu_round = [&](const float uq, u64 vm) {
        const int tvr = __popcll(vm);
        tv += tvr;
        if (tvr <= 6) {
          while (vm) {
            const int l = (int)__builtin_ctzll(vm);
            vm &= vm - 1;
            tq = __fadd_rn(tq, rlf(uq, l));
          }
        } else {
          // many visited edges (the nodes near the root): the values go to LDS compacted in lane (= edge) order and every lane
          // runs the dependent chain on broadcast 16-B reads instead of one readlane per element.  The tail is padded with
          // +0.0f: the running sum starts at +0.0f and can never be -0.0f, so x + 0.0f == x bit for bit.
          if ((vm >> lane) & 1) uqs[__popcll(vm & lt_mask)] = uq;
          if (lane < 4) uqs[tvr + lane] = 0.0f;
          Board<N>::wsync();
          const float4* u4 = reinterpret_cast<const float4*>(uqs);
          const int n4 = (tvr + 3) >> 2;
          for (int c4 = 0; c4 < n4; ++c4) {
            const float4 q4 = u4[c4];
            tq = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(tq, q4.x), q4.y), q4.z), q4.w);
          }
          Board<N>::wsync();
        }
      };
      // one round of 64 entries of the scoring order; returns whether the next round has to be looked at.  The first round works
      // on the entries that arrived with the header (one instantiation without loads or waits); further rounds (nodes with more
      // than ~60 followed edges, or a long run of equal priors) are read on demand by a second instantiation.
      auto score_round = [&](const int base, const float prior, const TStat ts, const u32 cd, const int e) -> bool {
        const bool in = base + lane < ne;
        // only the followed prefix [0, nt) carries statistics; a never-followed edge has N = 0, vl = 0, reward = 0 and no child
        const bool tch = base + lane < nt;
        const float reward = tch ? ts.reward : 0.0f, vl = tch ? ts.vloss : 0.0f;
        const int nv = tch ? ts.visits : 0;
        const int ch = tch ? ts.child : -1;
        // one formula for both kinds of edge: with N = 0, vl = 0, reward = 0 it yields nvl = 0, Q = first-play urgency,
        // unsigned_q = umq and prior / 1 = prior
        float r = h.flip ? -reward : reward;
        r = __fsub_rn(r, vl);
        const int nvl = (int)__fadd_rn((float)nv, vl);                       // int + float -> float -> int
        const float q = nvl > 0 ? __fdiv_rn(r, (float)nvl) : fpu;
        const float uq = nv > 0 ? __fdiv_rn(reward, (float)nv) : umq;
        const float pp = (float)((double)__fdiv_rn(prior, (float)(1 + nv)) * sq);   // float / int, then * double sqrt, stored to float
        const float score = cfg.use_prior ? __fadd_rn(__fmul_rn(pp, cfg.c_puct), q) : q;
        // keys compare like the floats do (-0.0 == +0.0: canonicalised by + 0.0f; NaN never wins a '>': key 0)
        const u32 key = (in && score == score) ? f2ukey(__fadd_rn(score, 0.0f)) : 0u;
        uq_r = uq; vm_r = __ballot(in && nvl != 0);
        // strict '>' in iteration order = the lowest edge index among the maxima.  One 32-bit maximum of the score keys on the DPP
        // network (6 dependent steps); only when several entries share it -- equal priors among never-followed edges -- a second one
        // over their indices.  (Until round 6b: one 64-bit maximum of (key, ~index), 5 instructions per step, on every level.)
        const u32 kmax = wave_max_u32(key);
        if (kmax != 0 && kmax >= best_key) {
          const u64 mm = ballot64(key == kmax);
          int bl, emin;
          if (__popcll(mm) == 1) {
            bl = (int)__builtin_ctzll(mm);
            emin = rl(e, bl);
          } else {
            emin = 0xFFFF - (int)wave_max_u32(lane_of(mm) ? (u32)(0xFFFF - e) : 0u);
            bl = (int)__builtin_ctzll(ballot64(lane_of(mm) && e == emin));
          }
          if (kmax > best_key || emin < best_e) {
            best_key = kmax; best_e = emin; best_pos = base + bl;
            best_child = rl(ch, bl); best_mv = rl((int)cd, bl); best_vl = rlf(vl, bl); best_prior_v = rlf(prior, bl);
          }
        }
        if (base + 64 >= ne) return false;                   // every edge has been scored
        if (base + 64 <= nt) return true;                    // followed edges (or the first never-followed one) are still ahead
        if (nt >= base) unt_key0 = (u32)rl((int)key, nt - base);   // this round holds the head of the prior-sorted run
        return (u32)rl((int)key, 63) == unt_key0;            // false: the run of maximal never-followed scores ends inside this round
      };
      bool more = score_round(0, pr0, ts0, cd0, (int)og0);
      const u32 og_first = og0;                              // the first round's `orig` entries (the request below reuses the registers)
      for (int base = 64; more; base += 64) {
        fpu_round(uq_r, vm_r);
        const int pc = base + lane < ne ? base + lane : 0;
        TStat t2 = tst_none();
        if (base + lane < nt) t2 = nd.tst()[base + lane];   // more than 64 followed edges: a big record
        more = score_round(base, nd.prior()[pc], t2, (u32)nd.coord()[pc], (int)(u32)nd.orig()[pc]);
      }
      if (best_e == 0x7FFFFFFF) { err |= MCTS_ERR_FORWARD; break; }   // every score NaN: cannot happen with finite priors
      // the next level's round trip starts HERE, as soon as the child is known: the rest of this level (the last round's FPU sum, the
      // running mean, the virtual loss) runs while the child's header and scoring order are on their way
      if (best_child >= 0) request(best_child);              // existing children always own a state (created together with the node)
      fpu_round(uq_r, vm_r);
      SEL_PHASE(1);   // statistics gather, scores, reductions, FPU sum
      const float new_umq = __fdiv_rn(__fadd_rn(h.upq, tq), (float)(tv + 1));   // :227-228
      // ---- addVirtualLoss :233-251 (an edge followed for the first time gets it together with its move into the prefix below)
      const float new_vl = cfg.virtual_loss > 0 ? __fadd_rn(best_vl, vl_f) : best_vl;
      if (lane == 0) {
        nd.h().unsigned_mean_q = new_umq;
        if (cfg.virtual_loss > 0 && best_child >= 0) nd.tst()[best_pos].vloss = new_vl;
      }
      ++depth;
      int child = best_child;
      const int mv = best_mv;
      if (child < 0) {
        // ---- followEdge :280-302 -> SearchTreeT::addNode(unsignedMeanQ_) :439-443
        if (free_top <= 0) { err |= MCTS_ERR_POOL; --depth; break; }
        NodeRef<N> nw = nd;      // the record that receives the new followed edge
        if (nt >= nodes.cap(node)) {
          // ---- the 17th followed edge of a small record: the node MOVES to the big pool.  Copy the record, tell the parent (its
          // child id) and the 16 children (their parent id), return the small record to the stash.  Cb = Cs / TCS + 1 big records cannot run out.
          const int bid = pop_big(tp, lane);
          if (bid < 0) { err |= MCTS_ERR_POOL; --depth; break; }
          const NodeRef<N> dst = nodes[bid];
          promote_record<N>(nodes, nd, dst, po, bid);
          if (h.parent >= 0) {
            if (lane == 0) nodes[h.parent].tst()[h.parent_edge].child = bid;
          } else {
            root = bid;
          }
          if (lane == 0) { po[bid] = h.parent; tp.owner[bid] = g; po[node] = -2; fs[free_top] = node; }
          ++free_top;
          ++promotions;
          node = bid;
          nw = dst;
          mem_sync();            // the copy and the returned id are visible before either is used
        }
        child = rfl(fs[free_top - 1]);
        --free_top;
        ++created;
        // the edge joins the orig-sorted prefix of the scoring order at position p: entries [p, best_pos) move up by one and
        // the child nodes of the moved FOLLOWED edges learn their new position
        int p = 0;
        for (int b2 = 0; b2 < nt; b2 += 64) {
          const int pos = b2 + lane;
          const bool t = pos < nt;
          const int v = t ? (int)(b2 == 0 ? og_first : (u32)nw.orig()[pos]) : 0;
          p += __popcll(__ballot(t && v < best_e));
        }
        for (int b2 = best_pos & ~63; b2 >= (p & ~63); b2 -= 64) {   // high rounds first: a round's loads precede its stores
          const int pos = b2 + lane;
          const bool mvd = pos >= p && pos < best_pos;
          const bool tmv = mvd && pos < nt;
          const int pc = mvd ? pos : 0;
          const float mp = nw.prior()[pc];
          const u16 md = nw.coord()[pc], mo = nw.orig()[pc];
          TStat mt = tst_none();
          if (tmv) mt = nw.tst()[pos];
          if (mvd) { nw.prior()[pos + 1] = mp; nw.coord()[pos + 1] = md; nw.orig()[pos + 1] = mo; }
          if (tmv) { nw.tst()[pos + 1] = mt; nodes[mt.child].h().parent_edge = pos + 1; }
        }
        node_init(tp, g, child, node, p, new_umq, lane);
        if (lane == 0) {
          nw.prior()[p] = best_prior_v; nw.coord()[p] = (u16)mv; nw.orig()[p] = (u16)best_e;
          nw.tst()[p] = tst_make(0.0f, 0, new_vl, child);
          nw.h().n_touched = nt + 1;
        }
        SEL_PHASE(2);   // new node: id, header, scoring-order insertion
        node = child;
        h.status = NS_NOT_VISITED; h.n_edges = 0;   // the fresh node is this rollout's leaf: no need to read it back
        break;
      }
      node = child;
    }
    visited_nodes += depth;
    SEL_PHASE(5);   // slot store issue, loop exit
    if (root_only && h.status == NS_VISITED) break;   // runPolicyOnly on an evaluated root: no evaluation, no leaf
    // ---- leaf bookkeeping: batch_rollouts :211-233 (requestEvaluation, duplicate leaves of THIS thread's batch)
    int dup_idx = -1;
    for (int b0 = thread_start & ~63; b0 < n_unique; b0 += 64) {
      const int i = b0 + lane;
      const u64 dup = __ballot(i >= thread_start && i < n_unique && lf_node[i] == node);
      if (dup) { dup_idx = b0 + (int)__builtin_ctzll(dup); break; }
    }
    if (dup_idx >= 0) {
      if (lane == 0) ++lf_count[dup_idx];
    } else {
      // a leaf nobody has requested yet (a node this descent created, or the root of a fresh tree): its state, the terminal test
      // (MCTSActor::pre_evaluate) and the D4 draw follow in k_mcts_leafstate / k_mcts_leafindex
      int kind = LK_REVISIT;
      if (h.status == NS_NOT_VISITED) {
        kind = LK_STATE_PENDING;
        if (lane == 0) nodes[node].h().status = NS_EVAL_REQUESTED;
      }
      if (lane == 0) {
        lf_node[n_unique] = node; lf_count[n_unique] = 1; lf_meta[n_unique] = kind | (cur_t << 8) | (depth << 20);
        lf_value[n_unique] = 0.0f; lf_nn[n_unique] = 0;
      }
      ++n_unique;
    }
    SEL_PHASE(6);   // leaf bookkeeping (terminal test, D4 draw)
    mem_sync();
    SEL_PHASE(7);   // fence: this rollout's stores are visible to the next descent
  }
#ifdef ELF_PROFILE_SELECT
  if (lane == 0) {
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += sel_acc[k];
    const int step = (int)((gs.rollouts_done / (KT > 0 ? KT : 1)) & 1023);
    atomicAdd(&g_select_step[step][0], tot);
    atomicMax(&g_select_step[step][1], tot);
    atomicAdd(&g_select_step[step][2], 1ull);
    atomicMax(&g_select_step[step][3], (unsigned long long)visited_nodes);
  }
  if (lane < 8 && g < 4096)
    g_select_phase[g][lane] += lane == 0 ? sel_acc[0] : lane == 1 ? sel_acc[1] : lane == 2 ? sel_acc[2] : lane == 3 ? sel_acc[3]
                             : lane == 4 ? sel_acc[4] : lane == 5 ? sel_acc[5] : lane == 6 ? sel_acc[6] : sel_acc[7];
#endif
  for (int i = lane; i < n_unique; i += 64) {
    LeafRec& lr = tp.leaves[(size_t)g * MCTS_KMAX + i];
    const int meta = lf_meta[i];
    lr.node = lf_node[i]; lr.count = lf_count[i]; lr.kind = meta & 0xFF; lr.d4 = 0; lr.value = 0.0f;
    lr.nn_index = 0; lr.depth = meta >> 20; lr.thread = (meta >> 8) & 0xFFF;
  }
  if (lane == 0) {
    gs.root = root; gs.free_top = free_top; gs.n_unique = n_unique; gs.n_nn = 0;
    if (created) { const int lv = gs.live + created; gs.live = lv; if (lv > gs.live_peak) gs.live_peak = lv; }
    gs.rollouts_done += KT;
    gs.node_visits += visited_nodes;
    if (promotions) gs.promotions += promotions;
    if (err) gs.err |= err;
  }
}

// ------------------------------------------------------------------------------------------------
// leaf states: one wave per (game, unique leaf) that nobody had requested before this step.
//   a node the descent created:  allocateState, tree_search.h:174-190 -- new State(parent) + actor.forward(state, action): coalesced
//                                load of the parent's compact board -> Board::forward in LDS -> coalesced store; the superko filter of the
//                                parent's position is the game board's own Bloom words (the game's records up to the root) + the hashes
//                                the descent wrote into the leaf's path row (root .. parent; a filter may hold more, never less)
//   the root of a fresh tree:    its state is there (k_mcts_set_root)
// then MCTSActor::pre_evaluate (go/mcts/mcts.h:185-207): a terminated position is its own evaluation (+-1 by Tromp-Taylor), anything
// else needs the net.  The serial part of a step (k_mcts_select: one wave per game, 16 dependent descents) is free of all this.
// ------------------------------------------------------------------------------------------------
template <int N, class PoolT>
__global__ __launch_bounds__(64) void k_mcts_leafstate(TreePool<N> tp, PoolT pool, const int32_t* board_ids, int KT, TreeCfg cfg) {
  using GEO = Geo<N>;
  constexpr u32 BMASK = GEO::BLOOM * 32 - 1;
  __shared__ Slot<N> lds;
  const int g = blockIdx.x / KT, u = blockIdx.x % KT, lane = threadIdx.x;
  GameState& gs = tp.gs[g];
  if (u >= rfl(gs.n_unique)) return;
  LeafRec& lr = tp.leaves[(size_t)g * MCTS_KMAX + u];
  if (rfl(lr.kind) != LK_STATE_PENDING) return;
  const GameNodes<N> nodes(tp, g);
  const int node = rfl(lr.node), depth = rfl(lr.depth);
  const NodeRef<N> nd = nodes[node];
  Board<N> bd;
  bd.init(&lds, pool.zob, nullptr);
  if (rfl(nd.h().has_state) == 0) {
    const int bslot = board_ids ? board_ids[g] : g;
    const int parent = rfl(nd.h().parent);
    const NodeRef<N> pn = nodes[parent];
    const int mv = rfl((int)pn.coord()[rfl(nd.h().parent_edge)]);
    // the three inputs are requested together: the parent's board, the game's Bloom words, the path's hashes
    const u32* prow = tp.pathbuf + ((size_t)g * tp.KTA + u) * (2 * MCTS_PATH_LV);
    uint4 gb = make_uint4(0u, 0u, 0u, 0u);
    if (lane < GEO::BLOOM / 4) gb = reinterpret_cast<const uint4*>(pool.slots[bslot].bloom)[lane];
    u32 ph1 = 0, ph2 = 0;
    const bool on_path = depth <= MCTS_PATH_LV && lane < depth;
    if (on_path) { ph1 = prow[2 * lane]; ph2 = prow[2 * lane + 1]; }
    const int root_sk_len = rfl((int)nodes[rfl(gs.root)].board().h.sk_len);
    bd.load(&pn.board());
    if (depth <= MCTS_PATH_LV) {
      if (lane < GEO::BLOOM / 4) reinterpret_cast<uint4*>(lds.bloom)[lane] = gb;
      Board<N>::wsync();
      if (on_path) {
        ph1 &= BMASK; ph2 &= BMASK;
        Board<N>::lds_or(&lds.bloom[ph1 >> 5], 1u << (ph1 & 31));
        Board<N>::lds_or(&lds.bloom[ph2 >> 5], 1u << (ph2 & 31));
      }
    } else {
      for (int l = lane; l < GEO::BLOOM; l += 64) lds.bloom[l] = ~0u;   // a path longer than the row: every probe goes to the exact check
    }
    Board<N>::wsync();
    TreeSK<N> sk{nodes, parent, mv, GameSK<N>{pool.skr(bslot)}, root_sk_len};
    if (!bd.forward(mv, sk)) {
      if (lane == 0) atomicOr(&gs.err, MCTS_ERR_FORWARD);
      return;
    }
    bd.store(&nd.board());
    if (lane == 0) { nd.h().has_state = 1; nd.h().hash_lo = (u32)bd.hash; nd.h().hash_hi = (u32)(bd.hash >> 32); }
  } else {
    bd.load(&nd.board());
  }
  int kind = LK_NN;
  float value = 0.0f;
  if (bd.terminated()) {                                 // MCTSActor::pre_evaluate :185-207
    kind = LK_TERMINAL;
    value = bd.evaluate(cfg.komi) > 0.0f ? 1.0f : -1.0f;
  } else {
    // What the expansion of this leaf will need from its BOARD (k_mcts_expand, after the net has answered) is computed here, where the
    // position sits in LDS anyway: post_nn_result's pass rule (go/mcts/mcts.h:209-242: ply_pass_enabled, remove_pass_if_dangerous =
    // Tromp-Taylor of this position) and the legal-move mask (s.checkMove x N*N, :308-309).  64 B parked at the head of the node's
    // prior array (unused until the expansion writes the edges): legal words [0, R), word 7 = pass_enabled | flipQSign << 1.
    bool pass_enabled = bd.ply >= cfg.ply_pass_enabled;
    if (cfg.remove_pass_if_dangerous && pass_enabled && bd.lm0 != M_PASS) {   // :232-242
      const bool black_win = bd.evaluate(cfg.komi) > 0.0f;
      if ((black_win && bd.next_player == S_WHITE) || (!black_win && bd.next_player == S_BLACK)) pass_enabled = false;
    }
    u64 legal, cand;
    bd.template legal_moves<false>(legal, cand);
    u64* const park = reinterpret_cast<u64*>(nd.prior());
    static_assert(GEO::R <= 7, "legal words + one flag word fit 64 B");
    if (lane < GEO::R) park[lane] = legal;
    else if (lane == 7) park[7] = (u64)(pass_enabled ? 1 : 0) | ((u64)(bd.next_player == S_WHITE ? 1 : 0) << 1);
  }
  if (lane == 0) { lr.kind = kind; lr.value = value; }
}

// One wave per game, after the states: which leaves need the net (in leaf order: the row order of the batch) and their D4 codes.
// get_extractor (go/mcts/mcts.h:175-183) draws rng() % 8 once per net leaf, in the order the search thread's actor.evaluate walks its
// locked leaves -- the leaf order of that thread's batch; every search thread's MCTSActor owns a generator (all seeded alike,
// game_selfplay.cc:45-47,77): thread t's net leaves take the next codes of window t.
template <int N>
__global__ __launch_bounds__(64) void k_mcts_leafindex(TreePool<N> tp, TreeCfg cfg) {
  const int g = blockIdx.x, lane = threadIdx.x;
  GameState& gs = tp.gs[g];
  const int nu = rfl(gs.n_unique);
  if (nu == 0) { if (lane == 0) gs.n_nn = 0; return; }
  LeafRec* leaves = tp.leaves + (size_t)g * MCTS_KMAX;
  const int Wt = tp.W / tp.NT;
  int* const tpos = tp.rng_pos_t + (size_t)g * tp.NT;
  const u64 lt_mask = (1ull << lane) - 1ull;
  int n_nn = 0, err = 0;
  for (int c0 = 0; c0 < nu; c0 += 64) {
    const int i = c0 + lane;
    const bool have = i < nu;
    int kind = LK_REVISIT, th = 0;
    if (have) { kind = leaves[i].kind; th = leaves[i].thread; }
    const bool nn = have && kind == LK_NN;
    u64 todo = __ballot(nn);
    const u64 all_nn = todo;
    int d4 = 0;
    while (todo) {                                       // the search threads present in this chunk, in leaf (= thread) order
      const int t = rl(th, (int)__builtin_ctzll(todo));
      const u64 mine = __ballot(nn && th == t);
      todo &= ~mine;
      if (cfg.rotation_flip) {
        const int pos = rfl(tpos[t]);
        const int k = pos + __popcll(mine & lt_mask);
        if (nn && th == t) {
          if (k < Wt) d4 = (int)tp.d4buf[((size_t)g * tp.NT + t) * Wt + k];
          else err |= MCTS_ERR_RNG;
        }
        const int np = pos + __popcll(mine);
        if (lane == 0) { tpos[t] = np; if (t == 0) gs.rng_pos = np; }   // thread 0's actor is also the Dirichlet generator: RootInfo reports it
        mem_sync();                                      // the next chunk of the same thread reads the position back
      }
    }
    if (nn) { leaves[i].nn_index = n_nn + __popcll(all_nn & lt_mask); leaves[i].d4 = d4; }
    n_nn += __popcll(all_nn);
  }
  const bool bad = __ballot(err != 0) != 0;
  if (lane == 0) { gs.n_nn = n_nn; if (bad) gs.err |= MCTS_ERR_RNG; }
}

// rows of the net batch: game-major, leaf order within a game; row_base = exclusive prefix of n_nn over games.  One block scans
// all games once per step (every feature wave used to recompute its own prefix over all G games); it also publishes the
// total row count and the OR of the games' error bits.
template <int N>
__global__ __launch_bounds__(1024) void k_mcts_rowbase(TreePool<N> tp, int32_t* counts /* [0]=rows [1]=err-or [2..3]=u64 running total of rows */) {
  __shared__ int wsum[16], weor[16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int base = 0, eor_all = 0;
  for (int i0 = 0; i0 < tp.G; i0 += 1024) {
    const int i = i0 + tid;
    const int nn = i < tp.G ? tp.gs[i].n_nn : 0;
    int eor = i < tp.G ? tp.gs[i].err : 0;
    int inc = nn;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) eor |= __shfl_xor(eor, o, 64);
    if (lane == 63) { wsum[wv] = inc; weor[wv] = eor; }
    __syncthreads();
    int before = 0, total = 0, e = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int t = wsum[w]; before += w < wv ? t : 0; total += t; e |= weor[w]; }
    if (i < tp.G) tp.gs[i].row_base = base + before + inc - nn;
    base += total; eor_all |= e;
    __syncthreads();
  }
  if (tid == 0) {
    counts[0] = base; counts[1] = eor_all;
    *reinterpret_cast<unsigned long long*>(counts + 2) += (unsigned long long)base;   // running total (statistics)
  }
}

// One wave per (game, unique leaf): the leaf's feature row at row_base + its index among the game's net leaves.
template <int N>
__global__ __launch_bounds__(64) void k_mcts_features(TreePool<N> tp, int K, void* __restrict__ s_out, int64_t stride, int fmt, RowRec* rowmap) {
  using G = Geo<N>;
  __shared__ u64 hist[HIST][2][G::R];
  __shared__ u64 tpl[AGZ_SCRATCH_BYTES / 8];
  const int g = blockIdx.x / K, u = blockIdx.x % K, lane = threadIdx.x;
  const GameState& gs = tp.gs[g];
  if (u >= rfl(gs.n_unique)) return;
  const int base = rfl(gs.row_base);
  const LeafRec& lr = tp.leaves[(size_t)g * MCTS_KMAX + u];
  if (rfl(lr.kind) != LK_NN) return;
  const int row = base + rfl(lr.nn_index), node = rfl(lr.node), d4 = rfl(lr.d4);
  const CBoard<N>* sl = &GameNodes<N>(tp, g)[node].board();
  const u64* gh = &sl->hist[0][0][0];
  for (int j = lane; j < HIST * 2 * G::R; j += 64) (&hist[0][0][0])[j] = gh[j];
  const int cnt = sl->h.hist_cnt, player = sl->h.next_player;
  __syncthreads();
  extract_agz_row<N>(hist, tpl, cnt, player, d4, (char*)s_out + (size_t)row * stride * (fmt == FEAT_F16_NHWC ? 2 : 4), fmt, lane);
  if (lane == 0) { rowmap[row].game = g; rowmap[row].node = node; rowmap[row].d4 = d4; rowmap[row].pad = 0; }
}

// ------------------------------------------------------------------------------------------------
// expand: MCTSActor::post_nn_result / pi2response + NodeT::setEvaluation, one wave per net row
// ------------------------------------------------------------------------------------------------
template <int N>
struct ExpandLds {
  static constexpr int NA = N * N + 1;
  static constexpr int NE = NodeL<N>::NE;
  static constexpr int PP = Geo<N>::PP;
  // Two consecutive lifetimes share one region (round 6: the board is not loaded here any more -- the legal mask and the pass rule
  // arrive with the node, computed by k_mcts_leafstate):
  //   1. prob, key  -- the net's priors and coords in ACTION order: inputs of the register sort and of the exact std::sort replay
  //   2. seq..sx    -- scratch of umap_order_wave, after the sorted candidates sit in sprob/skey
  union {
    // candidate (prior, coord) pairs in ACTION order, 8 B each {prior bits, coord | valid << 16}: the input of the exact std::sort
    // replay (prior ties only), which swaps whole pairs; [NA, NE + 16) is padding that compares below every prior (the 16-wide
    // windows of the final ranks may run past the last pair)
    u64 pk[NE + 16];
    struct {
      u16 seq[NE];      // epoch insertion sequence (indices into skey)
      u16 nseq[NE];
      u16 tkey[PP];     // time of key in the current epoch, 0xFFFF = absent
      u16 sx[NE];       // exclusive prefix of group sizes at group-first times
    };
  };
  float sprob[NE];      // sorted (insertion order of the reference's map)
  u16 skey[NE];
  // 19x19: the replay's rank-indexed positions and the tie probe's table live in sprob (free while they are needed); 9x9: sprob is too
  // small for them, they use skey + scr
  static constexpr bool UD_IN_SPROB = N >= 19;
  // u16 offset into sprob of the position arrays: behind the replay's stack, heap-sort list and cut flags (SortScratch)
  static constexpr int UD_OFF = 2 * (((25 + (N * N + 1) / 17 + 1 + 3) & ~3) + (NE + 7) / 8 * 2);
  static constexpr int UD_LEN = 2 * ((N * N + 1) / 2 + 1) + 64;
  static_assert(!UD_IN_SPROB || UD_OFF + UD_LEN <= 2 * NE, "the position arrays fit sprob behind the cut flags");
  static_assert(!UD_IN_SPROB || 512 <= 2 * NE, "the tie probe's table fits sprob");
  u16 scr[UD_IN_SPROB ? 4 : NE];
  u64 legalw[8];        // legal-move bitboard words (D4-0 action order)
};
static_assert(sizeof(ExpandLds<19>) <= 5632, "expand LDS per wave: 29 waves per CU by LDS, so the 72 VGPRs decide (7 per SIMD)");
// the rank-indexed position arrays of the partitions (2 (n / 2 + 1) slots + one per lane that does not swap) / the tie probe's table
template <int N>
__device__ __forceinline__ u16* replay_ud(ExpandLds<N>& L) {
  if constexpr (ExpandLds<N>::UD_IN_SPROB) return reinterpret_cast<u16*>(L.sprob) + ExpandLds<N>::UD_OFF;
  else return L.skey;
}
template <int N>
__device__ __forceinline__ u16* probe_table(ExpandLds<N>& L) {
  if constexpr (ExpandLds<N>::UD_IN_SPROB) return reinterpret_cast<u16*>(L.sprob);
  else return L.skey;
}

// inclusive prefix sum over the 64 lanes on the DPP network: Hillis-Steele inside each row of 16 (row_shr 1 / 2 / 4 / 8, zero fill),
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  EXEC must be full.
__device__ __forceinline__ int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// iteration order of the reference's unordered_map after inserting skey[0..n) (see stl_emul.h); result in L.seq.
// One epoch = one bucket count of libstdc++'s rehash policy (13, 29, 59, 127, 257, 541): the elements so far, in their current
// iteration order, followed by the epoch's new elements, re-grouped by bucket (groups in order of first appearance, members in
// sequence order) and reversed.  The epochs are compiled one by one (round 4): the bucket count is a constant, so `key % nb` is a
// multiply-shift, the scan of an element's residue class is a fixed number of independent LDS reads issued together (34 for 13
// buckets, 2 for 257; as a run-time loop each read waited for the previous one), the early epochs (<= 59 elements) are one round
// of 64 lanes with no loop around it, and the coord -> time table is cleared once (a coord that is present is rewritten by every
// epoch, one that is absent never was written).
template <int N, int EP>
__device__ __forceinline__ void umap_epoch_wave(ExpandLds<N>& L, int n, int lane, int& have, int& done, u16*& cur, u16*& nxt) {
  constexpr int P = Geo<N>::P;     // coords are < (N+2)^2
  constexpr int NE = NodeL<N>::NE;
  constexpr int nb = stl_emul::epoch_buckets(EP);
  if (done >= n) return;
  const int take = (n < nb ? n : nb) - done;
  const int m = have + take;
  for (int t = have + lane; t < m; t += 64) cur[t] = (u16)(done + (t - have));
  done += take;
  Board<N>::wsync();
  if (nb >= P) {
    // every key has its own bucket: the epoch iterates as the plain reverse of its insertion sequence
    for (int t = lane; t < m; t += 64) nxt[m - 1 - t] = cur[t];
  } else {
    constexpr int MAXM = nb < NE ? nb : NE;            // elements this epoch can hold
    constexpr int RR = (MAXM + 63) / 64;               // rounds of 64 lanes
    constexpr int CNT = (P + nb - 1) / nb;             // coords of one residue class
    int kkv[RR];
#pragma unroll
    for (int k = 0; k < RR; ++k) {
      const int t = k * 64 + lane;
      kkv[k] = t < m ? (int)L.skey[cur[t]] : 0;
      if (t < m) L.tkey[kkv[k]] = (u16)t;
    }
    Board<N>::wsync();
    // per element: first time / size of its bucket and its rank inside the bucket, from the coords of its residue class; group
    // sizes are prefix-summed over first times
    int ftv[RR], rkv[RR];
    int carry = 0;
#pragma unroll
    for (int k = 0; k < RR; ++k) {
      ftv[k] = 0xFFFF; rkv[k] = 0;
      if (k * 64 >= m) continue;   // wave-uniform
      const int t = k * 64 + lane;
      int ft = 0xFFFF, cnt = 0, rk = 0;
      const int r0 = kkv[k] % nb;
      int tts[CNT];
#pragma unroll
      for (int j = 0; j < CNT; ++j) { const int c = r0 + j * nb; tts[j] = c < P ? (int)L.tkey[c] : 0xFFFF; }
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        const int tt = tts[j];
        if (tt != 0xFFFF) { ++cnt; ft = tt < ft ? tt : ft; rk += tt < t; }
      }
      if (!(t < m)) { ft = 0xFFFF; cnt = 0; rk = 0; }
      ftv[k] = ft; rkv[k] = rk;
      const int hv = (t < m && ft == t) ? cnt : 0;   // group size, placed at the group's first time
      const int inc = wave_inclusive_sum(hv);
      if (t < m) L.sx[t] = (u16)(carry + inc - hv);
      carry += rl(inc, 63);
    }
    Board<N>::wsync();
#pragma unroll
    for (int k = 0; k < RR; ++k) {
      const int t = k * 64 + lane;
      if (t < m) nxt[m - 1 - ((int)L.sx[ftv[k]] + rkv[k])] = cur[t];
    }
  }
  Board<N>::wsync();
  u16* const sw = cur; cur = nxt; nxt = sw;          // the new order is the next epoch's sequence: no copy
  have = m;
}

template <int N>
__device__ __forceinline__ void umap_order_wave(ExpandLds<N>& L, int n, int lane) {
  int have = 0, done = 0;
  u16* cur = L.seq;
  u16* nxt = L.nseq;
  for (int i = lane; i < ExpandLds<N>::PP; i += 64) L.tkey[i] = 0xFFFF;
  Board<N>::wsync();
  umap_epoch_wave<N, 0>(L, n, lane, have, done, cur, nxt);
  umap_epoch_wave<N, 1>(L, n, lane, have, done, cur, nxt);
  umap_epoch_wave<N, 2>(L, n, lane, have, done, cur, nxt);
  umap_epoch_wave<N, 3>(L, n, lane, have, done, cur, nxt);
  umap_epoch_wave<N, 4>(L, n, lane, have, done, cur, nxt);
  umap_epoch_wave<N, 5>(L, n, lane, have, done, cur, nxt);
  static_assert(stl_emul::kNumEpochs == 6, "one call per epoch");
  if (cur != L.seq) {                                  // an odd number of epochs ran: the result sits in nseq
    for (int t = lane; t < n; t += 64) L.seq[t] = cur[t];
    Board<N>::wsync();
  }
}

// ascending bitonic sort of 512 u64 slots held 8 per lane (slot e = k*64 + lane)
__device__ __forceinline__ void bitonic_sort512(u64 (&sx)[8], int lane) {
  constexpr int SK = 8;
#pragma unroll
  for (int size = 2; size <= 64 * SK; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64) {
#pragma unroll
        for (int k = 0; k < SK; ++k) {
          const int kp = k ^ (stride >> 6);
          if (kp > k) {
            const bool up = ((k * 64) & size) == 0;
            const u64 a = sx[k], b = sx[kp];
            const bool sw = up ? (a > b) : (a < b);
            sx[k] = sw ? b : a;
            sx[kp] = sw ? a : b;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < SK; ++k) {
          const u64 y = __shfl_xor(sx[k], stride, 64);
          const bool up = (((k * 64 + lane) & size) == 0);
          const bool keep_min = (((lane & stride) == 0) == up);
          const bool less = sx[k] < y;
          sx[k] = (keep_min == less) ? sx[k] : y;
        }
      }
    }
  }
}


// The same sort with the slots held BLOCKED, slot e = lane * 8 + k: the strides 1, 2, 4 of every merge are then exchanges between a
// lane's own registers (24 of the 45 stages), and of the 21 stages that cross lanes those with lane distance 1, 2 and 8 ride the
// DPP network (quad_perm / row_ror:8) -- 12 stages through the LDS crossbar instead of 39.  Any sorting network yields the same
// result here: the keys are distinct (they embed the payload) except for the all-ones padding.
template <int LS>
__device__ __forceinline__ u64 xor_lane(u64 v) {
  if (LS == 1) return dpp_u64<0xB1>(v);        // quad_perm [1,0,3,2]
  if (LS == 2) return dpp_u64<0x4E>(v);        // quad_perm [2,3,0,1]
  if (LS == 8) return dpp_u64<0x128>(v);       // row_ror:8
  return __shfl_xor(v, LS, 64);
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_stage_blocked(u64 (&sx)[8], int lane) {
  if (STRIDE < 8) {
    // in-register: slots k and k | STRIDE of the same lane
    const bool dn_lane = SIZE >= 8 && SIZE < 512 && (lane & (SIZE >> 3)) != 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k & STRIDE) continue;
      const int kp = k | STRIDE;
      const u64 a = sx[k], b = sx[kp];
      const bool dn = SIZE < 8 ? ((k & SIZE) != 0) : dn_lane;
      const bool sw = (a > b) != dn;            // equal slots (padding) may swap: same value
      sx[k] = sw ? b : a;
      sx[kp] = sw ? a : b;
    }
  } else {
    constexpr int LS = STRIDE >> 3;
    const bool up = SIZE >= 512 || (lane & (SIZE >> 3)) == 0;
    const bool keep_min = ((lane & LS) == 0) == up;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u64 y = xor_lane<LS>(sx[k]);
      const bool less = sx[k] < y;
      sx[k] = (keep_min == less) ? sx[k] : y;
    }
  }
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void bitonic_merge_blocked(u64 (&sx)[8], int lane) {
  bitonic_stage_blocked<SIZE, STRIDE>(sx, lane);
  if constexpr (STRIDE > 1) bitonic_merge_blocked<SIZE, STRIDE / 2>(sx, lane);
}
template <int SIZE>
__device__ __forceinline__ void bitonic_sizes_blocked(u64 (&sx)[8], int lane) {
  bitonic_merge_blocked<SIZE, SIZE / 2>(sx, lane);
  if constexpr (SIZE < 512) bitonic_sizes_blocked<SIZE * 2>(sx, lane);
}
// ascending; slot e = lane * 8 + k
__device__ __forceinline__ void bitonic_sort512_blocked(u64 (&sx)[8], int lane) { bitonic_sizes_blocked<2>(sx, lane); }



// __move_median_to_first's choice among (a, b, c) = (first + 1, mid, last - 1) for wave-uniform values: 0 = a, 1 = b, 2 = c.
// The if-chain of libstdc++ (`a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b))`, ties included) as a table
// indexed by its three comparisons.  Computed by every lane alike, on the VECTOR unit: a CU has four of those and ONE scalar unit
// (1.00 scalar instruction per clock and CU against 1.82 vector ones, "Issue roofs" in DESIGN.md), and the replay's control flow
// keeps the scalar unit busy enough.
__device__ __forceinline__ int median3_choice(float va, float vb, float vc) {
  const int code = (va > vb ? 4 : 0) | (vb > vc ? 2 : 0) | (va > vc ? 1 : 0);
  return (0x5821 >> (2 * code)) & 3;
}
// a wave-uniform value as a vector register (keeps what is computed from it on the vector unit)
__device__ __forceinline__ float in_vgpr(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int in_vgpr(int v) { asm volatile("" : "+v"(v)); return v; }

// One __unguarded_partition of a segment with wave-uniform bounds by the whole wave -- stl_emul.h (e):
//   * with nub(e) = up-stops (v <= P) strictly before e and nda(e) = down-stops (v >= P) strictly after e, both inside the segment:
//     the up-stop at e is U[nub] and swaps iff nda > nub; the down-stop at e is D[nda] and swaps iff nub > nda
//   * ud[2 t] / ud[2 t + 1]: position of the swapping up-stop / down-stop of rank t (t <= n / 2); every swapper stores ITS OWN OLD pair
//     at its partner's place
//   * the cut is the lowest position that holds an up-stop which does not swap or a down-stop which does, else `last`
// The two stop flags and the two swap flags are lane masks that come straight out of a compare each (lanes outside the segment
// compare a NaN; a lane that is no up-stop compares against INT_MAX); the cut is a scalar find-first-set.  No lane is switched off
// anywhere: a lane that does not swap stores its position in a slot of its own, reads it back and rewrites its own pair in place.
//
// partition_small_wave: a segment of <= 64 pairs, pivot selection included.  Lane i holds pair first + i, so __move_median_to_first
// reads its three candidates from registers, the pivot swap is two lanes storing their pairs at each other's place plus one register
// move, and the partition needs no further load.  Most of a row's ~36 partitions are this kind.
template <int N>
__device__ __forceinline__ int partition_small_wave(ExpandLds<N>& L, const int first, const int last, const int lane) {
  constexpr int DUMMY = 2 * ((N * N + 1) / 2 + 1);          // ud slots [DUMMY, DUMMY + 64): one per lane that does not swap
  static_assert(DUMMY + 64 <= ExpandLds<N>::UD_LEN, "slots of the lanes that do not swap");
  u16* const ud = replay_ud<N>(L);
  u64* const pk = L.pk;
  const int len = last - first;                             // 17 .. 64
  u64 x = pk[first + lane];                                 // lanes >= len read pairs beyond the segment: masked below
  const float v0 = __uint_as_float((u32)x);
  const int ib = len >> 1, ic = len - 1;
  const int ch = median3_choice(in_vgpr(rlf(v0, 1)), in_vgpr(rlf(v0, ib)), in_vgpr(rlf(v0, ic)));
  const int im = rfl(ch == 0 ? 1 : (ch == 1 ? ib : ic));
  const float P = rlf(v0, im);
  // the pair at `first` and the median change places: in LDS now (every other lane rewrites its own pair where it is), and in lane
  // im's register (lane 0 takes no part in the partition)
  const bool is_m = lane == im;
  pk[first + (is_m ? 0 : (lane == 0 ? im : lane))] = x;
  {
    const u32 lo0 = (u32)rl((int)(u32)x, 0), hi0 = (u32)rl((int)(u32)(x >> 32), 0);
    if (is_m) x = ((u64)hi0 << 32) | lo0;
  }
  const bool in = (u32)(lane - 1) < (u32)(len - 1);         // lanes 1 .. len - 1
  const float v = in ? __uint_as_float((u32)x) : __uint_as_float(0x7FC00000u);   // NaN: neither stop
  const bool u = v <= P, d = v >= P;
  const u64 mu = ballot64(u), md = ballot64(d);
  const int nub = mbcnt64(mu);
  const int nda = (int)__popcll(md) - mbcnt64(md) - (d ? 1 : 0);      // down-stops above this lane
  const bool su = nda > (u ? nub : 0x7FFFFFFF), sd = nub > (d ? nda : 0x7FFFFFFF);
  const u64 msu = ballot64(su), msd = ballot64(sd);
  const int slot = (su || sd) ? 2 * (su ? nub : nda) + (sd ? 1 : 0) : DUMMY + lane;
  ud[slot] = (u16)(first + lane);
  Board<N>::wsync();
  // (lane 0 still holds the pair that sat at `first` before the pivot moved there: its store goes to the last padding pair, which no
  // window of the final ranks reaches)
  const int partner = ud[(su || sd) ? (slot ^ 1) : slot];
  pk[lane == 0 ? ExpandLds<N>::NE + 15 : partner] = x;
  Board<N>::wsync();
  const u64 mF = (mu & ~msu) | msd;
  return mF != 0 ? first + (int)__builtin_ctzll(mF) : last;
}

// partition_segment_wave: longer segments, pivot value P already at `first`; element first + 1 + 64 j + lane in round j of RM.
template <int N, int RM>
__device__ __forceinline__ int partition_segment_wave(ExpandLds<N>& L, const int first, const int last, const float P, const int lane) {
  constexpr int DUMMY = 2 * ((N * N + 1) / 2 + 1);          // ud slots [DUMMY, DUMMY + 64): written by lanes that do not swap, never read
  u16* const ud = replay_ud<N>(L);
  u64* const pk = L.pk;
  u64 x[RM], mu[RM], md[RM];
  int cu[RM], cd[RM];
  int nu = 0, nd = 0;
#pragma unroll
  for (int j = 0; j < RM; ++j) {
    x[j] = 0; mu[j] = 0; md[j] = 0; cu[j] = nu; cd[j] = nd;
    const int left = last - (first + 1 + 64 * j);           // elements of this round and the following ones
    if (j > 0 && left <= 0) continue;                       // wave-uniform
    x[j] = pk[first + 1 + 64 * j + lane];                   // lanes beyond the segment read pairs that compare as NaN below
    const float v = lane < left ? __uint_as_float((u32)x[j]) : __uint_as_float(0x7FC00000u);
    mu[j] = ballot64(v <= P);
    md[j] = ballot64(v >= P);
    nu += __popcll(mu[j]);
    nd += __popcll(md[j]);
  }
  u64 mF[RM], msw[RM];
  int slot[RM];                                             // this lane's slot in ud; its partner's is slot ^ 1
#pragma unroll
  for (int j = 0; j < RM; ++j) {
    mF[j] = 0; msw[j] = 0; slot[j] = 0;
    if (j > 0 && first + 1 + 64 * j >= last) continue;
    const bool u = lane_of(mu[j]), d = lane_of(md[j]);
    const int nub = cu[j] + mbcnt64(mu[j]);
    const int nda = (nd - cd[j]) - mbcnt64(md[j]) - (d ? 1 : 0);     // down-stops strictly after this element
    const bool su = nda > (u ? nub : 0x7FFFFFFF), sd = nub > (d ? nda : 0x7FFFFFFF);
    const u64 msu = ballot64(su), msd = ballot64(sd);
    mF[j] = (mu[j] & ~msu) | msd;
    msw[j] = msu | msd;
    slot[j] = (su || sd) ? 2 * (su ? nub : nda) + (sd ? 1 : 0) : DUMMY + lane;
    ud[slot[j]] = (u16)(first + 1 + 64 * j + lane);
  }
  Board<N>::wsync();
#pragma unroll
  for (int j = 0; j < RM; ++j) {
    if (j > 0 && first + 1 + 64 * j >= last) continue;
    const int partner = ud[slot[j] ^ 1];                    // (a lane that does not swap reads a slot nobody cares about)
    pk[lane_of(msw[j]) ? partner : first + 1 + 64 * j + lane] = x[j];
  }
  Board<N>::wsync();
  int cut = last;
#pragma unroll
  for (int j = RM - 1; j >= 0; --j)
    if (mF[j] != 0) cut = first + 1 + 64 * j + (int)__builtin_ctzll(mF[j]);
  return cut;
}

// std::sort(pairs, a.second > b.second) (go/mcts/mcts.h:292-297) over the pairs L.pk[0, n), exact, as the wave runs it since round 6b
// (stl_emul.h (e), checked against std::sort on the host): the partition tree of __introsort_loop is walked ONE SEGMENT AT A TIME by
// the whole wave -- a row of 362 priors has ~36 partitions, all but ~9 of <= 64 pairs (one round of 64 lanes), and with the
// segment's bounds and pivot in scalar registers a partition is two compares, four lane counts, one u16 store / load and one pair
// store per element.  (Rounds 5-6a partitioned all segments of one recursion depth at once: ~10 generations x 6 rounds x 5 passes of
// per-element segment tables, 9 069 vector instructions per row with ties against ~5 200 now.)
// The benchmark's random-init fp16 net answers with a near-uniform policy on the fp16 grid (~240 distinct values among 362 priors),
// so EVERY row of the headline takes this path.  Scratch (all dead outside this function and the window ranks that follow it):
//   sprob as u32:  [0, 24) the stack of segments that wait (first | last << 9 | depth << 18); [24] number of heap-sort segments,
//                  [25, 25 + MS) the list of them; from CUT on, one byte per position: "a segment the loop leaves starts here"
//   replay_ud():   the rank-indexed positions of the partitions (at most n / 2 swaps: 2 (n / 2 + 1) u16, + one slot per lane that does
//                  not swap): 19x19 behind the cut flags in sprob, 9x9 in skey + scr
template <int N>
struct SortScratch {
  static constexpr int NE = ExpandLds<N>::NE;
  static constexpr int MS = (N * N + 1) / 17 + 1;          // segments longer than 16 that n pairs can hold
  static constexpr int STK = 0, TODO = 24, CUT = (25 + MS + 3) & ~3;   // u32 offsets into sprob
  static constexpr int CUTW = (NE + 7) / 8 * 2;            // u32 words of flag bytes (8-byte granular)
  static_assert((CUT + CUTW) * 4 <= NE * 4, "the cut flags fit sprob");
  static_assert(CUTW <= 128, "two words per lane clear the cut flags");
};

template <int N>
__device__ __forceinline__ void introsort_segments_wave(ExpandLds<N>& L, const int n, const int lane) {
  using SS = SortScratch<N>;
  constexpr int NE = ExpandLds<N>::NE;
  static_assert(ExpandLds<N>::UD_IN_SPROB || offsetof(ExpandLds<N>, scr) == offsetof(ExpandLds<N>, skey) + NE * 2, "skey and scr are one array of 2 NE u16");
  static_assert(!ExpandLds<N>::UD_IN_SPROB || ExpandLds<N>::UD_OFF == 2 * (SS::CUT + SS::CUTW), "the position arrays start behind the cut flags");
  u32* const w32 = reinterpret_cast<u32*>(L.sprob);
  u32* const stk = w32 + SS::STK;
  u32* const todo = w32 + SS::TODO;
  unsigned char* const cutf = reinterpret_cast<unsigned char*>(w32 + SS::CUT);
  u32* const pk32 = reinterpret_cast<u32*>(L.pk);
  if (lane == 0) todo[0] = 0u;
  if (2 * lane < SS::CUTW) { w32[SS::CUT + 2 * lane] = lane == 0 ? 1u : 0u; w32[SS::CUT + 2 * lane + 1] = 0u; }   // position 0 starts a segment
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) depth += 2;              // 2 * floor(lg n)
  int first = 0, last = n, sp = 0;
  for (;;) {
    while (last - first > 16) {
      if (ELF_RARE(depth == 0)) {                           // __partial_sort fallback (median-of-3 killers): after the loop
        if (lane == 0) { const u32 c = todo[0]; todo[0] = c + 1u; todo[1 + (c < (u32)SS::MS ? c : 0u)] = (u32)first | ((u32)last << 16); }
        break;
      }
      --depth;
      int cut;
      if (last - first <= 64) {
        cut = partition_small_wave<N>(L, first, last, lane);
      } else {
        // __move_median_to_first(first, first + 1, mid, last - 1): three reads, a scalar select chain, one swap of pairs
        const int a = first + 1, b = first + ((last - first) >> 1), c = last - 1;
        const float pv = __uint_as_float(pk32[2 * (lane == 0 ? a : (lane == 1 ? b : c))]);
        const int ch = rfl(median3_choice(in_vgpr(rlf(pv, 0)), in_vgpr(rlf(pv, 1)), in_vgpr(rlf(pv, 2))));
        const int m = ch == 0 ? a : (ch == 1 ? b : c);
        const float P = rlf(pv, ch);
        if (lane < 2) {
          const u64 xx = L.pk[lane == 0 ? first : m];
          L.pk[lane == 0 ? m : first] = xx;
        }
        Board<N>::wsync();
        if (last - first <= 129) cut = partition_segment_wave<N, 2>(L, first, last, P, lane);
        else cut = partition_segment_wave<N, (N * N + 63) / 64>(L, first, last, P, lane);
      }
      // every cut starts a segment; of two parts that are both longer than 16 the left one waits (any order: disjoint)
      if (lane == 0) cutf[cut] = 1;
      if (last - cut > 16) {
        if (cut - first > 16) {
          if (lane == 0) stk[sp] = (u32)first | ((u32)cut << 9) | ((u32)depth << 18);
          sp = rfl(sp + 1);
        }
        first = cut;
      } else {
        last = cut;
      }
    }
    if (sp == 0) break;
    --sp;
    Board<N>::wsync();
    const u32 w = (u32)rfl((int)stk[sp]);
    first = (int)(w & 0x1FFu); last = (int)((w >> 9) & 0x1FFu); depth = (int)(w >> 18);
  }
  Board<N>::wsync();
  const int n_heap = rfl((int)todo[0]);
  if (ELF_RARE(n_heap > 0)) {
    // median-of-3 killers only: serial heap sorts, one segment after the other; a heap-sorted segment is in its final order, so every
    // position of it is a segment of its own
    for (int i = 0; i < n_heap && i < SS::MS; ++i) {
      const u32 fl = (u32)rfl((int)todo[1 + i]);
      const int hf = (int)(fl & 0xFFFFu), hl = (int)(fl >> 16);
      if (lane == 0) stl_emul::heap_sort(stl_emul::PairRefInterleaved{{pk32}, {pk32}}, hf, hl);
      for (int q = hf + lane; q < hl; q += 64) cutf[q] = 1;
      Board<N>::wsync();
    }
  }
}

// phase markers for cycle attribution (make FLAGS+=-DELF_PROFILE_EXPAND on the GPU box; compiled out of the library)
#ifdef ELF_PROFILE_EXPAND
__device__ unsigned long long g_expand_phase[65536][8];   // per block id: no atomics, summed on the host
__device__ unsigned long long g_expand_rowmax[65536];     // per block id: longest row (ticks) over all launches
#define EXP_PHASE(k) do { unsigned long long _t = __builtin_amdgcn_s_memtime(); ph_acc[k] += _t - ph_t; ph_t = _t; } while (0)
#else
#define EXP_PHASE(k)
#endif

// at least 6 waves per SIMD asked for; round 6b: 60 VGPRs and 5.4 KB of LDS (29 waves per CU by LDS: 7.25 per SIMD)
template <int N>
__global__ __launch_bounds__(64, 6) void k_mcts_expand(TreePool<N> tp, const RowRec* rowmap, const float* __restrict__ pi,
                                                     int64_t pi_stride, const float* __restrict__ value, const int64_t* __restrict__ rv,
                                                     int n_rows_host, const int32_t* __restrict__ counts, TreeCfg cfg) {
  using G = Geo<N>;
  constexpr int NA = N * N + 1, R = (NA + 63) / 64;
  __shared__ ExpandLds<N> L;
  const int row = blockIdx.x, lane = threadIdx.x;
  // n_rows_host < 0: the row count of the last select stays on the device (no host round trip per step); the grid then
  // covers the maximum and surplus blocks leave here
  const int n_rows = n_rows_host >= 0 ? n_rows_host : rfl(counts[0]);
  if (row >= n_rows) return;
  const int g = rowmap[row].game, node = rowmap[row].node, d4 = rowmap[row].d4;
  const NodeRef<N> nd = GameNodes<N>(tp, g)[node];
  // MCTSActor::post_nn_result :210-217: the reply's model version must be the requested one (the reference throws)
  const long long need_ver = tp.req_ver ? tp.req_ver[g] : cfg.required_version;
  if (rv != nullptr && need_ver >= 0 && lane == 0 && rv[row] != need_ver) atomicOr(&tp.gs[g].err, MCTS_ERR_VERSION);
#ifdef ELF_PROFILE_EXPAND
  unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime();
#endif
  // what the expansion needs from the leaf's BOARD was computed by k_mcts_leafstate while the position sat in LDS (the pass rule of
  // post_nn_result :209-242 and the legal-move mask, s.checkMove :308-309): 64 B at the head of the node's prior array
  const u64* const park = reinterpret_cast<const u64*>(nd.prior());
  const u64 pw = lane < 8 ? park[lane] : 0ull;
  const u32 pflags = (u32)__builtin_amdgcn_readlane((int)(u32)pw, 7);
  const bool pass_enabled = (pflags & 1u) != 0;
  const int flip = (int)((pflags >> 1) & 1u);
  EXP_PHASE(0);   // row map + parked mask
  // ---- pi2response :256-332.  The reference sorts all N*N+1 (coord, prior) pairs by prior (descending) and then keeps
  // the valid ones in that order.  If no two VALID candidates share a prior the result is the unique descending order
  // of the valid ones: a register bitonic sort of 512 slots (8 per lane), invalid candidates keyed last.
  EXP_PHASE(1);   // (round 5: pass rule + legal mask; now in k_mcts_leafstate)
  const float* prow = pi + (size_t)row * pi_stride;
  if (lane < G::R) L.legalw[lane] = pw;
  Board<N>::wsync();
  constexpr int SK = 8;   // 512 sort slots: element e = k*64 + lane
  u64 sx[SK];
  int nvalid = 0;
  // Equal priors among the valid candidates send a row to the exact std::sort replay below, which needs nothing from the register sort.
  // Whether a row has them is only known after that sort -- unless the game's previous row had them (fp16 nets tie in every row): then
  // a probe looks for ONE equal pair first (a 512-slot table of action ids keyed by a hash of the prior bits: last writer wins, a reader
  // that finds another id compares the two priors).  A hit skips the register sort; a miss (no ties, or all of them hidden by
  // collisions) costs ~100 instructions and falls through to the sort and its exact tie test.  Either way the result is the same.
  const bool probe = ELF_EXP_TIE_PROBE && rfl(tp.gs[g].tie_hint) != 0;
  u16* const ptab = probe_table<N>(L);                       // 19x19: sprob (free until the sort); 9x9: skey + scr
  constexpr int PSH = 2 * ExpandLds<N>::NE >= 512 ? 23 : 25; // 512 / 128 slots
  static_assert((1 << (32 - PSH)) <= 2 * ExpandLds<N>::NE, "the probe's table fits skey + scr");
#pragma unroll
  for (int k = 0; k < SK; ++k) {
    const int i = ELF_EXP_BLOCKED ? lane * SK + k : k * 64 + lane;   // sort slot <-> action id (any bijection does)
    bool valid = false;
    int coord = 0;
    float p = 0.0f;
    if (i < NA) {
      int a0;
      action_to_coord<N>(i, d4, coord, a0);
      p = prow[i];
      if (coord == M_PASS) valid = pass_enabled;
      else valid = (L.legalw[a0 >> 6] >> (a0 & 63)) & 1;      // s.checkMove(v.first) :308-309
      // kept in action order for the exact std::sort replay
      L.pk[i] = (u64)__float_as_uint(p) | ((u64)((u32)coord | (valid ? 0x10000u : 0u)) << 32);
      if (probe && valid) ptab[(__float_as_uint(p) * 0x9E3779B1u) >> PSH] = (u16)i;
    } else if (i < ExpandLds<N>::NE + 16) {
      L.pk[i] = 0xFF800000ull;                               // -inf, invalid
    }
    nvalid += __popcll(__ballot(valid));
    // monotone float -> uint key (ascending with the value), inverted for a descending sort; coord as payload
    const u32 bits = __float_as_uint(p);
    const u32 ukey = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    sx[k] = valid ? (((u64)(~ukey) << 32) | (u32)coord) : ~0ull;
  }
  static_assert(ExpandLds<N>::NE + 16 <= 64 * SK, "the key loop also writes the padding of pk");
  EXP_PHASE(2);   // reply read, action->coord, validity, sort keys
  int n = 0;   // number of edges
  bool tie = false;
  float sp[R];
#pragma unroll
  for (int k = 0; k < R; ++k) sp[k] = 0.0f;
  if (nvalid == 0) {
    // "Add pass if there is no valid move" :321-324 (only reachable with pass disabled); normalize: 1 / (1e-10 + 1)
    if (lane == 0) { L.skey[0] = M_PASS; L.sprob[0] = __fdiv_rn(1.0f, __fadd_rn(1e-10f, 1.0f)); }
    n = 1;
  } else {
    n = nvalid;
    if (probe) {
      Board<N>::wsync();
      const u32* const pk32 = reinterpret_cast<const u32*>(L.pk);
#pragma unroll
      for (int k = 0; k < SK; ++k) {
        const int i = ELF_EXP_BLOCKED ? lane * SK + k : k * 64 + lane;
        if (sx[k] != ~0ull) {                                 // valid
          const u32 bits = pk32[2 * i];
          const int o = ptab[(bits * 0x9E3779B1u) >> PSH];
          tie |= o != i && pk32[2 * o] == bits;
        }
      }
      tie = __any(tie);
      Board<N>::wsync();
    }
    if (!tie) {
    if (ELF_EXP_BLOCKED) {
      bitonic_sort512_blocked(sx, lane);
      EXP_PHASE(3);   // register bitonic sort of 512 slots
      // sorted element e = lane*8 + k: prior back from the key, coord from the payload; the rest of the kernel holds element
      // k*64 + lane in sp[k], read back from LDS
#pragma unroll
      for (int k = 0; k < SK; ++k) {
        const int e = lane * SK + k;
        const u32 ukey = ~(u32)(sx[k] >> 32);
        const u32 bits = (ukey & 0x80000000u) ? (ukey & 0x7FFFFFFFu) : ~ukey;
        if (e < n) { L.sprob[e] = __uint_as_float(bits); L.skey[e] = (u16)(sx[k] & 0xFFFFu); }
      }
      Board<N>::wsync();
#pragma unroll
      for (int k = 0; k < R; ++k) { const int e = k * 64 + lane; sp[k] = e < n ? L.sprob[e] : 0.0f; }
    } else {
    bitonic_sort512(sx, lane);
    EXP_PHASE(3);   // register bitonic sort of 512 slots
    // sorted element e = k*64 + lane (e < n <= N*N+1 <= 6*64): prior back from the key, coord from the payload
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int e = k * 64 + lane;
      const u32 ukey = ~(u32)(sx[k] >> 32);
      const u32 bits = (ukey & 0x80000000u) ? (ukey & 0x7FFFFFFFu) : ~ukey;
      sp[k] = __uint_as_float(bits);
      if (e < n) { L.sprob[e] = sp[k]; L.skey[e] = (u16)(sx[k] & 0xFFFFu); }
    }
    }
    Board<N>::wsync();
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int e = k * 64 + lane;
      if (e + 1 < n) tie |= (sp[k] == L.sprob[e + 1]);
    }
    tie = __any(tie);
    }
    if (tie) {
      // equal priors among valid candidates: the order is whatever libstdc++'s unstable std::sort makes of ALL N*N+1 pairs
      // (go/mcts/mcts.h:292-297).  Replayed exactly: the introsort loop on the action-order pairs in LDS, one segment at a time
      // (introsort_segments_wave); __final_insertion_sort as what it is, the stable sort of every segment the loop leaves -- a
      // pair's final place is its segment's start plus the pairs of a 16-wide window that go before it; then the filter.
      EXP_PHASE(4);
      introsort_segments_wave<N>(L, NA, lane);
      EXP_PHASE(1);   // the introsort loop of the exact std::sort replay (prior ties only)
      {
        const u32* const pk32 = reinterpret_cast<const u32*>(L.pk);
        const unsigned char* const cutf = reinterpret_cast<const unsigned char*>(reinterpret_cast<const u32*>(L.sprob) + SortScratch<N>::CUT);
        // segment starts as lane masks: word k = positions 64 k .. 64 k + 63
        u64 bw[R];
#pragma unroll
        for (int k = 0; k < R; ++k) bw[k] = ballot64(cutf[k * 64 + lane] != 0);   // bytes past NE: scratch of this block, masked by e < NA below
        const u64 le_mask = ~0ull >> (63 - lane);            // lanes 0 .. lane
        u64 xe[R];
        int fp[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int e = k * 64 + lane;
          xe[k] = 0; fp[k] = -1;
          if (e < NA) {
            // the highest segment start at or below e: in this word, else the highest of the word before (a segment has <= 16 pairs)
            const u64 m = bw[k] & le_mask;
            int sf;
            if (k == 0) sf = 63 - (int)__builtin_clzll(m);                   // bit 0 is always set
            else sf = m != 0 ? k * 64 + 63 - (int)__builtin_clzll(m) : k * 64 - 1 - (int)__builtin_clzll(bw[k - 1]);
            xe[k] = L.pk[e];
            const float v = __uint_as_float((u32)xe[k]);
            const u32* const win = pk32 + 2 * sf;
            // r: pairs of the window that compare above this one; eq: bit 15 - t = the pair at sf + t compares equal
            u32 r = 0, eq = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
              const float vj = __uint_as_float(win[2 * t]);
              r += vj > v ? 1u : 0u;
              eq = eq + eq + (vj == v ? 1u : 0u);
            }
            fp[k] = sf + (int)r + __popc(eq >> (16 - (e - sf)));   // e - sf <= 15; e == sf: eq >> 16 = 0
          }
        }
        Board<N>::wsync();
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (fp[k] >= 0) L.pk[fp[k]] = xe[k];
        Board<N>::wsync();
        // the filter (s.checkMove :308-309), in sorted order
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const int e = k * 64 + lane;
          const u64 y = L.pk[e < NA ? e : 0];
          const bool valid = e < NA && ((y >> 48) & 1ull) != 0;
          const u64 mv = __ballot(valid);
          if (valid) {
            const int dst = cnt + mbcnt64(mv);
            L.sprob[dst] = __uint_as_float((u32)y);
            L.skey[dst] = (u16)((y >> 32) & 0xFFFFu);
          }
          cnt += __popcll(mv);
        }
      }
      Board<N>::wsync();
#pragma unroll
      for (int k = 0; k < R; ++k) { const int e = k * 64 + lane; sp[k] = e < n ? L.sprob[e] : 0.0f; }
    }
    if (ELF_EXP_TIE_PROBE && lane == 0 && probe != tie) tp.gs[g].tie_hint = tie ? 1 : 0;
    EXP_PHASE(4);   // sorted rows to LDS, tie test (+ exact std::sort replay on ties)
    // normalize :244-254: total = 1e-10 + sequential fp32 sum in sorted order.  The chain of n dependent adds is the cost; every
    // lane runs it redundantly on broadcast 16-B LDS reads of the sorted priors (issued ahead of the adds), which beats a
    // readlane per element.  The tail is padded with +0.0f: x + 0.0f == x for the positive running total, so summing to the
    // next multiple of 16 is the same sequence of roundings.
    for (int e = n + lane; e < ExpandLds<N>::NE; e += 64) L.sprob[e] = 0.0f;
    Board<N>::wsync();
    float total = 1e-10f;
    {
      const float4* s4 = reinterpret_cast<const float4*>(L.sprob);
      const int n16 = (n + 15) >> 4;
      for (int c = 0; c < n16; ++c) {
        const float4 q0 = s4[4 * c], q1 = s4[4 * c + 1], q2 = s4[4 * c + 2], q3 = s4[4 * c + 3];
        total = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(total, q0.x), q0.y), q0.z), q0.w);
        total = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(total, q1.x), q1.y), q1.z), q1.w);
        total = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(total, q2.x), q2.y), q2.z), q2.w);
        total = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(total, q3.x), q3.y), q3.z), q3.w);
      }
    }
    Board<N>::wsync();
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int e = k * 64 + lane;
      if (e < n) L.sprob[e] = __fdiv_rn(sp[k], total);
    }
  }
  Board<N>::wsync();
  EXP_PHASE(5);   // sequential fp32 normalisation
  // ---- setEvaluation :176-203: insert in this order; store edges in the map's ITERATION order
  umap_order_wave<N>(L, n, lane);
  EXP_PHASE(6);   // unordered_map iteration order
  // the edges are stored in scoring order: no edge followed yet, all of them by descending prior = pi2response's sorted order;
  // `orig` carries each edge's index in the map's iteration order
  // (a never-followed edge has no statistics record: 8 B per legal move leave for HBM, not 24)
  {
    float* const e_prior = nd.prior();
    u16* const e_coord = nd.coord();
    u16* const e_orig = nd.orig();
    for (int jn = lane; jn < n; jn += 64) {
      const int src = L.seq[jn];
      e_prior[src] = L.sprob[src];
      e_coord[src] = L.skey[src];
      e_orig[src] = (u16)jn;
    }
  }
  if (lane == 0) {
    NodeHdr& nh = nd.h();
    nh.n_touched = 0;
    nh.n_edges = n;
    nh.V = value[row];                                  // resp->value = reply.value :222
    nh.flip = flip;                                     // pre_evaluate :186
    nh.status = NS_VISITED;
  }
  EXP_PHASE(7);   // edge records to HBM
#ifdef ELF_PROFILE_EXPAND
  if (lane == 0) { unsigned long long t = 0; for (int k = 0; k < 8; ++k) t += ph_acc[k]; if (t > g_expand_rowmax[blockIdx.x & 65535]) g_expand_rowmax[blockIdx.x & 65535] = t; }
  if (lane < 8) g_expand_phase[blockIdx.x & 65535][lane] += lane == 0 ? ph_acc[0] : lane == 1 ? ph_acc[1] : lane == 2 ? ph_acc[2] : lane == 3 ? ph_acc[3] : lane == 4 ? ph_acc[4] : lane == 5 ? ph_acc[5] : lane == 6 ? ph_acc[6] : ph_acc[7];
#endif
}

// ------------------------------------------------------------------------------------------------
// backup: batch_rollouts :245-259, one wave per game, one LANE per unique leaf (64 leaves at a time).
// The trajectories are walked level by level from the deepest leaf's level up to the root's children, every lane at the
// ancestor of its leaf on that level (the leaf's depth comes from select, so all lanes sit on the same tree level at the same
// time).  Lanes whose trajectories have merged (same node) form a group; the edge above the group receives the group's rewards
// in leaf order -- the order of the serial loop, which for fp32 sums is the result -- by every member running the same
// readlane chain redundantly; the lowest lane of the group stores.  Visit counts are integers (order-free): one atomic add
// per group on the parent node.
// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_mcts_backup(TreePool<N> tp, TreeCfg cfg) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const GameNodes<N> nodes(tp, g);
  const int nu = rfl(tp.gs[g].n_unique);
  if (nu == 0) return;
  const LeafRec* leaves = tp.leaves + (size_t)g * MCTS_KMAX;
  for (int i = lane; i < nu; i += 64) {      // pre_evaluate result -> setEvaluation with an empty pi
    const LeafRec lr = leaves[i];
    if (lr.kind == LK_TERMINAL) {
      const NodeRef<N> leaf = nodes[lr.node];
      leaf.h().V = lr.value;
      leaf.h().flip = leaf.board().h.next_player == S_WHITE;
      leaf.h().n_edges = 0;
      leaf.h().status = NS_VISITED;
    }
  }
  mem_sync();                                // a terminal leaf may be another search thread's revisited leaf in this step
  // more than 64 unique leaves (num_threads x num_rollouts_per_batch up to MCTS_KMAX): chunks of 64 in leaf order, one after the
  // other -- an edge then receives chunk 0's rewards in leaf order, then chunk 1's: the serial order
  for (int c0 = 0; c0 < nu; c0 += 64) {
    const bool have = c0 + lane < nu;
    int c = 0, d = 0, count = 0;
    if (have) {
      const LeafRec lr = leaves[c0 + lane];
      c = lr.node; d = lr.depth; count = lr.count;
    }
    const float reward = have ? nodes[c].h().V : 0.0f;   // MCTSActor::reward (go/mcts/mcts.h:163-165)
    const float vsub = (float)(cfg.virtual_loss * count);
    const int maxd = (int)wave_max_u32((u32)d);
    for (int lvl = maxd; lvl >= 1; --lvl) {    // updateEdgeStats :253-278 along the trajectories
      const bool act = have && d >= lvl;
      const u64 am = __ballot(act);
      int p = 0;
      TStat* sp = nullptr;
      TStat s = tst_none();
      if (act) {
        const NodeHdr& ch = nodes[c].h();
        p = ch.parent;
        sp = &nodes[p].tst()[ch.parent_edge];
        s = *sp;
      }
      bool leader = true;
      int zc = 0;
      u64 m = am;
      while (m) {
        const int jl = (int)__builtin_ctzll(m);
        m &= m - 1;
        const int cj = rl(c, jl);
        const float rj = rlf(reward, jl), vj = rlf(vsub, jl);
        if (c == cj) {
          if (jl < lane) leader = false;
          s.reward = __fadd_rn(s.reward, rj);
          s.vloss = __fsub_rn(s.vloss, vj);
          ++zc;
        }
      }
      if (act && leader) {
        s.visits += zc;
        *sp = s;
        atomicAdd(&nodes[p].h().num_visits, zc);
      }
      if (act) c = p;
    }
    if (c0 + 64 < nu) mem_sync();            // the next chunk reads the edge sums this one stored
  }
}

// ------------------------------------------------------------------------------------------------
// root services
// ------------------------------------------------------------------------------------------------
// NodeT::enhanceExploration :132-155; etas/Z drawn on the host with libstdc++'s gamma_distribution
template <int N>
__global__ __launch_bounds__(64) void k_mcts_dirichlet(TreePool<N> tp, const float* etas, const float* Z, float epsilon) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (rfl(tp.game_mode(g)) == GM_IDLE) return;
  const NodeRef<N> r = GameNodes<N>(tp, g)[rfl(tp.gs[g].root)];
  const int n = rfl(r.h().n_edges);
  if (rfl(r.h().status) != NS_VISITED) return;
  float* const e_prior = r.prior();
  u16* const e_coord = r.coord();
  u16* const e_orig = r.orig();
  const float z = Z[g], ome = __fsub_rn(1.0f, epsilon);
  for (int i = lane; i < n; i += 64) {
    const float p = e_prior[i];
    const int o = e_orig[i];   // eta_o belongs to the o-th edge of the map's iteration order
    // (1 - epsilon) * p + epsilon * etas[o] / Z, left to right, no contraction
    e_prior[i] = __fadd_rn(__fmul_rn(ome, p), __fdiv_rn(__fmul_rn(epsilon, etas[(size_t)g * NodeL<N>::NE + o]), z));
  }
  // the never-followed part of the scoring order follows the NEW priors: descending prior (ties: ascending orig).  Those
  // entries carry no statistics and no child: sort (prior, position) keys, then move prior / coord / orig accordingly.
  mem_sync();
  const int nt = rfl(r.h().n_touched);
  u64 sx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int pos = nt + k * 64 + lane;
    sx[k] = ~0ull;
    if (pos < n) sx[k] = ((u64)(~f2ukey(e_prior[pos])) << 32) | ((u32)e_orig[pos] << 16) | (u32)(pos - nt);
  }
  bitonic_sort512(sx, lane);
  float np[8];
  u16 nc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = k * 64 + lane;
    np[k] = 0.0f; nc[k] = 0;
    if (nt + i < n) {
      const int src = nt + (int)(sx[k] & 0xFFFFu);
      np[k] = e_prior[src];
      nc[k] = e_coord[src];
    }
  }
  mem_sync();   // every source entry has been read before any destination is written
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = k * 64 + lane;
    if (nt + i < n) {
      e_prior[nt + i] = np[k];
      e_coord[nt + i] = nc[k];
      e_orig[nt + i] = (u16)((sx[k] >> 16) & 0xFFFFu);
    }
  }
}

struct RootInfo {   // 32 B per game
  int n_edges, num_visits, status, root;
  float V;
  int rng_pos, err, live;      // live: node ids the game's tree holds
};

// root edges in iteration order (MCTSResultT::addActions walks exactly this, tree_search_base.h:248-292)
template <int N>
__global__ __launch_bounds__(64) void k_mcts_root(TreePool<N> tp, RootInfo* info, int32_t* coord, int32_t* visits, float* prior,
                                                   float* reward, int32_t* child) {
  constexpr int NE = NodeL<N>::NE;
  const int g = blockIdx.x, lane = threadIdx.x;
  const GameState& gs = tp.gs[g];
  const NodeRef<N> r = GameNodes<N>(tp, g)[rfl(gs.root)];
  const int n = rfl(r.h().n_edges), nt = rfl(r.h().n_touched);
  if (lane == 0) {
    RootInfo ri;
    ri.n_edges = n; ri.num_visits = r.h().num_visits; ri.status = r.h().status; ri.root = gs.root; ri.V = r.h().V;
    ri.rng_pos = gs.rng_pos; ri.err = gs.err | tp.tops->err; ri.live = gs.live;
    info[g] = ri;
  }
  // the arrays are in scoring order; the caller gets them in the map's iteration order (entry -> slot orig)
  for (int i = n + lane; i < NE; i += 64) {
    const size_t o = (size_t)g * NE + i;
    if (coord) coord[o] = -1;
    if (visits) visits[o] = 0;
    if (prior) prior[o] = 0.f;
    if (reward) reward[o] = 0.f;
    if (child) child[o] = -1;
  }
  for (int i = lane; i < n; i += 64) {
    const size_t o = (size_t)g * NE + r.orig()[i];
    TStat t = tst_none();
    if (i < nt) t = r.tst()[i];
    if (coord) coord[o] = r.coord()[i];
    if (visits) visits[o] = t.visits;
    if (prior) prior[o] = r.prior()[i];
    if (reward) reward[o] = t.reward;
    if (child) child[o] = t.child;
  }
}

// Invariants of the node records (test / debug service, elfmcts_validate): for every live node that has been expanded
//   entries [0, n_touched): child >= 0, the child's header points back (parent, parent_edge == position), orig ascending;
//                           n_touched fits the record's class (16 in a small record)
//   entries [n_touched, n_edges): priors descending
//   every live node that owns a state carries its position's hash in the header
// out[0] = number of violations, out[1..4] = code, game, node, position of one of them.
template <int N>
__global__ __launch_bounds__(64) void k_mcts_validate(TreePool<N> tp, int32_t* out) {
  const int lane = threadIdx.x;
  const GameNodes<N> nodes(tp, 0);
  const int* po = tp.parent_of;
  for (int id = blockIdx.x; id < tp.C; id += gridDim.x) {
    if (po[id] == -2) continue;
    const int g = tp.owner[id];
    auto fail = [&](int code, int node, int pos) {
      if (atomicAdd(&out[0], 1) == 0) { out[1] = code; out[2] = g; out[3] = node; out[4] = pos; }
    };
    if (g < 0 || g >= tp.G) { if (lane == 0) fail(13, id, g); continue; }
    const NodeRef<N> r = nodes[id];
    const NodeHdr& rh = r.h();
    if (lane == 0 && rh.parent != po[id]) fail(11, id, po[id]);
    if (lane == 0 && rh.parent >= 0 && tp.owner[rh.parent] != g) fail(14, id, rh.parent);
    if (lane == 0 && rh.parent == -1 && tp.gs[g].root != id) fail(15, id, tp.gs[g].root);
    if (lane == 0 && rh.has_state && (((u64)rh.hash_hi << 32) | rh.hash_lo) != r.board().h.hash) fail(12, id, 0);
    if (rh.status != NS_VISITED) continue;
    const int n = rh.n_edges, nt = rh.n_touched;
    if (nt < 0 || nt > n || nt > nodes.cap(id)) { if (lane == 0) fail(1, id, nt); continue; }
    int visits = 0;
    for (int i = lane; i < n; i += 64) {
      if (i < nt) {
        const TStat t = r.tst()[i];
        const int ch = t.child;
        visits += t.visits;
        if (ch < 0 || ch >= tp.C) { fail(2, id, i); continue; }
        if (nodes[ch].h().parent != id || po[ch] != id) fail(3, id, i);
        if (nodes[ch].h().parent_edge != i) fail(4, id, i);
        if (i + 1 < nt && r.orig()[i] >= r.orig()[i + 1]) fail(5, id, i);
      } else {
        if (i + 1 < n && !(r.prior()[i] >= r.prior()[i + 1])) fail(8, id, i);
      }
      if (r.orig()[i] >= n) fail(9, id, i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) visits += __shfl_xor(visits, o, 64);
    if (lane == 0 && visits != rh.num_visits) fail(10, id, visits);
  }
}

// per-game node counts by a scan of the owner array (statistics / tests): out[g] = ids whose owner is g
template <int N>
__global__ __launch_bounds__(256) void k_mcts_count_live(TreePool<N> tp, int32_t* out) {
  for (int id = blockIdx.x * 256 + threadIdx.x; id < tp.C; id += gridDim.x * 256)
    if (tp.parent_of[id] != -2) atomicAdd(&out[tp.owner[id]], 1);
}

// SearchTreeT::treeAdvance :420-436: the child reached by `move` becomes the root, everything else is freed.
// Four launches over the context's dense id arrays (parent 4 B + owner 4 B per node id: coalesced):
//   k_mcts_advance_prepare  one wave per game: the id of the next root (or "none" / "this game does not move")
//   k_mcts_advance_mark     all ids: keep[id] = the parent chain of id reaches its game's next root; kept nodes counted per game
//   k_mcts_advance_sweep    all ids: dead ids go back on the context's free stacks (one atomic per wave; the ORDER of the ids on
//                           the stack depends on the waves' timing -- ids are internal, no result depends on them)
//   k_mcts_advance_reroot   one wave per game: re-root, or a fresh root for a game whose move had no child
// SearchTreeT::clear is the same sweep with "free everything" for the listed games.
template <int N>
__device__ __forceinline__ int advance_next_root(const NodeRef<N> r, int mv, int lane) {
  const int n = rfl(r.h().n_edges), nt = rfl(r.h().n_touched);
  int next_root = -1;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const u64 b = __ballot(i < n && r.coord()[i] == mv);
    if (b) {
      const int pos = base + (int)__builtin_ctzll(b);
      next_root = pos < nt ? rfl(r.tst()[pos].child) : -1;     // a never-followed edge has no child node
    }
  }
  return next_root;
}

// one wave per game: adv[g] = -2 (no move for this game: elfsp_play with a partial move list), the id of the next root, or -1
template <int N>
__global__ __launch_bounds__(64) void k_mcts_advance_prepare(TreePool<N> tp, const int32_t* moves) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int mv = rfl(moves[g]);
  int a = -2;
  if (mv >= 0) a = advance_next_root<N>(GameNodes<N>(tp, g)[rfl(tp.gs[g].root)], mv, lane);
  if (lane == 0) { tp.adv[g] = a; tp.live_tmp[g] = 0; }
}

// every id of the context: keep[id] = its game moves and its parent chain reaches the game's next root
template <int N>
__global__ __launch_bounds__(256) void k_mcts_advance_mark(TreePool<N> tp) {
  const int* po = tp.parent_of;
  for (int id = blockIdx.x * 256 + threadIdx.x; id < tp.C; id += gridDim.x * 256) {
    int k = 0;
    if (po[id] != -2) {
      const int g = tp.owner[id];
      const int next_root = tp.adv[g];
      if (next_root >= 0) {
        int a = id;
        for (;;) {
          if (a == next_root) { k = 1; break; }
          a = po[a];
          if (a < 0) break;
        }
        if (k) atomicAdd(&tp.live_tmp[g], 1);
      } else if (next_root == -2) {
        k = 1;                               // the game does not move: nothing of it is freed
      }
    }
    tp.keep[id] = (unsigned char)k;
  }
}

// dead ids go back on the context's free stacks: one atomic per wave and class
template <int N>
__global__ __launch_bounds__(256) void k_mcts_advance_sweep(TreePool<N> tp) {
  const int lane = threadIdx.x & 63;
  int* po = tp.parent_of;
  const int rounds = (tp.C + 63) >> 6;       // Cs is a multiple of 64: a round is all small ids or all big ids
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rounds; r += gridDim.x * 4) {
    const int id = r * 64 + lane;
    const bool dead = id < tp.C && po[id] != -2 && !tp.keep[id];
    const u64 b = __ballot(dead);
    if (b == 0) continue;
    const int rank = __popcll(b & ((1ull << lane) - 1));
    const bool sm = r * 64 < tp.Cs;
    int base = 0;
    if (lane == 0) base = atomicAdd(sm ? &tp.tops->small : &tp.tops->big, __popcll(b));
    base = rfl(base);
    if (dead) {
      po[id] = -2;
      if (sm) tp.gstack[base + rank] = id; else tp.gstack_big[base + rank] = id - tp.Cs;
    }
  }
}

// one wave per game that moved: the child reached by the move becomes the root, or (no such child / clear) a fresh root
template <int N>
__global__ __launch_bounds__(64) void k_mcts_advance_reroot(TreePool<N> tp) {
  const int g = blockIdx.x, lane = threadIdx.x;
  int next_root = rfl(tp.adv[g]);
  if (next_root == -2) return;
  const GameNodes<N> nodes(tp, g);
  GameState& gs = tp.gs[g];
  int free_top = rfl(gs.free_top), live = rfl(tp.live_tmp[g]);
  if (next_root < 0) {                       // allocateRoot -> addNode(0.0)
    if (free_top <= 0) { free_top = stash_refill(tp, g, 0, 1, lane); mem_sync(); }
    if (free_top <= 0) { if (lane == 0) gs.err |= MCTS_ERR_POOL; return; }
    next_root = rfl(tp.free_stack[(size_t)g * tp.SC + free_top - 1]);
    --free_top;
    node_init(tp, g, next_root, -1, -1, 0.0f, lane);
    live = 1;
  } else if (lane == 0) {
    nodes[next_root].h().parent = -1;
    nodes[next_root].h().parent_edge = -1;
    tp.parent_of[next_root] = -1;
  }
  if (lane == 0) { gs.root = next_root; gs.free_top = free_top; gs.live = live; if (live > gs.live_peak) gs.live_peak = live; tp.adv[g] = -2; }
}

}  // namespace elfgo
