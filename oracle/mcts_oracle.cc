// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// CPU restatement of the search half of the self-play hot path, serial and as plain as possible,
// over the C restatement of the board engine (go_oracle.c).  Every function cites the reference lines it follows
// (paths relative to /root/reference/src_cpp).  It is built into oracle/libgo_oracle{19,9}.so from source that travels with the
// repository, so it exists wherever g++ does -- also where neither /root/reference nor a prebuilt oracle/_ref is present.
//
// Pinned: tests/test_oracle_mcts.py runs it against every MCTS golden fixture that the REAL reference stack produced
// (tests/golden/mcts_*.npz, records_*.npz: root edges in iteration order, priors, visit counts, rewards, moves), bit for bit.
//
// What makes the result implementation-defined in the reference is taken from the same libstdc++ here, not re-derived:
// std::unordered_map<unsigned short, ...> iteration order, std::sort tie order, std::gamma_distribution,
// std::uniform_real_distribution, std::mt19937.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <ctime>
#include <random>
#include <unordered_map>
#include <utility>
#include <vector>

extern "C" {
typedef struct OrcState OrcState;
OrcState* orc_new(void);
OrcState* orc_clone(const OrcState*);
void orc_free(OrcState*);
void orc_reset(OrcState*);
int orc_board_size(void);
int orc_forward(OrcState*, int c);
int orc_check_move(OrcState*, int c);
int orc_terminated(const OrcState*);
int orc_ply(const OrcState*);
int orc_next_player(const OrcState*);
int orc_last_move(const OrcState*);
uint64_t orc_hash(const OrcState*);
float orc_evaluate(const OrcState*, float komi);
void orc_extract_agz(const OrcState*, int d4, float* out);
void orc_stub_net(const float* s, int batch, uint32_t salt, int tie_levels, float* pi, float* v);
}

namespace {

typedef unsigned short Coord;   // elfgames/go/base/common.h:20
enum { M_PASS = 0, M_INVALID = 3, S_BLACK = 1, S_WHITE = 2 };

// same layout as RefSpConfig / RefSpSearch in oracle/ref_selfplay.cc (one ctypes definition serves both)
struct SpConfig {
  int32_t num_games, batchsize, mcts_threads, rollouts_per_thread, rollouts_per_batch, virtual_loss, persistent_tree, use_prior;
  int32_t unexplored_q_zero, root_unexplored_q_zero;
  float c_puct, root_epsilon, root_alpha;
  uint32_t seed;
  float komi;
  int32_t ply_pass_enabled, policy_distri_cutoff, move_cutoff;
  float resign_thres, never_resign_prob;
  uint32_t net_salt;
  int32_t net_tie_levels, max_searches, timeout_usec;
  // round 3: evaluation games, pick methods, policy-only play (same layout as RefSpConfig's tail)
  int32_t black_ver, white_ver, player_swap;
  float white_puct;
  int32_t white_rollouts_per_batch, white_rollouts_per_thread;
  uint32_t white_net_salt;
  int32_t pick_method, black_policy_only, white_policy_only, thread_used;
  int32_t req2_after_searches, req2_black_ver, req2_async;   // a second request mid-run: see orcsp_run
  int32_t cheat_eval_new_model_wins_half, cheat_selfplay_random_result;
  int32_t online, following_pass, net_value_on;   // online mode: oracle/ref_selfplay.cc only
  float net_value;
  int32_t req2_ts, req2_rollouts_per_thread, req2_rollouts_per_batch;   // the second request's own TSOptions and white_ver
  float req2_c_puct, req2_root_epsilon, req2_root_alpha;
  int32_t req2_unexplored_q_zero, req2_root_unexplored_q_zero, req2_white_ver;
};
struct SpSearch {
  int32_t game, move_played, best_action, total_visits, n_edges;
  float root_value, max_score;
  int32_t pad;
};

// elf/ai/tree_search/tree_search_base.h:102-157 EdgeInfo
struct Edge {
  float prior;
  int child = -1;
  float reward = 0;
  int num_visits = 0;
  float virtual_loss = 0;
  explicit Edge(float p) : prior(p) {}
};

enum { NOT_VISITED = 0, EVAL_REQUESTED = 1, VISITED = 2 };

// elf/ai/tree_search/tree_search_node.h:84-399 NodeT (the state is owned by the node, :62-82)
struct Node {
  int status = NOT_VISITED;
  std::unordered_map<Coord, Edge> sa;   // stateActions_ :310
  int num_visits = 0;
  float V = 0, unsigned_mean_q, unsigned_parent_q;
  bool flip = false;
  OrcState* state = nullptr;
  explicit Node(float upq) : unsigned_mean_q(upq), unsigned_parent_q(upq) {}   // :97-102
  ~Node() { if (state) orc_free(state); }
};

// tree_search_node.h:401-562 SearchTreeT
struct Tree {
  std::unordered_map<int, std::unique_ptr<Node>> nodes;
  int count = 0, root = -1;
  Tree() { clear(); }
  int add_node(float upq) { nodes[count].reset(new Node(upq)); return count++; }          // :447-451
  Node* at(int id) { auto it = nodes.find(id); return it == nodes.end() ? nullptr : it->second.get(); }
  void clear() { nodes.clear(); count = 0; root = -1; root = add_node(0.0f); }             // :411-416, allocateRoot :555-561
  void recursive_free(int id) {                                                            // :457-467
    if (id < 0) return;
    Node* n = at(id);
    for (const auto& p : n->sa) recursive_free(p.second.child);
    nodes.erase(id);
  }
  void advance(Coord action) {                                                             // treeAdvance :420-436
    int next_root = -1;
    Node* r = at(root);
    for (const auto& p : r->sa) {
      if (p.first == action) next_root = p.second.child;
      else recursive_free(p.second.child);
    }
    nodes.erase(root);
    root = next_root;
    if (root < 0) root = add_node(0.0f);
  }
};

struct Response {   // NodeResponseT, tree_search_base.h:33-38
  std::vector<std::pair<Coord, float>> pi;
  float value = 0;
  bool q_flip = false;
};

typedef void (*net_fn)(const float* s, int batch, float* pi, float* v, void* user);

struct Actor {   // elfgames/go/mcts/mcts.h:40-333 MCTSActor
  int n, na;
  float komi;
  int ply_pass_enabled;
  std::mt19937 rng;
  const SpConfig* cfg;
  net_fn net;
  void* user;
  uint32_t salt = 0;          // stub net of this actor's batch group ("actor_black" / "actor_white")
  int64_t rows = 0, batches = 0;

  // BoardFeature::action2Coord (base/board_feature.h:139-144) with InvTransform (:115-130)
  Coord action2coord(int a, int d4) const {
    if (a == n * n) return M_PASS;
    int x = a / n, y = a % n;
    if ((d4 >> 2) == 1) std::swap(x, y);
    const int rot = d4 % 4;
    int ox = x, oy = y;
    if (rot == 1) { ox = n - y - 1; oy = x; }
    else if (rot == 2) { ox = n - x - 1; oy = n - y - 1; }
    else if (rot == 3) { ox = y; oy = n - x - 1; }
    return (Coord)((oy + 1) * (n + 2) + (ox + 1));
  }

  // pi2response :256-332 + normalize :244-254
  void pi2response(const OrcState* s, int d4, const float* pi, bool pass_enabled, std::vector<std::pair<Coord, float>>* out) {
    out->clear();
    if (orc_terminated(s)) return;
    for (int i = 0; i < na; ++i) out->push_back(std::make_pair(action2coord(i, d4), pi[i]));
    typedef std::pair<Coord, float> D;
    std::sort(out->begin(), out->end(), [](const D& a, const D& b) { return a.second > b.second; });
    std::vector<D> tmp;
    for (const D& v : *out) {
      const bool valid = (v.first == M_PASS && pass_enabled) || (v.first != M_PASS && orc_check_move(const_cast<OrcState*>(s), v.first));
      if (valid) tmp.push_back(v);
    }
    if (tmp.empty() && !pass_enabled) tmp.push_back(std::make_pair((Coord)M_PASS, 1.0f));
    *out = tmp;
    float total = 1e-10;
    for (const D& p : *out) total += p.second;
    for (D& p : *out) p.second /= total;
  }

  // evaluate :73-121 (batch): pre_evaluate :185-207, get_extractor :175-183, post_nn_result :209-230
  void evaluate(const std::vector<const OrcState*>& states, std::vector<Response>* resps) {
    if (states.empty()) return;
    resps->assign(states.size(), Response());
    std::vector<size_t> sel;
    std::vector<int> d4s;
    for (size_t i = 0; i < states.size(); ++i) {
      Response& r = (*resps)[i];
      const OrcState* s = states[i];
      r.q_flip = orc_next_player(s) == S_WHITE;
      if (orc_terminated(s)) {
        r.value = orc_evaluate(s, komi) > 0 ? 1.0f : -1.0f;
        r.pi.clear();
      } else {
        sel.push_back(i);
        d4s.push_back((int)(rng() % 8));   // BoardFeature::RandomShuffle, base/board_feature.h:74-78
      }
    }
    if (sel.empty()) return;
    const int b = (int)sel.size(), fs = 18 * n * n;
    std::vector<float> feat((size_t)b * fs), pi((size_t)b * na), v(b);
    for (int j = 0; j < b; ++j) orc_extract_agz(states[sel[j]], d4s[j], &feat[(size_t)j * fs]);
    if (net) net(feat.data(), b, pi.data(), v.data(), user);
    else orc_stub_net(feat.data(), b, salt, cfg->net_tie_levels, pi.data(), v.data());
    rows += b; batches++;
    for (int j = 0; j < b; ++j) {
      Response& r = (*resps)[sel[j]];
      const OrcState* s = states[sel[j]];
      r.value = v[j];
      bool pass_enabled = orc_ply(s) >= ply_pass_enabled;
      if (pass_enabled && orc_last_move(s) != M_PASS) {   // remove_pass_if_dangerous :232-242 (params default true)
        const bool black_win = orc_evaluate(s, komi) > 0;
        if ((black_win && orc_next_player(s) == S_WHITE) || (!black_win && orc_next_player(s) == S_BLACK)) pass_enabled = false;
      }
      pi2response(s, d4s[j], &pi[(size_t)j * na], pass_enabled, &r.pi);
    }
  }
};

struct Search {   // elf/ai/tree_search/tree_search.h TreeSearchSingleThreadT + TreeSearchT, one thread
  Tree tree;
  const SpConfig* cfg;
  Actor* actor;
  // TSOptions of this AI after init_ai's overrides (game_selfplay.cc:51-70)
  float c_puct = 0;
  int rollouts_per_batch = 0, rollouts_per_thread = 0;
  size_t next_move_number = 0;   // MCTSAI_T::nextMoveNumber_
  // the generators of the search threads 1 .. T-1's own MCTSActors (TreeSearchT's actor_gen makes one actor per thread, every one
  // seeded with the SAME params.seed: game_selfplay.cc:45-47,77); thread 0's is actor->rng, which also feeds the Dirichlet draws
  std::vector<std::mt19937> thread_rng;

  // EdgeInfo::getScore (tree_search_base.h:132-157) + NodeT::UCT :361-397 + BestAction :321-358 + findMove :205-231
  bool find_move(Node* nd, int depth, Coord* action) {
    if (nd->status != VISITED) return false;
    if (nd->sa.empty()) return false;
    if (cfg->unexplored_q_zero || (cfg->root_unexplored_q_zero && depth == 0)) nd->unsigned_mean_q = 0.0;
    Coord best = M_INVALID;
    float max_score = std::numeric_limits<float>::lowest(), total_unsigned_q = 0;
    int total_visits = 0;
    for (const auto& ap : nd->sa) {
      const Edge& e = ap.second;
      const int all_visits = nd->num_visits + 1;
      float r = e.reward;
      if (nd->flip) r = -r;
      r -= e.virtual_loss;
      const int nvl = e.num_visits + e.virtual_loss;   // int + float -> float -> int
      const float q = nvl > 0 ? r / nvl : (nd->flip ? -nd->unsigned_mean_q : nd->unsigned_mean_q);
      const float unsigned_q = e.num_visits > 0 ? e.reward / e.num_visits : nd->unsigned_mean_q;
      const float prior = e.prior / (1 + e.num_visits) * std::sqrt(all_visits);   // float * double sqrt(int) -> float
      const bool first_visit = nvl == 0;
      const float score = cfg->use_prior ? (prior * c_puct + q) : q;
      if (score > max_score) { max_score = score; best = ap.first; }
      if (!first_visit) { total_unsigned_q += unsigned_q; total_visits++; }
    }
    *action = best;
    nd->unsigned_mean_q = (nd->unsigned_parent_q + total_unsigned_q) / (total_visits + 1);
    return true;
  }

  struct Traj {
    std::vector<std::pair<Node*, Coord>> traj;
    Node* leaf;
  };

  // single_rollout :264-322
  Traj single_rollout() {
    Node* node = tree.at(tree.root);
    Traj t;
    int depth = 0;
    while (node->status == VISITED) {
      Coord action;
      if (!find_move(node, depth, &action)) break;
      if (cfg->virtual_loss > 0) node->sa.find(action)->second.virtual_loss += cfg->virtual_loss;   // addVirtualLoss :233-251
      t.traj.push_back(std::make_pair(node, action));
      Edge& e = node->sa.find(action)->second;                                                      // followEdge :280-302
      if (e.child < 0) e.child = tree.add_node(node->unsigned_mean_q);
      Node* next = tree.at(e.child);
      if (next == nullptr) break;
      if (next->state == nullptr) {                                                                 // allocateState :174-190
        OrcState* st = orc_clone(node->state);
        if (!orc_forward(st, action)) { orc_free(st); break; }
        next->state = st;
      }
      node = next;
      ++depth;
    }
    t.leaf = node;
    return t;
  }

  // batch_rollouts :200-262.  The reference walks `traj_counts` (keyed by node address) in hash order; this restatement backs
  // the unique leaves up in first-occurrence order (with values on a 1/256 grid the sums are order-independent).
  // mcts_threads = T > 1 (TreeSearchT's thread pool, tree_search.h:345-368) is restated as ONE of the interleavings the
  // reference's racing threads can produce: the T batch_rollouts calls of a round run their descents back to back (thread t
  // sees the virtual losses of threads < t; a leaf another thread has already requested is not requested again, :142-153),
  // then every thread evaluates its own locked leaves with ITS OWN actor (own mt19937 for the D4 draws, all seeded alike:
  // game_selfplay.cc:45-47,77), then every thread sets its evaluations and backs up its own trajectories (waitEvaluation :250).
  // Round 5: this is exactly the schedule the turnstile build of the real reference runs (oracle/Makefile, libelfsp*_ts.so;
  // ref_selfplay.cc elf_ts_hook): the fixtures mcts_*_T2 / _T4 are that reference's output.
  struct Batch {
    std::vector<Traj> trajs;
    std::vector<Node*> locked;
    std::vector<const OrcState*> states;
    std::vector<std::pair<Node*, std::pair<Traj*, int>>> counts;
  };
  void descend(Batch& b) {
    b.trajs.reserve(rollouts_per_batch);
    for (int j = 0; j < rollouts_per_batch; ++j) b.trajs.push_back(single_rollout());
    for (Traj& t : b.trajs) {
      if (t.leaf->status == NOT_VISITED) {   // requestEvaluation :142-153
        t.leaf->status = EVAL_REQUESTED;
        b.locked.push_back(t.leaf);
        b.states.push_back(t.leaf->state);
      }
      bool found = false;
      for (auto& c : b.counts) if (c.first == t.leaf) { c.second.second++; found = true; break; }
      if (!found) b.counts.push_back(std::make_pair(t.leaf, std::make_pair(&t, 1)));
    }
  }
  void evaluate(Batch& b) {
    std::vector<Response> resps;
    actor->evaluate(b.states, &resps);
    for (size_t j = 0; j < b.locked.size(); ++j) {   // setEvaluation :176-203
      Node* nd = b.locked[j];
      for (const auto& ap : resps[j].pi) nd->sa.insert(std::make_pair(ap.first, Edge(ap.second)));
      nd->V = resps[j].value;
      nd->flip = resps[j].q_flip;
      nd->status = VISITED;
    }
  }
  void backup(Batch& b) {
    for (auto& c : b.counts) {
      const float reward = c.first->V;   // MCTSActor::reward :163-165
      for (const auto& p : c.second.first->traj) {   // updateEdgeStats :253-278
        Edge& e = p.first->sa.find(p.second)->second;
        p.first->num_visits++;
        e.reward += reward;
        e.num_visits++;
        e.virtual_loss -= (float)(cfg->virtual_loss * c.second.second);
      }
    }
  }
  void batch_rollouts() {
    const int T = cfg->mcts_threads > 1 ? cfg->mcts_threads : 1;
    std::vector<Batch> bs(T);
    for (int t = 0; t < T; ++t) descend(bs[t]);
    for (int t = 0; t < T; ++t) {
      if (t > 0) std::swap(actor->rng, thread_rng[t - 1]);   // thread t's MCTSActor draws from its own generator
      evaluate(bs[t]);
      if (t > 0) std::swap(actor->rng, thread_rng[t - 1]);
    }
    for (int t = 0; t < T; ++t) backup(bs[t]);
  }

  // TreeSearchT::run :410-426 (setRootNodeState :478-493, enhanceExploration tree_search_node.h:132-155, chooseAction :495-528)
  bool run(const OrcState* root_state) {
    Node* root = tree.at(tree.root);
    if (root->state == nullptr) root->state = orc_clone(root_state);
    if (orc_hash(root_state) != orc_hash(root->state)) return false;   // "Root state is not the same as the input state"
    if (cfg->root_epsilon > 0.0f) {
      std::gamma_distribution<> dis(cfg->root_alpha);
      std::vector<float> etas(root->sa.size());
      float Z = 1e-10;
      for (size_t i = 0; i < root->sa.size(); ++i) { etas[i] = dis(actor->rng); Z += etas[i]; }
      int i = 0;
      for (auto& p : root->sa) {
        p.second.prior = (1 - cfg->root_epsilon) * p.second.prior + cfg->root_epsilon * etas[i] / Z;
        i++;
      }
    }
    for (int idx = 0; idx < rollouts_per_thread; idx += rollouts_per_batch) batch_rollouts();   // tree_search.h:112-117
    return true;
  }

  // TreeSearchT::runPolicyOnly :385-407: the root is evaluated if it has not been yet; no noise, no rollouts
  bool run_policy_only(const OrcState* root_state) {
    Node* root = tree.at(tree.root);
    if (root->state == nullptr) root->state = orc_clone(root_state);
    if (orc_hash(root_state) != orc_hash(root->state)) return false;
    if (root->status != VISITED) {
      std::vector<const OrcState*> one(1, root->state);
      std::vector<Response> resps;
      actor->evaluate(one, &resps);
      for (const auto& ap : resps[0].pi) root->sa.insert(std::make_pair(ap.first, Edge(ap.second)));
      root->V = resps[0].value;
      root->flip = resps[0].q_flip;
      root->status = VISITED;
    }
    return true;
  }

  // MCTSAI_T::align_state (elf/ai/tree_search/mcts.h:141-167)
  void align_state(const std::vector<Coord>& moves) {
    if (!cfg->persistent_tree) { tree.clear(); next_move_number = 0; }
    else if (next_move_number > moves.size()) { tree.clear(); next_move_number = 0; }   // moves_since fails -> resetTree
    else {
      for (size_t i = next_move_number; i < moves.size(); ++i) tree.advance(moves[i]);
      next_move_number = moves.size();
    }
  }
  void end_game() { tree.clear(); next_move_number = 0; }   // MCTSAI_T::endGame -> resetTree
};

std::vector<Coord> g_preload;   // GameOptions.preload_sgf as Coords for the following orcsp_run calls
int g_preload_move_to = -1;

}  // namespace

extern "C" {

// GameOptions.preload_sgf / preload_sgf_move_to (game_selfplay.cc:202-219); n = 0 switches it off
void orcsp_set_preload(const uint16_t* moves, int n, int move_to) {
  g_preload.assign(moves, moves + (n > 0 ? n : 0));
  g_preload_move_to = move_to;
}

// Self-play of one game slot with one MCTS AI: GoGameSelfPlay::act (elfgames/go/common/game_selfplay.cc:272-430) with
// init_ai :30-78, mcts_make_diverse_move :80-95, mcts_update_info :97-119, finish_game :121-149, ResignCheck
// (common/game_utils.h:14-54), MCTSAI_T::act / align_state / advanceMoves (elf/ai/tree_search/mcts.h:59-81,141-167).
// Outputs as refsp_run of oracle/ref_selfplay.cc: one SpSearch per search + the root edges in iteration order.
static int64_t g_fixed_time = 0;
static bool g_reseed = true;
void orcsp_set_time(int64_t t) { g_fixed_time = t; g_reseed = true; }

int orcsp_run(const SpConfig* cfg_in, net_fn net, void* user, SpSearch* out_search, int32_t* out_coord, int32_t* out_visits,
              float* out_prior, float* out_reward, int64_t* stats) {
  SpConfig cur_cfg = *cfg_in;                    // the request the game plays under: a second request may replace its versions / TSOptions
  const SpConfig* cfg = &cur_cfg;
  const int n = orc_board_size(), na = n * n + 1, max_move = 2 * n * n;
  std::mt19937 game_rng;
  game_rng.seed(cfg->seed);                      // GoGameBase, common/game_base.h:32-38
  // GoGameSelfPlay::restart :158-200: "actor_black" first, then (white_ver >= 0) "actor_white" with the white_* overrides, each
  // seeded with the next draw of the game's generator (init_ai :45-47); player_swap exchanges the two
  bool two = cfg->white_ver >= 0;
  Actor actors[2];
  Search searches[2];
  int64_t batches_before = 0, rows_before = 0;
  auto init_ais = [&]() {
  for (int a = 0; a < (two ? 2 : 1); ++a) {
    batches_before += actors[a].batches; rows_before += actors[a].rows;
    actors[a] = Actor{};
    searches[a] = Search{};
    Actor& actor = actors[a];
    actor.n = n; actor.na = na; actor.komi = cfg->komi; actor.ply_pass_enabled = cfg->ply_pass_enabled;
    actor.cfg = cfg; actor.net = net; actor.user = user;
    actor.salt = a == 0 ? cfg->net_salt : cfg->white_net_salt;
    actor.rng.seed(game_rng());                  // params.seed = _rng() :47, MCTSActor::rng_(params.seed) mcts.h:52
    Search& se = searches[a];
    se.cfg = cfg; se.actor = &actor;
    se.thread_rng.assign(cfg->mcts_threads > 1 ? cfg->mcts_threads - 1 : 0, actor.rng);   // the same params.seed for every thread's actor
    se.c_puct = cfg->c_puct; se.rollouts_per_batch = cfg->rollouts_per_batch; se.rollouts_per_thread = cfg->rollouts_per_thread;
    if (a == 1) {
      if (cfg->white_puct > 0.0f) se.c_puct = cfg->white_puct;
      if (cfg->white_rollouts_per_batch > 0) se.rollouts_per_batch = cfg->white_rollouts_per_batch;
      if (cfg->white_rollouts_per_thread > 0) se.rollouts_per_thread = cfg->white_rollouts_per_thread;
    }
  }
  };
  init_ais();
  Search* ai = &searches[0];
  Search* ai2 = two ? &searches[1] : nullptr;
  if (two && cfg->player_swap) std::swap(ai, ai2);
  bool req2_pending = cfg->req2_after_searches > 0;
  // MCTSResultT::addActions' static generator (tree_search_base.h:238), uniform_random only: one per process, seeded with
  // time(NULL) at the first search; orcsp_set_time fixes that value (before the first run of the process, as for the reference)
  // (the restatement, unlike the reference, can be re-seeded within a process: orcsp_set_time starts the sequence again)
  static std::mt19937 pick_rng;
  if (g_reseed) { pick_rng.seed((unsigned)(g_fixed_time != 0 ? g_fixed_time : (int64_t)time(nullptr))); g_reseed = false; }
  OrcState* st = orc_new();
  std::vector<Coord> moves;                      // GoState::_moves
  bool never_resign = false, has_never = false;  // ResignCheck
  size_t sgf_iter = 0;                           // GoGameSelfPlay::restart :202-219: forward the first preload_sgf_move_to moves
  for (int i = 0; sgf_iter < g_preload.size() && i < g_preload_move_to; ++i, ++sgf_iter) {
    if (!orc_forward(st, g_preload[sgf_iter])) { orc_free(st); return -4; }   // "Preload sgf: move not valid!"
    moves.push_back(g_preload[sgf_iter]);
  }
  int k = 0;
  while (k < cfg->max_searches) {
    // GoGameSelfPlay::act :272-290: the mailbox is read at every fifth act (_online_counter % 5; here one act = one search).  The
    // second request of oracle/ref_selfplay.cc reaches the mailbox during search number req2_after_searches (0-based), so the
    // first look that finds it is the next multiple of five after that.  OnReceive :222-270: other versions and not async ->
    // restart() :159-220 (both AIs rebuilt with fresh seeds from the game's generator, the state reset, nothing recorded);
    // async -> the game goes on (the versions only show in the record)
    if (req2_pending && k > cfg->req2_after_searches && k % 5 == 0) {
      req2_pending = false;
      // ModelPair::operator== (record.h): versions and mcts_opt (TSOptions::operator==, every field)
      const bool same_ts = !cfg->req2_ts || (cfg->req2_rollouts_per_thread == cfg->rollouts_per_thread && cfg->req2_rollouts_per_batch == cfg->rollouts_per_batch &&
                                             cfg->req2_c_puct == cfg->c_puct && cfg->req2_root_epsilon == cfg->root_epsilon && cfg->req2_root_alpha == cfg->root_alpha &&
                                             (cfg->req2_unexplored_q_zero != 0) == (cfg->unexplored_q_zero != 0) &&
                                             (cfg->req2_root_unexplored_q_zero != 0) == (cfg->root_unexplored_q_zero != 0));
      if (!cfg->req2_async && (cfg->req2_black_ver != cfg->black_ver || cfg->req2_white_ver != cfg->white_ver || !same_ts)) {
        // restart() builds the AIs from the NEW request: its versions (a second AI if white_ver >= 0) and its TSOptions :166-180
        cur_cfg.black_ver = cfg_in->req2_black_ver; cur_cfg.white_ver = cfg_in->req2_white_ver;
        if (cfg_in->req2_ts) {
          cur_cfg.rollouts_per_thread = cfg_in->req2_rollouts_per_thread; cur_cfg.rollouts_per_batch = cfg_in->req2_rollouts_per_batch;
          cur_cfg.c_puct = cfg_in->req2_c_puct; cur_cfg.root_epsilon = cfg_in->req2_root_epsilon; cur_cfg.root_alpha = cfg_in->req2_root_alpha;
          cur_cfg.unexplored_q_zero = cfg_in->req2_unexplored_q_zero; cur_cfg.root_unexplored_q_zero = cfg_in->req2_root_unexplored_q_zero;
        }
        two = cur_cfg.white_ver >= 0;
        init_ais();
        ai = &searches[0];
        ai2 = two ? &searches[1] : nullptr;
        if (two && cfg->player_swap) std::swap(ai, ai2);
        orc_reset(st); moves.clear();
        never_resign = false; has_never = false;
        sgf_iter = 0;
        for (int i = 0; sgf_iter < g_preload.size() && i < g_preload_move_to; ++i, ++sgf_iter) {
          if (!orc_forward(st, g_preload[sgf_iter])) { orc_free(st); return -4; }
          moves.push_back(g_preload[sgf_iter]);
        }
      }
    }
    // GoGameSelfPlay::act :354-372: the AI of the colour to move; MCTSAI_T::act or actPolicyOnly (align_state first)
    const bool white_to_move = orc_next_player(st) == S_WHITE;
    Search& search = (ai2 != nullptr && white_to_move) ? *ai2 : *ai;
    const bool policy_only = white_to_move ? cfg->white_policy_only != 0 : cfg->black_policy_only != 0;
    search.align_state(moves);
    if (policy_only) { if (!search.run_policy_only(st)) { orc_free(st); return -2; } }
    else if (!search.run(st)) { orc_free(st); return -2; }
    // chooseAction :495-528 / runPolicyOnly :401-405 with MCTSResultT::addActions (tree_search_base.h:237-294)
    const int method = policy_only ? 1 : cfg->pick_method;   // 0 most_visited, 1 strongest_prior, 2 uniform_random
    Node* root = search.tree.at(search.tree.root);
    SpSearch& S = out_search[k];
    memset(&S, 0, sizeof(S));
    S.game = 0; S.best_action = M_INVALID; S.max_score = std::numeric_limits<float>::lowest(); S.root_value = root->V;
    std::vector<std::pair<Coord, float>> policy;
    const Edge* best_edge = nullptr;
    int i = 0;
    for (int j = 0; j < na; ++j) { out_coord[(size_t)k * na + j] = -1; out_visits[(size_t)k * na + j] = 0; out_prior[(size_t)k * na + j] = 0; out_reward[(size_t)k * na + j] = 0; }
    const int random_idx = (method == 2 && !root->sa.empty()) ? (int)(pick_rng() % root->sa.size()) : 0;
    for (const auto& ap : root->sa) {
      const float score = method == 0 ? (float)ap.second.num_visits : method == 1 ? ap.second.prior : 1.0f;
      policy.push_back(std::make_pair(ap.first, score));
      S.total_visits += ap.second.num_visits;
      if (method == 2) { if (i == random_idx) { S.max_score = score; S.best_action = ap.first; best_edge = &ap.second; } }
      else if (score > S.max_score) { S.max_score = score; S.best_action = ap.first; best_edge = &ap.second; }
      out_coord[(size_t)k * na + i] = ap.first; out_visits[(size_t)k * na + i] = ap.second.num_visits;
      out_prior[(size_t)k * na + i] = ap.second.prior; out_reward[(size_t)k * na + i] = ap.second.reward;
      ++i;
    }
    S.n_edges = i;
    Coord c = (Coord)S.best_action;
    // mcts_make_diverse_move: MCTSPolicy::normalize (t = 1) + sample_multinomial (elf/utils/utils.h:159-182)
    if (!policy_only && orc_ply(st) <= cfg->policy_distri_cutoff) {
      float exp_sum = 0;
      for (auto& e : policy) { const float v = std::pow(e.second, 1.0 / 1.0f); e.second = v; exp_sum += v; }
      for (auto& e : policy) e.second /= exp_sum;
      float Z = 0.0;
      for (const auto& e : policy) Z += e.second;
      std::uniform_real_distribution<> dis(0, Z);
      const float rd = dis(game_rng);
      std::vector<float> accu(policy.size() + 1);
      accu[0] = 0;
      size_t pick = policy.size() - 1;
      for (size_t t = 1; t < accu.size(); t++) {
        accu[t] = policy[t - 1].second + accu[t - 1];
        if (rd < accu[t]) { pick = t - 1; break; }
      }
      c = policy[pick].first;
    }
    // mcts_update_info: MCTSGoAI::getValue (go/mcts/mcts.h:358-365)
    const float predicted = (S.total_visits == 0 || best_edge == nullptr) ? root->V : best_edge->reward / best_edge->num_visits;
    S.move_played = c;
    ++k;
    // shouldResign (go_state_ext.h:207-214) -> ResignCheck::check
    const bool black = orc_next_player(st) == S_BLACK;
    const float value = black ? predicted : -predicted;
    if (!has_never) {
      std::uniform_real_distribution<> dis(0.0, 1.0);
      never_resign = dis(game_rng) < cfg->never_resign_prob;
      has_never = true;
    }
    const bool resign = !never_resign && !(value >= -1.0 + cfg->resign_thres);
    bool finished = false;
    if (resign && orc_ply(st) >= 50) finished = true;                                  // finish_game(FR_RESIGN) :387-390
    else if (!g_preload.empty() && sgf_iter >= g_preload.size()) finished = true;      // SGF exhausted: finish_game(FR_MAX_STEP) :392-396
    else {
      if (!g_preload.empty()) c = g_preload[sgf_iter++];                               // "Move changes from {} to {}" :397-405
      if (!orc_forward(st, c)) { orc_free(st); return -3; }                            // "Something is wrong! Move cannot be applied"
      moves.push_back(c);
      if (orc_terminated(st)) finished = true;                                         // :420-425
      if (cfg->move_cutoff > 0 && orc_ply(st) >= cfg->move_cutoff) finished = true;    // :427-429
      (void)max_move;
    }
    if (finished) {   // finish_game :121-149: _ai->endGame / _ai2->endGame (resetTree), _state_ext.restart()
      // FR_CHEAT_SELFPLAY_RANDOM_RESULT (:126-129, GoStateExt::setFinalValue go_state_ext.h:96-99): the result of a self-play game
      // is a draw of the game's generator -- one more draw in the stream the next game's move sampling reads
      if (!two && cfg->cheat_selfplay_random_result) (void)game_rng();
      ai->end_game();
      if (ai2 != nullptr) ai2->end_game();
      orc_reset(st); moves.clear();
      never_resign = false; has_never = false;
    }
  }
  if (stats) { stats[0] = batches_before + actors[0].batches + actors[1].batches; stats[1] = rows_before + actors[0].rows + actors[1].rows; stats[2] = 0; }
  orc_free(st);
  return k;
}

}  // extern "C"
