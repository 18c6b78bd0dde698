// Host side of the self-play loop over the device engines (include/elf_amd.h, elfsp_*).
//
// Mirrors, for G games advanced step by step on one GPU, what the reference does on one std::thread per game:
//   src_cpp/elfgames/go/common/game_selfplay.cc   GoGameSelfPlay::act :272-430, init_ai :30-78, restart :158-220, OnReceive :222-270,
//                                                  mcts_make_diverse_move :80-95, mcts_update_info :97-119, finish_game :121-149
//   src_cpp/elfgames/go/common/dispatcher_callback.h  OnFirstSend :28-44 (idle game threads), OnReply :46-103 (game_start)
//   src_cpp/elf/base/dispatcher.h                  checkMessage :42-62, process_request :104-152 (request -> every game -> barrier)
//   src_cpp/elf/ai/tree_search/mcts.h             MCTSAI_T::act / actPolicyOnly / align_state / advanceMoves :59-90,141-167
//   src_cpp/elf/ai/tree_search/tree_search.h      TreeSearchT::run :410-426, runPolicyOnly :385-407, chooseAction :495-528
//   src_cpp/elf/ai/tree_search/tree_search_base.h MCTSResultT::addActions :237-294 (most_visited / strongest_prior / uniform_random)
//   src_cpp/elfgames/go/mcts/mcts.h               MCTSGoAI::getValue / getMCTSPolicy :358-372
//   src_cpp/elfgames/go/common/game_utils.h       ResignCheck :14-54
//   src_cpp/elfgames/go/common/go_state_ext.h     setFinalValue :76-103, restart :112-124
// A game is in one of three phases: waiting for a request that gives it something to play (ModelPair::wait), restarted by a
// request and waiting for the other games to acknowledge it (UPDATE_MODEL -> UPDATE_COMPLETE), or playing.  A playing game
// is either between two searches or inside one; searches of different games need not be in step (two AIs with different
// rollout budgets, policy-only moves).  Each game owns up to two MCTS AIs ("actor_black" = tree pool 0 and, when the request
// names a model for White, "actor_white" = tree pool 1), each with its own tree, options and std::mt19937.
// Everything that depends on libstdc++'s implementation-defined distributions stays on the host and uses the
// very same library: std::gamma_distribution (Dirichlet noise), std::uniform_real_distribution (move sampling,
// never-resign draw), std::mt19937 streams per game and per actor.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <new>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <deque>
#include <functional>
#include <string>
#include <thread>

#include "engine_host.h"
#include "host_workers.h"
#include "record_host.h"

struct ElfSpSearchRec {   // == ElfSpSearch in include/elf_amd.h
  int32_t game, move_played, best_action, total_visits, n_edges;
  float root_value, max_score, predicted_value;
};

enum { PH_WAIT = 0, PH_BARRIER = 1, PH_PLAY = 2 };
enum { GMASK_IDLE = 0, GMASK_SEARCH = 1, GMASK_POLICY_ONLY = 2 };   // == GM_* of mcts.cuh

struct SpRequest {          // MsgRequest (common/record.h:119-149): ModelPair + ClientCtrl
  int64_t black_ver = -1, white_ver = -1;
  float black_thres = 0.f, white_thres = 0.f, never_resign_prob = 0.f;
  bool async = false, player_swap = false;
  int thread_used = -1;
  int client_type = 1;         // ClientCtrl.client_type: carried into the records only
  ElfTsOptions ts;             // ModelPair.mcts_opt: the search options the AIs of this request are built with
  int id = 0;
  bool wait() const { return black_ver < 0; }                           // ModelPair::wait
  bool is_selfplay() const { return black_ver >= 0 && white_ver == -1; }
  void set_wait() { black_ver = white_ver = -1; }
};

struct SpGame {
  std::mt19937 rng;            // GoGameBase::_rng (game_base.h:32-38)
  std::mt19937 actor_rng[2];   // MCTSActor::rng_ (go/mcts/mcts.h:52) of "actor_black" / "actor_white", seeded with _rng() (game_selfplay.cc:47)
  // the generators of search threads 1 .. T-1 of each AI: TreeSearchT makes one MCTSActor per thread (tree_search.h:339-343), every one
  // from the same MCTSActorParams, i.e. the same seed (game_selfplay.cc:45-47,77); thread 0's is actor_rng (also the Dirichlet source)
  std::vector<std::mt19937> thread_rng[2];
  std::mt19937 actor_rng0[2];  // the freshly seeded state of the AI's actors (what a search thread's generator starts from)
  int64_t actor_ver[2] = {-1, -1};   // MCTSActorParams.required_version of the two AIs (-1: any)
  int pool_of_colour[2] = {0, 0};    // [0] Black's AI, [1] White's AI (curr_ai, game_selfplay.cc:364-366, after player_swap :181-185)
  SpRequest req;               // GoStateExt::curr_request_
  int seen_req = 0;            // id of the last request this game has received
  int phase = PH_WAIT;
  int online_counter = 0;      // GoGameSelfPlay::_online_counter: requests are looked at every 5 acts (:273-289)
  // ResignCheck (game_utils.h:14-54)
  bool never_resign = false, has_calculated_never_resign = false;
  float last_predicted = 0.0f;
  int ply = 1;                 // GoState::getPly of the game board
  int seq = 1;                 // GoStateExt::_seq: 1 after the constructor's restart(), +1 per restart()
  float last_final = 0.0f;     // GoStateExt::getLastGameFinalValue (go_state_ext.h:153-155)
  int last_move = -1;          // what the last finished search of this game did: the Coord it forwarded, M_RESIGN, or -1 (none yet)
  int sgf_iter = 0;            // GoGameSelfPlay::_sgf_iter (game_selfplay.h): next move of the preloaded SGF
  // the search in progress
  int ai = -1;                 // tree pool searching now, -1 = between two searches
  int step = 0, steps = 0;     // batches done / batches of this search
  bool policy_only = false;    // MCTSAI_T::actPolicyOnly instead of act
  SpRecord rec;                // GoStateExt::_mcts_policies / _predicted_values / the game's moves (go_state_ext.h:131-148)
  std::set<int64_t> using_models;   // GoStateExt::using_models_: every model version this game has been played with
  void add_current_model() {   // GoStateExt::addCurrentModel (go_state_ext.h:68-73)
    if (req.black_ver >= 0) using_models.insert(req.black_ver);
    if (req.white_ver >= 0) using_models.insert(req.white_ver);
  }
};

struct SpPool {              // one MCTSGoAI role: its tree pool and options
  ElfMcts* mcts = nullptr;
  ElfMctsOptions mo{};
  int rollouts_per_thread = 0, K = 0, T = 1, KT = 0, steps_per_move = 0, W = 0;
  int32_t* d_counts = nullptr;           // [4]: rows, error bits, running total of rows (u64)
  int32_t h_counts[2] = {0, 0};
  int last_rows = 0;                     // rows of the last select, -1 = left on the device
  uint8_t *d_start = nullptr, *d_active = nullptr;
  int64_t* d_ver = nullptr;
  std::vector<uint8_t> h_start, h_active, up_active;   // up_active: what d_active holds
  std::vector<int64_t> h_ver, up_ver;
  std::vector<uint8_t> h_d4;
  std::vector<int32_t> h_tdraws;         // [G][T] D4 draws per search thread of the move that just ended (elfmcts_thread_draws)
  int n_active = 0;
  bool selected = false;                 // a select of this step is waiting for its expand
};

struct ElfSelfPlay {
  int64_t ts_deferred = 0;               // requests whose TSOptions could not be applied because other games were still playing (sp_poll_requests)
  ElfSpOptions opt;
  ElfGoEngine* eng = nullptr;
  SpPool pool[2];
  int G = 0, NE = 0, NA = 0;
  hipStream_t stream = nullptr;
  std::vector<SpGame> games;
  // device scratch
  int32_t* d_info = nullptr;     // [G][8]
  int32_t *d_coord = nullptr, *d_visits = nullptr, *d_moves = nullptr, *d_ids = nullptr, *d_binfo = nullptr;
  float *d_prior = nullptr, *d_reward = nullptr, *d_etas = nullptr, *d_Z = nullptr, *d_val = nullptr;
  uint8_t* d_ok = nullptr;
  // host mirrors
  std::vector<int32_t> h_info, h_coord, h_visits, h_moves, h_binfo;
  std::vector<float> h_prior, h_reward, h_etas, h_Z, h_val;
  std::vector<uint8_t> h_ok;
  bool step_open = false;        // between begin_step and end_step
  // requests: `cur` is the one being delivered (every game must receive it before the next one goes out, dispatcher.h:104-152)
  std::deque<SpRequest> mailbox;
  SpRequest cur;
  bool cur_done = true, cur_restarted = false;   // cur_restarted: the request changed the model of some game ("game_start" is due)
  int cur_n_restart = 0;                         // games that actually restarted under `cur` (RestartReply::UPDATE_MODEL)
  int next_req_id = 1;
  int64_t start_black = 0, start_white = -1;   // versions of the last request that (re)started games
  int game_starts = 0;
  bool explicit_request = false;
  ElfSpOptions opt0;             // the options at creation: the TSOptions of a request that names none (GameContext::setRequest)
  std::mt19937 pick_rng;         // MCTSResultT::addActions' static rng (tree_search_base.h:238), uniform_random only
  std::chrono::steady_clock::time_point t_after_drain;
  // statistics / capture
  int64_t n_moves = 0, n_games = 0, n_rollouts = 0, n_rows = 0, n_steps = 0;
  // move boundaries: boundary_wait_ns = the first wait of sp_finish_moves (the stream drains whatever the host had queued ahead:
  // pipeline depth, not boundary work); boundary_ns = everything after it up to the end of the next search start
  int64_t boundary_ns = 0, boundary_wait_ns = 0, n_boundaries = 0;
  double sum_final = 0.0;
  std::vector<ElfSpSearchRec> log_search;
  std::vector<int32_t> log_coord, log_visits;
  std::vector<float> log_prior, log_reward;
  int log_cap = 0;
  // finished-game records (GameNotifier::OnGameEnd -> GoStateExt::dumpRecord), newest at the back
  std::deque<std::string> records;
  std::deque<float> finished_values;   // final value of every finished game not yet taken (GameStats::feedWinRate, game_stats.h:41-44)
  std::vector<int32_t> sgf;   // GameOptions.preload_sgf as reference Coords (elfsp_preload)
  int sgf_move_to = -1;       // GameOptions.preload_sgf_move_to
};

#define SPCHK(x)                  \
  do {                            \
    int _rc = (x);                \
    if (_rc != 0) return _rc;     \
  } while (0)

// Per-game host work of a move boundary (gamma draws for the Dirichlet noise, the D4 draws of the coming search) touches only
// that game's generators: spread over a few host threads when many games are at the boundary together.  The threads are a
// process-wide pool created at the first use and parked on a condition variable in between (a move boundary every 200 steps
// would otherwise pay for 16 thread creations + joins each time).
template <class F>
static void sp_for_games(const std::vector<int32_t>& ids, F fn) {
  const size_t n = ids.size();
  unsigned nt = host_worker_count(16);
  if (n < 16 || nt < 2) { for (int g : ids) fn(g); return; }
  if (nt > n / 4) nt = (unsigned)(n / 4);
  HostWorkers::get().run(n, nt, [&](size_t i) { fn(ids[i]); });
}

static ElfTsOptions sp_ts_of(const ElfSpOptions& o);
static bool sp_ts_pool_equal(const ElfTsOptions& a, const ElfTsOptions& b);
static SpRecordMeta sp_meta(const ElfSelfPlay* sp, const SpGame& gm) {
  // Record.request = curr_request_ (go_state_ext.h:134): the game's own request incl. the mcts_opt it carried
  SpRecordMeta m = elfrec_meta_from_options(sp->opt);
  m.black_ver = gm.req.black_ver; m.white_ver = gm.req.white_ver;
  m.black_resign_thres = gm.req.black_thres; m.white_resign_thres = gm.req.white_thres; m.never_resign_prob = gm.req.never_resign_prob;
  m.num_game_thread_used = gm.req.thread_used;
  m.player_swap = gm.req.player_swap; m.async = gm.req.async;
  m.client_type = gm.req.client_type;
  // vers.mcts_opt as the request carried it -- except where this game searched with OTHER options: a request that restarted only
  // some games while the rest played on could not rebuild the context's tree pools (sp_poll_requests, "deferred"); those games'
  // records carry the search options that were actually used, not the ones the request asked for
  ElfTsOptions used = gm.req.ts;
  const ElfTsOptions ctx = sp_ts_of(sp->opt);
  if (!sp_ts_pool_equal(used, ctx)) {
    used.num_threads = ctx.num_threads; used.num_rollouts_per_thread = ctx.num_rollouts_per_thread; used.num_rollouts_per_batch = ctx.num_rollouts_per_batch;
    used.persistent_tree = ctx.persistent_tree; used.pick_method = ctx.pick_method; used.root_epsilon = ctx.root_epsilon; used.root_alpha = ctx.root_alpha;
    used.virtual_loss = ctx.virtual_loss; used.use_prior = ctx.use_prior; used.c_puct = ctx.c_puct; used.unexplored_q_zero = ctx.unexplored_q_zero;
    used.root_unexplored_q_zero = ctx.root_unexplored_q_zero;
  }
  elfrec_meta_set_ts(&m, used);
  return m;
}

static void sp_finish_record(ElfSelfPlay* sp, int g, float final_value, int final_ply) {
  SpGame& gm = sp->games[g];
  gm.last_final = final_value;
  sp->finished_values.push_back(final_value);
  if (sp->finished_values.size() > 65536) sp->finished_values.pop_front();
  if (sp->opt.keep_records > 0) {
    SpRecord& r = gm.rec;
    r.reward = final_value;                      // _state.getFinalValue()
    r.never_resign = gm.never_resign;
    r.num_move = final_ply - 1;                  // _state.getPly() - 1
    r.thread_id = (uint64_t)(sp->opt.game_idx_base + g);   // _game_idx: the same id the game's ThreadState carries (elfsp_thread_states)
    r.seq = gm.seq;                              // _seq: ctor restart() -> 1, first request restart() -> 2, then +1 per game
    r.using_models.assign(gm.using_models.begin(), gm.using_models.end());
    r.timestamp = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    if ((int)sp->records.size() >= sp->opt.keep_records) sp->records.pop_front();
    sp->records.push_back(elfrec_record_json(sp_meta(sp, gm), r));
  }
  gm.rec = SpRecord();                           // GoStateExt::restart(): _mcts_policies.clear(), _predicted_values.clear()
}

// GoStateExt::restart (go_state_ext.h:112-124), host half: resign check reset, _seq++ (the board reset is the caller's)
static void sp_state_restart(SpGame& gm) {
  gm.ply = 1; gm.never_resign = false; gm.has_calculated_never_resign = false; gm.last_predicted = 0.0f;
  gm.seq++;
  gm.rec = SpRecord();
  gm.using_models.clear();
  gm.add_current_model();
}

// GoGameSelfPlay::restart :202-219: forward the first preload_sgf_move_to moves of the preloaded SGF on the listed (fresh) game boards
static int sp_forward_preload(ElfSelfPlay* sp, const std::vector<int32_t>& ids) {
  const int k = (int)ids.size(), n = (int)sp->sgf.size();
  if (k == 0) return 0;
  HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
  int fwd = 0;
  for (; fwd < n && fwd < sp->sgf_move_to; ++fwd) {            // while (!_sgf_iter.done() && i < preload_sgf_move_to)
    std::vector<int32_t> mv(k, (int32_t)sp->sgf[fwd]);
    HIPCHK(hipMemcpyAsync(sp->d_moves, mv.data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_forward(sp->eng, sp->d_ids, sp->d_moves, k, sp->d_ok, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_ok.data(), sp->d_ok, k, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (int j = 0; j < k; ++j)
      if (sp->h_ok[j] != 1) return ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD;   // "Preload sgf: move not valid!" :211-215
    for (int g : ids) {
      sp->games[g].ply++;
      if (sp->opt.keep_records > 0) sp->games[g].rec.moves.push_back((uint16_t)sp->sgf[fwd]);
    }
  }
  for (int g : ids) sp->games[g].sgf_iter = fwd;
  return 0;
}

// effective options of the "actor_white" AI: init_ai's overrides (game_selfplay.cc:51-70) of GameOptions.white_puct /
// white_mcts_rollout_per_batch / white_mcts_rollout_per_thread
// TSOptions <-> the fields of ElfSpOptions they live in
static ElfTsOptions sp_ts_of(const ElfSpOptions& o) {
  ElfTsOptions t;
  memset(&t, 0, sizeof(t));
  t.num_threads = o.mcts.num_threads; t.num_rollouts_per_thread = o.num_rollouts_per_thread; t.num_rollouts_per_batch = o.mcts.num_rollouts_per_batch;
  t.persistent_tree = o.persistent_tree != 0; t.pick_method = o.pick_method; t.root_epsilon = o.root_epsilon; t.root_alpha = o.root_alpha;
  t.virtual_loss = o.mcts.virtual_loss; t.use_prior = o.mcts.use_prior != 0; t.unexplored_q_zero = o.mcts.unexplored_q_zero != 0;
  t.root_unexplored_q_zero = o.mcts.root_unexplored_q_zero != 0; t.c_puct = o.mcts.c_puct;
  return t;
}
static void sp_ts_into(const ElfTsOptions& t, ElfSpOptions* o) {
  o->mcts.num_threads = t.num_threads; o->num_rollouts_per_thread = t.num_rollouts_per_thread; o->mcts.num_rollouts_per_batch = t.num_rollouts_per_batch;
  o->persistent_tree = t.persistent_tree != 0; o->pick_method = t.pick_method; o->root_epsilon = t.root_epsilon; o->root_alpha = t.root_alpha;
  o->mcts.virtual_loss = t.virtual_loss; o->mcts.use_prior = t.use_prior != 0; o->mcts.unexplored_q_zero = t.unexplored_q_zero != 0;
  o->mcts.root_unexplored_q_zero = t.root_unexplored_q_zero != 0; o->mcts.c_puct = t.c_puct;
}
// the fields of a request's TSOptions that the tree pools are built from (what sp_ts_into stores): max_num_moves, seed, verbose*,
// log_prefix take part in ModelPair::operator== (sp_ts_equal) but never in the shape of a pool
static bool sp_ts_pool_equal(const ElfTsOptions& a, const ElfTsOptions& b) {
  return a.num_threads == b.num_threads && a.num_rollouts_per_thread == b.num_rollouts_per_thread && a.num_rollouts_per_batch == b.num_rollouts_per_batch &&
         (a.persistent_tree != 0) == (b.persistent_tree != 0) && a.pick_method == b.pick_method && a.root_epsilon == b.root_epsilon &&
         a.root_alpha == b.root_alpha && a.virtual_loss == b.virtual_loss && (a.use_prior != 0) == (b.use_prior != 0) && a.c_puct == b.c_puct &&
         (a.unexplored_q_zero != 0) == (b.unexplored_q_zero != 0) && (a.root_unexplored_q_zero != 0) == (b.root_unexplored_q_zero != 0);
}
static bool sp_ts_equal(const ElfTsOptions& a, const ElfTsOptions& b) {      // TSOptions::operator== (tree_search_options.h:133-180)
  return a.max_num_moves == b.max_num_moves && a.num_threads == b.num_threads && a.num_rollouts_per_thread == b.num_rollouts_per_thread &&
         a.num_rollouts_per_batch == b.num_rollouts_per_batch && (a.verbose != 0) == (b.verbose != 0) && (a.verbose_time != 0) == (b.verbose_time != 0) &&
         a.seed == b.seed && (a.persistent_tree != 0) == (b.persistent_tree != 0) && a.pick_method == b.pick_method &&
         !strncmp(a.log_prefix, b.log_prefix, sizeof(a.log_prefix)) && a.root_epsilon == b.root_epsilon && a.root_alpha == b.root_alpha &&
         a.virtual_loss == b.virtual_loss && (a.use_prior != 0) == (b.use_prior != 0) && a.c_puct == b.c_puct &&
         (a.unexplored_q_zero != 0) == (b.unexplored_q_zero != 0) && (a.root_unexplored_q_zero != 0) == (b.root_unexplored_q_zero != 0);
}

static void sp_pool_options(const ElfSpOptions& o, int a, ElfMctsOptions* mo, int* rollouts_per_thread) {
  *mo = o.mcts;
  *rollouts_per_thread = o.num_rollouts_per_thread;
  if (a == 1) {
    if (o.white_puct > 0.0f) mo->c_puct = o.white_puct;
    if (o.white_mcts_rollout_per_batch > 0) mo->num_rollouts_per_batch = o.white_mcts_rollout_per_batch;
    if (o.white_mcts_rollout_per_thread > 0) *rollouts_per_thread = o.white_mcts_rollout_per_thread;
  }
}

static void sp_pool_free(SpPool& p) {
  void* ptrs[] = {p.d_counts, p.d_start, p.d_active, p.d_ver};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  p.d_counts = nullptr; p.d_start = p.d_active = nullptr; p.d_ver = nullptr;
  if (p.mcts) elfmcts_destroy(p.mcts);
  p.mcts = nullptr;
}

static int sp_pool_create(ElfSelfPlay* sp, int a) {
  SpPool& p = sp->pool[a];
  if (p.mcts) return 0;
  const int G = sp->G;
  sp_pool_options(sp->opt, a, &p.mo, &p.rollouts_per_thread);
  if (p.mo.num_rollouts_per_batch <= 0 || p.mo.num_threads <= 0 || p.rollouts_per_thread <= 0) return ELFGO_E_BADARG;
  p.K = p.mo.num_rollouts_per_batch; p.T = p.mo.num_threads; p.KT = p.K * p.T;
  p.steps_per_move = (p.rollouts_per_thread + p.K - 1) / p.K;   // for (idx = 0; idx < num_rollout; idx += batch) tree_search.h:112-117, in every search thread
  p.W = p.steps_per_move * p.KT;
  SPCHK(elfmcts_create(sp->eng, G, sp->opt.nodes_per_game, p.W, &p.mo, &p.mcts));
  if (a == 1 && sp->pool[0].mcts) {     // the feature row format is the context's
    int fmt = 0;
    if (elfmcts_get_feature_format(sp->pool[0].mcts, &fmt) == 0) (void)elfmcts_set_feature_format(p.mcts, fmt);
  }
  HIPCHK(hipMalloc((void**)&p.d_counts, 16));
  HIPCHK(hipMemset(p.d_counts, 0, 16));
  HIPCHK(hipMalloc((void**)&p.d_start, G));
  HIPCHK(hipMalloc((void**)&p.d_active, G));
  HIPCHK(hipMalloc((void**)&p.d_ver, 8 * (size_t)G));
  p.h_start.assign(G, 0); p.h_active.assign(G, 0); p.up_active.assign(G, 0xFF);
  p.h_ver.assign(G, -1); p.up_ver.assign(G, -2);
  p.h_d4.assign((size_t)G * p.W, 0);
  p.h_tdraws.assign((size_t)G * p.T, 0);
  return 0;
}

// GoGameSelfPlay::restart (:158-220) for the listed games under their (already stored) request: new AIs seeded with the next
// draws of the game's generator (init_ai :45-47: "actor_black" first, then "actor_white"), player_swap, fresh trees, fresh state
static int sp_restart_games(ElfSelfPlay* sp, const std::vector<int32_t>& ids) {
  const int k = (int)ids.size();
  if (k == 0) return 0;
  bool need2 = false;
  for (int g : ids) need2 = need2 || sp->games[g].req.white_ver >= 0;
  if (need2) SPCHK(sp_pool_create(sp, 1));
  for (int g : ids) {
    SpGame& gm = sp->games[g];
    const bool two = gm.req.white_ver >= 0;
    gm.actor_rng[0].seed(gm.rng());
    gm.actor_rng0[0] = gm.actor_rng[0];      // every search thread's actor starts from this state (the same params.seed)
    gm.thread_rng[0].clear();                // sized at the AI's first search (the pools may be rebuilt for the request's threads first)
    gm.actor_ver[0] = gm.req.async ? -1 : gm.req.black_ver;
    if (two) {
      gm.actor_rng[1].seed(gm.rng());
      gm.actor_rng0[1] = gm.actor_rng[1];
      gm.thread_rng[1].clear();
      gm.actor_ver[1] = gm.req.async ? -1 : gm.req.white_ver;
    }
    gm.pool_of_colour[0] = 0;
    gm.pool_of_colour[1] = two ? 1 : 0;
    if (!gm.req.is_selfplay() && gm.req.player_swap && two) std::swap(gm.pool_of_colour[0], gm.pool_of_colour[1]);
    sp_state_restart(gm);
    gm.sgf_iter = 0;
    gm.ai = -1;
  }
  HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_reset(sp->eng, sp->d_ids, k, sp->stream));
  for (int a = 0; a < 2; ++a)
    if (sp->pool[a].mcts) SPCHK(elfmcts_clear(sp->pool[a].mcts, sp->d_ids, k, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));   // ids may be a temporary
  if (!sp->sgf.empty()) SPCHK(sp_forward_preload(sp, ids));
  return 0;
}

// GoGameSelfPlay::OnReceive (game_selfplay.cc:222-270) of game g for request r (already specialised by OnFirstSend).
// Returns true when the game must restart (RestartReply::UPDATE_MODEL).
static bool sp_on_receive(ElfSelfPlay* sp, int g, const SpRequest& r, bool* model_changed) {
  SpGame& gm = sp->games[g];
  const bool is_waiting = r.wait(), is_prev_waiting = gm.req.wait();
  const bool same_vers = r.black_ver == gm.req.black_ver && r.white_ver == gm.req.white_ver && sp_ts_equal(r.ts, gm.req.ts);   // ModelPair::operator==
  const bool same_swap = r.player_swap == gm.req.player_swap;
  const bool no_restart = (same_vers || r.async) && same_swap && !is_prev_waiting;
  gm.req = r;                                    // _state_ext.setRequest: thresholds follow the request (go_state_ext.h:57-66)
  gm.seen_req = r.id;
  if (is_waiting) { gm.phase = PH_WAIT; return false; }           // ONLY_WAIT
  if (!no_restart) { gm.phase = PH_BARRIER; *model_changed = true; return true; }   // UPDATE_MODEL
  gm.phase = PH_PLAY;
  if (r.async) {                                 // setAsync :150-156
    gm.actor_ver[0] = gm.actor_ver[1] = -1;
    gm.add_current_model();
    if (!same_vers) *model_changed = true;       // UPDATE_MODEL_ASYNC
  }
  return false;
}

// New search options for the whole context (all playing games have just restarted and hold no tree): rebuild the tree pools
static int sp_apply_ts(ElfSelfPlay* sp, const ElfTsOptions& t) {
  HIPCHK(hipStreamSynchronize(sp->stream));
  int fmt = 0;
  const bool have_fmt = sp->pool[0].mcts && elfmcts_get_feature_format(sp->pool[0].mcts, &fmt) == 0;
  const bool had_white = sp->pool[1].mcts != nullptr;
  const int64_t old_w = sp->pool[0].W > 0 ? sp->pool[0].W : 1;
  sp_pool_free(sp->pool[0]);
  sp_pool_free(sp->pool[1]);
  sp_ts_into(t, &sp->opt);
  // the node pool was sized by the caller for the rollouts of the options it created the context with: keep the same head room
  const int64_t new_w = (int64_t)((t.num_rollouts_per_thread + t.num_rollouts_per_batch - 1) / t.num_rollouts_per_batch) * t.num_rollouts_per_batch * t.num_threads;
  if (new_w > old_w)   // (a multiple of 64: elfmcts_create)
    sp->opt.nodes_per_game = (int)std::min<int64_t>((((int64_t)sp->opt.nodes_per_game * new_w / old_w + 63) / 64) * 64, (int64_t)1 << 30);
  SPCHK(sp_pool_create(sp, 0));
  if (have_fmt) SPCHK(elfmcts_set_feature_format(sp->pool[0].mcts, fmt));
  if (had_white) SPCHK(sp_pool_create(sp, 1));
  for (SpGame& gm : sp->games) gm.ai = -1;
  return 0;
}

// What the dispatcher thread and the games' checkMessage calls do between two searches: deliver the current request to every
// game that looks at its mailbox now (a waiting game always does, a playing one at every fifth act), restart those that must,
// and once every game has answered send "game_start" and release the restarted ones.
static int sp_poll_requests(ElfSelfPlay* sp) {
  for (;;) {
    if (sp->cur_done) {
      if (sp->mailbox.empty()) return 0;
      sp->cur = sp->mailbox.front();
      sp->mailbox.pop_front();
      sp->cur_done = false; sp->cur_restarted = false; sp->cur_n_restart = 0;
    }
    std::vector<int32_t> restart;
    int pending = 0;
    for (int g = 0; g < sp->G; ++g) {
      SpGame& gm = sp->games[g];
      if (gm.seen_req == sp->cur.id) continue;
      const bool looks = gm.phase == PH_WAIT || (gm.phase == PH_PLAY && gm.ai < 0 && gm.online_counter % 5 == 0);
      if (!looks) { ++pending; continue; }
      SpRequest r = sp->cur;
      if (r.thread_used >= 0 && g >= r.thread_used) r.set_wait();   // DispatcherCallback::OnFirstSend :28-44
      if (sp_on_receive(sp, g, r, &sp->cur_restarted)) restart.push_back(g);
    }
    sp->cur_n_restart += (int)restart.size();
    SPCHK(sp_restart_games(sp, restart));
    if (pending) return 0;
    // every game has replied: OnReply :46-103
    if (sp->cur_restarted) {
      sp->game_starts++;
      sp->start_black = sp->cur.black_ver; sp->start_white = sp->cur.white_ver;
      // the restarted games' AIs are built from the request's TSOptions (restart() :166-180).  The pools belong to the whole
      // context, so they are rebuilt for other options only when some game actually restarted (an async model update restarts
      // nobody: setAsync :150-156 only clears required_version, the AIs keep the TSOptions they were built with) and no game is
      // in the middle of play under the old options -- then every playing game is idle at the barrier and holds no tree.
      bool playing_on = false;
      for (const SpGame& gm : sp->games) playing_on = playing_on || gm.phase == PH_PLAY;
      const bool ts_differ = !sp_ts_pool_equal(sp->cur.ts, sp_ts_of(sp->opt));
      if (sp->cur_n_restart > 0 && !playing_on && ts_differ) SPCHK(sp_apply_ts(sp, sp->cur.ts));
      else if (sp->cur_n_restart > 0 && playing_on && ts_differ) {
        // some games restarted under a request whose search options differ from the pools', while others play on under the old
        // ones: the pools belong to the whole context, so the restarted games search with the OLD options (the reference would build
        // their AIs from the request's).  Not silent: counted (elfsp_ts_requests_deferred) and reported once per context.
        if (sp->ts_deferred++ == 0)
          fprintf(stderr, "elf_amd: a request's search options differ from the context's while games are still playing: the %d restarted "
                          "game(s) keep the context's options until a request restarts every game\n", sp->cur_n_restart);
      }
    }
    for (SpGame& gm : sp->games) if (gm.phase == PH_BARRIER) gm.phase = PH_PLAY;
    sp->cur_done = true;
  }
}

static int sp_upload_masks(ElfSelfPlay* sp, SpPool& p) {
  if (p.h_active != p.up_active) {
    HIPCHK(hipMemcpyAsync(p.d_active, p.h_active.data(), sp->G, hipMemcpyHostToDevice, sp->stream));
    p.up_active = p.h_active;
  }
  if (p.h_ver != p.up_ver) {
    HIPCHK(hipMemcpyAsync(p.d_ver, p.h_ver.data(), 8 * (size_t)sp->G, hipMemcpyHostToDevice, sp->stream));
    p.up_ver = p.h_ver;
  }
  return 0;
}

// act() up to the first batch for every playing game that is between two searches: request check, curr_ai, align_state,
// setRootNodeState, Dirichlet noise, the D4 draws of this move
static int sp_begin_searches(ElfSelfPlay* sp) {
  const int G = sp->G;
  // games between two searches reach the top of act(): `_online_counter % 5 == 0` -> checkMessage
  SPCHK(sp_poll_requests(sp));
  std::vector<int32_t> starting[2];
  for (int g = 0; g < G; ++g) {
    SpGame& gm = sp->games[g];
    if (gm.phase != PH_PLAY || gm.ai >= 0) continue;
    gm.online_counter++;
    const int colour = (gm.ply & 1) ? 0 : 1;      // ply 1 = Black to move
    const int a = gm.pool_of_colour[colour];
    gm.ai = a;
    gm.policy_only = colour == 0 ? sp->opt.black_use_policy_network_only != 0 : sp->opt.white_use_policy_network_only != 0;
    gm.step = 0;
    gm.steps = gm.policy_only ? 1 : sp->pool[a].steps_per_move;
    starting[a].push_back(g);
  }
  for (int a = 0; a < 2; ++a) {
    if (starting[a].empty()) continue;
    SpPool& p = sp->pool[a];
    const int k = (int)starting[a].size();
    // MCTSAI_T::act -> align_state (mcts.h:141-167): advanceMoves happened when the moves were played (both trees follow every
    // move); a tree that is not persistent is reset now
    if (!sp->opt.persistent_tree) {
      HIPCHK(hipMemcpyAsync(sp->d_ids, starting[a].data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
      SPCHK(elfmcts_clear(p.mcts, sp->d_ids, k, sp->stream));
      HIPCHK(hipStreamSynchronize(sp->stream));
    }
    std::fill(p.h_start.begin(), p.h_start.end(), (uint8_t)GMASK_IDLE);
    bool any_search = false;
    for (int g : starting[a]) {
      p.h_start[g] = sp->games[g].policy_only ? GMASK_POLICY_ONLY : GMASK_SEARCH;
      p.h_active[g] = p.h_start[g];
      p.h_ver[g] = sp->games[g].actor_ver[a];
      any_search = any_search || !sp->games[g].policy_only;
    }
    const bool all = k == G;
    if (!all) HIPCHK(hipMemcpyAsync(p.d_start, p.h_start.data(), G, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfmcts_set_game_mask(p.mcts, all ? nullptr : p.d_start));
    // TreeSearchT::run :410-417 / runPolicyOnly :385-392: setRootNodeState
    SPCHK(elfmcts_set_root(p.mcts, nullptr, sp->stream));
    if (sp->opt.root_epsilon > 0.0f && any_search) {
      SPCHK(elfmcts_root(p.mcts, sp->d_info, nullptr, nullptr, nullptr, nullptr, nullptr, sp->stream));
      HIPCHK(hipMemcpyAsync(sp->h_info.data(), sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS, hipMemcpyDeviceToHost, sp->stream));
      HIPCHK(hipStreamSynchronize(sp->stream));
      // NodeT::enhanceExploration (tree_search_node.h:132-155), draws from actors_[0]->rng(); not part of runPolicyOnly
      for (int g : starting[a])
        if (sp->h_info[g * ELFMCTS_ROOT_WORDS + 6]) return ELFGO_E_MCTS_BASE - sp->h_info[g * ELFMCTS_ROOT_WORDS + 6];
      sp_for_games(starting[a], [&](int g) {
        float* et = &sp->h_etas[(size_t)g * sp->NE];
        if (sp->games[g].policy_only) { sp->h_Z[g] = 1.0f; return; }
        const int n = sp->h_info[g * ELFMCTS_ROOT_WORDS + 0];
        std::gamma_distribution<> dis(sp->opt.root_alpha);
        float Z = 1e-10;
        for (int i = 0; i < n; ++i) {
          et[i] = dis(sp->games[g].actor_rng[a]);
          Z += et[i];
        }
        sp->h_Z[g] = Z;
      });
      // policy-only games of this batch must not receive noise: they are masked out for the noise launch
      bool mixed = false;
      for (int g : starting[a]) mixed = mixed || sp->games[g].policy_only;
      if (mixed) {
        std::vector<uint8_t> m2(p.h_start);
        for (int g : starting[a]) if (sp->games[g].policy_only) m2[g] = GMASK_IDLE;
        HIPCHK(hipMemcpyAsync(p.d_start, m2.data(), G, hipMemcpyHostToDevice, sp->stream));
        SPCHK(elfmcts_set_game_mask(p.mcts, p.d_start));
      }
      HIPCHK(hipMemcpyAsync(sp->d_etas, sp->h_etas.data(), sizeof(float) * (size_t)G * sp->NE, hipMemcpyHostToDevice, sp->stream));
      HIPCHK(hipMemcpyAsync(sp->d_Z, sp->h_Z.data(), sizeof(float) * G, hipMemcpyHostToDevice, sp->stream));
      SPCHK(elfmcts_dirichlet(p.mcts, sp->d_etas, sp->d_Z, sp->opt.root_epsilon, sp->stream));
      if (mixed) HIPCHK(hipStreamSynchronize(sp->stream));   // m2 is a temporary
    }
    // BoardFeature::RandomShuffle draws of this move (go/mcts/mcts.h:175-183), from a copy of the actor stream
    // one window per search thread: thread t's actor draws from its own generator
    sp_for_games(starting[a], [&](int g) {
      const int wt = p.W / p.T;
      std::vector<std::mt19937>& tr = sp->games[g].thread_rng[a];
      if ((int)tr.size() != p.T - 1) tr.assign(p.T > 1 ? p.T - 1 : 0, sp->games[g].actor_rng0[a]);   // a new AI: untouched generators
      for (int t = 0; t < p.T; ++t) {
        if (t > 0 && sp->games[g].policy_only) break;      // runPolicyOnly evaluates with actors_[0] (tree_search.h:396-399)
        std::mt19937 c = t == 0 ? sp->games[g].actor_rng[a] : sp->games[g].thread_rng[a][t - 1];
        uint8_t* d = &p.h_d4[(size_t)g * p.W + (size_t)t * wt];
        const int w = sp->games[g].policy_only ? 1 : wt;
        for (int i = 0; i < w; ++i) d[i] = (uint8_t)(c() % 8);
      }
    });
    SPCHK(elfmcts_set_d4(p.mcts, p.h_d4.data(), sp->stream));
  }
  return 0;
}

// finish_game :121-149 for the listed games with final values already known: records, tree resets, state restart
static int sp_restart_finished(ElfSelfPlay* sp, const std::vector<int32_t>& ids) {
  const int k = (int)ids.size();
  if (k == 0) return 0;
  // _ai->endGame / _ai2->endGame (resetTree), _state_ext.restart() (state reset, resign check reset)
  HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_reset(sp->eng, sp->d_ids, k, sp->stream));
  for (int a = 0; a < 2; ++a)
    if (sp->pool[a].mcts) SPCHK(elfmcts_clear(sp->pool[a].mcts, sp->d_ids, k, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));   // ids may be a temporary
  for (int g : ids) sp_state_restart(sp->games[g]);
  sp->n_games += k;
  return 0;
}

// GoStateExt::setFinalValue (go_state_ext.h:76-103) with finish_game's cheat overrides (:122-129); `evaluated` = GoState::evaluate(komi)
static float sp_final_value(ElfSelfPlay* sp, SpGame& gm, int reason, float evaluated) {
  if (!gm.req.is_selfplay() && sp->opt.cheat_eval_new_model_wins_half) {
    const size_t h = std::hash<std::string>{}(std::to_string(gm.req.black_ver)) ^ std::hash<std::string>{}(std::to_string(gm.req.white_ver));
    float fv = h % 2 == 0 ? 1.0f : -1.0f;
    if (gm.req.player_swap) fv = -fv;
    return fv;
  }
  if (gm.req.is_selfplay() && sp->opt.cheat_selfplay_random_result) return gm.rng() % 2 == 0 ? 1.0f : -1.0f;
  if (reason == ELFSP_FR_RESIGN) return ((gm.ply & 1) == 1) ? -1.0f : 1.0f;   // nextPlayer() == S_WHITE ? 1 : -1; ply 1 = Black to move
  return evaluated;
}
// the part of act() after the search (:373-429) for every game whose search has just had its last batch
static int sp_finish_moves(ElfSelfPlay* sp, const std::vector<int32_t> (&done)[2]) {
  const int G = sp->G, NE = sp->NE;
  std::vector<int32_t> finished, sgf_done, movers;
  std::fill(sp->h_moves.begin(), sp->h_moves.end(), -1);
  bool first_wait = true;
  for (int a = 0; a < 2; ++a) {
    if (done[a].empty()) continue;
    SpPool& p = sp->pool[a];
    SPCHK(elfmcts_root(p.mcts, sp->d_info, sp->d_coord, sp->d_visits, sp->d_prior, sp->d_reward, nullptr, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_info.data(), sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_coord.data(), sp->d_coord, sizeof(int32_t) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_visits.data(), sp->d_visits, sizeof(int32_t) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_prior.data(), sp->d_prior, sizeof(float) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_reward.data(), sp->d_reward, sizeof(float) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
    if (p.T > 1) SPCHK(elfmcts_thread_draws(p.mcts, p.h_tdraws.data(), sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    if (first_wait) { sp->t_after_drain = std::chrono::steady_clock::now(); first_wait = false; }
    // online mode, following_pass (mcts_update_info :104-111): Tromp-Taylor score and last move of the game boards
    std::vector<float> tt_score;
    if (sp->opt.following_pass) {
      const int k = (int)done[a].size();
      tt_score.resize(k);
      HIPCHK(hipMemcpyAsync(sp->d_ids, done[a].data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
      SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, k, sp->opt.mcts.komi, sp->d_val, sp->stream));
      SPCHK(elfgo_info(sp->eng, sp->d_ids, k, sp->d_binfo, sp->stream));
      HIPCHK(hipMemcpyAsync(tt_score.data(), sp->d_val, 4 * k, hipMemcpyDeviceToHost, sp->stream));
      HIPCHK(hipMemcpyAsync(sp->h_binfo.data(), sp->d_binfo, sizeof(int32_t) * k * ELFGO_INFO_WORDS, hipMemcpyDeviceToHost, sp->stream));
      HIPCHK(hipStreamSynchronize(sp->stream));
    }
    for (size_t di = 0; di < done[a].size(); ++di) {
      const int g = done[a][di];
      SpGame& gm = sp->games[g];
      const int32_t* info = &sp->h_info[g * ELFMCTS_ROOT_WORDS];
      if (info[6]) return ELFGO_E_MCTS_BASE - info[6];
      gm.actor_rng[a].discard((unsigned long long)info[5]);   // D4 draws the search consumed (thread 0's actor)
      for (int t = 1; t < p.T && t - 1 < (int)gm.thread_rng[a].size(); ++t)
        gm.thread_rng[a][t - 1].discard((unsigned long long)p.h_tdraws[(size_t)g * p.T + t]);
      p.h_active[g] = GMASK_IDLE;
      const int n = info[0];
      const int32_t* coord = &sp->h_coord[(size_t)g * NE];
      const int32_t* visits = &sp->h_visits[(size_t)g * NE];
      const float* prior = &sp->h_prior[(size_t)g * NE];
      const float* reward = &sp->h_reward[(size_t)g * NE];
      float root_value;
      memcpy(&root_value, &info[4], 4);
      // chooseAction :495-528 / runPolicyOnly :401-405 with MCTSResultT::addActions (tree_search_base.h:237-294)
      const int method = gm.policy_only ? ELFSP_PICK_STRONGEST_PRIOR : sp->opt.pick_method;
      int best_action = M_INVALID, total_visits = 0, best_i = -1, random_idx = 0;
      float max_score = -3.402823466e+38f;
      if (method == ELFSP_PICK_UNIFORM_RANDOM && n > 0) random_idx = (int)(sp->pick_rng() % (unsigned)n);
      std::vector<float> scores(n);
      for (int i = 0; i < n; ++i) {
        const float score = method == ELFSP_PICK_MOST_VISITED ? (float)visits[i] : method == ELFSP_PICK_STRONGEST_PRIOR ? prior[i] : 1.0f;
        scores[i] = score;
        total_visits += visits[i];
        if (method == ELFSP_PICK_UNIFORM_RANDOM) {
          if (i == random_idx) { max_score = score; best_action = coord[i]; best_i = i; }
        } else if (score > max_score) { max_score = score; best_action = coord[i]; best_i = i; }
      }
      int c = best_action;
      if (!gm.policy_only) {
        // mcts_make_diverse_move (game_selfplay.cc:80-95): MCTSPolicy::normalize (tree_search_base.h:190-203) + sampleAction
        const bool diverse = gm.ply <= sp->opt.policy_distri_cutoff;
        const bool keep_policy = sp->opt.keep_records > 0 && (diverse || sp->opt.policy_distri_training_for_all);
        std::vector<std::pair<int, float>> policy;
        if ((diverse && n > 0) || keep_policy) {
          policy.resize(n);
          float exp_sum = 0;
          for (int i = 0; i < n; ++i) {
            float e = std::pow(scores[i], 1.0 / 1.0f);
            policy[i] = std::make_pair(coord[i], e);
            exp_sum += e;
          }
          for (auto& pe : policy) pe.second /= exp_sum;
        }
        if (diverse && n > 0) {
          // elf_utils::sample_multinomial (elf/utils/utils.h:159-182)
          float Z = 0.0;
          for (const auto& pe : policy) Z += pe.second;
          std::uniform_real_distribution<> dis(0, Z);
          float rd = dis(gm.rng);
          std::vector<float> accu(n + 1);
          accu[0] = 0;
          size_t pick = n - 1;
          for (size_t i = 1; i < accu.size(); i++) {
            accu[i] = policy[i - 1].second + accu[i - 1];
            if (rd < accu[i]) { pick = i - 1; break; }
          }
          c = policy[pick].first;
        }
        if (keep_policy) {   // _state_ext.addMCTSPolicy(policy) :89-92
          std::vector<int32_t> pc(n);
          std::vector<float> pp(n);
          for (int i = 0; i < n; ++i) { pc[i] = policy[i].first; pp[i] = policy[i].second; }
          elfrec_append_policy(sp->opt.board_size, pc.data(), pp.data(), n, &gm.rec.policies);
        }
      }
      // mcts_update_info :97-119 with MCTSGoAI::getValue (go/mcts/mcts.h:358-365)
      float predicted = root_value;
      if (total_visits != 0 && best_i >= 0) predicted = reward[best_i] / visits[best_i];
      gm.last_predicted = predicted;
      if (sp->opt.keep_records > 0) gm.rec.values.push_back(predicted);   // addPredictedValue, mcts_update_info :98-100
      if (sp->opt.following_pass) {   // "If the opponent wants pass, and we are in good, we follow." :104-111 (human games)
        const bool black = (gm.ply & 1) == 1;
        const float sc = tt_score[di];
        const bool we_are_good = black ? (sc > 0 && predicted > 0.9) : (sc < 0 && predicted < -0.9);
        if (we_are_good && sp->h_binfo[di * ELFGO_INFO_WORDS + 2] == M_PASS && gm.ply > 1) c = M_PASS;
      }
      if (sp->log_cap > 0 && (int)sp->log_search.size() < sp->log_cap) {
        ElfSpSearchRec r;
        r.game = g; r.move_played = c; r.best_action = best_action; r.total_visits = total_visits; r.n_edges = n;
        r.root_value = root_value; r.max_score = max_score; r.predicted_value = predicted;
        sp->log_search.push_back(r);
        sp->log_coord.insert(sp->log_coord.end(), coord, coord + NE);
        sp->log_visits.insert(sp->log_visits.end(), visits, visits + NE);
        sp->log_prior.insert(sp->log_prior.end(), prior, prior + NE);
        sp->log_reward.insert(sp->log_reward.end(), reward, reward + NE);
      }
      // shouldResign (go_state_ext.h:207-214) -> ResignCheck::check (game_utils.h:24-40); side to move = parity of ply
      bool resign = false;
      {
        const bool black = (gm.ply & 1) == 1;   // ply 1 = Black to move
        const float value = black ? predicted : -predicted;
        if (!gm.has_calculated_never_resign) {
          std::uniform_real_distribution<> dis(0.0, 1.0);
          gm.never_resign = (dis(gm.rng) < gm.req.never_resign_prob);
          gm.has_calculated_never_resign = true;
        }
        const float thres = (gm.req.black_thres + gm.req.white_thres) / 2.0;   // setRequest, go_state_ext.h:62-66
        if (!gm.never_resign && !(value >= -1.0 + thres)) resign = true;
      }
      gm.ai = -1;
      sp->n_moves++;
      if (resign && gm.ply >= 50) {
        gm.last_move = M_RESIGN;
        const float fv = sp_final_value(sp, gm, ELFSP_FR_RESIGN, 0.0f);   // finish_game(FR_RESIGN) :387-391
        sp->sum_final += fv;
        sp_finish_record(sp, g, fv, gm.ply);
        finished.push_back(g);
      } else if (!sp->sgf.empty() && gm.sgf_iter >= (int)sp->sgf.size()) {
        sgf_done.push_back(g);         // preloaded SGF exhausted: finish_game(FR_MAX_STEP), game_selfplay.cc:392-396
      } else {
        if (!sp->sgf.empty()) c = sp->sgf[gm.sgf_iter++];       // "Move changes from {} to {}" :397-405
        sp->h_moves[g] = c;
        gm.last_move = c;
        movers.push_back(g);
      }
    }
  }
  if (!sgf_done.empty()) {
    // setFinalValue(FR_MAX_STEP) = GoState::evaluate(komi) of the position the search started from; no move is forwarded
    const int k = (int)sgf_done.size();
    HIPCHK(hipMemcpyAsync(sp->d_ids, sgf_done.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, k, sp->opt.mcts.komi, sp->d_val, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * k, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (int i = 0; i < k; ++i) {
      SpGame& gm = sp->games[sgf_done[i]];
      const float fv = sp_final_value(sp, gm, ELFSP_FR_MAX_STEP, sp->h_val[i]);
      sp->sum_final += fv;
      sp_finish_record(sp, sgf_done[i], fv, gm.ply);
      finished.push_back(sgf_done[i]);
    }
  }
  std::vector<int32_t> by_end;
  std::sort(movers.begin(), movers.end());   // with every game moving, list position == game index (the ids == NULL launches below)
  if (!movers.empty()) {
    // GoStateExt::forward(c) (game_selfplay.cc:408) on the real game boards; the trees of both AIs follow (advanceMoves -> treeAdvance)
    const int k = (int)movers.size();
    std::vector<int32_t> mv(k);
    for (int i = 0; i < k; ++i) mv[i] = sp->h_moves[movers[i]];
    const bool all = k == G;
    if (all) {
      HIPCHK(hipMemcpyAsync(sp->d_moves, sp->h_moves.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, sp->stream));
      SPCHK(elfgo_forward(sp->eng, nullptr, sp->d_moves, G, sp->d_ok, sp->stream));
    } else {
      HIPCHK(hipMemcpyAsync(sp->d_ids, movers.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, sp->stream));
      HIPCHK(hipMemcpyAsync(sp->d_moves, mv.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, sp->stream));
      SPCHK(elfgo_forward(sp->eng, sp->d_ids, sp->d_moves, k, sp->d_ok, sp->stream));
      HIPCHK(hipStreamSynchronize(sp->stream));   // d_moves is reused for the per-game list below
      HIPCHK(hipMemcpyAsync(sp->d_moves, sp->h_moves.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, sp->stream));
    }
    if (sp->opt.persistent_tree)
      for (int a = 0; a < 2; ++a)
        if (sp->pool[a].mcts) SPCHK(elfmcts_advance(sp->pool[a].mcts, sp->d_moves, sp->stream));   // games without a move: -1
    SPCHK(elfgo_info(sp->eng, all ? nullptr : sp->d_ids, k, sp->d_binfo, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_binfo.data(), sp->d_binfo, sizeof(int32_t) * k * ELFGO_INFO_WORDS, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_ok.data(), sp->d_ok, k, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (int i = 0; i < k; ++i) {
      const int g = movers[i];
      SpGame& gm = sp->games[g];
      if (sp->h_ok[i] != 1) return ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD;   // "Something is wrong! Move cannot be applied" :409-418
      const int32_t* bi = &sp->h_binfo[i * ELFGO_INFO_WORDS];
      gm.ply = bi[0];
      if (sp->opt.keep_records > 0) gm.rec.moves.push_back((uint16_t)sp->h_moves[g]);   // GoState::_moves
      const bool terminated = bi[9] != 0;
      if (terminated || (sp->opt.move_cutoff > 0 && gm.ply >= sp->opt.move_cutoff)) by_end.push_back(g);   // :420-429
    }
  }
  if (!by_end.empty()) {
    // finish_game -> setFinalValue: GoState::evaluate(komi) (go_state_ext.h:100-102)
    const int k = (int)by_end.size();
    HIPCHK(hipMemcpyAsync(sp->d_ids, by_end.data(), sizeof(int32_t) * k, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, k, sp->opt.mcts.komi, sp->d_val, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * k, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (int i = 0; i < k; ++i) {
      SpGame& gm = sp->games[by_end[i]];
      const float fv = sp_final_value(sp, gm, ELFSP_FR_MAX_STEP, sp->h_val[i]);
      sp->sum_final += fv;
      sp_finish_record(sp, by_end[i], fv, gm.ply);
    }
    finished.insert(finished.end(), by_end.begin(), by_end.end());
  }
  SPCHK(sp_restart_finished(sp, finished));
  return 0;
}

static void sp_push_request(ElfSelfPlay* sp, SpRequest r) {
  r.id = sp->next_req_id++;
  sp->mailbox.push_back(r);
}

extern "C" {

int elfsp_create(const ElfSpOptions* o, int device, const uint64_t* zobrist_host, ElfSelfPlay** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return ELFGO_E_BADARG;
  DevGuard _dg(device);
  if (!o || !out || !zobrist_host || o->num_games <= 0 || o->num_rollouts_per_thread <= 0 || o->mcts.num_threads <= 0 ||
      o->mcts.num_rollouts_per_batch <= 0 || o->pick_method < ELFSP_PICK_MOST_VISITED || o->pick_method > ELFSP_PICK_UNIFORM_RANDOM)
    return ELFGO_E_BADARG;
  ElfSelfPlay* sp = new (std::nothrow) ElfSelfPlay();
  if (!sp) return ELFGO_E_NOMEM;
  sp->opt = *o;
  const int G = o->num_games;
  int rc = elfgo_create(o->board_size, G, device, zobrist_host, &sp->eng);
  if (rc) { delete sp; return rc; }
  sp->G = G;
  rc = sp_pool_create(sp, 0);
  if (rc) { elfsp_destroy(sp); return rc; }
  sp->NE = elfmcts_edge_stride(sp->pool[0].mcts); sp->NA = o->board_size * o->board_size + 1;
  sp->games.resize(G);
  const uint64_t now_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                              std::chrono::system_clock::now().time_since_epoch()).count();
  for (int g = 0; g < G; ++g) {
    // GoGameBase ctor: _rng.seed(_seed) (game_base.h:32-38).  Per-game seed rule: include/elf_amd.h (ElfSpOptions).
    const uint64_t gi = (uint64_t)(uint32_t)(o->game_idx_base + g);
    uint64_t seed;
    if (o->seed != 0) seed = (uint64_t)o->seed + gi;
    else {
      // elf_utils::get_seed(int game_idx) (elf/utils/utils.h:50-57): the game term is an int product (32-bit, wrapping)
      const int32_t idx = (int32_t)(uint32_t)(gi ^ o->job_hash);
      const int32_t term = (int32_t)((uint32_t)idx * 2341479u);
      seed = (uint64_t)(((int64_t)(now_ms / 1000) * 1000 + (int64_t)now_ms + (int64_t)term) % 100000000ll);
    }
    sp->games[g].rng.seed((std::mt19937::result_type)seed);
  }
  sp->pick_rng.seed(o->seed != 0 ? (std::mt19937::result_type)(o->seed ^ 0x5EEDu) : (std::mt19937::result_type)time(NULL));
#define A(ptr, bytes) do { hipError_t _e = hipMalloc((void**)&(ptr), (bytes)); if (_e != hipSuccess) { elfsp_destroy(sp); return (int)_e; } } while (0)
  const size_t GE = (size_t)G * sp->NE;
  A(sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS);
  A(sp->d_coord, 4 * GE); A(sp->d_visits, 4 * GE); A(sp->d_prior, 4 * GE); A(sp->d_reward, 4 * GE); A(sp->d_etas, 4 * GE);
  A(sp->d_Z, 4 * G); A(sp->d_moves, 4 * G); A(sp->d_ids, 4 * G); A(sp->d_val, 4 * G); A(sp->d_ok, G);
  A(sp->d_binfo, sizeof(int32_t) * G * ELFGO_INFO_WORDS);
#undef A
  sp->h_info.resize(G * ELFMCTS_ROOT_WORDS); sp->h_coord.resize(GE); sp->h_visits.resize(GE); sp->h_prior.resize(GE);
  sp->h_reward.resize(GE); sp->h_etas.assign(GE, 0.f); sp->h_Z.resize(G); sp->h_moves.resize(G); sp->h_val.resize(G);
  sp->h_ok.resize(G); sp->h_binfo.resize(G * ELFGO_INFO_WORDS);
  sp->log_cap = o->log_searches;
  // the request the games start under unless the caller sends one before the first step: self-play with ElfSpOptions.model_ver
  SpRequest r;
  r.black_ver = o->model_ver; r.white_ver = -1;
  r.black_thres = r.white_thres = o->resign_thres; r.never_resign_prob = o->never_resign_prob;
  r.thread_used = o->num_games;
  sp->opt0 = *o;
  r.ts = sp_ts_of(*o);
  sp->cur.ts = r.ts;
  for (SpGame& gm : sp->games) gm.req.ts = r.ts;
  sp_push_request(sp, r);
  *out = sp;
  return 0;
}

int elfsp_destroy(ElfSelfPlay* sp) {
  if (!sp) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng ? sp->eng->device : 0);
  void* ptrs[] = {sp->d_info, sp->d_coord, sp->d_visits, sp->d_prior, sp->d_reward, sp->d_etas, sp->d_Z,
                  sp->d_moves, sp->d_ids, sp->d_val, sp->d_ok, sp->d_binfo};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  sp_pool_free(sp->pool[0]);
  sp_pool_free(sp->pool[1]);
  if (sp->eng) elfgo_destroy(sp->eng);
  delete sp;
  return 0;
}

ElfGoEngine* elfsp_engine(ElfSelfPlay* sp) { return sp ? sp->eng : nullptr; }
int64_t elfsp_ts_requests_deferred(const ElfSelfPlay* sp) { return sp ? sp->ts_deferred : ELFGO_E_BADARG; }
int elfsp_ts_games_deferred(const ElfSelfPlay* sp) {
  if (!sp) return ELFGO_E_BADARG;
  const ElfTsOptions ctx = sp_ts_of(sp->opt);
  int k = 0;
  for (const SpGame& gm : sp->games) k += gm.phase == PH_PLAY && !gm.req.wait() && !sp_ts_pool_equal(gm.req.ts, ctx);
  return k;
}
ElfMcts* elfsp_mcts(ElfSelfPlay* sp) { return sp ? sp->pool[0].mcts : nullptr; }
ElfMcts* elfsp_mcts_actor(ElfSelfPlay* sp, int actor) { return (sp && actor >= 0 && actor < 2) ? sp->pool[actor].mcts : nullptr; }
int elfsp_max_rows(const ElfSelfPlay* sp) { return sp ? sp->G * sp->pool[0].KT : ELFGO_E_BADARG; }
int elfsp_max_rows_actor(const ElfSelfPlay* sp, int actor) {
  if (!sp || actor < 0 || actor > 1) return ELFGO_E_BADARG;
  if (actor == 0) return sp->G * sp->pool[0].KT;
  ElfMctsOptions mo; int rpt;
  sp_pool_options(sp->opt, 1, &mo, &rpt);
  return sp->G * mo.num_rollouts_per_batch * mo.num_threads;
}

int elfsp_begin_step2(ElfSelfPlay* sp, void* const* s_dst, int64_t stride_elems, int* n_rows, void* stream) {
  if (!sp || !s_dst || !s_dst[0] || sp->step_open) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  {
    const auto t0 = std::chrono::steady_clock::now();
    SPCHK(sp_begin_searches(sp));
    sp->boundary_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    p.selected = false;
    p.last_rows = 0;
    p.n_active = 0;
    if (!p.mcts) continue;
    for (int g = 0; g < sp->G; ++g) p.n_active += p.h_active[g] != GMASK_IDLE;
    if (p.n_active > 0 && !s_dst[a]) return ELFGO_E_BADARG;   // this AI has rows to write and nowhere to write them
  }
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    if (p.n_active == 0) continue;
    bool plain = p.n_active == sp->G;
    for (int g = 0; plain && g < sp->G; ++g) plain = p.h_active[g] == GMASK_SEARCH;
    SPCHK(sp_upload_masks(sp, p));
    SPCHK(elfmcts_set_game_mask(p.mcts, plain ? nullptr : p.d_active));
    SPCHK(elfmcts_set_required_versions(p.mcts, p.d_ver));
    SPCHK(elfmcts_select(p.mcts, nullptr, s_dst[a], stride_elems, p.d_counts, sp->stream));
    p.selected = true;
    p.last_rows = -1;
  }
  sp->step_open = true;
  if (!n_rows) return 0;          // row counts and error words stay on the device until the move boundary
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    n_rows[a] = 0;
    if (!p.selected) continue;
    HIPCHK(hipMemcpyAsync(p.h_counts, p.d_counts, 8, hipMemcpyDeviceToHost, sp->stream));
  }
  HIPCHK(hipStreamSynchronize(sp->stream));
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    if (!p.selected) continue;
    if (p.h_counts[1]) return ELFGO_E_MCTS_BASE - p.h_counts[1];
    p.last_rows = p.h_counts[0];
    n_rows[a] = p.last_rows;
  }
  return 0;
}

int elfsp_begin_step(ElfSelfPlay* sp, void* s_dst, int64_t stride_elems, int* n_rows, void* stream) {
  if (!sp || !s_dst) return ELFGO_E_BADARG;
  void* dst[2] = {s_dst, nullptr};
  int rows[2] = {0, 0};
  const int rc = elfsp_begin_step2(sp, dst, stride_elems, n_rows ? rows : nullptr, stream);
  if (rc == 0 && n_rows) *n_rows = rows[0];
  return rc;
}

int elfsp_last_rows2(ElfSelfPlay* sp, int* n_rows) {
  if (!sp || !n_rows) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    n_rows[a] = 0;
    if (!p.selected) continue;
    HIPCHK(hipMemcpyAsync(p.h_counts, p.d_counts, 8, hipMemcpyDeviceToHost, sp->stream));
  }
  HIPCHK(hipStreamSynchronize(sp->stream));
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    if (!p.selected) continue;
    if (p.h_counts[1]) return ELFGO_E_MCTS_BASE - p.h_counts[1];
    n_rows[a] = p.h_counts[0];
  }
  return 0;
}

int elfsp_last_rows(ElfSelfPlay* sp, int* n_rows) {
  int rows[2] = {0, 0};
  if (!n_rows) return ELFGO_E_BADARG;
  const int rc = elfsp_last_rows2(sp, rows);
  if (rc == 0) *n_rows = rows[0];
  return rc;
}

int elfsp_end_step2(ElfSelfPlay* sp, const float* const* pi, int64_t pi_stride_floats, const float* const* value, const int64_t* const* rv,
                    void* stream) {
  if (!sp || !sp->step_open) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  std::vector<int32_t> done[2];
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    if (!p.selected) continue;
    const float* pia = pi ? pi[a] : nullptr;
    const float* va = value ? value[a] : nullptr;
    SPCHK(elfmcts_expand(p.mcts, pia, pi_stride_floats, va, rv ? rv[a] : nullptr, p.last_rows, sp->stream));
    p.selected = false;
    for (int g = 0; g < sp->G; ++g) {
      if (p.h_active[g] == GMASK_IDLE) continue;
      SpGame& gm = sp->games[g];
      sp->n_rollouts += gm.policy_only ? 1 : p.KT;
      if (++gm.step >= gm.steps) done[a].push_back(g);
    }
  }
  sp->step_open = false;
  sp->n_steps++;
  if (!done[0].empty() || !done[1].empty()) {
    const auto t0 = std::chrono::steady_clock::now();
    sp->t_after_drain = t0;
    SPCHK(sp_finish_moves(sp, done));
    const auto t1 = std::chrono::steady_clock::now();
    sp->boundary_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(sp->t_after_drain - t0).count();
    sp->boundary_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - sp->t_after_drain).count();
    sp->n_boundaries++;
  }
  return 0;
}

int elfsp_end_step(ElfSelfPlay* sp, const float* pi, int64_t pi_stride_floats, const float* value, const int64_t* rv, void* stream) {
  const float* pis[2] = {pi, nullptr};
  const float* vs[2] = {value, nullptr};
  const int64_t* rvs[2] = {rv, nullptr};
  if (sp && sp->pool[1].selected) return ELFGO_E_BADARG;   // the second AI's rows need their own replies: elfsp_end_step2
  return elfsp_end_step2(sp, pis, pi_stride_floats, vs, rvs, stream);
}

int elfsp_set_request2(ElfSelfPlay* sp, const ElfSpRequest* q) { return elfsp_set_request3(sp, q, nullptr); }

int elfsp_set_request3(ElfSelfPlay* sp, const ElfSpRequest* q, const ElfTsOptions* mcts_opt) {
  if (!sp || !q) return ELFGO_E_BADARG;
  if (q->white_ver >= 0 && q->black_ver < 0) return ELFGO_E_BADARG;
  const ElfTsOptions ts = mcts_opt ? *mcts_opt : sp_ts_of(sp->opt0);
  if (ts.num_threads < 1 || ts.num_rollouts_per_batch < 1 || ts.num_rollouts_per_thread < 1 || ts.pick_method < ELFSP_PICK_MOST_VISITED ||
      ts.pick_method > ELFSP_PICK_UNIFORM_RANDOM || (int64_t)ts.num_rollouts_per_batch * ts.num_threads > elfmcts_max_rollouts_per_step())
    return ELFGO_E_BADARG;
  if (q->white_ver >= 0) {
    // the second AI's tree pool must fit the leaf table of a step
    ElfSpOptions o2 = sp->opt;
    sp_ts_into(ts, &o2);
    ElfMctsOptions mo; int rpt;
    sp_pool_options(o2, 1, &mo, &rpt);
    if ((int64_t)mo.num_rollouts_per_batch * mo.num_threads > elfmcts_max_rollouts_per_step() || rpt <= 0) return ELFGO_E_BADARG;
  }
  SpRequest r;
  r.ts = ts;
  r.black_ver = q->black_ver < 0 ? -1 : q->black_ver;
  r.white_ver = q->black_ver < 0 ? -1 : (q->white_ver < 0 ? -1 : q->white_ver);
  r.black_thres = q->black_resign_thres; r.white_thres = q->white_resign_thres; r.never_resign_prob = q->never_resign_prob;
  r.async = q->async != 0; r.player_swap = q->player_swap != 0;
  r.thread_used = q->num_game_thread_used;
  r.client_type = q->client_type != 0 ? q->client_type : 1;
  if (!sp->explicit_request && sp->n_steps == 0 && !sp->step_open && !sp->mailbox.empty() && sp->cur_done) {
    // nothing has been played yet: this request replaces the implicit one of elfsp_create (the reference's games wait for their
    // first request, game_selfplay.cc:277-279)
    sp->mailbox.clear();
  }
  sp->explicit_request = true;
  // the dispatcher forwards a message only if it differs from the last one (dispatcher.h:92-100)
  const SpRequest& last = sp->mailbox.empty() ? sp->cur : sp->mailbox.back();
  const bool have_last = !sp->mailbox.empty() || sp->cur.id != 0;
  if (have_last && last.black_ver == r.black_ver && last.white_ver == r.white_ver && last.black_thres == r.black_thres &&
      last.white_thres == r.white_thres && last.never_resign_prob == r.never_resign_prob && last.async == r.async &&
      last.player_swap == r.player_swap && last.thread_used == r.thread_used && last.client_type == r.client_type && sp_ts_equal(last.ts, r.ts))
    return 0;
  sp_push_request(sp, r);
  if (!sp->step_open) {                              // games between two searches may look at their mailbox now
    DevGuard _dg(sp->eng->device);
    return sp_poll_requests(sp);
  }
  return 0;
}

int elfsp_set_request(ElfSelfPlay* sp, int64_t black_ver, int64_t white_ver, float resign_thres, float never_resign_prob, int async) {
  if (!sp) return ELFGO_E_BADARG;
  ElfSpRequest q;
  memset(&q, 0, sizeof(q));
  q.black_ver = black_ver; q.white_ver = white_ver;
  q.black_resign_thres = q.white_resign_thres = resign_thres;
  q.never_resign_prob = never_resign_prob;
  q.num_game_thread_used = -1;
  q.async = async;
  return elfsp_set_request2(sp, &q);
}

int elfsp_take_game_starts(ElfSelfPlay* sp, int64_t* black_ver, int64_t* white_ver) {
  if (!sp) return ELFGO_E_BADARG;
  const int n = sp->game_starts;
  sp->game_starts = 0;
  if (black_ver) *black_ver = sp->start_black;
  if (white_ver) *white_ver = sp->start_white;
  return n;
}

// GoStateExt::getThreadState (go_state_ext.h:149-157) of every game; host-only
int elfsp_thread_states(const ElfSelfPlay* sp, ElfThreadState* out, int capacity) {
  if (!sp || !out || capacity < sp->G) return ELFGO_E_BADARG;
  for (int g = 0; g < sp->G; ++g) {
    const SpGame& gm = sp->games[g];
    ElfThreadState& t = out[g];
    t.thread_id = sp->opt.game_idx_base + g;
    t.seq = gm.seq;
    t.move_idx = gm.ply - 1;
    t.reserved = 0;
    t.black = gm.req.black_ver;
    t.white = gm.req.white_ver;
  }
  return sp->G;
}

int elfsp_set_pick_seed(ElfSelfPlay* sp, uint32_t seed) {
  if (!sp) return ELFGO_E_BADARG;
  sp->pick_rng.seed((std::mt19937::result_type)seed);
  return 0;
}

// host-only progress counters (no device synchronisation): 0 searches finished (moves played or games resigned), 1 games finished,
// 2 searches open, 3 steps, 4 games waiting for a request, 5 games waiting at a request barrier
int elfsp_progress(const ElfSelfPlay* sp, int64_t* out6) {
  if (!sp || !out6) return ELFGO_E_BADARG;
  int open = 0, waiting = 0, barrier = 0;
  for (const SpGame& gm : sp->games) { open += gm.ai >= 0; waiting += gm.phase == PH_WAIT; barrier += gm.phase == PH_BARRIER; }
  out6[0] = sp->n_moves; out6[1] = sp->n_games; out6[2] = open; out6[3] = sp->n_steps; out6[4] = waiting; out6[5] = barrier;
  return 0;
}

// which AI of game g is searching now (ELFSP_ACTOR_*), -1 if none; host-only
int elfsp_game_actor(const ElfSelfPlay* sp, int game) {
  if (!sp || game < 0 || game >= sp->G) return ELFGO_E_BADARG;
  return sp->games[game].ai;
}

static bool sp_any_search_open(const ElfSelfPlay* sp, const int32_t* games, int n) {
  if (sp->step_open) return true;
  if (!games) { for (const SpGame& gm : sp->games) if (gm.ai >= 0) return true; return false; }
  for (int j = 0; j < n; ++j) if (sp->games[games[j]].ai >= 0) return true;
  return false;
}

// the human half of GoGameSelfPlay::act (game_selfplay.cc:290-330): an externally chosen move is forwarded on the game board,
// the trees follow (MCTSAI_T::align_state -> advanceMoves, mcts.h:141-167).  moves_host[g] < 0 = no move.
int elfsp_play(ElfSelfPlay* sp, const int32_t* moves_host, void* stream) {
  if (!sp || !moves_host) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  const int G = sp->G;
  std::vector<int32_t> ids, mv;
  for (int g = 0; g < G; ++g) if (moves_host[g] >= 0) { ids.push_back(g); mv.push_back(moves_host[g]); }
  if (ids.empty()) return 0;
  if (sp_any_search_open(sp, ids.data(), (int)ids.size())) return ELFGO_E_BADARG;
  SPCHK(sp_poll_requests(sp));   // the games of a fresh context start with their first request
  const int k = (int)ids.size();
  HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->d_moves, mv.data(), 4 * k, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_forward(sp->eng, sp->d_ids, sp->d_moves, k, sp->d_ok, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_ok.data(), sp->d_ok, k, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  // "Invalid move ... please try again" (:323-327): refused moves leave their game untouched
  std::vector<int32_t> adv(G, -1);
  int bad = 0;
  for (int j = 0; j < k; ++j) {
    if (sp->h_ok[j] != 1) { bad++; continue; }
    adv[ids[j]] = mv[j];
    sp->games[ids[j]].online_counter++;
  }
  if (sp->opt.persistent_tree) {
    HIPCHK(hipMemcpyAsync(sp->d_moves, adv.data(), 4 * G, hipMemcpyHostToDevice, sp->stream));
    for (int a = 0; a < 2; ++a)
      if (sp->pool[a].mcts) SPCHK(elfmcts_advance(sp->pool[a].mcts, sp->d_moves, sp->stream));
  }
  SPCHK(elfgo_info(sp->eng, nullptr, G, sp->d_binfo, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_binfo.data(), sp->d_binfo, sizeof(int32_t) * G * ELFGO_INFO_WORDS, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  std::vector<int32_t> two_pass;
  for (int g = 0; g < G; ++g) {
    if (adv[g] < 0) continue;
    SpGame& gm = sp->games[g];
    const int32_t* bi = &sp->h_binfo[g * ELFGO_INFO_WORDS];
    gm.ply = bi[0];
    if (sp->opt.keep_records > 0) gm.rec.moves.push_back((uint16_t)adv[g]);
    if (bi[2] == M_PASS && bi[3] == M_PASS) two_pass.push_back(g);   // "If the human opponent pass, we pass as well" :319-322
  }
  sp->n_moves += k - bad;
  if (!two_pass.empty()) SPCHK(elfsp_finish(sp, two_pass.data(), (int)two_pass.size(), ELFSP_FR_TWO_PASSES, stream));
  return bad ? ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD : 0;
}

// GameOptions.preload_sgf / preload_sgf_move_to (GoGameSelfPlay::restart, game_selfplay.cc:202-219): every game follows the
// given move list -- the first move_to moves are forwarded when the games (re)start, afterwards each search's move is replaced by
// the next SGF move (:392-405) and the game is finished (FR_MAX_STEP) by the search that finds the list exhausted.
int elfsp_preload(ElfSelfPlay* sp, const uint16_t* moves_host, int n, int move_to, void* stream) {
  if (!sp || n < 0 || (n > 0 && !moves_host) || sp_any_search_open(sp, nullptr, 0) || sp->n_moves != 0) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  sp->sgf.assign(moves_host, moves_host + n);
  sp->sgf_move_to = move_to;
  // games that have already started (their request arrived before the preload) are at the empty board: forward now; the others
  // forward when their first request starts them
  std::vector<int32_t> ids;
  for (int g = 0; g < sp->G; ++g) if (sp->games[g].phase != PH_WAIT) ids.push_back(g);
  return sp_forward_preload(sp, ids);
}

// finish_game(reason) + restart (game_selfplay.cc:121-149): the listed games are scored (setFinalValue, go_state_ext.h:76-103: FR_RESIGN
// = the side to move loses, every other reason = GoState::evaluate(komi)), leave their record and start over from the empty board
int elfsp_finish(ElfSelfPlay* sp, const int32_t* games_host, int n, int reason, void* stream) {
  if (!sp || n < 0 || n > sp->G || (n > 0 && !games_host)) return ELFGO_E_BADARG;
  if (reason < ELFSP_FR_RESIGN || reason > ELFSP_FR_ILLEGAL) return ELFGO_E_BADARG;
  if (n == 0) return 0;
  for (int j = 0; j < n; ++j) if (games_host[j] < 0 || games_host[j] >= sp->G) return ELFGO_E_BADARG;
  if (sp_any_search_open(sp, games_host, n)) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  SPCHK(sp_poll_requests(sp));
  HIPCHK(hipMemcpyAsync(sp->d_ids, games_host, 4 * n, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, n, sp->opt.mcts.komi, sp->d_val, sp->stream));   // setFinalValue: evaluate(komi)
  HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * n, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  std::vector<int32_t> ids(games_host, games_host + n);
  for (int j = 0; j < n; ++j) {
    SpGame& gm = sp->games[games_host[j]];
    const float fv = sp_final_value(sp, gm, reason, sp->h_val[j]);
    sp->sum_final += fv;
    sp_finish_record(sp, games_host[j], fv, gm.ply);
    gm.online_counter++;
  }
  return sp_restart_finished(sp, ids);
}

int elfsp_restart(ElfSelfPlay* sp, const int32_t* games_host, int n, void* stream) {
  return elfsp_finish(sp, games_host, n, ELFSP_FR_CLEAR, stream);
}

int elfsp_take_finished(ElfSelfPlay* sp, float* out_host, int cap) {
  if (!sp || cap < 0 || (cap > 0 && !out_host)) return ELFGO_E_BADARG;
  int k = 0;
  while (k < cap && !sp->finished_values.empty()) {
    out_host[k++] = sp->finished_values.front();
    sp->finished_values.pop_front();
  }
  return k;
}

// GoGameSelfPlay::getLastScore (GoStateExt::getLastGameFinalValue): final value of the last finished game of each game slot
int elfsp_last_score(const ElfSelfPlay* sp, float* out_host) {
  if (!sp || !out_host) return ELFGO_E_BADARG;
  for (int g = 0; g < sp->G; ++g) out_host[g] = sp->games[g].last_final;
  return 0;
}

int elfsp_last_moves(const ElfSelfPlay* sp, int32_t* out_host) {
  if (!sp || !out_host) return ELFGO_E_BADARG;
  for (int g = 0; g < sp->G; ++g) out_host[g] = sp->games[g].last_move;
  return 0;
}

int elfsp_records_pending(const ElfSelfPlay* sp) { return sp ? (int)sp->records.size() : ELFGO_E_BADARG; }

int elfsp_pop_record(ElfSelfPlay* sp, char* buf, size_t cap, size_t* len) {
  if (!sp || !len) return ELFGO_E_BADARG;
  if (sp->records.empty()) { *len = 0; return 0; }
  const std::string& r = sp->records.front();
  *len = r.size();
  if (!buf || cap < r.size() + 1) return ELFGO_E_BADSIZE;   // *len tells the caller what to allocate; nothing is consumed
  memcpy(buf, r.data(), r.size());
  buf[r.size()] = 0;
  sp->records.pop_front();
  return 0;
}

int64_t elfsp_games_finished(const ElfSelfPlay* sp) { return sp ? sp->n_games : -1; }

int elfsp_stats(ElfSelfPlay* sp, int64_t* out) {
  if (!sp || !out) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  out[8] = 0;
  uint64_t rows = 0;
  for (int a = 0; a < 2; ++a) {
    SpPool& p = sp->pool[a];
    if (!p.mcts) continue;
    int64_t nv = 0;
    SPCHK(elfmcts_node_visits(p.mcts, &nv));      // synchronises the device
    out[8] += nv;
    uint64_t total_rows = 0;
    HIPCHK(hipMemcpy(&total_rows, p.d_counts + 2, 8, hipMemcpyDeviceToHost));
    rows += total_rows;
  }
  sp->n_rows = (int64_t)rows;
  out[0] = sp->n_moves; out[1] = sp->n_games; out[2] = sp->n_rollouts; out[3] = sp->n_rows; out[4] = sp->n_steps;
  out[5] = (int64_t)sp->log_search.size(); out[6] = sp->pool[0].steps_per_move; out[7] = sp->games[0].ai >= 0 ? sp->games[0].step : 0;
  out[9] = sp->boundary_ns; out[10] = sp->n_boundaries; out[11] = sp->boundary_wait_ns;
  return 0;
}

int elfsp_search_log(const ElfSelfPlay* sp, int first, int n, ElfSpSearch* rec, int32_t* coord, int32_t* visits, float* prior, float* reward) {
  if (!sp || first < 0 || n < 0 || first + n > (int)sp->log_search.size()) return ELFGO_E_BADARG;
  static_assert(sizeof(ElfSpSearch) == sizeof(ElfSpSearchRec), "ElfSpSearch layout");
  const size_t NE = sp->NE;
  if (rec) memcpy(rec, sp->log_search.data() + first, sizeof(ElfSpSearchRec) * n);
  if (coord) memcpy(coord, sp->log_coord.data() + first * NE, 4 * NE * n);
  if (visits) memcpy(visits, sp->log_visits.data() + first * NE, 4 * NE * n);
  if (prior) memcpy(prior, sp->log_prior.data() + first * NE, 4 * NE * n);
  if (reward) memcpy(reward, sp->log_reward.data() + first * NE, 4 * NE * n);
  return 0;
}

}  // extern "C"
