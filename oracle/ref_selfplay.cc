// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// The REAL reference self-play stack, compiled in place from /root/reference by oracle/Makefile into
// oracle/_ref/libelfsp{19,9}.so:  elf::Context batcher (elf/base/context.h) + ThreadedDispatcher +
// GoGameSelfPlay (elfgames/go/common/game_selfplay.cc) + MCTSGoAI / MCTSActor (elfgames/go/mcts/mcts.h)
// + the generic tree search (elf/ai/tree_search/*.h).  This file only plays the role of the Python
// half (GCWrapper, src_py/elf/utils_elf.py:291-437, and GameContext, inference/game_context.h:31-69):
// it allocates the batch buffers, serves "actor_black"/"actor_white" batches with a net function, and
// records what GameNotifierBase::OnMCTSResult (common/notifier.h:13) reports after every search.
// Nothing from the reference is copied into this repository.
//
// Used (a) to generate tests/golden/mcts_*.npz (oracle/gen_golden_mcts.py), (b) as the CPU baseline of
// kind "reference" for MCTS rollouts/s in bench.py, (c) for the self-play Record JSON of finished games
// (GoStateExt::dumpRecord via GameNotifierBase::OnGameEnd) and the trainer's "train" batch extractors
// (reftrain_*, the real GoStateExtOffline + GoFeature functions) behind tests/golden/records_*.{json,npz}.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <bits/stdc++.h>   // every std header first, so that the access hack below only touches reference headers
#include <nlohmann/json.hpp>
#include "elf/logging/IndexedLoggerFactory.h"
#define private public   // test infrastructure: reach GoStateExtOffline::_bf / _state to set a chosen D4 code and ply
#include "elfgames/go/common/go_state_ext.h"
#undef private
#include "elf/base/context.h"
#include "elf/base/dispatcher.h"
#include "elfgames/go/common/dispatcher_callback.h"
#include "elfgames/go/common/game_feature.h"
#include "elfgames/go/common/game_selfplay.h"
#include "elfgames/go/common/go_game_specific.h"
#include "elfgames/go/train/game_train.h"

#include "stub_net.h"

extern "C" {

// Mirrors the option structs the reference exposes to Python (TSOptions tree_search_options.h:77-229,
// GameOptions go_game_specific.h:16-268); plain C so ctypes can fill it.
struct RefSpConfig {
  int32_t num_games;            // ContextOptions.num_games
  int32_t batchsize;            // ContextOptions.batchsize
  int32_t mcts_threads;         // TSOptions.num_threads
  int32_t rollouts_per_thread;  // TSOptions.num_rollouts_per_thread
  int32_t rollouts_per_batch;   // TSOptions.num_rollouts_per_batch
  int32_t virtual_loss;         // TSOptions.virtual_loss
  int32_t persistent_tree;      // TSOptions.persistent_tree
  int32_t use_prior;            // alg_opt.use_prior
  int32_t unexplored_q_zero, root_unexplored_q_zero;
  float c_puct, root_epsilon, root_alpha;
  uint32_t seed;                // GameOptions.seed (game idx i uses seed + i when num_games > 1: see below)
  float komi;
  int32_t ply_pass_enabled, policy_distri_cutoff, move_cutoff;
  float resign_thres;           // ClientCtrl black/white_resign_thres (setRequest)
  float never_resign_prob;      // ClientCtrl.never_resign_prob
  // stub net
  uint32_t net_salt;
  int32_t net_tie_levels;
  // stop after this many searches (summed over games) have been recorded
  int32_t max_searches;
  int32_t timeout_usec;         // batch collector timeout (game.py:365-402 --gpu path uses 10)
  // ---- round 3: evaluation games (two AIs), pick methods, policy-only play, idle game threads
  int32_t black_ver;            // MsgRequest.vers.black_ver (replies carry it in "rv")
  int32_t white_ver;            // MsgRequest.vers.white_ver: -1 = self-play, >= 0 = a second AI for White
  int32_t player_swap;          // ClientCtrl.player_swap
  float white_puct;             // GameOptions.white_puct / white_mcts_rollout_per_batch / white_mcts_rollout_per_thread (<= 0: no override)
  int32_t white_rollouts_per_batch, white_rollouts_per_thread;
  uint32_t white_net_salt;      // stub net of the "actor_white" group
  int32_t pick_method;          // 0 most_visited, 1 strongest_prior, 2 uniform_random
  int32_t black_policy_only, white_policy_only;   // GameOptions.*_use_policy_network_only
  int32_t thread_used;          // ClientCtrl.num_game_thread_used (0 = num_games, the harness' historical value; -1 = all)
  // a second request while the games play (num_games = 1 for a reproducible arrival): sent when `req2_after_searches` searches
  // have been captured and the next search has its first batch waiting for a reply -- the harness then sleeps 1.5 s before it
  // replies, so that the dispatcher thread (500 ms poll, elf/base/dispatcher.h:22) has put the request into the game's mailbox
  // before the search can finish: the game receives it at its next mailbox look, i.e. at the top of its next fifth act
  int32_t req2_after_searches;  // 0 = no second request
  int32_t req2_black_ver;
  int32_t req2_async;           // ClientCtrl.async of the second request
  int32_t cheat_eval_new_model_wins_half, cheat_selfplay_random_result;   // GameOptions.cheat_* (finish_game, game_selfplay.cc:122-129)
  // GameOptions.mode = "online" (one game): the "human_actor" prompts of GoGameSelfPlay::act :290-330 are answered from the script
  // of refsp_set_human_script ("a" = an action index or ACTION_SKIP / PASS / RESIGN / CLEAR); the run ends at the prompt after the
  // last scripted answer; every prompt's "s" rows are kept (refsp_last_prompts)
  int32_t online;
  int32_t following_pass;       // GameOptions.following_pass (mcts_update_info :104-111)
  int32_t net_value_on;         // stub net: V = net_value for every row (a position the AI is sure about)
  float net_value;
  // the second request's own ModelPair.mcts_opt (the reference's server dictates the search options: ctrl_selfplay.h:426, and for
  // evaluation games ctrl_eval.h:227-237 with the noise and both q_zero flags off) and white_ver
  int32_t req2_ts;              // 1 = the fields below replace the context's TSOptions in the second request
  int32_t req2_rollouts_per_thread, req2_rollouts_per_batch;
  float req2_c_puct, req2_root_epsilon, req2_root_alpha;
  int32_t req2_unexplored_q_zero, req2_root_unexplored_q_zero;
  int32_t req2_white_ver;       // white_ver of the second request (-1 = self-play)
};

// One record per finished search (MCTSAI_T::act), in completion order.
struct RefSpSearch {
  int32_t game;          // game index (thread)
  int32_t move_played;   // Coord passed to OnMCTSResult: after mcts_make_diverse_move / mcts_update_info
  int32_t best_action;   // MCTSResult.best_action (most visited)
  int32_t total_visits;  // MCTSResult.total_visits
  int32_t n_edges;
  float root_value;      // MCTSResult.root_value
  float max_score;
  int32_t pad;
};

typedef void (*refsp_net_fn)(const float* s, int batch, float* pi, float* v, void* user);

}  // extern "C"

namespace {

struct Capture : public GameNotifierBase {
  std::mutex m;
  int max_searches = 0;
  std::vector<RefSpSearch> searches;
  std::vector<int32_t> coord, visits, child;   // [search][BOARD_NUM_ACTION], root edges in ITERATION order
  std::vector<float> prior, reward;
  std::atomic<int> count{0};
  int game_of_thread(const std::thread::id&) { return 0; }

  int total_calls = 0;   // every search, recorded or not
  void OnMCTSResult(Coord c, const MCTSResult& r) override { add(0, c, r); }
  void add(int game, Coord c, const MCTSResult& r) {
    std::lock_guard<std::mutex> l(m);
    ++total_calls;
    if ((int)searches.size() >= max_searches) return;
    RefSpSearch s;
    memset(&s, 0, sizeof(s));
    s.game = game;
    s.move_played = c;
    s.best_action = r.best_action;
    s.total_visits = r.total_visits;
    s.n_edges = (int)r.action_edge_pairs.size();
    s.root_value = r.root_value;
    s.max_score = r.max_score;
    searches.push_back(s);
    const size_t base = coord.size();
    coord.resize(base + BOARD_NUM_ACTION, -1);
    visits.resize(base + BOARD_NUM_ACTION, 0);
    child.resize(base + BOARD_NUM_ACTION, -1);
    prior.resize(base + BOARD_NUM_ACTION, 0.f);
    reward.resize(base + BOARD_NUM_ACTION, 0.f);
    for (size_t i = 0; i < r.action_edge_pairs.size(); ++i) {
      coord[base + i] = r.action_edge_pairs[i].first;
      visits[base + i] = r.action_edge_pairs[i].second.num_visits;
      child[base + i] = r.action_edge_pairs[i].second.child_node;
      prior[base + i] = r.action_edge_pairs[i].second.prior_probability;
      reward[base + i] = r.action_edge_pairs[i].second.reward;
    }
    count++;
  }
  std::vector<std::string> records;   // Record JSON of every finished game, in completion order
  std::vector<std::string> sgfs;      // GoStateExt::dumpSgf of the same games
  void OnGameEnd(const GoStateExt& s) override {
    std::lock_guard<std::mutex> l(m);
    // searches past max_searches run while the driver is already shutting the context down (replies are FAILED batches):
    // a game whose last search is one of those is not a valid reference game
    if (total_calls > max_searches) return;
    json j;
    s.dumpRecord().setJsonFields(j);
    records.push_back(j.dump());
    // what finish_game writes when GameOptions.dump_record_prefix is set (:133-135 -> GoStateExt::dumpSgf, go_state_ext.cc:26-82)
    const ThreadState ts = s.getThreadState();
    const std::string fname = "game_" + std::to_string(ts.thread_id) + "_" + std::to_string(s.seq()) + "_" +
                              (s.state().getFinalValue() > 0 ? "B" : "W") + ".sgf";
    sgfs.push_back(s.dumpSgf(fname));
  }
};

// one notifier per game thread: GameNotifierBase callbacks carry no game index (common/notifier.h:13), the search log needs one
struct GameCapture : public GameNotifierBase {
  Capture* cap = nullptr;
  int game = 0;
  void OnMCTSResult(Coord c, const MCTSResult& r) override { cap->add(game, c, r); }
  void OnGameEnd(const GoStateExt& s) override { cap->OnGameEnd(s); }
};

// ---- the search threads' turnstile (libelfsp*_ts.so only: see oracle/Makefile, target ts_patched) --------------------------------
// The reference's T search threads race on the shared tree (tree_search.h:345-368), so a T > 1 search has no single answer.  The
// build `libelfsp{19,9}_ts.so` compiles a COPY of elf/ai/tree_search/tree_search.h (made at build time inside oracle/_ref/, never
// committed) in which four calls of elf_ts_hook() have been inserted into TreeSearchSingleThreadT::batch_rollouts -- no other line
// differs, no arithmetic and no data structure is touched.  With the turnstile on, the hooks make the threads of one TreeSearchT
// take turns in ONE fixed order per round of batch_rollouts calls:
//     D_0 .. D_{T-1}   thread t runs its K single_rollouts + requestEvaluation loop (:205-233) after thread t-1 has finished its own
//     (evaluate)       every thread blocks in actor.evaluate on the batcher as always (:236-237); replies arrive in any order
//     B_0 .. B_{T-1}   thread t runs setEvaluation + its backups (:239-259) after thread t-1 has finished its own
// and the next round's D_0 starts after B_{T-1}.  Every such run is one of the interleavings the unpatched reference can produce;
// it is the one elf_amd's engine implements (elf_amd/csrc/mcts.cuh).  With the turnstile off (the default, and always for T = 1)
// the hooks return at once.
struct TsTurn {
  std::mutex m;
  std::condition_variable cv;
  int slot = 0;            // 0..T-1: D_t may run; T..2T-1: B_{slot-T} may run
};
std::mutex g_ts_m;
std::map<const void*, std::unique_ptr<TsTurn>> g_ts;   // one per TreeSearchT (keyed by the address of its TSOptions member)
std::atomic<int> g_ts_on{0};

std::string g_preload_sgf;     // GameOptions.preload_sgf for the next refsp_run ("" = none)
int g_preload_move_to = -1;
std::string g_last_records;   // JSON array text of the records of the last refsp_run
std::string g_last_sgfs;      // JSON array of the dumpSgf texts of the same games
int64_t g_white_rows = 0;     // rows served to the "actor_white" group by the last refsp_run
std::vector<int64_t> g_human_script;   // answers to the "human_actor" prompts of an online run
std::vector<uint8_t> g_prompts;         // "s" of every prompt of the last online run, one byte per plane point
int64_t g_n_prompts = 0;
int64_t g_game_starts = 0;    // "game_start" batches of the last refsp_run
int64_t g_start_vers[8] = {0};   // black_ver of the first 8 of them

struct Buffers {
  std::vector<float> s, pi, V;
  std::vector<int64_t> a, rv, black_ver, white_ver;
};

}  // namespace

extern "C" {

int refsp_board_size() { return BOARD_SIZE; }

// 1 = this library was built from the turnstile copy of tree_search.h (libelfsp*_ts.so)
int refsp_has_turnstile() {
#ifdef ELF_TS_TURNSTILE
  return 1;
#else
  return 0;
#endif
}
// 1 = this library was built from the copy of tree_search.h whose batch_rollouts backs the leaves up in first-occurrence order
// (libelfsp*_h2.so, oracle/Makefile: SURVEY.md hazard H2)
int refsp_has_canonical_backup() {
#ifdef ELF_H2_CANONICAL
  return 1;
#else
  return 0;
#endif
}
// switch the turnstile on / off for the following refsp_run calls (no effect in the stock build, which has no hooks)
void refsp_set_turnstile(int on) { g_ts_on.store(on ? 1 : 0); }

// called from the four hook lines of the patched batch_rollouts: where = 0 before the descents, 1 before actor.evaluate,
// 2 after actor.evaluate, 3 at the end of the backups; key = the address of the thread's TSOptions (one per TreeSearchT)
void elf_ts_hook(int where, int thread_id, const void* key, int num_threads) {
  if (!g_ts_on.load() || num_threads <= 1) return;
  TsTurn* t;
  {
    std::lock_guard<std::mutex> l(g_ts_m);
    auto& p = g_ts[key];
    if (!p) p.reset(new TsTurn());
    t = p.get();
  }
  const int T = num_threads;
  std::unique_lock<std::mutex> l(t->m);
  switch (where) {
    case 0: t->cv.wait(l, [&] { return t->slot == thread_id || !g_ts_on.load(); }); break;
    case 1: t->slot = thread_id + 1; t->cv.notify_all(); break;
    case 2: t->cv.wait(l, [&] { return t->slot == T + thread_id || !g_ts_on.load(); }); break;
    default: t->slot = (T + thread_id + 1) % (2 * T); t->cv.notify_all(); break;
  }
}

// Runs reference self-play until cfg->max_searches searches have been captured.
// out arrays sized [max_searches] / [max_searches][BOARD_NUM_ACTION]. Returns the number captured
// (<0 on error).  stats[0] = batches served, stats[1] = rows served, stats[2] = wall seconds * 1e6.
int refsp_run(const RefSpConfig* cfg, refsp_net_fn net, void* net_user, RefSpSearch* out_search, int32_t* out_coord,
              int32_t* out_visits, float* out_prior, float* out_reward, int64_t* stats) {
  try {
    ContextOptions co;
    co.num_games = cfg->num_games;
    co.batchsize = cfg->batchsize;
    auto& ts = co.mcts_options;
    ts.num_threads = cfg->mcts_threads;
    ts.num_rollouts_per_thread = cfg->rollouts_per_thread;
    ts.num_rollouts_per_batch = cfg->rollouts_per_batch;
    ts.virtual_loss = cfg->virtual_loss;
    ts.persistent_tree = cfg->persistent_tree != 0;
    ts.root_epsilon = cfg->root_epsilon;
    ts.root_alpha = cfg->root_alpha;
    ts.alg_opt.use_prior = cfg->use_prior != 0;
    ts.alg_opt.c_puct = cfg->c_puct;
    ts.alg_opt.unexplored_q_zero = cfg->unexplored_q_zero != 0;
    ts.alg_opt.root_unexplored_q_zero = cfg->root_unexplored_q_zero != 0;
    ts.pick_method = cfg->pick_method == 1 ? "strongest_prior" : cfg->pick_method == 2 ? "uniform_random" : "most_visited";

    GameOptions opt;
    opt.mode = cfg->online ? "online" : "selfplay";
    opt.following_pass = cfg->following_pass != 0;
    opt.seed = cfg->seed;
    opt.komi = cfg->komi;
    opt.ply_pass_enabled = cfg->ply_pass_enabled;
    opt.policy_distri_cutoff = cfg->policy_distri_cutoff;
    opt.move_cutoff = cfg->move_cutoff;
    opt.use_mcts = true;
    opt.port = 0;
    opt.preload_sgf = g_preload_sgf;
    opt.preload_sgf_move_to = g_preload_move_to;
    opt.white_puct = cfg->white_puct > 0.0f ? cfg->white_puct : -1.0f;
    opt.cheat_eval_new_model_wins_half = cfg->cheat_eval_new_model_wins_half != 0;
    opt.cheat_selfplay_random_result = cfg->cheat_selfplay_random_result != 0;
    opt.white_mcts_rollout_per_batch = cfg->white_rollouts_per_batch > 0 ? cfg->white_rollouts_per_batch : -1;
    opt.white_mcts_rollout_per_thread = cfg->white_rollouts_per_thread > 0 ? cfg->white_rollouts_per_thread : -1;
    opt.black_use_policy_network_only = cfg->black_policy_only != 0;
    opt.white_use_policy_network_only = cfg->white_policy_only != 0;

    const int n = cfg->num_games, B = cfg->batchsize;
    elf::Context ctx;
    GoFeature gf(opt);
    gf.registerExtractor(B, ctx.getExtractor());

    using ThreadedDispatcher = GoGameSelfPlay::ThreadedDispatcher;
    Ctrl ctrl;
    ThreadedDispatcher disp(ctrl, n);
    DispatcherCallback dcb(&disp, ctx.getClient());

    Capture cap;
    cap.max_searches = cfg->max_searches;
    std::vector<std::unique_ptr<GoGameSelfPlay>> games;
    std::vector<GameCapture> gcaps(n);
    for (int i = 0; i < n; ++i) {
      GameOptions gopt = opt;
      if (n > 1 && cfg->seed != 0) gopt.seed = cfg->seed + i;   // distinct games (the reference proper seeds them all alike)
      gcaps[i].cap = &cap; gcaps[i].game = i;
      games.emplace_back(new GoGameSelfPlay(i, ctx.getClient(), co, gopt, &disp, &gcaps[i]));
    }
    ctx.setStartCallback(n, [&](int i, elf::GameClient*) {
      disp.RegGame(i);
      games[i]->mainLoop();
    });

    // what Allocator.spec2batches does (utils_elf.py:59-99) with the desc of game.py:365-402, num_recv = 2
    struct Group { const char* name; std::vector<std::string> keys; int bs; int timeout; };
    std::vector<Group> groups = {
        {"actor_black", {"s", "pi", "V", "a", "rv"}, B, cfg->timeout_usec},
        {"actor_white", {"s", "pi", "V", "a", "rv"}, B, cfg->timeout_usec},
        {"game_start", {"black_ver", "white_ver"}, 1, 0},
        {"game_end", {}, 1, 0},
    };
    // game.py:366-378.  A finite collector timeout, unlike game.py's: a collector that waits for game messages without one never sees
    // Context::stop's PREPARE_TO_STOP (self-play wakes every collector with dummy messages, game_selfplay.cc:334-357; online mode
    // has no such path for "human_actor"), and this harness has to stop
    if (cfg->online) groups = {{"human_actor", {"s", "pi", "V", "a"}, 1, 10}, {"actor_black", {"s", "pi", "V", "a", "rv"}, B, cfg->timeout_usec}};
    std::vector<Buffers> bufs;
    bufs.reserve(groups.size() * 2 + 1);
    std::vector<int> idx2buf;
    for (auto& g : groups) {
      for (int r = 0; r < 2; ++r) {
        auto smo = ctx.createSharedMemOptions(g.name, g.bs);
        smo.setTimeout(g.timeout);
        elf::SharedMem& sm = ctx.allocateSharedMem(smo, g.keys);
        bufs.emplace_back();
        Buffers& b = bufs.back();
        const int idx = sm.getSharedMemOptions().getIdx();
        if ((int)idx2buf.size() <= idx) idx2buf.resize(idx + 1, -1);
        idx2buf[idx] = (int)bufs.size() - 1;
        for (const auto& k : g.keys) {
          elf::AnyP* p = sm[k];
          const auto& f = p->field();
          const size_t ne = f.getSize().nelement();
          void* addr = nullptr;
          if (k == "s") { b.s.assign(ne, 0.f); addr = b.s.data(); }
          else if (k == "pi") { b.pi.assign(ne, 0.f); addr = b.pi.data(); }
          else if (k == "V") { b.V.assign(ne, 0.f); addr = b.V.data(); }
          else if (k == "a") { b.a.assign(ne, 0); addr = b.a.data(); }
          else if (k == "rv") { b.rv.assign(ne, 0); addr = b.rv.data(); }
          else if (k == "black_ver") { b.black_ver.assign(ne, 0); addr = b.black_ver.data(); }
          else if (k == "white_ver") { b.white_ver.assign(ne, 0); addr = b.white_ver.data(); }
          p->setAddress((uint64_t)addr, f.getSize().getContinuousStrides(f.getSizeOfType()).vec());
        }
      }
    }

    ctx.start();
    {
      // Client::setRequest (train/distri_client.h:318-331) / GameContext::setRequest (inference/game_context.h:76-88)
      MsgRequest req;
      req.vers.black_ver = cfg->black_ver;
      req.vers.white_ver = cfg->white_ver;   // -1 = self-play: one AI plays both colours
      req.client_ctrl.player_swap = cfg->player_swap != 0;
      req.vers.mcts_opt = co.mcts_options;
      req.client_ctrl.black_resign_thres = cfg->resign_thres;
      req.client_ctrl.white_resign_thres = cfg->resign_thres;
      req.client_ctrl.never_resign_prob = cfg->never_resign_prob;
      req.client_ctrl.num_game_thread_used = cfg->thread_used == 0 ? n : cfg->thread_used;
      disp.sendToThread(req);
    }

    int64_t batches = 0, rows = 0;
    g_white_rows = 0;
    g_game_starts = 0;
    int64_t cur_black_rv = cfg->black_ver;   // the version the "actor_black" model answers with: follows the game_start batches
    int64_t cur_white_rv = cfg->white_ver;
    bool req2_sent = false;
    const auto t0 = std::chrono::steady_clock::now();
    const int NA = BOARD_NUM_ACTION;
    g_prompts.clear();
    g_n_prompts = 0;
    while (cap.count.load() < cfg->max_searches) {
      const elf::SharedMem* sm = ctx.wait(100000);
      if (sm == nullptr) { continue; }
      const int eb = sm->getEffectiveBatchSize();
      const auto& smo = sm->getSharedMemOptions();
      Buffers& b = bufs[idx2buf[smo.getIdx()]];
      const std::string& label = smo.getLabel();
      if (label == "game_start") {           // Python loads the models the batch names (selfplay.py game_start callback)
        if (g_game_starts < 8) g_start_vers[g_game_starts] = b.black_ver[0];
        g_game_starts++;
        cur_black_rv = b.black_ver[0];
        cur_white_rv = b.white_ver[0];
      }
      if (label == "human_actor") {
        for (size_t i = 0; i < (size_t)18 * BOARD_SIZE * BOARD_SIZE; ++i) g_prompts.push_back(b.s[i] != 0.f ? 1 : 0);
        if (g_n_prompts >= (int64_t)g_human_script.size()) {   // the prompt after the last answer: the run is over (answered SKIP,
          g_n_prompts++;                                        // so that the game blocks in a search batch as at any other stop)
          { std::lock_guard<std::mutex> l(cap.m); cap.max_searches = (int)cap.searches.size(); }   // what runs during the shutdown is not part of the run
          b.a[0] = SA_SKIP;
          ctx.step();
          break;
        }
        b.a[0] = g_human_script[g_n_prompts++];
        b.V[0] = 0.f;
      }
      if (label == "actor_black" || label == "actor_white") {
        const bool white_group = label == "actor_white";
        if (cfg->req2_after_searches > 0 && !req2_sent && cap.count.load() >= cfg->req2_after_searches) {
          MsgRequest req2;
          req2.vers.black_ver = cfg->req2_black_ver;
          req2.vers.white_ver = cfg->req2_white_ver;
          req2.vers.mcts_opt = co.mcts_options;
          if (cfg->req2_ts) {
            auto& t2 = req2.vers.mcts_opt;
            t2.num_rollouts_per_thread = cfg->req2_rollouts_per_thread; t2.num_rollouts_per_batch = cfg->req2_rollouts_per_batch;
            t2.alg_opt.c_puct = cfg->req2_c_puct; t2.root_epsilon = cfg->req2_root_epsilon; t2.root_alpha = cfg->req2_root_alpha;
            t2.alg_opt.unexplored_q_zero = cfg->req2_unexplored_q_zero != 0;
            t2.alg_opt.root_unexplored_q_zero = cfg->req2_root_unexplored_q_zero != 0;
          }
          req2.client_ctrl.black_resign_thres = cfg->resign_thres;
          req2.client_ctrl.white_resign_thres = cfg->resign_thres;
          req2.client_ctrl.never_resign_prob = cfg->never_resign_prob;
          req2.client_ctrl.num_game_thread_used = cfg->thread_used == 0 ? n : cfg->thread_used;
          req2.client_ctrl.async = cfg->req2_async != 0;
          disp.sendToThread(req2);
          std::this_thread::sleep_for(std::chrono::milliseconds(1500));
          req2_sent = true;
        }
        if (net) net(b.s.data(), eb, b.pi.data(), b.V.data(), net_user);
        else stubnet_eval(b.s.data(), eb, BOARD_SIZE, white_group ? cfg->white_net_salt : cfg->net_salt, cfg->net_tie_levels, b.pi.data(), b.V.data());
        if (cfg->net_value_on) for (int i = 0; i < eb; ++i) b.V[i] = cfg->net_value;
        for (int i = 0; i < eb; ++i) { b.rv[i] = white_group ? cur_white_rv : cur_black_rv; b.a[i] = 0; }
        if (white_group) g_white_rows += eb;
        (void)NA;
        batches++; rows += eb;
      }
      ctx.step();
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) {
      stats[0] = batches; stats[1] = rows;
      stats[2] = std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
    }
    ctx.stop();
    {
      g_last_records = "[";
      for (size_t i = 0; i < cap.records.size(); ++i) g_last_records += (i ? "," : "") + cap.records[i];
      g_last_records += "]";
      json js = json::array();
      for (const auto& t : cap.sgfs) js.push_back(t);
      g_last_sgfs = js.dump();
    }

    const int k = (int)cap.searches.size();
    for (int i = 0; i < k; ++i) out_search[i] = cap.searches[i];
    if (out_coord) memcpy(out_coord, cap.coord.data(), sizeof(int32_t) * (size_t)k * BOARD_NUM_ACTION);
    if (out_visits) memcpy(out_visits, cap.visits.data(), sizeof(int32_t) * (size_t)k * BOARD_NUM_ACTION);
    if (out_prior) memcpy(out_prior, cap.prior.data(), sizeof(float) * (size_t)k * BOARD_NUM_ACTION);
    if (out_reward) memcpy(out_reward, cap.reward.data(), sizeof(float) * (size_t)k * BOARD_NUM_ACTION);
    return k;
  } catch (const std::exception& e) {
    fprintf(stderr, "refsp_run: %s\n", e.what());
    return -1;
  }
}

int64_t refsp_white_rows() { return g_white_rows; }

// MCTSResultT::addActions owns `static std::mt19937 rng(time(NULL))` (tree_search_base.h:238): the generator of the uniform_random
// pick method, seeded at the first search of the process.  This library is linked with -Bsymbolic-functions, so that the
// reference's call of time() binds to the definition below: a fixture generator sets the value BEFORE the first search of its
// process and the draws become reproducible.  0 (the default) = the real clock.
static std::atomic<int64_t> g_fixed_time{0};
void refsp_set_time(int64_t t) { g_fixed_time = t; }
time_t time(time_t* out) noexcept {
  time_t t = (time_t)g_fixed_time.load();
  if (t == 0) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    t = ts.tv_sec;
  }
  if (out) *out = t;
  return t;
}
void refsp_set_human_script(const int64_t* actions, int n) { g_human_script.assign(actions, actions + (n > 0 ? n : 0)); }
// prompts of the last online run: returns their number, copies min(number * 18 * N * N, cap) bytes
int64_t refsp_last_prompts(uint8_t* buf, int64_t cap) {
  const int64_t len = (int64_t)g_prompts.size();
  if (buf && cap > 0) memcpy(buf, g_prompts.data(), (size_t)std::min(len, cap));
  return g_n_prompts;
}
int64_t refsp_game_starts(int64_t* vers8) {
  if (vers8) memcpy(vers8, g_start_vers, sizeof(g_start_vers));
  return g_game_starts;
}

// GameOptions.preload_sgf / preload_sgf_move_to for the following refsp_run calls (path "" switches it off)
void refsp_set_preload(const char* path, int move_to) {
  g_preload_sgf = path ? path : "";
  g_preload_move_to = move_to;
}

// Record JSON (array text) of the games that finished during the last refsp_run; returns the length, copies min(len, cap)
int64_t refsp_last_records(char* buf, int64_t cap) {
  const int64_t n = (int64_t)g_last_records.size();
  if (buf && cap > 0) memcpy(buf, g_last_records.data(), (size_t)std::min(n, cap));
  return n;
}

// dumpSgf's header for a given final value (which overload of abs() decides between "B+R" and "B+1.500000"?)
int64_t refsp_sgf_of_value(float final_value, float komi, char* buf, int64_t cap) {
  GameOptions opt;
  opt.komi = komi;
  struct Open : public GoStateExt {
    using GoStateExt::GoStateExt;
    GoState& board() { return _state; }
  } st(0, opt);
  st.board().setFinalValue(final_value);
  const std::string t = st.dumpSgf("f.sgf");
  if (buf && cap > 0) { const int64_t k = std::min<int64_t>((int64_t)t.size(), cap - 1); memcpy(buf, t.data(), (size_t)k); buf[k] = 0; }
  return (int64_t)t.size();
}

int64_t refsp_last_sgfs(char* buf, int64_t cap) {
  const int64_t n = (int64_t)g_last_sgfs.size();
  if (buf && cap > 0) memcpy(buf, g_last_sgfs.data(), (size_t)std::min(n, cap));
  return n;
}

// ---- trainer side: the real GoStateExtOffline + GoFeature "train" extractors on one record --------------------
// record_json = one Record object; move_to / d4 chosen by the caller (the reference draws them with rng() in
// GoGameTrain::act :33-40).  Returns 0, or -1 when the reference throws.
int reftrain_sample(const char* record_json, int move_to, int d4, int num_future_actions, float* s, int64_t* offline_a,
                    float* winner, float* mcts_scores, float* predicted_value, int32_t* move_idx, int32_t* num_move,
                    int32_t* aug_code, int64_t* selfplay_ver) {
  try {
    Record r = Record::createFromJson(json::parse(record_json));
    GameOptions opt;
    opt.num_future_actions = num_future_actions;
    GoStateExtOffline st(0, opt);
    st.fromRecord(r);
    st.switchBeforeMove((size_t)move_to);
    st._bf.setD4Code(d4);
    GoFeature::extractStateExtAGZ(st, s);
    GoFeature::extractOfflineAction(st, offline_a);
    GoFeature::extractWinner(st, winner);
    GoFeature::extractMCTSPi(st, mcts_scores);
    GoFeature::extractMoveIdx(st, move_idx);
    GoFeature::extractNumMove(st, num_move);
    GoFeature::extractAugCode(st, aug_code);
    GoFeature::extractStateSelfplayVersion(st, selfplay_ver);
    *predicted_value = (size_t)*move_idx < r.result.values.size() ? st.getPredictedValue(*move_idx) : 0.0f;
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "reftrain_sample: %s\n", e.what());
    return -1;
  }
}

// The trainer's input path as the reference runs it: a ReaderQueuesT<Record> replay buffer (elf/distributed/shared_reader.h) filled the
// way TrainCtrl::OnReceive fills it (InsertWithParity(Record, &rng, reward > 0), train/game_ctrl.h:306-311) and ONE real GoGameTrain
// game thread whose act() (train/game_train.cc:23-58: getSamplerWithParity, sample, fromRecord, switchRandomMove, generateD4Code, 64
// states per act) feeds the "train" batch group (game.py:403-409) of a real elf::Context.  One game thread and batchsize 64: every
// act is one batch, so the rows come out in the order the thread drew them.  Outputs hold num_acts * 64 rows.  Returns rows, < 0 on error.
int reftrain_act(const char* records_json_array, int num_reader, int q_min_size, int q_max_size, uint32_t insert_seed, int64_t game_seed,
                 int num_acts, int num_future_actions, uint8_t* s_bits, int64_t* offline_a, float* winner, float* mcts_scores,
                 int32_t* move_idx, int32_t* num_move, int32_t* aug_code, int64_t* selfplay_ver) {
  try {
    std::vector<Record> recs = Record::createBatchFromJson(std::string(records_json_array));
    if (recs.empty() || num_reader < 2 || (num_reader & 1)) return -1;
    elf::shared::RQCtrl rq;
    rq.num_reader = num_reader;
    rq.ctrl.queue_min_size = (size_t)q_min_size;
    rq.ctrl.queue_max_size = (size_t)q_max_size;
    elf::shared::ReaderQueuesT<Record> reader(rq);
    std::mt19937 insert_rng(insert_seed);
    for (const Record& r : recs) reader.InsertWithParity(Record(r), &insert_rng, r.result.reward > 0);
    for (size_t q = 0; q < reader.nqueue(); ++q)
      if (reader[q]->size() < (size_t)q_min_size) return -2;   // the reference would sleep for minutes (wait_for_sufficient_data)

    const int B = 64, NA = BOARD_NUM_ACTION, PL = MAX_NUM_AGZ_FEATURE * BOARD_SIZE * BOARD_SIZE;
    ContextOptions co;
    co.num_games = 1;
    co.batchsize = B;
    GameOptions opt;
    opt.mode = "train";
    opt.seed = game_seed;
    opt.num_future_actions = num_future_actions;
    elf::Context ctx;
    GoFeature gf(opt);
    gf.registerExtractor(B, ctx.getExtractor());
    std::unique_ptr<GoGameTrain> game(new GoGameTrain(0, ctx.getClient(), co, opt, &reader));
    ctx.setStartCallback(1, [&](int, elf::GameClient*) { game->mainLoop(); });
    const std::vector<std::string> keys = {"s", "offline_a", "winner", "mcts_scores", "move_idx", "selfplay_ver", "aug_code", "num_move"};
    struct TB { std::vector<float> s, winner, ms; std::vector<int64_t> oa, ver; std::vector<int32_t> mi, ac, nm; };
    std::vector<TB> tb(2);
    std::vector<int> idx2tb;
    for (int r = 0; r < 2; ++r) {
      auto smo = ctx.createSharedMemOptions("train", B);
      smo.setTimeout(0);
      elf::SharedMem& sm = ctx.allocateSharedMem(smo, keys);
      const int idx = sm.getSharedMemOptions().getIdx();
      if ((int)idx2tb.size() <= idx) idx2tb.resize(idx + 1, -1);
      idx2tb[idx] = r;
      TB& b = tb[r];
      for (const auto& k : keys) {
        elf::AnyP* p = sm[k];
        const auto& f = p->field();
        const size_t ne = f.getSize().nelement();
        void* addr = nullptr;
        if (k == "s") { b.s.assign(ne, 0.f); addr = b.s.data(); }
        else if (k == "winner") { b.winner.assign(ne, 0.f); addr = b.winner.data(); }
        else if (k == "mcts_scores") { b.ms.assign(ne, 0.f); addr = b.ms.data(); }
        else if (k == "offline_a") { b.oa.assign(ne, 0); addr = b.oa.data(); }
        else if (k == "selfplay_ver") { b.ver.assign(ne, 0); addr = b.ver.data(); }
        else if (k == "move_idx") { b.mi.assign(ne, 0); addr = b.mi.data(); }
        else if (k == "aug_code") { b.ac.assign(ne, 0); addr = b.ac.data(); }
        else if (k == "num_move") { b.nm.assign(ne, 0); addr = b.nm.data(); }
        p->setAddress((uint64_t)addr, f.getSize().getContinuousStrides(f.getSizeOfType()).vec());
      }
    }
    ctx.start();
    int rows = 0;
    for (int a = 0; a < num_acts;) {
      const elf::SharedMem* sm = ctx.wait(100000);
      if (sm == nullptr) continue;
      const int eb = sm->getEffectiveBatchSize();
      if (sm->getSharedMemOptions().getLabel() == "train") {
        if (eb != B) { ctx.step(); ctx.stop(); return -3; }
        const TB& b = tb[idx2tb[sm->getSharedMemOptions().getIdx()]];
        for (int i = 0; i < eb; ++i, ++rows) {
          for (int j = 0; j < PL; ++j) s_bits[(size_t)rows * PL + j] = b.s[(size_t)i * PL + j] != 0.f ? 1 : 0;
          for (int j = 0; j < num_future_actions; ++j) offline_a[(size_t)rows * num_future_actions + j] = b.oa[(size_t)i * num_future_actions + j];
          for (int j = 0; j < NA; ++j) mcts_scores[(size_t)rows * NA + j] = b.ms[(size_t)i * NA + j];
          winner[rows] = b.winner[i]; move_idx[rows] = b.mi[i]; num_move[rows] = b.nm[i]; aug_code[rows] = b.ac[i]; selfplay_ver[rows] = b.ver[i];
        }
        ++a;
      }
      ctx.step();
    }
    ctx.stop();
    return rows;
  } catch (const std::exception& e) {
    fprintf(stderr, "reftrain_act: %s\n", e.what());
    return -1;
  }
}

// ---- the client's wire formats: the real Records / ThreadState / MsgRequestSeq (common/record.h) ---------------------------------
// Records as GuardedRecords (train/distri_client.h:111-170) keeps it: one object, fed and dumped repeatedly (clear() keeps the
// unordered_map's buckets, which the order of "states" in later messages depends on)
static std::unique_ptr<Records> g_client_records;
static int64_t copy_out(const std::string& t, char* buf, int64_t cap) {
  if (buf && cap > 0) { const int64_t k = std::min<int64_t>((int64_t)t.size(), cap - 1); memcpy(buf, t.data(), (size_t)k); buf[k] = 0; }
  return (int64_t)t.size();
}
void refrec_client_reset(const char* identity) { g_client_records.reset(new Records(identity ? identity : "")); }
int refrec_client_feed(const char* record_json) {
  try { g_client_records->addRecord(Record::createFromJson(json::parse(record_json))); return 0; }
  catch (const std::exception& e) { fprintf(stderr, "refrec_client_feed: %s\n", e.what()); return -1; }
}
void refrec_client_update_state(int thread_id, int seq, int move_idx, int64_t black, int64_t white) {
  ThreadState ts;
  ts.thread_id = thread_id; ts.seq = seq; ts.move_idx = move_idx; ts.black = black; ts.white = white;
  g_client_records->updateState(ts);
}
int64_t refrec_client_dump_and_clear(char* buf, int64_t cap) {     // GuardedRecords::dumpAndClear :156-169
  const std::string t = g_client_records->dumpJsonString();
  g_client_records->clear();
  return copy_out(t, buf, cap);
}
// the server's side of the same message (TrainCtrl::OnReceive, train/game_ctrl.h:288): Records::createFromJsonString; out3 =
// {records, states, sum of the states' move_idx}; identity copied to buf.  < 0 when the reference throws.
int64_t refrec_records_parse(const char* text, int64_t* out3, char* buf, int64_t cap) {
  try {
    Records rs = Records::createFromJsonString(std::string(text));
    int64_t mv = 0;
    for (const auto& t : rs.states) mv += t.second.move_idx;
    out3[0] = (int64_t)rs.records.size(); out3[1] = (int64_t)rs.states.size(); out3[2] = mv;
    return copy_out(rs.identity, buf, cap);
  } catch (const std::exception& e) { fprintf(stderr, "refrec_records_parse: %s\n", e.what()); return -1; }
}
// what the reference's server sends: MsgRequestSeq::dumpJsonString with the TSOptions of cfg
int64_t refrec_request_seq_dump(const RefSpConfig* cfg, int64_t black_ver, int64_t white_ver, int client_type, int num_game_thread_used,
                                float black_thres, float white_thres, float never_resign_prob, int player_swap, int async, int64_t seq,
                                char* buf, int64_t cap) {
  MsgRequestSeq m;
  m.seq = seq;
  m.request.vers.black_ver = black_ver; m.request.vers.white_ver = white_ver;
  auto& ts = m.request.vers.mcts_opt;
  ts.num_threads = cfg->mcts_threads; ts.num_rollouts_per_thread = cfg->rollouts_per_thread; ts.num_rollouts_per_batch = cfg->rollouts_per_batch;
  ts.virtual_loss = cfg->virtual_loss; ts.persistent_tree = cfg->persistent_tree != 0; ts.root_epsilon = cfg->root_epsilon;
  ts.root_alpha = cfg->root_alpha; ts.alg_opt.use_prior = cfg->use_prior != 0; ts.alg_opt.c_puct = cfg->c_puct;
  ts.alg_opt.unexplored_q_zero = cfg->unexplored_q_zero != 0; ts.alg_opt.root_unexplored_q_zero = cfg->root_unexplored_q_zero != 0;
  ts.pick_method = cfg->pick_method == 1 ? "strongest_prior" : cfg->pick_method == 2 ? "uniform_random" : "most_visited";
  auto& c = m.request.client_ctrl;
  c.client_type = (ClientType)client_type; c.num_game_thread_used = num_game_thread_used; c.black_resign_thres = black_thres;
  c.white_resign_thres = white_thres; c.never_resign_prob = never_resign_prob; c.player_swap = player_swap != 0; c.async = async != 0;
  return copy_out(m.dumpJsonString(), buf, cap);
}
// text -> MsgRequestSeq -> text; < 0 when the reference throws (a missing field)
int64_t refrec_request_seq_roundtrip(const char* text, char* buf, int64_t cap) {
  try { return copy_out(MsgRequestSeq::createFromJson(json::parse(text)).dumpJsonString(), buf, cap); }
  catch (const std::exception& e) { return -1; }
}

// CPU baseline of the trainer's input pipeline: what GoGameTrain::act does per sample (fromRecord, switchRandomMove,
// generateD4Code, then every "train" extractor the batcher would call on the sending thread), records parsed once,
// `threads` host threads each producing samples into its own row buffers.  Returns the number of board steps replayed.
int64_t reftrain_bench(const char* records_json_array, int n_samples, int threads, int num_future_actions, double* seconds) {
  try {
    std::vector<Record> recs = Record::createBatchFromJson(std::string(records_json_array));
    if (recs.empty() || threads < 1) return -1;
    GameOptions opt;
    opt.num_future_actions = num_future_actions;
    std::atomic<int64_t> steps{0};
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t) {
      th.emplace_back([&, t]() {
        std::mt19937 rng(1234 + t);
        GoStateExtOffline st(t, opt);
        std::vector<float> s(MAX_NUM_AGZ_FEATURE * BOARD_SIZE * BOARD_SIZE), ms(BOARD_NUM_ACTION);
        std::vector<int64_t> oa(num_future_actions);
        float w, pv = 0; int32_t mi, nm, ac; int64_t sv;
        int64_t local = 0;
        for (int i = t; i < n_samples; i += threads) {
          while (true) {
            st.fromRecord(recs[rng() % recs.size()]);
            if (st.switchRandomMove(&rng)) break;
          }
          st.generateD4Code(&rng);
          GoFeature::extractStateExtAGZ(st, s.data());
          GoFeature::extractOfflineAction(st, oa.data());
          GoFeature::extractWinner(st, &w);
          GoFeature::extractMCTSPi(st, ms.data());
          GoFeature::extractMoveIdx(st, &mi);
          GoFeature::extractNumMove(st, &nm);
          GoFeature::extractAugCode(st, &ac);
          GoFeature::extractStateSelfplayVersion(st, &sv);
          local += mi;
        }
        (void)pv;
        steps += local;
      });
    }
    for (auto& x : th) x.join();
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    return steps.load();
  } catch (const std::exception& e) {
    fprintf(stderr, "reftrain_bench: %s\n", e.what());
    return -1;
  }
}

// Record::createFromJson -> setJsonFields -> dump: the reference's reading of a record, re-serialised
int64_t reftrain_record_roundtrip(const char* record_json, char* buf, int64_t cap) {
  try {
    Record r = Record::createFromJson(json::parse(record_json));
    json j;
    r.setJsonFields(j);
    const std::string t = j.dump();
    if (buf && cap > 0) memcpy(buf, t.data(), (size_t)std::min<int64_t>((int64_t)t.size(), cap));
    return (int64_t)t.size();
  } catch (const std::exception& e) {
    fprintf(stderr, "reftrain_record_roundtrip: %s\n", e.what());
    return -1;
  }
}

// sgf/sgf.h helpers on their own
int reftrain_sgfstr2coords(const char* sgf, uint16_t* out, int cap) {
  std::vector<Coord> m = sgfstr2coords(std::string(sgf));
  for (size_t i = 0; i < m.size() && (int)i < cap; ++i) out[i] = m[i];
  return (int)m.size();
}
int64_t reftrain_coords2sgfstr(const uint16_t* coords, int n, char* buf, int64_t cap) {
  std::vector<Coord> m(coords, coords + n);
  const std::string t = coords2sgfstr(m);
  if (buf && cap > 0) memcpy(buf, t.data(), (size_t)std::min<int64_t>((int64_t)t.size(), cap));
  return (int64_t)t.size();
}

void refsp_stub_net(const float* s, int batch, uint32_t salt, int tie_levels, float* pi, float* v) {
  stubnet_eval(s, batch, BOARD_SIZE, salt, tie_levels, pi, v);
}

}  // extern "C"
