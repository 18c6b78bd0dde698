// Host side of the self-play loop over the device engines (include/elf_amd.h, elfsp_*).
//
// Mirrors, for G games advanced in lock-step on one GPU, what the reference does on one std::thread per game:
//   src_cpp/elfgames/go/common/game_selfplay.cc   GoGameSelfPlay::act :272-430, init_ai :30-78,
//                                                  mcts_make_diverse_move :80-95, mcts_update_info :97-119, finish_game :121-149
//   src_cpp/elf/ai/tree_search/mcts.h             MCTSAI_T::act / align_state / advanceMoves :59-81,141-167
//   src_cpp/elf/ai/tree_search/tree_search.h      TreeSearchT::run :410-426, chooseAction :495-528
//   src_cpp/elfgames/go/mcts/mcts.h               MCTSGoAI::getValue / getMCTSPolicy :358-372
//   src_cpp/elfgames/go/common/game_utils.h       ResignCheck :14-54
// Everything that depends on libstdc++'s implementation-defined distributions stays on the host and uses the
// very same library: std::gamma_distribution (Dirichlet noise), std::uniform_real_distribution (move sampling,
// never-resign draw), std::mt19937 streams per game and per actor.
#include <math.h>
#include <string.h>

#include <new>
#include <random>
#include <utility>
#include <vector>

#include <chrono>
#include <deque>
#include <string>

#include "engine_host.h"
#include "record_host.h"

struct ElfSpSearchRec {   // == ElfSpSearch in include/elf_amd.h
  int32_t game, move_played, best_action, total_visits, n_edges;
  float root_value, max_score, predicted_value;
};

struct SpGame {
  std::mt19937 rng;        // GoGameBase::_rng (game_base.h:32-38)
  std::mt19937 actor_rng;  // MCTSActor::rng_ (go/mcts/mcts.h:52), seeded with params.seed = _rng() (game_selfplay.cc:47)
  // ResignCheck (game_utils.h:14-54)
  bool never_resign = false, has_calculated_never_resign = false;
  float last_predicted = 0.0f;
  int ply = 1;             // GoState::getPly of the game board
  int seq = 0;             // games finished by this slot
  float last_final = 0.0f; // GoStateExt::getLastGameFinalValue (go_state_ext.h:153-155)
  int last_move = -1;      // what the last finished search of this game did: the Coord it forwarded, M_RESIGN, or -1 (none yet)
  int sgf_iter = 0;        // GoGameSelfPlay::_sgf_iter (game_selfplay.h): next move of the preloaded SGF
  SpRecord rec;            // GoStateExt::_mcts_policies / _predicted_values / the game's moves (go_state_ext.h:131-148)
};

struct ElfSelfPlay {
  ElfSpOptions opt;
  ElfGoEngine* eng = nullptr;
  ElfMcts* mcts = nullptr;
  int G = 0, NE = 0, NA = 0, K = 0, T = 1, KT = 0, steps_per_move = 0, step_in_move = 0, W = 0;
  hipStream_t stream = nullptr;
  std::vector<SpGame> games;
  // device scratch
  int32_t* d_counts = nullptr;   // [4]: rows, error bits, running total of rows (u64)
  int32_t* d_info = nullptr;     // [G][8]
  int32_t *d_coord = nullptr, *d_visits = nullptr, *d_moves = nullptr, *d_ids = nullptr, *d_binfo = nullptr;
  float *d_prior = nullptr, *d_reward = nullptr, *d_etas = nullptr, *d_Z = nullptr, *d_val = nullptr;
  uint8_t* d_ok = nullptr;
  // host mirrors
  std::vector<int32_t> h_info, h_coord, h_visits, h_moves, h_binfo;
  std::vector<float> h_prior, h_reward, h_etas, h_Z, h_val;
  std::vector<uint8_t> h_d4, h_ok;
  int32_t h_counts[2] = {0, 0};
  int last_rows = 0;             // rows of the last begin_step, -1 = left on the device (begin_step without n_rows)
  bool search_open = false;
  // request state (MsgRequest, common/record.h): versions + client_ctrl of the current and of a pending request
  int64_t black_ver = 0, white_ver = -1;
  bool have_pending = false, pending_async = false, cur_async = false, had_request = false;
  int64_t pend_black = 0, pend_white = -1;
  float pend_thres = 0.f, pend_never = 0.f;
  int game_starts = 0;
  std::chrono::steady_clock::time_point t_after_drain;
  // statistics / capture
  int64_t n_moves = 0, n_games = 0, n_rollouts = 0, n_rows = 0, n_steps = 0;
  // move boundaries: boundary_wait_ns = the first wait of sp_finish_move (the stream drains whatever the host had queued ahead:
  // pipeline depth, not boundary work); boundary_ns = everything after it up to the end of the next sp_begin_search
  int64_t boundary_ns = 0, boundary_wait_ns = 0, n_boundaries = 0;
  double sum_final = 0.0;
  std::vector<ElfSpSearchRec> log_search;
  std::vector<int32_t> log_coord, log_visits;
  std::vector<float> log_prior, log_reward;
  int log_cap = 0;
  // finished-game records (GameNotifier::OnGameEnd -> GoStateExt::dumpRecord), newest at the back
  std::deque<std::string> records;
  std::deque<float> finished_values;   // final value of every finished game not yet taken (GameStats::feedWinRate, game_stats.h:41-44)
  SpRecordMeta meta{};
  std::vector<int32_t> sgf;   // GameOptions.preload_sgf as reference Coords (elfsp_preload)
  int sgf_move_to = -1;       // GameOptions.preload_sgf_move_to
};

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
    r.thread_id = (uint64_t)g;
    r.seq = gm.seq + 2;                          // _seq: ctor restart() -> 1, first request restart() -> 2, then +1 per game
    r.timestamp = (uint64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    if ((int)sp->records.size() >= sp->opt.keep_records) sp->records.pop_front();
    sp->records.push_back(elfrec_record_json(sp->meta, r));
  }
  gm.rec = SpRecord();                           // GoStateExt::restart(): _mcts_policies.clear(), _predicted_values.clear()
}

#define SPCHK(x)                  \
  do {                            \
    int _rc = (x);                \
    if (_rc != 0) return _rc;     \
  } while (0)

// GoGameSelfPlay::restart :202-219: forward the first preload_sgf_move_to moves of the preloaded SGF on every (fresh) game board
static int sp_forward_preload(ElfSelfPlay* sp) {
  const int G = sp->G, n = (int)sp->sgf.size();
  int fwd = 0;
  for (; fwd < n && fwd < sp->sgf_move_to; ++fwd) {            // while (!_sgf_iter.done() && i < preload_sgf_move_to)
    std::vector<int32_t> mv(G, (int32_t)sp->sgf[fwd]);
    HIPCHK(hipMemcpyAsync(sp->d_moves, mv.data(), 4 * G, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_forward(sp->eng, nullptr, sp->d_moves, G, sp->d_ok, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_ok.data(), sp->d_ok, G, hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (int g = 0; g < G; ++g)
      if (sp->h_ok[g] != 1) return ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD;   // "Preload sgf: move not valid!" :211-215
    for (int g = 0; g < G; ++g) {
      sp->games[g].ply++;
      if (sp->opt.keep_records > 0) sp->games[g].rec.moves.push_back((uint16_t)sp->sgf[fwd]);
    }
  }
  for (int g = 0; g < G; ++g) sp->games[g].sgf_iter = fwd;
  return 0;
}

// GoGameSelfPlay::OnReceive (game_selfplay.cc:222-270) at a move boundary: new versions -> restart() of every game (board,
// tree, record, resign check; a fresh MCTS actor seeded with the next draw of the game's generator, init_ai :45-47); same
// versions (or async) -> only the request (thresholds) changes.
static int sp_apply_request(ElfSelfPlay* sp) {
  sp->have_pending = false;
  // the first request is what starts the reference's games (they wait for it, game_selfplay.cc:277-279): elfsp_create has
  // already done that restart (boards empty, actor seeded with the first draw), so it only counts as a game start here
  const bool first = !sp->had_request;
  sp->had_request = true;
  const bool same_vers = first || (sp->pend_black == sp->black_ver && sp->pend_white == sp->white_ver);
  if (first) sp->game_starts++;
  sp->opt.resign_thres = sp->pend_thres;            // (black + white) / 2 with both equal (go_state_ext.h:62-66)
  sp->opt.never_resign_prob = sp->pend_never;
  sp->cur_async = sp->pending_async;
  if (!(same_vers || sp->pending_async)) {
    const int G = sp->G;
    SPCHK(elfgo_reset(sp->eng, nullptr, G, sp->stream));
    SPCHK(elfmcts_clear(sp->mcts, nullptr, G, sp->stream));
    for (int g = 0; g < G; ++g) {
      SpGame& gm = sp->games[g];
      gm.actor_rng.seed(gm.rng());
      gm.ply = 1; gm.never_resign = false; gm.has_calculated_never_resign = false; gm.last_predicted = 0.0f;
      gm.seq++; gm.sgf_iter = 0;
      gm.rec = SpRecord();
    }
    if (!sp->sgf.empty()) SPCHK(sp_forward_preload(sp));
    sp->game_starts++;
  }
  sp->black_ver = sp->pend_black; sp->white_ver = sp->pend_white;
  sp->opt.model_ver = (int32_t)sp->black_ver;
  sp->meta = elfrec_meta_from_options(sp->opt);
  sp->opt.mcts.required_version = sp->cur_async ? -1 : sp->black_ver;
  SPCHK(elfmcts_set_options(sp->mcts, &sp->opt.mcts));
  return 0;
}

static int sp_begin_search(ElfSelfPlay* sp) {
  // MCTSAI_T::act -> align_state (mcts.h:141-167) happened at the end of the previous move (treeAdvance) or is a
  // clear when the tree is not persistent; then TreeSearchT::run :410-417
  const int G = sp->G;
  if (sp->have_pending) SPCHK(sp_apply_request(sp));
  if (!sp->opt.persistent_tree) SPCHK(elfmcts_clear(sp->mcts, nullptr, G, sp->stream));
  SPCHK(elfmcts_set_root(sp->mcts, nullptr, sp->stream));
  SPCHK(elfmcts_root(sp->mcts, sp->d_info, nullptr, nullptr, nullptr, nullptr, nullptr, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_info.data(), sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  for (int g = 0; g < G; ++g)
    if (sp->h_info[g * ELFMCTS_ROOT_WORDS + 6]) return ELFGO_E_MCTS_BASE - sp->h_info[g * ELFMCTS_ROOT_WORDS + 6];
  if (sp->opt.root_epsilon > 0.0f) {
    // NodeT::enhanceExploration (tree_search_node.h:132-155), draws from actors_[0]->rng()
    for (int g = 0; g < G; ++g) {
      const int n = sp->h_info[g * ELFMCTS_ROOT_WORDS + 0];
      std::gamma_distribution<> dis(sp->opt.root_alpha);
      float Z = 1e-10;
      float* et = &sp->h_etas[(size_t)g * sp->NE];
      for (int i = 0; i < n; ++i) {
        et[i] = dis(sp->games[g].actor_rng);
        Z += et[i];
      }
      sp->h_Z[g] = Z;
    }
    HIPCHK(hipMemcpyAsync(sp->d_etas, sp->h_etas.data(), sizeof(float) * (size_t)G * sp->NE, hipMemcpyHostToDevice, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->d_Z, sp->h_Z.data(), sizeof(float) * G, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfmcts_dirichlet(sp->mcts, sp->d_etas, sp->d_Z, sp->opt.root_epsilon, sp->stream));
  }
  // BoardFeature::RandomShuffle draws of this move (go/mcts/mcts.h:175-183), from a copy of the actor stream
  for (int g = 0; g < G; ++g) {
    std::mt19937 c = sp->games[g].actor_rng;
    uint8_t* d = &sp->h_d4[(size_t)g * sp->W];
    for (int i = 0; i < sp->W; ++i) d[i] = (uint8_t)(c() % 8);
  }
  SPCHK(elfmcts_set_d4(sp->mcts, sp->h_d4.data(), sp->stream));
  sp->step_in_move = 0;
  sp->search_open = true;
  return 0;
}

static int sp_finish_move(ElfSelfPlay* sp) {
  const int G = sp->G, NE = sp->NE;
  SPCHK(elfmcts_root(sp->mcts, sp->d_info, sp->d_coord, sp->d_visits, sp->d_prior, sp->d_reward, nullptr, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_info.data(), sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_coord.data(), sp->d_coord, sizeof(int32_t) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_visits.data(), sp->d_visits, sizeof(int32_t) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_prior.data(), sp->d_prior, sizeof(float) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_reward.data(), sp->d_reward, sizeof(float) * (size_t)G * NE, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  sp->t_after_drain = std::chrono::steady_clock::now();
  std::vector<int> finished, sgf_done;
  for (int g = 0; g < G; ++g) {
    SpGame& gm = sp->games[g];
    const int32_t* info = &sp->h_info[g * ELFMCTS_ROOT_WORDS];
    if (info[6]) return ELFGO_E_MCTS_BASE - info[6];
    gm.actor_rng.discard((unsigned long long)info[5]);   // D4 draws the search consumed
    const int n = info[0];
    const int32_t* coord = &sp->h_coord[(size_t)g * NE];
    const int32_t* visits = &sp->h_visits[(size_t)g * NE];
    const float* reward = &sp->h_reward[(size_t)g * NE];
    float root_value;
    memcpy(&root_value, &info[4], 4);
    // chooseAction :495-528 with MCTSResultT::addActions (tree_search_base.h:237-294), MOST_VISITED
    int best_action = M_INVALID, total_visits = 0, best_i = -1;
    float max_score = -3.402823466e+38f;
    for (int i = 0; i < n; ++i) {
      const float score = (float)visits[i];
      total_visits += visits[i];
      if (score > max_score) { max_score = score; best_action = coord[i]; best_i = i; }
    }
    int c = best_action;
    // mcts_make_diverse_move (game_selfplay.cc:80-95): MCTSPolicy::normalize (tree_search_base.h:190-203) + sampleAction
    const bool diverse = gm.ply <= sp->opt.policy_distri_cutoff;
    const bool keep_policy = sp->opt.keep_records > 0 && (diverse || sp->opt.policy_distri_training_for_all);
    std::vector<std::pair<int, float>> policy;
    if ((diverse && n > 0) || keep_policy) {
      policy.resize(n);
      float exp_sum = 0;
      for (int i = 0; i < n; ++i) {
        float e = std::pow((float)visits[i], 1.0 / 1.0f);
        policy[i] = std::make_pair(coord[i], e);
        exp_sum += e;
      }
      for (auto& p : policy) p.second /= exp_sum;
    }
    if (diverse && n > 0) {
      // elf_utils::sample_multinomial (elf/utils/utils.h:159-182)
      float Z = 0.0;
      for (const auto& p : policy) Z += p.second;
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
    // mcts_update_info :97-119 with MCTSGoAI::getValue (go/mcts/mcts.h:358-365)
    float predicted = root_value;
    if (total_visits != 0 && best_i >= 0) predicted = reward[best_i] / visits[best_i];
    gm.last_predicted = predicted;
    if (sp->opt.keep_records > 0) gm.rec.values.push_back(predicted);   // addPredictedValue, mcts_update_info :98-100
    if (sp->log_cap > 0 && (int)sp->log_search.size() < sp->log_cap) {
      ElfSpSearchRec r;
      r.game = g; r.move_played = c; r.best_action = best_action; r.total_visits = total_visits; r.n_edges = n;
      r.root_value = root_value; r.max_score = max_score; r.predicted_value = predicted;
      sp->log_search.push_back(r);
      sp->log_coord.insert(sp->log_coord.end(), coord, coord + NE);
      sp->log_visits.insert(sp->log_visits.end(), visits, visits + NE);
      sp->log_prior.insert(sp->log_prior.end(), &sp->h_prior[(size_t)g * NE], &sp->h_prior[(size_t)g * NE] + NE);
      sp->log_reward.insert(sp->log_reward.end(), reward, reward + NE);
    }
    // shouldResign (go_state_ext.h:207-214) -> ResignCheck::check (game_utils.h:24-40); side to move = parity of ply
    bool resign = false;
    {
      const bool black = (gm.ply & 1) == 1;   // ply 1 = Black to move
      const float value = black ? predicted : -predicted;
      if (!gm.has_calculated_never_resign) {
        std::uniform_real_distribution<> dis(0.0, 1.0);
        gm.never_resign = (dis(gm.rng) < sp->opt.never_resign_prob);
        gm.has_calculated_never_resign = true;
      }
      if (!gm.never_resign && !(value >= -1.0 + sp->opt.resign_thres)) resign = true;
    }
    if (resign && gm.ply >= 50) {
      finished.push_back(g);
      gm.last_move = M_RESIGN;
      sp->h_moves[g] = M_PASS;       // placeholder; the board is reset below
      const float fv = ((gm.ply & 1) == 1) ? -1.0f : 1.0f;   // setFinalValue FR_RESIGN (go_state_ext.h:83-85)
      sp->sum_final += fv;
      sp_finish_record(sp, g, fv, gm.ply);
      gm.ply = -1;                   // marks "finished without a move"
    } else if (!sp->sgf.empty() && gm.sgf_iter >= (int)sp->sgf.size()) {
      sgf_done.push_back(g);         // preloaded SGF exhausted: finish_game(FR_MAX_STEP), game_selfplay.cc:392-396
      sp->h_moves[g] = M_PASS;
    } else {
      if (!sp->sgf.empty()) c = sp->sgf[gm.sgf_iter++];       // "Move changes from {} to {}" :397-405
      sp->h_moves[g] = c;
      gm.last_move = c;
    }
  }
  if (!sgf_done.empty()) {
    // setFinalValue(FR_MAX_STEP) = GoState::evaluate(komi) of the position the search started from; no move is forwarded
    std::vector<int32_t> ids(sgf_done.begin(), sgf_done.end());
    HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, (int)ids.size(), sp->opt.mcts.komi, sp->d_val, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * ids.size(), hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (size_t i = 0; i < ids.size(); ++i) {
      SpGame& gm = sp->games[ids[i]];
      sp->sum_final += sp->h_val[i];
      sp_finish_record(sp, ids[i], sp->h_val[i], gm.ply);
      gm.ply = -1;
      finished.push_back(ids[i]);
    }
  }
  sp->n_moves += G;
  // GoStateExt::forward(c) (game_selfplay.cc:408) on the real game boards; tree follows (advanceMoves -> treeAdvance)
  HIPCHK(hipMemcpyAsync(sp->d_moves, sp->h_moves.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_forward(sp->eng, nullptr, sp->d_moves, G, sp->d_ok, sp->stream));
  if (sp->opt.persistent_tree) SPCHK(elfmcts_advance(sp->mcts, sp->d_moves, sp->stream));
  SPCHK(elfgo_info(sp->eng, nullptr, G, sp->d_binfo, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_binfo.data(), sp->d_binfo, sizeof(int32_t) * G * ELFGO_INFO_WORDS, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipMemcpyAsync(sp->h_ok.data(), sp->d_ok, G, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  std::vector<int> by_end;
  for (int g = 0; g < G; ++g) {
    SpGame& gm = sp->games[g];
    if (gm.ply < 0) continue;   // resigned
    if (sp->h_ok[g] != 1) return ELFGO_E_MCTS_BASE - ELFMCTS_E_FORWARD;   // "Something is wrong! Move cannot be applied" :409-418
    const int32_t* bi = &sp->h_binfo[g * ELFGO_INFO_WORDS];
    gm.ply = bi[0];
    if (sp->opt.keep_records > 0) gm.rec.moves.push_back((uint16_t)sp->h_moves[g]);   // GoState::_moves
    const bool terminated = bi[9] != 0;
    if (terminated || (sp->opt.move_cutoff > 0 && gm.ply >= sp->opt.move_cutoff)) by_end.push_back(g);   // :420-429
  }
  if (!by_end.empty()) {
    // finish_game -> setFinalValue: GoState::evaluate(komi) (go_state_ext.h:100-102)
    std::vector<int32_t> ids(by_end.begin(), by_end.end());
    HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, (int)ids.size(), sp->opt.mcts.komi, sp->d_val, sp->stream));
    HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * ids.size(), hipMemcpyDeviceToHost, sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));
    for (size_t i = 0; i < ids.size(); ++i) {
      sp->sum_final += sp->h_val[i];
      sp_finish_record(sp, ids[i], sp->h_val[i], sp->games[ids[i]].ply);
    }
    finished.insert(finished.end(), by_end.begin(), by_end.end());
  }
  if (!finished.empty()) {
    // finish_game :121-149: _ai->endGame (resetTree), _state_ext.restart() (state reset, resign check reset)
    std::vector<int32_t> ids(finished.begin(), finished.end());
    HIPCHK(hipMemcpyAsync(sp->d_ids, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfgo_reset(sp->eng, sp->d_ids, (int)ids.size(), sp->stream));
    SPCHK(elfmcts_clear(sp->mcts, sp->d_ids, (int)ids.size(), sp->stream));
    HIPCHK(hipStreamSynchronize(sp->stream));   // ids is a stack vector
    for (int g : finished) {
      SpGame& gm = sp->games[g];
      gm.ply = 1; gm.never_resign = false; gm.has_calculated_never_resign = false; gm.last_predicted = 0.0f;
      gm.seq++;
    }
    sp->n_games += (int64_t)finished.size();
  }
  sp->search_open = false;
  return 0;
}

extern "C" {

int elfsp_create(const ElfSpOptions* o, int device, const uint64_t* zobrist_host, ElfSelfPlay** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return ELFGO_E_BADARG;
  DevGuard _dg(device);
  if (!o || !out || !zobrist_host || o->num_games <= 0 || o->num_rollouts_per_thread <= 0 || o->mcts.num_threads <= 0 ||
      o->mcts.num_rollouts_per_batch <= 0)
    return ELFGO_E_BADARG;
  ElfSelfPlay* sp = new (std::nothrow) ElfSelfPlay();
  if (!sp) return ELFGO_E_NOMEM;
  sp->opt = *o;
  const int G = o->num_games;
  int rc = elfgo_create(o->board_size, G, device, zobrist_host, &sp->eng);
  if (rc) { delete sp; return rc; }
  sp->K = o->mcts.num_rollouts_per_batch;
  sp->T = o->mcts.num_threads;
  sp->KT = sp->K * sp->T;
  sp->steps_per_move = (o->num_rollouts_per_thread + sp->K - 1) / sp->K;   // for (idx = 0; idx < num_rollout; idx += batch) tree_search.h:112-117, in every search thread
  sp->W = sp->steps_per_move * sp->KT;
  sp->black_ver = o->model_ver; sp->white_ver = -1;
  rc = elfmcts_create(sp->eng, G, o->nodes_per_game, sp->W, &o->mcts, &sp->mcts);
  if (rc) { elfgo_destroy(sp->eng); delete sp; return rc; }
  sp->G = G; sp->NE = elfmcts_edge_stride(sp->mcts); sp->NA = o->board_size * o->board_size + 1;
  sp->games.resize(G);
  const uint64_t now_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                              std::chrono::system_clock::now().time_since_epoch()).count();
  for (int g = 0; g < G; ++g) {
    // GoGameBase ctor: _rng.seed(_seed) (game_base.h:32-38); restart() -> init_ai: params.seed = _rng() (game_selfplay.cc:47).
    // Per-game seed rule: include/elf_amd.h (ElfSpOptions).
    const uint64_t gi = (uint64_t)(uint32_t)(o->game_idx_base + g);
    uint64_t seed;
    if (o->seed != 0) seed = (uint64_t)o->seed + gi;
    else seed = ((now_ms / 1000) * 1000 + now_ms + (uint64_t)(int)(gi ^ o->job_hash) * 2341479ull) % 100000000ull;
    sp->games[g].rng.seed((std::mt19937::result_type)seed);
    const uint64_t aseed = sp->games[g].rng();
    sp->games[g].actor_rng.seed(aseed);
  }
#define A(ptr, bytes) do { hipError_t _e = hipMalloc((void**)&(ptr), (bytes)); if (_e != hipSuccess) { elfsp_destroy(sp); return (int)_e; } } while (0)
  const size_t GE = (size_t)G * sp->NE;
  A(sp->d_counts, 16); A(sp->d_info, sizeof(int32_t) * G * ELFMCTS_ROOT_WORDS);
  A(sp->d_coord, 4 * GE); A(sp->d_visits, 4 * GE); A(sp->d_prior, 4 * GE); A(sp->d_reward, 4 * GE); A(sp->d_etas, 4 * GE);
  A(sp->d_Z, 4 * G); A(sp->d_moves, 4 * G); A(sp->d_ids, 4 * G); A(sp->d_val, 4 * G); A(sp->d_ok, G);
  A(sp->d_binfo, sizeof(int32_t) * G * ELFGO_INFO_WORDS);
#undef A
  sp->h_info.resize(G * ELFMCTS_ROOT_WORDS); sp->h_coord.resize(GE); sp->h_visits.resize(GE); sp->h_prior.resize(GE);
  sp->h_reward.resize(GE); sp->h_etas.assign(GE, 0.f); sp->h_Z.resize(G); sp->h_moves.resize(G); sp->h_val.resize(G);
  sp->h_ok.resize(G); sp->h_binfo.resize(G * ELFGO_INFO_WORDS); sp->h_d4.resize((size_t)G * sp->W);
  if (hipMemset(sp->d_counts, 0, 16) != hipSuccess) { elfsp_destroy(sp); return ELFGO_E_NOMEM; }
  sp->log_cap = o->log_searches;
  sp->meta = elfrec_meta_from_options(*o);
  *out = sp;
  return 0;
}

int elfsp_destroy(ElfSelfPlay* sp) {
  if (!sp) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng ? sp->eng->device : 0);
  void* ptrs[] = {sp->d_counts, sp->d_info, sp->d_coord, sp->d_visits, sp->d_prior, sp->d_reward, sp->d_etas, sp->d_Z,
                  sp->d_moves, sp->d_ids, sp->d_val, sp->d_ok, sp->d_binfo};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (sp->mcts) elfmcts_destroy(sp->mcts);
  if (sp->eng) elfgo_destroy(sp->eng);
  delete sp;
  return 0;
}

ElfGoEngine* elfsp_engine(ElfSelfPlay* sp) { return sp ? sp->eng : nullptr; }
ElfMcts* elfsp_mcts(ElfSelfPlay* sp) { return sp ? sp->mcts : nullptr; }
int elfsp_max_rows(const ElfSelfPlay* sp) { return sp ? sp->G * sp->KT : ELFGO_E_BADARG; }

int elfsp_begin_step(ElfSelfPlay* sp, void* s_dst, int64_t stride_elems, int* n_rows, void* stream) {
  if (!sp || !s_dst) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  if (!sp->search_open) {
    const auto t0 = std::chrono::steady_clock::now();
    SPCHK(sp_begin_search(sp));
    sp->boundary_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  }
  SPCHK(elfmcts_select(sp->mcts, nullptr, s_dst, stride_elems, sp->d_counts, sp->stream));
  if (!n_rows) {          // row count and error word stay on the device until the move boundary
    sp->last_rows = -1;
    return 0;
  }
  HIPCHK(hipMemcpyAsync(sp->h_counts, sp->d_counts, 8, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  if (sp->h_counts[1]) return ELFGO_E_MCTS_BASE - sp->h_counts[1];
  sp->last_rows = sp->h_counts[0];
  *n_rows = sp->last_rows;
  return 0;
}

int elfsp_last_rows(ElfSelfPlay* sp, int* n_rows) {
  if (!sp || !n_rows) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  HIPCHK(hipMemcpyAsync(sp->h_counts, sp->d_counts, 8, hipMemcpyDeviceToHost, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  if (sp->h_counts[1]) return ELFGO_E_MCTS_BASE - sp->h_counts[1];
  *n_rows = sp->h_counts[0];
  return 0;
}

int elfsp_end_step(ElfSelfPlay* sp, const float* pi, int64_t pi_stride_floats, const float* value, const int64_t* rv, void* stream) {
  if (!sp || !sp->search_open) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  SPCHK(elfmcts_expand(sp->mcts, pi, pi_stride_floats, value, rv, sp->last_rows, sp->stream));
  sp->n_rollouts += (int64_t)sp->G * sp->KT;
  sp->n_steps++;
  if (++sp->step_in_move >= sp->steps_per_move) {
    const auto t0 = std::chrono::steady_clock::now();
    SPCHK(sp_finish_move(sp));
    const auto t1 = std::chrono::steady_clock::now();
    sp->boundary_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(sp->t_after_drain - t0).count();
    sp->boundary_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - sp->t_after_drain).count();
    sp->n_boundaries++;
  }
  return 0;
}

int elfsp_set_request(ElfSelfPlay* sp, int64_t black_ver, int64_t white_ver, float resign_thres, float never_resign_prob, int async) {
  if (!sp || black_ver < 0) return ELFGO_E_BADARG;     // black_ver < 0 is the reference's "wait" request: nothing to play
  if (white_ver >= 0) return ELFGO_E_BADARG;           // second AI for White: not supported (out of scope)
  sp->have_pending = true;
  sp->pend_black = black_ver; sp->pend_white = white_ver;
  sp->pend_thres = resign_thres; sp->pend_never = never_resign_prob;
  sp->pending_async = async != 0;
  if (!sp->search_open) {                              // between two moves: the move boundary is now
    DevGuard _dg(sp->eng->device);
    return sp_apply_request(sp);
  }
  return 0;
}

int elfsp_take_game_starts(ElfSelfPlay* sp, int64_t* black_ver, int64_t* white_ver) {
  if (!sp) return ELFGO_E_BADARG;
  const int n = sp->game_starts;
  sp->game_starts = 0;
  if (black_ver) *black_ver = sp->black_ver;
  if (white_ver) *white_ver = sp->white_ver;
  return n;
}

// the human half of GoGameSelfPlay::act (game_selfplay.cc:290-330): an externally chosen move is forwarded on the game board,
// the tree follows at the next search (MCTSAI_T::align_state -> advanceMoves, mcts.h:141-167).  moves_host[g] < 0 = no move.
int elfsp_play(ElfSelfPlay* sp, const int32_t* moves_host, void* stream) {
  if (!sp || !moves_host || sp->search_open) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  const int G = sp->G;
  std::vector<int32_t> ids, mv;
  for (int g = 0; g < G; ++g) if (moves_host[g] >= 0) { ids.push_back(g); mv.push_back(moves_host[g]); }
  if (ids.empty()) return 0;
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
  }
  if (sp->opt.persistent_tree) {
    HIPCHK(hipMemcpyAsync(sp->d_moves, adv.data(), 4 * G, hipMemcpyHostToDevice, sp->stream));
    SPCHK(elfmcts_advance(sp->mcts, sp->d_moves, sp->stream));
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
// given move list -- the first move_to moves are forwarded now, afterwards each search's move is replaced by the next SGF move
// (:392-405) and the game is finished (FR_MAX_STEP) by the search that finds the list exhausted.
int elfsp_preload(ElfSelfPlay* sp, const uint16_t* moves_host, int n, int move_to, void* stream) {
  if (!sp || n < 0 || (n > 0 && !moves_host) || sp->search_open || sp->n_moves != 0) return ELFGO_E_BADARG;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  sp->sgf.assign(moves_host, moves_host + n);
  sp->sgf_move_to = move_to;
  return sp_forward_preload(sp);
}

// finish_game(reason) + restart (game_selfplay.cc:121-149): the listed games are scored (setFinalValue, go_state_ext.h:76-103: FR_RESIGN
// = the side to move loses, every other reason = GoState::evaluate(komi)), leave their record and start over from the empty board
int elfsp_finish(ElfSelfPlay* sp, const int32_t* games_host, int n, int reason, void* stream) {
  if (!sp || n < 0 || n > sp->G || (n > 0 && !games_host) || sp->search_open) return ELFGO_E_BADARG;
  if (reason < ELFSP_FR_RESIGN || reason > ELFSP_FR_ILLEGAL) return ELFGO_E_BADARG;
  if (n == 0) return 0;
  DevGuard _dg(sp->eng->device);
  sp->stream = (hipStream_t)stream;
  for (int j = 0; j < n; ++j) if (games_host[j] < 0 || games_host[j] >= sp->G) return ELFGO_E_BADARG;
  HIPCHK(hipMemcpyAsync(sp->d_ids, games_host, 4 * n, hipMemcpyHostToDevice, sp->stream));
  SPCHK(elfgo_evaluate(sp->eng, sp->d_ids, n, sp->opt.mcts.komi, sp->d_val, sp->stream));   // setFinalValue: evaluate(komi)
  HIPCHK(hipMemcpyAsync(sp->h_val.data(), sp->d_val, sizeof(float) * n, hipMemcpyDeviceToHost, sp->stream));
  SPCHK(elfgo_reset(sp->eng, sp->d_ids, n, sp->stream));
  SPCHK(elfmcts_clear(sp->mcts, sp->d_ids, n, sp->stream));
  HIPCHK(hipStreamSynchronize(sp->stream));
  for (int j = 0; j < n; ++j) {
    SpGame& gm = sp->games[games_host[j]];
    float fv = sp->h_val[j];
    if (reason == ELFSP_FR_RESIGN) fv = ((gm.ply & 1) == 1) ? -1.0f : 1.0f;   // nextPlayer() == S_WHITE ? 1 : -1; ply 1 = Black to move
    sp->sum_final += fv;
    sp_finish_record(sp, games_host[j], fv, gm.ply);
    gm.ply = 1; gm.never_resign = false; gm.has_calculated_never_resign = false; gm.last_predicted = 0.0f;
    gm.seq++;
  }
  sp->n_games += n;
  return 0;
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
  SPCHK(elfmcts_node_visits(sp->mcts, &out[8]));      // synchronises the device
  uint64_t total_rows = 0;
  HIPCHK(hipMemcpy(&total_rows, sp->d_counts + 2, 8, hipMemcpyDeviceToHost));
  sp->n_rows = (int64_t)total_rows;
  out[0] = sp->n_moves; out[1] = sp->n_games; out[2] = sp->n_rollouts; out[3] = sp->n_rows; out[4] = sp->n_steps;
  out[5] = (int64_t)sp->log_search.size(); out[6] = sp->steps_per_move; out[7] = sp->step_in_move;
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
