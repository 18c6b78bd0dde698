// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Thin extern "C" wrapper around the *real* reference board engine
// (/root/reference/src_cpp/elfgames/go/base/{board,go_state,board_feature}.cc), compiled in
// place by oracle/Makefile into oracle/_ref/libelfref{19,9}.so.  Nothing from the reference is
// copied into this repository: this file only calls the reference's public functions.
// Used (a) to pin the C restatement in oracle/go_oracle.c, (b) to generate tests/golden/*,
// (c) as bench.py's cpu_baseline of kind "reference".
//
// Every ref_* entry point names the reference function it forwards to.
#include <cstdint>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "elfgames/go/base/board_feature.h"
#include "elfgames/go/base/go_state.h"
#include "elfgames/go/sgf/sgf.h"

extern "C" {

int ref_board_size() { return BOARD_SIZE; }
int ref_sizeof_board() { return (int)sizeof(Board); }

void* ref_new() { return new GoState(); }                        // go_state.h:96-98
void ref_free(void* s) { delete (GoState*)s; }
void ref_reset(void* s) { ((GoState*)s)->reset(); }              // go_state.cc:134-141
void* ref_clone(void* s) { return new GoState(*(GoState*)s); }   // go_state.h:117-124

// GoState::forward (go_state.cc:74-94). -1 when the reference throws (M_INVALID).
int ref_forward(void* s, int c) {
  try {
    return ((GoState*)s)->forward((Coord)c) ? 1 : 0;
  } catch (const std::range_error&) {
    return -1;
  }
}
int ref_check_move(void* s, int c) { return ((GoState*)s)->checkMove((Coord)c) ? 1 : 0; }  // go_state.cc:123-128
int ref_terminated(void* s) { return ((GoState*)s)->terminated() ? 1 : 0; }                 // go_state.h:145-147
int ref_ply(void* s) { return ((GoState*)s)->getPly(); }
int ref_next_player(void* s) { return ((GoState*)s)->nextPlayer(); }
int ref_last_move(void* s) { return ((GoState*)s)->lastMove(); }
uint64_t ref_hash(void* s) { return ((GoState*)s)->getHashCode(); }
float ref_evaluate(void* s, float komi) { return ((GoState*)s)->evaluate(komi); }           // go_state.h:194-203

// info[0..9] = ply, next_player, last_move, last_move2, ko_age, simple_ko, simple_ko_color, b_cap, w_cap, num_groups-1
void ref_info(void* s, int32_t* info) {
  const Board& b = ((GoState*)s)->board();
  info[0] = b._ply; info[1] = b._next_player; info[2] = b._last_move; info[3] = b._last_move2;
  info[4] = b._ko_age; info[5] = b._simple_ko; info[6] = b._simple_ko_color;
  info[7] = b._b_cap; info[8] = b._w_cap; info[9] = b._num_groups - 1;
}

// Legal-move mask in NN action order under D4 code 0 (action = x*N + y, pass last): what
// MCTSActor::pi2response filters with (go/mcts/mcts.h:300-312): checkMove(action2Coord(a)).
void ref_legal_mask(void* sp, uint8_t* mask) {
  GoState* s = (GoState*)sp;
  BoardFeature bf(*s);
  for (int a = 0; a < (int)BOARD_NUM_ACTION; ++a) mask[a] = s->checkMove(bf.action2Coord(a)) ? 1 : 0;
}

// Per-point colours and the liberty count of the group at that point (0 for empty), action order.
void ref_board(void* sp, uint8_t* colour, int16_t* libs) {
  const Board& b = ((GoState*)sp)->board();
  for (int x = 0; x < BOARD_SIZE; ++x)
    for (int y = 0; y < BOARD_SIZE; ++y) {
      Coord c = OFFSETXY(x, y);
      int a = EXPORT_OFFSET_XY(x, y);
      colour[a] = b._infos[c].color;
      libs[a] = b._infos[c].id ? b._groups[b._infos[c].id].liberties : 0;
    }
}

// BoardFeature::extractAGZ under D4 code d4 (board_feature.cc:247-290).
void ref_extract_agz(void* sp, int d4, float* out) {
  BoardFeature bf(*(GoState*)sp);
  bf.setD4Code(d4);
  bf.extractAGZ(out);
}
int64_t ref_coord2action(void* sp, int d4, int c) {
  BoardFeature bf(*(GoState*)sp);
  bf.setD4Code(d4);
  return bf.coord2Action((Coord)c);
}
int ref_action2coord(void* sp, int d4, int64_t a) {
  BoardFeature bf(*(GoState*)sp);
  bf.setD4Code(d4);
  return bf.action2Coord(a);
}
int ref_is_true_eye(void* sp, int c, int player) {
  return isTrueEye(&((GoState*)sp)->board(), (Coord)c, (Stone)player) ? 1 : 0;  // board.cc:1912-1914
}

// SGF -> main-line moves (sgf/sgf.h Sgf::load + iterator). Returns count, or -1 on load failure.
int ref_sgf_moves(const char* path, int32_t* moves, int32_t* players, int cap) {
  Sgf sgf;
  if (!sgf.load(path)) return -1;
  int n = 0;
  for (auto it = sgf.begin(); !it.done() && n < cap; ++it) {
    auto m = it.getCurrMove();
    moves[n] = m.move; players[n] = m.player; ++n;
  }
  return n;
}

// the same from a game string (Sgf::load(filename, game_string), sgf.cc:28-57) + the header fields; header5 = {size, komi, handi,
// winner (Stone), win_margin}
int ref_sgf_parse(const char* text, int32_t* moves, int32_t* players, int cap, float* header5) {
  Sgf sgf;
  if (!sgf.load("", std::string(text))) return -1;
  int n = 0;
  for (auto it = sgf.begin(); !it.done() && n < cap; ++it) {
    auto m = it.getCurrMove();
    moves[n] = m.move; players[n] = m.player; ++n;
  }
  if (header5) {
    const SgfHeader& h = sgf.getHeader();
    header5[0] = (float)h.size; header5[1] = h.komi; header5[2] = (float)h.handi; header5[3] = (float)h.winner; header5[4] = h.win_margin;
  }
  return n;
}

// ---- config-2 protocol: random legal non-true-eye play to game end -------------
// Counter-based RNG shared verbatim with the HIP kernel and oracle/go_oracle.c.
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
/* rand(seed, ply) = fmix32(key(seed) + ply * 0x9E3779B9), key(seed) = fmix32(lo) ^ fmix32(hi + 0x7F4A7C15) */
static inline uint32_t playout_rng(uint64_t seed, uint32_t t) {
  uint32_t key = fmix32((uint32_t)seed) ^ fmix32((uint32_t)(seed >> 32) + 0x7F4A7C15u);
  return fmix32(key + t * 0x9E3779B9u);
}

// Plays one game on *s from its current position. Returns number of successful forwards.
static int playout_one(GoState* s, uint64_t seed, int max_steps, float* feat, int with_feat) {
  int steps = 0;
  while (!s->terminated() && steps < max_steps) {
    AllMoves am;
    Stone p = s->nextPlayer();
    FindAllValidMoves(&s->board(), p, &am);  // board.cc:949-968 (x-major order)
    Coord cand[BOARD_SIZE * BOARD_SIZE];
    int n = 0;
    for (int i = 0; i < am.num_moves; ++i)
      if (!isTrueEye(&s->board(), am.moves[i], p)) cand[n++] = am.moves[i];
    Coord pick = M_PASS;
    if (n > 0) pick = cand[playout_rng(seed, (uint32_t)s->getPly()) % (uint32_t)n];
    if (!s->forward(pick)) break;
    ++steps;
    if (with_feat) {
      BoardFeature bf(*s);
      bf.setD4Code(playout_rng(seed ^ 0xD4D4D4D4ULL, (uint32_t)s->getPly()) % 8);
      bf.extractAGZ(feat);
    }
  }
  return steps;
}

// out[4*i+0..3] = hash_lo32, hash_hi32 (as u32), ply, steps for game i with seeds[i]; threads OS threads
// (reference threading model: one thread per game, elf/base/context.h:284-291).
int64_t ref_playout(const uint64_t* seeds, int n, int max_steps, int threads, int with_feat, uint32_t* out) {
  std::vector<std::thread> th;
  std::vector<int64_t> tot(threads, 0);
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&, t]() {
      std::vector<float> feat(18 * BOARD_SIZE * BOARD_SIZE);
      for (int i = t; i < n; i += threads) {
        GoState s;
        int steps = playout_one(&s, seeds[i], max_steps, feat.data(), with_feat);
        uint64_t h = s.getHashCode();
        out[4 * i + 0] = (uint32_t)h; out[4 * i + 1] = (uint32_t)(h >> 32);
        out[4 * i + 2] = (uint32_t)s.getPly(); out[4 * i + 3] = (uint32_t)steps;
        tot[t] += steps;
      }
    });
  for (auto& t : th) t.join();
  int64_t sum = 0;
  for (auto v : tot) sum += v;
  return sum;
}

// Same protocol on an existing state, also returning the move list (for replay through other engines).
int ref_playout_moves(void* sp, uint64_t seed, int max_steps, int32_t* moves) {
  GoState* s = (GoState*)sp;
  int steps = 0;
  while (!s->terminated() && steps < max_steps) {
    AllMoves am;
    Stone p = s->nextPlayer();
    FindAllValidMoves(&s->board(), p, &am);
    Coord cand[BOARD_SIZE * BOARD_SIZE];
    int n = 0;
    for (int i = 0; i < am.num_moves; ++i)
      if (!isTrueEye(&s->board(), am.moves[i], p)) cand[n++] = am.moves[i];
    Coord pick = M_PASS;
    if (n > 0) pick = cand[playout_rng(seed, (uint32_t)s->getPly()) % (uint32_t)n];
    if (!s->forward(pick)) break;
    moves[steps++] = pick;
  }
  return steps;
}

// The 441 / 121 Zobrist constants actually used by this build (hash_num.h:12, indexed by Coord).
void ref_zobrist(uint64_t* out) {
  for (int i = 0; i < BOUND_COORD; ++i) out[i] = _board_hash[i];
}

}  // extern "C"
