/* TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
 *
 * CPU restatement ("port") of the ELF OpenGo board engine for the self-play hot path.
 * Plain C, one translation unit, board size chosen at compile time (-DORC_N=19 or 9).
 * Written from the reference's observable semantics, not its data structures: groups and
 * liberties are recomputed from first principles (flood fill) after every move instead of the
 * reference's linked lists + incremental counters (SURVEY.md Appendix A "Empirical invariants").
 * Citations are relative to /root/reference/src_cpp/elfgames/go/.
 *
 * PARITY PINNED: tests/test_oracle.py checks this file against (a) tests/golden/ vectors generated
 * from the real reference build (oracle/gen_golden.py), (b) the reference's own 9x9 gtest known
 * answers (base/test/go_test.cc, board_feature_test.cc, symmetry_test.cc) and (c) when
 * oracle/_ref/libelfref*.so is present, the real reference move by move on random games.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_N
#define ORC_N 19
#endif
#define N ORC_N
#define S (N + 2)              /* base/board.h:45  BOARD_EXPAND_SIZE */
#define P (S * S)              /* base/board.h:98  BOUND_COORD */
#define NP (N * N)
#define NUM_ACTION (NP + 1)    /* base/go_common.h:11-12 */
#define MAX_MOVE (2 * NP)      /* base/go_common.h:15  BOARD_MAX_MOVE */
#define HIST 8                 /* base/board_feature.h:39 MAX_NUM_AGZ_HISTORY */

enum { S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2, S_OFF = 3 };            /* base/common.h:36-39 */
enum { M_PASS = 0, M_RESIGN = 1, M_SKIP = 2, M_INVALID = 3, M_CLEAR = 4 }; /* base/common.h:43-47 */

#define CX(c) ((c) % S - 1)                       /* base/board.h:178 */
#define CY(c) ((c) / S - 1)                       /* base/board.h:179 */
#define OFFSETXY(x, y) (((y) + 1) * S + (x) + 1)  /* base/board.h:183-184 */
#define OPP(p) (S_BLACK + S_WHITE - (p))          /* base/board.h:166 */
static const int delta4[4] = {-1, -S, +1, +S};    /* base/board.h:220 (L, T, R, B) */
static const int diag4[4] = {-1 - S, -1 + S, 1 - S, 1 + S}; /* base/board.h:223-226 */

static uint64_t g_zob[P];
static int g_zob_set = 0;

typedef struct {
  uint8_t color[P];
  uint64_t hash;
  int ply, next_player;
  int last_move[4];
  int ko_age, simple_ko, simple_ko_color;
  int b_cap, w_cap;
  /* GoState::_history: <= 8 most recent post-move positions, newest last (base/go_state.cc:90-92) */
  uint8_t hist[HIST][P];
  int hist_len;
  /* GoState::_board_hash: pre-move positions keyed by Zobrist hash (base/go_state.cc:113-121) */
  int sk_len;
  uint64_t sk_hash[MAX_MOVE + 2];
  uint8_t sk_img[MAX_MOVE + 2][P];
  /* derived, recomputed lazily */
  int dirty;
  int16_t label[P];
  int16_t libs[P]; /* indexed by label */
  int16_t stones[P];
} OrcState;

/* base/board.cc:24-36 transform_hash: white uses the hi/lo 32-bit swap of the black constant */
static uint64_t zob(int c, int s) {
  uint64_t h = g_zob[c];
  if (s == S_BLACK) return h;
  if (s == S_WHITE) return (h >> 32) | ((h & 0xFFFFFFFFULL) << 32);
  return 0;
}

/* base/board.cc:38-51 set_color (bit image is implied by color[] here) */
static void set_color(OrcState* st, int c, int s) {
  st->hash ^= zob(c, st->color[c]);
  st->color[c] = (uint8_t)s;
  st->hash ^= zob(c, s);
  st->dirty = 1;
}

/* Group labelling + liberty counts from first principles. Stands in for the reference's
 * incrementally maintained Group{stones,liberties} (base/board.h:71-76, board.cc:526-782). */
static void analyze(OrcState* st) {
  if (!st->dirty) return;
  int stack[P];
  uint8_t seen_lib[P];
  memset(st->label, 0, sizeof st->label);
  int next = 1;
  for (int c = 0; c < P; ++c) {
    if ((st->color[c] != S_BLACK && st->color[c] != S_WHITE) || st->label[c]) continue;
    int id = next++;
    int sp = 0, nlib = 0, nst = 0;
    memset(seen_lib, 0, sizeof seen_lib);
    stack[sp++] = c;
    st->label[c] = (int16_t)id;
    while (sp) {
      int u = stack[--sp];
      ++nst;
      for (int i = 0; i < 4; ++i) {
        int v = u + delta4[i];
        if (st->color[v] == S_EMPTY) {
          if (!seen_lib[v]) { seen_lib[v] = 1; ++nlib; }
        } else if (st->color[v] == st->color[c] && !st->label[v]) {
          st->label[v] = (int16_t)id;
          stack[sp++] = v;
        }
      }
    }
    st->libs[id] = (int16_t)nlib;
    st->stones[id] = (int16_t)nst;
  }
  st->dirty = 0;
}

/* ---------------------------------------------------------------- API ---------------------- */
void orc_set_zobrist(const uint64_t* z) { memcpy(g_zob, z, sizeof g_zob); g_zob_set = 1; }
int orc_board_size(void) { return N; }

/* base/board.cc:79-107 clearBoard + base/go_state.cc:134-141 GoState::reset */
void orc_reset(OrcState* st) {
  memset(st, 0, sizeof *st);
  for (int c = 0; c < P; ++c) {
    int x = CX(c), y = CY(c);
    st->color[c] = (x < 0 || x >= N || y < 0 || y >= N) ? S_OFF : S_EMPTY;
  }
  st->hash = 0; /* border/empty contribute 0 (board.cc:26-28) */
  st->next_player = S_BLACK;
  for (int i = 0; i < 4; ++i) st->last_move[i] = M_INVALID;
  st->ply = 1;
  st->dirty = 1;
}
OrcState* orc_new(void) {
  OrcState* st = (OrcState*)malloc(sizeof(OrcState));
  orc_reset(st);
  return st;
}
void orc_free(OrcState* st) { free(st); }
/* base/go_state.h:117-124 copy constructor */
OrcState* orc_clone(const OrcState* src) {
  OrcState* st = (OrcState*)malloc(sizeof(OrcState));
  memcpy(st, src, sizeof *st);
  return st;
}

/* base/board.cc:788-827 TryPlay (+ :161-240 StoneLibertyAnalysis / isSuicideMove / isSimpleKoViolation) */
static int try_play(OrcState* st, int c, int player) {
  int x = CX(c), y = CY(c);
  if (c == M_PASS || c == M_RESIGN) return 1;               /* :794-800 */
  if (c < 0 || c >= P) return 0;
  if (x < 0 || x >= N || y < 0 || y >= N) return 0;         /* :803 */
  if (st->color[c] != S_EMPTY) return 0;                    /* :808 */
  if (st->simple_ko == c && st->ko_age == 0 && st->simple_ko_color == player) return 0; /* :234-240 */
  analyze(st);
  int own_safe = 0, enemy_atari = 0;
  for (int i = 0; i < 4; ++i) {
    int v = c + delta4[i];
    if (st->color[v] == S_EMPTY) return 1;                  /* :203 liberty > 0 */
    if (st->color[v] == S_OFF) continue;
    int l = st->libs[st->label[v]];
    if (st->color[v] == player) { if (l > 1) ++own_safe; }  /* :213-215 */
    else { if (l == 1) ++enemy_atari; }                     /* :216-218 */
  }
  return (own_safe || enemy_atari) ? 1 : 0;                 /* :228-231 */
}

/* base/go_state.h:141-147 */
static int is_two_pass(const OrcState* st) { return st->last_move[0] == M_PASS && st->last_move[1] == M_PASS; }
/* base/go_state.cc:96-111 _check_superko */
static int check_superko(const OrcState* st) {
  if (st->last_move[0] == M_PASS) return 0;
  for (int i = 0; i < st->sk_len; ++i)
    if (st->sk_hash[i] == st->hash && memcmp(st->sk_img[i], st->color, P) == 0) return 1;
  return 0;
}
int orc_terminated(const OrcState* st) { return is_two_pass(st) || st->ply >= MAX_MOVE || check_superko(st); }

/* base/board.cc:1225-1238 update_next_move */
static void update_next_move(OrcState* st, int c, int player) {
  st->next_player = OPP(player);
  st->last_move[3] = st->last_move[2];
  st->last_move[2] = st->last_move[1];
  st->last_move[1] = st->last_move[0];
  st->last_move[0] = c;
  st->ply++;
}

/* base/board.cc:1297-1401 Play, restated: place, capture enemy neighbour groups left without
 * liberties (counts go to the mover's capture tally, :1348-1352), then simple-ko bookkeeping (:1384-1393). */
static void play(OrcState* st, int c, int player) {
  if (c == M_PASS || c == M_RESIGN) { update_next_move(st, c, player); return; }  /* :1306-1309 */
  set_color(st, c, player);
  analyze(st);
  int total_capture = 0, capture_c = 0;
  int dead[4], nd = 0;
  for (int i = 0; i < 4; ++i) {
    int v = c + delta4[i];
    if (st->color[v] != OPP(player)) continue;
    int id = st->label[v];
    if (st->libs[id] != 0) continue;
    int dup = 0;
    for (int j = 0; j < nd; ++j) dup |= (dead[j] == id);
    if (dup) continue;
    dead[nd++] = id;
    total_capture += st->stones[id];
    capture_c = v;                                          /* :1355 */
  }
  if (nd) {
    int16_t lab[P];
    memcpy(lab, st->label, sizeof lab);
    for (int u = 0; u < P; ++u)
      for (int j = 0; j < nd; ++j)
        if (lab[u] == dead[j]) set_color(st, u, S_EMPTY);   /* :526-572 EmptyGroup */
    if (player == S_BLACK) st->b_cap += total_capture; else st->w_cap += total_capture;
  }
  analyze(st);
  int id = st->label[c];
  if (st->libs[id] == 1 && st->stones[id] == 1 && total_capture == 1) {  /* :1386-1389 */
    st->simple_ko = capture_c;
    st->simple_ko_color = OPP(player);
    st->ko_age = 0;
  } else {
    st->ko_age++;                                           /* :1391 */
  }
  update_next_move(st, c, player);
}

/* base/go_state.cc:74-94 GoState::forward. Returns 1/0, -1 for M_INVALID (reference throws). */
int orc_forward(OrcState* st, int c) {
  if (c == M_INVALID) return -1;
  if (orc_terminated(st)) return 0;
  if (!try_play(st, c, st->next_player)) return 0;
  if (c != M_PASS) {                                        /* go_state.cc:113-121 _add_board_hash */
    st->sk_hash[st->sk_len] = st->hash;
    memcpy(st->sk_img[st->sk_len], st->color, P);
    st->sk_len++;
  }
  play(st, c, st->next_player);
  if (st->hist_len == HIST) {                               /* go_state.cc:90-92 */
    memmove(st->hist[0], st->hist[1], (HIST - 1) * P);
    st->hist_len--;
  }
  memcpy(st->hist[st->hist_len++], st->color, P);
  return 1;
}

/* base/go_state.cc:123-128 checkMove */
int orc_check_move(OrcState* st, int c) {
  if (c == M_INVALID) return 0;
  return try_play(st, c, st->next_player);
}

int orc_ply(const OrcState* st) { return st->ply; }
int orc_next_player(const OrcState* st) { return st->next_player; }
int orc_last_move(const OrcState* st) { return st->last_move[0]; }
uint64_t orc_hash(const OrcState* st) { return st->hash; }

/* same 10 fields as ref_info (group count from the labelling) */
void orc_info(OrcState* st, int32_t* info) {
  analyze(st);
  int ng = 0;
  for (int c = 0; c < P; ++c) if (st->label[c] > ng) ng = st->label[c];
  info[0] = st->ply; info[1] = st->next_player; info[2] = st->last_move[0]; info[3] = st->last_move[1];
  info[4] = st->ko_age; info[5] = st->simple_ko; info[6] = st->simple_ko_color;
  info[7] = st->b_cap; info[8] = st->w_cap; info[9] = ng;
}

/* ---- D4 symmetry and action map: base/board_feature.h:97-144 -------------------------------- */
static void d4_transform(int d4, int x, int y, int* ox, int* oy) {
  int rot = d4 % 4, flip = (d4 >> 2) == 1;
  int a = x, b = y;
  if (rot == 1) { a = y; b = N - x - 1; }                   /* CCW90  :100-101 */
  else if (rot == 2) { a = N - x - 1; b = N - y - 1; }      /* CCW180 :102-104 */
  else if (rot == 3) { a = N - y - 1; b = x; }              /* CCW270 :105-106 */
  if (flip) { int t = a; a = b; b = t; }                    /* :110-111 */
  *ox = a; *oy = b;
}
static void d4_inv_transform(int d4, int x, int y, int* ox, int* oy) {
  int rot = d4 % 4, flip = (d4 >> 2) == 1;
  int a = x, b = y;
  if (flip) { int t = a; a = b; b = t; }                    /* :118-119 */
  int c = a, d = b;
  if (rot == 1) { c = N - b - 1; d = a; }                   /* :121-122 */
  else if (rot == 2) { c = N - a - 1; d = N - b - 1; }      /* :123-125 */
  else if (rot == 3) { c = b; d = N - a - 1; }              /* :126-127 */
  *ox = c; *oy = d;
}
int64_t orc_coord2action(int d4, int c) {                   /* :132-137 */
  if (c == M_PASS) return NP;
  int x, y;
  d4_transform(d4, CX(c), CY(c), &x, &y);
  return (int64_t)x * N + y;                                /* board.h:189 EXPORT_OFFSET_XY */
}
int orc_action2coord(int d4, int64_t a) {                   /* :139-144 */
  if (a == -1 || a == NP) return M_PASS;
  int x, y;
  d4_inv_transform(d4, (int)(a / N), (int)(a % N), &x, &y);
  return OFFSETXY(x, y);
}

/* legal mask in action order under D4 code 0: go/mcts/mcts.h:300-312 (checkMove per action) */
void orc_legal_mask(OrcState* st, uint8_t* mask) {
  for (int a = 0; a < NUM_ACTION; ++a) mask[a] = (uint8_t)orc_check_move(st, orc_action2coord(0, a));
}

void orc_board(OrcState* st, uint8_t* colour, int16_t* libs) {
  analyze(st);
  for (int x = 0; x < N; ++x)
    for (int y = 0; y < N; ++y) {
      int c = OFFSETXY(x, y), a = x * N + y;
      colour[a] = st->color[c];
      libs[a] = st->label[c] ? st->libs[st->label[c]] : 0;
    }
}

/* base/board_feature.cc:247-290 extractAGZ: planes 2k/2k+1 = mover's / opponent's stones k
 * positions ago, 16 = all ones if Black to move, 17 = all ones if White to move. */
void orc_extract_agz(const OrcState* st, int d4, float* out) {
  memset(out, 0, sizeof(float) * 18 * NP);
  int player = st->next_player;
  for (int k = 0; k < st->hist_len; ++k) {
    const uint8_t* img = st->hist[st->hist_len - 1 - k];    /* rbegin → newest first (:266) */
    for (int x = 0; x < N; ++x)
      for (int y = 0; y < N; ++y) {
        int s = img[OFFSETXY(x, y)];
        if (s != S_BLACK && s != S_WHITE) continue;
        int tx, ty;
        d4_transform(d4, x, y, &tx, &ty);
        int plane = 2 * k + (s == player ? 0 : 1);
        out[plane * NP + tx * N + ty] = 1.0f;
      }
  }
  float* ind = out + (player == S_BLACK ? 16 : 17) * NP;    /* :284-289 */
  for (int i = 0; i < NP; ++i) ind[i] = 1.0f;
}

/* base/go_state.h:32-93 simple_flood_fill + simple_tt_scoring, :194-203 evaluate */
static void flood(const OrcState* st, int player, uint8_t* f) {
  int q[P * 5], qh = 0, qt = 0;
  uint8_t open[P];
  memset(open, 0, sizeof open);
  memset(f, 0, P);
  for (int c = 0; c < P; ++c) if (st->color[c] == player) q[qt++] = c;
  while (qh < qt) {
    int c = q[qh++];
    f[c] = 1;
    for (int i = 0; i < 4; ++i) {
      int v = c + delta4[i];
      if (st->color[v] == S_EMPTY && !open[v]) { open[v] = 1; q[qt++] = v; }
    }
  }
}
float orc_evaluate(const OrcState* st, float komi) {
  if (check_superko(st)) return st->next_player == S_BLACK ? 1.0f : -1.0f;
  uint8_t fb[P], fw[P];
  flood(st, S_BLACK, fb);
  flood(st, S_WHITE, fw);
  int bv = 0, wv = 0;
  for (int c = 0; c < P; ++c) {
    if (fb[c] && !fw[c]) ++bv;
    else if (fw[c] && !fb[c]) ++wv;
  }
  return (float)(bv - wv) - komi;
}

/* base/board.cc:1850-1860 isEye, :1887-1906 isFakeEye, :1912-1914 isTrueEye */
int orc_is_true_eye(const OrcState* st, int c, int player) {
  if (st->color[c] != S_EMPTY) return 0;
  for (int i = 0; i < 4; ++i) {
    int s = st->color[c + delta4[i]];
    if (s != player && s != S_OFF) return 0;
  }
  int nopp = 0, nb = 0;
  for (int i = 0; i < 4; ++i) {
    int s = st->color[c + diag4[i]];
    if (s == OPP(player)) ++nopp;
    else if (s == S_OFF) ++nb;
  }
  int fake = (nb > 0 && nopp >= 1) || (nb == 0 && nopp >= 2);
  return !fake;
}

/* ---- config-2 protocol, RNG shared verbatim with ref_capi.cc and the HIP kernel */
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
/* rand(seed, ply) = fmix32(key(seed) + ply * 0x9E3779B9), key(seed) = fmix32(lo) ^ fmix32(hi + 0x7F4A7C15) */
static inline uint32_t playout_rng(uint64_t seed, uint32_t t) {
  uint32_t key = fmix32((uint32_t)seed) ^ fmix32((uint32_t)(seed >> 32) + 0x7F4A7C15u);
  return fmix32(key + t * 0x9E3779B9u);
}
/* candidates enumerated x-major like FindAllValidMoves (base/board.cc:949-968), minus true eyes */
int orc_playout_moves(OrcState* st, uint64_t seed, int max_steps, int32_t* moves) {
  int steps = 0;
  while (!orc_terminated(st) && steps < max_steps) {
    int cand[NP], n = 0, p = st->next_player;
    for (int x = 0; x < N; ++x)
      for (int y = 0; y < N; ++y) {
        int c = OFFSETXY(x, y);
        if (st->color[c] != S_EMPTY) continue;
        if (!try_play(st, c, p)) continue;
        if (orc_is_true_eye(st, c, p)) continue;
        cand[n++] = c;
      }
    int pick = M_PASS;
    if (n > 0) pick = cand[playout_rng(seed, (uint32_t)st->ply) % (uint32_t)n];
    if (orc_forward(st, pick) != 1) break;
    if (moves) moves[steps] = pick;
    ++steps;
  }
  return steps;
}

/* ---- deterministic stub net shared with oracle/ref_selfplay.cc (oracle/stub_net.h) ---- */
#include "stub_net.h"
void orc_stub_net(const float* s, int batch, uint32_t salt, int tie_levels, float* pi, float* v) {
  stubnet_eval(s, batch, ORC_N, salt, tie_levels, pi, v);
}
