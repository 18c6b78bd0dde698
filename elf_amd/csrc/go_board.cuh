// Device-side Go board engine for gfx950 (CDNA4): one wave64 per board, position + group tables in LDS.
//
// Replaces, for the self-play hot path, the reference's CPU board engine
//   src_cpp/elfgames/go/base/board.cc   (TryPlay :788-827, Play :1297-1401, liberty/group upkeep :526-782)
//   src_cpp/elfgames/go/base/go_state.cc (GoState::forward :74-94, superko :96-121)
// Only reference-VISIBLE results are reproduced (legality, captures, liberties per group, simple-ko
// point/age/colour, Zobrist hash, 8-deep history, termination); group numbering and list order are
// internal to the reference and are not mirrored.
//
// Layout choices (all gfx950-specific):
//  * wave64, one wave = one board: the engine never synchronises across waves (its only fence is a wavefront-scope compiler
//    fence), every control decision is wave-uniform (readlane -> SGPR) so branches are scalar.  Kernels pack several such
//    waves into a workgroup where they share read-only tables in LDS (k_playout: Zobrist constants, division table).
//  * points live in LDS on a padded (N+2)^2 grid in x-major order so that NN action ids
//    (a = x*N + y, board.h:189) map to consecutive LDS addresses: idx = (x+1)*(N+2) + (y+1).
//    The reference Coord is the transpose, c = (y+1)*(N+2) + (x+1) (board.h:183); i<->c is an involution.
//  * pt[idx] (u16): 0 empty, 0x8000 border ("white, root 0"), else (white?0x8000:0) | root, root = idx of the
//    group's representative point (never 0).  libs[root] (u16) = liberties of that group; libs[0] stays 0, so
//    "libs[label & 0x7FFF]" is a valid, branch-free read for empty and border points too (it yields 0).
//    Lane l owns points a = 64*k + l, k < R (R = 6 for 19x19, 2 for 9x9).
//  * captures / merges / liberty recounts are wave-parallel scans over the R rounds with __ballot +
//    __popcll; liberty give-back after a capture uses LDS atomics.
//  * history = ring of 8 x {black,white} bitboards in action order (W u64 words each); superko keeps a Bloom
//    filter of every pre-move hash in LDS and the full (hash, bitboards) records in HBM, read only on a filter hit.
//  * the current position also lives in registers as two lane-distributed bitboards (lane k = actions [64k, 64k+64));
//    the whole legality test is bitboard algebra on them (dilate / funnel shifts on 32-bit halves, DPP for the carries).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// phase markers for tools/playout_phases.hip (cycle attribution); compiled out of the library
#ifndef ELF_PHASE
#define ELF_PHASE(bd, k)
#define ELF_PHASE_END(bd)
#endif

// wave-uniform conditions that are rarely true on the board step's critical path (a capture: 12 % of the steps; a merge of two or
// more groups; a Bloom hit; a live simple-ko point): the cold code moves out of the fall-through path (+1.7 % on k_playout;
// hinting the 27 % / 73 % atari-set refresh or the pick's "any candidate" test the same way costs 1.3 %)
#define ELF_RARE(x) __builtin_expect(!!(x), 0)

namespace elfgo {

typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

enum { S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2 };                       // base/common.h:36-38
enum { M_PASS = 0, M_RESIGN = 1, M_SKIP = 2, M_INVALID = 3, M_CLEAR = 4 };  // base/common.h:43-47
constexpr u16 PT_BORDER = 0x8000;  // colour bit with root 0: no stone ever carries it
constexpr int HIST = 8;  // base/board_feature.h:39 MAX_NUM_AGZ_HISTORY

// 64-byte header, wave-uniform; lives at the front of every board slot.
struct Hdr {
  u64 hash;         // Board::_hash (board.h:107)
  u16 ply;          // Board::_ply, starts at 1 (board.h:153)
  u16 ko_pt;        // Board::_simple_ko as reference Coord (board.h:145)
  u16 ko_age;       // Board::_ko_age (board.h:144)
  u16 last_move[4]; // _last_move.._last_move4 (board.h:127-130)
  u16 b_cap, w_cap; // board.h:120-121
  u16 hist_cnt;     // number of history pushes so far (ring position); len = min(cnt, 8)
  u16 sk_len;       // number of superko records (= non-pass forwards so far)
  unsigned char next_player;  // Board::_next_player
  unsigned char ko_color;     // Board::_simple_ko_color
  unsigned char superko;      // cached GoState::_check_superko() of the current position
  unsigned char pad0;
  u32 pad1[7];
};
static_assert(sizeof(Hdr) == 64, "Hdr must be 64 bytes");

template <int N>
struct Geo {
  static constexpr int S = N + 2;
  static constexpr int P = S * S;
  static constexpr int NP = N * N;
  static constexpr int NA = NP + 1;               // BOARD_NUM_ACTION (go_common.h:12)
  static constexpr int R = (NP + 63) / 64;        // rounds of 64 lanes / u64 words per bitboard
  static constexpr int PP = (P + 7) & ~7;         // u16 arrays padded to 16 B
  static constexpr int MAXMOVE = 2 * NP;          // BOARD_MAX_MOVE (go_common.h:15)
  static constexpr int SKW = 2 * R + 1;           // u64 words per superko record: hash, black words, white words
  static constexpr int BLOOM = N > 9 ? 256 : 64;  // u32 words of the superko Bloom filter
  static constexpr int ZOBW = PP;                 // zobrist table words; geometry masks follow it
  // floor(v / N) = (v * DN_M) >> DN_S for 0 <= v < 64*R, floor(v / S) = (v * DS_M) >> DS_S for 0 <= v < PP (exhaustively
  // checked by static_assert below): a wave-uniform multiply + shift instead of the compiler's general 32-bit sequences
  static constexpr int DN_M = N == 19 ? 27 : 57, DN_S = 9;
  static constexpr int DS_M = N == 19 ? 781 : 187, DS_S = N == 19 ? 14 : 11;
};
template <int N>
constexpr bool geo_div_ok() {
  using G = Geo<N>;
  for (int v = 0; v < 64 * G::R; ++v) if (((v * G::DN_M) >> G::DN_S) != v / N) return false;
  for (int v = 0; v < G::PP; ++v) if (((v * G::DS_M) >> G::DS_S) != v / G::S) return false;
  return true;
}
static_assert(geo_div_ok<19>() && geo_div_ok<9>(), "small-range division constants");

// One board slot = the LDS image, also the HBM image (copied 16 B per lane).
template <int N>
struct alignas(16) Slot {
  using G = Geo<N>;
  Hdr h;
  u16 pt[G::PP];
  u16 libs[G::PP];
  u64 hist[HIST][2][G::R];
  u32 bloom[G::BLOOM];
  static constexpr int RAW = 64 + 2 * G::PP * 2 + HIST * 2 * G::R * 8 + G::BLOOM * 4;
  static constexpr int PADB = ((RAW + 255) & ~255) - RAW;
  unsigned char pad[PADB > 0 ? PADB : 256];
};
static_assert(sizeof(Slot<19>) == 3840, "19x19 slot is 3.75 KiB");
static_assert(sizeof(Slot<19>) % 256 == 0 && sizeof(Slot<9>) % 256 == 0, "slot is 256-B granular");

// The board of a search-tree node in HBM: the slot WITHOUT its superko Bloom words (and padding) -- header, labels, liberties, history.
// A tree node needs no filter of its own: what the filter has to cover is the game's records up to the root (the game board's own
// Bloom words, one copy per game) plus the positions on the path root -> node, which a descent passes through anyway (k_mcts_select
// rebuilds the filter in LDS from those before it forwards).  2624 B instead of 3840 B at 19x19.  CBoard is a prefix of Slot.
template <int N>
struct alignas(16) CBoard {
  using G = Geo<N>;
  Hdr h;
  u16 pt[G::PP];
  u16 libs[G::PP];
  u64 hist[HIST][2][G::R];
};
static_assert(sizeof(CBoard<19>) == 2624 && sizeof(CBoard<9>) == 832, "compact node board");
static_assert(sizeof(CBoard<19>) % 16 == 0 && sizeof(CBoard<9>) % 16 == 0, "compact node board is copied 16 B per lane");
static_assert(offsetof(Slot<19>, bloom) == sizeof(CBoard<19>) && offsetof(Slot<9>, bloom) == sizeof(CBoard<9>), "CBoard is the slot's prefix");

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ u64 rl64(u64 v, int lane) {
  u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, lane), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}
// lane l receives the value of lane l-1 / l+1 (zero at the ends of a 16-lane row; only lanes < R <= 6 matter)
__device__ __forceinline__ u64 dpp_prev(u64 v) {
  u32 lo = __builtin_amdgcn_update_dpp(0u, (u32)v, 0x111, 0xf, 0xf, true);
  u32 hi = __builtin_amdgcn_update_dpp(0u, (u32)(v >> 32), 0x111, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 dpp_next(u64 v) {
  u32 lo = __builtin_amdgcn_update_dpp(0u, (u32)v, 0x101, 0xf, 0xf, true);
  u32 hi = __builtin_amdgcn_update_dpp(0u, (u32)(v >> 32), 0x101, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
// v_writelane_b32 through the LLVM intrinsic (this clang has no __builtin_amdgcn_writelane): the compiler picks an immediate or
// m0 lane select and handles the hazards itself.
extern "C" __device__ int elf_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane");
// X (lane-distributed) gets the wave-uniform 64-bit value `val` in lane k (wave-uniform): two v_writelane_b32 instead of
// compare + two selects through VGPR copies.
__device__ __forceinline__ void set_lane64(u64& X, int k, u64 val) {
  const u32 lo = (u32)elf_llvm_writelane((int)(u32)val, k, (int)(u32)X);
  const u32 hi = (u32)elf_llvm_writelane((int)(u32)(val >> 32), k, (int)(u32)(X >> 32));
  X = ((u64)hi << 32) | lo;
}
__device__ __forceinline__ bool lane_bit(u64 uniform_mask) { return __builtin_amdgcn_inverse_ballot_w64(uniform_mask); }
// XOR over the 64 lanes on the DPP network (same butterfly as wave_max_u32 in mcts.cuh: no LDS crossbar round trips);
// every lane of the last row -- and lane 63 in particular -- ends with the total, returned wave-uniform.  EXEC must be full.
__device__ __forceinline__ u64 wave_xor64(u64 v) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
#define ELF_DPP_XOR(ctrl, rmask)                                                     \
  lo ^= (u32)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, rmask, 0xf, false);       \
  hi ^= (u32)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, rmask, 0xf, false);
  ELF_DPP_XOR(0xB1, 0xf)    // quad_perm [1,0,3,2]
  ELF_DPP_XOR(0x4E, 0xf)    // quad_perm [2,3,0,1]
  ELF_DPP_XOR(0x124, 0xf)   // row_ror:4
  ELF_DPP_XOR(0x128, 0xf)   // row_ror:8   -> every lane of a row holds the row's XOR
  ELF_DPP_XOR(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
  ELF_DPP_XOR(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> row 3 holds the total
#undef ELF_DPP_XOR
  lo = (u32)__builtin_amdgcn_readlane((int)lo, 63);
  hi = (u32)__builtin_amdgcn_readlane((int)hi, 63);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ bool is_stone(u32 v) { return (v & 0x7FFFu) != 0; }   // neither empty (0) nor border (0x8000)
// Wave masks straight from a vector compare (v_cmp -> SGPR pair).  __ballot(a && b) makes the compiler rebuild the mask through
// v_cndmask + v_cmp_ne; "bal_xx(..) & bal_yy(..)" is two v_cmp and one s_and_b64.  Inactive lanes contribute 0, like __ballot.
__device__ __forceinline__ u64 bal_eq(u32 a, u32 b) { return __builtin_amdgcn_uicmp(a, b, 32); }
__device__ __forceinline__ u64 bal_ne(u32 a, u32 b) { return __builtin_amdgcn_uicmp(a, b, 33); }
__device__ __forceinline__ u64 bal_gt(u32 a, u32 b) { return __builtin_amdgcn_uicmp(a, b, 34); }   // unsigned a > b
__device__ __forceinline__ u64 bal_le(u32 a, u32 b) { return __builtin_amdgcn_uicmp(a, b, 37); }   // unsigned a <= b
__device__ __forceinline__ u64 bal_ne64(u64 a, u64 b) { return __builtin_amdgcn_uicmpl(a, b, 33); }   // 64-bit operands: uicmpl (uicmp would truncate)
// base/board.cc:24-36 transform_hash
__device__ __forceinline__ u64 zob_col(u64 h, int s) { return s == S_BLACK ? h : ((h >> 32) | (h << 32)); }

// Scalar (SMEM) load of a 64-bit table entry at a wave-uniform index.  The played point's Zobrist constant is read this way:
// SMEM completion is tracked by lgkmcnt, so consuming it does not have to wait on vmcnt -- and on gfx9 vmcnt also counts the
// fire-and-forget superko record STORES issued in between (an `s_waitcnt vmcnt(0)` for a vector load of this constant was found
// to stall ~1 200 cycles per board step behind those stores).  A constant-address-space load with a wave-uniform index: the
// compiler emits s_load_dwordx2 and places the lgkmcnt wait itself (an inline-asm s_load + `s_waitcnt lgkmcnt(0)` also drained
// every LDS operation in flight at the point of use).
__device__ __forceinline__ u64 sload_u64(const u64* base, int uniform_index) {
  typedef const __attribute__((address_space(4))) u64 cu64;
  return reinterpret_cast<cu64*>(reinterpret_cast<uintptr_t>(base))[uniform_index];
}
__device__ __forceinline__ u64 sload_wait(u64 v) { return v; }

// config-2 counter RNG ("a counter-based RNG shared verbatim by CPU and GPU harness"); oracle/go_oracle.c and
// oracle/ref_capi.cc restate the same two functions.  rand(seed, ply) = fmix32(key(seed) + ply * 0x9E3779B9) with the 32-bit
// murmur3 finaliser: two 32-bit multiplies on the per-step chain (a splitmix64 round costs eleven on a 32-bit scalar ALU, and
// the pick sits on the critical path of a board step); the per-board key is loop-invariant.
__device__ __forceinline__ u32 fmix32(u32 h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ u32 playout_key(u64 seed) { return fmix32((u32)seed) ^ fmix32((u32)(seed >> 32) + 0x7F4A7C15u); }
__device__ __forceinline__ u32 playout_rng_k(u32 key, u32 t) { return fmix32(key + t * 0x9E3779B9u); }
__device__ __forceinline__ u32 playout_rng(u64 seed, u32 t) { return playout_rng_k(playout_key(seed), t); }

// Superko record store of a GAME board (GoState::_board_hashes, go_state.h:218): every pre-move position of the game so far,
// one record per non-pass forward, in HBM: rec[MAXMOVE+2][SKW] u64 = {hash, black words [R], white words [R]} -- 104 B at
// 19x19, contiguous, so one record is ONE store instruction (lanes 0..2R) and a run of records is one contiguous block.
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
  u32 lo = __builtin_amdgcn_update_dpp(0u, (u32)v, CTRL, 0xf, 0xf, true);
  u32 hi = __builtin_amdgcn_update_dpp(0u, (u32)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
// the record as one word per lane: lane 0 hash, lanes 1..R black words, lanes R+1..2R white words (bitboards are
// lane-distributed over lanes 0..R-1 and zero elsewhere: row_shr:1 / row_shr:R+1 move them into place)
template <int N>
__device__ __forceinline__ u64 sk_record_word(u64 hash, u64 Bw, u64 Ww, int lane) {
  constexpr int R = Geo<N>::R;
  const u64 w = dpp_u64<0x110 + 1>(Bw) | dpp_u64<0x110 + R + 1>(Ww);
  return lane == 0 ? hash : w;
}

template <int N>
struct GameSK {
  using G = Geo<N>;
  u64* rec;   // [MAXMOVE+2][SKW]
  __device__ __forceinline__ void record(int sk_len, u64 hash, u64 Bw, u64 Ww, int lane) const {
    const u64 w = sk_record_word<N>(hash, Bw, Ww, lane);
    if (lane < G::SKW) rec[(size_t)sk_len * G::SKW + lane] = w;
  }
  // go_state.cc:96-111: hash match, then full image compare, over `cnt` records starting at `base`
  __device__ __forceinline__ static bool scan(const u64* base, int cnt, u64 hash, u64 w, int lane) {
    bool hit = false;
    for (int b0 = 0; b0 < cnt; b0 += 64) {
      const int t = b0 + lane;
      u64 bal = __ballot(t < cnt && base[(size_t)t * G::SKW] == hash);
      while (bal) {
        const int tt = b0 + (int)__builtin_ctzll(bal);
        bal &= bal - 1;
        bool same = true;
        if (lane >= 1 && lane < G::SKW) same = base[(size_t)tt * G::SKW + lane] == w;
        if (__all(same)) hit = true;
      }
    }
    return hit;
  }
  __device__ __forceinline__ bool exact_hit(int sk_len, u64 hash, u64 Bw, u64 Ww, int lane) const {
    return scan(rec, sk_len, hash, sk_record_word<N>(hash, Bw, Ww, lane), lane);
  }
};

template <int N>
struct Board {
  using G = Geo<N>;
  static constexpr int S = G::S, NP = G::NP, R = G::R;

  Slot<N>* L;          // LDS image of this wave's board
  const u64* zob;      // Zobrist constants in INTERNAL index order (global memory) + geometry masks
  bool zob_lds_on;     // zob_v points at an LDS copy (k_playout, k_replay_extract): every Zobrist read of forward goes there
  const u64* zob_v;    // the same constants for per-lane (vector) reads: global memory by default; k_playout points it at a
                       // copy in LDS, because on gfx9 a vector load's vmcnt wait also waits for every superko record store
                       // issued before it (a capture stalled ~2 board steps' worth of time behind those stores)
  u64* sk_rec;         // this board's superko records  [MAXMOVE+2][SKW]   (HBM)
  int lane;
  int idx[R];          // LDS index of this lane's point in round k (clamped for invalid lanes)
  int dl4;             // lanes 0..3: LDS offset of the neighbour in delta4 order (dir4(lane & 3))
  int kk3, off12;      // lanes 0..11: neighbour kk3 = lane / 3 and the offset from the played point to that neighbour's
                       // jj-th side not facing it (MergeToGroup liberty test)
  // lane-distributed bitboards in NN action order: lane k < R holds bits [64k, 64k+64); other lanes 0
  u64 Bw, Ww;          // black / white stones of the current position
  u64 mValid, mEdge;   // per-lane geometry masks (lanes 0..R-1, 0 elsewhere): a < N*N, point on the first/last line
  // y != 0, y != N-1, a < N*N for the shifts.  These live in lanes 0..R-1 AND again in lanes 8..8+R-1: a second bitboard parked
  // in lanes 8.. of the same row rides through dilate / the shifts for free (dilate2, eye test); the DPP row shifts never mix
  // the two because lanes R..7 and 8+R..15 hold 0
  u64 mTop, mBot, pValid;
  // wave-uniform copy of the header (SGPRs); LDS/HBM copy is refreshed by store_hdr()
  u64 hash;
  int ply, next_player, ko_pt, ko_age, ko_color, lm0, lm1, lm2, lm3, b_cap, w_cap, hist_cnt, sk_len, superko;
  int ko_a;            // action id of ko_pt (derived; valid whenever ko_pt != 0)
  // k_playout only: the atari set of the current position (stones of groups with exactly one liberty), carried from one
  // legal_moves to the next; forward_legal_action updates it in place when the move changed it by at most the played stone
  // and marks it dirty otherwise (a capture, a neighbour group falling to one liberty, the mover's group changing status)
  u64 at_cache;
  int at_dirty;
  u64* sk_wp;          // where this lane's word of the next superko record goes when the record is written through the running
                       // pointer (forward_legal_action); kept in step with sk_len by load() / reset() / the general forward
  int nb_addr, t12_off;   // per-lane LDS byte offsets for the neighbour reads of forward (see init)
#ifdef ELF_PROFILE
  unsigned long long ph_t, ph_acc[16];   // tools/playout_phases.hip
#endif

  __device__ __forceinline__ static int a2i(int a) { return a + S + 1 + 2 * (a / N); }
  __device__ __forceinline__ static int tr(int i) { return (i % S) * S + i / S; }  // idx <-> reference Coord
  __device__ __forceinline__ static int i2a(int i) { return (i / S - 1) * N + (i % S - 1); }
  // the same maps for wave-uniform in-range arguments, division by multiply + shift (Geo::DN_M / DS_M)
  __device__ __forceinline__ static int div_n(int a) { return (int)(((u32)a * (u32)G::DN_M) >> G::DN_S); }   // a / N, 0 <= a < 64 R
  __device__ __forceinline__ static int div_s(int i) { return (int)(((u32)i * (u32)G::DS_M) >> G::DS_S); }   // i / S, 0 <= i < PP
  __device__ __forceinline__ static int a2i_u(int a) { return a + S + 1 + 2 * div_n(a); }
  __device__ __forceinline__ static int a2c_u(int a) { const int x = div_n(a), y = a - x * N; return (y + 1) * S + (x + 1); }   // action -> reference Coord
  __device__ __forceinline__ static int c2a_u(int c) { const int y1 = div_s(c), x1 = c - y1 * S; return (x1 - 1) * N + (y1 - 1); }  // on-board Coord -> action
  // reference delta4 order L,T,R,B = x-1,y-1,x+1,y+1 (board.h:220) -> internal -S,-1,+S,+1
  __device__ __forceinline__ static int dir4(int q) { return (q & 1) ? ((q & 2) ? 1 : -1) : ((q & 2) ? S : -S); }
  // single-wave workgroup: LDS ops of one wave execute in order, so a wavefront-scope fence (no
  // instructions, compiler ordering only) is all the synchronisation the engine needs -- in particular
  // no s_waitcnt vmcnt(0) behind the fire-and-forget superko stores to HBM.
  __device__ __forceinline__ static void wsync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

  __device__ __forceinline__ void init(Slot<N>* lds, const u64* z, u64* skr) { init(lds, z, skr, threadIdx.x & 63); }
  // the lane index as an argument: a kernel that wants the per-lane constants set up late (k_mcts_select: after the descent, so that
  // they are not live across its level loop) passes an opaque copy of its lane index, which keeps the set-up where it is written
  __device__ __forceinline__ void init(Slot<N>* lds, const u64* z, u64* skr, int lane_) {
    L = lds; zob = z; zob_v = z; zob_lds_on = false; sk_rec = skr;
    lane = lane_;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      int a = k * 64 + lane;
      idx[k] = a2i(a < NP ? a : 0);
    }
    dl4 = dir4(lane & 3);
    kk3 = (lane < 12 ? lane : 0) / 3;
    off12 = dir4(kk3) + dir4((kk3 + 3 + ((lane < 12 ? lane : 0) - 3 * kk3)) & 3);
    // branch-free neighbour reads of the played point i (all 64 lanes issue the read, no EXEC games):
    //   lanes 0..3  : pt[i + dir4(lane)]              byte offset = pt + 2 i + nb_addr
    //   lanes >= 4  : libs[0], which is always 0      ("an empty non-neighbour": label 0, liberties 0)
    //   lanes 0..11 : pt[i + off12] for the MergeToGroup test; other lanes re-read pt[i]
    nb_addr = 2 * dl4;
    t12_off = lane < 12 ? 2 * off12 : 0;
    ko_a = 0; at_cache = 0; at_dirty = 1;
    const u64* geo = z + G::ZOBW;
    const int gl = lane & 7;
    const bool g2 = lane < 16 && gl < R;
    mTop = g2 ? geo[gl * 4 + 0] : 0ull;
    mBot = g2 ? geo[gl * 4 + 1] : 0ull;
    pValid = g2 ? geo[gl * 4 + 2] : 0ull;
    mValid = lane < R ? geo[lane * 4 + 2] : 0ull;
    mEdge = lane < R ? geo[lane * 4 + 3] : 0ull;
    Bw = Ww = 0;
  }

  // The word shifts of a lane-distributed bitboard on 32-bit halves.  Only two values cross lanes: the previous word's high half
  // (it supplies the bits a left shift pulls in: shifts are < 32) and the next word's low half; every shifted half is then ONE
  // v_alignbit_b32 (a funnel shift {a, b} >> s) instead of a 64-bit shift + 32-bit shift + or.
  struct Halves { u32 lo, hi, phi, nlo; };
  __device__ __forceinline__ static Halves halves(u64 X) {
    Halves h;
    h.lo = (u32)X; h.hi = (u32)(X >> 32);
    h.phi = (u32)__builtin_amdgcn_update_dpp(0, (int)h.hi, 0x111, 0xf, 0xf, true);   // row_shr:1, 0 into lane 0 of a row
    h.nlo = (u32)__builtin_amdgcn_update_dpp(0, (int)h.lo, 0x101, 0xf, 0xf, true);   // row_shl:1
    return h;
  }
  __device__ __forceinline__ static u64 join(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
  // (X << s) | (prev >> (64 - s)) and (X >> s) | (next << (64 - s)), 0 < s < 32
  template <int SH> __device__ __forceinline__ static u64 shl_w(const Halves& h) {
    return join(__builtin_amdgcn_alignbit(h.lo, h.phi, 32 - SH), __builtin_amdgcn_alignbit(h.hi, h.lo, 32 - SH));
  }
  template <int SH> __device__ __forceinline__ static u64 shr_w(const Halves& h) {
    return join(__builtin_amdgcn_alignbit(h.hi, h.lo, SH), __builtin_amdgcn_alignbit(h.nlo, h.hi, SH));
  }
  // points adjacent to X (4-neighbourhood), X lane-distributed
  __device__ __forceinline__ u64 dilate(u64 X) const {
    const Halves h = halves(X);
    u64 d = shl_w<1>(h) & mTop;                     // a-1 in X (same column run)
    d |= shr_w<1>(h) & mBot;                        // a+1 in X
    d |= shl_w<N>(h);                               // a-N in X
    d |= shr_w<N>(h);                               // a+N in X
    return d & pValid;
  }
  // two dilations for the price of one: B rides in lanes 8..8+R-1.  dA is valid in lanes 0..R-1 (its lanes 8.. hold dilate(B):
  // AND it with a lanes-0..R-1 bitboard before use); dB is clean.
  __device__ __forceinline__ void dilate2(u64 A, u64 B, u64& dA, u64& dB) const {
    dA = dilate(A | dpp_u64<0x118>(B));   // row_shr:8
    dB = dpp_u64<0x108>(dA);              // row_shl:8
  }

  // single-direction shifts of a lane-distributed bitboard: result bit a = X bit (a -/+ 1) within the column run, (a -/+ N)
  __device__ __forceinline__ u64 sh_m1(u64 X) const { return shl_w<1>(halves(X)) & mTop; }
  __device__ __forceinline__ u64 sh_p1(u64 X) const { return shr_w<1>(halves(X)) & mBot; }
  __device__ __forceinline__ u64 sh_mN(u64 X) const { return shl_w<N>(halves(X)); }
  __device__ __forceinline__ u64 sh_pN(u64 X) const { return shr_w<N>(halves(X)) & pValid; }

  // k_playout, after load(): point the per-lane Zobrist reads at the workgroup's LDS copy of the constants (forward_legal_action
  // reads the played stone's constant per lane; from global memory that read would queue behind the superko record stores).
  // The running record pointer of forward_legal_action is armed by load() / reset() themselves.
  __device__ __forceinline__ void playout_begin(const u64* zob_lds) { zob_v = zob_lds; zob_lds_on = true; }
  __device__ __forceinline__ void arm_record_pointer() { sk_wp = sk_rec ? sk_rec + (size_t)sk_len * G::SKW + lane : nullptr; }
  __device__ __forceinline__ void load_hdr() {
    u32 w = lane < 16 ? reinterpret_cast<const u32*>(&L->h)[lane] : 0u;
    u32 w0 = rl(w, 0), w1 = rl(w, 1), w2 = rl(w, 2), w3 = rl(w, 3), w4 = rl(w, 4), w5 = rl(w, 5), w6 = rl(w, 6),
        w7 = rl(w, 7), w8 = rl(w, 8);
    hash = ((u64)w1 << 32) | w0;
    ply = w2 & 0xFFFF; ko_pt = w2 >> 16;
    ko_age = w3 & 0xFFFF; lm0 = w3 >> 16;
    lm1 = w4 & 0xFFFF; lm2 = w4 >> 16;
    lm3 = w5 & 0xFFFF; b_cap = w5 >> 16;
    w_cap = w6 & 0xFFFF; hist_cnt = w6 >> 16;
    sk_len = w7 & 0xFFFF; next_player = (w7 >> 16) & 0xFF; ko_color = w7 >> 24;
    superko = w8 & 0xFF;
    ko_a = ko_pt ? c2a_u(ko_pt) : 0;
    at_dirty = 1;
    arm_record_pointer();
  }
  __device__ __forceinline__ void store_hdr() {
    if (lane == 0) {
      Hdr& h = L->h;
      h.hash = hash; h.ply = (u16)ply; h.ko_pt = (u16)ko_pt; h.ko_age = (u16)ko_age;
      h.last_move[0] = (u16)lm0; h.last_move[1] = (u16)lm1; h.last_move[2] = (u16)lm2; h.last_move[3] = (u16)lm3;
      h.b_cap = (u16)b_cap; h.w_cap = (u16)w_cap; h.hist_cnt = (u16)hist_cnt; h.sk_len = (u16)sk_len;
      h.next_player = (unsigned char)next_player; h.ko_color = (unsigned char)ko_color;
      h.superko = (unsigned char)superko;
    }
    wsync();
  }

  // ---- slot <-> HBM, 16 B per lane, fully coalesced ------------------------------------------
  __device__ __forceinline__ void load(const Slot<N>* g) {
    const uint4* s = reinterpret_cast<const uint4*>(g);
    uint4* d = reinterpret_cast<uint4*>(L);
#pragma unroll
    for (int k = 0; k < ((int)(sizeof(Slot<N>) / 16) + 63) / 64; ++k) {
      const int j = k * 64 + lane;
      if (j < (int)(sizeof(Slot<N>) / 16)) d[j] = s[j];
    }
    wsync();
    load_hdr();
    const int newest = (hist_cnt + HIST - 1) & (HIST - 1);
    Bw = (lane < R && hist_cnt != 0) ? L->hist[newest][0][lane] : 0ull;
    Ww = (lane < R && hist_cnt != 0) ? L->hist[newest][1][lane] : 0ull;
  }
  __device__ __forceinline__ void store(Slot<N>* g) {
    store_hdr();
    const uint4* s = reinterpret_cast<const uint4*>(L);
    uint4* d = reinterpret_cast<uint4*>(g);
#pragma unroll
    for (int k = 0; k < ((int)(sizeof(Slot<N>) / 16) + 63) / 64; ++k) {
      const int j = k * 64 + lane;
      if (j < (int)(sizeof(Slot<N>) / 16)) d[j] = s[j];
    }
  }
  // the same for a tree node's compact board: the Bloom words of the LDS image are left alone (the caller owns them)
  __device__ __forceinline__ void load(const CBoard<N>* g) {
    const uint4* s = reinterpret_cast<const uint4*>(g);
    uint4* d = reinterpret_cast<uint4*>(L);
#pragma unroll
    for (int k = 0; k < ((int)(sizeof(CBoard<N>) / 16) + 63) / 64; ++k) {
      const int j = k * 64 + lane;
      if (j < (int)(sizeof(CBoard<N>) / 16)) d[j] = s[j];
    }
    wsync();
    load_hdr();
    const int newest = (hist_cnt + HIST - 1) & (HIST - 1);
    Bw = (lane < R && hist_cnt != 0) ? L->hist[newest][0][lane] : 0ull;
    Ww = (lane < R && hist_cnt != 0) ? L->hist[newest][1][lane] : 0ull;
  }
  __device__ __forceinline__ void store(CBoard<N>* g) {
    store_hdr();
    const uint4* s = reinterpret_cast<const uint4*>(L);
    uint4* d = reinterpret_cast<uint4*>(g);
#pragma unroll
    for (int k = 0; k < ((int)(sizeof(CBoard<N>) / 16) + 63) / 64; ++k) {
      const int j = k * 64 + lane;
      if (j < (int)(sizeof(CBoard<N>) / 16)) d[j] = s[j];
    }
  }

  // base/board.cc:79-107 clearBoard + base/go_state.cc:134-141 reset, straight into LDS
  __device__ __forceinline__ void reset() {
    u32* w = reinterpret_cast<u32*>(L);
    for (int j = lane; j < (int)(sizeof(Slot<N>) / 4); j += 64) w[j] = 0;
    wsync();
    for (int j = lane; j < G::PP; j += 64) {
      int a = j / S, b = j % S;
      bool on = j < G::P && a >= 1 && a <= N && b >= 1 && b <= N;
      L->pt[j] = on ? 0 : PT_BORDER;
    }
    hash = 0; ply = 1; next_player = S_BLACK; ko_pt = 0; ko_age = 0; ko_color = 0;
    lm0 = lm1 = lm2 = lm3 = M_INVALID; b_cap = w_cap = 0; hist_cnt = 0; sk_len = 0; superko = 0;
    ko_a = 0; at_cache = 0; at_dirty = 1;
    arm_record_pointer();
    Bw = Ww = 0;
    wsync();
  }

  // base/go_state.h:141-147 terminated()
  __device__ __forceinline__ bool terminated() const {
    return (lm0 == M_PASS && lm1 == M_PASS) || ply >= G::MAXMOVE || superko;
  }

  __device__ __forceinline__ static void atomic_inc_u16(u16* p) {
    // LDS has no 16-bit atomic add: add into the containing dword (never carries: liberties < 2^15)
    // the slot pointer is generic: say "LDS" explicitly so that this is a ds_add_u32, not a flat atomic
    const u32 a = (u32)reinterpret_cast<size_t>((__attribute__((address_space(3))) void*)p);
    __attribute__((address_space(3))) u32* w = (__attribute__((address_space(3))) u32*)(size_t)(a & ~3u);
    __hip_atomic_fetch_add(w, (a & 2) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  // fire-and-forget ds_or_b32 on a dword of this wave's slot
  __device__ __forceinline__ static void lds_or(u32* p, u32 bits) {
    __attribute__((address_space(3))) u32* w = (__attribute__((address_space(3))) u32*)p;
    __hip_atomic_fetch_or(w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int popc_lanes(u64 X) const {   // population of a lane-distributed bitboard
    int c = __popcll(X), t = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) t += rl(c, k);
    return t;
  }

  // ---- GoState::forward (go_state.cc:74-94). c = reference Coord, wave-uniform. ---------------
  // returns 1 played, 0 refused (terminated / illegal). M_INVALID is rejected by the caller.
  __device__ int forward(int c) {
    GameSK<N> sk{sk_rec};
    return forward(c, sk);
  }
  template <class SK>
  __device__ int forward(int c, const SK& sk) { return forward_impl<false>(c, 0, sk); }
  // a move the caller took from legal_moves() of THIS position (k_playout): action id a, neither pass nor resign.  TryPlay's
  // verdict is known, so the Coord decode, the occupancy / simple-ko / suicide tests and the terminated() test are skipped;
  // the neighbour analysis that Play needs is not.
  template <class SK>
  __device__ int forward_legal_action(int a, const SK& sk) { return forward_impl<true>(0, a, sk); }
  template <bool TRUSTED, class SK>
  __device__ int forward_impl(int c, int a_trusted, const SK& sk) {
    if (!TRUSTED) c = rfl(c);
    if (!TRUSTED && terminated()) return 0;
    const int player = next_player, opp = S_BLACK + S_WHITE - player;
    const bool is_move = TRUSTED || !(c == M_PASS || c == M_RESIGN);
    const u64 mblack = player == S_BLACK ? ~0ull : 0ull;   // wave-uniform colour mask: X_black |= v & mblack, X_white |= v & ~mblack
    int i = 0, ka = 0, a = 0;
    u32 iv = 0;           // i as a vector value (same in every lane)
    u64 abit = 0, zi = 0;
    u32 nv = 0, nl = 0;   // lanes 0..3: label of the neighbour in delta4 order / liberties of its group; other lanes 0
    u32 emp4 = 0;         // bit j: neighbour j is empty
    if (is_move) {
      // ---- TryPlay, board.cc:788-827
      if (TRUSTED) {
        a = rfl(a_trusted);
        // the ds_read address of the neighbours hangs on i: do the arithmetic on the vector ALU (24-bit multiplies issue at
        // full rate; the scalar ALU's 32-bit multiply is several times slower) from a vector copy of the wave-uniform a
        u32 av;
        asm("v_mov_b32 %0, %1" : "=v"(av) : "s"(a));
        const u32 xv = __umul24(av, (u32)G::DN_M) >> G::DN_S, yv = av - __umul24(xv, (u32)N);
        iv = av + S + 1 + 2 * xv;
        i = rfl((int)iv);
        c = rfl((int)(__umul24(yv + 1, (u32)S) + xv + 1));
      } else {
        if (c < 0 || c >= G::P) return 0;
        const int yq = div_s(c);                     // c / S by multiply + shift (0 <= c < P)
        int x = c - yq * S - 1, y = yq - 1;
        if (x < 0 || x >= N || y < 0 || y >= N) return 0;                     // :803
        i = (x + 1) * S + (y + 1);
        a = x * N + y;
        iv = (u32)i;
      }
      ka = a >> 6; abit = 1ull << (a & 63);
      if (!TRUSTED) {
        if (rl64(Bw | Ww, ka) & abit) return 0;                               // :808 occupied
        if (ko_pt == c && ko_age == 0 && ko_color == player) return 0;        // :234-240
      }
      // the played point's Zobrist constant, hashed in after Play.  k_playout: from the LDS copy (every lane reads the same
      // address; with no scalar load left in the loop every LDS wait is an exact lgkmcnt(n)); elsewhere through the scalar cache
      if (TRUSTED || zob_lds_on) zi = zob_v[iv];
      else zi = sload_u64(zob, i);
      // StoneLibertyAnalysis :161-199, branch-free: every lane reads (lanes >= 4 read libs[0] = 0 as their "label"), and the
      // liberty lookup needs no stone test because libs[0] = 0 serves empty and border labels
      nv = *(lane < 4 ? &L->pt[iv + dl4] : &L->libs[0]);
      nl = L->libs[nv & 0x7FFFu];
      emp4 = (u32)bal_eq(nv, 0u) & 0xFu;
      if (!TRUSTED && emp4 == 0) {                                            // isSuicideMove :201-232
        const u32 pb = player == S_WHITE ? 0x8000u : 0u;
        const bool saves = lane < 4 && nv != PT_BORDER && (((nv & 0x8000u) == pb) ? nl > 1 : nl == 1);
        if (__ballot(saves) == 0) return 0;
      }
    }
    ELF_PHASE(*this, 2);   // TryPlay done
    // ---- _add_board_hash (go_state.cc:113-121): record the PRE-move position, skipped for pass
    const bool not_pass = TRUSTED || c != M_PASS;
    if (not_pass) {
      if (TRUSTED) {
        // k_playout: running per-lane record pointer (two adds per step instead of a 64-bit multiply-add)
        const u64 w = sk_record_word<N>(hash, Bw, Ww, lane);
        if (lane < G::SKW) *sk_wp = w;
        sk_wp += G::SKW;
      } else {
        sk.record(sk_len, hash, Bw, Ww, lane);
        sk_wp += G::SKW;   // stays where forward_legal_action expects it (k_playout mixes both: resignation never occurs there, but keep the invariant)
      }
      // Bloom insert: lane 0 sets the bit of the low hash word, lane 1 of the high word -- one ds_or_b32 for both (a
      // lane-dependent address keeps the compiler from wrapping a uniform atomic in its single-lane election sequence)
      const u32 hb = (lane == 0 ? (u32)hash : (u32)(hash >> 32)) & (G::BLOOM * 32 - 1);
      if (lane < 2) lds_or(&L->bloom[hb >> 5], 1u << (hb & 31));
    }
    ELF_PHASE(*this, 3);   // superko record + bloom insert
    int total_cap = 0, ko_c = 0, cap_a = 0;
    u32 bl1 = 0, bl2 = 0;   // Bloom words of the new position's hash, shifted to bit 0 (moves: probed early, see below)
    bool new_ko = false;
    if (is_move) {
      // ---- Play, board.cc:1297-1401
      const u32 ownbit = player == S_WHITE ? 0x8000u : 0u;
      // classify the <=4 distinct neighbour groups on lanes 0..3 (first occurrence wins, like GroupId4 slots)
      const u32 p1 = __builtin_amdgcn_update_dpp(0u, nv, 0x111, 0xf, 0xf, true), p2 = __builtin_amdgcn_update_dpp(0u, nv, 0x112, 0xf, 0xf, true),
                p3 = __builtin_amdgcn_update_dpp(0u, nv, 0x113, 0xf, 0xf, true);   // lane j <- lanes j-1, j-2, j-3 (0 before lane 0)
      const u64 firstm = bal_ne(nv & 0x7FFFu, 0u) & bal_ne(nv, p1) & bal_ne(nv, p2) & bal_ne(nv, p3);   // lanes >= 4 hold label 0
      const u64 ownm = bal_eq(nv & 0x8000u, ownbit);
      const u64 lib1 = bal_eq(nl, 1u);
      const u32 bo = (u32)(firstm & ownm);                        // own groups touched (merge candidates)
      const u64 enem = firstm & ~ownm;
      const u32 bc = (u32)(enem & lib1);                          // enemy groups captured by this move
      if (lane_bit(enem & ~lib1)) L->libs[nv & 0x7FFFu] = (u16)(nl - 1);   // surviving enemy groups lose the liberty at i (:1327)
      const int m = __popc(bo);
      const bool anycap = bc != 0;
      // label of the mover's group: the first own neighbour group's, else a new root at i (lane 31 holds label 0: the read is unconditional)
      const u32 nv_first = (u32)rl((int)nv, (int)__builtin_ctz(bo | 0x80000000u));
      const u32 newv = bo ? nv_first : (ownbit | (u32)i);
      bool at_changed = false;
      if (TRUSTED) at_changed = anycap || (enem & bal_eq(nl, 2u)) != 0;   // a capture, or an enemy neighbour group falls into atari
      ELF_PHASE(*this, 8);    // neighbour classification + enemy liberty decrement
      u64 capw = 0;   // lane-distributed bitboard of the stones captured by this move
      if (ELF_RARE(anycap)) {
        // EmptyGroup / RemoveStoneAndAddLiberty (:526-572): wave-parallel removal
        const u32 cv = lane_bit((u64)bc) ? nv : ~0u;   // captured labels on their lanes, ~0 (no label) elsewhere
        const u32 c0 = (u32)rl((int)cv, 0), c1 = (u32)rl((int)cv, 1), c2 = (u32)rl((int)cv, 2), c3 = (u32)rl((int)cv, 3);
        u64 xh = 0;
        u32 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = L->pt[idx[k]];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          // c* are stone labels or ~0: never matches an empty / clamped point
          const u64 bal = (bal_eq(v[k], c0) | bal_eq(v[k], c1) | bal_eq(v[k], c2) | bal_eq(v[k], c3)) & rl64(mValid, k);
          if (bal) {
            set_lane64(capw, k, bal);
            if (total_cap == 0) cap_a = k * 64 + (int)__builtin_ctzll(bal);   // :1355 capture_c: the stone, when exactly one is captured
            total_cap += __popcll(bal);
            if (lane_bit(bal)) { L->pt[idx[k]] = 0; xh ^= zob_col(zob_v[idx[k]], opp); }
          }
        }
        ko_c = a2c_u(cap_a);
        hash ^= wave_xor64(xh);
        Ww &= ~(capw & mblack);
        Bw &= ~(capw & ~mblack);
      }
      // the hash of the new position is final here: issue the Bloom probes of _check_superko now, consume them at the end
      hash ^= zob_col(sload_wait(zi), player);
      {
        const u32 h1 = (u32)hash & (G::BLOOM * 32 - 1), h2 = (u32)(hash >> 32) & (G::BLOOM * 32 - 1);
        bl1 = L->bloom[h1 >> 5] >> (h1 & 31);
        bl2 = L->bloom[h2 >> 5] >> (h2 & 31);
      }
      ELF_PHASE(*this, 9);    // capture removal
      // place the stone with its final label; fold further own groups into it (MergeGroups :712-752)
      if (lane == 0) L->pt[i] = (u16)newv;
      const u64 addw = lane == ka ? abit : 0ull;
      Bw |= addw & mblack;
      Ww |= addw & ~mblack;
      if (ELF_RARE(m >= 2)) {
        const u32 ov = lane_bit((u64)bo) ? nv : ~0u;
        const u32 o0 = (u32)rl((int)ov, 0), o1 = (u32)rl((int)ov, 1), o2 = (u32)rl((int)ov, 2), o3 = (u32)rl((int)ov, 3);
        u32 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = L->pt[idx[k]];
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const u64 mm = (bal_eq(v[k], o0) | bal_eq(v[k], o1) | bal_eq(v[k], o2) | bal_eq(v[k], o3)) & bal_ne(v[k], newv);
          if (lane_bit(mm)) L->pt[idx[k]] = (u16)newv;
        }
      }
      wsync();
      ELF_PHASE(*this, 10);   // placement + merge relabel
      if (ELF_RARE(anycap)) {
        // liberty give-back: every removed stone returns one liberty to each DISTINCT adjacent group (:533-538)
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const u64 ck = rl64(capw, k);
          if (ck) {
            if (lane_bit(ck)) {
              int p = idx[k];
              u32 a0 = L->pt[p - 1], a1 = L->pt[p + 1], a2 = L->pt[p - S], a3 = L->pt[p + S];
              if (is_stone(a0)) atomic_inc_u16(&L->libs[a0 & 0x7FFF]);
              if (is_stone(a1) && a1 != a0) atomic_inc_u16(&L->libs[a1 & 0x7FFF]);
              if (is_stone(a2) && a2 != a0 && a2 != a1) atomic_inc_u16(&L->libs[a2 & 0x7FFF]);
              if (is_stone(a3) && a3 != a0 && a3 != a1 && a3 != a2) atomic_inc_u16(&L->libs[a3 & 0x7FFF]);
            }
          }
        }
        wsync();
      }
      ELF_PHASE(*this, 4);   // liberty give-back
      // liberties of the mover's group
      int newlibs;
      const int root = newv & 0x7FFF;
      if (m == 0) {
        // createNewGroup (:661-671): its liberties are the empty neighbours after captures
        if (anycap) {
          const u32 e = *(lane < 4 ? &L->pt[i + dl4] : &L->pt[0]);   // pt[0] is a border point: not empty
          newlibs = __popcll(bal_eq(e, 0u));
        } else {
          newlibs = __popc(emp4);
        }
      } else if (m == 1) {
        // MergeToGroup (:677-708) restated: the played point stops being a liberty; each previously
        // empty neighbour e counts only if no other stone of the group already touches it.
        // lanes 0..11: neighbour kk3, its jj-th side not facing i; the read is unconditional, the emp4 test below drops
        // the triples of non-empty neighbours
        const u32 side = *reinterpret_cast<const u16*>(reinterpret_cast<const char*>(&L->pt[i]) + t12_off);
        const u32 t = (u32)bal_eq(side, newv) & 0xFFFu;
        const int add = ((emp4 & 1u) && ((t & 7u) == 0)) + ((emp4 & 2u) && (((t >> 3) & 7u) == 0)) +
                        ((emp4 & 4u) && (((t >> 6) & 7u) == 0)) + ((emp4 & 8u) && (((t >> 9) & 7u) == 0));
        const int old = (int)L->libs[root];
        newlibs = old - 1 + add;
        if (TRUSTED) at_changed |= (old == 1) != (newlibs == 1);
      } else {
        // RecomputeGroupLiberties (:754-782): |dilate(group) & empty| on bitboards
        u64 gw = 0;
        u32 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = L->pt[idx[k]];
#pragma unroll
        for (int k = 0; k < R; ++k) set_lane64(gw, k, bal_eq(v[k], newv));
        gw &= mValid;
        newlibs = popc_lanes(dilate(gw) & ~(Bw | Ww));
        if (TRUSTED) at_changed = true;
      }
      newlibs = rfl(newlibs);
      if (lane == 0) L->libs[root] = (u16)newlibs;
      new_ko = (m == 0 && total_cap == 1 && newlibs == 1);                    // :1386
      if (TRUSTED) {
        // the played stone joins the atari set when its group ends with one liberty (exact when nothing else changed)
        if (newlibs == 1) at_cache |= addw;
        at_dirty |= at_changed;
      }
    }
    ELF_PHASE(*this, 5);   // liberties of the mover's group
    // ---- history push (go_state.cc:90-92; BoardHistory(board) board_feature.h:45-56) as bitboards
    {
      const int slot = hist_cnt & (HIST - 1);
      if (lane < R) { L->hist[slot][0][lane] = Bw; L->hist[slot][1][lane] = Ww; }
    }
    // ---- header update: caps :1348-1351, ko :1384-1393, update_next_move :1225-1238
    if (is_move) {
      if (total_cap) { if (player == S_BLACK) b_cap += total_cap; else w_cap += total_cap; }
      if (new_ko) { ko_pt = ko_c; ko_a = cap_a; ko_color = opp; ko_age = 0; }
      else ko_age = (ko_age + 1) & 0xFFFF;
    }
    next_player = opp;
    lm3 = lm2; lm2 = lm1; lm1 = lm0; lm0 = c;
    ply++;
    hist_cnt = (hist_cnt + 1) & 0xFFFF;
    if (not_pass) sk_len++;
    superko = 0;
    wsync();
    // ---- _check_superko (go_state.cc:96-111) for the new position, cached in the header.
    // Bloom filter in LDS first (two probes); the exact (hash, image) records in HBM only on a hit.
    if (not_pass) {
      if (!is_move) {   // resign: the position (and its hash) is unchanged
        const u32 h1 = (u32)hash & (G::BLOOM * 32 - 1), h2 = (u32)(hash >> 32) & (G::BLOOM * 32 - 1);
        bl1 = L->bloom[h1 >> 5] >> (h1 & 31);
        bl2 = L->bloom[h2 >> 5] >> (h2 & 31);
      }
      if (ELF_RARE(rfl((int)(bl1 & bl2 & 1u)))) {
        if (sk.exact_hit(sk_len, hash, Bw, Ww, lane)) superko = 1;
      }
    }
    ELF_PHASE(*this, 6);   // history, header, superko check
    return 1;
  }

  // ---- legal moves for the side to move (TryPlay :788-827 over every point) --------------------
  // Results are lane-distributed bitboards (lane k holds actions [64k, 64k+64)).  Everything is settled on bitboards:
  //   At    = stones whose group has exactly one liberty (the only LDS work: 2 x R independent reads, pt then libs[root])
  //   legal = E & dilate(E | (Own & ~At) | (Opp & At))      an empty neighbour, or an own neighbour group that keeps a liberty,
  //                                                         or an enemy neighbour group in atari  (isSuicideMove :201-232)
  //   eye   = E & ~dilate(E | Opp) & legal & ~fake          isEye :1850-1860, isFakeEye :1887-1906 via the four diagonal shifts
  // minus the simple-ko point (:234-240).  cand = legal minus the mover's own true eyes, for the config-2 policy.
  // the atari set from the labels: libs[label & 0x7FFF] == 1 (empty points read libs[0] = 0)
  __device__ __forceinline__ u64 atari_set() const {
    u32 v[R], lb[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = L->pt[idx[k]];
#pragma unroll
    for (int k = 0; k < R; ++k) lb[k] = L->libs[v[k] & 0x7FFFu];   // on-board points hold 0 or a stone label (never the border mark)
    u64 At = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) set_lane64(At, k, bal_eq(lb[k], 1u));
    return At & (Bw | Ww);   // drops the clamped lanes of the last round (they re-read point 0)
  }
  // INCR (k_playout): reuse the atari set carried in at_cache unless the last move invalidated it
  template <bool WITH_EYES, bool INCR = false>
  __device__ __forceinline__ void legal_moves(u64& legal, u64& cand) {
    const int player = next_player;
    const u64 mblack = player == S_BLACK ? ~0ull : 0ull;
    const u64 Own = (Bw & mblack) | (Ww & ~mblack), Opp = (Ww & mblack) | (Bw & ~mblack);
    const u64 E = ~(Bw | Ww) & mValid;
    u64 At;
    if (INCR) {
      if (at_dirty) { at_cache = atari_set(); at_dirty = 0; }
      At = at_cache;
    } else {
      At = atari_set();
    }
    u64 okw;
    u64 eyew = 0;
    if (!WITH_EYES) {
      okw = E & dilate(E | (Own & ~At) | (Opp & At));
    } else {
      u64 d1w, d2w;
      dilate2(E | (Own & ~At) | (Opp & At), E | Opp, d1w, d2w);
      okw = E & d1w;
      const u64 allown = okw & ~d2w;
      if (bal_ne64(allown, 0ull)) {
        // the four diagonal neighbours in Opp: (a-1 | a+1) first, the pair packed into one register, then -N / +N on both at once
        const u64 pk = sh_m1(Opp) | dpp_u64<0x118>(sh_p1(Opp));
        const u64 d1 = sh_mN(pk), d3 = sh_pN(pk);                   // lanes 0..R-1: from a-1; lanes 8..: from a+1
        const u64 d2 = dpp_u64<0x108>(d1), d4 = dpp_u64<0x108>(d3);
        const u64 ge1 = d1 | d2 | d3 | d4;
        const u64 ge2 = (d1 & d2) | (d3 & d4) | ((d1 | d2) & (d3 | d4));
        const u64 fake = (mEdge & ge1) | (~mEdge & ge2);
        eyew = allown & ~fake;
      }
    }
    if (ELF_RARE(ko_age == 0 && ko_color == player && ko_pt != 0)) {
      if (lane == (ko_a >> 6)) okw &= ~(1ull << (ko_a & 63));
    }
    legal = okw;
    cand = okw & ~eyew;
  }

  // ---- Tromp-Taylor area score (go_state.h:32-93 simple_flood_fill x2 + simple_tt_scoring) ----
  // Row-bitboard flood fill in registers: lane x holds column-bits y of row x.
  __device__ int tt_area() const {
    // re-slice the action-order bitboards into rows: row x = bits [x*N, x*N+N)
    const int bit0 = (lane < N ? lane : 0) * N, w = bit0 >> 6, s = bit0 & 63;
    const u64 b0 = rl64(Bw, 0), b1 = R > 1 ? rl64(Bw, 1 % R) : 0, b2 = R > 2 ? rl64(Bw, 2 % R) : 0, b3 = R > 3 ? rl64(Bw, 3 % R) : 0,
              b4 = R > 4 ? rl64(Bw, 4 % R) : 0, b5 = R > 5 ? rl64(Bw, 5 % R) : 0;
    const u64 w0 = rl64(Ww, 0), w1 = R > 1 ? rl64(Ww, 1 % R) : 0, w2 = R > 2 ? rl64(Ww, 2 % R) : 0, w3 = R > 3 ? rl64(Ww, 3 % R) : 0,
              w4 = R > 4 ? rl64(Ww, 4 % R) : 0, w5 = R > 5 ? rl64(Ww, 5 % R) : 0;
    auto sel = [&](int q, u64 x0, u64 x1, u64 x2, u64 x3, u64 x4, u64 x5) -> u64 {
      return q == 0 ? x0 : q == 1 ? x1 : q == 2 ? x2 : q == 3 ? x3 : q == 4 ? x4 : q == 5 ? x5 : 0ull;
    };
    const u64 bl = sel(w, b0, b1, b2, b3, b4, b5), bh = sel(w + 1, b0, b1, b2, b3, b4, b5);
    const u64 wl = sel(w, w0, w1, w2, w3, w4, w5), wh = sel(w + 1, w0, w1, w2, w3, w4, w5);
    const u32 rowmask = (1u << N) - 1;
    u32 B = 0, Wt = 0;
    if (lane < N) {
      u64 bb = s ? ((bl >> s) | (bh << (64 - s))) : bl;
      u64 ww = s ? ((wl >> s) | (wh << (64 - s))) : wl;
      B = (u32)bb & rowmask;
      Wt = (u32)ww & rowmask;
    }
    const u32 E = (lane < N) ? (~(B | Wt) & rowmask) : 0u;
    u32 rb = B, rw = Wt;
    for (;;) {
      u32 ub = __shfl_up(rb, 1, 64), db = __shfl_down(rb, 1, 64);
      u32 uw = __shfl_up(rw, 1, 64), dw = __shfl_down(rw, 1, 64);
      if (lane == 0) { ub = 0; uw = 0; }
      if (lane >= N - 1) { db = 0; dw = 0; }
      u32 nb = rb | (E & ((rb << 1) | (rb >> 1) | ub | db));
      u32 nw = rw | (E & ((rw << 1) | (rw >> 1) | uw | dw));
      // finish the in-row run before the next vertical exchange
      for (int q = 0; q < 5; ++q) {
        nb |= E & ((nb << 1) | (nb >> 1));
        nw |= E & ((nw << 1) | (nw >> 1));
      }
      bool ch = (nb != rb) || (nw != rw);
      rb = nb; rw = nw;
      if (!__any(ch)) break;
    }
    int bv = __popc(rb & ~rw), wv = __popc(rw & ~rb);
    int d = (lane < N) ? (bv - wv) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    return d;
  }

  // GoState::evaluate (go_state.h:194-203)
  __device__ __forceinline__ float evaluate(float komi) const {
    if (superko) return next_player == S_BLACK ? 1.0f : -1.0f;
    return (float)tt_area() - komi;
  }
};

// BoardFeature::extractAGZ (board_feature.cc:247-290) under Transform (board_feature.h:97-113), two stages:
//  1. agz_bitplanes: the 18 planes as bit rows in OUTPUT order (bit o of plane p = value at transformed point o).  Lane l
//     looks up the source bit of output points 64k+l in the history ring; one ballot per (plane, k) is the output word.
//  2. agz_store: the row is one flat array (fp32 [18][N][N], the reference's layout, or fp16 [N][N][18] = channels_last
//     for an fp16 net); lanes own 16 consecutive BYTES of it, so every store instruction of the body is one
//     contiguous 1-KiB segment (global_store_dwordx4) whatever the alignment of the row (head/tail peeled per element).
enum { FEAT_F32_NCHW = 0, FEAT_F16_NHWC = 1 };
// LDS scratch one wave needs while it extracts a row: 64 points x 18 halves (fp16 path; the fp32 bit string and the staged
// bit planes are smaller)
constexpr int AGZ_SCRATCH_BYTES = 64 * 36;
// One wave extracts one row with its own LDS scratch; LDS operations of a wave execute in order, so the only synchronisation
// between the phases is a compiler fence (several waves of a workgroup may be in different phases of different rows).
__device__ __forceinline__ void agz_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

template <int N>
__device__ __forceinline__ void agz_bitplanes(const u64 (*hist)[2][Geo<N>::R], int cnt, int player, int d4,
                                              u64 (*tpl)[Geo<N>::R], int lane) {
  using G = Geo<N>;
  const int len = cnt < HIST ? cnt : HIST;
  const int rot = d4 & 3;
  const bool flip = ((d4 >> 2) & 1) != 0;
  const bool blk = player == S_BLACK;
#pragma unroll
  for (int k = 0; k < G::R; ++k) {
    int o = k * 64 + lane;
    const bool valid = o < G::NP;
    if (!valid) o = 0;
    int xo = o / N, yo = o % N;
    if (flip) { int t = xo; xo = yo; yo = t; }
    int x = xo, y = yo;
    if (rot == 1) { x = N - 1 - yo; y = xo; }
    else if (rot == 2) { x = N - 1 - xo; y = N - 1 - yo; }
    else if (rot == 3) { x = yo; y = N - 1 - xo; }
    const int a = x * N + y, w = a >> 6, sft = a & 63;
#pragma unroll
    for (int hk = 0; hk < HIST; ++hk) {
      const int slot = (cnt - 1 - hk) & (HIST - 1);
      const bool on = valid && hk < len;
      const u64 mb = __ballot(on && ((hist[slot][0][w] >> sft) & 1));
      const u64 mw = __ballot(on && ((hist[slot][1][w] >> sft) & 1));
      if (lane == 0) { tpl[2 * hk][k] = blk ? mb : mw; tpl[2 * hk + 1][k] = blk ? mw : mb; }
    }
    const u64 full = __ballot(valid);
    if (lane == 0) { tpl[16][k] = blk ? full : 0; tpl[17][k] = blk ? 0 : full; }
  }
}

template <int N>
__device__ __forceinline__ u32 agz_bit(const u64 (*tpl)[Geo<N>::R], int p, int o) { return (u32)(tpl[p][o >> 6] >> (o & 63)) & 1u; }

template <int N, int FMT>
__device__ __forceinline__ void agz_store(const u64 (*tpl)[Geo<N>::R], void* __restrict__ row, int lane) {
  using G = Geo<N>;
  constexpr int TOTAL = 18 * G::NP;
  const uintptr_t addr = (uintptr_t)row;
  if (FMT == FEAT_F32_NCHW) {
    float* out = (float*)row;
    const int head = (int)(((16 - (addr & 15)) & 15) >> 2);     // floats up to the first 16-B boundary
    if (lane < head) out[lane] = agz_bit<N>(tpl, lane / G::NP, lane % G::NP) ? 1.0f : 0.0f;
    const int body = (TOTAL - head) >> 2;
    float4* o4 = (float4*)(out + head);
    for (int j = lane; j < body; j += 64) {
      const int f = head + 4 * j;
      int p = f / G::NP, o = f - p * G::NP;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = agz_bit<N>(tpl, p, o) ? 1.0f : 0.0f;
        if (++o == G::NP) { o = 0; ++p; }
      }
      o4[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
    const int f = head + 4 * body + lane;
    if (f < TOTAL) out[f] = agz_bit<N>(tpl, f / G::NP, f % G::NP) ? 1.0f : 0.0f;
  } else {
    u16* out = (u16*)row;                                       // IEEE half bits: 1.0 = 0x3C00
    const int head = (int)(((16 - (addr & 15)) & 15) >> 1);
    if (lane < head) out[lane] = agz_bit<N>(tpl, lane % 18, lane / 18) ? 0x3C00 : 0;
    const int body = (TOTAL - head) >> 3;
    uint4* o4 = (uint4*)(out + head);
    for (int j = lane; j < body; j += 64) {
      const int f = head + 8 * j;
      int o = f / 18, p = f - o * 18;
      u32 v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        u32 lo = agz_bit<N>(tpl, p, o) ? 0x3C00u : 0u;
        if (++p == 18) { p = 0; ++o; }
        u32 hi = agz_bit<N>(tpl, p, o) ? 0x3C00u : 0u;
        if (++p == 18) { p = 0; ++o; }
        v[e] = lo | (hi << 16);
      }
      o4[j] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    const int f = head + 8 * body + lane;
    if (f < TOTAL) out[f] = agz_bit<N>(tpl, f % 18, f / 18) ? 0x3C00 : 0;
  }
}

// Fast paths (rows aligned to 4 bytes, i.e. every fp32 row and every fp16 row a tensor allocator hands out).  Per round of 64
// output points the wave gathers, for each of the 16 history planes, the dword of the source bitboard that holds the
// transformed point's bit (LDS, all 16 reads in flight together) and turns "bit set" into a wave mask with one v_and + v_cmp.
//   fp16 NHWC (agz_nhwc_f16): a lane builds its point's 18 halves = 36 contiguous bytes straight from the masks (v_cndmask on
//     the SGPR pairs), the 64 points of the round are transposed through LDS and leave as 1-KiB contiguous stores.
//   fp32 NCHW (agz_flat_f32): the masks are parked lane-wise (v_writelane) and then ORed, 64 words at a time, into ONE contiguous
//     bit string in LDS whose bit f is element f of the flat [18][N][N] row; the store pass then is 16 bytes per lane, 1 KiB per
//     instruction, 16-byte aligned: one ds_read2_b32 + v_alignbit gives a lane its 4 bits (a step of 64 lanes is exactly 8 dwords
//     of the bit string, so the LDS address is an immediate and the shift a per-lane constant).
//     (256-byte dword stores straight from the masks were measured at 0.7x of the old staged path: the memory pipe wants 16 B / lane
//     and long contiguous runs per instruction.)
// History entries that were never pushed read as empty boards: reset() zero-fills the ring and nothing else rewinds hist_cnt,
// so "fewer than 8 positions so far" needs no test here (the staged path above keeps its explicit `hk < len`).
struct __attribute__((packed, aligned(4))) AgzQuad { u32 x, y, z, w; };
template <int N, bool BLK>
__device__ __forceinline__ void agz_nhwc_f16(const u64 (*hist)[2][Geo<N>::R], u32* scratch, int cnt, int d4, void* __restrict__ row, int lane) {
  using G = Geo<N>;
  constexpr int R = G::R, NP = G::NP;
  const int rot = d4 & 3;
  const bool flip = ((d4 >> 2) & 1) != 0;
  // dwords of the ring: [slot][colour][2 R]; the mover's colour first (planes 2 hk), then the opponent's (planes 2 hk + 1)
  const u32* h_own = reinterpret_cast<const u32*>(hist) + (BLK ? 0 : 2 * R);
  const u32* h_opp = reinterpret_cast<const u32*>(hist) + (BLK ? 2 * R : 0);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const u32 o = (u32)(k * 64 + lane);
    const bool valid = (k + 1) * 64 <= NP || o < (u32)NP;
    // Transform (board_feature.h:97-113): output point o = (xo, yo) shows source point (x, y)
    u32 xo = __umul24(o, (u32)G::DN_M) >> G::DN_S, yo = o - __umul24(xo, (u32)N);
    if (flip) { const u32 t = xo; xo = yo; yo = t; }
    u32 x = xo, y = yo;
    if (rot == 1) { x = N - 1 - yo; y = xo; }
    else if (rot == 2) { x = N - 1 - xo; y = N - 1 - yo; }
    else if (rot == 3) { x = yo; y = N - 1 - xo; }
    const u32 a = valid ? __umul24(x, (u32)N) + y : 0u;
    const u32 dw = a >> 5, bm = valid ? (1u << (a & 31)) : 0u;
    u32 wo[HIST], wp[HIST];
#pragma unroll
    for (int hk = 0; hk < HIST; ++hk) {   // all 16 gathers in flight before the first is consumed
      const int slot = (cnt - 1 - hk) & (HIST - 1);
      wo[hk] = h_own[slot * (4 * R) + dw];
      wp[hk] = h_opp[slot * (4 * R) + dw];
    }
    u64 m[18];
#pragma unroll
    for (int hk = 0; hk < HIST; ++hk) {
      m[2 * hk] = bal_ne(wo[hk] & bm, 0u);
      m[2 * hk + 1] = bal_ne(wp[hk] & bm, 0u);
    }
    const u64 full = bal_ne(bm, 0u);
    m[16] = BLK ? full : 0ull;
    m[17] = BLK ? 0ull : full;
    if (valid) {
      u32 d[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) d[j] = (lane_bit(m[2 * j]) ? 0x3C00u : 0u) | (lane_bit(m[2 * j + 1]) ? 0x3C000000u : 0u);
      u32* t = scratch + lane * 9;   // 9-dword stride: conflict-free
#pragma unroll
      for (int j = 0; j < 9; ++j) t[j] = d[j];
    }
    {
      // the round's 64 x 36 bytes leave as contiguous 16-byte pieces, 1 KiB per store instruction (a lane storing its own 36
      // bytes -- 16, 16, 4 at a 36-byte lane stride -- ran at 0.65x of this)
      agz_wave_sync();
      const int npts = (k + 1) * 64 <= NP ? 64 : NP - k * 64;
      const int nq = npts * 9 / 4;                          // whole 16-byte pieces of this round
      char* outk = (char*)row + (size_t)k * 64 * 36;
      const uint4* s4 = reinterpret_cast<const uint4*>(scratch);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int q = t * 64 + lane;
        if (q < nq) { const uint4 v = s4[q]; *reinterpret_cast<AgzQuad*>(outk + 16 * q) = AgzQuad{v.x, v.y, v.z, v.w}; }
      }
      const int rem = npts * 9 - nq * 4;                    // dwords after the last whole piece (0..3)
      if (lane < rem) reinterpret_cast<u32*>(outk)[nq * 4 + lane] = scratch[nq * 4 + lane];
      agz_wave_sync();
    }
  }
}

template <int N, bool BLK>
__device__ __forceinline__ void agz_flat_f32(const u64 (*hist)[2][Geo<N>::R], u32* bs /* LDS, >= 18*NP/32 + 3 dwords */, int cnt, int d4,
                                             float* __restrict__ out, int lane) {
  using G = Geo<N>;
  constexpr int R = G::R, NP = G::NP, TOTAL = 18 * NP, NDW = (TOTAL + 31) / 32 + 2, NW = 18 * R, NB = (NW + 63) / 64;
  const int rot = d4 & 3;
  const bool flip = ((d4 >> 2) & 1) != 0;
  const u32* h_own = reinterpret_cast<const u32*>(hist) + (BLK ? 0 : 2 * R);
  const u32* h_opp = reinterpret_cast<const u32*>(hist) + (BLK ? 2 * R : 0);
  for (int j = lane; j < NDW; j += 64) bs[j] = 0;
  u64 acc[NB];   // word w = p R + k of the plane-major mask array lives in lane w & 63 of acc[w >> 6]
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = 0;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const u32 o = (u32)(k * 64 + lane);
    const bool valid = (k + 1) * 64 <= NP || o < (u32)NP;
    u32 xo = __umul24(o, (u32)G::DN_M) >> G::DN_S, yo = o - __umul24(xo, (u32)N);
    if (flip) { const u32 t = xo; xo = yo; yo = t; }
    u32 x = xo, y = yo;
    if (rot == 1) { x = N - 1 - yo; y = xo; }
    else if (rot == 2) { x = N - 1 - xo; y = N - 1 - yo; }
    else if (rot == 3) { x = yo; y = N - 1 - xo; }
    const u32 a = valid ? __umul24(x, (u32)N) + y : 0u;
    const u32 dw = a >> 5, bm = valid ? (1u << (a & 31)) : 0u;
    u32 wo[HIST], wp[HIST];
#pragma unroll
    for (int hk = 0; hk < HIST; ++hk) {
      const int slot = (cnt - 1 - hk) & (HIST - 1);
      wo[hk] = h_own[slot * (4 * R) + dw];
      wp[hk] = h_opp[slot * (4 * R) + dw];
    }
#pragma unroll
    for (int hk = 0; hk < HIST; ++hk) {
      const int w0 = (2 * hk) * R + k, w1 = (2 * hk + 1) * R + k;
      set_lane64(acc[w0 >> 6], w0 & 63, bal_ne(wo[hk] & bm, 0u));
      set_lane64(acc[w1 >> 6], w1 & 63, bal_ne(wp[hk] & bm, 0u));
    }
    const int wc = (BLK ? 16 : 17) * R + k;   // the "mover's colour" plane is all ones (plane 16 if Black is to move, else 17)
    set_lane64(acc[wc >> 6], wc & 63, bal_ne(bm, 0u));
  }
  agz_wave_sync();   // the zero fill is ordered before the ORs: LDS operations of one wave execute in order
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const u32 w = (u32)(b * 64 + lane);
    if (w < (u32)NW) {
      const u32 p = R == 6 ? (__umul24(w, 43u) >> 8) : (w >> 1);   // w / R for R = 6 (w < 128) or 2
      const u32 k = w - p * R;
      const u32 B = p * NP + 64 * k, sft = B & 31;
      const u64 v = acc[b], t = v << sft;
      const u32 top = sft ? (u32)(v >> (64 - sft)) : 0u;
      u32* d = bs + (B >> 5);
      Board<N>::lds_or(d, (u32)t);
      Board<N>::lds_or(d + 1, (u32)(t >> 32));
      Board<N>::lds_or(d + 2, top);
    }
  }
  agz_wave_sync();
  // flat store pass: element f of the row is bit f of bs
  const uintptr_t addr = (uintptr_t)out;
  const int head = (int)(((16 - (addr & 15)) & 15) >> 2);     // floats up to the first 16-B boundary (aligning the 1-KiB store
                                                              // instructions to 128 B instead, and nt stores, were measured: no gain)
  if (lane < head) out[lane] = (float)((bs[0] >> lane) & 1u);
  const int body = (TOTAL - head) >> 2;
  float4* o4 = (float4*)(out + head);
  const u32 f0 = (u32)(head + 4 * lane), sft = f0 & 31;
  const u32* b0 = bs + (f0 >> 5);
  constexpr int ITER = (TOTAL / 4 + 63) / 64;
#pragma unroll
  for (int t = 0; t < ITER; ++t) {
    const int j = lane + 64 * t;
    const u32 win = __builtin_amdgcn_alignbit(b0[8 * t + 1], b0[8 * t], sft);
    if (j < body)
      o4[j] = make_float4((float)(win & 1u), (float)((win >> 1) & 1u), (float)((win >> 2) & 1u), (float)((win >> 3) & 1u));
  }
  const int f = head + 4 * body + lane;
  if (f < TOTAL) out[f] = (float)((bs[f >> 5] >> (f & 31)) & 1u);
}

// whole extraction of one position by one wave; `hist` and `scratch` are LDS
template <int N>
__device__ __forceinline__ void extract_agz_row(const u64 (*hist)[2][Geo<N>::R], u64* scratch /* LDS, AGZ_SCRATCH_BYTES */, int cnt, int player, int d4,
                                                void* __restrict__ row, int fmt, int lane) {
  u64 (*tpl)[Geo<N>::R] = reinterpret_cast<u64 (*)[Geo<N>::R]>(scratch);
  if (rfl((int)(((uintptr_t)row & 3) == 0))) {
    const bool blk = player == S_BLACK;
    if (fmt == FEAT_F16_NHWC) {
      u32* sc = reinterpret_cast<u32*>(scratch);
      if (blk) agz_nhwc_f16<N, true>(hist, sc, cnt, d4, row, lane); else agz_nhwc_f16<N, false>(hist, sc, cnt, d4, row, lane);
    } else {
      u32* bs = reinterpret_cast<u32*>(scratch);
      if (blk) agz_flat_f32<N, true>(hist, bs, cnt, d4, (float*)row, lane); else agz_flat_f32<N, false>(hist, bs, cnt, d4, (float*)row, lane);
    }
    return;
  }
  // rows at odd 2-byte offsets (fp16 only): staged path
  agz_bitplanes<N>(hist, cnt, player, d4, tpl, lane);
  agz_wave_sync();
  if (fmt == FEAT_F16_NHWC) agz_store<N, FEAT_F16_NHWC>(tpl, row, lane);
  else agz_store<N, FEAT_F32_NCHW>(tpl, row, lane);
}

// BoardFeature::action2Coord (board_feature.h:139-144): NN action id under D4 code d4 -> (reference Coord, D4-0 action id)
template <int N>
__device__ __forceinline__ void action_to_coord(int i, int d4, int& coord, int& a0) {
  constexpr int S = N + 2;
  if (i >= N * N) { coord = M_PASS; a0 = N * N; return; }
  int xo = (int)(((u32)i * (u32)Geo<N>::DN_M) >> Geo<N>::DN_S), yo = i - xo * N;   // i / N, i % N for 0 <= i < N*N (Geo::DN_M)
  if ((d4 >> 2) & 1) { int t = xo; xo = yo; yo = t; }
  const int rot = d4 & 3;
  int x = xo, y = yo;
  if (rot == 1) { x = N - 1 - yo; y = xo; }
  else if (rot == 2) { x = N - 1 - xo; y = N - 1 - yo; }
  else if (rot == 3) { x = yo; y = N - 1 - xo; }
  coord = (y + 1) * S + (x + 1);
  a0 = x * N + y;
}

}  // namespace elfgo
