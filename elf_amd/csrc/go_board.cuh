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
//  * wave64, one wave = one workgroup = one board: no cross-wave sync, __syncthreads() degenerates to a
//    waitcnt, every control decision is wave-uniform (readlane -> SGPR) so branches are scalar.
//  * points live in LDS on a padded (N+2)^2 grid in x-major order so that NN action ids
//    (a = x*N + y, board.h:189) map to consecutive LDS addresses: idx = (x+1)*(N+2) + (y+1).
//    The reference Coord is the transpose, c = (y+1)*(N+2) + (x+1) (board.h:183); i<->c is an involution.
//  * pt[idx] (u16): 0 empty, 0xFFFF border, else (white?0x8000:0) | root, root = idx of the group's
//    representative point.  libs[root] (u16) = liberties of that group.  Lane l owns points
//    a = 64*k + l, k < R (R = 6 for 19x19, 2 for 9x9).
//  * captures / merges / liberty recounts are wave-parallel scans over the R rounds with __ballot +
//    __popcll; liberty give-back after a capture uses LDS atomics.
//  * history = ring of 8 x {black,white} bitboards in action order (W u64 words each); superko keeps
//    16-bit tags of every pre-move hash in LDS and the full (hash, bitboards) records in HBM, read only
//    on a tag hit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace elfgo {

typedef unsigned short u16;
typedef unsigned int u32;
typedef unsigned long long u64;

enum { S_EMPTY = 0, S_BLACK = 1, S_WHITE = 2 };                       // base/common.h:36-38
enum { M_PASS = 0, M_RESIGN = 1, M_SKIP = 2, M_INVALID = 3, M_CLEAR = 4 };  // base/common.h:43-47
constexpr u16 PT_BORDER = 0xFFFF;
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
  static constexpr int TAGS = (MAXMOVE + 2 + 7) & ~7;
  static constexpr int SKW = 2 * R;               // u64 words per superko image (black, white)
};

// One board slot = the LDS image, also the HBM image (copied 16 B per lane).
template <int N>
struct alignas(16) Slot {
  using G = Geo<N>;
  Hdr h;
  u16 pt[G::PP];
  u16 libs[G::PP];
  u64 hist[HIST][2][G::R];
  u16 tags[G::TAGS];
  static constexpr int RAW = 64 + 2 * G::PP * 2 + HIST * 2 * G::R * 8 + G::TAGS * 2;
  unsigned char pad[((RAW + 255) & ~255) - RAW];
};
static_assert(sizeof(Slot<19>) == 4096, "19x19 slot is 4 KiB");
static_assert(sizeof(Slot<9>) % 256 == 0, "slot is 256-B granular");

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ u64 rfl64(u64 v) {
  u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_xor64(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ bool is_stone(u32 v) { return v != 0 && v != PT_BORDER; }
// base/board.cc:24-36 transform_hash
__device__ __forceinline__ u64 zob_col(u64 h, int s) { return s == S_BLACK ? h : ((h >> 32) | (h << 32)); }
__device__ __forceinline__ u16 sk_tag(u64 h) { return (u16)(h ^ (h >> 16) ^ (h >> 32) ^ (h >> 48)); }

// config-2 counter RNG, the CPU checkers under oracle/ restate the same function
__device__ __forceinline__ u32 playout_rng(u64 seed, u32 t) {
  u64 z = seed + (u64)(t + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (u32)(z >> 32);
}

template <int N>
struct Board {
  using G = Geo<N>;
  static constexpr int S = G::S, NP = G::NP, R = G::R;

  Slot<N>* L;          // LDS image of this wave's board
  const u64* zob;      // Zobrist constants in INTERNAL index order (global memory)
  u64* sk_hash;        // this board's superko hashes   [MAXMOVE+2]        (HBM)
  u64* sk_img;         // this board's superko images   [MAXMOVE+2][SKW]   (HBM)
  int lane;
  int idx[R];          // LDS index of this lane's point in round k (clamped for invalid lanes)
  bool valid[R];

  __device__ __forceinline__ static int a2i(int a) { return a + S + 1 + 2 * (a / N); }
  __device__ __forceinline__ static int tr(int i) { return (i % S) * S + i / S; }  // idx <-> reference Coord

  __device__ __forceinline__ void init(Slot<N>* lds, const u64* z, u64* skh, u64* ski) {
    L = lds; zob = z; sk_hash = skh; sk_img = ski;
    lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      int a = k * 64 + lane;
      valid[k] = a < NP;
      idx[k] = a2i(valid[k] ? a : 0);
    }
  }

  // ---- slot <-> HBM, 16 B per lane, fully coalesced ------------------------------------------
  __device__ __forceinline__ void load(const Slot<N>* g) {
    const uint4* s = reinterpret_cast<const uint4*>(g);
    uint4* d = reinterpret_cast<uint4*>(L);
#pragma unroll
    for (int j = lane; j < (int)(sizeof(Slot<N>) / 16); j += 64) d[j] = s[j];
    __syncthreads();
  }
  __device__ __forceinline__ void store(Slot<N>* g) const {
    __syncthreads();
    const uint4* s = reinterpret_cast<const uint4*>(L);
    uint4* d = reinterpret_cast<uint4*>(g);
#pragma unroll
    for (int j = lane; j < (int)(sizeof(Slot<N>) / 16); j += 64) d[j] = s[j];
  }

  // base/board.cc:79-107 clearBoard + base/go_state.cc:134-141 reset, straight into LDS
  __device__ __forceinline__ void reset() {
    u32* w = reinterpret_cast<u32*>(L);
    for (int j = lane; j < (int)(sizeof(Slot<N>) / 4); j += 64) w[j] = 0;
    __syncthreads();
    for (int j = lane; j < G::PP; j += 64) {
      int a = j / S, b = j % S;
      bool on = j < G::P && a >= 1 && a <= N && b >= 1 && b <= N;
      L->pt[j] = on ? 0 : PT_BORDER;
    }
    if (lane == 0) {
      L->h.ply = 1;
      L->h.next_player = S_BLACK;
      for (int j = 0; j < 4; ++j) L->h.last_move[j] = M_INVALID;
    }
    __syncthreads();
  }

  // current position as bitboards = newest history entry (zeros for a fresh board)
  __device__ __forceinline__ const u64* cur_bits() const {
    int cnt = L->h.hist_cnt;
    return &L->hist[(cnt + HIST - 1) & (HIST - 1)][0][0];
  }

  // base/go_state.h:141-147 terminated()
  __device__ __forceinline__ bool terminated() const {
    const Hdr& h = L->h;
    return (h.last_move[0] == M_PASS && h.last_move[1] == M_PASS) || h.ply >= G::MAXMOVE || h.superko;
  }

  // ---- GoState::forward (go_state.cc:74-94). c = reference Coord, wave-uniform. ---------------
  // returns 1 played, 0 refused (terminated / illegal). M_INVALID is rejected by the host (-1).
  __device__ int forward(int c) {
    c = rfl(c);
    Hdr& h = L->h;
    if (terminated()) return 0;
    const int player = h.next_player, opp = S_BLACK + S_WHITE - player;
    const bool is_move = !(c == M_PASS || c == M_RESIGN);
    int i = 0;
    u32 nv = 0, nl = 0;
    int n[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
    // reference delta4 order L,T,R,B = x-1,y-1,x+1,y+1 (board.h:220) -> internal -S,-1,+S,+1
    const int dl = (lane & 1) ? ((lane & 2) ? 1 : -1) : ((lane & 2) ? S : -S);
    if (is_move) {
      // ---- TryPlay, board.cc:788-827
      if (c >= G::P) return 0;
      int x = c % S - 1, y = c / S - 1;
      if (x < 0 || x >= N || y < 0 || y >= N) return 0;                       // :803
      i = (x + 1) * S + (y + 1);
      if (L->pt[i] != 0) return 0;                                            // :808
      if (h.ko_pt == c && h.ko_age == 0 && h.ko_color == player) return 0;    // :234-240
      if (lane < 4) {                                                         // StoneLibertyAnalysis :161-199
        nv = L->pt[i + dl];
        nl = is_stone(nv) ? L->libs[nv & 0x7FFF] : 0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { n[k] = rl((int)nv, k); l[k] = rl((int)nl, k); }
      int nempty = 0, own_safe = 0, enemy_atari = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (n[k] == 0) { ++nempty; continue; }
        if (n[k] == PT_BORDER) continue;
        bool own = ((n[k] >> 15) + 1) == player;
        if (own) own_safe += l[k] > 1; else enemy_atari += l[k] == 1;
      }
      if (nempty == 0 && own_safe == 0 && enemy_atari == 0) return 0;         // isSuicideMove :201-232
    }
    // ---- _add_board_hash (go_state.cc:113-121): record the PRE-move position, skipped for pass
    if (c != M_PASS) {
      int t = h.sk_len;
      const u64* cb = cur_bits();
      bool have = h.hist_cnt != 0;
      if (lane < G::SKW) sk_img[(size_t)t * G::SKW + lane] = have ? cb[lane] : 0ull;
      if (lane == 0) { sk_hash[t] = h.hash; L->tags[t] = sk_tag(h.hash); }
    }
    __syncthreads();
    u64 hash = h.hash;
    int total_cap = 0, ko_c = 0;
    bool new_ko = false;
    if (is_move) {
      // ---- Play, board.cc:1297-1401
      const u32 ownbit = player == S_WHITE ? 0x8000u : 0u;
      int own[4], m = 0, cap[4], nc = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        own[k] = -1; cap[k] = -1;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (n[k] == 0 || n[k] == PT_BORDER) continue;
        bool dup = false;
#pragma unroll
        for (int j = 0; j < k; ++j) dup |= n[j] == n[k];
        if (dup) continue;
        if ((u32)(n[k] & 0x8000) == ownbit) {
          own[m++] = n[k];
        } else if (l[k] == 1) {
          cap[nc++] = n[k];                                                   // :1346 liberties hit 0
        } else if (lane == 0) {
          L->libs[n[k] & 0x7FFF] = (u16)(l[k] - 1);                           // :1327 --g->liberties
        }
      }
      // final representative of the mover's group
      const u32 newv = m > 0 ? (u32)own[0] : (ownbit | (u32)i);
      bool capf[R];
#pragma unroll
      for (int k = 0; k < R; ++k) capf[k] = false;
      if (nc > 0) {
        // EmptyGroup / RemoveStoneAndAddLiberty (:526-572): wave-parallel removal
        u64 xh = 0;
        u64 capbal = 0; int capk = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          u32 v = valid[k] ? L->pt[idx[k]] : 0;
          bool isc = v != 0 && ((int)v == cap[0] || (int)v == cap[1] || (int)v == cap[2] || (int)v == cap[3]);
          u64 bal = __ballot(isc);
          if (bal) { capbal = bal; capk = k; }
          total_cap += __popcll(bal);
          capf[k] = isc;
          if (isc) { L->pt[idx[k]] = 0; xh ^= zob_col(zob[idx[k]], opp); }
        }
        hash ^= wave_xor64(xh);
        if (total_cap == 1) ko_c = tr(a2i(capk * 64 + (int)__builtin_ctzll(capbal)));   // :1355 capture_c
      }
      // place the stone with its final label; fold further own groups into it (MergeGroups :712-752)
      if (lane == 0) L->pt[i] = (u16)newv;
      if (m >= 2) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
          u32 v = valid[k] ? L->pt[idx[k]] : 0;
          if (v != 0 && ((int)v == own[1] || (int)v == own[2] || (int)v == own[3])) L->pt[idx[k]] = (u16)newv;
        }
      }
      __syncthreads();
      if (nc > 0) {
        // liberty give-back: every removed stone returns one liberty to each DISTINCT adjacent group (:533-538)
#pragma unroll
        for (int k = 0; k < R; ++k) {
          if (capf[k]) {
            int p = idx[k];
            u32 a0 = L->pt[p - 1], a1 = L->pt[p + 1], a2 = L->pt[p - S], a3 = L->pt[p + S];
            if (is_stone(a0)) atomicAdd_u16(&L->libs[a0 & 0x7FFF]);
            if (is_stone(a1) && a1 != a0) atomicAdd_u16(&L->libs[a1 & 0x7FFF]);
            if (is_stone(a2) && a2 != a0 && a2 != a1) atomicAdd_u16(&L->libs[a2 & 0x7FFF]);
            if (is_stone(a3) && a3 != a0 && a3 != a1 && a3 != a2) atomicAdd_u16(&L->libs[a3 & 0x7FFF]);
          }
        }
        __syncthreads();
      }
      // liberties of the mover's group
      int newlibs;
      const int root = newv & 0x7FFF;
      if (m == 0) {
        // createNewGroup (:661-671): its liberties are the empty neighbours after captures
        u32 e = lane < 4 ? L->pt[i + dl] : 1u;
        newlibs = __popcll(__ballot(e == 0));
      } else if (m == 1) {
        // MergeToGroup (:677-708) restated: the played point stops being a liberty; each previously
        // empty neighbour counts only if no other stone of the group already touches it.
        int kk = lane / 3, jj = lane % 3;          // lanes 0..11: neighbour kk, its j-th other side
        bool fresh = false;
        if (lane < 12) {
          int dk = (kk & 1) ? ((kk & 2) ? 1 : -1) : ((kk & 2) ? S : -S);
          // the three directions from e that do not lead back to i
          int dj;
          {
            int cand0 = -S, cand1 = -1, cand2 = S, cand3 = 1;
            int back = -dk;
            int arr[3]; int q = 0;
            if (cand0 != back) arr[q++] = cand0;
            if (cand1 != back) arr[q++] = cand1;
            if (cand2 != back) arr[q++] = cand2;
            if (cand3 != back) arr[q++] = cand3;
            dj = arr[jj];
          }
          u32 nk = kk == 0 ? (u32)n[0] : kk == 1 ? (u32)n[1] : kk == 2 ? (u32)n[2] : (u32)n[3];
          if (nk == 0) fresh = L->pt[i + dk + dj] == newv;   // another stone of the group touches e
          else fresh = true;                                  // not an (originally) empty point: ignore
        }
        u64 touched = __ballot(fresh);
        int add = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) add += (n[k] == 0) && (((touched >> (3 * k)) & 7ull) == 0);
        newlibs = (int)L->libs[root] - 1 + add;
      } else {
        // RecomputeGroupLiberties (:754-782): count empty points touching the merged group
        newlibs = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          bool lib = false;
          if (valid[k]) {
            int p = idx[k];
            if (L->pt[p] == 0)
              lib = L->pt[p - 1] == newv || L->pt[p + 1] == newv || L->pt[p - S] == newv || L->pt[p + S] == newv;
          }
          newlibs += __popcll(__ballot(lib));
        }
      }
      if (lane == 0) L->libs[root] = (u16)newlibs;
      hash ^= zob_col(zob[i], player);
      new_ko = (m == 0 && total_cap == 1 && newlibs == 1);                    // :1386
    }
    // ---- history push (go_state.cc:90-92; BoardHistory(board) board_feature.h:45-56) as bitboards
    {
      const int slot = h.hist_cnt & (HIST - 1);
      if (is_move) {
        u64 myb = 0, myw = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
          u32 v = valid[k] ? L->pt[idx[k]] : 0;
          bool st = v != 0;   // valid points are never border
          u64 bb = __ballot(st && !(v & 0x8000)), wb = __ballot(st && (v & 0x8000));
          if (lane == k) { myb = bb; myw = wb; }
        }
        if (lane < R) { L->hist[slot][0][lane] = myb; L->hist[slot][1][lane] = myw; }
      } else {
        const u64* cb = cur_bits();
        bool have = h.hist_cnt != 0;
        if (lane < 2 * R) (&L->hist[slot][0][0])[lane] = have ? cb[lane] : 0ull;
      }
    }
    // ---- header update: caps :1348-1351, ko :1384-1393, update_next_move :1225-1238
    if (lane == 0) {
      h.hash = hash;
      if (is_move) {
        if (player == S_BLACK) h.b_cap += total_cap; else h.w_cap += total_cap;
        if (new_ko) { h.ko_pt = (u16)ko_c; h.ko_color = (unsigned char)opp; h.ko_age = 0; }
        else h.ko_age++;
      }
      h.next_player = (unsigned char)opp;
      h.last_move[3] = h.last_move[2]; h.last_move[2] = h.last_move[1]; h.last_move[1] = h.last_move[0];
      h.last_move[0] = (u16)c;
      h.ply++;
      h.hist_cnt++;
      if (c != M_PASS) h.sk_len++;
      h.superko = 0;
    }
    __syncthreads();
    // ---- _check_superko (go_state.cc:96-111) for the new position, cached in the header
    if (c != M_PASS) {
      const int len = h.sk_len;
      const u16 tag = sk_tag(hash);
      bool hit = false;
      for (int base = 0; base < len; base += 64) {
        int t = base + lane;
        bool cand = t < len && L->tags[t] == tag;
        u64 bal = __ballot(cand);
        while (bal) {                       // rare: 16-bit tag match -> full hash, then full image
          int tl = (int)__builtin_ctzll(bal);
          bal &= bal - 1;
          int tt = base + tl;
          if (sk_hash[tt] == hash) {
            const u64* cb = cur_bits();
            bool same = lane < G::SKW ? sk_img[(size_t)tt * G::SKW + lane] == cb[lane] : true;
            if (__all(same)) hit = true;
          }
        }
      }
      if (hit && lane == 0) h.superko = 1;
      __syncthreads();
    }
    return 1;
  }

  __device__ __forceinline__ static void atomicAdd_u16(u16* p) {
    // LDS has no 16-bit atomic add: add into the containing dword (never carries: liberties < 2^15)
    size_t a = reinterpret_cast<size_t>(p);
    u32* w = reinterpret_cast<u32*>(a & ~size_t(3));
    atomicAdd(w, (a & 2) ? 0x10000u : 1u);
  }

  // ---- legal moves for the side to move (TryPlay :788-827 over every point) --------------------
  // legal[k] bit l = action 64k+l is playable; optionally also the "not own true eye" candidate set
  // used by the config-2 playout policy (isTrueEye, board.cc:1850-1914).
  template <bool WITH_EYES>
  __device__ __forceinline__ void legal_moves(u64 (&legal)[R], u64 (&cand)[R]) const {
    const Hdr& h = L->h;
    const int player = h.next_player;
    const u32 ownbit = player == S_WHITE ? 0x8000u : 0u;
    const int ko_i = (h.ko_age == 0 && h.ko_color == player) ? tr(h.ko_pt) : -1;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      bool ok = false, eye = false;
      if (valid[k]) {
        const int p = idx[k];
        if (L->pt[p] == 0) {
          u32 a[4] = {L->pt[p - S], L->pt[p - 1], L->pt[p + S], L->pt[p + 1]};
          ok = a[0] == 0 || a[1] == 0 || a[2] == 0 || a[3] == 0;
          if (!ok) {
            bool allown = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (a[j] == PT_BORDER) continue;
              int lb = L->libs[a[j] & 0x7FFF];
              bool own = (a[j] & 0x8000) == ownbit;
              ok |= own ? lb > 1 : lb == 1;
              allown &= own;
            }
            if (WITH_EYES && allown) {
              // isEye holds; isFakeEye :1887-1906 on the diagonals
              u32 d[4] = {L->pt[p - S - 1], L->pt[p - S + 1], L->pt[p + S - 1], L->pt[p + S + 1]};
              int nopp = 0, nb = 0;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (d[j] == PT_BORDER) ++nb;
                else if (d[j] != 0 && (d[j] & 0x8000) != ownbit) ++nopp;
              }
              bool fake = (nb > 0 && nopp >= 1) || (nb == 0 && nopp >= 2);
              eye = !fake;
            }
          }
          if (p == ko_i) ok = false;
        }
      }
      legal[k] = __ballot(ok);
      if (WITH_EYES) cand[k] = __ballot(ok && !eye);
    }
  }

  // ---- Tromp-Taylor area score (go_state.h:32-93 simple_flood_fill x2 + simple_tt_scoring) ----
  // Row-bitboard flood fill in registers: lane x holds column-bits y of row x.
  __device__ int tt_area() const {
    const u64* cb = cur_bits();
    const bool have = L->h.hist_cnt != 0;
    u32 B = 0, Wt = 0;
    const u32 rowmask = (1u << N) - 1;
    if (lane < N && have) {
      int bit0 = lane * N, w = bit0 >> 6, s = bit0 & 63;
      u64 b0 = cb[w], w0 = cb[R + w];
      u64 b1 = (w + 1 < R) ? cb[w + 1] : 0ull, w1 = (w + 1 < R) ? cb[R + w + 1] : 0ull;
      u64 bb = s ? ((b0 >> s) | (b1 << (64 - s))) : b0;
      u64 ww = s ? ((w0 >> s) | (w1 << (64 - s))) : w0;
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
    if (L->h.superko) return L->h.next_player == S_BLACK ? 1.0f : -1.0f;
    return (float)tt_area() - komi;
  }
};

}  // namespace elfgo
