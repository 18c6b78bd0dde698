// Self-play record format, host side: what the reference's GoStateExt::dumpRecord (go_state_ext.h:131-148),
// MsgResult/Record::setJsonFields (record.h:203-262), GoStateExt::addMCTSPolicy (go_state_ext.h:158-181) and the SGF string
// helpers coords2sgfstr / sgfstr2coords / str2coord / coord2str (sgf/sgf.h:21-57,87-125) produce and consume.  The JSON text
// follows nlohmann::json::dump() of the reference (compact separators, object keys in std::map order, floats widened to double
// and printed with the shortest round-trip digits in nlohmann's fixed/exponent layout), so that the reference's
// Record::createFromJson reads it back field for field.
#include "record_host.h"
#include "host_workers.h"

#include <atomic>
#include <math.h>
#include <string.h>

#include <time.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/elf_amd.h"

namespace {

// ---- shortest-digit generation: Grisu2 (F. Loitsch, "Printing Floating-Point Numbers Quickly and Accurately with Integers",
// PLDI 2010), the algorithm nlohmann::json 3.1 prints floats with.  Grisu2's digits are short and round-trip, but they are not
// always the closest/shortest decimal (std::to_chars differs on ~0.7 % of doubles), so text identity with the reference's
// json::dump() needs this very algorithm: 64-bit "do-it-yourself" floats, a cached power of ten chosen so that the scaled value's
// binary exponent lies in [-60, -32], digit generation inside the rounding interval shrunk by one unit on each side, and the
// final weeding step towards the exact scaled value.
struct DiyFp {
  uint64_t f;
  int e;
};

DiyFp diy_normalize(DiyFp x) {
  while ((x.f >> 63) == 0) { x.f <<= 1; x.e--; }
  return x;
}

// upper 64 bits of the 128-bit product, rounded (ties up)
DiyFp diy_mul(DiyFp x, DiyFp y) {
  const unsigned __int128 p = (unsigned __int128)x.f * y.f + ((unsigned __int128)1 << 63);
  return DiyFp{(uint64_t)(p >> 64), x.e + y.e + 64};
}

// 10^k as a normalized 64-bit significand, k = -300, -292, ..., 324, computed exactly with a small big-integer and rounded to nearest
struct Pow10 { uint64_t f; int e; };
struct BigUInt {
  std::vector<uint32_t> w;   // little endian
  void mul_small(uint32_t m) {
    uint64_t c = 0;
    for (auto& x : w) { c += (uint64_t)x * m; x = (uint32_t)c; c >>= 32; }
    if (c) w.push_back((uint32_t)c);
  }
  uint32_t div_small(uint32_t d) {   // in place, returns the remainder
    uint64_t r = 0;
    for (size_t i = w.size(); i-- > 0;) { r = (r << 32) | w[i]; w[i] = (uint32_t)(r / d); r %= d; }
    while (!w.empty() && w.back() == 0) w.pop_back();
    return (uint32_t)r;
  }
  int bits() const {
    if (w.empty()) return 0;
    int b = 32 * (int)(w.size() - 1);
    for (uint32_t t = w.back(); t; t >>= 1) ++b;
    return b;
  }
  bool bit(int i) const { return i >= 0 && (size_t)(i >> 5) < w.size() && ((w[i >> 5] >> (i & 31)) & 1u); }
  bool any_below(int i) const {   // any set bit at a position < i
    for (int j = 0; j < i && (size_t)(j >> 5) < w.size(); ++j) if (bit(j)) return true;
    return false;
  }
  uint64_t top64(int* exp2, bool sticky_extra) const {   // round to nearest (ties to even; `sticky_extra` = truncated tail beyond w)
    const int L = bits();
    uint64_t f = 0;
    for (int i = 0; i < 64; ++i) f = (f << 1) | (bit(L - 1 - i) ? 1u : 0u);
    const bool half = bit(L - 65);
    const bool rest = any_below(L - 65) || sticky_extra;
    int e = L - 64;
    if (half && (rest || (f & 1))) { if (++f == 0) { f = (uint64_t)1 << 63; ++e; } }
    *exp2 = e;
    return f;
  }
};

const Pow10& cached_pow10(int index) {   // index 0 <-> k = -300
  static Pow10 table[79];
  static bool ready = false;
  if (!ready) {
    for (int i = 0; i < 79; ++i) {
      const int k = -300 + 8 * i;
      BigUInt b;
      int e = 0;
      if (k >= 0) {
        b.w = {1u};
        for (int j = 0; j < k; ++j) b.mul_small(10u);
        table[i].f = b.top64(&e, false);
        table[i].e = e;
      } else {
        const int m = 1200;   // 2^m / 10^|k| keeps > 190 significant bits for |k| <= 300
        b.w.assign(m / 32 + 1, 0u);
        b.w[m / 32] = 1u << (m % 32);
        bool sticky = false;
        for (int j = 0; j < -k; ++j) sticky |= b.div_small(10u) != 0;
        table[i].f = b.top64(&e, sticky);
        table[i].e = e - m;
      }
    }
    ready = true;
  }
  return table[index];
}

// digits of v > 0 (finite): v ~= digits * 10^decimal_exponent
void grisu2_digits(double v, char* buf, int* len, int* decimal_exponent) {
  uint64_t bits;
  memcpy(&bits, &v, 8);
  const uint64_t E = bits >> 52, F = bits & (((uint64_t)1 << 52) - 1);
  const DiyFp x = E == 0 ? DiyFp{F, 1 - 1075} : DiyFp{F + ((uint64_t)1 << 52), (int)E - 1075};
  // rounding interval: half way to the neighbours; the lower neighbour is twice as close at a power of two
  const bool closer_below = F == 0 && E > 1;
  const DiyFp mp = diy_normalize(DiyFp{2 * x.f + 1, x.e - 1});
  DiyFp mm = closer_below ? DiyFp{4 * x.f - 1, x.e - 2} : DiyFp{2 * x.f - 1, x.e - 1};
  mm.f <<= (mm.e - mp.e); mm.e = mp.e;
  const DiyFp w0 = diy_normalize(x);
  // cached power: smallest k with 10^k >= 2^(alpha - e - 1), alpha = -60
  const int fexp = -60 - mp.e - 1;
  const int k = (fexp * 78913) / (1 << 18) + (fexp > 0);
  const int index = (300 + k + 7) / 8;
  const Pow10& c = cached_pow10(index);
  const int ck = -300 + 8 * index;
  const DiyFp cp{c.f, c.e};
  const DiyFp W = diy_mul(w0, cp), Wm = diy_mul(mm, cp), Wp = diy_mul(mp, cp);
  const DiyFp Mm{Wm.f + 1, Wm.e}, Mp{Wp.f - 1, Wp.e};   // shrink by one unit: the products carry an error below one unit
  *decimal_exponent = -ck;
  // ---- digit generation in [Mm, Mp]
  uint64_t delta = Mp.f - Mm.f, dist = Mp.f - W.f;
  const int sh = -Mp.e;                       // 32 <= sh <= 60
  const uint64_t one = (uint64_t)1 << sh;
  uint32_t p1 = (uint32_t)(Mp.f >> sh);
  uint64_t p2 = Mp.f & (one - 1);
  auto weed = [&](uint64_t rest, uint64_t unit) {   // move the last digit down while that brings the number closer to W
    while (rest < dist && delta - rest >= unit && (rest + unit < dist || dist - rest > rest + unit - dist)) {
      buf[*len - 1]--;
      rest += unit;
    }
  };
  uint32_t pow10 = 1;
  int n = 1;
  while (n < 10 && p1 >= pow10 * 10u) { pow10 *= 10u; ++n; }
  *len = 0;
  while (n > 0) {
    const uint32_t d = p1 / pow10;
    p1 %= pow10;
    buf[(*len)++] = (char)('0' + d);
    --n;
    const uint64_t rest = ((uint64_t)p1 << sh) + p2;
    if (rest <= delta) {
      *decimal_exponent += n;
      weed(rest, (uint64_t)pow10 << sh);
      return;
    }
    pow10 /= 10u;
  }
  int m = 0;
  for (;;) {
    p2 *= 10;
    const uint64_t d = p2 >> sh;
    p2 &= one - 1;
    buf[(*len)++] = (char)('0' + d);
    ++m;
    delta *= 10;
    dist *= 10;
    if (p2 <= delta) break;
  }
  *decimal_exponent -= m;
  weed(p2, one);
}

// nlohmann's number layout: fixed notation for 10^-4 <= v < 10^15, else d[.ddd]e+-XX
void put_double(std::string& o, double v) {
  if (!std::isfinite(v)) { o += "null"; return; }
  if (std::signbit(v)) { o += '-'; v = -v; }
  if (v == 0) { o += "0.0"; return; }
  char dg[32];
  int k = 0, dexp = 0;
  grisu2_digits(v, dg, &k, &dexp);
  const std::string digits(dg, dg + k);
  const int n = k + dexp;               // position of the decimal point relative to the first digit
  if (k <= n && n <= 15) { o += digits; o.append((size_t)(n - k), '0'); o += ".0"; return; }
  if (0 < n && n <= 15) { o.append(digits, 0, (size_t)n); o += '.'; o.append(digits, (size_t)n, std::string::npos); return; }
  if (-4 < n && n <= 0) { o += "0."; o.append((size_t)(-n), '0'); o += digits; return; }
  o += digits[0];
  if (k > 1) { o += '.'; o.append(digits, 1, std::string::npos); }
  o += 'e';
  int ex = n - 1;
  if (ex < 0) { o += '-'; ex = -ex; } else { o += '+'; }
  if (ex < 10) o += '0';
  o += std::to_string(ex);
}
void put_float(std::string& o, float v) { put_double(o, (double)v); }
void put_bool(std::string& o, bool b) { o += b ? "true" : "false"; }

inline int X(int c, int S) { return c % S - 1; }
inline int Y(int c, int S) { return c / S - 1; }

}  // namespace

SpRecordMeta elfrec_meta_from_options(const ElfSpOptions& o) {
  SpRecordMeta m{};
  m.board_size = o.board_size; m.black_ver = o.model_ver; m.white_ver = -1;   // self-play: one AI plays both colours
  m.num_threads = o.mcts.num_threads > 0 ? o.mcts.num_threads : 1; m.num_rollouts_per_thread = o.num_rollouts_per_thread; m.num_rollouts_per_batch = o.mcts.num_rollouts_per_batch;
  m.virtual_loss = o.mcts.virtual_loss; m.persistent_tree = o.persistent_tree != 0; m.use_prior = o.mcts.use_prior != 0;
  m.unexplored_q_zero = o.mcts.unexplored_q_zero != 0; m.root_unexplored_q_zero = o.mcts.root_unexplored_q_zero != 0;
  m.c_puct = o.mcts.c_puct; m.root_epsilon = o.root_epsilon; m.root_alpha = o.root_alpha;
  m.black_resign_thres = o.resign_thres; m.white_resign_thres = o.resign_thres; m.never_resign_prob = o.never_resign_prob;
  m.num_game_thread_used = o.num_games;
  m.pick_method = o.pick_method;
  return m;
}

namespace { void put_json_string(std::string& o, const std::string& t); }

void elfrec_meta_set_ts(SpRecordMeta* m, const ElfTsOptions& t) {
  m->num_threads = t.num_threads; m->num_rollouts_per_thread = t.num_rollouts_per_thread; m->num_rollouts_per_batch = t.num_rollouts_per_batch;
  m->virtual_loss = t.virtual_loss; m->persistent_tree = t.persistent_tree != 0; m->use_prior = t.use_prior != 0;
  m->unexplored_q_zero = t.unexplored_q_zero != 0; m->root_unexplored_q_zero = t.root_unexplored_q_zero != 0;
  m->c_puct = t.c_puct; m->root_epsilon = t.root_epsilon; m->root_alpha = t.root_alpha; m->pick_method = t.pick_method;
  m->max_num_moves = t.max_num_moves; m->ts_seed = t.seed; m->verbose = t.verbose != 0; m->verbose_time = t.verbose_time != 0;
  m->log_prefix.assign(t.log_prefix, strnlen(t.log_prefix, sizeof(t.log_prefix)));
}

// MsgRequest::setJsonFields (record.h:119-127): {"client_ctrl":{...},"vers":{...}} as nlohmann dumps it (keys in std::map order)
static void put_request(std::string& o, const SpRecordMeta& m) {
  o += "{\"client_ctrl\":{\"async\":";
  put_bool(o, m.async);
  o += ",\"black_resign_thres\":";
  put_float(o, m.black_resign_thres);
  o += ",\"client_type\":" + std::to_string((uint32_t)m.client_type) + ",\"never_resign_prob\":"; put_float(o, m.never_resign_prob);
  o += ",\"num_game_thread_used\":" + std::to_string(m.num_game_thread_used);
  o += ",\"player_swap\":"; put_bool(o, m.player_swap);
  o += ",\"white_resign_thres\":"; put_float(o, m.white_resign_thres);
  o += "},\"vers\":{\"black_ver\":" + std::to_string(m.black_ver) + ",\"mcts_opt\":{\"alg_opt\":{\"c_puct\":"; put_float(o, m.c_puct);
  o += ",\"root_unexplored_q_zero\":"; put_bool(o, m.root_unexplored_q_zero);
  o += ",\"unexplored_q_zero\":"; put_bool(o, m.unexplored_q_zero);
  o += ",\"use_prior\":"; put_bool(o, m.use_prior);
  o += "},\"log_prefix\":"; put_json_string(o, m.log_prefix);
  o += ",\"max_num_moves\":" + std::to_string(m.max_num_moves) + ",\"num_rollouts_per_batch\":" + std::to_string(m.num_rollouts_per_batch);
  o += ",\"num_rollouts_per_thread\":" + std::to_string(m.num_rollouts_per_thread);
  o += ",\"num_threads\":" + std::to_string(m.num_threads) + ",\"persistent_tree\":"; put_bool(o, m.persistent_tree);
  o += std::string(",\"pick_method\":\"") + (m.pick_method == 1 ? "strongest_prior" : m.pick_method == 2 ? "uniform_random" : "most_visited") +
       "\",\"root_alpha\":";
  put_float(o, m.root_alpha);
  o += ",\"root_epsilon\":"; put_float(o, m.root_epsilon);
  o += ",\"seed\":" + std::to_string(m.ts_seed) + ",\"verbose\":"; put_bool(o, m.verbose);
  o += ",\"verbose_time\":"; put_bool(o, m.verbose_time);
  o += ",\"virtual_loss\":" + std::to_string(m.virtual_loss);
  o += "},\"white_ver\":" + std::to_string(m.white_ver) + "}}";
}

std::string elfrec_record_json(const SpRecordMeta& m, const SpRecord& r) {
  std::string o;
  o.reserve(4096 + r.policies.size() * 3);
  const int P = (m.board_size + 2) * (m.board_size + 2);
  o += "{\"offline\":false,\"pri\":0.0,\"request\":";
  put_request(o, m);
  o += ",\"result\":{\"black_never_resign\":"; put_bool(o, r.never_resign);
  o += ",\"content\":\"";
  {
    std::vector<char> buf(r.moves.size() * 6 + 8);
    const int len = elfrec_coords_to_sgfstr(m.board_size, r.moves.data(), (int)r.moves.size(), buf.data(), buf.size());
    for (int i = 0; i < len; ++i) {   // coord2str of an off-board Coord can emit '`' or '\\'-range bytes; escape as nlohmann does
      const unsigned char ch = (unsigned char)buf[i];
      if (ch == '"') o += "\\\"";
      else if (ch == '\\') o += "\\\\";
      else if (ch < 0x20) { char t[8]; snprintf(t, sizeof(t), "\\u%04x", ch); o += t; }
      else o += (char)ch;
    }
  }
  o += "\",\"num_move\":" + std::to_string(r.num_move);
  const size_t np = r.policies.size() / (size_t)P;
  if (np) {   // MsgResult::setJsonFields only creates the key when there is at least one policy (record.h:212-218)
    o += ",\"policies\":[";
    for (size_t i = 0; i < np; ++i) {
      o += i ? ",[" : "[";
      for (int k = 0; k < P; ++k) { if (k) o += ','; o += std::to_string((int)r.policies[i * P + k]); }
      o += ']';
    }
    o += ']';
  }
  o += ",\"reward\":"; put_float(o, r.reward);
  o += ",\"using_models\":[";
  {
    bool first = true;   // std::set<int64_t>: ascending, without negatives (addCurrentModel, go_state_ext.h:69-74)
    if (!r.using_models.empty()) {
      for (int64_t v : r.using_models) { if (!first) o += ','; o += std::to_string(v); first = false; }
    } else {
      int64_t a = m.black_ver, b = m.white_ver;
      if (a > b) { int64_t t = a; a = b; b = t; }
      if (a >= 0) { o += std::to_string(a); first = false; }
      if (b >= 0 && b != a) { if (!first) o += ','; o += std::to_string(b); }
    }
  }
  o += "],\"values\":[";
  for (size_t i = 0; i < r.values.size(); ++i) { if (i) o += ','; put_float(o, r.values[i]); }
  o += "],\"white_never_resign\":"; put_bool(o, r.never_resign);
  o += "},\"seq\":" + std::to_string(r.seq) + ",\"thread_id\":" + std::to_string(r.thread_id) + ",\"timestamp\":" + std::to_string(r.timestamp) + "}";
  return o;
}

void elfrec_append_policy(int board_size, const int32_t* coord, const float* prob, int n, std::vector<uint8_t>* policies) {
  const size_t P = (size_t)(board_size + 2) * (board_size + 2);
  const size_t base = policies->size();
  policies->resize(base + P, 0);
  (void)elfrec_quantise_policy(board_size, coord, prob, n, policies->data() + base);
}

extern "C" {

// coords2sgfstr (sgf/sgf.h:87-95) with coord2str (:48-57): "(;B[xy];W[xy]...)", pass = "[]"
int elfrec_coords_to_sgfstr(int board_size, const uint16_t* coords, int n, char* out, size_t cap) {
  if (board_size < 1 || n < 0 || (n > 0 && !coords)) return ELFGO_E_BADARG;
  const int S = board_size + 2;
  size_t len = 0;
  auto put = [&](char c) { if (out && len < cap) out[len] = c; ++len; };
  put('(');
  for (int i = 0; i < n; ++i) {
    put(';'); put(i % 2 == 0 ? 'B' : 'W'); put('[');
    const int c = coords[i];
    if (c != 0 /* M_PASS */) { put((char)('a' + X(c, S))); put((char)('a' + Y(c, S))); }
    put(']');
  }
  put(')');
  if (out && len < cap) out[len] = 0;
  if (out && len >= cap) return ELFGO_E_BADSIZE;
  return (int)len;
}

// Sgf::load(filename, game_string) + the iterator over its entries (sgf/sgf.cc:28-57,72-127,129-240, sgf/sgf.h:21-46,125-245), restated:
//  * the header is the node after the first ';': its key/value pairs up to the next ';' or ')' (RE, SZ, KM, HA are read here);
//  * every following ';' starts an entry whose B[..] / W[..] property (the last one, if several) is the move; any other key --
//    and the second, third ... value of a multi-value property such as AB[gc][cg], whose key is then empty -- is ignored;
//  * '(' and ')' do not nest: a key/value scan stops at ';' or ')' and the next entry starts at the next ';' wherever it is, so
//    the variations of a file follow each other as if they were one line of play;
//  * a backslash hides the character after it from the scanner (an escaped ']' does not end a value);
//  * keys and values are trimmed by sgf.cc's own trim (blanks and newlines; its right end only when the left needed none);
//  * str2coord: fewer than two characters = pass; blanks and newlines before each letter are skipped; off the board (e.g. "tt" on
//    19x19) = M_INVALID (3), which GoState::forward refuses.
// An entry without a move has uninitialised fields in the reference; here it is reported as (player 0, M_INVALID).
// Returns the number of entries (all are counted, the first `cap` stored); 0 when the text has no header or no entry
// (Sgf::load returns false).
int elfrec_sgf_parse(int board_size, const char* text, int32_t* players, uint16_t* coords, int cap, ElfSgfHeader* header) {
  if (board_size < 1 || !text || cap < 0 || (cap > 0 && (!players || !coords))) return ELFGO_E_BADARG;
  const int N = board_size, S = N + 2;
  const int len = (int)strlen(text);
  const char* s = text;
  // sgf.cc:15-24 trim: blanks and newlines off both ends -- except that it returns substr(l, r + 1), a COUNT of r + 1 characters
  // from l, so the right end is only trimmed when nothing was trimmed on the left
  auto trim = [](const std::string& t) {
    int l = 0;
    while (l < (int)t.size() && (t[l] == ' ' || t[l] == '\n')) l++;
    int r = (int)t.size() - 1;
    while (r >= 0 && (t[r] == ' ' || t[r] == '\n')) r--;
    return t.substr((size_t)l, (size_t)(r + 1));
  };
  auto str2coord = [&](const std::string& v) -> int {
    if (v.size() < 2) return 0;                          // M_PASS
    size_t i = 0;
    while (i < v.size() && (v[i] == '\n' || v[i] == ' ')) i++;
    if (i == v.size()) return 3;                         // M_INVALID
    const int x = v[i] - 'a';
    i++;
    while (i < v.size() && (v[i] == '\n' || v[i] == ' ')) i++;
    if (i == v.size()) return 3;
    const int y = v[i] - 'a';
    if (x < 0 || x >= N || y < 0 || y >= N) return 3;
    return (y + 1) * S + (x + 1);
  };
  // get_key_values: calls cb(key, value) for every pair from `from`; returns where the scan stopped
  auto scan = [&](int from, const std::function<void(const std::string&, const std::string&)>& cb) {
    int i, start = from, k0 = 0, k1 = 0;
    bool in_value = false, done = false, backslash = false;
    for (i = from; i < len && !done; ++i) {
      if (s[i] == '\\') { backslash = !backslash; continue; }
      if (backslash) { backslash = false; continue; }
      const char c = s[i];
      if (!in_value) {
        if (c == '[') { k0 = start; k1 = i; start = i + 1; in_value = true; }
        else if (c == ';' || c == ')') { --i; done = true; }
      } else if (c == ']') {
        cb(std::string(s + k0, (size_t)(k1 - k0)), std::string(s + start, (size_t)(i - start)));
        start = i + 1;
        in_value = false;
      }
    }
    return i;
  };
  ElfSgfHeader h;
  h.size = N; h.komi = 7.5f; h.handi = 0; h.winner = 3 /* S_OFF_BOARD */; h.win_margin = 0.0f;
  int i = 0;
  while (i < len && s[i] != ';') i++;
  if (i >= len) return 0;
  i++;
  int next = scan(i, [&](const std::string& key, const std::string& value) {
    const std::string v = trim(value), k = trim(key);
    if (k == "RE") {
      if (!v.empty()) {
        h.winner = (v[0] == 'B' || v[0] == 'b') ? 1 : 2;
        if (v.size() >= 3) {                             // stof(v.substr(2)), or a reason such as "R" / "id"
          const std::string m = v.substr(2);
          char* e = nullptr;
          const float f = strtof(m.c_str(), &e);
          if (e != m.c_str()) h.win_margin = f;
        }
      }
    } else if (k == "SZ") h.size = atoi(v.c_str());
    else if (k == "KM") h.komi = strtof(v.c_str(), nullptr);
    else if (k == "HA") h.handi = atoi(v.c_str());
  });
  int count = 0;
  while (true) {
    int j = next;
    while (j < len && s[j] != ';') ++j;
    if (j >= len) break;
    ++j;
    int player = 0, move = 3;
    if (j < len && s[j] == '(') {
      // ";(": the reference parses a child list from here to the end of the text, then looks for ')' at the offset its last,
      // failing load call has reset to 0 -- finds '(' there, calls the file corrupted and drops this entry and everything after
      // it (sgf.cc:216-226): the list of entries ends here
      break;
    } else {
      next = scan(j, [&](const std::string& key, const std::string& value) {
        const std::string v = trim(value), k = trim(key);
        if (k.size() == 1 && (k[0] == 'B' || k[0] == 'W')) { player = k[0] == 'B' ? 1 : 2; move = str2coord(v); }
      });
    }
    if (count < cap) { players[count] = player; coords[count] = (uint16_t)move; }
    ++count;
  }
  if (count == 0) return 0;
  if (header) *header = h;
  return count;
}

// GoStateExt::dumpSgf (go_state_ext.cc:26-82): the SGF text finish_game writes to <dump_record_prefix>_<game>_<seq>_<B|W>.sgf
// (game_selfplay.cc:133-135, go_state_ext.h:48-56): result, player names, komi, every move with its predicted value.  The same
// iostream / std::to_string calls as the reference, so that the numbers print alike.  git_hash / git_staged: the two lines of the
// opening comment (the reference prints its build's GIT_COMMIT_HASH / GIT_STAGED); NULL = this library's version and "0".
int64_t elfrec_game_sgf(const ElfSpOptions* opt, const uint16_t* moves, int num_moves, const float* values, int num_values,
                        float final_value, const char* filename, const char* git_hash, const char* git_staged, char* out, size_t cap) {
  if (!opt || num_moves < 0 || num_values < 0 || (num_moves && !moves) || (num_values && !values) || !filename) return ELFGO_E_BADARG;
  const int S = opt->board_size + 2;
  std::stringstream ss;
  const float value = final_value;
  std::string result;
  if (std::abs(value) == 1.0) result = (value > 0.0 ? "B+R" : "W+R");
  else result = (value > 0.0 ? "B+" + std::to_string(value) : "W+" + std::to_string(-value));
  std::stringstream cm;
  cm << "Filename: " << filename << std::endl;
  cm << "Git hash: " << (git_hash ? git_hash : elfgo_version()) << std::endl;
  cm << "Staged: " << (git_staged ? git_staged : "0") << std::endl;
  ss << "(;SZ[" << opt->board_size << "]RE[" << result << "]C[" + cm.str() + "]";
  std::string black_name = "MCTS", white_name = "MCTS";      // use_mcts: this engine's AIs always search
  if (opt->black_use_policy_network_only) black_name += "(policy only)";
  if (opt->white_use_policy_network_only) white_name += "(policy only)";
  ss << "PB[" << black_name << "]PW[" << white_name << "]KM[" << opt->mcts.komi << "]";
  for (int i = 0; i < num_moves; ++i) {
    const int c = moves[i];
    std::string mv;                                           // coord2str (sgf/sgf.h:48-57)
    if (c != 0 /* M_PASS */) { mv += (char)('a' + X(c, S)); mv += (char)('a' + Y(c, S)); }
    ss << ";" << (i % 2 == 0 ? "B" : "W") << "[" << mv << "]";
    std::string comments = std::to_string(i + 1) + ": ";
    if (i < num_values) comments += "PredV: " + std::to_string(values[i]);
    ss << "C[" << comments << "]";
  }
  ss << ")";
  const std::string t = ss.str();
  if (!out) return (int64_t)t.size();
  if (cap <= t.size()) return ELFGO_E_BADSIZE;
  memcpy(out, t.data(), t.size());
  out[t.size()] = 0;
  return (int64_t)t.size();
}

// sgfstr2coords (sgf/sgf.h:97-125) with str2coord (:21-46); returns the number of moves (all of them are counted, the
// first `cap` are stored)
int elfrec_sgfstr_to_coords(int board_size, const char* sgf, uint16_t* out, int cap) {
  if (board_size < 1 || !sgf || cap < 0 || (cap > 0 && !out)) return ELFGO_E_BADARG;
  const int N = board_size, S = N + 2;
  const size_t L = strlen(sgf);
  if (L == 0 || sgf[0] != '(') return 0;
  int cnt = 0;
  size_t i = 1;
  while (true) {
    if (i >= L || sgf[i] != ';') break;
    while (i < L && sgf[i] != '[') i++;
    if (i == L) break;
    i++;
    size_t j = i;
    while (j < L && sgf[j] != ']') j++;
    if (j == L) break;
    // str2coord(sgf.substr(i, j - i))
    int coord;
    const size_t sl = j - i;
    const char* s = sgf + i;
    if (sl < 2) coord = 0;   // M_PASS
    else {
      size_t k = 0;
      while (k < sl && (s[k] == '\n' || s[k] == ' ')) k++;
      if (k == sl) coord = 3;   // M_INVALID
      else {
        const int x = s[k] - 'a';
        k++;
        while (k < sl && (s[k] == '\n' || s[k] == ' ')) k++;
        if (k == sl) coord = 3;
        else {
          const int y = s[k] - 'a';
          coord = (x >= 0 && x < N && y >= 0 && y < N) ? (y + 1) * S + (x + 1) : 3;   // ON_BOARD, OFFSETXY (board.h:183-186)
        }
      }
    }
    if (cnt < cap) out[cnt] = (uint16_t)coord;
    cnt++;
    i = j + 1;
  }
  return cnt;
}

// Record JSON from plain arrays (what elfsp_pop_record returns for a finished game); usable without a GPU
int elfrec_record_to_json2(const ElfSpOptions* opt, const ElfSpRequest* request, const uint16_t* moves, int num_moves,
                           const uint8_t* policies, int num_policies, const float* values, int num_values, float reward, int never_resign,
                           int seq, uint64_t thread_id, uint64_t timestamp, char* out, size_t cap) {
  if (!opt || num_moves < 0 || num_policies < 0 || num_values < 0 || (num_moves && !moves) || (num_policies && !policies) ||
      (num_values && !values)) return ELFGO_E_BADARG;
  SpRecord r;
  const size_t P = (size_t)(opt->board_size + 2) * (opt->board_size + 2);
  r.moves.assign(moves, moves + num_moves);
  r.policies.assign(policies, policies + (size_t)num_policies * P);
  r.values.assign(values, values + num_values);
  r.reward = reward; r.never_resign = never_resign != 0; r.num_move = num_moves; r.seq = seq; r.thread_id = thread_id; r.timestamp = timestamp;
  SpRecordMeta m = elfrec_meta_from_options(*opt);
  if (request) {
    m.black_ver = request->black_ver; m.white_ver = request->white_ver;
    m.black_resign_thres = request->black_resign_thres; m.white_resign_thres = request->white_resign_thres;
    m.never_resign_prob = request->never_resign_prob; m.num_game_thread_used = request->num_game_thread_used;
    m.player_swap = request->player_swap != 0; m.async = request->async != 0;
    m.client_type = request->client_type != 0 ? request->client_type : 1;
  }
  const std::string t = elfrec_record_json(m, r);
  if (out && cap > t.size()) { memcpy(out, t.data(), t.size()); out[t.size()] = 0; }
  else if (out) return ELFGO_E_BADSIZE;
  return (int)t.size();
}

int elfrec_record_to_json(const ElfSpOptions* opt, const uint16_t* moves, int num_moves, const uint8_t* policies, int num_policies,
                          const float* values, int num_values, float reward, int never_resign, int seq, uint64_t thread_id,
                          uint64_t timestamp, char* out, size_t cap) {
  return elfrec_record_to_json2(opt, nullptr, moves, num_moves, policies, num_policies, values, num_values, reward, never_resign, seq,
                                thread_id, timestamp, out, cap);
}

// GoStateExt::addMCTSPolicy (go_state_ext.h:158-181): out[(N+2)^2] <- 0; out[coord[k]] = (unsigned char)(prob[k] / max * 255)
int elfrec_quantise_policy(int board_size, const int32_t* coord, const float* prob, int n, uint8_t* out) {
  if (board_size < 1 || n < 0 || !out || (n > 0 && (!coord || !prob))) return ELFGO_E_BADARG;
  const int P = (board_size + 2) * (board_size + 2);
  float max_val = 0.0;
  for (int k = 0; k < n; ++k) max_val = prob[k] > max_val ? prob[k] : max_val;   // std::max(max_val, entry.second)
  memset(out, 0, (size_t)P);
  for (int k = 0; k < n; ++k) {
    if (coord[k] < 0 || coord[k] >= P) return ELFGO_E_BADARG;
    out[coord[k]] = static_cast<unsigned char>(prob[k] / max_val * 255);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The trainer's replay buffer and GoGameTrain::act's draws (host only).  elf::shared::ReaderQueuesT<Record>
// (elf/distributed/shared_reader.h:165-340): num_reader queues, each a deque bounded by queue_max_size (ReaderQueueT::Insert :88-100),
// filled by TrainCtrl::OnReceive with InsertWithParity(record, rng, reward > 0) (train/game_ctrl.h:306-311; shared_reader.h:212-219):
// games Black won go to the odd queues, the others to the even ones.  A record here is a handle (a slot of an ElfReplay store).
struct ElfReaderQueues {
  struct Rec { int32_t slot, num_moves; };
  std::vector<std::deque<Rec>> qs;
  size_t queue_min_size = 1, queue_max_size = 1000;
  int parity_sizes[2] = {0, 0};
  bool min_size_satisfied = false;
  std::mt19937 insert_rng;                 // TrainCtrl::rng_ (game_ctrl.h:244,367)
  std::vector<std::mt19937> thread_rng;    // GoGameBase::_rng of every GoGameTrain game thread (common/game_base.h:32-38)
  int64_t next_act = 0;                    // acts go round the threads
};

int elfrq_create(int num_reader, int queue_min_size, int queue_max_size, uint32_t insert_seed, ElfReaderQueues** out) {
  if (!out || num_reader < 2 || (num_reader & 1) || queue_min_size < 1 || queue_max_size < queue_min_size) return ELFGO_E_BADARG;
  ElfReaderQueues* q = new ElfReaderQueues();
  q->qs.resize((size_t)num_reader);
  q->queue_min_size = (size_t)queue_min_size;
  q->queue_max_size = (size_t)queue_max_size;
  q->insert_rng.seed(insert_seed != 0 ? insert_seed : (uint32_t)time(NULL));   // the reference: rng_(time(NULL))
  q->thread_rng.resize(1);
  q->thread_rng[0].seed(1);
  *out = q;
  return 0;
}

int elfrq_destroy(ElfReaderQueues* q) {
  delete q;
  return 0;
}

int elfrq_insert(ElfReaderQueues* q, int32_t slot, int32_t num_moves, int black_win, int32_t* evicted) {
  if (!q || slot < 0 || num_moves < 0) return ELFGO_E_BADARG;
  const int ii = (int)(q->insert_rng() % (q->qs.size() / 2));
  const int idx = 2 * ii + (black_win ? 1 : 0);
  std::deque<ElfReaderQueues::Rec>& buf = q->qs[(size_t)idx];
  buf.push_back({slot, num_moves});
  int delta = 1;
  int32_t out = -1;
  while (buf.size() > q->queue_max_size) {       // at most one: every insert adds one
    out = buf.front().slot;
    buf.pop_front();
    delta--;
  }
  q->parity_sizes[idx % 2] += delta;
  if (evicted) *evicted = out;
  return idx;
}

int elfrq_sizes(const ElfReaderQueues* q, int32_t* per_queue) {
  if (!q || !per_queue) return ELFGO_E_BADARG;
  for (size_t i = 0; i < q->qs.size(); ++i) per_queue[i] = (int32_t)q->qs[i].size();
  return (int)q->qs.size();
}

// One generator per GoGameTrain game thread.  seed != 0: thread t is seeded seed + t (this repository's rule for games, see
// ElfSpOptions: the reference proper gives every thread `seed` itself, i.e. identical streams); seed == 0: elf_utils::get_seed
// (elf/utils/utils.h:50-57) of t ^ job_hash, the reference's time-based default.
int elfrq_set_threads(ElfReaderQueues* q, int num_threads, int64_t seed, uint64_t job_hash) {
  if (!q || num_threads < 1) return ELFGO_E_BADARG;
  q->thread_rng.resize((size_t)num_threads);
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  const int64_t now_ms = (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
  for (int t = 0; t < num_threads; ++t) {
    uint64_t sd;
    if (seed != 0) sd = (uint64_t)(seed + t);
    else {
      const int32_t idx = (int32_t)(uint32_t)((uint64_t)t ^ job_hash);
      const int32_t term = (int32_t)((uint32_t)idx * 2341479u);
      sd = (uint64_t)(((now_ms / 1000) * 1000 + now_ms + (int64_t)term) % 100000000ll);
    }
    q->thread_rng[(size_t)t].seed((std::mt19937::result_type)sd);
  }
  q->next_act = 0;
  return 0;
}

// GoGameTrain::act (train/game_train.cc:23-58) num_acts times, the game threads taking turns; kNumState = 64 states per act, each:
//   getSamplerWithParity (shared_reader.h:251-274): queue pair rng() % (nq / 2); the odd one if uniform_real(0, 1) > even_ratio,
//     even_ratio = even / (even + odd + 1e-6) clamped to [0.45, 0.55] -- Black's and White's wins are sampled about equally often
//     whatever their share of the buffer;
//   Sampler::sample (:47-61): record rng() % queue size (a queue below queue_min_size gives none: the state is drawn again);
//   switchRandomMove (go_state_ext.h:282-296): a record with fewer than num_future_actions moves is dropped and the state drawn
//     again, else move_to = rng() % (num_moves - num_future_actions + 1);
//   generateD4Code (:298-300): rng() % 8.
// Host arrays of num_acts * 64.  Before the first draw every queue must hold queue_min_size records (the reference waits there,
// wait_for_sufficient_data :327-336): ELFGO_E_BADARG otherwise, and when no record is long enough.
int elfrq_draw(ElfReaderQueues* q, int num_acts, int num_future_actions, int32_t* slot, int32_t* move_to, int32_t* d4) {
  if (!q || num_acts < 0 || num_future_actions < 1 || !slot || !move_to || !d4) return ELFGO_E_BADARG;
  if (!q->min_size_satisfied) {
    for (const auto& b : q->qs) if (b.size() < q->queue_min_size) return ELFGO_E_BADARG;
    q->min_size_satisfied = true;
  }
  bool any = false;   // a record long enough, in a queue the sampler accepts (Sampler::sample refuses queues below queue_min_size)
  for (const auto& b : q->qs)
    if (b.size() >= q->queue_min_size)
      for (const auto& r : b) any = any || r.num_moves > num_future_actions - 1;
  if (!any) return ELFGO_E_BADARG;
  const float kSafeMargin = 0.45;
  const int kNumState = 64;
  // The acts go round the game threads; every thread draws from its own generator and only READS the queues, so the acts of
  // different threads are independent of each other: thread tt's acts (in their order) are one unit of work for the host pool.
  const int64_t T = (int64_t)q->thread_rng.size(), first = q->next_act;
  std::atomic<bool> stuck{false};
  const std::vector<std::mt19937> rng_before(q->thread_rng);   // a failed call leaves the draw state as it found it
  auto acts_of_thread = [&](size_t tt) {
    std::mt19937& rng = q->thread_rng[tt];
    for (int64_t a = (((int64_t)tt - first) % T + T) % T; a < num_acts; a += T) {
      for (int i = 0; i < kNumState; ++i) {
        const size_t o = (size_t)a * kNumState + i;
        // The reference's game thread retries for ever (it waits for data); a synchronous call must come back: the eligibility of
        // some record was checked above, so a draw that has not found one after 2^20 tries means the queues changed under it
        // (or an eligible record sits in a queue that fell below queue_min_size): the call fails instead of hanging the host pool.
        int tries = 0;
        while (true) {
          if (++tries > (1 << 20)) { stuck.store(true); slot[o] = 0; move_to[o] = 0; break; }
          const int even = q->parity_sizes[0], odd = q->parity_sizes[1];
          float even_ratio = static_cast<float>(even) / (even + odd + 1e-6);
          even_ratio = std::max(even_ratio, kSafeMargin);
          even_ratio = std::min(even_ratio, 1.0f - kSafeMargin);
          std::uniform_real_distribution<> dis(0.0, 1.0);
          int idx = (int)(rng() % (q->qs.size() / 2));
          idx *= 2;
          if (dis(rng) > even_ratio) idx++;
          const std::deque<ElfReaderQueues::Rec>& buf = q->qs[(size_t)idx];
          if (buf.size() < q->queue_min_size) continue;
          const ElfReaderQueues::Rec& r = buf[rng() % buf.size()];
          if (r.num_moves <= num_future_actions - 1) continue;
          slot[o] = r.slot;
          move_to[o] = (int32_t)(rng() % (size_t)(r.num_moves - num_future_actions + 1));
          break;
        }
        d4[o] = (int32_t)(rng() % 8);
      }
    }
  };
  const unsigned nt = (unsigned)std::min<int64_t>(std::min<int64_t>(T, num_acts), host_worker_count(32));   // one game thread per worker at the trainer's 2048 / 64 = 32
  if (nt < 2 || num_acts < 8) { for (int64_t tt = 0; tt < T; ++tt) acts_of_thread((size_t)tt); }
  else HostWorkers::get().run((size_t)T, nt, acts_of_thread);
  if (stuck.load()) {
    // no eligible record was found within the bound: the generators and the act counter go back to their state before the call, so
    // that a retry (after the queues have been refilled) draws what a clean call would have drawn; ELFGO_E_NODATA, not BADARG
    q->thread_rng = rng_before;
    return ELFGO_E_NODATA;
  }
  q->next_act += num_acts;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The client's wire formats (train/distri_client.h): what it sends -- Records = {identity, states, records} (common/record.h:401-470),
// kept and dumped by GuardedRecords (:111-170) -- and what it receives, MsgRequestSeq (common/record.h:152-171).  Text as
// nlohmann::json::dump() writes it: compact, object keys in std::map order.
namespace {

void put_json_string(std::string& o, const std::string& t) {   // nlohmann escape_string, ensure_ascii = false
  o += '"';
  for (unsigned char ch : t) {
    switch (ch) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (ch < 0x20) { char t8[8]; snprintf(t8, sizeof(t8), "\\u%04x", ch); o += t8; }
        else o += (char)ch;
    }
  }
  o += '"';
}

// a reader for the JSON the reference's server sends: objects, arrays, strings, numbers, true / false / null
struct JValue {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  double num = 0;
  int64_t inum = 0;
  bool is_int = false;
  std::string str;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;
  const JValue* get(const char* key) const {
    if (kind != OBJ) return nullptr;
    for (const auto& kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct JReader {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool lit(const char* t) { const size_t n = strlen(t); if ((size_t)(end - p) >= n && !memcmp(p, t, n)) { p += n; return true; } return false; }
  JValue value(int depth = 0) {
    JValue v;
    ws();
    if (p >= end || depth > 32) { ok = false; return v; }
    if (*p == '{') {
      v.kind = JValue::OBJ; ++p; ws();
      if (p < end && *p == '}') { ++p; return v; }
      while (ok) {
        ws();
        JValue k = value(depth + 1);
        if (!ok || k.kind != JValue::STR) { ok = false; break; }
        ws();
        if (p >= end || *p != ':') { ok = false; break; }
        ++p;
        v.obj.emplace_back(k.str, value(depth + 1));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; break; }
        ok = false;
      }
    } else if (*p == '[') {
      v.kind = JValue::ARR; ++p; ws();
      if (p < end && *p == ']') { ++p; return v; }
      while (ok) {
        v.arr.push_back(value(depth + 1));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; break; }
        ok = false;
      }
    } else if (*p == '"') {
      v.kind = JValue::STR; ++p;
      while (p < end && *p != '"') {
        if ((unsigned char)*p < 0x20) { ok = false; return v; }   // raw control characters are not JSON (nlohmann: parse_error 101)
        if (*p == '\\' && p + 1 < end) {
          ++p;
          switch (*p) {
            case 'n': v.str += '\n'; break; case 't': v.str += '\t'; break; case 'r': v.str += '\r'; break;
            case 'b': v.str += '\b'; break; case 'f': v.str += '\f'; break;
            case '"': case '\\': case '/': v.str += *p; break;
            case 'u': {
              auto hex4 = [&](const char* q, unsigned* out) -> bool {
                if (end - q < 4) return false;
                unsigned code = 0;
                for (int i = 0; i < 4; ++i) {
                  const char c = q[i];
                  if (!((c >= '0' && c <= '9') || ((c | 32) >= 'a' && (c | 32) <= 'f'))) return false;
                  code = code * 16 + (c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
                }
                *out = code;
                return true;
              };
              unsigned code = 0;
              if (!hex4(p + 1, &code)) { ok = false; return v; }
              p += 4;
              if (code >= 0xD800 && code <= 0xDBFF) {                      // a high surrogate must be followed by \uDC00..DFFF
                unsigned lo = 0;
                if (end - p < 7 || p[1] != '\\' || p[2] != 'u' || !hex4(p + 3, &lo) || lo < 0xDC00 || lo > 0xDFFF) { ok = false; return v; }
                p += 6;
                code = 0x10000 + ((code - 0xD800) << 10) + (lo - 0xDC00);
              } else if (code >= 0xDC00 && code <= 0xDFFF) { ok = false; return v; }
              if (code < 0x80) v.str += (char)code;
              else if (code < 0x800) { v.str += (char)(0xC0 | (code >> 6)); v.str += (char)(0x80 | (code & 63)); }
              else if (code < 0x10000) { v.str += (char)(0xE0 | (code >> 12)); v.str += (char)(0x80 | ((code >> 6) & 63)); v.str += (char)(0x80 | (code & 63)); }
              else { v.str += (char)(0xF0 | (code >> 18)); v.str += (char)(0x80 | ((code >> 12) & 63)); v.str += (char)(0x80 | ((code >> 6) & 63)); v.str += (char)(0x80 | (code & 63)); }
              break;
            }
            default: ok = false; return v;                                 // unknown escape
          }
          ++p;
        } else v.str += *p++;
      }
      if (p >= end) { ok = false; return v; }
      ++p;
    } else if (lit("true")) { v.kind = JValue::BOOL; v.b = true; }
    else if (lit("false")) { v.kind = JValue::BOOL; v.b = false; }
    else if (lit("null")) { v.kind = JValue::NUL; }
    else {
      // the JSON number grammar first: -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?  (strtod alone also takes hex, inf / nan, a leading
      // '+' and leading zeros, all of which nlohmann refuses)
      const char* c = p;
      if (c < end && *c == '-') ++c;
      if (c >= end || *c < '0' || *c > '9') { ok = false; return v; }
      if (*c == '0') ++c; else while (c < end && *c >= '0' && *c <= '9') ++c;
      v.is_int = true;
      if (c < end && *c == '.') {
        v.is_int = false; ++c;
        if (c >= end || *c < '0' || *c > '9') { ok = false; return v; }
        while (c < end && *c >= '0' && *c <= '9') ++c;
      }
      if (c < end && (*c == 'e' || *c == 'E')) {
        v.is_int = false; ++c;
        if (c < end && (*c == '+' || *c == '-')) ++c;
        if (c >= end || *c < '0' || *c > '9') { ok = false; return v; }
        while (c < end && *c >= '0' && *c <= '9') ++c;
      }
      const std::string tok(p, c);                  // the text need not be NUL-terminated at `end`
      char* q = nullptr;
      v.kind = JValue::NUM;
      v.num = strtod(tok.c_str(), &q);
      v.inum = v.is_int ? strtoll(tok.c_str(), nullptr, 10) : (int64_t)v.num;
      q = const_cast<char*>(c);
      p = q;
    }
    return v;
  }
};

// nlohmann's implicit conversions as the reference's JSON_LOAD uses them (target.field = j["field"]), by target type:
//   int          numbers (a float is truncated) and booleans (1 / 0)
//   int64_t      numbers only (it is the library's own integer type: no conversion from a boolean)
//   float        numbers and booleans
//   bool         booleans only
//   std::string  strings only
//   ClientType   numbers only, stored in the enum's unsigned underlying type
// Return: 1 = converted, 0 = the key is absent, -1 = present with a type the conversion throws on.
int jget(const JValue* o, const char* k, const JValue** v) { *v = o ? o->get(k) : nullptr; return *v ? 1 : 0; }
int jint32(const JValue* o, const char* k, int32_t* out) {
  const JValue* v;
  if (!jget(o, k, &v)) return 0;
  if (v->kind == JValue::NUM) { *out = v->is_int ? (int32_t)v->inum : (int32_t)v->num; return 1; }
  if (v->kind == JValue::BOOL) { *out = v->b ? 1 : 0; return 1; }
  return -1;
}
int jint64(const JValue* o, const char* k, int64_t* out) {
  const JValue* v;
  if (!jget(o, k, &v)) return 0;
  if (v->kind != JValue::NUM) return -1;
  *out = v->is_int ? v->inum : (int64_t)v->num;
  return 1;
}
int jfloat(const JValue* o, const char* k, float* out) {
  const JValue* v;
  if (!jget(o, k, &v)) return 0;
  if (v->kind == JValue::NUM) { *out = (float)v->num; return 1; }
  if (v->kind == JValue::BOOL) { *out = v->b ? 1.0f : 0.0f; return 1; }
  return -1;
}
int jbool(const JValue* o, const char* k, bool* out) {
  const JValue* v;
  if (!jget(o, k, &v)) return 0;
  if (v->kind != JValue::BOOL) return -1;
  *out = v->b;
  return 1;
}
int jenum(const JValue* o, const char* k, int32_t* out) {
  const JValue* v;
  if (!jget(o, k, &v)) return 0;
  if (v->kind != JValue::NUM) return -1;
  *out = (int32_t)(uint32_t)(v->is_int ? v->inum : (int64_t)v->num);
  return 1;
}
// the number / reward fields of a Record this library wrote itself
bool jnum(const JValue* o, const char* k, double* out) { const JValue* v = o ? o->get(k) : nullptr; if (!v || v->kind != JValue::NUM) return false; *out = v->num; return true; }
bool jint(const JValue* o, const char* k, int64_t* out) { return jint64(o, k, out) == 1; }

}  // namespace

struct ElfClientRecords {
  std::string identity;
  std::unordered_map<int, ElfThreadState> states;   // the reference's own container: its iteration order is the order of "states"
  std::vector<std::string> records;                 // Record JSON texts
};

int elfrec_client_create(const char* identity, ElfClientRecords** out) {
  if (!out) return ELFGO_E_BADARG;
  ElfClientRecords* c = new ElfClientRecords();
  c->identity = identity ? identity : "";
  *out = c;
  return 0;
}

int elfrec_client_destroy(ElfClientRecords* c) {
  delete c;
  return 0;
}

// GuardedRecords::feed (distri_client.h:118-121): records_.addRecord(s.dumpRecord()); record_json = one Record as elfsp_pop_record /
// elfrec_record_to_json give it
int elfrec_client_feed(ElfClientRecords* c, const char* record_json) {
  if (!c || !record_json || record_json[0] != '{') return ELFGO_E_BADARG;
  {                                               // the text is spliced into the message verbatim: it has to be one JSON object
    JReader rd{record_json, record_json + strlen(record_json)};
    const JValue v = rd.value();
    rd.ws();
    if (!rd.ok || v.kind != JValue::OBJ || rd.p != rd.end) return ELFGO_E_BADARG;
  }
  c->records.emplace_back(record_json);
  return 0;
}

// Records::updateState (record.h:422-424): states[ts.thread_id] = ts
int elfrec_client_update_state(ElfClientRecords* c, const ElfThreadState* ts) {
  if (!c || !ts) return ELFGO_E_BADARG;
  c->states[ts->thread_id] = *ts;
  return 0;
}

int elfrec_client_size(const ElfClientRecords* c) { return c ? (int)c->records.size() : ELFGO_E_BADARG; }

// GuardedRecords::dumpAndClear (distri_client.h:156-169): Records::dumpJsonString() (record.h:426-439,459-463), then clear().
// Returns the length of the text; with out == NULL nothing is cleared (a size query); with a buffer too small ELFGO_E_BADSIZE.
int64_t elfrec_client_dump_and_clear(ElfClientRecords* c, char* out, size_t cap) {
  if (!c) return ELFGO_E_BADARG;
  std::string o = "{\"identity\":";
  put_json_string(o, c->identity);
  if (!c->records.empty()) {
    o += ",\"records\":[";
    for (size_t i = 0; i < c->records.size(); ++i) { if (i) o += ','; o += c->records[i]; }
    o += ']';
  }
  if (!c->states.empty()) {
    o += ",\"states\":[";
    bool first = true;
    for (const auto& t : c->states) {
      if (!first) o += ',';
      first = false;
      const ElfThreadState& ts = t.second;
      o += "{\"black\":" + std::to_string(ts.black) + ",\"move_idx\":" + std::to_string(ts.move_idx) + ",\"seq\":" + std::to_string(ts.seq) +
           ",\"thread_id\":" + std::to_string(ts.thread_id) + ",\"white\":" + std::to_string(ts.white) + "}";
    }
    o += ']';
  }
  o += '}';
  if (!out) return (int64_t)o.size();
  if (cap <= o.size()) return ELFGO_E_BADSIZE;
  memcpy(out, o.data(), o.size());
  out[o.size()] = 0;
  c->states.clear();      // Records::clear: the maps keep their bucket counts, as the reference's do
  c->records.clear();
  return (int64_t)o.size();
}

// MsgRequestSeq::createFromJson (record.h:161-166) -> MsgRequest::createFromJson (:129-134) -> ModelPair (black_ver, white_ver,
// mcts_opt = TSOptions::createFromJson, tree_search_options.h:196-213) and ClientCtrl::createFromJson (record.h:49-66: player_swap
// may be missing in a self-play request, async may be missing always).  A missing mandatory field is the reference's
// "... cannot not be found!" exception: ELFGO_E_BADARG here.
int elfrec_parse_request_seq(const char* text, ElfSpRequest* request, int64_t* seq, ElfTsOptions* mcts_opt) {
  if (!text || !request) return ELFGO_E_BADARG;
  JReader rd{text, text + strlen(text)};
  const JValue root = rd.value();
  rd.ws();
  if (!rd.ok || rd.p != rd.end || root.kind != JValue::OBJ) return ELFGO_E_BADARG;
  const JValue* req = root.get("request");
  const JValue* vers = req ? req->get("vers") : nullptr;
  const JValue* ctrl = req ? req->get("client_ctrl") : nullptr;
  const JValue* mo = vers ? vers->get("mcts_opt") : nullptr;
  const JValue* alg = mo ? mo->get("alg_opt") : nullptr;
  int64_t sq = -1;
  if (!vers || vers->kind != JValue::OBJ || !ctrl || ctrl->kind != JValue::OBJ || !mo || mo->kind != JValue::OBJ || !alg ||
      alg->kind != JValue::OBJ || jint64(&root, "seq", &sq) != 1)
    return ELFGO_E_BADARG;
  ElfSpRequest q;
  memset(&q, 0, sizeof(q));
  if (jint64(vers, "black_ver", &q.black_ver) != 1 || jint64(vers, "white_ver", &q.white_ver) != 1) return ELFGO_E_BADARG;
  ElfTsOptions t;
  memset(&t, 0, sizeof(t));
  bool b = false;
  int32_t i32 = 0;
#define NEED_INT(obj, key, dst) do { if (jint32(obj, key, &i32) != 1) return ELFGO_E_BADARG; dst = i32; } while (0)
#define NEED_FLT(obj, key, dst) do { if (jfloat(obj, key, &dst) != 1) return ELFGO_E_BADARG; } while (0)
#define NEED_BOOL(obj, key, dst) do { if (jbool(obj, key, &b) != 1) return ELFGO_E_BADARG; dst = b ? 1 : 0; } while (0)
  NEED_INT(mo, "max_num_moves", t.max_num_moves); NEED_INT(mo, "num_threads", t.num_threads);
  NEED_INT(mo, "num_rollouts_per_thread", t.num_rollouts_per_thread); NEED_INT(mo, "num_rollouts_per_batch", t.num_rollouts_per_batch);
  NEED_BOOL(mo, "verbose", t.verbose); NEED_BOOL(mo, "verbose_time", t.verbose_time); NEED_INT(mo, "seed", t.seed);
  NEED_BOOL(mo, "persistent_tree", t.persistent_tree);
  {
    const JValue* pm = mo->get("pick_method");
    const JValue* lp = mo->get("log_prefix");
    if (!pm || pm->kind != JValue::STR || !lp || lp->kind != JValue::STR) return ELFGO_E_BADARG;
    t.pick_method = pm->str == "most_visited" ? ELFSP_PICK_MOST_VISITED : pm->str == "strongest_prior" ? ELFSP_PICK_STRONGEST_PRIOR :
                    pm->str == "uniform_random" ? ELFSP_PICK_UNIFORM_RANDOM : -1;   // -1: the search would throw "Unknown pick method"
    snprintf(t.log_prefix, sizeof(t.log_prefix), "%s", lp->str.c_str());
  }
  NEED_FLT(mo, "root_epsilon", t.root_epsilon); NEED_FLT(mo, "root_alpha", t.root_alpha); NEED_INT(mo, "virtual_loss", t.virtual_loss);
  NEED_BOOL(alg, "use_prior", t.use_prior); NEED_FLT(alg, "c_puct", t.c_puct);
  NEED_BOOL(alg, "unexplored_q_zero", t.unexplored_q_zero); NEED_BOOL(alg, "root_unexplored_q_zero", t.root_unexplored_q_zero);
  if (jenum(ctrl, "client_type", &q.client_type) != 1) return ELFGO_E_BADARG;
  NEED_INT(ctrl, "num_game_thread_used", q.num_game_thread_used);
  NEED_FLT(ctrl, "black_resign_thres", q.black_resign_thres); NEED_FLT(ctrl, "white_resign_thres", q.white_resign_thres);
  NEED_FLT(ctrl, "never_resign_prob", q.never_resign_prob);
  const bool is_selfplay = q.black_ver >= 0 && q.white_ver == -1;           // ModelPair::is_selfplay
  {
    const int r = jbool(ctrl, "player_swap", &b);
    if (r < 0 || (r == 0 && !is_selfplay)) return ELFGO_E_BADARG;            // JSON_LOAD, not _OPTIONAL, for evaluation requests
    if (r == 1) q.player_swap = b ? 1 : 0;
  }
  {
    const int r = jbool(ctrl, "async", &b);
    if (r < 0) return ELFGO_E_BADARG;
    if (r == 1) q.async = b ? 1 : 0;
  }
#undef NEED_INT
#undef NEED_FLT
#undef NEED_BOOL
  *request = q;
  if (seq) *seq = sq;
  if (mcts_opt) *mcts_opt = t;
  return 0;
}

// The SGF file of a finished game from its Record text (what elfsp_pop_record returns): name = <prefix>_<thread_id>_<seq>_<B|W>.sgf
// (GoStateExt::dumpSgf(), go_state_ext.h:48-56), text = elfrec_game_sgf.  Returns the text length; ELFGO_E_BADARG for a text that
// is not a Record.
int64_t elfrec_record_to_sgf(const ElfSpOptions* opt, const char* record_json, const char* prefix, char* name_out, size_t name_cap,
                             char* out, size_t cap) {
  if (!opt || !record_json || !prefix) return ELFGO_E_BADARG;
  JReader rd{record_json, record_json + strlen(record_json)};
  const JValue root = rd.value();
  if (!rd.ok || root.kind != JValue::OBJ) return ELFGO_E_BADARG;
  const JValue* res = root.get("result");
  const JValue* content = res ? res->get("content") : nullptr;
  const JValue* vals = res ? res->get("values") : nullptr;
  double reward = 0;
  int64_t seq = 0, thread_id = 0;
  if (!content || content->kind != JValue::STR || !jnum(res, "reward", &reward) || !jint(&root, "seq", &seq) || !jint(&root, "thread_id", &thread_id))
    return ELFGO_E_BADARG;
  const int nm = elfrec_sgfstr_to_coords(opt->board_size, content->str.c_str(), nullptr, 0);
  if (nm < 0) return nm;
  std::vector<uint16_t> moves((size_t)nm + 1);
  elfrec_sgfstr_to_coords(opt->board_size, content->str.c_str(), moves.data(), nm);
  std::vector<float> values;
  if (vals && vals->kind == JValue::ARR) for (const JValue& v : vals->arr) values.push_back((float)v.num);
  const std::string name = std::string(prefix) + "_" + std::to_string(thread_id) + "_" + std::to_string(seq) + "_" + ((float)reward > 0 ? "B" : "W") + ".sgf";
  if (name_out) {
    if (name_cap <= name.size()) return ELFGO_E_BADSIZE;
    memcpy(name_out, name.data(), name.size() + 1);
  }
  return elfrec_game_sgf(opt, moves.data(), nm, values.data(), (int)values.size(), (float)reward, name.c_str(), nullptr, nullptr, out, cap);
}

// MsgRequestSeq::dumpJsonString (record.h:167-171): {"request":{...},"seq":n} -- what the reference's server writes
int64_t elfrec_request_seq_to_json(const ElfSpRequest* request, const ElfTsOptions* t, int64_t seq, char* out, size_t cap) {
  if (!request || !t) return ELFGO_E_BADARG;
  SpRecordMeta m{};
  m.board_size = 0;
  m.black_ver = request->black_ver; m.white_ver = request->white_ver;
  elfrec_meta_set_ts(&m, *t);
  m.black_resign_thres = request->black_resign_thres; m.white_resign_thres = request->white_resign_thres;
  m.never_resign_prob = request->never_resign_prob; m.num_game_thread_used = request->num_game_thread_used;
  m.player_swap = request->player_swap != 0; m.async = request->async != 0;
  m.client_type = request->client_type;          // as given (a parsed request is written back unchanged)
  std::string o = "{\"request\":";
  put_request(o, m);
  o += ",\"seq\":" + std::to_string(seq) + "}";
  if (!out) return (int64_t)o.size();
  if (cap <= o.size()) return ELFGO_E_BADSIZE;
  memcpy(out, o.data(), o.size());
  out[o.size()] = 0;
  return (int64_t)o.size();
}

}  // extern "C"
