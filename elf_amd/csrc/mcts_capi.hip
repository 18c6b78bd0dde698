// C ABI of the device-resident MCTS (include/elf_amd.h, elfmcts_*): host side of mcts.cuh.
#include <math.h>
#include <stddef.h>

#include <new>
#include <vector>

#include "engine_host.h"
#include "mcts.cuh"

struct ElfMcts {
  ElfGoEngine* eng = nullptr;
  int G = 0, C = 0, W = 0, NE = 0;       // C = node ids of the context = Cs + Cb
  int Cs = 0, Cb = 0;                    // small / big node records of the context, shared by its games (mcts.cuh)
  int npg = 0;                           // nodes_per_game the context was created with (Cs = G x npg)
  int SC = 0;                            // stash capacity per game
  void* small = nullptr;
  void* big = nullptr;
  int* gstack = nullptr;
  int* gstack_big = nullptr;
  PoolTops* tops = nullptr;
  int* free_stack = nullptr;
  int* parent_of = nullptr;
  int* owner = nullptr;
  unsigned char* keep = nullptr;
  int* adv = nullptr;
  int* live_tmp = nullptr;
  GameState* gs = nullptr;
  LeafRec* leaves = nullptr;
  unsigned char* d4buf = nullptr;
  int* rng_pos_t = nullptr;              // [G][NT]
  uint32_t* pathbuf = nullptr;           // [G][KTA][MCTS_PATH_LV][2]
  int KTA = 0;                           // leaves per game and step the path buffer is laid out for
  int NT = 1;                            // search threads the D4 windows are laid out for
  double* sqrt_tab = nullptr;
  int sqrt_n = 0;
  RowRec* rowmap = nullptr;
  int32_t* last_counts = nullptr;   // counts pointer of the last elfmcts_select (device), for elfmcts_expand(n_rows < 0)
  const unsigned char* mask = nullptr;   // elfmcts_set_game_mask (device bytes, GM_*), nullptr = every game searches
  const long long* req_ver = nullptr;    // elfmcts_set_required_versions (device int64 per game), nullptr = ElfMctsOptions.required_version
  TreeCfg cfg;
  size_t small_bytes = 0, big_bytes = 0; // bytes of one small / big record
  int feat_fmt = ELFGO_FEAT_F32_NCHW;
};

template <int N>
static TreePool<N> tree_of(const ElfMcts* m) {
  TreePool<N> t;
  t.small = reinterpret_cast<char*>(m->small);
  t.big = reinterpret_cast<char*>(m->big);
  t.gstack = m->gstack;
  t.gstack_big = m->gstack_big;
  t.tops = m->tops;
  t.free_stack = m->free_stack;
  t.parent_of = m->parent_of;
  t.owner = m->owner;
  t.keep = m->keep;
  t.adv = m->adv;
  t.live_tmp = m->live_tmp;
  t.SC = m->SC;
  t.gs = m->gs;
  t.leaves = m->leaves;
  t.d4buf = m->d4buf;
  t.rng_pos_t = m->rng_pos_t;
  t.pathbuf = m->pathbuf;
  t.KTA = m->KTA;
  t.NT = m->NT;
  t.sqrt_tab = m->sqrt_tab;
  t.sqrt_n = m->sqrt_n;
  t.mask = m->mask;
  t.req_ver = m->req_ver;
  t.Cs = m->Cs; t.Cb = m->Cb; t.C = m->C; t.W = m->W; t.G = m->G;
  return t;
}

// big records per `nodes` small ones: a node moves to a big record when its (TCS+1)-th edge is followed, so a big node has >= TCS + 1
// children that are nodes themselves (mcts.cuh): nodes / TCS + 1 big records cannot run out before the small ones do
static_assert(NodeL<19>::TCS == NodeL<9>::TCS, "one touched-edge capacity for both board sizes");
static int64_t big_records_for(int64_t nodes) { return nodes / NodeL<19>::TCS + 1; }
static void record_bytes(int n, size_t* small, size_t* big) {
  *small = n == 19 ? NodeL<19>::SMALL : NodeL<9>::SMALL;
  *big = n == 19 ? NodeL<19>::BIG : NodeL<9>::BIG;
}

static int cfg_from(const ElfMctsOptions* o, TreeCfg* c) {
  // leaf table of a step: num_threads x num_rollouts_per_batch <= MCTS_KMAX = 1024 (larger products are rejected, not truncated)
  if (!o || o->num_rollouts_per_batch <= 0 || o->num_threads <= 0 ||
      (int64_t)o->num_rollouts_per_batch * o->num_threads > MCTS_KMAX)
    return ELFGO_E_BADARG;
  c->rollouts_per_batch = o->num_rollouts_per_batch;
  c->virtual_loss = o->virtual_loss;
  c->use_prior = o->use_prior;
  c->unexplored_q_zero = o->unexplored_q_zero;
  c->root_unexplored_q_zero = o->root_unexplored_q_zero;
  c->c_puct = o->c_puct;
  c->komi = o->komi;
  c->ply_pass_enabled = o->ply_pass_enabled;
  c->remove_pass_if_dangerous = o->remove_pass_if_dangerous;
  c->rotation_flip = o->rotation_flip;
  c->num_threads = o->num_threads;
  c->required_version = o->required_version;
  if (!(c->c_puct >= 0.0f)) return ELFGO_E_BADARG;   // the scoring order of never-followed edges relies on U growing with the prior
  return 0;
}

extern "C" {

int elfmcts_create(ElfGoEngine* e, int num_games, int nodes_per_game, int d4_window, const ElfMctsOptions* opt, ElfMcts** out) {
  if (!e || !out || num_games <= 0 || num_games > e->capacity || nodes_per_game < 64 || (nodes_per_game & 63) || d4_window <= 0)
    return ELFGO_E_BADARG;
  // one window per search thread: d4_window = num_threads x (draws one thread can make per move)
  if (!opt || opt->num_threads <= 0 || d4_window % opt->num_threads) return ELFGO_E_BADARG;
  ElfMcts* m = new (std::nothrow) ElfMcts();
  if (!m) return ELFGO_E_NOMEM;
  int rc = cfg_from(opt, &m->cfg);
  if (rc) { delete m; return rc; }
  DevGuard _dg(e->device);
  m->eng = e; m->G = num_games; m->W = d4_window; m->NT = opt->num_threads; m->npg = nodes_per_game;
  // ONE pool for the context: num_games x nodes_per_game small records (+ their share of big ones), any game may take any of them
  const int64_t cs = (int64_t)num_games * nodes_per_game, cb = big_records_for(cs);
  if (cs + cb >= ((int64_t)1 << 31) - 64) { delete m; return ELFGO_E_BADARG; }     // node ids are 32-bit
  m->Cs = (int)cs; m->Cb = (int)cb; m->C = m->Cs + m->Cb;
  m->KTA = opt->num_threads * opt->num_rollouts_per_batch;
  m->SC = (MCTS_REFILL + 1) * m->KTA;
  record_bytes(e->n, &m->small_bytes, &m->big_bytes);
  m->NE = e->n == 19 ? NodeL<19>::NE : NodeL<9>::NE;
  const size_t G = num_games, C = m->C;
#define MCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { elfmcts_destroy(m); return (int)_e; } } while (0)
  MCHK(hipMalloc(&m->small, (size_t)m->Cs * m->small_bytes));
  MCHK(hipMalloc(&m->big, (size_t)m->Cb * m->big_bytes));
  MCHK(hipMalloc((void**)&m->gstack, (size_t)m->Cs * sizeof(int)));
  MCHK(hipMalloc((void**)&m->gstack_big, (size_t)m->Cb * sizeof(int)));
  MCHK(hipMalloc((void**)&m->tops, sizeof(PoolTops)));
  MCHK(hipMalloc((void**)&m->free_stack, G * m->SC * sizeof(int)));
  MCHK(hipMalloc((void**)&m->parent_of, C * sizeof(int)));
  MCHK(hipMalloc((void**)&m->owner, C * sizeof(int)));
  MCHK(hipMalloc((void**)&m->keep, C));
  MCHK(hipMalloc((void**)&m->adv, G * sizeof(int)));
  MCHK(hipMalloc((void**)&m->live_tmp, G * sizeof(int)));
  MCHK(hipMalloc((void**)&m->gs, G * sizeof(GameState)));
  MCHK(hipMemset(m->gs, 0, G * sizeof(GameState)));
  MCHK(hipMalloc((void**)&m->leaves, G * MCTS_KMAX * sizeof(LeafRec)));
  MCHK(hipMalloc((void**)&m->d4buf, G * (size_t)d4_window));
  MCHK(hipMemset(m->d4buf, 0, G * (size_t)d4_window));
  MCHK(hipMalloc((void**)&m->pathbuf, G * (size_t)m->KTA * MCTS_PATH_LV * 2 * sizeof(uint32_t)));
  MCHK(hipMalloc((void**)&m->rng_pos_t, G * (size_t)m->NT * sizeof(int)));
  MCHK(hipMemset(m->rng_pos_t, 0, G * (size_t)m->NT * sizeof(int)));
  MCHK(hipMalloc((void**)&m->rowmap, G * MCTS_KMAX * sizeof(RowRec)));
  // std::sqrt(int) of the reference (tree_search_base.h:153) tabulated with the host libm
  m->sqrt_n = 1 << 17;
  {
    std::vector<double> t(m->sqrt_n);
    for (int i = 0; i < m->sqrt_n; ++i) t[i] = std::sqrt((double)i);
    MCHK(hipMalloc((void**)&m->sqrt_tab, t.size() * sizeof(double)));
    MCHK(hipMemcpy(m->sqrt_tab, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  {
    size_t most = C > G ? C : G;
    DISPATCH(e, {
      hipLaunchKernelGGL(k_mcts_pool_init<N>, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, (hipStream_t)0, tree_of<N>(m));
      hipLaunchKernelGGL(k_mcts_advance_reroot<N>, dim3(m->G), dim3(64), 0, (hipStream_t)0, tree_of<N>(m));   // adv = -1: a root for every game
    });
    MCHK(hipGetLastError());
  }
  MCHK(hipDeviceSynchronize());
#undef MCHK
  *out = m;
  return 0;
}

int elfmcts_destroy(ElfMcts* m) {
  if (!m) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  if (m->small) (void)hipFree(m->small);
  if (m->big) (void)hipFree(m->big);
  if (m->gstack) (void)hipFree(m->gstack);
  if (m->gstack_big) (void)hipFree(m->gstack_big);
  if (m->tops) (void)hipFree(m->tops);
  if (m->free_stack) (void)hipFree(m->free_stack);
  if (m->parent_of) (void)hipFree(m->parent_of);
  if (m->owner) (void)hipFree(m->owner);
  if (m->keep) (void)hipFree(m->keep);
  if (m->adv) (void)hipFree(m->adv);
  if (m->live_tmp) (void)hipFree(m->live_tmp);
  if (m->gs) (void)hipFree(m->gs);
  if (m->leaves) (void)hipFree(m->leaves);
  if (m->d4buf) (void)hipFree(m->d4buf);
  if (m->rng_pos_t) (void)hipFree(m->rng_pos_t);
  if (m->pathbuf) (void)hipFree(m->pathbuf);
  if (m->sqrt_tab) (void)hipFree(m->sqrt_tab);
  if (m->rowmap) (void)hipFree(m->rowmap);
  delete m;
  return 0;
}

int elfmcts_set_options(ElfMcts* m, const ElfMctsOptions* opt) {
  if (!m || !opt) return ELFGO_E_BADARG;
  if (opt->num_threads != m->NT) return ELFGO_E_BADARG;   // the D4 windows are laid out per search thread at creation
  if ((int64_t)opt->num_threads * opt->num_rollouts_per_batch > m->KTA) return ELFGO_E_BADARG;   // ... and the path rows per leaf of a step
  return cfg_from(opt, &m->cfg);
}
int elfmcts_num_threads(const ElfMcts* m) { return m ? m->NT : ELFGO_E_BADARG; }
int elfmcts_thread_draws(ElfMcts* m, int32_t* out_host, void* stream) {
  if (!m || !out_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  HIPCHK(hipMemcpyAsync(out_host, m->rng_pos_t, sizeof(int32_t) * (size_t)m->G * m->NT, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int elfmcts_set_feature_format(ElfMcts* m, int fmt) {
  if (!m || (fmt != ELFGO_FEAT_F32_NCHW && fmt != ELFGO_FEAT_F16_NHWC)) return ELFGO_E_BADARG;
  m->feat_fmt = fmt;
  return 0;
}
int elfmcts_get_feature_format(const ElfMcts* m, int* fmt) {
  if (!m || !fmt) return ELFGO_E_BADARG;
  *fmt = m->feat_fmt;
  return 0;
}
int elfmcts_set_game_mask(ElfMcts* m, const uint8_t* mask) {
  if (!m) return ELFGO_E_BADARG;
  m->mask = mask;
  return 0;
}
int elfmcts_set_required_versions(ElfMcts* m, const int64_t* versions) {
  if (!m) return ELFGO_E_BADARG;
  m->req_ver = reinterpret_cast<const long long*>(versions);
  return 0;
}
int elfmcts_max_rollouts_per_step(void) { return MCTS_KMAX; }
int elfmcts_num_games(const ElfMcts* m) { return m ? m->G : ELFGO_E_BADARG; }
int elfmcts_edge_stride(const ElfMcts* m) { return m ? m->NE : ELFGO_E_BADARG; }
size_t elfmcts_tree_bytes_per_game(int board_size, int nodes_per_game) {
  return elfmcts_tree_bytes_per_game2(board_size, nodes_per_game, 1, 16, 1024);
}
size_t elfmcts_tree_bytes_per_game2(int board_size, int nodes_per_game, int num_threads, int rollouts_per_batch, int d4_window) {
  if ((board_size != 19 && board_size != 9) || nodes_per_game <= 0 || num_threads <= 0 || rollouts_per_batch <= 0 || d4_window < 0) return 0;
  size_t sm, bg;
  record_bytes(board_size, &sm, &bg);
  const size_t Cs = nodes_per_game, kta = (size_t)num_threads * rollouts_per_batch;
  // a game's share of the context's pool: records (the big class at 1 per TCS small ones), the per-id arrays (free-stack entry, parent,
  // owner, keep byte), and what IS per game: its stash, leaf / row tables, path rows of a step, D4 windows
  return Cs * sm + (Cs * bg + NodeL<19>::TCS - 1) / NodeL<19>::TCS + (Cs + Cs / NodeL<19>::TCS + 1) * (sizeof(int) * 3 + 1) +
         sizeof(GameState) + MCTS_KMAX * (sizeof(LeafRec) + sizeof(RowRec)) + kta * MCTS_PATH_LV * 2 * sizeof(uint32_t) +
         (MCTS_REFILL + 1) * kta * sizeof(int) + (size_t)d4_window + (size_t)num_threads * sizeof(int) + 2 * sizeof(int);
}
/* average bytes per node id of the pool: small record + its share of the big class and of the id arrays */
size_t elfmcts_node_bytes(const ElfMcts* m) {
  return m ? (elfmcts_tree_bytes_per_game2(m->eng->n, m->npg, m->NT, m->KTA / m->NT, m->W) + m->npg - 1) / m->npg : 0;
}

/* Node ids of the context: out[0] small records in all, [1] of them free (on the context's stack or in a game's stash), [2] big records
 * in all, [3] free, [4] the largest number of ids one game has held since the last reset, [5] the largest sum over games ... (see header) */
int elfmcts_pool_info(ElfMcts* m, int64_t* out8_host, int reset_peaks) {
  if (!m || !out8_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  HIPCHK(hipDeviceSynchronize());
  PoolTops t;
  HIPCHK(hipMemcpy(&t, m->tops, sizeof(t), hipMemcpyDeviceToHost));
  std::vector<GameState> gs(m->G);
  HIPCHK(hipMemcpy(gs.data(), m->gs, sizeof(GameState) * m->G, hipMemcpyDeviceToHost));
  int64_t stash = 0, live = 0, live_max = 0, peak_max = 0, peak_sum = 0;
  for (auto& g : gs) {
    stash += g.free_top; live += g.live;
    if (g.live > live_max) live_max = g.live;
    if (g.live_peak > peak_max) peak_max = g.live_peak;
    peak_sum += g.live_peak;
  }
  out8_host[0] = m->Cs; out8_host[1] = (int64_t)t.small + stash; out8_host[2] = m->Cb; out8_host[3] = t.big;
  out8_host[4] = live; out8_host[5] = live_max; out8_host[6] = peak_max; out8_host[7] = peak_sum;
  if (reset_peaks) {
    for (auto& g : gs) g.live_peak = g.live;
    // only the peak field is written back (the games may not be touched concurrently: the device was synchronised above)
    for (int i = 0; i < m->G; ++i)
      HIPCHK(hipMemcpy(reinterpret_cast<char*>(m->gs + i) + offsetof(GameState, live_peak), &gs[i].live_peak, sizeof(int), hipMemcpyHostToDevice));
  }
  return 0;
}

/* test / debug service: node ids per game counted by a scan of the owner array (what RootInfo word 7 tracks incrementally) */
int elfmcts_count_live(ElfMcts* m, int32_t* out_host) {
  if (!m || !out_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  HIPCHK(hipDeviceSynchronize());
  int32_t* d = nullptr;
  HIPCHK(hipMalloc(&d, m->G * sizeof(int32_t)));
  HIPCHK(hipMemset(d, 0, m->G * sizeof(int32_t)));
  unsigned blocks = (unsigned)(((size_t)m->C + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_count_live<N>, dim3(blocks), dim3(256), 0, (hipStream_t)0, tree_of<N>(m), d));
  hipError_t e = hipMemcpy(out_host, d, m->G * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIPCHK(e);
  return 0;
}

static int sweep_launches(ElfMcts* m, hipStream_t st) {
  // enough waves to fill the chip; each thread covers >= 1 id
  unsigned blocks = (unsigned)(((size_t)m->C + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  DISPATCH(m->eng, {
    hipLaunchKernelGGL(k_mcts_advance_mark<N>, dim3(blocks), dim3(256), 0, st, tree_of<N>(m));
    hipLaunchKernelGGL(k_mcts_advance_sweep<N>, dim3(blocks), dim3(256), 0, st, tree_of<N>(m));
    hipLaunchKernelGGL(k_mcts_advance_reroot<N>, dim3(m->G), dim3(64), 0, st, tree_of<N>(m));
  });
  HIPCHK(hipGetLastError());
  return 0;
}

int elfmcts_clear(ElfMcts* m, const int32_t* games, int n, void* stream) {
  if (!m || n < 0 || n > m->G) return ELFGO_E_BADARG;
  if (n == 0) return 0;
  if (!games && n != m->G) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_clear_mark<N>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tree_of<N>(m), games, n));
  HIPCHK(hipGetLastError());
  return sweep_launches(m, (hipStream_t)stream);
}

int elfmcts_set_root(ElfMcts* m, const int32_t* board_ids, void* stream) {
  if (!m) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  DISPATCH(m->eng, {
    hipLaunchKernelGGL((k_mcts_set_root<N, Pool<N>>), dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), pool_of<N>(m->eng),
                       board_ids);
  });
  HIPCHK(hipGetLastError());
  return 0;
}

int elfmcts_set_d4(ElfMcts* m, const uint8_t* d4_host, void* stream) {
  if (!m || !d4_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  HIPCHK(hipMemcpyAsync(m->d4buf, d4_host, (size_t)m->G * m->W, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}

int elfmcts_dirichlet(ElfMcts* m, const float* etas, const float* Z, float epsilon, void* stream) {
  if (!m || !etas || !Z) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_dirichlet<N>, dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), etas, Z, epsilon));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfmcts_select(ElfMcts* m, const int32_t* board_ids, void* s_dst, int64_t stride_elems, int32_t* counts, void* stream) {
  if (!m || !s_dst || !counts || stride_elems < (int64_t)18 * m->eng->n * m->eng->n) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  const int KT = m->cfg.rollouts_per_batch * m->cfg.num_threads;
  DISPATCH(m->eng, {
    hipLaunchKernelGGL((k_mcts_select<N, Pool<N>>), dim3(m->G), dim3(64), (size_t)20 * ((KT + 63) & ~63), (hipStream_t)stream, tree_of<N>(m),
                       pool_of<N>(m->eng), board_ids, m->cfg);
    hipLaunchKernelGGL((k_mcts_leafstate<N, Pool<N>>), dim3(m->G * KT), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), pool_of<N>(m->eng),
                       board_ids, KT, m->cfg);
    hipLaunchKernelGGL(k_mcts_leafindex<N>, dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), m->cfg);
    hipLaunchKernelGGL(k_mcts_rowbase<N>, dim3(1), dim3(1024), 0, (hipStream_t)stream, tree_of<N>(m), counts);
    hipLaunchKernelGGL(k_mcts_features<N>, dim3(m->G * KT), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), KT, s_dst, stride_elems,
                       m->feat_fmt, m->rowmap);
  });
  HIPCHK(hipGetLastError());
  m->last_counts = counts;
  return 0;
}

int elfmcts_expand(ElfMcts* m, const float* pi, int64_t pi_stride_floats, const float* value, const int64_t* rv, int n_rows, void* stream) {
  if (!m || n_rows > m->G * MCTS_KMAX) return ELFGO_E_BADARG;
  if (n_rows < 0 && !m->last_counts) return ELFGO_E_BADARG;
  if (n_rows != 0 && (!pi || !value || pi_stride_floats < (int64_t)m->eng->n * m->eng->n + 1)) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  const int grid = n_rows >= 0 ? n_rows : m->G * m->cfg.rollouts_per_batch * m->cfg.num_threads;
  DISPATCH(m->eng, {
    if (grid > 0)
      hipLaunchKernelGGL(k_mcts_expand<N>, dim3(grid), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), (const RowRec*)m->rowmap, pi, pi_stride_floats, value, rv, n_rows, (const int32_t*)m->last_counts, m->cfg);
    hipLaunchKernelGGL(k_mcts_backup<N>, dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), m->cfg);
  });
  HIPCHK(hipGetLastError());
  return 0;
}

#ifdef ELF_PROFILE_SELECT
// profile builds only: accumulated s_memtime ticks per k_mcts_select phase (see SEL_PHASE in mcts.cuh)
extern "C" int elfprof_select_phases(unsigned long long* out8) {
  HIPCHK(hipDeviceSynchronize());
  std::vector<unsigned long long> all((size_t)4096 * 8);
  HIPCHK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(elfgo::g_select_phase), all.size() * sizeof(unsigned long long)));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (size_t i = 0; i < all.size(); ++i) out8[i & 7] += all[i];
  return 0;
}
#endif

#ifdef ELF_PROFILE_SELECT
// profile builds only: per step {sum, max of the waves' ticks, waves, max visited nodes of a wave}
extern "C" int elfprof_select_steps(unsigned long long* out4096) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out4096, HIP_SYMBOL(elfgo::g_select_step), 4096 * sizeof(unsigned long long)));
  return 0;
}
#endif

#ifdef ELF_PROFILE_EXPAND
// profile builds only: accumulated s_memtime ticks per k_mcts_expand phase (see EXP_PHASE in mcts.cuh)
extern "C" int elfprof_expand_phases(unsigned long long* out8) {
  HIPCHK(hipDeviceSynchronize());
  std::vector<unsigned long long> all((size_t)65536 * 8);
  HIPCHK(hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(elfgo::g_expand_phase), all.size() * sizeof(unsigned long long)));
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  for (size_t i = 0; i < all.size(); ++i) out8[i & 7] += all[i];
  return 0;
}
extern "C" int elfprof_expand_rowmax(unsigned long long* out65536) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out65536, HIP_SYMBOL(elfgo::g_expand_rowmax), 65536 * sizeof(unsigned long long)));
  return 0;
}
#endif

int elfmcts_root(ElfMcts* m, int32_t* info, int32_t* coord, int32_t* visits, float* prior, float* reward, int32_t* child, void* stream) {
  if (!m || !info) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  static_assert(sizeof(RootInfo) == ELFMCTS_ROOT_WORDS * 4, "RootInfo layout");
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_root<N>, dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), (RootInfo*)info, coord,
                                      visits, prior, reward, child));
  HIPCHK(hipGetLastError());
  return 0;
}

int elfmcts_validate(ElfMcts* m, int32_t* out5_host) {
  if (!m || !out5_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  HIPCHK(hipDeviceSynchronize());
  int32_t* d = nullptr;
  HIPCHK(hipMalloc(&d, 5 * sizeof(int32_t)));
  HIPCHK(hipMemset(d, 0, 5 * sizeof(int32_t)));
  int gx = m->C < 8192 ? m->C : 8192;
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_validate<N>, dim3(gx), dim3(64), 0, (hipStream_t)0, tree_of<N>(m), d));
  hipError_t e = hipMemcpy(out5_host, d, 5 * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIPCHK(e);
  return 0;
}

int elfmcts_node_visits(ElfMcts* m, int64_t* out_host) {
  if (!m || !out_host) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  std::vector<GameState> gs(m->G);
  HIPCHK(hipMemcpy(gs.data(), m->gs, sizeof(GameState) * m->G, hipMemcpyDeviceToHost));
  long long t = 0;
  for (auto& g : gs) t += g.node_visits;
  *out_host = t;
  return 0;
}

int elfmcts_advance(ElfMcts* m, const int32_t* moves, void* stream) {
  if (!m || !moves) return ELFGO_E_BADARG;
  DevGuard _dg(m->eng->device);
  DISPATCH(m->eng, hipLaunchKernelGGL(k_mcts_advance_prepare<N>, dim3(m->G), dim3(64), 0, (hipStream_t)stream, tree_of<N>(m), moves));
  HIPCHK(hipGetLastError());
  return sweep_launches(m, (hipStream_t)stream);
}

}  // extern "C"
