// Host side of the C-ABI declared in include/crbm_amd.h: owns device memory,
// the HIP stream and (optionally) the RCCL communicator of one GPU, and turns
// each entry point into kernel launches.  No torch, no Python types.
#define CRBM_DEFINE_MISC_KERNELS
#include "crbm_kernels.h"
#include "crbm_jit.h"
#include "../../include/crbm_amd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace crbm;

namespace {

std::string g_create_error;

// ---- RCCL, loaded on demand so that the library itself has no link-time
// dependency on it (single-GPU users never touch it) --------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
RcclApi g_rccl;

bool load_rccl() {
  if (g_rccl.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    g_rccl.error = std::string("cannot dlopen librccl: ") + dlerror();
    return false;
  }
#define CRBM_SYM(field, name)                                                    \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name)); \
  if (!g_rccl.field) {                                                           \
    g_rccl.error = std::string("librccl lacks ") + name;                         \
    dlclose(g_rccl.lib);                                                         \
    g_rccl.lib = nullptr;                                                        \
    return false;                                                                \
  }
  CRBM_SYM(GetUniqueId, "ncclGetUniqueId")
  CRBM_SYM(CommInitRank, "ncclCommInitRank")
  CRBM_SYM(CommDestroy, "ncclCommDestroy")
  CRBM_SYM(AllReduce, "ncclAllReduce")
  CRBM_SYM(Broadcast, "ncclBroadcast")
  CRBM_SYM(GetErrorString, "ncclGetErrorString")
#undef CRBM_SYM
  return true;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// Letter grouping of PLAIN chain launches (crbm_gibbs_steps*: one 1024-thread block per CU whose LDS is otherwise idle):
// the largest grouping whose table stays within 48 KB -- config #2: G = 4 (49 152 B, four gathers per position) where the
// fused training launch, four blocks per CU beside the statistics slices, takes G = 3 (five gathers): 21.7 -> 21.4 us per
// launch.  Only for models whose training step never launches the plain kernel (Cfg::FUSE_STATS) -- the others would
// rebuild the second table every step -- and not beyond 48 KB: every block copies its table once per launch (config #5
// with G = 4, 82 KB: 144.6 -> 150 us).  CRBM_GROUP_SOLO overrides.
int solo_group(int K, int M, int ds, int G, int pool) {
  const int forced = env_int("CRBM_GROUP_SOLO", 0);
  if (forced >= 1 && forced <= 4) return forced;
  if (!model_shape(K, M, ds, G, pool).FUSE_STATS) return G;
  return std::max(G, choose_group(K, M, ds, 48 * 1024));
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;   // elements
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

// One host thread per extra partition of the plain chain launches (crbm_handle::chain_parts): it enqueues that
// partition's launches on the partition's stream while the caller's thread enqueues partition 0's.  Alternating two
// streams from ONE thread costs ~6 us of host time per launch (2.6 us when a thread stays on one stream), which at two
// launches per 17.6-us step leaves the GPU waiting for the host in short bursts.
struct PartJob {
  hipFunction_t fn;
  crbm::GibbsArgs args;
  unsigned grid, threads, lds;
  unsigned long long delay_ticks;   // > 0: a delay kernel in front (the partition starts this much later)
};
struct PartWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, idle_cv;
  std::deque<PartJob> jobs;
  bool stop = false, busy = false;
  hipError_t error = hipSuccess;
  int device = 0;
  hipStream_t stream = nullptr;
};

struct crbm_handle {
  crbm_config cfg;
  int K = 0, M = 0, ds = 0, NW = 0, G = 0, KAM = 0;
  int A = 4;                // letters of the alphabet (input_dims); anything but 4 runs on the generic kernels, rows of bytes
  int Lf = 0, Lv = 0, B = 0;
  int device = 0, num_cu = 256;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;          // model phase of a training step runs beside the data phase
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
  bool overlap = false;
  bool big = false;         // model beyond the LDS-resident kernels: the generic "big" kernels (crbm_kernels.h) serve every entry point
  ModelShape ms;
  ModelShape ms_solo;       // the same model with the letter grouping of plain chain launches (GS; == ms when GS == G)
  int GS = 0;
  JitKernels jk;            // kernels specialised for this model (hiprtc)
  float* d_tables = nullptr;   // precomputed LDS images (gather / top-down tables, c)
  float* d_tf_solo = nullptr;  // gather table in the solo grouping (GS != G only), rebuilt lazily before a plain chain launch
  bool tables_dirty = true;
  uint64_t params_version = 1, tf_solo_version = 0;   // bumped with every change of W, b, c / version d_tf_solo was built from
  // parameters and optimiser state
  float *dW = nullptr, *db = nullptr, *dc = nullptr, *dvW = nullptr, *dvb = nullptr, *dvc = nullptr;
  // the update writes into a second set of buffers; the two sets swap after every step (launch_update)
  float *dW2 = nullptr, *db2 = nullptr, *dc2 = nullptr, *dvW2 = nullptr, *dvb2 = nullptr, *dvc2 = nullptr;
  // persistent chains: K-bit masks per hidden position, letters of the last visible sample
  uint32_t *d_hm = nullptr, *d_hmp = nullptr, *d_vf = nullptr;
  uint32_t* d_flags = nullptr;
  unsigned long long* d_ones = nullptr;
  DevBuf<float> stage, stage2, out_a, out_b, out_c;
  DevBuf<uint32_t> letters, letters2, masks_tmp;   // (…2: second set of the double-buffered host-input sweeps)
  DevBuf<float> out_a2, out_b2;
  DevBuf<uint32_t> dataset[CRBM_DATASET_SLOTS];   // resident data sets (slot 0: training, slot 1: test by convention)
  DevBuf<float> partials, partials2;
  DevBuf<unsigned long long> eval_ones;            // sampled ones per mini-batch (crbm_eval_epoch_resident)
  float* d_sums = nullptr;
  int dataset_n[CRBM_DATASET_SLOTS] = {0, 0}, dataset_L[CRBM_DATASET_SLOTS] = {0, 0};
  int slot = 0;
  // sampler
  uint64_t seed = 0;
  uint32_t gibbs_step = 0, eval_step = 0, chain_offset = 0;
  // launch geometry
  GibbsLayout gl;                      // of the variant in use
  int gibbs_threads = 256, gibbs_grid = 0;
  // top-down variants of the Gibbs kernel: [0] dense tables (small models only), [1] set-bit walk
  GibbsLayout glv[2];
  GibbsLayout gl_solo;                 // geometry of plain chain launches of the sparse variant when it differs (solo_threads > 0)
  int solo_threads = 0, solo_grid = 0;
  // Plain chain launches of short kernels go out as `chain_parts` launches of `part_chains` chains each, one stream per
  // partition: chains are independent, so partition p's step t+1 only waits for partition p's step t, and the drain of
  // one partition's kernel, the dispatch and the ramp of its next one are filled by the other partition's blocks on the
  // same CUs (config #2: 20.9 -> 17.6 us per step of the whole batch).
  int chain_parts = 1, part_chains = 0;
  GibbsLayout gl_part;                 // geometry of ONE partition's launch (chain_parts > 1); solo_* / gl_solo stay the unpartitioned one,
  int part_threads = 0, part_grid = 0; // which the chain launch INSIDE a training step takes (one launch, then the statistics wait for it)
  hipStream_t part_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  PartWorker* part_worker[4] = {nullptr, nullptr, nullptr, nullptr};   // [0] stays null: the caller's thread
  hipEvent_t part_done[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_parts_fork = nullptr;
  bool forked = false;                 // the partition streams hold work the main stream has not waited for yet
  bool main_idle_hint = false;         // the main stream was idle when the caller last looked (only its own event record is pending)
  // Partitions that start together stay in lockstep for hundreds of launches -- their gaps coincide and nothing is
  // gained (config #2: 20.8 us per step over the first 20 steps, 19.3 over 200, 17.7 once they have drifted apart).
  // So partition p starts p/P of a launch late: a one-wave delay kernel in front of its first launch after a fork.
  // The length of a launch is measured on the way (events around partition 0's first launch of the previous fork).
  double part_launch_us = 0.0;         // 0: not known yet (cost model below)
  hipEvent_t ev_cal0 = nullptr, ev_cal1 = nullptr;
  bool cal_pending = false;
  int wall_khz = 100000;               // ticks of the GPU's wall clock per millisecond
  int threadsv[2] = {256, 256}, gridv[2] = {0, 0};
  bool has_dense = false;
  int gibbs_wpe = 0;                   // register-allocation hint compiled into the sparse Gibbs kernel
  int variant = 1;
  int topdown_mode = 0;                // CRBM_TOPDOWN: 1 dense, anything else the set-bit walk (fixed per handle)
  uint32_t* d_nset = nullptr;          // per wave of the last Gibbs launch: set bits of the final state
  int nset_slots = 0;
  uint32_t launches_since_read = 0;
  double activity = -1.0;              // fraction of hidden units on after the last launch that was read back
  int stats_rows = 0;
  bool fuse_stats = false;             // model half of the statistics inside the Gibbs kernel (Cfg::FUSE_STATS; CRBM_STATS=split: off)
  bool one_launch = true;              // ... and the data half in the same launch (CRBM_STATS=two: separate launch)
  SumsLayout sl;
  // data parallel
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  // ... without a collective launch: every rank's sums buffer mapped into every rank (crbm_ipc_*)
  float* ipc_buf = nullptr;            // own: [2 parities][ipc_stride floats], then 2 flag words and a status word
  void* ipc_peer[IPC_MAX_RANKS] = {};  // all ranks' buffers as this process sees them (own included)
  int ipc_stride = 0;                  // floats per parity (sums count rounded up to a 128-byte line)
  bool ipc_on = false;
  uint32_t ipc_step = 0;               // steps published so far (the flag value of the next one is ipc_step + 1)
  bool ipc_published = false;          // the column reduction of the running step has written and flagged the published buffer itself
  uint32_t* d_ticket = nullptr;        // arrival counter of that launch (zero between launches)
  unsigned long long* d_probe = nullptr;   // {wall ticks, shader cycles, scratch, scratch} summed over the launches of all crbm_time_gibbs calls
  unsigned long long probe_base[2] = {0, 0}, probe_last[2] = {0, 0};   // the sums before / after the last call
  bool probe_on = false;
  unsigned long long* d_timeline = nullptr;   // CRBM_GIBBS_TIMELINE: first / last tick of block 0 of every launch of a crbm_time_gibbs call
  int timeline_cap = 0, timeline_next = 0;
  unsigned long long ipc_timeout_ticks = 0;   // bound of an update launch's wait for its peers, in ticks of the GPU's wall clock
  // Statistics of a generic DNA model on the matrix cores, a slab of motifs at a time (slab_launch_stats): a shadow handle
  // of the slab model -- slab->K motifs, its specialised kernels, its partial rows and nothing else (the stream is this
  // handle's) -- and one table image per slab, rebuilt when the parameters have changed
  crbm_handle* slab = nullptr;
  float* d_slab_tables = nullptr;
  int slab_n = 0;
  bool slab_hgv = false;               // the chain's h|v takes the slabs too (slab a multiple of ten motifs, or one slab: sampler groups)
  uint64_t slab_tables_version = 0;
  std::string slab_note;               // why the slabs are off, if they are (crbm_launch_info prints nothing of it; a debugging aid)
  std::string err;
};

namespace {

int fail(crbm_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return fail(h, CRBM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));      \
  } while (0)

#define ARGCHK(cond, msg)                                 \
  do {                                                    \
    if (!(cond)) return fail(h, CRBM_ERR_INVALID, msg);   \
  } while (0)

RngView rng_view(const crbm_handle* h, uint32_t step, uint32_t seq_offset) {
  RngView r;
  r.seed_lo = (uint32_t)(h->seed & 0xffffffffu);
  r.seed_hi = (uint32_t)(h->seed >> 32);
  r.step = step;
  r.seq_offset = seq_offset;
  return r;
}

int tab_bytes(const crbm_handle* h) { return h->ms.TAB * 4; }

// words per packed row of L letters (2-bit letters for DNA, bytes for any other alphabet)
int lw(const crbm_handle* h, int L) { return letter_words_any(h->A, L); }

// (re)build the LDS table images after a parameter change
int ensure_tables(crbm_handle* h) {
  if (!h->tables_dirty || h->big) return CRBM_OK;
  TablesArgs t;
  t.W = h->dW; t.b = h->db; t.c = h->dc; t.out = h->d_tables;
  const unsigned grid = (unsigned)std::max(1, std::min((h->ms.TABLES_ALL + 255) / 256, h->num_cu * 4));
  HIPCHK(jit_launch(h->jk.build_tables, t, grid, 1, 256, 0, h->stream));
  h->tables_dirty = false;
  return CRBM_OK;
}

// the gather table of the solo grouping follows the parameters lazily: built when a plain chain launch finds it stale
int ensure_solo_table(crbm_handle* h) {
  if (h->GS == h->G || h->tf_solo_version == h->params_version) return CRBM_OK;
  TablesArgs t;
  t.W = h->dW; t.b = h->db; t.c = h->dc; t.out = h->d_tf_solo;
  const unsigned grid = (unsigned)std::max(1, std::min((h->ms_solo.TAB + 255) / 256, h->num_cu * 4));
  HIPCHK(jit_launch(h->jk.build_gather_solo, t, grid, 1, 256, 0, h->stream));
  h->tf_solo_version = h->params_version;
  return CRBM_OK;
}

// ---- the "big" path (crbm_kernels.h: models beyond the LDS-resident kernels) --------------------------------------
BigModel big_model(const crbm_handle* h) {
  BigModel m;
  m.W = h->dW; m.b = h->db; m.c = h->dc;
  m.K = h->K; m.M = h->M; m.ds = h->ds; m.NW = h->NW; m.A = h->A;
  return m;
}

constexpr int SLAB_FALLBACK = -1000;     // slab_launch_stats / slab_launch_hgv: nothing launched, take the generic kernel
int slab_launch_hgv(crbm_handle* h, const uint32_t* d_letters, int n, int L, int mode, unsigned long long* ones, uint32_t* masks,
                    uint32_t kind, uint32_t step, uint32_t seq_offset, hipStream_t st);
int slab_launch_stats(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool data_half, hipStream_t st, ReduceArgs* reduce);
int slab_launch_fe(crbm_handle* h, const uint32_t* rows, int n, int L, float* fe, float* fem, hipStream_t st);

// h | v on packed rows: dense outputs (API), a count of sampled ones (evaluateData) or the masks of one strand (chain)
int big_launch_hgv(crbm_handle* h, const uint32_t* d_letters, int n, int L, int mode, float* act, float* prob, float* sample,
                   unsigned long long* ones, uint32_t* masks, uint32_t kind, uint32_t step, uint32_t seq_offset, hipStream_t st) {
  if (h->slab && h->slab_hgv && (masks || ones) && !act && !prob && !sample) {
    // the chain's h|v, or the count of a sample (evaluateData, the per-epoch evaluation: into scratch masks): the
    // specialised kernel, slab by slab
    uint32_t* out = masks;
    if (!out) {
      HIPCHK(h->masks_tmp.ensure((size_t)n * (L - h->M + 1) * h->NW));
      out = h->masks_tmp.p;
    }
    const int rc = slab_launch_hgv(h, d_letters, n, L, mode, ones, out, kind, step, seq_offset, st);
    if (rc != SLAB_FALLBACK) return rc;      // (a fallback comes before anything is launched)
  }
  BigHgvArgs a;
  a.m = big_model(h);
  a.letters = d_letters;
  a.n = n; a.L = L; a.Lh = L - h->M + 1; a.LW = lw(h, L);
  const int pool = h->ms.POOL;
  int ks = pool > 1 ? 8 : 32;                                // motifs per staged slab: a divisor of 32 whose filters fit 96 KB
  while (ks > 1 && (size_t)big_hgv_ksp(ks) * h->M * h->A * 4 > 96 * 1024) ks >>= 1;   // (pooled: at most 8, the capacity of big_hgv_pooled_kernel)
  a.KS = ks;
  a.pool = pool;
  // rows per tile: enough tiles to cover the chip a few times over (every tile stages the filters once per slab), at most
  // what the LDS takes of packed letters and ~4096 positions
  a.TS = std::max(1, std::min(std::min((n + 4 * h->num_cu - 1) / (4 * h->num_cu), (32 * 1024) / (a.LW * 4)), std::max(1, 4096 / a.Lh)));
  a.mode = mode;
  a.act = act; a.prob = prob; a.sample = sample; a.ones = ones; a.masks = masks;
  a.rng = rng_view(h, step, seq_offset);
  a.kind = kind;
  const size_t lds = ((size_t)big_hgv_ksp(ks) * h->M * h->A + 32) * 4 + (size_t)a.TS * a.LW * 4;
  ARGCHK(lds <= 160 * 1024, "motif_length too large for the h|v kernel");
  const int ntiles = (n + a.TS - 1) / a.TS;
  // few tiles (small batches): one block per (tile, mask word) -- every slab lies inside one mask word, so blocks of
  // different words never touch the same output
  a.split = (h->NW > 1 && ntiles < 4 * h->num_cu) ? 1 : 0;
  const dim3 grid(std::max(1, std::min(ntiles, h->num_cu * 8)), a.split ? h->NW : 1);
  if (pool > 1) hipLaunchKernelGGL(big_hgv_pooled_kernel, grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL(big_hgv_kernel, grid, dim3(256), lds, st, a);
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}

// `steps` Gibbs steps of the persistent chains: v | h from the masks, then h | v per strand into the masks
int big_launch_gibbs(crbm_handle* h, int steps, hipStream_t st) {
  BigVghArgs v;
  v.m = big_model(h);
  v.hm = h->d_hm; v.hmp = h->ds ? h->d_hmp : nullptr; v.vout = h->d_vf;
  v.nchains = h->B; v.Lf = h->Lf; v.Lv = h->Lv; v.LWs = h->gl.LWs;
  v.JS = std::max(1, std::min(h->M, (64 * 1024) / (32 * 4 * h->A)));
  // DNA: activations in registers; any other alphabet: one position per thread, activations in LDS (big_vgh_any_kernel)
  const size_t lds = h->A == 4 ? (size_t)v.JS * 32 * 16 + (size_t)BIG_VR * 256 + (size_t)2 * (std::min(BIG_VR * 256, h->Lv) + h->M - 1) * 4   // (+ one mask word of both strands for a chunk)
                               : ((size_t)v.JS * 32 * h->A + (size_t)h->A * 256) * 4 + 256;
  HIPCHK(hipMemsetAsync(h->d_ones, 0, sizeof(unsigned long long), st));
  for (int t = 0; t < steps; ++t) {
    v.rng = rng_view(h, h->gibbs_step + (uint32_t)t, h->chain_offset);
    if (h->A == 4) hipLaunchKernelGGL(big_vgh_kernel, dim3(std::max(1, std::min(h->B, h->num_cu * 8))), dim3(256), lds, st, v);
    else hipLaunchKernelGGL(big_vgh_any_kernel, dim3(std::max(1, std::min(h->B, h->num_cu * 8))), dim3(256), lds, st, v);
    HIPCHK(hipGetLastError());
    const bool last = t == steps - 1;      // the last step also counts the units that are on (activity monitor)
    for (int strand = 0; strand <= h->ds; ++strand) {
      const int rc = big_launch_hgv(h, h->d_vf, h->B, h->Lv, strand, nullptr, nullptr, nullptr, last ? h->d_ones : nullptr,
                                    strand ? h->d_hmp : h->d_hm, KIND_CHAIN_H, h->gibbs_step + (uint32_t)t, h->chain_offset, st);
      if (rc) return rc;
    }
  }
  h->gibbs_step += (uint32_t)steps;
  h->launches_since_read += 1;
  return CRBM_OK;
}

// raw statistic sums of (letters, n, L) into partial rows; the column reduction is handed back like launch_stats does
// (reduce->row == 0: nothing left to reduce -- the slabbed form below has written the sums itself)
int big_launch_stats(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool data_half, hipStream_t st, ReduceArgs* reduce) {
  if (h->slab) {
    const int rc = slab_launch_stats(h, d_letters, n, L, data_half, st, reduce);
    if (rc != SLAB_FALLBACK) return rc;
  }
  DevBuf<float>& pbuf = data_half ? h->partials : h->partials2;
  const int K = h->K, M = h->M, KAM = h->KAM;
  BigStatsArgs a;
  a.m = big_model(h);
  a.letters = d_letters;
  a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = lw(h, L);
  a.want_sparsity = data_half ? 1 : 0;
  a.R = std::max(1, std::min(n, std::max(1, (h->num_cu * 8) / K)));
  a.pool = h->ms.POOL;
  a.CH = std::max(64, std::min(4096, a.Lh));
  a.CH = std::max(a.pool, a.CH - a.CH % a.pool);            // chunks start on pooling-group boundaries
  a.row = 3 * KAM + 3 * K + h->A;
  a.off_vh0 = 0; a.off_vh1 = KAM; a.off_h0 = 2 * KAM; a.off_h1 = 2 * KAM + K;
  a.off_sw = 2 * KAM + 2 * K; a.off_sb = 3 * KAM + 2 * K; a.off_v = 3 * KAM + 3 * K;
  HIPCHK(pbuf.ensure((size_t)a.R * a.row));
  a.partials = pbuf.p;
  ARGCHK(h->A * M <= BIG_ST * 256, "motif_length too large for the statistics kernel");
  const size_t lds = (((size_t)h->A * M + 3) & ~(size_t)3) * 4 + (size_t)(a.pool > 1 ? 5 : 3) * a.CH * 4 + 64 + 12 * 256 * 4 +
                     (size_t)((h->A + 3) & ~3) * 4 + (size_t)a.CH + M;
  ARGCHK(lds <= 160 * 1024, "motif_length too large for the statistics kernel");
  hipLaunchKernelGGL(big_stats_kernel, dim3(K, a.R), dim3(256), lds, st, a);
  HIPCHK(hipGetLastError());
  ReduceArgs& r = *reduce;
  r.partials = pbuf.p;
  r.nrows = a.R; r.row = a.row;
  r.K = K; r.KAM = KAM; r.ds = h->ds; r.want_sparsity = a.want_sparsity;
  if (data_half) {
    r.sums = h->d_sums + h->sl.data_off;
    r.skip_begin = a.row; r.skip_len = 0;
  } else {
    r.sums = h->d_sums + h->sl.model_off;
    r.skip_begin = h->sl.model_skip_begin; r.skip_len = h->sl.model_skip_len;
  }
  r.n_value = (float)n;
  return CRBM_OK;
}

// free energies (hits = 0) or motif-hit summaries (hits = 1) of n packed rows
int big_launch_eval(crbm_handle* h, const uint32_t* rows, int n, int L, int hits, float* fe, float* fem, float* hmax, float* hmean,
                    unsigned long long* pos_fx, hipStream_t st) {
  if (h->slab && !hits && env_int("CRBM_SLAB_FE", 1) != 0) {       // free energies: the specialised kernel, slab by slab
    const int rc = slab_launch_fe(h, rows, n, L, fe, fem, st);
    if (rc != SLAB_FALLBACK) return rc;
  }
  BigEvalArgs a;
  a.m = big_model(h);
  a.letters = rows;
  a.n = n; a.L = L; a.Lh = L - h->M + 1; a.LW = lw(h, L);
  a.fe = fe; a.fem = fem; a.hmax = hmax; a.hmean = hmean; a.pos_fx = pos_fx; a.hits = hits; a.pool = h->ms.POOL;
  const size_t lds = (((size_t)h->A * h->M + 3) & ~(size_t)3) * 4 + 64 + (size_t)L;
  ARGCHK(lds <= 160 * 1024, "sequence too long for the evaluation kernel of a model of this size");
  hipLaunchKernelGGL(big_eval_kernel, dim3(std::max(1, std::min(n, h->num_cu * 8))), dim3(256), lds, st, a);
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}

int grid_for(long items, int threads, int cap) {
  long g = (items + threads - 1) / threads;
  return (int)std::max<long>(1, std::min<long>(g, cap));
}

// sequences per tile so that a tile has ~target items and ts*len*mult < 2^20
int tile_seqs(int len, int target) { return std::max(1, std::min(target / std::max(1, len), (1 << 20) / std::max(1, len))); }

int check_flags(crbm_handle* h) {
  uint32_t flags = 0;
  HIPCHK(hipMemcpyAsync(&flags, h->d_flags, sizeof(flags), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (flags) {
    HIPCHK(hipMemsetAsync(h->d_flags, 0, sizeof(uint32_t), h->stream));
    if (flags & 1u) return fail(h, CRBM_ERR_NOT_ONEHOT, "visible data is not exactly one-hot");
    return fail(h, CRBM_ERR_NOT_BINARY, "hidden state is not exactly 0/1");
  }
  return CRBM_OK;
}

// host one-hot (n,1,A,L) -> packed letters on the device
int encode_host(crbm_handle* h, const float* v, int n, int L, uint32_t* d_letters) {
  const size_t count = (size_t)n * h->A * L;
  HIPCHK(h->stage.ensure(count));
  HIPCHK(hipMemcpyAsync(h->stage.p, v, count * sizeof(float), hipMemcpyHostToDevice, h->stream));
  EncodeArgs a;
  a.v = h->stage.p; a.letters = d_letters; a.flags = h->d_flags;
  a.n = n; a.L = L; a.LW = lw(h, L); a.A = h->A;
  const int grid = grid_for((long)n * a.LW, 256, h->num_cu * 8);
  if (h->A == 4) hipLaunchKernelGGL(encode_onehot_kernel, dim3(grid), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(encode_onehot_any_kernel, dim3(grid), dim3(256), 0, h->stream, a);
  HIPCHK(hipGetLastError());
  return check_flags(h);
}

// host bytes (n,L) -> packed letters on the device
int encode_codes_host(crbm_handle* h, const uint8_t* codes, int n, int L, uint32_t* d_letters) {
  const size_t bytes = (size_t)n * L;
  HIPCHK(h->stage.ensure((bytes + 3) / 4));
  HIPCHK(hipMemcpyAsync(h->stage.p, codes, bytes, hipMemcpyHostToDevice, h->stream));
  EncodeCodesArgs a;
  a.codes = reinterpret_cast<const unsigned char*>(h->stage.p);
  a.letters = d_letters; a.flags = h->d_flags;
  a.n = n; a.L = L; a.LW = lw(h, L); a.A = h->A;
  if (h->A == 4) hipLaunchKernelGGL(encode_codes_kernel, dim3(grid_for((long)n * a.LW, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(encode_codes_any_kernel, dim3(grid_for((long)n * a.LW, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
  HIPCHK(hipGetLastError());
  return check_flags(h);
}

int launch_hgv(crbm_handle* h, const uint32_t* d_letters, int n, int L, int mode, float* act, float* prob,
               float* sample, unsigned long long* ones, uint32_t kind, uint32_t step, uint32_t seq_offset) {
  if (h->big) return big_launch_hgv(h, d_letters, n, L, mode, act, prob, sample, ones, nullptr, kind, step, seq_offset, h->stream);
  int rc = ensure_tables(h);
  if (rc) return rc;
  HgvArgs a;
  a.tables = h->d_tables;
  a.letters = d_letters;
  a.n = n; a.L = L; a.Lh = L - h->M + 1; a.LW = lw(h, L);
  a.TS = tile_seqs(a.Lh, 4096);
  a.divLh = make_fastdiv((uint32_t)a.Lh);
  a.mode = mode;
  a.act = act; a.prob = prob; a.sample = sample; a.ones = ones;
  a.rng = rng_view(h, step, seq_offset);
  a.kind = kind;
  const int ntiles = (n + a.TS - 1) / a.TS;
  const unsigned gx = (unsigned)std::max(1, std::min(ntiles, h->num_cu * 8));
  HIPCHK(jit_launch(h->jk.hgv, a, gx, 1, 256, (unsigned)tab_bytes(h), h->stream));
  return CRBM_OK;
}

void use_variant(crbm_handle* h, int v) {
  h->variant = v;
  h->gl = h->glv[v];
  h->gibbs_threads = h->threadsv[v];
  h->gibbs_grid = h->gridv[v];
}

StatsGeom stats_geom(const StatsMfmaLayout& st, float* partials, long ngroups, int lds_bytes);

// Arguments of a Gibbs launch of `steps` steps.  model_reduce != null selects the STATS variant: the
// launch also leaves the model half of the gradient statistics as one partial row per block in
// h->partials2 and the column reduction is handed back through `model_reduce`.
int prepare_gibbs(crbm_handle* h, int steps, ReduceArgs* model_reduce, GibbsArgs* out, unsigned* lds_out) {
  const bool with_stats = model_reduce != nullptr;
  GibbsArgs& a = *out;
  a.tables = h->d_tables;
  a.tables_tf = nullptr; a.off_ws = 0;
  a.hm = h->d_hm; a.hmp = h->ds ? h->d_hmp : nullptr; a.vout = h->d_vf;
  a.nchains = h->B; a.Lf = h->Lf; a.Lv = h->gl.Lv; a.S = h->gl.S;
  a.nvb = h->gl.nvb; a.nhb = h->gl.nhb; a.Lrow = h->gl.Lrow; a.LWs = h->gl.LWs;
  a.divVB = make_fastdiv((uint32_t)a.nvb);
  a.divHB = make_fastdiv((uint32_t)a.nhb);
  a.divRow = make_fastdiv((uint32_t)(a.Lrow * h->NW));
  a.divLfw = make_fastdiv((uint32_t)(a.Lf * h->NW));
  a.steps = steps;
  a.rng = rng_view(h, h->gibbs_step, h->chain_offset);
  a.ones = h->d_nset;
  a.clock = h->probe_on ? h->d_probe : nullptr;
  a.timeline = nullptr;
  a.debug = env_int("CRBM_GIBBS_DEBUG", 0);
  h->nset_slots = h->gibbs_grid * (h->gibbs_threads / 64);
  a.nblocks = h->gibbs_grid;
  a.stats_off = 0;
  a.sg = StatsGeom();
  unsigned lds = (unsigned)h->gl.lds_bytes;
  if (with_stats) {
    const StatsMfmaLayout st = stats_mfma_layout(h->ms, 0, h->Lf, h->gibbs_threads, 0, false);
    HIPCHK(h->partials2.ensure((size_t)h->gibbs_grid * st.row));
    a.stats_off = (h->gl.lds_bytes / 4 + 3) & ~3;
    lds = (unsigned)std::max((a.stats_off + st.region_floats) * 4, st.combine_bytes);
    a.sg = stats_geom(st, h->partials2.p, (long)h->gl.S * st.GPC, (int)lds);
    ReduceArgs& r = *model_reduce;
    r.partials = h->partials2.p;
    r.nrows = h->gibbs_grid; r.row = st.row;
    r.K = h->K; r.KAM = h->KAM; r.ds = h->ds; r.want_sparsity = 0;
    r.sums = h->d_sums + h->sl.model_off;
    r.skip_begin = h->sl.model_skip_begin; r.skip_len = h->sl.model_skip_len;
    r.n_value = (float)h->B;
  }
  *lds_out = lds;
  return CRBM_OK;
}

void part_worker_main(PartWorker* w) {
  (void)hipSetDevice(w->device);
  std::unique_lock<std::mutex> lock(w->mu);
  for (;;) {
    // launches come in bursts: poll for a couple of milliseconds before going to sleep (a sleeping thread picks its
    // first job up 20-35 us late, and that is how late the partition's first launch of a burst then starts)
    const auto spin_until_t = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
    while (!w->stop && w->jobs.empty() && std::chrono::steady_clock::now() < spin_until_t) {
      lock.unlock();
      for (int i = 0; i < 64; ++i) __builtin_ia32_pause();
      lock.lock();
    }
    w->cv.wait(lock, [&] { return w->stop || !w->jobs.empty(); });
    if (w->jobs.empty()) return;       // stop, nothing left
    PartJob job = w->jobs.front();
    w->jobs.pop_front();
    w->busy = true;
    lock.unlock();
    hipError_t e = hipSuccess;
    if (job.delay_ticks > 0) {
      hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, w->stream, job.delay_ticks);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = jit_launch(job.fn, job.args, job.grid, 1, job.threads, job.lds, w->stream);
    lock.lock();
    if (e != hipSuccess && w->error == hipSuccess) w->error = e;
    w->busy = false;
    if (w->jobs.empty()) w->idle_cv.notify_all();
  }
}

// every launch handed to the partitions' threads has been enqueued on its stream
int drain_part_workers(crbm_handle* h) {
  for (int p = 1; p < h->chain_parts; ++p) {
    PartWorker* w = h->part_worker[p];
    if (!w) continue;
    std::unique_lock<std::mutex> lock(w->mu);
    w->idle_cv.wait(lock, [&] { return w->jobs.empty() && !w->busy; });
    if (w->error != hipSuccess) {
      const hipError_t e = w->error;
      w->error = hipSuccess;
      return fail(h, CRBM_ERR_HIP, std::string("chain launch of partition ") + std::to_string(p) + ": " + hipGetErrorString(e));
    }
  }
  return CRBM_OK;
}

// every partition stream has finished what it was given as far as the main stream is concerned
int join_parts(crbm_handle* h) {
  if (!h->forked) return CRBM_OK;
  {
    const int rc = drain_part_workers(h);
    if (rc) return rc;
  }
  for (int p = 1; p < h->chain_parts; ++p) {          // (partition 0 is the main stream itself)
    HIPCHK(hipEventRecord(h->part_done[p], h->part_stream[p]));
    HIPCHK(hipStreamWaitEvent(h->stream, h->part_done[p], 0));
  }
  h->forked = false;
  return CRBM_OK;
}

// `steps` Gibbs steps of all chains as chain_parts launches on the partition streams (see crbm_handle::chain_parts)
int launch_gibbs_parts(crbm_handle* h, int steps) {
  bool first_of_fork = false;
  double stagger_ticks = 0.0;      // first launch after a fork: partition p starts p times this late
  if (!h->forked) {     // the partitions start after whatever the main stream holds (parameter updates, state uploads)
    if (!h->main_idle_hint && hipStreamQuery(h->stream) != hipSuccess) {     // ... unless it is idle: nothing to wait for, no cross-stream dependency to resolve
      (void)hipGetLastError();
      HIPCHK(hipEventRecord(h->ev_parts_fork, h->stream));
      for (int p = 1; p < h->chain_parts; ++p) HIPCHK(hipStreamWaitEvent(h->part_stream[p], h->ev_parts_fork, 0));
    }
    h->main_idle_hint = false;
    h->forked = true;
    first_of_fork = true;
    if (h->cal_pending && hipEventQuery(h->ev_cal1) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev_cal0, h->ev_cal1) == hipSuccess && ms > 0.f) h->part_launch_us = 1e3 * ms;
      h->cal_pending = false;
    }
    (void)hipGetLastError();
    if (env_int("CRBM_PART_STAGGER", 1)) {
      // (before the first measurement: ~1 ns per hidden unit and Gibbs step, what configs #2 and #5 run at)
      const double est = h->part_launch_us > 0.0 ? h->part_launch_us
                                                 : 1e-3 * (double)h->B * h->Lf * h->K * (1 + h->ds) * steps;
      stagger_ticks = 0.9 * est / h->chain_parts * (h->wall_khz / 1000.0);
    }
  }
  GibbsArgs a;
  unsigned lds = 0;
  const GibbsLayout gl_keep = h->gl;
  const int threads_keep = h->gibbs_threads, grid_keep = h->gibbs_grid;
  h->gl = h->gl_part; h->gibbs_threads = h->part_threads; h->gibbs_grid = h->part_grid;
  int rc = prepare_gibbs(h, steps, nullptr, &a, &lds);
  const unsigned grid = (unsigned)h->gibbs_grid, threads = (unsigned)h->gibbs_threads;
  h->gl = gl_keep; h->gibbs_threads = threads_keep; h->gibbs_grid = grid_keep;
  if (rc) return rc;
  const size_t per = (size_t)h->Lf * h->NW;
  const int waves = (int)threads / 64;
  h->nset_slots = h->chain_parts * (int)grid * waves;
  for (int p = 0; p < h->chain_parts; ++p) {
    const int c0 = p * h->part_chains, n = std::min(h->part_chains, h->B - c0);
    if (n <= 0) break;
    GibbsArgs ap = a;
    ap.hm = a.hm + (size_t)c0 * per;
    if (ap.hmp) ap.hmp = a.hmp + (size_t)c0 * per;
    ap.vout = a.vout + (size_t)c0 * a.LWs;
    ap.nchains = n;
    ap.rng.seq_offset = a.rng.seq_offset + (uint32_t)c0;
    ap.ones = a.ones + (size_t)p * grid * waves;
    if (p > 0) ap.clock = nullptr;
    if (h->d_timeline && h->timeline_next < h->timeline_cap) ap.timeline = h->d_timeline + 2 * (size_t)(h->timeline_next++);
    ap.nblocks = (int)grid;
    if (p > 0) {          // the partition's own thread enqueues it
      PartWorker* w = h->part_worker[p];
      {
        std::lock_guard<std::mutex> lock(w->mu);
        w->jobs.push_back(PartJob{h->jk.gibbs_sparse, ap, grid, threads, lds, (unsigned long long)(stagger_ticks * p)});
      }
      w->cv.notify_one();
      continue;
    }
    const bool calibrate = first_of_fork && !h->cal_pending;
    if (calibrate) HIPCHK(hipEventRecord(h->ev_cal0, h->part_stream[0]));
    HIPCHK(jit_launch(h->jk.gibbs_sparse, ap, grid, 1, threads, lds, h->part_stream[0]));
    if (calibrate) {
      HIPCHK(hipEventRecord(h->ev_cal1, h->part_stream[0]));
      h->cal_pending = true;
    }
  }
  h->gibbs_step += (uint32_t)steps;
  h->launches_since_read += 1;
  return CRBM_OK;
}

// plain: a launch that only advances the chains, from the API (crbm_gibbs_steps*, crbm_time_gibbs) -- those may go out in
// partitions; the chain launch inside a training step is followed at once by kernels that wait for all of it and stays whole
int launch_gibbs(crbm_handle* h, int steps, hipStream_t s = nullptr, ReduceArgs* model_reduce = nullptr, bool plain = false) {
  if (!s) s = h->stream;
  if (h->big) return big_launch_gibbs(h, steps, s);
  int rc = ensure_tables(h);
  if (rc) return rc;
  if (plain && !model_reduce && h->variant == 1 && h->chain_parts > 1 && s == h->stream) return launch_gibbs_parts(h, steps);
  GibbsArgs a;
  unsigned lds = 0;
  const bool solo = !model_reduce && h->variant == 1 && h->solo_threads > 0;
  const GibbsLayout gl_keep = h->gl;
  const int threads_keep = h->gibbs_threads, grid_keep = h->gibbs_grid;
  if (solo) { h->gl = h->gl_solo; h->gibbs_threads = h->solo_threads; h->gibbs_grid = h->solo_grid; }
  rc = prepare_gibbs(h, steps, model_reduce, &a, &lds);
  if (!model_reduce && h->variant == 1 && h->GS != h->G) {     // crbm_gibbs_sparse is compiled for the solo grouping
    a.tables_tf = h->d_tf_solo; a.off_ws = h->ms.OFF_WS;
    if (rc == CRBM_OK) rc = ensure_solo_table(h);
  }
  const unsigned grid = (unsigned)h->gibbs_grid, threads = (unsigned)h->gibbs_threads;
  if (solo) { h->gl = gl_keep; h->gibbs_threads = threads_keep; h->gibbs_grid = grid_keep; }
  if (rc) return rc;
  if (plain && h->d_timeline && h->timeline_next < h->timeline_cap) a.timeline = h->d_timeline + 2 * (size_t)(h->timeline_next++);
  hipFunction_t fn = model_reduce ? h->jk.gibbs_sparse_stats : (h->variant ? h->jk.gibbs_sparse : h->jk.gibbs);
  HIPCHK(jit_launch(fn, a, grid, 1, threads, lds, s));
  h->gibbs_step += (uint32_t)steps;
  h->launches_since_read += 1;
  return CRBM_OK;
}

// Reads the activity monitor of the last Gibbs launch back (crbm_sync /
// crbm_gibbs_steps only: a blocking copy, kept out of every timed or training
// path).  It is a monitor, nothing else: the top-down variant is fixed per
// handle, so every rank of a data-parallel job -- and a 1-GPU run of the same
// chains -- rounds identically and draws identical samples.
int refresh_activity(crbm_handle* h) {
  if (h->launches_since_read == 0) return CRBM_OK;
  if (h->big) {          // the last step's h|v launches counted into d_ones
    unsigned long long on = 0;
    HIPCHK(hipMemcpy(&on, h->d_ones, sizeof(on), hipMemcpyDeviceToHost));
    h->launches_since_read = 0;
    h->activity = (double)on / ((double)h->B * h->Lf * h->K * (1 + h->ds));
    return CRBM_OK;
  }
  std::vector<uint32_t> slots((size_t)h->nset_slots);
  HIPCHK(hipMemcpy(slots.data(), h->d_nset, slots.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  unsigned long long last = 0;
  for (uint32_t v : slots) last += v;
  h->launches_since_read = 0;
  h->activity = (double)last / ((double)h->B * h->Lf * h->K * (1 + h->ds));
  return CRBM_OK;
}

StatsGeom stats_geom(const StatsMfmaLayout& st, float* partials, long ngroups, int lds_bytes) {
  StatsGeom g;
  g.GPC = st.GPC;
  g.lds_floats = lds_bytes / 4;
  g.off_slices = st.off_slices; g.slice = st.slice;
  g.off_win = st.off_win; g.off_gw = st.off_gw; g.off_pt = st.off_pt;
  g.divGPC = make_fastdiv((uint32_t)st.GPC, (uint64_t)std::max<long>(ngroups, 1));
  g.row = st.row; g.off_vh0 = st.off_vh[0]; g.off_vh1 = st.off_vh[1]; g.off_h0 = st.off_h[0]; g.off_h1 = st.off_h[1];
  g.off_sw = st.off_sw; g.off_sb = st.off_sb; g.off_v = st.off_v;
  g.partials = partials;
  return g;
}

// Arguments of the MFMA statistics kernel (stats_mfma_body) for (letters, n, L): raw statistic sums ->
// partial rows of the data or the model half.  `threads` 0: the kernel's own block size and the byte
// LUT; otherwise the geometry of a host kernel it rides in (nibble LUT).
int prepare_stats(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool data_half, int threads,
                  StatsMfmaArgs* out, int* lds_out, int* grid_out, int* block_out, ReduceArgs* reduce) {
  DevBuf<float>& pbuf = data_half ? h->partials : h->partials2;
  const int want_sp = data_half ? 1 : 0;
  const int Lh = L - h->M + 1;
  const int tabs = tab_bytes(h);
  const bool own = threads <= 0;
  // CRBM_STATS_MAX_TILES is an experiment knob: it must come with CRBM_JIT_DEFINES=-DCRBM_STATS_MAX_TILES=<same>
  const StatsMfmaLayout st = stats_mfma_layout(h->ms, want_sp, Lh, own ? env_int("CRBM_STATS_THREADS", 0) : threads, tabs, own,
                                               env_int("CRBM_STATS_MAX_TILES", CRBM_STATS_MAX_TILES));
  StatsMfmaArgs& a = *out;
  a.tables = h->d_tables;
  a.letters = d_letters;
  a.n = n; a.L = L; a.Lh = Lh; a.LW = lw(h, L);
  const long ngroups = (long)n * st.GPC;
  const long nunits = (ngroups + 1) / 2;
  a.off_tab = st.region_floats;
  a.debug = env_int("CRBM_STATS_DEBUG", 0);
  ARGCHK(ngroups < (1L << 31), "batch too large for the statistics kernel");
  const int lds = std::max(st.region_floats * 4 + tabs, st.combine_bytes);
  ARGCHK(lds <= 160 * 1024, "model too large for the statistics kernel");
  const int wpr = (st.threads / 64) / st.NR;                  // waves per role = units a block works on at a time
  // Grid: a persistent set of blocks, every wave with the same number of units (a ragged last round costs
  // a whole unit time: at config #2 768 blocks of 8 units per wave take 73 us per training step, 832 blocks
  // 78).  Stand-alone: as many blocks as are resident at once; riding in the training launch: three per CU
  // beside the four chain blocks (measured optimum, fewer partial rows to combine and reduce).
  const int per_cu = own ? std::max(1, std::min(2048 / st.threads, (160 * 1024) / std::max(1, lds))) : 3;
  // (riding beside the chain blocks: three 256-thread blocks per CU, the same number of waves with larger blocks)
  const long cap = h->stats_rows > 0 ? h->stats_rows : own ? (long)h->num_cu * per_cu : std::max<long>(1, (long)h->num_cu * 3 * 256 / st.threads);
  const long upw = std::max<long>(1, (nunits + cap * wpr - 1) / (cap * wpr));        // units per wave
  const int gx = (int)std::max<long>(1, (nunits + upw * wpr - 1) / (upw * wpr));
  HIPCHK(pbuf.ensure((size_t)gx * st.row));
  a.nblocks = gx;
  a.sg = stats_geom(st, pbuf.p, ngroups, lds);
  *lds_out = lds; *grid_out = gx; *block_out = st.threads;
  ReduceArgs& r = *reduce;
  r.partials = pbuf.p;
  r.nrows = gx; r.row = st.row;
  r.K = h->K; r.KAM = h->KAM; r.ds = h->ds; r.want_sparsity = want_sp;
  if (data_half) {
    r.sums = h->d_sums + h->sl.data_off;
    r.skip_begin = st.row; r.skip_len = 0;
  } else {
    r.sums = h->d_sums + h->sl.model_off;
    r.skip_begin = h->sl.model_skip_begin; r.skip_len = h->sl.model_skip_len;
  }
  r.n_value = (float)n;
  return CRBM_OK;
}

// The statistics of a generic DNA model (motif_length <= 64) on the matrix cores.  VH[k], H[k] and the sparsity sums of motif
// k depend on that motif's filter alone, so the model is a row of independent sub-models of `slab->K` motifs: the
// specialised statistics kernel of that sub-model (stats_mfma_body: gather table in LDS, P split into f16 halves,
// v_mfma_f32_16x16x32_f16) runs once per slab on the slab's own table image -- W and b of a slab are contiguous pieces of
// the model's (K,4,M) and (K) arrays -- and slab_reduce_kernel adds the slab's partial rows into its columns of d_sums.
// The last slab of a model whose K is not a multiple of the slab is moved back to end at K: the motifs it shares with its
// neighbour get the same sums twice (an accumulator's value does not depend on the column it sits in).
// first motif of slab i: slabs of Ks motifs side by side; the last one of a model whose K is no multiple of Ks starts on the
// first multiple of ten at or behind K - Ks (sampler groups are ten units wide: every unit keeps its group and field), so it
// overlaps its neighbour and may reach up to nine motifs past K -- into the zero padding of W and b (crbm_create)
int slab_origin(const crbm_handle* h, int i) {
  const int Ks = h->slab->K;
  if ((i + 1) * Ks <= h->K) return i * Ks;
  return h->slab_hgv ? ((h->K - Ks + 9) / 10) * 10 : h->K - Ks;
}

SlabPlan slab_plan(const crbm_handle* h) {
  SlabPlan p;
  p.Ks = h->slab->K; p.K = h->K; p.last_k0 = slab_origin(h, h->slab_n - 1);
  return p;
}

// one launch builds the table images of all slabs (blockIdx.y = slab: slab_tables_body)
int slab_ensure_tables(crbm_handle* h, hipStream_t st) {
  crbm_handle* s = h->slab;
  if (h->slab_tables_version == h->params_version) return CRBM_OK;
  SlabTablesArgs a;
  a.t.W = h->dW; a.t.b = h->db; a.t.c = h->dc; a.t.out = h->d_slab_tables;
  a.plan = slab_plan(h);
  a.stride = s->ms.TABLES_ALL;
  const unsigned grid = (unsigned)std::max(1, std::min((s->ms.TABLES_ALL + 255) / 256, h->num_cu * 4));
  HIPCHK(jit_launch(s->jk.slab_tables, a, grid, (unsigned)h->slab_n, 256, 0, st));
  h->slab_tables_version = h->params_version;
  return CRBM_OK;
}

// h | v of a chain of a generic DNA model: the specialised h|v pass of the slab model (hgv_masks_body: gather table in LDS,
// all of the slab's units of a position in one lane, sign-word sampling), all slabs in one launch (blockIdx.y = slab); every
// unit draws from the counter it has in the whole model (HgvMasksArgs::group0) and lands in its bit of the model's mask rows
// (HgvMasksArgs::masks, atomicOr: neighbouring slabs share words).
int slab_launch_hgv(crbm_handle* h, const uint32_t* d_letters, int n, int L, int mode, unsigned long long* ones, uint32_t* masks,
                    uint32_t kind, uint32_t step, uint32_t seq_offset, hipStream_t st) {
  crbm_handle* s = h->slab;
  const int Lh = L - h->M + 1;
  if (tab_bytes(s) > 160 * 1024) return SLAB_FALLBACK;
  int rc = slab_ensure_tables(h, st);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(masks, 0, (size_t)n * Lh * h->NW * sizeof(uint32_t), st));
  SlabHgvArgs sa;
  HgvArgs& a = sa.m.g;
  a.tables = h->d_slab_tables;
  a.letters = d_letters;
  a.n = n; a.L = L; a.Lh = Lh; a.LW = lw(h, L);
  // rows per tile: ~4096 positions, fewer for small batches (enough blocks, with the slabs, to cover the chip a few times over)
  a.TS = std::max(1, std::min(tile_seqs(a.Lh, 4096), (n * h->slab_n + 4 * h->num_cu - 1) / (4 * h->num_cu)));
  a.divLh = make_fastdiv((uint32_t)a.Lh);
  a.mode = mode;
  a.act = nullptr; a.prob = nullptr; a.sample = nullptr; a.ones = ones;
  a.rng = rng_view(h, step, seq_offset);
  a.kind = kind;
  sa.m.masks = masks; sa.m.NWfull = h->NW; sa.m.Kfull = h->K;
  sa.m.k0 = 0; sa.m.kskip = 0; sa.m.group0 = 0;          // per slab: slab_hgv_body
  sa.plan = slab_plan(h);
  sa.table_stride = s->ms.TABLES_ALL;
  const int ntiles = (n + a.TS - 1) / a.TS;
  const unsigned gx = (unsigned)std::max(1, std::min(ntiles, h->num_cu * 8));
  HIPCHK(jit_launch(s->jk.slab_hgv, sa, gx, (unsigned)h->slab_n, 256, (unsigned)tab_bytes(s), st));
  return CRBM_OK;
}

// The statistics of a generic DNA model: the specialised statistics kernel of the slab model, all slabs in one launch
// (blockIdx.y = slab: its table image, its partial rows), then slab_reduce_kernel -- one launch too -- reduces every slab's
// partial rows into the slab's columns of the model's sums.
int slab_launch_stats(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool data_half, hipStream_t st, ReduceArgs* reduce) {
  crbm_handle* s = h->slab;
  const int Ks = s->K, M = h->M;
  s->err.clear();
  {
    const int trc = slab_ensure_tables(h, st);
    if (trc) return trc;
  }
  s->d_tables = h->d_slab_tables;
  SlabStatsArgs sa;
  ReduceArgs r;
  int lds = 0, gx = 0, block = 0;
  const int rc = prepare_stats(s, d_letters, n, L, data_half, 0, &sa.a, &lds, &gx, &block, &r);
  if (rc == CRBM_ERR_INVALID) return SLAB_FALLBACK;      // (a data set whose rows the slab kernel's LDS does not take)
  if (rc) return fail(h, rc, "slabbed statistics: " + s->err);
  DevBuf<float>& pbuf = data_half ? s->partials : s->partials2;
  const size_t per_slab = (size_t)r.nrows * r.row;
  HIPCHK(pbuf.ensure(per_slab * h->slab_n));
  sa.a.sg.partials = pbuf.p;
  sa.table_stride = s->ms.TABLES_ALL; sa.pad_ = 0;
  sa.partial_stride = (long long)per_slab;
  if (jit_launch(data_half ? s->jk.slab_stats_data : s->jk.slab_stats_model, sa, (unsigned)gx, (unsigned)h->slab_n, (unsigned)block, (unsigned)lds, st) != hipSuccess)
    return fail(h, CRBM_ERR_HIP, "launch of the slab statistics kernel failed");
  SlabReduceArgs ra;
  ra.partials = pbuf.p;
  ra.sums = h->d_sums + (data_half ? h->sl.data_off : h->sl.model_off);
  ra.nrows = r.nrows; ra.row = r.row;
  ra.Ks = Ks; ra.k0 = slab_origin(h, h->slab_n - 1); ra.K = h->K; ra.M4 = 4 * M;
  ra.partial_stride = (long long)per_slab;
  ra.ds = h->ds; ra.want_sparsity = data_half ? 1 : 0;
  const int full_row = 3 * h->KAM + 3 * h->K + 4;
  ra.skip_begin = data_half ? full_row : h->sl.model_skip_begin;
  ra.skip_len = data_half ? 0 : h->sl.model_skip_len;
  ra.n_value = (float)n;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((ra.row + 31) / 32, h->slab_n), dim3(1024), 0, st, ra);
  HIPCHK(hipGetLastError());
  *reduce = ReduceArgs();
  reduce->nrows = 0; reduce->row = 0;     // nothing left for the caller's column reduction
  return CRBM_OK;
}

// Free energies of a generic DNA model (freeEnergy, evaluateData, the per-epoch evaluation of fit(): convRBM.py:657-697, :517-522):
// the slab model's free-energy kernel leaves every slab's per-motif terms in a scratch (one launch, blockIdx.y = slab),
// slab_fe_combine_kernel adds them up per sequence.
int slab_launch_fe(crbm_handle* h, const uint32_t* rows, int n, int L, float* fe, float* fem, hipStream_t st) {
  crbm_handle* s = h->slab;
  if (tab_bytes(s) > 160 * 1024) return SLAB_FALLBACK;
  int rc = slab_ensure_tables(h, st);
  if (rc) return rc;
  const size_t per_slab = (size_t)n * s->K;
  HIPCHK(s->out_b.ensure(per_slab * h->slab_n));
  SlabFeArgs sa;
  sa.a.tables = h->d_slab_tables;
  sa.a.letters = rows;
  sa.a.n = n; sa.a.L = L; sa.a.Lh = L - h->M + 1; sa.a.LW = lw(h, L);
  sa.a.fe = nullptr; sa.a.fem = s->out_b.p;
  sa.table_stride = s->ms.TABLES_ALL; sa.pad_ = 0;
  sa.fem_stride = (long long)per_slab;
  const unsigned gx = (unsigned)std::max(1, std::min((n + 3) / 4, h->num_cu * 8));
  HIPCHK(jit_launch(s->jk.slab_fe, sa, gx, (unsigned)h->slab_n, 256, (unsigned)tab_bytes(s), st));
  SlabFeCombineArgs ca;
  ca.scratch = s->out_b.p;
  ca.c_log2e = h->d_slab_tables + s->ms.OFF_C;
  ca.letters = rows;
  ca.n = n; ca.L = L; ca.LW = lw(h, L);
  ca.Ks = s->K; ca.K = h->K; ca.last_k0 = slab_origin(h, h->slab_n - 1); ca.nslab = h->slab_n;
  ca.fe = fe; ca.fem = fem;
  hipLaunchKernelGGL(slab_fe_combine_kernel, dim3(gx), dim3(256), 0, st, ca);
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}

// The slab model of a generic DNA model: up to `want` motifs are one slab, a larger model takes slabs of a multiple of ten
// motifs (the sampler's groups); the largest candidate whose statistics kernel fits the LDS.  Returns 0 motifs when none does.
int slab_choose(int K, int M, int ds, int pool, int Lf, int* G_out, ModelShape* ms_out) {
  const int want = std::max(10, std::min(env_int("CRBM_SLAB_MOTIFS", 60), 64));      // (the slab kernels are compiled for models of up to 64 motifs: crbm_jit.h)
  const int first = K <= want ? K : want / 10 * 10;
  for (int cand : {first, 40, 30, 20, 10}) {
    if (cand > K || (cand != first && cand >= first)) continue;
    int G = env_int("CRBM_SLAB_GROUP", 0);
    if (G < 1 || G > 4) {
      // the table budget of the specialised kernels; long motifs that it leaves with single letters take pairs where those fit
      // twice 48 KB (60 x 40 double-stranded: 77 KB, training step 1.55 -> 1.19 ms at 2048 chains; a larger budget for ALL slabs
      // costs the statistics kernel waves: 256 x 4 double-stranded 1.72 -> 1.99 ms)
      const int budget = env_int("CRBM_SLAB_TABLE_BUDGET", 26 * 1024);
      G = choose_group(cand, M, ds, budget);
      if (G == 1) G = choose_group(cand, M, ds, std::max(budget, 48 * 1024));
    }
    const ModelShape ms = model_shape(cand, M, ds, G, pool);
    bool fit = true;
    for (int want_sp = 0; want_sp <= 1 && fit; ++want_sp) {
      const int tabs = ms.TAB * 4;
      const StatsMfmaLayout st = stats_mfma_layout(ms, want_sp, Lf, 0, tabs, true);
      if (st.threads > 1024 || std::max(st.region_floats * 4 + tabs, st.combine_bytes) > 160 * 1024) fit = false;
    }
    if (fit) { *G_out = G; *ms_out = ms; return cand; }
  }
  return 0;
}

// The shadow handle of the slab model of a generic handle (crbm_create); leaves h->slab null, with the reason in
// h->slab_note, when the model is not one for slabs.  CRBM_SLAB_STATS=0 switches them off (A/B runs, tests).
int slab_setup(crbm_handle* h) {
  if (!h->big) return CRBM_OK;
  if (env_int("CRBM_SLAB_STATS", 1) == 0) { h->slab_note = "CRBM_SLAB_STATS=0"; return CRBM_OK; }
  if (h->A != 4 || h->M > MAX_MOTIF_LENGTH) { h->slab_note = "other alphabet, or motifs beyond 64 letters"; return CRBM_OK; }
  int G = 0;
  ModelShape ms;
  const int Ks = slab_choose(h->K, h->M, h->ds, h->ms.POOL, h->Lf, &G, &ms);
  if (!Ks) { h->slab_note = "no slab of this motif length fits the LDS"; return CRBM_OK; }
  crbm_handle* s = new crbm_handle();
  std::string err;
  if (jit_load(Ks, h->M, h->ds, G, G, ms.POOL, 0, 256, &s->jk, &err, true) != 0) {
    h->slab_note = "kernel specialisation of the slab failed: " + err;
    if (s->jk.module) (void)hipModuleUnload(s->jk.module);
    delete s;
    return CRBM_OK;
  }
  s->cfg = h->cfg; s->cfg.num_motifs = Ks;
  s->K = Ks; s->M = h->M; s->ds = h->ds; s->A = 4; s->G = G; s->GS = G; s->KAM = Ks * 4 * h->M;
  s->ms = ms; s->ms_solo = ms; s->NW = ms.NW;
  s->Lf = h->Lf; s->Lv = h->Lv; s->B = h->B;
  s->device = h->device; s->num_cu = h->num_cu;
  s->stream = h->stream;                 // borrowed: never destroyed through the shadow
  s->big = false; s->tables_dirty = false;
  s->sl = sums_layout(Ks, h->M);
  s->stats_rows = h->stats_rows;
  h->slab_n = (h->K + Ks - 1) / Ks;
  h->slab_hgv = (h->slab_n == 1 || Ks % 10 == 0) && env_int("CRBM_SLAB_HGV", 1) != 0;
  if (hipMalloc((void**)&s->d_sums, (size_t)s->sl.count * 4) != hipSuccess ||
      hipMalloc((void**)&h->d_slab_tables, (size_t)h->slab_n * ms.TABLES_ALL * 4) != hipSuccess) {
    (void)hipGetLastError();
    if (s->d_sums) (void)hipFree(s->d_sums);
    if (h->d_slab_tables) { (void)hipFree(h->d_slab_tables); h->d_slab_tables = nullptr; }
    (void)hipModuleUnload(s->jk.module);
    delete s;
    h->slab_note = "no memory for the slab tables";
    return CRBM_OK;
  }
  h->slab = s;
  return CRBM_OK;
}

void slab_destroy(crbm_handle* h) {
  if (crbm_handle* s = h->slab) {
    if (s->d_sums) (void)hipFree(s->d_sums);
    s->partials.release(); s->partials2.release(); s->out_b.release();
    if (s->jk.module) (void)hipModuleUnload(s->jk.module);
    delete s;
    h->slab = nullptr;
  }
  if (h->d_slab_tables) { (void)hipFree(h->d_slab_tables); h->d_slab_tables = nullptr; }
}

// stand-alone launch; the column reduction is handed back to the caller (`defer`, to pair it with the
// other half) or launched here
int launch_stats(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool data_half, hipStream_t s = nullptr,
                 ReduceArgs* defer = nullptr) {
  if (!s) s = h->stream;
  int rc = ensure_tables(h);
  if (rc) return rc;
  StatsMfmaArgs a;
  ReduceArgs r;
  int lds = 0, gx = 0, block = 0;
  if (h->big) rc = big_launch_stats(h, d_letters, n, L, data_half, s, &r);
  else {
    rc = prepare_stats(h, d_letters, n, L, data_half, 0, &a, &lds, &gx, &block, &r);
    if (rc) return rc;
    rc = jit_launch(data_half ? h->jk.stats_mfma_data : h->jk.stats_mfma_model, a, (unsigned)gx, 1, (unsigned)block, (unsigned)lds, s) == hipSuccess
             ? CRBM_OK : fail(h, CRBM_ERR_HIP, "launch of the statistics kernel failed");
  }
  if (rc) return rc;
  if (defer) { *defer = r; return CRBM_OK; }
  if (r.row == 0) return CRBM_OK;        // slabbed statistics: the sums are written
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((r.row + 31) / 32), dim3(1024), 0, s, r);
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}

void fill_update_args(crbm_handle* h, int L_data, UpdateTablesArgs& a) {
  UpdateArgs& u = a.u;
  u.sums = h->d_sums;
  u.W = h->dW; u.b = h->db; u.c = h->dc; u.vW = h->dvW; u.vb = h->dvb; u.vc = h->dvc;
  u.oW = h->dW2; u.ob = h->db2; u.oc = h->dc2; u.ovW = h->dvW2; u.ovb = h->dvb2; u.ovc = h->dvc2;
  u.K = h->K; u.M = h->M; u.ds = h->ds;
  u.L_data = L_data; u.Lf = h->Lf;
  u.data_off = h->sl.data_off; u.n_d = h->sl.n_d; u.model_off = h->sl.model_off; u.n_m = h->sl.n_m;
  u.lr = h->cfg.learning_rate; u.momentum = h->cfg.momentum; u.rho = h->cfg.rho; u.lambda_rate = h->cfg.lambda_rate;
  u.A = h->A;
  a.tables = h->d_tables;
}

// the update wrote the other set of buffers: it is the current one from now on
void swap_param_sets(crbm_handle* h) {
  std::swap(h->dW, h->dW2); std::swap(h->db, h->db2); std::swap(h->dc, h->dc2);
  std::swap(h->dvW, h->dvW2); std::swap(h->dvb, h->dvb2); std::swap(h->dvc, h->dvc2);
  h->tables_dirty = false;
  h->params_version += 1;
}

// update + rebuild of the table images in one launch (update_tables_body)
int launch_update(crbm_handle* h, int L_data) {
  UpdateTablesArgs a;
  fill_update_args(h, L_data, a);
  if (h->big) {        // element-wise and in place: no table images to rebuild, no second buffer set
    UpdateArgs& u = a.u;
    u.oW = h->dW; u.ob = h->db; u.oc = h->dc; u.ovW = h->dvW; u.ovb = h->dvb; u.ovc = h->dvc;
    hipLaunchKernelGGL(big_update_kernel, dim3(grid_for((long)h->KAM + h->K + h->A, 256, h->num_cu * 4)), dim3(256), 0, h->stream, u);
    HIPCHK(hipGetLastError());
    h->params_version += 1;
    return CRBM_OK;
  }
  const unsigned grid = (unsigned)std::max(1, std::min((h->ms.TABLES_ALL + 4095) / 4096, 32));
  HIPCHK(jit_launch(h->jk.update_tables, a, grid, 1, UPDATE_THREADS, (unsigned)((h->KAM + h->K + 4) * 4), h->stream));
  swap_param_sets(h);
  return CRBM_OK;
}

size_t ipc_bytes(const crbm_handle* h);
float* ipc_slot_of(void* base, const crbm_handle* h, int parity, int source);
uint32_t* ipc_flag_of(void* base, const crbm_handle* h, int parity, int source);

// the column reductions of both halves in one launch; with the mapped-buffer all-reduce on, straight into this
// rank's published buffer, flag included (reduce_publish_pair_kernel)
int launch_reduce_pair(crbm_handle* h, ReducePair pair, bool publish) {
  if (pair.half[0].row == 0 || pair.half[1].row == 0) {      // slabbed statistics have written their sums themselves (slab_launch_stats)
    for (auto& half : pair.half)
      if (half.row > 0) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((half.row + 31) / 32), dim3(1024), 0, h->stream, half);
        HIPCHK(hipGetLastError());
      }
    return CRBM_OK;
  }
  const dim3 grid((pair.half[0].row + 31) / 32, 2);
  if (publish && h->ipc_on) {
    // this rank's slot of the step's parity in every rank's buffer, the own one as the base the others are offsets of
    const int parity = (int)(h->ipc_step & 1u);
    float* base = ipc_slot_of(h->ipc_buf, h, parity, h->rank);
    for (auto& half : pair.half) half.sums = base + (half.sums - h->d_sums);
    PublishTail t;
    t.ticket = h->d_ticket; t.value = h->ipc_step + 1u;
    t.push.n = h->nranks;
    for (int r = 0; r < IPC_MAX_RANKS; ++r) {
      void* peer = r < h->nranks ? h->ipc_peer[r] : h->ipc_buf;
      t.push.delta[r] = (long long)(reinterpret_cast<char*>(ipc_slot_of(peer, h, parity, h->rank)) - reinterpret_cast<char*>(base));
      t.flag[r] = ipc_flag_of(peer, h, parity, h->rank);
    }
    hipLaunchKernelGGL(reduce_publish_pair_kernel, grid, dim3(1024), 0, h->stream, pair, t);
    h->ipc_published = true;
  } else {
    hipLaunchKernelGGL(reduce_partials_pair_kernel, grid, dim3(1024), 0, h->stream, pair);
  }
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}

// data statistics + k Gibbs steps + model statistics -> d_sums (local)
// publish: the caller runs the mapped-buffer all-reduce next (train_core), so the sums may go straight to the
// published buffer; crbm_train_local wants them in d_sums
int train_local_dev(crbm_handle* h, const uint32_t* d_letters, int n, int L, bool publish = false) {
  int rc = ensure_tables(h);
  if (rc) return rc;
  if (h->slab) {      // the slab tables too, on the main stream: both halves read them (CRBM_OVERLAP: from two streams)
    rc = slab_ensure_tables(h, h->stream);
    if (rc) return rc;
  }
  hipStream_t sm = h->overlap ? h->stream2 : h->stream;
  if (h->overlap) {
    HIPCHK(hipEventRecord(h->ev_fork, h->stream));          // tables ready, earlier work on the chains done
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
  }
  // single stream: the two column reductions share one launch
  ReducePair pair;
  const bool paired = !h->overlap && n > 0;
  if (h->fuse_stats && paired && h->one_launch) {
    // the whole local phase in one launch (train_local_body): chain + model half in the first
    // gibbs_grid blocks, data half in the blocks behind them
    TrainLocalArgs t;
    unsigned glds = 0;
    rc = prepare_gibbs(h, h->cfg.cd_k, &pair.half[1], &t.g, &glds);
    if (rc) return rc;
    int dlds = 0, dgrid = 0, dblock = 0;
    rc = prepare_stats(h, d_letters, n, L, true, h->gibbs_threads, &t.d, &dlds, &dgrid, &dblock, &pair.half[0]);
    if (rc) return rc;
    const unsigned lds = std::max(glds, (unsigned)dlds);
    t.g.sg.lds_floats = (int)(lds / 4);
    t.d.sg.lds_floats = (int)(lds / 4);
    HIPCHK(jit_launch(h->jk.train_local, t, (unsigned)(h->gibbs_grid + dgrid), 1, (unsigned)h->gibbs_threads, lds, h->stream));
    h->gibbs_step += (uint32_t)h->cfg.cd_k;
    h->launches_since_read += 1;
    return launch_reduce_pair(h, pair, publish);
  }
  if (h->fuse_stats) {
    // the Gibbs launch itself leaves the model half of the statistics (last-step probabilities
    // and visible sample never leave the chip)
    ReduceArgs mr;
    rc = launch_gibbs(h, h->cfg.cd_k, sm, &mr);
    if (rc) return rc;
    if (paired) pair.half[1] = mr;
    else {
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((mr.row + 31) / 32), dim3(1024), 0, sm, mr);
      HIPCHK(hipGetLastError());
    }
  } else {
    rc = launch_gibbs(h, h->cfg.cd_k, sm);
    if (rc) return rc;
    rc = join_parts(h);      // (a partitioned chain launch: the statistics read what all partitions leave)
    if (rc) return rc;
    rc = launch_stats(h, h->d_vf, h->B, h->Lv, false, sm, paired ? &pair.half[1] : nullptr);
    if (rc) return rc;
  }
  if (h->overlap) HIPCHK(hipEventRecord(h->ev_join, h->stream2));
  if (n > 0) {
    rc = launch_stats(h, d_letters, n, L, true, nullptr, paired ? &pair.half[0] : nullptr);
    if (rc) return rc;
    if (paired) {
      rc = launch_reduce_pair(h, pair, publish);
      if (rc) return rc;
    }
  } else {   // a rank may own no rows of a short last mini-batch: contribute zeros
    HIPCHK(hipMemsetAsync(h->d_sums + h->sl.data_off, 0, (size_t)(h->sl.n_d + 1 - h->sl.data_off) * sizeof(float), h->stream));
  }
  if (h->overlap) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  return CRBM_OK;
}

// ---- IPC all-reduce (update_tables_ipc_body) -------------------------------------------------
// A rank's buffer: [2 parities][8 source ranks][ipc_stride floats] of sums, then [2][8] flag words on 64-byte lines
// of their own, then the status word
size_t ipc_bytes(const crbm_handle* h) { return ((size_t)2 * IPC_MAX_RANKS * h->ipc_stride + (2 * IPC_MAX_RANKS + 1) * 16) * sizeof(float); }
float* ipc_slot_of(void* base, const crbm_handle* h, int parity, int source) {
  return static_cast<float*>(base) + ((size_t)parity * IPC_MAX_RANKS + source) * h->ipc_stride;
}
uint32_t* ipc_flag_of(void* base, const crbm_handle* h, int parity, int source) {
  return reinterpret_cast<uint32_t*>(static_cast<float*>(base) + (size_t)2 * IPC_MAX_RANKS * h->ipc_stride) + 16 * (parity * IPC_MAX_RANKS + source);
}
uint32_t* ipc_status_of(void* base, const crbm_handle* h) {
  return reinterpret_cast<uint32_t*>(static_cast<float*>(base) + (size_t)2 * IPC_MAX_RANKS * h->ipc_stride) + 16 * 2 * IPC_MAX_RANKS;
}

int ipc_allocate(crbm_handle* h) {
  if (h->ipc_buf) return CRBM_OK;
  h->ipc_stride = (h->sl.count + 31) & ~31;
  // fine-grained device memory: coherent between agents, so a peer's system-scope stores are what this GPU's
  // system-scope loads see.  Plain (coarse-grained) device memory may keep stale lines of the rank's own buffer in
  // L2 after a peer's write over xGMI: refused unless CRBM_IPC_COARSE_OK=1 asks for it (one-GPU experiments).
  void* p = nullptr;
  const hipError_t fe = hipExtMallocWithFlags(&p, ipc_bytes(h), hipDeviceMallocFinegrained);
  if (fe != hipSuccess) {
    (void)hipGetLastError();
    if (env_int("CRBM_IPC_COARSE_OK", 0) == 0)
      return fail(h, CRBM_ERR_HIP, std::string("hipExtMallocWithFlags(hipDeviceMallocFinegrained) for the mapped sums buffer: ") + hipGetErrorString(fe));
    HIPCHK(hipMalloc(&p, ipc_bytes(h)));
  }
  HIPCHK(hipMemset(p, 0, ipc_bytes(h)));
  h->ipc_buf = static_cast<float*>(p);
  return CRBM_OK;
}

// this rank's sums of the step -> its slot in every rank's buffer + flags; then the update that sums the slots of its own buffer
int launch_ipc_allreduce_update(crbm_handle* h, int L_data) {
  const int parity = (int)(h->ipc_step & 1u);
  const uint32_t value = h->ipc_step + 1u;
  if (!h->ipc_published) {
    // the sums of this step were formed in d_sums (separate reductions of the two halves, or a rank without data
    // rows): push them and raise the flags in a launch of its own
    PublishArgs pa;
    pa.src = h->d_sums; pa.value = value; pa.count = h->sl.count; pa.n = h->nranks;
    for (int r = 0; r < IPC_MAX_RANKS; ++r) {
      void* peer = r < h->nranks ? h->ipc_peer[r] : h->ipc_buf;
      pa.dst[r] = ipc_slot_of(peer, h, parity, h->rank);
      pa.flag[r] = ipc_flag_of(peer, h, parity, h->rank);
    }
    hipLaunchKernelGGL(publish_sums_kernel, dim3(1), dim3(1024), 0, h->stream, pa);
    HIPCHK(hipGetLastError());
  }
  h->ipc_published = false;
  UpdateIpcArgs a;
  fill_update_args(h, L_data, a.ut);
  for (int r = 0; r < IPC_MAX_RANKS; ++r) {            // everything the update reads is in this rank's own buffer
    const int src = r < h->nranks ? r : 0;
    a.ipc.sums[r] = ipc_slot_of(h->ipc_buf, h, parity, src);
    a.ipc.flags[r] = ipc_flag_of(h->ipc_buf, h, parity, src);
  }
  a.ipc.status = ipc_status_of(h->ipc_buf, h);
  a.ipc.expect = value; a.ipc.nranks = h->nranks; a.ipc.count = h->sl.count;
  a.ipc.timeout_ticks = h->ipc_timeout_ticks;
  const unsigned grid = (unsigned)std::max(1, std::min((h->ms.TABLES_ALL + 4095) / 4096, 32));
  // the new parameters and one word (update_tables_ipc_body): the published sums are read where they lie, not staged
  const unsigned lds = (unsigned)((h->KAM + h->K + 4 + 4) * 4);
  HIPCHK(jit_launch(h->jk.update_tables_ipc, a, grid, 1, UPDATE_THREADS, lds, h->stream));
  swap_param_sets(h);
  h->ipc_step += 1;
  return CRBM_OK;
}

int train_core(crbm_handle* h, const uint32_t* d_letters, int n, int L) {
  int rc = train_local_dev(h, d_letters, n, L, true);
  if (rc) return rc;
  if (h->ipc_on) return launch_ipc_allreduce_update(h, L);
  if (h->comm) {
    ncclResult_t r = g_rccl.AllReduce(h->d_sums, h->d_sums, (size_t)h->sl.count, ncclFloat, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) return fail(h, CRBM_ERR_RCCL, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  }
  return launch_update(h, L);
}

// The status word of the mapped-buffer all-reduce, read at a point where the stream is idle anyway: a wait for a
// peer that ran out (update_tables_ipc_body) is an error of every training entry point and of crbm_sync from then on.
int ipc_check(crbm_handle* h) {
  if (!h->ipc_on || !h->ipc_buf) return CRBM_OK;
  uint32_t st = 0;
  HIPCHK(hipMemcpy(&st, ipc_status_of(h->ipc_buf, h), sizeof(st), hipMemcpyDeviceToHost));
  if (st != 0)
    return fail(h, CRBM_ERR_IPC_TIMEOUT, "mapped-buffer all-reduce: waited " + std::to_string(env_int("CRBM_IPC_TIMEOUT_MS", 30000)) +
                " ms in vain for a peer's statistic sums (a rank died or fell out of step); no update has been applied since, "
                "the parameters are those of the last complete step");
  return CRBM_OK;
}

int check_data_shape(crbm_handle* h, int n, int L) {
  ARGCHK(n >= 1, "n must be positive");
  ARGCHK(L >= h->M, "sequence length must be >= motif_length");
  ARGCHK((long)L * 4 < (1 << 20), "sequence too long");
  // the reference reshapes the hidden layer into (…, Lh / pooling, pooling) (convRBM.py:250-252); fit() truncates for it
  ARGCHK((L - h->M + 1) % h->ms.POOL == 0, "hidden length L - motif_length + 1 must be a multiple of pooling");
  return CRBM_OK;
}

int copy_out(crbm_handle* h, float* host, const float* dev, size_t count) {
  HIPCHK(hipMemcpyAsync(host, dev, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  return CRBM_OK;
}

}  // namespace

// =============================================================================
extern "C" {

int crbm_abi_version(void) { return CRBM_AMD_ABI_VERSION; }

int crbm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* crbm_last_error(const crbm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"

namespace {

int validate_config(const crbm_config* cfg) {
  crbm_handle* h = nullptr;   // fail() routes to the create-error slot
  if (!cfg) return fail(h, CRBM_ERR_INVALID, "null argument");
  ARGCHK(cfg->num_motifs >= 1, "Number of motifs must be positive.");
  // (beyond 256 motifs / 64 letters / what the LDS holds the generic kernels take over: crbm_create)
  ARGCHK(cfg->num_motifs <= 65536, "num_motifs > 65536 is not supported");
  ARGCHK(cfg->motif_length >= 1, "Motif length must be positive.");
  ARGCHK(cfg->input_dims >= 1, "input_dims must be positive.");
  ARGCHK(cfg->input_dims <= 64, "input_dims > 64 is not supported (v|h keeps a position's activations of all letters in LDS)");
  ARGCHK(cfg->input_dims * cfg->motif_length <= 256 * BIG_ST, "input_dims * motif_length > 2048 is not supported (statistics kernel: one accumulator per letter and filter column, 8 per thread of a 256-thread block; DNA: motif_length <= 512)");
  ARGCHK((long)cfg->num_motifs * cfg->motif_length * cfg->input_dims <= (1L << 26), "num_motifs * input_dims * motif_length too large");
  ARGCHK(cfg->pooling >= 1 && cfg->pooling <= 64, "pooling must be between 1 and 64");
  ARGCHK(cfg->fantasy_hidden_len % cfg->pooling == 0, "pooling must divide the hidden length of the fantasy chains");
  ARGCHK(cfg->batchsize >= 1, "batchsize must be positive.");
  ARGCHK(cfg->cd_k >= 1, "cd_k must be positive.");
  ARGCHK(cfg->fantasy_hidden_len >= 1 && cfg->fantasy_hidden_len <= 65536, "fantasy_hidden_len out of range");
  ARGCHK(cfg->rho > 0.f && cfg->rho < 1.f, "rho must be in (0,1)");
  ARGCHK(cfg->learning_rate > 0.f, "learning_rate must be positive.");
  ARGCHK(cfg->momentum >= 0.f && cfg->momentum < 1.f, "momentum must be between zero and one.");
  ARGCHK(cfg->lambda_rate >= 0.f, "lambda_rate must be non-negative.");
  return CRBM_OK;
}

// Gibbs launch geometry.  A thread owns 4 consecutive positions, a block owns
// tiles of S whole chains (grid-stride).  Pick (S, threads) that keeps lanes
// busy in both phases and splits the tiles evenly over the CUs.
struct GibbsGeom {
  int S, threads, grid, lds;
};

// solo: geometry of launches that only advance the chains (crbm_gibbs_steps*, the chain launch of models
// whose statistics run in kernels of their own): free of the 256-thread build of the fused variants.
GibbsGeom choose_gibbs_geometry(const ModelShape& ms, int Lf, int B, int num_cu, bool sparse, bool solo = false) {
  const int forceS = env_int("CRBM_GIBBS_S", 0), forceT = env_int("CRBM_GIBBS_THREADS", 0);
  GibbsGeom best{1, 256, 1, 0};
  double best_score = -1.0;
  // 256 first: larger blocks (one table copy shared by up to 16 waves; the chain kernels are then compiled
  // with that launch bound) win only where they score strictly better -- models whose tables leave room
  // for fewer than four 256-thread blocks per CU.  The fused statistics variants are built for 256.
  double best_occupancy = 0.0;
  for (int threads : {256, 128, 64, 512, 1024}) {
    // measured: config #4 (two 256-thread blocks per CU) 2.62 -> 2.26 ms per launch with one 1024-thread
    // block; config #5 (three blocks per CU) is no faster with two 512-thread blocks -- so only when the
    // small blocks reach at most half the waves
    // Solo launches of single-stranded models take the large blocks whenever they score better (config #2:
    // one 1024-thread block of 32 chains per CU 22.7 us per launch against 23.8 with four 256-thread blocks:
    // one table copy per CU, and the short last round of the v|h pass spreads over all SIMDs).  Double-stranded
    // models too since they gather from one table and their queue of undecided units holds 254 entries
    // (config #5: 158 us per launch with three 256-thread blocks of 3 chains per CU, 145 with two 512-thread
    // blocks of 8, 143.6 with one 1024-thread block of 16 -- 147 when the queue overflows at 62 entries).
    const bool big_ok = solo ? true : (!ms.FUSE_STATS && best_occupancy <= 0.5);
    // (experiment: CRBM_FUSED_THREADS=512 with CRBM_JIT_DEFINES=-DCRBM_FUSED_TB=512 gives the fused training launch larger blocks)
    const int fusedT = (!solo && ms.FUSE_STATS) ? env_int("CRBM_FUSED_THREADS", 0) : 0;
    if (fusedT > 0 && threads != fusedT) continue;
    if (fusedT == 0 && threads > 256 && (!big_ok || env_int("CRBM_GIBBS_MAX_THREADS", 1024) < threads)) continue;
    if (forceT > 0 && threads != forceT) continue;
    for (int S = 1; S <= std::min(B, 64); ++S) {
      if (forceS > 0 && S != forceS) continue;
      const GibbsLayout gl = gibbs_layout(ms, Lf, S, sparse);
      if (gl.lds_bytes > 150 * 1024 && !(forceS > 0)) continue;
      if (gl.lds_bytes > 160 * 1024 || (long)S * gl.Lrow * ms.NW >= (1 << 20)) continue;
      // lanes are used at wave granularity (an idle wave of a pass costs nothing);
      // 4 hidden positions cost ~1.5x a 4-position visible block
      const double iv = (double)S * gl.nvb, ih = (double)S * gl.nhb;
      const double util = (iv + 0.375 * ih) / (64.0 * (std::ceil(iv / 64.0) + 0.375 * std::ceil(ih / 64.0)));
      const double ntiles = std::ceil((double)B / S);
      const double per_cu_tiles = ntiles / num_cu;
      const double balance = per_cu_tiles / std::ceil(per_cu_tiles);
      const int blocks_cu = std::max(1, std::min((160 * 1024) / gl.lds_bytes, 1024 / threads));   // <= 16 waves / CU
      const double waves_cu = std::min(per_cu_tiles, (double)blocks_cu) * std::min((double)threads, iv) / 64.0;
      const double occupancy = std::min(1.0, waves_cu / 16.0);                // 4 waves / SIMD hide the LDS latency
      const double score = util * balance * (0.4 + 0.6 * occupancy) - 1e-4 * S;
      if (score > best_score) {
        best_score = score;
        best = GibbsGeom{S, threads, (int)std::min(ntiles, (double)num_cu * blocks_cu), gl.lds_bytes};
        if (threads <= 256) best_occupancy = occupancy;
      }
    }
  }
  const int forceG = env_int("CRBM_GIBBS_GRID", 0);
  if (forceG > 0) best.grid = forceG;
  return best;
}

// Partitions of a plain chain launch (crbm_handle::chain_parts).  Worth it where a launch is short and its blocks are
// small: the partitions' kernels then share every CU (several blocks of each resident at once) and one partition's
// drain / dispatch / ramp is filled by the other's work.  Measured (two handles of half the chains, alternating launches):
// config #2 with 256-thread blocks of 8 chains 21.7 -> 17.6 us per step of the whole batch (17.4 with four partitions);
// with one 1024-thread block per CU per partition nothing (20.5 vs 20.6): the partitions then own disjoint CUs.
// In the library: config #2 20.3 -> 17.8 us, config #5 137.9 -> 136.1 us, config #4 2191 -> 2180 us (not worth a second
// geometry there); more partitions do not help (config #2: three 17.8 us against 17.4 with two, four 28 us: a process has
// four hardware queues, and partitions that share one serialise).
// Auto: two partitions when the small-block geometry of half the batch puts at least two blocks on a CU, still covers
// every CU, and a launch is short (by the number of hidden units per step); CRBM_CHAIN_PARTS forces 1..4.
struct PartPlan {
  int parts, part_chains;
  GibbsGeom geom;      // of one partition
};
PartPlan plan_chain_parts(const ModelShape& ms, int Lf, int B, int num_cu) {
  PartPlan one{1, B, GibbsGeom{0, 0, 0, 0}};
  const int forced = env_int("CRBM_CHAIN_PARTS", 0);
  if (forced == 1 || B < 2) return one;
  const int parts = forced >= 2 ? std::min(forced, 4) : 2;
  // small blocks: the geometry the fused training launch uses (at most 256 threads unless they reach half the waves)
  const int half = (B + parts - 1) / parts;
  GibbsGeom g = choose_gibbs_geometry(ms, Lf, half, num_cu, true, false);
  if (g.lds <= 0) return one;
  const int tiles = (B + g.S - 1) / g.S, tiles_part = (tiles + parts - 1) / parts;
  const int blocks_cu = std::max(1, std::min((160 * 1024) / g.lds, 1024 / g.threads));
  if (forced < 2) {
    const double items = (double)B * Lf * ms.K * (1 + ms.DS);          // hidden units per step
    if (blocks_cu < 2 || tiles_part < num_cu || items > 256e6) return one;
  }
  if (tiles_part < 1) return one;
  PartPlan p{parts, tiles_part * g.S, g};
  p.geom.grid = std::min(tiles_part, num_cu * blocks_cu);
  // (Measured and NOT adopted: twice the chains per tile where the partition's tiles do not divide over its resident blocks
  //  -- config #5: 2048 tiles of 2 chains on 768 blocks.  Forced for every kernel of the handle, CRBM_GIBBS_S=4, it runs
  //  128.6 instead of 135.2 us per step; chosen here for the partitioned launch alone 150.7: the kernel is compiled with the
  //  occupancy hint of the handle's regular geometry, three blocks per CU, and the larger tiles leave two.)
  return p;
}

// Does the model need the generic ("big") kernels?  Beyond 256 motifs or 64 letters the specialised templates do not
// exist; within them the LDS decides: the chain kernel holds its tables and at least one chain, the statistics kernel a
// column image per 16 motifs beside the gather table (at most 16 roles of 64 threads).  CRBM_FORCE_BIG=1 puts any
// model on the generic path -- the tests compare the two paths on the same model with it.
bool model_needs_big(const ModelShape& ms, int Lf, int B, int num_cu) {
  if (env_int("CRBM_FORCE_BIG", 0)) return true;
  if (ms.K > MAX_MOTIFS || ms.M > MAX_MOTIF_LENGTH) return true;
  if (choose_gibbs_geometry(ms, Lf, B, num_cu, true).lds <= 0) return true;
  for (int want_sp = 0; want_sp <= 1; ++want_sp) {
    const int tabs = ms.TAB * 4;
    const StatsMfmaLayout st = stats_mfma_layout(ms, want_sp, Lf, 0, tabs, true);
    if (st.threads > 1024 || std::max(st.region_floats * 4 + tabs, st.combine_bytes) > 160 * 1024) return true;
  }
  return false;
}

// waves per SIMD the sparse Gibbs variant reaches with its geometry; 0 when >= 4 (no hint needed)
int gibbs_block_bound(int threads) { return threads > 512 ? 1024 : threads > 256 ? 512 : 256; }

int gibbs_wpe_hint(const GibbsGeom& g) {
  if (g.lds <= 0) return 0;
  const int blocks_cu = std::max(1, std::min((160 * 1024) / g.lds, 2048 / g.threads));
  const int wpe = std::max(1, blocks_cu * (g.threads / 64) / 4);
  return wpe < 4 ? wpe : 0;
}

}  // namespace

extern "C" {

// Compile the kernels of a model into the on-disk cache without touching a GPU
// (used by __graft_entry__.build()).
int crbm_precompile(const crbm_config* cfg) {
  int rc = validate_config(cfg);
  if (rc) return rc;
  const int ds = cfg->doublestranded ? 1 : 0;
  int G = env_int("CRBM_GROUP", 0);
  if (G < 1 || G > 4) G = choose_group(cfg->num_motifs, cfg->motif_length, ds, env_int("CRBM_TABLE_BUDGET", 26 * 1024));
  const ModelShape ms = model_shape(cfg->num_motifs, cfg->motif_length, ds, G, cfg->pooling);
  std::vector<char> code;
  bool cached = false;
  std::string file, err;
  const int Lf_pc = cfg->fantasy_hidden_len > 0 ? cfg->fantasy_hidden_len : 200;
  const int ncu = env_int("CRBM_NUM_CU", 256);
  if (cfg->num_motifs > MAX_MOTIFS || cfg->motif_length > MAX_MOTIF_LENGTH || cfg->input_dims != 4 || model_needs_big(ms, Lf_pc, cfg->batchsize, ncu)) {
    // the generic kernels are compiled ahead of time; a DNA model with motifs of up to 64 letters also takes the kernels of
    // its slab model (slab_setup): those are specialised
    if (cfg->input_dims == 4 && cfg->motif_length <= MAX_MOTIF_LENGTH && env_int("CRBM_SLAB_STATS", 1) != 0) {
      int Gs = 0;
      ModelShape mss;
      const int Ks = slab_choose(cfg->num_motifs, cfg->motif_length, ds, cfg->pooling, Lf_pc, &Gs, &mss);
      if (Ks > 0 && jit_compile(Ks, cfg->motif_length, ds, Gs, Gs, mss.POOL, 0, 256, &code, &cached, &file, &err, true) != 0) {
        g_create_error = err;
        return CRBM_ERR_HIP;
      }
    }
    return CRBM_OK;
  }
  const GibbsGeom gs = choose_gibbs_geometry(ms, Lf_pc, cfg->batchsize, ncu, true);
  int tb = gs.threads;
  if (ms.DENSE) tb = std::max(tb, choose_gibbs_geometry(ms, Lf_pc, cfg->batchsize, ncu, false).threads);
  const PartPlan plan = plan_chain_parts(ms, Lf_pc, cfg->batchsize, ncu);
  if (plan.parts > 1) tb = std::max(tb, plan.geom.threads);
  int GS = plan.parts > 1 ? G : solo_group(ms.K, ms.M, ms.DS, G, ms.POOL);
  GibbsGeom solo = choose_gibbs_geometry(model_shape(ms.K, ms.M, ms.DS, GS, ms.POOL), Lf_pc, cfg->batchsize, ncu, true, true);
  if (solo.lds <= 0 && GS != G) { GS = G; solo = choose_gibbs_geometry(ms, Lf_pc, cfg->batchsize, ncu, true, true); }
  tb = std::max(tb, solo.threads);
  if (jit_compile(ms.K, ms.M, ms.DS, ms.G, GS, ms.POOL, gibbs_wpe_hint(gs), gibbs_block_bound(tb), &code, &cached, &file, &err) != 0) {
    g_create_error = err;
    return CRBM_ERR_HIP;
  }
  return CRBM_OK;
}

int crbm_create(const crbm_config* cfg, crbm_handle** out) {
  crbm_handle* h = nullptr;   // fail() routes to the create-error slot
  if (!cfg || !out) return fail(h, CRBM_ERR_INVALID, "null argument");
  *out = nullptr;
  int rc = validate_config(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(h, CRBM_ERR_NO_GPU, "no HIP device visible (this library has no CPU path)");
  ARGCHK(cfg->device >= 0 && cfg->device < ndev, "device ordinal out of range");

  crbm_handle* hh = new crbm_handle();
  hh->cfg = *cfg;
  hh->K = cfg->num_motifs; hh->M = cfg->motif_length; hh->ds = cfg->doublestranded ? 1 : 0;
  hh->A = cfg->input_dims;
  hh->KAM = hh->K * hh->A * hh->M;
  hh->Lf = cfg->fantasy_hidden_len; hh->Lv = hh->Lf + hh->M - 1; hh->B = cfg->batchsize;
  hh->seed = cfg->seed;
  hh->device = cfg->device;
  hh->sl = sums_layout(hh->K, hh->M, hh->A);
  // gather-table group size, derived shapes
  hh->G = env_int("CRBM_GROUP", 0);
  if (hh->G < 1 || hh->G > 4) hh->G = choose_group(hh->K, hh->M, hh->ds, env_int("CRBM_TABLE_BUDGET", 26 * 1024));
  hh->ms = model_shape(hh->K, hh->M, hh->ds, hh->G, cfg->pooling);
  hh->NW = hh->ms.NW;
  auto bail = [&](int code) { crbm_destroy(hh); return code; };
  hipError_t e;
#define TRY(expr)                                                                                 \
  if ((e = (expr)) != hipSuccess) {                                                               \
    g_create_error = std::string(#expr) + ": " + hipGetErrorString(e);                            \
    return bail(CRBM_ERR_HIP);                                                                    \
  }
  TRY(hipSetDevice(hh->device));
  hipDeviceProp_t prop;
  TRY(hipGetDeviceProperties(&prop, hh->device));
  hh->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    int khz = 0;      // ticks of the GPU's constant-rate clock (s_memrealtime) per millisecond
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, hh->device) == hipSuccess && khz > 0) hh->wall_khz = khz;
    (void)hipGetLastError();
  }
  // (the specialised kernels are DNA kernels: 2-bit letters, tables over letter tuples; any other alphabet is generic)
  hh->big = (hh->K > MAX_MOTIFS || hh->M > MAX_MOTIF_LENGTH || hh->A != 4) ? true : model_needs_big(hh->ms, hh->Lf, hh->B, hh->num_cu);
  if (hh->big) {
    // no specialised kernels, no table images: the state layout is the same (K-bit masks, 2-bit letters)
    hh->gl = GibbsLayout();
    hh->gl.S = 1; hh->gl.Lv = hh->Lv; hh->gl.LWs = letter_words_any(hh->A, hh->Lv);
    hh->glv[0] = hh->glv[1] = hh->gl;
    hh->variant = 1; hh->topdown_mode = 2;
    hh->GS = hh->G; hh->ms_solo = hh->ms;
    hh->chain_parts = 1; hh->part_chains = hh->B;
    hh->gridv[0] = hh->gridv[1] = 0;
  }
  // launch geometry, then the model-specific kernels (hiprtc; cached on disk)
  hh->has_dense = !hh->big && hh->ms.DENSE != 0;
  for (int v = hh->has_dense ? 0 : 1; v < 2 && !hh->big; ++v) {
    const GibbsGeom geom = choose_gibbs_geometry(hh->ms, hh->Lf, hh->B, hh->num_cu, v == 1);
    if (geom.lds <= 0) {
      g_create_error = "model too large for the LDS-resident Gibbs kernel";
      return bail(CRBM_ERR_INVALID);
    }
    hh->glv[v] = gibbs_layout(hh->ms, hh->Lf, geom.S, v == 1);
    if (v == 1) hh->gibbs_wpe = gibbs_wpe_hint(geom);
    hh->threadsv[v] = geom.threads;
    hh->gridv[v] = geom.grid;
  }
  if (!hh->big) {
    // plain chain launches: their own geometry; short launches in partitions on streams of their own (chain_parts);
    // unpartitioned ones of small fused models with their own letter grouping (solo_group)
    const PartPlan plan = plan_chain_parts(hh->ms, hh->Lf, hh->B, hh->num_cu);
    hh->chain_parts = plan.parts;
    hh->part_chains = plan.parts > 1 ? plan.part_chains : hh->B;
    if (plan.parts > 1) {
      hh->gl_part = gibbs_layout(hh->ms, hh->Lf, plan.geom.S, true);
      hh->part_threads = plan.geom.threads;
      hh->part_grid = plan.geom.grid;
    }
    // the unpartitioned launch: its own geometry and, where nothing else launches that kernel, its own letter grouping
    hh->GS = plan.parts > 1 ? hh->G : solo_group(hh->K, hh->M, hh->ds, hh->G, cfg->pooling);
    hh->ms_solo = model_shape(hh->K, hh->M, hh->ds, hh->GS, cfg->pooling);
    GibbsGeom solo = choose_gibbs_geometry(hh->ms_solo, hh->Lf, hh->B, hh->num_cu, true, true);
    if (solo.lds <= 0 && hh->GS != hh->G) {           // the larger table leaves no room for a chain: the model's grouping
      hh->GS = hh->G; hh->ms_solo = hh->ms;
      solo = choose_gibbs_geometry(hh->ms_solo, hh->Lf, hh->B, hh->num_cu, true, true);
    }
    if (solo.lds > 0 && (hh->GS != hh->G || solo.threads != hh->threadsv[1] || solo.S != hh->glv[1].S || solo.grid != hh->gridv[1])) {
      hh->gl_solo = gibbs_layout(hh->ms_solo, hh->Lf, solo.S, true);
      hh->solo_threads = solo.threads;
      hh->solo_grid = solo.grid;
    }
  }
  if (!hh->big) {
    // The set-bit walk is the variant of every model (its cost follows the hidden activity,
    // ~2 % under the reference's sparsity target); CRBM_TOPDOWN=dense pins the table variant
    // of small models for A/B runs.  Never switched at run time: the two round differently.
    const char* td = getenv("CRBM_TOPDOWN");
    hh->topdown_mode = (td && !strcmp(td, "dense") && hh->has_dense) ? 1 : 2;
    use_variant(hh, hh->topdown_mode == 1 ? 0 : 1);
  }
  if (!hh->big) {
    std::string err;
    const int tb = gibbs_block_bound(std::max(std::max(std::max(hh->has_dense ? hh->threadsv[0] : 0, hh->threadsv[1]), hh->solo_threads), hh->part_threads));
    if (jit_load(hh->K, hh->M, hh->ds, hh->G, hh->GS, hh->ms.POOL, hh->gibbs_wpe, tb, &hh->jk, &err) != 0) {
      g_create_error = "kernel specialisation failed: " + err;
      return bail(CRBM_ERR_HIP);
    }
  }
  TRY(hipStreamCreateWithFlags(&hh->stream, hipStreamNonBlocking));
  TRY(hipStreamCreateWithFlags(&hh->stream2, hipStreamNonBlocking));
  TRY(hipEventCreate(&hh->ev0));
  TRY(hipEventCreate(&hh->ev1));
  TRY(hipEventCreateWithFlags(&hh->ev_fork, hipEventDisableTiming));
  TRY(hipEventCreateWithFlags(&hh->ev_join, hipEventDisableTiming));
  if (hh->chain_parts > 1) {
    TRY(hipEventCreateWithFlags(&hh->ev_parts_fork, hipEventDisableTiming));
    TRY(hipEventCreate(&hh->ev_cal0));
    TRY(hipEventCreate(&hh->ev_cal1));
    hh->part_stream[0] = hh->stream;          // partition 0 rides on the handle's own stream: one hardware queue fewer
    for (int p = 1; p < hh->chain_parts; ++p) {
      TRY(hipStreamCreateWithFlags(&hh->part_stream[p], hipStreamNonBlocking));
      TRY(hipEventCreate(&hh->part_done[p]));     // with timestamps: crbm_time_gibbs reads them
      {
        PartWorker* w = new PartWorker();
        w->device = hh->device;
        w->stream = hh->part_stream[p];
        w->th = std::thread(part_worker_main, w);
        hh->part_worker[p] = w;
      }
    }
  }
  hh->overlap = env_int("CRBM_OVERLAP", 0) != 0;   // measured: no gain once the statistics kernel fills the chip
  // (generic handles: W and b carry ten motifs of zero padding behind motif K -- the last slab of the slabbed kernels
  //  starts on a multiple of ten and may reach past the model's end, slab_origin)
  const size_t pad_k = hh->big ? 10 : 0;
  const size_t kam = (size_t)hh->KAM + pad_k * hh->A * hh->M, k = (size_t)hh->K + pad_k;
  TRY(hipMalloc((void**)&hh->dW, kam * 4)); TRY(hipMalloc((void**)&hh->dvW, kam * 4));
  TRY(hipMalloc((void**)&hh->db, k * 4));   TRY(hipMalloc((void**)&hh->dvb, k * 4));
  const size_t cb = (size_t)std::max(4, hh->A) * 4;
  TRY(hipMalloc((void**)&hh->dc, cb));      TRY(hipMalloc((void**)&hh->dvc, cb));
  TRY(hipMemset(hh->dW, 0, kam * 4)); TRY(hipMemset(hh->dvW, 0, kam * 4));
  TRY(hipMemset(hh->db, 0, k * 4));   TRY(hipMemset(hh->dvb, 0, k * 4));
  TRY(hipMemset(hh->dc, 0, cb));      TRY(hipMemset(hh->dvc, 0, cb));
  TRY(hipMalloc((void**)&hh->dW2, kam * 4)); TRY(hipMalloc((void**)&hh->dvW2, kam * 4));
  TRY(hipMalloc((void**)&hh->db2, k * 4));   TRY(hipMalloc((void**)&hh->dvb2, k * 4));
  TRY(hipMalloc((void**)&hh->dc2, cb));      TRY(hipMalloc((void**)&hh->dvc2, cb));
  TRY(hipMalloc((void**)&hh->d_tables, hh->big ? 64 : (size_t)hh->ms.TABLES_ALL * 4));
  if (hh->GS != hh->G) TRY(hipMalloc((void**)&hh->d_tf_solo, (size_t)hh->ms_solo.TAB * 4));
  const size_t mwords = (size_t)hh->B * hh->Lf * hh->NW;
  TRY(hipMalloc((void**)&hh->d_hm, mwords * 4)); TRY(hipMemset(hh->d_hm, 0, mwords * 4));
  TRY(hipMalloc((void**)&hh->d_hmp, mwords * 4)); TRY(hipMemset(hh->d_hmp, 0, mwords * 4));
  const size_t vwords = (size_t)hh->B * hh->gl.LWs;
  TRY(hipMalloc((void**)&hh->d_vf, vwords * 4)); TRY(hipMemset(hh->d_vf, 0, vwords * 4));
  TRY(hipMalloc((void**)&hh->d_flags, 16)); TRY(hipMemset(hh->d_flags, 0, 16));
  TRY(hipMalloc((void**)&hh->d_ones, 16)); TRY(hipMemset(hh->d_ones, 0, 16));
  {
    const size_t slots = (size_t)std::max(std::max(hh->gridv[0] * (hh->threadsv[0] / 64), hh->gridv[1] * (hh->threadsv[1] / 64)), std::max(hh->solo_grid * (hh->solo_threads / 64), hh->chain_parts * hh->part_grid * (hh->part_threads / 64))) + 64;
    TRY(hipMalloc((void**)&hh->d_nset, slots * 4)); TRY(hipMemset(hh->d_nset, 0, slots * 4));
  }
  TRY(hipMalloc((void**)&hh->d_sums, (size_t)hh->sl.count * 4)); TRY(hipMemset(hh->d_sums, 0, (size_t)hh->sl.count * 4));
  TRY(hipMalloc((void**)&hh->d_ticket, 16)); TRY(hipMemset(hh->d_ticket, 0, 16));
  TRY(hipMalloc((void**)&hh->d_probe, 32)); TRY(hipMemset(hh->d_probe, 0, 32));
#undef TRY
  hh->tables_dirty = true;
  {
    const char* sv = getenv("CRBM_STATS");
    hh->fuse_stats = !hh->big && hh->ms.FUSE_STATS && hh->variant == 1 && !(sv && !strcmp(sv, "split"));
    hh->one_launch = !(sv && !strcmp(sv, "two"));
    if (hh->fuse_stats) {   // the fused launch appends the statistics slices to the chain image: it must fit the LDS
      const StatsMfmaLayout st = stats_mfma_layout(hh->ms, 0, hh->Lf, hh->gibbs_threads, 0, false);
      const int lds = std::max((((hh->gl.lds_bytes / 4 + 3) & ~3) + st.region_floats) * 4, st.combine_bytes);
      if (lds > 160 * 1024) hh->fuse_stats = false;
    }
  }
  hh->stats_rows = env_int("CRBM_STATS_ROWS", 0);   // 0: one resident wave of blocks
  slab_setup(hh);                                   // generic DNA models: their statistics on the matrix cores, slab by slab
  *out = hh;
  return CRBM_OK;
}

int crbm_destroy(crbm_handle* h) {
  if (!h) return CRBM_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  for (int p = 0; p < 4; ++p) {
    if (PartWorker* w = h->part_worker[p]) {
      {
        std::lock_guard<std::mutex> lock(w->mu);
        w->stop = true;
      }
      w->cv.notify_one();
      if (w->th.joinable()) w->th.join();
      delete w;
      h->part_worker[p] = nullptr;
    }
    if (p > 0 && h->part_stream[p]) { (void)hipStreamSynchronize(h->part_stream[p]); (void)hipStreamDestroy(h->part_stream[p]); }
    if (h->part_done[p]) (void)hipEventDestroy(h->part_done[p]);
  }
  if (h->ev_parts_fork) (void)hipEventDestroy(h->ev_parts_fork);
  if (h->ev_cal0) (void)hipEventDestroy(h->ev_cal0);
  if (h->ev_cal1) (void)hipEventDestroy(h->ev_cal1);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  for (int r = 0; r < IPC_MAX_RANKS; ++r)
    if (h->ipc_peer[r] && h->ipc_peer[r] != h->ipc_buf) (void)hipIpcCloseMemHandle(h->ipc_peer[r]);
  if (h->ipc_buf) (void)hipFree(h->ipc_buf);
  slab_destroy(h);
  void* ptrs[] = {h->dW2, h->db2, h->dc2, h->dvW2, h->dvb2, h->dvc2, h->dW, h->db, h->dc, h->dvW, h->dvb, h->dvc, h->d_hm, h->d_hmp, h->d_vf, h->d_flags, h->d_ones, h->d_nset, h->d_sums, h->d_ticket, h->d_tables, h->d_tf_solo, h->d_probe, h->d_timeline};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  h->stage.release(); h->stage2.release(); h->out_a.release(); h->out_b.release(); h->out_c.release();
  h->letters.release(); h->letters2.release(); h->masks_tmp.release(); h->out_a2.release(); h->out_b2.release();
  for (auto& d : h->dataset) d.release(); h->partials.release(); h->partials2.release();
  if (h->jk.module) (void)hipModuleUnload(h->jk.module);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return CRBM_OK;
}

// ENTER_ASYNC: entry points that may leave work on the partition streams (plain chain launches); every other entry point
// first lets the main stream wait for them (join_parts: two event calls per partition, only when something is pending)
#define ENTER_ASYNC()                                         \
  if (!h) return CRBM_ERR_INVALID;                            \
  h->err.clear();                                             \
  HIPCHK(hipSetDevice(h->device))
#define ENTER()                                               \
  ENTER_ASYNC();                                              \
  do {                                                        \
    const int jrc__ = join_parts(h);                          \
    if (jrc__) return jrc__;                                  \
  } while (0)

int crbm_set_params(crbm_handle* h, const float* W, const float* b, const float* c) {
  ENTER();
  ARGCHK(W && b && c, "null argument");
  HIPCHK(hipMemcpyAsync(h->dW, W, (size_t)h->KAM * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->db, b, (size_t)h->K * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->dc, c, (size_t)h->A * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->tables_dirty = true;
  h->params_version += 1;
  return CRBM_OK;
}

int crbm_get_params(crbm_handle* h, float* W, float* b, float* c) {
  ENTER();
  ARGCHK(W && b && c, "null argument");
  HIPCHK(hipMemcpyAsync(W, h->dW, (size_t)h->KAM * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(b, h->db, (size_t)h->K * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(c, h->dc, (size_t)h->A * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_set_velocities(crbm_handle* h, const float* vW, const float* vb, const float* vc) {
  ENTER();
  ARGCHK(vW && vb && vc, "null argument");
  HIPCHK(hipMemcpyAsync(h->dvW, vW, (size_t)h->KAM * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->dvb, vb, (size_t)h->K * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->dvc, vc, (size_t)h->A * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_get_velocities(crbm_handle* h, float* vW, float* vb, float* vc) {
  ENTER();
  ARGCHK(vW && vb && vc, "null argument");
  HIPCHK(hipMemcpyAsync(vW, h->dvW, (size_t)h->KAM * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(vb, h->dvb, (size_t)h->K * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(vc, h->dvc, (size_t)h->A * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_set_fantasy(crbm_handle* h, const float* hid, const float* hid_prime) {
  ENTER();
  ARGCHK(hid, "null argument");
  ARGCHK(!h->ds || hid_prime, "doublestranded model needs hid_prime");
  const size_t count = (size_t)h->B * h->K * h->Lf;
  HIPCHK(h->stage.ensure(count));
  const float* src[2] = {hid, hid_prime};
  uint32_t* dst[2] = {h->d_hm, h->d_hmp};
  for (int sidx = 0; sidx < 1 + h->ds; ++sidx) {
    HIPCHK(hipMemcpyAsync(h->stage.p, src[sidx], count * 4, hipMemcpyHostToDevice, h->stream));
    HiddenPackArgs a;
    a.dense = h->stage.p; a.masks = dst[sidx]; a.flags = h->d_flags;
    a.n = h->B; a.K = h->K; a.Lh = h->Lf; a.NW = h->NW;
    hipLaunchKernelGGL(pack_hidden_kernel, dim3(grid_for((long)h->B * h->Lf, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));   // staging buffer is reused
  }
  return check_flags(h);
}

int crbm_get_fantasy(crbm_handle* h, float* hid, float* hid_prime) {
  ENTER();
  ARGCHK(hid, "null argument");
  const size_t count = (size_t)h->B * h->K * h->Lf;
  HIPCHK(h->stage.ensure(count));
  float* dsth[2] = {hid, hid_prime};
  uint32_t* src[2] = {h->d_hm, h->d_hmp};
  for (int sidx = 0; sidx < 1 + h->ds; ++sidx) {
    if (!dsth[sidx]) continue;
    HiddenPackArgs a;
    a.dense = h->stage.p; a.masks = src[sidx]; a.flags = h->d_flags;
    a.n = h->B; a.K = h->K; a.Lh = h->Lf; a.NW = h->NW;
    hipLaunchKernelGGL(unpack_hidden_kernel, dim3(grid_for((long)h->B * h->Lf, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dsth[sidx], h->stage.p, count * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return CRBM_OK;
}

int crbm_get_fantasy_visible(crbm_handle* h, float* v) {
  ENTER();
  ARGCHK(v, "null argument");
  const size_t count = (size_t)h->B * h->A * h->Lv;
  HIPCHK(h->stage.ensure(count));
  DecodeArgs a;
  a.letters = h->d_vf; a.v = h->stage.p; a.n = h->B; a.L = h->Lv; a.LW = h->gl.LWs; a.A = h->A;
  if (h->A == 4) hipLaunchKernelGGL(decode_onehot_kernel, dim3(grid_for((long)h->B * h->Lv, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(decode_onehot_any_kernel, dim3(grid_for((long)h->B * h->Lv, 256, h->num_cu * 8)), dim3(256), 0, h->stream, a);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(v, h->stage.p, count * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_set_rng(crbm_handle* h, uint64_t seed, uint32_t gibbs_step, uint32_t eval_step) {
  if (!h) return CRBM_ERR_INVALID;
  h->seed = seed; h->gibbs_step = gibbs_step; h->eval_step = eval_step;
  return CRBM_OK;
}

int crbm_get_rng(crbm_handle* h, uint64_t* seed, uint32_t* gibbs_step, uint32_t* eval_step) {
  if (!h) return CRBM_ERR_INVALID;
  if (seed) *seed = h->seed;
  if (gibbs_step) *gibbs_step = h->gibbs_step;
  if (eval_step) *eval_step = h->eval_step;
  return CRBM_OK;
}

int crbm_set_shard(crbm_handle* h, uint32_t chain_offset) {
  if (!h) return CRBM_ERR_INVALID;
  h->chain_offset = chain_offset;
  return CRBM_OK;
}

// ---- training ----------------------------------------------------------------
int crbm_train_step(crbm_handle* h, const float* D, int32_t n, int32_t L) {
  ENTER();
  ARGCHK(D, "null argument");
  int rc = check_data_shape(h, n, L);
  if (rc) return rc;
  HIPCHK(h->letters.ensure((size_t)n * lw(h, L)));
  rc = encode_host(h, D, n, L, h->letters.p);
  if (rc) return rc;
  rc = train_core(h, h->letters.p, n, L);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return ipc_check(h);
}

int crbm_dataset_select(crbm_handle* h, int32_t slot) {
  ENTER();
  ARGCHK(slot >= 0 && slot < CRBM_DATASET_SLOTS, "no such data-set slot");
  h->slot = slot;
  return CRBM_OK;
}

int crbm_dataset_upload(crbm_handle* h, const float* data, int32_t n, int32_t L) {
  ENTER();
  ARGCHK(data, "null argument");
  int rc = check_data_shape(h, n, L);
  if (rc) return rc;
  const int LW = lw(h, L);
  DevBuf<uint32_t>& ds = h->dataset[h->slot];
  h->dataset_n[h->slot] = 0;
  HIPCHK(ds.ensure((size_t)n * LW));
  // stream the fp32 array through the staging buffer in slabs of <= 64 MiB
  const int slab = std::max(1, (int)std::min<long>(n, (64L << 20) / ((long)h->A * L * 4)));
  for (int start = 0; start < n; start += slab) {
    const int cnt = std::min(slab, n - start);
    rc = encode_host(h, data + (size_t)start * h->A * L, cnt, L, ds.p + (size_t)start * LW);
    if (rc) return rc;
  }
  h->dataset_n[h->slot] = n; h->dataset_L[h->slot] = L;
  return CRBM_OK;
}

int crbm_dataset_upload_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L) {
  ENTER();
  ARGCHK(codes, "null argument");
  int rc = check_data_shape(h, n, L);
  if (rc) return rc;
  const int LW = lw(h, L);
  DevBuf<uint32_t>& ds = h->dataset[h->slot];
  h->dataset_n[h->slot] = 0;
  HIPCHK(ds.ensure((size_t)n * LW));
  const int slab = std::max(1, (int)std::min<long>(n, (256L << 20) / (long)L));
  for (int start = 0; start < n; start += slab) {
    const int cnt = std::min(slab, n - start);
    rc = encode_codes_host(h, codes + (size_t)start * L, cnt, L, ds.p + (size_t)start * LW);
    if (rc) return rc;
  }
  h->dataset_n[h->slot] = n; h->dataset_L[h->slot] = L;
  return CRBM_OK;
}

int crbm_train_step_resident(crbm_handle* h, int32_t start, int32_t end) {
  ENTER();
  const int slot = h->slot;
  ARGCHK(h->dataset_n[slot] > 0, "no resident data set (call crbm_dataset_upload)");
  ARGCHK(start >= 0 && end >= start && end <= h->dataset_n[slot], "row range out of bounds");
  ARGCHK(end > start || h->comm || h->ipc_on, "empty row range");
  const int LW = lw(h, h->dataset_L[slot]);
  int rc = train_core(h, h->dataset[slot].p + (size_t)start * LW, end - start, h->dataset_L[slot]);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return ipc_check(h);
}

// One pass over the resident data set in sequential mini-batches of `batchsize`
// rows (the slices of convRBM.py:722-726, last one short), each a full PCD-k
// update; rank r of a communicator takes its contiguous share of every batch.
// The steps are enqueued back to back: one host synchronisation per epoch
// instead of one per step.
int crbm_train_epoch_resident(crbm_handle* h, int32_t batchsize) {
  ENTER();
  const int slot = h->slot;
  ARGCHK(h->dataset_n[slot] > 0, "no resident data set (call crbm_dataset_upload)");
  ARGCHK(batchsize >= 1, "batchsize must be positive");
  const int total = h->dataset_n[slot], L = h->dataset_L[slot], LW = lw(h, L);
  for (int start = 0; start < total; start += batchsize) {
    const int end = std::min(total, start + batchsize), n = end - start;
    const int lo = start + (int)(((long)n * h->rank) / h->nranks), hi = start + (int)(((long)n * (h->rank + 1)) / h->nranks);
    int rc = train_core(h, h->dataset[slot].p + (size_t)lo * LW, hi - lo, L);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return ipc_check(h);
}

// The same loop when the selected slot holds only THIS rank's rows: for every
// slice [start, end) of the global data set of `total_rows` rows, rank r owns rows
// [start + n*r/R, start + n*(r+1)/R) (n = end - start) and the resident set is their
// concatenation in slice order (crbm_amd.dist.shard_rows builds it), so a rank
// uploads 1/R of the data instead of all of it.
int crbm_train_epoch_sharded(crbm_handle* h, int32_t batchsize, int32_t total_rows, int32_t L) {
  ENTER();
  const int slot = h->slot;
  ARGCHK(batchsize >= 1 && total_rows >= 1, "batchsize and total_rows must be positive");
  // L comes from the caller, not from this rank's slot: a rank with an empty share has nothing resident,
  // and all ranks must divide the all-reduced sums by the same counts
  int rc = check_data_shape(h, 1, L);
  if (rc) return rc;
  const int LW = lw(h, L);
  long have = 0;
  for (int start = 0; start < total_rows; start += batchsize) {
    const long n = std::min(total_rows, start + batchsize) - start;
    have += (n * (h->rank + 1)) / h->nranks - (n * h->rank) / h->nranks;
  }
  ARGCHK(have == h->dataset_n[slot], "resident rows do not match this rank's share of total_rows");
  ARGCHK(have == 0 || h->dataset_L[slot] == L, "resident rows have another sequence length than L");
  ARGCHK(have > 0 || h->comm || h->ipc_on, "no resident data set (call crbm_dataset_upload)");
  size_t off = 0;
  for (int start = 0; start < total_rows; start += batchsize) {
    const long n = std::min(total_rows, start + batchsize) - start;
    const int mine = (int)((n * (h->rank + 1)) / h->nranks - (n * h->rank) / h->nranks);
    rc = train_core(h, h->dataset[slot].p + off * LW, mine, L);
    if (rc) return rc;
    off += (size_t)mine;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return ipc_check(h);
}

int crbm_gibbs_steps_async(crbm_handle* h, int32_t k) {
  ENTER_ASYNC();
  ARGCHK(k >= 1, "k must be positive");
  return launch_gibbs(h, k, nullptr, nullptr, true);
}

int crbm_sync(crbm_handle* h) {
  ENTER();
  HIPCHK(hipStreamSynchronize(h->stream));
  int rc = ipc_check(h);
  if (rc) return rc;
  return refresh_activity(h);
}

// every launch of this handle has completed (no read-back of the activity monitor: crbm_sync does that too)
int crbm_wait_idle(crbm_handle* h) {
  ENTER();
  HIPCHK(hipStreamSynchronize(h->stream));
  return ipc_check(h);
}

int crbm_gibbs_steps(crbm_handle* h, int32_t k) {
  int rc = crbm_gibbs_steps_async(h, k);
  if (rc) return rc;
  return crbm_sync(h);
}

// waits for an event by polling it: the timed entry points return microseconds after their last launch has completed
// (a blocking wait wakes up tens of microseconds late, which a 20-launch benchmark run would see in its host clock)
static hipError_t spin_until(hipEvent_t ev) {
  for (long i = 0;; ++i) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (i > 2000000) return hipEventSynchronize(ev);   // seconds have passed: nothing left to gain from polling, block like everybody else
  }
}

int crbm_time_gibbs(crbm_handle* h, int32_t k, int32_t launches, float* total_ms) {
  ENTER();
  ARGCHK(k >= 0 && launches >= 1 && total_ms, "bad argument");   // k = 0: state load/store only (profiling)
  // (the launches add their durations to d_probe; the sums before this call were read when the previous one ended:
  //  nothing touches the GPU here before the opening event)
  h->probe_base[0] = h->probe_last[0]; h->probe_base[1] = h->probe_last[1];
  h->probe_on = true;
  const bool want_timeline = env_int("CRBM_GIBBS_TIMELINE", 0) != 0 && launches <= 4096;
  if (want_timeline) {
    const int need = launches * std::max(1, h->chain_parts);
    if (need > h->timeline_cap) {
      if (h->d_timeline) (void)hipFree(h->d_timeline);
      h->d_timeline = nullptr;
      HIPCHK(hipMalloc((void**)&h->d_timeline, (size_t)need * 16));
      h->timeline_cap = need;
    }
    HIPCHK(hipMemset(h->d_timeline, 0, (size_t)h->timeline_cap * 16));
    h->timeline_next = 0;
  } else {
    h->timeline_cap = h->d_timeline ? h->timeline_cap : 0;
    h->timeline_next = h->timeline_cap;      // no slots handed out
  }
  struct TimelineDump {
    crbm_handle* h; bool on;
    ~TimelineDump() {
      if (!on || !h->d_timeline) return;
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> t((size_t)h->timeline_next * 2);
      if (hipMemcpy(t.data(), h->d_timeline, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess || t.empty()) return;
      unsigned long long t0 = ~0ull;
      for (size_t i = 0; i < t.size(); i += 2) if (t[i]) t0 = std::min(t0, t[i]);
      const double us = 1000.0 / h->wall_khz;
      fprintf(stderr, "chain launches (block 0: start-end in us from the first start; %d partition(s), launch order):", h->chain_parts);
      for (size_t i = 0; i < t.size(); i += 2) {
        if (i == 2 * 60 && t.size() > 2 * 120) { fprintf(stderr, " ..."); i = t.size() - 2 * 60; }
        fprintf(stderr, " %.1f-%.1f", (t[i] - t0) * us, (t[i + 1] - t0) * us);
      }
      fprintf(stderr, "\n");
      h->timeline_next = h->timeline_cap;
    }
  } timeline_dump{h, want_timeline};
  struct ProbeOff { crbm_handle* h; ~ProbeOff() { h->probe_on = false; } } probe_off{h};
  h->main_idle_hint = hipStreamQuery(h->stream) == hipSuccess;   // (the opening event below does not count as work to wait for)
  (void)hipGetLastError();
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  const auto t_enq0 = std::chrono::steady_clock::now();
  const bool trace_enq = env_int("CRBM_TIME_ENQUEUE", 0) >= 2;
  std::vector<double> enq_us;
  for (int i = 0; i < launches; ++i) {
    int rc = launch_gibbs(h, k, nullptr, nullptr, true);
    if (rc) return rc;
    if (trace_enq) enq_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq0).count());
  }
  if (trace_enq) {
    fprintf(stderr, "enqueue completion times (us):");
    for (size_t i = 0; i < enq_us.size() && i < 60; ++i) fprintf(stderr, " %.0f", enq_us[i]);
    fprintf(stderr, "\n");
  }
  if (env_int("CRBM_TIME_ENQUEUE", 0))     // measurement aid: host time spent enqueuing the launches
    fprintf(stderr, "crbm_time_gibbs: %d launch steps enqueued in %.1f us of host time\n", launches,
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq0).count());
  if (h->forked) {
    // partitioned launches: the interval ends when the last partition's last launch does -- an event behind each
    // partition's launches, polled by the host (no device-side join in the timed region; the streams stay forked)
    float worst = 0.f;
    {
      const int rc = drain_part_workers(h);
      if (rc) return rc;
    }
    HIPCHK(hipEventRecord(h->ev1, h->stream));                       // partition 0 = the main stream
    for (int p = 1; p < h->chain_parts; ++p) HIPCHK(hipEventRecord(h->part_done[p], h->part_stream[p]));
    HIPCHK(spin_until(h->ev1));
    HIPCHK(hipEventElapsedTime(&worst, h->ev0, h->ev1));
    for (int p = 1; p < h->chain_parts; ++p) {
      float ms = 0.f;
      HIPCHK(spin_until(h->part_done[p]));
      HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->part_done[p]));
      worst = std::max(worst, ms);
    }
    *total_ms = worst;
    return CRBM_OK;
  }
  h->main_idle_hint = false;
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(spin_until(h->ev1));
  HIPCHK(hipEventElapsedTime(total_ms, h->ev0, h->ev1));
  return CRBM_OK;
}

int crbm_time_train(crbm_handle* h, int32_t start, int32_t end, int32_t launches, float* total_ms) {
  ENTER();
  const int slot = h->slot;
  ARGCHK(h->dataset_n[slot] > 0, "no resident data set (call crbm_dataset_upload)");
  ARGCHK(start >= 0 && end > start && end <= h->dataset_n[slot] && launches >= 1 && total_ms, "bad argument");
  const int LW = lw(h, h->dataset_L[slot]);
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < launches; ++i) {
    int rc = train_core(h, h->dataset[slot].p + (size_t)start * LW, end - start, h->dataset_L[slot]);
    if (rc) return rc;
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(spin_until(h->ev1));
  HIPCHK(hipEventElapsedTime(total_ms, h->ev0, h->ev1));
  return ipc_check(h);
}

// ---- stand-alone passes ------------------------------------------------------
int crbm_h_given_v(crbm_handle* h, const float* v, int32_t n, int32_t L, int32_t flip, uint32_t rng_step,
                   float* act, float* prob, float* sample) {
  ENTER();
  ARGCHK(v, "null argument");
  int rc = check_data_shape(h, n, L);
  if (rc) return rc;
  const int Lh = L - h->M + 1;
  const size_t count = (size_t)n * h->K * Lh;
  HIPCHK(h->letters.ensure((size_t)n * lw(h, L)));
  rc = encode_host(h, v, n, L, h->letters.p);
  if (rc) return rc;
  if (act) HIPCHK(h->out_a.ensure(count));
  if (prob) HIPCHK(h->out_b.ensure(count));
  if (sample) HIPCHK(h->out_c.ensure(count));
  rc = launch_hgv(h, h->letters.p, n, L, flip ? 1 : 0, act ? h->out_a.p : nullptr, prob ? h->out_b.p : nullptr,
                  sample ? h->out_c.p : nullptr, nullptr, KIND_API_H, rng_step, h->chain_offset);
  if (rc) return rc;
  if (act && (rc = copy_out(h, act, h->out_a.p, count))) return rc;
  if (prob && (rc = copy_out(h, prob, h->out_b.p, count))) return rc;
  if (sample && (rc = copy_out(h, sample, h->out_c.p, count))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_v_given_h(crbm_handle* h, const float* hid, const float* hid_prime, int32_t n, int32_t Lh,
                   uint32_t rng_step, float* act, float* prob, float* sample) {
  ENTER();
  ARGCHK(hid, "null argument");
  ARGCHK(n >= 1 && Lh >= 1, "bad shape");
  const int L = Lh + h->M - 1;
  ARGCHK((long)L * 4 < (1 << 20), "sequence too long");
  const size_t hcount = (size_t)n * h->K * Lh, vcount = (size_t)n * h->A * L;
  HIPCHK(h->stage.ensure(hcount));
  HIPCHK(hipMemcpyAsync(h->stage.p, hid, hcount * 4, hipMemcpyHostToDevice, h->stream));
  if (hid_prime) {
    HIPCHK(h->stage2.ensure(hcount));
    HIPCHK(hipMemcpyAsync(h->stage2.p, hid_prime, hcount * 4, hipMemcpyHostToDevice, h->stream));
  }
  if (act) HIPCHK(h->out_a.ensure(vcount));
  if (prob) HIPCHK(h->out_b.ensure(vcount));
  if (sample) HIPCHK(h->out_c.ensure(vcount));
  VghArgs a;
  a.W = h->dW; a.c = h->dc; a.K = h->K; a.M = h->M;
  a.hid = h->stage.p; a.hidp = hid_prime ? h->stage2.p : nullptr;
  a.n = n; a.Lh = Lh; a.L = L;
  a.TS = tile_seqs(L, 4096);
  a.divL = make_fastdiv((uint32_t)L);
  a.act = act ? h->out_a.p : nullptr; a.prob = prob ? h->out_b.p : nullptr; a.sample = sample ? h->out_c.p : nullptr;
  a.rng = rng_view(h, rng_step, h->chain_offset);
  a.kind = KIND_API_V;
  const int ntiles = (n + a.TS - 1) / a.TS;
  if (h->A == 4) {
    hipLaunchKernelGGL(vgh_dense_kernel, dim3(std::max(1, std::min(ntiles, h->num_cu * 8))), dim3(256),
                       (size_t)h->M * h->K * 16, h->stream, a);
  } else {      // any other alphabet: a position's activations of all letters in LDS, filters from global memory
    VghAnyArgs aa;
    aa.g = a; aa.A = h->A;
    hipLaunchKernelGGL(vgh_dense_any_kernel, dim3(grid_for((long)n * L, 256, h->num_cu * 8)), dim3(256), (size_t)h->A * 256 * 4, h->stream, aa);
  }
  HIPCHK(hipGetLastError());
  int rc;
  if (act && (rc = copy_out(h, act, h->out_a.p, vcount))) return rc;
  if (prob && (rc = copy_out(h, prob, h->out_b.p, vcount))) return rc;
  if (sample && (rc = copy_out(h, sample, h->out_c.p, vcount))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

// ---- evaluation --------------------------------------------------------------
// Every evaluation entry point exists in three forms: fp32 one-hot host input
// (the reference's layout), one byte per base host input (`_codes`), and rows of
// a data set already resident in HBM (`_resident`).  Host input is streamed in
// slabs so that the staged input plus the outputs of one slab stay around
// 256 MB (data sets far larger than the scratch: SURVEY 8(f)-1).
static int slab_rows(int n, size_t bytes_per_row) {
  const size_t budget = (size_t)env_int("CRBM_SLAB_BYTES", 256 << 20);
  size_t rows = budget / std::max<size_t>(bytes_per_row, 1);
  if (rows < 1) rows = 1;
  return (int)std::min<size_t>(rows, (size_t)n);
}

namespace {

// where the letters of an evaluation call come from
struct RowSource {
  const float* onehot = nullptr;      // host (n,1,A,L) fp32
  const uint8_t* codes = nullptr;     // host (n,L) bytes 0..A-1
  const uint32_t* resident = nullptr; // device packed rows
  int n = 0, L = 0, A = 4;
  size_t in_bytes_per_row() const { return onehot ? (size_t)4 * A * L : (codes ? (size_t)L : 0); }
};

// packed letters of rows [start, start+cnt) of a source; host sources go through h->letters
int source_rows(crbm_handle* h, const RowSource& src, int start, int cnt, const uint32_t** out) {
  const int LW = lw(h, src.L);
  if (src.resident) {
    *out = src.resident + (size_t)start * LW;
    return CRBM_OK;
  }
  HIPCHK(h->letters.ensure((size_t)cnt * LW));
  *out = h->letters.p;
  if (src.onehot) return encode_host(h, src.onehot + (size_t)start * h->A * src.L, cnt, src.L, h->letters.p);
  return encode_codes_host(h, src.codes + (size_t)start * src.L, cnt, src.L, h->letters.p);
}

int resident_source(crbm_handle* h, int start, int end, RowSource* src) {
  const int slot = h->slot;
  ARGCHK(h->dataset_n[slot] > 0, "no resident data set (call crbm_dataset_upload)");
  ARGCHK(start >= 0 && end > start && end <= h->dataset_n[slot], "row range out of bounds");
  src->L = h->dataset_L[slot];
  src->n = end - start;
  src->resident = h->dataset[slot].p + (size_t)start * lw(h, src->L);
  return CRBM_OK;
}

// ---- double-buffered sweeps over host input ---------------------
// Slab i is staged, encoded and processed on stream (i & 1) with buffer set (i & 1);
// its outputs are collected only after slab i+1 has been enqueued, so the host->device
// copy of one slab overlaps the kernels of the other.  Resident sources use set 0 only.
struct SweepSet {
  hipStream_t st;
  DevBuf<float>* stage;
  DevBuf<uint32_t>* letters;
  DevBuf<float>* oa;
  DevBuf<float>* ob;
};
SweepSet sweep_set(crbm_handle* h, int i) {
  if (i & 1) return SweepSet{h->stream2, &h->stage2, &h->letters2, &h->out_a2, &h->out_b2};
  return SweepSet{h->stream, &h->stage, &h->letters, &h->out_a, &h->out_b};
}
// rows per slab: host input uses small slabs so that there is something to overlap
int sweep_slab(const RowSource& src, size_t out_bytes_per_row) {
  const size_t per_row = src.in_bytes_per_row() + out_bytes_per_row;
  int slab = slab_rows(src.n, per_row);
  if (!src.resident && !getenv("CRBM_SLAB_BYTES")) slab = std::min<size_t>(slab, std::max<size_t>(1, (32u << 20) / std::max<size_t>(per_row, 1)));
  return slab;
}
// enqueue staging + encoding of rows [start, start+cnt) on the set's stream (no flag check: sweep_finish)
int sweep_rows(crbm_handle* h, const RowSource& src, int start, int cnt, const SweepSet& set, const uint32_t** out) {
  const int L = src.L, LW = lw(h, L);
  if (src.resident) {
    *out = src.resident + (size_t)start * LW;
    return CRBM_OK;
  }
  HIPCHK(set.letters->ensure((size_t)cnt * LW));
  *out = set.letters->p;
  const int grid = grid_for((long)cnt * LW, 256, h->num_cu * 8);
  if (src.onehot) {
    const size_t count = (size_t)cnt * h->A * L;
    HIPCHK(set.stage->ensure(count));
    HIPCHK(hipMemcpyAsync(set.stage->p, src.onehot + (size_t)start * h->A * L, count * sizeof(float), hipMemcpyHostToDevice, set.st));
    EncodeArgs a;
    a.v = set.stage->p; a.letters = set.letters->p; a.flags = h->d_flags;
    a.n = cnt; a.L = L; a.LW = LW; a.A = h->A;
    if (h->A == 4) hipLaunchKernelGGL(encode_onehot_kernel, dim3(grid), dim3(256), 0, set.st, a);
    else hipLaunchKernelGGL(encode_onehot_any_kernel, dim3(grid), dim3(256), 0, set.st, a);
  } else {
    const size_t bytes = (size_t)cnt * L;
    HIPCHK(set.stage->ensure((bytes + 3) / 4));
    HIPCHK(hipMemcpyAsync(set.stage->p, src.codes + (size_t)start * L, bytes, hipMemcpyHostToDevice, set.st));
    EncodeCodesArgs a;
    a.codes = reinterpret_cast<const unsigned char*>(set.stage->p);
    a.letters = set.letters->p; a.flags = h->d_flags;
    a.n = cnt; a.L = L; a.LW = LW; a.A = h->A;
    if (h->A == 4) hipLaunchKernelGGL(encode_codes_kernel, dim3(grid), dim3(256), 0, set.st, a);
    else hipLaunchKernelGGL(encode_codes_any_kernel, dim3(grid), dim3(256), 0, set.st, a);
  }
  HIPCHK(hipGetLastError());
  return CRBM_OK;
}
// tables built and visible to both streams before a sweep starts
int sweep_begin(crbm_handle* h) {
  int rc = ensure_tables(h);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}
// both streams idle, then the one-hot / code validity flags of every slab at once
int sweep_finish(crbm_handle* h, const RowSource& src) {
  HIPCHK(hipStreamSynchronize(h->stream2));
  HIPCHK(hipStreamSynchronize(h->stream));
  return src.resident ? CRBM_OK : check_flags(h);
}

int hit_probs_any(crbm_handle* h, const RowSource& src, float* out) {
  int rc = check_data_shape(h, src.n, src.L);
  if (rc) return rc;
  const int L = src.L, Lh = L - h->M + 1;
  const size_t per_seq_out = (size_t)h->K * Lh;
  const int slab = slab_rows(src.n, src.in_bytes_per_row() + per_seq_out * sizeof(float));
  HIPCHK(h->out_b.ensure((size_t)slab * per_seq_out));
  for (int start = 0; start < src.n; start += slab) {
    const int cnt = std::min(slab, src.n - start);
    const uint32_t* rows = nullptr;
    rc = source_rows(h, src, start, cnt, &rows);
    if (rc) return rc;
    // convRBM.py:507-514: doublestranded -> sigma(x); single-stranded -> sigma(x + x')
    rc = launch_hgv(h, rows, cnt, L, h->ds ? 0 : 2, nullptr, h->out_b.p, nullptr, nullptr, KIND_API_H, 0, 0);
    if (rc) return rc;
    rc = copy_out(h, out + (size_t)start * per_seq_out, h->out_b.p, (size_t)cnt * per_seq_out);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return CRBM_OK;
}

// free energies of n packed rows into the set's outputs (per sequence / per motif), no copy
int launch_free_energy(crbm_handle* h, const uint32_t* rows, int n, int L, const SweepSet& set) {
  int rc = ensure_tables(h);   // no-op inside a sweep (sweep_begin has done it)
  if (rc) return rc;
  HIPCHK(set.oa->ensure((size_t)n));
  HIPCHK(set.ob->ensure((size_t)n * h->K));
  if (h->big) return big_launch_eval(h, rows, n, L, 0, set.oa->p, set.ob->p, nullptr, nullptr, nullptr, set.st);
  FeArgs a;
  a.tables = h->d_tables;
  a.letters = rows;
  a.n = n; a.L = L; a.Lh = L - h->M + 1; a.LW = lw(h, L);
  a.fe = set.oa->p; a.fem = set.ob->p;
  const unsigned gx = (unsigned)std::max(1, std::min((n + 3) / 4, h->num_cu * 8));
  HIPCHK(jit_launch(h->jk.free_energy, a, gx, 1, 256, (unsigned)tab_bytes(h), set.st));
  return CRBM_OK;
}

int free_energy_any(crbm_handle* h, const RowSource& src, float* fe, float* fem) {
  int rc = check_data_shape(h, src.n, src.L);
  if (rc) return rc;
  rc = sweep_begin(h);
  if (rc) return rc;
  const int slab = sweep_slab(src, ((size_t)h->K + 1) * sizeof(float));
  int prev_start = -1, prev_cnt = 0;
  SweepSet prev = sweep_set(h, 0);
  auto collect = [&]() -> int {     // outputs of the previous slab -> host
    if (prev_start < 0) return CRBM_OK;
    if (fe) HIPCHK(hipMemcpyAsync(fe + prev_start, prev.oa->p, (size_t)prev_cnt * sizeof(float), hipMemcpyDeviceToHost, prev.st));
    if (fem) HIPCHK(hipMemcpyAsync(fem + (size_t)prev_start * h->K, prev.ob->p, (size_t)prev_cnt * h->K * sizeof(float), hipMemcpyDeviceToHost, prev.st));
    HIPCHK(hipStreamSynchronize(prev.st));
    return CRBM_OK;
  };
  for (int start = 0, i = 0; start < src.n; start += slab, ++i) {
    const int cnt = std::min(slab, src.n - start);
    const SweepSet set = sweep_set(h, src.resident ? 0 : i);
    const uint32_t* rows = nullptr;
    rc = sweep_rows(h, src, start, cnt, set, &rows);
    if (rc) return rc;
    rc = launch_free_energy(h, rows, cnt, src.L, set);
    if (rc) return rc;
    rc = collect();
    if (rc) return rc;
    prev = set; prev_start = start; prev_cnt = cnt;
  }
  rc = collect();
  if (rc) return rc;
  return sweep_finish(h, src);
}

int eval_data_any(crbm_handle* h, const RowSource& src, float* mfe, float* nmh) {
  int rc = check_data_shape(h, src.n, src.L);
  if (rc) return rc;
  const int n = src.n, L = src.L;
  const int slab = slab_rows(n, src.in_bytes_per_row() + ((size_t)h->K + 1) * sizeof(float));
  std::vector<float> fe((size_t)slab);
  double tot = 0.0;
  HIPCHK(hipMemsetAsync(h->d_ones, 0, sizeof(unsigned long long), h->stream));
  for (int start = 0; start < n; start += slab) {
    const int cnt = std::min(slab, n - start);
    const uint32_t* rows = nullptr;
    rc = source_rows(h, src, start, cnt, &rows);
    if (rc) return rc;
    rc = launch_free_energy(h, rows, cnt, L, sweep_set(h, 0));
    if (rc) return rc;
    rc = copy_out(h, fe.data(), h->out_a.p, (size_t)cnt);
    if (rc) return rc;
    // mean of a fresh forward-strand sample (convRBM.py:469-472); rows keep their index within the call
    rc = launch_hgv(h, rows, cnt, L, 0, nullptr, nullptr, nullptr, h->d_ones, KIND_EVAL_H, h->eval_step, (uint32_t)start);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < cnt; ++i) tot += fe[i];
  }
  *mfe = (float)(tot / n);                       // convRBM.py:636-638
  h->eval_step += 1;
  unsigned long long ones = 0;
  HIPCHK(hipMemcpyAsync(&ones, h->d_ones, sizeof(ones), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *nmh = (float)((double)ones / ((double)n * h->K * (L - h->M + 1)));
  return CRBM_OK;
}

// max / mean over positions per (sequence, motif) and mean over sequences per (motif, position)
int hit_summary_any(crbm_handle* h, const RowSource& src, float* hmax, float* hmean, float* posmean) {
  int rc = check_data_shape(h, src.n, src.L);
  if (rc) return rc;
  const int n = src.n, L = src.L, Lh = L - h->M + 1, K = h->K;
  const int tabs = h->big ? 0 : tab_bytes(h);
  // a block covers one chunk of 64*HIT_NI positions; its (PC,K) sums share the LDS with the tables
  const int PC = 64 * h->ms.HIT_NI;
  const int nchunks = h->big ? 1 : (Lh + PC - 1) / PC;           // (the big path's kernel takes a whole sequence per block)
  const unsigned lds = (unsigned)(tabs + (size_t)PC * K * 8);    // tables + the block's fixed-point position sums
  ARGCHK(h->big || lds <= 160u * 1024u, "model too large for the hit-summary kernel");
  // sums over sequences per (motif, position): 64-bit fixed point (HIT_FX units), so that the order in which blocks,
  // slabs and the two streams add does not show in the result
  unsigned long long* pos_fx = nullptr;
  if (posmean) {
    HIPCHK(h->out_c.ensure((size_t)2 * K * Lh + 2));
    pos_fx = reinterpret_cast<unsigned long long*>(h->out_c.p);
    HIPCHK(hipMemsetAsync(pos_fx, 0, (size_t)K * Lh * 8, h->stream));
  }
  rc = sweep_begin(h);          // also orders the memset before both streams' kernels
  if (rc) return rc;
  const int slab = sweep_slab(src, (size_t)2 * K * sizeof(float));
  int prev_start = -1, prev_cnt = 0;
  SweepSet prev = sweep_set(h, 0);
  // layout of a set's `ob` when the positions come in several chunks: [cnt*K fixed-point sums][cnt*K means]
  auto mean_of = [&](const SweepSet& set, int cnt) { return nchunks > 1 ? set.ob->p + (size_t)2 * cnt * K : set.ob->p; };
  auto collect = [&]() -> int {
    if (prev_start < 0) return CRBM_OK;
    if (hmax) HIPCHK(hipMemcpyAsync(hmax + (size_t)prev_start * K, prev.oa->p, (size_t)prev_cnt * K * sizeof(float), hipMemcpyDeviceToHost, prev.st));
    if (hmean) HIPCHK(hipMemcpyAsync(hmean + (size_t)prev_start * K, mean_of(prev, prev_cnt), (size_t)prev_cnt * K * sizeof(float), hipMemcpyDeviceToHost, prev.st));
    HIPCHK(hipStreamSynchronize(prev.st));
    return CRBM_OK;
  };
  for (int start = 0, i = 0; start < n; start += slab, ++i) {
    const int cnt = std::min(slab, n - start);
    const SweepSet set = sweep_set(h, src.resident ? 0 : i);
    const uint32_t* rows = nullptr;
    rc = sweep_rows(h, src, start, cnt, set, &rows);
    if (rc) return rc;
    HIPCHK(set.oa->ensure((size_t)cnt * K));
    HIPCHK(set.ob->ensure((size_t)cnt * K * (nchunks > 1 ? 3 : 1)));
    if (nchunks > 1) {
      HIPCHK(hipMemsetAsync(set.oa->p, 0, (size_t)cnt * K * 4, set.st));
      HIPCHK(hipMemsetAsync(set.ob->p, 0, (size_t)cnt * K * 8, set.st));
    }
    HitArgs a;
    a.tables = h->d_tables; a.letters = rows;
    a.n = cnt; a.L = L; a.Lh = Lh; a.LW = lw(h, L);
    a.hmax = hmax ? set.oa->p : nullptr;
    a.hsum = (hmean && nchunks == 1) ? set.ob->p : nullptr;
    a.hsum_fx = (hmean && nchunks > 1) ? reinterpret_cast<unsigned long long*>(set.ob->p) : nullptr;
    a.inv_Lh = 1.0f / (float)Lh;
    a.pos_fx = pos_fx;                          // both streams add (integers: any order)
    const unsigned gx = (unsigned)std::max(1, std::min((cnt + 3) / 4, std::max(1, h->num_cu * 8 / nchunks)));
    if (h->big) {
      rc = big_launch_eval(h, rows, cnt, L, 1, nullptr, nullptr, a.hmax, a.hsum, pos_fx, set.st);
      if (rc) return rc;
    } else
    HIPCHK(jit_launch(h->jk.hit_summary, a, gx, (unsigned)nchunks, 256, lds, set.st));
    if (a.hsum_fx) {
      hipLaunchKernelGGL(hit_finalize_kernel, dim3(grid_for((long)cnt * K, 256, h->num_cu * 4)), dim3(256), 0, set.st, a.hsum_fx,
                         mean_of(set, cnt), (size_t)cnt * K, a.inv_Lh / HIT_FX);
      HIPCHK(hipGetLastError());
    }
    rc = collect();
    if (rc) return rc;
    prev = set; prev_start = start; prev_cnt = cnt;
  }
  rc = collect();
  if (rc) return rc;
  rc = sweep_finish(h, src);
  if (rc) return rc;
  if (posmean) {
    std::vector<unsigned long long> fx((size_t)K * Lh);
    HIPCHK(hipMemcpyAsync(fx.data(), pos_fx, fx.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const double inv = 1.0 / ((double)n * (double)HIT_FX);
    for (size_t i = 0; i < fx.size(); ++i) posmean[i] = (float)((double)fx[i] * inv);
  }
  return CRBM_OK;
}

static RowSource host_onehot_(const float* v, int n, int L, int A) { RowSource s; s.onehot = v; s.n = n; s.L = L; s.A = A; return s; }
static RowSource host_codes_(const uint8_t* c, int n, int L, int A) { RowSource s; s.codes = c; s.n = n; s.L = L; s.A = A; return s; }
#define host_onehot(v, n, L) host_onehot_(v, n, L, h->A)
#define host_codes(c, n, L) host_codes_(c, n, L, h->A)

}  // namespace

int crbm_hit_probs(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out) {
  ENTER();
  ARGCHK(v && out, "null argument");
  return hit_probs_any(h, host_onehot(v, n, L), out);
}

int crbm_hit_probs_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* out) {
  ENTER();
  ARGCHK(codes && out, "null argument");
  return hit_probs_any(h, host_codes(codes, n, L), out);
}

int crbm_hit_probs_resident(crbm_handle* h, int32_t start, int32_t end, float* out) {
  ENTER();
  ARGCHK(out, "null argument");
  RowSource src;
  int rc = resident_source(h, start, end, &src);
  if (rc) return rc;
  return hit_probs_any(h, src, out);
}

int crbm_free_energy(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out) {
  ENTER();
  ARGCHK(v && out, "null argument");
  return free_energy_any(h, host_onehot(v, n, L), out, nullptr);
}

int crbm_free_energy_per_motif(crbm_handle* h, const float* v, int32_t n, int32_t L, float* out) {
  ENTER();
  ARGCHK(v && out, "null argument");
  return free_energy_any(h, host_onehot(v, n, L), nullptr, out);
}

int crbm_free_energy_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* fe, float* fe_per_motif) {
  ENTER();
  ARGCHK(codes && (fe || fe_per_motif), "null argument");
  return free_energy_any(h, host_codes(codes, n, L), fe, fe_per_motif);
}

int crbm_free_energy_resident(crbm_handle* h, int32_t start, int32_t end, float* fe, float* fe_per_motif) {
  ENTER();
  ARGCHK(fe || fe_per_motif, "null argument");
  RowSource src;
  int rc = resident_source(h, start, end, &src);
  if (rc) return rc;
  return free_energy_any(h, src, fe, fe_per_motif);
}

int crbm_eval_data(crbm_handle* h, const float* v, int32_t n, int32_t L, float* mfe, float* nmh) {
  ENTER();
  ARGCHK(v && mfe && nmh, "null argument");
  return eval_data_any(h, host_onehot(v, n, L), mfe, nmh);
}

// The per-epoch evaluation of fit() (convRBM.py:616-625) in one call: the selected resident set is walked in mini-batches
// of `batchsize` rows exactly as a loop over crbm_eval_data_resident would -- batch b draws its hidden sample with sampler
// step eval_step + b and rows indexed from 0 -- but every launch is enqueued before the one read-back.  Returns the MEAN
// OVER BATCHES of the batches' mean free energy and of their mean sampled hidden activity (what the reference's status
// line prints: a short last batch weighs as much as a full one), each batch value rounded to float as the per-batch
// entry point returns it.
int crbm_eval_epoch_resident(crbm_handle* h, int32_t batchsize, double* mean_fe, double* mean_nmh) {
  ENTER();
  ARGCHK(mean_fe && mean_nmh, "null argument");
  ARGCHK(batchsize >= 1, "batchsize must be positive");
  const int slot = h->slot;
  ARGCHK(h->dataset_n[slot] > 0, "no resident data set (call crbm_dataset_upload)");
  const int n = h->dataset_n[slot], L = h->dataset_L[slot], LW = lw(h, L), Lh = L - h->M + 1;
  int rc = check_data_shape(h, n, L);
  if (rc) return rc;
  const int nb = (n + batchsize - 1) / batchsize;
  HIPCHK(h->eval_ones.ensure((size_t)nb));
  HIPCHK(hipMemsetAsync(h->eval_ones.p, 0, (size_t)nb * sizeof(unsigned long long), h->stream));
  const uint32_t* rows = h->dataset[slot].p;
  rc = launch_free_energy(h, rows, n, L, sweep_set(h, 0));        // a row's free energy does not depend on its batch
  if (rc) return rc;
  for (int b = 0; b < nb; ++b) {
    const int start = b * batchsize, cnt = std::min(batchsize, n - start);
    rc = launch_hgv(h, rows + (size_t)start * LW, cnt, L, 0, nullptr, nullptr, nullptr, h->eval_ones.p + b, KIND_EVAL_H,
                    h->eval_step + (uint32_t)b, 0u);
    if (rc) return rc;
  }
  h->eval_step += (uint32_t)nb;
  std::vector<float> fe((size_t)n);
  std::vector<unsigned long long> ones((size_t)nb);
  HIPCHK(hipMemcpyAsync(fe.data(), h->out_a.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(ones.data(), h->eval_ones.p, (size_t)nb * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double sfe = 0.0, snmh = 0.0;
  for (int b = 0; b < nb; ++b) {
    const int start = b * batchsize, cnt = std::min(batchsize, n - start);
    double tot = 0.0;
    for (int i = 0; i < cnt; ++i) tot += fe[(size_t)start + i];
    sfe += (double)(float)(tot / cnt);                                              // convRBM.py:636-638 per batch
    snmh += (double)(float)((double)ones[(size_t)b] / ((double)cnt * h->K * Lh));     // :469-472
  }
  *mean_fe = sfe / nb;
  *mean_nmh = snmh / nb;
  return CRBM_OK;
}

int crbm_eval_data_resident(crbm_handle* h, int32_t start, int32_t end, float* mfe, float* nmh) {
  ENTER();
  ARGCHK(mfe && nmh, "null argument");
  RowSource src;
  int rc = resident_source(h, start, end, &src);
  if (rc) return rc;
  return eval_data_any(h, src, mfe, nmh);
}

int crbm_hit_summary(crbm_handle* h, const float* v, int32_t n, int32_t L, float* hit_max, float* hit_mean,
                     float* position_mean) {
  ENTER();
  ARGCHK(v && (hit_max || hit_mean || position_mean), "null argument");
  return hit_summary_any(h, host_onehot(v, n, L), hit_max, hit_mean, position_mean);
}

int crbm_hit_summary_codes(crbm_handle* h, const uint8_t* codes, int32_t n, int32_t L, float* hit_max, float* hit_mean,
                           float* position_mean) {
  ENTER();
  ARGCHK(codes && (hit_max || hit_mean || position_mean), "null argument");
  return hit_summary_any(h, host_codes(codes, n, L), hit_max, hit_mean, position_mean);
}

int crbm_hit_summary_resident(crbm_handle* h, int32_t start, int32_t end, float* hit_max, float* hit_mean,
                              float* position_mean) {
  ENTER();
  ARGCHK(hit_max || hit_mean || position_mean, "null argument");
  RowSource src;
  int rc = resident_source(h, start, end, &src);
  if (rc) return rc;
  return hit_summary_any(h, src, hit_max, hit_mean, position_mean);
}

int crbm_eval_params(crbm_handle* h, float* twn, float* ic, float* medic) {
  ENTER();
  ARGCHK(twn && ic && medic, "null argument");
  const int K = h->K, M = h->M;
  std::vector<float> W((size_t)h->KAM);
  HIPCHK(hipMemcpyAsync(W.data(), h->dW, W.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  // convRBM.py:475-485 on A*K*M numbers: host arithmetic in double
  const int A = h->A;
  double sq = 0.0, entsum = 0.0, medsum = 0.0;
  std::vector<double> ent((size_t)M), w((size_t)A);
  for (int k = 0; k < K; ++k) {
    for (int j = 0; j < M; ++j) {
      double mx = -1e300, z = 0.0, e = 0.0;
      for (int a = 0; a < A; ++a) {
        w[a] = W[(size_t)(k * A + a) * M + j];
        sq += w[a] * w[a];
        mx = std::max(mx, w[a]);
      }
      for (int a = 0; a < A; ++a) z += std::exp(w[a] - mx);
      for (int a = 0; a < A; ++a) {
        const double p = std::exp(w[a] - mx) / z;
        if (p > 0.0) e -= p * std::log2(p);
      }
      ent[j] = e;
      entsum += e;
    }
    std::sort(ent.begin(), ent.end());
    medsum += ent[(size_t)(M / 2)];
  }
  *twn = (float)std::sqrt(sq / (double)h->KAM);
  const double full = std::log2((double)A);     // log2 of the alphabet size (:481-484); 2 bits for DNA
  *ic = (float)(full - entsum / ((double)K * M));
  *medic = (float)(full - medsum / (double)K);
  return CRBM_OK;
}

// ---- data parallel -------------------------------------------------------------
int crbm_comm_unique_id(uint8_t id[CRBM_UNIQUE_ID_BYTES]) {
  if (!id) return CRBM_ERR_INVALID;
  if (!load_rccl()) { g_create_error = g_rccl.error; return CRBM_ERR_RCCL; }
  ncclUniqueId uid;
  static_assert(sizeof(uid) == CRBM_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return CRBM_ERR_RCCL; }
  memcpy(id, &uid, sizeof(uid));
  return CRBM_OK;
}

int crbm_comm_init(crbm_handle* h, const uint8_t id[CRBM_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank) {
  ENTER();
  ARGCHK(id && nranks >= 1 && rank >= 0 && rank < nranks, "bad argument");
  if (!load_rccl()) return fail(h, CRBM_ERR_RCCL, g_rccl.error);
  if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclResult_t r = g_rccl.CommInitRank(&h->comm, nranks, uid, rank);
  if (r != ncclSuccess) {
    h->comm = nullptr;
    return fail(h, CRBM_ERR_RCCL, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  h->nranks = nranks; h->rank = rank;
  return CRBM_OK;
}

int crbm_comm_destroy(crbm_handle* h) {
  ENTER();
  if (h->comm) {
    HIPCHK(hipStreamSynchronize(h->stream));
    g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
  }
  h->nranks = 1; h->rank = 0;
  return CRBM_OK;
}

// Replicas must start identical: only raw statistic sums are all-reduced, so a
// rank that begins from different parameters would apply the common update to a
// different model for ever.  ncclBroadcast of W, b, c and the three velocities
// from `root` (no-op without a communicator).
int crbm_comm_broadcast_state(crbm_handle* h, int32_t root) {
  ENTER();
  if (!h->comm) return CRBM_OK;
  ARGCHK(root >= 0 && root < h->nranks, "root out of range");
  struct { float* p; size_t n; } bufs[] = {{h->dW, (size_t)h->KAM}, {h->db, (size_t)h->K}, {h->dc, (size_t)h->A},
                                           {h->dvW, (size_t)h->KAM}, {h->dvb, (size_t)h->K}, {h->dvc, (size_t)h->A}};
  for (auto& b : bufs) {
    ncclResult_t r = g_rccl.Broadcast(b.p, b.p, b.n, ncclFloat, root, h->comm, h->stream);
    if (r != ncclSuccess) return fail(h, CRBM_ERR_RCCL, std::string("ncclBroadcast: ") + g_rccl.GetErrorString(r));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->tables_dirty = true;
  h->params_version += 1;
  return CRBM_OK;
}

// ---- all-reduce through mapped buffers (no collective launch) -------------------------------------
int crbm_ipc_export(crbm_handle* h, uint8_t handle[CRBM_IPC_HANDLE_BYTES]) {
  ENTER();
  ARGCHK(handle, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) == CRBM_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  int rc = ipc_allocate(h);
  if (rc) return rc;
  // flags and status start from zero with every attachment (the step count does too): cleared here, before the
  // handle leaves -- no peer pushes into this buffer before it has seen the handles of all ranks
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemset(h->ipc_buf, 0, ipc_bytes(h)));
  hipIpcMemHandle_t mh;
  HIPCHK(hipIpcGetMemHandle(&mh, h->ipc_buf));
  memcpy(handle, &mh, sizeof(mh));
  return CRBM_OK;
}

int crbm_ipc_attach(crbm_handle* h, const uint8_t* handles, int32_t nranks, int32_t rank) {
  ENTER();
  ARGCHK(handles && nranks >= 1 && nranks <= IPC_MAX_RANKS && rank >= 0 && rank < nranks, "bad argument (at most 8 ranks: one node)");
  ARGCHK(!h->comm, "the handle already has an RCCL communicator");
  ARGCHK(!h->big, "the mapped-buffer all-reduce serves models of the LDS-resident kernels; this one (generic kernels) takes RCCL");
  int rc = ipc_allocate(h);
  if (rc) return rc;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { h->ipc_peer[r] = h->ipc_buf; continue; }
    hipIpcMemHandle_t mh;
    memcpy(&mh, handles + (size_t)r * CRBM_IPC_HANDLE_BYTES, sizeof(mh));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(h, CRBM_ERR_HIP, std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
    h->ipc_peer[r] = p;
  }
  h->nranks = nranks; h->rank = rank;
  h->ipc_on = true;
  h->ipc_step = 0;
  // the bound of an update launch's wait for its peers, in ticks of the GPU's constant-rate clock (s_memrealtime;
  // its rate in kHz = ticks per millisecond, 100 MHz on every part so far)
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) {
    (void)hipGetLastError();
    khz = 100000;
  }
  h->ipc_timeout_ticks = (unsigned long long)std::max(1, env_int("CRBM_IPC_TIMEOUT_MS", 30000)) * (unsigned long long)khz;
  return CRBM_OK;
}

int crbm_ipc_detach(crbm_handle* h) {
  ENTER();
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int r = 0; r < IPC_MAX_RANKS; ++r) {
    if (h->ipc_peer[r] && h->ipc_peer[r] != h->ipc_buf) (void)hipIpcCloseMemHandle(h->ipc_peer[r]);
    h->ipc_peer[r] = nullptr;
  }
  h->ipc_on = false;
  h->nranks = 1; h->rank = 0;
  return CRBM_OK;
}

int crbm_ipc_status(crbm_handle* h, int32_t* timed_out) {
  ENTER();
  ARGCHK(timed_out, "null argument");
  *timed_out = 0;
  if (!h->ipc_buf) return CRBM_OK;
  uint32_t st = 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(&st, ipc_status_of(h->ipc_buf, h), sizeof(st), hipMemcpyDeviceToHost));
  *timed_out = (int32_t)st;
  return CRBM_OK;
}

int crbm_sums_count(const crbm_handle* h) { return h ? h->sl.count : 0; }

int crbm_train_local(crbm_handle* h, const float* D, int32_t n, int32_t L, float* sums_out) {
  ENTER();
  ARGCHK(sums_out && (D || n == 0), "null argument");
  int rc = check_data_shape(h, std::max(n, 1), L);
  if (rc) return rc;
  ARGCHK(n >= 0, "n must be non-negative");
  if (n > 0) {   // n == 0: a rank that owns no row of a short last mini-batch contributes zeros
    HIPCHK(h->letters.ensure((size_t)n * lw(h, L)));
    rc = encode_host(h, D, n, L, h->letters.p);
    if (rc) return rc;
  }
  rc = train_local_dev(h, h->letters.p, n, L);
  if (rc) return rc;
  rc = copy_out(h, sums_out, h->d_sums, (size_t)h->sl.count);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_train_apply(crbm_handle* h, const float* sums_in, int32_t L_data) {
  ENTER();
  ARGCHK(sums_in && L_data >= h->M, "bad argument");
  HIPCHK(hipMemcpyAsync(h->d_sums, sums_in, (size_t)h->sl.count * 4, hipMemcpyHostToDevice, h->stream));
  int rc = launch_update(h, L_data);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return CRBM_OK;
}

int crbm_time_allreduce(crbm_handle* h, int32_t launches, float* total_ms) {
  ENTER();
  ARGCHK(launches >= 1 && total_ms, "bad argument");
  *total_ms = 0.f;
  if (!h->comm) return CRBM_OK;
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < launches; ++i) {
    ncclResult_t r = g_rccl.AllReduce(h->d_sums, h->d_sums, (size_t)h->sl.count, ncclFloat, ncclSum, h->comm, h->stream);
    if (r != ncclSuccess) return fail(h, CRBM_ERR_RCCL, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(spin_until(h->ev1));
  HIPCHK(hipEventElapsedTime(total_ms, h->ev0, h->ev1));
  return CRBM_OK;
}

int crbm_get_launch_info(const crbm_handle* h, crbm_launch_info* out) {
  if (!h || !out) return CRBM_ERR_INVALID;
  if (h->big) {          // generic kernels: no specialised geometry to report
    *out = crbm_launch_info();
    out->gibbs_sparse = 1; out->chain_parts = 1; out->gibbs_block = 256;
    out->activity_ppm = h->activity < 0.0 ? -1 : (int32_t)(h->activity * 1e6 + 0.5);
    return CRBM_OK;
  }
  // the data-half statistics kernel at the chains' shape (stats_mfma_body)
  const int tabs = h->ms.TAB * 4;
  const StatsMfmaLayout st = stats_mfma_layout(h->ms, 1, h->Lf, 0, tabs);
  const int lds = std::max(st.region_floats * 4 + tabs, st.combine_bytes);
  out->nq = h->ms.NQ; out->group = h->GS;   // of the plain chain launch this structure describes
  // the geometry of a plain chain launch (crbm_gibbs_steps*): the solo one where the model has it
  const bool parts = h->variant == 1 && h->chain_parts > 1;
  const bool solo = h->variant == 1 && h->solo_threads > 0;
  out->chain_parts = parts ? h->chain_parts : 1;
  out->gibbs_grid = parts ? h->part_grid : solo ? h->solo_grid : h->gibbs_grid;
  out->gibbs_block = parts ? h->part_threads : solo ? h->solo_threads : h->gibbs_threads;
  out->gibbs_seqs_per_tile = parts ? h->gl_part.S : solo ? h->gl_solo.S : h->gl.S;
  out->gibbs_lds_bytes = parts ? h->gl_part.lds_bytes : solo ? h->gl_solo.lds_bytes : h->gl.lds_bytes;
  out->stats_grid_x = h->stats_rows > 0 ? h->stats_rows
                                        : h->num_cu * std::max(1, std::min(2048 / st.threads, (160 * 1024) / std::max(1, lds)));
  out->stats_grid_y = 1;
  out->stats_block = st.threads; out->stats_lds_bytes = lds;
  out->stats_fused = h->fuse_stats ? 1 : 0;
  out->gibbs_sparse = h->variant;
  out->activity_ppm = h->activity < 0.0 ? -1 : (int32_t)(h->activity * 1e6 + 0.5);
  return CRBM_OK;
}

// Device-copy bandwidth of this GPU, measured with a float4 copy kernel on the
// library's stream (HIP events): the practical ceiling next to the 8 TB/s spec
// that bench.py quotes its roofline fraction against as well.
int crbm_copy_bandwidth(crbm_handle* h, int64_t bytes, int32_t reps, float* gb_per_s) {
  ENTER();
  ARGCHK(bytes >= (1 << 20) && reps >= 1 && gb_per_s, "bad argument");
  const size_t n4 = (size_t)bytes / 16;
  float4 *src = nullptr, *dst = nullptr;
  HIPCHK(hipMalloc((void**)&src, n4 * 16));
  hipError_t e = hipMalloc((void**)&dst, n4 * 16);
  if (e != hipSuccess) { (void)hipFree(src); return fail(h, CRBM_ERR_HIP, "hipMalloc (copy buffer)"); }
  (void)hipMemsetAsync(src, 1, n4 * 16, h->stream);
  const unsigned grid = (unsigned)((n4 + 1023) / 1024);
  for (int i = 0; i < 1 + reps; ++i) {
    if (i == 1) (void)hipEventRecord(h->ev0, h->stream);
    hipLaunchKernelGGL(copy_float4_kernel, dim3(grid), dim3(256), 0, h->stream, src, dst, n4);
  }
  (void)hipEventRecord(h->ev1, h->stream);
  e = hipEventSynchronize(h->ev1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev0, h->ev1);
  (void)hipFree(src); (void)hipFree(dst);
  if (e != hipSuccess) return fail(h, CRBM_ERR_HIP, std::string("copy bandwidth: ") + hipGetErrorString(e));
  *gb_per_s = (float)(2.0 * (double)n4 * 16.0 * reps / (ms * 1e-3) / 1e9);   // bytes read + written
  return CRBM_OK;
}

// Shader clock during the launches of the last crbm_time_gibbs call: block 0 of every launch (of partition 0) adds its
// duration in wall-clock ticks and in shader cycles (GibbsArgs::clock); MHz = cycles / ticks x tick rate.  0 if unknown.
int crbm_last_shader_clock(crbm_handle* h, float* mhz) {
  ENTER();
  ARGCHK(mhz, "null argument");
  *mhz = 0.f;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->probe_last, h->d_probe, sizeof(h->probe_last), hipMemcpyDeviceToHost));
  const unsigned long long ticks = h->probe_last[0] - h->probe_base[0], cycles = h->probe_last[1] - h->probe_base[1];
  if (ticks) *mhz = (float)((double)cycles / ((double)ticks / (h->wall_khz / 1000.0)));
  return CRBM_OK;
}

int64_t crbm_gibbs_state_bytes(const crbm_handle* h) {
  if (!h) return 0;
  const int64_t masks = (int64_t)h->B * h->Lf * h->NW * 4 * (1 + h->ds);
  const int64_t vout = (int64_t)h->B * h->gl.LWs * 4;
  return 2 * masks + vout;   // masks read + written, last visible sample written
}

}  // extern "C"
