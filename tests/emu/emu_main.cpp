// TEST INFRASTRUCTURE: runs the HIP kernels of crbm_amd/csrc/crbm_kernels.h on
// CPU threads (see shim/hip/hip_runtime.h) so that indexing, LDS layouts and
// barrier placement are checked under ASan/UBSan before a kernel ever reaches
// a GPU.  Exposes plain C entry points for tests/emu/run_emu.py (ctypes).
#define CRBM_DEFINE_MISC_KERNELS
#include "crbm_kernels.h"

#include <thread>
#include <vector>

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx;

template <typename F>
void launch(F kernel, dim3 grid, dim3 block, size_t lds) {
  const unsigned nthr = block.x, nwaves = (block.x + 63) / 64;
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      BlockCtx ctx;
      pthread_barrier_init(&ctx.bar, nullptr, nthr);
      std::vector<pthread_barrier_t> wb(nwaves);
      for (unsigned w = 0; w < nwaves; ++w) pthread_barrier_init(&wb[w], nullptr, std::min(64u, nthr - w * 64));
      std::vector<float> scratch(nwaves * 64);
      std::vector<uint32_t> frag((size_t)nwaves * 64 * 8);
      // exact size (16-byte aligned base): out-of-bounds LDS accesses trip AddressSanitizer
      std::vector<float4> smem((lds + 15) / 16 + 1);
      memset(smem.data(), 0xAB, smem.size() * 16);
      ctx.wave_bar = wb.data();
      ctx.wave_scratch = scratch.data();
      ctx.wave_frag = frag.data();
      ctx.smem = reinterpret_cast<unsigned char*>(smem.data());
      std::vector<std::thread> threads;
      threads.reserve(nthr);
      for (unsigned t = 0; t < nthr; ++t)
        threads.emplace_back([&, t]() {
          t_threadIdx = dim3(t, 0, 0);
          t_blockIdx = dim3(bx, by, 0);
          t_blockDim = block;
          t_gridDim = grid;
          t_ctx = &ctx;
          kernel();
        });
      for (auto& th : threads) th.join();
      pthread_barrier_destroy(&ctx.bar);
      for (auto& b : wb) pthread_barrier_destroy(&b);
    }
}
}  // namespace emu

using namespace crbm;

namespace {
RngView make_rng(uint64_t seed, uint32_t step, uint32_t off) {
  RngView r;
  r.seed_lo = (uint32_t)seed; r.seed_hi = (uint32_t)(seed >> 32); r.step = step; r.seq_offset = off;
  return r;
}

// the model configurations the emulation is instantiated for (K, M, DS, G)
#define CFG_DISPATCH(id, ...)                                    \
  switch (id) {                                                  \
    case 0: { using C = Cfg<10, 5, 0, 2>; __VA_ARGS__; break; }         \
    case 1: { using C = Cfg<10, 15, 1, 3>; __VA_ARGS__; break; }        \
    case 2: { using C = Cfg<2, 5, 1, 4>; __VA_ARGS__; break; }          \
    case 3: { using C = Cfg<3, 4, 0, 1>; __VA_ARGS__; break; }          \
    case 4: { using C = Cfg<20, 15, 1, 2>; __VA_ARGS__; break; }        \
    case 5: { using C = Cfg<50, 25, 0, 2>; __VA_ARGS__; break; }        \
    case 6: { using C = Cfg<7, 32, 1, 3>; __VA_ARGS__; break; }         \
    case 7: { using C = Cfg<10, 15, 0, 3>; __VA_ARGS__; break; }        \
    case 8: { using C = Cfg<64, 32, 1, 1>; __VA_ARGS__; break; }        \
    case 9: { using C = Cfg<4, 5, 1, 2, 2>; __VA_ARGS__; break; }       \
    case 10: { using C = Cfg<10, 15, 0, 3, 3>; __VA_ARGS__; break; }    \
    case 11: { using C = Cfg<100, 15, 0, 1>; __VA_ARGS__; break; }      \
    case 12: { using C = Cfg<20, 40, 1, 2>; __VA_ARGS__; break; }       \
    case 13: { using C = Cfg<70, 33, 1, 1>; __VA_ARGS__; break; }       \
    default: return -1;                                          \
  }

template <class C>
ModelShape shape_of() { return model_shape(C::K, C::M, C::DS, C::G, C::POOL); }
}  // namespace

extern "C" {

int emu_case_info(int id, int* out) {   // K, M, DS, G, TABLES, NW, DENSE, POOL
  CFG_DISPATCH(id, (out[0] = C::K, out[1] = C::M, out[2] = C::DS, out[3] = C::G, out[4] = C::TABLES_ALL, out[5] = C::NW,
                    out[6] = C::DENSE ? 1 : 0, out[7] = C::POOL));
  return 0;
}

int emu_encode(const float* v, uint32_t* letters, uint32_t* flags, int n, int L, int grid) {
  EncodeArgs a{v, letters, flags, n, L, letter_words(L), 4};
  emu::launch([&] { encode_onehot_kernel(a); }, dim3(grid), dim3(64), 0);
  return 0;
}

int emu_decode(const uint32_t* letters, float* v, int n, int L, int LW, int grid) {
  DecodeArgs a{letters, v, n, L, LW, 4};
  emu::launch([&] { decode_onehot_kernel(a); }, dim3(grid), dim3(64), 0);
  return 0;
}

// any alphabet (A != 4: rows of bytes); codes != null: letter codes instead of one-hot floats
int emu_encode_any(const float* v, const unsigned char* codes, uint32_t* letters, uint32_t* flags, int n, int L, int A, int grid) {
  if (codes) {
    EncodeCodesArgs a{codes, letters, flags, n, L, letter_words_any(A, L), A};
    if (A == 4) emu::launch([&] { encode_codes_kernel(a); }, dim3(grid), dim3(64), 0);
    else emu::launch([&] { encode_codes_any_kernel(a); }, dim3(grid), dim3(64), 0);
    return 0;
  }
  EncodeArgs a{v, letters, flags, n, L, letter_words_any(A, L), A};
  if (A == 4) emu::launch([&] { encode_onehot_kernel(a); }, dim3(grid), dim3(64), 0);
  else emu::launch([&] { encode_onehot_any_kernel(a); }, dim3(grid), dim3(64), 0);
  return 0;
}

int emu_decode_any(const uint32_t* letters, float* v, int n, int L, int LW, int A, int grid) {
  DecodeArgs a{letters, v, n, L, LW, A};
  if (A == 4) emu::launch([&] { decode_onehot_kernel(a); }, dim3(grid), dim3(64), 0);
  else emu::launch([&] { decode_onehot_any_kernel(a); }, dim3(grid), dim3(64), 0);
  return 0;
}

// dense v | h of any alphabet (vgh_dense_any_kernel)
int emu_vgh_any(const float* W, const float* c, int K, int M, int A, const float* hid, const float* hidp, int n, int Lh,
                float* act, float* prob, float* sample, uint64_t seed, uint32_t step, uint32_t off, int grid, int threads) {
  VghAnyArgs aa;
  VghArgs& a = aa.g;
  a.W = W; a.c = c; a.K = K; a.M = M; a.hid = hid; a.hidp = hidp; a.n = n; a.Lh = Lh; a.L = Lh + M - 1;
  a.TS = 1; a.divL = make_fastdiv((uint32_t)a.L);
  a.act = act; a.prob = prob; a.sample = sample;
  a.rng = make_rng(seed, step, off); a.kind = KIND_API_V;
  aa.A = A;
  emu::launch([&] { vgh_dense_any_kernel(aa); }, dim3(grid), dim3(threads), (size_t)A * threads * 4);
  return 0;
}

int emu_pack_hidden(float* dense, uint32_t* masks, uint32_t* flags, int n, int K, int Lh, int NW, int unpack) {
  HiddenPackArgs a{dense, masks, flags, n, K, Lh, NW};
  if (unpack) emu::launch([&] { unpack_hidden_kernel(a); }, dim3(2), dim3(64), 0);
  else emu::launch([&] { pack_hidden_kernel(a); }, dim3(2), dim3(64), 0);
  return 0;
}

int emu_tables(int id, const float* W, const float* b, const float* c, float* out) {
  TablesArgs a{W, b, c, out};
  CFG_DISPATCH(id, emu::launch([&] { build_tables_body<C>(a); }, dim3(3), dim3(64), 0));
  return 0;
}

int emu_hgv(int id, const float* tables, const uint32_t* letters, int n, int L, int mode, float* act, float* prob,
            float* sample, unsigned long long* ones, uint64_t seed, uint32_t step, uint32_t off, uint32_t kind,
            int TS, int grid, int threads) {
  HgvArgs a;
  a.tables = tables; a.letters = letters; a.n = n; a.L = L; a.LW = letter_words(L);
  a.TS = TS; a.mode = mode;
  a.act = act; a.prob = prob; a.sample = sample; a.ones = ones;
  a.rng = make_rng(seed, step, off); a.kind = kind;
  CFG_DISPATCH(id, (a.Lh = L - C::M + 1, a.divLh = make_fastdiv((uint32_t)a.Lh),
                    emu::launch([&] { hgv_body<C>(a); }, dim3(grid), dim3(threads),
                                (size_t)C::TAB * 4)));
  return 0;
}

// the table images of all slabs of a larger model in one launch (slab_tables_body; blockIdx.y = slab)
int emu_slab_tables(int id, const float* W, const float* b, const float* c, float* out, int K, int last_k0, int nslab) {
  SlabTablesArgs a;
  a.t.W = W; a.t.b = b; a.t.c = c; a.t.out = out;
  CFG_DISPATCH(id, (a.plan.Ks = C::K, a.plan.K = K, a.plan.last_k0 = last_k0, a.stride = C::TABLES_ALL,
                    emu::launch([&] { slab_tables_body<C>(a); }, dim3(2, nslab), dim3(64), 0)));
  return 0;
}

// h | v of a larger model's chain, all slabs in one launch (crbm_api.hip, slab_launch_hgv): the sample as bits of the larger
// model's mask rows
int emu_slab_hgv(int id, const float* tables, const uint32_t* letters, int n, int L, int mode, uint32_t* masks, int NWfull,
                 int Kfull, int last_k0, int nslab, unsigned long long* ones, uint64_t seed, uint32_t step, uint32_t off,
                 uint32_t kind, int TS, int grid, int threads) {
  SlabHgvArgs sa;
  HgvArgs& a = sa.m.g;
  a.tables = tables; a.letters = letters; a.n = n; a.L = L; a.LW = letter_words(L);
  a.TS = TS; a.mode = mode;
  a.act = nullptr; a.prob = nullptr; a.sample = nullptr; a.ones = ones;
  a.rng = make_rng(seed, step, off); a.kind = kind;
  sa.m.masks = masks; sa.m.NWfull = NWfull; sa.m.Kfull = Kfull; sa.m.k0 = 0; sa.m.kskip = 0; sa.m.group0 = 0;
  CFG_DISPATCH(id, (a.Lh = L - C::M + 1, a.divLh = make_fastdiv((uint32_t)a.Lh),
                    sa.plan.Ks = C::K, sa.plan.K = Kfull, sa.plan.last_k0 = last_k0, sa.table_stride = C::TABLES_ALL,
                    emu::launch([&] { slab_hgv_body<C>(sa); }, dim3(grid, nslab), dim3(threads), (size_t)C::TAB * 4)));
  return 0;
}

// free energy of a larger model slab by slab (slab_fe_body, blockIdx.y = slab) and the combine (slab_fe_combine_kernel)
int emu_slab_fe(int id, const float* tables, const uint32_t* letters, int n, int L, float* scratch, int K, int last_k0, int nslab,
                float* fe, float* fem, int grid, int threads) {
  SlabFeArgs sa;
  sa.a.tables = tables; sa.a.letters = letters; sa.a.n = n; sa.a.L = L; sa.a.LW = letter_words(L);
  sa.a.fe = nullptr; sa.a.fem = scratch; sa.pad_ = 0;
  SlabFeCombineArgs ca;
  ca.scratch = scratch; ca.letters = letters; ca.n = n; ca.L = L; ca.LW = letter_words(L);
  ca.K = K; ca.last_k0 = last_k0; ca.nslab = nslab; ca.fe = fe; ca.fem = fem;
  CFG_DISPATCH(id, (sa.a.Lh = L - C::M + 1, sa.table_stride = C::TABLES_ALL, sa.fem_stride = (long long)n * C::K,
                    ca.Ks = C::K, ca.c_log2e = tables + C::OFF_C,
                    emu::launch([&] { slab_fe_body<C>(sa); }, dim3(grid, nslab), dim3(threads), (size_t)C::TAB * 4)));
  emu::launch([&] { slab_fe_combine_kernel(ca); }, dim3(grid), dim3(threads), 0);
  return 0;
}

// column reduction of all slabs' partial rows into their columns of the larger model's sums (slab_reduce_kernel)
int emu_slab_reduce(const float* partials, float* sums, int nrows, int row, int Ks, int last_k0, int nslab, int K, int M4, int ds,
                    int want_sparsity, int skip_begin, int skip_len, float n_value, int threads) {
  SlabReduceArgs a;
  a.partials = partials; a.sums = sums; a.nrows = nrows; a.row = row;
  a.Ks = Ks; a.k0 = last_k0; a.K = K; a.M4 = M4; a.ds = ds; a.want_sparsity = want_sparsity;
  a.partial_stride = (long long)nrows * row;
  a.skip_begin = skip_begin; a.skip_len = skip_len; a.n_value = n_value;
  emu::launch([&] { slab_reduce_kernel(a); }, dim3((row + 31) / 32, nslab), dim3(threads), 0);
  return 0;
}

int emu_vgh(const float* W, const float* c, int K, int M, const float* hid, const float* hidp,
            int n, int Lh, float* act, float* prob, float* sample, uint64_t seed, uint32_t step, uint32_t off,
            int TS, int grid, int threads) {
  VghArgs a;
  a.W = W; a.c = c; a.K = K; a.M = M;
  a.hid = hid; a.hidp = hidp; a.n = n; a.Lh = Lh; a.L = Lh + M - 1;
  a.TS = TS; a.divL = make_fastdiv((uint32_t)a.L);
  a.act = act; a.prob = prob; a.sample = sample;
  a.rng = make_rng(seed, step, off); a.kind = KIND_API_V;
  emu::launch([&] { vgh_dense_kernel(a); }, dim3(grid), dim3(threads), (size_t)M * K * 16);
  return 0;
}

// returns LWs (letter words per chain row of vout) or -1
// sparse != 0: set-bit walk; 0: dense top-down tables (returns -3 if the model has none)
int emu_gibbs(int id, const float* tables, uint32_t* hm, uint32_t* hmp, uint32_t* vout, int nchains, int Lf, int S,
              int steps, uint64_t seed, uint32_t step, uint32_t off, int grid, int threads, int sparse,
              uint32_t* ones) {
  GibbsArgs a;
  a.tables = tables; a.hm = hm; a.hmp = hmp; a.vout = vout; a.ones = ones; a.debug = 0; a.nblocks = 0; a.stats_off = 0;
  a.nchains = nchains; a.Lf = Lf; a.S = S; a.steps = steps; a.rng = make_rng(seed, step, off);
  int lws = -1;
  CFG_DISPATCH(id, {
    const ModelShape ms = shape_of<C>();
    if (!sparse && !C::DENSE) return -3;
    const GibbsLayout gl = gibbs_layout(ms, Lf, S, sparse != 0);
    a.Lv = gl.Lv; a.nvb = gl.nvb; a.nhb = gl.nhb; a.Lrow = gl.Lrow; a.LWs = gl.LWs;
    a.divVB = make_fastdiv((uint32_t)gl.nvb); a.divHB = make_fastdiv((uint32_t)gl.nhb);
    a.divRow = make_fastdiv((uint32_t)(gl.Lrow * ms.NW)); a.divLfw = make_fastdiv((uint32_t)(Lf * ms.NW));
    if (!C::DS) a.hmp = nullptr;
    lws = gl.LWs;
    if (vout) {
      if (sparse) emu::launch([&] { gibbs_body<C, true>(a); }, dim3(grid), dim3(threads), (size_t)gl.lds_bytes);
      else emu::launch([&] { gibbs_body<C, C::DENSE ? false : true>(a); }, dim3(grid), dim3(threads), (size_t)gl.lds_bytes);
    }
  });
  return lws;
}

// fills the StatsGeom of a layout
static StatsGeom geom_of(const StatsMfmaLayout& st, float* partials, long ngroups, size_t lds_bytes) {
  StatsGeom g;
  g.GPC = st.GPC;
  g.lds_floats = (int)(lds_bytes / 4);
  g.off_slices = st.off_slices; g.slice = st.slice;
  g.off_win = st.off_win; g.off_gw = st.off_gw; g.off_pt = st.off_pt;
  g.divGPC = make_fastdiv((uint32_t)st.GPC, (uint64_t)ngroups);
  g.row = st.row; g.off_vh0 = st.off_vh[0]; g.off_vh1 = st.off_vh[1]; g.off_h0 = st.off_h[0]; g.off_h1 = st.off_h[1];
  g.off_sw = st.off_sw; g.off_sb = st.off_sb; g.off_v = st.off_v;
  g.partials = partials;
  return g;
}

// column sums of the partial rows on the host, same validity rules as reduce_partials_kernel
static void host_reduce(const float* partials, int rows, int row, int K, int KAM, int ds, int want_sparsity, int skip_begin,
                        int skip_len, float n_value, float* sums) {
  const int sb = skip_begin < 0 ? row : skip_begin, sl = skip_begin < 0 ? 0 : skip_len;
  for (int r = 0; r < row; ++r) {
    if (r >= sb && r < sb + sl) continue;
    bool valid;
    if (r < KAM) valid = true;
    else if (r < 2 * KAM) valid = ds != 0;
    else if (r < 2 * KAM + K) valid = true;
    else if (r < 2 * KAM + 2 * K) valid = ds != 0;
    else if (r < 3 * KAM + 3 * K) valid = want_sparsity != 0;
    else valid = true;
    float t = 0.f;
    if (valid)
      for (int i = 0; i < rows; ++i) t += partials[(size_t)i * row + r];
    sums[r < sb ? r : r - sl] = t;
  }
  sums[row - sl] = n_value;
}

// MFMA statistics of one half + deterministic reduction into `sums` (row floats + n).
// threads > 0 overrides the block size (a multiple of 64 * roles).
int emu_stats_mfma(int id, const float* tables, const uint32_t* letters, int n, int L, int LW, int want_sparsity,
                   int threads, int gx, float* partials, int partials_cap, float* sums, int skip_begin, int skip_len) {
  StatsMfmaArgs a;
  a.tables = tables; a.letters = letters; a.n = n; a.L = L; a.LW = LW;
  int row = -1;
  CFG_DISPATCH(id, {
    const ModelShape ms = shape_of<C>();
    const int Lh = L - C::M + 1;
    a.Lh = Lh;
    const StatsMfmaLayout st = stats_mfma_layout(ms, want_sparsity, Lh, threads, C::TAB * 4);
    if ((st.threads / 64) % st.NR != 0) return -3;
    const long ngroups = (long)n * st.GPC;
    const long nunits = (ngroups + 1) / 2;
    const int wpr = (st.threads / 64) / st.NR;
    if (gx > (nunits + wpr - 1) / wpr) gx = (int)((nunits + wpr - 1) / wpr);
    if ((long)gx * st.row > partials_cap) return -2;
    for (size_t i = 0; i < (size_t)gx * st.row; ++i) partials[i] = 1e30f;   // never cleared on the GPU either
    a.off_tab = st.region_floats;
    a.debug = 0;
    a.nblocks = 0;
    const size_t lds = std::max((size_t)st.region_floats * 4 + (size_t)C::TAB * 4, (size_t)st.combine_bytes);
    a.sg = geom_of(st, partials, ngroups, lds);
    if (want_sparsity) emu::launch([&] { stats_mfma_body<C, true>(a); }, dim3(gx), dim3(st.threads), lds);
    else emu::launch([&] { stats_mfma_body<C, false>(a); }, dim3(gx), dim3(st.threads), lds);
    host_reduce(partials, gx, st.row, C::K, C::K * 4 * C::M, C::DS, want_sparsity, skip_begin, skip_len, (float)n, sums);
    row = st.row;
  });
  return row;
}

}  // extern "C"

// The fused launch: `steps` Gibbs steps whose last h|v pass also leaves the model half of the
// statistics (gibbs_body<C, true, true>); the partial rows are reduced into `sums` on the host.
// Returns the partial-row length, -3 if the model has no fused variant.
template <class C>
static int run_gibbs_stats(GibbsArgs a, int Lf, int S, int grid, int threads, float* partials, int partials_cap, float* sums,
                           int skip_begin, int skip_len) {
  if constexpr (C::FUSE_STATS) {
    const ModelShape ms = shape_of<C>();
    const GibbsLayout gl = gibbs_layout(ms, Lf, S, true);
    a.Lv = gl.Lv; a.nvb = gl.nvb; a.nhb = gl.nhb; a.Lrow = gl.Lrow; a.LWs = gl.LWs;
    a.divVB = make_fastdiv((uint32_t)gl.nvb); a.divHB = make_fastdiv((uint32_t)gl.nhb);
    a.divRow = make_fastdiv((uint32_t)(gl.Lrow * ms.NW)); a.divLfw = make_fastdiv((uint32_t)(Lf * ms.NW));
    if (!C::DS) a.hmp = nullptr;
    const StatsMfmaLayout st = stats_mfma_layout(ms, 0, Lf, threads, 0, false);
    if ((long)grid * st.row > partials_cap) return -2;
    for (size_t i = 0; i < (size_t)grid * st.row; ++i) partials[i] = 1e30f;
    a.stats_off = (gl.lds_bytes / 4 + 3) & ~3;
    const size_t lds = std::max((size_t)(a.stats_off + st.region_floats) * 4, (size_t)st.combine_bytes);
    a.sg = geom_of(st, partials, (long)S * st.GPC, lds);
    emu::launch([&] { gibbs_body<C, true, true>(a); }, dim3(grid), dim3(threads), lds);
    host_reduce(partials, grid, st.row, C::K, C::K * 4 * C::M, C::DS, 0, skip_begin, skip_len, (float)a.nchains, sums);
    return st.row;
  } else {
    return -3;
  }
}

// The whole local phase of a training step in one launch (train_local_body): chain + model half in
// `ggrid` blocks, data half in `dgrid` blocks, interleaved; both partial sets reduced on the host.
template <class C>
static int run_train_local(GibbsArgs g, int Lf, int S, int ggrid, int dgrid, int threads, const uint32_t* letters, int n, int L,
                           int LW, float* partials_m, float* partials_d, int cap, float* sums_d, float* sums_m, int skip_begin,
                           int skip_len) {
  if constexpr (C::FUSE_STATS) {
    const ModelShape ms = shape_of<C>();
    const GibbsLayout gl = gibbs_layout(ms, Lf, S, true);
    g.Lv = gl.Lv; g.nvb = gl.nvb; g.nhb = gl.nhb; g.Lrow = gl.Lrow; g.LWs = gl.LWs;
    g.divVB = make_fastdiv((uint32_t)gl.nvb); g.divHB = make_fastdiv((uint32_t)gl.nhb);
    g.divRow = make_fastdiv((uint32_t)(gl.Lrow * ms.NW)); g.divLfw = make_fastdiv((uint32_t)(Lf * ms.NW));
    if (!C::DS) g.hmp = nullptr;
    const StatsMfmaLayout sm = stats_mfma_layout(ms, 0, Lf, threads, 0, false);
    const int tabs = C::TAB * 4, Lh = L - C::M + 1;
    const StatsMfmaLayout sd = stats_mfma_layout(ms, 1, Lh, threads, tabs, false);
    const long ngroups = (long)n * sd.GPC;
    const int wpr = (sd.threads / 64) / sd.NR;
    const long nunits = (ngroups + 1) / 2;
    if (dgrid > (nunits + wpr - 1) / wpr) dgrid = (int)((nunits + wpr - 1) / wpr);
    if ((long)ggrid * sm.row > cap || (long)dgrid * sd.row > cap) return -2;
    for (size_t i = 0; i < (size_t)ggrid * sm.row; ++i) partials_m[i] = 1e30f;
    for (size_t i = 0; i < (size_t)dgrid * sd.row; ++i) partials_d[i] = 1e30f;
    g.nblocks = ggrid;
    g.stats_off = (gl.lds_bytes / 4 + 3) & ~3;
    const size_t glds = std::max((size_t)(g.stats_off + sm.region_floats) * 4, (size_t)sm.combine_bytes);
    const size_t dlds = std::max((size_t)sd.region_floats * 4 + tabs, (size_t)sd.combine_bytes);
    const size_t lds = std::max(glds, dlds);
    g.sg = geom_of(sm, partials_m, (long)S * sm.GPC, lds);
    TrainLocalArgs t;
    t.g = g;
    t.d.tables = g.tables; t.d.letters = letters; t.d.n = n; t.d.L = L; t.d.Lh = Lh; t.d.LW = LW;
    t.d.off_tab = sd.region_floats; t.d.debug = 0; t.d.nblocks = dgrid;
    t.d.sg = geom_of(sd, partials_d, ngroups, lds);
    emu::launch([&] { train_local_body<C>(t); }, dim3(ggrid + dgrid), dim3(threads), lds);
    host_reduce(partials_m, ggrid, sm.row, C::K, C::K * 4 * C::M, C::DS, 0, skip_begin, skip_len, (float)g.nchains, sums_m);
    host_reduce(partials_d, dgrid, sd.row, C::K, C::K * 4 * C::M, C::DS, 1, -1, 0, (float)n, sums_d);
    return sm.row;
  } else {
    return -3;
  }
}

extern "C" int emu_train_local(int id, const float* tables, uint32_t* hm, uint32_t* hmp, uint32_t* vout, int nchains, int Lf, int S,
                               int steps, uint64_t seed, uint32_t step, uint32_t off, int ggrid, int dgrid, int threads,
                               const uint32_t* letters, int n, int L, int LW, float* partials_m, float* partials_d, int cap,
                               float* sums_d, float* sums_m, int skip_begin, int skip_len) {
  GibbsArgs g;
  g.tables = tables; g.hm = hm; g.hmp = hmp; g.vout = vout; g.ones = nullptr; g.debug = 0;
  g.nchains = nchains; g.Lf = Lf; g.S = S; g.steps = steps; g.rng = make_rng(seed, step, off);
  int row = -1;
  CFG_DISPATCH(id, row = run_train_local<C>(g, Lf, S, ggrid, dgrid, threads, letters, n, L, LW, partials_m, partials_d, cap,
                                            sums_d, sums_m, skip_begin, skip_len));
  return row;
}

extern "C" {

int emu_gibbs_stats(int id, const float* tables, uint32_t* hm, uint32_t* hmp, uint32_t* vout, int nchains, int Lf, int S,
                    int steps, uint64_t seed, uint32_t step, uint32_t off, int grid, int threads, float* partials,
                    int partials_cap, float* sums, int skip_begin, int skip_len) {
  GibbsArgs a;
  a.tables = tables; a.hm = hm; a.hmp = hmp; a.vout = vout; a.ones = nullptr; a.debug = 0; a.nblocks = 0;
  a.nchains = nchains; a.Lf = Lf; a.S = S; a.steps = steps; a.rng = make_rng(seed, step, off);
  int row = -1;
  CFG_DISPATCH(id, row = run_gibbs_stats<C>(a, Lf, S, grid, threads, partials, partials_cap, sums, skip_begin, skip_len));
  return row;
}

int emu_reduce(const float* partials, float* sums, int nrows, int row, int K, int KAM, int ds, int want,
               int skip_begin, int skip_len, float n_value) {
  // a synthetic layout: K = 1, KAM = 2 -> row = 3*2 + 3 + 4 = 13 columns when row == 13
  ReduceArgs r{partials, sums, nrows, row, K, KAM, ds, want, skip_begin, skip_len, n_value};
  emu::launch([&] { reduce_partials_kernel(r); }, dim3((row + 31) / 32), dim3(1024), 0);
  return 0;
}

int emu_update(const float* sums, float* W, float* b, float* c, float* vW, float* vb, float* vc, int K, int M,
               int ds, int L_data, int Lf, float lr, float momentum, float rho, float lambda_rate) {
  const SumsLayout sl = sums_layout(K, M);
  UpdateArgs u{sums, W, b, c, vW, vb, vc, W, b, c, vW, vb, vc, K, M, ds, L_data, Lf, sl.data_off, sl.n_d, sl.model_off, sl.n_m,
               lr, momentum, rho, lambda_rate};
  emu::launch([&] { apply_update_kernel(u); }, dim3(1), dim3(64), 0);
  return 0;
}

// the fused end of a training step: update + table images of the new parameters; `grid` blocks, the
// new parameters and velocities go to the o* buffers (the old ones stay untouched)
int emu_update_tables(int id, const float* sums, const float* W, const float* b, const float* c, const float* vW,
                      const float* vb, const float* vc, float* oW, float* ob, float* oc, float* ovW, float* ovb, float* ovc,
                      int L_data, int Lf, float lr, float momentum, float rho, float lambda_rate, float* tables, int grid,
                      int threads) {
  CFG_DISPATCH(id, {
    const SumsLayout sl = sums_layout(C::K, C::M);
    UpdateTablesArgs a;
    a.u = UpdateArgs{sums, W, b, c, vW, vb, vc, oW, ob, oc, ovW, ovb, ovc, C::K, C::M, C::DS, L_data, Lf,
                     sl.data_off, sl.n_d, sl.model_off, sl.n_m, lr, momentum, rho, lambda_rate};
    a.tables = tables;
    emu::launch([&] { update_tables_body<C>(a); }, dim3(grid), dim3(threads), (size_t)(C::K * 4 * C::M + C::K + 4) * 4);
  });
  return 0;
}

// The all-reduce through mapped buffers with `nranks` emulated ranks in one process: every rank's sums are
// published into its own buffer (publish_sums_kernel: sums, fence, flag), then ONE rank's update launch waits
// for all flags and adds the copies in rank order (update_tables_ipc_body).  bufs: nranks x stride floats,
// flags: nranks words, status: one word.
int emu_ipc_update(int id, int nranks, const float* rank_sums, int count, float* bufs, int stride, uint32_t* flags,
                   uint32_t* status, uint32_t step_value, const float* W, const float* b, const float* c, const float* vW,
                   const float* vb, const float* vc, float* oW, float* ob, float* oc, float* ovW, float* ovb, float* ovc,
                   int L_data, int Lf, float lr, float momentum, float rho, float lambda_rate, float* tables, int grid, int threads,
                   int silent_rank) {   // silent_rank >= 0: that rank never publishes (the update must time out, not hang)
  for (int r = 0; r < nranks; ++r) {
    if (r == silent_rank) continue;
    PublishArgs pa{};     // rank r pushes into its slot of the observing rank's buffer
    pa.src = rank_sums + (size_t)r * count; pa.dst[0] = bufs + (size_t)r * stride; pa.flag[0] = flags + r;
    pa.value = step_value; pa.count = count; pa.n = 1;
    emu::launch([&] { publish_sums_kernel(pa); }, dim3(1), dim3(64), 0);
  }
  CFG_DISPATCH(id, {
    const SumsLayout sl = sums_layout(C::K, C::M);
    if (count != sl.count || nranks > IPC_MAX_RANKS) return -2;
    UpdateIpcArgs a;
    a.ut.u = UpdateArgs{nullptr, W, b, c, vW, vb, vc, oW, ob, oc, ovW, ovb, ovc, C::K, C::M, C::DS, L_data, Lf,
                        sl.data_off, sl.n_d, sl.model_off, sl.n_m, lr, momentum, rho, lambda_rate};
    a.ut.tables = tables;
    for (int r = 0; r < IPC_MAX_RANKS; ++r) {
      a.ipc.sums[r] = bufs + (size_t)(r < nranks ? r : 0) * stride;
      a.ipc.flags[r] = flags + (r < nranks ? r : 0);
    }
    a.ipc.status = status; a.ipc.expect = step_value; a.ipc.nranks = nranks; a.ipc.count = count;
    a.ipc.timeout_ticks = 300000;    // 0.3 s of the emulator's microsecond clock
    emu::launch([&] { update_tables_ipc_body<C>(a); }, dim3(grid), dim3(threads),
                (size_t)(C::K * 4 * C::M + C::K + 4 + 4) * 4);
  });
  return 0;
}

int emu_free_energy(int id, const float* tables, const uint32_t* letters, int n, int L, float* fe, float* fem,
                    int grid, int threads) {
  FeArgs a;
  a.tables = tables; a.letters = letters; a.n = n; a.L = L; a.LW = letter_words(L);
  a.fe = fe; a.fem = fem;
  CFG_DISPATCH(id, (a.Lh = L - C::M + 1,
                    emu::launch([&] { free_energy_body<C>(a); }, dim3(grid), dim3(threads),
                                (size_t)C::TAB * 4)));
  return 0;
}

int emu_hit_summary(int id, const float* tables, const uint32_t* letters, int n, int L, float* hmax, float* hsum,
                     float* pos, int grid, int threads) {
  HitArgs a;
  a.tables = tables; a.letters = letters; a.n = n; a.L = L; a.LW = letter_words(L);
  a.hmax = hmax; a.inv_Lh = 1.0f;   // the harness divides by Lh itself
  std::vector<unsigned long long> pos_fx, hsum_fx;
  int chunks = 1, K = 0, Lh = 0;
  CFG_DISPATCH(id, (a.Lh = Lh = L - C::M + 1, K = C::K, chunks = (a.Lh + 64 * C::HIT_NI - 1) / (64 * C::HIT_NI),
                    pos_fx.assign((size_t)C::K * a.Lh, 0ull), hsum_fx.assign((size_t)n * C::K, 0ull),
                    a.pos_fx = pos_fx.data(), a.hsum = chunks == 1 ? hsum : nullptr, a.hsum_fx = chunks > 1 ? hsum_fx.data() : nullptr,
                    emu::launch([&] { hit_summary_body<C>(a); }, dim3(grid, chunks),
                                dim3(threads), (size_t)C::TAB * 4 + (size_t)64 * C::HIT_NI * C::K * 8)));
  // fixed point (HIT_FX units) -> the floats the harness compares
  for (size_t i = 0; i < pos_fx.size(); ++i) pos[i] = (float)((double)pos_fx[i] / (double)HIT_FX);
  if (chunks > 1)
    for (size_t i = 0; i < hsum_fx.size(); ++i) hsum[i] = (float)((double)hsum_fx[i] / (double)HIT_FX);
  (void)K; (void)Lh;
  return 0;
}


// ---- the generic ("big") kernels: run-time K and M -------------------------------------------------------------------
static BigModel big_model_of(const float* W, const float* b, const float* c, int K, int M, int ds, int A) {
  BigModel m;
  m.W = W; m.b = b; m.c = c; m.K = K; m.M = M; m.ds = ds; m.NW = (K + 31) / 32; m.A = A;
  return m;
}

int emu_big_hgv(const float* W, const float* b, const float* c, int K, int M, int ds, const uint32_t* letters, int n, int L,
                int mode, float* act, float* prob, float* sample, unsigned long long* ones, uint32_t* masks, uint64_t seed,
                uint32_t step, uint32_t off, uint32_t kind, int TS, int KS, int grid, int threads, int pool, int A) {
  BigHgvArgs a;
  a.m = big_model_of(W, b, c, K, M, ds, A);
  a.pool = pool;
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = letter_words_any(A, L);
  a.TS = TS; a.KS = KS; a.mode = mode;
  a.split = (K > 32 && (n % 2) == 0) ? 1 : 0;      // both forms are exercised: blockIdx.y = mask word where a model has several
  a.act = act; a.prob = prob; a.sample = sample; a.ones = ones; a.masks = masks;
  a.rng = make_rng(seed, step, off); a.kind = kind;
  const size_t lds = ((size_t)big_hgv_ksp(KS) * M * A + 32) * 4 + (size_t)TS * a.LW * 4;
  const dim3 g(grid, a.split ? (K + 31) / 32 : 1);
  if (pool > 1) {
    if (KS > 8) return -2;
    emu::launch([&] { big_hgv_pooled_kernel(a); }, g, dim3(threads), lds);
  } else
    emu::launch([&] { big_hgv_kernel(a); }, g, dim3(threads), lds);
  return 0;
}

// one Gibbs step: v | h from the masks, then h | v per strand into the masks; returns the letter words per row of vout
int emu_big_gibbs_step(const float* W, const float* b, const float* c, int K, int M, int ds, uint32_t* hm, uint32_t* hmp,
                       uint32_t* vout, int nchains, int Lf, uint64_t seed, uint32_t step, uint32_t off, int JS, int KS,
                       int grid, int threads, int pool, int A) {
  const int Lv = Lf + M - 1, LWs = letter_words_any(A, Lv);
  if (!vout) return LWs;
  BigVghArgs v;
  v.m = big_model_of(W, b, c, K, M, ds, A);
  v.hm = hm; v.hmp = ds ? hmp : nullptr; v.vout = vout;
  v.nchains = nchains; v.Lf = Lf; v.Lv = Lv; v.LWs = LWs; v.JS = JS;
  v.rng = make_rng(seed, step, off);
  if (A == 4) emu::launch([&] { big_vgh_kernel(v); }, dim3(grid), dim3(threads), (size_t)JS * 32 * 16 + (size_t)BIG_VR * threads + (size_t)2 * (std::min(BIG_VR * threads, Lv) + M - 1) * 4);
  else emu::launch([&] { big_vgh_any_kernel(v); }, dim3(grid), dim3(threads), ((size_t)JS * 32 * A + (size_t)A * threads) * 4 + threads);
  for (int strand = 0; strand <= ds; ++strand)
    emu_big_hgv(W, b, c, K, M, ds, vout, nchains, Lv, strand, nullptr, nullptr, nullptr, nullptr, strand ? hmp : hm, seed, step, off,
                KIND_CHAIN_H, 2, KS, grid, threads, pool, A);
  return LWs;
}

// raw sums of one half: partial rows through big_stats_kernel, then the host column reduce of the harness
int emu_big_stats(const float* W, const float* b, const float* c, int K, int M, int ds, const uint32_t* letters, int n, int L,
                  int LW, int want_sparsity, int R, int CH, int threads, float* sums, int skip_begin, int skip_len, int pool, int A) {
  const int KAM = K * A * M, row = 3 * KAM + 3 * K + A;
  std::vector<float> partials((size_t)R * row, -777.0f);      // what a launch does not write must not be read
  BigStatsArgs a;
  a.m = big_model_of(W, b, c, K, M, ds, A);
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = LW;
  a.want_sparsity = want_sparsity; a.R = R; a.CH = CH; a.pool = pool;
  if (CH % pool != 0) return -3;
  a.partials = partials.data();
  a.row = row; a.off_vh0 = 0; a.off_vh1 = KAM; a.off_h0 = 2 * KAM; a.off_h1 = 2 * KAM + K;
  a.off_sw = 2 * KAM + 2 * K; a.off_sb = 3 * KAM + 2 * K; a.off_v = 3 * KAM + 3 * K;
  if (A * M > BIG_ST * threads) return -2;
  emu::launch([&] { big_stats_kernel(a); }, dim3(K, R), dim3(threads),
              (((size_t)A * M + 3) & ~(size_t)3) * 4 + (size_t)(pool > 1 ? 5 : 3) * CH * 4 + 64 + (size_t)12 * threads * 4 +
                  (size_t)((A + 3) & ~3) * 4 + (size_t)CH + M);
  host_reduce(partials.data(), R, row, K, KAM, ds, want_sparsity, skip_begin, skip_len, (float)n, sums);
  return row;
}

int emu_big_update(const float* sums, float* W, float* b, float* c, float* vW, float* vb, float* vc, int K, int M, int ds,
                   int L_data, int Lf, float lr, float momentum, float rho, float lambda_rate, int grid, int threads, int A) {
  const SumsLayout sl = sums_layout(K, M, A);
  UpdateArgs u{sums, W, b, c, vW, vb, vc, W, b, c, vW, vb, vc, K, M, ds, L_data, Lf,
               sl.data_off, sl.n_d, sl.model_off, sl.n_m, lr, momentum, rho, lambda_rate, A};
  emu::launch([&] { big_update_kernel(u); }, dim3(grid), dim3(threads), 0);
  return 0;
}

int emu_big_eval(const float* W, const float* b, const float* c, int K, int M, int ds, const uint32_t* letters, int n, int L,
                 int hits, float* fe, float* fem, float* hmax, float* hmean, float* pos, int grid, int threads, int pool, int A) {
  BigEvalArgs a;
  a.pool = pool;
  a.m = big_model_of(W, b, c, K, M, ds, A);
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = letter_words_any(A, L);
  a.fe = fe; a.fem = fem; a.hmax = hmax; a.hmean = hmean; a.hits = hits;
  std::vector<unsigned long long> pos_fx((size_t)K * a.Lh, 0ull);
  a.pos_fx = (hits && pos) ? pos_fx.data() : nullptr;
  emu::launch([&] { big_eval_kernel(a); }, dim3(grid), dim3(threads), (((size_t)A * M + 3) & ~(size_t)3) * 4 + 64 + (size_t)L);
  if (hits && pos)
    for (size_t i = 0; i < pos_fx.size(); ++i) pos[i] = (float)((double)pos_fx[i] / (double)HIT_FX);
  return 0;
}

int emu_sums_layout_any(int K, int M, int A, int* out);
int emu_sums_layout(int K, int M, int* out) { return emu_sums_layout_any(K, M, 4, out); }
int emu_sums_layout_any(int K, int M, int A, int* out) {
  const SumsLayout s = sums_layout(K, M, A);
  out[0] = s.data_off; out[1] = s.n_d; out[2] = s.model_off; out[3] = s.n_m; out[4] = s.count;
  out[5] = s.model_skip_begin; out[6] = s.model_skip_len;
  return 0;
}

int emu_letter_words(int L) { return letter_words(L); }
int emu_letter_words_any(int A, int L) { return letter_words_any(A, L); }

}  // extern "C"
