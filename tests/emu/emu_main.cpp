// TEST INFRASTRUCTURE: runs the HIP kernels of crbm_amd/csrc/crbm_kernels.h on
// CPU threads (see shim/hip/hip_runtime.h) so that indexing, LDS layouts and
// barrier placement are checked under ASan/UBSan before a kernel ever reaches
// a GPU.  Exposes plain C entry points for tests/test_emu.py (ctypes).
#define CRBM_DEFINE_MISC_KERNELS
#include "crbm_kernels.h"

#include <thread>
#include <vector>

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx;

template <typename Args>
void launch(void (*kernel)(Args), dim3 grid, dim3 block, size_t lds, const Args& args) {
  const unsigned nthr = block.x, nwaves = (block.x + 63) / 64;
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
      BlockCtx ctx;
      pthread_barrier_init(&ctx.bar, nullptr, nthr);
      std::vector<pthread_barrier_t> wb(nwaves);
      for (unsigned w = 0; w < nwaves; ++w) pthread_barrier_init(&wb[w], nullptr, std::min(64u, nthr - w * 64));
      std::vector<float> scratch(nwaves * 64);
      // exact size: out-of-bounds LDS accesses trip AddressSanitizer
      std::vector<unsigned char> smem(lds ? lds : 1, 0xAB);
      ctx.wave_bar = wb.data();
      ctx.wave_scratch = scratch.data();
      ctx.smem = smem.data();
      std::vector<std::thread> threads;
      threads.reserve(nthr);
      for (unsigned t = 0; t < nthr; ++t)
        threads.emplace_back([&, t]() {
          t_threadIdx = dim3(t, 0, 0);
          t_blockIdx = dim3(bx, by, 0);
          t_blockDim = block;
          t_gridDim = grid;
          t_ctx = &ctx;
          kernel(args);
        });
      for (auto& th : threads) th.join();
      pthread_barrier_destroy(&ctx.bar);
      for (auto& b : wb) pthread_barrier_destroy(&b);
    }
}
}  // namespace emu

using namespace crbm;

namespace {
ModelView make_mv(const float* W, const float* b, const float* c, int K, int M, int G, int ds) {
  ModelView mv;
  mv.W = W; mv.b = b; mv.c = c; mv.K = K; mv.M = M; mv.G = G;
  mv.ngroups = (M + G - 1) / G; mv.rows = pow4(G); mv.ds = ds;
  return mv;
}
RngView make_rng(uint64_t seed, uint32_t step, uint32_t off) {
  RngView r;
  r.seed_lo = (uint32_t)seed; r.seed_hi = (uint32_t)(seed >> 32); r.step = step; r.seq_offset = off;
  return r;
}

#define NQ_DISPATCH(nq, CALL)                \
  switch (nq) {                              \
    case 1: { constexpr int NQ = 1; CALL; break; }   \
    case 2: { constexpr int NQ = 2; CALL; break; }   \
    case 3: { constexpr int NQ = 3; CALL; break; }   \
    case 5: { constexpr int NQ = 5; CALL; break; }   \
    case 13: { constexpr int NQ = 13; CALL; break; } \
    default: return -1;                      \
  }
}  // namespace

extern "C" {

int emu_encode(const float* v, uint32_t* letters, uint32_t* flags, int n, int L, int grid) {
  EncodeArgs a{v, letters, flags, n, L, letter_words(L)};
  emu::launch(encode_onehot_kernel, dim3(grid), dim3(64), 0, a);
  return 0;
}

int emu_decode(const uint32_t* letters, float* v, int n, int L, int grid) {
  DecodeArgs a{letters, v, n, L, letter_words(L)};
  emu::launch(decode_onehot_kernel, dim3(grid), dim3(64), 0, a);
  return 0;
}

int emu_pack_hidden(float* dense, uint32_t* masks, uint32_t* flags, int n, int K, int Lh, int NW, int unpack) {
  HiddenPackArgs a{dense, masks, flags, n, K, Lh, NW};
  emu::launch(unpack ? unpack_hidden_kernel : pack_hidden_kernel, dim3(2), dim3(64), 0, a);
  return 0;
}

int emu_hgv(int nq, const float* W, const float* b, const float* c, int K, int M, int G, int ds,
            const uint32_t* letters, int n, int L, int mode, float* act, float* prob, float* sample,
            unsigned long long* ones, uint64_t seed, uint32_t step, uint32_t off, uint32_t kind,
            int TS, int grid, int threads) {
  HgvArgs a;
  a.mv = make_mv(W, b, c, K, M, G, ds);
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = letter_words(L);
  a.TS = TS; a.divLh = make_fastdiv((uint32_t)a.Lh); a.mode = mode;
  a.act = act; a.prob = prob; a.sample = sample; a.ones = ones;
  a.rng = make_rng(seed, step, off); a.kind = kind;
  const size_t lds = (size_t)(mode == 2 ? 2 : 1) * gather_table_floats(M, G, 4 * nq) * 4;
  NQ_DISPATCH(nq, emu::launch(hgv_kernel<NQ>, dim3(grid), dim3(threads), lds, a));
  return 0;
}

int emu_vgh(const float* W, const float* b, const float* c, int K, int M, const float* hid, const float* hidp,
            int n, int Lh, float* act, float* prob, float* sample, uint64_t seed, uint32_t step, uint32_t off,
            int TS, int grid, int threads) {
  VghArgs a;
  a.mv = make_mv(W, b, c, K, M, 1, hidp ? 1 : 0);
  a.hid = hid; a.hidp = hidp; a.n = n; a.Lh = Lh; a.L = Lh + M - 1;
  a.TS = TS; a.divL = make_fastdiv((uint32_t)a.L);
  a.act = act; a.prob = prob; a.sample = sample;
  a.rng = make_rng(seed, step, off); a.kind = KIND_API_V;
  emu::launch(vgh_dense_kernel, dim3(grid), dim3(threads), (size_t)M * K * 16, a);
  return 0;
}

int emu_gibbs(int nq, const float* W, const float* b, const float* c, int K, int M, int G, int ds,
              uint32_t* hm, uint32_t* hmp, uint32_t* vout, int nchains, int Lf, int S, int steps,
              uint64_t seed, uint32_t step, uint32_t off, int grid, int threads) {
  const GibbsLayout gl = gibbs_layout(K, M, ds, nq, G, Lf, S);
  GibbsArgs a;
  a.mv = make_mv(W, b, c, K, M, G, ds);
  a.hm = hm; a.hmp = ds ? hmp : nullptr; a.vout = vout;
  a.nchains = nchains; a.Lf = Lf; a.Lv = gl.Lv; a.Lhp = gl.Lhp; a.LWs = gl.LWs; a.S = S;
  a.divLv = make_fastdiv((uint32_t)gl.Lv); a.divLf = make_fastdiv((uint32_t)Lf);
  a.divRow = make_fastdiv((uint32_t)(gl.Lhp * gl.NW));
  a.steps = steps; a.rng = make_rng(seed, step, off);
  NQ_DISPATCH(nq, emu::launch(gibbs_kernel<NQ>, dim3(grid), dim3(threads), (size_t)gl.lds_bytes, a));
  return 0;
}

// statistics of one half + deterministic reduction into `sums` (row floats + n)
int emu_stats(int nq, const float* W, const float* b, const float* c, int K, int M, int G, int ds,
              const uint32_t* letters, int n, int L, int want_sparsity, int TS, int rows, int threads,
              float* partials, float* sums, int skip_begin, int skip_len) {
  const StatsLayout st = stats_layout(K, M, ds, nq, G, want_sparsity, threads);
  StatsArgs a;
  a.mv = make_mv(W, b, c, K, M, G, ds);
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = letter_words(L);
  a.TS = TS; a.divLh = make_fastdiv((uint32_t)a.Lh); a.divL = make_fastdiv((uint32_t)L);
  a.want_sparsity = want_sparsity; a.ntk = st.ntk; a.ntj = st.ntj; a.ntiles = st.ntiles;
  a.row = st.row; a.off_vh0 = st.off_vh[0]; a.off_vh1 = st.off_vh[1]; a.off_h0 = st.off_h[0]; a.off_h1 = st.off_h[1];
  a.off_sw = st.off_sw; a.off_sb = st.off_sb; a.off_v = st.off_v;
  a.partials = partials;
  memset(partials, 0, sizeof(float) * (size_t)rows * st.row);
  NQ_DISPATCH(nq, emu::launch(stats_kernel<NQ>, dim3(rows, st.grid_y), dim3(threads), (size_t)st.lds_bytes, a));
  ReduceArgs r{partials, sums, rows, st.row, skip_begin < 0 ? st.row : skip_begin, skip_begin < 0 ? 0 : skip_len, (float)n};
  emu::launch(reduce_partials_kernel, dim3((st.row + 63) / 64), dim3(64), 0, r);
  return st.row;
}

int emu_update(const float* sums, float* W, float* b, float* c, float* vW, float* vb, float* vc, int K, int M,
               int ds, int L_data, int Lf, float lr, float momentum, float rho, float lambda_rate) {
  const SumsLayout sl = sums_layout(K, M);
  UpdateArgs u{sums, W, b, c, vW, vb, vc, K, M, ds, L_data, Lf, sl.data_off, sl.n_d, sl.model_off, sl.n_m,
               lr, momentum, rho, lambda_rate};
  emu::launch(apply_update_kernel, dim3(1), dim3(64), 0, u);
  return 0;
}

int emu_free_energy(int nq, const float* W, const float* b, const float* c, int K, int M, int G, int ds,
                    const uint32_t* letters, int n, int L, float* fe, float* fem, int grid, int threads) {
  FeArgs a;
  a.mv = make_mv(W, b, c, K, M, G, ds);
  a.letters = letters; a.n = n; a.L = L; a.Lh = L - M + 1; a.LW = letter_words(L);
  a.fe = fe; a.fem = fem;
  const size_t lds = (size_t)(1 + ds) * gather_table_floats(M, G, 4 * nq) * 4;
  NQ_DISPATCH(nq, emu::launch(free_energy_kernel<NQ>, dim3(grid), dim3(threads), lds, a));
  return 0;
}

int emu_sums_layout(int K, int M, int* out) {
  const SumsLayout s = sums_layout(K, M);
  out[0] = s.data_off; out[1] = s.n_d; out[2] = s.model_off; out[3] = s.n_m; out[4] = s.count;
  out[5] = s.model_skip_begin; out[6] = s.model_skip_len;
  return 0;
}

int emu_letter_words(int L) { return letter_words(L); }

}  // extern "C"
