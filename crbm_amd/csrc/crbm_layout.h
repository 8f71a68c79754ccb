// Shapes, LDS layouts and launch geometry shared by the host API and the
// kernels.  Pure C++ (no HIP types) so the same arithmetic sizes a launch in
// crbm_api.hip and in the CPU emulation harness under tests/emu/.
#pragma once
#include <stdint.h>

namespace crbm {

// Philox counter "kind" field (bits 28..31 of counter word 2); mirrored in
// oracle/crbm_oracle.py.
enum : uint32_t {
  KIND_CHAIN_H = 1, KIND_CHAIN_V = 2, KIND_EVAL_H = 3, KIND_API_H = 4, KIND_API_V = 5
};

// Exact n / d for n*d < 2^31 via one mul_hi (d >= 1).
struct FastDiv {
  uint32_t d, inv;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.inv = (d <= 1) ? 0u : (uint32_t)(0x100000000ull / d) + 1u;
  return f;
}

// What every kernel needs to know about the model.  Passed by value.
struct ModelView {
  const float* W;   // (K,4,M) row-major == reference (K,1,4,M)
  const float* b;   // (K)
  const float* c;   // (4)
  int32_t K, M;
  int32_t G;        // letters per gather-table group
  int32_t ngroups;  // ceil(M / G)
  int32_t rows;     // 4^G
  int32_t ds;       // doublestranded
};

struct RngView {
  uint32_t seed_lo, seed_hi;
  uint32_t step;        // counter word 3
  uint32_t seq_offset;  // global index of local sequence 0
};

// Packed one-hot letters: 16 letters per 32-bit word, two zero pad words so a
// 64+32-bit window read never leaves the row.
inline int letter_words(int L) { return (L + 15) / 16 + 2; }

inline int nq_for(int K) { return (K + 3) / 4; }
inline int mask_words_for_nq(int NQ) { return NQ <= 8 ? 1 : 2; }

// Instantiated specialisations; a model with NQ quads runs on the smallest
// instantiated NQ' >= NQ (pad columns are self-masking, see build_gather_table).
static const int kInstantiatedNQ[] = {1, 2, 3, 4, 5, 6, 8, 10, 13, 16};
inline int instantiated_nq(int NQ) {
  for (int v : kInstantiatedNQ)
    if (v >= NQ) return v;
  return -1;
}

inline int pow4(int g) { return 1 << (2 * g); }
inline int gather_table_floats(int M, int G, int KP) { return ((M + G - 1) / G) * pow4(G) * KP; }

// Largest G whose table(s) fit the LDS budget.
inline int choose_group(int M, int KP, int ds, int budget_bytes) {
  for (int G = 4; G >= 2; --G)
    if ((1 + ds) * gather_table_floats(M, G, KP) * 4 <= budget_bytes) return G;
  return 1;
}

// ---- Gibbs kernel -----------------------------------------------------------
struct GibbsLayout {
  int S;        // chains per tile
  int Lv;       // visible length of a chain = Lf + M - 1
  int Lhp;      // padded hidden row = Lf + 2(M-1)
  int LWs;      // letter words per chain row
  int NW;       // mask words per hidden position
  int tab;      // floats per gather table
  int wt;       // floats of the scatter table  M * NW*32 * 4
  int lds_bytes;
};
inline GibbsLayout gibbs_layout(int K, int M, int ds, int NQ, int G, int Lf, int S) {
  GibbsLayout g;
  g.S = S;
  g.Lv = Lf + M - 1;
  g.Lhp = Lf + 2 * (M - 1);
  g.LWs = letter_words(g.Lv);
  g.NW = mask_words_for_nq(NQ);
  g.tab = gather_table_floats(M, G, 4 * NQ);
  g.wt = M * g.NW * 32 * 4;
  long words = (long)(1 + ds) * g.tab + g.wt + 4 + (long)(1 + ds) * S * g.Lhp * g.NW + (long)S * g.LWs;
  g.lds_bytes = (int)(words * 4);
  (void)K;
  return g;
}

// ---- statistics kernel --------------------------------------------------------
// Accumulator tile owned by one wave: [4 letters][JC filter columns][KC motifs].
inline int stats_nqc(int NQ) { return NQ < 4 ? NQ : 4; }
inline int stats_jc(int NQ) { return stats_nqc(NQ) <= 3 ? 4 : 3; }
struct StatsLayout {
  int ntk, ntj, ntiles;   // k-tiles, j-tiles, total (class x k x j)
  int grid_y;
  int row;                // floats per partial row: 3*KAM + 3K + 4
  int off_vh[2], off_h[2], off_sw, off_sb, off_v;
  int lds_bytes;
};
inline StatsLayout stats_layout(int K, int M, int ds, int NQ, int G, int want_sparsity, int threads) {
  StatsLayout s;
  int KAM = K * 4 * M;
  s.ntk = (NQ + stats_nqc(NQ) - 1) / stats_nqc(NQ);
  s.ntj = (M + stats_jc(NQ) - 1) / stats_jc(NQ);
  s.ntiles = (1 + ds + want_sparsity) * s.ntk * s.ntj;   // classes: vh, vh' (ds), sw
  int waves = threads / 64;
  s.grid_y = (s.ntiles + waves - 1) / waves;
  s.off_vh[0] = 0;
  s.off_vh[1] = KAM;
  s.off_h[0] = 2 * KAM;
  s.off_h[1] = 2 * KAM + K;
  s.off_sw = 2 * KAM + 2 * K;
  s.off_sb = 3 * KAM + 2 * K;
  s.off_v = 3 * KAM + 3 * K;
  s.row = 3 * KAM + 3 * K + 4;
  long words = (long)(1 + ds) * gather_table_floats(M, G, 4 * NQ) + (long)(1 + ds) * threads * 4 * NQ +
               2L * threads + 64;
  s.lds_bytes = (int)(words * 4);
  return s;
}

// Packed sums buffer (what the all-reduce carries), see include/crbm_amd.h.
struct SumsLayout {
  int data_off, n_d, model_off, n_m, count;
  int model_skip_begin, model_skip_len;   // sw,sb are not carried for the model half
};
inline SumsLayout sums_layout(int K, int M) {
  SumsLayout s;
  int KAM = K * 4 * M;
  int row = 3 * KAM + 3 * K + 4;
  s.data_off = 0;
  s.n_d = row;
  s.model_off = row + 1;
  s.model_skip_begin = 2 * KAM + 2 * K;
  s.model_skip_len = KAM + K;
  s.n_m = s.model_off + row - s.model_skip_len;
  s.count = s.n_m + 1;
  return s;
}

}  // namespace crbm
