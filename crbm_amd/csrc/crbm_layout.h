// Shapes, LDS layouts and launch geometry shared by the host API, the kernels
// (compiled per model by hiprtc) and the CPU emulation harness under
// tests/emu/.  Pure C++ (no HIP types).
#pragma once
#include <stdint.h>

namespace crbm {

// Philox counter "kind" field (bits 28..31 of counter word 2); mirrored in
// oracle/crbm_oracle.py.
enum : uint32_t {
  KIND_CHAIN_H = 1, KIND_CHAIN_V = 2, KIND_EVAL_H = 3, KIND_API_H = 4, KIND_API_V = 5
};

// n / d via one mul_hi (d >= 1).  With inv = floor(2^32 / d) + 1 the estimate
// mulhi(n, inv) is exact while n*d < 2^32 and otherwise at most one too large
// (n*inv/2^32 = n/d + n*e/(d*2^32), 0 < e <= d): `fix` asks the device for the
// one-step correction.  The host sets it from the largest dividend the kernel
// will present (items of one tile), so short rows keep the single instruction.
struct FastDiv {
  uint32_t d, inv, fix;
};
inline FastDiv make_fastdiv(uint32_t d, uint64_t max_n = (1ull << 20)) {
  FastDiv f;
  f.d = d;
  f.inv = (d <= 1) ? 0u : (uint32_t)(0x100000000ull / d) + 1u;
  f.fix = (max_n * (uint64_t)d >= 0x100000000ull) ? 1u : 0u;
  return f;
}

struct RngView {
  uint32_t seed_lo, seed_hi;
  uint32_t step;        // counter word 3
  uint32_t seq_offset;  // global index of local sequence 0
};

// Packed one-hot letters: 16 letters per 32-bit word, two zero pad words so a
// window read (three words from the word of its first letter for M <= 32, five for
// M <= 64: crbm_kernels.h, letter_window) never leaves the row.
inline int letter_words(int L) { return (L + 15) / 16 + 2; }
// Alphabets of A != 4 letters (input_dims, convRBM.py:68; generic kernels only): one byte per letter, four per word,
// the same two pad words.
inline int letter_words_any(int A, int L) { return A == 4 ? letter_words(L) : (L + 3) / 4 + 2; }
// Largest motif length the letter windows hold (two 64-bit words); the number of motifs is
// bounded by what the LDS holds (tables + one chain: choose_gibbs_geometry refuses beyond)
// and by the statistics kernel (one role of 64 threads per 16 motifs, at most 1024 threads per block).
constexpr int MAX_MOTIF_LENGTH = 64, MAX_MOTIFS = 256;
// 16-bit pieces of a statistics letter window (32 positions of a group + M - 1 behind them)
constexpr int stats_npw(int M) { return M <= 32 ? 4 : 8; }

constexpr int cpow4(int g) { return 1 << (2 * g); }
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// Everything that is fixed per model.  The kernels are compiled for one Cfg
// (hiprtc at crbm_create, like the reference compiles its Theano graph at
// construction, convRBM.py:175); the host uses the same formulas through
// ModelShape below.
// POOL: hidden units compete in groups of POOL consecutive positions (convRBM.py:245-267,
// "pooling"; 1 = independent sigmoid units, the reference's default and what every
// BASELINE configuration uses).
template <int K_, int M_, int DS_, int G_, int POOL_ = 1>
struct Cfg {
  static constexpr int K = K_, M = M_, DS = DS_, G = G_, POOL = POOL_;
  static constexpr int NQ = cdiv(K, 4), KP = 4 * NQ;      // motifs padded to float4
  static constexpr int NW = cdiv(K, 32);                  // 32-bit mask words per hidden position
  static constexpr int NG = cdiv(M, G), ROWS = cpow4(G);  // gather table: groups x letter tuples
  static constexpr int TAB = NG * ROWS * KP;              // floats per gather table
  static constexpr int NCH = cdiv(K, 5);                  // 5-bit chunks of a position's mask
  // v|h top-down pass, two kernel variants:
  //  dense  (small K*M only): table look-ups per (filter column, 5-bit chunk of a mask)
  //  sparse (every model): a walk over the set bits of the hidden masks
  static constexpr bool DENSE = (NW == 1) && (NCH * M * (1 + DS) <= 64);
  static constexpr int TV = DENSE ? M * NCH * 32 * 4 : 0; // floats per dense top-down table
  // sparse table Ws[jr+4][k] = float4 over letters of W[k][:][M-1-jr], rows jr = -4..-1 and
  // M..M+2 all zero (a thread's 4 positions see jr = q-i, q = window slot, without range checks;
  // the four leading zero rows also serve as the target of the walk's idle look-ups);
  // Wsr is the same for the reverse-complement strand
  static constexpr int WSROWS = M + 7;
  static constexpr int WS = WSROWS * K * 4;               // floats per sparse table
  static constexpr int NGRP = cdiv(K, 10);                // sampler groups of 10 hidden units
  // Precomputed tables buffer (floats), global memory:
  //   [Tf][Tv][Tvr*][c]        TABLES floats = LDS image of the dense Gibbs variant
  //   [Ws][Wsr*][c]            with [Tf] the LDS image of the sparse variant (SP_TABLES floats)
  // (* = doublestranded only.  There is ONE gather table: the reverse-complement strand gathers from it
  //  with the reverse-complemented letter window, crbm_kernels.h revcomp_window)
  static constexpr int OFF_TF = 0;
  static constexpr int OFF_TV = TAB;
  static constexpr int OFF_TVR = OFF_TV + TV;
  static constexpr int OFF_C = OFF_TV + TV * (1 + DS);
  static constexpr int TABLES = OFF_C + 4;                // multiple of 4 floats
  static constexpr int OFF_WS = TABLES;
  static constexpr int OFF_WSR = OFF_WS + WS;
  static constexpr int OFF_C2 = OFF_WS + WS * (1 + DS);
  static constexpr int END2 = OFF_C2 + 4;
  // sparse variant, offsets inside LDS
  static constexpr int SP_WS = TAB;
  static constexpr int SP_WSR = SP_WS + WS;
  static constexpr int SP_C = SP_WS + WS * (1 + DS);
  static constexpr int SP_TABLES = SP_C + 4;
  // MFMA statistics (stats_mfma_body, fused tail of gibbs_body): one v_mfma_f32_16x16x32_f16
  // contracts 32 hidden positions; its 16 output rows are 16 filter columns of one letter, its
  // 16 output columns are 16 motifs of one column kind (P, P' of the rc strand, P(1-P))
  static constexpr int NT = cdiv(K, 16);                  // motif tiles per column kind
  static constexpr int JT = cdiv(M, 16);                  // filter-column tiles per letter
  static constexpr int NPW = stats_npw(M);                // 16-bit pieces per letter window (64 or 128 bits)
  // Letters contracted on the matrix cores.  Every visible position carries exactly one letter, so
  // sum_a VH[k,a,j] = sum_s P[k,s] = H[k] for every filter column j: with a spare row in the last
  // filter-column tile (M not a multiple of 16) that row is fed all ones and yields H, and the fourth
  // letter follows as H - (the other three) -- a quarter fewer MFMAs and A fragments.
  static constexpr int NL = (M % 16 != 0) ? 3 : 4;
  static_assert(M <= MAX_MOTIF_LENGTH && K <= MAX_MOTIFS, "model beyond the kernels' limits");
  // the model half of the statistics rides in the Gibbs kernel's last h|v pass when all motifs of
  // a position fit one wave's accumulator set of at most 8 tiles (32 registers; measured: with 16 tiles,
  // config #5, the fused kernel drops to one wave per SIMD and loses to the separate launch)
  static constexpr bool FUSE_STATS = 4 * JT * (1 + DS) * NT <= 8 && POOL == 1;
  static constexpr int TABLES_ALL = END2;
  // hit-summary kernel: a lane keeps the position sums of HIT_NI positions in registers
  static constexpr int HIT_NI = (48 / KP) < 1 ? 1 : ((48 / KP) > 4 ? 4 : (48 / KP));
};

// Host-side mirror of Cfg (runtime values, same arithmetic).
struct ModelShape {
  int K, M, DS, G, NT, JT, POOL, NPW;
  int NQ, KP, NW, NG, ROWS, TAB, NCH, DENSE, TV, WS, NGRP;
  int OFF_TF, OFF_TV, OFF_TVR, OFF_C, TABLES, OFF_WS, END2, SP_TABLES, TABLES_ALL;
  int HIT_NI, FUSE_STATS;
};
inline ModelShape model_shape(int K, int M, int DS, int G, int POOL = 1) {
  ModelShape s;
  s.K = K; s.M = M; s.DS = DS; s.G = G; s.POOL = POOL;
  s.NQ = cdiv(K, 4); s.KP = 4 * s.NQ; s.NW = cdiv(K, 32);
  s.NT = cdiv(K, 16); s.JT = cdiv(M, 16); s.NPW = stats_npw(M);
  s.FUSE_STATS = 4 * s.JT * (1 + DS) * s.NT <= 8 && POOL == 1;
  s.NG = cdiv(M, G); s.ROWS = cpow4(G); s.TAB = s.NG * s.ROWS * s.KP;
  s.NCH = cdiv(K, 5);
  s.DENSE = (s.NW == 1) && (s.NCH * M * (1 + DS) <= 64);
  s.TV = s.DENSE ? M * s.NCH * 32 * 4 : 0;
  s.WS = (M + 7) * K * 4;
  s.NGRP = cdiv(K, 10);
  s.OFF_TF = 0; s.OFF_TV = s.TAB; s.OFF_TVR = s.OFF_TV + s.TV;
  s.OFF_C = s.OFF_TV + s.TV * (1 + DS); s.TABLES = s.OFF_C + 4;
  s.OFF_WS = s.TABLES; s.END2 = s.OFF_WS + s.WS * (1 + DS) + 4;
  s.SP_TABLES = s.TAB + s.WS * (1 + DS) + 4;
  s.TABLES_ALL = s.END2;
  s.HIT_NI = (48 / s.KP) < 1 ? 1 : ((48 / s.KP) > 4 ? 4 : (48 / s.KP));
  return s;
}

// Letters per gather-table group: the largest G whose table stays small
// enough (budget) to leave the CU several resident blocks; failing that G = 2
// if it fits twice the budget, else G = 1.  (One table serves both strands.)
inline int choose_group(int K, int M, int /*ds*/, int budget_bytes) {
  const int KP = 4 * cdiv(K, 4);
  auto bytes = [&](int G) { return cdiv(M, G) * cpow4(G) * KP * 4; };
  for (int G = 4; G >= 2; --G)
    if (bytes(G) <= budget_bytes) return G;
  return bytes(2) <= 2 * budget_bytes ? 2 : 1;
}

// ---- Gibbs kernel -------------------------------------------------------------
// v|h: a thread owns 4 consecutive visible positions (nvb = ceil(Lv/4) items per
// chain); h|v: one hidden position per item (nhb = Lf).  Hidden position s sits at index
// s + M-1 of a zero-padded mask row of Lrow positions (multiple of 4).
// Behind the letter rows sits the block's queue of undecided hidden units (crbm_kernels.h, gibbs_body):
// word 0 = count, words 2.. = entries (item | unit << 20 | strand << 26).
#ifndef CRBM_FIXQ_CAP
#define CRBM_FIXQ_CAP 254    // tests shrink it (CRBM_JIT_DEFINES) to drive the overflow path; never above FIXQ_WORDS - 2
#endif
constexpr int FIXQ_WORDS = 256, FIXQ_CAP = CRBM_FIXQ_CAP;
static_assert(FIXQ_CAP >= 0 && FIXQ_CAP <= FIXQ_WORDS - 2, "queue capacity");
struct GibbsLayout {
  int S, Lv, nvb, nhb, Lrow, LWs;
  int lds_bytes;
};
inline GibbsLayout gibbs_layout(const ModelShape& ms, int Lf, int S, bool sparse) {
  GibbsLayout g;
  g.S = S;
  g.Lv = Lf + ms.M - 1;
  g.nvb = cdiv(g.Lv, 4);
  g.nhb = Lf;
  g.Lrow = 4 * cdiv(4 * g.nvb + ms.M - 1 + 3, 4);
  g.LWs = letter_words(4 * g.nvb);
  const long words = (long)(sparse ? ms.SP_TABLES : ms.TABLES) + (long)(1 + ms.DS) * S * g.Lrow * ms.NW + (long)S * g.LWs + FIXQ_WORDS;
  g.lds_bytes = (int)(words * 4);
  return g;
}

// ---- MFMA statistics ----------------------------------------------------------------
// Hidden positions are handled in groups of 32 (one MFMA step).  A chain of Lh hidden
// positions is GPC = ceil(Lh / 32) groups, the tail of the last one padded with P = 0.
// A wave works on units of two consecutive groups (64 positions, one per lane; a unit may
// span two chains) and is autonomous: it computes P of its 64 positions, parks them
// transposed in its own LDS slice and runs the MFMA steps on them -- no block barrier in
// the loop.  Column kinds: P (forward strand), P' (rc strand, ds), Q = P(1-P) (forward,
// data half).  A wave owns the 16-motif tiles [nt0, nt0 + NTW) of every kind (its role,
// NR = NT / NTW roles), so that its accumulator set stays <= 32 tiles (128 registers);
// a unit is processed by one wave of every role.
//   LDS of a block:  [lut][slice of wave 0][slice of wave 1] ...   (+ gather tables)
//   slice:           win [2][4] letter windows of NPW 16-bit pieces (64 bits for M <= 32, 128 beyond):
//                        bit t of win[g][a] = [letter(32*gi + t) == a]
//                    gw  [2][NPW] packed letter words of the two groups (stats_mfma_body only)
//                    Pt  [kinds*KW + 1 rows][68] floats: row kind*KW + i = motif 16*nt0 + i of the
//                        kind, 64 positions + 4 pad (row stride 17 x 16 B: conflict-free b128 reads);
//                        the last row is all zero (motifs beyond K)
//   lut:             byte -> eight f16 (bit e set -> 2^-14, else 0): one 16-byte read makes an A fragment
//                    (stand-alone kernel, 4 KB); the Gibbs kernel, short of LDS, uses the nibble form
//                    (16 entries of four f16, two reads per fragment, 128 bytes)
constexpr int STATS_RS = 68;                 // Pt row stride in floats
#ifndef CRBM_STATS_MAX_TILES
#define CRBM_STATS_MAX_TILES 32     // accumulator tiles (4 registers each) a wave may hold
#endif
constexpr int stats_ntw(int NT, int JT, int kinds, int max_tiles = CRBM_STATS_MAX_TILES) {   // motif tiles per wave
  int ntw = NT;
  while (ntw > 1 && (4 * JT * kinds * ntw > max_tiles || NT % ntw != 0)) --ntw;
  return ntw;
}
// block size: 8 waves, rounded down to whole sets of roles
constexpr int stats_mfma_threads(int NR) { return 64 * NR * (8 / NR > 0 ? 8 / NR : 1); }
struct StatsMfmaLayout {
  int kinds, NTW, NR, KW, rows;   // column kinds, motif tiles per wave, roles, Pt rows per kind, Pt rows
  int threads, GPC;
  int off_win, off_gw, off_pt, slice;   // floats inside a slice / per slice
  int off_slices;                       // first slice (after the LUT), floats
  int region_floats;                    // LUT + all slices
  int combine_bytes;                    // end-of-kernel combine buffer, from the LDS base (everything is dead by then)
  int row, off_vh[2], off_h[2], off_sw, off_sb, off_v;   // partial row: [vh KAM][vh' KAM][h K][h' K][sw KAM][sb K][v 4]
};
// threads = 0: as many waves per role as the LDS (160 KiB minus `other_lds_bytes`: gather tables, or the
// chain image of the Gibbs kernel) holds, at most 8 waves in all
inline StatsMfmaLayout stats_mfma_layout(const ModelShape& ms, int want_sparsity, int Lh, int threads = 0,
                                         int other_lds_bytes = 0, bool byte_lut = true, int max_tiles = CRBM_STATS_MAX_TILES) {
  StatsMfmaLayout s;
  const int K = ms.K, M = ms.M, KAM = K * 4 * M;
  s.kinds = 1 + ms.DS + (want_sparsity ? 1 : 0);
  s.NTW = stats_ntw(ms.NT, ms.JT, s.kinds, max_tiles);
  s.NR = ms.NT / s.NTW;
  s.KW = 16 * s.NTW < K ? 16 * s.NTW : K;
  // the sparsity columns are derived from the P fragments in registers, not parked (pooled units excepted): crbm_kernels.h, StatsRole
  const int parked = s.kinds - ((want_sparsity && ms.POOL == 1) ? 1 : 0);
  s.rows = parked * s.KW + 1;
  s.GPC = cdiv(Lh, 32);
  s.off_win = 0;                        // 8 windows x NPW/2 words
  s.off_gw = 4 * ms.NPW;                // 2 x NPW words
  s.off_pt = 6 * ms.NPW;
  s.slice = (s.off_pt + s.rows * STATS_RS + 3) & ~3;
  s.off_slices = byte_lut ? 1024 : 32;      // 256 x 16 B (stand-alone kernel) or 16 x 8 B (fused tail of the Gibbs kernel)
  s.threads = threads > 0 ? threads : stats_mfma_threads(s.NR);
  if (threads <= 0)
    while (s.threads > 64 * s.NR && (s.off_slices + (s.threads / 64) * s.slice) * 4 + other_lds_bytes > 160 * 1024)
      s.threads -= 64 * s.NR;
  s.region_floats = s.off_slices + (s.threads / 64) * s.slice;
  // per kind: total [KAM] + exchange + one set of roles parked ([KW][4][M] per wave); more waves at a time if there is room
  s.combine_bytes = (KAM + 64 + s.NR * s.KW * 4 * M) * 4;
  s.off_vh[0] = 0;
  s.off_vh[1] = KAM;
  s.off_h[0] = 2 * KAM;
  s.off_h[1] = 2 * KAM + K;
  s.off_sw = 2 * KAM + 2 * K;
  s.off_sb = 3 * KAM + 2 * K;
  s.off_v = 3 * KAM + 3 * K;
  s.row = 3 * KAM + 3 * K + 4;
  return s;
}

// Packed sums buffer (what the all-reduce carries), see include/crbm_amd.h.
struct SumsLayout {
  int data_off, n_d, model_off, n_m, count;
  int model_skip_begin, model_skip_len;   // sw,sb are not carried for the model half
};
inline SumsLayout sums_layout(int K, int M, int A = 4) {
  SumsLayout s;
  const int KAM = K * A * M;
  const int row = 3 * KAM + 3 * K + A;
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
