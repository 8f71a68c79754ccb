// The generic kernels of crbm_kernels.h (run-time K, M and alphabet size): a fragment of that header, included by it
// inside namespace crbm and behind CRBM_DEFINE_MISC_KERNELS (the ahead-of-time build of crbm_api.hip and the CPU emulator;
// never part of the per-model hiprtc compilation, so editing this file does not invalidate the cache of specialised kernels).
#pragma once
// ===========================================================================
// The "big" path: models the LDS-resident kernels above do not take -- more than 256 motifs, motifs longer than 64
// letters, or tables that exceed the LDS (120 x 40 double-stranded, 300 x 10, 8 x 100 ...).  The reference accepts any
// positive num_motifs and motif_length (convRBM.py:72-108, :127-133); these kernels do too.  Plain kernels with
// run-time K and M, compiled ahead of time: the filters are staged through LDS a slab of motifs (or one motif) at a
// time, hidden masks are walked word by word, everything else follows the same formulas, the same bit-packed state
// (K-bit masks, 2-bit letters) and the same Philox counters as the specialised kernels, so a model may cross the
// boundary between the two paths without its samples changing beyond p == u ties (pooled models included).
// ===========================================================================
struct BigModel {
  const float* W;   // (K,A,M)
  const float* b;   // (K)
  const float* c;   // (A)
  int32_t K, M, ds, NW;
  int32_t A;        // letters of the alphabet (input_dims, convRBM.py:68): 4 = DNA, rows of 2-bit letters; anything else: rows of bytes
};

// letter p of a packed row: 16 two-bit letters per word (A == 4), else one byte per letter
__device__ __forceinline__ uint32_t letter_at(const BigModel& m, const uint32_t* row, int p) {
  if (m.A == 4) return (row[p >> 4] >> (2 * (p & 15))) & 3u;
  return reinterpret_cast<const unsigned char*>(row)[p];
}
__device__ __forceinline__ float sigmoid_x(float x) { return fast_rcp(1.0f + __expf(-x)); }   // v_rcp_f32 (1 ulp), as sigmoid_z

// block-wide sum in a fixed order (waves through DPP, then wave totals through LDS); all threads get the total
__device__ __forceinline__ float big_block_sum(float v, float* xch) {
  const float w = wave_sum(v);
  __syncthreads();                                   // xch may still be read from the previous call
  if ((threadIdx.x & 63) == 0) xch[threadIdx.x >> 6] = w;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += xch[i];
  return t;
}
__device__ __forceinline__ float big_block_max(float v, float* xch) {
  const float w = wave_max_nonneg(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) xch[threadIdx.x >> 6] = w;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, xch[i]);
  return t;
}

// ---- h | v: _bottomUpActivity / _bottomUpProbability / _bottomUpSample (convRBM.py:238-275), motifHitProbs (:507-514),
// and the h|v half of a Gibbs step (masks out).  A thread owns one hidden position of a tile of rows and takes the
// motifs KS (<= 32, a divisor of 32) at a time: the slab's filters sit in LDS as Ws[k][j][letter].
struct BigHgvArgs {
  BigModel m;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  int32_t TS, KS;
  int32_t split;               // 1: blockIdx.y = mask word, the block takes the slabs of that word only (few tiles: more blocks)
  int32_t pool;                // pooling (1: independent units; > 1: big_hgv_pooled_kernel, KS <= 8)
  int32_t mode;                // 0: forward strand, 1: reverse-complement strand, 2: sigma(x + x')
  float* act;                  // (n,K,1,Lh) each, may be null
  float* prob;
  float* sample;
  unsigned long long* ones;    // += sampled ones, may be null
  uint32_t* masks;             // [n][Lh][NW] sampled units of the strand (chain state), may be null
  RngView rng;
  uint32_t kind;
};

// floats per (column, letter) row of a slab of KS motifs in LDS
__host__ __device__ inline int big_hgv_ksp(int KS) { return ((KS + 3) & ~3) + 4; }
// KSM: slab capacity of the register arrays (32 unpooled; 8 pooled, which keeps five arrays per unit).
// POOLED: the units of `pool` consecutive positions compete (convRBM.py:245-267): P_i = exp(x_i) / (pool + sum_j exp(x_j)),
// one draw per group -- the uniform of its first position -- against the cumulative probabilities; a thread evaluates its
// whole group itself (two passes over the group's activations: maximum, then sums), like the specialised kernels do.
template <int KSM, bool POOLED>
__device__ __forceinline__ void big_hgv_body(const BigHgvArgs& a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const int K = a.m.K, M = a.m.M, KS = a.KS;
  const int A = a.m.A;
  // the slab's filters as Ws[column][letter][motif], motifs contiguous (four per LDS read); rows padded by four floats so
  // that the rows of different letters start in different banks
  const int KSP = big_hgv_ksp(KS);
  float* Ws = smem;                                            // [M][A][KSP]
  float* bs = Ws + (size_t)KSP * M * A;                        // [32]
  uint32_t* let = reinterpret_cast<uint32_t*>(bs + 32);        // [TS][LW]
  const bool want_sample = a.sample || a.ones || a.masks;
  const uint32_t strand = a.mode == 1 ? 1u : 0u;
  unsigned long long cnt = 0;
  const int ntiles = (a.n + a.TS - 1) / a.TS;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * a.TS, ns = min(a.TS, a.n - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < ns * a.LW; i += blockDim.x) let[i] = a.letters[(size_t)n0 * a.LW + i];
    const int items = ns * a.Lh;
    const int kbeg = a.split ? 32 * (int)blockIdx.y : 0, kend = a.split ? min(K, kbeg + 32) : K;
    for (int k0 = kbeg; k0 < kend; k0 += KS) {
      const int kc = min(KS, K - k0);
      __syncthreads();
      for (int i = threadIdx.x; i < KSP * M * A; i += blockDim.x) {
        const int k = i % KSP, al = (i / KSP) % A, j = i / (KSP * A);
        Ws[i] = k < kc ? a.m.W[((size_t)(k0 + k) * A + al) * M + j] : 0.f;
      }
      if ((int)threadIdx.x < kc) bs[threadIdx.x] = a.m.b[k0 + threadIdx.x];
      __syncthreads();
      for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int nl = it / a.Lh, s = it - nl * a.Lh, nn = n0 + nl;
        const uint32_t* lrow = let + (size_t)nl * a.LW;
        // x[k] = b[k] + sum_j W[k, l(pos+j), j]; rc strand: W[k, A-1 - l(pos+j), M-1-j] (convRBM.py:241); mode 2: both, each with b
        auto activations = [&](int pos, float (&x)[KSM]) {
#pragma unroll
          for (int k = 0; k < KSM; ++k) x[k] = 0.f;
          for (int pass = 0; pass < (a.mode == 2 ? 2 : 1); ++pass) {
            const bool rc = a.mode == 1 || pass == 1;
            for (int j = 0; j < M; ++j) {
              const uint32_t l = letter_at(a.m, lrow, pos + j);
              const float4* col = reinterpret_cast<const float4*>(Ws + (size_t)((rc ? M - 1 - j : j) * A + (int)(rc ? (uint32_t)(A - 1) - l : l)) * KSP);
#pragma unroll
              for (int q = 0; q < KSM / 4; ++q)
                if (4 * q < kc) {            // wave-uniform; motifs beyond kc are zero in the slab
                  const float4 t = col[q];
                  x[4 * q] += t.x; x[4 * q + 1] += t.y; x[4 * q + 2] += t.z; x[4 * q + 3] += t.w;
                }
            }
#pragma unroll
            for (int k = 0; k < KSM; ++k)
              if (k < kc) x[k] += bs[k];
          }
        };
        float x[KSM], p[KSM], cb[KSM];
        int su = s;                                  // position whose uniform decides (the group's first when pooled)
        if constexpr (POOLED) {
          const int g0 = s - s % a.pool, me = s - g0;
          su = g0;
          float mx[KSM], den[KSM], xt[KSM];
#pragma unroll
          for (int k = 0; k < KSM; ++k) mx[k] = 0.f;                 // the "pool" term is pool * exp(0)
          for (int j = 0; j < a.pool; ++j) {
            activations(g0 + j, xt);
#pragma unroll
            for (int k = 0; k < KSM; ++k) mx[k] = fmaxf(mx[k], xt[k]);
          }
#pragma unroll
          for (int k = 0; k < KSM; ++k) { den[k] = (float)a.pool * __expf(-mx[k]); cb[k] = 0.f; p[k] = 0.f; x[k] = 0.f; }
          for (int j = 0; j < a.pool; ++j) {
            activations(g0 + j, xt);
#pragma unroll
            for (int k = 0; k < KSM; ++k) {
              const float e = __expf(xt[k] - mx[k]);
              den[k] += e;
              cb[k] += j < me ? e : 0.f;
              p[k] = j == me ? e : p[k];
              x[k] = j == me ? xt[k] : x[k];
            }
          }
#pragma unroll
          for (int k = 0; k < KSM; ++k) {
            const float inv = 1.0f / den[k];
            p[k] *= inv;
            cb[k] *= inv;
          }
        } else {
          activations(s, x);
#pragma unroll
          for (int k = 0; k < KSM; ++k) { p[k] = sigmoid_x(x[k]); cb[k] = 0.f; }
        }
        uint32_t bits = 0u;
        uint32_t last_g = 0xFFFFFFFFu;
        bool have_fine = false;
        Philox4 rcs = {}, rfs = {};
        const uint32_t gn = a.rng.seq_offset + (uint32_t)nn;
#pragma unroll
        for (int k = 0; k < KSM; ++k) {
          if (k < kc) {
            const int gk = k0 + k;
            const size_t idx = ((size_t)nn * K + gk) * a.Lh + s;
            if (a.act) a.act[idx] = x[k];
            if (a.prob) a.prob[idx] = p[k];
            if (want_sample) {
              const uint32_t g = (uint32_t)gk / 10u, i10 = (uint32_t)gk - 10u * g;
              if (g != last_g) {
                rcs = philox4x32(gn, (uint32_t)su, rng_word2(a.kind, strand, 0, g), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
                last_g = g;
                have_fine = false;
              }
              // u = (coarse * 4096 + fine) / 2^24 (crbm_kernels.h, sample_hidden).  The coarse field alone decides unless a
              // threshold falls inside its bucket of 4096 (2^-12 of the unpooled units): the fine call is made only then.
              // Exact: p * 2^24 and the integers below 2^24 are floats, the comparisons are those of p > u itself.
              const uint32_t coarse = philox_field12_dyn(rcs, (int)i10);
              const float lo24 = (float)(coarse * 4096u), hi24 = (float)(coarse * 4096u + 4096u);
              const float t0 = (POOLED ? cb[k] : p[k]) * 16777216.0f, t1 = POOLED ? (cb[k] + p[k]) * 16777216.0f : t0;
              const bool inside0 = t0 > lo24 && t0 < hi24, inside1 = POOLED && t1 > lo24 && t1 < hi24;
              uint32_t fine = 0u;
              if (inside0 || inside1) {
                if (!have_fine) {
                  rfs = philox4x32(gn, (uint32_t)su, rng_word2(a.kind, strand, 1, g), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
                  have_fine = true;
                }
                fine = philox_field12_dyn(rfs, (int)i10);
              }
              const float u = (float)(coarse * 4096u + fine) * 5.9604644775390625e-8f;
              // unpooled: h = 1 if p > u; pooled: the first unit of the group whose cumulative probability exceeds u
              const uint32_t hb = POOLED ? ((cb[k] + p[k] > u && cb[k] <= u) ? 1u : 0u) : (p[k] > u ? 1u : 0u);
              if (a.sample) a.sample[idx] = (float)hb;
              bits |= hb << (gk & 31);
              cnt += hb;
            }
          }
        }
        if (a.masks) {      // the slab lies inside one mask word; the same thread meets this item for every slab
          uint32_t* wp = a.masks + ((size_t)nn * a.Lh + s) * a.m.NW + (k0 >> 5);
          *wp = (k0 & 31) ? (*wp | bits) : bits;
        }
      }
    }
  }
  if (a.ones && cnt) atomicAdd(a.ones, cnt);
}

__global__ void __launch_bounds__(256) big_hgv_kernel(BigHgvArgs a) { big_hgv_body<32, false>(a); }
__global__ void __launch_bounds__(256) big_hgv_pooled_kernel(BigHgvArgs a) { big_hgv_body<8, true>(a); }

// ---- v | h of the chain from the masks (convRBM.py:277-325): y[a,p] = c[a] + sum_k sum_j W[k,a,j] h[k,p-j] (+ rc strand),
// softmax over the four letters, one categorical draw per position.  One block per chain; a thread owns up to BIG_VR
// positions and keeps their activations in registers while the filters pass through LDS 32 motifs x JS columns at a time.
constexpr int BIG_VR = 8;
struct BigVghArgs {
  BigModel m;
  const uint32_t* hm;       // [nchains][Lf][NW]
  const uint32_t* hmp;      // reverse-complement strand or null
  uint32_t* vout;           // [nchains][LWs] packed letters of the sample
  int32_t nchains, Lf, Lv, LWs;
  int32_t JS;               // filter columns per staged slab
  RngView rng;
};

__global__ void __launch_bounds__(256) big_vgh_kernel(BigVghArgs a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const int K = a.m.K, M = a.m.M, NW = a.m.NW, JS = a.JS;
  float4* Wt = reinterpret_cast<float4*>(smem);                       // [JS][32]: W[k, 0..3, j] of the slab
  unsigned char* lb = reinterpret_cast<unsigned char*>(Wt + (size_t)JS * 32);   // [BIG_VR * blockDim] letters of a chunk
  const int CH = BIG_VR * (int)blockDim.x;
  // mask word w of the hidden positions a chunk's visible positions meet (s = p - j: CH + M - 1 of them), both strands: read
  // from global memory once per (chunk, word) instead of once per (position, filter column, word)
  const int HSW = min(CH, a.Lv) + M - 1;              // (a chain shorter than a chunk needs no more than its own length)
  uint32_t* hs = reinterpret_cast<uint32_t*>(lb + (size_t)CH);         // [2][HSW]
  for (int chain = blockIdx.x; chain < a.nchains; chain += gridDim.x) {
    const uint32_t gn = a.rng.seq_offset + (uint32_t)chain;
    for (int c0 = 0; c0 < a.Lv; c0 += CH) {
      float y[BIG_VR][4];
#pragma unroll
      for (int r = 0; r < BIG_VR; ++r) { y[r][0] = a.m.c[0]; y[r][1] = a.m.c[1]; y[r][2] = a.m.c[2]; y[r][3] = a.m.c[3]; }
      const int sbase = max(0, c0 - (M - 1)), send = min(a.Lf, c0 + CH);      // hidden positions [sbase, send) matter to this chunk
      for (int w = 0; w < NW; ++w)
        for (int j0 = 0; j0 < M; j0 += JS) {
          const int jc = min(JS, M - j0), kc = min(32, K - 32 * w);
          __syncthreads();
          if (j0 == 0)
            for (int i = threadIdx.x; i < send - sbase; i += blockDim.x) {
              hs[i] = a.hm[((size_t)chain * a.Lf + sbase + i) * NW + w];
              if (a.hmp) hs[HSW + i] = a.hmp[((size_t)chain * a.Lf + sbase + i) * NW + w];
            }
          for (int i = threadIdx.x; i < jc * 32; i += blockDim.x) {
            const int j = i >> 5, k = i & 31;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kc) {
              const float* wk = a.m.W + (size_t)(32 * w + k) * 4 * M + (j0 + j);
              t = make_float4(wk[0], wk[M], wk[2 * M], wk[3 * M]);
            }
            Wt[i] = t;
          }
          __syncthreads();
#pragma unroll
          for (int r = 0; r < BIG_VR; ++r) {
            const int p = c0 + (int)threadIdx.x + r * (int)blockDim.x;
            if (p < a.Lv)
              for (int strand = 0; strand <= (a.hmp ? 1 : 0); ++strand) {
                const uint32_t* hrow = hs + (strand ? HSW : 0) - sbase;
                // forward strand: column j of the filter meets hidden position p - j; rc strand: rc(W)[k,a,j] = W[k,3-a,M-1-j]
                for (int j = 0; j < jc; ++j) {
                  const int jj = j0 + j;                       // column of W that is staged at row j
                  const int s = strand ? p - (M - 1 - jj) : p - jj;
                  if (s < 0 || s >= a.Lf) continue;
                  uint32_t bits = hrow[s];
                  while (bits) {
                    const int k = __ffs(bits) - 1;
                    bits &= bits - 1u;
                    const float4 t = Wt[j * 32 + k];
                    if (strand) { y[r][0] += t.w; y[r][1] += t.z; y[r][2] += t.y; y[r][3] += t.x; }
                    else { y[r][0] += t.x; y[r][1] += t.y; y[r][2] += t.z; y[r][3] += t.w; }
                  }
                }
              }
          }
        }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BIG_VR; ++r) {
        const int pl = (int)threadIdx.x + r * (int)blockDim.x, p = c0 + pl;
        if (p < a.Lv) {
          const Philox4 rr = philox4x32(gn, (uint32_t)(p >> 2), rng_word2(KIND_CHAIN_V, 0, 0, 0), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
          const float mx = fmaxf(fmaxf(y[r][0], y[r][1]), fmaxf(y[r][2], y[r][3]));
          const float e0 = __expf(y[r][0] - mx), e1 = __expf(y[r][1] - mx), e2 = __expf(y[r][2] - mx), e3 = __expf(y[r][3] - mx);
          const float t = u01(philox_pick(rr, p & 3)) * ((e0 + e1) + (e2 + e3));
          lb[pl] = (unsigned char)((t >= e0) + (t >= e0 + e1) + (t >= (e0 + e1) + e2));
        }
      }
      __syncthreads();
      // 16 letters per word (CH is a multiple of 16: the chunks start on word boundaries); the row's pad words stay zero
      const int nw = (min(CH, a.Lv - c0) + 15) / 16;
      for (int wi = threadIdx.x; wi < nw; wi += blockDim.x) {
        uint32_t word = 0u;
        for (int t = 0; t < 16; ++t)
          if (c0 + 16 * wi + t < a.Lv) word |= (uint32_t)lb[16 * wi + t] << (2 * t);
        a.vout[(size_t)chain * a.LWs + (c0 >> 4) + wi] = word;
      }
    }
    for (int wi = (a.Lv + 15) / 16 + (int)threadIdx.x; wi < a.LWs; wi += blockDim.x) a.vout[(size_t)chain * a.LWs + wi] = 0u;
  }
}

// The same for an alphabet of A != 4 letters (input_dims, convRBM.py:68-71, :277-325): a thread owns ONE position of a
// chunk of blockDim positions and keeps its A activations in LDS (yl[letter][thread]); the sample is written one byte
// per letter.  The categorical draw is the reference's: the first letter whose cumulative probability exceeds u.
__global__ void __launch_bounds__(256) big_vgh_any_kernel(BigVghArgs a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const int K = a.m.K, M = a.m.M, NW = a.m.NW, JS = a.JS, A = a.m.A;
  const int CH = (int)blockDim.x, tid = (int)threadIdx.x;
  float* Wt = smem;                                                   // [JS][32][A]: W[k, 0..A-1, j] of the slab
  float* yl = Wt + (size_t)JS * 32 * A;                               // [A][CH]
  unsigned char* lb = reinterpret_cast<unsigned char*>(yl + (size_t)A * CH);   // [CH] letters of a chunk
  for (int chain = blockIdx.x; chain < a.nchains; chain += gridDim.x) {
    const uint32_t gn = a.rng.seq_offset + (uint32_t)chain;
    for (int c0 = 0; c0 < a.Lv; c0 += CH) {
      const int p = c0 + tid;
      for (int al = 0; al < A; ++al) yl[al * CH + tid] = a.m.c[al];
      for (int w = 0; w < NW; ++w)
        for (int j0 = 0; j0 < M; j0 += JS) {
          const int jc = min(JS, M - j0), kc = min(32, K - 32 * w);
          __syncthreads();
          for (int i = tid; i < jc * 32 * A; i += CH) {
            const int al = i % A, k = (i / A) & 31, j = i / (32 * A);
            Wt[i] = k < kc ? a.m.W[((size_t)(32 * w + k) * A + al) * M + (j0 + j)] : 0.f;
          }
          __syncthreads();
          if (p < a.Lv)
            for (int strand = 0; strand <= (a.hmp ? 1 : 0); ++strand) {
              const uint32_t* hrow = (strand ? a.hmp : a.hm) + (size_t)chain * a.Lf * NW + w;
              for (int j = 0; j < jc; ++j) {
                const int jj = j0 + j;
                const int s = strand ? p - (M - 1 - jj) : p - jj;       // rc(W)[k,a,j] = W[k,A-1-a,M-1-j]
                if (s < 0 || s >= a.Lf) continue;
                uint32_t bits = hrow[(size_t)s * NW];
                while (bits) {
                  const int k = __ffs(bits) - 1;
                  bits &= bits - 1u;
                  const float* t = Wt + (size_t)(j * 32 + k) * A;
                  for (int al = 0; al < A; ++al) yl[al * CH + tid] += t[strand ? A - 1 - al : al];
                }
              }
            }
        }
      if (p < a.Lv) {
        float mx = yl[tid];
        for (int al = 1; al < A; ++al) mx = fmaxf(mx, yl[al * CH + tid]);
        float tot = 0.f;
        for (int al = 0; al < A; ++al) {
          const float e = __expf(yl[al * CH + tid] - mx);
          yl[al * CH + tid] = e;
          tot += e;
        }
        const Philox4 rr = philox4x32(gn, (uint32_t)(p >> 2), rng_word2(KIND_CHAIN_V, 0, 0, 0), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
        const float t = u01(philox_pick(rr, p & 3)) * tot;
        float cum = 0.f;
        int l = 0;
        for (int al = 0; al < A - 1; ++al) {
          cum += yl[al * CH + tid];
          l += t >= cum ? 1 : 0;
        }
        lb[tid] = (unsigned char)l;
      }
      __syncthreads();
      // four letters per word (CH is a multiple of 4: the chunks start on word boundaries); the row's pad words stay zero
      const int nw = (min(CH, a.Lv - c0) + 3) / 4;
      for (int wi = tid; wi < nw; wi += CH) {
        uint32_t word = 0u;
        for (int t = 0; t < 4; ++t)
          if (c0 + 4 * wi + t < a.Lv) word |= (uint32_t)lb[4 * wi + t] << (8 * t);
        a.vout[(size_t)chain * a.LWs + (c0 >> 2) + wi] = word;
      }
      __syncthreads();
    }
    for (int wi = (a.Lv + 3) / 4 + tid; wi < a.LWs; wi += CH) a.vout[(size_t)chain * a.LWs + wi] = 0u;
  }
}

// ---- gradient statistics (convRBM.py:327-371) and the sparsity sums (:440-451), raw sums into partial rows of the layout
// the column reduction expects.  Block (k, r): motif k, rows r, r + R, ...; the motif's filter, a chunk of letters and
// the probabilities of the chunk's positions sit in LDS; thread t accumulates VH[k, a, j] for (a, j) = t / M, t % M.
constexpr int BIG_ST = 8;                                   // (a, j) slots per thread: A M <= BIG_ST * blockDim
struct BigStatsArgs {
  BigModel m;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  int32_t want_sparsity, R, CH;
  int32_t pool;                                             // pooling (CH is a multiple of it)
  float* partials;                                          // [R][row]
  int32_t row, off_vh0, off_vh1, off_h0, off_h1, off_sw, off_sb, off_v;
};

__global__ void __launch_bounds__(256) big_stats_kernel(BigStatsArgs a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const int M = a.m.M, k = blockIdx.x, r = blockIdx.y, A = a.m.A, AM = A * M, CH = a.CH;
  float* Wk = smem;                                         // [A][M]
  float* P = Wk + ((AM + 3) & ~3);                          // [CH] each
  float* Pp = P + CH;
  float* Q = Pp + CH;
  float* X = Q + CH;                                        // [CH] each: activations of the chunk (pooled models only)
  float* Xp = X + (a.pool > 1 ? CH : 0);
  float* xch = Xp + (a.pool > 1 ? CH : 0);                  // [16]
  float* red = xch + 16;                                    // [12][blockDim]: the slices' sums meet here (3 kinds x 4 letters, or 3 kinds)
  uint32_t* cnt = reinterpret_cast<uint32_t*>(red + 12 * blockDim.x);  // [A] letter counts (blocks of motif 0)
  unsigned char* lb = reinterpret_cast<unsigned char*>(cnt + ((A + 3) & ~3));   // [CH + M]
  for (int i = threadIdx.x; i < AM; i += blockDim.x) Wk[i] = a.m.W[(size_t)k * AM + i];
  for (int i = threadIdx.x; i < A; i += blockDim.x) cnt[i] = 0u;
  const float bk = a.m.b[k];
  float vh[BIG_ST], vhp[BIG_ST], sw[BIG_ST];
#pragma unroll
  for (int t = 0; t < BIG_ST; ++t) vh[t] = vhp[t] = sw[t] = 0.f;
  const int nsl = AM <= (int)blockDim.x ? (int)blockDim.x / AM : 0;    // position slices per (letter, column); 0: A M > blockDim
  // DNA with at most blockDim filter columns: a thread owns a COLUMN and a slice of positions and keeps one accumulator per
  // letter and kind -- one letter read and three probability reads per (column, position) instead of per (letter, column,
  // position): the loop is bound by its LDS reads (300 x 10 on 4096 chains: 6.05 -> 2.2 ms per half)
  const bool by_column = A == 4 && M <= (int)blockDim.x;
  const int ncs = by_column ? (int)blockDim.x / M : 0;                  // position slices per column
  float c_vh[4] = {0.f, 0.f, 0.f, 0.f}, c_vhp[4] = {0.f, 0.f, 0.f, 0.f}, c_sw[4] = {0.f, 0.f, 0.f, 0.f};
  float hsum = 0.f, hpsum = 0.f, qsum = 0.f;
  for (int nn = r; nn < a.n; nn += a.R) {
    const uint32_t* lrow = a.letters + (size_t)nn * a.LW;
    for (int c0 = 0; c0 < a.Lh; c0 += CH) {
      const int cl = min(CH, a.Lh - c0), nl = cl + M - 1;     // positions of the chunk, letters they see
      __syncthreads();
      for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        const uint32_t l = letter_at(a.m, lrow, c0 + i);
        lb[i] = (unsigned char)l;
        // every visible position once: the chunk's own positions, the last chunk also the M - 1 behind them
        // (integer counts: the same total whatever the order)
        if (k == 0 && (i < cl || c0 + cl == a.Lh)) atomicAdd(&cnt[l], 1u);
      }
      __syncthreads();
      for (int s = threadIdx.x; s < cl; s += blockDim.x) {
        float x = bk, xr = bk;
        for (int j = 0; j < M; ++j) {
          const int l = lb[s + j];
          x += Wk[l * M + j];
          if (a.m.ds) xr += Wk[(A - 1 - l) * M + (M - 1 - j)];
        }
        if (a.pool > 1) { X[s] = x; Xp[s] = xr; continue; }
        const float p = sigmoid_x(x), pp = a.m.ds ? sigmoid_x(xr) : 0.f, q = a.want_sparsity ? p * (1.0f - p) : 0.f;
        P[s] = p; Pp[s] = pp; Q[s] = q;
        hsum += p; hpsum += pp; qsum += q;
      }
      __syncthreads();
      if (a.pool > 1) {
        // pooled units (convRBM.py:245-257): P_s = exp(x_s) / (pool + sum_group exp(x_j)); the sparsity slope is
        // P_s (1 - sum of the group's P).  Chunks start on group boundaries.
        for (int s = threadIdx.x; s < cl; s += blockDim.x) {
          const int g0 = s - s % a.pool;
          float p2[2] = {0.f, 0.f}, S2[2] = {0.f, 0.f};
          for (int st = 0; st <= (a.m.ds ? 1 : 0); ++st) {
            const float* xs = st ? Xp : X;
            float mx = 0.f;
            for (int j = 0; j < a.pool; ++j) mx = fmaxf(mx, xs[g0 + j]);
            const float base = (float)a.pool * __expf(-mx);
            float den = base;
            for (int j = 0; j < a.pool; ++j) den += __expf(xs[g0 + j] - mx);
            p2[st] = __expf(xs[s] - mx) / den;
            S2[st] = (den - base) / den;
          }
          const float q = a.want_sparsity ? p2[0] * (1.0f - S2[0]) : 0.f;
          P[s] = p2[0]; Pp[s] = p2[1]; Q[s] = q;
          hsum += p2[0]; hpsum += p2[1]; qsum += q;
        }
        __syncthreads();
      }
      if (by_column) {
        const int j = (int)threadIdx.x % M, sl = (int)threadIdx.x / M;
        if (sl < ncs)
          for (int s = sl; s < cl; s += ncs) {
            const int l = lb[s + j];
            const float p = P[s], pp = Pp[s], q = Q[s];
#pragma unroll
            for (int al = 0; al < 4; ++al) {
              const float mk = l == al ? 1.0f : 0.0f;
              c_vh[al] = fmaf(mk, p, c_vh[al]); c_vhp[al] = fmaf(mk, pp, c_vhp[al]); c_sw[al] = fmaf(mk, q, c_sw[al]);
            }
          }
      } else if (nsl > 0) {
        // 4 M <= blockDim: thread (e, slice) takes every nsl-th position of the chunk for its (letter, column) -- branch-free
        // (a match contributes P * 1, anything else P * 0), so that the loads of consecutive positions overlap
        const int e = (int)threadIdx.x % AM, sl = (int)threadIdx.x / AM;
        if (sl < nsl) {
          const int al = e / M, j = e - al * M;
          float s0 = 0.f, s1 = 0.f, s2 = 0.f;
          for (int s = sl; s < cl; s += nsl) {
            const float mk = lb[s + j] == al ? 1.0f : 0.0f;
            s0 = fmaf(mk, P[s], s0); s1 = fmaf(mk, Pp[s], s1); s2 = fmaf(mk, Q[s], s2);
          }
          vh[0] += s0; vhp[0] += s1; sw[0] += s2;
        }
      } else {
#pragma unroll
        for (int t = 0; t < BIG_ST; ++t) {
          const int e = (int)threadIdx.x + t * (int)blockDim.x;
          if (e < AM) {
            const int al = e / M, j = e - al * M;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int s = 0; s < cl; ++s) {
              const float mk = lb[s + j] == al ? 1.0f : 0.0f;
              s0 = fmaf(mk, P[s], s0); s1 = fmaf(mk, Pp[s], s1); s2 = fmaf(mk, Q[s], s2);
            }
            vh[t] += s0; vhp[t] += s1; sw[t] += s2;
          }
        }
      }
    }
  }
  float* out = a.partials + (size_t)r * a.row;
  if (by_column) {
    // the slices of a column are added in slice order through LDS: red[kind * 4 + letter][slice * M + column]
    __syncthreads();
    const int j = (int)threadIdx.x % M, sl = (int)threadIdx.x / M, RS = (int)blockDim.x;
    if (sl < ncs)
#pragma unroll
      for (int al = 0; al < 4; ++al) {
        red[(0 + al) * RS + sl * M + j] = c_vh[al]; red[(4 + al) * RS + sl * M + j] = c_vhp[al]; red[(8 + al) * RS + sl * M + j] = c_sw[al];
      }
    __syncthreads();
    for (int e = (int)threadIdx.x; e < AM; e += (int)blockDim.x) {
      const int al = e / M, jj = e - al * M;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int i = 0; i < ncs; ++i) { t0 += red[(0 + al) * RS + i * M + jj]; t1 += red[(4 + al) * RS + i * M + jj]; t2 += red[(8 + al) * RS + i * M + jj]; }
      out[a.off_vh0 + (size_t)k * AM + e] = t0;
      if (a.m.ds) out[a.off_vh1 + (size_t)k * AM + e] = t1;
      if (a.want_sparsity) out[a.off_sw + (size_t)k * AM + e] = t2;
    }
    __syncthreads();
  } else if (nsl > 0) {
    // the slices of a (letter, column) are added in slice order through LDS
    __syncthreads();
    const int e = (int)threadIdx.x % AM, sl = (int)threadIdx.x / AM;
    const int RS = (int)blockDim.x;                         // stride of a kind (nsl * AM <= blockDim)
    if (sl < nsl) { red[0 * RS + sl * AM + e] = vh[0]; red[1 * RS + sl * AM + e] = vhp[0]; red[2 * RS + sl * AM + e] = sw[0]; }
    __syncthreads();
    if ((int)threadIdx.x < AM) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
      for (int i = 0; i < nsl; ++i) { t0 += red[0 * RS + i * AM + e]; t1 += red[1 * RS + i * AM + e]; t2 += red[2 * RS + i * AM + e]; }
      out[a.off_vh0 + (size_t)k * AM + e] = t0;
      if (a.m.ds) out[a.off_vh1 + (size_t)k * AM + e] = t1;
      if (a.want_sparsity) out[a.off_sw + (size_t)k * AM + e] = t2;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int t = 0; t < BIG_ST; ++t) {
      const int e = (int)threadIdx.x + t * (int)blockDim.x;
      if (e < AM) {
        out[a.off_vh0 + (size_t)k * AM + e] = vh[t];
        if (a.m.ds) out[a.off_vh1 + (size_t)k * AM + e] = vhp[t];
        if (a.want_sparsity) out[a.off_sw + (size_t)k * AM + e] = sw[t];
      }
    }
  }
  const float H = big_block_sum(hsum, xch), Hp = big_block_sum(hpsum, xch), Sb = big_block_sum(qsum, xch);
  if (threadIdx.x == 0) {
    out[a.off_h0 + k] = H;
    if (a.m.ds) out[a.off_h1 + k] = Hp;
    if (a.want_sparsity) out[a.off_sb + k] = Sb;
  }
  if (k == 0) {
    __syncthreads();
    for (int l = threadIdx.x; l < A; l += blockDim.x) out[a.off_v + l] = (float)cnt[l];
  }
}

// ---- the update (convRBM.py:358-371, :415-436, :440-451), element-wise and in place, any number of blocks
__global__ void __launch_bounds__(256) big_update_kernel(UpdateArgs a) {
  const int K = a.K, M = a.M, A = a.A, KAM = K * A * M;
  const float* S = a.sums;
  const float n_d = S[a.n_d], n_m = S[a.n_m];
  const float cnt_d = n_d * (float)(a.L_data - M + 1), cnt_m = n_m * (float)a.Lf;
  const int d_vh = a.data_off, d_vhp = d_vh + KAM, d_h = d_vh + 2 * KAM, d_hp = d_h + K;
  const int d_sw = d_vh + 2 * KAM + 2 * K, d_sb = d_sw + KAM, d_v = d_sb + K;
  const int m_vh = a.model_off, m_vhp = m_vh + KAM, m_h = m_vh + 2 * KAM, m_hp = m_h + K, m_v = m_hp + K;
  const float q = a.rho;
  const int total = KAM + K + A;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (i < KAM) {
      const int k = i / (A * M), al = (i / M) % A, j = i % M;
      const int ri = (k * A + (A - 1 - al)) * M + (M - 1 - j);
      float gd = S[d_vh + i] / cnt_d, gm = S[m_vh + i] / cnt_m;
      if (a.ds) {
        gd = 0.5f * (gd + S[d_vhp + ri] / cnt_d);
        gm = 0.5f * (gm + S[m_vhp + ri] / cnt_m);
      }
      const float p = S[d_h + k] / cnt_d;
      const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
      const float reg = -g * S[d_sw + i] / cnt_d;
      const float v = a.momentum * a.vW[i] + a.lr * (gd - gm - a.lambda_rate * reg);
      a.ovW[i] = v;
      a.oW[i] = a.W[i] + v;
    } else if (i < KAM + K) {
      const int k = i - KAM;
      const float dh = S[d_h + k];
      float gd = dh / cnt_d, gm = S[m_h + k] / cnt_m;
      if (a.ds) {
        gd = 0.5f * (gd + S[d_hp + k] / cnt_d);
        gm = 0.5f * (gm + S[m_hp + k] / cnt_m);
      }
      const float p = dh / cnt_d;
      const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
      const float reg = -g * S[d_sb + k] / cnt_d;
      const float v = a.momentum * a.vb[k] + a.lr * (gd - gm - a.lambda_rate * reg);
      a.ovb[k] = v;
      a.ob[k] = a.b[k] + v;
    } else {
      const int al = i - KAM - K;
      const float nd = n_d * (float)a.L_data, nm = n_m * (float)(a.Lf + M - 1);
      const float gd = S[d_v + al] / nd + S[d_v + A - 1 - al] / nd;     // a += a[::-1]  (:345)
      const float gm = S[m_v + al] / nm + S[m_v + A - 1 - al] / nm;
      const float v = a.momentum * a.vc[al] + a.lr * (gd - gm);
      a.ovc[al] = v;
      a.oc[al] = a.c[al] + v;
    }
  }
}

// ---- free energy (convRBM.py:657-697) and the motif-hit summaries (utils.py:113-116, :154, :242-244, :305): one block per
// sequence, the motifs one after the other (filter in LDS), threads over the hidden positions.
struct BigEvalArgs {
  BigModel m;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  float* fe;                       // (n)   free-energy mode, may be null
  float* fem;                      // (n,K) free-energy mode, may be null
  float* hmax;                     // (n,K) hit mode: max over positions
  float* hmean;                    // (n,K) hit mode: mean over positions
  unsigned long long* pos_fx;      // (K,Lh) hit mode: fixed-point sums over sequences (HIT_FX units), may be null
  int32_t hits;                    // 0: free energy, 1: hit summaries
  int32_t pool;                    // pooling: the units of `pool` consecutive positions compete
};

__global__ void __launch_bounds__(256) big_eval_kernel(BigEvalArgs a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const int K = a.m.K, M = a.m.M, A = a.m.A, AM = A * M;
  float* Wk = smem;                                                  // [A][M]
  float* xch = Wk + ((AM + 3) & ~3);                                 // [16]
  unsigned char* lb = reinterpret_cast<unsigned char*>(xch + 16);    // [L]
  for (int nn = blockIdx.x; nn < a.n; nn += gridDim.x) {
    const uint32_t* lrow = a.letters + (size_t)nn * a.LW;
    __syncthreads();
    float csl = 0.f;
    for (int p = threadIdx.x; p < a.L; p += blockDim.x) {
      const uint32_t l = letter_at(a.m, lrow, p);
      lb[p] = (unsigned char)l;
      csl += a.m.c[l];
    }
    const float cs = big_block_sum(csl, xch);
    float tot = 0.f;
    for (int k = 0; k < K; ++k) {
      __syncthreads();
      for (int i = threadIdx.x; i < AM; i += blockDim.x) Wk[i] = a.m.W[(size_t)k * AM + i];
      __syncthreads();
      const float bk = a.m.b[k];
      float part = 0.f, mx = 0.f;
      auto act = [&](int s, float& x, float& xr) {
        x = bk; xr = bk;
        for (int j = 0; j < M; ++j) {
          const int l = lb[s + j];
          x += Wk[l * M + j];
          xr += Wk[(A - 1 - l) * M + (M - 1 - j)];
        }
      };
      if (a.pool > 1) {
        // a thread takes whole pooling groups: log(1 + sum_j exp(x_j)) per group (free energy, convRBM.py:664-665) or the
        // pooled probabilities exp(x_j) / (pool + sum exp) (hits), in two passes over the group's activations
        for (int g0 = (int)threadIdx.x * a.pool; g0 < a.Lh; g0 += (int)blockDim.x * a.pool) {
          float m0 = 0.f, m1 = 0.f, x, xr;
          for (int j = 0; j < a.pool; ++j) {
            act(g0 + j, x, xr);
            const float z = a.hits && !a.m.ds ? x + xr : x;
            m0 = fmaxf(m0, z);
            m1 = fmaxf(m1, xr);
          }
          float d0 = 0.f, d1 = 0.f;
          for (int j = 0; j < a.pool; ++j) {
            act(g0 + j, x, xr);
            d0 += __expf((a.hits && !a.m.ds ? x + xr : x) - m0);
            d1 += __expf(xr - m1);
          }
          if (a.hits) {
            const float den = d0 + (float)a.pool * __expf(-m0);
            for (int j = 0; j < a.pool; ++j) {
              act(g0 + j, x, xr);
              const float p = __expf((a.m.ds ? x : x + xr) - m0) / den;
              part += p;
              mx = fmaxf(mx, p);
              if (a.pos_fx) atomicAdd(a.pos_fx + (size_t)k * a.Lh + g0 + j, to_fx(p));
            }
          } else {
            part += m0 + __logf(d0 + __expf(-m0));
            if (a.m.ds) part += m1 + __logf(d1 + __expf(-m1));
          }
        }
      } else
      for (int s = threadIdx.x; s < a.Lh; s += blockDim.x) {
        float x, xr;
        act(s, x, xr);
        if (a.hits) {
          // convRBM.py:507-514: doublestranded -> sigma(x); single-stranded -> sigma(x + x')
          const float p = sigmoid_x(a.m.ds ? x : x + xr);
          part += p;
          mx = fmaxf(mx, p);
          if (a.pos_fx) atomicAdd(a.pos_fx + (size_t)k * a.Lh + s, to_fx(p));
        } else {
          // softplus(x) = max(x, 0) + log1p(exp(-|x|))
          part += fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
          if (a.m.ds) part += fmaxf(xr, 0.f) + log1pf(__expf(-fabsf(xr)));
        }
      }
      const float v = big_block_sum(part, xch);
      if (a.hits) {
        const float m = big_block_max(mx, xch);
        if (threadIdx.x == 0) {
          if (a.hmax) a.hmax[(size_t)nn * K + k] = m;
          if (a.hmean) a.hmean[(size_t)nn * K + k] = v / (float)a.Lh;
        }
      } else {
        tot += v;
        if (threadIdx.x == 0 && a.fem) a.fem[(size_t)nn * K + k] = -v - cs;
      }
    }
    if (!a.hits && threadIdx.x == 0 && a.fe) a.fe[nn] = (-tot - cs) / (float)a.L;
  }
}


// ---- Statistics of large DNA models on the matrix cores, a slab of motifs at a time -----------------------------------
// VH[k], H[k] and the sparsity sums of motif k depend on that motif's filter alone (convRBM.py:327-371, :440-451), so a model
// whose tables exceed the LDS is, for the statistics, a row of independent sub-models: crbm_api.hip runs the SPECIALISED
// statistics kernel (stats_mfma_body, compiled for a slab of <= 64 motifs) once per slab on the slab's own gather table,
// and this kernel reduces the slab's partial rows column by column -- like reduce_partials_body -- straight into the
// slab's columns of the full model's sums ([vh K*4*M][vh' K*4*M][h K][h' K][sw K*4*M][sb K][v 4] n, the model half
// without sw, sb).  The letter counts and n do not depend on the motifs: every slab writes the same values.  Slabs may
// overlap (the same sums twice) and the last one may reach past motif K (zero padding of W and b: those columns are dropped).
struct SlabReduceArgs {
  const float* partials;   // [nrows][row] partial rows of the slab model
  float* sums;             // the full model's sums of this half (d_sums + data_off or model_off)
  int32_t nrows, row;      // row = 3 Ks 4M + 3 Ks + 4
  int32_t Ks, k0, K, M4;   // motifs of the slab, first motif of the last slab (slab_k0: blockIdx.y is the slab), motifs of the model, 4 * motif_length
  long long partial_stride;   // floats between the partial rows of consecutive slabs
  int32_t ds, want_sparsity;
  int32_t skip_begin, skip_len;   // columns of the FULL row that are not carried (model half: sw, sb)
  float n_value;
};

__global__ void __launch_bounds__(1024) slab_reduce_kernel(SlabReduceArgs a) {
  __shared__ float part[32][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int ngrp = (int)(blockDim.x >> 5);
  const int r = blockIdx.x * 32 + col;
  const int Ks = a.Ks, K = a.K, KAMs = a.Ks * a.M4, KAM = a.K * a.M4;
  const SlabPlan plan = {a.Ks, a.K, a.k0};
  const int k0 = slab_k0(plan, (int)blockIdx.y);
  const float* partials = a.partials + (size_t)blockIdx.y * a.partial_stride;
  int dst = -1;          // column of the full row
  bool valid = false;
  if (r < a.row) {
    int kk;              // motif of the slab this column belongs to (-1: none)
    if (r < KAMs) { kk = r / a.M4; dst = k0 * a.M4 + r; valid = true; }
    else if (r < 2 * KAMs) { kk = (r - KAMs) / a.M4; dst = KAM + k0 * a.M4 + (r - KAMs); valid = a.ds != 0; }
    else if (r < 2 * KAMs + Ks) { kk = r - 2 * KAMs; dst = 2 * KAM + k0 + kk; valid = true; }
    else if (r < 2 * KAMs + 2 * Ks) { kk = r - 2 * KAMs - Ks; dst = 2 * KAM + K + k0 + kk; valid = a.ds != 0; }
    else if (r < 3 * KAMs + 2 * Ks) { kk = (r - 2 * KAMs - 2 * Ks) / a.M4; dst = 2 * KAM + 2 * K + k0 * a.M4 + (r - 2 * KAMs - 2 * Ks); valid = a.want_sparsity != 0; }
    else if (r < 3 * KAMs + 3 * Ks) { kk = r - 3 * KAMs - 2 * Ks; dst = 3 * KAM + 2 * K + k0 + kk; valid = a.want_sparsity != 0; }
    else { kk = -1; dst = 3 * KAM + 3 * K + (r - 3 * KAMs - 3 * Ks); valid = true; }
    if (kk >= 0 && k0 + kk >= K) { dst = -1; valid = false; }      // the last slab may reach past the model's end (its padding)
  }
  float tsum = 0.f;
  if (valid) {      // eight rows in flight per thread, fixed order (reduce_partials_body)
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f;
    int i = grp;
    for (; i + 7 * ngrp < a.nrows; i += 8 * ngrp) {
      const float v0 = partials[(size_t)i * a.row + r], v1 = partials[(size_t)(i + ngrp) * a.row + r];
      const float v2 = partials[(size_t)(i + 2 * ngrp) * a.row + r], v3 = partials[(size_t)(i + 3 * ngrp) * a.row + r];
      const float v4 = partials[(size_t)(i + 4 * ngrp) * a.row + r], v5 = partials[(size_t)(i + 5 * ngrp) * a.row + r];
      const float v6 = partials[(size_t)(i + 6 * ngrp) * a.row + r], v7 = partials[(size_t)(i + 7 * ngrp) * a.row + r];
      t0 += v0; t1 += v1; t2 += v2; t3 += v3; t4 += v4; t5 += v5; t6 += v6; t7 += v7;
    }
    for (; i < a.nrows; i += ngrp) t0 += partials[(size_t)i * a.row + r];
    tsum = ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));
  }
  part[grp][col] = tsum;
  __syncthreads();
  if (grp == 0 && dst >= 0) {
    const bool skipped = dst >= a.skip_begin && dst < a.skip_begin + a.skip_len;
    if (!skipped) {
      float s = 0.f;
      for (int g = 0; g < ngrp; ++g) s += part[g][col];
      a.sums[dst < a.skip_begin ? dst : dst - a.skip_len] = s;     // (columns of a kind the model does not have: zero, as the plain reduction writes them)
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.sums[(3 * KAM + 3 * K + 4) - a.skip_len] = a.n_value;
}

// Free energy of a model evaluated slab by slab (slab_fe_body: every slab leaves fem[n][kk] = -v_k - cs, v_k = sum over positions
// and strands of softplus(x_k), cs = sum_p c[letter(p)] -- free_energy_body).  One wave per sequence: cs exactly as
// free_energy_body forms it (same terms, same order), then -v_k = fem + cs motif by motif, F = (-sum_k v_k - cs) / L
// (convRBM.py:657-697) and, where asked for, the per-motif free energies in the model's own (n, K) layout.
struct SlabFeCombineArgs {
  const float* scratch;       // [nslab][n][Ks]
  const float* c_log2e;       // log2(e) * c as the slab tables hold it (C::OFF_C of any slab image)
  const uint32_t* letters;
  int32_t n, L, LW;
  int32_t Ks, K, last_k0, nslab;
  float* fe;                  // (n), may be null
  float* fem;                 // (n, K), may be null
};

__global__ void __launch_bounds__(256) slab_fe_combine_kernel(SlabFeCombineArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const float c0 = LN2 * a.c_log2e[0], c1 = LN2 * a.c_log2e[1], c2 = LN2 * a.c_log2e[2], c3 = LN2 * a.c_log2e[3];
  for (int nn = blockIdx.x * nwaves + wave; nn < a.n; nn += gridDim.x * nwaves) {
    const uint32_t* row = a.letters + (size_t)nn * a.LW;
    float cs = 0.f;
    for (int p = lane; p < a.L; p += 64) {
      const uint32_t l = (row[p >> 4] >> (2 * (p & 15))) & 3u;
      cs += l == 0u ? c0 : l == 1u ? c1 : l == 2u ? c2 : c3;
    }
    cs = wave_sum(cs);
    float part = 0.f;            // - sum of v_k over this lane's motifs
    for (int k = lane; k < a.K; k += 64) {
      const int y = k >= a.last_k0 ? a.nslab - 1 : k / a.Ks;        // (the last slab owns everything from its first motif on)
      const int k0 = y == a.nslab - 1 ? a.last_k0 : y * a.Ks;
      const float f = a.scratch[((size_t)y * a.n + nn) * a.Ks + (k - k0)];
      if (a.fem) a.fem[(size_t)nn * a.K + k] = f;
      part += f + cs;
    }
    const float tot = wave_sum(part);
    if (lane == 0 && a.fe) a.fe[nn] = (tot - cs) / (float)a.L;
  }
}
