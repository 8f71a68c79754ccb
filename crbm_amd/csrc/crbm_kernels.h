// CDNA4 (gfx950) kernels of the CRBM hot path.  Wave64 throughout.
//
// Design (details in DESIGN.md):
//  * The visible layer is one-hot wherever the forward correlation is applied
//    (reference sequences.py:28-31, convRBM.py:304-310), so a sequence is kept
//    as 2-bit letters and x[k,s] = b[k] + sum_j W[k, letter[s+j], j] becomes a
//    gather-add.  Letters are taken G at a time: LDS holds pre-summed rows
//    T[g][letter-tuple][k], one ds_read_b128 feeds four motifs.
//  * The hidden layer is binary wherever the transposed convolution is applied
//    (convRBM.py:259-267), so chain state is a K-bit mask per hidden position
//    and the top-down pass adds W[k,:,j] (one float4) per set bit.
//  * One workgroup owns whole chains, so a k-step Gibbs chain runs entirely in
//    LDS; HBM sees the masks once in and once out per launch.
//  * MFMA is not used: the transposed conv has output width 4 and the forward
//    has one-hot operands (BASELINE.json north_star).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "crbm_layout.h"

namespace crbm {

// ---------------------------------------------------------------------------
// Philox-4x32-10 (Random123 constants); same counters as oracle/crbm_oracle.py
// ---------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// component i (0..3) without dynamic register indexing
__device__ __forceinline__ uint32_t philox_pick(const Philox4& r, int i) {
  const uint32_t lo = (i & 1) ? r.v[1] : r.v[0];
  const uint32_t hi = (i & 1) ? r.v[3] : r.v[2];
  return (i & 2) ? hi : lo;
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

__device__ __forceinline__ uint32_t rng_word2(uint32_t kind, uint32_t strand, uint32_t kgroup) {
  return (kind << 28) | (strand << 24) | kgroup;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t i, const FastDiv& f) {
  return f.d <= 1 ? i : __umulhi(i, f.inv);
}

__device__ __forceinline__ float sigmoidf_fast(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// Gather tables.  T[g][r][k] = sum_{t<G, j=gG+t<M} Wf[k][(r>>2t)&3][j]
// (+ b[k] folded into group 0), Wf = W or rc(W) = W[k][3-a][M-1-j]
// (convRBM.py:241, :285).  Pad columns k >= K get -1e30 in group 0 so that
// sigmoid -> 0, softplus -> 0 and no bit is ever sampled there.
// ---------------------------------------------------------------------------
__device__ inline void build_gather_table(float* T, const ModelView& mv, int KP, bool rc) {
  const int total = mv.ngroups * mv.rows * KP;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int k = idx % KP;
    const int r = (idx / KP) % mv.rows;
    const int g = idx / (KP * mv.rows);
    float acc = 0.f;
    if (k < mv.K) {
      for (int t = 0; t < mv.G; ++t) {
        const int j = g * mv.G + t;
        if (j < mv.M) {
          const int a = (r >> (2 * t)) & 3;
          acc += rc ? mv.W[(k * 4 + (3 - a)) * mv.M + (mv.M - 1 - j)] : mv.W[(k * 4 + a) * mv.M + j];
        }
      }
      if (g == 0) acc += mv.b[k];
    } else if (g == 0) {
      acc = -1e30f;
    }
    T[idx] = acc;
  }
}

// M letters (2 bits each) starting at position s of a packed row.
__device__ __forceinline__ uint64_t letter_window(const uint32_t* w, int s, int M) {
  const int i = s >> 4;
  const int sh = (s & 15) * 2;
  const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
  uint64_t win = lo >> sh;
  if (2 * M + sh > 64) win |= (uint64_t)w[i + 2] << (64 - sh);
  if (M < 32) win &= (1ull << (2 * M)) - 1ull;
  return win;
}

template <int NQ, bool ACCUMULATE = false>
__device__ __forceinline__ void conv_gather(const float* T, uint64_t win, const ModelView& mv, float (&x)[4 * NQ]) {
  constexpr int KP = 4 * NQ;
  if (!ACCUMULATE) {
#pragma unroll
    for (int q = 0; q < KP; ++q) x[q] = 0.f;
  }
  const int gb = 2 * mv.G;
  const uint32_t rmask = (uint32_t)mv.rows - 1u;
  for (int g = 0; g < mv.ngroups; ++g) {
    const uint32_t r = (uint32_t)win & rmask;
    win >>= gb;
    const float4* row = reinterpret_cast<const float4*>(T + (size_t)(g * mv.rows + r) * KP);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 t = row[q];
      x[4 * q + 0] += t.x;
      x[4 * q + 1] += t.y;
      x[4 * q + 2] += t.z;
      x[4 * q + 3] += t.w;
    }
  }
}

#ifdef CRBM_DEFINE_MISC_KERNELS
// ---------------------------------------------------------------------------
// one-hot fp32 (n,1,4,L) -> packed letters [n][LW]; flags[0] |= 1 on a column
// that is not exactly one-hot.
// ---------------------------------------------------------------------------
struct EncodeArgs {
  const float* v;
  uint32_t* letters;
  uint32_t* flags;
  int32_t n, L, LW;
};

__global__ void encode_onehot_kernel(EncodeArgs a) {
  const long total = (long)a.n * a.LW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.LW);
    const int w = (int)(i - (long)nn * a.LW);
    uint32_t word = 0;
    bool bad = false;
    const float* base = a.v + (size_t)nn * 4 * a.L;
    for (int t = 0; t < 16; ++t) {
      const int p = w * 16 + t;
      if (p < a.L) {
        const float v0 = base[p], v1 = base[a.L + p], v2 = base[2 * a.L + p], v3 = base[3 * a.L + p];
        const int ones = (v0 == 1.f) + (v1 == 1.f) + (v2 == 1.f) + (v3 == 1.f);
        const int zeros = (v0 == 0.f) + (v1 == 0.f) + (v2 == 0.f) + (v3 == 0.f);
        bad |= (ones != 1) | (zeros != 3);
        const uint32_t l = (v1 == 1.f) ? 1u : (v2 == 1.f) ? 2u : (v3 == 1.f) ? 3u : 0u;
        word |= l << (2 * t);
      }
    }
    a.letters[i] = word;
    if (bad) atomicOr(a.flags, 1u);
  }
}

// packed letters -> one-hot fp32 (n,1,4,L)
struct DecodeArgs {
  const uint32_t* letters;
  float* v;
  int32_t n, L, LW;
};

__global__ void decode_onehot_kernel(DecodeArgs a) {
  const long total = (long)a.n * a.L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.L);
    const int p = (int)(i - (long)nn * a.L);
    const uint32_t l = (a.letters[(size_t)nn * a.LW + (p >> 4)] >> (2 * (p & 15))) & 3u;
    float* base = a.v + (size_t)nn * 4 * a.L + p;
    base[0] = l == 0 ? 1.f : 0.f;
    base[a.L] = l == 1 ? 1.f : 0.f;
    base[2 * a.L] = l == 2 ? 1.f : 0.f;
    base[3 * a.L] = l == 3 ? 1.f : 0.f;
  }
}

// dense hidden (n,K,1,Lh) <-> K-bit masks [n][Lh][NW]; flags[0] |= 2 if not 0/1
struct HiddenPackArgs {
  float* dense;
  uint32_t* masks;
  uint32_t* flags;
  int32_t n, K, Lh, NW;
};

__global__ void pack_hidden_kernel(HiddenPackArgs a) {
  const long total = (long)a.n * a.Lh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.Lh);
    const int s = (int)(i - (long)nn * a.Lh);
    bool bad = false;
    for (int w = 0; w < a.NW; ++w) {
      uint32_t m = 0;
      for (int bit = 0; bit < 32; ++bit) {
        const int k = w * 32 + bit;
        if (k < a.K) {
          const float v = a.dense[((size_t)nn * a.K + k) * a.Lh + s];
          bad |= !(v == 0.f || v == 1.f);
          m |= (v == 1.f ? 1u : 0u) << bit;
        }
      }
      a.masks[i * a.NW + w] = m;
    }
    if (bad) atomicOr(a.flags, 2u);
  }
}

__global__ void unpack_hidden_kernel(HiddenPackArgs a) {
  const long total = (long)a.n * a.Lh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.Lh);
    const int s = (int)(i - (long)nn * a.Lh);
    for (int k = 0; k < a.K; ++k) {
      const uint32_t m = a.masks[i * a.NW + (k >> 5)];
      a.dense[((size_t)nn * a.K + k) * a.Lh + s] = (float)((m >> (k & 31)) & 1u);
    }
  }
}

#endif  // CRBM_DEFINE_MISC_KERNELS

// ---------------------------------------------------------------------------
// h_given_v, dense outputs: _bottomUpActivity / _bottomUpProbability /
// _bottomUpSample (convRBM.py:238-275) and motifHitProbs (:507-514).
// mode 0: forward strand, 1: reverse-complement strand, 2: sigma(x + x').
// ---------------------------------------------------------------------------
struct HgvArgs {
  ModelView mv;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  int32_t TS;          // sequences per tile
  FastDiv divLh;
  int32_t mode;
  float* act;
  float* prob;
  float* sample;
  unsigned long long* ones;   // += number of sampled ones (may be null)
  RngView rng;
  uint32_t kind;
};

template <int NQ>
__global__ void __launch_bounds__(256) hgv_kernel(HgvArgs a) {
  constexpr int KP = 4 * NQ;
  HIP_DYNAMIC_SHARED(float, smem);
  const ModelView& mv = a.mv;
  const int tab = mv.ngroups * mv.rows * KP;
  float* T0 = smem;
  float* T1 = smem + tab;
  build_gather_table(T0, mv, KP, a.mode == 1);
  if (a.mode == 2) build_gather_table(T1, mv, KP, true);
  __syncthreads();
  const bool want_sample = (a.sample != nullptr) || (a.ones != nullptr);
  const uint32_t strand = a.mode == 1 ? 1u : 0u;
  unsigned long long cnt = 0;
  const int ntiles = (a.n + a.TS - 1) / a.TS;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * a.TS;
    const int ns = min(a.TS, a.n - n0);
    const uint32_t items = (uint32_t)ns * (uint32_t)a.Lh;
    for (uint32_t i = threadIdx.x; i < items; i += blockDim.x) {
      const uint32_t nl = fastdiv(i, a.divLh);
      const int s = (int)(i - nl * (uint32_t)a.Lh);
      const int nn = n0 + (int)nl;
      const uint64_t win = letter_window(a.letters + (size_t)nn * a.LW, s, mv.M);
      float x[KP];
      conv_gather<NQ>(T0, win, mv, x);
      if (a.mode == 2) conv_gather<NQ, true>(T1, win, mv, x);
      Philox4 r;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        if (k < mv.K) {
          const size_t idx = ((size_t)nn * mv.K + k) * a.Lh + s;
          const float xv = x[k];
          if (a.act) a.act[idx] = xv;
          const float p = sigmoidf_fast(xv);
          if (a.prob) a.prob[idx] = p;
          if (want_sample) {
            if ((k & 3) == 0)
              r = philox4x32_10(a.rng.seq_offset + (uint32_t)nn, (uint32_t)s,
                                rng_word2(a.kind, strand, (uint32_t)(k >> 2)), a.rng.step,
                                a.rng.seed_lo, a.rng.seed_hi);
            const float hv = p > u01(r.v[k & 3]) ? 1.f : 0.f;
            if (a.sample) a.sample[idx] = hv;
            cnt += (unsigned long long)hv;
          }
        }
      }
    }
  }
  if (a.ones && cnt) atomicAdd(a.ones, cnt);
}

#ifdef CRBM_DEFINE_MISC_KERNELS
// ---------------------------------------------------------------------------
// v_given_h from DENSE hidden tensors (any finite values): _topDownActivity,
// _topDownProbability, _topDownSample (convRBM.py:277-325).  API/test pass;
// the training chain uses gibbs_kernel.
// ---------------------------------------------------------------------------
struct VghArgs {
  ModelView mv;
  const float* hid;
  const float* hidp;    // null when single-stranded
  int32_t n, Lh, L;
  int32_t TS;
  FastDiv divL;
  float* act;
  float* prob;
  float* sample;
  RngView rng;
  uint32_t kind;
};

__global__ void __launch_bounds__(256) vgh_dense_kernel(VghArgs a) {
  HIP_DYNAMIC_SHARED(float, smem);
  const ModelView& mv = a.mv;
  const int K = mv.K, M = mv.M;
  float4* W4 = reinterpret_cast<float4*>(smem);   // [j][k] -> W[k][0..3][j]
  for (int idx = threadIdx.x; idx < M * K; idx += blockDim.x) {
    const int j = idx / K, k = idx - j * K;
    W4[idx] = make_float4(mv.W[(k * 4 + 0) * M + j], mv.W[(k * 4 + 1) * M + j],
                          mv.W[(k * 4 + 2) * M + j], mv.W[(k * 4 + 3) * M + j]);
  }
  __syncthreads();
  const float c0 = mv.c[0], c1 = mv.c[1], c2 = mv.c[2], c3 = mv.c[3];
  const int ntiles = (a.n + a.TS - 1) / a.TS;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * a.TS;
    const int ns = min(a.TS, a.n - n0);
    const uint32_t items = (uint32_t)ns * (uint32_t)a.L;
    for (uint32_t i = threadIdx.x; i < items; i += blockDim.x) {
      const uint32_t nl = fastdiv(i, a.divL);
      const int p = (int)(i - nl * (uint32_t)a.L);
      const int nn = n0 + (int)nl;
      float y0 = c0, y1 = c1, y2 = c2, y3 = c3;
      const int jlo = max(0, p - a.Lh + 1), jhi = min(M - 1, p);
      for (int k = 0; k < K; ++k) {
        const float* hrow = a.hid + ((size_t)nn * K + k) * a.Lh;
        const float* hprow = a.hidp ? a.hidp + ((size_t)nn * K + k) * a.Lh : nullptr;
        for (int j = jlo; j <= jhi; ++j) {
          const float hv = hrow[p - j];
          const float4 w = W4[j * K + k];
          y0 = fmaf(w.x, hv, y0); y1 = fmaf(w.y, hv, y1); y2 = fmaf(w.z, hv, y2); y3 = fmaf(w.w, hv, y3);
          if (hprow) {   // rc(W)[k,a,j] = W[k,3-a,M-1-j]
            const float hp = hprow[p - j];
            const float4 wr = W4[(M - 1 - j) * K + k];
            y0 = fmaf(wr.w, hp, y0); y1 = fmaf(wr.z, hp, y1); y2 = fmaf(wr.y, hp, y2); y3 = fmaf(wr.x, hp, y3);
          }
        }
      }
      const size_t o = (size_t)nn * 4 * a.L + p;
      if (a.act) { a.act[o] = y0; a.act[o + a.L] = y1; a.act[o + 2 * a.L] = y2; a.act[o + 3 * a.L] = y3; }
      const float mx = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
      const float e0 = __expf(y0 - mx), e1 = __expf(y1 - mx), e2 = __expf(y2 - mx), e3 = __expf(y3 - mx);
      const float sum = (e0 + e1) + (e2 + e3);
      if (a.prob) {
        const float inv = 1.0f / sum;
        a.prob[o] = e0 * inv; a.prob[o + a.L] = e1 * inv; a.prob[o + 2 * a.L] = e2 * inv; a.prob[o + 3 * a.L] = e3 * inv;
      }
      if (a.sample) {
        const Philox4 r = philox4x32_10(a.rng.seq_offset + (uint32_t)nn, (uint32_t)(p >> 2),
                                        rng_word2(a.kind, 0, 0), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
        const float t = u01(philox_pick(r, p & 3)) * sum;
        const int l = (t >= e0) + (t >= e0 + e1) + (t >= (e0 + e1) + e2);
        a.sample[o] = l == 0 ? 1.f : 0.f; a.sample[o + a.L] = l == 1 ? 1.f : 0.f;
        a.sample[o + 2 * a.L] = l == 2 ? 1.f : 0.f; a.sample[o + 3 * a.L] = l == 3 ? 1.f : 0.f;
      }
    }
  }
}

#endif  // CRBM_DEFINE_MISC_KERNELS

// ---------------------------------------------------------------------------
// The persistent-chain kernel: `steps` Gibbs steps
//   v ~ P(v|h,h')  (convRBM.py:317-325)   then   h,h' ~ P(h|v)  (:269-275)
// for every chain of a tile, entirely in LDS (convRBM.py:397-408).
// ---------------------------------------------------------------------------
struct GibbsArgs {
  ModelView mv;
  uint32_t* hm;        // [nchains][Lf][NW] in/out
  uint32_t* hmp;       // reverse strand (ds) or null
  uint32_t* vout;      // [nchains][LWs] letters of the last visible sample
  int32_t nchains, Lf, Lv, Lhp, LWs, S;
  FastDiv divLv, divLf, divRow;   // divRow: / (Lhp*NW)
  int32_t steps;
  RngView rng;
};

template <int NQ>
__global__ void __launch_bounds__(512) gibbs_kernel(GibbsArgs a) {
  constexpr int KP = 4 * NQ;
  constexpr int NW = (NQ <= 8) ? 1 : 2;
  HIP_DYNAMIC_SHARED(float, smem);
  const ModelView& mv = a.mv;
  const int M = mv.M;
  const int tab = mv.ngroups * mv.rows * KP;
  float* Tf = smem;
  float* Tr = Tf + tab;
  float4* Wt = reinterpret_cast<float4*>(Tr + (mv.ds ? tab : 0));   // [jr][slot] -> W[k][0..3][M-1-jr]
  float* cv = reinterpret_cast<float*>(Wt + M * NW * 32);
  uint32_t* hm = reinterpret_cast<uint32_t*>(cv + 4);
  uint32_t* hmp = hm + (size_t)a.S * a.Lhp * NW;
  uint32_t* let = hmp + (mv.ds ? (size_t)a.S * a.Lhp * NW : 0);

  build_gather_table(Tf, mv, KP, false);
  if (mv.ds) build_gather_table(Tr, mv, KP, true);
  for (int idx = threadIdx.x; idx < M * NW * 32; idx += blockDim.x) {
    const int k = idx % (NW * 32), jr = idx / (NW * 32);
    const int j = M - 1 - jr;
    Wt[idx] = k < mv.K ? make_float4(mv.W[(k * 4 + 0) * M + j], mv.W[(k * 4 + 1) * M + j],
                                     mv.W[(k * 4 + 2) * M + j], mv.W[(k * 4 + 3) * M + j])
                       : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (threadIdx.x < 4) cv[threadIdx.x] = mv.c[threadIdx.x];

  const int rowW = a.Lhp * NW;
  const int ntiles = (a.nchains + a.S - 1) / a.S;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * a.S;
    const int ns = min(a.S, a.nchains - n0);
    __syncthreads();
    // chain state -> zero-padded LDS rows (hidden position s sits at s + M-1)
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)(ns * rowW); idx += blockDim.x) {
      const uint32_t nl = fastdiv(idx, a.divRow);
      const uint32_t r = idx - nl * (uint32_t)rowW;
      const int q = (int)(r / NW), w = (int)(r % NW);
      const int s = q - (M - 1);
      const bool in = s >= 0 && s < a.Lf;
      const size_t g = ((size_t)(n0 + nl) * a.Lf + (in ? s : 0)) * NW + w;
      hm[idx] = in ? a.hm[g] : 0u;
      if (mv.ds) hmp[idx] = in ? a.hmp[g] : 0u;
    }
    for (int st = 0; st < a.steps; ++st) {
      for (int idx = threadIdx.x; idx < ns * a.LWs; idx += blockDim.x) let[idx] = 0u;
      __syncthreads();
      // ---- v | h : y[a,p] = c[a] + sum over set bits (k, s=p-j) of W[k,a,j] ----
      for (uint32_t i = threadIdx.x; i < (uint32_t)(ns * a.Lv); i += blockDim.x) {
        const uint32_t nl = fastdiv(i, a.divLv);
        const int p = (int)(i - nl * (uint32_t)a.Lv);
        float y0 = cv[0], y1 = cv[1], y2 = cv[2], y3 = cv[3];
        const uint32_t* mrow = hm + (size_t)nl * rowW + (size_t)p * NW;
        for (int jr = 0; jr < M; ++jr) {
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            uint32_t m = mrow[jr * NW + w];
            while (m) {
              const int bit = __ffs(m) - 1;
              m &= m - 1;
              const float4 t = Wt[(jr * NW + w) * 32 + bit];
              y0 += t.x; y1 += t.y; y2 += t.z; y3 += t.w;
            }
          }
        }
        if (mv.ds) {   // rc strand: rc(W)[k,a,j] = W[k,3-a,M-1-j]
          const uint32_t* prow = hmp + (size_t)nl * rowW + (size_t)p * NW;
          for (int jr = 0; jr < M; ++jr) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              uint32_t m = prow[jr * NW + w];
              while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                const float4 t = Wt[((M - 1 - jr) * NW + w) * 32 + bit];
                y0 += t.w; y1 += t.z; y2 += t.y; y3 += t.x;
              }
            }
          }
        }
        const float mx = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
        const float e0 = __expf(y0 - mx), e1 = __expf(y1 - mx), e2 = __expf(y2 - mx), e3 = __expf(y3 - mx);
        const float sum = (e0 + e1) + (e2 + e3);
        const Philox4 r = philox4x32_10(a.rng.seq_offset + (uint32_t)(n0 + nl), (uint32_t)(p >> 2),
                                        rng_word2(KIND_CHAIN_V, 0, 0), a.rng.step + (uint32_t)st,
                                        a.rng.seed_lo, a.rng.seed_hi);
        const float t = u01(philox_pick(r, p & 3)) * sum;
        const uint32_t l = (uint32_t)(t >= e0) + (uint32_t)(t >= e0 + e1) + (uint32_t)(t >= (e0 + e1) + e2);
        atomicOr(&let[nl * a.LWs + (p >> 4)], l << (2 * (p & 15)));
      }
      __syncthreads();
      // ---- h | v : x[k,s] = b[k] + sum_j W[k, letter[s+j], j]; h = [sigma(x) > u] ----
      for (uint32_t i = threadIdx.x; i < (uint32_t)(ns * a.Lf); i += blockDim.x) {
        const uint32_t nl = fastdiv(i, a.divLf);
        const int s = (int)(i - nl * (uint32_t)a.Lf);
        const uint64_t win = letter_window(let + (size_t)nl * a.LWs, s, M);
        const uint32_t gn = a.rng.seq_offset + (uint32_t)(n0 + nl);
#pragma unroll
        for (int strand = 0; strand < 2; ++strand) {
          if (strand == 1 && !mv.ds) break;
          float x[KP];
          conv_gather<NQ>(strand ? Tr : Tf, win, mv, x);
          uint32_t mask[NW];
#pragma unroll
          for (int w = 0; w < NW; ++w) mask[w] = 0u;
#pragma unroll
          for (int kg = 0; kg < NQ; ++kg) {
            if (4 * kg < mv.K) {
              const Philox4 r = philox4x32_10(gn, (uint32_t)s, rng_word2(KIND_CHAIN_H, (uint32_t)strand, (uint32_t)kg),
                                              a.rng.step + (uint32_t)st, a.rng.seed_lo, a.rng.seed_hi);
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int k = 4 * kg + t;
                const float p = sigmoidf_fast(x[k]);
                const uint32_t bit = p > u01(r.v[t]) ? 1u : 0u;
                mask[k >> 5] |= bit << (k & 31);
              }
            }
          }
          uint32_t* dst = (strand ? hmp : hm) + (size_t)nl * rowW + (size_t)(s + M - 1) * NW;
#pragma unroll
          for (int w = 0; w < NW; ++w) dst[w] = mask[w];
        }
      }
      __syncthreads();
    }
    // LDS -> chain state
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)(ns * a.Lf * NW); idx += blockDim.x) {
      const uint32_t per = (uint32_t)(a.Lf * NW);
      const uint32_t nl = idx / per;
      const uint32_t r = idx - nl * per;
      const size_t src = (size_t)nl * rowW + (size_t)(M - 1) * NW + r;
      const size_t g = (size_t)(n0 + nl) * per + r;
      a.hm[g] = hm[src];
      if (mv.ds) a.hmp[g] = hmp[src];
    }
    if (a.vout)
      for (int idx = threadIdx.x; idx < ns * a.LWs; idx += blockDim.x)
        a.vout[(size_t)n0 * a.LWs + idx] = let[idx];
  }
}

// ---------------------------------------------------------------------------
// Gradient statistics (convRBM.py:327-371 and the closed form of :440-451):
// raw sums over (n,s) of
//   P[k,s] * onehot[a,s+j]          -> vh   (per strand)
//   P(1-P)[k,s] * onehot[a,s+j]     -> sw   (forward strand, if want_sparsity)
//   P[k,s], P(1-P)[k,s]             -> h, sb
//   onehot[a,p]                     -> v    (letter counts)
// Phase A: one thread per hidden position computes P (and P') and parks it in
// LDS.  Phase B: each wave owns one accumulator tile [4][JC][KC] in registers
// for the whole kernel and streams every parked item through it; per-block
// partial sums are written once at the end (deterministic order).
// ---------------------------------------------------------------------------
struct StatsArgs {
  ModelView mv;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  int32_t TS;
  FastDiv divLh, divL;
  int32_t want_sparsity;
  int32_t ntk, ntj, ntiles;
  int32_t row, off_vh0, off_vh1, off_h0, off_h1, off_sw, off_sb, off_v;
  float* partials;     // [gridDim.x][row], zero-initialised by the host
};

template <int NQ>
__global__ void __launch_bounds__(256) stats_kernel(StatsArgs a) {
  constexpr int KP = 4 * NQ;
  constexpr int NQC = NQ < 4 ? NQ : 4;
  constexpr int KC = 4 * NQC;
  constexpr int JC = NQC <= 3 ? 4 : 3;
  HIP_DYNAMIC_SHARED(float, smem);
  const ModelView& mv = a.mv;
  const int M = mv.M, K = mv.K;
  const int tab = mv.ngroups * mv.rows * KP;
  const int nthr = blockDim.x;
  float* Tf = smem;
  float* Tr = Tf + tab;
  float* Pb0 = Tr + (mv.ds ? tab : 0);               // [nthr][KP]
  float* Pb1 = Pb0 + (size_t)nthr * KP;              // [nthr][KP] (ds)
  uint32_t* Win = reinterpret_cast<uint32_t*>(Pb1 + (mv.ds ? (size_t)nthr * KP : 0));   // [nthr][2]
  float* red = reinterpret_cast<float*>(Win + 2 * nthr);                                 // [64]

  build_gather_table(Tf, mv, KP, false);
  if (mv.ds) build_gather_table(Tr, mv, KP, true);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = nthr >> 6;
  // accumulator tile of this wave: (class, kt, jt)
  const int tile_id = blockIdx.y * nwaves + wave;
  const bool active = tile_id < a.ntiles;
  int t_jt = 0, t_kt = 0, t_kind = 0, t_strand = 0;
  if (active) {
    int t = tile_id;
    t_jt = t % a.ntj; t /= a.ntj;
    t_kt = t % a.ntk; t /= a.ntk;
    // accumulator classes: 0 = vh (forward strand), 1 = vh' (ds only), last = sw (forward, P(1-P))
    t_kind = (a.want_sparsity && t == mv.ds + 1) ? 1 : 0;
    t_strand = (!t_kind && t == 1) ? 1 : 0;
  }
  float acc[4][JC][KC];
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int jj = 0; jj < JC; ++jj)
#pragma unroll
      for (int q = 0; q < KC; ++q) acc[l][jj][q] = 0.f;

  const bool owner = blockIdx.y == 0;   // h / sb / letter counts are accumulated once
  float hs0[KP], hs1[KP], sb[KP];
#pragma unroll
  for (int q = 0; q < KP; ++q) { hs0[q] = 0.f; hs1[q] = 0.f; sb[q] = 0.f; }
  float vc0 = 0.f, vc1 = 0.f, vc2 = 0.f, vc3 = 0.f;

  const int nseqtiles = (a.n + a.TS - 1) / a.TS;
  for (int tile = blockIdx.x; tile < nseqtiles; tile += gridDim.x) {
    const int n0 = tile * a.TS;
    const int ns = min(a.TS, a.n - n0);
    const uint32_t items = (uint32_t)ns * (uint32_t)a.Lh;
    for (uint32_t base = 0; base < items; base += nthr) {
      __syncthreads();   // tables built / previous batch consumed
      const uint32_t i = base + threadIdx.x;
      float* p0 = Pb0 + (size_t)threadIdx.x * KP;
      float* p1 = Pb1 + (size_t)threadIdx.x * KP;
      if (i < items) {
        const uint32_t nl = fastdiv(i, a.divLh);
        const int s = (int)(i - nl * (uint32_t)a.Lh);
        const uint64_t win = letter_window(a.letters + (size_t)(n0 + nl) * a.LW, s, M);
        Win[2 * threadIdx.x] = (uint32_t)win;
        Win[2 * threadIdx.x + 1] = (uint32_t)(win >> 32);
        float x[KP];
        conv_gather<NQ>(Tf, win, mv, x);
#pragma unroll
        for (int q = 0; q < KP; ++q) {
          const float p = sigmoidf_fast(x[q]);
          p0[q] = p;
          if (owner) { hs0[q] += p; sb[q] += p * (1.f - p); }
        }
        if (mv.ds) {
          conv_gather<NQ>(Tr, win, mv, x);
#pragma unroll
          for (int q = 0; q < KP; ++q) {
            const float p = sigmoidf_fast(x[q]);
            p1[q] = p;
            if (owner) hs1[q] += p;
          }
        }
      } else {
        Win[2 * threadIdx.x] = 0u;
        Win[2 * threadIdx.x + 1] = 0u;
#pragma unroll
        for (int q = 0; q < KP; ++q) { p0[q] = 0.f; if (mv.ds) p1[q] = 0.f; }
      }
      __syncthreads();
      if (active) {
        const float* Pb = t_strand ? Pb1 : Pb0;
        for (int c = 0; c < nwaves; ++c) {
          const int it = c * 64 + lane;
          const uint64_t win = (uint64_t)Win[2 * it] | ((uint64_t)Win[2 * it + 1] << 32);
          float pk[KC];
          const float4* src = reinterpret_cast<const float4*>(Pb + (size_t)it * KP + t_kt * KC);
#pragma unroll
          for (int q = 0; q < NQC; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t_kt * NQC + q < NQ) v = src[q];
            pk[4 * q] = v.x; pk[4 * q + 1] = v.y; pk[4 * q + 2] = v.z; pk[4 * q + 3] = v.w;
          }
          if (t_kind) {
#pragma unroll
            for (int q = 0; q < KC; ++q) pk[q] = pk[q] * (1.f - pk[q]);
          }
#pragma unroll
          for (int jj = 0; jj < JC; ++jj) {
            const int j = t_jt * JC + jj;
            if (j < M) {
              const uint32_t l = (uint32_t)(win >> (2 * j)) & 3u;
              const float m0 = l == 0u ? 1.f : 0.f, m1 = l == 1u ? 1.f : 0.f;
              const float m2 = l == 2u ? 1.f : 0.f, m3 = l == 3u ? 1.f : 0.f;
#pragma unroll
              for (int q = 0; q < KC; ++q) {
                acc[0][jj][q] = fmaf(m0, pk[q], acc[0][jj][q]);
                acc[1][jj][q] = fmaf(m1, pk[q], acc[1][jj][q]);
                acc[2][jj][q] = fmaf(m2, pk[q], acc[2][jj][q]);
                acc[3][jj][q] = fmaf(m3, pk[q], acc[3][jj][q]);
              }
            }
          }
        }
      }
    }
    // letter counts of the whole visible rows of this tile
    if (owner) {
      const uint32_t vitems = (uint32_t)ns * (uint32_t)a.L;
      for (uint32_t i = threadIdx.x; i < vitems; i += nthr) {
        const uint32_t nl = fastdiv(i, a.divL);
        const int p = (int)(i - nl * (uint32_t)a.L);
        const uint32_t l = (a.letters[(size_t)(n0 + nl) * a.LW + (p >> 4)] >> (2 * (p & 15))) & 3u;
        vc0 += l == 0u ? 1.f : 0.f; vc1 += l == 1u ? 1.f : 0.f;
        vc2 += l == 2u ? 1.f : 0.f; vc3 += l == 3u ? 1.f : 0.f;
      }
    }
  }

  float* out = a.partials + (size_t)blockIdx.x * a.row;
  // accumulator tiles: wave reduction, lane 0 writes (each slot has exactly one writer)
  if (active) {
    const int off = t_kind ? a.off_sw : (t_strand ? a.off_vh1 : a.off_vh0);
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int jj = 0; jj < JC; ++jj)
#pragma unroll
        for (int q = 0; q < KC; ++q) {
          const float v = wave_sum(acc[l][jj][q]);
          const int k = t_kt * KC + q, j = t_jt * JC + jj;
          if (lane == 0 && k < K && j < M) out[off + (k * 4 + l) * M + j] = v;
        }
  }
  if (owner) {
    // per-thread sums -> wave -> block (through LDS), fixed order
    auto block_sum_store = [&](float v, int dst) {
      v = wave_sum(v);
      __syncthreads();
      if (lane == 0) red[wave] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += red[w];
        out[dst] = t;
      }
    };
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      if (q < K) {
        block_sum_store(hs0[q], a.off_h0 + q);
        if (mv.ds) block_sum_store(hs1[q], a.off_h1 + q);
        if (a.want_sparsity) block_sum_store(sb[q], a.off_sb + q);
      }
    }
    block_sum_store(vc0, a.off_v + 0);
    block_sum_store(vc1, a.off_v + 1);
    block_sum_store(vc2, a.off_v + 2);
    block_sum_store(vc3, a.off_v + 3);
  }
}

#ifdef CRBM_DEFINE_MISC_KERNELS
// sums[dst(r)] = sum over partial rows of column r, fixed order.
struct ReduceArgs {
  const float* partials;
  float* sums;
  int32_t nrows, row;
  int32_t skip_begin, skip_len;   // columns [skip_begin, skip_begin+skip_len) are dropped
  float n_value;                  // written after the last kept column
};

__global__ void reduce_partials_kernel(ReduceArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < a.row) {
    const bool skipped = r >= a.skip_begin && r < a.skip_begin + a.skip_len;
    if (!skipped) {
      float t = 0.f;
      for (int i = 0; i < a.nrows; ++i) t += a.partials[(size_t)i * a.row + r];
      a.sums[r < a.skip_begin ? r : r - a.skip_len] = t;
    }
  }
  if (r == 0) a.sums[a.row - a.skip_len] = a.n_value;
}

// ---------------------------------------------------------------------------
// Normalise the (all-reduced) raw sums and apply the SGD+momentum update
// (convRBM.py:358-371, :415-436, :440-451).
// ---------------------------------------------------------------------------
struct UpdateArgs {
  const float* sums;
  float* W; float* b; float* c;
  float* vW; float* vb; float* vc;
  int32_t K, M, ds;
  int32_t L_data, Lf;
  int32_t data_off, n_d, model_off, n_m;   // offsets into sums
  float lr, momentum, rho, lambda_rate;
};

__global__ void apply_update_kernel(UpdateArgs a) {
  const int K = a.K, M = a.M, KAM = K * 4 * M;
  const float n_d = a.sums[a.n_d], n_m = a.sums[a.n_m];
  const float cnt_d = n_d * (float)(a.L_data - M + 1);
  const float cnt_m = n_m * (float)a.Lf;
  const float* d = a.sums + a.data_off;    // [vh][vh'][h][h'][sw][sb][v]
  const float* m = a.sums + a.model_off;   // [vh][vh'][h][h'][v]
  const float* d_vh = d, *d_vhp = d + KAM, *d_h = d + 2 * KAM, *d_hp = d_h + K;
  const float* d_sw = d + 2 * KAM + 2 * K, *d_sb = d_sw + KAM, *d_v = d_sb + K;
  const float* m_vh = m, *m_vhp = m + KAM, *m_h = m + 2 * KAM, *m_hp = m_h + K, *m_v = m_hp + K;
  const float q = a.rho;
  for (int idx = threadIdx.x; idx < KAM; idx += blockDim.x) {
    const int k = idx / (4 * M), al = (idx / M) & 3, j = idx % M;
    const int ridx = (k * 4 + (3 - al)) * M + (M - 1 - j);
    float gd = d_vh[idx] / cnt_d, gm = m_vh[idx] / cnt_m;
    if (a.ds) {
      gd = 0.5f * (gd + d_vhp[ridx] / cnt_d);
      gm = 0.5f * (gm + m_vhp[ridx] / cnt_m);
    }
    const float p = d_h[k] / cnt_d;
    const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
    const float reg = -g * d_sw[idx] / cnt_d;
    const float v = a.momentum * a.vW[idx] + a.lr * (gd - gm - a.lambda_rate * reg);
    a.vW[idx] = v;
    a.W[idx] += v;
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float gd = d_h[k] / cnt_d, gm = m_h[k] / cnt_m;
    if (a.ds) {
      gd = 0.5f * (gd + d_hp[k] / cnt_d);
      gm = 0.5f * (gm + m_hp[k] / cnt_m);
    }
    const float p = d_h[k] / cnt_d;
    const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
    const float reg = -g * d_sb[k] / cnt_d;
    const float v = a.momentum * a.vb[k] + a.lr * (gd - gm - a.lambda_rate * reg);
    a.vb[k] = v;
    a.b[k] += v;
  }
  if (threadIdx.x < 4) {
    const int al = threadIdx.x;
    const float nd = n_d * (float)a.L_data, nm = n_m * (float)(a.Lf + M - 1);
    const float gd = d_v[al] / nd + d_v[3 - al] / nd;     // a += a[::-1]  (:345)
    const float gm = m_v[al] / nm + m_v[3 - al] / nm;
    const float v = a.momentum * a.vc[al] + a.lr * (gd - gm);
    a.vc[al] = v;
    a.c[al] += v;
  }
}

#endif  // CRBM_DEFINE_MISC_KERNELS

// ---------------------------------------------------------------------------
// Free energy (convRBM.py:657-697): one wave per sequence.
//   fe[n]    = ( -sum_{k,s} softplus(x) [- rc strand] - sum_p c[letter_p] ) / L
//   fem[n,k] =   -sum_s softplus(x[k]) [- rc strand] - sum_p c[letter_p]
// ---------------------------------------------------------------------------
struct FeArgs {
  ModelView mv;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  float* fe;
  float* fem;
};

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

template <int NQ>
__global__ void __launch_bounds__(256) free_energy_kernel(FeArgs a) {
  constexpr int KP = 4 * NQ;
  HIP_DYNAMIC_SHARED(float, smem);
  const ModelView& mv = a.mv;
  const int tab = mv.ngroups * mv.rows * KP;
  float* Tf = smem;
  float* Tr = Tf + tab;
  build_gather_table(Tf, mv, KP, false);
  if (mv.ds) build_gather_table(Tr, mv, KP, true);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const float c0 = mv.c[0], c1 = mv.c[1], c2 = mv.c[2], c3 = mv.c[3];
  for (int nn = blockIdx.x * nwaves + wave; nn < a.n; nn += gridDim.x * nwaves) {
    const uint32_t* row = a.letters + (size_t)nn * a.LW;
    float acc[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) acc[q] = 0.f;
    for (int s = lane; s < a.Lh; s += 64) {
      const uint64_t win = letter_window(row, s, mv.M);
      float x[KP];
      conv_gather<NQ>(Tf, win, mv, x);
#pragma unroll
      for (int q = 0; q < KP; ++q) acc[q] += softplusf(x[q]);
      if (mv.ds) {
        conv_gather<NQ>(Tr, win, mv, x);
#pragma unroll
        for (int q = 0; q < KP; ++q) acc[q] += softplusf(x[q]);
      }
    }
    float cs = 0.f;
    for (int p = lane; p < a.L; p += 64) {
      const uint32_t l = (row[p >> 4] >> (2 * (p & 15))) & 3u;
      cs += l == 0u ? c0 : l == 1u ? c1 : l == 2u ? c2 : c3;
    }
    cs = wave_sum(cs);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      const float v = wave_sum(acc[q]);
      if (q < mv.K) {
        tot += v;
        if (lane == 0 && a.fem) a.fem[(size_t)nn * mv.K + q] = -v - cs;
      }
    }
    if (lane == 0 && a.fe) a.fe[nn] = (-tot - cs) / (float)a.L;
  }
}

// ---------------------------------------------------------------------------
// Dispatch table: one entry per instantiated NQ.
// ---------------------------------------------------------------------------
struct LaunchCfg {
  uint32_t gx, gy, block, lds;
  hipStream_t stream;
};
struct KernelTable {
  int nq;
  void (*hgv)(const HgvArgs&, const LaunchCfg&);
  void (*gibbs)(const GibbsArgs&, const LaunchCfg&);
  void (*stats)(const StatsArgs&, const LaunchCfg&);
  void (*free_energy)(const FeArgs&, const LaunchCfg&);
};
const KernelTable* kernel_table(int nq);

}  // namespace crbm
