/*
 * crbm_cpu.c -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * A plain C, float32, OpenMP restatement ("port") of one Gibbs step of the
 * reference's persistent chain (schulter/crbm secomo/convRBM.py:397-408) in
 * the reference's own dense tensor layout, the way its Theano graph evaluates
 * it: a full transposed convolution (convRBM.py:277-292), a 4-way softmax
 * (:699-702) and categorical sample (:301-315), then a valid cross-correlation
 * (:238-243), sigmoid (:245-257) and Bernoulli sample (:259-267), for both
 * strands when doublestranded.  No sparsity or one-hot shortcuts: dense MACs.
 *
 * It is validated against oracle/crbm_oracle.py in tests/test_cpu_port.py and
 * timed by bench.py's `cpu_baseline` leg ("kind": "port").  The reference
 * itself cannot run here (Theano absent), so this is a restatement, not Theano.
 * Nothing in crbm_amd/ links or calls it.
 *
 * Randomness: the same Philox-4x32-7 counters as the oracle and the kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
  for (int r = 0; r < 7; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

/* 12-bit field i (0..9) of the 128-bit little-endian Philox output */
static inline uint32_t field12(const uint32_t r[4], int i) {
  const int w = (12 * i) / 32, b = (12 * i) % 32;
  uint32_t v = r[w] >> b;
  if (b > 20) v |= r[(w + 1) & 3] << (32 - b);
  return v & 0xFFFu;
}

int crbm_cpu_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* One Gibbs step on n chains.
 *   W (K,4,M), b (K), c (4)
 *   h, hp: (n,K,Lh) in/out, 0/1 floats (hp ignored unless ds)
 *   v: (n,4,L) out, one-hot floats, L = Lh + M - 1
 *   ph, php: (n,K,Lh) out, hidden probabilities of this step (may be NULL)
 * Philox counters: see oracle/crbm_oracle.py (kind 1 = hidden, 2 = visible). */
void crbm_cpu_gibbs_step(const float* W, const float* b, const float* c, int K, int M, int ds,
                         float* h, float* hp, float* v, float* ph, float* php, int n, int Lh,
                         uint64_t seed, uint32_t step, uint32_t seq_offset, int nthreads) {
  const int L = Lh + M - 1;
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    /* per-thread scratch, allocated once per thread and kept between calls: y (4,L) top-down activations, x (K,Lh)
     * bottom-up activations of one strand */
    static __thread float* scratch = NULL;
    static __thread size_t scratch_cap = 0;
    const size_t need = (size_t)4 * L + (size_t)K * Lh;
    if (scratch_cap < need) {
      free(scratch);
      scratch = (float*)malloc(sizeof(float) * need);
      scratch_cap = need;
    }
    float* y = scratch;
    float* x = scratch + (size_t)4 * L;
    /* static schedule: a chain is always worked on by the same thread, which is also the first to touch its rows of
     * h, h', v (the caller hands freshly allocated, untouched arrays to the first step) */
#pragma omp for schedule(static)
    for (int nn = 0; nn < n; ++nn) {
      float* hn = h + (size_t)nn * K * Lh;
      float* hpn = ds ? hp + (size_t)nn * K * Lh : NULL;
      float* vn = v + (size_t)nn * 4 * L;
      /* top-down: y[a][p] = c[a] + sum_{k,j} W[k,a,j] h[k,p-j] + rc(W)[k,a,j] h'[k,p-j] */
      for (int a = 0; a < 4; ++a)
        for (int p = 0; p < L; ++p) y[a * L + p] = c[a];
      for (int k = 0; k < K; ++k)
        for (int a = 0; a < 4; ++a)
          for (int j = 0; j < M; ++j) {
            const float w = W[(k * 4 + a) * M + j];
            const float wr = W[(k * 4 + (3 - a)) * M + (M - 1 - j)];
            float* yr = y + a * L + j;
            const float* hr = hn + (size_t)k * Lh;
            for (int s = 0; s < Lh; ++s) yr[s] += w * hr[s];
            if (ds) {
              const float* hpr = hpn + (size_t)k * Lh;
              for (int s = 0; s < Lh; ++s) yr[s] += wr * hpr[s];
            }
          }
      /* softmax over the 4 letters + categorical sample: one Philox call serves 4 positions */
      for (int p0 = 0; p0 < L; p0 += 4) {
        uint32_t r[4];
        philox4x32_7(seq_offset + (uint32_t)nn, (uint32_t)(p0 >> 2), 2u << 28, step, k0, k1, r);
        for (int p = p0; p < p0 + 4 && p < L; ++p) {
          const float y0 = y[p], y1 = y[L + p], y2 = y[2 * L + p], y3 = y[3 * L + p];
          const float mx = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
          const float e0 = expf(y0 - mx), e1 = expf(y1 - mx), e2 = expf(y2 - mx), e3 = expf(y3 - mx);
          const float sum = (e0 + e1) + (e2 + e3);
          const float t = u01(r[p & 3]) * sum;
          const int l = (t >= e0) + (t >= e0 + e1) + (t >= (e0 + e1) + e2);
          vn[p] = l == 0; vn[L + p] = l == 1; vn[2 * L + p] = l == 2; vn[3 * L + p] = l == 3;
        }
      }
      /* bottom-up, both strands: x[k][s] = b[k] + sum_{a,j} Wf[k,a,j] v[a,s+j] */
      for (int strand = 0; strand <= ds; ++strand) {
        float* out = strand ? hpn : hn;
        float* pout = strand ? (php ? php + (size_t)nn * K * Lh : NULL) : (ph ? ph + (size_t)nn * K * Lh : NULL);
        for (int k = 0; k < K; ++k) {
          float* xk = x + (size_t)k * Lh;
          for (int s = 0; s < Lh; ++s) xk[s] = b[k];
          for (int a = 0; a < 4; ++a)
            for (int j = 0; j < M; ++j) {
              const float w = strand ? W[(k * 4 + (3 - a)) * M + (M - 1 - j)] : W[(k * 4 + a) * M + j];
              const float* vr = vn + a * L + j;
              for (int s = 0; s < Lh; ++s) xk[s] += w * vr[s];
            }
        }
        /* 24-bit uniform of unit k: coarse and fine 12-bit fields of two Philox calls shared by the 10 units of
         * group k/10 -- drawn once per (position, group) */
        for (int s = 0; s < Lh; ++s)
          for (int g = 0; 10 * g < K; ++g) {
            uint32_t rc[4], rf[4];
            const uint32_t w2 = (1u << 28) | ((uint32_t)strand << 24) | (uint32_t)g;
            philox4x32_7(seq_offset + (uint32_t)nn, (uint32_t)s, w2, step, k0, k1, rc);
            philox4x32_7(seq_offset + (uint32_t)nn, (uint32_t)s, w2 | (1u << 16), step, k0, k1, rf);
            for (int i = 0; i < 10 && 10 * g + i < K; ++i) {
              const int k = 10 * g + i;
              const float p = 1.0f / (1.0f + expf(-x[(size_t)k * Lh + s]));
              const float u = (float)(field12(rc, i) * 4096u + field12(rf, i)) * 5.9604644775390625e-8f;
              out[(size_t)k * Lh + s] = p > u ? 1.0f : 0.0f;
              if (pout) pout[(size_t)k * Lh + s] = p;
            }
          }
      }
    }
  }
}
