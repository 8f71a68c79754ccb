// CDNA4 (gfx950) kernels of the CRBM hot path.  Wave64 throughout.
//
// The model-dependent kernels are templates over crbm::Cfg<K,M,DS,G> and are
// compiled per model by hiprtc when a handle is created (crbm_jit.h) -- the
// counterpart of the reference compiling its Theano graph at construction
// (convRBM.py:175, :453-515).  K, M, the strand count and the table grouping
// are therefore compile-time constants: every motif loop is unrolled into
// registers and every LDS table offset is an instruction immediate.
//
// Design (details in DESIGN.md):
//  * The visible layer is one-hot wherever the forward correlation is applied
//    (reference sequences.py:28-31, convRBM.py:304-310), so a sequence is kept
//    as 2-bit letters and x[k,s] = b[k] + sum_j W[k, letter[s+j], j] becomes a
//    gather-add.  Letters are taken G at a time: LDS holds pre-summed rows
//    T[g][letter-tuple][k], one ds_read_b128 feeds four motifs.
//  * The hidden layer is binary wherever the transposed convolution is applied
//    (convRBM.py:259-267), so chain state is a K-bit mask per hidden position.
//    For small K*M the top-down pass is a dense walk over pre-summed tables
//    indexed by 5-bit mask chunks; otherwise it adds W[k,:,j] per set bit.
//  * One workgroup owns whole chains and one thread owns 4 consecutive
//    positions, so a k-step Gibbs chain runs entirely in LDS/registers; HBM
//    sees the masks once in and once out per launch.
//  * The convolution and its transpose do not use MFMA: the transposed conv has
//    output width 4 and the forward has one-hot operands (BASELINE.json
//    north_star).  The gradient statistics -- a K x 4M output contracted over
//    all positions of the batch -- do (stats_mfma_*).
//  * The reverse-complement strand needs no table of its own: its activation is
//    the forward gather of the reverse-complemented letter window.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "crbm_layout.h"

namespace crbm {

template <int I>
struct IC {
  static constexpr int value = I;
};

// ---------------------------------------------------------------------------
// Philox-4x32-7 (Random123 constants; seven rounds is the fewest that its authors report as passing BigCrush --
// Salmon et al., SC'11, table 2 -- and what Random123 ships as philox4x32_R<7>); same counters and the same
// round count as oracle/crbm_oracle.py and its C port.  The stream is this library's own definition: the
// reference seeds Theano's MRG31k3p from the wall clock (convRBM.py:155), so there is no stream to match.
// ---------------------------------------------------------------------------
struct Philox4 {
  uint32_t v[4];
};

// ---------------------------------------------------------------------------
// The one place that depends on the compiler: hipcc / hiprtc (clang) get the
// gfx950 instructions, the CPU emulation build of the same header (g++,
// tests/emu) gets portable equivalents supplied by its shim.
// ---------------------------------------------------------------------------
struct HalfFrag {     // eight f16 = one A or B fragment of v_mfma_f32_16x16x32_f16
  uint32_t r[4];
};
#if defined(__clang__)
typedef float floatx4 __attribute__((ext_vector_type(4)));
// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96); the compiler does not fuse it itself
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t bitrev32(uint32_t x) { return __builtin_bitreverse32(x); }   // v_bfrev_b32
// 16-byte store that does not linger in the caches (the chain state is written once
// per launch and next read by another launch: measured 0.45 us per launch at config #2)
__device__ __forceinline__ void store4_streaming(uint32_t* dst, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  const u4v val = {x, y, z, w};
  __builtin_nontemporal_store(val, reinterpret_cast<u4v*>(dst));
}
__device__ __forceinline__ void store1_streaming(uint32_t* dst, uint32_t x) { __builtin_nontemporal_store(x, dst); }
// all vector-memory operations of this wave have completed (gfx9 encoding: vmcnt = 0, expcnt and lgkmcnt untouched)
__device__ __forceinline__ void wait_vector_memory() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// System-scope accesses for memory another process / another GPU reads and writes (the mapped sums buffers of the
// IPC all-reduce): loads and stores that bypass this GPU's caches, and the fence that orders them.
__device__ __forceinline__ float load_system(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint32_t load_system(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_system(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_system(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void fence_system() { __threadfence_system(); }
// what a consumer needs after it has seen a producer's flag: later loads may not be served from lines cached before
__device__ __forceinline__ void fence_acquire_system() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }
// arrival counter of the blocks of one launch: the increment publishes the block's stores (release) and the block
// that draws the last ticket sees everybody's (acquire), at the scope of this GPU
__device__ __forceinline__ uint32_t ticket_add(uint32_t* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void short_sleep() { __builtin_amdgcn_s_sleep(8); }
// the GPU's constant-rate wall clock (s_memrealtime; hipDeviceAttributeWallClockRate ticks per millisecond): bounds
// the waits for another process in TIME, whatever the shader clock does
__device__ __forceinline__ uint64_t realtime_ticks() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ uint64_t shader_cycles() { return __builtin_amdgcn_s_memtime(); }   // counts shader clocks
// x = hi + lo with both halves f16 (round to nearest): 22 significant bits, v_cvt_pk_f16_f32 +
// v_cvt_f32_f16 + v_pk_add_f32 per pair
__device__ __forceinline__ void split_f16(const float (&x)[8], HalfFrag& hi, HalfFrag& lo) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (_Float16)x[e];
    l[e] = (_Float16)(x[e] - (float)h[e]);
  }
  hi = __builtin_bit_cast(HalfFrag, h);
  lo = __builtin_bit_cast(HalfFrag, l);
}
// D = A (16 x 32) * B (32 x 16) + C on the matrix core.  Lane l holds A[row l&15][8(l>>4) .. +7],
// B[8(l>>4) .. +7][col l&15] and D[rows 4(l>>4) .. +3][col l&15] (checked with integer data on gfx950:
// tools/mfma_probe.hip)
__device__ __forceinline__ floatx4 mfma_16x16x32_f16(const HalfFrag& a, const HalfFrag& b, floatx4 c) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
// two floats = one v_pk_*_f32 operand; a * b + c as one v_pk_fma_f32; `opaque` hides where a value came
// from (no instruction), so that float(field) + 1 stays a packed add instead of a second conversion
typedef float floatx2 __attribute__((vector_size(8)));
__device__ __forceinline__ floatx2 fma2(floatx2 a, floatx2 b, floatx2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ floatx2 opaque(floatx2 v) { asm volatile("" : "+v"(v)); return v; }
#else   // CPU emulation build (tests/emu/shim/hip/hip_runtime.h implements the wave-wide parts)
typedef float floatx4 __attribute__((vector_size(16)));
typedef float floatx2 __attribute__((vector_size(8)));
__device__ __forceinline__ floatx2 fma2(floatx2 a, floatx2 b, floatx2 c) { return floatx2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
__device__ __forceinline__ floatx2 opaque(floatx2 v) { return v; }
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
__device__ __forceinline__ uint32_t bitrev32(uint32_t x) {
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
}
__device__ __forceinline__ void store4_streaming(uint32_t* dst, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  *reinterpret_cast<uint4*>(dst) = make_uint4(x, y, z, w);
}
__device__ __forceinline__ void store1_streaming(uint32_t* dst, uint32_t x) { *dst = x; }
__device__ __forceinline__ void wait_vector_memory() {}
__device__ __forceinline__ float load_system(const float* p) { float v; __atomic_load(p, &v, __ATOMIC_RELAXED); return v; }
__device__ __forceinline__ uint32_t load_system(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
__device__ __forceinline__ void store_system(float* p, float v) { __atomic_store(p, &v, __ATOMIC_RELAXED); }
__device__ __forceinline__ void store_system(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELAXED); }
__device__ __forceinline__ void fence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
__device__ __forceinline__ void fence_acquire_system() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
__device__ __forceinline__ uint32_t ticket_add(uint32_t* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_ACQ_REL); }
__device__ __forceinline__ void short_sleep() {}
__device__ __forceinline__ uint64_t realtime_ticks() { return emu::realtime_ticks(); }   // microseconds
__device__ __forceinline__ uint64_t shader_cycles() { return emu::realtime_ticks(); }
__device__ __forceinline__ void split_f16(const float (&x)[8], HalfFrag& hi, HalfFrag& lo) { emu::split_f16(x, hi.r, lo.r); }
__device__ __forceinline__ floatx4 mfma_16x16x32_f16(const HalfFrag& a, const HalfFrag& b, floatx4 c) {
  float d[4] = {c[0], c[1], c[2], c[3]};
  emu::mfma_16x16x32_f16(a.r, b.r, d);
  return floatx4{d[0], d[1], d[2], d[3]};
}
#endif

#ifndef CRBM_PHILOX_ROUNDS
#define CRBM_PHILOX_ROUNDS 7      // anything else (CRBM_JIT_DEFINES) is a timing experiment: the oracle draws seven
#endif
__device__ __forceinline__ Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < CRBM_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
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

// 12-bit field I (0..9) of the 128-bit little-endian value v0 | v1<<32 | ...
template <int I>
__device__ __forceinline__ uint32_t philox_field12(const Philox4& r) {
  constexpr int w = (12 * I) / 32, b = (12 * I) % 32;
  if (b <= 20) return (r.v[w] >> b) & 0xFFFu;
  return ((r.v[w] >> b) | (r.v[(w + 1) & 3] << ((32 - b) & 31))) & 0xFFFu;
}

// the same for a field index known only at run time (the rare exact path of the chain kernel)
__device__ __forceinline__ uint32_t philox_field12_dyn(const Philox4& r, int i) {
  const int b = 12 * i, w = b >> 5, sh = b & 31;
  uint32_t v = philox_pick(r, w) >> sh;
  if (sh > 20) v |= philox_pick(r, (w + 1) & 3) << (32 - sh);
  return v & 0xFFFu;
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }

// counter word 2: kind | strand | sub-stream (0 coarse, 1 fine) | group
__device__ __forceinline__ uint32_t rng_word2(uint32_t kind, uint32_t strand, uint32_t sub, uint32_t group) {
  return (kind << 28) | (strand << 24) | (sub << 16) | group;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t i, const FastDiv& f) {
  if (f.d <= 1) return i;
  uint32_t q = __umulhi(i, f.inv);
  if (f.fix && q * f.d > i) --q;   // long rows only (crbm_layout.h): the estimate is q or q + 1
  return q;
}
// The Gibbs kernel divides indices of one LDS-resident tile: dividend and divisor are both at most
// the tile's word count (<= 40 960: 160 KB), so n*d < 2^32 and the single mul_hi is always exact.
__device__ __forceinline__ uint32_t fastdiv_tile(uint32_t i, const FastDiv& f) {
  return f.d <= 1 ? i : __umulhi(i, f.inv);
}

// The MFMA statistics carry probabilities scaled by 2^14 (and a one-hot operand of 2^-14):
// the f16 halves of P then stay normal numbers down to P ~ 4e-9 instead of 6e-5.
constexpr float STATS_PSCALE_INV = 6.103515625e-05f;   // 2^-14
// z = -x*log2(e) (what conv_gather returns): exp(-x) = 2^z
__device__ __forceinline__ float exp_neg_x(float z) { return __builtin_amdgcn_exp2f(z); }
// v_rcp_f32 (1 ulp); __fdividef expands to a full division sequence under hiprtc
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_z(float z) { return fast_rcp(1.0f + exp_neg_x(z)); }
__device__ __forceinline__ float x_of_z(float z) { return -0.6931471805599453f * z; }

// sum over the 16 lanes that share lane & 3 (all lanes of the class get the total):
// two row rotations (DPP, VALU only) + two cross-row exchanges
__device__ __forceinline__ float class_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// Wave-wide reductions on the VALU (DPP): four row rotations give every lane its
// 16-lane row total, row_bcast15 / row_bcast31 fold the four rows into lane 63,
// one v_readlane hands the result to all lanes.  (The LDS-crossbar shuffles are
// an order of magnitude slower when dozens of values are reduced back to back.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0x121, 0xf>(v);   // row_ror:1
  v += dpp_move<0x122, 0xf>(v);   // row_ror:2
  v += dpp_move<0x124, 0xf>(v);   // row_ror:4
  v += dpp_move<0x128, 0xf>(v);   // row_ror:8
  v += dpp_move<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// maximum of non-negative values (rows a DPP move does not reach contribute 0)
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = fmaxf(v, dpp_move<0x121, 0xf>(v));
  v = fmaxf(v, dpp_move<0x122, 0xf>(v));
  v = fmaxf(v, dpp_move<0x124, 0xf>(v));
  v = fmaxf(v, dpp_move<0x128, 0xf>(v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------------------
// Sampling K hidden units from z = -x*log2(e) of their activations x
// (convRBM.py:259-267: h = 1 if p > u).  The uniform of unit k is 24 bits wide, u = (coarse*4096 + fine) /
// 2^24, where coarse and fine are the (k % 10)-th 12-bit fields of two Philox
// calls shared by the 10 units of group k / 10.  Almost every decision is
// settled by the coarse field alone: with e = exp(-x),
//     (coarse + 1)(1 + e) <= 4096  =>  h = 1,      coarse (1 + e) > 4096  =>  h = 0,
// and the fine call is made only when some lane of the wave lands in between
// (2^-12 per unit).  Returns the K-bit mask; optionally the probabilities.
// ---------------------------------------------------------------------------
// WANT_P: 0 none, 1 p = sigma(x), 2 p = 2^14 * sigma(x) (the scale of the MFMA statistics, below)
//
// The two coarse tests are taken as sign bits: d1 = 4096 - coarse (1 + e) and d2 = 4096 - (coarse + 1)(1 + e)
// (one fused multiply-add each, two units per packed instruction); the sign of d2 is "not certainly 1",
// the sign of d1 "certainly 0".  Both are shifted into a word per group (v_alignbit_b32, units in
// descending order so that unit 0 ends in bit 0): two instructions per unit where compares, selects
// and ors took six.  e = inf (x < -88) gives d2 = -inf (h = 0, correct); d1 can then be NaN for
// coarse = 0, whose sign only decides whether the exact path below is taken -- it returns 0 as well.
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t acc, float d) { return (acc << 1) | (__float_as_uint(d) >> 31); }

// DEFER: the undecided units are not resolved here; their bits come back in pending[group] (mask has 0
// there) and the caller resolves them later (gibbs_body queues them per block).
template <class C, int WANT_P, bool DEFER = false>
__device__ __forceinline__ void sample_hidden(const float (&z)[C::KP], uint32_t n, uint32_t s, uint32_t kind,
                                              uint32_t strand, const RngView& rng, uint32_t step,
                                              uint32_t (&mask)[C::NW], float (&p)[C::KP],
                                              uint32_t* pending = nullptr, uint32_t group0 = 0) {   // group0: sampler group of unit 0 (a slab of a larger model: hgv_masks_body)
#pragma unroll
  for (int w = 0; w < C::NW; ++w) mask[w] = 0u;
#pragma unroll
  for (int g = 0; g < C::NGRP; ++g) {
    const Philox4 rc = philox4x32(n, s, rng_word2(kind, strand, 0, group0 + (uint32_t)g), step, rng.seed_lo, rng.seed_hi);
    constexpr int kFull = 10;
    const int cnt = C::K - 10 * g < kFull ? C::K - 10 * g : kFull;   // units of this group
    uint32_t not_one = 0u, zero = 0u;     // bit i: unit 10g+i is not certainly 1 / is certainly 0
    auto pair = [&](auto I) {             // units i + 1 and i of the group (i even), in that order
      constexpr int i = decltype(I)::value;
      const int k = 10 * g + i;
      if (k < C::K) {
        const bool two = k + 1 < C::K;
        const floatx2 e = {exp_neg_x(z[k]), exp_neg_x(z[two ? k + 1 : k])};
        const floatx2 ope = e + 1.0f;
        const floatx2 af = opaque(floatx2{(float)philox_field12<i>(rc), (float)philox_field12<i + 1>(rc)});
        const floatx2 c4096 = {4096.0f, 4096.0f};
        const floatx2 d1 = fma2(-af, ope, c4096);             // >= 0: coarse (1 + e) <= 4096
        const floatx2 d2 = fma2(-(af + 1.0f), ope, c4096);    // >= 0: (coarse + 1)(1 + e) <= 4096, h = 1
        if (two) {
          zero = shift_in_sign(zero, d1[1]);
          not_one = shift_in_sign(not_one, d2[1]);
        }
        zero = shift_in_sign(zero, d1[0]);
        not_one = shift_in_sign(not_one, d2[0]);
        if (WANT_P == 1) { p[k] = fast_rcp(ope[0]); if (two) p[k + 1] = fast_rcp(ope[1]); }
        if (WANT_P == 2) {
          const floatx2 os = ope * STATS_PSCALE_INV;
          p[k] = fast_rcp(os[0]);
          if (two) p[k + 1] = fast_rcp(os[1]);
        }
      }
    };
    pair(IC<8>{}); pair(IC<6>{}); pair(IC<4>{}); pair(IC<2>{}); pair(IC<0>{});
    uint32_t ones = ~not_one & ((1u << cnt) - 1u);
    const uint32_t amb = not_one & ~zero;   // units of this group that need the fine field
    if constexpr (DEFER) pending[g] = amb;
    if (!DEFER && __any(amb != 0u)) {
      const Philox4 rf = philox4x32(n, s, rng_word2(kind, strand, 1, group0 + (uint32_t)g), step, rng.seed_lo, rng.seed_hi);
      auto fix = [&](auto I) {
        constexpr int i = decltype(I)::value;
        const int k = 10 * g + i;
        if (k < C::K) {
          if (amb & (1u << i)) {
            // P*4096 - coarse lies in [0,1) up to rounding; compare with fine/4096
            const float t = 4096.0f * fast_rcp(1.0f + exp_neg_x(z[k]));
            const float frac = t - (float)philox_field12<i>(rc);
            ones |= (frac * 4096.0f > (float)philox_field12<i>(rf) ? 1u : 0u) << i;
          }
        }
      };
      fix(IC<0>{}); fix(IC<1>{}); fix(IC<2>{}); fix(IC<3>{}); fix(IC<4>{});
      fix(IC<5>{}); fix(IC<6>{}); fix(IC<7>{}); fix(IC<8>{}); fix(IC<9>{});
    }
    constexpr int kBitsPerWord = 32;
    const int w0 = (10 * g) / kBitsPerWord, sh = (10 * g) % kBitsPerWord;
    mask[w0] |= ones << sh;
    if (sh + cnt > kBitsPerWord) mask[w0 + 1] |= ones >> (kBitsPerWord - sh);
  }
}

// M letters (2 bits each) starting at position s of a packed row: one 64-bit word for M <= 32,
// a second one (letters 32..) for M <= 64 (MAX_MOTIF_LENGTH).
template <int M>
struct LetterWin {
  uint64_t lo, hi;   // hi is never touched for M <= 32
};

template <int M>
__device__ __forceinline__ LetterWin<M> letter_window(const uint32_t* w, int s) {
  static_assert(M <= MAX_MOTIF_LENGTH, "letter windows hold at most 64 letters");
  const int i = s >> 4;
  const int sh = (s & 15) * 2;
  LetterWin<M> out;
  out.hi = 0ull;
  const uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
  if constexpr (M <= 32) {
    uint64_t win = lo >> sh;
    if (2 * M + 30 > 64) {
      if (sh > 0) win |= (uint64_t)w[i + 2] << (64 - sh);
    }
    if (M < 32) win &= (1ull << (2 * M)) - 1ull;
    out.lo = win;
  } else {
    // bits [sh, sh + 2M) of the five words from w[i] on (the row's two pad words cover the last window: crbm_layout.h)
    const uint64_t mid = (uint64_t)w[i + 2] | ((uint64_t)w[i + 3] << 32);
    const uint64_t top = (uint64_t)w[i + 4];
    out.lo = sh > 0 ? (lo >> sh) | (mid << (64 - sh)) : lo;
    out.hi = sh > 0 ? (mid >> sh) | (top << (64 - sh)) : mid;
    if (M < 64) out.hi &= (1ull << (2 * (M - 32))) - 1ull;
  }
  return out;
}

// bits [off, off + 8) of a window (off < 2 M; a group's letter tuple is at most 8 bits wide)
template <int M>
__device__ __forceinline__ uint32_t window_bits(const LetterWin<M>& win, int off) {
  if constexpr (M <= 32) return (uint32_t)(win.lo >> off);
  else {
    if (off >= 64) return (uint32_t)(win.hi >> (off - 64));
    return (uint32_t)(off > 0 ? (win.lo >> off) | (win.hi << (64 - off)) : win.lo);
  }
}

// The window of the reverse-complement strand: letter o of the result is the complement of letter M-1-o.
// The reference flips the FILTER for that strand (rc(W)[k,a,j] = W[k,3-a,M-1-j], convRBM.py:241); flipping the
// data instead gives the same activation -- sum_j W[k, 3 - l(s+j), M-1-j] = sum_j' W[k, rc_window[j'], j'] -- as
// the FORWARD gather of the one table Tf (bias and pad columns of group 0 included exactly once, as they
// should be: both strands share the bias, :242).  No second gather table exists anywhere.
__device__ __forceinline__ uint32_t revcomp_word(uint32_t x) {   // 16 letters: order reversed, each complemented
  x = bitrev32(x);                                                // reverses the two bits inside every letter as well ...
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);        // ... swap them back
  return ~x;
}

template <int M>
__device__ __forceinline__ LetterWin<M> revcomp_window(const LetterWin<M>& w) {
  LetterWin<M> out;
  out.hi = 0ull;
  // reversing a 64-bit word of 32 letters: both halves reversed, then swapped
  const uint64_t rlo = ((uint64_t)revcomp_word((uint32_t)w.lo) << 32) | revcomp_word((uint32_t)(w.lo >> 32));
  if constexpr (M <= 32) {
    out.lo = M < 32 ? rlo >> (64 - 2 * M) : rlo;        // the M letters sit at the top (the complemented padding below them leaves)
  } else {
    const uint64_t rhi = ((uint64_t)revcomp_word((uint32_t)w.hi) << 32) | revcomp_word((uint32_t)(w.hi >> 32));
    constexpr int sh = 128 - 2 * M;                     // the 128-bit value rlo:rhi (rlo on top), shifted down
    out.lo = sh > 0 ? (rhi >> sh) | (rlo << (64 - sh)) : rhi;
    out.hi = sh > 0 ? rlo >> sh : rlo;
  }
  return out;
}

// z[k] (+)= sum over letter groups of T[g][tuple][k]  (z = -log2(e) * activation).  The sums are kept
// as pairs of floats: one v_pk_add_f32 per two motifs (the compiler leaves scalar adds otherwise).
template <class C, bool ACCUMULATE = false>
__device__ __forceinline__ void conv_gather(const float* T, const LetterWin<C::M>& win, float (&x)[C::KP]) {
  floatx2 acc[2 * C::NQ];
  if (ACCUMULATE) {
#pragma unroll
    for (int q = 0; q < 2 * C::NQ; ++q) acc[q] = floatx2{x[2 * q], x[2 * q + 1]};
  }
  auto group = [&](int g, auto FIRST) {
    constexpr bool first = decltype(FIRST)::value != 0;   // the first group of a fresh sum assigns (saves KP adds of 0)
    const uint32_t r = window_bits<C::M>(win, 2 * C::G * g) & (uint32_t)(C::ROWS - 1);
    const float4* row = reinterpret_cast<const float4*>(T + (size_t)g * C::ROWS * C::KP) + (size_t)r * C::NQ;
#pragma unroll
    for (int q = 0; q < C::NQ; ++q) {
      const float4 t = row[q];
      if (first) {
        acc[2 * q] = floatx2{t.x, t.y}; acc[2 * q + 1] = floatx2{t.z, t.w};
      } else {
        acc[2 * q] += floatx2{t.x, t.y}; acc[2 * q + 1] += floatx2{t.z, t.w};
      }
    }
  };
  if (ACCUMULATE) group(0, IC<0>{}); else group(0, IC<1>{});
  if constexpr (C::NG * C::NQ <= 48) {
#pragma unroll
    for (int g = 1; g < C::NG; ++g) group(g, IC<0>{});
  } else {
    // large models: a rolled loop keeps the in-flight LDS reads (and registers) bounded
#pragma unroll 1
    for (int g = 1; g < C::NG; ++g) group(g, IC<0>{});
  }
#pragma unroll
  for (int q = 0; q < 2 * C::NQ; ++q) { x[2 * q] = acc[q][0]; x[2 * q + 1] = acc[q][1]; }
}

// ---------------------------------------------------------------------------
// Pooling (convRBM.py:245-267; Cfg::POOL > 1): the hidden units of POOL consecutive
// positions compete, P_i = exp(x_i) / (POOL + sum_j exp(x_j)), and at most one of them
// fires: unit i is on iff cum_{i-1} <= u < cum_i for the one uniform of the group
// (that of its first position).  A lane owns one position and evaluates its whole
// group itself (POOL gathers, two passes: maximum, then sums) -- plain and without
// cross-lane traffic; the reference calls pooling "not relevant for cRBM" and every
// BASELINE configuration uses POOL = 1, whose code never sees any of this.
//   zfun(pos, z): z[k] = -log2(e) * activation of the units at hidden position pos
// Results for the lane's position s: p = P_s, cb = sum of P over the group's earlier
// positions, S = sum of P over the group (all in the overflow-free form).
// ---------------------------------------------------------------------------
template <int POOL, int KP, class ZFun>
__device__ __forceinline__ void pooled_probs(ZFun zfun, int s, float (&p)[KP], float (&cb)[KP], float (&S)[KP]) {
  const int g0 = s - s % POOL, me = s - g0;
  float zmin[KP], z[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) zmin[k] = 0.f;               // the "POOL" term is POOL * exp(0)
#pragma unroll 1
  for (int j = 0; j < POOL; ++j) {
    zfun(g0 + j, z);
#pragma unroll
    for (int k = 0; k < KP; ++k) zmin[k] = fminf(zmin[k], z[k]);
  }
  float den[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) { den[k] = (float)POOL * __builtin_amdgcn_exp2f(zmin[k]); cb[k] = 0.f; p[k] = 0.f; }
#pragma unroll 1
  for (int j = 0; j < POOL; ++j) {
    zfun(g0 + j, z);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const float e = __builtin_amdgcn_exp2f(zmin[k] - z[k]);   // exp(x_j - max), <= 1
      den[k] += e;
      cb[k] += j < me ? e : 0.f;
      p[k] = j == me ? e : p[k];
    }
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const float inv = 1.0f / den[k];
    S[k] = (den[k] - (float)POOL * __builtin_amdgcn_exp2f(zmin[k])) * inv;
    p[k] *= inv;
    cb[k] *= inv;
  }
}

// 24-bit uniforms of the K units at (sequence n, position s): both 12-bit fields of every sampler
// group (the pooled sampler compares against cumulative sums, so the lazy fine call of
// sample_hidden does not apply)
template <class C>
__device__ __forceinline__ void hidden_uniforms24(uint32_t n, uint32_t s, uint32_t kind, uint32_t strand, const RngView& rng,
                                                  uint32_t step, float (&u)[C::KP], uint32_t group0 = 0) {
#pragma unroll
  for (int g = 0; g < C::NGRP; ++g) {
    const Philox4 rc = philox4x32(n, s, rng_word2(kind, strand, 0, group0 + (uint32_t)g), step, rng.seed_lo, rng.seed_hi);
    const Philox4 rf = philox4x32(n, s, rng_word2(kind, strand, 1, group0 + (uint32_t)g), step, rng.seed_lo, rng.seed_hi);
    auto unit = [&](auto I) {
      constexpr int i = decltype(I)::value;
      const int k = 10 * g + i;
      if (k < C::K) u[k] = (float)(philox_field12<i>(rc) * 4096u + philox_field12<i>(rf)) * 5.9604644775390625e-8f;
    };
    unit(IC<0>{}); unit(IC<1>{}); unit(IC<2>{}); unit(IC<3>{}); unit(IC<4>{});
    unit(IC<5>{}); unit(IC<6>{}); unit(IC<7>{}); unit(IC<8>{}); unit(IC<9>{});
  }
}

// sum_k log(1 + sum_j exp(x_j)) of the group that starts at position g0, per unit into acc[] (free energy,
// convRBM.py:664-665): ln2 * (-zmin) + ln(exp(-max) + sum_j exp(x_j - max))
template <int POOL, int KP, class ZFun>
__device__ __forceinline__ void pooled_softplus(ZFun zfun, int g0, float (&acc)[KP], int nvalid) {
  float zmin[KP], z[KP], den[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) zmin[k] = 0.f;
#pragma unroll 1
  for (int j = 0; j < POOL; ++j) {
    zfun(g0 + j, z);
#pragma unroll
    for (int k = 0; k < KP; ++k) zmin[k] = fminf(zmin[k], z[k]);
  }
#pragma unroll
  for (int k = 0; k < KP; ++k) den[k] = __builtin_amdgcn_exp2f(zmin[k]);
#pragma unroll 1
  for (int j = 0; j < POOL; ++j) {
    zfun(g0 + j, z);
#pragma unroll
    for (int k = 0; k < KP; ++k) den[k] += __builtin_amdgcn_exp2f(zmin[k] - z[k]);
  }
#pragma unroll
  for (int k = 0; k < KP; ++k)
    if (k < nvalid) acc[k] += 0.6931471805599453f * (__builtin_amdgcn_logf(den[k]) - zmin[k]);
}

// K-bit mask of the pooled sample at the lane's position: on iff cb <= u < cb + p (first index whose
// cumulative probability exceeds the group's uniform, convRBM.py:259-267)
template <class C>
__device__ __forceinline__ void pooled_sample(const float (&p)[C::KP], const float (&cb)[C::KP], const float (&u)[C::KP],
                                              uint32_t (&mask)[C::NW]) {
#pragma unroll
  for (int w = 0; w < C::NW; ++w) mask[w] = 0u;
#pragma unroll
  for (int k = 0; k < C::K; ++k) {
    const uint32_t one = (cb[k] + p[k] > u[k] && cb[k] <= u[k]) ? 1u : 0u;
    mask[k >> 5] |= one << (k & 31);
  }
}

// global precomputed tables -> LDS (plain float4 copy; batching the loads of a thread measured slower)
template <int NFLOATS>
__device__ __forceinline__ void copy_tables(float* dst, const float* src) {
  static_assert(NFLOATS % 4 == 0, "tables are float4 granular");
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int i = threadIdx.x; i < NFLOATS / 4; i += blockDim.x) d4[i] = s4[i];
}

// ---------------------------------------------------------------------------
// Tables, rebuilt whenever W, b or c change (set_params / apply_update):
//  Tf[g][r][k]  = -log2(e) * ( sum_{t<G, j=gG+t<M} W[k][(r>>2t)&3][j]  (+ b[k] in group 0) )
//    (ONE table: the reverse-complement strand, rc(W) = W[k][3-a][M-1-j] of convRBM.py:241, gathers from it
//     with the reverse-complemented letter window -- revcomp_window)
//    The gather therefore yields z = -x*log2(e), so exp(-x) is one v_exp_f32
//    (2^z) with no multiply; kernels that need x itself use x = -ln(2)*z.
//    Pad columns k >= K get +1e30 in group 0: exp(-x) = inf, sigmoid -> 0.
//  Tv[jr][ch][pat] (float4 over letters) = log2(e) * sum_{bit in pat} W[5ch+bit][:][M-1-jr]
//  Tvr          = same for rc(W)                           (convRBM.py:279-287)
//  Ws[jr+4][k]  (float4 over letters) = log2(e) * W[k][:][M-1-jr] for 0 <= jr < M, else 0   (sparse top-down)
//  Wsr          = same for rc(W)
//  c            = log2(e) * c
//    The top-down images carry log2(e) so that the softmax over the four letters is four v_exp_f32
//    (2^(y - max)) without a multiply each (sample_letter); free_energy_body, which wants c itself, multiplies back.
// ---------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
struct TablesArgs {
  const float* W;   // (K,4,M)
  const float* b;   // (K)
  const float* c;   // (4)
  float* out;       // Cfg::TABLES_ALL floats
};

// entries first, first + stride, ... of the table images
// LIMIT: entries [0, LIMIT) only (Cfg::TAB: the gather table alone)
template <class C, int LIMIT = C::TABLES_ALL>
__device__ void build_tables_range(const TablesArgs& a, int first, int stride) {
  constexpr int K = C::K, M = C::M;
  for (int idx = first; idx < LIMIT; idx += stride) {
    float val = 0.f;
    if (idx < C::TAB) {                                      // gather table
      const int t0 = idx;
      const int k = t0 % C::KP, r = (t0 / C::KP) % C::ROWS, g = t0 / (C::KP * C::ROWS);
      if (k < K) {
        for (int t = 0; t < C::G; ++t) {
          const int j = g * C::G + t;
          if (j < M) {
            const int al = (r >> (2 * t)) & 3;
            val += a.W[(k * 4 + al) * M + j];
          }
        }
        if (g == 0) val += a.b[k];
        val *= -LOG2E;
      } else if (g == 0) {
        val = 1e30f;
      }
    } else if (idx < C::OFF_C) {                             // dense top-down tables
      const bool rc = C::DS && idx >= C::OFF_TVR;
      const int t0 = idx - (rc ? C::OFF_TVR : C::OFF_TV);
      const int al = t0 & 3, pat = (t0 >> 2) & 31, ch = (t0 >> 7) % C::NCH, jr = (t0 >> 7) / C::NCH;
      for (int bit = 0; bit < 5; ++bit) {
        const int k = 5 * ch + bit;
        if (k < K && ((pat >> bit) & 1))
          val += rc ? a.W[(k * 4 + (3 - al)) * M + jr] : a.W[(k * 4 + al) * M + (M - 1 - jr)];
      }
      val *= LOG2E;
    } else if (idx < C::TABLES) {
      val = LOG2E * a.c[idx - C::OFF_C];
    } else if (idx < C::OFF_C2) {                            // sparse top-down tables, zero rows at both ends
      const bool rc = C::DS && idx >= C::OFF_WSR;
      const int t0 = idx - (rc ? C::OFF_WSR : C::OFF_WS);
      const int al = t0 & 3, k = (t0 >> 2) % K, jr = (t0 >> 2) / K - 4;
      if (jr >= 0 && jr < M) val = LOG2E * (rc ? a.W[(k * 4 + (3 - al)) * M + jr] : a.W[(k * 4 + al) * M + (M - 1 - jr)]);
    } else {
      val = LOG2E * a.c[idx - C::OFF_C2];
    }
    a.out[idx] = val;
  }
}

template <class C>
__device__ void build_tables_body(const TablesArgs& a) {
  build_tables_range<C>(a, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}
// the gather table of another letter grouping (plain chain launches of small models take larger groups than the
// fused training launch, whose LDS is shared with the statistics: crbm_api.hip, solo_group)
template <class C>
__device__ void build_gather_table_body(const TablesArgs& a) {
  build_tables_range<C, C::TAB>(a, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// ---------------------------------------------------------------------------
// h_given_v, dense outputs: _bottomUpActivity / _bottomUpProbability /
// _bottomUpSample (convRBM.py:238-275) and motifHitProbs (:507-514).
// mode 0: forward strand, 1: reverse-complement strand, 2: sigma(x + x').
// ---------------------------------------------------------------------------
struct HgvArgs {
  const float* tables;
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

// The kernel's model as a SLAB of a larger one (crbm_api.hip, slab_launch_hgv: h|v of a chain on the generic path): the
// sampled units go, as bits [k0, k0 + K) and below Kfull, into the larger model's mask rows (OR: the rows start at zero and
// slabs may overlap); group0 = k0 / 10 keeps every unit on its own counter (units draw in groups of ten).  Units below
// kskip are not counted in `ones`.  (A struct of its own: HgvArgs, and with it crbm_hgv, stay what they were.)
struct HgvMasksArgs {
  HgvArgs g;                  // act, prob, sample unused
  uint32_t* masks;            // [n][Lh][NWfull]
  int32_t NWfull, k0, Kfull, kskip;
  uint32_t group0;
};

template <class C>
__device__ void hgv_body(const HgvArgs& a) {
  constexpr int KP = C::KP, K = C::K, M = C::M;
  HIP_DYNAMIC_SHARED(float, smem);
  float* T0 = smem;
  copy_tables<C::TAB>(T0, a.tables + C::OFF_TF);
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
      const uint32_t* lrow = a.letters + (size_t)nn * a.LW;
      auto zfun = [&](int pos, float (&z)[KP]) {
        const LetterWin<M> w = letter_window<M>(lrow, pos);
        conv_gather<C>(T0, a.mode == 1 ? revcomp_window<M>(w) : w, z);
        if (a.mode == 2) conv_gather<C, true>(T0, revcomp_window<M>(w), z);
      };
      float x[KP];
      zfun(s, x);
      uint32_t mask[C::NW];
      float p[KP];
      if constexpr (C::POOL > 1) {
        float cb[KP], S[KP];
        pooled_probs<C::POOL, KP>(zfun, s, p, cb, S);
        if (want_sample) {
          float u[KP];
          hidden_uniforms24<C>(a.rng.seq_offset + (uint32_t)nn, (uint32_t)(s - s % C::POOL), a.kind, strand, a.rng, a.rng.step, u);
          pooled_sample<C>(p, cb, u, mask);
        }
      } else if (want_sample) {
        sample_hidden<C, 1>(x, a.rng.seq_offset + (uint32_t)nn, (uint32_t)s, a.kind, strand, a.rng, a.rng.step,
                               mask, p);
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) p[k] = sigmoid_z(x[k]);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const size_t idx = ((size_t)nn * K + k) * a.Lh + s;
        if (a.act) a.act[idx] = x_of_z(x[k]);
        if (a.prob) a.prob[idx] = p[k];
        if (want_sample) {
          const uint32_t hb = (mask[k >> 5] >> (k & 31)) & 1u;
          if (a.sample) a.sample[idx] = (float)hb;
          cnt += hb;
        }
      }
    }
  }
  if (a.ones && cnt) atomicAdd(a.ones, cnt);
}

// The same pass for a SLAB of a larger model's chain (HgvArgs::masks; crbm_api.hip, slab_launch_hgv): the sample only, as bits
// of the larger model's mask rows.  A kernel of its own so that crbm_hgv stays the code it was (its register and scratch
// budget at 256 motifs is tight); compiled for models of up to 64 motifs only (crbm_jit.h).
template <class C>
__device__ void hgv_masks_body(const HgvMasksArgs& ma) {
  const HgvArgs& a = ma.g;
  constexpr int KP = C::KP, K = C::K, M = C::M;
  HIP_DYNAMIC_SHARED(float, smem);
  float* T0 = smem;
  copy_tables<C::TAB>(T0, a.tables + C::OFF_TF);
  __syncthreads();
  const uint32_t strand = a.mode == 1 ? 1u : 0u;
  unsigned long long cnt = 0;
  const int ntiles = (a.n + a.TS - 1) / a.TS;
  const int w0 = ma.k0 >> 5, sh = ma.k0 & 31;
  const int kend = min(K, ma.Kfull - ma.k0);          // local units that exist in the larger model
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n0 = tile * a.TS;
    const int ns = min(a.TS, a.n - n0);
    const uint32_t items = (uint32_t)ns * (uint32_t)a.Lh;
    for (uint32_t i = threadIdx.x; i < items; i += blockDim.x) {
      const uint32_t nl = fastdiv(i, a.divLh);
      const int s = (int)(i - nl * (uint32_t)a.Lh);
      const int nn = n0 + (int)nl;
      const uint32_t* lrow = a.letters + (size_t)nn * a.LW;
      auto zfun = [&](int pos, float (&z)[KP]) {
        const LetterWin<M> w = letter_window<M>(lrow, pos);
        conv_gather<C>(T0, a.mode == 1 ? revcomp_window<M>(w) : w, z);
      };
      float x[KP];
      zfun(s, x);
      uint32_t mask[C::NW];
      float p[KP];
      if constexpr (C::POOL > 1) {
        float cb[KP], S[KP], u[KP];
        pooled_probs<C::POOL, KP>(zfun, s, p, cb, S);
        hidden_uniforms24<C>(a.rng.seq_offset + (uint32_t)nn, (uint32_t)(s - s % C::POOL), a.kind, strand, a.rng, a.rng.step, u, ma.group0);
        pooled_sample<C>(p, cb, u, mask);
      } else {
        sample_hidden<C, 0>(x, a.rng.seq_offset + (uint32_t)nn, (uint32_t)s, a.kind, strand, a.rng, a.rng.step, mask, p, nullptr, ma.group0);
      }
      // bits [k0, k0 + K) of the row's NWfull words, cut at Kfull (the last slab of a model may reach past its end);
      // units below kskip belong to the neighbouring slab as well and are counted there
      uint32_t* wp = ma.masks + ((size_t)nn * a.Lh + s) * ma.NWfull;
#pragma unroll
      for (int w = 0; w < C::NW; ++w) {
        const int lo = min(32, max(0, ma.kskip - 32 * w)), hi = min(32, max(0, kend - 32 * w));
        const uint32_t keep = hi >= 32 ? 0xFFFFFFFFu : (1u << hi) - 1u;
        const uint32_t counted = keep & (lo >= 32 ? 0u : ~((1u << lo) - 1u));
        mask[w] &= keep;
        cnt += (unsigned)__popc(mask[w] & counted);
      }
#pragma unroll
      for (int w = 0; w <= C::NW; ++w) {
        const uint32_t cur = w < C::NW ? mask[w < C::NW ? w : 0] : 0u, prev = w > 0 ? mask[w - 1] : 0u;
        const uint32_t bits = sh ? (cur << sh) | (prev >> (32 - sh)) : cur;
        if (bits && w0 + w < ma.NWfull) atomicOr(&wp[w0 + w], bits);      // (the slabs of a launch share words: slab_hgv_body)
      }
    }
  }
  if (a.ones && cnt) atomicAdd(a.ones, cnt);
}

// ===========================================================================
// Gradient statistics on the matrix cores.
//
// VH[k,a,j] = sum_{n,s} P[n,k,s] * [letter(n, s+j) == a]   (convRBM.py:327-337)
// is a K x (4*M) output contracted over N*Lh positions: a GEMM whose one operand
// (the unfolded one-hot) is exact in any float format and whose other operand, P,
// is split into two f16 halves (22 significant bits, products exact in f32).  One
// v_mfma_f32_16x16x32_f16 contracts 32 consecutive hidden positions of a chain:
//   A (16 x 32): row i = filter column j = 16*jt + i of letter a, element (i, t) =
//                2^-14 * [letter(s0 + t + j) == a] -- eight consecutive bits of the
//                letter's bit plane, expanded to eight f16 by one 16-byte LUT read;
//   B (32 x 16): column i = motif 16*nt + i of one column kind (P of the forward
//                strand, P' of the rc strand, Q = P(1-P) for the sparsity gradient,
//                convRBM.py:440-451), element (t, i) = 2^14 * P[k, s0 + t], read
//                from an LDS image the h|v pass writes transposed ([column][position]).
// The accumulators (one 16x16 f32 tile per letter x column tile x motif tile) stay in
// registers for the whole kernel; waves of a block are combined through LDS at the
// end and the block writes one partial row.  Against the letter-bucketed LDS walk of
// round 1 (one LDS read and one VALU add per (k, j, position); 42.5 us per half at
// config #2) the parked probabilities are read once instead of M
// times and the adds run on the matrix pipe beside the VALU work of other waves.
// north_star's "no MFMA" is about the convolution and its transpose (4-wide output,
// one-hot operand: gathers); this contraction is the one dense product on the path.
// ===========================================================================
struct StatsGeom {
  int32_t GPC;                                   // groups per chain
  int32_t off_slices, slice, off_win, off_gw, off_pt;   // LDS layout (floats): first slice, slice size, offsets inside a slice
  int32_t lds_floats;                            // dynamic LDS of the launch (the end-of-kernel combine uses all of it)
  FastDiv divGPC;
  int32_t row, off_vh0, off_vh1, off_h0, off_h1, off_sw, off_sb, off_v;
  float* partials;                               // [blocks of the statistics grid][row]
};

// kinds, motif tiles per wave, roles; mirrors stats_mfma_layout()
template <class C, bool SP>
struct StatsRole {
  static constexpr int KINDS = 1 + C::DS + (SP ? 1 : 0);
  static constexpr int NTW = stats_ntw(C::NT, C::JT, KINDS);
  static constexpr int NR = C::NT / NTW;
  static constexpr int KW = 16 * NTW < C::K ? 16 * NTW : C::K;
  // The sparsity columns Q = P(1-P) are not parked: the wave that reads a P fragment derives the Q fragment from
  // it in registers (same formula, same bits), so the column image holds the strands' P only -- half the rows
  // of a single-stranded data half, a third fewer LDS instructions.  (Pooled units need the sum over their
  // pooling group for Q: those are parked.)
  static constexpr bool QDERIVE = SP && C::POOL == 1;
  static constexpr int PARKED = KINDS - (QDERIVE ? 1 : 0);
  static constexpr int ROWS = PARKED * KW + 1;
  // Column tiles of the accumulators.  A wave that holds all motifs (NR == 1, so KW == K) has the parked kinds
  // back to back in its image, rows kind * K + k: the tiles then cut that list every 16 rows regardless of
  // where a kind ends (PACKED) -- 20 motifs on two strands are 40 columns = 3 tiles instead of 2 x 2.  The
  // derived sparsity tiles follow; each comes from the P fragment of the same tile index (columns of that
  // fragment that belong to the other strand give garbage nobody reads).
  static constexpr int NPT_PACKED = cdiv(PARKED * C::K, 16);
#ifdef CRBM_STATS_UNPACKED     // A/B knob (CRBM_JIT_DEFINES): whole tiles per kind everywhere
  static constexpr bool PACKED = false;
#else
  static constexpr bool PACKED = NR == 1 && NPT_PACKED < PARKED * NTW;
#endif
  static constexpr int NPT = PACKED ? NPT_PACKED : PARKED * NTW;        // tiles read from the image
  static constexpr int NCT = NPT + (QDERIVE ? NTW : 0);                 // + derived ones
  static constexpr int NACC = C::NL * C::JT * NCT;
  // tile and lane-in-tile of column `col` of kind `kind` (kind == KINDS - 1 with QDERIVE: the derived tiles)
  static constexpr int tile_of(int kind, int col) {
    return (QDERIVE && kind == KINDS - 1) ? NPT + col / 16 : PACKED ? (kind * C::K + col) / 16 : kind * NTW + col / 16;
  }
  static constexpr int lane_of(int kind, int col) {
    return (QDERIVE && kind == KINDS - 1) ? col % 16 : PACKED ? (kind * C::K + col) % 16 : col % 16;
  }
  static constexpr int THREADS = stats_mfma_threads(NR);
  static constexpr int NQW = 4 * NTW;            // float4 quads of motifs a wave gathers
};

// A-fragment look-up tables: bit e of the index set -> f16 2^-14 (0x0400), else 0.
//   nibble form: 16 entries of 8 bytes (four f16), two reads per fragment -- 128 bytes, for the Gibbs
//                kernel whose LDS is spoken for;
//   byte form:   256 entries of 16 bytes (eight f16), one read per fragment -- 4 KB, stand-alone kernel.
template <bool BYTE_LUT>
__device__ __forceinline__ void stats_build_lut(uint32_t* lut) {
  constexpr int WORDS = BYTE_LUT ? 1024 : 32, PER = BYTE_LUT ? 4 : 2;
  for (int i = threadIdx.x; i < WORDS; i += blockDim.x) {
    const uint32_t n = (uint32_t)i / PER, q = (uint32_t)i % PER;
    lut[i] = ((n >> (2 * q)) & 1u ? 0x0400u : 0u) | ((n >> (2 * q + 1)) & 1u ? 0x04000000u : 0u);
  }
}

// bit t of the result = [letter t of the packed word == a], t = 0..15
__device__ __forceinline__ uint32_t letter_plane16(uint32_t word, uint32_t a) {
  uint32_t x = word ^ (a * 0x55555555u);          // 00 where the letter is a
  x = ~(x | (x >> 1)) & 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  return (x | (x >> 8)) & 0xFFFFu;
}

// The letter windows of a unit (2 groups x 4 letters x NPW 16-bit pieces: 64 bits for M <= 32, 128
// beyond) are built by 8 NPW lanes: lane t owns letter (t & 3), letter word (t >> 2) % NPW (16
// positions) of group slot t / (4 NPW) and stores its 16 plane bits straight into the window
// (win as 16-bit pieces: piece index (slot*4 + letter)*NPW + word).
// Returns the number of its positions that count for the letter statistics: a group owns its 32
// positions (words 0, 1), the last group of a chain also the tail up to L (words 2 ..).
template <int NPW>
__device__ __forceinline__ float stats_window_piece(unsigned short* win16, int t, uint32_t word, bool group_valid, int gi, int GPC, int L) {
  const int a = t & 3, w = (t >> 2) % NPW, slot = t / (4 * NPW);
  const uint32_t bits = group_valid ? letter_plane16(word, (uint32_t)a) : 0u;
  win16[(slot * 4 + a) * NPW + w] = (unsigned short)bits;
  const int p0 = 32 * gi + 16 * w;                              // first position of this piece
  const int limit = (w < 2 || gi == GPC - 1) ? L : 0;           // positions >= limit do not count
  const int nbits = limit - p0 < 0 ? 0 : (limit - p0 > 16 ? 16 : limit - p0);
  return (float)__popc(bits & ((1u << nbits) - 1u));
}

// z[] = the gather of motif quads [q0, q0 + NQW) (clamped to the model's NQ); z = -log2(e) * activation
template <class C, int NQW>
__device__ __forceinline__ void conv_gather_quads(const float* T, const LetterWin<C::M>& win, int q0, float (&z)[4 * NQW]) {
#pragma unroll
  for (int i = 0; i < 4 * NQW; ++i) z[i] = 0.f;   // quads beyond the model's NQ
  auto group = [&](int g, auto FIRST) {
    constexpr bool first = decltype(FIRST)::value != 0;   // the first group assigns
    const uint32_t r = window_bits<C::M>(win, 2 * C::G * g) & (uint32_t)(C::ROWS - 1);
    const float4* row = reinterpret_cast<const float4*>(T + (size_t)g * C::ROWS * C::KP) + (size_t)r * C::NQ + q0;
#pragma unroll
    for (int q = 0; q < NQW; ++q)
      if (q0 + q < C::NQ) {        // wave-uniform
        const float4 t = row[q];
        if (first) {
          z[4 * q + 0] = t.x; z[4 * q + 1] = t.y; z[4 * q + 2] = t.z; z[4 * q + 3] = t.w;
        } else {
          z[4 * q + 0] += t.x; z[4 * q + 1] += t.y; z[4 * q + 2] += t.z; z[4 * q + 3] += t.w;
        }
      }
  };
  group(0, IC<1>{});
  if constexpr (C::NG * NQW <= 48) {
#pragma unroll
    for (int g = 1; g < C::NG; ++g) group(g, IC<0>{});
  } else {
#pragma unroll 1
    for (int g = 1; g < C::NG; ++g) group(g, IC<0>{});
  }
}

// One 32-position group (slot 0 or 1 of the wave's unit): all accumulator tiles of the wave.
//   Pt  : the wave's column image (row kind*KW + i, stride STATS_RS, position slot*32 + t)
//   win : the group's four letter windows (NPW/2 words each)
//   R   : the wave's role (StatsRole): column tiles read from the image (per kind, or cut from the kinds back to
//         back: PACKED) and, with QDERIVE, the tiles of the last kind (Q = P(1-P), sparsity), which are not in the
//         image but derived from the fragments of kind 0
template <class C, class R, bool BYTE_LUT>
__device__ __forceinline__ void stats_mfma_group(const float* Pt, const uint32_t* win, const uint32_t* lut, int nt0, int slot,
                                                 floatx4 (&acc)[R::NACC]) {
  constexpr int NTW = R::NTW, NCT = R::NCT, NPT = R::NPT, KW = R::KW, ZROW = R::PARKED * KW;
  const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
  HalfFrag bhi[NCT], blo[NCT];
#pragma unroll
  for (int c = 0; c < NPT; ++c) {
    int row;
    if constexpr (R::PACKED) {
      row = 16 * c + i16 < R::PARKED * C::K ? 16 * c + i16 : ZROW;               // the kinds back to back; behind them the all-zero row
    } else {
      const int kind = c / NTW, t = c - kind * NTW, kl = 16 * t + i16;
      row = (16 * (nt0 + t) + i16 < C::K) ? kind * KW + kl : ZROW;              // motifs beyond K: the all-zero row
    }
    const float4* src = reinterpret_cast<const float4*>(Pt + (size_t)row * STATS_RS + slot * 32 + 8 * g);
    const float4 x0 = src[0], x1 = src[1];
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    split_f16(x, bhi[c], blo[c]);
    if (R::QDERIVE && c < NTW) {          // kind 0 fills the first tiles in either layout
      float q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) q[e] = fmaf(-x[e] * STATS_PSCALE_INV, x[e], x[e]);   // 2^14 * P(1-P) from 2^14 * P
      split_f16(q, bhi[NPT + c], blo[NPT + c]);
    }
  }
  // the spare row (filter column M of letter 0, NL == 3): all ones -> sum over positions of P, i.e. H (crbm_layout.h, NL)
  const uint32_t ones_row = (C::NL == 3 && i16 == C::M % 16) ? 0x04000400u : 0u;
#pragma unroll
  for (int a = 0; a < C::NL; ++a) {
    unsigned long long bits = 0ull;
    if constexpr (C::NPW == 4) {
      const uint2 w = reinterpret_cast<const uint2*>(win)[a];
      bits = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
    }
#pragma unroll
    for (int jt = 0; jt < C::JT; ++jt) {
      uint32_t byte;
      if constexpr (C::NPW == 4) byte = (uint32_t)(bits >> (8 * g + i16 + 16 * jt));
      else {
        // 128-bit window: the two words around bit 8 g + i16 + 16 jt (<= 87: word index <= 2), funnel-shifted
        const int sa = 8 * g + i16 + 16 * jt;
        const uint32_t* w32 = win + a * (C::NPW / 2) + (sa >> 5);
        const unsigned long long two = (unsigned long long)w32[0] | ((unsigned long long)w32[1] << 32);
        byte = (uint32_t)(two >> (sa & 31));
      }
      HalfFrag af;
      if constexpr (BYTE_LUT) {
        const uint4 f = reinterpret_cast<const uint4*>(lut)[byte & 255u];
        af.r[0] = f.x; af.r[1] = f.y; af.r[2] = f.z; af.r[3] = f.w;
      } else {
        const uint2 f0 = reinterpret_cast<const uint2*>(lut)[byte & 15u], f1 = reinterpret_cast<const uint2*>(lut)[(byte >> 4) & 15u];
        af.r[0] = f0.x; af.r[1] = f0.y; af.r[2] = f1.x; af.r[3] = f1.y;
      }
      if (C::NL == 3 && a == 0 && jt == C::JT - 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) af.r[e] |= ones_row;
      }
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        floatx4& d = acc[(a * C::JT + jt) * NCT + c];
        d = mfma_16x16x32_f16(af, bhi[c], d);
        d = mfma_16x16x32_f16(af, blo[c], d);
      }
    }
  }
}

// Combines the waves of a block and writes the block's partial row: vh / vh' / sw blocks, the H
// and sparsity-bias sums (filter column 0 pairs every hidden position with exactly one letter:
// sum_s P[k,s] = sum_a VH[k,a,0]; with NL == 3 the all-ones row gives H directly and the fourth letter is
// derived from it: crbm_layout.h) and the letter counts.  Kind by kind, every wave parks its
// accumulator tiles of that kind in a buffer of its own (as many waves at a time as the LDS
// holds: normally all), then every thread sums the copies of its output elements in wave order
// -- a barrier pair per kind instead of one per wave, and the same result for every launch
// geometry of a block.
// Uses the LDS of the block from its base on (everything is dead by now); all threads call it.
template <class C, bool SP>
__device__ __forceinline__ void stats_mfma_finish(const StatsGeom& sg, float* lds, int nt0, int block_row,
                                                  const floatx4 (&acc)[StatsRole<C, SP>::NACC], float vcount) {
  using R = StatsRole<C, SP>;
  constexpr int K = C::K, M = C::M, KAM = K * 4 * M, KINDS = R::KINDS, NTW = R::NTW, NR = R::NR, KW = R::KW;
  constexpr int PW = KW * 4 * M;                  // floats a wave parks per kind: [letter][column][motif of the role]
  const int nthr = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = nthr >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  float* total = lds;                   // [KAM], one kind at a time
  float* xch = lds + KAM;               // [nwaves][4]
  float* wbuf = xch + 64;
  int cw = min(nwaves, (sg.lds_floats - KAM - 64) / PW);   // waves parked at a time: whole sets of roles
  cw = max(NR, cw / NR * NR);
  float* out = sg.partials + (size_t)block_row * sg.row;
  __syncthreads();
#pragma unroll
  for (int kind = 0; kind < KINDS; ++kind) {
    for (int w0 = 0; w0 < nwaves; w0 += cw) {
      if (wave >= w0 && wave < w0 + cw) {
        float* mine = wbuf + (size_t)(wave - w0) * PW;
#pragma unroll
        for (int a = 0; a < C::NL; ++a)
#pragma unroll
          for (int jt = 0; jt < C::JT; ++jt)
#pragma unroll
            for (int c = R::tile_of(kind, 0); c <= R::tile_of(kind, KW - 1); ++c) {   // the tiles that hold columns of this kind
              // this lane's column of the tile, as a motif of the kind (negative or >= KW: another kind's or padding)
              const int kl = 16 * (c - R::tile_of(kind, 0)) + i16 - R::lane_of(kind, 0);
              const floatx4 d = acc[(a * C::JT + jt) * R::NCT + c];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * g + r;
                const bool mine_k = kl >= 0 && kl < KW && 16 * nt0 + kl < K;
                if (j < M && mine_k) mine[(a * M + j) * KW + kl] = d[r];   // motif fastest: lanes of a tile row hit consecutive banks
                // the all-ones row: H, parked in the (unused) slot of letter 3, column 0
                if (C::NL == 3 && a == 0 && j == M && mine_k) mine[(3 * M) * KW + kl] = d[r];
              }
            }
      }
      __syncthreads();
      for (int e = threadIdx.x; e < (C::NL == 3 ? (3 * M + 1) * K : KAM); e += nthr) {   // e = (letter*M + column)*K + motif
        const int aj = e / K, k = e - aj * K;
        const int role = (k >> 4) / NTW, kl = k - 16 * NTW * role;
        const int o = k * 4 * M + aj;                                     // its place in the (K,4,M) output block
        float t = w0 == 0 ? 0.f : total[o];
        for (int w = w0 + role; w < min(nwaves, w0 + cw); w += NR)       // the waves of this motif's role, in wave order
          t += wbuf[(size_t)(w - w0) * PW + aj * KW + kl];
        total[o] = t;
      }
      __syncthreads();
    }
    // kinds: 0 = P (vh, h), 1 = P' when doublestranded (vh', h'), last = Q when SP (sw, sb)
    const int off_w = kind == 0 ? sg.off_vh0 : (C::DS && kind == 1) ? sg.off_vh1 : sg.off_sw;
    const int off_k = kind == 0 ? sg.off_h0 : (C::DS && kind == 1) ? sg.off_h1 : sg.off_sb;
    if constexpr (C::NL == 3) {
      // letter 3 from H (in its column-0 slot) and the three contracted letters
      for (int i = threadIdx.x; i < KAM; i += nthr) {
        const int k = i / (4 * M), aj = i - k * 4 * M;
        const float* tk = total + k * 4 * M;
        out[off_w + i] = aj < 3 * M ? tk[aj] : tk[3 * M] - ((tk[aj - 3 * M] + tk[aj - 2 * M]) + tk[aj - M]);
      }
      for (int k = threadIdx.x; k < K; k += nthr) out[off_k + k] = total[k * 4 * M + 3 * M];
    } else {
      for (int i = threadIdx.x; i < KAM; i += nthr) out[off_w + i] = total[i];
      for (int k = threadIdx.x; k < K; k += nthr)
        out[off_k + k] = (total[(k * 4) * M] + total[(k * 4 + 1) * M]) + (total[(k * 4 + 2) * M] + total[(k * 4 + 3) * M]);
    }
    // (the next kind's first barrier orders these reads before `total` is rewritten)
  }
  // letter counts: lane l counted letter (l & 3); lanes of one class -> wave -> block, fixed order
  const float cls = class_sum(vcount);
  if (lane < 4) xch[wave * 4 + lane] = cls;
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
    for (int w = 0; w < nwaves; ++w) t += xch[w * 4 + threadIdx.x];
    out[sg.off_v + threadIdx.x] = t;
  }
}

struct StatsMfmaArgs {
  const float* tables;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  int32_t off_tab;      // gather tables inside LDS (floats), after the slices
  int32_t debug;        // profiling only (results are wrong when set): 1 skips the MFMA steps, 2 the h|v arithmetic, 4 the
                        // staging, 8 the combine + output, 16 the table copy, 32 the whole loop
  int32_t nblocks;      // blocks of the statistics grid (0: the whole launch)
  StatsGeom sg;
};

// Stand-alone statistics of (letters, n, L): the data half of a training step (SP: with the
// sparsity columns) and the model half of models whose accumulator set is too large to ride
// in the Gibbs kernel.  A wave loops over units of two groups: lanes 0-7 stage the groups'
// letter words (fetched one unit ahead) and build the letter windows, every lane computes P of
// one hidden position for the wave's motifs and parks it transposed, then the MFMA steps run.
template <class C, bool SP, bool BYTE_LUT = true>
__device__ void stats_mfma_body(const StatsMfmaArgs& a, int bid = -1) {   // bid: block of the statistics grid (default: blockIdx.x)
  using R = StatsRole<C, SP>;
  constexpr int K = C::K, M = C::M, NTW = R::NTW, NR = R::NR, KW = R::KW;
  HIP_DYNAMIC_SHARED(float, smem);
  const StatsGeom& sg = a.sg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
  float* slice = smem + sg.off_slices + (size_t)wave * sg.slice;
  constexpr int NPW = C::NPW;
  uint32_t* win = reinterpret_cast<uint32_t*>(slice + sg.off_win);
  uint32_t* gw = reinterpret_cast<uint32_t*>(slice + sg.off_gw);
  float* Pt = slice + sg.off_pt;
  float* Tf = smem + a.off_tab;
  if (!(a.debug & 16)) {
    copy_tables<C::TAB>(Tf, a.tables + C::OFF_TF);
    stats_build_lut<BYTE_LUT>(lut);
    for (int i = lane; i < R::ROWS * STATS_RS; i += 64) Pt[i] = 0.f;   // the zero row (and the pad columns) stay zero
  }
  __syncthreads();

  const int role = wave % NR, wave_in_role = wave / NR, waves_per_role = nwaves / NR;
  const int nt0 = role * NTW;
  floatx4 acc[R::NACC];
#pragma unroll
  for (int t = 0; t < R::NACC; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  float vcount = 0.f;                                  // letter (lane & 3), lanes 0-7 of role-0 waves

  const int GPC = sg.GPC;
  const int ngroups = a.n * GPC;                       // the host keeps n * GPC below 2^31
  const int nunits = (ngroups + 1) / 2;
  if (bid < 0) bid = (int)blockIdx.x;
  const int nblk = a.nblocks > 0 ? a.nblocks : (int)gridDim.x;
  const int ustride = nblk * waves_per_role;
  // letter words of a unit: lane t < 8 NPW fetches word (t >> 2) % NPW of group slot t / (4 NPW) (four lanes,
  // one per letter, fetch the same word: one transaction) and later turns it into its window piece
  unsigned short* win16 = reinterpret_cast<unsigned short*>(win);
  auto fetch_word = [&](int u) -> uint32_t {
    if (lane >= 8 * NPW || u >= nunits) return 0u;
    const int G = 2 * u + lane / (4 * NPW);
    if (G >= ngroups) return 0u;
    const uint32_t chain = fastdiv((uint32_t)G, sg.divGPC);
    const int w = 2 * (G - (int)chain * GPC) + (lane >> 2) % NPW;
    return w < a.LW ? a.letters[(size_t)chain * a.LW + w] : 0u;
  };
  int u = bid * waves_per_role + wave_in_role;
  if (a.debug & 32) u = nunits;
  uint32_t pre = fetch_word(u);
  for (; u < nunits; u += ustride) {
    const int G0 = 2 * u;
    if (lane < 8 * NPW && !(a.debug & 4)) {
      const int G = G0 + lane / (4 * NPW);
      int gi = 0;
      if (G < ngroups) gi = G - (int)fastdiv((uint32_t)G, sg.divGPC) * GPC;
      if ((lane & 3) == 0) gw[lane >> 2] = pre;        // the packed words themselves feed the h|v gather below
      const float cnt = stats_window_piece<NPW>(win16, lane, pre, G < ngroups, gi, GPC, a.L);
      if (role == 0) vcount += cnt;
    }
    __builtin_amdgcn_wave_barrier();
    pre = fetch_word(u + ustride);                     // in flight during this unit
    // ---- P (and P', Q) of this lane's hidden position for the wave's motifs, parked transposed ----
    {
      const int slot = lane >> 5, e = lane & 31;
      const int G = G0 + slot;
      bool valid = G < ngroups && !(a.debug & 2);
      uint32_t chain = 0u;
      int s = 0;
      if (valid) {
        chain = fastdiv((uint32_t)G, sg.divGPC);
        s = 32 * (G - (int)chain * GPC) + e;
        valid = s < a.Lh;
      }
      float* col = Pt + lane;
      if constexpr (C::POOL > 1) {
        // pooled units: the lane evaluates its whole pooling group (it may reach back over the staged
        // window: letters come from the global row); Q = dP_group/dx_s = P_s (1 - sum of the group's P)
        if (valid) {
          const uint32_t* lrow = a.letters + (size_t)chain * a.LW;
          constexpr int NV = 4 * R::NQW;
          float p[NV], cb[NV], S[NV];
#pragma unroll
          for (int strand = 0; strand <= C::DS; ++strand) {
            auto zfun = [&](int pos, float (&z)[NV]) {
              const LetterWin<M> w = letter_window<M>(lrow, pos);
              conv_gather_quads<C, R::NQW>(Tf, strand ? revcomp_window<M>(w) : w, 4 * nt0, z);
            };
            pooled_probs<C::POOL, NV>(zfun, s, p, cb, S);
#pragma unroll
            for (int kl = 0; kl < KW; ++kl)
              if (16 * nt0 + kl < K) {
                const float ps = 16384.0f * p[kl];
                col[(size_t)(strand * KW + kl) * STATS_RS] = ps;
                if (SP && strand == 0) col[(size_t)((1 + C::DS) * KW + kl) * STATS_RS] = ps * (1.0f - S[kl]);
              }
          }
        } else {
#pragma unroll
          for (int r = 0; r < R::PARKED * KW; ++r) col[(size_t)r * STATS_RS] = 0.f;
        }
      } else
      if (valid) {
        const LetterWin<M> wl = letter_window<M>(gw + NPW * slot, e);
        float z[4 * R::NQW];
        conv_gather_quads<C, R::NQW>(Tf, wl, 4 * nt0, z);
#pragma unroll
        for (int kl = 0; kl < KW; ++kl)
          if (16 * nt0 + kl < K) {                     // wave-uniform
            const float ps = fast_rcp(fmaf(exp_neg_x(z[kl]), STATS_PSCALE_INV, STATS_PSCALE_INV));   // 2^14 * sigma(x)
            col[(size_t)kl * STATS_RS] = ps;
            if (SP && !R::QDERIVE) col[(size_t)((1 + C::DS) * KW + kl) * STATS_RS] = fmaf(-ps * STATS_PSCALE_INV, ps, ps);   // 2^14 * P(1-P)
          }
        if (C::DS) {
          conv_gather_quads<C, R::NQW>(Tf, revcomp_window<M>(wl), 4 * nt0, z);
#pragma unroll
          for (int kl = 0; kl < KW; ++kl)
            if (16 * nt0 + kl < K)
              col[(size_t)(KW + kl) * STATS_RS] = fast_rcp(fmaf(exp_neg_x(z[kl]), STATS_PSCALE_INV, STATS_PSCALE_INV));
        }
      } else {
#pragma unroll
        for (int r = 0; r < R::PARKED * KW; ++r) col[(size_t)r * STATS_RS] = 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- the MFMA steps of the unit ----
#pragma unroll
    for (int slot = 0; slot < 2; ++slot)
      if (G0 + slot < ngroups && !(a.debug & 1))
        stats_mfma_group<C, R, BYTE_LUT>(Pt, win + 2 * NPW * slot, lut, nt0, slot, acc);
    __builtin_amdgcn_wave_barrier();                   // the slice is rewritten by the next unit
  }
  if (a.debug & 8) return;
  stats_mfma_finish<C, SP>(sg, smem, nt0, bid, acc, vcount);
}

// ---------------------------------------------------------------------------
// The persistent-chain kernel: `steps` Gibbs steps
//   v ~ P(v|h,h')  (convRBM.py:317-325)   then   h,h' ~ P(h|v)  (:269-275)
// for every chain of a tile, entirely in LDS (convRBM.py:397-408).
// ---------------------------------------------------------------------------
struct GibbsArgs {
  const float* tables;    // the model's table images (crbm_layout.h)
  const float* tables_tf = nullptr;   // sparse variant: the gather table of THIS kernel's letter grouping when it is not the
  int32_t off_ws = 0;                 // model's (then the top-down tables start at tables + off_ws); null: tables, Cfg::OFF_WS
  uint32_t* hm;        // [nchains][Lf][NW] in/out
  uint32_t* hmp;       // reverse strand (ds) or null
  uint32_t* vout;      // [nchains][LWs] letters of the last visible sample
  int32_t nchains, Lf, Lv, S;
  int32_t nvb, nhb;    // items per chain: 4-position visible blocks, hidden positions
  int32_t Lrow;        // padded mask row (positions), multiple of 4
  int32_t LWs;         // letter words per chain row
  FastDiv divVB, divHB, divRow, divLfw;   // / nvb, / nhb, / (Lrow*NW), / (Lf*NW)
  int32_t steps;
  RngView rng;
  uint32_t* ones;      // [gridDim.x * waves per block] set bits of the final hidden state per wave (activity monitor), may be null
  unsigned long long* clock = nullptr;   // null, or {sum of wall ticks, sum of shader cycles, scratch, scratch}: block 0 adds the duration of
                               // this launch in both clocks (their ratio is the shader clock WHILE the kernel ran: crbm_time_gibbs)
  unsigned long long* timeline = nullptr;   // measurement aid: null, or two words: block 0's first and last wall-clock tick of this launch
  int32_t debug;       // profiling only: 1 skips the table copy, 2 the state load, 4 the state store
  // STATS variant only: the model half of the gradient statistics rides in the last h|v pass
  int32_t nblocks;     // blocks [0, nblocks) of the launch run the chain (0: the whole grid)
  int32_t stats_off;   // statistics region inside LDS (floats, 16-byte aligned), after the chain image
  StatsGeom sg;        // divGPC divides group indices of one tile
};

// letter of one visible position from its 4 top-down activations, given in units of log 2 (the tables carry log2(e))
__device__ __forceinline__ uint32_t sample_letter(float y0, float y1, float y2, float y3, float u) {
  const float mx = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
  const float e0 = __builtin_amdgcn_exp2f(y0 - mx), e1 = __builtin_amdgcn_exp2f(y1 - mx), e2 = __builtin_amdgcn_exp2f(y2 - mx),
              e3 = __builtin_amdgcn_exp2f(y3 - mx);
  const float t = u * ((e0 + e1) + (e2 + e3));
  return (uint32_t)(t >= e0) + (uint32_t)(t >= e0 + e1) + (uint32_t)(t >= (e0 + e1) + e2);
}

// Adds the top-down contributions of every set bit of one window word to the 4
// visible positions of a thread.  Bit b of the word is hidden unit k of window slot
// q = q_base + ql (b = ql*K + k); visible position i sees it through filter column
// jr = q - i, i.e. table row (q - i + 4): address (q_base*K + b)*16 + (4 - i)*K*16,
// out-of-range columns hit the zero rows of the table.  The rows of the next bit are
// requested before those of the current one are added (the LDS latency of a set
// bit overlaps the adds of the previous one); a lane that runs out of bits points
// at the leading zero rows.
// unit_base: hidden unit of bit 0 (models with more than 64 motifs walk a mask in several words).
template <class C>
__device__ __forceinline__ void topdown_bits(unsigned long long w, const char* tab, int q_base, float (&y)[4][4], int unit_base = 0) {
  constexpr int K = C::K;
  const char* tab_q = tab + ((size_t)q_base * K + unit_base) * 16;
  const char* idle = tab - K * 16;                     // + (4-i)*K*16 = rows 3-i: all zero
  auto rows_of = [&](unsigned long long bits) { return bits ? tab_q + (__ffsll(bits) - 1) * 16 : idle; };
  auto fetch = [&](const char* p, float4 (&t)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const float4*>(p + (4 - i) * K * 16);
  };
  float4 cur[4];
  fetch(rows_of(w), cur);
  while (w) {
    w &= w - 1ull;
    float4 nxt[4];
    fetch(rows_of(w), nxt);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[i][0] += cur[i].x; y[i][1] += cur[i].y; y[i][2] += cur[i].z; y[i][3] += cur[i].w;
      cur[i] = nxt[i];
    }
  }
}

// SPARSE selects the top-down variant (Cfg comment in crbm_layout.h); the LDS image
// differs accordingly.  The dense variant exists only for Cfg::DENSE models.
// STATS: the last h|v pass also produces the model half of the gradient statistics
// (convRBM.py:411-413: last-step probabilities, last-step visible sample): the pass runs
// in wave units of two 32-position groups, every lane parks the probabilities it has in
// registers anyway in the wave's LDS slice, and the wave contracts them with the visible
// sample (still in LDS) on the matrix cores -- no second read of v, no recomputation of P.
template <class C, bool SPARSE, bool STATS = false>
__device__ void gibbs_body(const GibbsArgs& a, int bid = -1) {   // bid: block of the chain grid (default: blockIdx.x)
  constexpr int KP = C::KP, M = C::M, NW = C::NW, NCH = C::NCH;
  static_assert(SPARSE || C::DENSE, "no dense top-down tables for this model");
  using SR = StatsRole<C, false>;
  static_assert(!STATS || SR::NR == 1, "the fused statistics need all motifs of a position in one wave");
  constexpr int LTAB = SPARSE ? C::SP_TABLES : C::TABLES;
  HIP_DYNAMIC_SHARED(float, smem);
  const float* Tf = smem;
  const float* cv = smem + (SPARSE ? C::SP_C : C::OFF_C);
  uint32_t* hm = reinterpret_cast<uint32_t*>(smem + LTAB);
  uint32_t* hmp = hm + (size_t)a.S * a.Lrow * NW;
  uint32_t* let = hmp + (C::DS ? (size_t)a.S * a.Lrow * NW : 0);
  uint32_t* fixq = let + (size_t)a.S * a.LWs;   // [count][-][FIXQ_CAP entries]: undecided units of the running h|v pass
#ifdef CRBM_INLINE_EXACT_PATH                     // A/B knob (CRBM_JIT_DEFINES): every wave resolves its own undecided units
  constexpr bool DEFER = false;
#else
  // Measured: double-stranded models gain (config #5: 180 -> 155 us per step), single-stranded
  // ones lose to the extra barrier and the serial drain (config #2: 16.9 -> 17.5 us per step; with the
  // 1024-thread blocks of round 3 21.7 -> 22.1 us per one-step launch, config #4 2.273 -> 2.279 ms).  Pooled
  // models draw both fields for every group anyway.
  constexpr bool DEFER = C::POOL == 1 && C::DS;
#endif

  // The tables are copied while the state loads of the block's first tile are in flight (below).
  bool tables_done = (a.debug & 1) != 0;
  auto copy_all_tables = [&]() {
    if (SPARSE) {
      copy_tables<C::TAB>(smem, a.tables_tf ? a.tables_tf : a.tables);
      copy_tables<C::WS * (1 + C::DS) + 4>(smem + C::SP_WS, a.tables + (a.tables_tf ? a.off_ws : C::OFF_WS));
    } else {
      copy_tables<C::TABLES>(smem, a.tables);
    }
    tables_done = true;
  };

  const int rowW = a.Lrow * NW;
  const uint32_t per = (uint32_t)(a.Lf * NW);           // state words per chain
  const int ntiles = (a.nchains + a.S - 1) / a.S;
  if (bid < 0) bid = (int)blockIdx.x;
  if (a.clock && bid == 0 && threadIdx.x == 0) {        // both clocks at the start of the launch, parked in memory (no live registers)
    a.clock[2] = realtime_ticks();
    a.clock[3] = shader_cycles();
  }
  if (a.timeline && bid == 0 && threadIdx.x == 0) a.timeline[0] = realtime_ticks();
  const int nblk = a.nblocks > 0 ? a.nblocks : (int)gridDim.x;
  int nset = 0;
  // statistics state (STATS): the wave's LDS slice, accumulator tiles, letter counts
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  float* sreg = nullptr;
  uint32_t* swin = nullptr;
  float* sPt = nullptr;
  floatx4 sacc[STATS ? SR::NACC : 1];
  float vcount = 0.f;
  if constexpr (STATS) {
    sreg = smem + a.stats_off;
    float* sslice = sreg + a.sg.off_slices + (size_t)wave * a.sg.slice;
    swin = reinterpret_cast<uint32_t*>(sslice + a.sg.off_win);
    sPt = sslice + a.sg.off_pt;
#pragma unroll
    for (int t = 0; t < SR::NACC; ++t) sacc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    stats_build_lut<false>(reinterpret_cast<uint32_t*>(sreg));
    for (int i = lane; i < SR::ROWS * STATS_RS; i += 64) sPt[i] = 0.f;   // the zero row stays zero
  }
  if (threadIdx.x == 0) fixq[0] = 0u;
  // the pads of the mask rows (M-1 positions in front, the rest behind) stay zero for the whole kernel
  {
    const int padw = rowW - (int)per, front = (M - 1) * NW;
    for (int idx = threadIdx.x; idx < a.S * padw; idx += blockDim.x) {
      const int nl = idx / padw, e = idx - nl * padw;
      const int d = nl * rowW + (e < front ? e : e + (int)per);
      hm[d] = 0u;
      if (C::DS) hmp[d] = 0u;
    }
  }
  for (int tile = bid; tile < ntiles; tile += nblk) {
    const int n0 = tile * a.S;
    const int ns = min(a.S, a.nchains - n0);
    __syncthreads();
    // chain state -> zero-padded LDS rows (hidden position s sits at s + M-1).  The
    // chains of a tile are one contiguous range of ns*per words in global memory:
    // all loads of a thread are issued before the first LDS store (one memory round
    // trip per tile), 16 bytes per lane where the alignment allows.
    if (!(a.debug & 2)) {
      const uint32_t nwords = (uint32_t)ns * per;
      auto lds_index = [&](uint32_t i) {           // word i of the tile -> padded row position
        const uint32_t nl = fastdiv_tile(i, a.divLfw);
        return nl * (uint32_t)rowW + (uint32_t)((M - 1) * NW) + (i - nl * per);
      };
      const size_t g0 = (size_t)n0 * per;
      if (((g0 | nwords) & 3u) == 0u) {
        constexpr int UN = 2;
        const uint4* src = reinterpret_cast<const uint4*>(a.hm + g0);
        const uint4* srcp = C::DS ? reinterpret_cast<const uint4*>(a.hmp + g0) : nullptr;
        for (uint32_t base0 = 0; base0 < nwords / 4; base0 += UN * blockDim.x) {   // the same rounds for every thread
          const uint32_t base = base0 + threadIdx.x;
          uint4 t[UN], tp[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const uint32_t i4 = base + u * blockDim.x;
            if (i4 < nwords / 4) {
              t[u] = src[i4];
              if (C::DS) tp[u] = srcp[i4];
            }
          }
          if (!tables_done) copy_all_tables();
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const uint32_t i4 = base + u * blockDim.x;
            if (i4 < nwords / 4) {
              const uint32_t e[4] = {t[u].x, t[u].y, t[u].z, t[u].w};
              const uint32_t ep[4] = {tp[u].x, tp[u].y, tp[u].z, tp[u].w};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const uint32_t d = lds_index(4 * i4 + c);
                hm[d] = e[c];
                if (C::DS) hmp[d] = ep[c];
              }
            }
          }
        }
      } else {
        constexpr int UN = 8;
        for (uint32_t base0 = 0; base0 < nwords; base0 += UN * blockDim.x) {
          const uint32_t base = base0 + threadIdx.x;
          uint32_t t[UN], tp[UN];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const uint32_t i = base + u * blockDim.x;
            if (i < nwords) {
              t[u] = a.hm[g0 + i];
              if (C::DS) tp[u] = a.hmp[g0 + i];
            }
          }
          if (!tables_done) copy_all_tables();
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const uint32_t i = base + u * blockDim.x;
            if (i < nwords) {
              const uint32_t d = lds_index(i);
              hm[d] = t[u];
              if (C::DS) hmp[d] = tp[u];
            }
          }
        }
      }
    }
    if (!tables_done) copy_all_tables();
    for (int idx = threadIdx.x; idx < ns * a.LWs; idx += blockDim.x) let[idx] = 0u;
    // The exact path of the hidden sampler, block-wide (DEFER).  A unit whose coarse 12-bit field cannot
    // decide it (2^-12 of all units, but some lane of a wave in 14.5 % of all rounds at K = 10) is not
    // resolved by its wave -- that costs the whole wave a second Philox call and a pass over its ten
    // units -- but queued: (item, unit, strand) in one word.  After the pass the block drains the queue,
    // one entry per thread: activation of that one unit (same table rows, same order of additions),
    // both Philox calls, the exact comparison of sample_hidden, one atomic OR into the mask word.
    auto resolve = [&](uint32_t e, int st, bool last) -> uint32_t {
      const uint32_t it = e & 0xFFFFFu, k = (e >> 20) & 0x7FFu, strand = e >> 31;
      const uint32_t nl = fastdiv_tile(it, a.divHB);
      const int s = (int)(it - nl * (uint32_t)a.nhb);
      const uint32_t gn = a.rng.seq_offset + (uint32_t)(n0 + nl);
      LetterWin<M> win = letter_window<M>(let + nl * (uint32_t)a.LWs, s);
      if (strand) win = revcomp_window<M>(win);
      const float* T = Tf + k;
      float z = 0.f;
#pragma unroll
      for (int g = 0; g < C::NG; ++g) {
        const uint32_t r = window_bits<C::M>(win, 2 * C::G * g) & (uint32_t)(C::ROWS - 1);
        const float t = T[((size_t)g * C::ROWS + r) * KP];
        z = g == 0 ? t : z + t;
      }
      const uint32_t g10 = k / 10u, i10 = k - 10u * g10, step = a.rng.step + (uint32_t)st;
      const Philox4 rc = philox4x32(gn, (uint32_t)s, rng_word2(KIND_CHAIN_H, strand, 0, g10), step, a.rng.seed_lo, a.rng.seed_hi);
      const Philox4 rf = philox4x32(gn, (uint32_t)s, rng_word2(KIND_CHAIN_H, strand, 1, g10), step, a.rng.seed_lo, a.rng.seed_hi);
      const float t = 4096.0f * fast_rcp(1.0f + exp_neg_x(z));
      const float frac = t - (float)philox_field12_dyn(rc, (int)i10);
      const uint32_t one = frac * 4096.0f > (float)philox_field12_dyn(rf, (int)i10) ? 1u : 0u;
      if (one) {
        const uint32_t bit = 1u << (k & 31u), w = k >> 5;
        if (last) { if (!(a.debug & 4)) atomicOr((strand ? a.hmp : a.hm) + (size_t)n0 * per + it * (uint32_t)NW + w, bit); }
        else atomicOr((strand ? hmp : hm) + (nl * (uint32_t)(rowW - (int)per) + it * (uint32_t)NW + (uint32_t)((M - 1) * NW)) + w, bit);
      }
      return one;
    };
    // queue the undecided units of one item (rare; lanes without any do nothing)
    auto defer_units = [&](const uint32_t (&pending)[C::NGRP], uint32_t it, uint32_t strand, int st, bool last) -> uint32_t {
      uint32_t any = 0u, ones = 0u;
#pragma unroll
      for (int g = 0; g < C::NGRP; ++g) any |= pending[g];
      if (__any(any != 0u)) {
#pragma unroll
        for (int g = 0; g < C::NGRP; ++g) {
          uint32_t m = pending[g];
          while (m) {
            const uint32_t i = (uint32_t)__ffs(m) - 1u;
            m &= m - 1u;
            const uint32_t e = it | ((10u * (uint32_t)g + i) << 20) | (strand << 31);
            const uint32_t slot = atomicAdd(&fixq[0], 1u);
            if (slot < (uint32_t)FIXQ_CAP) fixq[2 + slot] = e;
            else ones += resolve(e, st, last);        // queue full (never at the sizes that run): resolve in place
          }
        }
      }
      return ones;
    };
    auto drain_queue = [&](int st, bool last) {
      // the last pass stored its mask words to global memory: they must have arrived before another
      // wave's atomic OR may touch them (s_waitcnt vmcnt(0); the barrier itself only waits for LDS)
      if (last) wait_vector_memory();
      __syncthreads();                                // every mask word of the pass is stored, every entry queued
      const uint32_t nq = min(fixq[0], (uint32_t)FIXQ_CAP);
      for (uint32_t e = threadIdx.x; e < nq; e += blockDim.x) {
        const uint32_t one = resolve(fixq[2 + e], st, last);
        if (last) nset += (int)one;
      }
    };
    for (int st = 0; st < a.steps; ++st) {
      __syncthreads();
      if (threadIdx.x == 0) fixq[0] = 0u;             // read by drain_queue before this barrier, filled again after the next
      // ---- v | h : y[a,p] = c[a] + sum_{k,j} W[k,a,j] h[k,p-j] (+ rc strand) ----
      for (uint32_t it = threadIdx.x; it < (uint32_t)(ns * a.nvb); it += blockDim.x) {
        const uint32_t nl = fastdiv_tile(it, a.divVB);
        const int pb = (int)(it - nl * (uint32_t)a.nvb);
        const int p0 = 4 * pb;
        float y[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { y[i][0] = cv[0]; y[i][1] = cv[1]; y[i][2] = cv[2]; y[i][3] = cv[3]; }
        if constexpr (!SPARSE) {
          // masks p0 .. p0+M+2 (padded row), then one float4 table row per
          // (filter column, 5-bit chunk) of each mask
          constexpr int NMV = (M + 3 + 3) / 4;
#pragma unroll
          for (int strand = 0; strand <= C::DS; ++strand) {
            const uint4* mrow = reinterpret_cast<const uint4*>((strand ? hmp : hm) + (size_t)nl * a.Lrow + p0);
            const char* Tv = reinterpret_cast<const char*>(smem + (strand ? C::OFF_TVR : C::OFF_TV));
            uint32_t off[4 * NMV][NCH];
#pragma unroll
            for (int v4 = 0; v4 < NMV; ++v4) {
              const uint4 mm = mrow[v4];
              const uint32_t m4[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch)
                  off[4 * v4 + e][ch] = (5 * ch >= 4 ? (m4[e] >> (5 * ch - 4)) : (m4[e] << (4 - 5 * ch))) & 0x1F0u;
            }
#pragma unroll
            for (int jr = 0; jr < M; ++jr)
#pragma unroll
              for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const float4 t = *reinterpret_cast<const float4*>(Tv + (jr * NCH + ch) * 512 + off[i + jr][ch]);
                  y[i][0] += t.x; y[i][1] += t.y; y[i][2] += t.z; y[i][3] += t.w;
                }
          }
        } else {
          // window slots q = 0 .. M+2 are the masks at padded-row positions p0 .. p0+M+2;
          // their bits are packed into 64-bit words (as many whole masks as fit) and every
          // set bit adds one table row to each of the 4 positions
          constexpr int NSLOT = M + 3;
          constexpr int PPW = NW == 1 ? 64 / C::K : 1;         // masks per word
          constexpr int NWORD = (NSLOT + PPW - 1) / PPW;
#pragma unroll
          for (int strand = 0; strand <= C::DS; ++strand) {
            const uint32_t* mrow = (strand ? hmp : hm) + ((size_t)nl * a.Lrow + p0) * NW;
            const char* tab = reinterpret_cast<const char*>(smem + (strand ? C::SP_WSR : C::SP_WS));
            if constexpr (NW == 1) {
              constexpr int NMV = (NSLOT + 3) / 4;
              uint32_t m[4 * NMV];
#pragma unroll
              for (int v4 = 0; v4 < NMV; ++v4) {
                const uint4 mm = reinterpret_cast<const uint4*>(mrow)[v4];
                m[4 * v4] = mm.x; m[4 * v4 + 1] = mm.y; m[4 * v4 + 2] = mm.z; m[4 * v4 + 3] = mm.w;
              }
#pragma unroll
              for (int wi = 0; wi < NWORD; ++wi) {
                unsigned long long w = 0ull;
#pragma unroll
                for (int t = 0; t < PPW; ++t)
                  if (wi * PPW + t < NSLOT) w |= (unsigned long long)m[wi * PPW + t] << (t * C::K);
                topdown_bits<C>(w, tab, wi * PPW, y);
              }
            } else {
#pragma unroll 1
              for (int q = 0; q < NSLOT; ++q) {
                if constexpr (NW == 2) {
                  const uint2 mm = *reinterpret_cast<const uint2*>(mrow + 2 * q);
                  topdown_bits<C>((unsigned long long)mm.x | ((unsigned long long)mm.y << 32), tab, q, y);
                } else {
                  // more than 64 motifs: 64 units at a time (a mask is NW words, 4-byte aligned only)
#pragma unroll
                  for (int w2 = 0; 2 * w2 < NW; ++w2) {
                    const uint32_t lo = mrow[NW * q + 2 * w2], hi = 2 * w2 + 1 < NW ? mrow[NW * q + 2 * w2 + 1] : 0u;
                    topdown_bits<C>((unsigned long long)lo | ((unsigned long long)hi << 32), tab, q, y, 64 * w2);
                  }
                }
              }
            }
          }
        }
        // one Philox call serves the 4 positions of the block (convRBM.py:301-315)
        const Philox4 r = philox4x32(a.rng.seq_offset + (uint32_t)(n0 + nl), (uint32_t)pb,
                                        rng_word2(KIND_CHAIN_V, 0, 0, 0), a.rng.step + (uint32_t)st,
                                        a.rng.seed_lo, a.rng.seed_hi);
        uint32_t byte = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t l = sample_letter(y[i][0], y[i][1], y[i][2], y[i][3], u01(r.v[i]));
          byte |= (p0 + i < a.Lv ? l : 0u) << (2 * i);
        }
        reinterpret_cast<unsigned char*>(let + (size_t)nl * a.LWs)[pb] = (unsigned char)byte;
      }
      __syncthreads();
      bool fused_pass = false;
      if constexpr (STATS) if (st == a.steps - 1) {
        fused_pass = true;
        // ---- last h | v with the model statistics: wave units of two 32-position groups ----
        constexpr int KW = SR::KW;
        const int GPC = a.sg.GPC, ngl = ns * GPC;
        for (int u = wave; 2 * u < ngl; u += nwaves) {
          const int G0 = 2 * u;
          constexpr int NPW = C::NPW;
          if (lane < 8 * NPW) {                 // letter windows of the two groups (and the letter counts)
            const int G = G0 + lane / (4 * NPW), w = (lane >> 2) % NPW;
            int gi = 0;
            uint32_t word = 0u;
            if (G < ngl) {
              const uint32_t nl = fastdiv_tile((uint32_t)G, a.sg.divGPC);
              gi = G - (int)nl * GPC;
              if (2 * gi + w < a.LWs) word = let[(size_t)nl * a.LWs + 2 * gi + w];
            }
            vcount += stats_window_piece<NPW>(reinterpret_cast<unsigned short*>(swin), lane, word, G < ngl, gi, GPC, a.Lv);
          }
          __builtin_amdgcn_wave_barrier();
          {
            const int slot = lane >> 5, G = G0 + slot;
            bool valid = G < ngl;
            uint32_t nl = 0u;
            int s = 0;
            if (valid) {
              nl = fastdiv_tile((uint32_t)G, a.sg.divGPC);
              s = 32 * (G - (int)nl * GPC) + (lane & 31);
              valid = s < a.Lf;
            }
            float* col = sPt + lane;
            if (valid) {
              const uint32_t gn = a.rng.seq_offset + (uint32_t)(n0 + nl);
              const LetterWin<M> win = letter_window<M>(let + (size_t)nl * a.LWs, s);
#pragma unroll
              for (int strand = 0; strand <= C::DS; ++strand) {
                float x[KP], p[KP];
                conv_gather<C>(Tf, strand ? revcomp_window<M>(win) : win, x);
                uint32_t mask[NW];
                uint32_t pend[C::NGRP];
                sample_hidden<C, 2, DEFER>(x, gn, (uint32_t)s, KIND_CHAIN_H, (uint32_t)strand, a.rng,
                                           a.rng.step + (uint32_t)st, mask, p, pend);
                // the last pass of the launch: the new state goes straight to global memory
                uint32_t* dst = (strand ? a.hmp : a.hm) + (size_t)(n0 + nl) * per + (size_t)s * NW;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                  if (!(a.debug & 4)) store1_streaming(dst + w, mask[w]);
                  nset += __popc(mask[w]);
                }
                if constexpr (DEFER) nset += (int)defer_units(pend, nl * (uint32_t)a.nhb + (uint32_t)s, (uint32_t)strand, st, true);
#pragma unroll
                for (int k = 0; k < C::K; ++k) col[(size_t)(strand * KW + k) * STATS_RS] = p[k];
              }
            } else {
#pragma unroll
              for (int r = 0; r < SR::KINDS * KW; ++r) col[(size_t)r * STATS_RS] = 0.f;
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int slot = 0; slot < 2; ++slot)
            if (G0 + slot < ngl)
              stats_mfma_group<C, SR, false>(sPt, swin + 2 * C::NPW * slot, reinterpret_cast<const uint32_t*>(sreg), 0, slot, sacc);
          __builtin_amdgcn_wave_barrier();       // the slice is rewritten by the next unit
        }
      }
      if (!fused_pass) {
      // ---- h | v : x[k,s] = b[k] + sum_j W[k, letter[s+j], j]; h = [sigma(x) > u] ----
      // one hidden position per item (nhb = Lf items per chain): the K units of a position
      // already give the instruction-level parallelism, and single positions spread evenly
      // over the waves of the block
      for (uint32_t it = threadIdx.x; it < (uint32_t)(ns * a.nhb); it += blockDim.x) {
        const uint32_t nl = fastdiv_tile(it, a.divHB);
        const int s = (int)(it - nl * (uint32_t)a.nhb);
        const uint32_t gn = a.rng.seq_offset + (uint32_t)(n0 + nl);
        const uint32_t* lrow = let + nl * (uint32_t)a.LWs;        // LDS offsets: 32-bit arithmetic
        const LetterWin<M> win = letter_window<M>(lrow, s);
#pragma unroll
        for (int strand = 0; strand <= C::DS; ++strand) {
          float x[KP], p[KP];
          uint32_t mask[NW], pend[C::NGRP];
          if constexpr (C::POOL > 1) {
            auto zfun = [&](int pos, float (&z)[KP]) {
              const LetterWin<M> w = letter_window<M>(lrow, pos);
              conv_gather<C>(Tf, strand ? revcomp_window<M>(w) : w, z);
            };
            float cb[KP], S[KP], u[KP];
            pooled_probs<C::POOL, KP>(zfun, s, p, cb, S);
            hidden_uniforms24<C>(gn, (uint32_t)(s - s % C::POOL), KIND_CHAIN_H, (uint32_t)strand, a.rng, a.rng.step + (uint32_t)st, u);
            pooled_sample<C>(p, cb, u, mask);
          } else {
          conv_gather<C>(Tf, strand ? revcomp_window<M>(win) : win, x);
          sample_hidden<C, 0, DEFER>(x, gn, (uint32_t)s, KIND_CHAIN_H, (uint32_t)strand, a.rng,
                                     a.rng.step + (uint32_t)st, mask, p, pend);
          }
          if (st == a.steps - 1) {
            // the last pass of the launch: the new state goes straight to global memory (the stores
            // drain behind the remaining items instead of in a phase of their own after a barrier)
            // (item `it` of the tile is word it * NW of its contiguous state: nhb = Lf positions per chain)
            uint32_t* dst = (strand ? a.hmp : a.hm) + (size_t)n0 * per + it * (uint32_t)NW;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              if (!(a.debug & 4)) store1_streaming(dst + w, mask[w]);
              nset += __popc(mask[w]);
            }
          } else {
            uint32_t* dst = (strand ? hmp : hm) + (nl * (uint32_t)(rowW - (int)per) + it * (uint32_t)NW + (uint32_t)((M - 1) * NW));
#pragma unroll
            for (int w = 0; w < NW; ++w) dst[w] = mask[w];
          }
          if constexpr (DEFER) {
            const uint32_t ones = defer_units(pend, it, (uint32_t)strand, st, st == a.steps - 1);
            if (st == a.steps - 1) nset += (int)ones;
          }
        }
      }
      }
      if constexpr (DEFER) drain_queue(st, st == a.steps - 1);
    }
    // LDS -> chain state (contiguous in global memory, 16-byte stores where aligned): only a launch
    // without steps (state round trip) still has the state in LDS here
    if (a.steps == 0) __syncthreads();
    if (a.steps == 0 && !(a.debug & 4)) {
      const uint32_t nwords = (uint32_t)ns * per;
      const size_t g0 = (size_t)n0 * per;
      auto lds_index = [&](uint32_t i) {
        const uint32_t nl = fastdiv_tile(i, a.divLfw);
        return nl * (uint32_t)rowW + (uint32_t)((M - 1) * NW) + (i - nl * per);
      };
      if (((g0 | nwords) & 3u) == 0u) {
        for (uint32_t i4 = threadIdx.x; i4 < nwords / 4; i4 += blockDim.x) {
          const uint32_t d0 = lds_index(4 * i4), d1 = lds_index(4 * i4 + 1), d2 = lds_index(4 * i4 + 2), d3 = lds_index(4 * i4 + 3);
          store4_streaming(a.hm + g0 + 4 * (size_t)i4, hm[d0], hm[d1], hm[d2], hm[d3]);
          if (C::DS) store4_streaming(a.hmp + g0 + 4 * (size_t)i4, hmp[d0], hmp[d1], hmp[d2], hmp[d3]);
        }
      } else {
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) {
          const uint32_t d = lds_index(i);
          a.hm[g0 + i] = hm[d];
          if (C::DS) a.hmp[g0 + i] = hmp[d];
        }
      }
    }
    if (a.vout && !(a.debug & 4))
      for (int idx = threadIdx.x; idx < ns * a.LWs; idx += blockDim.x)
        a.vout[(size_t)n0 * a.LWs + idx] = let[idx];
  }
  if (a.ones) {   // one plain store per wave, summed by the host (counts stay far below 2^24: exact in float)
    const float tot = wave_sum((float)nset);
    if ((threadIdx.x & 63) == 0) a.ones[bid * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (uint32_t)tot;
  }
  if (a.clock && bid == 0 && threadIdx.x == 0) {
    a.clock[0] += realtime_ticks() - a.clock[2];
    a.clock[1] += shader_cycles() - a.clock[3];
  }
  if (a.timeline && bid == 0 && threadIdx.x == 0) a.timeline[1] = realtime_ticks();
  if constexpr (STATS) stats_mfma_finish<C, false>(a.sg, smem, 0, bid, sacc, vcount);
}

// The local phase of a training step in ONE launch (convRBM.py:373-413): g.nblocks blocks advance the
// persistent chains and leave the model half of the statistics (gibbs_body, STATS variant), d.nblocks
// blocks compute the data half (stats_mfma_body with the sparsity columns) -- the two halves are
// independent given (W, b, c).  Same block size, similar LDS footprint; the two kinds alternate in
// the grid, so every CU runs chain blocks (VALU and LDS gathers) beside statistics blocks (matrix
// pipe) from the first cycle, and there is no second launch floor and no idle tail between them.
struct TrainLocalArgs {
  GibbsArgs g;
  StatsMfmaArgs d;
};

template <class C>
__device__ void train_local_body(const TrainLocalArgs& a) {
  if constexpr (C::FUSE_STATS) {
    const int nG = a.g.nblocks, nD = a.d.nblocks, m = min(nG, nD), b = (int)blockIdx.x;
    const bool chain = b < 2 * m ? !(b & 1) : nG > nD;      // alternate while both kinds last, then the rest
    const int idx = b < 2 * m ? b >> 1 : b - m;
    if (chain) gibbs_body<C, true, true>(a.g, idx);
    else stats_mfma_body<C, true, false>(a.d, idx);
  }
}

// ---------------------------------------------------------------------------
// Free energy (convRBM.py:657-697): one wave per sequence.
//   fe[n]    = ( -sum_{k,s} softplus(x) [- rc strand] - sum_p c[letter_p] ) / L
//   fem[n,k] =   -sum_s softplus(x[k]) [- rc strand] - sum_p c[letter_p]
// ---------------------------------------------------------------------------
struct FeArgs {
  const float* tables;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  float* fe;
  float* fem;
};

// softplus(x) = max(x,0) + log1p(exp(-|x|)) from z = -x*log2(e), on the hardware
// exp2/log2: log1p by its series below 2^-5 (where 1+t would round t away), by
// log2(1+t) above; relative error of either branch < 3e-7.
__device__ __forceinline__ float softplus_of_z(float z) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(z));
  const float series = t * (1.0f - t * (0.5f - t * (0.33333334f - 0.25f * t)));
  const float vialog = 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + t);
  return fmaxf(x_of_z(z), 0.f) + (t < 0.03125f ? series : vialog);
}

template <class C>
__device__ void free_energy_body(const FeArgs& a) {
  constexpr int KP = C::KP, K = C::K, M = C::M;
  HIP_DYNAMIC_SHARED(float, smem);
  float* Tf = smem;
  copy_tables<C::TAB>(Tf, a.tables + C::OFF_TF);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const float* cg = a.tables + C::OFF_C;          // log2(e) * c
  const float c0 = LN2 * cg[0], c1 = LN2 * cg[1], c2 = LN2 * cg[2], c3 = LN2 * cg[3];
  for (int nn = blockIdx.x * nwaves + wave; nn < a.n; nn += gridDim.x * nwaves) {
    const uint32_t* row = a.letters + (size_t)nn * a.LW;
    float acc[KP];
#pragma unroll
    for (int q = 0; q < KP; ++q) acc[q] = 0.f;
    if constexpr (C::POOL > 1) {
      // one term per pooling group, log(1 + sum_j exp(x_j)) (convRBM.py:664-665): a lane takes whole groups
      for (int g0 = lane * C::POOL; g0 < a.Lh; g0 += 64 * C::POOL)
#pragma unroll
        for (int strand = 0; strand <= C::DS; ++strand) {
          auto zfun = [&](int pos, float (&z)[KP]) {
            const LetterWin<M> w = letter_window<M>(row, pos);
            conv_gather<C>(Tf, strand ? revcomp_window<M>(w) : w, z);
          };
          pooled_softplus<C::POOL, KP>(zfun, g0, acc, K);
        }
    } else
    for (int s = lane; s < a.Lh; s += 64) {
      const LetterWin<M> win = letter_window<M>(row, s);
      float x[KP];
      conv_gather<C>(Tf, win, x);
#pragma unroll
      for (int q = 0; q < K; ++q) acc[q] += softplus_of_z(x[q]);
      if (C::DS) {
        conv_gather<C>(Tf, revcomp_window<M>(win), x);
#pragma unroll
        for (int q = 0; q < K; ++q) acc[q] += softplus_of_z(x[q]);
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
    for (int q = 0; q < K; ++q) {
      const float v = wave_sum(acc[q]);
      tot += v;
      if (lane == 0 && a.fem) a.fem[(size_t)nn * K + q] = -v - cs;
    }
    if (lane == 0 && a.fe) a.fe[nn] = (-tot - cs) / (float)a.L;
  }
}

// ===========================================================================
// Motif-hit summaries for data-set scale sweeps.  The reference's
// analysis code only ever reduces motifHitProbs() (convRBM.py:507-514): the max
// and the mean over positions per (sequence, motif) (utils.py:154, :242-244,
// :305) and the mean over sequences per (motif, position) (utils.py:113-116).
// This kernel produces those three reductions directly, so the dense
// (n,K,1,Lh) tensor never exists.  One wave per sequence; a lane owns HIT_NI
// positions (stride 64) of the chunk blockIdx.y of 64*HIT_NI positions and keeps
// their sums over sequences in registers.  Whatever is combined across waves, blocks
// or chunks is combined in FIXED POINT (2^-30 units, 64-bit integer atomics): integer
// addition does not care about the order, so the sums are the same bits in every run
// (float atomics gave run-to-run differences in the last place).
// ===========================================================================
constexpr float HIT_FX = 1073741824.0f;          // 2^30 units per 1.0: probabilities are <= 1, sums stay far below 2^63
struct HitArgs {
  const float* tables;
  const uint32_t* letters;
  int32_t n, L, Lh, LW;
  float* hmax;      // (n,K) max over positions   (zero-initialised when gridDim.y > 1)
  float* hsum;      // (n,K) mean over positions, written directly when gridDim.y == 1
  unsigned long long* hsum_fx;   // (n,K) fixed-point sums over positions when gridDim.y > 1 (zero-initialised; hit_finalize_kernel)
  float inv_Lh;     // 1 / Lh
  unsigned long long* pos_fx;    // (K,Lh) fixed-point sums over sequences (zero-initialised), or null
};
__device__ __forceinline__ unsigned long long to_fx(float v) { return (unsigned long long)(v * HIT_FX); }

template <class C>
__device__ void hit_summary_body(const HitArgs& a) {
  constexpr int KP = C::KP, K = C::K, M = C::M, NI = C::HIT_NI, PC = 64 * NI;
  constexpr bool BOTH = !C::DS;   // single-stranded models report sigma(x + x'), convRBM.py:511-514
  HIP_DYNAMIC_SHARED(float, smem);
  float* Tf = smem;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(Tf + C::TAB);   // [PC][K], block total of the waves' register sums (TAB: a multiple of 4 floats)
  copy_tables<C::TAB>(Tf, a.tables + C::OFF_TF);
  const int s0 = blockIdx.y * PC;
  if (a.pos_fx)
    for (int i = threadIdx.x; i < PC * K; i += blockDim.x) acc[i] = 0ull;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  float pacc[NI][K];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int q = 0; q < K; ++q) pacc[i][q] = 0.f;
  for (int nn = blockIdx.x * nwaves + wave; nn < a.n; nn += gridDim.x * nwaves) {
    const uint32_t* row = a.letters + (size_t)nn * a.LW;
    float mx[K], sm[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { mx[q] = 0.f; sm[q] = 0.f; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int s = s0 + lane + 64 * i;
      if (s < a.Lh) {
        auto zfun = [&](int pos, float (&zz)[KP]) {
          const LetterWin<M> w = letter_window<M>(row, pos);
          conv_gather<C>(Tf, w, zz);
          if (BOTH) conv_gather<C, true>(Tf, revcomp_window<M>(w), zz);
        };
        float z[KP], pp[KP];
        if constexpr (C::POOL > 1) {
          float cb[KP], S[KP];
          pooled_probs<C::POOL, KP>(zfun, s, pp, cb, S);
        } else {
          zfun(s, z);
#pragma unroll
          for (int q = 0; q < K; ++q) pp[q] = sigmoid_z(z[q]);
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const float p = pp[q];
          mx[q] = fmaxf(mx[q], p);
          sm[q] += p;
          pacc[i][q] += p;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const float m = wave_max_nonneg(mx[q]);
      const float t = wave_sum(sm[q]);
      if (lane == 0) {
        const size_t idx = (size_t)nn * K + q;
        if (gridDim.y == 1) {
          if (a.hmax) a.hmax[idx] = m;
          if (a.hsum) a.hsum[idx] = t * a.inv_Lh;
        } else {   // probabilities are >= 0: their bit patterns order like unsigned integers
          if (a.hmax) atomicMax(reinterpret_cast<unsigned int*>(a.hmax) + idx, __float_as_uint(m));
          if (a.hsum_fx) atomicAdd(a.hsum_fx + idx, to_fx(t));
        }
      }
    }
  }
  if (a.pos_fx) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int q = 0; q < K; ++q) atomicAdd(&acc[(lane + 64 * i) * K + q], to_fx(pacc[i][q]));
    __syncthreads();
    const int npos = min(PC, a.Lh - s0);
    for (int i = threadIdx.x; i < npos * K; i += blockDim.x) {
      const int sl = i / K, q = i - sl * K;
      if (acc[i]) atomicAdd(a.pos_fx + (size_t)q * a.Lh + s0 + sl, acc[i]);
    }
  }
}

// ---------------------------------------------------------------------------
// Normalise the (all-reduced) raw sums and apply the SGD+momentum update
// (convRBM.py:358-371, :415-436, :440-451).
// ---------------------------------------------------------------------------
struct UpdateArgs {
  const float* sums;
  const float *W, *b, *c, *vW, *vb, *vc;      // parameters and velocities before the step
  float *oW, *ob, *oc, *ovW, *ovb, *ovc;      // ... after it (the same buffers when a single block updates in place)
  int32_t K, M, ds;
  int32_t L_data, Lf;
  int32_t data_off, n_d, model_off, n_m;   // offsets into sums
  float lr, momentum, rho, lambda_rate;
  int32_t A;                                  // letters (big_update_kernel; the specialised update is compiled for 4)
};

// `nw`: when not null, the new W, b, c are also left there ([KAM][K][4], LDS of the caller);
// `store`: whether this block writes the new parameters and velocities to global memory.
// NE: filter weights a thread may have to take (a compile-time bound on ceil(KAM / blockDim), 0 = run-time loop).
// The kernel is a chain of memory round trips (sums, parameters, velocities -> a few hundred numbers): with
// NE known the loads of all of a thread's weights are issued before the first is used -- a run-time loop
// paid one round trip per iteration (three at config #2 with 256 threads: 7.3 us for the launch).
// Sums: how element i of the packed sums is read -- PlainSums: from a.sums; RankSums (update_tables_ipc_body): the
// ranks' published copies added in rank order.
struct PlainSums {
  const float* p;
  __device__ __forceinline__ float operator()(int i) const { return p[i]; }
};
template <int NE = 0, class Sums = PlainSums>
__device__ __forceinline__ void apply_update_body(const UpdateArgs& a, float* nw, bool store, const Sums& S) {
  const int K = a.K, M = a.M, KAM = K * 4 * M;
  const float n_d = S(a.n_d), n_m = S(a.n_m);
  const float cnt_d = n_d * (float)(a.L_data - M + 1);
  const float cnt_m = n_m * (float)a.Lf;
  // offsets into the packed sums: data half [vh][vh'][h][h'][sw][sb][v], model half [vh][vh'][h][h'][v]
  const int d_vh = a.data_off, d_vhp = d_vh + KAM, d_h = d_vh + 2 * KAM, d_hp = d_h + K;
  const int d_sw = d_vh + 2 * KAM + 2 * K, d_sb = d_sw + KAM, d_v = d_sb + K;
  const int m_vh = a.model_off, m_vhp = m_vh + KAM, m_h = m_vh + 2 * KAM, m_hp = m_h + K, m_v = m_hp + K;
  const float q = a.rho;
  struct WeightIn { float dvh, mvh, dvhp, mvhp, dh, dsw, vw, w; };
  auto load_weight = [&](int idx) {
    const int k = idx / (4 * M), al = (idx / M) & 3, j = idx % M;
    const int ridx = (k * 4 + (3 - al)) * M + (M - 1 - j);
    WeightIn in;
    in.dvh = S(d_vh + idx); in.mvh = S(m_vh + idx);
    in.dvhp = a.ds ? S(d_vhp + ridx) : 0.f; in.mvhp = a.ds ? S(m_vhp + ridx) : 0.f;
    in.dh = S(d_h + k); in.dsw = S(d_sw + idx); in.vw = a.vW[idx]; in.w = a.W[idx];
    return in;
  };
  auto finish_weight = [&](int idx, const WeightIn& in) {
    float gd = in.dvh / cnt_d, gm = in.mvh / cnt_m;
    if (a.ds) {
      gd = 0.5f * (gd + in.dvhp / cnt_d);
      gm = 0.5f * (gm + in.mvhp / cnt_m);
    }
    const float p = in.dh / cnt_d;
    const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
    const float reg = -g * in.dsw / cnt_d;
    const float v = a.momentum * in.vw + a.lr * (gd - gm - a.lambda_rate * reg);
    const float w = in.w + v;
    if (store) { a.ovW[idx] = v; a.oW[idx] = w; }
    if (nw) nw[idx] = w;
  };
  if constexpr (NE > 0) {
    WeightIn in[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = (int)threadIdx.x + e * (int)blockDim.x;
      if (idx < KAM) in[e] = load_weight(idx);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = (int)threadIdx.x + e * (int)blockDim.x;
      if (idx < KAM) finish_weight(idx, in[e]);
    }
    for (int idx = (int)threadIdx.x + NE * (int)blockDim.x; idx < KAM; idx += blockDim.x) finish_weight(idx, load_weight(idx));   // smaller blocks than assumed
  } else {
    for (int idx = threadIdx.x; idx < KAM; idx += blockDim.x) finish_weight(idx, load_weight(idx));
  }
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const float dh = S(d_h + k);
    float gd = dh / cnt_d, gm = S(m_h + k) / cnt_m;
    if (a.ds) {
      gd = 0.5f * (gd + S(d_hp + k) / cnt_d);
      gm = 0.5f * (gm + S(m_hp + k) / cnt_m);
    }
    const float p = dh / cnt_d;
    const float g = (q / p - (1.f - q) / (1.f - p)) / (float)K;
    const float reg = -g * S(d_sb + k) / cnt_d;
    const float v = a.momentum * a.vb[k] + a.lr * (gd - gm - a.lambda_rate * reg);
    const float bn = a.b[k] + v;
    if (store) { a.ovb[k] = v; a.ob[k] = bn; }
    if (nw) nw[KAM + k] = bn;
  }
  if (threadIdx.x < 4) {
    const int al = threadIdx.x;
    const float nd = n_d * (float)a.L_data, nm = n_m * (float)(a.Lf + M - 1);
    const float gd = S(d_v + al) / nd + S(d_v + 3 - al) / nd;     // a += a[::-1]  (:345)
    const float gm = S(m_v + al) / nm + S(m_v + 3 - al) / nm;
    const float v = a.momentum * a.vc[al] + a.lr * (gd - gm);
    const float cn = a.c[al] + v;
    if (store) { a.ovc[al] = v; a.oc[al] = cn; }
    if (nw) nw[KAM + K + al] = cn;
  }
}

// The end of a training step in one launch: the update, then the LDS table images of the new
// parameters for the next step's kernels.  Every block of the grid forms the complete update for
// itself (a few hundred numbers from the packed sums) into its LDS and builds its share of the
// tables from there; block 0 also stores the new parameters and velocities -- into the OTHER set of
// buffers (the host swaps the two sets after the launch), because the remaining blocks may still be
// reading the old ones.
constexpr int UPDATE_THREADS = 1024;   // block size of the update launches of the product (the emulator uses smaller ones)
struct UpdateTablesArgs {
  UpdateArgs u;
  float* tables;     // Cfg::TABLES_ALL floats
};

template <class C>
__device__ void update_tables_body(const UpdateTablesArgs& a) {
  HIP_DYNAMIC_SHARED(float, smem);
  constexpr int KAM = C::K * 4 * C::M;
  apply_update_body<cdiv(KAM, UPDATE_THREADS) <= 8 ? cdiv(KAM, UPDATE_THREADS) : 0>(a.u, smem, blockIdx.x == 0, PlainSums{a.u.sums});
  __syncthreads();
  TablesArgs t;
  t.W = smem; t.b = smem + KAM; t.c = smem + KAM + C::K; t.out = a.tables;
  build_tables_body<C>(t);
}

// ---------------------------------------------------------------------------
// The all-reduce of a data-parallel training step WITHOUT a collective launch (an alternative to
// ncclAllReduce for the few-KB sums buffer, whose cost is pure latency).  Every rank owns a buffer that all
// ranks of the node have mapped (hipIpcOpenMemHandle): per step parity one slot of sums and one flag word per
// SOURCE rank.  A rank PUSHES: the column reduction of its step writes its packed sums into its slot of every
// rank's buffer (its own included) and then raises its flag there to the step number
// (reduce_publish_pair_kernel; publish_sums_kernel where the sums were formed in d_sums).  The update launch of
// every rank waits for the R flags in ITS OWN buffer, adds the R slots IN RANK ORDER (so every rank forms the
// bit-identical sum, and with it the bit-identical update) and goes on as update_tables_body does: no
// collective launch, no extra kernel boundary, and nothing on the critical path reads remote memory -- a
// step's traffic over xGMI is one posted copy of the sums per peer (a few KB), where pulling would have every
// block of every rank's update fetch every peer's sums and poll remote flags.
// Buffer reuse: a rank writes parity p of a peer's buffer only after its own update of the step before, which
// waited for that peer's flag of that step -- raised after the peer's update two steps back had read parity p.
// The wait is bounded in TIME (the GPU's constant-rate clock, `timeout_ticks`; the host sets it from
// CRBM_IPC_TIMEOUT_MS, default 30 s): a peer that died must not hang this GPU, and ordinary skew between the ranks'
// hosts -- a rank that uploads or evaluates while the others already wait -- must not trip it (crbm_amd.crbm.fit
// also puts a host barrier in front of every epoch's first launch).  A wait that runs out raises `status[0]`,
// which is sticky: the launch that saw it and every later one apply NO update (parameters and velocities are
// carried over unchanged, the tables stay), and the host turns the status word into an error at its next
// synchronisation point (CRBM_ERR_IPC_TIMEOUT) -- the run fails with the last consistent model instead of going on
// with sums that never arrived.
// ---------------------------------------------------------------------------
constexpr int IPC_MAX_RANKS = 8;
struct IpcArgs {
  const float* sums[IPC_MAX_RANKS];       // the slots of this step's parity in THIS rank's buffer, one per source rank
  const uint32_t* flags[IPC_MAX_RANKS];   // ... and their flag words: == expect once a source's sums are complete
  uint32_t* status;                       // [0] != 0: a wait timed out (sticky)
  uint32_t expect;
  int32_t nranks, count;
  unsigned long long timeout_ticks;       // of realtime_ticks()
};
struct UpdateIpcArgs {
  UpdateTablesArgs ut;
  IpcArgs ipc;
};

// element i of the all-reduced sums: the R published copies added IN RANK ORDER (every rank forms the same fp32
// sum); all loads of an element are in flight together, nothing is staged in LDS
struct RankSums {
  const IpcArgs* ipc;
  __device__ __forceinline__ float operator()(int i) const {
    float v[IPC_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < IPC_MAX_RANKS; ++r) v[r] = r < ipc->nranks ? load_system(ipc->sums[r] + i) : 0.f;
    float t = v[0];
#pragma unroll
    for (int r = 1; r < IPC_MAX_RANKS; ++r) t += v[r];          // the zeros of absent ranks change nothing
    return t;
  }
};

template <class C>
__device__ void update_tables_ipc_body(const UpdateIpcArgs& a) {
  HIP_DYNAMIC_SHARED(float, smem);
  constexpr int KAM = C::K * 4 * C::M;
  const IpcArgs& ipc = a.ipc;
  float* nw = smem;                                                       // the new W, b, c
  uint32_t* gave_up = reinterpret_cast<uint32_t*>(smem + KAM + C::K + 4);   // this block does not apply the step
  if (threadIdx.x == 0) *gave_up = load_system(ipc.status);              // an earlier launch timed out: sticky
  __syncthreads();
  if ((int)threadIdx.x < ipc.nranks && *gave_up == 0u) {
    const uint32_t* f = ipc.flags[threadIdx.x];
    const uint64_t t0 = realtime_ticks();
    while ((int32_t)(load_system(f) - ipc.expect) < 0) {                  // flags only grow (step numbers)
      if (realtime_ticks() - t0 > ipc.timeout_ticks) {
        store_system(ipc.status, 1u);
        atomicOr(gave_up, 1u);
        break;
      }
      short_sleep();
    }
    fence_acquire_system();                                               // the sums behind the flag, not an older copy
  }
  __syncthreads();
  const UpdateArgs& u = a.ut.u;
  if (*gave_up != 0u) {
    // the host has already made the other buffer set the current one: carry the state over unchanged
    if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < KAM; i += blockDim.x) { u.oW[i] = u.W[i]; u.ovW[i] = u.vW[i]; }
      for (int i = threadIdx.x; i < C::K; i += blockDim.x) { u.ob[i] = u.b[i]; u.ovb[i] = u.vb[i]; }
      if (threadIdx.x < 4) { u.oc[threadIdx.x] = u.c[threadIdx.x]; u.ovc[threadIdx.x] = u.vc[threadIdx.x]; }
    }
    return;
  }
  apply_update_body<cdiv(KAM, UPDATE_THREADS) <= 8 ? cdiv(KAM, UPDATE_THREADS) : 0>(u, nw, blockIdx.x == 0, RankSums{&ipc});
  __syncthreads();
  TablesArgs t;
  t.W = nw; t.b = nw + KAM; t.c = nw + KAM + C::K; t.out = a.ut.tables;
  build_tables_body<C>(t);
}

// sums[dst(r)] = sum over partial rows of column r in a fixed order.  Block =
// 32 columns x (blockDim / 32) row groups: each thread adds the rows of its group (128-byte
// segments per row), the 32 groups are combined through LDS.  The partial
// buffer is never cleared: the kernel knows which column classes a statistics
// launch wrote and yields 0 for the classes that launch did not compute.
struct ReduceArgs {
  const float* partials;
  float* sums;
  int32_t nrows, row;
  int32_t K, KAM, ds, want_sparsity;
  int32_t skip_begin, skip_len;   // columns [skip_begin, skip_begin+skip_len) are dropped
  float n_value;                  // written after the last kept column
};

// Where a pushing rank's sums go: the same offset in its slot of every rank's buffer (byte distances from the
// rank's own buffer, which ReduceArgs::sums points into; own = 0)
struct PushTargets {
  int32_t n;
  long long delta[8];
};
__device__ __forceinline__ void push_store(float* own, const PushTargets& t, float v) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
    if (r < t.n) store_system(reinterpret_cast<float*>(reinterpret_cast<char*>(own) + t.delta[r]), v);
}

// SYSTEM: the sums go out with system-scope stores into the buffers the ranks of a node have mapped
// (reduce_publish_pair_kernel)
template <bool SYSTEM = false>
__device__ __forceinline__ void reduce_partials_body(const ReduceArgs& a, const PushTargets* push = nullptr) {
  __shared__ float part[32][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int ngrp = (int)(blockDim.x >> 5);          // 32 row groups in the 1024-thread launches of the product
  const int r = blockIdx.x * 32 + col;
  float tsum = 0.f;
  if (r < a.row) {
    // column classes of a partial row: [vh KAM][vh' KAM][h K][h' K][sw KAM][sb K][v 4]
    const int K = a.K, KAM = a.KAM;
    bool valid;
    if (r < KAM) valid = true;
    else if (r < 2 * KAM) valid = a.ds != 0;
    else if (r < 2 * KAM + K) valid = true;
    else if (r < 2 * KAM + 2 * K) valid = a.ds != 0;
    else if (r < 3 * KAM + 3 * K) valid = a.want_sparsity != 0;
    else valid = true;
    if (valid) {
      // eight rows in flight per thread (the kernel is a chain of memory round trips: ~2 000 rows of a few KB;
      // sixteen in flight measured slower, 5.1 -> 5.7 us); fixed order
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f;
      int i = grp;
      for (; i + 7 * ngrp < a.nrows; i += 8 * ngrp) {
        const float v0 = a.partials[(size_t)i * a.row + r], v1 = a.partials[(size_t)(i + ngrp) * a.row + r];
        const float v2 = a.partials[(size_t)(i + 2 * ngrp) * a.row + r], v3 = a.partials[(size_t)(i + 3 * ngrp) * a.row + r];
        const float v4 = a.partials[(size_t)(i + 4 * ngrp) * a.row + r], v5 = a.partials[(size_t)(i + 5 * ngrp) * a.row + r];
        const float v6 = a.partials[(size_t)(i + 6 * ngrp) * a.row + r], v7 = a.partials[(size_t)(i + 7 * ngrp) * a.row + r];
        t0 += v0; t1 += v1; t2 += v2; t3 += v3; t4 += v4; t5 += v5; t6 += v6; t7 += v7;
      }
      for (; i < a.nrows; i += ngrp) t0 += a.partials[(size_t)i * a.row + r];
      tsum = ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));
    }
  }
  part[grp][col] = tsum;
  __syncthreads();
  if (grp == 0 && r < a.row) {
    const bool skipped = r >= a.skip_begin && r < a.skip_begin + a.skip_len;
    if (!skipped) {
      float s = 0.f;
      for (int g = 0; g < ngrp; ++g) s += part[g][col];
      float* dst = a.sums + (r < a.skip_begin ? r : r - a.skip_len);
      if (SYSTEM) push_store(dst, *push, s); else *dst = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (SYSTEM) push_store(a.sums + (a.row - a.skip_len), *push, a.n_value); else a.sums[a.row - a.skip_len] = a.n_value;
  }
}


// both halves of a training step in one launch (blockIdx.y = 0: data, 1: model)
struct ReducePair {
  ReduceArgs half[2];
};

// ===========================================================================
// A model as a row of SLABS of the kernels' model (crbm_api.hip: slab_launch_stats, slab_launch_hgv -- generic DNA
// models take their statistics and the h|v of their chains from the specialised kernels, Ks motifs at a time).  One launch
// serves all slabs: blockIdx.y is the slab, whose table image, partial rows and first motif follow from it; the bodies
// above run unchanged (they look at blockIdx.x only).
// ===========================================================================
struct SlabPlan {
  int32_t Ks, K, last_k0;     // motifs per slab, motifs of the model, first motif of the last slab (it may overlap its neighbour)
};
__device__ __forceinline__ int slab_k0(const SlabPlan& p, int y) { return (y + 1) * p.Ks <= p.K ? y * p.Ks : p.last_k0; }

struct SlabTablesArgs {
  TablesArgs t;               // W, b of the whole model; out: the first slab's image
  SlabPlan plan;
  int32_t stride;             // floats between the images of consecutive slabs
};
template <class C>
__device__ void slab_tables_body(const SlabTablesArgs& s) {
  TablesArgs a = s.t;
  const int k0 = slab_k0(s.plan, (int)blockIdx.y);
  a.W += (size_t)k0 * 4 * C::M;
  a.b += k0;
  a.out += (size_t)blockIdx.y * s.stride;
  build_tables_body<C>(a);
}

struct SlabStatsArgs {
  StatsMfmaArgs a;            // tables, partial rows: the first slab's
  int32_t table_stride;       // floats
  int32_t pad_;
  long long partial_stride;   // floats between the partial rows of consecutive slabs
};
template <class C, bool SP, bool BYTE_LUT>
__device__ void slab_stats_body(const SlabStatsArgs& s) {
  StatsMfmaArgs a = s.a;
  a.tables += (size_t)blockIdx.y * s.table_stride;
  a.sg.partials += (size_t)blockIdx.y * s.partial_stride;
  stats_mfma_body<C, SP, BYTE_LUT>(a);
}

struct SlabHgvArgs {
  HgvMasksArgs m;             // k0, kskip, group0 are filled in per slab
  SlabPlan plan;
  int32_t table_stride;       // floats
};
template <class C>
__device__ void slab_hgv_body(const SlabHgvArgs& s) {
  HgvMasksArgs ma = s.m;
  const int y = (int)blockIdx.y, k0 = slab_k0(s.plan, y);
  ma.g.tables += (size_t)y * s.table_stride;
  ma.k0 = k0;
  ma.group0 = (uint32_t)(k0 / 10);
  ma.kskip = y > 0 ? max(0, slab_k0(s.plan, y - 1) + s.plan.Ks - k0) : 0;     // units the neighbouring slab counts
  hgv_masks_body<C>(ma);
}

// free energy: the per-motif terms -v_k - cs of every slab (free_energy_body's `fem`) into a scratch [slab][n][Ks];
// slab_fe_combine_kernel (crbm_kernels_generic.h) puts them together
struct SlabFeArgs {
  FeArgs a;                   // fe unused; fem: the first slab's scratch
  int32_t table_stride;       // floats
  int32_t pad_;
  long long fem_stride;       // floats between the scratch of consecutive slabs
};
template <class C>
__device__ void slab_fe_body(const SlabFeArgs& s) {
  FeArgs a = s.a;
  a.tables += (size_t)blockIdx.y * s.table_stride;
  a.fem += (size_t)blockIdx.y * s.fem_stride;
  a.fe = nullptr;
  free_energy_body<C>(a);
}

#ifdef CRBM_DEFINE_MISC_KERNELS
// ===========================================================================
// Model-independent kernels, compiled ahead of time into libcrbm_hip.so.
// ===========================================================================

// plain streaming copy, 16 bytes per lane (crbm_copy_bandwidth): a block moves one contiguous 16 KB
// chunk, four loads in flight per lane
__global__ void __launch_bounds__(256) copy_float4_kernel(const float4* src, float4* dst, size_t n4) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (base + 768 < n4) {
    const float4 a = src[base], b = src[base + 256], c = src[base + 512], d = src[base + 768];
    dst[base] = a; dst[base + 256] = b; dst[base + 512] = c; dst[base + 768] = d;
  } else {
    for (size_t i = base; i < n4 && i < base + 1024; i += 256) dst[i] = src[i];
  }
}

// one-hot fp32 (n,1,4,L) -> packed letters [n][LW]; flags[0] |= 1 on a column
// that is not exactly one-hot.
struct EncodeArgs {
  const float* v;
  uint32_t* letters;
  uint32_t* flags;
  int32_t n, L, LW;
  int32_t A;          // letters (the *_any kernels; the 2-bit kernels are DNA)
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

// one-hot fp32 (n,1,A,L) of any alphabet -> rows of bytes [n][LW] (four letters per word)
__global__ void encode_onehot_any_kernel(EncodeArgs a) {
  const long total = (long)a.n * a.LW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.LW);
    const int w = (int)(i - (long)nn * a.LW);
    uint32_t word = 0;
    bool bad = false;
    const float* base = a.v + (size_t)nn * a.A * a.L;
    for (int t = 0; t < 4; ++t) {
      const int p = w * 4 + t;
      if (p < a.L) {
        int ones = 0, zeros = 0;
        uint32_t l = 0u;
        for (int al = 0; al < a.A; ++al) {
          const float v = base[(size_t)al * a.L + p];
          if (v == 1.f) { if (!ones) l = (uint32_t)al; ++ones; }
          zeros += v == 0.f;
        }
        bad |= (ones != 1) | (zeros != a.A - 1);
        word |= l << (8 * t);
      }
    }
    a.letters[i] = word;
    if (bad) atomicOr(a.flags, 1u);
  }
}

// letter codes (one byte per base: 0..3 = A,C,G,T as in sequences.py:9-17) -> packed
// letters [n][LW]; flags[0] |= 1 on any other code.
struct EncodeCodesArgs {
  const unsigned char* codes;   // [n][L]
  uint32_t* letters;
  uint32_t* flags;
  int32_t n, L, LW;
  int32_t A;
};

__global__ void encode_codes_kernel(EncodeCodesArgs a) {
  const long total = (long)a.n * a.LW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.LW);
    const int w = (int)(i - (long)nn * a.LW);
    uint32_t word = 0;
    bool bad = false;
    const unsigned char* base = a.codes + (size_t)nn * a.L;
    for (int t = 0; t < 16; ++t) {
      const int p = w * 16 + t;
      if (p < a.L) {
        const uint32_t c = base[p];
        bad |= c > 3u;
        word |= (c & 3u) << (2 * t);
      }
    }
    a.letters[i] = word;
    if (bad) atomicOr(a.flags, 1u);
  }
}

// letter codes of any alphabet (one byte per letter, 0 .. A-1) -> rows of bytes [n][LW]; flags[0] |= 1 on any other code
__global__ void encode_codes_any_kernel(EncodeCodesArgs a) {
  const long total = (long)a.n * a.LW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.LW);
    const int w = (int)(i - (long)nn * a.LW);
    uint32_t word = 0;
    bool bad = false;
    const unsigned char* base = a.codes + (size_t)nn * a.L;
    for (int t = 0; t < 4; ++t) {
      const int p = w * 4 + t;
      if (p < a.L) {
        const uint32_t c = base[p];
        bad |= c >= (uint32_t)a.A;
        word |= (bad ? 0u : c) << (8 * t);
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
  int32_t A;
};

// rows of bytes -> one-hot fp32 (n,1,A,L)
__global__ void decode_onehot_any_kernel(DecodeArgs a) {
  const long total = (long)a.n * a.L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(i / a.L);
    const int p = (int)(i - (long)nn * a.L);
    const uint32_t l = reinterpret_cast<const unsigned char*>(a.letters + (size_t)nn * a.LW)[p];
    float* base = a.v + (size_t)nn * a.A * a.L + p;
    for (int al = 0; al < a.A; ++al) base[(size_t)al * a.L] = l == (uint32_t)al ? 1.f : 0.f;
  }
}

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

// ---------------------------------------------------------------------------
// v_given_h from DENSE hidden tensors (any finite values): _topDownActivity,
// _topDownProbability, _topDownSample (convRBM.py:277-325).  API/test pass;
// the training chain uses the Gibbs kernel.
// ---------------------------------------------------------------------------
struct VghArgs {
  const float* W;       // (K,4,M)
  const float* c;       // (4)
  int32_t K, M;
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
  const int K = a.K, M = a.M;
  float4* W4 = reinterpret_cast<float4*>(smem);   // [j][k] -> W[k][0..3][j]
  for (int idx = threadIdx.x; idx < M * K; idx += blockDim.x) {
    const int j = idx / K, k = idx - j * K;
    W4[idx] = make_float4(a.W[(k * 4 + 0) * M + j], a.W[(k * 4 + 1) * M + j],
                          a.W[(k * 4 + 2) * M + j], a.W[(k * 4 + 3) * M + j]);
  }
  __syncthreads();
  const float c0 = a.c[0], c1 = a.c[1], c2 = a.c[2], c3 = a.c[3];
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
        const Philox4 r = philox4x32(a.rng.seq_offset + (uint32_t)nn, (uint32_t)(p >> 2),
                                        rng_word2(a.kind, 0, 0, 0), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
        const float t = u01(philox_pick(r, p & 3)) * sum;
        const int l = (t >= e0) + (t >= e0 + e1) + (t >= (e0 + e1) + e2);
        a.sample[o] = l == 0 ? 1.f : 0.f; a.sample[o + a.L] = l == 1 ? 1.f : 0.f;
        a.sample[o + 2 * a.L] = l == 2 ? 1.f : 0.f; a.sample[o + 3 * a.L] = l == 3 ? 1.f : 0.f;
      }
    }
  }
}


// The same for an alphabet of A != 4 letters: a thread keeps the A activations of its position in LDS
// (yl[letter][thread]); filters straight from global memory (an API / test pass).
struct VghAnyArgs {
  VghArgs g;
  int32_t A;
};

__global__ void __launch_bounds__(256) vgh_dense_any_kernel(VghAnyArgs aa) {
  HIP_DYNAMIC_SHARED(float, smem);
  const VghArgs& a = aa.g;
  const int K = a.K, M = a.M, A = aa.A, CH = (int)blockDim.x, tid = (int)threadIdx.x;
  float* yl = smem;   // [A][CH]
  const long items = (long)a.n * a.L;
  for (long i0 = (long)blockIdx.x * CH; i0 < items; i0 += (long)gridDim.x * CH) {
    const long i = i0 + tid;
    if (i >= items) continue;
    const int nn = (int)(i / a.L), p = (int)(i - (long)nn * a.L);
    for (int al = 0; al < A; ++al) yl[al * CH + tid] = a.c[al];
    const int jlo = max(0, p - a.Lh + 1), jhi = min(M - 1, p);
    for (int k = 0; k < K; ++k) {
      const float* hrow = a.hid + ((size_t)nn * K + k) * a.Lh;
      const float* hprow = a.hidp ? a.hidp + ((size_t)nn * K + k) * a.Lh : nullptr;
      for (int j = jlo; j <= jhi; ++j) {
        const float hv = hrow[p - j];
        const float hp = hprow ? hprow[p - j] : 0.f;
        for (int al = 0; al < A; ++al) {
          float y = fmaf(a.W[((size_t)k * A + al) * M + j], hv, yl[al * CH + tid]);
          if (hprow) y = fmaf(a.W[((size_t)k * A + (A - 1 - al)) * M + (M - 1 - j)], hp, y);   // rc(W)[k,a,j] = W[k,A-1-a,M-1-j]
          yl[al * CH + tid] = y;
        }
      }
    }
    const size_t o = (size_t)nn * A * a.L + p;
    float mx = yl[tid];
    for (int al = 0; al < A; ++al) {
      const float y = yl[al * CH + tid];
      if (a.act) a.act[o + (size_t)al * a.L] = y;
      mx = fmaxf(mx, y);
    }
    float sum = 0.f;
    for (int al = 0; al < A; ++al) {
      const float e = __expf(yl[al * CH + tid] - mx);
      yl[al * CH + tid] = e;
      sum += e;
    }
    if (a.prob) {
      const float inv = 1.0f / sum;
      for (int al = 0; al < A; ++al) a.prob[o + (size_t)al * a.L] = yl[al * CH + tid] * inv;
    }
    if (a.sample) {
      const Philox4 r = philox4x32(a.rng.seq_offset + (uint32_t)nn, (uint32_t)(p >> 2),
                                      rng_word2(a.kind, 0, 0, 0), a.rng.step, a.rng.seed_lo, a.rng.seed_hi);
      const float t = u01(philox_pick(r, p & 3)) * sum;
      float cum = 0.f;
      int l = 0;
      for (int al = 0; al < A - 1; ++al) {
        cum += yl[al * CH + tid];
        l += t >= cum ? 1 : 0;
      }
      for (int al = 0; al < A; ++al) a.sample[o + (size_t)al * a.L] = l == al ? 1.f : 0.f;
    }
  }
}

#include "crbm_kernels_generic.h"   // models the LDS-resident kernels do not take: more motifs, longer motifs, other alphabets

// fixed-point sums over the position chunks of a sequence (hit_summary_body) -> mean over positions
__global__ void hit_finalize_kernel(const unsigned long long* fx, float* out, size_t count, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (float)((double)fx[i] * (double)scale);
}

// One wave that does nothing for `ticks` of the GPU's wall clock: put in front of a partition's first chain launch
// so that the partitions' launches interleave from the start (crbm_api.hip, launch_gibbs_parts)
__global__ void __launch_bounds__(64) delay_kernel(unsigned long long ticks) {
  const uint64_t t0 = realtime_ticks();
  while (realtime_ticks() - t0 < ticks) short_sleep();
}

__global__ void __launch_bounds__(1024) reduce_partials_kernel(ReduceArgs a) { reduce_partials_body<false>(a); }

// IPC all-reduce: the column reduction writes this rank's sums straight into its slot of every rank's buffer,
// and the block that arrives last raises the rank's flag in all of them -- the all-reduce then costs no launch of
// its own (the stores are system-scope and drained before a block's ticket; the last block's fence orders them
// before the flags)
struct PublishTail {
  uint32_t* ticket;      // zero before the launch, left zero
  uint32_t* flag[8];     // this rank's flag word of the step's parity in every rank's buffer
  uint32_t value;
  PushTargets push;
};
__global__ void __launch_bounds__(1024) reduce_publish_pair_kernel(ReducePair p, PublishTail t) {
  reduce_partials_body<true>(p.half[blockIdx.y], &t.push);
  wait_vector_memory();
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t prev = ticket_add(t.ticket);
    if (prev == gridDim.x * gridDim.y - 1u) {
      atomicExch(t.ticket, 0u);
      fence_system();
      for (int r = 0; r < t.push.n; ++r) store_system(t.flag[r], t.value);
    }
  }
}

__global__ void __launch_bounds__(1024) reduce_partials_pair_kernel(ReducePair p) {
  reduce_partials_body<false>(p.half[blockIdx.y]);
}

__global__ void apply_update_kernel(UpdateArgs a) { apply_update_body<0>(a, nullptr, true, PlainSums{a.sums}); }   // one block, in place

// IPC all-reduce: this rank's sums of the step (formed in d_sums) -> its slot of every rank's buffer, then its
// flags (one block)
struct PublishArgs {
  const float* src;
  float* dst[8];         // this rank's slot of the step's parity in every rank's buffer
  uint32_t* flag[8];
  uint32_t value;
  int32_t count, n;
};
__global__ void __launch_bounds__(1024) publish_sums_kernel(PublishArgs a) {
  for (int i = threadIdx.x; i < a.count; i += blockDim.x) {
    const float v = a.src[i];
    for (int r = 0; r < a.n; ++r) store_system(a.dst[r] + i, v);
  }
  wait_vector_memory();                             // this wave's (write-through) stores have left ...
  __syncthreads();                                  // ... and so have everybody's
  if (threadIdx.x == 0) {
    fence_system();                                 // one release for the block (a fence per thread costs microseconds)
    for (int r = 0; r < a.n; ++r) store_system(a.flag[r], a.value);
  }
}
#endif  // CRBM_DEFINE_MISC_KERNELS

}  // namespace crbm
