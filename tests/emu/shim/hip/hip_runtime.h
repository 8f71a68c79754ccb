// TEST INFRASTRUCTURE: a minimal "HIP on CPU threads" shim so that the kernel
// header crbm_amd/csrc/crbm_kernels.h can be compiled with g++ and run under
// AddressSanitizer / UBSan (GPU sanitizers are not available on this pool).
// One OS thread per GPU thread, pthread barriers for __syncthreads() and for
// wave-level shuffles.  Never shipped, never used by the product path.
#pragma once
#include <immintrin.h>   // F16C: f16 <-> f32 exactly as the GPU's conversions round (nearest even, subnormals kept)
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static   /* blocks run one at a time in the emulator */

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float4 {
  float x, y, z, w;
};
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct alignas(8) float2 { float x, y; };
struct alignas(8) uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
struct uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
typedef void* hipStream_t;

namespace emu {
struct BlockCtx {
  pthread_barrier_t bar;
  pthread_barrier_t* wave_bar;   // one per wave
  float* wave_scratch;           // 64 floats per wave
  uint32_t* wave_frag;           // 64 lanes x 8 words per wave: the A and B fragments of an emulated MFMA
  unsigned char* smem;
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx* t_ctx;
}  // namespace emu

#define threadIdx emu::t_threadIdx
#define blockIdx emu::t_blockIdx
#define blockDim emu::t_blockDim
#define gridDim emu::t_gridDim

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::t_ctx->smem);

static inline void __syncthreads() { pthread_barrier_wait(&emu::t_ctx->bar); }

static inline float __shfl_down(float v, int off, int width = 64) {
  const unsigned lane = emu::t_threadIdx.x & 63u, wave = emu::t_threadIdx.x >> 6;
  float* s = emu::t_ctx->wave_scratch + wave * 64;
  s[lane] = v;
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  const float r = (lane + (unsigned)off < (unsigned)width) ? s[lane + off] : v;
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  return r;
}

static inline float __shfl_xor(float v, int mask, int width = 64) {
  const unsigned lane = emu::t_threadIdx.x & 63u, wave = emu::t_threadIdx.x >> 6;
  float* s = emu::t_ctx->wave_scratch + wave * 64;
  s[lane] = v;
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  const float r = s[(lane ^ (unsigned)mask) & 63u];
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  (void)width;
  return r;
}

// DPP moves used by crbm_kernels.h: row_ror:n (ctrl 0x120 + n: lane l of a 16-lane row
// reads lane (l - n) mod 16), row_bcast15 (0x142: lane 15 of the previous row) and
// row_bcast31 (0x143: lane 31); rows outside row_mask keep `old`.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const unsigned lane = emu::t_threadIdx.x & 63u, wave = emu::t_threadIdx.x >> 6;
  float* s = emu::t_ctx->wave_scratch + wave * 64;
  memcpy(&s[lane], &src, 4);
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  int r = old;
  const unsigned row = lane >> 4;
  if ((row_mask >> row) & 1) {
    if (ctrl == 0x142) { if (row >= 1) memcpy(&r, &s[16 * row - 1], 4); }
    else if (ctrl == 0x143) { if (row >= 2) memcpy(&r, &s[31], 4); }
    else {
      const unsigned n = (unsigned)ctrl - 0x120u;
      const unsigned from = (lane & ~15u) | ((lane - n) & 15u);
      memcpy(&r, &s[from], 4);
    }
  }
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  (void)bank_mask; (void)bound_ctrl;
  return r;
}

static inline int __builtin_amdgcn_readlane(int v, int src_lane) {
  const unsigned lane = emu::t_threadIdx.x & 63u, wave = emu::t_threadIdx.x >> 6;
  float* s = emu::t_ctx->wave_scratch + wave * 64;
  memcpy(&s[lane], &v, 4);
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  int r;
  memcpy(&r, &s[src_lane], 4);
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  return r;
}

// all 64 lanes of the wave must call it (true for the uses in crbm_kernels.h)
static inline unsigned long long __ballot(int pred) {
  const unsigned lane = emu::t_threadIdx.x & 63u, wave = emu::t_threadIdx.x >> 6;
  float* s = emu::t_ctx->wave_scratch + wave * 64;
  s[lane] = pred ? 1.f : 0.f;
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  unsigned long long m = 0;
  for (unsigned i = 0; i < 64; ++i) m |= (unsigned long long)(s[i] != 0.f) << i;
  pthread_barrier_wait(&emu::t_ctx->wave_bar[wave]);
  return m;
}
// lanes of a wave run in lockstep on the GPU; here they are OS threads
static inline void __builtin_amdgcn_wave_barrier() {
  pthread_barrier_wait(&emu::t_ctx->wave_bar[emu::t_threadIdx.x >> 6]);
}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

namespace emu {
// stand-in for the GPU's constant-rate clock (crbm_kernels.h: realtime_ticks): microseconds
static inline uint64_t realtime_ticks() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000ull + (uint64_t)ts.tv_nsec / 1000ull;
}
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
// x = hi + lo, both f16 (crbm_kernels.h: split_f16)
static inline void split_f16(const float (&x)[8], uint32_t (&hi)[4], uint32_t (&lo)[4]) {
  for (int q = 0; q < 4; ++q) {
    uint16_t h[2], l[2];
    for (int t = 0; t < 2; ++t) {
      h[t] = f2h(x[2 * q + t]);
      l[t] = f2h(x[2 * q + t] - h2f(h[t]));
    }
    hi[q] = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    lo[q] = (uint32_t)l[0] | ((uint32_t)l[1] << 16);
  }
}
// v_mfma_f32_16x16x32_f16: lane l holds A[l&15][8(l>>4)+e], B[8(l>>4)+e][l&15], D[4(l>>4)+r][l&15].
// All 64 lanes of the wave call it.  Products of two f16 are exact in f32; the sum is taken in
// double and rounded once (the hardware's internal order is not specified).
static inline void mfma_16x16x32_f16(const uint32_t (&a)[4], const uint32_t (&b)[4], float (&d)[4]) {
  const unsigned lane = t_threadIdx.x & 63u, wave = t_threadIdx.x >> 6;
  uint32_t* s = t_ctx->wave_frag + (size_t)wave * 64 * 8;
  for (int q = 0; q < 4; ++q) { s[lane * 8 + q] = a[q]; s[lane * 8 + 4 + q] = b[q]; }
  pthread_barrier_wait(&t_ctx->wave_bar[wave]);
  const unsigned col = lane & 15u, rg = lane >> 4;
  for (int r = 0; r < 4; ++r) {
    const unsigned row = 4 * rg + r;
    double acc = d[r];
    for (unsigned kk = 0; kk < 32; ++kk) {
      const unsigned la = row + 16 * (kk >> 3), lb = col + 16 * (kk >> 3), e = kk & 7u;
      const uint16_t ha = (uint16_t)(s[la * 8 + (e >> 1)] >> (16 * (e & 1u)));
      const uint16_t hb = (uint16_t)(s[lb * 8 + 4 + (e >> 1)] >> (16 * (e & 1u)));
      acc += (double)h2f(ha) * (double)h2f(hb);
    }
    d[r] = (float)acc;
  }
  pthread_barrier_wait(&t_ctx->wave_bar[wave]);
}
}  // namespace emu

// Only used to decide whether a wave takes a slow path that is a no-op for lanes
// that do not need it, so the lane-local answer is an exact emulation.
static inline int __any(int pred) { return pred; }

static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicExch(uint32_t* p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
static inline float atomicAdd(float* p, float v) {   // CAS loop: the blocks of a grid run as concurrent OS threads
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), want;
  float f;
  do {
    __builtin_memcpy(&f, &old, 4);
    f += v;
    __builtin_memcpy(&want, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  __builtin_memcpy(&f, &old, 4);
  return f;
}
static inline unsigned int atomicMax(unsigned int* p, unsigned int v) {
  unsigned int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline unsigned int __float_as_uint(float f) { unsigned int u; __builtin_memcpy(&u, &f, 4); return u; }
static inline int __ffs(uint32_t v) { return __builtin_ffs((int)v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __fdividef(float a, float b) { return a / b; }
using std::max;
using std::min;
