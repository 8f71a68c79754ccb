// Instruction-throughput microbenchmarks for gfx950 (development tool, not part
// of the product): how many cycles one wave64 instruction of each kind costs a
// SIMD when 8 waves per SIMD issue independent chains.  Used to price the
// Philox rounds, the exp/rcp pipeline and LDS gathers of the Gibbs kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void __launch_bounds__(256) ubench(uint32_t* out, int iters, uint32_t k) {
  uint32_t x[8];
  float f[8];
  uint64_t q[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 2654435761u + i * 977u + k; f[i] = 1.0f + (float)(x[i] & 1023) * 1e-4f; q[i] = x[i]; }
  float kf = 1.0f + (float)k * 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#define S(i) \
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(kf)); \
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i+1)&7])); \
      if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 3) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 4) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q[i]) : "v"(x[i]), "v"(k) : "vcc"); \
      if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i])); \
      if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i])); \
      if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(k) : "vcc"); \
      if (OP == 9) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 10) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 11) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 12) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(x[i])); \
      if (OP == 13) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(x[i]) : "v"(k)); \
      if (OP == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(kf)); \
      if (OP == 15) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(k)); \
      if (OP == 16) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(k)); \
      if (OP == 17) asm volatile("v_log_f32 %0, %0" : "+v"(f[i])); \
      if (OP == 18) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(f[i]), "v"(kf) : "vcc"); \
      if (OP == 19) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x[i])); \
      if (OP == 20) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i+1)&7])); \
      if (OP == 21) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i+1)&7]));
      REP8(S)
#undef S
    }
  }
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= x[i] ^ __float_as_uint(f[i]) ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// LDS gather: every lane reads `NQ` float4 from a random row (row stride = rs floats)
template <int NQ, int WIDE>
__global__ void __launch_bounds__(256) lds_gather(float* out, int iters, int rows, int rs, uint32_t seed) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < rows * rs; i += 256) lds[i] = (float)i;
  __syncthreads();
  uint32_t r = threadIdx.x * 2654435761u + seed;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    r = r * 1664525u + 1013904223u;
    const int row = (r >> 10) % rows;
    if (WIDE) {
      const float4* p = reinterpret_cast<const float4*>(lds + row * rs);
#pragma unroll
      for (int q = 0; q < NQ; ++q) { float4 t = p[q]; acc += t.x + t.y + t.z + t.w; }
    } else {
#pragma unroll
      for (int q = 0; q < 4 * NQ; ++q) acc += lds[row * rs + q];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
double run(const char* name, int blocks_per_cu, int iters) {
  int ncu = 256;
  const int blocks = ncu * blocks_per_cu;
  uint32_t* out;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(ubench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 12345u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(ubench<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per SIMD = (blocks_per_cu * 4 waves / 4 SIMDs) * iters * 32
  const double winst_per_simd = (double)blocks_per_cu * iters * 32.0;
  const double ns_per_winst = ms * 1e6 / winst_per_simd;
  printf("%-16s blocks/CU=%d  %8.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name,
         blocks_per_cu, ms, ns_per_winst, ns_per_winst * 2.4);
  CHK(hipFree(out));
  return ns_per_winst;
}

template <int NQ, int WIDE>
void run_lds(const char* name, int rows, int rs, int iters) {
  const int blocks = 256 * 4;
  float* out;
  CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const size_t lds = (size_t)rows * rs * 4;
  hipLaunchKernelGGL((lds_gather<NQ, WIDE>), dim3(blocks), dim3(256), lds, 0, out, iters / 10, rows, rs, 1u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL((lds_gather<NQ, WIDE>), dim3(blocks), dim3(256), lds, 0, out, iters, rows, rs, 1u);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes_per_cu = 4.0 * 256 * (double)iters * NQ * 16;   // 4 blocks/CU
  printf("%-28s rows=%d stride=%d floats: %8.3f ms  %.1f B/clk/CU @2.4GHz (%.2f ns per row-gather per wave)\n", name, rows, rs,
         ms, bytes_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e6 / ((double)iters * 4.0));
  CHK(hipFree(out));
}

int main() {
  const int it = 4000;
  for (int bpc : {8, 2}) {
    run<0>("v_fma_f32", bpc, it);
    run<14>("v_add_f32", bpc, it);
    run<1>("v_pk_fma_f32", bpc, it);
    run<20>("v_pk_add_f32", bpc, it);
    run<21>("v_pk_mul_f32", bpc, it);
    run<5>("v_xor_b32", bpc, it);
    run<9>("v_xad_u32", bpc, it);
    run<11>("v_add3_u32", bpc, it);
    run<16>("v_lshl_add_u32", bpc, it);
    run<12>("v_bfe_u32", bpc, it);
    run<13>("v_alignbit_b32", bpc, it);
    run<8>("v_cndmask_b32", bpc, it);
    run<18>("v_cmp_gt_f32", bpc, it);
    run<19>("v_ffbl_b32", bpc, it);
    run<2>("v_mul_lo_u32", bpc, it);
    run<3>("v_mul_hi_u32", bpc, it);
    run<4>("v_mad_u64_u32", bpc, it);
    run<10>("v_mul_u32_u24", bpc, it);
    run<15>("v_mad_u32_u24", bpc, it);
    run<6>("v_exp_f32", bpc, it);
    run<7>("v_rcp_f32", bpc, it);
    run<17>("v_log_f32", bpc, it);
  }
  run_lds<3, 1>("ds_read_b128 x3 random row", 64, 12, 4000);
  run_lds<3, 1>("ds_read_b128 x3 random row", 64, 16, 4000);
  run_lds<3, 1>("ds_read_b128 x3 random row", 64, 20, 4000);
  run_lds<3, 1>("ds_read_b128 x3 random row", 16, 12, 4000);
  run_lds<3, 1>("ds_read_b128 x3 random row", 256, 12, 4000);
  run_lds<3, 0>("ds_read_b32 x12 random row", 64, 12, 4000);
  run_lds<1, 1>("ds_read_b128 x1 random row", 64, 4, 4000);
  run_lds<5, 1>("ds_read_b128 x5 random row", 64, 20, 4000);
  return 0;
}
