// Development tool (not part of the product): checks the lane -> element maps of
// v_mfma_f32_16x16x32_f16 on gfx950 with exact integer data before the statistics
// kernel relies on them (the guide documents the bf16 form; "other dtypes: check").
//   A fragment: lane l holds A[row l&15][k = 8*(l>>4) + e], e = 0..7
//   B fragment: lane l holds B[k = 8*(l>>4) + e][col l&15]
//   C/D:        lane l, register r holds D[row 4*(l>>4) + r][col l&15]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D) {   // A[16][32], B[32][16], D[16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)A[r * 32 + 8 * g + e];
    b[e] = (_Float16)B[(8 * g + e) * 16 + r];
  }
  floatx4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = c[q];
}

// throughput of the statistics inner loop shape: per 32 positions 4 A fragments (LUT reads) x 2 B
// fragments (hi/lo) = 8 MFMAs + the f16 split of 8 floats
__global__ void __launch_bounds__(256) loop(const float* src, float* out, int iters) {
  __shared__ float4 lut[256];
  __shared__ float pt[16 * 264];
  for (int i = threadIdx.x; i < 256; i += 256) lut[i] = make_float4(1.f * i, 2.f, 3.f, 4.f);
  for (int i = threadIdx.x; i < 16 * 264; i += 256) pt[i] = src[i % 1024];
  __syncthreads();
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  floatx4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const float4 p0 = *reinterpret_cast<const float4*>(&pt[r * 264 + ((8 * g + 32 * it) & 255)]);
    const float4 p1 = *reinterpret_cast<const float4*>(&pt[r * 264 + ((8 * g + 32 * it) & 255) + 4]);
    const float x[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    half8 hi, lo;
    for (int e = 0; e < 8; ++e) {
      hi[e] = (_Float16)x[e];
      lo[e] = (_Float16)(x[e] - (float)hi[e]);
    }
    for (int t = 0; t < 4; ++t) {
      const float4 f = lut[(it * 7 + t * 13 + l) & 255];
      half8 a = __builtin_bit_cast(half8, f);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, hi, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, lo, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  std::vector<float> A(16 * 32), B(32 * 16), D(256), ref(256, 0.f);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 2 + j * 11) % 9 - 4 + (j == 3));
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) ref[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
  float *dA, *dB, *dD;
  CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dD, 1 << 22));
  CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  CHK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += D[i] != ref[i];
  printf("mfma_f32_16x16x32_f16 layout: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  // loop timing
  float* src; CHK(hipMalloc(&src, 4096)); CHK(hipMemset(src, 0, 4096));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int iters = 2000, grid = 1024;
  hipLaunchKernelGGL(loop, dim3(grid), dim3(256), 0, 0, src, dD, iters);
  CHK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(loop, dim3(grid), dim3(256), 0, 0, src, dD, iters);
  CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double pos = (double)grid * 4 * iters * 32;      // positions processed (32 per wave-iteration)
  printf("stats-shaped loop: %.3f ms, %.2f ns per 1000 positions, %.1f cycles/position/SIMD at 2.4 GHz\n",
         ms, ms * 1e6 / (pos / 1000), ms * 1e-3 * 2.4e9 / (pos / 1024));
  return bad ? 1 : 0;
}
