// Development tool (not part of the product): what it costs a training launch to add its blocks' partial sums into ONE
// row of 64-bit fixed-point accumulators with device-scope atomics (deterministic: integer adds commute) instead of
// storing a row per block for a column-reduction launch.   hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.hip -o tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// mode 0: a row of floats per block; 1: u64 atomics into one row; 2: u64 atomics into `spread` rows (block % spread)
__global__ void __launch_bounds__(256) producer(float* rows, unsigned long long* acc, int ncols, int mode, int spread, int spin, float* sink) {
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = fmaf(x, 1.0000001f, 0.5f);
  if (sink && x == 12345.f) sink[0] = x;
  if (mode == 0) {
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) rows[(size_t)blockIdx.x * ncols + c] = x + (float)c;
  } else {
    unsigned long long* dst = acc + (size_t)(mode == 2 ? blockIdx.x % spread : 0) * ncols;
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) atomicAdd(dst + c, (unsigned long long)(c + 1) + (x == 12345.f ? 1ull : 0ull));   // depends on the work before it
  }
}

__global__ void __launch_bounds__(1024) reduce_rows(const float* rows, int nrows, int ncols, float* out) {
  __shared__ float red[1024];
  // 16 columns x 64 row slices per block
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
  float s = 0.f;
  if (col < ncols)
    for (int r = sl; r < nrows; r += 64) s += rows[(size_t)r * ncols + col];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 512; h >= 16; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x < 16 && col < ncols) out[col] = red[threadIdx.x];
}

__global__ void __launch_bounds__(1024) consume_acc(unsigned long long* acc, int ncols, int spread, float* out) {
  for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
    unsigned long long s = 0;
    for (int r = 0; r < spread; ++r) { s += acc[(size_t)r * ncols + c]; acc[(size_t)r * ncols + c] = 0; }
    out[c] = (float)s;
  }
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1792, ncols = argc > 2 ? atoi(argv[2]) : 1536, spin = argc > 3 ? atoi(argv[3]) : 4000;
  float *rows, *out, *sink;
  unsigned long long* acc;
  CHK(hipMalloc(&rows, (size_t)blocks * ncols * 4)); CHK(hipMalloc(&out, ncols * 4)); CHK(hipMalloc(&sink, 4));
  CHK(hipMalloc(&acc, (size_t)64 * ncols * 8)); CHK(hipMemset(acc, 0, (size_t)64 * ncols * 8));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int n = 400;
  auto timeit = [&](const char* what, auto&& step) {
    for (int i = 0; i < 50; ++i) step();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) step();
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %.2f us per step\n", what, 1e3 * ms / n);
  };
  printf("%d blocks x 256 threads, %d columns, spin %d\n", blocks, ncols, spin);
  timeit("producer alone, no output", [&] { hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, 0, 0, 1, spin, sink); });
  timeit("producer storing rows", [&] { hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 0, 1, spin, sink); });
  timeit("producer storing rows + column reduction launch", [&] {
    hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 0, 1, spin, sink);
    hipLaunchKernelGGL(reduce_rows, dim3((ncols + 15) / 16), dim3(1024), 0, 0, rows, blocks, ncols, out);
  });
  timeit("  ... + a consumer launch (one block)", [&] {
    hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 0, 1, spin, sink);
    hipLaunchKernelGGL(reduce_rows, dim3((ncols + 15) / 16), dim3(1024), 0, 0, rows, blocks, ncols, out);
    hipLaunchKernelGGL(consume_acc, dim3(1), dim3(1024), 0, 0, acc, ncols, 1, out);
  });
  timeit("producer with u64 atomics into one row", [&] { hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 1, 1, spin, sink); });
  timeit("  ... + a consumer launch (one block)", [&] {
    hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 1, 1, spin, sink);
    hipLaunchKernelGGL(consume_acc, dim3(1), dim3(1024), 0, 0, acc, ncols, 1, out);
  });
  for (int spread : {4, 16, 64}) {
    char what[96];
    snprintf(what, sizeof what, "producer with u64 atomics into %d rows (block %% %d)", spread, spread);
    timeit(what, [&] { hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 2, spread, spin, sink); });
    snprintf(what, sizeof what, "  ... + a consumer launch adding the %d rows", spread);
    timeit(what, [&] {
      hipLaunchKernelGGL(producer, dim3(blocks), dim3(256), 0, 0, rows, acc, ncols, 2, spread, spin, sink);
      hipLaunchKernelGGL(consume_acc, dim3(1), dim3(1024), 0, 0, acc, ncols, spread, out);
    });
  }
  return 0;
}
