// Development tool (not part of the product): what a dependent back-to-back launch costs on gfx950 as a function of
// its geometry -- blocks, threads per block, dynamic LDS, registers -- with a kernel that does nothing but touch one
// LDS word and (optionally) meet a barrier.   hipcc --offload-arch=gfx950 -O2 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int THREADS>
__global__ void __launch_bounds__(THREADS) empty_kernel(float* out, int barriers) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) lds[0] = 1.f;
  for (int i = 0; i < barriers; ++i) __syncthreads();
  if (out && threadIdx.x == 0 && lds[0] == 2.f) out[blockIdx.x] = 1.f;
}

template <int THREADS>
void run(int blocks, int lds_bytes, int barriers, int n, float* out) {
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel<THREADS>, dim3(blocks), dim3(THREADS), lds_bytes, 0, out, barriers);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel<THREADS>, dim3(blocks), dim3(THREADS), lds_bytes, 0, out, barriers);
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("blocks %5d x %4d threads, LDS %6d B, %d barriers: %.2f us per launch\n", blocks, THREADS, lds_bytes, barriers, 1e3 * ms / n);
}

// the same launches replayed from a graph of `per` kernel nodes captured from the stream
template <int THREADS>
void run_graph(int blocks, int lds_bytes, int per, int n, float* out) {
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < per; ++i) hipLaunchKernelGGL(empty_kernel<THREADS>, dim3(blocks), dim3(THREADS), lds_bytes, st, out, 2);
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipStreamSynchronize(st));
  CHK(hipEventRecord(e0, st));
  for (int i = 0; i < n / per; ++i) CHK(hipGraphLaunch(ge, st));
  CHK(hipEventRecord(e1, st));
  CHK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("graph of %3d nodes: blocks %5d x %4d threads, LDS %6d B: %.2f us per kernel\n", per, blocks, THREADS, lds_bytes, 1e3 * ms / (n / per * per));
  CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); CHK(hipStreamDestroy(st));
}

int main() {
  float* out;
  CHK(hipMalloc(&out, 1 << 20));
  const int n = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    run<64>(1, 0, 0, n, out);
    run<256>(256, 0, 0, n, out);
    run<1024>(256, 0, 0, n, out);
    run<1024>(256, 110 * 1024, 0, n, out);
    run<1024>(256, 110 * 1024, 2, n, out);
    run<1024>(256, 50 * 1024, 2, n, out);
    run<512>(512, 55 * 1024, 2, n, out);
    run<256>(1024, 27 * 1024, 2, n, out);
    run<256>(1792, 22 * 1024, 2, n, out);
    run<512>(256, 110 * 1024, 2, n, out);
    run<768>(256, 110 * 1024, 2, n, out);
  }
  for (int per : {1, 3, 20, 100}) {
    run_graph<1024>(256, 110 * 1024, per, 2000, out);
    run_graph<256>(1792, 22 * 1024, per, 2000, out);
  }
  return 0;
}
