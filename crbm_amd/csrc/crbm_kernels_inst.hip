// One translation unit per instantiated motif-quad count: compiled with
// -DCRBM_NQ=<n> (see build.py).  Exposes the launchers through a KernelTable.
#include "crbm_kernels.h"

#ifndef CRBM_NQ
#error "compile with -DCRBM_NQ=<n>"
#endif

#define CRBM_CAT2(a, b) a##b
#define CRBM_CAT(a, b) CRBM_CAT2(a, b)

namespace crbm {
namespace {

template <typename K>
void allow_big_lds(K kernel, uint32_t lds) {
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
}

void launch_hgv(const HgvArgs& a, const LaunchCfg& c) {
  allow_big_lds(hgv_kernel<CRBM_NQ>, c.lds);
  hipLaunchKernelGGL(hgv_kernel<CRBM_NQ>, dim3(c.gx, c.gy), dim3(c.block), c.lds, c.stream, a);
}
void launch_gibbs(const GibbsArgs& a, const LaunchCfg& c) {
  allow_big_lds(gibbs_kernel<CRBM_NQ>, c.lds);
  hipLaunchKernelGGL(gibbs_kernel<CRBM_NQ>, dim3(c.gx, c.gy), dim3(c.block), c.lds, c.stream, a);
}
void launch_stats(const StatsArgs& a, const LaunchCfg& c) {
  allow_big_lds(stats_kernel<CRBM_NQ>, c.lds);
  hipLaunchKernelGGL(stats_kernel<CRBM_NQ>, dim3(c.gx, c.gy), dim3(c.block), c.lds, c.stream, a);
}
void launch_fe(const FeArgs& a, const LaunchCfg& c) {
  allow_big_lds(free_energy_kernel<CRBM_NQ>, c.lds);
  hipLaunchKernelGGL(free_energy_kernel<CRBM_NQ>, dim3(c.gx, c.gy), dim3(c.block), c.lds, c.stream, a);
}

}  // namespace

// accessor (not a const global: hipcc would try to emit that on the device side too)
const KernelTable* CRBM_CAT(kernel_table_nq, CRBM_NQ)() {
  static KernelTable t = {CRBM_NQ, launch_hgv, launch_gibbs, launch_stats, launch_fe};
  return &t;
}

}  // namespace crbm
