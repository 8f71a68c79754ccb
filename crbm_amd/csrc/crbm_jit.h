// Per-model kernel specialisation with hiprtc.
//
// The reference compiles its Theano graph when a CRBM is constructed
// (convRBM.py:175, :453-515).  The MI355X-native counterpart: crbm_create
// instantiates the kernel templates of crbm_kernels.h for this model's
// Cfg<K,M,DS,G>, compiles them for gfx950 with hiprtc and loads the code
// object.  Code objects are cached on disk next to the library
// (csrc/jit_cache/crbm_<hash>.hsaco, keyed by the full source text and
// options), so a configuration is compiled once per source revision;
// __graft_entry__.build() pre-populates the cache for the BASELINE configs.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace crbm {

struct JitKernels {
  hipModule_t module = nullptr;
  hipFunction_t build_tables = nullptr, update_tables = nullptr, update_tables_ipc = nullptr, hgv = nullptr, gibbs = nullptr /* dense top-down; null if the model has none */,
                build_gather_solo = nullptr /* gather table of the solo letter grouping */, slab_hgv = nullptr, slab_tables = nullptr, slab_stats_data = nullptr, slab_stats_model = nullptr, slab_fe = nullptr /* the model as a slab of a larger one: empty beyond 64 motifs */,
                gibbs_sparse = nullptr, gibbs_sparse_stats = nullptr /* empty unless Cfg::FUSE_STATS */, train_local = nullptr /* ditto */, stats_mfma_data = nullptr, stats_mfma_model = nullptr,
                free_energy = nullptr, hit_summary = nullptr;
  bool from_cache = false;
  std::string cache_file;
};

inline std::string jit_dirname(const std::string& p) {
  const size_t pos = p.find_last_of('/');
  return pos == std::string::npos ? std::string(".") : p.substr(0, pos);
}

// Directory holding crbm_kernels.h / crbm_layout.h: next to this shared library.
inline std::string jit_source_dir() {
  if (const char* e = getenv("CRBM_KERNEL_SRC_DIR")) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&jit_source_dir), &info) && info.dli_fname) return jit_dirname(info.dli_fname);
  return ".";
}

inline bool jit_read_file(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

inline uint64_t jit_fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char ch : s) {
    h ^= ch;
    h *= 1099511628211ull;
  }
  return h;
}

// gibbs_wpe: waves per SIMD the launch geometry can actually reach when that is
// below 4 (LDS-limited large models): tells the register allocator not to squeeze
// the kernel for an occupancy it will never see; 0 = no hint.
// gibbs_tb: block-size bound of the chain kernels (256, 512 or 1024: large models share one table copy per CU)
// GS: letters per gather-table group of the plain chain kernel (crbm_gibbs_sparse), G of everything else
// slab: the module also carries the kernels that run the model as a slab of a larger one (crbm_slab_*; models of up to 64 motifs):
// compiled for the shadow handles of generic models only, so that an ordinary model does not pay for them at crbm_create
inline std::string jit_stub(int K, int M, int DS, int G, int GS, int POOL, int gibbs_wpe, int gibbs_tb = 256, bool slab = false) {
  char attr[96] = "", sattr[96] = "";
  if (gibbs_wpe > 0) snprintf(attr, sizeof(attr), "__attribute__((amdgpu_waves_per_eu(1, %d)))", gibbs_wpe);
  else snprintf(attr, sizeof(attr), "__attribute__((amdgpu_waves_per_eu(4)))");   // 4 blocks of 4 waves per CU: at most 128 registers
  // the variant that also carries the statistics accumulators must keep the occupancy of the plain
  // kernel's geometry (4 blocks of 4 waves per CU for small models): cap it at 128 registers there
  if (gibbs_wpe > 0) snprintf(sattr, sizeof(sattr), "__attribute__((amdgpu_waves_per_eu(1, %d)))", gibbs_wpe);
  else snprintf(sattr, sizeof(sattr), "__attribute__((amdgpu_waves_per_eu(4)))");
  char buf[16384];
  snprintf(buf, sizeof(buf),
           "#define CRBM_SLAB_KERNELS %d\n"
           "#include \"crbm_kernels.h\"\n"
           "#ifndef CRBM_GIBBS_ATTR\n#define CRBM_GIBBS_ATTR %s\n#endif\n"
           "#ifndef CRBM_GIBBS_STATS_ATTR\n#define CRBM_GIBBS_STATS_ATTR %s\n#endif\n"
           "#ifndef CRBM_STATS_BYTE_LUT\n#define CRBM_STATS_BYTE_LUT true\n#endif\n"
           "#ifndef CRBM_FUSED_TB\n#define CRBM_FUSED_TB 256\n#endif\n"
           "using ModelCfg = crbm::Cfg<%d, %d, %d, %d, %d>;\n"
           "using SoloCfg = crbm::Cfg<%d, %d, %d, %d, %d>;\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_build_tables(crbm::TablesArgs a) { crbm::build_tables_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_build_gather_solo(crbm::TablesArgs a) { crbm::build_gather_table_body<SoloCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(1024) crbm_update_tables(crbm::UpdateTablesArgs a) { crbm::update_tables_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(1024) crbm_update_tables_ipc(crbm::UpdateIpcArgs a) { crbm::update_tables_ipc_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_hgv(crbm::HgvArgs a) { crbm::hgv_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_slab_hgv(crbm::SlabHgvArgs a) { if constexpr (CRBM_SLAB_KERNELS && ModelCfg::K <= 64) crbm::slab_hgv_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_slab_tables(crbm::SlabTablesArgs a) { if constexpr (CRBM_SLAB_KERNELS && ModelCfg::K <= 64) crbm::slab_tables_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(%d) CRBM_GIBBS_ATTR crbm_gibbs_sparse(crbm::GibbsArgs a) { crbm::gibbs_body<SoloCfg, true>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(CRBM_FUSED_TB) CRBM_GIBBS_STATS_ATTR crbm_gibbs_sparse_stats(crbm::GibbsArgs a) { if constexpr (ModelCfg::FUSE_STATS) crbm::gibbs_body<ModelCfg, true, true>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(CRBM_FUSED_TB) CRBM_GIBBS_STATS_ATTR crbm_train_local(crbm::TrainLocalArgs a) { crbm::train_local_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(%d) crbm_gibbs(crbm::GibbsArgs a) { if constexpr (ModelCfg::DENSE) crbm::gibbs_body<ModelCfg, ModelCfg::DENSE ? false : true>(a); }\n"
           "using RoleData = crbm::StatsRole<ModelCfg, true>;\nusing RoleModel = crbm::StatsRole<ModelCfg, false>;\n"
           "extern \"C\" __global__ void __launch_bounds__(RoleData::THREADS) crbm_stats_mfma_data(crbm::StatsMfmaArgs a) { crbm::stats_mfma_body<ModelCfg, true, CRBM_STATS_BYTE_LUT>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(RoleModel::THREADS) crbm_stats_mfma_model(crbm::StatsMfmaArgs a) { crbm::stats_mfma_body<ModelCfg, false, CRBM_STATS_BYTE_LUT>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(RoleData::THREADS) crbm_slab_stats_data(crbm::SlabStatsArgs a) { if constexpr (CRBM_SLAB_KERNELS && ModelCfg::K <= 64) crbm::slab_stats_body<ModelCfg, true, CRBM_STATS_BYTE_LUT>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(RoleModel::THREADS) crbm_slab_stats_model(crbm::SlabStatsArgs a) { if constexpr (CRBM_SLAB_KERNELS && ModelCfg::K <= 64) crbm::slab_stats_body<ModelCfg, false, CRBM_STATS_BYTE_LUT>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_free_energy(crbm::FeArgs a) { crbm::free_energy_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_slab_fe(crbm::SlabFeArgs a) { if constexpr (CRBM_SLAB_KERNELS && ModelCfg::K <= 64) crbm::slab_fe_body<ModelCfg>(a); }\n"
           "extern \"C\" __global__ void __launch_bounds__(256) crbm_hit_summary(crbm::HitArgs a) { crbm::hit_summary_body<ModelCfg>(a); }\n",
           slab ? 1 : 0, attr, sattr, K, M, DS, G, POOL, K, M, DS, GS, POOL, gibbs_tb, gibbs_tb);
  return buf;
}

// Compile (or fetch from the cache) the code object; no device needed.
inline int jit_compile(int K, int M, int DS, int G, int GS, int POOL, int gibbs_wpe, int gibbs_tb, std::vector<char>* code, bool* from_cache,
                       std::string* cache_file, std::string* err, bool slab = false) {
  const std::string dir = jit_source_dir();
  std::string kernels, layout;
  if (!jit_read_file(dir + "/crbm_kernels.h", &kernels) || !jit_read_file(dir + "/crbm_layout.h", &layout)) {
    *err = "kernel sources not found in " + dir + " (set CRBM_KERNEL_SRC_DIR)";
    return -1;
  }
  const std::string stub = jit_stub(K, M, DS, G, GS, POOL, gibbs_wpe, gibbs_tb, slab);
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-I" + dir, "-I/opt/rocm/include"};
  if (const char* e = getenv("CRBM_JIT_DEFINES")) {   // tuning knobs, e.g. "-DCRBM_STATS_MAX_TILES=16"
    std::istringstream ss(e);
    std::string tok;
    while (ss >> tok) opts.push_back(tok);
  }
  int rtc_major = 0, rtc_minor = 0;
  hiprtcVersion(&rtc_major, &rtc_minor);
  uint64_t h = jit_fnv1a(stub);
  h = jit_fnv1a(kernels, h);
  h = jit_fnv1a(layout, h);
  for (const auto& o : opts)
    if (o.rfind("-I", 0) != 0) h = jit_fnv1a(o, h);
  h = jit_fnv1a(std::to_string(rtc_major) + "." + std::to_string(rtc_minor), h);
  std::string cdir = dir + "/jit_cache";
  if (const char* e = getenv("CRBM_JIT_CACHE")) cdir = e;
  char name[64];
  snprintf(name, sizeof(name), "/crbm_%016llx.hsaco", (unsigned long long)h);
  const std::string cfile = cdir + name;
  if (cache_file) *cache_file = cfile;
  std::string cached;
  if (!getenv("CRBM_JIT_NOCACHE") && jit_read_file(cfile, &cached) && cached.size() > 64) {
    code->assign(cached.begin(), cached.end());
    *from_cache = true;
    return 0;
  }
  *from_cache = false;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, stub.c_str(), "crbm_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *err = "hiprtcCreateProgram failed";
    return -1;
  }
  std::vector<const char*> copts;
  for (const auto& o : opts) copts.push_back(o.c_str());
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)copts.size(), copts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    *err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log;
    hiprtcDestroyProgram(&prog);
    return -1;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->resize(cs);
  hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  // best-effort cache write (atomic rename); failures are not errors
  mkdir(cdir.c_str(), 0755);
  const std::string tmp = cfile + "." + std::to_string((long)getpid()) + ".tmp";
  {
    std::ofstream f(tmp, std::ios::binary);
    if (f) {
      f.write(code->data(), (std::streamsize)code->size());
      f.close();
      if (rename(tmp.c_str(), cfile.c_str()) != 0) unlink(tmp.c_str());
    }
  }
  return 0;
}

inline int jit_load(int K, int M, int DS, int G, int GS, int POOL, int gibbs_wpe, int gibbs_tb, JitKernels* out, std::string* err, bool slab = false) {
  std::vector<char> code;
  if (jit_compile(K, M, DS, G, GS, POOL, gibbs_wpe, gibbs_tb, &code, &out->from_cache, &out->cache_file, err, slab) != 0) return -1;
  hipError_t e = hipModuleLoadData(&out->module, code.data());
  if (e != hipSuccess) {
    *err = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
    return -1;
  }
  struct { const char* name; hipFunction_t* f; } syms[] = {
      {"crbm_build_tables", &out->build_tables}, {"crbm_build_gather_solo", &out->build_gather_solo}, {"crbm_update_tables", &out->update_tables}, {"crbm_update_tables_ipc", &out->update_tables_ipc}, {"crbm_hgv", &out->hgv}, {"crbm_slab_hgv", &out->slab_hgv}, {"crbm_slab_tables", &out->slab_tables}, {"crbm_slab_stats_data", &out->slab_stats_data}, {"crbm_slab_stats_model", &out->slab_stats_model}, {"crbm_slab_fe", &out->slab_fe}, {"crbm_gibbs", &out->gibbs},
      {"crbm_gibbs_sparse", &out->gibbs_sparse}, {"crbm_gibbs_sparse_stats", &out->gibbs_sparse_stats}, {"crbm_train_local", &out->train_local},
      {"crbm_stats_mfma_data", &out->stats_mfma_data},
      {"crbm_stats_mfma_model", &out->stats_mfma_model}, {"crbm_free_energy", &out->free_energy},
      {"crbm_hit_summary", &out->hit_summary}};
  for (auto& s : syms) {
    e = hipModuleGetFunction(s.f, out->module, s.name);
    if (e != hipSuccess) {
      *err = std::string("hipModuleGetFunction(") + s.name + "): " + hipGetErrorString(e);
      return -1;
    }
  }
  return 0;
}

template <typename Args>
inline hipError_t jit_launch(hipFunction_t f, const Args& args, unsigned gx, unsigned gy, unsigned block, unsigned lds,
                             hipStream_t stream) {
  Args copy = args;
  void* params[] = {&copy};
  if (lds > 48 * 1024) {
    // opt-in for large dynamic LDS where the runtime wants one; module functions do not need it on
    // ROCm and the call may fail for them -- its error must not linger for a later hipGetLastError().
    // Once per function and size: the call costs microseconds of host time, a training step has three launches.
    static std::mutex mu;
    static std::unordered_map<hipFunction_t, unsigned> opted;
    std::lock_guard<std::mutex> lock(mu);
    unsigned& have = opted[f];
    if (lds > have) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipGetLastError();
      have = lds;
    }
  }
  return hipModuleLaunchKernel(f, gx, gy, 1, block, 1, 1, lds, stream, params, nullptr);
}

}  // namespace crbm
