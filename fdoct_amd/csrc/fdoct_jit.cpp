// fdoct_jit.cpp -- the wave-per-row kernel for ANY shape, compiled when a handle first needs it.
//
// fdoct_wave.hip compiles wave_kernel<W, M, N, ...> for the shapes the reference ships and their neighbours
// (FDOCT_WAVE_SHAPES*).  An operator who types another region of interest, bin factor or numfftpoints into the ini
// (build/BscanFFT.ini:9-12, 25-26, 31-32, 51-52) would drop to the workgroup-per-row kernel, 2.5-5x slower
// (tools/bench_jit.py).  Unless fdoct_set_jit(h, 0) / FDOCT_JIT=0 says otherwise, the same template -- its source travels
// inside this library, fdoct_wave_src.inc -- is instantiated for the handle's own (W, M, N, sample type, depth) by hipRTC
// instead: one compile of under a second, kept in a process-wide table and on disk, then the compile-time-specialised
// kernel like a built-in one.
// libhiprtc is loaded on first use only (dlopen): a host that never asks for it does not depend on it.
#include "fdoct_jit.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "fdoct_kernels.h"
#include "fdoct_wave.h"
#include "fdoct_wave_src.inc"

namespace fdoct {

namespace {

struct Rtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcAddNameExpression) add_name = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetLoweredName) lowered = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    err.clear();  // (a library that was missing is looked for again: wave_jit_get decides how often)
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      err = "libhiprtc.so not found";
      return false;
    }
    auto sym = [&](const char* n) { return dlsym(lib, n); };
#define RTC_SYM(field, name)                                  \
  field = reinterpret_cast<decltype(field)>(sym(#name));      \
  if (!field) err = "libhiprtc.so lacks " #name;
    RTC_SYM(create, hiprtcCreateProgram)
    RTC_SYM(destroy, hiprtcDestroyProgram)
    RTC_SYM(add_name, hiprtcAddNameExpression)
    RTC_SYM(compile, hiprtcCompileProgram)
    RTC_SYM(log_size, hiprtcGetProgramLogSize)
    RTC_SYM(log, hiprtcGetProgramLog)
    RTC_SYM(lowered, hiprtcGetLoweredName)
    RTC_SYM(code_size, hiprtcGetCodeSize)
    RTC_SYM(code, hiprtcGetCode)
    RTC_SYM(version, hiprtcVersion)
#undef RTC_SYM
    if (!err.empty()) {
      dlclose(lib);
      lib = nullptr;
      return false;
    }
    return true;
  }
};

struct Entry {
  hipFunction_t fn = nullptr;
  std::string why;          // non-empty: this shape failed
  bool permanent = true;    // false: a failure of the environment (no libhiprtc, no memory for the module), tried again ...
  std::chrono::steady_clock::time_point retry_at{};  // ... from here on
};

std::mutex g_mu;
Rtc g_rtc;
std::map<std::tuple<int, int, int, int, int, int, int, int>, Entry> g_kernels;   // W, M, N, sample type, depth bins per lane, depth bound, opt, device

const char* sample_type(int kdtype) {
  switch (kdtype) {
    case FDOCT_K_U8: return "unsigned char";
    case FDOCT_K_U16: return "unsigned short";
    case FDOCT_K_F32: return "float";
    default: return nullptr;
  }
}

uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}
uint64_t fnv1a(uint64_t h, const std::string& s) { return fnv1a(h, s.data(), s.size() + 1); }

std::string cache_dir() {
  if (const char* e = std::getenv("FDOCT_JIT_CACHE")) return *e ? std::string(e) : std::string();  // empty: no disk cache
  if (const char* e = std::getenv("XDG_CACHE_HOME"))
    if (*e) return std::string(e) + "/fdoct_amd";
  if (const char* e = std::getenv("HOME"))
    if (*e) return std::string(e) + "/.cache/fdoct_amd";
  return std::string();
}

void make_dirs(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); i++)
    if (i == dir.size() || dir[i] == '/') mkdir(dir.substr(0, i).c_str(), i == dir.size() ? 0700 : 0755);  // the cache itself: private
}

// A code object read from the disk runs on the GPU with the caller's rights, so the cache is trusted only where nobody else
// can have put it: the directory must be a real directory (not a symbolic link) that the effective user owns and that neither
// group nor others can write; the same goes for the file.  Anything else and the disk cache is left alone -- not read, not
// written -- and `why_not` says so (the kernel is compiled and kept in the process-wide table as usual).
bool cache_dir_trusted(const std::string& dir, std::string* why_not) {
  struct stat st;
  if (lstat(dir.c_str(), &st) != 0) return true;  // not there yet: make_dirs creates it 0700
  if (!S_ISDIR(st.st_mode)) {
    *why_not = "run-time compile cache " + dir + " is not a directory (or is a symbolic link): not used";
    return false;
  }
  if (st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) {
    *why_not = "run-time compile cache " + dir + " is not owned by this user, or group / others may write to it: not used (chmod go-w, or set FDOCT_JIT_CACHE to a private directory)";
    return false;
  }
  return true;
}

// file = "FDOCTJIT2\n" lowered-name "\n" <code bytes> " " <fnv1a of the code, hex> "\n" code object
bool read_cached(const std::string& path, std::string* lowered, std::vector<char>* code) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) {
    close(fd);
    return false;  // somebody else's file, or one others may rewrite: compiled afresh (and replaced by our own)
  }
  FILE* f = fdopen(fd, "rb");
  if (!f) {
    close(fd);
    return false;
  }
  std::vector<char> all;
  char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + n);
  std::fclose(f);
  static const char magic[] = "FDOCTJIT2\n";
  const size_t ml = sizeof magic - 1;
  if (all.size() < ml + 2 || std::memcmp(all.data(), magic, ml) != 0) return false;
  const char* p0 = all.data() + ml;
  const char* end = all.data() + all.size();
  const char* nl1 = static_cast<const char*>(std::memchr(p0, '\n', (size_t)(end - p0)));
  if (!nl1 || nl1 == p0) return false;
  const char* nl2 = static_cast<const char*>(std::memchr(nl1 + 1, '\n', (size_t)(end - nl1 - 1)));
  if (!nl2 || nl2 - nl1 > 64) return false;
  unsigned long long bytes = 0, sum = 0;
  const std::string meta(nl1 + 1, nl2);
  if (std::sscanf(meta.c_str(), "%llu %llx", &bytes, &sum) != 2) return false;
  if ((unsigned long long)(end - nl2 - 1) != bytes || bytes < 64) return false;        // truncated or padded: not trusted
  if (fnv1a(0xcbf29ce484222325ull, nl2 + 1, (size_t)bytes) != sum) return false;        // damaged: not trusted
  lowered->assign(p0, nl1);
  code->assign(nl2 + 1, end);
  return true;
}

void write_cached(const std::string& dir, const std::string& path, const std::string& lowered, const std::vector<char>& code) {
  make_dirs(dir);
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
  if (!f) {  // a read-only home: the process-wide table still holds the kernel
    if (fd >= 0) close(fd);
    return;
  }
  bool ok = std::fprintf(f, "FDOCTJIT2\n%s\n%llu %llx\n", lowered.c_str(), (unsigned long long)code.size(),
                         (unsigned long long)fnv1a(0xcbf29ce484222325ull, code.data(), code.size())) > 0 &&
            std::fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

hipError_t load_function(const std::vector<char>& code, const std::string& lowered, hipFunction_t* fn, std::string* why) {
  hipModule_t mod = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());  // kept for the life of the process, like the built-in kernels
  if (e != hipSuccess) {
    *why = std::string("hipModuleLoadData: ") + hipGetErrorString(e);
    return e;
  }
  e = hipModuleGetFunction(fn, mod, lowered.c_str());
  if (e != hipSuccess) {
    *why = std::string("hipModuleGetFunction: ") + hipGetErrorString(e);
    (void)hipModuleUnload(mod);
    return e;
  }
  // the workgroup's LDS (shared tables + one buffer per wave) goes beyond the 64 KB a kernel may use unasked
  // (160 KB less the 64 bytes the launch leaves free -- fdoct_route.cpp::launch_family_wave never asks for more --: the kernel's
  // own static LDS, the row-ticket counter, has to fit next to the dynamic part; a failure here must not stay behind as the
  // runtime's "last error" for the launch that follows on the fallback kernel)
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(*fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  if (e != hipSuccess) {
    *why = std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e);
    (void)hipGetLastError();
    (void)hipModuleUnload(mod);
    return e;
  }
  return hipSuccess;
}

// The translation unit, options and name expression of one instantiation.
struct Job {
  std::string tu, arch, expr;
  std::vector<const char*> opts;
};
bool make_job(int W, int M, int N, int kdtype, int D, int opt, const char* gcn_arch, Job* j, std::string* why) {
  const int TD = (D + 63) / 64, DK = wave_depth_bound(N, opt, D);
  const char* st = sample_type(kdtype);
  if (!st) {
    *why = "no wave-per-row kernel for this sample type";
    return false;
  }
  // this library holds gfx950 code only (and hipRTC does not survive an architecture name it does not know)
  if (!gcn_arch || std::strncmp(gcn_arch, "gfx950", 6) != 0 || (gcn_arch[6] != '\0' && gcn_arch[6] != ':')) {
    *why = std::string("not a gfx950 device: ") + (gcn_arch ? gcn_arch : "(null)");
    return false;
  }
  j->arch = std::string("--offload-arch=") + gcn_arch;
  char expr[160];
  std::snprintf(expr, sizeof expr, "fdoct::wave_kernel<%d, %d, %d, %s, %d, %d, %d>", W, M, N, st, TD, opt, DK);
  j->expr = expr;
  // fixed-width names the run-time compiler may lack, then the device code
  j->tu = "typedef unsigned char uint8_t;\ntypedef unsigned short uint16_t;\ntypedef unsigned int uint32_t;\n"
          "typedef decltype(sizeof(0)) size_t;\n#include \"fdoct_wave_dev.h\"\n";
  j->opts = {j->arch.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function"};  // the flags of the Makefile
  // (the rows-per-wave rule sizes the launch on the host: the run-time compiled kernel follows the LIBRARY's setting)
  j->opts.push_back(FDOCT_WAVE_ROWS2 ? "-DFDOCT_WAVE_ROWS2=1" : "-DFDOCT_WAVE_ROWS2=0");
  j->opts.push_back(FDOCT_WAVE_R20PAD ? "-DFDOCT_WAVE_R20PAD=1" : "-DFDOCT_WAVE_R20PAD=0");   // (the buffers' length follows it)
  // tuning aid (tools/ab_jit.sh): extra -D options for the run-time compiled kernels, e.g. FDOCT_JIT_DEFINES="-DFDOCT_WAVE_RESGI=0"
  // (part of the cache key like every option)
  static const std::vector<std::string> extra = [] {
    std::vector<std::string> v;
    if (const char* e = std::getenv("FDOCT_JIT_DEFINES")) {
      std::string cur;
      for (const char* p = e;; p++) {
        if (*p == ' ' || *p == '\0') {
          if (cur.rfind("-D", 0) == 0) v.push_back(cur);
          cur.clear();
          if (!*p) break;
        } else {
          cur.push_back(*p);
        }
      }
    }
    return v;
  }();
  for (const std::string& x : extra) j->opts.push_back(x.c_str());
  return true;
}

hipError_t compile_job(const Job& j, std::string* lowered, std::vector<char>* code, std::string* why) {
  hiprtcProgram prog = nullptr;
  if (g_rtc.create(&prog, j.tu.c_str(), "fdoct_wave_jit.hip", k_jit_src_count, const_cast<const char**>(k_jit_src_texts),
                   const_cast<const char**>(k_jit_src_names)) != HIPRTC_SUCCESS) {
    *why = "hiprtcCreateProgram failed";
    return hipErrorUnknown;
  }
  hipError_t rc = hipSuccess;
  if (g_rtc.add_name(prog, j.expr.c_str()) != HIPRTC_SUCCESS) {
    *why = "hiprtcAddNameExpression failed";
    rc = hipErrorUnknown;
  } else if (g_rtc.compile(prog, (int)j.opts.size(), const_cast<const char**>(j.opts.data())) != HIPRTC_SUCCESS) {
    size_t ls = 0;
    g_rtc.log_size(prog, &ls);
    std::string log(ls ? ls : 1, '\0');
    if (ls) g_rtc.log(prog, &log[0]);
    // the first error line says why (a static_assert of the template: this shape is not one the kernel can take)
    const size_t ep = log.find("error:");
    const size_t ee = ep == std::string::npos ? std::string::npos : log.find('\n', ep);
    *why = std::string("run-time compile of ") + j.expr + " failed: " +
           (ep == std::string::npos ? std::string("(no diagnostic)") : log.substr(ep, ee == std::string::npos ? 200 : ee - ep));
    rc = hipErrorInvalidValue;
  } else {
    const char* low = nullptr;
    size_t cs = 0;
    if (g_rtc.lowered(prog, j.expr.c_str(), &low) != HIPRTC_SUCCESS || !low || g_rtc.code_size(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
      *why = "hiprtc returned no code object";
      rc = hipErrorUnknown;
    } else {
      *lowered = low;
      code->resize(cs);
      if (g_rtc.code(prog, code->data()) != HIPRTC_SUCCESS) {
        *why = "hiprtcGetCode failed";
        rc = hipErrorUnknown;
      }
    }
  }
  g_rtc.destroy(&prog);
  return rc;
}

// *transient: the failure is one of the environment (libhiprtc absent, the module did not load), not of the shape.
hipError_t build(int W, int M, int N, int kdtype, int D, int opt, int device, hipFunction_t* fn, std::string* why, bool* transient,
                 std::string* cache_note) {
  *transient = false;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    *why = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
    return e;
  }
  Job j;
  if (!make_job(W, M, N, kdtype, D, opt, prop.gcnArchName, &j, why)) return hipErrorInvalidValue;
  if (!g_rtc.load()) {
    *why = g_rtc.err;
    *transient = true;
    return hipErrorNotSupported;
  }
  // disk cache: the key covers everything the code object depends on
  int vmaj = 0, vmin = 0;
  g_rtc.version(&vmaj, &vmin);
  uint64_t key = 0xcbf29ce484222325ull;
  key = fnv1a(key, j.tu);
  for (int i = 0; i < k_jit_src_count; i++) key = fnv1a(key, std::string(k_jit_src_texts[i]));
  for (const char* o : j.opts) key = fnv1a(key, std::string(o));
  key = fnv1a(key, j.expr);
  key = fnv1a(key, &vmaj, sizeof vmaj);
  key = fnv1a(key, &vmin, sizeof vmin);
  std::string dir = cache_dir();
  if (!dir.empty() && !cache_dir_trusted(dir, cache_note)) dir.clear();
  char fname[128];
  std::snprintf(fname, sizeof fname, "/wave_%dx%d_%d_t%d_d%d_k%d_o%d_%016llx.co", W, M, N, kdtype, (D + 63) / 64, wave_depth_bound(N, opt, D), opt,
                (unsigned long long)key);
  const std::string path = dir + fname;

  std::string lowered;
  std::vector<char> code;
  if (!dir.empty() && read_cached(path, &lowered, &code)) {
    if (load_function(code, lowered, fn, why) == hipSuccess) return hipSuccess;
    why->clear();  // a stale or damaged file: compile again
  }
  hipError_t rc = compile_job(j, &lowered, &code, why);
  if (rc != hipSuccess) return rc;
  if ((rc = load_function(code, lowered, fn, why)) != hipSuccess) {
    *transient = true;
    return rc;
  }
  if (!dir.empty()) write_cached(dir, path, lowered, code);
  return hipSuccess;
}

}  // namespace

bool wave_jit_shape_ok(int W, int M, int N, int D, int opt) {
  if (W < 2 || M < 1 || N < 4 || D < 1) return false;
  const bool cplx = (opt & FDOCT_WAVE_OPT_CPLX) != 0, deep = (opt & FDOCT_WAVE_OPT_DEEP) != 0;
  const long long mw = (long long)W * M;
  if (mw < 128 || mw >= 65536) return false;                              // two upsampled samples per lane at least; 16-bit gather sources
  if ((M > 1 && W % 2 != 0) || D > N) return false;                       // half-length zero-pad transforms
  if (!cplx && (N % 2 != 0 || (D > N / 2) != deep)) return false;         // real rows: half-length final transform; beyond N/2 only with the mirror option
  if (cplx && deep) return false;
  const int nc = wave_final_points(N, opt);
  if (nc > 16384 || wave_plan(nc).npass <= 0) return false;               // lengths of 2^a 3^b 5^c
  if (M > 1 && (wave_plan(W / 2).npass <= 0 || wave_plan((int)mw / 2).npass <= 0)) return false;
  // at least four waves' buffers next to the shared tables (below that the workgroup-per-row kernel is the better one)
  const size_t priv = wave_private_lds_bytes(W, M, N, opt);
  const size_t shared_floor = ((size_t)nc + (size_t)mw + 2 * (size_t)W + (cplx ? 2 * (size_t)N : 0)) * 4;
  return shared_floor + 4 * priv <= 160 * 1024 - 64;
}

hipError_t wave_jit_get(int W, int M, int N, int kdtype, int D, int opt, int device, hipFunction_t* fn, std::string* why) {
  std::lock_guard<std::mutex> lock(g_mu);
  // (one kernel per depth CLASS: bins per lane and the bound of wave_depth_bound)
  const auto key = std::make_tuple(W, M, N, kdtype, (D + 63) / 64, wave_depth_bound(N, opt, D), opt, device);
  auto it = g_kernels.find(key);
  const auto now = std::chrono::steady_clock::now();
  // a shape the template cannot take is remembered for good; a failure of the environment (libhiprtc missing, the module did
  // not load) is tried again, at most every five seconds
  if (it == g_kernels.end() || (!it->second.fn && !it->second.permanent && now >= it->second.retry_at)) {
    Entry e;
    bool transient = false;
    std::string cache_note;
    if (build(W, M, N, kdtype, D, opt, device, &e.fn, &e.why, &transient, &cache_note) != hipSuccess) {
      e.fn = nullptr;
      if (e.why.empty()) e.why = "run-time compile failed";
      e.permanent = !transient;
      e.retry_at = now + std::chrono::seconds(5);
    } else if (!cache_note.empty()) {
      e.why = cache_note;  // the kernel runs; the note says why the disk cache was left alone
    }
    if (it == g_kernels.end())
      it = g_kernels.emplace(key, e).first;
    else
      it->second = e;
  }
  *why = it->second.why;
  if (!it->second.fn) return hipErrorNotSupported;
  *fn = it->second.fn;
  return hipSuccess;
}

long long wave_jit_compile_only(int W, int M, int N, int kdtype, int D, int opt, const char* gcn_arch, std::string* why) {
  std::lock_guard<std::mutex> lock(g_mu);
  Job j;
  if (!make_job(W, M, N, kdtype, D, opt, gcn_arch, &j, why)) return -1;
  if (!g_rtc.load()) {
    *why = g_rtc.err;
    return -1;
  }
  std::string lowered;
  std::vector<char> code;
  if (compile_job(j, &lowered, &code, why) != hipSuccess) return -1;
  return (long long)code.size();
}

hipError_t wave_jit_launch(hipFunction_t fn, const WaveArgs& a, int grid, int waves, size_t lds, hipStream_t st) {
  WaveArgs copy = a;
  void* args[] = {&copy};
  return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64u * (unsigned)waves, 1, 1, (unsigned)lds, st, args, nullptr);
}

}  // namespace fdoct
