// rollout_plugin.hip -- the fused roll-out for dimension sets without a built-in instantiation (host code only).
//
// The reference's dimensions are constants edited in its scripts (duffing.py:66 Nlift, :632-633 MPCHorizon; the MATLAB twin runs
// L = 10, N = 10: Koopman_update.m:67, 70, 113).  libkoopmpc.so carries rollout_kernel<L, N, q, ...> for the sets BASELINE.json and
// the reference's scripts use; for any other set the kernel is made when a handle of that set is created:
//
//   key = (L, N, q, trajectories per workgroup, lift variant, panel type)
//   1. the process's table of loaded plug-ins,
//   2. the kernel cache on disk -- $KMPC_KERNEL_CACHE, <library directory>/kernel_cache (what __graft_entry__.build() pre-builds
//      travels with the tree), ~/.cache/koopmpc, /tmp/koopmpc-<uid> -- file rollout_L.._N.._q.._nw.._ks.._f64|f32_<hash>.so, the hash over
//      the sources and the compiler flags, so that a changed header never meets a stale object,
//   3. hipcc on csrc/rollout_jit.hip (the sources ship next to the library) with the flags of the library's own build, 4-8 s per kernel,
//      under a file lock (the ranks of a node build an object once), written under a temporary name and renamed,
//   then dlopen.  The plug-in has no undefined symbol of the library; its entry point gets the launch arguments and the workgroup size
//   the library chose.  A set that cannot be served (no compiler, no sources, no writable cache) leaves the handle on per-step
//   launches and says why (kmpc_rollout_plugin_status).
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "kernels.h"

extern char** environ;

namespace kmpc {

namespace {

struct Loaded {
  rollout_plugin_fn fn = nullptr;
  std::string path, how;
  double build_s = 0.0;
};
std::mutex g_mu;
typedef std::tuple<int, int, int, int, int, int, int> KeyTuple;
KeyTuple tuple_of(const RolloutPluginKey& k) { return std::make_tuple(k.L, k.N, k.q, k.nw, k.ks, k.io32, k.term); }
std::map<KeyTuple, Loaded> g_loaded;
std::map<KeyTuple, std::string> g_failed;  // (a set that failed once is not compiled again and again)

const char* const kFlags[] = {"-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-Wno-pass-failed", "-Wno-unused-function", "-shared"};
const char* const kSources[] = {"rollout_jit.hip", "rollout_kernel.hip", "step_body.h", "step_v2.h", "qp_rl.h", "kernels.h", "plant_device.h", "dare_device.h"};

std::string lib_dir() {
  Dl_info info{};
  if (!dladdr(reinterpret_cast<const void*>(&rollout_plugin_dims), &info) || !info.dli_fname) return "";
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
bool read_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}
// FNV-1a over the sources and the flags: the name of a cached object says what it was made of
bool source_hash(const std::string& src_dir, unsigned long long* h, std::string* err) {
  static std::mutex mu;
  static std::map<std::string, unsigned long long> memo;
  std::lock_guard<std::mutex> lk(mu);
  auto it = memo.find(src_dir);
  if (it != memo.end()) { *h = it->second; return true; }
  unsigned long long x = 1469598103934665603ull;
  auto mix = [&](const char* p, size_t n) { for (size_t i = 0; i < n; ++i) { x ^= (unsigned char)p[i]; x *= 1099511628211ull; } };
  for (const char* s : kSources) {
    std::string body;
    if (!read_file(src_dir + "/" + s, &body)) { *err = "kernel source " + src_dir + "/" + s + " not found (the plug-in sources ship next to libkoopmpc.so)"; return false; }
    mix(body.data(), body.size());
  }
  for (const char* f : kFlags) mix(f, strlen(f));
  const int abi = KMPC_PLUGIN_ABI;
  mix(reinterpret_cast<const char*>(&abi), sizeof(abi));
  memo[src_dir] = x;
  *h = x;
  return true;
}
bool is_file(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
bool mkdir_p(const std::string& p) {
  if (p.empty()) return false;
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') {
      if (!cur.empty() && cur != "/") { if (mkdir(cur.c_str(), 0755) != 0 && errno != EEXIST) return false; }
    }
    if (i < p.size()) cur.push_back(p[i]);
  }
  return access(p.c_str(), W_OK | X_OK) == 0;
}
std::vector<std::string> cache_dirs() {
  std::vector<std::string> d;
  if (const char* e = getenv("KMPC_KERNEL_CACHE")) { if (*e) d.push_back(e); }
  const std::string ld = lib_dir();
  if (!ld.empty()) d.push_back(ld + "/kernel_cache");
  if (const char* x = getenv("XDG_CACHE_HOME")) { if (*x) d.push_back(std::string(x) + "/koopmpc"); }
  if (const char* h = getenv("HOME")) { if (*h) d.push_back(std::string(h) + "/.cache/koopmpc"); }
  d.push_back("/tmp/koopmpc-" + std::to_string((long)getuid()));
  return d;
}
std::string hipcc_path() {
  if (const char* e = getenv("KMPC_HIPCC")) { if (*e) return e; }
  if (access("/opt/rocm/bin/hipcc", X_OK) == 0) return "/opt/rocm/bin/hipcc";
  return "hipcc";
}
std::string tail_of(const std::string& path, size_t n) {
  std::string body;
  read_file(path, &body);
  return body.size() > n ? body.substr(body.size() - n) : body;
}

// hipcc as a child process (posix_spawn: the calling process, which may have initialised the GPU, is not replaced), output into `log`
bool run_hipcc(const std::vector<std::string>& args, const std::string& log, std::string* err) {
  std::vector<char*> argv;
  for (const auto& a : args) argv.push_back(const_cast<char*>(a.c_str()));
  argv.push_back(nullptr);
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
  pid_t pid = 0;
  const int rc = posix_spawnp(&pid, argv[0], &fa, nullptr, argv.data(), environ);
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0) { *err = std::string("could not start ") + argv[0] + ": " + strerror(rc); return false; }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) {
    if (errno != EINTR) { *err = std::string("waitpid: ") + strerror(errno); return false; }
  }
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
    *err = std::string(argv[0]) + " failed (" + (WIFEXITED(status) ? "exit code " + std::to_string(WEXITSTATUS(status)) : std::string("signal")) + "): ..." + tail_of(log, 600);
    return false;
  }
  return true;
}

// measurement aid (KMPC_DEBUG only): extra compiler flags for the plug-ins, e.g. -DSOME_SWITCH=1 -- part of the object's hash, so that
// two settings are two objects (tools/dbg/ab_plugin.sh alternates them on one box; with KMPC_FORCE_PLUGIN the built-in sets run on plug-ins)
std::vector<std::string> extra_flags() {
  std::vector<std::string> out;
  const char* e = dbg_env("KMPC_PLUGIN_FLAGS");
  if (!e) return out;
  std::string cur;
  for (const char* p = e;; ++p) {
    if (*p == ' ' || *p == 0) { if (!cur.empty()) out.push_back(cur); cur.clear(); if (!*p) break; }
    else cur.push_back(*p);
  }
  return out;
}
std::string object_name(const RolloutPluginKey& k, unsigned long long h) {
  for (const auto& f : extra_flags())
    for (char ch : f) { h ^= (unsigned char)ch; h *= 1099511628211ull; }
  char buf[160];
  snprintf(buf, sizeof(buf), "rollout_L%d_N%d_q%d_nw%d_ks%s%d_%s%s_%016llx.so", k.L, k.N, k.q, k.nw, k.ks < 0 ? "m" : "", k.ks < 0 ? -k.ks : k.ks,
           k.io32 ? "f32" : "f64", k.term ? "_term" : "", h);
  return buf;
}

bool load_object(const std::string& path, Loaded* out, std::string* err) {
  void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!lib) { *err = std::string("dlopen ") + path + ": " + dlerror(); return false; }
  typedef int (*int_fn)(void);
  const int_fn abi = reinterpret_cast<int_fn>(dlsym(lib, "kmpc_rollout_plugin_abi"));
  const int_fn nb = reinterpret_cast<int_fn>(dlsym(lib, "kmpc_rollout_plugin_args_bytes"));
  const rollout_plugin_fn fn = reinterpret_cast<rollout_plugin_fn>(dlsym(lib, "kmpc_rollout_plugin_launch"));
  if (!abi || !nb || !fn || abi() != KMPC_PLUGIN_ABI || nb() != (int)sizeof(RolloutArgs<double>)) {
    dlclose(lib);
    *err = path + " is not a roll-out plug-in of this library version";
    return false;
  }
  out->fn = fn;
  out->path = path;
  return true;
}

}  // namespace

// Dimension sets a plug-in can be generated for: what rollout_kernel's two step bodies cover with one wave per trajectory --
//   y = C x with q <= 2 output rows: the register-state step (step_v2.h) for L + 2 <= 32, N <= 32; the LDS step (step_body.h) beyond,
//   y = psi (q = L): the 8 x 8-grid step of step_body.h,
// as far as sixteen / eight / four trajectories fit into a CU's LDS (the caller checks rollout_waves) and n = 2 (the plants).
bool rollout_plugin_dims(int n, int L, int N, int q) {
  if (n != 2 || L < 2 || L > 64 || N < 2 || N > 64 || q < 1) return false;
  if (q != L && q > 2) return false;
  if ((long)(L + 1) * (L + 1) > 2048 || (long)N * N > 2048) return false;  // (beyond: four waves per trajectory, the per-step kernels)
  return true;
}

rollout_plugin_fn rollout_plugin_get(const RolloutPluginKey& k, std::string* err, bool build_if_missing) {
  const auto key = tuple_of(k);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_loaded.find(key);
  if (it != g_loaded.end()) return it->second.fn;
  auto fi = g_failed.find(key);
  if (fi != g_failed.end()) { if (err) *err = fi->second; return nullptr; }
  std::string e;
  auto fail = [&](const std::string& msg) -> rollout_plugin_fn {
    g_failed[key] = msg;
    if (err) *err = msg;
    return nullptr;
  };
  const std::string ld = lib_dir();
  if (ld.empty()) return fail("roll-out plug-in: cannot locate libkoopmpc.so (dladdr)");
  const std::string src = ld + "/csrc";
  unsigned long long h = 0;
  if (!source_hash(src, &h, &e)) return fail("roll-out plug-in: " + e);
  const std::string name = object_name(k, h);
  const std::vector<std::string> dirs = cache_dirs();
  Loaded L;
  for (const auto& d : dirs) {
    const std::string p = d + "/" + name;
    if (is_file(p)) {
      if (load_object(p, &L, &e)) {
        L.how = "loaded from the kernel cache";
        g_loaded[key] = L;
        return L.fn;
      }
    }
  }
  if (!build_if_missing) { if (err) *err = "roll-out plug-in " + name + " is not in the kernel cache"; return nullptr; }
  std::string dir;
  for (const auto& d : dirs)
    if (mkdir_p(d)) { dir = d; break; }
  if (dir.empty()) return fail("roll-out plug-in: no writable kernel cache directory (set KMPC_KERNEL_CACHE)");
  const std::string obj = dir + "/" + name, lock = obj + ".lock", log = obj + ".log." + std::to_string((long)getpid());
  const int lfd = open(lock.c_str(), O_CREAT | O_RDWR, 0644);
  if (lfd >= 0) (void)flock(lfd, LOCK_EX);  // (the ranks of a node: one of them builds, the others find the object when they get the lock)
  const auto t0 = std::chrono::steady_clock::now();
  bool built = false;
  if (!is_file(obj)) {
    const std::string tmp = obj + ".tmp." + std::to_string((long)getpid());
    std::vector<std::string> args;
    args.push_back(hipcc_path());
    for (const char* f : kFlags) args.push_back(f);
    args.push_back("-DKMPC_JIT_L=" + std::to_string(k.L));
    args.push_back("-DKMPC_JIT_N=" + std::to_string(k.N));
    args.push_back("-DKMPC_JIT_Q=" + std::to_string(k.q));
    args.push_back("-DKMPC_JIT_NW=" + std::to_string(k.nw));
    args.push_back("-DKMPC_JIT_KS=" + std::to_string(k.ks));
    args.push_back("-DKMPC_JIT_IO32=" + std::to_string(k.io32 ? 1 : 0));
    args.push_back("-DKMPC_JIT_TERM=" + std::to_string(k.term ? 1 : 0));
    for (const auto& f : extra_flags()) args.push_back(f);
    args.push_back("-I" + src);
    args.push_back(src + "/rollout_jit.hip");
    args.push_back("-o");
    args.push_back(tmp);
    const bool ok = run_hipcc(args, log, &e);
    if (ok && rename(tmp.c_str(), obj.c_str()) != 0) { e = "rename " + tmp + ": " + strerror(errno); }
    if (!ok || !e.empty()) {
      (void)unlink(tmp.c_str());
      (void)unlink(log.c_str());
      if (lfd >= 0) { (void)unlink(lock.c_str()); (void)flock(lfd, LOCK_UN); close(lfd); }
      return fail("roll-out plug-in " + name + ": " + e);
    }
    (void)unlink(log.c_str());
    built = true;
  }
  if (lfd >= 0) { (void)unlink(lock.c_str()); (void)flock(lfd, LOCK_UN); close(lfd); }
  if (!load_object(obj, &L, &e)) return fail("roll-out plug-in: " + e);
  L.build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  char hb[96];
  snprintf(hb, sizeof(hb), built ? "compiled with hipcc in %.1f s" : "built by another process of this node (waited %.1f s)", L.build_s);
  L.how = hb;
  g_loaded[key] = L;
  return L.fn;
}

std::string rollout_plugin_describe(const RolloutPluginKey& k) {
  const auto key = tuple_of(k);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_loaded.find(key);
  if (it != g_loaded.end()) return "plug-in " + it->second.path + " (" + it->second.how + ")";
  auto fi = g_failed.find(key);
  if (fi != g_failed.end()) return "unavailable: " + fi->second;
  return "plug-in not loaded yet";
}

}  // namespace kmpc
