// rware_jit.cpp — see rware_jit.h.  Host code only; hipRTC is reached through dlopen / dlsym, nothing here links against it.
#include "rware_jit.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "rware_hooks.h"
#include "rware_jit_sources.inc"

namespace rw_jit {
namespace {

// the handful of hipRTC entry points, by their C signatures (hiprtc.h): every one returns 0 on success
struct Rtc {
    void *lib = nullptr;
    int (*CreateProgram)(void **prog, const char *src, const char *name, int n_headers, const char **headers, const char **names) = nullptr;
    int (*CompileProgram)(void *prog, int n_opts, const char **opts) = nullptr;
    int (*AddNameExpression)(void *prog, const char *expr) = nullptr;
    int (*GetLoweredName)(void *prog, const char *expr, const char **lowered) = nullptr;
    int (*GetProgramLogSize)(void *prog, size_t *n) = nullptr;
    int (*GetProgramLog)(void *prog, char *log) = nullptr;
    int (*GetCodeSize)(void *prog, size_t *n) = nullptr;
    int (*GetCode)(void *prog, char *code) = nullptr;
    int (*DestroyProgram)(void **prog) = nullptr;
    std::string why;
};

Rtc *rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch-ROCm bundles one next to its HIP runtime) wins; then the system's
        // (RWARE_JIT_LIBRARY: load exactly this file instead — tests point it at a path that does not exist to take the
        //  "no hipRTC on this box" road, which has to end in the generic kernel, not in a crash)
        const char *names[] = {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"};
        const char *only = rw_hook("RWARE_JIT_LIBRARY");
        if (only && *only) {
            r.lib = dlopen(only, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *n : names) {
                r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
                if (r.lib) break;
            }
            for (size_t i = 0; !r.lib && i < sizeof names / sizeof names[0]; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) {
            const char *e = dlerror();  // (ONE call: dlerror() hands the message out once and returns NULL after that)
            r.why = std::string("hipRTC not available: ") + (e ? e : "dlopen failed");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.lib, n);
            if (!p) { ok = false; r.why = std::string("hipRTC lacks ") + n; }
            return p;
        };
        r.CreateProgram = reinterpret_cast<decltype(r.CreateProgram)>(sym("hiprtcCreateProgram"));
        r.CompileProgram = reinterpret_cast<decltype(r.CompileProgram)>(sym("hiprtcCompileProgram"));
        r.AddNameExpression = reinterpret_cast<decltype(r.AddNameExpression)>(sym("hiprtcAddNameExpression"));
        r.GetLoweredName = reinterpret_cast<decltype(r.GetLoweredName)>(sym("hiprtcGetLoweredName"));
        r.GetProgramLogSize = reinterpret_cast<decltype(r.GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        r.GetProgramLog = reinterpret_cast<decltype(r.GetProgramLog)>(sym("hiprtcGetProgramLog"));
        r.GetCodeSize = reinterpret_cast<decltype(r.GetCodeSize)>(sym("hiprtcGetCodeSize"));
        r.GetCode = reinterpret_cast<decltype(r.GetCode)>(sym("hiprtcGetCode"));
        r.DestroyProgram = reinterpret_cast<decltype(r.DestroyProgram)>(sym("hiprtcDestroyProgram"));
        if (!ok) { dlclose(r.lib); r.lib = nullptr; }
    });
    return &r;
}

std::string join(const char *const *pieces) {
    std::string s;
    for (; *pieces; ++pieces) s += *pieces;
    return s;
}

uint64_t fnv1a(const std::string &s, uint64_t h) {
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ULL; }
    return h;
}

// Where compiled code objects are kept: $RWARE_JIT_CACHE, else ~/.cache/rware_amd/jit.  "" = no disk cache: without HOME
// there is no directory that is this user's alone, and a code object read back from a shared one (/tmp) would be somebody
// else's code running in this process — so such a process compiles every time and keeps nothing.
std::string cache_dir() {
    if (const char *e = getenv("RWARE_JIT_CACHE")) return e;
    const char *home = getenv("HOME");
    if (!home || !*home) return "";
    return std::string(home) + "/.cache/rware_amd/jit";
}

void mkdirs(const std::string &path) {
    for (size_t i = 1; i <= path.size(); ++i)
        if (i == path.size() || path[i] == '/') mkdir(path.substr(0, i).c_str(), 0700);  // (new directories: this user's only)
}

// the cache directory is trusted only if it belongs to this user and nobody else may write to it
bool dir_is_private(const std::string &dir) {
    struct stat st;
    if (stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

// cache file: "RWJIT1\n<step name>\n<rollout name>\n" + code object
bool cache_read(const std::string &file, Result *out) {
    FILE *f = fopen(file.c_str(), "rb");
    if (!f) return false;
    std::vector<char> all;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + n);
    fclose(f);
    size_t pos = 0;
    auto line = [&](std::string *dst) {
        size_t e = pos;
        while (e < all.size() && all[e] != '\n') ++e;
        if (e >= all.size()) return false;
        dst->assign(all.data() + pos, e - pos);
        pos = e + 1;
        return true;
    };
    std::string magic;
    if (!line(&magic) || magic != "RWJIT1" || !line(&out->step_name) || !line(&out->rollout_name) || pos >= all.size()) return false;
    out->code.assign(all.begin() + (long)pos, all.end());
    return true;
}

void cache_write(const std::string &file, const Result &r) {
    const std::string tmp = file + "." + std::to_string((long)getpid()) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    fprintf(f, "RWJIT1\n%s\n%s\n", r.step_name.c_str(), r.rollout_name.c_str());
    const bool ok = fwrite(r.code.data(), 1, r.code.size(), f) == r.code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), file.c_str()) != 0) unlink(tmp.c_str());  // (atomic: several ranks may compile the same shape)
}

}  // namespace

namespace {
// the two name expressions of a shape and the cache file its code object is kept in
struct Names { std::string expr_step, expr_roll, dir, file; };
Names names_of(const Shape &s, const char *arch) {
    char cfg[256], expr_step[512], expr_roll[512];
    auto static_cfg = [&](int nt) {
        snprintf(cfg, sizeof cfg, "rw::StaticCfg<%d, %d, %d, %d, %d, %d, %d, %d, %d, %uu, %d, %d>", s.H, s.W, s.N, s.Q, s.S, s.E, s.T, s.M, s.NL,
                 s.layers, s.directional, nt);
        return cfg;
    };
    const char *cell = s.wide ? "uint16_t" : "uint8_t";
    snprintf(expr_step, sizeof expr_step, "rw::rware_step_kernel<%d, %s, %s, false, %d>", s.R, cell, static_cfg(s.nt), s.obs);
    snprintf(expr_roll, sizeof expr_roll, "rw::rware_step_kernel<%d, %s, %s, true, %d>", s.R, cell, static_cfg(0), s.obs);
    const std::string opts_key = std::string(arch) + " -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16" + (s.stats ? " -DRW_STATS_BUILD=1" : "");
    const std::string key = std::string(kJitSourcesSha) + "|" + opts_key + "|" + expr_step + "|" + expr_roll;
    char name[64];
    snprintf(name, sizeof name, "%016llx%016llx.hsaco", (unsigned long long)fnv1a(key, 0xcbf29ce484222325ULL),
             (unsigned long long)fnv1a(key, 0x84222325cbf29ce4ULL));
    Names n;
    n.expr_step = expr_step; n.expr_roll = expr_roll; n.dir = cache_dir();
    n.file = n.dir.empty() ? std::string() : n.dir + "/" + name;
    return n;
}
}  // namespace

std::string cache_file(const Shape &s, const char *arch) { return names_of(s, arch).file; }

bool compile(const Shape &s, const char *arch, Result *out) {
    const Names nm = names_of(s, arch);
    const char *expr_step = nm.expr_step.c_str(), *expr_roll = nm.expr_roll.c_str();
    const std::string &dir = nm.dir, &file = nm.file;
    const char *nocache = rw_hook("RWARE_JIT_NO_CACHE");
    const bool use_cache = !dir.empty() && !(nocache && nocache[0] == '1');
    if (use_cache && dir_is_private(dir) && cache_read(file, out)) {
        out->from_cache = true;
        out->log = "loaded " + file;
        return true;
    }
    Rtc *r = rtc();
    if (!r->lib) { out->log = r->why; return false; }
    // the program: fixed-width integer names (hipRTC has no <stdint.h>; the headers skip their system includes under
    // __HIPCC_RTC__), the kernel header, nothing else — the two kernels are instantiated through name expressions
    std::string src =
        "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long uint64_t;\n"
        "typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long int64_t; typedef unsigned long uintptr_t;\n"
        "#include \"rware_kernels.h\"\n";
    std::vector<std::string> bodies;
    std::vector<const char *> hdr, hname;
    for (int i = 0; i < kJitHeaderCount; ++i) bodies.push_back(join(kJitHeaderPieces[i]));
    for (int i = 0; i < kJitHeaderCount; ++i) { hdr.push_back(bodies[(size_t)i].c_str()); hname.push_back(kJitHeaderNames[i]); }
    const auto t0 = std::chrono::steady_clock::now();
    void *prog = nullptr;
    if (r->CreateProgram(&prog, src.c_str(), "rware_jit.hip", kJitHeaderCount, hdr.data(), hname.data()) != 0) {
        out->log = "hiprtcCreateProgram failed";
        return false;
    }
    bool ok = r->AddNameExpression(prog, expr_step) == 0 && r->AddNameExpression(prog, expr_roll) == 0;
    const std::string arch_opt = std::string("--offload-arch=") + arch;
    const char *opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-DRW_STATS_BUILD=1"};
    ok = ok && r->CompileProgram(prog, s.stats ? 6 : 5, opts) == 0;
    size_t ln = 0;
    if (r->GetProgramLogSize(prog, &ln) == 0 && ln > 1) {
        std::string log(ln, '\0');
        if (r->GetProgramLog(prog, &log[0]) == 0) out->log = log.c_str();
    }
    const char *ls = nullptr, *lr = nullptr;
    ok = ok && r->GetLoweredName(prog, expr_step, &ls) == 0 && r->GetLoweredName(prog, expr_roll, &lr) == 0 && ls && lr;
    size_t cn = 0;
    ok = ok && r->GetCodeSize(prog, &cn) == 0 && cn > 0;
    if (ok) {
        out->step_name = ls;
        out->rollout_name = lr;
        out->code.resize(cn);
        ok = r->GetCode(prog, out->code.data()) == 0;
    }
    r->DestroyProgram(&prog);
    out->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) {
        out->log = std::string("hipRTC could not build ") + expr_step + ": " + out->log;
        out->code.clear();
        return false;
    }
    char note[160];
    snprintf(note, sizeof note, "compiled in %.2f s", out->compile_seconds);
    out->log = note;
    if (use_cache) {
        mkdirs(dir);
        if (dir_is_private(dir)) {
            cache_write(file, *out);
            out->log += " -> " + file;
        } else {
            out->log += " (NOT CACHED — every construction of this shape recompiles: " + dir + " is not a private directory of this user; chmod go-w it or point RWARE_JIT_CACHE elsewhere)";
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "librware_hip: run-time specialised kernels are not cached: %s is not a private directory of this user (group- or world-writable, or not yours)\n", dir.c_str());
        }
    } else {
        out->log += " (not cached: no HOME / RWARE_JIT_CACHE)";
    }
    return true;
}

}  // namespace rw_jit
