// Model-specialised cascade kernels: hiprtc compiles wb_cascade_tile.h -- the SAME source the generic kernels are
// built from -- behind a generated prelude that makes the stage records of ONE model compile-time constants
// (wb_model_specialize).  Feature offsets become LDS instruction offsets, thresholds, leaf values and theta become
// literals, the scalar record loads and the moves in front of the selects disappear from phase A and from the
// wave-synchronous segments.  The generic kernels stay the cold path (and the only path for float32 tiles).
//
// Replaces nothing new of the reference: it is wb_cascade_launch's kernel (model.py:216-259, training.py:84-96)
// for a model that is scanned often enough to be worth ~2 s of compilation.  Compiled code objects are kept in
// the process and in a cache directory ($WB_JIT_CACHE, default ~/.cache/waldboost_amd), keyed by a hash of the
// generated source, the target and the hiprtc version.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE        // dlmopen
#endif
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <stdlib.h>
#include <string.h>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "wb_common.h"

int wb_cascade_group(int depth);   // wb_cascade.hip

namespace {

const char kTileSrc[] =
#include "wb_cascade_tile.inc"
    ;

struct JitKernel {
    hipModule_t module = nullptr;
    hipFunction_t func = nullptr;
    int users = 0;                            // models holding the function (wb_jit_get / wb_jit_release)
    uint64_t released = 0;                    // when the last of them let go (a counter): the oldest idle module goes first
    int scratch = 0;                          // bytes of scratch memory per lane the build asks for
};

std::mutex g_mu;
std::map<uint64_t, JitKernel> g_loaded;      // per process: hash -> loaded module
uint64_t g_release_clock = 0;

// Two compilers can stand behind a specialised kernel:
//   0  the hiprtc the process already holds -- the library links libhiprtc.so.7; under PyTorch that name resolves to the
//      copy its wheel bundles (ROCm 7.0 there), under rocprofv3 to the toolkit's.  The default, and the only one by default;
//   1  the hiprtc of the ROCm toolkit the library was built with (WB_ROCM_LIB_DIR, set by the Makefile from hipcc's
//      location; WB_HIPRTC_LIB=<file> picks another), for WB_JIT_COMPILERS=both / toolkit.  A second libhiprtc +
//      libamd_comgr with the sonames of loaded ones can only live in a link-map namespace of its own: dlmopen(LM_ID_NEWLM),
//      on first use -- which also gives it a private copy of libc, and that is where it crashed in long processes.
// Round 4: compiler 0's code for some cascades of depth-3 trees wrote wrong records on nine scans of ten (tests/
// test_gpu_fuzz.py seeds 558, 569, 644; profiles/r04/jit_selftest.txt) and fails the self-test; compiler 1's code for the
// same source passed 60 of 60.  wb_model_specialize trusts a build only after the self-test, whichever compiler made it.
struct Rtc {
    decltype(&hiprtcCreateProgram) create = &hiprtcCreateProgram;
    decltype(&hiprtcCompileProgram) compile = &hiprtcCompileProgram;
    decltype(&hiprtcGetProgramLogSize) log_size = &hiprtcGetProgramLogSize;
    decltype(&hiprtcGetProgramLog) log = &hiprtcGetProgramLog;
    decltype(&hiprtcGetCodeSize) code_size = &hiprtcGetCodeSize;
    decltype(&hiprtcGetCode) code = &hiprtcGetCode;
    decltype(&hiprtcDestroyProgram) destroy = &hiprtcDestroyProgram;
    decltype(&hiprtcGetErrorString) error_string = &hiprtcGetErrorString;
    decltype(&hiprtcVersion) version = &hiprtcVersion;
    std::string origin = "process";
    bool ok = true;
};

const Rtc &rtc_process() {
    static const Rtc r;
    return r;
}

const Rtc &rtc_toolkit() {
    static const Rtc r = [] {
        Rtc y;
        y.ok = false;
        const char *env = getenv("WB_HIPRTC_LIB");
#ifdef WB_ROCM_LIB_DIR
        const std::string path = env && *env ? std::string(env) : std::string(WB_ROCM_LIB_DIR) + "/libhiprtc.so.7";
#else
        const std::string path = env && *env ? std::string(env) : std::string();
#endif
        y.origin = path;
        if (path.empty()) return y;
        void *h = dlmopen(LM_ID_NEWLM, path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            if (getenv("WB_JIT_VERBOSE")) fprintf(stderr, "[wb_jit] %s: %s\n", path.c_str(), dlerror());
            return y;
        }
        bool ok = true;
        auto sym = [&](auto &fp, const char *name) {
            void *p = dlsym(h, name);
            if (!p) ok = false;
            else fp = reinterpret_cast<std::remove_reference_t<decltype(fp)>>(p);
        };
        sym(y.create, "hiprtcCreateProgram");
        sym(y.compile, "hiprtcCompileProgram");
        sym(y.log_size, "hiprtcGetProgramLogSize");
        sym(y.log, "hiprtcGetProgramLog");
        sym(y.code_size, "hiprtcGetCodeSize");
        sym(y.code, "hiprtcGetCode");
        sym(y.destroy, "hiprtcDestroyProgram");
        sym(y.error_string, "hiprtcGetErrorString");
        sym(y.version, "hiprtcVersion");
        if (!ok && getenv("WB_JIT_VERBOSE")) fprintf(stderr, "[wb_jit] %s lacks a hiprtc entry point\n", path.c_str());
        y.ok = ok;
        return y;
    }();
    return r;
}

const Rtc &rtc_of(int which) { return which == 0 ? rtc_process() : rtc_toolkit(); }

uint64_t fnv1a(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

// the stage segments run_segments walks for a cascade of T stages (wb_cascade_tile.h: t_end = min(2 t, T, t + 64),
// from stage 8 on; the re-count at stage 16 splits nothing: 16 is on the chain)
// Only the segments that end by stage WB_JIT_BAKE_STAGES (default 384) are unrolled with their records as constants: the
// windows of a long soft cascade that get that far are a handful per tile, and a wave that walks them streams through its
// code once -- unrolled, stages 384..1023 would be hundreds of KB of instructions nobody executes twice; they run the
// generic segment loop of the same kernel (records through the scalar cache).
std::string segment_list(int T) {
    static const int cap = getenv("WB_JIT_BAKE_STAGES") ? atoi(getenv("WB_JIT_BAKE_STAGES")) : 384;
    std::string s;
    for (int t = 8; t < T;) {
        int e = 2 * t < T ? 2 * t : T;
        e = e < t + 64 ? e : t + 64;
        if (e > cap && t >= 16) break;
        s += " X(" + std::to_string(t) + "," + std::to_string(e) + ")";
        t = e;
    }
    return s;
}

std::string make_source(const int32_t *words, size_t n_words, int T, int D, int rpw, int waves, int C, int rows, int pitch, int eb = 1, int lds_stages = -1, int occ_min = 8) {
    std::string s;
    s.reserve(n_words * 12 + 1024);
    s += "#define WB_JIT_BAKED 1\n#define WB_JIT_STAGE_WORDS ";
    for (size_t i = 0; i < n_words; ++i) {
        if (i) s += ',';
        s += std::to_string(words[i]);
    }
    s += "\n#define WB_JIT_SEGMENTS(X)" + segment_list(T) + "\n";
    s += "#define WB_JIT_LDS_STAGES " + std::to_string(lds_stages < 0 ? T : lds_stages) + "\n";
    // (the ending and the occupancy floor can be overridden per build through WB_JIT_DEFS: diagnostics)
    s += "#define WB_JIT_WAVES " + std::to_string(waves) + "\n#define WB_CASC_QFULL " + std::to_string((int)WB_CASC_QFULL) +
         "\n#ifndef WB_CASC_END_BARRIER\n#define WB_CASC_END_BARRIER " + std::to_string((int)WB_CASC_END_BARRIER) +
         "\n#endif\n#ifndef WB_JIT_OCC_MIN\n#define WB_JIT_OCC_MIN " + std::to_string(occ_min) + "\n#endif\n#ifndef WB_JIT_ATTR_EXTRA\n#define WB_JIT_ATTR_EXTRA\n#endif\n";
    s += "#define WB_JIT_T " + std::to_string(T) + "\n#define WB_JIT_C " + std::to_string(C) + "\n#define WB_JIT_ROWS " +
         std::to_string(rows) + "\n#define WB_JIT_PITCH " + std::to_string(pitch) + "\n";
    s += "#include \"wb_cascade_tile.h\"\n";
    // (8 waves per SIMD, as the generic kernel reaches on its own with 41 registers: the byte tile admits 4 workgroups per CU)
    s += "extern \"C\" __global__ __launch_bounds__(" + std::to_string(waves * 64) +
         ") __attribute__((" + std::string(occ_min == 8 ? "amdgpu_num_sgpr(80), " : "") + "amdgpu_waves_per_eu(WB_JIT_OCC_MIN, 8) WB_JIT_ATTR_EXTRA)) void wb_casc_jit(CascArgs a, const int32_t *stages) {\n"
         "    cascade_tile_body<" + std::to_string(D) + ", " + std::to_string(rpw) + ", " + std::to_string(waves) +
         ", " + std::to_string(eb) + ", true>(a, stages);\n}\n";
    return s;
}

std::string cache_dir() {
    if (const char *e = getenv("WB_JIT_CACHE")) return *e ? std::string(e) : std::string();
    if (const char *h = getenv("HOME")) return std::string(h) + "/.cache/waldboost_amd";
    return std::string();
}

bool read_file(const std::string &path, std::vector<char> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// The disk cache is bounded: before a new code object is written, the oldest ones (by modification time) go until at most
// WB_JIT_CACHE_MAX - 1 (default 256) are left -- a loop that changes its model every iteration (training with the uint8
// channel functions, threshold sweeps) otherwise leaves one file per model content behind, for ever.
void prune_cache(const std::string &dir) {
    const int cap = getenv("WB_JIT_CACHE_MAX") ? atoi(getenv("WB_JIT_CACHE_MAX")) : 256;
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    std::vector<std::pair<time_t, std::string>> files;
    while (struct dirent *e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() != 19 || n.compare(16, 3, ".co") != 0) continue;          // (only what this cache wrote: <16 hex digits>.co)
        struct stat st;
        if (stat((dir + "/" + n).c_str(), &st) == 0) files.emplace_back(st.st_mtime, n);
    }
    closedir(d);
    if ((int)files.size() < cap) return;
    std::sort(files.begin(), files.end());
    for (size_t i = 0; i + (size_t)(cap > 0 ? cap - 1 : 0) < files.size(); ++i) (void)unlink((dir + "/" + files[i].second).c_str());
}

// a cached file is only handed to the module loader if it looks like what compile() produces: an ELF image of plausible
// size, owned by this user (WB_JIT_CACHE may point at a shared directory)
bool plausible_code_object(const std::string &path, const std::vector<char> &code) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return false;
    return code.size() > 64 && code[0] == 0x7f && code[1] == 'E' && code[2] == 'L' && code[3] == 'F';
}

void write_file_atomic(const std::string &dir, const std::string &path, const std::vector<char> &data) {
    if (dir.empty()) return;
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos && slash > 0) (void)mkdir(dir.substr(0, slash).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    prune_cache(dir);
    const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

// source -> code object for `arch` (no HIP runtime call: works without a GPU)
int compile(const Rtc &R, const std::string &src, const char *arch, std::vector<char> &code, std::string &log) {
    hiprtcProgram prog;
    const char *hdr_src[] = {kTileSrc};
    const char *hdr_name[] = {"wb_cascade_tile.h"};
    hiprtcResult r = R.create(&prog, src.c_str(), "wb_casc_jit.hip", 1, hdr_src, hdr_name);
    if (r != HIPRTC_SUCCESS) {
        log = std::string("hiprtcCreateProgram: ") + R.error_string(r);
        return WB_ERR_HIP;
    }
    const std::string a = std::string("--offload-arch=") + arch;
    std::vector<std::string> extra;                          // diagnostic: WB_JIT_DEFS="-DWB_TAIL_W=1 -D..." (A/B builds)
    if (const char *e = getenv("WB_JIT_DEFS")) {
        std::string cur;
        for (const char *c = e;; ++c) {
            if (*c == ' ' || *c == 0) {
                if (!cur.empty()) extra.push_back(cur);
                cur.clear();
                if (!*c) break;
            } else {
                cur += *c;
            }
        }
    }
    std::vector<const char *> opts = {a.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-pragma-once-outside-header", "-Wno-inline-asm"};
    for (const std::string &x : extra) opts.push_back(x.c_str());
    r = R.compile(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    if (R.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        log.resize(ls);
        (void)R.log(prog, &log[0]);
    }
    if (getenv("WB_JIT_VERBOSE") && !log.empty()) fprintf(stderr, "[wb_jit] %s\n", log.c_str());
    if (r != HIPRTC_SUCCESS) {
        const size_t at = log.find("error:");                     // (warnings first: show the first error)
        if (at != std::string::npos) log = log.substr(at > 120 ? at - 120 : 0);
        log = std::string("hiprtcCompileProgram: ") + R.error_string(r) + "\n" + log;
        (void)R.destroy(&prog);
        return WB_ERR_HIP;
    }
    size_t cs = 0;
    r = R.code_size(prog, &cs);
    if (r == HIPRTC_SUCCESS) {
        code.resize(cs);
        r = R.code(prog, code.data());
    }
    (void)R.destroy(&prog);
    if (const char *d = getenv("WB_JIT_DUMP_DIR")) {            // diagnostic: the generated source and the code object
        const std::string base = std::string(d) + "/wb_casc_jit";
        if (FILE *f = fopen((base + ".hip").c_str(), "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
        if (FILE *f = fopen((base + ".co").c_str(), "wb")) { fwrite(code.data(), 1, code.size(), f); fclose(f); }
    }
    if (r != HIPRTC_SUCCESS || code.empty()) {
        log = std::string("hiprtcGetCode: ") + R.error_string(r);
        return WB_ERR_HIP;
    }
    return WB_OK;
}

// Bytes of scratch (private segment) per work-item the code object's kernel asks for, from its metadata note (msgpack:
// the key ".private_segment_fixed_size" followed by an unsigned integer); -1 when the note cannot be read.
int scratch_bytes(const std::vector<char> &code) {
    static const char key[] = ".private_segment_fixed_size";
    const size_t kl = sizeof(key) - 1;
    for (size_t i = 0; i + kl + 1 <= code.size(); ++i) {
        if (code[i] != key[0] || memcmp(code.data() + i, key, kl) != 0) continue;
        const unsigned char *v = reinterpret_cast<const unsigned char *>(code.data()) + i + kl;
        const size_t left = code.size() - i - kl;
        if (v[0] <= 0x7f) return v[0];
        if (v[0] == 0xcc && left >= 2) return v[1];
        if (v[0] == 0xcd && left >= 3) return (v[1] << 8) | v[2];
        if (v[0] == 0xce && left >= 5) return (int)(((uint32_t)v[1] << 24) | (v[2] << 16) | (v[3] << 8) | v[4]);
        return -1;
    }
    return -1;
}

// Scratch memory: every cascade kernel of the library keeps its state in registers and LDS, and a specialised build is
// expected to.  What used to send two lane addresses of a depth-3 or 1024-stage build to scratch is dealt with in the
// source (wb_cascade_tile.h: relane, WB_INLINE_LAMBDA).  A build that asks for scratch anyway is not wrong -- it was
// suspected in round 4 and cleared (profiles/r04/jit_selftest.txt) -- but second choice: wb_model_specialize takes it only
// when neither compiler has a scratch-free build that passes the self-test (the toolkit's compiler spills 60 bytes per
// lane in the 128-stage benchmark kernel; under rocprofv3, whose libraries put that compiler into the process first, this
// is the build that runs).  wb_jit_compile_check (the CPU suite) is strict: scratch fails the check.
int build_checked(const Rtc &R, const std::string &src, const char *arch, std::vector<char> &code, std::string &log) {
    const int rc = compile(R, src, arch, code, log);
    if (rc != WB_OK) return rc;
    const int sb = scratch_bytes(code);
    if (sb == 0) return WB_OK;
    code.clear();
    log = sb < 0 ? "the code object's metadata has no readable .private_segment_fixed_size"
                 : "the specialised kernel would spill registers to scratch memory (" + std::to_string(sb) + " bytes per lane)";
    return WB_ERR_UNSUPPORTED;
}

}  // namespace

namespace {
void evict_idle_modules();
}

// Build (or fetch) the specialised kernel for one stage table of `M`.  words: (T + G) records of SD dwords as uploaded.
// compiler: 0 = the hiprtc in the process, 1 = the toolkit's (see Rtc above); WB_ERR_UNSUPPORTED when that one cannot be had,
// and when its build asks for scratch memory while allow_scratch is 0 (the code object stays in the cache either way).
int wb_jit_get(const int32_t *words, size_t n_words, int T, int D, int rpw, int waves, int C, int rows, int pitch, int eb, int lds_stages,
               int compiler, int allow_scratch, void **func_out) {
    *func_out = nullptr;
    const Rtc &R = rtc_of(compiler);
    if (!R.ok) {
        wb_set_error("wb_model_specialize: the toolkit's hiprtc (%s) could not be loaded", R.origin.c_str());
        return WB_ERR_UNSUPPORTED;
    }
    hipDeviceProp_t prop;
    int dev = 0;
    WB_HIP_CHECK(hipGetDevice(&dev));
    WB_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    const std::string src = make_source(words, n_words, T, D, rpw, waves, C, rows, pitch, eb, lds_stages);
    int rtc_major = 0, rtc_minor = 0;
    (void)R.version(&rtc_major, &rtc_minor);
    uint64_t h = fnv1a(src.data(), src.size());
    h = fnv1a(kTileSrc, sizeof(kTileSrc), h);
    h = fnv1a(prop.gcnArchName, strlen(prop.gcnArchName), h);
    h = fnv1a(&rtc_major, sizeof(int), fnv1a(&rtc_minor, sizeof(int), h));
    h = fnv1a(R.origin.data(), R.origin.size(), h);
    h = fnv1a(&dev, sizeof(int), h);                         // (a module is loaded per device)
    if (const char *e = getenv("WB_JIT_DEFS")) h = fnv1a(e, strlen(e), h);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_loaded.find(h);
    if (it != g_loaded.end()) {
        if (it->second.scratch != 0 && !allow_scratch) {
            wb_set_error("wb_model_specialize: this build asks for scratch memory (%d bytes per lane)", it->second.scratch);
            return WB_ERR_UNSUPPORTED;
        }
        ++it->second.users;
        *func_out = it->second.func;
        return WB_OK;
    }
    char name[40];
    snprintf(name, sizeof(name), "%016llx.co", (unsigned long long)h);
    const std::string dir = cache_dir(), path = dir.empty() ? std::string() : dir + "/" + name;
    std::vector<char> code;
    if (path.empty() || !read_file(path, code) || !plausible_code_object(path, code)) {
        std::string log;
        const int rc = compile(R, src, prop.gcnArchName, code, log);
        if (rc != WB_OK) {
            wb_set_error("wb_model_specialize: %.400s", log.c_str());
            return rc;
        }
        if (!path.empty()) write_file_atomic(dir, path, code);
    }
    const int sb = scratch_bytes(code);
    if (sb != 0 && !allow_scratch) {
        wb_set_error("wb_model_specialize: this build asks for scratch memory (%d bytes per lane)", sb);
        return WB_ERR_UNSUPPORTED;
    }
    JitKernel k;
    hipError_t e = hipModuleLoadData(&k.module, code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&k.func, k.module, "wb_casc_jit");
    if (e != hipSuccess) {
        if (!path.empty()) (void)unlink(path.c_str());          // (a stale or damaged cache entry: compile again next time)
        wb_set_error("wb_model_specialize: loading the compiled kernel failed: %s", hipGetErrorString(e));
        return WB_ERR_HIP;
    }
    k.users = 1;
    k.scratch = sb;
    g_loaded[h] = k;
    evict_idle_modules();
    *func_out = k.func;
    return WB_OK;
}

// A model lets go of a specialised kernel (wb_model_destroy, a build that failed the self-test): its module becomes idle.
// It is not unloaded on the spot -- a hipGraph captured with the kernel may outlive the model by a moment (Python drops a
// scan state's model and graph in no particular order) -- but when the next module is loaded and more than
// WB_JIT_MODULES_MAX (64) are: the longest-idle one goes first, behind a device synchronisation.  A sweep over thousands
// of models (training, threshold search) stays bounded; a model that comes back within the window finds its kernel loaded.
void wb_jit_release(void *func) {
    if (!func) return;
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_loaded) {
        if (kv.second.func != (hipFunction_t)func) continue;
        if (kv.second.users > 0 && --kv.second.users == 0) kv.second.released = ++g_release_clock;
        return;
    }
}

namespace {
// (g_mu held) unload idle modules, longest idle first, while more than the cap are loaded
void evict_idle_modules() {
    static const size_t cap = getenv("WB_JIT_MODULES_MAX") ? (size_t)atoi(getenv("WB_JIT_MODULES_MAX")) : 64;
    if (cap == 0) return;                                    // (0: never unload)
    bool synced = false;
    while (g_loaded.size() > cap) {
        auto victim = g_loaded.end();
        for (auto it = g_loaded.begin(); it != g_loaded.end(); ++it)
            if (it->second.users == 0 && (victim == g_loaded.end() || it->second.released < victim->second.released)) victim = it;
        if (victim == g_loaded.end()) return;                // (every module has a user)
        if (!synced) (void)hipDeviceSynchronize();           // (nothing of an idle module's last scans in flight)
        synced = true;
        (void)hipModuleUnload(victim->second.module);
        g_loaded.erase(victim);
    }
}
}  // namespace

// Compile check without a GPU (the CPU test suite): a synthetic cascade of n_stages random depth-`depth` trees through
// the same generator and hiprtc for `arch`.  Returns the code object's size in *code_bytes.
extern "C" int wb_jit_compile_check(int depth, int n_stages, const char *arch, int64_t *code_bytes) {
    return wb_jit_compile_check2(depth, n_stages, 1, arch, code_bytes);
}

extern "C" int wb_jit_compile_check2(int depth, int n_stages, int elem_bytes, const char *arch, int64_t *code_bytes) {
    WB_REQUIRE(depth >= 1 && depth <= WB_CASC_MAX_DEPTH && n_stages >= 1 && n_stages <= 4096 && arch && code_bytes &&
               (elem_bytes == 1 || elem_bytes == 2), "wb_jit_compile_check: bad argument");
    const int SD = WB_STAGE_DWORDS(depth), NI = WB_STAGE_NI(depth), NL = WB_STAGE_NL(depth), G = wb_cascade_group(depth);
    std::vector<int32_t> words((size_t)(n_stages + G) * SD, 0);
    uint32_t x = 12345u;
    auto rnd = [&]() { return x = x * 1664525u + 1013904223u; };
    for (int s = 0; s < n_stages + G; ++s) {
        int32_t *rec = words.data() + (size_t)s * SD;
        float *f = reinterpret_cast<float *>(rec);
        if (s >= n_stages) {
            f[2 * NI + NL] = -INFINITY;
            continue;
        }
        for (int i = 0; i < NI; ++i) {
            rec[i] = (int32_t)(rnd() % 3000u) * elem_bytes;
            rec[NI + i] = (int32_t)(rnd() % (elem_bytes == 2 ? 1000u : 255u));
        }
        for (int i = 0; i < NL; ++i) f[2 * NI + i] = (float)(rnd() % 2000u) * 1e-3f - 1.0f;
        f[2 * NI + NL] = s % 7 == 3 ? -INFINITY : -0.5f * (float)s;
    }
    // (as wb_model_create decides it: a stage table beyond 16 KiB is not mirrored in LDS)
    const int lds_stages = n_stages * SD * 4 <= 16 * 1024 ? n_stages : 0;
    std::vector<char> code;
    std::string log;
    // (the compiler in the process, or -- WB_JIT_CHECK_COMPILER=1 -- the toolkit's)
    const Rtc &R = rtc_of(getenv("WB_JIT_CHECK_COMPILER") ? atoi(getenv("WB_JIT_CHECK_COMPILER")) : 0);
    if (!R.ok) {
        wb_set_error("wb_jit_compile_check: the toolkit's hiprtc (%s) could not be loaded", R.origin.c_str());
        return WB_ERR_UNSUPPORTED;
    }
    const int rc = build_checked(R, make_source(words.data(), words.size(), n_stages, depth, 4, 8, 4, 4 * 8 + 11, WB_CASC_TC + 12, elem_bytes, lds_stages),
                                 arch, code, log);
    if (rc != WB_OK) {
        wb_set_error("wb_jit_compile_check: %.400s", log.c_str());
        return rc;
    }
    *code_bytes = (int64_t)code.size();
    return WB_OK;
}
