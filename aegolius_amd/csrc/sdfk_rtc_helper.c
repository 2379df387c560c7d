/* sdfk_rtc_helper — hiprtc in a child process.
 *
 *   sdfk_rtc_helper <libhiprtc.so> <source file> <output file> [hiprtc option ...]
 *
 * libsdfk.so builds kernels in the background while the interpreter kernel serves the first calls of a new tree
 * shape. Inside the calling process that is not safe: hiprtcCompileProgram holds comgr's process-wide mutex for
 * seconds, and a dlopen of any library with HIP fat binaries on another thread (`import torch`) then deadlocks —
 * loader lock -> comgr mutex on one side, comgr mutex -> loader lock on the other
 * (profiles/r03_hang_import_during_build.txt). Here the compiler has a process of its own: it never touches the GPU,
 * shares no lock with the caller, and several of them can run side by side.
 * The code object is written to <output file>.tmp and renamed; on failure the hiprtc log goes to <output file>.log.
 * Exit code 0 = code object written. Plain C, linked against libdl only: the hiprtc it loads is the caller's. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void* rtc_prog;
typedef int (*fn_create)(rtc_prog*, const char*, const char*, int, const char**, const char**);
typedef int (*fn_compile)(rtc_prog, int, const char**);
typedef int (*fn_size)(rtc_prog, size_t*);
typedef int (*fn_get)(rtc_prog, char*);
typedef int (*fn_destroy)(rtc_prog*);

static void write_log(const char* out, const char* msg, const char* detail) {
    char path[4096];
    snprintf(path, sizeof path, "%s.log", out);
    FILE* f = fopen(path, "wb");
    if (!f) return;
    fprintf(f, "%s\n%s\n", msg, detail ? detail : "");
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <libhiprtc.so> <source> <output> [option ...]\n", argv[0]);
        return 2;
    }
    const char *libpath = argv[1], *srcpath = argv[2], *out = argv[3];
    void* lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        write_log(out, "dlopen of hiprtc failed", dlerror());
        return 3;
    }
    fn_create create = (fn_create)dlsym(lib, "hiprtcCreateProgram");
    fn_compile compile = (fn_compile)dlsym(lib, "hiprtcCompileProgram");
    fn_size log_size = (fn_size)dlsym(lib, "hiprtcGetProgramLogSize"), code_size = (fn_size)dlsym(lib, "hiprtcGetCodeSize");
    fn_get get_log = (fn_get)dlsym(lib, "hiprtcGetProgramLog"), get_code = (fn_get)dlsym(lib, "hiprtcGetCode");
    fn_destroy destroy = (fn_destroy)dlsym(lib, "hiprtcDestroyProgram");
    if (!create || !compile || !log_size || !code_size || !get_log || !get_code || !destroy) {
        write_log(out, "hiprtc entry points missing", libpath);
        return 3;
    }
    FILE* f = fopen(srcpath, "rb");
    if (!f) {
        write_log(out, "cannot read the source file", srcpath);
        return 4;
    }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* src = (char*)malloc((size_t)n + 1);
    if (!src || fread(src, 1, (size_t)n, f) != (size_t)n) {
        write_log(out, "short read of the source file", srcpath);
        return 4;
    }
    src[n] = 0;
    fclose(f);
    rtc_prog prog = 0;
    if (create(&prog, src, "sdfk_spec.hip", 0, 0, 0) != 0) {
        write_log(out, "hiprtcCreateProgram failed", 0);
        return 5;
    }
    const int rc = compile(prog, argc - 4, (const char**)(argv + 4));
    if (rc != 0) {
        size_t ls = 0;
        char* log = 0;
        if (log_size(prog, &ls) == 0 && ls > 1 && (log = (char*)malloc(ls + 1)) != 0) {
            get_log(prog, log);
            log[ls] = 0;
        }
        write_log(out, "hiprtc: compilation failed", log);
        return 6;
    }
    size_t cs = 0;
    if (code_size(prog, &cs) != 0 || cs == 0) {
        write_log(out, "hiprtcGetCodeSize failed", 0);
        return 7;
    }
    char* code = (char*)malloc(cs);
    if (!code || get_code(prog, code) != 0) {
        write_log(out, "hiprtcGetCode failed", 0);
        return 7;
    }
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp", out);
    f = fopen(tmp, "wb");
    if (!f || fwrite(code, 1, cs, f) != cs || fclose(f) != 0 || rename(tmp, out) != 0) {
        write_log(out, "cannot write the code object", tmp);
        remove(tmp);
        return 8;
    }
    destroy(&prog);
    return 0;
}
