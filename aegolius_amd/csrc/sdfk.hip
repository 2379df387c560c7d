// sdfk.hip — kernels and C-ABI of libsdfk.so (gfx950 only).
//
//   * sdfk_interp_kernel<VEC,SRC> : generic register-machine interpreter. The program and its
//     parameters are wave-uniform, so the dispatch runs on the scalar unit (s_load of the
//     instruction word, scalar branch) and parameters arrive in SGPRs through the scalar cache.
//   * specialised kernels        : the same per-point functions (sdfk_device.h) called in
//     straight-line order, generated per tree TOPOLOGY (parameters stay runtime data) by
//     sdfk_codegen.cpp, compiled with hiprtc on first use and cached per (device, topology).
//   * access pattern             : one thread owns 4 consecutive points — three 16-byte loads
//     (x, y, z rows of the (3,N) array), one 16-byte store; a wave covers 1 KiB per row per
//     instruction, fully coalesced. Algorithmic traffic 16 B/point.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "../../include/sdfk.h"
#include "sdfk_device.h"
#include "sdfk_access.h"
#include "sdfk_codegen.h"

// ------------------------------------------------------------------------------------------------
// opcode tables from the single source of truth
// ------------------------------------------------------------------------------------------------
enum {
#define SDFK_OP(NAME, KIND, NP, FUNC) SDFK_OP_##NAME,
#include "sdfk_ops.def"
#undef SDFK_OP
    SDFK_OP_COUNT
};

static const sdfk_opinfo g_ops[] = {
#define SDFK_OP(NAME, KIND, NP, FUNC) {#NAME, SDFK_KIND_##KIND, NP, #FUNC},
#include "sdfk_ops.def"
#undef SDFK_OP
};

extern "C" const sdfk_opinfo* sdfk_op_table(int* count) {
    if (count) *count = SDFK_OP_COUNT;
    return g_ops;
}

// interpreter register-file limits (the specialised path has none beyond the 8-bit operand fields)
#define SDFK_NC 8
#define SDFK_NV 8
// six programs in seven need at most 2 coordinate and 3 value registers (every BASELINE config does): that register
// file fits the VGPRs, the full one lives in scratch
#define SDFK_NC_SMALL 2
#define SDFK_NV_SMALL 3

// ------------------------------------------------------------------------------------------------
// interpreter kernel
// ------------------------------------------------------------------------------------------------
template <int VEC, int NC, int NV, typename SRC>
__global__ __launch_bounds__(SDFK_BLOCK) void sdfk_interp_kernel(const uint2* __restrict__ code, int n_instr,
                                                                const float* __restrict__ prm,
                                                                const float* __restrict__ tab, SRC src, long long off,
                                                                long long n, float* __restrict__ out, int result_reg,
                                                                const float* __restrict__ aux, long long aux_stride) {
    const long long block_base = (long long)sdfk_bx() * (SDFK_BLOCK * VEC);
    const unsigned lane_off = sdfk_tx() * VEC;
    if (block_base + lane_off >= n) return;
    // Register files indexed by the (wave-uniform) register number. The small coordinate file is three plain float
    // arrays, which the compiler keeps in VGPRs next to V; the full-size one is an array of structs and lives in
    // scratch — as plain arrays it takes 173 VGPRs, two waves per SIMD, and is slower (52.7 against 44.7 ms on the
    // north-star tree). (Wrapping either in a struct sends V to scratch as well: 62 ms.)
    constexpr bool SPLIT = NC <= SDFK_NC_SMALL;
    float CX[SPLIT ? NC : 1][VEC], CY[SPLIT ? NC : 1][VEC], CZ[SPLIT ? NC : 1][VEC];
    V3 CS[SPLIT ? 1 : NC][VEC];
    float V[NV][VEC];
#define SDFK_CGET(r, v) (SPLIT ? V3{CX[SPLIT ? (r) : 0][v], CY[SPLIT ? (r) : 0][v], CZ[SPLIT ? (r) : 0][v]} : CS[SPLIT ? 0 : (r)][v])
#define SDFK_CSET(r, v, q)                                                                      \
    do {                                                                                        \
        const V3 q_ = (q);                                                                      \
        if constexpr (SPLIT) { CX[SPLIT ? (r) : 0][v] = q_.x; CY[SPLIT ? (r) : 0][v] = q_.y; CZ[SPLIT ? (r) : 0][v] = q_.z; } \
        else CS[SPLIT ? 0 : (r)][v] = q_;                                                       \
    } while (0)
    {
        V3 p0[VEC];
        sdfk_load<VEC>(src, off + block_base, lane_off, p0);
        _Pragma("unroll") for (int v = 0; v < VEC; ++v) SDFK_CSET(0, v, p0[v]);
    }
    uint2 fetched = code[0];         // wave-uniform -> scalar load; the next word is requested before this one executes
    for (int pc = 0; pc < n_instr; ++pc) {
        const uint2 ins = fetched;
        fetched = code[min(pc + 1, n_instr - 1)];
        const unsigned op = ins.x & 255u, a = (ins.x >> 8) & 255u, b = (ins.x >> 16) & 255u, c = ins.x >> 24;
        const float* __restrict__ P = prm + ins.y;
        if (op == SDFK_OP_V_FIELD) {   // auxiliary field c at this lane's points (validated: aux != nullptr)
            _Pragma("unroll") for (int v = 0; v < VEC; ++v)
                V[a][v] = sdfk_aux<float>(aux + off + block_base + lane_off + v, aux_stride, (int)c);
            continue;
        }
        switch (op) {
#define SDFK_EXEC_C_C(F) \
    _Pragma("unroll") for (int v = 0; v < VEC; ++v) SDFK_CSET(a, v, F(SDFK_CGET(b, v), P, tab, (int)c))
#define SDFK_EXEC_V_C(F) \
    _Pragma("unroll") for (int v = 0; v < VEC; ++v) V[a][v] = F(SDFK_CGET(b, v), P, tab)
#define SDFK_EXEC_V_V(F) \
    _Pragma("unroll") for (int v = 0; v < VEC; ++v) V[a][v] = F(V[b][v], P)
#define SDFK_EXEC_V_VV(F) \
    _Pragma("unroll") for (int v = 0; v < VEC; ++v) V[a][v] = F(V[b][v], V[c][v], P)
#define SDFK_OP(NAME, KIND, NP, FUNC) \
    case SDFK_OP_##NAME:              \
        SDFK_EXEC_##KIND(FUNC);       \
        break;
#include "sdfk_ops.def"
#undef SDFK_OP
            default:
                break;
        }
    }
    sdfk_store<VEC>(out, off + block_base + lane_off, V[result_reg]);
}

// (3,n) -> (n) streaming probe with the evaluation kernels' access pattern
__global__ __launch_bounds__(SDFK_BLOCK) void sdfk_probe_kernel(SrcArray src, long long n, float* __restrict__ out) {
    const long long block_base = (long long)sdfk_bx() * (SDFK_BLOCK * 4);
    const unsigned lane_off = sdfk_tx() * 4;
    if (block_base + lane_off >= n) return;
    V3 p[4];
    sdfk_load<4>(src, block_base, lane_off, p);
    float v[4] = {p[0].x + p[0].y + p[0].z, p[1].x + p[1].y + p[1].z, p[2].x + p[2].y + p[2].z,
                  p[3].x + p[3].y + p[3].z};
    sdfk_store<4>(out, block_base + lane_off, v);
}

__global__ __launch_bounds__(SDFK_BLOCK) void sdfk_gridfill_kernel(SrcGrid src, long long n, float* __restrict__ co,
                                                                  long long stride) {
    const long long block_base = (long long)sdfk_bx() * (SDFK_BLOCK * 4);
    const unsigned lane_off = sdfk_tx() * 4;
    const long long i = block_base + lane_off;
    if (i >= n) return;
    V3 p[4];
    sdfk_load<4>(src, block_base, lane_off, p);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if (i + v < n) {
            co[i + v] = p[v].x;
            co[stride + i + v] = p[v].y;
            co[2 * stride + i + v] = p[v].z;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int g_default_mode = SDFK_MODE_AUTO;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (void)hipGetLastError(); /* the runtime keeps it: the next launch check would see it */ \
            return fail(-100 - (int)e_, std::string(#expr) + ": " + hipGetErrorString(e_));       \
        }                                                                                         \
    } while (0)

// candidate lists of one evaluation in flight (chain mode, csrc/sdfk_codegen.cpp "sdfk_spec_cells"): cell spheres, list
// spans, the pool the lists are allocated from and its head — one set per STREAM, because calls on different streams
// (the two slots of the host pipeline) run side by side on one program
struct CellScratch {
    char* buf = nullptr;
    size_t bytes = 0;
};
struct DevState {
    uint2* d_code = nullptr;
    float* d_params = nullptr;
    float* d_tables = nullptr;
    unsigned long long params_version = 0;
    std::map<hipStream_t, CellScratch> cells;
};

// One hiprtc translation unit = (program topology, kernel flavour, build options). The code object is built once
// per PROCESS — by whichever thread asks first, outside every global lock — and shared by all devices; a device
// only loads it (hipModuleLoadData). state: 0 idle, 1 building, 2 ready, 3 failed.
struct CodeObject {
    std::mutex mu;
    std::condition_variable cv;
    int state = 0;
    std::vector<char> co;
    std::string error;
    std::string disk_path;                // non-empty: `co` was read from this file of the on-disk cache
    double build_seconds = 0.0;
    std::chrono::steady_clock::time_point failed_at;
};
struct SpecModule {                   // a code object loaded on one device
    std::mutex mu;
    bool loaded = false, failed = false;
    hipModule_t mod = nullptr;
    hipFunction_t fn[2] = {nullptr, nullptr};
    std::string error;
};

struct sdfk_program {
    std::vector<uint32_t> code;  // 2 words / instruction
    std::vector<float> params, tables;
    int result_reg = 0;
    bool interp_ok = true;  // fits the interpreter's register file
    bool interp_small = false;  // ... and its small instantiation (SDFK_NC_SMALL coordinate / SDFK_NV_SMALL value registers)
    int n_aux = 0;          // auxiliary per-point fields read by V_FIELD instructions
    unsigned long long params_version = 1;
    std::string key;
    std::string source;
    std::vector<sdfk_cullsite> sites;  // brick-culling sites the mask kernels use: the SDFK_MASK_SITES widest (sdfk_program_set_cull)
    std::vector<sdfk_cullsite> sites_all;  // every site that was passed in
    bool chain_mode = false;           // long n-ary min / max chain: table-driven kernels (sdfk_codegen.cpp)
    int chain_members = 0;             // its members (the program may hold more sites: combiners above the chain)
    std::mutex mu;
    std::map<int, DevState> dev;
};

static std::mutex g_code_mu;                                    // guards the two maps only, never a build
static std::map<std::string, std::shared_ptr<CodeObject>> g_code;
static std::map<std::pair<int, std::string>, std::shared_ptr<SpecModule>> g_mods;

extern "C" int sdfk_abi_version(void) { return SDFK_ABI_VERSION; }

extern "C" int sdfk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" const char* sdfk_last_error(void) { return g_err.c_str(); }
#define SDFK_MASK_SITES 512   // sites a row-block kernel carries skip bits for (16 words of 64 bits per brick at most)
extern "C" void sdfk_set_default_mode(int mode) { g_default_mode = mode; }

// ---- validation -------------------------------------------------------------------------------
static int table_rows_ok(const sdfk_program* p, int op, const float* P, std::string* why) {
    auto bad = [&](const char* m) {
        *why = m;
        return 0;
    };
    long long cnt = (long long)P[0], off = (long long)P[1];
    if (op == SDFK_OP_P_POLYSIGN && P[0] == -1.0f) cnt = 1;      // one piece, raw side value (interior_convex)
    else if (!(P[0] >= 0.0f) || (float)cnt != P[0]) return bad("table count/offset not integral");
    if (!(P[1] >= 0.0f) || (float)off != P[1]) return bad("table count/offset not integral");
    long long nt = (long long)p->tables.size();
    long long row = 0;
    switch (op) {
        case SDFK_OP_P_SEGLINE3: row = 7; break;
        case SDFK_OP_P_NEAREST3: row = 3; break;
        case SDFK_OP_P_SEGLINE2: row = 5; break;
        case SDFK_OP_P_NEAREST2: row = 2; break;
        case SDFK_OP_P_SHAPESIGN: row = 6; break;
        case SDFK_OP_CURVEINST: row = (P[2] != 0.0f) ? 12 : 3; break;
        case SDFK_OP_CURVEINSTT:
        case SDFK_OP_P_NEARTREE: {
            // every index the kernel will follow stays inside the table; no box is empty (root -> middle -> leaf -> points)
            if (cnt < 1) return bad("point tree without root boxes");
            if (op == SDFK_OP_P_NEARTREE && P[2] != 2.0f && P[2] != 3.0f) return bad("point tree dimension must be 2 or 3");
            if (off + 8 * cnt > nt) return bad("point tree: root boxes out of range");
            const float* t = p->tables.data() + off;
            const long long room = nt - off;
            auto kids = [&](const float* box, long long width, long long* first, long long* n) {
                const float ff = box[6], fn = box[7];
                *first = (long long)ff;
                *n = (long long)fn;
                return ff >= 0.0f && fn >= 1.0f && (float)*first == ff && (float)*n == fn && *first + width * *n <= room;
            };
            for (long long r = 0; r < cnt; ++r) {
                long long mfirst, nm;
                if (!kids(t + 8 * r, 8, &mfirst, &nm)) return bad("point tree: middle boxes out of range");
                for (long long i = 0; i < nm; ++i) {
                    long long first, nl;
                    if (!kids(t + mfirst + 8 * i, 8, &first, &nl)) return bad("point tree: leaf boxes out of range");
                    for (long long l = 0; l < nl; ++l) {
                        long long pfirst, pcount;
                        if (!kids(t + first + 8 * l, 3, &pfirst, &pcount)) return bad("point tree: points out of range");
                        if (op == SDFK_OP_CURVEINSTT) {
                            // the original index of every point the kernel can pick is readable and names a row inside the table
                            const long long base = (long long)P[4], rows = (long long)P[3], ids = (long long)P[5];
                            const long long stride = (P[2] != 0.0f) ? 12 : 3;
                            if (!(P[3] >= 0.0f) || !(P[4] >= 0.0f) || !(P[5] >= 0.0f) || (float)base != P[4] || (float)rows != P[3] ||
                                (float)ids != P[5] || pfirst < base || (pfirst - base) % 3 != 0 ||
                                ids + (pfirst - base) / 3 + pcount > nt)
                                return bad("instancing tree: original indices out of range");
                            for (long long q = 0; q < pcount; ++q) {
                                const float of = p->tables[(size_t)(ids + (pfirst - base) / 3 + q)];
                                const long long o = (long long)of;
                                if (!(of >= 0.0f) || (float)o != of || rows + (o + 1) * stride > nt)
                                    return bad("instancing tree: instance row out of range");
                            }
                        }
                    }
                }
            }
            return 1;
        }
        case SDFK_OP_P_POLYSIGN: {
            long long pos = off;
            for (long long j = 0; j < cnt; ++j) {
                if (pos >= nt) return bad("polygon piece header out of range");
                float kf = p->tables[pos];
                long long k = (long long)kf;
                if (!(kf >= 0.0f) || (float)k != kf) return bad("polygon piece size not integral");
                pos += 1 + 4 * k;
                if (pos > nt) return bad("polygon piece out of range");
            }
            return 1;
        }
        default: return 1;
    }
    if (off + cnt * row > nt) return bad("table rows out of range");
    if (op == SDFK_OP_CURVEINST && cnt < 1) return bad("curve instancing needs at least one instance");
    return 1;
}

static bool is_table_op(int op) {
    return op == SDFK_OP_P_SEGLINE3 || op == SDFK_OP_P_NEAREST3 || op == SDFK_OP_P_SEGLINE2 ||
           op == SDFK_OP_P_NEAREST2 || op == SDFK_OP_CURVEINST || op == SDFK_OP_P_POLYSIGN || op == SDFK_OP_P_NEARTREE ||
           op == SDFK_OP_CURVEINSTT ||
           op == SDFK_OP_P_SHAPESIGN;
}

static int validate(sdfk_program* p, std::string* why) {
    const size_t n_instr = p->code.size() / 2;
    std::vector<char> cdef(256, 0), vdef(256, 0);
    cdef[0] = 1;  // C0 = input point
    unsigned max_c = 0, max_v = 0;
    char buf[160];
    for (size_t i = 0; i < n_instr; ++i) {
        const uint32_t w = p->code[2 * i], poff = p->code[2 * i + 1];
        const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
        if (op >= SDFK_OP_COUNT) {
            snprintf(buf, sizeof buf, "instruction %zu: unknown opcode %u", i, op);
            *why = buf;
            return 0;
        }
        const sdfk_opinfo& info = g_ops[op];
        if ((size_t)poff + (size_t)info.nparams > p->params.size()) {
            snprintf(buf, sizeof buf, "instruction %zu (%s): parameters out of range", i, info.name);
            *why = buf;
            return 0;
        }
        bool ok = true;
        switch (info.kind) {
            case SDFK_KIND_C_C: ok = cdef[b]; cdef[a] = 1; max_c = std::max(max_c, std::max(a, b)); break;
            case SDFK_KIND_V_C: ok = cdef[b]; vdef[a] = 1; max_c = std::max(max_c, b); max_v = std::max(max_v, a); break;
            case SDFK_KIND_V_V: ok = vdef[b]; vdef[a] = 1; max_v = std::max(max_v, std::max(a, b)); break;
            case SDFK_KIND_V_VV:
                ok = vdef[b] && vdef[c];
                vdef[a] = 1;
                max_v = std::max(max_v, std::max(a, std::max(b, c)));
                break;
        }
        if (!ok) {
            snprintf(buf, sizeof buf, "instruction %zu (%s): reads a register that was never written", i, info.name);
            *why = buf;
            return 0;
        }
        if (op == SDFK_OP_V_FIELD) {
            if (c >= 32) {
                snprintf(buf, sizeof buf, "instruction %zu: auxiliary field index %u out of range (32)", i, c);
                *why = buf;
                return 0;
            }
            p->n_aux = std::max(p->n_aux, (int)c + 1);
        }
        if (op == SDFK_OP_SYMMETRY && c > 2) {
            snprintf(buf, sizeof buf, "instruction %zu: symmetry axis %u out of range", i, c);
            *why = buf;
            return 0;
        }
        if (is_table_op((int)op)) {
            std::string w2;
            if (!table_rows_ok(p, (int)op, p->params.data() + poff, &w2)) {
                snprintf(buf, sizeof buf, "instruction %zu (%s): %s", i, info.name, w2.c_str());
                *why = buf;
                return 0;
            }
        }
    }
    if (p->result_reg < 0 || p->result_reg > 255 || !vdef[p->result_reg]) {
        *why = "result register is never written";
        return 0;
    }
    p->interp_ok = (max_c < SDFK_NC) && (max_v < SDFK_NV) && (p->result_reg < SDFK_NV);
    p->interp_small = (max_c < SDFK_NC_SMALL) && (max_v < SDFK_NV_SMALL) && (p->result_reg < SDFK_NV_SMALL);
    return 1;
}

extern "C" sdfk_program* sdfk_program_create(const uint32_t* code, size_t n_instr, const float* params,
                                             size_t n_params, const float* tables, size_t n_tables, int result_reg) {
    if (!code || n_instr == 0 || n_instr > (1u << 20)) {
        fail(-1, "sdfk_program_create: empty or oversized program");
        return nullptr;
    }
    std::unique_ptr<sdfk_program> p(new sdfk_program);
    p->code.assign(code, code + 2 * n_instr);
    if (params && n_params) p->params.assign(params, params + n_params);
    if (tables && n_tables) p->tables.assign(tables, tables + n_tables);
    p->result_reg = result_reg;
    std::string why;
    if (!validate(p.get(), &why)) {
        fail(-2, "sdfk_program_create: " + why);
        return nullptr;
    }
    p->key.assign(reinterpret_cast<const char*>(p->code.data()), p->code.size() * sizeof(uint32_t));
    p->key.push_back((char)result_reg);
    return p.release();
}

static void free_dev_state(sdfk_program* p) {
    int cur = 0;
    bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto& kv : p->dev) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        if (kv.second.d_code) (void)hipFree(kv.second.d_code);
        if (kv.second.d_params) (void)hipFree(kv.second.d_params);
        if (kv.second.d_tables) (void)hipFree(kv.second.d_tables);
        for (auto& cs : kv.second.cells)
            if (cs.second.buf) (void)hipFree(cs.second.buf);
    }
    if (have) (void)hipSetDevice(cur);
    p->dev.clear();
}

extern "C" void sdfk_program_destroy(sdfk_program* p) {
    if (!p) return;
    free_dev_state(p);
    delete p;
}

extern "C" int sdfk_program_set_params(sdfk_program* p, const float* params, size_t n_params) {
    if (!p) return fail(-1, "null program");
    std::lock_guard<std::mutex> lk(p->mu);
    if (n_params != p->params.size()) return fail(-2, "sdfk_program_set_params: parameter count differs from the program's");
    std::vector<float> old = p->params;
    p->params.assign(params, params + n_params);
    std::string why;
    if (!validate(p, &why)) {  // table counts / offsets live in the parameters
        p->params = old;
        return fail(-2, "sdfk_program_set_params: " + why);
    }
    p->params_version++;
    return 0;
}

// ---- brick culling sites ---------------------------------------------------------------------
static void instr_fields(const sdfk_program* p, size_t i, unsigned* op, unsigned* a, unsigned* b, unsigned* c) {
    const uint32_t w = p->code[2 * i];
    *op = w & 255u; *a = (w >> 8) & 255u; *b = (w >> 16) & 255u; *c = w >> 24;
}

// May the instructions [lo, hi] be skipped when the combiner `comb` does not need the value they
// produce? Only if nothing executed later reads a register they would have written.
static bool range_skippable(const sdfk_program* p, uint32_t lo, uint32_t hi, uint32_t comb, unsigned result_vreg) {
    const size_t n = p->code.size() / 2;
    std::vector<char> wc(256, 0), wv(256, 0);
    unsigned op, a, b, c;
    for (size_t i = lo; i <= hi; ++i) {
        instr_fields(p, i, &op, &a, &b, &c);
        (g_ops[op].kind == SDFK_KIND_C_C ? wc : wv)[a] = 1;
    }
    instr_fields(p, hi, &op, &a, &b, &c);
    if (g_ops[op].kind == SDFK_KIND_C_C || a != result_vreg) return false;  // range must end by producing the operand
    for (size_t j = hi + 1; j < n; ++j) {
        instr_fields(p, j, &op, &a, &b, &c);
        const int kind = g_ops[op].kind;
        if (kind == SDFK_KIND_C_C || kind == SDFK_KIND_V_C) {
            if (wc[b]) return false;
        } else if (kind == SDFK_KIND_V_V) {
            if (wv[b]) return false;
        } else {
            const bool is_comb = (j == comb);
            if (wv[b] && !(is_comb && b == result_vreg)) return false;
            if (wv[c] && !(is_comb && c == result_vreg)) return false;
        }
        (kind == SDFK_KIND_C_C ? wc : wv)[a] = 0;   // redefined: later reads see the new value
    }
    return !wv[(unsigned)p->result_reg];
}

static bool is_cullable_op(unsigned op) {
    return op == SDFK_OP_VMIN || op == SDFK_OP_VMAX || op == SDFK_OP_VSUBTRACT || op == SDFK_OP_SMIN2 ||
           op == SDFK_OP_SMIN3 || op == SDFK_OP_SMAX3 || op == SDFK_OP_SSUB3;
}

extern "C" int sdfk_program_set_cull(sdfk_program* p, const uint32_t* rows, size_t n_sites, const float* k) {
    if (!p) return fail(-1, "null program");
    if (n_sites > 32767) return fail(-2, "sdfk_program_set_cull: at most 32767 sites");
    if (n_sites && (!rows || !k)) return fail(-1, "sdfk_program_set_cull: null arrays");
    std::lock_guard<std::mutex> lk(p->mu);
    if (!p->source.empty() || !p->dev.empty())
        return fail(-2, "sdfk_program_set_cull: must be called before the program is first used");
    if (p->n_aux > 0) n_sites = 0;   // auxiliary fields have no Lipschitz bound and the culling kernels do not carry them
    const size_t n = p->code.size() / 2;
    std::vector<sdfk_cullsite> sites;
    for (size_t i = 0; i < n_sites; ++i) {
        sdfk_cullsite t{rows[5 * i], rows[5 * i + 1], rows[5 * i + 2], rows[5 * i + 3], rows[5 * i + 4], k[i], 0, 0};
        if (!(t.comb < n && t.a0 <= t.a1 && t.a1 + 1 == t.b0 && t.b0 <= t.b1 && t.b1 + 1 == t.comb))
            return fail(-2, "sdfk_program_set_cull: malformed site ranges");
        if (!(t.k >= 0.0f) || !std::isfinite(t.k)) return fail(-2, "sdfk_program_set_cull: bad Lipschitz sum");
        unsigned op, a, b, c;
        instr_fields(p, t.comb, &op, &a, &b, &c);
        if (!is_cullable_op(op)) return fail(-2, "sdfk_program_set_cull: site is not at a min/max-type combiner");
        if (i && rows[5 * i] <= rows[5 * (i - 1)]) return fail(-2, "sdfk_program_set_cull: sites must be sorted");
        for (const sdfk_cullsite& u : sites) {   // spans [a0, comb] nest or are disjoint
            const bool disjoint = u.comb < t.a0;
            const bool nested = u.a0 >= t.a0 && (u.comb <= t.a1 || (u.a0 >= t.b0 && u.comb <= t.b1));
            if (!disjoint && !nested) return fail(-2, "sdfk_program_set_cull: sites overlap without nesting");
        }
        t.skip_a_ok = range_skippable(p, t.a0, t.a1, t.comb, b) ? 1 : 0;
        t.skip_b_ok = range_skippable(p, t.b0, t.b1, t.comb, c) ? 1 : 0;
        sites.push_back(t);
    }
    // the row-block kernels carry two mask bits per site, 32 sites per 64-bit word and brick (SDFK_NMASK words; rounds 1-3:
    // two words, 64 sites — a left-deep union of 200 primitives then evaluated 135 of them at every point): up to
    // SDFK_MASK_SITES = 512 sites, the widest ones, in program order (the line-brick kernel picks its 31 among them)
    p->sites_all = sites;
    if (sites.size() > SDFK_MASK_SITES) {
        std::vector<size_t> order(sites.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            return (sites[x].a1 - sites[x].a0) + (sites[x].b1 - sites[x].b0) > (sites[y].a1 - sites[y].a0) + (sites[y].b1 - sites[y].b0);
        });
        order.resize(SDFK_MASK_SITES);
        std::sort(order.begin(), order.end());
        std::vector<sdfk_cullsite> widest;
        for (size_t i : order) widest.push_back(sites[i]);
        sites.swap(widest);
    }
    p->sites = sites;
    p->chain_members = sdfk_chain_mode(g_ops, SDFK_OP_COUNT, p->code.data(), p->code.size() / 2, p->result_reg, p->sites_all);
    p->chain_mode = p->chain_members > 0;
    p->key.append("|cull");
    for (const sdfk_cullsite& t : p->sites_all) {
        p->key.append(reinterpret_cast<const char*>(&t), sizeof t);
    }
    return 0;
}

extern "C" const char* sdfk_program_source(sdfk_program* p) {
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(p->mu);
    if (p->source.empty())
        p->source = sdfk_generate_source(g_ops, SDFK_OP_COUNT, p->code.data(), p->code.size() / 2, p->result_reg,
                                         p->sites, SDFK_FL_ALL, &p->sites_all);
    return p->source.c_str();
}

// tile geometry of the brick-culling kernel: SDFK_TWAVES waves x SDFK_WBRICKS bricks of 128 points per
// workgroup (overridable through the environment for experiments)
// (experiments: "-DSDFK_TWAVES=n" / "-DSDFK_RWBRICKS=n" inside the extra build switches override the launch geometry)
static std::atomic<int> g_twaves_override{0}, g_rwbricks_override{0};
static int tile_waves() {
    static int v = [] { const char* e = getenv("SDFK_TWAVES"); int t = e ? atoi(e) : 4; return (t >= 1 && t <= 16) ? t : 4; }();
    const int o = g_twaves_override.load();
    return o ? o : v;
}
static int tile_wbricks() {
    static int v = [] {
        const char* e = getenv("SDFK_WBRICKS");
        int t = e ? atoi(e) : 4;
        if (t < 1 || t > 32) t = 4;
        return t;
    }();
    return v;
}
static int tile_points() { return tile_waves() * tile_wbricks() * 128; }
// bricks per wave of the row-block kernel: 2 — except for big trees (> 150 instructions, e.g. the 50-primitive 2-D
// union), whose whole-tree probe is better shared by 16 bricks per workgroup than by 8 (measured -11 %)
// waves per workgroup of the row-block kernel: 4. (Round 2 measured 2 best, when ONE lane per brick probed the whole tree and
// a bigger workgroup only made more waves wait for it. Since the probe runs on all lanes — round 3 — the workgroup's serial
// steps, centres and fold on the first wave, are shared by more bricks: round 4, one box, 2 -> 4 waves: north-star tree
// 2.944 -> 2.913 ms, 20-primitive tree 3.356 -> 3.186 at 1025^3 and 25.4 -> 24.9 at 2049^3, 50-member flat union 0.847 ->
// 0.833, 513^3 1-2 %; 3, 6 and 8 waves are slower everywhere: profiles/r04_rwaves_sweep.txt.)
static std::atomic<int> g_rwaves_override{0};
static int rows_waves(const sdfk_program*) {
    if (const int o = g_rwaves_override.load()) return o;
    return 4;
}
static int rows_wbricks(const sdfk_program* p);
// Programs that are not chains and hold more than SDFK_BIG_PROGRAM instructions (300) are built with two LLVM passes off
// (big_build_options): bit 16 of the geometry word, which selects the compiler options of a build and is part of its key
static long long big_program_limit() {
    static const long long v = [] {
        const char* e = getenv("SDFK_BIG_PROGRAM");
        const long long t = e ? atoll(e) : 300;
        return t > 0 ? t : 300;
    }();
    return v;
}
static int rows_geo(const sdfk_program* p) {
    const bool big = p && !p->chain_mode && (long long)(p->code.size() / 2) > big_program_limit();
    return rows_wbricks(p) | (rows_waves(p) << 4) | (big ? 1 << 16 : 0);
}
static int rows_wbricks(const sdfk_program* p) {
    static int forced = [] { const char* e = getenv("SDFK_RWBRICKS"); int t = e ? atoi(e) : 0; return (t >= 1 && t <= 16) ? t : 0; }();
    if (const int o = g_rwbricks_override.load()) return o;
    if (forced) return forced;
    // chain mode (measured, 513^3 sphere unions and the 50-child flat union): every brick of a wave costs a fold and an
    // evaluation pass one after the other, and the leaf values take 6 bytes of LDS per child and brick — few bricks per
    // wave win: 1000 spheres 21.9 / 11.4 / 5.5 ms with 4 / 2 / 1, the flat union 1.08 / 0.99 / 1.03 ms
    if (p && p->chain_mode) return p->chain_members <= 64 ? 2 : 1;
    // (rounds 2-3 gave programs beyond 150 instructions 4 bricks per wave; with skip bits for every site — SDFK_MASK_SITES —
    //  2 win at every size: 70 / 100 / 150 / 200 primitives at 513^3 1.40 / 1.87 / 2.79 / 3.51 ms against 1.55 / 2.34 / 3.18 /
    //  4.00, profiles/r04_bigtree_wbricks.txt)
    return 2;
}
struct RowGeom {           // mirrors sdfk_rowgeom of the generated source
    unsigned L, nchunk, nbricks;
    long long R;
    long long row0;
    int yrows;
    // row blocks never straddle a PLANE of the grid (rows of one x): the slab's rows are the rest of a first plane
    // (seg0 rows, nb0 blocks), then planes of prow rows (bpp blocks each; the last block of a plane may be partial)
    unsigned prow, seg0, nb0, bpp;
    unsigned inv_nchunk, inv_bpp;                              // floor(2^32 / nchunk), floor(2^32 / bpp) (sdfk_udiv)
};
// can the row-block kernel take n points in rows of row_len? (brick ids are 32-bit)
// plane_rows: rows per grid plane (0 / >= R: one plane — blocks of 16 consecutive rows throughout);
// plane_phase: index within its plane of the first row. Both are layout hints like row_len: they only decide which
// 16 rows form a block (a block of rows from two planes has a bounding sphere as wide as the grid and culls nothing).
static bool rows_geometry(long long n, long long row_len, RowGeom* g, long long plane_rows = 0, long long plane_phase = 0,
                          bool planes_on = false) {
    if (row_len < 32 || row_len > 0x7fffffffLL || n <= 0 || n % row_len != 0) return false;
    const long long R = n / row_len, brows = 16;
    // windows of 32 points aligned in the flat array: one more than ceil(L / 32) can overlap a row
    const long long nchunk = (row_len % 32 == 0) ? row_len / 32 : (row_len + 62) / 32;
    long long prow = plane_rows, seg0 = 0;
    // Measured on 513^3 / 1025^3 (tools/rows_ab.py `noplanes:`): the partial block that ends every plane of 2^k + 1 rows
    // costs as much as the one straddling block it replaces saves (513^3: 0.413 vs 0.401 ms, 1025^3 equal) — so the hint
    // is honoured only on request (SDFK_PLANE_BLOCKS=1); the default is blocks of 16 consecutive rows throughout.
    static const bool plane_blocks = [] { const char* e = getenv("SDFK_PLANE_BLOCKS"); return e && e[0] == '1'; }();
    if (!(plane_blocks || planes_on) || prow <= 0 || prow >= R || prow > 0x7fffffffLL) {
        prow = R > 0x7fffffffLL ? 0 : R;                       // one plane
        if (prow == 0) return false;
    } else if (plane_phase > 0) {
        seg0 = std::min(R, (prow - plane_phase % prow) % prow);
    }
    const long long nb0 = (seg0 + brows - 1) / brows, bpp = (prow + brows - 1) / brows;
    const long long planes = (R - seg0 + prow - 1) / prow;
    const long long nb = nchunk * (nb0 + planes * bpp);
    if (nb > 0x7fffffffLL - 1024) return false;
    g->L = (unsigned)row_len;
    g->nchunk = (unsigned)nchunk;
    g->nbricks = (unsigned)nb;
    g->R = R;
    g->row0 = 0;
    g->yrows = 0;
    g->prow = (unsigned)prow;
    g->seg0 = (unsigned)seg0;
    g->nb0 = (unsigned)nb0;
    g->bpp = (unsigned)bpp;
    g->inv_nchunk = (unsigned)std::min<unsigned long long>(0xffffffffull, (1ull << 32) / (unsigned long long)nchunk);
    g->inv_bpp = (unsigned)std::min<unsigned long long>(0xffffffffull, (1ull << 32) / (unsigned long long)bpp);
    return true;
}
static int tile_threads() { return 64 * tile_waves(); }
// On-disk cache of hiprtc code objects, ON by default: a new process loads the kernels of tree / chain shapes it has
// seen before instead of compiling them. Directory: $SDFK_CACHE_DIR, else $XDG_CACHE_HOME/sdfk, else $HOME/.cache/sdfk;
// SDFK_CACHE_DIR= (empty), "off" or "0" disables it. The file name is a 64-bit FNV-1a hash of the source, the options
// and the hiprtc version, plus the source length. A cache that cannot be created, read or written is never an error.
static std::string rtc_cache_dir() {
    static const std::string dir = [] {
        std::string d;
        if (const char* e = getenv("SDFK_CACHE_DIR")) {
            d = e;
            if (d.empty() || d == "off" || d == "0") return std::string();
        } else if (const char* x = getenv("XDG_CACHE_HOME"); x && *x) {
            d = std::string(x) + "/sdfk";
        } else if (const char* h = getenv("HOME"); h && *h) {
            (void)mkdir((std::string(h) + "/.cache").c_str(), 0700);
            d = std::string(h) + "/.cache/sdfk";
        } else {
            return std::string();
        }
        (void)mkdir(d.c_str(), 0700);
        return d;
    }();
    return dir;
}
static std::string rtc_cache_path(const std::string& src, const std::string& opts) {
    const std::string dir = rtc_cache_dir();
    if (dir.empty()) return std::string();
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const std::string& t) {
        for (unsigned char c : t) {
            h ^= c;
            h *= 1099511628211ull;
        }
    };
    mix(src);
    mix(opts);
    mix(std::to_string(major) + "." + std::to_string(minor) + "/abi" + std::to_string(SDFK_ABI_VERSION));
    char name[96];
    snprintf(name, sizeof name, "/sdfk-%016llx-%zu.co", h, src.size());
    return dir + name;
}
// File = code object + 24-byte trailer {magic, payload length, FNV-1a of the payload}: a truncated or foreign file is
// never handed to hipModuleLoadData (it is deleted instead).
static const unsigned long long kCacheMagic = 0x53444643'4f424a31ull;           // "SDFCOBJ1"
static unsigned long long fnv1a(const char* p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= (unsigned char)p[i];
        h *= 1099511628211ull;
    }
    return h;
}
static bool rtc_cache_read(const std::string& path, std::vector<char>* out) {
    if (path.empty()) return false;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long size = ftell(f);
        if (size > 24 && fseek(f, 0, SEEK_SET) == 0) {
            out->resize((size_t)size);
            ok = fread(out->data(), 1, (size_t)size, f) == (size_t)size;
            if (ok) {
                unsigned long long tr[3];
                memcpy(tr, out->data() + size - 24, 24);
                ok = tr[0] == kCacheMagic && tr[1] == (unsigned long long)(size - 24) && tr[2] == fnv1a(out->data(), (size_t)size - 24);
                out->resize((size_t)size - 24);
            }
        }
    }
    fclose(f);
    if (!ok) {
        out->clear();
        (void)remove(path.c_str());                            // truncated / corrupt / older format: rebuilt and rewritten
    } else {
        (void)utimes(path.c_str(), nullptr);                   // most recently used (the eviction below goes by mtime)
    }
    return ok;
}
// keep the directory below SDFK_CACHE_MAX_MB (default 512): oldest files go first, down to three quarters of the cap
static void rtc_cache_evict(const std::string& dir) {
    static const long long cap = [] {
        const char* e = getenv("SDFK_CACHE_MAX_MB");
        const long long v = e ? atoll(e) : 512;
        return (v > 0 ? v : 512) * (1ll << 20);
    }();
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    std::vector<std::pair<long long, std::pair<std::string, long long>>> files;   // (mtime, (path, size))
    long long total = 0;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.compare(0, 5, "sdfk-") != 0 || name.compare(0, 9, "sdfk-rtc-") == 0) continue;   // (not the hand-over directories of builds in flight)
        struct stat st;
        const std::string path = dir + "/" + name;
        if (stat(path.c_str(), &st) != 0) continue;
        total += (long long)st.st_size;
        files.push_back({(long long)st.st_mtime, {path, (long long)st.st_size}});
    }
    closedir(d);
    if (total <= cap) return;
    std::sort(files.begin(), files.end());
    for (const auto& f : files) {
        if (total <= cap / 4 * 3) break;
        if (remove(f.second.first.c_str()) == 0) total -= f.second.second;
    }
}
static void rtc_cache_write(const std::string& path, const std::vector<char>& co) {
    if (path.empty() || co.empty()) return;
    const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                            // a cache that cannot be written is no error
    const unsigned long long tr[3] = {kCacheMagic, (unsigned long long)co.size(), fnv1a(co.data(), co.size())};
    bool ok = fwrite(co.data(), 1, co.size(), f) == co.size() && fwrite(tr, 1, sizeof tr, f) == sizeof tr;
    ok = (fclose(f) == 0) && ok;                               // (a short write on a full disk may only show here)
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());   // atomic: readers never see a partial file
    else rtc_cache_evict(rtc_cache_dir());
}

static std::mutex g_rtc_mu;   // hiprtc and hipModuleLoadData: one thread at a time (see BuildWorker)
// extra -D switches for the generated source (experiments): SDFK_RTC_DEFS="-DSDFK_TWAVES=2 ..." or sdfk_debug_set_rtc_defs
static std::mutex g_defs_mu;
static std::string g_rtc_defs = [] { const char* e = getenv("SDFK_RTC_DEFS"); return std::string(e ? e : ""); }();
extern "C" void sdfk_debug_set_rtc_defs(const char* defs) {
    std::lock_guard<std::mutex> lk(g_defs_mu);
    std::string rest;
    int tw = 0, rwb = 0, rwv = 0;
    const std::string all = defs ? defs : "";
    size_t pos = 0;
    while (pos < all.size()) {
        size_t sp = all.find(' ', pos);
        if (sp == std::string::npos) sp = all.size();
        const std::string tok = all.substr(pos, sp - pos);
        if (tok.compare(0, 14, "-DSDFK_TWAVES=") == 0) tw = atoi(tok.c_str() + 14);
        else if (tok.compare(0, 16, "-DSDFK_RWBRICKS=") == 0) rwb = atoi(tok.c_str() + 16);
        else if (tok.compare(0, 14, "-DSDFK_RWAVES=") == 0) rwv = atoi(tok.c_str() + 14);
        else if (!tok.empty()) rest += tok + " ";
        pos = sp + 1;
    }
    g_twaves_override = (tw >= 1 && tw <= 16) ? tw : 0;
    g_rwbricks_override = (rwb >= 1 && rwb <= 15) ? rwb : 0;
    g_rwaves_override = (rwv >= 1 && rwv <= 16) ? rwv : 0;
    g_rtc_defs = rest;
}
// geo: bricks per wave | waves per workgroup << 4 of the row-block kernel (rows_geo)
static std::vector<std::string> rtc_options(int geo) {
    const int rwb = geo & 15, rwaves = ((geo >> 4) & 0xff) ? ((geo >> 4) & 0xff) : 4;
    const bool big = (geo >> 16) & 1;
    const char* opt = getenv("SDFK_RTC_OPT");                 // experiments: "-O1" ... (the cache key carries the options)
    std::vector<std::string> o = {"--offload-arch=gfx950", (opt && opt[0] == '-') ? opt : "-O3", "-ffp-contract=off", "-std=c++17",
                                  // -fno-honor-nans: v_min/v_max without the canonicalising pre-op. -mno-amdgpu-ieee (same
                                  // flags as the hipcc build of the interpreter kernel: both flavours stay bit-identical)
                                  // keeps the device library's sincos / atan2 / pow out of line — the inliner refuses
                                  // across the attribute — which is what a 50-primitive 2-D tree wants: 296 KB of code
                                  // instead of 490 KB, 10 s of compile instead of 15 s, 1.19 vs 1.22 ms at 16385^2
                                  "-fno-honor-nans", "-mno-amdgpu-ieee",
                                  "-DSDFK_TWAVES=" + std::to_string(tile_waves()), "-DSDFK_WBRICKS=" + std::to_string(tile_wbricks()),
                                  "-DSDFK_RWBRICKS=" + std::to_string(rwb), "-DSDFK_RWAVES=" + std::to_string(rwaves)};
    if (big) {
        // Big programs (round 4): hiprtc's time grows with the square of a straight-line program, and -ftime-report on a
        // 599-instruction tree names the pass: CodeGenPrepare, 458 of 630 s (then VectorCombine, 31 of the remaining 151).
        // Without the two a row-block build takes 30 s instead of 250 at 599 instructions, line bricks 33 s at 1199
        // instead of 105 (profiles/r04_build_time.txt). CodeGenPrepare is worth 2 % on the north-star tree and 12 % on the
        // 20-primitive one (profiles/r04_nocgp.txt) — so small programs keep it — but a culled kernel without it is still
        // several times the interpreter kernel, which is what served these programs before. Same FP semantics: same bits.
        o.push_back("-mllvm");
        o.push_back("-disable-cgp");
        o.push_back("-mllvm");
        o.push_back("-disable-vector-combine");
    }
    if (const char* extra = getenv("SDFK_RTC_EXTRA")) {       // experiments: raw compiler options, space-separated
        std::string e = extra;
        size_t q = 0;
        while (q < e.size()) {
            size_t sp = e.find(' ', q);
            if (sp == std::string::npos) sp = e.size();
            if (sp > q) o.push_back(e.substr(q, sp - q));
            q = sp + 1;
        }
    }
    std::string all;
    {
        std::lock_guard<std::mutex> lk(g_defs_mu);
        all = g_rtc_defs;
    }
    size_t pos = 0;
    while (pos < all.size()) {
        size_t sp = all.find(' ', pos);
        if (sp == std::string::npos) sp = all.size();
        if (sp > pos && all.compare(pos, 2, "-D") == 0) o.push_back(all.substr(pos, sp - pos));
        pos = sp + 1;
    }
    return o;
}
static std::string rtc_option_key(int rwb) {
    std::string k;
    for (const std::string& o : rtc_options(rwb)) k += o + " ";
    return k;
}
static int rtc_compile_uncached(const std::string& src, std::vector<char>* out, std::string* log, int rwb) {
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "sdfk_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        *log = "hiprtcCreateProgram failed";
        return -1;
    }
    const std::vector<std::string> o = rtc_options(rwb);
    std::vector<const char*> opts;
    for (const std::string& x : o) opts.push_back(x.c_str());
    hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) {
        log->resize(ls);
        hiprtcGetProgramLog(prog, &(*log)[0]);
    }
    if (r != HIPRTC_SUCCESS) {
        *log = std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + *log;
        hiprtcDestroyProgram(&prog);
        return -1;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    out->resize(cs);
    hiprtcGetCode(prog, out->data());
    hiprtcDestroyProgram(&prog);
    return 0;
}
// ---- hiprtc in a child process (background builds) -------------------------------------------------------------------
// hiprtcCompileProgram holds comgr's process-wide mutex for the whole build; a dlopen of any library with HIP fat
// binaries on another thread of the same process (`import torch`) deadlocks against it — loader lock -> comgr mutex there,
// comgr mutex -> loader lock here (profiles/r03_hang_import_during_build.txt). Builds that run BESIDE the caller
// therefore run in aegolius_amd/sdfk_rtc_helper (csrc/sdfk_rtc_helper.c): no GPU, no shared lock. Builds the caller
// waits for stay in-process (the caller cannot dlopen while it waits). No helper next to the library: no background
// builds — the call waits.
extern char** environ;
static std::string rtc_helper_path() {
    static const std::string path = [] {
        if (const char* e = getenv("SDFK_RTC_HELPER")) return std::string(strcmp(e, "off") && strcmp(e, "0") ? e : "");
        Dl_info info;
        if (!dladdr((void*)&sdfk_abi_version, &info) || !info.dli_fname) return std::string();
        std::string p = info.dli_fname;
        const size_t slash = p.rfind('/');
        p = (slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/sdfk_rtc_helper";
        return access(p.c_str(), X_OK) == 0 ? p : std::string();
    }();
    return path;
}
static std::string rtc_library_path() {                        // the hiprtc THIS process uses (torch's or the system's)
    Dl_info info;
    if (!dladdr((void*)&hiprtcCompileProgram, &info) || !info.dli_fname) return std::string();
    return info.dli_fname;
}
static bool rtc_helper_available() { return !rtc_helper_path().empty() && !rtc_library_path().empty(); }
static std::atomic<bool> g_cancel_builds{false};              // set by sdfk_jit_cancel: running compiler children are killed, queued builds dropped
static int rtc_compile_external(const std::string& src, std::vector<char>* out, std::string* log, int rwb) {
    static std::atomic<unsigned> serial{0};
    const std::string helper = rtc_helper_path(), lib = rtc_library_path();
    if (helper.empty() || lib.empty()) {
        *log = "sdfk_rtc_helper is not available";
        return -2;
    }
    std::string dir = rtc_cache_dir();
    if (dir.empty()) {
        const char* t = getenv("TMPDIR");
        dir = (t && *t) ? t : "/tmp";
    }
    // the hand-over files live in a directory of their own that mkdtemp creates (mode 0700, unpredictable name): nobody
    // can plant a file or a link where the source is written or the code object is read, two processes with the same pid in
    // different namespaces that share the cache directory cannot meet, and the cache eviction skips the "sdfk-rtc-" prefix
    (void)serial;
    std::string priv = dir + "/sdfk-rtc-XXXXXX";
    if (!mkdtemp(&priv[0])) {
        *log = "cannot create a private directory under " + dir + ": " + strerror(errno);
        return -2;
    }
    const std::string srcf = priv + "/src.hip", outf = priv + "/out.co";
    auto cleanup = [&] {
        (void)remove(srcf.c_str());
        (void)remove(outf.c_str());
        (void)remove((outf + ".tmp").c_str());
        (void)remove((outf + ".log").c_str());
        (void)rmdir(priv.c_str());
    };
    {
        const int fd = open(srcf.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_CLOEXEC, 0600);
        FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
        if (!f && fd >= 0) close(fd);
        const bool ok = f && fwrite(src.data(), 1, src.size(), f) == src.size();
        if (!f || fclose(f) != 0 || !ok) {
            cleanup();
            *log = "cannot write " + srcf;
            return -2;
        }
    }
    const std::vector<std::string> o = rtc_options(rwb);
    std::vector<char*> argv = {const_cast<char*>(helper.c_str()), const_cast<char*>(lib.c_str()), const_cast<char*>(srcf.c_str()),
                               const_cast<char*>(outf.c_str())};
    for (const std::string& x : o) argv.push_back(const_cast<char*>(x.c_str()));
    argv.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addclosefrom_np(&fa, 3);          // the child inherits nothing of the GPU runtime's
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) {
        cleanup();
        *log = std::string("posix_spawn of sdfk_rtc_helper: ") + strerror(rc);
        return -2;
    }
    // a compiler that never returns (wedged inside comgr, a stale network file system) must not hold its worker thread —
    // and with it sdfk_jit_drain at interpreter exit — for ever: SDFK_RTC_TIMEOUT seconds (default 900), then it is killed
    static const double limit_s = [] { const char* e = getenv("SDFK_RTC_TIMEOUT"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 900.0; }();
    int status = 0;
    bool timed_out = false, cancelled = false;
    const auto t_spawn = std::chrono::steady_clock::now();
    for (;;) {
        const pid_t w = waitpid(pid, &status, WNOHANG);
        if (w == pid) break;
        if (w < 0 && errno != EINTR) { status = -1; break; }
        if (g_cancel_builds.load(std::memory_order_relaxed)) {  // the process is leaving (sdfk_jit_cancel): nobody will use the kernel
            cancelled = true;
            (void)kill(pid, SIGKILL);
            while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
            }
            break;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_spawn).count() > limit_s) {
            timed_out = true;
            (void)kill(pid, SIGKILL);
            while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
            }
            break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    int result = -1;
    if (cancelled) {
        *log = "build cancelled: the process is shutting down";
    } else if (timed_out) {
        *log = "sdfk_rtc_helper did not finish within " + std::to_string((long long)limit_s) + " s (SDFK_RTC_TIMEOUT) and was killed";
    } else if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
        FILE* f = fopen(outf.c_str(), "rb");
        if (f && fseek(f, 0, SEEK_END) == 0) {
            const long size = ftell(f);
            if (size > 0 && fseek(f, 0, SEEK_SET) == 0) {
                out->resize((size_t)size);
                if (fread(out->data(), 1, (size_t)size, f) == (size_t)size) result = 0;
            }
        }
        if (f) fclose(f);
        if (result) *log = "sdfk_rtc_helper left no code object";
    } else {
        *log = "sdfk_rtc_helper failed (status " + std::to_string(status) + ")";
        if (FILE* f = fopen((outf + ".log").c_str(), "rb")) {
            char buf[8192];
            const size_t n = fread(buf, 1, sizeof buf - 1, f);
            buf[n] = 0;
            *log += std::string(": ") + buf;
            fclose(f);
        }
    }
    cleanup();
    return result;
}

// *from_disk (optional): the code object came from the on-disk cache
static int rtc_compile(const std::string& src, std::vector<char>* out, std::string* log, int rwb, bool* from_disk = nullptr,
                       std::string* disk_path = nullptr, bool external = false) {
    if (from_disk) *from_disk = false;
    const std::string path = rtc_cache_path(src, rtc_option_key(rwb));
    if (rtc_cache_read(path, out)) {
        if (from_disk) *from_disk = true;
        if (disk_path) *disk_path = path;
        return 0;
    }
    // Builds run in the compiler CHILD process whenever it is there — the background ones (never hiprtc inside this process
    // while the caller is free to dlopen something: profiles/r03_hang_import_during_build.txt) and the ones the caller waits
    // for alike (ctypes releases the GIL during the wait: another Python thread that imports a HIP library would meet the same
    // lock inversion). In-process hiprtc is the last resort, and says so once.
    int rc = -2;
    if (rtc_helper_available()) rc = rtc_compile_external(src, out, log, rwb);
    if (rc == -2 && !external) {
        static std::atomic<bool> told{false};
        if (!told.exchange(true))
            fprintf(stderr, "[sdfk] compiler helper unavailable (%s): building inside this process — do not import HIP libraries on "
                            "other threads meanwhile\n", log->empty() ? "sdfk_rtc_helper not found next to libsdfk.so" : log->c_str());
        std::lock_guard<std::mutex> lk(g_rtc_mu);
        rc = rtc_compile_uncached(src, out, log, rwb);
    }
    if (rc == 0) rtc_cache_write(path, *out);
    return rc;
}

// ---- code objects (per process) and modules (per device) -----------------------------------------
static const char* const kFlavourFn[SDFK_FL_COUNT][2] = {
    {"sdfk_spec_v4", "sdfk_spec_v1"}, {"sdfk_spec_g4", "sdfk_spec_g1"}, {"sdfk_spec_t", nullptr}, {"sdfk_spec_tg", nullptr},
    {"sdfk_spec_tmask", nullptr},     {"sdfk_spec_r", nullptr},         {"sdfk_spec_rg", nullptr}, {"sdfk_spec_rmask", nullptr},
    {"sdfk_spec_r", nullptr},         {"sdfk_spec_rg", nullptr}};

// hiprtc is entered by ONE thread at a time, and never while a code object is being loaded (hipModuleLoadData):
// g_rtc_mu. Background builds are queued to one worker thread, which is drained before the interpreter / the
// library's statics (and with them hiprtc) go away: sdfk_jit_drain (Python: atexit) and the destructor below.
struct BuildWorker {
    static constexpr int kThreads = 2;                       // compiler processes that may run side by side
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> jobs;
    std::vector<std::thread> threads;
    bool stop = false;
    int busy = 0;
    void post(std::function<void()> job) {
        std::lock_guard<std::mutex> lk(mu);
        jobs.push_back(std::move(job));
        if ((int)threads.size() < kThreads && (int)threads.size() < busy + (int)jobs.size())
            threads.emplace_back([this] { loop(); });
        cv.notify_all();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [this] { return stop || !jobs.empty(); });
            if (jobs.empty()) return;                        // (stop: the queue is finished first)
            std::function<void()> job = std::move(jobs.front());
            jobs.pop_front();
            ++busy;
            lk.unlock();
            job();
            lk.lock();
            --busy;
            cv.notify_all();
        }
    }
    void drain() {                                           // wait until nothing is queued or running
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return jobs.empty() && busy == 0; });
    }
    void drop_queued() {
        std::lock_guard<std::mutex> lk(mu);
        jobs.clear();
        cv.notify_all();
    }
    ~BuildWorker() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        for (std::thread& t : threads)
            if (t.joinable()) t.join();
    }
};
static BuildWorker g_builds;
extern "C" void sdfk_jit_drain(void) { g_builds.drain(); }
// At interpreter exit: a background build nobody will use any more (a big tree evaluated once: up to a minute of hiprtc)
// must not hold the process. Queued builds are dropped, running compiler children killed, then the workers are idle.
extern "C" void sdfk_jit_cancel(void) {
    g_cancel_builds.store(true);
    g_builds.drop_queued();
    g_builds.drain();
}
static std::atomic<long long> g_compile_count{0};             // hiprtc builds this process has actually run
static std::atomic<long long> g_compile_micros{0};
extern "C" void sdfk_debug_jit_stats(int64_t* builds, double* seconds) {
    if (builds) *builds = g_compile_count.load();
    if (seconds) *seconds = (double)g_compile_micros.load() * 1e-6;
}

static std::shared_ptr<CodeObject> code_entry(const std::string& key) {
    std::lock_guard<std::mutex> lk(g_code_mu);
    std::shared_ptr<CodeObject>& e = g_code[key];
    if (!e) e = std::make_shared<CodeObject>();
    return e;
}
// run one build; the caller has moved the entry to state 1
static void code_build(const std::shared_ptr<CodeObject>& e, const std::string& src, int rwb, bool external = false) {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<char> co;
    std::string log;
    bool from_disk = false;
    std::string disk_path;
    const int rc = rtc_compile(src, &co, &log, rwb, &from_disk, &disk_path, external);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!from_disk) {
        g_compile_count++;
        g_compile_micros += (long long)(dt * 1e6);
    }
    std::lock_guard<std::mutex> lk(e->mu);
    e->build_seconds = dt;
    if (rc == 0) {
        e->co.swap(co);
        e->disk_path = disk_path;
        e->state = 2;
    } else {
        e->error = log;
        e->state = 3;
        e->failed_at = std::chrono::steady_clock::now();
    }
    e->cv.notify_all();
}
// The code object of (source, options): wait = build here (or wait for the thread that is building); !wait = make sure
// a build is under way (background thread) and return at once. A failed build is retried after 30 s at the earliest.
template <typename MakeSource>
static std::shared_ptr<CodeObject> code_get(const std::string& key, MakeSource make_source, int rwb, bool wait) {
    std::shared_ptr<CodeObject> e = code_entry(key);
    std::unique_lock<std::mutex> lk(e->mu);
    if (e->state == 3 && std::chrono::steady_clock::now() - e->failed_at > std::chrono::seconds(30)) e->state = 0;
    if (e->state == 0) {
        e->state = 1;
        lk.unlock();
        const std::string src = make_source();                 // (the program may be gone before a background build ends)
        if (wait || !rtc_helper_available()) {                 // no compiler process to hand the build to: the caller waits
            code_build(e, src, rwb);
        } else {
            g_builds.post([e, src, rwb] { code_build(e, src, rwb, true); });
        }
        return e;
    }
    if (wait) e->cv.wait(lk, [&] { return e->state != 1; });
    return e;
}
// with_flags: the build of a flavour that writes one flag bit per point (value <= threshold) instead of the field — a
// translation unit of its own (#define SDFK_FLAGS), so the field kernels carry none of it.
// with_flags is a set of build VARIANTS: bit 0 = flag-writing build (SDFK_FLAGS), bit 1 = two-row coordinates, z = 0 by
// contract (SDFK_XY: the array kernels never read a third row — sdfk_eval_device_rows2d_xy)
static std::string flavour_key(const sdfk_program* p, int flavour, int rwb, int with_flags = 0) {
    return p->key + "|f" + std::to_string(flavour) + ((with_flags & 1) ? "s" : "") + ((with_flags & 2) ? "x" : "") + "|" + rtc_option_key(rwb);
}
static std::string flavour_source(const sdfk_program* p, int flavour, int with_flags = 0) {
    return std::string((with_flags & 1) ? "#define SDFK_FLAGS 1\n" : "") + ((with_flags & 2) ? "#define SDFK_XY 1\n" : "") +
           sdfk_generate_source(g_ops, SDFK_OP_COUNT, p->code.data(), p->code.size() / 2, p->result_reg, p->sites, flavour,
                                &p->sites_all);
}

extern "C" int sdfk_program_chain_members(const sdfk_program* p) {
    return p && p->chain_mode ? p->chain_members : 0;
}
extern "C" int sdfk_program_compile_check(sdfk_program* p, size_t* code_size) {
    // every flavour this program can be launched with, each as its own translation unit (what a run would build)
    if (!p) return fail(-1, "null program");
    size_t total = 0;
    const int rwb = rows_geo(p);
    for (int f = 0; f < SDFK_FL_COUNT; ++f) {
        if (p->sites.empty() && f != SDFK_FL_PLAIN_ARRAY && f != SDFK_FL_PLAIN_GRID) continue;
        if (p->chain_mode && (f == SDFK_FL_TILE_ARRAY || f == SDFK_FL_TILE_GRID || f == SDFK_FL_TILE_MASK || f == SDFK_FL_ROWS_MASK)) continue;
        std::shared_ptr<CodeObject> e = code_get(flavour_key(p, f, rwb), [&] { return flavour_source(p, f); }, rwb, true);
        if (e->state != 2) return fail(-3, e->error);
        total += e->co.size();
    }
    if (code_size) *code_size = total;
    return 0;
}
/* Build (or fetch) ONE flavour without a GPU: 0 + seconds the build took (0 when it was already there). */
extern "C" int sdfk_program_compile_flavour(sdfk_program* p, int flavour, size_t* code_size, double* seconds) {
    if (!p) return fail(-1, "null program");
    int with_flags = 0;                                                              // build variants (see flavour_key)
    if (flavour >= 0 && (flavour & SDFK_FLAVOUR_FLAGS)) with_flags |= 1;             // the flag-writing build of the flavour
    if (flavour >= 0 && (flavour & SDFK_FLAVOUR_XY)) with_flags |= 2;                // two-row coordinates
    if (flavour >= 0) flavour &= ~(SDFK_FLAVOUR_FLAGS | SDFK_FLAVOUR_XY);
    if (flavour < 0 || flavour >= SDFK_FL_COUNT) return fail(-1, "sdfk_program_compile_flavour: unknown flavour");
    if (p->sites.empty() && flavour != SDFK_FL_PLAIN_ARRAY && flavour != SDFK_FL_PLAIN_GRID)
        return fail(-2, "sdfk_program_compile_flavour: the program has no cull sites");
    if ((with_flags & 2) && flavour != SDFK_FL_PLAIN_ARRAY && flavour != SDFK_FL_ROWS2D_ARRAY)
        return fail(-2, "sdfk_program_compile_flavour: two-row coordinates exist for the plain and the flat row-block array kernels");
    if ((with_flags & 1) && (flavour == SDFK_FL_TILE_ARRAY || flavour == SDFK_FL_TILE_GRID || flavour == SDFK_FL_TILE_MASK ||
                       flavour == SDFK_FL_ROWS_MASK))
        return fail(-2, "sdfk_program_compile_flavour: this flavour has no flag-writing build");
    const int rwb = rows_geo(p);
    const auto t0 = std::chrono::steady_clock::now();
    std::shared_ptr<CodeObject> e =
        code_get(flavour_key(p, flavour, rwb, with_flags), [&] { return flavour_source(p, flavour, with_flags); }, rwb, true);
    if (e->state != 2) return fail(-3, e->error);
    if (code_size) *code_size = e->co.size();
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

/* Test aid: build one flavour in the compiler child process (what a background build does), GPU or not. */
extern "C" int sdfk_debug_compile_external(sdfk_program* p, int flavour, size_t* code_size) {
    if (!p) return fail(-1, "null program");
    if (flavour < 0 || flavour >= SDFK_FL_COUNT) return fail(-1, "sdfk_debug_compile_external: unknown flavour");
    if (p->sites.empty() && flavour != SDFK_FL_PLAIN_ARRAY && flavour != SDFK_FL_PLAIN_GRID)
        return fail(-2, "sdfk_debug_compile_external: the program has no cull sites");
    if (!rtc_helper_available()) return fail(-9, "sdfk_rtc_helper is not next to libsdfk.so (or hiprtc cannot be located)");
    std::vector<char> co;
    std::string log;
    if (rtc_compile_external(flavour_source(p, flavour), &co, &log, rows_geo(p)) != 0) return fail(-3, log);
    if (code_size) *code_size = co.size();
    return 0;
}

// The module of one flavour on one device. wait = false: nullptr while the code object is still being built in the
// background (the caller serves this call from the interpreter kernel — same bits). *err is set on failure.
static std::shared_ptr<SpecModule> get_module(sdfk_program* p, int device, int flavour, bool wait, std::string* err,
                                              int with_flags = 0) {
    const int rwb = rows_geo(p);
    const std::string key = flavour_key(p, flavour, rwb, with_flags);
    std::shared_ptr<SpecModule> m;
    {
        std::lock_guard<std::mutex> lk(g_code_mu);
        std::shared_ptr<SpecModule>& slot = g_mods[std::make_pair(device, key)];
        if (!slot) slot = std::make_shared<SpecModule>();
        m = slot;
    }
    std::lock_guard<std::mutex> lk(m->mu);                     // per (device, flavour): loads never block other devices
    if (m->loaded) return m;
    for (int attempt = 0;; ++attempt) {
        std::shared_ptr<CodeObject> e;
        {
            std::lock_guard<std::mutex> ce(g_code_mu);
            auto it = g_code.find(key);
            if (it != g_code.end()) e = it->second;
        }
        int state = 0;
        if (e) {
            std::lock_guard<std::mutex> el(e->mu);
            state = e->state;
        }
        if (state != 2) {
            e = code_get(key, [&] { return flavour_source(p, flavour, with_flags); }, rwb, wait);
            std::lock_guard<std::mutex> el(e->mu);
            state = e->state;
        }
        if (state == 1) return nullptr;                            // still building (wait == false)
        if (state != 2) {
            std::lock_guard<std::mutex> el(e->mu);
            *err = e->error;
            m->failed = true;
            m->error = e->error;
            return m;                                              // (not marked loaded: a later call asks code_get again)
        }
        hipError_t he;
        {
            std::lock_guard<std::mutex> rl(g_rtc_mu);
            he = hipModuleLoadData(&m->mod, e->co.data());
        }
        for (int i = 0; i < 2 && he == hipSuccess; ++i)
            if (kFlavourFn[flavour][i]) he = hipModuleGetFunction(&m->fn[i], m->mod, kFlavourFn[flavour][i]);
        if (he == hipSuccess && p->chain_mode && !kFlavourFn[flavour][1]) {
            // chain-mode row-block kernels come with the pre-pass of their candidate lists (absent from -DSDFK_NO_CELLS builds)
            const bool grid_fl = flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS2D_GRID;
            if (hipModuleGetFunction(&m->fn[1], m->mod, grid_fl ? "sdfk_spec_cellsg" : "sdfk_spec_cells") != hipSuccess) {
                m->fn[1] = nullptr;
                (void)hipGetLastError();
            }
        }
        if (he == hipSuccess) break;
        // A code object that came from the on-disk cache and does not load (another driver / compiler generation, a
        // damaged file that still passed the checksum): delete the file, forget the blob and build from source once.
        bool retry = false;
        {
            std::lock_guard<std::mutex> el(e->mu);
            if (attempt == 0 && e->state == 2 && !e->disk_path.empty()) {
                (void)remove(e->disk_path.c_str());
                fprintf(stderr, "[sdfk] cached code object %s does not load (%s): rebuilding\n", e->disk_path.c_str(),
                        hipGetErrorString(he));
                e->disk_path.clear();
                e->co.clear();
                e->state = 0;
                retry = true;
            }
        }
        if (m->mod) {
            (void)hipModuleUnload(m->mod);
            m->mod = nullptr;
        }
        (void)hipGetLastError();
        if (retry) {
            wait = true;                                           // the caller gets the rebuilt kernel, not a second failure
            continue;
        }
        m->failed = true;
        m->error = std::string("hipModuleLoadData/GetFunction: ") + hipGetErrorString(he);
        *err = m->error;
        return m;
    }
    m->failed = false;
    m->loaded = true;
    return m;
}

// make sure code / params / tables of `p` are resident on the current device
static int ensure_resident(sdfk_program* p, int device, hipStream_t stream, DevState** out) {
    std::lock_guard<std::mutex> lk(p->mu);
    DevState& d = p->dev[device];
    if (!d.d_code) {
        HIPCHK(hipMalloc(&d.d_code, p->code.size() * sizeof(uint32_t)));
        HIPCHK(hipMemcpy(d.d_code, p->code.data(), p->code.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&d.d_params, std::max<size_t>(p->params.size(), 1) * sizeof(float)));
        HIPCHK(hipMalloc(&d.d_tables, std::max<size_t>(p->tables.size(), 1) * sizeof(float)));
        if (!p->tables.empty())
            HIPCHK(hipMemcpy(d.d_tables, p->tables.data(), p->tables.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (d.params_version != p->params_version) {
        // enqueued on the caller's stream (it must not overtake kernels of that stream that still read the old values)
        // and WAITED for: other streams of the device (the two slots of the host pipeline) launch right after this
        if (!p->params.empty()) {
            HIPCHK(hipMemcpyAsync(d.d_params, p->params.data(), p->params.size() * sizeof(float),
                                  hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        d.params_version = p->params_version;
    }
    *out = &d;
    return 0;
}

static inline unsigned blocks_for(long long n, int vec) {
    return (unsigned)((n + (long long)SDFK_BLOCK * vec - 1) / ((long long)SDFK_BLOCK * vec));
}

// test aid: statistics of the candidate lists of the last chain-mode launch (sdfk_debug_cells_stats)
static std::atomic<bool> g_cells_stats_on{false};
static std::mutex g_cells_stats_mu;
static long long g_cells_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" void sdfk_debug_cells_stats(int enable, long long* out8) {
    g_cells_stats_on.store(enable != 0);
    if (out8) {
        std::lock_guard<std::mutex> lk(g_cells_stats_mu);
        for (int i = 0; i < 8; ++i) out8[i] = g_cells_stats[i];
        for (int i = 0; i < 8; ++i) g_cells_stats[i] = 0;
    }
}
// ---- candidate lists of chain-mode programs (sdfk_codegen.cpp: sdfk_cells / sdfk_cellpass / sdfk_spec_cells) ------------------
struct CellLevelH {           // mirrors sdfk_celllevel
    unsigned lx, ly, lz, ncx, ncy, ncz, xoff, pad;
};
struct CellsArg {             // mirrors sdfk_cells
    CellLevelH lv;
    const void *sph, *span, *cand;
    unsigned enabled, ncells;
};
struct CellPassArg {          // mirrors sdfk_cellpass
    CellLevelH lv, parent;
    void *sph, *span;
    const void *psph, *pspan;
    void* cand;
    unsigned* head;
    unsigned base, shard_cap;
    unsigned ncells, pad0;
    float inflate, pad;
};
static bool parse3(const char* e, unsigned* v) {
    int a = 0, b = 0, c = 0;
    if (!e || sscanf(e, "%d,%d,%d", &a, &b, &c) != 3 || a < 0 || b < 0 || c < 0 || a > 12 || b > 12 || c > 12) return false;
    v[0] = (unsigned)a; v[1] = (unsigned)b; v[2] = (unsigned)c;
    return true;
}
static CellLevelH cell_level(const RowGeom& rg, const unsigned l[3]) {
    CellLevelH lv{};
    lv.lx = l[0]; lv.ly = l[1]; lv.lz = l[2];
    const long long planes = rg.prow ? ((rg.R - rg.seg0) + rg.prow - 1) / rg.prow : 0;
    lv.xoff = rg.seg0 > 0 ? (1u << lv.lx) : 0u;
    lv.ncx = planes > 0 ? (unsigned)(((long long)lv.xoff + planes - 1) >> lv.lx) + 1u : 1u;
    lv.ncy = ((std::max(rg.bpp, rg.nb0) - 1u) >> lv.ly) + 1u;
    lv.ncz = ((rg.nchunk - 1u) >> lv.lz) + 1u;
    return lv;
}
// Lists for this launch: sizes the levels, (re)allocates the stream's scratch, enqueues the pre-pass (coarse level, then
// fine) on `stream` and fills what the row-block kernel is handed. cells_fn: sdfk_spec_cells / sdfk_spec_cellsg of the
// module, `src`: its first kernel arguments after PRM / TAB (array: co, stride; grid: the SrcGrid), n_src of them.
static int prepare_cells(sdfk_program* p, DevState* d, hipFunction_t cells_fn, const RowGeom& rg, void** src, int n_src,
                         const float* prm, const float* tab, hipStream_t stream, CellsArg* out) {
    memset(out, 0, sizeof *out);
    static const int min_members = [] { const char* e = getenv("SDFK_CELLS_MIN"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 17; }();
    static const bool off = [] { const char* e = getenv("SDFK_CELLS"); return e && e[0] == '0'; }();
    if (!cells_fn || off || p->chain_members < min_members) return 0;
    const bool is3d = rg.prow < (unsigned long long)rg.R;
    unsigned lf[3] = {is3d ? 3u : 0u, is3d ? 1u : 2u, is3d ? 0u : 1u};            // 8 planes x 32 rows x 32 points | 64 rows x 64 points
    unsigned lc[3] = {lf[0] + (is3d ? 2u : 0u), lf[1] + 2u, lf[2] + 2u};          // 4 x 4 x 4 (4 x 4) fine cells
    static const char* e_fine = getenv("SDFK_CELL_FINE");
    static const char* e_coarse = getenv("SDFK_CELL_COARSE");
    (void)parse3(e_fine, lf);
    bool coarse = p->chain_members >= 128;
    if (e_coarse) coarse = parse3(e_coarse, lc);
    if (coarse && (lc[0] < lf[0] || lc[1] < lf[1] || lc[2] < lf[2])) coarse = false;
    const CellLevelH fine = cell_level(rg, lf);
    const CellLevelH crs = coarse ? cell_level(rg, lc) : CellLevelH{};
    const unsigned long long nf = (unsigned long long)fine.ncx * fine.ncy * fine.ncz;
    const unsigned long long nc = coarse ? (unsigned long long)crs.ncx * crs.ncy * crs.ncz : 0ull;
    if (nf == 0 || nf > 0x3fffffffull || nc > 0x3fffffffull) return 0;
    // pool: room for 48 entries per fine cell and 1024 per coarse cell (measured lists: a handful / a few hundred); a cell
    // that finds the pool full makes its bricks probe every member — slower, never wrong
    // Pool of list entries, per level 256 shards with an allocation head each (sdfk_cells_kernel). The coarse level can
    // never run out — a shard holds every member for each of its cells —; the fine level gets 256 entries per cell plus
    // slack (measured lists: a handful to a few dozen entries, a few hundred in scenes where thousands of members overlap).
    const unsigned long long members = (unsigned long long)p->chain_members, shards = 256;
    unsigned long long cshard = ((nc + shards - 1) / shards) * members;
    // (a fine shard serves ceil(cells / 256) cells: every member for each of them, or the budget — but never less than one
    //  whole list)
    unsigned long long fshard = std::max(members, std::min(((nf + shards - 1) / shards) * members, (256ull * nf + (16ull << 20) + shards - 1) / shards));
    if (const char* e = getenv("SDFK_CELLS_POOL")) {             // (tests: a pool too small for the lists)
        const long long v = atoll(e);
        if (v > 0) fshard = std::min<unsigned long long>(fshard, (unsigned long long)v);
    }
    if (shards * (cshard + fshard) > 0x3fffffffull) return 0;   // (no lists: still correct)
    const unsigned long long cap = shards * (cshard + fshard);
    const size_t o_fsph = 0, o_fspan = o_fsph + 16 * nf, o_csph = o_fspan + 8 * nf, o_cspan = o_csph + 16 * nc,
                 o_head = (o_cspan + 8 * nc + 63) & ~(size_t)63, o_pool = o_head + 2 * 64 * shards, total = o_pool + 4 * cap + 64;
    CellScratch* cs;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        cs = &d->cells[stream];
    }
    if (cs->bytes < total) {
        if (cs->buf) {
            HIPCHK(hipStreamSynchronize(stream));              // (the stream's earlier launches read the old buffer)
            (void)hipFree(cs->buf);
            cs->buf = nullptr;
            cs->bytes = 0;
        }
        if (hipMalloc(&cs->buf, total + total / 4) != hipSuccess) { (void)hipGetLastError(); return 0; }   // (no lists: still correct)
        cs->bytes = total + total / 4;
    }
    char* b = cs->buf;
    HIPCHK(hipMemsetAsync(b + o_head, 0, 2 * 64 * shards, stream));
    CellPassArg cp{};
    cp.cand = b + o_pool;
    if (coarse) {
        // a coarse cell answers for 1.3 x its circumsphere: room for the circumspheres of the fine cells inside it
        cp.lv = crs; cp.parent = CellLevelH{}; cp.sph = b + o_csph; cp.span = b + o_cspan; cp.psph = nullptr; cp.pspan = nullptr;
        cp.ncells = (unsigned)nc; cp.inflate = 1.3f;
        cp.head = reinterpret_cast<unsigned*>(b + o_head);
        cp.base = 0u; cp.shard_cap = (unsigned)cshard;
        std::vector<void*> args = {(void*)&prm, (void*)&tab};
        for (int i = 0; i < n_src; ++i) args.push_back(src[i]);
        RowGeom g2 = rg;
        args.push_back(&g2);
        args.push_back(&cp);
        HIPCHK(hipModuleLaunchKernel(cells_fn, (unsigned)((nc + 3) / 4), 1, 1, 256, 1, 1, 0, stream, args.data(), nullptr));
    }
    cp.lv = fine; cp.parent = coarse ? crs : CellLevelH{}; cp.sph = b + o_fsph; cp.span = b + o_fspan;
    cp.psph = coarse ? b + o_csph : nullptr; cp.pspan = coarse ? b + o_cspan : nullptr;
    cp.ncells = (unsigned)nf; cp.inflate = 1.0f;
    cp.head = reinterpret_cast<unsigned*>(b + o_head + 64 * shards);
    cp.base = (unsigned)(shards * cshard); cp.shard_cap = (unsigned)fshard;
    {
        std::vector<void*> args = {(void*)&prm, (void*)&tab};
        for (int i = 0; i < n_src; ++i) args.push_back(src[i]);
        RowGeom g2 = rg;
        args.push_back(&g2);
        args.push_back(&cp);
        HIPCHK(hipModuleLaunchKernel(cells_fn, (unsigned)((nf + 3) / 4), 1, 1, 256, 1, 1, 0, stream, args.data(), nullptr));
    }
    static const bool trace = [] { const char* e = getenv("SDFK_CELLS_TRACE"); return e && e[0] == '1'; }();
    if (trace || g_cells_stats_on.load()) {                      // (debug: synchronises and reads the lists' statistics back)
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<uint2> sp(nf);
        unsigned head = 0;
        HIPCHK(hipMemcpy(sp.data(), b + o_fspan, 8 * nf, hipMemcpyDeviceToHost));
        {
            std::vector<unsigned> heads(2 * 16 * shards);
            HIPCHK(hipMemcpy(heads.data(), b + o_head, 2 * 64 * shards, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < heads.size(); i += 16) head += heads[i];
        }
        unsigned long long sum = 0, all = 0, empty = 0, mx = 0;
        for (const uint2& x : sp) {
            if (x.y == 0xffffffffu) ++all;
            else { sum += x.y; mx = std::max<unsigned long long>(mx, x.y); if (x.y == 0) ++empty; }
        }
        {
            std::lock_guard<std::mutex> lk(g_cells_stats_mu);
            g_cells_stats[0] = (long long)nf; g_cells_stats[1] = (long long)nc; g_cells_stats[2] = (long long)head;
            g_cells_stats[3] = (long long)cap; g_cells_stats[4] = (long long)sum; g_cells_stats[5] = (long long)mx;
            g_cells_stats[6] = (long long)all; g_cells_stats[7] = (long long)empty;
        }
        if (trace) fprintf(stderr, "[sdfk cells] %d members: fine %ux%ux%u = %llu cells (2^%u planes x 2^%u blocks x 2^%u windows), coarse %llu; pool %u of %llu entries; "
                "fine lists: mean %.1f max %llu, %llu without a list, %llu empty\n", p->chain_members, fine.ncx, fine.ncy, fine.ncz, nf, fine.lx, fine.ly, fine.lz, nc,
                head, cap, (double)sum / (double)std::max<unsigned long long>(1, nf - all - empty), mx, all, empty);
    }
    out->lv = fine;
    out->sph = b + o_fsph;
    out->span = b + o_fspan;
    out->cand = b + o_pool;
    out->enabled = 1u;
    out->ncells = (unsigned)nf;
    return 0;
}

// flat: the caller states that the rows of the array are rows of a flat grid (z = 0, rows along y); grids know it
static int run(sdfk_program* p, const SrcArray* arr, const SrcGrid* grid, long long n, float* d_out, void* stream_,
               int mode, bool vec_ok, long long row_len = 0, const float* aux = nullptr, long long aux_stride = 0,
               bool flat = false, long long plane_rows = 0, long long plane_phase = 0, unsigned* d_flags = nullptr,
               unsigned thr_key = 0, bool xy = false) {
    if (!p) return fail(-1, "null program");
    if (n < 0) return fail(-1, "negative point count");
    if (p->n_aux > 0 && (!aux || aux_stride < n))
        return fail(-1, "this program reads auxiliary fields (staged evaluation): use sdfk_eval_device_aux / sdfk_eval_grid_aux");
    if (n == 0) return 0;
    if (mode == SDFK_MODE_AUTO) mode = g_default_mode;
    // flags instead of the field (fused selection): the specialised plain / row-block kernels only — the call waits for
    // their build instead of starting on the interpreter kernel
    if (d_flags && (mode == SDFK_MODE_AUTO || mode == SDFK_MODE_INTERPRET)) mode = SDFK_MODE_SPECIALIZED;
    // two-row coordinates (z = 0 by contract): builds of the plain and the flat row-block array kernels that never touch a
    // third row; the interpreter kernel has no such build, so these calls wait for the specialised kernel too
    if (xy) {
        if (!arr || p->n_aux > 0) return fail(-1, "two-row coordinates: array source, no auxiliary fields");
        if (mode == SDFK_MODE_AUTO || mode == SDFK_MODE_INTERPRET) mode = SDFK_MODE_SPECIALIZED;
    }
    hipStream_t stream = (hipStream_t)stream_;
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    DevState* d = nullptr;
    int rc = ensure_resident(p, device, stream, &d);
    if (rc) return rc;

    // split into a 4-wide body and a scalar tail
    long long n4 = vec_ok ? (n / 4) * 4 : 0;
    long long tail = n - n4;

    // Build time bounds (programs that are not chains: those are table-driven and build in under a second whatever their
    // size). hiprtc's time grows faster than the program — profiles/r04_build_time.txt, left-deep smooth-union chains on
    // the build container's CPU: row blocks 1 / 4 / 8 / 21 / 102 s at 29 / 89 / 179 / 299 / 449 instructions with the full
    // pipeline; beyond SDFK_BIG_PROGRAM (300) instructions builds run without CodeGenPrepare and VectorCombine
    // (rtc_options): row blocks 21 / 30 / 49 / 85 s at 449 / 599 / 899 / 1199, line bricks 8 / 11 / 22 / 33 s, the plain
    // kernel 4 / 6 / 12 / 25 s (on the GPU boxes' CPUs less than half of that). With skip bits for 512 sites a row-block
    // kernel is 2-3 x a line-brick one on these programs, so both limits are the same now:
    //   row blocks up to SDFK_ROWS_LIMIT instructions (1200), line bricks (or, for unaligned arrays, the plain kernel) up to
    //   SDFK_SPECIALIZE_LIMIT (1200); beyond that AUTO stays on the interpreter kernel, which needs no compilation.
    // A background build that is still running when the process leaves is killed (sdfk_jit_cancel).
    // MODE_SPECIALIZED / NOCULL always build (the caller asked for the kernel and waits), with the same choice of flavour.
    static const long long spec_limit = [] {
        const char* e = getenv("SDFK_SPECIALIZE_LIMIT");
        const long long v = e ? atoll(e) : 1200;
        return v > 0 ? v : 1200;
    }();
    static const long long rows_limit = [] {
        const char* e = getenv("SDFK_ROWS_LIMIT");
        const long long v = e ? atoll(e) : 1200;
        return v > 0 ? v : 1200;
    }();
    if (mode == SDFK_MODE_AUTO && (long long)(p->code.size() / 2) > spec_limit && p->interp_ok && !p->chain_mode && !d_flags)
        mode = SDFK_MODE_INTERPRET;

    // Which flavour does this call launch? (row blocks > line bricks > plain; NOCULL and programs without sites: plain)
    RowGeom rg;
    int flavour = arr ? SDFK_FL_PLAIN_ARRAY : SDFK_FL_PLAIN_GRID;
    const long long grow = grid ? (grid->n2 > 1 ? (long long)grid->n2 : (long long)grid->n1) : 0;
    if (!p->sites.empty() && mode != SDFK_MODE_NOCULL && mode != SDFK_MODE_INTERPRET) {
        // (chain mode: row blocks of ONE plane each — the cells of its candidate lists are boxes of the grid)
        if (arr && rows_geometry(n, row_len, &rg, (flat || d_flags) ? 0 : plane_rows, plane_phase, p->chain_mode))
            flavour = (flat || xy) ? SDFK_FL_ROWS2D_ARRAY : SDFK_FL_ROWS_ARRAY;   // rows need no alignment beyond 4 bytes
        else if (arr && vec_ok && !p->chain_mode && !d_flags && !xy) flavour = SDFK_FL_TILE_ARRAY;
        else if (grid && grid->start % grow == 0 &&
                 rows_geometry(n, grow, &rg, (grid->n2 > 1 && !d_flags) ? (long long)grid->n1 : 0,    // (flags: the slot layout
                               grid->n2 > 1 ? (grid->start / grow) % (long long)grid->n1 : 0, p->chain_mode))   //  knows blocks of 16 rows)
            flavour = grid->n2 > 1 ? SDFK_FL_ROWS_GRID : SDFK_FL_ROWS2D_GRID;
        else if (grid && vec_ok && !p->chain_mode && !d_flags) flavour = SDFK_FL_TILE_GRID;
        // (too big for a row-block build within the budget: the line-brick kernel where the call allows it, else un-culled)
        if (!p->chain_mode && (long long)(p->code.size() / 2) > rows_limit &&
            (flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS2D_ARRAY || flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS2D_GRID)) {
            const bool is_arr = arr != nullptr;
            if (vec_ok && !d_flags && !xy) flavour = is_arr ? SDFK_FL_TILE_ARRAY : SDFK_FL_TILE_GRID;
            else if (!d_flags) flavour = is_arr ? SDFK_FL_PLAIN_ARRAY : SDFK_FL_PLAIN_GRID;
        }
    }
    std::shared_ptr<SpecModule> sk;
    if (mode != SDFK_MODE_INTERPRET) {
        // AUTO: while hiprtc is still building this flavour (background thread, SDFK_ASYNC_JIT=0 turns that off) the
        // call is served by the interpreter kernel — the same device functions, the same bits — and later calls
        // switch over. SPECIALIZED / NOCULL always wait for the build.
        static const bool async_jit = [] { const char* e = getenv("SDFK_ASYNC_JIT"); return !(e && e[0] == '0'); }();
        const bool wait = mode != SDFK_MODE_AUTO || !p->interp_ok || !async_jit;
        std::string err;
        sk = get_module(p, device, flavour, wait, &err, (d_flags ? 1 : 0) | (xy ? 2 : 0));
        if (sk && sk->failed) {
            if (mode == SDFK_MODE_SPECIALIZED || !p->interp_ok)
                return fail(-3, "specialised kernel unavailable: " + err);
            static bool warned = false;
            if (!warned) {
                fprintf(stderr, "[sdfk] hiprtc specialisation failed, using the interpreter kernel: %s\n", err.c_str());
                warned = true;
            }
            sk.reset();
        }
    }
    if (!sk && !p->interp_ok)
        return fail(-4, "program needs more registers than the interpreter kernel has (use the specialised mode)");

    const float* prm = d->d_params;
    const float* tab = d->d_tables;
    if (sk) {
        if (flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS2D_ARRAY ||
            flavour == SDFK_FL_ROWS2D_GRID) {
            const unsigned per_tile = (unsigned)(rows_waves(p) * rows_wbricks(p));
            const unsigned tiles = ((rg.nbricks + per_tile - 1) / per_tile + 127u) & ~127u;   // whole rounds of 8 XCDs x SDFK_XGROUP = 16 tiles (sdfk_codegen.cpp)
            const unsigned rthreads = 64u * (unsigned)rows_waves(p);
            // (chain-mode builds take one more argument, their candidate lists: prepare_cells; fn[1] = the pre-pass kernel)
            CellsArg cells{};
            const bool with_cells = p->chain_mode && sk->fn[1] != nullptr;
            if (arr) {
                const float* co = arr->co;
                long long stride = arr->stride;
                if (with_cells) {
                    void* src[] = {&co, &stride};
                    rc = prepare_cells(p, d, sk->fn[1], rg, src, 2, prm, tab, stream, &cells);
                    if (rc) return rc;
                }
                void* args[] = {&prm, &tab, &co, &stride, &rg, &d_out, &d_flags, &thr_key, &cells};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], tiles, 1, 1, rthreads, 1, 1, 0, stream, args, nullptr));
            } else {
                // whole grid rows (x-slabs of a sharded evaluation always are); rows along the third axis, or along
                // the second one when the grid is flat (n2 == 1)
                SrcGrid g = *grid;
                rg.row0 = grid->start / grow;
                rg.yrows = grid->n2 > 1 ? 0 : 1;
                if (with_cells) {
                    void* src[] = {&g};
                    rc = prepare_cells(p, d, sk->fn[1], rg, src, 1, prm, tab, stream, &cells);
                    if (rc) return rc;
                }
                void* args[] = {&prm, &tab, &g, &rg, &d_out, &d_flags, &thr_key, &cells};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], tiles, 1, 1, rthreads, 1, 1, 0, stream, args, nullptr));
            }
            return 0;
        }
        if (flavour == SDFK_FL_TILE_ARRAY || flavour == SDFK_FL_TILE_GRID) {
            // brick-culling tile kernel: handles the ragged end itself
            const unsigned tiles = (unsigned)((n + tile_points() - 1) / tile_points());
            if (arr) {
                const float* co = arr->co;
                long long stride = arr->stride;
                void* args[] = {&prm, &tab, &co, &stride, &n, &d_out};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], tiles, 1, 1, tile_threads(), 1, 1, 0, stream, args, nullptr));
            } else {
                SrcGrid g = *grid;
                void* args[] = {&prm, &tab, &g, &n, &d_out};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], tiles, 1, 1, tile_threads(), 1, 1, 0, stream, args, nullptr));
            }
            return 0;
        }
        if (arr) {
            const float* co = arr->co;
            long long stride = arr->stride;
            if (n4) {
                long long off = 0;
                void* args[] = {&prm, &tab, &co, &stride, &off, &n4, &d_out, &aux, &aux_stride, &d_flags, &thr_key};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], blocks_for(n4, 4), 1, 1, SDFK_BLOCK, 1, 1, 0, stream, args,
                                             nullptr));
            }
            if (tail) {
                long long off = n4;
                void* args[] = {&prm, &tab, &co, &stride, &off, &tail, &d_out, &aux, &aux_stride, &d_flags, &thr_key};
                HIPCHK(hipModuleLaunchKernel(sk->fn[1], blocks_for(tail, 1), 1, 1, SDFK_BLOCK, 1, 1, 0, stream, args,
                                             nullptr));
            }
        } else {
            SrcGrid g = *grid;
            if (n4) {
                long long off = 0;
                void* args[] = {&prm, &tab, &g, &off, &n4, &d_out, &aux, &aux_stride, &d_flags, &thr_key};
                HIPCHK(hipModuleLaunchKernel(sk->fn[0], blocks_for(n4, 4), 1, 1, SDFK_BLOCK, 1, 1, 0, stream, args,
                                             nullptr));
            }
            if (tail) {
                long long off = n4;
                void* args[] = {&prm, &tab, &g, &off, &tail, &d_out, &aux, &aux_stride, &d_flags, &thr_key};
                HIPCHK(hipModuleLaunchKernel(sk->fn[1], blocks_for(tail, 1), 1, 1, SDFK_BLOCK, 1, 1, 0, stream, args,
                                             nullptr));
            }
        }
        return 0;
    }
    // interpreter
    if (d_flags) return fail(-3, "fused selection needs the specialised kernels");
    const int n_instr = (int)(p->code.size() / 2);
    const long long zero = 0;
    auto launch = [&](auto src, int vec, long long off, long long cnt) {
        using SRC = decltype(src);
        const dim3 grid(blocks_for(cnt, vec)), block(SDFK_BLOCK);
#define SDFK_INTERP_GO(VEC, NC, NV) hipLaunchKernelGGL((sdfk_interp_kernel<VEC, NC, NV, SRC>), grid, block, 0, stream, d->d_code, \
                                                        n_instr, prm, tab, src, off, cnt, d_out, p->result_reg, aux, aux_stride)
        if (vec == 4) {
            if (p->interp_small) SDFK_INTERP_GO(4, SDFK_NC_SMALL, SDFK_NV_SMALL);
            else SDFK_INTERP_GO(4, SDFK_NC, SDFK_NV);
        } else {
            if (p->interp_small) SDFK_INTERP_GO(1, SDFK_NC_SMALL, SDFK_NV_SMALL);
            else SDFK_INTERP_GO(1, SDFK_NC, SDFK_NV);
        }
#undef SDFK_INTERP_GO
    };
    if (arr) {
        if (n4) launch(*arr, 4, zero, n4);
        if (tail) launch(*arr, 1, n4, tail);
    } else {
        if (n4) launch(*grid, 4, zero, n4);
        if (tail) launch(*grid, 1, n4, tail);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int sdfk_eval_device(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, float* d_out,
                                void* stream, int mode) {
    if (!d_co || !d_out) return fail(-1, "sdfk_eval_device: null device pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device: row stride smaller than the point count");
    SrcArray a = {d_co, (long long)row_stride};
    bool vec_ok = aligned16(d_co) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok);
}

extern "C" int sdfk_eval_device_rows(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                                     float* d_out, void* stream, int mode) {
    if (!d_co || !d_out) return fail(-1, "sdfk_eval_device_rows: null device pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device_rows: row stride smaller than the point count");
    if (row_len < 1 || (n > 0 && n % row_len != 0))
        return fail(-1, "sdfk_eval_device_rows: the point count is not a multiple of the row length");
    SrcArray a = {d_co, (long long)row_stride};
    bool vec_ok = aligned16(d_co) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok, row_len);
}

/* include/sdfk.h: two coordinate rows, z = 0 by contract */
extern "C" int sdfk_eval_device_rows2d_xy(sdfk_program* p, const float* d_xy, int64_t n, int64_t row_stride, int64_t row_len,
                                          float* d_out, void* stream, int mode) {
    if (!d_xy || !d_out) return fail(-1, "sdfk_eval_device_rows2d_xy: null device pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device_rows2d_xy: row stride smaller than the point count");
    if (row_len < 0 || (row_len > 0 && n > 0 && n % row_len != 0))
        return fail(-1, "sdfk_eval_device_rows2d_xy: the point count is not a multiple of the row length");
    SrcArray a = {d_xy, (long long)row_stride};
    bool vec_ok = aligned16(d_xy) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok, row_len, nullptr, 0, row_len > 0, 0, 0, nullptr, 0, true);
}

extern "C" int sdfk_eval_device_rows3d(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                                       int64_t plane_rows, int64_t first_row_in_plane, float* d_out, void* stream, int mode) {
    if (!d_co || !d_out) return fail(-1, "sdfk_eval_device_rows3d: null device pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device_rows3d: row stride smaller than the point count");
    if (row_len < 1 || (n > 0 && n % row_len != 0))
        return fail(-1, "sdfk_eval_device_rows3d: the point count is not a multiple of the row length");
    if (plane_rows < 0 || first_row_in_plane < 0 || (plane_rows > 0 && first_row_in_plane >= plane_rows))
        return fail(-1, "sdfk_eval_device_rows3d: plane_rows >= 0 and 0 <= first_row_in_plane < plane_rows");
    SrcArray a = {d_co, (long long)row_stride};
    bool vec_ok = aligned16(d_co) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok, row_len, nullptr, 0, false, plane_rows, first_row_in_plane);
}

extern "C" int sdfk_eval_device_rows2d(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                                       float* d_out, void* stream, int mode) {
    if (!d_co || !d_out) return fail(-1, "sdfk_eval_device_rows2d: null device pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device_rows2d: row stride smaller than the point count");
    if (row_len < 1 || (n > 0 && n % row_len != 0))
        return fail(-1, "sdfk_eval_device_rows2d: the point count is not a multiple of the row length");
    SrcArray a = {d_co, (long long)row_stride};
    bool vec_ok = aligned16(d_co) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok, row_len, nullptr, 0, true);
}

extern "C" int sdfk_eval_device_aux(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, const float* d_aux,
                                    int n_aux, int64_t aux_stride, float* d_out, void* stream, int mode) {
    if (!p || !d_co || !d_out) return fail(-1, "sdfk_eval_device_aux: null pointer");
    if (row_stride < n) return fail(-1, "sdfk_eval_device_aux: row stride smaller than the point count");
    if (n_aux < p->n_aux) return fail(-1, "sdfk_eval_device_aux: the program reads more auxiliary fields than were passed");
    SrcArray a = {d_co, (long long)row_stride};
    bool vec_ok = aligned16(d_co) && aligned16(d_out) && (row_stride % 4 == 0);
    return run(p, &a, nullptr, n, d_out, stream, mode, vec_ok, 0, d_aux, aux_stride);
}

extern "C" int sdfk_debug_row_masks(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                                    uint64_t* d_masks, int64_t* n_bricks, int* brick_rows, void* stream_) {
    if (!p || !d_co) return fail(-1, "sdfk_debug_row_masks: null argument");
    if (p->sites.empty()) return fail(-2, "sdfk_debug_row_masks: program has no cull sites");
    if (p->chain_mode) return fail(-2, "sdfk_debug_row_masks: a chain-mode program keeps lists of surviving children, not mask words");
    RowGeom rg;
    if (row_stride < n || !rows_geometry(n, row_len, &rg))
        return fail(-1, "sdfk_debug_row_masks: the row-block kernel does not take this shape");
    if (n_bricks) *n_bricks = rg.nbricks;
    if (brick_rows) *brick_rows = 16;
    if (!d_masks) return 0;                      // size query
    hipStream_t stream = (hipStream_t)stream_;
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    DevState* d = nullptr;
    int rc = ensure_resident(p, device, stream, &d);
    if (rc) return rc;
    std::string err;
    std::shared_ptr<SpecModule> sk = get_module(p, device, SDFK_FL_ROWS_MASK, true, &err);
    if (!sk || sk->failed) return fail(-3, "specialised kernel unavailable: " + err);
    const float* prm = d->d_params;
    const float* tab = d->d_tables;
    long long stride = row_stride;
    void* args[] = {&prm, &tab, &d_co, &stride, &rg, &d_masks};
    const unsigned per_tile = (unsigned)(rows_waves(p) * rows_wbricks(p));
    HIPCHK(hipModuleLaunchKernel(sk->fn[0], (rg.nbricks + per_tile - 1) / per_tile, 1, 1, 64 * rows_waves(p), 1, 1, 0, stream,
                                 args, nullptr));
    return 0;
}

extern "C" int sdfk_debug_brick_masks(sdfk_program* p, const float* d_co, int64_t n, int64_t row_stride,
                                      uint64_t* d_masks, void* stream_) {
    if (!p || !d_co || !d_masks) return fail(-1, "sdfk_debug_brick_masks: null argument");
    if (!(aligned16(d_co) && row_stride % 4 == 0 && row_stride >= n && n > 0))
        return fail(-1, "sdfk_debug_brick_masks: needs 16-byte aligned rows");
    if (p->sites.empty()) return fail(-2, "sdfk_debug_brick_masks: program has no cull sites");
    if (p->chain_mode) return fail(-2, "sdfk_debug_brick_masks: chain-mode programs have no line-brick flavour");
    hipStream_t stream = (hipStream_t)stream_;
    int device = 0;
    HIPCHK(hipGetDevice(&device));
    DevState* d = nullptr;
    int rc = ensure_resident(p, device, stream, &d);
    if (rc) return rc;
    std::string err;
    std::shared_ptr<SpecModule> sk = get_module(p, device, SDFK_FL_TILE_MASK, true, &err);
    if (!sk || sk->failed) return fail(-3, "specialised kernel unavailable: " + err);
    const float* prm = d->d_params;
    const float* tab = d->d_tables;
    long long stride = row_stride, nn = n;
    void* args[] = {&prm, &tab, &d_co, &stride, &nn, &d_masks};
    HIPCHK(hipModuleLaunchKernel(sk->fn[0], (unsigned)((n + tile_points() - 1) / tile_points()), 1, 1, tile_threads(),
                                 1, 1, 0, stream, args, nullptr));
    return 0;
}

// ---- grids -------------------------------------------------------------------------------------
extern "C" int sdfk_linspace_f32(double lo, double hi, int64_t n, float* out) {
    if (n < 0 || (n > 0 && !out)) return fail(-1, "sdfk_linspace_f32: bad arguments");
    if (n == 0) return 0;
    if (n == 1) {
        out[0] = (float)lo;
        return 0;
    }
    const double div = (double)(n - 1);
    const double delta = hi - lo;
    const double step = delta / div;
    for (int64_t i = 0; i < n; ++i) {
        // numpy: y = arange(n) * step + start  (step != 0) ; y = arange(n)/div * delta + start (step == 0)
        volatile double t = (step != 0.0) ? (double)i * step : ((double)i / div) * delta;
        out[i] = (float)(t + lo);
    }
    out[n - 1] = (float)hi;
    return 0;
}

struct AxisTables {
    float* d = nullptr;
    ~AxisTables() {
        if (d) (void)hipFree(d);
    }
};

static int upload_axes(const float* ax0, int64_t n0, const float* ax1, int64_t n1, const float* ax2, int64_t n2,
                       hipStream_t stream, AxisTables* t, SrcGrid* g, int64_t start) {
    if (!ax0 || !ax1 || !ax2 || n0 < 1 || n1 < 1 || n2 < 1) return fail(-1, "grid axes missing or empty");
    if (n1 > 0x7fffffff || n2 > 0x7fffffff) return fail(-1, "grid axis too long");
    HIPCHK(hipMalloc(&t->d, (size_t)(n0 + n1 + n2) * sizeof(float)));
    HIPCHK(hipMemcpyAsync(t->d, ax0, (size_t)n0 * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(t->d + n0, ax1, (size_t)n1 * sizeof(float), hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(t->d + n0 + n1, ax2, (size_t)n2 * sizeof(float), hipMemcpyHostToDevice, stream));
    g->ax0 = t->d;
    g->ax1 = t->d + n0;
    g->ax2 = t->d + n0 + n1;
    g->n1 = (unsigned)n1;
    g->n2 = (unsigned)n2;
    g->start = start;
    return 0;
}

extern "C" int sdfk_eval_grid(sdfk_program* p, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                              const float* ax2, int64_t n2, int64_t start, int64_t count, float* d_out, void* stream,
                              int mode) {
    if (!d_out) return fail(-1, "sdfk_eval_grid: null output");
    if (start < 0 || count < 0 || start + count > n0 * n1 * n2) return fail(-1, "sdfk_eval_grid: range outside the grid");
    AxisTables t;
    SrcGrid g;
    int rc = upload_axes(ax0, n0, ax1, n1, ax2, n2, (hipStream_t)stream, &t, &g, start);
    if (rc) return rc;
    rc = run(p, nullptr, &g, count, d_out, stream, mode, aligned16(d_out));
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));  // the axis tables are freed on return
    return 0;
}

extern "C" int sdfk_eval_grid_aux(sdfk_program* p, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                                  const float* ax2, int64_t n2, int64_t start, int64_t count, const float* d_aux, int n_aux,
                                  int64_t aux_stride, float* d_out, void* stream, int mode) {
    if (!p || !d_out) return fail(-1, "sdfk_eval_grid_aux: null pointer");
    if (start < 0 || count < 0 || start + count > n0 * n1 * n2) return fail(-1, "sdfk_eval_grid_aux: range outside the grid");
    if (n_aux < p->n_aux) return fail(-1, "sdfk_eval_grid_aux: the program reads more auxiliary fields than were passed");
    AxisTables t;
    SrcGrid g;
    int rc = upload_axes(ax0, n0, ax1, n1, ax2, n2, (hipStream_t)stream, &t, &g, start);
    if (rc) return rc;
    rc = run(p, nullptr, &g, count, d_out, stream, mode, aligned16(d_out), 0, d_aux, aux_stride);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));  // the axis tables are freed on return
    return 0;
}

// Host-buffer convenience for grids: evaluate flat indices [start, start+count) of the grid in device chunks
// and copy the field back; no coordinate array ever exists (host or device).
extern "C" int sdfk_eval_grid_host(sdfk_program* p, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                                   const float* ax2, int64_t n2, int64_t start, int64_t count, float* out, int device,
                                   int mode) {
    if (!p) return fail(-1, "null program");
    if (count < 0 || (count > 0 && !out)) return fail(-1, "sdfk_eval_grid_host: bad arguments");
    if (start < 0 || start + count > n0 * n1 * n2) return fail(-1, "sdfk_eval_grid_host: range outside the grid");
    if (count == 0) return 0;
    HIPCHK(hipSetDevice(device));
    int64_t chunk = std::min<int64_t>(count, (int64_t)1 << 27);   // 128 Mi points = 512 MiB of device memory
    const int64_t grow = n2 > 1 ? n2 : n1;                         // whole grid rows per chunk (row-block kernel)
    if (start % grow == 0 && chunk > grow) chunk = chunk / grow * grow;
    float* d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, (size_t)chunk * sizeof(float)));
    AxisTables t;
    SrcGrid g;
    int rc = upload_axes(ax0, n0, ax1, n1, ax2, n2, nullptr, &t, &g, start);
    for (int64_t s = 0; s < count && rc == 0; s += chunk) {
        const int64_t m = std::min(chunk, count - s);
        g.start = start + s;
        rc = run(p, nullptr, &g, m, d_out, nullptr, mode, true);
        if (rc == 0 && hipMemcpy(out + s, d_out, (size_t)m * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(-6, "sdfk_eval_grid_host: device-to-host copy failed");
    }
    (void)hipFree(d_out);
    return rc;
}

// One process, several devices: device d evaluates the d-th slab of whole grid rows and copies it into its part of
// the host field; the slabs run concurrently (one host thread per device). The per-device state of a program and
// the kernel cache are keyed by device, so this is sdfk_eval_grid_host once per slab.
extern "C" int sdfk_eval_grid_sharded(sdfk_program* p, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                                      const float* ax2, int64_t n2, int n_shards, const int* devices, float* out,
                                      int mode) {
    if (!p || !out) return fail(-1, "sdfk_eval_grid_sharded: null argument");
    if (n_shards < 1 || n_shards > 64) return fail(-1, "sdfk_eval_grid_sharded: 1..64 shards");
    if (!ax0 || !ax1 || !ax2 || n0 < 1 || n1 < 1 || n2 < 1) return fail(-1, "grid axes missing or empty");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(-8, "sdfk_eval_grid_sharded: no HIP device");
    const int64_t total = n0 * n1 * n2, unit = n2 > 1 ? n2 : n1;
    const int64_t per = (total / unit / n_shards) * unit;
    std::vector<int> rc((size_t)n_shards, 0);
    std::vector<std::string> msg((size_t)n_shards);
    std::vector<std::thread> workers;
    for (int d = 0; d < n_shards; ++d) {
        const int dev = devices ? devices[d] : d % n_dev;
        if (dev < 0 || dev >= n_dev) return fail(-1, "sdfk_eval_grid_sharded: device index out of range");
    }
    for (int d = 0; d < n_shards; ++d) {
        const int dev = devices ? devices[d] : d % n_dev;
        const int64_t start = d * per, count = d < n_shards - 1 ? per : total - start;
        workers.emplace_back([=, &rc, &msg] {
            rc[(size_t)d] = count > 0 ? sdfk_eval_grid_host(p, ax0, n0, ax1, n1, ax2, n2, start, count, out + start, dev, mode) : 0;
            if (rc[(size_t)d]) msg[(size_t)d] = sdfk_last_error();     // thread-local: carry it to the caller's thread
        });
    }
    for (std::thread& t : workers) t.join();
    for (int d = 0; d < n_shards; ++d)
        if (rc[(size_t)d]) return fail(rc[(size_t)d], "shard " + std::to_string(d) + ": " + msg[(size_t)d]);
    return 0;
}

// The same partition with the field left ON THE DEVICES: shard d is evaluated on devices[d] and lands in its place of
// `d_full`, a buffer of n0 * n1 * n2 floats on `gather_device` — written in place by the shards that run on that device,
// moved by hipMemcpyPeerAsync (device to device over xGMI, no host buffer) by the others. A C consumer without torch gets
// the reassembled field on one GPU this way; the multi-process route with RCCL is aegolius_amd/distributed.py.
extern "C" int sdfk_eval_grid_sharded_device(sdfk_program* p, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                                             const float* ax2, int64_t n2, int n_shards, const int* devices,
                                             int gather_device, float* d_full, int mode) {
    if (!p || !d_full) return fail(-1, "sdfk_eval_grid_sharded_device: null argument");
    if (n_shards < 1 || n_shards > 64) return fail(-1, "sdfk_eval_grid_sharded_device: 1..64 shards");
    if (!ax0 || !ax1 || !ax2 || n0 < 1 || n1 < 1 || n2 < 1) return fail(-1, "grid axes missing or empty");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(-8, "sdfk_eval_grid_sharded_device: no HIP device");
    if (gather_device < 0 || gather_device >= n_dev) return fail(-1, "sdfk_eval_grid_sharded_device: gather device out of range");
    for (int d = 0; d < n_shards; ++d) {
        const int dev = devices ? devices[d] : d % n_dev;
        if (dev < 0 || dev >= n_dev) return fail(-1, "sdfk_eval_grid_sharded_device: device index out of range");
    }
    const int64_t total = n0 * n1 * n2, unit = n2 > 1 ? n2 : n1;
    const int64_t per = (total / unit / n_shards) * unit;
    std::vector<int> rc((size_t)n_shards, 0);
    std::vector<std::string> msg((size_t)n_shards);
    std::vector<std::thread> workers;
    for (int d = 0; d < n_shards; ++d) {
        const int dev = devices ? devices[d] : d % n_dev;
        const int64_t start = d * per, count = d < n_shards - 1 ? per : total - start;
        workers.emplace_back([=, &rc, &msg] {
            auto shard = [&]() -> int {
                if (count <= 0) return 0;
                HIPCHK(hipSetDevice(dev));
                hipStream_t stream = nullptr;
                HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
                // (SDFK_FORCE_PEER_COPY=1: the copy path also for the shards of the gather device — how a one-GPU box tests it)
                static const bool force_copy = [] { const char* e = getenv("SDFK_FORCE_PEER_COPY"); return e && e[0] == '1'; }();
                const bool in_place = dev == gather_device && !force_copy;
                float* d_slab = in_place ? d_full + start : nullptr;
                int r = 0;
                if (!d_slab && hipMalloc(&d_slab, (size_t)count * sizeof(float)) != hipSuccess) r = fail(-5, "out of device memory for the slab");
                if (r == 0) r = sdfk_eval_grid(p, ax0, n0, ax1, n1, ax2, n2, start, count, d_slab, stream, mode);   // (synchronises)
                if (r == 0 && !in_place) {
                    if (hipMemcpyPeerAsync(d_full + start, gather_device, d_slab, dev, (size_t)count * sizeof(float), stream) != hipSuccess ||
                        hipStreamSynchronize(stream) != hipSuccess)
                        r = fail(-6, "peer copy of the slab failed");
                }
                if (!in_place && d_slab) (void)hipFree(d_slab);
                (void)hipStreamDestroy(stream);
                return r;
            };
            rc[(size_t)d] = shard();
            if (rc[(size_t)d]) msg[(size_t)d] = sdfk_last_error();
        });
    }
    for (std::thread& t : workers) t.join();
    for (int d = 0; d < n_shards; ++d)
        if (rc[(size_t)d]) return fail(rc[(size_t)d], "shard " + std::to_string(d) + ": " + msg[(size_t)d]);
    return 0;
}

extern "C" int sdfk_grid_fill(float* d_co, int64_t row_stride, const float* ax0, int64_t n0, const float* ax1,
                              int64_t n1, const float* ax2, int64_t n2, int64_t start, int64_t count, void* stream) {
    if (!d_co) return fail(-1, "sdfk_grid_fill: null output");
    if (row_stride < count) return fail(-1, "sdfk_grid_fill: row stride smaller than count");
    if (start < 0 || count < 0 || start + count > n0 * n1 * n2) return fail(-1, "sdfk_grid_fill: range outside the grid");
    if (count == 0) return 0;
    AxisTables t;
    SrcGrid g;
    int rc = upload_axes(ax0, n0, ax1, n1, ax2, n2, (hipStream_t)stream, &t, &g, start);
    if (rc) return rc;
    hipLaunchKernelGGL(sdfk_gridfill_kernel, dim3(blocks_for(count, 4)), dim3(SDFK_BLOCK), 0, (hipStream_t)stream, g,
                       (long long)count, d_co, (long long)row_stride);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ---- host-buffer convenience --------------------------------------------------------------------
// Row length of a host (3, n) array that looks like a flattened meshgrid: the index at which x or y first
// changes (3-D grids: rows along z) or, failing that, at which x first changes (2-D grids: rows along y).
// Only a layout hint for the row-block kernel — a wrong guess costs speed, never correctness.
template <typename T>
static int64_t detect_row_len(const T* co, int64_t n, int64_t stride, bool* flat, int64_t* plane_rows) {
    const int64_t scan = std::min<int64_t>(n, (int64_t)1 << 22);
    const T *x = co, *y = co + stride, *z = co + 2 * stride;
    int64_t a = 0, b = 0;
    *flat = false;
    *plane_rows = 0;
    for (int64_t i = 1; i < scan && (!a || !b); ++i) {
        if (!b && x[i] != x[0]) b = i;
        if (!a && (x[i] != x[0] || y[i] != y[0])) a = i;
    }
    if (a >= 32 && n % a == 0) {
        if (b > a && b % a == 0) *plane_rows = b / a;            // x first changes after b / a rows: one grid plane
        return a;
    }
    if (b >= 32 && n % b == 0) {
        *flat = z[0] == (T)0 && z[b - 1] == (T)0 && z[n - 1] == (T)0;   // (a hint: the kernel checks every point it reads)
        return b;
    }
    return 0;
}

// ---- host arrays in, host array out: a two-slot pipeline over pinned staging buffers ------------------------------
// A pageable (3, n) array cannot be DMA'd directly: hipMemcpy stages it through an internal bounce buffer, one
// synchronous chunk at a time (measured 24-26 GB/s over a 63 GB/s link, kernel and copies never overlapping). Here the
// array is cut into chunks; a few host threads copy (or narrow float64 -> float32) chunk i + 1 into pinned slot B and
// copy chunk i - 1's field out of it, while slot A's stream runs H2D -> kernel -> D2H of chunk i. The pinned slots
// and their device buffers are allocated once per device and kept (256 MiB pinned for 8 Mi-point chunks).
struct HostSlot {
    float *h_co = nullptr, *h_out = nullptr, *d_co = nullptr, *d_out = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    int64_t pending_start = -1, pending_count = 0;        // field in h_out still to be handed to the caller
};
struct HostStage {
    std::mutex mu;                                          // one host-path call per device at a time
    int64_t chunk = 0;
    HostSlot slot[2];
};
static std::mutex g_stage_mu;
static std::map<int, std::unique_ptr<HostStage>> g_stage;
static int host_threads() {
    static int v = [] {
        const char* e = getenv("SDFK_HOST_THREADS");
        int t = e ? atoi(e) : 0;
        if (t < 1) t = std::min(8, std::max(1, (int)std::thread::hardware_concurrency() / 4));
        return std::min(t, 64);
    }();
    return v;
}
// run fn(lo, hi) over [0, count) on the calling thread plus helpers
template <typename F>
static void parallel_ranges(int64_t count, F fn) {
    const int t = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, count >> 18));
    if (t <= 1) {
        fn(0, count);
        return;
    }
    std::vector<std::thread> helpers;
    const int64_t step = (count + t - 1) / t;
    for (int i = 1; i < t; ++i) helpers.emplace_back([=] { fn(std::min(count, i * step), std::min(count, (i + 1) * step)); });
    fn(0, std::min(count, step));
    for (std::thread& h : helpers) h.join();
}
static HostStage* stage_of(int device) {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    std::unique_ptr<HostStage>& st = g_stage[device];
    if (!st) st.reset(new HostStage);
    return st.get();
}
// the caller holds st->mu: no other call of this device has anything in flight in the slots being replaced
static int stage_grow(HostStage* st, int64_t want) {
    if (st->chunk >= want) return 0;
    for (HostSlot& sl : st->slot) {
        if (sl.stream) HIPCHK(hipStreamSynchronize(sl.stream));
        if (sl.h_co) (void)hipHostFree(sl.h_co);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_co) (void)hipFree(sl.d_co);
        if (sl.d_out) (void)hipFree(sl.d_out);
        sl.h_co = sl.h_out = sl.d_co = sl.d_out = nullptr;
        if (!sl.stream) HIPCHK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        if (!sl.done) HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        st->chunk = 0;
        HIPCHK(hipHostMalloc((void**)&sl.h_co, (size_t)want * 3 * sizeof(float), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void**)&sl.h_out, (size_t)want * sizeof(float), hipHostMallocDefault));
        HIPCHK(hipMalloc(&sl.d_co, (size_t)want * 3 * sizeof(float)));
        HIPCHK(hipMalloc(&sl.d_out, (size_t)want * sizeof(float)));
    }
    st->chunk = want;
    return 0;
}

// out_on_device: `out` is device memory of the same device (the field stays resident, nothing comes back)
static int eval_host_impl(sdfk_program* p, const void* co, int co_dtype, int64_t n, int64_t row_stride, float* out,
                          int device, int mode, bool out_on_device) {
    if (!p) return fail(-1, "null program");
    if (n < 0 || (n > 0 && (!co || !out))) return fail(-1, "sdfk_eval_host: bad arguments");
    if (co_dtype != 0 && co_dtype != 1) return fail(-1, "sdfk_eval_host: co_dtype must be 0 (fp32) or 1 (fp64)");
    if (row_stride < n) return fail(-1, "sdfk_eval_host: row stride smaller than the point count");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(device));
    static const int64_t max_chunk = [] {
        const char* e = getenv("SDFK_HOST_CHUNK");
        const int64_t v = e ? atoll(e) : 0;
        return v >= 4096 ? v : ((int64_t)1 << 23);           // 8 Mi points: 96 MiB in + 32 MiB out per slot
    }();
    int64_t chunk = std::min<int64_t>(n, max_chunk);
    bool flat = false;
    int64_t plane_rows = 0;
    // SDFK_HOST_TRACE=1: where the call's time goes (stderr, one line per call)
    static const bool trace = [] { const char* e = getenv("SDFK_HOST_TRACE"); return e && e[0] == '1'; }();
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const clk::time_point t_call = clk::now();
    double t_stage_in = 0.0, t_wait = 0.0, t_copy_out = 0.0, t_enqueue = 0.0;
    const int64_t row_len = p->sites.empty() ? 0
                            : co_dtype == 0 ? detect_row_len(static_cast<const float*>(co), n, row_stride, &flat, &plane_rows)
                                            : detect_row_len(static_cast<const double*>(co), n, row_stride, &flat, &plane_rows);
    if (row_len > 0 && chunk > row_len) chunk = chunk / row_len * row_len;   // whole rows per chunk
    const int64_t stride = (chunk + 63) & ~(int64_t)63;
    HostStage* st = stage_of(device);
    std::lock_guard<std::mutex> lk(st->mu);                  // one host-path call per device at a time, growth included
    int rc = stage_grow(st, stride);
    if (rc) return rc;
    // the first call of a program builds its kernel: do that before anything is in flight (the build may take seconds)
    auto hand_over = [&](HostSlot& sl) -> int {             // wait for the slot's chunk and give its field to the caller
        if (sl.pending_start < 0) return 0;
        clk::time_point t0 = clk::now();
        HIPCHK(hipEventSynchronize(sl.done));
        t_wait += ms_since(t0);
        if (!out_on_device) {
            t0 = clk::now();
            const float* src = sl.h_out;
            float* dst = out + sl.pending_start;
            parallel_ranges(sl.pending_count, [=](int64_t lo, int64_t hi) { memcpy(dst + lo, src + lo, (size_t)(hi - lo) * sizeof(float)); });
            t_copy_out += ms_since(t0);
        }
        sl.pending_start = -1;
        return 0;
    };
    int k = 0;
    for (int64_t s = 0; s < n && rc == 0; s += chunk, ++k) {
        HostSlot& sl = st->slot[k & 1];
        rc = hand_over(sl);                                  // the slot's previous chunk (two chunks ago) is done: reuse it
        if (rc) break;
        const int64_t m = std::min(chunk, n - s);
        float* h = sl.h_co;
        const clk::time_point t_in = clk::now();
        if (co_dtype == 0) {
            const float* base = static_cast<const float*>(co) + s;
            parallel_ranges(m, [=](int64_t lo, int64_t hi) {
                for (int r = 0; r < 3; ++r) memcpy(h + r * stride + lo, base + r * row_stride + lo, (size_t)(hi - lo) * sizeof(float));
            });
        } else {
            const double* base = static_cast<const double*>(co) + s;
            parallel_ranges(m, [=](int64_t lo, int64_t hi) {
                for (int r = 0; r < 3; ++r) {
                    const double* src = base + r * row_stride;
                    float* dst = h + r * stride;
                    for (int64_t i = lo; i < hi; ++i) dst[i] = (float)src[i];
                }
            });
        }
        t_stage_in += ms_since(t_in);
        const clk::time_point t_enq = clk::now();
        for (int r = 0; r < 3 && rc == 0; ++r)
            if (hipMemcpyAsync(sl.d_co + r * stride, h + r * stride, (size_t)m * sizeof(float), hipMemcpyHostToDevice, sl.stream) != hipSuccess)
                rc = fail(-6, "sdfk_eval_host: host-to-device copy failed");
        if (rc == 0)
            rc = !(row_len > 0 && m % row_len == 0) ? sdfk_eval_device(p, sl.d_co, m, stride, sl.d_out, sl.stream, mode)
                 : flat                             ? sdfk_eval_device_rows2d(p, sl.d_co, m, stride, row_len, sl.d_out, sl.stream, mode)
                 : sdfk_eval_device_rows3d(p, sl.d_co, m, stride, row_len, plane_rows,
                                           plane_rows > 0 ? (s / row_len) % plane_rows : 0, sl.d_out, sl.stream, mode);
        if (rc == 0) {
            const hipError_t e = out_on_device
                                     ? hipMemcpyAsync(out + s, sl.d_out, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, sl.stream)
                                     : hipMemcpyAsync(sl.h_out, sl.d_out, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, sl.stream);
            if (e != hipSuccess) rc = fail(-6, "sdfk_eval_host: copy of the result failed");
        }
        if (rc == 0 && hipEventRecord(sl.done, sl.stream) != hipSuccess) rc = fail(-6, "sdfk_eval_host: event record failed");
        if (rc == 0) {
            sl.pending_start = s;
            sl.pending_count = m;
        }
        t_enqueue += ms_since(t_enq);
    }
    for (HostSlot& sl : st->slot) {                         // drain (also on errors: nothing of this call stays in flight)
        if (rc == 0) rc = hand_over(sl);
        else {
            (void)hipStreamSynchronize(sl.stream);
            sl.pending_start = -1;
        }
    }
    if (trace)
        fprintf(stderr, "[sdfk host] %lld points, %d chunks of %lld: total %.2f ms = stage-in %.2f + enqueue %.2f + wait %.2f + copy-out %.2f (+ %.2f other)\n",
                (long long)n, k, (long long)chunk, ms_since(t_call), t_stage_in, t_enqueue, t_wait, t_copy_out,
                ms_since(t_call) - t_stage_in - t_enqueue - t_wait - t_copy_out);
    return rc;
}

extern "C" int sdfk_eval_host(sdfk_program* p, const void* co, int co_dtype, int64_t n, int64_t row_stride, float* out,
                              int device, int mode) {
    return eval_host_impl(p, co, co_dtype, n, row_stride, out, device, mode, false);
}
extern "C" int sdfk_eval_host_resident(sdfk_program* p, const void* co, int co_dtype, int64_t n, int64_t row_stride,
                                       float* d_out, int device, int mode) {
    return eval_host_impl(p, co, co_dtype, n, row_stride, d_out, device, mode, true);
}

// ---- plumbing -----------------------------------------------------------------------------------
extern "C" int sdfk_set_device(int device) {
    HIPCHK(hipSetDevice(device));
    return 0;
}
extern "C" void* sdfk_malloc(size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        fail(-5, std::string("hipMalloc: ") + hipGetErrorString(e));
        return nullptr;
    }
    return p;
}
extern "C" int sdfk_free(void* d_ptr) {
    HIPCHK(hipFree(d_ptr));
    return 0;
}
extern "C" int sdfk_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
    HIPCHK(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" int sdfk_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
    HIPCHK(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sdfk_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
    HIPCHK(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
    return 0;
}
extern "C" int sdfk_sync(void* stream) {
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
extern "C" void* sdfk_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        fail(-7, "hipEventCreate failed");
        return nullptr;
    }
    return e;
}
extern "C" int sdfk_event_destroy(void* ev) {
    HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
extern "C" int sdfk_event_record(void* ev, void* stream) {
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int sdfk_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    HIPCHK(hipEventSynchronize((hipEvent_t)ev_stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}
extern "C" int sdfk_stream_probe(const float* d_co, int64_t n, int64_t row_stride, float* d_out, void* stream) {
    if (!d_co || !d_out) return fail(-1, "sdfk_stream_probe: null pointer");
    if (!(aligned16(d_co) && aligned16(d_out) && row_stride % 4 == 0 && n % 4 == 0))
        return fail(-1, "sdfk_stream_probe: needs 16-byte aligned rows and n % 4 == 0");
    SrcArray a = {d_co, (long long)row_stride};
    hipLaunchKernelGGL(sdfk_probe_kernel, dim3(blocks_for(n, 4)), dim3(SDFK_BLOCK), 0, (hipStream_t)stream, a,
                       (long long)n, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

#include "sdfk_gridops.inc"
#include "sdfk_fieldops.inc"
#include "sdfk_vector.inc"
#include "sdfk_hosttree.inc"
