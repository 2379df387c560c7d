// sdfk_codegen.h — turns a validated register-machine program into straight-line HIP source.
#ifndef SDFK_CODEGEN_H
#define SDFK_CODEGEN_H
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

enum sdfk_kind { SDFK_KIND_C_C = 0, SDFK_KIND_V_C = 1, SDFK_KIND_V_V = 2, SDFK_KIND_V_VV = 3 };

struct sdfk_opinfo {
    const char* name;
    int kind;
    int nparams;
    const char* func;
};

// One brick-culling site: the V_VV combiner at instruction `comb` whose first operand is produced by
// instructions [a0, a1] and whose second operand by [b0, b1] (b1 == comb - 1). k = L_a + L_b
// (Lipschitz constants of the two operand fields w.r.t. the root point). skip_*_ok: skipping that
// range has no side effect on registers read later (checked by sdfk_program_set_cull).
struct sdfk_cullsite {
    uint32_t comb, a0, a1, b0, b1;
    float k;
    int skip_a_ok, skip_b_ok;
};

// Specialised per TOPOLOGY: opcodes, register operands and parameter offsets are baked into the
// text; parameter VALUES stay in the runtime table (PRM), so one compiled kernel serves every
// tree of the same shape.
// One hiprtc translation unit per kernel FLAVOUR: a call needs one of them, and hiprtc's time is spent per
// __global__ function (every flavour inlines the whole tree), so only what is launched is ever compiled.
enum sdfk_flavour {
    SDFK_FL_PLAIN_ARRAY = 0,   // sdfk_spec_v4 + sdfk_spec_v1
    SDFK_FL_PLAIN_GRID,        // sdfk_spec_g4 + sdfk_spec_g1
    SDFK_FL_TILE_ARRAY,        // sdfk_spec_t
    SDFK_FL_TILE_GRID,         // sdfk_spec_tg
    SDFK_FL_TILE_MASK,         // sdfk_spec_tmask (test aid)
    SDFK_FL_ROWS_ARRAY,        // sdfk_spec_r
    SDFK_FL_ROWS_GRID,         // sdfk_spec_rg
    SDFK_FL_ROWS_MASK,         // sdfk_spec_rmask (test aid)
    SDFK_FL_ROWS2D_ARRAY,      // sdfk_spec_r  built for flat grids (rows along y, z = 0: SDFK_FLAT)
    SDFK_FL_ROWS2D_GRID,       // sdfk_spec_rg built for flat grids
    SDFK_FL_COUNT,
    SDFK_FL_ALL = SDFK_FL_COUNT   // everything in one unit (sdfk_program_source, developer tools)
};
// sites: what the mask kernels use (at most 64); sites_all (optional): every site of the program — a long n-ary
// min / max chain whose children read nothing but the input point is generated TABLE-DRIVEN from them ("chain mode":
// one function per kind of child, loops over tables of parameter offsets; compile time and code size no longer grow
// with the number of children, culling keeps a LIST of surviving children per brick instead of mask bits).
std::string sdfk_generate_source(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr,
                                 int result_reg, const std::vector<sdfk_cullsite>& sites, int flavour,
                                 const std::vector<sdfk_cullsite>* sites_all = nullptr);
// members of the chain when the program runs in chain mode, else 0 (the launcher then has no line-brick flavour and never
// falls back to the interpreter for size)
int sdfk_chain_mode(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr, int result_reg,
                    const std::vector<sdfk_cullsite>& sites_all);

// Text every chain-specialised vector kernel starts with: sdfk_device.h followed by sdfk_vecdev.h.
std::string sdfk_vector_prelude();

#endif
