// sdfk_codegen.h — turns a validated register-machine program into straight-line HIP source.
#ifndef SDFK_CODEGEN_H
#define SDFK_CODEGEN_H
#include <cstddef>
#include <cstdint>
#include <string>

enum sdfk_kind { SDFK_KIND_C_C = 0, SDFK_KIND_V_C = 1, SDFK_KIND_V_V = 2, SDFK_KIND_V_VV = 3 };

struct sdfk_opinfo {
    const char* name;
    int kind;
    int nparams;
    const char* func;
};

// Specialised per TOPOLOGY: opcodes, register operands and parameter offsets are baked into the
// text; parameter VALUES stay in the runtime table (PRM), so one compiled kernel serves every
// tree of the same shape.
std::string sdfk_generate_source(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr,
                                 int result_reg);

#endif
