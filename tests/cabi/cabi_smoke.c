/* Plain-C consumer of include/sdfk.h (C99, no C++): proves the boundary is a C ABI. Built and run by
 * tests/test_native_cpu.py::test_header_is_valid_c_and_links (no GPU needed: only calls that work without a device).
 * The program is the lowered form of Sphere(0.5): P_SPHERE V0 <- C0, parameter table {0.5}. */
#include <stdio.h>
#include <string.h>

#include "sdfk.h"

int main(int argc, char** argv) {
    unsigned op_sphere;
    if (argc != 2 || sscanf(argv[1], "%u", &op_sphere) != 1) {
        fprintf(stderr, "usage: cabi_smoke <opcode of P_SPHERE>\n");
        return 2;
    }
    if (sdfk_abi_version() != SDFK_ABI_VERSION) return 3;
    if (sdfk_device_count() < 0) return 4;

    const uint32_t code[2] = {op_sphere | (0u << 8) | (0u << 16), 0u};   /* V[0] = sphere(C[0]), params at 0 */
    const float params[1] = {0.5f};
    sdfk_program* prog = sdfk_program_create(code, 1, params, 1, NULL, 0, 0);
    if (!prog) {
        fprintf(stderr, "create failed: %s\n", sdfk_last_error());
        return 5;
    }
    const char* src = sdfk_program_source(prog);
    if (!src || !strstr(src, "sdfk_spec_v4")) return 6;
    size_t code_size = 0;
    if (sdfk_program_compile_check(prog, &code_size) != 0 || code_size < 1000) {
        fprintf(stderr, "hiprtc: %s\n", sdfk_last_error());
        return 7;
    }
    /* a malformed program is refused with a message, never launched */
    const uint32_t bad[2] = {255u, 0u};
    if (sdfk_program_create(bad, 1, params, 1, NULL, 0, 0) != NULL || strlen(sdfk_last_error()) == 0) return 8;
    float ax[5];
    if (sdfk_linspace_f32(-1.0, 1.0, 5, ax) != 0 || ax[0] != -1.0f || ax[2] != 0.0f || ax[4] != 1.0f) return 9;
    sdfk_program_destroy(prog);
    printf("cabi ok: code object %zu bytes\n", code_size);
    return 0;
}
