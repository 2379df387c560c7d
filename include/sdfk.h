/* sdfk.h — C-ABI of libsdfk.so, the MI355X (gfx950) SDF grid-evaluation engine.
 *
 * The reference (peterropac/Aegolius = SPOMSO, pure Python) has no FFI seam; its seam is the
 * Python object protocol `GenericGeometry.create(co) -> (N,)`
 * (reference Code/spomso/spomso/cores/geom.py:29-43). The Python layer in aegolius_amd/ keeps that
 * protocol and lowers the expression tree it records to the register-machine program consumed here.
 * Each entry point below names the reference interface whose work it replaces.
 *
 * Conventions: plain C types only; 0 = success, negative = error (text via sdfk_last_error(),
 * thread-local); the caller owns every host buffer for the duration of a call; the library owns
 * programs and the device memory it allocates; programs are immutable after creation and may be
 * shared between threads; "device pointers" are ordinary HIP device addresses (for example
 * torch.Tensor.data_ptr() of a ROCm tensor) and `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).
 */
#ifndef SDFK_H
#define SDFK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDFK_ABI_VERSION 1

/* evaluation modes for sdfk_eval_device / sdfk_set_default_mode */
#define SDFK_MODE_AUTO 0        /* topology-specialised kernel (hiprtc): only the kernel flavour a call launches is built,
                                   in a background thread, and cached (per process and on disk); until it is ready calls
                                   are served by the interpreter kernel (bit-identical results; SDFK_ASYNC_JIT=0: wait
                                   instead). The background build runs in a CHILD PROCESS (aegolius_amd/sdfk_rtc_helper:
                                   hiprtc inside the caller deadlocks against a dlopen of any HIP library on another
                                   thread); without the helper the first call waits. Build time is bounded by program
                                   size (programs that run in chain mode — sdfk_program_set_cull — build in under a
                                   second whatever their size): row-block kernels up to SDFK_ROWS_LIMIT (env, default
                                   1200) instructions, line bricks / plain up to SDFK_SPECIALIZE_LIMIT (1200), beyond that
                                   the interpreter kernel serves the program; from SDFK_BIG_PROGRAM (300) instructions on
                                   a build runs without LLVM's CodeGenPrepare and VectorCombine passes (quadratic in a
                                   straight-line program; same FP semantics, same bits) */
#define SDFK_MODE_INTERPRET 1   /* generic register-machine interpreter kernel */
#define SDFK_MODE_SPECIALIZED 2 /* wait for the specialised kernel; fail instead of falling back if hiprtc fails */
#define SDFK_MODE_NOCULL 3      /* specialised kernel with brick culling switched off (A/B runs, tests) */

typedef struct sdfk_program sdfk_program;

int sdfk_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU; never an error) */
int sdfk_device_count(void);
const char* sdfk_last_error(void);

/* ---- programs --------------------------------------------------------------------------------
 * A program is the lowered form of one expression tree: what the reference holds as nested Python
 * closures (cores/modifications.py:55-63 `modified_object`, cores/combine.py:129-138) plus the
 * Euclidean transform of every node (cores/transformations.py:232-242).
 *   code   : 2 words per instruction: {op | a<<8 | b<<16 | c<<24, parameter offset}
 *            (opcodes: aegolius_amd/csrc/sdfk_ops.def)
 *   params : fp32 parameter table (host-precomputed constants)
 *   tables : fp32 variable-length tables (poly-lines, point sets, convex pieces); may be NULL/0
 *   result_reg : value register that holds the field after the last instruction
 * The program is validated (opcodes, register indices, parameter ranges) — a malformed program is
 * rejected here, never launched. */
sdfk_program* sdfk_program_create(const uint32_t* code, size_t n_instr, const float* params, size_t n_params,
                                  const float* tables, size_t n_tables, int result_reg);
void sdfk_program_destroy(sdfk_program* prog);
/* Replace the parameter values of a program in place (same topology, new shape parameters). */
int sdfk_program_set_params(sdfk_program* prog, const float* params, size_t n_params);
/* Brick culling (optional, before first use of the program): n_sites rows (at most 32767) {combiner index, a_start,
 * a_end, b_start, b_end} naming, for min/max-type combiners, the instruction ranges that produce the
 * two operands, and k[i] = L_a + L_b, the sum of the Lipschitz constants of the operand fields with
 * respect to the input point. The specialised kernels then probe the tree per BRICK — 32 points x 16 grid rows when the
 * caller gives the row length (sdfk_eval_device_rows, grids), 128 consecutive points otherwise — and skip operand
 * subtrees that provably cannot change the result on that brick; results are bit-identical to the un-culled
 * evaluation. Sites whose ranges have side effects on registers read later are kept but never skipped.
 * How the sites are used: up to 64 of them as two mask bits each (the widest, in program order; the line-brick kernel
 * takes 31). The probe runs lane-parallel when every leaf range (a range without a site inside) reads nothing but the
 * input point: all leaves at all probe centres on the lanes of the workgroup, 8 / 4 / 1 centres per brick. A program
 * that holds an n-ary hard min / max over 17 to 32768 such leaves (CombineGeometry("UNION").combine(*many)) runs in
 * "chain mode" with all of the chain's sites: one function per kind of leaf, tables of parameter offsets, a list of
 * surviving leaves per brick; it builds in about a second whatever its size. The chain may be the whole program (with
 * value modifications of its result) or an operand of a small program around it — clipped, blended, subtracted: those
 * instructions (at most 64 + a quarter of the chain's, and 256) run per point around the chain's value, un-culled. */
int sdfk_program_set_cull(sdfk_program* prog, const uint32_t* sites, size_t n_sites, const float* k);
/* Members of the n-ary chain when the program runs in chain mode (after sdfk_program_set_cull), else 0. */
int sdfk_program_chain_members(const sdfk_program* prog);
/* Generated HIP source of the specialised kernel (for inspection / tests); NULL on error. */
const char* sdfk_program_source(sdfk_program* prog);
/* Compile the specialised kernel for gfx950 with hiprtc without needing a GPU (build check).
 * Returns 0 and the code-object size in *code_size. */
int sdfk_program_compile_check(sdfk_program* prog, size_t* code_size);
/* Kernel flavours: one hiprtc translation unit each, built only when a call needs it. */
#define SDFK_FLAVOUR_PLAIN_ARRAY 0 /* sdfk_spec_v4 / v1: straight-line body on a (3, n) array */
#define SDFK_FLAVOUR_PLAIN_GRID 1  /* the same from per-axis grid tables */
#define SDFK_FLAVOUR_TILE_ARRAY 2  /* exact culling on bricks of 128 consecutive points */
#define SDFK_FLAVOUR_TILE_GRID 3
#define SDFK_FLAVOUR_TILE_MASK 4   /* test aid (sdfk_debug_brick_masks) */
#define SDFK_FLAVOUR_ROWS_ARRAY 5  /* exact culling on blocks of 32 points x 16 grid rows (sdfk_eval_device_rows) */
#define SDFK_FLAVOUR_ROWS_GRID 6
#define SDFK_FLAVOUR_ROWS_MASK 7   /* test aid (sdfk_debug_row_masks) */
#define SDFK_FLAVOUR_ROWS2D_ARRAY 8 /* the row-block kernel built for flat grids (sdfk_eval_device_rows2d) */
#define SDFK_FLAVOUR_ROWS2D_GRID 9
/* OR-ed onto a PLAIN / ROWS / ROWS2D flavour: its flag-writing build (one bit per point, value <= threshold, instead of
   the field — what sdfk_eval_device_select / sdfk_eval_grid_select launch). A translation unit of its own: the field
   kernels carry none of it (as a run-time branch it cost the 20-primitive tree 10 % at 1025^3). */
#define SDFK_FLAVOUR_FLAGS 0x100
/* OR-ed onto PLAIN_ARRAY / ROWS2D_ARRAY: the build for two-row coordinates (z = 0 by contract: sdfk_eval_device_rows2d_xy) */
#define SDFK_FLAVOUR_XY 0x200
/* Build (or fetch from the caches) ONE flavour, GPU or not: its code-object size and the seconds this call took. */
int sdfk_program_compile_flavour(sdfk_program* prog, int flavour, size_t* code_size, double* seconds);
/* Test aid: with enable != 0 every chain-mode row-block launch synchronises after the pre-pass of its candidate lists and
 * records { fine cells, coarse cells, pool entries used, pool capacity, sum of the fine lists' lengths, longest fine list,
 * fine cells without a list (their bricks probe every member), empty cells }; out8 (nullable) receives the record of the
 * last such launch and clears it. */
void sdfk_debug_cells_stats(int enable, long long* out8);
/* Test aid: build one flavour the way BACKGROUND builds are run — in a child process (aegolius_amd/sdfk_rtc_helper,
 * csrc/sdfk_rtc_helper.c) — and return the code-object size; nothing is cached. While the interpreter kernel serves the
 * first calls of a new tree shape (SDFK_MODE_AUTO) the compiler never runs inside the calling process: hiprtc holds a
 * process-wide lock of its compiler library for the whole build, against which a dlopen of any HIP library on another
 * thread deadlocks. Without the helper next to the library there are no background builds: the first call waits. */
int sdfk_debug_compile_external(sdfk_program* prog, int flavour, size_t* code_size);
/* Wait until no background kernel build is queued or running. */
void sdfk_jit_drain(void);
/* The same for a process that is leaving: queued builds are dropped, running compiler child processes killed (a build of a
 * big tree takes up to a minute; nobody would use its kernel). What the Python layer registers with atexit. */
void sdfk_jit_cancel(void);
/* hiprtc builds this process has actually run (cache hits excluded) and the seconds they took. */
void sdfk_debug_jit_stats(int64_t* builds, double* seconds);
/* Extra -D switches handed to hiprtc for kernels built from now on (experiments; also env SDFK_RTC_DEFS). */
void sdfk_debug_set_rtc_defs(const char* defs);

/* ---- evaluation ------------------------------------------------------------------------------
 * Replaces GenericGeometry.create / propagate (cores/geom.py:29-60) for one whole tree:
 * out[i] = tree(co[0][i], co[1][i], co[2][i]).
 * d_co  : device pointer to a (3, n) fp32 array; row r starts at d_co + r*row_stride (elements).
 * d_out : device pointer to n fp32.
 * Uses 16-byte vector loads when d_co, d_out and row_stride allow (16-B aligned, stride % 4 == 0). */
int sdfk_eval_device(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, float* d_out,
                     void* stream, int mode);
/* The same with a LAYOUT HINT: the n points are consecutive rows of row_len points (n % row_len == 0) — for the
 * (3, N) array of generate_grid (cores/helper_functions.py:86-91, meshgrid "ij" flattened) row_len is the last
 * grid dimension. Brick culling then works on blocks of 32 points x 16 rows instead of 128 points in a line
 * (4x smaller bounding spheres, far fewer surviving subtrees), and rows need no 16-byte alignment. The hint
 * affects speed only: the field is bit-identical to sdfk_eval_device for ANY row_len (bounds, the "one x and
 * one y per row" test and every skip decision are derived from the coordinates actually read). */
int sdfk_eval_device_rows(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                          float* d_out, void* stream, int mode);
/* The same with the second layout hint of a 3-D grid: the rows come in PLANES of plane_rows rows (the second grid
 * dimension; rows of one plane share x) and the first row of the array is row first_row_in_plane of its plane
 * (0 for a whole grid, anything for an x-slab of whole rows). Row blocks then never straddle two planes — such a
 * block spans the whole y extent of the grid and culls nothing (1 block in 32 at 513^3). plane_rows = 0: unknown
 * (= sdfk_eval_device_rows). Hints only: the field is bit-identical for any values. Honoured when the environment
 * sets SDFK_PLANE_BLOCKS=1: measured on 2^k + 1 grids the partial block that then ends every plane costs what the
 * straddling block saved, so by default the call equals sdfk_eval_device_rows. */
int sdfk_eval_device_rows3d(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                            int64_t plane_rows, int64_t first_row_in_plane, float* d_out, void* stream, int mode);
/* The same for the array of a FLAT grid (generate_grid with two sizes, cores/helper_functions.py:63-75: rows run
 * along y, row_len = the second grid dimension, the z row is all zeros): the kernel is built so that the x part of
 * every root transform is computed once per row. Again a hint only — bricks whose z is not exactly 0 or whose x
 * varies along a row take the general path, and the field is bit-identical to sdfk_eval_device (up to the sign of
 * a zero). */
int sdfk_eval_device_rows2d(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                            float* d_out, void* stream, int mode);
/* Flat grids WITHOUT their z row: d_xy is a (2, n) array — row 0 = x, row 1 = y, row pitch row_stride elements — and
 * z = 0 is the CONTRACT of the call, not something the kernel reads (the reference's 2-D generate_grid,
 * cores/helper_functions.py:63-75, always appends a row of zeros; streaming it costs a quarter of the traffic: 12
 * instead of 16 bytes per point). row_len as for sdfk_eval_device_rows2d (0: no layout hint — the plain kernel). The
 * field is bit-identical to sdfk_eval_device_rows2d on the same x, y with a zero z row. Waits for the specialised kernel
 * (the interpreter kernel has no two-row build); programs that read auxiliary fields are refused. */
int sdfk_eval_device_rows2d_xy(sdfk_program* prog, const float* d_xy, int64_t n, int64_t row_stride, int64_t row_len,
                               float* d_out, void* stream, int mode);
/* Host-buffer convenience: stages co (dtype 0 = fp32, 1 = fp64; (3, n) with row stride in elements)
 * through device memory in chunks, evaluates and copies the fp32 field back. */
int sdfk_eval_host(sdfk_program* prog, const void* co, int co_dtype, int64_t n, int64_t row_stride, float* out,
                   int device, int mode);
/* Same staging of host coordinates, but the field is written to d_out — n floats of DEVICE memory on `device` —
 * and stays there for the field consumers below (nothing is copied back). */
int sdfk_eval_host_resident(sdfk_program* prog, const void* co, int co_dtype, int64_t n, int64_t row_stride,
                            float* d_out, int device, int mode);
/* Evaluate directly on a regular grid without materialising coordinates (4 B/point of traffic):
 * point `start + i` of the flat index n = (ix*n1 + iy)*n2 + iz takes (ax0[ix], ax1[iy], ax2[iz]).
 * Replaces generate_grid + create (cores/helper_functions.py:23-93). Axis tables are HOST pointers. */
int sdfk_eval_grid(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1, const float* ax2,
                   int64_t n2, int64_t start, int64_t count, float* d_out, void* stream, int mode);
/* Same, field copied back to a HOST buffer in device-sized chunks: the whole generate_grid + create round
 * trip without a coordinate array on either side. */
int sdfk_eval_grid_host(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                        const float* ax2, int64_t n2, int64_t start, int64_t count, float* out, int device, int mode);
/* Single-process multi-GPU: the grid is cut into n_shards contiguous slabs of whole rows (the remainder goes to
 * the last); shard d runs on devices[d] (devices == NULL: d modulo the device count), all shards concurrently,
 * each copying its slab into `out` (HOST, n0*n1*n2 floats). Slabs are independent (the path is pointwise): no
 * collective. The multi-process route (one rank per GPU, torch.distributed) is aegolius_amd/distributed.py. */
int sdfk_eval_grid_sharded(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                           const float* ax2, int64_t n2, int n_shards, const int* devices, float* out, int mode);
/* The same partition with the field left on the DEVICES: shard d runs on devices[d] and lands in its place of d_full, a
 * buffer of n0*n1*n2 floats on gather_device — in place for the shards of that device, by hipMemcpyPeerAsync (device to
 * device over xGMI, no host buffer) for the others. */
int sdfk_eval_grid_sharded_device(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                                  const float* ax2, int64_t n2, int n_shards, const int* devices, int gather_device,
                                  float* d_full, int mode);
void sdfk_set_default_mode(int mode);
/* Test / diagnostics aid for brick culling: writes one 64-bit skip mask per brick of 128 consecutive
 * points (ceil(n / 2048) * 16 entries; bit 2k = first operand of site k skipped, bit 2k+1 = second operand,
 * bit 63 = all points of the brick share x and y). */
int sdfk_debug_brick_masks(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, uint64_t* d_masks,
                           void* stream);
/* The same for the row-block kernel of sdfk_eval_device_rows: 3 words per brick {skip bits of sites 0-31, of sites
 * 32-63, kind: 1 = every row segment has one x and one y, 0 = not}. Bricks are WINDOWS of 32 points aligned in the flat
 * array: window k of row r is line floor(r * row_len / 32) + k, so it starts up to 31 points before the row when
 * row_len is no multiple of 32; brick q covers window q % nchunk of rows [brick_rows * (q / nchunk), + brick_rows),
 * nchunk = row_len / 32 if that is exact, else (row_len + 62) / 32. With d_masks == NULL only *n_bricks and
 * *brick_rows are returned. */
int sdfk_debug_row_masks(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len,
                         uint64_t* d_masks, int64_t* n_bricks, int* brick_rows, void* stream);

/* ---- staged evaluation: grid-neighbourhood modifications ------------------------------------------
 * signed / conv_averaging / conv_edge_detection (cores/modifications.py:220-275, 1589-1637) reshape the field to
 * the grid and look at neighbours, so a tree that contains them is evaluated in stages: the sub-tree below the
 * operator with an ordinary program, the operator on the resident field (below), and the rest of the tree with a
 * program whose V_FIELD instructions read that field as auxiliary input c: d_aux + c*aux_stride + point index.
 * Programs with V_FIELD instructions must be run through the _aux entry points (the others refuse them). */
int sdfk_eval_device_aux(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, const float* d_aux,
                         int n_aux, int64_t aux_stride, float* d_out, void* stream, int mode);
int sdfk_eval_grid_aux(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1, const float* ax2,
                       int64_t n2, int64_t start, int64_t count, const float* d_aux, int n_aux, int64_t aux_stride,
                       float* d_out, void* stream, int mode);
/* Operators on a DEVICE field of n0*n1*n2 fp32 laid out like the (N,) output (flat index (i*n1 + j)*n2 + k; 2-D
 * fields: n2 = 1), in place, synchronous:
 *   box average  = post_processing.conv_averaging (cores/post_processing.py:561-600): scipy.ndimage.convolve with
 *                  ones(k0,k1,k2)/(k0*k1*k2), mode "reflect", `iterations` times;
 *   edge detect  = post_processing.conv_edge_detection (:603-623): [[-1,-1,-1],[-1,8,-1],[-1,-1,-1]] on axes 0, 1;
 *   signed       = ModifyObject.signed (cores/modifications.py:220-275): unchanged if the field has a negative value,
 *                  else boundary = field < sep_min (the smallest grid spacing), scan-line parity along axes 0 and 1,
 *                  2x2x1 average, inner crop + edge pad (crop = 0: signed_old, :163-218, without it),
 *                  field *= (1 - 2*(average > 0.5)). */
/* d_field: 16-byte aligned for sdfk_field_min (it reads 16-byte pieces); sdfk_grid_signed takes any float alignment. */
int sdfk_field_min(const float* d_field, int64_t n, float* out_min, void* stream);
/* d_scratch: n0*n1*n2*4 bytes of device memory for the operator's work arrays (NULL: allocated and freed inside).
 * sdfk_grid_signed keeps its work as bit planes (six arrays of n0*n1*ceil(n2/32) words) inside it and allocates its
 * own few bytes for grids thinner than 11 points along the last axis. */
int sdfk_grid_box_average(float* d_field, int64_t n0, int64_t n1, int64_t n2, int k0, int k1, int k2, int iterations,
                          void* d_scratch, void* stream);
int sdfk_grid_edge_detect(float* d_field, int64_t n0, int64_t n1, int64_t n2, void* d_scratch, void* stream);
int sdfk_grid_signed(float* d_field, int64_t n0, int64_t n1, int64_t n2, float sep_min, int crop, void* d_scratch,
                     void* stream);
/* `signed` on ONE SLAB of a grid that is sharded over several GPUs: the scan lines cross every slab, but all they
 * read is one bit per point (field < sep_min). sdfk_grid_boundary_mask writes that test for n points as bytes; the
 * ranks exchange the bytes (aegolius_amd/distributed.py); sdfk_grid_signed_slab runs the scans on the WHOLE grid's
 * mask (n0 * n1 * n2 bytes) and flips the sign of the planes [plane0, plane0 + planes) held in d_slab. The "already
 * signed" test of the reference (modifications.py:236-237) is the caller's, as a minimum over all ranks. */
int sdfk_grid_boundary_mask(const float* d_field, int64_t n, float sep_min, unsigned char* d_mask, void* stream);
int sdfk_grid_signed_slab(float* d_slab, int64_t plane0, int64_t planes, const unsigned char* d_mask, int64_t n0, int64_t n1,
                          int64_t n2, int crop, void* d_scratch, void* stream);

/* ---- consumers of a resident field -------------------------------------------------------------
 * What the reference does with the (N,) field right after create(), on the device, synchronous:
 *   select    = the mask of GenericGeometry.point_cloud (cores/geom.py:62-74): the indices i, ascending, with
 *               field[i] <= threshold (NaN never selected). *count always receives the size of the selection;
 *               d_index == NULL counts only; otherwise d_index (DEVICE, capacity entries) must hold the selection.
 *               d_scratch: sdfk_field_select_scratch(n) bytes (about n / 8) of 8-byte-aligned device memory (NULL:
 *               allocated and freed inside). The field is read once: the count pass leaves 4 flag bits per quad in
 *               the scratch, the scatter pass works from those.
 *               d_field must be 16-byte aligned.
 *   gradient  = vector_functions.from_sdf (cores/vector_functions.py:130-140): numpy.gradient with unit spacing
 *               (central differences, one-sided on the faces) over the LAST ncomp axes of the (n0, n1, n2) field
 *               (the others must have length 1; every differentiated axis needs >= 2 points), then, if
 *               normalize != 0, batch_normalize (cores/vector_modification_functions.py:14-20): each vector divided
 *               by its norm unless the norm is 0. fp32 arithmetic: the raw gradient (normalize == 0) equals
 *               numpy's float64 result rounded to fp32, the direction is within 1e-6 per component. Row r of
 *               d_vec (DEVICE, ncomp rows of row_stride floats) is the derivative along axis 3 - ncomp + r;
 *               d_field, d_vec and row_stride * 4 must be 16-byte aligned. */
size_t sdfk_field_select_scratch(int64_t n);
/* Second half of a selection whose count-only call (d_index == NULL, caller-owned d_scratch) has just run on a field
 * of n points: writes the `count` indices that call announced without reading the field again. */
int sdfk_field_select_finish(int64_t n, int64_t count, int64_t* d_index, int64_t capacity, void* d_scratch, void* stream);

/* FUSED selection (GenericGeometry.point_cloud, cores/geom.py:62-74, without a field): the evaluation kernels write one
 * bit per point — field <= threshold — instead of the field (12 B/point of coordinates in, 1/8 B/point out), and count,
 * scan and scatter work on those flags alone; the indices are numpy.flatnonzero(tree(co) <= threshold), ascending, and
 * the bits are those of sdfk_eval_device's field. Same two-step protocol as sdfk_field_select: d_index == NULL returns
 * the count and keeps the flags in d_scratch (sdfk_eval_select_scratch(n, row_len) bytes of device memory, 8-byte aligned) for
 * sdfk_eval_select_finish. row_len / flat: the layout hints of sdfk_eval_device_rows / _rows2d (0: none). Runs on the
 * specialised kernels (the call waits for their build); programs with auxiliary fields are refused. */
size_t sdfk_eval_select_scratch(int64_t n, int64_t row_len);
int sdfk_eval_device_select(sdfk_program* prog, const float* d_co, int64_t n, int64_t row_stride, int64_t row_len, int flat,
                            float threshold, int64_t* d_index, int64_t capacity, int64_t* count, void* d_scratch,
                            void* stream, int mode);
int sdfk_eval_grid_select(sdfk_program* prog, const float* ax0, int64_t n0, const float* ax1, int64_t n1, const float* ax2,
                          int64_t n2, int64_t start, int64_t count_points, float threshold, int64_t* d_index,
                          int64_t capacity, int64_t* count, void* d_scratch, void* stream, int mode);
/* row_len, mode: as in the count-only call before it (grids: the last grid dimension longer than one point when the range
 * starts at a row boundary, else 0) — they decide how the flags are laid out. */
int sdfk_eval_select_finish(sdfk_program* prog, int64_t n, int64_t row_len, int mode, int64_t count, int64_t* d_index,
                            int64_t capacity, void* d_scratch, void* stream);
int sdfk_field_select(const float* d_field, int64_t n, float threshold, int64_t* d_index, int64_t capacity,
                      int64_t* count, void* d_scratch, void* stream);
int sdfk_field_gradient(const float* d_field, int64_t n0, int64_t n1, int64_t n2, int ncomp, int normalize,
                        float* d_vec, int64_t row_stride, void* stream);

/* ---- vector-field programs ----------------------------------------------------------------------
 * The reference's vector-field path (cores/geom.py:213-362 VectorField; cores/vector_functions.py:15-127 field
 * definitions; cores/modifications.py:1666-1975 ModifyVectorObject; cores/vector_modification_functions.py:14-160)
 * as ONE pointwise kernel: instruction 0 turns the input triple p into a vector, the others are the modifications
 * in the order they were applied. op = opcode | kindA << 8 | kindB << 12; kinds: 0 none, 1 number (imm[0] for A,
 * imm[3] for B), 2 3-vector (imm[0..2]), 3 one row of `streams` (src = row index: a per-point number), 4 three rows
 * (src = first row: a per-point vector), 5 the input triple p itself (revolutions about the axes of the same
 * coordinates). Opcodes (A / B operands):
 *   0 p itself | 1 (r, phi, theta) = p | 2 (r, phi, z) = p | 3 p/|p| | 4 (x, y, 0)/|(x, y)| | 5 vortex |
 *   6 radial-cylindrical turned by A | 7 vortex turned by A | 8 constant A | 9 rows A                (initialisers)
 *   10 v + A | 11 v - A | 12 v * A | 13 turn about z by A | 14 about x | 15 about y | 16 polar turn by A |
 *   17 turn about axis A by angle B | 18 / 19 / 20 revolve about x / y / z with coordinates A | 21 normalise.
 * out_kind: 0 the vector (3 rows) | 1 x | 2 y | 3 z | 4 atan2(y, x) | 5 acos(z) | 6 length (1 row).
 * fp32 arithmetic; zero vectors stay zero under normalisation. Asynchronous on `stream` (device flavour). */
typedef struct {
    int32_t op;
    int32_t src[2];
    float imm[4];
} sdfk_vec_instr;
/* d_p, d_out: 16-byte aligned, row strides multiples of 4 floats; d_streams: n_streams rows of stream_stride floats. */
int sdfk_vec_eval_device(const sdfk_vec_instr* prog, int n_instr, const float* d_p, int64_t n, int64_t p_stride,
                         const float* d_streams, int n_streams, int64_t stream_stride, int out_kind, float* d_out,
                         int64_t out_stride, void* stream);
/* By default a chain's TOPOLOGY (opcodes, operand kinds, stream rows, read-out) is compiled once per process with
 * hiprtc into a straight-line kernel (immediates stay run-time values); sdfk_vec_set_interpret(1) — or
 * SDFK_VEC_INTERPRET=1 in the environment — runs the interpreter kernel instead. Both give the same bits.
 * sdfk_vec_source: the generated HIP source (valid until the next call on the thread); sdfk_vec_compile_check:
 * compile it for gfx950 without a device. */
void sdfk_vec_set_interpret(int on);
const char* sdfk_vec_source(const sdfk_vec_instr* prog, int n_instr, int n_streams, int out_kind);
int sdfk_vec_compile_check(const sdfk_vec_instr* prog, int n_instr, int n_streams, int out_kind, size_t* code_size);
/* HOST arrays staged through the device in chunks: p (dtype 0 = fp32, 1 = fp64; (3, n) contiguous) or, when ax0..ax2
 * are given, the generate_grid cloud of those tables (expanded on the device); streams (n_streams, n) fp32; out (3, n)
 * or (n,) fp32. */
int sdfk_vec_eval_host(const sdfk_vec_instr* prog, int n_instr, const void* p, int p_dtype, int64_t n, const float* ax0,
                       int64_t n0, const float* ax1, int64_t n1, const float* ax2, int64_t n2, const float* streams,
                       int n_streams, int out_kind, float* out, int device);

/* ---- grid builder -----------------------------------------------------------------------------
 * numpy.linspace(lo, hi, n) in float64 (step = (hi-lo)/(n-1); y[i] = i*step + lo; y[n-1] = hi),
 * rounded to fp32 — the per-axis table of generate_grid (cores/helper_functions.py:56-88). */
int sdfk_linspace_f32(double lo, double hi, int64_t n, float* out);
/* ---- nearest-point tables ---------------------------------------------------------------------
 * Box tree over a point table for the nearest-point leaves (point clouds, resampled / parametric curves, curve
 * instancing: C/sdf_2D.py:221-224, C/sdf_3D.py:283-286 build a scipy KDTree per evaluation; here the tree is part of the
 * program's tables, built once per lowering). Host only, no GPU. pts: (m, 3) fp32, row-major; leaf: points per leaf box
 * = children per box on every level above (the lowering uses 32); table: out — root boxes, middle boxes, leaf boxes (8
 * floats each: lo, hi, first child, children), then the points in leaf order —, capacity table_cap floats (3 m + 24 (m /
 * (leaf / 2) + 4) always suffices); order: out or NULL, the original index of every point in leaf order; n_root /
 * point_base: out, the number of root boxes and the offset of the first point. Returns the number of floats written,
 * < 0 on error (-2: more than 2^24 words — table indices are carried as fp32). */
int64_t sdfk_point_tree_build(const float* pts, int64_t m, int leaf, float* table, int64_t table_cap, int64_t* order,
                              int64_t* n_root, int64_t* point_base);

/* Fill a device (3, count) coordinate slab for flat indices [start, start+count) of the grid. */
int sdfk_grid_fill(float* d_co, int64_t row_stride, const float* ax0, int64_t n0, const float* ax1, int64_t n1,
                   const float* ax2, int64_t n2, int64_t start, int64_t count, void* stream);

/* ---- device memory / timing plumbing (so callers need no HIP binding of their own) ------------ */
int sdfk_set_device(int device);
void* sdfk_malloc(size_t bytes);
int sdfk_free(void* d_ptr);
int sdfk_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
int sdfk_memcpy_d2h(void* dst, const void* d_src, size_t bytes);
int sdfk_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);
int sdfk_sync(void* stream);
void* sdfk_event_create(void);
int sdfk_event_destroy(void* ev);
int sdfk_event_record(void* ev, void* stream);
int sdfk_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */
/* plain (3,n)->(n) streaming kernel out = x+y+z with the same access pattern: measured HBM ceiling */
int sdfk_stream_probe(const float* d_co, int64_t n, int64_t row_stride, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFK_H */
