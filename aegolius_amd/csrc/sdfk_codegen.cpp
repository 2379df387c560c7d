// sdfk_codegen.cpp — program -> HIP source for hiprtc (see sdfk_codegen.h).
#include "sdfk_codegen.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>

// text of sdfk_device.h / sdfk_access.h, generated at build time by __graft_entry__.build()
#include "sdfk_embedded.inc"

std::string sdfk_vector_prelude() { return std::string(kEmbeddedDevice) + "\n" + kEmbeddedVecdev + "\n"; }

static const char kWrappers[] = R"SDFKW(
template <int VEC, typename SRC>
static __device__ __forceinline__ void sdfk_body(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                 const SRC& src, long long off, long long n,
                                                 float* __restrict__ out, const float* __restrict__ aux,
                                                 long long aux_stride, unsigned* __restrict__ flags, unsigned thr) {
    const long long block_base = (long long)sdfk_bx() * (SDFK_BLOCK * VEC);
    const unsigned lane_off = sdfk_tx() * VEC;
#ifdef SDFK_FLAGS                                              // the flag-writing build of this flavour (its own code object:
    {                                                          // carried as a run-time branch it cost the field kernels registers)
        const bool active = block_base + lane_off < n;
        unsigned nib = 0u;
        if (active) {
            V3 p[VEC];
            sdfk_load<VEC>(src, off + block_base, lane_off, p);
            const float* ax = aux + (off + block_base + lane_off);
            if constexpr (VEC == 4) {
                const f2 ra = sdfk_point<f2>(sd_join(p[0], p[1]), PRM, TAB, ax, aux_stride);
                const f2 rb = sdfk_point<f2>(sd_join(p[2], p[3]), PRM, TAB, ax + 2, aux_stride);
                nib = (sdfk_sel_key(ra.x) <= thr ? 1u : 0u) | (sdfk_sel_key(ra.y) <= thr ? 2u : 0u) |
                      (sdfk_sel_key(rb.x) <= thr ? 4u : 0u) | (sdfk_sel_key(rb.y) <= thr ? 8u : 0u);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    nib |= (sdfk_sel_key(sdfk_point<float>(p[k], PRM, TAB, ax + k, aux_stride)) <= thr ? 1u : 0u) << k;
            }
        }
        sdfk_store_flags<VEC>(flags, off + block_base + lane_off, nib, active);
        return;
    }
#endif
    if (block_base + lane_off >= n) return;
    V3 p[VEC];
    sdfk_load<VEC>(src, off + block_base, lane_off, p);
    const float* ax = aux + (off + block_base + lane_off);   // only dereferenced by V_FIELD instructions
    float v[VEC];
    if constexpr (VEC == 4) {
        // two points per lane value: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
        const f2 ra = sdfk_point<f2>(sd_join(p[0], p[1]), PRM, TAB, ax, aux_stride);
        const f2 rb = sdfk_point<f2>(sd_join(p[2], p[3]), PRM, TAB, ax + 2, aux_stride);
        v[0] = ra.x; v[1] = ra.y; v[2] = rb.x; v[3] = rb.y;
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = sdfk_point<float>(p[k], PRM, TAB, ax + k, aux_stride);
    }
    sdfk_store<VEC>(out, off + block_base + lane_off, v);
}
)SDFKW";
static const char kWrappersArray[] = R"SDFKW(
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_v4(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long off, long long n, float* __restrict__ out, const float* __restrict__ aux, long long aux_stride,
    unsigned* __restrict__ flags, unsigned thr) {
    SrcArray s = {co, stride};
    sdfk_body<4>(PRM, TAB, s, off, n, out, aux, aux_stride, flags, thr);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_v1(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long off, long long n, float* __restrict__ out, const float* __restrict__ aux, long long aux_stride,
    unsigned* __restrict__ flags, unsigned thr) {
    SrcArray s = {co, stride};
    sdfk_body<1>(PRM, TAB, s, off, n, out, aux, aux_stride, flags, thr);
}
)SDFKW";
static const char kWrappersGrid[] = R"SDFKW(
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_g4(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid g, long long off, long long n,
    float* __restrict__ out, const float* __restrict__ aux, long long aux_stride, unsigned* __restrict__ flags, unsigned thr) {
    sdfk_body<4>(PRM, TAB, g, off, n, out, aux, aux_stride, flags, thr);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_g1(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid g, long long off, long long n,
    float* __restrict__ out, const float* __restrict__ aux, long long aux_stride, unsigned* __restrict__ flags, unsigned thr) {
    sdfk_body<1>(PRM, TAB, g, off, n, out, aux, aux_stride, flags, thr);
}
)SDFKW";

// Pieces shared by the two culling kernel families.
static const char kWaveHelpers[] = R"SDFKH(
#ifndef SDFK_TWAVES
#define SDFK_TWAVES 4           // waves per workgroup           (host launch code must agree: sdfk.hip)
#endif
#define SDFK_TTHREADS (64 * SDFK_TWAVES)
// wave-wide max of a non-negative value (DPP: row_shr 1,2,4,8, row_bcast 15 / 31); result valid in every lane
static __device__ __forceinline__ float sdfk_wave_max(float v) {
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false)));
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false)));
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false)));
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false)));
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false)));
    v = sd_rawmax(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false)));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
static __device__ __forceinline__ float sdfk_lane(float v, int l) {     // l wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

)SDFKH";

// Brick-culling tile kernel (only emitted when the program has cull sites).
// A brick = SDFK_BRICK (128) consecutive points = one wave x one packed f2 lane value. A workgroup of
// SDFK_TWAVES waves owns SDFK_TWAVES * SDFK_WBRICKS bricks (a "tile").
//   phase A  every wave loads its bricks (8-byte loads: a lane holds the same two points it will evaluate, so
//            z stays in registers), classifies each brick and reduces its bounding sphere(s) (wave-wide DPP)
//   phase B  one lane per run evaluates the WHOLE tree at the run's centre and turns the operand gaps at
//            every combiner into a skip mask (exact: see sdfk_probe); LDS carries only per-brick metadata
//   phase C  every wave evaluates its bricks, jumping over the subtrees the mask proves irrelevant
//            (the mask is wave-uniform: scalar branches)
// Brick kinds (bitwise comparisons of the coordinates):
//   1  all points share x and y (inside one row of a regular grid): x, y kept once, the first two thirds of
//      every root transform are computed once by the probe lane (op_xform_base)
//   2  two consecutive runs of kind 1 (a brick straddling the end of a row): probed once per run, a subtree
//      is skipped only if both runs allow it
//   0  anything else: one bounding sphere, x / y re-read from global memory in phase C
static const char kTileKernel[] = R"SDFKT(
#ifndef SDFK_WBRICKS
#define SDFK_WBRICKS 4          // bricks per wave
#endif
#define SDFK_BRICK 128
#define SDFK_NBRICK (SDFK_TWAVES * SDFK_WBRICKS)
#define SDFK_TILE (SDFK_NBRICK * SDFK_BRICK)
static_assert(2 * SDFK_NBRICK <= SDFK_TTHREADS, "one probe lane per run");

struct sdfk_tilemeta {
    float z[SDFK_TILE];                                // z of every point, written and read by the owning wave
    float4 bound[SDFK_NBRICK], bound2[SDFK_NBRICK];   // bounding spheres: centre, radius (second run of kind 2)
    unsigned long long mask[SDFK_NBRICK], mask2[SDFK_NBRICK];
    float2 xy[SDFK_NBRICK], xy2[SDFK_NBRICK];
    float base[SDFK_NBRICK][3 * SDFK_NROOT], base2[SDFK_NBRICK][3 * SDFK_NROOT];
    unsigned kind[SDFK_NBRICK], split[SDFK_NBRICK];   // split = index of the first point of the second run
};

// phase A for one brick: classify and bound; lane 0 publishes the metadata
static __device__ __forceinline__ void sdfk_brick_bounds(f2 x, f2 y, f2 z, int lane, sdfk_tilemeta* meta, int b) {
    const float fx = sdfk_lane(x.x, 0), fy = sdfk_lane(y.x, 0), fz = sdfk_lane(z.x, 0), lz = sdfk_lane(z.y, 63);
    const bool e0 = x.x == fx && y.x == fy, e1 = x.y == fx && y.y == fy;
    const unsigned long long q0 = __ballot(e0), q1 = __ballot(e1);
    unsigned kind;
    unsigned split = SDFK_BRICK;
    float4 bnd, bnd2;
    float2 xy2 = make_float2(fx, fy);
    if ((q0 & q1) == ~0ull) {                                   // one run: 1-D bound
        kind = 1u;
        const float cz = 0.5f * (fz + lz);
        const float r = sdfk_wave_max(sd_rawmax(sd_abs(z.x - cz), sd_abs(z.y - cz)));
        bnd = make_float4(fx, fy, cz, 1.00001f * r + 1e-30f);
        bnd2 = bnd;
    } else {
        // index of the first point that differs from the first one
        const int s0 = (~q0 == 0ull) ? 64 : __builtin_ctzll(~q0), s1 = (~q1 == 0ull) ? 64 : __builtin_ctzll(~q1);
        const int s = (2 * s0 < 2 * s1 + 1) ? 2 * s0 : 2 * s1 + 1;
        const int sl = s >> 1, pl = (s - 1) >> 1;
        const float x2 = (s & 1) ? sdfk_lane(x.y, sl) : sdfk_lane(x.x, sl), y2 = (s & 1) ? sdfk_lane(y.y, sl) : sdfk_lane(y.x, sl);
        const float zs = (s & 1) ? sdfk_lane(z.y, sl) : sdfk_lane(z.x, sl);
        const float zp = s > 0 ? (((s - 1) & 1) ? sdfk_lane(z.y, pl) : sdfk_lane(z.x, pl)) : fz;
        const bool k0 = (2 * lane < s) ? e0 : (x.x == x2 && y.x == y2), k1 = (2 * lane + 1 < s) ? e1 : (x.y == x2 && y.y == y2);
        const bool two = s > 0 && (__ballot(k0 && k1) == ~0ull);
        if (two) {
            kind = 2u;
            split = (unsigned)s;
            const float c1z = 0.5f * (fz + zp), c2z = 0.5f * (zs + lz);
            const float ra = sdfk_wave_max(sd_rawmax(2 * lane < s ? sd_abs(z.x - c1z) : 0.0f, 2 * lane + 1 < s ? sd_abs(z.y - c1z) : 0.0f));
            const float rb = sdfk_wave_max(sd_rawmax(2 * lane < s ? 0.0f : sd_abs(z.x - c2z), 2 * lane + 1 < s ? 0.0f : sd_abs(z.y - c2z)));
            bnd = make_float4(fx, fy, c1z, 1.00001f * ra + 1e-30f);
            bnd2 = make_float4(x2, y2, c2z, 1.00001f * rb + 1e-30f);
            xy2 = make_float2(x2, y2);
        } else {                                                // arbitrary points: sphere around the midpoint of the
            kind = 0u;                                          // first and the last point
            const float cx = 0.5f * (fx + sdfk_lane(x.y, 63)), cy = 0.5f * (fy + sdfk_lane(y.y, 63)), cz = 0.5f * (fz + lz);
            const f2 dx = x - cx, dy = y - cy, dz = z - cz;
            const f2 d2 = sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
            const float r2 = sdfk_wave_max(sd_rawmax(d2.x, d2.y));
            bnd = make_float4(cx, cy, cz, 1.00001f * sqrtf(r2) + 1e-30f);
            bnd2 = bnd;
        }
    }
    if (lane == 0) {
        meta->bound[b] = bnd;
        meta->bound2[b] = bnd2;
        meta->kind[b] = kind;
        meta->split[b] = split;
        meta->xy[b] = make_float2(fx, fy);
        meta->xy2[b] = xy2;
    }
}

struct sdfk_tileregs {
    f2 x[SDFK_WBRICKS], y[SDFK_WBRICKS], z[SDFK_WBRICKS];
};
// points (i, i+1) of a brick whose first point has index brick_base (wave-uniform); indices beyond n-1 repeat the
// last point (keeps the bounds valid at the ragged end)
static __device__ __forceinline__ void sdfk_pair(const SrcArray& s, long long brick_base, int lane, long long n,
                                                 f2& x, f2& y, f2& z) {
    const long long i = brick_base + 2 * lane;
    if (i + 1 < n) {
        x = *reinterpret_cast<const f2*>(s.co + i);
        y = *reinterpret_cast<const f2*>(s.co + s.stride + i);
        z = *reinterpret_cast<const f2*>(s.co + 2 * s.stride + i);
    } else {
        const long long last = n - 1;
        const long long i0 = i < last ? i : last;
        x = {s.co[i0], s.co[last]};
        y = {s.co[s.stride + i0], s.co[s.stride + last]};
        z = {s.co[2 * s.stride + i0], s.co[2 * s.stride + last]};
    }
}
// regular grid: the 64-bit divisions act on the wave-uniform brick base (scalar unit), the per-lane part is 32-bit
static __device__ __forceinline__ void sdfk_pair(const SrcGrid& s, long long brick_base, int lane, long long n,
                                                 f2& x, f2& y, f2& z) {
    const long long last = n - 1;
    const long long bb = brick_base < last ? brick_base : last;
    const unsigned long long base = (unsigned long long)(s.start + bb);
    const unsigned long long row = base / s.n2;
    const unsigned iz0 = (unsigned)(base - row * s.n2);
    const unsigned long long ix0 = row / s.n1;
    const unsigned iy0 = (unsigned)(row - ix0 * s.n1);
    const long long room64 = last - bb;                        // points of the array after the brick base
    const unsigned room = room64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)room64;
    if (iz0 + SDFK_BRICK <= s.n2 && room >= SDFK_BRICK - 1) {  // the whole brick lies inside one grid row (uniform)
        const float gx = s.ax0[ix0], gy = s.ax1[iy0];          // scalar loads
        x = {gx, gx};
        y = {gy, gy};
        z = {s.ax2[iz0 + 2 * lane], s.ax2[iz0 + 2 * lane + 1]};
        return;
    }
    float px[2], py[2], pz[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        unsigned off = 2u * (unsigned)lane + (unsigned)k;
        off = off < room ? off : room;
        const unsigned t = iz0 + off;
        const unsigned cz = t / s.n2, iz = t - cz * s.n2;
        const unsigned ty = iy0 + cz;
        const unsigned cy = ty / s.n1, iy = ty - cy * s.n1;
        px[k] = s.ax0[ix0 + cy];
        py[k] = s.ax1[iy];
        pz[k] = s.ax2[iz];
    }
    x = {px[0], px[1]};
    y = {py[0], py[1]};
    z = {pz[0], pz[1]};
}
// issue the loads of one tile (this wave's bricks)
template <typename SRC>
static __device__ __forceinline__ void sdfk_tile_load(const SRC& src, long long n, long long tile_base, int lane, int wave,
                                                      sdfk_tileregs& r) {
#pragma unroll
    for (int j = 0; j < SDFK_WBRICKS; ++j)
        sdfk_pair(src, tile_base + (long long)(wave * SDFK_WBRICKS + j) * SDFK_BRICK, lane, n, r.x[j], r.y[j], r.z[j]);
}
// phase A: bounds of this wave's bricks from registers, z to LDS
static __device__ __forceinline__ void sdfk_tile_bounds(const sdfk_tileregs& r, sdfk_tilemeta* meta, int lane, int wave) {
#pragma unroll
    for (int j = 0; j < SDFK_WBRICKS; ++j) {
        const int b = wave * SDFK_WBRICKS + j;
        *reinterpret_cast<f2*>(meta->z + b * SDFK_BRICK + 2 * lane) = r.z[j];
        sdfk_brick_bounds(r.x[j], r.y[j], r.z[j], lane, meta, b);
    }
}
// phase B: one lane per run (pl = index of the lane among the probing lanes)
static __device__ __forceinline__ void sdfk_tile_probe(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                       sdfk_tilemeta* meta, int pl) {
    if (pl < 2 * SDFK_NBRICK) {
        const int b = pl % SDFK_NBRICK, second = pl / SDFK_NBRICK;
        const unsigned kind = meta->kind[b];
        if (!second || kind == 2u) {
            const float4 bb = second ? meta->bound2[b] : meta->bound[b];
            V3T<float> c = {bb.x, bb.y, bb.z};     // for an x/y-constant run the centre has exactly that x and y
            const unsigned long long m = sdfk_probe(c, bb.w, second ? meta->base2[b] : meta->base[b], PRM, TAB);
            if (second) meta->mask2[b] = m;
            else meta->mask[b] = m | ((unsigned long long)(kind == 1u) << 63) | ((unsigned long long)(kind == 2u) << 62);
        }
    }
}
// skip bits of a brick: both runs of a two-run brick must agree
static __device__ __forceinline__ unsigned long long sdfk_brick_mask(const sdfk_tilemeta* meta, int b) {
    unsigned long long m = meta->mask[b];
    if ((m >> 62) & 1ull) m &= meta->mask2[b] | (3ull << 62);
    return m;
}
// phase C: evaluate this wave's bricks of the prepared tile
template <typename SRC>
static __device__ __forceinline__ void sdfk_tile_evaluate(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                          const SRC& src, long long n, long long tile_base,
                                                          const sdfk_tilemeta* meta, float* __restrict__ out, int lane,
                                                          int wave) {
#pragma unroll 1
    for (int j = 0; j < SDFK_WBRICKS; ++j) {
        const int b = wave * SDFK_WBRICKS + j;
        const long long i = tile_base + (long long)b * SDFK_BRICK + 2 * lane;
        if (tile_base + (long long)b * SDFK_BRICK >= n) break;                        // bricks past the end
        const unsigned long long mv = sdfk_brick_mask(meta, b);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)mv);            // wave-uniform (SGPRs)
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(mv >> 32));
        const f2 z = *reinterpret_cast<const f2*>(meta->z + b * SDFK_BRICK + 2 * lane);
        f2 r;
        if (hi >> 31) {                                                               // one run
            const float2 xy = meta->xy[b];
            V3P p = {sp<f2>(xy.x), sp<f2>(xy.y), z};
            r = sdfk_point_culled<f2, true>(p, lo, hi, meta->base[b], PRM, TAB);
        } else {
            V3P p;
            if ((hi >> 30) & 1u) {                                                    // two runs
                const float2 xa = meta->xy[b], xb = meta->xy2[b];
                const int s = (int)meta->split[b];
                const bool a0 = 2 * lane < s, a1 = 2 * lane + 1 < s;
                p = {{a0 ? xa.x : xb.x, a1 ? xa.x : xb.x}, {a0 ? xa.y : xb.y, a1 ? xa.y : xb.y}, z};
            } else {                                                                  // arbitrary points: x, y again
                f2 zz;                                                                // (L2-resident: loaded in phase A)
                sdfk_pair(src, tile_base + (long long)b * SDFK_BRICK, lane, n, p.x, p.y, zz);
                p.z = z;
            }
            r = sdfk_point_culled<f2, false>(p, lo, hi, meta->base[b], PRM, TAB);
        }
        if (i + 1 < n) *reinterpret_cast<f2*>(out + i) = r;
        else if (i < n) out[i] = r.x;
    }
}
// phases A and B of one tile (two workgroup barriers)
template <typename SRC>
static __device__ __forceinline__ void sdfk_tile_prepare(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                         const SRC& src, long long n, sdfk_tilemeta* meta) {
    const int lane = sdfk_tx() & 63, wave = sdfk_tx() >> 6;
    sdfk_tileregs r;
    sdfk_tile_load(src, n, (long long)sdfk_bx() * SDFK_TILE, lane, wave, r);
    sdfk_tile_bounds(r, meta, lane, wave);
    __syncthreads();
    sdfk_tile_probe(PRM, TAB, meta, sdfk_tx());
    __syncthreads();
}
// One tile per workgroup; many short-lived workgroups per CU sit in different phases at any time, which is
// what overlaps the memory phase (A) with the VALU phases (B, C). Two persistent variants (register prefetch
// of the next tile; a dedicated probing wave) were measured 1.5-2.5x slower: with few long-lived workgroups
// per CU the phases run in lockstep and the single-wave probe leaves the CU idle.
template <typename SRC>
static __device__ __forceinline__ void sdfk_tile_kernel(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                        const SRC& src, long long n, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) sdfk_tilemeta meta;
    sdfk_tile_prepare(PRM, TAB, src, n, &meta);
    sdfk_tile_evaluate(PRM, TAB, src, n, (long long)sdfk_bx() * SDFK_TILE, &meta, out, sdfk_tx() & 63,
                       sdfk_tx() >> 6);
}
)SDFKT";
static const char kTileArray[] = R"SDFKT(
extern "C" __global__ __launch_bounds__(SDFK_TTHREADS) void sdfk_spec_t(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long n, float* __restrict__ out) {
    const SrcArray s = {co, stride};
    sdfk_tile_kernel(PRM, TAB, s, n, out);
}
)SDFKT";
static const char kTileGrid[] = R"SDFKT(
// the same on a regular grid expanded from three per-axis tables: no coordinate array at all (4 B/point)
extern "C" __global__ __launch_bounds__(SDFK_TTHREADS) void sdfk_spec_tg(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid g, long long n, float* __restrict__ out) {
    sdfk_tile_kernel(PRM, TAB, g, n, out);
}
)SDFKT";
static const char kTileMask[] = R"SDFKT(
// debugging / test aid: the skip masks of every brick (2 bits per site: bit 2k = skip first operand,
// bit 2k+1 = skip second operand; bit 63 = x/y-constant run, bit 62 = two such runs)
extern "C" __global__ __launch_bounds__(SDFK_TTHREADS) void sdfk_spec_tmask(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long n, unsigned long long* __restrict__ masks) {
    __shared__ __attribute__((aligned(16))) sdfk_tilemeta meta;
    const SrcArray s = {co, stride};
    sdfk_tile_prepare(PRM, TAB, s, n, &meta);
    for (int b = sdfk_tx(); b < SDFK_NBRICK; b += SDFK_TTHREADS)
        masks[(long long)sdfk_bx() * SDFK_NBRICK + b] = sdfk_brick_mask(&meta, b);
}
)SDFKT";


// Geometry of the lane-parallel probe (emitted in front of the generated leaf functions).
static const char kSimtGeometry[] = R"SDFKR(
#ifndef SDFK_RWBRICKS
#define SDFK_RWBRICKS 2
#endif
#ifndef SDFK_RWAVES
#define SDFK_RWAVES 4
#endif
#ifndef SDFK_NSUB                        // sub-bricks (probe centres) per brick: 16 = 4 rows x 8 points, 8 = 8 rows x 8, 4 = 16 rows x 8
// measured (north-star tree, 20-primitive tree; 513^3 and 1025^3): 8 centres beat 16 — half the leaf evaluations, nearly
// the same radius — and beat 4 for 20 leaves too (513^3: 0.56 -> 0.49 ms, 1025^3: 1 %); beyond 32 leaves the probe itself
// becomes the cost: 4 centres up to 96 leaves (round 4, smooth unions of n primitives at 513^3, 1 -> 4 centres: n = 40
// 1.06 -> 0.95 ms, 60: 1.44 -> 1.30, 80: 1.58 -> 1.52, 100: 1.86 -> 1.85, 130: 2.48 -> 2.61), one centre beyond
#define SDFK_NSUB ((SDFK_NLEAF <= 32 && 8 * SDFK_RWBRICKS <= 64) ? 8 : ((SDFK_NLEAF <= 96 && 4 * SDFK_RWBRICKS <= 64) ? 4 : 1))
#endif
#define SDFK_NCEN (SDFK_RWAVES * SDFK_RWBRICKS * SDFK_NSUB)
static_assert(SDFK_NSUB == 1 || SDFK_NSUB == 4 || SDFK_NSUB == 8 || SDFK_NSUB == 16, "1, 4, 8 or 16 probe centres per brick");
static_assert(SDFK_NCEN <= 64 * SDFK_RWAVES, "one fold lane per probe centre");
)SDFKR";

// Row-block culling kernel (only emitted when the program has cull sites). The caller states that the
// points come as consecutive ROWS of L points (the last axis of a generate_grid meshgrid; any L dividing
// n is valid). A brick = SDFK_RZ (32) consecutive points of SDFK_RROWS (16) consecutive rows: on a
// regular grid a 32 x 16 x 1 block whose bounding sphere is 4x smaller than that of 128 points in a line,
// so far fewer subtrees survive the probe. Everything is still derived from the coordinates actually
// read — a wrong or meaningless L costs speed, never correctness.
//   phase A  8 rows x 32 points per load instruction (rows need only 4-byte alignment); bounding sphere
//            around the midpoint of the first and last point; "uniform" = every row has one x and one y
//   phase B  one lane per brick evaluates the whole tree at the centre -> skip mask (up to 64 sites)
//   phase C  a lane owns 2*SDFK_NP consecutive points of ONE row: on uniform bricks the x/y part of every
//            root transform is computed once per lane (op_xform_base) and shared by its points; control
//            flow, parameter loads and branches are shared by SDFK_NP packed pairs per lane
static const char kRowsKernel[] = R"SDFKR(
#ifndef SDFK_RWBRICKS
#define SDFK_RWBRICKS 2
#endif
#ifndef SDFK_RWAVES
#define SDFK_RWAVES 4                           // waves per workgroup (host launch code must agree: sdfk.hip)
#endif
#define SDFK_RZ 32
#define SDFK_RLPR (SDFK_RZ / (2 * SDFK_NP))     // lanes per row in phase C
#define SDFK_RROWS (64 / SDFK_RLPR)             // rows per brick
#define SDFK_RBRICK (SDFK_RZ * SDFK_RROWS)
#define SDFK_RLOADS (SDFK_RROWS / 8)            // load instructions per array and brick
#define SDFK_RNBRICK (SDFK_RWAVES * SDFK_RWBRICKS)
#ifndef SDFK_CHUNK
#define SDFK_CHUNK 16                            // chain mode: children whose parameters are staged in LDS at a time
#endif
#ifndef SDFK_STAGE_MIN
#define SDFK_STAGE_MIN 2                         // ... when more than this many survive on the brick
#endif
#ifndef SDFK_CHAIN_STAGED                        // Round 3 staged the survivors' parameters in LDS for chains of more than 64
#define SDFK_CHAIN_STAGED 0                      // children (three dependent loads per child: kind, base, parameters). With the
#endif                                           // survivors kept as RECORDS (round 4) one load is left between the list and the
                                                 // parameters, and the direct path is faster: 1000 spheres at 513^3 1.08 -> 0.94 ms,
                                                 // 4096: 2.21 -> 2.11 (SDFK_STAGE_MIN=64 against 2); -DSDFK_CHAIN_STAGED=1 brings it back
static_assert(SDFK_NP == 2 || SDFK_NP == 4, "2 or 4 packed pairs per lane");
static_assert(SDFK_RNBRICK <= 64 * SDFK_RWAVES, "one probe lane per brick");

typedef float sdfk_f4u __attribute__((ext_vector_type(4), aligned(4)));

#ifdef SDFK_CELLS
#ifndef SDFK_ALIST_CAP
#define SDFK_ALIST_CAP (SDFK_NLEAF < 192 ? SDFK_NLEAF : 192)     // survivors a brick's list holds in LDS
#endif
#define SDFK_ALL_ALIVE 0xffffffffu
#define SDFK_LIST_ALIVE 0xfffffffeu
#endif
struct sdfk_rowmeta {
    float z[SDFK_RNBRICK][SDFK_RBRICK];         // [row of the brick][point of the window]
    float2 xy[SDFK_RNBRICK][SDFK_RROWS];
    float4 bound[SDFK_RNBRICK];
    unsigned long long mask[SDFK_NMASK][SDFK_RNBRICK];   // two bits per site: SDFK_NMASK x 32 sites (2 words up to 64 sites)
    unsigned uniform[SDFK_RNBRICK];
#if defined(SDFK_SIMT) && !defined(SDFK_CELLS)
    float4 cen[SDFK_NCEN];                      // probe centres (x, y, z, radius): SDFK_NSUB per brick
    float leafval[SDFK_NLEAF * SDFK_NCEN];      // [leaf][centre]: every leaf of the tree at every centre
#endif
#ifdef SDFK_CHAIN
#ifdef SDFK_CELLS
    unsigned alist[SDFK_RNBRICK][SDFK_ALIST_CAP];                      // RECORDS of the children that run, in order
    uint2 aspan[SDFK_RNBRICK];                                         // more survivors than that: the cell's whole list runs (nalive = LIST)
    unsigned inside[SDFK_RNBRICK];                                     // every point of the brick lies in its cell's sphere
#else
    unsigned short alist[SDFK_RNBRICK][SDFK_NLEAF];                    // the children that run, in order
#endif
    unsigned nalive[SDFK_RNBRICK];
#if SDFK_CHAIN_STAGED
    float cprm[SDFK_RNBRICK][SDFK_CHUNK][SDFK_NPLMAX];                  // parameters of the chunk of children being evaluated
    unsigned cgrp[SDFK_RNBRICK][SDFK_CHUNK];                            // ... and their kinds
#endif
#endif
#ifdef SDFK_LDSPAD
    float pad[SDFK_LDSPAD / 4];                 // experiment: a larger LDS footprint per workgroup
#endif
};
struct sdfk_rowgeom {
    unsigned L, nchunk, nbricks;                // row length, windows per row, bricks in total
    long long R;                                // rows
    long long row0;                             // grid flavour: global row index of the first row of the slab
    int yrows;                                  // grid flavour, 2-D grid (n2 == 1): rows run along the second axis
    unsigned prow, seg0, nb0, bpp;              // planes of prow rows after a first partial plane of seg0 rows (nb0 blocks);
                                                // bpp blocks per plane: a block never holds rows of two planes
    unsigned inv_nchunk, inv_bpp;               // floor(2^32 / nchunk), floor(2^32 / bpp): divisions as one v_mul_hi
};
#ifdef SDFK_CELLS
// Candidate lists (chain mode). The bricks are grouped into CELLS in index space — 2^lx plane slots x 2^ly row blocks x
// 2^lz windows — and a pre-pass (sdfk_spec_cells) leaves per cell a bounding sphere, derived from the coordinates of the
// cell's eight corner points, and the LIST of members that can be the minimum somewhere in that sphere (the global-minimum
// rule of sdfk_chain_fold at the cell's radius; cells of a coarser level first, so that a fine cell only looks at its
// parent's list). A brick then probes the members on its cell's list instead of all of them — after checking, point by
// point, that it really lies inside the sphere: the corner points bound the cell only if the array is the lattice its
// layout hints say it is, and a brick that fails the check (or whose cell ran out of list space) probes every member,
// as before. Whatever the input, a member that is dropped cannot be the minimum at any point of the brick.
struct sdfk_celllevel {
    unsigned lx, ly, lz;                        // log2 of the cell size in plane slots, row blocks, windows
    unsigned ncx, ncy, ncz;                     // cells along each of them
    unsigned xoff;                              // slot number of the first whole plane (2^lx when a partial plane leads, else 0)
    unsigned pad;
};
struct sdfk_cells {
    sdfk_celllevel lv;                          // the level the bricks look at (the finest)
    const float4* __restrict__ sph;             // per cell: centre, radius (< 0: no such cell)
    const uint2* __restrict__ span;             // per cell: first entry, entries (SDFK_ALL_ALIVE: no list — probe every member)
    const unsigned* __restrict__ cand;          // the lists: one RECORD per member (sdfk_member_record: kind and parameter block)
    unsigned enabled, ncells;
};
#endif
// a / d for wave-uniform a, with inv = floor(2^32 / d) from the host: the estimate is never too large and at most a few
// units too small (a * (2^32 - inv * d) < 2^32 * d for every brick count the host accepts), the loop repairs it.
// (A plain 32-bit division is ~35 vector instructions; this kernel has two per brick.)
static __device__ __forceinline__ unsigned sdfk_udiv(unsigned a, unsigned d, unsigned inv) {
    unsigned q = __umulhi(a, inv);
    while (a - q * d >= d) ++q;
    return __builtin_amdgcn_readfirstlane(q);
}
// rows [r0, rend) of row block rb (wave-uniform): 16 consecutive rows of ONE plane, fewer at the end of a plane
// (slot, blk: plane slot — 0 = the leading partial plane, whole planes from 0 as well; the caller adds its offset — and the
//  block's index within it)
static __device__ __forceinline__ void sdfk_block_rows(const sdfk_rowgeom& g, unsigned rb, long long& r0, long long& rend,
                                                       unsigned* slot = nullptr, unsigned* blk = nullptr) {
    // ONE plane (no plane hint, flat grids, plane blocks not asked for): blocks of 16 consecutive rows, nothing to divide
    if (g.nb0 == 0u && (long long)g.prow >= g.R) {
        r0 = (long long)rb * SDFK_RROWS;
        rend = g.R;
        if (slot) { *slot = 0u; *blk = rb; }
        return;
    }
    if (rb < g.nb0) {
        r0 = (long long)rb * SDFK_RROWS;
        rend = g.seg0;
        if (slot) { *slot = 0xffffffffu; *blk = rb; }
    } else {
        const unsigned k = rb - g.nb0;
        const unsigned pl = sdfk_udiv(k, g.bpp, g.inv_bpp);
        const long long base = (long long)g.seg0 + (long long)pl * g.prow;
        r0 = base + (long long)(k - pl * g.bpp) * SDFK_RROWS;
        rend = base + g.prow < g.R ? base + g.prow : g.R;
        if (slot) { *slot = pl; *blk = k - pl * g.bpp; }
    }
}

// Rows are cut into WINDOWS of 32 points that are aligned in the flat array (128-byte lines when the array is):
// window k of a row starts at flat index 32 * (floor(row * L / 32) + k), i.e. up to 31 points before the row
// when L is no multiple of 32. Misaligned 128-byte segments cost 20-35 % of the streaming rate on MI355X
// (tools/rowstream.hip), whole lines cost nothing. Elements outside the row are replaced by the nearest point
// of the row (duplicates keep bounds and the uniformity test valid) and never stored; that only happens in the
// first and the last windows of a row ("edge" chunks, wave-uniform).
static __device__ __forceinline__ bool sdfk_interior(unsigned L, unsigned k) {      // wave-uniform
    return (L & 31u) == 0u || (k >= 1u && 32u * k + 32u <= L);
}
static __device__ __forceinline__ float4 sdfk_win_quad(const float* __restrict__ rowp, int z, int last, bool interior) {
    if (interior) {
        const sdfk_f4u v = __builtin_nontemporal_load(reinterpret_cast<const sdfk_f4u*>(rowp + z));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    // edge window, branch-free on purpose: per-lane branches around the loads (whole quad / one value / mixed)
    // serialise the memory round trips and were measured 15 % slower for the whole kernel
    return make_float4(rowp[min(max(z, 0), last)], rowp[min(max(z + 1, 0), last)], rowp[min(max(z + 2, 0), last)],
                       rowp[min(max(z + 3, 0), last)]);
}
// the 4 points [z, z+4) of row `row` (row = r0 + dr, r0 wave-uniform) as three quads
static __device__ __forceinline__ void sdfk_rows_fetch(const SrcArray& s, const sdfk_rowgeom& g, long long r0, int dr, int z,
                                                       bool interior, float4& X, float4& Y, float4& Z) {
    const float* p = s.co + (r0 + dr) * (long long)g.L;
    const int last = (int)g.L - 1;
    X = sdfk_win_quad(p, z, last, interior);
    Y = sdfk_win_quad(p + s.stride, z, last, interior);
#ifdef SDFK_XY                                                    // two-row coordinates (z = 0 by contract): 12 B/point
    Z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
    Z = sdfk_win_quad(p + 2 * s.stride, z, last, interior);
#endif
}
// regular grid: row -> (ix, iy) with one wave-uniform division per brick; z from the third axis table
static __device__ __forceinline__ void sdfk_rows_fetch(const SrcGrid& s, const sdfk_rowgeom& g, long long r0, int dr, int z,
                                                       bool interior, float4& X, float4& Y, float4& Z) {
    const unsigned long long g0 = (unsigned long long)(g.row0 + r0);
    if (g.yrows) {                                                 // (x, y, z) = (ax0[row], ax1[point], ax2[0])
        const float x = s.ax0[g0 + (unsigned)dr], zc = s.ax2[0];
        X = make_float4(x, x, x, x);
        Y = sdfk_win_quad(s.ax1, z, (int)g.L - 1, interior);
        Z = make_float4(zc, zc, zc, zc);
        return;
    }
    unsigned long long ix = g0 / s.n1;
    unsigned iy = (unsigned)(g0 - ix * s.n1) + (unsigned)dr;
    while (iy >= s.n1) { iy -= s.n1; ++ix; }
    const float x = s.ax0[ix], y = s.ax1[iy];
    X = make_float4(x, x, x, x);
    Y = make_float4(y, y, y, y);
    Z = sdfk_win_quad(s.ax2, z, (int)g.L - 1, interior);
}
static __device__ __forceinline__ float sdfk_prev_lane(float v) {   // value of lane-1 within a row of 16 lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
// point index (in its row) of the first point of window k of local row `row`, plus e0
static __device__ __forceinline__ int sdfk_win_z(long long row, unsigned L, unsigned k, int e0, long long* flat) {
    const long long rl = row * (long long)L;
    const long long f = (((rl >> 5) + k) << 5) + e0;
    *flat = f;
    return (int)(f - rl);
}

// phase A, part 1: issue the loads of one brick (wave-uniform r0, k); 8 rows x 32 points per instruction
struct sdfk_rowregs {
    float4 X[SDFK_RLOADS], Y[SDFK_RLOADS], Z[SDFK_RLOADS];
};
template <typename SRC>
static __device__ __forceinline__ void sdfk_rows_load(const SRC& s, const sdfk_rowgeom& g, long long r0, long long rend,
                                                      unsigned k, int lane, sdfk_rowregs& r) {
    // SDFK_FLAT (flat grids: rows run along y, z = 0 everywhere): the roles of y and z are swapped from here to the
    // end of phase A — r.Y holds z (the coordinate a row shares, which must be exactly 0) and r.Z holds y (the one
    // that varies along the row) — so that "uniform", the staging and the bounding sphere work unchanged
    const bool interior = sdfk_interior(g.L, k);
#pragma unroll
    for (int t = 0; t < SDFK_RLOADS; ++t) {
        int dr = 8 * t + (lane >> 3);
        if (r0 + dr >= rend) dr = (int)(rend - 1 - r0);
        long long flat;
        const int z = sdfk_win_z(r0 + dr, g.L, k, 4 * (lane & 7), &flat);
        #ifdef SDFK_FLAT
        sdfk_rows_fetch(s, g, r0, dr, z, interior, r.X[t], r.Z[t], r.Y[t]);
#else
        sdfk_rows_fetch(s, g, r0, dr, z, interior, r.X[t], r.Y[t], r.Z[t]);
#endif
    }
}
// phase A, part 2: z and the row heads to LDS, "every row has one x and one y", bounding sphere
// (SDFK_CELLS: S = sphere of the brick's cell, with y and z swapped for SDFK_FLAT like the registers; the brick also learns
//  whether every one of its points lies inside it)
#ifdef SDFK_CELLS
#define SDFK_CELL_ARG , float4 S
#else
#define SDFK_CELL_ARG
#endif
static __device__ __forceinline__ void sdfk_rows_bounds(const sdfk_rowregs& r, int lane, sdfk_rowmeta* meta, int b SDFK_CELL_ARG) {
#ifdef SDFK_ABLATE_BOUNDS
    *reinterpret_cast<float4*>(&meta->z[b][(lane >> 3) * SDFK_RZ + 4 * (lane & 7)]) = r.Z[0];
    if (SDFK_RLOADS > 1) *reinterpret_cast<float4*>(&meta->z[b][(8 + (lane >> 3)) * SDFK_RZ + 4 * (lane & 7)]) = r.Z[SDFK_RLOADS - 1];
    if ((lane & 7) == 0) { meta->xy[b][lane >> 3] = make_float2(r.X[0].x, r.Y[0].x); meta->xy[b][SDFK_RROWS - 8 + (lane >> 3)] = make_float2(r.X[SDFK_RLOADS - 1].x, r.Y[SDFK_RLOADS - 1].x); }
    if (lane == 0) { meta->bound[b] = make_float4(0.f, 0.f, 0.f, 1.f); meta->uniform[b] = 1u; }
#ifdef SDFK_CELLS
    if (lane == 0) meta->inside[b] = S.w > 0.0f ? 1u : 0u;
#endif
    return;
#endif
    bool uni = true;
#pragma unroll
    for (int t = 0; t < SDFK_RLOADS; ++t) {
        *reinterpret_cast<float4*>(&meta->z[b][(8 * t + (lane >> 3)) * SDFK_RZ + 4 * (lane & 7)]) = r.Z[t];
        const float px = sdfk_prev_lane(r.X[t].x), py = sdfk_prev_lane(r.Y[t].x);
        const bool head = (lane & 7) == 0;
        uni = uni && r.X[t].x == r.X[t].y && r.X[t].x == r.X[t].z && r.X[t].x == r.X[t].w && r.Y[t].x == r.Y[t].y &&
              r.Y[t].x == r.Y[t].z && r.Y[t].x == r.Y[t].w && (head || (px == r.X[t].x && py == r.Y[t].x));
#ifdef SDFK_FLAT
        uni = uni && r.Y[t].x == 0.0f;                             // (M2 * 0 drops out of every root transform exactly)
#endif
        if (head) meta->xy[b][8 * t + (lane >> 3)] = make_float2(r.X[t].x, r.Y[t].x);
    }
    const bool uniform = __ballot(uni) == ~0ull;
#if defined(SDFK_SIMT) && SDFK_NSUB > 1
    // bricks of whole row segments are bounded per SUB-brick by the probe lanes (sdfk_sub_centre, from the staged
    // coordinates): the sphere around the whole brick is only needed for the others
    if (uniform) {
        if (lane == 0) {
            meta->uniform[b] = 1u;
#pragma unroll
            for (int w = 0; w < SDFK_NMASK; ++w) meta->mask[w][b] = ~0ull;
        }
        return;
    }
#endif
    const float cx = 0.5f * (sdfk_lane(r.X[0].x, 0) + sdfk_lane(r.X[SDFK_RLOADS - 1].w, 63));
    const float cy = 0.5f * (sdfk_lane(r.Y[0].x, 0) + sdfk_lane(r.Y[SDFK_RLOADS - 1].w, 63));
    const float cz = 0.5f * (sdfk_lane(r.Z[0].x, 0) + sdfk_lane(r.Z[SDFK_RLOADS - 1].w, 63));
    float d2 = 0.0f;
#ifdef SDFK_CELLS
    float d2s = 0.0f;                                // the same maximum about the CELL's centre
#endif
    if (uniform) {                                   // one x, y per row segment: |p - c|^2 = dxy^2 + max |z - cz|^2
#pragma unroll
        for (int t = 0; t < SDFK_RLOADS; ++t) {
            const float dx = r.X[t].x - cx, dy = r.Y[t].x - cy;
            const float zm = sd_rawmax(sd_rawmax(sd_abs(r.Z[t].x - cz), sd_abs(r.Z[t].y - cz)),
                                       sd_rawmax(sd_abs(r.Z[t].z - cz), sd_abs(r.Z[t].w - cz)));
            d2 = sd_rawmax(d2, sd_fma(zm, zm, sd_fma(dx, dx, dy * dy)));
#ifdef SDFK_CELLS
            const float ex = r.X[t].x - S.x, ey = r.Y[t].x - S.y;
            const float em = sd_rawmax(sd_rawmax(sd_abs(r.Z[t].x - S.z), sd_abs(r.Z[t].y - S.z)),
                                       sd_rawmax(sd_abs(r.Z[t].z - S.z), sd_abs(r.Z[t].w - S.z)));
            d2s = sd_rawmax(d2s, sd_fma(em, em, sd_fma(ex, ex, ey * ey)));
#endif
        }
    } else {
#pragma unroll
        for (int t = 0; t < SDFK_RLOADS; ++t) {
            const f2 xa = {r.X[t].x, r.X[t].y}, xb = {r.X[t].z, r.X[t].w}, ya = {r.Y[t].x, r.Y[t].y}, yb = {r.Y[t].z, r.Y[t].w};
            const f2 za = {r.Z[t].x, r.Z[t].y}, zb = {r.Z[t].z, r.Z[t].w};
            const f2 ax = xa - cx, ay = ya - cy, az = za - cz, bx = xb - cx, by = yb - cy, bz = zb - cz;
            const f2 da = sd_fma(ax, ax, sd_fma(ay, ay, az * az)), db = sd_fma(bx, bx, sd_fma(by, by, bz * bz));
            d2 = sd_rawmax(d2, sd_rawmax(sd_rawmax(da.x, da.y), sd_rawmax(db.x, db.y)));
#ifdef SDFK_CELLS
            const f2 ex = xa - S.x, ey = ya - S.y, ez = za - S.z, fx = xb - S.x, fy = yb - S.y, fz = zb - S.z;
            const f2 ea = sd_fma(ex, ex, sd_fma(ey, ey, ez * ez)), eb = sd_fma(fx, fx, sd_fma(fy, fy, fz * fz));
            d2s = sd_rawmax(d2s, sd_rawmax(sd_rawmax(ea.x, ea.y), sd_rawmax(eb.x, eb.y)));
#endif
        }
    }
    const float r2 = sdfk_wave_max(d2);
#ifdef SDFK_CELLS
    // inside: max |p - S| <= radius of S, with room for the rounding of the distances (relative 1e-5; S.w < 0: no cell)
    const float rs2 = sdfk_wave_max(d2s);
    if (lane == 0) meta->inside[b] = (S.w > 0.0f && 1.00002f * sqrtf(rs2) <= S.w) ? 1u : 0u;
#endif
    if (lane == 0) {
#ifdef SDFK_FLAT
        meta->bound[b] = make_float4(cx, cz, cy, 1.00001f * sqrtf(r2) + 1e-30f);
#else
        meta->bound[b] = make_float4(cx, cy, cz, 1.00001f * sqrtf(r2) + 1e-30f);
#endif
        meta->uniform[b] = uniform ? 1u : 0u;
#ifdef SDFK_SIMT
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w) meta->mask[w][b] = ~0ull;   // the fold lanes AND their decisions into these
#endif
    }
}
#if defined(SDFK_SIMT) && SDFK_NSUB > 1
// Probe centre `sb` of brick b whose rows each have one x and one y: the sub-brick is rows [r0, r0 + RPS) x points
// [i0, i0 + ZPS) of the window; sphere around the midpoint of its first and last point, radius = the exact maximum
// over its points (read back from the staged coordinates), so any input is bounded correctly — on a regular grid
// 4 rows x 8 points have a quarter of the brick's radius.
static __device__ __forceinline__ float4 sdfk_sub_centre(const sdfk_rowmeta* meta, unsigned b, unsigned sb) {
    constexpr int NZP = 4, NRP = SDFK_NSUB / NZP, RPS = SDFK_RROWS / NRP, ZPS = SDFK_RZ / NZP;
    static_assert(ZPS % 4 == 0 && RPS >= 1, "sub-brick shape");
    int r0 = (int)(sb / NZP) * RPS, r1 = r0 + RPS;
    const int i0 = (int)(sb % NZP) * ZPS;
    if constexpr (NRP == 2) {
        // A brick whose 16 rows come from TWO planes of the grid (1 row block in 32 at 513^3: the last rows of one plane,
        // y near its maximum, and the first rows of the next, y near its minimum) is split where x changes instead of in the
        // middle: each part lies in one plane and is bounded tightly; split in the middle, the part that holds the plane
        // change spans the whole y extent and nothing of the brick could be culled.
        const float x0 = meta->xy[b][0].x;
        if (__builtin_bit_cast(unsigned, x0) != __builtin_bit_cast(unsigned, meta->xy[b][SDFK_RROWS - 1].x)) {
            int s = 1;
            while (s < SDFK_RROWS - 1 && __builtin_bit_cast(unsigned, meta->xy[b][s].x) == __builtin_bit_cast(unsigned, x0)) ++s;
            r0 = sb / NZP ? s : 0;
            r1 = sb / NZP ? SDFK_RROWS : s;
        }
    }
    const float2 pa = meta->xy[b][r0], pe = meta->xy[b][r1 - 1];
    const float za = meta->z[b][r0 * SDFK_RZ + i0], ze = meta->z[b][(r1 - 1) * SDFK_RZ + i0 + ZPS - 1];
    const float cx = 0.5f * (pa.x + pe.x), cy = 0.5f * (pa.y + pe.y), cz = 0.5f * (za + ze);
    float r2 = 0.0f;
#pragma unroll 1
    for (int rr = r0; rr < r1; ++rr) {
        const int r = rr - r0;
        const float2 p = meta->xy[b][r0 + r];
        const float dx = p.x - cx, dy = p.y - cy;
        float zm = 0.0f;
#pragma unroll
        for (int i = 0; i < ZPS; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(&meta->z[b][(r0 + r) * SDFK_RZ + i0 + i]);
            zm = sd_rawmax(sd_rawmax(zm, sd_abs(q.x - cz)), sd_rawmax(sd_abs(q.y - cz), sd_rawmax(sd_abs(q.z - cz), sd_abs(q.w - cz))));
        }
        r2 = sd_rawmax(r2, sd_fma(zm, zm, sd_fma(dx, dx, dy * dy)));
    }
    const float rho = 1.00001f * sqrtf(r2) + 1e-30f;
#ifdef SDFK_FLAT
    return make_float4(cx, cz, cy, rho);                         // (the staged coordinate is y, xy holds (x, z = 0))
#else
    return make_float4(cx, cy, cz, rho);
#endif
}
#endif

#if defined(SDFK_CHAIN) && !defined(SDFK_CELLS)
// Chain mode, one WAVE per brick: which children run on this brick, from the leaf values at its centre. A hard min (max:
// signs flipped, e_k = +-d_k) is exact, associative and commutative, so the ORDER of the chain does not matter for a
// decision: child k can be dropped wherever some other child j stays below it on the whole brick, and with j = the child
// that is smallest at the centre that is guaranteed by e_k - m >= thr, m = min_j e_j, thr = K rho + margins with
// K = the largest Lipschitz sum of the chain (any pair). No prefix minimum, no "restart" level: one reduction for m, then
// 64 children per step — compare, ballot, and the survivors written as a LIST in index order, so that the evaluation
// never looks at a child it skips and still combines the others in the chain's order. (Tighter than the sequential rule,
// which compares a child with its predecessors only.)
static __device__ __forceinline__ void sdfk_chain_fold(sdfk_rowmeta* meta, int b, int lane) {
    const float4 cc = meta->bound[b];
    const float rho = cc.w, cmag = 1e-6f * (fabsf(cc.x) + fabsf(cc.y) + fabsf(cc.z) + rho);
    constexpr int NW = (SDFK_NLEAF + 63) / 64;
    float m = 3.0e38f;
#pragma unroll 4
    for (int j = 0; j < NW; ++j) {
        const int k = 64 * j + lane;
        if (k < SDFK_NLEAF) m = fminf(m, SDFK_CHAIN_SGN * meta->leafval[k * SDFK_NCEN + b]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fminf(m, __shfl_xor(m, o));
    const float thr0 = 1.0001f * SDFK_CHAIN_KMAX * rho + SDFK_CHAIN_KMAX * cmag + 1e-6f * (1.0f + fabsf(m));
    int n_out = 0;
#pragma unroll 1
    for (int j = 0; j < NW; ++j) {
        const int k = 64 * j + lane;
        const bool in = k < SDFK_NLEAF;
        const float e = in ? SDFK_CHAIN_SGN * meta->leafval[k * SDFK_NCEN + b] : 3.0e38f;
        const bool run = in && !(e - m >= thr0 + 1e-6f * fabsf(e));
        const unsigned long long bits = __ballot(run);
        if (run) meta->alist[b][n_out + __builtin_popcountll(bits & ((1ull << lane) - 1ull))] = (unsigned short)k;
        n_out += __builtin_popcountll(bits);
    }
    if (lane == 0) meta->nalive[b] = (unsigned)n_out;
}
#endif

#ifdef SDFK_CELLS
// rows of row block rb plus the cell of brick (rb, window c) and that cell's sphere (wave-uniform; scalar loads)
static __device__ __forceinline__ void sdfk_brick_cell(const sdfk_rowgeom& g, const sdfk_cells& cl, unsigned rb, unsigned c,
                                                       long long& r0, long long& rend, uint2& span, float4& S) {
    unsigned slot, blk;
    sdfk_block_rows(g, rb, r0, rend, &slot, &blk);
    const unsigned sx = slot == 0xffffffffu ? 0u : cl.lv.xoff + slot;
    const unsigned cell = ((sx >> cl.lv.lx) * cl.lv.ncy + (blk >> cl.lv.ly)) * cl.lv.ncz + (c >> cl.lv.lz);
    S = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    span = make_uint2(0u, SDFK_ALL_ALIVE);
    // (the launch is padded with bricks beyond the last one: their row block does not exist, and neither does a cell for it)
    if (cl.enabled && r0 < rend && cell < cl.ncells) {
        const uint2 sp = cl.span[cell];
        span = make_uint2(__builtin_amdgcn_readfirstlane(sp.x), __builtin_amdgcn_readfirstlane(sp.y));
        const float4 t = cl.sph[cell];
#ifdef SDFK_FLAT
        S = make_float4(t.x, t.z, t.y, t.w);                   // (the registers of phase A hold (x, z, y))
#else
        S = t;
#endif
    }
}
#endif
#if defined(SDFK_CHAIN) && defined(SDFK_CELLS)
// The same decision from the CELL's candidate list: one member per lane, evaluated at the brick's centre on the spot (no
// table of leaf values in LDS) — the first 64 stay in a register, longer lists are evaluated a second time for the
// comparison. A brick that is not inside its cell's sphere, or whose cell has no list, takes every member.
// `sp` (the cell's span) and `first` (this lane's entry of the first batch) were loaded BEFORE the coordinates were waited
// for, so the only memory round trip left between the bounds and the decision is the members' parameters.
static __device__ __forceinline__ void sdfk_chain_fold_cells(sdfk_rowmeta* meta, int b, int lane, const sdfk_cells& cl, uint2 sp,
                                                             unsigned first, const float* __restrict__ PRM,
                                                             const float* __restrict__ TAB) {
    const float4 cc = meta->bound[b];
    const V3T<float> ctr = {cc.x, cc.y, cc.z};
    const float rho = cc.w, cmag = 1e-6f * (fabsf(cc.x) + fabsf(cc.y) + fabsf(cc.z) + rho);
    unsigned off = 0u, cnt = SDFK_NLEAF;
    bool listed = false;
    if (cl.enabled && __builtin_amdgcn_readfirstlane(meta->inside[b]) != 0u && sp.y != SDFK_ALL_ALIVE) {
        off = sp.x; cnt = sp.y; listed = true;
    }
    float m = 3.0e38f, e0 = 3.0e38f;
    unsigned r0 = 0u;
#pragma unroll 1
    for (unsigned j0 = 0u; j0 < cnt; j0 += 64u) {
        const unsigned j = j0 + (unsigned)lane;
        const bool in = j < cnt;
        unsigned rec = 0u;
        if (in) rec = listed ? (j0 == 0u ? first : cl.cand[off + j]) : sdfk_member_record(j);
        float e = 3.0e38f;
        if (in) e = SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(rec, ctr, PRM, TAB);
        if (j0 == 0u) { e0 = e; r0 = rec; }
        m = fminf(m, e);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fminf(m, __shfl_xor(m, o));
    const float thr0 = 1.0001f * SDFK_CHAIN_KMAX * rho + SDFK_CHAIN_KMAX * cmag + 1e-6f * (1.0f + fabsf(m));
    unsigned n_out = 0u;
#pragma unroll 1
    for (unsigned j0 = 0u; j0 < cnt; j0 += 64u) {
        const unsigned j = j0 + (unsigned)lane;
        const bool in = j < cnt;
        unsigned rec = r0;
        float e = e0;
        if (j0 != 0u) {
            rec = in ? (listed ? cl.cand[off + j] : sdfk_member_record(j)) : 0u;
            e = 3.0e38f;
            if (in) e = SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(rec, ctr, PRM, TAB);
        }
        const bool run = in && !(e - m >= thr0 + 1e-6f * fabsf(e));
        const unsigned long long bits = __ballot(run);
        const unsigned pos = n_out + (unsigned)__builtin_popcountll(bits & ((1ull << lane) - 1ull));
        if (run && pos < SDFK_ALIST_CAP) meta->alist[b][pos] = rec;
        n_out += (unsigned)__builtin_popcountll(bits);
    }
    // more survivors than the list in LDS holds (dense scenes): the brick runs its cell's whole candidate list straight from
    // memory — a superset in the chain's order — and every member only where there is no such list
    if (lane == 0) {
        meta->nalive[b] = n_out <= SDFK_ALIST_CAP ? n_out : (listed ? SDFK_LIST_ALIVE : SDFK_ALL_ALIVE);
        meta->aspan[b] = make_uint2(off, cnt);
    }
}
#endif

// phases A and B of the tile of this workgroup; returns this wave's first brick (row block, window)
#ifdef SDFK_CELLS
#define SDFK_CELLS_PARAM , const sdfk_cells& cl
#define SDFK_CELLS_PASS , cl
#else
#define SDFK_CELLS_PARAM
#define SDFK_CELLS_PASS
#endif
template <typename SRC>
static __device__ __forceinline__ void sdfk_rows_prepare(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                         const SRC& s, const sdfk_rowgeom& g, sdfk_rowmeta* meta,
                                                         unsigned tile, unsigned& rb0, unsigned& c0 SDFK_CELLS_PARAM) {
    const int lane = sdfk_tx() & 63, wave = __builtin_amdgcn_readfirstlane(sdfk_tx() >> 6);
    const unsigned q0 = tile * SDFK_RNBRICK + wave * SDFK_RWBRICKS;            // first brick of this wave
    rb0 = sdfk_udiv(q0, g.nchunk, g.inv_nchunk);                // (back in an SGPR: what is derived from it stays scalar)
    c0 = q0 - rb0 * g.nchunk;
    unsigned rb = rb0, c = c0;
#if SDFK_RWBRICKS <= 2 && !defined(SDFK_NOHOIST)
    // two bricks per wave: all of the wave's loads are in flight before the first one is used (-0.8 % together with
    // two-wave workgroups on the north-star grid; with four bricks per wave the registers this takes cost more)
    sdfk_rowregs hregs[SDFK_RWBRICKS];
    bool live[SDFK_RWBRICKS];
#ifdef SDFK_CELLS
    uint2 cspan[SDFK_RWBRICKS];
    unsigned cfirst[SDFK_RWBRICKS];
    float4 csph[SDFK_RWBRICKS];
#endif
#pragma unroll
    for (int j = 0; j < SDFK_RWBRICKS; ++j) {
        long long r0, rend;
#ifdef SDFK_CELLS
        sdfk_brick_cell(g, cl, rb, c, r0, rend, cspan[j], csph[j]);
        cfirst[j] = 0u;                                          // the lane's entry of the list's first batch: in flight with the coordinates
        if (cspan[j].y != SDFK_ALL_ALIVE && (unsigned)lane < cspan[j].y) cfirst[j] = cl.cand[cspan[j].x + (unsigned)lane];
#else
        sdfk_block_rows(g, rb, r0, rend);
#endif
        live[j] = q0 + j < g.nbricks && r0 < rend;               // (blocks of the last plane beyond the slab: nothing)
        if (live[j]) sdfk_rows_load(s, g, r0, rend, c, lane, hregs[j]);
        if (++c == g.nchunk) { c = 0; ++rb; }
    }
#pragma unroll
    for (int j = 0; j < SDFK_RWBRICKS; ++j)
#ifdef SDFK_CELLS
        if (live[j]) sdfk_rows_bounds(hregs[j], lane, meta, wave * SDFK_RWBRICKS + j, csph[j]);
#else
        if (live[j]) sdfk_rows_bounds(hregs[j], lane, meta, wave * SDFK_RWBRICKS + j);
#endif
        else if (lane == 0) { meta->bound[wave * SDFK_RWBRICKS + j] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); meta->uniform[wave * SDFK_RWBRICKS + j] = 0u; }   // (masks: never read)
#else
#ifdef SDFK_CELLS
    bool live[SDFK_RWBRICKS];
    uint2 cspan[SDFK_RWBRICKS];
    unsigned cfirst[SDFK_RWBRICKS];
#endif
#pragma unroll
    for (int j = 0; j < SDFK_RWBRICKS; ++j) {
        long long r0, rend;
#ifdef SDFK_CELLS
        float4 csph;
        sdfk_brick_cell(g, cl, rb, c, r0, rend, cspan[j], csph);
        cfirst[j] = 0u;
        if (cspan[j].y != SDFK_ALL_ALIVE && (unsigned)lane < cspan[j].y) cfirst[j] = cl.cand[cspan[j].x + (unsigned)lane];
        live[j] = q0 + j < g.nbricks && r0 < rend;
#else
        sdfk_block_rows(g, rb, r0, rend);
#endif
#ifdef SDFK_ABLATE_EDGE
        if (q0 + j < g.nbricks && r0 < rend && sdfk_interior(g.L, c)) {
#else
        if (q0 + j < g.nbricks && r0 < rend) {
#endif
            sdfk_rowregs regs;
            sdfk_rows_load(s, g, r0, rend, c, lane, regs);
#ifdef SDFK_CELLS
            sdfk_rows_bounds(regs, lane, meta, wave * SDFK_RWBRICKS + j, csph);
#else
            sdfk_rows_bounds(regs, lane, meta, wave * SDFK_RWBRICKS + j);
#endif
        } else if (lane == 0) {                                   // a dead brick still meets a probe lane: harmless values
            meta->bound[wave * SDFK_RWBRICKS + j] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            meta->uniform[wave * SDFK_RWBRICKS + j] = 0u;
        }
        if (++c == g.nchunk) { c = 0; ++rb; }
    }
#endif
    __syncthreads();
#if defined(SDFK_CHAIN) && defined(SDFK_CELLS)
    // every wave folds ITS bricks from their cells' lists: nothing crosses waves any more (no probe centres, no table of
    // leaf values, two barriers fewer than the path below)
#pragma unroll
    for (int j = 0; j < SDFK_RWBRICKS; ++j) {
        const int b = wave * SDFK_RWBRICKS + j;
#ifdef SDFK_ABLATE_PROBE
        if (lane == 0) { meta->alist[b][0] = sdfk_member_record(0u); meta->nalive[b] = live[j] ? 1u : 0u; }
#else
        if (live[j]) sdfk_chain_fold_cells(meta, b, lane, cl, cspan[j], cfirst[j], PRM, TAB);
        else if (lane == 0) meta->nalive[b] = 0u;
#endif
    }
    __syncthreads();
    return;
#endif
#if defined(SDFK_SIMT) && !defined(SDFK_CELLS)
    // lane-parallel probe: probe centres -> every leaf at every centre (all lanes) -> one fold lane per centre ANDs its
    // skip decisions into the brick's mask (a subtree is skipped only if every sub-brick allows it)
    if (sdfk_tx() < SDFK_NCEN) {
        const unsigned b = sdfk_tx() / SDFK_NSUB, sb = sdfk_tx() % SDFK_NSUB;
        float4 cc = meta->bound[b];
#if SDFK_NSUB > 1
        if (meta->uniform[b]) cc = sdfk_sub_centre(meta, b, sb);
#else
        (void)sb;
#endif
        meta->cen[sdfk_tx()] = cc;
    }
    __syncthreads();
#ifndef SDFK_ABLATE_PROBE
    sdfk_probe_leaves(meta->cen, meta->leafval, PRM, TAB);
    __syncthreads();
#ifdef SDFK_CHAIN
#pragma unroll 1
    for (int j = 0; j < SDFK_RWBRICKS; ++j)
        if (q0 + j < g.nbricks) sdfk_chain_fold(meta, wave * SDFK_RWBRICKS + j, lane);
    __syncthreads();
    return;
#else
    if (sdfk_tx() < SDFK_NCEN && tile * SDFK_RNBRICK + sdfk_tx() / SDFK_NSUB < g.nbricks) {
        const float4 cc = meta->cen[sdfk_tx()];
        V3T<float> ctr = {cc.x, cc.y, cc.z};
        unsigned long long mk[SDFK_NMASK];
        sdfk_probe_fold(meta->leafval, sdfk_tx(), ctr, cc.w, PRM, mk);
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w)
            __hip_atomic_fetch_and(&meta->mask[w][sdfk_tx() / SDFK_NSUB], mk[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#endif
#else
    if (sdfk_tx() < SDFK_RNBRICK) {
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w) meta->mask[w][sdfk_tx()] = 0ull;
    }
#endif
    __syncthreads();
    return;
#endif
    if (sdfk_tx() < SDFK_RNBRICK && tile * SDFK_RNBRICK + sdfk_tx() < g.nbricks) {
        const float4 bb = meta->bound[sdfk_tx()];
        V3T<float> ctr = {bb.x, bb.y, bb.z};
        unsigned long long mk[SDFK_NMASK];
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w) mk[w] = 0ull;
#if !defined(SDFK_ABLATE_PROBE) && !defined(SDFK_CHAIN)
        sdfk_probe_r(ctr, bb.w, PRM, TAB, mk);
#endif
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w) meta->mask[w][sdfk_tx()] = mk[w];
    }
    __syncthreads();
}

template <typename SRC>
static __device__ __forceinline__ void sdfk_rows_kernel(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                        const SRC& s, const sdfk_rowgeom& g, float* __restrict__ out,
                                                        unsigned* __restrict__ flags, unsigned thr SDFK_CELLS_PARAM) {
    __shared__ __attribute__((aligned(16))) sdfk_rowmeta meta;
#ifdef SDFK_LDSPAD
    if (g.L == 0xffffffffu) meta.pad[sdfk_tx()] = 1.0f;
#endif
    const int lane = sdfk_tx() & 63, wave = __builtin_amdgcn_readfirstlane(sdfk_tx() >> 6);
    const int lr = lane / SDFK_RLPR;
    const int zq = (lane % SDFK_RLPR) * (2 * SDFK_NP);
#ifndef SDFK_RTILES
#define SDFK_RTILES 1
#endif
#ifndef SDFK_XGROUP
#ifdef SDFK_FLAT                         // flat grids (rows of 16385 points: 513 windows): short runs — the 50-member union at
#define SDFK_XGROUP 4                    // 16385^2 0.833 -> 0.805 ms against runs of 8, 0.840 with 16 (profiles/r04_w4_cfg4.txt)
#else
#define SDFK_XGROUP (SDFK_RNBRICK <= 4 ? 16 : (SDFK_RNBRICK <= 8 ? 8 : 4))    // runs of about 64 bricks: two row blocks at 1025
#endif
#endif
#pragma unroll 1
  for (unsigned tt = 0; tt < SDFK_RTILES; ++tt) {
    // Workgroups go round-robin to the 8 XCDs: an XCD takes runs of SDFK_XGROUP consecutive tiles (the grid is a
    // multiple of 8 * SDFK_XGROUP) instead of every eighth tile, while all XCDs stay in the same neighbourhood of the
    // arrays. Measured, not derived: runs of 1 (plain round-robin) are 0.7 to 3 % slower on the north-star grid
    // depending on the box and 18 % on the MEDIAN at 513^3; runs aligned to whole row blocks (33, 66) are no better than
    // runs of 8; one contiguous eighth of the grid per XCD costs 7 % (eight far-apart streams). HBM traffic is the same
    // in all of them (DESIGN 9.1).
    const unsigned xg_t = sdfk_bx() / 8;
    const unsigned tile = (((xg_t / SDFK_XGROUP) * 8 + sdfk_bx() % 8) * SDFK_XGROUP + xg_t % SDFK_XGROUP) * SDFK_RTILES + tt;
    if (tile * SDFK_RNBRICK >= g.nbricks) break;
    if (tt) __syncthreads();
    unsigned rb, c;
    sdfk_rows_prepare(PRM, TAB, s, g, &meta, tile, rb, c SDFK_CELLS_PASS);
    const unsigned q0 = tile * SDFK_RNBRICK + wave * SDFK_RWBRICKS;
#pragma unroll 1
    for (int j = 0; j < SDFK_RWBRICKS; ++j) {
        if (q0 + j >= g.nbricks) break;
        const int b = wave * SDFK_RWBRICKS + j;
        const unsigned k = c;
        long long r0, rend;
        sdfk_block_rows(g, rb, r0, rend);
        if (++c == g.nchunk) { c = 0; ++rb; }
        if (r0 >= rend) continue;
        unsigned mw[2 * SDFK_NMASK];                             // the brick's skip bits as scalar words
#pragma unroll
        for (int w = 0; w < SDFK_NMASK; ++w) {
            const unsigned long long m = meta.mask[w][b];
            mw[2 * w] = __builtin_amdgcn_readfirstlane((unsigned)m);
            mw[2 * w + 1] = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        }
        const bool uniform = __builtin_amdgcn_readfirstlane(meta.uniform[b]) != 0u;
        const bool interior = sdfk_interior(g.L, k);
#ifdef SDFK_ABLATE_EDGE
        if (!interior) continue;
#endif
        const bool live_row = r0 + lr < rend;
        const int dr = live_row ? lr : (int)(rend - 1 - r0);
        long long f;                                              // flat index of this lane's first point
        const int z = sdfk_win_z(r0 + dr, g.L, k, zq, &f);        // its index in the row (edge chunks: may lie outside)
        const int last = (int)g.L - 1;
        V3P P[SDFK_NP];
        f2 res[SDFK_NP];
        // the staged coordinate: z (SDFK_FLAT: y, see sdfk_rows_load)
#ifdef SDFK_FLAT
        SDFK_EACH P[q].y = *reinterpret_cast<const f2*>(&meta.z[b][lr * SDFK_RZ + zq + 2 * q]);
#else
        SDFK_EACH P[q].z = *reinterpret_cast<const f2*>(&meta.z[b][lr * SDFK_RZ + zq + 2 * q]);
#endif
#ifdef SDFK_ABLATE_EVAL
        if (true) {
#ifdef SDFK_FLAT
            SDFK_EACH res[q] = P[q].y + __builtin_bit_cast(float, mw[0] ^ mw[1] ^ mw[2] ^ mw[3]);
#else
            SDFK_EACH res[q] = P[q].z + __builtin_bit_cast(float, mw[0] ^ mw[1] ^ mw[2] ^ mw[3]);
#endif
        } else
#endif
        if (uniform) {
            const float2 xy = meta.xy[b][lr];
#ifdef SDFK_FLAT
            SDFK_EACH { P[q].x = sp<f2>(xy.x); P[q].z = sp<f2>(xy.y); }      // (x, 0)
#else
            SDFK_EACH { P[q].x = sp<f2>(xy.x); P[q].y = sp<f2>(xy.y); }
#endif
#ifndef SDFK_CHAIN
            sdfk_rows_culled<true>(xy.x, xy.y, P, mw, PRM, TAB, res);
#endif
        } else {                                                  // the other coordinates of every point again (L2-resident)
#pragma unroll
            for (int q = 0; q < SDFK_NP; q += 2) {
                float4 X, Y, Z;
                sdfk_rows_fetch(s, g, r0, dr, z + 2 * q, interior, X, Y, Z);
                P[q].x = {X.x, X.y};
                P[q + 1].x = {X.z, X.w};
#ifdef SDFK_FLAT
                P[q].z = {Z.x, Z.y};
                P[q + 1].z = {Z.z, Z.w};
#else
                P[q].y = {Y.x, Y.y};
                P[q + 1].y = {Y.z, Y.w};
#endif
            }
#ifndef SDFK_CHAIN
            sdfk_rows_culled<false>(0.0f, 0.0f, P, mw, PRM, TAB, res);
#endif
        }
#ifdef SDFK_CHAIN
        {   // The children on the brick's list, in order; the first one starts the accumulator. Looked up one by one — kind of
            // the child, base of its parameters, then the parameters: three dependent scalar loads per child — the evaluation
            // was bound by memory latency (4200 cycles per child at 1000 children: 3.5 of 4.5 ms). So the wave first GATHERS
            // the next SDFK_CHUNK children lane-parallel — a lane per (child, parameter) — into LDS, and then evaluates them
            // from there: the latency is paid once per chunk.
#ifdef SDFK_CELLS
            // (more survivors than the list holds — nalive = ALL —: every member, in order, straight from its index)
            const unsigned nal = __builtin_amdgcn_readfirstlane(meta.nalive[b]);
            const bool all_alive = nal == SDFK_ALL_ALIVE, list_alive = nal == SDFK_LIST_ALIVE;
            const unsigned aoff = __builtin_amdgcn_readfirstlane(meta.aspan[b].x);
            const unsigned cnt = all_alive ? (unsigned)SDFK_NLEAF : (list_alive ? __builtin_amdgcn_readfirstlane(meta.aspan[b].y) : nal);
#define SDFK_AREC(i) (all_alive ? sdfk_member_record(i) : (list_alive ? cl.cand[aoff + (i)] : meta.alist[b][i]))
#else
            const unsigned cnt = __builtin_amdgcn_readfirstlane(meta.nalive[b]);
#define SDFK_AREC(i) sdfk_member_record((unsigned)meta.alist[b][i])
#endif
            f2 acc[SDFK_NP];
#if SDFK_CHAIN_STAGED
            if (cnt <= SDFK_STAGE_MIN)
#endif
            {
                // one or two survivors (the 50-child flat union: 1.06 per brick): staging costs more than it hides, and
                // parameters in SGPRs beat parameters read back from LDS — straight from the table
#pragma unroll 1
                for (unsigned i = 0; i < cnt; ++i) {
                    const unsigned rec = __builtin_amdgcn_readfirstlane(SDFK_AREC(i));
                    f2 val[SDFK_NP];
                    SDFK_EACH val[q] = sdfk_leaf_rec<f2>(rec, P[q], PRM, TAB);
                    if (i == 0u) { SDFK_EACH acc[q] = val[q]; }
                    else { SDFK_EACH acc[q] = SDFK_CHAIN_CMB(acc[q], val[q], PRM); }
                }
            }
#if SDFK_CHAIN_STAGED
            else
#pragma unroll 1
            for (unsigned c0 = 0; c0 < cnt; c0 += SDFK_CHUNK) {
                const unsigned nc = cnt - c0 < SDFK_CHUNK ? cnt - c0 : SDFK_CHUNK;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the previous chunk has been read)
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (unsigned it = (unsigned)lane; it < nc * SDFK_NPLMAX; it += 64u) {
                    const unsigned ch = it / SDFK_NPLMAX, w = it - ch * SDFK_NPLMAX;
                    const unsigned rec = SDFK_AREC(c0 + ch);
                    const unsigned at = (rec & 0xffffffu) + w;
                    meta.cprm[b][ch][w] = PRM[at < SDFK_NPARAMS ? at : SDFK_NPARAMS - 1u];
                    if (w == 0u) meta.cgrp[b][ch] = rec >> 24;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
                for (unsigned i = 0; i < nc; ++i) {
                    const unsigned grp = __builtin_amdgcn_readfirstlane(meta.cgrp[b][i]);
                    const float* cp = &meta.cprm[b][i][0];
                    f2 val[SDFK_NP];
                    SDFK_EACH val[q] = sdfk_leaf_at<f2>(grp, P[q], cp, TAB);
                    if (c0 + i == 0u) { SDFK_EACH acc[q] = val[q]; }
                    else { SDFK_EACH acc[q] = SDFK_CHAIN_CMB(acc[q], val[q], PRM); }
                }
            }
#endif
            SDFK_EACH res[q] = sdfk_chain_tail<f2>(acc[q], P[q], PRM, TAB);
#if defined(SDFK_DEBUG_NALIVE) && defined(SDFK_CELLS)
            // test aid: instead of the field, how the brick ran — survivors on its list; 100000 + entries when it ran its cell's
            // whole list (more survivors than SDFK_ALIST_CAP); -1 when it ran every member
            SDFK_EACH res[q] = sp<f2>(all_alive ? -1.0f : (list_alive ? 100000.0f + (float)cnt : (float)cnt));
#endif
        }
#endif
#ifdef SDFK_FLAGS                                                  // (the flag-writing build) one bit per point instead of the field
        {
            // BRICK-TILED layout: the word of (row r, window k) is flags[(brick * 16 + r % 16)], so the 64 lanes of the
            // wave — lane = 4 * row + octet of the window — write the brick's 64 bytes as ONE contiguous store, and every
            // (row, window) slot belongs to one brick, also where two rows share a flat window: no atomics (a linear bit
            // string needs an atomic OR in the edge windows and 16 partial-line stores per brick). The compaction
            // (sdfk_fieldops.inc) walks the slots in row order. (Measured at 1025^3: the kernel takes as long as when it
            // writes the field, 2.9 ms — it is not bound by its stores; the saving is the field that is never re-read.)
            unsigned bits = 0u;
            SDFK_EACH bits |= ((sdfk_sel_key(res[q].x) <= thr ? 1u : 0u) | (sdfk_sel_key(res[q].y) <= thr ? 2u : 0u)) << (2 * q);
            if (!interior) {                                       // an edge window: only the points of this row
                unsigned ok = 0u;
                SDFK_EACH {
                    if (z + 2 * q >= 0 && z + 2 * q <= last) ok |= 1u << (2 * q);
                    if (z + 2 * q + 1 >= 0 && z + 2 * q + 1 <= last) ok |= 2u << (2 * q);
                }
                bits &= ok;
            }
            if (!live_row) bits = 0u;                              // rows past the end of the last block: empty slots
            reinterpret_cast<unsigned char*>(flags)[(unsigned long long)(q0 + j) * 64ull + (unsigned)lane] = (unsigned char)bits;
        }
#else
        if (live_row) {
            float* po = out + f;
            if (interior) {
#pragma unroll
                for (int q = 0; q < SDFK_NP; q += 2) {
                    sdfk_f4u v = {res[q].x, res[q].y, res[q + 1].x, res[q + 1].y};
                    __builtin_nontemporal_store(v, reinterpret_cast<sdfk_f4u*>(po + 2 * q));
                }
            } else {
                SDFK_EACH {
                    if (z + 2 * q >= 0 && z + 2 * q <= last) po[2 * q] = res[q].x;
                    if (z + 2 * q + 1 >= 0 && z + 2 * q + 1 <= last) po[2 * q + 1] = res[q].y;
                }
            }
        }
#endif
    }
  }
}
#ifdef SDFK_WPE
#define SDFK_ROWS_ATTR __attribute__((amdgpu_waves_per_eu(SDFK_WPE, SDFK_WPE)))
#else
#define SDFK_ROWS_ATTR
#endif
#ifdef SDFK_CELLS
#define SDFK_CELLS_KARG , sdfk_cells cl
// ---- the pre-pass: one WAVE per cell ------------------------------------------------------------------------------
struct sdfk_cellpass {
    sdfk_celllevel lv, parent;                  // parent.ncx == 0: no coarser level — the candidates are all members
    float4* __restrict__ sph;                   // this level's cells: sphere ...
    uint2* __restrict__ span;                   // ... and list
    const float4* __restrict__ psph;            // the parent level's
    const uint2* __restrict__ pspan;
    unsigned* __restrict__ cand;                // the pool (records: sdfk_member_record)
    unsigned* __restrict__ head;                // this level's SDFK_POOL_SHARDS allocation heads, 64 bytes apart
    unsigned base, shard_cap;                   // shard h owns entries [base + h * shard_cap, base + (h + 1) * shard_cap)
    unsigned ncells, pad0;
    float inflate, pad;                         // factor on the radius the LIST is computed for (coarse level: > 1, see below)
};
// point z of local row `row`
static __device__ __forceinline__ float3 sdfk_cell_point(const SrcArray& s, const sdfk_rowgeom& g, long long row, unsigned z) {
    const float* p = s.co + row * (long long)g.L + z;
#ifdef SDFK_XY
    return make_float3(p[0], p[s.stride], 0.0f);
#else
    return make_float3(p[0], p[s.stride], p[2 * s.stride]);
#endif
}
static __device__ __forceinline__ float3 sdfk_cell_point(const SrcGrid& s, const sdfk_rowgeom& g, long long row, unsigned z) {
    const unsigned long long g0 = (unsigned long long)(g.row0 + row);
    if (g.yrows) return make_float3(s.ax0[g0], s.ax1[z], s.ax2[0]);
    const unsigned long long ix = g0 / s.n1;
    return make_float3(s.ax0[ix], s.ax1[(unsigned)(g0 - ix * s.n1)], s.ax2[z]);
}
// rows [r0, rend) and blocks of plane slot sl of a level (slot 0 = the leading partial plane when lv.xoff > 0)
static __device__ __forceinline__ bool sdfk_slot_rows(const sdfk_rowgeom& g, const sdfk_celllevel& lv, unsigned sl, long long& r0,
                                                      long long& rend) {
    if (lv.xoff > 0u && sl == 0u) { r0 = 0; rend = g.seg0; return g.seg0 > 0u; }
    if (sl < lv.xoff) return false;
    r0 = (long long)g.seg0 + (long long)(sl - lv.xoff) * g.prow;
    rend = r0 + g.prow < g.R ? r0 + g.prow : g.R;
    return r0 < g.R;
}
template <typename SRC>
static __device__ __forceinline__ void sdfk_cells_kernel(const float* __restrict__ PRM, const float* __restrict__ TAB, const SRC& s,
                                                         const sdfk_rowgeom& g, const sdfk_cellpass& cp) {
    const int lane = sdfk_tx() & 63;
    const unsigned cell = __builtin_amdgcn_readfirstlane(sdfk_bx() * 4u + (sdfk_tx() >> 6));
    if (cell >= cp.ncells) return;
    const sdfk_celllevel lv = cp.lv;
    const unsigned cz = cell % lv.ncz, t0 = cell / lv.ncz, cy = t0 % lv.ncy, cx = t0 / lv.ncy;
    // what the cell covers in index space: slots [sA, sB], blocks [b0, b1], points [zlo, zhi] of a row
    const unsigned sA = cx << lv.lx;
    unsigned sB = ((cx + 1u) << lv.lx) - 1u;
    long long ra0, ra1, rb0, rb1;
    bool some = sdfk_slot_rows(g, lv, sA, ra0, ra1);
    if (lv.xoff > 0u && cx == 0u) sB = 0u;
    while (some && sB > sA && !sdfk_slot_rows(g, lv, sB, rb0, rb1)) --sB;
    if (sB == sA) { rb0 = ra0; rb1 = ra1; }
    const long long lo = (long long)(cy << lv.ly) * SDFK_RROWS, hi = (long long)((cy + 1u) << lv.ly) * SDFK_RROWS;
    some = some && ra0 + lo < ra1;
    const unsigned w0 = cz << lv.lz, w1 = ((cz + 1u) << lv.lz) < g.nchunk ? ((cz + 1u) << lv.lz) - 1u : g.nchunk - 1u;
    some = some && w0 < g.nchunk;
    if (!some) {                                                // no brick maps here
        if (lane == 0) { cp.sph[cell] = make_float4(0.0f, 0.0f, 0.0f, -1.0f); cp.span[cell] = make_uint2(0u, 0u); }
        return;
    }
    // windows are aligned in the FLAT array: window k of a row starts up to 31 points before point 32 k of that row
    const unsigned zlo = 32u * w0 > 31u ? 32u * w0 - 31u : 0u;
    const unsigned zhi = 32u * w1 + 31u < g.L ? 32u * w1 + 31u : g.L - 1u;
    // the eight corner points (a lattice cell is their convex hull): lanes 0..7
    const long long rowsel[4] = {ra0 + lo, (ra0 + hi < ra1 ? ra0 + hi : ra1) - 1, (rb0 + lo < rb1 ? rb0 + lo : rb1 - 1),
                                 (rb0 + hi < rb1 ? rb0 + hi : rb1) - 1};
    const float3 pt = sdfk_cell_point(s, g, rowsel[(lane >> 1) & 3], (lane & 1) ? zhi : zlo);
    float mnx = pt.x, mxx = pt.x, mny = pt.y, mxy = pt.y, mnz = pt.z, mxz = pt.z;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
        mnz = fminf(mnz, __shfl_xor(mnz, o)); mxz = fmaxf(mxz, __shfl_xor(mxz, o));
    }
    const V3T<float> ctr = {0.5f * (mnx + mxx), 0.5f * (mny + mxy), 0.5f * (mnz + mxz)};
    const float dx = pt.x - ctr.x, dy = pt.y - ctr.y, dz = pt.z - ctr.z;
    float r2 = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
    r2 = __shfl(r2, 0);
    const float cxs = __shfl(ctr.x, 0), cys = __shfl(ctr.y, 0), czs = __shfl(ctr.z, 0);
    const V3T<float> c = {cxs, cys, czs};
    // the radius the cell ANSWERS for: bricks test their points against it. A coarse level answers for more than its own
    // sphere (cp.inflate): the circumsphere of a fine cell can reach beyond the circumsphere of the coarse cell it lies in
    // (fine box at a corner, boxes of different proportions), and a fine cell may only start from its parent's list if its
    // whole sphere lies inside what that list answers for.
    const float rho = (1.0001f * sqrtf(r2) + 1e-30f) * cp.inflate;
    if (lane == 0) cp.sph[cell] = make_float4(c.x, c.y, c.z, rho);
    // candidates: the parent's list if this cell's sphere lies inside the parent's answer, else every member
    unsigned off = 0u, cnt = SDFK_NLEAF;
    bool listed = false;
    if (cp.parent.ncx != 0u) {
        unsigned pcx = 0u;
        if (!(lv.xoff > 0u && cx == 0u)) pcx = (cp.parent.xoff + (sA - lv.xoff)) >> cp.parent.lx;
        const unsigned pcell = (pcx * cp.parent.ncy + (cy >> (cp.parent.ly - lv.ly))) * cp.parent.ncz + (cz >> (cp.parent.lz - lv.lz));
        const float4 ps = cp.psph[pcell];
        const uint2 sp = cp.pspan[pcell];
        const float ex = c.x - ps.x, ey = c.y - ps.y, ez = c.z - ps.z;
        if (ps.w > 0.0f && sp.y != SDFK_ALL_ALIVE && 1.00002f * (sqrtf(ex * ex + ey * ey + ez * ez) + rho) <= ps.w) {
            off = sp.x; cnt = sp.y; listed = true;
        }
    }
    const float cmag = 1e-6f * (fabsf(c.x) + fabsf(c.y) + fabsf(c.z) + rho);
    // The candidates' values at the centre: the first SDFK_CELL_NB batches of 64 stay in registers (one evaluation per
    // candidate); what lies beyond is evaluated again for the count and once more for the write.
#ifndef SDFK_CELL_NB
#define SDFK_CELL_NB 8
#endif
#define SDFK_POOL_SHARDS 256u                   // (host: prepare_cells)
    float ev[SDFK_CELL_NB];
    unsigned rv[SDFK_CELL_NB];
    float m = 3.0e38f;
#pragma unroll
    for (int bt = 0; bt < SDFK_CELL_NB; ++bt) {
        const unsigned j = 64u * bt + (unsigned)lane;
        ev[bt] = 3.0e38f;
        rv[bt] = 0u;
        if (64u * bt < cnt) {
            if (j < cnt) {
                rv[bt] = listed ? cp.cand[off + j] : sdfk_member_record(j);
                ev[bt] = SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(rv[bt], c, PRM, TAB);
            }
            m = fminf(m, ev[bt]);
        }
    }
#pragma unroll 4
    for (unsigned j0 = 64u * SDFK_CELL_NB; j0 < cnt; j0 += 64u) {
        const unsigned j = j0 + (unsigned)lane;
        if (j < cnt) m = fminf(m, SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(listed ? cp.cand[off + j] : sdfk_member_record(j), c, PRM, TAB));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fminf(m, __shfl_xor(m, o));
    const float thr0 = 1.0001f * SDFK_CHAIN_KMAX * rho + SDFK_CHAIN_KMAX * cmag + 1e-6f * (1.0f + fabsf(m));
    unsigned total = 0u;
#pragma unroll
    for (int bt = 0; bt < SDFK_CELL_NB; ++bt)
        if (64u * bt < cnt) total += (unsigned)__builtin_popcountll(__ballot(64u * bt + (unsigned)lane < cnt && !(ev[bt] - m >= thr0 + 1e-6f * fabsf(ev[bt]))));
#pragma unroll 4
    for (unsigned j0 = 64u * SDFK_CELL_NB; j0 < cnt; j0 += 64u) {
        const unsigned j = j0 + (unsigned)lane;
        bool run = false;
        if (j < cnt) {
            const float e = SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(listed ? cp.cand[off + j] : sdfk_member_record(j), c, PRM, TAB);
            run = !(e - m >= thr0 + 1e-6f * fabsf(e));
        }
        total += (unsigned)__builtin_popcountll(__ballot(run));
    }
    // room for the list: a returning atomic add — on ONE word 19,000 cells would queue for 0.2 ms (about 90 such atomics
    // per microsecond and address), so the pool is cut into SDFK_POOL_SHARDS shards with a head each, dealt by cell index
    unsigned at = 0u;
    const unsigned shard = cell & (SDFK_POOL_SHARDS - 1u);
    if (lane == 0) at = atomicAdd(cp.head + 16u * shard, total);
    at = __shfl(at, 0);
    if (at > cp.shard_cap || total > cp.shard_cap - at) {       // the shard is full: bricks of this cell probe every member
        if (lane == 0) cp.span[cell] = make_uint2(0u, SDFK_ALL_ALIVE);
        return;
    }
    at += cp.base + shard * cp.shard_cap;
    unsigned n_out = 0u;
#pragma unroll
    for (int bt = 0; bt < SDFK_CELL_NB; ++bt)
        if (64u * bt < cnt) {
            const bool run = 64u * bt + (unsigned)lane < cnt && !(ev[bt] - m >= thr0 + 1e-6f * fabsf(ev[bt]));
            const unsigned long long bits = __ballot(run);
            if (run) cp.cand[at + n_out + (unsigned)__builtin_popcountll(bits & ((1ull << lane) - 1ull))] = rv[bt];
            n_out += (unsigned)__builtin_popcountll(bits);
        }
#pragma unroll 4
    for (unsigned j0 = 64u * SDFK_CELL_NB; j0 < cnt; j0 += 64u) {
        const unsigned j = j0 + (unsigned)lane;
        bool run = false;
        unsigned rec = 0u;
        if (j < cnt) {
            rec = listed ? cp.cand[off + j] : sdfk_member_record(j);
            const float e = SDFK_CHAIN_SGN * sdfk_leaf_rec<float>(rec, c, PRM, TAB);
            run = !(e - m >= thr0 + 1e-6f * fabsf(e));
        }
        const unsigned long long bits = __ballot(run);
        if (run) cp.cand[at + n_out + (unsigned)__builtin_popcountll(bits & ((1ull << lane) - 1ull))] = rec;
        n_out += (unsigned)__builtin_popcountll(bits);
    }
    if (lane == 0) cp.span[cell] = make_uint2(at, total);
}
#else
#define SDFK_CELLS_KARG
#endif
)SDFKR";
static const char kRowsArray[] = R"SDFKR(
extern "C" __global__ __launch_bounds__(64 * SDFK_RWAVES) SDFK_ROWS_ATTR void sdfk_spec_r(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    sdfk_rowgeom g, float* __restrict__ out, unsigned* __restrict__ flags, unsigned thr SDFK_CELLS_KARG) {
    const SrcArray s = {co, stride};
    sdfk_rows_kernel(PRM, TAB, s, g, out, flags, thr SDFK_CELLS_PASS);
}
#ifdef SDFK_CELLS
extern "C" __global__ __launch_bounds__(256) void sdfk_spec_cells(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    sdfk_rowgeom g, sdfk_cellpass cp) {
    const SrcArray s = {co, stride};
    sdfk_cells_kernel(PRM, TAB, s, g, cp);
}
#endif
)SDFKR";
static const char kRowsGrid[] = R"SDFKR(
// the same on a regular grid expanded from three per-axis tables (no coordinate array: 4 B/point); the slab
// starts at a row boundary and out[0] is its first point
extern "C" __global__ __launch_bounds__(64 * SDFK_RWAVES) void sdfk_spec_rg(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid s, sdfk_rowgeom g, float* __restrict__ out,
    unsigned* __restrict__ flags, unsigned thr SDFK_CELLS_KARG) {
    sdfk_rows_kernel(PRM, TAB, s, g, out, flags, thr SDFK_CELLS_PASS);
}
#ifdef SDFK_CELLS
extern "C" __global__ __launch_bounds__(256) void sdfk_spec_cellsg(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid s, sdfk_rowgeom g, sdfk_cellpass cp) {
    sdfk_cells_kernel(PRM, TAB, s, g, cp);
}
#endif
)SDFKR";
static const char kRowsMask[] = R"SDFKR(
// test aid: the skip masks (two 64-bit words per brick: sites 0-31, 32-63; bit 2k = first operand of site k
// skipped, bit 2k+1 = second) followed by the "uniform rows" flag in a third word
extern "C" __global__ __launch_bounds__(64 * SDFK_RWAVES) void sdfk_spec_rmask(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    sdfk_rowgeom g, unsigned long long* __restrict__ masks) {
    __shared__ __attribute__((aligned(16))) sdfk_rowmeta meta;
    const SrcArray s = {co, stride};
    unsigned rb, c;
    sdfk_rows_prepare(PRM, TAB, s, g, &meta, sdfk_bx(), rb, c);
    const unsigned q = sdfk_bx() * SDFK_RNBRICK + sdfk_tx();
    if (sdfk_tx() < SDFK_RNBRICK && q < g.nbricks) {
        masks[3ull * q] = meta.mask[0][sdfk_tx()];
        masks[3ull * q + 1] = meta.mask[1][sdfk_tx()];
        masks[3ull * q + 2] = meta.uniform[sdfk_tx()];
    }
}
)SDFKR";

namespace {

struct Gen {
    const sdfk_opinfo* ops;
    int n_ops;
    const uint32_t* code;
    size_t n_instr;
    const std::vector<sdfk_cullsite>* sites;
    std::string s;
    bool rows = false;   // emitting for the row-block kernel: registers are arrays of SDFK_NP packed pairs

    // "root transforms": XFORM instructions that read the untouched input point C0. slot[i] = their
    // index (else -1). On a brick whose points share x and y their first six fmas are one value per
    // brick (op_xform_base), computed by the probe lane and read back from LDS by the consumers.
    std::vector<int> root_slot;
    int n_root = 0;
    void find_roots() {
        root_slot.assign(n_instr, -1);
        bool c0_intact = true;
        for (size_t i = 0; i < n_instr; ++i) {
            const uint32_t w = code[2 * i];
            const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u;
            if ((int)op >= n_ops) continue;
            if (ops[op].kind != SDFK_KIND_C_C) continue;
            if (!strcmp(ops[op].name, "XFORM") && b == 0 && c0_intact && n_root < 64) root_slot[i] = n_root++;
            if (a == 0) c0_intact = false;
        }
    }

    // mode 0: plain; 1: probe (stores the base of root transforms); 2: culled (reads it when ZRUN)
    void instr_x(size_t i, const char* indent, int mode) {
        const int r = root_slot.empty() ? -1 : root_slot[i];
        // (the row-block probe keeps no bases: every lane computes its own in phase C)
        if (r < 0 || mode == 0 || (rows && mode == 1)) {
            instr(i, indent, rows && mode == 2);
            return;
        }
        char buf[1400];
        const uint32_t w = code[2 * i], poff = code[2 * i + 1];
        const unsigned a = (w >> 8) & 255u;
        if (rows) {
            // (SDFK_FLAT: x shared, z = 0 exactly, y varies: M2 * 0 drops out of the sum without a rounding, so
            //  fma(M1, y, fma(M0, x, -c)) IS op_xform's value and the x part is one value per row)
            snprintf(buf, sizeof buf,
                     "%sif constexpr (ZRUN) {\n"
                     "#ifdef SDFK_FLAT\n"
                     "%s  const V3T<float> bs = op_xform_base_x(X, PRM + %u);\n"
                     "%s  const V3P bp = {sp<f2>(bs.x), sp<f2>(bs.y), sp<f2>(bs.z)};\n"
                     "%s  SDFK_EACH C_%u[q] = op_xform_y(bp, C_0[q].y, PRM + %u);\n"
                     "#else\n"
                     "%s  const V3T<float> bs = op_xform_base(X, Y, PRM + %u);\n"
                     "%s  const V3P bp = {sp<f2>(bs.x), sp<f2>(bs.y), sp<f2>(bs.z)};\n"
                     "%s  SDFK_EACH C_%u[q] = op_xform_z(bp, C_0[q].z, PRM + %u);\n"
                     "#endif\n"
                     "%s} else { SDFK_EACH C_%u[q] = op_xform(C_0[q], PRM + %u, TAB, 0); }\n",
                     indent, indent, poff, indent, indent, a, poff, indent, poff, indent, indent, a, poff, indent, a, poff);
        } else if (mode == 1) {
            snprintf(buf, sizeof buf,
                     "%s{ const V3T<float> bs = op_xform_base(C_0.x, C_0.y, PRM + %u); bases[%d] = bs.x; bases[%d] = bs.y; "
                     "bases[%d] = bs.z;\n%s  C_%u = op_xform_z(bs, C_0.z, PRM + %u); }\n",
                     indent, poff, 3 * r, 3 * r + 1, 3 * r + 2, indent, a, poff);
        } else {
            snprintf(buf, sizeof buf,
                     "%sif constexpr (ZRUN) { const V3T<T> bs = {sp<T>(bases[%d]), sp<T>(bases[%d]), sp<T>(bases[%d])};\n"
                     "%s  C_%u = op_xform_z(bs, C_0.z, PRM + %u); }\n%selse C_%u = op_xform(C_0, PRM + %u, TAB, 0);\n",
                     indent, 3 * r, 3 * r + 1, 3 * r + 2, indent, a, poff, indent, a, poff);
        }
        s += buf;
    }

    void instr(size_t i, const char* indent) { instr(i, indent, false); }
    void instr(size_t i, const char* indent, bool arr) {
        char buf[256];
        const uint32_t w = code[2 * i], poff = code[2 * i + 1];
        const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
        if ((int)op >= n_ops) return;
        const sdfk_opinfo& o = ops[op];
        if (!strcmp(o.name, "V_FIELD")) {   // only ever reached for the plain kernels (programs with fields have no sites)
            snprintf(buf, sizeof buf, "%sV_%u = sdfk_aux<T>(AUX, AUXS, %u);\n", indent, a, c);
            s += buf;
            return;
        }
        const char* e = arr ? "SDFK_EACH " : "";
        const char* x = arr ? "[q]" : "";
        switch (o.kind) {
            case SDFK_KIND_C_C:
                snprintf(buf, sizeof buf, "%s%sC_%u%s = %s(C_%u%s, PRM + %u, TAB, %u);\n", indent, e, a, x, o.func, b, x, poff, c);
                break;
            case SDFK_KIND_V_C:
                snprintf(buf, sizeof buf, "%s%sV_%u%s = %s(C_%u%s, PRM + %u, TAB);\n", indent, e, a, x, o.func, b, x, poff);
                break;
            case SDFK_KIND_V_V:
                snprintf(buf, sizeof buf, "%s%sV_%u%s = %s(V_%u%s, PRM + %u);\n", indent, e, a, x, o.func, b, x, poff);
                break;
            default:
                snprintf(buf, sizeof buf, "%s%sV_%u%s = %s(V_%u%s, V_%u%s, PRM + %u);\n", indent, e, a, x, o.func, b, x, c, x, poff);
                break;
        }
        s += buf;
    }

    // (no initialisers: skipped subtrees never have their registers read — set_cull's liveness check)
    // init: give every register a value up front. Skipped subtrees never have their registers READ (set_cull's
    // liveness check), but without it the compiler materialises a zero for each of them on every skip edge —
    // 8 moves per nesting level of a combiner chain in the row-block kernel; one initialisation costs them once.
    void declare(const char* ctype, const char* vtype, bool arr, bool init = false) {
        std::set<unsigned> cregs, vregs;
        for (size_t i = 0; i < n_instr; ++i) {
            const uint32_t w = code[2 * i];
            const unsigned op = w & 255u, a = (w >> 8) & 255u;
            if ((int)op >= n_ops) continue;
            if (ops[op].kind == SDFK_KIND_C_C) cregs.insert(a);
            else vregs.insert(a);
        }
        char buf[160];
        const char* dim = arr ? "[SDFK_NP]" : "";
        const char* ini = init ? " = {}" : "";
        for (unsigned c : cregs)
            if (c != 0) {
                snprintf(buf, sizeof buf, "    %s C_%u%s;\n", ctype, c, dim);
                s += buf;
            }
        for (unsigned v : vregs) {
            snprintf(buf, sizeof buf, "    %s V_%u%s%s;\n", vtype, v, dim, ini);
            s += buf;
        }
    }

    // which site (if any) opens at instruction i and closes no later than hi: the outermost one
    int site_opening_at(size_t i, size_t hi) const {
        int best = -1;
        for (size_t k = 0; k < sites->size(); ++k) {
            const sdfk_cullsite& t = (*sites)[k];
            if (t.a0 == i && t.comb <= hi && (best < 0 || t.comb > (*sites)[best].comb)) best = (int)k;
        }
        return best;
    }

    // gap expressions at a site, by combiner
    void site_ops(const sdfk_cullsite& t, char* gapB, char* gapA, char* wexpr, bool* neg_b, size_t cap) const {
        const uint32_t w = code[2 * t.comb], poff = code[2 * t.comb + 1];
        const unsigned b = (w >> 16) & 255u, c = w >> 24;
        const char* name = ops[w & 255u].name;
        *neg_b = false;
        snprintf(wexpr, cap, "0.0f");
        if (!strcmp(name, "VMIN") || !strcmp(name, "SMIN2") || !strcmp(name, "SMIN3")) {
            snprintf(gapB, cap, "(V_%u - V_%u)", c, b);
            snprintf(gapA, cap, "(V_%u - V_%u)", b, c);
        } else if (!strcmp(name, "VMAX") || !strcmp(name, "SMAX3")) {
            snprintf(gapB, cap, "(V_%u - V_%u)", b, c);
            snprintf(gapA, cap, "(V_%u - V_%u)", c, b);
        } else {  // VSUBTRACT, SSUB3: max(a, -b)
            snprintf(gapB, cap, "(V_%u + V_%u)", b, c);
            snprintf(gapA, cap, "(-V_%u - V_%u)", b, c);
            *neg_b = true;
        }
        if (name[0] == 'S') snprintf(wexpr, cap, "PRM[%u]", poff);
    }

    // ---- leaves: operand ranges without a site inside (the children of an n-ary UNION, the primitives of a chain) ----
    // When every instruction outside the combiners lives in such a range and a range reads nothing but the input
    // point, the probe needs no sequential walk of the tree: the leaves are evaluated by ALL lanes of the workgroup —
    // one (leaf, centre) pair per lane — and one lane per centre then folds the combiners over the stored leaf values.
    // Leaves with the same operation sequence (children built the same way with other numbers) share one function
    // that takes a table of parameter offsets, so 50 children of 4 kinds cost 4 functions, not 50 inlined bodies.
    struct Leaf {
        size_t lo, hi;
        int group, member;
        unsigned out;            // value register the range leaves its result in
    };
    struct Group {
        std::vector<uint32_t> sig;   // canonical instruction words
        std::vector<int> members;    // leaf indices
        unsigned n_c = 1, n_v = 0;   // canonical registers used
        unsigned out = 0;            // canonical result register
    };
    std::vector<Leaf> leaves;
    std::vector<Group> groups;
    std::vector<int> leaf_at;        // instruction -> leaf (or -1)
    bool simt = false;

    // [region_lo, region_hi]: the instructions the leaf analysis is about — the whole program, or (chain mode with a REST,
    // see chain_analyse) the chain's own instructions: what lies outside is evaluated per point by sdfk_chain_tail
    size_t region_lo = 0, region_hi = (size_t)-1;
    bool analyse_leaves() {
        leaves.clear();
        groups.clear();
        leaf_at.assign(n_instr, -1);
        if (sites->empty()) return false;
        auto holds_site = [&](size_t lo, size_t hi) {
            for (const sdfk_cullsite& t : *sites)
                if (t.comb >= lo && t.comb <= hi) return true;
            return false;
        };
        std::vector<std::pair<size_t, size_t>> rs;
        for (const sdfk_cullsite& t : *sites) {
            if (!holds_site(t.a0, t.a1)) {
                if (!t.skip_a_ok) return false;
                rs.push_back({t.a0, t.a1});
            }
            if (!holds_site(t.b0, t.b1)) {
                if (!t.skip_b_ok) return false;
                rs.push_back({t.b0, t.b1});
            }
        }
        std::sort(rs.begin(), rs.end());
        rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
        for (size_t k = 1; k < rs.size(); ++k)
            if (rs[k].first <= rs[k - 1].second) return false;
        for (size_t k = 0; k < rs.size(); ++k)
            for (size_t i = rs[k].first; i <= rs[k].second; ++i) leaf_at[i] = (int)k;
        int c0_writer = -1;                                              // leaf that has overwritten C_0 (-1: nobody yet)
        for (size_t i = 0; i < n_instr; ++i) {
            const uint32_t w = code[2 * i];
            const unsigned op = w & 255u, a = (w >> 8) & 255u;
            if ((int)op >= n_ops) return false;
            const int kind = ops[op].kind;
            if (!strcmp(ops[op].name, "V_FIELD")) return false;
            const bool inside = i >= region_lo && i <= region_hi;
            if (inside && leaf_at[i] < 0 && kind != SDFK_KIND_V_V && kind != SDFK_KIND_V_VV) return false;
            if (!inside && i < region_lo && kind == SDFK_KIND_C_C && a == 0) return false;   // the leaves start from the INPUT point
            // a leaf may rewrite the input point for itself (the last child of a combiner reuses C_0: inside the leaf's
            // function that is a local copy) — but nobody else may read it afterwards
            const unsigned b = (w >> 16) & 255u;
            if ((kind == SDFK_KIND_C_C || kind == SDFK_KIND_V_C) && b == 0 && c0_writer != -1 && c0_writer != leaf_at[i]) return false;
            if (kind == SDFK_KIND_C_C && a == 0) c0_writer = leaf_at[i];
        }
        for (size_t k = 0; k < rs.size(); ++k) {
            Leaf lf{rs[k].first, rs[k].second, -1, -1, 0};
            std::vector<int> cmap(256, -1), vmap(256, -1);
            cmap[0] = 0;
            unsigned nc = 1, nv = 0;
            std::vector<uint32_t> sig;
            for (size_t i = lf.lo; i <= lf.hi; ++i) {
                const uint32_t w = code[2 * i];
                const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
                const int kind = ops[op].kind;
                unsigned ca = 0, cb = 0, cc = c;
                if (kind == SDFK_KIND_C_C || kind == SDFK_KIND_V_C) {
                    if (cmap[b] < 0) return false;                       // reads a coordinate register from outside
                    cb = (unsigned)cmap[b];
                } else {
                    if (vmap[b] < 0) return false;
                    cb = (unsigned)vmap[b];
                    if (kind == SDFK_KIND_V_VV) {
                        if (vmap[c] < 0) return false;
                        cc = (unsigned)vmap[c];
                    }
                }
                if (kind == SDFK_KIND_C_C) {
                    if (cmap[a] < 0) cmap[a] = (int)nc++;
                    ca = (unsigned)cmap[a];
                } else {
                    if (vmap[a] < 0) vmap[a] = (int)nv++;
                    ca = (unsigned)vmap[a];
                }
                if (nc > 255 || nv > 255) return false;
                sig.push_back(op | (ca << 8) | (cb << 16) | (cc << 24));
            }
            const uint32_t wl = code[2 * lf.hi];
            if (ops[wl & 255u].kind == SDFK_KIND_C_C) return false;
            lf.out = (wl >> 8) & 255u;
            int g = -1;
            for (size_t q = 0; q < groups.size(); ++q)
                if (groups[q].sig == sig) g = (int)q;
            if (g < 0) {
                Group ng;
                ng.sig = sig;
                ng.n_c = nc;
                ng.n_v = nv;
                ng.out = (unsigned)vmap[lf.out];
                groups.push_back(ng);
                g = (int)groups.size() - 1;
            }
            lf.group = g;
            lf.member = (int)groups[g].members.size();
            groups[g].members.push_back((int)k);
            leaves.push_back(lf);
        }
        return !leaves.empty();
    }

    // one function per group of equal leaves + its tables (parameter offsets per member and instruction, leaf ids)
    // rel: every leaf's parameters are ONE contiguous block of the table (leaves_contiguous()): the function takes the
    // block's base and adds compile-time offsets — no offset table, and the base may as well point into LDS
    std::vector<unsigned> group_npl;     // parameters per leaf of each group (rel mode)
    bool leaves_contiguous() {
        group_npl.assign(groups.size(), 0);
        for (const Leaf& lf : leaves) {
            unsigned at = code[2 * lf.lo + 1], total = 0;
            for (size_t i = lf.lo; i <= lf.hi; ++i) {
                if (code[2 * i + 1] != at) return false;
                const int np = ops[code[2 * i] & 255u].nparams;
                if (np < 0) return false;
                at += (unsigned)np;
                total += (unsigned)np;
            }
            group_npl[lf.group] = total;
        }
        return true;
    }
    void emit_groups(bool rel = false) {
        char buf[512];
        for (size_t g = 0; g < groups.size(); ++g) {
            const Group& G = groups[g];
            if (rel)
                snprintf(buf, sizeof buf,
                         "\ntemplate <typename T> static __device__ __forceinline__ T sdfk_grp%zu(V3T<T> C_0, const float* __restrict__ PRM, "
                         "const float* __restrict__ TAB) {\n    constexpr unsigned OFF[] = {",
                         g);
            else
                snprintf(buf, sizeof buf,
                         "\ntemplate <typename T> static __device__ __forceinline__ T sdfk_grp%zu(V3T<T> C_0, const unsigned* __restrict__ OFF, "
                         "const float* __restrict__ PRM, const float* __restrict__ TAB) {\n",
                         g);
            s += buf;
            if (rel) {
                unsigned at = 0;
                for (size_t j = 0; j < G.sig.size(); ++j) {
                    snprintf(buf, sizeof buf, "%uu,", at);
                    s += buf;
                    at += (unsigned)ops[G.sig[j] & 255u].nparams;
                }
                s += "};\n";
            }
            for (unsigned c = 1; c < G.n_c; ++c) {
                snprintf(buf, sizeof buf, "    V3T<T> C_%u;\n", c);
                s += buf;
            }
            for (unsigned v = 0; v < G.n_v; ++v) {
                snprintf(buf, sizeof buf, "    T V_%u;\n", v);
                s += buf;
            }
            for (size_t j = 0; j < G.sig.size(); ++j) {
                const uint32_t w = G.sig[j];
                const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
                const sdfk_opinfo& o = ops[op];
                switch (o.kind) {
                    case SDFK_KIND_C_C:
                        snprintf(buf, sizeof buf, "    C_%u = %s(C_%u, PRM + OFF[%zu], TAB, %u);\n", a, o.func, b, j, c);
                        break;
                    case SDFK_KIND_V_C:
                        snprintf(buf, sizeof buf, "    V_%u = %s(C_%u, PRM + OFF[%zu], TAB);\n", a, o.func, b, j);
                        break;
                    case SDFK_KIND_V_V:
                        snprintf(buf, sizeof buf, "    V_%u = %s(V_%u, PRM + OFF[%zu]);\n", a, o.func, b, j);
                        break;
                    default:
                        snprintf(buf, sizeof buf, "    V_%u = %s(V_%u, V_%u, PRM + OFF[%zu]);\n", a, o.func, b, c, j);
                        break;
                }
                s += buf;
            }
            snprintf(buf, sizeof buf, "    return V_%u;\n}\n", G.out);
            s += buf;
            if (!rel) {
                snprintf(buf, sizeof buf, "static __constant__ const unsigned sdfk_grp%zu_off[%zu][%zu] = {\n", g, G.members.size(),
                         G.sig.size());
                s += buf;
                for (int m : G.members) {
                    s += "    {";
                    for (size_t i = leaves[m].lo; i <= leaves[m].hi; ++i) {
                        snprintf(buf, sizeof buf, "%uu,", code[2 * i + 1]);
                        s += buf;
                    }
                    s += "},\n";
                }
                s += "};\n";
            }
            snprintf(buf, sizeof buf, "static __constant__ const unsigned short sdfk_grp%zu_leaf[%zu] = {", g, G.members.size());
            s += buf;
            for (int m : G.members) {
                snprintf(buf, sizeof buf, "%d,", m);
                s += buf;
            }
            s += "};\n";
        }
    }

    // every (leaf, centre) pair on a lane of its own: leafval[leaf][centre] = leaf(centre)
    void emit_probe_leaves(bool per_leaf = false, bool rel = false) {
        char buf[900];
        if (per_leaf) {
            // chain mode (one centre per brick, few bricks per workgroup): a lane takes a LEAF and walks the centres, so the
            // leaf's parameters — two dependent memory round trips, the offsets and then the values — are fetched once
            s += "\nstatic __device__ __forceinline__ void sdfk_probe_leaves(const float4* __restrict__ cen, float* __restrict__ leafval, "
                 "const float* __restrict__ PRM, const float* __restrict__ TAB) {\n";
            for (size_t g = 0; g < groups.size(); ++g) {
                snprintf(buf, sizeof buf,
                         "    _Pragma(\"unroll 2\") for (unsigned m = sdfk_tx(); m < %zuu; m += 64u * SDFK_RWAVES) {\n"
                         "        const unsigned leaf = sdfk_grp%zu_leaf[m];\n"
                         "        _Pragma(\"unroll\") for (unsigned c = 0; c < SDFK_NCEN; ++c) {\n"
                         "            const float4 cc = cen[c];\n"
                         "            const V3T<float> p = {cc.x, cc.y, cc.z};\n"
                         "            leafval[leaf * SDFK_NCEN + c] = sdfk_grp%zu<float>(p, PRM + sdfk_leaf_base[leaf], TAB);\n"
                         "        }\n"
                         "    }\n",
                         groups[g].members.size(), g, g);
                s += buf;
            }
            s += "}\n";
            return;
        }
        // (several rounds of a big group in flight at once: each round is two dependent memory round trips — the offsets,
        //  then the parameters — and nothing else hides them)
        s += "\n#ifndef SDFK_LEAF_UNROLL\n#define SDFK_LEAF_UNROLL _Pragma(\"unroll 1\")\n#endif\n";
        s += "static __device__ __forceinline__ void sdfk_probe_leaves(const float4* __restrict__ cen, float* __restrict__ leafval, "
             "const float* __restrict__ PRM, const float* __restrict__ TAB) {\n";
        for (size_t g = 0; g < groups.size(); ++g) {
            snprintf(buf, sizeof buf,
                     "    SDFK_LEAF_UNROLL for (unsigned item = sdfk_tx(); item < %zuu * SDFK_NCEN; item += 64u * SDFK_RWAVES) {\n"
                     "        unsigned m = item / SDFK_NCEN;\n"
                     "        if (SDFK_NCEN %% 64 == 0) m = __builtin_amdgcn_readfirstlane(m);   // one member per wave: scalar parameter loads\n"
                     "        const unsigned c = item - m * SDFK_NCEN;\n"
                     "        const float4 cc = cen[c];\n"
                     "        const V3T<float> p = {cc.x, cc.y, cc.z};\n"
                     "        leafval[sdfk_grp%zu_leaf[m] * SDFK_NCEN + c] = sdfk_grp%zu<float>(p, %s, TAB);\n"
                     "    }\n",
                     groups[g].members.size(), g, g,
                     rel ? ("PRM + sdfk_leaf_base[sdfk_grp" + std::to_string(g) + "_leaf[m]]").c_str()
                         : ("sdfk_grp" + std::to_string(g) + "_off[m], PRM").c_str());
            s += buf;
        }
        s += "}\n";
    }

    // one lane per centre: the combiners (and whatever else lies outside the leaves) over the stored leaf values,
    // with the skip decision of every site exactly as in the sequential probe below
    void emit_probe_fold() {
        s += "\nstatic __device__ __forceinline__ void sdfk_probe_fold(const float* __restrict__ leafval, unsigned cen, V3T<float> C_0, "
             "float rho, const float* __restrict__ PRM, unsigned long long (&mask)[SDFK_NMASK]) {\n";
        std::set<unsigned> vregs;
        for (size_t i = 0; i < n_instr; ++i)
            if (leaf_at[i] < 0) vregs.insert((code[2 * i] >> 8) & 255u);
        for (const Leaf& lf : leaves) vregs.insert(lf.out);
        char buf[512], gB[64], gA[64], wx[32];
        for (unsigned v : vregs) {
            snprintf(buf, sizeof buf, "    float V_%u;\n", v);
            s += buf;
        }
        s += "    _Pragma(\"unroll\") for (int w_ = 0; w_ < SDFK_NMASK; ++w_) mask[w_] = 0ull;\n";
        s += "    const float cmag = 1e-6f * (fabsf(C_0.x) + fabsf(C_0.y) + fabsf(C_0.z) + rho);\n";
        for (size_t i = 0; i < n_instr; ++i) {
            if (leaf_at[i] >= 0) {
                const Leaf& lf = leaves[leaf_at[i]];
                snprintf(buf, sizeof buf, "    V_%u = leafval[%d * SDFK_NCEN + cen];\n", lf.out, leaf_at[i]);
                s += buf;
                i = lf.hi;
                continue;
            }
            for (size_t k = 0; k < sites->size(); ++k) {
                const sdfk_cullsite& t = (*sites)[k];
                if (t.comb != i) continue;
                bool neg;
                site_ops(t, gB, gA, wx, &neg, sizeof gB);
                const uint32_t w = code[2 * i];
                const unsigned b = (w >> 16) & 255u, c = w >> 24;
                char mv[24];
                if (rows) snprintf(mv, sizeof mv, "mask[%zu]", k / 32);
                else snprintf(mv, sizeof mv, "mask");
                const unsigned sh = 2 * (unsigned)(k & 31);
                snprintf(buf, sizeof buf,
                         "    { const float thr = %s + %.9ef * rho + %.9ef * cmag + 1e-6f * (1.0f + fabsf(V_%u) + fabsf(V_%u));\n"
                         "      if (%d && %s >= thr) %s |= %lluull;\n"
                         "      else if (%d && %s >= thr) %s |= %lluull; }\n",
                         wx, (double)t.k * 1.0001, (double)t.k, b, c, t.skip_b_ok, gB, mv, 2ull << sh, t.skip_a_ok, gA, mv,
                         1ull << sh);
                s += buf;
            }
            instr(i, "    ");
        }
        s += "}\n";
    }

    // ---- chain mode: one long n-ary hard min / max over leaves (CombineGeometry("UNION").combine(*many)) ----
    // leaves[k] (k = 0..n-1) in program order, combined by `V_acc = OP(V_acc, V_k)` right after each leaf k >= 1, optionally
    // followed by value modifications of the result. Requires analyse_leaves() over ALL sites of the program.
    struct Chain {
        bool ok = false;
        unsigned acc = 0;           // accumulator register
        bool is_max = false;        // VMAX (INTERSECT) instead of VMIN (UNION)
        size_t tail = 0;            // first instruction after the last combiner
        size_t lo = 0;              // first instruction of the first leaf
        bool rest = false;          // instructions before the chain, or after it other than in-place modifications of its value
        int result = 0;             // the program's result register
        std::vector<float> k;       // Lipschitz sum of level k (site k - 1)
    } chain;
    std::vector<sdfk_cullsite> chain_sites;      // the sites of the chain itself (a program with a rest has others too)

    bool analyse_chain(int result_reg) {
        chain = Chain();
        const size_t n = leaves.size();
        if (n < 2 || sites->size() != n - 1) return false;
        const uint32_t w0 = code[2 * (*sites)[0].comb];
        const char* name = ops[w0 & 255u].name;
        if (strcmp(name, "VMIN") && strcmp(name, "VMAX")) return false;
        chain.is_max = !strcmp(name, "VMAX");
        chain.acc = (w0 >> 8) & 255u;
        chain.k.assign(n, 0.0f);
        for (size_t j = 0; j + 1 < n; ++j) {
            const sdfk_cullsite& t = (*sites)[j];
            const uint32_t w = code[2 * t.comb];
            if ((w & 255u) != (w0 & 255u)) return false;
            const unsigned a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
            if (a != chain.acc || b != chain.acc) return false;
            if (t.a0 != leaves[0].lo || t.b0 != leaves[j + 1].lo || t.b1 != leaves[j + 1].hi || t.comb != t.b1 + 1) return false;
            if (leaves[j + 1].out != c || c == chain.acc) return false;
            if (j == 0 && (t.a1 != leaves[0].hi || leaves[0].out != chain.acc)) return false;
            if (j > 0 && t.a1 != (*sites)[j - 1].comb) return false;
            chain.k[j + 1] = t.k;
        }
        chain.tail = (*sites)[n - 2].comb + 1;
        chain.lo = leaves[0].lo;
        chain.result = result_reg;
        chain.rest = chain.lo > 0 || (unsigned)result_reg != chain.acc;
        for (size_t i = chain.tail; i < n_instr; ++i) {              // value modifications of the result, in place: no rest
            const uint32_t w = code[2 * i];
            if (ops[w & 255u].kind != SDFK_KIND_V_V || ((w >> 8) & 255u) != chain.acc || ((w >> 16) & 255u) != chain.acc) chain.rest = true;
        }
        // the rest runs un-culled, straight-line, for every point: worth it while it is small next to the chain (a clip, a
        // ground plane, a body) — a program of many medium-sized unions is better off on the mask kernels, whose widest
        // sites skip whole unions
        // (and at most 256 instructions: it is compiled inline)
        if (chain.rest && chain.lo + (n_instr - chain.tail) > std::min<size_t>(256, 64 + (chain.tail - chain.lo) / 4)) return false;
        if (chain.rest && !rest_ok()) return false;
        chain.ok = true;
        return true;
    }
    // A program with a REST: instructions before the chain (other operands of the combiners above it) and after it (those
    // combiners, value modifications). sdfk_chain_tail evaluates them per point around the chain's value — which the
    // chain kernels produce from the brick's survivor list, exact at every point of the brick whatever is done with it
    // afterwards. That is the original program iff nothing after the chain reads a register the chain's instructions wrote,
    // other than its accumulator (temporaries of the leaves; C_0 where the last leaf reused it).
    bool rest_ok() const {
        std::vector<char> wc(256, 0), wv(256, 0);                   // written by the chain's instructions
        for (size_t i = chain.lo; i < chain.tail; ++i) {
            const uint32_t w = code[2 * i];
            if (ops[w & 255u].kind == SDFK_KIND_C_C) wc[(w >> 8) & 255u] = 1;
            else wv[(w >> 8) & 255u] = 1;
        }
        std::vector<char> sc(256, 0), sv(256, 0);                   // written since, by the instructions after the chain
        sv[chain.acc] = 1;
        for (size_t i = chain.tail; i < n_instr; ++i) {
            const uint32_t w = code[2 * i];
            const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
            const int kind = ops[op].kind;
            if (kind == SDFK_KIND_C_C || kind == SDFK_KIND_V_C) {
                if (wc[b] && !sc[b]) return false;
            } else {
                if (wv[b] && !sv[b]) return false;
                if (kind == SDFK_KIND_V_VV && wv[c] && !sv[c]) return false;
            }
            if (kind == SDFK_KIND_C_C) sc[a] = 1;
            else sv[a] = 1;
        }
        return true;
    }

    // tables + dispatcher shared by the chain-mode kernels
    void emit_chain_common() {
        char buf[768];
        const size_t n = leaves.size();
        snprintf(buf, sizeof buf, "\n#define SDFK_CHAIN 1\n#define SDFK_NLEAF %zu\n#define SDFK_CHAIN_SGN %s\n", n,
                 chain.is_max ? "(-1.0f)" : "1.0f");
        s += buf;
        emit_groups(true);
        s += "static __constant__ const unsigned short sdfk_leaf_grp[SDFK_NLEAF] = {";
        for (const Leaf& lf : leaves) {
            snprintf(buf, sizeof buf, "%d,", lf.group);
            s += buf;
        }
        s += "};\nstatic __constant__ const unsigned sdfk_leaf_base[SDFK_NLEAF] = {";       // first parameter of the leaf's block
        unsigned npl_max = 1, n_params = 1;
        for (const Leaf& lf : leaves) {
            snprintf(buf, sizeof buf, "%uu,", code[2 * lf.lo + 1]);
            s += buf;
            npl_max = std::max(npl_max, group_npl[lf.group]);
            n_params = std::max(n_params, code[2 * lf.lo + 1] + group_npl[lf.group]);
        }
        s += "};\n";
        for (size_t i = 0; i < n_instr; ++i)
            if (i < chain.lo || i >= chain.tail)
                n_params = std::max(n_params, code[2 * i + 1] + (unsigned)std::max(0, ops[code[2 * i] & 255u].nparams));
        float kmax = 0.0f;
        for (float k : chain.k) kmax = std::max(kmax, k);
        snprintf(buf, sizeof buf,
                 "#define SDFK_CHAIN_KMAX %.9ef   // largest Lipschitz sum of a level: bounds any pair of children\n"
                 "#define SDFK_NPLMAX %u          // parameters of the largest leaf\n"
                 "#define SDFK_NPARAMS %u         // parameters the leaves and the tail touch (gathers are clamped to it)\n",
                 (double)kmax, npl_max, n_params);
        s += buf;
        // leaf k (wave-uniform) at the lane's points, parameters at P (the leaf's block: in the table or staged in LDS)
        s += "template <typename T> static __device__ __forceinline__ T sdfk_leaf_at(unsigned grp, V3T<T> C_0, const float* __restrict__ P, "
             "const float* __restrict__ TAB) {\n    switch (grp) {\n";
        for (size_t g = 0; g < groups.size(); ++g) {
            snprintf(buf, sizeof buf, "        case %zuu: return sdfk_grp%zu<T>(C_0, P, TAB);\n", g, g);
            s += buf;
        }
        s += "        default: return sp<T>(0.0f);\n    }\n}\n"
             "template <typename T> static __device__ __forceinline__ T sdfk_leaf(unsigned k, V3T<T> C_0, const float* __restrict__ PRM, "
             "const float* __restrict__ TAB) {\n    return sdfk_leaf_at<T>(sdfk_leaf_grp[k], C_0, PRM + sdfk_leaf_base[k], TAB);\n}\n"
             // a member as ONE word — kind << 24 | first parameter —: what the candidate lists and the bricks' survivor lists
             // carry, so that nothing lies between a list entry and the member's parameters
             "static __device__ __forceinline__ unsigned sdfk_member_record(unsigned k) { return ((unsigned)sdfk_leaf_grp[k] << 24) | sdfk_leaf_base[k]; }\n"
             "template <typename T> static __device__ __forceinline__ T sdfk_leaf_rec(unsigned rec, V3T<T> C_0, const float* __restrict__ PRM, "
             "const float* __restrict__ TAB) {\n    return sdfk_leaf_at<T>(rec >> 24, C_0, PRM + (rec & 0xffffffu), TAB);\n}\n";
        if (groups.size() > 255 || n_params >= (1u << 24)) s += "#define SDFK_NO_CELLS 1   // (records hold 8 bits of kind, 24 of parameter offset)\n";
        const char* cmb = chain.is_max ? "cmb_max" : "cmb_min";
        // the result modifications, applied to an accumulator of type T
        s += "template <typename T> static __device__ __forceinline__ T sdfk_chain_tail(T V_chain, V3T<T> C_0, const float* __restrict__ PRM, "
             "const float* __restrict__ TAB) {\n";
        if (!chain.rest) {
            s += "    T V_acc = V_chain;\n";
            for (size_t i = chain.tail; i < n_instr; ++i) {
                const uint32_t w = code[2 * i], poff = code[2 * i + 1];
                snprintf(buf, sizeof buf, "    V_acc = %s(V_acc, PRM + %u);\n", ops[w & 255u].func, poff);
                s += buf;
            }
            s += "    return V_acc;\n}\n";
        } else {
            // the rest of the program around the chain's value: what comes before the chain, then what comes after it
            declare("V3T<T>", "T", false);
            for (size_t i = 0; i < chain.lo; ++i) instr(i, "    ");
            snprintf(buf, sizeof buf, "    V_%u = V_chain;\n", chain.acc);
            s += buf;
            for (size_t i = chain.tail; i < n_instr; ++i) instr(i, "    ");
            snprintf(buf, sizeof buf, "    return V_%d;\n}\n", chain.result);
            s += buf;
        }
        // plain evaluation: every leaf in order
        snprintf(buf, sizeof buf,
                 "template <typename T> static __device__ __forceinline__ T sdfk_point(V3T<T> C_0, const float* __restrict__ PRM, "
                 "const float* __restrict__ TAB, const float* __restrict__, long long) {\n"
                 "    T acc = sdfk_leaf<T>(0u, C_0, PRM, TAB);\n"
                 "    _Pragma(\"unroll 1\") for (unsigned k = 1u; k < SDFK_NLEAF; ++k) acc = %s(acc, sdfk_leaf<T>(k, C_0, PRM, TAB), PRM);\n"
                 "    return sdfk_chain_tail<T>(acc, C_0, PRM, TAB);\n}\n",
                 cmb);
        s += buf;
        // culled evaluation of one row-block brick: the levels on the brick's list, in order
        snprintf(buf, sizeof buf,
                 "#define SDFK_CHAIN_CMB %s\n", cmb);
        s += buf;
    }

    // ---- probe: full evaluation at a brick centre + skip decisions ----
    // For a brick of radius rho around the centre c and a combiner with operand fields a, b of
    // Lipschitz constants L_a, L_b (k = L_a + L_b): gap(c) >= w + k*rho  =>  gap(p) >= w for every p
    // of the brick, and then the (smooth) min / max returns the other operand bit-exactly.
    void emit_probe() {
        if (rows)
            s += "\nstatic __device__ __forceinline__ void sdfk_probe_r(V3T<float> C_0, float rho, "
                 "const float* __restrict__ PRM, const float* __restrict__ TAB, unsigned long long (&mask)[SDFK_NMASK]) {\n";
        else
            s += "\nstatic __device__ __forceinline__ unsigned long long sdfk_probe(V3T<float> C_0, float rho, float* bases, "
                 "const float* __restrict__ PRM, const float* __restrict__ TAB) {\n";
        declare("V3", "float", false);
        s += rows ? "    _Pragma(\"unroll\") for (int w_ = 0; w_ < SDFK_NMASK; ++w_) mask[w_] = 0ull;\n" : "    unsigned long long mask = 0ull;\n";
        // fp32 rounding of an operand grows with the magnitude of the coordinates it is computed from (about
        // 6e-8 |p| per rounding step of a transform), not with its value: a scene far from the origin needs a margin
        // in |c| + rho, scaled by the Lipschitz sum of the site, on top of the margin in the operand values
        s += "    const float cmag = 1e-6f * (fabsf(C_0.x) + fabsf(C_0.y) + fabsf(C_0.z) + rho);\n";
        char buf[512], gB[64], gA[64], wx[32];
        for (size_t i = 0; i < n_instr; ++i) {
            for (size_t k = 0; k < sites->size(); ++k) {   // decisions read the operands BEFORE the combiner
                const sdfk_cullsite& t = (*sites)[k];      // overwrites its destination (dst may alias one)
                if (t.comb != i) continue;
                bool neg;
                site_ops(t, gB, gA, wx, &neg, sizeof gB);
                const uint32_t w = code[2 * i];
                const unsigned b = (w >> 16) & 255u, c = w >> 24;
                char mv[24];
                if (rows) snprintf(mv, sizeof mv, "mask[%zu]", k / 32);
                else snprintf(mv, sizeof mv, "mask");
                const unsigned sh = 2 * (unsigned)(k & 31);
                snprintf(buf, sizeof buf,
                         "    { const float thr = %s + %.9ef * rho + %.9ef * cmag + 1e-6f * (1.0f + fabsf(V_%u) + fabsf(V_%u));\n"
                         "      if (%d && %s >= thr) %s |= %lluull;\n"
                         "      else if (%d && %s >= thr) %s |= %lluull; }\n",
                         wx, (double)t.k * 1.0001, (double)t.k, b, c, t.skip_b_ok, gB, mv, 2ull << sh, t.skip_a_ok, gA, mv,
                         1ull << sh);
                s += buf;
            }
            instr_x(i, "    ", 1);
        }
        s += rows ? "}\n" : "    return mask;\n}\n";
    }

    // ---- culled evaluation ----
    // mask bit `bit` as a test on one of the 32-bit scalar words (row blocks: 2 * SDFK_NMASK of them; line bricks: 2)
    std::string bit_test(unsigned bit) const {
        char b[48];
        if (rows) snprintf(b, sizeof b, "(mw[%u] & %uu)", bit >> 5, 1u << (bit & 31));
        else snprintf(b, sizeof b, "(%s & %uu)", bit < 32 ? "mlo" : "mhi", 1u << (bit & 31));
        return b;
    }

    // the site (if any) whose whole span [a0, comb] is exactly [lo, hi]
    int site_spanning(size_t lo, size_t hi) const {
        for (size_t k = 0; k < sites->size(); ++k)
            if ((*sites)[k].a0 == lo && (*sites)[k].comb == hi) return (int)k;
        return -1;
    }

    // second operand of site k and its combiner; `replace`: expression that is true when the first operand is
    // irrelevant (the combiner's value is then the second operand alone)
    void emit_second(int k, const std::string& ind, const std::string& replace, int depth) {
        const sdfk_cullsite& t = (*sites)[k];
        const std::string tb = bit_test(2 * k + 1);
        const uint32_t w = code[2 * t.comb];
        const unsigned a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
        char gB[64], gA[64], wx[32];
        bool neg;
        site_ops(t, gB, gA, wx, &neg, sizeof gB);
        char ra[24], rb_[24], rc[24];
        snprintf(ra, sizeof ra, "V_%u", a);
        snprintf(rb_, sizeof rb_, "V_%u", b);
        snprintf(rc, sizeof rc, "V_%u", c);
        const std::string e = rows ? "{ SDFK_EACH " : "", x = rows ? "[q]" : "", z = rows ? " }" : "";
        const std::string in2 = ind + "    ";
        s += ind + "if (!" + tb + ") {\n";
        emit_span(t.b0, t.b1, depth + 1);
        s += in2 + "if (" + replace + ") " + e + ra + x + " = " + (neg ? "-" : "") + rc + x + ";" + z + "\n";
        s += in2 + "else {\n";
        instr(t.comb, (in2 + "    ").c_str(), rows);
        s += in2 + "}\n";
        s += ind + "}\n";
        if (a != b) s += ind + "else " + e + ra + x + " = " + rb_ + x + ";" + z + "\n";
    }

    void emit_span(size_t lo, size_t hi, int depth) {
        std::string ind(4 + 4 * std::min(depth, 10), ' ');   // (cosmetic; capped so lines fit the line buffers)
        size_t i = lo;
        while (i <= hi) {
            const int k = site_opening_at(i, hi);
            if (k < 0) {
                instr_x(i, ind.c_str(), 2);
                ++i;
                continue;
            }
            // A left-deep chain — fold(fold(fold(p0, p1), p2), p3): what CombineGeometry builds for n operands and what
            // every chain of pairwise unions is — comes as sites that all open at the same instruction, each one's
            // first operand being the whole site before it. Nested, every level would merge "operand skipped" with
            // "operand evaluated", and the compiler fills the skipped side's registers with zeros level after level
            // (8 moves per level in the row-block kernel). Flat instead: the highest level whose FIRST operand is
            // irrelevant is where evaluation starts (scalar arithmetic on the mask), levels below it do nothing.
            std::vector<int> chain = {k};                    // outermost first
            for (;;) {
                const sdfk_cullsite& t = (*sites)[chain.back()];
                const int inner = site_spanning(t.a0, t.a1);
                if (inner < 0) break;
                chain.push_back(inner);
            }
            const size_t m = chain.size();
            // (inside [a0, a1] a site is out of reach of itself: its combiner lies beyond a1)
            if (m == 1) {
                const sdfk_cullsite& t = (*sites)[k];
                const std::string ta = bit_test(2 * k);
                // first operand (unless irrelevant), then — only if the second one matters — the second operand and
                // the combiner. Everything the second operand computes lives and dies inside that block, so a
                // skipped operand costs one scalar branch: no merge copies, no placeholder values.
                s += ind + "if (!" + ta + ") {\n";
                emit_span(t.a0, t.a1, depth + 1);
                s += ind + "}\n";
                emit_second(k, ind, ta, depth);
                i = t.comb + 1;
                continue;
            }
            // Long chains (an n-ary UNION of 50 children is a chain of 49 sites): walking every level's two mask bits
            // costs more scalar instructions than the one or two surviving children cost vector instructions. When the
            // sites of the chain are numbered consecutively and every level combines in place (what the lowering of
            // CombineGeometry produces), the levels that must run are a BITSET computed with a dozen scalar operations
            // — st = highest level whose first operand is irrelevant, alive = levels >= st whose second operand is
            // not skipped — and a loop visits exactly those: ctz, clear the bit, jump to the level's code.
            bool loopable = rows && m >= 4 && m <= 64 && chain[m - 1] + (int)m <= 64;   // (the loop's bitset: sites 0..63)
            for (size_t idx = 1; idx <= m && loopable; ++idx) {
                const sdfk_cullsite& t = (*sites)[chain[m - idx]];
                const uint32_t w = code[2 * t.comb];
                if (chain[m - idx] != chain[m - 1] + (int)idx - 1) loopable = false;            // consecutive site numbers
                if (((w >> 8) & 255u) != ((w >> 16) & 255u)) loopable = false;                 // V_a = op(V_a, V_c)
            }
            if (loopable) {
                char hd[256];
                snprintf(hd, sizeof hd, "#if SDFK_CHAIN_LOOP_MIN <= %zu\n", m);
                s += hd;
                emit_chain_loop(chain, ind, depth);
                s += "#else\n";
            }
            char st[32];
            snprintf(st, sizeof st, "st_%d", k);
            s += ind + "{ unsigned " + st + " = 0u;\n";
            for (size_t idx = 1; idx <= m; ++idx) {          // level idx = site chain[m - idx]
                char buf[96];
                snprintf(buf, sizeof buf, "if %s %s = %zuu;\n", bit_test(2 * chain[m - idx]).c_str(), st, idx);
                s += ind + "  " + buf;
            }
            const sdfk_cullsite& first = (*sites)[chain[m - 1]];
            s += ind + "  if (" + st + " == 0u) {\n";
            emit_span(first.a0, first.a1, depth + 1);
            s += ind + "  }\n";
            for (size_t idx = 1; idx <= m; ++idx) {
                char cond[64], repl[64];
                snprintf(cond, sizeof cond, "%s <= %zuu", st, idx);
                snprintf(repl, sizeof repl, "%s == %zuu", st, idx);
                s += ind + "  if (" + cond + ") {\n";
                emit_second(chain[m - idx], ind + "    ", repl, depth + 1);
                s += ind + "  }\n";
            }
            s += ind + "}\n";
            if (loopable) s += "#endif\n";
            i = (*sites)[k].comb + 1;
        }
    }

    void emit_chain_loop(const std::vector<int>& chain, const std::string& ind, int depth) {
        const size_t m = chain.size();
        const int k0 = chain[m - 1];                              // site of level 1; level idx = site k0 + idx - 1
        const sdfk_cullsite& first = (*sites)[k0];
        char buf[512];
        s += ind + "{ const unsigned long long cm0 = (unsigned long long)mw[0] | ((unsigned long long)mw[1] << 32), "
                   "cm1 = (unsigned long long)mw[2] | ((unsigned long long)mw[3] << 32);\n";
        // bit k of sa / sb = "skip the first / the second operand" of site k (sites 0..63)
        s += ind + "  unsigned long long sa = sdfk_even_bits(cm0) | (sdfk_even_bits(cm1) << 32), "
                   "sb = sdfk_even_bits(cm0 >> 1) | (sdfk_even_bits(cm1 >> 1) << 32);\n";
        snprintf(buf, sizeof buf, "  const unsigned long long lm = %lluull;\n", m >= 64 ? ~0ull : ((1ull << m) - 1ull));
        s += ind + buf;
        snprintf(buf, sizeof buf, "  sa = (sa >> %d) & lm; sb = (sb >> %d) & lm;\n", k0, k0);
        s += ind + buf;
        s += ind + "  const unsigned st = sa ? 64u - (unsigned)__builtin_clzll(sa) : 0u;\n";
        s += ind + "  unsigned long long alive = ~sb & lm & (st ? ~((1ull << (st - 1u)) - 1ull) : ~0ull);\n";
        s += ind + "  if (st == 0u) {\n";
        emit_span(first.a0, first.a1, depth + 1);
        s += ind + "  }\n";
        s += ind + "  while (alive) {\n";
        s += ind + "    const unsigned lvl = (unsigned)__builtin_ctzll(alive) + 1u;\n";
        s += ind + "    alive &= alive - 1ull;\n";
        s += ind + "    switch (lvl) {\n";
        for (size_t idx = 1; idx <= m; ++idx) {
            const sdfk_cullsite& t = (*sites)[chain[m - idx]];
            const uint32_t w = code[2 * t.comb];
            const unsigned a = (w >> 8) & 255u, c = w >> 24;
            char gB[64], gA[64], wx[32];
            bool neg;
            site_ops(t, gB, gA, wx, &neg, sizeof gB);
            snprintf(buf, sizeof buf, "      case %zuu: {\n", idx);
            s += ind + buf;
            emit_span(t.b0, t.b1, depth + 2);
            snprintf(buf, sizeof buf, "        if (lvl == st) { SDFK_EACH V_%u[q] = %sV_%u[q]; }\n", a, neg ? "-" : "", c);
            s += ind + buf;
            s += ind + "        else {\n";
            instr(t.comb, (ind + "          ").c_str(), true);
            s += ind + "        }\n";
            s += ind + "      } break;\n";
        }
        s += ind + "      default: break;\n";
        s += ind + "    }\n";
        s += ind + "  }\n";
        s += ind + "}\n";
    }

    void emit_rows_culled(int result_reg) {
        s += "\ntemplate <bool ZRUN> static __device__ __forceinline__ void sdfk_rows_culled(float X, float Y, "
             "const V3P (&P0)[SDFK_NP], const unsigned (&mw)[2 * SDFK_NMASK], "
             "const float* __restrict__ PRM, const float* __restrict__ TAB, f2 (&R)[SDFK_NP]) {\n"
             "    typedef f2 T;\n    V3P C_0[SDFK_NP];\n    SDFK_EACH C_0[q] = P0[q];\n";
        declare("V3P", "f2", true, true);
        emit_span(0, n_instr - 1, 0);
        char buf[96];
        snprintf(buf, sizeof buf, "    SDFK_EACH R[q] = V_%d[q];\n}\n", result_reg);
        s += buf;
    }

    void emit_culled(int result_reg) {
        s += "\ntemplate <typename T, bool ZRUN> static __device__ __forceinline__ T sdfk_point_culled(V3T<T> C_0, "
             "unsigned mlo, unsigned mhi, const float* bases, const float* __restrict__ PRM, "
             "const float* __restrict__ TAB) {\n";
        declare("V3T<T>", "T", false);
        emit_span(0, n_instr - 1, 0);
        char buf[64];
        snprintf(buf, sizeof buf, "    return V_%d;\n}\n", result_reg);
        s += buf;
    }
};

}  // namespace

// leaves from which an n-ary min / max chain is generated table-driven (SDFK_CHAIN_MIN overrides: tests). Measured on
// the 50-child flat union at 16385^2: the table-driven kernel runs as fast as the fully specialised one (1.08 vs 1.07 ms)
// and builds in 1.5 s instead of 12 s, so short chains go this way too; below ~16 children the specialised code wins.
// Round 4, after the level loop of the specialised code went (profiles/r04_chain_min.txt, left-deep hard unions of n mixed
// primitives at 513^3, specialised / table-driven): n = 20 0.47 / 0.54 ms, 24: 0.51 / 0.51, 27: 0.55 / 0.52, 30: 0.58 / 0.56,
// 48: 0.74 / 0.63 — the threshold moves from 17 to 22.
static size_t chain_min_leaves() {
    static const size_t v = [] {
        const char* e = getenv("SDFK_CHAIN_MIN");
        const long t = e ? atol(e) : 0;
        return (size_t)(t >= 2 ? t : 22);
    }();
    return v;
}
static bool chain_analyse(Gen& g, int result_reg, const std::vector<sdfk_cullsite>* sites_all) {
    if (!sites_all || sites_all->size() + 1 < chain_min_leaves()) return false;
    // The chain: the largest set of sites that share their first operand's start and their combiner (the in-place fold
    // of an n-ary min / max; analyse_chain checks the pattern). Other sites — combiners above the chain, with the chain
    // in one operand — belong to the REST of the program, which runs un-culled around the chain's value.
    std::map<std::pair<uint32_t, uint32_t>, std::vector<sdfk_cullsite>> by_start;
    for (const sdfk_cullsite& t : *sites_all) by_start[{t.a0, g.code[2 * t.comb] & 0xffffffu}].push_back(t);
    const std::vector<sdfk_cullsite>* best = nullptr;
    for (const auto& kv : by_start)
        if (!best || kv.second.size() > best->size()) best = &kv.second;
    // (at most 32768 leaves — the lowering keeps 32767 sites; round 3: 4096, when every leaf's value and list slot lived in
    //  LDS. With candidate lists per cell the workgroup holds 128 survivors per brick whatever the size of the chain.)
    if (!best || best->size() + 1 < chain_min_leaves() || best->size() + 1 > 32768) return false;
    g.chain_sites = *best;
    std::sort(g.chain_sites.begin(), g.chain_sites.end(), [](const sdfk_cullsite& x, const sdfk_cullsite& y) { return x.comb < y.comb; });
    for (const sdfk_cullsite& t : *sites_all) {                     // a site inside the chain that is not of the chain: no
        const bool ours = std::any_of(g.chain_sites.begin(), g.chain_sites.end(), [&](const sdfk_cullsite& c) { return c.comb == t.comb; });
        if (!ours && t.comb >= g.chain_sites.front().a0 && t.comb <= g.chain_sites.back().comb) return false;
    }
    const std::vector<sdfk_cullsite>* keep = g.sites;
    g.sites = &g.chain_sites;
    g.region_lo = g.chain_sites.front().a0;
    g.region_hi = g.chain_sites.back().comb;
    bool ok = g.analyse_leaves() && g.analyse_chain(result_reg) && g.leaves_contiguous();
    if (ok && g.groups.size() > 255) ok = false;                   // (a member travels as kind << 24 | first parameter)
    // (a member is compiled inline: beyond 4096 members the lowering drops the sites of the first levels, and what they
    //  combined becomes ONE member of hundreds of primitives — minutes of hiprtc; such programs stay where they were)
    for (size_t k = 0; ok && k < g.leaves.size(); ++k)
        if (g.leaves[k].hi - g.leaves[k].lo + 1 > 256) ok = false;
    if (!ok) {
        g.sites = keep;
        g.region_lo = 0;
        g.region_hi = (size_t)-1;
    }
    return ok;
}
int sdfk_chain_mode(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr, int result_reg,
                    const std::vector<sdfk_cullsite>& sites_all) {
    Gen g{ops, n_ops, code, n_instr, &sites_all, std::string()};
    return chain_analyse(g, result_reg, &sites_all) ? (int)g.leaves.size() : 0;
}

std::string sdfk_generate_source(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr,
                                 int result_reg, const std::vector<sdfk_cullsite>& sites, int flavour,
                                 const std::vector<sdfk_cullsite>* sites_all) {
    const bool all = flavour == SDFK_FL_ALL;
    const bool plain = all || flavour == SDFK_FL_PLAIN_ARRAY || flavour == SDFK_FL_PLAIN_GRID;
    const bool tile = !sites.empty() && (all || flavour == SDFK_FL_TILE_ARRAY || flavour == SDFK_FL_TILE_GRID ||
                                         flavour == SDFK_FL_TILE_MASK);
    const bool flat = flavour == SDFK_FL_ROWS2D_ARRAY || flavour == SDFK_FL_ROWS2D_GRID;
    const bool rowk = !sites.empty() && (all || flat || flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS_GRID ||
                                         flavour == SDFK_FL_ROWS_MASK);
    Gen g{ops, n_ops, code, n_instr, &sites, std::string()};
    g.s.reserve(sizeof(kEmbeddedDevice) + sizeof(kEmbeddedAccess) + sizeof(kWrappers) + sizeof(kTileKernel) +
                sizeof(kRowsKernel) + 800 * n_instr + 4096);
    g.s += kEmbeddedDevice;
    g.s += "\n";
    g.s += kEmbeddedAccess;
    char buf[64];
    if (chain_analyse(g, result_reg, sites_all)) {
        // chain mode: table-driven plain kernels and row-block kernels (no line-brick flavour)
        const bool flat2 = flavour == SDFK_FL_ROWS2D_ARRAY || flavour == SDFK_FL_ROWS2D_GRID;
        g.emit_chain_common();
        if (plain) {
            g.s += kWrappers;
            if (all || flavour == SDFK_FL_PLAIN_ARRAY) g.s += kWrappersArray;
            if (all || flavour == SDFK_FL_PLAIN_GRID) g.s += kWrappersGrid;
        }
        if (all || flat2 || flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS_MASK) {
            g.s += kWaveHelpers;
            if (flat2) g.s += "\n#define SDFK_FLAT 1\n";
            g.s += "\n#define SDFK_NP 4\n#define SDFK_EACH _Pragma(\"unroll\") for (int q = 0; q < SDFK_NP; ++q)\n"
                   "#define SDFK_NMASK 2\n"
                   "#define SDFK_SIMT 1\n#define SDFK_NSUB 1\n"
                   // candidate lists per cell (sdfk_spec_cells) for chains of more than 64 members; up to 64 the probe of every
                   // member by the whole workgroup stays (measured on the 50-child flat union at 16385^2: 0.88 ms against 1.04
                   // with the per-wave fold and 1.68 with lists — three dependent memory round trips per brick for a probe
                   // that costs 0.09 ms)
                   "#ifndef SDFK_CELLS_MIN_LEAVES\n#define SDFK_CELLS_MIN_LEAVES 64\n#endif\n"
                   "#if !defined(SDFK_NO_CELLS) && SDFK_NLEAF > SDFK_CELLS_MIN_LEAVES\n#define SDFK_CELLS 1\n#endif\n"
                   "#ifndef SDFK_LEAF_UNROLL\n#define SDFK_LEAF_UNROLL _Pragma(\"unroll 4\")\n#endif\n";
            g.s += kSimtGeometry;
            // (a lane per LEAF walking the centres pays off once there are more leaves than lanes; below that a lane per
            //  (leaf, centre) pair keeps more lanes busy: the 50-child flat union has 200 pairs for 128 lanes)
            g.emit_probe_leaves(g.leaves.size() >= 128, true);
            g.s += kRowsKernel;
            if (all || flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS2D_ARRAY) g.s += kRowsArray;
            if (all || flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS2D_GRID) g.s += kRowsGrid;
        }
        return g.s;
    }
    if (plain) {
        g.s += "\ntemplate <typename T> static __device__ __forceinline__ T sdfk_point(V3T<T> C_0, "
               "const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ AUX, "
               "long long AUXS) {\n";
        g.declare("V3T<T>", "T", false);
        for (size_t i = 0; i < n_instr; ++i) g.instr(i, "    ");
        snprintf(buf, sizeof buf, "    return V_%d;\n}\n", result_reg);
        g.s += buf;
        g.s += kWrappers;
        if (all || flavour == SDFK_FL_PLAIN_ARRAY) g.s += kWrappersArray;
        if (all || flavour == SDFK_FL_PLAIN_GRID) g.s += kWrappersGrid;
    }
    if (tile || rowk) g.s += kWaveHelpers;
    if (tile) {
        // the flat tile kernel carries 62 mask bits: the first 31 sites; the row-block kernel takes 64
        // (the 31 widest, in program order)
        std::vector<size_t> order(sites.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
            return sites[x].comb - sites[x].a0 > sites[y].comb - sites[y].a0;
        });
        order.resize(std::min<size_t>(order.size(), 31));
        std::sort(order.begin(), order.end());
        std::vector<sdfk_cullsite> flat;
        for (size_t i : order) flat.push_back(sites[i]);
        g.sites = &flat;
        g.find_roots();
        snprintf(buf, sizeof buf, "\n#define SDFK_NROOT %d\n", g.n_root > 0 ? g.n_root : 1);
        g.s += buf;
        g.emit_probe();
        g.emit_culled(result_reg);
        g.s += kTileKernel;
        if (all || flavour == SDFK_FL_TILE_ARRAY) g.s += kTileArray;
        if (all || flavour == SDFK_FL_TILE_GRID) g.s += kTileGrid;
        if (all || flavour == SDFK_FL_TILE_MASK) g.s += kTileMask;
        g.sites = &sites;
    }
    if (rowk) {
        g.rows = true;
        g.root_slot.clear();
        g.n_root = 0;
        g.find_roots();
        if (flat) g.s += "\n#define SDFK_FLAT 1\n";
        g.s += "\n#define SDFK_NP 4\n#define SDFK_EACH _Pragma(\"unroll\") for (int q = 0; q < SDFK_NP; ++q)\n";
        {   // skip bits of the sites: two per site, 32 sites per 64-bit word, at least the two words of rounds 1-3
            char nm[64];
            snprintf(nm, sizeof nm, "#define SDFK_NMASK %zu\n", std::max<size_t>(2, (sites.size() + 31) / 32));
            g.s += nm;
        }
        // (the loop over surviving levels — a bitset walked with ctz — paid for the 49-level union of cfg 4 in round 2, before
        //  chain mode took the long hard chains; on what is left for it, smooth chains of 13 .. 64 primitives and hard ones of
        //  13 .. 16, the plain level-after-level form is 7-11 % faster at every size tried: profiles/r04_chain_loop.txt. Off.)
        g.s += "#ifndef SDFK_CHAIN_LOOP_MIN\n#define SDFK_CHAIN_LOOP_MIN 100000   // chains of at least this many sites run as a loop over the levels that survive\n#endif\n"
               "// the even bits of x, packed into the low 32 bits (wave-uniform: scalar unit)\n"
               "static __device__ __forceinline__ unsigned long long sdfk_even_bits(unsigned long long x) {\n"
               "    x &= 0x5555555555555555ull;\n"
               "    x = (x | (x >> 1)) & 0x3333333333333333ull;\n"
               "    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;\n"
               "    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;\n"
               "    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;\n"
               "    x = (x | (x >> 16)) & 0x00000000ffffffffull;\n"
               "    return x;\n}\n";
        g.simt = g.analyse_leaves();
        if (g.simt) {
            // sub-bricks per brick (every one gets a probe centre of its own): as many as the lanes of the workgroup
            // pay for — the leaf evaluations are the cost: (leaves x centres) / lanes rounds of one leaf each
            char def[160];
            snprintf(def, sizeof def, "\n#ifndef SDFK_NOSIMT\n#define SDFK_SIMT 1\n#endif\n#define SDFK_NLEAF %zu\n", g.leaves.size());
            g.s += def;
            g.s += kSimtGeometry;
            g.emit_groups();
            g.emit_probe_leaves();
            g.emit_probe_fold();
        }
        g.emit_probe();
        g.emit_rows_culled(result_reg);
        g.s += kRowsKernel;
        if (all || flavour == SDFK_FL_ROWS_ARRAY || flavour == SDFK_FL_ROWS2D_ARRAY) g.s += kRowsArray;
        if (all || flavour == SDFK_FL_ROWS_GRID || flavour == SDFK_FL_ROWS2D_GRID) g.s += kRowsGrid;
        if (all || flavour == SDFK_FL_ROWS_MASK) g.s += kRowsMask;
    }
    return g.s;
}
