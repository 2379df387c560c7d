// sdfk_access.h — how a kernel reads its points and writes the field. Shared (like sdfk_device.h)
// by the interpreter kernel and, as embedded text, by every hiprtc-specialised kernel.
//
// HBM layout: coordinates are a C-contiguous (3, N) fp32 array (row 0/1/2 = x/y/z, row pitch
// `stride` elements), exactly the reference's `co` (C/helper_functions.py:90-91) in fp32; the
// field is N fp32. One thread owns VEC consecutive points: with VEC = 4 a wave reads 1 KiB per
// row per instruction (global_load_dwordx4, fully coalesced) and writes 1 KiB.
#ifndef SDFK_ACCESS_H
#define SDFK_ACCESS_H

#define SDFK_BLOCK 256

struct SrcArray {  // (3, n) array resident in HBM
    const float* __restrict__ co;
    long long stride;
};
struct SrcGrid {  // regular grid expanded on the fly from three per-axis tables (generate_grid on device)
    const float* __restrict__ ax0;
    const float* __restrict__ ax1;
    const float* __restrict__ ax2;
    unsigned n1, n2;
    long long start;  // flat index of point 0 of this launch:  n = (ix*n1 + iy)*n2 + iz
};

// Coordinates and the field are touched exactly once per evaluation: non-temporal loads and stores (the
// streaming rate of a (3,N)->(N) pass on MI355X rises from 5.8 to 6.2 TB/s, tools/rowstream.hip).
typedef float sdfk_f4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float4 sdfk_stream_load4(const float* p) {
    const sdfk_f4 v = __builtin_nontemporal_load(reinterpret_cast<const sdfk_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
static __device__ __forceinline__ void sdfk_stream_store4(float* p, float a, float b, float c, float d) {
    const sdfk_f4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<sdfk_f4*>(p));
}

// block_base : index of the workgroup's first point (wave-uniform), lane_off : sdfk_tx() * VEC
template <int VEC>
static __device__ __forceinline__ void sdfk_load(const SrcArray& s, long long block_base, unsigned lane_off,
                                                 V3 (&p)[VEC]) {
    const long long i = block_base + lane_off;
    if constexpr (VEC == 4) {
        const float4 x = sdfk_stream_load4(s.co + i);
        const float4 y = sdfk_stream_load4(s.co + s.stride + i);
#ifdef SDFK_XY                                                   // two-row coordinates: z = 0 by contract, no third row
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
        const float4 z = sdfk_stream_load4(s.co + 2 * s.stride + i);
#endif
        p[0] = {x.x, y.x, z.x};
        p[1] = {x.y, y.y, z.y};
        p[2] = {x.z, y.z, z.z};
        p[3] = {x.w, y.w, z.w};
    } else {
#pragma unroll
#ifdef SDFK_XY
        for (int v = 0; v < VEC; ++v) p[v] = {s.co[i + v], s.co[s.stride + i + v], 0.0f};
#else
        for (int v = 0; v < VEC; ++v) p[v] = {s.co[i + v], s.co[s.stride + i + v], s.co[2 * s.stride + i + v]};
#endif
    }
}

// The 64-bit divisions act on wave-uniform values only (scalar unit, once per workgroup); the
// per-lane part is 32-bit.
template <int VEC>
static __device__ __forceinline__ void sdfk_load(const SrcGrid& s, long long block_base, unsigned lane_off,
                                                 V3 (&p)[VEC]) {
    const unsigned long long base = (unsigned long long)(s.start + block_base);
    const unsigned long long row = base / s.n2;
    const unsigned iz0 = (unsigned)(base - row * s.n2);
    const unsigned long long ix0 = row / s.n1;
    const unsigned iy0 = (unsigned)(row - ix0 * s.n1);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const unsigned t = iz0 + lane_off + v;
        const unsigned cz = t / s.n2;
        const unsigned iz = t - cz * s.n2;
        const unsigned ty = iy0 + cz;
        const unsigned cy = ty / s.n1;
        const unsigned iy = ty - cy * s.n1;
        p[v] = {s.ax0[ix0 + cy], s.ax1[iy], s.ax2[iz]};
    }
}

// ---- flags instead of the field (fused interior selection, GenericGeometry.point_cloud: C/geom.py:62-74) -----------
// An evaluation kernel that is handed a flag buffer writes ONE BIT per point — bit j of byte b = point 8 b + j,
// "field <= threshold" — and no field: 12 B/point of coordinates in, 1/8 B/point out. The compaction
// (sdfk_fieldops.inc: count, scan, scatter from the flags) then never sees the field either.
// Selection key: an unsigned integer that orders like the float (negative zero = zero), every NaN above all numbers so
// that it is never selected (NumPy: nan <= t is False). Integer compares only: the kernels are built with
// -fno-honor-nans, which leaves float compares with NaN undefined.
static __host__ __device__ __forceinline__ unsigned sdfk_sel_key(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN
    if (u == 0x80000000u) u = 0u;                             // -0.0 == 0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// VEC points per lane starting at flat index i (i % VEC == 0; a wave covers 64 * VEC consecutive points, its first one
// at a multiple of 64 * VEC): the lanes' VEC-bit groups are OR-ed into 32-bit words inside the wave (VEC = 4: 8 lanes per
// word, three exchange steps) and stored by one lane per word. nib = 0 for lanes past the end of the array — they must
// still get here (the exchange is wave-wide).
template <int VEC>
static __device__ __forceinline__ void sdfk_store_flags(unsigned* __restrict__ flags, long long i, unsigned nib, bool active) {
    if constexpr (VEC == 4) {
        const unsigned lane = __builtin_amdgcn_workitem_id_x() & 63u;
        unsigned w = nib << (4u * (lane & 7u));
        w |= __shfl_xor(w, 1);
        w |= __shfl_xor(w, 2);
        w |= __shfl_xor(w, 4);
        if ((lane & 7u) == 0u && active) flags[i >> 5] = w;
    } else {
        if (active && nib) atomicOr(&flags[i >> 5], nib << (unsigned)(i & 31));   // single points (tails, unaligned arrays)
    }
}

template <int VEC>
static __device__ __forceinline__ void sdfk_store(float* __restrict__ out, long long i, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        sdfk_stream_store4(out + i, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[i + k] = v[k];
    }
}

#endif  // SDFK_ACCESS_H
