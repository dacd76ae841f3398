// common.h -- shared device/host helpers for libmusev_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/musev_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MV_WAVE 64

// makes a VGPR value opaque to the optimiser at this point (an empty asm that "modifies" it): used to keep a per-tile LDS address
// in ONE register with immediate offsets.  The host simulator of tests/ defines it away before including this header.
#ifndef MV_KEEP_ONE_REGISTER
#define MV_KEEP_ONE_REGISTER(x) asm volatile("" : "+v"(x))
#endif

// marks the lanes a divergent region switches off for an LDS-DMA instruction: nothing on the GPU (EXEC does it); the host
// simulator of tests/ books the wave's instruction for those lanes too, so that their counted waits stay in step
#ifndef MV_DMA_LANE_OFF
#define MV_DMA_LANE_OFF() ((void)0)
#endif

// a * b + c as ONE scalar v_fma_f32 the optimiser cannot merge into a packed-fp32 instruction.  Used for the row affine of the
// LayerNorm-folded GEMM epilogue: hipcc packed that expression into v_pk_mul_f32 / v_pk_fma_f32 with op_sel operand swizzles
// (the per-row scalars of two 16-row passes share a register pair), and on the MI355X the LOW results of exactly those
// instructions came out wrong sporadically on lanes 48-63 (profiles/r03d, r03e: run-to-run different, always pass 1 / tile 0 /
// elements 0 and 2).  The host simulator of tests/ defines it as fmaf.
#ifndef MV_FMA_SCALAR
__device__ __forceinline__ float mv_fma_scalar(float a, float b, float c) {
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#define MV_FMA_SCALAR(a, b, c) mv_fma_scalar((a), (b), (c))
#endif

// host-side error plumbing ---------------------------------------------------------------------------
void mv_set_error(const char* fmt, ...);

#define MV_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            mv_set_error(__VA_ARGS__);   \
            return MV_ERR_INVALID;       \
        }                                \
    } while (0)

#define MV_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            mv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return MV_ERR_LAUNCH;                                                   \
        }                                                                           \
    } while (0)

// device helpers -------------------------------------------------------------------------------------
// x * sigmoid(x); the reciprocal is the hardware v_rcp_f32 (1 ulp) -- results are rounded to fp16 anyway
__device__ __forceinline__ float mv_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) gelu, matching torch.nn.functional.gelu default used by diffusers GEGLU
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of the result): one exp, one rcp and
// five fma instead of libm's branchy erff -- the GEGLU gate is evaluated 4C times per token in the FF1 epilogue
__device__ __forceinline__ float mv_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float y = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float mv_gelu(float x) { return 0.5f * x * (1.0f + mv_erf(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: blocks that the dispatcher places on the same XCD
// (observed: id % 8) get a contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__host__ __device__ __forceinline__ int mv_xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, k = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Logical id -> output tile.  An XCD runs a sliding window of ~64 consecutive logical ids (32 CUs x 2 resident blocks);
// what its 4 MB L2 has to fetch for that window is (distinct m-tiles) x BM x K + (distinct n-tiles) x BN x K halfs.  With
// plain m-major order and a wide N (the GEGLU projections: 20 / 40 / 80 n-tiles; fused QKV: 24) a window is 1-3 m-tiles x
// ALL n-tiles, i.e. every window streams most of the weight matrix again.  Groups of MV_TILE_GROUP m-tiles walked m-fastest
// make the window ~8 x 8 tiles instead (group = MV_TILE_GROUP; 0 / 1 = plain m-major, the A/B knob) (the fetch per window drops from (1 + tiles_n) to 16 tile-slabs).  For tiles_n <= 8 the
// m-major window already is 8 m-tiles (or more) x tiles_n, and the order is left exactly as it was.  Any order is a
// bijection onto the tile grid, each tile is still reduced over K in the same order by one block: results do not change.
#define MV_TILE_GROUP 8
__host__ __device__ __forceinline__ void mv_tile_order(int id, int tiles_m, int tiles_n, int group, int* tile_m, int* tile_n) {
    if (group < 0) {  // weight-stationary (mv_gemm_desc.tile_order): n-major, a contiguous id range = a few n-tiles x ALL m-tiles
        const int tn = id / tiles_m;
        *tile_n = tn;
        *tile_m = id - tn * tiles_m;
        return;
    }
    if (group <= 1 || tiles_n <= group) {
        const int tm = id / tiles_n;
        *tile_m = tm;
        *tile_n = id - tm * tiles_n;
        return;
    }
    const int per_group = group * tiles_n;
    const int grp = id / per_group;
    const int first_m = grp * group;
    const int rest = tiles_m - first_m;
    const int gsz = rest < group ? rest : group;  // the last group may be shorter
    const int r = id - grp * per_group;
    const int tn = r / gsz;
    *tile_m = first_m + (r - tn * gsz);
    *tile_n = tn;
}
