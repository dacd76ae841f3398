// gemm.hip -- fp16 implicit-GEMM family for gfx950 (MFMA 16x16x32 f16, fp32 accumulate).
//
// One kernel template covers the three contraction shapes of the UNet3D step:
//   MV_GEMM_LINEAR  : Linear / 1x1 conv               (K3)
//   MV_GEMM_CONV3X3 : NHWC 3x3 conv, pad 1, stride 1|2, optional fused nearest-x2 upsample   (K2, K9)
//   MV_GEMM_TCONV3  : Conv3d (3,1,1), pad (1,0,0) over a [B,T,HW,C] tensor                    (K4)
// all with an optional second channel-concatenated source (K10: the up-path torch.cat is never
// materialised) and a fused epilogue: bias, per-frame row bias (time / frame embedding), |alpha| scale
// (temporal_weight), SiLU, residual add, or the GEGLU gate.
//
// Tiling (CDNA4, 64-wide waves): block = 256 threads = 2x2 waves; wave tile = (16*TM) x (16*TN) built from
// 16x16x32 MFMAs; BK = 64.  Both operands are K-contiguous ("B^T" form: activations [m][k], weights [n][k]),
// so every MFMA fragment is one 16-byte ds_read_b128.  LDS tiles are row-major with 128-byte rows and the
// 16-byte slot index XOR-ed with (row & 7): conflict-free for the ds_read_b128 lane groups and for the
// 8-lane ds_write_b128 groups (see cdna_hip_programming.md T2).  Global->LDS staging goes through registers
// (issue loads for tile k+1, run the MFMAs of tile k, then write tile k+1: one barrier per K step) because the
// conv gathers need per-row predication (zero halo) that an LDS-DMA cannot express.
// The MFMA operands are swapped (weights as the "A" operand) so that each lane ends up holding four
// consecutive output channels of one output row -> 8-byte stores and 8-byte bias/residual loads.
#include "common.h"

namespace {

struct GemmArgs {
    const half_t* a;
    const half_t* a2;
    const half_t* w;
    half_t* c;
    const half_t* bias;
    const half_t* rowbias;
    const half_t* residual;
    const float* alpha;
    long M;
    int N, K;
    int lda, lda2, ldc, ldr, ldrb;
    int c1, cin;  // cin = c1 + c2
    int stride, upsample, hin, win, hout, wout;
    int t, hw;
    int rows_per_group, act, geglu;
    int tiles_m, tiles_n;
    int tile_group;  // mv_tile_order group size (0: m-major)
};

__device__ __attribute__((aligned(16))) uint4 g_zero_page[4];  // 64 zero bytes: target of predicated-off loads


template <int TM>
struct RowInfo {
    bool ok[TM];
    long base[TM];
    int y[TM], x[TM];
};

// Issue the global loads of K tile `kt` into registers (no wait).  Predicated-off elements read the zero page,
// so the loads are unconditional and the compiler keeps them in flight across the MFMA block.
template <int MODE, int TM, int TN>
__device__ __forceinline__ void load_tiles(const GemmArgs& p, const RowInfo<TM>& ri, u32x4 (&ra)[TM], u32x4 (&rb)[TN],
                                           int kt, int& kc, int& tap, int n0, int lrow, int lslot,
                                           const half_t* zero) {
    constexpr int BK = 64;
    // ---- A operand (activations) ----
    const bool second = (p.a2 != nullptr) && (kc >= p.c1);
    const half_t* src = second ? p.a2 : p.a;
    const long ld = second ? p.lda2 : p.lda;
    const long coff = (second ? kc - p.c1 : kc) + lslot * 8;
    const long zdelta = zero - src;  // element offset that redirects a load to the zero page
    const bool kok = (MODE != MV_GEMM_LINEAR) || (kc + lslot * 8 < p.cin);
    int dy = 0, dx = 0;
    if (MODE == MV_GEMM_CONV3X3) {
        dy = tap / 3 - 1;
        dx = tap - (tap / 3) * 3 - 1;
    } else if (MODE == MV_GEMM_TCONV3) {
        dy = tap - 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool ok = ri.ok[i] && kok;
        long row;
        if (MODE == MV_GEMM_LINEAR) {
            row = ri.base[i];
        } else if (MODE == MV_GEMM_CONV3X3) {
            int iy = ri.y[i] + dy, ix = ri.x[i] + dx;
            if (p.upsample) {
                ok = ok && iy >= 0 && iy < 2 * p.hin && ix >= 0 && ix < 2 * p.win;
                iy >>= 1;
                ix >>= 1;
            } else {
                ok = ok && iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win;
            }
            row = ri.base[i] + (long)(iy * p.win + ix);
        } else {
            int tt = ri.y[i] + dy;
            ok = ok && tt >= 0 && tt < p.t;
            row = ri.base[i] + (long)dy * p.hw;
        }
        long off = row * ld + coff;
        off = ok ? off : zdelta;
        ra[i] = *reinterpret_cast<const u32x4*>(src + off);
    }
    // ---- B operand (weights [N][K]) ----
    const int kg = kt * BK + lslot * 8;
    const bool wk_ok = kg < p.K;
    const long wz = zero - p.w;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + lrow + 32 * j;
        bool ok = wk_ok && n < p.N;
        long off = (long)n * p.K + kg;
        off = ok ? off : wz;
        rb[j] = *reinterpret_cast<const u32x4*>(p.w + off);
    }
    // advance the (tap, channel) cursor by one K tile
    kc += BK;
    if (MODE != MV_GEMM_LINEAR && kc >= p.cin) {
        kc -= p.cin;
        ++tap;
    }
}

template <int TM, int TN>
__device__ __forceinline__ void store_tiles(half_t* dA, half_t* dB, const u32x4 (&ra)[TM], const u32x4 (&rb)[TN],
                                            int lrow, int sw_off) {
    constexpr int BK = 64;
#pragma unroll
    for (int i = 0; i < TM; ++i) *reinterpret_cast<u32x4*>(dA + (lrow + 32 * i) * BK + sw_off) = ra[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) *reinterpret_cast<u32x4*>(dB + (lrow + 32 * j) * BK + sw_off) = rb[j];
}

template <int TM, int TN>
__device__ __forceinline__ void mma_tile(const half_t* cA, const half_t* cB, float4v (&acc)[TM][TN], int a_row0,
                                         int b_row0, int swz, int g) {
    constexpr int BK = 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int slot_off = (((kk * 4 + g) ^ swz) << 3);
        half8v af[TM], wf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[i] = *reinterpret_cast<const half8v*>(cA + (a_row0 + 16 * i) * BK + slot_off);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            wf[j] = *reinterpret_cast<const half8v*>(cB + (b_row0 + 16 * j) * BK + slot_off);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
}


// ---- LDS-DMA staging variant -------------------------------------------------------------------------------------
// global_load_lds_dwordx4 writes LDS at (wave-uniform base + lane*16): one wave instruction fills 8 rows x 128 B.
// The XOR swizzle therefore moves to the SOURCE side: lane l of chunk c owns row 8c + l/8 and LDS slot l%8, and fetches
// the logical slot (l%8) ^ (l/8) of that row -- still one full 128-byte line per 8 lanes.  Predicated-off lanes (conv
// halo, ragged M/N/K) point at the zero page, which an LDS-DMA can express because the source address is per lane.
template <int MODE, int TM, int TN>
__device__ __forceinline__ void issue_tiles(const GemmArgs& p, const RowInfo<TM>& ri, half_t* dA, half_t* dB, int kt,
                                            int& kc, int& tap, int n0, int wave, int lane, const half_t* zero) {
    constexpr int BK = 64;
    const int lslot = (lane & 7) ^ (lane >> 3);  // logical 16-byte slot this lane fetches
    const bool second = (p.a2 != nullptr) && (kc >= p.c1);
    const half_t* src = second ? p.a2 : p.a;
    const long ld = second ? p.lda2 : p.lda;
    const long coff = (second ? kc - p.c1 : kc) + lslot * 8;
    const long zdelta = zero - src;
    const bool kok = (MODE != MV_GEMM_LINEAR) || (kc + lslot * 8 < p.cin);
    int dy = 0, dx = 0;
    if (MODE == MV_GEMM_CONV3X3) {
        dy = tap / 3 - 1;
        dx = tap - (tap / 3) * 3 - 1;
    } else if (MODE == MV_GEMM_TCONV3) {
        dy = tap - 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool ok = ri.ok[i] && kok;
        long row;
        if (MODE == MV_GEMM_LINEAR) {
            row = ri.base[i];
        } else if (MODE == MV_GEMM_CONV3X3) {
            int iy = ri.y[i] + dy, ix = ri.x[i] + dx;
            if (p.upsample) {
                ok = ok && iy >= 0 && iy < 2 * p.hin && ix >= 0 && ix < 2 * p.win;
                iy >>= 1;
                ix >>= 1;
            } else {
                ok = ok && iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win;
            }
            row = ri.base[i] + (long)(iy * p.win + ix);
        } else {
            int tt = ri.y[i] + dy;
            ok = ok && tt >= 0 && tt < p.t;
            row = ri.base[i] + (long)dy * p.hw;
        }
        long off = row * ld + coff;
        off = ok ? off : zdelta;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                         (__attribute__((address_space(3))) void*)(dA + (wave * TM + i) * (8 * BK)), 16, 0, 0);
    }
    const int kg = kt * BK + lslot * 8;
    const bool wk_ok = kg < p.K;
    const long wz = zero - p.w;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + (wave * TN + j) * 8 + (lane >> 3);
        bool ok = wk_ok && n < p.N;
        long off = (long)n * p.K + kg;
        off = ok ? off : wz;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + off),
                                         (__attribute__((address_space(3))) void*)(dB + (wave * TN + j) * (8 * BK)), 16, 0, 0);
    }
    kc += BK;
    if (MODE != MV_GEMM_LINEAR && kc >= p.cin) {
        kc -= p.cin;
        ++tap;
    }
}

template <int MODE, int TM, int TN, int STAGE>  // STAGE 0: register staging, 1: LDS-DMA (global_load_lds)
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs p) {
    constexpr int BM = 32 * TM, BN = 32 * TN, BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = reinterpret_cast<half_t*>(smem);  // [2][BM*BK]
    half_t* sB = sA + 2 * BM * BK;                  // [2][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int id = mv_xcd_remap(blockIdx.x, nwg);
    const int tile_m = id / p.tiles_n, tile_n = id - tile_m * p.tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    const half_t* zero = reinterpret_cast<const half_t*>(g_zero_page);

    // ---- loader geometry: thread -> (row = tid/8 + 32*pass, 16-byte slot = tid%8) ----
    const int lrow = tid >> 3, lslot = tid & 7;
    const int sw_off = ((lslot ^ (lrow & 7)) << 3);  // swizzled slot, in halfs

    RowInfo<TM> ri;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        long gm = (STAGE == 0) ? (m0 + lrow + 32 * i) : (m0 + (wave * TM + i) * 8 + (lane >> 3));
        ri.ok[i] = gm < p.M;
        if (MODE == MV_GEMM_LINEAR) {
            ri.base[i] = gm;
            ri.y[i] = ri.x[i] = 0;
        } else if (MODE == MV_GEMM_CONV3X3) {
            int hwo = p.hout * p.wout;
            long n = gm / hwo;
            int rem = (int)(gm - n * hwo);
            int oy = rem / p.wout, ox = rem - oy * p.wout;
            ri.base[i] = n * (long)(p.hin * p.win);
            ri.y[i] = oy * p.stride;
            ri.x[i] = ox * p.stride;
        } else {
            ri.base[i] = gm;
            ri.y[i] = (int)((gm / p.hw) % p.t);
            ri.x[i] = 0;
        }
    }

    u32x4 ra[TM], rb[TN];
    int kc = 0, tap = 0;  // channel offset inside the current tap, tap index (uniform)

    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    // fragment read offsets (halfs): row = tile base + l15 ; slot = kk*4 + g ; swizzle with row & 7 == l15 & 7
    const int a_row0 = wm * 16 * TM + l15;
    const int b_row0 = wn * 16 * TN + l15;
    const int swz = l15 & 7;
    if constexpr (STAGE == 0) {
        load_tiles<MODE, TM, TN>(p, ri, ra, rb, 0, kc, tap, n0, lrow, lslot, zero);
        store_tiles<TM, TN>(sA, sB, ra, rb, lrow, sw_off);
        __syncthreads();
        for (int kt = 0; kt < nk - 1; ++kt) {
            const int cur = kt & 1;
            load_tiles<MODE, TM, TN>(p, ri, ra, rb, kt + 1, kc, tap, n0, lrow, lslot, zero);
            mma_tile<TM, TN>(sA + cur * (BM * BK), sB + cur * (BN * BK), acc, a_row0, b_row0, swz, g);
            store_tiles<TM, TN>(sA + (cur ^ 1) * (BM * BK), sB + (cur ^ 1) * (BN * BK), ra, rb, lrow, sw_off);
            __syncthreads();
        }
    } else {
        issue_tiles<MODE, TM, TN>(p, ri, sA, sB, 0, kc, tap, n0, wave, lane, zero);
        __syncthreads();  // the compiler drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        for (int kt = 0; kt < nk - 1; ++kt) {
            const int cur = kt & 1;
            issue_tiles<MODE, TM, TN>(p, ri, sA + (cur ^ 1) * (BM * BK), sB + (cur ^ 1) * (BN * BK), kt + 1, kc, tap, n0,
                                      wave, lane, zero);
            mma_tile<TM, TN>(sA + cur * (BM * BK), sB + cur * (BN * BK), acc, a_row0, b_row0, swz, g);
            __syncthreads();
        }
    }
    {
        const int cur = (nk - 1) & 1;
        mma_tile<TM, TN>(sA + cur * (BM * BK), sB + cur * (BN * BK), acc, a_row0, b_row0, swz, g);
    }

    // ---- epilogue: lane holds out[m = .. + l15][n = .. + 4g + {0..3}] ----
    const float alpha = p.alpha ? fabsf(*p.alpha) : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * 16 * TM + 16 * i + l15;
        if (m >= p.M) continue;
        const long grp = p.rowbias ? (m / p.rows_per_group) : 0;
        if (!p.geglu) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * 16 * TN + 16 * j + 4 * g;
                if (n >= p.N) continue;
                float4v v = acc[i][j];
                if (p.bias) {
                    half4v b = *reinterpret_cast<const half4v*>(p.bias + n);
                    v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                }
                if (p.rowbias) {
                    half4v b = *reinterpret_cast<const half4v*>(p.rowbias + grp * p.ldrb + n);
                    v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                }
                v *= alpha;
                if (p.act == MV_ACT_SILU) {
                    v[0] = mv_silu(v[0]); v[1] = mv_silu(v[1]); v[2] = mv_silu(v[2]); v[3] = mv_silu(v[3]);
                }
                if (p.residual) {
                    half4v r = *reinterpret_cast<const half4v*>(p.residual + m * p.ldr + n);
                    v += float4v{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                }
                half4v o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4v*>(p.c + m * p.ldc + n) = o;
            }
        } else {
            // packed rows: [16 value | 16 gate] per 32; even tile = value, odd tile = gate
            if constexpr ((TN & 1) == 0) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int nb = n0 + wn * 16 * TN + 16 * j;  // packed column of the value tile
                    if (nb >= p.N) continue;
                    float4v v = acc[i][j], gt = acc[i][j + 1];
                    if (p.bias) {
                        half4v b = *reinterpret_cast<const half4v*>(p.bias + nb + 4 * g);
                        half4v bg = *reinterpret_cast<const half4v*>(p.bias + nb + 16 + 4 * g);
                        v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                        gt += float4v{(float)bg[0], (float)bg[1], (float)bg[2], (float)bg[3]};
                    }
                    half4v o = {(half_t)(v[0] * mv_gelu(gt[0])), (half_t)(v[1] * mv_gelu(gt[1])),
                                (half_t)(v[2] * mv_gelu(gt[2])), (half_t)(v[3] * mv_gelu(gt[3]))};
                    *reinterpret_cast<half4v*>(p.c + m * p.ldc + (nb >> 1) + 4 * g) = o;
                }
            }
        }
    }
}

template <int MODE, int TM, int TN, int STAGE>
int launch_cfg_s(const GemmArgs& a0, hipStream_t stream) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int smem = 2 * (BM + BN) * 64 * (int)sizeof(half_t);
    GemmArgs a = a0;
    a.tiles_m = (int)((a.M + BM - 1) / BM);
    a.tiles_n = (a.N + BN - 1) / BN;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<MODE, TM, TN, STAGE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            mv_set_error("mv_gemm_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return MV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n));
    hipLaunchKernelGGL((gemm_kernel<MODE, TM, TN, STAGE>), grid, dim3(256), smem, stream, a);
    MV_CHECK_LAUNCH("mv_gemm_f16");
    return MV_OK;
}

int g_gemm_tile_group = MV_TILE_GROUP;  // mv_tile_order group size of the v2 kernel (mv_set_gemm_tile_group; 0 = m-major)
int g_gemm_stage = 2;  // tuning knob (mv_set_gemm_variant): 0 register staging, 1 LDS-DMA, 2 v2 kernel, 3 v2 + 8-wave tiles,
                       // 4 persistent v3 kernel, 5 v2 + 8-wave tiles on a 3-stage counted-wait ring, 6 BK-32 4-stage ring,
                       // 7 256x320 tiles wherever they fit, 8 = 2 + 256x256 tiles for the GEGLU GEMM

template <int MODE, int TM, int TN>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    return g_gemm_stage == 0 ? launch_cfg_s<MODE, TM, TN, 0>(a, stream) : launch_cfg_s<MODE, TM, TN, 1>(a, stream);  // 1: also the fallback of v2
}


// =====================================================================================================================
// v2 kernel: same tiling / LDS image / MFMA schedule as above, with the per-K-step address arithmetic removed.
//   * global->LDS copies are buffer_load_dwordx4 ... lds through three buffer descriptors (source 1, source 2,
//     weights): a lane's byte offset is a 32-bit VGPR that only changes when the conv tap (or the concat source)
//     changes; the K advance is the scalar soffset.  Predicated-off lanes (conv halo, ragged M/N/K) carry the offset
//     0x80000000, which is out of range for every descriptor (sizes are checked < 2 GiB on the host) and therefore
//     reads as zero -- no zero page, no per-lane pointer selects.
//   * the weight rows of every pair of 16-row MFMA tiles are interleaved in LDS so that a lane ends up holding 8
//     CONSECUTIVE output channels of its row: 16-byte epilogue stores / residual / bias loads (64 contiguous bytes per
//     row per wave instruction instead of 32).
//   * block shape is a template parameter (WGM x WGN waves): 2x2 waves (2 blocks per CU) or 4x2 waves (one
//     512-thread block per CU, A/B tiles shared by twice as many waves).
constexpr unsigned kOOB = 0x80000000u;

struct GemmArgs2 {
    GemmArgs g;
    unsigned a_bytes, a2_bytes, w_bytes;
    int wide;  // 1: N, ldc, ldr, ldrb multiples of 8 and 16-byte aligned pointers -> interleaved tiles + 16-byte epilogue
};

// narrow epilogue (N or a leading dimension not a multiple of 8, or unaligned pointers): 8-byte accesses straight from
// the accumulator layout (lane = one row, 4 channels per 16-wide tile)
template <int TM, int TN>
__device__ __forceinline__ void epilogue_narrow(const GemmArgs& p, float4v (&acc)[TM][TN], int mw0, int nw0, int lane, float alpha,
                                                int Mi) {
    const int l15 = lane & 15, g = lane >> 4;
    if (p.geglu) {
        if constexpr ((TN & 1) == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = mw0 + 16 * i + l15;
                if (m >= Mi) continue;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int nb = nw0 + 16 * j;  // packed column of the value tile
                    if (nb >= p.N) continue;
                    float4v v = acc[i][j], gt = acc[i][j + 1];
                    if (p.bias) {
                        half4v b = *reinterpret_cast<const half4v*>(p.bias + nb + 4 * g);
                        half4v bg = *reinterpret_cast<const half4v*>(p.bias + nb + 16 + 4 * g);
                        v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                        gt += float4v{(float)bg[0], (float)bg[1], (float)bg[2], (float)bg[3]};
                    }
                    half4v o = {(half_t)(v[0] * mv_gelu(gt[0])), (half_t)(v[1] * mv_gelu(gt[1])),
                                (half_t)(v[2] * mv_gelu(gt[2])), (half_t)(v[3] * mv_gelu(gt[3]))};
                    *reinterpret_cast<half4v*>(p.c + (long)m * p.ldc + (nb >> 1) + 4 * g) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw0 + 16 * i + l15;
        if (m >= Mi) continue;
        const long grp = p.rowbias ? (m / p.rows_per_group) : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nw0 + 16 * j + 4 * g;
            if (n >= p.N) continue;
            float4v v = acc[i][j];
            if (p.bias) {
                half4v b = *reinterpret_cast<const half4v*>(p.bias + n);
                v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
            if (p.rowbias) {
                half4v b = *reinterpret_cast<const half4v*>(p.rowbias + grp * p.ldrb + n);
                v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
            v *= alpha;
            if (p.act == MV_ACT_SILU) {
                v[0] = mv_silu(v[0]); v[1] = mv_silu(v[1]); v[2] = mv_silu(v[2]); v[3] = mv_silu(v[3]);
            }
            if (p.residual) {
                half4v r = *reinterpret_cast<const half4v*>(p.residual + (long)m * p.ldr + n);
                v += float4v{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            }
            half4v o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4v*>(p.c + (long)m * p.ldc + n) = o;
        }
    }
}

// LDS-staged epilogue of one wave's (16*TM) x (16*TN) accumulator tile (see the call site).  GEGLU: even tiles hold
// values, odd tiles gates; the output has 8*TN columns per wave.  Each wave owns a private staging region, so only
// wave-level ordering is needed between its write and read phases.
template <int TM, int TN, bool GEGLU, int IT>  // IT = 16-row tiles staged per pass
__device__ __forceinline__ void epilogue_staged(const GemmArgs& p, float4v (&acc)[TM][TN], float* stg_base, int wave, int mw0,
                                                int nw0, int lane, float alpha, int Mi) {
    constexpr int W = GEGLU ? 8 * TN : 16 * TN;  // output columns of the wave tile
    constexpr int LD = W + 4;                    // floats; +4 keeps the 16-byte row-strided writes conflict-free
    constexpr int CPR = W / 8;                   // 8-column chunks per row
    constexpr int ROWS = 16 * IT;
    constexpr int KI = (ROWS * CPR + 63) / 64;   // read iterations per pass (the last one may be partial)
    static_assert(TM % IT == 0, "passes must tile the wave rows");
    float* stg = stg_base + wave * (ROWS * LD);
    const int l15 = lane & 15, g = lane >> 4;
    const int Nout = GEGLU ? (p.N >> 1) : p.N;

    // per-column terms in the accumulator layout
    float4v bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = float4v{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // GEGLU: packed bias index of tile j is (2 * nw0 + 16 j + 4 g); plain: nw0 + 16 j + 4 g
            const int n = (GEGLU ? 2 * nw0 : nw0) + 16 * j + 4 * g;
            if (n < p.N) {
                half4v b = *reinterpret_cast<const half4v*>(p.bias + n);
                bv[j] = float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < TM / IT; ++pass) {
        // ---- residual loads of this pass first (row-contiguous layout), so they fly under the staging writes ----
        half8v rs[KI];
        if (!GEGLU && p.residual) {
#pragma unroll
            for (int k = 0; k < KI; ++k) {
                const int idx = lane + 64 * k;
                const int r = idx / CPR, ch = idx - r * CPR;
                const int m = mw0 + ROWS * pass + r, n = nw0 + 8 * ch;
                rs[k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
                if (idx < ROWS * CPR && m < Mi && n < Nout) rs[k] = *reinterpret_cast<const half8v*>(p.residual + (long)m * p.ldr + n);
            }
        }
        // ---- write phase ----
#pragma unroll
        for (int ii = 0; ii < IT; ++ii) {
            const int i = pass * IT + ii;
            const int m = mw0 + 16 * i + l15;
            float* row = stg + (16 * ii + l15) * LD;
            if constexpr (GEGLU) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const float4v v = acc[i][j] + bv[j], gt = acc[i][j + 1] + bv[j + 1];
                    *reinterpret_cast<float4v*>(row + 8 * j + 4 * g) =
                        float4v{v[0] * mv_gelu(gt[0]), v[1] * mv_gelu(gt[1]), v[2] * mv_gelu(gt[2]), v[3] * mv_gelu(gt[3])};
                }
            } else {
                const half_t* rbp = nullptr;
                if (p.rowbias && m < Mi) rbp = p.rowbias + (long)(m / p.rows_per_group) * p.ldrb;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float4v v = acc[i][j] + bv[j];
                    const int n = nw0 + 16 * j + 4 * g;
                    if (rbp && n < p.N) {
                        half4v b = *reinterpret_cast<const half4v*>(rbp + n);
                        v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                    }
                    v *= alpha;
                    if (p.act == MV_ACT_SILU) {
                        v[0] = mv_silu(v[0]); v[1] = mv_silu(v[1]); v[2] = mv_silu(v[2]); v[3] = mv_silu(v[3]);
                    }
                    *reinterpret_cast<float4v*>(row + 16 * j + 4 * g) = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- read phase: lane -> (row, 8-column chunk); consecutive lanes = consecutive 16-byte pieces of a row ----
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int idx = lane + 64 * k;
            const int r = idx / CPR, ch = idx - r * CPR;
            const int m = mw0 + ROWS * pass + r, n = nw0 + 8 * ch;
            if (idx >= ROWS * CPR) continue;
            const float4v f0 = *reinterpret_cast<const float4v*>(stg + r * LD + 8 * ch);
            const float4v f1 = *reinterpret_cast<const float4v*>(stg + r * LD + 8 * ch + 4);
            if (m < Mi && n < Nout) {
                half8v o;
                if (!GEGLU && p.residual) {
                    o = half8v{(half_t)(f0[0] + (float)rs[k][0]), (half_t)(f0[1] + (float)rs[k][1]), (half_t)(f0[2] + (float)rs[k][2]),
                               (half_t)(f0[3] + (float)rs[k][3]), (half_t)(f1[0] + (float)rs[k][4]), (half_t)(f1[1] + (float)rs[k][5]),
                               (half_t)(f1[2] + (float)rs[k][6]), (half_t)(f1[3] + (float)rs[k][7])};
                } else {
                    o = half8v{(half_t)f0[0], (half_t)f0[1], (half_t)f0[2], (half_t)f0[3],
                               (half_t)f1[0], (half_t)f1[1], (half_t)f1[2], (half_t)f1[3]};
                }
                *reinterpret_cast<half8v*>(p.c + (long)m * p.ldc + n) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next pass overwrites the staging rows
    }
}

template <int MODE, int TM, int TN, int WGM, int WGN, int SCHED>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm2_kernel(const GemmArgs2 q) {
    constexpr int NW = WGM * WGN;
    // SCHED: 0 = BK 64, two stages, __syncthreads ring; 6 = the same with BK 32 (half the LDS per block: more blocks per CU for
    // the short-K, memory-bound projections); 3 = BK 64, three stages, counted waits; 4 = BK 32, four stages,
    // counted waits (three K tiles in flight per block: 3/4 of the block's LDS is "in the air" instead of 1/2)
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN, BK = (SCHED == 4 || SCHED == 6) ? 32 : 64;
    constexpr int RP = 512 / BK;             // tile rows per 1-KiB LDS-DMA piece: 8 rows of 128 B or 16 rows of 64 B
    constexpr int CA = BM / RP, CB = BN / RP;
    constexpr int AI = (CA + NW - 1) / NW, BI = (CB + NW - 1) / NW;
    const GemmArgs& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = (SCHED == 4) ? 4 : (SCHED == 3) ? 3 : 2;  // LDS stages
    half_t* sA = reinterpret_cast<half_t*>(smem);  // [NST][BM*BK]
    half_t* sB = sA + NST * BM * BK;                // [NST][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int l15 = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int id = mv_xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    mv_tile_order(id, p.tiles_m, p.tiles_n, p.tile_group, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int Mi = (int)p.M;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, q.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? q.a2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, q.w_bytes, 0x00020000);

    // LDS-DMA lane geometry (the LDS image of a piece is lane-linear, so the swizzle lives on the SOURCE address):
    //   BK 64: lane -> row lane/8, slot lane%8, fetches logical 16-byte chunk slot ^ row
    //   BK 32: lane -> row lane/4, slot lane%4, fetches logical chunk (slot - 2*((row/4)&1)) & 3  (reader: slot =
    //          (g + 2*((row/4)&1)) & 3 -- conflict-free for the ds_read_b128 lane groups over 64-byte rows)
    const int lrow = (BK == 64) ? (lane >> 3) : (lane >> 2);
    const unsigned lsl = (BK == 64) ? (unsigned)((lane & 7) ^ lrow) : (unsigned)(((lane & 3) - 2 * ((lrow >> 2) & 1)) & 3);

    // ---- A rows owned by this lane: chunk c = wave + NW*i ----
    int a_row[AI];   // LINEAR/TCONV: global row; CONV: image base pixel n*hin*win
    int a_y[AI], a_x[AI];
    bool a_ok[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int c = wave + NW * i;
        const int gm = m0 + RP * c + lrow;
        a_ok[i] = (c < CA) && (gm < Mi);
        if (MODE == MV_GEMM_LINEAR) {
            a_row[i] = gm;
            a_y[i] = a_x[i] = 0;
        } else if (MODE == MV_GEMM_CONV3X3) {
            const int hwo = p.hout * p.wout;
            const int n = gm / hwo;
            const int rem = gm - n * hwo;
            const int oy = rem / p.wout, ox = rem - oy * p.wout;
            a_row[i] = n * (p.hin * p.win);
            a_y[i] = oy * p.stride;
            a_x[i] = ox * p.stride;
        } else {
            a_row[i] = gm;
            a_y[i] = (gm / p.hw) % p.t;
            a_x[i] = 0;
        }
    }
    // ---- weight rows owned by this lane ----
    unsigned b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int c = wave + NW * j;
        const int rr = RP * c + lrow;  // LDS row of the B tile
        const int wnt = rr / (16 * TN);
        const int within = rr - wnt * (16 * TN);
        const int jt = within >> 4, r16 = within & 15;
        const int n = n0 + wnt * (16 * TN) + 16 * jt + r16;
        const bool ok = (c < CB) && (n < p.N);
        b_off[j] = ok ? ((unsigned)n * (unsigned)p.K + lsl * 8u) * 2u : kOOB;
    }

    float4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const bool ragged = (p.K & (BK - 1)) != 0;
    unsigned a_off[AI];  // current byte offsets (valid for the current tap / source)
    int kc = 0, tap = 0;

    bool sec = false;      // source of the tile about to be issued
    unsigned soa = 0;      // its scalar byte offset inside a source row

    // cursor step (the only branchy part of the K loop): fixes (source, offsets) of the next tile to issue
    auto prepare = [&]() {
        const bool second = (p.a2 != nullptr) && (kc >= p.c1);
        if (kc == 0 || (second && kc == p.c1)) {  // tap or source changed: rebuild the lane offsets (wave-uniform branch)
            const unsigned ldb = (unsigned)(second ? p.lda2 : p.lda) * 2u;
            int dy = 0, dx = 0;
            if (MODE == MV_GEMM_CONV3X3) {
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            } else if (MODE == MV_GEMM_TCONV3) {
                dy = tap - 1;
            }
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                bool ok = a_ok[i];
                int row;
                if (MODE == MV_GEMM_LINEAR) {
                    row = a_row[i];
                } else if (MODE == MV_GEMM_CONV3X3) {
                    int iy = a_y[i] + dy, ix = a_x[i] + dx;
                    if (p.upsample) {
                        ok = ok && iy >= 0 && iy < 2 * p.hin && ix >= 0 && ix < 2 * p.win;
                        iy >>= 1;
                        ix >>= 1;
                    } else {
                        ok = ok && iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win;
                    }
                    row = a_row[i] + iy * p.win + ix;
                } else {
                    const int tt = a_y[i] + dy;
                    ok = ok && tt >= 0 && tt < p.t;
                    row = a_row[i] + dy * p.hw;
                }
                a_off[i] = ok ? (unsigned)row * ldb + lsl * 16u : kOOB;
            }
        }
        sec = second;
        soa = (unsigned)(second ? kc - p.c1 : kc) * 2u;
        kc += BK;
        if (MODE != MV_GEMM_LINEAR && kc >= p.cin) {
            kc -= p.cin;
            ++tap;
        }
    };
    // branch-free issue of LDS-DMA piece d (0 .. AI+BI-1) of tile kt into stage `buf`
    auto issue_piece = [&](int d, int buf, int kt) {
        const bool kcut = ragged && (kt == nk - 1) && ((int)(kt * BK + lsl * 8) >= p.K);  // ragged K: zero past K (LINEAR)
        if (d < AI) {
            const int c = wave + NW * d;
            if ((CA % NW) != 0 && c >= CA) return;
            const __amdgpu_buffer_rsrc_t rCur = sec ? rA2 : rA;
            const unsigned vo = kcut ? kOOB : a_off[d];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rCur, (__attribute__((address_space(3))) void*)(sA + buf * (BM * BK) + c * (RP * BK)), 16, (int)vo, (int)soa, 0, 0);
        } else {
            const int j = d - AI;
            const int c = wave + NW * j;
            if ((CB % NW) != 0 && c >= CB) return;
            const unsigned vo = kcut ? kOOB : b_off[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rW, (__attribute__((address_space(3))) void*)(sB + buf * (BN * BK) + c * (RP * BK)), 16, (int)vo,
                (int)((unsigned)kt * (BK * 2u)), 0, 0);
        }
    };
    auto issue = [&](int buf, int kt) {
#pragma unroll
        for (int d = 0; d < AI + BI; ++d) issue_piece(d, buf, kt);
    };

    const int a_row0 = wm * 16 * TM + l15;
    const int b_row0 = wn * 16 * TN + l15;
    const int swz = l15 & 7;

    // the MFMAs of one LDS stage: two 32-deep fragment sets of a 64-deep tile, or the single set of a 32-deep tile
    auto mma_stage = [&](int st) __attribute__((always_inline)) {
        if constexpr (BK == 64) {
            mma_tile<TM, TN>(sA + st * (BM * BK), sB + st * (BN * BK), acc, a_row0, b_row0, swz, g);
        } else {
            const half_t* cA = sA + st * (BM * BK);
            const half_t* cB = sB + st * (BN * BK);
            const int slot_off = ((g + 2 * ((l15 >> 2) & 1)) & 3) << 3;
            half8v af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8v*>(cA + (a_row0 + 16 * i) * BK + slot_off);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const half8v*>(cB + (b_row0 + 16 * j) * BK + slot_off);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (SCHED == 3 || SCHED == 4) {
        // NST-stage ring with COUNTED waits: NST-1 K tiles are in flight while one is multiplied, and nothing ever drains
        // the LDS-DMA queue inside the loop.  Per K step: this wave waits until its own pieces of tile kt have landed
        // (s_waitcnt vmcnt(pieces of the younger tiles)), one raw s_barrier makes every wave's pieces visible and proves
        // that all waves have finished reading the stage the next issue overwrites (it held tile kt-1), then the pieces of
        // tile kt+NST-1 are issued and the MFMAs of tile kt run.  (__syncthreads() would add vmcnt(0) and serialise the ring.)
        int pw = 0;  // LDS-DMA pieces this wave issues per K tile (wave-uniform)
#pragma unroll
        for (int d = 0; d < AI; ++d) pw += ((CA % NW) != 0 && wave + NW * d >= CA) ? 0 : 1;
#pragma unroll
        for (int d = 0; d < BI; ++d) pw += ((CB % NW) != 0 && wave + NW * d >= CB) ? 0 : 1;
#pragma unroll
        for (int t0 = 0; t0 < NST - 1; ++t0) {
            if (t0 < nk) {
                prepare();
                issue(t0, t0);
            }
        }
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            int younger = nk - 1 - kt;  // tiles issued after tile kt that may stay in flight
            if (younger > NST - 2) younger = NST - 2;
            switch (younger * pw) {  // vmcnt takes an immediate
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // over-waiting is always safe
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NST - 1 < nk) {
                prepare();
                issue(cur == 0 ? NST - 1 : cur - 1, kt + NST - 1);  // (cur + NST - 1) % NST: the stage tile kt-1 just left
            }
            mma_stage(cur);
            cur = (cur == NST - 1) ? 0 : cur + 1;
        }
    } else {
        prepare();
        issue(0, 0);
        __syncthreads();  // drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        for (int kt = 0; kt < nk - 1; ++kt) {
            const int cur = kt & 1;
            prepare();
            issue(cur ^ 1, kt + 1);
            mma_stage(cur);
            __syncthreads();
        }
        mma_stage((nk - 1) & 1);
    }

    // ---- epilogue ----
    const float alpha = p.alpha ? fabsf(*p.alpha) : 1.0f;
    const int nw0 = n0 + wn * 16 * TN;
    const int mw0 = m0 + wm * 16 * TM;
    if (q.wide) {
        // LDS-staged: the MFMA accumulator layout gives a lane 4 channels of ONE row (a wave store instruction would
        // touch 16 rows x 32 bytes); staging the fp32 tile through LDS turns it into 16-byte-per-lane accesses whose
        // consecutive lanes cover consecutive bytes of a row (residual loads and output stores in whole 128-byte lines).
        __syncthreads();  // every wave is done reading the operand tiles: their LDS is reused
        float* stg = reinterpret_cast<float*>(smem);
        // (the launcher sizes the dynamic LDS as max(operand stages, this staging area))
        // 16-row tiles staged per pass: the 128-row wave tiles of the 256-row blocks sit at the 256-register cap, where the
        // residual prefetch of a 32-row pass (20 registers) spilled an accumulator to scratch; 16-row passes keep it in registers
        constexpr int EIT = (TM >= 8 || SCHED == 6) ? 1 : 2;  // SCHED 6: 16-row passes keep the block's LDS small (blocks per CU)
        if (p.geglu) {
            if constexpr ((TN & 1) == 0) epilogue_staged<TM, TN, true, EIT>(p, acc, stg, wave, mw0, nw0 >> 1, lane, alpha, Mi);
        } else {
            epilogue_staged<TM, TN, false, EIT>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi);
        }
        return;
    }
    epilogue_narrow<TM, TN>(p, acc, mw0, nw0, lane, alpha, Mi);
}

// =====================================================================================================================
// v3 kernel: the v2 tile engine made PERSISTENT.  Most launches of the UNet step have short K loops (K = 320 ... 1280:
// 5-20 steps), so a one-tile-per-block grid spends as long in its prologue (row decode, first HBM round trip) and
// epilogue as in MFMAs.  Here a block walks tiles  blockIdx, blockIdx + grid, ...  and, during the LAST K step of a
// tile, already decodes the next tile and issues its first K tile into the LDS stage that has just become free; the
// epilogue (staged through the other, just-consumed stage) then runs under that load.
//   LDS: 2 stages of [A tile | B tile]; per tile the ring simply continues from whichever stage holds its K step 0.
template <int MODE, int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(const GemmArgs2 q) {
    constexpr int NW = 4, WGN = 2;
    constexpr int BM = 32 * TM, BN = 32 * TN, BK = 64;
    constexpr int CA = BM / 8, CB = BN / 8;
    constexpr int AI = CA / NW, BI = (CB + NW - 1) / NW;
    static_assert(CA % NW == 0, "A chunks must divide over the waves");
    constexpr int STAGE = (BM + BN) * BK;  // halfs per stage
    static_assert(NW * 16 * (16 * TN + 4) * 4 <= STAGE * 2, "output staging does not fit one operand stage");
    const GemmArgs& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* st = reinterpret_cast<half_t*>(smem);  // stage s: A at st + s*STAGE, B at st + s*STAGE + BM*BK

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int l15 = lane & 15, g = lane >> 4;
    const int Mi = (int)p.M;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int G = (int)gridDim.x;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, q.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? q.a2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, q.w_bytes, 0x00020000);

    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    const int nk = (p.K + BK - 1) / BK;
    const bool ragged = (p.K & (BK - 1)) != 0;

    // it-th tile of this block: tiles [it*G, it*G + cnt) are live together; inside that range the XCD remap keeps the
    // N-tiles of one M-tile on one XCD (shared A rows in that L2)
    auto tile_of = [&](int it, int& tm0, int& tn0) -> bool {
        const int base = it * G;
        const int cnt = (ntiles - base < G) ? (ntiles - base) : G;
        if ((int)blockIdx.x >= cnt) return false;
        const int id = base + mv_xcd_remap(blockIdx.x, cnt);
        const int tile_m = id / p.tiles_n;
        tm0 = tile_m * BM;
        tn0 = (id - tile_m * p.tiles_n) * BN;
        return true;
    };

    // ---- loader state (of the tile whose K tiles are being issued) ----
    int a_row[AI], a_y[AI], a_x[AI];
    bool a_ok[AI];
    unsigned a_off[AI], b_off[BI];
    int kc = 0, tap = 0;
    bool sec = false;
    unsigned soa = 0;

    auto setup_loader = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int gm = m0 + 8 * (wave + NW * i) + lrow;
            a_ok[i] = gm < Mi;
            if (MODE == MV_GEMM_LINEAR) {
                a_row[i] = gm;
                a_y[i] = a_x[i] = 0;
            } else if (MODE == MV_GEMM_CONV3X3) {
                const int hwo = p.hout * p.wout;
                const int n = gm / hwo;
                const int rem = gm - n * hwo;
                const int oy = rem / p.wout, ox = rem - oy * p.wout;
                a_row[i] = n * (p.hin * p.win);
                a_y[i] = oy * p.stride;
                a_x[i] = ox * p.stride;
            } else {
                a_row[i] = gm;
                a_y[i] = (gm / p.hw) % p.t;
                a_x[i] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int c = wave + NW * j;
            const int n = n0 + 8 * c + lrow;  // LDS row r of the B tile holds weight row n0 + r
            const bool ok = (c < CB) && (n < p.N);
            b_off[j] = ok ? ((unsigned)n * (unsigned)p.K + lsl * 8u) * 2u : kOOB;
        }
        kc = 0;
        tap = 0;
    };
    auto prepare = [&]() {
        const bool second = (p.a2 != nullptr) && (kc >= p.c1);
        if (kc == 0 || (second && kc == p.c1)) {
            const unsigned ldb = (unsigned)(second ? p.lda2 : p.lda) * 2u;
            int dy = 0, dx = 0;
            if (MODE == MV_GEMM_CONV3X3) {
                dy = tap / 3 - 1;
                dx = tap - (tap / 3) * 3 - 1;
            } else if (MODE == MV_GEMM_TCONV3) {
                dy = tap - 1;
            }
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                bool ok = a_ok[i];
                int row;
                if (MODE == MV_GEMM_LINEAR) {
                    row = a_row[i];
                } else if (MODE == MV_GEMM_CONV3X3) {
                    int iy = a_y[i] + dy, ix = a_x[i] + dx;
                    if (p.upsample) {
                        ok = ok && iy >= 0 && iy < 2 * p.hin && ix >= 0 && ix < 2 * p.win;
                        iy >>= 1;
                        ix >>= 1;
                    } else {
                        ok = ok && iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win;
                    }
                    row = a_row[i] + iy * p.win + ix;
                } else {
                    const int tt = a_y[i] + dy;
                    ok = ok && tt >= 0 && tt < p.t;
                    row = a_row[i] + dy * p.hw;
                }
                a_off[i] = ok ? (unsigned)row * ldb + lsl * 16u : kOOB;
            }
        }
        sec = second;
        soa = (unsigned)(second ? kc - p.c1 : kc) * 2u;
        kc += BK;
        if (MODE != MV_GEMM_LINEAR && kc >= p.cin) {
            kc -= p.cin;
            ++tap;
        }
    };
    auto issue = [&](int buf, int kt) {
        const bool kcut = ragged && (kt == nk - 1) && ((int)(kt * BK + lsl * 8) >= p.K);
        half_t* dA = st + buf * STAGE;
        half_t* dB = dA + BM * BK;
        const __amdgpu_buffer_rsrc_t rCur = sec ? rA2 : rA;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int c = wave + NW * i;
            const unsigned vo = kcut ? kOOB : a_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rCur, (__attribute__((address_space(3))) void*)(dA + c * (8 * BK)), 16, (int)vo,
                                                     (int)soa, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BI; ++j) {
            const int c = wave + NW * j;
            if ((CB % NW) != 0 && c >= CB) break;
            const unsigned vo = kcut ? kOOB : b_off[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)(dB + c * (8 * BK)), 16, (int)vo,
                                                     (int)((unsigned)kt * (BK * 2u)), 0, 0);
        }
    };

    const int a_row0 = wm * 16 * TM + l15;
    const int b_row0 = wn * 16 * TN + l15;
    const int swz = l15 & 7;
    const float alpha = p.alpha ? fabsf(*p.alpha) : 1.0f;

    int lm0 = 0, ln0 = 0;
    if (!tile_of(0, lm0, ln0)) return;  // block-uniform
    setup_loader(lm0, ln0);
    prepare();
    issue(0, 0);
    __syncthreads();
    int b0 = 0;
    for (int it = 0;; ++it) {
        const int cm0 = lm0, cn0 = ln0;  // the tile being computed
        int nm0 = 0, nn0 = 0;
        const bool has_next = tile_of(it + 1, nm0, nn0);
        float4v acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = (b0 + kt) & 1;
            if (kt + 1 < nk) {
                prepare();
                issue(cur ^ 1, kt + 1);
            } else if (has_next) {  // cross-tile prefetch: K step 0 of the next tile goes into the stage freed one step ago
                lm0 = nm0;
                ln0 = nn0;
                setup_loader(lm0, ln0);
                prepare();
                issue(cur ^ 1, 0);
            }
            mma_tile<TM, TN>(st + cur * STAGE, st + cur * STAGE + BM * BK, acc, a_row0, b_row0, swz, g);
            __syncthreads();  // stage `cur` fully consumed by every wave; the LDS-DMA issued above has landed
        }
        const int last = (b0 + nk - 1) & 1;
        const int mw0 = cm0 + wm * 16 * TM, nw0 = cn0 + wn * 16 * TN;
        if (q.wide) {
            float* stg = reinterpret_cast<float*>(st + last * STAGE);  // the stage consumed last is free; the other one
                                                                      // may already hold the next tile's first K tile
            if (p.geglu) {
                if constexpr ((TN & 1) == 0) epilogue_staged<TM, TN, true, 1>(p, acc, stg, wave, mw0, nw0 >> 1, lane, alpha, Mi);
            } else {
                epilogue_staged<TM, TN, false, 1>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi);
            }
        } else {
            epilogue_narrow<TM, TN>(p, acc, mw0, nw0, lane, alpha, Mi);
        }
        if (!has_next) break;
        __syncthreads();  // every wave has left the staging rows before the next tile's K step 1 is written there
        b0 = last ^ 1;
    }
}

int mv_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <int MODE, int TM, int TN>
int launch_cfg3(const GemmArgs2& a0, hipStream_t stream) {
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int smem = 2 * (BM + BN) * 64 * (int)sizeof(half_t);
    GemmArgs2 a = a0;
    a.g.tiles_m = (int)((a.g.M + BM - 1) / BM);
    a.g.tiles_n = (a.g.N + BN - 1) / BN;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<MODE, TM, TN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            mv_set_error("mv_gemm_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return MV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    const long ntiles = (long)a.g.tiles_m * a.g.tiles_n;
    const long resident = 2L * mv_num_cus();  // two 256-thread blocks per CU (LDS-limited)
    dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
    hipLaunchKernelGGL((gemm3_kernel<MODE, TM, TN>), grid, dim3(256), smem, stream, a);
    MV_CHECK_LAUNCH("mv_gemm_f16");
    return MV_OK;
}

template <int MODE>
int launch_mode3(const GemmArgs2& a, hipStream_t stream) {
    const GemmArgs& g = a.g;
    if (g.geglu) return launch_cfg3<MODE, 4, 4>(a, stream);
    const bool n160 = (g.N % 160) == 0;
    const long tiles_n = n160 ? g.N / 160 : (g.N + 127) / 128;
    const long tiles_m128 = (g.M + 127) / 128;
    const bool small = tiles_m128 * tiles_n < 512;
    if (n160) return small ? launch_cfg3<MODE, 2, 5>(a, stream) : launch_cfg3<MODE, 4, 5>(a, stream);
    return small ? launch_cfg3<MODE, 2, 4>(a, stream) : launch_cfg3<MODE, 4, 4>(a, stream);
}

template <int MODE, int TM, int TN, int WGM, int WGN, int SCHED>
int launch_cfg2s(const GemmArgs2& a0, hipStream_t stream) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int smem_ops = (SCHED == 4 ? 4 * 32 : SCHED == 3 ? 3 * 64 : SCHED == 6 ? 2 * 32 : 2 * 64) * (BM + BN) * (int)sizeof(half_t);
    // the LDS-staged epilogue reuses the operand LDS: 32 (or 16) fp32 rows of (16 TN + 4) floats per wave must fit as well
    constexpr int smem_epi = WGM * WGN * ((TM >= 8 || SCHED == 6) ? 16 : 32) * (16 * TN + 4) * 4;
    constexpr int smem = smem_ops > smem_epi ? smem_ops : smem_epi;
    static_assert(smem <= 160 * 1024, "tile does not fit LDS");
    GemmArgs2 a = a0;
    a.g.tiles_m = (int)((a.g.M + BM - 1) / BM);
    a.g.tiles_n = (a.g.N + BN - 1) / BN;
    a.g.tile_group = g_gemm_tile_group;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<MODE, TM, TN, WGM, WGN, SCHED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            mv_set_error("mv_gemm_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return MV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    dim3 grid((unsigned)(a.g.tiles_m * a.g.tiles_n));
    hipLaunchKernelGGL((gemm2_kernel<MODE, TM, TN, WGM, WGN, SCHED>), grid, dim3(64 * WGM * WGN), smem, stream, a);
    MV_CHECK_LAUNCH("mv_gemm_f16");
    return MV_OK;
}

template <int MODE, int TM, int TN, int WGM, int WGN>
int launch_cfg2(const GemmArgs2& a, hipStream_t stream) {
    return launch_cfg2s<MODE, TM, TN, WGM, WGN, 0>(a, stream);
}

// ---- tile-configuration catalogue --------------------------------------------------------------------------------------
// Every (wave tile TM x TN in 16-row / 16-column MFMA tiles, waves WGM x WGN, schedule) the kernel is instantiated for, under
// a stable id: block tile = 16 TM WGM rows x 16 TN WGN columns.  The ids are what mv_set_gemm_force() and the per-shape table
// of gemm_tuned.h (written by tools/gpu_gemm_tune.py from timings on the MI355X) refer to.  SCHED 0: BK 64, two LDS stages
// behind __syncthreads; 3: BK 64, three stages behind counted waits; 4: BK 32, four stages behind counted waits.
#define MV_GEMM_CFGS(X)                                                                                              \
    X(0, 4, 5, 2, 2, 0)  /* 128x160, 4 waves            */ X(1, 2, 5, 2, 2, 0)   /* 64x160                     */      \
    X(2, 4, 4, 2, 2, 0)  /* 128x128                     */ X(3, 2, 4, 2, 2, 0)   /* 64x128                     */      \
    X(4, 4, 5, 4, 2, 3)  /* 256x160, 8 waves, 3 stages  */ X(5, 4, 4, 4, 2, 3)   /* 256x128, 8 waves, 3 stages */      \
    X(6, 8, 5, 2, 4, 0)  /* 256x320, 8 waves            */ X(7, 8, 4, 2, 4, 0)   /* 256x256, 8 waves           */      \
    X(8, 4, 5, 4, 2, 0)  /* 256x160, 8 waves, 2 stages  */ X(9, 4, 4, 4, 2, 0)   /* 256x128, 8 waves, 2 stages */      \
    X(10, 4, 5, 2, 2, 4) /* 128x160, BK 32 x 4 stages   */ X(11, 2, 5, 2, 2, 4)  /* 64x160, BK 32 x 4          */      \
    X(12, 4, 4, 2, 2, 4) /* 128x128, BK 32 x 4          */ X(13, 2, 4, 2, 2, 4)  /* 64x128, BK 32 x 4          */      \
    X(14, 8, 4, 2, 4, 4) /* 256x256, 8 waves, BK 32 x 4 */ X(15, 4, 4, 2, 4, 3)  /* 128x256, 8 waves, 3 stages */      \
    X(16, 4, 5, 2, 4, 0) /* 128x320, 8 waves            */ X(17, 2, 5, 4, 2, 3)  /* 128x160, 8 waves, 3 stages */      \
    X(18, 8, 5, 2, 4, 4) /* 256x320, 8 waves, BK 32 x 4 */ X(19, 2, 5, 2, 1, 0)  /* 64x80, 2 waves (small grids)  */      \
    X(20, 2, 5, 1, 2, 0) /* 32x160, 2 waves             */ X(21, 2, 5, 2, 2, 6)  /* 64x160, BK 32 x 2 (4 blocks/CU) */   \
    X(22, 2, 4, 2, 2, 6) /* 64x128, BK 32 x 2           */ X(23, 2, 5, 4, 2, 0)  /* 128x160, 8 waves of 32x80: 4 waves/SIMD */ \
    X(24, 2, 4, 4, 2, 0) /* 128x128, 8 waves of 32x64   */
constexpr int kNumGemmCfgs = 25;
struct GemmCfgDesc { int tm, tn, wgm, wgn, sched; };
constexpr GemmCfgDesc kGemmCfgs[kNumGemmCfgs] = {
#define MV_X(id, tm, tn, wgm, wgn, sched) {tm, tn, wgm, wgn, sched},
    MV_GEMM_CFGS(MV_X)
#undef MV_X
};

// a configuration can run a problem iff its epilogue can: the GEGLU gate pairs 16-column tiles, so TN must be even
inline bool gemm_cfg_applies(int id, const GemmArgs& g) {
    return id >= 0 && id < kNumGemmCfgs && (!g.geglu || (kGemmCfgs[id].tn & 1) == 0);
}

template <int MODE>
int launch_by_id(const GemmArgs2& a, hipStream_t stream, int id) {
    switch (id) {
#define MV_X(cid, tm, tn, wgm, wgn, sched) \
    case cid: return launch_cfg2s<MODE, tm, tn, wgm, wgn, sched>(a, stream);
        MV_GEMM_CFGS(MV_X)
#undef MV_X
    }
    mv_set_error("mv_gemm_f16: unknown tile configuration %d", id);
    return MV_ERR_INVALID;
}

// per-shape choices measured on the MI355X (exact match on mode / M / N / K / geglu; anything else follows the rules below)
struct GemmTuned { int mode; long M; int N, K, geglu, cfg; };
#include "gemm_tuned.h"
int g_gemm_force = -1;      // mv_set_gemm_force: use this configuration wherever it applies (tuner / A-B runs); -1 = off
int g_gemm_use_tuned = 1;   // the table applies to the default variant only

// tile selection for the v2 kernel.  variant 2: the 4-wave tiles of v1; variant 3: 8-wave 256x160 tiles where the grid
// still fills the chip (>= 2 blocks per CU worth of work).
template <int MODE>
int launch_mode2(const GemmArgs2& a, hipStream_t stream, int variant) {
    const GemmArgs& g = a.g;
    if (g_gemm_force >= 0 && gemm_cfg_applies(g_gemm_force, g)) return launch_by_id<MODE>(a, stream, g_gemm_force);
    if (variant == 2 && g_gemm_force < 0 && g_gemm_use_tuned) {
        for (int i = 0; i < kNumGemmTuned; ++i) {
            const GemmTuned& e = kGemmTuned[i];
            if (e.mode == MODE && e.M == g.M && e.N == g.N && e.K == g.K && e.geglu == g.geglu && gemm_cfg_applies(e.cfg, g))
                return launch_by_id<MODE>(a, stream, e.cfg);
        }
    }
    if (g.geglu) {
        // 256x256 tile (8 waves as 2 x 4, wave tile 128x64) where it still gives every CU a block: +5..+14 % on the FF1
        // projections (profiles/r01m_gemm_variant_ab.log); opt-in (variant 8), see the note on the 256x320 tile below
        if (variant == 8 && (g.N % 256) == 0 && (g.M + 255) / 256 * (g.N / 256) >= 200)
            return launch_cfg2s<MODE, 8, 4, 2, 4, 0>(a, stream);
        if (variant == 5 && (g.M + 255) / 256 * ((g.N + 127) / 128) >= 200) return launch_cfg2s<MODE, 4, 4, 4, 2, 3>(a, stream);
        if (variant == 3 && (g.M + 255) / 256 * ((g.N + 127) / 128) >= 512) return launch_cfg2<MODE, 4, 4, 4, 2>(a, stream);
        return launch_cfg2<MODE, 4, 4, 2, 2>(a, stream);
    }
    const bool n160 = (g.N % 160) == 0;
    const long tiles_n = n160 ? g.N / 160 : (g.N + 127) / 128;
    const long tiles_m128 = (g.M + 127) / 128;
    const bool small = tiles_m128 * tiles_n < 512;
    // 256x320 block tile, 8 waves as 2(M) x 4(N), wave tile 128x80: twice the flops per LDS-DMA byte of the 128x160 tile and
    // 0.33 instead of 0.45 fragment reads per MFMA.  One block per CU, so it needs a K loop long enough to amortise its
    // un-overlapped prologue / epilogue: measured (profiles/r01l_gemm_variant_ab.log) +10..+20 % on the conv / temporal-conv
    // / K >= 640 linear shapes of levels 0-1, -20 % on the N = 320, K = 320 projections.
    const bool big320 = !g.geglu && (g.N % 320) == 0 && (g.M + 255) / 256 * (g.N / 320) >= 200;
    if (variant == 7 && big320) return launch_cfg2s<MODE, 8, 5, 2, 4, 0>(a, stream);
    // NOT in the default variant yet: parity and the micro-benchmarks are green (r01l / r01m), but the only whole-model run
    // with these tiles (bench.py under hipGraph replay + two streams, r01n) did not finish inside the GPU budget that was
    // left, so variant 2 stays exactly what r01k verified and variant 8 = 2 + these tiles (MUSEV_GEMM_VARIANT=8).
    if (variant == 8 && big320 && (g.K >= 640 || g.N >= 960)) return launch_cfg2s<MODE, 8, 5, 2, 4, 0>(a, stream);
    if (variant == 6 && !g.geglu) {  // experiment: BK 32, four-stage counted ring on the 4-wave tiles
        if (n160) return small ? launch_cfg2s<MODE, 2, 5, 2, 2, 4>(a, stream) : launch_cfg2s<MODE, 4, 5, 2, 2, 4>(a, stream);
        return small ? launch_cfg2s<MODE, 2, 4, 2, 2, 4>(a, stream) : launch_cfg2s<MODE, 4, 4, 2, 2, 4>(a, stream);
    }
    if (n160) {
        // 8-wave 256x160 tiles on a three-stage counted-wait ring.  Measured (profiles/r01e_gemm_variant_ab.log): +20 % where
        // they form ONE round of blocks over the CUs (the M = 6656 level: 26 x 8 = 208 blocks, against 832 small
        // 64x160 tiles), -10..-18 % on the large grids, where two independent 4-wave blocks per CU overlap better.
        const long blocks8 = (g.M + 255) / 256 * tiles_n;
        const long cus = mv_num_cus();
        if (variant == 5 && blocks8 >= 200) return launch_cfg2s<MODE, 4, 5, 4, 2, 3>(a, stream);
        if ((variant == 2 || variant == 8) && blocks8 * 5 >= cus * 4 && blocks8 <= cus) return launch_cfg2s<MODE, 4, 5, 4, 2, 3>(a, stream);
        if (variant == 3 && (g.M + 255) / 256 * tiles_n >= 512) return launch_cfg2<MODE, 4, 5, 4, 2>(a, stream);
        return small ? launch_cfg2<MODE, 2, 5, 2, 2>(a, stream) : launch_cfg2<MODE, 4, 5, 2, 2>(a, stream);
    }
    return small ? launch_cfg2<MODE, 2, 4, 2, 2>(a, stream) : launch_cfg2<MODE, 4, 4, 2, 2>(a, stream);
}

template <int MODE>
int launch_mode(const GemmArgs& a, hipStream_t stream) {
    // tile selection: BN = 160 when it divides N (all UNet widths are multiples of 320), else 128;
    // BM = 128 unless that leaves the 256 CUs under-filled, then 64.
    if (a.geglu) return launch_cfg<MODE, 4, 4>(a, stream);
    const bool n160 = (a.N % 160) == 0;
    const long tiles_n = n160 ? a.N / 160 : (a.N + 127) / 128;
    const long tiles_m128 = (a.M + 127) / 128;
    const bool small = tiles_m128 * tiles_n < 512;
    if (n160) return small ? launch_cfg<MODE, 2, 5>(a, stream) : launch_cfg<MODE, 4, 5>(a, stream);
    return small ? launch_cfg<MODE, 2, 4>(a, stream) : launch_cfg<MODE, 4, 4>(a, stream);
}

}  // namespace

extern "C" int mv_set_gemm_variant(int v) {
    MV_REQUIRE(v >= 0 && v <= 8, "mv_set_gemm_variant: variant %d not in [0, 8]", v);
    g_gemm_stage = v;
    return MV_OK;
}

extern "C" int mv_set_gemm_force(int cfg) {
    MV_REQUIRE(cfg >= -2 && cfg < kNumGemmCfgs, "mv_set_gemm_force: configuration %d not in [-2, %d)", cfg, kNumGemmCfgs);
    g_gemm_use_tuned = cfg != -2;  // -2: rules only (ignore the tuned table); -1: table + rules; >= 0: this configuration
    g_gemm_force = cfg < 0 ? -1 : cfg;
    return MV_OK;
}

extern "C" int mv_gemm_num_configs(void) { return kNumGemmCfgs; }

extern "C" int mv_gemm_config_desc(int cfg, int32_t* desc5) {
    MV_REQUIRE(cfg >= 0 && cfg < kNumGemmCfgs && desc5, "mv_gemm_config_desc: bad args");
    const GemmCfgDesc& c = kGemmCfgs[cfg];
    desc5[0] = 16 * c.tm * c.wgm;  // block rows
    desc5[1] = 16 * c.tn * c.wgn;  // block columns
    desc5[2] = c.wgm * c.wgn;      // waves
    desc5[3] = (c.sched == 4 || c.sched == 6) ? 32 : 64;  // BK
    desc5[4] = c.sched == 4 ? 4 : c.sched == 3 ? 3 : 2;   // LDS stages
    return MV_OK;
}

extern "C" int mv_set_gemm_tile_group(int group) {
    MV_REQUIRE(group >= 0 && group <= 64, "mv_set_gemm_tile_group: group %d not in [0, 64]", group);
    g_gemm_tile_group = group;
    return MV_OK;
}

// host-side evaluation of the workgroup -> tile map the v2 kernel uses (the same inline functions): introspection for
// tests and for reasoning about L2 locality; launches nothing
extern "C" int mv_gemm_tile_order(int tiles_m, int tiles_n, int group, int32_t* tile_m, int32_t* tile_n) {
    MV_REQUIRE(tiles_m > 0 && tiles_n > 0 && (long)tiles_m * tiles_n < (1L << 31) && tile_m && tile_n && group >= 0,
               "mv_gemm_tile_order: bad args");
    const int nwg = tiles_m * tiles_n;
    for (int b = 0; b < nwg; ++b) {
        int tm, tn;
        mv_tile_order(mv_xcd_remap(b, nwg), tiles_m, tiles_n, group, &tm, &tn);
        tile_m[b] = tm;
        tile_n[b] = tn;
    }
    return MV_OK;
}

extern "C" int mv_gemm_f16(const mv_gemm_desc* d, void* stream) {
    MV_REQUIRE(d != nullptr, "mv_gemm_f16: null descriptor");
    MV_REQUIRE(d->a && d->w && d->c, "mv_gemm_f16: null a/w/c pointer");
    MV_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "mv_gemm_f16: empty problem M=%ld N=%d K=%d", (long)d->M, d->N, d->K);
    MV_REQUIRE(d->N % 4 == 0 && d->K % 8 == 0, "mv_gemm_f16: need N %% 4 == 0 and K %% 8 == 0 (N=%d K=%d)", d->N, d->K);
    MV_REQUIRE(d->ldc % 4 == 0 && d->lda % 8 == 0, "mv_gemm_f16: lda must be a multiple of 8 and ldc of 4");
    const int c2 = d->a2 ? d->c2 : 0;
    const int cin = d->c1 + c2;
    MV_REQUIRE(d->c1 > 0 && d->c1 % 8 == 0 && c2 % 8 == 0, "mv_gemm_f16: c1/c2 must be multiples of 8");
    if (d->a2) {
        MV_REQUIRE(d->c1 % 64 == 0 && d->lda2 % 8 == 0, "mv_gemm_f16: two-source input needs c1 %% 64 == 0");
    }
    int taps = 1;
    if (d->mode == MV_GEMM_CONV3X3) taps = 9;
    else if (d->mode == MV_GEMM_TCONV3) taps = 3;
    else MV_REQUIRE(d->mode == MV_GEMM_LINEAR, "mv_gemm_f16: bad mode %d", d->mode);
    MV_REQUIRE(d->K == taps * cin, "mv_gemm_f16: K=%d != taps*cin=%d*%d", d->K, taps, cin);
    if (taps > 1) MV_REQUIRE(cin % 64 == 0, "mv_gemm_f16: conv modes need cin %% 64 == 0 (cin=%d)", cin);
    if (d->residual) MV_REQUIRE(d->ldr % 4 == 0, "mv_gemm_f16: ldr %% 4");
    if (d->rowbias) MV_REQUIRE(d->ldrb % 4 == 0 && d->rows_per_group > 0, "mv_gemm_f16: rowbias needs ldrb %% 4 and rows_per_group > 0");
    if (d->geglu) {
        MV_REQUIRE(d->N % 32 == 0 && !d->rowbias && !d->residual && !d->alpha && d->act == MV_ACT_NONE,
                   "mv_gemm_f16: geglu epilogue needs N %% 32 == 0 and no other epilogue terms");
    }
    GemmArgs a;
    a.a = (const half_t*)d->a; a.a2 = (const half_t*)d->a2; a.w = (const half_t*)d->w; a.c = (half_t*)d->c;
    a.bias = (const half_t*)d->bias; a.rowbias = (const half_t*)d->rowbias; a.residual = (const half_t*)d->residual;
    a.alpha = d->alpha;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.lda2 = d->lda2; a.ldc = d->ldc; a.ldr = d->ldr; a.ldrb = d->ldrb;
    a.c1 = d->c1; a.cin = cin;
    a.stride = d->stride; a.upsample = d->upsample; a.hin = d->hin; a.win = d->win; a.hout = d->hout; a.wout = d->wout;
    a.t = d->t; a.hw = d->hw;
    a.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1; a.act = d->act; a.geglu = d->geglu;
    a.tiles_m = a.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    if (d->mode == MV_GEMM_CONV3X3) {
        MV_REQUIRE(d->stride == 1 || d->stride == 2, "mv_gemm_f16: conv stride must be 1 or 2");
        MV_REQUIRE(!(d->upsample && d->stride != 1), "mv_gemm_f16: upsample requires stride 1");
        MV_REQUIRE(d->hin > 0 && d->win > 0 && d->hout > 0 && d->wout > 0, "mv_gemm_f16: conv geometry missing");
        MV_REQUIRE(d->M % ((long)d->hout * d->wout) == 0, "mv_gemm_f16: M is not a whole number of output images");
    }
    if (d->mode == MV_GEMM_TCONV3)
        MV_REQUIRE(d->t > 0 && d->hw > 0 && d->M % ((long)d->t * d->hw) == 0, "mv_gemm_f16: tconv geometry: M must be B*T*HW");

    // v2 (buffer-descriptor LDS-DMA) needs every source to span < 2 GiB (32-bit byte offsets, 0x80000000 = "zero" marker)
    if (g_gemm_stage >= 2) {
        const long rows_in = d->mode == MV_GEMM_CONV3X3 ? (d->M / ((long)d->hout * d->wout)) * d->hin * d->win : d->M;
        const long a_bytes = ((rows_in - 1) * (long)d->lda + d->c1) * 2;
        const long a2_bytes = d->a2 ? ((rows_in - 1) * (long)d->lda2 + c2) * 2 : 0;
        const long w_bytes = (long)d->N * d->K * 2;
        const long lim = 0x7fffffffL;
        if (a_bytes < lim && a2_bytes < lim && w_bytes < lim && d->M < lim) {
            GemmArgs2 b;
            b.g = a;
            b.a_bytes = (unsigned)a_bytes; b.a2_bytes = (unsigned)a2_bytes; b.w_bytes = (unsigned)w_bytes;
            auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
            b.wide = (d->N % 8 == 0) && (d->ldc % 8 == 0) && al16(d->c) && (!d->bias || al16(d->bias)) &&
                     (!d->rowbias || (d->ldrb % 8 == 0 && al16(d->rowbias))) &&
                     (!d->residual || (d->ldr % 8 == 0 && al16(d->residual)));
            if (g_gemm_stage == 4) {
                if (d->mode == MV_GEMM_CONV3X3) return launch_mode3<MV_GEMM_CONV3X3>(b, s);
                if (d->mode == MV_GEMM_TCONV3) return launch_mode3<MV_GEMM_TCONV3>(b, s);
                return launch_mode3<MV_GEMM_LINEAR>(b, s);
            }
            if (d->mode == MV_GEMM_CONV3X3) return launch_mode2<MV_GEMM_CONV3X3>(b, s, g_gemm_stage);
            if (d->mode == MV_GEMM_TCONV3) return launch_mode2<MV_GEMM_TCONV3>(b, s, g_gemm_stage);
            return launch_mode2<MV_GEMM_LINEAR>(b, s, g_gemm_stage);
        }
    }
    if (d->mode == MV_GEMM_CONV3X3) return launch_mode<MV_GEMM_CONV3X3>(a, s);
    if (d->mode == MV_GEMM_TCONV3) return launch_mode<MV_GEMM_TCONV3>(a, s);
    return launch_mode<MV_GEMM_LINEAR>(a, s);
}
