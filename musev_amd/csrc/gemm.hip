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

int g_gemm_stage = 1;  // tuning knob (mv_set_gemm_variant): 0 register staging, 1 LDS-DMA

template <int MODE, int TM, int TN>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    return g_gemm_stage == 0 ? launch_cfg_s<MODE, TM, TN, 0>(a, stream) : launch_cfg_s<MODE, TM, TN, 1>(a, stream);
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
    g_gemm_stage = v ? 1 : 0;
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
        return launch_mode<MV_GEMM_CONV3X3>(a, s);
    }
    if (d->mode == MV_GEMM_TCONV3) {
        MV_REQUIRE(d->t > 0 && d->hw > 0 && d->M % ((long)d->t * d->hw) == 0, "mv_gemm_f16: tconv geometry: M must be B*T*HW");
        return launch_mode<MV_GEMM_TCONV3>(a, s);
    }
    return launch_mode<MV_GEMM_LINEAR>(a, s);
}
