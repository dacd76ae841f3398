// gemm.hip -- fp16 implicit-GEMM family for gfx950 (MFMA 16x16x32 f16, fp32 accumulate).
//
// One kernel template covers the three contraction shapes of the UNet3D step:
//   MV_GEMM_LINEAR  : Linear / 1x1 conv               (K3)
//   MV_GEMM_CONV3X3 : NHWC 3x3 conv, pad 1, stride 1|2, optional fused nearest-x2 upsample   (K2, K9)
//   MV_GEMM_TCONV3  : Conv3d (3,1,1), pad (1,0,0) over a [B,T,HW,C] tensor                    (K4)
// all with an optional second channel-concatenated source (K10: the up-path torch.cat is never materialised) and a fused
// epilogue: bias, per-frame row bias (time / frame embedding), |alpha| scale (temporal_weight), SiLU, residual add, or the
// GEGLU gate.
//
// Tiling (CDNA4, 64-wide waves): block = WGM x WGN waves; wave tile = (16*TM) x (16*TN) built from 16x16x32 MFMAs; BK = 64.
// Both operands are K-contiguous ("B^T" form: activations [m][k], weights [n][k]), so every MFMA fragment is one 16-byte
// ds_read_b128.  LDS tiles are row-major with 128-byte rows and the 16-byte slot index XOR-ed with (row & 7): conflict-free
// for the ds_read_b128 lane groups (cdna_hip_programming.md T2).
//   * global->LDS copies are buffer_load_dwordx4 ... lds through three buffer descriptors (source 1, source 2, weights): a
//     lane's byte offset is a 32-bit VGPR that only changes when the conv tap (or the concat source) changes; the K advance
//     is the scalar soffset.  Predicated-off lanes (conv halo, ragged M/N/K) carry the offset 0x80000000, which is out of
//     range for every descriptor (sizes are checked < 2 GiB on the host) and therefore reads as zero.  The XOR swizzle is
//     applied on the SOURCE address (the LDS image of an LDS-DMA is lane-linear).
//   * The MFMA operands are swapped (weights as the "A" operand) so that each lane ends up holding four consecutive output
//     channels of one output row; the epilogue stages the fp32 tile through LDS so that stores / residual loads are 16 bytes
//     per lane with consecutive lanes on consecutive bytes of a row.
//   * split-K (small-M, long-K problems: the 8x8-latent level): blockIdx.y owns a contiguous range of K tiles and writes its
//     raw fp32 accumulators to a caller-provided workspace slab; splitk_reduce_kernel adds the slabs in fixed order (bit-
//     reproducible) and applies the epilogue.
// The library keeps no tuning state: the tile configuration and the split factor are either chosen here (measured per-shape
// table gemm_tuned.h, then rules) or forced per call through mv_gemm_desc.cfg / .splitk.
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
    int n_major;               // weight-stationary order: an XCD's contiguous id range = a few n-tiles x ALL m-tiles (mv_gemm_desc.tile_order)
    int nsplit, kt_per_split;  // split-K: blockIdx.y owns K tiles [y * kt_per_split, (y + 1) * kt_per_split)
    float* ws;                 // split-K workspace [nsplit][M][N] fp32
    // LayerNorm folded into the projection (LINEAR mode): w holds gamma-scaled weights, the block forms the row statistics of its A
    // rows from the fragments it multiplies anyway, and the epilogue applies  rstd_m * (acc - mean_m * colsum_n) + colbias_n
    const float* ln_colsum;    // [N] fp32: sum_k w[n][k] (of the fp16 values the MFMA sees), or nullptr
    const float* ln_colbias;   // [N] fp32: sum_k beta_k W[n][k] + bias_n
    float ln_eps;
    // column statistics of the OUTPUT for the GroupNorm that reads it next (wide epilogue, one K slice, no GEGLU): every wave adds
    // the sum and the sum of squares of the fp16 values it stores, per column over its 16 TM rows, and writes them to
    // colstats[mw0 / (16 TM)][n][2] -- mv_groupnorm_cs_f16 folds them instead of re-reading the tensor
    float* colstats;
    // two-fp16 carry of the residual stream's identity path (wide epilogue, one K slice, no GEGLU / LayerNorm fold): the sum
    // s = value + residual + residual_lo is formed in fp32 and stored as c = fp16(s), c_lo = fp16(s - c); layers read c only
    const half_t* residual_lo;  // [M][ldr] or nullptr
    half_t* c_lo;               // [M][ldc] or nullptr (no carry)
    // one weight matrix per GROUP of rows (LINEAR mode): rows [g * w_group_rows, (g + 1) * w_group_rows) multiply w + g * N * K -- a
    // GroupNorm folded into the projection behind it has one scaled copy of the weights per normalised item (mv_groupnorm_cs_fold_linear_f16).
    // A block tile never straddles two groups (the launcher picks a tile whose rows divide w_group_rows).  0 = one matrix.
    int w_group_rows;
    const half_t* rowbias_lo;   // second fp16 half of the row bias (same layout; the sum keeps ~22 bits), or nullptr
};

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

// the same with row statistics of the A tile: the wave adds its fragments' sum and sum of squares -- 8 v_dot2_f32_f16 per
// fragment, fp32 accumulation -- into per-lane partials (lane (l15, g) holds k = 8 g .. 8 g + 7 of row 16 i + l15).
// MV_LN_VARIANT (experiment builds; the default is what measured fastest, profiles/r03c):
//   0  the WGN waves that share a row block deal the 32-deep steps round-robin; statistics ahead of the step's MFMAs
//   1  the same deal, statistics behind the step's MFMAs (an in-order wave issues its MFMAs first, the dot products run in their shadow)
//   2  every wave keeps the statistics of all of its fragments (no deal, no LDS exchange), issued behind the MFMAs and interleaved
//      with them by the scheduler (two dot products per MFMA)
#ifndef MV_LN_VARIANT
#define MV_LN_VARIANT 0
#endif
template <int TM>
__device__ __forceinline__ void ln_stats_step(const half8v (&af)[TM], float (&s1)[TM], float (&s2)[TM]) {
#ifdef MV_LN_DOT2
    const half2v ones = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float a0 = s1[i], b0 = s2[i], a1 = 0.f, b1 = 0.f;  // two chains per statistic: half the dependent latency
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const half2v h0 = {af[i][2 * e], af[i][2 * e + 1]};
            const half2v h1 = {af[i][2 * e + 2], af[i][2 * e + 3]};
            a0 = __builtin_amdgcn_fdot2(h0, ones, a0, false);
            a1 = __builtin_amdgcn_fdot2(h1, ones, a1, false);
            b0 = __builtin_amdgcn_fdot2(h0, h0, b0, false);
            b1 = __builtin_amdgcn_fdot2(h1, h1, b1, false);
        }
        s1[i] = a0 + a1;
        s2[i] = b0 + b1;
    }
#else
    // mixed-precision FMAs (v_fma_mix_f32: fp16 operands, fp32 accumulate), two chains per statistic.  NOT v_dot2c_f32_f16: next to
    // MFMAs it produced sporadically corrupted accumulators on the MI355X (profiles/r03d: single elements of single 16 x 16 tiles,
    // run-to-run different) -- the in-place dot product overwrote registers an in-flight MFMA was still reading as its C operand.
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float a0 = s1[i], b0 = s2[i], a1 = 0.f, b1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float x0 = (float)af[i][e], x1 = (float)af[i][e + 1];
            a0 += x0;
            a1 += x1;
            b0 = fmaf(x0, x0, b0);
            b1 = fmaf(x1, x1, b1);
        }
        s1[i] = a0 + a1;
        s2[i] = b0 + b1;
    }
#endif
}

template <int TM, int TN, int WGN>
__device__ __forceinline__ void mma_tile_ln(const half_t* cA, const half_t* cB, float4v (&acc)[TM][TN], int a_row0, int b_row0, int swz,
                                            int g, float (&s1)[TM], float (&s2)[TM], int kstep0, int wn) {
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
#if MV_LN_VARIANT == 0
        if (((kstep0 + kk) & (WGN - 1)) == wn) ln_stats_step<TM>(af, s1, s2);  // wave-uniform
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
#if MV_LN_VARIANT == 1
        if (((kstep0 + kk) & (WGN - 1)) == wn) ln_stats_step<TM>(af, s1, s2);  // wave-uniform
#elif MV_LN_VARIANT == 2
        ln_stats_step<TM>(af, s1, s2);
        // issue order of this step: one MFMA, then two of the 10 TM VALU instructions of the statistics, ...
#pragma unroll
        for (int m = 0; m < TM * TN; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // 2 VALU
        }
#endif
    }
}

constexpr unsigned kOOB = 0x80000000u;

struct GemmArgs2 {
    GemmArgs g;
    unsigned a_bytes, a2_bytes, w_bytes;
    int wide;  // 1: N, ldc, ldr, ldrb multiples of 8 and 16-byte aligned pointers -> LDS-staged 16-byte epilogue
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
                if (p.rowbias_lo) {
                    b = *reinterpret_cast<const half4v*>(p.rowbias_lo + grp * p.ldrb + n);
                    v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                }
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

// LDS-staged epilogue of one wave's (16*TM) x (16*TN) accumulator tile.  The MFMA accumulator layout gives a lane 4 channels of
// ONE row (a wave store instruction would touch 16 rows x 32 bytes); staging the tile through LDS turns it into 16-byte-per-lane
// accesses whose consecutive lanes cover consecutive bytes of a row (residual loads and output stores in whole lines).
// Block timelines (profiles/r02l) showed the previous form of this epilogue -- fp32 staging, per-store index division, 64-bit
// address arithmetic and bounds branches, ~1100 VALU instructions per wave, and every pass's residual / row-bias loads waiting
// (in-order vmcnt) behind the previous pass's stores -- at 6-12 us per 256-row block, a third of a K = 320 launch.  This one:
//   * one pass = one 16-row accumulator row i, staged as fp16 (value after bias / row bias / alpha / activation, rounded once -- the
//     rounding torch applies to the layer output before its residual add) into one of two per-wave buffers: pass i+1 is written
//     BEFORE pass i is read back, so the write -> read turnaround of a pass hides under the conversions of the next one (DS
//     operations of a wave execute in order; the wave barriers are compiler / simulator ordering only);
//   * read phase: lane -> (row r0 = lane / CPR, 16-byte chunk lane % CPR) fixed for the whole tile, CPR chunks per row, RPI = 64 / CPR
//     rows per iteration: no division, LDS and global offsets advance by constants; uniform base pointers + 32-bit lane offsets;
//   * bias and row bias are already in the accumulators (the K loop starts from them); the residual chunks of pass i+1 are
//     requested BEFORE the stores of pass i are issued, so waiting for them need not wait for a store; the residual add is a
//     packed fp16 add (what torch's fp16 `x + residual` computes);
//   * bounds checks only on edge tiles (`full` is block-uniform).
// GEGLU: even accumulator tiles hold values, odd tiles gates; the output has 8*TN columns per wave.  Each wave owns a private
// staging region, so only wave-level ordering is needed between its write and read phases.
template <int TN, bool GEGLU> struct EpiGeom {
    static constexpr int W = GEGLU ? 8 * TN : 16 * TN;  // output columns of the wave tile
    static constexpr int LDW = W + 8;                   // staging row (halfs): conflict-free 8-byte row-strided writes
    static constexpr int CPR = W / 8;                   // 16-byte chunks per row
    static constexpr int RPI = 64 / CPR;                // rows per read iteration
    static constexpr int KI = (16 + RPI - 1) / RPI;     // read iterations per pass
    static constexpr int WAVE_HALFS = 2 * 16 * LDW;     // two buffers
};

// LN: the accumulators hold x . (gamma W)^T of the RAW rows; a pass first applies v = rstd_m * acc - (rstd_m * mean_m) * colsum_n +
// colbias_n (ln_r[i] = rstd, ln_mr[i] = rstd * mean of row 16 i + l15; the column vectors are read in the accumulator layout).
// CARRY: a pass is staged as TWO fp16 values per element (hi = fp16(v), lo = fp16(v - hi): v to ~22 bits) in two buffers, the read
// phase forms s = hi + lo + residual + residual_lo in fp32 and stores c = fp16(s), c_lo = fp16(s - c) -- the identity path of the
// residual stream keeps ~22 bits across the network while every layer reads the fp16 tensor c (column statistics are those of c).
template <int TM, int TN, bool GEGLU, bool RES, bool LN = false, bool CARRY = false>
__device__ __forceinline__ void epilogue_staged(const GemmArgs& p, float4v (&acc)[TM][TN], half_t* stg_base, int wave, int mw0,
                                                int nw0, int lane, float alpha, int Mi, bool full, const float* ln_r = nullptr,
                                                const float* ln_mr = nullptr, int nacc0 = 0) {
    using G = EpiGeom<TN, GEGLU>;
    constexpr int W = G::W, LDW = G::LDW, CPR = G::CPR, RPI = G::RPI, KI = G::KI;
    static_assert(!CARRY || (!GEGLU && !LN), "the carry is a feature of the plain / residual epilogue");
    constexpr int LO = G::WAVE_HALFS;  // CARRY: the lo buffers sit behind the wave's two hi buffers
    half_t* stg = stg_base + wave * ((CARRY ? 2 : 1) * G::WAVE_HALFS);
    const int l15 = lane & 15, g = lane >> 4;
    const int Nout = GEGLU ? (p.N >> 1) : p.N;
    // read-phase geometry of this lane
    const int r0 = lane / CPR, ch = lane - r0 * CPR;
    const bool lane_on = r0 < RPI;
    const int ncol = nw0 + 8 * ch;                       // first output column of this lane's chunk
    const bool col_ok = full || ncol < Nout;
    const half_t* rd = stg + (r0 < 16 ? r0 : 0) * LDW + 8 * ch;  // (lanes past the pass rows read row 0, their values are unused)
    half_t* wr = stg + l15 * LDW + 4 * g;
    half_t* const cbase = p.c + (long)mw0 * p.ldc + nw0;                                   // wave-uniform
    const half_t* const rbase = RES ? p.residual + (long)mw0 * p.ldr + nw0 : nullptr;
    half_t* const clo = CARRY ? p.c_lo + (long)mw0 * p.ldc + nw0 : nullptr;
    const half_t* const rlo = (CARRY && RES && p.residual_lo) ? p.residual_lo + (long)mw0 * p.ldr + nw0 : nullptr;  // (wave-uniform)
    const unsigned coff = (unsigned)(r0 * p.ldc + 8 * ch), roff = (unsigned)(r0 * p.ldr + 8 * ch);
    const bool silu = p.act == MV_ACT_SILU;

    // LN: colsum / colbias of this lane's accumulator columns nacc0 + 16 j + 4 g .. + 3 (nacc0 = first accumulator column of the wave
    // tile; with GEGLU the packed [value tile | gate tile] order is the accumulator's own)
    float4v lcs[LN ? TN : 1], lcb[LN ? TN : 1];
    if constexpr (LN) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = nacc0 + 16 * j + 4 * g;
            const bool ok = n < p.N;
            lcs[j] = ok ? *reinterpret_cast<const float4v*>(p.ln_colsum + n) : float4v{0.f, 0.f, 0.f, 0.f};
            lcb[j] = ok ? *reinterpret_cast<const float4v*>(p.ln_colbias + n) : float4v{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto ln_affine = [&](int i, int j) __attribute__((always_inline)) -> float4v {
        if constexpr (LN) {
            // rstd * acc - (rstd * mean) * colsum + colbias, element by element (MV_FMA_SCALAR: see common.h)
            const float r = ln_r[i], nmr = -ln_mr[i];
            float4v v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = MV_FMA_SCALAR(acc[i][j][e], r, MV_FMA_SCALAR(lcs[j][e], nmr, lcb[j][e]));
            return v;
        } else {
            return acc[i][j];
        }
    };
    // column statistics (see GemmArgs::colstats): this lane's 8 columns over the rows it stores
    const bool cs_on = !GEGLU && p.colstats != nullptr;
    float cs1[GEGLU ? 1 : 8], cs2[GEGLU ? 1 : 8];
    if constexpr (!GEGLU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) cs1[e] = cs2[e] = 0.f;
    }
    // (the 160-accumulator tile's carry form has no registers for a second set of residual chunks: its request for pass i + 1 goes
    // out right behind pass i's stores into the SAME registers -- the write phase of pass i + 2 is the latency window)
    constexpr int NRB = (CARRY && TM * TN >= 40) ? 1 : 2;
    half8v res[NRB][KI];
    half8v resl[(CARRY && RES) ? NRB : 1][(CARRY && RES) ? KI : 1];
    auto request = [&](int i, half8v* rs, half8v* rsl) {  // residual chunks of pass i, in the read layout
        if constexpr (RES) {
#pragma unroll
            for (int k = 0; k < KI; ++k) {
                const int row = r0 + k * RPI;
                rs[k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (CARRY) rsl[k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
                if (lane_on && row < 16 && col_ok && (full || mw0 + 16 * i + row < Mi)) {
                    rs[k] = *reinterpret_cast<const half8v*>(rbase + (roff + (unsigned)((16 * i + k * RPI) * p.ldr)));
                    if constexpr (CARRY) {
                        if (rlo) rsl[k] = *reinterpret_cast<const half8v*>(rlo + (roff + (unsigned)((16 * i + k * RPI) * p.ldr)));
                    }
                }
            }
        }
    };
    auto write_pass = [&](int i) __attribute__((always_inline)) {
        half_t* wrow = wr + (i & 1) * (16 * LDW);
        if constexpr (GEGLU) {
#pragma unroll
            for (int j = 0; j < TN; j += 2) {
                const float4v v = ln_affine(i, j), gt = ln_affine(i, j + 1);
                *reinterpret_cast<half4v*>(wrow + 8 * j) = half4v{(half_t)(v[0] * mv_gelu(gt[0])), (half_t)(v[1] * mv_gelu(gt[1])),
                                                                  (half_t)(v[2] * mv_gelu(gt[2])), (half_t)(v[3] * mv_gelu(gt[3]))};
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4v v = ln_affine(i, j) * alpha;
                if (silu) {
                    v[0] = mv_silu(v[0]); v[1] = mv_silu(v[1]); v[2] = mv_silu(v[2]); v[3] = mv_silu(v[3]);
                }
                const half4v hv = half4v{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4v*>(wrow + 16 * j) = hv;
                if constexpr (CARRY)
                    *reinterpret_cast<half4v*>(wrow + LO + 16 * j) = half4v{(half_t)(v[0] - (float)hv[0]), (half_t)(v[1] - (float)hv[1]),
                                                                            (half_t)(v[2] - (float)hv[2]), (half_t)(v[3] - (float)hv[3])};
            }
        }
    };
    request(0, res[0], resl[0]);
    write_pass(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int buf = i & 1;
        // ---- write phase of pass i+1 (other buffer; its last readers, pass i-1, come earlier in program order) ----
        __builtin_amdgcn_wave_barrier();
        if (i + 1 < TM) write_pass(i + 1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- the next pass's residual requests go out ahead of this pass's stores ----
        if constexpr (NRB == 2) {
            if (i + 1 < TM) request(i + 1, res[buf ^ 1], resl[(CARRY && RES) ? (buf ^ 1) : 0]);
        }
        const int rb = NRB == 2 ? buf : 0;
        // ---- read phase of pass i: every LDS read of the pass first (lanes past the pass rows read row 0, their values unused) ----
        const half_t* rrow = rd + buf * (16 * LDW);
        half8v ov[KI], ovl[CARRY ? KI : 1];
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int row = r0 + k * RPI;
            ov[k] = *reinterpret_cast<const half8v*>(rrow + ((lane_on && row < 16) ? k * RPI * LDW : 0));
            if constexpr (CARRY) ovl[k] = *reinterpret_cast<const half8v*>(rrow + LO + ((lane_on && row < 16) ? k * RPI * LDW : 0));
        }
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int row = r0 + k * RPI;
            const bool valid = lane_on && row < 16 && col_ok && (full || mw0 + 16 * i + row < Mi);
            half8v o = half8v{0, 0, 0, 0, 0, 0, 0, 0};
            if (valid) {
                if constexpr (CARRY) {
                    half8v ol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float sv = (float)ov[k][e] + (float)ovl[k][e];
                        if constexpr (RES) sv = (sv + (float)res[rb][k][e]) + (float)resl[rb][k][e];
                        o[e] = (half_t)sv;
                        ol[e] = (half_t)(sv - (float)o[e]);
                    }
                    *reinterpret_cast<half8v*>(cbase + (coff + (unsigned)((16 * i + k * RPI) * p.ldc))) = o;
                    *reinterpret_cast<half8v*>(clo + (coff + (unsigned)((16 * i + k * RPI) * p.ldc))) = ol;
                } else {
                    o = ov[k];
                    if constexpr (RES) o += res[rb][k];
                    *reinterpret_cast<half8v*>(cbase + (coff + (unsigned)((16 * i + k * RPI) * p.ldc))) = o;
                }
            }
            if constexpr (!GEGLU) {
                if (cs_on && valid) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)o[e];
                        cs1[e] += f;
                        cs2[e] = fmaf(f, f, cs2[e]);
                    }
                }
            }
        }
        if constexpr (NRB == 1) {
            if (i + 1 < TM) request(i + 1, res[0], resl[0]);
        }
        asm volatile("" ::: "memory");
    }
    if constexpr (!GEGLU) {
        // the RPI lanes that hold the same chunk of different rows meet in the wave's staging area ([RPI][W] floats, sums first,
        // then squares -- fixed order, bit-reproducible); lane c (and c + 64) owns column c of the wave tile
        static_assert(RPI * W * (int)sizeof(float) <= G::WAVE_HALFS * (int)sizeof(half_t), "column statistics do not fit the staging area");
        if (cs_on && mw0 < Mi) {
            float* const cst = reinterpret_cast<float*>(stg);
            float* const dst = p.colstats + ((long)(mw0 / (16 * TM)) * p.N + nw0) * 2;
            constexpr int CI = (W + 63) / 64;
            float tot[2][CI];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __builtin_amdgcn_wave_barrier();
                if (lane_on) {
                    const float* src = h ? cs2 : cs1;
                    *reinterpret_cast<float4v*>(cst + r0 * W + 8 * ch) = float4v{src[0], src[1], src[2], src[3]};
                    *reinterpret_cast<float4v*>(cst + r0 * W + 8 * ch + 4) = float4v{src[4], src[5], src[6], src[7]};
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const int c = lane + 64 * ci;
                    float t = 0.f;
                    if (c < W) {
#pragma unroll
                        for (int r = 0; r < RPI; ++r) t += cst[r * W + c];
                    }
                    tot[h][ci] = t;
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const int c = lane + 64 * ci;
                if (c < W && (full || nw0 + c < Nout)) *reinterpret_cast<float2v*>(dst + 2 * c) = float2v{tot[0][ci], tot[1][ci]};
            }
        }
    }
}

// Block timeline (experiment builds only, -DMV_TIMELINE; tools/gpu_gemm_timeline.py): thread 0 of every block of the first
// K slice records the 100 MHz wall clock at entry / prologue issued / first tile landed / K loop done / epilogue issued /
// stores drained, plus its HW_ID and XCC_ID, into a device array the tool reads back.
#ifdef MV_TIMELINE
constexpr int kTlBlocks = 32768;
__device__ unsigned long long mv_tl_buf[kTlBlocks * 8];
#define MV_TL(i) do { if (tl_on) tl[i] = wall_clock64(); } while (0)
#define MV_TL_FLUSH() do { if (tl_on) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tl[5] = wall_clock64(); \
        unsigned long long* o = mv_tl_buf + (size_t)blockIdx.x * 8; for (int i_ = 0; i_ < 6; ++i_) o[i_] = tl[i_]; \
        o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } while (0)
#else
#define MV_TL(i) ((void)0)
#define MV_TL_FLUSH() ((void)0)
#endif

template <int MODE, int TM, int TN, int WGM, int WGN, int SCHED, bool LNF = false, bool CARRY = false>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm2_kernel(const GemmArgs2 q) {
    static_assert(!LNF || MODE == MV_GEMM_LINEAR, "LayerNorm folding is a LINEAR-mode feature");
    static_assert(!(LNF && CARRY), "no carry behind a LayerNorm-folded projection");
    static_assert((WGN & (WGN - 1)) == 0, "the k-step deal of the row statistics needs a power-of-two WGN");
    constexpr int NW = WGM * WGN;
    // SCHED: 0 = two LDS stages behind __syncthreads (two blocks per CU overlap each other's stalls);
    //        3 = three stages behind counted waits (two K tiles in flight; for one-block-per-CU tiles and latency-bound grids)
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN, BK = 64;
    constexpr int RP = 8;  // tile rows per 1-KiB LDS-DMA piece (8 rows of 128 B)
    constexpr int CA = BM / RP, CB = BN / RP;
    constexpr int AI = (CA + NW - 1) / NW, BI = (CB + NW - 1) / NW;
    const GemmArgs& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = (SCHED == 3) ? 3 : 2;  // LDS stages
    half_t* sA = reinterpret_cast<half_t*>(smem);  // [NST][BM*BK]
    half_t* sB = sA + NST * BM * BK;                // [NST][BN*BK]

    const int tid = threadIdx.x;
#ifdef MV_TIMELINE
    unsigned long long tl[6] = {0, 0, 0, 0, 0, 0};
    const bool tl_on = tid == 0 && blockIdx.y == 0 && blockIdx.x < kTlBlocks;
#endif
    MV_TL(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int l15 = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int id = mv_xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    // (n_major, block-uniform: small M under a large weight matrix -- every XCD streams ITS n-tiles' weights only)
    mv_tile_order(id, p.tiles_m, p.tiles_n, p.n_major ? -1 : MV_TILE_GROUP, &tile_m, &tile_n);
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;
    const int Mi = (int)p.M;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, q.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? q.a2_bytes : 0u, 0x00020000);
    const half_t* const wmat = p.w_group_rows > 0 ? p.w + (long)(m0 / p.w_group_rows) * p.N * p.K : p.w;   // (block-uniform)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)wmat, 0, q.w_bytes, 0x00020000);

    // LDS-DMA lane geometry (the LDS image of a piece is lane-linear, so the swizzle lives on the SOURCE address):
    //   lane -> row lane/8, slot lane%8, fetches logical 16-byte chunk slot ^ row
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);

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
    // ---- weight rows owned by this lane (LDS row r of the B tile holds weight row n0 + r) ----
    unsigned b_off[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int c = wave + NW * j;
        const int n = n0 + RP * c + lrow;
        const bool ok = (c < CB) && (n < p.N);
        b_off[j] = ok ? ((unsigned)n * (unsigned)p.K + lsl * 8u) * 2u : kOOB;
    }

    // |alpha| (a device scalar) is requested here so that its latency hides under the prologue
    const float alpha = p.alpha ? fabsf(*p.alpha) : 1.0f;
    // the accumulators start from bias + row bias, requested right AFTER the prologue's LDS-DMA issue (in-order vmcnt: waiting for
    // them is waiting for the first tile, which the K loop does anyway), so the epilogue has no loads ahead of its first LDS write
    // and nothing is held in registers across the K loop; a split-K slice starts from zero, the reduce kernel adds both.  Lane
    // (l15, g) holds columns 16 j + 4 g .. + 3 of rows 16 i + l15 (GEGLU: the packed [value tile | gate tile] column order is the
    // accumulator's own).  The narrow epilogue adds them itself.
    float4v acc[TM][TN];
    float ln_s1[LNF ? TM : 1], ln_s2[LNF ? TM : 1];  // LNF: per-lane partial row sums / sums of squares of the A rows 16 i + l15
#pragma unroll
    for (int i = 0; i < (LNF ? TM : 1); ++i) ln_s1[i] = ln_s2[i] = 0.f;
    auto init_acc = [&]() __attribute__((always_inline)) {
        const int nw0 = n0 + wn * 16 * TN;
        const bool init = q.wide && p.nsplit <= 1 && !LNF;  // (LNF: the bias is part of ln_colbias, applied after the row affine)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float4v b0 = float4v{0.f, 0.f, 0.f, 0.f};
            const int n = nw0 + 16 * j + 4 * g;
            if (init && p.bias && n < p.N) {
                const half4v b = *reinterpret_cast<const half4v*>(p.bias + n);
                b0 = float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = b0;
        }
        if (init && p.rowbias) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * 16 * TM + 16 * i + l15;
                const long rbo = (long)((m < Mi ? m : Mi - 1) / p.rows_per_group) * p.ldrb + nw0 + 4 * g;
                const half_t* rbp = p.rowbias + rbo;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (nw0 + 16 * j + 4 * g < p.N) {
                        const half4v b = *reinterpret_cast<const half4v*>(rbp + 16 * j);
                        acc[i][j] += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                        if (p.rowbias_lo) {   // (block-uniform)
                            const half4v bl = *reinterpret_cast<const half4v*>(p.rowbias_lo + rbo + 16 * j);
                            acc[i][j] += float4v{(float)bl[0], (float)bl[1], (float)bl[2], (float)bl[3]};
                        }
                    }
                }
            }
        }
    };

    // K tiles of this block: all of them, or the slice blockIdx.y of a split-K launch
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = (int)blockIdx.y * p.kt_per_split;
    const int nk = (kt0 + p.kt_per_split < nk_all) ? (kt0 + p.kt_per_split) : nk_all;  // one past the last K tile of this block
    const bool ragged = (p.K & (BK - 1)) != 0;
    unsigned a_off[AI];  // current byte offsets (valid for the tile about to be issued)
    // K walk.  LINEAR: k tile j = columns [64 j, 64 j + 64).  Convolutions: CHANNEL-major with the taps innermost -- k tile j is
    // (64-channel chunk j / TAPS, tap j % TAPS) -- so that the 9 (3) shifted views of one 64-channel slab of the activations are
    // fetched back to back while the slab is still in the XCD's L2 (tap-major, the walk of rounds 1-3, re-fetched every tap from the
    // fabric: the ~32 blocks of an XCD stream ~10 MB between two taps of the same block; profiles/r03ah_pmc_by_problem.log: 3.5-7.6 x
    // the algorithmic bytes on the convolutions).  The weights stay packed tap-major ([N][taps][cin]): tile (chunk, tap) reads their
    // columns tap * cin + 64 chunk.  A tap moves a lane's source row by a UNIFORM number of rows (per lane with the fused upsample:
    // the parity of its output pixel decides), so a tile's offsets are the centre-tap offsets + a shift, masked by the lane's
    // per-tap validity bits (halo / frame range) -- no division or bounds arithmetic inside the walk.
    constexpr int TAPS = MODE == MV_GEMM_CONV3X3 ? 9 : MODE == MV_GEMM_TCONV3 ? 3 : 1;
    int chunk = 0, tap = 0;   // cursor of the next tile to issue (conv modes; cin % 64 == 0)
    int kc = 0;               // LINEAR: channel cursor
    if (MODE == MV_GEMM_LINEAR) {
        kc = kt0 * BK;
    } else {
        chunk = kt0 / TAPS;
        tap = kt0 - chunk * TAPS;
    }
    // conv modes: per piece the centre-tap source row, the validity bits of the taps and (upsample) the output pixel's parities
    int a_rowc[AI];
    unsigned a_mask[AI];
    unsigned a_base[AI];      // centre-tap byte offsets for the CURRENT source
    if (MODE != MV_GEMM_LINEAR) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            unsigned m = 0;
            if (MODE == MV_GEMM_CONV3X3) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int iy = a_y[i] + tp / 3 - 1, ix = a_x[i] + tp % 3 - 1;
                    const bool ok = p.upsample ? (iy >= 0 && iy < 2 * p.hin && ix >= 0 && ix < 2 * p.win)
                                               : (iy >= 0 && iy < p.hin && ix >= 0 && ix < p.win);
                    m |= (a_ok[i] && ok) ? (1u << tp) : 0u;
                }
                a_rowc[i] = p.upsample ? a_row[i] + (a_y[i] >> 1) * p.win + (a_x[i] >> 1) : a_row[i] + a_y[i] * p.win + a_x[i];
            } else {
#pragma unroll
                for (int tp = 0; tp < 3; ++tp) {
                    const int tt = a_y[i] + tp - 1;
                    m |= (a_ok[i] && tt >= 0 && tt < p.t) ? (1u << tp) : 0u;
                }
                a_rowc[i] = a_row[i];
            }
            a_mask[i] = m;
            a_base[i] = 0;
        }
    }
    bool rebuild = true;   // the centre-tap offsets must be (re)built before the next issue (start, source switch)
    bool cur_second = false;

    bool sec = false;      // source of the tile about to be issued
    unsigned soa = 0;      // its scalar byte offset inside a source row
    unsigned sob = 0;      // scalar byte offset of its weight columns inside a weight row

    // cursor step (the only branchy part of the K loop): fixes (source, offsets) of the next tile to issue
    auto prepare = [&]() {
        if (MODE == MV_GEMM_LINEAR) {
            const bool second = (p.a2 != nullptr) && (kc >= p.c1);
            if (rebuild || (second && kc == p.c1)) {  // source changed: rebuild the lane offsets (wave-uniform branch)
                rebuild = false;
                const unsigned ldb = (unsigned)(second ? p.lda2 : p.lda) * 2u;
#pragma unroll
                for (int i = 0; i < AI; ++i) a_off[i] = a_ok[i] ? (unsigned)a_row[i] * ldb + lsl * 16u : kOOB;
            }
            sec = second;
            soa = (unsigned)(second ? kc - p.c1 : kc) * 2u;
            sob = (unsigned)kc * 2u;
            kc += BK;
            return;
        }
        const int c0 = chunk * BK;  // first channel of the tile (of the concatenation)
        const bool second = (p.a2 != nullptr) && (c0 >= p.c1);
        const int ldh = second ? p.lda2 : p.lda;
        if (rebuild || second != cur_second) {  // (wave-uniform branch)
            rebuild = false;
            cur_second = second;
#pragma unroll
            for (int i = 0; i < AI; ++i) a_base[i] = (unsigned)a_rowc[i] * (unsigned)(ldh * 2) + lsl * 16u;
        }
        int dy = 0, dx = 0;
        if (MODE == MV_GEMM_CONV3X3) {
            dy = tap / 3 - 1;
            dx = tap - (tap / 3) * 3 - 1;
        } else {
            dy = tap - 1;
        }
        const int ushift = (MODE == MV_GEMM_CONV3X3 ? dy * p.win + dx : dy * p.hw) * (ldh * 2);  // uniform row shift in bytes
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            int sh = ushift;
            if (MODE == MV_GEMM_CONV3X3 && p.upsample) {
                // ((oy + dy) >> 1) - (oy >> 1): -1 for dy = -1 on an even oy, +1 for dy = +1 on an odd oy, else 0 (same in x)
                const int py = a_y[i] & 1, px = a_x[i] & 1;
                const int sy = dy < 0 ? py - 1 : dy > 0 ? py : 0;
                const int sx = dx < 0 ? px - 1 : dx > 0 ? px : 0;
                sh = (sy * p.win + sx) * (ldh * 2);
            }
            a_off[i] = ((a_mask[i] >> tap) & 1u) ? (unsigned)((int)a_base[i] + sh) : kOOB;
        }
        sec = second;
        soa = (unsigned)(second ? c0 - p.c1 : c0) * 2u;
        sob = (unsigned)(tap * p.cin + c0) * 2u;
        if (++tap == TAPS) {
            tap = 0;
            ++chunk;
        }
    };
    // branch-free issue of LDS-DMA piece d (0 .. AI+BI-1) of tile kt into stage `buf`
    auto issue_piece = [&](int d, int buf, int kt) {
        const bool kcut = ragged && (kt == nk_all - 1) && ((int)(kt * BK + lsl * 8) >= p.K);  // ragged K: zero past K (LINEAR)
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
                rW, (__attribute__((address_space(3))) void*)(sB + buf * (BN * BK) + c * (RP * BK)), 16, (int)vo, (int)sob, 0, 0);
        }
    };
    auto issue = [&](int buf, int kt) {
#pragma unroll
        for (int d = 0; d < AI + BI; ++d) issue_piece(d, buf, kt);
    };

    const int a_row0 = wm * 16 * TM + l15;
    const int b_row0 = wn * 16 * TN + l15;
    const int swz = l15 & 7;

    auto mma_stage = [&](int st, int kt) __attribute__((always_inline)) {
        if constexpr (LNF) mma_tile_ln<TM, TN, WGN>(sA + st * (BM * BK), sB + st * (BN * BK), acc, a_row0, b_row0, swz, g, ln_s1, ln_s2, 2 * kt, wn);
        else mma_tile<TM, TN>(sA + st * (BM * BK), sB + st * (BN * BK), acc, a_row0, b_row0, swz, g);
    };
    if constexpr (SCHED == 3) {
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
            if (kt0 + t0 < nk) {
                prepare();
                issue(t0, kt0 + t0);
            }
        }
        MV_TL(1);
        init_acc();
        int cur = 0;
        for (int kt = kt0; kt < nk; ++kt) {
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
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;  // over-waiting is always safe
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef MV_TIMELINE
            if (kt == kt0) MV_TL(2);
#endif
            if (kt + NST - 1 < nk) {
                prepare();
                issue(cur == 0 ? NST - 1 : cur - 1, kt + NST - 1);  // (cur + NST - 1) % NST: the stage tile kt-1 just left
            }
            mma_stage(cur, kt);
            cur = (cur == NST - 1) ? 0 : cur + 1;
        }
    } else {
        prepare();
        issue(0, kt0);
        MV_TL(1);
        init_acc();
        __syncthreads();  // drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        MV_TL(2);
        for (int kt = kt0; kt < nk - 1; ++kt) {
            const int cur = (kt - kt0) & 1;
            prepare();
            issue(cur ^ 1, kt + 1);
            mma_stage(cur, kt);
            __syncthreads();
        }
        mma_stage((nk - 1 - kt0) & 1, nk - 1);
    }

    // ---- epilogue ----
    MV_TL(3);
    const int nw0 = n0 + wn * 16 * TN;
    const int mw0 = m0 + wm * 16 * TM;
    if (p.nsplit > 1) {
        // split-K slice: raw fp32 accumulators into this slice's slab; lane = (row l15, 4 consecutive columns 4g..4g+3) per tile
        float* slab = p.ws + (long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = mw0 + 16 * i + l15;
            if (m >= Mi) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nw0 + 16 * j + 4 * g;
                if (n < p.N) *reinterpret_cast<float4v*>(slab + (long)m * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    if (q.wide) {
        // LNF: the waves' row-statistic partials meet in LDS behind the operand / staging area: [wave][16 TM rows][sum, sum of squares]
        constexpr int kOpsBytes = NST * 64 * (BM + BN) * (int)sizeof(half_t);
        constexpr int kEpiBytes = NW * EpiGeom<TN, false>::WAVE_HALFS * (int)sizeof(half_t);
        float* const sst = reinterpret_cast<float*>(smem + (kOpsBytes > kEpiBytes ? kOpsBytes : kEpiBytes));
        if constexpr (LNF) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float a = ln_s1[i], b = ln_s2[i];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);  // the 4 lane groups hold k = 8 g .. 8 g + 7 of the row
                b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
                ln_s1[i] = a;
                ln_s2[i] = b;
#if MV_LN_VARIANT != 2
                if (g == 0) {
                    sst[(wave * 16 * TM + 16 * i + l15) * 2] = a;
                    sst[(wave * 16 * TM + 16 * i + l15) * 2 + 1] = b;
                }
#endif
            }
        }
        __syncthreads();  // every wave is done reading the operand tiles: their LDS is reused as the staging buffers
        float ln_r[LNF ? TM : 1], ln_mr[LNF ? TM : 1];
        if constexpr (LNF) {
            const float inv_k = 1.0f / (float)p.K;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float a = 0.f, b = 0.f;
#if MV_LN_VARIANT == 2
                a = ln_s1[i];
                b = ln_s2[i];
#else
#pragma unroll
                for (int w2 = 0; w2 < WGN; ++w2) {  // fixed order: bit-reproducible
                    a += sst[((wm * WGN + w2) * 16 * TM + 16 * i + l15) * 2];
                    b += sst[((wm * WGN + w2) * 16 * TM + 16 * i + l15) * 2 + 1];
                }
#endif
                const float mean = a * inv_k;
                const float var = fmaxf(b * inv_k - mean * mean, 0.f);
                ln_r[i] = rsqrtf(var + p.ln_eps);
                ln_mr[i] = ln_r[i] * mean;

            }
        }
        half_t* stg = reinterpret_cast<half_t*>(smem);  // (the launcher sizes the dynamic LDS as max(operand stages, staging area))
        const bool full = m0 + BM <= Mi && n0 + BN <= p.N;
        if constexpr (CARRY) {
            if (p.residual) epilogue_staged<TM, TN, false, true, false, true>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi, full);
            else epilogue_staged<TM, TN, false, false, false, true>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi, full);
        } else if (p.geglu) {
            if constexpr ((TN & 1) == 0) epilogue_staged<TM, TN, true, false, LNF>(p, acc, stg, wave, mw0, nw0 >> 1, lane, alpha, Mi, full, ln_r, ln_mr, nw0);
        } else if (p.residual) {
            epilogue_staged<TM, TN, false, true, LNF>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi, full, ln_r, ln_mr, nw0);
        } else {
            epilogue_staged<TM, TN, false, false, LNF>(p, acc, stg, wave, mw0, nw0, lane, alpha, Mi, full, ln_r, ln_mr, nw0);
        }
        MV_TL(4);
        MV_TL_FLUSH();
        return;
    }
    epilogue_narrow<TM, TN>(p, acc, mw0, nw0, lane, alpha, Mi);
    MV_TL(4);
    MV_TL_FLUSH();
}

// split-K second pass: out = epilogue(sum over slices, in slice order) -- one thread per 4 consecutive columns of a row
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const long q4 = p.N >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * q4) return;
    const long m = idx / q4;
    const int n = (int)(idx - m * q4) * 4;
    const long slab = p.M * (long)p.N;
    const float* src = p.ws + m * p.N + n;
    float4v v = *reinterpret_cast<const float4v*>(src);
    for (int s = 1; s < p.nsplit; ++s) v += *reinterpret_cast<const float4v*>(src + s * slab);
    if (p.bias) {
        const half4v b = *reinterpret_cast<const half4v*>(p.bias + n);
        v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
    if (p.rowbias) {
        half4v b = *reinterpret_cast<const half4v*>(p.rowbias + (m / p.rows_per_group) * p.ldrb + n);
        v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
        if (p.rowbias_lo) {
            b = *reinterpret_cast<const half4v*>(p.rowbias_lo + (m / p.rows_per_group) * p.ldrb + n);
            v += float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
        }
    }
    v *= p.alpha ? fabsf(*p.alpha) : 1.0f;
    if (p.act == MV_ACT_SILU) {
        v[0] = mv_silu(v[0]); v[1] = mv_silu(v[1]); v[2] = mv_silu(v[2]); v[3] = mv_silu(v[3]);
    }
    if (p.residual) {
        const half4v r = *reinterpret_cast<const half4v*>(p.residual + m * p.ldr + n);
        v += float4v{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
    }
    *reinterpret_cast<half4v*>(p.c + m * p.ldc + n) = half4v{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
}

#ifdef MV_TIMELINE
}  // namespace
extern "C" int mv_debug_timeline(void* host, int n_blocks) {
    if (n_blocks > kTlBlocks) n_blocks = kTlBlocks;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mv_tl_buf), (size_t)n_blocks * 8 * sizeof(unsigned long long)) == hipSuccess ? n_blocks : -1;
}
namespace {
#endif

int mv_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// What the 8 XCDs fetch for one K slice of a launch under a workgroup -> tile order: workgroup b runs on XCD b % 8 and produces
// the tile mv_tile_order gives it; an XCD fetches every DISTINCT A row block and every DISTINCT weight column block its workgroups
// touch once (they are co-resident and walk K together).  tools/gemm_traffic_model.py holds this model against the per-problem PMC
// bytes of profiles/r04z_pmc_by_problem.log: within 5 % on the small-M levels (every XCD streams the whole weight matrix there:
// 3.4-8.6 x the algorithmic bytes).  The weight-stationary (n-major) order is taken where it fetches at least 5 % less.
inline double order_fetch_bytes(int tiles_m, int tiles_n, int group, double a_tile, double w_tile) {
    const int nwg = tiles_m * tiles_n;
    double total = 0.0;
    for (int x = 0; x < 8; ++x) {
        // the ids of XCD x are a contiguous logical range (mv_xcd_remap): count the distinct tiles by walking it in order
        unsigned long long seen_m[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seen_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // bitmaps: up to 512 tiles per axis
        int nm = 0, nn = 0;
        for (int b = x; b < nwg; b += 8) {
            int tm, tn;
            mv_tile_order(mv_xcd_remap(b, nwg), tiles_m, tiles_n, group, &tm, &tn);
            if (!((seen_m[(tm >> 6) & 7] >> (tm & 63)) & 1ull)) { seen_m[(tm >> 6) & 7] |= 1ull << (tm & 63); ++nm; }
            if (!((seen_n[(tn >> 6) & 7] >> (tn & 63)) & 1ull)) { seen_n[(tn >> 6) & 7] |= 1ull << (tn & 63); ++nn; }
        }
        total += nm * a_tile + nn * w_tile;
    }
    return total;
}
inline bool ws_model_prefers(long M, int N, int K, int cin, int bm, int bn) {
    const int tiles_m = (int)((M + bm - 1) / bm), tiles_n = (N + bn - 1) / bn;
    if (tiles_m > 512 || tiles_n > 512 || (long)tiles_m * tiles_n > 8192) return false;  // large grids: the activations dominate anyway
    const double a_tile = (double)(M < bm ? M : bm) * cin * 2.0, w_tile = (double)(N < bn ? N : bn) * K * 2.0;
    return order_fetch_bytes(tiles_m, tiles_n, -1, a_tile, w_tile) < 0.95 * order_fetch_bytes(tiles_m, tiles_n, MV_TILE_GROUP, a_tile, w_tile);
}

template <int MODE, int TM, int TN, int WGM, int WGN, int SCHED, bool LNF = false, bool CARRY = false>
int launch_cfg2s(const GemmArgs2& a0, hipStream_t stream) {
    constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN;
    constexpr int smem_ops = (SCHED == 3 ? 3 * 64 : 2 * 64) * (BM + BN) * (int)sizeof(half_t);
    // the LDS-staged epilogue reuses the operand LDS: two 16-row fp16 buffers of (16 TN + 8) halfs per wave must fit as well
    // (CARRY: two more for the lo halves)
    constexpr int smem_epi = (CARRY ? 2 : 1) * WGM * WGN * EpiGeom<TN, false>::WAVE_HALFS * (int)sizeof(half_t);
    // LNF: + the waves' row-statistic partials [wave][16 TM][2] fp32 behind them
    constexpr int smem = (smem_ops > smem_epi ? smem_ops : smem_epi) + (LNF ? WGM * WGN * 16 * TM * 2 * (int)sizeof(float) : 0);
    static_assert(smem <= 160 * 1024, "tile does not fit LDS");
    GemmArgs2 a = a0;
    a.g.tiles_m = (int)((a.g.M + BM - 1) / BM);
    a.g.tiles_n = (a.g.N + BN - 1) / BN;
    if (a.g.n_major) a.g.n_major = ws_model_prefers(a.g.M, a.g.N, a.g.K, a.g.cin, BM, BN) ? 1 : 0;  // (allowed -> taken)
    const int nk = (a.g.K + 63) / 64;
    a.g.kt_per_split = (nk + a.g.nsplit - 1) / a.g.nsplit;
    static bool attr_done = false;  // idempotent one-time attribute of this instantiation (not tuning state)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<MODE, TM, TN, WGM, WGN, SCHED, LNF, CARRY>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) {
            mv_set_error("mv_gemm_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return MV_ERR_LAUNCH;
        }
        attr_done = true;
    }
    dim3 grid((unsigned)(a.g.tiles_m * a.g.tiles_n), (unsigned)a.g.nsplit);
    hipLaunchKernelGGL((gemm2_kernel<MODE, TM, TN, WGM, WGN, SCHED, LNF, CARRY>), grid, dim3(64 * WGM * WGN), smem, stream, a);
    MV_CHECK_LAUNCH("mv_gemm_f16");
    if (a.g.nsplit > 1) {
        const long work = a.g.M * (long)(a.g.N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, a.g);
        MV_CHECK_LAUNCH("mv_gemm_f16(split-K reduce)");
    }
    return MV_OK;
}

// ---- tile-configuration catalogue --------------------------------------------------------------------------------------
// Every (wave tile TM x TN in 16-row / 16-column MFMA tiles, waves WGM x WGN, schedule) the kernel is instantiated for, under
// a stable id: block tile = 16 TM WGM rows x 16 TN WGN columns.  The ids are what mv_gemm_desc.cfg and the per-shape table
// of gemm_tuned.h (written by tools/gpu_gemm_tune.py from timings on the MI355X) refer to.  SCHED 0: two LDS stages behind
// __syncthreads; 3: three stages behind counted waits.  (Round 1 also measured BK-32 rings, a persistent tile loop and a
// register-staged kernel, round 2 a two-group ping-pong K loop for the 8-wave tiles: all slower or equal, numbers in
// profiles/r01*, r02c; removed from the library.)
#define MV_GEMM_CFGS(X)                                                                                              \
    X(0, 4, 5, 2, 2, 0)  /* 128x160, 4 waves            */ X(1, 2, 5, 2, 2, 0)   /* 64x160                     */      \
    X(2, 4, 4, 2, 2, 0)  /* 128x128                     */ X(3, 2, 4, 2, 2, 0)   /* 64x128                     */      \
    X(4, 4, 5, 4, 2, 3)  /* 256x160, 8 waves, 3 stages  */ X(5, 4, 4, 4, 2, 3)   /* 256x128, 8 waves, 3 stages */      \
    X(6, 8, 5, 2, 4, 0)  /* 256x320, 8 waves            */ X(7, 8, 4, 2, 4, 0)   /* 256x256, 8 waves           */      \
    X(8, 4, 5, 4, 2, 0)  /* 256x160, 8 waves, 2 stages  */ X(9, 4, 4, 4, 2, 0)   /* 256x128, 8 waves, 2 stages */      \
    X(10, 4, 4, 2, 4, 3) /* 128x256, 8 waves, 3 stages  */ X(11, 4, 5, 2, 4, 0)  /* 128x320, 8 waves           */      \
    X(12, 2, 5, 4, 2, 3) /* 128x160, 8 waves, 3 stages  */ X(13, 2, 5, 2, 1, 0)  /* 64x80, 2 waves (small grids) */    \
    X(14, 2, 5, 1, 2, 0) /* 32x160, 2 waves             */ X(15, 2, 5, 4, 2, 0)  /* 128x160, 8 waves of 32x80  */      \
    X(16, 2, 4, 4, 2, 0) /* 128x128, 8 waves of 32x64   */ X(17, 2, 5, 2, 2, 3)  /* 64x160, 4 waves, 3 stages  */      \
    X(18, 4, 5, 2, 2, 3) /* 128x160, 4 waves, 3 stages  */
constexpr int kNumGemmCfgs = 19;
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

// LayerNorm-folded launches (LINEAR mode): every catalogue entry whose tile leaves registers for the row statistics (the
// 160-accumulator 256x320 tile and the 3-stage 256-row tiles do not: their choice is mapped to the nearest tile that does)
inline int gemm_ln_cfg(int id) {
    switch (id) {
        case 6: return 7;    // 256x320 -> 256x256
        case 4: case 8: return 9;   // 256x160 -> 256x128 (two stages)
        case 5: return 9;
        default: return id;
    }
}

int launch_ln_by_id(const GemmArgs2& a, hipStream_t stream, int id) {
    switch (id) {
#define MV_X(cid, tm, tn, wgm, wgn, sched) \
    case cid: if constexpr (!(tm == 8 && tn == 5) && !(cid == 4 || cid == 5 || cid == 8)) return launch_cfg2s<MV_GEMM_LINEAR, tm, tn, wgm, wgn, sched, true>(a, stream); else break;
        MV_GEMM_CFGS(MV_X)
#undef MV_X
    }
    mv_set_error("mv_gemm_f16: tile configuration %d has no LayerNorm-folded form", id);
    return MV_ERR_INVALID;
}

// Carry launches (GemmArgs::c_lo): the two-stage tiles of up to 80 accumulators per lane (the carry's epilogue holds the residual's
// two halves and the staged value's two halves per pass) and, since round 5, the 160-accumulator 256x320 tile with ONE set of residual
// registers (epilogue_staged: NRB); any other choice is mapped to the nearest of them
inline int gemm_carry_cfg(int id) {
    switch (id) {
        case 0: case 1: case 2: case 3: case 6: case 8: case 13: case 15: return id;
        case 4: case 11: return 8;    // 256x160 three stages / 128x320 -> 256x160 (two stages)
        case 12: case 18: return 0;           // 128x160 three stages -> two stages
        case 5: case 7: case 9: case 10: case 16: return 2;   // the 128-wide family -> 128x128
        case 17: return 1;
        case 14: return 13;
        default: return 0;
    }
}

template <int MODE>
int launch_carry_by_id(const GemmArgs2& a, hipStream_t stream, int id) {
    switch (id) {
        case 0: return launch_cfg2s<MODE, 4, 5, 2, 2, 0, false, true>(a, stream);
        case 1: return launch_cfg2s<MODE, 2, 5, 2, 2, 0, false, true>(a, stream);
        case 2: return launch_cfg2s<MODE, 4, 4, 2, 2, 0, false, true>(a, stream);
        case 3: return launch_cfg2s<MODE, 2, 4, 2, 2, 0, false, true>(a, stream);
        case 6: return launch_cfg2s<MODE, 8, 5, 2, 4, 0, false, true>(a, stream);
        case 8: return launch_cfg2s<MODE, 4, 5, 4, 2, 0, false, true>(a, stream);
        case 13: return launch_cfg2s<MODE, 2, 5, 2, 1, 0, false, true>(a, stream);
        case 15: return launch_cfg2s<MODE, 2, 5, 4, 2, 0, false, true>(a, stream);
    }
    mv_set_error("mv_gemm_f16: tile configuration %d has no carry form", id);
    return MV_ERR_INVALID;
}

// per-shape choices measured on the MI355X (mode / N / K / geglu exact, nearest M within a factor of 3; anything else follows the
// rules below)
struct GemmTuned { int mode; long M; int N, K, geglu, ln, cfg, nsplit; };
struct GemmKeyed { int mode, geglu, ln, kb, fb, cfg; };
#include "gemm_tuned.h"

// the key of a problem for sizes the table does not hold: K tiles of 64 in five classes (the level-0 projections | the 640-wide |
// the 1280-wide | convolution K | long convolution K) and the fill of the CUs by the 128 x 160 grid in six (half a round | one |
// two | four | sixteen | more).  tools/tile_choice_study.py:key_of is the same arithmetic (it generates kGemmKeyed).
struct GemmKey {
    int kb, fb;
    bool operator==(const GemmKey& o) const { return kb == o.kb && fb == o.fb; }
};
inline GemmKey gemm_key(long M, int N, int K, int cus) {
    const int nk = (K + 63) / 64;
    const long grid2 = 2 * ((M + 127) / 128) * ((N + 159) / 160);
    GemmKey k;
    k.kb = (nk > 5) + (nk > 10) + (nk > 20) + (nk > 60);
    k.fb = (grid2 > 1L * cus) + (grid2 > 2L * cus) + (grid2 > 4L * cus) + (grid2 > 8L * cus) + (grid2 > 32L * cus);
    return k;
}

struct GemmChoice { int cfg, nsplit; };

// split factor for a tile grid of `blocks` workgroups with `nk` K tiles: fill ~2 workgroups per CU, keep >= 8 K tiles per slice
inline int splitk_rule(long blocks, int nk, int cus) {
    int s = (int)((2L * cus) / (blocks > 0 ? blocks : 1));
    if (s > nk / 8) s = nk / 8;
    if (s > 8) s = 8;
    return s < 1 ? 1 : s;
}

// the slabs of a split must stay small next to the operands: cap the workspace of a forced / tabled split
inline int splitk_clamp(const GemmArgs& g, int nsplit) {
    const int nk = (g.K + 63) / 64;
    if (g.geglu || nsplit < 1) return 1;
    while (nsplit > 1 && ((long)nsplit * g.M * g.N * 4 > (64L << 20) || nk / nsplit < 4)) nsplit >>= 1;
    return nsplit;
}

// tile configuration + split factor of a problem: the caller's choice (desc.cfg >= 0 / desc.splitk >= 1), else the measured
// table, else the rules
inline GemmChoice choose_config(int mode, const GemmArgs& g, int want_cfg, int want_split) {
    GemmChoice ch{-1, 0};
    if (want_cfg >= 0 && gemm_cfg_applies(want_cfg, g)) ch.cfg = want_cfg;
    const int cus = mv_num_cus();
    const int nk = (g.K + 63) / 64;
    if (ch.cfg < 0 && want_cfg == -1) {
        // measured table.  (1) The entry of this (mode, N, K, geglu) whose M is NEAREST on a log scale: within a factor of 1.26 it
        // IS the measurement (tile and split factor; cfg -2 = the rules measured best).  (2) Further away -- the table is measured on
        // the 512 x 512 and 768 x 768 benchmarks at 13 / 26 frames; 512 x 320, other window lengths and resolutions are not in it --
        // the same layer's entry still lends its tile if it sits in the SAME key bucket (same K class, same fill of the CUs: a
        // 256 x 320 tile tuned on a full grid is not handed to a grid of 104 blocks), (3) else the key's vote over every measured
        // problem (kGemmKeyed), (4) else the nearest entry within a factor of 3 anyway, (5) else the rules.  Priced on held-out
        // resolutions from the tuner's own measurements (tools/tile_choice_study.py, profiles/r04w_tile_choice_study.log): 2.2-3.1 %
        // over a fresh tune at 2.25x / 0.44x the size, against 3.8-6.1 % for (4) alone and 11-18 % for the rules.  An entry measured
        // on the LayerNorm-folded form of the launch wins over a plain one at the same distance; away from the measured M the
        // split rule decides the split.
        int best = -1;
        double best_d = 1e30;
        const bool want_ln = g.ln_colsum != nullptr;
        const bool want_carry = g.c_lo != nullptr;   // rows measured on the carry form of a launch (its own epilogue, never split): ln == 2
        for (int i = 0; i < kNumGemmTuned; ++i) {
            const GemmTuned& e = kGemmTuned[i];
            if (e.mode != mode || e.N != g.N || e.K != g.K || e.geglu != g.geglu || e.M <= 0 || !(e.cfg == -2 || gemm_cfg_applies(e.cfg, g))) continue;
            if ((e.ln == 1 && !want_ln) || (e.ln == 2 && !want_carry)) continue;
            const double r = (double)g.M / (double)e.M;
            double dist = r > 1.0 ? r : 1.0 / r;  // >= 1: the size ratio
            if (dist > 3.0) continue;
            if ((e.ln == 1) != want_ln || (e.ln == 2) != want_carry) dist *= 1.0001;  // tie-break only
            if (dist < best_d) {
                best_d = dist;
                best = i;
            }
        }
        const GemmKey key = gemm_key(g.M, g.N, g.K, cus);
        bool settled = false;
        if (best >= 0 && best_d <= 1.26) {
            settled = true;  // (cfg -2: the rules below measured best for this problem)
            if (kGemmTuned[best].cfg >= 0) {
                ch.cfg = kGemmTuned[best].cfg;
                ch.nsplit = kGemmTuned[best].nsplit;
            }
        } else if (best >= 0 && kGemmTuned[best].cfg >= 0 && (kGemmTuned[best].ln == 1) == want_ln && gemm_key(kGemmTuned[best].M, g.N, g.K, cus) == key) {
            settled = true;
            ch.cfg = kGemmTuned[best].cfg;
        }
        if (!settled) {
            for (int i = 0; i < kNumGemmKeyed; ++i) {
                const GemmKeyed& e = kGemmKeyed[i];
                if (e.mode != mode || e.geglu != g.geglu || (e.ln != 0) != want_ln || e.kb != key.kb || e.fb != key.fb || !gemm_cfg_applies(e.cfg, g)) continue;
                ch.cfg = e.cfg;
                settled = true;
                break;
            }
        }
        if (!settled && best >= 0 && kGemmTuned[best].cfg >= 0) ch.cfg = kGemmTuned[best].cfg;
    }
    if (ch.cfg < 0) {
        // rules: BN = 160 when it divides N (all UNet widths are multiples of 320), else 128; BM = 128 unless that leaves the
        // CUs under-filled, then 64; GEGLU (even TN) on 128x128.  One-round grids of 8-wave 256x160 tiles on the counted-wait
        // ring where they fit (measured +20 % on the M = 6656 level, profiles/r01e).
        if (g.geglu) {
            ch.cfg = 2;
        } else {
            const bool n160 = (g.N % 160) == 0;
            const long tiles_n = n160 ? g.N / 160 : (g.N + 127) / 128;
            const bool small = ((g.M + 127) / 128) * tiles_n < 512;
            const long blocks8 = (g.M + 255) / 256 * tiles_n;
            if (n160 && blocks8 * 5 >= cus * 4L && blocks8 <= cus) ch.cfg = 4;
            else if (n160) ch.cfg = small ? 1 : 0;
            else ch.cfg = small ? 3 : 2;
            // small grids with long K loops (the 8x8-latent level): latency-bound -> three-stage ring + split-K
            const GemmCfgDesc& c = kGemmCfgs[ch.cfg];
            const long blocks = ((g.M + 16 * c.tm * c.wgm - 1) / (16 * c.tm * c.wgm)) * ((g.N + 16 * c.tn * c.wgn - 1) / (16 * c.tn * c.wgn));
            if (ch.cfg == 1 && blocks <= cus && nk >= 16) ch.cfg = 17;
        }
    }
    if (want_split >= 1 && !g.ln_colsum) ch.nsplit = want_split;
    if (ch.nsplit < 1) {
        const GemmCfgDesc& c = kGemmCfgs[ch.cfg];
        const long blocks = ((g.M + 16 * c.tm * c.wgm - 1) / (16 * c.tm * c.wgm)) * ((g.N + 16 * c.tn * c.wgn - 1) / (16 * c.tn * c.wgn));
        ch.nsplit = (blocks <= cus && nk >= 16) ? splitk_rule(blocks, nk, cus) : 1;
    }
    ch.nsplit = splitk_clamp(g, ch.nsplit);
    const int per = (nk + ch.nsplit - 1) / ch.nsplit;
    ch.nsplit = (nk + per - 1) / per;  // no empty slice
    if (g.ln_colsum) {  // the row statistics span the whole K: one slice, a tile with room for them
        ch.cfg = gemm_ln_cfg(ch.cfg);
        ch.nsplit = 1;
    }
    if (g.c_lo) {  // the carry lives in the staged epilogue of an unsplit launch
        ch.cfg = gemm_carry_cfg(ch.cfg);
        ch.nsplit = 1;
    }
    if (g.w_group_rows > 0 && g.w_group_rows % (16 * kGemmCfgs[ch.cfg].tm * kGemmCfgs[ch.cfg].wgm) != 0) {
        // per-group weights: a block tile must not straddle two groups -- the largest 160-wide (GEGLU: 128-wide) tile whose rows divide the group
        const int cand160[] = {8, 0, 1, 14}, cand128[] = {9, 2, 3, 3};   // 256 / 128 / 64 / 32 rows (64 for the 128-wide family)
        for (int i = 0; i < 4; ++i) {
            const int c = g.geglu ? cand128[i] : cand160[i];
            if (g.w_group_rows % (16 * kGemmCfgs[c].tm * kGemmCfgs[c].wgm) == 0) {
                ch.cfg = c;
                break;
            }
        }
    }
    return ch;
}

}  // namespace

extern "C" int mv_gemm_num_configs(void) { return kNumGemmCfgs; }

extern "C" int mv_gemm_config_desc(int cfg, int32_t* desc5) {
    MV_REQUIRE(cfg >= 0 && cfg < kNumGemmCfgs && desc5, "mv_gemm_config_desc: bad args");
    const GemmCfgDesc& c = kGemmCfgs[cfg];
    desc5[0] = 16 * c.tm * c.wgm;  // block rows
    desc5[1] = 16 * c.tn * c.wgn;  // block columns
    desc5[2] = c.wgm * c.wgn;      // waves
    desc5[3] = 64;                 // BK
    desc5[4] = c.sched == 3 ? 3 : 2;   // LDS stages
    return MV_OK;
}

// host-side evaluation of the workgroup -> tile map the kernel uses (the same inline functions): introspection for
// tests and for reasoning about L2 locality; launches nothing
extern "C" int mv_gemm_tile_order(int tiles_m, int tiles_n, int group, int32_t* tile_m, int32_t* tile_n) {
    MV_REQUIRE(tiles_m > 0 && tiles_n > 0 && (long)tiles_m * tiles_n < (1L << 31) && tile_m && tile_n && group >= -1,
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

namespace {

// validation + translation of the public descriptor; returns MV_OK and fills `b`, the mode's tap count is checked here
int gemm_prepare(const mv_gemm_desc* d, GemmArgs2& b, const char* who) {
    MV_REQUIRE(d != nullptr, "%s: null descriptor", who);
    MV_REQUIRE(d->a && d->w && d->c, "%s: null a/w/c pointer", who);
    MV_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "%s: empty problem M=%ld N=%d K=%d", who, (long)d->M, d->N, d->K);
    MV_REQUIRE(d->N % 4 == 0 && d->K % 8 == 0, "%s: need N %% 4 == 0 and K %% 8 == 0 (N=%d K=%d)", who, d->N, d->K);
    MV_REQUIRE(d->ldc % 4 == 0 && d->lda % 8 == 0, "%s: lda must be a multiple of 8 and ldc of 4", who);
    const int c2 = d->a2 ? d->c2 : 0;
    const int cin = d->c1 + c2;
    MV_REQUIRE(d->c1 > 0 && d->c1 % 8 == 0 && c2 % 8 == 0, "%s: c1/c2 must be multiples of 8", who);
    if (d->a2) {
        MV_REQUIRE(d->c1 % 64 == 0 && d->lda2 % 8 == 0, "%s: two-source input needs c1 %% 64 == 0", who);
    }
    int taps = 1;
    if (d->mode == MV_GEMM_CONV3X3) taps = 9;
    else if (d->mode == MV_GEMM_TCONV3) taps = 3;
    else MV_REQUIRE(d->mode == MV_GEMM_LINEAR, "%s: bad mode %d", who, d->mode);
    MV_REQUIRE(d->K == taps * cin, "%s: K=%d != taps*cin=%d*%d", who, d->K, taps, cin);
    if (taps > 1) MV_REQUIRE(cin % 64 == 0, "%s: conv modes need cin %% 64 == 0 (cin=%d)", who, cin);
    if (d->residual) MV_REQUIRE(d->ldr % 4 == 0, "%s: ldr %% 4", who);
    if (d->rowbias) MV_REQUIRE(d->ldrb % 4 == 0 && d->rows_per_group > 0, "%s: rowbias needs ldrb %% 4 and rows_per_group > 0", who);
    if (d->geglu) {
        MV_REQUIRE(d->N % 32 == 0 && !d->rowbias && !d->residual && !d->alpha && d->act == MV_ACT_NONE,
                   "%s: geglu epilogue needs N %% 32 == 0 and no other epilogue terms", who);
    }
    MV_REQUIRE(d->cfg >= -2 && d->cfg < kNumGemmCfgs, "%s: tile configuration %d not in [-2, %d)", who, d->cfg, kNumGemmCfgs);
    MV_REQUIRE(d->splitk >= 0 && d->splitk <= 16, "%s: splitk %d not in [0, 16]", who, d->splitk);
    GemmArgs& a = b.g;
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
    // weight-stationary tile order: mv_gemm_desc.tile_order = 1 ALLOWS it; the launcher takes it where the traffic model says the
    // XCDs fetch less that way (ws_model_prefers, once the tile shape is known)
    a.n_major = d->tile_order == 1 ? 1 : 0;
    a.nsplit = 1; a.kt_per_split = 0; a.ws = nullptr;
    a.ln_colsum = d->ln_colsum; a.ln_colbias = d->ln_colbias; a.ln_eps = d->ln_eps;
    a.colstats = nullptr;  // (set by mv_gemm_f16 once the choice is known to support it)
    a.residual_lo = (const half_t*)d->residual_lo; a.c_lo = (half_t*)d->c_lo;
    MV_REQUIRE(!d->residual_lo || (d->residual && d->c_lo), "%s: residual_lo needs residual and c_lo", who);
    a.w_group_rows = d->w_group_rows; a.rowbias_lo = (const half_t*)d->rowbias_lo;
    MV_REQUIRE(d->w_group_rows >= 0, "%s: w_group_rows < 0", who);
    if (d->w_group_rows > 0) {
        MV_REQUIRE(d->mode == MV_GEMM_LINEAR && !d->ln_colsum && !d->c_lo, "%s: per-group weights (w_group_rows): LINEAR mode, no LayerNorm folding, no carry", who);
        MV_REQUIRE(d->M % d->w_group_rows == 0 && d->w_group_rows % 32 == 0, "%s: M = %ld is not a whole number of w_group_rows = %d (a multiple of 32) row groups",
                   who, (long)d->M, d->w_group_rows);
        MV_REQUIRE((d->M / d->w_group_rows) * (long)d->N * d->K * 2 < 0x7fffffffL, "%s: the per-group weight matrices span 2 GiB or more", who);
    }
    MV_REQUIRE(!d->rowbias_lo || d->rowbias, "%s: rowbias_lo needs rowbias", who);
    if (d->c_lo)
        MV_REQUIRE(!d->geglu && !d->ln_colsum && d->splitk <= 1, "%s: the carry (c_lo) excludes GEGLU, LayerNorm folding and a forced K split", who);
    if (d->ln_colsum || d->ln_colbias) {
        MV_REQUIRE(d->ln_colsum && d->ln_colbias && d->ln_eps > 0.f, "%s: LayerNorm folding needs ln_colsum, ln_colbias and ln_eps > 0", who);
        MV_REQUIRE(d->mode == MV_GEMM_LINEAR && !d->a2 && !d->bias && !d->rowbias && d->K % 64 == 0,
                   "%s: LayerNorm folding: LINEAR mode, one source, K %% 64 == 0, bias folded into ln_colbias (no bias / rowbias)", who);
        MV_REQUIRE((reinterpret_cast<uintptr_t>(d->ln_colsum) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->ln_colbias) & 15) == 0,
                   "%s: ln_colsum / ln_colbias must be 16-byte aligned", who);
    }
    if (d->mode == MV_GEMM_CONV3X3) {
        MV_REQUIRE(d->stride == 1 || d->stride == 2, "%s: conv stride must be 1 or 2", who);
        MV_REQUIRE(!(d->upsample && d->stride != 1), "%s: upsample requires stride 1", who);
        MV_REQUIRE(d->hin > 0 && d->win > 0 && d->hout > 0 && d->wout > 0, "%s: conv geometry missing", who);
        MV_REQUIRE(d->M % ((long)d->hout * d->wout) == 0, "%s: M is not a whole number of output images", who);
    }
    if (d->mode == MV_GEMM_TCONV3)
        MV_REQUIRE(d->t > 0 && d->hw > 0 && d->M % ((long)d->t * d->hw) == 0, "%s: tconv geometry: M must be B*T*HW", who);
    // every source must span < 2 GiB (32-bit byte offsets through buffer descriptors, 0x80000000 = "reads zero" marker)
    const long rows_in = d->mode == MV_GEMM_CONV3X3 ? (d->M / ((long)d->hout * d->wout)) * d->hin * d->win : d->M;
    const long a_bytes = ((rows_in - 1) * (long)d->lda + d->c1) * 2;
    const long a2_bytes = d->a2 ? ((rows_in - 1) * (long)d->lda2 + c2) * 2 : 0;
    const long w_bytes = (long)d->N * d->K * 2;
    const long lim = 0x7fffffffL;
    MV_REQUIRE(a_bytes < lim && a2_bytes < lim && w_bytes < lim && d->M < lim,
               "%s: an operand spans 2 GiB or more (a %ld, a2 %ld, w %ld bytes): split the call", who, a_bytes, a2_bytes, w_bytes);
    b.a_bytes = (unsigned)a_bytes; b.a2_bytes = (unsigned)a2_bytes; b.w_bytes = (unsigned)w_bytes;
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    b.wide = (d->N % 8 == 0) && (d->ldc % 8 == 0) && al16(d->c) && (!d->bias || al16(d->bias)) &&
             (!d->rowbias || (d->ldrb % 8 == 0 && al16(d->rowbias) && (!d->rowbias_lo || al16(d->rowbias_lo)))) &&
             (!d->residual || (d->ldr % 8 == 0 && al16(d->residual)));
    if (d->ln_colsum) MV_REQUIRE(b.wide, "%s: LayerNorm folding needs the 16-byte epilogue (N, ldc, ldr %% 8, aligned pointers)", who);
    if (d->c_lo)
        MV_REQUIRE(b.wide && al16(d->c_lo) && (!d->residual_lo || al16(d->residual_lo)),
                   "%s: the carry needs the 16-byte epilogue (N, ldc, ldr %% 8; c_lo / residual_lo 16-byte aligned, same leading dimensions as c / residual)", who);
    return MV_OK;
}

}  // namespace

extern "C" int64_t mv_gemm_workspace_bytes(const mv_gemm_desc* d) {
    GemmArgs2 b;
    if (gemm_prepare(d, b, "mv_gemm_workspace_bytes") != MV_OK) return -1;
    const GemmChoice ch = choose_config(d->mode, b.g, d->cfg, d->splitk);
    return ch.nsplit > 1 ? (int64_t)ch.nsplit * b.g.M * b.g.N * 4 : 0;
}

extern "C" int mv_gemm_choice(const mv_gemm_desc* d, int32_t* cfg, int32_t* nsplit) {
    GemmArgs2 b;
    if (int rc = gemm_prepare(d, b, "mv_gemm_choice")) return rc;
    MV_REQUIRE(cfg && nsplit, "mv_gemm_choice: null output");
    const GemmChoice ch = choose_config(d->mode, b.g, d->cfg, d->splitk);
    *cfg = ch.cfg;
    *nsplit = ch.nsplit;
    return MV_OK;
}

extern "C" int mv_gemm_weight_stationary(const mv_gemm_desc* d) {
    GemmArgs2 b;
    if (gemm_prepare(d, b, "mv_gemm_weight_stationary") != MV_OK) return -1;
    if (!b.g.n_major) return 0;
    GemmChoice ch = choose_config(d->mode, b.g, d->cfg, d->splitk);
    const GemmCfgDesc& c = kGemmCfgs[ch.cfg];
    return ws_model_prefers(b.g.M, b.g.N, b.g.K, b.g.cin, 16 * c.tm * c.wgm, 16 * c.tn * c.wgn) ? 1 : 0;
}

// output statistics a launch can emit: the staged (16-byte) epilogue of an unsplit, non-GEGLU launch
inline bool stats_capable(const GemmArgs2& b, const GemmChoice& ch) { return b.wide && ch.nsplit == 1 && !b.g.geglu; }
inline int colstats_rows_per_tile(const GemmArgs2& b, const GemmChoice& ch) { return stats_capable(b, ch) ? 16 * kGemmCfgs[ch.cfg].tm : 0; }

extern "C" int mv_gemm_stats_layout(const mv_gemm_desc* d, int32_t* col_rows_per_tile, int64_t* col_floats) {
    GemmArgs2 b;
    if (int rc = gemm_prepare(d, b, "mv_gemm_stats_layout")) return rc;
    MV_REQUIRE(col_rows_per_tile && col_floats, "mv_gemm_stats_layout: null output");
    const GemmChoice ch = choose_config(d->mode, b.g, d->cfg, d->splitk);
    const int rpt = colstats_rows_per_tile(b, ch);
    *col_rows_per_tile = rpt;
    *col_floats = rpt ? ((b.g.M + rpt - 1) / rpt) * (int64_t)b.g.N * 2 : 0;
    return MV_OK;
}

extern "C" int mv_gemm_f16(const mv_gemm_desc* d, void* stream) {
    GemmArgs2 b;
    if (int rc = gemm_prepare(d, b, "mv_gemm_f16")) return rc;
    const GemmChoice ch = choose_config(d->mode, b.g, d->cfg, d->splitk);
    if (d->colstats) {
        const int rpt = colstats_rows_per_tile(b, ch);
        MV_REQUIRE(rpt > 0, "mv_gemm_f16: this launch cannot emit column statistics (split K, GEGLU or narrow epilogue): ask mv_gemm_stats_layout first");
        MV_REQUIRE(d->colstats_floats >= ((b.g.M + rpt - 1) / rpt) * (int64_t)b.g.N * 2 && (reinterpret_cast<uintptr_t>(d->colstats) & 7) == 0,
                   "mv_gemm_f16: colstats needs %ld floats (8-byte aligned), got %ld", (long)(((b.g.M + rpt - 1) / rpt) * (int64_t)b.g.N * 2), (long)d->colstats_floats);
        b.g.colstats = d->colstats;
    }
    if (ch.nsplit > 1) {
        const int64_t need = (int64_t)ch.nsplit * b.g.M * b.g.N * 4;
        MV_REQUIRE(d->workspace && d->workspace_bytes >= need && (reinterpret_cast<uintptr_t>(d->workspace) & 15) == 0,
                   "mv_gemm_f16: split-K x%d needs a 16-byte aligned workspace of %ld bytes (mv_gemm_workspace_bytes); got %ld",
                   ch.nsplit, (long)need, (long)d->workspace_bytes);
        b.g.nsplit = ch.nsplit;
        b.g.ws = (float*)d->workspace;
    }
    hipStream_t s = (hipStream_t)stream;
    if (d->ln_colsum) {
        MV_REQUIRE(ch.nsplit == 1, "mv_gemm_f16: LayerNorm folding cannot be combined with a forced K split");
        return launch_ln_by_id(b, s, ch.cfg);
    }
    if (d->c_lo) {
        MV_REQUIRE(ch.nsplit == 1, "mv_gemm_f16: the carry cannot be combined with a K split");
        if (d->mode == MV_GEMM_CONV3X3) return launch_carry_by_id<MV_GEMM_CONV3X3>(b, s, ch.cfg);
        if (d->mode == MV_GEMM_TCONV3) return launch_carry_by_id<MV_GEMM_TCONV3>(b, s, ch.cfg);
        return launch_carry_by_id<MV_GEMM_LINEAR>(b, s, ch.cfg);
    }
    if (d->mode == MV_GEMM_CONV3X3) return launch_by_id<MV_GEMM_CONV3X3>(b, s, ch.cfg);
    if (d->mode == MV_GEMM_TCONV3) return launch_by_id<MV_GEMM_TCONV3>(b, s, ch.cfg);
    return launch_by_id<MV_GEMM_LINEAR>(b, s, ch.cfg);
}
