// ffn.hip -- the feed-forward of a BasicTransformerBlock as ONE kernel for the level-0 stream (C = 320, hidden 4 C = 1280):
//
//     out = residual + FF2( GEGLU( FF1( LayerNorm(x) ) ) )           (musev/models/attention.py:398-429; diffusers FeedForward / GEGLU)
//
// The three launches it replaces (mv_layernorm_f16, the GEGLU projection, the output projection + residual) round-trip the
// [M, 4 C] activation and the normalised rows through HBM: at M = 53 248 that is 340 MB of the chain's 442 MB, and the GEGLU launch
// is bound by writing its output (635 TFLOP/s against 800+ for every other level-0 GEMM, profiles/r04b_gemm_by_problem.json).
// Here a block owns 128 rows for the whole chain and the hidden activation never leaves the CU:
//   * prologue: the block's 128 rows of x are normalised in registers (two-pass LayerNorm, 4 lanes per row, fp32), rounded to fp16 --
//     the values mv_layernorm_f16 would have written to HBM -- and turned into MFMA-operand FRAGMENTS through LDS (5 K tiles of
//     [128 rows][64], 128-byte rows, XOR-(row & 7) slot swizzle: every fragment is one conflict-free ds_read_b128): a wave keeps the
//     2 x 10 fragments of its 32 rows in registers for the whole kernel, and the LDS they passed through becomes part of the ring;
//   * the hidden dimension is walked in 20 chunks of 64: per chunk  S = xn . W1[chunk]^T + b1  (128 packed columns = 64 values +
//     64 gates, K = 320), g = value * gelu(gate) as fp16 into an LDS tile in the same operand layout, then  acc += g . W2[:, chunk]^T
//     (K = 64, all 320 output columns) -- the accumulators of the OUTPUT (32 rows x 160 columns per wave; 8 waves = 4 row groups x 2
//     column groups) stay in registers across the chunks;
//   * both weight matrices stream through ONE ring of 16-KiB LDS tiles ([128 weight rows][64 k], same layout, filled by
//     buffer_load ... lds through two buffer descriptors): 5 W1 tiles + 3 W2 tiles per chunk, 160 tiles per block, every tile = 16
//     MFMAs per wave (8 for the half-empty third W2 tile); counted vmcnt waits + one raw s_barrier per tile keep RING - 1 = 7 tiles
//     (112 KB) in flight -- one 8-wave block per CU cannot hide a late tile behind another block, only behind its own prefetch depth
//     (the first form of this kernel kept x in LDS and had room for 3 stages: 212 us at M = 53 248, profiles/r04d_ffn_bench.log);
//     nothing else issues vector-memory loads inside the loop (b1 sits in LDS: vmcnt is in-order, a stray load would drain the ring
//     when its value is waited for);
//   * the epilogue stages the fp32 result through the idle LDS so that the residual loads and the stores are 16 bytes per lane on
//     consecutive bytes of a row; b2 is the initial value of the accumulators; the residual add is fp32, rounded once.
// Weights are read from L2 / the infinity cache (2.4 MB per block, the same bytes for every block); HBM sees x, the residual and
// the output: 3 x M x 320 x 2 bytes.
#include "common.h"
#include <type_traits>

namespace {

struct FfnArgs {
    const half_t* x;
    const half_t* gamma;     // [C] LayerNorm weight
    const half_t* beta;      // [C] LayerNorm bias
    const half_t* w1;        // [2 H][C] GEGLU-packed rows ([16 value | 16 gate] blocks)
    const half_t* bias1;     // [2 H] packed like the rows of w1, or nullptr
    const half_t* w2;        // [C][H]
    const half_t* bias2;     // [C] or nullptr
    const half_t* residual;  // [M][ldr]
    half_t* out;             // [M][ldo]
    long M;
    int ldx, ldr, ldo;
    float eps;
    unsigned w1_bytes, w2_bytes;
    int rotate;              // blocks start their walk over the hidden chunks at different chunks (see ffn_geglu_kernel)
    int ablate;              // experiment builds only (MV_EXPERIMENT, tools/gpu_ffn_bench.py --ablate): bit 0 no gelu, 1 no phase-1 MFMAs,
                             // 2 no phase-2 MFMAs, 3 no g tile write  (wrong results: timing only)
};
#ifdef MV_EXPERIMENT
#define MV_FFN_ABL(bit) (p.ablate & (1 << (bit)))
#else
#define MV_FFN_ABL(bit) 0
#endif

constexpr int kC = 320, kH = 1280;
constexpr int kBM = 128;                 // rows per block
constexpr int kHC = 64;                  // hidden units per chunk (= 128 packed FF1 columns)
constexpr int kChunks = kH / kHC;        // 20
constexpr int kKT1 = kC / 64;            // 5 K tiles of the first projection
constexpr int kTilesPerChunk = kKT1 + 3; // + 3 tiles of W2 (output columns 0-127, 128-255, 256-319)
constexpr int kTileHalfs = 128 * 64;     // one operand tile: [128 rows][64 k] halfs = 16 KiB
constexpr int kRing = 8;                 // LDS stages of the weight stream; stages 0 .. 4 first carry the normalised rows to the registers
constexpr int kGHalfs = kBM * kHC;       // the chunk's gated activation: one tile
constexpr int kLdsHalfs = kRing * kTileHalfs + kGHalfs + 2 * kH;
constexpr unsigned kOob = 0x80000000u;
constexpr int kSLd = kC + 4;             // epilogue: floats per staging row
static_assert(64 * kSLd * 4 <= kRing * kTileHalfs * 2, "a 64-row fp32 staging tile must fit the ring");
static_assert(kRing - 1 - kKT1 >= 1 && kRing - 1 <= kTilesPerChunk, "prologue: tiles 0 .. RING - 2 of chunk 0, the first of them beside the x tiles");

__global__ __launch_bounds__(512, 2) void ffn_geglu_kernel(const FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* const ring = lds;                           // [kRing][128][64]
    half_t* const xs = lds;                             // [5][128][64]: stages 0 .. 4, until the rows are in registers
    half_t* const gbuf = ring + kRing * kTileHalfs;     // [128][64]
    half_t* const b1s = gbuf + kGHalfs;                 // [2 H]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = (int)((p.M + kBM - 1) / kBM);
    const int bid = mv_xcd_remap(blockIdx.x, nblk);
    const int m0 = bid * kBM;
    const int Mi = (int)p.M;
    // rotate: block b walks the chunks from chunk (7 b) % 20 on, so that the ~256 blocks in flight do not all ask the L2 for the same
    // 16-KiB weight tile at the same moment (the sum over chunks is in a different -- fixed per row block -- order: bit-reproducible)
    const int cbase = p.rotate ? (bid * 7) % kChunks : 0;

    const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, p.w2_bytes, 0x00020000);

    // ---- the block's rows of x: thread -> (row tid / 4, 16-byte chunks (tid % 4) + 4 q, q = 0 .. 9); requested AHEAD of the ring's
    // first tiles (in-order vmcnt: waiting for them does not wait for a tile) ----
    const int xrow = tid >> 2, xq = tid & 3;
    half8v xv[10];
    {
        const int row = m0 + xrow;
        const half_t* xr = p.x + (long)(row < Mi ? row : Mi - 1) * p.ldx + 8 * xq;
#pragma unroll
        for (int q = 0; q < 10; ++q) xv[q] = *reinterpret_cast<const half8v*>(xr + 32 * q);
    }

    // ---- LDS-DMA geometry of the weight stream: a tile is 16 pieces of 8 rows x 128 B; wave w issues pieces w and w + 8.  Lane ->
    // (row lane / 8, 16-byte slot lane % 8); the swizzle (slot ^ row) lives on the SOURCE address, a piece's LDS image is lane-linear ----
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off1[2], off2[2];  // per piece: byte offset of this lane's chunk relative to (tile row 0, k 0) in W1 / W2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * (wave + 8 * q) + lrow;  // tile row 0 .. 127
        off1[q] = ((unsigned)r * (unsigned)kC + lsl * 8u) * 2u;
        off2[q] = ((unsigned)r * (unsigned)kH + lsl * 8u) * 2u;
    }
    // the weight stream: tile (chunk, slot), slots 0 .. 4 = W1 k tiles, 5 .. 7 = W2 column tiles, into ring stage `stage`.  The slot is
    // a compile-time constant (the tile loop below is unrolled over a chunk's 8 steps): no branch, no division in the stream
    auto issue = [&](auto slot_c, int chunk_i, int stage) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        const int chunk = chunk_i + cbase < kChunks ? chunk_i + cbase : chunk_i + cbase - kChunks;
        half_t* dst = ring + stage * kTileHalfs;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_t* d = dst + (wave + 8 * q) * (8 * 64);
            if constexpr (slot < kKT1) {
                // W1 rows 128 chunk .. + 127, columns 64 slot .. + 63
                const unsigned so = ((unsigned)(128 * chunk) * (unsigned)kC + 64u * (unsigned)slot) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, (__attribute__((address_space(3))) void*)d, 16, (int)off1[q], (int)so, 0, 0);
            } else {
                // W2 rows (output columns) 128 t .. + 127 (rows >= 320 read zero), columns 64 chunk .. + 63
                constexpr int t = slot - kKT1;
                const int r = 128 * t + 8 * (wave + 8 * q) + lrow;
                const unsigned vo = r < kC ? off2[q] : kOob;
                const unsigned so = ((unsigned)(128 * t) * (unsigned)kH + 64u * (unsigned)chunk) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (__attribute__((address_space(3))) void*)d, 16, (int)vo, (int)so, 0, 0);
            }
        }
    };
    // tile seq lives in stage (seq + 5) % RING: the first RING - 1 - 5 tiles go to the stages beside the x tiles at once, the rest of
    // the prologue's tiles follow when the rows have left stages 0 .. 4
    issue(std::integral_constant<int, 0>{}, 0, 5);
    issue(std::integral_constant<int, 1>{}, 0, 6);

    // ---- LayerNorm of the rows (two-pass in registers, the 4 lanes of a row meet by xor-shuffles) -> xs, operand layout ----
    {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 10; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)xv[q][e];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        const float mu = sum * (1.0f / (float)kC);
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < 10; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dlt = (float)xv[q][e] - mu;
                sq = fmaf(dlt, dlt, sq);
            }
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        const float rs = rsqrtf(sq * (1.0f / (float)kC) + p.eps);
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            const int ch = xq + 4 * q;  // 16-byte chunk of the row: channels 8 ch .. + 7 -> K tile ch / 8, slot ch % 8
            const half8v gm = *reinterpret_cast<const half8v*>(p.gamma + 8 * ch);
            const half8v bt = *reinterpret_cast<const half8v*>(p.beta + 8 * ch);
            half8v o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)xv[q][e] - mu) * rs * (float)gm[e] + (float)bt[e]);
            *reinterpret_cast<half8v*>(xs + (ch >> 3) * kTileHalfs + xrow * 64 + ((((ch & 7) ^ (xrow & 7))) << 3)) = o;
        }
    }
    // the first projection's bias -> LDS (packed order, 5 KB); zeros when there is none
    for (int i = tid; i < 2 * kH / 8; i += 512)
        *reinterpret_cast<half8v*>(b1s + 8 * i) = p.bias1 ? *reinterpret_cast<const half8v*>(p.bias1 + 8 * i) : half8v{0, 0, 0, 0, 0, 0, 0, 0};
    // ---- output accumulators, starting from b2 (requested ahead of the barrier below: no stray load in the ring's queue): acc2[i][jj] = rows 32 wm + 16 i + l15, columns ocol(jj) + 4 g .. + 3 ----
    // column tiles jj 0-3: output columns 64 wn + 16 jj (W2 tile 0), 4-7: 128 + 64 wn + 16 (jj - 4) (tile 1), 8-9: 256 + 32 wn + 16 (jj - 8)
    auto ocol = [&](int jj) { return jj < 4 ? 64 * wn + 16 * jj : jj < 8 ? 128 + 64 * wn + 16 * (jj - 4) : 256 + 32 * wn + 16 * (jj - 8); };
    float4v acc2[2][10];
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) {
        float4v b0 = float4v{0.f, 0.f, 0.f, 0.f};
        if (p.bias2) {
            const half4v b = *reinterpret_cast<const half4v*>(p.bias2 + ocol(jj) + 4 * g);
            b0 = float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
        }
        acc2[0][jj] = b0;
        acc2[1][jj] = b0;
    }

    // ---- the rows as B-operand fragments in registers: xreg[i][2 kt + kk] = xn[row 32 wm + 16 i + l15][64 kt + 32 kk + 8 g .. + 7] ----
    half8v xreg[2][10];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int sk = 0; sk < 10; ++sk)
            xreg[i][sk] = *reinterpret_cast<const half8v*>(xs + (sk >> 1) * kTileHalfs + (32 * wm + 16 * i + l15) * 64 +
                                                           (((((sk & 1) * 4 + g) ^ (l15 & 7))) << 3));
    __syncthreads();  // every wave holds its rows: stages 0 .. 4 join the ring
    issue(std::integral_constant<int, 2>{}, 0, 7);
    issue(std::integral_constant<int, 3>{}, 0, 0);
    issue(std::integral_constant<int, 4>{}, 0, 1);
    issue(std::integral_constant<int, 5>{}, 0, 2);
    issue(std::integral_constant<int, 6>{}, 0, 3);

    const int swz = l15 & 7;
    const int arow = (32 * wm + l15) * 64;   // this lane's row inside an operand tile ([row][64]); + 16 * 64 for the second row tile
    float4v acc1[2][4];

    // one tile step: wait for tile (chunk, SLOT) -- ring stage `stage` --, issue the tile RING - 1 ahead into the stage tile seq - 1 just
    // left, multiply.  `stage` / `istage` run mod RING on the scalar unit.
    int stage = 5, istage = 4;  // tile 0 sits in stage 5; the first tile issued by the loop (seq RING - 1 = 7) goes to stage (7 + 5) % 8
    auto step = [&](auto slot_c, int chunk) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        const int chunk_w = chunk + cbase < kChunks ? chunk + cbase : chunk + cbase - kChunks;   // the chunk of the weights this step multiplies
        // this wave's 2 pieces of the tile have landed when at most its pieces of the younger tiles are in flight
        constexpr int left_in_chunk = kTilesPerChunk - 1 - slot;
        int younger = (kChunks - 1 - chunk) * kTilesPerChunk + left_in_chunk;  // tiles after this one
        if (younger > kRing - 2) younger = kRing - 2;
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's LDS writes -- xs, b1s, the g tile -- are complete before the barrier)
        __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile are visible; every wave has left the tile before it (its stage is free)
        asm volatile("" ::: "memory");
        {
            constexpr int ahead = slot + kRing - 1;
            constexpr int aslot = ahead % kTilesPerChunk, achunk = ahead / kTilesPerChunk;
            if (chunk + achunk < kChunks) issue(std::integral_constant<int, aslot>{}, chunk + achunk, istage);
            istage = istage == kRing - 1 ? 0 : istage + 1;
        }
        const half_t* tile = ring + stage * kTileHalfs;
        stage = stage == kRing - 1 ? 0 : stage + 1;

        if constexpr (slot < kKT1) {
            // ---- first projection, K tile `slot`: acc1[i][jj] += W1 tile rows 64 wn + 16 jj + l15 (A operand) x xn rows (B operand) ----
            if constexpr (slot == 0) {
                // start from b1: the wave's 4 packed column tiles [value 0 | gate 0 | value 1 | gate 1], columns 128 chunk + 64 wn + 16 jj + 4 g
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const half4v b = *reinterpret_cast<const half4v*>(b1s + 128 * chunk_w + 64 * wn + 16 * jj + 4 * g);
                    const float4v b0 = float4v{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
                    acc1[0][jj] = b0;
                    acc1[1][jj] = b0;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot_off = (((kk * 4 + g) ^ swz) << 3);
                half8v wf[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) wf[jj] = *reinterpret_cast<const half8v*>(tile + (64 * wn + 16 * jj + l15) * 64 + slot_off);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (!MV_FFN_ABL(1)) acc1[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj], xreg[i][2 * slot + kk], acc1[i][jj], 0, 0, 0);
            }
            if constexpr (slot == kKT1 - 1) {
                // ---- GEGLU of the chunk -> g[row][hidden] (fp16, operand layout): hidden units 32 wn + 16 pr + 4 g .. + 3 of rows 16 i + l15.
                // (The previous chunk's readers of the g tile left it 5 barriers ago.) ----
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 32 * wm + 16 * i + l15;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const float4v v = acc1[i][2 * pr], gt = acc1[i][2 * pr + 1];
                        half4v o = half4v{(half_t)(v[0] * mv_gelu(gt[0])), (half_t)(v[1] * mv_gelu(gt[1])),
                                          (half_t)(v[2] * mv_gelu(gt[2])), (half_t)(v[3] * mv_gelu(gt[3]))};
                        if (MV_FFN_ABL(0)) o = half4v{(half_t)(v[0] * gt[0]), (half_t)(v[1] * gt[1]), (half_t)(v[2] * gt[2]), (half_t)(v[3] * gt[3])};
                        const int hcol = 32 * wn + 16 * pr + 4 * g;  // hidden index inside the chunk
                        if (!MV_FFN_ABL(3)) *reinterpret_cast<half4v*>(gbuf + row * 64 + ((((hcol >> 3) ^ (row & 7)) << 3) | (hcol & 7))) = o;
                    }
                }
            }
        } else {
            // ---- second projection, W2 tile t = slot - 5: acc2 += W2 tile rows (output columns, A operand) x g rows (B operand), K = 64 ----
            constexpr int t = slot - kKT1;
            constexpr int TN = t < 2 ? 4 : 2;                 // the third tile holds output columns 256 .. 319 only
            const int wrow0 = (t < 2 ? 64 : 32) * wn + l15;   // this wave's first weight row inside the tile
            const half_t* gt_ = gbuf + arow;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot_off = (((kk * 4 + g) ^ swz) << 3);
                half8v gf[2], wf[TN];
#pragma unroll
                for (int i = 0; i < 2; ++i) gf[i] = *reinterpret_cast<const half8v*>(gt_ + i * (16 * 64) + slot_off);
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) wf[jj] = *reinterpret_cast<const half8v*>(tile + (wrow0 + 16 * jj) * 64 + slot_off);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        if (!MV_FFN_ABL(2)) acc2[i][4 * t + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj], gf[i], acc2[i][4 * t + jj], 0, 0, 0);
            }
        }
    };
#pragma unroll 1
    for (int chunk = 0; chunk < kChunks; ++chunk) {
        step(std::integral_constant<int, 0>{}, chunk); step(std::integral_constant<int, 1>{}, chunk);
        step(std::integral_constant<int, 2>{}, chunk); step(std::integral_constant<int, 3>{}, chunk);
        step(std::integral_constant<int, 4>{}, chunk); step(std::integral_constant<int, 5>{}, chunk);
        step(std::integral_constant<int, 6>{}, chunk); step(std::integral_constant<int, 7>{}, chunk);
    }

    // ---- epilogue: the accumulator layout gives a lane 4 columns of ONE row; the fp32 tile goes through the idle LDS (the ring)
    // in two halves of 64 rows (64 x 324 x 4 B = 83 KB) so that the residual loads and the stores are 16 bytes per lane on consecutive
    // bytes of a row; the residual add is fp32, rounded once ----
    float* const stg = reinterpret_cast<float*>(lds);
    for (int hrow = 0; hrow < 2; ++hrow) {
        __syncthreads();  // every wave is done with the operand tiles (hrow 0) / with reading the previous half (hrow 1)
        if ((wm >> 1) == hrow) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 32 * (wm & 1) + 16 * i + l15;
#pragma unroll
                for (int jj = 0; jj < 10; ++jj) *reinterpret_cast<float4v*>(stg + row * kSLd + ocol(jj) + 4 * g) = acc2[i][jj];
            }
        }
        __syncthreads();
        // 64 rows x 40 chunks of 8 columns: thread -> (row, chunk), 5 iterations of 512 threads
        for (int idx = tid; idx < 64 * (kC / 8); idx += 512) {
            const int row = idx / (kC / 8), ch = idx - row * (kC / 8);
            const int grow = m0 + 64 * hrow + row;
            if (grow >= Mi) continue;
            const float4v v0 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch);
            const float4v v1 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch + 4);
            const half8v r = *reinterpret_cast<const half8v*>(p.residual + (long)grow * p.ldr + 8 * ch);
            half8v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)(v0[e] + (float)r[e]);
                o[4 + e] = (half_t)(v1[e] + (float)r[4 + e]);
            }
            *reinterpret_cast<half8v*>(p.out + (long)grow * p.ldo + 8 * ch) = o;
        }
    }
}

}  // namespace

extern "C" int mv_ffn_geglu_f16(const mv_ffn_desc* d, void* stream) {
    MV_REQUIRE(d != nullptr, "mv_ffn_geglu_f16: null descriptor");
    MV_REQUIRE(d->x && d->w1 && d->w2 && d->ln_gamma && d->ln_beta && d->residual && d->out, "mv_ffn_geglu_f16: null pointer");
    MV_REQUIRE(d->C == kC && d->H == kH, "mv_ffn_geglu_f16: built for C = %d, hidden = %d (got C = %d, hidden = %d): use the three-launch form",
               kC, kH, d->C, d->H);
    MV_REQUIRE(d->M > 0 && d->M < 0x7fffffffL, "mv_ffn_geglu_f16: bad M");
    MV_REQUIRE(d->ldx % 8 == 0 && d->ldr % 8 == 0 && d->ldo % 8 == 0 && d->ldx >= kC && d->ldr >= kC && d->ldo >= kC,
               "mv_ffn_geglu_f16: leading dimensions must be multiples of 8 and >= C");
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    MV_REQUIRE(al16(d->x) && al16(d->w1) && al16(d->w2) && al16(d->residual) && al16(d->out) && al16(d->ln_gamma) && al16(d->ln_beta) &&
               (!d->bias1 || al16(d->bias1)) && (!d->bias2 || (reinterpret_cast<uintptr_t>(d->bias2) & 7) == 0),
               "mv_ffn_geglu_f16: pointers must be 16-byte aligned (bias2: 8)");
    MV_REQUIRE(d->ln_eps > 0.f, "mv_ffn_geglu_f16: ln_eps must be positive");
    FfnArgs a;
    a.x = (const half_t*)d->x; a.gamma = (const half_t*)d->ln_gamma; a.beta = (const half_t*)d->ln_beta;
    a.w1 = (const half_t*)d->w1; a.bias1 = (const half_t*)d->bias1;
    a.w2 = (const half_t*)d->w2; a.bias2 = (const half_t*)d->bias2; a.residual = (const half_t*)d->residual; a.out = (half_t*)d->out;
    a.M = d->M; a.ldx = d->ldx; a.ldr = d->ldr; a.ldo = d->ldo; a.eps = d->ln_eps;
    a.w1_bytes = (unsigned)(2L * kH * kC * 2); a.w2_bytes = (unsigned)((long)kC * kH * 2);
    a.rotate = (d->flags & 1) ? 1 : 0;
    a.ablate = (d->flags >> 8) & 0xff;
    constexpr int smem = kLdsHalfs * (int)sizeof(half_t);
    static_assert(smem <= 160 * 1024, "ffn tile does not fit LDS");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_geglu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        MV_REQUIRE(e == hipSuccess, "mv_ffn_geglu_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const unsigned nblk = (unsigned)((d->M + kBM - 1) / kBM);
    hipLaunchKernelGGL(ffn_geglu_kernel, dim3(nblk), dim3(512), smem, (hipStream_t)stream, a);
    MV_CHECK_LAUNCH("mv_ffn_geglu_f16");
    return MV_OK;
}
