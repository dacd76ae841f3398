// xab.hip -- the text cross-attention sub-block of a spatial BasicTransformerBlock as ONE kernel for the level-0 stream
// (C = 320 = 8 heads x 40, at most 80 keys in one softmax group):
//
//     out = x + to_out( softmax( q K^T * scale ) V ) + b_o,      q = LayerNorm(x) Wq^T
//
// (musev/models/attention.py:345-396 -- norm2 / attn2 of the spatial block; attention_processor.py:233-300 with the prompt's K / V
// projected once per prompt by the caller: they are constant over the denoise loop).  The three launches it replaces (LayerNorm-folded
// to_q projection, the resident-K/V cross-attention, to_out + residual) move q and the attention output through HBM once each way:
// 136 of the chain's 272 MB at M = 53 248.
//
// It is the fused temporal sub-block's structure (tsa.hip) with the keys / values coming from outside:
//   * a block owns 128 consecutive rows (all of one key batch) for the whole chain; prologue: two-pass LayerNorm in registers, the
//     normalised rows as MFMA-operand fragments (2 x 10 per wave, in registers for the whole kernel);
//   * the heads are walked in PAIRS: per pair  [q_a | q_b] = xn . Wq_pair^T  with the pair's 128 packed weight rows [q_a (40) | 24 zero
//     rows | q_b (40) | 24 zero rows] (K = 320: 5 ring tiles), written as fp16 into two LDS operand tiles (one per head);
//   * per head, K_h and V_h ([keys][40 h .. 40 h + 63] column windows of the caller's row-major K / V: the 24 columns past the head
//     meet zero columns of q / zero rows of the packed to_out weights) arrive as two more ring tiles; wave w runs the attention of rows
//     16 w .. 16 w + 15 on the matrix cores: S^T = K Q^T (5 key tiles: one 32-deep + one 16-deep step each, the tail in its OWN
//     accumulator -- see tsa.hip), softmax over the lane's 20 keys + two xor-shuffles, O^T = V^T P^T (3 x 5 16-deep steps, V^T through
//     ds_read_b64_tr_b16), O_h written over q_h in the head's operand tile;
//   * acc += O_h . Wo[:, head]^T (3 ring tiles per head, K = 64 with 40 used), the output accumulators (32 rows x 160 columns per wave)
//     in registers across the heads;
//   * ONE ring of 16-KiB LDS tiles carries the weights and the keys / values (buffer_load ... lds, counted vmcnt waits, one raw s_barrier
//     per tile): 15 tiles per head pair, 60 per block.  A tile is issued RING - 2 steps ahead into the stage of the tile TWO steps back,
//     so that the step of a V tile still finds its K tile in the stage before it;
//   * the epilogue adds the residual (the block's own rows of x) in fp32.
// HBM sees x twice (rows, residual) and the output once: 3 x M x 320 x 2 bytes.
#include "common.h"
#include <type_traits>

namespace {

struct XabArgs {
    const half_t* x;         // [M][ldx]; also the residual
    const half_t* gamma;     // [C] LayerNorm weight
    const half_t* beta;      // [C] LayerNorm bias
    const half_t* wq;        // [4 pairs][128][C] packed rows per head pair: q_a | 24 zero rows | q_b | 24 zero rows
    const half_t* k;         // [key batches][len][ldk]: column 40 h + d = key component d of head h
    const half_t* v;         // [key batches][len][ldv]
    const half_t* wo;        // [C][heads * 64]: column 64 h + d = to_out column 40 h + d (d < 40), zero for 40 <= d < 64
    const half_t* bias_o;    // [C] or nullptr
    half_t* out;             // [M][ldo]
    long M;
    int rows_per_kvb;        // rows of x per key batch (a multiple of 128)
    int len;                 // keys per batch (<= 80)
    int ldk, ldv, ldx, ldo;
    float eps, scale_log2e;
    unsigned wq_bytes, wo_bytes;
    int rotate;              // blocks start their walk over the head pairs at different pairs
};

constexpr int kC = 320, kHeads = 8, kD = 40, kPairs = 4;
constexpr int kBM = 128;
constexpr int kKT1 = kC / 64;               // 5 K tiles of the q projection of a pair
constexpr int kTilesPerPair = kKT1 + 4 + 6; // + K_a, V_a, K_b, V_b + 3 tiles of Wo per head
constexpr int kTiles = kPairs * kTilesPerPair;
constexpr int kTileHalfs = 128 * 64;        // one operand tile: [128 rows][64 k] halfs = 16 KiB
constexpr int kRing = 7;                    // LDS stages; stages 0 .. 4 first carry the normalised rows to the registers
constexpr int kAhead = kRing - 2;           // a tile is issued this many steps ahead (the tile one step back stays readable)
constexpr int kWoLd = kHeads * 64;
constexpr int kKeyTiles = 5;                // 80 keys
constexpr int kLdsHalfs = (kRing + 2) * kTileHalfs;   // ring + the two q / O operand tiles (one per head of the pair)
constexpr unsigned kOob = 0x80000000u;
constexpr int kSLd = kC + 4;                // epilogue: floats per staging row
static_assert(64 * kSLd * 4 <= kRing * kTileHalfs * 2, "a 64-row fp32 staging tile must fit the ring");
static_assert(kRing - kKT1 >= 2 && kAhead <= kTilesPerPair, "prologue: tiles 0, 1 beside the x tiles, tiles 2 .. 4 behind them");

__global__ __launch_bounds__(512, 2) void xab_kernel(const XabArgs p) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* const ring = lds;                           // [kRing][128][64]
    half_t* const xs = lds;                             // [5][128][64]: stages 0 .. 4, until the rows are in registers
    half_t* const qb = ring + kRing * kTileHalfs;       // [2][128][64]: q_h (columns 0-39, zero up to 63) of the pair's heads; O_h afterwards

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int nblk = (int)((p.M + kBM - 1) / kBM);
    const int bid = mv_xcd_remap(blockIdx.x, nblk);
    const int m0 = bid * kBM;
    const int Mi = (int)p.M;
    const int kvb = m0 / p.rows_per_kvb;
    const int cbase = p.rotate ? (bid * 3) % kPairs : 0;

    const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq, 0, p.wq_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo, 0, p.wo_bytes, 0x00020000);
    // descriptors over the rows [0, len) of this block's key batch (a head's 64-column window past its 40 columns reads the next head's --
    // finite -- values; past the row: zero, see issue)
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k + (long)kvb * p.len * p.ldk), 0,
                                                                        (unsigned)(((p.len - 1) * p.ldk + kC) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v + (long)kvb * p.len * p.ldv), 0,
                                                                        (unsigned)(((p.len - 1) * p.ldv + kC) * 2), 0x00020000);

    // ---- the block's rows of x: thread -> (row tid / 4, 16-byte chunks (tid % 4) + 4 q, q = 0 .. 9); requested AHEAD of the ring's first tiles ----
    const int xrow = tid >> 2, xq = tid & 3;
    half8v xv[10];
    {
        const int row = m0 + xrow;
        const half_t* xr = p.x + (long)(row < Mi ? row : Mi - 1) * p.ldx + 8 * xq;
#pragma unroll
        for (int q = 0; q < 10; ++q) xv[q] = *reinterpret_cast<const half8v*>(xr + 32 * q);
    }

    // ---- LDS-DMA geometry (as in ffn.hip / tsa.hip): a tile is 16 pieces of 8 rows x 128 B; wave w issues pieces w and w + 8.  Lane ->
    // (row lane / 8, 16-byte slot lane % 8); the swizzle (slot ^ row) lives on the SOURCE address ----
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off1[2], off2[2], offk[2], offv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * (wave + 8 * q) + lrow;  // tile row 0 .. 127
        off1[q] = ((unsigned)r * (unsigned)kC + lsl * 8u) * 2u;
        off2[q] = ((unsigned)r * (unsigned)kWoLd + lsl * 8u) * 2u;
        offk[q] = r < p.len ? ((unsigned)r * (unsigned)p.ldk + lsl * 8u) * 2u : kOob;   // tile row = key; rows past the keys read zero
        offv[q] = r < p.len ? ((unsigned)r * (unsigned)p.ldv + lsl * 8u) * 2u : kOob;
    }
    // tile (pair, slot): slots 0 .. 4 = K tiles of the pair's packed q rows, 5 / 7 = keys of head a / b, 6 / 8 = values, 9 .. 11 / 12 .. 14 =
    // to_out column tiles of head a / b
    auto issue = [&](auto slot_c, int pair_i, int stage) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        const int pair = pair_i + cbase < kPairs ? pair_i + cbase : pair_i + cbase - kPairs;
        half_t* dst = ring + stage * kTileHalfs;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_t* d = dst + (wave + 8 * q) * (8 * 64);
            if constexpr (slot < kKT1) {
                const unsigned so = ((unsigned)(128 * pair) * (unsigned)kC + 64u * (unsigned)slot) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, (__attribute__((address_space(3))) void*)d, 16, (int)off1[q], (int)so, 0, 0);
            } else if constexpr (slot < kKT1 + 4) {
                constexpr int e = (slot - kKT1) >> 1;
                const int col0 = kD * (2 * pair + e);                       // the head's first column
                const unsigned so = (unsigned)col0 * 2u;
                const bool inrow = col0 + 8 * (int)lsl < kC;                // (the last head's window runs past the row: those chunks read zero)
                if constexpr (((slot - kKT1) & 1) == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (__attribute__((address_space(3))) void*)d, 16, (int)(inrow ? offk[q] : kOob), (int)so, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (__attribute__((address_space(3))) void*)d, 16, (int)(inrow ? offv[q] : kOob), (int)so, 0, 0);
            } else {
                constexpr int e = (slot - kKT1 - 4) / 3, t = (slot - kKT1 - 4) % 3;
                const int r = 128 * t + 8 * (wave + 8 * q) + lrow;
                const unsigned vo = r < kC ? off2[q] : kOob;
                const unsigned so = ((unsigned)(128 * t) * (unsigned)kWoLd + 64u * (unsigned)(2 * pair + e)) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (__attribute__((address_space(3))) void*)d, 16, (int)vo, (int)so, 0, 0);
            }
        }
    };
    // tile seq lives in stage (seq + 5) % RING
    issue(std::integral_constant<int, 0>{}, 0, 5);
    issue(std::integral_constant<int, 1>{}, 0, 6);

    // ---- the q / O operand tiles start as zeros: their columns 48 .. 63 are never written and must multiply as zeros ----
    for (int i = tid; i < 2 * kTileHalfs / 8; i += 512) reinterpret_cast<uint4*>(qb)[i] = uint4{0, 0, 0, 0};

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
    // ---- output accumulators, starting from the to_out bias: acc2[i][jj] = rows 32 wm + 16 i + l15, columns ocol(jj) + 4 g .. + 3 ----
    auto ocol = [&](int jj) { return jj < 4 ? 64 * wn + 16 * jj : jj < 8 ? 128 + 64 * wn + 16 * (jj - 4) : 256 + 32 * wn + 16 * (jj - 8); };
    float4v acc2[2][10];
#pragma unroll
    for (int jj = 0; jj < 10; ++jj) {
        float4v b0 = float4v{0.f, 0.f, 0.f, 0.f};
        if (p.bias_o) {
            const half4v b = *reinterpret_cast<const half4v*>(p.bias_o + ocol(jj) + 4 * g);
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
    issue(std::integral_constant<int, 2>{}, 0, 0);
    issue(std::integral_constant<int, 3>{}, 0, 1);
    issue(std::integral_constant<int, 4>{}, 0, 2);

    const int swz = l15 & 7;
    const int arow = (32 * wm + l15) * 64;   // this lane's row inside an operand tile ([row][64]); + 16 * 64 for the second row tile
    float4v acc1[2][3];
    typedef __attribute__((address_space(3))) half_t lds_half_t;

    int stage = 5, pstage = 4, istage = 3;  // tile 0 sits in stage 5; the first tile issued by the loop (seq kAhead = 5) goes to stage (5 + 5) % 7
    auto step = [&](auto slot_c, int pair) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        constexpr int left_in_pair = kTilesPerPair - 1 - slot;
        int younger = (kPairs - 1 - pair) * kTilesPerPair + left_in_pair;  // tiles after this one
        if (younger > kAhead - 1) younger = kAhead - 1;                    // ... of which already issued: at most kAhead - 1
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's LDS writes -- xs, the q / O tiles -- are complete before the barrier)
        __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile are visible; every wave has left the tile TWO steps back (its stage is free)
        asm volatile("" ::: "memory");
        {
            constexpr int ahead = slot + kAhead;
            constexpr int aslot = ahead % kTilesPerPair, ahead_p = ahead / kTilesPerPair;
            if (pair + ahead_p < kPairs) issue(std::integral_constant<int, aslot>{}, pair + ahead_p, istage);
            istage = istage == kRing - 1 ? 0 : istage + 1;
        }
        const half_t* tile = ring + stage * kTileHalfs;
        const half_t* ptile = ring + pstage * kTileHalfs;   // the tile one step back (the keys, on a values step)
        pstage = stage;
        stage = stage == kRing - 1 ? 0 : stage + 1;

        if constexpr (slot < kKT1) {
            // ---- q projection of the pair, K tile `slot`: acc1[i][jj] += Wq tile rows 64 wn + 16 jj + l15 (A operand) x xn rows (B operand);
            // wave column wn = head a / b of the pair; the fourth 16-row tile of a head's 64 packed rows is all zeros and is skipped ----
            if constexpr (slot == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) acc1[i][jj] = float4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot_off = (((kk * 4 + g) ^ swz) << 3);
                half8v wf[3];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) wf[jj] = *reinterpret_cast<const half8v*>(tile + (64 * wn + 16 * jj + l15) * 64 + slot_off);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
                        acc1[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj], xreg[i][2 * slot + kk], acc1[i][jj], 0, 0, 0);
            }
            if constexpr (slot == kKT1 - 1) {
                // ---- q_h (fp16, operand layout): columns 16 jj + 4 g .. + 3 of rows 32 wm + 16 i + l15 -> tile wn.  (The previous pair's readers
                // of the tiles -- its to_out steps -- left them at least 5 barriers ago; the first reader of these values is 2 barriers ahead.) ----
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 32 * wm + 16 * i + l15;
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {
                        const float4v v = acc1[i][jj];
                        const int kcol = 16 * jj + 4 * g;
                        *reinterpret_cast<half4v*>(qb + wn * kTileHalfs + row * 64 + ((((kcol >> 3) ^ (row & 7)) << 3) | (kcol & 7))) =
                            half4v{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    }
                }
            }
        } else if constexpr (slot < kKT1 + 4) {
            if constexpr (((slot - kKT1) & 1) == 1) {
                // ---- attention of head e over rows 16 wave .. + 15: keys in `ptile`, values in `tile` (rows = keys, columns 0 .. 39 = the head) ----
                constexpr int e = (slot - kKT1) >> 1;
                half_t* const qtile = qb + e * kTileHalfs;
                const half_t* r0 = qtile + (16 * wave + l15) * 64;                       // this lane's query row (B operand)
                const half8v qf = *reinterpret_cast<const half8v*>(r0 + ((g ^ swz) << 3));                 // q d = 8 g .. + 7
                half4v qt = half4v{0, 0, 0, 0};
                if (g < 2) qt = *reinterpret_cast<const half4v*>(r0 + ((4 ^ swz) << 3) + 4 * g);           // q d = 32 + 4 g .. + 3
                float4v s[kKeyTiles];
#pragma unroll
                for (int kt = 0; kt < kKeyTiles; ++kt) {
                    const half_t* kr = ptile + (16 * kt + l15) * 64;                     // key 16 kt + l15 (A operand); (16 kt + l15) & 7 == swz
                    const half8v kf = *reinterpret_cast<const half8v*>(kr + ((g ^ swz) << 3));
                    half4v k4 = half4v{0, 0, 0, 0};
                    if (g < 2) k4 = *reinterpret_cast<const half4v*>(kr + ((4 ^ swz) << 3) + 4 * g);
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const float4v st = __builtin_amdgcn_mfma_f32_16x16x16f16(k4, qt, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    s[kt] += st;   // s[kt][r] = S[query l15][key 16 kt + 4 g + r] (raw dot products)
                }
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < kKeyTiles; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (16 * kt + 4 * g + r >= p.len) s[kt][r] = -INFINITY;
                        m = fmaxf(m, s[kt][r]);
                    }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const float nms = -m * p.scale_log2e;   // (scale > 0: the maximum of the raw scores is the maximum of the scaled ones)
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < kKeyTiles; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], p.scale_log2e, nms));
                        l += s[kt][r];
                    }
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
                const float inv = 1.0f / l;
                half4v pf[kKeyTiles];
#pragma unroll
                for (int kt = 0; kt < kKeyTiles; ++kt)
                    pf[kt] = half4v{(half_t)(s[kt][0] * inv), (half_t)(s[kt][1] * inv), (half_t)(s[kt][2] * inv), (half_t)(s[kt][3] * inv)};
                // V^T fragments: the 16-lane group g addresses the [4 keys][16 d] block of keys 16 kt + 4 g .. + 3 row-wise (lane -> key 4 g + l15 / 4,
                // d = 16 dt + 4 (l15 % 4)): ds_read_b64_tr_b16 hands lane l15 the column d = 16 dt + l15 of those 4 keys
                float4v o[3] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kt = 0; kt < kKeyTiles; ++kt) {
                    const int krow = 16 * kt + 4 * g + (l15 >> 2);
                    const lds_half_t* v3 = (const lds_half_t*)(tile + krow * 64);
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        const int vslot = 2 * dt + ((l15 & 3) >> 1);
                        const lds_half_t* va = v3 + (((vslot ^ (krow & 7)) << 3) + 4 * (l15 & 1));
                        const short4v tv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v, tv), pf[kt], o[dt], 0, 0, 0);
                    }
                }
                // O_h[query l15][d = 16 dt + 4 g + r] over q_h: row 16 wave + l15, column d (d >= 40: the columns past the head's values times P --
                // finite; the zero rows of the packed to_out weights drop them); columns 48 .. 63 stay zero
                half_t* orow = qtile + (16 * wave + l15) * 64;
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
                    *reinterpret_cast<half4v*>(orow + ((((2 * dt + (g >> 1)) ^ swz) << 3) | (4 * (g & 1)))) =
                        half4v{(half_t)o[dt][0], (half_t)o[dt][1], (half_t)o[dt][2], (half_t)o[dt][3]};
            }
            // (a keys step only waits for its tile and passes the barrier: its tile is read on the values step behind it)
        } else {
            // ---- to_out of head e, Wo tile t: acc2 += Wo tile rows (output columns, A operand) x O_h rows (B operand), K = 64 (40 used) ----
            constexpr int e = (slot - kKT1 - 4) / 3, t = (slot - kKT1 - 4) % 3;
            constexpr int TN = t < 2 ? 4 : 2;                 // the third tile holds output columns 256 .. 319 only
            const int wrow0 = (t < 2 ? 64 : 32) * wn + l15;   // this wave's first weight row inside the tile
            const half_t* gt_ = qb + e * kTileHalfs + arow;
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
                        acc2[i][4 * t + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj], gf[i], acc2[i][4 * t + jj], 0, 0, 0);
            }
        }
    };
#pragma unroll 1
    for (int pair = 0; pair < kPairs; ++pair) {
        step(std::integral_constant<int, 0>{}, pair); step(std::integral_constant<int, 1>{}, pair);
        step(std::integral_constant<int, 2>{}, pair); step(std::integral_constant<int, 3>{}, pair);
        step(std::integral_constant<int, 4>{}, pair); step(std::integral_constant<int, 5>{}, pair);
        step(std::integral_constant<int, 6>{}, pair); step(std::integral_constant<int, 7>{}, pair);
        step(std::integral_constant<int, 8>{}, pair); step(std::integral_constant<int, 9>{}, pair);
        step(std::integral_constant<int, 10>{}, pair); step(std::integral_constant<int, 11>{}, pair);
        step(std::integral_constant<int, 12>{}, pair); step(std::integral_constant<int, 13>{}, pair);
        step(std::integral_constant<int, 14>{}, pair);
    }

    // ---- epilogue: the fp32 tile goes through the idle ring in two halves of 64 rows so that the residual loads and the stores are
    // 16 bytes per lane on consecutive bytes of a row; the residual add is fp32, rounded once ----
    float* const stg = reinterpret_cast<float*>(lds);
    for (int hrow = 0; hrow < 2; ++hrow) {
        __syncthreads();
        if ((wm >> 1) == hrow) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 32 * (wm & 1) + 16 * i + l15;
#pragma unroll
                for (int jj = 0; jj < 10; ++jj) *reinterpret_cast<float4v*>(stg + row * kSLd + ocol(jj) + 4 * g) = acc2[i][jj];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * (kC / 8); idx += 512) {
            const int row = idx / (kC / 8), ch = idx - row * (kC / 8);
            const int grow = m0 + 64 * hrow + row;
            if (grow >= Mi) continue;
            const float4v v0 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch);
            const float4v v1 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch + 4);
            const half8v r = *reinterpret_cast<const half8v*>(p.x + (long)grow * p.ldx + 8 * ch);
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

extern "C" int mv_xattn_block_f16(const mv_xab_desc* d, void* stream) {
    MV_REQUIRE(d != nullptr, "mv_xattn_block_f16: null descriptor");
    MV_REQUIRE(d->x && d->wq && d->k && d->v && d->wo && d->ln_gamma && d->ln_beta && d->out, "mv_xattn_block_f16: null pointer");
    MV_REQUIRE(d->C == kC && d->heads == kHeads && d->d == kD, "mv_xattn_block_f16: built for C = %d = %d heads x %d (got C = %d, %d heads x %d): use the three-launch form",
               kC, kHeads, kD, d->C, d->heads, d->d);
    MV_REQUIRE(d->M > 0 && d->M < 0x7fffffffL, "mv_xattn_block_f16: M = %ld out of range", (long)d->M);
    MV_REQUIRE(d->len >= 1 && d->len <= 16 * kKeyTiles, "mv_xattn_block_f16: %d keys not in [1, %d]", d->len, 16 * kKeyTiles);
    MV_REQUIRE(d->rows_per_kvb > 0 && d->rows_per_kvb % kBM == 0, "mv_xattn_block_f16: rows_per_kvb = %d must be a positive multiple of %d (a block's rows share their keys)",
               d->rows_per_kvb, kBM);
    MV_REQUIRE(d->ldx % 8 == 0 && d->ldo % 8 == 0 && d->ldx >= kC && d->ldo >= kC && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldk >= kC && d->ldv >= kC,
               "mv_xattn_block_f16: leading dimensions must be multiples of 8 and >= C");
    const long nkvb = (d->M + d->rows_per_kvb - 1) / d->rows_per_kvb;
    MV_REQUIRE(nkvb * d->len * (long)(d->ldk > d->ldv ? d->ldk : d->ldv) * 2 < 0x7fffffffL, "mv_xattn_block_f16: keys / values span 2 GiB or more");
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    MV_REQUIRE(al16(d->x) && al16(d->wq) && al16(d->wo) && al16(d->k) && al16(d->v) && al16(d->out) && al16(d->ln_gamma) && al16(d->ln_beta) &&
               (!d->bias_o || (reinterpret_cast<uintptr_t>(d->bias_o) & 7) == 0),
               "mv_xattn_block_f16: pointers must be 16-byte aligned (bias_o: 8)");
    MV_REQUIRE(d->ln_eps > 0.f && d->scale > 0.f, "mv_xattn_block_f16: ln_eps and scale must be positive");
    XabArgs a;
    a.x = (const half_t*)d->x; a.gamma = (const half_t*)d->ln_gamma; a.beta = (const half_t*)d->ln_beta;
    a.wq = (const half_t*)d->wq; a.k = (const half_t*)d->k; a.v = (const half_t*)d->v; a.wo = (const half_t*)d->wo;
    a.bias_o = (const half_t*)d->bias_o; a.out = (half_t*)d->out;
    a.M = d->M; a.rows_per_kvb = d->rows_per_kvb; a.len = d->len; a.ldk = d->ldk; a.ldv = d->ldv; a.ldx = d->ldx; a.ldo = d->ldo;
    a.eps = d->ln_eps; a.scale_log2e = d->scale * 1.4426950408889634f;
    a.wq_bytes = (unsigned)((long)kPairs * 128 * kC * 2); a.wo_bytes = (unsigned)((long)kC * kWoLd * 2);
    a.rotate = (d->flags & 1) ? 1 : 0;
    constexpr int smem = kLdsHalfs * (int)sizeof(half_t);
    static_assert(smem <= 160 * 1024, "xab tiles do not fit LDS");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xab_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        MV_REQUIRE(e == hipSuccess, "mv_xattn_block_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const unsigned nblk = (unsigned)((d->M + kBM - 1) / kBM);
    hipLaunchKernelGGL(xab_kernel, dim3(nblk), dim3(512), smem, (hipStream_t)stream, a);
    MV_CHECK_LAUNCH("mv_xattn_block_f16");
    return MV_OK;
}
