// tsa.hip -- one temporal self-attention sub-block of a BasicTransformerBlock as ONE kernel for the level-0 stream
// (C = 320 = 8 heads x 40, T <= 16 frames):
//
//     out = x + to_out( softmax_T( q k^T * scale ) v ),   [q | k | v] = LayerNorm(x) Wqkv^T     over the T frames of every pixel
//
// (musev/models/attention.py:293-345 -- norm1 / attn1 and norm2 / attn2 of the temporal BasicTransformerBlock,
// double_self_attention = True; musev/models/temporal_transformer.py:250-279: the block runs on "(b h w) t c" sequences of T frames).
// The three launches it replaces (LayerNorm-folded QKV projection, mv_temporal_attention_f16, to_out + residual) move the [M, 3 C]
// projection through HBM twice and the attention output once more: 410 of the chain's 480 MB at M = 53 248, and 127 us per
// sub-block, 24 sub-blocks per denoise step (profiles/r04w_gemm_roofline_by_problem.log, r04z_rocprofv3_kernel_stats.csv).
//
// It is the fused feed-forward's structure (ffn.hip) with the attention in the place of the GEGLU gate:
//   * a block owns 8 PIXELS of one batch item for the whole chain, as 128 rows: pixel j, frame t -> local row 16 j + t (frames T .. 15
//     repeat frame T - 1: computed, never stored -- a pixel's sequence then IS one 16-row MFMA tile);
//   * prologue: the rows are normalised in registers (two-pass LayerNorm, 4 lanes per row, fp32), rounded to fp16 and turned into
//     MFMA-operand fragments through LDS; a wave keeps the 2 x 10 fragments of its 32 rows (2 pixels) in registers for the whole kernel;
//   * the HEADS are the chunks: per head  S = xn . W_h^T  with W_h the head's 128 packed weight rows [q_h (40) | k_h (40) | v_h (40) |
//     8 zero rows] (K = 320: 5 ring tiles), written as fp16 into two LDS operand tiles; after one raw barrier wave w runs the T x T
//     attention of pixel w on the matrix cores (S^T = K Q^T: one 32-deep + one 16-deep step, softmax over a lane's 4 keys + two
//     xor-shuffles, O^T = V^T P^T: three 16-deep steps with V^T through ds_read_b64_tr_b16) and writes O_h (40 columns, zero up to
//     64) over the q / k columns of its own rows -- that tile is the B operand of  acc += O_h . Wo[:, head]^T  (3 ring tiles, K = 64),
//     with the output accumulators (32 rows x 160 columns per wave) in registers across the heads;
//   * both weight matrices stream through ONE ring of 16-KiB LDS tiles (buffer_load ... lds, counted vmcnt waits, one raw s_barrier
//     per tile, RING - 1 tiles in flight): 8 tiles per head, 64 per block = 1 MB from the L2 per block (the three launches: a
//     128 x 160 tile GEMM at 71 FLOP per byte staged into the CU; here 104 useful FLOP per byte);
//   * the epilogue adds the residual (the block's own rows of x) in fp32 and stores the rows of frames < T.
// HBM sees x twice (rows, residual) and the output once: 3 x M x 320 x 2 bytes.
//
// The 16-deep tail of S goes into its OWN accumulator and is added on the vector ALU: a v_mfma_f32_16x16x16_f16 issued right
// behind the v_mfma_f32_16x16x32_f16 whose result it accumulates onto read a half-written accumulator on the MI355X (round 5,
// profiles/r05d_debug_dump.log; see xattn_kernel in attention.hip).
#include "common.h"
#include <type_traits>

namespace {

struct TsaArgs {
    const half_t* x;         // [B T HW][ldx], rows in (b, t, p) order; also the residual
    const half_t* gamma;     // [C] LayerNorm weight
    const half_t* beta;      // [C] LayerNorm bias
    const half_t* wqkv;      // [heads][128][C] packed rows per head: q_h | k_h | v_h | 8 zero rows
    const half_t* wo;        // [C][heads * 64]: column 64 h + d = to_out column 40 h + d (d < 40), zero for 40 <= d < 64
    const half_t* bias_o;    // [C] or nullptr
    half_t* out;             // [B T HW][ldo]
    int B, T, HW;
    int ldx, ldo;
    float eps, scale_log2e;
    unsigned wqkv_bytes, wo_bytes;
    int rotate;              // blocks start their walk over the heads at different heads (see ffn.hip)
};

constexpr int kC = 320, kHeads = 8, kD = 40;
constexpr int kPix = 8;                  // pixels per block
constexpr int kBM = 16 * kPix;           // 128 local rows
constexpr int kKT1 = kC / 64;            // 5 K tiles of the QKV projection
constexpr int kTilesPerHead = kKT1 + 3;  // + 3 tiles of Wo (output columns 0-127, 128-255, 256-319)
constexpr int kTileHalfs = 128 * 64;     // one operand tile: [128 rows][64 k] halfs = 16 KiB
constexpr int kRing = 7;                 // LDS stages of the weight stream; stages 0 .. 4 first carry the normalised rows to the registers
constexpr int kWoLd = kHeads * 64;       // leading dimension of the packed to_out weights
constexpr int kLdsHalfs = (kRing + 2) * kTileHalfs;   // ring + the two q/k/v operand tiles (the first one doubles as the O_h tile)
constexpr unsigned kOob = 0x80000000u;
constexpr int kSLd = kC + 4;             // epilogue: floats per staging row
static_assert(64 * kSLd * 4 <= kRing * kTileHalfs * 2, "a 64-row fp32 staging tile must fit the ring");
static_assert(kRing - kKT1 >= 2 && kRing - 1 <= kTilesPerHead, "prologue: tiles 0 .. RING - 2 of head 0, the first two of them beside the x tiles");

__global__ __launch_bounds__(512, 2) void tsa_kernel(const TsaArgs p) {
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* const ring = lds;                           // [kRing][128][64]
    half_t* const xs = lds;                             // [5][128][64]: stages 0 .. 4, until the rows are in registers
    half_t* const qb = ring + kRing * kTileHalfs;       // [2][128][64]: packed q | k | v columns 0-63, 64-127 of the current head; tile 0 = O_h afterwards

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int bpi = p.HW / kPix;                        // blocks per batch item
    const int nblk = p.B * bpi;
    const int bid = mv_xcd_remap(blockIdx.x, nblk);
    const int bi = bid / bpi;
    const int p0 = (bid - bi * bpi) * kPix;
    const int Tn = p.T;
    // global row of local row 16 j + t
    auto grow = [&](int j, int t) -> long { return ((long)bi * Tn + (t < Tn ? t : Tn - 1)) * p.HW + p0 + j; };
    const int cbase = p.rotate ? (bid * 7) % kHeads : 0;

    const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqkv, 0, p.wqkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo, 0, p.wo_bytes, 0x00020000);

    // ---- the block's rows of x: thread -> (frame (tid / 4) / 8, pixel (tid / 4) % 8 -- consecutive row indices are consecutive rows of
    // one frame --, 16-byte chunks (tid % 4) + 4 q, q = 0 .. 9); requested AHEAD of the ring's first tiles ----
    const int xidx = tid >> 2, xq = tid & 3;
    const int xrow = 16 * (xidx & 7) + (xidx >> 3);     // local row 16 j + t
    half8v xv[10];
    {
        const half_t* xr = p.x + grow(xidx & 7, xidx >> 3) * p.ldx + 8 * xq;
#pragma unroll
        for (int q = 0; q < 10; ++q) xv[q] = *reinterpret_cast<const half8v*>(xr + 32 * q);
    }

    // ---- LDS-DMA geometry of the weight stream (as in ffn.hip): a tile is 16 pieces of 8 rows x 128 B; wave w issues pieces w and w + 8.
    // Lane -> (row lane / 8, 16-byte slot lane % 8); the swizzle (slot ^ row) lives on the SOURCE address ----
    const int lrow = lane >> 3;
    const unsigned lsl = (unsigned)((lane & 7) ^ lrow);
    unsigned off1[2], off2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 8 * (wave + 8 * q) + lrow;  // tile row 0 .. 127
        off1[q] = ((unsigned)r * (unsigned)kC + lsl * 8u) * 2u;
        off2[q] = ((unsigned)r * (unsigned)kWoLd + lsl * 8u) * 2u;
    }
    // tile (head, slot): slots 0 .. 4 = K tiles of the head's 128 packed QKV rows, 5 .. 7 = to_out column tiles, into ring stage `stage`
    auto issue = [&](auto slot_c, int head_i, int stage) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        const int head = head_i + cbase < kHeads ? head_i + cbase : head_i + cbase - kHeads;
        half_t* dst = ring + stage * kTileHalfs;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half_t* d = dst + (wave + 8 * q) * (8 * 64);
            if constexpr (slot < kKT1) {
                const unsigned so = ((unsigned)(128 * head) * (unsigned)kC + 64u * (unsigned)slot) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, (__attribute__((address_space(3))) void*)d, 16, (int)off1[q], (int)so, 0, 0);
            } else {
                constexpr int t = slot - kKT1;
                const int r = 128 * t + 8 * (wave + 8 * q) + lrow;
                const unsigned vo = r < kC ? off2[q] : kOob;
                const unsigned so = ((unsigned)(128 * t) * (unsigned)kWoLd + 64u * (unsigned)head) * 2u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, (__attribute__((address_space(3))) void*)d, 16, (int)vo, (int)so, 0, 0);
            }
        }
    };
    // tile seq lives in stage (seq + 5) % RING
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
    issue(std::integral_constant<int, 5>{}, 0, 3);

    const int swz = l15 & 7;
    const int arow = (32 * wm + l15) * 64;   // this lane's row inside an operand tile ([row][64]); + 16 * 64 for the second row tile
    float4v acc1[2][4];
    typedef __attribute__((address_space(3))) half_t lds_half_t;

    int stage = 5, istage = 4;  // tile 0 sits in stage 5; the first tile issued by the loop (seq RING - 1 = 6) goes to stage (6 + 5) % 7
    auto step = [&](auto slot_c, int head) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        constexpr int left_in_head = kTilesPerHead - 1 - slot;
        int younger = (kHeads - 1 - head) * kTilesPerHead + left_in_head;  // tiles after this one
        if (younger > kRing - 2) younger = kRing - 2;
        switch (younger) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's LDS writes -- xs, the O_h tile -- are complete before the barrier)
        __builtin_amdgcn_s_barrier();  // every wave's pieces of this tile are visible; every wave has left the tile before it (its stage is free)
        asm volatile("" ::: "memory");
        {
            constexpr int ahead = slot + kRing - 1;
            constexpr int aslot = ahead % kTilesPerHead, ahead_h = ahead / kTilesPerHead;
            if (head + ahead_h < kHeads) issue(std::integral_constant<int, aslot>{}, head + ahead_h, istage);
            istage = istage == kRing - 1 ? 0 : istage + 1;
        }
        const half_t* tile = ring + stage * kTileHalfs;
        stage = stage == kRing - 1 ? 0 : stage + 1;

        if constexpr (slot < kKT1) {
            // ---- QKV projection of the head, K tile `slot`: acc1[i][jj] += W_h tile rows 64 wn + 16 jj + l15 (A operand) x xn rows (B operand) ----
            if constexpr (slot == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc1[i][jj] = float4v{0.f, 0.f, 0.f, 0.f};
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
                        acc1[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jj], xreg[i][2 * slot + kk], acc1[i][jj], 0, 0, 0);
            }
            if constexpr (slot == kKT1 - 1) {
                // ---- the head's q | k | v (fp16, operand layout): packed columns 64 wn + 16 jj + 4 g .. + 3 of rows 32 wm + 16 i + l15 -> tile wn.
                // (The previous head's readers of tile 0 -- the to_out steps -- left it 5 barriers ago.) ----
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = 32 * wm + 16 * i + l15;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const float4v v = acc1[i][jj];
                        const int kcol = 16 * jj + 4 * g;   // column inside the tile
                        *reinterpret_cast<half4v*>(qb + wn * kTileHalfs + row * 64 + ((((kcol >> 3) ^ (row & 7)) << 3) | (kcol & 7))) =
                            half4v{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // the 8 waves' columns of every row are in place (LDS-DMA tiles stay in flight: no vmcnt wait)
                asm volatile("" ::: "memory");
                // ---- attention of pixel `wave`: rows 16 wave + t.  Packed columns: q 0-39 | k 40-79 | v 80-119 | zeros 120-127 ----
                {
                    const half_t* r0 = qb + (16 * wave + l15) * 64;          // this lane's row (query l15 as B operand, key l15 as A operand) in tile 0
                    const half_t* r1 = r0 + kTileHalfs;                      // ... in tile 1
                    const half8v qf = *reinterpret_cast<const half8v*>(r0 + ((g ^ swz) << 3));                       // q d = 8 g .. + 7
                    half4v qt = half4v{0, 0, 0, 0}, kt = half4v{0, 0, 0, 0};
                    if (g < 2) {
                        qt = *reinterpret_cast<const half4v*>(r0 + ((4 ^ swz) << 3) + 4 * g);                        // q d = 32 + 4 g .. + 3
                        kt = *reinterpret_cast<const half4v*>(r1 + ((1 ^ swz) << 3) + 4 * g);                        // k d = 32 + 4 g: packed 72 + 4 g
                    }
                    // k d = 8 g .. + 7: packed column 40 + 8 g = tile 0 slots 5, 6, 7 (g < 3), tile 1 slot 0 (g = 3)
                    const half8v kf = *reinterpret_cast<const half8v*>((g < 3 ? r0 : r1) + ((((g < 3 ? 5 + g : 0)) ^ swz) << 3));
                    float4v s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const float4v st = __builtin_amdgcn_mfma_f32_16x16x16f16(kt, qt, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    s += st;   // s[r] = S[query l15][key 4 g + r] (raw dot products)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r >= Tn) s[r] = -INFINITY;
                    float m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const float nms = -m * p.scale_log2e;
                    float pr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, nms));
                    float l = (pr[0] + pr[1]) + (pr[2] + pr[3]);
                    l += __shfl_xor(l, 16, 64);
                    l += __shfl_xor(l, 32, 64);
                    const float inv = 1.0f / l;
                    const half4v pf = half4v{(half_t)(pr[0] * inv), (half_t)(pr[1] * inv), (half_t)(pr[2] * inv), (half_t)(pr[3] * inv)};
                    // V^T fragments: the 16-lane group g addresses the [4 keys][16 d] block of keys 4 g .. + 3 row-wise (lane -> key 4 g + l15 / 4,
                    // d = 16 dt + 4 (l15 % 4)); v d sits at packed column 80 + d = tile 1 column 16 + d
                    const int krow = 16 * wave + 4 * g + (l15 >> 2);
                    const lds_half_t* v3 = (const lds_half_t*)(qb + kTileHalfs + krow * 64);
                    float4v o[3];
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt) {
                        const int vslot = 2 + 2 * dt + ((l15 & 3) >> 1);
                        const lds_half_t* va = v3 + (((vslot ^ (krow & 7)) << 3) + 4 * (l15 & 1));
                        const short4v tv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(va));
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v, tv), pf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    }
                    // O_h[query l15][d = 16 dt + 4 g + r] -> tile 0, row 16 wave + l15, column d (d >= 40: the zero rows of W_h made them 0);
                    // columns 48 .. 63 (still k_h) are zeroed: the tile is the K = 64 operand of the to_out steps
                    half_t* orow = qb + (16 * wave + l15) * 64;
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
                        *reinterpret_cast<half4v*>(orow + ((((2 * dt + (g >> 1)) ^ swz) << 3) | (4 * (g & 1)))) =
                            half4v{(half_t)o[dt][0], (half_t)o[dt][1], (half_t)o[dt][2], (half_t)o[dt][3]};
                    if (g < 2) *reinterpret_cast<half8v*>(orow + (((6 + g) ^ swz) << 3)) = half8v{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        } else {
            // ---- to_out, Wo tile t = slot - 5: acc2 += Wo tile rows (output columns, A operand) x O_h rows (B operand), K = 64 (40 used) ----
            constexpr int t = slot - kKT1;
            constexpr int TN = t < 2 ? 4 : 2;                 // the third tile holds output columns 256 .. 319 only
            const int wrow0 = (t < 2 ? 64 : 32) * wn + l15;   // this wave's first weight row inside the tile
            const half_t* gt_ = qb + arow;
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
    for (int head = 0; head < kHeads; ++head) {
        step(std::integral_constant<int, 0>{}, head); step(std::integral_constant<int, 1>{}, head);
        step(std::integral_constant<int, 2>{}, head); step(std::integral_constant<int, 3>{}, head);
        step(std::integral_constant<int, 4>{}, head); step(std::integral_constant<int, 5>{}, head);
        step(std::integral_constant<int, 6>{}, head); step(std::integral_constant<int, 7>{}, head);
    }

    // ---- epilogue: the fp32 tile goes through the idle ring in two halves of 64 rows so that the residual loads and the stores are
    // 16 bytes per lane on consecutive bytes of a row; the residual add is fp32, rounded once; frames >= T are not stored ----
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
        // 64 rows x 40 chunks of 8 columns; thread -> (frame, pixel, chunk) with the pixel next to the chunk: consecutive rows of a frame
        for (int idx = tid; idx < 64 * (kC / 8); idx += 512) {
            const int ch = idx % (kC / 8), rr = idx / (kC / 8);   // rr 0 .. 63 -> frame rr / 4, pixel 4 hrow + rr % 4
            const int t = rr >> 2, j = 4 * hrow + (rr & 3);
            if (t >= Tn) continue;
            const int row = 16 * (rr & 3) + t;                    // row inside this half's staging tile (local row 64 hrow + row)
            const long gr = grow(j, t);
            const float4v v0 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch);
            const float4v v1 = *reinterpret_cast<const float4v*>(stg + row * kSLd + 8 * ch + 4);
            const half8v r = *reinterpret_cast<const half8v*>(p.x + gr * p.ldx + 8 * ch);
            half8v o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (half_t)(v0[e] + (float)r[e]);
                o[4 + e] = (half_t)(v1[e] + (float)r[4 + e]);
            }
            *reinterpret_cast<half8v*>(p.out + gr * p.ldo + 8 * ch) = o;
        }
    }
}

}  // namespace

extern "C" int mv_temporal_attn_block_f16(const mv_tsa_desc* d, void* stream) {
    MV_REQUIRE(d != nullptr, "mv_temporal_attn_block_f16: null descriptor");
    MV_REQUIRE(d->x && d->wqkv && d->wo && d->ln_gamma && d->ln_beta && d->out, "mv_temporal_attn_block_f16: null pointer");
    MV_REQUIRE(d->C == kC && d->heads == kHeads && d->d == kD, "mv_temporal_attn_block_f16: built for C = %d = %d heads x %d (got C = %d, %d heads x %d): use the three-launch form",
               kC, kHeads, kD, d->C, d->heads, d->d);
    MV_REQUIRE(d->T >= 1 && d->T <= 16, "mv_temporal_attn_block_f16: T = %d not in [1, 16]", d->T);
    MV_REQUIRE(d->B >= 1 && d->HW >= kPix && d->HW % kPix == 0, "mv_temporal_attn_block_f16: HW = %d must be a positive multiple of %d", d->HW, kPix);
    MV_REQUIRE((long)d->B * d->T * d->HW < 0x7fffffffL && (long)d->B * (d->HW / kPix) < 0x7fffffffL, "mv_temporal_attn_block_f16: problem too large");
    MV_REQUIRE(d->ldx % 8 == 0 && d->ldo % 8 == 0 && d->ldx >= kC && d->ldo >= kC, "mv_temporal_attn_block_f16: leading dimensions must be multiples of 8 and >= C");
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    MV_REQUIRE(al16(d->x) && al16(d->wqkv) && al16(d->wo) && al16(d->out) && al16(d->ln_gamma) && al16(d->ln_beta) &&
               (!d->bias_o || (reinterpret_cast<uintptr_t>(d->bias_o) & 7) == 0),
               "mv_temporal_attn_block_f16: pointers must be 16-byte aligned (bias_o: 8)");
    MV_REQUIRE(d->ln_eps > 0.f && d->scale > 0.f, "mv_temporal_attn_block_f16: ln_eps and scale must be positive");
    TsaArgs a;
    a.x = (const half_t*)d->x; a.gamma = (const half_t*)d->ln_gamma; a.beta = (const half_t*)d->ln_beta;
    a.wqkv = (const half_t*)d->wqkv; a.wo = (const half_t*)d->wo; a.bias_o = (const half_t*)d->bias_o; a.out = (half_t*)d->out;
    a.B = (int)d->B; a.T = d->T; a.HW = d->HW; a.ldx = d->ldx; a.ldo = d->ldo; a.eps = d->ln_eps;
    a.scale_log2e = d->scale * 1.4426950408889634f;
    a.wqkv_bytes = (unsigned)((long)kHeads * 128 * kC * 2); a.wo_bytes = (unsigned)((long)kC * kWoLd * 2);
    a.rotate = (d->flags & 1) ? 1 : 0;
    constexpr int smem = kLdsHalfs * (int)sizeof(half_t);
    static_assert(smem <= 160 * 1024, "tsa tiles do not fit LDS");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tsa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        MV_REQUIRE(e == hipSuccess, "mv_temporal_attn_block_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const unsigned nblk = (unsigned)((long)d->B * (d->HW / kPix));
    hipLaunchKernelGGL(tsa_kernel, dim3(nblk), dim3(512), smem, (hipStream_t)stream, a);
    MV_CHECK_LAUNCH("mv_temporal_attn_block_f16");
    return MV_OK;
}
