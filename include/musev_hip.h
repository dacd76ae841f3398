/*
 * musev_hip.h -- C ABI of libmusev_hip.so, the MI355X (gfx950) kernel library behind
 * musev_amd.models.UNet3DConditionModel and musev_amd.pipelines (the MuseV parallel-denoise hot path).
 *
 * The reference (TMElyralab/MuseV) is 100 % Python and has no FFI boundary of its own: every entry point
 * below replaces a *torch / xformers / diffusers call site* on the hot path, cited per function as
 * "replaces: <reference file:line>".  Conventions:
 *   - extern "C", plain device pointers + explicit sizes/strides; no torch types, no ownership transfer,
 *     no hidden allocation (scratch is passed in by the caller), no global mutable state except the
 *     thread-local last-error string.
 *   - every launch goes to the hipStream_t passed as `stream` (void* so that C callers need no HIP headers).
 *   - return value: 0 = ok, negative = MV_ERR_*; mv_last_error() returns a human-readable message.
 *   - storage dtype: IEEE fp16 ("half") for activations and packed weights, fp32 accumulation and
 *     statistics.  Activation layout is channels-last: [B, T, H, W, C] (== rows x C matrices).
 */
#ifndef MUSEV_HIP_H
#define MUSEV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_OK 0
#define MV_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define MV_ERR_LAUNCH (-2)    /* HIP launch error */

#define MV_ABI_VERSION 13

/* ---- library ------------------------------------------------------------------------------------ */
int mv_abi_version(void);
const char* mv_last_error(void);

/* ---- implicit GEMM family (K2 conv3x3, K3 linear / 1x1 conv, K4 temporal conv) --------------------
 * out[m, n] = act( alpha * ( sum_k A(m, k) * W[n, k] + bias[n] + rowbias[m / rows_per_group, n] ) )
 *             + residual[m, n]
 * replaces: torch.nn.Linear / Conv2d(1x1) / Conv2d(3x3) / Conv3d((3,1,1)) calls inside
 *   diffusers ResnetBlock2D / Downsample2D / Upsample2D (constructed at musev/models/unet_3d_blocks.py:272-345,486-571),
 *   musev/models/resnet.py:57-82 (TemporalConvLayer conv1..4), musev/models/transformer_2d.py:257-271,365-389
 *   (proj_in / proj_out), diffusers Attention.to_q/to_k/to_v/to_out (musev/models/attention_processor.py:501-533),
 *   FeedForward (musev/models/attention.py:398-429), musev/models/temporal_transformer.py:243-251,276.
 */
#define MV_GEMM_LINEAR 0   /* A(m,k) = a[m*lda + k]  (k < c1) or a2[m*lda2 + k - c1]                   */
#define MV_GEMM_CONV3X3 1  /* A gathers a 3x3 neighbourhood of an NHWC image, zero padding 1           */
#define MV_GEMM_TCONV3 2   /* A gathers frames t-1, t, t+1 of a [B,T,HW,C] tensor, zero padding in t   */

#define MV_ACT_NONE 0
#define MV_ACT_SILU 1

typedef struct mv_gemm_desc {
    const void* a;         /* fp16 source 1                                                            */
    const void* a2;        /* fp16 source 2 (channel-concatenated after source 1) or NULL              */
    const void* w;         /* fp16 packed weight [N][K], K = taps * (c1 + c2), tap-major, channel-minor */
    void* c;               /* fp16 output [M][ldc]                                                     */
    const void* bias;      /* fp16 [N] or NULL                                                         */
    const void* rowbias;   /* fp16 [M / rows_per_group][ldrb] or NULL (time / frame embedding add)     */
    const void* residual;  /* fp16 [M][ldr] or NULL                                                    */
    const float* alpha;    /* device scalar; |*alpha| is used (temporal_weight); NULL -> 1.0           */
    int64_t M;
    int32_t N, K;
    int32_t lda, lda2, ldc, ldr, ldrb;  /* leading dimensions in elements                              */
    int32_t c1, c2;        /* channels of source 1 / 2 (c2 = 0 when a2 == NULL)                        */
    int32_t mode;          /* MV_GEMM_*                                                                */
    int32_t stride;        /* conv3x3: 1 or 2                                                          */
    int32_t upsample;      /* conv3x3: 1 = input is nearest-upsampled x2 on the fly                    */
    int32_t hin, win;      /* conv3x3: source image size (before upsample)                             */
    int32_t hout, wout;    /* conv3x3: output image size                                               */
    int32_t t, hw;         /* tconv3: frames per batch item, pixels per frame                          */
    int32_t rows_per_group;/* rowbias row = m / rows_per_group                                         */
    int32_t act;           /* MV_ACT_*                                                                 */
    int32_t geglu;         /* 1: w rows are [value | gate] interleaved per 16; out[m, n] for n < N/2 = */
                           /*    value * gelu(gate) (N is the full 2x width, ldc counts N/2 columns)   */
    int32_t cfg;           /* tile configuration: -1 = measured per-shape table, then rules (default); */
                           /*    -2 = rules only; >= 0 = this catalogue id wherever it applies          */
    int32_t splitk;        /* K slices: 0 = library's choice, >= 1 = this many (clamped, see below)     */
    void* workspace;       /* split-K scratch (fp32 slabs), 16-byte aligned; may be NULL when           */
    int64_t workspace_bytes; /*  mv_gemm_workspace_bytes(d) == 0                                        */
    /* LayerNorm folded into the projection (LINEAR mode; replaces nn.LayerNorm norm1 / norm2 / norm3 + the Linear behind  */
    /* it, musev/models/attention.py:293-308,345-362,398-429): `a` holds the RAW rows, `w` = W * gamma (column-wise), the  */
    /* kernel forms each row's mean / rstd over its K = C values from the fragments it multiplies and the epilogue applies */
    /*   out[m][n] = rstd_m * (acc[m][n] - mean_m * ln_colsum[n]) + ln_colbias[n]   (then GEGLU / residual as usual).      */
    const float* ln_colsum;  /* fp32 [N]: sum_k w[n][k] of the fp16 values in `w`; NULL = no folding                      */
    const float* ln_colbias; /* fp32 [N]: sum_k beta_k W[n][k] + bias_n (bias / rowbias must be NULL)                      */
    float ln_eps;            /* LayerNorm epsilon                                                                          */
    /* One weight matrix per GROUP of rows (ABI 12; LINEAR mode, no LayerNorm folding / carry): rows [g * w_group_rows,            */
    /* (g + 1) * w_group_rows) multiply the [N][K] matrix at w + g * N * K.  A GroupNorm without activation folded into the        */
    /* projection behind it (transformer_2d.py:260-271, temporal_transformer.py:239-247) has one scaled copy of the weights per    */
    /* normalised item: mv_groupnorm_cs_fold_linear_f16 writes them.  M must be a whole number of groups; the launcher picks a     */
    /* tile whose rows divide w_group_rows (a multiple of 32).  0 = one matrix.                                                     */
    int32_t w_group_rows;
    /* Statistics of the OUTPUT, formed by the epilogue from the fp16 values it stores, for the normalisation that reads the    */
    /* tensor next (16-byte epilogue, one K slice, no GEGLU; mv_gemm_stats_layout says whether a launch can and how big):      */
    /*   colstats[m / rows_per_tile][n] = {sum, sum of squares} over that row tile -> mv_groupnorm_cs_f16 (replaces the        */
    /*   statistics pass of nn.GroupNorm over the conv / proj_out outputs: resnet.py:57-82, transformer_2d.py:260, unet blocks) */
    float* colstats;         /* fp32 [ceil(M / rows_per_tile)][N][2], 8-byte aligned, or NULL                                   */
    int64_t colstats_floats; /* capacity of colstats in floats                                                                 */
    /* Two-fp16 carry of the residual stream's identity path (`hidden_states = hidden_states + residual` of ResnetBlock2D,         */
    /* resnet.py:133, transformer_2d.py:389, temporal_transformer.py:283-287; conv_in): with c_lo the epilogue forms               */
    /*   s = value + residual + residual_lo in fp32 and stores c = fp16(s), c_lo = fp16(s - c).  Layers read c (an ordinary fp16   */
    /* tensor); only the next residual add picks c_lo up again, so the identity path keeps ~22 bits instead of rounding to fp16 at */
    /* every block (16-byte epilogue, one K slice, no GEGLU / LayerNorm folding; leading dimensions ldc / ldr as c / residual).    */
    const void* residual_lo; /* fp16 [M][ldr] or NULL (a residual without a lo half)                                              */
    void* c_lo;              /* fp16 [M][ldc] or NULL (no carry)                                                                   */
    /* workgroup -> tile order (ABI 8).  0: m-major in groups of 8 m-tiles (an XCD's id range = a few m-tiles x all n-tiles).        */
    /* 1: ALLOW the weight-stationary order -- an XCD's id range = a few n-tiles x ALL m-tiles (and all K slices): each XCD streams */
    /*    only its share of the weight matrix -- taken where the fetch model (distinct A row blocks + distinct weight column       */
    /*    blocks per XCD) says the XCDs fetch at least 5 % less that way (mv_gemm_weight_stationary reports it): the small-M       */
    /*    levels.  Results are identical (every tile is still reduced over K in the same order by one block).                      */
    int32_t tile_order;
    /* second fp16 half of the row bias (ABI 12): out gets rowbias[g][n] + rowbias_lo[g][n] (same layout and leading dimension) --  */
    /* a bias formed in fp32 keeps ~22 bits as two fp16 rows (the folded GroupNorm's mean term); NULL = none                       */
    const void* rowbias_lo;
} mv_gemm_desc;

/* The library holds no tuning state: everything that selects a kernel travels in the descriptor.  A call with a split-K
 * choice (small-M, long-K problems) writes fp32 partial slabs [slices][M][N] into d->workspace and reduces them in fixed
 * slice order (bit-reproducible); no allocation happens inside the library. */
int mv_gemm_f16(const mv_gemm_desc* d, void* stream);
/* bytes of workspace mv_gemm_f16 needs for this descriptor (0 = none; -1 = invalid descriptor, see mv_last_error) */
int64_t mv_gemm_workspace_bytes(const mv_gemm_desc* d);
/* 1 if mv_gemm_f16 would run this descriptor in the weight-stationary workgroup order (tile_order = 1 and the fetch model prefers
 * it), 0 if not, -1 = invalid descriptor: introspection, launches nothing */
int mv_gemm_weight_stationary(const mv_gemm_desc* d);
/* the (tile configuration id, K slices) mv_gemm_f16 would use for this descriptor: introspection for tuners and tests */
int mv_gemm_choice(const mv_gemm_desc* d, int32_t* cfg, int32_t* nsplit);
/* output statistics this descriptor's launch can emit (d->colstats itself is ignored here): *col_rows_per_tile = rows per
 * colstats row tile (0 = the launch cannot emit them: split K, GEGLU or the narrow epilogue), *col_floats = floats to allocate */
int mv_gemm_stats_layout(const mv_gemm_desc* d, int32_t* col_rows_per_tile, int64_t* col_floats);

/* tile-configuration catalogue of the implicit-GEMM kernel (block tile, waves, K depth, LDS stages), for the per-shape
 * tuner (tools/gpu_gemm_tune.py -> musev_amd/csrc/gemm_tuned.h).  mv_gemm_config_desc fills {block rows, block columns,
 * waves, BK, LDS stages}. */
int mv_gemm_num_configs(void);
int mv_gemm_config_desc(int cfg, int32_t* desc5);

/* workgroup -> output-tile order of the implicit-GEMM kernel: logical ids (contiguous per XCD) walk groups of `group`
 * m-tiles m-fastest when the grid is more than `group` n-tiles wide, so that the ~64 blocks resident on one XCD cover
 * ~8 x 8 tiles (what its L2 must fetch per window) instead of 1-3 m-tiles x the whole weight matrix (the kernel uses
 * group 8; group -1 = the weight-stationary order of mv_gemm_desc.tile_order: n-major, a contiguous id range = a few
 * n-tiles x all m-tiles).  Results are independent of the order.
 * Host-side evaluation of that map (launches nothing): tile_m[b], tile_n[b] of workgroup b, for b < tiles_m*tiles_n */
int mv_gemm_tile_order(int tiles_m, int tiles_n, int group, int32_t* tile_m, int32_t* tile_n);

/* ---- feed-forward of a BasicTransformerBlock as one launch (K5 + K7 + K3) ------------------------------
 * replaces: norm3 -> ff (FeedForward(GEGLU) of diffusers) -> + hidden_states in BasicTransformerBlock.forward
 *   (musev/models/attention.py:398-429) for the 320-channel (level 0) stream:
 *     out = residual + ( GEGLU( LayerNorm(x) W1^T + b1 ) ) W2^T + b2
 * w1 / bias1: the GEGLU-packed first projection ([16 value rows | 16 gate rows] blocks, as mv_gemm_f16's geglu epilogue takes
 * it).  The normalised rows and the [M, 4 C] activation never leave the compute unit.  C = 320, hidden = 1280 only
 * (MV_ERR_INVALID otherwise: the caller keeps the three-launch form there). */
typedef struct mv_ffn_desc {
    const void* x;           /* fp16 [M][ldx], C columns: the rows norm3 reads                                  */
    const void* ln_gamma;    /* fp16 [C], 16-byte aligned                                                       */
    const void* ln_beta;     /* fp16 [C], 16-byte aligned                                                       */
    const void* w1;          /* fp16 [2 H][C] packed                                                            */
    const void* bias1;       /* fp16 [2 H] packed like w1's rows, or NULL                                       */
    const void* w2;          /* fp16 [C][H] (torch Linear layout)                                               */
    const void* bias2;       /* fp16 [C] or NULL                                                                */
    const void* residual;    /* fp16 [M][ldr] (the block passes x)                                              */
    void* out;               /* fp16 [M][ldo]                                                                   */
    int64_t M;
    int32_t C, H;            /* 320, 1280                                                                       */
    int32_t ldx, ldr, ldo;   /* leading dimensions in elements (multiples of 8)                                 */
    float ln_eps;
    int32_t flags;           /* bit 0: row blocks walk the hidden chunks from different starting chunks (spreads the weight reads) */
} mv_ffn_desc;
int mv_ffn_geglu_f16(const mv_ffn_desc* d, void* stream);

/* ---- one temporal self-attention sub-block as one launch (K5 + K3 + K6c + K3) ---------------------------
 * replaces: norm1 -> attn1 -> + hidden_states (and norm2 -> attn2 -> + hidden_states: double_self_attention) of the temporal
 *   BasicTransformerBlock (musev/models/attention.py:293-345) on the "(b h w) t c" sequences TransformerTemporalModel forms
 *   (musev/models/temporal_transformer.py:250-279), for the 320-channel (level 0) stream, 8 heads x 40, T <= 16 frames:
 *     out = x + to_out( softmax_T( q k^T scale ) v ) + bias_o,   [q | k | v] = LayerNorm(x) Wqkv^T   over the T frames of a pixel
 * Rows stay in (b, t, p) order (row = (b T + t) HW + p): a workgroup owns 8 pixels of one batch item with all their frames.
 * wqkv: per head 128 rows of C columns: to_q rows 40 h .. + 39, to_k rows 40 h .. + 39, to_v rows 40 h .. + 39, 8 zero rows.
 * wo: [C][heads * 64]: column 64 h + d holds to_out.0.weight[:, 40 h + d] for d < 40, zero for 40 <= d < 64.
 * The [M, 3 C] projection and the attention output never leave the compute unit.  MV_ERR_INVALID for any other geometry (the
 * caller keeps the three-launch form there). */
typedef struct mv_tsa_desc {
    const void* x;           /* fp16 [B T HW][ldx], C columns: the rows the LayerNorm reads, also the residual  */
    const void* ln_gamma;    /* fp16 [C], 16-byte aligned                                                       */
    const void* ln_beta;     /* fp16 [C], 16-byte aligned                                                       */
    const void* wqkv;        /* fp16 [heads][128][C] packed (see above)                                         */
    const void* wo;          /* fp16 [C][heads * 64] packed (see above)                                         */
    const void* bias_o;      /* fp16 [C] or NULL                                                                */
    void* out;               /* fp16 [B T HW][ldo]                                                              */
    int64_t B;
    int32_t T, HW;           /* frames (<= 16), pixels per frame (a multiple of 8)                              */
    int32_t C, heads, d;     /* 320, 8, 40                                                                      */
    int32_t ldx, ldo;        /* leading dimensions in elements (multiples of 8)                                 */
    float ln_eps, scale;     /* LayerNorm epsilon; softmax scale (d^-0.5)                                       */
    int32_t flags;           /* bit 0: workgroups walk the heads from different starting heads (spreads the weight reads) */
} mv_tsa_desc;
int mv_temporal_attn_block_f16(const mv_tsa_desc* d, void* stream);

/* ---- the text cross-attention sub-block of a spatial BasicTransformerBlock as ONE launch (ABI 12; level 0: C = 320 = 8 heads x 40) ----
 *   out = x + to_out( softmax( q K^T scale ) V ) + bias_o,   q = LayerNorm(x) Wq^T
 * replaces norm2 -> attn2.to_q -> attention over the prompt's keys / values -> attn2.to_out + residual of
 * musev/models/attention.py:345-396 / attention_processor.py:233-300 (one softmax group: the text tokens) where the three-launch form
 * (LayerNorm-folded projection, mv_attention_f16 with resident_kv, mv_gemm_f16 + residual) moves q and the attention output through
 * HBM.  k / v are the prompt's PROJECTED keys / values (constant over the denoise loop: the caller projects them once): rows
 * [kvb * len, (kvb + 1) * len) belong to key batch kvb = row / rows_per_kvb of x.
 * wq: [4][128][C] -- per head pair (2 p, 2 p + 1) the rows of to_q.weight [q_a (40 rows) | 24 zero rows | q_b (40) | 24 zero rows];
 * wo: [C][heads * 64] as for mv_tsa_desc.  MV_ERR_INVALID for any other geometry (more than 80 keys, several softmax groups, other
 * widths: the caller keeps the three-launch form there). */
typedef struct mv_xab_desc {
    const void* x;           /* fp16 [M][ldx], C columns: the rows the LayerNorm reads, also the residual       */
    const void* ln_gamma;    /* fp16 [C], 16-byte aligned                                                       */
    const void* ln_beta;     /* fp16 [C], 16-byte aligned                                                       */
    const void* wq;          /* fp16 [4][128][C] packed (see above)                                             */
    const void* k;           /* fp16 [key batches][len][ldk], 16-byte aligned                                   */
    const void* v;           /* fp16 [key batches][len][ldv]                                                    */
    const void* wo;          /* fp16 [C][heads * 64] packed                                                     */
    const void* bias_o;      /* fp16 [C] or NULL                                                                */
    void* out;               /* fp16 [M][ldo]                                                                   */
    int64_t M;
    int32_t rows_per_kvb;    /* rows of x per key batch (a multiple of 128)                                     */
    int32_t len;             /* keys per batch, 1 .. 80                                                         */
    int32_t C, heads, d;     /* 320, 8, 40                                                                      */
    int32_t ldk, ldv, ldx, ldo; /* leading dimensions in elements (multiples of 8, >= C)                        */
    float ln_eps, scale;     /* LayerNorm epsilon; softmax scale (d^-0.5), > 0                                  */
    int32_t flags;           /* bit 0: workgroups walk the head pairs from different starting pairs             */
} mv_xab_desc;
int mv_xattn_block_f16(const mv_xab_desc* d, void* stream);

/* ---- GroupNorm (K1) --------------------------------------------------------------------------------
 * replaces: torch.nn.GroupNorm(32, C)(+SiLU) in ResnetBlock2D.norm1/norm2, Transformer2DModel.norm
 *   (musev/models/transformer_2d.py:260), TransformerTemporalModel.norm (temporal_transformer.py:117,239),
 *   TemporalConvLayer conv*[0] (resnet.py:57-75; statistics span T*H*W: pass rows = T*H*W, groups_n = B),
 *   conv_norm_out (unet_3d_condition.py:562-568).
 * x: [n_items][rows][c1] (+ optional second source [n_items][rows][c2], channel-concatenated),
 * y: [n_items][rows][c1+c2].  Three launches: statistics (per-split group partials), fold, apply.
 * partial: fp32 scratch [n_items][nsplit][num_groups][2]; stat: fp32 scratch [n_items][num_groups][2] (mean, rstd).
 * gamma / beta 16-byte aligned.
 * x1_lo / y_lo (optional, single source, the three-launch form): the lo halves of a two-fp16 carry (see mv_gemm_desc.c_lo) -- the
 *   apply pass normalises hi + lo in fp32 and writes y = fp16(v), y_lo = fp16(v - y) (leading dimensions ld1 / ldy): conv_norm_out
 *   reads the carried residual stream and hands the output convolution ~22 bits (unet_3d_condition.py:1258-1263).
 */
int mv_groupnorm_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                     int64_t n_items, int64_t rows, int32_t num_groups, float eps,
                     const void* gamma, const void* beta, int32_t silu,
                     void* y, int32_t ldy, float* partial, int32_t nsplit, float* stat, const void* x1_lo, void* y_lo, void* stream);
/* the same with the statistics folded from producer-side column statistics (mv_gemm_desc.colstats) instead of a pass over x:
 * cs1 = colstats of the launch that wrote x1 (its N == c1, row tiles of rpt1 rows, rows %% rpt1 == 0), cs2 / rpt2 those of
 * x2 (required when x2 != NULL).  Two launches: fold (one block per (item, group)), apply.  Slabs small enough for the
 * one-launch kernel of mv_groupnorm_f16 still take it (the column statistics are then unused). */
int mv_groupnorm_cs_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                        int64_t n_items, int64_t rows, int32_t num_groups, float eps,
                        const void* gamma, const void* beta, int32_t silu, void* y, int32_t ldy,
                        const float* cs1, int32_t rpt1, const float* cs2, int32_t rpt2, int32_t nsplit, float* stat,
                        const void* x1_lo, void* y_lo, void* stream);
/* GroupNorm WITHOUT activation folded into the Linear / 1x1 convolution that reads it (ABI 12; replaces the apply pass -- an HBM
 * round trip of the activation -- of Transformer2DModel.norm -> proj_in, musev/models/transformer_2d.py:260-271,365-368, and of
 * TransformerTemporalModel.norm -> proj_in, temporal_transformer.py:239-247):
 *     proj(GN(x))[m][n] = sum_c (W[n][c] gamma_c rstd_{i,g(c)}) x[m][c]  +  b_n + sum_c W[n][c] beta_c - sum_c W'[n][c] mean_{i,g(c)}
 * for row m of item i.  From the producer's column statistics of x (cs / rpt as in mv_groupnorm_cs_f16, single source) this call
 * folds the group statistics (stat: fp32 scratch [n_items][num_groups][2]) and writes, per item, the scaled weights
 * w_out[i][n][c] = fp16(W gamma rstd) and the bias as two fp16 halves rb_hi / rb_lo[i * rb_per_item + j][n] (fp32 sum, the mean term
 * formed from the ROUNDED w_out so that it cancels what the product accumulates; rb_in[i * rb_per_item + j][n], ld ldrb_in, is added
 * when given: the frame-embedding projection of the temporal transformer, rb_per_item = frames per item; else rb_per_item = 1).
 * The projection is then mv_gemm_f16 on the RAW x with w = w_out, w_group_rows = rows, rowbias = rb_hi, rowbias_lo = rb_lo,
 * rows_per_group = rows / rb_per_item, bias = NULL.  Two launches (fold of the statistics, weights); c <= 2048, c % 8 == 0. */
int mv_groupnorm_cs_fold_linear_f16(const float* cs, int32_t rpt, int32_t c, int64_t n_items, int64_t rows, int32_t num_groups, float eps,
                                    const void* gamma, const void* beta, const void* w, const void* bias, int32_t n_out,
                                    const void* rb_in, int32_t ldrb_in, int32_t rb_per_item,
                                    void* w_out, void* rb_hi, void* rb_lo, float* stat, void* stream);
/* scratch size (in floats) of `partial` for the call above */
int64_t mv_groupnorm_partial_floats(int64_t n_items, int32_t num_groups, int32_t nsplit);
int32_t mv_groupnorm_default_nsplit(int64_t n_items, int64_t rows, int32_t c);

/* ---- LayerNorm over the channel dim (K5) -------------------------------------------------------------
 * replaces: nn.LayerNorm norm1/norm2/norm3 of BasicTransformerBlock (musev/models/attention.py:186,345,399). */
int mv_layernorm_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t c,
                     const void* gamma, const void* beta, float eps, void* stream);

/* ---- fused softmax attention (K6a/b/d) ---------------------------------------------------------------
 * out[n, q, h*d:(h+1)*d] (=, or += out_scale *)  softmax_j( scale * <Q[n,q,h], K_seg[.., j, h]> ) V_seg[.., j, h]
 * over the concatenation of up to MV_ATTN_MAX_SEG key/value segments (no concat copy is made).
 * Segment s serves query batch n from key/value batch  kvb = (n / div) * mul + add.
 * replaces: xformers.ops.memory_efficient_attention at musev/models/attention_processor.py:258 (text
 *   cross-attn), :292 (IP-Adapter second attention -> accumulate=1,out_scale=ip_adapter_scale), :519
 *   (reference-only self-attn: segments = [self frame, vision-condition frame(s), (referencenet tokens)]),
 *   :724 (ReferEmbFuseAttention: segments = [referencenet tokens, self frame]).
 */
#define MV_ATTN_MAX_SEG 4
typedef struct mv_attn_seg {
    const void* k;     /* fp16, row (kvb * len + j) at k + row*ldk, head h at column h*d                */
    const void* v;
    int32_t ldk, ldv;
    int32_t len;       /* keys per batch item                                                          */
    int32_t div, mul, add;
    /* softmax groups (d = 40 / 80): a segment with new_group = 1 starts its own softmax; the output is                */
    /*   sum_g group_scale_g * softmax_g(Q K_g^T) V_g  -- the text cross-attention + ip_adapter_scale * image-prompt  */
    /*   attention (+ FaceID) of attention_processor.py:258-300 in ONE launch.  All zero = one group of weight 1.      */
    int32_t new_group; /* segment 0: 1 = group_scale applies to the first group (else 1.0)                      */
    float group_scale;
} mv_attn_seg;

typedef struct mv_attn_desc {
    const void* q;     /* fp16 [nb][lq][ldq]                                                           */
    void* out;         /* fp16 [nb][lq][ldo]                                                           */
    int32_t ldq, ldo;
    int32_t nb, lq, heads, d;
    float scale;
    int32_t nseg;
    mv_attn_seg seg[MV_ATTN_MAX_SEG];
    int32_t accumulate; /* 0: out = attn ; 1: out += out_scale * attn                                   */
    float out_scale;
    /* 1: the resident-K/V kernel (ABI 8) -- a block owns whole query rows (wave = head), every head's keys / values stay in     */
    /*   registers for the block's lifetime: the text cross-attention (77 keys + image-prompt tokens as further softmax groups),  */
    /*   attention_processor.py:258-300.  Needs mv_attention_resident_ok(d) == 1: d in {40, 80}, heads <= 8, at most 8 key      */
    /*   tiles of 16 over all segments, <= 3 groups, accumulate == 0, heads*d-wide LDS image of V under 160 KB.  0: the tiled    */
    /*   kernels (any key count).  >= 16: the same with this many query rows per block (a multiple of 16) instead of the launcher's   */
    /*   choice (whole rounds of the CUs).                                                                                       */
    int32_t resident_kv;
} mv_attn_desc;

int mv_attention_f16(const mv_attn_desc* d, void* stream);
/* 1 if the problem fits the resident-K/V kernel (host-side check, launches nothing) */
int mv_attention_resident_ok(const mv_attn_desc* d);

/* ---- temporal self-attention over T <= 32 frames per pixel (K6c) -------------------------------------
 * rows are ordered (b, t, p): sequence of pixel (b, p) = rows (b*T + t)*HW + p, t = 0..T-1.
 * replaces: F.scaled_dot_product_attention through diffusers AttnProcessor2_0 in the temporal
 *   BasicTransformerBlock (attn1 and attn2, double_self_attention) musev/models/temporal_transformer.py:157-177,
 *   musev/models/attention.py:293-308,354-366 -- without the (b t) c h w <-> (b h w) t c permute copies
 *   of temporal_transformer.py:234-241,277-279. */
int mv_temporal_attention_f16(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk, int32_t ldv,
                              void* out, int32_t ldo, int32_t b, int32_t t, int32_t hw, int32_t heads,
                              int32_t d, float scale, void* stream);

/* ---- GEGLU gate (K7): y[m, j] = x[m, j] * gelu(x[m, half + j]) -----------------------------------------
 * replaces: diffusers GEGLU inside FeedForward (musev/models/attention.py:398-429). */
int mv_geglu_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t half_cols, void* stream);

/* ---- small-channel 3x3 convolutions (conv_in: Cin = 4, conv_out: Cout = 4) ----------------------------
 * replaces: self.conv_in / self.conv_out (musev/models/unet_3d_condition.py:334-339,570-575,1009,1262). */
int mv_conv3x3_cin_small_f16(const void* x, int32_t cin, const void* w /* [cout][3][3][cin] */, const void* bias,
                             const void* add /* optional [rows][cout] (pose_guider_emb) */, void* y,
                             int32_t cout, int64_t n_img, int32_t h, int32_t w_, void* stream);
/* im2col of a 3x3 / pad-1 convolution with tiny Cin: y[pix][tap*cin + ci], zero-filled up to kpad columns (kpad % 8 == 0);
 * conv_in then is mv_gemm_f16(LINEAR) with the weight rows zero-padded to kpad.                                          */
int mv_im2col3x3_f16(const void* x, int32_t cin, void* y, int32_t kpad, int64_t n_img, int32_t h, int32_t w_, void* stream);
/* y: [rows][cout] fp16, or fp32 when y_is_f32 (the UNet's noise prediction leaves the network unrounded); x_lo (optional): the lo
 * half of a two-fp16 input (the convolution reads x + x_lo in fp32) */
int mv_conv3x3_cout_small_f16(const void* x, const void* x_lo, int32_t cin, const void* w /* [cout][3][3][cin] */, const void* bias,
                              void* y, int32_t y_is_f32, int32_t cout, int64_t n_img, int32_t h, int32_t w_, void* stream);

/* direct 3x3 convolution, padding 1, stride 1 | 2, fused bias (+ SiLU when act = MV_ACT_SILU), any cin <= 455 (the 8 x 9 cin fp16 weight slab of a block must fit 64 KB of LDS), cout % 8 == 0:
 * x [n_img*h*w][cin] -> y [n_img*ho*wo][cout], w packed [cout][3][3][cin] (mv_pack_conv_weight_f16).
 * replaces: InflatedConv3d + F.silu of PoseGuider.forward (musev/models/controlnet.py:308-316,363-373) -- the conv stack
 *   that turns pose images into pose_guider_emb, once per call (pipeline_controlnet.py:1774-1781). */
int mv_conv3x3_direct_f16(const void* x, int32_t cin, const void* w /* [cout][3][3][cin] */, const void* bias, void* y,
                          int32_t cout, int64_t n_img, int32_t h, int32_t w_, int32_t stride, int32_t act, void* stream);

/* ---- elementwise helpers -------------------------------------------------------------------------------*/
/* sinusoidal Timesteps(dim, flip_sin_to_cos=True, shift=0): out[i, :] = [cos(t_i f), sin(t_i f)]           */
/* replaces: diffusers Timesteps in self.time_proj / self.frame_proj (unet_3d_condition.py:343,354,888,918) */
int mv_timestep_embedding_f16(const float* t, int32_t n, int32_t dim, void* out, void* stream);
int mv_silu_f16(const void* x, void* y, int64_t n, void* stream);
/* y = a + b (fp16), used for ControlNet residual adds (unet_3d_condition.py:1146-1156,1195)               */
/* y = a + b.  With a_lo / y_lo (may be NULL): a is a carried residual-stream tensor (mv_gemm_desc.c_lo) -- the sum (a + a_lo) + b is
 * formed in fp32 and stored as two fp16 halves again (the ControlNet residuals added to the UNet's skips, pipeline_controlnet.py:
 * 2045-2067 / unet_3d_condition.py:1160-1175, join the unrounded stream instead of rounding it a second time). */
int mv_add_f16(const void* a, const void* a_lo, const void* b, void* y, void* y_lo, int64_t n, void* stream);
/* nearest-neighbour resize of channels-last fp16 images [n_img, hin, win, c] -> [n_img, hout, wout, c] to an EXPLICIT size: torch's rule
 * src = min(floor(dst * (float)in / out), in - 1).  replaces: diffusers Upsample2D.forward(hidden_states, output_size) =
 * F.interpolate(size=output_size, mode="nearest"), reached through upsampler(hidden_states, upsample_size) (unet_3d_blocks.py:1235,1397)
 * when the latent size is not a multiple of 2^(number of upsamplers) (forward_upsample_size, unet_3d_condition.py:841-849,1209-1210).
 * The exact x 2 case never comes here: it is fused into mv_gemm_f16(CONV3X3, upsample = 1). */
int mv_upsample_nearest_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t n_img, int32_t hin, int32_t win, int32_t hout,
                            int32_t wout, int32_t c, void* stream);
/* rows of x selected by zeroing: y[g, :] = 0 for group rows flagged in mask (temb zeroing of the          */
/* vision-condition frames, unet_3d_condition.py:898-906)                                                   */
int mv_zero_rows_f16(void* x, int32_t ld, const int32_t* row_idx, int32_t n_idx, int32_t cols, void* stream);
/* layout: [B, C, T, H, W] (fp16 or fp32) -> channels-last fp16 [B, T, H, W, C] and back                    */
/* replaces: rearrange(sample, "b c t h w -> (b t) c h w") and its inverse (unet_3d_condition.py:1008,1263) */
int mv_bcthw_to_bthwc_f16(const void* x, int32_t x_is_f32, void* y, int32_t b, int32_t c, int32_t t, int32_t hw, void* stream);
int mv_bthwc_to_bcthw_f16(const void* x, int32_t x_is_f32, void* y, int32_t y_is_f32, int32_t b, int32_t c, int32_t t, int32_t hw,
                          void* stream);

/* ---- sliding-window denoise loop glue (K12) -----------------------------------------------------------
 * mv_window_gather: builds the UNet input of one window, channels-last fp16 [2?][n_cond + win][HW][C]:
 *   cond frames first (from cond_latents), then latents[:, :, idx[k]] (both CFG halves get the same data).
 * replaces: musev/pipelines/pipeline_controlnet.py:1902-1946 (gather, CFG repeat, scale_model_input (identity
 *   for DDIM), batch_concat_two_tensor_with_index).
 * latents: fp32 [C][T_total][HW] (batch 1), cond: fp32 [C][n_cond][HW].
 * hi_lo = 1: rows of 2 C columns [fp16(v) | fp16(v - fp16(v))]: the fp32 latents as two fp16 halves -- conv_in with its weight
 *   duplicated over the two channel groups then convolves the unrounded input (conv is linear).
 * cond_slot (ABI 11; int32 [n_cond] on the device, or NULL = condition frame k in slot k): the window slot of every condition frame
 *   = the reference's `vision_condition_latent_index` (prepare_condition_latents_and_index, pipeline_controlnet.py:966-1040; -1 ->
 *   the last of the n_cond + video_length frames, :995-1003).  As in batch_concat_two_tensor_with_index (data_util.py:242-268, call
 *   site :1939-1946) the condition frames are written into a zero tensor first and the window's frames at n_cond.. afterwards: a
 *   slot < n_cond no condition frame names stays ZERO, a condition frame whose slot is >= n_cond is overwritten by the window's
 *   frame there.  The caller checks 0 <= cond_slot[k] < n_cond + win (torch raises IndexError otherwise).
 */
int mv_window_gather(const float* latents, const float* cond, const int32_t* idx, const int32_t* cond_slot, int32_t win,
                     int32_t n_cond, int32_t c, int32_t t_total, int32_t hw, int32_t cfg_copies, int32_t hi_lo, void* out,
                     void* stream);
/* mv_window_scatter_add: eps_acc[half][C][T_total][HW] += eps_win (channels-last fp16|fp32 [halves][n_cond+win][HW][C],
 *   cond frames dropped); counter[T_total] += 1.
 * replaces: pipeline_controlnet.py:2068-2078. */
int mv_window_scatter_add(const void* eps_win, int32_t eps_is_f32, const int32_t* idx, int32_t win, int32_t n_cond, int32_t c,
                          int32_t t_total, int32_t hw, int32_t halves, int32_t half_offset,
                          float* eps_acc, float* counter, int32_t add_counter, void* stream);
/* ---- row softmax (side model: VAE decoder mid-block attention, single head of d = 512, run as GEMM -> softmax -> GEMM) ------
 * replaces: the softmax inside diffusers' Attention processor of AutoencoderKL.decoder.mid_block.attentions[0] (un-vendored);
 * call site in the reference: StableDiffusionPipeline.decode_latents via musev/pipelines/pipeline_controlnet.py:233-238.
 * x: fp16 [rows][ldx], softmax over the first `cols` entries of every row, in place (fp32 arithmetic). */
int mv_softmax_rows_f16(void* x, int64_t ldx, int64_t rows, int32_t cols, void* stream);

/* mv_window_units_reduce: the multi-rank form of the accumulation (reference :2076-2078 after the RCCL all-gather of SURVEY 8e):
 * eps_acc[half][C][T_total][HW] = sum, in table order, of units[slot][j*HW + p][C] (fp32 channels-last rows of one window's
 * generated frames; slot = rank * max_units + k in the gathered buffer, unit_stride elements apart).  table: int32
 * [halves][T_total][maxc][2] = (slot, j) pairs covering (half, frame), -1 terminated.  Overwrites eps_acc (no zero fill needed);
 * the order of the sum is the table's, identical on every rank. */
int mv_window_units_reduce(const float* units, int64_t unit_stride, const int32_t* table, int32_t maxc, int32_t c,
                           int32_t t_total, int32_t hw, int32_t halves, float* eps_acc, void* stream);
/* mv_cfg_ddim_step: eps = acc / counter; eps = eps_u + g (eps_t - eps_u); DDIM (eta = 0, epsilon prediction):
 *   x0 = (x - sqrt(1-a_t) eps) / sqrt(a_t);  x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps   (in place on latents)
 * replaces: pipeline_controlnet.py:2079,2101-2117 + musev/schedulers/scheduling_ddim.py:198-264. */
int mv_cfg_ddim_step(float* latents, const float* eps_acc, const float* counter, int32_t c, int32_t t_total,
                     int32_t hw, int32_t halves, float guidance, float alpha_t, float alpha_prev, void* stream);

/* mv_cfg_affine_step: eps = acc / counter; CFG; x <- cx * x + ce * eps (in place).  Euler-discrete with s_churn = 0:
 *   cx = 1, ce = sigma_{i+1} - sigma_i.
 * replaces: pipeline_controlnet.py:2079,2101-2117 + musev/schedulers/scheduling_euler_discrete.py:110-167
 *   (pred_original_sample = x - sigma*eps; derivative = (x - x0)/sigma = eps; prev = x + derivative*(sigma_next - sigma)). */
int mv_cfg_affine_step(float* latents, const float* eps_acc, const float* counter, int32_t c, int32_t t_total, int32_t hw,
                       int32_t halves, float guidance, float cx, float ce, void* stream);

/* ---- weight packing -------------------------------------------------------------------------------------
 * conv weight [O][I][kh][kw] (torch layout, fp16 or fp32) -> [O][kh][kw][I] fp16;  Conv3d [O][I][3][1][1] -> [O][3][I]
 * replaces: nothing in the reference (torch consumes its native layout); lets real checkpoints (state_dict keys
 * of SURVEY.md 8b) feed the kernels above. */
int mv_pack_conv_weight_f16(const void* w, int32_t w_is_f32, void* out, int32_t o, int32_t i, int32_t taps, void* stream);

/* ---- instrumentation: hardware layout probes used by tests (not on the product path) ------------------ */
int mv_probe_tr16(const void* lds_image /* 1024 x int16 */, void* out /* 64 x 4 x int16 */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSEV_HIP_H */
