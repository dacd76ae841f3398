// attention.hip -- fused softmax attention for gfx950 (MFMA 16x16x32 f16, fp32 online softmax).
//
// mv_attention_f16: flash-style forward over up to 4 key/value SEGMENTS that share one softmax.  The
// reference concatenates [self tokens | vision-condition-frame tokens | referencenet tokens] into one K/V
// source before to_k/to_v (attention_processor.py:431-493); because to_k/to_v are row-wise linear, the
// projections of the shared tokens are identical for every frame, so here K/V are projected ONCE per source
// tensor and each query frame walks the segments in place (no concat copy, no repeated projection).
//
// Work decomposition: grid = (q tiles of 128, heads, frames); block = 4 waves; each wave owns 32 query rows
// (two 16-row MFMA tiles) and the block streams 64-row K/V tiles through LDS (register-prefetched).
//   S^T = K Q^T  (operands swapped so a lane holds ONE query column and 16 key scores -> the row max / row sum
//                 are in-lane plus two xor-shuffles over the 4 lane groups; no LDS round trip for P)
//   O^T = V^T P^T with the MFMA k-index permuted as kv = 32c + 16*(j/4) + 4g + (j%4): the exp'd scores feed the
//                 second MFMA straight from the accumulator registers, and V^T fragments come from the row-major
//                 V tile through ds_read_b64_tr_b16 (hardware 4x16 transpose read).
// Head dims 40 / 80 / 160 (SD-1.5: C/8): the QK^T contraction is zero-padded to 64 / 96 / 160, O^T uses 3 / 5 / 10
// d-tiles of 16.
//
// mv_temporal_attention_f16: the temporal transformer's sequences are 13 tokens long (12 frames + 1 condition
// frame), one sequence per pixel and head; rows stay in (b, t, p) order so no permute copy is needed.  T <= 16 at the
// UNet's head widths: one wave per (pixel, head), one 16x16 MFMA tile per product (tattn3_kernel).  Other shapes
// (T <= 32, any d % 8 == 0): one 16/32-lane group per item, lane = query frame, fp32 VALU dot products (tattn_kernel).
#include "common.h"

namespace {

struct SegArgs {
    const half_t* k;
    const half_t* v;
    int ldk, ldv, len, div, mul, add;
    int new_group;   // 1: this segment starts a new softmax group (its own normalisation); segment 0 always starts one
    float gscale;    // weight of the group this segment starts in the output sum (read for group starts only)
};

struct AttnArgs {
    const half_t* q;
    half_t* out;
    int ldq, ldo, nb, lq, heads;
    float scale_log2e;
    int nseg;
    SegArgs seg[MV_ATTN_MAX_SEG];
    int accumulate;
    float out_scale;
};

template <int D, int KPAD = 8, int VPAD = 8>  // KPAD / VPAD: halfs of padding per K- / V-tile row (see KRS, VRS)
struct AttnCfg {
    static constexpr int DP = ((D + 31) / 32) * 32;  // padded QK^T contraction length
    static constexpr int NC = DP / 32;               // 32-wide k chunks
    static constexpr int NDT = (D + 15) / 16;        // O^T d-tiles
    static constexpr int DCH = D / 8;                // 16-byte chunks per row of real data
    // K tile row stride (halfs).  DP + 8 (an odd number of 16-byte slots) is conflict-free if a ds_read_b128 is served in
    // passes of 16 CONSECUTIVE lanes; under the lane groups MI355X_MICROARCH.md measures ({0-3, 12-15, 20-27}, ...) it is
    // 2-way conflicted for the fragment pattern (row l%16, slot l/16) and DP + 16 is the conflict-free stride.  Both are
    // instantiated; which one the hardware prefers is an A/B (mv_set_attn_variant +32).
    static constexpr int KRS = DP + KPAD;
    // V tile row stride (halfs).  The transpose read (ds_read_b64_tr_b16: 8 bytes per lane, rows 4g + l15/4) is served 32
    // lanes at a time: with the 16-byte pad, rows 0 and 7 of a pass overlap in 4 banks (28 r mod 64 dwords); without it the
    // eight rows land on disjoint 8-bank ranges (24 r / 40 r mod 64 at d = 40 / 80).  A/B: mv_set_attn_variant +64.
    static constexpr int VRS = NDT * 16 + VPAD;
    static constexpr int KV = 64;                    // keys per tile
    static constexpr int QT = (D > 80) ? 1 : 2;      // 16-row query tiles per wave (register budget at d = 160)
    static constexpr int QB = 64 * QT;               // query rows per block
    static constexpr int CHUNKS = KV * DCH;          // 16-byte chunks per K (or V) tile
    static constexpr int PF = (2 * CHUNKS + 255) / 256;  // prefetch registers (uint4) per thread for K+V
    static constexpr int LDS_HALFS = KV * KRS + KV * VRS;
};


__device__ __attribute__((aligned(16))) uint4 g_attn_zero[4];

// uniform per-field selects instead of p.seg[s]: a dynamic index into the by-value kernel argument (or a struct
// copy of it) would be materialised in scratch memory
#define ATTN_SEG_FIELD(p, s, f) ((s) == 0 ? (p).seg[0].f : (s) == 1 ? (p).seg[1].f : (s) == 2 ? (p).seg[2].f : (p).seg[3].f)

// Per-thread description of the 16-byte chunks this thread stages for every K/V tile of the current segment.
template <int D>
struct AttnStage {
    long goff[AttnCfg<D>::PF];  // element offset of the chunk inside a tile (row * ld + ch * 8), per-segment
    int row[AttnCfg<D>::PF];    // tile row of the chunk (for the tail predicate); >= KV marks "no chunk"
    int loff[AttnCfg<D>::PF];   // LDS offset in halfs (K region first, V region after)
    bool isv[AttnCfg<D>::PF];
};

// Issue the global loads of the K/V tile starting at key `b0` into registers (no wait).  Rows past the segment
// end read the zero page, so masked keys carry finite (zero) K and V rows.
template <int D>
__device__ __forceinline__ void attn_prefetch(u32x4 (&pf)[AttnCfg<D>::PF], const AttnStage<D>& st, const half_t* kb,
                                              const half_t* vb, long ldk, long ldv, int b0, int len,
                                              const half_t* zero) {
    using C = AttnCfg<D>;
    const half_t* kt = kb + (long)b0 * ldk;  // uniform tile bases
    const half_t* vt = vb + (long)b0 * ldv;
#pragma unroll
    for (int i = 0; i < C::PF; ++i) {
        const bool ok = (b0 + st.row[i]) < len;
        const half_t* ptr = (st.isv[i] ? vt : kt) + st.goff[i];
        ptr = ok ? ptr : zero;
        pf[i] = *reinterpret_cast<const u32x4*>(ptr);
    }
}

template <int D>
__device__ __forceinline__ void attn_commit(const u32x4 (&pf)[AttnCfg<D>::PF], const AttnStage<D>& st, half_t* lds) {
    using C = AttnCfg<D>;
#pragma unroll
    for (int i = 0; i < C::PF; ++i) {
        if (st.row[i] < C::KV) *reinterpret_cast<u32x4*>(lds + st.loff[i]) = pf[i];
    }
}

// OPT = 3: OPT 1 with the K/V tile fetch through buffer descriptors (one per operand and segment): a thread's chunk offsets
// are fixed for the whole segment, the tile advance is the scalar offset, and rows past the segment end use an out-of-range
// offset that reads zero -- the per-tile 64-bit pointer selects of the register-staged prefetch (~25 VALU ops of a
// VALU-bound loop) disappear.  K and V chunks are fetched by separate instructions so that the descriptor is uniform.
// OPT = 1 (d = 40 only, where the softmax VALU work -- not the MFMAs -- bounds the kernel): the row sums come out of
// the P.V MFMA itself through a column of ones parked in the unused d-columns [40, 48) of the V tile, the running max
// uses 3-input maxima, and the O rescale is skipped (exactly: alpha == 1) while no row maximum of the wave moves.
template <int D, int OPT, int KPAD = 8, int VPAD = 8>
__global__ __launch_bounds__(256, (OPT >= 1 && D == 40) ? 4 : 2) void attn_kernel(const AttnArgs p) {
    using C = AttnCfg<D, KPAD, VPAD>;
    constexpr bool ONES = (OPT >= 1) && (C::NDT * 16 > D);
    __shared__ __attribute__((aligned(16))) half_t lds[C::LDS_HALFS];
    half_t* sK = lds;
    half_t* sV = lds + C::KV * C::KRS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int h = blockIdx.y;
    const int n = blockIdx.z;
    const int q0 = blockIdx.x * C::QB + wave * (16 * C::QT);

    // zero the K tile once: the padded contraction columns [D, DP) must read as 0 forever
    for (int i = tid; i < C::KV * C::KRS / 8; i += 256) reinterpret_cast<uint4*>(sK)[i] = uint4{0, 0, 0, 0};
    if constexpr (ONES) {  // V[kv][D] = 1 for every key row (the commits never touch columns >= D)
        if (tid < C::KV) sV[tid * C::VRS + D] = (half_t)1.0f;
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = l15][d = 32c + 8g .. +8] ----
    half8v qf[C::QT][C::NC];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int qr = q0 + 16 * qt + l15;
#pragma unroll
        for (int c = 0; c < C::NC; ++c) {
            const int dcol = 32 * c + 8 * g;
            half8v v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qr < p.lq && dcol < D)
                v = *reinterpret_cast<const half8v*>(p.q + ((long)n * p.lq + qr) * p.ldq + h * D + dcol);
            qf[qt][c] = v;
        }
    }

    float4v acc_o[C::QT][C::NDT];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] = float4v{0.f, 0.f, 0.f, 0.f};
    float m_run[C::QT], l_run[C::QT];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
    }

    // ---- walk the key/value segments; inside a segment the next tile is prefetched under the MFMAs ----
    const half_t* zero = reinterpret_cast<const half_t*>(g_attn_zero);
    u32x4 pf[C::PF];
    AttnStage<D> stg;
#pragma unroll
    for (int i = 0; i < C::PF; ++i) {
        const int idx = tid + 256 * i;  // [0, 2*CHUNKS): first the K chunks, then the V chunks
        const bool isv = idx >= C::CHUNKS;
        const int cidx = isv ? idx - C::CHUNKS : idx;
        const int row = cidx / C::DCH, ch = cidx - row * C::DCH;
        stg.isv[i] = isv;
        stg.row[i] = (idx < 2 * C::CHUNKS) ? row : (1 << 20);
        stg.loff[i] = isv ? C::KV * C::KRS + row * C::VRS + ch * 8 : row * C::KRS + ch * 8;
        stg.goff[i] = ch * 8;  // + row * ld, filled per segment
    }

    constexpr bool BUF = (OPT == 3);
    constexpr int PFH = (C::CHUNKS + 255) / 256;  // BUF: 16-byte chunks per thread and operand
    constexpr unsigned kOOBA = 0x80000000u;       // out of range for every descriptor below (num_records < 2 GiB)
    u32x4 pfk[PFH], pfv[PFH];
    // chunk (tid + 256 i) of a tile: row = chunk / DCH, 16-byte slot = chunk % DCH; only the LDS offsets are kept in
    // registers, row / slot are recomputed where a segment or a partial tile needs them (register budget: 128 at d = 40)
    int b_loffk[PFH], b_loffv[PFH];
#pragma unroll
    for (int i = 0; i < PFH; ++i) {
        const int cidx = tid + 256 * i;
        const int row = cidx / C::DCH, ch = cidx - row * C::DCH;
        b_loffk[i] = row * C::KRS + ch * 8;
        b_loffv[i] = C::KV * C::KRS + row * C::VRS + ch * 8;
    }

#pragma unroll 1
    for (int seg = 0; seg < p.nseg; ++seg) {
        const int len = ATTN_SEG_FIELD(p, seg, len);
        const long ldk = ATTN_SEG_FIELD(p, seg, ldk), ldv = ATTN_SEG_FIELD(p, seg, ldv);
        const int sdiv = ATTN_SEG_FIELD(p, seg, div), smul = ATTN_SEG_FIELD(p, seg, mul), sadd = ATTN_SEG_FIELD(p, seg, add);
        const long kvb = (long)(n / sdiv) * smul + sadd;
        const half_t* kb = ATTN_SEG_FIELD(p, seg, k) + kvb * len * ldk + h * D;
        const half_t* vb = ATTN_SEG_FIELD(p, seg, v) + kvb * len * ldv + h * D;
        AttnStage<D> st = stg;
        // BUF: per-segment byte offsets of this thread's chunks inside a tile; descriptors over this (frame, head)'s rows
        unsigned b_vk[PFH], b_vv[PFH];
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
        auto buf_prefetch = [&](int b0) __attribute__((always_inline)) {
            if (b0 >= len) return;  // uniform: nothing follows the last tile of a segment
            const int sk = b0 * (int)ldk * 2, sv = b0 * (int)ldv * 2;  // scalar byte offsets of the tile
            if (b0 + C::KV > len) {  // uniform: the partial last tile masks its rows past the end
#pragma unroll
                for (int i = 0; i < PFH; ++i) {
                    const bool ok = b0 + (tid + 256 * i) / C::DCH < len;  // inactive chunks already carry kOOBA
                    pfk[i] = __builtin_amdgcn_raw_buffer_load_b128(rK, (int)(ok ? b_vk[i] : kOOBA), sk, 0);
                    pfv[i] = __builtin_amdgcn_raw_buffer_load_b128(rV, (int)(ok ? b_vv[i] : kOOBA), sv, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < PFH; ++i) {
                    pfk[i] = __builtin_amdgcn_raw_buffer_load_b128(rK, (int)b_vk[i], sk, 0);
                    pfv[i] = __builtin_amdgcn_raw_buffer_load_b128(rV, (int)b_vv[i], sv, 0);
                }
            }
        };
        if constexpr (BUF) {
#pragma unroll
            for (int i = 0; i < PFH; ++i) {
                const int cidx = tid + 256 * i;
                const int row = cidx / C::DCH, ch = cidx - row * C::DCH;
                const bool act = cidx < C::CHUNKS;
                b_vk[i] = act ? (unsigned)((row * (int)ldk + ch * 8) * 2) : kOOBA;
                b_vv[i] = act ? (unsigned)((row * (int)ldv + ch * 8) * 2) : kOOBA;
            }
            buf_prefetch(0);
        } else {
#pragma unroll
            for (int i = 0; i < C::PF; ++i) {
                const int r = st.row[i] < C::KV ? st.row[i] : 0;
                st.goff[i] += (long)r * (st.isv[i] ? ldv : ldk);
            }
            attn_prefetch<D>(pf, st, kb, vb, ldk, ldv, 0, len, zero);
        }
      for (int cur_base = 0; cur_base < len; cur_base += C::KV) {
        __syncthreads();  // previous tile fully consumed (also orders the initial zero fill)
        if constexpr (BUF) {
#pragma unroll
            for (int i = 0; i < PFH; ++i) {
                if (tid + 256 * i < C::CHUNKS) {
                    *reinterpret_cast<u32x4*>(lds + b_loffk[i]) = pfk[i];
                    *reinterpret_cast<u32x4*>(lds + b_loffv[i]) = pfv[i];
                }
            }
        } else {
            attn_commit<D>(pf, st, lds);
        }
        __syncthreads();
        // prefetch the next tile of this segment while this one is computed (past the end: zero page / nothing)
        if constexpr (BUF) buf_prefetch(cur_base + C::KV);
        else attn_prefetch<D>(pf, st, kb, vb, ldk, ldv, cur_base + C::KV, len, zero);

        // ---- S^T = K Q^T : acc_s[qt][st][r] = S[q = l15][kv = 16 st + 4 g + r] ----
        float4v acc_s[C::QT][4];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
            for (int st = 0; st < 4; ++st) acc_s[qt][st] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C::NC; ++c) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                half8v kf = *reinterpret_cast<const half8v*>(sK + (16 * st + l15) * C::KRS + 32 * c + 8 * g);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_s[qt][st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][c], acc_s[qt][st], 0, 0, 0);
            }
        }
        // ---- mask the tail of the segment ----
        if (cur_base + C::KV > len) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (cur_base + 16 * st + 4 * g + r >= len) {
#pragma unroll
                        for (int qt = 0; qt < C::QT; ++qt) acc_s[qt][st][r] = -INFINITY;
                    }
                }
        }
        // ---- online softmax (per lane: one query column, 16 scores; lane groups g hold the other 48) ----
        half8v pfrag[C::QT][2];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            float mx;
            if constexpr (OPT >= 1) {
                // 16 scores -> 1 in a tree of 3-input maxima.  Written as fmaxf(fmaxf(a, b), c) throughout: that shape
                // selects v_max3_f32 on the raw MFMA results, whereas a 2-input fmaxf of two accumulator registers is
                // preceded by a canonicalising v_max_f32 x, x per operand (10 VALU ops instead of ~25; the maximum
                // itself is exact in any association, so the result does not change)
                const float4v &s0 = acc_s[qt][0], &s1 = acc_s[qt][1], &s2 = acc_s[qt][2], &s3 = acc_s[qt][3];
                const float t0 = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
                const float t1 = fmaxf(fmaxf(s0[3], s1[0]), s1[1]);
                const float t2 = fmaxf(fmaxf(s1[2], s1[3]), s2[0]);
                const float t3 = fmaxf(fmaxf(s2[1], s2[2]), s2[3]);
                const float t4 = fmaxf(fmaxf(s3[0], s3[1]), s3[2]);
                const float u0 = fmaxf(fmaxf(t0, t1), t2);
                const float u1 = fmaxf(fmaxf(t3, t4), s3[3]);
                mx = fmaxf(u0, u1);
            } else {
                mx = -INFINITY;
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc_s[qt][st][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx * p.scale_log2e);
            if constexpr (OPT >= 1) {
                if (__any(m_new != m_run[qt])) {  // otherwise alpha == exp2(0) == 1: skipping the rescale is bit-exact
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                    if constexpr (!ONES) l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] *= alpha;
                    m_run[qt] = m_new;
                }
            } else {
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                m_run[qt] = m_new;
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] *= alpha;
            }
            float ps = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = __builtin_amdgcn_exp2f(fmaf(acc_s[qt][st][r], p.scale_log2e, -m_new));
                    acc_s[qt][st][r] = e;
                    if constexpr (!ONES) ps += e;
                }
            if constexpr (!ONES) l_run[qt] += ps;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                if constexpr (OPT == 2) {
                    // one v_cvt_pkrtz_f16_f32 per two probabilities (round toward zero); with the ones-column row sums
                    // numerator and denominator see the SAME rounded values, so the bias cancels in O = sum(p v) / sum(p)
                    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                    union { fp16x2 h2[4]; half8v h8; } u;
                    u.h2[0] = __builtin_amdgcn_cvt_pkrtz(acc_s[qt][2 * cc][0], acc_s[qt][2 * cc][1]);
                    u.h2[1] = __builtin_amdgcn_cvt_pkrtz(acc_s[qt][2 * cc][2], acc_s[qt][2 * cc][3]);
                    u.h2[2] = __builtin_amdgcn_cvt_pkrtz(acc_s[qt][2 * cc + 1][0], acc_s[qt][2 * cc + 1][1]);
                    u.h2[3] = __builtin_amdgcn_cvt_pkrtz(acc_s[qt][2 * cc + 1][2], acc_s[qt][2 * cc + 1][3]);
                    pfrag[qt][cc] = u.h8;
                } else {
                    half8v f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = (half_t)acc_s[qt][2 * cc][r];
                        f[4 + r] = (half_t)acc_s[qt][2 * cc + 1][r];
                    }
                    pfrag[qt][cc] = f;
                }
            }
        }
        // ---- O^T += V^T P^T : A = V^T fragment via transpose read, B = P fragment ----
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                // 16-lane group g reads the [4 kv][16 d] blocks at kv = 32cc + 4g (+16), d = 16 dt
                const half_t* b0 = sV + (32 * cc + 4 * g + (l15 >> 2)) * C::VRS + 16 * dt + (l15 & 3) * 4;
                short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4v*)(b0));
                short4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) short4v*)(b0 + 16 * C::VRS));
                typedef short short8v __attribute__((ext_vector_type(8)));
                short8v tv = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                half8v vf = __builtin_bit_cast(half8v, tv);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfrag[qt][cc], acc_o[qt][dt], 0, 0, 0);
            }
        }
      }
    }

    // ---- epilogue: O^T[d = 16 dt + 4 g + r][q = l15] / l ----
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        float l;
        if constexpr (ONES) {
            // row D of O^T holds sum_k P[k][q] (the ones column of V): tile D / 16, lane group (D % 16) / 4, register D % 4
            l = __shfl(acc_o[qt][D / 16][D % 4], ((D % 16) / 4) * 16 + l15, 64);
        } else {
            l = l_run[qt];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int qr = q0 + 16 * qt + l15;
        if (qr >= p.lq) continue;
        half_t* orow = p.out + ((long)n * p.lq + qr) * p.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
            const int dcol = 16 * dt + 4 * g;
            if (dcol >= D) continue;
            float4v o = acc_o[qt][dt] * inv;
            if (p.accumulate) {
                half4v prev = *reinterpret_cast<const half4v*>(orow + dcol);
                o = float4v{(float)prev[0], (float)prev[1], (float)prev[2], (float)prev[3]} + o * p.out_scale;
            }
            half4v w = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            *reinterpret_cast<half4v*>(orow + dcol) = w;
        }
    }
}

// =====================================================================================================================
// attn3_kernel (d = 40 / 80): the same math and fragment layouts as attn_kernel, with the per-tile instruction count cut to a
// third.  attn_kernel<40, 1> issues ~530 instructions per 64-key tile and wave for 28 MFMAs (static count of its loop: ~290 of
// them are the register-staged K/V prefetch -- per-chunk 64-bit pointer selects, tail predicates, ds_write commits -- and 64
// are the scale / subtract fma + exp pairs) and runs at ~1 instruction per 4 cycles per SIMD: it is instruction-issue-bound
// (MFMA pipe ~30 % busy).  Here:
//   * K / V tiles travel global -> LDS by LDS-DMA through buffer descriptors (scalar tile advance, rows past the segment end
//     carry an out-of-range offset and read as zero): no staging registers, no ds_write, no per-tile address arithmetic; dense
//     rows of D halfs (a whole number of 1-KiB pieces per tile); a three-stage ring with counted waits keeps two tiles in
//     flight behind one raw s_barrier per tile.
//   * the score MFMA starts from the accumulator -m instead of 0 and the query fragment is pre-multiplied by scale * log2(e):
//     S' = q.k * scale * log2(e) - m leaves the matrix pipe ready for exp2 with NO per-score VALU work.
//   * m is a lazily updated reference, not the exact running maximum (softmax is invariant to the reference as long as
//     numerators and row sums use the same one): a tile rescales only when some score exceeds the reference by more than kThr
//     (P <= 2^kThr in fp16) or on the first tile; then O, the row sums and the tile's scores shift by the same amount.
//   * the row sums come out of the matrix pipe as well: one extra MFMA per 32 keys against a constant fragment whose row 0 is
//     all ones (fp32 accumulation of exactly the fp16-rounded probabilities the P.V MFMA sees).
// Contraction columns past D (the 32-deep chunks cover 64 / 96) read the next LDS row: finite data (the LDS is zero-filled
// once, DMA data is finite) times the zero tail of the query fragment.
// Per tile and wave at d = 40: 32 MFMA + 32 exp + 16 packed converts + 16 max3 + 1 compare + 20 LDS reads + 2-3 DMA issues.
// VAR (bit set; the dispatcher picks per head dim what the MI355X measured fastest, tools/gpu_attn_bench.py):
//   1  the contraction tail past the last whole 32-deep chunk (d 32..39 at d = 40, d 64..79 at d = 80) runs as ONE 16-deep
//      v_mfma_f32_16x16x16_f16 step instead of a zero-padded 32-deep one: 40 -> 48 instead of 64, 80 -> 80 instead of 96
//   2  (d = 40) K and V tile rows are 48 halfs apart: the sixth 16-byte slot of a row is never written by the LDS-DMA (its lanes
//      are switched off), stays 0 in K and holds {1, 0, ...} in V, so that
//        * the row sums come out of the P.V MFMA's unused output row d = 40 (no ones-fragment MFMAs: 28 instead of 32 per tile),
//        * the 96-byte row stride makes both the ds_read_b128 K fragments and the ds_read_b64_tr_b16 V reads conflict-free under
//          the hardware's lane groups (80-byte rows: 2-way on both).
//   4  (d = 40) register budget of four waves per SIMD (128 VGPRs) instead of three
//   8  eight waves = 256 query rows per block share every K / V tile (half the L2 -> LDS traffic and half the barriers per MFMA)
//  32  s_setprio 1 around the two matrix clusters of a tile (S^T = K Q^T, O^T += V^T P^T), 0 around the softmax: the SIMD's other waves'
//      exp / convert work yields issue slots to the wave that feeds the matrix pipe.  Round 5, same box (profiles/r05i_attn_variants.log):
//      26 frames 1.630 -> 1.578 ms, 13 frames 0.868 -> 0.781 ms; in the two-stream step -0.25 ms (r05j: 55.76 / 55.89 against 56.06 /
//      56.13 / 56.08).  A static priority for the second-dispatched half of the block instead measured 0.4 % (not kept).
//  16  softmax GROUPS: a segment flagged new_group closes the running softmax -- O / row sum, times the group's weight, is added to
//      a second accumulator -- and starts a fresh one: out = sum_g gscale_g * softmax_g(Q K_g^T) V_g in ONE launch (text
//      cross-attention + IP-Adapter (+ FaceID) image-prompt attention: Q read once, the output written once, no read-modify-write)
template <int D, int VAR>
struct Attn3Cfg {
    static constexpr bool GRP = (VAR & 16) != 0;
    static constexpr bool TAIL16 = (VAR & 1) != 0 && (D % 32) != 0 && (D % 32) <= 16;
    static constexpr bool PADR = (VAR & 2) != 0 && D == 40;
    static constexpr int CH = D / 8;                   // 16-byte chunks of data per row
    static constexpr int RCH = PADR ? CH + 1 : CH;     // 16-byte slots per LDS row
    static constexpr int DR = 8 * RCH;                 // LDS row stride (halfs)
    static constexpr int NC = (D + 31) / 32;           // 32-deep contraction chunks (the last one 16 deep with TAIL16)
    static constexpr int NC32 = TAIL16 ? NC - 1 : NC;  // ... of which whole 32-deep MFMA steps
    static constexpr int NDT = (D + 15) / 16;          // O^T d-tiles
    static constexpr bool ONES = PADR && NDT * 16 > D; // the row sums ride in output row D of the P.V product
    static constexpr int NW = (VAR & 8) ? 8 : 4;       // waves per block (each owns 32 query rows)
    static constexpr int KV = 64, QT = 2, QB = 32 * NW, NST = 3;
    static constexpr int TILE = KV * DR;               // halfs per K (or V) tile = RCH pieces of 1 KiB
    static constexpr int PIECES = 2 * RCH;             // LDS-DMA pieces per (K, V) tile pair, dealt round-robin to the waves
    static constexpr int PPW = (PIECES + NW - 1) / NW; // ... at most per wave
    static constexpr int LDS_HALFS = NST * 2 * TILE + 64;  // slack: a 32-deep fragment read of the last row runs past it
};

template <int D, int VAR>
__global__ __launch_bounds__((64 * Attn3Cfg<D, VAR>::NW), (D == 40 ? ((VAR & 4) ? 4 : 3) : 2))  // (threads, waves per SIMD)
void attn3_kernel(const AttnArgs p) {
    using C = Attn3Cfg<D, VAR>;
    constexpr int NT = 64 * C::NW;
    static_assert(D % 8 == 0 && (64 * C::RCH) % 64 == 0, "a tile must be a whole number of 1-KiB pieces");
    constexpr float kThr = 6.0f;  // scores may exceed the reference by 2^6: P <= 64 in fp16, sums in fp32
    constexpr int DR = C::DR;
    extern __shared__ __attribute__((aligned(16))) half_t lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware: the dispatcher deals workgroups round-robin to the 8 XCDs; the remap hands every XCD a CONTIGUOUS range
    // of logical ids = all query blocks of a few (frame, head) pairs, so the ~96 blocks resident on an XCD stream the K / V of ~3
    // pairs (1.3 MB each at level 0) through its 4 MB L2.  With the plain (q block, head, frame) grid every XCD held blocks of
    // ~24 pairs at once and re-fetched K / V from the fabric: removing the tile DMA sped the kernel up by 17 % (ablation, r02).
    const int qblocks = (p.lq + C::QB - 1) / C::QB;
    const int id = mv_xcd_remap(blockIdx.x, gridDim.x);
    // (frame, head) pairs HEAD-major: with 8 heads an XCD walks one head over all frames -- every XCD gets its share of the frames whose
    // condition segment is not walked (dup_seg below; frame-major, the first XCD held all of them and the launch ended with the others),
    // and the condition frame's K / V of that head (shared by all frames of a batch item) stay in its L2
    const int hn = id / qblocks;
    const int h = hn / p.nb;
    const int n = hn - h * p.nb;
    const int q0 = (id - hn * qblocks) * C::QB + wave * (16 * C::QT);
    const int npw = (C::PIECES - wave + C::NW - 1) / C::NW;  // pieces this wave issues per tile pair (wave-uniform)

    for (int i = tid; i < C::LDS_HALFS / 8; i += NT) reinterpret_cast<uint4*>(lds)[i] = uint4{0, 0, 0, 0};
    if constexpr (C::ONES) {
        __syncthreads();
        // the sixth slot of every V row: {1, 0, 0, 0, 0, 0, 0, 0} -- written once, the LDS-DMA never touches it
        for (int i = tid; i < C::NST * C::KV; i += NT)
            lds[(i / C::KV) * (2 * C::TILE) + C::TILE + (i % C::KV) * DR + D] = (half_t)1.0f;
    }
    typedef __attribute__((address_space(3))) half_t lds_half_t;
    const lds_half_t* lds3 = (const lds_half_t*)lds;                           // the LDS image in its own address space (one cast)
    const int v_lane_off = (4 * g + (l15 >> 2)) * DR + (l15 & 3) * 4;          // this lane's place inside a [4 kv][16 d] block of V

    // ---- Q fragments (B operand of S^T = K Q^T), pre-multiplied by scale * log2(e); zero past column D ----
    half8v qf[C::QT][C::NC32];
    half4v qt16[C::QT];  // TAIL16: the 16-deep tail d = 32 NC32 + 4 g .. + 3
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int qr = q0 + 16 * qt + l15;
        const half_t* qrow = p.q + ((long)n * p.lq + (qr < p.lq ? qr : 0)) * p.ldq + h * D;
#pragma unroll
        for (int c = 0; c < C::NC32; ++c) {
            const int dcol = 32 * c + 8 * g;
            half8v v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qr < p.lq && dcol < D) {
                const half8v raw = *reinterpret_cast<const half8v*>(qrow + dcol);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)raw[e] * p.scale_log2e);
            }
            qf[qt][c] = v;
        }
        qt16[qt] = half4v{0, 0, 0, 0};
        if constexpr (C::TAIL16) {
            const int dcol = 32 * C::NC32 + 4 * g;
            if (qr < p.lq && dcol < D) {
                const half4v raw = *reinterpret_cast<const half4v*>(qrow + dcol);
#pragma unroll
                for (int e = 0; e < 4; ++e) qt16[qt][e] = (half_t)((float)raw[e] * p.scale_log2e);
            }
        }
    }
    // constant A fragment of the row-sum MFMA: row 0 (lanes with l15 == 0) all ones
    half8v ones_f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones_f[e] = (half_t)(l15 == 0 ? 1.0f : 0.0f);

    float4v acc_o[C::QT][C::NDT], acc_l[C::QT];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        acc_l[qt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] = float4v{0.f, 0.f, 0.f, 0.f};
    }
    float m_ref[C::QT] = {0.f, 0.f};  // reference of the exponents (log2 domain)
    // GRP: the finished groups' weighted outputs; a group is closed by dividing its O by its row sum (see the epilogue)
    float4v acc_t[C::GRP ? C::QT : 1][C::GRP ? C::NDT : 1];
    float gscale = p.seg[0].gscale;
    if constexpr (C::GRP) {
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) acc_t[qt][dt] = float4v{0.f, 0.f, 0.f, 0.f};
    }
    auto row_sum = [&](int qt) __attribute__((always_inline)) -> float {
        // row 0 of the ones-fragment product (lane group g = 0, register 0), or output row D of P.V (ONES: d-tile D / 16, row
        // D % 16 = 4 g' + r' -> lane group g', register r')
        if constexpr (C::ONES) return __shfl(acc_o[qt][D / 16][(D % 16) % 4], 16 * ((D % 16) / 4) + l15, 64);
        else return __shfl(acc_l[qt][0], l15, 64);
    };

    __syncthreads();  // the zero fill (and the ones column) is complete before the first DMA may land
    bool first = true;    // no tile processed yet (the first tile sets the reference unconditionally)
    // A later segment of the (single) softmax that addresses the very rows of segment 0 is not walked.  Reference-only self-attention
    // appends the vision-condition frame's keys / values to every frame's own (attention_processor.py:431-468): for the condition
    // frame itself that is its own key set a second time.  A key set counted twice has twice the softmax weight, i.e.
    // exp2(score + 1): segment 0 runs with its reference lowered by one instead (exact: a power of two on numerators and row sums
    // alike, relative to the other segments), and the block saves the duplicate's tiles (1 frame in 13 at config 2: half its work).
    int dup_seg = -1;
    if constexpr (!C::GRP) {
        const long kvb0 = (long)(n / p.seg[0].div) * p.seg[0].mul + p.seg[0].add;
#pragma unroll
        for (int s2 = 1; s2 < MV_ATTN_MAX_SEG; ++s2) {
            if (s2 < p.nseg && dup_seg < 0 && p.seg[s2].k == p.seg[0].k && p.seg[s2].v == p.seg[0].v && p.seg[s2].len == p.seg[0].len &&
                p.seg[s2].ldk == p.seg[0].ldk && p.seg[s2].ldv == p.seg[0].ldv &&
                (long)(n / p.seg[s2].div) * p.seg[s2].mul + p.seg[s2].add == kvb0)
                dup_seg = s2;
        }
    }
    // ---- walk the key/value segments; per segment a three-stage LDS ring with two tiles in flight (the ring drains at a segment
    // boundary: a segment is 64 tiles at level 0, and everything that selects it -- base pointers, strides, descriptors, the
    // lanes' chunk offsets -- is computed here once instead of per tile) ----
#pragma unroll 1
    for (int seg = 0; seg < p.nseg; ++seg) {
        if constexpr (C::GRP) {
            if (seg > 0 && ATTN_SEG_FIELD(p, seg, new_group)) {  // block-uniform: close the running group, start a fresh softmax
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) {
                    const float w = gscale / row_sum(qt);
#pragma unroll
                    for (int dt = 0; dt < C::NDT; ++dt) {
                        acc_t[qt][dt] += acc_o[qt][dt] * w;
                        acc_o[qt][dt] = float4v{0.f, 0.f, 0.f, 0.f};
                    }
                    acc_l[qt] = float4v{0.f, 0.f, 0.f, 0.f};
                    m_ref[qt] = 0.f;
                }
                first = true;
                gscale = ATTN_SEG_FIELD(p, seg, gscale);
            }
        }
        if (seg == dup_seg) continue;   // (block-uniform) the rows of segment 0 again: counted there
        const float seg_bias = (seg == 0 && dup_seg > 0) ? 1.0f : 0.0f;
        const int len = ATTN_SEG_FIELD(p, seg, len);
        const int ldk = ATTN_SEG_FIELD(p, seg, ldk), ldv = ATTN_SEG_FIELD(p, seg, ldv);
        const int sdiv = ATTN_SEG_FIELD(p, seg, div), smul = ATTN_SEG_FIELD(p, seg, mul), sadd = ATTN_SEG_FIELD(p, seg, add);
        const long kvb = (long)(n / sdiv) * smul + sadd;
        const half_t* kb = ATTN_SEG_FIELD(p, seg, k) + kvb * len * ldk + h * D;
        const half_t* vb = ATTN_SEG_FIELD(p, seg, v) + kvb * len * ldv + h * D;
        // descriptors over the rows [0, len) of this (frame, head)
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (unsigned)(((len - 1) * ldk + D) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (unsigned)(((len - 1) * ldv + D) * 2), 0x00020000);
        unsigned voff[C::PPW];   // byte offset of this lane's chunk inside a tile, per piece
        int prow[C::PPW];        // its tile row (tail predicate of the partial last tile)
        bool pon[C::PPW];        // PADR: the lane owns a data slot (the pad slot's lanes never issue)
#pragma unroll
        for (int j = 0; j < C::PPW; ++j) {
            const int q = wave + C::NW * j;
            const int ci = (q % C::RCH) * 64 + lane;
            prow[j] = ci / C::RCH;
            const int slot = ci - prow[j] * C::RCH;
            pon[j] = slot < C::CH;
            voff[j] = (unsigned)((prow[j] * (q >= C::RCH ? ldv : ldk) + slot * 8) * 2);
        }
        const int total = (len + C::KV - 1) / C::KV;
        const int tile_k = C::KV * ldk * 2, tile_v = C::KV * ldv * 2;  // scalar byte advance per tile

        auto issue = [&](int tile) {  // tile `tile` of this segment -> stage tile % NST
            half_t* base = lds + (tile % C::NST) * (2 * C::TILE);
            const int row0 = tile * C::KV;
            const bool partial = len - row0 < C::KV;  // wave-uniform: rows past the end read as zero
#pragma unroll
            for (int j = 0; j < C::PPW; ++j) {
                const int q = wave + C::NW * j;
                if (q >= C::PIECES) break;  // wave-uniform
                const bool isv = q >= C::RCH;
                unsigned vo = voff[j];
                if (partial) vo = (row0 + prow[j] < len) ? vo : 0x80000000u;  // out of range for every descriptor
                half_t* dst = base + (isv ? C::TILE : 0) + (q % C::RCH) * 512;
                if (!C::PADR || pon[j]) {  // (PADR: a divergent region -- the pad slot's lanes are off for this instruction)
                    if (isv) __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, tile * tile_v, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, tile * tile_k, 0, 0);
                } else {
                    MV_DMA_LANE_OFF();
                }
            }
        };

        // every wave has left the previous segment's last tile before its stages are refilled
        if (seg > 0) __builtin_amdgcn_s_barrier();
        issue(0);
        if (total > 1) issue(1);

      for (int t = 0; t < total; ++t) {
        const int stage = t % C::NST;
        // this wave's pieces of tile t have landed when at most its pieces of tile t+1 are still in flight
        if (t + 1 < total) {
            if (npw == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (npw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (npw == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave's pieces of tile t are visible; every wave has left tile t-1 (its stage is free)
        asm volatile("" ::: "memory");
        if (t + 2 < total) issue(t + 2);
        const int rows = len - t * C::KV < C::KV ? len - t * C::KV : C::KV;
        const half_t* sK = lds + stage * (2 * C::TILE);
        // V^T fragments: ONE per-lane LDS address per tile (stage base + this lane's row / column inside a [4 kv][16 d] block);
        // the 4 * NDT transpose reads of the tile use immediate offsets from it.  (Casting a generic pointer to the LDS address
        // space per read costs a v_subrev + v_add3 each: 24 of the ~150 VALU instructions per tile before this.)
        const lds_half_t* vrd = lds3 + (stage * (2 * C::TILE) + C::TILE + v_lane_off);
        MV_KEEP_ONE_REGISTER(vrd);  // opaque: ONE address register + immediates (LLVM otherwise hoists 12 addresses and re-adds the stage per read)

        // ---- S'^T = K Q^T - m : acc_s[qt][st][r] = S'[q = l15][kv = 16 st + 4 g + r] ----
        float4v acc_s[C::QT][4];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const float m0 = seg_bias - m_ref[qt];
#pragma unroll
            for (int st = 0; st < 4; ++st) acc_s[qt][st] = float4v{m0, m0, m0, m0};
        }
        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(1);   // VAR bit 32: raised priority around the matrix clusters
#pragma unroll
        for (int c = 0; c < C::NC32; ++c) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const half8v kf = *reinterpret_cast<const half8v*>(sK + (16 * st + l15) * DR + 32 * c + 8 * g);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_s[qt][st] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][c], acc_s[qt][st], 0, 0, 0);
                if constexpr (C::TAIL16) __builtin_amdgcn_sched_barrier(0x7F6);   // (the steps keep their (st, qt) order: see the tail below)
            }
        }
        if constexpr (C::TAIL16) {
            // A 16-deep MFMA that accumulates onto the 32-deep MFMA issued right before it reads a half-written accumulator on gfx950
            // (second hardware finding; hipcc inserts no wait states between the two shapes).  The matrix instructions may
            // therefore NOT be reordered across these points (mask: everything but MFMA may cross): every 32-deep step of the tile is
            // issued before the first 16-deep one, and the tails keep the steps' (st, qt) order -- the tail of an accumulator is at
            // least 4 QT - 2 = 6 independent MFMAs behind the step it accumulates onto.
            static_assert(C::QT >= 2, "the distance argument needs two query tiles per wave");
            __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const half4v kf = *reinterpret_cast<const half4v*>(sK + (16 * st + l15) * DR + 32 * C::NC32 + 4 * g);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_s[qt][st] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, qt16[qt], acc_s[qt][st], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0x7F6);
            }
        }
        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);
        if (rows < C::KV) {  // wave-uniform: the partial last tile of a segment masks its missing keys
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * st + 4 * g + r >= rows) {
#pragma unroll
                        for (int qt = 0; qt < C::QT; ++qt) acc_s[qt][st][r] = -INFINITY;
                    }
        }
        // ---- softmax numerators ----
        // per-lane maxima of the 16 scores each lane holds per query tile (3-input maxima on the raw MFMA results); whether ANY
        // score of the wave exceeds the reference by more than kThr needs no cross-lane traffic -- the per-query maximum (two
        // cross-lane steps through the LDS crossbar) is only formed on the rare rescale path
        float lmx[C::QT];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const float4v &s0 = acc_s[qt][0], &s1 = acc_s[qt][1], &s2 = acc_s[qt][2], &s3 = acc_s[qt][3];
            const float t0 = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
            const float t1 = fmaxf(fmaxf(s0[3], s1[0]), s1[1]);
            const float t2 = fmaxf(fmaxf(s1[2], s1[3]), s2[0]);
            const float t3 = fmaxf(fmaxf(s2[1], s2[2]), s2[3]);
            const float t4 = fmaxf(fmaxf(s3[0], s3[1]), s3[2]);
            lmx[qt] = fmaxf(fmaxf(fmaxf(t0, t1), t2), fmaxf(fmaxf(t3, t4), s3[3]));
        }
        if (first || __any(fmaxf(lmx[0], lmx[1]) > kThr)) {  // wave-uniform; rare after the first tiles of a row
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) {
                float mx = lmx[qt];
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // maximum of this query's 64 scores, relative to its reference
                // move the reference up to the new maximum (on the first tile: to the first estimate, whatever its sign)
                const float delta = first ? mx : fmaxf(mx, 0.f);  // every tile holds >= 1 real key: mx is finite
                m_ref[qt] += delta;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                acc_l[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::NDT; ++dt) acc_o[qt][dt] *= alpha;
#pragma unroll
                for (int st = 0; st < 4; ++st) acc_s[qt][st] -= delta;
            }
        }
        half8v pfrag[C::QT][2];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_s[qt][st][r] = __builtin_amdgcn_exp2f(acc_s[qt][st][r]);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                half8v f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[r] = (half_t)acc_s[qt][2 * cc][r];
                    f[4 + r] = (half_t)acc_s[qt][2 * cc + 1][r];
                }
                pfrag[qt][cc] = f;
            }
        }
        // ---- O^T += V^T P^T; row sums += 1^T P^T (ONES: output row D of the same product) ----
        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            if constexpr (!C::ONES) {
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_l[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones_f, pfrag[qt][cc], acc_l[qt], 0, 0, 0);
            }
#pragma unroll
            for (int dt = 0; dt < C::NDT; ++dt) {
                const lds_half_t* b0 = vrd + (32 * cc * DR + 16 * dt);  // compile-time offset (the loops are fully unrolled)
                short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(b0));
                short4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(b0 + 16 * DR));
                typedef short short8v __attribute__((ext_vector_type(8)));
                short8v tv = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                half8v vf = __builtin_bit_cast(half8v, tv);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt)
                    acc_o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pfrag[qt][cc], acc_o[qt][dt], 0, 0, 0);
            }
        }
        if constexpr ((VAR & 32) != 0) __builtin_amdgcn_s_setprio(0);
        first = false;
      }
    }

    // ---- epilogue: O^T[d = 16 dt + 4 g + r][q = l15] / row sum ----
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const float inv = (C::GRP ? gscale : 1.0f) / row_sum(qt);
        const int qr = q0 + 16 * qt + l15;
        if (qr >= p.lq) continue;
        half_t* orow = p.out + ((long)n * p.lq + qr) * p.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
            const int dcol = 16 * dt + 4 * g;
            if (dcol >= D) continue;
            float4v o = acc_o[qt][dt] * inv;
            if constexpr (C::GRP) o += acc_t[qt][dt];
            if (p.accumulate) {
                half4v prev = *reinterpret_cast<const half4v*>(orow + dcol);
                o = float4v{(float)prev[0], (float)prev[1], (float)prev[2], (float)prev[3]} + o * p.out_scale;
            }
            half4v w = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            *reinterpret_cast<half4v*>(orow + dcol) = w;
        }
    }
}

// ---- cross-attention with RESIDENT keys / values (d = 40 / 80, at most 128 keys over all segments) -------------------------------
// The text cross-attention (77 keys, + 4 IP-Adapter tokens, + FaceID tokens as further softmax groups) under attn3_kernel is a
// (q block, head, frame) grid: every block fetches an 80-byte slice of each query row and writes an 80-byte slice of each output
// row, a key tile ring is set up for two tiles, and the launch sits at ~1.5 TB/s (41-45 us at level 0 whatever the variant,
// profiles/r03g_attn_variants.log).  Here a block owns WHOLE rows: wave h = head h, the block walks 16-row tiles of one query
// batch; every wave holds its head's K fragments (A operand of S^T = K Q^T, loaded once from global: row-contiguous 16-byte
// pieces) and V^T fragments (A operand of O^T = V^T P^T, once through an LDS image of V and ds_read_b64_tr_b16) in registers for
// the block's lifetime.  Per tile a wave loads its 2 D bytes of each query row (the eight waves together: the whole row), forms
// all scores at once (no online rescale: every key is resident), normalises per softmax group BEFORE rounding to fp16
// (P~ = gscale_g * exp2(s - m_g) / l_g in fp32 -- the groups' weighted sum then is ONE P.V product), and stores its slice.
// No barrier after the prologue.  HBM traffic: q read once, out written once (68 MB at level 0 = 17 us at 4 TB/s).
struct XAttnArgs {
    const half_t* q;
    half_t* out;
    int ldq, ldo, nb, lq, heads;
    float scale_log2e;
    int rows_per_block;   // multiple of 16
    int ngroups;
    float gscale[4];
    // per key tile of 16 (tiles never straddle a segment)
    const half_t* tk[8];
    const half_t* tv[8];
    int tldk[8], tldv[8], tlen[8], tdiv[8], tmul[8], tadd[8], tk0[8], tgrp[8];
};

// G2 (compile-time group layout): 0 = a single softmax group (the text-only cross-attention of `musev`: no group selects, the
// weight gscale / l goes on the 4 NDT output accumulators instead of the 4 KT probabilities); KT - 1 = two groups, the second one
// exactly the LAST key tile (text + up to 16 IP-Adapter tokens, `musev_referencenet`: group membership is `t >= G2`, no selects);
// -1 = any layout of up to three groups (membership from XAttnArgs::tgrp, wave-uniform selects).
template <int D, int KT, int G2>
__global__ __launch_bounds__(512, 2) void xattn_kernel(const XAttnArgs p) {
    constexpr bool ONEG = G2 == 0;
    constexpr bool TWOG = G2 > 0;
    constexpr int NC32 = D / 32;            // whole 32-deep contraction chunks
    constexpr bool TAIL = (D % 32) != 0;    // + one 16-deep step (d = 40: columns 32..39 | zeros; d = 80: columns 64..79)
    constexpr int NDT = (D + 15) / 16;      // O^T d-tiles
    constexpr int NCC = (KT + 1) / 2;       // 32-key chunks of the P.V contraction
    static_assert(D % 8 == 0 && (D % 32 == 0 || D % 32 <= 16) && KT >= 1 && KT <= 8, "head dim / key tiles");
    extern __shared__ __attribute__((aligned(16))) half_t lds[];   // V image: [16 KT keys][heads * D + 8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = head
    const int l15 = lane & 15, g = lane >> 4;
    const int bpb = (p.lq + p.rows_per_block - 1) / p.rows_per_block;
    const int n = blockIdx.x / bpb;
    const int r_begin = (blockIdx.x - n * bpb) * p.rows_per_block;
    const int r_end = r_begin + p.rows_per_block < p.lq ? r_begin + p.rows_per_block : p.lq;
    const int C = p.heads * D;
    const int vrs = C + 8;                  // LDS row stride (halfs): 16-byte rows + one pad slot

    // ---- prologue 1: the V rows of every key tile -> LDS (16-byte pieces, rows past a segment's end and the pad slot zero) ----
    {
        const int cpr = vrs / 8;            // 16-byte slots per LDS row (the last one is the pad)
        const int total = KT * 16 * cpr;
        for (int i = tid; i < total; i += blockDim.x) {
            const int row = i / cpr, slot = i - row * cpr;
            const int t = row >> 4, r = row & 15;
            uint4 v = uint4{0, 0, 0, 0};
#pragma unroll
            for (int tt = 0; tt < KT; ++tt) {
                if (tt == t) {
                    const int key = p.tk0[tt] + r;
                    if (key < p.tlen[tt] && slot * 8 < C) {
                        const long kvb = (long)(n / p.tdiv[tt]) * p.tmul[tt] + p.tadd[tt];
                        v = *reinterpret_cast<const uint4*>(p.tv[tt] + (kvb * p.tlen[tt] + key) * p.tldv[tt] + slot * 8);
                    }
                }
            }
            *reinterpret_cast<uint4*>(lds + (long)row * vrs + slot * 8) = v;
        }
    }
    // ---- prologue 2: this head's K fragments, straight from global, as loaded (scale * log2(e) goes on the fp32 scores, inside the
    // fused multiply-add that subtracts the row maximum: no fp16 rounding of a pre-scaled operand, no conversions here) ----
    half8v kf[KT][NC32 > 0 ? NC32 : 1];
    half4v kt16[KT];
    int nvalid[KT];   // valid keys of the tile (16 except a segment's last tile)
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int key = p.tk0[t] + l15;
        const bool ok = key < p.tlen[t];
        nvalid[t] = p.tlen[t] - p.tk0[t];
        const long kvb = (long)(n / p.tdiv[t]) * p.tmul[t] + p.tadd[t];
        const half_t* krow = p.tk[t] + (kvb * p.tlen[t] + (ok ? key : 0)) * p.tldk[t] + h * D;
#pragma unroll
        for (int c = 0; c < NC32; ++c) {
            kf[t][c] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) kf[t][c] = *reinterpret_cast<const half8v*>(krow + 32 * c + 8 * g);
        }
        kt16[t] = half4v{0, 0, 0, 0};
        if constexpr (TAIL) {
            const int dcol = 32 * NC32 + 4 * g;
            if (ok && dcol < D) kt16[t] = *reinterpret_cast<const half4v*>(krow + dcol);
        }
    }
    __syncthreads();
    // ---- prologue 3: this head's V^T fragments out of the LDS image (hardware transpose read, k-slot 8 g' + e of chunk cc =
    // key 32 cc + 16 (e / 4) + 4 g' + e % 4: the order the score accumulators already have) ----
    half8v vf[NDT][NCC];
    {
        typedef __attribute__((address_space(3))) half_t lds_half_t;
        const lds_half_t* lds3 = (const lds_half_t*)lds;
        const int v_lane_off = (4 * g + (l15 >> 2)) * vrs + (l15 & 3) * 4 + h * D;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const lds_half_t* b0 = lds3 + (v_lane_off + 32 * cc * vrs + 16 * dt);
                short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(b0));
                short4v t1 = short4v{0, 0, 0, 0};
                if (2 * cc + 1 < KT) t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(b0 + 16 * vrs));
                typedef short short8v __attribute__((ext_vector_type(8)));
                short8v tv = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                vf[dt][cc] = __builtin_bit_cast(half8v, tv);
            }
        }
    }

    // ---- the block's rows, 16 at a time; the next tile's query slice is in flight while this one is worked on ----
    half8v qraw[NC32 > 0 ? NC32 : 1];
    half4v qtraw = half4v{0, 0, 0, 0};
    auto fetch_q = [&](int r0) __attribute__((always_inline)) {
        const int qr = r0 + l15;
        const half_t* qrow = p.q + ((long)n * p.lq + (qr < r_end ? qr : r_begin)) * p.ldq + h * D;
#pragma unroll
        for (int c = 0; c < NC32; ++c) qraw[c] = *reinterpret_cast<const half8v*>(qrow + 32 * c + 8 * g);
        if constexpr (TAIL) {
            const int dcol = 32 * NC32 + 4 * g;
            qtraw = half4v{0, 0, 0, 0};
            if (dcol < D) qtraw = *reinterpret_cast<const half4v*>(qrow + dcol);
        }
    };
    fetch_q(r_begin);
#pragma unroll 1
    for (int r0 = r_begin; r0 < r_end; r0 += 16) {
        // B operand of S^T = K Q^T (the scale sits in the K fragments)
        half8v qf[NC32 > 0 ? NC32 : 1];
#pragma unroll
        for (int c = 0; c < NC32; ++c) qf[c] = qraw[c];
        const half4v qt16 = qtraw;
        if (r0 + 16 < r_end) fetch_q(r0 + 16);

        // acc[t][r] = S[q = l15][key 16 t + 4 g + r]  (raw dot products: the scale goes into the exponential's fused multiply-add)
        float4v acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            acc[t] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC32; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][c], qf[c], acc[t], 0, 0, 0);
            // the 16-deep tail goes into its OWN accumulator and is added on the vector ALU: a v_mfma_f32_16x16x16_f16 issued right
            // behind the v_mfma_f32_16x16x32_f16 whose result it accumulates onto read a half-written accumulator on the MI355X
            // (profiles/r05d_debug_dump.log: registers 0 and 1 of the 4-register result wrong, run-to-run and wave-to-wave different,
            // right whenever the SIMD's other wave happened to issue in between) -- hipcc inserts no wait states between the two
            if constexpr (TAIL) {
                const float4v tl = __builtin_amdgcn_mfma_f32_16x16x16f16(kt16[t], qt16, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc[t] += tl;
            }
            if (nvalid[t] < 16) {  // wave-uniform: a segment's last tile masks its missing keys
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r >= nvalid[t]) acc[t][r] = -INFINITY;
            }
        }
        // softmax: per group maximum, exponentials, row sum.  Per-tile maxima / sums first, then the groups pick their tiles
        // (wave-uniform selects): ~2 VALU per (group, tile) instead of a pass over the tile's scores per group
        float tmax[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) tmax[t] = fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3]));
        float mg[4] = {0.f, 0.f, 0.f, 0.f}, wg[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ONEG) {
            float m = tmax[0];
#pragma unroll
            for (int t = 1; t < KT; ++t) m = fmaxf(m, tmax[t]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            mg[0] = m;
        } else if constexpr (TWOG) {
            float m0 = tmax[0], m1 = tmax[G2];
#pragma unroll
            for (int t = 1; t < KT; ++t) {
                if (t < G2) m0 = fmaxf(m0, tmax[t]);
                else if (t > G2) m1 = fmaxf(m1, tmax[t]);
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 16, 64));
            m1 = fmaxf(m1, __shfl_xor(m1, 16, 64));
            mg[0] = fmaxf(m0, __shfl_xor(m0, 32, 64));
            mg[1] = fmaxf(m1, __shfl_xor(m1, 32, 64));
        } else {
#pragma unroll
            for (int gi = 0; gi < 3; ++gi) {   // branch-free: an absent group comes out as m = -inf, l = 0 and is never selected
                float m = -INFINITY;
#pragma unroll
                for (int t = 0; t < KT; ++t) m = p.tgrp[t] == gi ? fmaxf(m, tmax[t]) : m;
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                mg[gi] = m;
            }
        }
        float tsum[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            float m = mg[0];
            if constexpr (TWOG) {
                m = t >= G2 ? mg[1] : mg[0];
            } else if constexpr (!ONEG) {
                const int gi = p.tgrp[t];
                m = gi == 0 ? mg[0] : gi == 1 ? mg[1] : mg[2];
            }
            const float nms = -m * p.scale_log2e;   // (scale > 0: the maximum of the raw scores is the maximum of the scaled ones)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_amdgcn_exp2f(fmaf(acc[t][r], p.scale_log2e, nms));
            tsum[t] = (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        }
        if constexpr (ONEG) {
            float l = tsum[0];
#pragma unroll
            for (int t = 1; t < KT; ++t) l += tsum[t];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            wg[0] = p.gscale[0] / l;   // a group holds at least one real key: l >= 1
        } else if constexpr (TWOG) {
            float l0 = tsum[0], l1 = tsum[G2];
#pragma unroll
            for (int t = 1; t < KT; ++t) {
                if (t < G2) l0 += tsum[t];
                else if (t > G2) l1 += tsum[t];
            }
            l0 += __shfl_xor(l0, 16, 64);
            l1 += __shfl_xor(l1, 16, 64);
            l0 += __shfl_xor(l0, 32, 64);
            l1 += __shfl_xor(l1, 32, 64);
            wg[0] = p.gscale[0] / l0;
            wg[1] = p.gscale[1] / l1;
        } else {
#pragma unroll
            for (int gi = 0; gi < 3; ++gi) {
                float l = 0.f;
#pragma unroll
                for (int t = 0; t < KT; ++t) l += p.tgrp[t] == gi ? tsum[t] : 0.f;
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
                wg[gi] = p.gscale[gi] / l;
            }
        }
        // O^T[d][q] = sum over the key chunks of V^T P^T  (several groups: P leaves normalised and weighted, P~ = w_g P, so that the
        // groups' weighted sum is this one product; one group: the weight goes on the output accumulators)
        float4v acc_o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) acc_o[dt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            half8v pf = half8v{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int t = 2 * cc + half;
                if (t < KT) {
                    float w = 1.0f;
                    if constexpr (TWOG) {
                        w = t >= G2 ? wg[1] : wg[0];
                    } else if constexpr (!ONEG) {
                        const int gi = p.tgrp[t];
                        w = gi == 0 ? wg[0] : gi == 1 ? wg[1] : wg[2];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf[4 * half + r] = (half_t)(ONEG ? acc[t][r] : acc[t][r] * w);
                }
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc_o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dt][cc], pf, acc_o[dt], 0, 0, 0);
        }
        if constexpr (ONEG) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc_o[dt] *= wg[0];
        }
        // acc_o[dt][r] = O[q = l15][d = 16 dt + 4 g + r]
        const int qr = r0 + l15;
        if (qr < r_end) {
            half_t* orow = p.out + ((long)n * p.lq + qr) * p.ldo + h * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int dcol = 16 * dt + 4 * g;
                if (dcol < D) {
                    const float4v o = acc_o[dt];
                    *reinterpret_cast<half4v*>(orow + dcol) = half4v{(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
struct TAttnArgs {
    const half_t* q;
    const half_t* k;
    const half_t* v;
    half_t* out;
    int ldq, ldk, ldv, ldo;
    int b, t, hw, heads, d;
    float scale;
    long items;  // b * hw * heads
};

template <int GL>  // lanes per (pixel, head) item: 16 or 32 (>= T)
__global__ __launch_bounds__(256) void tattn_kernel(const TAttnArgs a) {
    const int tid = threadIdx.x;
    const int gl = tid % GL;  // query frame of this lane
    const long item = (long)blockIdx.x * (256 / GL) + tid / GL;
    if (item >= a.items) return;
    const int h = (int)(item % a.heads);
    const long bp = item / a.heads;
    const int pix = (int)(bp % a.hw);
    const int b = (int)(bp / a.hw);
    const bool active = gl < a.t;
    const int tq = active ? gl : 0;
    const long qrow = ((long)b * a.t + tq) * a.hw + pix;
    const int dch = a.d >> 3;

    float s[GL];
#pragma unroll
    for (int j = 0; j < GL; ++j) s[j] = 0.f;
    // key/value rows of frame j (clamped so every load is unconditional; j >= T is masked in the softmax)
    const long row0 = (long)b * a.t * a.hw + pix;
    for (int ch = 0; ch < dch; ++ch) {
        half8v qv = *reinterpret_cast<const half8v*>(a.q + qrow * a.ldq + h * a.d + ch * 8);
        float qf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
#pragma unroll
        for (int jb = 0; jb < GL; jb += 16) {
            half8v kv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int jj = (jb + j) < a.t ? (jb + j) : a.t - 1;
                kv[j] = *reinterpret_cast<const half8v*>(a.k + (row0 + (long)jj * a.hw) * a.ldk + h * a.d + ch * 8);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf(qf[e], (float)kv[j][e], acc);
                s[jb + j] += acc;
            }
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < GL; ++j)
        if (j < a.t) mx = fmaxf(mx, s[j] * a.scale);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < GL; ++j) {
        float e = (j < a.t) ? __expf(s[j] * a.scale - mx) : 0.f;
        s[j] = e;
        l += e;
    }
    const float inv = 1.0f / l;
    for (int ch = 0; ch < dch; ++ch) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int jb = 0; jb < GL; jb += 16) {
            half8v vv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int jj = (jb + j) < a.t ? (jb + j) : a.t - 1;
                vv[j] = *reinterpret_cast<const half8v*>(a.v + (row0 + (long)jj * a.hw) * a.ldv + h * a.d + ch * 8);
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(s[jb + j], (float)vv[j][e], o[e]);
            }
        }
        if (active) {
            half8v w;
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (half_t)(o[e] * inv);
            *reinterpret_cast<half8v*>(a.out + qrow * a.ldo + h * a.d + ch * 8) = w;
        }
    }
}

// ---- temporal attention on the matrix cores (T <= 16, d = 40 / 80 / 160): one wave per (pixel, head) item ----------------------
// tattn_kernel spends ~1200 VALU/LDS instructions per item on 13 x 13 x d scalar FMAs and is instruction-issue bound (level 0:
// 107 us for 272 MB of q/k/v/out, profiles/r02e).  Here an item is two small MFMA products:
//   S^T[key][query] = K Q^T   16x16x32 steps over d; A = K rows, B = Q rows, both loaded straight from global memory in fragment
//                             shape (lane (l15 = frame, g) holds 8 halfs at d = 32c + 8g; frames >= T re-read frame T-1 and are masked)
//   softmax over keys         a lane holds keys 4g..4g+3 of query l15: 4 registers + two xor-shuffles (16, 32)
//   O^T[d][query]   = V^T P^T 16x16x16 steps; B = P^T is the S^T accumulator layout as it stands (keys 4g+r), A = V^T comes from the
//                             wave's private LDS image of V ([key][DP] row-major) through ds_read_b64_tr_b16
// The accumulator of O^T holds 4 consecutive d of one query per lane: 8-byte stores.  A wave walks `ipw` items (stride 4 inside its
// block, so the block covers 4 * ipw consecutive (pixel, head) items = whole 128-byte lines of the [.., heads * d] rows) and loads
// the next item's fragments before it computes the current one.  LDS rows are padded to an odd multiple of 32 bytes: the
// ds_write_b128 of 8 keys and the transposed reads of 8 keys x 32 bytes are conflict-free.
template <int D> struct TAttn3Cfg {
    static constexpr int NCH = (D + 31) / 32;  // 32-wide d chunks of the S^T product
    static constexpr int NJ = (D + 15) / 16;   // 16-wide d blocks of the O^T product
    static constexpr int DP = D == 40 ? 48 : (D == 80 ? 80 : 176);
    static_assert(DP >= 16 * NJ && (DP * 2 / 32) % 2 == 1, "LDS row: >= the columns read, odd multiple of 32 bytes");
};

template <int D>
__global__ __launch_bounds__(256) void tattn3_kernel(const TAttnArgs a, const int ipw) {
    using C = TAttn3Cfg<D>;
    __shared__ __attribute__((aligned(16))) half_t vs[4][16 * C::DP];
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // everything derived from the item id stays scalar
    half_t* myv = vs[wave];
    const int fr = l15 < a.t ? l15 : a.t - 1;
    const float c2 = a.scale * 1.4426950408889634f;
    const half8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // per-lane element offsets inside an item (frame fr, d piece 8g); the launcher guarantees 16 * hw * ld < 2^31
    const int lq = fr * a.hw * a.ldq + 8 * g, lk = fr * a.hw * a.ldk + 8 * g, lv = fr * a.hw * a.ldv + 8 * g;
    const int lo = fr * a.hw * a.ldo + 4 * g;
    // the wave's items: first + 4 * i, i < n (wave-uniform); (b, pix, h) advance incrementally
    const int first = mv_xcd_remap((int)blockIdx.x, (int)gridDim.x) * (4 * ipw) + wave;
    int n = first < (int)a.items ? ((int)a.items - first + 3) / 4 : 0;
    n = n < ipw ? n : ipw;
    int h = first % a.heads, pix = (first / a.heads) % a.hw, b = (first / a.heads) / a.hw;

    half8v q0[C::NCH], k0[C::NCH], v0[C::NCH], q1[C::NCH], k1[C::NCH], v1[C::NCH];
    half_t *o0 = nullptr, *o1 = nullptr;
    auto fetch = [&](half8v* q, half8v* k, half8v* v, half_t** o) {  // loads the item at (b, pix, h), then steps to the wave's next one
        const long srow = (long)b * a.t * a.hw + pix;
        const half_t* qs = a.q + srow * a.ldq + h * D;
        const half_t* ks = a.k + srow * a.ldk + h * D;
        const half_t* vsrc = a.v + srow * a.ldv + h * D;
        *o = a.out + srow * a.ldo + h * D;
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) {
            if (32 * c + 8 * g < D) {
                q[c] = *reinterpret_cast<const half8v*>(qs + lq + 32 * c);
                k[c] = *reinterpret_cast<const half8v*>(ks + lk + 32 * c);
                v[c] = *reinterpret_cast<const half8v*>(vsrc + lv + 32 * c);
            } else {
                q[c] = zero8;
                k[c] = zero8;
                v[c] = zero8;
            }
        }
        h += 4;
        while (h >= a.heads) {
            h -= a.heads;
            ++pix;
        }
        while (pix >= a.hw) {
            pix -= a.hw;
            ++b;
        }
    };
    auto compute = [&](const half8v* q, const half8v* k, const half8v* v, half_t* o) {
        // ---- S^T = K Q^T ----
        float4v st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C::NCH; ++c) st = __builtin_amdgcn_mfma_f32_16x16x32_f16(k[c], q[c], st, 0, 0, 0);
        // ---- V rows -> the wave's LDS image ----
#pragma unroll
        for (int c = 0; c < C::NCH; ++c)
            if (32 * c + 8 * g < D) *reinterpret_cast<half8v*>(myv + l15 * C::DP + 32 * c + 8 * g) = v[c];
        // ---- softmax over the keys of query l15 (this lane: keys 4g .. 4g+3) ----
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (4 * g + r < a.t) ? st[r] * c2 : -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(sc[r] - mx);
            l += sc[r];
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const half4v pf = {(half_t)(sc[0] * inv), (half_t)(sc[1] * inv), (half_t)(sc[2] * inv), (half_t)(sc[3] * inv)};
        // ---- O^T = V^T P^T: every lane of the wave has written its V piece; the DS operations of a wave execute in order ----
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const half_t* vrow = myv + (4 * g + (l15 >> 2)) * C::DP + (l15 & 3) * 4;
#pragma unroll
        for (int j = 0; j < C::NJ; ++j) {
            const short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(vrow + 16 * j));
            const half4v vt = __builtin_bit_cast(half4v, t);
            const float4v acc = __builtin_amdgcn_mfma_f32_16x16x16f16(vt, pf, float4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (l15 < a.t && 16 * j + 4 * g < D) {
                const half4v w = {(half_t)acc[0], (half_t)acc[1], (half_t)acc[2], (half_t)acc[3]};
                *reinterpret_cast<half4v*>(o + lo + 16 * j) = w;
            }
        }
        asm volatile("" ::: "memory");
    };
    if constexpr (D > 80) {  // 60 VGPRs of fragments per item: one register set, the other resident waves hide the loads
        for (int it = 0; it < n; ++it) {
            fetch(q0, k0, v0, &o0);
            compute(q0, k0, v0, o0);
        }
        return;
    }
    if (n > 0) fetch(q0, k0, v0, &o0);
    for (int it = 0; it < n; it += 2) {  // two register sets: the next item's loads are in flight while this one computes
        if (it + 1 < n) fetch(q1, k1, v1, &o1);
        compute(q0, k0, v0, o0);
        if (it + 1 < n) {
            if (it + 2 < n) fetch(q0, k0, v0, &o0);
            compute(q1, k1, v1, o1);
        }
    }
}

}  // namespace

namespace {

// kernel variants the library runs (Attn3Cfg: 1 = 16-deep contraction tail, 2 = 48-half rows with the ones column at d = 40);
// chosen from the same-box A/B of tools/gpu_attn_bench.py (profiles/r03*_attn_variants.log)
constexpr int kAttnVar40 = 47;  // r03a / r03g, level 0: var 0 1.870 ms -> var 7 1.715 (26 frames), 0.964 -> 0.809 (13 frames); var 15 a further -5.5 % / -1.5 %
constexpr int kAttnVar80 = 8;   // r03g: eight waves -5 % at 13 frames (the two-stream default), +4 % at 26; the 16-deep tail is 3-4 % slower at d = 80

template <int D, int VAR>
int launch_attn3(const AttnArgs& a, dim3 grid1, hipStream_t s) {
    constexpr int smem = Attn3Cfg<D, VAR>::LDS_HALFS * 2;
    static bool attr_done = false;  // idempotent one-time attribute of this instantiation (up to 62 KB of dynamic LDS)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_kernel<D, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        MV_REQUIRE(e == hipSuccess, "mv_attention_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_done = true;
    }
    constexpr int QB = Attn3Cfg<D, VAR>::QB;
    const dim3 grid((unsigned)(((a.lq + QB - 1) / QB) * a.heads * a.nb));  // 1-D: the kernel maps ids to (q block, head, frame)
    hipLaunchKernelGGL((attn3_kernel<D, VAR>), grid, dim3(64 * Attn3Cfg<D, VAR>::NW), smem, s, a);
    return MV_OK;
}


int attn_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// resident-K/V cross-attention (mv_attn_desc.resident_kv): eligibility is the caller's to check (mv_attention_resident_ok)
template <int D, int KT, int G2>
int launch_xattn(const XAttnArgs& a, unsigned grid, int smem, hipStream_t s) {
    static int attr_smem = 0;  // idempotent one-time attribute of this instantiation (the LDS image of V: up to 160 KB)
    if (smem > attr_smem) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel<D, KT, G2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        MV_REQUIRE(e == hipSuccess, "mv_attention_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        attr_smem = smem;
    }
    hipLaunchKernelGGL((xattn_kernel<D, KT, G2>), dim3(grid), dim3(64 * a.heads), smem, s, a);
    return MV_OK;
}

// key tiles of 16 over all segments (a tile never straddles a segment); -1 if the problem does not fit the resident kernel
int xattn_key_tiles(const mv_attn_desc* d) {
    if (!(d->d == 40 || d->d == 80) || d->heads < 1 || d->heads > 8 || d->accumulate || d->nseg < 1 || d->nseg > MV_ATTN_MAX_SEG) return -1;
    if (d->ldq % 8 || d->ldo % 4) return -1;
    int kt = 0, groups = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->seg[s].len <= 0 || d->seg[s].ldk % 8 || d->seg[s].ldv % 8) return -1;
        kt += (d->seg[s].len + 15) / 16;
        groups += (s == 0 || d->seg[s].new_group) ? 1 : 0;
    }
    if (kt > 8 || groups > 3) return -1;
    const long smem = (long)kt * 16 * (d->heads * d->d + 8) * 2;
    return smem <= 160 * 1024 ? kt : -1;
}

int xattn_launch(const mv_attn_desc* d, void* stream) {
    MV_REQUIRE(d->nb > 0 && d->lq > 0 && d->nb <= 65535, "mv_attention_f16: empty problem");
    // scale * log2(e) goes on the fp32 scores inside the exponential's fma while the row maximum is taken over the RAW scores:
    // only a positive scale keeps "max of the raw scores" the maximum of the scaled ones
    MV_REQUIRE(d->scale > 0.f, "mv_attention_f16: resident_kv needs scale > 0 (got %g)", (double)d->scale);
    const int kt = xattn_key_tiles(d);
    MV_REQUIRE(kt > 0, "mv_attention_f16: resident_kv needs d in {40, 80}, heads <= 8, <= 128 keys in tiles of 16 per segment, <= 3 groups, no accumulate, "
                       "and an LDS image of V under 160 KB (ask mv_attention_resident_ok)");
    XAttnArgs a;
    a.q = (const half_t*)d->q; a.out = (half_t*)d->out; a.ldq = d->ldq; a.ldo = d->ldo;
    a.nb = d->nb; a.lq = d->lq; a.heads = d->heads;
    a.scale_log2e = d->scale * 1.4426950408889634f;
    int t = 0, grp = -1;
    for (int i = 0; i < 4; ++i) a.gscale[i] = 1.0f;
    for (int s = 0; s < d->nseg; ++s) {
        const mv_attn_seg& g = d->seg[s];
        MV_REQUIRE(g.k && g.v && g.div > 0, "mv_attention_f16: bad segment %d", s);
        if (s == 0 || g.new_group) {
            ++grp;
            a.gscale[grp] = g.new_group ? g.group_scale : 1.0f;
        }
        for (int k0 = 0; k0 < g.len; k0 += 16, ++t) {
            a.tk[t] = (const half_t*)g.k; a.tv[t] = (const half_t*)g.v;
            a.tldk[t] = g.ldk; a.tldv[t] = g.ldv; a.tlen[t] = g.len; a.tdiv[t] = g.div; a.tmul[t] = g.mul; a.tadd[t] = g.add;
            a.tk0[t] = k0; a.tgrp[t] = grp;
        }
    }
    a.ngroups = grp + 1;
    for (int i = t; i < 8; ++i) {
        a.tk[i] = a.tk[0]; a.tv[i] = a.tv[0];
        a.tldk[i] = a.tldv[i] = 0; a.tlen[i] = 0; a.tdiv[i] = 1; a.tmul[i] = a.tadd[i] = a.tk0[i] = 0; a.tgrp[i] = 0;
    }
    // rows per block: whole rounds of the CUs with as little idle tail as possible; the prologue (K / V fragments) costs about as
    // much as six row tiles, so few, long blocks win over many short ones
    const int cus = attn_num_cus();
    int best_rt = 1;
    long best_cost = -1;
    for (int rt = 1; rt <= 64; ++rt) {
        const long blocks = (long)d->nb * ((d->lq + 16 * rt - 1) / (16 * rt));
        const long rounds = (blocks + cus - 1) / cus;
        const long cost = rounds * (rt + 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_rt = rt; }
    }
    a.rows_per_block = 16 * best_rt;
    if (d->resident_kv >= 16) a.rows_per_block = d->resident_kv / 16 * 16;  // the caller's choice (tuning: tools/gpu_xattn_bench.py sweeps it)
    const long blocks = (long)d->nb * ((d->lq + a.rows_per_block - 1) / a.rows_per_block);
    MV_REQUIRE(blocks <= 0x7fffffffL, "mv_attention_f16: grid too large");
    const int smem = kt * 16 * (d->heads * d->d + 8) * 2;
    hipStream_t s = (hipStream_t)stream;
    int rc = MV_ERR_INVALID;
    // the group layout the kernel is compiled for: one group; two groups with the second one = the last key tile; anything else
    const bool last_tile_group = a.ngroups == 2 && kt >= 2 && a.tgrp[kt - 1] == 1 && a.tgrp[kt - 2] == 0;
    switch (kt * 100 + d->d) {
#define MV_XA(KT_, D_)                                                                                                  \
    case KT_ * 100 + D_:                                                                                                \
        rc = a.ngroups == 1 ? launch_xattn<D_, KT_, 0>(a, (unsigned)blocks, smem, s)                                    \
             : (last_tile_group && KT_ >= 2) ? launch_xattn<D_, KT_, (KT_ >= 2 ? KT_ - 1 : -1)>(a, (unsigned)blocks, smem, s)  \
                                             : launch_xattn<D_, KT_, -1>(a, (unsigned)blocks, smem, s);                 \
        break;
        MV_XA(1, 40) MV_XA(2, 40) MV_XA(3, 40) MV_XA(4, 40) MV_XA(5, 40) MV_XA(6, 40) MV_XA(7, 40) MV_XA(8, 40)
        MV_XA(1, 80) MV_XA(2, 80) MV_XA(3, 80) MV_XA(4, 80) MV_XA(5, 80) MV_XA(6, 80) MV_XA(7, 80) MV_XA(8, 80)
#undef MV_XA
    }
    if (rc != MV_OK) return rc;
    MV_CHECK_LAUNCH("mv_attention_f16(resident)");
    return MV_OK;
}

int attention_launch(const mv_attn_desc* d, int var40, int var80, void* stream) {
    MV_REQUIRE(d && d->q && d->out, "mv_attention_f16: null pointer");
    if (d->resident_kv) return xattn_launch(d, stream);
    MV_REQUIRE(d->nseg >= 1 && d->nseg <= MV_ATTN_MAX_SEG, "mv_attention_f16: nseg=%d out of range", d->nseg);
    MV_REQUIRE(d->d == 40 || d->d == 80 || d->d == 160, "mv_attention_f16: head dim %d not in {40,80,160}", d->d);
    MV_REQUIRE(d->nb > 0 && d->lq > 0 && d->heads > 0, "mv_attention_f16: empty problem");
    MV_REQUIRE(d->ldq % 8 == 0 && d->ldo % 4 == 0, "mv_attention_f16: ldq %% 8 / ldo %% 4");
    MV_REQUIRE(d->nb <= 65535 && d->heads <= 65535, "mv_attention_f16: grid too large");
    AttnArgs a;
    a.q = (const half_t*)d->q; a.out = (half_t*)d->out; a.ldq = d->ldq; a.ldo = d->ldo;
    a.nb = d->nb; a.lq = d->lq; a.heads = d->heads;
    a.scale_log2e = d->scale * 1.4426950408889634f;
    a.nseg = d->nseg; a.accumulate = d->accumulate; a.out_scale = d->out_scale;
    bool grouped = false;
    for (int s = 0; s < MV_ATTN_MAX_SEG; ++s) {
        if (s < d->nseg) {
            const mv_attn_seg& g = d->seg[s];
            MV_REQUIRE(g.k && g.v && g.len > 0 && g.div > 0, "mv_attention_f16: bad segment %d", s);
            MV_REQUIRE(g.ldk % 8 == 0 && g.ldv % 8 == 0, "mv_attention_f16: segment %d ldk/ldv %% 8", s);
            a.seg[s] = SegArgs{(const half_t*)g.k, (const half_t*)g.v, g.ldk, g.ldv, g.len, g.div, g.mul, g.add, (s > 0 && g.new_group) ? 1 : 0,
                               g.new_group ? g.group_scale : 1.0f};
            grouped = grouped || (s > 0 && g.new_group) || (s == 0 && g.new_group && g.group_scale != 1.0f);
        } else {
            a.seg[s] = SegArgs{nullptr, nullptr, 0, 0, 0, 1, 0, 0, 0, 1.0f};
        }
    }
    MV_REQUIRE(!grouped || d->d == 40 || d->d == 80, "mv_attention_f16: softmax groups are built for head dims 40 and 80 (d=%d: one launch per group with accumulate)", d->d);
    const int qb = d->d > 80 ? 64 : 128;  // AttnCfg<D>::QB
    dim3 grid((unsigned)((d->lq + qb - 1) / qb), (unsigned)d->heads, (unsigned)d->nb);
    hipStream_t s = (hipStream_t)stream;
    // d = 40 / 80: OPT 1 (row sums out of the P.V MFMA, 3-input maxima, exact rescale skip); the round-1 / round-2 A/B of the
    // other variants (double-buffered tiles, pkrtz packing, buffer-descriptor K/V fetch, K / V row strides) measured within
    // +-3 % of it at d = 40 and -20..-30 % at d = 80 (profiles/r02a_attn_variant_ab.log): removed from the library
    if (d->d == 40 || d->d == 80) {
        // LDS-DMA kernel: every K / V source must span < 2 GiB from its (frame, head) base (32-bit descriptor offsets)
        for (int sg = 0; sg < d->nseg; ++sg)
            MV_REQUIRE((long)d->seg[sg].len * d->seg[sg].ldk * 2 < 0x7fffffffL && (long)d->seg[sg].len * d->seg[sg].ldv * 2 < 0x7fffffffL,
                       "mv_attention_f16: segment %d spans 2 GiB or more per key batch", sg);
        const dim3 grid1((unsigned)(((d->lq + 127) / 128) * d->heads * d->nb));  // 1-D: the kernel maps ids to (q block, head, frame)
        int rc = MV_OK;
        if (grouped) {
            // (three waves per SIMD at d = 40: the second accumulator does not fit the 128-register budget of variant bit 4)
            rc = d->d == 40 ? launch_attn3<40, 59>(a, grid1, s) : launch_attn3<80, 24>(a, grid1, s);
        } else if (d->d == 40) {
#ifdef MV_EXPERIMENT
            if (var40 == 0) rc = launch_attn3<40, 0>(a, grid1, s);
            else if (var40 == 1) rc = launch_attn3<40, 1>(a, grid1, s);
            else if (var40 == 2) rc = launch_attn3<40, 2>(a, grid1, s);
            else if (var40 == 6) rc = launch_attn3<40, 6>(a, grid1, s);
            else if (var40 == 7) rc = launch_attn3<40, 7>(a, grid1, s);
            else if (var40 == 15) rc = launch_attn3<40, 15>(a, grid1, s);
            else if (var40 == 14) rc = launch_attn3<40, 14>(a, grid1, s);
            else if (var40 == 11) rc = launch_attn3<40, 11>(a, grid1, s);
            else if (var40 == 47) rc = launch_attn3<40, 47>(a, grid1, s);
            else
#endif
            rc = launch_attn3<40, kAttnVar40>(a, grid1, s);
        } else {
#ifdef MV_EXPERIMENT
            if (var80 == 0) rc = launch_attn3<80, 0>(a, grid1, s);
            else if (var80 == 1) rc = launch_attn3<80, 1>(a, grid1, s);
            else if (var80 == 8) rc = launch_attn3<80, 8>(a, grid1, s);
            else
#endif
            rc = launch_attn3<80, kAttnVar80>(a, grid1, s);
        }
        if (rc != MV_OK) return rc;
    } else {
        hipLaunchKernelGGL((attn_kernel<160, 0>), grid, dim3(256), 0, s, a);
    }
    MV_CHECK_LAUNCH("mv_attention_f16");
    return MV_OK;
}

}  // namespace

extern "C" int mv_attention_f16(const mv_attn_desc* d, void* stream) { return attention_launch(d, kAttnVar40, kAttnVar80, stream); }

extern "C" int mv_attention_resident_ok(const mv_attn_desc* d) { return d && d->q && d->out && d->nb > 0 && d->lq > 0 && xattn_key_tiles(d) > 0 ? 1 : 0; }

#ifdef MV_EXPERIMENT
// experiment builds only (tools/gpu_attn_bench.py): the same entry with the kernel variant chosen per call
extern "C" int mv_attention_f16_var(const mv_attn_desc* d, int var40, int var80, void* stream) { return attention_launch(d, var40, var80, stream); }
#endif

extern "C" int mv_temporal_attention_f16(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk,
                                         int32_t ldv, void* out, int32_t ldo, int32_t b, int32_t t, int32_t hw,
                                         int32_t heads, int32_t d, float scale, void* stream) {
    MV_REQUIRE(q && k && v && out, "mv_temporal_attention_f16: null pointer");
    MV_REQUIRE(t >= 1 && t <= 32, "mv_temporal_attention_f16: T=%d not in [1,32]", t);
    MV_REQUIRE(d % 8 == 0 && d > 0, "mv_temporal_attention_f16: head dim %% 8");
    MV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "mv_temporal_attention_f16: ld %% 8");
    TAttnArgs a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.out = (half_t*)out;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.b = b; a.t = t; a.hw = hw; a.heads = heads; a.d = d;
    a.scale = scale; a.items = (long)b * hw * heads;
    hipStream_t s = (hipStream_t)stream;
    // T <= 16 at the UNet's head widths: the MFMA kernel; items per wave sized so that the grid still fills the 256 CUs
    if (t <= 16 && (d == 40 || d == 80 || d == 160) && a.items < 0x7fffffffL &&
        16L * hw * ldq < 0x7fffffffL && 16L * hw * ldk < 0x7fffffffL && 16L * hw * ldv < 0x7fffffffL && 16L * hw * ldo < 0x7fffffffL) {
        int ipw = (int)(a.items / 8192);
        ipw = ipw < 1 ? 1 : (ipw > 8 ? 8 : ipw);
        const long per_block = 4L * ipw;
        const unsigned grid = (unsigned)((a.items + per_block - 1) / per_block);
        if (d == 40) hipLaunchKernelGGL((tattn3_kernel<40>), dim3(grid), dim3(256), 0, s, a, ipw);
        else if (d == 80) hipLaunchKernelGGL((tattn3_kernel<80>), dim3(grid), dim3(256), 0, s, a, ipw);
        else hipLaunchKernelGGL((tattn3_kernel<160>), dim3(grid), dim3(256), 0, s, a, ipw);
        MV_CHECK_LAUNCH("mv_temporal_attention_f16");
        return MV_OK;
    }
    if (t <= 16) {
        const unsigned grid = (unsigned)((a.items + 15) / 16);
        hipLaunchKernelGGL(tattn_kernel<16>, dim3(grid), dim3(256), 0, s, a);
    } else {
        const unsigned grid = (unsigned)((a.items + 7) / 8);
        hipLaunchKernelGGL(tattn_kernel<32>, dim3(grid), dim3(256), 0, s, a);
    }
    MV_CHECK_LAUNCH("mv_temporal_attention_f16");
    return MV_OK;
}
