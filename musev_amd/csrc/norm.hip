// norm.hip -- GroupNorm(32)(+SiLU) over channels-last tensors and LayerNorm over the channel dim (gfx950).
//
// Both are HBM-bound streaming kernels: 16-byte (8 x fp16) vector loads, fp32 statistics, 64-lane
// shuffle reductions.  GroupNorm runs as three launches on one stream (small tensors: one, see gn_small_kernel):
//   1. gn_stats    : per (item, row-split): per-CHANNEL sum / sum-of-squares (fp32; per-channel first because on the up path the
//                    input is the channel concat of two tensors and groups of 30 / 60 channels straddle both the 8-channel
//                    vectors and the concat seam), folded to per-GROUP partials [item][split][group][2] inside the block.
//   2. gn_finalize : one wave per (item, group): fold the splits (double, fixed order), emit mean / rstd.
//   3. gn_apply    : y = silu?((x - mean) * rstd * gamma + beta), writing the concatenated [rows][c1+c2] tensor (this is the
//                    only place the up-path concat is ever materialised, already normalised).
// For TemporalConvLayer / TransformerTemporalModel the statistics span (T, H, W): the caller passes
// n_items = B and rows = T*H*W, which is contiguous in the [B,T,H,W,C] layout.
#include "common.h"

namespace {

struct GnArgs {
    const half_t* x1;
    const half_t* x2;
    int c1, c2, ld1, ld2;
    long rows;
    int nsplit;
    int oc;  // (c1+c2)/8 channel octets
    int rl;  // row lanes per block
    float* partial;      // [item][split][group][2]
    float* stat;         // [item][group][2] = mean, rstd
    long n_items;
    const half_t* gamma;
    const half_t* beta;
    half_t* y;
    int ldy;
    int silu;
    int groups;
    float eps;
    // producer-side column statistics (mv_groupnorm_cs_f16): [rows / rpt][c][2] per source
    const float* cs1;
    const float* cs2;
    int rpt1, rpt2;
    // two-fp16 carry through the apply pass (single source): x = x1 + x1_lo in fp32, y = fp16(v), y_lo = fp16(v - y)
    const half_t* x1_lo;
    half_t* y_lo;
};

__device__ __forceinline__ const half_t* gn_src(const GnArgs& a, long item, long row, int o) {
    const int ch = o * 8;
    if (ch < a.c1) return a.x1 + (item * a.rows + row) * a.ld1 + ch;
    return a.x2 + (item * a.rows + row) * a.ld2 + (ch - a.c1);
}

__global__ void gn_stats_kernel(const GnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [rl][C][2]
    const int C = a.oc * 8;
    const int o = threadIdx.x % a.oc, rl = threadIdx.x / a.oc;
    const long item = blockIdx.y;
    const long per = (a.rows + a.nsplit - 1) / a.nsplit;
    const long r0 = (long)blockIdx.x * per;
    const long r1 = (r0 + per < a.rows) ? r0 + per : a.rows;
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
    // four independent 16-byte loads in flight per thread (a single load per iteration left the kernel at a third of the
    // HBM rate: bytes in flight, not bandwidth, was the limit); the rows are still accumulated in ascending order
    long r = r0 + rl;
    const long st = a.rl;
    for (; r + 3 * st < r1; r += 4 * st) {
        half8v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(gn_src(a, item, r + u * st, o));
        __builtin_amdgcn_sched_barrier(0);  // all four loads are issued before the first one is consumed
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = (float)v[u][j];
                s[j] += f;
                ss[j] += f * f;
            }
        }
    }
    for (; r < r1; r += st) {
        half8v v = *reinterpret_cast<const half8v*>(gn_src(a, item, r, o));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j];
            s[j] += f;
            ss[j] += f * f;
        }
    }
    float* mine = red + ((long)rl * C + o * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mine[2 * j] = s[j];
        mine[2 * j + 1] = ss[j];
    }
    __syncthreads();
    // per-channel totals of this block (row lanes folded in lane order), parked in row lane 0's slots
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float ts = 0.f, tss = 0.f;
            for (int k = 0; k < a.rl; ++k) {
                const float* q = red + ((long)k * C + o * 8 + j) * 2;
                ts += q[0];
                tss += q[1];
            }
            red[(o * 8 + j) * 2] = ts;   // the k = 0 slot of this very thread: read above, written only here
            red[(o * 8 + j) * 2 + 1] = tss;
        }
    }
    __syncthreads();
    // per-group partials of this (item, split): channels of a group in ascending order
    const int cpg = C / a.groups;
    float* gp = a.partial + (item * a.nsplit + blockIdx.x) * a.groups * 2;
    for (int gI = threadIdx.x; gI < a.groups; gI += blockDim.x) {
        float ts = 0.f, tss = 0.f;
        for (int c = gI * cpg; c < (gI + 1) * cpg; ++c) {
            ts += red[c * 2];
            tss += red[c * 2 + 1];
        }
        gp[gI * 2] = ts;
        gp[gI * 2 + 1] = tss;
    }
}

// one wave per (item, group): fold the splits' group partials in split order (lanes take splits lane, lane + 64, ...; the lane
// totals are then added by a fixed xor tree), in double; emit mean / rstd.  Contiguous 8-byte reads of [split][group][2] -- the
// round-1 form folded per-CHANNEL partials (strided 4-byte reads of a 40x larger array) and cost 8-10 us per launch.
// (A last-arriving-block fold inside gn_stats was measured in round 2: the agent-scope release each block needs writes back
// its XCD's dirty L2 lines -- the previous kernels' activations -- and doubled the statistics kernel: 11.4 -> 25.4 us.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnArgs a) {
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (item, group) of this wave
    const long total = (long)a.groups * a.n_items;
    if (pair >= total) return;
    const long item = pair / a.groups;
    const int gI = (int)(pair - item * a.groups);
    const float* pp = a.partial + item * a.nsplit * a.groups * 2 + gI * 2;
    double ts = 0.0, tss = 0.0;
    for (int k = lane; k < a.nsplit; k += 64) {
        const float* v = pp + (long)k * a.groups * 2;
        ts += (double)v[0];
        tss += (double)v[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ts += __shfl_xor(ts, o, 64);
        tss += __shfl_xor(tss, o, 64);
    }
    if (lane == 0) {
        const int cpg = a.oc * 8 / a.groups;
        const double n = (double)cpg * (double)a.rows;
        const double mean = ts / n;
        double var = tss / n - mean * mean;
        if (var < 0.0) var = 0.0;
        a.stat[(item * a.groups + gI) * 2] = (float)mean;
        a.stat[(item * a.groups + gI) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

// mean / rstd of one (item, group) from its folded sums (shared by the fold kernel and the fold inside the apply pass)
__device__ __forceinline__ void gn_emit_stat(double s, double ss, int cpg, long rows, float eps, float& mean_out, float& rstd_out) {
    const double n = (double)cpg * (double)rows;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out = (float)mean;
    rstd_out = (float)(1.0 / sqrt(var + (double)eps));
}

// statistics from the producers' column statistics: one block per (item, group) walks the [row tile][channel] pairs of the group
// (of either source; a group may straddle the two) with all of its loads independent, double per-thread sums, a fixed xor tree
// per wave and a fixed-order fold of the waves: bit-reproducible.  Pairs per block: 64 row tiles x 10 channels at level 0 per
// frame, 832 x 10 for a temporal norm (statistics over T*H*W) -> the launcher sizes the block to ~8 pairs per thread.
__global__ void gn_finalize_cs_kernel(const GnArgs a) {
    __shared__ double red[16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const long item = blockIdx.x / a.groups;
    const int gI = (int)(blockIdx.x - item * a.groups);
    const int cpg = a.oc * 8 / a.groups;
    const int c0 = gI * cpg;
    double ts = 0.0, tss = 0.0;
#pragma unroll
    for (int src = 0; src < 2; ++src) {
        const float* cs = src ? a.cs2 : a.cs1;
        const int cb = src ? a.c1 : 0, cn = src ? a.c2 : a.c1;  // channels [cb, cb + cn) of the concatenation
        const int lo = c0 > cb ? c0 : cb, hi = (c0 + cpg < cb + cn) ? c0 + cpg : cb + cn;
        if (!cs || lo >= hi) continue;
        const int rpt = src ? a.rpt2 : a.rpt1;
        const long tiles = a.rows / rpt;
        const int w = hi - lo;
        const float* base = cs + (item * tiles * cn + (lo - cb)) * 2;
        const long total = tiles * w;
        for (long k = tid; k < total; k += nthr) {
            const long tile = k / w;
            const int c = (int)(k - tile * w);
            const float2v v = *reinterpret_cast<const float2v*>(base + (tile * cn + c) * 2);
            ts += (double)v[0];
            tss += (double)v[1];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ts += __shfl_xor(ts, o, 64);
        tss += __shfl_xor(tss, o, 64);
    }
    if (lane == 0) {
        red[wave][0] = ts;
        red[wave][1] = tss;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0, ss = 0.0;
        for (int w2 = 0; w2 < (nthr >> 6); ++w2) {
            s += red[w2][0];
            ss += red[w2][1];
        }
        gn_emit_stat(s, ss, cpg, a.rows, a.eps, a.stat[(item * a.groups + gI) * 2], a.stat[(item * a.groups + gI) * 2 + 1]);
    }
}

// GroupNorm folded into the projection behind it (mv_groupnorm_cs_fold_linear_f16): one block per (item, 4 output rows).  Thread t owns
// the channel octet 8 t .. 8 t + 7 of each of its rows: w' = fp16(w gamma rstd) is stored, and the row's bias term
//   sum_c w beta_c  -  sum_c w' mean_{g(c)}        (the ROUNDED w': it cancels what the product of the raw rows accumulates)
// is folded in fp32 -- a fixed xor tree per wave, the waves in order: bit-reproducible -- and handed on as two fp16 halves.
struct GnFoldArgs {
    const float* stat;      // [items][groups][2] = mean, rstd
    const half_t* gamma;
    const half_t* beta;
    const half_t* w;        // [n_out][c]
    const half_t* bias;     // [n_out] or nullptr
    const half_t* rb_in;    // [items * rb_per_item][ldrb_in] or nullptr
    half_t* w_out;          // [items][n_out][c]
    half_t* rb_hi;          // [items * rb_per_item][n_out]
    half_t* rb_lo;
    int c, n_out, groups, ldrb_in, rb_per_item;
};

__global__ __launch_bounds__(256) void gn_fold_weights_kernel(const GnFoldArgs a) {
    __shared__ float red[4][4];   // [row of the block][wave]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long item = blockIdx.y;
    const int n0 = blockIdx.x * 4;
    const int oc = a.c >> 3, cpg = a.c / a.groups;
    const float* st = a.stat + item * a.groups * 2;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int o = tid; o < oc; o += 256) {
        const half8v gm = *reinterpret_cast<const half8v*>(a.gamma + 8 * o);
        const half8v bt = *reinterpret_cast<const half8v*>(a.beta + 8 * o);
        float sc[8], mu[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gI = (8 * o + j) / cpg;
            mu[j] = st[2 * gI];
            sc[j] = st[2 * gI + 1] * (float)gm[j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + r;
            if (n >= a.n_out) break;
            const half8v wv = *reinterpret_cast<const half8v*>(a.w + (long)n * a.c + 8 * o);
            half8v wo;
            float acc = part[r];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wf = (float)wv[j];
                wo[j] = (half_t)(wf * sc[j]);
                acc = fmaf(wf, (float)bt[j], acc);
                acc = fmaf(-(float)wo[j], mu[j], acc);
            }
            part[r] = acc;
            *reinterpret_cast<half8v*>(a.w_out + ((long)item * a.n_out + n) * a.c + 8 * o) = wo;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = part[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[r][wave] = v;
    }
    __syncthreads();
    // thread (r, j): row r of the block, bias row j of the item
    for (int idx = tid; idx < 4 * a.rb_per_item; idx += 256) {
        const int r = idx & 3, j = idx >> 2;
        const int n = n0 + r;
        if (n >= a.n_out) continue;
        float v = ((red[r][0] + red[r][1]) + red[r][2]) + red[r][3];
        if (a.bias) v += (float)a.bias[n];
        const long row = item * a.rb_per_item + j;
        if (a.rb_in) v += (float)a.rb_in[row * a.ldrb_in + n];
        const half_t hi = (half_t)v;
        a.rb_hi[row * a.n_out + n] = hi;
        a.rb_lo[row * a.n_out + n] = (half_t)(v - (float)hi);
    }
}

__global__ void gn_apply_kernel(const GnArgs a) {
    const int C = a.oc * 8;
    const int o = threadIdx.x % a.oc, rl = threadIdx.x / a.oc;
    const long item = blockIdx.y;
    const long per = (a.rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per;
    const long r1 = (r0 + per < a.rows) ? r0 + per : a.rows;
    float sc[8], sh[8];
    {
        const int cpg = C / a.groups;
        const half8v gm = *reinterpret_cast<const half8v*>(a.gamma + o * 8);
        const half8v bt = *reinterpret_cast<const half8v*>(a.beta + o * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gI = (o * 8 + j) / cpg;  // groups of 30 / 60 channels straddle octets
            const float* st = a.stat + (item * a.groups + gI) * 2;
            sc[j] = st[1] * (float)gm[j];
            sh[j] = (float)bt[j] - st[0] * sc[j];
        }
    }
    auto norm = [&](const half8v& v) __attribute__((always_inline)) {
        half8v w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * sc[j] + sh[j];
            if (a.silu) f = mv_silu(f);
            w[j] = (half_t)f;
        }
        return w;
    };
    long r = r0 + rl;
    const long st = a.rl;
    if (a.x1_lo != nullptr || a.y_lo != nullptr) {  // (block-uniform) the carried form: conv_norm_out only, one launch per forward
        for (; r < r1; r += st) {
            const long off = (item * a.rows + r) * a.ld1 + o * 8;
            const half8v v = *reinterpret_cast<const half8v*>(a.x1 + off);
            half8v l = half8v{0, 0, 0, 0, 0, 0, 0, 0};
            if (a.x1_lo) l = *reinterpret_cast<const half8v*>(a.x1_lo + off);
            half8v wh, wl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = ((float)v[j] + (float)l[j]) * sc[j] + sh[j];
                if (a.silu) f = mv_silu(f);
                wh[j] = (half_t)f;
                wl[j] = (half_t)(f - (float)wh[j]);
            }
            *reinterpret_cast<half8v*>(a.y + (item * a.rows + r) * a.ldy + o * 8) = wh;
            if (a.y_lo) *reinterpret_cast<half8v*>(a.y_lo + (item * a.rows + r) * a.ldy + o * 8) = wl;
        }
        return;
    }
    for (; r + 3 * st < r1; r += 4 * st) {  // four loads in flight per thread (see gn_stats_kernel)
        half8v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(gn_src(a, item, r + u * st, o));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<half8v*>(a.y + (item * a.rows + r + u * st) * a.ldy + o * 8) = norm(v[u]);
    }
    for (; r < r1; r += st)
        *reinterpret_cast<half8v*>(a.y + (item * a.rows + r) * a.ldy + o * 8) =
            norm(*reinterpret_cast<const half8v*>(gn_src(a, item, r, o)));
}

// ---- small tensors (the 16x16 / 8x8-latent levels, C = 1280 / 2560): ONE launch ---------------------------------------------------
// One block per (item, group) when the group is a whole number of 8-channel octets that does not straddle the concat seam and its
// slab (rows x C/groups channels) is small.  Thread (row lane, octet): a fixed octet of the group, rows rl, rl + rpp, ... -- no index
// arithmetic in the loops.  The block sums its slab (fp32, fixed order: per-thread partials over its rows, xor tree, waves in
// order), derives mean / rstd in double and normalises.  Slabs of <= KEEP rows per thread stay in registers between the two
// passes (one trip to memory); larger ones are re-read (L2).  The three-launch form costs ~25 us on these tensors (three kernel
// boundaries around ~3 us of streaming).
template <int NT, int KEEP>
__global__ __launch_bounds__(NT) void gn_small_kernel(const GnArgs a) {
    constexpr int NW = NT / 64;
    __shared__ float red[2][NW];
    __shared__ float stat[2];
    const int C = a.oc * 8;
    const int cpg = C / a.groups, opg = cpg >> 3;     // channels / octets per group
    const int gI = blockIdx.x % a.groups;
    const long item = blockIdx.x / a.groups;
    const int rpp = NT / opg;                          // rows per pass of the block
    const int rl = threadIdx.x / opg, o = gI * opg + (int)(threadIdx.x - rl * opg);
    const bool live = rl < rpp;
    const int rows = (int)a.rows;
    const bool second = o * 8 >= a.c1;
    const long ld = second ? a.ld2 : a.ld1;
    const half_t* src = (second ? a.x2 + (o * 8 - a.c1) : a.x1 + o * 8) + item * a.rows * ld;
    const bool in_regs = rows <= KEEP * rpp;           // block-uniform
    half8v keep[KEEP];
    float s = 0.f, ss = 0.f;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int row = rl + k * rpp;
            if (live && row < rows) keep[k] = *reinterpret_cast<const half8v*>(src + row * ld);
            else keep[k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)keep[k][j];
                s += f;
                ss += f * f;
            }
        }
    } else if (live) {
        for (int r0 = rl; r0 < rows; r0 += 4 * rpp) {
            half8v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + u * rpp;
                v[u] = row < rows ? *reinterpret_cast<const half8v*>(src + row * ld) : half8v{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)v[u][j];
                    s += f;
                    ss += f * f;
                }
            }
        }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s;
        red[1][threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < NW; ++w) {
            s1 += (double)red[0][w];
            s2 += (double)red[1][w];
        }
        const double n = (double)cpg * (double)a.rows;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    if (!live) return;
    const float mean = stat[0], rstd = stat[1];
    const half8v gm = *reinterpret_cast<const half8v*>(a.gamma + o * 8);
    const half8v bt = *reinterpret_cast<const half8v*>(a.beta + o * 8);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = rstd * (float)gm[j];
        sh[j] = (float)bt[j] - mean * sc[j];
    }
    half_t* dst = a.y + item * a.rows * a.ldy + o * 8;
    auto put = [&](const half8v v, int row) {
        half8v w;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j] * sc[j] + sh[j];
            if (a.silu) f = mv_silu(f);
            w[j] = (half_t)f;
        }
        *reinterpret_cast<half8v*>(dst + (long)row * a.ldy) = w;
    };
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int row = rl + k * rpp;
            if (row < rows) put(keep[k], row);
        }
    } else {
        for (int r0 = rl; r0 < rows; r0 += 4 * rpp) {
            half8v v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + u * rpp;
                v[u] = row < rows ? *reinterpret_cast<const half8v*>(src + row * ld) : half8v{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = r0 + u * rpp;
                if (row < rows) put(v[u], row);
            }
        }
    }
}

// ---- LayerNorm: one wave per group of R rows, up to 3 octets per lane and row (C <= 1536) ----
// All R rows' loads are issued before any is reduced: with one row per wave a lane had a single 16-byte load in flight
// (C = 320) and the kernel ran at half the HBM rate.  Per row the arithmetic (and its order) is unchanged.
// Round 3: the R rows of a wave are reduced TOGETHER -- R independent xor trees step by step, gamma / beta requested with the rows
// -- instead of one row after the other (each row's two dependent 6-step shuffle trees + its own gamma / beta loads made a wave's
// 8 rows a ~25 us serial chain: 34 us per level-0 launch = 2.0 TB/s, against 15.6 us for GroupNorm's apply pass over the same
// bytes; trace profiles/r03k).  Per row the arithmetic and its order are unchanged (bit-identical results).
template <int NO, int kLnRows>  // kLnRows = R rows per wave (the launcher's choice: 2, or 1 at C > 1024)
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* x, int ldx, half_t* y, int ldy, long rows, int c,
                                                        const half_t* gamma, const half_t* beta, float eps) {
    constexpr int R = kLnRows;
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int oc = c >> 3;
    half8v v[R][NO], gm[NO], bt[NO];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const long row = row0 + q;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            v[q][k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
            if (o < oc && row < rows) v[q][k] = *reinterpret_cast<const half8v*>(x + row * ldx + o * 8);
        }
    }
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const int o = lane + 64 * k;
        gm[k] = bt[k] = half8v{0, 0, 0, 0, 0, 0, 0, 0};
        if (o < oc) {
            gm[k] = *reinterpret_cast<const half8v*>(gamma + o * 8);
            bt[k] = *reinterpret_cast<const half8v*>(beta + o * 8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            if (lane + 64 * k < oc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (float)v[q][k][j];
            }
        }
        acc[q] = sum;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
    }
    float mean[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        mean[q] = acc[q] / (float)c;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            if (lane + 64 * k < oc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = (float)v[q][k][j] - mean[q];
                    sq += d * d;
                }
            }
        }
        acc[q] = sq;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const long row = row0 + q;
        if (row >= rows) break;  // wave-uniform
        const float rstd = rsqrtf(acc[q] / (float)c + eps);
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            if (o < oc) {
                half8v w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (half_t)(((float)v[q][k][j] - mean[q]) * rstd * (float)gm[k][j] + (float)bt[k][j]);
                *reinterpret_cast<half8v*>(y + row * ldy + o * 8) = w;
            }
        }
    }
}

}  // namespace

extern "C" int32_t mv_groupnorm_default_nsplit(int64_t n_items, int64_t rows, int32_t c) {
    // aim for ~1024 workgroups (4 per CU) in the stats / apply passes, each streaming at least ~32 rows
    long want = (1024 + n_items - 1) / n_items;
    long maxsplit = rows / 32;
    if (maxsplit < 1) maxsplit = 1;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    (void)c;
    return (int32_t)want;
}

extern "C" int64_t mv_groupnorm_partial_floats(int64_t n_items, int32_t num_groups, int32_t nsplit) {
    return n_items * (int64_t)nsplit * 2 * num_groups;
}

namespace {
// shared by mv_groupnorm_f16 (cs1 == nullptr: statistics by a pass over x) and mv_groupnorm_cs_f16
int gn_launch(const char* who, const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2, int64_t n_items,
              int64_t rows, int32_t num_groups, float eps, const void* gamma, const void* beta, int32_t silu, void* y, int32_t ldy,
              float* partial, int32_t nsplit, float* stat, const float* cs1, int32_t rpt1, const float* cs2, int32_t rpt2,
              const void* x1_lo, void* y_lo, void* stream) {
    MV_REQUIRE(x1 && y && gamma && beta && stat && (partial || cs1), "%s: null pointer", who);
    MV_REQUIRE((!x1_lo && !y_lo) || !x2, "%s: the carried form (x1_lo / y_lo) takes one source", who);
    MV_REQUIRE(((reinterpret_cast<uintptr_t>(x1_lo) | reinterpret_cast<uintptr_t>(y_lo)) & 15) == 0, "%s: x1_lo / y_lo must be 16-byte aligned", who);
    if (!x2) c2 = 0;
    const int C = c1 + c2;
    MV_REQUIRE(c1 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "%s: channels must be multiples of 8 (c1=%d c2=%d)", who, c1, c2);
    MV_REQUIRE(num_groups > 0 && C % num_groups == 0, "%s: C=%d not divisible by groups=%d", who, C, num_groups);
    MV_REQUIRE(ld1 % 8 == 0 && (c2 == 0 || ld2 % 8 == 0) && ldy % 8 == 0, "%s: leading dims must be multiples of 8", who);
    MV_REQUIRE(n_items > 0 && rows > 0 && nsplit > 0 && nsplit <= 65535 && n_items <= 65535, "%s: bad sizes", who);
    MV_REQUIRE((reinterpret_cast<uintptr_t>(gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(beta) & 15) == 0,
               "%s: gamma / beta must be 16-byte aligned", who);
    if (cs1) {
        MV_REQUIRE(rpt1 > 0 && rows % rpt1 == 0 && (reinterpret_cast<uintptr_t>(cs1) & 7) == 0,
                   "%s: rows=%ld is not a whole number of the producer's %d-row statistic tiles", who, (long)rows, rpt1);
        MV_REQUIRE(c2 == 0 || (cs2 && rpt2 > 0 && rows % rpt2 == 0 && (reinterpret_cast<uintptr_t>(cs2) & 7) == 0),
                   "%s: the second source needs column statistics too (rows %% rpt2 == 0)", who);
        MV_REQUIRE(n_items * num_groups <= 0x7fffffffL, "%s: too many (item, group) pairs", who);
    }
    const int oc = C / 8;
    MV_REQUIRE(oc <= 1024, "%s: C=%d too large", who, C);
    int rl = 256 / oc;
    if (rl < 1) rl = 1;
    const int bs = oc * rl;
    MV_REQUIRE(bs >= num_groups, "%s: C=%d too small for %d groups", who, C, num_groups);
    GnArgs a;
    a.x1 = (const half_t*)x1; a.x2 = (const half_t*)x2; a.c1 = c1; a.c2 = c2; a.ld1 = ld1; a.ld2 = ld2;
    a.rows = rows; a.nsplit = nsplit; a.oc = oc; a.rl = rl; a.partial = partial; a.stat = stat; a.n_items = n_items;
    a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta; a.y = (half_t*)y; a.ldy = ldy; a.silu = silu;
    a.groups = num_groups; a.eps = eps;
    a.cs1 = cs1; a.cs2 = c2 ? cs2 : nullptr; a.rpt1 = rpt1; a.rpt2 = rpt2;
    a.x1_lo = (const half_t*)x1_lo; a.y_lo = (half_t*)y_lo;
    hipStream_t s = (hipStream_t)stream;
    if (!x1_lo && !y_lo) {   // small slabs: one launch (see gn_small_kernel); the carried form always takes the apply pass
        const int cpg = C / num_groups;
        const long slab_bytes = rows * (long)cpg * 2;
        const int opg = cpg / 8;
        // measured (profiles/r02i): a slab that fits the block's registers (<= 8 rows per thread of a 1024-thread block) beats the
        // three launches; a larger one (the temporal norms of the 16x16 level: 2 items x 32 groups = 64 blocks re-reading 266 KB
        // each) does not -- 34 us against ~25 -- and keeps the three-launch form
        if (cpg % 8 == 0 && cpg <= 512 && c1 % cpg == 0 && rows <= 8L * (1024 / (opg > 0 ? opg : 1)) && slab_bytes <= 320 * 1024 &&
            n_items * num_groups <= 0x7fffffffL) {
            const dim3 grid((unsigned)(n_items * num_groups));
            if (rows <= 8L * (256 / opg)) hipLaunchKernelGGL((gn_small_kernel<256, 8>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gn_small_kernel<1024, 8>), grid, dim3(1024), 0, s, a);
            MV_CHECK_LAUNCH("mv_groupnorm_f16(small)");
            return MV_OK;
        }
    }
    const long pairs = (long)num_groups * n_items;
    if (cs1) {
        // ~8 (row tile, channel) pairs per thread
        const int cpg = C / num_groups;
        const long per_block = (rows / rpt1) * (long)cpg;
        const int threads = per_block <= 512 ? 64 : per_block <= 2048 ? 256 : 1024;
        hipLaunchKernelGGL(gn_finalize_cs_kernel, dim3((unsigned)pairs), dim3(threads), 0, s, a);
        MV_CHECK_LAUNCH("mv_groupnorm_cs_f16(fold)");
    } else {
        // LDS: [rl][C][2] floats = 64 bytes per thread for the row-lane fold
        hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, (unsigned)n_items), dim3(bs), (size_t)bs * 16 * sizeof(float), s, a);
        MV_CHECK_LAUNCH("mv_groupnorm_f16(stats)");
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, a);
        MV_CHECK_LAUNCH("mv_groupnorm_f16(finalize)");
    }
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nsplit, (unsigned)n_items), dim3(bs), 0, s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_f16(apply)");
    return MV_OK;
}
}  // namespace

extern "C" int mv_groupnorm_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                                int64_t n_items, int64_t rows, int32_t num_groups, float eps, const void* gamma,
                                const void* beta, int32_t silu, void* y, int32_t ldy, float* partial, int32_t nsplit,
                                float* stat, const void* x1_lo, void* y_lo, void* stream) {
    MV_REQUIRE(partial, "mv_groupnorm_f16: null pointer");
    return gn_launch("mv_groupnorm_f16", x1, x2, c1, c2, ld1, ld2, n_items, rows, num_groups, eps, gamma, beta, silu, y, ldy, partial,
                     nsplit, stat, nullptr, 0, nullptr, 0, x1_lo, y_lo, stream);
}

extern "C" int mv_groupnorm_cs_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                                   int64_t n_items, int64_t rows, int32_t num_groups, float eps, const void* gamma,
                                   const void* beta, int32_t silu, void* y, int32_t ldy, const float* cs1, int32_t rpt1,
                                   const float* cs2, int32_t rpt2, int32_t nsplit, float* stat, const void* x1_lo, void* y_lo, void* stream) {
    MV_REQUIRE(cs1, "mv_groupnorm_cs_f16: null column statistics");
    return gn_launch("mv_groupnorm_cs_f16", x1, x2, c1, c2, ld1, ld2, n_items, rows, num_groups, eps, gamma, beta, silu, y, ldy, nullptr,
                     nsplit, stat, cs1, rpt1, cs2, rpt2, x1_lo, y_lo, stream);
}

extern "C" int mv_groupnorm_cs_fold_linear_f16(const float* cs, int32_t rpt, int32_t c, int64_t n_items, int64_t rows, int32_t num_groups, float eps,
                                               const void* gamma, const void* beta, const void* w, const void* bias, int32_t n_out,
                                               const void* rb_in, int32_t ldrb_in, int32_t rb_per_item,
                                               void* w_out, void* rb_hi, void* rb_lo, float* stat, void* stream) {
    const char* who = "mv_groupnorm_cs_fold_linear_f16";
    MV_REQUIRE(cs && gamma && beta && w && w_out && rb_hi && rb_lo && stat, "%s: null pointer", who);
    MV_REQUIRE(c > 0 && c % 8 == 0 && c <= 2048 && num_groups > 0 && c % num_groups == 0, "%s: need C %% 8 == 0, C <= 2048, C %% groups == 0 (C=%d groups=%d)", who, c, num_groups);
    MV_REQUIRE(n_items > 0 && n_items <= 65535 && rows > 0 && n_out > 0 && rb_per_item >= 1 && rows % rb_per_item == 0, "%s: bad sizes", who);
    MV_REQUIRE(rpt > 0 && rows % rpt == 0 && (reinterpret_cast<uintptr_t>(cs) & 7) == 0, "%s: rows=%ld is not a whole number of the producer's %d-row statistic tiles", who, (long)rows, rpt);
    MV_REQUIRE(n_items * num_groups <= 0x7fffffffL && n_items * (long)n_out * c * 2 < 0x7fffffffL, "%s: problem too large", who);
    auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    MV_REQUIRE(al16(gamma) && al16(beta) && al16(w) && al16(w_out), "%s: gamma / beta / w / w_out must be 16-byte aligned", who);
    MV_REQUIRE(!rb_in || ldrb_in >= n_out, "%s: ldrb_in < n_out", who);
    hipStream_t s = (hipStream_t)stream;
    GnArgs a{};
    a.c1 = c; a.c2 = 0; a.rows = rows; a.oc = c / 8; a.stat = stat; a.n_items = n_items; a.groups = num_groups; a.eps = eps;
    a.cs1 = cs; a.cs2 = nullptr; a.rpt1 = rpt; a.rpt2 = 0;
    const long pairs = (long)num_groups * n_items;
    const long per_block = (rows / rpt) * (long)(c / num_groups);
    const int threads = per_block <= 512 ? 64 : per_block <= 2048 ? 256 : 1024;
    hipLaunchKernelGGL(gn_finalize_cs_kernel, dim3((unsigned)pairs), dim3(threads), 0, s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_cs_fold_linear_f16(fold)");
    GnFoldArgs f;
    f.stat = stat; f.gamma = (const half_t*)gamma; f.beta = (const half_t*)beta; f.w = (const half_t*)w; f.bias = (const half_t*)bias;
    f.rb_in = (const half_t*)rb_in; f.w_out = (half_t*)w_out; f.rb_hi = (half_t*)rb_hi; f.rb_lo = (half_t*)rb_lo;
    f.c = c; f.n_out = n_out; f.groups = num_groups; f.ldrb_in = ldrb_in; f.rb_per_item = rb_per_item;
    hipLaunchKernelGGL(gn_fold_weights_kernel, dim3((unsigned)((n_out + 3) / 4), (unsigned)n_items), dim3(256), 0, s, f);
    MV_CHECK_LAUNCH("mv_groupnorm_cs_fold_linear_f16(weights)");
    return MV_OK;
}

extern "C" int mv_layernorm_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t c,
                                const void* gamma, const void* beta, float eps, void* stream) {
    MV_REQUIRE(x && y && gamma && beta, "mv_layernorm_f16: null pointer");
    MV_REQUIRE(c > 0 && c % 8 == 0 && c <= 1536, "mv_layernorm_f16: need C %% 8 == 0 and C <= 1536 (C=%d)", c);
    MV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && rows > 0, "mv_layernorm_f16: bad leading dims / rows");
    hipStream_t s = (hipStream_t)stream;
    const int oc = c / 8;
    // rows per wave (4 waves per block).  Measured on the config-2 shapes, inputs cycled through > 256 MB of buffers and right behind
    // a producer (tools/gpu_ln_bench.py, profiles/r03s_ln_variants.log): with the rows of a wave reduced together, FEWER rows per wave
    // win -- more waves in flight beat more loads per lane -- 53 248 x 320: 8 rows 21.4 us, 4 rows 18.2, 2 rows 17.4, 1 row 19.0;
    // 13 312 x 640: 4 rows 12.4, 2 rows 10.4; 3 328 x 1280: 4 rows 7.9, 2 rows 6.6, 1 row 6.1.
    const int rpw = oc <= 128 ? 2 : 1;
    const unsigned grid = (unsigned)((rows + 4 * rpw - 1) / (4 * rpw));
    const half_t* xp = (const half_t*)x;
    half_t* yp = (half_t*)y;
    const half_t* g = (const half_t*)gamma;
    const half_t* b = (const half_t*)beta;
    if (oc <= 64) hipLaunchKernelGGL((layernorm_kernel<1, 2>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    else if (oc <= 128) hipLaunchKernelGGL((layernorm_kernel<2, 2>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    else hipLaunchKernelGGL((layernorm_kernel<3, 1>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    MV_CHECK_LAUNCH("mv_layernorm_f16");
    return MV_OK;
}

#ifdef MV_EXPERIMENT
// experiment builds only (tools/gpu_ln_bench.py): the one-shot kernel with `rows_per_wave` in {1, 2, 4, 8, 16} (0 = the product's choice).
// A persistent form (fixed grid of waves walking the row groups, next group's loads requested before the current one is reduced) was
// measured as well (profiles/r03r_ln_variants.log: equal or slower than the one-shot form at every size) and is not kept.
extern "C" int mv_layernorm_f16_var(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t c, const void* gamma,
                                    const void* beta, float eps, int32_t rows_per_wave, void* stream) {
    if (rows_per_wave == 0) return mv_layernorm_f16(x, ldx, y, ldy, rows, c, gamma, beta, eps, stream);
    hipStream_t s = (hipStream_t)stream;
    const int oc = c / 8, r = rows_per_wave;
    MV_REQUIRE(oc <= 192 && (r == 1 || r == 2 || r == 4 || ((r == 8 || r == 16) && oc <= 64)), "mv_layernorm_f16_var: unsupported rows per wave %d at C=%d", r, c);
    const half_t* xp = (const half_t*)x;
    half_t* yp = (half_t*)y;
    const half_t* g = (const half_t*)gamma;
    const half_t* b = (const half_t*)beta;
    const unsigned grid = (unsigned)((rows + 4 * r - 1) / (4 * r));
#define MV_LN_LAUNCH(NO, R) hipLaunchKernelGGL((layernorm_kernel<NO, R>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps)
    if (oc <= 64) {
        if (r == 1) MV_LN_LAUNCH(1, 1); else if (r == 2) MV_LN_LAUNCH(1, 2); else if (r == 4) MV_LN_LAUNCH(1, 4); else if (r == 8) MV_LN_LAUNCH(1, 8); else MV_LN_LAUNCH(1, 16);
    } else if (oc <= 128) {
        if (r == 1) MV_LN_LAUNCH(2, 1); else if (r == 2) MV_LN_LAUNCH(2, 2); else MV_LN_LAUNCH(2, 4);
    } else {
        if (r == 1) MV_LN_LAUNCH(3, 1); else if (r == 2) MV_LN_LAUNCH(3, 2); else MV_LN_LAUNCH(3, 4);
    }
#undef MV_LN_LAUNCH
    MV_CHECK_LAUNCH("mv_layernorm_f16_var");
    return MV_OK;
}
#endif
