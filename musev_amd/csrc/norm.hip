// norm.hip -- GroupNorm(32)(+SiLU) over channels-last tensors and LayerNorm over the channel dim (gfx950).
//
// Both are HBM-bound streaming kernels: 16-byte (8 x fp16) vector loads, fp32 statistics, 64-lane
// shuffle reductions.  GroupNorm runs as three launches on one stream:
//   1. gn_stats    : per (item, row-split) per-CHANNEL partial sum / sum-of-squares (fp32).  Per-channel, not
//                    per-group, because on the up path the input is the channel concat of two tensors and
//                    groups of 30 / 60 channels straddle both the 8-channel vectors and the concat seam.
//   2. gn_finalize : per (item, group): fold splits and the group's channels (double), emit per-channel
//                    scale = rstd*gamma and shift = beta - mean*rstd*gamma.
//   3. gn_apply    : y = silu?(x*scale + shift), writing the concatenated [rows][c1+c2] tensor (this is the only
//                    place the up-path concat is ever materialised, already normalised).
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
    float* partial;      // [item][split][2][C]
    float* scale_shift;  // [item][2][C]
    const half_t* gamma;
    const half_t* beta;
    half_t* y;
    int ldy;
    int silu;
    int groups;
    float eps;
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
    if (rl == 0) {
        float* out = a.partial + (item * a.nsplit + blockIdx.x) * 2 * C;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float ts = 0.f, tss = 0.f;
            for (int k = 0; k < a.rl; ++k) {
                const float* q = red + ((long)k * C + o * 8 + j) * 2;
                ts += q[0];
                tss += q[1];
            }
            out[o * 8 + j] = ts;
            out[C + o * 8 + j] = tss;
        }
    }
}

// one block per (item, group): fold the row-split partials of the group's channels (fixed order -> deterministic),
// reduce across the block in double, emit scale/shift for the group's channels.  (A single block per item folding
// nsplit x C partials serially was 24 % of the whole UNet step: 1024 splits x 2 items at the temporal layers.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnArgs a) {
    __shared__ double red[2][4];
    __shared__ float stat[2];
    const int C = a.oc * 8;
    const long item = blockIdx.y;
    const int gI = blockIdx.x;
    const int cpg = C / a.groups;
    const int c0 = gI * cpg;
    const float* pp = a.partial + item * a.nsplit * 2 * C;
    double ts = 0.0, tss = 0.0;
    const int total = a.nsplit * cpg;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int k = i / cpg, c = c0 + (i - k * cpg);
        ts += (double)pp[(long)k * 2 * C + c];
        tss += (double)pp[(long)k * 2 * C + C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ts += __shfl_xor(ts, o, 64);
        tss += __shfl_xor(tss, o, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = ts;
        red[1][wave] = tss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const double s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double n = (double)cpg * (double)a.rows;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    for (int c = c0 + threadIdx.x; c < c0 + cpg; c += 256) {
        const float sc = rstd * (float)a.gamma[c];
        a.scale_shift[item * 2 * C + c] = sc;
        a.scale_shift[item * 2 * C + C + c] = (float)a.beta[c] - mean * sc;
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
    const float* ssp = a.scale_shift + item * 2 * C + o * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = ssp[j];
        sh[j] = ssp[C + j];
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

// ---- LayerNorm: one wave per group of R rows, up to 3 octets per lane and row (C <= 1536) ----
// All R rows' loads are issued before any is reduced: with one row per wave a lane had a single 16-byte load in flight
// (C = 320) and the kernel ran at half the HBM rate.  Per row the arithmetic (and its order) is unchanged.
template <int NO, int kLnRows>  // kLnRows = R: 4 at C <= 512 (one octet per lane), else 2
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* x, int ldx, half_t* y, int ldy, long rows, int c,
                                                        const half_t* gamma, const half_t* beta, float eps) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * kLnRows;
    if (row0 >= rows) return;
    const int oc = c >> 3;
    half8v v[kLnRows][NO];
#pragma unroll
    for (int q = 0; q < kLnRows; ++q) {
        const long row = row0 + q;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            if (o < oc && row < rows) v[q][k] = *reinterpret_cast<const half8v*>(x + row * ldx + o * 8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < kLnRows; ++q) {
        const long row = row0 + q;
        if (row >= rows) break;  // wave-uniform
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            if (o < oc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (float)v[q][k][j];
            }
        }
        const float mean = wave_sum(sum) / (float)c;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            if (o < oc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float d = (float)v[q][k][j] - mean;
                    sq += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)c + eps);
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const int o = lane + 64 * k;
            if (o < oc) {
                half8v gm = *reinterpret_cast<const half8v*>(gamma + o * 8);
                half8v bt = *reinterpret_cast<const half8v*>(beta + o * 8);
                half8v w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (half_t)(((float)v[q][k][j] - mean) * rstd * (float)gm[j] + (float)bt[j]);
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

extern "C" int64_t mv_groupnorm_partial_floats(int64_t n_items, int32_t c, int32_t nsplit) {
    return n_items * (int64_t)nsplit * 2 * c;
}

extern "C" int mv_groupnorm_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                                int64_t n_items, int64_t rows, int32_t num_groups, float eps, const void* gamma,
                                const void* beta, int32_t silu, void* y, int32_t ldy, float* partial, int32_t nsplit,
                                float* scale_shift, void* stream) {
    MV_REQUIRE(x1 && y && gamma && beta && partial && scale_shift, "mv_groupnorm_f16: null pointer");
    if (!x2) c2 = 0;
    const int C = c1 + c2;
    MV_REQUIRE(c1 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "mv_groupnorm_f16: channels must be multiples of 8 (c1=%d c2=%d)", c1, c2);
    MV_REQUIRE(num_groups > 0 && C % num_groups == 0, "mv_groupnorm_f16: C=%d not divisible by groups=%d", C, num_groups);
    MV_REQUIRE(ld1 % 8 == 0 && (c2 == 0 || ld2 % 8 == 0) && ldy % 8 == 0, "mv_groupnorm_f16: leading dims must be multiples of 8");
    MV_REQUIRE(n_items > 0 && rows > 0 && nsplit > 0 && nsplit <= 65535 && n_items <= 65535, "mv_groupnorm_f16: bad sizes");
    const int oc = C / 8;
    MV_REQUIRE(oc <= 1024, "mv_groupnorm_f16: C=%d too large", C);
    int rl = 256 / oc;
    if (rl < 1) rl = 1;
    GnArgs a;
    a.x1 = (const half_t*)x1; a.x2 = (const half_t*)x2; a.c1 = c1; a.c2 = c2; a.ld1 = ld1; a.ld2 = ld2;
    a.rows = rows; a.nsplit = nsplit; a.oc = oc; a.rl = rl; a.partial = partial; a.scale_shift = scale_shift;
    a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta; a.y = (half_t*)y; a.ldy = ldy; a.silu = silu;
    a.groups = num_groups; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    const int bs = oc * rl;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, (unsigned)n_items), dim3(bs), (size_t)bs * 16 * sizeof(float), s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_f16(stats)");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)num_groups, (unsigned)n_items), dim3(256), 0, s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_f16(finalize)");
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nsplit, (unsigned)n_items), dim3(bs), 0, s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_f16(apply)");
    return MV_OK;
}

extern "C" int mv_layernorm_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t c,
                                const void* gamma, const void* beta, float eps, void* stream) {
    MV_REQUIRE(x && y && gamma && beta, "mv_layernorm_f16: null pointer");
    MV_REQUIRE(c > 0 && c % 8 == 0 && c <= 1536, "mv_layernorm_f16: need C %% 8 == 0 and C <= 1536 (C=%d)", c);
    MV_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && rows > 0, "mv_layernorm_f16: bad leading dims / rows");
    hipStream_t s = (hipStream_t)stream;
    const int oc = c / 8;
    const int rpw = oc <= 64 ? 4 : 2;  // rows per wave (4 waves per block)
    const unsigned grid = (unsigned)((rows + 4 * rpw - 1) / (4 * rpw));
    const half_t* xp = (const half_t*)x;
    half_t* yp = (half_t*)y;
    const half_t* g = (const half_t*)gamma;
    const half_t* b = (const half_t*)beta;
    if (oc <= 64) hipLaunchKernelGGL((layernorm_kernel<1, 4>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    else if (oc <= 128) hipLaunchKernelGGL((layernorm_kernel<2, 2>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    else hipLaunchKernelGGL((layernorm_kernel<3, 2>), dim3(grid), dim3(256), 0, s, xp, ldx, yp, ldy, (long)rows, c, g, b, eps);
    MV_CHECK_LAUNCH("mv_layernorm_f16");
    return MV_OK;
}
