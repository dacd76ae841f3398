// norm.hip -- GroupNorm(32)(+SiLU) over channels-last tensors and LayerNorm over the channel dim (gfx950).
//
// Both are HBM-bound streaming kernels: 16-byte (8 x fp16) vector loads, fp32 statistics, 64-lane
// shuffle reductions.  GroupNorm runs as two launches on one stream:
//   1. gn_stats : per (item, row-split): per-CHANNEL sum / sum-of-squares (fp32; per-channel first because on the up path the
//                 input is the channel concat of two tensors and groups of 30 / 60 channels straddle both the 8-channel vectors
//                 and the concat seam), folded to per-GROUP partials [item][split][group][2] inside the block.  The block that
//                 arrives LAST for its item (ticket counter, agent-scope release / acquire as cdna_hip_programming.md G16) folds
//                 the item's splits in split order, in double -- the result does not depend on which block is last -- and emits
//                 mean / rstd per (item, group).  (Round 1 ran this fold as a third launch: 166 launches and 1.4 ms per step,
//                 most of it strided reads of per-channel partials.)
//   2. gn_apply : y = silu?((x - mean) * rstd * gamma + beta), writing the concatenated [rows][c1+c2] tensor (this is the only
//                 place the up-path concat is ever materialised, already normalised).
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
    int* counter;        // [item] arrival tickets: zero on entry, zero on exit
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
    // ---- last-arriver fold (cdna_hip_programming.md Guideline 16, counter form): every storing wave drains its stores, one
    // lane releases at agent scope and takes a ticket; the block that draws nsplit - 1 acquires and folds ----
    int* flag = reinterpret_cast<int*>(red + (long)a.rl * C * 2);  // one word behind the fold area (same LDS object)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(a.counter + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (ticket == a.nsplit - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    // fold the item's splits: thread (part, g) sums splits part, part + P, ... in double; parts are then added in part order
    double* dred = reinterpret_cast<double*>(red);  // [P][groups][2] doubles: P * groups <= blockDim threads, 16 B each <= the fold area
    const int P = blockDim.x / a.groups;            // >= 1 (checked on the host)
    const int gI = threadIdx.x % a.groups, part = threadIdx.x / a.groups;
    if (part < P) {
        double ts = 0.0, tss = 0.0;
        const float* pp = a.partial + item * a.nsplit * a.groups * 2 + gI * 2;
        for (int k = part; k < a.nsplit; k += P) {
            ts += (double)pp[(long)k * a.groups * 2];
            tss += (double)pp[(long)k * a.groups * 2 + 1];
        }
        dred[(part * a.groups + gI) * 2] = ts;
        dred[(part * a.groups + gI) * 2 + 1] = tss;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.groups) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < P; ++k) {
            s1 += dred[(k * a.groups + threadIdx.x) * 2];
            s2 += dred[(k * a.groups + threadIdx.x) * 2 + 1];
        }
        const double n = (double)cpg * (double)a.rows;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        a.stat[(item * a.groups + threadIdx.x) * 2] = (float)mean;
        a.stat[(item * a.groups + threadIdx.x) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    if (threadIdx.x == 0) a.counter[item] = 0;  // the next call on this stream finds it zero
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
            const float* st = a.stat + (item * a.groups + (o * 8 + j) / cpg) * 2;  // groups of 30 / 60 channels straddle octets
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

extern "C" int64_t mv_groupnorm_partial_floats(int64_t n_items, int32_t num_groups, int32_t nsplit) {
    return n_items * (int64_t)nsplit * 2 * num_groups;
}

extern "C" int mv_groupnorm_f16(const void* x1, const void* x2, int32_t c1, int32_t c2, int32_t ld1, int32_t ld2,
                                int64_t n_items, int64_t rows, int32_t num_groups, float eps, const void* gamma,
                                const void* beta, int32_t silu, void* y, int32_t ldy, float* partial, int32_t nsplit,
                                float* stat, int32_t* counters, void* stream) {
    MV_REQUIRE(x1 && y && gamma && beta && partial && stat && counters, "mv_groupnorm_f16: null pointer");
    if (!x2) c2 = 0;
    const int C = c1 + c2;
    MV_REQUIRE(c1 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "mv_groupnorm_f16: channels must be multiples of 8 (c1=%d c2=%d)", c1, c2);
    MV_REQUIRE(num_groups > 0 && C % num_groups == 0, "mv_groupnorm_f16: C=%d not divisible by groups=%d", C, num_groups);
    MV_REQUIRE(ld1 % 8 == 0 && (c2 == 0 || ld2 % 8 == 0) && ldy % 8 == 0, "mv_groupnorm_f16: leading dims must be multiples of 8");
    MV_REQUIRE(n_items > 0 && rows > 0 && nsplit > 0 && nsplit <= 65535 && n_items <= 65535, "mv_groupnorm_f16: bad sizes");
    MV_REQUIRE((reinterpret_cast<uintptr_t>(gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(beta) & 15) == 0,
               "mv_groupnorm_f16: gamma / beta must be 16-byte aligned");
    const int oc = C / 8;
    MV_REQUIRE(oc <= 1024, "mv_groupnorm_f16: C=%d too large", C);
    int rl = 256 / oc;
    if (rl < 1) rl = 1;
    const int bs = oc * rl;
    MV_REQUIRE(bs >= num_groups, "mv_groupnorm_f16: C=%d too small for %d groups", C, num_groups);
    GnArgs a;
    a.x1 = (const half_t*)x1; a.x2 = (const half_t*)x2; a.c1 = c1; a.c2 = c2; a.ld1 = ld1; a.ld2 = ld2;
    a.rows = rows; a.nsplit = nsplit; a.oc = oc; a.rl = rl; a.partial = partial; a.stat = stat; a.counter = counters;
    a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta; a.y = (half_t*)y; a.ldy = ldy; a.silu = silu;
    a.groups = num_groups; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    // LDS: [rl][C][2] floats = 16 bytes per thread for the row-lane fold (reused as [P][groups][2] doubles by the last
    // arriver: P * groups <= bs threads x 16 bytes) + the "I am last" word behind it (ONE LDS object, G16 / section 5 trap (a))
    const size_t lds = (size_t)bs * 16 * sizeof(float) + 16;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, (unsigned)n_items), dim3(bs), lds, s, a);
    MV_CHECK_LAUNCH("mv_groupnorm_f16(stats)");
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
