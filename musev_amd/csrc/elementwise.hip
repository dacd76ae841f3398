// elementwise.hip -- HBM-bound helpers of the UNet3D step and the sliding-window denoise loop (gfx950).
// All kernels use 8/16-byte vector accesses on the channels-last layout and grid-stride loops.
#include "common.h"

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(long work_items) {
    long g = (work_items + kBlock - 1) / kBlock;
    if (g > 8192) g = 8192;  // grid-stride beyond ~32 blocks per CU
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- GEGLU: y[m, j] = x[m, j] * gelu(x[m, half + j]) ----
__global__ void geglu_kernel(const half_t* x, int ldx, half_t* y, int ldy, long rows, int half_cols) {
    const int oc = half_cols >> 3;
    const long total = rows * oc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / oc;
        const int o = (int)(i - r * oc);
        half8v a = *reinterpret_cast<const half8v*>(x + r * ldx + o * 8);
        half8v g = *reinterpret_cast<const half8v*>(x + r * ldx + half_cols + o * 8);
        half8v w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (half_t)((float)a[j] * mv_gelu((float)g[j]));
        *reinterpret_cast<half8v*>(y + r * ldy + o * 8) = w;
    }
}

__global__ void silu_kernel(const half_t* x, half_t* y, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        half8v a = reinterpret_cast<const half8v*>(x)[i];
        half8v w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (half_t)mv_silu((float)a[j]);
        reinterpret_cast<half8v*>(y)[i] = w;
    }
}

// y = a + b; with a_lo (the lo half of a carried residual-stream tensor, mv_gemm_desc.c_lo) the sum is (a + a_lo) + b in fp32 and is
// stored as two halves again: the ControlNet residual joins the unrounded skip instead of rounding it a second time
__global__ void add_kernel(const half_t* a, const half_t* a_lo, const half_t* b, half_t* y, half_t* y_lo, long n8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        half8v u = reinterpret_cast<const half8v*>(a)[i];
        half8v v = reinterpret_cast<const half8v*>(b)[i];
        half8v ul = half8v{0, 0, 0, 0, 0, 0, 0, 0};
        if (a_lo) ul = reinterpret_cast<const half8v*>(a_lo)[i];
        half8v w, wl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = ((float)u[j] + (float)ul[j]) + (float)v[j];
            w[j] = (half_t)sv;
            wl[j] = (half_t)(sv - (float)w[j]);
        }
        reinterpret_cast<half8v*>(y)[i] = w;
        if (y_lo) reinterpret_cast<half8v*>(y_lo)[i] = wl;
    }
}

__global__ void zero_rows_kernel(half_t* x, int ld, const int* row_idx, int n_idx, int cols) {
    const long total = (long)n_idx * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
        x[(long)row_idx[r] * ld + c] = (half_t)0.f;
    }
}

// nearest-neighbour resize of channels-last images to an EXPLICIT size (diffusers Upsample2D with output_size: F.interpolate(size=...,
// mode="nearest")): y[n, oy, ox, :] = x[n, min(floor(oy * sy), hin - 1), min(floor(ox * sx), win - 1), :] with sy = hin / hout, sx = win / wout
// as floats -- torch's own index rule (the scales are formed on the host in IEEE float, the products rounded once here).  One thread =
// 8 channels of one output pixel (16-byte loads / stores).  Only taken for latent sizes that are not multiples of 2^(number of
// upsamplers); the exact x 2 case stays fused into the convolution's gather.
__global__ void upsample_nearest_kernel(const half_t* x, int ldx, half_t* y, int ldy, long n_img, int hin, int win, int hout, int wout,
                                        int c8, float sy, float sx) {
    const long total = n_img * hout * wout * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8);
        const long pix = i / c8;
        const int ox = (int)(pix % wout);
        const long t = pix / wout;
        const int oy = (int)(t % hout);
        const long n = t / hout;
        int iy = (int)floorf((float)oy * sy), ix = (int)floorf((float)ox * sx);
        iy = iy < hin - 1 ? iy : hin - 1;
        ix = ix < win - 1 ? ix : win - 1;
        const half8v v = *reinterpret_cast<const half8v*>(x + ((n * hin + iy) * win + ix) * (long)ldx + 8 * cc);
        *reinterpret_cast<half8v*>(y + pix * (long)ldy + 8 * cc) = v;
    }
}

// sinusoidal embedding, flip_sin_to_cos=True, downscale_freq_shift=0: out = [cos(t f_k), sin(t f_k)], f_k = 10000^(-k/half)
__global__ void timestep_embedding_kernel(const float* t, int n, int dim, half_t* out) {
    const int half_dim = dim / 2;
    const int total = n * half_dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / half_dim, k = i - r * half_dim;
        const float freq = expf(-logf(10000.0f) * (float)k / (float)half_dim);
        const float arg = t[r] * freq;
        out[(long)r * dim + k] = (half_t)cosf(arg);
        out[(long)r * dim + half_dim + k] = (half_t)sinf(arg);
    }
}

// ---- layout: [B, C, T, HW] <-> [B, T, HW, C] (C small: latent channels) ----
template <typename SrcT>
__global__ void bcthw_to_bthwc_kernel(const SrcT* x, half_t* y, int b, int c, int t, int hw) {
    const long total = (long)b * t * hw * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int p = (int)(r % hw);
        r /= hw;
        const int ti = (int)(r % t);
        const int bi = (int)(r / t);
        y[i] = (half_t)(float)x[(((long)bi * c + ci) * t + ti) * hw + p];
    }
}
template <typename SrcT, typename DstT>
__global__ void bthwc_to_bcthw_kernel(const SrcT* x, DstT* y, int b, int c, int t, int hw) {
    const long total = (long)b * t * hw * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the destination [b][c][t][hw] so that stores are coalesced
        const int p = (int)(i % hw);
        long r = i / hw;
        const int ti = (int)(r % t);
        r /= t;
        const int ci = (int)(r % c);
        const int bi = (int)(r / c);
        y[i] = (DstT)(float)x[(((long)bi * t + ti) * hw + p) * c + ci];
    }
}

// ---- conv3x3 with tiny Cin (conv_in: 4 -> 320): one thread per (pixel, 8 output channels) ----
// weights [cout][3][3][cin] are staged in LDS as fp32 [tap*cin][cout] so lanes read consecutive channels.
__global__ void conv3x3_cin_small_kernel(const half_t* x, int cin, const half_t* w, const half_t* bias, const half_t* add,
                                         half_t* y, int cout, long n_img, int h, int wd) {
    extern __shared__ __attribute__((aligned(16))) float sw[];  // [9*cin][cout]
    const int kk = 9 * cin;
    for (int i = threadIdx.x; i < kk * cout; i += blockDim.x) {
        const int o = i / kk, k = i - o * kk;
        sw[k * cout + o] = (float)w[i];
    }
    __syncthreads();
    const int oc8 = cout >> 3;
    const long total = n_img * h * wd * oc8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o8 = (int)(i % oc8);
        const long pix = i / oc8;
        const int px = (int)(pix % wd);
        const long r = pix / wd;
        const int py = (int)(r % h);
        const long img = r / h;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias ? (float)bias[o8 * 8 + j] : 0.f;
        if (cin == 4) {  // the latent case: one 8-byte load per tap, all 9 issued before the arithmetic
            half4v xv[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < wd;
                xv[tap] = half4v{0, 0, 0, 0};
                if (ok) xv[tap] = *reinterpret_cast<const half4v*>(x + ((img * h + iy) * wd + ix) * 4);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    const float xf = (float)xv[tap][ci];
                    const float4v w0 = *reinterpret_cast<const float4v*>(sw + (tap * 4 + ci) * cout + o8 * 8);
                    const float4v w1 = *reinterpret_cast<const float4v*>(sw + (tap * 4 + ci) * cout + o8 * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[j] = fmaf(xf, w0[j], acc[j]);
                        acc[4 + j] = fmaf(xf, w1[j], acc[4 + j]);
                    }
                }
            }
        } else {
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                if (iy < 0 || iy >= h || ix < 0 || ix >= wd) continue;
                const half_t* src = x + ((img * h + iy) * wd + ix) * cin;
                for (int ci = 0; ci < cin; ++ci) {
                    const float xv = (float)src[ci];
                    const float* wp = sw + (tap * cin + ci) * cout + o8 * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wp[j], acc[j]);
                }
            }
        }
        half8v o;
        if (add) {
            half8v a = *reinterpret_cast<const half8v*>(add + pix * cout + o8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)a[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
        *reinterpret_cast<half8v*>(y + pix * cout + o8 * 8) = o;
    }
}

// ---- im2col for a 3x3 / pad 1 convolution with tiny Cin: y[pix][tap*cin + ci] = x[neighbour(tap)][ci], zero beyond 9*cin.
// conv_in (4 -> 320) then runs as ONE 64-deep K step of the MFMA GEMM instead of a VALU direct convolution.
__global__ void im2col3x3_kernel(const half_t* x, int cin, half_t* y, int kpad, long n_img, int h, int wd) {
    const int oc = kpad >> 3;
    const long total = n_img * h * wd * oc;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % oc);
        const long pix = i / oc;
        const int px = (int)(pix % wd);
        const long r = pix / wd;
        const int py = (int)(r % h);
        const long img = r / h;
        half8v v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = o * 8 + j;
            const int tap = col / cin, ci = col - tap * cin;
            half_t e = (half_t)0.f;
            if (tap < 9) {
                const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
                if (iy >= 0 && iy < h && ix >= 0 && ix < wd) e = x[((img * h + iy) * wd + ix) * cin + ci];
            }
            v[j] = e;
        }
        *reinterpret_cast<half8v*>(y + pix * kpad + o * 8) = v;
    }
}

// ---- conv3x3 with tiny Cout (conv_out: 320 -> 4): one wave per output pixel, lanes split the 9*Cin reduction ----
template <typename OutT>
__global__ __launch_bounds__(256) void conv3x3_cout_small_kernel(const half_t* x, const half_t* x_lo, int cin, const half_t* w,
                                                                 const half_t* bias, OutT* y, int cout, long n_img,
                                                                 int h, int wd) {
    extern __shared__ __attribute__((aligned(16))) half_t swh[];  // [cout][9*cin]
    const int kk = 9 * cin;
    for (int i = threadIdx.x; i < cout * kk / 8; i += blockDim.x)
        reinterpret_cast<uint4*>(swh)[i] = reinterpret_cast<const uint4*>(w)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int oc = cin >> 3;  // octets per tap
    const long npix = n_img * h * wd;
    for (long pix = (long)blockIdx.x * 4 + wave; pix < npix; pix += (long)gridDim.x * 4) {
        const int px = (int)(pix % wd);
        const long r = pix / wd;
        const int py = (int)(r % h);
        const long img = r / h;
        float acc[8];  // cout <= 8
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int e = lane; e < 9 * oc; e += 64) {
            const int tap = e / oc, o = e - tap * oc;
            const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
            if (iy < 0 || iy >= h || ix < 0 || ix >= wd) continue;
            const long xoff = ((img * h + iy) * wd + ix) * cin + o * 8;
            half8v xv = *reinterpret_cast<const half8v*>(x + xoff);
            float xf[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xf[q] = (float)xv[q];
            if (x_lo) {  // (uniform) two-fp16 input: hi + lo in fp32
                const half8v xl = *reinterpret_cast<const half8v*>(x_lo + xoff);
#pragma unroll
                for (int q = 0; q < 8; ++q) xf[q] += (float)xl[q];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < cout) {
                    half8v wv = *reinterpret_cast<const half8v*>(swh + j * kk + tap * cin + o * 8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[j] = fmaf(xf[q], (float)wv[q], acc[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = wave_sum(acc[j]);
        if (lane < cout) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (lane == j) v = acc[j];
            if (bias) v += (float)bias[lane];
            y[pix * cout + lane] = (OutT)v;
        }
    }
}

// ---- sliding-window loop glue ----
// out: channels-last fp16 [copies][n_cond + win][hw][c]; frame f < n_cond from cond[c][n_cond][hw], else latents[:, idx[f-n_cond]].
// cond_slot (optional, [n_cond]): the window slot of every condition frame (the reference's vision_condition_latent_index, written
// FIRST into a zero tensor and then overwritten by the window's frames at n_cond.., pipeline_controlnet.py:1939-1946): slot f <
// n_cond holds cond frame k where cond_slot[k] == f, zeros where no k does; slots >= n_cond always hold the window's frames.
__global__ void window_gather_kernel(const float* latents, const float* cond, const int* idx, const int* cond_slot, int win, int n_cond,
                                     int c, int t_total, int hw, int copies, int hi_lo, half_t* out) {
    const int tw = n_cond + win;
    const long per = (long)tw * hw * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int p = (int)(r % hw);
        const int f = (int)(r / hw);
        float v;
        if (f < n_cond) {
            int k = f;
            if (cond_slot) {
                k = -1;
                for (int q = 0; q < n_cond; ++q)
                    if (cond_slot[q] == f) k = q;
            }
            v = k < 0 ? 0.f : cond[((long)ci * n_cond + k) * hw + p];
        } else {
            v = latents[((long)ci * t_total + idx[f - n_cond]) * hw + p];
        }
        const half_t hv = (half_t)v;
        if (hi_lo) {  // rows of 2 c columns: [hi | lo]
            const half_t lv = (half_t)(v - (float)hv);
            const long ro = r * (2 * c) + ci;
            for (int k = 0; k < copies; ++k) {
                out[k * 2 * per + ro] = hv;
                out[k * 2 * per + ro + c] = lv;
            }
            continue;
        }
        for (int k = 0; k < copies; ++k) out[k * per + i] = hv;
    }
}

// eps_acc[half_offset + k][c][t_total][hw] += eps_win[k][n_cond + j][hw][c]  for window frame j -> idx[j]
template <typename EpsT>
__global__ void window_scatter_add_kernel(const EpsT* eps_win, const int* idx, int win, int n_cond, int c, int t_total,
                                          int hw, int halves, int half_offset, float* eps_acc, float* counter,
                                          int add_counter) {
    const long total = (long)halves * c * win * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        long r = i / hw;
        const int j = (int)(r % win);
        r /= win;
        const int ci = (int)(r % c);
        const int k = (int)(r / c);
        const float v = (float)eps_win[(((long)k * (n_cond + win) + n_cond + j) * hw + p) * c + ci];
        eps_acc[(((long)(half_offset + k) * c + ci) * t_total + idx[j]) * hw + p] += v;
    }
    if (add_counter && blockIdx.x == 0 && threadIdx.x < win) counter[idx[threadIdx.x]] += 1.0f;
}

// Row softmax in place over fp16 rows (the single-head, d = 512 mid-block attention of the VAE decoder is run as GEMM -> this ->
// GEMM: once per decoded frame, outside the denoise loop).  One 256-thread block per row, the row held in registers (cols <= 16384),
// fp32 maximum / sum, exp2 with the log2(e) factor folded.
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* x, long ld, int cols) {
    __shared__ float red[8];
    half_t* row = x + (long)blockIdx.x * ld;
    const int oc = cols >> 3;
    constexpr int MAXV = 8;  // 8 octets per thread x 256 threads x 8 = 16384 columns
    half8v v[MAXV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int o = threadIdx.x + 256 * k;
        if (o < oc) {
            v[k] = *reinterpret_cast<const half8v*>(row + o * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[k][j]);
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float e[MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int o = threadIdx.x + 256 * k;
        if (o < oc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                e[k][j] = __builtin_amdgcn_exp2f(((float)v[k][j] - mx) * 1.4426950408889634f);
                sum += e[k][j];
            }
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int o = threadIdx.x + 256 * k;
        if (o < oc) {
            half8v w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (half_t)(e[k][j] * inv);
            *reinterpret_cast<half8v*>(row + o * 8) = w;
        }
    }
}

// Multi-rank update of one denoise step: eps_acc[h][ci][f][p] = sum over the units that cover frame f (window order) of the
// gathered predictions  units[slot][j * hw + p][ci]  (fp32, channels-last rows; slot = rank * max_units + k).  A GATHER over a host
// table (up to `maxc` (slot, j) pairs per (half, frame), -1 = none) instead of world x max_units scatter-add launches: one launch,
// no read-modify-write on eps_acc (no zero fill either), and a summation order fixed by the table -> bit-identical on every rank.
__global__ void window_units_reduce_kernel(const float* units, long unit_stride, const int* tab, int maxc, int c, int t_total, int hw,
                                           int halves, float* eps_acc) {
    const long total = (long)halves * t_total * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const long r = i / hw;
        const int f = (int)(r % t_total);
        const int h = (int)(r / t_total);
        const int* e = tab + ((long)h * t_total + f) * maxc * 2;
        for (int ci = 0; ci < c; ++ci) {
            float acc = 0.f;
            for (int k = 0; k < maxc; ++k) {
                const int slot = e[2 * k];
                if (slot < 0) break;
                acc += units[(long)slot * unit_stride + ((long)e[2 * k + 1] * hw + p) * c + ci];
            }
            eps_acc[(((long)h * c + ci) * t_total + f) * hw + p] = acc;
        }
    }
}

__global__ void cfg_ddim_step_kernel(float* latents, const float* eps_acc, const float* counter, int c, int t_total, int hw,
                                     int halves, float guidance, float sqrt_at, float sqrt_1mat, float sqrt_ap,
                                     float sqrt_1map) {
    const long n = (long)c * t_total * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tf = (int)((i / hw) % t_total);
        const float inv = 1.0f / counter[tf];
        float eps = eps_acc[i] * inv;
        if (halves == 2) {
            const float et = eps_acc[n + i] * inv;
            eps = eps + guidance * (et - eps);
        }
        const float x = latents[i];
        const float x0 = (x - sqrt_1mat * eps) / sqrt_at;
        latents[i] = sqrt_ap * x0 + sqrt_1map * eps;
    }
}

// generic one-step scheduler update after CFG: x <- cx * x + ce * eps   (Euler-discrete: cx = 1, ce = sigma_next - sigma)
__global__ void cfg_affine_step_kernel(float* latents, const float* eps_acc, const float* counter, int c, int t_total, int hw,
                                       int halves, float guidance, float cx, float ce) {
    const long n = (long)c * t_total * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int tf = (int)((i / hw) % t_total);
        const float inv = 1.0f / counter[tf];
        float eps = eps_acc[i] * inv;
        if (halves == 2) {
            const float et = eps_acc[n + i] * inv;
            eps = eps + guidance * (et - eps);
        }
        latents[i] = cx * latents[i] + ce * eps;
    }
}

// conv weight repack: [O][I][taps] -> [O][taps][I] fp16
template <typename SrcT>
__global__ void pack_conv_weight_kernel(const SrcT* w, half_t* out, int o, int ic, int taps) {
    const long total = (long)o * ic * taps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % ic);
        long r = i / ic;
        const int tp = (int)(r % taps);
        const long oo = r / taps;
        out[i] = (half_t)(float)w[(oo * ic + ci) * taps + tp];
    }
}

// ds_read_b64_tr_b16 semantics probe: lane l reads through the transpose path at byte offset addr[l]
__global__ void probe_tr16_kernel(const short* image, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = image[i];
    __syncthreads();
    const int l = threadIdx.x;
    // same addressing the attention kernel uses with row stride 64 shorts: 16-lane group g, lane i -> row 4g + i/4, col (i%4)*4
    const int l15 = l & 15, g = l >> 4;
    const short* p = lds + (4 * g + (l15 >> 2)) * 64 + (l15 & 3) * 4;
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    out[l * 4 + 0] = v[0];
    out[l * 4 + 1] = v[1];
    out[l * 4 + 2] = v[2];
    out[l * 4 + 3] = v[3];
}

// Direct 3x3 convolution (padding 1, stride 1 | 2) for the small-channel conv stacks either side of the hot path -- the
// PoseGuider (3 -> 16 -> 16 -> 32 -> 32 -> 96 -> 96 -> 256 -> 320 from 512^2 down to 64^2, once per call): channels-last
// fp16, fp32 accumulation, fused bias (+ SiLU).  One thread = one output pixel x 8 output channels; the 8 x (9 cin) weight
// slab of the block's channel octet sits in LDS and is read as a broadcast (every lane the same address).  VALU FMAs on
// purpose: 15 GFLOP per 512^2 frame, outside the per-step path, and Cin = 3 / 16 / 32 / 96 do not tile the MFMA K = 32.
// [cpu-sim:begin conv3x3_direct]  (tests/cpu_sim: this kernel is also compiled for the host and run thread-per-thread)
template <bool VEC>  // VEC: cin % 8 == 0 -> 16-byte input and weight reads
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const half_t* __restrict__ x, int cin, const half_t* __restrict__ w,
                                                             const half_t* __restrict__ bias, half_t* __restrict__ y, int cout,
                                                             long n_img, int h, int w_, int ho, int wo, int stride, int act) {
    extern __shared__ __attribute__((aligned(16))) half_t dsw[];  // [8][9 * cin], row = channel inside the octet
    const int oc0 = blockIdx.y * 8;
    const int K = 9 * cin;
    for (int i = threadIdx.x; i < 8 * K; i += 256) {
        const int r = i / K, k = i - r * K;
        dsw[i] = (oc0 + r < cout) ? w[(long)(oc0 + r) * K + k] : (half_t)0.0f;
    }
    __syncthreads();
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= n_img * ho * wo) return;
    const int ox = (int)(pix % wo);
    const long rest = pix / wo;
    const int oy = (int)(rest % ho);
    const long n = rest / ho;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride + ky - 1;
        if (iy < 0 || iy >= h) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride + kx - 1;
            if (ix < 0 || ix >= w_) continue;
            const half_t* xp = x + ((n * h + iy) * (long)w_ + ix) * cin;
            const half_t* wp = dsw + (ky * 3 + kx) * cin;
            if constexpr (VEC) {
                for (int c = 0; c < cin; c += 8) {
                    const half8v v = *reinterpret_cast<const half8v*>(xp + c);
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const half8v wv = *reinterpret_cast<const half8v*>(wp + r * K + c);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[r] = fmaf((float)v[j], (float)wv[j], acc[r]);
                    }
                }
            } else {
                for (int c = 0; c < cin; ++c) {
                    const float v = (float)xp[c];
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = fmaf(v, (float)wp[r * K + c], acc[r]);
                }
            }
        }
    }
    half8v o;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float v = acc[r];
        if (bias && oc0 + r < cout) v += (float)bias[oc0 + r];
        if (act == MV_ACT_SILU) v = mv_silu(v);
        o[r] = (half_t)v;
    }
    *reinterpret_cast<half8v*>(y + pix * cout + oc0) = o;  // cout % 8 == 0 (checked on the host)
}
// [cpu-sim:end conv3x3_direct]

}  // namespace

extern "C" int mv_geglu_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t rows, int32_t half_cols, void* stream) {
    MV_REQUIRE(x && y && rows > 0 && half_cols > 0 && half_cols % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "mv_geglu_f16: bad args");
    hipLaunchKernelGGL(geglu_kernel, dim3(grid_for(rows * (half_cols / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                       (const half_t*)x, ldx, (half_t*)y, ldy, (long)rows, half_cols);
    MV_CHECK_LAUNCH("mv_geglu_f16");
    return MV_OK;
}

extern "C" int mv_silu_f16(const void* x, void* y, int64_t n, void* stream) {
    MV_REQUIRE(x && y && n > 0 && n % 8 == 0, "mv_silu_f16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n / 8)), dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, (long)(n / 8));
    MV_CHECK_LAUNCH("mv_silu_f16");
    return MV_OK;
}

extern "C" int mv_add_f16(const void* a, const void* a_lo, const void* b, void* y, void* y_lo, int64_t n, void* stream) {
    MV_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "mv_add_f16: n must be a positive multiple of 8");
    MV_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(a_lo) |
                 reinterpret_cast<uintptr_t>(y_lo)) & 15) == 0, "mv_add_f16: 16-byte aligned pointers");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)a, (const half_t*)a_lo, (const half_t*)b,
                       (half_t*)y, (half_t*)y_lo, (long)(n / 8));
    MV_CHECK_LAUNCH("mv_add_f16");
    return MV_OK;
}

extern "C" int mv_upsample_nearest_f16(const void* x, int32_t ldx, void* y, int32_t ldy, int64_t n_img, int32_t hin, int32_t win, int32_t hout,
                                       int32_t wout, int32_t c, void* stream) {
    MV_REQUIRE(x && y && n_img > 0 && hin > 0 && win > 0 && hout > 0 && wout > 0 && c > 0 && c % 8 == 0 && ldx >= c && ldy >= c && ldx % 8 == 0 && ldy % 8 == 0,
               "mv_upsample_nearest_f16: bad args (channels and leading dimensions in multiples of 8)");
    MV_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "mv_upsample_nearest_f16: 16-byte aligned pointers");
    const float sy = (float)hin / (float)hout, sx = (float)win / (float)wout;   // torch: compute_scales_value<float>(nullopt, in, out)
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(grid_for(n_img * hout * wout * (c / 8))), dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)x, ldx,
                       (half_t*)y, ldy, (long)n_img, hin, win, hout, wout, c / 8, sy, sx);
    MV_CHECK_LAUNCH("mv_upsample_nearest_f16");
    return MV_OK;
}

extern "C" int mv_zero_rows_f16(void* x, int32_t ld, const int32_t* row_idx, int32_t n_idx, int32_t cols, void* stream) {
    MV_REQUIRE(x && row_idx && n_idx > 0 && cols > 0, "mv_zero_rows_f16: bad args");
    hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for((long)n_idx * cols)), dim3(kBlock), 0, (hipStream_t)stream, (half_t*)x, ld, row_idx, n_idx, cols);
    MV_CHECK_LAUNCH("mv_zero_rows_f16");
    return MV_OK;
}

extern "C" int mv_timestep_embedding_f16(const float* t, int32_t n, int32_t dim, void* out, void* stream) {
    MV_REQUIRE(t && out && n > 0 && dim > 0 && dim % 2 == 0, "mv_timestep_embedding_f16: bad args");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long)n * dim / 2)), dim3(kBlock), 0, (hipStream_t)stream, t, n, dim, (half_t*)out);
    MV_CHECK_LAUNCH("mv_timestep_embedding_f16");
    return MV_OK;
}

extern "C" int mv_bcthw_to_bthwc_f16(const void* x, int32_t x_is_f32, void* y, int32_t b, int32_t c, int32_t t, int32_t hw, void* stream) {
    MV_REQUIRE(x && y && b > 0 && c > 0 && t > 0 && hw > 0, "mv_bcthw_to_bthwc_f16: bad args");
    const long n = (long)b * c * t * hw;
    if (x_is_f32) hipLaunchKernelGGL(bcthw_to_bthwc_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (const float*)x, (half_t*)y, b, c, t, hw);
    else hipLaunchKernelGGL(bcthw_to_bthwc_kernel<half_t>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)x, (half_t*)y, b, c, t, hw);
    MV_CHECK_LAUNCH("mv_bcthw_to_bthwc_f16");
    return MV_OK;
}

extern "C" int mv_bthwc_to_bcthw_f16(const void* x, int32_t x_is_f32, void* y, int32_t y_is_f32, int32_t b, int32_t c, int32_t t,
                                     int32_t hw, void* stream) {
    MV_REQUIRE(x && y && b > 0 && c > 0 && t > 0 && hw > 0, "mv_bthwc_to_bcthw_f16: bad args");
    const long n = (long)b * c * t * hw;
    const dim3 g(grid_for(n)), blk(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (x_is_f32 && y_is_f32) hipLaunchKernelGGL((bthwc_to_bcthw_kernel<float, float>), g, blk, 0, s, (const float*)x, (float*)y, b, c, t, hw);
    else if (x_is_f32) hipLaunchKernelGGL((bthwc_to_bcthw_kernel<float, half_t>), g, blk, 0, s, (const float*)x, (half_t*)y, b, c, t, hw);
    else if (y_is_f32) hipLaunchKernelGGL((bthwc_to_bcthw_kernel<half_t, float>), g, blk, 0, s, (const half_t*)x, (float*)y, b, c, t, hw);
    else hipLaunchKernelGGL((bthwc_to_bcthw_kernel<half_t, half_t>), g, blk, 0, s, (const half_t*)x, (half_t*)y, b, c, t, hw);
    MV_CHECK_LAUNCH("mv_bthwc_to_bcthw_f16");
    return MV_OK;
}

extern "C" int mv_conv3x3_cin_small_f16(const void* x, int32_t cin, const void* w, const void* bias, const void* add, void* y,
                                        int32_t cout, int64_t n_img, int32_t h, int32_t w_, void* stream) {
    MV_REQUIRE(x && w && y && cin > 0 && cin <= 16 && cout % 8 == 0 && n_img > 0 && h > 0 && w_ > 0, "mv_conv3x3_cin_small_f16: bad args (cin=%d cout=%d)", cin, cout);
    const size_t smem = (size_t)9 * cin * cout * sizeof(float);
    MV_REQUIRE(smem <= 64 * 1024, "mv_conv3x3_cin_small_f16: weights do not fit LDS");
    // persistent-ish grid: every block first converts the whole weight tensor into LDS, so few, long-lived blocks
    long gcin = (n_img * h * w_ * (cout / 8) + kBlock - 1) / kBlock;
    if (gcin > 768) gcin = 768;  // 3 blocks per CU (46 KB of LDS each at 4 -> 320)
    hipLaunchKernelGGL(conv3x3_cin_small_kernel, dim3((unsigned)gcin), dim3(kBlock), smem, (hipStream_t)stream,
                       (const half_t*)x, cin, (const half_t*)w, (const half_t*)bias, (const half_t*)add, (half_t*)y, cout, (long)n_img, h, w_);
    MV_CHECK_LAUNCH("mv_conv3x3_cin_small_f16");
    return MV_OK;
}

extern "C" int mv_conv3x3_direct_f16(const void* x, int32_t cin, const void* w, const void* bias, void* y, int32_t cout,
                                     int64_t n_img, int32_t h, int32_t w_, int32_t stride, int32_t act, void* stream) {
    MV_REQUIRE(x && w && y && cin > 0 && cin <= 512 && cout > 0 && cout % 8 == 0 && n_img > 0 && h > 0 && w_ > 0,
               "mv_conv3x3_direct_f16: bad args (cin=%d cout=%d)", cin, cout);
    MV_REQUIRE((stride == 1 || stride == 2) && (act == MV_ACT_NONE || act == MV_ACT_SILU), "mv_conv3x3_direct_f16: stride %d / act %d", stride, act);
    const int ho = (h + 2 - 3) / stride + 1, wo = (w_ + 2 - 3) / stride + 1;
    const long blocks = (n_img * ho * wo + 255) / 256;
    MV_REQUIRE(blocks < (1L << 31) && cout / 8 <= 65535, "mv_conv3x3_direct_f16: grid too large");
    const size_t smem = (size_t)8 * 9 * cin * sizeof(half_t);  // <= 72 KB at cin = 512; 36 KB at the PoseGuider's 256
    MV_REQUIRE(smem <= 64 * 1024, "mv_conv3x3_direct_f16: weight slab does not fit LDS (cin=%d)", cin);
    dim3 grid((unsigned)blocks, (unsigned)(cout / 8));
    if (cin % 8 == 0)
        hipLaunchKernelGGL(conv3x3_direct_kernel<true>, grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)x, cin,
                           (const half_t*)w, (const half_t*)bias, (half_t*)y, cout, (long)n_img, h, w_, ho, wo, stride, act);
    else
        hipLaunchKernelGGL(conv3x3_direct_kernel<false>, grid, dim3(256), smem, (hipStream_t)stream, (const half_t*)x, cin,
                           (const half_t*)w, (const half_t*)bias, (half_t*)y, cout, (long)n_img, h, w_, ho, wo, stride, act);
    MV_CHECK_LAUNCH("mv_conv3x3_direct_f16");
    return MV_OK;
}

extern "C" int mv_im2col3x3_f16(const void* x, int32_t cin, void* y, int32_t kpad, int64_t n_img, int32_t h, int32_t w_, void* stream) {
    MV_REQUIRE(x && y && cin > 0 && kpad % 8 == 0 && kpad >= 9 * cin && n_img > 0 && h > 0 && w_ > 0, "mv_im2col3x3_f16: bad args (cin=%d kpad=%d)", cin, kpad);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(n_img * h * w_ * (kpad / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                       (const half_t*)x, cin, (half_t*)y, kpad, (long)n_img, h, w_);
    MV_CHECK_LAUNCH("mv_im2col3x3_f16");
    return MV_OK;
}

extern "C" int mv_conv3x3_cout_small_f16(const void* x, const void* x_lo, int32_t cin, const void* w, const void* bias, void* y, int32_t y_is_f32,
                                         int32_t cout, int64_t n_img, int32_t h, int32_t w_, void* stream) {
    MV_REQUIRE(x && w && y && cin % 8 == 0 && cout > 0 && cout <= 8 && n_img > 0 && h > 0 && w_ > 0, "mv_conv3x3_cout_small_f16: bad args (cin=%d cout=%d)", cin, cout);
    const size_t smem = (size_t)9 * cin * cout * sizeof(half_t);
    MV_REQUIRE(smem <= 64 * 1024, "mv_conv3x3_cout_small_f16: weights do not fit LDS");
    long g = (n_img * h * w_ + 3) / 4;
    if (g > 4096) g = 4096;
    if (y_is_f32)
        hipLaunchKernelGGL(conv3x3_cout_small_kernel<float>, dim3((unsigned)g), dim3(kBlock), smem, (hipStream_t)stream,
                           (const half_t*)x, (const half_t*)x_lo, cin, (const half_t*)w, (const half_t*)bias, (float*)y, cout, (long)n_img, h, w_);
    else
        hipLaunchKernelGGL(conv3x3_cout_small_kernel<half_t>, dim3((unsigned)g), dim3(kBlock), smem, (hipStream_t)stream,
                           (const half_t*)x, (const half_t*)x_lo, cin, (const half_t*)w, (const half_t*)bias, (half_t*)y, cout, (long)n_img, h, w_);
    MV_CHECK_LAUNCH("mv_conv3x3_cout_small_f16");
    return MV_OK;
}

extern "C" int mv_window_gather(const float* latents, const float* cond, const int32_t* idx, const int32_t* cond_slot, int32_t win,
                                int32_t n_cond, int32_t c, int32_t t_total, int32_t hw, int32_t cfg_copies, int32_t hi_lo, void* out,
                                void* stream) {
    MV_REQUIRE(latents && idx && out && win > 0 && n_cond >= 0 && (n_cond == 0 || cond) && c > 0 && t_total > 0 && hw > 0 && cfg_copies > 0,
               "mv_window_gather: bad args");
    hipLaunchKernelGGL(window_gather_kernel, dim3(grid_for((long)(n_cond + win) * hw * c)), dim3(kBlock), 0, (hipStream_t)stream,
                       latents, cond, idx, cond_slot, win, n_cond, c, t_total, hw, cfg_copies, hi_lo, (half_t*)out);
    MV_CHECK_LAUNCH("mv_window_gather");
    return MV_OK;
}

extern "C" int mv_window_scatter_add(const void* eps_win, int32_t eps_is_f32, const int32_t* idx, int32_t win, int32_t n_cond,
                                     int32_t c, int32_t t_total, int32_t hw, int32_t halves, int32_t half_offset, float* eps_acc,
                                     float* counter, int32_t add_counter, void* stream) {
    MV_REQUIRE(eps_win && idx && eps_acc && counter && win > 0 && win <= kBlock && halves > 0, "mv_window_scatter_add: bad args");
    const dim3 g(grid_for((long)halves * c * win * hw));
    if (eps_is_f32)
        hipLaunchKernelGGL(window_scatter_add_kernel<float>, g, dim3(kBlock), 0, (hipStream_t)stream, (const float*)eps_win, idx, win,
                           n_cond, c, t_total, hw, halves, half_offset, eps_acc, counter, add_counter);
    else
        hipLaunchKernelGGL(window_scatter_add_kernel<half_t>, g, dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)eps_win, idx, win,
                           n_cond, c, t_total, hw, halves, half_offset, eps_acc, counter, add_counter);
    MV_CHECK_LAUNCH("mv_window_scatter_add");
    return MV_OK;
}

extern "C" int mv_window_units_reduce(const float* units, int64_t unit_stride, const int32_t* table, int32_t maxc, int32_t c,
                                      int32_t t_total, int32_t hw, int32_t halves, float* eps_acc, void* stream) {
    MV_REQUIRE(units && table && eps_acc && maxc > 0 && c > 0 && t_total > 0 && hw > 0 && (halves == 1 || halves == 2) && unit_stride > 0,
               "mv_window_units_reduce: bad args");
    hipLaunchKernelGGL(window_units_reduce_kernel, dim3(grid_for((long)halves * t_total * hw)), dim3(kBlock), 0, (hipStream_t)stream, units,
                       (long)unit_stride, table, maxc, c, t_total, hw, halves, eps_acc);
    MV_CHECK_LAUNCH("mv_window_units_reduce");
    return MV_OK;
}

extern "C" int mv_softmax_rows_f16(void* x, int64_t ldx, int64_t rows, int32_t cols, void* stream) {
    MV_REQUIRE(x && rows > 0 && rows < (1L << 31) && cols > 0 && cols % 8 == 0 && cols <= 16384 && ldx % 8 == 0 && ldx >= cols &&
                   (reinterpret_cast<uintptr_t>(x) & 15) == 0,
               "mv_softmax_rows_f16: need a 16-byte aligned x, cols %% 8 == 0, cols <= 16384, ldx %% 8 == 0 (rows=%ld cols=%d ldx=%ld)", (long)rows, cols, (long)ldx);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (half_t*)x, (long)ldx, cols);
    MV_CHECK_LAUNCH("mv_softmax_rows_f16");
    return MV_OK;
}

extern "C" int mv_cfg_ddim_step(float* latents, const float* eps_acc, const float* counter, int32_t c, int32_t t_total,
                                int32_t hw, int32_t halves, float guidance, float alpha_t, float alpha_prev, void* stream) {
    MV_REQUIRE(latents && eps_acc && counter && (halves == 1 || halves == 2) && alpha_t > 0.f && alpha_t <= 1.f && alpha_prev > 0.f && alpha_prev <= 1.f,
               "mv_cfg_ddim_step: bad args");
    hipLaunchKernelGGL(cfg_ddim_step_kernel, dim3(grid_for((long)c * t_total * hw)), dim3(kBlock), 0, (hipStream_t)stream, latents,
                       eps_acc, counter, c, t_total, hw, halves, guidance, sqrtf(alpha_t), sqrtf(1.f - alpha_t), sqrtf(alpha_prev),
                       sqrtf(1.f - alpha_prev));
    MV_CHECK_LAUNCH("mv_cfg_ddim_step");
    return MV_OK;
}

extern "C" int mv_cfg_affine_step(float* latents, const float* eps_acc, const float* counter, int32_t c, int32_t t_total,
                                  int32_t hw, int32_t halves, float guidance, float cx, float ce, void* stream) {
    MV_REQUIRE(latents && eps_acc && counter && (halves == 1 || halves == 2) && c > 0 && t_total > 0 && hw > 0,
               "mv_cfg_affine_step: bad args");
    hipLaunchKernelGGL(cfg_affine_step_kernel, dim3(grid_for((long)c * t_total * hw)), dim3(kBlock), 0, (hipStream_t)stream, latents,
                       eps_acc, counter, c, t_total, hw, halves, guidance, cx, ce);
    MV_CHECK_LAUNCH("mv_cfg_affine_step");
    return MV_OK;
}

extern "C" int mv_pack_conv_weight_f16(const void* w, int32_t w_is_f32, void* out, int32_t o, int32_t i, int32_t taps, void* stream) {
    MV_REQUIRE(w && out && o > 0 && i > 0 && taps > 0, "mv_pack_conv_weight_f16: bad args");
    const long n = (long)o * i * taps;
    if (w_is_f32) hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (const float*)w, (half_t*)out, o, i, taps);
    else hipLaunchKernelGGL(pack_conv_weight_kernel<half_t>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, (const half_t*)w, (half_t*)out, o, i, taps);
    MV_CHECK_LAUNCH("mv_pack_conv_weight_f16");
    return MV_OK;
}

extern "C" int mv_probe_tr16(const void* lds_image, void* out, void* stream) {
    MV_REQUIRE(lds_image && out, "mv_probe_tr16: null pointer");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const short*)lds_image, (short*)out);
    MV_CHECK_LAUNCH("mv_probe_tr16");
    return MV_OK;
}
