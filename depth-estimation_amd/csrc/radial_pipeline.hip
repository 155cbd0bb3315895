// radial_pipeline.hip -- the radial (polar) flow -> depth path as one call (BASELINE configs[2]).
//   replaces, composed: radial/test_radial_opticalflow.lua:186-225 and radial/train_radial_opticalflow.lua:161-182
//     polar_prev, polar_img = cartesian2polar(frame, getC2PMask(wImg, hImg, wInput, hInput, e2, floor((wKernel-1)/2),
//                                                              ceil((wKernel-1)/2), rmax, alpha))        data.lua:242-248
//     output = getTesterNetwork(networkp):forward{polar_prev, polar_img}                 radial_opticalflow_network.lua:56-74
//              = SpatialRadialMatching(hWin){ filter(crop(polar_prev, hWin-1 rows at the bottom)), filter(polar_img) }
//              with filter = conv(C -> n1, 1 x kW) [tanh] conv(n1 -> n2, kH x 1) -- the default separable stack
//              {{3,1,17,5},{5,17,1,10}} (train_radial:27), shared weights                radial_opticalflow_network.lua:6-30
//     idx = output:min(3) - 1                                                           test_radial:205-207
//     cartidx = cartesian2polar(idx, getP2CMaskOF(networkp, e2, alpha))                 test_radial:217-218, polar.lua:18-30
//     depth, confs = flow2depth(networkp, cartidx, e2 * getKOutput(networkp), 0.65)     test_radial:224-225, display.lua:6-58
// Kernels (every arithmetic expression is the one of the stand-alone ops in polar.hip / filters.hip /
// ssd_cost_volume.hip, so the one-call result is bit-identical to the staged host path):
//   polar_warp_pair_kernel   C2P grid evaluated in place (never stored) + bilinear gather of BOTH frames, wrap columns included
//   conv_rows_kernel         1 x kW correlation, all nOut planes per thread from an LDS row tile (accumulation order of
//                            nn.SpatialConvolution: bias, then input plane, then tap; separately rounded multiply and add)
//   conv_cols_kernel         kH x 1 correlation, RY output rows per thread: a column of RY + kH - 1 inputs is read once per plane
//   radial_match_kernel      A1r + arg-min: lanes over x (coalesced planes), RY rows per thread share their hWin + RY - 1
//                            frame-1 rows, features summed in order k = 0..K-1; the volume leaves through an LDS transpose
//                            as whole rows of W * hWin contiguous floats; first-minimum index - 1 = the radial flow
//   p2c_flow_depth_kernel    P2C grid in place + bilinear sample of the polar flow + flow2depth
#include "dfe_internal.h"
#include <memory>
#include <cmath>

#ifndef DFE_RADIAL_RY
#define DFE_RADIAL_RY 8   // output rows per thread of the radial matcher (measured 720p, K = 10, hWin = 15: 4 rows 33.7 us, 6 rows 35.5, 8 rows 27.6, 10 rows 48.6 -- 256 VGPRs, one wave per SIMD)
#endif

namespace {

int grid1d(long long n, int bs = 256) {
    long long b = (n + bs - 1) / bs;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float bilinear_at(const float *__restrict__ p, int H, int W, float fy, float fx) {
#pragma clang fp contract(off)
    fy = fy < 0 ? 0 : (fy > (float)(H - 1) ? (float)(H - 1) : fy);
    fx = fx < 0 ? 0 : (fx > (float)(W - 1) ? (float)(W - 1) : fx);
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float top = (1 - wx) * p[(long long)y0 * W + x0] + wx * p[(long long)y0 * W + x1];
    const float bot = (1 - wx) * p[(long long)y1 * W + x0] + wx * p[(long long)y1 * W + x1];
    return (1 - wy) * top + wy * bot;
}

// polar image [C][hdst][Wp], Wp = lpad + wdst + rpad; padded column jp shows grid column (jp - lpad) mod wdst
// (cartesian2polar.lua:42-47: the left pad repeats the last lpad columns, the right pad the first rpad)
// the grid's radius depends on the row only and its angle on the column only: r(i), sin / cos(theta(j)) -- the double-precision
// pow / sin / cos of the reference's inline C -- are tabulated once per call (hdst + wdst values) instead of per pixel
__global__ void polar_tables_kernel(int wdst, int hdst, float kr, float ktheta, float alpha, float *__restrict__ rt, double *__restrict__ sn,
                                    double *__restrict__ cs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < hdst) rt[t] = (float)((double)kr * pow((double)(float)t, (double)alpha));   // cartesian2polar.lua:34
    if (t < wdst) {
        const float th = ktheta * (float)t;                                             // :35
        sn[t] = sin((double)th);
        cs[t] = cos((double)th);
    }
}

// both frames as channel-interleaved float4 pixels (C <= 4): the warp's bilinear taps become one 16-B load each instead of C
// scattered 4-B loads from planes H*W apart (the polar rows are circles in the frame: every tap is its own cache line)
__global__ __launch_bounds__(256) void interleave_pair_kernel(const float *__restrict__ f0, const float *__restrict__ f1, int C, long long HW,
                                                             float4 *__restrict__ o0, float4 *__restrict__ o1) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < HW; e += (long long)gridDim.x * blockDim.x) {
        o0[e] = make_float4(f0[e], C > 1 ? f0[HW + e] : 0.f, C > 2 ? f0[2 * HW + e] : 0.f, C > 3 ? f0[3 * HW + e] : 0.f);
        o1[e] = make_float4(f1[e], C > 1 ? f1[HW + e] : 0.f, C > 2 ? f1[2 * HW + e] : 0.f, C > 3 ? f1[3 * HW + e] : 0.f);
    }
}

__device__ __forceinline__ float4 bilinear4_at(const float4 *__restrict__ p, int H, int W, float fy, float fx) {
#pragma clang fp contract(off)
    fy = fy < 0 ? 0 : (fy > (float)(H - 1) ? (float)(H - 1) : fy);
    fx = fx < 0 ? 0 : (fx > (float)(W - 1) ? (float)(W - 1) : fx);
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float4 a = p[(long long)y0 * W + x0], b = p[(long long)y0 * W + x1], c = p[(long long)y1 * W + x0], d = p[(long long)y1 * W + x1];
    float4 o;
#define DFE_BL(m) { const float top = (1 - wx) * a.m + wx * b.m; const float bot = (1 - wx) * c.m + wx * d.m; o.m = (1 - wy) * top + wy * bot; }
    DFE_BL(x) DFE_BL(y) DFE_BL(z) DFE_BL(w)
#undef DFE_BL
    return o;
}

template <bool IL>
__global__ __launch_bounds__(256) void polar_warp_pair_kernel(const float *__restrict__ f0, const float *__restrict__ f1, int C, int H, int W,
                                                             int wdst, int hdst, int Wp, int lpad, float xc, float yc, const float *__restrict__ rt,
                                                             const double *__restrict__ sn, const double *__restrict__ cs, float *__restrict__ o0,
                                                             float *__restrict__ o1) {
    // A block = a patch of 8 radii x 32 angles of the polar image: its 1024 taps fall into a compact piece of the frame (an arc of ~50
    // pixels, 8 deep: ~60 cache lines), where 256 consecutive angles of ONE radius -- the first version -- ran along 400 pixels of a
    // circle, nearly every tap its own line (176 MB gathered for 45 MB written, 76 % of the wave-cycles waiting: profiles/r04_af).  The
    // stores stay whole 128-B lines (32 angles of a row).
    const long long total = (long long)hdst * Wp, HW = (long long)H * W;
    const int npx = (Wp + 31) >> 5;
    {
        const int pb = blockIdx.x, pi = pb / npx, pj = pb - pi * npx;
        const int i = pi * 8 + (int)(threadIdx.x >> 5), jp = pj * 32 + (int)(threadIdx.x & 31);
        if (i >= hdst || jp >= Wp) return;
        const long long e = (long long)i * Wp + jp;
        int j = jp - lpad;
        j = j < 0 ? j + wdst : (j >= wdst ? j - wdst : j);
        const float r = rt[i];
        const float fy = (float)((double)r * sn[j] + (double)yc);                       // :36
        const float fx = (float)((double)r * cs[j] + (double)xc);                       // :37
        if constexpr (IL) {   // f0 / f1 are the interleaved float4 copies
            const float4 v0 = bilinear4_at(reinterpret_cast<const float4 *>(f0), H, W, fy, fx);
            const float4 v1 = bilinear4_at(reinterpret_cast<const float4 *>(f1), H, W, fy, fx);
            const float a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w};
            for (int c = 0; c < C; ++c) {
                o0[c * total + e] = a0[c];
                o1[c * total + e] = a1[c];
            }
        } else {
            for (int c = 0; c < C; ++c) {
                o0[c * total + e] = bilinear_at(f0 + c * HW, H, W, fy, fx);
                o1[c * total + e] = bilinear_at(f1 + c * HW, H, W, fy, fx);
            }
        }
    }
}

// N consecutive wave-uniform floats through the scalar cache in as few s_load_dwordxN as possible (the compiler picks x8 / x4 /
// x2 / x1 from the constant address space): a weight row costs 3 scalar instructions instead of 17.  One scalar load per
// multiply-add -- the first version of these kernels -- left them bound by scalar-memory latency at a third of the VALU rate.
typedef const float __attribute__((address_space(4))) *rp_cfptr;
template <int N> __device__ __forceinline__ void load_uniform(const float *p, float (&v)[N]) {
    rp_cfptr q = (rp_cfptr)p;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = q[i];
}

// out[o][y][x] = bias[o] + sum_i sum_v w[o][i][v] * in[i][y][x+v]   (kH = 1)
template <int NOUT, bool TANH, int KW = 0>
__global__ __launch_bounds__(256) void conv_rows_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                       int nIn, int H, int W, int kW, float *__restrict__ out) {
#pragma clang fp contract(off)
    extern __shared__ float tile[];                      // [nIn][256 + kW - 1]
    const int Wo = W - kW + 1, TW = 256 + kW - 1;
    const int y = blockIdx.y, x0 = blockIdx.x * 256, tx = threadIdx.x;
    for (int i = 0; i < nIn; ++i)
        for (int s = tx; s < TW; s += 256) {
            const int x = x0 + s;
            tile[i * TW + s] = x < W ? in[((long long)i * H + y) * W + x] : 0.f;
        }
    __syncthreads();
    const int x = x0 + tx;
    if (x >= Wo) return;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = bias ? bias[o] : 0.f;
    if constexpr (KW > 0) {
        for (int i = 0; i < nIn; ++i) {
            float a[KW];
#pragma unroll
            for (int v = 0; v < KW; ++v) a[v] = tile[i * TW + tx + v];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float wv[KW];
                load_uniform<KW>(w + ((long long)o * nIn + i) * KW, wv);
#pragma unroll
                for (int v = 0; v < KW; ++v) acc[o] = acc[o] + wv[v] * a[v];
            }
        }
    } else {
    for (int i = 0; i < nIn; ++i)
        for (int v = 0; v < kW; ++v) {
            const float a = tile[i * TW + tx + v];
#pragma unroll
            for (int o = 0; o < NOUT; ++o) acc[o] = acc[o] + w[((long long)o * nIn + i) * kW + v] * a;   // (wave-uniform weights: scalar loads)
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) out[((long long)o * H + y) * Wo + x] = TANH ? tanhf(acc[o]) : acc[o];
}

// The same sums for TWO image rows per thread (KW taps unrolled): the multiply and the add of a tap are v_pk_mul_f32 / v_pk_add_f32 on
// the row pair (the weight an SGPR pair with its low half broadcast), each element rounded as the scalar operations are, in the same
// order -- same results.  With one row per thread every operation had a scalar operand (the VALU's slow path, 0.9 per cycle and CU:
// DESIGN 4.6) and conv_rows_kernel<5, false, 17> ran at 46 % of the rate its 255 multiply-adds per pixel allow; the tile holds the two
// rows interleaved ([plane][column][2]), so a tap's operand pair is one aligned 8-B piece of an LDS read.
template <int NOUT, bool TANH, int KW>
__global__ __launch_bounds__(256) void conv_rows_pk_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                          int nIn, int H, int W, float *__restrict__ out) {
#pragma clang fp contract(off)
    typedef float f2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [nIn][TW][2]
    constexpr int TW = 256 + KW - 1 + ((256 + KW - 1) & 1);        // (even: 16-B alignment of the planes)
    const int Wo = W - KW + 1;
    const int y = 2 * blockIdx.y, y1 = min(y + 1, H - 1), x0 = blockIdx.x * 256, tx = threadIdx.x;
    for (int i = 0; i < nIn; ++i)
        for (int s = tx; s < TW; s += 256) {
            const int x = x0 + s;
            const float a = x < W ? in[((long long)i * H + y) * W + x] : 0.f, b = x < W ? in[((long long)i * H + y1) * W + x] : 0.f;
            *reinterpret_cast<f2 *>(tile + ((long long)i * TW + s) * 2) = f2{a, b};
        }
    __syncthreads();
    const int x = x0 + tx;
    if (x >= Wo) return;
    f2 acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) { const float b = bias ? bias[o] : 0.f; acc[o] = f2{b, b}; }
    for (int i = 0; i < nIn; ++i) {
        f2 a[KW];
#pragma unroll
        for (int v = 0; v < KW; ++v) a[v] = *reinterpret_cast<const f2 *>(tile + ((long long)i * TW + tx + v) * 2);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float wv[KW];
            load_uniform<KW>(w + ((long long)o * nIn + i) * KW, wv);
#pragma unroll
            for (int v = 0; v < KW; ++v) {
                const f2 w2 = f2{wv[v], wv[v]};
                acc[o] = acc[o] + w2 * a[v];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        out[((long long)o * H + y) * Wo + x] = TANH ? tanhf(acc[o][0]) : acc[o][0];
        if (y + 1 < H) out[((long long)o * H + y + 1) * Wo + x] = TANH ? tanhf(acc[o][1]) : acc[o][1];
    }
}

// out[o][y][x] = bias[o] + sum_i sum_u w[o][i][u] * in[i][y+u][x]   (kW = 1); RY output rows per thread
template <int NOUT, int RY, int KH>
__global__ __launch_bounds__(256) void conv_cols_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                       int nIn, int H, int W, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H - KH + 1;
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * RY;
    if (x >= W) return;
    // two output rows per instruction (v_pk_mul_f32 / v_pk_add_f32 on row pairs, the weight an SGPR pair with its low half broadcast):
    // every element is rounded as the scalar multiply and add are, in the same order -- same results, half the VALU issues
    static_assert(RY % 2 == 0, "row pairs");
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[RY / 2][NOUT];
#pragma unroll
    for (int r = 0; r < RY / 2; ++r)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) { const float b = bias ? bias[o] : 0.f; acc[r][o] = f2{b, b}; }
    for (int i = 0; i < nIn; ++i) {
        float col[RY + KH - 1];
#pragma unroll
        for (int s = 0; s < RY + KH - 1; ++s) col[s] = in[((long long)i * H + min(y0 + s, H - 1)) * W + x];
        f2 pr[RY + KH - 2];                       // (col[s], col[s + 1]): the odd ones are register copies, shared by every output plane
#pragma unroll
        for (int s = 0; s < RY + KH - 2; ++s) pr[s] = f2{col[s], col[s + 1]};
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            float wv[KH];
            load_uniform<KH>(w + ((long long)o * nIn + i) * KH, wv);
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                const f2 w2 = f2{wv[u], wv[u]};
#pragma unroll
                for (int r = 0; r < RY / 2; ++r) acc[r][o] = acc[r][o] + w2 * pr[2 * r + u];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r)
        if (y0 + r < Ho)
#pragma unroll
            for (int o = 0; o < NOUT; ++o) out[((long long)o * Ho + y0 + r) * W + x] = acc[r >> 1][o][r & 1];
}

// A1r + arg-min.  in1 [K][H1p >= H1][W] (only the first H1 rows of a plane are used: the cropped previous frame), in2 [K][H1+HW-1][W]; vol [H1][W][HW] (may be NULL), flow [H1][W] = first-min index (0-based)
// as float, the last row zeroed when zero_last (train_radial:180).  Block = 64 columns x RY rows per thread x 4 row groups.
template <int HWIN, int RY>
__global__ __launch_bounds__(256) void radial_match_kernel(const float *__restrict__ in1, int H1p, const float *__restrict__ in2, int K, int H1, int W,
                                                          float *__restrict__ vol, float *__restrict__ flow, int zero_last) {
#pragma clang fp contract(off)
    __shared__ float stage[4][64 * HWIN + 1];            // one output row of the block's 64 columns per row group
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y0 = (blockIdx.y * 4 + grp) * RY;
    const int H2 = H1 + HWIN - 1;
    const bool inx = x < W;
    const int xs = inx ? x : W - 1;
    float acc[RY][HWIN];
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int d = 0; d < HWIN; ++d) acc[r][d] = 0.f;
    if (y0 < H1) {
        for (int k = 0; k < K; ++k) {
            float b[RY + HWIN - 1], a[RY];
#pragma unroll
            for (int s = 0; s < RY + HWIN - 1; ++s) b[s] = in2[((long long)k * H2 + min(y0 + s, H2 - 1)) * W + xs];
#pragma unroll
            for (int r = 0; r < RY; ++r) a[r] = in1[((long long)k * H1p + min(y0 + r, H1 - 1)) * W + xs];
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int d = 0; d < HWIN; ++d) {
                    const float t = a[r] - b[r + d];
                    acc[r][d] = acc[r][d] + t * t;
                }
        }
    }
    const int wcols = min(64, W - (int)blockIdx.x * 64);   // columns of this block inside the frame
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const int y = y0 + r;                              // (wave-uniform)
        if (y < H1) {
            if (inx) {
                float best = acc[r][0];
                int bi = 0;
#pragma unroll
                for (int d = 1; d < HWIN; ++d)
                    if (acc[r][d] < best) { best = acc[r][d]; bi = d; }   // strict: TH min keeps the first minimum
                flow[(long long)y * W + x] = (zero_last && y == H1 - 1) ? 0.f : (float)bi;
            }
            if (vol) {
#pragma unroll
                for (int d = 0; d < HWIN; ++d) stage[grp][lane * HWIN + d] = acc[r][d];
                __builtin_amdgcn_wave_barrier();
                float *dst = vol + ((long long)y * W + (long long)blockIdx.x * 64) * HWIN;
                for (int s = lane; s < wcols * HWIN; s += 64) dst[s] = stage[grp][s];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// cart[i][j] = bilinear(polar flow, P2C grid(i, j)); depth / conf = flow2depth(cart)
__global__ __launch_bounds__(256) void p2c_flow_depth_kernel(const float *__restrict__ pflow, int hPolar, int wPolar, int wdst, int hdst, float xc,
                                                            float yc, float kx, float ky, float pi2, float invalpha, float cx2, float cy2,
                                                            float infty, float *__restrict__ cart, float *__restrict__ depth,
                                                            float *__restrict__ conf) {
    const long long total = (long long)hdst * wdst;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / wdst), j = (int)(e - (long long)i * wdst);
        const float x = (float)j - xc, y = (float)i - yc;                                                       // cartesian2polar.lua:80-81
        const float my = (float)(pow((double)(x * x + y * y), (double)invalpha) * (double)ky);                  // :82
        const float mx = (float)(fmod(atan2((double)y, (double)x) + (double)pi2, (double)pi2) * (double)kx);    // :83
        const float f = bilinear_at(pflow, hPolar, wPolar, my, mx);
        if (cart) cart[e] = f;
        const float a = (float)j - cx2, b = (float)i - cy2;
        const float d = (float)sqrt((double)(a * a + b * b));     // radial_opticalflow_display.lua:39
        float o = 0.f, c = 1.f;
        if (d > 10.0f) o = (f < 0.1f) ? infty : d / f;            // :42-46
        else c = 0.f;                                             // :48
        if (depth) depth[e] = o / infty;                          // :57
        if (conf) conf[e] = c;
    }
}

double lua_rmax(double h, double w, double ex, double ey) {   // getRMax radial/radial_opticalflow_polar.lua:4-10
    const double a = ex * ex + ey * ey, b = (w - ex) * (w - ex) + ey * ey, c = ex * ex + (h - ey) * (h - ey), d = (w - ex) * (w - ex) + (h - ey) * (h - ey);
    return floor(sqrt(fmax(fmax(a, b), fmax(c, d))));
}

template <int NOUT>
int launch_conv_rows(dfe_ctx *ctx, const float *in, const float *w, const float *b, int nIn, int H, int W, int kW, bool tanh_after, float *out) {
    const int Wo = W - kW + 1;
    const size_t lds = (size_t)nIn * (256 + kW - 1) * sizeof(float);
    dim3 grid(dfe_cdiv(Wo, 256), H);
    if (kW == 17) {
        const dim3 grid2(dfe_cdiv(Wo, 256), dfe_cdiv(H, 2));
        const size_t lds2 = (size_t)nIn * (256 + 17 - 1) * 2 * sizeof(float);
        if (tanh_after) hipLaunchKernelGGL((conv_rows_pk_kernel<NOUT, true, 17>), grid2, dim3(256), lds2, ctx->stream, in, w, b, nIn, H, W, out);
        else hipLaunchKernelGGL((conv_rows_pk_kernel<NOUT, false, 17>), grid2, dim3(256), lds2, ctx->stream, in, w, b, nIn, H, W, out);
    } else if (tanh_after) hipLaunchKernelGGL((conv_rows_kernel<NOUT, true>), grid, dim3(256), lds, ctx->stream, in, w, b, nIn, H, W, kW, out);
    else hipLaunchKernelGGL((conv_rows_kernel<NOUT, false>), grid, dim3(256), lds, ctx->stream, in, w, b, nIn, H, W, kW, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // namespace

// separable stack: fast kernels for the shapes with an instantiation, the generic direct convolution otherwise (same sums)
static int radial_filter(dfe_ctx *ctx, const dfe_radial_params *p, const float *polar, int H, int Wp, const float *w1, const float *b1,
                         const float *w2, const float *b2, float *tmp, float *feat) {
    const int W = Wp - p->kW1 + 1, Ho = H - p->kH2 + 1;
    int rc = DFE_OK;
    bool done = false;
    // (the 17-tap kernel stages two interleaved rows: twice the LDS of the one-row kernel; both stay within the 64 KB a launch gets
    //  without hipFuncAttributeMaxDynamicSharedMemorySize)
    if ((size_t)p->C * (256 + p->kW1 - 1) * sizeof(float) * (p->kW1 == 17 ? 2 : 1) <= 48 * 1024) {
        switch (p->n1) {
            case 4: rc = launch_conv_rows<4>(ctx, polar, w1, b1, p->C, H, Wp, p->kW1, p->tanh_between != 0, tmp); done = true; break;
            case 5: rc = launch_conv_rows<5>(ctx, polar, w1, b1, p->C, H, Wp, p->kW1, p->tanh_between != 0, tmp); done = true; break;
            case 8: rc = launch_conv_rows<8>(ctx, polar, w1, b1, p->C, H, Wp, p->kW1, p->tanh_between != 0, tmp); done = true; break;
            default: break;
        }
    }
    if (!done) {
        rc = dfe_spatial_convolution_f32(ctx, polar, w1, b1, p->C, p->n1, H, Wp, 1, p->kW1, tmp);
        if (!rc && p->tanh_between) rc = dfe_tanh_f32(ctx, tmp, (int64_t)p->n1 * H * W, tmp);
    }
    if (rc) return rc;
    if (p->kH2 == 17 && (p->n2 == 10 || p->n2 == 8)) {
        dim3 grid(dfe_cdiv(W, 256), dfe_cdiv(Ho, 4));
        if (p->n2 == 10) hipLaunchKernelGGL((conv_cols_kernel<10, 4, 17>), grid, dim3(256), 0, ctx->stream, tmp, w2, b2, p->n1, H, W, feat);
        else hipLaunchKernelGGL((conv_cols_kernel<8, 4, 17>), grid, dim3(256), 0, ctx->stream, tmp, w2, b2, p->n1, H, W, feat);
        DFE_LAUNCH_CHECK(ctx);
        return DFE_OK;
    }
    return dfe_spatial_convolution_f32(ctx, tmp, w2, b2, p->n1, p->n2, H, W, p->kH2, 1, feat);
}

extern "C" {

int dfe_radial_out_shape(const dfe_radial_params *p, int *hMatch, int *hOut, int *wOut) {
    if (!p || p->hInput <= 0 || p->wInput <= 0 || p->kH2 <= 0 || p->hWin <= 0) return DFE_E_ARG;
    const int hm = p->hInput - (p->kH2 - 1) - (p->hWin - 1);
    if (hm < 1) return DFE_E_SHAPE;
    const double kOut = (double)hm / (double)p->hInput;   // getP2CMaskOF: hPolar = hInput - hKernel - hWin + 2
    if (hMatch) *hMatch = hm;
    if (hOut) *hOut = (int)((double)p->hImg * kOut);       // (Torch truncates the fractional tensor sizes)
    if (wOut) *wOut = (int)((double)p->wImg * kOut);
    return DFE_OK;
}

int dfe_radial_match_argmin_f32(dfe_ctx *ctx, const float *in1, int in1_plane_rows, const float *in2, int K, int H1, int W, int hWin,
                                float *volume, float *flow, int zero_last_row) {
    DFE_ENTER(ctx);
    if (in1_plane_rows <= 0) in1_plane_rows = H1;
    DFE_REQUIRE(ctx, in1_plane_rows >= H1, DFE_E_SHAPE, "dfe_radial_match_argmin_f32: in1 planes of %d rows for %d output rows", in1_plane_rows, H1);
    DFE_REQUIRE(ctx, in1 && in2 && flow, DFE_E_ARG, "dfe_radial_match_argmin_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W > 0 && hWin > 0, DFE_E_SHAPE, "dfe_radial_match_argmin_f32: K=%d H1=%d W=%d hWin=%d", K, H1, W, hWin);
    DFE_REQUIRE(ctx, hWin == 15 || hWin == 12 || hWin == 8 || hWin == 16, DFE_E_UNSUPPORTED,
                "dfe_radial_match_argmin_f32: hWin %d has no instantiation (8, 12, 15, 16); use dfe_radial_matching_f32 + dfe_argbest_center", hWin);
    constexpr int RYM = DFE_RADIAL_RY;
    dim3 grid(dfe_cdiv(W, 64), dfe_cdiv(H1, 4 * RYM));
    {
        DfeProfScope prof(ctx);
        switch (hWin) {
            case 15: hipLaunchKernelGGL((radial_match_kernel<15, RYM>), grid, dim3(256), 0, ctx->stream, in1, in1_plane_rows, in2, K, H1, W, volume, flow, zero_last_row); break;
            case 12: hipLaunchKernelGGL((radial_match_kernel<12, RYM>), grid, dim3(256), 0, ctx->stream, in1, in1_plane_rows, in2, K, H1, W, volume, flow, zero_last_row); break;
            case 16: hipLaunchKernelGGL((radial_match_kernel<16, RYM>), grid, dim3(256), 0, ctx->stream, in1, in1_plane_rows, in2, K, H1, W, volume, flow, zero_last_row); break;
            default: hipLaunchKernelGGL((radial_match_kernel<8, RYM>), grid, dim3(256), 0, ctx->stream, in1, in1_plane_rows, in2, K, H1, W, volume, flow, zero_last_row); break;
        }
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "radial_match_kernel";
    return DFE_OK;
}

int dfe_radial_flow_depth_pair_f32(dfe_ctx *ctx, const dfe_radial_params *p, const float *prev, const float *cur, double e2x, double e2y,
                                   const float *w1, const float *b1, const float *w2, const float *b2, float *volume, float *polar_flow,
                                   float *cart_flow, float *depth, float *conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, p && prev && cur && w1 && w2, DFE_E_ARG, "dfe_radial_flow_depth_pair_f32: NULL argument");
    DFE_REQUIRE(ctx, p->C > 0 && p->hImg > 0 && p->wImg > 0 && p->hInput > 0 && p->wInput > 0 && p->n1 > 0 && p->n2 > 0 && p->kW1 > 0 && p->kH2 > 0 &&
                         p->hWin > 0 && p->alpha_polar > 0 && p->kinfty > 0,
                DFE_E_ARG, "dfe_radial_flow_depth_pair_f32: bad parameter block");
    int hm, hOut, wOut;
    DFE_REQUIRE(ctx, dfe_radial_out_shape(p, &hm, &hOut, &wOut) == DFE_OK && hOut > 0 && wOut > 0, DFE_E_SHAPE,
                "dfe_radial_flow_depth_pair_f32: polar height %d too small for kernel %d + window %d", p->hInput, p->kH2, p->hWin);
    const int lpad = (p->kW1 - 1) / 2, rpad = (p->kW1 - 1) - lpad;        // floor / ceil((wKernel-1)/2): test_radial:190-191
    DFE_REQUIRE(ctx, lpad <= p->wInput && rpad <= p->wInput, DFE_E_SHAPE, "dfe_radial_flow_depth_pair_f32: wInput %d below the kernel width", p->wInput);
    const int Wp = p->wInput + lpad + rpad, H = p->hInput, W = p->wInput;
    const int Hf2 = H - (p->kH2 - 1);                                        // feature rows of a full polar frame
    // scratch: two polar frames, one row-filter buffer, two feature maps, the polar flow
    const size_t polar_b = ((size_t)p->C * H * Wp * 4 + 255) / 256 * 256, tmp_b = ((size_t)p->n1 * H * W * 4 + 255) / 256 * 256;
    const size_t f2_b = ((size_t)p->n2 * Hf2 * W * 4 + 255) / 256 * 256, f1_b = f2_b;
    const size_t pf_b = ((size_t)hm * W * 4 + 255) / 256 * 256;
    const size_t tab_b = ((size_t)H * 4 + 255) / 256 * 256 + 2 * (((size_t)W * 8 + 255) / 256 * 256);
    const bool il = p->C <= 4;   // (planar taps measured with the patch mapping too: 0.194 against 0.185 ms at 720p)
    const size_t il_b = il ? ((size_t)p->hImg * p->wImg * 16 + 255) / 256 * 256 : 0;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, 2 * polar_b + tmp_b + f1_b + f2_b + pf_b + tab_b + 2 * il_b, &scr);
    if (rc) return rc;
    float *pol0 = (float *)scr, *pol1 = (float *)((char *)scr + polar_b), *tmp = (float *)((char *)scr + 2 * polar_b);
    float *feat1 = (float *)((char *)tmp + tmp_b), *feat2 = (float *)((char *)feat1 + f1_b), *pflow = (float *)((char *)feat2 + f2_b);
    if (polar_flow) pflow = polar_flow;

    const double rmax = lua_rmax(p->hImg, p->wImg, e2x, e2y);
    std::unique_ptr<DfeStageScope> stage(new DfeStageScope(ctx, DFE_STAGE_FILTER));   // polar warps + filters
    {   // 1. both frames to polar (getC2PMask's constants: cartesian2polar.lua:13-14)
        const float kr = (float)(rmax / pow((double)H, (double)p->alpha_polar));
        const float ktheta = (float)(2 * M_PI / W);
        char *tb = (char *)scr + 2 * polar_b + tmp_b + f1_b + f2_b + pf_b;
        float *rt = (float *)tb;
        double *sn = (double *)(tb + ((size_t)H * 4 + 255) / 256 * 256), *cs = (double *)((char *)sn + ((size_t)W * 8 + 255) / 256 * 256);
        hipLaunchKernelGGL(polar_tables_kernel, dim3(dfe_cdiv(H > W ? H : W, 256)), dim3(256), 0, ctx->stream, W, H, kr, ktheta, p->alpha_polar, rt, sn, cs);
        if (il) {
            float4 *i0 = (float4 *)(tb + tab_b), *i1 = (float4 *)(tb + tab_b + il_b);
            hipLaunchKernelGGL(interleave_pair_kernel, dim3(grid1d((long long)p->hImg * p->wImg)), dim3(256), 0, ctx->stream, prev, cur, p->C,
                               (long long)p->hImg * p->wImg, i0, i1);
            hipLaunchKernelGGL(polar_warp_pair_kernel<true>, dim3(dfe_cdiv(H, 8) * dfe_cdiv(Wp, 32)), dim3(256), 0, ctx->stream, (const float *)i0,
                               (const float *)i1, p->C, p->hImg, p->wImg, W, H, Wp, lpad, (float)e2x, (float)e2y, rt, sn, cs, pol0, pol1);
        } else {
            hipLaunchKernelGGL(polar_warp_pair_kernel<false>, dim3(dfe_cdiv(H, 8) * dfe_cdiv(Wp, 32)), dim3(256), 0, ctx->stream, prev, cur, p->C, p->hImg,
                               p->wImg, W, H, Wp, lpad, (float)e2x, (float)e2y, rt, sn, cs, pol0, pol1);
        }
        DFE_LAUNCH_CHECK(ctx);
    }
    // 2. shared filter on both polar frames.  The reference crops the previous frame's last hWin-1 rows BEFORE its filter
    //    (SpatialPadding(0,0,0,-hWin+1), network.lua:59); a valid correlation's surviving rows do not see the cropped ones, so
    //    the full frame is filtered and the matcher reads the first hm rows of each feature plane (plane pitch Hf2 rows).
    rc = radial_filter(ctx, p, pol0, H, Wp, w1, b1, w2, b2, tmp, feat1);
    if (rc) return rc;
    rc = radial_filter(ctx, p, pol1, H, Wp, w1, b1, w2, b2, tmp, feat2);
    if (rc) return rc;
    stage.reset();
    stage.reset(new DfeStageScope(ctx, DFE_STAGE_MATCH));
    // 3. matcher + arg-min (+ the volume when asked for); the last flow row zeroed only on request (train_radial:178-180 does
    //    it for its display; test_radial:204-207, the path this call replaces, does not)
    rc = dfe_radial_match_argmin_f32(ctx, feat1, Hf2, feat2, p->n2, hm, W, p->hWin, volume, pflow, p->zero_last_row != 0);
    if (rc) return rc;
    stage.reset();
    stage.reset(new DfeStageScope(ctx, DFE_STAGE_EXTRACT));
    {   // 4. polar flow -> cartesian -> depth (getP2CMaskOF + flow2depth with center2 = e2 * getKOutput)
        const double kOut = (double)hm / (double)H;
        const double nrmax = rmax * kOut;
        const float pi2 = (float)(2 * M_PI);
        const float kx = (float)((double)W / (2 * M_PI));
        const float ky = (float)((double)hm / pow(nrmax, 1.0 / (double)p->alpha_polar));
        const float invalpha = (float)(1.0 / (double)p->alpha_polar) * 0.5f;
        const double kOut2 = (double)(H - (p->kH2 - 1) / 2 - p->hWin + 1) / (double)H;           // getKOutput polar.lua:12-16 (sic: not kOut)
        const double c2x = e2x * kOut2, c2y = e2y * kOut2;
        const float infty = (float)(lua_rmax(p->hImg, p->wImg, c2x, c2y) * p->kinfty);
        hipLaunchKernelGGL(p2c_flow_depth_kernel, dim3(grid1d((long long)hOut * wOut)), dim3(256), 0, ctx->stream, pflow, hm, W, wOut, hOut,
                           (float)(e2x * kOut), (float)(e2y * kOut), kx, ky, pi2, invalpha, (float)c2x, (float)c2y, infty,
                           cart_flow, depth, conf);
        DFE_LAUNCH_CHECK(ctx);
    }
    ctx->last_kernel = "radial_match_kernel";
    return DFE_OK;
}

}  // extern "C"
