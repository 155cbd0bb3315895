// polar.hip -- A13 polar sampling grids, A14 bilinear warp, A12(ii) radial flow -> depth.
//   replaces: getC2PMask / getP2CMask (inline C)      radial/cartesian2polar.lua:4-49, 51-89
//             cartesian2polar = image.warp(.., 'bilinear', false)   radial/cartesian2polar.lua:91-93
//             flow2depth (inline C do_depths)          radial/radial_opticalflow_display.lua:6-58
// The reference's inline C keeps `float` variables but calls the double libm functions (pow, sin, cos,
// atan2, fmod, sqrt) on them; the kernels do the same promotions so results agree to the last float bit
// wherever the device libm is correctly rounded.
#include "dfe_internal.h"
#include <cmath>

namespace {

__global__ void c2p_kernel(int wdst, int hdst, int Wp, int lpad, float xc, float yc, float kr, float ktheta, float alpha,
                           float *__restrict__ mask) {
    const long long total = (long long)hdst * wdst;
    float *m0 = mask, *m1 = mask + (long long)hdst * Wp;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int i = (int)(e / wdst), j = (int)(e - (long long)i * wdst);
        float r = (float)((double)kr * pow((double)(float)i, (double)alpha));   // cartesian2polar.lua:34
        float th = ktheta * (float)j;                                          // :35
        m0[(long long)i * Wp + lpad + j] = (float)((double)r * sin((double)th) + (double)yc);   // :36
        m1[(long long)i * Wp + lpad + j] = (float)((double)r * cos((double)th) + (double)xc);   // :37
    }
}

// circular column padding: left pad <- last lpad columns, right pad <- first rpad columns (:42-47)
__global__ void c2p_pad_kernel(int wdst, int hdst, int Wp, int lpad, int rpad, float *__restrict__ mask) {
    const int np = lpad + rpad;
    const long long total = 2ll * hdst * np;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int j = (int)(e % np);
        long long t = e / np;
        int i = (int)(t % hdst), pl = (int)(t / hdst);
        float *m = mask + ((long long)pl * hdst + i) * Wp;
        if (j < lpad) m[j] = m[lpad + wdst - lpad + j];
        else m[lpad + wdst + (j - lpad)] = m[lpad + (j - lpad)];
    }
}

__global__ void p2c_kernel(int wdst, int hdst, float xc, float yc, float kx, float ky, float pi2, float invalpha,
                           float *__restrict__ mask) {
    const long long total = (long long)hdst * wdst;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int i = (int)(e / wdst), j = (int)(e - (long long)i * wdst);
        float x = (float)j - xc, y = (float)i - yc;                                              // :80-81
        mask[e] = (float)(pow((double)(x * x + y * y), (double)invalpha) * (double)ky);          // :82
        mask[total + e] = (float)(fmod(atan2((double)y, (double)x) + (double)pi2, (double)pi2) * (double)kx);   // :83
    }
}

// image.warp(img, mask, 'bilinear', false): absolute coordinates, plane 0 = y, plane 1 = x, 0-based;
// coordinates outside the image are clamped (border policy of the un-vendored `image` package: parity unpinned)
__global__ void warp_bilinear_kernel(const float *__restrict__ img, int C, int H, int W, const float *__restrict__ mask, int Hd,
                                     int Wd, float *__restrict__ out) {
#pragma clang fp contract(off)
    const long long P = (long long)Hd * Wd;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        float fy = mask[e], fx = mask[P + e];
        fy = fy < 0 ? 0 : (fy > (float)(H - 1) ? (float)(H - 1) : fy);
        fx = fx < 0 ? 0 : (fx > (float)(W - 1) ? (float)(W - 1) : fx);
        int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        float wy = fy - (float)y0, wx = fx - (float)x0;
        for (int c = 0; c < C; ++c) {
            const float *p = img + (long long)c * H * W;
            float top = (1 - wx) * p[(long long)y0 * W + x0] + wx * p[(long long)y0 * W + x1];
            float bot = (1 - wx) * p[(long long)y1 * W + x0] + wx * p[(long long)y1 * W + x1];
            out[(long long)c * P + e] = (1 - wy) * top + wy * bot;
        }
    }
}

__global__ void flow_to_depth_radial_kernel(const float *__restrict__ rflow, int H, int W, float xc, float yc, float infty,
                                            float *__restrict__ depth, float *__restrict__ conf) {
    const long long P = (long long)H * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        int i = (int)(e / W), j = (int)(e - (long long)i * W);
        float a = (float)j - xc, b = (float)i - yc;
        float d = (float)sqrt((double)(a * a + b * b));     // radial_opticalflow_display.lua:39
        float o = 0.f, c = 1.f;                             // ret zero-filled, confs filled with 1 (:13-14)
        if (d > 10.0f) {
            float f = rflow[e];
            o = (f < 0.1f) ? infty : d / f;                 // :42-46
        } else {
            c = 0.f;                                        // :48
        }
        depth[e] = o / infty;                               // :57 ret/infty
        conf[e] = c;
    }
}

// A12(iii) ardrone/ardrone_api.cpp:99-140.  One thread per pixel; the 20-bin histogram of the 6x6 window lives in two
// 64-bit registers (ten 6-bit counters each: a bin counts at most 36).
__global__ void flow_to_depth_ardrone_kernel(const float *__restrict__ xflow, const float *__restrict__ mask, int H, int W, float m,
                                             float *__restrict__ depth, float *__restrict__ conf) {
    const long long P = (long long)H * W;
    const int k = 3, middlex = W / 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e / W), i = (int)(e - (long long)j * W);
        const float mk = mask[e];
        float mode = 0.f;
        if (mk != 0.f) {
            unsigned long long lo = 0, hi = 0;              // bins 0..9 / 10..19
            for (int j2 = max(0, j - k); j2 < min(H, j + k); ++j2)
                for (int i2 = max(0, i - k); i2 < min(W, i + k); ++i2)
                    if (mask[(long long)j2 * W + i2] != 0.f) {
                        const int f = (int)roundf(xflow[(long long)j2 * W + i2]) + 8;
                        if (f >= 0 && f < 10) lo += 1ull << (6 * f);
                        else if (f >= 10 && f < 20) hi += 1ull << (6 * (f - 10));
                    }
            int best = 0, im = 0;
            for (int iv = 0; iv < 20; ++iv) {               // first maximum wins (:117-121)
                const int c = (int)(((iv < 10 ? lo >> (6 * iv) : hi >> (6 * (iv - 10)))) & 63);
                if (c > best) { best = c; im = iv - 8; }
            }
            mode = (float)im;
        }
        float d = 0.f, c = 0.f;
        if (mk > 0.5f && i - middlex != 0) {
            d = fabsf(mode) < 1.1f ? 100.0f : m * (float)abs(i - middlex) / fabsf(mode);
            c = 1.0f;
        }
        depth[e] = d;
        conf[e] = c;
    }
}

int grid1d(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int dfe_polar_grid_c2p_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter, float ycenter, int lpadding,
                           int rpadding, float rmax, float alpha, float *mask) {
    (void)wsrc; (void)hsrc;
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, mask && wdst > 0 && hdst > 0 && lpadding >= 0 && rpadding >= 0 && lpadding <= wdst && rpadding <= wdst,
                DFE_E_ARG, "dfe_polar_grid_c2p_f32: bad argument");
    const int Wp = wdst + lpadding + rpadding;
    const float kr = (float)((double)rmax / pow((double)hdst, (double)alpha));   // cartesian2polar.lua:13
    const float ktheta = (float)(2 * M_PI / wdst);                                 // :14
    hipLaunchKernelGGL(c2p_kernel, dim3(grid1d((long long)hdst * wdst)), dim3(256), 0, ctx->stream, wdst, hdst, Wp, lpadding, xcenter,
                       ycenter, kr, ktheta, alpha, mask);
    if (lpadding + rpadding > 0)
        hipLaunchKernelGGL(c2p_pad_kernel, dim3(grid1d(2ll * hdst * (lpadding + rpadding))), dim3(256), 0, ctx->stream, wdst, hdst, Wp,
                           lpadding, rpadding, mask);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_polar_grid_p2c_f32(dfe_ctx *ctx, int wsrc, int hsrc, int wdst, int hdst, float xcenter, float ycenter, float rmax,
                           float alpha, float *mask) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, mask && wdst > 0 && hdst > 0 && rmax > 0 && alpha > 0, DFE_E_ARG, "dfe_polar_grid_p2c_f32: bad argument");
    const float pi2 = (float)(2 * M_PI);                                                   // :58
    const float kx = (float)((double)wsrc / (2 * M_PI));                                   // :59
    const float ky = (float)((double)hsrc / pow((double)rmax, 1.0 / (double)alpha));       // :60
    const float invalpha = (float)(1.0 / (double)alpha) * 0.5f;                            // :70
    hipLaunchKernelGGL(p2c_kernel, dim3(grid1d((long long)hdst * wdst)), dim3(256), 0, ctx->stream, wdst, hdst, xcenter, ycenter, kx, ky,
                       pi2, invalpha, mask);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_warp_bilinear_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const float *mask, int Hd, int Wd, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && Hd >= 0 && Wd >= 0, DFE_E_SHAPE, "dfe_warp_bilinear_f32: bad shape");
    if ((long long)Hd * Wd == 0) return DFE_OK;
    DFE_REQUIRE(ctx, img && mask && out, DFE_E_ARG, "dfe_warp_bilinear_f32: NULL tensor");
    hipLaunchKernelGGL(warp_bilinear_kernel, dim3(grid1d((long long)Hd * Wd)), dim3(256), 0, ctx->stream, img, C, H, W, mask, Hd, Wd, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_flow_to_depth_radial(dfe_ctx *ctx, const float *rflow, int H, int W, float xcenter, float ycenter, float infty, float *depth,
                             float *conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0, DFE_E_SHAPE, "dfe_flow_to_depth_radial: H=%d W=%d", H, W);
    if ((long long)H * W == 0) return DFE_OK;
    DFE_REQUIRE(ctx, rflow && depth && conf, DFE_E_ARG, "dfe_flow_to_depth_radial: NULL tensor");
    hipLaunchKernelGGL(flow_to_depth_radial_kernel, dim3(grid1d((long long)H * W)), dim3(256), 0, ctx->stream, rflow, H, W, xcenter,
                       ycenter, infty, depth, conf);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_flow_to_depth_ardrone(dfe_ctx *ctx, const float *xflow, const float *mask, int H, int W, float imu_tx, float *depth,
                              float *conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0, DFE_E_SHAPE, "dfe_flow_to_depth_ardrone: H=%d W=%d", H, W);
    if ((long long)H * W == 0) return DFE_OK;
    DFE_REQUIRE(ctx, xflow && mask && depth && conf, DFE_E_ARG, "dfe_flow_to_depth_ardrone: NULL tensor");
    hipLaunchKernelGGL(flow_to_depth_ardrone_kernel, dim3(grid1d((long long)H * W)), dim3(256), 0, ctx->stream, xflow, mask, H, W,
                       imu_tx, depth, conf);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"
