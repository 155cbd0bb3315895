// egomotion.hip -- next-row N4: the step in front of the polar warp (radial/radial_opticalflow_data.lua:211-231,
// depth_estimation_api.lua:139-147, test_opticalflow.lua:280-284).  The reference takes it from the un-vendored, OpenCV-backed
// `sfm2` package: undistortImage(img, K, distP), getEgoMotion(2)(im1, im2, K, ...) -> R, T, removeEgoMotion(img, K, R) -> warped,
// mask, and computes the epipole e2 = K T / (K T)_3 itself (data.lua:218-219).  Nothing of sfm2 is in the repository, so these
// are restated from the calling convention and from what the functions must do (parity unpinned):
//   dfe_undistort_image_f32     the radial-tangential (k1, k2, p1, p2, k3) model of the 5-entry `distortion` vectors in the
//                               .cal files: out(p) = bilinear(img, distort(K^-1 p)) -- the usual inverse-map undistortion
//   dfe_remove_ego_motion_f32   rotation-only warp: out(p) = bilinear(img, K R K^-1 p), mask(p) = 1 where the source lies
//                               inside the frame (the callers zero the mask border and polar-warp it, data.lua:233-240)
//   dfe_epipole                 e2 = K T / (K T)_3 scaled to the working resolution (data.lua:218-220), host arithmetic
//   dfe_foe_from_flow_f32       focus of expansion of a dense flow field: the point minimising the weighted squared
//                               distances to the lines (p, flow(p)), a 2 x 2 normal system from five wave/block-reduced
//                               sums, re-weighted twice against outliers.  The MI355X-native stand-in for the sparse
//                               LK-tracks + RANSAC of sfm2.getEgoMotion2 when the motion is (rectified to) a pure
//                               translation: it consumes the dense flow the matcher already produced.
#include "dfe_internal.h"
#include <cmath>

namespace {

int grid_e(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float bilin_masked(const float *__restrict__ p, int H, int W, float fy, float fx) {
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

struct Mat3 { float m[9]; };

// out(y, x) = img(Hm * (x, y, 1)); mask = source inside [0, W-1] x [0, H-1]
__global__ void homography_warp_kernel(const float *__restrict__ img, int C, int H, int W, Mat3 Hm, float *__restrict__ out, float *__restrict__ mask) {
    const long long P = (long long)H * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(e / W), x = (int)(e - (long long)y * W);
        const float X = Hm.m[0] * x + Hm.m[1] * y + Hm.m[2], Y = Hm.m[3] * x + Hm.m[4] * y + Hm.m[5], Z = Hm.m[6] * x + Hm.m[7] * y + Hm.m[8];
        const float sx = X / Z, sy = Y / Z;
        const bool in = Z > 0 && sx >= 0 && sx <= (float)(W - 1) && sy >= 0 && sy <= (float)(H - 1);
        for (int c = 0; c < C; ++c) out[c * P + e] = in ? bilin_masked(img + c * P, H, W, sy, sx) : 0.f;
        if (mask) mask[e] = in ? 1.f : 0.f;
    }
}

__global__ void undistort_kernel(const float *__restrict__ img, int C, int H, int W, float fx, float fy, float cx, float cy, float k1, float k2,
                                 float p1, float p2, float k3, float *__restrict__ out) {
    const long long P = (long long)H * W;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(e / W), x = (int)(e - (long long)y * W);
        const float xn = ((float)x - cx) / fx, yn = ((float)y - cy) / fy;
        const float r2 = xn * xn + yn * yn;
        const float rad = 1.f + r2 * (k1 + r2 * (k2 + r2 * k3));
        const float xd = xn * rad + 2.f * p1 * xn * yn + p2 * (r2 + 2.f * xn * xn);
        const float yd = yn * rad + p1 * (r2 + 2.f * yn * yn) + 2.f * p2 * xn * yn;
        const float sx = xd * fx + cx, sy = yd * fy + cy;
        const bool in = sx >= 0 && sx <= (float)(W - 1) && sy >= 0 && sy <= (float)(H - 1);
        for (int c = 0; c < C; ++c) out[c * P + e] = in ? bilin_masked(img + c * P, H, W, sy, sx) : 0.f;
    }
}

// partial sums of the FOE normal equations over pixels with |flow| >= min_flow and conf > 0:
//   n = (-v, u)/|flow| (unit normal of the flow line through p), residual r = n . (c - p), weight w
//   A = sum w n n^T, b = sum w n (n . p)  ->  5 doubles per block: Axx, Axy, Ayy, bx, by
__global__ __launch_bounds__(256) void foe_sums_kernel(const float *__restrict__ fy, const float *__restrict__ fx, const float *__restrict__ conf, int H,
                                                      int W, float min_flow, float cx, float cy, float huber, int use_center, double *__restrict__ part) {
    __shared__ double sm[5][4];
    double s[5] = {0, 0, 0, 0, 0};
    const long long P = (long long)H * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < P; e += (long long)gridDim.x * 256) {
        const float u = fx[e], v = fy[e];
        const float mag = sqrtf(u * u + v * v);
        if (mag < min_flow || (conf && conf[e] <= 0.f)) continue;
        const int y = (int)(e / W), x = (int)(e - (long long)y * W);
        const double nx = -v / mag, ny = u / mag;
        double w = 1.0;
        if (use_center) {
            const double r = fabs(nx * ((double)cx - x) + ny * ((double)cy - y));
            w = r <= huber ? 1.0 : huber / r;               // Huber re-weighting against outliers
        }
        const double np = nx * x + ny * y;
        s[0] += w * nx * nx; s[1] += w * nx * ny; s[2] += w * ny * ny; s[3] += w * nx * np; s[4] += w * ny * np;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double v = s[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) part[blockIdx.x * 5 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

void mat3_mul(const double *a, const double *b, double *o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

bool mat3_inv(const double *m, double *o) {
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (fabs(d) < 1e-300) return false;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return true;
}

}  // namespace

extern "C" {

int dfe_epipole(const double *K9, const double *T3, double scale, double *e2) {
    if (!K9 || !T3 || !e2) return DFE_E_ARG;
    const double x = K9[0] * T3[0] + K9[1] * T3[1] + K9[2] * T3[2], y = K9[3] * T3[0] + K9[4] * T3[1] + K9[5] * T3[2],
                 z = K9[6] * T3[0] + K9[7] * T3[1] + K9[8] * T3[2];
    if (z == 0) return DFE_E_ARG;                    // translation parallel to the image plane: the epipole is at infinity
    e2[0] = x / z * scale;                           // e2 = K T; e2 = e2 / e2[3]; e2 = e2 * wImg / calibration.wImg  (data.lua:218-220)
    e2[1] = y / z * scale;
    return DFE_OK;
}

int dfe_remove_ego_motion_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *R9, int inverse, float *out,
                              float *mask) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, img && K9 && R9 && out, DFE_E_ARG, "dfe_remove_ego_motion_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0, DFE_E_SHAPE, "dfe_remove_ego_motion_f32: C=%d %dx%d", C, H, W);
    double Ki[9], Rt[9], t[9], Hd[9];
    DFE_REQUIRE(ctx, mat3_inv(K9, Ki), DFE_E_ARG, "dfe_remove_ego_motion_f32: K is singular");
    const double *Ru = R9;
    if (inverse) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = R9[j * 3 + i];
        Ru = Rt;
    }
    mat3_mul(K9, Ru, t);
    mat3_mul(t, Ki, Hd);
    Mat3 Hm;
    for (int i = 0; i < 9; ++i) Hm.m[i] = (float)Hd[i];
    hipLaunchKernelGGL(homography_warp_kernel, dim3(grid_e((long long)H * W)), dim3(256), 0, ctx->stream, img, C, H, W, Hm, out, mask);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_undistort_image_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, const double *K9, const double *dist5, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, img && K9 && dist5 && out, DFE_E_ARG, "dfe_undistort_image_f32: NULL argument");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && K9[0] != 0 && K9[4] != 0, DFE_E_SHAPE, "dfe_undistort_image_f32: C=%d %dx%d fx=%g fy=%g", C, H, W, K9[0], K9[4]);
    hipLaunchKernelGGL(undistort_kernel, dim3(grid_e((long long)H * W)), dim3(256), 0, ctx->stream, img, C, H, W, (float)K9[0], (float)K9[4], (float)K9[2],
                       (float)K9[5], (float)dist5[0], (float)dist5[1], (float)dist5[2], (float)dist5[3], (float)dist5[4], out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_foe_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W, float min_flow, int iterations,
                          double *foe_xy, double *n_used) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, flow_y && flow_x && foe_xy, DFE_E_ARG, "dfe_foe_from_flow_f32: NULL argument");
    DFE_REQUIRE(ctx, H > 0 && W > 0 && iterations >= 0 && iterations <= 16, DFE_E_SHAPE, "dfe_foe_from_flow_f32: %dx%d, %d iterations", H, W, iterations);
    const int nb = grid_e((long long)H * W) > 256 ? 256 : grid_e((long long)H * W);
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (size_t)nb * 5 * sizeof(double), &scr);
    if (rc) return rc;
    std::vector<double> host((size_t)nb * 5);
    double cx = W / 2.0, cy = H / 2.0;
    for (int it = 0; it <= iterations; ++it) {
        hipLaunchKernelGGL(foe_sums_kernel, dim3(nb), dim3(256), 0, ctx->stream, flow_y, flow_x, conf, H, W, min_flow, (float)cx, (float)cy, 2.0f, it > 0,
                           (double *)scr);
        DFE_LAUNCH_CHECK(ctx);
        DFE_HIP(ctx, hipMemcpyAsync(host.data(), scr, host.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double s[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 5; ++k) s[k] += host[(size_t)b * 5 + k];
        const double det = s[0] * s[2] - s[1] * s[1];
        if (n_used) *n_used = s[0] + s[2];           // sum of the weights (n is a unit vector)
        DFE_REQUIRE(ctx, fabs(det) > 1e-9 * (s[0] + s[2]) * (s[0] + s[2]) + 1e-300, DFE_E_ARG,
                    "dfe_foe_from_flow_f32: the flow lines do not intersect in a point (parallel flow or too few vectors)");
        cx = (s[2] * s[3] - s[1] * s[4]) / det;
        cy = (s[0] * s[4] - s[1] * s[3]) / det;
    }
    foe_xy[0] = cx;
    foe_xy[1] = cy;
    return DFE_OK;
}

}  // extern "C"
