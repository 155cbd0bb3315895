// egopose.hip -- next-row N4, the core: relative camera pose (R, T) of two frames from point correspondences -- what the
// reference takes from `sfm2.getEgoMotion2{im1, im2, K, maxPoints, pointsQuality, ransacMaxDist, pointsMinDistance}` ->
// R, T, nFound, nInliers, fundmat (radial/radial_opticalflow_data.lua:211-217, radial/test_radial_opticalflow.lua:122-126;
// sfm2.getEgoMotion at depth_estimation_api.lua:141, test_opticalflow.lua:282).  sfm2 is an un-vendored OpenCV wrapper (corner
// detection + pyramidal LK tracks + a RANSAC fundamental matrix): nothing of it is in the repository, so this is restated from
// the calling convention and from what the results are used for -- e2 = K T (data.lua:218), removeEgoMotion(prev, K, R)
// (:231), nInliers / nFound against bad_image_threshold (:222) -- and parity is unpinned.
//
// MI355X form: RANSAC is embarrassingly parallel.  All `iterations` 8-point hypotheses are built at once (one thread each:
// the 9 x 9 normal matrix of 8 random correspondences, its null vector by cyclic Jacobi, the rank-2 / equal-singular-value
// projection onto the essential manifold), every hypothesis is scored against every correspondence by its own block (Sampson
// distance in pixels), the best one is refitted over its inliers (45 block-reduced sums -> one 9 x 9 eigenproblem on the
// host), decomposed into the four (R, T) candidates and disambiguated by cheirality.  Correspondences come either from the
// caller (dfe_ego_motion_from_points_f32: any tracker) or from the dense flow the matcher has just produced
// (dfe_ego_motion_from_flow_f32: p2 = p1 + flow(p1) on a regular grid of at most max_points samples).
// Convention: x2 ~ R x1 + T for camera coordinates of frame 1 (previous) and frame 2 (current), |T| = 1, E = [T]x R,
// F = K^-T E K^-1 with p2^T F p1 = 0; the epipole in the current frame is K T (what data.lua:218 computes).
#include "dfe_internal.h"
#include <cmath>
#include <vector>

namespace {

// ---- small dense algebra in double, host and device ---------------------------------------------------------------
// cyclic Jacobi for a symmetric N x N matrix: on return A's diagonal holds the eigenvalues, V's columns the eigenvectors
template <int N> __host__ __device__ inline void jacobi_sym(double (&A)[N][N], double (&V)[N][N]) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < N; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p][q];
                if (apq == 0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// E (row-major 9) -> the closest matrix with singular values (1, 1, 0): U diag(1,1,0) V^T; also returns U and V (columns) for the
// pose decomposition.  false when E is (numerically) of rank < 2.
__host__ __device__ inline bool essential_project(const double *E, double *Eo, double (*Uo)[3], double (*Vo)[3]) {
    double M[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = E[0 * 3 + i] * E[0 * 3 + j] + E[1 * 3 + i] * E[1 * 3 + j] + E[2 * 3 + i] * E[2 * 3 + j];   // E^T E
    jacobi_sym<3>(M, V);
    int o[3] = {0, 1, 2};                                  // eigenvalues descending
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (M[o[b]][o[b]] > M[o[a]][o[a]]) { const int t = o[a]; o[a] = o[b]; o[b] = t; }
    const double s1 = sqrt(fmax(M[o[0]][o[0]], 0.0)), s2 = sqrt(fmax(M[o[1]][o[1]], 0.0));
    if (!(s2 > 1e-9 * s1) || !(s1 > 0)) return false;
    double v[3][3], u[3][3];                               // v[k] = k-th right singular vector, u[k] = E v[k] / s_k
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i) v[k][i] = V[i][o[k]];
    // v3 = v1 x v2 (a proper right-handed triple whatever signs Jacobi chose)
    v[2][0] = v[0][1] * v[1][2] - v[0][2] * v[1][1];
    v[2][1] = v[0][2] * v[1][0] - v[0][0] * v[1][2];
    v[2][2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
    for (int k = 0; k < 2; ++k) {
        const double s = k == 0 ? s1 : s2;
        for (int i = 0; i < 3; ++i) u[k][i] = (E[i * 3] * v[k][0] + E[i * 3 + 1] * v[k][1] + E[i * 3 + 2] * v[k][2]) / s;
    }
    {   // re-orthonormalise u2 against u1 (E is noisy), u3 = u1 x u2
        double n1 = sqrt(u[0][0] * u[0][0] + u[0][1] * u[0][1] + u[0][2] * u[0][2]);
        for (int i = 0; i < 3; ++i) u[0][i] /= n1;
        const double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
        for (int i = 0; i < 3; ++i) u[1][i] -= d * u[0][i];
        double n2 = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
        if (!(n2 > 1e-12)) return false;
        for (int i = 0; i < 3; ++i) u[1][i] /= n2;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Eo[i * 3 + j] = u[0][i] * v[0][j] + u[1][i] * v[1][j];
    if (Uo)
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 3; ++i) { Uo[i][k] = u[k][i]; Vo[i][k] = v[k][i]; }
    return true;
}

__host__ __device__ inline void mat3_mul_d(const double *a, const double *b, double *o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
// F = Ki^T E Ki
__host__ __device__ inline void fund_from_essential(const double *E, const double *Ki, double *F) {
    double KiT[9], t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KiT[i * 3 + j] = Ki[j * 3 + i];
    mat3_mul_d(KiT, E, t);
    mat3_mul_d(t, Ki, F);
}
// squared Sampson distance of (p1 -> p2) to F, in pixels^2
__host__ __device__ inline double sampson2(const double *F, double x1, double y1, double x2, double y2) {
    const double a0 = F[0] * x1 + F[1] * y1 + F[2], a1 = F[3] * x1 + F[4] * y1 + F[5], a2 = F[6] * x1 + F[7] * y1 + F[8];   // F p1
    const double b0 = F[0] * x2 + F[3] * y2 + F[6], b1 = F[1] * x2 + F[4] * y2 + F[7];                                       // F^T p2
    const double r = x2 * a0 + y2 * a1 + a2;
    const double den = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
    return den > 0 ? r * r / den : 1e300;
}

// counter-based generator: the k-th draw of hypothesis h (no state to carry, the same sequence on any launch shape)
__host__ __device__ inline unsigned ego_rand(unsigned seed, unsigned h, unsigned k) {
    unsigned x = seed * 0x9E3779B1u ^ (h + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (k + 0x165667B1u) * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

struct EgoK { double Ki[9]; };

// one thread per hypothesis: 8 distinct valid correspondences -> E on the essential manifold -> F in pixels.  hypF[h][9] (all zero:
// no model, scores no inlier)
__global__ __launch_bounds__(64) void ego_hypotheses_kernel(const float *__restrict__ p1, const float *__restrict__ p2, const float *__restrict__ w, int N,
                                                           EgoK kk, unsigned seed, int nh, double *__restrict__ hypF) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= nh) return;
    double *Fo = hypF + (size_t)h * 9;
    for (int i = 0; i < 9; ++i) Fo[i] = 0;
    int pick[8];
    unsigned k = 0;
    for (int n = 0; n < 8; ++n) {
        int tries = 0;
        for (;; ++tries) {
            if (tries > 64) return;                        // too few valid correspondences
            const int c = (int)(ego_rand(seed, (unsigned)h, k++) % (unsigned)N);
            if (w && !(w[c] > 0.f)) continue;
            bool dup = false;
            for (int m = 0; m < n; ++m) dup = dup || pick[m] == c;
            if (!dup) { pick[n] = c; break; }
        }
    }
    double A[9][9], V[9][9];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) A[i][j] = 0;
    for (int n = 0; n < 8; ++n) {
        const double u1 = p1[2 * pick[n]], v1 = p1[2 * pick[n] + 1], u2 = p2[2 * pick[n]], v2 = p2[2 * pick[n] + 1];
        const double a[3] = {kk.Ki[0] * u1 + kk.Ki[1] * v1 + kk.Ki[2], kk.Ki[3] * u1 + kk.Ki[4] * v1 + kk.Ki[5], kk.Ki[6] * u1 + kk.Ki[7] * v1 + kk.Ki[8]};
        const double b[3] = {kk.Ki[0] * u2 + kk.Ki[1] * v2 + kk.Ki[2], kk.Ki[3] * u2 + kk.Ki[4] * v2 + kk.Ki[5], kk.Ki[6] * u2 + kk.Ki[7] * v2 + kk.Ki[8]};
        double r[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r[3 * i + j] = b[i] * a[j];            // x2^T E x1 = sum_ij x2_i E_ij x1_j
        for (int i = 0; i < 9; ++i)
            for (int j = 0; j < 9; ++j) A[i][j] += r[i] * r[j];
    }
    jacobi_sym<9>(A, V);
    int best = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i][i] < A[best][best]) best = i;
    double E[9], Ep[9];
    for (int i = 0; i < 9; ++i) E[i] = V[i][best];
    if (!essential_project(E, Ep, nullptr, nullptr)) return;
    fund_from_essential(Ep, kk.Ki, Fo);
}

// one block per hypothesis: inliers = valid correspondences within max_dist pixels (Sampson) of its F
__global__ __launch_bounds__(256) void ego_score_kernel(const float *__restrict__ p1, const float *__restrict__ p2, const float *__restrict__ w, int N,
                                                       const double *__restrict__ hypF, double max_d2, int *__restrict__ counts) {
    __shared__ int sm[4];
    double F[9];
    for (int i = 0; i < 9; ++i) F[i] = hypF[(size_t)blockIdx.x * 9 + i];
    int c = 0;
    const bool model = F[0] != 0 || F[1] != 0 || F[2] != 0 || F[3] != 0 || F[4] != 0 || F[5] != 0 || F[6] != 0 || F[7] != 0 || F[8] != 0;
    if (model)
        for (int n = threadIdx.x; n < N; n += 256) {
            if (w && !(w[n] > 0.f)) continue;
            c += sampson2(F, p1[2 * n], p1[2 * n + 1], p2[2 * n], p2[2 * n + 1]) <= max_d2;
        }
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// refit: the 45 unique entries of sum r r^T over the inliers of F (r = the 9 products x2_i x1_j in camera coordinates), per block;
// also the inlier mask.  part[block][46] (last = the count)
__global__ __launch_bounds__(256) void ego_refit_sums_kernel(const float *__restrict__ p1, const float *__restrict__ p2, const float *__restrict__ w, int N,
                                                            EgoK kk, const double *__restrict__ Fd, double max_d2, unsigned char *__restrict__ mask,
                                                            double *__restrict__ part) {
    __shared__ double sm[46][4];
    double F[9];
    for (int i = 0; i < 9; ++i) F[i] = Fd[i];
    double s[46];
    for (int i = 0; i < 46; ++i) s[i] = 0;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
        const double u1 = p1[2 * n], v1 = p1[2 * n + 1], u2 = p2[2 * n], v2 = p2[2 * n + 1];
        const bool in = (!w || w[n] > 0.f) && sampson2(F, u1, v1, u2, v2) <= max_d2;
        mask[n] = in ? 1 : 0;
        if (!in) continue;
        const double a[3] = {kk.Ki[0] * u1 + kk.Ki[1] * v1 + kk.Ki[2], kk.Ki[3] * u1 + kk.Ki[4] * v1 + kk.Ki[5], kk.Ki[6] * u1 + kk.Ki[7] * v1 + kk.Ki[8]};
        const double b[3] = {kk.Ki[0] * u2 + kk.Ki[1] * v2 + kk.Ki[2], kk.Ki[3] * u2 + kk.Ki[4] * v2 + kk.Ki[5], kk.Ki[6] * u2 + kk.Ki[7] * v2 + kk.Ki[8]};
        double r[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r[3 * i + j] = b[i] * a[j];
        int k = 0;
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) s[k++] += r[i] * r[j];
        s[45] += 1.0;
    }
    for (int k = 0; k < 46; ++k) {
        double v = s[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) sm[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 46) part[(size_t)blockIdx.x * 46 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

// correspondences from a dense flow field: sample (gy, gx) of a regular grid -> p1 = (x, y), p2 = p1 + flow, w = 1 where the
// sample is usable (conf > 0 when given, finite flow, both points inside the frame)
__global__ void ego_sample_flow_kernel(const float *__restrict__ fy, const float *__restrict__ fx, const float *__restrict__ conf, int H, int W, int step,
                                       int y0, int x0, int gh, int gw, float *__restrict__ p1, float *__restrict__ p2, float *__restrict__ w) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= gh * gw) return;
    const int gy = n / gw, gx = n - gy * gw;
    const int y = y0 + gy * step, x = x0 + gx * step;
    const long long e = (long long)y * W + x;
    const float u = fx[e], v = fy[e];
    const float X = (float)x + u, Y = (float)y + v;
    const bool ok = (!conf || conf[e] > 0.f) && u == u && v == v && X >= 0.f && X <= (float)(W - 1) && Y >= 0.f && Y <= (float)(H - 1);
    p1[2 * n] = (float)x; p1[2 * n + 1] = (float)y;
    p2[2 * n] = X; p2[2 * n + 1] = Y;
    w[n] = ok ? 1.f : 0.f;
}

bool mat3_inv_d(const double *m, double *o) {
    const double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (fabs(d) < 1e-300) return false;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return true;
}

// number of inliers in front of both cameras for the pose (R, t): x2 ~ R x1 + t.  Depths from the two-view linear system
// z2 b = z1 R a + t, solved in the least-squares sense per point.
int cheirality(const double *R, const double *t, const std::vector<float> &p1, const std::vector<float> &p2, const std::vector<unsigned char> &mask,
               const double *Ki) {
    int good = 0;
    for (size_t n = 0; n < mask.size(); ++n) {
        if (!mask[n]) continue;
        const double u1 = p1[2 * n], v1 = p1[2 * n + 1], u2 = p2[2 * n], v2 = p2[2 * n + 1];
        const double a[3] = {Ki[0] * u1 + Ki[1] * v1 + Ki[2], Ki[3] * u1 + Ki[4] * v1 + Ki[5], Ki[6] * u1 + Ki[7] * v1 + Ki[8]};
        const double b[3] = {Ki[0] * u2 + Ki[1] * v2 + Ki[2], Ki[3] * u2 + Ki[4] * v2 + Ki[5], Ki[6] * u2 + Ki[7] * v2 + Ki[8]};
        const double ra[3] = {R[0] * a[0] + R[1] * a[1] + R[2] * a[2], R[3] * a[0] + R[4] * a[1] + R[5] * a[2], R[6] * a[0] + R[7] * a[1] + R[8] * a[2]};
        // [ra  -b] [z1 z2]^T = -t  ->  normal equations
        const double m00 = ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2], m01 = -(ra[0] * b[0] + ra[1] * b[1] + ra[2] * b[2]);
        const double m11 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        const double r0 = -(ra[0] * t[0] + ra[1] * t[1] + ra[2] * t[2]), r1 = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
        const double det = m00 * m11 - m01 * m01;
        if (fabs(det) < 1e-18) continue;
        const double z1 = (r0 * m11 - m01 * r1) / det, z2 = (m00 * r1 - m01 * r0) / det;
        good += z1 > 0 && z2 > 0;
    }
    return good;
}

// arena layout behind `scr_off` bytes of the caller's own data: hypotheses' F, their inlier counts, the best F, the inlier mask, the
// refit partial sums
struct EgoLayout { size_t off_f, off_c, off_b, off_m, off_p, total; int nblk; };
EgoLayout ego_layout(int N, int nh, size_t scr_off) {
    EgoLayout l;
    l.nblk = N < 256 * 64 ? (N + 255) / 256 : 64;
    l.off_f = (scr_off + 255) / 256 * 256;
    l.off_c = l.off_f + ((size_t)nh * 9 * 8 + 255) / 256 * 256;
    l.off_b = l.off_c + ((size_t)nh * 4 + 255) / 256 * 256;
    l.off_m = l.off_b + 256;
    l.off_p = l.off_m + ((size_t)N + 255) / 256 * 256;
    l.total = l.off_p + (size_t)l.nblk * 46 * 8;
    return l;
}

int ego_from_points(dfe_ctx *ctx, const float *p1, const float *p2, const float *w, int N, const double *K9, double max_dist, int iterations, unsigned seed,
                    double *R9, double *T3, int *n_inliers, double *F9, size_t scr_off) {
    EgoK kk;
    DFE_REQUIRE(ctx, mat3_inv_d(K9, kk.Ki), DFE_E_ARG, "ego motion: K is singular");
    const int nh = iterations;
    const EgoLayout lay = ego_layout(N, nh, scr_off);
    const int nblk = lay.nblk;
    const size_t off_f = lay.off_f, off_c = lay.off_c, off_b = lay.off_b, off_m = lay.off_m, off_p = lay.off_p;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, lay.total, &scr);
    if (rc) return rc;
    // (a caller that placed its correspondences in the arena -- the flow sampler -- passes their extent as scr_off and has reserved
    //  lay.total itself: the arena only ever grows, so this call does not move it)
    double *hypF = (double *)((char *)scr + off_f);
    int *counts = (int *)((char *)scr + off_c);
    double *bestF = (double *)((char *)scr + off_b);
    unsigned char *mask = (unsigned char *)scr + off_m;
    double *part = (double *)((char *)scr + off_p);
    hipLaunchKernelGGL(ego_hypotheses_kernel, dim3((nh + 63) / 64), dim3(64), 0, ctx->stream, p1, p2, w, N, kk, seed, nh, hypF);
    hipLaunchKernelGGL(ego_score_kernel, dim3(nh), dim3(256), 0, ctx->stream, p1, p2, w, N, (const double *)hypF, max_dist * max_dist, counts);
    DFE_LAUNCH_CHECK(ctx);
    std::vector<int> hc(nh);
    DFE_HIP(ctx, hipMemcpyAsync(hc.data(), counts, (size_t)nh * 4, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int best = 0;
    for (int h = 1; h < nh; ++h)
        if (hc[h] > hc[best]) best = h;                    // ties: the first hypothesis (deterministic)
    DFE_REQUIRE(ctx, hc[best] >= 8, DFE_E_ARG, "ego motion: no hypothesis with 8 inliers (%d correspondences, best %d)", N, hc[best]);
    DFE_HIP(ctx, hipMemcpyAsync(bestF, hypF + (size_t)best * 9, 72, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(ego_refit_sums_kernel, dim3(nblk), dim3(256), 0, ctx->stream, p1, p2, w, N, kk, (const double *)bestF, max_dist * max_dist, mask, part);
    DFE_LAUNCH_CHECK(ctx);
    std::vector<double> hp((size_t)nblk * 46);
    std::vector<unsigned char> hm(N);
    std::vector<float> h1((size_t)2 * N), h2((size_t)2 * N);
    DFE_HIP(ctx, hipMemcpyAsync(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipMemcpyAsync(hm.data(), mask, (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipMemcpyAsync(h1.data(), p1, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipMemcpyAsync(h2.data(), p2, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double A[9][9], V[9][9];
    {
        double s[46] = {0};
        for (int b = 0; b < nblk; ++b)
            for (int k = 0; k < 46; ++k) s[k] += hp[(size_t)b * 46 + k];
        int k = 0;
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) { A[i][j] = A[j][i] = s[k]; ++k; }
    }
    jacobi_sym<9>(A, V);
    int mn = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i][i] < A[mn][mn]) mn = i;
    double E[9], Ep[9], U[3][3], Vv[3][3];
    for (int i = 0; i < 9; ++i) E[i] = V[i][mn];
    DFE_REQUIRE(ctx, essential_project(E, Ep, U, Vv), DFE_E_ARG, "ego motion: degenerate configuration (the refitted E has rank < 2)");
    // E = U diag(1,1,0) V^T with det(U) = det(V) = +1 (both built as right-handed triples): R = U W V^T or U W^T V^T, T = +-u3
    const double Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Ur[9], VT[9], t1[9], Ra[9], Rb[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Ur[i * 3 + j] = U[i][j]; VT[i * 3 + j] = Vv[j][i]; }
    mat3_mul_d(Ur, Wm, t1); mat3_mul_d(t1, VT, Ra);
    mat3_mul_d(Ur, Wt, t1); mat3_mul_d(t1, VT, Rb);
    const double u3[3] = {U[0][2], U[1][2], U[2][2]}, nu3[3] = {-u3[0], -u3[1], -u3[2]};
    const double *Rs[4] = {Ra, Ra, Rb, Rb};
    const double *Ts[4] = {u3, nu3, u3, nu3};
    int bc = -1, bg = -1;
    for (int c = 0; c < 4; ++c) {
        const int g = cheirality(Rs[c], Ts[c], h1, h2, hm, kk.Ki);
        if (g > bg) { bg = g; bc = c; }
    }
    for (int i = 0; i < 9; ++i) R9[i] = Rs[bc][i];
    for (int i = 0; i < 3; ++i) T3[i] = Ts[bc][i];
    double F[9];
    fund_from_essential(Ep, kk.Ki, F);
    double fn = 0;
    for (int i = 0; i < 9; ++i) fn += F[i] * F[i];
    fn = sqrt(fn);
    // nInliers: the correspondences of the winning hypothesis' consensus set that the REFITTED model keeps within max_dist
    int cnt = 0;
    for (int n = 0; n < N; ++n)
        if (hm[n]) cnt += sampson2(F, h1[2 * n], h1[2 * n + 1], h2[2 * n], h2[2 * n + 1]) <= max_dist * max_dist;
    if (n_inliers) *n_inliers = cnt;
    if (F9)
        for (int i = 0; i < 9; ++i) F9[i] = F[i] / (fn > 0 ? fn : 1);
    return DFE_OK;
}

}  // namespace

extern "C" {

int dfe_ego_motion_from_points_f32(dfe_ctx *ctx, const float *pts1, const float *pts2, const float *weights, int N, const double *K9, double ransac_max_dist,
                                   int iterations, unsigned seed, double *R9, double *T3, int *n_inliers, double *F9) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, pts1 && pts2 && K9 && R9 && T3, DFE_E_ARG, "dfe_ego_motion_from_points_f32: NULL argument");
    DFE_REQUIRE(ctx, N >= 8 && iterations >= 1 && iterations <= 65536 && ransac_max_dist > 0, DFE_E_ARG,
                "dfe_ego_motion_from_points_f32: N=%d (>= 8) iterations=%d (1..65536) ransac_max_dist=%g", N, iterations, ransac_max_dist);
    return ego_from_points(ctx, pts1, pts2, weights, N, K9, ransac_max_dist, iterations, seed, R9, T3, n_inliers, F9, 0);
}

int dfe_ego_motion_from_flow_f32(dfe_ctx *ctx, const float *flow_y, const float *flow_x, const float *conf, int H, int W, const double *K9, int max_points,
                                 double ransac_max_dist, int iterations, unsigned seed, double *R9, double *T3, int *n_found, int *n_inliers, double *F9) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, flow_y && flow_x && K9 && R9 && T3, DFE_E_ARG, "dfe_ego_motion_from_flow_f32: NULL argument");
    DFE_REQUIRE(ctx, H >= 8 && W >= 8 && max_points >= 8 && iterations >= 1 && iterations <= 65536 && ransac_max_dist > 0, DFE_E_ARG,
                "dfe_ego_motion_from_flow_f32: %dx%d max_points=%d iterations=%d ransac_max_dist=%g", H, W, max_points, iterations, ransac_max_dist);
    int step = (int)ceil(sqrt((double)H * W / max_points));
    if (step < 1) step = 1;
    const int gh = (H - 1) / step + 1, gw = (W - 1) / step + 1;
    const int y0 = ((H - 1) - (gh - 1) * step) / 2, x0 = ((W - 1) - (gw - 1) * step) / 2;    // the grid centred in the frame
    const int N = gh * gw;
    const size_t pts_bytes = ((size_t)N * 5 * 4 + 255) / 256 * 256;
    // reserve the whole arena first (samples + what ego_from_points lays out behind them), so that its own dfe_scratch call cannot move it
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, ego_layout(N, iterations, pts_bytes).total, &scr);
    if (rc) return rc;
    float *p1 = (float *)scr, *p2 = p1 + 2 * (size_t)N, *w = p2 + 2 * (size_t)N;
    hipLaunchKernelGGL(ego_sample_flow_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, flow_y, flow_x, conf, H, W, step, y0, x0, gh, gw, p1, p2, w);
    DFE_LAUNCH_CHECK(ctx);
    if (n_found) {
        std::vector<float> hw(N);
        DFE_HIP(ctx, hipMemcpyAsync(hw.data(), w, (size_t)N * 4, hipMemcpyDeviceToHost, ctx->stream));
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int c = 0;
        for (int n = 0; n < N; ++n) c += hw[n] > 0.f;
        *n_found = c;
        DFE_REQUIRE(ctx, c >= 8, DFE_E_ARG, "dfe_ego_motion_from_flow_f32: only %d usable flow samples", c);
    }
    return ego_from_points(ctx, p1, p2, w, N, K9, ransac_max_dist, iterations, seed, R9, T3, n_inliers, F9, pts_bytes);
}

}  // extern "C"
