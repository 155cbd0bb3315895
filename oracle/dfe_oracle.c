/*
 * dfe_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see dfe_oracle.h for the rules and the pinning status: the reference's native files
 * cannot be built here, so this restatement is pinned by the reference's own data-free
 * known-answer tests; pieces living in un-vendored nnx/nn are "parity unpinned").
 *
 * Build (reference flags, /root/reference Makefile:2):
 *   gcc -O3 -funroll-loops -fopenmp -fPIC -shared -o libdfe_oracle.so dfe_oracle.c -lm
 * -ffast-math is deliberately NOT used: summation order is part of the contract.
 */
#include "dfe_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
void orc_set_num_threads(int n) { g_threads = n; }
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
#ifdef _OPENMP
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(orc_get_max_threads())")
#else
#define ORC_PAR_FOR
#endif

static inline int ifloor_div2(int a) { return a >> 1; }           /* floor(a/2), a>=0 */
static inline int iceil_div2(int a) { return (a + 1) >> 1; }      /* ceil(a/2),  a>=0 */
/* ref: common.lua:24-26 round(x)=floor(x+0.5) */
static inline double lua_round(double x) { return floor(x + 0.5); }

/* ---- A0 ---------------------------------------------------------------- */
void orc_unfold(const float *img, int C, int H, int W, int kh, int kw, float *out) {
    /* ref: radial/radial_opticalflow_groundtruth.lua:9-21: imgc[{{},i,j}] = reshape(C x hKer x wKer) */
    int Ho = H - kh + 1, Wo = W - kw + 1;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
                float *o = out + (size_t)((c * kh + i) * kw + j) * Ho * Wo;
                for (int y = 0; y < Ho; ++y)
                    for (int x = 0; x < Wo; ++x)
                        o[(size_t)y * Wo + x] = img[((size_t)c * H + y + i) * W + x + j];
            }
}

/* ---- A1 ---------------------------------------------------------------- */
void orc_spatial_matching(const float *in1, const float *in2, int K, int H1, int W1,
                          int maxh, int maxw, float *out) {
    /* out[y][x][dy][dx] = sum_k (in1[k][y][x] - in2[k][y+dy][x+dx])^2, accumulated in
     * float in feature order (nnx `real`=float). Layout pinned by opticalflow_model.lua:98
     * (SmartReshape({-1,-2},{-3,-4})) and the index convention by tests/test_multiscale.lua:149-166. */
    int H2 = H1 + maxh - 1, W2 = W1 + maxw - 1;
    ORC_PAR_FOR
    for (int y = 0; y < H1; ++y)
        for (int x = 0; x < W1; ++x)
            for (int dy = 0; dy < maxh; ++dy)
                for (int dx = 0; dx < maxw; ++dx) {
                    float s = 0.f;
                    for (int k = 0; k < K; ++k) {
                        float d = in1[((size_t)k * H1 + y) * W1 + x] -
                                  in2[((size_t)k * H2 + y + dy) * W2 + x + dx];
                        s += d * d;
                    }
                    out[(((size_t)y * W1 + x) * maxh + dy) * maxw + dx] = s;
                }
    (void)H2;
}

/* ---- A15 (next-row N1): the learned patch-feature stack ------------------------------------------ */
/* nn.SpatialConvolution(nIn, nOut, kW, kH):updateOutput -- valid cross-correlation + bias.  ref call sites:
 * opticalflow_model.lua:45-79 (getFilter), radial/radial_opticalflow_network.lua:6-30.  [3P nn: out = bias, then for
 * every input plane out += xcorr2(in_i, w[o][i]) -- per output element that is the order i, then ky, then kx; float.] */
void orc_spatial_convolution(const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                             int kH, int kW, float *out) {
    int Ho = H - kH + 1, Wo = W - kW + 1;
    ORC_PAR_FOR
    for (int o = 0; o < nOut; ++o)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float s = bias ? bias[o] : 0.f;
                for (int i = 0; i < nIn; ++i)
                    for (int u = 0; u < kH; ++u)
                        for (int v = 0; v < kW; ++v)
                            s += weight[(((size_t)o * nIn + i) * kH + u) * kW + v] * in[((size_t)i * H + y + u) * W + x + v];
                out[((size_t)o * Ho + y) * Wo + x] = s;
            }
}
/* The same convolution with FUSED multiply-adds in the same (i, u, v) order -- what dfe_spatial_convolution_mfma_f32 computes
 * (v_mfma_f32_16x16x4_f32 is an fmaf chain over k): test infrastructure for that kernel only. */
void orc_spatial_convolution_fma(const float *in, const float *weight, const float *bias, int nIn, int nOut, int H, int W,
                                 int kH, int kW, float *out) {
    int Ho = H - kH + 1, Wo = W - kW + 1;
    ORC_PAR_FOR
    for (int o = 0; o < nOut; ++o)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float s = bias ? bias[o] : 0.f;
                for (int i = 0; i < nIn; ++i)
                    for (int u = 0; u < kH; ++u)
                        for (int v = 0; v < kW; ++v)
                            s = fmaf(in[((size_t)i * H + y + u) * W + x + v], weight[(((size_t)o * nIn + i) * kH + u) * kW + v], s);
                out[((size_t)o * Ho + y) * Wo + x] = s;
            }
}
/* nn.SpatialConvolutionMap(connTable, kW, kH): connection c = (from, to) 1-based, one kH x kW kernel per connection,
 * accumulated into out[to] in table order. ref: opticalflow_model.lua:56-59 (nn.tables.random fan-in tables) */
void orc_spatial_convolution_map(const float *in, const float *weight, const float *bias, const int *conn, int nConn,
                                 int nIn, int nOut, int H, int W, int kH, int kW, float *out) {
    int Ho = H - kH + 1, Wo = W - kW + 1;
    (void)nIn;
    ORC_PAR_FOR
    for (int o = 0; o < nOut; ++o)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float s = bias ? bias[o] : 0.f;
                for (int c = 0; c < nConn; ++c) {
                    if (conn[2 * c + 1] - 1 != o) continue;
                    int i = conn[2 * c] - 1;
                    for (int u = 0; u < kH; ++u)
                        for (int v = 0; v < kW; ++v)
                            s += weight[((size_t)c * kH + u) * kW + v] * in[((size_t)i * H + y + u) * W + x + v];
                }
                out[((size_t)o * Ho + y) * Wo + x] = s;
            }
}
void orc_tanh(const float *in, int64_t n, float *out) { /* nn.Tanh: tanhf */
    for (int64_t i = 0; i < n; ++i) out[i] = tanhf(in[i]);
}

/* nn.SpatialContrastiveNormalization(nIn, kernel1d, threshold, thresval) = SpatialSubtractiveNormalization followed by
 * SpatialDivisiveNormalization with a 1-D kernel (version2/network.lua:12: image.gaussian1D(normalization_k)).
 * [3P-recall: nn is un-vendored and nothing in the reference tests this module -- parity unpinned.]  As recalled from Torch7:
 *   kn = kernel / (sum(kernel) * nIn); estimator(x) = zero-pad by floor(k/2) (ceil-1 on the far side for even k), a horizontal
 *   pass per plane (SpatialConvolutionMap one-to-one, taps in order), a vertical pass that also sums the planes
 *   (SpatialConvolution nIn -> 1: planes, then taps); coef = estimator(ones): the border correction.
 *   subtractive: y = x - estimator(x) / coef (the one-plane mean replicated over the planes)
 *   divisive:    s = sqrt(estimator(y^2)) / coef; s = s > threshold ? s : thresval; out = y / s                        */
static void cn_estimator(const float *in, int C, int H, int W, const float *kn, int k, float *tmp, float *out) {
    int pl = k / 2, pr = k - 1 - pl; /* SpatialZeroPadding(floor(k/2), ceil(k/2)-1 ...) so that the output keeps H x W */
    (void)pr;
    ORC_PAR_FOR
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float s = 0.f;
                for (int v = 0; v < k; ++v) {
                    int xx = x + v - pl;
                    float a = (xx >= 0 && xx < W) ? in[((size_t)c * H + y) * W + xx] : 0.f;
                    s += kn[v] * a;
                }
                tmp[((size_t)c * H + y) * W + x] = s;
            }
    ORC_PAR_FOR
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float s = 0.f;
            for (int c = 0; c < C; ++c)
                for (int u = 0; u < k; ++u) {
                    int yy = y + u - pl;
                    float a = (yy >= 0 && yy < H) ? tmp[((size_t)c * H + yy) * W + x] : 0.f;
                    s += kn[u] * a;
                }
            out[(size_t)y * W + x] = s;
        }
}
void orc_contrastive_normalization(const float *in, int C, int H, int W, const float *kernel, int k, float threshold,
                                   float thresval, float *out) {
    size_t P = (size_t)H * W;
    float *kn = (float *)malloc(sizeof(float) * k), *tmp = (float *)malloc(sizeof(float) * C * P);
    float *ones = (float *)malloc(sizeof(float) * C * P), *coef = (float *)malloc(sizeof(float) * P);
    float *est = (float *)malloc(sizeof(float) * P), *y = (float *)malloc(sizeof(float) * C * P), *sq = (float *)malloc(sizeof(float) * C * P);
    float ks = 0.f;
    for (int i = 0; i < k; ++i) ks += kernel[i];
    for (int i = 0; i < k; ++i) kn[i] = kernel[i] / (ks * (float)C);
    for (size_t i = 0; i < (size_t)C * P; ++i) ones[i] = 1.f;
    cn_estimator(ones, C, H, W, kn, k, tmp, coef);
    cn_estimator(in, C, H, W, kn, k, tmp, est);
    for (int c = 0; c < C; ++c)
        for (size_t p = 0; p < P; ++p) {
            float v = in[c * P + p] - est[p] / coef[p];
            y[c * P + p] = v;
            sq[c * P + p] = v * v;
        }
    cn_estimator(sq, C, H, W, kn, k, tmp, est);
    for (size_t p = 0; p < P; ++p) {
        float sd = sqrtf(est[p]) / coef[p];
        est[p] = sd > threshold ? sd : thresval;
    }
    for (int c = 0; c < C; ++c)
        for (size_t p = 0; p < P; ++p) out[c * P + p] = y[c * P + p] / est[p];
    free(kn); free(tmp); free(ones); free(coef); free(est); free(y); free(sq);
}

/* N2: gradients of the filter stack (un-vendored nn; pinned as the Jacobian of the forward above, the way
 * tests/test_cascad.lua:21-25 pins the cascade).  conn == NULL: dense nn.SpatialConvolution, weight [nOut][nIn][kH][kW];
 * else nn.SpatialConvolutionMap, weight [nConn][kH][kW], conn (from, to) 1-based.
 *   gradIn[i][y][x]      = sum_o sum_u sum_v w[o][i][u][v] gO[o][y-u][x-v]        (terms in that order)
 *   gradW[o][i][u][v]   += scale * sum_{y,x} gO[o][y][x] in[i][y+u][x+v]           (row-major float sum)
 *   gradB[o]            += scale * sum_{y,x} gO[o][y][x] */
void orc_spatial_convolution_grad_input(const float *go, const float *weight, const int *conn, int nConn, int nIn, int nOut,
                                        int H, int W, int kH, int kW, float *gi) {
    int Ho = H - kH + 1, Wo = W - kW + 1;
    ORC_PAR_FOR
    for (int i = 0; i < nIn; ++i)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float s = 0.f;
                int nq = conn ? nConn : nOut;
                for (int q = 0; q < nq; ++q) {
                    int o = q;
                    if (conn) {
                        if (conn[2 * q] - 1 != i) continue;
                        o = conn[2 * q + 1] - 1;
                    }
                    const float *wk = conn ? weight + (size_t)q * kH * kW : weight + ((size_t)o * nIn + i) * kH * kW;
                    for (int u = 0; u < kH; ++u) {
                        int yy = y - u;
                        if (yy < 0 || yy >= Ho) continue;
                        for (int v = 0; v < kW; ++v) {
                            int xx = x - v;
                            if (xx < 0 || xx >= Wo) continue;
                            s += wk[u * kW + v] * go[((size_t)o * Ho + yy) * Wo + xx];
                        }
                    }
                }
                gi[((size_t)i * H + y) * W + x] = s;
            }
}
void orc_spatial_convolution_acc_grad(const float *in, const float *go, const int *conn, int nConn, int nIn, int nOut, int H,
                                      int W, int kH, int kW, float scale, float *gw, float *gb) {
    int Ho = H - kH + 1, Wo = W - kW + 1;
    int nq = conn ? nConn : nOut * nIn;
    ORC_PAR_FOR
    for (int q = 0; q < nq; ++q) {
        int i = conn ? conn[2 * q] - 1 : q % nIn, o = conn ? conn[2 * q + 1] - 1 : q / nIn;
        for (int u = 0; u < kH; ++u)
            for (int v = 0; v < kW; ++v) {
                float s = 0.f;
                for (int y = 0; y < Ho; ++y)
                    for (int x = 0; x < Wo; ++x) s += go[((size_t)o * Ho + y) * Wo + x] * in[((size_t)i * H + y + u) * W + x + v];
                gw[((size_t)q * kH + u) * kW + v] += scale * s;
            }
    }
    if (gb)
        for (int o = 0; o < nOut; ++o) {
            float s = 0.f;
            for (int p = 0; p < Ho * Wo; ++p) s += go[(size_t)o * Ho * Wo + p];
            gb[o] += scale * s;
        }
}
void orc_tanh_backward(const float *out, const float *go, int64_t n, float *gi) {
    for (int64_t i = 0; i < n; ++i) gi[i] = go[i] * (1.f - out[i] * out[i]);
}
/* nn.LogSoftMax over rows of N (radial/radial_opticalflow_network.lua:50) and its gradient; nn.SoftMax's gradient
 * (the window soft-max of getModel, opticalflow_model.lua:96-109) */
void orc_log_softmax(const float *in, int64_t P, int N, float *out) {
    for (int64_t p = 0; p < P; ++p) {
        const float *r = in + p * N;
        float m = r[0];
        for (int j = 1; j < N; ++j) m = r[j] > m ? r[j] : m;
        double s = 0;
        for (int j = 0; j < N; ++j) s += exp((double)r[j] - m);
        float l = (float)(m + log(s));
        for (int j = 0; j < N; ++j) out[p * N + j] = r[j] - l;
    }
}
void orc_log_softmax_backward(const float *out, const float *go, int64_t P, int N, float *gi) {
    for (int64_t p = 0; p < P; ++p) {
        double s = 0;
        for (int j = 0; j < N; ++j) s += go[p * N + j];
        for (int j = 0; j < N; ++j) gi[p * N + j] = (float)(go[p * N + j] - exp((double)out[p * N + j]) * s);
    }
}
void orc_softmax_backward(const float *out, const float *go, int64_t P, int N, float *gi) {
    for (int64_t p = 0; p < P; ++p) {
        double s = 0;
        for (int j = 0; j < N; ++j) s += (double)go[p * N + j] * out[p * N + j];
        for (int j = 0; j < N; ++j) gi[p * N + j] = (float)(out[p * N + j] * (go[p * N + j] - s));
    }
}

/* N2: gradients of A1 / A1r w.r.t. both feature maps (un-vendored nnx; nothing in the reference tests them -- pinned as
 * the Jacobian of orc_spatial_matching / orc_radial_matching, the way tests/test_cascad.lua:22 pins the cascade).
 * g1[k][y][x]  = sum_{dy,dx}  2 (in1[k][y][x] - in2[k][y+dy][x+dx]) go[y][x][dy][dx]
 * g2[k][v][u]  = sum_{dy,dx} -2 (in1[k][v-dy][u-dx] - in2[k][v][u]) go[v-dy][u-dx][dy][dx]   over the (dy,dx) that keep
 * (v-dy, u-dx) inside in1; float accumulation in (dy, dx) order. */
void orc_spatial_matching_backward(const float *in1, const float *in2, const float *go, int K, int H1, int W1,
                                   int maxh, int maxw, float *g1, float *g2) {
    int H2 = H1 + maxh - 1, W2 = W1 + maxw - 1;
    ORC_PAR_FOR
    for (int k = 0; k < K; ++k) {
        for (int y = 0; y < H1; ++y)
            for (int x = 0; x < W1; ++x) {
                float a = in1[((size_t)k * H1 + y) * W1 + x], s = 0.f;
                for (int dy = 0; dy < maxh; ++dy)
                    for (int dx = 0; dx < maxw; ++dx)
                        s += 2.0f * (a - in2[((size_t)k * H2 + y + dy) * W2 + x + dx]) * go[(((size_t)y * W1 + x) * maxh + dy) * maxw + dx];
                g1[((size_t)k * H1 + y) * W1 + x] = s;
            }
        for (int v = 0; v < H2; ++v)
            for (int u = 0; u < W2; ++u) {
                float b = in2[((size_t)k * H2 + v) * W2 + u], s = 0.f;
                for (int dy = 0; dy < maxh; ++dy)
                    for (int dx = 0; dx < maxw; ++dx) {
                        int y = v - dy, x = u - dx;
                        if (y < 0 || y >= H1 || x < 0 || x >= W1) continue;
                        s += -2.0f * (in1[((size_t)k * H1 + y) * W1 + x] - b) * go[(((size_t)y * W1 + x) * maxh + dy) * maxw + dx];
                    }
                g2[((size_t)k * H2 + v) * W2 + u] = s;
            }
    }
}
void orc_radial_matching_backward(const float *in1, const float *in2, const float *go, int K, int H1, int W, int hWin,
                                  float *g1, float *g2) {
    int H2 = H1 + hWin - 1;
    ORC_PAR_FOR
    for (int k = 0; k < K; ++k) {
        for (int y = 0; y < H1; ++y)
            for (int x = 0; x < W; ++x) {
                float a = in1[((size_t)k * H1 + y) * W + x], s = 0.f;
                for (int d = 0; d < hWin; ++d) s += 2.0f * (a - in2[((size_t)k * H2 + y + d) * W + x]) * go[((size_t)y * W + x) * hWin + d];
                g1[((size_t)k * H1 + y) * W + x] = s;
            }
        for (int v = 0; v < H2; ++v)
            for (int x = 0; x < W; ++x) {
                float b = in2[((size_t)k * H2 + v) * W + x], s = 0.f;
                for (int d = 0; d < hWin; ++d) {
                    int y = v - d;
                    if (y < 0 || y >= H1) continue;
                    s += -2.0f * (in1[((size_t)k * H1 + y) * W + x] - b) * go[((size_t)y * W + x) * hWin + d];
                }
                g2[((size_t)k * H2 + v) * W + x] = s;
            }
    }
}

void orc_ssd_cost_volume(const float *I0, const float *I1, int C, int H, int W,
                         int kh, int kw, int hWin, int wWin, float *out, int row0, int row1) {
    /* ref: radial/radial_opticalflow_groundtruth.lua:79-84.  Feature k=(c,i,j) of frame0 at
     * output (y,x) is I0[c][y+oy+i][x+ox+j] with oy=floor((hWin-1)/2) (the SpatialPadding crop,
     * :81-82); of frame1 at cell (dy,dx) it is I1[c][y+dy+i][x+dx+j]. Same k order as A0. */
    int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    int oy = (hWin - 1) / 2, ox = (wWin - 1) / 2;
    if (row0 < 0) row0 = 0;
    if (row1 > Ho) row1 = Ho;
    ORC_PAR_FOR
    for (int y = row0; y < row1; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int dy = 0; dy < hWin; ++dy)
                for (int dx = 0; dx < wWin; ++dx) {
                    float s = 0.f;
                    for (int c = 0; c < C; ++c)
                        for (int i = 0; i < kh; ++i) {
                            const float *a = I0 + ((size_t)c * H + y + oy + i) * W + x + ox;
                            const float *b = I1 + ((size_t)c * H + y + dy + i) * W + x + dx;
                            for (int j = 0; j < kw; ++j) {
                                float d = a[j] - b[j];
                                s += d * d;
                            }
                        }
                    out[(((size_t)y * Wo + x) * hWin + dy) * wWin + dx] = s;
                }
}

/* ---- A1r --------------------------------------------------------------- */
void orc_radial_matching(const float *in1, const float *in2, int K, int H1, int W, int hWin,
                         float *out) {
    /* ref: radial/radial_opticalflow_network.lua:33,59-72; consumer `min(3)-1`
     * radial/train_radial_opticalflow.lua:166-167 */
    int H2 = H1 + hWin - 1;
    ORC_PAR_FOR
    for (int y = 0; y < H1; ++y)
        for (int x = 0; x < W; ++x)
            for (int d = 0; d < hWin; ++d) {
                float s = 0.f;
                for (int k = 0; k < K; ++k) {
                    float t = in1[((size_t)k * H1 + y) * W + x] - in2[((size_t)k * H2 + y + d) * W + x];
                    s += t * t;
                }
                out[((size_t)y * W + x) * hWin + d] = s;
            }
}

/* ---- A6 ---------------------------------------------------------------- */
void orc_argbest_center(const float *vol, int64_t P, int N, int middle, int take_max,
                        int64_t *idx, float *best) {
    /* ref: radial/radial_opticalflow_groundtruth.lua:88-94: m,idx=output:min(3);
     * flat=m:eq(output[middleidx]); idx=flat*middleidx+(1-flat)*idx.  TH min/max keep the
     * first extremum (strict comparison). opticalflow_model.lua:153-161 is the max twin. */
    for (int64_t p = 0; p < P; ++p) {
        const float *v = vol + p * N;
        float b = v[0];
        int bi = 0;
        for (int n = 1; n < N; ++n) {
            if (take_max ? (v[n] > b) : (v[n] < b)) { b = v[n]; bi = n; }
        }
        int64_t id = bi + 1;
        if (middle > 0 && b == v[middle - 1]) id = middle;
        idx[p] = id;
        if (best) best[p] = b;
    }
}

/* ---- A7 / A8 ----------------------------------------------------------- */
/* ref: version2/extract_output.cpp:17-26 sortswap: swaps value and index iff *b > *a */
static inline void sortswap(float *v, float *ix, int a, int b) {
    if (v[b] > v[a]) {
        float t = v[b]; v[b] = v[a]; v[a] = t;
        t = ix[b]; ix[b] = ix[a]; ix[a] = t;
    }
}
static void sort4(float *v, float *ix) { /* ref: :27-33 */
    sortswap(v, ix, 0, 2); sortswap(v, ix, 1, 3); sortswap(v, ix, 0, 1);
    sortswap(v, ix, 2, 3); sortswap(v, ix, 1, 2);
}
static void sort8(float *v, float *ix) { /* ref: :35-61 */
    sortswap(v, ix, 0, 1); sortswap(v, ix, 2, 3); sortswap(v, ix, 4, 5); sortswap(v, ix, 6, 7);
    sortswap(v, ix, 0, 2); sortswap(v, ix, 1, 3); sortswap(v, ix, 4, 6); sortswap(v, ix, 5, 7);
    sortswap(v, ix, 1, 2); sortswap(v, ix, 5, 6); sortswap(v, ix, 0, 4); sortswap(v, ix, 3, 7);
    sortswap(v, ix, 1, 5); sortswap(v, ix, 2, 6);
    sortswap(v, ix, 1, 4); sortswap(v, ix, 3, 6);
    sortswap(v, ix, 2, 4); sortswap(v, ix, 3, 5);
    sortswap(v, ix, 3, 4);
}
/* shared body of ExtractOutput / ExtractOutputMarginalized: returns 0 if the pixel is skipped */
static int extract_pixel(const float *v, int N, double threshold, int M, int64_t *imax, double *acc_out) {
    float hv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* ref: :86-90 zeroed highs */
    int n = 0;
    for (int k = 0; k < N; ++k) { /* ref: :99-112 */
        if (v[k] > threshold) {   /* float promoted to double vs double threshold, as in the reference */
            hv[n] = v[k];
            hi[n] = (float)(k + 1); /* index is stored in a float temp, ref :105 */
            if (++n == M) break;
        }
    }
    if (!(hv[0] > 0)) return 0; /* ref: :121 */
    if (M == 4) sort4(hv, hi); else sort8(hv, hi);
    *imax = (int64_t)hi[0];     /* ref: :123 */
    for (int k = 1; k < M; ++k) hv[k] += hv[k - 1]; /* ref: :124-125 float prefix sums */
    double acc = 0;             /* ref: :126-128 accreal=double */
    for (int k = 0; k < M; ++k) acc += hv[k];
    *acc_out = acc;
    return 1;
}
void orc_extract_output(const float *input, int64_t P, int N, double threshold, int64_t *imaxs,
                        float *scores) {
    int M = threshold < 0.2 ? 8 : 4; /* ref: :83-85 */
    for (int64_t p = 0; p < P; ++p) {
        int64_t im; double acc;
        if (extract_pixel(input + p * N, N, threshold, M, &im, &acc)) {
            imaxs[p] = im;
            scores[p] = (float)acc;
        }
    }
}
void orc_extract_output_marginalized(const float *input, int64_t P, int N, double threshold,
                                     double threshold_acc, int64_t *ret, int64_t *retgd) {
    int M = threshold < 0.2 ? 8 : 4;
    for (int64_t p = 0; p < P; ++p) retgd[p] = 0; /* ref: :166 THLongTensor_zero(retgd) */
    for (int64_t p = 0; p < P; ++p) {
        int64_t im; double acc;
        if (extract_pixel(input + p * N, N, threshold, M, &im, &acc)) {
            ret[p] = im;
            if (acc >= threshold_acc) retgd[p] = 1; /* ref: :227-228 */
        }
    }
}

/* ---- A9 ---------------------------------------------------------------- */
void orc_x2yx(const int64_t *idx, int64_t P, int maxh, int maxw, int64_t *y, int64_t *x) {
    /* ref: radial/radial_opticalflow_groundtruth.lua:97-100 */
    for (int64_t p = 0; p < P; ++p) {
        int64_t fl = (idx[p] - 1) / maxw; /* idx>=1 so trunc == floor */
        y[p] = fl - (maxh - 1) / 2;
        x[p] = idx[p] - 1 - fl * maxw - (maxw - 1) / 2;
    }
}

/* ---- A10 --------------------------------------------------------------- */
static int ring_d(int maxw, const int *ratios, int i) { /* i: 0-based scale index >=1 */
    /* ref: opticalflow_model_multiscale.lua:94-95 / :296 round(maxw*(r_i-r_{i-1})/(2 r_i)) */
    return (int)lua_round((double)maxw * (ratios[i] - ratios[i - 1]) / (2.0 * ratios[i]));
}
int64_t orc_multi_nclasses(int maxh, int maxw, const int *ratios, int nratios) {
    int64_t n = (int64_t)maxh * maxw;
    for (int i = 1; i < nratios; ++i) {
        int d = ring_d(maxw, ratios, i);
        n += 2 * d * maxw + 2 * (maxh - 2 * d) * d;
    }
    return n;
}
static int is_in(double size, double v) { /* ref: :13-15 */
    return (v >= -ceil(size / 2) + 1) && (v <= floor(size / 2));
}
int64_t orc_yx2x_multi(int maxh, int maxw, const int *ratios, int nratios, double y, double x) {
    /* ref: opticalflow_model_multiscale.lua:10-52 */
    x = lua_round(x);
    y = lua_round(y);
    int i = 0;
    double tx = 0, ty = 0;
    for (; i < nratios; ++i) {
        if (is_in((double)maxw * ratios[i], x) && is_in((double)maxh * ratios[i], y)) {
            tx = ceil(x / ratios[i]) + ceil(maxw / 2.0);
            ty = ceil(y / ratios[i]) + ceil(maxh / 2.0);
            break;
        }
    }
    if (i >= nratios) return -1; /* ref :29 assert */
    int64_t targetx = (int64_t)tx, targety = (int64_t)ty, it;
    if (i == 0) return (targety - 1) * maxw + targetx;
    int d = (int)floor((double)maxw * (ratios[i] - ratios[i - 1]) / (2.0 * ratios[i]) + 0.5);
    if (targety <= d) it = (targety - 1) * maxw + targetx;
    else if (targety > maxh - d)
        it = d * maxw + 2 * (maxh - 2 * d) * d + (targety - (maxh - d) - 1) * maxw + targetx;
    else if (targetx <= d) it = d * maxw + (targety - d - 1) * d + targetx;
    else if (targetx > maxw - d)
        it = d * maxw + (maxh - 2 * d) * d + (targety - d - 1) * d + targetx - (maxw - d);
    else return -1; /* ref :45-46 assert(false) */
    /* ref :48-49: NOTE uses the ring length of scale i for every skipped scale */
    return (int64_t)maxw * maxh + (int64_t)(i - 1) * (2 * d * maxw + 2 * (maxh - 2 * d) * d) + it;
}
int orc_x2yx_multi_number(int maxh, int maxw, const int *ratios, int nratios, int64_t id,
                          int64_t *oy, int64_t *ox) {
    /* ref: opticalflow_model_multiscale.lua:83-132 */
    int chh = iceil_div2(maxh), chw = iceil_div2(maxw);
    int64_t x = id;
    if (x < 1) return -1;
    if (x <= (int64_t)maxh * maxw) {
        int64_t ty = (x - 1) / maxw + 1, tx = (x - 1) % maxw + 1;
        *oy = ty - chh; *ox = tx - chw;
        return 0;
    }
    x -= (int64_t)maxh * maxw;
    for (int i = 1; i < nratios; ++i) {
        int d = ring_d(maxw, ratios, i);
        int64_t len = 2 * d * maxw + 2 * (maxh - 2 * d) * d, ty, tx;
        if (x <= len) {
            if (x <= (int64_t)d * maxw) { /* top block d x maxw */
                ty = (x - 1) / maxw + 1; tx = (x - 1) % maxw + 1;
            } else {
                x -= (int64_t)d * maxw;
                if (x <= (int64_t)(maxh - 2 * d) * d) { /* left block */
                    ty = (x - 1) / d + 1 + d; tx = (x - 1) % d + 1;
                } else {
                    x -= (int64_t)(maxh - 2 * d) * d;
                    if (x <= (int64_t)(maxh - 2 * d) * d) { /* right block */
                        ty = (x - 1) / d + 1 + d; tx = (x - 1) % d + 1 + maxw - d;
                    } else {
                        x -= (int64_t)(maxh - 2 * d) * d;
                        if (x > (int64_t)d * maxw) return -1;
                        ty = (x - 1) / maxw + 1 + maxh - d; tx = (x - 1) % maxw + 1; /* bottom */
                    }
                }
            }
            *oy = (ty - chh) * ratios[i]; *ox = (tx - chw) * ratios[i];
            return 0;
        }
        x -= len;
    }
    return -1; /* ref :131 assert(false) */
}
int orc_x2yx_multi(int maxh, int maxw, const int *ratios, int nratios, const int64_t *idx,
                   int64_t P, int64_t *y, int64_t *x) {
    int rc = 0;
    for (int64_t p = 0; p < P; ++p)
        if (orc_x2yx_multi_number(maxh, maxw, ratios, nratios, idx[p], &y[p], &x[p])) rc = -1;
    return rc;
}
void orc_x2yx_multi_compat_c(int maxh, int maxw, const int *ratios_lua, int nratios,
                             const int64_t *idx, int64_t P, int64_t *rety, int64_t *retx) {
    /* ref: x2yxMulti2.c:1-95, bug for bug.  (1) :15-19 reads Lua keys 0..n-1 of a 1-based
     * table => ratios[0]=0 (nil->0), ratios[k]=lua[k]; (2) :41 lengths drop the factor d;
     * (3) strict '<' at :50,:61,:79; ceil(maxh/2) on ints (:24-25) is maxh/2;
     * (4) unmatched ids write nothing. C integer / and % on x-1=-1 kept as is. */
    int ratios[ORC_MAX_RATIOS];
    for (int i = 0; i < nratios && i < ORC_MAX_RATIOS; ++i) ratios[i] = (i == 0) ? 0 : ratios_lua[i - 1];
    int chmaxh = maxh / 2, chmaxw = maxw / 2;
    int patcharea = maxh * maxw;
    int borders[ORC_MAX_RATIOS], lengths[ORC_MAX_RATIOS];
    for (int i = 1; i < nratios; ++i) {
        borders[i] = (int)roundf((float)maxw * ((float)ratios[i] - (float)ratios[i - 1]) / (2.0f * (float)ratios[i]));
        lengths[i] = 2 * maxw + 2 * (maxh - 2 * borders[i]) * borders[i];
    }
    for (int64_t p = 0; p < P; ++p) {
        long x = (long)idx[p];
        if (x < patcharea) {
            rety[p] = (long)floor((double)((x - 1) / maxw)) + 1 - chmaxh;
            retx[p] = (x - 1) % maxw + 1 - chmaxw;
        } else {
            x -= patcharea;
            for (int k = 1; k < nratios; ++k) {
                int d = borders[k];
                int mH = (maxh - 2 * d) * d;
                if (x <= lengths[k]) {
                    if (x < d * maxw) {
                        rety[p] = ((x - 1) / maxw + 1 - chmaxh) * ratios[k];
                        retx[p] = ((x - 1) % maxw + 1 - chmaxw) * ratios[k];
                        break;
                    }
                    x -= d * maxw;
                    if (x <= mH) {
                        rety[p] = (d ? ((x - 1) / d + 1 + d - chmaxh) : 0) * ratios[k];
                        retx[p] = (d ? ((x - 1) % d + 1 - chmaxw) : 0) * ratios[k];
                        break;
                    }
                    x -= mH;
                    if (x <= mH) {
                        rety[p] = (d ? ((x - 1) / d + 1 + d - chmaxh) : 0) * ratios[k];
                        retx[p] = (d ? ((x - 1) % d + 1 + maxw - d - chmaxw) : 0) * ratios[k];
                        break;
                    }
                    x -= mH;
                    if (x < d * maxw) {
                        rety[p] = ((x - 1) / maxw + 1 + maxh - d - chmaxh) * ratios[k];
                        retx[p] = ((x - 1) % maxw + 1 - chmaxw) * ratios[k];
                        break;
                    }
                } else {
                    x -= lengths[k];
                }
            }
        }
    }
}

/* ---- A2 pieces --------------------------------------------------------- */
void orc_downsample_box(const float *img, int C, int H, int W, int r, float *out) {
    /* nn.SpatialDownSampling(r,r): mean of each r x r block. ref: opticalflow_model_multiscale.lua:146.
     * [3P nnx: accumulation order row-major inside the block, scaled by 1/(r*r) -- parity unpinned] */
    int Ho = H / r, Wo = W / r;
    float inv = 1.0f / (float)(r * r);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float s = 0.f;
                for (int i = 0; i < r; ++i)
                    for (int j = 0; j < r; ++j) s += img[((size_t)c * H + y * r + i) * W + x * r + j];
                out[((size_t)c * Ho + y) * Wo + x] = s * inv;
            }
}
void orc_zero_pad(const float *img, int C, int H, int W, int pl, int pr, int pt, int pb, float *out) {
    /* nn.SpatialZeroPadding(l,r,t,b), non-negative pads. ref: :136-141,147 */
    int Ho = H + pt + pb, Wo = W + pl + pr;
    memset(out, 0, sizeof(float) * (size_t)C * Ho * Wo);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            memcpy(out + ((size_t)c * Ho + y + pt) * Wo + pl, img + ((size_t)c * H + y) * W, sizeof(float) * W);
}
void orc_pyramid_scale_volume(const float *I0, const float *I1, int C, int H, int W, int r,
                              int kh, int kw, int maxh, int maxw, float *out) {
    /* ref: opticalflow_model_multiscale.lua:134-173 (downsample, pad hPatch2-1 = maxh-1+kh-1 split
     * floor/ceil), :196-216 (frame0 cropped by maxw-1 split floor/ceil, identity kxk patch
     * features, SpatialMatching(maxh,maxw)).  Result at native scale [H/r][W/r][maxh][maxw]. */
    int Hs = H / r, Ws = W / r;
    int hp = maxh - 1 + kh - 1, wp = maxw - 1 + kw - 1;
    int pt = hp / 2, pb = hp - pt, pl = wp / 2, pr = wp - pl;
    int Hp = Hs + hp, Wp = Ws + wp;
    float *d0 = (float *)malloc(sizeof(float) * (size_t)C * Hs * Ws);
    float *d1 = (float *)malloc(sizeof(float) * (size_t)C * Hs * Ws);
    float *p0 = (float *)malloc(sizeof(float) * (size_t)C * Hp * Wp);
    float *p1 = (float *)malloc(sizeof(float) * (size_t)C * Hp * Wp);
    orc_downsample_box(I0, C, H, W, r, d0);
    orc_downsample_box(I1, C, H, W, r, d1);
    orc_zero_pad(d0, C, Hs, Ws, pl, pr, pt, pb, p0);
    orc_zero_pad(d1, C, Hs, Ws, pl, pr, pt, pb, p1);
    /* frame0 crop floor((maxh-1)/2) top / ceil bottom == the oy offset of orc_ssd_cost_volume */
    orc_ssd_cost_volume(p0, p1, C, Hp, Wp, kh, kw, maxh, maxw, out, 0, Hs);
    free(d0); free(d1); free(p0); free(p1);
}

/* ---- A3 ---------------------------------------------------------------- */
void orc_softmin(const float *cost, int64_t P, int N, float *prob) {
    /* nn.Minus -> nn.SoftMax over the N window cells. ref: opticalflow_model_multiscale.lua:270-279.
     * [3P nn.SoftMax: max-subtracted, exact exp, float sum in index order -- parity unpinned] */
    for (int64_t p = 0; p < P; ++p) {
        const float *c = cost + p * N;
        float *o = prob + p * N;
        float m = -c[0];
        for (int n = 1; n < N; ++n) if (-c[n] > m) m = -c[n];
        float s = 0.f;
        for (int n = 0; n < N; ++n) { float e = expf(-c[n] - m); o[n] = e; s += e; }
        float inv = 1.0f / s;
        for (int n = 0; n < N; ++n) o[n] *= inv;
    }
}

/* ---- A4 / A5 ----------------------------------------------------------- */
static int cascade_check(int nratios, const int *ratios, int maxh, int maxw) {
    for (int i = 0; i + 1 < nratios; ++i) { /* ref: CascadingAddTable.lua:121-124 */
        int r = ratios[i], r2 = ratios[i + 1];
        if ((maxh * (r2 - r)) % (2 * r2) != 0 || (maxw * (r2 - r)) % (2 * r2) != 0) return -1;
        if (r2 % r != 0) return -1;
    }
    return 0;
}
/* out_i = in_i + replicate_{r2/r}( crop(out_{i+1}, dh, dw) )   ref: CascadingAddTable.lua:117-132 */
static void cascade_window(float *const *win, int nratios, const int *ratios, int maxh, int maxw) {
    for (int i = nratios - 2; i >= 0; --i) {
        int r = ratios[i], r2 = ratios[i + 1], q = r2 / r;
        int dh = maxh * (r2 - r) / (2 * r2), dw = maxw * (r2 - r) / (2 * r2);
        for (int a = 0; a < maxh; ++a)
            for (int b = 0; b < maxw; ++b)
                win[i][a * maxw + b] += win[i + 1][(dh + a / q) * maxw + dw + b / q];
    }
}
int orc_cascading_add(const float *const *in, int nratios, const int *ratios, int64_t P, int maxh,
                      int maxw, float *const *out) {
    if (cascade_check(nratios, ratios, maxh, maxw)) return -1;
    int N = maxh * maxw;
    float *w[ORC_MAX_RATIOS];
    for (int64_t p = 0; p < P; ++p) {
        for (int s = 0; s < nratios; ++s) {
            w[s] = out[s] + p * N;
            memcpy(w[s], in[s] + p * N, sizeof(float) * N);
        }
        cascade_window(w, nratios, ratios, maxh, maxw);
    }
    return 0;
}
/* A4b: gradient of the cascade w.r.t. its inputs.  ref: CascadingAddTable.lua:137-154 (updateGradInput) with HEAD's
 * transformer graph (:26-60: reshape | crop -> replicate -> reshape, CAddTable; Mul2/Power commented out, so there are
 * no parameter gradients).  g_0 = go_0; g_{i+1} = go_{i+1} + pad(blocksum_q(g_i)); gradInput_i = g_i.
 * Pinned the way the reference pins it (tests/test_cascad.lua:22: nn.Jacobian.testJacobian): adjoint of the forward. */
int orc_cascading_add_backward(const float *const *gradOut, int nratios, const int *ratios, int64_t P, int maxh,
                               int maxw, float *const *gradIn) {
    if (cascade_check(nratios, ratios, maxh, maxw)) return -1;
    int N = maxh * maxw;
    for (int64_t p = 0; p < P; ++p) {
        memcpy(gradIn[0] + p * N, gradOut[0] + p * N, sizeof(float) * N);
        for (int i = 0; i + 1 < nratios; ++i) {
            int r = ratios[i], r2 = ratios[i + 1], q = r2 / r;
            int dh = maxh * (r2 - r) / (2 * r2), dw = maxw * (r2 - r) / (2 * r2);
            const float *g = gradIn[i] + p * N;
            float *gn = gradIn[i + 1] + p * N;
            memcpy(gn, gradOut[i + 1] + p * N, sizeof(float) * N);
            for (int a = 0; a < maxh / q; ++a)
                for (int b = 0; b < maxw / q; ++b) {
                    float acc = 0.f; /* row-major over the q x q block, float accumulation */
                    for (int u = 0; u < q; ++u)
                        for (int v = 0; v < q; ++v) acc += g[(a * q + u) * maxw + b * q + v];
                    gn[(dh + a) * maxw + dw + b] += acc;
                }
        }
    }
    return 0;
}
int orc_cascade_ring(const float *const *prob, int nratios, const int *ratios, int H, int W,
                     int maxh, int maxw, float *out) {
    if (cascade_check(nratios, ratios, maxh, maxw) || nratios > ORC_MAX_RATIOS) return -1;
    int N = maxh * maxw;
    int64_t ncls = orc_multi_nclasses(maxh, maxw, ratios, nratios);
    float *buf = (float *)malloc(sizeof(float) * (size_t)N * nratios);
    float *w[ORC_MAX_RATIOS];
    for (int s = 0; s < nratios; ++s) w[s] = buf + (size_t)s * N;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            for (int s = 0; s < nratios; ++s) { /* nearest-neighbour upsample x r of dims (1,2) */
                int r = ratios[s], Ws = W / r;
                memcpy(w[s], prob[s] + ((size_t)(y / r) * Ws + x / r) * N, sizeof(float) * N);
            }
            cascade_window(w, nratios, ratios, maxh, maxw);
            float *o = out + ((size_t)y * W + x) * ncls;
            memcpy(o, w[0], sizeof(float) * N); /* scale 1: all cells. ref: :294 */
            o += N;
            for (int s = 1; s < nratios; ++s) { /* ring blocks top,left,right,bottom. ref: :295-324 */
                int d = ring_d(maxw, ratios, s);
                for (int a = 0; a < d; ++a) for (int b = 0; b < maxw; ++b) *o++ = w[s][a * maxw + b];
                for (int a = d; a < maxh - d; ++a) for (int b = 0; b < d; ++b) *o++ = w[s][a * maxw + b];
                for (int a = d; a < maxh - d; ++a) for (int b = maxw - d; b < maxw; ++b) *o++ = w[s][a * maxw + b];
                for (int a = maxh - d; a < maxh; ++a) for (int b = 0; b < maxw; ++b) *o++ = w[s][a * maxw + b];
            }
        }
    free(buf);
    return 0;
}

/* ---- A11 --------------------------------------------------------------- */
void orc_paste_center(const float *src, int h, int w, float *dst, int H, int W) {
    /* ref: opticalflow_model.lua:227-249: zero, offset floor((H-h)/2) */
    int ho = (H - h) / 2, wo = (W - w) / 2;
    memset(dst, 0, sizeof(float) * (size_t)H * W);
    for (int y = 0; y < h; ++y) memcpy(dst + (size_t)(y + ho) * W + wo, src + (size_t)y * w, sizeof(float) * w);
}

/* ---- A12 --------------------------------------------------------------- */
void orc_flow_to_depth_cartesian(const float *flow, int H, int W, float mw, float mh, int fix_dot,
                                 float *depth, float *conf) {
    /* ref: test_opticalflow.lua:143-189.  Arguments (cx,cy)=(mw,mh). float variables, double
     * sqrt (C promotion) rounded back to float on assignment, exactly as the inline C. */
    float infty = (float)((double)W / 2); /* ref :148 geometry.wImg/2 passed as lua number -> float */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float py = (float)i - mh, px = (float)j - mw;
            float pn = (float)sqrt((double)(px * px + py * py));
            float dy = flow[(size_t)i * W + j], dx = flow[(size_t)H * W + (size_t)i * W + j];
            float dn = (float)sqrt((double)(dx * dx + dy * dy));
            float r = 0.f, c = 0.f; /* ref :146-147 zero-initialised outputs */
            if (dn >= 0.2f) {
                float q = pn / dn;
                r = q < infty ? q : infty;
                float dot = fix_dot ? (px * dx + py * dy) : (px * dx + dy * dy); /* ref :181 (sic) */
                if (dot > 0.125f) c = 1.0f;
            } else {
                c = 1.0f;
                r = infty;
            }
            depth[(size_t)i * W + j] = r;
            conf[(size_t)i * W + j] = c;
        }
}
void orc_flow_to_depth_radial(const float *rflow, const float *unused, int H, int W, float xc,
                              float yc, float infty, float *depth, float *conf) {
    /* ref: radial/radial_opticalflow_display.lua:6-58; returns ret/infty and confs */
    (void)unused;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float a = (float)j - xc, b = (float)i - yc;
            float d = (float)sqrt((double)(a * a + b * b));
            float o = 0.f, c = 1.f;
            if (d > 10.0f) {
                float f = rflow[(size_t)i * W + j];
                o = (f < 0.1f) ? infty : d / f;
            } else c = 0.f;
            depth[(size_t)i * W + j] = o / infty;
            conf[(size_t)i * W + j] = c;
        }
}

/* A11 'mean' extraction helper: input:reshape(H,W,maxh,maxw):sum(4). ref: opticalflow_model.lua:192.
 * [3P-recall: TH sums float tensors in a double accumulator] in [P][A][B] -> out [P][A] */
void orc_marginal_sum(const float *in, int64_t P, int A, int B, float *out) {
    for (int64_t p = 0; p < P * A; ++p) {
        double acc = 0;
        for (int b = 0; b < B; ++b) acc += in[p * B + b];
        out[p] = (float)acc;
    }
}

/* A12(iii): ARdroneAPI::computeDepthMapFromFlow. ref: ardrone/ardrone_api.cpp:99-140.
 * Mode filter of the rounded x-flow over the window [i-3, i+3) x [j-3, j+3) (sic: half-open, 6x6) of pixels with a
 * non-zero mask, 20 bins for values -8..11 (the reference indexes values[f+8] unchecked: samples outside that range
 * are undefined behaviour there and are skipped here), first maximum wins; depth = m*|j-W/2|/|mode| (100 where
 * |mode| < 1.1) and conf = 1 where mask > 0.5 and j != W/2, else conf = 0 and depth 0 (left uninitialised there). */
void orc_flow_to_depth_ardrone(const float *xflow, const float *mask, int H, int W, float m, float *depth, float *conf) {
    const int k = 3, middlex = W / 2;
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < W; ++i) {
            float mode = 0.f;
            if (mask[(size_t)j * W + i] != 0.f) {
                int values[20] = {0};
                int i2a = i - k > 0 ? i - k : 0, i2b = i + k < W ? i + k : W;
                int j2a = j - k > 0 ? j - k : 0, j2b = j + k < H ? j + k : H;
                for (int i2 = i2a; i2 < i2b; ++i2)
                    for (int j2 = j2a; j2 < j2b; ++j2)
                        if (mask[(size_t)j2 * W + i2] != 0.f) {
                            int f = (int)roundf(xflow[(size_t)j2 * W + i2]);
                            if (f >= -8 && f < 12) ++values[f + 8];
                        }
                int best = 0, im = 0;
                for (int iv = 0; iv < 20; ++iv)
                    if (values[iv] > best) { best = values[iv]; im = iv - 8; }
                mode = (float)im;
            }
            float d = 0.f, c = 0.f;
            if (mask[(size_t)j * W + i] > 0.5f && i - middlex != 0) {
                d = fabsf(mode) < 1.1f ? 100.0f : m * (float)abs(i - middlex) / fabsf(mode);
                c = 1.0f;
            }
            depth[(size_t)j * W + i] = d;
            conf[(size_t)j * W + i] = c;
        }
}

/* ---- A13 / A14 --------------------------------------------------------- */
void orc_polar_grid_c2p(int wsrc, int hsrc, int wdst, int hdst, float xc, float yc, int lpad,
                        int rpad, float rmax, float alpha, float *mask) {
    /* ref: radial/cartesian2polar.lua:4-49 */
    (void)wsrc; (void)hsrc;
    int Wp = wdst + lpad + rpad;
    float kr = (float)((double)rmax / pow((double)hdst, (double)alpha));
    float ktheta = (float)(2 * M_PI / wdst);
    float *m0 = mask, *m1 = mask + (size_t)hdst * Wp;
    for (int i = 0; i < hdst; ++i)
        for (int j = 0; j < wdst; ++j) {
            float r = (float)(kr * pow((double)(float)i, (double)alpha));
            float th = ktheta * (float)j;
            m0[(size_t)i * Wp + lpad + j] = (float)(r * sin((double)th) + yc);
            m1[(size_t)i * Wp + lpad + j] = (float)(r * cos((double)th) + xc);
        }
    for (int pl = 0; pl < 2; ++pl) { /* circular column padding :42-47 */
        float *m = pl ? m1 : m0;
        for (int i = 0; i < hdst; ++i) {
            for (int j = 0; j < lpad; ++j) m[(size_t)i * Wp + j] = m[(size_t)i * Wp + lpad + wdst - lpad + j];
            for (int j = 0; j < rpad; ++j) m[(size_t)i * Wp + lpad + wdst + j] = m[(size_t)i * Wp + lpad + j];
        }
    }
}
void orc_polar_grid_p2c(int wsrc, int hsrc, int wdst, int hdst, float xc, float yc, float rmax,
                        float alpha, float *mask) {
    /* ref: radial/cartesian2polar.lua:51-89 */
    float pi2 = (float)(2 * M_PI);
    float kx = (float)((double)wsrc / (2 * M_PI));
    float ky = (float)((double)hsrc / pow((double)rmax, 1.0 / (double)alpha));
    float invalpha = (float)(1.0 / (double)alpha) * 0.5f;
    float *m0 = mask, *m1 = mask + (size_t)hdst * wdst;
    for (int i = 0; i < hdst; ++i)
        for (int j = 0; j < wdst; ++j) {
            float x = (float)j - xc, y = (float)i - yc;
            m0[(size_t)i * wdst + j] = (float)(pow((double)(x * x + y * y), (double)invalpha) * ky);
            m1[(size_t)i * wdst + j] = (float)(fmod(atan2((double)y, (double)x) + pi2, (double)pi2) * kx);
        }
}
void orc_warp_bilinear(const float *img, int C, int H, int W, const float *mask, int Hd, int Wd,
                       float *out) {
    /* image.warp(img, mask, 'bilinear', false): absolute coords, plane0=y plane1=x, 0-based.
     * ref: radial/cartesian2polar.lua:91-93.  [3P image: border policy = clamp -- parity unpinned] */
    const float *my = mask, *mx = mask + (size_t)Hd * Wd;
    for (int i = 0; i < Hd; ++i)
        for (int j = 0; j < Wd; ++j) {
            float fy = my[(size_t)i * Wd + j], fx = mx[(size_t)i * Wd + j];
            fy = fy < 0 ? 0 : (fy > (float)(H - 1) ? (float)(H - 1) : fy);
            fx = fx < 0 ? 0 : (fx > (float)(W - 1) ? (float)(W - 1) : fx);
            int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
            float wy = fy - (float)y0, wx = fx - (float)x0;
            for (int c = 0; c < C; ++c) {
                const float *p = img + (size_t)c * H * W;
                float v = (1 - wy) * ((1 - wx) * p[(size_t)y0 * W + x0] + wx * p[(size_t)y0 * W + x1]) +
                          wy * ((1 - wx) * p[(size_t)y1 * W + x0] + wx * p[(size_t)y1 * W + x1]);
                out[((size_t)c * Hd + i) * Wd + j] = v;
            }
        }
}

/* ---- A16 --------------------------------------------------------------- */
static int cmp_float(const void *a_, const void *b_) { /* ref: opticalflow_model.lua:328-340 */
    float a = *(const float *)a_, b = *(const float *)b_;
    return a == b ? 0 : (a < b ? -1 : 1);
}
int orc_postprocess_image(const float *flow, const float *mask, int H, int W, int k, int method, float *out) {
    const size_t HW = (size_t)H * W;
    const int halfk = k / 2;
    memset(out, 0, sizeof(float) * 2 * HW);
    if (method == 0) {
        /* inputR = floor(input+0.5); m = inputR:min(); fmax(inputR-m, ...); output = output + m  (:436-440) */
        float *R = (float *)malloc(sizeof(float) * 2 * HW);
        float m = 0.f, mx = 0.f;
        for (size_t i = 0; i < 2 * HW; ++i) {
            R[i] = floorf(flow[i] + 0.5f);
            if (i == 0 || R[i] < m) m = R[i];
            if (i == 0 || R[i] > mx) mx = R[i];
        }
        if (mx - m > 15.f) { free(R); return -1; }   /* the reference's tmp[256] / ROWSIZE=16 would overflow */
        for (int i = 0; i < H - k; ++i)
            for (int j = 0; j < W - k; ++j) {
                int tmp[256];
                memset(tmp, 0, sizeof(tmp));
                for (int ik = i; ik < i + k; ++ik)
                    for (int jk = j; jk < j + k; ++jk)
                        if (mask[(size_t)ik * W + jk]) {
                            int vx = (int)(R[HW + (size_t)ik * W + jk] - m);
                            int vy = (int)(R[(size_t)ik * W + jk] - m);
                            ++tmp[vx + 16 * vy];
                        }
                int im = 0;
                for (int l = 0; l < 256; ++l)
                    if (tmp[l] > tmp[im]) im = l;
                out[HW + (size_t)(i + halfk) * W + j + halfk] = (float)(im % 16);
                out[(size_t)(i + halfk) * W + j + halfk] = (float)(im / 16);
            }
        for (size_t i = 0; i < 2 * HW; ++i) out[i] += m;
        free(R);
        return 0;
    }
    if (k * k > 32) return -1;   /* the reference's float tmp[32] would overflow */
    for (int i = 0; i < H - k; ++i)
        for (int j = 0; j < W - k; ++j) {
            float tmp[32], tmp2[32];
            memset(tmp, 0, sizeof(tmp));
            memset(tmp2, 0, sizeof(tmp2));
            int n = 0;
            for (int ik = i; ik < i + k; ++ik)
                for (int jk = j; jk < j + k; ++jk)
                    if (mask[(size_t)ik * W + jk]) {
                        tmp[n] = flow[(size_t)ik * W + jk];
                        tmp2[n++] = flow[HW + (size_t)ik * W + jk];
                    }
            qsort(tmp, n, sizeof(float), cmp_float);
            qsort(tmp2, n, sizeof(float), cmp_float);
            out[HW + (size_t)(i + halfk) * W + j + halfk] = tmp2[n / 2];
            out[(size_t)(i + halfk) * W + j + halfk] = tmp[n / 2];
        }
    return 0;
}

/* ---- A17 --------------------------------------------------------------- */
void orc_enlarge_mask(float *mask, int H, int W, int ix, int iy) {
    /* ref: depth_estimation_api.lua:93-126, literally */
    for (int i = 0; i < H; ++i) {
        for (int j = 0; j < W; ++j)
            if (mask[(size_t)i * W + j] > 0.5) {
                for (int k = j; k < (j + ix < W ? j + ix : W); ++k) mask[(size_t)i * W + k] = 0.0f;
                break;
            }
        for (int j = W - 1; j >= 0; --j)
            if (mask[(size_t)i * W + j] > 0.5) {
                for (int k = j; k >= (j - ix + 1 > 0 ? j - ix + 1 : 0); --k) mask[(size_t)i * W + k] = 0.0f;
                break;
            }
    }
    for (int j = 0; j < W; ++j) {
        for (int i = 0; i < H; ++i)
            if (mask[(size_t)i * W + j] > 0.5) {
                for (int k = i; k < (i + iy < H ? i + iy : H); ++k) mask[(size_t)k * W + j] = 0.0f;
                break;
            }
        for (int i = H - 1; i >= 0; --i)
            if (mask[(size_t)i * W + j] > 0.5) {
                for (int k = i; k >= (i - iy + 1 > 0 ? i - iy + 1 : 0); --k) mask[(size_t)k * W + j] = 0.0f;
                break;
            }
    }
}

/* ---- A18 --------------------------------------------------------------- */
void orc_output_extractor(const float *input, int64_t P, int maxh, int maxw, float *x, float *y) {
    /* ref: OutputExtractor.lua:7-35: xmul[k]=j, ymul[k]=i for k=(i-1)*maxw+j; x = sum(input.*xmul) over dim 3 */
    int N = maxh * maxw;
    for (int64_t p = 0; p < P; ++p) {
        float sx = 0.f, sy = 0.f;
        for (int k = 0; k < N; ++k) {
            sx += input[p * N + k] * (float)(k % maxw + 1);
            sy += input[p * N + k] * (float)(k / maxw + 1);
        }
        x[p] = sx; y[p] = sy;
    }
}

/* ======================================================================================================================
 * next-row N4: ego-motion rectification and relative pose.  The reference takes all of it from the un-vendored,
 * OpenCV-backed `sfm2` package -- nothing of sfm2 is in the repository -- so these follow the CALL SITES (what goes in,
 * what the results are used for); "parity unpinned (3P)".
 * ====================================================================================================================== */
static float orc_bilin_clamped(const float *p, int H, int W, float fy, float fx) {
    fy = fy < 0 ? 0 : (fy > (float)(H - 1) ? (float)(H - 1) : fy);
    fx = fx < 0 ? 0 : (fx > (float)(W - 1) ? (float)(W - 1) : fx);
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    float wy = fy - (float)y0, wx = fx - (float)x0;
    float top = (1 - wx) * p[(size_t)y0 * W + x0] + wx * p[(size_t)y0 * W + x1];
    float bot = (1 - wx) * p[(size_t)y1 * W + x0] + wx * p[(size_t)y1 * W + x1];
    return (1 - wy) * top + wy * bot;
}

static int orc_inv3(const double *m, double *o) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (fabs(d) < 1e-300) return -1;
    o[0] = (m[4] * m[8] - m[5] * m[7]) / d; o[1] = (m[2] * m[7] - m[1] * m[8]) / d; o[2] = (m[1] * m[5] - m[2] * m[4]) / d;
    o[3] = (m[5] * m[6] - m[3] * m[8]) / d; o[4] = (m[0] * m[8] - m[2] * m[6]) / d; o[5] = (m[2] * m[3] - m[0] * m[5]) / d;
    o[6] = (m[3] * m[7] - m[4] * m[6]) / d; o[7] = (m[1] * m[6] - m[0] * m[7]) / d; o[8] = (m[0] * m[4] - m[1] * m[3]) / d;
    return 0;
}
static void orc_mul3(const double *a, const double *b, double *o) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

/* ref: `e2 = calibrationp.K * T; e2 = e2 / e2[3]; e2 = e2*networkp.wImg/calibrationp.wImg`
 *      radial/radial_opticalflow_data.lua:218-220 (= radial/test_radial_opticalflow.lua:128-130,170) */
int orc_epipole(const double *K9, const double *T3, double scale, double *e2) {
    double x = K9[0] * T3[0] + K9[1] * T3[1] + K9[2] * T3[2], y = K9[3] * T3[0] + K9[4] * T3[1] + K9[5] * T3[2],
           z = K9[6] * T3[0] + K9[7] * T3[1] + K9[8] * T3[2];
    if (z == 0) return -1;
    e2[0] = x / z * scale;
    e2[1] = y / z * scale;
    return 0;
}

/* ref: `prev_img, prev_img_mask = sfm2.removeEgoMotion(prev_img, Ksmall, R, 'bilinear')` radial/radial_opticalflow_data.lua:231,
 *      depth_estimation_api.lua:147, test_opticalflow.lua:284.  A rotation between two views of a static scene moves pixels by the
 *      homography K R K^-1; out(p) = bilinear(img, K R K^-1 p) (inverse: R^T), mask = 1 where the source is inside the frame (the
 *      callers erode and polar-warp the mask, data.lua:233-240). */
int orc_remove_ego_motion(const float *img, int C, int H, int W, const double *K9, const double *R9, int inverse, float *out, float *mask) {
    double Ki[9], Rt[9], t[9], Hd[9];
    if (orc_inv3(K9, Ki)) return -1;
    const double *Ru = R9;
    if (inverse) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[i * 3 + j] = R9[j * 3 + i];
        Ru = Rt;
    }
    orc_mul3(K9, Ru, t);
    orc_mul3(t, Ki, Hd);
    float Hm[9];
    for (int i = 0; i < 9; ++i) Hm[i] = (float)Hd[i];
    size_t P = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float X = Hm[0] * x + Hm[1] * y + Hm[2], Y = Hm[3] * x + Hm[4] * y + Hm[5], Z = Hm[6] * x + Hm[7] * y + Hm[8];
            float sx = X / Z, sy = Y / Z;
            int in = Z > 0 && sx >= 0 && sx <= (float)(W - 1) && sy >= 0 && sy <= (float)(H - 1);
            for (int c = 0; c < C; ++c) out[c * P + (size_t)y * W + x] = in ? orc_bilin_clamped(img + c * P, H, W, sy, sx) : 0.f;
            if (mask) mask[(size_t)y * W + x] = in ? 1.f : 0.f;
        }
    return 0;
}

/* ref: `img = sfm2.undistortImage(img, calibrationp.K, calibrationp.distortion)` radial/radial_opticalflow_data.lua:24,
 *      depth_estimation_api.lua:139 with the 5-entry distortion vectors of the .cal files = (k1, k2, p1, p2, k3), the
 *      radial-tangential model: out(p) = bilinear(img, K distort(K^-1 p)). */
void orc_undistort_image(const float *img, int C, int H, int W, const double *K9, const double *d5, float *out) {
    float fx = (float)K9[0], fy = (float)K9[4], cx = (float)K9[2], cy = (float)K9[5];
    float k1 = (float)d5[0], k2 = (float)d5[1], p1 = (float)d5[2], p2 = (float)d5[3], k3 = (float)d5[4];
    size_t P = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float xn = ((float)x - cx) / fx, yn = ((float)y - cy) / fy;
            float r2 = xn * xn + yn * yn;
            float rad = 1.f + r2 * (k1 + r2 * (k2 + r2 * k3));
            float xd = xn * rad + 2.f * p1 * xn * yn + p2 * (r2 + 2.f * xn * xn);
            float yd = yn * rad + p1 * (r2 + 2.f * yn * yn) + 2.f * p2 * xn * yn;
            float sx = xd * fx + cx, sy = yd * fy + cy;
            int in = sx >= 0 && sx <= (float)(W - 1) && sy >= 0 && sy <= (float)(H - 1);
            for (int c = 0; c < C; ++c) out[c * P + (size_t)y * W + x] = in ? orc_bilin_clamped(img + c * P, H, W, sy, sx) : 0.f;
        }
}

/* NOT IN THE REFERENCE (the library's dense-flow focus-of-expansion estimator, dfe_foe_from_flow_f32): least-squares intersection
 * of the flow lines, Huber re-weighted.  Restated here only so that the device code has a CPU counterpart. */
int orc_foe_from_flow(const float *fy, const float *fx, const float *conf, int H, int W, float min_flow, int iterations, double *foe, double *n_used) {
    double cx = W / 2.0, cy = H / 2.0;
    for (int it = 0; it <= iterations; ++it) {
        double s[5] = {0, 0, 0, 0, 0};
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t e = (size_t)y * W + x;
                float u = fx[e], v = fy[e];
                float mag = sqrtf(u * u + v * v);
                if (mag < min_flow || (conf && conf[e] <= 0.f)) continue;
                double nx = -v / mag, ny = u / mag, w = 1.0;
                if (it > 0) {
                    double r = fabs(nx * ((double)(float)cx - x) + ny * ((double)(float)cy - y));
                    w = r <= 2.0 ? 1.0 : 2.0 / r;
                }
                double np = nx * x + ny * y;
                s[0] += w * nx * nx; s[1] += w * nx * ny; s[2] += w * ny * ny; s[3] += w * nx * np; s[4] += w * ny * np;
            }
        double det = s[0] * s[2] - s[1] * s[1];
        if (n_used) *n_used = s[0] + s[2];
        if (!(fabs(det) > 1e-9 * (s[0] + s[2]) * (s[0] + s[2]) + 1e-300)) return -1;
        cx = (s[2] * s[3] - s[1] * s[4]) / det;
        cy = (s[0] * s[4] - s[1] * s[3]) / det;
    }
    foe[0] = cx; foe[1] = cy;
    return 0;
}

/* ---- relative pose from correspondences --------------------------------------------------------------------------------
 * ref: `R, T, nFound, nInliers, fundmat = sfm2.getEgoMotion2{im1, im2, K, maxPoints, pointsQuality, ransacMaxDist,
 *      pointsMinDistance}` radial/radial_opticalflow_data.lua:211-217, radial/test_radial_opticalflow.lua:122-126;
 *      `sfm2.getEgoMotion(last_im, im, Kf, 400)` depth_estimation_api.lua:141.  Used as: e2 = K T (data.lua:218),
 *      removeEgoMotion(prev, K, R) (:231), nInliers / nFound < bad_image_threshold (:222).
 * sfm2 finds its correspondences itself (OpenCV corners + LK); here they are an input.  Then the textbook pipeline: RANSAC over
 * 8-point fundamental-matrix hypotheses (Sampson distance), least-squares refit over the consensus set, projection onto the
 * essential manifold, the four-fold decomposition, cheirality.  Same counter-based sampling as the device code (so that both draw
 * the same hypotheses), everything else written independently. */
static void orc_jacobi(double *A, double *V, int n) {   /* symmetric A (n x n, row-major) -> eigenvalues on the diagonal, vectors = columns of V */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int i = 0; i < n; ++i) {
            dg += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-32 * dg) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p * n + q];
                if (apq == 0) continue;
                double th = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double a = A[k * n + p], b = A[k * n + q];
                    A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b;
                }
                for (int k = 0; k < n; ++k) {
                    double a = A[p * n + k], b = A[q * n + k];
                    A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b;
                }
                for (int k = 0; k < n; ++k) {
                    double a = V[k * n + p], b = V[k * n + q];
                    V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b;
                }
            }
    }
}
static unsigned orc_ego_rand(unsigned seed, unsigned h, unsigned k) {
    unsigned x = seed * 0x9E3779B1u ^ (h + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (k + 0x165667B1u) * 0xC2B2AE35u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
static void orc_cam(const double *Ki, double u, double v, double *o) {
    o[0] = Ki[0] * u + Ki[1] * v + Ki[2]; o[1] = Ki[3] * u + Ki[4] * v + Ki[5]; o[2] = Ki[6] * u + Ki[7] * v + Ki[8];
}
/* E -> U diag(1,1,0) V^T (U, V right-handed, columns); 0 on success */
static int orc_essential_svd(const double *E, double *Ep, double *U, double *V) {
    double M[9], Q[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
    orc_jacobi(M, Q, 3);
    int o[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (M[o[b] * 4] > M[o[a] * 4]) { int t = o[a]; o[a] = o[b]; o[b] = t; }
    double s1 = sqrt(M[o[0] * 4] > 0 ? M[o[0] * 4] : 0), s2 = sqrt(M[o[1] * 4] > 0 ? M[o[1] * 4] : 0);
    if (!(s1 > 0) || !(s2 > 1e-9 * s1)) return -1;
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
    for (int i = 0; i < 3; ++i) { v1[i] = Q[i * 3 + o[0]]; v2[i] = Q[i * 3 + o[1]]; }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1]; v3[1] = v1[2] * v2[0] - v1[0] * v2[2]; v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    for (int i = 0; i < 3; ++i) {
        u1[i] = (E[i * 3] * v1[0] + E[i * 3 + 1] * v1[1] + E[i * 3 + 2] * v1[2]) / s1;
        u2[i] = (E[i * 3] * v2[0] + E[i * 3 + 1] * v2[1] + E[i * 3 + 2] * v2[2]) / s2;
    }
    double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (int i = 0; i < 3; ++i) u1[i] /= n1;
    double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
    for (int i = 0; i < 3; ++i) u2[i] -= d * u1[i];
    double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
    if (!(n2 > 1e-12)) return -1;
    for (int i = 0; i < 3; ++i) u2[i] /= n2;
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1]; u3[1] = u1[2] * u2[0] - u1[0] * u2[2]; u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Ep[i * 3 + j] = u1[i] * v1[j] + u2[i] * v2[j];
    if (U)
        for (int i = 0; i < 3; ++i) {
            U[i * 3] = u1[i]; U[i * 3 + 1] = u2[i]; U[i * 3 + 2] = u3[i];
            V[i * 3] = v1[i]; V[i * 3 + 1] = v2[i]; V[i * 3 + 2] = v3[i];
        }
    return 0;
}
static void orc_fund(const double *E, const double *Ki, double *F) {
    double KiT[9], t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KiT[i * 3 + j] = Ki[j * 3 + i];
    orc_mul3(KiT, E, t);
    orc_mul3(t, Ki, F);
}
static double orc_sampson2(const double *F, double x1, double y1, double x2, double y2) {
    double a0 = F[0] * x1 + F[1] * y1 + F[2], a1 = F[3] * x1 + F[4] * y1 + F[5], a2 = F[6] * x1 + F[7] * y1 + F[8];
    double b0 = F[0] * x2 + F[3] * y2 + F[6], b1 = F[1] * x2 + F[4] * y2 + F[7];
    double r = x2 * a0 + y2 * a1 + a2, den = a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
    return den > 0 ? r * r / den : 1e300;
}
static void orc_normal_rows(const double *Ki, const float *p1, const float *p2, int n, double *A /* 81, accumulated */) {
    double a[3], b[3], r[9];
    orc_cam(Ki, p1[2 * n], p1[2 * n + 1], a);
    orc_cam(Ki, p2[2 * n], p2[2 * n + 1], b);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = b[i] * a[j];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) A[i * 9 + j] += r[i] * r[j];
}
static int orc_null_essential(double *A, double *Ep, double *U, double *V) {
    double Q[81], E[9];
    orc_jacobi(A, Q, 9);
    int mn = 0;
    for (int i = 1; i < 9; ++i)
        if (A[i * 10] < A[mn * 10]) mn = i;
    for (int i = 0; i < 9; ++i) E[i] = Q[i * 9 + mn];
    return orc_essential_svd(E, Ep, U, V);
}

int orc_ego_motion_from_points(const float *p1, const float *p2, const float *w, int N, const double *K9, double max_dist, int iterations, unsigned seed,
                               double *R9, double *T3, int *n_inliers, double *F9) {
    double Ki[9];
    if (N < 8 || orc_inv3(K9, Ki)) return -1;
    double bestF[9];
    int bestc = -1;
    const double md2 = max_dist * max_dist;
    for (int h = 0; h < iterations; ++h) {
        int pick[8], ok = 1;
        unsigned k = 0;
        for (int n = 0; n < 8 && ok; ++n)
            for (int tries = 0;; ++tries) {
                if (tries > 64) { ok = 0; break; }
                int c = (int)(orc_ego_rand(seed, (unsigned)h, k++) % (unsigned)N);
                if (w && !(w[c] > 0.f)) continue;
                int dup = 0;
                for (int m = 0; m < n; ++m) dup |= pick[m] == c;
                if (!dup) { pick[n] = c; break; }
            }
        if (!ok) continue;
        double A[81] = {0}, Ep[9], F[9];
        for (int n = 0; n < 8; ++n) orc_normal_rows(Ki, p1, p2, pick[n], A);
        if (orc_null_essential(A, Ep, NULL, NULL)) continue;
        orc_fund(Ep, Ki, F);
        int c = 0;
        for (int n = 0; n < N; ++n)
            if (!w || w[n] > 0.f) c += orc_sampson2(F, p1[2 * n], p1[2 * n + 1], p2[2 * n], p2[2 * n + 1]) <= md2;
        if (c > bestc) { bestc = c; memcpy(bestF, F, sizeof F); }
    }
    if (bestc < 8) return -2;
    double A[81] = {0}, Ep[9], U[9], V[9];
    unsigned char *mask = (unsigned char *)calloc((size_t)N, 1);
    for (int n = 0; n < N; ++n)
        if ((!w || w[n] > 0.f) && orc_sampson2(bestF, p1[2 * n], p1[2 * n + 1], p2[2 * n], p2[2 * n + 1]) <= md2) { mask[n] = 1; orc_normal_rows(Ki, p1, p2, n, A); }
    if (orc_null_essential(A, Ep, U, V)) { free(mask); return -3; }
    const double Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double VT[9], t[9], Ra[9], Rb[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) VT[i * 3 + j] = V[j * 3 + i];
    orc_mul3(U, Wm, t); orc_mul3(t, VT, Ra);
    orc_mul3(U, Wt, t); orc_mul3(t, VT, Rb);
    double u3[3] = {U[2], U[5], U[8]};
    int bc = -1, bg = -1;
    for (int c = 0; c < 4; ++c) {
        const double *R = c < 2 ? Ra : Rb;
        double tt[3] = {(c & 1) ? -u3[0] : u3[0], (c & 1) ? -u3[1] : u3[1], (c & 1) ? -u3[2] : u3[2]};
        int good = 0;
        for (int n = 0; n < N; ++n) {
            if (!mask[n]) continue;
            double a[3], b[3], ra[3];
            orc_cam(Ki, p1[2 * n], p1[2 * n + 1], a);
            orc_cam(Ki, p2[2 * n], p2[2 * n + 1], b);
            for (int i = 0; i < 3; ++i) ra[i] = R[i * 3] * a[0] + R[i * 3 + 1] * a[1] + R[i * 3 + 2] * a[2];
            double m00 = ra[0] * ra[0] + ra[1] * ra[1] + ra[2] * ra[2], m01 = -(ra[0] * b[0] + ra[1] * b[1] + ra[2] * b[2]);
            double m11 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
            double r0 = -(ra[0] * tt[0] + ra[1] * tt[1] + ra[2] * tt[2]), r1 = b[0] * tt[0] + b[1] * tt[1] + b[2] * tt[2];
            double det = m00 * m11 - m01 * m01;
            if (fabs(det) < 1e-18) continue;
            double z1 = (r0 * m11 - m01 * r1) / det, z2 = (m00 * r1 - m01 * r0) / det;
            good += z1 > 0 && z2 > 0;
        }
        if (good > bg) { bg = good; bc = c; }
    }
    memcpy(R9, bc < 2 ? Ra : Rb, sizeof Ra);
    for (int i = 0; i < 3; ++i) T3[i] = (bc & 1) ? -u3[i] : u3[i];
    double F[9], fn = 0;
    orc_fund(Ep, Ki, F);
    for (int i = 0; i < 9; ++i) fn += F[i] * F[i];
    fn = sqrt(fn);
    int cnt = 0;
    for (int n = 0; n < N; ++n)
        if (mask[n]) cnt += orc_sampson2(F, p1[2 * n], p1[2 * n + 1], p2[2 * n], p2[2 * n + 1]) <= md2;
    if (n_inliers) *n_inliers = cnt;
    if (F9)
        for (int i = 0; i < 9; ++i) F9[i] = F[i] / (fn > 0 ? fn : 1);
    free(mask);
    return 0;
}

/* image.rgb2y as prepareInput calls it (opticalflow_model.lua:136-138): y = 0.299 R + 0.587 G + 0.114 B, accumulated in that order
 * (output:zero():add(0.299, R):add(0.587, G):add(0.114, B) -- THTensor cadd, float products and sums rounded separately).
 * `image` is an un-vendored package: restated from recall, parity unpinned. */
void orc_rgb2y(const float *rgb, int H, int W, float *y) {
    const long long P = (long long)H * W;
    for (long long e = 0; e < P; ++e) {
        volatile float v = 0.299f * rgb[e];
        volatile float t = 0.587f * rgb[P + e];
        v = v + t;
        t = 0.114f * rgb[2 * P + e];
        v = v + t;
        y[e] = v;
    }
}
