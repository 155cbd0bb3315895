// ssd_cost_volume.hip -- A0+A1 / A1 / A1r: the dense SSD cost-volume build for gfx950.
//
// Two kernels produce out[y][x][dy][dx] = sum_k (a_k(y,x) - b_k(y+dy,x+dx))^2 in the reference's
// pixel-major layout (nn.SpatialMatching output H1 x W1 x maxh x maxw,
// opticalflow_model.lua:93-98, radial/radial_opticalflow_groundtruth.lua:83-87):
//
//  ssd_cv_ref_kernel    any C / kernel / window / feature maps.  One thread per output element,
//                       features summed in the reference order (c, i, j) with separately rounded
//                       multiply and add, so it is bit-identical to the CPU loop nest for any
//                       float input.  O(C*kh*kw) flops per output: correct everywhere, fast nowhere.
//
//  ssd_cv_tiled_kernel  the HBM-bound kernel (raw frames, C in {1,3}, square kernel K).
//                       The patch sum is a K x K box filter of the per-displacement squared-
//                       difference plane e_d(u,v) = sum_c (I0[c][u][v] - I1[c][u+dy-oy][v+dx-ox])^2,
//                       evaluated separably (horizontal K-sum, then a K-row register ring), so an
//                       output costs ~25 VALU ops instead of 3*C*K*K.
//                         * lane <-> displacement d = dy*wWin+dx: the 64 lanes of a wave own 64
//                           consecutive cells of one pixel's window, so every global store is a
//                           full-wave 256-B contiguous burst in the reference layout and the
//                           volume is written exactly once, with no LDS transpose;
//                         * frame-0 values are wave-uniform -> scalar (SMEM) loads, zero VALU/LDS;
//                         * the frame-1 tile (+ search halo) is staged once per block in LDS as
//                           channel-interleaved pixels; consecutive lanes read consecutive pixels
//                           (row pitch == wWin mod 16 keeps a wave that spans several dy rows
//                           conflict-free), one ds_read_b128 per squared difference;
//                         * edge tiles are shifted inwards instead of masked, so no lane ever
//                           reads out of bounds; duplicate pixels are not stored twice.
//                       Summation order is fixed relative to the output pixel (independent of the
//                       tile), so results do not depend on the launch geometry.
#include "dfe_internal.h"

// ------------------------------------------------------------------------------------------
// reference-order kernel
// ------------------------------------------------------------------------------------------
struct CvRefArgs {
    const float *a; long long a_plane; int a_pitch; int a_oy, a_ox;   // frame0 / in1
    const float *b; long long b_plane; int b_pitch;                   // frame1 / in2
    int C, kh, kw, Wo, hWin, wWin;
    long long total;                                                  // Ho*Wo*hWin*wWin
    float *out;
};

__global__ __launch_bounds__(256) void ssd_cv_ref_kernel(CvRefArgs p) {
#pragma clang fp contract(off)
    const int D = p.hWin * p.wWin;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < p.total; e += (long long)gridDim.x * 256) {
        long long pix = e / D;
        int d = (int)(e - pix * D);
        int y = (int)(pix / p.Wo), x = (int)(pix - (long long)y * p.Wo);
        int dy = d / p.wWin, dx = d - dy * p.wWin;
        float s = 0.f;
        for (int c = 0; c < p.C; ++c)
            for (int i = 0; i < p.kh; ++i) {
                const float *ap = p.a + c * p.a_plane + (long long)(y + p.a_oy + i) * p.a_pitch + x + p.a_ox;
                const float *bp = p.b + c * p.b_plane + (long long)(y + dy + i) * p.b_pitch + x + dx;
                for (int j = 0; j < p.kw; ++j) {
                    float t = ap[j] - bp[j];
                    float t2 = t * t;
                    s = s + t2;
                }
            }
        p.out[e] = s;
    }
}

static int launch_cv_ref(dfe_ctx *ctx, const CvRefArgs &a) {
    if (a.total <= 0) return DFE_OK;
    long long blocks = (a.total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(ssd_cv_ref_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "ssd_cv_ref_kernel";
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// tiled kernel
// ------------------------------------------------------------------------------------------
template <int C> struct Px;
template <> struct Px<1> { using type = float;  static constexpr int bank_mod = 32; };
template <> struct Px<3> { using type = float4; static constexpr int bank_mod = 16; };

template <int C> __device__ __forceinline__ float sqdiff(const float (&a)[C], const typename Px<C>::type &b);
template <> __device__ __forceinline__ float sqdiff<1>(const float (&a)[1], const float &b) {
    float d = a[0] - b;
    return d * d;
}
template <> __device__ __forceinline__ float sqdiff<3>(const float (&a)[3], const float4 &b) {
    float d0 = a[0] - b.x, d1 = a[1] - b.y, d2 = a[2] - b.z;
    return fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
}

// horizontal K-sum of e[0..TX+K-2] -> h[0..TX-1]; fixed association relative to x.
template <int K, int TX> __device__ __forceinline__ void hsum(const float (&e)[TX + K - 1], float (&h)[TX]) {
    if constexpr (K == 7) {
        float p2[TX + 5], p4[TX + 3];
#pragma unroll
        for (int s = 0; s < TX + 5; ++s) p2[s] = e[s] + e[s + 1];
#pragma unroll
        for (int s = 0; s < TX + 3; ++s) p4[s] = p2[s] + p2[s + 2];
#pragma unroll
        for (int x = 0; x < TX; ++x) h[x] = (p4[x] + p2[x + 4]) + e[x + 6];
    } else {
#pragma unroll
        for (int x = 0; x < TX; ++x) {
            float s = e[x];
#pragma unroll
            for (int j = 1; j < K; ++j) s += e[x + j];
            h[x] = s;
        }
    }
}

// Wave-uniform loads of N consecutive floats through the scalar cache (s_load_dwordx8/x4/x2/x1).
// The constant address space makes the backend select SMEM; dword alignment is all SMEM needs.
typedef const float __attribute__((address_space(4))) *cfptr;
typedef float f8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef f8_t f8u_t __attribute__((aligned(4)));
typedef f4_t f4u_t __attribute__((aligned(4)));
typedef f2_t f2u_t __attribute__((aligned(4)));
template <int N, int OFF = 0> __device__ __forceinline__ void uload(cfptr p, float *v) {
    if constexpr (N >= 8) {
        f8_t t = *(const __attribute__((address_space(4))) f8u_t *)(p + OFF);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[OFF + i] = t[i];
        uload<N - 8, OFF + 8>(p, v);
    } else if constexpr (N >= 4) {
        f4_t t = *(const __attribute__((address_space(4))) f4u_t *)(p + OFF);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[OFF + i] = t[i];
        uload<N - 4, OFF + 4>(p, v);
    } else if constexpr (N >= 2) {
        f2_t t = *(const __attribute__((address_space(4))) f2u_t *)(p + OFF);
        v[OFF] = t[0];
        v[OFF + 1] = t[1];
        uload<N - 2, OFF + 2>(p, v);
    } else if constexpr (N == 1) {
        v[OFF] = p[OFF];
    }
}

struct CvTiledArgs {
    int H, W, hWin, wWin, Ho, Wo;
    long long plane;   // channel plane stride in elements (H*W of the full frame, also for row bands)
    int pitch;   // LDS row pitch in pixels
    int lrows;   // LDS rows = TYQ*K + hWin - 1
    int lcols;   // staged columns = TX + K - 1 + wWin - 1
};

extern __shared__ __attribute__((aligned(16))) char dfe_smem[];

// TYQ*K image rows are swept per tile -> TY = TYQ*K-(K-1) output rows.
template <int C, int K, int TX, int TYQ, int NW>
__global__ __launch_bounds__(NW * 64) void ssd_cv_tiled_kernel(const float *__restrict__ I0, const float *__restrict__ I1,
                                                               float *__restrict__ out, CvTiledArgs p) {
    using px_t = typename Px<C>::type;
    constexpr int TY = TYQ * K - (K - 1);
    px_t *lds = reinterpret_cast<px_t *>(dfe_smem);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0n = blockIdx.x * TX, y0n = blockIdx.y * TY;           // nominal origin: what this tile stores
    const int x0 = min(x0n, p.Wo - TX), y0 = min(y0n, p.Ho - TY);      // shifted origin: what it computes
    const long long HW = p.plane;

    // ---- stage the frame-1 tile: rows y0.., cols x0.. (always inside the frame, see header) ----
    for (int r = wave; r < p.lrows; r += NW) {
        const float *src = I1 + (long long)(y0 + r) * p.W + x0;
        for (int s = lane; s < p.lcols; s += 64) {
            if constexpr (C == 1) {
                lds[r * p.pitch + s] = src[s];
            } else {
                lds[r * p.pitch + s] = make_float4(src[s], src[HW + s], src[2 * HW + s], 0.f);
            }
        }
    }
    __syncthreads();

    const int D = p.hWin * p.wWin;
    const int nchunks = (D + 63) >> 6;
    const int oy = (p.hWin - 1) >> 1, ox = (p.wWin - 1) >> 1;

    for (int chunk = wave; chunk < nchunks; chunk += NW) {
        const int d = chunk * 64 + lane;
        if (d < D) {   // one divergent region per chunk (only the last chunk is partial): no per-store branch
            const int dy = d / p.wWin, dx = d - dy * p.wWin;
            const px_t *lp = lds + dy * p.pitch + dx;
            const unsigned dbytes = (unsigned)d * 4u;   // 32-bit lane offset -> saddr+voffset stores

            float ring[K][TX];
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int x = 0; x < TX; ++x) ring[i][x] = 0.f;

            for (int q = 0; q < TYQ; ++q) {
#pragma unroll
                for (int m = 0; m < K; ++m) {
                    const int r = q * K + m;
                    cfptr a = (cfptr)(I0 + (long long)(y0 + oy + r) * p.W + (x0 + ox));   // wave-uniform
                    const px_t *lr = lp + r * p.pitch;
                    float av[C][TX + K - 1];
#pragma unroll
                    for (int c = 0; c < C; ++c) uload<TX + K - 1>(a + c * HW, av[c]);
                    float e[TX + K - 1];
#pragma unroll
                    for (int s = 0; s < TX + K - 1; ++s) {
                        float a3[C];
#pragma unroll
                        for (int c = 0; c < C; ++c) a3[c] = av[c][s];
                        px_t b = lr[s];
                        if constexpr (C == 3) asm volatile("" ::"v"(b.w));   // keep the 16-B read (b128, not the slower b96)
                        e[s] = sqdiff<C>(a3, b);
                    }
                    hsum<K, TX>(e, ring[m]);
                    if (q > 0 || m == K - 1) {
                        const int y = y0 + r - (K - 1);
                        if (y >= y0n) {
#pragma unroll
                            for (int x = 0; x < TX; ++x) {
                                // oldest row first: association fixed relative to the output pixel
                                float v = ring[(m + 1) % K][x];
#pragma unroll
                                for (int i = 2; i <= K; ++i) v += ring[(m + i) % K][x];
                                if (x0 + x >= x0n) {
                                    char *op = (char *)(out + ((long long)y * p.Wo + (x0 + x)) * D);   // wave-uniform
                                    *(float *)(op + dbytes) = v;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int C, int K, int TX, int TYQ, int NW>
static int launch_cv_tiled(dfe_ctx *ctx, const float *I0, const float *I1, int H, int W, long long plane, int hWin,
                           int wWin, float *out, bool *handled) {
    constexpr int TY = TYQ * K - (K - 1);
    using px_t = typename Px<C>::type;
    const int Ho = H - K + 1 - hWin + 1, Wo = W - K + 1 - wWin + 1;
    *handled = false;
    if (Ho < TY || Wo < TX) return DFE_OK;
    CvTiledArgs a;
    a.plane = plane;
    a.H = H; a.W = W; a.hWin = hWin; a.wWin = wWin; a.Ho = Ho; a.Wo = Wo;
    a.lrows = TYQ * K + hWin - 1;
    a.lcols = TX + K - 1 + wWin - 1;
    const int M = Px<C>::bank_mod;
    int pitch = a.lcols;                        // smallest pitch >= lcols with pitch == wWin (mod M)
    while ((pitch - wWin) % M != 0) ++pitch;
    a.pitch = pitch;
    size_t lds_bytes = (size_t)a.lrows * pitch * sizeof(px_t);
    if (lds_bytes > 160 * 1024) return DFE_OK;  // window too large for one tile: caller falls back
    auto kern = ssd_cv_tiled_kernel<C, K, TX, TYQ, NW>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    dim3 grid(dfe_cdiv(Wo, TX), dfe_cdiv(Ho, TY));
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, ctx->stream, I0, I1, out, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = "ssd_cv_tiled_kernel";
    *handled = true;
    return DFE_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
// H is the number of frame rows visible to this call (a row band of a taller frame when plane > H*W)
static int cv_frames_dispatch(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, long long plane, int kh,
                              int kw, int hWin, int wWin, float *out) {
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    if (ctx->cv_mode != 1 && kh == kw) {
        bool handled = false;
        int rc = DFE_OK;
        if (C == 3 && kh == 7) rc = launch_cv_tiled<3, 7, 8, 5, 6>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 1 && kh == 7) rc = launch_cv_tiled<1, 7, 8, 5, 6>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 3 && kh == 5) rc = launch_cv_tiled<3, 5, 8, 6, 6>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        else if (C == 3 && kh == 3) rc = launch_cv_tiled<3, 3, 8, 8, 6>(ctx, I0, I1, H, W, plane, hWin, wWin, out, &handled);
        if (rc != DFE_OK || handled) return rc;
    }
    if (ctx->cv_mode == 2)
        return dfe_fail(ctx, DFE_E_UNSUPPORTED, "no tiled cost-volume kernel for C=%d k=%dx%d win=%dx%d out=%dx%d", C, kh,
                        kw, hWin, wWin, Ho, Wo);
    CvRefArgs a;
    a.a = I0; a.a_plane = plane; a.a_pitch = W; a.a_oy = (hWin - 1) / 2; a.a_ox = (wWin - 1) / 2;
    a.b = I1; a.b_plane = plane; a.b_pitch = W;
    a.C = C; a.kh = kh; a.kw = kw; a.Wo = Wo; a.hWin = hWin; a.wWin = wWin;
    a.total = (long long)Ho * Wo * hWin * wWin;
    a.out = out;
    return launch_cv_ref(ctx, a);
}

extern "C" {

int dfe_ssd_cost_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw,
                            int hWin, int wWin, float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, I0 && I1 && out, DFE_E_ARG, "dfe_ssd_cost_volume_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && kh > 0 && kw > 0 && hWin > 0 && wWin > 0, DFE_E_ARG,
                "dfe_ssd_cost_volume_f32: C=%d k=%dx%d win=%dx%d must be positive", C, kh, kw, hWin, wWin);
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE,
                "dfe_ssd_cost_volume_f32: frame %dx%d too small for kernel %dx%d + window %dx%d", H, W, kh, kw, hWin, wWin);
    return cv_frames_dispatch(ctx, I0, I1, C, H, W, (long long)H * W, kh, kw, hWin, wWin, out);
}

// rows of the output volume that fit one scratch band (ctx->scratch_limit, 16 GiB by default:
// one launch per pair up to 1080p/33x33; 288 GB of HBM make the whole volume the natural unit)
static int band_rows(const dfe_ctx *ctx, int Ho, int Wo, int D) {
    long long row_bytes = (long long)Wo * D * sizeof(float);
    long long band = (long long)ctx->scratch_limit / row_bytes;
    if (band < 1) band = 1;
    if (band > Ho) band = Ho;
    return (int)band;
}

int dfe_ssd_flow_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int kh, int kw, int hWin,
                     int wWin, double extract_threshold, int64_t *idx, float *best, float *flow_y, float *flow_x,
                     float *scores, int64_t *imaxs) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, I0 && I1, DFE_E_ARG, "dfe_ssd_flow_f32: NULL frame");
    DFE_REQUIRE(ctx, C > 0 && kh > 0 && kw > 0 && hWin > 0 && wWin > 0, DFE_E_ARG,
                "dfe_ssd_flow_f32: C=%d k=%dx%d win=%dx%d must be positive", C, kh, kw, hWin, wWin);
    const int Ho = H - kh + 1 - hWin + 1, Wo = W - kw + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_ssd_flow_f32: frame %dx%d too small for kernel %dx%d + window %dx%d",
                H, W, kh, kw, hWin, wWin);
    DFE_REQUIRE(ctx, (scores == nullptr) == (imaxs == nullptr), DFE_E_ARG, "dfe_ssd_flow_f32: scores and imaxs go together");
    const int D = hWin * wWin;
    const int band = band_rows(ctx, Ho, Wo, D);
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (size_t)band * Wo * D * sizeof(float), &scr);
    if (rc) return rc;
    float *vol = (float *)scr;
    for (int r0 = 0; r0 < Ho; r0 += band) {
        int nr = (r0 + band <= Ho) ? band : Ho - r0;
        int Hb = nr + kh - 1 + hWin - 1;
        rc = cv_frames_dispatch(ctx, I0 + (long long)r0 * W, I1 + (long long)r0 * W, C, Hb, W, (long long)H * W, kh, kw, hWin,
                                wWin, vol);
        if (rc) return rc;
        rc = dfe_flow_tail(ctx, vol, nr, Wo, hWin, wWin, extract_threshold, r0, idx, best, flow_y, flow_x, scores, imaxs, Wo, 0,
                           0, 0);
        if (rc) return rc;
    }
    return DFE_OK;
}

int dfe_flow_depth_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int hWin, int wWin,
                            float foe_x, float foe_y, double extract_threshold, float *flow, float *scores, float *depth,
                            float *depth_conf) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, I0 && I1 && flow, DFE_E_ARG, "dfe_flow_depth_pair_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && k > 0 && hWin > 0 && wWin > 0, DFE_E_ARG, "dfe_flow_depth_pair_f32: C=%d k=%d win=%dx%d", C, k,
                hWin, wWin);
    DFE_REQUIRE(ctx, (depth == nullptr) == (depth_conf == nullptr), DFE_E_ARG,
                "dfe_flow_depth_pair_f32: depth and depth_conf go together");
    const int Ho = H - k + 1 - hWin + 1, Wo = W - k + 1 - wWin + 1;
    DFE_REQUIRE(ctx, Ho > 0 && Wo > 0, DFE_E_SHAPE, "dfe_flow_depth_pair_f32: frame %dx%d too small for kernel %d + window %dx%d",
                H, W, k, hWin, wWin);
    const int D = hWin * wWin;
    const long long HW = (long long)H * W;
    // centre-paste offsets: opticalflow_model.lua:228-230 floor((hImg-h)/2)
    //   == radial/radial_opticalflow_groundtruth.lua:27-32 floor((hWin-1)/2)+floor((k-1)/2)
    const int pad_t = (H - Ho) / 2, pad_l = (W - Wo) / 2;
    DFE_HIP(ctx, hipMemsetAsync(flow, 0, 2 * HW * sizeof(float), ctx->stream));
    if (scores) DFE_HIP(ctx, hipMemsetAsync(scores, 0, HW * sizeof(float), ctx->stream));
    const int band = band_rows(ctx, Ho, Wo, D);
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (size_t)band * Wo * D * sizeof(float), &scr);
    if (rc) return rc;
    float *vol = (float *)scr;
    for (int r0 = 0; r0 < Ho; r0 += band) {
        int nr = (r0 + band <= Ho) ? band : Ho - r0;
        int Hb = nr + k - 1 + hWin - 1;
        rc = cv_frames_dispatch(ctx, I0 + (long long)r0 * W, I1 + (long long)r0 * W, C, Hb, W, HW, k, k, hWin, wWin, vol);
        if (rc) return rc;
        rc = dfe_flow_tail(ctx, vol, nr, Wo, hWin, wWin, extract_threshold, r0, nullptr, nullptr, flow, flow + HW, scores,
                           nullptr, W, pad_t, pad_l, 1);
        if (rc) return rc;
    }
    if (depth) {
        rc = dfe_flow_to_depth_cartesian(ctx, flow, H, W, foe_x, foe_y, 0, depth, depth_conf);
        if (rc) return rc;
    }
    return DFE_OK;
}

int dfe_spatial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh,
                             int maxw, float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_spatial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W1 > 0 && maxh > 0 && maxw > 0, DFE_E_SHAPE,
                "dfe_spatial_matching_f32: K=%d H1=%d W1=%d maxh=%d maxw=%d must be positive", K, H1, W1, maxh, maxw);
    CvRefArgs a;
    a.a = in1; a.a_plane = (long long)H1 * W1; a.a_pitch = W1; a.a_oy = 0; a.a_ox = 0;
    a.b = in2; a.b_plane = (long long)(H1 + maxh - 1) * (W1 + maxw - 1); a.b_pitch = W1 + maxw - 1;
    a.C = K; a.kh = 1; a.kw = 1; a.Wo = W1; a.hWin = maxh; a.wWin = maxw;
    a.total = (long long)H1 * W1 * maxh * maxw;
    a.out = out;
    return launch_cv_ref(ctx, a);
}

int dfe_radial_matching_f32(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W, int hWin,
                            float *out) {
    DFE_REQUIRE(ctx, ctx, DFE_E_ARG, "ctx is NULL");
    DFE_REQUIRE(ctx, in1 && in2 && out, DFE_E_ARG, "dfe_radial_matching_f32: NULL tensor");
    DFE_REQUIRE(ctx, K > 0 && H1 > 0 && W > 0 && hWin > 0, DFE_E_SHAPE,
                "dfe_radial_matching_f32: K=%d H1=%d W=%d hWin=%d must be positive", K, H1, W, hWin);
    return dfe_spatial_matching_f32(ctx, in1, in2, K, H1, W, hWin, 1, out);
}

}  // extern "C"
