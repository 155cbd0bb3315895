// multiscale.hip -- A2..A5: pyramid volumes, per-scale softmin, CascadingAddTable, ring extraction.
//   replaces: getMultiscalePrefilter / SpatialPyramid geometry  opticalflow_model_multiscale.lua:134-229
//             cascad_preproc (Minus -> SoftMax)                  opticalflow_model_multiscale.lua:270-279
//             nn.CascadingAddTable:updateOutput                  CascadingAddTable.lua:108-135
//             the "middle remover" + JoinTable + SmartReshape    opticalflow_model_multiscale.lua:293-333
// The reference materialises the x r nearest-neighbour upsampling of every coarse volume and ~15 full
// H x W x 64 temporaries; here volumes stay at native scale and one kernel per finest pixel gathers its
// window from each scale, cascades and ring-selects in LDS, writing only the H x W x nclasses result.
#include "dfe_internal.h"
#include <cstring>
#include <type_traits>
#include <cmath>
#include <memory>

namespace {

constexpr int kWaves = 4;
template <int I, int N, class F> __device__ __forceinline__ void static_for_ms(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_ms<I + 1, N>(f);
    }
}

// ---- nn.SpatialDownSampling(r,r): mean of r x r blocks (row-major accumulation, then * 1/(r*r)) ----
// the r x r box of one output pixel, summed row-major like the generic loop; for the usual ratios every load is issued
// before the first add (a run-time loop of load-add pairs makes the r = 4 scale a chain of 16 memory latencies)
// a frame pixel as the pipeline's fp32 value: fp32 frames as they are, uint8 frames as float(byte) * scale -- the expression of
// u8_to_f32_kernel (ingest.hip), so the preparation kernels reading bytes give the bits of conversion pass + fp32 preparation
__device__ __forceinline__ float px_ld(const float *__restrict__ p, long long i, float) { return p[i]; }
__device__ __forceinline__ float px_ld(const unsigned char *__restrict__ p, long long i, float sc) {
#pragma clang fp contract(off)
    return (float)p[i] * sc;
}
template <int R, typename T = float> __device__ __forceinline__ float box_sum(const T *__restrict__ src, int W, float sc = 0.f) {
#pragma clang fp contract(off)
    float t[R * R];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) t[i * R + j] = px_ld(src, (long long)i * W + j, sc);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < R * R; ++k) s = s + t[k];
    return s;
}
__global__ void downsample_box_kernel(const float *__restrict__ img, int C, int H, int W, int r, float *__restrict__ out) {
#pragma clang fp contract(off)
    const int Ho = H / r, Wo = W / r;
    const long long total = (long long)C * Ho * Wo;
    const float inv = 1.0f / (float)(r * r);
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int x = (int)(e % Wo);
        long long t = e / Wo;
        int y = (int)(t % Ho), c = (int)(t / Ho);
        const float *src = img + ((long long)c * H + y * r) * W + x * r;
        float s = 0.f;
        if (r == 2) s = box_sum<2>(src, W);
        else if (r == 4) s = box_sum<4>(src, W);
        else if (r == 8) s = box_sum<8>(src, W);
        else {
            for (int i = 0; i < r; ++i)
                for (int j = 0; j < r; ++j) s = s + src[(long long)i * W + j];
        }
        out[e] = s * inv;
    }
}

// ---- nn.SpatialZeroPadding(l,r,t,b) with non-negative pads --------------------------------------
__global__ void zero_pad_kernel(const float *__restrict__ img, int C, int H, int W, int pl, int pt, int Hp, int Wp,
                                float *__restrict__ out) {
    const long long total = (long long)C * Hp * Wp;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int x = (int)(e % Wp);
        long long t = e / Wp;
        int y = (int)(t % Hp), c = (int)(t / Hp);
        int sy = y - pt, sx = x - pl;
        out[e] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((long long)c * H + sy) * W + sx] : 0.f;
    }
}

// ---- down-sample by r (same arithmetic as downsample_box_kernel) and zero-pad, both frames, one launch ----
template <typename T>
__device__ __forceinline__ void prep_scale_body(const T *__restrict__ I0, const T *__restrict__ I1, int C, int H, int W, int r, int pl,
                                                int pt, int Hp, int Wp, float *__restrict__ p0, float *__restrict__ p1, float sc) {
#pragma clang fp contract(off)
    const int Hs = H / r, Ws = W / r;
    // 32-bit element indices (the launcher checks 2 * C * Hp * Wp < 2^31): with 64-bit div/mod per element this kernel
    // spent most of its time dividing
    const unsigned total = (unsigned)C * Hp * Wp;
    const float inv = 1.0f / (float)(r * r);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * total; e += gridDim.x * blockDim.x) {
        const bool second = e >= total;
        const unsigned ee = second ? e - total : e;
        const T *img = second ? I1 : I0;
        const unsigned t = ee / (unsigned)Wp;
        const int x = (int)(ee - t * (unsigned)Wp);
        const int c = (int)(t / (unsigned)Hp), y = (int)(t - (unsigned)c * Hp);
        const int sy = y - pt, sx = x - pl;
        float v = 0.f;
        if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
            const T *src = img + ((long long)c * H + sy * r) * W + sx * r;
            float s = 0.f;
            if (r == 1) s = 0.f + px_ld(src, 0, sc);
            else if (r == 2) s = box_sum<2, T>(src, W, sc);
            else if (r == 4) s = box_sum<4, T>(src, W, sc);
            else if (r == 8) s = box_sum<8, T>(src, W, sc);
            else {
                for (int i = 0; i < r; ++i)
                    for (int j = 0; j < r; ++j) s = s + px_ld(src, (long long)i * W + j, sc);
            }
            v = r > 1 ? s * inv : s;
        }
        (second ? p1 : p0)[ee] = v;
    }
}
// every scale of the pyramid in one launch: blockIdx.y = scale (the scales only read the full-resolution frames)
struct PrepScales {
    int r[DFE_MAX_RATIOS], Hp[DFE_MAX_RATIOS], Wp[DFE_MAX_RATIOS];
    float *p0[DFE_MAX_RATIOS], *p1[DFE_MAX_RATIOS];
};
template <typename T>
__global__ void prep_scales_kernel(const T *__restrict__ I0, const T *__restrict__ I1, int C, int H, int W, int pl, int pt, PrepScales ps, float sc) {
    const int s = blockIdx.y;
    prep_scale_body<T>(I0, I1, C, H, W, ps.r[s], pl, pt, ps.Hp[s], ps.Wp[s], ps.p0[s], ps.p1[s], sc);
}

// ---- every scale of both frames from ONE read of the frames (ratios 1, 2, 4, 8, 16) ---------------------------------------------
// prep_scales_kernel re-reads both full-resolution frames once per scale (a hierarchical box sum would change the rounding, so every
// scale sums its own r x r boxes from the original pixels): 266 MB at 1080p for 50 MB of input, 383 us at 4K with five scales.  Here
// a block loads one 64 x 64 tile of one channel of one frame into LDS and produces that tile's part of EVERY scale from it -- the same
// row-major sequential box sums, then * 1/r^2: bit-identical -- including the zero padding next to it where the tile touches the
// frame edge.  Traffic: the frames once + the padded scales once.
struct PrepTile {
    float *p[2][5];          // [frame][scale] padded output [C][Hp][Wp]
    int r[5], Hp[5], Wp[5];
    int ns;
};
constexpr int PT = 64;
template <typename T>
__global__ __launch_bounds__(256) void prep_tiles_kernel(const T *__restrict__ I0, const T *__restrict__ I1, int C, int H, int W, int pl, int pt,
                                                        PrepTile ps, float sc) {
#pragma clang fp contract(off)
    __shared__ float tile[PT][PT + 1];
    const int f = blockIdx.z / C, c = blockIdx.z - f * C;
    const T *__restrict__ img = (f ? I1 : I0) + (long long)c * H * W;
    const int x0 = blockIdx.x * PT, y0 = blockIdx.y * PT;
    const int tw = min(PT, W - x0), th = min(PT, H - y0);
    {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                    // 16 rows of 64 per pass: all loads in flight before the LDS writes
            const int e = i * 256 + threadIdx.x, yy = e >> 6, xx = e & 63;
            v[i] = px_ld(img, (long long)min(y0 + yy, H - 1) * W + min(x0 + xx, W - 1), sc);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = i * 256 + threadIdx.x;
            tile[e >> 6][e & 63] = v[i];
        }
    }
    __syncthreads();
    const bool left = x0 == 0, top = y0 == 0, right = x0 + PT >= W, bottom = y0 + PT >= H;
    for (int s = 0; s < ps.ns; ++s) {
        const int r = ps.r[s], Hp = ps.Hp[s], Wp = ps.Wp[s];
        const int Hs = H / r, Ws = W / r;
        // this tile's part of the padded plane: its own outputs, widened to the plane's border where the tile touches the frame edge
        const int xlo = left ? 0 : x0 / r + pl, xhi = right ? Wp : (x0 + tw) / r + pl;
        const int ylo = top ? 0 : y0 / r + pt, yhi = bottom ? Hp : (y0 + th) / r + pt;
        const int nx = xhi - xlo, n = nx * (yhi - ylo);
        float *__restrict__ out = ps.p[f][s] + (long long)c * Hp * Wp;
        const float inv = 1.0f / (float)(r * r);
        for (int e = threadIdx.x; e < n; e += 256) {
            const int yy = e / nx, y = ylo + yy, x = xlo + (e - yy * nx);
            const int sy = y - pt, sx = x - pl;
            float v = 0.f;
            if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
                const int ty = sy * r - y0, tx = sx * r - x0;
                float acc = 0.f;
                if (r == 1) acc = 0.f + tile[ty][tx];
                else {
                    for (int i = 0; i < r; ++i)
                        for (int j = 0; j < r; ++j) acc = acc + tile[ty + i][tx + j];
                }
                v = r > 1 ? acc * inv : acc;
            }
            out[(long long)y * Wp + x] = v;
        }
    }
}

// the same per FRAME, each entry with its own padding (the learned-filter matcher crops frame 0 by the search window before
// it filters: its padded frame is smaller than frame 1's): blockIdx.y = entry
struct PrepFrames {
    const float *img[2 * DFE_MAX_RATIOS];
    float *out[2 * DFE_MAX_RATIOS];
    int r[2 * DFE_MAX_RATIOS], pl[2 * DFE_MAX_RATIOS], pt[2 * DFE_MAX_RATIOS], Hp[2 * DFE_MAX_RATIOS], Wp[2 * DFE_MAX_RATIOS];
};
__global__ void prep_frames_kernel(int C, int H, int W, PrepFrames pf) {
#pragma clang fp contract(off)
    const int z = blockIdx.y;
    const int r = pf.r[z], pl = pf.pl[z], pt = pf.pt[z], Hp = pf.Hp[z], Wp = pf.Wp[z];
    const float *__restrict__ img = pf.img[z];
    float *__restrict__ out = pf.out[z];
    const int Hs = H / r, Ws = W / r;
    const unsigned total = (unsigned)C * Hp * Wp;
    const float inv = 1.0f / (float)(r * r);
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned t = e / (unsigned)Wp;
        const int x = (int)(e - t * (unsigned)Wp);
        const int c = (int)(t / (unsigned)Hp), y = (int)(t - (unsigned)c * Hp);
        const int sy = y - pt, sx = x - pl;
        float v = 0.f;
        if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
            const float *src = img + ((long long)c * H + sy * r) * W + sx * r;
            float s = 0.f;
            if (r == 1) s = 0.f + src[0];
            else if (r == 2) s = box_sum<2>(src, W);
            else if (r == 4) s = box_sum<4>(src, W);
            else if (r == 8) s = box_sum<8>(src, W);
            else {
                for (int i = 0; i < r; ++i)
                    for (int j = 0; j < r; ++j) s = s + src[(long long)i * W + j];
            }
            v = r > 1 ? s * inv : s;
        }
        out[e] = v;
    }
}

// ---- A3: p = softmax(-cost) over the N cells of each pixel (one wave per pixel) --------------------
__device__ __forceinline__ void softmin_body(const float *__restrict__ cost, long long P, int N, float *__restrict__ prob) {
    const int lane = threadIdx.x & 63;
    if (N <= 64) {
        // one cell per lane: NPX pixels per step, all loads issued before the first is used, reductions on the VALU (the
        // shuffle version is LDS-crossbar- and latency-bound: 62 us for the 157 MB of the VGA scale-1 volume)
        constexpr int NPX = 8;
        const bool on = lane < N;
        for (long long p0 = ((long long)blockIdx.x * kWaves + (threadIdx.x >> 6)) * NPX; p0 < P; p0 += (long long)gridDim.x * kWaves * NPX) {
            float c[NPX];
#pragma unroll
            for (int i = 0; i < NPX; ++i) c[i] = (on && p0 + i < P) ? cost[(p0 + i) * N + lane] : 0.f;
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                if (p0 + i >= P) break;                                   // wave-uniform
                const float m = wave_max_f32(on ? -c[i] : -INFINITY);
                const float e = on ? dfe_exp_nonpos(-c[i] - m) : 0.f;
                const float sum = wave_sum_f32_ordered(e);
                if (on) prob[(p0 + i) * N + lane] = e * (1.0f / sum);
            }
        }
        return;
    }
    // windows of more than 64 cells (the single-scale models: 16 x 16, 17 x 17, 33 x 33): SIXTEEN lanes per pixel, lane t of them takes
    // the cells 64 j + 4 t + i (i < 4) -- 256 contiguous bytes per 16 lanes and step -- and sums their exponentials in that order; the
    // 16 partial sums meet over partners at distance 8, 4, 2, 1.  The feature matcher's soft-max epilogue (feat_matching_flat.hip,
    // FF_SOFT) reads a pixel's window from LDS in the same pattern: the same association, the same bits.
    const int t = lane & 15;
    for (long long p = ((long long)blockIdx.x * kWaves + (threadIdx.x >> 6)) * 4 + (lane >> 4); p < P; p += (long long)gridDim.x * kWaves * 4) {
        const float *c = cost + p * N;
        float *o = prob + p * N;
        float m = -INFINITY;
        for (int n0 = 4 * t; n0 < N; n0 += 64)
            for (int i = 0; i < 4 && n0 + i < N; ++i) m = fmaxf(m, -c[n0 + i]);
        m = row16_max_f32(m);
        float s = 0.f;
        for (int n0 = 4 * t; n0 < N; n0 += 64)
            for (int i = 0; i < 4 && n0 + i < N; ++i) {
                const float e = dfe_exp_nonpos(-c[n0 + i] - m);
                o[n0 + i] = e;
                s = s + e;
            }
        s = row16_sum_f32_ordered(s);
        const float inv = 1.0f / s;
        for (int n0 = 4 * t; n0 < N; n0 += 64)
            for (int i = 0; i < 4 && n0 + i < N; ++i) o[n0 + i] *= inv;
    }
}

__global__ __launch_bounds__(kWaves * 64) void softmin_kernel(const float *__restrict__ cost, long long P, int N,
                                                             float *__restrict__ prob) {
    softmin_body(cost, P, N, prob);
}
// the soft-min of every scale's volume in one launch: blockIdx.y = scale
struct SoftScales {
    const float *cost[DFE_MAX_RATIOS];
    float *prob[DFE_MAX_RATIOS];
    long long P[DFE_MAX_RATIOS];
};
__global__ __launch_bounds__(kWaves * 64) void softmin_scales_kernel(SoftScales ss, int N) {
    const int s = blockIdx.y;
    softmin_body(ss.cost[s], ss.P[s], N, ss.prob[s]);
}

struct CascadeGeom {
    int nratios, maxh, maxw, H, W;
    int ratios[DFE_MAX_RATIOS];
    int d[DFE_MAX_RATIOS];      // ring width of scale s >= 1
    int base[DFE_MAX_RATIOS];   // 0-based class offset of scale s in the joined vector
    int ncls;
    const float *in[DFE_MAX_RATIOS];
    float *out_scale[DFE_MAX_RATIOS];   // A4-only mode: per-scale outputs [P][maxh][maxw]
};

// out_s = in_s + replicate_q(crop(out_{s+1}, dh, dw)), coarse -> fine, in one wave's LDS window buffers.
// CascadingAddTable.lua:117-132: dh = maxh*(r2-r)/(2*r2), q = r2/r.
template <bool RING>
__global__ __launch_bounds__(kWaves * 64) void cascade_kernel(CascadeGeom g, float *__restrict__ out) {
    extern __shared__ float sh[];
    const int N = g.maxh * g.maxw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *cur = sh + (size_t)w * 2 * N, *prev = cur + N;
    const long long P = (long long)g.H * g.W;
    for (long long p = (long long)blockIdx.x * kWaves + w; p < P; p += (long long)gridDim.x * kWaves) {
        const int y = (int)(p / g.W), x = (int)(p - (long long)y * g.W);
        for (int s = g.nratios - 1; s >= 0; --s) {
            const int r = g.ratios[s];
            // RING: inputs at native scale, nearest-neighbour upsampling = index (y/r, x/r); else full-res inputs
            const float *src = RING ? g.in[s] + ((long long)(y / r) * (g.W / r) + x / r) * N : g.in[s] + p * N;
            if (s == g.nratios - 1) {
                for (int n = lane; n < N; n += 64) cur[n] = src[n];
            } else {
                const int r2 = g.ratios[s + 1], q = r2 / r;
                const int dh = g.maxh * (r2 - r) / (2 * r2), dw = g.maxw * (r2 - r) / (2 * r2);
                for (int n = lane; n < N; n += 64) {
                    int a = n / g.maxw, b = n - a * g.maxw;
                    cur[n] = src[n] + prev[(dh + a / q) * g.maxw + dw + b / q];
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            if (RING) {
                float *o = out + p * g.ncls + g.base[s];
                if (s == 0) {
                    for (int n = lane; n < N; n += 64) o[n] = cur[n];
                } else {   // ring blocks in the order top, left, right, bottom  (opticalflow_model_multiscale.lua:301-315)
                    const int d = g.d[s], mh = g.maxh, mw = g.maxw;
                    for (int n = lane; n < N; n += 64) {
                        int a = n / mw, b = n - a * mw, idx = -1;
                        if (a < d) idx = a * mw + b;
                        else if (a >= mh - d) idx = d * mw + 2 * (mh - 2 * d) * d + (a - (mh - d)) * mw + b;
                        else if (b < d) idx = d * mw + (a - d) * d + b;
                        else if (b >= mw - d) idx = d * mw + (mh - 2 * d) * d + (a - d) * d + (b - (mw - d));
                        if (idx >= 0) o[idx] = cur[n];
                    }
                }
            } else {
                float *o = g.out_scale[s] + p * N;
                for (int n = lane; n < N; n += 64) o[n] = cur[n];
            }
            float *t = cur; cur = prev; prev = t;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// A4b: gradient of the cascade w.r.t. its inputs, fine -> coarse: g_0 = go_0, g_{s+1} = go_{s+1} + pad(blocksum_q(g_s)).
// CascadingAddTable.lua:137-154.  One wave per pixel; g_s sits in LDS while the next scale gathers its q x q blocks
// (row-major float accumulation, as in the CPU restatement).
__global__ __launch_bounds__(kWaves * 64) void cascade_backward_kernel(CascadeGeom g) {
    extern __shared__ float sh[];
    const int N = g.maxh * g.maxw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *cur = sh + (size_t)w * 2 * N, *prev = cur + N;
    const long long P = (long long)g.H * g.W;
    for (long long p = (long long)blockIdx.x * kWaves + w; p < P; p += (long long)gridDim.x * kWaves) {
        for (int s = 0; s < g.nratios; ++s) {
            const float *go = g.in[s] + p * N;
            if (s == 0) {
                for (int n = lane; n < N; n += 64) cur[n] = go[n];
            } else {
                const int r = g.ratios[s - 1], r2 = g.ratios[s], q = r2 / r;
                const int dh = g.maxh * (r2 - r) / (2 * r2), dw = g.maxw * (r2 - r) / (2 * r2);
                for (int n = lane; n < N; n += 64) {
                    const int a = n / g.maxw - dh, b = n - (n / g.maxw) * g.maxw - dw;
                    float v = go[n];
                    if (a >= 0 && a < g.maxh / q && b >= 0 && b < g.maxw / q) {
                        float acc = 0.f;
                        for (int u = 0; u < q; ++u)
                            for (int t = 0; t < q; ++t) acc += prev[(a * q + u) * g.maxw + b * q + t];
                        v += acc;
                    }
                    cur[n] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            float *o = g.out_scale[s] + p * N;
            for (int n = lane; n < N; n += 64) o[n] = cur[n];
            float *t = cur; cur = prev; prev = t;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// launch shape of cascade_argmax_kernel's one-cell-per-lane path: a wave takes 8 adjacent pixels per step, a block of kWaves
// waves one row; 2 steps per wave while that keeps the grid under ~8k blocks, else 4 (measured: VGA 1 / 2 / 4 steps ->
// 0.1507 / 0.142 / 0.1456 ms per pair, 1080p 2 / 4 steps -> 0.973 / 0.933 ms; blocks that walk down several rows
// instead measured slower)
static dim3 cascade_fast_grid(int H, int W) {
    int spw = 2;
    if ((long long)((W + kWaves * 8 * spw - 1) / (kWaves * 8 * spw)) * H > 8192) spw = 4;
    const int gx = (W + kWaves * 8 * spw - 1) / (kWaves * 8 * spw);
    return dim3((unsigned)gx, (unsigned)H);
}

// class id -> decoded displacement, (oy << 16) | (ox & 0xffff), for the one-cell-per-lane path (at most 5 scales of <= 64 cells)
constexpr int kMaxDecode = 5 * 64;
struct DecodeTab {
    int v[kMaxDecode];
    int pow2;                     // ratios are exactly 1, 2, 4, ... (the coarse pixels under 8 aligned fine pixels are shared)
    unsigned short cell[5][64];   // per (scale, lane = cell): (class in the joined vector + 1) << 6 | lane it reads the coarser window from
};
// What depends on the cell only: where it reads the coarser window (crop + replicate, CascadingAddTable.lua:117-132) and which
// class it is in the joined vector (ring blocks top, left, right, bottom: opticalflow_model_multiscale.lua:301-315; -1 =
// inside the ring hole).
static void fill_cell_maps(const CascadeGeom &g, DecodeTab &dt) {
    const int N = g.maxh * g.maxw;
    dt.pow2 = 1;
    for (int s = 0; s < g.nratios; ++s)
        if (g.ratios[s] != (1 << s)) dt.pow2 = 0;
    for (int s = 0; s < 5; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            int gsrc = 0, c = -1;
            if (s < g.nratios && lane < N) {
                const int a = lane / g.maxw, b = lane - a * g.maxw, d = g.d[s], mh = g.maxh, mw = g.maxw;
                if (s < g.nratios - 1) {
                    const int r = g.ratios[s], r2 = g.ratios[s + 1], q = r2 / r;
                    const int dh = mh * (r2 - r) / (2 * r2), dw = mw * (r2 - r) / (2 * r2);
                    gsrc = (dh + a / q) * mw + dw + b / q;
                }
                if (s == 0) c = lane;
                else if (a < d) c = a * mw + b;
                else if (a >= mh - d) c = d * mw + 2 * (mh - 2 * d) * d + (a - (mh - d)) * mw + b;
                else if (b < d) c = d * mw + (a - d) * d + b;
                else if (b >= mw - d) c = d * mw + (mh - 2 * d) * d + (a - d) * d + (b - (mw - d));
                if (c >= 0) c += g.base[s];
            }
            dt.cell[s][lane] = (unsigned short)(((c + 1) << 6) | (gsrc & 0x3f));   // c + 1 <= 320, gsrc <= 63
        }
}
static void fill_decode_tab(const MultiGeom &mg, int ncls, DecodeTab &dt) {
    for (int c = 0; c < ncls && c < kMaxDecode; ++c) {
        long long oy = 0, ox = 0;
        multi_decode(mg, c + 1, &oy, &ox);
        dt.v[c] = (int)((unsigned)((int)oy << 16) | ((unsigned)(int)ox & 0xffffu));
    }
}

// One cascade step for ratios 1, 2, 4, ... and 8 aligned pixels x0 .. x0+7 inside the row: the coarse pixel of scale s under
// fine pixel i is (x0 >> s) + (i >> s), so scale s has only max(1, 8 >> s) distinct inputs -- each is loaded (immediate
// offsets from one address per scale), cascaded and compared ONCE and shared by the fine pixels below it.  Same adds and
// comparisons per fine pixel, in the same order, as the generic loop: bit-identical.
template <int NR, bool SOFT0>
__device__ __forceinline__ void cascade_inputs_pow2(const CascadeGeom &g, const int (&gsrc)[5], const int (&cls)[5], int lane, int N, int y, int x0,
                                                    int (&key)[8], int (&bcls)[8], int (&cbits)[8]) {
    float pv[8], bv[8];
    int bi[8];
    static_for_ms<0, NR>([&](auto sc) {
        constexpr int s = NR - 1 - decltype(sc)::value;          // coarse -> fine
        constexpr int cnt = (8 >> s) > 0 ? (8 >> s) : 1;
        const float *src = g.in[s] + ((long long)(y >> s) * (g.W >> s) + (x0 >> s)) * N + lane;
        float v[cnt];
#pragma unroll
        for (int j = 0; j < cnt; ++j) v[j] = lane < N ? src[j * N] : 0.f;
        if constexpr (SOFT0 && s == 0) {
            const bool on = lane < N;
#pragma unroll
            for (int j = 0; j < cnt; ++j) {
                const float c = v[j];
                const float m = wave_max_f32(on ? -c : -INFINITY);
                const float e = on ? dfe_exp_nonpos(-c - m) : 0.f;
                const float sum = wave_sum_f32_ordered(e);
                v[j] = on ? e * (1.0f / sum) : 0.f;
            }
        }
        // children first (descending j): slot j of this scale reads the parent slot j >> 1 of the coarser scale before it is overwritten
#pragma unroll
        for (int j = cnt - 1; j >= 0; --j) {
            float val = v[j];
            float pbv = -INFINITY;
            int pbi = 0x7fffffff;
            if constexpr (s < NR - 1) {
                constexpr int pcnt = (8 >> (s + 1)) > 0 ? (8 >> (s + 1)) : 1;
                const int pj = (cnt == 2 * pcnt) ? (j >> 1) : j;   // (cnt == pcnt == 1 above ratio 8)
                val += __int_as_float(__builtin_amdgcn_ds_bpermute(gsrc[s], __float_as_int(pv[pj])));
                pbv = bv[pj];
                pbi = bi[pj];
            }
            const int c = cls[s];
            if (c >= 0 && (val > pbv || (val == pbv && c < pbi))) { pbv = val; pbi = c; }
            pv[j] = val; bv[j] = pbv; bi[j] = pbi;
        }
    });
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int bb = __float_as_int(bv[i] + 0.0f);
        key[i] = ~(bb >= 0 ? bb : bb ^ 0x7fffffff);
        bcls[i] = bi[i];
        cbits[i] = __float_as_int(pv[i]);
    }
}

// SOFT0 (one-cell-per-lane path only): g.in[0] is the RAW scale-1 cost volume and its soft-min is taken here, with the
// arithmetic of softmin_kernel (wave maximum, expf, wave sum in the same association order, e * (1 / sum)), so the scale-1
// probabilities -- the largest tensor of the pipeline -- never make their round trip through HBM.
template <bool SOFT0>
__global__ __launch_bounds__(kWaves * 64) void cascade_argmax_kernel(CascadeGeom g, MultiGeom mg, int middle, long long *__restrict__ idx,
                                                                     float *__restrict__ best_out, float *__restrict__ fy,
                                                                     float *__restrict__ fx, int pitch, int pad_t, int pad_l, DecodeTab dt) {
    extern __shared__ float sh[];
    const int N = g.maxh * g.maxw;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *cur = sh + (size_t)w * 2 * N, *prev = cur + N;
    const long long P = (long long)g.H * g.W;
    if (N <= 64 && g.nratios <= 5) {
        // One cell per lane: everything that depends on the cell only -- where it reads the coarser window (crop +
        // replicate) and which class it is in the joined vector -- is worked out once per wave; per pixel and scale
        // remain one coalesced load, one cross-lane read of the coarser result (no LDS buffer, no barrier), one add and
        // the running arg-max.
        // (both maps come from the host, packed per (scale, lane) in the kernel arguments: fill_decode_tab)
        int gsrc[5], cls[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int pk = dt.cell[s][lane];                   // (cls + 1) << 6 | gsrc
            gsrc[s] = (pk & 0x3f) << 2;                        // byte address for ds_bpermute
            cls[s] = (pk >> 6) - 1;
        }
        // 2-D launch on this path: blockIdx.y = row, a wave takes NPX adjacent pixels per step and issues all of their
        // loads before it touches any (one pixel at a time the kernel was bound by load latency: 154 us at VGA); all
        // index arithmetic in 32 bits (a 64-bit divide per pixel cost more than the whole cascade), x / r through a float
        // reciprocal (exact: (x + 0.5) / r is never within 1/(2r) of an integer); class -> displacement from a table
        // built once per block instead of divisions per pixel.
        // Eight pixels per step: their wave reductions share one halving butterfly (dfe_internal.h: wave_min8, ~40 VALU for 8
        // columns instead of 8 x 12 dependent DPP steps), on keys that order like the floats (sign-folded bits, inverted so
        // that the minimum is the maximum value); the smallest class among the lanes holding a column's maximum comes
        // from a second butterfly; lanes 0..7 then finish one pixel each and store together.
        constexpr int NPX = 8, MAXS = 5;
        // class -> displacement: a table the host built once per launch, passed by value (kernel arguments live in constant
        // memory): no per-block rebuild, no barrier on this path
        const int mlane = (middle - 1) & 63;
        const int y = blockIdx.y;
        const float *rowp[MAXS];
        float rinv[MAXS];
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int r = s < g.nratios ? g.ratios[s] : 1;
            rowp[s] = g.in[s < g.nratios ? s : 0] + (long long)(y / r) * (g.W / r) * N;
            rinv[s] = 1.0f / (float)r;
        }
        for (int x0 = (blockIdx.x * kWaves + w) * NPX; x0 < g.W; x0 += gridDim.x * kWaves * NPX) {
            int key[NPX], bcls[NPX], cbits[NPX];
            if (dt.pow2 && x0 + NPX <= g.W && g.nratios >= 2) {   // wave-uniform
                if (g.nratios == 2) cascade_inputs_pow2<2, SOFT0>(g, gsrc, cls, lane, N, y, x0, key, bcls, cbits);
                else if (g.nratios == 3) cascade_inputs_pow2<3, SOFT0>(g, gsrc, cls, lane, N, y, x0, key, bcls, cbits);
                else if (g.nratios == 4) cascade_inputs_pow2<4, SOFT0>(g, gsrc, cls, lane, N, y, x0, key, bcls, cbits);
                else cascade_inputs_pow2<5, SOFT0>(g, gsrc, cls, lane, N, y, x0, key, bcls, cbits);
            } else {
            float vin[NPX][MAXS];
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                const int x = min(x0 + i, g.W - 1);
#pragma unroll
                for (int s = 0; s < MAXS; ++s) {
                    const int xs = (int)(((float)x + 0.5f) * rinv[s]);
                    vin[i][s] = (s < g.nratios && lane < N) ? rowp[s][xs * N + lane] : 0.f;
                }
            }
            if constexpr (SOFT0) {
                const bool on = lane < N;
#pragma unroll
                for (int i = 0; i < NPX; ++i) {
                    const float c = vin[i][0];
                    const float m = wave_max_f32(on ? -c : -INFINITY);
                    const float e = on ? dfe_exp_nonpos(-c - m) : 0.f;
                    const float sum = wave_sum_f32_ordered(e);
                    vin[i][0] = on ? e * (1.0f / sum) : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < NPX; ++i) {
                float bv = -INFINITY, pv = 0.f;
                int bi = 0x7fffffff;
#pragma unroll
                for (int s = MAXS - 1; s >= 0; --s) {
                    if (s >= g.nratios) continue;
                    float v = vin[i][s];
                    if (s < g.nratios - 1) v += __int_as_float(__builtin_amdgcn_ds_bpermute(gsrc[s], __float_as_int(pv)));
                    pv = v;
                    const int c = cls[s];
                    if (c >= 0 && (v > bv || (v == bv && c < bi))) { bv = v; bi = c; }
                }
                // order-preserving int image of the float (-0 folded onto +0 first, so that equal floats give equal keys),
                // inverted: the smallest key is the largest value
                const int bb = __float_as_int(bv + 0.0f);
                key[i] = ~(bb >= 0 ? bb : bb ^ 0x7fffffff);
                bcls[i] = bi;
                cbits[i] = __float_as_int(pv);                // scale-1 value of this lane's cell (the centre class sits in lane mlane)
            }
            }
            const int wk = wave_min8<NPX>(key, lane);         // lane L: key of the maximum of pixel L & 7
            int cand[NPX];
#pragma unroll
            for (int i = 0; i < NPX; ++i) cand[i] = key[i] == __builtin_amdgcn_readlane(wk, i) ? bcls[i] : 0x7fffffff;
            const int wc = wave_min8<NPX>(cand, lane);        // lane L: smallest class holding that maximum
            int cen = 0;                                      // lane i: the centre class's value of pixel i
#define DFE_CEN(i) { const int t = __builtin_amdgcn_readlane(cbits[i], mlane); asm("v_writelane_b32 %0, %1, " #i : "+v"(cen) : "s"(t)); }
            DFE_CEN(0) DFE_CEN(1) DFE_CEN(2) DFE_CEN(3) DFE_CEN(4) DFE_CEN(5) DFE_CEN(6) DFE_CEN(7)
#undef DFE_CEN
            const int x = x0 + lane;
            if (lane < NPX && x < g.W) {
                const int t = ~wk;
                const float wmax = __int_as_float(t >= 0 ? t : t ^ 0x7fffffff);
                int id = wc + 1;
                if (middle > 0 && wmax == __int_as_float(cen)) id = middle;
                const long long p = (long long)y * g.W + x;
                if (idx) idx[p] = id;
                if (best_out) best_out[p] = wmax;
                if (fy) {
                    const int d = dt.v[id - 1];                         // (oy << 16) | (ox & 0xffff)
                    const long long fo = (long long)(y + pad_t) * pitch + x + pad_l;
                    fy[fo] = (float)(d >> 16);
                    fx[fo] = (float)(short)(d & 0xffff);
                }
            }
        }
        return;
    }
    for (long long p = (long long)blockIdx.x * kWaves + w; p < P; p += (long long)gridDim.x * kWaves) {
        const int y = (int)(p / g.W), x = (int)(p - (long long)y * g.W);
        float bv = -INFINITY;
        int bi = 0x7fffffff;           // 0-based class id
        float centre = 0.f;
        for (int s = g.nratios - 1; s >= 0; --s) {
            const int r = g.ratios[s];
            const float *src = g.in[s] + ((long long)(y / r) * (g.W / r) + x / r) * N;
            const int d = g.d[s], mh = g.maxh, mw = g.maxw;
            int r2 = 1, q = 1, dh = 0, dw = 0;
            if (s < g.nratios - 1) {
                r2 = g.ratios[s + 1]; q = r2 / r;
                dh = mh * (r2 - r) / (2 * r2); dw = mw * (r2 - r) / (2 * r2);
            }
            for (int n = lane; n < N; n += 64) {
                const int a = n / mw, b = n - a * mw;
                float v = src[n];
                if (s < g.nratios - 1) v += prev[(dh + a / q) * mw + dw + b / q];
                cur[n] = v;
                int cls = -1;          // class of this cell in the joined vector, -1 = not emitted (inside the ring hole)
                if (s == 0) cls = n;
                else if (a < d) cls = a * mw + b;
                else if (a >= mh - d) cls = d * mw + 2 * (mh - 2 * d) * d + (a - (mh - d)) * mw + b;
                else if (b < d) cls = d * mw + (a - d) * d + b;
                else if (b >= mw - d) cls = d * mw + (mh - 2 * d) * d + (a - d) * d + (b - (mw - d));
                if (cls >= 0) {
                    cls += g.base[s];
                    if (v > bv || (v == bv && cls < bi)) { bv = v; bi = cls; }
                    if (cls == middle - 1) centre = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            float *t = cur; cur = prev; prev = t;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            centre += __shfl_xor(centre, off);       // exactly one lane holds it, the others 0
        }
        if (lane == 0) {
            long long id = (long long)bi + 1;
            if (middle > 0 && bv == centre) id = middle;
            if (idx) idx[p] = id;
            if (best_out) best_out[p] = bv;
            if (fy) {
                long long oy = 0, ox = 0;
                multi_decode(mg, id, &oy, &ox);
                const long long fo = (long long)(y + pad_t) * pitch + x + pad_l;
                fy[fo] = (float)oy;
                fx[fo] = (float)ox;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- lane <-> PIXEL cascade for 8 x 8 windows and ratios 1, 2, 4, ... (the configurations of BASELINE configs[1], [3], [4]) ------------
// cascade_argmax_kernel above keeps a window's 64 cells on the 64 lanes of a wave: every soft-min needs two wave
// reductions, every cascade add a cross-lane read, the arg-max two 8-column butterflies -- ~50 VALU per pixel and latency
// chains nothing overlaps (63.9 us at VGA, 11 % of the step's roofline).  Here a lane owns a whole pixel: its 64 costs sit in
// registers, the maximum, the exponentials, the sum (added in the SAME tree as the wave reduction: partners at distance 32,
// 16, 8, 4, 2, 1 -- bit-identical) and the arg-max are straight-line code with no cross-lane traffic at all (~17 VALU per
// pixel).  One launch per scale, coarse to fine: scale s leaves its cascaded window [P_s][64] and its running best (value,
// class) per pixel for scale s-1, whose pixel (y, x) reads the 4 x 4 cells (2 + a/2, 2 + b/2) of pixel (y/2, x/2)
// (CascadingAddTable.lua:117-132 with maxh = 8: dh = dw = 2, q = 2); scale 1 writes class ids / flow only.  The same float
// operations in the same order as softmin_kernel + cascade_ring + arg-max: bit-identical to the staged path.
struct CascadePxArgs {
    const float *cost;        // [Hs][Ws][64] raw SSD costs of this scale
    const float *pcasc;       // cascaded windows of the coarser scale [Hs/2][Ws/2][64], or NULL (coarsest)
    const float2 *pbest;      // its running best (value, class as int bits), or NULL
    float *casc;              // out (scales > 1): this scale's cascaded windows
    float2 *best;             // out (scales > 1)
    long long *idx;           // out (scale 1): 1-based class ids, or NULL
    float *fy, *fx;           // out (scale 1): decoded flow planes, or NULL
    int Hs, Ws, scale, middle;
    int cls_base;             // scales > 1: 0-based class id of the ring's first cell (g.base[scale])
    const float *pcost;       // INLINE: raw costs of the (coarsest) parent scale
    int pcls_base;            // INLINE: its class base
    float inv_scale;          // H16: cost and pcost are half volumes holding half(cost * scale); inv_scale = 1 / scale
};

// soft-min of the 64 costs in v (in place): p = e / sum, e = expf(-c - max(-c)), the sum in the association order of
// wave_sum_f32_ordered (partners at distance 32, 16, 8, 4, 2, 1)
__device__ __forceinline__ void px_softmin64(float (&v)[64]) {
#pragma clang fp contract(off)
    float m = -v[0];
#pragma unroll
    for (int j = 1; j < 64; ++j) m = fmaxf(m, -v[j]);
    float t[32];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = dfe_exp_nonpos(-v[j] - m);
#pragma unroll
    for (int j = 0; j < 32; ++j) t[j] = v[j] + v[j + 32];
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1)
#pragma unroll
        for (int j = 0; j < h; ++j) t[j] = t[j] + t[j + h];
    const float rs = 1.0f / t[0];
#pragma unroll
    for (int j = 0; j < 64; ++j) v[j] = v[j] * rs;
}
__device__ __forceinline__ void px_load64(const float *__restrict__ src_, float (&v)[64]) {
    const float4 *src = reinterpret_cast<const float4 *>(src_);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float4 t = src[j];
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
}
// The same load for the 64 pixels of a wave, coalesced: a lane's own 256 B sit 256 B from its neighbour's, so the plain form
// touches 64 cache lines with every instruction.  Here the wave reads its 16 KB front to back (eight lanes per 128-B line),
// parks them in an LDS scratch of 9 float4 per pixel and half window (one float4 of padding: conflict-free on the way out),
// and every lane picks up its own pixel.  No barrier: the scratch belongs to the wave, and a wave's LDS operations execute
// in order.  `first`: the wave's first pixel, `P`: pixels in the plane (lanes past the end still help loading).
__device__ __forceinline__ void px_load64_staged(const float *__restrict__ plane, long long first, long long P, int lane, float4 *scr, float (&v)[64]) {
    const float4 *src = reinterpret_cast<const float4 *>(plane) + first * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 t[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int g = jj * 64 + lane, q = g >> 3, w = g & 7;
            t[jj] = first + q < P ? src[q * 16 + h * 8 + w] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int g = jj * 64 + lane, q = g >> 3, w = g & 7;
            scr[q * 9 + w] = t[jj];
        }
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float4 u = scr[lane * 9 + w];
            v[h * 32 + 4 * w] = u.x; v[h * 32 + 4 * w + 1] = u.y; v[h * 32 + 4 * w + 2] = u.z; v[h * 32 + 4 * w + 3] = u.w;
        }
    }
}
// fp16 volumes (dfe_multiscale_flow_pair_f16): the stored value is half(cost * scale); the cascade works on
// float(stored) * inv_scale in fp32 registers, exactly as if the fp32 volume had been rounded to half precision in place.
__device__ __forceinline__ void px_unpack8h(const uint4 &t, float inv, float *v) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    union { uint4 u; h2_t h[4]; } c;
    c.u = t;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = (float)c.h[i][0] * inv; v[2 * i + 1] = (float)c.h[i][1] * inv; }
}
__device__ __forceinline__ void px_load64_h(const void *__restrict__ src_, float inv, float (&v)[64]) {
    const uint4 *src = reinterpret_cast<const uint4 *>(src_);
#pragma unroll
    for (int j = 0; j < 8; ++j) px_unpack8h(src[j], inv, &v[8 * j]);
}
// the wave's 64 windows = 8 KB of halves, front to back (eight lanes per 128-B window), through the same LDS scratch
__device__ __forceinline__ void px_load64_staged_h(const void *__restrict__ plane, long long first, long long P, int lane, float4 *scr_, float inv,
                                                   float (&v)[64]) {
    const uint4 *src = reinterpret_cast<const uint4 *>(plane) + first * 8;
    uint4 *scr = reinterpret_cast<uint4 *>(scr_);
    uint4 t[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int g = jj * 64 + lane, q = g >> 3, w = g & 7;
        t[jj] = first + q < P ? src[q * 8 + w] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int g = jj * 64 + lane, q = g >> 3, w = g & 7;
        scr[q * 9 + w] = t[jj];
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) px_unpack8h(scr[lane * 9 + w], inv, &v[8 * w]);
}
// ... and the way back (the cascaded window a coarser scale leaves for its child)
__device__ __forceinline__ void px_store64_staged(float *__restrict__ plane, long long first, long long P, int lane, float4 *scr, const float (&v)[64]) {
    float4 *dst = reinterpret_cast<float4 *>(plane) + first * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int w = 0; w < 8; ++w) scr[lane * 9 + w] = make_float4(v[h * 32 + 4 * w], v[h * 32 + 4 * w + 1], v[h * 32 + 4 * w + 2], v[h * 32 + 4 * w + 3]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int g = jj * 64 + lane, q = g >> 3, w = g & 7;
            const float4 u = scr[q * 9 + w];
            if (first + q < P) dst[q * 16 + h * 8 + w] = u;
        }
    }
}
// Arg-max over a ring scale's 48 classes (see the comment in the kernel): (value, 0-based class)
__device__ __forceinline__ void px_ring_best(const float (&v)[64], int cls_base, float &fv, int &fi) {
    constexpr int ORD[48] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15,      // top d x maxw
                             16, 17, 24, 25, 32, 33, 40, 41,                                      // left (maxh-2d) x d
                             22, 23, 30, 31, 38, 39, 46, 47,                                      // right
                             48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63};     // bottom
    fv = v[ORD[0]];
    fi = 0;
#pragma unroll
    for (int k = 1; k < 48; ++k) fv = fmaxf(fv, v[ORD[k]]);
#pragma unroll
    for (int k = 47; k >= 0; --k) fi = v[ORD[k]] == fv ? k : fi;
    fi += cls_base;
}

// INLINE: the coarser scale is the coarsest one and is not launched at all: every lane recomputes its parent pixel's window
// from the raw costs (a.pcost) with exactly the parent kernel's operations -- four lanes repeat the same ~1300 instructions,
// which is cheaper than a launch of its own for a scale of a few hundred waves (VGA scale 4: 75 blocks, 11 us of latency).
template <bool FINEST, bool INLINE, bool H16 = false>
__global__ __launch_bounds__(256) void cascade_px_kernel(CascadePxArgs a, DecodeTab dt) {
#pragma clang fp contract(off)
    __shared__ float4 stage[4][64 * 9];
    const long long P = (long long)a.Hs * a.Ws;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long wfirst = (long long)blockIdx.x * 256 + wv * 64;      // the wave's first pixel
    if (wfirst >= P) return;                                             // (wave-uniform)
    const bool live = p < P;                                             // lanes past the end help with the staged loads / stores
    const long long pc_ = live ? p : P - 1;
    const int y = (int)(pc_ / a.Ws), x = (int)(pc_ - (long long)y * a.Ws);
    float v[64];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    float par[16];
    const long long pp = (long long)(y >> 1) * (a.Ws >> 1) + (x >> 1);
    if constexpr (INLINE) {
        if constexpr (H16) px_load64_h(reinterpret_cast<const _Float16 *>(a.pcost) + pp * 64, a.inv_scale, v);
        else px_load64(a.pcost + pp * 64, v);
        px_softmin64(v);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) par[4 * r + c] = v[(2 + r) * 8 + 2 + c];
        px_ring_best(v, a.pcls_base, bv, bi);       // (the coarsest scale has no chain above it: its best is its own)
    }
    if constexpr (H16) px_load64_staged_h(a.cost, wfirst, P, lane, stage[wv], a.inv_scale, v);
    else px_load64_staged(a.cost, wfirst, P, lane, stage[wv], v);
    px_softmin64(v);
    if (INLINE || a.pcasc) {                                           // (launch-uniform)
        if constexpr (!INLINE) {
            const float *pc = a.pcasc + pp * 64;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float2 lo = *reinterpret_cast<const float2 *>(pc + (2 + r) * 8 + 2), hi = *reinterpret_cast<const float2 *>(pc + (2 + r) * 8 + 4);
                par[4 * r] = lo.x; par[4 * r + 1] = lo.y; par[4 * r + 2] = hi.x; par[4 * r + 3] = hi.y;
            }
            const float2 pb = a.pbest[pp];
            bv = pb.x;
            bi = __float_as_int(pb.y);
        }
#pragma unroll
        for (int j = 0; j < 64; ++j) v[j] = v[j] + par[((j >> 3) >> 1) * 4 + ((j & 7) >> 1)];
    }
    // Arg-max over this scale's classes, then against the coarser chain's best.  Rule (the cell-per-lane kernel's, i.e. TH's
    // max + the reference's class order): the largest value, among equal values the smallest class id.  This scale's ids are all
    // smaller than the coarser chain's, so: fv = max over the scale's class cells; fi = the first cell IN CLASS ORDER equal to
    // fv; the scale wins ties against the chain (fv >= bv).  Straight-line v_max / v_cmp_eq / v_cndmask with static register
    // indices -- the first version walked (value, class) pairs through a 3-way compare per cell: 400 v_cmp + 1400 scalar mask
    // instructions per pixel-lane.
    // 8 x 8 window, ring width 2 (round(8 (r - r/2) / (2 r)) = 2 for every ratio pair 2:1): class order of the ring cells =
    // top two rows, left 4 x 2, right 4 x 2, bottom two rows (opticalflow_model_multiscale.lua:301-315).
    {
        float fv;
        int fi = 0;
        if constexpr (FINEST) {
            fv = v[0];
#pragma unroll
            for (int j = 1; j < 64; ++j) fv = fmaxf(fv, v[j]);
#pragma unroll
            for (int j = 63; j >= 0; --j) fi = v[j] == fv ? j : fi;
        } else {
            px_ring_best(v, a.cls_base, fv, fi);
        }
        if (fv >= bv) { bv = fv; bi = fi; }
    }
    if constexpr (!FINEST) {
        px_store64_staged(a.casc, wfirst, P, lane, stage[wv], v);
        if (live) a.best[p] = make_float2(bv, __int_as_float(bi));
    } else {
        if (!live) return;
        int id = bi + 1;
        const int mc = (a.middle - 1) & 63;
        float cen = 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j) cen = j == mc ? v[j] : cen;      // (mc is launch-uniform: folds to one select chain)
        if (a.middle > 0 && bv == cen) id = a.middle;
        if (a.idx) a.idx[p] = id;
        if (a.fy) {
            const int d = dt.v[id - 1];
            a.fy[p] = (float)(d >> 16);
            a.fx[p] = (float)(short)(d & 0xffff);
        }
    }
}

// shapes without an fp16 volume kernel: the fp32 volume rounded to half precision in place, v = float(half(v * scale)) / scale
__global__ __launch_bounds__(256) void round_half_kernel(float *__restrict__ v, long long n, float scale, float inv) {
#pragma clang fp contract(off)
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) v[e] = (float)(_Float16)(v[e] * scale) * inv;
}

int ring_width(int maxw, int r, int rprev) { return (int)floor((double)maxw * (r - rprev) / (2.0 * r) + 0.5); }

int fill_cascade(dfe_ctx *ctx, CascadeGeom &g, const int *ratios, int nratios, int maxh, int maxw) {
    DFE_REQUIRE(ctx, ratios && nratios >= 1 && nratios <= DFE_MAX_RATIOS, DFE_E_ARG, "nratios=%d not in 1..%d", nratios, DFE_MAX_RATIOS);
    DFE_REQUIRE(ctx, maxh > 0 && maxw > 0, DFE_E_ARG, "maxh=%d maxw=%d must be positive", maxh, maxw);
    g.nratios = nratios; g.maxh = maxh; g.maxw = maxw;
    int base = 0;
    for (int i = 0; i < nratios; ++i) {
        DFE_REQUIRE(ctx, ratios[i] > 0, DFE_E_ARG, "ratios[%d]=%d", i, ratios[i]);
        g.ratios[i] = ratios[i];
        g.d[i] = i ? ring_width(maxw, ratios[i], ratios[i - 1]) : 0;
        g.base[i] = base;
        base += i ? 2 * g.d[i] * maxw + 2 * (maxh - 2 * g.d[i]) * g.d[i] : maxh * maxw;
        if (i + 1 < nratios) {   // CascadingAddTable.lua:121-124
            int r = ratios[i], r2 = ratios[i + 1];
            DFE_REQUIRE(ctx, r2 % r == 0 && (maxh * (r2 - r)) % (2 * r2) == 0 && (maxw * (r2 - r)) % (2 * r2) == 0, DFE_E_SHAPE,
                        "nn.CascadingAddTable: ratios and input sizes not compatible");
        }
    }
    g.ncls = base;
    return DFE_OK;
}

#ifndef DFE_PREP_EPT
#define DFE_PREP_EPT 8     // elements per thread of prep_scales_kernel (tuning)
#endif
int grid1d(long long n, int per_block) {
    long long b = (n + per_block - 1) / per_block;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int dfe_downsample_box_f32(dfe_ctx *ctx, const float *img, int C, int H, int W, int r, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, img && out && C > 0 && r > 0 && H >= r && W >= r, DFE_E_ARG, "dfe_downsample_box_f32: bad argument");
    hipLaunchKernelGGL(downsample_box_kernel, dim3(grid1d((long long)C * (H / r) * (W / r), 256)), dim3(256), 0, ctx->stream, img, C, H, W,
                       r, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_pyramid_scale_volume_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int r, int kh, int kw,
                                 int maxh, int maxw, float *out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1 && out, DFE_E_ARG, "dfe_pyramid_scale_volume_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && r > 0 && kh > 0 && kw > 0 && maxh > 0 && maxw > 0, DFE_E_ARG, "dfe_pyramid_scale_volume_f32: bad size");
    DFE_REQUIRE(ctx, H % r == 0 && W % r == 0, DFE_E_SHAPE,
                "dfe_pyramid_scale_volume_f32: frame %dx%d is not a multiple of ratio %d (opticalflow_model_multiscale.lua:238-243)", H, W, r);
    const int Hs = H / r, Ws = W / r;
    const int hp = maxh - 1 + kh - 1, wp = maxw - 1 + kw - 1;   // hPatch2-1 (:136-141)
    const int pt = hp / 2, pl = wp / 2;
    const int Hp = Hs + hp, Wp = Ws + wp;
    const size_t nd = (size_t)C * Hs * Ws, np = (size_t)C * Hp * Wp;
    void *scr = nullptr;
    int rc = dfe_scratch(ctx, (2 * nd + 2 * np) * sizeof(float), &scr);
    if (rc) return rc;
    float *d0 = (float *)scr, *d1 = d0 + nd, *p0 = d1 + nd, *p1 = p0 + np;
    const float *s0 = I0, *s1 = I1;
    if (r > 1) {
        hipLaunchKernelGGL(downsample_box_kernel, dim3(grid1d((long long)nd, 256)), dim3(256), 0, ctx->stream, I0, C, H, W, r, d0);
        hipLaunchKernelGGL(downsample_box_kernel, dim3(grid1d((long long)nd, 256)), dim3(256), 0, ctx->stream, I1, C, H, W, r, d1);
        s0 = d0; s1 = d1;
    }
    hipLaunchKernelGGL(zero_pad_kernel, dim3(grid1d((long long)np, 256)), dim3(256), 0, ctx->stream, s0, C, Hs, Ws, pl, pt, Hp, Wp, p0);
    hipLaunchKernelGGL(zero_pad_kernel, dim3(grid1d((long long)np, 256)), dim3(256), 0, ctx->stream, s1, C, Hs, Ws, pl, pt, Hp, Wp, p1);
    DFE_LAUNCH_CHECK(ctx);
    // frame-0 crop floor/ceil((maxw-1)/2) (:198-202) is the oy/ox offset of the cost-volume op
    return cv_frames_dispatch(ctx, p0, p1, C, Hp, Wp, (long long)Hp * Wp, kh, kw, maxh, maxw, out);
}

// f16_scale != 0: every scale's volume is stored as half(cost * f16_scale) (dfe_multiscale_flow_pair_f16)
// the learned patch filters of the matcher (NULL = raw patches, the identity filter): layers [share ? 1 : nratios][nlayers]
struct MsFilter { const dfe_filter_layer *layers; int nlayers, share; };
// u8_scale > 0: I0 / I1 point to uint8 frames (raw-patch pyramid only: the preparation kernels, the frames' only readers, convert)
static int multiscale_flow_pair(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int maxh, int maxw,
                                const int *ratios, int nratios, float *flow, int64_t *idx, float f16_scale, const MsFilter *filt = nullptr,
                                float u8_scale = 0.f) {
    DFE_REQUIRE(ctx, I0 && I1 && (flow || idx), DFE_E_ARG, "dfe_multiscale_flow_pair_f32: NULL tensor");
    DFE_REQUIRE(ctx, C > 0 && k > 0 && maxh > 0 && maxw > 0, DFE_E_ARG, "dfe_multiscale_flow_pair_f32: bad size");
    CascadeGeom g;
    int rc = fill_cascade(ctx, g, ratios, nratios, maxh, maxw);
    if (rc) return rc;
    const int N = maxh * maxw;
    // receptive field of the patch filter: k x k raw patches, or hKernel = sum kH - (nlayers - 1) of the learned stack (opticalflow.lua:154-171)
    int hk = k, wk = k, maxplanes = C;
    if (filt) {
        for (int v = 0; v < (filt->share ? 1 : nratios); ++v) {
            int h = 1, w = 1, nin = C;
            for (int l = 0; l < filt->nlayers; ++l) {
                const dfe_filter_layer &L = filt->layers[v * filt->nlayers + l];
                DFE_REQUIRE(ctx, L.weight && L.nIn > 0 && L.nOut > 0 && L.kH > 0 && L.kW > 0, DFE_E_ARG, "dfe_multiscale_flow_pair_filtered_f32: layer %d: bad description", l);
                DFE_REQUIRE(ctx, L.conn ? L.nIn <= nin : L.nIn == nin, DFE_E_SHAPE, "dfe_multiscale_flow_pair_filtered_f32: layer %d reads %d planes, the layer before it makes %d", l, L.nIn, nin);
                h += L.kH - 1; w += L.kW - 1; nin = L.nOut;
                if (L.nOut > maxplanes) maxplanes = L.nOut;
            }
            DFE_REQUIRE(ctx, v == 0 || (h == hk && w == wk), DFE_E_SHAPE, "dfe_multiscale_flow_pair_filtered_f32: the scales' filter stacks have different receptive fields");
            hk = h; wk = w;
        }
    }
    const int hp = maxh - 1 + hk - 1, wp = maxw - 1 + wk - 1;   // hPatch2-1 (opticalflow_model_multiscale.lua:136-141)
    const int pt = hp / 2, pl = wp / 2;
    size_t off_p[DFE_MAX_RATIOS], off_v[DFE_MAX_RATIOS], off_q[DFE_MAX_RATIOS], off_b[DFE_MAX_RATIOS], off_f[DFE_MAX_RATIOS], fbuf[DFE_MAX_RATIOS], total = 0;
    for (int s = 0; s < nratios; ++s) {
        const int r = ratios[s];
        DFE_REQUIRE(ctx, H % r == 0 && W % r == 0, DFE_E_SHAPE,
                    "dfe_multiscale_flow_pair_f32: frame %dx%d is not a multiple of ratio %d (opticalflow_model_multiscale.lua:238-243)", H, W, r);
        const size_t np = (size_t)C * (H / r + hp) * (W / r + wp), nv = (size_t)(H / r) * (W / r) * N;
        // learned filters: four feature buffers per scale (ping-pong per frame), each the largest layer output
        fbuf[s] = filt ? ((size_t)maxplanes * (H / r + hp) * (W / r + wp) * sizeof(float) + 255) / 256 * 256 : 0;
        off_f[s] = total; total += 4 * fbuf[s];
        off_p[s] = total; total += (2 * np * sizeof(float) + 255) / 256 * 256;
        off_v[s] = total; total += (nv * sizeof(float) + 255) / 256 * 256;
        off_q[s] = total; total += (nv * sizeof(float) + 255) / 256 * 256;
        off_b[s] = total; total += ((size_t)(H / r) * (W / r) * sizeof(float2) + 255) / 256 * 256;
    }
    // the per-scale cost volumes below use the same arena for their own temporaries only through cv_frames_dispatch, which
    // needs none; one allocation up front keeps every stage's buffers alive until the cascade has read them
    void *scr = nullptr;
    rc = dfe_scratch(ctx, total, &scr, filt != nullptr);   // (learned filters: the convolutions' arena, see dfe_scratch)
    if (rc) return rc;
    // the same call again (same buffers, shapes and arena): replay its launches as a graph
    struct { const void *I0, *I1, *flow, *idx, *scr; int C, H, W, k, maxh, maxw, nratios, ratios[DFE_MAX_RATIOS]; float f16, u8; } gkey;
    memset(&gkey, 0, sizeof gkey);
    gkey.I0 = I0; gkey.I1 = I1; gkey.flow = flow; gkey.idx = idx; gkey.scr = scr;
    gkey.C = C; gkey.H = H; gkey.W = W; gkey.k = k; gkey.maxh = maxh; gkey.maxw = maxw; gkey.nratios = nratios; gkey.f16 = f16_scale; gkey.u8 = u8_scale;
    for (int s = 0; s < nratios; ++s) gkey.ratios[s] = ratios[s];
    const int gmode = filt ? 0 : dfe_graph_lookup(ctx, ctx->ms_graph, &gkey, sizeof gkey);
    if (gmode == 2) {
        DFE_HIP(ctx, hipGraphLaunch(ctx->ms_graph.exec, ctx->stream));
        ctx->last_kernel = "multiscale graph";
        return DFE_OK;
    }
    auto launch_all = [&]() -> int {
    MultiGeom mg;
    mg.maxh = maxh; mg.maxw = maxw; mg.nratios = nratios;
    // the scales are independent until the cascade: one launch prepares every scale's frames, one per scale builds its
    // volume, one takes every soft-min (2 + nratios launches + the cascade instead of 3 nratios + 1)
    PrepScales ps;
    SoftScales ss;
    long long prep_max = 0, soft_max = 0;
    for (int s = 0; s < nratios; ++s) {
        const int r = ratios[s], Hs = H / r, Ws = W / r, Hp = Hs + hp, Wp = Ws + wp;
        ps.r[s] = r; ps.Hp[s] = Hp; ps.Wp[s] = Wp;
        ps.p0[s] = (float *)((char *)scr + off_p[s]);
        ps.p1[s] = ps.p0[s] + (size_t)C * Hp * Wp;
        ss.cost[s] = (float *)((char *)scr + off_v[s]);
        ss.prob[s] = (float *)((char *)scr + off_q[s]);
        ss.P[s] = (long long)Hs * Ws;
        DFE_REQUIRE(ctx, 2ll * C * Hp * Wp < (1ll << 31), DFE_E_SHAPE, "dfe_multiscale_flow_pair_f32: frame too large");
        if (2ll * C * Hp * Wp > prep_max) prep_max = 2ll * C * Hp * Wp;
        if (ss.P[s] > soft_max) soft_max = ss.P[s];
        g.in[s] = ss.prob[s];
        g.out_scale[s] = nullptr;
        mg.ratios[s] = r;
        mg.d[s] = g.d[s];
    }
    const bool fast = N <= 64 && nratios <= 5;   // one-cell-per-lane path of the cascade kernel
    // lane <-> pixel path (cascade_px_kernel): 8 x 8 windows, ratios 1, 2, 4, ...; DFE_CASCADE_PX=0 keeps the lane <-> cell kernels
    bool px_path = fast && maxh == 8 && maxw == 8;
    for (int s = 0; s < nratios; ++s)
        if (ratios[s] != (1 << s) || (s > 0 && g.d[s] != 2)) px_path = false;   // (ring width 2: the kernel's compile-time class order)
    px_path = px_path && ctx->opt[DFE_OPT_CASCADE_PX] != 0;
    // raw patches, 3 channels, 7 x 7, on the lane <-> pixel path: the finest scale is fused into its volume kernel (no volume)
    // -- where it pays: the fused kernel is VALU-bound (three 64-lane reductions per pixel), the volume it saves is HBM traffic that the
    // small frames hide behind the coarse scales' launches.  Measured, volume path -> fused: VGA 0.076 -> 0.099 ms, 720p 0.213 ->
    // 0.185, 1080p 0.430 -> 0.366, 4K 1.94 -> 1.40; fp16 volumes (half the bytes to save): 1080p 0.344 -> 0.368, 4K 1.48 -> 1.32.
    // DFE_FINE_FUSE=0 / 1 forces the choice.
    // Learned filters: the same epilogue behind the one-chunk feature matcher (feat_matching_win64_fine_kernel), where every scale's
    // stack ends in the same number of planes (<= 16).
    int fK = 0;
    bool fine_shape = C == 3 && k == 7;
    if (filt) {
        fK = filt->layers[filt->nlayers - 1].nOut;
        fine_shape = dfe_feat_matching_win64_ok(ctx, fK, maxh, maxw);   // the launcher's own conditions (cv_mode, fm64, K, LDS)
        for (int s = 1; s < nratios && !filt->share; ++s) fine_shape = fine_shape && filt->layers[s * filt->nlayers + filt->nlayers - 1].nOut == fK;
    }
    // (learned filters, volume path -> fused: VGA 0.193 -> 0.183 ms, 1080p 1.064 -> 0.912: the matcher's launch is long enough at VGA already)
    bool try_fine = px_path && fine_shape && (long long)H * W >= (filt ? 250000ll : f16_scale != 0.f ? 3000000ll : 600000ll);
    if (ctx->opt[DFE_OPT_FINE_FUSE] >= 0) try_fine = px_path && fine_shape && ctx->opt[DFE_OPT_FINE_FUSE] != 0;
    // the second scale the same way (its volume, 21 % of the rest, is otherwise written by the volume kernel and read back by its
    // cascade launch): wherever the finest scale is fused and a coarser scale exists above it.  DFE_MID_FUSE=0 / 1 forces the choice.
    // Measured, finest scale fused -> both: 1080p 0.363 -> 0.346 ms, 4K 1.40 -> 1.29; 720p 0.179 -> 0.205 (230 k pixels: one partial round
    // of blocks, latency-bound); fp16 volumes (half the bytes to save) 4K 1.315 -> 1.320.
    const bool mid_shape = nratios >= 3 && (filt ? (H / ratios[1] >= 16 && W / ratios[1] >= 8 && ((H / ratios[1]) | (W / ratios[1])) % 2 == 0)
                                                 : cv_finest_plan_ok(ctx, H / ratios[1] + k - 1 + maxh - 1, W / ratios[1] + k - 1 + maxw - 1, maxh, maxw));
    bool try_mid = try_fine && f16_scale == 0.f && (long long)H * W >= 1500000ll && mid_shape;
    if (ctx->opt[DFE_OPT_MID_FUSE] >= 0) try_mid = try_fine && ctx->opt[DFE_OPT_MID_FUSE] != 0 && mid_shape;
    const int s0 = try_mid ? 2 : try_fine ? 1 : 0;      // the first scale whose volume is materialised
    if (!filt) {
        DfeStageScope st(ctx, DFE_STAGE_FILTER);
        bool pow2 = nratios <= 5;
        for (int s = 0; s < nratios; ++s) pow2 = pow2 && (ratios[s] == 1 || ratios[s] == 2 || ratios[s] == 4 || ratios[s] == 8 || ratios[s] == 16);
        // (large frames only: at VGA the per-scale kernel's 8.8 us are all latency and the tile kernel's sequential r x r sums are
        //  slower -- 0.082 against 0.076 ms per pair; 1080p 0.451 -> 0.430 ms, 4K five levels 1.69 -> 1.48 ms)
        // (tried: the finest scale's padded frames made on a second stream next to the coarse scales' chain, forked here and joined in
        //  front of the fused kernel, which alone reads them -- 720p 0.178 -> 0.190 ms, 1080p 0.347 -> 0.348, 4K 1.27 -> 1.31: the
        //  frames are read twice and the cross-stream waits cost more than the overlap gives)
        if (pow2 && (long long)H * W >= 1500000 && ctx->opt[DFE_OPT_PREP_TILES] != 0) {       // every scale from one read of the frames
            PrepTile pq;
            pq.ns = nratios;
            for (int s = 0; s < nratios; ++s) { pq.r[s] = ps.r[s]; pq.Hp[s] = ps.Hp[s]; pq.Wp[s] = ps.Wp[s]; pq.p[0][s] = ps.p0[s]; pq.p[1][s] = ps.p1[s]; }
            if (u8_scale > 0.f)
                hipLaunchKernelGGL(prep_tiles_kernel<unsigned char>, dim3(dfe_cdiv(W, PT), dfe_cdiv(H, PT), 2 * C), dim3(256), 0, ctx->stream,
                                   (const unsigned char *)I0, (const unsigned char *)I1, C, H, W, pl, pt, pq, u8_scale);
            else
                hipLaunchKernelGGL(prep_tiles_kernel<float>, dim3(dfe_cdiv(W, PT), dfe_cdiv(H, PT), 2 * C), dim3(256), 0, ctx->stream, I0, I1, C, H, W, pl, pt, pq, 0.f);
        } else {
            if (u8_scale > 0.f)
                hipLaunchKernelGGL(prep_scales_kernel<unsigned char>, dim3(grid1d(prep_max, 256 * DFE_PREP_EPT), nratios), dim3(256), 0, ctx->stream,
                                   (const unsigned char *)I0, (const unsigned char *)I1, C, H, W, pl, pt, ps, u8_scale);
            else
                hipLaunchKernelGGL(prep_scales_kernel<float>, dim3(grid1d(prep_max, 256 * DFE_PREP_EPT), nratios), dim3(256), 0, ctx->stream, I0, I1, C, H, W, pl, pt, ps, 0.f);
        }
        DFE_LAUNCH_CHECK(ctx);
    }
    bool merged = false, soft_done = false, half_vol = false;
    const float *feat0[2] = {nullptr, nullptr}, *feat1[2] = {nullptr, nullptr};      // learned filters: the two finest scales' feature planes (frame 0, frame 1)
    // stage "match": volumes, soft-min, cascade + fused arg-max (with learned filters it opens behind the per-scale filter / matching scopes)
    std::unique_ptr<DfeStageScope> match_rest;
    if (!filt) match_rest.reset(new DfeStageScope(ctx, DFE_STAGE_MATCH));
    if (filt) {
        // learned filters (getModelMultiscale's filter1 / filter2, opticalflow_model_multiscale.lua:196-211): frame 0 is cropped by
        // the search window BEFORE the filter (its zero padding shrinks by floor / ceil((maxh-1)/2)), both padded frames go
        // through the stack, nn.SpatialMatching(maxh, maxw) runs on the K-plane features
        const int ct = (maxh - 1) / 2, cl = (maxw - 1) / 2;
        PrepFrames pf;
        long long pmax = 0;
        for (int s = 0; s < nratios; ++s) {
            const int r = ratios[s], Hs = H / r, Ws = W / r;
            pf.img[2 * s] = I0; pf.out[2 * s] = ps.p0[s]; pf.r[2 * s] = r;
            pf.pl[2 * s] = pl - cl; pf.pt[2 * s] = pt - ct; pf.Hp[2 * s] = Hs + hk - 1; pf.Wp[2 * s] = Ws + wk - 1;
            pf.img[2 * s + 1] = I1; pf.out[2 * s + 1] = ps.p1[s]; pf.r[2 * s + 1] = r;
            pf.pl[2 * s + 1] = pl; pf.pt[2 * s + 1] = pt; pf.Hp[2 * s + 1] = Hs + hp; pf.Wp[2 * s + 1] = Ws + wp;
            if ((long long)C * (Hs + hp) * (Ws + wp) > pmax) pmax = (long long)C * (Hs + hp) * (Ws + wp);
        }
        {
            DfeStageScope st(ctx, DFE_STAGE_FILTER);
            hipLaunchKernelGGL(prep_frames_kernel, dim3(grid1d(pmax, 256 * 8), 2 * nratios), dim3(256), 0, ctx->stream, C, H, W, pf);
        }
        DFE_LAUNCH_CHECK(ctx);
        // layer by layer, both frames of every scale in ONE launch (18 launches of a few microseconds each otherwise)
        const float *cur[2 * DFE_MAX_RATIOS];
        int ch[2 * DFE_MAX_RATIOS], cw[2 * DFE_MAX_RATIOS];
        for (int e = 0; e < 2 * nratios; ++e) { cur[e] = pf.out[e]; ch[e] = pf.Hp[e]; cw[e] = pf.Wp[e]; }
        for (int l = 0; l < filt->nlayers; ++l) {
            DfeStageScope st(ctx, DFE_STAGE_FILTER);
            const dfe_filter_layer *Lp[2 * DFE_MAX_RATIOS];
            float *dst[2 * DFE_MAX_RATIOS];
            for (int e = 0; e < 2 * nratios; ++e) {
                const int s = e >> 1, f = e & 1;
                Lp[e] = filt->layers + (filt->share ? 0 : s) * filt->nlayers + l;
                dst[e] = (float *)((char *)scr + off_f[s] + (size_t)(2 * f + (l & 1)) * fbuf[s]);
            }
            rc = dfe_filter_layer_forward_batch(ctx, 2 * nratios, cur, Lp, ch, cw, dst);
            if (rc) return rc;
            for (int e = 0; e < 2 * nratios; ++e) { cur[e] = dst[e]; ch[e] -= Lp[e]->kH - 1; cw[e] -= Lp[e]->kW - 1; }
        }
        {
            // every scale's nn.SpatialMatching in ONE launch where the one-chunk matcher applies (8 x 8 windows) -- with fp16 volumes
            // written as halves directly when the lane <-> pixel cascade will read them; else scale by scale
            DfeStageScope st(ctx, DFE_STAGE_MATCH);
            const int K = filt->layers[filt->nlayers - 1].nOut;
            bool same_k = true;
            for (int s = 1; s < nratios && !filt->share; ++s) same_k = same_k && filt->layers[s * filt->nlayers + filt->nlayers - 1].nOut == K;
            const float *m1[DFE_MAX_RATIOS], *m2[DFE_MAX_RATIOS];
            float *mo[DFE_MAX_RATIOS];
            int mh[DFE_MAX_RATIOS], mw[DFE_MAX_RATIOS];
            for (int s = 0; s < nratios; ++s) { m1[s] = cur[2 * s]; m2[s] = cur[2 * s + 1]; mo[s] = (float *)ss.cost[s]; mh[s] = H / ratios[s]; mw[s] = W / ratios[s]; }
            feat0[0] = m1[0]; feat0[1] = m2[0];
            if (nratios > 1) { feat1[0] = m1[1]; feat1[1] = m2[1]; }
            bool done = nratios - s0 < 1;       // (s0 = 1: the finest scale's matcher runs last, with the fused epilogue -- below)
            if (same_k && !done) {
                rc = dfe_feat_matching_win64_batch(ctx, nratios - s0, m1 + s0, m2 + s0, K, mh + s0, mw + s0, maxh, maxw, mo + s0, (f16_scale != 0.f && px_path) ? f16_scale : 0.f, &done);
                if (rc) return rc;
                half_vol = done && f16_scale != 0.f && px_path;
            }
            for (int s = s0; s < nratios && !done; ++s) {
                const dfe_filter_layer *Ls = filt->layers + (filt->share ? 0 : s) * filt->nlayers;
                rc = dfe_spatial_matching_dispatch(ctx, m1[s], m2[s], Ls[filt->nlayers - 1].nOut, mh[s], mw[s], maxh, maxw, mo[s]);
                if (rc) return rc;
            }
        }
        merged = true;
        match_rest.reset(new DfeStageScope(ctx, DFE_STAGE_MATCH));
    } else {
        // one launch for every scale's volume; on the fast path the coarser scales leave it as soft-min probabilities already
        // (their blocks run next to the scale-1 blocks that dominate the launch), scale 1 as costs for the cascade's SOFT0
        // (s0 = 1: the finest scale has NO volume -- its task rows are consumed inside the volume kernel, cv_frames_finest_fused below)
        const float *f0[DFE_MAX_RATIOS], *f1[DFE_MAX_RATIOS];
        float *vo[DFE_MAX_RATIOS], *pr[DFE_MAX_RATIOS];
        int vh[DFE_MAX_RATIOS], vw[DFE_MAX_RATIOS];
        const int nv = nratios - s0;
        const int nq_hint = (try_fine && (long long)H * W < 4000000ll) ? 3 : 0;
        for (int s = s0; s < nratios; ++s) {
            f0[s - s0] = ps.p0[s]; f1[s - s0] = ps.p1[s]; vo[s - s0] = (float *)ss.cost[s]; vh[s - s0] = ps.Hp[s]; vw[s - s0] = ps.Wp[s];
            pr[s - s0] = (fast && s > 0 && !px_path) ? ss.prob[s] : nullptr;
        }
        // fp16 volumes: written as halves by the volume kernel itself where the lane <-> pixel cascade will read them (8 x 8
        // windows, C = 3, k = 7); any other shape builds fp32 volumes and rounds them to half precision in place -- same values
        if (f16_scale != 0.f && px_path && nv >= 1) {
            rc = cv_frames_dispatch_multi(ctx, nv, f0, f1, C, vh, vw, k, maxh, maxw, vo, nullptr, &merged, &soft_done, f16_scale, nq_hint);
            if (rc) return rc;
            half_vol = merged;
        }
        if (!merged && nv >= 1) {
            rc = cv_frames_dispatch_multi(ctx, nv, f0, f1, C, vh, vw, k, maxh, maxw, vo, f16_scale != 0.f ? nullptr : pr, &merged, &soft_done, 0.f, nq_hint);
            if (rc) return rc;
        }
    }
    for (int s = s0; s < nratios && !merged; ++s) {
        rc = cv_frames_dispatch(ctx, ps.p0[s], ps.p1[s], C, ps.Hp[s], ps.Wp[s], (long long)ps.Hp[s] * ps.Wp[s], k, k, maxh, maxw,
                                (float *)ss.cost[s]);
        if (rc) return rc;
    }
    if (f16_scale != 0.f && !half_vol) {
        for (int s = s0; s < nratios; ++s) {
            const long long n = ss.P[s] * N;
            hipLaunchKernelGGL(round_half_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, (float *)ss.cost[s], n, f16_scale, 1.0f / f16_scale);
        }
        DFE_LAUNCH_CHECK(ctx);
    }
    const int middle = ((maxh + 1) / 2 - 1) * maxw + (maxw + 1) / 2;   // yx2xMulti(0, 0)
    if (px_path) {
        DecodeTab dt;
        fill_decode_tab(mg, g.ncls, dt);
        fill_cell_maps(g, dt);
        // coarse -> fine: one launch per scale, except that the coarsest scale is recomputed inside its child's launch
        const int top = nratios >= 2 ? nratios - 2 : 0;
        bool fine_done = false;
        // where the coarsest scale is the parent of a fused scale it needs a launch of its own (it cannot be recomputed inside its child's)
        const bool lone_parent = (try_mid && nratios == 3) || (try_fine && !try_mid && nratios == 2);
        for (int s = lone_parent ? nratios - 1 : top; s >= 0; --s) {
            const int r = ratios[s];
            if (s == 1 && try_mid) {
                CvFineArgs mid{};
                mid.pcasc = (const float *)((char *)scr + off_q[2]);
                mid.pbest = (const float2 *)((char *)scr + off_b[2]);
                mid.casc = (float *)((char *)scr + off_q[1]);
                mid.best = (float2 *)((char *)scr + off_b[1]);
                mid.cls_base = g.base[1];
                mid.f16_scale = f16_scale;
                mid.f16_inv = f16_scale != 0.f ? 1.0f / f16_scale : 0.f;
                bool mid_done = false;
                if (filt) {
                    float *none = nullptr;
                    const int h1 = H / ratios[1], w1 = W / ratios[1];
                    rc = dfe_feat_matching_win64_batch(ctx, 1, &feat1[0], &feat1[1], fK, &h1, &w1, maxh, maxw, &none, 0.f, &mid_done, &mid);
                } else {
                    rc = cv_frames_finest_fused(ctx, ps.p0[1], ps.p1[1], C, ps.Hp[1], ps.Wp[1], k, maxh, maxw, mid, &mid_done);
                }
                if (rc) return rc;
                DFE_REQUIRE(ctx, mid_done, DFE_E_UNSUPPORTED, "multiscale: no plan for the fused second scale (%d x %d) although cv_finest_plan_ok said so", ps.Hp[1], ps.Wp[1]);
                continue;
            }
            if (s == 0 && try_fine) {
                CvFineArgs fine{};
                if (nratios >= 2) {
                    fine.pcasc = (const float *)((char *)scr + off_q[1]);
                    fine.pbest = (const float2 *)((char *)scr + off_b[1]);
                }
                fine.idx = (long long *)idx;
                fine.fy = flow;
                fine.fx = flow ? flow + (size_t)H * W : nullptr;
                fine.middle = middle;
                fine.f16_scale = f16_scale;
                fine.f16_inv = f16_scale != 0.f ? 1.0f / f16_scale : 0.f;
                for (int c = 0; c < 5 * 64; ++c) fine.dec[c] = c < g.ncls ? dt.v[c] : 0;
                if (filt) {
                    float *none = nullptr;
                    rc = dfe_feat_matching_win64_batch(ctx, 1, &feat0[0], &feat0[1], fK, &H, &W, maxh, maxw, &none, 0.f, &fine_done, &fine);
                } else {
                    rc = cv_frames_finest_fused(ctx, ps.p0[0], ps.p1[0], C, ps.Hp[0], ps.Wp[0], k, maxh, maxw, fine, &fine_done);
                }
                if (rc) return rc;
                if (fine_done) break;
                // no plan for this frame: the scale-1 volume after all (fp32; rounded in place for the fp16 entry), then the px kernel
                if (filt) rc = dfe_spatial_matching_dispatch(ctx, feat0[0], feat0[1], fK, H, W, maxh, maxw, (float *)ss.cost[0]);
                else rc = cv_frames_dispatch(ctx, ps.p0[0], ps.p1[0], C, ps.Hp[0], ps.Wp[0], (long long)ps.Hp[0] * ps.Wp[0], k, k, maxh, maxw, (float *)ss.cost[0]);
                if (rc) return rc;
                if (f16_scale != 0.f) {
                    const long long n = ss.P[0] * N;
                    hipLaunchKernelGGL(round_half_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, (float *)ss.cost[0], n, f16_scale, 1.0f / f16_scale);
                }
            }
            CascadePxArgs a{};
            a.cost = (const float *)ss.cost[s];
            a.Hs = H / r; a.Ws = W / r; a.scale = s; a.middle = middle; a.cls_base = g.base[s];
            a.inv_scale = (half_vol && !(s == 0 && try_fine)) ? 1.0f / f16_scale : 1.0f;
            const bool inl = nratios >= 2 && s == top && !lone_parent && !(s == 0 && try_fine);
            if (inl) {
                a.pcost = (const float *)ss.cost[s + 1];
                a.pcls_base = g.base[s + 1];
            } else if (s + 1 < nratios) {
                a.pcasc = (const float *)((char *)scr + off_q[s + 1]);
                a.pbest = (const float2 *)((char *)scr + off_b[s + 1]);
            }
            const int blocks = (int)(((long long)a.Hs * a.Ws + 255) / 256);
            if (s > 0) {
                a.casc = (float *)((char *)scr + off_q[s]);
                a.best = (float2 *)((char *)scr + off_b[s]);
                if (half_vol) {
                    if (inl) hipLaunchKernelGGL((cascade_px_kernel<false, true, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                    else hipLaunchKernelGGL((cascade_px_kernel<false, false, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                } else if (inl) hipLaunchKernelGGL((cascade_px_kernel<false, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                else hipLaunchKernelGGL((cascade_px_kernel<false, false>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
            } else {
                a.idx = (long long *)idx;
                a.fy = flow;
                a.fx = flow ? flow + (size_t)H * W : nullptr;
                if (half_vol && !try_fine) {
                    if (inl) hipLaunchKernelGGL((cascade_px_kernel<true, true, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                    else hipLaunchKernelGGL((cascade_px_kernel<true, false, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                } else if (inl) hipLaunchKernelGGL((cascade_px_kernel<true, true>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
                else hipLaunchKernelGGL((cascade_px_kernel<true, false>), dim3(blocks), dim3(256), 0, ctx->stream, a, dt);
            }
        }
        DFE_LAUNCH_CHECK(ctx);
        (void)fine_done;
        return DFE_OK;
    }
    // one-cell-per-lane path: the scale-1 soft-min happens inside the cascade kernel (SOFT0), the coarser scales' here
    int nsoft = nratios;
    if (fast) {   // scale 1 is skipped here: the remaining scales move up one slot, so that no idle blocks are launched for it
        g.in[0] = ss.cost[0];
        soft_max = 0;
        for (int s = 1; s < nratios; ++s) {
            ss.cost[s - 1] = ss.cost[s]; ss.prob[s - 1] = ss.prob[s]; ss.P[s - 1] = ss.P[s];
            if (ss.P[s] > soft_max) soft_max = ss.P[s];
        }
        nsoft = nratios - 1;
    }
    if (soft_max > 0 && nsoft > 0 && !soft_done) {   // (soft_done: the volume launch has taken the coarser scales' soft-mins)
        hipLaunchKernelGGL(softmin_scales_kernel, dim3(grid1d(soft_max, kWaves * (N <= 64 ? 8 : 4)), nsoft), dim3(kWaves * 64), 0, ctx->stream, ss, N);
        DFE_LAUNCH_CHECK(ctx);
    }
    g.H = H; g.W = W;
    size_t lds = (size_t)kWaves * 2 * N * sizeof(float);
    DFE_REQUIRE(ctx, lds <= 64 * 1024, DFE_E_UNSUPPORTED, "dfe_multiscale_flow_pair_f32: window %dx%d too large", maxh, maxw);
    if (lds < (size_t)g.ncls * sizeof(int2)) lds = (size_t)g.ncls * sizeof(int2);
    dim3 grid(grid1d((long long)H * W, kWaves));
    DecodeTab dt;
    fill_decode_tab(mg, g.ncls, dt);
    if (g.maxh * g.maxw <= 64 && g.nratios <= 5) fill_cell_maps(g, dt);
    if (fast) {
        grid = cascade_fast_grid(H, W);
        hipLaunchKernelGGL(cascade_argmax_kernel<true>, grid, dim3(kWaves * 64), lds, ctx->stream, g, mg, middle, (long long *)idx, (float *)nullptr,
                           flow, flow ? flow + (size_t)H * W : nullptr, W, 0, 0, dt);
    } else {
        hipLaunchKernelGGL(cascade_argmax_kernel<false>, grid, dim3(kWaves * 64), lds, ctx->stream, g, mg, middle, (long long *)idx, (float *)nullptr,
                           flow, flow ? flow + (size_t)H * W : nullptr, W, 0, 0, dt);
    }
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
    };
    if (gmode == 1) return dfe_graph_finish(ctx, ctx->ms_graph, launch_all());
    return launch_all();
}

int dfe_multiscale_flow_pair_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int maxh, int maxw,
                                 const int *ratios, int nratios, float *flow, int64_t *idx) {
    DFE_ENTER(ctx);
    return multiscale_flow_pair(ctx, I0, I1, C, H, W, k, maxh, maxw, ratios, nratios, flow, idx, 0.f);
}

int dfe_multiscale_flow_pair_f16(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int k, int maxh, int maxw,
                                 const int *ratios, int nratios, float scale, float *flow, int64_t *idx) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, scale > 0.f && scale < INFINITY, DFE_E_ARG, "dfe_multiscale_flow_pair_f16: scale=%g must be positive", (double)scale);
    return multiscale_flow_pair(ctx, I0, I1, C, H, W, k, maxh, maxw, ratios, nratios, flow, idx, scale);
}

int dfe_multiscale_flow_pair_filtered_f32(dfe_ctx *ctx, const float *I0, const float *I1, int C, int H, int W, int maxh, int maxw,
                                          const int *ratios, int nratios, const dfe_filter_layer *layers, int nlayers, int share_filters,
                                          float f16_scale, float *flow, int64_t *idx) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, layers && nlayers >= 1 && nlayers <= 16, DFE_E_ARG, "dfe_multiscale_flow_pair_filtered_f32: nlayers=%d", nlayers);
    DFE_REQUIRE(ctx, f16_scale >= 0.f && f16_scale < INFINITY, DFE_E_ARG, "dfe_multiscale_flow_pair_filtered_f32: f16_scale=%g", (double)f16_scale);
    DFE_REQUIRE(ctx, layers[0].nIn == C || layers[0].conn, DFE_E_SHAPE, "dfe_multiscale_flow_pair_filtered_f32: frames have %d channels, the first layer reads %d", C,
                layers[0].nIn);
    const MsFilter filt{layers, nlayers, share_filters};
    return multiscale_flow_pair(ctx, I0, I1, C, H, W, 1, maxh, maxw, ratios, nratios, flow, idx, f16_scale, &filt);
}

int dfe_softmin_f32(dfe_ctx *ctx, const float *cost, int64_t P, int N, float *prob) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && N > 0, DFE_E_SHAPE, "dfe_softmin_f32: P=%lld N=%d", (long long)P, N);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, cost && prob, DFE_E_ARG, "dfe_softmin_f32: NULL tensor");
    hipLaunchKernelGGL(softmin_kernel, dim3(grid1d(P, kWaves * (N <= 64 ? 8 : 4))), dim3(kWaves * 64), 0, ctx->stream, cost, (long long)P, N, prob);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_cascade_ring_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios, int H, int W, int maxh, int maxw,
                         float *out) {
    DFE_ENTER(ctx);
    CascadeGeom g;
    int rc = fill_cascade(ctx, g, ratios, nratios, maxh, maxw);
    if (rc) return rc;
    DFE_REQUIRE(ctx, prob && out && H > 0 && W > 0, DFE_E_ARG, "dfe_cascade_ring_f32: bad argument");
    for (int s = 0; s < nratios; ++s) {
        DFE_REQUIRE(ctx, prob[s], DFE_E_ARG, "dfe_cascade_ring_f32: prob[%d] is NULL", s);
        DFE_REQUIRE(ctx, H % ratios[s] == 0 && W % ratios[s] == 0, DFE_E_SHAPE, "dfe_cascade_ring_f32: %dx%d not a multiple of ratio %d", H, W, ratios[s]);
        g.in[s] = prob[s];
        g.out_scale[s] = nullptr;
    }
    g.H = H; g.W = W;
    size_t lds = (size_t)kWaves * 2 * maxh * maxw * sizeof(float);
    DFE_REQUIRE(ctx, lds <= 64 * 1024, DFE_E_UNSUPPORTED, "dfe_cascade_ring_f32: window %dx%d too large", maxh, maxw);
    hipLaunchKernelGGL(cascade_kernel<true>, dim3(grid1d((long long)H * W, kWaves)), dim3(kWaves * 64), lds, ctx->stream, g, out);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_cascade_flow_f32(dfe_ctx *ctx, const float *const *prob, const int *ratios, int nratios, int H, int W, int maxh, int maxw,
                         int64_t *idx, float *best, float *flow_y, float *flow_x) {
    DFE_ENTER(ctx);
    CascadeGeom g;
    int rc = fill_cascade(ctx, g, ratios, nratios, maxh, maxw);
    if (rc) return rc;
    DFE_REQUIRE(ctx, prob && H > 0 && W > 0, DFE_E_ARG, "dfe_cascade_flow_f32: bad argument");
    DFE_REQUIRE(ctx, (flow_y == nullptr) == (flow_x == nullptr), DFE_E_ARG, "dfe_cascade_flow_f32: flow_y and flow_x go together");
    DFE_REQUIRE(ctx, idx || flow_y || best, DFE_E_ARG, "dfe_cascade_flow_f32: no output requested");
    MultiGeom mg;
    mg.maxh = maxh; mg.maxw = maxw; mg.nratios = nratios;
    for (int s = 0; s < nratios; ++s) {
        DFE_REQUIRE(ctx, prob[s], DFE_E_ARG, "dfe_cascade_flow_f32: prob[%d] is NULL", s);
        DFE_REQUIRE(ctx, H % ratios[s] == 0 && W % ratios[s] == 0, DFE_E_SHAPE, "dfe_cascade_flow_f32: %dx%d not a multiple of ratio %d", H, W, ratios[s]);
        g.in[s] = prob[s];
        g.out_scale[s] = nullptr;
        mg.ratios[s] = ratios[s];
        mg.d[s] = g.d[s];
    }
    g.H = H; g.W = W;
    // middle class = yx2xMulti(0, 0): the centre cell of scale 1 (opticalflow_model.lua:36-43)
    const int middle = ((maxh + 1) / 2 - 1) * maxw + (maxw + 1) / 2;
    size_t lds = (size_t)kWaves * 2 * maxh * maxw * sizeof(float);
    DFE_REQUIRE(ctx, lds <= 64 * 1024, DFE_E_UNSUPPORTED, "dfe_cascade_flow_f32: window %dx%d too large", maxh, maxw);
    if (lds < (size_t)g.ncls * sizeof(int2)) lds = (size_t)g.ncls * sizeof(int2);   // fast path: class -> displacement table
    dim3 grid(grid1d((long long)H * W, kWaves));
    if (maxh * maxw <= 64 && nratios <= 5) grid = cascade_fast_grid(H, W);   // fast path: one row per blockIdx.y
    DecodeTab dt;
    fill_decode_tab(mg, g.ncls, dt);
    if (g.maxh * g.maxw <= 64 && g.nratios <= 5) fill_cell_maps(g, dt);
    hipLaunchKernelGGL(cascade_argmax_kernel<false>, grid, dim3(kWaves * 64), lds, ctx->stream, g, mg, middle,
                       (long long *)idx, best, flow_y, flow_x, W, 0, 0, dt);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_cascading_add_f32(dfe_ctx *ctx, const float *const *in, const int *ratios, int nratios, int64_t P, int maxh, int maxw,
                          float *const *out) {
    DFE_ENTER(ctx);
    CascadeGeom g;
    int rc = fill_cascade(ctx, g, ratios, nratios, maxh, maxw);
    if (rc) return rc;
    DFE_REQUIRE(ctx, in && out && P >= 0, DFE_E_ARG, "dfe_cascading_add_f32: bad argument");
    if (P == 0) return DFE_OK;
    for (int s = 0; s < nratios; ++s) {
        DFE_REQUIRE(ctx, in[s] && out[s], DFE_E_ARG, "dfe_cascading_add_f32: tensor %d is NULL", s);
        g.in[s] = in[s];
        g.out_scale[s] = out[s];
    }
    DFE_REQUIRE(ctx, P <= 0x7fffffff, DFE_E_SHAPE, "dfe_cascading_add_f32: P too large");
    g.H = 1; g.W = (int)P;
    size_t lds = (size_t)kWaves * 2 * maxh * maxw * sizeof(float);
    DFE_REQUIRE(ctx, lds <= 64 * 1024, DFE_E_UNSUPPORTED, "dfe_cascading_add_f32: window %dx%d too large", maxh, maxw);
    hipLaunchKernelGGL(cascade_kernel<false>, dim3(grid1d(P, kWaves)), dim3(kWaves * 64), lds, ctx->stream, g, (float *)nullptr);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_cascading_add_backward_f32(dfe_ctx *ctx, const float *const *gradOut, const int *ratios, int nratios, int64_t P, int maxh,
                                   int maxw, float *const *gradIn) {
    DFE_ENTER(ctx);
    CascadeGeom g;
    int rc = fill_cascade(ctx, g, ratios, nratios, maxh, maxw);
    if (rc) return rc;
    DFE_REQUIRE(ctx, gradOut && gradIn && P >= 0, DFE_E_ARG, "dfe_cascading_add_backward_f32: bad argument");
    if (P == 0) return DFE_OK;
    for (int s = 0; s < nratios; ++s) {
        DFE_REQUIRE(ctx, gradOut[s] && gradIn[s], DFE_E_ARG, "dfe_cascading_add_backward_f32: tensor %d is NULL", s);
        g.in[s] = gradOut[s];
        g.out_scale[s] = gradIn[s];
    }
    DFE_REQUIRE(ctx, P <= 0x7fffffff, DFE_E_SHAPE, "dfe_cascading_add_backward_f32: P too large");
    g.H = 1; g.W = (int)P;
    size_t lds = (size_t)kWaves * 2 * maxh * maxw * sizeof(float);
    DFE_REQUIRE(ctx, lds <= 64 * 1024, DFE_E_UNSUPPORTED, "dfe_cascading_add_backward_f32: window %dx%d too large", maxh, maxw);
    hipLaunchKernelGGL(cascade_backward_kernel, dim3(grid1d(P, kWaves)), dim3(kWaves * 64), lds, ctx->stream, g);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"

// uint8 frames straight into the preparation kernels (dfe_multiscale_flow_pair_u8, ingest.hip); f16_scale as in the _f16 entry, 0 = fp32 volumes
int dfe_multiscale_flow_pair_bytes(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int maxh, int maxw,
                                   const int *ratios, int nratios, float u8_scale, float f16_scale, float *flow, int64_t *idx) {
    return multiscale_flow_pair(ctx, (const float *)I0, (const float *)I1, C, H, W, k, maxh, maxw, ratios, nratios, flow, idx, f16_scale, nullptr, u8_scale);
}
