// postops.hip -- per-pixel consumers of a cost / probability volume: A6 arg-best with the centre
// tie-break, A7/A8 extractOutput(+Marginalized), A9/A10 class-id decode.  All of them stream the
// [P][N] volume once (one wave per pixel, 256-B coalesced reads) or are elementwise over [P].
#include "dfe_internal.h"
#include <cmath>

namespace {

constexpr int kWavesPerBlock = 4;   // one wave per pixel; 4 waves per block

// ---- A6 ------------------------------------------------------------------------------------
// replaces: m,idx = output:min(3) + centre override, radial/radial_opticalflow_groundtruth.lua:88-94;
//           input:max(3) twin, opticalflow_model.lua:153-161.  First extremum wins (strict compare).
template <bool TAKE_MAX>
__global__ __launch_bounds__(kWavesPerBlock * 64) void argbest_kernel(const float *__restrict__ vol, long long P, int N,
                                                                     int middle, long long *__restrict__ idx,
                                                                     float *__restrict__ best) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    for (long long p = wave0; p < P; p += (long long)gridDim.x * kWavesPerBlock) {
        const float *v = vol + p * N;
        float b = 0.f;
        int bi = 0x7fffffff;   // "no candidate yet"
        for (int n = lane; n < N; n += 64) {
            float t = v[n];
            bool better = (bi == 0x7fffffff) || (TAKE_MAX ? (t > b) : (t < b));
            if (better) { b = t; bi = n; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float ob = __shfl_xor(b, off);
            int oi = __shfl_xor(bi, off);
            bool take = (oi != 0x7fffffff) &&
                        ((bi == 0x7fffffff) || (TAKE_MAX ? (ob > b) : (ob < b)) || (ob == b && oi < bi));
            if (take) { b = ob; bi = oi; }
        }
        if (lane == 0) {
            long long id = (long long)bi + 1;
            if (middle > 0 && b == v[middle - 1]) id = middle;
            idx[p] = id;
            if (best) best[p] = b;
        }
    }
}

// ---- A7 / A8 -------------------------------------------------------------------------------
// replaces: extract_output.cpp:63-155 / :157-255.  The reference scans the N values of a pixel in
// index order and keeps the first M that exceed the threshold; a wave does the same 64 values at a
// time with a ballot, so the kept set and its order are identical.
// sorting networks: dfe_sort4 / dfe_sort8 in dfe_internal.h (extract_output.cpp:17-61)
#define sort4 dfe_sort4
#define sort8 dfe_sort8

template <int M, bool MARG>
__global__ __launch_bounds__(kWavesPerBlock * 64) void extract_kernel(const float *__restrict__ input, long long P, int N,
                                                                     double threshold, double threshold_acc,
                                                                     long long *__restrict__ imaxs,
                                                                     float *__restrict__ scores,
                                                                     long long *__restrict__ retgd) {
    __shared__ float sh_v[kWavesPerBlock][8];
    __shared__ float sh_i[kWavesPerBlock][8];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long long wave0 = (long long)blockIdx.x * kWavesPerBlock + w;
    for (long long p = wave0; p < P; p += (long long)gridDim.x * kWavesPerBlock) {
        const float *v = input + p * N;
        if (lane < 8) { sh_v[w][lane] = 0.f; sh_i[w][lane] = 0.f; }   // :86-90 zeroed highs
        int n = 0;
        for (int base = 0; base < N && n < M; base += 64) {
            int k = base + lane;
            float t = (k < N) ? v[k] : 0.f;
            bool hit = (k < N) && ((double)t > threshold);            // :103 float vs double threshold
            unsigned long long mask = __ballot(hit);
            int rank = n + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && rank < M) { sh_v[w][rank] = t; sh_i[w][rank] = (float)(k + 1); }   // :104-105
            n += __popcll(mask);                                      // early exit when M are found :107-110
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane == 0) {
            float hv[8], hi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { hv[k] = sh_v[w][k]; hi[k] = sh_i[w][k]; }
            if (hv[0] > 0) {                                          // :121
                if (M == 4) sort4(hv, hi); else sort8(hv, hi);
                imaxs[p] = (long long)hi[0];                          // :123
                for (int k = 1; k < M; ++k) hv[k] += hv[k - 1];       // :124-125 float prefix sums
                double acc = 0;                                       // :126-128 double accumulator
                for (int k = 0; k < M; ++k) acc += hv[k];
                if (MARG) { if (acc >= threshold_acc) retgd[p] = 1; } // :227-228
                else scores[p] = (float)acc;                          // :129
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

// ---- A9 / A10 ------------------------------------------------------------------------------
__global__ void x2yx_kernel(const long long *__restrict__ idx, long long P, int maxh, int maxw, long long *__restrict__ y,
                            long long *__restrict__ x) {
    // replaces: radial/radial_opticalflow_groundtruth.lua:97-100
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        long long id = idx[p];
        long long fl = (id - 1) / maxw;
        y[p] = fl - (maxh - 1) / 2;
        x[p] = id - 1 - fl * maxw - (maxw - 1) / 2;
    }
}

__global__ void x2yx_multi_kernel(MultiGeom g, const long long *__restrict__ idx, long long P, long long *__restrict__ y,
                                  long long *__restrict__ x, int *__restrict__ flag) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        long long oy, ox;
        if (multi_decode(g, idx[p], &oy, &ox) == 0) { y[p] = oy; x[p] = ox; }
        else atomicOr(flag, 1);
    }
}

// bug-compatible with the shipped vectorised body x2yxMulti2.c:1-95 (see include/dfe.h)
struct CompatGeom {
    int maxh, maxw, nratios;
    int ratios[DFE_MAX_RATIOS], borders[DFE_MAX_RATIOS], lengths[DFE_MAX_RATIOS];
};
__global__ void x2yx_multi_compat_kernel(CompatGeom g, const long long *__restrict__ idx, long long P,
                                         long long *__restrict__ rety, long long *__restrict__ retx) {
    const int maxh = g.maxh, maxw = g.maxw;
    const int chmaxh = maxh / 2, chmaxw = maxw / 2, patcharea = maxh * maxw;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        long long x = idx[p];
        if (x < patcharea) {
            rety[p] = (x - 1) / maxw + 1 - chmaxh;
            retx[p] = (x - 1) % maxw + 1 - chmaxw;
            continue;
        }
        x -= patcharea;
        for (int k = 1; k < g.nratios; ++k) {
            const int d = g.borders[k];
            const int mH = (maxh - 2 * d) * d;
            if (x <= g.lengths[k]) {
                if (x < d * maxw) {
                    rety[p] = ((x - 1) / maxw + 1 - chmaxh) * g.ratios[k];
                    retx[p] = ((x - 1) % maxw + 1 - chmaxw) * g.ratios[k];
                    break;
                }
                x -= d * maxw;
                if (x <= mH) {
                    rety[p] = (d ? ((x - 1) / d + 1 + d - chmaxh) : 0) * g.ratios[k];
                    retx[p] = (d ? ((x - 1) % d + 1 - chmaxw) : 0) * g.ratios[k];
                    break;
                }
                x -= mH;
                if (x <= mH) {
                    rety[p] = (d ? ((x - 1) / d + 1 + d - chmaxh) : 0) * g.ratios[k];
                    retx[p] = (d ? ((x - 1) % d + 1 + maxw - d - chmaxw) : 0) * g.ratios[k];
                    break;
                }
                x -= mH;
                if (x < d * maxw) {
                    rety[p] = ((x - 1) / maxw + 1 + maxh - d - chmaxh) * g.ratios[k];
                    retx[p] = ((x - 1) % maxw + 1 - chmaxw) * g.ratios[k];
                    break;
                }
            } else {
                x -= g.lengths[k];
            }
        }
    }
}

// ---- fused single-scale tail: A6 (min + centre tie-break) + A7 + A9 in one pass over the volume ----
// replaces: radial/radial_opticalflow_groundtruth.lua:87-105 (min(3), tie-break, decode, extractOutput)
// (TailOut, pair_depth_px and the record path of the finalize -- dfe_finalize_rec_pixel -- live in dfe_internal.h: the fused sweep
//  finishes its own pixels with the same code)

// Finishes what the fused cost-volume epilogue started; one thread per pixel, everything it normally reads is compact
// and coalesced (8*nchunks + 4 + 64 bytes per pixel):
//  A6: the pixel minimum is the minimum of its chunk minima, taken from the FIRST chunk that attains it together
//      with that chunk's first attaining cell (== the strict '<' scan of the reference); centre override.
//  A9: decode.  A7: extractOutput looks for the first M values above the threshold among the pixel's first DFE_LEAD
//      cells (for cost volumes they are there); only if fewer are found does it walk on through the volume in index
//      order (extract_output.cpp:99-112 stops at M as well).
// The record path by itself (the fused single-scale step: rec != nullptr): the same per-pixel code as flow_finalize_kernel's record
// branch, without the three-plane branch next to it (its registers, its scalar spills and 1 900 lines of code that never run here).
template <int M>
__global__ __launch_bounds__(256) void flow_finalize_rec_kernel(const float *__restrict__ vol, long long Pband, int N, int hWin, int wWin, int middle,
                                                                double threshold, TailOut o, const float *__restrict__ rec, int rec_rows) {
    const long long nthreads_work = o.frame_H ? (long long)o.frame_H * o.frame_W : Pband;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nthreads_work; q += (long long)gridDim.x * blockDim.x) {
        long long p = q;
        int fi = 0, fj = 0;
        if (o.frame_H) {
            fi = (int)((unsigned)q / (unsigned)o.frame_W);
            fj = (int)((unsigned)q - (unsigned)fi * (unsigned)o.frame_W);
            const int iy = fi - o.pad_t, ix = fj - o.pad_l;
            if (iy < 0 || ix < 0 || ix >= o.Wo || (long long)iy * o.Wo + ix >= Pband) {   // border pixel
                o.fy[q] = 0.f;
                o.fx[q] = 0.f;
                if (o.scores) o.scores[q] = 0.f;
                if (o.depth) pair_depth_px(fi, fj, 0.f, 0.f, o.mw, o.mh, o.infty, &o.depth[q], &o.conf[q]);
                continue;
            }
            p = (long long)iy * o.Wo + ix;
            dfe_finalize_rec_pixel<M>(rec, rec_rows, vol, p, N, hWin, wWin, middle, threshold, o, fi, fj, iy, ix);
            continue;
        }
        dfe_finalize_rec_pixel<M>(rec, rec_rows, vol, p, N, hWin, wWin, middle, threshold, o, fi, fj);
    }
}

template <int M>
__global__ __launch_bounds__(256) void flow_finalize_kernel(const float2 *__restrict__ part, const float *__restrict__ centre,
                                                            const float *__restrict__ lead, int nchunks, long long Ptot,
                                                            const float *__restrict__ vol, long long Pband, int N, int hWin,
                                                            int wWin, int middle, double threshold, TailOut o, const float *__restrict__ rec,
                                                            int rec_rows) {
    // rec != nullptr (launch-uniform): minimum / first index / centre cost of a pixel come from its tile row's record
    // [column group][rec_rows][DFE_REC] (CvFuseArgs::rec), its first DFE_REC_NLEAD lead cells from the record too; else from the three planes
    const long long nthreads_work = o.frame_H ? (long long)o.frame_H * o.frame_W : Pband;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nthreads_work; q += (long long)gridDim.x * blockDim.x) {
        long long p = q;
        int fi = 0, fj = 0;
        if (o.frame_H) {
            fi = (int)((unsigned)q / (unsigned)o.frame_W);
            fj = (int)((unsigned)q - (unsigned)fi * (unsigned)o.frame_W);
            const int iy = fi - o.pad_t, ix = fj - o.pad_l;
            if (iy < 0 || ix < 0 || ix >= o.Wo || (long long)iy * o.Wo + ix >= Pband) {   // border pixel
                o.fy[q] = 0.f;
                o.fx[q] = 0.f;
                if (o.scores) o.scores[q] = 0.f;
                if (o.depth) pair_depth_px(fi, fj, 0.f, 0.f, o.mw, o.mh, o.infty, &o.depth[q], &o.conf[q]);
                continue;
            }
            p = (long long)iy * o.Wo + ix;
        }
        if (rec) {
            dfe_finalize_rec_pixel<M>(rec, rec_rows, vol, p, N, hWin, wWin, middle, threshold, o, fi, fj);
            continue;
        }
        const long long pg = o.p_off + p;
        const int y = (int)(p / o.Wo) + o.row_off, x = (int)(p % o.Wo);
        float2 b;
        float cen;
        if (rec) {
            const int ncols = (o.Wo + 7) >> 3;
            const int g = min(x >> 3, ncols - 1), xb = g == ncols - 1 ? o.Wo - 8 : g << 3;   // (the last tile column is shifted inwards)
            const float *rp = rec + ((long long)g * rec_rows + y) * DFE_REC;
            // (non-temporal: what this kernel reads is REWRITTEN by the next frame's cost-volume launch -- lines left in the memory-side
            //  cache by these reads made that launch's stores slower: 1080p 2.4 against 1.8 ms)
            b.x = __builtin_nontemporal_load(rp + 2 * (x - xb));
            b.y = __builtin_nontemporal_load(rp + 2 * (x - xb) + 1);
            cen = __builtin_nontemporal_load(rp + DFE_REC_CENTRE + x - xb);
        } else {
            b = part[pg];
            cen = centre[pg];
        }
        for (int c0 = 1; c0 < nchunks; c0 += 6) {   // six loads in flight, compared in chunk order (first chunk wins ties)
            float2 t[6];
#pragma unroll
            for (int j = 0; j < 6; ++j)
                t[j] = c0 + j < nchunks ? part[(long long)(c0 + j) * Ptot + pg] : make_float2(__int_as_float(0x7f800000), 0.f);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (t[j].x < b.x) b = t[j];
        }
        long long id = (long long)__float_as_int(b.y) + 1;
        if (middle > 0 && b.x == cen) id = middle;
        if (o.idx) o.idx[pg] = id;
        if (o.best) o.best[pg] = b.x;
        const long long fo = (long long)(y + o.pad_t) * o.pitch + x + o.pad_l;
        const long long fl = (id - 1) / wWin;
        const float dyf = (float)(fl - (hWin - 1) / 2), dxf = (float)(id - 1 - fl * wWin - (wWin - 1) / 2);
        if (o.fy) o.fy[fo] = dyf;
        if (o.fx) o.fx[fo] = dxf;
        if (o.frame_H && o.depth) pair_depth_px(fi, fj, dyf, dxf, o.mw, o.mh, o.infty, &o.depth[fo], &o.conf[fo]);
        if (o.scores) {
            float hv[M], hi[M];
#pragma unroll
            for (int j = 0; j < M; ++j) { hv[j] = 0.f; hi[j] = 0.f; }
            int n = 0;
            float qq[DFE_LEAD];
#pragma unroll
            for (int g4 = 0; g4 < DFE_LEAD / 4; ++g4) {                                   // pixel-major [P][DFE_LEAD]: 4 x 16 B
                // (record mode: the pixel's first DFE_REC_NLEAD cells from its tile row's record; DFE_REC_NLEAD = 0 builds read them straight
                //  from the volume -- p * N * 4 bytes is 4-B aligned only: scalar loads)
                float4 q4;
                if (rec) {
                    if (4 * g4 < DFE_REC_NLEAD) {          // the record's copy of the pixel's first cells (32 B, whole-line reads across the tile row)
                        const int ncols = (o.Wo + 7) >> 3;
                        const int g = min(x >> 3, ncols - 1), xb = g == ncols - 1 ? o.Wo - 8 : g << 3;
                        const float *lv = rec + ((long long)g * rec_rows + y) * DFE_REC + DFE_REC_LEAD + (x - xb) * DFE_REC_NLEAD + 4 * g4;
                        q4 = make_float4(__builtin_nontemporal_load(lv), __builtin_nontemporal_load(lv + 1), __builtin_nontemporal_load(lv + 2), __builtin_nontemporal_load(lv + 3));
                    } else if (DFE_REC_NLEAD == 0) {
                        const float *lv = vol + p * N + 4 * g4;
                        q4 = make_float4(__builtin_nontemporal_load(lv), __builtin_nontemporal_load(lv + 1), __builtin_nontemporal_load(lv + 2), __builtin_nontemporal_load(lv + 3));
                    } else {
                        q4 = make_float4(0.f, 0.f, 0.f, 0.f);   // (not looked at: nlead below)
                    }
                }
                else q4 = reinterpret_cast<const float4 *>(lead + pg * DFE_LEAD)[g4];
                qq[4 * g4] = q4.x; qq[4 * g4 + 1] = q4.y; qq[4 * g4 + 2] = q4.z; qq[4 * g4 + 3] = q4.w;
            }
            // (record mode with lead cells in the record: only the first DFE_REC_NLEAD are at hand; the walk through the volume below
            //  takes over behind them -- the same cells in the same order)
            const int nlead = (rec && DFE_REC_NLEAD > 0) ? DFE_REC_NLEAD : DFE_LEAD;
#pragma unroll
            for (int kk = 0; kk < DFE_LEAD; ++kk) qq[kk] = kk < N ? qq[kk] : 0.f;
#pragma unroll
            for (int kk = 0; kk < DFE_LEAD; ++kk) {
                if (kk < nlead && kk < N && n < M && (double)qq[kk] > threshold) {
#pragma unroll
                    for (int j = 0; j < M; ++j)
                        if (j == n) { hv[j] = qq[kk]; hi[j] = (float)(kk + 1); }
                    ++n;
                }
            }
            if (n < M && N > nlead) {   // rare: keep scanning the volume itself
                const float *v = vol + p * N;
                for (int kk = nlead; kk < N && n < M; ++kk) {
                    const float t = v[kk];
                    if ((double)t > threshold) {
#pragma unroll
                        for (int j = 0; j < M; ++j)
                            if (j == n) { hv[j] = t; hi[j] = (float)(kk + 1); }
                        ++n;
                    }
                }
            }
            if (hv[0] > 0) {
                if (M == 4) dfe_sort4(hv, hi); else dfe_sort8(hv, hi);
                if (o.imaxs) o.imaxs[pg] = (long long)hi[0];
#pragma unroll
                for (int j = 1; j < M; ++j) hv[j] += hv[j - 1];
                double acc = 0;
#pragma unroll
                for (int j = 0; j < M; ++j) acc += hv[j];
                o.scores[o.padded ? fo : pg] = (float)acc;
            } else if (o.padded) {
                o.scores[fo] = 0.f;   // pair mode: the caller's buffer is not pre-zeroed (pixels without a hit read 0)
            }
        }
    }
}

// NCH = compile-time bound on ceil(N/64): all of a pixel's loads are issued before the first is consumed
// (one wave keeps up to NCH x 256 B in flight), then min / first-wins arg-min over registers; the
// first-M-above-threshold scan walks the same registers in index order and normally stops after one chunk.
template <int M, int NCH>
__global__ __launch_bounds__(kWavesPerBlock * 64) void flow_tail_kernel(const float *__restrict__ vol, long long Pband, int N,
                                                                       int hWin, int wWin, int middle, double threshold,
                                                                       TailOut o) {
    __shared__ float sh_v[kWavesPerBlock][8];
    __shared__ float sh_i[kWavesPerBlock][8];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const bool want_extract = o.scores != nullptr;
    for (long long p = (long long)blockIdx.x * kWavesPerBlock + w; p < Pband; p += (long long)gridDim.x * kWavesPerBlock) {
        const float *v = vol + p * N;
        float t[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = c * 64 + lane;
            t[c] = (k < N) ? v[k] : 0.f;
        }
        const float centre = (middle > 0) ? v[middle - 1] : 0.f;   // wave-uniform address
        if (lane < 8) { sh_v[w][lane] = 0.f; sh_i[w][lane] = 0.f; }
        float b = t[0];            // lane < N always holds for chunk 0 when N >= 64; guarded below otherwise
        int bi = (lane < N) ? lane : 0x7fffffff;
#pragma unroll
        for (int c = 1; c < NCH; ++c) {
            const int k = c * 64 + lane;
            if (k < N && (bi == 0x7fffffff || t[c] < b)) { b = t[c]; bi = k; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float ob = __shfl_xor(b, off);
            int oi = __shfl_xor(bi, off);
            bool take = (oi != 0x7fffffff) && ((bi == 0x7fffffff) || (ob < b) || (ob == b && oi < bi));
            if (take) { b = ob; bi = oi; }
        }
        if (want_extract) {
            int n = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (n < M && c * 64 < N) {                       // wave-uniform
                    const int k = c * 64 + lane;
                    bool hit = (k < N) && ((double)t[c] > threshold);
                    unsigned long long mask = __ballot(hit);
                    int rank = n + __popcll(mask & ((1ull << lane) - 1ull));
                    if (hit && rank < M) { sh_v[w][rank] = t[c]; sh_i[w][rank] = (float)(k + 1); }
                    n += __popcll(mask);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        if (lane == 0) {
            long long id = (long long)bi + 1;
            if (middle > 0 && b == centre) id = middle;
            const long long pg = o.p_off + p;
            if (o.idx) o.idx[pg] = id;
            if (o.best) o.best[pg] = b;
            const int y = (int)(p / o.Wo) + o.row_off, x = (int)(p % o.Wo);
            const long long fo = (long long)(y + o.pad_t) * o.pitch + x + o.pad_l;
            long long fl = (id - 1) / wWin;
            if (o.fy) o.fy[fo] = (float)(fl - (hWin - 1) / 2);
            if (o.fx) o.fx[fo] = (float)(id - 1 - fl * wWin - (wWin - 1) / 2);
            if (want_extract) {
                float hv[8], hi[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { hv[k] = sh_v[w][k]; hi[k] = sh_i[w][k]; }
                if (hv[0] > 0) {
                    if (M == 4) dfe_sort4(hv, hi); else dfe_sort8(hv, hi);
                    if (o.imaxs) o.imaxs[pg] = (long long)hi[0];
                    for (int k = 1; k < M; ++k) hv[k] += hv[k - 1];
                    double acc = 0;
                    for (int k = 0; k < M; ++k) acc += hv[k];
                    o.scores[o.padded ? fo : pg] = (float)acc;
                } else if (o.padded) {
                    o.scores[fo] = 0.f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

// ---- A12 (i): flow -> depth, replaces the inline-C `radial` of test_opticalflow.lua:143-189 ----
__global__ void flow_to_depth_cart_kernel(const float *__restrict__ flow, int H, int W, float mw, float mh, float infty,
                                          int fix_dot, float *__restrict__ depth, float *__restrict__ conf) {
    const long long P = (long long)H * W;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        int i = (int)(p / W), j = (int)(p - (long long)i * W);
        float py = (float)i - mh, px = (float)j - mw;
        float pn = (float)sqrt((double)(px * px + py * py));     // C `sqrt` on a float promotes to double
        float dy = flow[p], dx = flow[P + p];
        float dn = (float)sqrt((double)(dx * dx + dy * dy));
        float r = 0.f, c = 0.f;
        if (dn >= 0.2f) {
            float q = pn / dn;
            r = q < infty ? q : infty;
            float dot = fix_dot ? (px * dx + py * dy) : (px * dx + dy * dy);   // :181 (sic) unless fix_dot
            if (dot > 0.125f) c = 1.0f;
        } else {
            c = 1.0f;
            r = infty;
        }
        depth[p] = r;
        conf[p] = c;
    }
}

// Pair epilogue: zeroes flow / scores outside the centre-pasted interior (instead of two full-frame memsets before the
// build) and turns the flow into depth in the same pass (same arithmetic as flow_to_depth_cart_kernel).
__global__ void pair_border_depth_kernel(float *__restrict__ flow, float *__restrict__ scores, int H, int W, int pad_t, int pad_l,
                                         int Ho, int Wo, float mw, float mh, float infty, float *__restrict__ depth,
                                         float *__restrict__ conf) {
    const long long P = (long long)H * W;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(p / W), j = (int)(p - (long long)i * W);
        float dy = 0.f, dx = 0.f;
        if (i >= pad_t && i < pad_t + Ho && j >= pad_l && j < pad_l + Wo) {
            dy = flow[p]; dx = flow[P + p];
        } else {
            flow[p] = 0.f; flow[P + p] = 0.f;
            if (scores) scores[p] = 0.f;
        }
        if (depth) pair_depth_px(i, j, dy, dx, mw, mh, infty, &depth[p], &conf[p]);
    }
}

int ring_d(int maxw, int r, int rprev) {   // opticalflow_model_multiscale.lua:94-95: round(maxw*(r-r')/(2r))
    return (int)floor((double)maxw * (r - rprev) / (2.0 * r) + 0.5);
}

int fill_geom(dfe_ctx *ctx, MultiGeom &g, int maxh, int maxw, const int *ratios, int nratios) {
    DFE_REQUIRE(ctx, ratios && nratios >= 1 && nratios <= DFE_MAX_RATIOS, DFE_E_ARG, "nratios=%d not in 1..%d", nratios,
                DFE_MAX_RATIOS);
    DFE_REQUIRE(ctx, maxh > 0 && maxw > 0, DFE_E_ARG, "maxh=%d maxw=%d must be positive", maxh, maxw);
    g.maxh = maxh; g.maxw = maxw; g.nratios = nratios;
    for (int i = 0; i < nratios; ++i) {
        DFE_REQUIRE(ctx, ratios[i] > 0, DFE_E_ARG, "ratios[%d]=%d must be positive", i, ratios[i]);
        g.ratios[i] = ratios[i];
        g.d[i] = i ? ring_d(maxw, ratios[i], ratios[i - 1]) : 0;
    }
    return DFE_OK;
}

int grid_for(long long n, int block) {
    long long b = (n + block - 1) / block;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

void dfe_make_tailout(TailOut *po, int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int Wo, int pitch, int pad_t, int pad_l,
                      int scores_padded, int row_off, const DfePairDepth *pd) {
    TailOut &o = *po;
    o.frame_H = 0; o.frame_W = 0; o.depth = nullptr; o.conf = nullptr; o.mw = o.mh = o.infty = 0.f;
    if (pd) {   // frame mode: this call owns the whole frame (one band), fy / fx / scores are full-frame planes
        o.frame_H = pd->H; o.frame_W = pd->W; o.depth = pd->depth; o.conf = pd->conf;
        o.mw = pd->cx; o.mh = pd->cy; o.infty = (float)((double)pd->W / 2);   // test_opticalflow.lua:148 geometry.wImg/2
    }
    o.idx = (long long *)idx; o.best = best; o.fy = fy; o.fx = fx; o.scores = scores; o.imaxs = (long long *)imaxs;
    o.Wo = Wo; o.pitch = pitch; o.pad_t = pad_t; o.pad_l = pad_l; o.padded = scores_padded;
    o.p_off = (long long)row_off * Wo; o.row_off = row_off;
}

int dfe_flow_finalize(dfe_ctx *ctx, const float2 *part, const float *centre, const float *lead, int nchunks, long long Ptot,
                      const float *vol, double threshold, int rows, int Wo, int hWin, int wWin, int row_off, int64_t *idx, float *best,
                      float *fy, float *fx, float *scores, int64_t *imaxs, int pitch, int pad_t, int pad_l, int scores_padded,
                      const DfePairDepth *pd, const float *rec, int rec_rows) {
    if (rec) nchunks = 1;
    TailOut o;
    dfe_make_tailout(&o, idx, best, fy, fx, scores, imaxs, Wo, pitch, pad_t, pad_l, scores_padded, row_off, pd);
    const long long Pb = (long long)rows * Wo;
    const int N = hWin * wWin;
    const int middle = (wWin + 1) / 2 + wWin * ((hWin + 1) / 2 - 1);
    const int grid = grid_for(pd ? (long long)pd->H * pd->W : Pb, 256);
    if (rec) {
        DFE_REQUIRE(ctx, Pb < (1ll << 31), DFE_E_SHAPE, "flow finalize: %lld pixels in one band", Pb);   // (32-bit pixel arithmetic in the record path)
        DFE_REQUIRE(ctx, !pd || (long long)pd->H * pd->W < (1ll << 31), DFE_E_SHAPE, "flow finalize: frame of %d x %d pixels", pd ? pd->H : 0, pd ? pd->W : 0);
        if (threshold < 0.2)   // extract_output.cpp:83-85
            hipLaunchKernelGGL(flow_finalize_rec_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, vol, Pb, N, hWin, wWin, middle, threshold, o, rec, rec_rows);
        else
            hipLaunchKernelGGL(flow_finalize_rec_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, vol, Pb, N, hWin, wWin, middle, threshold, o, rec, rec_rows);
        DFE_LAUNCH_CHECK(ctx);
        return DFE_OK;
    }
    if (threshold < 0.2)   // extract_output.cpp:83-85
        hipLaunchKernelGGL(flow_finalize_kernel<8>, dim3(grid), dim3(256), 0, ctx->stream, part, centre, lead, nchunks, Ptot, vol, Pb, N,
                           hWin, wWin, middle, threshold, o, rec, rec_rows);
    else
        hipLaunchKernelGGL(flow_finalize_kernel<4>, dim3(grid), dim3(256), 0, ctx->stream, part, centre, lead, nchunks, Ptot, vol, Pb, N,
                           hWin, wWin, middle, threshold, o, rec, rec_rows);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_pair_border_depth(dfe_ctx *ctx, float *flow, float *scores, int H, int W, int pad_t, int pad_l, int Ho, int Wo, float cx,
                          float cy, float *depth, float *conf) {
    const float infty = (float)((double)W / 2);   // test_opticalflow.lua:148 geometry.wImg/2
    hipLaunchKernelGGL(pair_border_depth_kernel, dim3(grid_for((long long)H * W, 256)), dim3(256), 0, ctx->stream, flow, scores, H, W,
                       pad_t, pad_l, Ho, Wo, cx, cy, infty, depth, conf);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

extern "C" {

int dfe_argbest_center(dfe_ctx *ctx, const float *vol, int64_t P, int N, int middle, int take_max, int64_t *idx,
                       float *best) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && N > 0, DFE_E_SHAPE, "dfe_argbest_center: P=%lld N=%d", (long long)P, N);
    DFE_REQUIRE(ctx, middle <= N, DFE_E_ARG, "dfe_argbest_center: middle=%d > N=%d", middle, N);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, vol && idx, DFE_E_ARG, "dfe_argbest_center: NULL tensor");
    int grid = grid_for(P, kWavesPerBlock);
    if (take_max)
        hipLaunchKernelGGL(argbest_kernel<true>, dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, vol, (long long)P, N,
                           middle, (long long *)idx, best);
    else
        hipLaunchKernelGGL(argbest_kernel<false>, dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, vol, (long long)P, N,
                           middle, (long long *)idx, best);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_extract_output(dfe_ctx *ctx, const float *input, int H, int W, int N, float *scores, double threshold,
                       int64_t *imaxs) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0 && N > 0, DFE_E_SHAPE, "dfe_extract_output: H=%d W=%d N=%d", H, W, N);
    long long P = (long long)H * W;
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, input && scores && imaxs, DFE_E_ARG, "dfe_extract_output: NULL tensor");
    int grid = grid_for(P, kWavesPerBlock);
    if (threshold < 0.2)   // extract_output.cpp:83-85
        hipLaunchKernelGGL((extract_kernel<8, false>), dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, input, P, N,
                           threshold, 0.0, (long long *)imaxs, scores, (long long *)nullptr);
    else
        hipLaunchKernelGGL((extract_kernel<4, false>), dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, input, P, N,
                           threshold, 0.0, (long long *)imaxs, scores, (long long *)nullptr);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_extract_output_marginalized(dfe_ctx *ctx, const float *input, int H, int W, int N, double threshold,
                                    double threshold_acc, int64_t *ret, int64_t *retgd) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0 && N > 0, DFE_E_SHAPE, "dfe_extract_output_marginalized: H=%d W=%d N=%d", H, W, N);
    long long P = (long long)H * W;
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, input && ret && retgd, DFE_E_ARG, "dfe_extract_output_marginalized: NULL tensor");
    DFE_HIP(ctx, hipMemsetAsync(retgd, 0, sizeof(int64_t) * P, ctx->stream));   // :166 THLongTensor_zero(retgd)
    int grid = grid_for(P, kWavesPerBlock);
    if (threshold < 0.2)
        hipLaunchKernelGGL((extract_kernel<8, true>), dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, input, P, N,
                           threshold, threshold_acc, (long long *)ret, (float *)nullptr, (long long *)retgd);
    else
        hipLaunchKernelGGL((extract_kernel<4, true>), dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, input, P, N,
                           threshold, threshold_acc, (long long *)ret, (float *)nullptr, (long long *)retgd);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_x2yx(dfe_ctx *ctx, const int64_t *idx, int64_t P, int maxh, int maxw, int64_t *y, int64_t *x) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, P >= 0 && maxh > 0 && maxw > 0, DFE_E_ARG, "dfe_x2yx: P=%lld maxh=%d maxw=%d", (long long)P, maxh, maxw);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, idx && y && x, DFE_E_ARG, "dfe_x2yx: NULL tensor");
    hipLaunchKernelGGL(x2yx_kernel, dim3(grid_for(P, 256)), dim3(256), 0, ctx->stream, (const long long *)idx, (long long)P,
                       maxh, maxw, (long long *)y, (long long *)x);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_x2yx_multi(dfe_ctx *ctx, int maxh, int maxw, const int *ratios, int nratios, const int64_t *idx, int64_t P,
                   int64_t *y, int64_t *x, int compat_c) {
    DFE_ENTER(ctx);
    MultiGeom g;
    int rc = fill_geom(ctx, g, maxh, maxw, ratios, nratios);
    if (rc) return rc;
    DFE_REQUIRE(ctx, P >= 0, DFE_E_ARG, "dfe_x2yx_multi: P=%lld", (long long)P);
    if (P == 0) return DFE_OK;
    DFE_REQUIRE(ctx, idx && y && x, DFE_E_ARG, "dfe_x2yx_multi: NULL tensor");
    if (compat_c) {
        CompatGeom c;
        c.maxh = maxh; c.maxw = maxw; c.nratios = nratios;
        for (int i = 0; i < nratios; ++i) c.ratios[i] = i ? ratios[i - 1] : 0;   // x2yxMulti2.c:15-19 off-by-one
        for (int i = 1; i < nratios; ++i) {
            c.borders[i] = (int)roundf((float)maxw * ((float)c.ratios[i] - (float)c.ratios[i - 1]) / (2.0f * (float)c.ratios[i]));
            c.lengths[i] = 2 * maxw + 2 * (maxh - 2 * c.borders[i]) * c.borders[i];   // :41
        }
        hipLaunchKernelGGL(x2yx_multi_compat_kernel, dim3(grid_for(P, 256)), dim3(256), 0, ctx->stream, c,
                           (const long long *)idx, (long long)P, (long long *)y, (long long *)x);
        DFE_LAUNCH_CHECK(ctx);
        return DFE_OK;
    }
    DFE_HIP(ctx, hipMemsetAsync(ctx->dflag, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(x2yx_multi_kernel, dim3(grid_for(P, 256)), dim3(256), 0, ctx->stream, g, (const long long *)idx,
                       (long long)P, (long long *)y, (long long *)x, ctx->dflag);
    DFE_LAUNCH_CHECK(ctx);
    int flag = 0;
    DFE_HIP(ctx, hipMemcpyAsync(&flag, ctx->dflag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DFE_REQUIRE(ctx, flag == 0, DFE_E_ARG, "dfe_x2yx_multi: a class id is outside 1..%lld",
                (long long)dfe_multi_nclasses(maxh, maxw, ratios, nratios));
    return DFE_OK;
}

int64_t dfe_multi_nclasses(int maxh, int maxw, const int *ratios, int nratios) {
    if (!ratios || nratios < 1 || nratios > DFE_MAX_RATIOS) return -1;
    int64_t n = (int64_t)maxh * maxw;
    for (int i = 1; i < nratios; ++i) {
        int d = ring_d(maxw, ratios[i], ratios[i - 1]);
        n += 2ll * d * maxw + 2ll * (maxh - 2 * d) * d;
    }
    return n;
}

int dfe_x2yx_multi_number(int maxh, int maxw, const int *ratios, int nratios, int64_t id, int64_t *y, int64_t *x) {
    if (!ratios || !y || !x || nratios < 1 || nratios > DFE_MAX_RATIOS) return DFE_E_ARG;
    MultiGeom g;
    g.maxh = maxh; g.maxw = maxw; g.nratios = nratios;
    for (int i = 0; i < nratios; ++i) { g.ratios[i] = ratios[i]; g.d[i] = i ? ring_d(maxw, ratios[i], ratios[i - 1]) : 0; }
    long long oy, ox;
    if (multi_decode(g, id, &oy, &ox)) return DFE_E_ARG;
    *y = oy; *x = ox;
    return DFE_OK;
}

int64_t dfe_yx2x_multi(int maxh, int maxw, const int *ratios, int nratios, double y, double x) {
    // replaces: yx2xMulti opticalflow_model_multiscale.lua:10-52 (host scalar, used for middleIndex)
    if (!ratios || nratios < 1 || nratios > DFE_MAX_RATIOS) return -1;
    x = floor(x + 0.5);
    y = floor(y + 0.5);
    auto is_in = [](double size, double v) { return (v >= -ceil(size / 2) + 1) && (v <= floor(size / 2)); };
    int i = 0;
    double tx = 0, ty = 0;
    for (; i < nratios; ++i)
        if (is_in((double)maxw * ratios[i], x) && is_in((double)maxh * ratios[i], y)) {
            tx = ceil(x / ratios[i]) + ceil(maxw / 2.0);
            ty = ceil(y / ratios[i]) + ceil(maxh / 2.0);
            break;
        }
    if (i >= nratios) return -1;
    long long targetx = (long long)tx, targety = (long long)ty, it;
    if (i == 0) return (targety - 1) * maxw + targetx;
    int d = ring_d(maxw, ratios[i], ratios[i - 1]);
    if (targety <= d) it = (targety - 1) * maxw + targetx;
    else if (targety > maxh - d) it = (long long)d * maxw + 2ll * (maxh - 2 * d) * d + (targety - (maxh - d) - 1) * maxw + targetx;
    else if (targetx <= d) it = (long long)d * maxw + (targety - d - 1) * d + targetx;
    else if (targetx > maxw - d) it = (long long)d * maxw + (long long)(maxh - 2 * d) * d + (targety - d - 1) * d + targetx - (maxw - d);
    else return -1;
    return (long long)maxw * maxh + (long long)(i - 1) * (2ll * d * maxw + 2ll * (maxh - 2 * d) * d) + it;
}

int dfe_flow_tail(dfe_ctx *ctx, const float *vol, int rows, int Wo, int hWin, int wWin, double threshold, int row_off,
                  int64_t *idx, float *best, float *fy, float *fx, float *scores, int64_t *imaxs, int pitch, int pad_t,
                  int pad_l, int scores_padded) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, vol && rows >= 0 && Wo > 0 && hWin > 0 && wWin > 0, DFE_E_ARG, "dfe_flow_tail: bad argument");
    if (rows == 0) return DFE_OK;
    TailOut o;
    o.idx = (long long *)idx; o.best = best; o.fy = fy; o.fx = fx; o.scores = scores; o.imaxs = (long long *)imaxs;
    o.Wo = Wo; o.pitch = pitch; o.pad_t = pad_t; o.pad_l = pad_l; o.padded = scores_padded;
    o.p_off = (long long)row_off * Wo; o.row_off = row_off;
    const long long Pb = (long long)rows * Wo;
    const int N = hWin * wWin;
    const int middle = (wWin + 1) / 2 + wWin * ((hWin + 1) / 2 - 1);   // radial/radial_opticalflow_groundtruth.lua:91
    DFE_REQUIRE(ctx, N <= 64 * 36, DFE_E_UNSUPPORTED, "dfe_flow_tail: window %dx%d has more than 2304 cells", hWin, wWin);
    int grid = grid_for(Pb, kWavesPerBlock);
    const bool m8 = threshold < 0.2;   // extract_output.cpp:83-85
#define DFE_TAIL(MM, NCH) hipLaunchKernelGGL((flow_tail_kernel<MM, NCH>), dim3(grid), dim3(kWavesPerBlock * 64), 0, ctx->stream, vol, Pb, N, hWin, wWin, middle, threshold, o)
    if (N <= 64 * 2) { if (m8) DFE_TAIL(8, 2); else DFE_TAIL(4, 2); }
    else if (N <= 64 * 5) { if (m8) DFE_TAIL(8, 5); else DFE_TAIL(4, 5); }
    else if (N <= 64 * 18) { if (m8) DFE_TAIL(8, 18); else DFE_TAIL(4, 18); }
    else { if (m8) DFE_TAIL(8, 36); else DFE_TAIL(4, 36); }
#undef DFE_TAIL
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_flow_to_depth_cartesian(dfe_ctx *ctx, const float *flow, int H, int W, float cx, float cy, int fix_dot, float *depth,
                                float *conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, H >= 0 && W >= 0, DFE_E_SHAPE, "dfe_flow_to_depth_cartesian: H=%d W=%d", H, W);
    if ((long long)H * W == 0) return DFE_OK;
    DFE_REQUIRE(ctx, flow && depth && conf, DFE_E_ARG, "dfe_flow_to_depth_cartesian: NULL tensor");
    float infty = (float)((double)W / 2);   // test_opticalflow.lua:148 geometry.wImg/2
    hipLaunchKernelGGL(flow_to_depth_cart_kernel, dim3(grid_for((long long)H * W, 256)), dim3(256), 0, ctx->stream, flow, H, W, cx,
                       cy, infty, fix_dot, depth, conf);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

}  // extern "C"
