// ingest.hip -- frames as the camera delivers them: uint8 planes at the C ABI (a quarter of the host-to-device bytes of fp32 frames;
// SURVEY section 8(e) "upload frames as uint8, not fp32"), and image.rgb2y for prepareInput (opticalflow_model.lua:131-151).
//   dfe_u8_to_f32                      [n] uint8 -> float(value) * scale
//   dfe_rgb2y_f32                      [3][H][W] -> [1][H][W], 0.299 R + 0.587 G + 0.114 B accumulated in that order, each product and
//                                      sum rounded separately (image.rgb2y's THTensor cadd chain; `image` is un-vendored: parity unpinned)
//   dfe_flow_depth_pair_u8             dfe_flow_depth_pair_f32 on uint8 frames
//   dfe_multiscale_flow_pair_u8        dfe_multiscale_flow_pair_f32 / _f16 on uint8 frames
// The uint8 entries convert into a per-ctx frame buffer (one pass: 0.9 MB read, 3.7 MB written per VGA frame, ~2 us) and run the fp32
// pipeline on it -- bit-identical to the fp32 entry called on float(frame) * scale.  The conversion is NOT folded into the cost-volume
// kernel: its frame-0 operands are scalar loads of whole fp32 rows (42 SGPRs a row, requested a row ahead), and unpacking bytes on the
// scalar unit would put ~60 more scalar instructions into every row of a kernel that has no scalar register left.
#include "dfe_internal.h"

namespace {

__global__ void u8_to_f32_kernel(const unsigned char *__restrict__ src, long long n, float scale, float *__restrict__ dst) {
    // 4 pixels per thread: one 32-bit load, one 16-B store
    const long long n4 = n >> 2;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
        const unsigned v = reinterpret_cast<const unsigned *>(src)[e];
        reinterpret_cast<float4 *>(dst)[e] = make_float4((float)(v & 255u) * scale, (float)((v >> 8) & 255u) * scale, (float)((v >> 16) & 255u) * scale,
                                                         (float)(v >> 24) * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = (float)src[(n4 << 2) + threadIdx.x] * scale;
}
// BOTH frames of a pair converted in one launch, 16 source bytes per thread and step (the pipelined ingest's conversion)
__global__ __launch_bounds__(256) void u8_fetch_pair_kernel(const unsigned char *__restrict__ s0, const unsigned char *__restrict__ s1, long long n, float scale,
                                                            float *__restrict__ d0, float *__restrict__ d1) {
    const long long n16 = n >> 4;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < 2 * n16; e += (long long)gridDim.x * blockDim.x) {
        const bool second = e >= n16;
        const long long i = second ? e - n16 : e;
        const uint4 v = reinterpret_cast<const uint4 *>(second ? s1 : s0)[i];
        float4 *o = reinterpret_cast<float4 *>(second ? d1 : d0) + 4 * i;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = make_float4((float)(w[q] & 255u) * scale, (float)((w[q] >> 8) & 255u) * scale, (float)((w[q] >> 16) & 255u) * scale, (float)(w[q] >> 24) * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * (n & 15)) {
        const int f = threadIdx.x >= (n & 15), t = threadIdx.x - f * (int)(n & 15);
        (f ? d1 : d0)[(n16 << 4) + t] = (float)(f ? s1 : s0)[(n16 << 4) + t] * scale;
    }
}
__global__ void u8_to_f32_bytes_kernel(const unsigned char *__restrict__ src, long long n, float scale, float *__restrict__ dst) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) dst[e] = (float)src[e] * scale;
}

__global__ void rgb2y_kernel(const float *__restrict__ rgb, long long P, float *__restrict__ y) {
#pragma clang fp contract(off)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
        float v = 0.299f * rgb[e];
        v = v + 0.587f * rgb[P + e];
        v = v + 0.114f * rgb[2 * P + e];
        y[e] = v;
    }
}

// minimum over the leading dimension of an [n][M] view (first row attaining it): tests/time_matching.lua:41-43, where the view is the
// matcher's volume read as [wsize * wsize][W' * H'].  A block = 64 columns x MD_RG row groups: group g walks its n / MD_RG rows (loads of
// 256 contiguous bytes per wave and row, eight in flight), the groups' first minima meet in LDS and are merged in row order with the
// same strict '<' -- the row the sequential loop `best = in[0]; if (v < best) ...` ends on, NaN handling included (a NaN in row 0
// stays, a NaN elsewhere never wins).  (First version: one thread per column, 256 dependent loads each: 101 us for the script's 46 MB.)
constexpr int MD_RG = 16;
__global__ __launch_bounds__(64 * MD_RG) void min_dim0_kernel(const float *__restrict__ in, int n, long long M, float *__restrict__ val,
                                                              long long *__restrict__ idx) {
    __shared__ float sv[MD_RG][64];
    __shared__ int si[MD_RG][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long long m = (long long)blockIdx.x * 64 + tx;
    const int per = (n + MD_RG - 1) / MD_RG;
    const int r0 = g * per, r1 = min(n, r0 + per);
    float best = __int_as_float(0x7f800000);
    int bi = -1;
    if (m < M) {
        const float *p = in + m;
        int r = r0;
        if (g == 0 && r < r1) { best = p[0]; bi = 0; r = 1; }
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long long)(r + u) * M];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] < best) { best = v[u]; bi = r + u; }
        }
        for (; r < r1; ++r) {
            const float v = p[(long long)r * M];
            if (v < best) { best = v; bi = r; }
        }
    }
    sv[g][tx] = best;
    si[g][tx] = bi;
    __syncthreads();
    if (g == 0 && m < M) {
#pragma unroll
        for (int h = 1; h < MD_RG; ++h) {
            const float v = sv[h][tx];
            if (si[h][tx] >= 0 && v < best) { best = v; bi = si[h][tx]; }
        }
        if (val) val[m] = best;
        if (idx) idx[m] = (long long)bi + 1;
    }
}

int in_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

int launch_u8(dfe_ctx *ctx, const unsigned char *src, long long n, float scale, float *dst) {
    if ((((uintptr_t)src) & 3) == 0 && (((uintptr_t)dst) & 15) == 0)
        hipLaunchKernelGGL(u8_to_f32_kernel, dim3(in_grid(n >> 2)), dim3(256), 0, ctx->stream, src, n, scale, dst);
    else
        hipLaunchKernelGGL(u8_to_f32_bytes_kernel, dim3(in_grid(n)), dim3(256), 0, ctx->stream, src, n, scale, dst);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

// the per-ctx fp32 copy of a uint8 frame pair (grow-only, next to the scratch arena that the pipelines themselves use)
int ingest_pair(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, long long n, float scale, float **f0, float **f1) {
    const size_t bytes = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    if (2 * bytes > ctx->ingest_bytes) {
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ingest) DFE_HIP(ctx, hipFree(ctx->ingest));
        ctx->ingest = nullptr;
        ctx->ingest_bytes = 0;
        hipError_t e = hipMalloc(&ctx->ingest, 2 * bytes);
        if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "frame buffer hipMalloc(%zu): %s", 2 * bytes, hipGetErrorString(e));
        ctx->ingest_bytes = 2 * bytes;
    }
    *f0 = (float *)ctx->ingest;
    *f1 = (float *)((char *)ctx->ingest + bytes);
    DfeStageScope st(ctx, DFE_STAGE_LOAD);
    int rc = launch_u8(ctx, I0, n, scale, *f0);
    if (rc) return rc;
    return launch_u8(ctx, I1, n, scale, *f1);
}

}  // namespace

extern "C" {

int dfe_u8_to_f32(dfe_ctx *ctx, const uint8_t *src, int64_t n, float scale, float *dst) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, n >= 0, DFE_E_SHAPE, "dfe_u8_to_f32: n=%lld", (long long)n);
    if (n == 0) return DFE_OK;
    DFE_REQUIRE(ctx, src && dst, DFE_E_ARG, "dfe_u8_to_f32: NULL tensor");
    return launch_u8(ctx, src, n, scale, dst);
}

int dfe_min_dim0_f32(dfe_ctx *ctx, const float *in, int n, int64_t M, float *val, int64_t *idx) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, in && (val || idx), DFE_E_ARG, "dfe_min_dim0_f32: NULL tensor");
    DFE_REQUIRE(ctx, n > 0 && M > 0, DFE_E_SHAPE, "dfe_min_dim0_f32: %d x %lld", n, (long long)M);
    DFE_REQUIRE(ctx, (M + 63) / 64 < (1ll << 31), DFE_E_SHAPE, "dfe_min_dim0_f32: %lld columns", (long long)M);
    hipLaunchKernelGGL(min_dim0_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64 * MD_RG), 0, ctx->stream, in, n, (long long)M, val, (long long *)idx);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_rgb2y_f32(dfe_ctx *ctx, const float *rgb, int H, int W, float *y) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, rgb && y, DFE_E_ARG, "dfe_rgb2y_f32: NULL tensor");
    DFE_REQUIRE(ctx, H > 0 && W > 0, DFE_E_SHAPE, "dfe_rgb2y_f32: %dx%d", H, W);
    hipLaunchKernelGGL(rgb2y_kernel, dim3(in_grid((long long)H * W)), dim3(256), 0, ctx->stream, rgb, (long long)H * W, y);
    DFE_LAUNCH_CHECK(ctx);
    return DFE_OK;
}

int dfe_flow_depth_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int hWin, int wWin, float foe_x,
                           float foe_y, double extract_threshold, float scale, float *flow, float *scores, float *depth, float *depth_conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1, DFE_E_ARG, "dfe_flow_depth_pair_u8: NULL frame");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && scale > 0, DFE_E_ARG, "dfe_flow_depth_pair_u8: C=%d %dx%d scale=%g", C, H, W, (double)scale);
    float *f0 = nullptr, *f1 = nullptr;
    int rc = ingest_pair(ctx, I0, I1, (long long)C * H * W, scale, &f0, &f1);
    if (rc) return rc;
    return dfe_flow_depth_pair_f32(ctx, f0, f1, C, H, W, k, hWin, wWin, foe_x, foe_y, extract_threshold, flow, scores, depth, depth_conf);
}

// ---- pipelined ingest: host frames of pair i+1 travel while pair i computes -------------------------------------------------------
// What the measurements on this stack say (tools/pipe_probe.py, pipe_probe2.py; profiles/r05_n_*):
//   * asynchronous copies from pinned host memory on a second stream run on the copy engines BESIDE the sweep: 258 us per step
//     against 245 resident (+ the conversion);
//   * but a copy that has to wait for an event of the COMPUTE queue (hipStreamWaitEvent on the copy stream) makes hipMemcpyAsync block
//     the HOST for a whole step: 160-360 us per submit, the loop slower than the serial one;
//   * an upload KERNEL beside the sweep does not overlap at all: the sweep's 16 waves of 128 registers fill every SIMD's register file,
//     the upload's waves are only placed when a sweep block leaves (249 us for a 40-us transfer, the next step 14 us late).
// So: copy-engine transfers into one of THREE uint8 slots, no event wait on the copy stream; a slot's reuse is ordered on the host
// (hipEventSynchronize on its consumed event -- recorded two pairs earlier, complete long before in a loop that stays one pair ahead);
// the compute stream waits for the slot's copied event (a barrier packet) and converts both frames in one launch.
int dfe_ingest_submit_u8(dfe_ctx *ctx, const uint8_t *hI0, const uint8_t *hI1, int64_t nbytes, int *slot_out) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, hI0 && hI1 && slot_out && nbytes > 0, DFE_E_ARG, "dfe_ingest_submit_u8: NULL frame or %lld bytes", (long long)nbytes);
    if (!ctx->copy_stream) {
        DFE_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < DFE_NSLOT; ++i) {
            DFE_HIP(ctx, hipEventCreateWithFlags(&ctx->copied[i], hipEventDisableTiming));
            DFE_HIP(ctx, hipEventCreateWithFlags(&ctx->consumed[i], hipEventDisableTiming));
        }
    }
    if ((size_t)nbytes > ctx->slot_bytes) {                      // (grow-only; sizes settle with the first pair)
        DFE_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < DFE_NSLOT; ++i) {
            if (ctx->slot[i]) DFE_HIP(ctx, hipFree(ctx->slot[i]));
            ctx->slot[i] = nullptr;
            ctx->slot_used[i] = false;
        }
        ctx->slot_bytes = 0;
        const size_t fb = ((size_t)nbytes + 255) / 256 * 256;
        for (int i = 0; i < DFE_NSLOT; ++i) {
            hipError_t e = hipMalloc(&ctx->slot[i], 2 * fb);
            if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "ingest slot hipMalloc(%zu): %s", 2 * fb, hipGetErrorString(e));
        }
        ctx->slot_bytes = fb;
    }
    const int s = ctx->slot_next;
    ctx->slot_next = (s + 1) % DFE_NSLOT;
    // the slot's previous pair must have been consumed before its bytes are overwritten: ordered on the host (see above)
    if (ctx->slot_used[s]) DFE_HIP(ctx, hipEventSynchronize(ctx->consumed[s]));
    DFE_HIP(ctx, hipMemcpyAsync(ctx->slot[s], hI0, (size_t)nbytes, hipMemcpyHostToDevice, ctx->copy_stream));
    DFE_HIP(ctx, hipMemcpyAsync((char *)ctx->slot[s] + ctx->slot_bytes, hI1, (size_t)nbytes, hipMemcpyHostToDevice, ctx->copy_stream));
    DFE_HIP(ctx, hipEventRecord(ctx->copied[s], ctx->copy_stream));
    ctx->slot_used[s] = true;
    *slot_out = s;
    return DFE_OK;
}

int dfe_flow_depth_pair_u8_slot(dfe_ctx *ctx, int slot, int C, int H, int W, int k, int hWin, int wWin, float foe_x, float foe_y, double extract_threshold,
                                float scale, float *flow, float *scores, float *depth, float *depth_conf) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, slot >= 0 && slot < DFE_NSLOT && ctx->slot_used[slot], DFE_E_ARG, "dfe_flow_depth_pair_u8_slot: slot %d holds no submitted pair", slot);
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && scale > 0 && (size_t)C * H * W <= ctx->slot_bytes, DFE_E_SHAPE,
                "dfe_flow_depth_pair_u8_slot: %dx%dx%d bytes (scale %g), the slot holds %zu", C, H, W, (double)scale, ctx->slot_bytes);
    DFE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->copied[slot], 0));
    const uint8_t *d0 = (const uint8_t *)ctx->slot[slot], *d1 = d0 + ctx->slot_bytes;
    // both frames converted by ONE launch (the serial entry's two conversion launches are 12 us of a 245-us step)
    const long long n = (long long)C * H * W;
    const size_t bytes = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    if (2 * bytes > ctx->ingest_bytes) {
        DFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ingest) DFE_HIP(ctx, hipFree(ctx->ingest));
        ctx->ingest = nullptr;
        ctx->ingest_bytes = 0;
        hipError_t e = hipMalloc(&ctx->ingest, 2 * bytes);
        if (e != hipSuccess) return dfe_fail(ctx, DFE_E_ALLOC, "frame buffer hipMalloc(%zu): %s", 2 * bytes, hipGetErrorString(e));
        ctx->ingest_bytes = 2 * bytes;
    }
    float *f0 = (float *)ctx->ingest, *f1 = (float *)((char *)ctx->ingest + bytes);
    {
        DfeStageScope st(ctx, DFE_STAGE_LOAD);
        hipLaunchKernelGGL(u8_fetch_pair_kernel, dim3(in_grid(2 * (n >> 4))), dim3(256), 0, ctx->stream, d0, d1, n, scale, f0, f1);
        DFE_LAUNCH_CHECK(ctx);
    }
    DFE_HIP(ctx, hipEventRecord(ctx->consumed[slot], ctx->stream));   // (the conversion is the slot's only reader)
    return dfe_flow_depth_pair_f32(ctx, f0, f1, C, H, W, k, hWin, wWin, foe_x, foe_y, extract_threshold, flow, scores, depth, depth_conf);
}

int dfe_multiscale_flow_pair_u8(dfe_ctx *ctx, const uint8_t *I0, const uint8_t *I1, int C, int H, int W, int k, int maxh, int maxw,
                                const int *ratios, int nratios, float scale, float f16_scale, float *flow, int64_t *idx) {
    DFE_ENTER(ctx);
    DFE_REQUIRE(ctx, I0 && I1, DFE_E_ARG, "dfe_multiscale_flow_pair_u8: NULL frame");
    DFE_REQUIRE(ctx, C > 0 && H > 0 && W > 0 && scale > 0, DFE_E_ARG, "dfe_multiscale_flow_pair_u8: C=%d %dx%d scale=%g", C, H, W, (double)scale);
    DFE_REQUIRE(ctx, f16_scale >= 0.f && f16_scale < INFINITY, DFE_E_ARG, "dfe_multiscale_flow_pair_u8: f16_scale=%g", (double)f16_scale);
    // no conversion pass: the pyramid's preparation kernels are the only readers of the frames and take the bytes as they are
    // (float(byte) * scale at the load: the bits of dfe_u8_to_f32 followed by the fp32 entry)
    return dfe_multiscale_flow_pair_bytes(ctx, I0, I1, C, H, W, k, maxh, maxw, ratios, nratios, scale, f16_scale, flow, idx);
}

}  // extern "C"
