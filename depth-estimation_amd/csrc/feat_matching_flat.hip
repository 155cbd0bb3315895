// feat_matching_flat.hip -- nn.SpatialMatching on K-plane FEATURE maps with 16- and 17-wide windows: the matcher behind every trained
// single-scale model (version2/network.lua:30 SpatialMatching(17, 17) on 32 planes; opticalflow_model.lua:93; tests/time_matching.lua:18
// SpatialMatching(16, 16) on 10 planes).
//   out[y][x][dy][dx] = sum_k (in1[k][y][x] - in2[k][y+dy][x+dx])^2,  accumulated over k in order with a separate multiply and add --
// the arithmetic of the CPU restatement and of every other matcher here: results are bit-identical.
//
// Round 3's row kernel (feat_matching.hip: one block = one output row x 256 columns, wave <-> dy, lane <-> 4 pixels x the window row's
// cells) had the right register tile -- 0.09 LDS reads per output and plane -- and lost its time around it (profiles/r04_a_fm_*.txt:
// 1.57e8 vector instructions for a minimum of 1.12e8, 45 % of all wave cycles waiting): a third of the lanes idle at 625 columns
// (3 x 256), a staging loop of per-element index arithmetic, __syncthreads() draining the store queue in front of every barrier of
// the copy-out, a new block (and an exposed memory round trip) per tile.  This kernel keeps the register tile and changes the rest:
//   * FLAT TILES: the pixels of the frame are taken in row-major order in groups of PX = 4, a tile is 64 consecutive groups (lane <->
//     group) and may run over the end of an image row: 1141 tiles instead of 1395 at 625 x 465, no idle lanes.  A tile touches at most
//     two output rows (frames at least 64 groups wide); the in2 rows it needs sit in LDS as two column pieces per row (A: the end of
//     row y, B: the start of row y+1), and a lane's window row is at one precomputed offset in them.
//   * PERSISTENT blocks (one per CU) walk a contiguous range of tiles: the output of consecutive tiles is one contiguous run of memory.
//   * STAGING by LDS-DMA (global_load_lds_dword): wave <-> tile row, 5 loads per wave and plane with per-lane source offsets that are
//     computed once per tile; no staging registers, no LDS write pass; plane k+1 is requested before the arithmetic of plane k.
//   * ONE LDS-only barrier per plane (s_waitcnt lgkmcnt(0); s_barrier) -- never a vmcnt(0) in front of a barrier except for the
//     plane that was requested a whole plane of arithmetic earlier.
//   * COPY-OUT through a double-buffered LDS image of 32 pixels' windows (a contiguous run of the output, congruent to the global float
//     index mod 32): whole 128-B lines, fire-and-forget, one LDS-only barrier per phase; the stores of a tile drain behind the
//     arithmetic of the next.
//   * 17 WINDOW ROWS on 16 waves (version2's 17 x 17): wave w sweeps window row w for its lanes' 4 pixels (68 accumulators) and, as an
//     EXTRA task, row 16 for the 16 pixels of lanes 4w .. 4w+3 (lane <-> (pixel, 4-cell group): 5 accumulators, 15 vector
//     instructions a plane next to the main task's 204) -- the seventeenth row costs 7 % instead of a second round.
#include "dfe_internal.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int FF_PX = 4;                     // pixels per lane
constexpr int FF_GROUPS = 64;                // groups (lanes) per tile
#ifndef FF_ST_FLAGS
#define FF_ST_FLAGS " nt"   // cache policy of the copy-out stores
#endif
#ifndef FF_DEEP
#define FF_DEEP 1     // 1: requests stay in flight for TWO planes (counted vmcnt), operands are read behind the barrier
#endif
#ifndef FF_BATCH
#define FF_BATCH 8                           // (pixel, cell) pairs whose differences / squares / adds are issued as three groups
#endif
template <int I, int N, class F> __device__ __forceinline__ void static_for_q(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_q<I + 1, N>(f);
    }
}

typedef float ff_f4 __attribute__((ext_vector_type(4)));
// LDS pointers keep their address space (32-bit arithmetic, ds_ instructions with immediate offsets): through generic pointers the
// compiler carried 64-bit lane addresses across the plane loop
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(3))) ff_f4 lds_f4;
typedef __attribute__((address_space(3))) void *ff_lds_ptr;
typedef const __attribute__((address_space(1))) void *ff_gbl_ptr;

struct FfArgs {
    const float *in1, *in2;
    float *out;
    int K, H1, W1, maxh, H2, W2;
    int G;            // groups per image row = ceil(W1 / PX)
    int NG;           // H1 * G
    int ntiles;       // ceil(NG / 64)
    long long *idx;   // ARGMIN: [H1][W1] 1-based window index of the first minimum, or NULL
    float *xflow, *yflow;   // ARGMIN: [H1][W1] decoded displacement (index % wWin - lWin, index / wWin - tWin), or NULL
    int lWin, tWin;
    int S;            // blocks per tile (1, or 2: the window's rows dealt to two half blocks that share a CU)
    int nd;           // window rows of a block = maxh / S
    int pitch1;       // floats between rows of in1 (W1 for a contiguous map; more for a narrowed view: opticalflow_model.lua:147-149)
    long long plane1; // floats between planes of in1
    // SOFT (FF_SOFT): getModel's Minus -> SoftMax over the window and processOutput on it (opticalflow_model.lua:94-109,153-252)
    int middle;       // 1-based index of the window's centre cell: ceil(maxw/2) + maxw * (ceil(maxh/2) - 1)
    int use_thr;      // 0: arg-max with the centre tie-break, confidences 1;  1: extractOutput(p, 0.11) and scores > thr
    float thr;
    int wFull, ho, wo;                 // the centre paste: full / full_conf are [..][hFull][wFull], the output's pixel (y, x) lands at (ho + y, wo + x)
    long long fullplane;               // hFull * wFull
    float *full, *full_conf, *scores;  // [2][hFull][wFull] (plane 0 = y), [hFull][wFull], [H1][W1]; each may be NULL (idx above: [H1][W1])
};
enum { FF_VOLUME = 0, FF_ARGMIN = 1, FF_SOFT = 2 };
#ifndef DFE_FF_SOFT_FAST
#define DFE_FF_SOFT_FAST 1
#endif
constexpr float FF_TIE = 1e-6f;   // FF_SOFT without a threshold: cells this close to a window's minimum may share its maximal probability

template <int MW> struct FfGeom {
    static constexpr int PITCH = (64 * FF_PX + 2 * (MW - 1) + 8 + 3) / 4 * 4;   // floats per LDS tile row (piece A | piece B)
    static constexpr int NLOAD = (PITCH + 63) / 64;                             // LDS-DMA loads per tile row
    static constexpr int NB4 = (FF_PX + MW - 1 + 3) / 4;                        // b128 reads of a lane's window row
};

extern __shared__ __attribute__((aligned(128))) float ff_smem[];

// floats per window slot of the copy-out image: a multiple of 4 with room for the alignment shift (<= 3 floats) of a window whose
// place in the output is not 16-B aligned, and not a multiple of 8 (the 64 lanes' 16-B writes then fall on different banks)
__host__ __device__ inline int ff_wnp(int WN) {
    int w = (WN + 3 + 3) & ~3;
    return (w & 7) ? w : w + 4;
}

// One LDS-DMA load of 64 floats: lane l fetches the float at sbase + voff (bytes; its own offset) into LDS at lds_dst + 4 l.
// Written as assembly for the addressing form: scalar base + 32-bit vector offset -- through the builtin the compiler built a 64-bit
// address per lane and load (two more registers live at every request, in a kernel that has none to spare).  M0 (the destination
// base) is the compiler's: saved and restored around the instruction (cdna_hip_programming.md, LDS-DMA recipe).
__device__ __forceinline__ void ff_glds16(unsigned voff, const void *sbase, const lds_f *lds_dst) {   // 16 bytes per lane: LDS at lds_dst + 16 l
    const unsigned la = (unsigned)(size_t)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(la) : "memory");
}
__device__ __forceinline__ void ff_glds4(unsigned voff, const void *sbase, const lds_f *lds_dst) {
    const unsigned la = (unsigned)(size_t)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(la) : "memory");
}

// Both requests of a tile row (PITCH floats: 64 lanes' 16-B pieces + a tail of NT lanes) as ONE assembly block: M0 saved and restored
// once, the tail request under an execution mask set and restored by two scalar instructions (as C++: save / move / restore of M0 per
// request and a saveexec / branch / restore around the second one -- twelve scalar instructions where seven do; every instruction of
// this kernel's plane loop costs 0.24 % of it).  Called with all 64 lanes active.
template <int NT> __device__ __forceinline__ void ff_glds_row(unsigned voff0, unsigned voff1, const void *sbase, const lds_f *lds_dst) {
    const unsigned la = (unsigned)(size_t)lds_dst;
    unsigned keep;
    unsigned long long ex;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                 "s_add_u32 m0, m0, 0x400\n\ts_and_saveexec_b64 %1, %6\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ex) : "v"(voff0), "v"(voff1), "s"(la), "s"(sbase), "n"((1u << NT) - 1) : "memory", "scc");
}

// maxh <= 16: waves 0 .. maxh-1 each sweep one window row.  EXTRA (maxh == 17): 16 waves, row 16 as the extra task.
// ARGMIN: no volume -- the arithmetic's result goes through the first-minimum decode of version2/test.lua:45-51 (FfArgs::idx / xflow /
// yflow) instead of the copy-out: the volume of a 17 x 17 window on VGA features is 316 MB that the one-call model never reads back.
template <int MW, bool EXTRA, int MODE = FF_VOLUME>
__global__ __launch_bounds__(1024) void feat_matching_flat_kernel(FfArgs p) {
#pragma clang fp contract(off)
    constexpr bool ARGMIN = MODE == FF_ARGMIN, SOFT = MODE == FF_SOFT;
    constexpr int PX = FF_PX;
    constexpr int PITCH = FfGeom<MW>::PITCH, NLOAD = FfGeom<MW>::NLOAD, NB4 = FfGeom<MW>::NB4;
    const int nd = EXTRA ? 16 : p.nd;                      // window rows of this block (a half tile: maxh / 2)
    const int NW = nd;                                     // waves of the block = blockDim.x / 64
    const int nrows = EXTRA ? p.maxh + 1 : nd + 1;         // in2 rows of a (half) tile (EXTRA: 18 -- left a run-time value: as a constant it cost 160 B more scratch)
    const int WN = p.maxh * MW;                            // floats per window
    const int WL = (EXTRA ? 17 : nd) * MW;                 // ... of them this block's
    // LDS: 3 x ([nrows][PITCH] tile | [256] in1 piece) | [64] lane offsets (extra task) | [64][WNP] image.  A buffer is addressed by its
    // float offset from `tile`; the plane loop carries the three offsets and rotates them (no index -> address arithmetic per plane)
    lds_f *tile = (lds_f *)ff_smem;
    const int aoff = nrows * PITCH, BUFSZ = aoff + 64 * PX;  // the in1 piece of a buffer, floats per buffer
    lds_i *gtab = (lds_i *)(tile + 3 * BUFSZ);
    lds_f *img = (lds_f *)(gtab + 64);                    // [64][WNP]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dy = wave;
    const long long plane1 = p.plane1, plane2 = (long long)p.H2 * p.W2;
    const int pgap = p.pitch1 - p.W1;                       // floats between the end of an in1 row and the start of the next
    // Tiles of this block: tile (round r, virtual block vb) = r * gridDim.x + vb, where the virtual index puts the 32 blocks of an XCD
    // (the hardware deals linear block ids round-robin to the 8 XCDs) on 32 CONSECUTIVE tiles -- about 13 output rows of the frame,
    // all planes of whose in2 rows (2.4 MB) stay in that XCD's 4-MB L2 while its CUs, in step with each other, walk through the planes.
    // (First version: a contiguous range of tiles per block -- the tiles in flight were spread over the whole frame, 76 % of the
    //  L2 requests missed, 790 MB were fetched for 39 MB of in2 and every plane waited out a memory round trip: r04_g, r04_j.)
    const int nbx = gridDim.x;
    int vb = (int)blockIdx.x;
    const int S = (MODE == FF_VOLUME && !EXTRA) ? p.S : 1, lgS = S == 4 ? 2 : S == 2 ? 1 : 0;   // (the arg-min form and the 17-row window: always whole tiles)
    if (S > 1 && !(nbx & (8 * S - 1))) {
        // two blocks per CU: the dispatcher fills every CU once (blocks 0 .. nbx/2 - 1), then a second time (blocks b and b + nbx/2
        // share a CU: tools/ubench/cuid.hip, profiles/r04_ah_cu_residency.txt).  The second set takes the upper half of every round's
        // tiles -- of a last, partial round none, so that its half tiles run one to a CU, 1.5 x faster than two to a CU.
        // (Tried: the second set starting 0.25 .. 1.5 tiles late, to keep the two blocks of a CU in different phases -- no gain, and
        //  a loss from 8 sleeps on: profiles/r04_ag_fm_split_sweep.txt.  The gain of the split is the last round: 4 full rounds take
        //  0.213 ms either way, r04_ag_fm_split_rounds.txt.)
        // (S == 4: four sets, blocks b, b + nbx/4, ... on one CU.)
        const int hb = nbx >> lgS, set = (int)blockIdx.x / hb, b = (int)blockIdx.x - set * hb;
        vb = set * hb + (b & 7) * (hb >> 3) + (b >> 3);
    } else if (!(nbx & 7)) {
        vb = (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3);
    }
    // (Also tried: CU j of an XCD starting j x 2048 cycles late, so that the 32 CUs of an XCD -- which run in step -- do not copy out
    //  at the same moment through the XCD's one fabric port: 4 full rounds 0.213 -> 0.232 ms, i.e. the delay itself minus ~3 us a
    //  round; the copy-out's 21 kilocycles a tile are not a shared-port effect either.  profiles/r04_aj_fm_cu_stagger.txt)

    // (Tried: the blocks of XCD x starting x * 1 .. 8 us later, so that the XCDs' copy-out bursts come one after the other -- no gain,
    //  profiles/r04_u: the bursts are not what the copy-out waits for.)
    // Half tiles (S == 2): block vt handles window rows dy0 .. dy0 + nd - 1 of tile vt / 2, TWO blocks of nd waves share a CU, each
    // with its own barriers.  1141 tiles on 256 CUs are 4.46 rounds that cost one block per CU 5; as half tiles the last round's
    // blocks have their CU to themselves and finish in 0.65 of a round (K = 32, 625 x 465: 0.265 -> 0.232 ms, K = 10: 0.134 -> 0.101).
    // The price: the in1 piece and one in2 row are staged by both halves.
    const bool halves = S > 1;
    const int nvt = p.ntiles << lgS;
    for (int vt = vb; vt < nvt; vt += nbx) {
        const int t = vt >> lgS;
        const int dy0 = (vt & (S - 1)) * nd;
        // ---- tile geometry (wave-uniform scalars, then per-lane offsets) ----
        const int g0 = t * FF_GROUPS;
        const int y_first = g0 / p.G, xgA0 = g0 - y_first * p.G, xA0 = xgA0 * PX;
        const int nA = min(FF_GROUPS, p.G - xgA0);                        // groups of the tile in row y_first
        const int LA = min(p.W2 - xA0, FF_GROUPS * PX + MW - 1);          // columns of piece A
        const int LA_pad = (LA + 3) & ~3;
        // (per-lane geometry is derived from a laundered lane id where it is used -- here for the LDS offsets, again in the copy-out --
        //  so that nothing but boff and the five source offsets is live across the plane loop: the 17 x 17 kernel has no register to spare)
        auto lane_fresh = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
        int boff;
        unsigned voff16[2];     // byte offsets (from the start of an in2 row) of my 16-B pieces of a tile row: piece 4 (l + 64 m) .. + 3
        {
            const int l = lane_fresh();
            const int isB = l >= nA ? 1 : 0;
            const int x = isB ? (l - nA) * PX : xA0 + l * PX;             // first pixel column of my group
            const int xin = isB ? LA_pad + x : x - xA0;                   // ... inside its tile row (piece A | piece B)
            boff = (dy + isB) * PITCH + xin;                              // my window row in the LDS tile (floats)
            if (wave == 0) {
                gtab[l] = isB * PITCH + xin;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = 4 * (l + 64 * m);                           // (LA_pad is a multiple of 4: a piece of 4 floats lies in A or in B)
                voff16[m] = 4u * (unsigned)(c < LA_pad ? xA0 + c : c - LA_pad);
            }
        }
        // Staging by LDS-DMA, wave <-> tile row (the rows behind the waves and the in1 piece are dealt to the first waves): 16 bytes per
        // lane -- two instructions for a tile row of 296 floats, one for the in1 piece.  (First version: dword loads, five per row.)
        // A 16-B piece is not clamped at the end of its image row: it may read up to 3 floats of the NEXT row, which no valid pixel
        // uses.  Where that would leave the buffer -- the last row of the last plane -- the row is staged float by float, clamped.
        // Everything a request needs is a per-tile scalar (row pointers of plane 0, LDS offsets) + k * plane size: the barrier puts the
        // sixteen waves in phase, so scalar work at the top of a plane is not hidden behind other waves' arithmetic -- the first version
        // recomputed its row pointers with 64-bit multiplies per plane, ~70 scalar instructions a wave, and a plane took 2900 cycles
        // where arithmetic and requests by themselves take 2100 and 1650 (r04_n).
        const int padpx = p.G * PX - p.W1;
        const int j1 = wave + NW;                                         // my second task: a row, the in1 piece (j1 == nrows) or none
        const int yr0 = min(y_first + dy0 + wave, p.H2 - 1), yr1 = min(y_first + dy0 + j1, p.H2 - 1);
        const char *rp0 = reinterpret_cast<const char *>(p.in2) + (long long)yr0 * p.W2 * 4;
        const char *rp1 = j1 < nrows ? reinterpret_cast<const char *>(p.in2) + (long long)yr1 * p.W2 * 4
                                     : reinterpret_cast<const char *>(p.in1) + ((long long)y_first * p.pitch1 + xA0) * 4;
        const long long pstep0 = plane2 * 4, pstep1 = (j1 < nrows ? plane2 : plane1) * 4;
        const bool last0 = yr0 == p.H2 - 1, last1 = j1 < nrows ? yr1 == p.H2 - 1 : y_first + 2 >= p.H1;   // may end at the end of the buffer (plane K-1)
        unsigned voff_a = 0;                                              // my 16-B piece of the in1 slots
        if (j1 == nrows) { const int l = lane_fresh(); voff_a = 4u * (unsigned)(4 * l - (l >= nA ? padpx - pgap : 0)); }
        auto stage_row = [&](const char *rowp, lds_f *dst, bool careful) __attribute__((always_inline)) {
            if (careful) {                                                // (wave-uniform, once per frame)
#pragma unroll
                for (int m = 0; m < NLOAD; ++m) {
                    const int c = lane_fresh() + 64 * m;
                    if (c < PITCH) ff_glds4(4u * (unsigned)(c < LA_pad ? min(xA0 + c, p.W2 - 1) : min(c - LA_pad, p.W2 - 1)), rowp, dst + 64 * m);
                }
            } else {
                ff_glds16(voff16[0], rowp, dst);
                if (4 * (lane + 64) < PITCH) ff_glds16(voff16[1], rowp, dst + 256);
            }
        };
        // kslow: the plane (K - 1) in which this wave's rows may end at the end of the buffer, or -1 -- ONE comparison per plane decides
        // between the clamped float-by-float path and the plain one; role: what my second task is
        const int kslow = (last0 || last1) ? p.K - 1 : -1;
        const int role = j1 < nrows ? 2 : j1 == nrows ? 1 : 0;
        const bool is_row1 = role == 2, is_piece = role == 1;
        constexpr int NTAIL = (PITCH - 256) / 4;                          // lanes of a row's second request
        static_assert(NTAIL > 0 && NTAIL < 64 && PITCH % 4 == 0, "a tile row is 64 + NTAIL 16-B pieces");
        auto stage = [&](int k, lds_f *tb, const bool FAST) __attribute__((always_inline)) {   // tb: the buffer; FAST: k != kslow is known
            const char *r0 = rp0 + k * pstep0, *r1 = rp1 + k * pstep1;
            if (FAST || k != kslow) {                                     // (flat: two wave-uniform tests around straight-line requests)
                // (the second task's requests first: see the counted wait; two independent tests -- nested, the compiler spent eight
                //  scalar instructions on the role of a wave that has no second task)
                if (is_row1) ff_glds_row<NTAIL>(voff16[0], voff16[1], r1, tb + j1 * PITCH);
                if (is_piece) ff_glds16(voff_a, r1, tb + aoff);
                ff_glds_row<NTAIL>(voff16[0], voff16[1], r0, tb + wave * PITCH);
            } else {
                stage_row(r0, tb + wave * PITCH, last0);
                if (role == 2) stage_row(r1, tb + j1 * PITCH, last1);
                if (role == 1) {
                    // the in1 values of the tile's 256 pixel slots (group l, pixel q at slot 4 l + q): row y_first from xA0, then row
                    // y_first + 1 from column 0 -- linear in memory except for the W1p - W1 padding slots at the end of row y_first
                    if (last1) {
                        const int lim = (p.H1 - 1 - y_first) * p.pitch1 + p.W1 - 1 - xA0;   // the map's last element, from the piece's first
#pragma unroll
                        for (int m = 0; m < PX; ++m) {
                            const int e = lane_fresh() + 64 * m;
                            ff_glds4(4u * (unsigned)min(e - (e >= nA * PX ? padpx - pgap : 0), lim), r1, tb + aoff + 64 * m);
                        }
                    } else {
                        ff_glds16(voff_a, r1, tb + aoff);
                    }
                }
            }
        };

        float acc[PX][MW];
#pragma unroll
        for (int q = 0; q < PX; ++q)
#pragma unroll
            for (int d = 0; d < MW; ++d) acc[q][d] = 0.f;
        // extra task (EXTRA): lane e <-> pixel e >> 2 of the 16 pixels of groups 4 wave .. 4 wave + 3, cells 4 (e & 3) .. + 4 (cell 16 on e & 3 == 3)
        float accx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int xoff_b = 0;

        // Three tile buffers, requests two planes ahead, and the LDS reads of plane k+1 issued BEHIND the arithmetic of plane k, in front
        // of the barrier: with two buffers every wave read its operands right behind the barrier, i.e. all sixteen waited out the LDS
        // latency together, once per plane, with nothing to cover it (first version: 3000 cycles per plane against 1920 of arithmetic).
        // What crosses the barrier in registers is kept small: the in1 pixels and the first 16 floats of the window row (4 b128 reads);
        // the row's last b128 and the extra task's six operands are read at the top of the plane and used by the arithmetic that comes
        // last (the pairs q + d >= 16, then the extra task), behind ~150 instructions that cover their latency.
        static_assert(NB4 == 5, "16- / 17-wide windows, 4 pixels per lane: 19 / 20 floats of the window row");
        float b[4 * NB4], ax = 0.f, bx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        ff_f4 a4;
        auto read_head = [&](const lds_f *tb) __attribute__((always_inline)) {
            a4 = *(const lds_f4 *)(tb + aoff + PX * lane);
            const lds_f4 *br = (const lds_f4 *)(tb + boff);
#pragma unroll
            for (int j = 0; j < NB4 - 1; ++j) {
                const ff_f4 v = br[j];
                b[4 * j] = v[0]; b[4 * j + 1] = v[1]; b[4 * j + 2] = v[2]; b[4 * j + 3] = v[3];
            }
        };
        auto read_tail = [&](const lds_f *bt) __attribute__((always_inline)) {
            const ff_f4 v = ((const lds_f4 *)(bt + boff))[NB4 - 1];
            b[16] = v[0]; b[17] = v[1]; b[18] = v[2]; b[19] = v[3];
            if constexpr (EXTRA) {
                ax = bt[aoff + 16 * wave + (lane >> 2)];                // (slot of pixel (lane >> 2) & 3 of group 4 wave + (lane >> 4))
#pragma unroll
                for (int j = 0; j < 5; ++j) bx[j] = bt[xoff_b + j];
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the tables are in LDS
        if constexpr (EXTRA) {
            const int pp = lane >> 2, lg = 4 * wave + (pp >> 2), q = pp & 3, c = lane & 3;
            xoff_b = gtab[lg] + 16 * PITCH + q + 4 * c;
        }
        // ONE loop over kk = 0 .. K+1: iteration kk requests plane kk, does the arithmetic of plane kk-2 and reads the operands of plane
        // kk-1 (the two fill iterations included, so that the staging code exists once: inlined three times it cost 40 spilled scalars)
        const int kplain = kslow >= 0 ? kslow : p.K;
        // (Tried: the loop once per role -- what a wave's second staging task is -- so that the role tests fold away: with three inlined
        //  copies the staging pointers came out as vector registers and the requests' scalar-base form did not assemble.)
        lds_f *bs = tile, *bc = tile + BUFSZ, *bp = tile + 2 * BUFSZ;     // buffers of plane kk (requested now), kk - 2 (this iteration's arithmetic), kk - 1
        auto plane = [&](int kk, const bool STEADY) __attribute__((always_inline)) {   // STEADY: 2 <= kk < kplain, a constant at the call
            if (FF_DEEP) read_head(bc);
            read_tail(bc);                                                // (kk < 2: nothing there yet, nothing is computed from it)
            if (STEADY || kk < p.K) stage(kk, bs, STEADY);
            __builtin_amdgcn_sched_barrier(0);
            // the (pixel, cell) pairs in batches of FF_BATCH: all differences, then all squares, then all adds
            auto pairs = [&](auto first_part) {
                constexpr bool FIRST = decltype(first_part)::value;
                static_for_q<0, PX>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    constexpr int dlo = FIRST ? 0 : (16 - q < MW ? 16 - q : MW), dhi = FIRST ? (16 - q < MW ? 16 - q : MW) : MW;
                    static_for_q<0, (dhi - dlo + FF_BATCH - 1) / FF_BATCH>([&](auto bc) {
                        constexpr int d0 = dlo + decltype(bc)::value * FF_BATCH, n = (dhi - d0 < FF_BATCH) ? dhi - d0 : FF_BATCH;
                        float df[n > 0 ? n : 1];
#pragma unroll
                        for (int i = 0; i < n; ++i) df[i] = a4[q] - b[q + d0 + i];
#pragma unroll
                        for (int i = 0; i < n; ++i) df[i] = df[i] * df[i];
#pragma unroll
                        for (int i = 0; i < n; ++i) acc[q][d0 + i] = acc[q][d0 + i] + df[i];
                    });
                });
            };
            if (STEADY || kk >= 2) {
                pairs(std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
                pairs(std::false_type{});
                if constexpr (EXTRA) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const float df = ax - bx[j];
                        accx[j] = accx[j] + df * df;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!FF_DEEP) read_head(bp);     // (unconditional -- in the first and the last iteration the values are not used -- so that the operands stay
                               //  in ONE register set: with the read under a condition the compiler kept two sets and 20 moves per plane)
            // my requests of plane kk have landed; behind the barrier everyone's have, and every wave is past its reads of plane kk-3
            if (FF_DEEP) {
                // all but THIS iteration's requests have landed: the plane of the next iteration's arithmetic
                // (plain planes: kk < K, or < K - 1 where the last plane takes the clamped path, whose request count differs)
                // vmcnt(2) whatever the wave's role: everything but the LAST two requests has landed -- all of the previous plane's, and, for
                // the waves with a second task, that task's requests of this plane, which went out first, a whole plane of arithmetic ago
                // (counting the role's own 2 / 3 / 4 requests took a compare-and-branch chain per plane)
                if (!STEADY && kk >= kplain) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            { lds_f *const t = bs; bs = bc; bc = bp; bp = t; }            // (kk + 1) % 3, (kk - 1) % 3, kk % 3
        };
        // fill (planes 0, 1 requested), steady planes (request + arithmetic + counted wait, none of the per-plane tests), the rest (the
        // clamped last plane where this wave has one, and the two drain iterations)
        plane(0, false);
        plane(1, false);
        int kk = 2;
        for (; kk < kplain; ++kk) plane(kk, true);
        for (; kk < p.K + 2; ++kk) plane(kk, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        // ---- copy-out: one phase per pixel q of the groups -- ALL lanes deposit their pixel q's window row (an LDS instruction costs its
        // issue slot whatever its execution mask: the first version, 8 phases of 8 lanes each, spent 43 of a tile's 124 kilocycles here,
        // most of it in 2048 write instructions with 8 active lanes), then wave w copies the windows of lanes 4w .. 4w+3 out: a window is
        // one contiguous run of the output (1 KB, line-aligned, at 16 x 16).  The image is [lane][WNP]; WNP = WN + 4 where WN is a multiple
        // of 8 keeps the 64 lanes' 16-B writes on different banks.
        const long long tile_px0 = (long long)y_first * p.W1 + xA0;       // first pixel of the tile (row-major pixel index)
        const int WNP = ff_wnp(WL);
        if constexpr (ARGMIN) {
            // Every (lane, pixel, window row) leaves its row's minimum and the first cell attaining it (strict '<' in cell order) in LDS --
            // cand[slot 4 l + q][row] -- the extra task's lanes one candidate per 4-cell group of row 16 (cand[..][16 + c]); then one
            // thread per pixel slot walks its candidates in window order, again with strict '<': the first minimum of the whole window,
            // exactly what v2_argmin_decode_kernel finds in the stored volume (a NaN never wins; torch.min would return the NaN:
            // equality with the staged host path is for volumes without NaN).
            constexpr int NC = EXTRA ? 20 : 16;                           // candidates per pixel slot
            lds_f *cv = img;                                              // [256][NC] values
            lds_i *ci = (lds_i *)(img + 256 * NC);                        // [256][NC] window indices (0-based)
            {
                const int lc = lane_fresh();
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    float best = __int_as_float(0x7f800000);
                    int bi = 0x7fffffff;
#pragma unroll
                    for (int d = 0; d < MW; ++d)
                        if (acc[q][d] < best) { best = acc[q][d]; bi = dy * MW + d; }
                    cv[(lc * PX + q) * NC + dy] = best;
                    ci[(lc * PX + q) * NC + dy] = bi;
                }
                if (!EXTRA && NW < 16) {                                  // (fewer window rows than candidates: the rest never win)
                    for (int r = NW + wave; r < 16; r += NW)
#pragma unroll
                        for (int q = 0; q < PX; ++q) { cv[(lc * PX + q) * NC + r] = __int_as_float(0x7f800000); ci[(lc * PX + q) * NC + r] = 0x7fffffff; }
                }
            }
            if constexpr (EXTRA) {
                const int pp = lane >> 2, c = lane & 3;                   // slot 16 wave + pp, cells 4 c .. 4 c + 3 (+ cell 16 on c == 3) of row 16
                float best = __int_as_float(0x7f800000);
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if ((j < 4 || c == 3) && accx[j] < best) { best = accx[j]; bi = 16 * MW + 4 * c + j; }
                cv[(16 * wave + pp) * NC + 16 + c] = best;
                ci[(16 * wave + pp) * NC + 16 + c] = bi;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            for (int sl = threadIdx.x; sl < 256; sl += 64 * NW) {
                const int ll = sl >> 2, q = sl & 3;
                const int xg = ll >= nA ? (ll - nA) * PX : xA0 + ll * PX;
                if (g0 + ll >= p.NG || xg + q >= p.W1) continue;
                float best = __int_as_float(0x7f800000);
                int bi = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < NC; ++r) {
                    const float v = cv[sl * NC + r];
                    const int vi = ci[sl * NC + r];
                    if (v < best) { best = v; bi = vi; }
                }
                if (bi == 0x7fffffff) bi = 0;
                const long long px = tile_px0 + ll * PX - (ll >= nA ? padpx : 0) + q;
                const int fy = bi / MW;
                if (p.idx) p.idx[px] = (long long)bi + 1;
                if (p.yflow) p.yflow[px] = (float)(fy - p.tWin);
                if (p.xflow) p.xflow[px] = (float)(bi - fy * MW - p.lWin);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the next tile's tables / planes overwrite the candidates
        } else if constexpr (SOFT) {
            // getModel's tail and processOutput on the costs while they are on the chip (opticalflow_model.lua:94-109: Minus -> SoftMax over
            // the window; :153-169 getOutputConfidences; :201-252 decode and centre paste) -- the 300-MB volume of a VGA pair is never written,
            // never re-read by three more passes.  One phase per pixel q of the groups, as in the copy-out: every lane deposits pixel q's window
            // row in the LDS image [lane][WNP]; then SIXTEEN lanes take a window -- lane t of them the cells 64 j + 4 t + i (b128 reads) --
            // and run softmin_body's arithmetic on it (multiscale.hip, windows of more than 64 cells: the same maximum, the same exponential,
            // the same order of the sum, e * (1 / sum)), so the probabilities are the staged modules' bit for bit.  On them: the first maximum
            // with the centre override (argbest_kernel<true>), or extractOutput's first 8 values above 0.11 in index order, its sorting
            // network and its score (extract_kernel<8>), and the confidence score > threshold; the index is decoded (x2yx minus
            // centered2onebased(0, 0)) and lands, with the confidence, at its place in the full-frame planes.
            constexpr int NJ4 = ((EXTRA ? 17 : 16) * MW + 63) / 64;       // b128 pieces per lane, at most (16 x 16: 4; 17 wide: 5)
            lds_f *cand = img + 64 * WNP;                                 // [64 windows][8 values | 8 indices]
            const int yoffc = (p.maxh + 1) / 2, xoffc = (MW + 1) / 2;     // centered2onebased(geometry, 0, 0)
            static_for_q<0, PX>([&](auto qphase) {
                constexpr int q = decltype(qphase)::value;
                {
                    lds_f *w = img + lane_fresh() * WNP + dy * MW;
                    if constexpr (MW % 4 == 0) {
#pragma unroll
                        for (int d = 0; d < MW; d += 4) *(lds_f4 *)(w + d) = ff_f4{acc[q][d], acc[q][d + 1], acc[q][d + 2], acc[q][d + 3]};
                    } else {
#pragma unroll
                        for (int d = 0; d < MW; ++d) w[d] = acc[q][d];
                    }
                }
                if constexpr (EXTRA) {
                    const int pp = lane >> 2, lg = 4 * wave + (pp >> 2), c = lane & 3;
                    if ((pp & 3) == q) {
                        lds_f *w = img + lg * WNP + 16 * MW + 4 * c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = accx[j];
                        if (c == 3) w[4] = accx[4];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int l0 = 4 * wave; l0 < 64; l0 += 4 * NW) {          // (wave-uniform trip count: the ballots below need every lane)
                    const int lf = lane_fresh(), g = lf >> 4, t = lf & 15, ll = l0 + g;
                    const lds_f *wv = img + ll * WNP + 4 * t;
                    float v[4 * NJ4];
                    float m = -__int_as_float(0x7f800000);
#pragma unroll
                    for (int j = 0; j < NJ4; ++j) {
                        ff_f4 c4 = ff_f4{0.f, 0.f, 0.f, 0.f};
                        if (64 * j + 4 * t < WN) c4 = *(const lds_f4 *)(wv + 64 * j);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v[4 * j + i] = c4[i];
                            if (64 * j + 4 * t + i < WN) m = fmaxf(m, -c4[i]);
                        }
                    }
                    m = row16_max_f32(m);
                    const int xg = ll >= nA ? (ll - nA) * PX : xA0 + ll * PX;
                    const bool live = g0 + ll < p.NG && xg + q < p.W1;
                    long long id = 0;
                    float score = 0.f, conf = 1.f;
                    // Without a threshold only the arg-max of the probabilities leaves the kernel.  p = e * (1 / sum) with e = exp2((cmin - c) log2 e):
                    // the largest e is exactly 1 (the minimum's), and a cell more than FF_TIE above the minimum has e <= 1 - 2^-20, whose product
                    // with 1 / sum rounds strictly below 1 / sum.  So a window with ONE cell within FF_TIE of its minimum has that cell as its only
                    // maximal probability whatever the sum is -- no exponential, no sum, no division: 3 instead of 12 vector instructions per cell.
                    // Windows with several such cells (flat regions, exact ties) take the full arithmetic below, wave by wave: same results.
                    bool full = p.use_thr != 0;
                    int fi = 0x7fffffff;
                    if (!full && DFE_FF_SOFT_FAST) {
                        const float lim = FF_TIE - m;                     // (m = max(-c) = -cmin)
                        int cnt = 0;
#pragma unroll
                        for (int j = NJ4 - 1; j >= 0; --j)
#pragma unroll
                            for (int i = 3; i >= 0; --i) {
                                const bool on = 64 * j + 4 * t + i < WN && v[4 * j + i] <= lim;
                                cnt += on ? 1 : 0;
                                fi = on ? 64 * j + 4 * t + i : fi;
                            }
#define FF_STEP(ctrl) cnt += __builtin_amdgcn_update_dpp(0, cnt, ctrl, 0xf, 0xf, false); fi = min(fi, __builtin_amdgcn_update_dpp(0, fi, ctrl, 0xf, 0xf, false));
                        FF_STEP(0x128) FF_STEP(0x124) FF_STEP(0x4E) FF_STEP(0xB1)
#undef FF_STEP
                        full = __builtin_amdgcn_ballot_w64(cnt != 1) != 0;    // (wave-uniform)
                    } else {
                        full = true;
                    }
                    if (!full) {
                        id = (long long)fi + 1;
                    } else {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NJ4; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool on = 64 * j + 4 * t + i < WN;
                            const float e = on ? dfe_exp_nonpos(-v[4 * j + i] - m) : 0.f;
                            v[4 * j + i] = e;
                            if (on) s = s + e;
                        }
                    s = row16_sum_f32_ordered(s);
                    const float inv = 1.0f / s;
#pragma unroll
                    for (int n = 0; n < 4 * NJ4; ++n) v[n] = v[n] * inv;   // (cells beyond the window: 0 * inv = 0, below every probability's use)
                    if (!p.use_thr) {
                        // input:max(3), first maximum; where it equals the centre cell's probability the index is the centre's (:156-160)
                        float b = v[0];
                        int bi = 4 * t;
#pragma unroll
                        for (int j = 0; j < NJ4; ++j)
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (64 * j + 4 * t + i < WN && v[4 * j + i] > b) { b = v[4 * j + i]; bi = 64 * j + 4 * t + i; }
                        row16_argmax_first(b, bi);
                        const float pc = dfe_exp_nonpos(-img[ll * WNP + p.middle - 1] - m) * inv;
                        id = b == pc ? (long long)p.middle : (long long)bi + 1;
                    } else {
                        // extractoutput.extractOutput(input, scores, 0.11, imaxs): the first 8 probabilities above 0.11 in index order
                        // (extract_output.cpp:83-110), sorted by its network, imaxs = the largest one's index, scores = the sum of the prefix sums
                        lds_f *cw = cand + ll * 16;
                        cw[t] = 0.f;
                        int n = 0;
#pragma unroll
                        for (int j = 0; j < NJ4; ++j) {
                            int before = 0, mine = 0;
                            bool hit[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                hit[i] = 64 * j + 4 * t + i < WN && (double)v[4 * j + i] > 0.11;
                                const unsigned gm = (unsigned)(__ballot(hit[i]) >> (16 * g)) & 0xffffu;
                                before += __popc(gm & ((1u << t) - 1u));
                                mine += __popc(gm);
                            }
                            int r = n + before;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (hit[i]) {
                                    if (r < 8) { cw[r] = v[4 * j + i]; cw[8 + r] = (float)(64 * j + 4 * t + i + 1); }
                                    ++r;
                                }
                            n += mine;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the group's 16 lanes are lanes of this wave: its LDS operations are in order)
                        float hv[8], hi[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) { hv[k] = cw[k]; hi[k] = cw[8 + k]; }
                        id = p.middle;                                     // nothing above 0.11: the reference leaves imaxs / scores as they were --
                        if (hv[0] > 0) {                                   // defined here as the centre index and a zero score
                            dfe_sort8(hv, hi);
                            id = (long long)hi[0];
#pragma unroll
                            for (int k = 1; k < 8; ++k) hv[k] += hv[k - 1];
                            double a = 0;
#pragma unroll
                            for (int k = 0; k < 8; ++k) a += hv[k];
                            score = (float)a;
                        }
                        conf = score > p.thr ? 1.f : 0.f;                  // scores:gt(threshold)
                    }
                    }
                    if (live && t == 0) {
                        const int py = y_first + (ll >= nA ? 1 : 0), pxc = xg + q;
                        const long long px = (long long)py * p.W1 + pxc;
                        const int i0 = (int)id - 1, ty = i0 / MW;
                        if (p.idx) p.idx[px] = id;
                        if (p.scores) p.scores[px] = score;
                        const long long fo = (long long)(p.ho + py) * p.wFull + p.wo + pxc;
                        if (p.full) {
                            p.full[fo] = (float)(ty + 1 - yoffc);
                            p.full[p.fullplane + fo] = (float)(i0 - ty * MW + 1 - xoffc);
                        }
                        if (p.full_conf) p.full_conf[fo] = conf;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // before the next phase (or tile) overwrites the image
            });
        } else
        static_for_q<0, PX>([&](auto qphase) {
            constexpr int q = decltype(qphase)::value;
            {
                const int lc = lane_fresh();
                const int x = lc >= nA ? (lc - nA) * PX : xA0 + lc * PX;
                if (g0 + lc < p.NG && x + q < p.W1) {
                    // (the window's image is shifted by its output address mod 16 B, so that 16-B pieces are aligned on both sides)
                    const int prl = lc * PX - (lc >= nA ? padpx : 0);
                    const int ash = (int)(((uintptr_t)(p.out + (tile_px0 + prl + q) * WN + dy0 * MW) >> 2) & 3);
                    lds_f *w = img + lc * WNP + ash + dy * MW;
                    if constexpr (MW % 4 == 0) {
#pragma unroll
                        for (int d = 0; d < MW; d += 4) *(lds_f4 *)(w + d) = ff_f4{acc[q][d], acc[q][d + 1], acc[q][d + 2], acc[q][d + 3]};
                    } else {
#pragma unroll
                        for (int d = 0; d < MW; ++d) w[d] = acc[q][d];
                    }
                }
            }
            if constexpr (EXTRA) {
                // window row 16: the extra task's lanes that hold pixel q of their group
                const int pp = lane >> 2, lg = 4 * wave + (pp >> 2), c = lane & 3;
                const int xg = lg >= nA ? (lg - nA) * PX : xA0 + lg * PX;
                if ((pp & 3) == q && g0 + lg < p.NG && xg + q < p.W1) {
                    const int prl = lg * PX - (lg >= nA ? padpx : 0);
                    const int ash = (int)(((uintptr_t)(p.out + (tile_px0 + prl + q) * WN) >> 2) & 3);
                    lds_f *w = img + lg * WNP + ash + 16 * MW + 4 * c;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = accx[j];
                    if (c == 3) w[4] = accx[4];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (halves && MW == 16 && (WL == 128 || WL == 64) && !((uintptr_t)p.out & 15)) {
                // half / quarter windows of 16 x 16: 128 / 64 floats = 32 / 16 lanes' 16-B pieces, aligned -- a wave copies 2 / 4 windows per step
                const int lgl = WL == 128 ? 5 : 4, wps = 64 >> lgl, pc = lane & ((1 << lgl) - 1);
                for (int l2 = wps * wave; l2 < 64; l2 += wps * NW) {
                    const int ll = l2 + (lane >> lgl);
                    const int xg = ll >= nA ? (ll - nA) * PX : xA0 + ll * PX;
                    const int prl = ll * PX - (ll >= nA ? padpx : 0);
                    const ff_f4 v = *(const lds_f4 *)(img + ll * WNP + 4 * pc);
                    if (g0 + ll < p.NG && xg + q < p.W1) {
                        float *gw = p.out + (tile_px0 + prl + q) * WN + dy0 * MW + 4 * pc;
                        asm volatile("global_store_dwordx4 %0, %1, off" FF_ST_FLAGS ::"v"(gw), "v"(v) : "memory");
                    }
                }
            } else
            // the block's waves share the 64 windows: wave w takes lanes w, w + NW, ...
            for (int ll = wave; ll < 64; ll += NW) {
                const int xg = ll >= nA ? (ll - nA) * PX : xA0 + ll * PX;
                if (g0 + ll >= p.NG || xg + q >= p.W1) continue;         // (wave-uniform)
                const int prl = ll * PX - (ll >= nA ? padpx : 0);        // pixel index of the group relative to the tile's first pixel
                float *gw = p.out + (tile_px0 + prl + q) * WN + dy0 * MW;  // my rows of the window: WL floats
                const int ash = (int)(((uintptr_t)gw >> 2) & 3), head = (4 - ash) & 3;
                const lds_f *sw = img + ll * WNP + ash;
                const int nb4 = (WL - head) >> 2, tail0 = head + 4 * nb4;
                const lds_f4 *sb = (const lds_f4 *)(sw + head);
                const float *gb = gw + head;
                constexpr int NJ4 = (17 * MW / 4 + 63) / 64;              // 16-B pieces per lane and window (2)
                ff_f4 v[NJ4];
                float vh = 0.f;
#pragma unroll
                for (int j = 0; j < NJ4; ++j) v[j] = sb[min(lane + 64 * j, nb4 - 1)];
                if (lane < 8) vh = sw[lane < 4 ? min(lane, WL - 1) : min(tail0 + lane - 4, WL - 1)];   // lanes 0..3: head floats, 4..7: tail floats
                {
#pragma unroll
                    for (int j = 0; j < NJ4; ++j)
                        if (lane + 64 * j < nb4)
                            asm volatile("global_store_dwordx4 %0, %1, %2" FF_ST_FLAGS ::"v"((unsigned)(lane + 64 * j) * 16u), "v"(v[j]), "s"(gb) : "memory");
                    if (lane < head) gw[lane] = vh;
                    if (lane >= 4 && lane < 8 && tail0 + lane - 4 < WL) gw[tail0 + lane - 4] = vh;
                }
            }
            // every wave is past its reads of the image before the next phase (or the next tile's tables / first plane) overwrites it
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        });
    }
}

}  // namespace

// the shapes this kernel takes (the pointers' alignment apart): 16- / 17-wide windows of 4 .. 17 rows on frames at least 64 groups wide
static bool ff_shape_ok(const dfe_ctx *ctx, int K, int H1, int W1, int maxh, int maxw) {
    if (ctx->cv_mode == 1 || ctx->cv_mode == 2 || ctx->opt[DFE_OPT_FM_FLAT] == 0) return false;
    if (maxw != 16 && maxw != 17) return false;
    if (maxh < 4 || maxh > 17 || (maxh == 17 && maxw != 17)) return false;
    const int G = dfe_cdiv(W1, FF_PX);
    if (G < FF_GROUPS || K < 1 || H1 < 1) return false;                    // (a tile must not touch more than two image rows)
    const long long NGl = (long long)H1 * G;
    if (NGl > (1ll << 30) || (long long)H1 * W1 * maxh * maxw >= (1ll << 40)) return false;
    return true;
}

// whether dfe_feat_matching_flat_argmin will take the shape (4-byte aligned feature maps assumed): the one-call models decide with it
// whether the volume needs a place in the scratch arena at all
bool dfe_feat_matching_flat_argmin_takes(const dfe_ctx *ctx, int K, int H1, int W1, int maxh, int maxw) { return ff_shape_ok(ctx, K, H1, W1, maxh, maxw); }

// *handled stays false when the shape is not this kernel's (the caller goes on to the round-3 kernels)
static int ff_launch(dfe_ctx *ctx, const float *in1, int pitch1, long long plane1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out,
                     long long *idx, float *xflow, float *yflow, const DfeSoftOut *soft, bool *handled) {
    const int mode = soft ? FF_SOFT : out ? FF_VOLUME : FF_ARGMIN;
    const bool argmin = mode == FF_ARGMIN;
    *handled = false;
    if (!ff_shape_ok(ctx, K, H1, W1, maxh, maxw)) return DFE_OK;
    const int G = dfe_cdiv(W1, FF_PX);
    if (((uintptr_t)in1 | (uintptr_t)in2 | (uintptr_t)out) & 3) return DFE_OK;
    if (pitch1 < W1 || plane1 < (long long)(H1 - 1) * pitch1 + W1 || plane1 >= (1ll << 29)) return DFE_OK;
    const long long NGl = (long long)H1 * G;
    FfArgs a{};
    a.in1 = in1; a.in2 = in2; a.out = out;
    a.pitch1 = pitch1; a.plane1 = plane1;
    a.K = K; a.H1 = H1; a.W1 = W1; a.maxh = maxh; a.H2 = H1 + maxh - 1; a.W2 = W1 + maxw - 1;
    a.G = G; a.NG = (int)NGl; a.ntiles = dfe_cdiv(NGl, FF_GROUPS);
    a.idx = idx; a.xflow = xflow; a.yflow = yflow;
    a.lWin = (maxw + 1) / 2 - 1; a.tWin = (maxh + 1) / 2 - 1;            // version2/test.lua:18-19
    if (soft) {
        a.middle = (maxw + 1) / 2 + maxw * ((maxh + 1) / 2 - 1);           // getMiddleIndex: yx2x(centered2onebased(0, 0)), opticalflow_model.lua:12-14,28-43
        a.use_thr = soft->use_threshold; a.thr = soft->threshold;
        a.wFull = soft->wFull; a.fullplane = (long long)soft->hFull * soft->wFull;
        a.ho = (soft->hFull - H1) / 2; a.wo = (soft->wFull - W1) / 2;    // processOutput: floor((hImg - h) / 2), :228-230
        a.full = soft->full; a.full_conf = soft->full_conf; a.scores = soft->scores; a.idx = soft->index;
        if ((soft->full || soft->full_conf) && (soft->hFull < H1 || soft->wFull < W1)) return DFE_OK;
    }
    const bool extra = maxh == 17;
    // two half blocks per tile and CU where the window's rows split evenly into halves of >= 4 waves (the arg-min and soft-max forms need
    // the whole window in one block; 17 rows = 17 waves do not fit a CU's registers as 9 + 8)
    // (option fm_split: 0 whole tiles, 1 / 2 halves, 4 quarters; the launcher's own choice: quarters for few planes -- the copy-out is
    //  then a larger share of a tile, K = 10: 0.102 -> 0.098 ms, time_matching.lua's shape 0.024 -> 0.022 -- else halves, which cost
    //  less staging: profiles/r04_ao_fm_quarters.txt)
    const int want = ctx->opt[DFE_OPT_FM_SPLIT] < 0 ? (K <= 16 ? 4 : 2) : ctx->opt[DFE_OPT_FM_SPLIT] == 1 ? 2 : ctx->opt[DFE_OPT_FM_SPLIT];
    const bool can = mode == FF_VOLUME && !extra;
    a.S = can && want >= 4 && maxh == 16 ? 4 : can && want >= 2 && maxh >= 8 && maxh % 2 == 0 ? 2 : 1;
    a.nd = extra ? 16 : maxh / a.S;
    const int NW = a.nd;
    const int PITCH = maxw == 17 ? FfGeom<17>::PITCH : FfGeom<16>::PITCH;
    const int WNP = ff_wnp((extra ? 17 : a.nd) * maxw);
    // the copy-out image [64][WNP]; the arg-min form keeps its candidates there instead -- cv[256][NC] and ci[256][NC], NC = 16 (20 with
    // the extra task) whatever the window's height: larger than the image of a window of fewer than 8 rows; the soft-max form reads its
    // windows in 16-B pieces up to cell 4 * 16 * ceil(WN / 64) of the last window and keeps extractOutput's candidates [64][16] behind
    const size_t img_floats = mode == FF_SOFT ? (size_t)64 * WNP + 64 * 16 + 64
                                              : std::max((size_t)64 * WNP, argmin ? (size_t)2 * 256 * (extra ? 20 : 16) : (size_t)0);
    const size_t lds = ((size_t)3 * (extra ? 18 : a.nd + 1) * PITCH + 3 * 64 * FF_PX + 64 + img_floats) * sizeof(float);
    if (lds * a.S > 160 * 1024) return DFE_OK;
    void (*kern)(FfArgs);
    if (mode == FF_SOFT) kern = maxw == 17 ? (extra ? feat_matching_flat_kernel<17, true, FF_SOFT> : feat_matching_flat_kernel<17, false, FF_SOFT>)
                                           : feat_matching_flat_kernel<16, false, FF_SOFT>;
    else if (argmin) kern = maxw == 17 ? (extra ? feat_matching_flat_kernel<17, true, FF_ARGMIN> : feat_matching_flat_kernel<17, false, FF_ARGMIN>)
                                       : feat_matching_flat_kernel<16, false, FF_ARGMIN>;
    else kern = maxw == 17 ? (extra ? feat_matching_flat_kernel<17, true> : feat_matching_flat_kernel<17, false>) : feat_matching_flat_kernel<16, false>;
    DFE_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nblk = (a.ntiles < ctx->ncu ? a.ntiles : ctx->ncu) * a.S;
    {
        DfeProfScope prof(ctx);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * NW), lds, ctx->stream, a);
    }
    DFE_LAUNCH_CHECK(ctx);
    ctx->last_kernel = mode == FF_SOFT ? "feat_matching_flat_kernel+softmax" : argmin ? "feat_matching_flat_kernel+argmin" : "feat_matching_flat_kernel";
    *handled = true;
    return DFE_OK;
}

int dfe_feat_matching_flat(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, float *out, bool *handled) {
    *handled = false;
    if (!out) return DFE_OK;
    return ff_launch(ctx, in1, W1, (long long)H1 * W1, in2, K, H1, W1, maxh, maxw, out, nullptr, nullptr, nullptr, nullptr, handled);
}

// in1 as a view: rows pitch1 floats apart, planes plane1 floats apart (prepareInput's narrow of a feature map, opticalflow_model.lua:147-149)
int dfe_feat_matching_flat_strided(dfe_ctx *ctx, const float *in1, int pitch1, long long plane1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                   float *out, bool *handled) {
    *handled = false;
    if (!out) return DFE_OK;
    return ff_launch(ctx, in1, pitch1, plane1, in2, K, H1, W1, maxh, maxw, out, nullptr, nullptr, nullptr, nullptr, handled);
}

// nn.SpatialMatching(maxh, maxw) + `min` over the window + the decode of version2/test.lua:45-51, without the volume
int dfe_feat_matching_flat_argmin(dfe_ctx *ctx, const float *in1, const float *in2, int K, int H1, int W1, int maxh, int maxw, long long *idx, float *xflow,
                                  float *yflow, bool *handled) {
    if (dfe_feat_matching_mfma_takes(ctx, K, H1, W1, maxh, maxw)) {   // opt-in (fm_mfma = 1): the banded GEMM on the matrix cores
        void *nrm = nullptr;
        int rc = dfe_aux_scratch(ctx, dfe_feat_matching_mfma_scratch(H1, W1, maxh, maxw) * sizeof(float), &nrm);
        if (rc) return rc;
        rc = dfe_feat_matching_mfma(ctx, in1, in2, K, H1, W1, maxh, maxw, (float *)nrm, nullptr, idx, xflow, yflow, handled);
        if (rc != DFE_OK || *handled) return rc;
    }
    return ff_launch(ctx, in1, W1, (long long)H1 * W1, in2, K, H1, W1, maxh, maxw, nullptr, idx, xflow, yflow, nullptr, handled);
}

// nn.SpatialMatching -> nn.Minus -> SoftMax over the window -> processOutput, without the volume (the FF_SOFT epilogue)
int dfe_feat_matching_flat_soft(dfe_ctx *ctx, const float *in1, int pitch1, long long plane1, const float *in2, int K, int H1, int W1, int maxh, int maxw,
                                const DfeSoftOut *soft, bool *handled) {
    *handled = false;
    if (!soft) return DFE_OK;
    return ff_launch(ctx, in1, pitch1, plane1, in2, K, H1, W1, maxh, maxw, nullptr, nullptr, nullptr, nullptr, soft, handled);
}
