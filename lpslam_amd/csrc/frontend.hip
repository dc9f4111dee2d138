// frontend.hip -- ORB front end for gfx950 (MI355X): image pyramid, FAST-9/16 in 64-px cells, quad-tree keypoint
// distribution, intensity-centroid orientation + fused 7x7 Gaussian + rotated BRIEF-256.
//
// Replaces what the reference reaches through feed_stereo_frame / feed_monocular_frame
// (/root/reference/src/Trackers/OpenVSLAMStereoTracker.cpp:293-295, src/Trackers/OpenVSLAMTracker.cpp:120):
// [UPSTREAM] openvslam::feature::orb_extractor::extract with the Feature.* parameters of
// src/Trackers/OpenVSLAMTrackerBase.cpp:193-198.  All pixel arithmetic is integer / fixed point and all float
// arithmetic is written without FMA contraction, so results are bit-identical to the CPU definition.
//
// Layout in HBM: one slab per image holding all pyramid levels (row pitch = width rounded up to 64 B), images of a
// batch back to back; FAST results in fixed 1024-entry slots per 64x64 cell; keypoints/descriptors per image.
#include "internal.h"

#pragma clang fp contract(off)

using namespace lpslam;

// ------------------------------------------------------------------------------------------------------------
// K1  pyramid: level l = bilinear resize of level l-1, 11-bit fixed-point coefficients (cv::resize INTER_LINEAR 8u)
//     HBM-bound: reads ~1.44 B and writes 1 B per output pixel.  One thread = 4 output pixels = one dword store.
// ------------------------------------------------------------------------------------------------------------
// One launch builds levels 1..L-1 of every image.  A work-group is one horizontal band of one image: it computes its rows of
// level 1 from the input, synchronises itself, computes level 2 from those, and so on.  The row ranges (host table, see
// lpslam_hip_create) include the few rows above and below the band that its coarser rows are interpolated from, so bands never
// wait for one another; the overlap rows are written by both neighbours with identical bytes.
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// Four destination pixels of one row.  The right tap is always column sx0 + 1 and the lower tap the next row (where the table
// clamps them their weight is 0), so one unaligned 16-bit load fetches both taps of a source row, v_perm spreads them to 16-bit
// lanes and v_dot2_u32_u16 applies the weight pair (w0 | w1 << 16) as the table stores it.  (One 8-byte load per row with 64-bit
// shifts was measured slower.)
template <typename Tab>
__device__ __forceinline__ uint32_t pyr_down_dword(const uint8_t* __restrict__ src, int sp, int dw, int dx0, int2 ey, Tab xplanes, int per_row)
{
    const int b0 = ey.y & 0xFFFF, b1 = ey.y >> 16;
    const unsigned o0 = (unsigned)(ey.x & 0xFFFF) * (unsigned)sp, o1 = o0 + (unsigned)sp;   // uniform base + 32-bit offsets
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {       // branch-free: pad columns compute the last column and are zeroed by a select
        const int2 ex = xplanes[k * per_row + (dx0 >> 2)];       // plane k, dword q: column min(4 q + k, dw - 1)
        const unsigned sx0 = (unsigned)ex.x & 0xFFFFu;
        const unsigned t0 = *reinterpret_cast<const unsigned short*>(src + o0 + sx0);
        const unsigned t1 = *reinterpret_cast<const unsigned short*>(src + o1 + sx0);
        const int r0 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, t0, 0x0c010c00u)), __builtin_bit_cast(us2, ex.y), 0u, false);
        const int r1 = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, t1, 0x0c010c00u)), __builtin_bit_cast(us2, ex.y), 0u, false);
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        packed |= (dx0 + k < dw ? (uint32_t)(v & 0xFF) : 0u) << (8 * k);
    }
    return packed;
}

// ------------------------------------------------------------------------------------------------------------
// Mapping reserve in software (lpslam_hip_set_mapping_reserve).  The bundle adjustments that run beside the front end need compute
// units with their LDS free (a panel-pair workgroup: 128 KB, the band factorisation: 130 KB), which they do not find while the
// extraction kernels keep eight workgroups on every compute unit.  A CU mask on the front end's streams gave them room (round 3) but
// a masked queue in the process slows every solve that runs beside uploads.  Instead the extraction kernels
// reserve compute units THEMSELVES: launched as a chip-filling grid of persistent workgroups, a workgroup that finds itself on a
// reserved compute unit (HW_REG_XCC_ID / HW_REG_HW_ID against a table made at calibration, api.hip) leaves at once, the others take
// chunks of work items from one counter until it runs out.  Every item is done exactly once by some workgroup that stays -- and
// workgroups DO stay wherever the dispatcher puts the grid: a workgroup on a reserved compute unit takes a leave ticket
// (counter[1]) and leaves only while fewer than max_leave have left before it.  max_leave is what an evenly spread grid sheds
// (grid x reserved / all compute units) plus half of that again, at most three quarters of the grid: when other work (a solve of
// the mapping thread, the prefetch stream, another session) fills the unreserved compute units and the grid drains through the
// reserved ones instead, the rest of the grid stays there and does the work -- the reserve is a preference, completion is not.
// ------------------------------------------------------------------------------------------------------------
struct FeQueue {
    const uint32_t* cu_table;        // [8 XCC][8 words]: bit (hw_id >> 8) & 255 set = reserved compute unit
    int* counter;                    // [0] next chunk, [1] leave tickets (both zeroed in front of the launch)
    int n_items, chunk;
    int reserve_cus;                 // per XCD (of 32)
};
__device__ __forceinline__ bool fe_on_reserved_cu(const uint32_t* table)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned id = (hw >> 8) & 0xffu;
    return (table[(xcc & 7u) * 8u + (id >> 5)] >> (id & 31u)) & 1u;
}
// does this workgroup leave?  (uniform over the workgroup: its wavefronts share a compute unit; a barrier inside)
__device__ __forceinline__ bool fe_leaves(const FeQueue& q)
{
    if (!fe_on_reserved_cu(q.cu_table)) return false;
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(q.counter + 1, 1);
    __syncthreads();
    const int grid = (int)(gridDim.x * gridDim.y);
    const int even = grid * q.reserve_cus / 32;
    return __builtin_amdgcn_readfirstlane(s_ticket) < min(even + even / 2, grid - grid / 4);
}
// first item of the next chunk for the whole workgroup (every thread calls it; barriers inside)
__device__ __forceinline__ int fe_next_chunk(const FeQueue& q)
{
    __shared__ int s_chunk;
    __syncthreads();                 // the previous item is finished by everybody (its LDS may be reused) and s_chunk has been read
    if (threadIdx.x == 0) s_chunk = atomicAdd(q.counter, 1);
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(s_chunk) * q.chunk;      // uniform: what is derived from it (level, sizes, pointers) stays in scalar registers
}
// calibration: where does workgroup b run?  (xcc << 16 | hw_id & 0xffff)
__global__ void k_fe_where(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x] = (xcc & 0xfu) << 16 | (hw & 0xffffu);
    }
    // stay a little so that the grid spreads over every compute unit instead of draining through the first ones
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(8);
}

// Test hook (lpslam_hip_debug_occupy_unreserved): one workgroup per compute unit that claims the whole LDS; those on UNRESERVED
// compute units hold it for `ticks` of the 100 MHz wall clock (a bounded spin), the others leave.  While it runs, a workgroup that
// needs LDS -- every extraction kernel's -- only finds room on the reserved compute units: the placement the leave tickets exist for.
__global__ __launch_bounds__(1024) void k_fe_occupy(const uint32_t* cu_table, unsigned long long ticks, int* landed)
{
    extern __shared__ int s_hold[];
    if (fe_on_reserved_cu(cu_table)) return;
    if (threadIdx.x == 0) { s_hold[0] = 1; atomicAdd(landed, 1); }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

template <bool kTablesInLds>
__global__ __launch_bounds__(1024) void k_pyr_bands(uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt, ImgSel image0,
                                                    const int2* __restrict__ rs_pack, int rs_entries,
                                                    const int2* __restrict__ band_rows, FeQueue fq, int n_bands)
{
    extern __shared__ int2 s_tab[];      // the resize tables of all levels (57 KB at 1280x720, 8 levels)
    if (fq.cu_table && fe_leaves(fq)) return;
    if (kTablesInLds) {
        for (int i = threadIdx.x; i < rs_entries; i += 1024) s_tab[i] = rs_pack[i];
        __syncthreads();
    }
  // direct launch: workgroup (band, image), once; queued: items band + n_bands * image from the counter until it runs out
  for (bool once = true;; once = false) {
    int item;
    if (fq.cu_table) { item = fe_next_chunk(fq); if (item >= fq.n_items) break; }
    else { if (!once) break; item = (int)(blockIdx.x + n_bands * blockIdx.y); }
    const int band_x = item % n_bands, image_y = item / n_bands;
    uint8_t* img = pyr + (size_t)lp_image(image0, image_y) * image_slab;
    for (int level = 1; level < lt.n_levels; ++level) {
        const int2 rows = band_rows[level * kPyrMaxBands + band_x];
        const int dw = lt.w[level], dp = lt.pitch[level], sp = lt.pitch[level - 1];
        const uint8_t* src = img + lt.off[level - 1];
        uint8_t* dst = img + lt.off[level];
        const int per_row = dp >> 2;
        const int total = (rows.y - rows.x + 1) * per_row;
        const float inv = 1.0f / (float)per_row;
        const int ytab = lt.dy_start[level] + rows.x, xtab = lt.dx_start[level];
        constexpr int kIlp = 4;          // dwords in flight per thread: the loop is a chain of table read -> pixel loads
        for (int t = threadIdx.x; t < total; t += 1024 * kIlp) {
            uint32_t v[kIlp];
            int r[kIlp], q[kIlp];
#pragma unroll
            for (int u = 0; u < kIlp; ++u) {
                const int tu = min(t + 1024 * u, total - 1);     // past the end: recompute the last dword, store nothing
                r[u] = (int)(((float)tu + 0.5f) * inv);
                q[u] = tu - r[u] * per_row;
                if (q[u] < 0) { --r[u]; q[u] += per_row; } else if (q[u] >= per_row) { ++r[u]; q[u] -= per_row; }
                if (kTablesInLds) v[u] = pyr_down_dword(src, sp, dw, q[u] * 4, s_tab[ytab + r[u]], s_tab + xtab, per_row);
                else v[u] = pyr_down_dword(src, sp, dw, q[u] * 4, rs_pack[ytab + r[u]], rs_pack + xtab, per_row);
            }
#pragma unroll
            for (int u = 0; u < kIlp; ++u)
                if (t + 1024 * u < total) *reinterpret_cast<uint32_t*>(dst + (size_t)(rows.x + r[u]) * dp + q[u] * 4) = v[u];
        }
        __threadfence_block();
        __syncthreads();
    }
  }
}

// (Measured and dropped, round 6: the SEPARABLE form -- every source row of a chunk of output rows resampled horizontally once into LDS
// as 16-bit r >> 4, the vertical pass reading its two rows from there, the level's tables staged beside them: same bytes, 0.179 ms per
// 32-image step on half of the chip against 0.164 for the direct form above (0.237 with the tables read through the cache).  Half the
// resampling arithmetic, but two workgroup barriers and an LDS round trip per chunk; the direct form has no barrier inside a level.)
// ------------------------------------------------------------------------------------------------------------
// K2  FAST-9/16 + score + 3x3 NMS per 64-px cell (cv::FAST on the cell sub-image, ini threshold then min threshold)
//     One 256-thread workgroup per cell; the (64+6)^2 tile is staged in LDS with aligned dword loads.
//     S(p) = max over the 16 arcs of 9 contiguous ring pixels of min(v - ring) resp. min(ring - v):
//     p is a corner at threshold t iff S > t, and cv's cornerScore is S - 1.
// ------------------------------------------------------------------------------------------------------------
#define TILE_PITCH 80
#define SMAP_PITCH 72

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// (Round 6, measured and dropped: the strength of ONE polarity per candidate -- a pixel whose pre-test failed for a polarity has that
// polarity's strength <= t, which is stored as 0 either way, so only the polarities that passed need their 16 arcs: 81 instead of 129
// instructions per candidate, candidates in two lists (darker from the front of the queue, brighter-only from the back) so that a
// wavefront instruction works on one polarity.  Bit-exact (21 front-end tests); k_fast_cells 247.9 us per 32 images against 245.0 with
// both polarities, same box, alternating runs: what the strength pass saves, the second set of flags, ballots and list ends costs.)
__device__ __forceinline__ int fast_strength(const uint8_t* t /* points at centre in LDS tile */)
{
    const int v = t[0];
    int d[16];
    d[0] = v - t[3 * TILE_PITCH + 0];   d[1] = v - t[3 * TILE_PITCH + 1];   d[2] = v - t[2 * TILE_PITCH + 2];
    d[3] = v - t[1 * TILE_PITCH + 3];   d[4] = v - t[3];                    d[5] = v - t[-1 * TILE_PITCH + 3];
    d[6] = v - t[-2 * TILE_PITCH + 2];  d[7] = v - t[-3 * TILE_PITCH + 1];  d[8] = v - t[-3 * TILE_PITCH];
    d[9] = v - t[-3 * TILE_PITCH - 1];  d[10] = v - t[-2 * TILE_PITCH - 2]; d[11] = v - t[-1 * TILE_PITCH - 3];
    d[12] = v - t[-3];                  d[13] = v - t[1 * TILE_PITCH - 3];  d[14] = v - t[2 * TILE_PITCH - 2];
    d[15] = v - t[3 * TILE_PITCH - 1];
    int lo3[16], hi3[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lo3[i] = min3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
        hi3[i] = max3i(d[i], d[(i + 1) & 15], d[(i + 2) & 15]);
    }
    int a = -512, b = 512;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a = max(a, min3i(lo3[i], lo3[(i + 3) & 15], lo3[(i + 6) & 15]));
        b = min(b, max3i(hi3[i], hi3[(i + 3) & 15], hi3[(i + 6) & 15]));
    }
    return max(a, -b);
}

#ifdef LPSLAM_FAST_STAMPS
// development: shader-clock cycles of every workgroup's phases (thread 0's view; plain stores into the workgroup's own slots --
// atomics on shared counters made the launch four times slower) (tools/dev_fast_stamps.py)
constexpr int kFastStampWgs = 32768;
__device__ unsigned long long g_fast_stamps[8 * kFastStampWgs];
#define FAST_STAMP(k) do { if (threadIdx.x == 0 && fs_wg < kFastStampWgs) { const unsigned long long now_ = __builtin_readcyclecounter(); g_fast_stamps[8 * fs_wg + (k)] = now_ - fs_last; fs_last = now_; } } while (0)
#define FAST_STAMP_BEGIN unsigned long long fs_last = __builtin_readcyclecounter(); const int fs_wg = (int)(blockIdx.y * gridDim.x + blockIdx.x); if (threadIdx.x == 0 && fs_wg < kFastStampWgs) g_fast_stamps[8 * fs_wg + 7] = 1ull
extern "C" __attribute__((visibility("default"))) int lpslam_hip_debug_fast_stamps(unsigned long long* out, int n_wg, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_stamps), (size_t)8 * n_wg * sizeof(unsigned long long));
    if (reset) { void* p = nullptr; rc |= (int)hipGetSymbolAddress(&p, HIP_SYMBOL(g_fast_stamps)); if (p) rc |= (int)hipMemset(p, 0, sizeof(unsigned long long) * 8 * kFastStampWgs); }
    return rc;
}
#else
#define FAST_STAMP(k) do {} while (0)
#define FAST_STAMP_BEGIN do {} while (0)
#endif
__device__ __forceinline__ void fast_cells_body(int cell_arg, int image_arg, const uint8_t* __restrict__ pyr, size_t image_slab, const LevelTable& lt,
                                                int ini_thr, int min_thr, uint32_t* __restrict__ cell_keys,
                                                int32_t* __restrict__ cell_count, int cells_per_image, const ImgSel& image0, int dbg,
                                                const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[70 * TILE_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t smap[66 * SMAP_PITCH];
    __shared__ unsigned long long m_sel[64];     // per-row masks of the NMS survivors (LDS atomics)
    __shared__ int rowoff[64];
    __shared__ uint16_t queue[4][16 * 64];       // pre-test survivors, one queue per wavefront (its 16 rows): no shared counter
    __shared__ int n_keep;

    FAST_STAMP_BEGIN;
    const int cell = cell_arg, image = lp_image(image0, image_arg);
    int level = 0;
    while (level + 1 < lt.n_levels && cell >= lt.cell_start[level + 1]) ++level;
    const int lc = cell - lt.cell_start[level];
    const int ci = lc / lt.cells_x[level], cj = lc - ci * lt.cells_x[level];
    const int W = lt.w[level], H = lt.h[level], P = lt.pitch[level];
    const int min_x = kEdge + cj * kCell, min_y = kEdge + ci * kCell;
    const int cw = min(min_x + kCell + kOverlap, W - kEdge) - min_x;   // 7..70 (cells narrower than 7 px hold no corner)
    const int ch = min(min_y + kCell + kOverlap, H - kEdge) - min_y;
    const uint8_t* src = pyr + (size_t)image * image_slab + lt.off[level];
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));          // opaque per call: inside the queued kernel's item loop the lane-derived values are recomputed per cell instead of staying live across it (113 -> 6x VGPRs)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    // camera mask ([UPSTREAM] orb_extractor::is_in_mask: level coordinate x scale factor, truncated, looked up in the level-0 mask,
    // 0 = masked out): even image slots are left eyes, odd ones right eyes.  A cell with a corner in the mask is skipped.
    const uint8_t* mask = (image & 1) ? mask1 : mask0;
    const float mscale = lt.scale[level];
    const int W0 = lt.w[0], H0 = lt.h[0];
    auto masked = [&](int y, int x) -> bool {
        const int my = min(max((int)((float)y * mscale), 0), H0 - 1), mx = min(max((int)((float)x * mscale), 0), W0 - 1);
        return mask[(size_t)my * W0 + mx] == 0;
    };
    if (mask) {
        const int max_x = min_x + cw, max_y = min_y + ch;
        if (masked(min_y, min_x) || masked(max_y, min_x) || masked(min_y, max_x) || masked(max_y, max_x)) {
            if (tid == 0) cell_count[(size_t)image * cells_per_image + cell] = 0;
            return;
        }
    }

    // stage the tile: LDS column 1 + j holds pixel min_x + j, so that the first tested pixel (min_x + 3) sits on a dword boundary
    // (column 4) and a lane of the pre-test owns one aligned dword = four pixels.  Global reads stay aligned dwords; the byte shift
    // between the two alignments is a v_alignbyte per staged dword.
    const int x_al = min_x & ~3, shift = min_x - x_al;
    const int s_al = (shift + 3) & 3, k_al = shift == 0 ? -1 : 0;          // LDS dword d = global dwords (d + k_al, d + k_al + 1) >> 8 s_al
    {
        const int ndw = (cw + 4) >> 2;                   // the dwords that hold pixels: at most 7 bytes are read past the cell's last pixel
        uint32_t g0[6], g1[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {                    // 70 rows x 20 dwords = 1400 <= 6 x 256: every load of a thread is in flight before the first store
            const int i = tid + 256 * u, r = i / 20, c = i - r * 20;
            g0[u] = 0; g1[u] = 0;
            if (r < ch && c < ndw) {
                // 4-byte words at x_al + 4 (c + k_al): the first may start 4 bytes left of x_al (>= 16)
                const uint32_t* g = reinterpret_cast<const uint32_t*>(src + (size_t)(min_y + r) * P + x_al) + c + k_al;
                g0[u] = g[0]; g1[u] = g[1];
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int i = tid + 256 * u, r = i / 20, c = i - r * 20;
            if (r < ch) *reinterpret_cast<uint32_t*>(&tile[r * TILE_PITCH + 4 * c]) = __builtin_amdgcn_alignbyte(g1[u], g0[u], s_al);
        }
    }
    for (int i = tid; i < 66 * SMAP_PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(smap)[i] = 0;
    __syncthreads();
    FAST_STAMP(0);          // geometry, mask test, tile staged

    const int vw = cw - 6, vh = ch - 6;          // valid (corner-tested) interior, <= 64 x 64; may be <= 0
    // cv::FAST at the initial threshold; a cell without a single NMS survivor is redone at the minimum threshold.
    // The arc strength S does not depend on the threshold (corner at t  <=>  S > t), so each pass is:
    //   1. pre-test, every pixel: a 9-arc of the 16-ring always contains two neighbouring compass points (ring positions
    //      0, 4, 8, 12) and two neighbouring diagonal ones (2, 6, 10, 14), so a corner needs such a pair both darker or both
    //      brighter than the centre by more than t.  A lane tests FOUR neighbouring pixels from eleven dword LDS reads (the
    //      byte-per-pixel form issued nine ds_read_u8 per pixel: LDS instruction issue was what the kernel was made of);
    //      a wavefront covers four rows of 64 pixels;
    //   2. survivors only, dense lanes from an LDS queue: S (16 arcs of 9, v_min3 / v_max3), stored if S > t;
    //   3. survivors only: strict 3x3 non-maximum suppression on the S map, survivors set their bit in the row mask.
    int thr = ini_thr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (tid < 64) m_sel[tid] = 0;
        if (tid == 0) n_keep = 0;
        __syncthreads();
        const int l16 = lane & 15, rs = lane >> 4;
        int n_mine = 0;                                  // entries of this wavefront's queue (uniform)
        // Rows of the four 16-lane groups: the two groups of a 32-lane half (the unit the LDS serves a dword read in) take rows FOUR
        // apart -- 4 x 20 dwords = 80 = 16 (mod 32 banks), so their 16 consecutive dwords fall on disjoint banks; neighbouring rows
        // (20 dwords apart) shared four banks and every one of the eleven reads below took two passes.
        for (int r4 = wave * 8; r4 < vh; r4 += (r4 & 2) ? 30 : 2) {
            const int r = r4 + (rs >> 1) + 4 * (rs & 1);  // tested row (tile row r + 3); rows past vh read inside the tile and are masked out
            const uint32_t* t = reinterpret_cast<const uint32_t*>(&tile[min(r + 3, 66) * TILE_PITCH]) + 1 + l16;      // the lane's own dword
            const uint32_t* tu = t - 2 * (TILE_PITCH / 4), *td = t + 2 * (TILE_PITCH / 4);
            const uint32_t C = t[0], Cl = t[-1], Cr = t[1];
            const uint32_t P0 = t[3 * (TILE_PITCH / 4)], P8 = t[-3 * (TILE_PITCH / 4)];
            const uint32_t P4 = __builtin_amdgcn_alignbyte(Cr, C, 3), P12 = __builtin_amdgcn_alignbyte(C, Cl, 1);
            const uint32_t P2 = __builtin_amdgcn_alignbyte(td[1], td[0], 2), P14 = __builtin_amdgcn_alignbyte(td[0], td[-1], 2);
            const uint32_t P6 = __builtin_amdgcn_alignbyte(tu[1], tu[0], 2), P10 = __builtin_amdgcn_alignbyte(tu[0], tu[-1], 2);
            // two pixels per instruction: bytes spread to 16-bit halves (v_perm), then v_pk_min / v_pk_max / v_pk_add / v_pk_sub_u16.
            // "some neighbouring pair both darker than c - t" is max(min(p0, p8), min(p4, p12)) < c - t (a pair takes one point of
            // {0, 8} and one of {4, 12}); brighter likewise; the same for the diagonals, with the same polarity.  The comparisons are
            // saturating subtractions: x < y  <=>  sat(y - x) != 0.
            unsigned flags = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t sel = h ? 0x0c030c02u : 0x0c010c00u;
#define PK(x) __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, (x), sel))
                const us2 c = PK(C), thr2 = {(unsigned short)thr, (unsigned short)thr};
                const us2 lo = __builtin_elementwise_sub_sat(c, thr2), hi = c + thr2;
                const us2 p0 = PK(P0), p4 = PK(P4), p8 = PK(P8), p12 = PK(P12), p2 = PK(P2), p6 = PK(P6), p10 = PK(P10), p14 = PK(P14);
#undef PK
                const us2 dk = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(p0, p8), __builtin_elementwise_min(p4, p12)),
                                                         __builtin_elementwise_max(__builtin_elementwise_min(p2, p10), __builtin_elementwise_min(p6, p14)));
                const us2 br = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_max(p0, p8), __builtin_elementwise_max(p4, p12)),
                                                         __builtin_elementwise_min(__builtin_elementwise_max(p2, p10), __builtin_elementwise_max(p6, p14)));
                const uint32_t f = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(lo, dk)) | __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(br, hi));
                flags |= ((f & 0xFFFFu) ? 1u : 0u) << (2 * h) | ((f >> 16) ? 2u : 0u) << (2 * h);
            }
            // rows / columns outside the tested interior
            const int nvalid = r < vh ? min(max(vw - 4 * l16, 0), 4) : 0;
            flags &= (1u << nvalid) - 1u;
            // queue slots by ballot: pixel k of every lane, k = 0..3 (any order would do: the S map and the masks are positional)
            if (__ballot(flags != 0)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long b = __ballot((flags >> k) & 1u);
                    if ((flags >> k) & 1u) queue[wave][n_mine + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = (uint16_t)((r << 6) | (4 * l16 + k));
                    n_mine += __popcll(b);
                }
            }
        }
        FAST_STAMP(1);      // pre-test of every pixel + queue
        // every wavefront works through its own queue straight away (its rows interleave with the others': the load is even; the
        // strength reads the tile only, so no barrier is needed here)
        for (int i = lane; i < n_mine; i += 64) {
            const int e = queue[wave][i], r = e >> 6, x = e & 63;
            const int sv = fast_strength(&tile[(r + 3) * TILE_PITCH + 4 + x]);
            smap[(r + 1) * SMAP_PITCH + x + 1] = (uint8_t)(sv > thr ? sv : 0);
        }
        FAST_STAMP(2);      // strength of the candidates
        __syncthreads();
        FAST_STAMP(3);      // waiting for the other wavefronts' strengths
        int kept = 0;
        for (int i = lane; i < n_mine; i += 64) {
            const int e = queue[wave][i], r = e >> 6, x = e & 63;
            const uint8_t* q = &smap[(r + 1) * SMAP_PITCH + x + 1];
            const int sc = q[0];
            if (sc > 0 && sc > q[-1] && sc > q[1] && sc > q[-SMAP_PITCH - 1] && sc > q[-SMAP_PITCH] && sc > q[-SMAP_PITCH + 1] &&
                sc > q[SMAP_PITCH - 1] && sc > q[SMAP_PITCH] && sc > q[SMAP_PITCH + 1]) {
                // the threshold fallback looks at what FAST found; a corner whose own position is masked out is dropped afterwards
                if (!mask || !masked(min_y + r + 3, min_x + x + 3)) atomicOr(&m_sel[r], 1ull << x);
                kept = 1;
            }
        }
        if (kept) n_keep = 1;              // benign race: every writer stores 1
        __syncthreads();
        FAST_STAMP(4);      // suppression (+ barrier)
        if (n_keep || min_thr >= thr) break;
        thr = min_thr;
    }
    if (wave == 0) {
        const unsigned long long sel = m_sel[lane];
        const int c = __popcll(sel);
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        rowoff[lane] = incl - c;
        if (lane == 63) cell_count[(size_t)image * cells_per_image + cell] = incl;
    }
    __syncthreads();
    uint32_t* out = cell_keys + ((size_t)image * cells_per_image + cell) * kCellSlots;
    for (int r = wave; r < vh; r += 4) {
        const unsigned long long sel = m_sel[r];
        if ((sel >> lane) & 1ull) {
            const int pos = rowoff[r] + __popcll(sel & ((1ull << lane) - 1ull));
            const uint32_t score = (uint32_t)smap[(r + 1) * SMAP_PITCH + lane + 1] - 1u;   // cornerScore = S - 1
            const uint32_t x = (uint32_t)(lane + 3 + cj * kCell), y = (uint32_t)(r + 3 + ci * kCell);
            out[pos] = (score << 24) | (y << 12) | x;
        }
    }
    FAST_STAMP(5);          // row offsets, keys out
}
// direct launch: one workgroup per (cell, image)
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt,
                                                    int ini_thr, int min_thr, uint32_t* __restrict__ cell_keys,
                                                    int32_t* __restrict__ cell_count, int cells_per_image, ImgSel image0, int dbg,
                                                    const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1)
{
    fast_cells_body(blockIdx.x, blockIdx.y, pyr, image_slab, lt, ini_thr, min_thr, cell_keys, cell_count, cells_per_image, image0, dbg, mask0, mask1);
}
// queued (mapping reserve): persistent workgroups take chunks of cells; a kernel of its own, so that the direct one keeps its registers
// (both in one kernel: 110 instead of 63 VGPRs, four instead of eight wavefronts per SIMD, 246 -> 297 us per 32 images)
__global__ __launch_bounds__(256) void k_fast_cells_q(const uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt,
                                                      int ini_thr, int min_thr, uint32_t* __restrict__ cell_keys,
                                                      int32_t* __restrict__ cell_count, int cells_per_image, ImgSel image0, int dbg,
                                                      const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ mask1, FeQueue fq)
{
    if (fe_leaves(fq)) return;
    for (;;) {
        const int first = fe_next_chunk(fq);
        if (first >= fq.n_items) break;
        for (int it = first; it < min(first + fq.chunk, fq.n_items); ++it) {
            if (it != first) __syncthreads();          // the previous cell's LDS is free
            fast_cells_body(it % cells_per_image, it / cells_per_image, pyr, image_slab, lt, ini_thr, min_thr, cell_keys, cell_count, cells_per_image, image0, dbg, mask0, mask1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K3  quad-tree distribution (orb_extractor::distribute_keypoints_via_tree), one 1024-thread workgroup per
//     (image, level).  Nodes are kept in ascending creation order (= reverse of the upstream std::list); every
//     pass splits an ordered "visit list" of nodes with scans instead of list surgery:
//       breadth-first passes: all nodes with > 1 corner, visited from the newest to the oldest;
//       final passes: the pool of new children sorted by (count, creation) descending, cut where the node count
//       reaches the quota.  Ties go to the node created last.
// ------------------------------------------------------------------------------------------------------------
struct DistLds {
    int* cnt4;      // [16*Q]   quadrant histograms of the visit list
    int* order;     // [4*Q pow2] visit list (node index), doubles as bitonic sort buffer
    int* nodepos;   // [Q]      node -> position in visit list or -1
    int* kpre;      // [4*Q+1]  prefix of children count over the visit list
    int* keptrank;  // [Q+1]
    int* misc;      // [64]
};

__device__ __forceinline__ int block_excl_scan_1024(int v, int* wave_tot /* LDS[16] */, int* total)
{
    // exclusive scan of one value per thread over a 1024-thread block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int t = wave_tot[w]; if (w < wave) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// exclusive scan in place of a[0..n) (n <= per_thread*1024), returns total via *total; all threads call
__device__ void block_scan_array(int* a, int n, int* wave_tot, int* total)
{
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; ++i) s += a[i];
    int tot;
    int base = block_excl_scan_1024(s, wave_tot, &tot);
    for (int i = b; i < e; ++i) { const int v = a[i]; a[i] = base; base += v; }
    __syncthreads();
    *total = tot;
}

// Candidates live in registers: thread t owns candidates t, t + 1024, ... (CPT slots, fully unrolled so every slot is a
// fixed register); levels with more than 1024 * CPT candidates keep the rest in the global scratch arrays.  Nodes (box,
// count; ping-pong) live in LDS.  A pass therefore touches HBM/L2 only for the overflow candidates.
#define DIST_CPT 24

__device__ __forceinline__ void distribute_body(int image_arg, int level_arg, const LevelTable& lt, const uint32_t* __restrict__ cell_keys,
                                                const int32_t* __restrict__ cell_count, int cells_per_image,
                                                uint32_t* __restrict__ cand_key, uint32_t* __restrict__ cand_node,
                                                int32_t* __restrict__ cand_count, int cand_per_image,
                                                uint32_t* __restrict__ sel_key, int32_t* __restrict__ sel_count,
                                                int slots_per_image, const ImgSel& image0)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int level = level_arg, image = lp_image(image0, image_arg);
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));          // opaque per call (see fast_cells_body)
    const int tid = tid_;
    const int N = lt.quota[level];
    const int Q = lt.qcap[level];          // node count never exceeds max(N + 2, 4 * roots) < Q
    int sortcap = 1; while (sortcap < Q) sortcap <<= 1;
    const int ncell = lt.cells_x[level] * lt.cells_y[level];
    int* cnt4 = lds;                                   // max(4*Q, ncell + 1): quadrant histograms / cell offsets / winners
    int* order = cnt4 + max(4 * Q, ncell + 1);         // sortcap: visit list, bitonic sort buffer
    int* nodepos = order + sortcap;                    // Q
    int* kpre = nodepos + Q;                           // Q + 1
    int* keptrank = kpre + Q + 1;                      // Q + 1
    int* misc = keptrank + Q + 1;                      // 64
    int* wave_tot = misc + 32;
    uint2* nbox[2]; int* ncnt[2];
    nbox[0] = reinterpret_cast<uint2*>(misc + 64); nbox[1] = nbox[0] + Q;
    ncnt[0] = reinterpret_cast<int*>(nbox[1] + Q); ncnt[1] = ncnt[0] + Q;

    const uint32_t* ckeys = cell_keys + ((size_t)image * cells_per_image + lt.cell_start[level]) * kCellSlots;
    const int32_t* ccnt = cell_count + (size_t)image * cells_per_image + lt.cell_start[level];
    uint32_t* ckey = cand_key + (size_t)image * cand_per_image + lt.cand_start[level];
    uint32_t* cnode = cand_node + (size_t)image * cand_per_image + lt.cand_start[level];
    uint32_t* out_sel = sel_key + (size_t)image * slots_per_image + lt.slot_start[level];

    // ---- candidate list = per-cell slots in cell order (cells row-major, corners row-major inside a cell)
    int n_cand;
    int* coff = cnt4;
    for (int i = tid; i < ncell; i += 1024) coff[i] = ccnt[i];
    __syncthreads();
    block_scan_array(coff, ncell, wave_tot, &n_cand);
    if (tid == 0) { coff[ncell] = n_cand; cand_count[image * lt.n_levels + level] = n_cand; }
    __syncthreads();
    if (n_cand == 0) { if (tid == 0) sel_count[image * lt.n_levels + level] = 0; return; }

    // Thread t owns the CONTIGUOUS candidates [t * cpt, t * cpt + cpt): corners arrive in cell order, so a thread's corners
    // mostly share one quad-tree node and every histogram / maximum below is run-length aggregated inside the thread
    // (2-3 LDS atomics per thread and sweep instead of one per corner).
    uint32_t rk[DIST_CPT];     // key: score << 24 | y << 12 | x
    uint32_t rn[DIST_CPT];     // node index (bits 30..31: quadrant scratch)
    const int cpt = min(DIST_CPT, (n_cand + 1023) / 1024);
    const int n_reg = min(n_cand, 1024 * cpt);
    const int c_first = tid * cpt;
    // 1. the per-cell slots are compacted into the level's candidate array
    //    with coalesced loads and stores (below);
    // 2. every thread then fetches its own contiguous run from the compact array, 16 bytes at a time when the run length allows.
    //    (Before, each thread walked the cell offsets for its run and gathered straight from the cell slots: 24 scattered
    //    4-byte loads per thread, each touching 48 cache lines per wavefront -- 75 k of the level-0 workgroup's 218 k cycles.)
    {
        // wavefront w compacts candidates [w * 64 cpt, (w + 1) * 64 cpt): in step u its lanes take 64 consecutive candidates (coalesced on
        // both sides), so from step to step a lane's candidate moves on by 64 and its cell by one at most now and then -- one binary
        // search per lane, then a short walk; offsets first, then the loads of all steps together, then the stores
        const int wv = tid >> 6, ln = tid & 63;
        const int wbase = wv * 64 * cpt;
        int cell = 0;
        {
            const int c = min(wbase + ln, n_cand - 1);
            int lo = 0, hi = ncell;                    // largest cell with coff[cell] <= c
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= c) lo = mid; else hi = mid; }
            cell = lo;
        }
        int src_off[DIST_CPT];
#pragma unroll
        for (int u = 0; u < DIST_CPT; ++u) {
            const int c = min(wbase + 64 * u + ln, n_cand - 1);
            while (coff[cell + 1] <= c) ++cell;        // coff[ncell] = n_cand > c
            src_off[u] = cell * kCellSlots + (c - coff[cell]);
        }
        uint32_t kv[DIST_CPT];
#pragma unroll
        for (int u = 0; u < DIST_CPT; ++u) kv[u] = ckeys[src_off[u]];
#pragma unroll
        for (int u = 0; u < DIST_CPT; ++u) { const int c = wbase + 64 * u + ln; if (u < cpt && c < n_reg) ckey[c] = kv[u]; }
    }
    for (int c = n_reg + tid; c < n_cand; c += 1024) {      // beyond the register-resident part (more than 1024 x 24 candidates)
        int lo = 0, hi = ncell;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (coff[mid] <= c) lo = mid; else hi = mid; }
        ckey[c] = ckeys[(size_t)lo * kCellSlots + (c - coff[lo])];
    }
    __threadfence_block();
    __syncthreads();
    if ((cpt & 3) == 0) {
        const uint4* run = reinterpret_cast<const uint4*>(ckey + c_first);        // candidate arrays start on 4 KB boundaries
#pragma unroll
        for (int s4 = 0; s4 < DIST_CPT / 4; ++s4) {
            uint4 q4 = make_uint4(0, 0, 0, 0);
            if (4 * s4 < cpt && c_first + 4 * s4 < n_reg) q4 = run[s4];              // may run past n_cand inside the level's array: masked below
            rk[4 * s4] = q4.x; rk[4 * s4 + 1] = q4.y; rk[4 * s4 + 2] = q4.z; rk[4 * s4 + 3] = q4.w;
        }
#pragma unroll
        for (int s = 0; s < DIST_CPT; ++s) { if (!(s < cpt && c_first + s < n_reg)) rk[s] = 0; rn[s] = 0; }
    } else {
#pragma unroll
        for (int s = 0; s < DIST_CPT; ++s) {
            const int c = c_first + s;
            rk[s] = (s < cpt && c < n_reg) ? ckey[c] : 0u;
            rn[s] = 0;
        }
    }
    __syncthreads();
#define DIST_VALID(s) ((s) < cpt && c_first + (s) < n_reg)

    // ---- initialize_nodes
    const int nxg = lt.nxg[level], nyg = lt.nyg[level];
    const double delta_x = lt.delta_x[level], delta_y = lt.delta_y[level];
    const int nini = nxg * nyg;             // 4 * nini + 4 <= Q
    for (int i = tid; i < nini; i += 1024) cnt4[i] = 0;
    __syncthreads();
    // root column / row of every pixel coordinate of this level, tabulated once: two FP64 divisions per candidate were 26 k cycles of
    // the level-0 workgroup (four wavefronts per SIMD, 24 candidates per thread)
    uint8_t* rtab_x = reinterpret_cast<uint8_t*>(ncnt[1] + Q);
    uint8_t* rtab_y = rtab_x + lt.w[level];
    for (int i = tid; i < lt.w[level]; i += 1024) rtab_x[i] = (uint8_t)min(255u, (unsigned)((double)(float)i / delta_x));
    for (int i = tid; i < lt.h[level]; i += 1024) rtab_y[i] = (uint8_t)min(255u, (unsigned)((double)(float)i / delta_y));
    __syncthreads();
    auto root_of = [&](uint32_t k) -> uint32_t {
        const unsigned ix = rtab_x[k & 0xFFF], iy = rtab_y[(k >> 12) & 0xFFF];
        unsigned root = ix + iy * nxg;
        if (root >= (unsigned)nini) root = nini - 1;
        return root;
    };
    {
        int run_key = -1, run_cnt = 0;
#pragma unroll
        for (int s = 0; s < DIST_CPT; ++s) {
            int key = -1;
            if (DIST_VALID(s)) { rn[s] = root_of(rk[s]); key = (int)rn[s]; }
            if (key != run_key) { if (run_key >= 0) atomicAdd(&cnt4[run_key], run_cnt); run_key = key; run_cnt = 0; }
            run_cnt += key >= 0;
        }
        if (run_key >= 0) atomicAdd(&cnt4[run_key], run_cnt);
    }
    for (int c = n_reg + tid; c < n_cand; c += 1024) { const uint32_t r = root_of(ckey[c]); cnode[c] = r; atomicAdd(&cnt4[r], 1); }
    __syncthreads();
    // array position a <-> root (nini-1-a); drop empty roots
    for (int a = tid; a < nini; a += 1024) keptrank[a] = cnt4[nini - 1 - a] > 0 ? 1 : 0;
    __syncthreads();
    int alive;
    block_scan_array(keptrank, nini, wave_tot, &alive);
    int cur = 0;
    for (int a = tid; a < nini; a += 1024) {
        const int root = nini - 1 - a;
        if (cnt4[root] > 0) {
            const int ix = root % nxg, iy = root / nxg;
            const unsigned bx = (unsigned)(int)(delta_x * ix), by = (unsigned)(int)(delta_y * iy);
            const unsigned ex = (unsigned)(int)(delta_x * (ix + 1)), ey = (unsigned)(int)(delta_y * (iy + 1));
            nbox[cur][keptrank[a]] = make_uint2(bx | (by << 16), ex | (ey << 16));
            ncnt[cur][keptrank[a]] = cnt4[root];
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DIST_CPT; ++s) if (DIST_VALID(s)) rn[s] = (uint32_t)keptrank[nini - 1 - (int)rn[s]];
    for (int c = n_reg + tid; c < n_cand; c += 1024) cnode[c] = (uint32_t)keptrank[nini - 1 - (int)cnode[c]];
    __syncthreads();

    // ---- split passes.  Node bookkeeping (a few hundred entries) is done by wavefront 0 alone with wave-level scans, so a
    //      pass costs four workgroup barriers; the candidate sweeps (histogram, remap) use all 16 wavefronts.
    const int lane = tid & 63, wave = tid >> 6;
    auto wsync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    auto wave_scan = [&](int* arr, int n) -> int {      // exclusive scan in place by one wavefront; returns the total
        const int per = (n + 63) >> 6;
        const int b0 = lane * per, e0 = min(b0 + per, n);
        int sum = 0;
        for (int i = b0; i < e0; ++i) sum += arr[i];
        int incl = sum;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        int run = incl - sum;
        for (int i = b0; i < e0; ++i) { const int v = arr[i]; arr[i] = run; run += v; }
        const int total = __shfl(incl, 63);
        wsync();
        return total;
    };
    // misc: [0] m (visit-list length)  [1] jcut  [2] n_kept  [3] n_children  [4] pool_n
    if (tid == 0) { misc[5] = alive; misc[6] = 0 /* sorted_phase */; misc[7] = 0 /* pool_begin */; misc[8] = cur; }
    __syncthreads();
    for (int pass = 0; pass < 64; ++pass) {
        alive = misc[5];
        const bool sorted_phase = misc[6] != 0;
        const int pool_begin = misc[7];
        cur = misc[8];
        // 1. visit list (wavefront 0)
        if (wave == 0) {
            int m;
            if (!sorted_phase) {
                // all nodes with count > 1, newest first
                for (int i = lane; i < alive; i += 64) keptrank[i] = ncnt[cur][i] > 1 ? 1 : 0;
                wsync();
                m = wave_scan(keptrank, alive);
                for (int i = lane; i < alive; i += 64) {
                    if (ncnt[cur][i] > 1) { const int pos = m - 1 - keptrank[i]; order[pos] = i; nodepos[i] = pos; }
                    else nodepos[i] = -1;
                }
                wsync();
                // split centre of every visit-list node, by visit position (keptrank is free until the bookkeeping after the
                // histogram sweep): the sweep then needs one table read instead of the node's box and its halving
                for (int i = lane; i < m; i += 64) {
                    const uint2 bb = nbox[cur][order[i]];
                    const int bx = bb.x & 0xFFFF, by = bb.x >> 16, ex = bb.y & 0xFFFF, ey = bb.y >> 16;
                    keptrank[i] = (bx + ((ex - bx + 1) >> 1)) | ((by + ((ey - by + 1) >> 1)) << 16);
                }
            } else {
                // pool = new children with more than one corner; keys (count << 11 | index) into kpre, ranked below by all threads
                for (int i = lane; i < alive; i += 64) { nodepos[i] = -1; keptrank[i] = (i >= pool_begin && ncnt[cur][i] > 1) ? 1 : 0; }
                wsync();
                m = wave_scan(keptrank, alive);
                for (int i = lane; i < alive; i += 64)
                    if (i >= pool_begin && ncnt[cur][i] > 1) kpre[keptrank[i]] = (ncnt[cur][i] << 11) | i;
            }
            for (int i = lane; i < 4 * m; i += 64) cnt4[i] = 0;
            if (lane == 0) misc[0] = m;
        }
        __syncthreads();
        const int m = misc[0];
        if (m == 0) break;       // nothing dividable: node count cannot change any more
        if (sorted_phase) {
            // order by (count, index) descending = rank of every key among the m distinct keys (all threads, m^2 / 1024 compares
            // each, LDS broadcast reads) instead of a sorting network with log^2 m barriers
            for (int i = tid; i < m; i += 1024) {
                const unsigned ki = (unsigned)kpre[i];
                int rank = 0;
                for (int j = 0; j < m; ++j) rank += (unsigned)kpre[j] > ki;
                const int node = (int)(ki & 0x7FFu);
                order[rank] = node; nodepos[node] = rank;
                const uint2 bb = nbox[cur][node];
                const int bx = bb.x & 0xFFFF, by = bb.x >> 16, ex = bb.y & 0xFFFF, ey = bb.y >> 16;
                keptrank[rank] = (bx + ((ex - bx + 1) >> 1)) | ((by + ((ey - by + 1) >> 1)) << 16);      // split centre, see above
            }
            __syncthreads();
        }

        // 2. quadrant histograms of every visit-list node (all wavefronts)
        auto quadrant = [&](uint32_t node, uint32_t k, int* key) -> uint32_t {
            const int pos = nodepos[node];
            *key = -1;
            if (pos < 0) return node;
            const int centre = keptrank[pos];              // (bx + hx) | (by + hy) << 16 of the node at this visit position
            const int x = k & 0xFFF, y = (k >> 12) & 0xFFF;
            const int q = ((centre & 0xFFFF) <= x ? 1 : 0) + ((centre >> 16) <= y ? 2 : 0);
            *key = 4 * pos + q;
            return node | ((uint32_t)q << 30);
        };
        {
            int run_key = -1, run_cnt = 0;
#pragma unroll
            for (int s2 = 0; s2 < DIST_CPT; ++s2) {
                int key = -1;
                if (DIST_VALID(s2)) rn[s2] = quadrant(rn[s2] & 0x3FFFFFFFu, rk[s2], &key);
                if (key != run_key) { if (run_key >= 0) atomicAdd(&cnt4[run_key], run_cnt); run_key = key; run_cnt = 0; }
                run_cnt += key >= 0;
            }
            if (run_key >= 0) atomicAdd(&cnt4[run_key], run_cnt);
        }
        for (int c = n_reg + tid; c < n_cand; c += 1024) { int key; cnode[c] = quadrant(cnode[c] & 0x3FFFFFFFu, ckey[c], &key); if (key >= 0) atomicAdd(&cnt4[key], 1); }
        __syncthreads();

        // 3.-5. children per visit position, cut, kept ranks, new node arrays, phase decision (wavefront 0)
        const int nxt = cur ^ 1;
        if (wave == 0) {
            for (int i = lane; i < m; i += 64)
                kpre[i] = (cnt4[4 * i] > 0) + (cnt4[4 * i + 1] > 0) + (cnt4[4 * i + 2] > 0) + (cnt4[4 * i + 3] > 0);
            wsync();
            const int ktotal = wave_scan(kpre, m);
            if (lane == 0) kpre[m] = ktotal;
            wsync();
            int jcut = m;
            if (sorted_phase) {
                // smallest j >= 1 with alive + kpre[j] - j >= N (the node count after j splits; non-decreasing in j)
                int best = m;
                for (int j = lane + 1; j <= m; j += 64) if (alive + kpre[j] - j >= N) { best = j; break; }
                for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
                jcut = best;
            }
            for (int i = lane; i < alive; i += 64) { const int p = nodepos[i]; keptrank[i] = (p >= 0 && p < jcut) ? 0 : 1; }
            wsync();
            const int n_kept = wave_scan(keptrank, alive);
            const int n_children = kpre[jcut];
            for (int i = lane; i < alive; i += 64) {
                const int p = nodepos[i];
                if (!(p >= 0 && p < jcut)) { nbox[nxt][keptrank[i]] = nbox[cur][i]; ncnt[nxt][keptrank[i]] = ncnt[cur][i]; }
            }
            for (int p = lane; p < jcut; p += 64) {
                const int node = order[p];
                const uint2 bb = nbox[cur][node];
                const int bx = bb.x & 0xFFFF, by = bb.x >> 16, ex = bb.y & 0xFFFF, ey = bb.y >> 16;
                const int hx = (ex - bx + 1) >> 1, hy = (ey - by + 1) >> 1;
                int idx = n_kept + kpre[p];
                const int cbx[4] = {bx, bx + hx, bx, bx + hx}, cby[4] = {by, by, by + hy, by + hy};
                const int cex[4] = {bx + hx, ex, bx + hx, ex}, cey[4] = {by + hy, by + hy, ey, ey};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = cnt4[4 * p + q];
                    cnt4[4 * p + q] = idx;                 // the histogram entry becomes the child's node index: the remap sweep reads it
                    if (n > 0) {
                        nbox[nxt][idx] = make_uint2((unsigned)cbx[q] | ((unsigned)cby[q] << 16), (unsigned)cex[q] | ((unsigned)cey[q] << 16));
                        ncnt[nxt][idx] = n;
                        ++idx;
                    }
                }
            }
            wsync();
            const int new_alive = n_kept + n_children;
            int local = 0;       // pool for the phase decision: new children with more than one corner
            for (int i = n_kept + lane; i < new_alive; i += 64) local += ncnt[nxt][i] > 1;
            for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
            if (lane == 0) {
                misc[1] = jcut; misc[2] = n_kept; misc[3] = n_children; misc[4] = local;
                const bool done = (N <= new_alive) || (new_alive == alive);
                misc[9] = done ? 1 : 0;
                misc[5] = new_alive; misc[7] = n_kept; misc[8] = nxt;
                if (!sorted_phase && N < new_alive + 3 * local) misc[6] = 1;
            }
        }
        __syncthreads();
        // 6. remap candidates (all wavefronts)
        const int jcut = misc[1];
        auto remap = [&](uint32_t v) -> uint32_t {
            const int node = (int)(v & 0x3FFFFFFFu), q = (int)(v >> 30);
            const int p = nodepos[node];
            if (p >= 0 && p < jcut) return (uint32_t)cnt4[4 * p + q];     // child index written by the bookkeeping above
            return (uint32_t)keptrank[node];
        };
#pragma unroll
        for (int s2 = 0; s2 < DIST_CPT; ++s2) if (DIST_VALID(s2)) rn[s2] = remap(rn[s2]);
        for (int c = n_reg + tid; c < n_cand; c += 1024) cnode[c] = remap(cnode[c]);
        const bool done = misc[9] != 0;
        __syncthreads();
        if (done) break;
    }
    alive = misc[5];
    cur = misc[8];
    (void)cur;

    // ---- best response per node (first maximum in candidate order), nodes newest first
    int* win = cnt4;      // alive <= Q
    for (int i = tid; i < alive; i += 1024) win[i] = 0;
    __syncthreads();
    {
        int run_node = -1; unsigned run_best = 0;
#pragma unroll
        for (int s = 0; s < DIST_CPT; ++s) {
            int node = -1; unsigned key = 0;
            if (DIST_VALID(s)) { node = (int)(rn[s] & 0x3FFFFFFFu); key = ((rk[s] >> 24) << 24) | (0xFFFFFFu - (unsigned)(c_first + s)); }
            if (node != run_node) { if (run_node >= 0) atomicMax(reinterpret_cast<unsigned*>(&win[run_node]), run_best); run_node = node; run_best = 0; }
            run_best = max(run_best, key);
        }
        if (run_node >= 0) atomicMax(reinterpret_cast<unsigned*>(&win[run_node]), run_best);
    }
    for (int c = n_reg + tid; c < n_cand; c += 1024)
        atomicMax(reinterpret_cast<unsigned*>(&win[cnode[c] & 0x3FFFFFFFu]), ((ckey[c] >> 24) << 24) | (0xFFFFFFu - (unsigned)c));
    __syncthreads();
    const int cap = lt.slot_start[level + 1] - lt.slot_start[level];     // >= max(N + 3, 4 * roots) >= alive
    for (int i = tid; i < alive && i < cap; i += 1024) {
        const unsigned w = (unsigned)win[alive - 1 - i];
        out_sel[i] = ckey[0xFFFFFFu - (w & 0xFFFFFFu)];
    }
    if (tid == 0) sel_count[image * lt.n_levels + level] = min(alive, cap);
}
// direct: workgroup (image, level), level 0 of every image dispatched first
__global__ __launch_bounds__(1024) void k_distribute(LevelTable lt, const uint32_t* __restrict__ cell_keys,
                                                     const int32_t* __restrict__ cell_count, int cells_per_image,
                                                     uint32_t* __restrict__ cand_key, uint32_t* __restrict__ cand_node,
                                                     int32_t* __restrict__ cand_count, int cand_per_image,
                                                     uint32_t* __restrict__ sel_key, int32_t* __restrict__ sel_count,
                                                     int slots_per_image, ImgSel image0)
{
    distribute_body(blockIdx.x, blockIdx.y, lt, cell_keys, cell_count, cells_per_image, cand_key, cand_node, cand_count, cand_per_image, sel_key, sel_count, slots_per_image, image0);
}
// queued: items image + n_images * level in that order
__global__ __launch_bounds__(1024) void k_distribute_q(LevelTable lt, const uint32_t* __restrict__ cell_keys,
                                                       const int32_t* __restrict__ cell_count, int cells_per_image,
                                                       uint32_t* __restrict__ cand_key, uint32_t* __restrict__ cand_node,
                                                       int32_t* __restrict__ cand_count, int cand_per_image,
                                                       uint32_t* __restrict__ sel_key, int32_t* __restrict__ sel_count,
                                                       int slots_per_image, ImgSel image0, FeQueue fq, int n_images)
{
    if (fe_leaves(fq)) return;
    for (;;) {
        const int it = fe_next_chunk(fq);                // chunk = 1
        if (it >= fq.n_items) break;
        distribute_body(it % n_images, it / n_images, lt, cell_keys, cell_count, cells_per_image, cand_key, cand_node, cand_count, cand_per_image, sel_key, sel_count, slots_per_image, image0);
    }
}

size_t lp_distribute_lds_bytes(int Q, int ncell)
{
    int sortcap = 1; while (sortcap < Q) sortcap <<= 1;
    const int c4 = 4 * Q > ncell + 1 ? 4 * Q : ncell + 1;
    // cnt4 | order | nodepos | kpre | keptrank | misc | node boxes (2 x Q x uint2) | node counts (2 x Q) | root tables (w + h bytes, <= 8192)
    return sizeof(int) * ((size_t)c4 + sortcap + Q + (Q + 1) + (Q + 1) + 64 + 4 * (size_t)Q + 2 * (size_t)Q + 2) + 8192 + 16;
}

// ------------------------------------------------------------------------------------------------------------
// K4+K5+K6  per keypoint: intensity-centroid angle (radius-15 disc, cv::fastAtan2), 7x7 sigma-2 fixed-point Gaussian
//           of the 37x37 neighbourhood computed in LDS (never written to HBM), 256 rotated BRIEF tests.
//           One wavefront per keypoint; the descriptor's four 64-bit words are four wave ballots.
// ------------------------------------------------------------------------------------------------------------
__constant__ int8_t c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.2204460492503131e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.2204460492503131e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// sin/cos of an angle in degrees as one fixed sequence of IEEE double operations (the CPU definition uses the
// same sequence); rounded to float.
__device__ __forceinline__ void sincos_deg(float angle_deg, float* s_out, float* c_out)
{
    const double a = (double)angle_deg * 0.017453292519943295;
    const double qf = floor(a * 0.63661977236758138 + 0.5);
    const int q = (int)qf;
    double r = a - qf * 1.5707963267948966;
    r = r - qf * 6.123233995736766e-17;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6; ps = ps * z + S5; ps = ps * z + S4; ps = ps * z + S3; ps = ps * z + S2; ps = ps * z + S1;
    const double sn = r + r * (z * ps);
    double pc = C6; pc = pc * z + C5; pc = pc * z + C4; pc = pc * z + C3; pc = pc * z + C2; pc = pc * z + C1;
    const double cs = 1.0 - 0.5 * z + z * (z * pc);
    double s, c;
    switch (q & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = (float)s; *c_out = (float)c;
}

#define PR 21                 // raw patch radius: 18 (max rotated pattern radius) + 3 (blur)
#define PW 43                 // raw patch width
#define BW 37                 // blurred patch width (radius 18)
#define RAW_PITCH 48           // multiple of 4, and 12 bytes can be read from column 36 of any row
#define HB_PITCH 40            // row pitch of the horizontally blurred patch (u16): rows start 8-byte aligned
#define DESC_WAVES 4

__device__ __forceinline__ void describe_body(int slot_block, int image_arg, const uint8_t* __restrict__ pyr, size_t image_slab, const LevelTable& lt,
                                              const uint32_t* __restrict__ sel_key,
                                              const int32_t* __restrict__ sel_count, int slots_per_image,
                                              lpslam_hip_keypoint* __restrict__ kpts, uint8_t* __restrict__ desc,
                                              int32_t* __restrict__ kp_count, const ImgSel& image0)
{
    // per wavefront: the raw patch (2064 B), the horizontally blurred one (3440 B); the blurred patch (1369 B) takes the raw patch's
    // place, which nobody reads after the horizontal pass -- 5.5 KB instead of 6.9 KB per wavefront is 29 instead of 23 wavefronts
    // per compute unit, and this kernel is a chain of dependent phases that only other wavefronts can hide
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[DESC_WAVES][PW * RAW_PITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[DESC_WAVES][PW * HB_PITCH];
    static_assert(BW * BW <= PW * RAW_PITCH, "the blurred patch reuses the raw patch's storage");
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));          // opaque per call (see fast_cells_body)
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int image = lp_image(image0, image_arg);
    const int slot = slot_block * DESC_WAVES + wave;
    if (slot >= slots_per_image) return;          // wave-uniform; no block barriers below
    int level = 0;
    while (level + 1 < lt.n_levels && slot >= lt.slot_start[level + 1]) ++level;
    const int k = slot - lt.slot_start[level];
    const int32_t* cnt = sel_count + image * lt.n_levels;
    int dense = k;
    for (int l = 0; l < level; ++l) dense += cnt[l];
    if (slot == 0 && lane == 0) { int t = 0; for (int l = 0; l < lt.n_levels; ++l) t += cnt[l]; kp_count[image] = t; }
    if (k >= cnt[level]) return;

    const uint32_t key = sel_key[(size_t)image * slots_per_image + slot];
    const int cx = (int)(key & 0xFFF) + kEdge, cy = (int)((key >> 12) & 0xFFF) + kEdge;
    const float score = (float)(key >> 24);
    const int P = lt.pitch[level];
    const uint8_t* src = pyr + (size_t)image * image_slab + lt.off[level] + (size_t)(cy - PR) * P + (cx - PR);
    uint8_t* raw = s_raw[wave];
    uint16_t* hb = s_h[wave];
    uint8_t* bl = raw;                                   // from the vertical pass on
    // the 43 x 43 patch: ten (unaligned) dword loads and three byte loads per row instead of 43 byte loads
    // (all of a lane's loads first, then its LDS stores: one round trip to memory instead of ten in a row)
    {
        uint32_t wv[7]; uint8_t bv[3];
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int i = min(lane + 64 * it, PW * 10 - 1), r = i / 10, c4 = i - r * 10;
            __builtin_memcpy(&wv[it], src + (size_t)r * P + 4 * c4, 4);
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int i = min(lane + 64 * it, PW * 3 - 1), r = i / 3, c = 40 + (i - r * 3);
            bv[it] = src[(size_t)r * P + c];
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int i = lane + 64 * it, r = i / 10, c4 = i - r * 10;
            if (i < PW * 10) *reinterpret_cast<uint32_t*>(&raw[r * RAW_PITCH + 4 * c4]) = wv[it];
        }
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            const int i = lane + 64 * it, r = i / 3, c = 40 + (i - r * 3);
            if (i < PW * 3) raw[r * RAW_PITCH + c] = bv[it];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // orientation: m10 = sum u*I, m01 = sum v*I over the disc |u| <= umax[|v|], |v| <= 15
    int m10 = 0, m01 = 0;
    for (int i = lane; i < 31 * 9; i += 64) {            // a lane takes four neighbouring columns (4 .. 39 cover u = -15 .. 15)
        const int r = i / 9, g = i - r * 9;
        const int v = r - 15, lim = c_umax[abs(v)];
        const uint32_t w = *reinterpret_cast<const uint32_t*>(&raw[(PR + v) * RAW_PITCH + 4 + 4 * g]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = 4 + 4 * g + k - PR;
            const int I = abs(u) <= lim ? (int)((w >> (8 * k)) & 0xFF) : 0;
            m10 += u * I; m01 += v * I;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    // separable fixed-point Gaussian {18,34,48,56,48,34,18}/256: rows 0..42 x cols 3..39, then rows 3..39
    // horizontal pass: a lane takes four neighbouring columns of a row -- three dword reads feed 28 taps
    for (int i = lane; i < PW * 10; i += 64) {
        const int r = i / 10, g = i - r * 10;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&raw[r * RAW_PITCH + 4 * g]);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        // output j = taps j .. j + 6 of the 12 loaded bytes: two byte windows (v_alignbyte) and two 4-element dot products
        // (v_dot4_u32_u8) with the coefficient words {18, 34, 48, 56} and {48, 34, 18, 0} -- integer arithmetic, the same sums
        unsigned o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = j == 0 ? w0 : __builtin_amdgcn_alignbyte(w1, w0, j);
            const uint32_t b = j == 0 ? w1 : __builtin_amdgcn_alignbyte(w2, w1, j);
            o4[j] = __builtin_amdgcn_udot4(b, 0x00122230u, __builtin_amdgcn_udot4(a, 0x38302212u, 0u, false), false);
        }
        uint32_t* out = reinterpret_cast<uint32_t*>(&hb[r * HB_PITCH + 4 * g]);
        out[0] = o4[0] | (o4[1] << 16); out[1] = o4[2] | (o4[3] << 16);      // columns 37..39 of the last group are never read
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // vertical pass: a lane takes two neighbouring columns -- seven dword reads feed 14 taps
    for (int i = lane; i < BW * 19; i += 64) {
        const int r = i / 19, c = 2 * (i - r * 19);
        const uint32_t* p = reinterpret_cast<const uint32_t*>(&hb[r * HB_PITCH + c]);
        uint32_t v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = p[k * (HB_PITCH / 2)];
        // the low and the high halves are two columns: v_perm gathers the same half of two rows, v_dot2_u32_u16 applies a pair of taps
        uint32_t lo = 1u << 15, hi = 1u << 15;
        const uint32_t cpair[4] = {18u | (34u << 16), 48u | (56u << 16), 48u | (34u << 16), 18u};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t va = v[2 * k], vb = k < 3 ? v[2 * k + 1] : 0u;
            const uint32_t pl = __builtin_amdgcn_perm(vb, va, 0x05040100u);      // (lo16 of va) | (lo16 of vb) << 16
            const uint32_t ph = __builtin_amdgcn_perm(vb, va, 0x07060302u);      // (hi16 of va) | (hi16 of vb) << 16
            lo = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pl), __builtin_bit_cast(us2, cpair[k]), lo, false);
            hi = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, ph), __builtin_bit_cast(us2, cpair[k]), hi, false);
        }
        bl[r * BW + c] = (uint8_t)(lo >> 16);
        if (c + 1 < BW) bl[r * BW + c + 1] = (uint8_t)(hi >> 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float sa, ca;
    sincos_deg(angle, &sa, &ca);
    const uint8_t* centre = &bl[18 * BW + 18];
    unsigned long long words[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int8_t* pp = &c_pattern[(64 * j + lane) * 4];
        const float x0 = pp[0], y0 = pp[1], x1 = pp[2], y1 = pp[3];
        const int r0 = (int)rintf(x0 * sa + y0 * ca), c0 = (int)rintf(x0 * ca - y0 * sa);
        const int r1 = (int)rintf(x1 * sa + y1 * ca), c1 = (int)rintf(x1 * ca - y1 * sa);
        const int t0 = centre[r0 * BW + c0], t1 = centre[r1 * BW + c1];
        words[j] = __ballot(t0 < t1);
    }
    const size_t o = (size_t)image * slots_per_image + dense;
    if (lane < 4) reinterpret_cast<unsigned long long*>(desc + o * 32)[lane] = words[lane];
    if (lane == 0) {
        lpslam_hip_keypoint kp;
        const float sf = lt.scale[level];
        kp.x = (float)cx * sf; kp.y = (float)cy * sf;
        kp.size = (float)(unsigned)(31.0f * sf);
        kp.angle = angle; kp.response = score; kp.octave = level; kp.class_id = -1;
        kpts[o] = kp;
    }
}
// direct: workgroup = four keypoint slots of one image
__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe(const uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt,
                                                              const uint32_t* __restrict__ sel_key,
                                                              const int32_t* __restrict__ sel_count, int slots_per_image,
                                                              lpslam_hip_keypoint* __restrict__ kpts, uint8_t* __restrict__ desc,
                                                              int32_t* __restrict__ kp_count, ImgSel image0)
{
    describe_body(blockIdx.x, blockIdx.y, pyr, image_slab, lt, sel_key, sel_count, slots_per_image, kpts, desc, kp_count, image0);
}
// queued: chunks of slot blocks (slot block b of image i = item b + blocks * i)
__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe_q(const uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt,
                                                                const uint32_t* __restrict__ sel_key,
                                                                const int32_t* __restrict__ sel_count, int slots_per_image,
                                                                lpslam_hip_keypoint* __restrict__ kpts, uint8_t* __restrict__ desc,
                                                                int32_t* __restrict__ kp_count, ImgSel image0, FeQueue fq, int blocks_per_image)
{
    if (fe_leaves(fq)) return;
    for (;;) {
        const int first = fe_next_chunk(fq);
        if (first >= fq.n_items) break;
        for (int it = first; it < min(first + fq.chunk, fq.n_items); ++it)       // a wavefront's LDS is its own: no barrier between keypoints
            describe_body(it % blocks_per_image, it / blocks_per_image, pyr, image_slab, lt, sel_key, sel_count, slots_per_image, kpts, desc, kp_count, image0);
    }
}

// ------------------------------------------------------------------------------------------------------------
// K0: undistort / rectify.  cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with CV_32FC1 maps, which OpenCV first turns
// into fixed point (sx = cvRound(map_x * 32), integer part + 5-bit fraction; done once at lpslam_hip_set_rectify_map) and
// then interpolates with 15-bit weights (32 - fx)(32 - fy) * 32 ...: dst = (sum + 2^14) >> 15.
// Replaces the per-frame call of /root/reference/src/Utils/ImageProcessing.h:245-249.  One thread = 4 output pixels = one
// dword store; 6 map bytes + <= 4 gathered source bytes per pixel: HBM bound (8 B / pixel algorithmic).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_remap(const uint8_t* __restrict__ raw, const short2* __restrict__ map_xy,
                                               const uint16_t* __restrict__ map_frac, uint8_t* __restrict__ dst, int w, int h, int pitch)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= pitch || y >= h) return;
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = x0 + q;
        uint32_t val = 0;
        if (x < w) {
            const short2 xy = map_xy[(size_t)y * w + x];
            const int f = map_frac[(size_t)y * w + x], fx = f & 31, fy = f >> 5;
            const int sx = xy.x, sy = xy.y;
            if (!(sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0)) {
                auto tap = [&](int xx, int yy) -> int { return (xx >= 0 && yy >= 0 && xx < w && yy < h) ? (int)raw[(size_t)yy * w + xx] : 0; };
                const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
                const int sum = tap(sx, sy) * w00 + tap(sx + 1, sy) * w01 + tap(sx, sy + 1) * w10 + tap(sx + 1, sy + 1) * w11;
                val = (uint32_t)((sum + (1 << 14)) >> 15);
            }
        }
        out |= val << (8 * q);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x0) = out;
}

// ------------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------------
// the queue of one extraction launch under a mapping reserve (its counter zeroed on the stream in front of the launch); without a
// reserve: {nullptr, ...}, the kernels then run as plain grids
static FeQueue lp_fe_queue(lpslam_hip_ctx* c, int n_items, int chunk)
{
    FeQueue q{nullptr, nullptr, n_items, chunk, c->reserve_cus};
    if (c->reserve_cus <= 0 || !c->d_cu_table) return q;
    // a ring of 64 counters per stream (main / prefetch): a counter is zeroed and used on ONE stream, in order, so its re-use 64 launches
    // later is ordered behind the launch that used it before
    const bool on_prefetch = lp_fe_stream(c) != c->stream;
    int* ctr = c->d_fe_counters + 32 * ((on_prefetch ? 64 : 0) + (on_prefetch ? c->fe_counter_next_prefetch : c->fe_counter_next).fetch_add(1) % 64);
    if (hipMemsetAsync(ctr, 0, 2 * sizeof(int), lp_fe_stream(c)) != hipSuccess) { (void)hipGetLastError(); return q; }
    q.cu_table = c->d_cu_table; q.counter = ctr;
    return q;
}

// Calibration of the software reserve: where do workgroups land?  8192 one-wavefront workgroups report (XCC, hardware id); of the
// compute units seen on every XCC (32 on an MI355X, harvesting decides which ids) the r with the smallest ids are reserved.
int lp_fe_calibrate(lpslam_hip_ctx* c, int r)
{
    if (!c->d_cu_table) {
        LP_HIP(hipMalloc((void**)&c->d_cu_table, 64 * sizeof(uint32_t)));
        LP_HIP(hipMalloc((void**)&c->d_fe_counters, 2 * 64 * 32 * sizeof(int)));
    }
    std::vector<uint32_t> table(64, 0u);
    if (r > 0) {
        const int n = 8192;
        unsigned* d_out = nullptr;
        LP_HIP(hipMalloc((void**)&d_out, n * sizeof(unsigned)));
        hipLaunchKernelGGL(k_fe_where, dim3(n), dim3(64), 0, c->stream, d_out);
        std::vector<unsigned> h((size_t)n);
        hipError_t e = hipStreamSynchronize(c->stream);      // (a non-blocking stream: the copy below would not wait for it)
        if (e == hipSuccess) e = hipMemcpy(h.data(), d_out, n * sizeof(unsigned), hipMemcpyDeviceToHost);
        (void)hipFree(d_out);
        if (e != hipSuccess) { set_error("calibration of the mapping reserve failed"); return LPSLAM_HIP_ERR_DEVICE; }
        std::vector<std::vector<int>> ids(8);
        for (unsigned v : h) {
            const int xcc = (int)((v >> 16) & 7u), id = (int)((v >> 8) & 0xffu);
            if (std::find(ids[(size_t)xcc].begin(), ids[(size_t)xcc].end(), id) == ids[(size_t)xcc].end()) ids[(size_t)xcc].push_back(id);
        }
        for (int x = 0; x < 8; ++x) {
            std::sort(ids[(size_t)x].begin(), ids[(size_t)x].end());
            if ((int)ids[(size_t)x].size() < 2 * r) { set_error("mapping reserve %d: only %zu compute units of XCC %d were seen", r, ids[(size_t)x].size(), x); return LPSLAM_HIP_ERR_DEVICE; }
            // evenly over the shader engines (id = se << 5 | sh << 4 | cu): every engine's dispatcher keeps free compute units --
            // a workgroup handed to an engine whose units are all busy waits there, whatever is free elsewhere
            std::vector<std::vector<int>> by_se(8);
            for (int id : ids[(size_t)x]) by_se[(size_t)(id >> 5)].push_back(id);
            int chosen = 0;
            for (size_t depth = 0; chosen < r && depth < 32; ++depth)
                for (size_t se = 0; se < 8 && chosen < r; ++se)
                    if (depth < by_se[se].size()) { const int id = by_se[se][depth]; table[(size_t)(x * 8 + (id >> 5))] |= 1u << (id & 31); ++chosen; }
        }
    }
    LP_HIP(hipMemcpy(c->d_cu_table, table.data(), 64 * sizeof(uint32_t), hipMemcpyHostToDevice));
    return LPSLAM_HIP_OK;
}

int lp_fe_occupy_unreserved(lpslam_hip_ctx* c, int microseconds, int* landed)
{
    if (c->reserve_cus <= 0 || !c->d_cu_table) { set_error("debug_occupy_unreserved needs a mapping reserve"); return LPSLAM_HIP_ERR_INVALID; }
    if (microseconds < 0 || microseconds > 50000) { set_error("debug_occupy_unreserved: 0 .. 50000 us"); return LPSLAM_HIP_ERR_INVALID; }
    if (!c->debug_stream) LP_HIP(hipStreamCreateWithFlags(&c->debug_stream, hipStreamNonBlocking));
    const int lds = 160 * 1024;
    LP_HIP(hipFuncSetAttribute((const void*)k_fe_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int* d_landed = c->d_fe_counters + 2 * 64 * 32 - 1;            // the ring's last word: no queue uses it (a queue takes [0], [1] of its 32)
    LP_HIP(hipMemsetAsync(d_landed, 0, sizeof(int), c->debug_stream));
    hipLaunchKernelGGL(k_fe_occupy, dim3(256), dim3(1024), lds, c->debug_stream, c->d_cu_table, (unsigned long long)microseconds * 100ull, d_landed);
    LP_HIP(hipGetLastError());
    if (landed) {                                            // wait until the occupiers sit (they report at once), not until they are done
        *landed = 0;
        int h = 0, same = 0, last = -1;
        for (int spin = 0; spin < 2000 && same < 20; ++spin) {
            LP_HIP(hipMemcpyAsync(&h, d_landed, sizeof(int), hipMemcpyDeviceToHost, c->copy_stream ? c->copy_stream : c->stream));
            LP_HIP(hipStreamSynchronize(c->copy_stream ? c->copy_stream : c->stream));
            same = (h == last && h > 0) ? same + 1 : 0; last = h;
        }
        *landed = h;
    }
    return LPSLAM_HIP_OK;
}

int lp_launch_pyramid(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list)
{
    const ImgSel sel = lp_img_sel(first, n_images, list);
    if (c->lt.n_levels < 2 || n_images <= 0) return LPSLAM_HIP_OK;
    // one band work-group per CU (256 CUs, less what is reserved for the mapping solves): more bands would only add rows computed
    // twice -- or, with fewer CUs than work-groups, a second round --, fewer would leave CUs idle
    const int bands = std::max(4, std::min(kPyrMaxBands, (256 - 8 * c->reserve_cus) / n_images));
    const int set = bands;
    const int2* rows = c->d_band_rows + (size_t)set * kMaxLevels * kPyrMaxBands;
    const size_t lds = (size_t)c->rs_entries * sizeof(int2);
    FeQueue fq = lp_fe_queue(c, bands * n_images, 1);
    const dim3 grid = fq.cu_table ? dim3(256, 1) : dim3(bands, n_images);       // queued: one persistent workgroup per compute unit
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(k_pyr_bands<true>, grid, dim3(1024), lds, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, sel,
                           c->d_rs_pack, c->rs_entries, rows, fq, bands);
    else        // larger images: tables read through the cache instead
        hipLaunchKernelGGL(k_pyr_bands<false>, grid, dim3(1024), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, sel,
                           c->d_rs_pack, c->rs_entries, rows, fq, bands);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

int lp_launch_remap(lpslam_hip_ctx* c, int image, int eye)
{
    dim3 block(64, 4), grid((c->lt.pitch[0] / 4 + 63) / 64, (c->lt.h[0] + 3) / 4);
    hipLaunchKernelGGL(k_remap, grid, block, 0, lp_fe_stream(c), c->d_raw, c->d_map_xy[eye], c->d_map_frac[eye],
                       c->d_pyr + (size_t)image * c->image_slab, c->lt.w[0], c->lt.h[0], c->lt.pitch[0]);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

int lp_launch_fast(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list)
{
    const ImgSel sel = lp_img_sel(first, n_images, list);
    FeQueue fq = lp_fe_queue(c, c->cells_per_image * n_images, 1);       // one cell (~20 us of work) per fetch: chunks of 4 / 2 / 1 cells measured 1.083 / 1.064 / 1.047 ms per 16-frame step on half of the compute units -- the tail of an uneven last chunk costs more than the fetches
    if (fq.cu_table)                                     // queued: the eight workgroups a compute unit holds
        hipLaunchKernelGGL(k_fast_cells_q, dim3(256 * 8), dim3(256), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, c->cfg.ini_fast_threshold,
                           c->cfg.min_fast_threshold, c->d_cell_keys, c->d_cell_count, c->cells_per_image, sel, 0, c->d_mask[0], c->d_mask[1], fq);
    else
        hipLaunchKernelGGL(k_fast_cells, dim3(c->cells_per_image, n_images), dim3(256), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, c->cfg.ini_fast_threshold,
                           c->cfg.min_fast_threshold, c->d_cell_keys, c->d_cell_count, c->cells_per_image, sel, 0, c->d_mask[0], c->d_mask[1]);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

int lp_launch_distribute(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list)
{
    const ImgSel sel = lp_img_sel(first, n_images, list);
    FeQueue fq = lp_fe_queue(c, n_images * c->lt.n_levels, 1);
    if (fq.cu_table)
        hipLaunchKernelGGL(k_distribute_q, dim3(256), dim3(1024), c->distribute_lds, lp_fe_stream(c), c->lt, c->d_cell_keys, c->d_cell_count,
                           c->cells_per_image, c->d_cand_key, c->d_cand_node, c->d_cand_count, c->cand_per_image, c->d_sel_key, c->d_sel_count,
                           c->slots_per_image, sel, fq, n_images);
    else                                                 // level 0 of every image first: the long work-groups start first
        hipLaunchKernelGGL(k_distribute, dim3(n_images, c->lt.n_levels), dim3(1024), c->distribute_lds, lp_fe_stream(c), c->lt, c->d_cell_keys, c->d_cell_count,
                           c->cells_per_image, c->d_cand_key, c->d_cand_node, c->d_cand_count, c->cand_per_image, c->d_sel_key, c->d_sel_count,
                           c->slots_per_image, sel);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

int lp_launch_describe(lpslam_hip_ctx* c, int first, int n_images, const uint16_t* list)
{
    const ImgSel sel = lp_img_sel(first, n_images, list);
    if (!list) {
        for (int i = first; i < first + n_images && (size_t)i < c->h_kp_valid.size(); ++i) c->h_kp_valid[(size_t)i] = 0;      // counts are rewritten
        lp_pf_invalidate(c, first, n_images);
    }      // (a listed launch runs on the session pool's context: the sessions' own mirrors are invalidated by the caller, share.hip)
    const int blocks = (c->slots_per_image + DESC_WAVES - 1) / DESC_WAVES;
    FeQueue fq = lp_fe_queue(c, blocks * n_images, 2);         // (chunks of 32 / 16 / 8 / 4 / 2 / 1 workgroup-items: 1.18 / 1.055 / 1.039 / 1.033 / 1.027 / 1.039 ms per 16-frame step on half of the compute units)
    if (fq.cu_table)                                     // queued: seven workgroups (28 wavefronts) per compute unit
        hipLaunchKernelGGL(k_describe_q, dim3(256 * 7), dim3(64 * DESC_WAVES), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, c->d_sel_key,
                           c->d_sel_count, c->slots_per_image, c->d_kpts, c->d_desc, c->d_kp_count, sel, fq, blocks);
    else
        hipLaunchKernelGGL(k_describe, dim3(blocks, n_images), dim3(64 * DESC_WAVES), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, c->d_sel_key,
                           c->d_sel_count, c->slots_per_image, c->d_kpts, c->d_desc, c->d_kp_count, sel);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}
