// match.hip -- descriptor matching on gfx950: brute-force Hamming 2-NN and the stereo row-band matcher with SAD
// sub-pixel refinement.
//
// [UPSTREAM] openvslam::match::compute_descriptor_distance_32, match::robust (brute force), match::stereo::compute,
// reached from the reference through feed_stereo_frame (/root/reference/src/Trackers/OpenVSLAMStereoTracker.cpp:293-295);
// stereo parameters: focal_x_baseline (src/Trackers/OpenVSLAMTrackerBase.cpp:188-190, src/Interface/LpSlamTypes.h:219-222).
// Integer work (xor + popcount + argmin) is bit-exact; the few float operations are written without contraction.
#include "internal.h"
#include <chrono>
#include <vector>
#include <algorithm>
#include <cmath>

#pragma clang fp contract(off)

using namespace lpslam;

// ------------------------------------------------------------------------------------------------------------
// K7  brute-force Hamming 2-NN.  Workgroup = 16 queries (descriptor in 8 VGPRs), each spread over 16 lanes that split
//     each 1024-descriptor LDS tile of the train set; train descriptors are LDS broadcasts (ds_read_b128 x 2).
//     Integer-VALU bound: 8 v_xor + 8 v_bcnt per pair.  "First minimum wins", second = second smallest distance.
// ------------------------------------------------------------------------------------------------------------
#define BF_TILE 1024
#define BF_QPW 16                      // queries per workgroup

// merge of two partial 2-NN lists: best = smallest (distance, index); second = second smallest distance overall
__device__ __forceinline__ void bf_merge(int& b, int& s, int& bi, int wb, int ws, int wi)
{
    if (wi < 0) return;
    if (wb < b || (wb == b && wi < bi)) { s = min(s, b); b = wb; bi = wi; }
    else s = min(s, wb);
    s = min(s, ws);
}

// one workgroup's share of the brute-force 2-NN of nq query descriptors (qdesc) against nt train descriptors (tdesc): best index /
// best distance / second distance of query q into o[q], o[out_stride + q], o[2 out_stride + q]
__device__ __forceinline__ void bf_knn2_body(const uint8_t* __restrict__ qdesc, int nq, const uint8_t* __restrict__ tdesc, int nt, int32_t* __restrict__ o, int out_stride)
{
    // Workgroup = 16 queries; lane = (query, sub): the four wavefronts take the four quarters of each 1024-descriptor LDS tile
    // and the four subs of a wavefront the four sixteenths of a quarter, so a query is spread over 16 lanes and 2000 queries
    // make 125 workgroups per pair (the 64-queries-per-workgroup version left two thirds of the CUs idle).  A b128 LDS read
    // sees four distinct addresses per wavefront, one per 16-lane group: broadcast, conflict free.
    __shared__ __attribute__((aligned(16))) uint32_t tile[BF_TILE * 8];
    __shared__ int m_best[4][BF_QPW], m_second[4][BF_QPW], m_idx[4][BF_QPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ql = lane & 15, sub = lane >> 4;
    const int q = blockIdx.x * BF_QPW + ql;
    if (blockIdx.x * BF_QPW >= nq) return;                              // block-uniform
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(qdesc + (size_t)min(q, nq - 1) * 32);
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = qd[k];
    const uint4* td = reinterpret_cast<const uint4*>(tdesc);
    int best = 257, second = 257, bidx = -1;
    for (int base = 0; base < nt; base += BF_TILE) {
        const int n = min(BF_TILE, nt - base);
        __syncthreads();
        for (int i = threadIdx.x; i < n * 2; i += 256) reinterpret_cast<uint4*>(tile)[i] = td[(size_t)base * 2 + i];
        __syncthreads();
        const int jb = wave * (BF_TILE / 4) + sub * (BF_TILE / 16), je = min(jb + BF_TILE / 16, n);
        for (int j = jb; j < je; ++j) {
            const uint4 b0 = *reinterpret_cast<const uint4*>(&tile[j * 8]);
            const uint4 b1 = *reinterpret_cast<const uint4*>(&tile[j * 8 + 4]);
            int d = __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
                    __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
            if (d < best) { second = best; best = d; bidx = base + j; }
            else if (d < second) second = d;
        }
    }
    // the four subs of a wavefront (lanes ql, ql + 16, ql + 32, ql + 48), then the four wavefronts
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        const int wb = __shfl_xor(best, o), ws = __shfl_xor(second, o), wi = __shfl_xor(bidx, o);
        int b = best, s2 = second, bi = bidx;
        if (bi < 0) { b = 257; s2 = 257; }
        bf_merge(b, s2, bi, wb, ws, wi);
        best = b; second = s2; bidx = bi;
    }
    if (sub == 0) { m_best[wave][ql] = best; m_second[wave][ql] = second; m_idx[wave][ql] = bidx; }
    __syncthreads();
    if (threadIdx.x < BF_QPW && q < nq) {
        int b = 257, s = 257, bi = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w) bf_merge(b, s, bi, m_best[w][ql], m_second[w][ql], m_idx[w][ql]);
        o[q] = bi; o[out_stride + q] = b; o[2 * out_stride + q] = s;
    }
}


__global__ __launch_bounds__(256) void k_bf_knn2(const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts,
                                                 int slots_per_image, int q0, int t0, int stride, int32_t* __restrict__ out)
{
    const int pair = blockIdx.y;
    const int qs = q0 + pair * stride, ts = t0 + pair * stride;
    bf_knn2_body(desc + (size_t)qs * slots_per_image * 32, counts[qs], desc + (size_t)ts * slots_per_image * 32, counts[ts],
                 out + (size_t)qs * 3 * slots_per_image, slots_per_image);
}

// the same for a list of pairs whose descriptor sets live anywhere on the device (an image slot against stored keyframe descriptors,
// lpslam_hip_match_bf_stored); the list itself is read from page-locked host memory
struct BfPair { const uint8_t* q; const uint8_t* t; int32_t* out; int nq, nt, out_stride, pad_; };
__global__ __launch_bounds__(256) void k_bf_knn2_pairs(const BfPair* __restrict__ pairs)
{
    const BfPair p = pairs[blockIdx.y];
    bf_knn2_body(p.q, p.nq, p.t, p.nt, p.out, p.out_stride);
}

int lp_launch_bf_strided(lpslam_hip_ctx* c, int q0, int t0, int stride, int n_pairs)
{
    dim3 grid((c->slots_per_image + BF_QPW - 1) / BF_QPW, n_pairs);
    hipLaunchKernelGGL(k_bf_knn2, grid, dim3(256), 0, c->stream, c->d_desc, c->d_kp_count, c->slots_per_image, q0, t0, stride, c->d_bf);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// K8  stereo matcher: one wavefront per left keypoint.
//     (1) candidates = right keypoints whose row band [floor(y_r - 2 s_r), ceil(y_r + 2 s_r)] contains (int)y_l,
//         octave within +-1, x_r in [x_l - max_disp, x_l]; best Hamming < 75, first minimum in index order.
//     (2) 11x11 SAD (centre-subtracted, L1) over offsets -5..5 at the left keypoint's level, parabola fit.
//     (3) a second kernel applies the 2 x median correlation cut per image.
// ------------------------------------------------------------------------------------------------------------
#define ST_WAVES 4

// exclusive scan in place of a[0..n) by a 1024-thread workgroup
__device__ void block_scan_rows(int* a, int n, int* wave_tot /* LDS[16] */, int* total)
{
    const int per = (n + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(b + per, n);
    int s = 0;
    for (int i = b; i < e; ++i) s += a[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int t = wave_tot[w]; if (w < wave) base += t; tot += t; }
    base += incl - s;
    for (int i = b; i < e; ++i) { const int v = a[i]; a[i] = base; base += v; }
    __syncthreads();
    *total = tot;
}

// Row index of the right image ([UPSTREAM] match::stereo::get_right_keypoint_indices_in_each_row): keypoint j is a candidate for
// every row in [floor(y - r), ceil(y + r)], r = 2 x scale factor of its level.  One workgroup per stereo pair: histogram of the
// rows in LDS, scan, fill.  The order inside a row list is arbitrary (LDS atomics) and does not matter: the matcher takes the
// minimum of (distance << 16 | index).
__global__ __launch_bounds__(1024) void k_stereo_rows(LevelTable lt, const lpslam_hip_keypoint* __restrict__ kpts, const int32_t* __restrict__ counts,
                                                      int slots_per_image, int right0, int stride, ImgSel pairs, int32_t* __restrict__ row_start_all,
                                                      int32_t* __restrict__ row_list_all, int row_cap)
{
    extern __shared__ int st_rows[];                     // [H + 1] counts -> starts, [H] cursors
    __shared__ int wave_tot[16];
    const int H = lt.h[0];
    int* cnt = st_rows; int* cursor = st_rows + H + 1;
    const int right = pairs.listed ? lp_image(pairs, 2 * blockIdx.x + 1) : right0 + blockIdx.x * stride;      // listed: pair p = slots (list[2 p], list[2 p + 1])
    const int nr = counts[right];
    const lpslam_hip_keypoint* kr = kpts + (size_t)right * slots_per_image;
    int32_t* row_start = row_start_all + (size_t)right * (H + 1);
    int32_t* row_list = row_list_all + (size_t)right * row_cap;
    for (int r = threadIdx.x; r <= H; r += 1024) cnt[r] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < nr; j += 1024) {
        const lpslam_hip_keypoint k = kr[j];
        const float rad = 2.0f * lt.scale[k.octave];
        const int max_r = min((int)ceilf(k.y + rad), H - 1), min_r = max((int)floorf(k.y - rad), 0);
        for (int r = min_r; r <= max_r; ++r) atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    int total;
    block_scan_rows(cnt, H, wave_tot, &total);
    if (threadIdx.x == 0) cnt[H] = total;
    __syncthreads();
    for (int r = threadIdx.x; r <= H; r += 1024) { row_start[r] = cnt[r]; if (r < H) cursor[r] = cnt[r]; }
    __syncthreads();
    for (int j = threadIdx.x; j < nr; j += 1024) {
        const lpslam_hip_keypoint k = kr[j];
        const float rad = 2.0f * lt.scale[k.octave];
        const int max_r = min((int)ceilf(k.y + rad), H - 1), min_r = max((int)floorf(k.y - rad), 0);
        for (int r = min_r; r <= max_r; ++r) { const int pos = atomicAdd(&cursor[r], 1); if (pos < row_cap) row_list[pos] = j; }
    }
}

__global__ __launch_bounds__(64 * ST_WAVES) void k_stereo(const uint8_t* __restrict__ pyr, size_t image_slab, LevelTable lt,
                                                          const lpslam_hip_keypoint* __restrict__ kpts, const uint8_t* __restrict__ desc,
                                                          const int32_t* __restrict__ counts, int slots_per_image, int left0, int right0,
                                                          int stride, ImgSel pairs, float fxb, float max_disp, float* __restrict__ out_f,
                                                          int32_t* __restrict__ out_idx, int32_t* __restrict__ out_corr,
                                                          const int32_t* __restrict__ row_start_all, const int32_t* __restrict__ row_list_all, int row_cap)
{
    __shared__ uint8_t s_r[ST_WAVES][11 * 24];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int left = pairs.listed ? lp_image(pairs, 2 * blockIdx.y) : left0 + blockIdx.y * stride, right = pairs.listed ? lp_image(pairs, 2 * blockIdx.y + 1) : right0 + blockIdx.y * stride;
    const int i = blockIdx.x * ST_WAVES + wave;
    const int nl = counts[left], nr = counts[right];
    if (i >= nl) return;                                  // wave-uniform
    const lpslam_hip_keypoint kl = kpts[(size_t)left * slots_per_image + i];
    const lpslam_hip_keypoint* kr = kpts + (size_t)right * slots_per_image;
    const uint32_t* dl = reinterpret_cast<const uint32_t*>(desc + ((size_t)left * slots_per_image + i) * 32);
    const uint8_t* dr = desc + (size_t)right * slots_per_image * 32;
    float* o_xr = out_f + (size_t)left * 2 * slots_per_image;
    float* o_depth = o_xr + slots_per_image;
    int32_t* o_idx = out_idx + (size_t)left * slots_per_image;
    int32_t* o_corr = out_corr + (size_t)left * slots_per_image;

    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = dl[k];
    const int row = (int)kl.y;
    const float min_xr = kl.x - max_disp, max_xr = kl.x - 0.0f;
    unsigned best = 0xFFFFFFFFu;                          // (distance << 16) | index
    if (!(max_xr < 0) && nr > 0 && row >= 0 && row < lt.h[0]) {
        // candidates: the right keypoints whose row band covers this row (k_stereo_rows), not all of them
        const int32_t* row_start = row_start_all + (size_t)right * (lt.h[0] + 1);
        const int32_t* row_list = row_list_all + (size_t)right * row_cap;
        const int t_end = min(row_start[row + 1], row_cap);
        for (int t = row_start[row] + lane; t < t_end; t += 64) {
            const int j = row_list[t];
            const lpslam_hip_keypoint r = kr[j];
            if (r.octave < kl.octave - 1 || r.octave > kl.octave + 1) continue;
            if (r.x < min_xr || max_xr < r.x) continue;
            const uint32_t* b = reinterpret_cast<const uint32_t*>(dr + (size_t)j * 32);
            unsigned d = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += __popc(a[k] ^ b[k]);
            best = min(best, (d << 16) | (unsigned)j);
        }
    }
    for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
    const unsigned best_d = best >> 16;
    float res_xr = -1.0f, res_depth = -1.0f;
    int res_idx = -1, res_corr = -1;
    if (best != 0xFFFFFFFFu && best_d < 75u) {
        const int bj = (int)(best & 0xFFFFu);
        res_idx = bj;
        const int lvl = kl.octave;
        const float inv = lt.inv_scale[lvl], sc = lt.scale[lvl];
        const float x_right = kr[bj].x;
        const int sxl = (int)rintf(kl.x * inv), syl = (int)rintf(kl.y * inv), sxr = (int)rintf(x_right * inv);
        const int W = lt.w[lvl], P = lt.pitch[lvl];
        const int ini_x = sxr - 10, end_x = sxr + 11;
        if (!(ini_x < 0 || W <= end_x)) {
            const uint8_t* IL = pyr + (size_t)left * image_slab + lt.off[lvl];
            const uint8_t* IR = pyr + (size_t)right * image_slab + lt.off[lvl];
            // right strip 11 rows x 21 cols -> LDS; left patch values in registers (2 per lane)
            uint8_t* sr = s_r[wave];
            for (int t = lane; t < 11 * 21; t += 64) { const int r = t / 21, cc = t - r * 21; sr[r * 24 + cc] = IR[(size_t)(syl - 5 + r) * P + sxr - 10 + cc]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const int cl = IL[(size_t)syl * P + sxl];
            int lv[2], lr[2], lc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = lane + 64 * u;
                lr[u] = t / 11; lc[u] = t - lr[u] * 11;
                lv[u] = t < 121 ? (int)IL[(size_t)(syl - 5 + lr[u]) * P + sxl - 5 + lc[u]] - cl : 0;
            }
            float corr[11];
            float best_c = 4294967295.0f;
            int best_off = 0;
#pragma unroll
            for (int off = -5; off <= 5; ++off) {
                const int cr = sr[5 * 24 + 10 + off];
                int s = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = lane + 64 * u;
                    if (t < 121) s += abs(lv[u] - ((int)sr[lr[u] * 24 + 5 + off + lc[u]] - cr));
                }
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                const float c = (float)s;
                if (c < best_c) { best_c = c; best_off = off; }
                corr[off + 5] = c;
            }
            if (!(best_off == -5 || best_off == 5)) {
                float c1 = 0.f, c2 = 0.f, c3 = 0.f;
#pragma unroll
                for (int k = 1; k < 10; ++k) if (k == best_off + 5) { c1 = corr[k - 1]; c2 = corr[k]; c3 = corr[k + 1]; }
                const float x_delta = (float)((double)(c1 - c3) / (2.0 * ((double)(c1 + c3) - 2.0 * (double)c2)));
                if (!(x_delta < -1.0f || 1.0f < x_delta)) {
                    float bx = sc * ((float)(sxr + best_off) + x_delta);
                    float bd = kl.x - bx;
                    if (!(bd < 0.0f || max_disp <= bd)) {
                        if (bd <= 0.0f) { bd = 0.01f; bx = kl.x - bd; }
                        res_xr = bx; res_depth = fxb / bd; res_corr = (int)best_c;
                    }
                }
            }
        }
    }
    if (lane == 0) { o_xr[i] = res_xr; o_depth[i] = res_depth; o_idx[i] = res_idx; o_corr[i] = res_corr; }
}

// 2 x median cut: sort the correlations of the accepted matches of one image, read the median, invalidate weaker ones
__global__ __launch_bounds__(1024) void k_stereo_median(const int32_t* __restrict__ counts, int slots_per_image, int left0, int stride, ImgSel pairs,
                                                        float* __restrict__ out_f, const int32_t* __restrict__ corr)
{
    // median = element n / 2 of the sorted correlations, found by a four-pass radix select (8 bits per pass, LDS histogram +
    // one wavefront scan) instead of sorting them: the same value, 8 barriers instead of 66
    __shared__ int hist[256];
    __shared__ int s_n, s_prefix, s_target;
    const int left = pairs.listed ? lp_image(pairs, 2 * blockIdx.x) : left0 + blockIdx.x * stride;
    const int nl = counts[left];
    const int32_t* cr = corr + (size_t)left * slots_per_image;
    float* o_xr = out_f + (size_t)left * 2 * slots_per_image;
    float* o_depth = o_xr + slots_per_image;
    const int tid = threadIdx.x;
    if (tid == 0) s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 1024) mine += cr[i] >= 0 ? 1 : 0;
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    if (tid == 0) { s_prefix = 0; s_target = n / 2; }
    unsigned mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = (unsigned)s_prefix;
        for (int i = tid; i < nl; i += 1024) {
            const int v = cr[i];
            if (v >= 0 && ((unsigned)v & mask) == prefix) atomicAdd(&hist[((unsigned)v >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid < 64) {
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const int sum = h0 + h1 + h2 + h3;
            int incl = sum;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
            const int target = s_target, excl = incl - sum;
            if (excl <= target && target < incl) {                  // exactly one lane
                int r = target - excl, bin = 4 * tid;
                if (r >= h0) { r -= h0; ++bin; if (r >= h1) { r -= h1; ++bin; if (r >= h2) { r -= h2; ++bin; } } }
                s_prefix = (int)(prefix | ((unsigned)bin << shift));
                s_target = r;
            }
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    const float median = (float)s_prefix;
    const float thr = (float)(2.0 * (double)median);
    for (int i = tid; i < nl; i += 1024)
        if (cr[i] >= 0 && thr < (float)cr[i]) { o_xr[i] = -1.0f; o_depth[i] = -1.0f; }
}

int lp_launch_stereo_strided(lpslam_hip_ctx* c, int left0, int right0, int stride, int n_pairs, float fxb, float baseline, const uint16_t* pair_list)
{
    // pair_list: n_pairs (left, right) slot pairs in any order (the pending frames of several sessions in the session pool, share.hip)
    if (!pair_list) for (int i = 0; i < n_pairs; ++i) lp_pf_invalidate(c, left0 + i * stride, 1);        // the left slots' stereo columns are rewritten
    const ImgSel pairs = lp_img_sel(0, 2 * n_pairs, pair_list);
    const float max_disp = fxb / baseline;
    hipLaunchKernelGGL(k_stereo_rows, dim3(n_pairs), dim3(1024), (size_t)(2 * c->lt.h[0] + 2) * sizeof(int), lp_fe_stream(c), c->lt, c->d_kpts, c->d_kp_count,
                       c->slots_per_image, right0, stride, pairs, c->d_st_row_start, c->d_st_row_list, c->st_row_cap);
    dim3 grid((c->slots_per_image + ST_WAVES - 1) / ST_WAVES, n_pairs);
    hipLaunchKernelGGL(k_stereo, grid, dim3(64 * ST_WAVES), 0, lp_fe_stream(c), c->d_pyr, c->image_slab, c->lt, c->d_kpts, c->d_desc,
                       c->d_kp_count, c->slots_per_image, left0, right0, stride, pairs, fxb, max_disp, c->d_stereo, c->d_stereo_idx,
                       c->d_stereo_corr, c->d_st_row_start, c->d_st_row_list, c->st_row_cap);
    hipLaunchKernelGGL(k_stereo_median, dim3(n_pairs), dim3(1024), 0, lp_fe_stream(c), c->d_kp_count,
                       c->slots_per_image, left0, stride, pairs, c->d_stereo, c->d_stereo_corr);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------
// K8  projection matching ([UPSTREAM] match::projection::match_frame_and_landmarks / match_current_and_last_frames).
//     One wavefront per query: lanes stride over the image's keypoints, test the search window (|dx|, |dy| < radius, level
//     range, right-image x), compute the Hamming distance and keep their four smallest keys; the wavefront then merges them
//     into the query's four best candidates.  key = distance << 32 | grid cell << 20 | index << 4 | level, so "smallest key"
//     is upstream's "first strictly smaller distance in cell-scan order" (64 x 48 grid, column-major, index inside a cell).
//     `taken` keypoints are skipped before any distance is computed.  The sequential part of the algorithm (a keypoint
//     taken by an earlier query is invisible to later ones) is replayed on the host over these short lists.
// ------------------------------------------------------------------------------------------------------------
struct ProjQuery { float x, y, x_right, radius; int min_level, max_level; };
// mode 0: window + right-image window (projection / area matching); mode 1: chi-square gate of the keypoint's level (match::fuse)
struct ProjGate { int mode; float inv_sigma_sq[LPSLAM_HIP_MAX_LEVELS]; };

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// (a device function: launched as k_proj_topk for one call and as k_proj_topk_req, blockIdx.y = request, for the pending calls of
// several sessions at once)
__device__ __forceinline__ void proj_topk_body(const lpslam_hip_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                               const float* __restrict__ stereo_xr, const int32_t* __restrict__ kp_count,
                                               const ProjQuery* __restrict__ queries, const uint8_t* __restrict__ q_desc,
                                               const int* __restrict__ q_ids, int nq, const int16_t* __restrict__ best_so_far,
                                               float inv_w, float inv_h, const ProjGate& gate, unsigned long long* __restrict__ out_keys, int* __restrict__ out_count)
{
    // (out_keys / out_count may be page-locked host memory: with done_flag the kernel delivers the lists itself, lp_signal_done)
    const int lane = threadIdx.x & 63, qslot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qslot < nq) {
    const int qi = q_ids ? q_ids[qslot] : qslot;
    const ProjQuery q = queries[qi];
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(q_desc + 32 * (size_t)qi);
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = qd[k];
    const int n = *kp_count;
    const unsigned long long NONE = ~0ull;
    unsigned long long top[4] = {NONE, NONE, NONE, NONE};
    int cnt = 0;
    // Two passes, so that a wavefront waits for memory a handful of times instead of once per keypoint and once per candidate (2000
    // keypoints: 32 + up to 32 waits in a row, 19 us for a kernel that computes next to nothing).  Pass 1: sixteen keypoints of a lane in
    // flight at a time, the window / level test, survivors appended to the wavefront's list in LDS (ballot + prefix count).  Pass 2: a
    // lane per listed keypoint, everything it needs (stereo column, best-so-far, descriptor) loaded together.  Keys are unique and the
    // four smallest are wanted: the order candidates are met in does not matter.
    constexpr int PT_U = 16, PT_CAP = 2048;
    __shared__ uint16_t s_list[4][PT_CAP];
    uint16_t* list = s_list[threadIdx.x >> 6];
    int n_list = 0;                                                     // uniform over the wavefront
    auto second_pass = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < n_list; j += 64) {
            const int i = list[j];
            const lpslam_hip_keypoint* kpi = kp + i;
            struct { float x, y; int octave; } k{kpi->x, kpi->y, kpi->octave};
            const float xr_i = stereo_xr ? stereo_xr[i] : -1.0f;
            const int bsf_i = best_so_far ? (int)best_so_far[i] : 32767;
            const uint32_t* d = reinterpret_cast<const uint32_t*>(desc + 32 * (size_t)i);
            uint32_t dw[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) dw[w] = d[w];
            if (gate.mode == 0) {
                if (stereo_xr) { const float xr = xr_i; if (0 < xr && q.x_right >= 0 && q.radius < fabsf(q.x_right - xr)) continue; }
            } else {
                const float ex = q.x - k.x, ey = q.y - k.y;
                float isq = gate.inv_sigma_sq[0];
#pragma unroll
                for (int l = 1; l < LPSLAM_HIP_MAX_LEVELS; ++l) isq = k.octave == l ? gate.inv_sigma_sq[l] : isq;
                const float xr = xr_i;
                if (0 <= xr && q.x_right >= 0) { const float er = q.x_right - xr; if ((ex * ex + ey * ey + er * er) * isq > 7.81473f) continue; }
                else if ((ex * ex + ey * ey) * isq > 5.99146f) continue;
            }
            int dist = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) dist += __popc(a[w] ^ dw[w]);
            if (bsf_i <= dist) continue;                                // taken (0), or already matched at an equal or smaller distance
            int cx = (int)floorf(k.x * inv_w), cy = (int)floorf(k.y * inv_h);
            cx = min(max(cx, 0), 63); cy = min(max(cy, 0), 47);
            unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)(cx * 48 + cy) << 20) | ((unsigned long long)i << 4) | (unsigned)k.octave;
            ++cnt;
#pragma unroll
            for (int t = 0; t < 4; ++t) if (key < top[t]) { const unsigned long long tmp = top[t]; top[t] = key; key = tmp; }     // sorted insert
        }
        __builtin_amdgcn_wave_barrier();
        n_list = 0;
    };
    for (int i0 = lane; i0 - lane < n; i0 += 64 * PT_U) {               // (uniform trip count: the ballots below want every lane)
        float kxs[PT_U], kys[PT_U]; int ocs[PT_U];
#pragma unroll
        for (int u = 0; u < PT_U; ++u) {
            const lpslam_hip_keypoint* p = kp + min(i0 + 64 * u, n - 1);
            kxs[u] = p->x; kys[u] = p->y; ocs[u] = p->octave;
        }
        if (n_list + 64 * PT_U > PT_CAP) second_pass();
#pragma unroll
        for (int u = 0; u < PT_U; ++u) {
            const int i = i0 + 64 * u;
            bool pass = i < n && fabsf(kxs[u] - q.x) < q.radius && fabsf(kys[u] - q.y) < q.radius;
            if (q.min_level >= 0 && ocs[u] < q.min_level) pass = false;
            if (q.max_level >= 0 && ocs[u] > q.max_level) pass = false;
            const unsigned long long b = __ballot(pass);
            if (pass) list[n_list + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = (uint16_t)i;
            n_list += __popcll(b);
        }
    }
    second_pass();
    for (int o2 = 32; o2 > 0; o2 >>= 1) cnt += __shfl_xor(cnt, o2);
    unsigned long long res[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned long long m = wave_min_u64(top[0]);
        res[r] = m;
        if (top[0] == m && m != NONE) { top[0] = top[1]; top[1] = top[2]; top[2] = top[3]; top[3] = NONE; }      // keys are unique (index bits)
    }
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out_keys[4 * (size_t)qslot + r] = res[r];
        out_count[qslot] = cnt;
    }
    }
}

__global__ __launch_bounds__(256) void k_proj_topk(const lpslam_hip_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
                                                   const float* __restrict__ stereo_xr, const int32_t* __restrict__ kp_count,
                                                   const ProjQuery* __restrict__ queries, const uint8_t* __restrict__ q_desc,
                                                   const int* __restrict__ q_ids, int nq, const int16_t* __restrict__ best_so_far,
                                                   float inv_w, float inv_h, ProjGate gate, unsigned long long* __restrict__ out_keys, int* __restrict__ out_count,
                                                   unsigned* done_counter, int* done_flag, int done_seq)
{
    proj_topk_body(kp, desc, stereo_xr, kp_count, queries, q_desc, q_ids, nq, best_so_far, inv_w, inv_h, gate, out_keys, out_count);
    if (done_flag) lp_signal_done(done_counter, done_flag, done_seq);
}

// The first scan of several window-matcher calls in one launch: blockIdx.y = request, the request table in page-locked memory (the
// combiner of share.hip writes it), every request delivers its own lists and releases its own flag.
static_assert(sizeof(LpProjGate) == sizeof(ProjGate), "gate layout");
__global__ __launch_bounds__(256) void k_proj_topk_req(const LpProjReq* __restrict__ table)
{
    const LpProjReq* r = table + blockIdx.y;
    const int grid_x = r->grid_x;
    if ((int)blockIdx.x >= grid_x) return;
    const ProjGate& gate = *reinterpret_cast<const ProjGate*>(&r->gate);
    proj_topk_body((const lpslam_hip_keypoint*)r->kp, r->desc, r->stereo_xr, r->kp_count, (const ProjQuery*)r->queries, r->q_desc, (const int*)nullptr, r->nq,
                   r->best_so_far, r->inv_w, r->inv_h, gate, r->out_keys, r->out_count);
    // lp_signal_done with this request's own row of workgroups as the total
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (grid_x == 1 || atomicAdd(r->done_counter, 1u) == (unsigned)grid_x - 1) {
            if (grid_x > 1) { *r->done_counter = 0; __threadfence(); }
            __hip_atomic_store(r->done_flag, r->done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

extern "C" {

static int chk(lpslam_hip_ctx* c, int a, int b)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    if (a < 0 || a >= c->cfg.max_images || b < 0 || b >= c->cfg.max_images) {
        set_error("image slot out of range [0,%d)", c->cfg.max_images); return LPSLAM_HIP_ERR_CAPACITY;
    }
    LP_HIP(hipSetDevice(c->cfg.device));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_match_bf(lpslam_hip_ctx* c, int query, int train)
{
    int rc = chk(c, query, train); if (rc) return rc;
    return lp_launch_bf_strided(c, query, train, 0, 1);
}

int lpslam_hip_match_bf_strided(lpslam_hip_ctx* c, int query0, int train0, int stride, int n_pairs)
{
    if (n_pairs < 1) { set_error("n_pairs < 1"); return LPSLAM_HIP_ERR_INVALID; }
    int rc = chk(c, query0, train0); if (rc) return rc;
    if ((rc = chk(c, query0 + (n_pairs - 1) * stride, train0 + (n_pairs - 1) * stride))) return rc;
    return lp_launch_bf_strided(c, query0, train0, stride, n_pairs);
}

int lpslam_hip_get_bf_knn2(lpslam_hip_ctx* c, int query, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                           int32_t capacity, int32_t* count)
{
    int32_t n = 0;
    int rc = lpslam_hip_keypoint_count(c, query, &n); if (rc) return rc;
    if (count) *count = n;
    if (n > capacity) { set_error("result buffer too small (%d < %d)", capacity, n); return LPSLAM_HIP_ERR_CAPACITY; }
    const int32_t* src = c->d_bf + (size_t)query * 3 * c->slots_per_image;
    if (n && best_idx) LP_HIP(hipMemcpyAsync(best_idx, src, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (n && best_dist) LP_HIP(hipMemcpyAsync(best_dist, src + c->slots_per_image, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (n && second_dist) LP_HIP(hipMemcpyAsync(second_dist, src + 2 * c->slots_per_image, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_get_bf_matches(lpslam_hip_ctx* c, int query, int train, int32_t max_dist, float ratio, int32_t cross_check,
                              int32_t* out_q, int32_t* out_t, int32_t* out_d, int32_t capacity, int32_t* count)
{
    int rc = chk(c, query, train); if (rc) return rc;
    int32_t nq = 0, nt = 0;
    if ((rc = lpslam_hip_keypoint_count(c, query, &nq))) return rc;
    if ((rc = lpslam_hip_keypoint_count(c, train, &nt))) return rc;
    std::vector<int32_t> bi(nq), bd(nq), sd(nq), rbi;
    if ((rc = lpslam_hip_get_bf_knn2(c, query, bi.data(), bd.data(), sd.data(), nq, nullptr))) return rc;
    if (cross_check) {       // reverse direction (train -> query) computed on the device as well
        if ((rc = lp_launch_bf_strided(c, train, query, 0, 1))) return rc;
        rbi.resize(nt);
        if ((rc = lpslam_hip_get_bf_knn2(c, train, rbi.data(), nullptr, nullptr, nt, nullptr))) return rc;
    }
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (bi[i] < 0) continue;
        if (bd[i] > max_dist) continue;
        if (ratio > 0.f && ratio * (float)sd[i] < (float)bd[i]) continue;
        if (cross_check && rbi[bi[i]] != i) continue;
        if (n >= capacity) { set_error("match buffer too small"); return LPSLAM_HIP_ERR_CAPACITY; }
        out_q[n] = i; out_t[n] = bi[i]; out_d[n] = bd[i]; ++n;
    }
    if (count) *count = n;
    return LPSLAM_HIP_OK;
}

}  // extern "C"

namespace {
// up to four runs of 32-bit words into page-locked host memory, then the completion flag (internal.h, lp_signal_done)
struct WordRuns { const uint32_t* src[4]; int n[4]; int dst[4]; };
__global__ __launch_bounds__(256) void k_words_to_host(WordRuns r, uint32_t* __restrict__ st, unsigned* counter, int* flag, int seq)
{
    const int total = r.n[0] + r.n[1] + r.n[2] + r.n[3];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int j = i, k = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) if (k == q && j >= r.n[q]) { j -= r.n[q]; k = q + 1; }
        st[r.dst[k] + j] = r.src[k][j];
    }
    lp_signal_done(counter, flag, seq);
}
}  // namespace

extern "C" {

int lpslam_hip_match_bf_descriptors(lpslam_hip_ctx* c, int query, int scratch, const uint8_t* train_desc32, int32_t n_train, int32_t max_dist, float ratio,
                                    int32_t cross_check, int32_t* out_q, int32_t* out_t, int32_t* out_d, int32_t capacity, int32_t* count)
{
    int rc = chk(c, query, scratch); if (rc) return rc;
    if (query == scratch || n_train < 0 || n_train > c->slots_per_image || (n_train && !train_desc32) || !out_q || !out_t || !out_d) { set_error("bad match_bf_descriptors arguments"); return LPSLAM_HIP_ERR_INVALID; }
    int32_t nq = 0;
    if ((rc = lpslam_hip_keypoint_count(c, query, &nq))) return rc;
    if (count) *count = 0;
    hipStream_t s = c->stream;
    const size_t S = (size_t)c->slots_per_image;
    // the scratch slot receives the descriptors and their count (lpslam_hip_set_descriptors, without its wait)
    if (n_train) LP_HIP(hipMemcpyAsync(c->d_desc + (size_t)scratch * S * 32, train_desc32, (size_t)n_train * 32, hipMemcpyHostToDevice, s));
    LP_HIP(hipMemcpyAsync(c->d_kp_count + scratch, &n_train, sizeof(int32_t), hipMemcpyHostToDevice, s));
    if ((size_t)scratch < c->h_kp_valid.size()) { c->h_kp_count[(size_t)scratch] = n_train; c->h_kp_valid[(size_t)scratch] = 1; }
    lp_pf_invalidate(c, scratch, 1);
    if (nq == 0 || n_train == 0) { LP_HIP(hipStreamSynchronize(s)); return LPSLAM_HIP_OK; }
    if ((rc = lp_launch_bf_strided(c, query, scratch, 0, 1))) return rc;
    if (cross_check && (rc = lp_launch_bf_strided(c, scratch, query, 0, 1))) return rc;
    // page-locked block: flag | best index, best distance, second distance of the queries | best index of the train side
    const size_t o_bi = 16, o_bd = o_bi + (size_t)nq, o_sd = o_bd + (size_t)nq, o_rbi = o_sd + (size_t)nq, words = o_rbi + (size_t)n_train;
    if (c->h_match_bytes < words * 4) {
        if (c->h_match) { LP_HIP(hipStreamSynchronize(s)); (void)hipHostFree(c->h_match); }
        c->h_match = nullptr; c->h_match_bytes = 0;
        LP_HIP(hipHostMalloc((void**)&c->h_match, words * 8, hipHostMallocDefault));
        c->h_match_bytes = words * 8;
    }
    unsigned* done_counter = lp_done_counter(c, 2);
    if (!done_counter) { set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    uint32_t* st = (uint32_t*)c->h_match;
    int* flag = (int*)st;
    const int seq = lp_next_seq(c->done_seq);
    __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
    const uint32_t* fq = (const uint32_t*)(c->d_bf + (size_t)query * 3 * S);
    const uint32_t* ft = (const uint32_t*)(c->d_bf + (size_t)scratch * 3 * S);
    WordRuns r{{fq, fq + S, fq + 2 * S, ft}, {nq, nq, nq, cross_check ? n_train : 0}, {(int)o_bi, (int)o_bd, (int)o_sd, (int)o_rbi}};
    hipLaunchKernelGGL(k_words_to_host, dim3(std::max(1, std::min(16, (int)(words / 1024) + 1))), dim3(256), 0, s, r, st, done_counter, flag, seq);
    LP_HIP(hipGetLastError());
    if (!lp_wait_done(flag, seq, s)) { (void)lp_wait_recover(c, 2, s); set_error("lpslam_hip_match_bf_descriptors: the read-back kernel did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    const int32_t* bi = (const int32_t*)(st + o_bi); const int32_t* bd = (const int32_t*)(st + o_bd); const int32_t* sd = (const int32_t*)(st + o_sd);
    const int32_t* rbi = (const int32_t*)(st + o_rbi);
    int n = 0;
    for (int i = 0; i < nq; ++i) {       // the filter of lpslam_hip_get_bf_matches
        if (bi[i] < 0) continue;
        if (bd[i] > max_dist) continue;
        if (ratio > 0.f && ratio * (float)sd[i] < (float)bd[i]) continue;
        if (cross_check && rbi[bi[i]] != i) continue;
        if (n >= capacity) { set_error("match buffer too small"); return LPSLAM_HIP_ERR_CAPACITY; }
        out_q[n] = i; out_t[n] = bi[i]; out_d[n] = bd[i]; ++n;
    }
    if (count) *count = n;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_desc_store_put(lpslam_hip_ctx* c, int32_t key, const uint8_t* desc32, int32_t n)
{
    if (!c || n < 0 || (n > 0 && !desc32)) { set_error("bad desc_store_put arguments"); return LPSLAM_HIP_ERR_INVALID; }
    LP_HIP(hipSetDevice(c->cfg.device));
    lpslam_hip_ctx::StoredDesc& e = c->desc_store[key];
    const size_t bytes = (size_t)std::max(n, 1) * 32;
    hipStream_t aux = lp_aux_stream(c);                  // (the stream lpslam_hip_match_bf_stored reads the sets on)
    if (e.cap < bytes) {
        if (e.blk) { LP_HIP(hipStreamSynchronize(aux)); lp_pool_free(c, e.blk, e.cap); e.blk = nullptr; e.cap = 0; }
        const int rc = lp_pool_alloc(c, bytes, &e.blk, &e.cap);
        if (rc) { c->desc_store.erase(key); return rc; }
    }
    e.n = n;
    if (n) LP_HIP(hipMemcpyAsync(e.blk, desc32, (size_t)n * 32, hipMemcpyHostToDevice, aux));
    return LPSLAM_HIP_OK;
}

int lpslam_hip_desc_store_drop(lpslam_hip_ctx* c, int32_t key)
{
    if (!c) { set_error("null context"); return LPSLAM_HIP_ERR_INVALID; }
    auto it = c->desc_store.find(key);
    if (it == c->desc_store.end()) return LPSLAM_HIP_OK;
    if (it->second.blk) { LP_HIP(hipStreamSynchronize(lp_aux_stream(c))); lp_pool_free(c, it->second.blk, it->second.cap); }
    c->desc_store.erase(it);
    return LPSLAM_HIP_OK;
}

}  // extern "C"

namespace {
// runs of 32-bit words listed in page-locked memory -> page-locked memory, one run per blockIdx.y
struct WordRun { const uint32_t* src; int n, dst; };
__global__ __launch_bounds__(256) void k_runs_to_host(const WordRun* __restrict__ runs, uint32_t* __restrict__ st, unsigned* counter, int* flag, int seq)
{
    const WordRun r = runs[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r.n; i += gridDim.x * blockDim.x) st[r.dst + i] = r.src[i];
    lp_signal_done(counter, flag, seq);
}
}  // namespace

extern "C" {

int lpslam_hip_match_bf_stored(lpslam_hip_ctx* c, int query, const int32_t* keys, int32_t n_keys, int32_t max_dist, float ratio, int32_t cross_check,
                               int32_t* out_q, int32_t* out_t, int32_t* out_d, int32_t capacity_per_key, int32_t* counts)
{
    int rc = chk(c, query, query); if (rc) return rc;
    if (n_keys < 0 || (n_keys > 0 && (!keys || !out_q || !out_t || !out_d || !counts)) || capacity_per_key < 0) { set_error("bad match_bf_stored arguments"); return LPSLAM_HIP_ERR_INVALID; }
    for (int k = 0; k < n_keys; ++k) { counts[k] = 0; if (!c->desc_store.count(keys[k])) { set_error("descriptor set %d is not in the store", keys[k]); return LPSLAM_HIP_ERR_INVALID; } }
    int32_t nq = 0;
    if ((rc = lpslam_hip_keypoint_count(c, query, &nq))) return rc;
    if (n_keys == 0 || nq == 0) return LPSLAM_HIP_OK;
    hipStream_t s = lp_aux_stream(c);
    const size_t S = (size_t)c->slots_per_image;
    // device block: per key the forward arrays (best index, best distance, second distance of the nq queries) and the reverse best index
    std::vector<size_t> o_fwd((size_t)n_keys), o_rev((size_t)n_keys);
    size_t dev_words = 0, host_words = 16;
    int nt_max = 1;
    for (int k = 0; k < n_keys; ++k) {
        const int nt = c->desc_store[keys[k]].n;
        nt_max = std::max(nt_max, nt);
        o_fwd[(size_t)k] = dev_words; dev_words += 3 * (size_t)nq;
        o_rev[(size_t)k] = dev_words; dev_words += 3 * (size_t)std::max(nt, 1);
        host_words += 3 * (size_t)nq + (size_t)(cross_check ? nt : 0);
    }
    // page-locked block: flag | pair list | run list | results
    const size_t o_pairs = 64, o_runs = o_pairs + 2 * (size_t)n_keys * sizeof(BfPair), o_res = (o_runs + 2 * (size_t)n_keys * sizeof(WordRun) + 63) & ~(size_t)63;
    const size_t host_bytes = o_res + host_words * 4;
    if (c->h_match_bytes < host_bytes) {
        // sized at once for a search over 48 full sets (a tracker's candidate list grows one keyframe at a time: growing the block with
        // it cost a page-locked reallocation -- 1 ms -- on every frame that crossed the previous size)
        const size_t roomy = std::max(host_bytes * 2, (size_t)4096 + 96 * (sizeof(BfPair) + sizeof(WordRun)) + (size_t)48 * 4 * S * 4);
        if (c->h_match) { LP_HIP(hipStreamSynchronize(s)); (void)hipHostFree(c->h_match); }
        c->h_match = nullptr; c->h_match_bytes = 0;
        LP_HIP(hipHostMalloc((void**)&c->h_match, roomy, hipHostMallocDefault));
        c->h_match_bytes = roomy;
    }
    void* blk = nullptr; size_t cap = 0;
    // (always the size of a 48-set search: the context's block cache then serves every call from the same block -- a request that
    // grows by one keyframe per call would miss it every time, 1 ms of hipMalloc)
    if ((rc = lp_pool_alloc(c, std::max(dev_words * 4, (size_t)48 * 6 * S * 4), &blk, &cap))) return rc;
    auto release = [&]() { lp_pool_free(c, blk, cap); };
    unsigned* done_counter = lp_done_counter(c, 2);
    if (!done_counter) { release(); set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    uint8_t* hb = c->h_match;
    BfPair* pairs = (BfPair*)(hb + o_pairs); WordRun* runs = (WordRun*)(hb + o_runs);
    int32_t* d_res = (int32_t*)blk;
    const uint8_t* qdesc = c->d_desc + (size_t)query * S * 32;
    size_t w = 0;
    std::vector<size_t> h_fwd((size_t)n_keys), h_rev((size_t)n_keys);
    int n_pairs = 0, n_runs = 0;
    for (int k = 0; k < n_keys; ++k) {
        const lpslam_hip_ctx::StoredDesc& e = c->desc_store[keys[k]];
        pairs[n_pairs++] = BfPair{qdesc, (const uint8_t*)e.blk, d_res + o_fwd[(size_t)k], nq, e.n, nq, 0};
        if (cross_check) pairs[n_pairs++] = BfPair{(const uint8_t*)e.blk, qdesc, d_res + o_rev[(size_t)k], e.n, nq, std::max(e.n, 1), 0};
        h_fwd[(size_t)k] = o_res / 4 + w;
        runs[n_runs++] = WordRun{(const uint32_t*)(d_res + o_fwd[(size_t)k]), 3 * nq, (int)(o_res / 4 + w)}; w += 3 * (size_t)nq;
        h_rev[(size_t)k] = o_res / 4 + w;
        if (cross_check && e.n) { runs[n_runs++] = WordRun{(const uint32_t*)(d_res + o_rev[(size_t)k]), e.n, (int)(o_res / 4 + w)}; w += (size_t)e.n; }
    }
    // a set without descriptors leaves its forward arrays unwritten by the kernel (no train descriptor: best index -1): preset them
    for (int k = 0; k < n_keys; ++k)
        if (c->desc_store[keys[k]].n == 0) {
            if (hipMemsetAsync(d_res + o_fwd[(size_t)k], 0xff, (size_t)nq * 4, s) != hipSuccess) { release(); set_error("hipMemsetAsync failed"); return LPSLAM_HIP_ERR_DEVICE; }
        }
    int* flag = (int*)hb;
    const int seq = lp_next_seq(c->done_seq);
    __atomic_store_n(flag, 0, __ATOMIC_RELAXED);
    const int q_blocks = (std::max(nq, nt_max) + BF_QPW - 1) / BF_QPW;
    hipLaunchKernelGGL(k_bf_knn2_pairs, dim3(q_blocks, n_pairs), dim3(256), 0, s, (const BfPair*)pairs);
    hipLaunchKernelGGL(k_runs_to_host, dim3(4, n_runs), dim3(256), 0, s, (const WordRun*)runs, (uint32_t*)hb, done_counter, flag, seq);
    if (hipGetLastError() != hipSuccess) { release(); set_error("launch failed"); return LPSLAM_HIP_ERR_DEVICE; }
    // (a failed wait: the block goes back to the pool only when the stream still synchronises -- its kernels are then complete --, and the
    // arrival counter they left part-way is zeroed; on a dead stream the block is leaked rather than handed to the next caller)
    if (!lp_wait_done(flag, seq, s)) { if (lp_wait_recover(c, 2, s)) release(); set_error("lpslam_hip_match_bf_stored: the kernels did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    release();
    const int32_t* hw = (const int32_t*)hb;
    for (int k = 0; k < n_keys; ++k) {
        const int nt = c->desc_store[keys[k]].n;
        if (nt == 0) continue;
        const int32_t* bi = hw + h_fwd[(size_t)k]; const int32_t* bd = bi + nq; const int32_t* sd = bd + nq;
        const int32_t* rbi = hw + h_rev[(size_t)k];
        int n = 0;
        for (int i = 0; i < nq; ++i) {       // the filter of lpslam_hip_get_bf_matches
            if (bi[i] < 0) continue;
            if (bd[i] > max_dist) continue;
            if (ratio > 0.f && ratio * (float)sd[i] < (float)bd[i]) continue;
            if (cross_check && rbi[bi[i]] != i) continue;
            if (n >= capacity_per_key) { set_error("match buffer too small"); return LPSLAM_HIP_ERR_CAPACITY; }
            const size_t at = (size_t)k * capacity_per_key + n;
            out_q[at] = i; out_t[at] = bi[i]; out_d[at] = bd[i]; ++n;
        }
        counts[k] = n;
    }
    return LPSLAM_HIP_OK;
}

int lpslam_hip_match_stereo(lpslam_hip_ctx* c, int left, int right, float fxb, float baseline)
{
    int rc = chk(c, left, right); if (rc) return rc;
    if (!(baseline > 0.f) || !(fxb > 0.f)) { set_error("focal_x_baseline and baseline must be positive"); return LPSLAM_HIP_ERR_INVALID; }
    return lp_launch_stereo_strided(c, left, right, 0, 1, fxb, baseline);
}

int lpslam_hip_match_stereo_strided(lpslam_hip_ctx* c, int left0, int right0, int stride, int n_pairs, float fxb, float baseline)
{
    if (n_pairs < 1) { set_error("n_pairs < 1"); return LPSLAM_HIP_ERR_INVALID; }
    int rc = chk(c, left0, right0); if (rc) return rc;
    if ((rc = chk(c, left0 + (n_pairs - 1) * stride, right0 + (n_pairs - 1) * stride))) return rc;
    if (!(baseline > 0.f) || !(fxb > 0.f)) { set_error("focal_x_baseline and baseline must be positive"); return LPSLAM_HIP_ERR_INVALID; }
    return lp_launch_stereo_strided(c, left0, right0, stride, n_pairs, fxb, baseline);
}

int lpslam_hip_get_stereo(lpslam_hip_ctx* c, int left, float* stereo_x_right, float* depths, int32_t* best_right_idx,
                          int32_t capacity, int32_t* count)
{
    int32_t n = 0;
    int rc = lpslam_hip_keypoint_count(c, left, &n); if (rc) return rc;
    if (count) *count = n;
    if (n > capacity) { set_error("result buffer too small (%d < %d)", capacity, n); return LPSLAM_HIP_ERR_CAPACITY; }
    const float* f = c->d_stereo + (size_t)left * 2 * c->slots_per_image;
    if (n && stereo_x_right) LP_HIP(hipMemcpyAsync(stereo_x_right, f, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (n && depths) LP_HIP(hipMemcpyAsync(depths, f + c->slots_per_image, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (n && best_right_idx) LP_HIP(hipMemcpyAsync(best_right_idx, c->d_stereo_idx + (size_t)left * c->slots_per_image, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    LP_HIP(hipStreamSynchronize(c->stream));
    return LPSLAM_HIP_OK;
}

// ---- window matchers: device top-4 per query + the order-dependent part replayed on the host ----------------------------------
// policy 0  match::projection: a keypoint taken by an earlier query is invisible (best_so_far = 0), Lowe ratio between equal levels
// policy 1  match::fuse:       no exclusivity, chi-square gate in the kernel, best only
// policy 2  match::area:       a keypoint matched at an equal or smaller distance is invisible; strict ratio; a better later
//                              query takes the keypoint away from the earlier one
static int window_match(lpslam_hip_ctx* c, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                        int32_t hamming_thr, float lowe_ratio, const uint8_t* taken_in, int32_t use_stereo, int policy,
                        int32_t* match_idx, int32_t* match_dist, int32_t* n_matches)
{
    int rc = chk(c, image, image); if (rc) return rc;
    if (nq < 0 || (nq > 0 && (!queries || !q_desc32 || !match_idx))) { set_error("bad window-match arguments"); return LPSLAM_HIP_ERR_INVALID; }
    static const bool trace = getenv("LPSLAM_HIP_MATCH_TRACE") != nullptr;
    const auto tr0 = std::chrono::steady_clock::now();
    auto tr_us = [&tr0]() { return 1e-3 * (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tr0).count(); };
    double tr_alloc = 0, tr_launch = 0, tr_wait = 0; int tr_rescans = 0;
    static_assert(sizeof(lpslam_hip_proj_query) == sizeof(ProjQuery), "query layout");
    if (n_matches) *n_matches = 0;
    if (nq == 0) return LPSLAM_HIP_OK;
    hipStream_t s = c->stream;
    const size_t o = (size_t)image * c->slots_per_image;
    // one block of the context's cache: keys | queries | descriptors | counts | ids | best-so-far (one entry per keypoint SLOT:
    // the keypoint count stays on the device, the kernel reads it there, so no round trip is needed before the launch)
    const size_t nk = (size_t)std::max(c->slots_per_image, 1);
    const size_t o_keys = 0, o_q = o_keys + (size_t)nq * 4 * sizeof(unsigned long long), o_qd = o_q + (size_t)nq * sizeof(ProjQuery), o_cnt = o_qd + (size_t)nq * 32,
                 o_ids = o_cnt + (size_t)nq * sizeof(int), o_bsf = o_ids + 64;
    const size_t total = o_bsf + nk * sizeof(int16_t);
    void* blk = nullptr; size_t cap = 0;
    { const int rc2 = lp_pool_alloc(c, total, &blk, &cap); if (rc2) return rc2; }
    auto release = [&]() { lp_pool_free(c, blk, cap); };
#define P_HIP(x) do { if ((x) != hipSuccess) { release(); set_error("HIP call failed: %s", #x); return LPSLAM_HIP_ERR_DEVICE; } } while (0)
    // pinned mirror of that block on the host: inputs are assembled there and go down as ONE copy (queries, descriptors,
    // counts, ids and best-so-far are contiguous), the candidate lists come back as one copy
    if (c->h_match_bytes < total) {
        if (c->h_match) (void)hipHostFree(c->h_match);
        c->h_match = nullptr; c->h_match_bytes = 0;
        P_HIP(hipHostMalloc((void**)&c->h_match, total + total / 2, hipHostMallocDefault));
        c->h_match_bytes = total + total / 2;
    }
    tr_alloc = tr_us();
    uint8_t* base = (uint8_t*)blk;
    uint8_t* hb = c->h_match;
    unsigned long long* d_keys = (unsigned long long*)(base + o_keys); ProjQuery* d_q = (ProjQuery*)(base + o_q); uint8_t* d_qd = base + o_qd;
    int* d_cnt = (int*)(base + o_cnt); int* d_ids = (int*)(base + o_ids); int16_t* d_bsf = (int16_t*)(base + o_bsf);
    int16_t* bsf = (int16_t*)(hb + o_bsf);
    for (size_t i = 0; i < nk; ++i) bsf[i] = (int16_t)32767;
    if (taken_in) {                        // one byte per keypoint of the image: the count is needed (host mirror, else one round trip)
        int32_t n_kp = 0;
        { const int rc3 = lpslam_hip_keypoint_count(c, image, &n_kp); if (rc3) { release(); return rc3; } }
        for (int i = 0; i < n_kp; ++i) if (taken_in[i]) bsf[(size_t)i] = 0;
    }
    std::vector<int> owner(policy == 2 ? nk : 0, -1);
    memcpy(hb + o_q, queries, (size_t)nq * sizeof(ProjQuery));
    memcpy(hb + o_qd, q_desc32, (size_t)nq * 32);
    // Without a `taken` mask the first scan needs no best-so-far table, and every wavefront reads its one query and descriptor once:
    // the kernel takes them straight from the page-locked mirror (one PCIe read per wavefront, all in flight together) and the block
    // goes down to the device only if a rescan asks for it -- one copy-engine packet less in front of every matcher call.
    bool on_device = false;
    auto to_device = [&]() -> bool {
        if (on_device) return true;
        if (hipMemcpyAsync(base + o_q, hb + o_q, total - o_q, hipMemcpyHostToDevice, s) != hipSuccess) return false;
        on_device = true;
        return true; };
    // (a `taken` mask does not send the block down either: the second pass of the scan reads a candidate's best-so-far entry from the
    // page-locked mirror together with its descriptor -- one more PCIe read in a round of loads that is in flight anyway, one copy-engine
    // packet less in front of the kernel)
    const float inv_w = (float)(64.0 / c->lt.w[0]), inv_h = (float)(48.0 / c->lt.h[0]);
    const float* sxr = use_stereo ? c->d_stereo + (size_t)image * 2 * c->slots_per_image : nullptr;
    ProjGate gate{};
    gate.mode = policy == 1 ? 1 : 0;
    for (int l = 0; l < LPSLAM_HIP_MAX_LEVELS; ++l) { const float sc = l < c->lt.n_levels ? c->lt.scale[l] : 1.0f; gate.inv_sigma_sq[l] = 1.0f / (sc * sc); }
    // the candidate lists and counts are written by the kernel straight into the page-locked mirror, and its last store releases a
    // sequence number there (internal.h, lp_signal_done): no copy-engine packet on the way back
    unsigned* done_counter = lp_done_counter(c, 1);
    if (!done_counter) { release(); set_error("device memory for the completion counters"); return LPSLAM_HIP_ERR_DEVICE; }
    int* done_flag = (int*)(hb + o_ids + 32);
    const int done_seq = lp_next_seq(c->done_seq);
    __atomic_store_n(done_flag, 0, __ATOMIC_RELAXED);
    const int16_t* first_bsf = taken_in ? (const int16_t*)(hb + o_bsf) : (const int16_t*)nullptr;
    // several sessions tracking at once: the first scan joins the other sessions' pending scans in one launch (share.hip)
    int shared = LP_SHARE_DIRECT;
    {
        LpProjReq rq{};
        rq.kp = c->d_kpts + o; rq.desc = c->d_desc + o * 32; rq.stereo_xr = sxr; rq.kp_count = c->d_kp_count + image;
        rq.queries = hb + o_q; rq.q_desc = hb + o_qd; rq.best_so_far = first_bsf;
        rq.out_keys = (unsigned long long*)(hb + o_keys); rq.out_count = (int*)(hb + o_cnt); rq.done_counter = done_counter; rq.done_flag = done_flag;
        rq.nq = nq; rq.grid_x = (nq + 3) / 4; rq.done_seq = done_seq; rq.inv_w = inv_w; rq.inv_h = inv_h;
        rq.gate.mode = gate.mode; for (int l = 0; l < LPSLAM_HIP_MAX_LEVELS; ++l) rq.gate.inv_sigma_sq[l] = gate.inv_sigma_sq[l];
        shared = lp_share_proj(c, rq);
        if (shared < 0) { release(); return -shared; }
    }
    if (shared == LP_SHARE_DIRECT) {
        hipLaunchKernelGGL(k_proj_topk, dim3((nq + 3) / 4), dim3(256), 0, s, c->d_kpts + o, c->d_desc + o * 32, sxr, c->d_kp_count + image,
                           (const ProjQuery*)(hb + o_q), (const uint8_t*)(hb + o_qd), (const int*)nullptr, nq,
                           first_bsf, inv_w, inv_h, gate, (unsigned long long*)(hb + o_keys), (int*)(hb + o_cnt),
                           done_counter, done_flag, done_seq);
        P_HIP(hipGetLastError());
    }
    const unsigned long long* keys = (const unsigned long long*)(hb + o_keys);
    const int* cnt = (const int*)(hb + o_cnt);
    tr_launch = tr_us();
    if (shared == LP_SHARE_DIRECT && !lp_wait_done(done_flag, done_seq, s)) { if (lp_wait_recover(c, 1, s)) release(); set_error("window matcher: the kernel did not complete"); return LPSLAM_HIP_ERR_DEVICE; }
    tr_wait = tr_us();
    int found = 0;
    for (int k = 0; k < nq; ++k) {
        match_idx[k] = -1;
        if (match_dist) match_dist[k] = 256;
        unsigned long long cand[4] = {keys[4 * (size_t)k], keys[4 * (size_t)k + 1], keys[4 * (size_t)k + 2], keys[4 * (size_t)k + 3]};
        auto free_ones = [&](unsigned long long* out) {
            int m = 0;
            for (int r = 0; r < 4; ++r) if (cand[r] != ~0ull && (int)bsf[(size_t)((cand[r] >> 4) & 0xffff)] > (int)(cand[r] >> 32)) out[m++] = cand[r];
            return m; };
        unsigned long long fr[4];
        int m = free_ones(fr);
        if (policy != 1 && m < 2 && cnt[k] > 4) {
            // the short list was eaten by earlier queries: scan again for this query with the current assignment
            ++tr_rescans;
            P_HIP(to_device() ? hipSuccess : hipErrorUnknown);
            P_HIP(hipMemcpyAsync(d_bsf, bsf, nk * sizeof(int16_t), hipMemcpyHostToDevice, s));
            P_HIP(hipMemcpyAsync(d_ids, &k, sizeof(int), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_proj_topk, dim3(1), dim3(256), 0, s, c->d_kpts + o, c->d_desc + o * 32, sxr, c->d_kp_count + image,
                               d_q, d_qd, (const int*)d_ids, 1, (const int16_t*)d_bsf, inv_w, inv_h, gate, d_keys, d_cnt, (unsigned*)nullptr, (int*)nullptr, 0);
            P_HIP(hipGetLastError());
            P_HIP(hipMemcpyAsync(cand, d_keys, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            P_HIP(hipStreamSynchronize(s));
            m = free_ones(fr);
        }
        if (m == 0) continue;
        const int best = (int)(fr[0] >> 32), best_idx = (int)((fr[0] >> 4) & 0xffff), best_lvl = (int)(fr[0] & 15);
        const int second = m > 1 ? (int)(fr[1] >> 32) : 256, second_lvl = m > 1 ? (int)(fr[1] & 15) : -1;
        if (best > hamming_thr) continue;
        if (policy == 0 && best_lvl == second_lvl && (float)best > lowe_ratio * (float)second) continue;
        if (policy == 2 && !((float)best < lowe_ratio * (float)second)) continue;
        if (policy == 2) {
            const int prev = owner[(size_t)best_idx];
            if (prev >= 0) { match_idx[prev] = -1; if (match_dist) match_dist[prev] = 256; --found; }
            owner[(size_t)best_idx] = k;
            bsf[(size_t)best_idx] = (int16_t)best;
        } else if (policy == 0) {
            bsf[(size_t)best_idx] = 0;
        }
        match_idx[k] = best_idx;
        if (match_dist) match_dist[k] = best;
        ++found;
    }
#undef P_HIP
    release();
    if (trace) fprintf(stderr, "window_match: policy %d, %d queries, %d matches, %d rescans; us: setup %.1f, staged + launched %.1f, lists back %.1f, replayed %.1f\n",
                       policy, nq, found, tr_rescans, tr_alloc, tr_launch, tr_wait, tr_us());
    if (n_matches) *n_matches = found;
    return LPSLAM_HIP_OK;
}

int lpslam_hip_match_projection(lpslam_hip_ctx* c, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                                int32_t hamming_thr, float lowe_ratio, const uint8_t* taken_in, int32_t use_stereo,
                                int32_t* match_idx, int32_t* match_dist, int32_t* n_matches)
{
    return window_match(c, image, queries, q_desc32, nq, hamming_thr, lowe_ratio, taken_in, use_stereo, 0, match_idx, match_dist, n_matches);
}

int lpslam_hip_match_fuse(lpslam_hip_ctx* c, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                          int32_t hamming_thr, int32_t use_stereo, int32_t* match_idx, int32_t* match_dist, int32_t* n_matches)
{
    return window_match(c, image, queries, q_desc32, nq, hamming_thr, 1.0f, nullptr, use_stereo, 1, match_idx, match_dist, n_matches);
}

int lpslam_hip_match_area(lpslam_hip_ctx* c, int image, const lpslam_hip_proj_query* queries, const uint8_t* q_desc32, int32_t nq,
                          int32_t hamming_thr, float lowe_ratio, int32_t* match_idx, int32_t* match_dist, int32_t* n_matches)
{
    return window_match(c, image, queries, q_desc32, nq, hamming_thr, lowe_ratio, nullptr, 0, 2, match_idx, match_dist, n_matches);
}

int lpslam_hip_match_orientation_filter(const float* angle_q, const float* angle_t, int32_t* match_idx, int32_t nq, int32_t* n_kept)
{
    if (nq < 0 || (nq > 0 && (!angle_q || !angle_t || !match_idx))) { set_error("bad orientation-filter arguments"); return LPSLAM_HIP_ERR_INVALID; }
    constexpr int HL = 30;                       // match::angle_checker: bin = round(delta / 30), three best bins survive
    int hist[HL + 1] = {0};
    std::vector<int> bin_of((size_t)std::max(nq, 1), -1);
    for (int k = 0; k < nq; ++k) {
        if (match_idx[k] < 0) continue;
        float rot = angle_q[k] - angle_t[match_idx[k]];
        if (rot < 0.0f) rot += 360.0f;
        if (360.0f <= rot) rot -= 360.0f;
        int b = (int)lrintf(rot * (1.0f / HL));
        if (b == HL) b = 0;
        bin_of[k] = b; hist[b]++;
    }
    int i1 = -1, i2 = -1, i3 = -1, m1 = 0, m2 = 0, m3 = 0;
    for (int b = 0; b < HL; ++b) {
        const int sz = hist[b];
        if (sz > m1) { m3 = m2; m2 = m1; m1 = sz; i3 = i2; i2 = i1; i1 = b; }
        else if (sz > m2) { m3 = m2; m2 = sz; i3 = i2; i2 = b; }
        else if (sz > m3) { m3 = sz; i3 = b; }
    }
    if (m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
    else if (m3 < 0.1f * (float)m1) i3 = -1;
    int kept = 0;
    for (int k = 0; k < nq; ++k) {
        if (match_idx[k] < 0) continue;
        if (bin_of[k] == i1 || bin_of[k] == i2 || bin_of[k] == i3) ++kept; else match_idx[k] = -1;
    }
    if (n_kept) *n_kept = kept;
    return LPSLAM_HIP_OK;
}

}  // extern "C"

int lp_launch_proj_batch(hipStream_t s, const LpProjReq* table, int n, int grid_x_max)
{
    if (n < 1) return LPSLAM_HIP_OK;
    hipLaunchKernelGGL(k_proj_topk_req, dim3((unsigned)std::max(grid_x_max, 1), (unsigned)n), dim3(256), 0, s, table);
    LP_HIP(hipGetLastError());
    return LPSLAM_HIP_OK;
}
