/*
 * oracle/ora_match.c -- CPU restatement of descriptor matching.  TEST INFRASTRUCTURE ONLY (see ora.h).
 * PARITY UNPINNED (see ora.h).  [UPSTREAM] OpenVSLAM match::compute_descriptor_distance_32,
 * match::robust (brute force), match::stereo::compute; thresholds HAMMING_DIST_THR_LOW = 50,
 * HAMMING_DIST_THR_HIGH = 100 (stereo accepts best < 75).  Parameters reach the path through
 * src/Trackers/OpenVSLAMTrackerBase.cpp:188-201 (focal_x_baseline, depth_threshold, y margin).
 */
#include "ora.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

int ora_hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32); memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

void ora_match_bf_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    for (int i = 0; i < nq; ++i) {
        int best = 256 + 1, second = 256 + 1, bi = -1;
        for (int j = 0; j < nt; ++j) {
            const int d = ora_hamming256(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) second = d;
        }
        best_idx[i] = bi; best_dist[i] = best; second_dist[i] = second;
    }
}

int ora_match_bf(const uint8_t* q, int nq, const uint8_t* t, int nt, int max_dist, float ratio,
                 int cross_check, int32_t* out_q, int32_t* out_t, int32_t* out_d)
{
    int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(nq > 0 ? nq : 1));
    int32_t* bd = bi + nq; int32_t* sd = bd + nq;
    ora_match_bf_knn2(q, nq, t, nt, bi, bd, sd);
    int32_t* rbi = NULL;
    if (cross_check) {
        rbi = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(nt > 0 ? nt : 1));
        ora_match_bf_knn2(t, nt, q, nq, rbi, rbi + nt, rbi + 2 * nt);
    }
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (bi[i] < 0) continue;
        if (bd[i] > max_dist) continue;
        if (ratio > 0.f && ratio * (float)sd[i] < (float)bd[i]) continue;
        if (cross_check && rbi[bi[i]] != i) continue;
        out_q[n] = i; out_t[n] = bi[i]; out_d[n] = bd[i]; ++n;
    }
    free(bi); free(rbi);
    return n;
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] match::stereo
 */
typedef struct { int corr; int idx; } corr_entry;
static int corr_cmp(const void* a, const void* b) {
    const corr_entry* x = (const corr_entry*)a; const corr_entry* y = (const corr_entry*)b;
    if (x->corr != y->corr) return (x->corr > y->corr) - (x->corr < y->corr);
    return (x->idx > y->idx) - (x->idx < y->idx);
}
static inline int iround_half_even(double v) { return (int)nearbyint(v); }

static int subpixel_disparity(const uint8_t* img_l, const uint8_t* img_r, int w, int h,
                              const ora_keypoint* kl, const ora_keypoint* kr, float scale, float inv_scale,
                              float min_disp, float max_disp,
                              float* best_x_right, float* best_disp, float* best_corr)
{
    enum { WIN = 5, SLIDE = 5 };
    const float x_right = kr->x;
    const int sxl = iround_half_even(kl->x * inv_scale);
    const int syl = iround_half_even(kl->y * inv_scale);
    const int sxr = iround_half_even(x_right * inv_scale);
    (void)h;
    const int ini_x = sxr - SLIDE - WIN;
    const int end_x = sxr + SLIDE + WIN + 1;
    if (ini_x < 0 || w <= end_x) return 0;

    const int cl = img_l[(size_t)syl * w + sxl];
    float corr[2 * SLIDE + 1];
    int best_off = 0;
    *best_corr = (float)UINT_MAX;
    for (int off = -SLIDE; off <= SLIDE; ++off) {
        const int cr = img_r[(size_t)syl * w + sxr + off];
        int sum = 0;     /* L1 norm of (patch_l - centre_l) - (patch_r - centre_r): integer valued, exact in float */
        for (int dy = -WIN; dy <= WIN; ++dy)
            for (int dx = -WIN; dx <= WIN; ++dx) {
                const int a = img_l[(size_t)(syl + dy) * w + sxl + dx] - cl;
                const int b = img_r[(size_t)(syl + dy) * w + sxr + off + dx] - cr;
                sum += abs(a - b);
            }
        const float c = (float)sum;
        if (c < *best_corr) { *best_corr = c; best_off = off; }
        corr[SLIDE + off] = c;
    }
    if (best_off == -SLIDE || best_off == SLIDE) return 0;
    const float c1 = corr[SLIDE + best_off - 1], c2 = corr[SLIDE + best_off], c3 = corr[SLIDE + best_off + 1];
    const float x_delta = (float)((c1 - c3) / (2.0 * (c1 + c3 - 2.0 * c2)));
    if (x_delta < -1.0 || 1.0 < x_delta) return 0;   /* NaN (flat parabola) compares false -> kept, as upstream */
    *best_x_right = scale * ((float)(sxr + best_off) + x_delta);
    *best_disp = kl->x - *best_x_right;
    if (*best_disp < min_disp || max_disp <= *best_disp) return 0;
    if (*best_disp <= 0.0f) { *best_disp = 0.01f; *best_x_right = kl->x - *best_disp; }
    return 1;
}

int ora_match_stereo(const uint8_t* const* pyr_l, const uint8_t* const* pyr_r,
                     const int* lw, const int* lh, const ora_orb_params* p,
                     const ora_keypoint* kl, const uint8_t* dl, int nl,
                     const ora_keypoint* kr, const uint8_t* dr, int nr,
                     float focal_x_baseline, float true_baseline,
                     float* stereo_x_right, float* depths, int32_t* best_right_idx)
{
    float sf[ORA_MAX_LEVELS], isf[ORA_MAX_LEVELS];
    ora_scale_factors(p, sf, isf);
    const int rows = lh[0];
    const float min_disp = 0.0f, max_disp = focal_x_baseline / true_baseline;
    const unsigned hamm_thr = (100 + 50) / 2;

    /* get_right_keypoint_indices_in_each_row(margin = 2.0) */
    int* row_cnt = (int*)calloc(rows + 1, sizeof(int));
    for (int pass = 0; pass < 1; ++pass)
        for (int i = 0; i < nr; ++i) {
            const float r = 2.0f * sf[kr[i].octave];
            const int max_r = (int)ceilf(kr[i].y + r), min_r = (int)floorf(kr[i].y - r);
            for (int y = min_r; y <= max_r; ++y) if (y >= 0 && y < rows) row_cnt[y]++;
        }
    int* row_ofs = (int*)malloc(sizeof(int) * (rows + 1));
    int tot = 0;
    for (int y = 0; y < rows; ++y) { row_ofs[y] = tot; tot += row_cnt[y]; }
    row_ofs[rows] = tot;
    int* row_idx = (int*)malloc(sizeof(int) * (tot > 0 ? tot : 1));
    memset(row_cnt, 0, sizeof(int) * rows);
    for (int i = 0; i < nr; ++i) {
        const float r = 2.0f * sf[kr[i].octave];
        const int max_r = (int)ceilf(kr[i].y + r), min_r = (int)floorf(kr[i].y - r);
        for (int y = min_r; y <= max_r; ++y) if (y >= 0 && y < rows) row_idx[row_ofs[y] + row_cnt[y]++] = i;
    }

    corr_entry* corr = (corr_entry*)malloc(sizeof(corr_entry) * (nl > 0 ? nl : 1));
    int ncorr = 0;
    for (int i = 0; i < nl; ++i) {
        stereo_x_right[i] = -1.0f; depths[i] = -1.0f;
        if (best_right_idx) best_right_idx[i] = -1;
        const int lvl = kl[i].octave;
        const float yl = kl[i].y, xl = kl[i].x;
        const int row = (int)yl;                 /* indices_right_in_row.at(y_left): float -> index truncation */
        if (row < 0 || row >= rows) continue;
        if (row_cnt[row] == 0) continue;
        const float min_xr = xl - max_disp, max_xr = xl - min_disp;
        if (max_xr < 0) continue;
        unsigned best_j = 0, best_d = hamm_thr;
        for (int c = 0; c < row_cnt[row]; ++c) {
            const int j = row_idx[row_ofs[row] + c];
            if (kr[j].octave < lvl - 1 || kr[j].octave > lvl + 1) continue;
            const float xr = kr[j].x;
            if (xr < min_xr || max_xr < xr) continue;
            const unsigned d = (unsigned)ora_hamming256(dl + (size_t)i * 32, dr + (size_t)j * 32);
            if (d < best_d) { best_j = j; best_d = d; }
        }
        if (hamm_thr <= best_d) continue;
        if (best_right_idx) best_right_idx[i] = (int32_t)best_j;
        float bx = -1.0f, bd = -1.0f, bc = 0.f;
        if (!subpixel_disparity(pyr_l[lvl], pyr_r[lvl], lw[lvl], lh[lvl], &kl[i], &kr[best_j],
                                sf[lvl], isf[lvl], min_disp, max_disp, &bx, &bd, &bc)) continue;
        stereo_x_right[i] = bx;
        depths[i] = focal_x_baseline / bd;
        corr[ncorr].corr = (int)bc; corr[ncorr].idx = i; ++ncorr;
    }
    /* reject correlations weaker than 2 x median */
    qsort(corr, ncorr, sizeof(corr_entry), corr_cmp);
    const int median_i = ncorr / 2;
    const float median = ncorr == 0 ? 0.0f : (float)corr[median_i].corr;
    const float thr = (float)(2.0 * median);
    int valid = ncorr;
    for (int k = median_i; k < ncorr; ++k)
        if (thr < (float)corr[k].corr) { stereo_x_right[corr[k].idx] = -1; depths[corr[k].idx] = -1; --valid; }
    free(corr); free(row_cnt); free(row_ofs); free(row_idx);
    return valid;
}

/* ---- projection matching --------------------------------------------------------------------------------------------
 * [UPSTREAM] OpenVSLAM match::projection::match_frame_and_landmarks (local-map tracking) / match_current_and_last_frames
 * (motion-model tracking), data::frame::get_keypoints_in_cell (64 x 48 grid over the image bounds), match::angle_checker.
 * Queries are processed IN ORDER: a keypoint taken by an earlier query is skipped by the later ones before any distance is
 * computed (upstream: "if (frm.landmarks_.at(idx) && ...has_observation()) continue").  Candidates are visited cell column by
 * cell column, inside a column cell row by cell row, inside a cell by keypoint index; the first strictly smaller distance
 * wins.  Thresholds come from the caller (upstream: HAMMING_DIST_THR_HIGH = 100, lowe_ratio per call site). */
static int proj_cell(float v, float vmin, double inv, int n)
{
    int c = (int)floor((v - vmin) * inv);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

int ora_match_projection(const ora_keypoint* kp, const uint8_t* desc, const float* stereo_x_right, int n_kp, int width, int height,
                         const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr, float lowe_ratio,
                         uint8_t* taken /* n_kp, in/out, may be NULL */, int32_t* match_idx, int32_t* match_dist)
{
    const int GC = 64, GR = 48;
    const double inv_w = (double)GC / (double)width, inv_h = (double)GR / (double)height;
    uint8_t* own_taken = NULL;
    if (!taken) { own_taken = (uint8_t*)calloc((size_t)(n_kp > 0 ? n_kp : 1), 1); taken = own_taken; }
    /* grid: keypoint indices per cell (CSR), index order inside a cell */
    int* cell_of = (int*)malloc(sizeof(int) * (size_t)(n_kp > 0 ? n_kp : 1));
    int* cstart = (int*)calloc((size_t)GC * GR + 1, sizeof(int));
    int* clist = (int*)malloc(sizeof(int) * (size_t)(n_kp > 0 ? n_kp : 1));
    for (int i = 0; i < n_kp; ++i) { cell_of[i] = proj_cell(kp[i].x, 0.f, inv_w, GC) * GR + proj_cell(kp[i].y, 0.f, inv_h, GR); cstart[cell_of[i] + 1]++; }
    for (int c = 0; c < GC * GR; ++c) cstart[c + 1] += cstart[c];
    {
        int* fill = (int*)malloc(sizeof(int) * (size_t)GC * GR);
        memcpy(fill, cstart, sizeof(int) * (size_t)GC * GR);
        for (int i = 0; i < n_kp; ++i) clist[fill[cell_of[i]]++] = i;
        free(fill);
    }
    int n_match = 0;
    for (int k = 0; k < nq; ++k) {
        match_idx[k] = -1; match_dist[k] = 256;
        int best = 256, second = 256, best_lvl = -1, second_lvl = -1, best_idx = -1;
        /* cells that can hold a keypoint of the window (a superset is harmless: every keypoint is tested itself) */
        const int cx0 = proj_cell(q[k].x - q[k].radius, 0.f, inv_w, GC), cx1 = proj_cell(q[k].x + q[k].radius, 0.f, inv_w, GC);
        const int cy0 = proj_cell(q[k].y - q[k].radius, 0.f, inv_h, GR), cy1 = proj_cell(q[k].y + q[k].radius, 0.f, inv_h, GR);
        for (int cx = cx0; cx <= cx1; ++cx)
            for (int cy = cy0; cy <= cy1; ++cy)
                for (int s = cstart[cx * GR + cy]; s < cstart[cx * GR + cy + 1]; ++s) {
                    const int i = clist[s];
                    if (!(fabsf(kp[i].x - q[k].x) < q[k].radius && fabsf(kp[i].y - q[k].y) < q[k].radius)) continue;
                    if (q[k].min_level >= 0 && kp[i].octave < q[k].min_level) continue;
                    if (q[k].max_level >= 0 && kp[i].octave > q[k].max_level) continue;
                    if (taken[i]) continue;
                    if (stereo_x_right && 0 < stereo_x_right[i] && q[k].x_right >= 0) {
                        if (q[k].radius < fabsf(q[k].x_right - stereo_x_right[i])) continue;
                    }
                    const int d = ora_hamming256(q_desc + 32 * (size_t)k, desc + 32 * (size_t)i);
                    if (d < best) { second = best; second_lvl = best_lvl; best = d; best_lvl = kp[i].octave; best_idx = i; }
                    else if (d < second) { second = d; second_lvl = kp[i].octave; }
                }
        if (best_idx >= 0 && best <= hamming_thr) {
            if (best_lvl == second_lvl && (float)best > lowe_ratio * (float)second) continue;
            match_idx[k] = best_idx; match_dist[k] = best;
            taken[best_idx] = 1;
            ++n_match;
        }
    }
    free(cell_of); free(cstart); free(clist); free(own_taken);
    return n_match;
}

/* match::angle_checker (ORB-SLAM2 ComputeThreeMaxima): matches whose angle difference falls outside the three most populated
 * bins (the second / third kept only if >= 0.1 x the first) are dropped.  bin = round(delta / 30), delta in [0, 360). */
int ora_match_orientation_filter(const float* angle_q, const float* angle_t, int32_t* match_idx, int nq)
{
    enum { HL = 30 };
    int hist[HL + 1]; memset(hist, 0, sizeof(hist));
    int* bin_of = (int*)malloc(sizeof(int) * (size_t)(nq > 0 ? nq : 1));
    for (int k = 0; k < nq; ++k) {
        bin_of[k] = -1;
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
        const int s = hist[b];
        if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = b; }
        else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = b; }
        else if (s > m3) { m3 = s; i3 = b; }
    }
    if (m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
    else if (m3 < 0.1f * (float)m1) i3 = -1;
    int kept = 0;
    for (int k = 0; k < nq; ++k) {
        if (match_idx[k] < 0) continue;
        if (bin_of[k] == i1 || bin_of[k] == i2 || bin_of[k] == i3) ++kept; else match_idx[k] = -1;
    }
    free(bin_of);
    return kept;
}

/* ---- [UPSTREAM] match::fuse (detect / replace duplication): for every landmark projected into a keyframe the best keypoint
 * inside the window, level in [pred - 1, pred], reprojection error inside the chi-square gate of the keypoint's level
 * (5.99146 with two residuals, 7.81473 with the right-image x as third), smallest Hamming distance <= hamming_thr
 * (HAMMING_DIST_THR_LOW = 50).  No exclusivity: what happens to a keypoint that already has a landmark is the caller's
 * decision (the one with more observations survives).  Candidates in cell-scan order, first strictly smaller wins. */
int ora_match_fuse(const ora_keypoint* kp, const uint8_t* desc, const float* stereo_x_right, int n_kp, int width, int height,
                   const float* inv_level_sigma_sq, const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr,
                   int32_t* match_idx, int32_t* match_dist)
{
    const int GC = 64, GR = 48;
    const double inv_w = (double)GC / (double)width, inv_h = (double)GR / (double)height;
    int n_match = 0;
    for (int k = 0; k < nq; ++k) {
        long best_key = -1; int best = 256, best_idx = -1;
        for (int i = 0; i < n_kp; ++i) {
            if (!(fabsf(kp[i].x - q[k].x) < q[k].radius && fabsf(kp[i].y - q[k].y) < q[k].radius)) continue;
            if (q[k].min_level >= 0 && kp[i].octave < q[k].min_level) continue;
            if (q[k].max_level >= 0 && kp[i].octave > q[k].max_level) continue;
            const float ex = q[k].x - kp[i].x, ey = q[k].y - kp[i].y;
            if (stereo_x_right && 0 <= stereo_x_right[i] && q[k].x_right >= 0) {
                const float er = q[k].x_right - stereo_x_right[i];
                if ((ex * ex + ey * ey + er * er) * inv_level_sigma_sq[kp[i].octave] > 7.81473f) continue;
            } else {
                if ((ex * ex + ey * ey) * inv_level_sigma_sq[kp[i].octave] > 5.99146f) continue;
            }
            const int d = ora_hamming256(q_desc + 32 * (size_t)k, desc + 32 * (size_t)i);
            const long key = (long)(proj_cell(kp[i].x, 0.f, inv_w, GC) * GR + proj_cell(kp[i].y, 0.f, inv_h, GR)) * 65536 + i;
            if (d < best || (d == best && key < best_key)) { best = d; best_idx = i; best_key = key; }
        }
        match_idx[k] = -1; match_dist[k] = 256;
        if (best_idx >= 0 && best <= hamming_thr) { match_idx[k] = best_idx; match_dist[k] = best; ++n_match; }
    }
    return n_match;
}

/* ---- [UPSTREAM] match::area::match_in_consistent_area (monocular initialiser): every level-0 keypoint of frame 1 looks for
 * its match among the level-0 keypoints of frame 2 inside a window around the position it was matched to before; a frame-2
 * keypoint already matched at an equal or smaller distance is not a candidate; best <= hamming_thr and best < ratio * second;
 * a frame-2 keypoint won by a later, better query is taken away from the earlier one.  Queries in order; candidates in
 * cell-scan order.  The angle check is ora_match_orientation_filter. */
int ora_match_area(const ora_keypoint* kp2, const uint8_t* desc2, int n2, int width, int height,
                   const ora_proj_query* q, const uint8_t* q_desc, int nq, int hamming_thr, float lowe_ratio, int32_t* match_idx)
{
    const int GC = 64, GR = 48;
    const double inv_w = (double)GC / (double)width, inv_h = (double)GR / (double)height;
    int* dist2 = (int*)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    int* owner2 = (int*)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    long* key2 = (long*)malloc(sizeof(long) * (size_t)(n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; ++i) { dist2[i] = 256; owner2[i] = -1; key2[i] = (long)(proj_cell(kp2[i].x, 0.f, inv_w, GC) * GR + proj_cell(kp2[i].y, 0.f, inv_h, GR)) * 65536 + i; }
    int n_match = 0;
    for (int k = 0; k < nq; ++k) {
        match_idx[k] = -1;
        int best = 256, second = 256, best_idx = -1; long best_key = -1, second_key = -1;
        for (int i = 0; i < n2; ++i) {
            if (!(fabsf(kp2[i].x - q[k].x) < q[k].radius && fabsf(kp2[i].y - q[k].y) < q[k].radius)) continue;
            if (q[k].min_level >= 0 && kp2[i].octave < q[k].min_level) continue;
            if (q[k].max_level >= 0 && kp2[i].octave > q[k].max_level) continue;
            const int d = ora_hamming256(q_desc + 32 * (size_t)k, desc2 + 32 * (size_t)i);
            if (dist2[i] <= d) continue;
            /* scan order = ascending key: "d < best" on a scan in key order == lexicographic minimum of (d, key) */
            if (d < best || (d == best && key2[i] < best_key)) { second = best; second_key = best_key; best = d; best_key = key2[i]; best_idx = i; }
            else if (d < second || (d == second && key2[i] < second_key)) { second = d; second_key = key2[i]; }
        }
        if (best_idx >= 0 && best <= hamming_thr && (float)best < lowe_ratio * (float)second) {
            const int prev = owner2[best_idx];
            if (prev >= 0) { match_idx[prev] = -1; --n_match; }
            match_idx[k] = best_idx; owner2[best_idx] = k; dist2[best_idx] = best;
            ++n_match;
        }
    }
    free(dist2); free(owner2); free(key2);
    return n_match;
}
