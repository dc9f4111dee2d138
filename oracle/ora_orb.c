/*
 * oracle/ora_orb.c -- CPU restatement of the ORB front end.  TEST INFRASTRUCTURE ONLY (see ora.h).
 * PARITY UNPINNED (see ora.h): follows the published upstream algorithms, entered from the
 * reference at src/Trackers/OpenVSLAMStereoTracker.cpp:293-295 / src/Trackers/OpenVSLAMTracker.cpp:120
 * with the parameters of src/Trackers/OpenVSLAMTrackerBase.cpp:193-198.
 *
 * [UPSTREAM] tags name the absent third-party function each routine restates.
 */
#include "ora.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE 19          /* orb_patch_radius_ */
#define CELL 64
#define OVERLAP 6

static inline int iround_half_even(double v) { return (int)nearbyint(v); } /* cvRound (default FE_TONEAREST) */

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] orb_params::calc_scale_factors / calc_inv_scale_factors (float recurrences)
 */
void ora_scale_factors(const ora_orb_params* p, float* scale, float* inv_scale)
{
    scale[0] = 1.0f;
    for (int l = 1; l < p->num_levels; ++l) scale[l] = p->scale_factor * scale[l - 1];
    if (inv_scale)
        for (int l = 0; l < p->num_levels; ++l) inv_scale[l] = 1.0f / scale[l];
}

/* [UPSTREAM] orb_extractor::compute_image_pyramid: size = round(cols / scale), round(rows / scale) */
void ora_pyramid_sizes(int w, int h, const ora_orb_params* p, int* lw, int* lh)
{
    float sf[ORA_MAX_LEVELS];
    ora_scale_factors(p, sf, NULL);
    lw[0] = w; lh[0] = h;
    for (int l = 1; l < p->num_levels; ++l) {
        const double scale = sf[l];
        lw[l] = (int)round(w * 1.0 / scale);
        lh[l] = (int)round(h * 1.0 / scale);
    }
}

/* [UPSTREAM] orb_extractor::initialize: geometric share per level, remainder on the last level */
void ora_keypts_per_level(const ora_orb_params* p, int* quota)
{
    const double f = 1.0 / (double)p->scale_factor;
    double desired = p->max_num_keypts * (1.0 - f) / (1.0 - pow(f, (double)p->num_levels));
    int total = 0;
    for (int l = 0; l < p->num_levels - 1; ++l) {
        quota[l] = (int)round(desired);
        total += quota[l];
        desired *= f;
    }
    int rest = p->max_num_keypts - total;
    quota[p->num_levels - 1] = rest > 0 ? rest : 0;
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] cv::resize(..., INTER_LINEAR) for CV_8UC1: resizeGeneric_ with HResizeLinear<uchar,int,short>
 * (11-bit coefficients) and VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>:
 *   dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2
 */
static void resize_axis_table(int ssize, int dsize, int* ofs, short* coef)
{
    const double scale = 1.0 / ((double)dsize / ssize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        float c0 = 1.f - f, c1 = f;
        coef[2 * d] = (short)iround_half_even(c0 * 2048.f);
        coef[2 * d + 1] = (short)iround_half_even(c1 * 2048.f);
    }
}

void ora_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* alpha = (short*)malloc(sizeof(short) * 2 * dw);
    short* beta = (short*)malloc(sizeof(short) * 2 * dh);
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    resize_axis_table(sw, dw, xofs, alpha);
    resize_axis_table(sh, dh, yofs, beta);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy], sy1 = sy0 + 1;
        if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx0 = xofs[dx], sx1 = sx0 + 1;
            if (sx1 > sw - 1) sx1 = sw - 1;
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            row0[dx] = S0[sx0] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx0] * a0 + S1[sx1] * a1;
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] cv::FAST_t<16> + cornerScore<16> (features2d/src/fast.cpp, fast_score.cpp)
 */
static const int ring16[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold)
{
    enum { K = 8, N = K * 3 + 1 };
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int m = 4; m <= 8; ++m) a = a < d[k + m] ? a : d[k + m];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 5; ++m) b = b > d[k + m] ? b : d[k + m];
        if (b >= b0) continue;
        for (int m = 6; m <= 8; ++m) b = b > d[k + m] ? b : d[k + m];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

int ora_fast9_16(const uint8_t* img, int w, int h, int stride, int thr, int nms,
                 ora_corner* out, int max_out)
{
    enum { K = 8, N = 25 };
    int pixel[N];
    for (int k = 0; k < 16; ++k) pixel[k] = ring16[k][0] + ring16[k][1] * stride;
    for (int k = 16; k < N; ++k) pixel[k] = pixel[k - 16];
    if (thr < 0) thr = 0;
    if (thr > 255) thr = 255;
    if (w < 7 || h < 7) return 0;

    uint8_t* score = (uint8_t*)calloc((size_t)w * h, 1);
    uint8_t* is_corner = (uint8_t*)calloc((size_t)w * h, 1);
    for (int i = 3; i < h - 3; ++i) {
        const uint8_t* row = img + (size_t)i * stride;
        for (int j = 3; j < w - 3; ++j) {
            const uint8_t* ptr = row + j;
            int v = ptr[0];
            int found = 0;
            {   /* cv::FAST_t quick reject: every 9-arc contains one pixel of each opposite pair (k, k+8) */
                const int lo = v - thr, hi = v + thr;
                #define FCLS(k) ((ptr[pixel[k]] < lo ? 1 : 0) | (ptr[pixel[k]] > hi ? 2 : 0))
                int d = FCLS(0) | FCLS(8);
                if (d == 0) continue;
                d &= FCLS(2) | FCLS(10); d &= FCLS(4) | FCLS(12); d &= FCLS(6) | FCLS(14);
                if (d == 0) continue;
                d &= FCLS(1) | FCLS(9); d &= FCLS(3) | FCLS(11); d &= FCLS(5) | FCLS(13); d &= FCLS(7) | FCLS(15);
                if (d == 0) continue;
                #undef FCLS
            }
            {   /* darker arc: 9 contiguous ring pixels with x < v - thr */
                int vt = v - thr, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                }
            }
            if (!found) {   /* brighter arc */
                int vt = v + thr, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { found = 1; break; } }
                    else count = 0;
                }
            }
            if (found) {
                is_corner[(size_t)i * w + j] = 1;
                score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, thr);
            }
        }
    }
    int n = 0;
    for (int i = 3; i < h - 3; ++i) {
        for (int j = 3; j < w - 3; ++j) {
            if (!is_corner[(size_t)i * w + j]) continue;
            const uint8_t* s = score + (size_t)i * w + j;
            int sc = s[0];
            if (!nms || (sc > s[-1] && sc > s[1] && sc > s[-w - 1] && sc > s[-w] && sc > s[-w + 1] &&
                         sc > s[w - 1] && sc > s[w] && sc > s[w + 1])) {
                if (n < max_out) { out[n].x = j; out[n].y = i; out[n].score = sc; }
                ++n;
            }
        }
    }
    free(score); free(is_corner);
    return n;
}

/* [UPSTREAM] orb_extractor::is_in_mask: mask.at<uchar>(y * scale_factor, x * scale_factor) == 0 -- the level coordinate times the
 * level's scale factor (float), truncated to the level-0 pixel (clamped here; upstream trusts the range) */
static int in_mask(const uint8_t* mask, int mw, int mh, int mstride, float y, float x, float scale)
{
    int my = (int)(y * scale), mx = (int)(x * scale);
    if (my < 0) my = 0;
    if (my >= mh) my = mh - 1;
    if (mx < 0) mx = 0;
    if (mx >= mw) mx = mw - 1;
    return mask[(size_t)my * mstride + mx] == 0;
}

/* [UPSTREAM] orb_extractor::compute_fast_keypoints, per-level cell loop (serial order: rows, then cols).  With a mask (level-0
 * size, 0 = masked out): a cell with one of its four corners in the mask is skipped; the threshold fallback looks at what
 * cv::FAST found, then every corner is dropped whose own position is masked. */
int ora_fast_level_masked(const uint8_t* img, int w, int h, int stride, int ini_thr, int min_thr,
                          const uint8_t* mask, int mw, int mh, int mstride, float scale,
                          ora_corner* out, int max_out)
{
    const int min_bx = EDGE, min_by = EDGE;
    const int max_bx = w - EDGE, max_by = h - EDGE;
    if (max_bx <= min_bx || max_by <= min_by) return 0;
    const int width = max_bx - min_bx, height = max_by - min_by;
    const int num_cols = width / CELL + 1;   /* std::ceil(width / cell_size) + 1 with unsigned division */
    const int num_rows = height / CELL + 1;
    ora_corner* tmp = (ora_corner*)malloc(sizeof(ora_corner) * (CELL + OVERLAP) * (CELL + OVERLAP));
    int n = 0;
    for (int i = 0; i < num_rows; ++i) {
        const int min_y = min_by + i * CELL;
        if (max_by - OVERLAP <= min_y) continue;
        int max_y = min_y + CELL + OVERLAP;
        if (max_by < max_y) max_y = max_by;
        for (int j = 0; j < num_cols; ++j) {
            const int min_x = min_bx + j * CELL;
            if (max_bx - OVERLAP <= min_x) continue;
            int max_x = min_x + CELL + OVERLAP;
            if (max_bx < max_x) max_x = max_bx;
            if (mask && (in_mask(mask, mw, mh, mstride, (float)min_y, (float)min_x, scale) || in_mask(mask, mw, mh, mstride, (float)max_y, (float)min_x, scale) ||
                         in_mask(mask, mw, mh, mstride, (float)min_y, (float)max_x, scale) || in_mask(mask, mw, mh, mstride, (float)max_y, (float)max_x, scale))) continue;
            const uint8_t* sub = img + (size_t)min_y * stride + min_x;
            const int cap = (CELL + OVERLAP) * (CELL + OVERLAP);
            int m = ora_fast9_16(sub, max_x - min_x, max_y - min_y, stride, ini_thr, 1, tmp, cap);
            if (m == 0) m = ora_fast9_16(sub, max_x - min_x, max_y - min_y, stride, min_thr, 1, tmp, cap);
            for (int k = 0; k < m; ++k) {
                if (mask && in_mask(mask, mw, mh, mstride, (float)(min_y + tmp[k].y), (float)(min_x + tmp[k].x), scale)) continue;
                if (n < max_out) {
                    out[n].x = tmp[k].x + j * CELL;
                    out[n].y = tmp[k].y + i * CELL;
                    out[n].score = tmp[k].score;
                }
                ++n;
            }
        }
    }
    free(tmp);
    return n;
}

int ora_fast_level(const uint8_t* img, int w, int h, int stride, int ini_thr, int min_thr,
                   ora_corner* out, int max_out)
{
    return ora_fast_level_masked(img, w, h, stride, ini_thr, min_thr, NULL, 0, 0, 0, 1.0f, out, max_out);
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] orb_extractor::distribute_keypoints_via_tree / initialize_nodes / orb_extractor_node::divide_node
 * The upstream keeps nodes in a std::list (children are push_front'ed) and sorts the pool of
 * dividable leaves by (count, node address).  The address tie-break is unspecified behaviour; this
 * restatement makes it deterministic: ties go to the node created LAST (what ascending heap
 * addresses give in practice).  The list order is therefore "descending creation sequence".
 */
typedef struct qnode {
    int bx, by, ex, ey;
    int* idx; int n;      /* candidate indices in insertion order */
    int leaf;             /* exactly one keypoint */
    long seq;             /* creation sequence (larger = closer to list front) */
    struct qnode *prev, *next;
} qnode;

typedef struct { qnode *head, *tail; int size; long next_seq; } qlist;

static qnode* qnode_new(int cap) {
    qnode* n = (qnode*)calloc(1, sizeof(qnode));
    n->idx = (int*)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    return n;
}
static void qlist_push_front(qlist* l, qnode* n) {
    n->seq = l->next_seq++;
    n->prev = NULL; n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n; l->size++;
}
static void qlist_push_back(qlist* l, qnode* n) {
    n->prev = l->tail; n->next = NULL;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n; l->size++;
}
static qnode* qlist_erase(qlist* l, qnode* n) { /* returns next */
    qnode* nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    free(n->idx); free(n);
    return nx;
}

typedef struct { int count; qnode* node; } qpool_entry;

static int qpool_cmp_desc(const void* a, const void* b) {
    const qpool_entry* x = (const qpool_entry*)a; const qpool_entry* y = (const qpool_entry*)b;
    if (x->count != y->count) return y->count - x->count;
    return (y->node->seq > x->node->seq) - (y->node->seq < x->node->seq);
}

/* divide `n` into up to 4 children pushed to the list front; dividable children appended to pool */
static void qdivide(qlist* l, qnode* n, const ora_corner* cand, qpool_entry* pool, int* pool_n)
{
    const int half_x = (int)ceil((n->ex - n->bx) / 2.0);
    const int half_y = (int)ceil((n->ey - n->by) / 2.0);
    qnode* c[4];
    for (int k = 0; k < 4; ++k) c[k] = qnode_new(n->n);
    c[0]->bx = n->bx;          c[0]->by = n->by;          c[0]->ex = n->bx + half_x; c[0]->ey = n->by + half_y;
    c[1]->bx = n->bx + half_x; c[1]->by = n->by;          c[1]->ex = n->ex;          c[1]->ey = n->by + half_y;
    c[2]->bx = n->bx;          c[2]->by = n->by + half_y; c[2]->ex = n->bx + half_x; c[2]->ey = n->ey;
    c[3]->bx = n->bx + half_x; c[3]->by = n->by + half_y; c[3]->ex = n->ex;          c[3]->ey = n->ey;
    for (int i = 0; i < n->n; ++i) {
        const ora_corner* kp = &cand[n->idx[i]];
        int q = 0;
        if (n->bx + half_x <= kp->x) q += 1;
        if (n->by + half_y <= kp->y) q += 2;
        c[q]->idx[c[q]->n++] = n->idx[i];
    }
    for (int k = 0; k < 4; ++k) {
        if (c[k]->n == 0) { free(c[k]->idx); free(c[k]); continue; }
        c[k]->leaf = (c[k]->n == 1);
        qlist_push_front(l, c[k]);
        if (c[k]->n == 1) continue;
        pool[*pool_n].count = c[k]->n; pool[*pool_n].node = c[k]; (*pool_n)++;
    }
}

int ora_distribute(const ora_corner* cand, int n, int min_x, int max_x, int min_y, int max_y,
                   int num_keypts, int32_t* out_idx, int max_out)
{
    if (n <= 0) return 0;
    /* initialize_nodes */
    const double ratio = (double)(max_x - min_x) / (max_y - min_y);
    double delta_x, delta_y; int nxg, nyg;
    if (ratio > 1) { nxg = (int)round(ratio); nyg = 1; delta_x = (double)(max_x - min_x) / nxg; delta_y = max_y - min_y; }
    else { nxg = 1; nyg = (int)round(1 / ratio); delta_x = max_x - min_x; delta_y = (double)(max_y - min_y) / nyg; }
    const int nini = nxg * nyg;
    qlist list = {NULL, NULL, 0, 0};
    qnode** ini = (qnode**)malloc(sizeof(qnode*) * nini);
    for (int i = 0; i < nini; ++i) {
        qnode* nd = qnode_new(n);
        const int ix = i % nxg, iy = i / nxg;
        nd->bx = (int)(delta_x * ix); nd->by = (int)(delta_y * iy);
        nd->ex = (int)(delta_x * (ix + 1)); nd->ey = (int)(delta_y * (iy + 1));
        nd->seq = -(long)(i + 1);       /* push_back: list stays in descending-seq order */
        qlist_push_back(&list, nd);
        ini[i] = nd;
    }
    for (int k = 0; k < n; ++k) {
        unsigned ix = (unsigned)((float)cand[k].x / delta_x);
        unsigned iy = (unsigned)((float)cand[k].y / delta_y);
        unsigned ni = ix + iy * nxg;
        if (ni >= (unsigned)nini) ni = nini - 1;   /* cannot happen for in-range corners */
        ini[ni]->idx[ini[ni]->n++] = k;
    }
    free(ini);
    for (qnode* it = list.head; it;) {
        if (it->n == 0) { it = qlist_erase(&list, it); continue; }
        it->leaf = (it->n == 1);
        it = it->next;
    }

    qpool_entry* pool = (qpool_entry*)malloc(sizeof(qpool_entry) * ((size_t)n + 16));
    qpool_entry* prev_pool = (qpool_entry*)malloc(sizeof(qpool_entry) * ((size_t)n + 16));
    int pool_n = 0;
    int is_filled = 0;
    for (;;) {
        const int prev_size = list.size;
        pool_n = 0;
        for (qnode* it = list.head; it;) {
            if (it->leaf) { it = it->next; continue; }
            qdivide(&list, it, cand, pool, &pool_n);
            it = qlist_erase(&list, it);
        }
        if (num_keypts <= list.size || list.size == prev_size) { is_filled = 1; break; }
        /* every dividable leaf can add up to 3 nodes (ORB-SLAM2: lNodes.size()+nToExpand*3 > N).  The exact
         * constant in the lp-research fork is unverifiable (absent submodule); 3 keeps the count at N..N+3. */
        if (num_keypts < list.size + 3 * pool_n) { is_filled = 0; break; }
    }
    while (!is_filled) {
        const int prev_size = list.size;
        int prev_n = pool_n;
        memcpy(prev_pool, pool, sizeof(qpool_entry) * prev_n);
        pool_n = 0;
        qsort(prev_pool, prev_n, sizeof(qpool_entry), qpool_cmp_desc);
        for (int k = 0; k < prev_n; ++k) {
            qdivide(&list, prev_pool[k].node, cand, pool, &pool_n);
            qlist_erase(&list, prev_pool[k].node);
            if (num_keypts <= list.size) { is_filled = 1; break; }
        }
        if (is_filled || num_keypts <= list.size || list.size == prev_size) { is_filled = 1; break; }
    }
    free(pool); free(prev_pool);

    /* find_keypoints_with_max_response: first maximum in insertion order, nodes in list order */
    int m = 0;
    for (qnode* it = list.head; it; it = it->next) {
        int best = it->idx[0];
        for (int k = 1; k < it->n; ++k)
            if (cand[it->idx[k]].score > cand[best].score) best = it->idx[k];
        if (m < max_out) out_idx[m] = best;
        ++m;
    }
    while (list.head) qlist_erase(&list, list.head);
    return m;
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] cv::fastAtan2 scalar path (core/src/mathfuncs_core.simd.hpp, atan_f32), degrees
 */
float ora_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* [UPSTREAM] orb_extractor::initialize (u_max_) + orb_extractor::ic_angle */
static int g_umax[HALF_PATCH + 2];
static int g_umax_ready = 0;
static void init_umax(void)
{
    if (g_umax_ready) return;
    const int vmax = (int)floor(HALF_PATCH * sqrt(2.0) / 2 + 1);
    const int vmin = (int)ceil(HALF_PATCH * sqrt(2.0) / 2);
    for (int v = 0; v <= vmax; ++v) g_umax[v] = (int)round(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (g_umax[v0] == g_umax[v0 + 1]) ++v0;
        g_umax[v] = v0;
        ++v0;
    }
    g_umax_ready = 1;
}

float ora_ic_angle(const uint8_t* img, int stride, int x, int y)
{
    init_umax();
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0;
        const int d = g_umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * stride];
            const int val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return ora_fast_atan2((float)m_01, (float)m_10);
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) for CV_8U: fixed-point (8.8) separable
 * smoothing, kernel from getGaussianKernelFixedPoint_ED = {18,34,48,56,48,34,18}/256; horizontal pass
 * keeps 8.8, vertical pass accumulates 16.16 and rounds: (sum + 2^15) >> 16.
 */
static const int g_gk[7] = {18, 34, 48, 56, 48, 34, 18};
static inline int reflect101(int i, int n) { if (i < 0) return -i; if (i >= n) return 2 * n - 2 - i; return i; }

void ora_gauss7x7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned s = 0;
            for (int k = -3; k <= 3; ++k) s += g_gk[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = (uint16_t)s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; ++k) s += (uint32_t)g_gk[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
    free(tmp);
}

/* ---------------------------------------------------------------------------------------------
 * Deterministic sin/cos of an angle in degrees.  Upstream uses float cos/sin of angle*pi/180; libm and
 * the GPU math library differ in the last ulp, which would flip rounded sample coordinates, so both the
 * oracle and the HIP kernel evaluate this one fixed sequence of IEEE double operations (no FMA
 * contraction): quadrant reduction + the fdlibm kernel polynomials, result rounded to float.
 */
void ora_sincos_deg(float angle_deg, float* s_out, float* c_out)
{
    const double a = (double)angle_deg * 0.017453292519943295;       /* pi/180 */
    const double qf = floor(a * 0.63661977236758138 + 0.5);          /* 2/pi */
    const int q = (int)qf;
    double r = a - qf * 1.5707963267948966;
    r = r - qf * 6.123233995736766e-17;
    const double z = r * r;
    /* fdlibm __kernel_sin / __kernel_cos coefficients */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
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

/* [UPSTREAM] orb_extractor::compute_orb_descriptor (rotated BRIEF-256, bit k = I(p0) < I(p1), LSB first) */
static const int8_t g_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

void ora_brief256(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t* desc32)
{
    float sin_a, cos_a;
    ora_sincos_deg(angle_deg, &sin_a, &cos_a);
    const uint8_t* center = blurred + (size_t)y * stride + x;
    for (int i = 0; i < 32; ++i) {
        unsigned byte = 0;
        for (int b = 0; b < 8; ++b) {
            const int8_t* p = &g_pattern[(i * 8 + b) * 4];
            const float x0 = p[0], y0 = p[1], x1 = p[2], y1 = p[3];
            const int r0 = iround_half_even(x0 * sin_a + y0 * cos_a), c0 = iround_half_even(x0 * cos_a - y0 * sin_a);
            const int r1 = iround_half_even(x1 * sin_a + y1 * cos_a), c1 = iround_half_even(x1 * cos_a - y1 * sin_a);
            const int t0 = center[r0 * stride + c0], t1 = center[r1 * stride + c1];
            byte |= (unsigned)(t0 < t1) << b;
        }
        desc32[i] = (uint8_t)byte;
    }
}

/* ---------------------------------------------------------------------------------------------
 * [UPSTREAM] orb_extractor::extract; mask (optional): level-0 size, stride `mask_stride`, 0 = masked out
 */
int ora_orb_extract_masked(const uint8_t* img, int w, int h, int stride, const ora_orb_params* p,
                           const uint8_t* mask, int mask_stride,
                           ora_keypoint* kpts, uint8_t* descs, int max_out,
                           uint8_t* pyr_out, int32_t* cand_count);
int ora_orb_extract(const uint8_t* img, int w, int h, int stride, const ora_orb_params* p,
                    ora_keypoint* kpts, uint8_t* descs, int max_out,
                    uint8_t* pyr_out, int32_t* cand_count)
{
    return ora_orb_extract_masked(img, w, h, stride, p, NULL, 0, kpts, descs, max_out, pyr_out, cand_count);
}
int ora_orb_extract_masked(const uint8_t* img, int w, int h, int stride, const ora_orb_params* p,
                           const uint8_t* mask, int mask_stride,
                           ora_keypoint* kpts, uint8_t* descs, int max_out,
                           uint8_t* pyr_out, int32_t* cand_count)
{
    const int L = p->num_levels;
    int lw[ORA_MAX_LEVELS], lh[ORA_MAX_LEVELS], quota[ORA_MAX_LEVELS];
    float sf[ORA_MAX_LEVELS];
    ora_scale_factors(p, sf, NULL);
    ora_pyramid_sizes(w, h, p, lw, lh);
    ora_keypts_per_level(p, quota);

    uint8_t* pyr[ORA_MAX_LEVELS];
    pyr[0] = (uint8_t*)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y) memcpy(pyr[0] + (size_t)y * w, img + (size_t)y * stride, w);
    for (int l = 1; l < L; ++l) {
        pyr[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        ora_resize_linear_u8(pyr[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], pyr[l], lw[l], lh[l], lw[l]);
    }
    if (pyr_out) {
        size_t off = 0;
        for (int l = 0; l < L; ++l) { memcpy(pyr_out + off, pyr[l], (size_t)lw[l] * lh[l]); off += (size_t)lw[l] * lh[l]; }
    }

    int n_out = 0;
    for (int l = 0; l < L; ++l) {
        const int W = lw[l], H = lh[l];
        int cap = W * H / 4 + 16;
        ora_corner* cand = (ora_corner*)malloc(sizeof(ora_corner) * cap);
        int nc = ora_fast_level_masked(pyr[l], W, H, W, p->ini_fast_thr, p->min_fast_thr, mask, w, h, mask_stride, sf[l], cand, cap);
        if (cand_count) cand_count[l] = nc;
        if (nc == 0) { free(cand); continue; }
        int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * ((size_t)nc + 8));
        int ns = ora_distribute(cand, nc, EDGE, W - EDGE, EDGE, H - EDGE, quota[l], sel, nc + 8);
        if (ns > 0) {
            uint8_t* blurred = (uint8_t*)malloc((size_t)W * H);
            ora_gauss7x7_u8(pyr[l], W, H, W, blurred, W);
            const unsigned scaled_patch = (unsigned)(PATCH_SIZE * sf[l]);
            for (int k = 0; k < ns; ++k) {
                const ora_corner* c = &cand[sel[k]];
                const float fx = (float)(c->x + EDGE), fy = (float)(c->y + EDGE);
                const int ix = iround_half_even(fx), iy = iround_half_even(fy);
                const float angle = ora_ic_angle(pyr[l], W, ix, iy);
                if (n_out < max_out) {
                    ora_keypoint* kp = &kpts[n_out];
                    ora_brief256(blurred, W, ix, iy, angle, descs + (size_t)n_out * 32);
                    kp->x = fx * sf[l];            /* correct_keypoint_scale */
                    kp->y = fy * sf[l];
                    kp->size = (float)scaled_patch;
                    kp->angle = angle;
                    kp->response = (float)c->score;
                    kp->octave = l;
                    kp->class_id = -1;
                }
                ++n_out;
            }
            free(blurred);
        }
        free(sel); free(cand);
    }
    for (int l = 0; l < L; ++l) free(pyr[l]);
    return n_out;
}
