/* mslam_oracle.c — CPU ORACLE (test infrastructure only; see mslam_oracle.h).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (oracle/Makefile).  All float
 * arithmetic is plain IEEE binary32 with no contraction, which is how the reference is built
 * (no -march/-mfma anywhere in its CMake files; USE_SSE_ORB / USE_OPENMP undefined).
 *
 * "REF" = /root/reference/src/lib/modular_slam/ ; "DCF" = REF/distributed_cv_feature.cpp ;
 * "ORBF" = REF/orb_feature.cpp ; "PATCH" = conan_recipes/dbow3/dbow3.patch.
 */
#include "mslam_oracle.h"
#include "../include/mslam_orb_pattern.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int8_t k_pattern[1024] = MSLAM_ORB_PATTERN_INIT;

/* ---- OpenCV scalar helpers (cvRound/cvFloor/cvCeil, A.5) -------------------------------- */
static inline int cv_round_f(float v) { return (int)lrintf(v); } /* round-half-even (default FE mode) */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v)
{
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil_d(double v)
{
    int i = (int)v;
    return i + (i < v);
}
static inline int16_t sat_short_from_float(float v)
{
    int iv = cv_round_f(v);
    return (int16_t)(iv < SHRT_MIN ? SHRT_MIN : iv > SHRT_MAX ? SHRT_MAX : iv);
}

void mso_default_params(mso_orb_params* p)
{
    /* DCF:1184-1186 : orb_params("orb", 1.2f, 8, 20, 7), OrbExtractorPimpl(params, 1000) */
    p->n_levels = 8;
    p->scale_factor = 1.2f;
    p->ini_fast_thr = 20;
    p->min_fast_thr = 7;
    p->min_size = 1000;
}

/* ---- REF/frame.cpp:6-27 ------------------------------------------------------------------ */
void mso_gray(const uint8_t* bgr, size_t n_px, uint8_t* gray)
{
    for(size_t i = 0; i < n_px; ++i)
    {
        const uint8_t c0 = bgr[3 * i], c1 = bgr[3 * i + 1], c2 = bgr[3 * i + 2];
        /* frame.cpp:18-19: 0.299f*r + 0.587f*g + 0.114f*b evaluated left to right in float */
        float v = 0.299f * (float)c0 + 0.587f * (float)c1;
        v = v + 0.114f * (float)c2;
        if(255.0f < v) /* std::min(255.f, v) */
            v = 255.0f;
        gray[i] = (uint8_t)v; /* static_cast<uint8_t> truncates */
    }
}

/* ---- DCF:411-420, :836-837 --------------------------------------------------------------- */
void mso_level_geometry(int W, int H, const mso_orb_params* p, int* w, int* h, float* scale)
{
    scale[0] = 1.0f;
    for(int l = 1; l < p->n_levels; ++l)
        scale[l] = p->scale_factor * scale[l - 1]; /* DCF:416 float chain */
    w[0] = W;
    h[0] = H;
    for(int l = 1; l < p->n_levels; ++l)
    {
        const double s = scale[l]; /* DCF:836 */
        w[l] = (int)round(W * 1.0 / s); /* DCF:837 std::round = half away from zero */
        h[l] = (int)round(H * 1.0 / s);
    }
}

/* ---- cv::resize INTER_LINEAR, CV_8UC1 (OpenCV 4.x imgproc/resize.cpp) --------------------- */
void mso_resize_tables(int ssize, int dsize, int32_t* ofs, int16_t* coef)
{
    /* cv::resize(): inv_scale = (double)dsize/ssize ; hal::resize(): scale = 1./inv_scale */
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for(int d = 0; d < dsize; ++d)
    {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor_f(f);
        f -= s;
        if(s < 0)
        {
            f = 0;
            s = 0;
        }
        if(s >= ssize - 1)
        {
            f = 0;
            s = ssize - 1;
        }
        ofs[d] = s;
        coef[2 * d] = sat_short_from_float((1.f - f) * 2048); /* INTER_RESIZE_COEF_SCALE */
        coef[2 * d + 1] = sat_short_from_float(f * 2048);
    }
}

void mso_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    int32_t* xofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw);
    int16_t* alpha = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)dw);
    int32_t* yofs = (int32_t*)malloc(sizeof(int32_t) * (size_t)dh);
    int16_t* beta = (int16_t*)malloc(sizeof(int16_t) * 2 * (size_t)dh);
    int32_t* row0 = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw);
    int32_t* row1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)dw);
    mso_resize_tables(sw, dw, xofs, alpha);

    /* vertical table: same fractional rule but NO clamp of the coefficient; the row index is
     * clipped in resizeGeneric_Invoker (clip(sy, 0, ssize.height)). */
    {
        const double inv_scale = (double)dh / sh;
        const double scale = 1. / inv_scale;
        for(int d = 0; d < dh; ++d)
        {
            float f = (float)((d + 0.5) * scale - 0.5);
            int s = cv_floor_f(f);
            f -= s;
            yofs[d] = s;
            beta[2 * d] = sat_short_from_float((1.f - f) * 2048);
            beta[2 * d + 1] = sat_short_from_float(f * 2048);
        }
    }

    for(int dy = 0; dy < dh; ++dy)
    {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sw;
        const uint8_t* S1 = src + (size_t)sy1 * sw;
        for(int dx = 0; dx < dw; ++dx)
        {
            /* HResizeLinear<uchar,int,short,2048>: for sx at the right edge a1 == 0 and the
             * tail loop uses S[sx]*ONE — identical value, and S[sx+1] is never read. */
            const int sx = xofs[dx];
            const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        /* VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>: */
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dw;
        for(int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(alpha);
    free(yofs);
    free(beta);
    free(row0);
    free(row1);
}

/* ---- cv::FAST TYPE_9_16 with NMS (OpenCV 4.x features2d/fast.cpp, fast_score.cpp) --------- */
static const int k_circle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold)
{
    /* cornerScore<16> scalar branch */
    int d[25];
    const int v = ptr[0];
    for(int k = 0; k < 25; ++k)
        d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for(int k = 0; k < 16; k += 2)
    {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if(a <= a0)
            continue;
        for(int j = 4; j <= 8; ++j)
            a = a < d[k + j] ? a : d[k + j];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for(int k = 0; k < 16; k += 2)
    {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for(int j = 3; j <= 5; ++j)
            b = b > d[k + j] ? b : d[k + j];
        if(b >= b0)
            continue;
        for(int j = 6; j <= 8; ++j)
            b = b > d[k + j] ? b : d[k + j];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

int mso_fast(const uint8_t* img, int step, int cols, int rows, int threshold, mso_cand* out, int cap)
{
    int pixel[25];
    for(int k = 0; k < 16; ++k)
        pixel[k] = k_circle[k][0] + k_circle[k][1] * step;
    for(int k = 16; k < 25; ++k)
        pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    if(cols <= 0 || rows <= 0)
        return 0;

    /* full score map; untested pixels stay 0 exactly like FAST_t's rolling zeroed buffers */
    uint8_t* score = (uint8_t*)calloc((size_t)cols * rows, 1);
    uint8_t* is_corner = (uint8_t*)calloc((size_t)cols * rows, 1);
    for(int i = 3; i < rows - 3; ++i)
    {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        for(int j = 3; j < cols - 3; ++j, ++ptr)
        {
            const int v = ptr[0];
            int found = 0;
            { /* darker run: x < v - threshold, more than K=8 contiguous out of N=25 */
                const int vt = v - threshold;
                int count = 0;
                for(int k = 0; k < 25; ++k)
                {
                    if(ptr[pixel[k]] < vt)
                    {
                        if(++count > 8)
                        {
                            found = 1;
                            break;
                        }
                    }
                    else
                        count = 0;
                }
            }
            if(!found)
            { /* brighter run */
                const int vt = v + threshold;
                int count = 0;
                for(int k = 0; k < 25; ++k)
                {
                    if(ptr[pixel[k]] > vt)
                    {
                        if(++count > 8)
                        {
                            found = 1;
                            break;
                        }
                    }
                    else
                        count = 0;
                }
            }
            if(found)
            {
                is_corner[(size_t)i * cols + j] = 1;
                score[(size_t)i * cols + j] = (uint8_t)corner_score16(ptr, pixel, threshold);
            }
        }
    }
    int n = 0;
    for(int i = 3; i < rows - 3; ++i)
        for(int j = 3; j < cols - 3; ++j)
        {
            if(!is_corner[(size_t)i * cols + j])
                continue;
            const uint8_t* s = score + (size_t)i * cols + j;
            const int sc = s[0];
            if(sc > s[1] && sc > s[-1] && sc > s[-cols - 1] && sc > s[-cols] && sc > s[-cols + 1] &&
               sc > s[cols - 1] && sc > s[cols] && sc > s[cols + 1])
            {
                if(n < cap)
                {
                    out[n].x = (float)j;
                    out[n].y = (float)i;
                    out[n].response = (float)sc;
                }
                ++n;
            }
        }
    free(score);
    free(is_corner);
    return n;
}

/* ---- DCF:858-952 -------------------------------------------------------------------------- */
int mso_fast_level(const uint8_t* img, int cols, int rows, const mso_orb_params* p, mso_cand* out, int cap)
{
    const unsigned overlap = 6, cell_size = 64;                    /* DCF:852-853 */
    const unsigned min_border_x = MSO_PATCH_RADIUS, min_border_y = MSO_PATCH_RADIUS;
    if(cols <= 2 * MSO_PATCH_RADIUS + (int)overlap || rows <= 2 * MSO_PATCH_RADIUS + (int)overlap)
        return 0; /* the reference's unsigned arithmetic is undefined for such tiny levels */
    const unsigned max_border_x = (unsigned)cols - MSO_PATCH_RADIUS; /* DCF:864-865 */
    const unsigned max_border_y = (unsigned)rows - MSO_PATCH_RADIUS;
    const unsigned width = max_border_x - min_border_x, height = max_border_y - min_border_y;
    const unsigned num_cols = width / cell_size + 1; /* DCF:870-871: ceil of an integer quotient */
    const unsigned num_rows = height / cell_size + 1;
    int n = 0;
    mso_cand* tmp = (mso_cand*)malloc(sizeof(mso_cand) * 70 * 70);
    for(unsigned i = 0; i < num_rows; ++i)
    {
        const unsigned min_y = min_border_y + i * cell_size;
        if(max_border_y - overlap <= min_y)
            continue;
        unsigned max_y = min_y + cell_size + overlap;
        if(max_border_y < max_y)
            max_y = max_border_y;
        for(unsigned j = 0; j < num_cols; ++j)
        {
            const unsigned min_x = min_border_x + j * cell_size;
            if(max_border_x - overlap <= min_x)
                continue;
            unsigned max_x = min_x + cell_size + overlap;
            if(max_border_x < max_x)
                max_x = max_border_x;
            const uint8_t* sub = img + (size_t)min_y * cols + min_x;
            int m = mso_fast(sub, cols, (int)(max_x - min_x), (int)(max_y - min_y), p->ini_fast_thr, tmp, 70 * 70);
            if(m == 0) /* DCF:922-926 */
                m = mso_fast(sub, cols, (int)(max_x - min_x), (int)(max_y - min_y), p->min_fast_thr, tmp, 70 * 70);
            for(int k = 0; k < m; ++k)
            {
                if(n < cap)
                {
                    out[n].x = tmp[k].x + (float)(j * cell_size); /* DCF:940-941 */
                    out[n].y = tmp[k].y + (float)(i * cell_size);
                    out[n].response = tmp[k].response;
                }
                ++n;
            }
        }
    }
    free(tmp);
    return n;
}

/* ---- quadtree: DCF:981-1155, DCF:306-355 -------------------------------------------------- */
typedef struct
{
    int bx, by, ex, ey;
    int* kp; /* indices into the input array, in insertion order */
    int n;
    int prev, next; /* std::list links (pool indices) */
} qnode;

typedef struct
{
    qnode* pool;
    int used, cap;
    int head, tail, size;
} qlist;

static int ql_new(qlist* L)
{
    if(L->used == L->cap)
    {
        L->cap = L->cap ? L->cap * 2 : 64;
        L->pool = (qnode*)realloc(L->pool, sizeof(qnode) * (size_t)L->cap);
    }
    qnode* q = &L->pool[L->used];
    memset(q, 0, sizeof(*q));
    q->prev = q->next = -1;
    return L->used++;
}
static void ql_push_back(qlist* L, int id)
{
    L->pool[id].prev = L->tail;
    L->pool[id].next = -1;
    if(L->tail >= 0)
        L->pool[L->tail].next = id;
    else
        L->head = id;
    L->tail = id;
    L->size++;
}
static void ql_push_front(qlist* L, int id)
{
    L->pool[id].next = L->head;
    L->pool[id].prev = -1;
    if(L->head >= 0)
        L->pool[L->head].prev = id;
    else
        L->tail = id;
    L->head = id;
    L->size++;
}
static int ql_erase(qlist* L, int id) /* returns the following element */
{
    const int p = L->pool[id].prev, nx = L->pool[id].next;
    if(p >= 0)
        L->pool[p].next = nx;
    else
        L->head = nx;
    if(nx >= 0)
        L->pool[nx].prev = p;
    else
        L->tail = p;
    free(L->pool[id].kp);
    L->pool[id].kp = NULL;
    L->size--;
    return nx;
}

int mso_quadtree(const mso_cand* in, int n, int min_x, int max_x, int min_y, int max_y, float scale_factor,
                 unsigned min_size, mso_cand* out, int cap)
{
    qlist L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;

    /* initialize_nodes, DCF:1025-1105 */
    const double ratio = (double)(max_x - min_x) / (max_y - min_y);
    double delta_x, delta_y;
    unsigned num_x_grid, num_y_grid;
    if(ratio > 1)
    {
        num_x_grid = (unsigned)round(ratio);
        num_y_grid = 1;
        delta_x = (double)(max_x - min_x) / num_x_grid;
        delta_y = max_y - min_y;
    }
    else
    {
        num_x_grid = 1;
        num_y_grid = (unsigned)round(1 / ratio);
        delta_x = max_x - min_y; /* sic, DCF:1050 */
        delta_y = (double)(max_y - min_y) / num_y_grid;
    }
    const unsigned num_initial = num_x_grid * num_y_grid;
    int* initial = (int*)malloc(sizeof(int) * num_initial);
    for(unsigned i = 0; i < num_initial; ++i)
    {
        const unsigned ix = i % num_x_grid, iy = i / num_x_grid;
        const int id = ql_new(&L);
        qnode* q = &L.pool[id];
        q->bx = (int)(delta_x * ix);
        q->by = (int)(delta_y * iy);
        q->ex = (int)(delta_x * (ix + 1));
        q->ey = (int)(delta_y * (iy + 1));
        q->kp = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        ql_push_back(&L, id);
        initial[i] = id;
    }
    for(int k = 0; k < n; ++k)
    {
        const unsigned ix = (unsigned)(in[k].x / delta_x); /* float / double */
        const unsigned iy = (unsigned)(in[k].y / delta_y);
        const unsigned node_idx = ix + iy * num_x_grid;
        if(node_idx >= num_initial)
            continue; /* .at() would throw in the reference; unreachable for in-range keypoints */
        qnode* q = &L.pool[initial[node_idx]];
        q->kp[q->n++] = k;
    }
    for(int it = L.head; it >= 0;)
    {
        if(L.pool[it].n == 0)
            it = ql_erase(&L, it);
        else
            it = L.pool[it].next;
    }
    free(initial);

    /* distribute_keypoints_via_tree main loop, DCF:992-1020 */
    for(;;)
    {
        const int prev_size = L.size;
        int it = L.head;
        while(it >= 0)
        {
            qnode q = L.pool[it];
            const unsigned area = (unsigned)((q.ex - q.bx) * (q.ey - q.by)); /* DCF:294 */
            if(q.n == 1 || (float)area * scale_factor * scale_factor <= (float)min_size) /* DCF:1002 */
            {
                it = q.next;
                continue;
            }
            /* divide_node, DCF:306-355 */
            const unsigned half_x = (unsigned)cv_ceil_d((q.ex - q.bx) / 2.0);
            const unsigned half_y = (unsigned)cv_ceil_d((q.ey - q.by) / 2.0);
            int cb[4][4];
            const int cx = q.bx + (int)half_x, cy = q.by + (int)half_y;
            cb[0][0] = q.bx, cb[0][1] = q.by, cb[0][2] = cx, cb[0][3] = cy;     /* begin .. center   */
            cb[1][0] = cx, cb[1][1] = q.by, cb[1][2] = q.ex, cb[1][3] = cy;     /* top .. right      */
            cb[2][0] = q.bx, cb[2][1] = cy, cb[2][2] = cx, cb[2][3] = q.ey;     /* left .. bottom    */
            cb[3][0] = cx, cb[3][1] = cy, cb[3][2] = q.ex, cb[3][3] = q.ey;     /* center .. end     */
            int* ckp[4];
            int cn[4] = {0, 0, 0, 0};
            for(int c = 0; c < 4; ++c)
                ckp[c] = (int*)malloc(sizeof(int) * (size_t)q.n);
            for(int k = 0; k < q.n; ++k)
            {
                const mso_cand* kp = &in[q.kp[k]];
                unsigned idx = 0;
                if((float)((unsigned)q.bx + half_x) <= kp->x) /* unsigned -> float, DCF:343 */
                    idx += 1;
                if((float)((unsigned)q.by + half_y) <= kp->y)
                    idx += 2;
                ckp[idx][cn[idx]++] = q.kp[k];
            }
            /* assign_child_nodes, DCF:1107-1126: non-empty children are push_front'ed in order */
            for(int c = 0; c < 4; ++c)
            {
                if(cn[c] == 0)
                {
                    free(ckp[c]);
                    continue;
                }
                const int id = ql_new(&L);
                qnode* ch = &L.pool[id];
                ch->bx = cb[c][0], ch->by = cb[c][1], ch->ex = cb[c][2], ch->ey = cb[c][3];
                ch->kp = ckp[c];
                ch->n = cn[c];
                ql_push_front(&L, id);
            }
            it = ql_erase(&L, it); /* DCF:1012 */
        }
        if(L.size == prev_size) /* DCF:1016-1019 */
            break;
    }

    /* find_keypoints_with_max_response, DCF:1128-1155: first maximum wins */
    int m = 0;
    for(int it = L.head; it >= 0; it = L.pool[it].next)
    {
        const qnode* q = &L.pool[it];
        int best = q->kp[0];
        double max_response = in[best].response;
        for(int k = 1; k < q->n; ++k)
            if(in[q->kp[k]].response > max_response)
            {
                best = q->kp[k];
                max_response = in[best].response;
            }
        if(m < cap)
            out[m] = in[best];
        ++m;
    }
    for(int i = 0; i < L.used; ++i)
        free(L.pool[i].kp);
    free(L.pool);
    return m;
}

/* ---- cv::fastAtan2 scalar (core/mathfuncs_core.simd.hpp atan_f32) ------------------------- */
float mso_fast_atan2(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if(ax >= ay)
    {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else
    {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if(x < 0)
        a = 180.f - a;
    if(y < 0)
        a = 360.f - a;
    return a;
}

/* ---- util::cos / util::sin, DCF:456-503 --------------------------------------------------- */
static const float K_PI = 3.14159265358979f;
static inline float poly_cos(float v)
{
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}
float mso_util_cos(float v)
{
    const float PI_2 = K_PI / 2.0f, TWO_PI = 2.0f * K_PI, INV_TWO_PI = 1.0f / TWO_PI, THREE_PI_2 = 3.0f * PI_2;
    v = v - (float)cv_floor_f(v * INV_TWO_PI) * TWO_PI;
    v = (0.0f < v) ? v : -v;
    if(v < PI_2)
        return poly_cos(v);
    else if(v < K_PI)
        return -poly_cos(K_PI - v);
    else if(v < THREE_PI_2)
        return -poly_cos(v - K_PI);
    else
        return poly_cos(TWO_PI - v);
}
float mso_util_sin(float v)
{
    const float PI_2 = K_PI / 2.0f;
    return mso_util_cos(PI_2 - v);
}

/* ---- DCF:522-541 -------------------------------------------------------------------------- */
void mso_umax(int* umax)
{
    const int hp = 15; /* fast_half_patch_size_ = 31/2 */
    const unsigned vmax = (unsigned)floor(hp * sqrt(2.0) / 2 + 1);
    const unsigned vmin = (unsigned)ceil(hp * sqrt(2.0) / 2);
    for(unsigned v = 0; v <= vmax; ++v)
        umax[v] = (int)round(sqrt((double)(hp * hp) - (double)(v * v)));
    for(unsigned v = hp, v0 = 0; vmin <= v; --v)
    {
        while(umax[v0] == umax[v0 + 1])
            ++v0;
        umax[v] = (int)v0;
        ++v0;
    }
}

/* ---- DCF:543-570 -------------------------------------------------------------------------- */
float mso_ic_angle(const uint8_t* img, int step, int x, int y)
{
    int umax[16];
    mso_umax(umax);
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * step + x;
    for(int u = -15; u <= 15; ++u)
        m_10 += u * center[u];
    for(int v = 1; v <= 15; ++v)
    {
        int v_sum = 0;
        const int d = umax[v];
        for(int u = -d; u <= d; ++u)
        {
            const int val_plus = center[u + v * step];
            const int val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return mso_fast_atan2((float)m_01, (float)m_10);
}

/* ---- cv::GaussianBlur 7x7 sigma=2, CV_8U bit-exact path (imgproc/smooth.dispatch.cpp) ----- */
void mso_gaussian_kernel_fixed(int taps[7])
{
    /* getGaussianKernelBitExact(n=7, sigma=2) then getGaussianKernelFixedPoint_ED(fractionBits=8).
     * OpenCV evaluates this in softdouble; IEEE double gives the same taps (every rounding below is
     * > 0.03 away from a tie).  tests/test_oracle_primitives.py pins the result {18,34,48,56,48,34,18}. */
    const int n = 7, n2 = 3;
    const double sigma = 2.0, scale2x = -0.125 / (sigma * sigma);
    double values[3], sum = 0;
    for(int i = 0, x = 1 - n; i < n2; ++i, x += 2)
    {
        values[i] = exp((double)(x * x) * scale2x);
        sum += values[i];
    }
    sum *= 2;
    sum += 1;
    const double mul1 = 1.0 / sum;
    double kern[7];
    for(int i = 0; i < n2; ++i)
        kern[i] = kern[n - 1 - i] = values[i] * mul1;
    kern[n2] = mul1;
    double err = 0;
    long total = 0;
    for(int i = 0; i < n2; ++i)
    {
        const double adj = kern[i] * 256.0 + err;
        const long v0 = lrint(adj);
        err = adj - (double)v0;
        taps[i] = taps[n - 1 - i] = (int)v0;
        total += 2 * v0;
    }
    taps[n2] = (int)(256 - total);
}

static inline int reflect101(int p, int len)
{
    if(len == 1)
        return 0;
    while(p < 0 || p >= len)
    {
        if(p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

void mso_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst)
{
    int taps[7];
    mso_gaussian_kernel_fixed(taps);
    /* horizontal pass: u8 x 8.8 -> 8.8 (ufixedpoint16, exact because the taps sum to 256) */
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
    for(int y = 0; y < h; ++y)
        for(int x = 0; x < w; ++x)
        {
            uint32_t acc = 0;
            for(int k = -3; k <= 3; ++k)
                acc += (uint32_t)taps[k + 3] * src[(size_t)y * w + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = (uint16_t)acc;
        }
    /* vertical pass: 8.8 x 8.8 -> 16.16 (ufixedpoint32), rounded to u8 with (v + 2^15) >> 16 */
    for(int y = 0; y < h; ++y)
        for(int x = 0; x < w; ++x)
        {
            uint32_t acc = 0;
            for(int k = -3; k <= 3; ++k)
                acc += (uint32_t)taps[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            const uint32_t r = (acc + 32768u) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(r > 255 ? 255 : r);
        }
    free(tmp);
}

/* ---- DCF:572-629 (scalar branch) ---------------------------------------------------------- */
void mso_orb_descriptor(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc)
{
    const float angle = (float)(angle_deg * M_PI / 180.0); /* DCF:574: double product, then float */
    const float cos_angle = mso_util_cos(angle);
    const float sin_angle = mso_util_sin(angle);
    const uint8_t* center = blurred + (size_t)y * step + x;
#define MSO_VALUE(s)                                                                                                   \
    (center[cv_round_f((float)k_pattern[(s)] * sin_angle + (float)k_pattern[(s) + 1] * cos_angle) * step +            \
            cv_round_f((float)k_pattern[(s)] * cos_angle - (float)k_pattern[(s) + 1] * sin_angle)])
    for(int i = 0; i < 32; ++i)
    {
        int val = 0;
        for(int b = 0; b < 8; ++b)
        {
            const int s = i * 32 + b * 4;
            val |= (MSO_VALUE(s) < MSO_VALUE(s + 2)) << b; /* DCF:607, :616-623 */
        }
        desc[i] = (uint8_t)val;
    }
#undef MSO_VALUE
}

/* ---- DCF:719-809 + :1190-1222 -------------------------------------------------------------- */
int mso_detect(const uint8_t* bgr, int W, int H, const mso_orb_params* p, int max_out, float* xy, uint8_t* desc,
               int32_t* octave, float* angle, float* response, int* n_out)
{
    *n_out = 0;
    if(W <= 0 || H <= 0 || p->n_levels < 1 || p->n_levels > MSO_MAX_LEVELS)
        return 0; /* DCF:724-727: empty image => empty outputs */
    int w[MSO_MAX_LEVELS], h[MSO_MAX_LEVELS];
    float sf[MSO_MAX_LEVELS];
    mso_level_geometry(W, H, p, w, h, sf);

    uint8_t* pyr[MSO_MAX_LEVELS];
    pyr[0] = (uint8_t*)malloc((size_t)W * H);
    mso_gray(bgr, (size_t)W * H, pyr[0]);
    for(int l = 1; l < p->n_levels; ++l) /* DCF:830-841: chained */
    {
        pyr[l] = (uint8_t*)malloc((size_t)w[l] * h[l]);
        mso_resize_linear(pyr[l - 1], w[l - 1], h[l - 1], pyr[l], w[l], h[l]);
    }

    int total = 0, rc = 0;
    const int cand_cap = W * H / 4 + 16;
    mso_cand* cand = (mso_cand*)malloc(sizeof(mso_cand) * (size_t)cand_cap);
    mso_cand* sel = (mso_cand*)malloc(sizeof(mso_cand) * (size_t)cand_cap);
    uint8_t* blurred = (uint8_t*)malloc((size_t)W * H);
    for(int l = 0; l < p->n_levels; ++l)
    {
        const int nc = mso_fast_level(pyr[l], w[l], h[l], p, cand, cand_cap);
        if(nc == 0)
            continue; /* an empty list yields no nodes, hence no keypoints */
        const int ns = mso_quadtree(cand, nc, MSO_PATCH_RADIUS, w[l] - MSO_PATCH_RADIUS, MSO_PATCH_RADIUS,
                                    h[l] - MSO_PATCH_RADIUS, sf[l], p->min_size, sel, cand_cap);
        if(ns == 0)
            continue; /* DCF:792-795 */
        mso_gaussian_blur7(pyr[l], w[l], h[l], blurred); /* DCF:797-798 */
        for(int k = 0; k < ns; ++k)
        {
            const float px = sel[k].x + (float)MSO_PATCH_RADIUS; /* DCF:966-967 */
            const float py = sel[k].y + (float)MSO_PATCH_RADIUS;
            const int ix = cv_round_f(px), iy = cv_round_f(py);
            const float ang = mso_ic_angle(pyr[l], w[l], ix, iy); /* DCF:977: unblurred level */
            if(total < max_out)
            {
                mso_orb_descriptor(blurred, w[l], ix, iy, ang, desc + (size_t)total * 32);
                /* correct_keypoint_scale, DCF:1166-1179 */
                xy[2 * total] = l == 0 ? px : px * sf[l];
                xy[2 * total + 1] = l == 0 ? py : py * sf[l];
                octave[total] = l;
                angle[total] = ang;
                response[total] = sel[k].response;
            }
            else
                rc = -1;
            ++total;
        }
    }
    *n_out = total;
    for(int l = 0; l < p->n_levels; ++l)
        free(pyr[l]);
    free(cand);
    free(sel);
    free(blurred);
    return rc;
}

/* ---- BFMatcher(NORM_HAMMING)::knnMatch k=2 (core/batch_distance.cpp) ---------------------- */
static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

void mso_match_knn2_raw(const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to, int32_t* idx0,
                        int32_t* idx1, int32_t* d0, int32_t* d1)
{
    for(int q = 0; q < n_to; ++q)
    {
        int dist[2] = {INT_MAX, INT_MAX}, nidx[2] = {-1, -1};
        for(int j = 0; j < n_from; ++j)
        {
            const int d = hamming256(to_desc + (size_t)q * 32, from_desc + (size_t)j * 32);
            if(d < dist[1])
            {
                int k;
                for(k = 0; k >= 0 && dist[k] > d; --k)
                {
                    nidx[k + 1] = nidx[k];
                    dist[k + 1] = dist[k];
                }
                nidx[k + 1] = j;
                dist[k + 1] = d;
            }
        }
        idx0[q] = nidx[0], idx1[q] = nidx[1], d0[q] = dist[0], d1[q] = dist[1];
    }
}

/* ---- ORBF:84-117 -------------------------------------------------------------------------- */
int mso_match(const uint8_t* from_desc, int n_from, const uint8_t* to_desc, int n_to, double ratio,
              int32_t* from_idx, int32_t* to_idx)
{
    if(n_from < 2 || n_to < 1)
        return 0; /* reference reads match[1] out of bounds here (ORBF:101); defined as "no matches" */
    int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)n_to);
    int32_t *i0 = buf, *i1 = buf + n_to, *d0 = buf + 2 * n_to, *d1 = buf + 3 * n_to;
    mso_match_knn2_raw(from_desc, n_from, to_desc, n_to, i0, i1, d0, d1);
    int n = 0;
    for(int q = 0; q < n_to; ++q)
    {
        const float f0 = (float)d0[q], f1 = (float)d1[q]; /* DMatch::distance is float */
        if((double)f0 < ratio * (double)f1)                /* ORBF:101 with ratio = 0.7 */
        {
            from_idx[n] = i0[q]; /* trainIdx  -> fromIndex, ORBF:112 */
            to_idx[n] = q;       /* queryIdx  -> toIndex */
            ++n;
        }
    }
    free(buf);
    return n;
}

/* ---- REF/rgbd_feature_frontend.cpp:101-138 -------------------------------------------------- */
void mso_backproject(const uint16_t* depth, int width, int height, float factor, double fx, double fy, double cx,
                     double cy, const float* xy, int n, double* xyz, uint8_t* valid)
{
    const double inv_fx = 1.0 / fx, inv_fy = 1.0 / fy; /* :126 invFocal = 1.0 / focal.array() */
    for(int i = 0; i < n; ++i)
    {
        const double x = xy[2 * i], y = xy[2 * i + 1];
        const int ix = (int)x, iy = (int)y; /* coordinates.cast<int>() (:129) */
        float d = 0.f;
        if(ix >= 0 && ix < width && iy >= 0 && iy < height)
            d = (float)depth[(size_t)width * iy + ix] * factor; /* getDepth, depth_frame.hpp:20-25 */
        const int ok = d > FLT_EPSILON;                         /* isDepthValid, depth_frame.hpp:27-30 */
        const double z = d;                                     /* :111 */
        xyz[3 * i] = ok ? (x - cx) * z * inv_fx : 0.0;          /* :112 */
        xyz[3 * i + 1] = ok ? (y - cy) * z * inv_fy : 0.0;      /* :113 */
        xyz[3 * i + 2] = ok ? z : 0.0;
        valid[i] = (uint8_t)ok;
    }
}

/* ---- DBoW3 --------------------------------------------------------------------------------- */
struct mso_voc
{
    int k, L, scoring, weighting;
    uint32_t n_nodes, n_words;
    uint32_t* parent;
    double* weight;
    uint8_t* desc;       /* n_nodes x 32 */
    uint32_t* child_off; /* CSR, children in stream order (PATCH:2626 push_back) */
    uint32_t* child;
    uint32_t* word_id;
    uint32_t* words; /* word -> node */
};

#define RD(dst, type)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        if(pos + sizeof(type) > size)                                                                                  \
            goto fail;                                                                                                 \
        memcpy(&(dst), p + pos, sizeof(type));                                                                         \
        pos += sizeof(type);                                                                                           \
    } while(0)

mso_voc* mso_voc_load(const void* blob, size_t size)
{
    /* Vocabulary::fromStream, PATCH:2544-2651 */
    const uint8_t* p = (const uint8_t*)blob;
    size_t pos = 0;
    mso_voc* v = (mso_voc*)calloc(1, sizeof(mso_voc));
    uint32_t* cnt = NULL;
    uint64_t sig;
    uint8_t compressed;
    RD(sig, uint64_t);
    if(sig != 88877711233ull)
        goto fail;
    RD(compressed, uint8_t);
    RD(v->n_nodes, uint32_t);
    if(compressed || v->n_nodes == 0)
        goto fail; /* QuickLZ streams are not handled by the oracle */
    RD(v->k, int32_t);
    RD(v->L, int32_t);
    RD(v->scoring, int32_t);
    RD(v->weighting, int32_t);
    const uint32_t n = v->n_nodes;
    v->parent = (uint32_t*)calloc(n, sizeof(uint32_t));
    v->weight = (double*)calloc(n, sizeof(double));
    v->desc = (uint8_t*)calloc(n, 32);
    v->child_off = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    v->child = (uint32_t*)calloc(n, sizeof(uint32_t));
    v->word_id = (uint32_t*)calloc(n, sizeof(uint32_t));
    uint32_t* order = (uint32_t*)calloc(n, sizeof(uint32_t)); /* stream order of node ids */
    cnt = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    for(uint32_t i = 1; i < n; ++i)
    {
        uint32_t nid, par;
        int32_t cols, rows, type;
        double w;
        RD(nid, uint32_t);
        RD(par, uint32_t);
        RD(w, double);
        RD(cols, int32_t); /* DescManip::fromStream: cols, rows, type, then elemSize*cols bytes */
        RD(rows, int32_t);
        RD(type, int32_t);
        if(nid >= n || par >= n || cols != 32 || rows != 1 || type != 0 /* CV_8UC1 */ || pos + 32 > size)
        {
            free(order);
            goto fail;
        }
        v->parent[nid] = par;
        v->weight[nid] = w;
        memcpy(v->desc + (size_t)nid * 32, p + pos, 32);
        pos += 32;
        order[i] = nid;
        cnt[par]++;
    }
    for(uint32_t i = 0; i < n; ++i)
        v->child_off[i + 1] = v->child_off[i] + cnt[i];
    memset(cnt, 0, sizeof(uint32_t) * ((size_t)n + 1));
    for(uint32_t i = 1; i < n; ++i)
    {
        const uint32_t nid = order[i], par = v->parent[nid];
        v->child[v->child_off[par] + cnt[par]++] = nid;
    }
    free(order);
    RD(v->n_words, uint32_t);
    v->words = (uint32_t*)calloc(v->n_words ? v->n_words : 1, sizeof(uint32_t));
    for(uint32_t i = 0; i < v->n_words; ++i)
    {
        uint32_t wid, nid;
        RD(wid, uint32_t);
        RD(nid, uint32_t);
        if(wid >= v->n_words || nid >= n)
            goto fail;
        v->word_id[nid] = wid;
        v->words[wid] = nid;
    }
    free(cnt);
    return v;
fail:
    free(cnt);
    mso_voc_free(v);
    return NULL;
}
#undef RD

void mso_voc_free(mso_voc* v)
{
    if(!v)
        return;
    free(v->parent);
    free(v->weight);
    free(v->desc);
    free(v->child_off);
    free(v->child);
    free(v->word_id);
    free(v->words);
    free(v);
}

int mso_voc_info(const mso_voc* v, int* k, int* L, int* n_nodes, int* n_words, int* scoring, int* weighting)
{
    if(!v)
        return -1;
    *k = v->k, *L = v->L, *n_nodes = (int)v->n_nodes, *n_words = (int)v->n_words;
    *scoring = v->scoring, *weighting = v->weighting;
    return 0;
}

/* PATCH:1760-1860 — greedy descent; strict `<` so the first child wins ties */
void mso_bow_words(const mso_voc* v, const uint8_t* desc, int n, uint32_t* word, double* weight)
{
    for(int r = 0; r < n; ++r)
    {
        uint32_t final_id = 0;
        do
        {
            const uint32_t b = v->child_off[final_id], e = v->child_off[final_id + 1];
            uint64_t best_d = UINT64_MAX;
            for(uint32_t c = b; c < e; ++c)
            {
                const uint32_t id = v->child[c];
                const uint64_t dist = (uint64_t)hamming256(desc + (size_t)r * 32, v->desc + (size_t)id * 32);
                if(dist < best_d)
                {
                    best_d = dist;
                    final_id = id;
                }
            }
        } while(v->child_off[final_id] != v->child_off[final_id + 1]); /* !isLeaf() */
        word[r] = v->word_id[final_id];
        weight[r] = v->weight[final_id];
    }
}

static int cmp_u32(const void* a, const void* b)
{
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : x > y;
}

/* PATCH:1432-1530 with BowVector::addWeight / addIfNotExist / normalize (DBoW3 BowVector.cpp) */
int mso_bow_vector(const mso_voc* v, const uint8_t* desc, int n, uint32_t* words, double* values)
{
    if(n <= 0)
        return 0;
    uint32_t* wid = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    double* wgt = (double*)malloc(sizeof(double) * (size_t)n);
    mso_bow_words(v, desc, n, wid, wgt);
    /* std::map<WordId, WordValue>: gather the distinct ids in ascending order, then replay the
     * insertions in feature order so every += happens in the reference's order. */
    uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    int nk = 0;
    for(int r = 0; r < n; ++r)
        if(wgt[r] > 0) /* "not stopped" */
            keys[nk++] = wid[r];
    qsort(keys, (size_t)nk, sizeof(uint32_t), cmp_u32);
    int m = 0;
    for(int i = 0; i < nk; ++i)
        if(m == 0 || keys[i] != words[m - 1])
            words[m++] = keys[i];
    uint8_t* seen = (uint8_t*)calloc((size_t)(m ? m : 1), 1);
    for(int i = 0; i < m; ++i)
        values[i] = 0;
    const int tf = v->weighting == MSO_TF || v->weighting == MSO_TF_IDF;
    for(int r = 0; r < n; ++r)
    {
        if(!(wgt[r] > 0))
            continue;
        int lo = 0, hi = m - 1;
        while(lo < hi)
        {
            const int mid = (lo + hi) / 2;
            if(words[mid] < wid[r])
                lo = mid + 1;
            else
                hi = mid;
        }
        if(tf)
        { /* addWeight: insert with v, or += v */
            if(!seen[lo])
                values[lo] = wgt[r], seen[lo] = 1;
            else
                values[lo] += wgt[r];
        }
        else if(!seen[lo]) /* addIfNotExist */
            values[lo] = wgt[r], seen[lo] = 1;
    }
    /* ScoringObject::mustNormalize: L1 for L1/CHI/KL/BHATTACHARYYA, L2 for L2, none for DOT */
    const int must = v->scoring != MSO_DOT_PRODUCT;
    if(tf && m > 0 && !must)
    {
        const double nd = (double)m;
        for(int i = 0; i < m; ++i)
            values[i] /= nd;
    }
    if(must)
    {
        double norm = 0.0;
        if(v->scoring == MSO_L2_NORM)
        {
            for(int i = 0; i < m; ++i)
                norm += values[i] * values[i];
            norm = sqrt(norm);
        }
        else
            for(int i = 0; i < m; ++i)
                norm += fabs(values[i]);
        if(norm > 0.0)
            for(int i = 0; i < m; ++i)
                values[i] /= norm;
    }
    free(seen);
    free(keys);
    free(wid);
    free(wgt);
    return m;
}

/* DBoW3 L1Scoring::score */
double mso_bow_score_l1(const uint32_t* w1, const double* v1, int n1, const uint32_t* w2, const double* v2, int n2)
{
    int i = 0, j = 0;
    double score = 0;
    while(i < n1 && j < n2)
    {
        if(w1[i] == w2[j])
        {
            score += fabs(v1[i] - v2[j]) - fabs(v1[i]) - fabs(v2[j]);
            ++i, ++j;
        }
        else if(w1[i] < w2[j])
            ++i; /* lower_bound jump: same visiting order of the common words */
        else
            ++j;
    }
    return -score / 2.0;
}

/* ==== cv::ORB detector mode (orb_feature.cpp:25,33-65 -> OpenCV 4.8.1 features2d/src/orb.cpp) ================ */

void mso_cvorb_default_params(mso_cvorb_params* p)
{
    p->n_features = 1000; /* orb_feature.cpp:25 */
    p->scale_factor = 1.2f;
    p->n_levels = 8;
    p->edge_threshold = 31;
    p->fast_threshold = 20;
    p->order = MSO_ORDER_LIBSTDCXX;
}

/* orb.cpp: getScale(level, firstLevel = 0, scaleFactor) = (float)std::pow(scaleFactor, (double)level) with the
 * double member scaleFactor holding the float argument of ORB::create; sizes in detectAndCompute:
 * cvRound(cols * inv_scale), inv_scale = 1.0f / scale (float arithmetic); quota in computeKeyPoints. */
void mso_cvorb_geometry(int W, int H, const mso_cvorb_params* p, int* w, int* h, float* scale, int* quota)
{
    const double sf = (double)p->scale_factor;
    for(int l = 0; l < p->n_levels; ++l)
    {
        scale[l] = (float)pow(sf, (double)l);
        const float inv = 1.0f / scale[l];
        w[l] = cv_round_f((float)W * inv);
        h[l] = cv_round_f((float)H * inv);
    }
    const float factor = (float)(1.0 / sf);
    float desired = (float)p->n_features * (1 - factor) / (1 - (float)pow((double)factor, (double)p->n_levels));
    int sum = 0;
    for(int l = 0; l < p->n_levels - 1; ++l)
    {
        quota[l] = cv_round_f(desired);
        sum += quota[l];
        desired *= factor;
    }
    quota[p->n_levels - 1] = p->n_features - sum > 0 ? p->n_features - sum : 0;
}

/* interpolationLinear<uchar>::getCoeffs (resize.cpp): offsets, 8.8 coefficients (ufixedpoint16), and the range
 * [dmin, dmax) of destination positions that interpolate; positions outside copy the first / last source sample */
static void exact_axis(int ssize, int dsize, int* ofs, uint16_t* c0, uint16_t* c1, int* dmin, int* dmax)
{
    const double scale = 1.0 / ((double)dsize / (double)ssize); /* softdouble::one() / softdouble(inv_scale) */
    int mn = 0, mx = dsize;
    for(int d = 0; d < dsize; ++d)
    {
        const double fval = scale * ((double)d + 0.5) - 0.5;
        const int ival = (int)floor(fval);
        ofs[d] = 0, c0[d] = 256, c1[d] = 0;
        if(ival >= 0 && ssize > 1)
        {
            if(ival < ssize - 1)
            {
                ofs[d] = ival;
                c1[d] = (uint16_t)cv_round_d((fval - (double)ival) * 256.0); /* ufixedpoint16(softdouble) */
                c0[d] = (uint16_t)(256 - c1[d]);
            }
            else
            {
                ofs[d] = ssize - 1;
                if(d < mx)
                    mx = d;
            }
        }
        else if(d + 1 > mn)
            mn = d + 1;
    }
    *dmin = mn;
    *dmax = mx;
}

void mso_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    int* xo = (int*)malloc(sizeof(int) * (size_t)(dw + dh));
    int* yo = xo + dw;
    uint16_t* xc0 = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)(dw + dh));
    uint16_t *xc1 = xc0 + dw, *yc0 = xc1 + dw, *yc1 = yc0 + dh;
    int xmin, xmax, ymin, ymax;
    exact_axis(sw, dw, xo, xc0, xc1, &xmin, &xmax);
    exact_axis(sh, dh, yo, yc0, yc1, &ymin, &ymax);
    uint16_t* line0 = (uint16_t*)malloc(sizeof(uint16_t) * 2 * (size_t)dw);
    uint16_t* line1 = line0 + dw;
    for(int y = 0; y < dh; ++y)
    {
        /* hlineResizeCn<uchar, ufixedpoint16, 2, true, 1>: 8.8 values; left / right of the range = edge sample */
        const int r0 = y < ymin ? 0 : (y >= ymax ? sh - 1 : yo[y]);
        const int two = y >= ymin && y < ymax;
        for(int k = 0; k <= two; ++k)
        {
            const uint8_t* s = src + (size_t)(r0 + k) * sw;
            uint16_t* ln = k ? line1 : line0;
            for(int x = 0; x < dw; ++x)
            {
                if(x < xmin)
                    ln[x] = (uint16_t)(s[0] << 8);
                else if(x >= xmax)
                    ln[x] = (uint16_t)(s[sw - 1] << 8);
                else
                    ln[x] = (uint16_t)(xc0[x] * s[xo[x]] + xc1[x] * s[xo[x] + 1]);
            }
        }
        uint8_t* d = dst + (size_t)y * dw;
        for(int x = 0; x < dw; ++x)
        {
            if(two) /* vlineResize<uchar, ufixedpoint16, 2>: 16.16 sum, (v + 2^15) >> 16 */
                d[x] = (uint8_t)(((uint32_t)line0[x] * yc0[y] + (uint32_t)line1[x] * yc1[y] + 32768u) >> 16);
            else /* vlineSet: ufixedpoint16 -> uchar = (v + 128) >> 8 */
                d[x] = (uint8_t)((line0[x] + 128u) >> 8);
        }
    }
    free(line0);
    free(xc0);
    free(xo);
}

/* orb.cpp HarrisResponses: 7x7 block of 3x3 Sobel-like gradients, integer sums, float response */
float mso_harris_response(const uint8_t* img, int step, int x, int y)
{
    const float harris_k = 0.04f;
    const float scale = 1.f / ((1 << 2) * 7 * 255.f);
    const float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for(int i = -3; i <= 3; ++i)
        for(int j = -3; j <= 3; ++j)
        {
            const uint8_t* p = img + (size_t)(y + i) * step + (x + j);
            const int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
            const int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* orb.cpp computeOrbDescriptors, WTA_K == 2 */
void mso_cvorb_descriptor(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc)
{
    float angle = angle_deg;
    angle *= (float)(3.1415926535897932384626433832795 / 180.f);
    /* a = cos, b = sin.  The reference calls the host libm's cosf / sinf (implementation-defined in the last bit); the
     * oracle takes the double-precision libm routines rounded to float, i.e. the correctly rounded float values — an
     * implementation INDEPENDENT of include/mslam_sincos.h, which only the product (k_describe.hip) evaluates, so the
     * GPU-vs-oracle comparison of the cv::ORB mode's descriptors is not a self-comparison. */
    const float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t* center = blurred + (size_t)y * step + x;
#define CV_VALUE(s)                                                                                                    \
    (center[cv_round_f((float)k_pattern[(s)] * b + (float)k_pattern[(s) + 1] * a) * step +                            \
            cv_round_f((float)k_pattern[(s)] * a - (float)k_pattern[(s) + 1] * b)])
    for(int i = 0; i < 32; ++i)
    {
        int val = 0;
        for(int k = 0; k < 8; ++k)
        {
            const int s = i * 32 + k * 4;
            val |= (CV_VALUE(s) < CV_VALUE(s + 2)) << k;
        }
        desc[i] = (uint8_t)val;
    }
#undef CV_VALUE
}

/* ---- KeyPointsFilter::retainBest in the ORDER a GCC build leaves (features2d/src/keypoint.cpp -> libstdc++) ----------------
 * retainBest = std::nth_element(begin, begin + n - 1, end, KeypointResponseGreater()), then std::partition(begin + n, end,
 * response >= keypoints[n - 1].response), then resize: which elements survive is defined by the standard (the SET
 * {response >= the n-th largest}), where they end up is not — it is whatever the library's introselect and partition do.
 * The reference is built with GCC (its CI and conan profile), i.e. libstdc++, whose algorithms have been the same since
 * GCC 4.9 (bits/stl_algo.h, bits/stl_heap.h): restated here function by function on an array of (response, payload index)
 * pairs, `comp(a, b)` = a.response > b.response.  tests/test_oracle_std_order.py compiles a C++ harness against the REAL
 * <algorithm> of this image and compares the two on random, tie-heavy and adversarial inputs: this part of the oracle is
 * pinned by the library itself. */
typedef struct
{
    float r;     /* response */
    int32_t idx; /* which keypoint */
} mso_rk;
#define RK_COMP(a, b) ((a).r > (b).r)
static void rk_swap(mso_rk* a, mso_rk* b)
{
    const mso_rk t = *a;
    *a = *b;
    *b = t;
}
/* std::__push_heap (stl_heap.h), comparator __iter_comp_val */
static void rk_push_heap(mso_rk* first, int hole, int top, mso_rk value)
{
    int parent = (hole - 1) / 2;
    while(hole > top && RK_COMP(first[parent], value))
    {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
/* std::__adjust_heap */
static void rk_adjust_heap(mso_rk* first, int hole, int len, mso_rk value)
{
    const int top = hole;
    int second = hole;
    while(second < (len - 1) / 2)
    {
        second = 2 * (second + 1);
        if(RK_COMP(first[second], first[second - 1]))
            second--;
        first[hole] = first[second];
        hole = second;
    }
    if((len & 1) == 0 && second == (len - 2) / 2)
    {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    rk_push_heap(first, hole, top, value);
}
/* std::__make_heap */
static void rk_make_heap(mso_rk* first, mso_rk* last)
{
    const int len = (int)(last - first);
    if(len < 2)
        return;
    int parent = (len - 2) / 2;
    for(;;)
    {
        const mso_rk value = first[parent];
        rk_adjust_heap(first, parent, len, value);
        if(parent == 0)
            return;
        parent--;
    }
}
/* std::__pop_heap(first, last, result) */
static void rk_pop_heap(mso_rk* first, mso_rk* last, mso_rk* result)
{
    const mso_rk value = *result;
    *result = *first;
    rk_adjust_heap(first, 0, (int)(last - first), value);
}
/* std::__heap_select */
static int g_rk_heap_selects = 0; /* (test hook: how often the depth-limit branch ran) */
int mso_std_heap_select_calls(void) { return g_rk_heap_selects; }
static void rk_heap_select(mso_rk* first, mso_rk* middle, mso_rk* last)
{
    ++g_rk_heap_selects;
    rk_make_heap(first, middle);
    for(mso_rk* i = middle; i < last; ++i)
        if(RK_COMP(*i, *first))
            rk_pop_heap(first, middle, i);
}
/* std::__move_median_to_first(result, a, b, c) */
static void rk_move_median_to_first(mso_rk* result, mso_rk* a, mso_rk* b, mso_rk* c)
{
    if(RK_COMP(*a, *b))
    {
        if(RK_COMP(*b, *c))
            rk_swap(result, b);
        else if(RK_COMP(*a, *c))
            rk_swap(result, c);
        else
            rk_swap(result, a);
    }
    else if(RK_COMP(*a, *c))
        rk_swap(result, a);
    else if(RK_COMP(*b, *c))
        rk_swap(result, c);
    else
        rk_swap(result, b);
}
/* std::__unguarded_partition(first, last, pivot) */
static mso_rk* rk_unguarded_partition(mso_rk* first, mso_rk* last, const mso_rk* pivot)
{
    for(;;)
    {
        while(RK_COMP(*first, *pivot))
            ++first;
        --last;
        while(RK_COMP(*pivot, *last))
            --last;
        if(!(first < last))
            return first;
        rk_swap(first, last);
        ++first;
    }
}
/* std::__unguarded_partition_pivot */
static mso_rk* rk_unguarded_partition_pivot(mso_rk* first, mso_rk* last)
{
    mso_rk* mid = first + (last - first) / 2;
    rk_move_median_to_first(first, first + 1, mid, last - 1);
    return rk_unguarded_partition(first + 1, last, first);
}
/* std::__insertion_sort (with std::__unguarded_linear_insert) */
static void rk_insertion_sort(mso_rk* first, mso_rk* last)
{
    if(first == last)
        return;
    for(mso_rk* i = first + 1; i != last; ++i)
    {
        if(RK_COMP(*i, *first))
        {
            const mso_rk val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(mso_rk)); /* std::move_backward(first, i, i + 1) */
            *first = val;
        }
        else
        {
            const mso_rk val = *i;
            mso_rk* lastp = i;
            mso_rk* next = i - 1;
            while(RK_COMP(val, *next))
            {
                *lastp = *next;
                lastp = next;
                --next;
            }
            *lastp = val;
        }
    }
}
/* std::nth_element = std::__introselect(first, nth, last, 2 * std::__lg(last - first)) */
static void rk_nth_element(mso_rk* first, mso_rk* nth, mso_rk* last)
{
    if(first == last || nth == last)
        return;
    int depth = 0;
    for(long v = last - first; v > 1; v >>= 1)
        ++depth; /* std::__lg */
    depth *= 2;
    while(last - first > 3)
    {
        if(depth == 0)
        {
            rk_heap_select(first, nth + 1, last);
            rk_swap(first, nth); /* "Place the nth largest element in its final position." */
            return;
        }
        --depth;
        mso_rk* cut = rk_unguarded_partition_pivot(first, last);
        if(cut <= nth)
            first = cut;
        else
            last = cut;
    }
    rk_insertion_sort(first, last);
}
/* std::partition for bidirectional iterators (std::__partition(..., bidirectional_iterator_tag)), pred = response >= thr */
static mso_rk* rk_partition_ge(mso_rk* first, mso_rk* last, float thr)
{
    for(;;)
    {
        for(;;)
            if(first == last)
                return first;
            else if(first->r >= thr)
                ++first;
            else
                break;
        --last;
        for(;;)
            if(first == last)
                return first;
            else if(!(last->r >= thr))
                --last;
            else
                break;
        rk_swap(first, last);
        ++first;
    }
}
/* test hook (tests/test_oracle_std_order.py): retainBest's two library calls on n responses; order[] receives the payload
 * indices of the survivors in their final places; returns how many survive */
int mso_std_retain_best_order(const float* response, int n, int n_points, int32_t* order)
{
    mso_rk* a = (mso_rk*)malloc(sizeof(mso_rk) * (size_t)(n > 0 ? n : 1));
    for(int i = 0; i < n; ++i)
        a[i].r = response[i], a[i].idx = i;
    int m = n;
    if(n_points >= 0 && n > n_points)
    {
        if(n_points == 0)
            m = 0;
        else
        {
            rk_nth_element(a, a + n_points - 1, a + n);
            const float ambiguous = a[n_points - 1].r;
            m = (int)(rk_partition_ge(a + n_points, a + n, ambiguous) - a);
        }
    }
    for(int i = 0; i < m; ++i)
        order[i] = a[i].idx;
    free(a);
    return m;
}
/* KeyPointsFilter::retainBest on the oracle's keypoint records, libstdc++ order */
static int retain_best_std(mso_cand* kp, int n, int n_points)
{
    if(n_points < 0 || n <= n_points)
        return n;
    float* r = (float*)malloc(sizeof(float) * (size_t)n);
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    mso_cand* tmp = (mso_cand*)malloc(sizeof(mso_cand) * (size_t)n);
    for(int i = 0; i < n; ++i)
        r[i] = kp[i].response, tmp[i] = kp[i];
    const int m = mso_std_retain_best_order(r, n, n_points, order);
    for(int i = 0; i < m; ++i)
        kp[i] = tmp[order[i]];
    free(r);
    free(order);
    free(tmp);
    return m;
}

/* KeyPointsFilter::retainBest as a SET: keep every element whose response is >= the n-th largest (ties kept);
 * stable (raster order preserved).  Returns the new count. */
static int retain_best(mso_cand* kp, int n, int n_points)
{
    if(n_points < 0 || n <= n_points)
        return n;
    if(n_points == 0)
        return 0;
    float* r = (float*)malloc(sizeof(float) * (size_t)n);
    for(int i = 0; i < n; ++i)
        r[i] = kp[i].response;
    /* n-th largest by selection (n is a few thousand at most) */
    for(int i = 0; i < n_points; ++i)
    {
        int m = i;
        for(int j = i + 1; j < n; ++j)
            if(r[j] > r[m])
                m = j;
        const float t = r[i];
        r[i] = r[m];
        r[m] = t;
    }
    const float thr = r[n_points - 1];
    free(r);
    int k = 0;
    for(int i = 0; i < n; ++i)
        if(kp[i].response >= thr)
            kp[k++] = kp[i];
    return k;
}

int mso_cvorb_level_keypoints(const uint8_t* img, int w, int h, const mso_cvorb_params* p, int quota, int stage,
                              mso_cand* out, int cap)
{
    /* FastFeatureDetector(fastThreshold, true)->detect on the level; runByImageBorder(edgeThreshold) */
    int n = mso_fast(img, w, w, h, p->fast_threshold, out, cap);
    int k = 0;
    const int e = p->edge_threshold;
    for(int i = 0; i < n; ++i)
        if(out[i].x >= (float)e && out[i].x < (float)(w - e) && out[i].y >= (float)e && out[i].y < (float)(h - e))
            out[k++] = out[i];
    /* p->order: MSO_ORDER_LIBSTDCXX = the order a GCC build of the reference leaves (default), MSO_ORDER_RASTER = the kept
     * set in FAST's raster order (what the standard alone defines) */
    n = p->order == MSO_ORDER_RASTER ? retain_best(out, k, 2 * quota) : retain_best_std(out, k, 2 * quota); /* HARRIS_SCORE: keep 2x, orb.cpp computeKeyPoints */
    if(stage == 0)
        return n;
    for(int i = 0; i < n; ++i)
        out[i].response = mso_harris_response(img, w, cv_round_f(out[i].x), cv_round_f(out[i].y));
    return p->order == MSO_ORDER_RASTER ? retain_best(out, n, quota) : retain_best_std(out, n, quota);
}

int mso_cvorb_detect(const uint8_t* bgr, int W, int H, const mso_cvorb_params* p, int max_out, float* xy, uint8_t* desc,
                     int32_t* octave, float* angle, float* response, int* n_out)
{
    *n_out = 0;
    if(W <= 0 || H <= 0 || p->n_levels < 1 || p->n_levels > MSO_MAX_LEVELS)
        return 0;
    int w[MSO_MAX_LEVELS], h[MSO_MAX_LEVELS], quota[MSO_MAX_LEVELS];
    float sf[MSO_MAX_LEVELS];
    mso_cvorb_geometry(W, H, p, w, h, sf, quota);
    uint8_t* pyr[MSO_MAX_LEVELS];
    pyr[0] = (uint8_t*)malloc((size_t)W * H);
    mso_gray(bgr, (size_t)W * H, pyr[0]); /* orb_feature.cpp:35 */
    for(int l = 1; l < p->n_levels; ++l)
    {
        pyr[l] = (uint8_t*)malloc((size_t)w[l] * h[l] + 1);
        mso_resize_linear_exact(pyr[l - 1], w[l - 1], h[l - 1], pyr[l], w[l], h[l]);
    }
    const int cap = W * H / 4 + 16;
    mso_cand* kp = (mso_cand*)malloc(sizeof(mso_cand) * (size_t)cap);
    uint8_t* blurred = (uint8_t*)malloc((size_t)W * H);
    int total = 0, rc = 0;
    for(int l = 0; l < p->n_levels; ++l)
    {
        if(w[l] <= 2 * p->edge_threshold || h[l] <= 2 * p->edge_threshold)
            continue;
        const int n = mso_cvorb_level_keypoints(pyr[l], w[l], h[l], p, quota[l], 1, kp, cap);
        if(n == 0)
            continue;
        mso_gaussian_blur7(pyr[l], w[l], h[l], blurred);
        for(int i = 0; i < n; ++i)
        {
            const int ix = cv_round_f(kp[i].x), iy = cv_round_f(kp[i].y);
            const float ang = mso_ic_angle(pyr[l], w[l], ix, iy); /* ICAngles: unblurred level */
            if(total < max_out)
            {
                mso_cvorb_descriptor(blurred, w[l], ix, iy, ang, desc + (size_t)total * 32);
                xy[2 * total] = kp[i].x * sf[l]; /* allKeypoints[i].pt *= scale */
                xy[2 * total + 1] = kp[i].y * sf[l];
                octave[total] = l;
                angle[total] = ang;
                response[total] = kp[i].response;
            }
            else
                rc = -1;
            ++total;
        }
    }
    *n_out = total;
    for(int l = 0; l < p->n_levels; ++l)
        free(pyr[l]);
    free(kp);
    free(blurred);
    return rc;
}

/* test hooks: the PRODUCT's routine compiled for the host (so that a CPU test can examine it against libm; no algorithm of
 * the oracle calls it) and the host libm's float cos/sin (what the reference calls) */
#include "../include/mslam_sincos.h"
void mso_sincos_f32(float x, float* s, float* c) { mslam_sincos_f32(x, s, c); }
void mso_libm_sincosf(float x, float* s, float* c)
{
    *s = sinf(x);
    *c = cosf(x);
}

/* flat word assignment (SURVEY.md §8d bow_flat; not a DBoW3 function): the word whose leaf descriptor has the least
 * Hamming distance over ALL words, lower word id on ties — the exhaustive search the descent approximates */
void mso_bow_words_flat(const mso_voc* v, const uint8_t* desc, int n, uint32_t* word, double* weight)
{
    for(int r = 0; r < n; ++r)
    {
        int best_d = 1 << 30;
        uint32_t best_w = 0;
        for(uint32_t w = 0; w < v->n_words; ++w)
        {
            const int d = hamming256(desc + (size_t)r * 32, v->desc + (size_t)v->words[w] * 32);
            if(d < best_d)
                best_d = d, best_w = w;
        }
        word[r] = best_w;
        weight[r] = v->weight[v->words[best_w]];
    }
}
