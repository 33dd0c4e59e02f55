/*
 * oracle.c -- CPU restatement of the Map2DFusion multi-band hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h for the parity status header).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>
#ifdef ORC_OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ helpers */

/* cv::borderInterpolate for BORDER_REFLECT (delta=0) / BORDER_REFLECT_101 (delta=1) */
static inline int orc_border(int p, int len, int delta)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p - 1 + delta;
        else       p = len - 1 - (p - len) - delta;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* cvRound: SSE2 cvtsd2si, round-half-to-even under the default MXCSR */
static inline int orc_cvround(double v) { return (int)lrint(v); }

static inline int orc_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int16_t orc_sat_short_f(float t) { return (int16_t)orc_sat_short(orc_cvround((double)t)); }
static inline uint8_t orc_sat_uchar(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ----------------------------------------------------------------- geometry */
/* quaternion product, GSLAM/GSLAM/core/SO3.h:435-442; q = (x,y,z,w)          */
static void q_mul(const double a[4], const double b[4], double o[4])
{
    const double x = a[0], y = a[1], z = a[2], w = a[3];
    double r0 = w * b[0] + x * b[3] + y * b[2] - z * b[1];
    double r1 = w * b[1] + y * b[3] + z * b[0] - x * b[2];
    double r2 = w * b[2] + z * b[3] + x * b[1] - y * b[0];
    double r3 = w * b[3] - x * b[0] - y * b[1] - z * b[2];
    o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}

/* SO3 * Point3: (q * (p,0)) * q^-1, SO3.h:445-450 */
void orc_so3_rotate(const double q[4], const double p[3], double out[3])
{
    double pq[4] = { p[0], p[1], p[2], 0 }, qi[4] = { -q[0], -q[1], -q[2], q[3] }, t[4], r[4];
    q_mul(q, pq, t);
    q_mul(t, qi, r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}

/* SE3::inverse, SE3.h:70-73: rinv = r.inv(); t' = -(rinv * t) */
void orc_se3_inverse(const double a[7], double out[7])
{
    double qi[4] = { -a[3], -a[4], -a[5], a[6] }, t[3];
    orc_so3_rotate(qi, a, t);
    out[0] = -t[0]; out[1] = -t[1]; out[2] = -t[2];
    out[3] = qi[0]; out[4] = qi[1]; out[5] = qi[2]; out[6] = qi[3];
}

/* SE3 * SE3, SE3.h:85-90: (r*r2, t + r*t2) */
void orc_se3_mul(const double a[7], const double b[7], double out[7])
{
    double q[4], t[3];
    q_mul(a + 3, b + 3, q);
    orc_so3_rotate(a + 3, b, t);
    out[0] = a[0] + t[0]; out[1] = a[1] + t[1]; out[2] = a[2] + t[2];
    out[3] = q[0]; out[4] = q[1]; out[5] = q[2]; out[6] = q[3];
}

/* --------------------------------------------------------------- OpenCV ops */

/* cv::getPerspectiveTransform (imgwarp.cpp): system set-up as published; the
 * -sx*dx products are formed in float (Point2f) before widening.  Solver:
 * partial-pivot Gaussian elimination in double (documented deviation).      */
void orc_get_perspective_transform(const float src[8], const float dst[8], double M[9])
{
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        const float sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        double* r0 = a[i]; double* r1 = a[i + 4];
        r0[0] = r1[3] = sx; r0[1] = r1[4] = sy; r0[2] = r1[5] = 1;
        r0[3] = r0[4] = r0[5] = r1[0] = r1[1] = r1[2] = 0;
        r0[6] = (double)(-sx * dx); r0[7] = (double)(-sy * dx);
        r1[6] = (double)(-sx * dy); r1[7] = (double)(-sy * dy);
        r0[8] = dx; r1[8] = dy;
    }
    for (int c = 0; c < 8; c++) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 8; r++) { double v = fabs(a[r][c]); if (v > best) { best = v; piv = r; } }
        if (piv != c) for (int k = 0; k < 9; k++) { double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
        if (a[c][c] == 0) continue;
        for (int r = c + 1; r < 8; r++) {
            const double f = a[r][c] / a[c][c];
            if (f == 0) continue;
            for (int k = c; k < 9; k++) a[r][k] = a[r][k] - f * a[c][k];
        }
    }
    double x[8];
    for (int r = 7; r >= 0; r--) {
        double s = a[r][8];
        for (int k = r + 1; k < 8; k++) s = s - a[r][k] * x[k];
        x[r] = a[r][r] != 0 ? s / a[r][r] : 0;
    }
    for (int i = 0; i < 8; i++) M[i] = x[i];
    M[8] = 1.;
}

/* cv::invert 3x3 double closed form (core/src/lapack.cpp, n==3 branch) */
int orc_invert3x3(const double S[9], double D[9])
{
#define Sd(r,c) S[(r)*3+(c)]
    double d = Sd(0,0) * (Sd(1,1) * Sd(2,2) - Sd(1,2) * Sd(2,1))
             - Sd(0,1) * (Sd(1,0) * Sd(2,2) - Sd(1,2) * Sd(2,0))
             + Sd(0,2) * (Sd(1,0) * Sd(2,1) - Sd(1,1) * Sd(2,0));
    if (d == 0.) return 0;
    d = 1. / d;
    double t[9];
    t[0] = (Sd(1,1) * Sd(2,2) - Sd(1,2) * Sd(2,1)) * d;
    t[1] = (Sd(0,2) * Sd(2,1) - Sd(0,1) * Sd(2,2)) * d;
    t[2] = (Sd(0,1) * Sd(1,2) - Sd(0,2) * Sd(1,1)) * d;
    t[3] = (Sd(1,2) * Sd(2,0) - Sd(1,0) * Sd(2,2)) * d;
    t[4] = (Sd(0,0) * Sd(2,2) - Sd(0,2) * Sd(2,0)) * d;
    t[5] = (Sd(0,2) * Sd(1,0) - Sd(0,0) * Sd(1,2)) * d;
    t[6] = (Sd(1,0) * Sd(2,1) - Sd(1,1) * Sd(2,0)) * d;
    t[7] = (Sd(0,1) * Sd(2,0) - Sd(0,0) * Sd(2,1)) * d;
    t[8] = (Sd(0,0) * Sd(1,1) - Sd(0,1) * Sd(1,0)) * d;
#undef Sd
    memcpy(D, t, sizeof(t));
    return 1;
}

/* Mat::convertTo call sites MultiBandMap2DCPU.cpp:445,447,156,839 */
void orc_convert_8u_16s(const uint8_t* s, size_t n, int16_t* d) { for (size_t i = 0; i < n; i++) d[i] = s[i]; }
void orc_convert_8u_32f_scaled(const uint8_t* s, size_t n, float* d)
{
    const float a = (float)(1. / 255.);
    for (size_t i = 0; i < n; i++) d[i] = (float)s[i] * a + 0.f;
}
void orc_convert_16s_8u(const int16_t* s, size_t n, uint8_t* d) { for (size_t i = 0; i < n; i++) d[i] = orc_sat_uchar(s[i]); }

/* radial weight image, MultiBandMap2DCPU.cpp:400-418 (all float; w/2 is an
 * integer division first) */
void orc_weight_image(float* p, int h, int w, int weight_type)
{
    float x_center = w / 2;
    float y_center = h / 2;
    float dis_max = sqrtf(x_center * x_center + y_center * y_center);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float dis = (i - y_center) * (i - y_center) + (j - x_center) * (j - x_center);
            dis = 1 - sqrtf(dis) / dis_max;
            if (0 == weight_type) *p = dis;
            else *p = dis * dis;
            if (*p <= 1e-5) *p = 1e-5;
            p++;
        }
}

/* warpPerspective INTER_NEAREST + BORDER_CONSTANT(0), 1 or more channels */
void orc_warp_nearest_const_32f(const float* src, int srows, int scols, int cn,
                                float* dst, int drows, int dcols, const double M0[9])
{
    double M[9];
    if (!orc_invert3x3(M0, M)) memset(M, 0, sizeof(M));
    int bh0 = drows < 16 ? drows : 16;
    int bw0 = (1024 / bh0) < dcols ? (1024 / bh0) : dcols;
#ifdef ORC_OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < drows; y++) {
        float* D = dst + (size_t)y * dcols * cn;
        for (int xb = 0; xb < dcols; xb += bw0) {
            const int bw = bw0 < dcols - xb ? bw0 : dcols - xb;
            const double X0 = M[0] * xb + M[1] * y + M[2];
            const double Y0 = M[3] * xb + M[4] * y + M[5];
            const double W0 = M[6] * xb + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; x1++) {
                double W = W0 + M[6] * x1;
                W = W ? 1. / W : 0;
                double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
                fX = fX < (double)INT_MAX ? fX : (double)INT_MAX; fX = fX > (double)INT_MIN ? fX : (double)INT_MIN;
                fY = fY < (double)INT_MAX ? fY : (double)INT_MAX; fY = fY > (double)INT_MIN ? fY : (double)INT_MIN;
                const int sx = orc_sat_short(orc_cvround(fX)), sy = orc_sat_short(orc_cvround(fY));
                float* d = D + (size_t)(xb + x1) * cn;
                if ((unsigned)sx < (unsigned)scols && (unsigned)sy < (unsigned)srows) {
                    const float* S = src + ((size_t)sy * scols + sx) * cn;
                    for (int k = 0; k < cn; k++) d[k] = S[k];
                } else
                    for (int k = 0; k < cn; k++) d[k] = 0.f;
            }
        }
    }
}

/* ---- 8-bit fixed-point bilinear remap (imgwarp.cpp remapBilinear<FixedPtCast<int,uchar,15>,...,short>) ----
 * integer tap weights = saturate_cast<short>(w*32768); all bilinear products on the 1/32 grid are exact
 * integers summing to 32768 except at (0,0), where 32768 saturates to 32767 and initInterTab2D's fix-up
 * adds the missing 1 to another tap -- neither changes any output ((S*32767 + T + 16384) >> 15 == S). */
static void orc_itab(int fxi, int fyi, int w[4])
{
    const float fx = fxi * (1.f / 32), fy = fyi * (1.f / 32);
    const float v[4] = { (1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx };
    int sum = 0;
    for (int k = 0; k < 4; k++) { int t = orc_cvround((double)(v[k] * 32768.f)); w[k] = t > 32767 ? 32767 : t; sum += w[k]; }
    if (sum != 32768) w[3] += 32768 - sum;
}

void orc_warp_linear_const_8u(const uint8_t* src, int srows, int scols, int cn,
                              uint8_t* dst, int drows, int dcols, const double M0[9])
{
    double M[9];
    if (!orc_invert3x3(M0, M)) memset(M, 0, sizeof(M));
    int bh0 = drows < 16 ? drows : 16;
    int bw0 = (1024 / bh0) < dcols ? (1024 / bh0) : dcols;
    const size_t sstep = (size_t)scols * cn;
    for (int y = 0; y < drows; y++) {
        uint8_t* D = dst + (size_t)y * dcols * cn;
        for (int xb = 0; xb < dcols; xb += bw0) {
            const int bw = bw0 < dcols - xb ? bw0 : dcols - xb;
            const double X0 = M[0] * xb + M[1] * y + M[2];
            const double Y0 = M[3] * xb + M[4] * y + M[5];
            const double W0 = M[6] * xb + M[7] * y + M[8];
            for (int x1 = 0; x1 < bw; x1++) {
                double W = W0 + M[6] * x1;
                W = W ? 32. / W : 0;
                double fX = (X0 + M[0] * x1) * W, fY = (Y0 + M[3] * x1) * W;
                fX = fX < (double)INT_MAX ? fX : (double)INT_MAX; fX = fX > (double)INT_MIN ? fX : (double)INT_MIN;
                fY = fY < (double)INT_MAX ? fY : (double)INT_MAX; fY = fY > (double)INT_MIN ? fY : (double)INT_MIN;
                const int X = orc_cvround(fX), Y = orc_cvround(fY);
                const int sx = orc_sat_short(X >> 5), sy = orc_sat_short(Y >> 5);
                int w[4];
                orc_itab(X & 31, Y & 31, w);
                uint8_t* d = D + (size_t)(xb + x1) * cn;
                if (sx >= scols || sx + 1 < 0 || sy >= srows || sy + 1 < 0) {
                    for (int k = 0; k < cn; k++) d[k] = 0;
                    continue;
                }
                const int in00 = sx >= 0 && sy >= 0, in01 = sx + 1 < scols && sy >= 0;
                const int in10 = sx >= 0 && sy + 1 < srows, in11 = sx + 1 < scols && sy + 1 < srows;
                for (int k = 0; k < cn; k++) {
                    const int v0 = in00 ? src[(size_t)sy * sstep + (size_t)sx * cn + k] : 0;
                    const int v1 = in01 ? src[(size_t)sy * sstep + (size_t)(sx + 1) * cn + k] : 0;
                    const int v2 = in10 ? src[(size_t)(sy + 1) * sstep + (size_t)sx * cn + k] : 0;
                    const int v3 = in11 ? src[(size_t)(sy + 1) * sstep + (size_t)(sx + 1) * cn + k] : 0;
                    d[k] = orc_sat_uchar((v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << 14)) >> 15);
                }
            }
        }
    }
}

/* Map2DCPU.cpp:245-263 */
void orc_weight_image_8uc4(uint8_t* p, int h, int w, int weight_type)
{
    float x_center = w / 2;
    float y_center = h / 2;
    float dis_max = sqrtf(x_center * x_center + y_center * y_center);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float dis = (i - y_center) * (i - y_center) + (j - x_center) * (j - x_center);
            dis = 1 - sqrtf(dis) / dis_max;
            p[1] = p[2] = p[0] = 0;
            if (0 == weight_type) p[3] = (uint8_t)(dis * 254.);
            else p[3] = (uint8_t)(dis * dis * 254);
            if (p[3] < 2) p[3] = 2;
            p += 4;
        }
}

/* --- 16S instantiation --- */
#define T int16_t
#define WT int
#define SFX 16s
#define ORC_IS_FLOAT 0
#define CAST_DOWN(v) ((int16_t)(((v) + 128) >> 8))
#define CAST_UP(v)   ((int16_t)(((v) + 32) >> 6))
#define SAT_ADD(a,b) ((int16_t)orc_sat_short((int)(a) + (int)(b)))
#define SAT_SUB(a,b) ((int16_t)orc_sat_short((int)(a) - (int)(b)))
#define CAST_REMAP(t) orc_sat_short_f(t)
#include "oracle_pyr.inc"
#undef T
#undef WT
#undef SFX
#undef ORC_IS_FLOAT
#undef CAST_DOWN
#undef CAST_UP
#undef SAT_ADD
#undef SAT_SUB
#undef CAST_REMAP

/* --- 32F instantiation --- */
#define T float
#define WT float
#define SFX 32f
#define ORC_IS_FLOAT 1
#define CAST_DOWN(v) ((v) * (1.f / 256))
#define CAST_UP(v)   ((v) * (1.f / 64))
#define SAT_ADD(a,b) ((a) + (b))
#define SAT_SUB(a,b) ((a) - (b))
#define CAST_REMAP(t) (t)
#include "oracle_pyr.inc"
#undef T
#undef WT
#undef SFX
#undef ORC_IS_FLOAT
#undef CAST_DOWN
#undef CAST_UP
#undef SAT_ADD
#undef SAT_SUB
#undef CAST_REMAP

/* ------------------------------------------------------------------ the map */

typedef struct orc_tile {
    void*  lap[ORC_MAX_LEVELS];
    float* w[ORC_MAX_LEVELS];
    uint8_t* bgra;      /* Map2DCPU mode: ele->img (256x256 8UC4) */
    int    has_pyr;     /* pyr_laplace.size() != 0 (Map2DCPU mode: !img.empty()) */
    int    changed;
} orc_tile;

struct orc_map {
    orc_options opt;
    int    band_num;                 /* effective, .cpp:263 */
    int    valid;
    double plane[7];
    double cam_w, cam_h, fx, fy, cx, cy, fxinv, fyinv;
    /* MultiBandMap2DCPUData */
    double ele_size, ele_size_inv, length_pixel, length_pixel_inv;
    double min[3], max[3];
    int    w, h;
    orc_tile** data;                 /* dense row-major, lazily allocated */
    int    off_x, off_y;             /* stable tile coord = dense + off   */
    float* weight_image; int wi_rows, wi_cols;
    uint8_t* weight_image8;          /* Map2DCPU mode */
    /* test hooks */
    int    keep_last;
    int    last_x0, last_y0, last_tx, last_ty; double last_M[9];
    void*  last_lap[ORC_MAX_LEVELS]; float* last_w[ORC_MAX_LEVELS];
};

static size_t elt_size(const orc_map* m) { return m->opt.force_float ? sizeof(float) : sizeof(int16_t); }

void orc_default_options(orc_options* o)
{
    o->band_num = 5; o->force_float = 0; o->weight_type = 0; o->high_quality = 1;
    o->bg_color = 0; o->resolution = 0; o->scale = 1; o->single_band = 0;
}

orc_map* orc_map_create(const orc_options* o)
{
    orc_map* m = (orc_map*)calloc(1, sizeof(orc_map));
    if (o) m->opt = *o; else orc_default_options(&m->opt);
    /* .cpp:260-263: min(BandNumber, ceil(log(256)/log(2))) */
    int lim = (int)ceil(log((double)ORC_ELE_PIXELS) / log(2.0));
    m->band_num = m->opt.band_num < lim ? m->opt.band_num : lim;
    return m;
}

static void free_tile(orc_tile* t)
{
    if (!t) return;
    for (int i = 0; i < ORC_MAX_LEVELS; i++) { free(t->lap[i]); free(t->w[i]); }
    free(t->bgra);
    free(t);
}

static void free_grid(orc_map* m)
{
    if (m->data) { for (int i = 0; i < m->w * m->h; i++) free_tile(m->data[i]); free(m->data); }
    m->data = NULL; m->w = m->h = 0;
}

static void free_last(orc_map* m)
{
    for (int i = 0; i < ORC_MAX_LEVELS; i++) { free(m->last_lap[i]); free(m->last_w[i]); m->last_lap[i] = NULL; m->last_w[i] = NULL; }
}

void orc_map_destroy(orc_map* m)
{
    if (!m) return;
    free_grid(m); free(m->weight_image); free(m->weight_image8); free_last(m); free(m);
}

/* Map2DPrepare::prepare (Map2D.cpp:32-49) + MultiBandMap2DCPUData::prepare
 * (MultiBandMap2DCPU.cpp:199-255).  A failed prepare leaves the old state. */
int orc_map_prepare(orc_map* m, const double plane[7], const double cam[6], int n, const double* poses7)
{
    if (n == 0 || cam[0] <= 0 || cam[1] <= 0 || cam[2] == 0 || cam[3] == 0) {
        fprintf(stderr, "Map2D::prepare:Not valid prepare!\n");
        return 0;
    }
    const double fxinv = 1. / cam[2], fyinv = 1. / cam[3];
    double pinv[7];
    orc_se3_inverse(plane, pinv);
    double mx[3] = { -1e10, -1e10, -1e10 }, mn[3] = { 1e10, 1e10, 1e10 };
    for (int i = 0; i < n; i++) {
        double p[7];
        orc_se3_mul(pinv, poses7 + 7 * i, p);
        for (int k = 0; k < 3; k++) {
            mx[k] = p[k] > mx[k] ? p[k] : mx[k];
            mn[k] = p[k] < mn[k] ? p[k] : mn[k];
        }
    }
    if (mn[2] * mx[2] <= 0) return 0;
    double maxh = mx[2] > 0 ? mx[2] : -mn[2];
    /* line = UnProject(w,h) - UnProject(0,0), Map2D.h:60-64 */
    double lx = (cam[0] - cam[4]) * fxinv - (0 - cam[4]) * fxinv;
    double ly = (cam[1] - cam[5]) * fyinv - (0 - cam[5]) * fyinv;
    double radius = 0.5 * maxh * sqrt((lx * lx + ly * ly));
    double length_pixel = m->opt.resolution;
    if (!length_pixel) {
        length_pixel = 2 * radius / sqrt(cam[0] * cam[0] + cam[1] * cam[1]);
        length_pixel /= m->opt.scale;
    }
    mn[0] = mn[0] - radius; mn[1] = mn[1] - radius; mn[2] = mn[2] - 0;
    mx[0] = mx[0] + radius; mx[1] = mx[1] + radius; mx[2] = mx[2] + 0;
    double c[3];
    for (int k = 0; k < 3; k++) c[k] = 0.5 * (mn[k] + mx[k]);
    for (int k = 0; k < 3; k++) { mn[k] = 2 * mn[k] - c[k]; mx[k] = 2 * mx[k] - c[k]; }
    double ele_size = ORC_ELE_PIXELS * length_pixel;
    int w = (int)ceil((mx[0] - mn[0]) / ele_size);
    int h = (int)ceil((mx[1] - mn[1]) / ele_size);
    mx[0] = mn[0] + ele_size * w;
    mx[1] = mn[1] + ele_size * h;

    /* commit (MultiBandMap2DCPU::prepare, .cpp:276-283) */
    free_grid(m);
    memcpy(m->plane, plane, sizeof(double) * 7);
    m->cam_w = cam[0]; m->cam_h = cam[1]; m->fx = cam[2]; m->fy = cam[3]; m->cx = cam[4]; m->cy = cam[5];
    m->fxinv = fxinv; m->fyinv = fyinv;
    m->length_pixel = length_pixel; m->length_pixel_inv = 1. / length_pixel;
    m->ele_size = ele_size; m->ele_size_inv = 1. / ele_size;
    memcpy(m->min, mn, sizeof(mn)); memcpy(m->max, mx, sizeof(mx));
    m->w = w; m->h = h; m->off_x = m->off_y = 0;
    m->data = (orc_tile**)calloc((size_t)w * h, sizeof(orc_tile*));
    free(m->weight_image); m->weight_image = NULL; m->wi_rows = m->wi_cols = 0;
    free(m->weight_image8); m->weight_image8 = NULL;
    m->valid = 1;
    return 1;
}

/* renderFrame step 1, .cpp:324-347 (pose already in plane coordinates) */
static int footprint_plane(const orc_map* m, const double pose[7], double pts[8])
{
    const double img[8] = { 0, 0, m->cam_w, 0, 0, m->cam_h, m->cam_w, m->cam_h };
    double down[3] = { 0, 0, -1 };
    if (pose[2] < 0) down[2] = 1;
    for (int i = 0; i < 4; i++) {
        double p[3] = { (img[2 * i] - m->cx) * m->fxinv, (img[2 * i + 1] - m->cy) * m->fyinv, 1. }, axis[3];
        orc_so3_rotate(pose + 3, p, axis);
        if (axis[0] * down[0] + axis[1] * down[1] + axis[2] * down[2] < 0.4) return 0;
        const double s = pose[2] / axis[2];
        pts[2 * i]     = pose[0] - s * axis[0];
        pts[2 * i + 1] = pose[1] - s * axis[1];
    }
    return 1;
}

int orc_map_footprint(orc_map* m, const double pose_world[7], double pts8[8])
{
    if (!m->valid) return 0;
    double pinv[7], pose[7];
    orc_se3_inverse(m->plane, pinv);
    orc_se3_mul(pinv, pose_world, pose);
    return footprint_plane(m, pose, pts8);
}

/* spreadMap, .cpp:561-604 */
static int spread_map(orc_map* m, double xmin, double ymin, double xmax, double ymax)
{
    int xminInt = (int)floor((xmin - m->min[0]) * m->ele_size_inv);
    int yminInt = (int)floor((ymin - m->min[1]) * m->ele_size_inv);
    int xmaxInt = (int)ceil((xmax - m->min[0]) * m->ele_size_inv);
    int ymaxInt = (int)ceil((ymax - m->min[1]) * m->ele_size_inv);
    xminInt = xminInt < 0 ? xminInt : 0; yminInt = yminInt < 0 ? yminInt : 0;
    xmaxInt = xmaxInt > m->w ? xmaxInt : m->w; ymaxInt = ymaxInt > m->h ? ymaxInt : m->h;
    const int w = xmaxInt - xminInt, h = ymaxInt - yminInt;
    double mnx = m->min[0] + m->ele_size * xminInt;
    double mny = m->min[1] + m->ele_size * yminInt;
    double mxx = mnx + w * m->ele_size;
    double mxy = mny + h * m->ele_size;
    orc_tile** nd = (orc_tile**)calloc((size_t)w * h, sizeof(orc_tile*));
    for (int x = 0; x < m->w; x++)
        for (int y = 0; y < m->h; y++)
            nd[x - xminInt + (y - yminInt) * w] = m->data[y * m->w + x];
    free(m->data);
    m->data = nd; m->w = w; m->h = h;
    m->min[0] = mnx; m->min[1] = mny; m->max[0] = mxx; m->max[1] = mxy;
    m->off_x += xminInt; m->off_y += yminInt;
    return 1;
}

static size_t level_pixels(int tiles_x, int tiles_y, int level)
{
    return (size_t)((tiles_x * ORC_ELE_PIXELS) >> level) * ((tiles_y * ORC_ELE_PIXELS) >> level);
}

/* renderFrame, .cpp:311-558 */
static int render_frame(orc_map* m, const uint8_t* bgr, int rows, int cols, const double pose[7])
{
    if (cols != m->cam_w || rows != m->cam_h) {
        fprintf(stderr, "MultiBandMap2DCPU::renderFrame: frame.first.cols!=p->_camera.w||frame.first.rows!=p->_camera.h||frame.first.type()!=CV_8UC3\n");
        return 0;
    }
    double pts[8];
    if (!footprint_plane(m, pose, pts)) return 0;
    double xmin = pts[0], xmax = xmin, ymin = pts[1], ymax = ymin;
    for (int i = 1; i < 4; i++) {
        if (pts[2 * i] < xmin) xmin = pts[2 * i];
        if (pts[2 * i + 1] < ymin) ymin = pts[2 * i + 1];
        if (pts[2 * i] > xmax) xmax = pts[2 * i];
        if (pts[2 * i + 1] > ymax) ymax = pts[2 * i + 1];
    }
    if (xmin < m->min[0] || xmax > m->max[0] || ymin < m->min[1] || ymax > m->max[1])
        if (!spread_map(m, xmin, ymin, xmax, ymax)) return 0;
    int xminInt = (int)floor((xmin - m->min[0]) * m->ele_size_inv);
    int yminInt = (int)floor((ymin - m->min[1]) * m->ele_size_inv);
    int xmaxInt = (int)ceil((xmax - m->min[0]) * m->ele_size_inv);
    int ymaxInt = (int)ceil((ymax - m->min[1]) * m->ele_size_inv);
    if (xminInt < 0 || yminInt < 0 || xmaxInt > m->w || ymaxInt > m->h || xminInt >= xmaxInt || yminInt >= ymaxInt) {
        fprintf(stderr, "MultiBandMap2DCPU::renderFrame:should never happen!\n");
        return 0;
    }
    xmin = m->min[0] + m->ele_size * xminInt;
    ymin = m->min[1] + m->ele_size * yminInt;
    xmax = m->min[0] + m->ele_size * xmaxInt;
    ymax = m->min[1] + m->ele_size * ymaxInt;
    (void)xmax; (void)ymax;

    if (m->opt.single_band) {
        /* Map2DCPU::renderFrame, Map2DCPU.cpp:236-334: BGRA source (alpha = weight byte), one 8-bit warp,
         * select `ele.a < dst.a` */
        if (!m->weight_image8 || m->wi_cols != cols || m->wi_rows != rows) {
            free(m->weight_image8);
            m->weight_image8 = (uint8_t*)malloc((size_t)rows * cols * 4);
            m->wi_rows = rows; m->wi_cols = cols;
            orc_weight_image_8uc4(m->weight_image8, rows, cols, m->opt.weight_type);
        }
        const size_t n = (size_t)rows * cols;
        uint8_t* srcb = (uint8_t*)malloc(n * 4);
        memcpy(srcb, m->weight_image8, n * 4);
        for (size_t i = 0; i < n; i++) { srcb[4 * i] = bgr[3 * i]; srcb[4 * i + 1] = bgr[3 * i + 1]; srcb[4 * i + 2] = bgr[3 * i + 2]; }
        const float s4[8] = { 0.f, 0.f, (float)m->cam_w, 0.f, 0.f, (float)m->cam_h, (float)m->cam_w, (float)m->cam_h };
        float d4[8];
        for (int i = 0; i < 4; i++) {
            d4[2 * i]     = (float)((pts[2 * i] - xmin) * m->length_pixel_inv);
            d4[2 * i + 1] = (float)((pts[2 * i + 1] - ymin) * m->length_pixel_inv);
        }
        double Ms[9];
        orc_get_perspective_transform(s4, d4, Ms);
        const int txs = xmaxInt - xminInt, tys = ymaxInt - yminInt;
        const int dr = tys * ORC_ELE_PIXELS, dc = txs * ORC_ELE_PIXELS;
        uint8_t* dstb = (uint8_t*)malloc((size_t)dr * dc * 4);
        orc_warp_linear_const_8u(srcb, rows, cols, 4, dstb, dr, dc, Ms);
        free(srcb);
        for (int x = xminInt; x < xmaxInt; x++)
            for (int y = yminInt; y < ymaxInt; y++) {
                orc_tile* ele = m->data[y * m->w + x];
                if (!ele) ele = m->data[y * m->w + x] = (orc_tile*)calloc(1, sizeof(orc_tile));
                if (!ele->bgra) ele->bgra = (uint8_t*)calloc((size_t)ORC_ELE_PIXELS * ORC_ELE_PIXELS, 4);
                for (int r = 0; r < ORC_ELE_PIXELS; r++) {
                    const uint8_t* dp = dstb + (((size_t)(y - yminInt) * ORC_ELE_PIXELS + r) * dc + (size_t)(x - xminInt) * ORC_ELE_PIXELS) * 4;
                    uint8_t* ep = ele->bgra + (size_t)r * ORC_ELE_PIXELS * 4;
                    for (int c = 0; c < ORC_ELE_PIXELS; c++)
                        if (ep[4 * c + 3] < dp[4 * c + 3]) memcpy(ep + 4 * c, dp + 4 * c, 4);
                }
                ele->has_pyr = 1; ele->changed = 1;
            }
        free(dstb);
        m->last_x0 = xminInt + m->off_x; m->last_y0 = yminInt + m->off_y; m->last_tx = txs; m->last_ty = tys;
        memcpy(m->last_M, Ms, sizeof(Ms));
        return 1;
    }
    /* 3. weight image (built once, cloned per frame: .cpp:396-425) */
    if (!m->weight_image || m->wi_cols != cols || m->wi_rows != rows) {
        free(m->weight_image);
        m->weight_image = (float*)malloc((size_t)rows * cols * sizeof(float));
        m->wi_rows = rows; m->wi_cols = cols;
        orc_weight_image(m->weight_image, rows, cols, m->opt.weight_type);
    }
    float* weight_src = (float*)malloc((size_t)rows * cols * sizeof(float));
    memcpy(weight_src, m->weight_image, (size_t)rows * cols * sizeof(float));

    /* .cpp:427-441 */
    const float src4[8] = { 0.f, 0.f, (float)m->cam_w, 0.f, 0.f, (float)m->cam_h, (float)m->cam_w, (float)m->cam_h };
    float dst4[8];
    for (int i = 0; i < 4; i++) {
        dst4[2 * i]     = (float)((pts[2 * i] - xmin) * m->length_pixel_inv);
        dst4[2 * i + 1] = (float)((pts[2 * i + 1] - ymin) * m->length_pixel_inv);
    }
    double M[9];
    orc_get_perspective_transform(src4, dst4, M);

    const int tx = xmaxInt - xminInt, ty = ymaxInt - yminInt, L = m->band_num;
    const int crow = ty * ORC_ELE_PIXELS, ccol = tx * ORC_ELE_PIXELS;
    const size_t es = elt_size(m);
    void* lap[ORC_MAX_LEVELS]; float* wp[ORC_MAX_LEVELS];
    for (int i = 0; i <= L; i++) {
        lap[i] = malloc(level_pixels(tx, ty, i) * 3 * es);
        wp[i]  = (float*)malloc(level_pixels(tx, ty, i) * sizeof(float));
    }
    /* .cpp:443-452 */
    const size_t npx = (size_t)rows * cols;
    if (m->opt.force_float) {
        float* img_src = (float*)malloc(npx * 3 * sizeof(float));
        orc_convert_8u_32f_scaled(bgr, npx * 3, img_src);
        orc_warp_linear_reflect_32f(img_src, rows, cols, 3, (float*)lap[0], crow, ccol, M);
        free(img_src);
    } else {
        int16_t* img_src = (int16_t*)malloc(npx * 3 * sizeof(int16_t));
        orc_convert_8u_16s(bgr, npx * 3, img_src);
        orc_warp_linear_reflect_16s(img_src, rows, cols, 3, (int16_t*)lap[0], crow, ccol, M);
        free(img_src);
    }
    orc_warp_nearest_const_32f(weight_src, rows, cols, 1, wp[0], crow, ccol, M);
    free(weight_src);

    /* 4. .cpp:468-474 */
    if (m->opt.force_float) orc_create_laplace_pyr_32f((float**)lap, crow, ccol, 3, L);
    else                    orc_create_laplace_pyr_16s((int16_t**)lap, crow, ccol, 3, L);
    for (int i = 0; i < L; i++)
        orc_pyr_down_32f(wp[i], crow >> i, ccol >> i, 1, wp[i + 1]);

    /* Apply, .cpp:476-555 (tiles are independent) */
#ifdef ORC_OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
#endif
    for (int x = xminInt; x < xmaxInt; x++)
        for (int y = yminInt; y < ymaxInt; y++) {
            orc_tile* ele = m->data[y * m->w + x];
            if (!ele) ele = m->data[y * m->w + x] = (orc_tile*)calloc(1, sizeof(orc_tile));
            int width = ORC_ELE_PIXELS, height = ORC_ELE_PIXELS;
            for (int i = 0; i <= L; i++) {
                const int pcols = ccol >> i;
                const size_t org = (size_t)(x - xminInt) * width + (size_t)(y - yminInt) * height * pcols;
                if (!ele->lap[i]) {
                    ele->lap[i] = malloc((size_t)width * height * 3 * es);
                    ele->w[i] = (float*)malloc((size_t)width * height * sizeof(float));
                    for (int r = 0; r < height; r++) {
                        memcpy((char*)ele->lap[i] + (size_t)r * width * 3 * es,
                               (char*)lap[i] + (org + (size_t)r * pcols) * 3 * es, (size_t)width * 3 * es);
                        memcpy(ele->w[i] + (size_t)r * width, wp[i] + org + (size_t)r * pcols, (size_t)width * sizeof(float));
                    }
                } else {
                    for (int r = 0; r < height; r++) {
                        const float* sW = wp[i] + org + (size_t)r * pcols;
                        float* dW = ele->w[i] + (size_t)r * width;
                        const char* sL = (const char*)lap[i] + (org + (size_t)r * pcols) * 3 * es;
                        char* dL = (char*)ele->lap[i] + (size_t)r * width * 3 * es;
                        for (int c = 0; c < width; c++)
                            if (sW[c] >= dW[c]) {
                                memcpy(dL + (size_t)c * 3 * es, sL + (size_t)c * 3 * es, 3 * es);
                                dW[c] = sW[c];
                            }
                    }
                }
                width /= 2; height /= 2;
            }
            ele->has_pyr = 1;
            ele->changed = 1;
        }

    m->last_x0 = xminInt + m->off_x; m->last_y0 = yminInt + m->off_y; m->last_tx = tx; m->last_ty = ty;
    memcpy(m->last_M, M, sizeof(M));
    if (m->keep_last) {
        free_last(m);
        for (int i = 0; i <= L; i++) { m->last_lap[i] = lap[i]; m->last_w[i] = wp[i]; }
    } else
        for (int i = 0; i <= L; i++) { free(lap[i]); free(wp[i]); }
    return 1;
}

/* feed, thread=false branch, .cpp:288-309 */
int orc_map_feed(orc_map* m, const uint8_t* bgr, int rows, int cols, const double pose_world[7])
{
    if (!m->valid) return 0;
    double pinv[7], pose[7];
    orc_se3_inverse(m->plane, pinv);
    orc_se3_mul(pinv, pose_world, pose);
    return render_frame(m, bgr, rows, cols, pose);
}

void orc_map_grid(orc_map* m, int dims[4], double geo[6])
{
    dims[0] = m->w; dims[1] = m->h; dims[2] = m->off_x; dims[3] = m->off_y;
    geo[0] = m->min[0]; geo[1] = m->min[1]; geo[2] = m->max[0]; geo[3] = m->max[1];
    geo[4] = m->ele_size; geo[5] = m->length_pixel;
}

int orc_map_num_levels(orc_map* m) { return m->opt.single_band ? 1 : m->band_num + 1; }

int orc_map_get_tile_bgra(orc_map* m, int ix, int iy, uint8_t* bgra)
{
    const int x = ix - m->off_x, y = iy - m->off_y;
    if (!m->opt.single_band || x < 0 || y < 0 || x >= m->w || y >= m->h) return 0;
    orc_tile* t = m->data[y * m->w + x];
    if (!t || !t->bgra) return 0;
    memcpy(bgra, t->bgra, (size_t)ORC_ELE_PIXELS * ORC_ELE_PIXELS * 4);
    return 1;
}

static orc_tile* tile_at(orc_map* m, int ix, int iy)
{
    const int x = ix - m->off_x, y = iy - m->off_y;
    if (x < 0 || y < 0 || x >= m->w || y >= m->h) return NULL;
    orc_tile* t = m->data[y * m->w + x];
    return (t && t->has_pyr) ? t : NULL;
}

int orc_map_tile_count(orc_map* m)
{
    int n = 0;
    for (int i = 0; i < m->w * m->h; i++) if (m->data[i] && m->data[i]->has_pyr) n++;
    return n;
}

int orc_map_tile_coords(orc_map* m, int* xy, int cap)
{
    int n = 0;
    for (int y = 0; y < m->h; y++)
        for (int x = 0; x < m->w; x++) {
            orc_tile* t = m->data[y * m->w + x];
            if (!t || !t->has_pyr) continue;
            if (n < cap) { xy[2 * n] = x + m->off_x; xy[2 * n + 1] = y + m->off_y; }
            n++;
        }
    return n;
}

int orc_map_get_tile_level(orc_map* m, int ix, int iy, int level, void* lap, float* w)
{
    orc_tile* t = tile_at(m, ix, iy);
    if (!t || level < 0 || level > m->band_num) return 0;
    const size_t n = (size_t)(ORC_ELE_PIXELS >> level) * (ORC_ELE_PIXELS >> level);
    if (lap) memcpy(lap, t->lap[level], n * 3 * elt_size(m));
    if (w) memcpy(w, t->w[level], n * sizeof(float));
    return 1;
}

static void restore_any(orc_map* m, void** lv, int rows, int cols, int n)
{
    if (m->opt.force_float) orc_restore_from_laplace_pyr_32f((float**)lv, rows, cols, 3, n);
    else                    orc_restore_from_laplace_pyr_16s((int16_t**)lv, rows, cols, 3, n);
}

/* Ele::blend, .cpp:77-146, with the neighbour gathering of draw(), .cpp:724-741 */
int orc_map_blend_tile_raw(orc_map* m, int ix, int iy, void* out)
{
    orc_tile* self = tile_at(m, ix, iy);
    if (!self) return 0;
    const int L = m->band_num, nl = L + 1;
    const size_t es = elt_size(m), px = 3 * es;
    orc_tile* nb[9]; int all = 1;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            orc_tile* t = m->opt.high_quality ? tile_at(m, ix + dx, iy + dy) : NULL;
            nb[3 * (dy + 1) + (dx + 1)] = t;
            if (!t) all = 0;
        }
    void* lv[ORC_MAX_LEVELS];
    if (all) {
        for (int i = 0; i < nl; i++) {
            const int border = 1 << (nl - i - 1), srows = ORC_ELE_PIXELS >> i, drows = srows + (border << 1);
            lv[i] = malloc((size_t)drows * drows * px);
            for (int y = 0; y < 3; y++)
                for (int x = 0; x < 3; x++) {
                    const orc_tile* e = nb[3 * y + x];
                    const int sw = (x == 1) ? srows : border, sh = (y == 1) ? srows : border;
                    const int sx = (x == 0) ? (srows - border) : 0, sy = (y == 0) ? (srows - border) : 0;
                    const int dx = (x == 0) ? 0 : ((x == 1) ? border : (drows - border));
                    const int dy = (y == 0) ? 0 : ((y == 1) ? border : (drows - border));
                    for (int r = 0; r < sh; r++)
                        memcpy((char*)lv[i] + ((size_t)(dy + r) * drows + dx) * px,
                               (const char*)e->lap[i] + ((size_t)(sy + r) * srows + sx) * px, (size_t)sw * px);
                }
        }
        const int b0 = 1 << (nl - 1), d0 = ORC_ELE_PIXELS + 2 * b0;
        restore_any(m, lv, d0, d0, L);
        for (int r = 0; r < ORC_ELE_PIXELS; r++)
            memcpy((char*)out + (size_t)r * ORC_ELE_PIXELS * px,
                   (char*)lv[0] + ((size_t)(b0 + r) * d0 + b0) * px, (size_t)ORC_ELE_PIXELS * px);
    } else {
        for (int i = 0; i < nl; i++) {
            const size_t n = (size_t)(ORC_ELE_PIXELS >> i) * (ORC_ELE_PIXELS >> i) * px;
            lv[i] = malloc(n); memcpy(lv[i], self->lap[i], n);
        }
        restore_any(m, lv, ORC_ELE_PIXELS, ORC_ELE_PIXELS, L);
        memcpy(out, lv[0], (size_t)ORC_ELE_PIXELS * ORC_ELE_PIXELS * px);
    }
    for (int i = 0; i < nl; i++) free(lv[i]);
    /* setTo(0, weights[0]==0) */
    const float* w0 = self->w[0];
    for (size_t j = 0; j < (size_t)ORC_ELE_PIXELS * ORC_ELE_PIXELS; j++)
        if (w0[j] == 0) memset((char*)out + j * px, 0, px);
    return 1;
}

/* updateTexture's pixel conversion, .cpp:154-160.  The reference uploads 32F
 * tiles as GL_FLOAT; for an 8-bit view of them this build defines
 * saturate(cvRound(v*255)) (the convertTo(CV_8UC3,255) rule).               */
int orc_map_blend_tile(orc_map* m, int ix, int iy, uint8_t* bgr)
{
    const size_t n = (size_t)ORC_ELE_PIXELS * ORC_ELE_PIXELS * 3;
    void* raw = malloc(n * elt_size(m));
    int ok = orc_map_blend_tile_raw(m, ix, iy, raw);
    if (ok) {
        if (m->opt.force_float) {
            const float* f = (const float*)raw;
            for (size_t j = 0; j < n; j++) bgr[j] = orc_sat_uchar(orc_cvround((double)(f[j] * 255.f)));
        } else orc_convert_16s_8u((const int16_t*)raw, n, bgr);
    }
    free(raw);
    return ok;
}

/* save, .cpp:779-847: bbox of tiles with pyramids */
int orc_map_save_size(orc_map* m, int* rows, int* cols, int* tile_x0, int* tile_y0)
{
    if (!m->valid || m->w == 0 || m->h == 0) return 0;
    int mnx = 1000000, mny = 1000000, mxx = -1000000, mxy = -1000000, cnt = 0;
    for (int x = 0; x < m->w; x++)
        for (int y = 0; y < m->h; y++) {
            orc_tile* e = m->data[x + y * m->w];
            if (!e || !e->has_pyr) continue;
            cnt++;
            mnx = mnx < x ? mnx : x; mny = mny < y ? mny : y;
            mxx = mxx > x ? mxx : x; mxy = mxy > y ? mxy : y;
        }
    if (!cnt) return 0;
    *cols = (mxx + 1 - mnx) * ORC_ELE_PIXELS; *rows = (mxy + 1 - mny) * ORC_ELE_PIXELS;
    *tile_x0 = mnx + m->off_x; *tile_y0 = mny + m->off_y;
    return cnt;
}

int orc_map_save(orc_map* m, uint8_t* bgr)
{
    int rows, cols, tx0, ty0;
    if (!orc_map_save_size(m, &rows, &cols, &tx0, &ty0)) return 0;
    const int L = m->band_num;
    const size_t es = elt_size(m), px = 3 * es;
    const int wx = cols / ORC_ELE_PIXELS, wy = rows / ORC_ELE_PIXELS;
    void* lv[ORC_MAX_LEVELS];
    for (int i = 0; i <= L; i++) lv[i] = calloc((size_t)(rows >> i) * (cols >> i), px);
    float* w0 = (float*)calloc((size_t)rows * cols, sizeof(float));
    for (int x = 0; x < wx; x++)
        for (int y = 0; y < wy; y++) {
            orc_tile* e = tile_at(m, tx0 + x, ty0 + y);
            if (!e) continue;
            int width = ORC_ELE_PIXELS;
            for (int i = 0; i <= L; i++) {
                const int pcols = cols >> i;
                for (int r = 0; r < width; r++) {
                    memcpy((char*)lv[i] + ((size_t)(y * width + r) * pcols + (size_t)x * width) * px,
                           (char*)e->lap[i] + (size_t)r * width * px, (size_t)width * px);
                    if (i == 0)
                        memcpy(w0 + (size_t)(y * width + r) * pcols + (size_t)x * width,
                               e->w[0] + (size_t)r * width, (size_t)width * sizeof(float));
                }
                width >>= 1;
            }
        }
    restore_any(m, lv, rows, cols, L);
    const size_t n = (size_t)rows * cols;
    if (m->opt.force_float) {
        /* reference leaves CV_32FC3 and hands it to imwrite (which would
         * saturate_cast the [0,1] floats to 0/1); this build writes v*255. */
        const float* f = (const float*)lv[0];
        for (size_t j = 0; j < n * 3; j++) bgr[j] = orc_sat_uchar(orc_cvround((double)(f[j] * 255.f)));
    } else orc_convert_16s_8u((const int16_t*)lv[0], n * 3, bgr);
    const uint8_t bg = orc_sat_uchar(m->opt.bg_color);
    for (size_t j = 0; j < n; j++)
        if (w0[j] == 0) { bgr[3 * j] = bg; bgr[3 * j + 1] = bg; bgr[3 * j + 2] = bg; }
    for (int i = 0; i <= L; i++) free(lv[i]);
    free(w0);
    return 1;
}

int orc_map_last_canvas(orc_map* m, int dims[4], double M[9])
{
    if (!m->last_tx) return 0;
    dims[0] = m->last_x0; dims[1] = m->last_y0; dims[2] = m->last_tx; dims[3] = m->last_ty;
    memcpy(M, m->last_M, sizeof(double) * 9);
    return 1;
}

void orc_map_keep_last(orc_map* m, int on) { m->keep_last = on; if (!on) free_last(m); }

int orc_map_last_level(orc_map* m, int level, void* lap, float* w)
{
    if (!m->keep_last || level < 0 || level > m->band_num || !m->last_lap[level]) return 0;
    const size_t n = level_pixels(m->last_tx, m->last_ty, level);
    if (lap) memcpy(lap, m->last_lap[level], n * 3 * elt_size(m));
    if (w) memcpy(w, m->last_w[level], n * sizeof(float));
    return 1;
}
