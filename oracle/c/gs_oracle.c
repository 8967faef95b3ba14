/*
 * gs_oracle.c -- scalar CPU restatement (plain C + OpenMP) of the rasterization hot path.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / reported baseline.  Never linked into the product.
 *
 * PARITY UNPINNED: the algorithm is gsplat 1.0.0's (un-vendored dependency of the reference,
 * requirements.txt:1, README.md:16); restated from its published behaviour as summarised in
 * SURVEY.md Appendix A and anchored on the reference call site
 * /root/reference/model/gaussian.py:353-372 and :188-197.  It is cross-checked against the
 * differentiable PyTorch restatement (oracle/torch_oracle.py) in tests/test_oracle.py.
 *
 * Stages (one function each, same order the reference's single call executes them):
 *   gso_project_fwd   A.1   world->camera, covariance, perspective Jacobian, conic, radius, cull
 *   gso_sh_fwd        A.2   view-dependent colour, +0.5, clamp at 0
 *   gso_isect_count / gso_isect_build   A.3   tile lists, (cam,tile,depth,index) order, offsets
 *   gso_blend_fwd     A.4   per-pixel front-to-back alpha blend with early-out
 *   gso_blend_bwd     A.5   per-pixel back-to-front replay, incl. absgrad accumulation
 *   gso_sh_bwd, gso_project_bwd   A.6
 *
 * Build: see oracle/c/Makefile (real = float by default; -DGSO_DOUBLE for an fp64 shadow).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef GSO_DOUBLE
typedef double real;
#define RSQRT(x) sqrt(x)
#define REXP(x) exp(x)
#define RLOG(x) log(x)
#define RCEIL(x) ceil(x)
#define RFLOOR(x) floor(x)
#define RABS(x) fabs(x)
#else
typedef float real;
#define RSQRT(x) sqrtf(x)
#define REXP(x) expf(x)
#define RLOG(x) logf(x)
#define RCEIL(x) ceilf(x)
#define RFLOOR(x) floorf(x)
#define RABS(x) fabsf(x)
#endif

#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.999)
#define T_MIN ((real)1e-4)
/* (as float32 values, like every scalar of the path: gsplat writes 1.3f and 0.01f, and the device chain widens the same floats --
 *  0.01 as a double differs from 0.01f by 2e-10, enough to move a ceil(3 sqrt(lambda)) once in ~1e9 near-isotropic Gaussians) */
#define FOV_CLAMP ((real)1.3f)
#define RADIUS_DISC_FLOOR ((real)0.01f)

static const real SH_C0 = (real)0.2820947917738781;
static const real SH_C1 = (real)0.4886025119029199;
static const real SH_C20 = (real)1.0925484305920792, SH_C21 = (real)0.31539156525252005,
                  SH_C22 = (real)0.5462742152960396;
static const real SH_C30 = (real)0.5900435899266435, SH_C31 = (real)2.890611442640554,
                  SH_C32 = (real)0.4570457994644658, SH_C33 = (real)0.3731763325901154,
                  SH_C34 = (real)1.445305721320277;

int gso_real_bytes(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ helpers */
static void quat_to_R(const real* q, real* R, real* qn_out, real* norm_out) {
    real n = RSQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    real w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
    if (qn_out) { qn_out[0] = w; qn_out[1] = x; qn_out[2] = y; qn_out[3] = z; }
    if (norm_out) *norm_out = n;
}

static void mat3_mul(const real* A, const real* B, real* C) { /* C = A B */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}
static void mat3_mul_T(const real* A, const real* B, real* C) { /* C = A B^T */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[j * 3 + k];
            C[i * 3 + j] = s;
        }
}
static void mat3_Tmul(const real* A, const real* B, real* C) { /* C = A^T B */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

/* Shared forward chain up to the 2-D covariance; returns 0 if depth-culled. */
typedef struct {
    real Rq[9], qn[4], qnorm, M[9], cov[9], Rv[9], pc[3], covc[9];
    real fx, fy, cx, cy, tx, ty, J[6], cov2[3]; /* cov2 = (a,b,c) AFTER blur */
    int clamp_x, clamp_y; /* -1,0,+1 */
    real limx, limy, det;
} proj_t;

static int proj_chain(const real* mean, const real* quat, const real* scale, const real* V,
                      const real* K, int W, int H, real eps2d, real near_p, real far_p,
                      proj_t* o) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o->Rv[i * 3 + j] = V[i * 4 + j];
    for (int i = 0; i < 3; i++)
        o->pc[i] = o->Rv[i * 3] * mean[0] + o->Rv[i * 3 + 1] * mean[1] + o->Rv[i * 3 + 2] * mean[2] + V[i * 4 + 3];
    real z = o->pc[2];
    if (z < near_p || z > far_p) return 0;
    quat_to_R(quat, o->Rq, o->qn, &o->qnorm);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o->M[i * 3 + j] = o->Rq[i * 3 + j] * scale[j];
    mat3_mul_T(o->M, o->M, o->cov);
    real tmp[9];
    mat3_mul(o->Rv, o->cov, tmp);
    mat3_mul_T(tmp, o->Rv, o->covc);
    o->fx = K[0]; o->fy = K[4]; o->cx = K[2]; o->cy = K[5];
    o->limx = FOV_CLAMP * ((real)0.5 * W / o->fx);
    o->limy = FOV_CLAMP * ((real)0.5 * H / o->fy);
    real rx = o->pc[0] / z, ry = o->pc[1] / z;
    o->clamp_x = rx > o->limx ? 1 : (rx < -o->limx ? -1 : 0);
    o->clamp_y = ry > o->limy ? 1 : (ry < -o->limy ? -1 : 0);
    real crx = o->clamp_x > 0 ? o->limx : (o->clamp_x < 0 ? -o->limx : rx);
    real cry = o->clamp_y > 0 ? o->limy : (o->clamp_y < 0 ? -o->limy : ry);
    o->tx = z * crx; o->ty = z * cry;
    real* J = o->J;
    J[0] = o->fx / z; J[1] = 0; J[2] = -o->fx * o->tx / (z * z);
    J[3] = 0; J[4] = o->fy / z; J[5] = -o->fy * o->ty / (z * z);
    /* cov2 = J covc J^T */
    real JC[6];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            real s = 0;
            for (int k = 0; k < 3; k++) s += J[i * 3 + k] * o->covc[k * 3 + j];
            JC[i * 3 + j] = s;
        }
    real s00 = 0, s01 = 0, s11 = 0;
    for (int k = 0; k < 3; k++) {
        s00 += JC[k] * J[k];
        s01 += JC[k] * J[3 + k];
        s11 += JC[3 + k] * J[3 + k];
    }
    o->cov2[0] = s00 + eps2d; o->cov2[1] = s01; o->cov2[2] = s11 + eps2d;
    o->det = o->cov2[0] * o->cov2[2] - o->cov2[1] * o->cov2[1];
    return 1;
}

/* ------------------------------------------------------------------ A.1 */
int gso_project_fwd(int C, int N, const real* means, const real* quats, const real* scales,
                    const real* viewmats, const real* Ks, int W, int H, real eps2d, real near_p,
                    real far_p, real radius_clip, int32_t* radii, real* means2d, real* depths,
                    real* conics) {
#pragma omp parallel for schedule(static)
    for (long f = 0; f < (long)C * N; f++) {
        int c = (int)(f / N), n = (int)(f % N);
        radii[f] = 0; means2d[2 * f] = means2d[2 * f + 1] = 0; depths[f] = 0;
        conics[3 * f] = conics[3 * f + 1] = conics[3 * f + 2] = 0;
        proj_t p;
        if (!proj_chain(means + 3 * n, quats + 4 * n, scales + 3 * n, viewmats + 16 * c,
                        Ks + 9 * c, W, H, eps2d, near_p, far_p, &p)) continue;
        if (!(p.det > 0)) continue;
        real a = p.cov2[0], b = p.cov2[1], cc = p.cov2[2];
        real mid = (real)0.5 * (a + cc);
        real disc = mid * mid - p.det;
        if (disc < RADIUS_DISC_FLOOR) disc = RADIUS_DISC_FLOOR;
        real lam = mid + RSQRT(disc);
        real radius = RCEIL((real)3.0 * RSQRT(lam));
        if (radius <= radius_clip) continue;
        real z = p.pc[2];
        real mx = p.fx * p.pc[0] / z + p.cx, my = p.fy * p.pc[1] / z + p.cy;
        if (mx + radius <= 0 || mx - radius >= W || my + radius <= 0 || my - radius >= H) continue;
        radii[f] = (int32_t)radius;
        means2d[2 * f] = mx; means2d[2 * f + 1] = my; depths[f] = z;
        conics[3 * f] = cc / p.det; conics[3 * f + 1] = -b / p.det; conics[3 * f + 2] = a / p.det;
    }
    return 0;
}

/* ------------------------------------------------------------------ A.2 */
static void cam_position(const real* V, real* pos) { /* -R^T t (rigid world->camera) is NOT assumed:
    general inverse of the 3x3 block, as torch.inverse(viewmats)[:3,3] would give */
    real a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[6], g = V[8], h = V[9], i = V[10];
    real det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    real inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                   (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                   (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
    real t[3] = {V[3], V[7], V[11]};
    for (int r = 0; r < 3; r++) pos[r] = -(inv[r * 3] * t[0] + inv[r * 3 + 1] * t[1] + inv[r * 3 + 2] * t[2]);
}

static void sh_basis(int degree, real x, real y, real z, real* Y) {
    Y[0] = SH_C0;
    if (degree < 1) return;
    Y[1] = -SH_C1 * y; Y[2] = SH_C1 * z; Y[3] = -SH_C1 * x;
    if (degree < 2) return;
    real xx = x * x, yy = y * y, zz = z * z;
    Y[4] = SH_C20 * x * y; Y[5] = -SH_C20 * y * z; Y[6] = SH_C21 * (2 * zz - xx - yy);
    Y[7] = -SH_C20 * x * z; Y[8] = SH_C22 * (xx - yy);
    if (degree < 3) return;
    Y[9] = -SH_C30 * y * (3 * xx - yy); Y[10] = SH_C31 * x * y * z;
    Y[11] = -SH_C32 * y * (4 * zz - xx - yy); Y[12] = SH_C33 * z * (2 * zz - 3 * xx - 3 * yy);
    Y[13] = -SH_C32 * x * (4 * zz - xx - yy); Y[14] = SH_C34 * z * (xx - yy);
    Y[15] = -SH_C30 * x * (xx - 3 * yy);
}

/* dY[k][3] = gradient of Y_k wrt (x,y,z) as independent variables */
static void sh_basis_grad(int degree, real x, real y, real z, real (*dY)[3]) {
    dY[0][0] = dY[0][1] = dY[0][2] = 0;
    if (degree < 1) return;
    dY[1][0] = 0; dY[1][1] = -SH_C1; dY[1][2] = 0;
    dY[2][0] = 0; dY[2][1] = 0; dY[2][2] = SH_C1;
    dY[3][0] = -SH_C1; dY[3][1] = 0; dY[3][2] = 0;
    if (degree < 2) return;
    dY[4][0] = SH_C20 * y; dY[4][1] = SH_C20 * x; dY[4][2] = 0;
    dY[5][0] = 0; dY[5][1] = -SH_C20 * z; dY[5][2] = -SH_C20 * y;
    dY[6][0] = -2 * SH_C21 * x; dY[6][1] = -2 * SH_C21 * y; dY[6][2] = 4 * SH_C21 * z;
    dY[7][0] = -SH_C20 * z; dY[7][1] = 0; dY[7][2] = -SH_C20 * x;
    dY[8][0] = 2 * SH_C22 * x; dY[8][1] = -2 * SH_C22 * y; dY[8][2] = 0;
    if (degree < 3) return;
    real xx = x * x, yy = y * y, zz = z * z;
    dY[9][0] = -6 * SH_C30 * x * y; dY[9][1] = -SH_C30 * (3 * xx - 3 * yy); dY[9][2] = 0;
    dY[10][0] = SH_C31 * y * z; dY[10][1] = SH_C31 * x * z; dY[10][2] = SH_C31 * x * y;
    dY[11][0] = 2 * SH_C32 * x * y; dY[11][1] = -SH_C32 * (4 * zz - xx - 3 * yy); dY[11][2] = -8 * SH_C32 * y * z;
    dY[12][0] = -6 * SH_C33 * x * z; dY[12][1] = -6 * SH_C33 * y * z; dY[12][2] = SH_C33 * (6 * zz - 3 * xx - 3 * yy);
    dY[13][0] = -SH_C32 * (4 * zz - 3 * xx - yy); dY[13][1] = 2 * SH_C32 * x * y; dY[13][2] = -8 * SH_C32 * x * z;
    dY[14][0] = 2 * SH_C34 * x * z; dY[14][1] = -2 * SH_C34 * y * z; dY[14][2] = SH_C34 * (xx - yy);
    dY[15][0] = -SH_C30 * (3 * xx - 3 * yy); dY[15][1] = 6 * SH_C30 * x * y; dY[15][2] = 0;
}

int gso_sh_fwd(int C, int N, int K, int degree, const real* means, const real* viewmats,
               const real* shs, const int32_t* radii, real* colors) {
    int Ka = (degree + 1) * (degree + 1);
    if (Ka > K || degree > 3) return -1;
#pragma omp parallel for schedule(static)
    for (long f = 0; f < (long)C * N; f++) {
        int c = (int)(f / N), n = (int)(f % N);
        real rgb[3] = {0, 0, 0};
        if (radii[f] > 0) {
            real pos[3];
            cam_position(viewmats + 16 * c, pos);
            real d[3] = {means[3 * n] - pos[0], means[3 * n + 1] - pos[1], means[3 * n + 2] - pos[2]};
            real nn = RSQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            if (nn > 0) { d[0] /= nn; d[1] /= nn; d[2] /= nn; }
            real Y[16];
            sh_basis(degree, d[0], d[1], d[2], Y);
            for (int k = 0; k < Ka; k++)
                for (int ch = 0; ch < 3; ch++) rgb[ch] += Y[k] * shs[((long)n * K + k) * 3 + ch];
        }
        for (int ch = 0; ch < 3; ch++) {
            real v = rgb[ch] + (real)0.5;
            colors[3 * f + ch] = v > 0 ? v : 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A.3 */
static void tile_bbox(real mx, real my, int32_t radius, int tile, int tw, int th, int* x0, int* x1,
                      int* y0, int* y1) {
    real r = (real)radius / tile, tx = mx / tile, ty = my / tile;
    real fx0 = RFLOOR(tx - r), fx1 = RCEIL(tx + r), fy0 = RFLOOR(ty - r), fy1 = RCEIL(ty + r);
    *x0 = (int)(fx0 < 0 ? 0 : (fx0 > tw ? tw : fx0));
    *x1 = (int)(fx1 < 0 ? 0 : (fx1 > tw ? tw : fx1));
    *y0 = (int)(fy0 < 0 ? 0 : (fy0 > th ? th : fy0));
    *y1 = (int)(fy1 < 0 ? 0 : (fy1 > th ? th : fy1));
}

int64_t gso_isect_count(int C, int N, const real* means2d, const int32_t* radii, int tile, int tw,
                        int th, int32_t* tiles_per_gauss) {
    int64_t total = 0;
    for (long f = 0; f < (long)C * N; f++) {
        int cnt = 0;
        if (radii[f] > 0) {
            int x0, x1, y0, y1;
            tile_bbox(means2d[2 * f], means2d[2 * f + 1], radii[f], tile, tw, th, &x0, &x1, &y0, &y1);
            cnt = (x1 - x0) * (y1 - y0);
        }
        tiles_per_gauss[f] = cnt;
        total += cnt;
    }
    return total;
}

typedef struct { int64_t key; int32_t val; } kv_t;
static int kv_cmp(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->val < y->val ? -1 : (x->val > y->val ? 1 : 0); /* stable: ties by flatten index */
}

int gso_isect_build(int C, int N, const real* means2d, const int32_t* radii, const real* depths,
                    int tile, int tw, int th, int64_t I, int64_t* isect_ids, int32_t* flatten_ids,
                    int32_t* isect_offsets) {
    int n_tiles = tw * th;
    int tile_bits = 0;
    { int v = n_tiles; while (v > 0) { tile_bits++; v >>= 1; } if (tile_bits == 0) tile_bits = 1; }
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(I > 0 ? I : 1));
    int64_t pos = 0;
    for (long f = 0; f < (long)C * N; f++) {
        if (radii[f] <= 0) continue;
        int c = (int)(f / N);
        int x0, x1, y0, y1;
        tile_bbox(means2d[2 * f], means2d[2 * f + 1], radii[f], tile, tw, th, &x0, &x1, &y0, &y1);
        float df = (float)depths[f];
        int32_t dbits; memcpy(&dbits, &df, 4);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                int64_t tid = (int64_t)y * tw + x;
                kv[pos].key = ((int64_t)c << (32 + tile_bits)) | (tid << 32) | (int64_t)(uint32_t)dbits;
                kv[pos].val = (int32_t)f;
                pos++;
            }
    }
    if (pos != I) { free(kv); return -1; }
    qsort(kv, (size_t)I, sizeof(kv_t), kv_cmp);
    for (long t = 0; t < (long)C * n_tiles; t++) isect_offsets[t] = 0;
    int64_t* cnt = (int64_t*)calloc((size_t)C * n_tiles + 1, sizeof(int64_t));
    for (int64_t i = 0; i < I; i++) {
        isect_ids[i] = kv[i].key; flatten_ids[i] = kv[i].val;
        int64_t cam = kv[i].key >> (32 + tile_bits);
        int64_t tid = (kv[i].key >> 32) & (((int64_t)1 << tile_bits) - 1);
        cnt[cam * n_tiles + tid]++;
    }
    int64_t run = 0;
    for (long t = 0; t < (long)C * n_tiles; t++) { isect_offsets[t] = (int32_t)run; run += cnt[t]; }
    free(cnt); free(kv);
    return 0;
}

/* ------------------------------------------------------------------ A.4 */
int gso_blend_fwd(int C, int N, int W, int H, int tile, const real* means2d, const real* conics,
                  const real* colors, const real* opac, const real* bg /* [C,3] or NULL */,
                  const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t I,
                  real* out_colors, real* out_alphas, int32_t* last_ids) {
    int tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    (void)N;
#pragma omp parallel for schedule(dynamic, 64) collapse(2)
    for (int c = 0; c < C; c++)
        for (long pix = 0; pix < (long)H * W; pix++) {
            int i = (int)(pix / W), j = (int)(pix % W);
            long t = ((long)c * th + i / tile) * tw + j / tile;
            int64_t lo = isect_offsets[t];
            int64_t hi = (t + 1 < (long)C * tw * th) ? isect_offsets[t + 1] : I;
            real px = j + (real)0.5, py = i + (real)0.5;
            real T = 1, acc[3] = {0, 0, 0};
            int32_t last = 0;
            for (int64_t k = lo; k < hi; k++) {
                int32_t g = flatten_ids[k];
                real dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                real sigma = (real)0.5 * (conics[3 * g] * dx * dx + conics[3 * g + 2] * dy * dy) + conics[3 * g + 1] * dx * dy;
                real alpha = opac[g] * REXP(-sigma);
                if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
                if (sigma < 0 || alpha < ALPHA_MIN) continue;
                real Tn = T * (1 - alpha);
                if (Tn <= T_MIN) break;
                real w = alpha * T;
                acc[0] += colors[3 * g] * w; acc[1] += colors[3 * g + 1] * w; acc[2] += colors[3 * g + 2] * w;
                last = (int32_t)k;
                T = Tn;
            }
            long o = (long)c * H * W + pix;
            for (int ch = 0; ch < 3; ch++) out_colors[3 * o + ch] = acc[ch] + (bg ? T * bg[3 * c + ch] : 0);
            out_alphas[o] = 1 - T;
            last_ids[o] = last;
        }
    return 0;
}

/* Per-pixel distance to the blend's discontinuities (alpha >= 1/255 test, T <= 1e-4 early-out,
 * sigma >= 0 test): min over the pairs the forward visits of the gap to the threshold, RELATIVE to the
 * threshold and NORMALISED by how far fp32 inputs can move the tested quantity, so that "margin < 1e-4"
 * reads "an fp32 evaluation of this pixel may legitimately take the other branch":
 *   - the path under test (like gsplat itself) holds means2d in fp32: a few ulps of a ~1000 px coordinate
 *     (view transform + perspective divide, each rounded) are dmu ~ 3e-4 px, which moves
 *     ln(alpha) = ln(op) - sigma by |d sigma / d mu| * dmu -- 1e-3 and more for sharp splats;
 *   - the conic's own rounding moves sigma by ~1e-5 of the magnitude of its terms;
 *   - T inherits the accumulated relative error of every alpha blended before.
 * margin = gap * 1e-4 / (1e-4 + reachable relative perturbation).  Parity tests exempt pixels with
 * margin < 1e-4 from the strict bound (they must stay rare and are still bounded by one contributor's weight).
 * means2d_o / conics_o (optional): the OTHER implementation's means2d / conics.  When given, the reachable
 * perturbation of sigma is measured, not bounded: twice |sigma(other's mean, other's conic) - sigma| at this very
 * pixel, plus the rounding of an fp32 evaluation of the form (8 ulps of the magnitude of its terms). */
#define GSO_EPS32 ((real)1.1920929e-7)
int gso_blend_margin_tol(int C, int N, int W, int H, int tile, const real* means2d, const real* conics,
                         const real* opac, const int32_t* isect_offsets, const int32_t* flatten_ids,
                         int64_t I, const real* means2d_o, const real* conics_o, real mu_tol_ulps, real conic_rtol,
                         real* margin);
int gso_blend_margin(int C, int N, int W, int H, int tile, const real* means2d, const real* conics,
                     const real* opac, const int32_t* isect_offsets, const int32_t* flatten_ids,
                     int64_t I, const real* means2d_o, const real* conics_o, real* margin) {
    return gso_blend_margin_tol(C, N, W, H, tile, means2d, conics, opac, isect_offsets, flatten_ids, I, means2d_o, conics_o,
                                (real)0, (real)0, margin);
}

/* The a-priori form with the storage bounds as PARAMETERS: mu_tol_ulps > 0 -- the tested path's means2d are within that many
 * fp32 ulps (of max(|coordinate|, 32 px)) of this run's, its conics within conic_rtol of their largest entry (both are
 * asserted by the parity tests on the path's outputs, so the exemption follows from bounds the implementation is held to and
 * not from its measured error); <= 0: the legacy model (4 ulps of max(|coordinate|, 64), 1e-5 of the form's terms). */
int gso_blend_margin_tol(int C, int N, int W, int H, int tile, const real* means2d, const real* conics,
                         const real* opac, const int32_t* isect_offsets, const int32_t* flatten_ids,
                         int64_t I, const real* means2d_o, const real* conics_o, real mu_tol_ulps, real conic_rtol,
                         real* margin) {
    int tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    const real base = (real)1e-4;
    (void)N;
#pragma omp parallel for schedule(dynamic, 64) collapse(2)
    for (int c = 0; c < C; c++)
        for (long pix = 0; pix < (long)H * W; pix++) {
            int i = (int)(pix / W), j = (int)(pix % W);
            long t = ((long)c * th + i / tile) * tw + j / tile;
            int64_t lo = isect_offsets[t];
            int64_t hi = (t + 1 < (long)C * tw * th) ? isect_offsets[t + 1] : I;
            real px = j + (real)0.5, py = i + (real)0.5;
            real T = 1, m = 1, eT = 0;
            for (int64_t k = lo; k < hi; k++) {
                int32_t g = flatten_ids[k];
                real mux = means2d[2 * g], muy = means2d[2 * g + 1];
                real A = conics[3 * g], B = conics[3 * g + 1], Cc = conics[3 * g + 2];
                real dx = mux - px, dy = muy - py;
                real sigma = (real)0.5 * (A * dx * dx + Cc * dy * dy) + B * dx * dy;
                real alpha = opac[g] * REXP(-sigma);
                if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
                real mag = (real)0.5 * (RABS(A * dx * dx) + RABS(Cc * dy * dy)) + RABS(B * dx * dy);
                real amax = RABS(mux) > RABS(muy) ? RABS(mux) : RABS(muy);
                real es;   /* reachable |d sigma| */
                if (means2d_o && conics_o) {
                    real dxo = means2d_o[2 * g] - px, dyo = means2d_o[2 * g + 1] - py;
                    real so = (real)0.5 * (conics_o[3 * g] * dxo * dxo + conics_o[3 * g + 2] * dyo * dyo) + conics_o[3 * g + 1] * dxo * dyo;
                    es = 2 * RABS(so - sigma) + 8 * GSO_EPS32 * mag;
                } else if (mu_tol_ulps > 0) {   /* a-priori from the bounds the tested path is held to + fp32 evaluation of the form */
                    real dmu = mu_tol_ulps * GSO_EPS32 * (amax > 32 ? amax : 32);
                    real cmax = RABS(A) > RABS(Cc) ? RABS(A) : RABS(Cc);
                    if (RABS(B) > cmax) cmax = RABS(B);
                    real dcon = conic_rtol * cmax * ((real)0.5 * (dx * dx + dy * dy) + RABS(dx * dy));
                    es = (RABS(A * dx + B * dy) + RABS(B * dx + Cc * dy)) * dmu + dcon + 8 * GSO_EPS32 * mag;
                } else {   /* legacy a-priori: mean moved by a few ulps of its coordinate, conic by 1e-5, fp32 evaluation */
                    real dmu = 4 * GSO_EPS32 * (amax > 64 ? amax : 64);
                    es = (RABS(A * dx + B * dy) + RABS(B * dx + Cc * dy)) * dmu + ((real)1e-5 + 4 * GSO_EPS32) * mag;
                }
                /* gap of ln(alpha) to ln(1/255) -- in the log domain a perturbation of sigma is additive, also far from
                 * the threshold (|sigma| ~ 1e5 at pixels a sharp splat can never reach) */
                real ma = RABS(RLOG(opac[g] > 0 ? opac[g] : (real)1e-30) - sigma - RLOG(ALPHA_MIN));
                if (ma > 1) ma = 1;
                ma = ma * base / (base + es);
                if (ma < m) m = ma;
                /* sigma >= 0: the form is positive (semi-)definite, so only rounding inside its own evaluation can
                 * make it negative -- the gap is measured relative to the magnitude of its three terms */
                if (opac[g] >= ALPHA_MIN) {
                    real ms = mag > 0 ? RABS(sigma) / mag : 1;
                    if (ms < m) m = ms;
                }
                if (sigma < 0 || alpha < ALPHA_MIN) continue;
                real Tn = T * (1 - alpha);
                eT += alpha / (1 - alpha) * es;
                real mt = RABS(RLOG(Tn > 0 ? Tn : (real)1e-30) - RLOG(T_MIN));
                if (mt > 1) mt = 1;
                mt = mt * base / (base + eT);
                if (mt < m) m = mt;
                if (Tn <= T_MIN) break;
                T = Tn;
            }
            margin[(long)c * H * W + pix] = m;
        }
    return 0;
}

/* ------------------------------------------------------------------ A.5 */
static inline void atomic_add(real* p, real v) {
#pragma omp atomic
    *p += v;
}

int gso_blend_bwd(int C, int N, int W, int H, int tile, const real* means2d, const real* conics,
                  const real* colors, const real* opac, const real* bg,
                  const int32_t* isect_offsets, const int32_t* flatten_ids, int64_t I,
                  const real* out_alphas, const int32_t* last_ids, const real* v_colors,
                  const real* v_alphas, real* v_means2d, real* v_means2d_abs, real* v_conics,
                  real* v_rgb, real* v_opac) {
    int tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    long CN = (long)C * N;
    memset(v_means2d, 0, sizeof(real) * 2 * CN); memset(v_means2d_abs, 0, sizeof(real) * 2 * CN);
    memset(v_conics, 0, sizeof(real) * 3 * CN); memset(v_rgb, 0, sizeof(real) * 3 * CN);
    memset(v_opac, 0, sizeof(real) * CN);
#pragma omp parallel for schedule(dynamic, 64) collapse(2)
    for (int c = 0; c < C; c++)
        for (long pix = 0; pix < (long)H * W; pix++) {
            int i = (int)(pix / W), j = (int)(pix % W);
            long t = ((long)c * th + i / tile) * tw + j / tile;
            int64_t lo = isect_offsets[t];
            int64_t hi = (t + 1 < (long)C * tw * th) ? isect_offsets[t + 1] : I;
            if (hi <= lo) continue;
            long o = (long)c * H * W + pix;
            real px = j + (real)0.5, py = i + (real)0.5;
            real Tf = 1 - out_alphas[o];
            real T = Tf, buf[3] = {0, 0, 0};
            const real* vc = v_colors + 3 * o;
            real va = v_alphas ? v_alphas[o] : 0;
            real bgdot = bg ? (bg[3 * c] * vc[0] + bg[3 * c + 1] * vc[1] + bg[3 * c + 2] * vc[2]) : 0;
            for (int64_t k = last_ids[o]; k >= lo; k--) {
                int32_t g = flatten_ids[k];
                real dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                real A = conics[3 * g], B = conics[3 * g + 1], Cc = conics[3 * g + 2];
                real sigma = (real)0.5 * (A * dx * dx + Cc * dy * dy) + B * dx * dy;
                real vis = REXP(-sigma);
                real alpha = opac[g] * vis;
                if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
                if (sigma < 0 || alpha < ALPHA_MIN) continue;
                real ra = 1 / (1 - alpha);
                T *= ra;
                real fac = alpha * T;
                real v_alpha = 0;
                for (int ch = 0; ch < 3; ch++) {
                    atomic_add(&v_rgb[3 * g + ch], fac * vc[ch]);
                    v_alpha += (colors[3 * g + ch] * T - buf[ch] * ra) * vc[ch];
                }
                v_alpha += Tf * ra * va;
                if (bg) v_alpha += -Tf * ra * bgdot;
                if (opac[g] * vis <= ALPHA_MAX) {
                    real v_sigma = -opac[g] * vis * v_alpha;
                    atomic_add(&v_conics[3 * g], (real)0.5 * v_sigma * dx * dx);
                    atomic_add(&v_conics[3 * g + 1], v_sigma * dx * dy);
                    atomic_add(&v_conics[3 * g + 2], (real)0.5 * v_sigma * dy * dy);
                    real gx = v_sigma * (A * dx + B * dy), gy = v_sigma * (B * dx + Cc * dy);
                    atomic_add(&v_means2d[2 * g], gx); atomic_add(&v_means2d[2 * g + 1], gy);
                    atomic_add(&v_means2d_abs[2 * g], RABS(gx)); atomic_add(&v_means2d_abs[2 * g + 1], RABS(gy));
                    atomic_add(&v_opac[g], vis * v_alpha);
                }
                for (int ch = 0; ch < 3; ch++) buf[ch] += colors[3 * g + ch] * fac;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------ A.6 colour path */
int gso_sh_bwd(int C, int N, int K, int degree, const real* means, const real* viewmats,
               const real* shs, const int32_t* radii, const real* colors /* post-activation */,
               const real* v_colors, real* v_shs /* [N,K,3] */, real* v_means /* [N,3] += */) {
    int Ka = (degree + 1) * (degree + 1);
    if (Ka > K || degree > 3) return -1;
    memset(v_shs, 0, sizeof(real) * (size_t)N * K * 3);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        for (int c = 0; c < C; c++) {
            long f = (long)c * N + n;
            if (radii[f] <= 0) continue;
            real vp[3];
            for (int ch = 0; ch < 3; ch++) vp[ch] = colors[3 * f + ch] > 0 ? v_colors[3 * f + ch] : 0;
            real pos[3];
            cam_position(viewmats + 16 * c, pos);
            real d[3] = {means[3 * n] - pos[0], means[3 * n + 1] - pos[1], means[3 * n + 2] - pos[2]};
            real nn = RSQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            real u[3] = {d[0], d[1], d[2]};
            if (nn > 0) { u[0] /= nn; u[1] /= nn; u[2] /= nn; }
            real Y[16]; real dY[16][3];
            sh_basis(degree, u[0], u[1], u[2], Y);
            sh_basis_grad(degree, u[0], u[1], u[2], dY);
            real vu[3] = {0, 0, 0};
            for (int k = 0; k < Ka; k++) {
                real dot = 0;
                for (int ch = 0; ch < 3; ch++) {
                    v_shs[((long)n * K + k) * 3 + ch] += Y[k] * vp[ch];
                    dot += shs[((long)n * K + k) * 3 + ch] * vp[ch];
                }
                vu[0] += dY[k][0] * dot; vu[1] += dY[k][1] * dot; vu[2] += dY[k][2] * dot;
            }
            if (nn > 0) {
                real ud = u[0] * vu[0] + u[1] * vu[1] + u[2] * vu[2];
                for (int r = 0; r < 3; r++) v_means[3 * n + r] += (vu[r] - u[r] * ud) / nn;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A.6 projection VJP */
int gso_project_bwd(int C, int N, const real* means, const real* quats, const real* scales,
                    const real* viewmats, const real* Ks, int W, int H, real eps2d, real near_p,
                    real far_p, const int32_t* radii, const real* v_means2d, const real* v_depths,
                    const real* v_conics, real* v_means /* [N,3] += */, real* v_quats /* [N,4] = */,
                    real* v_scales /* [N,3] = */) {
    memset(v_quats, 0, sizeof(real) * 4 * (size_t)N);
    memset(v_scales, 0, sizeof(real) * 3 * (size_t)N);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        for (int c = 0; c < C; c++) {
            long f = (long)c * N + n;
            if (radii[f] <= 0) continue;
            proj_t p;
            if (!proj_chain(means + 3 * n, quats + 4 * n, scales + 3 * n, viewmats + 16 * c,
                            Ks + 9 * c, W, H, eps2d, near_p, far_p, &p)) continue;
            real a = p.cov2[0], b = p.cov2[1], cc = p.cov2[2], det = p.det;
            real X[4] = {cc / det, -b / det, -b / det, a / det};
            real vA = v_conics[3 * f], vB = v_conics[3 * f + 1], vC = v_conics[3 * f + 2];
            real Vm[4] = {vA, (real)0.5 * vB, (real)0.5 * vB, vC};
            /* G = -X V X */
            real XV[4] = {X[0] * Vm[0] + X[1] * Vm[2], X[0] * Vm[1] + X[1] * Vm[3],
                          X[2] * Vm[0] + X[3] * Vm[2], X[2] * Vm[1] + X[3] * Vm[3]};
            real G[4] = {-(XV[0] * X[0] + XV[1] * X[2]), -(XV[0] * X[1] + XV[1] * X[3]),
                         -(XV[2] * X[0] + XV[3] * X[2]), -(XV[2] * X[1] + XV[3] * X[3])};
            const real* J = p.J;
            /* v_covc = J^T G J (3x3) ; v_J = 2 G J covc (2x3) */
            real GJ[6];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) GJ[i * 3 + j] = G[i * 2] * J[j] + G[i * 2 + 1] * J[3 + j];
            real v_covc[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) v_covc[i * 3 + j] = J[i] * GJ[j] + J[3 + i] * GJ[3 + j];
            real vJ[6];
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) {
                    real s = 0;
                    for (int k = 0; k < 3; k++) s += GJ[i * 3 + k] * p.covc[k * 3 + j];
                    vJ[i * 3 + j] = 2 * s;
                }
            real x = p.pc[0], y = p.pc[1], z = p.pc[2];
            real z2 = z * z, z3 = z2 * z;
            real vpc[3] = {0, 0, 0};
            vpc[2] += vJ[0] * (-p.fx / z2) + vJ[4] * (-p.fy / z2) + vJ[2] * (2 * p.fx * p.tx / z3) + vJ[5] * (2 * p.fy * p.ty / z3);
            real v_tx = vJ[2] * (-p.fx / z2), v_ty = vJ[5] * (-p.fy / z2);
            if (p.clamp_x == 0) vpc[0] += v_tx; else vpc[2] += v_tx * (p.clamp_x > 0 ? p.limx : -p.limx);
            if (p.clamp_y == 0) vpc[1] += v_ty; else vpc[2] += v_ty * (p.clamp_y > 0 ? p.limy : -p.limy);
            real vmx = v_means2d[2 * f], vmy = v_means2d[2 * f + 1];
            vpc[0] += vmx * p.fx / z; vpc[1] += vmy * p.fy / z;
            vpc[2] += -(vmx * p.fx * x + vmy * p.fy * y) / z2;
            if (v_depths) vpc[2] += v_depths[f];
            for (int r = 0; r < 3; r++)
                v_means[3 * n + r] += p.Rv[r] * vpc[0] + p.Rv[3 + r] * vpc[1] + p.Rv[6 + r] * vpc[2];
            /* v_cov = Rv^T v_covc Rv */
            real tmp[9], v_cov[9];
            mat3_Tmul(p.Rv, v_covc, tmp);
            mat3_mul(tmp, p.Rv, v_cov);
            /* v_M = (v_cov + v_cov^T) M */
            real S[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) S[i * 3 + j] = v_cov[i * 3 + j] + v_cov[j * 3 + i];
            real vM[9];
            mat3_mul(S, p.M, vM);
            const real* s = scales + 3 * n;
            real vR[9];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) vR[i * 3 + j] = vM[i * 3 + j] * s[j];
            for (int j = 0; j < 3; j++)
                v_scales[3 * n + j] += vM[j] * p.Rq[j] + vM[3 + j] * p.Rq[3 + j] + vM[6 + j] * p.Rq[6 + j];
            real w = p.qn[0], qx = p.qn[1], qy = p.qn[2], qz = p.qn[3];
            real vq[4];
            vq[0] = 2 * (-qz * vR[1] + qy * vR[2] + qz * vR[3] - qx * vR[5] - qy * vR[6] + qx * vR[7]);
            vq[1] = 2 * (qy * vR[1] + qz * vR[2] + qy * vR[3] - 2 * qx * vR[4] - w * vR[5] + qz * vR[6] + w * vR[7] - 2 * qx * vR[8]);
            vq[2] = 2 * (-2 * qy * vR[0] + qx * vR[1] + w * vR[2] + qx * vR[3] + qz * vR[5] - w * vR[6] + qz * vR[7] - 2 * qy * vR[8]);
            vq[3] = 2 * (-2 * qz * vR[0] - w * vR[1] + qx * vR[2] + w * vR[3] - 2 * qz * vR[4] + qy * vR[5] + qx * vR[6] + qy * vR[7]);
            real dot = w * vq[0] + qx * vq[1] + qy * vq[2] + qz * vq[3];
            for (int r = 0; r < 4; r++) v_quats[4 * n + r] += (vq[r] - p.qn[r] * dot) / p.qnorm;
        }
    }
    return 0;
}
