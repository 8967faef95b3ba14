// gs_math.h -- per-Gaussian arithmetic of the splat rasterizer (projection, SH colour and their
// VJPs), written once and used by the gfx950 kernels in gs_project.hip (projection forward / backward) and gs_blend.hip.  The functions are plain
// scalar fp32 code behind GS_HD so that a host build (tests/hostmath) can exercise exactly the
// same source on the CPU against the oracle without a GPU.
//
// Semantics follow gsplat 1.0.0 `fully_fused_projection` / `spherical_harmonics` as called from
// /root/reference/model/gaussian.py:353-367.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD static inline
#endif

namespace gs {

constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.999f;
constexpr float kTMin = 1e-4f;
constexpr float kRadiusDiscFloor = 0.01f;
constexpr float kFovClamp = 1.3f;   // (the float32 value widened into the chain's type, like eps2d and the discriminant floor)

// Per-camera constants, prepared once per launch on the device (see gs_project.hip: make_camera).
struct Camera {
    float R[9];      // world->camera rotation block of viewmat (row-major)
    float t[3];      // translation column
    float pos[3];    // camera centre in world space = inverse(viewmat)[:3,3]
    float fx, fy, cx, cy;
    float half_w, half_h;  // W/2, H/2 (exact): the FOV clamp 1.3 * tan(half fov) = 1.3 * half_w / fx is formed in the chain's own type
};

// Clamp limits of the perspective Jacobian (A.1): 1.3 * tan(half fov), in the arithmetic type of the projection chain --
// formed in fp32 they moved the conic of every clamped (off-screen) Gaussian by ~4e-7 relative against the fp64 oracle,
// the one deviation above half an ulp the round-3 parity bounds found (tests/test_gpu_parity.py CONICS_RTOL).
template <typename T>
GS_HD void fov_limits(const Camera& cam, T& limx, T& limy) {
    limx = (T)kFovClamp * ((T)cam.half_w / (T)cam.fx);
    limy = (T)kFovClamp * ((T)cam.half_h / (T)cam.fy);
}

struct Splat2D {
    float mx, my, depth;
    float A, B, C;   // conic
    float cxx, cyy;  // diagonal of the blurred 2-D covariance (for opacity-aware extents)
    int radius;      // 0 => culled
    int x0, x1, y0, y1;  // tile rectangle of the 3-sigma square (project_gaussian with tile > 0)
};

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Half-extents (pixels) of the region where opacity * exp(-sigma) can reach 1/255: the level set
// sigma = ln(255 * opacity) of a Gaussian with covariance diag entries (cxx, cyy) spans
// +-sqrt(2 tau cxx) in x.  Inflated slightly so that it is a strict superset of what the blend's
// own per-pixel test accepts; negative when the Gaussian can never pass the test.
GS_HD void alpha_extent(float opacity, float cxx, float cyy, float& ex, float& ey) {
    const float v = 255.0f * opacity;
    if (!(v > 1.0f)) { ex = -1.f; ey = -1.f; return; }
    const float tau2 = 2.0f * (logf(v) + 0.01f);
    ex = sqrtf(tau2 * cxx) * 1.0005f + 0.02f;
    ey = sqrtf(tau2 * cyy) * 1.0005f + 0.02f;
}

// Arithmetic type of the per-Gaussian projection chain (mean -> camera space -> covariance -> conic, its VJP, the
// radius and the tile rectangle).  fp64 by default: CDNA4 issues vector fp64 at half the fp32 rate, the chain is
// ~10^3 flops per Gaussian against ~10^5 per Gaussian in the blend, and its inverse (det = ac - b^2 of a 60:1 needle
// loses 3 digits, twice in the VJP) is the one badly conditioned step of the whole path.  Inputs and outputs stay fp32.
typedef double preal;

GS_HD float r_sqrt(float x) { return sqrtf(x); }
GS_HD double r_sqrt(double x) { return sqrt(x); }
GS_HD float r_ceil(float x) { return ceilf(x); }
GS_HD double r_ceil(double x) { return ceil(x); }
GS_HD float r_floor(float x) { return floorf(x); }
GS_HD double r_floor(double x) { return floor(x); }
GS_HD float r_max(float x, float y) { return fmaxf(x, y); }
GS_HD double r_max(double x, double y) { return fmax(x, y); }
GS_HD float r_min(float x, float y) { return fminf(x, y); }
GS_HD double r_min(double x, double y) { return fmin(x, y); }

// Intermediates the VJP needs again; recomputed in backward rather than stored.
template <typename T>
struct ProjChainT {
    T qw, qx, qy, qz, qinv;           // normalised quaternion, 1/|q|
    T R[9];                           // rotation of the Gaussian
    T M[9];                           // R diag(s)
    T cc00, cc01, cc02, cc11, cc12, cc22;  // camera-space covariance
    T x, y, z;                        // camera-space mean
    T tx, ty;                         // clamped x, y used inside J
    int clampx, clampy;               // -1 / 0 / +1
    T j00, j02, j11, j12;
    T a, b, c, det;                   // blurred 2-D covariance and determinant
};
typedef ProjChainT<preal> ProjChain;

template <typename T>
GS_HD bool project_chain(const float* mean, const float* quat, const float* scale,
                         const Camera& cam, float eps2d, float near_p, float far_p,
                         ProjChainT<T>& o) {
    const T px = mean[0], py = mean[1], pz = mean[2];
    T V[9];
    for (int i = 0; i < 9; ++i) V[i] = (T)cam.R[i];
    o.x = V[0] * px + V[1] * py + V[2] * pz + (T)cam.t[0];
    o.y = V[3] * px + V[4] * py + V[5] * pz + (T)cam.t[1];
    o.z = V[6] * px + V[7] * py + V[8] * pz + (T)cam.t[2];
    if (o.z < (T)near_p || o.z > (T)far_p) return false;

    const T q0 = quat[0], q1 = quat[1], q2 = quat[2], q3 = quat[3];
    const T qn2 = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    o.qinv = T(1) / r_sqrt(qn2);
    const T w = q0 * o.qinv, x = q1 * o.qinv, y = q2 * o.qinv, z = q3 * o.qinv;
    o.qw = w; o.qx = x; o.qy = y; o.qz = z;
    T* R = o.R;
    R[0] = T(1) - T(2) * (y * y + z * z); R[1] = T(2) * (x * y - w * z); R[2] = T(2) * (x * z + w * y);
    R[3] = T(2) * (x * y + w * z); R[4] = T(1) - T(2) * (x * x + z * z); R[5] = T(2) * (y * z - w * x);
    R[6] = T(2) * (x * z - w * y); R[7] = T(2) * (y * z + w * x); R[8] = T(1) - T(2) * (x * x + y * y);
    T* M = o.M;
    const T s0 = scale[0], s1 = scale[1], s2 = scale[2];
    M[0] = R[0] * s0; M[1] = R[1] * s1; M[2] = R[2] * s2;
    M[3] = R[3] * s0; M[4] = R[4] * s1; M[5] = R[5] * s2;
    M[6] = R[6] * s0; M[7] = R[7] * s1; M[8] = R[8] * s2;
    // world covariance (symmetric)
    const T c00 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    const T c01 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    const T c02 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    const T c11 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    const T c12 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    const T c22 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    // T = Rv * cov
    const T t00 = V[0] * c00 + V[1] * c01 + V[2] * c02, t01 = V[0] * c01 + V[1] * c11 + V[2] * c12, t02 = V[0] * c02 + V[1] * c12 + V[2] * c22;
    const T t10 = V[3] * c00 + V[4] * c01 + V[5] * c02, t11 = V[3] * c01 + V[4] * c11 + V[5] * c12, t12 = V[3] * c02 + V[4] * c12 + V[5] * c22;
    const T t20 = V[6] * c00 + V[7] * c01 + V[8] * c02, t21 = V[6] * c01 + V[7] * c11 + V[8] * c12, t22 = V[6] * c02 + V[7] * c12 + V[8] * c22;
    o.cc00 = t00 * V[0] + t01 * V[1] + t02 * V[2];
    o.cc01 = t00 * V[3] + t01 * V[4] + t02 * V[5];
    o.cc02 = t00 * V[6] + t01 * V[7] + t02 * V[8];
    o.cc11 = t10 * V[3] + t11 * V[4] + t12 * V[5];
    o.cc12 = t10 * V[6] + t11 * V[7] + t12 * V[8];
    o.cc22 = t20 * V[6] + t21 * V[7] + t22 * V[8];

    const T fx = cam.fx, fy = cam.fy;
    T limx, limy;
    fov_limits<T>(cam, limx, limy);
    const T rz = T(1) / o.z;
    const T rx = o.x * rz, ry = o.y * rz;
    o.clampx = rx > limx ? 1 : (rx < -limx ? -1 : 0);
    o.clampy = ry > limy ? 1 : (ry < -limy ? -1 : 0);
    o.tx = o.z * (o.clampx > 0 ? limx : (o.clampx < 0 ? -limx : rx));
    o.ty = o.z * (o.clampy > 0 ? limy : (o.clampy < 0 ? -limy : ry));
    o.j00 = fx * rz; o.j11 = fy * rz;
    o.j02 = -fx * o.tx * rz * rz; o.j12 = -fy * o.ty * rz * rz;
    o.a = o.j00 * o.j00 * o.cc00 + T(2) * o.j00 * o.j02 * o.cc02 + o.j02 * o.j02 * o.cc22 + (T)eps2d;
    o.b = o.j00 * o.j11 * o.cc01 + o.j00 * o.j12 * o.cc02 + o.j02 * o.j11 * o.cc12 + o.j02 * o.j12 * o.cc22;
    o.c = o.j11 * o.j11 * o.cc11 + T(2) * o.j11 * o.j12 * o.cc12 + o.j12 * o.j12 * o.cc22 + (T)eps2d;
    o.det = o.a * o.c - o.b * o.b;
    return true;
}

// Appendix A.3: tile rectangle [x0,x1) x [y0,y1) touched by the 3-sigma square.
template <typename T>
GS_HD void tile_rect(T mx, T my, int radius, int tile, int tw, int th, int& x0, int& x1, int& y0, int& y1) {
    const T inv = T(1) / (T)tile;  // exact for power-of-two tiles; division kept for others
    const T r = (tile & (tile - 1)) ? (T)radius / (T)tile : (T)radius * inv;
    const T cx = (tile & (tile - 1)) ? mx / (T)tile : mx * inv;
    const T cy = (tile & (tile - 1)) ? my / (T)tile : my * inv;
    const T fx0 = r_floor(cx - r), fx1 = r_ceil(cx + r), fy0 = r_floor(cy - r), fy1 = r_ceil(cy + r);
    x0 = (int)r_min(r_max(fx0, T(0)), (T)tw); x1 = (int)r_min(r_max(fx1, T(0)), (T)tw);
    y0 = (int)r_min(r_max(fy0, T(0)), (T)th); y1 = (int)r_min(r_max(fy1, T(0)), (T)th);
}

// Appendix A.1: full forward projection of one Gaussian for one camera.  With tile > 0 the tile rectangle of the
// 3-sigma square is taken from the unrounded centre as well (s.x0..s.y1).
template <typename T>
GS_HD Splat2D project_gaussian_t(const float* mean, const float* quat, const float* scale,
                                 const Camera& cam, int W, int H, float eps2d, float near_p,
                                 float far_p, float radius_clip, int tile, int tw, int th) {
    Splat2D s;
    s.mx = s.my = s.depth = s.A = s.B = s.C = s.cxx = s.cyy = 0.f;
    s.radius = 0; s.x0 = s.x1 = s.y0 = s.y1 = 0;
    ProjChainT<T> p;
    if (!project_chain<T>(mean, quat, scale, cam, eps2d, near_p, far_p, p)) return s;
    if (!(p.det > T(0))) return s;
    const T mid = T(0.5) * (p.a + p.c);
    const T lam = mid + r_sqrt(r_max((T)kRadiusDiscFloor, mid * mid - p.det));
    const T radius = r_ceil(T(3) * r_sqrt(lam));
    if (radius <= (T)radius_clip) return s;
    const T mx = (T)cam.fx * p.x / p.z + (T)cam.cx, my = (T)cam.fy * p.y / p.z + (T)cam.cy;
    if (mx + radius <= T(0) || mx - radius >= (T)W || my + radius <= T(0) || my - radius >= (T)H) return s;
    const T rdet = T(1) / p.det;
    s.mx = (float)mx; s.my = (float)my; s.depth = (float)p.z;
    s.A = (float)(p.c * rdet); s.B = (float)(-p.b * rdet); s.C = (float)(p.a * rdet);
    s.cxx = (float)p.a; s.cyy = (float)p.c;
    s.radius = (int)radius;
    if (tile > 0) tile_rect<T>(mx, my, s.radius, tile, tw, th, s.x0, s.x1, s.y0, s.y1);
    return s;
}

GS_HD Splat2D project_gaussian(const float* mean, const float* quat, const float* scale,
                               const Camera& cam, int W, int H, float eps2d, float near_p,
                               float far_p, float radius_clip, int tile = 0, int tw = 0, int th = 0) {
    return project_gaussian_t<preal>(mean, quat, scale, cam, W, H, eps2d, near_p, far_p, radius_clip, tile, tw, th);
}

// Minimum of the positive-definite form a dx^2 + b dx dy + c dy^2 over the axis-aligned
// rectangle [dx0,dx1] x [dy0,dy1] (offsets from the Gaussian's centre): 0 when the centre is
// inside, otherwise attained on an edge, where the form is a 1-D parabola in the free coordinate.
GS_HD float quad_min_on_rect(float a, float b, float c, float dx0, float dx1, float dy0, float dy1) {
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return 0.f;
    const float hbc = -0.5f * b / c, hba = -0.5f * b / a;
    float m = 3.0e38f;
    {   // vertical edges: dx fixed, minimise over dy
        const float X0 = dx0, X1 = dx1;
        const float t0 = fminf(fmaxf(hbc * X0, dy0), dy1), t1 = fminf(fmaxf(hbc * X1, dy0), dy1);
        m = fminf(m, a * X0 * X0 + t0 * (b * X0 + c * t0));
        m = fminf(m, a * X1 * X1 + t1 * (b * X1 + c * t1));
    }
    {   // horizontal edges: dy fixed, minimise over dx
        const float Y0 = dy0, Y1 = dy1;
        const float t0 = fminf(fmaxf(hba * Y0, dx0), dx1), t1 = fminf(fmaxf(hba * Y1, dx0), dx1);
        m = fminf(m, c * Y0 * Y0 + t0 * (b * Y0 + a * t0));
        m = fminf(m, c * Y1 * Y1 + t1 * (b * Y1 + a * t1));
    }
    return m;
}

// Shrinks a tile rectangle to the tiles that contain at least one pixel centre (j + 0.5) with
// |mx - (j+0.5)| <= ex and |my - (i+0.5)| <= ey; empties it when there is none.
GS_HD void tile_rect_tight(float mx, float my, float ex, float ey, int W, int H, int tile, int& x0,
                           int& x1, int& y0, int& y1) {
    if (ex < 0.f || ey < 0.f) { x1 = x0; y1 = y0; return; }
    const float jlo = fmaxf(ceilf(mx - ex - 0.5f), 0.f), jhi = fminf(floorf(mx + ex - 0.5f), (float)(W - 1));
    const float ilo = fmaxf(ceilf(my - ey - 0.5f), 0.f), ihi = fminf(floorf(my + ey - 0.5f), (float)(H - 1));
    if (jlo > jhi || ilo > ihi) { x1 = x0; y1 = y0; return; }
    const int tx0 = (int)jlo / tile, tx1 = (int)jhi / tile + 1, ty0 = (int)ilo / tile, ty1 = (int)ihi / tile + 1;
    x0 = x0 > tx0 ? x0 : tx0; x1 = x1 < tx1 ? x1 : tx1;
    y0 = y0 > ty0 ? y0 : ty0; y1 = y1 < ty1 ? y1 : ty1;
    if (x1 <= x0 || y1 <= y0) { x1 = x0; y1 = y0; }
}

// ---------------------------------------------------------------------------------- SH
constexpr float kC0 = 0.2820947917738781f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC20 = 1.0925484305920792f, kC21 = 0.31539156525252005f, kC22 = 0.5462742152960396f;
constexpr float kC30 = 0.5900435899266435f, kC31 = 2.890611442640554f, kC32 = 0.4570457994644658f,
                kC33 = 0.3731763325901154f, kC34 = 1.445305721320277f;

GS_HD void sh_basis(int degree, float x, float y, float z, float* Y) {
    Y[0] = kC0;
    if (degree < 1) return;
    Y[1] = -kC1 * y; Y[2] = kC1 * z; Y[3] = -kC1 * x;
    if (degree < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z;
    Y[4] = kC20 * x * y; Y[5] = -kC20 * y * z; Y[6] = kC21 * (2.f * zz - xx - yy);
    Y[7] = -kC20 * x * z; Y[8] = kC22 * (xx - yy);
    if (degree < 3) return;
    Y[9] = -kC30 * y * (3.f * xx - yy); Y[10] = kC31 * x * y * z;
    Y[11] = -kC32 * y * (4.f * zz - xx - yy); Y[12] = kC33 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    Y[13] = -kC32 * x * (4.f * zz - xx - yy); Y[14] = kC34 * z * (xx - yy);
    Y[15] = -kC30 * x * (xx - 3.f * yy);
}

// Unit view direction from the camera centre to the Gaussian; returns |d| (0 => degenerate).
GS_HD float view_dir(const float* mean, const Camera& cam, float& ux, float& uy, float& uz) {
    const float dx = mean[0] - cam.pos[0], dy = mean[1] - cam.pos[1], dz = mean[2] - cam.pos[2];
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    const float inv = n > 0.f ? 1.0f / n : 1.0f;
    ux = dx * inv; uy = dy * inv; uz = dz * inv;
    return n;
}

// Appendix A.2: rgb = max(sum_k Y_k sh[k] + 0.5, 0).  `sh` points at this Gaussian's [K,3] block
// (any float pointer: global memory or an LDS staging row).
GS_HD void sh_to_rgb(int degree, const float* sh, float ux, float uy, float uz, float* rgb) {
    float Y[16];
    sh_basis(degree, ux, uy, uz, Y);
    const int Ka = (degree + 1) * (degree + 1);
    float r = 0.f, g = 0.f, b = 0.f;
    for (int k = 0; k < Ka; ++k) {
        r += Y[k] * sh[3 * k]; g += Y[k] * sh[3 * k + 1]; b += Y[k] * sh[3 * k + 2];
    }
    rgb[0] = fmaxf(r + 0.5f, 0.f); rgb[1] = fmaxf(g + 0.5f, 0.f); rgb[2] = fmaxf(b + 0.5f, 0.f);
}

// g = sum_k d[k] * dY_k/du at the unit direction (x, y, z), u treated as three free variables (k >= 1; degree >= 1).
GS_HD void sh_dir_grad(int degree, const float* d, float x, float y, float z, float& gx, float& gy, float& gz) {
    gx = -kC1 * d[3]; gy = -kC1 * d[1]; gz = kC1 * d[2];
    if (degree >= 2) {
        gx += kC20 * y * d[4] - 2.f * kC21 * x * d[6] - kC20 * z * d[7] + 2.f * kC22 * x * d[8];
        gy += kC20 * x * d[4] - kC20 * z * d[5] - 2.f * kC21 * y * d[6] - 2.f * kC22 * y * d[8];
        gz += -kC20 * y * d[5] + 4.f * kC21 * z * d[6] - kC20 * x * d[7];
    }
    if (degree >= 3) {
        const float xx = x * x, yy = y * y, zz = z * z;
        gx += -6.f * kC30 * x * y * d[9] + kC31 * y * z * d[10] + 2.f * kC32 * x * y * d[11]
              - 6.f * kC33 * x * z * d[12] - kC32 * (4.f * zz - 3.f * xx - yy) * d[13]
              + 2.f * kC34 * x * z * d[14] - kC30 * (3.f * xx - 3.f * yy) * d[15];
        gy += -kC30 * (3.f * xx - 3.f * yy) * d[9] + kC31 * x * z * d[10]
              - kC32 * (4.f * zz - xx - 3.f * yy) * d[11] - 6.f * kC33 * y * z * d[12]
              + 2.f * kC32 * x * y * d[13] - 2.f * kC34 * y * z * d[14] + 6.f * kC30 * x * y * d[15];
        gz += kC31 * x * y * d[10] - 8.f * kC32 * y * z * d[11]
              + kC33 * (6.f * zz - 3.f * xx - 3.f * yy) * d[12] - 8.f * kC32 * x * z * d[13]
              + kC34 * (xx - yy) * d[14];
    }
}

// v_mean += d(unit direction)/d(mean)^T g   (direction = (mean - camera centre) / dnorm)
GS_HD void dir_grad_to_mean(float gx, float gy, float gz, float x, float y, float z, float dnorm, float* v_mean) {
    const float ud = x * gx + y * gy + z * gz;
    const float inv = 1.0f / dnorm;
    v_mean[0] += (gx - x * ud) * inv; v_mean[1] += (gy - y * ud) * inv; v_mean[2] += (gz - z * ud) * inv;
}

// Appendix A.6 colour path.  In: post-activation rgb (for the clamp mask), v_rgb, SH block.
// Out: v_sh[k][3] += Y_k v_pre  (written to `v_sh`, k < Ka), returns v_mean contribution.
GS_HD void sh_vjp(int degree, const float* sh, const float* rgb, const float* v_rgb, float ux,
                  float uy, float uz, float dnorm, float* v_sh, float* v_mean, bool accumulate) {
    const float vr = rgb[0] > 0.f ? v_rgb[0] : 0.f;
    const float vg = rgb[1] > 0.f ? v_rgb[1] : 0.f;
    const float vb = rgb[2] > 0.f ? v_rgb[2] : 0.f;
    float Y[16];
    sh_basis(degree, ux, uy, uz, Y);
    const int Ka = (degree + 1) * (degree + 1);
    float d[16];  // d[k] = sh[k] . v_pre
    for (int k = 0; k < Ka; ++k) {
        d[k] = sh[3 * k] * vr + sh[3 * k + 1] * vg + sh[3 * k + 2] * vb;
        if (accumulate) {
            v_sh[3 * k] += Y[k] * vr; v_sh[3 * k + 1] += Y[k] * vg; v_sh[3 * k + 2] += Y[k] * vb;
        } else {
            v_sh[3 * k] = Y[k] * vr; v_sh[3 * k + 1] = Y[k] * vg; v_sh[3 * k + 2] = Y[k] * vb;
        }
    }
    if (degree < 1 || !(dnorm > 0.f)) return;
    float gx, gy, gz;
    sh_dir_grad(degree, d, ux, uy, uz, gx, gy, gz);
    dir_grad_to_mean(gx, gy, gz, ux, uy, uz, dnorm, v_mean);
}

// The direction Jacobian of the pre-clamp colour, G[i][c] = d(sum_k Y_k(u) sh[k][c]) / du_i  (3 x 3, rows padded to four
// floats): what the forward leaves for the backward so that the latter needs no SH coefficient (sh_vjp_jac).
GS_HD void sh_dir_jacobian(int degree, const float* sh, float ux, float uy, float uz, float* G) {
    const int Ka = (degree + 1) * (degree + 1);
    for (int c = 0; c < 3; ++c) {
        float d[16];
        for (int k = 0; k < Ka; ++k) d[k] = sh[3 * k + c];
        float gx, gy, gz;
        sh_dir_grad(degree, d, ux, uy, uz, gx, gy, gz);
        G[c] = gx; G[4 + c] = gy; G[8 + c] = gz;
    }
    G[3] = G[7] = G[11] = 0.f;
}

// sh_vjp without the coefficients: v_sh[k][:] = Y_k v_pre, and the direction term of v_mean through G (degree >= 1).
GS_HD void sh_vjp_jac(int degree, const float* G, const float* rgb, const float* v_rgb, float ux, float uy, float uz,
                      float dnorm, float* v_sh, float* v_mean) {
    const float vr = rgb[0] > 0.f ? v_rgb[0] : 0.f;
    const float vg = rgb[1] > 0.f ? v_rgb[1] : 0.f;
    const float vb = rgb[2] > 0.f ? v_rgb[2] : 0.f;
    float Y[16];
    sh_basis(degree, ux, uy, uz, Y);
    const int Ka = (degree + 1) * (degree + 1);
    for (int k = 0; k < Ka; ++k) { v_sh[3 * k] = Y[k] * vr; v_sh[3 * k + 1] = Y[k] * vg; v_sh[3 * k + 2] = Y[k] * vb; }
    if (degree < 1 || !(dnorm > 0.f)) return;
    const float gx = G[0] * vr + G[1] * vg + G[2] * vb;
    const float gy = G[4] * vr + G[5] * vg + G[6] * vb;
    const float gz = G[8] * vr + G[9] * vg + G[10] * vb;
    dir_grad_to_mean(gx, gy, gz, ux, uy, uz, dnorm, v_mean);
}

// Appendix A.6 projection VJP for one (camera, Gaussian): adds into v_mean[3], v_quat[4], v_scale[3].
template <typename T>
GS_HD void project_vjp(const float* scale, const Camera& cam, const ProjChainT<T>& p, float v_mx_f,
                       float v_my_f, float vA_f, float vB_f, float vC_f, float v_depth_f, float* v_mean,
                       float* v_quat, float* v_scale) {
    const T v_mx = v_mx_f, v_my = v_my_f, vA = vA_f, vB = vB_f, vC = vC_f, v_depth = v_depth_f;
    const T fx = cam.fx, fy = cam.fy;
    T limx, limy;
    fov_limits<T>(cam, limx, limy);
    T V[9];
    for (int i = 0; i < 9; ++i) V[i] = (T)cam.R[i];
    // conic = inverse(cov2'); G = -X V X with V = [[vA, vB/2],[vB/2, vC]]
    const T rdet = T(1) / p.det;
    const T X00 = p.c * rdet, X01 = -p.b * rdet, X11 = p.a * rdet;
    const T h = T(0.5) * vB;
    const T xv00 = X00 * vA + X01 * h, xv01 = X00 * h + X01 * vC;
    const T xv10 = X01 * vA + X11 * h, xv11 = X01 * h + X11 * vC;
    const T g00 = -(xv00 * X00 + xv01 * X01);
    const T g01 = -(xv00 * X01 + xv01 * X11);
    const T g11 = -(xv10 * X01 + xv11 * X11);
    const T j00 = p.j00, j02 = p.j02, j11 = p.j11, j12 = p.j12;
    // GJ (2x3)
    const T a0 = g00 * j00, a1 = g01 * j11, a2 = g00 * j02 + g01 * j12;
    const T b0 = g01 * j00, b1 = g11 * j11, b2 = g01 * j02 + g11 * j12;
    // v_covc = J^T G J (symmetric)
    const T vc00 = j00 * a0, vc01 = j00 * a1, vc02 = j00 * a2;
    const T vc11 = j11 * b1, vc12 = j11 * b2, vc22 = j02 * a2 + j12 * b2;
    // v_J = 2 (GJ) covc ; only the four non-constant entries
    const T vJ00 = T(2) * (a0 * p.cc00 + a1 * p.cc01 + a2 * p.cc02);
    const T vJ02 = T(2) * (a0 * p.cc02 + a1 * p.cc12 + a2 * p.cc22);
    const T vJ11 = T(2) * (b0 * p.cc01 + b1 * p.cc11 + b2 * p.cc12);
    const T vJ12 = T(2) * (b0 * p.cc02 + b1 * p.cc12 + b2 * p.cc22);
    const T rz = T(1) / p.z, rz2 = rz * rz, rz3 = rz2 * rz;
    T vx = T(0), vy = T(0);
    T vz = -vJ00 * fx * rz2 - vJ11 * fy * rz2 + T(2) * vJ02 * fx * p.tx * rz3 + T(2) * vJ12 * fy * p.ty * rz3;
    const T v_tx = -vJ02 * fx * rz2, v_ty = -vJ12 * fy * rz2;
    if (p.clampx == 0) vx += v_tx; else vz += v_tx * (p.clampx > 0 ? limx : -limx);
    if (p.clampy == 0) vy += v_ty; else vz += v_ty * (p.clampy > 0 ? limy : -limy);
    vx += v_mx * fx * rz; vy += v_my * fy * rz;
    vz += -(v_mx * fx * p.x + v_my * fy * p.y) * rz2 + v_depth;
    v_mean[0] += (float)(V[0] * vx + V[3] * vy + V[6] * vz);
    v_mean[1] += (float)(V[1] * vx + V[4] * vy + V[7] * vz);
    v_mean[2] += (float)(V[2] * vx + V[5] * vy + V[8] * vz);
    // v_cov = V^T v_covc V (symmetric): first U = v_covc V
    const T u00 = vc00 * V[0] + vc01 * V[3] + vc02 * V[6], u01 = vc00 * V[1] + vc01 * V[4] + vc02 * V[7], u02 = vc00 * V[2] + vc01 * V[5] + vc02 * V[8];
    const T u10 = vc01 * V[0] + vc11 * V[3] + vc12 * V[6], u11 = vc01 * V[1] + vc11 * V[4] + vc12 * V[7], u12 = vc01 * V[2] + vc11 * V[5] + vc12 * V[8];
    const T u20 = vc02 * V[0] + vc12 * V[3] + vc22 * V[6], u21 = vc02 * V[1] + vc12 * V[4] + vc22 * V[7], u22 = vc02 * V[2] + vc12 * V[5] + vc22 * V[8];
    const T w00 = V[0] * u00 + V[3] * u10 + V[6] * u20;
    const T w01 = V[0] * u01 + V[3] * u11 + V[6] * u21;
    const T w02 = V[0] * u02 + V[3] * u12 + V[6] * u22;
    const T w11 = V[1] * u01 + V[4] * u11 + V[7] * u21;
    const T w12 = V[1] * u02 + V[4] * u12 + V[7] * u22;
    const T w22 = V[2] * u02 + V[5] * u12 + V[8] * u22;
    // v_M = 2 v_cov M
    const T* M = p.M;
    T vM[9];
    vM[0] = T(2) * (w00 * M[0] + w01 * M[3] + w02 * M[6]); vM[1] = T(2) * (w00 * M[1] + w01 * M[4] + w02 * M[7]); vM[2] = T(2) * (w00 * M[2] + w01 * M[5] + w02 * M[8]);
    vM[3] = T(2) * (w01 * M[0] + w11 * M[3] + w12 * M[6]); vM[4] = T(2) * (w01 * M[1] + w11 * M[4] + w12 * M[7]); vM[5] = T(2) * (w01 * M[2] + w11 * M[5] + w12 * M[8]);
    vM[6] = T(2) * (w02 * M[0] + w12 * M[3] + w22 * M[6]); vM[7] = T(2) * (w02 * M[1] + w12 * M[4] + w22 * M[7]); vM[8] = T(2) * (w02 * M[2] + w12 * M[5] + w22 * M[8]);
    const T* R = p.R;
    v_scale[0] += (float)(vM[0] * R[0] + vM[3] * R[3] + vM[6] * R[6]);
    v_scale[1] += (float)(vM[1] * R[1] + vM[4] * R[4] + vM[7] * R[7]);
    v_scale[2] += (float)(vM[2] * R[2] + vM[5] * R[5] + vM[8] * R[8]);
    T vR[9];
    for (int i = 0; i < 3; ++i) { vR[3 * i] = vM[3 * i] * (T)scale[0]; vR[3 * i + 1] = vM[3 * i + 1] * (T)scale[1]; vR[3 * i + 2] = vM[3 * i + 2] * (T)scale[2]; }
    const T w = p.qw, x = p.qx, y = p.qy, z = p.qz;
    const T vq0 = T(2) * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
    const T vq1 = T(2) * (y * vR[1] + z * vR[2] + y * vR[3] - T(2) * x * vR[4] - w * vR[5] + z * vR[6] + w * vR[7] - T(2) * x * vR[8]);
    const T vq2 = T(2) * (-T(2) * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] - w * vR[6] + z * vR[7] - T(2) * y * vR[8]);
    const T vq3 = T(2) * (-T(2) * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - T(2) * z * vR[4] + y * vR[5] + x * vR[6] + y * vR[7]);
    const T dot = w * vq0 + x * vq1 + y * vq2 + z * vq3;
    v_quat[0] += (float)((vq0 - w * dot) * p.qinv); v_quat[1] += (float)((vq1 - x * dot) * p.qinv);
    v_quat[2] += (float)((vq2 - y * dot) * p.qinv); v_quat[3] += (float)((vq3 - z * dot) * p.qinv);
}

// Camera constants from raw viewmat[16] / K[9] (row-major), incl. the general 3x3 inverse for the
// camera centre (what torch.inverse(viewmats)[:, :3, 3] yields for an affine view matrix).
GS_HD void make_camera(const float* V, const float* K, int W, int H, Camera& cam) {
    cam.R[0] = V[0]; cam.R[1] = V[1]; cam.R[2] = V[2];
    cam.R[3] = V[4]; cam.R[4] = V[5]; cam.R[5] = V[6];
    cam.R[6] = V[8]; cam.R[7] = V[9]; cam.R[8] = V[10];
    cam.t[0] = V[3]; cam.t[1] = V[7]; cam.t[2] = V[11];
    const float a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[6], g = V[8], h = V[9], i = V[10];
    const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const float r = 1.0f / det;
    const float i00 = (e * i - f * h) * r, i01 = (c * h - b * i) * r, i02 = (b * f - c * e) * r;
    const float i10 = (f * g - d * i) * r, i11 = (a * i - c * g) * r, i12 = (c * d - a * f) * r;
    const float i20 = (d * h - e * g) * r, i21 = (b * g - a * h) * r, i22 = (a * e - b * d) * r;
    cam.pos[0] = -(i00 * cam.t[0] + i01 * cam.t[1] + i02 * cam.t[2]);
    cam.pos[1] = -(i10 * cam.t[0] + i11 * cam.t[1] + i12 * cam.t[2]);
    cam.pos[2] = -(i20 * cam.t[0] + i21 * cam.t[1] + i22 * cam.t[2]);
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    cam.half_w = 0.5f * (float)W;
    cam.half_h = 0.5f * (float)H;
}

}  // namespace gs
