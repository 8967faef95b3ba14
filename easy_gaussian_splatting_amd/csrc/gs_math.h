// gs_math.h -- per-Gaussian arithmetic of the splat rasterizer (projection, SH colour and their
// VJPs), written once and used by the gfx950 kernels in gs_kernels.hip.  The functions are plain
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
constexpr float kFovClamp = 1.3f;
constexpr float kRadiusDiscFloor = 0.01f;

// Per-camera constants, prepared once per launch on the device (see gs_kernels.hip: camera_prep).
struct Camera {
    float R[9];      // world->camera rotation block of viewmat (row-major)
    float t[3];      // translation column
    float pos[3];    // camera centre in world space = inverse(viewmat)[:3,3]
    float fx, fy, cx, cy;
    float limx, limy;  // 1.3 * tan(half fov)
};

struct Splat2D {
    float mx, my, depth;
    float A, B, C;   // conic
    float cxx, cyy;  // diagonal of the blurred 2-D covariance (for opacity-aware extents)
    int radius;      // 0 => culled
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

// Intermediates the VJP needs again; recomputed in backward rather than stored.
struct ProjChain {
    float qw, qx, qy, qz, qinv;           // normalised quaternion, 1/|q|
    float R[9];                           // rotation of the Gaussian
    float M[9];                           // R diag(s)
    float cc00, cc01, cc02, cc11, cc12, cc22;  // camera-space covariance
    float x, y, z;                        // camera-space mean
    float tx, ty;                         // clamped x, y used inside J
    int clampx, clampy;                   // -1 / 0 / +1
    float j00, j02, j11, j12;
    float a, b, c, det;                   // blurred 2-D covariance and determinant
};

GS_HD bool project_chain(const float* mean, const float* quat, const float* scale,
                         const Camera& cam, float eps2d, float near_p, float far_p,
                         ProjChain& o) {
    const float px = mean[0], py = mean[1], pz = mean[2];
    o.x = cam.R[0] * px + cam.R[1] * py + cam.R[2] * pz + cam.t[0];
    o.y = cam.R[3] * px + cam.R[4] * py + cam.R[5] * pz + cam.t[1];
    o.z = cam.R[6] * px + cam.R[7] * py + cam.R[8] * pz + cam.t[2];
    if (o.z < near_p || o.z > far_p) return false;

    const float qn2 = quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3];
    o.qinv = 1.0f / sqrtf(qn2);
    const float w = quat[0] * o.qinv, x = quat[1] * o.qinv, y = quat[2] * o.qinv, z = quat[3] * o.qinv;
    o.qw = w; o.qx = x; o.qy = y; o.qz = z;
    float* R = o.R;
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
    float* M = o.M;
    const float s0 = scale[0], s1 = scale[1], s2 = scale[2];
    M[0] = R[0] * s0; M[1] = R[1] * s1; M[2] = R[2] * s2;
    M[3] = R[3] * s0; M[4] = R[4] * s1; M[5] = R[5] * s2;
    M[6] = R[6] * s0; M[7] = R[7] * s1; M[8] = R[8] * s2;
    // world covariance (symmetric)
    const float c00 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    const float c01 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    const float c02 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    const float c11 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    const float c12 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    const float c22 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    // T = Rv * cov
    const float* V = cam.R;
    const float t00 = V[0] * c00 + V[1] * c01 + V[2] * c02, t01 = V[0] * c01 + V[1] * c11 + V[2] * c12, t02 = V[0] * c02 + V[1] * c12 + V[2] * c22;
    const float t10 = V[3] * c00 + V[4] * c01 + V[5] * c02, t11 = V[3] * c01 + V[4] * c11 + V[5] * c12, t12 = V[3] * c02 + V[4] * c12 + V[5] * c22;
    const float t20 = V[6] * c00 + V[7] * c01 + V[8] * c02, t21 = V[6] * c01 + V[7] * c11 + V[8] * c12, t22 = V[6] * c02 + V[7] * c12 + V[8] * c22;
    o.cc00 = t00 * V[0] + t01 * V[1] + t02 * V[2];
    o.cc01 = t00 * V[3] + t01 * V[4] + t02 * V[5];
    o.cc02 = t00 * V[6] + t01 * V[7] + t02 * V[8];
    o.cc11 = t10 * V[3] + t11 * V[4] + t12 * V[5];
    o.cc12 = t10 * V[6] + t11 * V[7] + t12 * V[8];
    o.cc22 = t20 * V[6] + t21 * V[7] + t22 * V[8];

    const float rz = 1.0f / o.z;
    const float rx = o.x * rz, ry = o.y * rz;
    o.clampx = rx > cam.limx ? 1 : (rx < -cam.limx ? -1 : 0);
    o.clampy = ry > cam.limy ? 1 : (ry < -cam.limy ? -1 : 0);
    o.tx = o.z * (o.clampx > 0 ? cam.limx : (o.clampx < 0 ? -cam.limx : rx));
    o.ty = o.z * (o.clampy > 0 ? cam.limy : (o.clampy < 0 ? -cam.limy : ry));
    o.j00 = cam.fx * rz; o.j11 = cam.fy * rz;
    o.j02 = -cam.fx * o.tx * rz * rz; o.j12 = -cam.fy * o.ty * rz * rz;
    o.a = o.j00 * o.j00 * o.cc00 + 2.f * o.j00 * o.j02 * o.cc02 + o.j02 * o.j02 * o.cc22 + eps2d;
    o.b = o.j00 * o.j11 * o.cc01 + o.j00 * o.j12 * o.cc02 + o.j02 * o.j11 * o.cc12 + o.j02 * o.j12 * o.cc22;
    o.c = o.j11 * o.j11 * o.cc11 + 2.f * o.j11 * o.j12 * o.cc12 + o.j12 * o.j12 * o.cc22 + eps2d;
    o.det = o.a * o.c - o.b * o.b;
    return true;
}

// Appendix A.1: full forward projection of one Gaussian for one camera.
GS_HD Splat2D project_gaussian(const float* mean, const float* quat, const float* scale,
                               const Camera& cam, int W, int H, float eps2d, float near_p,
                               float far_p, float radius_clip) {
    Splat2D s;
    s.mx = s.my = s.depth = s.A = s.B = s.C = s.cxx = s.cyy = 0.f;
    s.radius = 0;
    ProjChain p;
    if (!project_chain(mean, quat, scale, cam, eps2d, near_p, far_p, p)) return s;
    if (!(p.det > 0.f)) return s;
    const float mid = 0.5f * (p.a + p.c);
    const float lam = mid + sqrtf(fmaxf(kRadiusDiscFloor, mid * mid - p.det));
    const float radius = ceilf(3.0f * sqrtf(lam));
    if (radius <= radius_clip) return s;
    const float mx = cam.fx * p.x / p.z + cam.cx, my = cam.fy * p.y / p.z + cam.cy;
    if (mx + radius <= 0.f || mx - radius >= (float)W || my + radius <= 0.f || my - radius >= (float)H) return s;
    const float rdet = 1.0f / p.det;
    s.mx = mx; s.my = my; s.depth = p.z;
    s.A = p.c * rdet; s.B = -p.b * rdet; s.C = p.a * rdet;
    s.cxx = p.a; s.cyy = p.c;
    s.radius = (int)radius;
    return s;
}

// Appendix A.3: tile rectangle [x0,x1) x [y0,y1) touched by the 3-sigma square.
GS_HD void tile_rect(float mx, float my, int radius, int tile, int tw, int th, int& x0, int& x1,
                     int& y0, int& y1) {
    const float inv = 1.0f / (float)tile;  // exact for power-of-two tiles; division kept for others
    const float r = (tile & (tile - 1)) ? (float)radius / (float)tile : (float)radius * inv;
    const float cx = (tile & (tile - 1)) ? mx / (float)tile : mx * inv;
    const float cy = (tile & (tile - 1)) ? my / (float)tile : my * inv;
    const float fx0 = floorf(cx - r), fx1 = ceilf(cx + r), fy0 = floorf(cy - r), fy1 = ceilf(cy + r);
    x0 = (int)fminf(fmaxf(fx0, 0.f), (float)tw); x1 = (int)fminf(fmaxf(fx1, 0.f), (float)tw);
    y0 = (int)fminf(fmaxf(fy0, 0.f), (float)th); y1 = (int)fminf(fmaxf(fy1, 0.f), (float)th);
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
    const float x = ux, y = uy, z = uz;
    float gx = -kC1 * d[3], gy = -kC1 * d[1], gz = kC1 * d[2];
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
    const float ud = x * gx + y * gy + z * gz;
    const float inv = 1.0f / dnorm;
    v_mean[0] += (gx - x * ud) * inv; v_mean[1] += (gy - y * ud) * inv; v_mean[2] += (gz - z * ud) * inv;
}

// Appendix A.6 projection VJP for one (camera, Gaussian): adds into v_mean[3], v_quat[4], v_scale[3].
GS_HD void project_vjp(const float* scale, const Camera& cam, const ProjChain& p, float v_mx,
                       float v_my, float vA, float vB, float vC, float v_depth, float* v_mean,
                       float* v_quat, float* v_scale) {
    // conic = inverse(cov2'); G = -X V X with V = [[vA, vB/2],[vB/2, vC]]
    const float rdet = 1.0f / p.det;
    const float X00 = p.c * rdet, X01 = -p.b * rdet, X11 = p.a * rdet;
    const float h = 0.5f * vB;
    const float xv00 = X00 * vA + X01 * h, xv01 = X00 * h + X01 * vC;
    const float xv10 = X01 * vA + X11 * h, xv11 = X01 * h + X11 * vC;
    const float g00 = -(xv00 * X00 + xv01 * X01);
    const float g01 = -(xv00 * X01 + xv01 * X11);
    const float g11 = -(xv10 * X01 + xv11 * X11);
    const float j00 = p.j00, j02 = p.j02, j11 = p.j11, j12 = p.j12;
    // GJ (2x3)
    const float a0 = g00 * j00, a1 = g01 * j11, a2 = g00 * j02 + g01 * j12;
    const float b0 = g01 * j00, b1 = g11 * j11, b2 = g01 * j02 + g11 * j12;
    // v_covc = J^T G J (symmetric)
    const float vc00 = j00 * a0, vc01 = j00 * a1, vc02 = j00 * a2;
    const float vc11 = j11 * b1, vc12 = j11 * b2, vc22 = j02 * a2 + j12 * b2;
    // v_J = 2 (GJ) covc ; only the four non-constant entries
    const float vJ00 = 2.f * (a0 * p.cc00 + a1 * p.cc01 + a2 * p.cc02);
    const float vJ02 = 2.f * (a0 * p.cc02 + a1 * p.cc12 + a2 * p.cc22);
    const float vJ11 = 2.f * (b0 * p.cc01 + b1 * p.cc11 + b2 * p.cc12);
    const float vJ12 = 2.f * (b0 * p.cc02 + b1 * p.cc12 + b2 * p.cc22);
    const float rz = 1.0f / p.z, rz2 = rz * rz, rz3 = rz2 * rz;
    float vx = 0.f, vy = 0.f;
    float vz = -vJ00 * cam.fx * rz2 - vJ11 * cam.fy * rz2 + 2.f * vJ02 * cam.fx * p.tx * rz3 + 2.f * vJ12 * cam.fy * p.ty * rz3;
    const float v_tx = -vJ02 * cam.fx * rz2, v_ty = -vJ12 * cam.fy * rz2;
    if (p.clampx == 0) vx += v_tx; else vz += v_tx * (p.clampx > 0 ? cam.limx : -cam.limx);
    if (p.clampy == 0) vy += v_ty; else vz += v_ty * (p.clampy > 0 ? cam.limy : -cam.limy);
    vx += v_mx * cam.fx * rz; vy += v_my * cam.fy * rz;
    vz += -(v_mx * cam.fx * p.x + v_my * cam.fy * p.y) * rz2 + v_depth;
    const float* V = cam.R;
    v_mean[0] += V[0] * vx + V[3] * vy + V[6] * vz;
    v_mean[1] += V[1] * vx + V[4] * vy + V[7] * vz;
    v_mean[2] += V[2] * vx + V[5] * vy + V[8] * vz;
    // v_cov = V^T v_covc V (symmetric): first U = v_covc V
    const float u00 = vc00 * V[0] + vc01 * V[3] + vc02 * V[6], u01 = vc00 * V[1] + vc01 * V[4] + vc02 * V[7], u02 = vc00 * V[2] + vc01 * V[5] + vc02 * V[8];
    const float u10 = vc01 * V[0] + vc11 * V[3] + vc12 * V[6], u11 = vc01 * V[1] + vc11 * V[4] + vc12 * V[7], u12 = vc01 * V[2] + vc11 * V[5] + vc12 * V[8];
    const float u20 = vc02 * V[0] + vc12 * V[3] + vc22 * V[6], u21 = vc02 * V[1] + vc12 * V[4] + vc22 * V[7], u22 = vc02 * V[2] + vc12 * V[5] + vc22 * V[8];
    const float w00 = V[0] * u00 + V[3] * u10 + V[6] * u20;
    const float w01 = V[0] * u01 + V[3] * u11 + V[6] * u21;
    const float w02 = V[0] * u02 + V[3] * u12 + V[6] * u22;
    const float w11 = V[1] * u01 + V[4] * u11 + V[7] * u21;
    const float w12 = V[1] * u02 + V[4] * u12 + V[7] * u22;
    const float w22 = V[2] * u02 + V[5] * u12 + V[8] * u22;
    // v_M = 2 v_cov M
    const float* M = p.M;
    float vM[9];
    vM[0] = 2.f * (w00 * M[0] + w01 * M[3] + w02 * M[6]); vM[1] = 2.f * (w00 * M[1] + w01 * M[4] + w02 * M[7]); vM[2] = 2.f * (w00 * M[2] + w01 * M[5] + w02 * M[8]);
    vM[3] = 2.f * (w01 * M[0] + w11 * M[3] + w12 * M[6]); vM[4] = 2.f * (w01 * M[1] + w11 * M[4] + w12 * M[7]); vM[5] = 2.f * (w01 * M[2] + w11 * M[5] + w12 * M[8]);
    vM[6] = 2.f * (w02 * M[0] + w12 * M[3] + w22 * M[6]); vM[7] = 2.f * (w02 * M[1] + w12 * M[4] + w22 * M[7]); vM[8] = 2.f * (w02 * M[2] + w12 * M[5] + w22 * M[8]);
    const float* R = p.R;
    v_scale[0] += vM[0] * R[0] + vM[3] * R[3] + vM[6] * R[6];
    v_scale[1] += vM[1] * R[1] + vM[4] * R[4] + vM[7] * R[7];
    v_scale[2] += vM[2] * R[2] + vM[5] * R[5] + vM[8] * R[8];
    float vR[9];
    for (int i = 0; i < 3; ++i) { vR[3 * i] = vM[3 * i] * scale[0]; vR[3 * i + 1] = vM[3 * i + 1] * scale[1]; vR[3 * i + 2] = vM[3 * i + 2] * scale[2]; }
    const float w = p.qw, x = p.qx, y = p.qy, z = p.qz;
    const float vq0 = 2.f * (-z * vR[1] + y * vR[2] + z * vR[3] - x * vR[5] - y * vR[6] + x * vR[7]);
    const float vq1 = 2.f * (y * vR[1] + z * vR[2] + y * vR[3] - 2.f * x * vR[4] - w * vR[5] + z * vR[6] + w * vR[7] - 2.f * x * vR[8]);
    const float vq2 = 2.f * (-2.f * y * vR[0] + x * vR[1] + w * vR[2] + x * vR[3] + z * vR[5] - w * vR[6] + z * vR[7] - 2.f * y * vR[8]);
    const float vq3 = 2.f * (-2.f * z * vR[0] - w * vR[1] + x * vR[2] + w * vR[3] - 2.f * z * vR[4] + y * vR[5] + x * vR[6] + y * vR[7]);
    const float dot = w * vq0 + x * vq1 + y * vq2 + z * vq3;
    v_quat[0] += (vq0 - w * dot) * p.qinv; v_quat[1] += (vq1 - x * dot) * p.qinv;
    v_quat[2] += (vq2 - y * dot) * p.qinv; v_quat[3] += (vq3 - z * dot) * p.qinv;
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
    cam.limx = kFovClamp * (0.5f * (float)W / cam.fx);
    cam.limy = kFovClamp * (0.5f * (float)H / cam.fy);
}

}  // namespace gs
