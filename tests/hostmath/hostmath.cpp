// Host (g++) build of the per-Gaussian device math in easy_gaussian_splatting_amd/csrc/gs_math.h.
// TEST-ONLY: lets the CPU test-suite check the exact source the gfx950 kernels compile against the
// oracle without a GPU.  It is never loaded by the product package.
#include "../../easy_gaussian_splatting_amd/csrc/gs_math.h"
#include <cstring>

extern "C" {

int hm_forward(int C, int N, int K, int degree, const float* means, const float* quats,
               const float* scales, const float* shs, const float* viewmats, const float* Ks, int W,
               int H, int tile, float eps2d, float near_p, float far_p, float radius_clip,
               int32_t* radii, float* means2d, float* depths, float* conics, float* colors,
               int32_t* tiles_per_gauss) {
    const int tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    for (int c = 0; c < C; ++c) {
        gs::Camera cam;
        gs::make_camera(viewmats + 16 * c, Ks + 9 * c, W, H, cam);
        for (int n = 0; n < N; ++n) {
            const long f = (long)c * N + n;
            gs::Splat2D s = gs::project_gaussian(means + 3 * n, quats + 4 * n, scales + 3 * n, cam, W, H,
                                                 eps2d, near_p, far_p, radius_clip, tile, tw, th);
            radii[f] = s.radius; means2d[2 * f] = s.mx; means2d[2 * f + 1] = s.my; depths[f] = s.depth;
            conics[3 * f] = s.A; conics[3 * f + 1] = s.B; conics[3 * f + 2] = s.C;
            float rgb[3] = {0.5f, 0.5f, 0.5f};
            int cnt = 0;
            if (s.radius > 0) {
                float ux, uy, uz;
                gs::view_dir(means + 3 * n, cam, ux, uy, uz);
                gs::sh_to_rgb(degree, shs + (long)n * K * 3, ux, uy, uz, rgb);
                cnt = (s.x1 - s.x0) * (s.y1 - s.y0);
            }
            colors[3 * f] = rgb[0]; colors[3 * f + 1] = rgb[1]; colors[3 * f + 2] = rgb[2];
            tiles_per_gauss[f] = cnt;
        }
    }
    return 0;
}

int hm_backward(int C, int N, int K, int degree, const float* means, const float* quats,
                const float* scales, const float* shs, const float* viewmats, const float* Ks, int W,
                int H, float eps2d, float near_p, float far_p, const int32_t* radii,
                const float* colors, const float* v_means2d, const float* v_conics,
                const float* v_colors, float* v_means, float* v_quats, float* v_scales, float* v_shs, int use_jac) {
    if (N <= 0) return 0;   // (memset on the null data() of an empty buffer is undefined: found by the UBSan leg)
    std::memset(v_means, 0, sizeof(float) * 3 * N);
    std::memset(v_quats, 0, sizeof(float) * 4 * N);
    std::memset(v_scales, 0, sizeof(float) * 3 * N);
    std::memset(v_shs, 0, sizeof(float) * 3 * (size_t)K * N);
    for (int c = 0; c < C; ++c) {
        gs::Camera cam;
        gs::make_camera(viewmats + 16 * c, Ks + 9 * c, W, H, cam);
        for (int n = 0; n < N; ++n) {
            const long f = (long)c * N + n;
            if (radii[f] <= 0) continue;
            gs::ProjChain p;
            if (!gs::project_chain<gs::preal>(means + 3 * n, quats + 4 * n, scales + 3 * n, cam, eps2d, near_p, far_p, p)) continue;
            float ux, uy, uz;
            const float dn = gs::view_dir(means + 3 * n, cam, ux, uy, uz);
            if (use_jac) {   // the coefficient-free backward: direction Jacobian from the forward (gs_project_fwd's sh_jac)
                float G[12], row[48];
                gs::sh_dir_jacobian(degree, shs + (long)n * K * 3, ux, uy, uz, G);
                gs::sh_vjp_jac(degree, G, colors + 3 * f, v_colors + 3 * f, ux, uy, uz, dn, row, v_means + 3 * n);
                for (int o = 0; o < 3 * (degree + 1) * (degree + 1); ++o) v_shs[(long)n * K * 3 + o] += row[o];
            } else {
                gs::sh_vjp(degree, shs + (long)n * K * 3, colors + 3 * f, v_colors + 3 * f, ux, uy, uz, dn,
                           v_shs + (long)n * K * 3, v_means + 3 * n, true);
            }
            gs::project_vjp<gs::preal>(scales + 3 * n, cam, p, v_means2d[2 * f], v_means2d[2 * f + 1], v_conics[3 * f],
                            v_conics[3 * f + 1], v_conics[3 * f + 2], 0.f, v_means + 3 * n, v_quats + 4 * n,
                            v_scales + 3 * n);
        }
    }
    return 0;
}
}

// opacity-aware extents + tight tile rectangle (same source the projection kernel uses)
extern "C" int hm_extents(int n, const float* opac, const float* cxx, const float* cyy, const float* mx,
                          const float* my, const int32_t* radius, int W, int H, int tile, float* ex, float* ey,
                          int32_t* rect_gsplat, int32_t* rect_tight) {
    const int tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    for (int i = 0; i < n; ++i) {
        gs::alpha_extent(opac[i], cxx[i], cyy[i], ex[i], ey[i]);
        int x0, x1, y0, y1;
        gs::tile_rect<float>(mx[i], my[i], radius[i], tile, tw, th, x0, x1, y0, y1);
        rect_gsplat[4 * i] = x0; rect_gsplat[4 * i + 1] = x1; rect_gsplat[4 * i + 2] = y0; rect_gsplat[4 * i + 3] = y1;
        gs::tile_rect_tight(mx[i], my[i], ex[i], ey[i], W, H, tile, x0, x1, y0, y1);
        rect_tight[4 * i] = x0; rect_tight[4 * i + 1] = x1; rect_tight[4 * i + 2] = y0; rect_tight[4 * i + 3] = y1;
    }
    return 0;
}
