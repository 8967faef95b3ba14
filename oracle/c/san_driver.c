/* san_driver.c -- AddressSanitizer / UBSan leg of the CPU checker (SURVEY.md section 5 "Race detection": sanitizers run on
 * the CPU build only, never on the GPU).  TEST INFRASTRUCTURE ONLY.
 *
 * Drives every entry point of gs_oracle.c (the restatement of the seam, /root/reference/model/gaussian.py:353-367)
 * through a forward + backward on small deterministic scenes sized to hit the edges: ragged image sizes (not a
 * multiple of the tile), Gaussians behind the camera / off screen, N = 0, N = 1, several cameras, SH degree 0..3,
 * no background.  Built by `make -C oracle/c asan` with -fsanitize=address,undefined -fno-sanitize-recover; any heap
 * overflow, use-after-free, misaligned access, signed overflow or invalid shift aborts with a non-zero status.
 * Prints one checksum line per case so that the run is also a smoke test of the sanitized build.                     */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef GSO_DOUBLE
typedef double real;
#else
typedef float real;
#endif

int gso_project_fwd(int, int, const real*, const real*, const real*, const real*, const real*, int, int, real, real, real,
                    real, int32_t*, real*, real*, real*);
int gso_sh_fwd(int, int, int, int, const real*, const real*, const real*, const int32_t*, real*);
int64_t gso_isect_count(int, int, const real*, const int32_t*, int, int, int, int32_t*);
int gso_isect_build(int, int, const real*, const int32_t*, const real*, int, int, int, int64_t, int64_t*, int32_t*, int32_t*);
int gso_blend_fwd(int, int, int, int, int, const real*, const real*, const real*, const real*, const real*, const int32_t*,
                  const int32_t*, int64_t, real*, real*, int32_t*);
int gso_blend_margin(int, int, int, int, int, const real*, const real*, const real*, const int32_t*, const int32_t*, int64_t,
                     const real*, const real*, real*);
int gso_blend_bwd(int, int, int, int, int, const real*, const real*, const real*, const real*, const real*, const int32_t*,
                  const int32_t*, int64_t, const real*, const int32_t*, const real*, const real*, real*, real*, real*, real*,
                  real*);
int gso_sh_bwd(int, int, int, int, const real*, const real*, const real*, const int32_t*, const real*, const real*, real*, real*);
int gso_project_bwd(int, int, const real*, const real*, const real*, const real*, const real*, int, int, real, real, real,
                    const int32_t*, const real*, const real*, const real*, real*, real*, real*);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {   /* xorshift64*, uniform in [0,1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double nrand(void) { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

/* exact-size allocations (no slack), so that an off-by-one in the oracle lands in a red zone */
static void* xalloc(size_t n, size_t sz) { void* p = calloc(n ? n : 1, sz); if (!p) abort(); return p; }

static double run_case(int C, int N, int W, int H, int degree, int K, int with_bg, int with_valpha) {
    const int tile = 16, tw = (W + tile - 1) / tile, th = (H + tile - 1) / tile;
    real *means = xalloc(3 * (size_t)N, sizeof(real)), *quats = xalloc(4 * (size_t)N, sizeof(real));
    real *scales = xalloc(3 * (size_t)N, sizeof(real)), *opac1 = xalloc(N, sizeof(real));
    real *shs = xalloc(3 * (size_t)N * K, sizeof(real)), *viewmats = xalloc(16 * (size_t)C, sizeof(real)), *Ks = xalloc(9 * (size_t)C, sizeof(real));
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < 3; ++k) means[3 * n + k] = (real)(3.0 * (urand() - 0.5) * (k == 2 ? 4.0 : 2.0));   /* some behind the camera */
        for (int k = 0; k < 4; ++k) quats[4 * n + k] = (real)nrand();
        for (int k = 0; k < 3; ++k) scales[3 * n + k] = (real)exp(log(0.02) + urand() * (log(0.6) - log(0.02)));
        opac1[n] = (real)(1.0 / (1.0 + exp(-1.5 * nrand())));
        for (int k = 0; k < 3 * K; ++k) shs[(size_t)n * 3 * K + k] = (real)(k < 3 ? 3.5 * (urand() - 0.5) : 0.1 * nrand());
    }
    for (int c = 0; c < C; ++c) {
        const double a = 0.7 * c;
        real* V = viewmats + 16 * c;
        memset(V, 0, 16 * sizeof(real));
        V[0] = (real)cos(a); V[2] = (real)-sin(a); V[5] = 1; V[8] = (real)sin(a); V[10] = (real)cos(a); V[11] = (real)3.0; V[15] = 1;
        real* Kc = Ks + 9 * c;
        memset(Kc, 0, 9 * sizeof(real));
        Kc[0] = Kc[4] = (real)(0.5 * W / 0.5773502691896257); Kc[2] = (real)(0.5 * W); Kc[5] = (real)(0.5 * H); Kc[8] = 1;
    }
    const size_t CN = (size_t)C * N, P = (size_t)C * H * W;
    int32_t *radii = xalloc(CN, 4), *tpg = xalloc(CN, 4);
    real *m2 = xalloc(2 * CN, sizeof(real)), *dep = xalloc(CN, sizeof(real)), *con = xalloc(3 * CN, sizeof(real));
    real *cols = xalloc(3 * CN, sizeof(real)), *opac = xalloc(CN, sizeof(real));
    for (int c = 0; c < C; ++c) memcpy(opac + (size_t)c * N, opac1, (size_t)N * sizeof(real));
    gso_project_fwd(C, N, means, quats, scales, viewmats, Ks, W, H, (real)0.3, (real)0.01, (real)1e10, (real)0.0, radii, m2, dep, con);
    if (gso_sh_fwd(C, N, K, degree, means, viewmats, shs, radii, cols) != 0) abort();
    const int64_t I = gso_isect_count(C, N, m2, radii, tile, tw, th, tpg);
    int64_t* isect_ids = xalloc((size_t)I, 8);
    int32_t *flat = xalloc((size_t)I, 4), *offs = xalloc((size_t)C * tw * th, 4);
    if (gso_isect_build(C, N, m2, radii, dep, tile, tw, th, I, isect_ids, flat, offs) != 0) abort();
    real* bg = with_bg ? xalloc(3 * (size_t)C, sizeof(real)) : NULL;
    if (bg) for (int k = 0; k < 3 * C; ++k) bg[k] = (real)urand();
    real *img = xalloc(3 * P, sizeof(real)), *alpha = xalloc(P, sizeof(real)), *margin = xalloc(P, sizeof(real));
    int32_t* last = xalloc(P, 4);
    gso_blend_fwd(C, N, W, H, tile, m2, con, cols, opac, bg, offs, flat, I, img, alpha, last);
    gso_blend_margin(C, N, W, H, tile, m2, con, opac, offs, flat, I, NULL, NULL, margin);
    gso_blend_margin(C, N, W, H, tile, m2, con, opac, offs, flat, I, m2, con, margin);
    real *vc = xalloc(3 * P, sizeof(real)), *va = with_valpha ? xalloc(P, sizeof(real)) : NULL;
    for (size_t i = 0; i < 3 * P; ++i) vc[i] = (real)nrand();
    if (va) for (size_t i = 0; i < P; ++i) va[i] = (real)nrand();
    real *v_m2 = xalloc(2 * CN, sizeof(real)), *v_abs = xalloc(2 * CN, sizeof(real)), *v_cn = xalloc(3 * CN, sizeof(real));
    real *v_rgb = xalloc(3 * CN, sizeof(real)), *v_op = xalloc(CN, sizeof(real));
    gso_blend_bwd(C, N, W, H, tile, m2, con, cols, opac, bg, offs, flat, I, alpha, last, vc, va, v_m2, v_abs, v_cn, v_rgb, v_op);
    real *v_means = xalloc(3 * (size_t)N, sizeof(real)), *v_quats = xalloc(4 * (size_t)N, sizeof(real)), *v_scales = xalloc(3 * (size_t)N, sizeof(real));
    real* v_shs = xalloc(3 * (size_t)N * K, sizeof(real));
    if (gso_sh_bwd(C, N, K, degree, means, viewmats, shs, radii, cols, v_rgb, v_shs, v_means) != 0) abort();
    gso_project_bwd(C, N, means, quats, scales, viewmats, Ks, W, H, (real)0.3, (real)0.01, (real)1e10, radii, v_m2, NULL, v_cn, v_means,
                    v_quats, v_scales);
    double sum = (double)I;
    for (size_t i = 0; i < 3 * P; ++i) sum += (double)img[i];
    for (size_t i = 0; i < 3 * (size_t)N; ++i) sum += fabs((double)v_means[i]) + fabs((double)v_scales[i]);
    for (size_t i = 0; i < 4 * (size_t)N; ++i) sum += fabs((double)v_quats[i]);
    for (size_t i = 0; i < 3 * (size_t)N * K; ++i) sum += fabs((double)v_shs[i]);
    void* all[] = {means, quats, scales, opac1, shs, viewmats, Ks, radii, tpg, m2, dep, con, cols, opac, isect_ids, flat, offs, bg,
                   img, alpha, margin, last, vc, va, v_m2, v_abs, v_cn, v_rgb, v_op, v_means, v_quats, v_scales, v_shs};
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i]);
    if (!(sum == sum)) { fprintf(stderr, "NaN checksum\n"); exit(3); }
    return sum;
}

int main(void) {
    /*            C   N    W    H  deg  K  bg  v_alpha */
    const int cases[][8] = {{1, 0, 33, 17, 0, 1, 1, 0},      /* empty scene */
                            {1, 1, 16, 16, 0, 1, 0, 1},      /* one Gaussian, one tile, no background */
                            {1, 300, 70, 45, 3, 16, 1, 1},   /* ragged size */
                            {3, 500, 64, 48, 2, 16, 1, 0},   /* K > (deg+1)^2, three cameras */
                            {2, 200, 129, 31, 1, 4, 0, 0},   /* one-row-of-tiles image */
                            {1, 2000, 96, 96, 3, 16, 1, 1}};
    for (size_t i = 0; i < sizeof(cases) / sizeof(cases[0]); ++i) {
        const int* c = cases[i];
        printf("case %zu: C=%d N=%d %dx%d SH%d K=%d -> checksum %.9g\n", i, c[0], c[1], c[2], c[3], c[4], c[5],
               run_case(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]));
    }
    puts("sanitizer leg ok");
    return 0;
}
