// san_driver.cpp -- ASan / UBSan leg for the per-Gaussian device math (easy_gaussian_splatting_amd/csrc/gs_math.h) in its
// host build: the exact source the gfx950 kernels compile, driven over random scenes with exact-size heap buffers.  CPU only
// (sanitizers never run on the GPU pool); built and run by tests/test_sanitizers.py.  TEST INFRASTRUCTURE ONLY.
#include "hostmath.cpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

static double run_case(int C, int N, int W, int H, int degree, int K) {
    std::vector<float> means(3 * (size_t)N), quats(4 * (size_t)N), scales(3 * (size_t)N), shs(3 * (size_t)N * K), viewmats(16 * (size_t)C, 0.f), Ks(9 * (size_t)C, 0.f);
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < 3; ++k) means[3 * n + k] = (float)(3.0 * (urand() - 0.5) * (k == 2 ? 4.0 : 2.0));
        for (int k = 0; k < 4; ++k) quats[4 * n + k] = (float)nrand();
        for (int k = 0; k < 3; ++k) scales[3 * n + k] = (float)std::exp(std::log(0.005) + urand() * (std::log(0.8) - std::log(0.005)));   // needles included
        for (int k = 0; k < 3 * K; ++k) shs[(size_t)n * 3 * K + k] = (float)(k < 3 ? 3.5 * (urand() - 0.5) : 0.1 * nrand());
    }
    for (int c = 0; c < C; ++c) {
        const double a = 0.7 * c;
        float* V = viewmats.data() + 16 * c;
        V[0] = (float)std::cos(a); V[2] = (float)-std::sin(a); V[5] = 1; V[8] = (float)std::sin(a); V[10] = (float)std::cos(a); V[11] = 3.f; V[15] = 1;
        float* Kc = Ks.data() + 9 * c;
        Kc[0] = Kc[4] = (float)(0.5 * W / 0.5773502691896257); Kc[2] = 0.5f * W; Kc[5] = 0.5f * H; Kc[8] = 1;
    }
    const size_t CN = (size_t)C * N;
    std::vector<int32_t> radii(CN), tpg(CN), rect_a(4 * CN), rect_b(4 * CN);
    std::vector<float> m2(2 * CN), dep(CN), con(3 * CN), col(3 * CN), ex(CN), ey(CN), opac(CN), cxx(CN), cyy(CN), mx(CN), my(CN);
    hm_forward(C, N, K, degree, means.data(), quats.data(), scales.data(), shs.data(), viewmats.data(), Ks.data(), W, H, 16, 0.3f, 0.01f,
               1e10f, 0.f, radii.data(), m2.data(), dep.data(), con.data(), col.data(), tpg.data());
    for (size_t f = 0; f < CN; ++f) {
        opac[f] = (float)(1.0 / (1.0 + std::exp(-1.5 * nrand())));
        // covariance diagonal from the conic (cxx = C / det(conic), cyy = A / det(conic)); culled splats keep zeros
        const double A = con[3 * f], B = con[3 * f + 1], Cc = con[3 * f + 2], det = A * Cc - B * B;
        cxx[f] = det > 0 ? (float)(Cc / det) : 0.f;
        cyy[f] = det > 0 ? (float)(A / det) : 0.f;
        mx[f] = m2[2 * f]; my[f] = m2[2 * f + 1];
    }
    hm_extents((int)CN, opac.data(), cxx.data(), cyy.data(), mx.data(), my.data(), radii.data(), W, H, 16, ex.data(), ey.data(), rect_a.data(),
               rect_b.data());
    std::vector<float> v_m2(2 * CN), v_cn(3 * CN), v_col(3 * CN), v_means(3 * (size_t)N), v_quats(4 * (size_t)N), v_scales(3 * (size_t)N), v_shs(3 * (size_t)N * K);
    for (auto* v : {&v_m2, &v_cn, &v_col})
        for (auto& x : *v) x = (float)nrand();
    hm_backward(C, N, K, degree, means.data(), quats.data(), scales.data(), shs.data(), viewmats.data(), Ks.data(), W, H, 0.3f, 0.01f, 1e10f,
                radii.data(), col.data(), v_m2.data(), v_cn.data(), v_col.data(), v_means.data(), v_quats.data(), v_scales.data(), v_shs.data(), 0);
    hm_backward(C, N, K, degree, means.data(), quats.data(), scales.data(), shs.data(), viewmats.data(), Ks.data(), W, H, 0.3f, 0.01f, 1e10f,
                radii.data(), col.data(), v_m2.data(), v_cn.data(), v_col.data(), v_means.data(), v_quats.data(), v_scales.data(), v_shs.data(), 1);
    double sum = 0;
    for (size_t f = 0; f < CN; ++f) sum += radii[f] + tpg[f];
    for (float x : v_means) sum += std::fabs((double)x);
    for (float x : v_quats) sum += std::fabs((double)x);
    for (float x : v_shs) sum += std::fabs((double)x);
    if (!(sum == sum)) { std::fprintf(stderr, "NaN checksum\n"); std::exit(3); }
    return sum;
}

int main() {
    const int cases[][6] = {{1, 0, 33, 17, 0, 1}, {1, 1, 16, 16, 0, 1}, {1, 3000, 70, 45, 3, 16}, {3, 2000, 64, 48, 2, 16}, {2, 1000, 1920, 1080, 1, 4}};
    for (const auto& c : cases)
        std::printf("C=%d N=%d %dx%d SH%d K=%d -> checksum %.9g\n", c[0], c[1], c[2], c[3], c[4], c[5], run_case(c[0], c[1], c[2], c[3], c[4], c[5]));
    std::puts("sanitizer leg ok");
    return 0;
}
