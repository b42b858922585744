// A host that is NOT Python: plain C++ + the HIP runtime calls libsnag_hip.so through include/snag_hip.h, and checks the scores of
// snag_score_margin_fwd against a scalar restatement of the reference formula (model/MSNEA.py:110-122: rows L2-normalised with
// eps 1e-12, s = |h + r - t|^2, loss = sum relu(margin + pos - neg)).  Built and run by tests/test_native_gpu.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "snag_hip.h"

#define CHECK(x)                                                             \
    do {                                                                     \
        if ((x) != hipSuccess) {                                             \
            std::printf("HIP error at %s:%d\n", __FILE__, __LINE__);         \
            return 10;                                                       \
        }                                                                    \
    } while (0)

static void normalise(const float* x, int d, double* out) {
    double ss = 0;
    for (int c = 0; c < d; ++c) ss += (double)x[c] * x[c];
    const double den = std::fmax(std::sqrt(ss), 1e-12);
    for (int c = 0; c < d; ++c) out[c] = x[c] / den;
}

int main() {
    const int N = 300, R = 7, D = 128, B = 37, K = 9;
    const float margin = 1.5f;
    std::vector<float> ent((size_t)N * D), rel((size_t)R * D);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : ent) v = rnd();
    for (auto& v : rel) v = rnd();
    std::vector<int32_t> h(B), r(B), t(B), neg((size_t)B * K);
    for (int b = 0; b < B; ++b) {
        h[b] = (s = s * 1664525u + 1013904223u) % N; r[b] = (s = s * 1664525u + 1013904223u) % R; t[b] = (s = s * 1664525u + 1013904223u) % N;
        for (int k = 0; k < K; ++k) neg[(size_t)b * K + k] = (s = s * 1664525u + 1013904223u) % N;
    }
    float *d_ent, *d_rel, *d_pos, *d_neg, *d_part;
    int32_t *d_h, *d_r, *d_t, *d_n;
    CHECK(hipMalloc(&d_ent, ent.size() * 4)); CHECK(hipMalloc(&d_rel, rel.size() * 4));
    CHECK(hipMalloc(&d_pos, B * 4)); CHECK(hipMalloc(&d_neg, (size_t)B * K * 4)); CHECK(hipMalloc(&d_part, B * 4));
    CHECK(hipMalloc(&d_h, B * 4)); CHECK(hipMalloc(&d_r, B * 4)); CHECK(hipMalloc(&d_t, B * 4)); CHECK(hipMalloc(&d_n, (size_t)B * K * 4));
    CHECK(hipMemcpy(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rel, rel.data(), rel.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_h, h.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_r, r.data(), B * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_t, t.data(), B * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_n, neg.data(), (size_t)B * K * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const int rc = snag_score_margin_fwd(d_ent, D, N, d_rel, R, D, d_h, d_r, d_t, d_n, 4, B, K, margin, d_pos, d_neg, d_part, st);
    if (rc != SNAG_OK) { std::printf("snag_score_margin_fwd refused: %s\n", snag_last_error()); return 11; }
    CHECK(hipStreamSynchronize(st));
    std::vector<float> pos(B), ng((size_t)B * K), part(B);
    CHECK(hipMemcpy(pos.data(), d_pos, B * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ng.data(), d_neg, (size_t)B * K * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(part.data(), d_part, B * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    std::vector<double> hn(D), rn(D), tn(D);
    for (int b = 0; b < B; ++b) {
        normalise(&ent[(size_t)h[b] * D], D, hn.data()); normalise(&rel[(size_t)r[b] * D], D, rn.data()); normalise(&ent[(size_t)t[b] * D], D, tn.data());
        double p = 0;
        for (int c = 0; c < D; ++c) { const double v = hn[c] + rn[c] - tn[c]; p += v * v; }
        worst = std::fmax(worst, std::fabs(p - pos[b]));
        double lp = 0;
        for (int k = 0; k < K; ++k) {
            normalise(&ent[(size_t)neg[(size_t)b * K + k] * D], D, tn.data());
            double q = 0;
            for (int c = 0; c < D; ++c) { const double v = hn[c] + rn[c] - tn[c]; q += v * v; }
            worst = std::fmax(worst, std::fabs(q - ng[(size_t)b * K + k]));
            lp += std::fmax(0.0, margin + p - q);
        }
        worst = std::fmax(worst, std::fabs(lp - part[b]) / std::fmax(1.0, lp));
    }
    std::printf("max abs err %.3e\n", worst);
    return worst < 1e-4 ? 0 : 12;          // north_star: scores within 1e-4
}
