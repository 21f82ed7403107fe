// The lane emulation of the wave engine run under AddressSanitizer and
// UndefinedBehaviorSanitizer (host code only): every mode of the iteration on small ragged
// batches at both transform sizes, the forward and the inverse transform.  Built and run by
// tests/test_engine_emulation.py::test_emulation_under_sanitizers; prints "ok" on success.
#include <cstdio>
#include <cmath>
#include <vector>

#include "emu.cpp"

static int run_case(int n_fft, int hop, const std::vector<int>& lens, int fpr) {
    const int F = n_fft / 2 + 1, B = (int)lens.size();
    int T = 0;
    for (int t : lens) T = std::max(T, t);
    std::vector<float> S((size_t)B * T * F), ph((size_t)B * T * F * 2), y((size_t)B * hop * (T - 1) + 8), sc(B * 8);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto& v : S) v = rnd() * rnd();
    for (size_t i = 0; i < ph.size(); i += 2) {
        const float a = 6.2831853f * rnd();
        ph[i] = std::cos(a);
        ph[i + 1] = std::sin(a);
    }
    int bad = 0;
    for (int mode = 0; mode <= 6; ++mode) {
        if (mode >= 5 && n_fft != 1024) continue;
        const int rc = emu_griffinlim(n_fft, hop, n_fft, S.data(), ph.data(), 7, lens.data(), B, T, 3, 0.99f,
                                      mode == 2 ? 0 : fpr, y.data(), (int64_t)hop * (T - 1), mode == 2 ? nullptr : sc.data(),
                                      mode);
        if (rc != 0) { std::printf("mode %d n_fft %d: rc %d\n", mode, n_fft, rc); ++bad; }
        for (float v : y) if (!std::isfinite(v)) { std::printf("mode %d: non-finite output\n", mode); ++bad; break; }
    }
    std::vector<float> D((size_t)B * T * F * 2);
    bad += emu_stft(n_fft, hop, n_fft, y.data(), (int64_t)hop * (T - 1), lens.data(), B, T, fpr, D.data()) != 0;
    bad += emu_istft(n_fft, hop, n_fft, D.data(), lens.data(), B, T, fpr, y.data(), (int64_t)hop * (T - 1)) != 0;
    return bad;
}

int main() {
    int bad = 0;
    bad += run_case(1024, 256, {33, 7, 2, 20}, 5);
    bad += run_case(1024, 128, {19, 9}, 8);
    bad += run_case(2048, 256, {24, 3, 11}, 9);
    bad += run_case(2048, 512, {9, 5}, 4);
    std::printf(bad ? "failed\n" : "ok\n");
    return bad ? 1 : 0;
}
