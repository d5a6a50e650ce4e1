// Unit test of ptrs_plan_bands (include/ptrs.h; SURVEY 8e row bands) -- runs on the CPU, no GPU call.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/ptrs.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

static std::vector<int32_t> plan(int h, unsigned n, const std::vector<float> *cost = nullptr) {
    std::vector<int32_t> b(n + 1, -1);
    CHECK(ptrs_plan_bands(h, n, cost ? cost->data() : nullptr, b.data()) == PTRS_OK);
    return b;
}
static void invariants(const std::vector<int32_t> &b, int h) {
    CHECK(b.front() == 0 && b.back() == h);
    for (size_t k = 0; k + 1 < b.size(); ++k) CHECK(b[k] <= b[k + 1]);
    // every band has a row while rows remain
    const int n = (int)b.size() - 1;
    if (h >= n) for (int k = 0; k < n; ++k) CHECK(b[k + 1] > b[k]);
}

int main() {
    { auto b = plan(1024, 8); invariants(b, 1024); for (int k = 0; k <= 8; ++k) CHECK(b[k] == 128 * k); }
    { auto b = plan(10, 3); invariants(b, 10); CHECK(b[1] == 4 && b[2] == 7); }               // earlier bands take the remainder
    { auto b = plan(5, 8); invariants(b, 5); CHECK(b[5] == 5 && b[8] == 5); }                  // more devices than rows: empty bands at the end
    { auto b = plan(2160, 8); invariants(b, 2160); for (int k = 0; k < 8; ++k) CHECK(b[k + 1] - b[k] == 270); }
    { // cost-weighted: the top half costs three times the bottom half
        std::vector<float> c(100, 1.0f); for (int y = 0; y < 50; ++y) c[y] = 3.0f;
        auto b = plan(100, 4, &c); invariants(b, 100);
        double tot = 200.0; for (int k = 0; k < 4; ++k) { double s = 0; for (int y = b[k]; y < b[k + 1]; ++y) s += c[y]; CHECK(s > tot / 4 - 3.0 && s < tot / 4 + 3.0); }
    }
    { // all the cost in one row: bands still cover the film and keep a row each
        std::vector<float> c(16, 0.0f); c[3] = 1.0f;
        auto b = plan(16, 4, &c); invariants(b, 16);
    }
    { // zero / negative costs fall back to equal rows
        std::vector<float> c(64, 0.0f); auto b = plan(64, 4, &c); for (int k = 0; k <= 4; ++k) CHECK(b[k] == 16 * k);
        std::vector<float> d(64, -1.0f); auto e = plan(64, 4, &d); for (int k = 0; k <= 4; ++k) CHECK(e[k] == 16 * k);
    }
    { // a smooth ramp: the balance is within one row's cost
        std::vector<float> c(720); double tot = 0; for (int y = 0; y < 720; ++y) { c[y] = 1.0f + (float)y / 100.0f; tot += c[y]; }
        auto b = plan(720, 8, &c); invariants(b, 720);
        for (int k = 0; k < 8; ++k) { double s = 0; for (int y = b[k]; y < b[k + 1]; ++y) s += c[y]; CHECK(s > tot / 8 - 9.0 && s < tot / 8 + 9.0); }
    }
    { int32_t out[2]; CHECK(ptrs_plan_bands(0, 1, nullptr, out) == PTRS_ERR_INVALID); CHECK(ptrs_plan_bands(8, 0, nullptr, out) == PTRS_ERR_INVALID); CHECK(ptrs_plan_bands(8, 1, nullptr, nullptr) == PTRS_ERR_INVALID); }
    if (fails) { std::fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
    std::printf("band planning ok\n");
    return 0;
}
