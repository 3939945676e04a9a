// The reference crate's Rust unit tests and doctests, re-created against the C++ mirror of its API
// (include/interpn_hip.hpp -> C ABI -> HIP kernels).  Each test names the Rust test it restates;
// grids, observation points, fields and tolerances are the reference's (the rectilinear tests'
// grid noise comes from a splitmix64 stream here: the reference draws it from rand's StdRng,
// whose stream is not reproducible outside Rust; magnitude and acceptance rule are the same).
//
// Build (tests/test_cpp_mirror.py does this):
//   g++ -std=c++17 -O1 -Iinclude tests/cpp/reference_tests.cpp -Linterpn_amd -linterpn_hip
//       -Wl,-rpath,$PWD/interpn_amd -o reference_tests
// Prints one line per test and "ALL PASSED" / exit code 0 when every assertion held.
// Needs a GPU: the library has no CPU path.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include "interpn_hip.hpp"

using namespace interpn_hip;
using utils::linspace;
using utils::meshgrid;

static int g_failures = 0;
static int g_checks = 0;
#define EXPECT(cond)                                                                   \
  do {                                                                                 \
    ++g_checks;                                                                        \
    if (!(cond)) {                                                                     \
      if (g_failures < 20) std::printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      ++g_failures;                                                                    \
    }                                                                                  \
  } while (0)

static void run(const char* name, const std::function<void()>& f) {
  const int before = g_failures;
  f();
  std::printf("%s %s\n", g_failures == before ? "PASS" : "FAIL", name);
}

// uniform [0, 1) stream standing in for the reference's `randn(&mut rng, n)` (src/testing.rs:18)
struct Rng {
  std::uint64_t s;
  double next() {
    std::uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }
};

typedef std::vector<std::vector<double>> Table;

static std::vector<const std::vector<double>*> refs(const Table& t) {
  std::vector<const std::vector<double>*> r;
  for (const auto& x : t) r.push_back(&x);
  return r;
}

// `gridobs_t`: one array per dimension from a list of points
static Table transpose(const Table& pts, std::size_t ndims) {
  Table t(ndims, std::vector<double>(pts.size()));
  for (std::size_t p = 0; p < pts.size(); ++p)
    for (std::size_t d = 0; d < ndims; ++d) t[d][p] = pts[p][d];
  return t;
}

static std::vector<double> field(const Table& pts, const std::function<double(double)>& f) {
  std::vector<double> u(pts.size());
  for (std::size_t p = 0; p < pts.size(); ++p) {
    double v = 0.0;
    for (double x : pts[p]) v += f(x);
    u[p] = v;
  }
  return u;
}

// The grid of the extrapolation tests: axis i = linspace(-5 i, 5 (i + 1), n) [+ noise]
static Table test_axes(std::size_t ndims, std::size_t n, Rng* noise) {
  Table xs;
  for (std::size_t i = 0; i < ndims; ++i) {
    std::vector<double> x = linspace(-5.0 * (double)i, 5.0 * (double)(i + 1), n);
    if (noise) {
      for (double& xi : x) xi += (noise->next() - 0.5) / 10.0;
      for (std::size_t k = 0; k + 1 < x.size(); ++k) EXPECT(x[k + 1] > x[k]);
    }
    xs.push_back(x);
  }
  return xs;
}

static Table obs_axes(std::size_t ndims, double lo, double hi, std::size_t n) {
  Table xs;
  for (std::size_t i = 0; i < ndims; ++i) xs.push_back(linspace(lo * (double)i, hi * (double)(i + 1), n));
  return xs;
}

static double hat_func(double x) { return x <= 1.0 ? x : 2.0 - x; }

// ---------------------------------------------------------------------------------------------
// src/multilinear/regular.rs:436-477 test_interp_extrap_1d_to_6d and
// src/multilinear/regular_recursive.rs:402 test_interp_extrap_1d_to_8d (the dispatch function
// reaches the recursive arm for N = 7, 8)
static void multilinear_regular_extrap() {
  for (std::size_t n = 1; n <= 8; ++n) {
    const std::vector<std::size_t> dims(n, 2);
    const Table xs = test_axes(n, 2, nullptr);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    std::vector<double> starts, steps;
    for (const auto& x : xs) {
      starts.push_back(x[0]);
      steps.push_back(x[1] - x[0]);
    }
    const Table gridobs = meshgrid(refs(obs_axes(n, -7.0, 7.0, 3)));
    const Table gridobs_t = transpose(gridobs, n);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x; });
    std::vector<double> out(uobs.size(), 0.0);
    multilinear::regular::interpn<double>(dims, starts, steps, u, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-12);
    // interpn_alloc (regular.rs:124) gives the same vector
    const std::vector<double> out2 = multilinear::regular::interpn_alloc<double>(dims, starts, steps, u, slices(gridobs_t)).unwrap();
    EXPECT(out2 == out);
  }
}

// src/multilinear/regular.rs:481-495 test_interp_hat_func (assert_eq!: exact)
static void multilinear_regular_hat() {
  std::vector<double> y;
  for (int x = 0; x < 3; ++x) y.push_back(hat_func((double)x));
  const std::vector<double> obs = linspace(-2.0, 4.0, 100);
  auto interpolator = MultilinearRegular<double, 1>::new_({3}, {0.0}, {1.0}, y).unwrap();
  for (double x : obs) EXPECT(hat_func(x) == interpolator.interp_one({x}).unwrap());
  // the same points as one batch through `interp`
  std::vector<double> out(obs.size());
  interpolator.interp({Slice<double>(obs)}, out).unwrap();
  for (std::size_t i = 0; i < obs.size(); ++i) EXPECT(hat_func(obs[i]) == out[i]);
}

// src/multilinear/rectilinear.rs:380-407 test_interp_extrap_2d_small
static void multilinear_rectilinear_2d_small() {
  const std::size_t nx = 3, ny = 2;
  const std::vector<double> x = linspace(-1.0, 1.0, nx), y = {0.5, 0.6};
  const Table xy = meshgrid<double>({&x, &y});
  std::vector<double> z(nx * ny);
  for (std::size_t i = 0; i < nx * ny; ++i) z[i] = xy[i][0] + xy[i][1];
  const std::vector<double> xobs = linspace(-10.0, 10.0, 5), yobs = linspace(-10.0, 10.0, 5);
  const Table xyobs = meshgrid<double>({&xobs, &yobs});
  auto interpolator = MultilinearRectilinear<double, 2>::new_({Slice<double>(x), Slice<double>(y)}, z).unwrap();
  for (const auto& p : xyobs) {
    const double zii = interpolator.interp_one({p[0], p[1]}).unwrap();
    EXPECT(std::fabs((p[0] + p[1]) - zii) < 1e-12);
  }
}

// src/multilinear/rectilinear.rs:413-456 test_interp_extrap_1d_to_6d and
// rectilinear_recursive.rs:380 test_interp_extrap_1d_to_8d
static void multilinear_rectilinear_extrap() {
  Rng rng{0x2};
  for (std::size_t ndims = 1; ndims <= 8; ++ndims) {
    const Table xs = test_axes(ndims, 2, &rng);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 3)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x; });
    std::vector<double> out(uobs.size(), 0.0);
    multilinear::rectilinear::interpn<double>(slices(xs), u, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-12);
  }
}

// src/multilinear/rectilinear.rs:460-476 test_interp_hat_func
static void multilinear_rectilinear_hat() {
  const std::vector<double> x = {0.0, 1.0, 2.0};
  std::vector<double> y;
  for (double xi : x) y.push_back(hat_func(xi));
  const std::vector<double> obs = linspace(-2.0, 4.0, 100);
  auto interpolator = MultilinearRectilinear<double, 1>::new_({Slice<double>(x)}, y).unwrap();
  for (double xo : obs) EXPECT(hat_func(xo) == interpolator.interp_one({xo}).unwrap());
}

// ---------------------------------------------------------------------------------------------
// src/multicubic/regular.rs:634-676 test_interp_extrap_1d_to_4d_linear, and
// regular_recursive.rs:622 test_interp_extrap_1d_to_6d_linear (`for ndims in 1..6`: up to 5; the
// dispatch function takes the recursive arm for N = 5)
static void multicubic_regular_linear() {
  for (std::size_t ndims = 1; ndims <= 5; ++ndims) {
    const std::vector<std::size_t> dims(ndims, 4);
    const Table xs = test_axes(ndims, 4, nullptr);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    std::vector<double> starts, steps;
    for (const auto& x : xs) {
      starts.push_back(x[0]);
      steps.push_back(x[1] - x[0]);
    }
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 6)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x; });
    std::vector<double> out(uobs.size(), 0.0);
    for (bool linearize : {false, true}) {
      multicubic::regular::interpn<double>(dims, starts, steps, u, linearize, slices(gridobs_t), out).unwrap();
      for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-12);
    }
  }
}

// src/multicubic/regular.rs:680-730 test_interp_extrap_1d_to_4d_quadratic, and
// regular_recursive.rs:668 test_interp_extrap_1d_to_6d_quadratic (`1..6`: up to 5)
static void multicubic_regular_quadratic() {
  for (std::size_t ndims = 1; ndims <= 5; ++ndims) {
    const std::vector<std::size_t> dims(ndims, 4);
    const Table xs = test_axes(ndims, 4, nullptr);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x * x; });
    std::vector<double> starts, steps;
    for (const auto& x : xs) {
      starts.push_back(x[0]);
      steps.push_back(x[1] - x[0]);
    }
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 6)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x * x; });
    std::vector<double> out(uobs.size(), 0.0);
    multicubic::regular::interpn<double>(dims, starts, steps, u, false, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-10);
  }
}

// src/multicubic/regular.rs:737-792 test_interp_1d_to_3d_sine (`for ndims in 1..3`: 1 and 2)
static void multicubic_regular_sine() {
  for (std::size_t ndims = 1; ndims < 3; ++ndims) {
    const std::vector<std::size_t> dims(ndims, 10);
    const Table xs = test_axes(ndims, 10, nullptr);
    const auto f = [](double x) { return std::sin(x * 6.28 / 10.0); };
    const std::vector<double> u = field(meshgrid(refs(xs)), f);
    std::vector<double> starts, steps;
    for (const auto& x : xs) {
      starts.push_back(x[0]);
      steps.push_back(x[1] - x[0]);
    }
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -5.0, 5.0, 12)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, f);
    std::vector<double> out(uobs.size(), 0.0);
    multicubic::regular::interpn<double>(dims, starts, steps, u, false, slices(gridobs_t), out).unwrap();
    const double tol = 2e-2 * (double)ndims;
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < tol);
  }
}

// src/multicubic/rectilinear.rs:558-604 test_interp_extrap_1d_to_4d_linear (tolerance 1e-10 there)
// and rectilinear_recursive.rs:552 test_interp_extrap_1d_to_6d_linear (`1..=6`)
static void multicubic_rectilinear_linear() {
  Rng rng{0x3};
  for (std::size_t ndims = 1; ndims <= 6; ++ndims) {
    const Table xs = test_axes(ndims, 4, &rng);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 6)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x; });
    std::vector<double> out(uobs.size(), 0.0);
    for (bool linearize : {true, false}) {
      multicubic::rectilinear::interpn<double>(slices(xs), u, linearize, slices(gridobs_t), out).unwrap();
      for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-10);
    }
  }
}

// src/multicubic/rectilinear.rs:609-667 test_interp_extrap_1d_to_4d_quadratic (`1..4`: 1, 2, 3)
// and rectilinear_recursive.rs:603 test_interp_extrap_1d_to_6d_quadratic (`1..6`: up to 5)
static void multicubic_rectilinear_quadratic() {
  Rng rng{0x4};
  for (std::size_t ndims = 1; ndims < 6; ++ndims) {
    const Table xs = test_axes(ndims, 4, &rng);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x * x; });
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 6)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, [](double x) { return x * x; });
    std::vector<double> out(uobs.size(), 0.0);
    multicubic::rectilinear::interpn<double>(slices(xs), u, false, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < 1e-10);
  }
}

// src/multicubic/rectilinear.rs:674-736 test_interp_1d_to_3d_sine
static void multicubic_rectilinear_sine() {
  Rng rng{0x5};
  for (std::size_t ndims = 1; ndims < 3; ++ndims) {
    const Table xs = test_axes(ndims, 10, &rng);
    const auto f = [](double x) { return std::sin(x * 6.28 / 10.0); };
    const std::vector<double> u = field(meshgrid(refs(xs)), f);
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -5.0, 5.0, 12)));
    const Table gridobs_t = transpose(gridobs, ndims);
    const std::vector<double> uobs = field(gridobs, f);
    std::vector<double> out(uobs.size(), 0.0);
    multicubic::rectilinear::interpn<double>(slices(xs), u, false, slices(gridobs_t), out).unwrap();
    const double tol = 2e-2 * (double)ndims;
    for (std::size_t i = 0; i < uobs.size(); ++i) EXPECT(std::fabs(out[i] - uobs[i]) < tol);
  }
}

// ---------------------------------------------------------------------------------------------
// the reference tests' own helper, src/nearest/regular.rs:324-337
static std::size_t nearest_regular_index(double value, double start, double step, std::size_t dim) {
  const double floc = std::floor((value - start) / step);
  const long long n = (long long)dim;
  const long long dimmax = n - 2 > 0 ? n - 2 : 0;
  long long origin = (long long)floc;
  origin = origin < 0 ? 0 : (origin > dimmax ? dimmax : origin);
  const double index_zero = start + step * (double)origin;
  const double dt = (value - index_zero) / step;
  if (dt <= 0.5) return (std::size_t)origin;
  return (std::size_t)(origin + 1 < n - 1 ? origin + 1 : n - 1);
}

// src/nearest/rectilinear.rs:274-283
static std::size_t nearest_rectilinear_index(double value, const std::vector<double>& grid) {
  long long pp = 0;
  while (pp < (long long)grid.size() && grid[(std::size_t)pp] < value) ++pp;  // partition_point on a sorted grid
  const long long n = (long long)grid.size();
  const long long dimmax = n - 2 > 0 ? n - 2 : 0;
  long long origin = pp - 1;
  origin = origin < 0 ? 0 : (origin > dimmax ? dimmax : origin);
  const double x0 = grid[(std::size_t)origin], x1 = grid[(std::size_t)origin + 1];
  const double dt = (value - x0) / (x1 - x0);
  return (std::size_t)(dt <= 0.5 ? origin : origin + 1);
}

// src/nearest/regular.rs:343-398 test_interp_extrap_1d_to_6d
static void nearest_regular_extrap() {
  for (std::size_t n = 1; n <= 6; ++n) {
    const std::vector<std::size_t> dims(n, 2);
    const Table xs = test_axes(n, 2, nullptr);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    std::vector<double> starts, steps;
    for (const auto& x : xs) {
      starts.push_back(x[0]);
      steps.push_back(x[1] - x[0]);
    }
    const Table gridobs = meshgrid(refs(obs_axes(n, -7.0, 7.0, 3)));
    const Table gridobs_t = transpose(gridobs, n);
    std::vector<double> expected;
    for (const auto& p : gridobs) {
      double v = 0.0;
      for (std::size_t d = 0; d < n; ++d)
        v += starts[d] + steps[d] * (double)nearest_regular_index(p[d], starts[d], steps[d], dims[d]);
      expected.push_back(v);
    }
    std::vector<double> out(expected.size(), 0.0);
    nearest::regular::interpn<double>(dims, starts, steps, u, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < expected.size(); ++i) EXPECT(std::fabs(out[i] - expected[i]) < 1e-12);
  }
}

// src/nearest/regular.rs:402-418 test_interp_hat_func
static void nearest_regular_hat() {
  std::vector<double> y;
  for (int x = 0; x < 3; ++x) y.push_back(hat_func((double)x));
  const std::vector<double> obs = linspace(-2.0, 4.0, 100);
  auto interpolator = NearestRegular<double, 1>::new_({3}, {0.0}, {1.0}, y).unwrap();
  for (double x : obs) EXPECT(y[nearest_regular_index(x, 0.0, 1.0, y.size())] == interpolator.interp_one({x}).unwrap());
}

// src/nearest/rectilinear.rs:287-312 test_interp_extrap_2d_small
static void nearest_rectilinear_2d_small() {
  const std::size_t nx = 3, ny = 2;
  const std::vector<double> x = linspace(-1.0, 1.0, nx), y = {0.5, 0.6};
  const Table xy = meshgrid<double>({&x, &y});
  std::vector<double> z(nx * ny);
  for (std::size_t i = 0; i < nx * ny; ++i) z[i] = xy[i][0] + xy[i][1];
  const std::vector<double> xobs = linspace(-10.0, 10.0, 5), yobs = linspace(-10.0, 10.0, 5);
  const Table xyobs = meshgrid<double>({&xobs, &yobs});
  auto interpolator = NearestRectilinear<double, 2>::new_({Slice<double>(x), Slice<double>(y)}, z).unwrap();
  for (const auto& p : xyobs) {
    const double zii = interpolator.interp_one({p[0], p[1]}).unwrap();
    const double expected = x[nearest_rectilinear_index(p[0], x)] + y[nearest_rectilinear_index(p[1], y)];
    EXPECT(std::fabs(expected - zii) < 1e-12);
  }
}

// The struct form of the multicubic interpolators (multicubic/regular.rs:239-313,
// rectilinear.rs:193-253): `new` + `interp` give what the dispatch function gives.
static void multicubic_structs() {
  const Table xs = test_axes(3, 5, nullptr);
  const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x * x; });
  const Table gridobs = meshgrid(refs(obs_axes(3, -7.0, 7.0, 7)));
  const Table t = transpose(gridobs, 3);
  std::vector<double> a(gridobs.size()), b(gridobs.size()), c(gridobs.size());
  const std::vector<std::size_t> dims(3, 5);
  std::vector<double> starts, steps;
  for (const auto& x : xs) {
    starts.push_back(x[0]);
    steps.push_back(x[1] - x[0]);
  }
  multicubic::regular::interpn<double>(dims, starts, steps, u, true, slices(t), a).unwrap();
  auto reg = MulticubicRegular<double, 3>::new_({5, 5, 5}, {starts[0], starts[1], starts[2]}, {steps[0], steps[1], steps[2]}, u, true).unwrap();
  reg.interp({Slice<double>(t[0]), Slice<double>(t[1]), Slice<double>(t[2])}, b).unwrap();
  EXPECT(a == b);
  auto rect = MulticubicRectilinear<double, 3>::new_({Slice<double>(xs[0]), Slice<double>(xs[1]), Slice<double>(xs[2])}, u, true).unwrap();
  rect.interp({Slice<double>(t[0]), Slice<double>(t[1]), Slice<double>(t[2])}, c).unwrap();
  for (std::size_t i = 0; i < a.size(); ++i) EXPECT(std::fabs(a[i] - c[i]) < 1e-10);  // regular and rectilinear forms of one grid
  EXPECT(std::fabs(reg.interp_one({gridobs[10][0], gridobs[10][1], gridobs[10][2]}).unwrap() - a[10]) == 0.0);
}

// src/nearest/rectilinear.rs:319-371 test_interp_extrap_1d_to_6d
static void nearest_rectilinear_extrap() {
  Rng rng{0x6};
  for (std::size_t ndims = 1; ndims <= 6; ++ndims) {
    const Table xs = test_axes(ndims, 2, &rng);
    const std::vector<double> u = field(meshgrid(refs(xs)), [](double x) { return x; });
    const Table gridobs = meshgrid(refs(obs_axes(ndims, -7.0, 7.0, 3)));
    const Table gridobs_t = transpose(gridobs, ndims);
    std::vector<double> expected;
    for (const auto& p : gridobs) {
      double v = 0.0;
      for (std::size_t d = 0; d < ndims; ++d) v += xs[d][nearest_rectilinear_index(p[d], xs[d])];
      expected.push_back(v);
    }
    std::vector<double> out(expected.size(), 0.0);
    nearest::rectilinear::interpn<double>(slices(xs), u, slices(gridobs_t), out).unwrap();
    for (std::size_t i = 0; i < expected.size(); ++i) EXPECT(std::fabs(out[i] - expected[i]) < 1e-12);
  }
}

// src/nearest/rectilinear.rs:375-392 test_interp_hat_func
static void nearest_rectilinear_hat() {
  const std::vector<double> x = {0.0, 1.0, 2.0};
  std::vector<double> y;
  for (double xi : x) y.push_back(hat_func(xi));
  const std::vector<double> obs = linspace(-2.0, 4.0, 100);
  auto interpolator = NearestRectilinear<double, 1>::new_({Slice<double>(x)}, y).unwrap();
  for (double xo : obs) EXPECT(y[nearest_rectilinear_index(xo, x)] == interpolator.interp_one({xo}).unwrap());
}

// ---------------------------------------------------------------------------------------------
// src/lib.rs:29-80 doctests: a constant field stays constant under interpolation and extrapolation
static void lib_doctests() {
  const std::array<double, 4> x = {1.0, 2.0, 3.0, 4.0}, y = {0.0, 1.0, 2.0, 3.0};
  const std::vector<Slice<double>> grids = {Slice<double>(x), Slice<double>(y)};
  const std::array<std::size_t, 2> dims = {x.size(), y.size()};
  const std::array<double, 2> starts = {x[0], y[0]}, steps = {x[1] - x[0], y[1] - y[0]};
  std::array<double, 16> z;
  z.fill(2.0);
  const std::array<double, 2> xobs = {0.0, 5.0}, yobs = {-1.0, 3.0};
  const std::vector<Slice<double>> obs = {Slice<double>(xobs), Slice<double>(yobs)};
  std::array<double, 2> out = {0.0, 0.0};
  multilinear::regular::interpn<double>(dims, starts, steps, z, obs, out).unwrap();
  EXPECT(out[0] == 2.0 && out[1] == 2.0);
  out.fill(0.0);
  multicubic::regular::interpn<double>(dims, starts, steps, z, false, obs, out).unwrap();
  EXPECT(out[0] == 2.0 && out[1] == 2.0);
  out.fill(0.0);
  multilinear::rectilinear::interpn<double>(grids, z, obs, out).unwrap();
  EXPECT(out[0] == 2.0 && out[1] == 2.0);
  out.fill(0.0);
  multicubic::rectilinear::interpn<double>(grids, z, false, obs, out).unwrap();
  EXPECT(out[0] == 2.0 && out[1] == 2.0);
}

// Error strings of `new` / `interpn` (regular.rs:60,111-113,239-252; rectilinear.rs:185-199;
// multicubic/regular.rs:254-270; nearest/regular.rs:97) and the abort-at-first-bad-point contract
// of `interp` (regular.rs:277-280 with :418).
static void error_strings() {
  const std::vector<double> v4(4, 1.0);
  EXPECT(!std::strcmp((MultilinearRegular<double, 2>::new_({2, 3}, {0., 0.}, {1., 1.}, v4)).err(), "Dimension mismatch"));
  EXPECT(!std::strcmp((MultilinearRegular<double, 2>::new_({4, 1}, {0., 0.}, {1., 1.}, v4)).err(),
                      "All grids must have at least two entries"));
  EXPECT(!std::strcmp((MultilinearRegular<double, 2>::new_({2, 2}, {0., 0.}, {1., 0.}, v4)).err(),
                      "All grids must be monotonically increasing"));
  const std::vector<double> g2 = {0.0, 1.0}, g1 = {0.0}, gdown = {1.0, 0.0};
  EXPECT(!std::strcmp((MultilinearRectilinear<double, 2>::new_({Slice<double>(g2), Slice<double>(g1)}, std::vector<double>(2, 0.0))).err(),
                      "All grids must have at least 2 entries"));
  EXPECT(!std::strcmp((MultilinearRectilinear<double, 2>::new_({Slice<double>(g2), Slice<double>(gdown)}, v4)).err(),
                      "All grids must be monotonically increasing"));
  const std::vector<double> v9(9, 1.0);
  EXPECT(!std::strcmp((MulticubicRegular<double, 2>::new_({3, 3}, {0., 0.}, {1., 1.}, v9, false)).err(),
                      "All grids must have at least four entries"));
  const std::vector<double> g3 = {0.0, 1.0, 2.0};
  EXPECT(!std::strcmp((MulticubicRectilinear<double, 2>::new_({Slice<double>(g3), Slice<double>(g3)}, v9, false)).err(),
                      "All grids must have at least 4 entries"));
  // nine dimensions through the dispatch function
  {
    const std::vector<std::size_t> dims(9, 2);
    const std::vector<double> starts(9, 0.0), steps(9, 1.0), vals(512, 0.0);
    const Table obs(9, std::vector<double>(1, 0.5));
    std::vector<double> out(1);
    auto r = multilinear::regular::interpn<double>(dims, starts, steps, vals, slices(obs), out);
    EXPECT(!std::strcmp(r.err(), "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions."));
    bool threw = false;
    try {
      r.unwrap();
    } catch (const std::runtime_error& e) {
      threw = std::string(e.what()).find("Dimension exceeds maximum") == 0;
    }
    EXPECT(threw);  // `.unwrap()` of an Err: Rust panics with the message, the mirror throws it
  }
  // obs / out length mismatch in `interp` (regular.rs:271-274)
  {
    auto it = MultilinearRegular<double, 1>::new_({3}, {0.0}, {1.0}, std::vector<double>{0., 1., 0.}).unwrap();
    const std::vector<double> obs = {0.5, 1.5};
    std::vector<double> out(3);
    EXPECT(!std::strcmp(it.interp({Slice<double>(obs)}, out).err(), "Dimension mismatch"));
    // a coordinate that cannot be converted to an index stops the batch there: out[0..i) written,
    // out[i..] untouched
    const std::vector<double> bad = {0.5, 1.5, NAN, 0.25};
    std::vector<double> o4(4, -1.0);
    auto r = it.interp({Slice<double>(bad)}, o4);
    EXPECT(!std::strcmp(r.err(), "Unrepresentable coordinate value"));
    EXPECT(o4[0] == 0.5 && o4[1] == 0.5 && o4[2] == -1.0 && o4[3] == -1.0);
    EXPECT(!std::strcmp(it.interp_one({NAN}).err(), "Unrepresentable coordinate value"));
  }
}

// check_bounds (regular.rs:145-182, rectilinear.rs:109-134), f32 flavour of the generic functions
static void check_bounds_and_f32() {
  const std::vector<std::size_t> dims = {5, 3};
  const std::vector<double> starts = {0.0, 20.0}, steps = {2.5, 5.0};
  const Table inside = {{0.0, 10.0, 5.0}, {20.0, 30.0, 25.0}};
  const Table outside = {{0.0, 10.0 + 1e-6, 5.0}, {20.0, 30.0, 25.0}};
  bool flags[2] = {true, true};
  multilinear::regular::check_bounds<double>(dims, starts, steps, slices(inside), 1e-8, flags).unwrap();
  EXPECT(!flags[0] && !flags[1]);
  multicubic::regular::check_bounds<double>(dims, starts, steps, slices(outside), 1e-8, flags).unwrap();
  EXPECT(flags[0] && !flags[1]);
  const Table grids = {{0.0, 2.5, 5.0, 7.5, 10.0}, {20.0, 25.0, 30.0}};
  nearest::rectilinear::check_bounds<double>(slices(grids), slices(outside), 1e-8, flags).unwrap();
  EXPECT(flags[0] && !flags[1]);
  bool one[1];
  EXPECT(!std::strcmp(multilinear::rectilinear::check_bounds<double>(slices(grids), slices(inside), 1e-8, one).err(), "Dimension mismatch"));
  // f32 through the same generic functions (test/test_multilinear_regular.py runs both dtypes):
  // on-grid points return the grid values exactly
  const std::vector<float> xf = {0.0f, 2.5f, 5.0f, 7.5f, 10.0f}, yf = {20.0f, 25.0f, 30.0f};
  std::vector<float> zf, ox, oy;
  for (float a : xf)
    for (float b : yf) {
      zf.push_back(a + 2.0f * b);
      ox.push_back(a);
      oy.push_back(b);
    }
  std::vector<float> outf(zf.size());
  const std::vector<Slice<float>> obsf = {Slice<float>(ox), Slice<float>(oy)};
  multilinear::regular::interpn<float>(dims, std::vector<float>{0.0f, 20.0f}, std::vector<float>{2.5f, 5.0f}, zf, obsf, outf).unwrap();
  for (std::size_t i = 0; i < zf.size(); ++i) EXPECT(outf[i] == zf[i]);
  const std::vector<Slice<float>> gridsf = {Slice<float>(xf), Slice<float>(yf)};
  multilinear::rectilinear::interpn<float>(gridsf, zf, obsf, outf).unwrap();
  for (std::size_t i = 0; i < zf.size(); ++i) EXPECT(outf[i] == zf[i]);
}

int main() {
  if (interpn_hip_device_count() < 1) {
    std::printf("no HIP device: the library has no CPU path\n");
    return 2;
  }
  run("multilinear::regular test_interp_extrap_1d_to_6d / _1d_to_8d", multilinear_regular_extrap);
  run("multilinear::regular test_interp_hat_func", multilinear_regular_hat);
  run("multilinear::rectilinear test_interp_extrap_2d_small", multilinear_rectilinear_2d_small);
  run("multilinear::rectilinear test_interp_extrap_1d_to_6d / _1d_to_8d", multilinear_rectilinear_extrap);
  run("multilinear::rectilinear test_interp_hat_func", multilinear_rectilinear_hat);
  run("multicubic::regular test_interp_extrap_1d_to_4d_linear (+ recursive arm, 5d)", multicubic_regular_linear);
  run("multicubic::regular test_interp_extrap_1d_to_4d_quadratic", multicubic_regular_quadratic);
  run("multicubic::regular test_interp_1d_to_3d_sine", multicubic_regular_sine);
  run("multicubic::rectilinear test_interp_extrap_1d_to_4d_linear", multicubic_rectilinear_linear);
  run("multicubic::rectilinear test_interp_extrap_1d_to_4d_quadratic", multicubic_rectilinear_quadratic);
  run("multicubic::rectilinear test_interp_1d_to_3d_sine", multicubic_rectilinear_sine);
  run("nearest::regular test_interp_extrap_1d_to_6d", nearest_regular_extrap);
  run("nearest::regular test_interp_hat_func", nearest_regular_hat);
  run("nearest::rectilinear test_interp_extrap_2d_small", nearest_rectilinear_2d_small);
  run("multicubic structs: new + interp + interp_one", multicubic_structs);
  run("nearest::rectilinear test_interp_extrap_1d_to_6d", nearest_rectilinear_extrap);
  run("nearest::rectilinear test_interp_hat_func", nearest_rectilinear_hat);
  run("lib.rs doctests (constant field)", lib_doctests);
  run("error strings and abort-at-first-bad-point", error_strings);
  run("check_bounds, f32 instantiations", check_bounds_and_f32);
  std::printf("%d checks, %d failures\n", g_checks, g_failures);
  if (g_failures == 0) std::printf("ALL PASSED\n");
  return g_failures == 0 ? 0 : 1;
}
