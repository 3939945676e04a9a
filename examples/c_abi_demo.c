/* Pure-C consumer of libinterpn_hip.so: what a C / Rust-FFI / cgo caller of the drop-in boundary
 * does.  Build (from the repo root):
 *     gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Linterpn_amd -linterpn_hip \
 *         -Wl,-rpath,$PWD/interpn_amd -lm -o examples/c_abi_demo
 * Exercises BASELINE configs[0] (2-D multilinear-regular, 4x4 grid, 1e3 points) through the
 * one-shot entry point and through a persistent handle, a 3-D multicubic call, the reference's
 * error strings, and the abort-at-first-bad-point contract.  Prints PASS / FAIL lines; exit code
 * 0 only if everything passed.  Needs a GPU (the library has no CPU path). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "interpn_hip.h"

static int failures = 0;
#define CHECK(cond, what)                                   \
  do {                                                      \
    if (cond) printf("PASS %s\n", what);                    \
    else { printf("FAIL %s\n", what); ++failures; }         \
  } while (0)

static double lcg(uint64_t* s) {
  *s = *s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(*s >> 11) * (1.0 / 9007199254740992.0);
}

int main(void) {
  if (interpn_hip_device_count() < 1) {
    printf("no HIP device: the library has no CPU path\n");
    return 2;
  }
  /* --- configs[0]: 2-D linear on a 4x4 regular grid holding an affine field --------------- */
  enum { NX = 4, NY = 4, P = 1000 };
  const size_t dims[2] = {NX, NY};
  const double starts[2] = {-1.0, 2.0}, steps[2] = {0.5, 0.25};
  double vals[NX * NY];
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NY; ++j) vals[i * NY + j] = 3.0 * (starts[0] + steps[0] * i) - 2.0 * (starts[1] + steps[1] * j) + 0.5;
  static double x[P], y[P], out[P], out2[P];
  uint64_t seed = 42;
  for (int k = 0; k < P; ++k) {
    x[k] = -1.2 + 1.9 * lcg(&seed); /* includes extrapolation on both sides */
    y[k] = 1.9 + 1.0 * lcg(&seed);
  }
  const double* obs[2] = {x, y};
  const size_t obs_lens[2] = {P, P};
  int st = interpn_hip_linear_regular_f64(dims, 2, starts, 2, steps, 2, vals, NX * NY, obs, obs_lens, 2, out, P);
  CHECK(st == INTERPN_HIP_OK, "one-shot linear_regular_f64 returns OK");
  double worst = 0.0;
  for (int k = 0; k < P; ++k) {
    const double want = 3.0 * x[k] - 2.0 * y[k] + 0.5; /* multilinear reproduces affine fields */
    const double err = fabs(out[k] - want);
    if (err > worst) worst = err;
  }
  CHECK(worst < 1e-12, "affine field reproduced to 1e-12 (interpolation and extrapolation)");

  interpn_hip_interp* h = NULL;
  st = interpn_hip_create_regular_f64(INTERPN_HIP_LINEAR, dims, 2, starts, 2, steps, 2, vals, NX * NY,
                                      INTERPN_HIP_MEM_HOST, 0, -1, &h);
  CHECK(st == INTERPN_HIP_OK && h != NULL, "create_regular_f64 (persistent handle)");
  st = interpn_hip_eval_host(h, (const void* const*)obs, obs_lens, 2, out2, P);
  CHECK(st == INTERPN_HIP_OK && memcmp(out, out2, sizeof out) == 0, "handle eval_host bit-identical to the one-shot call");

  /* --- which kernel ran; a device-to-device clone; both handles sharing one batch ------------ */
  {
    char kname[160];
    st = interpn_hip_kernel_name(h, kname, sizeof kname);
    CHECK(st == INTERPN_HIP_OK && strncmp(kname, "interpn::k_", 11) == 0, "kernel_name reports the instantiation that ran");
    printf("       (%s)\n", kname);
    interpn_hip_interp* clone = NULL;
    st = interpn_hip_replicate(h, interpn_hip_device(h), &clone); /* another GPU of the node in a multi-GPU process */
    CHECK(st == INTERPN_HIP_OK && clone != NULL, "replicate (grid copied device to device)");
    interpn_hip_interp* pair[2];
    pair[0] = h;
    pair[1] = clone;
    uint64_t bad = 0;
    for (int k = 0; k < P; ++k) out2[k] = 0.0;
    st = interpn_hip_eval_host_sharded(pair, 2, (const void* const*)obs, obs_lens, 2, out2, P, &bad);
    CHECK(st == INTERPN_HIP_OK && memcmp(out, out2, sizeof out) == 0, "eval_host_sharded over two handles bit-identical");
    long long ppl = -1;
    CHECK(interpn_hip_get_option(h, "ppl", &ppl) == INTERPN_HIP_OK && ppl == 0 &&
              interpn_hip_set_option(h, "no_such_option", 1) == INTERPN_HIP_ERR_INVALID_ARGUMENT,
          "per-handle options by name");
    interpn_hip_destroy(clone);
  }

  /* --- abort at the first unrepresentable coordinate: prefix written, rest untouched -------- */
  for (int k = 0; k < P; ++k) out2[k] = -7.0;
  const double keep = x[500];
  x[500] = NAN;
  st = interpn_hip_eval_host(h, (const void* const*)obs, obs_lens, 2, out2, P);
  CHECK(st == INTERPN_HIP_ERR_UNREPRESENTABLE, "NaN coordinate -> INTERPN_HIP_ERR_UNREPRESENTABLE");
  CHECK(strcmp(interpn_hip_strerror(st), "Unrepresentable coordinate value") == 0, "strerror gives the reference's message");
  CHECK(memcmp(out, out2, 500 * sizeof(double)) == 0 && out2[500] == -7.0 && out2[P - 1] == -7.0,
        "out[0..500) written, out[500..] untouched");
  x[500] = keep;

  /* --- the reference's validation errors, in its order ---------------------------------- */
  st = interpn_hip_linear_regular_f64(dims, 2, starts, 2, steps, 2, vals, NX * NY - 1, obs, obs_lens, 2, out, P);
  CHECK(st == INTERPN_HIP_ERR_DIM_MISMATCH && strcmp(interpn_hip_strerror(st), "Dimension mismatch") == 0,
        "wrong vals length -> \"Dimension mismatch\"");
  const double bad_steps[2] = {0.5, -0.25};
  st = interpn_hip_linear_regular_f64(dims, 2, starts, 2, bad_steps, 2, vals, NX * NY, obs, obs_lens, 2, out, P);
  CHECK(strcmp(interpn_hip_strerror(st), "All grids must be monotonically increasing") == 0,
        "negative step -> \"All grids must be monotonically increasing\"");
  interpn_hip_destroy(h);

  /* --- 3-D multicubic on a rectilinear grid holding a quadratic field ------------------- */
  enum { N3 = 6, Q = 512 };
  double gx[N3], gy[N3], gz[N3], v3[N3 * N3 * N3];
  for (int i = 0; i < N3; ++i) {
    gx[i] = -1.0 + 0.4 * i + 0.05 * (i % 2);
    gy[i] = 0.5 * i * (1.0 + 0.1 * i);
    gz[i] = 2.0 + 0.3 * i;
  }
  for (int i = 0; i < N3; ++i)
    for (int j = 0; j < N3; ++j)
      for (int k = 0; k < N3; ++k) v3[(i * N3 + j) * N3 + k] = gx[i] * gx[i] - 0.5 * gy[j] + gz[k] * gx[i];
  static double ox[Q], oy[Q], oz[Q], o3[Q];
  for (int k = 0; k < Q; ++k) {
    ox[k] = gx[1] + (gx[N3 - 2] - gx[1]) * lcg(&seed);
    oy[k] = gy[1] + (gy[N3 - 2] - gy[1]) * lcg(&seed);
    oz[k] = gz[1] + (gz[N3 - 2] - gz[1]) * lcg(&seed);
  }
  const double* grids[3] = {gx, gy, gz};
  const size_t glens[3] = {N3, N3, N3};
  const double* obs3[3] = {ox, oy, oz};
  const size_t olens3[3] = {Q, Q, Q};
  st = interpn_hip_cubic_rectilinear_f64(grids, glens, 3, v3, N3 * N3 * N3, 1, obs3, olens3, 3, o3, Q);
  CHECK(st == INTERPN_HIP_OK, "one-shot cubic_rectilinear_f64 returns OK");
  worst = 0.0;
  for (int k = 0; k < Q; ++k) {
    const double want = ox[k] * ox[k] - 0.5 * oy[k] + oz[k] * ox[k];
    const double err = fabs(o3[k] - want) / fmax(fabs(want), 1.0);
    if (err > worst) worst = err;
  }
  CHECK(worst < 1e-10, "quadratic field reproduced to 1e-10 inside the grid (cubic Hermite)");

  printf("%s (%d failure%s)\n", failures ? "FAILED" : "ALL PASSED", failures, failures == 1 ? "" : "s");
  return failures ? 1 : 0;
}
