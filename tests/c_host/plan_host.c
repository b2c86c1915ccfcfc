/* A host WITHOUT Python: loads a plan blob (score_sde_pytorch_amd/plan_export.py), runs one U-Net evaluation through
 * the plan-level C ABI (include/ssde.h: ssde_plan_load_file, ssde_unet_forward) and compares the result with a golden
 * output of the REFERENCE implementation (tests/golden/<case>.npz, dumped to raw float32 files by the pytest).
 * Optionally (argv[6] = "pc") runs a sampler plan for a few predictor-corrector iterations and prints a checksum.
 *
 *   plan_host <plan.blob> <x.f32> <cond.f32> <y_gold.f32> <tolerance> [refresh]
 *   plan_host train <plan.blob> <batch.f32> <z.f32> <a.f32> <s.f32> <labels.f32> <hyper.f32> <seed> <loss_ref> <tolerance>
 *       one optimisation step of a training plan (ssde_train_step): the loss must match the Python-driven fused step and
 *       the parameters must move
 *
 * Plain C; device memory through the HIP runtime API (hipMalloc / hipMemcpy), exactly what any non-Python host has.
 * Under the test-only CPU emulator the same source is built against a three-function shim (HOST_IS_DEVICE). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssde.h"

#ifdef HOST_IS_DEVICE
static int dev_alloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 1; }
static int h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); return 0; }
static int d2h(void* d, const void* s, size_t n) { memcpy(d, s, n); return 0; }
static int dev_sync(void) { return 0; }
#else
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
static int dev_alloc(void** p, size_t n) { return hipMalloc(p, n) != hipSuccess; }
static int h2d(void* d, const void* s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyHostToDevice) != hipSuccess; }
static int d2h(void* d, const void* s, size_t n) { return hipMemcpy(d, s, n, hipMemcpyDeviceToHost) != hipSuccess; }
static int dev_sync(void) { return hipDeviceSynchronize() != hipSuccess; }
#endif

static float* read_f32(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  float* v = (float*)malloc((size_t)bytes);
  if (fread(v, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "short read %s\n", path); exit(2); }
  fclose(f);
  *n = (size_t)bytes / 4;
  return v;
}

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ssde_last_error()); return 1; } \
  } while (0)

static int run_train(int argc, char** argv) {
  if (argc < 12) { fprintf(stderr, "usage: %s train plan.blob batch z a s labels hyper seed loss_ref tol\n", argv[0]); return 2; }
  ssde_plan* plan = NULL;
  CHECK(ssde_plan_load_file(argv[2], &plan));
  ssde_plan_header h;
  CHECK(ssde_plan_info(plan, &h));
  if (h.kind != SSDE_PLAN_TRAIN || h.seg[3] <= 0) { fprintf(stderr, "not a training plan with an optimizer segment\n"); return 2; }
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width, vec = (size_t)h.batch;
  size_t n[6];
  float* host[6];
  for (int i = 0; i < 6; ++i) host[i] = read_f32(argv[3 + i], &n[i]);
  if (n[0] != img || n[1] != img || n[2] != vec || n[3] != vec || n[4] != vec || n[5] < 9) { fprintf(stderr, "input sizes do not match the plan\n"); return 2; }
  void* dev[5];
  for (int i = 0; i < 5; ++i)
    if (dev_alloc(&dev[i], n[i] * 4) || h2d(dev[i], host[i], n[i] * 4)) { fprintf(stderr, "upload failed\n"); return 1; }
  void* dloss;
  if (dev_alloc(&dloss, 4)) return 1;
  /* the flat parameter buffer before and after the step (SSDE_IO_PARAMS: reference layouts in state_dict order) */
  void *pb, *pa;
  const size_t nflat = (size_t)h.n_flat;
  if (dev_alloc(&pb, nflat * 4) || dev_alloc(&pa, nflat * 4)) return 1;
  CHECK(ssde_plan_copy_io(plan, SSDE_IO_PARAMS, pb, (int64_t)nflat * 4, 0, NULL));
  float* before = (float*)malloc(nflat * 4);
  float* after = (float*)malloc(nflat * 4);
  if (dev_sync() || d2h(before, pb, nflat * 4)) return 1;
  const uint32_t seed = (uint32_t)strtoul(argv[9], NULL, 10);
  CHECK(ssde_train_step(plan, (const float*)dev[0], (const float*)dev[1], (const float*)dev[2], (const float*)dev[3], (const float*)dev[4],
                        NULL, host[5], seed, (float*)dloss, NULL));
  CHECK(ssde_plan_copy_io(plan, SSDE_IO_PARAMS, pa, (int64_t)nflat * 4, 0, NULL));
  if (dev_sync()) return 1;
  float loss = 0.f;
  if (d2h(&loss, dloss, 4) || d2h(after, pa, nflat * 4)) return 1;
  double moved = 0.0;
  for (size_t i = 0; i < nflat; ++i) moved += fabs((double)after[i] - (double)before[i]);
  const char* name = "all parameters";
  const double ref = atof(argv[10]), tol = atof(argv[11]);
  const double rel = fabs((double)loss - ref) / (fabs(ref) + 1e-30);
  printf("plan_host train: %d ops (forward from %d, loss head %d, backward %d, optimizer %d), loss %.9g (reference %.9g, rel %.3g), "
         "sum |delta %s| = %.6g\n", h.n_ops, h.seg[0], h.seg[1], h.seg[2], h.seg[3], (double)loss, ref, rel, name, moved);
  CHECK(ssde_plan_destroy(plan));
  return (rel <= tol && moved > 0.0) ? 0 : 3;
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "train") == 0) return run_train(argc, argv);
  if (argc < 6) { fprintf(stderr, "usage: %s plan.blob x.f32 cond.f32 y_gold.f32 tol [refresh]\n", argv[0]); return 2; }
  ssde_plan* plan = NULL;
  CHECK(ssde_plan_load_file(argv[1], &plan));
  ssde_plan_header h;
  CHECK(ssde_plan_info(plan, &h));
  size_t nx, nc, ny;
  float* x = read_f32(argv[2], &nx);
  float* cond = read_f32(argv[3], &nc);
  float* gold = read_f32(argv[4], &ny);
  const double tol = atof(argv[5]);
  const size_t img = (size_t)h.batch * h.channels * h.height * h.width;
  if (h.kind != SSDE_PLAN_UNET || nx != img || ny != img || nc != (size_t)h.batch) {
    fprintf(stderr, "plan is [%d,%d,%d,%d] kind %d; inputs have %zu / %zu / %zu floats\n", h.batch, h.channels, h.height, h.width, h.kind, nx, nc, ny);
    return 2;
  }
  if (argc > 6 && strcmp(argv[6], "refresh") == 0) {
    /* what a host does after copying a checkpoint into the parameter regions: here the blob's own parameters are
     * left as they are, every packed copy is wiped by re-packing from them -- the output must not change */
    int64_t numel = 0; float* dev = NULL; const char* name = NULL;
    CHECK(ssde_plan_param(plan, NULL, 0, &dev, &numel, &name));
    printf("first parameter: %s (%lld floats)\n", name, (long long)numel);
    CHECK(ssde_plan_refresh_weights(plan, NULL));
  }
  void *dx, *dc, *dy;
  if (dev_alloc(&dx, img * 4) || dev_alloc(&dc, (size_t)h.batch * 4) || dev_alloc(&dy, img * 4)) { fprintf(stderr, "device allocation failed\n"); return 1; }
  if (h2d(dx, x, img * 4) || h2d(dc, cond, (size_t)h.batch * 4)) { fprintf(stderr, "upload failed\n"); return 1; }
  CHECK(ssde_unet_forward(plan, (const float*)dx, (const float*)dc, NULL, NULL, (float*)dy, NULL));
  if (dev_sync()) { fprintf(stderr, "device synchronisation failed\n"); return 1; }
  float* y = (float*)malloc(img * 4);
  if (d2h(y, dy, img * 4)) { fprintf(stderr, "download failed\n"); return 1; }
  double max_err = 0.0, max_ref = 0.0;
  for (size_t i = 0; i < img; ++i) {
    const double e = fabs((double)y[i] - (double)gold[i]);
    if (!(e == e)) { max_err = INFINITY; break; }
    if (e > max_err) max_err = e;
    if (fabs((double)gold[i]) > max_ref) max_ref = fabs((double)gold[i]);
  }
  const double rel = max_err / (max_ref + 1e-30);
  printf("plan_host: %d ops, [%d,%d,%d,%d], max |y - y_ref| / max |y_ref| = %.3g (tolerance %.3g)\n", h.n_ops, h.batch, h.channels, h.height,
         h.width, rel, tol);
  CHECK(ssde_plan_destroy(plan));
  return rel < tol ? 0 : 3;
}
