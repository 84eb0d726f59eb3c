// Error plumbing and version of libctta_hip.so (see include/ctta.h).
#include <stdarg.h>

#include "common.h"

#include <stdlib.h>

static thread_local char g_err[1024] = "";

void ctta_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ctta_last_error(void) { return g_err; }
extern "C" int ctta_version(void) { return 100; }

// ------------------------------------------------------------------------------------------
// Opt-in launch profiler: brackets conv_gemm / attention launches with HIP events ON THE
// LAUNCH STREAM (bench.py's live roofline measurement; off by default, zero cost when off).
#include <mutex>
#include <vector>

struct ProfRec { hipEvent_t a, b; int kind, variant; long long m, n, k, groups; double flops; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

bool ctta_prof_active() { return g_prof_on; }

void ctta_prof_begin(int kind, int variant, long long m, long long n, long long k, long long groups,
                     hipStream_t s) {
  ProfRec r;
  r.kind = kind; r.variant = variant; r.m = m; r.n = n; r.k = k; r.groups = groups;
  r.flops = 2.0 * (double)m * (double)n * (double)k * (double)groups;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}
void ctta_prof_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof.empty()) (void)hipEventRecord(g_prof.back().b, s);
}

extern "C" void ctta_prof_enable(int on) { g_prof_on = on != 0; }

int ctta_cu_count() {      // compute units of the current device (256 on MI355X); cached per device
  static int per_dev[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!per_dev[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    per_dev[dev] = n;
  }
  return per_dev[dev];
}

// ------------------------------------------------------------------------------------------
// Options: one table, one setter (include/ctta.h).  No environment variable is read anywhere in the library.
struct OptDef { const char* name; int def; int lo, hi; };
static const OptDef kOpts[CTTA_OPT_COUNT] = {
    {"xcd", 1, 0, 1},          {"splitk", 1, 0, 1},       {"streamk", 1, 0, 1},   {"streamk_grid", 0, 0, 1984},
    {"wgrad_stream", 1, 0, 1}, {"gn_fuse", 1, 0, 1},      {"fused_res", 1, 0, 1},
    {"ffn_fuse", 1, 0, 1},
};
int g_ctta_opt[CTTA_OPT_COUNT] = {1, 1, 1, 0, 1, 1, 1, 1};
static int opt_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < CTTA_OPT_COUNT; ++i) if (strcmp(name, kOpts[i].name) == 0) return i;
  return -1;
}
extern "C" ctta_status ctta_set_option(const char* name, int value) {
  const int i = opt_index(name);
  CTTA_REQUIRE(i >= 0, "ctta_set_option: unknown option '%s'", name ? name : "(null)");
  CTTA_REQUIRE(value >= kOpts[i].lo && value <= kOpts[i].hi, "ctta_set_option: %s = %d outside [%d, %d]", name, value, kOpts[i].lo, kOpts[i].hi);
  g_ctta_opt[i] = value;
  return CTTA_OK;
}
extern "C" ctta_status ctta_get_option(const char* name, int* value) {
  const int i = opt_index(name);
  CTTA_REQUIRE(i >= 0 && value, "ctta_get_option: unknown option '%s'", name ? name : "(null)");
  *value = g_ctta_opt[i];
  return CTTA_OK;
}
extern "C" int ctta_num_options(void) { return CTTA_OPT_COUNT; }
extern "C" const char* ctta_option_name(int i) { return (i >= 0 && i < CTTA_OPT_COUNT) ? kOpts[i].name : nullptr; }
extern "C" int ctta_option_default(int i) { return (i >= 0 && i < CTTA_OPT_COUNT) ? kOpts[i].def : 0; }

// GroupNorm statistics from the producing convolution's epilogue (engine_common.h: gn_fuse_enabled)
bool ctta_gn_fuse_on() { return ctta_opt(CTTA_OPT_GN_FUSE) != 0; }
extern "C" void ctta_set_gn_fuse(int on) { g_ctta_opt[CTTA_OPT_GN_FUSE] = on ? 1 : 0; }
extern "C" int ctta_get_gn_fuse(void) { return ctta_gn_fuse_on() ? 1 : 0; }

// Synchronises, sums the records of `kind` (0 conv_gemm, 1 attention, -1 all), optionally appends
// one CSV line per launch to `csv_path`, and clears the log.
extern "C" ctta_status ctta_prof_collect(int kind, double* total_ms, double* total_flops, int64_t* launches,
                                         const char* csv_path) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0.0, fl = 0.0;
  int64_t cnt = 0;
  FILE* f = csv_path ? fopen(csv_path, "a") : nullptr;
  for (ProfRec& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      if (kind < 0 || kind == r.kind) { ms += t; fl += r.flops; ++cnt; }
      if (f) fprintf(f, "%d,%d,%lld,%lld,%lld,%lld,%.6f,%.3f\n", r.kind, r.variant, r.m, r.n, r.k, r.groups, t,
                     r.flops / (t * 1e-3) / 1e12);
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  if (f) fclose(f);
  g_prof.clear();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = cnt;
  return CTTA_OK;
}

// Zero-fill as a KERNEL.  hipMemsetAsync becomes a memset node under stream capture, and on this ROCm (7.0 runtime) a
// replayed memset node was observed to run out of order with its neighbour kernels in small graphs: the segmented
// distillation-step capture got garbage in exactly the gradients behind memset + atomicAdd buffers (tools/dbg_seg.py).
// Everything the engines zero on a capturable path goes through here.
__global__ void ctta_zero_kernel(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
  const uint4 z = {0u, 0u, 0u, 0u};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
hipError_t ctta_zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (!bytes) return hipSuccess;
  unsigned char* b = (unsigned char*)ptr;
  const size_t head = (16 - ((uintptr_t)b & 15)) & 15;
  if (head >= bytes || head) {                      // unaligned start (never the case for arena / tensor memory)
    const size_t h = head < bytes ? head : bytes;
    hipLaunchKernelGGL(ctta_zero_kernel, dim3(1), dim3(64), 0, s, (uint4*)nullptr, (size_t)0, b, (int)h);
    b += h; bytes -= h;
    if (!bytes) return hipGetLastError();
  }
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes - n16 * 16);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ctta_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint4*)b, n16, b + n16 * 16, ntail);
  return hipGetLastError();
}
