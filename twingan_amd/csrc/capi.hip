// extern "C" entry points that dispatch between algorithms, plus error plumbing.
#include "tg_common.h"

static thread_local char tg_err[512] = "";

static thread_local char tg_kname[96] = "";

void tg_note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tg_kname, sizeof(tg_kname), fmt, ap);
  va_end(ap);
}

void tg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tg_err, sizeof(tg_err), fmt, ap);
  va_end(ap);
}

__global__ void tg_zero_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
    if (i < na) a[i] = 0u;
    else b[i - na] = 0u;
  }
}

int tg_zero_async(void* a, size_t a_bytes, void* b, size_t b_bytes, hipStream_t s) {
  TG_CHECK((a_bytes & 3) == 0 && (b_bytes & 3) == 0, TG_EALIGN, "tg_zero_async: sizes must be multiples of 4");
  const size_t na = a ? a_bytes / 4 : 0, nb = b ? b_bytes / 4 : 0;
  if (na + nb == 0) return TG_OK;
  hipLaunchKernelGGL(tg_zero_kernel, dim3(tg_grid_for((int64_t)(na + nb), 256, 1024)), dim3(256), 0, s, (uint32_t*)a, na,
                     (uint32_t*)b, nb);
  TG_LAUNCH_CHECK("tg_zero_async");
  return TG_OK;
}

int tg_conv2d_fwd_direct(const TgConvDesc*, const void*, const void*, const float*, void*, hipStream_t);
int tg_conv2d_bwd_data_direct(const TgConvDesc*, const void*, const void*, void*, hipStream_t);
int tg_conv2d_bwd_weight_direct(const TgConvDesc*, const void*, const void*, float*, int, hipStream_t, void* ws = nullptr,
                                size_t ws_bytes = 0);
size_t tg_conv2d_bwd_weight_workspace_direct(const TgConvDesc*);
int tg_conv2d_fwd_mfma(const TgConvDesc*, const void*, const void*, const float*, void*, hipStream_t);
int tg_conv2d_bwd_data_mfma(const TgConvDesc*, const void*, const void*, void*, hipStream_t, const void* mask = nullptr);
bool tg_conv2d_bwd_data_unpool_supported_mfma(const TgConvDesc*);
int tg_conv2d_bwd_data_unpool_mfma(const TgConvDesc*, const void*, const void*, const void*, void*, hipStream_t, const void* mask,
                                   void* gy_out, const void* y_act);
bool tg_conv2d_bwd_data_mask_fusable_mfma(const TgConvDesc*);
bool tg_conv2d_fwd_mask_fusable_mfma(const TgConvDesc*);
int tg_conv2d_fwd_masked_mfma(const TgConvDesc*, const void*, const void*, const void*, void*, hipStream_t);
size_t tg_conv2d_bwd_weight_workspace_mfma(const TgConvDesc*);
bool tg_conv2d_bwd_weight2_supported_mfma(const TgConvDesc* d);
size_t tg_conv2d_bwd_weight2_workspace_mfma(const TgConvDesc* d, int nb);
int tg_conv2d_bwd_weight2_mfma(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb, const void* gyb,
                               float* gw, int accumulate, void* ws, size_t ws_bytes, hipStream_t s, float* gbias = nullptr,
                               int bias_segs = 3);
bool tg_conv2d_bwd_weight_bias_fused_mfma(const TgConvDesc* d);

bool tg_conv_tile_upcat_supported(int h, int w, int c0, int c1, int cout);
int tg_conv_tile_upcat_bwd_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, int n1, const void* gy,
                               const void* wp, void* g0, void* g1, hipStream_t s);
int tg_conv_tile_upcat_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, const void* x0,
                           const void* x1, const void* wp, void* y, hipStream_t s, float* stats = nullptr,
                           int stat_chunks = 0, int* chunks_query = nullptr);
bool tg_conv2d_fwd_pool_supported_mfma(const TgConvDesc* d);
int tg_conv2d_fwd_pool_mfma(const TgConvDesc* d, const void* x, const void* wp, const float* bias, void* y, void* ypool,
                            hipStream_t s, void* ymask = nullptr);
int tg_conv2d_fwd_stats_chunks_mfma(const TgConvDesc* d);
int tg_conv2d_fwd_stats_mfma(const TgConvDesc* d, const void* x, const void* wp, void* y, float* partials, int chunks,
                             hipStream_t s);
size_t tg_wgrad_tile_workspace(int n, int h, int w, int cin, int cout);
int tg_wgrad_tile_upcat_run(int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, const void* x0,
                            const void* x1, const void* gy, float* gw, int accumulate, void* ws, size_t ws_bytes,
                            hipStream_t s);
int tg_conv2d_bwd_weight_mfma(const TgConvDesc*, const void*, const void*, float*, int, void*, size_t, hipStream_t,
                              float* gbias = nullptr);

static int check_desc(const char* who, const TgConvDesc* d) {
  TG_CHECK(d != nullptr, TG_EINVAL, "%s: null descriptor", who);
  TG_CHECK(d->n > 0 && d->hin > 0 && d->win > 0 && d->cin > 0 && d->hout > 0 && d->wout > 0 && d->cout > 0, TG_EINVAL,
           "%s: non-positive dimension", who);
  // the MFMA kernels take 1x1, 3x3 and the dense 4x4 VALID; the direct kernels any size up to 7x7 (the 7x7 to-RGB
  // layers of --use_larger_filter_at_rgb_layer, nets/pggan.py:172,194)
  const int kmax = d->algo == TG_ALGO_DIRECT ? 7 : 4;
  TG_CHECK(d->kh >= 1 && d->kh <= kmax && d->kw >= 1 && d->kw <= kmax, TG_EINVAL, "%s: kernel %dx%d out of range", who,
           d->kh, d->kw);
  TG_CHECK(d->pad_t >= 0 && d->pad_t < d->kh && d->pad_l >= 0 && d->pad_l < d->kw, TG_EINVAL, "%s: bad padding", who);
  // stride 1: the high-side padding implied by hout must be within the kernel
  const int pb = d->hout + d->kh - 1 - d->hin - d->pad_t, pr = d->wout + d->kw - 1 - d->win - d->pad_l;
  TG_CHECK(pb >= 0 && pb < d->kh && pr >= 0 && pr < d->kw, TG_EINVAL, "%s: output size %dx%d inconsistent with input", who,
           d->hout, d->wout);
  TG_CHECK(d->dtype == TG_F32 || d->dtype == TG_BF16 || d->dtype == TG_F16, TG_EINVAL, "%s: dtype %d", who, d->dtype);
  tg_set_elem_f16(d->dtype == TG_F16);      // read by the MFMA launchers this call reaches
  TG_CHECK(d->algo == TG_ALGO_DIRECT || d->algo == TG_ALGO_MFMA || d->algo == TG_ALGO_MFMA_V1, TG_EINVAL, "%s: algo %d", who,
           d->algo);
  TG_CHECK(d->groups >= 0 && d->groups <= TG_MAX_GROUPS && (d->groups <= 1 || d->n % d->groups == 0), TG_EINVAL,
           "%s: groups %d does not divide the batch of %d (at most %d weight sets)", who, d->groups, d->n, TG_MAX_GROUPS);
  return TG_OK;
}

// ---- weight-set groups (TgConvDesc::groups) ---------------------------------------------------------------------------
// The batch of a grouped call is G equal image ranges, range g convolved with weight set g (the two discriminators of a
// TwinGAN step as ONE launch per layer).  Kernels that select the weight set per image take the call whole (*_grouped_native_
// mfma); every other kernel is launched once per group on that group's rows, so a grouped call is always valid.
namespace {
struct Groups {
  int G;
  TgConvDesc d1;              // one group's descriptor (n / G images, groups = 1)
  size_t xin, yout;           // bytes of one group's input / output activations
  size_t wset[2];             // bytes of one weight set as the forward (mode 0) / backward-data (mode 1) operand
  size_t wmaster, bias;       // bytes of one fp32 HWIO master / bias row
  Groups(const TgConvDesc* d) {
    G = d->groups > 1 ? d->groups : 1;
    d1 = *d;
    d1.n = d->n / G;
    d1.groups = 1;
    const size_t es = d->dtype == TG_F32 ? 4 : 2;
    xin = (size_t)d1.n * d->hin * d->win * d->cin * es;
    yout = (size_t)d1.n * d->hout * d->wout * d->cout * es;
    wmaster = (size_t)d->kh * d->kw * d->cin * d->cout * sizeof(float);
    bias = (size_t)d->cout * sizeof(float);
    for (int m = 0; m < 2; ++m) wset[m] = d->algo == TG_ALGO_DIRECT ? wmaster : tg_conv2d_pack_elems(&d1, m) * 2;
  }
};
template <typename P>
inline P* at(P* p, size_t bytes) {
  return p ? reinterpret_cast<P*>(reinterpret_cast<uintptr_t>(p) + bytes) : nullptr;
}
inline bool grouped(const TgConvDesc* d) { return d->groups > 1; }
}  // namespace
// which grouped calls the MFMA dispatch takes as one launch (conv_mfma.hip); TG_GRP_*: the operation
enum { TG_GRP_FWD = 0, TG_GRP_DGRAD = 1, TG_GRP_WGRAD = 2 };
bool tg_conv2d_grouped_native_mfma(const TgConvDesc* d, int op);

static thread_local bool tg_elem_is_f16 = false;
bool tg_elem_f16() { return tg_elem_is_f16; }
void tg_set_elem_f16(bool f16) { tg_elem_is_f16 = f16; }

// -1 = not decided yet: the first question reads TG_DETERMINISTIC from the environment; tg_set_deterministic overrides.
// Process-wide on purpose (not thread-local): a trainer's side-stream threads must see what the main thread set.
static int tg_det_mode = -1;
int tg_deterministic_mode() {
  int m = __atomic_load_n(&tg_det_mode, __ATOMIC_RELAXED);
  if (m < 0) {
    const char* v = getenv("TG_DETERMINISTIC");
    m = (v && *v && atoi(v) != 0) ? 1 : 0;
    __atomic_store_n(&tg_det_mode, m, __ATOMIC_RELAXED);
  }
  return m;
}

extern "C" {

int tg_version(void) { return 100; }
int tg_set_deterministic(int on) {
  const int was = tg_deterministic_mode();
  __atomic_store_n(&tg_det_mode, on ? 1 : 0, __ATOMIC_RELAXED);
  return was;
}
int tg_get_deterministic(void) { return tg_deterministic_mode(); }
const char* tg_last_error(void) { return tg_err; }
const char* tg_last_kernel(void) { return tg_kname; }

int tg_conv2d_fwd(const TgConvDesc* d, const void* x, const void* w, const float* bias, void* y, void* stream) {
  int rc = check_desc("tg_conv2d_fwd", d);
  if (rc) return rc;
  TG_CHECK(x && w && y, TG_EINVAL, "tg_conv2d_fwd: null pointer");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(w) && tg_aligned16(y), TG_EALIGN, "tg_conv2d_fwd: pointers must be 16 B aligned");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_grouped_native_mfma(d, TG_GRP_FWD))) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_fwd(&gr.d1, at(x, g * gr.xin), at(w, g * gr.wset[0]), at(bias, g * gr.bias), at(y, g * gr.yout), stream);
    return rc;
  }
  if (d->algo != TG_ALGO_DIRECT) return tg_conv2d_fwd_mfma(d, x, w, bias, y, (hipStream_t)stream);
  return tg_conv2d_fwd_direct(d, x, w, bias, y, (hipStream_t)stream);
}

int tg_conv2d_fwd_masked(const TgConvDesc* d, const void* x, const void* w, const void* mask_src, void* y, void* stream) {
  int rc = check_desc("tg_conv2d_fwd_masked", d);
  if (rc) return rc;
  TG_CHECK(x && w && y && mask_src, TG_EINVAL, "tg_conv2d_fwd_masked: null pointer");
  TG_CHECK(d->epilogue == 0, TG_EINVAL, "tg_conv2d_fwd_masked: no bias / activation epilogue next to the mask");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(w) && tg_aligned16(y) && tg_aligned16(mask_src), TG_EALIGN,
           "tg_conv2d_fwd_masked: pointers must be 16 B aligned");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_fwd_mask_fusable_mfma(d) && tg_conv2d_grouped_native_mfma(d, TG_GRP_FWD))) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_fwd_masked(&gr.d1, at(x, g * gr.xin), at(w, g * gr.wset[0]), at(mask_src, g * gr.yout), at(y, g * gr.yout), stream);
    return rc;
  }
  if (d->algo != TG_ALGO_DIRECT && tg_conv2d_fwd_mask_fusable_mfma(d))
    return tg_conv2d_fwd_masked_mfma(d, x, w, mask_src, y, (hipStream_t)stream);
  // not fusable for this shape / algorithm: the plain conv, then the mask in place
  rc = tg_conv2d_fwd(d, x, w, nullptr, y, stream);
  if (rc) return rc;
  return tg_lrelu_bwd(y, mask_src, y, (int64_t)d->n * d->hout * d->wout * d->cout, d->lrelu_alpha, d->dtype, stream);
}

int tg_conv2d_bwd_data(const TgConvDesc* d, const void* gy, const void* w, void* gx, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_data", d);
  if (rc) return rc;
  TG_CHECK(gy && w && gx, TG_EINVAL, "tg_conv2d_bwd_data: null pointer");
  TG_CHECK(tg_aligned16(gy) && tg_aligned16(w) && tg_aligned16(gx), TG_EALIGN,
           "tg_conv2d_bwd_data: pointers must be 16 B aligned");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_grouped_native_mfma(d, TG_GRP_DGRAD))) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_data(&gr.d1, at(gy, g * gr.yout), at(w, g * gr.wset[1]), at(gx, g * gr.xin), stream);
    return rc;
  }
  if (d->algo != TG_ALGO_DIRECT) return tg_conv2d_bwd_data_mfma(d, gy, w, gx, (hipStream_t)stream);
  return tg_conv2d_bwd_data_direct(d, gy, w, gx, (hipStream_t)stream);
}

int tg_conv2d_bwd_data_masked(const TgConvDesc* d, const void* gy, const void* w, const void* x_act, void* gx,
                              void* stream) {
  int rc = check_desc("tg_conv2d_bwd_data_masked", d);
  if (rc) return rc;
  TG_CHECK(gy && w && gx && x_act, TG_EINVAL, "tg_conv2d_bwd_data_masked: null pointer");
  TG_CHECK(tg_aligned16(gy) && tg_aligned16(w) && tg_aligned16(gx) && tg_aligned16(x_act), TG_EALIGN,
           "tg_conv2d_bwd_data_masked: pointers must be 16 B aligned");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_bwd_data_mask_fusable_mfma(d) && tg_conv2d_grouped_native_mfma(d, TG_GRP_DGRAD))) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_data_masked(&gr.d1, at(gy, g * gr.yout), at(w, g * gr.wset[1]), at(x_act, g * gr.xin), at(gx, g * gr.xin), stream);
    return rc;
  }
  if (d->algo != TG_ALGO_DIRECT && tg_conv2d_bwd_data_mask_fusable_mfma(d))
    return tg_conv2d_bwd_data_mfma(d, gy, w, gx, (hipStream_t)stream, x_act);
  // not fusable for this shape / algorithm: plain backward-data, then the mask in place
  rc = tg_conv2d_bwd_data(d, gy, w, gx, stream);
  if (rc) return rc;
  return tg_lrelu_bwd(gx, x_act, gx, (int64_t)d->n * d->hin * d->win * d->cin, d->lrelu_alpha, d->dtype, stream);
}

int tg_conv2d_bwd_data_unpool_supported(const TgConvDesc* d) {
  if (!d || check_desc("tg_conv2d_bwd_data_unpool_supported", d) || d->algo == TG_ALGO_DIRECT) return 0;
  const Groups gr(d);
  return tg_conv2d_bwd_data_unpool_supported_mfma(&gr.d1) ? 1 : 0;
}

int tg_conv2d_bwd_data_unpool(const TgConvDesc* d, const void* gy_pooled, const void* y_signs, const void* w, const void* x_act,
                              void* gx, void* gy_out, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_data_unpool", d);
  if (rc) return rc;
  TG_CHECK(gy_pooled && y_signs && w && gx, TG_EINVAL, "tg_conv2d_bwd_data_unpool: null pointer");
  TG_CHECK(tg_aligned16(gy_pooled) && tg_aligned16(w) && tg_aligned16(gx) && (!x_act || tg_aligned16(x_act)) &&
               (!gy_out || tg_aligned16(gy_out)),
           TG_EALIGN, "tg_conv2d_bwd_data_unpool: pointers must be 16 B aligned");
  TG_CHECK(d->algo != TG_ALGO_DIRECT, TG_ENOSUP, "tg_conv2d_bwd_data_unpool: MFMA path only (tg_conv2d_bwd_data_unpool_supported)");
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_DGRAD)) {
    const Groups gr(d);      // one group's pooled gradient is a quarter of its gradient, its sign bytes a sixteenth
    const size_t es = d->dtype == TG_F32 ? 4 : 2;
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_data_unpool(&gr.d1, at(gy_pooled, g * (gr.yout / 4)), at(y_signs, g * (gr.yout / es / 8)), at(w, g * gr.wset[1]),
                                     at(x_act, g * gr.xin), at(gx, g * gr.xin), at(gy_out, g * gr.yout), stream);
    return rc;
  }
  return tg_conv2d_bwd_data_unpool_mfma(d, gy_pooled, y_signs, w, gx, (hipStream_t)stream, x_act, gy_out, nullptr);
}

int tg_conv2d_bwd_data_unpool_act(const TgConvDesc* d, const void* gy_pooled, const void* y_act, const void* w, const void* x_act,
                                  void* gx, void* gy_out, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_data_unpool_act", d);
  if (rc) return rc;
  TG_CHECK(gy_pooled && y_act && w && gx, TG_EINVAL, "tg_conv2d_bwd_data_unpool_act: null pointer");
  TG_CHECK(tg_aligned16(gy_pooled) && tg_aligned16(y_act) && tg_aligned16(w) && tg_aligned16(gx) && (!x_act || tg_aligned16(x_act)) &&
               (!gy_out || tg_aligned16(gy_out)),
           TG_EALIGN, "tg_conv2d_bwd_data_unpool_act: pointers must be 16 B aligned");
  TG_CHECK(d->algo != TG_ALGO_DIRECT, TG_ENOSUP, "tg_conv2d_bwd_data_unpool_act: MFMA path only (tg_conv2d_bwd_data_unpool_supported)");
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_DGRAD)) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_data_unpool_act(&gr.d1, at(gy_pooled, g * (gr.yout / 4)), at(y_act, g * gr.yout), at(w, g * gr.wset[1]),
                                         at(x_act, g * gr.xin), at(gx, g * gr.xin), at(gy_out, g * gr.yout), stream);
    return rc;
  }
  return tg_conv2d_bwd_data_unpool_mfma(d, gy_pooled, nullptr, w, gx, (hipStream_t)stream, x_act, gy_out, y_act);
}

size_t tg_conv2d_bwd_weight_workspace(const TgConvDesc* d) {
  if (!d) return 0;
  if (grouped(d)) {      // G disjoint parts: a queued slab reduction (tg_wgrad_defer) reads its part after the call returned
    if (d->n % d->groups) return 0;
    const Groups gr(d);
    return gr.G * ((tg_conv2d_bwd_weight_workspace(&gr.d1) + 255) & ~(size_t)255);
  }
  if (d->algo == TG_ALGO_DIRECT) return tg_conv2d_bwd_weight_workspace_direct(d);
  return tg_conv2d_bwd_weight_workspace_mfma(d);
}

size_t tg_conv2d_bwd_weight2_workspace(const TgConvDesc* d, int nb) {
  if (!d || nb <= 0 || check_desc("tg_conv2d_bwd_weight2_workspace", d) || d->algo == TG_ALGO_DIRECT ||
      !tg_conv2d_bwd_weight2_supported_mfma(d))
    return 0;
  if (grouped(d)) {
    if (nb % d->groups) return 0;
    const Groups gr(d);
    return gr.G * ((tg_conv2d_bwd_weight2_workspace(&gr.d1, nb / gr.G) + 255) & ~(size_t)255);
  }
  return tg_conv2d_bwd_weight2_workspace_mfma(d, nb);
}

int tg_conv2d_bwd_weight2(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb, const void* gyb,
                          float* gw, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_weight2", d);
  if (rc) return rc;
  TG_CHECK(xa && gya && xb && gyb && gw && nb > 0, TG_EINVAL, "tg_conv2d_bwd_weight2: bad arguments");
  TG_CHECK(d->algo != TG_ALGO_DIRECT && tg_conv2d_bwd_weight2_supported_mfma(d), TG_ENOSUP,
           "tg_conv2d_bwd_weight2: layer not taken by the tile kernel (query tg_conv2d_bwd_weight2_workspace first)");
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_WGRAD)) {
    TG_CHECK(nb % d->groups == 0, TG_EINVAL, "tg_conv2d_bwd_weight2: groups %d does not divide the second batch of %d", d->groups, nb);
    const Groups gr(d);
    const size_t part = ws_bytes / gr.G & ~(size_t)255, xb1 = gr.xin / gr.d1.n * (nb / gr.G), yb1 = gr.yout / gr.d1.n * (nb / gr.G);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_weight2(&gr.d1, nb / gr.G, at(xa, g * gr.xin), at(gya, g * gr.yout), at(xb, g * xb1), at(gyb, g * yb1),
                                 at(gw, g * gr.wmaster), accumulate, at(ws, g * part), part, stream);
    return rc;
  }
  return tg_conv2d_bwd_weight2_mfma(d, nb, xa, gya, xb, gyb, gw, accumulate, ws, ws_bytes, (hipStream_t)stream);
}

int tg_conv2d_upcat_supported(int h, int w, int c0, int c1, int cout) {
  return tg_conv_tile_upcat_supported(h, w, c0, c1, cout) ? 1 : 0;
}

static int check_upcat(const char* who, int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, int dtype) {
  TG_CHECK(dtype == TG_BF16 || dtype == TG_F16, TG_EINVAL, "%s: 16-bit storage only (dtype %d)", who, dtype);
  tg_set_elem_f16(dtype == TG_F16);      // read by the MFMA launchers this call reaches
  TG_CHECK(n > 0 && tg_conv_tile_upcat_supported(h, w, c0, c1, cout), TG_ENOSUP,
           "%s: needs h %% 8 == 0, w %% 16 == 0, c0 and c1 multiples of 32 (got %dx%d, %d+%d -> %d)", who, h, w, c0, c1, cout);
  TG_CHECK(gsz >= 0 && (gsz == 0 || (n % gsz == 0 && n / gsz <= 4)), TG_EINVAL, "%s: bad skip groups (n %d, gsz %d)", who, n,
           gsz);
  (void)perm;
  return TG_OK;
}

int tg_conv2d_upcat_fwd(const void* x0, const void* x1, const void* w_pack, void* y, int n, int h, int w, int c0, int c1,
                        int cout, int gsz, unsigned perm, int dtype, void* stream) {
  TG_CHECK(x0 && x1 && w_pack && y, TG_EINVAL, "tg_conv2d_upcat_fwd: null pointer");
  int rc = check_upcat("tg_conv2d_upcat_fwd", n, h, w, c0, c1, cout, gsz, perm, dtype);
  if (rc) return rc;
  return tg_conv_tile_upcat_run(n, h, w, c0, c1, cout, gsz, perm, x0, x1, w_pack, y, (hipStream_t)stream);
}

int tg_conv2d_fwd_pool_supported(const TgConvDesc* d) {
  if (check_desc("tg_conv2d_fwd_pool_supported", d) || d->algo == TG_ALGO_DIRECT) return 0;
  const Groups gr(d);
  return tg_conv2d_fwd_pool_supported_mfma(&gr.d1) ? 1 : 0;
}

int tg_conv2d_fwd_pool(const TgConvDesc* d, const void* x, const void* w_pack, const float* bias, void* y, void* y_pooled,
                       void* stream) {
  int rc = check_desc("tg_conv2d_fwd_pool", d);
  if (rc) return rc;
  TG_CHECK(x && w_pack && y && y_pooled, TG_EINVAL, "tg_conv2d_fwd_pool: null pointer");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(w_pack) && tg_aligned16(y) && tg_aligned16(y_pooled), TG_EALIGN,
           "tg_conv2d_fwd_pool: pointers must be 16 B aligned");
  TG_CHECK(d->algo != TG_ALGO_DIRECT, TG_ENOSUP, "tg_conv2d_fwd_pool: MFMA path only (query tg_conv2d_fwd_pool_supported)");
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_FWD)) {
    const Groups gr(d);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_fwd_pool(&gr.d1, at(x, g * gr.xin), at(w_pack, g * gr.wset[0]), at(bias, g * gr.bias), at(y, g * gr.yout),
                              at(y_pooled, g * (gr.yout / 4)), stream);
    return rc;
  }
  return tg_conv2d_fwd_pool_mfma(d, x, w_pack, bias, y, y_pooled, (hipStream_t)stream);
}

int tg_conv2d_fwd_pool_signs(const TgConvDesc* d, const void* x, const void* w_pack, const float* bias, void* y_signs,
                             void* y_pooled, void* stream) {
  int rc = check_desc("tg_conv2d_fwd_pool_signs", d);
  if (rc) return rc;
  TG_CHECK(x && w_pack && y_signs && y_pooled, TG_EINVAL, "tg_conv2d_fwd_pool_signs: null pointer");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(w_pack) && tg_aligned16(y_pooled) && (reinterpret_cast<uintptr_t>(y_signs) & 3u) == 0,
           TG_EALIGN, "tg_conv2d_fwd_pool_signs: pointers must be 16 B aligned (the sign bits: 4 B)");
  TG_CHECK(d->algo != TG_ALGO_DIRECT, TG_ENOSUP, "tg_conv2d_fwd_pool_signs: MFMA path only (query tg_conv2d_fwd_pool_supported)");
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_FWD)) {
    const Groups gr(d);
    const size_t es = d->dtype == TG_F32 ? 4 : 2;
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_fwd_pool_signs(&gr.d1, at(x, g * gr.xin), at(w_pack, g * gr.wset[0]), at(bias, g * gr.bias),
                                    at(y_signs, g * (gr.yout / es / 8)), at(y_pooled, g * (gr.yout / 4)), stream);
    return rc;
  }
  return tg_conv2d_fwd_pool_mfma(d, x, w_pack, bias, nullptr, y_pooled, (hipStream_t)stream, y_signs);
}

int tg_conv2d_fwd_stats_chunks(const TgConvDesc* d) {
  if (check_desc("tg_conv2d_fwd_stats_chunks", d) || d->algo == TG_ALGO_DIRECT || grouped(d)) return 0;
  return tg_conv2d_fwd_stats_chunks_mfma(d);
}

int tg_conv2d_fwd_stats(const TgConvDesc* d, const void* x, const void* w_pack, void* y, float* partials, int chunks,
                        void* stream) {
  int rc = check_desc("tg_conv2d_fwd_stats", d);
  if (rc) return rc;
  TG_CHECK(x && w_pack && y && partials, TG_EINVAL, "tg_conv2d_fwd_stats: null pointer");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(w_pack) && tg_aligned16(y), TG_EALIGN,
           "tg_conv2d_fwd_stats: pointers must be 16 B aligned");
  TG_CHECK(d->algo != TG_ALGO_DIRECT && d->epilogue == 0 && !grouped(d), TG_ENOSUP,
           "tg_conv2d_fwd_stats: MFMA path, plain epilogue, one weight set (query tg_conv2d_fwd_stats_chunks first)");
  return tg_conv2d_fwd_stats_mfma(d, x, w_pack, y, partials, chunks, (hipStream_t)stream);
}

// Input gradient of tg_conv2d_upcat_fwd's conv, straight into the two sources' gradients: g0 [n, h/2, w/2, c0] (2x2 sums of
// the first c0 channels of conv3x3^T(gy, w)) and g1 [n1, h, w, c1] (the other channels, summed over the groups that read one
// skip image) -- the backward-data kernel's epilogue instead of a concat-layout tensor + tg_upsample2x_concat_bwd.
int tg_conv2d_upcat_bwd_data(const void* gy, const void* w_pack, void* g0, void* g1, int n, int h, int w, int c0, int c1,
                             int cout, int gsz, unsigned perm, int dtype, void* stream) {
  TG_CHECK(gy && w_pack && (g0 || g1), TG_EINVAL, "tg_conv2d_upcat_bwd_data: null pointer");
  TG_CHECK(tg_aligned16(gy) && tg_aligned16(w_pack) && tg_aligned16(g0) && tg_aligned16(g1), TG_EALIGN,
           "tg_conv2d_upcat_bwd_data: pointers must be 16 B aligned");
  int rc = check_upcat("tg_conv2d_upcat_bwd_data", n, h, w, c0, c1, cout, gsz, perm, dtype);
  if (rc) return rc;
  int n1 = n;
  if (gsz) {
    int mx = 0;
    for (int k = 0; k < n / gsz; ++k) {
      const int v = (int)((perm >> (8 * k)) & 0xffu);
      if (v > mx) mx = v;
    }
    TG_CHECK(mx < 4, TG_EINVAL, "tg_conv2d_upcat_bwd_data: permutation entry %d out of range", mx);
    n1 = (mx + 1) * gsz;
  }
  return tg_conv_tile_upcat_bwd_run(n, h, w, c0, c1, cout, gsz, perm, n1, gy, w_pack, g0, g1, (hipStream_t)stream);
}

int tg_conv2d_upcat_fwd_stats_chunks(int n, int h, int w, int c0, int c1, int cout) {
  if (n <= 0 || !tg_conv_tile_upcat_supported(h, w, c0, c1, cout)) return 0;
  int chunks = 0;
  if (tg_conv_tile_upcat_run(n, h, w, c0, c1, cout, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &chunks) !=
      TG_OK)
    return 0;
  return chunks;
}

int tg_conv2d_upcat_fwd_stats(const void* x0, const void* x1, const void* w_pack, void* y, float* partials, int chunks, int n,
                              int h, int w, int c0, int c1, int cout, int gsz, unsigned perm, int dtype, void* stream) {
  TG_CHECK(x0 && x1 && w_pack && y && partials, TG_EINVAL, "tg_conv2d_upcat_fwd_stats: null pointer");
  int rc = check_upcat("tg_conv2d_upcat_fwd_stats", n, h, w, c0, c1, cout, gsz, perm, dtype);
  if (rc) return rc;
  TG_CHECK(chunks > 0 && chunks == tg_conv2d_upcat_fwd_stats_chunks(n, h, w, c0, c1, cout), TG_EINVAL,
           "tg_conv2d_upcat_fwd_stats: chunks %d does not match tg_conv2d_upcat_fwd_stats_chunks()", chunks);
  return tg_conv_tile_upcat_run(n, h, w, c0, c1, cout, gsz, perm, x0, x1, w_pack, y, (hipStream_t)stream, partials, chunks,
                                nullptr);
}

size_t tg_conv2d_upcat_bwd_weight_workspace(int n, int h, int w, int c0, int c1, int cout) {
  if (n <= 0 || !tg_conv_tile_upcat_supported(h, w, c0, c1, cout)) return 0;
  return tg_wgrad_tile_workspace(n, h, w, c0 + c1, cout);
}

int tg_conv2d_upcat_bwd_weight(const void* x0, const void* x1, const void* gy, float* gw, int accumulate, void* ws,
                               size_t ws_bytes, int n, int h, int w, int c0, int c1, int cout, int gsz, unsigned perm,
                               int dtype, void* stream) {
  TG_CHECK(x0 && x1 && gy && gw, TG_EINVAL, "tg_conv2d_upcat_bwd_weight: null pointer");
  int rc = check_upcat("tg_conv2d_upcat_bwd_weight", n, h, w, c0, c1, cout, gsz, perm, dtype);
  if (rc) return rc;
  return tg_wgrad_tile_upcat_run(n, h, w, c0, c1, cout, gsz, perm, x0, x1, gy, gw, accumulate, ws, ws_bytes,
                                 (hipStream_t)stream);
}

// gbias += sum over pixels of gy, either inside the filter-gradient kernel (tile kernels) or by tg_channel_sum
static int bias_fallback(const TgConvDesc* d, const void* gy, float* gbias, int nimg, void* stream) {
  return tg_channel_sum(gy, gbias, (int64_t)nimg * d->hout * d->wout, d->cout, 1, d->dtype, stream);
}

int tg_conv2d_bwd_weight_bias(const TgConvDesc* d, const void* x, const void* gy, float* gw, float* gbias, int accumulate,
                              void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_weight_bias", d);
  if (rc) return rc;
  TG_CHECK(x && gy && gw && gbias, TG_EINVAL, "tg_conv2d_bwd_weight_bias: null pointer");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_grouped_native_mfma(d, TG_GRP_WGRAD))) {
    const Groups gr(d);
    const size_t part = ws_bytes / gr.G & ~(size_t)255;
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_weight_bias(&gr.d1, at(x, g * gr.xin), at(gy, g * gr.yout), at(gw, g * gr.wmaster), at(gbias, g * gr.bias),
                                     accumulate, at(ws, g * part), part, stream);
    return rc;
  }
  // the fused form ends in one float atomic per workgroup and channel: not taken in deterministic mode
  if (d->algo != TG_ALGO_DIRECT && !tg_deterministic_mode() && tg_conv2d_bwd_weight_bias_fused_mfma(d))
    return tg_conv2d_bwd_weight_mfma(d, x, gy, gw, accumulate, ws, ws_bytes, (hipStream_t)stream, gbias);
  rc = tg_conv2d_bwd_weight(d, x, gy, gw, accumulate, ws, ws_bytes, stream);
  if (rc) return rc;
  return bias_fallback(d, gy, gbias, d->n, stream);
}

int tg_conv2d_bwd_weight2_bias(const TgConvDesc* d, int nb, const void* xa, const void* gya, const void* xb,
                               const void* gyb, float* gw, float* gbias, int bias_segs, int accumulate, void* ws,
                               size_t ws_bytes, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_weight2_bias", d);
  if (rc) return rc;
  TG_CHECK(xa && gya && xb && gyb && gw && gbias && nb > 0, TG_EINVAL, "tg_conv2d_bwd_weight2_bias: bad arguments");
  TG_CHECK(d->algo != TG_ALGO_DIRECT && tg_conv2d_bwd_weight2_supported_mfma(d), TG_ENOSUP,
           "tg_conv2d_bwd_weight2_bias: layer not taken by the tile kernel");
  TG_CHECK(bias_segs >= 1 && bias_segs <= 3, TG_EINVAL, "tg_conv2d_bwd_weight2_bias: bias_segs %d", bias_segs);
  if (grouped(d) && !tg_conv2d_grouped_native_mfma(d, TG_GRP_WGRAD)) {
    TG_CHECK(nb % d->groups == 0, TG_EINVAL, "tg_conv2d_bwd_weight2_bias: groups %d does not divide the second batch of %d", d->groups, nb);
    const Groups gr(d);
    const size_t part = ws_bytes / gr.G & ~(size_t)255, xb1 = gr.xin / gr.d1.n * (nb / gr.G), yb1 = gr.yout / gr.d1.n * (nb / gr.G);
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_weight2_bias(&gr.d1, nb / gr.G, at(xa, g * gr.xin), at(gya, g * gr.yout), at(xb, g * xb1), at(gyb, g * yb1),
                                      at(gw, g * gr.wmaster), at(gbias, g * gr.bias), bias_segs, accumulate, at(ws, g * part), part, stream);
    return rc;
  }
  if (tg_deterministic_mode()) {      // filter gradient without the bias MFMA, bias sums by one workgroup each
    rc = tg_conv2d_bwd_weight2_mfma(d, nb, xa, gya, xb, gyb, gw, accumulate, ws, ws_bytes, (hipStream_t)stream, nullptr, 3);
    if (rc) return rc;
    if (bias_segs & 1) rc = bias_fallback(d, gya, gbias, d->n, stream);
    if (!rc && (bias_segs & 2)) rc = bias_fallback(d, gyb, gbias, nb, stream);
    return rc;
  }
  return tg_conv2d_bwd_weight2_mfma(d, nb, xa, gya, xb, gyb, gw, accumulate, ws, ws_bytes, (hipStream_t)stream, gbias,
                                    bias_segs);
}

int tg_conv2d_bwd_weight(const TgConvDesc* d, const void* x, const void* gy, float* gw, int accumulate, void* ws,
                         size_t ws_bytes, void* stream) {
  int rc = check_desc("tg_conv2d_bwd_weight", d);
  if (rc) return rc;
  TG_CHECK(x && gy && gw, TG_EINVAL, "tg_conv2d_bwd_weight: null pointer");
  TG_CHECK(tg_aligned16(x) && tg_aligned16(gy), TG_EALIGN, "tg_conv2d_bwd_weight: pointers must be 16 B aligned");
  if (grouped(d) && !(d->algo != TG_ALGO_DIRECT && tg_conv2d_grouped_native_mfma(d, TG_GRP_WGRAD))) {
    const Groups gr(d);
    const size_t part = ws_bytes / gr.G & ~(size_t)255;
    for (int g = 0; g < gr.G && !rc; ++g)
      rc = tg_conv2d_bwd_weight(&gr.d1, at(x, g * gr.xin), at(gy, g * gr.yout), at(gw, g * gr.wmaster), accumulate, at(ws, g * part), part,
                                stream);
    return rc;
  }
  if (d->algo != TG_ALGO_DIRECT) return tg_conv2d_bwd_weight_mfma(d, x, gy, gw, accumulate, ws, ws_bytes, (hipStream_t)stream);
  return tg_conv2d_bwd_weight_direct(d, x, gy, gw, accumulate, (hipStream_t)stream, ws, ws_bytes);
}

}  // extern "C"
