// metrics.hip — ensemble metrics on CA traces, float64, on the device.
//
// What the reference computes with numpy / scipy on the host after decoding
// (/root/reference/slm/utils/eval_utils.py): pairwise_distance_ca :90-102, radius_of_gyration :105-129,
// _steric_clash / validity :132-173, bonding_validity :176-188, js_pwd :227-255, js_rg :290-316 (per-frame `weights` and the
// kl=True variants included), and the histogram / JS tail of js_tica :258-289 (esmdiff_metrics_js_columns).  The arithmetic is
// restated step by step — including numpy.histogram's equal-width binning with its edge correction and
// scipy.spatial.distance.jensenshannon — in oracle/metrics_ref.py, which reproduces the reference's own outputs
// (tests/golden/g9_metrics.npz) to 1e-16; these kernels follow the same steps in f64 (compiled with
// -ffp-contract=off: numpy does not fuse (dx*dx + dy*dy) + dz*dz or i*step + first).
// All of it is HBM- / latency-bound integer and f64 work on small arrays: one thread per (frame, pair) for the
// distances, one thread per histogram column afterwards (a column's bins are private to its thread: no atomics).
#include <math.h>

#include <vector>

#include "kernels.h"

namespace ed {
namespace {

__global__ __launch_bounds__(256) void pwd_kernel(const double* __restrict__ ca, const int* __restrict__ row,
                                                  const int* __restrict__ col, int L, int D, double* __restrict__ out) {
  const int d = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (d >= D) return;
  const double* a = ca + ((int64_t)n * L + row[d]) * 3;
  const double* b = ca + ((int64_t)n * L + col[d]) * 3;
  const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  out[(int64_t)n * D + d] = sqrt((dx * dx + dy * dy) + dz * dz);
}

__global__ __launch_bounds__(256) void rg_kernel(const double* __restrict__ ca, int L, int N, double* __restrict__ rg) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double* p = ca + (int64_t)n * L * 3;
  double m[3] = {0, 0, 0};
  for (int l = 0; l < L; ++l)
    for (int i = 0; i < 3; ++i) m[i] += p[l * 3 + i];
  for (int i = 0; i < 3; ++i) m[i] /= L;
  const double w = 1.0 / L;
  double s = 0;
  for (int l = 0; l < L; ++l) {
    const double x = p[l * 3] - m[0], y = p[l * 3 + 1] - m[1], z = p[l * 3 + 2] - m[2];
    s += ((x * x + y * y) + z * z) * w;
  }
  rg[n] = sqrt(s);
}

__global__ __launch_bounds__(256) void col_minmax_kernel(const double* __restrict__ x, int N, int D, double* __restrict__ lo,
                                                         double* __restrict__ hi) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  double a = x[d], b = x[d];
  for (int n = 1; n < N; ++n) {
    const double v = x[(int64_t)n * D + d];
    a = v < a ? v : a;
    b = v > b ? v : b;
  }
  lo[d] = a;
  hi[d] = b;
}

// numpy.histogram(x[:, d], bins = n_bins, range = (lo[d], hi[d])) + pseudo count; counts [n_bins, D]
__global__ __launch_bounds__(256) void col_hist_kernel(const double* __restrict__ x, int N, int D,
                                                       const double* __restrict__ lo, const double* __restrict__ hi,
                                                       int n_bins, double pseudo, const double* __restrict__ w,
                                                       double* __restrict__ counts) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  double first = lo[d], last = hi[d];
  if (first == last) {
    first -= 0.5;
    last += 0.5;
  }
  for (int b = 0; b < n_bins; ++b) counts[(int64_t)b * D + d] = pseudo;
  const double step = (last - first) / n_bins, width = last - first;
  auto edge = [&](int i) { return i == n_bins ? last : (double)i * step + first; };
  for (int n = 0; n < N; ++n) {
    const double v = x[(int64_t)n * D + d];
    if (!(v >= first && v <= last)) continue;
    int idx = (int)(((v - first) / width) * n_bins);
    if (idx == n_bins) idx -= 1;
    if (v < edge(idx)) idx -= 1;
    if (v >= edge(idx + 1) && idx != n_bins - 1) idx += 1;
    counts[(int64_t)idx * D + d] += w ? w[n] : 1.0;  // numpy.histogram(weights=): per-frame weights, accumulated in frame order
  }
}

// scipy.special.kl_div(p, q) = p log(p / q) - p + q on the (un-normalised, pseudo-counted) histograms, summed per column;
// the reference takes .mean() over all n_bins x D elements (eval_utils.py:247-249, :305-307)
__global__ __launch_bounds__(256) void col_kl_kernel(const double* __restrict__ p, const double* __restrict__ q, int n_bins,
                                                     int D, double* __restrict__ kl) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  double acc = 0;
  for (int b = 0; b < n_bins; ++b) {
    const double a = p[(int64_t)b * D + d], c = q[(int64_t)b * D + d];
    acc += (a * log(a / c) - a) + c;
  }
  kl[d] = acc;
}

// scipy.spatial.distance.jensenshannon(p, q, axis = 0) per column
__global__ __launch_bounds__(256) void col_js_kernel(const double* __restrict__ p, const double* __restrict__ q, int n_bins,
                                                     int D, double* __restrict__ js) {
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  double sp = 0, sq = 0;
  for (int b = 0; b < n_bins; ++b) {
    sp += p[(int64_t)b * D + d];
    sq += q[(int64_t)b * D + d];
  }
  double left = 0, right = 0;
  for (int b = 0; b < n_bins; ++b) {
    const double a = p[(int64_t)b * D + d] / sp, c = q[(int64_t)b * D + d] / sq, m = (a + c) / 2.0;
    if (a > 0) left += a * log(a / m);
    if (c > 0) right += c * log(c / m);
  }
  js[d] = sqrt((left + right) / 2.0);
}

__global__ __launch_bounds__(256) void frame_any_below_kernel(const double* __restrict__ pwd, int D, double bar,
                                                              int* __restrict__ flag) {
  const int n = blockIdx.x;
  int hit = 0;
  for (int d = threadIdx.x; d < D; d += 256) hit |= pwd[(int64_t)n * D + d] < bar;
  if (__syncthreads_or(hit) && threadIdx.x == 0) flag[n] = 1;
}

__global__ __launch_bounds__(256) void frame_all_below_kernel(const double* __restrict__ adj, int D, const double* __restrict__ hi,
                                                              int* __restrict__ flag) {
  const int n = blockIdx.x;
  double thres = hi[0];
  for (int d = 1; d < D; ++d) thres = hi[d] > thres ? hi[d] : thres;
  thres += 1e-6;
  int bad = 0;
  for (int d = threadIdx.x; d < D; d += 256) bad |= !(adj[(int64_t)n * D + d] < thres);
  if (!__syncthreads_or(bad) && threadIdx.x == 0) flag[n] = 1;
}

struct Scratch {  // device allocations of one call, freed on scope exit
  std::vector<void*> p;
  ~Scratch() {
    for (void* q : p) hipFree(q);
  }
  template <typename T>
  T* get(size_t n) {
    void* v = nullptr;
    if (hipMalloc(&v, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    p.push_back(v);
    return (T*)v;
  }
};

// pair lists in numpy.triu_indices(L, k) order
void triu(int L, int k, std::vector<int>& row, std::vector<int>& col) {
  for (int i = 0; i < L; ++i)
    for (int j = i + k; j < L; ++j) {
      row.push_back(i);
      col.push_back(j);
    }
}

int pairs_to_device(Scratch& s, int L, int k, int** row, int** col, int* D) {
  std::vector<int> r, c;
  triu(L, k, r, c);
  *D = (int)r.size();
  *row = s.get<int>(r.size());
  *col = s.get<int>(c.size());
  if (!*row || !*col) return ESMDIFF_E_HIP;
  if (*D) {
    hipMemcpy(*row, r.data(), r.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(*col, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  }
  return 0;
}

// mean over columns of JS(hist(model col), hist(ref col)); x arrays are [N, D] on the device
int js_columns(Scratch& s, const double* xm, int nm, const double* wm, const double* xr, int nr, const double* wr, int D,
               int n_bins, int kl, double* out, hipStream_t st) {
  if (D <= 0 || n_bins <= 0) return ESMDIFF_E_INVALID;
  double *lo = s.get<double>(D), *hi = s.get<double>(D), *cm = s.get<double>((size_t)n_bins * D),
         *cr = s.get<double>((size_t)n_bins * D), *js = s.get<double>(D);
  if (!lo || !hi || !cm || !cr || !js) return ESMDIFF_E_HIP;
  const dim3 g((D + 255) / 256), b(256);
  hipLaunchKernelGGL(col_minmax_kernel, g, b, 0, st, xr, nr, D, lo, hi);
  hipLaunchKernelGGL(col_hist_kernel, g, b, 0, st, xm, nm, D, lo, hi, n_bins, 1e-6, wm, cm);
  hipLaunchKernelGGL(col_hist_kernel, g, b, 0, st, xr, nr, D, lo, hi, n_bins, 1e-6, wr, cr);
  if (kl) hipLaunchKernelGGL(col_kl_kernel, g, b, 0, st, cm, cr, n_bins, D, js);
  else hipLaunchKernelGGL(col_js_kernel, g, b, 0, st, cm, cr, n_bins, D, js);
  std::vector<double> h(D);
  if (hipMemcpyAsync(h.data(), js, (size_t)D * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ESMDIFF_E_HIP;
  double acc = 0;
  for (double v : h) acc += v;
  *out = kl ? acc / ((double)D * n_bins) : acc / D;
  return 0;
}

}  // namespace
}  // namespace ed

using namespace ed;

extern "C" {

int esmdiff_metrics_js_pwd(const double* ca_model, int32_t n_model, const double* w_model, const double* ca_ref,
                           int32_t n_ref, const double* w_ref, int32_t L, int32_t n_bins, int32_t pwd_offset, int32_t kl,
                           double* js_out, void* stream) {
  if (!ca_model || !ca_ref || !js_out || n_model <= 0 || n_ref <= 0 || L <= 0 || pwd_offset < 0) return ESMDIFF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  int *row, *col, D;
  if (int r = pairs_to_device(s, L, pwd_offset, &row, &col, &D)) return r;
  if (D == 0) return ESMDIFF_E_INVALID;
  double *pm = s.get<double>((size_t)n_model * D), *pr = s.get<double>((size_t)n_ref * D);
  if (!pm || !pr) return ESMDIFF_E_HIP;
  hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n_model), dim3(256), 0, st, ca_model, row, col, L, D, pm);
  hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n_ref), dim3(256), 0, st, ca_ref, row, col, L, D, pr);
  return js_columns(s, pm, n_model, w_model, pr, n_ref, w_ref, D, n_bins, kl, js_out, st);
}

int esmdiff_metrics_pwd(const double* ca, int32_t n, int32_t L, int32_t pwd_offset, double* out, void* stream) {
  if (!ca || !out || n <= 0 || L <= 0 || pwd_offset < 0) return ESMDIFF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  int *row, *col, D;
  if (int r = pairs_to_device(s, L, pwd_offset, &row, &col, &D)) return r;
  if (D == 0) return ESMDIFF_E_INVALID;
  hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n), dim3(256), 0, st, ca, row, col, L, D, out);
  return hipStreamSynchronize(st) == hipSuccess ? 0 : ESMDIFF_E_HIP;  // row / col are freed on return
}

int esmdiff_metrics_js_columns(const double* x_model, int32_t n_model, const double* w_model, const double* x_ref,
                               int32_t n_ref, const double* w_ref, int32_t D, int32_t n_bins, int32_t kl, double* out,
                               void* stream) {
  if (!x_model || !x_ref || !out || n_model <= 0 || n_ref <= 0) return ESMDIFF_E_INVALID;
  Scratch s;
  return js_columns(s, x_model, n_model, w_model, x_ref, n_ref, w_ref, D, n_bins, kl, out, (hipStream_t)stream);
}

int esmdiff_metrics_js_rg(const double* ca_model, int32_t n_model, const double* w_model, const double* ca_ref,
                          int32_t n_ref, const double* w_ref, int32_t L, int32_t n_bins, int32_t kl, double* js_out,
                          void* stream) {
  if (!ca_model || !ca_ref || !js_out || n_model <= 0 || n_ref <= 0 || L <= 0) return ESMDIFF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  double *gm = s.get<double>(n_model), *gr = s.get<double>(n_ref);
  if (!gm || !gr) return ESMDIFF_E_HIP;
  hipLaunchKernelGGL(rg_kernel, dim3((n_model + 255) / 256), dim3(256), 0, st, ca_model, L, n_model, gm);
  hipLaunchKernelGGL(rg_kernel, dim3((n_ref + 255) / 256), dim3(256), 0, st, ca_ref, L, n_ref, gr);
  return js_columns(s, gm, n_model, w_model, gr, n_ref, w_ref, 1, n_bins, kl, js_out, st);
}

int esmdiff_metrics_validity(const double* ca, int32_t n, int32_t L, double ca_vdw_radius, double allowable_overlap,
                             int32_t k_exclusion, double* out, void* stream) {
  if (!ca || !out || n <= 0 || L <= 0 || k_exclusion < 0) return ESMDIFF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  int *row, *col, D;
  if (int r = pairs_to_device(s, L, k_exclusion + 1, &row, &col, &D)) return r;
  double* pw = s.get<double>((size_t)n * (D ? D : 1));
  int* flag = s.get<int>(n);
  if (!pw || !flag) return ESMDIFF_E_HIP;
  hipMemsetAsync(flag, 0, (size_t)n * 4, st);
  if (D) {
    hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n), dim3(256), 0, st, ca, row, col, L, D, pw);
    hipLaunchKernelGGL(frame_any_below_kernel, dim3(n), dim3(256), 0, st, pw, D, 2 * ca_vdw_radius - allowable_overlap, flag);
  }
  std::vector<int> h(n);
  if (hipMemcpyAsync(h.data(), flag, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ESMDIFF_E_HIP;
  int clashing = 0;
  for (int v : h) clashing += v;
  *out = 1.0 - (double)clashing / n;
  return 0;
}

int esmdiff_metrics_bonding_validity(const double* ca_model, int32_t n_model, const double* ca_ref, int32_t n_ref,
                                     int32_t L, double* out, void* stream) {
  if (!ca_model || !ca_ref || !out || n_model <= 0 || n_ref <= 0 || L < 2) return ESMDIFF_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  Scratch s;
  const int D = L - 1;
  std::vector<int> r(D), c(D);
  for (int i = 0; i < D; ++i) {
    r[i] = i;
    c[i] = i + 1;
  }
  int *row = s.get<int>(D), *col = s.get<int>(D), *flag = s.get<int>(n_model);
  double *am = s.get<double>((size_t)n_model * D), *ar = s.get<double>((size_t)n_ref * D), *lo = s.get<double>(D),
         *hi = s.get<double>(D);
  if (!row || !col || !flag || !am || !ar || !lo || !hi) return ESMDIFF_E_HIP;
  hipMemcpy(row, r.data(), (size_t)D * 4, hipMemcpyHostToDevice);
  hipMemcpy(col, c.data(), (size_t)D * 4, hipMemcpyHostToDevice);
  hipMemsetAsync(flag, 0, (size_t)n_model * 4, st);
  hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n_model), dim3(256), 0, st, ca_model, row, col, L, D, am);
  hipLaunchKernelGGL(pwd_kernel, dim3((D + 255) / 256, n_ref), dim3(256), 0, st, ca_ref, row, col, L, D, ar);
  hipLaunchKernelGGL(col_minmax_kernel, dim3((D + 255) / 256), dim3(256), 0, st, ar, n_ref, D, lo, hi);
  hipLaunchKernelGGL(frame_all_below_kernel, dim3(n_model), dim3(256), 0, st, am, D, hi, flag);
  std::vector<int> h(n_model);
  if (hipMemcpyAsync(h.data(), flag, (size_t)n_model * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return ESMDIFF_E_HIP;
  int ok = 0;
  for (int v : h) ok += v;
  *out = (double)ok / n_model;
  return 0;
}

}  // extern "C"
