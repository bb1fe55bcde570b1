// Host-side diffusion schedule: respacing + fp64 coefficient tables, in the reference's order of
// operations so that integer outputs are bit-exact and fp64 tables bit-identical wherever libm is.
//
//   space_timesteps            /root/reference/diffusion/respace.py:12-62
//   SpacedDiffusion.__init__   respace.py:73-87   (new_beta = 1 - abar_i / abar_prev_kept)
//   get_named_beta_schedule    gaussian_diffusion.py:98-141
//   GaussianDiffusion.__init__ gaussian_diffusion.py:153-201
//
// Compiled with -ffp-contract=off: a fused multiply-add would change linspace / posterior tables.
#include <cmath>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "common.h"

namespace latte {
namespace {

// numpy.linspace(start, stop, n, dtype=float64): y_i = i * ((stop-start)/(n-1)) + start, y_{n-1} = stop
std::vector<double> linspace(double start, double stop, int n) {
  std::vector<double> y(n);
  if (n == 1) {
    y[0] = start;
    return y;
  }
  const double step = (stop - start) / (double)(n - 1);
  for (int i = 0; i < n; ++i) y[i] = (double)i * step + start;
  y[n - 1] = stop;
  return y;
}

int named_betas(const std::string& name, int n, std::vector<double>& out) {
  if (name == "linear") {
    const double scale = 1000.0 / (double)n;  // gd:106
    out = linspace(scale * 0.0001, scale * 0.02, n);
    return LATTE_OK;
  }
  if (name == "squaredcos_cap_v2") {  // gd:113-117,125-141
    auto f = [](double t) {
      const double c = std::cos((t + 0.008) / 1.008 * M_PI / 2);
      return c * c;
    };
    out.resize(n);
    for (int i = 0; i < n; ++i) {
      const double t1 = (double)i / n, t2 = (double)(i + 1) / n;
      out[i] = std::fmin(1 - f(t2) / f(t1), 0.999);
    }
    return LATTE_OK;
  }
  return fail(LATTE_ERR_INVALID, "unknown beta schedule: " + name);
}

// respace.py:12-62; returns the kept indices as a set
int space_timesteps(int num_timesteps, const std::string& spec, std::set<int>& kept) {
  std::vector<int> counts;
  if (spec.rfind("ddim", 0) == 0) {
    const int want = std::atoi(spec.c_str() + 4);
    for (int stride = 1; stride < num_timesteps; ++stride) {
      const int len = (num_timesteps + stride - 1) / stride;  // len(range(0, n, stride))
      if (len == want) {
        for (int v = 0; v < num_timesteps; v += stride) kept.insert(v);
        return LATTE_OK;
      }
    }
    return fail(LATTE_ERR_INVALID, "cannot create exactly " + std::to_string(num_timesteps) + " steps with an integer stride");
  }
  size_t pos = 0;
  while (pos <= spec.size()) {
    size_t comma = spec.find(',', pos);
    if (comma == std::string::npos) comma = spec.size();
    const std::string tok = spec.substr(pos, comma - pos);
    if (tok.empty()) return fail(LATTE_ERR_INVALID, "bad timestep_respacing: '" + spec + "'");
    char* end = nullptr;
    const long v = std::strtol(tok.c_str(), &end, 10);
    if (*end != '\0') return fail(LATTE_ERR_INVALID, "bad timestep_respacing: '" + spec + "'");
    counts.push_back((int)v);
    pos = comma + 1;
  }
  const int nsec = (int)counts.size();
  const int size_per = num_timesteps / nsec, extra = num_timesteps % nsec;
  int start = 0;
  for (int i = 0; i < nsec; ++i) {
    const int size = size_per + (i < extra ? 1 : 0);
    const int count = counts[i];
    if (size < count)
      return fail(LATTE_ERR_INVALID, "cannot divide section of " + std::to_string(size) + " steps into " + std::to_string(count));
    const double frac = count <= 1 ? 1.0 : (double)(size - 1) / (double)(count - 1);
    double cur = 0.0;
    for (int k = 0; k < count; ++k) {
      kept.insert(start + (int)std::nearbyint(cur));  // Python round(): half to even (default FP mode)
      cur += frac;
    }
    start += size;
  }
  return LATTE_OK;
}

}  // namespace
}  // namespace latte

extern "C" {

int latte_schedule_create(int diffusion_steps, const char* timestep_respacing, const char* noise_schedule,
                          latte_schedule_t** out) {
  using namespace latte;
  if (out == nullptr || diffusion_steps <= 0) return fail(LATTE_ERR_INVALID, "schedule: bad arguments");
  std::string spec = timestep_respacing ? timestep_respacing : "";
  if (spec.empty()) spec = std::to_string(diffusion_steps);  // diffusion/__init__.py:29-30
  std::vector<double> base;
  int rc = named_betas(noise_schedule ? noise_schedule : "linear", diffusion_steps, base);
  if (rc) return rc;
  std::set<int> kept;
  rc = space_timesteps(diffusion_steps, spec, kept);
  if (rc) return rc;

  auto* s = new latte_schedule();
  // respace.py:78-86
  double ac = 1.0, last = 1.0;
  for (int i = 0; i < diffusion_steps; ++i) {
    ac = ac * (1.0 - base[i]);  // np.cumprod(1 - betas)
    if (kept.count(i)) {
      s->betas.push_back(1 - ac / last);
      last = ac;
      s->timestep_map.push_back(i);
    }
  }
  const int n = (int)s->betas.size();
  s->num_timesteps = n;
  for (int i = 0; i < n; ++i)
    if (!(s->betas[i] > 0 && s->betas[i] <= 1)) {  // gd:170
      delete s;
      return fail(LATTE_ERR_INVALID, "schedule: betas out of (0, 1]");
    }
  auto& B = s->betas;
  s->alphas_cumprod.resize(n);
  s->alphas_cumprod_prev.resize(n);
  double cp = 1.0;
  for (int i = 0; i < n; ++i) {
    s->alphas_cumprod_prev[i] = cp;
    cp = cp * (1.0 - B[i]);
    s->alphas_cumprod[i] = cp;
  }
  s->sqrt_recip_alphas_cumprod.resize(n);
  s->sqrt_recipm1_alphas_cumprod.resize(n);
  s->posterior_variance.resize(n);
  s->posterior_mean_coef1.resize(n);
  s->posterior_mean_coef2.resize(n);
  s->log_betas.resize(n);
  s->sqrt_alphas_cumprod.resize(n);
  s->sqrt_one_minus_alphas_cumprod.resize(n);
  for (int i = 0; i < n; ++i) {
    const double a = s->alphas_cumprod[i], ap = s->alphas_cumprod_prev[i];
    s->sqrt_recip_alphas_cumprod[i] = std::sqrt(1.0 / a);
    s->sqrt_recipm1_alphas_cumprod[i] = std::sqrt(1.0 / a - 1);
    s->posterior_variance[i] = B[i] * (1.0 - ap) / (1.0 - a);
    s->posterior_mean_coef1[i] = B[i] * std::sqrt(ap) / (1.0 - a);
    s->posterior_mean_coef2[i] = (1.0 - ap) * std::sqrt(1.0 - B[i]) / (1.0 - a);
    s->log_betas[i] = std::log(B[i]);
    s->sqrt_alphas_cumprod[i] = std::sqrt(a);                // gd:176
    s->sqrt_one_minus_alphas_cumprod[i] = std::sqrt(1.0 - a);  // gd:177
  }
  if (n > 1) {  // gd:191-193: log of [pv[1], pv[1:]]
    s->posterior_log_variance_clipped.resize(n);
    s->posterior_log_variance_clipped[0] = std::log(s->posterior_variance[1]);
    for (int i = 1; i < n; ++i) s->posterior_log_variance_clipped[i] = std::log(s->posterior_variance[i]);
  }
  *out = s;
  return LATTE_OK;
}

int latte_schedule_set_model_types(latte_schedule_t* s, int predict_xstart, int learn_sigma, int sigma_small) {
  if (!s) return latte::fail(LATTE_ERR_INVALID, "schedule_set_model_types: null schedule");
  s->mean_type = predict_xstart ? 1 : 0;                     // diffusion/__init__.py:33-35
  s->var_type = learn_sigma ? 0 : (sigma_small ? 2 : 1);     // :36-44
  return LATTE_OK;
}

void latte_schedule_destroy(latte_schedule_t* s) {
  if (s && s->dev_tables) (void)hipFree(s->dev_tables);
  delete s;
}

int latte_schedule_num_timesteps(const latte_schedule_t* s) { return s ? s->num_timesteps : -1; }

int latte_schedule_timestep_map(const latte_schedule_t* s, int64_t* out, int n) {
  if (!s || !out || n != s->num_timesteps) return latte::fail(LATTE_ERR_INVALID, "timestep_map: size mismatch");
  std::memcpy(out, s->timestep_map.data(), sizeof(int64_t) * n);
  return LATTE_OK;
}

int latte_schedule_table(const latte_schedule_t* s, const char* name, double* out, int n) {
  if (!s || !name || !out) return latte::fail(LATTE_ERR_INVALID, "schedule_table: null argument");
  const std::string k = name;
  const std::vector<double>* v = nullptr;
  if (k == "betas") v = &s->betas;
  else if (k == "alphas_cumprod") v = &s->alphas_cumprod;
  else if (k == "alphas_cumprod_prev") v = &s->alphas_cumprod_prev;
  else if (k == "sqrt_recip_alphas_cumprod") v = &s->sqrt_recip_alphas_cumprod;
  else if (k == "sqrt_recipm1_alphas_cumprod") v = &s->sqrt_recipm1_alphas_cumprod;
  else if (k == "posterior_variance") v = &s->posterior_variance;
  else if (k == "posterior_log_variance_clipped") v = &s->posterior_log_variance_clipped;
  else if (k == "posterior_mean_coef1") v = &s->posterior_mean_coef1;
  else if (k == "posterior_mean_coef2") v = &s->posterior_mean_coef2;
  else if (k == "log_betas") v = &s->log_betas;
  else if (k == "sqrt_alphas_cumprod") v = &s->sqrt_alphas_cumprod;
  else if (k == "sqrt_one_minus_alphas_cumprod") v = &s->sqrt_one_minus_alphas_cumprod;
  else return latte::fail(LATTE_ERR_INVALID, "schedule_table: unknown table '" + k + "'");
  if ((int)v->size() != n) return latte::fail(LATTE_ERR_INVALID, "schedule_table: size mismatch for '" + k + "'");
  std::memcpy(out, v->data(), sizeof(double) * n);
  return LATTE_OK;
}

}  // extern "C"
