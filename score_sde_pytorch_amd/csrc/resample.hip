// upfirdn2d for NHWC fp32: zero-insert upsample, zero pad, 2-D FIR (true
// convolution, i.e. correlation with the flipped kernel), decimate.
// Reference semantics: upfirdn2d_native (op/upfirdn2d.py:159-200), which the CUDA
// kernel op/upfirdn2d_kernel.cu:107-207 reproduces; callers upsample_2d /
// downsample_2d / conv_downsample_2d (models/up_or_down_sampling.py:144-257) and
// the 2x2 box filters of naive_upsample_2d / naive_downsample_2d (:59-69).
// The optional GroupNorm(+SiLU) prologue implements "h = act(GroupNorm_0(x))"
// of an up/down ResnetBlockBigGANpp (layerspp.py:243-258) without writing h.
//
// HBM-bound.  One lane = 4 consecutive channels of one output pixel (16-byte
// loads/stores, a wave covers 256 contiguous channels); the <= 16 taps re-read
// neighbours from L1/L2, so HBM traffic is (in + out) * 4 B.
#include "ssde_common.h"

namespace {

struct FirParams {
  ssde_src src;
  int n, h_in, w_in, c, h_out, w_out, up, down, pad0, pad1, kh, kw;
  float kf[16];   // flipped kernel
  float* dst;
  int accumulate;
};

__global__ __launch_bounds__(256) void upfirdn_kernel(const FirParams p) {
  const int c4n = p.c >> 2;
  const size_t total = (size_t)p.n * p.h_out * p.w_out * c4n;
  const SsdePro pro = ssde_pro_decode(p.src);
  const bool use_gn = pro.gn;
  const int cpg = use_gn ? p.c / p.src.gn_groups : 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    size_t t = idx / c4n;
    const int ox = (int)(t % p.w_out); t /= p.w_out;
    const int oy = (int)(t % p.h_out);
    const int n = (int)(t / p.h_out);
    const int ch = c4 * 4;
    float mu = 0.f, rs = 1.f;
    float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
    if (use_gn) {
      const int g = ch / cpg;
      mu = p.src.gn_mean[n * p.src.gn_groups + g];
      rs = p.src.gn_rstd[n * p.src.gn_groups + g];
      gam = *reinterpret_cast<const float4*>(p.src.gn_gamma + ch);
      bet = *reinterpret_cast<const float4*>(p.src.gn_beta + ch);
    }
    const float* xin = p.src.p0 + (size_t)n * p.h_in * p.w_in * p.c + ch;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < p.kh; ++ky) {
      const int uy = oy * p.down + ky - p.pad0;
      if (uy < 0 || uy % p.up != 0) continue;
      const int iy = uy / p.up;
      if (iy >= p.h_in) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ux = ox * p.down + kx - p.pad0;
        if (ux < 0 || ux % p.up != 0) continue;
        const int ix = ux / p.up;
        if (ix >= p.w_in) continue;
        float4 v = *reinterpret_cast<const float4*>(xin + ((size_t)iy * p.w_in + ix) * p.c);
        v = ssde_pro_apply(v, mu, rs, gam, bet,
                           (uint32_t)(((size_t)n * p.h_in + iy) * p.w_in + ix) * (uint32_t)p.c + (uint32_t)ch, pro);
        const float w = p.kf[ky * p.kw + kx];
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      }
    }
    float4* o = reinterpret_cast<float4*>(p.dst + (((size_t)n * p.h_out + oy) * p.w_out + ox) * p.c + ch);
    if (p.accumulate) { const float4 old = *o; acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w; }
    *o = acc;
  }
}

}  // namespace

extern "C" int ssde_upfirdn2d(const ssde_upfirdn_args* a, void* stream) {
  SSDE_REQUIRE(a && a->src.p0 && a->dst, "upfirdn2d: null args");
  SSDE_REQUIRE(a->src.p1 == nullptr && a->src.c1 == 0, "upfirdn2d: concatenated source not supported");
  SSDE_REQUIRE(a->c > 0 && a->c % 4 == 0 && a->src.c0 == a->c, "upfirdn2d: channels must be a multiple of 4 and match src.c0");
  SSDE_REQUIRE(a->kh >= 1 && a->kh <= 4 && a->kw >= 1 && a->kw <= 4, "upfirdn2d: kernel larger than 4x4");
  SSDE_REQUIRE(a->up >= 1 && a->down >= 1, "upfirdn2d: bad up/down");
  // output size exactly as upfirdn2d_native computes it (op/upfirdn2d.py:196-197)
  const int eh = (a->h_in * a->up + a->pad0 + a->pad1 - a->kh) / a->down + 1;
  const int ew = (a->w_in * a->up + a->pad0 + a->pad1 - a->kw) / a->down + 1;
  SSDE_REQUIRE(eh == a->h_out && ew == a->w_out, "upfirdn2d: output must be %dx%d (got %dx%d)", eh, ew, a->h_out, a->w_out);
  SSDE_REQUIRE(a->pad0 >= 0 && a->pad1 >= 0, "upfirdn2d: negative pads unsupported");
  if (a->src.pro_mode == SSDE_PRO_GN || a->src.pro_mode == SSDE_PRO_GN_SILU) {
    SSDE_REQUIRE(a->src.gn_groups > 0 && a->c % a->src.gn_groups == 0 && (a->c / a->src.gn_groups) % 4 == 0,
                 "upfirdn2d: GroupNorm channels-per-group must be a multiple of 4");
    SSDE_REQUIRE(a->src.gn_mean && a->src.gn_rstd && a->src.gn_gamma && a->src.gn_beta, "upfirdn2d: GroupNorm pointers missing");
  }
  FirParams p;
  p.src = a->src;
  p.n = a->n; p.h_in = a->h_in; p.w_in = a->w_in; p.c = a->c; p.h_out = a->h_out; p.w_out = a->w_out;
  p.up = a->up; p.down = a->down; p.pad0 = a->pad0; p.pad1 = a->pad1; p.kh = a->kh; p.kw = a->kw;
  for (int y = 0; y < a->kh; ++y)
    for (int x = 0; x < a->kw; ++x) p.kf[y * a->kw + x] = a->k[(a->kh - 1 - y) * a->kw + (a->kw - 1 - x)];
  p.dst = a->dst;
  p.accumulate = a->accumulate;
  const size_t total = (size_t)a->n * a->h_out * a->w_out * (a->c / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(upfirdn_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  SSDE_LAUNCH_CHECK();
  return SSDE_OK;
}
