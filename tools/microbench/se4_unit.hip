// Unit check of the Se4 helpers (squeeze-excite FCs on v_mfma_f32_4x4x1) outside the block kernels: one workgroup, G = 4 clips, Cexp = 480, se = 20.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I multilingual_kws_amd/csrc -I include -o /tmp/se4_unit tools/microbench/se4_unit.hip
#include "../../multilingual_kws_amd/csrc/mkws_embed.hip"
#include <vector>
#include <random>
using namespace mkws;
constexpr int G = 4, CE = 480, SE = 20, NW = 8, HW = 12, LDR = 52;
__global__ __launch_bounds__(512) void unit(const float* mean, const float* WrQ, const float* We2Q, const float* br, const float* be, int T0, int NQ,
                                            float* r_out, float* gate_out, float* part_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_S = sm; float* s_P = s_S + G * CE; float* s_R = s_P + G * CE; float* s_be = s_R + 16 * LDR; float* s_E = s_be + CE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < G * CE; i += 512) s_S[i] = mean[i];
  for (int i = tid; i < CE; i += 512) s_be[i] = be[i];
  for (int i = tid; i < G * HW * (CE + 4); i += 512) s_E[i] = 1.0f;
  __syncthreads();
  f32x4 wq4[kSe4MaxTQ];
  se4_request_c1(wq4, WrQ, wave, lane);
  se4_c1<G, NW>(wq4, T0, CE, s_S, s_P, wave, lane);
  f32x4 wg4[kSe4MaxGroups][kSe4MaxNQ];
  se4_request_c2<NW>(wg4, We2Q, CE, wave, lane);
  const float br4 = (tid < 32 * G && (tid & 31) < SE) ? br[tid & 31] : 0.0f;
  __syncthreads();
  for (int i = tid; i < NW * G * 32; i += 512) part_out[i] = s_P[i];
  se4_fold<G, NW, LDR>(s_P, s_R, SE, br4, tid);
  __syncthreads();
  for (int i = tid; i < G * 32; i += 512) r_out[i] = s_R[(i >> 5) * LDR + (i & 31)];
  se4_c2_gate<G, HW, HW, NW, LDR>(wg4, CE, s_R, s_be, s_E, wave, lane, gate_out, G);
}
int main() {
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 0.3f);
  std::vector<float> wr(CE * SE), we(SE * CE), mean(G * CE), br(32, 0.f), be(CE);
  for (auto& v : wr) v = nd(rng); for (auto& v : we) v = nd(rng); for (auto& v : mean) v = nd(rng); for (int i = 0; i < SE; ++i) br[i] = nd(rng); for (auto& v : be) v = nd(rng);
  const int cpw = CE / NW, T0 = 4 * ((cpw + 7) / 8), NQ = (SE + 3) / 4, NG = (CE + 63) / 64;
  std::vector<float> qr((size_t)NW * kSe4MaxTQ * 256, 0.f), qe((size_t)NG * kSe4MaxNQ * 256, 0.f);
  for (int w = 0; w < NW; ++w) for (int t = 0; t < T0; ++t) for (int l = 0; l < 64; ++l) {
    const int kh = l / 32, n = l % 32, ch = w * cpw + (kh ? cpw - T0 + t : t);
    const bool ok = n < SE && (kh || t < cpw - T0);
    qr[(((size_t)w * kSe4MaxTQ + t / 4) * 64 + l) * 4 + t % 4] = ok ? wr[(size_t)ch * SE + n] : 0.f;
  }
  for (int gq = 0; gq < NG; ++gq) for (int n = 0; n < 4 * NQ; ++n) for (int l = 0; l < 64; ++l) {
    const int ch = 64 * gq + l;
    qe[(((size_t)gq * kSe4MaxNQ + n / 4) * 64 + l) * 4 + n % 4] = (n < SE && ch < CE) ? we[(size_t)n * CE + ch] : 0.f;
  }
  auto up = [](const std::vector<float>& h) { float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
  float *d_mean = up(mean), *d_qr = up(qr), *d_qe = up(qe), *d_br = up(br), *d_be = up(be), *d_r, *d_g, *d_p;
  hipMalloc(&d_r, G * 32 * 4); hipMalloc(&d_g, G * CE * 4); hipMalloc(&d_p, NW * G * 32 * 4);
  const size_t lds = (size_t)(3 * G * CE + 16 * LDR + CE + G * HW * (CE + 4)) * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&unit), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(unit, dim3(1), dim3(512), lds, 0, d_mean, d_qr, d_qe, d_br, d_be, T0, NQ, d_r, d_g, d_p);
  std::vector<float> r(G * 32), g(G * CE), pp(NW * G * 32);
  hipMemcpy(r.data(), d_r, r.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(g.data(), d_g, g.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(pp.data(), d_p, pp.size() * 4, hipMemcpyDeviceToHost);
  printf("launch: %s, lds %zu\n", hipGetErrorString(hipGetLastError()), lds);
  double er = 0, eg = 0, ep = 0;
  std::vector<float> rr(G * 32, 0.f);
  for (int c = 0; c < G; ++c) for (int n = 0; n < SE; ++n) {
    double v = 0; for (int ch = 0; ch < CE; ++ch) v += (double)mean[c * CE + ch] * wr[ch * SE + n];
    double ps = 0; for (int w = 0; w < NW; ++w) ps += pp[(w * G + c) * 32 + n];
    ep = std::max(ep, std::abs(ps - v));
    v += br[n]; rr[c * 32 + n] = (float)(v / (1 + std::exp(-v)));
    er = std::max(er, (double)std::abs(r[c * 32 + n] - rr[c * 32 + n]));
  }
  for (int c = 0; c < G; ++c) for (int ch = 0; ch < CE; ++ch) {
    double v = be[ch]; for (int n = 0; n < SE; ++n) v += (double)rr[c * 32 + n] * we[n * CE + ch];
    eg = std::max(eg, std::abs(g[c * CE + ch] - 1 / (1 + std::exp(-v))));
  }
  printf("max error: partial sums %.3g, r %.3g, gate %.3g\n", ep, er, eg);
  // pre-activation sums: got (from the partials) against the reference for every (clip, unit), and where a wrong value DOES occur in the reference
  std::vector<double> want(G * SE), gotp(G * SE);
  for (int c = 0; c < G; ++c) for (int n = 0; n < SE; ++n) {
    double v = 0; for (int ch = 0; ch < CE; ++ch) v += (double)mean[c * CE + ch] * wr[ch * SE + n];
    want[c * SE + n] = v; double ps = 0; for (int w = 0; w < NW; ++w) ps += pp[(w * G + c) * 32 + n]; gotp[c * SE + n] = ps;
  }
  for (int c = 0; c < G; ++c) for (int n = 0; n < 8; ++n) {
    int mc = -1, mn = -1;
    for (int c2 = 0; c2 < G; ++c2) for (int n2 = 0; n2 < SE; ++n2) if (std::abs(want[c2 * SE + n2] - gotp[c * SE + n]) < 1e-3) { mc = c2; mn = n2; }
    printf("(clip %d unit %d) got %8.4f want %8.4f  = reference of (clip %d unit %d)\n", c, n, gotp[c * SE + n], want[c * SE + n], mc, mn);
  }
  printf("r[0][0..3] got %g %g %g %g want %g %g %g %g\n", r[0], r[1], r[2], r[3], rr[0], rr[1], rr[2], rr[3]);
  return 0;
}
