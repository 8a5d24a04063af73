// Denoise filter behind Denoiser::Denoise(Color*, Color*) (reference denoiser.cpp:22-64, called from StopRender,
// RayTracer.cpp:64-77).  The reference delegates to Intel OIDN's pretrained "RT" network (ldr, srgb, colour only); its
// weights are not part of the reference checkout, so this build ships its OWN filter with the same interface and
// makes no parity claim against OIDN (DESIGN.md "Denoiser": parity unpinned).  The filter is an edge-avoiding
// a-trous wavelet (Dammertz et al. 2010): RT_DN_ITERS passes of a 5x5 B3-spline kernel with hole size 2^i, each tap
// weighted by exp(-(|dc|^2 / sc_i^2 + |dn|^2 / sn^2 + |da|^2 / sa^2)), sc_i = sc * 2^-i; the albedo / normal guides are
// the overload the reference only declares (denoiser.h:16-17).  Input is the LDR, sRGB-encoded image, as in the
// reference.  One kernel per pass, ping-pong between two RGBA float images; HBM-bound (25 gathered taps per pixel
// served from L2, 16 B written).
#pragma once
#include "rt_device.h"

#define RT_DN_ITERS 5
#define RT_DN_SIGMA_C 0.45f
#define RT_DN_SIGMA_N 0.35f
#define RT_DN_SIGMA_A 0.25f

struct DenoiseP {
  int32_t W, H, step;
  float inv_sc2, inv_sn2, inv_sa2;
  int32_t has_albedo, has_normal;
};

RT_KERNEL(k_dn_pack, const float* rgb, float4* out, uint32_t n) {
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) out[i] = make_float4(rgb[3 * (size_t)i], rgb[3 * (size_t)i + 1], rgb[3 * (size_t)i + 2], 0.0f);
}
RT_KERNEL(k_dn_unpack, const float4* in, float* rgb, uint32_t n) {
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    float4 c = in[i];
    rgb[3 * (size_t)i] = c.x; rgb[3 * (size_t)i + 1] = c.y; rgb[3 * (size_t)i + 2] = c.z;
  }
}

RT_KERNEL(k_dn_atrous, DenoiseP D, const float4* in, float4* out, const float4* albedo, const float4* normal) {
  const float hk[3] = {3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
  uint32_t total = (uint32_t)D.W * (uint32_t)D.H;
  for (uint32_t p = RT_TID; p < total; p += RT_NTHREADS) {
    int x = (int)(p % (uint32_t)D.W), y = (int)(p / (uint32_t)D.W);
    float4 cp = in[p];
    float4 ap = make_float4(0, 0, 0, 0), np = make_float4(0, 0, 0, 0);
    if (D.has_albedo) ap = albedo[p];
    if (D.has_normal) np = normal[p];
    float sr = 0.0f, sg = 0.0f, sb = 0.0f, ws = 0.0f;
    for (int dy = -2; dy <= 2; dy++) {
      int qy = y + dy * D.step;
      if (qy < 0 || qy >= D.H) continue;
      for (int dx = -2; dx <= 2; dx++) {
        int qx = x + dx * D.step;
        if (qx < 0 || qx >= D.W) continue;
        uint32_t q = (uint32_t)qy * (uint32_t)D.W + (uint32_t)qx;
        float4 cq = in[q];
        float ex = cq.x - cp.x, ey = cq.y - cp.y, ez = cq.z - cp.z;
        float e = (ex * ex + ey * ey + ez * ez) * D.inv_sc2;
        if (D.has_normal) {
          float4 nq = normal[q];
          float nx = nq.x - np.x, ny = nq.y - np.y, nz = nq.z - np.z;
          e = e + (nx * nx + ny * ny + nz * nz) * D.inv_sn2;
        }
        if (D.has_albedo) {
          float4 aq = albedo[q];
          float ax = aq.x - ap.x, ay = aq.y - ap.y, az = aq.z - ap.z;
          e = e + (ax * ax + ay * ay + az * az) * D.inv_sa2;
        }
        float w = (hk[dx < 0 ? -dx : dx] * hk[dy < 0 ? -dy : dy]) * expf(-e);
        sr = sr + cq.x * w; sg = sg + cq.y * w; sb = sb + cq.z * w;
        ws = ws + w;
      }
    }
    out[p] = make_float4(sr / ws, sg / ws, sb / ws, 0.0f);
  }
}
