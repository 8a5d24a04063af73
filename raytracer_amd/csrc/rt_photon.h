// Photon pass and photon-map irradiance gather (reference RayTracer.cpp:262-334, photonmap.h, materials.cpp:315-390).
//
//   k_photon_trace   RayTracer::GeneratePhotons' emission + TracePhoton's random walk, one photon path per lane,
//                    every draw from the Philox stream (photon index, bounce) so the maps are reproducible
//                    (SURVEY Q14).  Deposits are appended unordered with their (path, bounce) key; the host orders
//                    them, keeps the first n per map (exactly the photons a sequential emitter would have kept,
//                    photonmap.h:158-172) and balances the kd-tree (photonmap.h:176-264).
//   k_photon_gather  PhotonMap::EstimateIrradiance<k> (photonmap.h:268-414) for the shade kernel's gather queue:
//                    one lookup per lane, the reference's own kd-tree walk and max-heap, step for step — the
//                    heap's first replacement evicts its root whatever the newcomer's distance (dist2[0] is only
//                    refreshed on replacement, photonmap.h:383-397), so the result depends on the visiting order
//                    and only the same walk reproduces it.  Heap and walk stack live in LDS, [entry][lane], which
//                    is bank-conflict free for per-lane indices.
#pragma once
#include <vector>

#include "rtcore.h"
#include "rt_kernels.h"

#define RT_PHOTON_MAX_K 256
#define RT_PHOTON_TSTACK 40   // kd-tree depth bound: a heap-ordered tree over < 2^31 photons is < 32 deep

struct DPhotonMap {
  const float4* ph;   // 2 float4 per photon (rt_photon layout), 1-based heap order, slot 0 unused
  int32_t stored;
  int32_t half;       // halfStoredPhotons = stored / 2 - 1 (photonmap.h:197)
};

struct PhotonTraceP {
  uint32_t seed_lo, seed_hi;
  uint32_t first, count;   // photon paths [first, first + count)
  int32_t max_bounces;
  int32_t n_plights;
  int32_t accept[2];       // 0: that map already holds its n photons from earlier batches
  uint32_t cap;            // record capacity per map
};
struct PhotonOut {
  float4* rec[2];          // 2 float4 per record
  uint2* key[2];           // (photon path, bounce)
  uint32_t* count;         // [2]
  uint32_t* overflow;
  const int32_t* plights;  // indices of the photon-source lights (Light::IsPhotonSource, lights.h:63)
  DCounters* counters;
};

RT_DEV float photon_draw_unit(uint32_t u) { return u32_to_unit(u); }

// PhotonData::SetPower (photonmap.h:148-154) + position / direction, packed as rt_photon
RT_DEV void photon_pack(V3 pos, V3 dir, V3 c, float4& a, float4& b) {
  float power = c.x;
  if (power < c.y) power = c.y;
  if (power < c.z) power = c.z;
  uint32_t bits = (uint32_t)to_byte(c.x / power) | ((uint32_t)to_byte(c.y / power) << 8) | ((uint32_t)to_byte(c.z / power) << 16);
  a = make_float4(pos.x, pos.y, pos.z, power);
  b = make_float4(dir.x, dir.y, dir.z, __uint_as_float(bits));
}

template <bool CNT>
RT_KERNEL(k_photon_trace, DScene S, PhotonTraceP PP, PhotonOut O) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t rays = 0;
  for (uint32_t t = RT_TID; t < PP.count; t += RT_NTHREADS) {
    uint32_t e = PP.first + t;
    const DLight& L = S.lights[O.plights[e % (uint32_t)PP.n_plights]];  // round robin, RayTracer.cpp:278,286
    // PointLight::RandomPhoton, lights.cpp:74-109
    uint32_t r[4];
    philox4x32_10(e, 0xFFFFFFFFu, 0, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r);
    float u1 = u32_to_unit(r[0]), u2 = u32_to_unit(r[1]), u3 = u32_to_unit(r[2]), u4 = u32_to_unit(r[3]);
    float phi = two_pi_times(u2);
    float cosTheta = 1.0f - 2.0f * u1;
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    V3 spherePoint = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    V3 o = v3(L.position[0], L.position[1], L.position[2]) + (spherePoint * L.size);
    float r_hemi = sqrtf(u3);
    float phi_hemi = two_pi_times(u4);
    V3 tt, bb;
    orthonormals(spherePoint, tt, bb);
    V3 d = ((r_hemi * cosf(phi_hemi)) * tt) + ((r_hemi * sinf(phi_hemi)) * bb) + (sqrtf(1.0f - u3) * spherePoint);
    d = normalized(d);
    V3 c = v3(L.intensity[0], L.intensity[1], L.intensity[2]) * (float)(4 * RT_PI_D) * L.size * L.size;

    // RayTracer::TracePhoton (RayTracer.cpp:296-334), iterative; MtlBlinn::GenerateSample (materials.cpp:335-390)
    int prevLobe = 0;
    for (int bounce = 0; bounce < PP.max_bounces; bounce++) {
      Hit h;
      rays++;
      if (!scene_closest<CNT>(S, o, d, 1, h, stk, stride, cnt)) break;
      if (h.inst < 0) break;  // a renderable light: hInfo.node is null (RayTracer.cpp:299)
      const DMaterial& M = S.materials[S.inst[h.inst].mtl];
      if (M.kind != 1) break;  // MtlPhong / MtlMicrofacet::GenerateSample return false
      ShadePt sp;
      sp.P = h.p;
      sp.N = normalized(h.N);
      sp.V = -normalized(d);
      sp.uvw = h.uvw;
      sp.front = h.front;
      sp.pixel = 0; sp.nid = 0; sp.sample = 0;
      uint32_t r0[4], r1[4];
      philox4x32_10(e, (uint32_t)bounce, 0, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r0);
      float random = u32_to_unit(r0[0]);
      V3 kdv = v3(M.diffuse.v[0], M.diffuse.v[1], M.diffuse.v[2]);
      V3 ksv = v3(M.specular.v[0], M.specular.v[1], M.specular.v[2]);
      V3 rfv = v3(M.refraction.v[0], M.refraction.v[1], M.refraction.v[2]);
      float diffuseProb = (kdv.x + kdv.y + kdv.z) / 3.0f;   // Color::Gray
      float reflectProb = (ksv.x + ksv.y + ksv.z) / 3.0f;
      float refractProb = (rfv.x + rfv.y + rfv.z) / 3.0f;
      V3 newDir, mult;
      float prob;
      int lobe;
      if (random < diffuseProb) {
        float a1 = u32_to_unit(r0[1]), a2 = u32_to_unit(r0[2]);
        float ph = two_pi_times(a2);
        float ct = a1;
        float st = sqrtf(1.0f - a1 * a1);
        float x = st * cosf(ph), y = st * sinf(ph), z = ct;
        V3 t2, b2;
        orthonormals(sp.N, t2, b2);
        newDir = normalized((x * t2) + (y * b2) + (z * sp.N));
        mult = kdv;
        prob = diffuseProb;
        lobe = 1;
      } else if (random < diffuseProb + reflectProb) {
        philox4x32_10(e, (uint32_t)bounce, 1, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r1);
        newDir = reflect_dir(S, sp, M.glossiness.v, u32_to_unit(r1[0]), u32_to_unit(r1[1]));
        mult = ksv * dot(newDir, sp.N);
        prob = reflectProb;
        lobe = 2;
      } else if (random < diffuseProb + reflectProb + refractProb) {
        philox4x32_10(e, (uint32_t)bounce, 1, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r1);
        float ra[2] = {u32_to_unit(r1[0]), u32_to_unit(r1[1])}, rb[2] = {u32_to_unit(r1[2]), u32_to_unit(r1[3])};
        V3 ro;
        refract_ray(S, sp, M.ior, M.glossiness.v, ra, rb, ro, newDir);  // only the direction is used (materials.cpp:380)
        mult = rfv * fabsf(dot(newDir, sp.N));
        prob = refractProb;
        lobe = 4;
      } else {
        break;  // absorbed
      }
      V3 newC = (c * mult) / prob;
      if (lobe & 1) {
        int which = (prevLobe & (4 | 2)) ? 1 : 0;  // caustics map after a specular / transmission bounce
        if (PP.accept[which]) {
          uint32_t at = rt_atomic_add_u32(O.count + which, 1u);
          if (at < PP.cap) {
            float4 a, b;
            photon_pack(sp.P, -newDir, newC, a, b);
            O.rec[which][2 * (size_t)at] = a;
            O.rec[which][2 * (size_t)at + 1] = b;
            uint2 k;
            k.x = e; k.y = (uint32_t)bounce;
            O.key[which][at] = k;
          } else {
            *O.overflow = 1u;
          }
        }
      }
      prevLobe = lobe;
      o = h.p;
      d = newDir;
      c = newC;
    }
  }
  if (CNT) {
    cnt.flush(O.counters);
    if (rays) rt_atomic_add_u64(&O.counters->rays_photon, rays);
  }
}

// ------------------------------------------------------------------------------------------------ gather
struct HeapMem {     // per-lane views: entry i of this lane is at [i * stride]
  float* d2;
  uint32_t* idx;
  uint32_t* walk;
  uint32_t stride;
};

// PhotonMap::IrradianceEstimate<true, k, CONSTANT> (photonmap.h:268-309) with LocatePhotons<true> (:339-414)
// unrolled into an explicit walk: state 0 = entering a node, 1 = first child done, 2 = second child done.
RT_DEV V3 photon_estimate(const DPhotonMap& M, V3 pos, V3 normal, float radius, float ellipticity, int k, const HeapMem& hm,
                          uint32_t& visits) {
  V3 irrad = v3(0, 0, 0);
  if (M.stored == 0) return irrad;
  const uint32_t hs = hm.stride;
  float normScale = ellipticity == 1.0f ? 0.0f : 1.0f / ellipticity - 1.0f;
  int found = 0;
  float bound = radius * radius;  // np.dist2[0]
  int sp = 0;
  uint32_t index = 1, state = 0;
  for (;;) {
    if (state == 0) visits++;  // one LocatePhotons call per node entered
    float4 pa = M.ph[2 * (size_t)index];
    float4 pb = M.ph[2 * (size_t)index + 1];
    int axis = (int)((__float_as_uint(pb.w) >> 24) & 3u);
    bool process = true;
    if ((int)index < M.half && state < 2) {
      float pc = axis == 0 ? pa.x : (axis == 1 ? pa.y : pa.z);
      float qc = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
      float dist = qc - pc;
      uint32_t first = dist > 0.0f ? 2 * index + 1 : 2 * index;
      uint32_t second = dist > 0.0f ? 2 * index : 2 * index + 1;
      if (state == 0) {
        hm.walk[(sp++) * hs] = (index << 2) | 1u;
        index = first; state = 0;
        process = false;
      } else if (dist * dist < bound) {
        hm.walk[(sp++) * hs] = (index << 2) | 2u;
        index = second; state = 0;
        process = false;
      }
    }
    if (!process) continue;
    // photonmap.h:366-413
    V3 dif = v3(pa.x, pa.y, pa.z) - pos;
    float dist2 = dot(dif, dif);
    if (dist2 < bound) {
      V3 dir = v3(pb.x, pb.y, pb.z);
      bool keep = !(dot(dir, normal) >= 0.0f);
      if (keep && normScale > 0.0f) {
        float perp = dot(dif, normal);
        dif = dif + normal * (perp * normScale);
        dist2 = dot(dif, dif);
        if (dist2 >= bound) keep = false;
      }
      if (keep) {
        if (found < k) {
          found++;
          hm.d2[found * hs] = dist2;
          hm.idx[found * hs] = index;
          if (found == k) {  // build the max-heap
            int half_found = found >> 1;
            for (int kk = half_found; kk >= 1; kk--) {
              int parent = kk;
              uint32_t tp = hm.idx[kk * hs];
              float td2 = hm.d2[kk * hs];
              while (parent <= half_found) {
                int j = parent + parent;
                if (j < found && hm.d2[j * hs] < hm.d2[(j + 1) * hs]) j++;
                float dj = hm.d2[j * hs];
                if (td2 >= dj) break;
                hm.d2[parent * hs] = dj;
                hm.idx[parent * hs] = hm.idx[j * hs];
                parent = j;
              }
              hm.idx[parent * hs] = tp;
              hm.d2[parent * hs] = td2;
            }
          }
        } else {
          int parent = 1;
          int j = 2;
          while (j <= found) {
            if (j < found && hm.d2[j * hs] < hm.d2[(j + 1) * hs]) j++;
            float dj = hm.d2[j * hs];
            if (dist2 > dj) break;
            hm.d2[parent * hs] = dj;
            hm.idx[parent * hs] = hm.idx[j * hs];
            parent = j;
            j <<= 1;
          }
          hm.idx[parent * hs] = index;
          hm.d2[parent * hs] = dist2;
          bound = hm.d2[1 * hs];
        }
      }
    }
    if (sp == 0) break;
    uint32_t w = hm.walk[(--sp) * hs];
    index = w >> 2; state = w & 3u;
  }
  for (int i = 1; i <= found; i++) {  // photonmap.h:289-297, GetPower = Color24::ToColor() * power
    uint32_t id = hm.idx[i * hs];
    float4 pa = M.ph[2 * (size_t)id];
    uint32_t bits = __float_as_uint(M.ph[2 * (size_t)id + 1].w);
    V3 col = v3((float)(bits & 255u) / 255.0f, (float)((bits >> 8) & 255u) / 255.0f, (float)((bits >> 16) & 255u) / 255.0f);
    irrad = irrad + col * pa.w;
  }
  if (found > 0) {
    float area = RT_PI_F * bound;
    if (area > 0.0f) {
      float one_over_area = 1.0f / area;
      irrad = irrad * one_over_area;
    }
  }
  return irrad;
}

#ifndef RT_EMU
#define RT_GATHER_BLOCK 64
// dynamic LDS: d2[(k+1)][64], idx[(k+1)][64], walk[RT_PHOTON_TSTACK][64]
#define RT_HEAP_DECL(k_)                                                                  \
  extern __shared__ uint32_t s_heap_raw[];                                                 \
  HeapMem hm;                                                                              \
  hm.stride = RT_GATHER_BLOCK;                                                             \
  hm.d2 = (float*)s_heap_raw + threadIdx.x;                                                \
  hm.idx = s_heap_raw + (size_t)((k_) + 1) * RT_GATHER_BLOCK + threadIdx.x;                \
  hm.walk = s_heap_raw + 2 * (size_t)((k_) + 1) * RT_GATHER_BLOCK + threadIdx.x;
static inline size_t photon_heap_bytes(int k) { return ((size_t)2 * (k + 1) + RT_PHOTON_TSTACK) * RT_GATHER_BLOCK * 4; }
#else
#define RT_GATHER_BLOCK 1
#define RT_HEAP_DECL(k_)                                                                   \
  static float e_d2[RT_PHOTON_MAX_K + 1];                                                  \
  static uint32_t e_idx[RT_PHOTON_MAX_K + 1], e_walk[RT_PHOTON_TSTACK];                    \
  HeapMem hm;                                                                              \
  hm.stride = 1; hm.d2 = e_d2; hm.idx = e_idx; hm.walk = e_walk;
static inline size_t photon_heap_bytes(int) { return 0; }
#endif

struct GatherP {
  DPhotonMap map[2];
  float radius;
  float ellipticity[2];
  int32_t k;
};

// the shade kernel's deferred lookups (materials.cpp:315-322): indirect += (1/pi) * kd * E, weighted by the path weight
template <bool CNT>
RT_KERNEL(k_photon_gather, GatherP G, FrameBufs F, GatherQ Q) {
  RT_HEAP_DECL(G.k)
  uint32_t visits = 0, gathers = 0;
  RT_SEG_BEGIN(Q.counts, Q.nseg, Q.segcap, 1u)
    if (!seg_valid) continue;
    uint32_t g = seg_base_ + (uint32_t)seg_e;
    float4 a = Q.a[g], b = Q.b[g], c = Q.c[g];
    uint32_t which = __float_as_uint(b.w) & 1u;
    const DPhotonMap& M = which ? G.map[1] : G.map[0];
    if (M.stored > 0) gathers++;
    V3 E = photon_estimate(M, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), G.radius, which ? G.ellipticity[1] : G.ellipticity[0], G.k, hm, visits);
    add_radiance(F, __float_as_uint(a.w), v3(c.x, c.y, c.z) * E);
  RT_SEG_END
  if (CNT) {
    if (visits) rt_atomic_add_u64(&F.counters->photon_visits, visits);
    if (gathers) rt_atomic_add_u64(&F.counters->gathers, gathers);
  }
}

// batched PhotonMap::EstimateIrradiance for the C-ABI (rt_photon_gather)
RT_KERNEL(k_api_photon_gather, DPhotonMap M, const float* P, const float* N, uint32_t n, float radius, float ellipticity, int k,
          float* out) {
  RT_HEAP_DECL(k)
  uint32_t visits = 0;
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    V3 E = photon_estimate(M, v3(P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2]),
                           v3(N[3 * (size_t)i], N[3 * (size_t)i + 1], N[3 * (size_t)i + 2]), radius, ellipticity, k, hm, visits);
    out[3 * (size_t)i] = E.x; out[3 * (size_t)i + 1] = E.y; out[3 * (size_t)i + 2] = E.z;
  }
}

// ------------------------------------------------------------------------------------------------ host state
struct PhotonState {
  bool built = false;
  uint32_t key[8] = {0};   // parameters the maps were built with
  void* d_map[2] = {nullptr, nullptr};
  int32_t count[2] = {0, 0};
  uint32_t emitted = 0;
  std::vector<rt_photon> raw[2];  // insertion order, powers scaled (what GetPhotonMap()'s array holds before balancing)
};

static inline void photon_state_free(PhotonState& s) {
  for (int i = 0; i < 2; i++) {
    if (s.d_map[i]) (void)hipFree(s.d_map[i]);
    s.d_map[i] = nullptr;
    s.count[i] = 0;
    s.raw[i].clear();
  }
  s.built = false;
}
