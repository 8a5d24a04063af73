// Photon pass and photon-map irradiance gather (reference RayTracer.cpp:262-334, photonmap.h, materials.cpp:315-390).
//
//   k_photon_trace   RayTracer::GeneratePhotons' emission + TracePhoton's random walk, one photon path per lane,
//                    every draw from the Philox stream (photon index, bounce) so the maps are reproducible
//                    (SURVEY Q14).  Deposits are appended unordered with their (path, bounce) key; the host orders
//                    them, keeps the first n per map (exactly the photons a sequential emitter would have kept,
//                    photonmap.h:158-172) and balances the kd-tree (photonmap.h:176-264).
//   k_photon_gather  PhotonMap::EstimateIrradiance<k> (photonmap.h:268-414) for the shade kernel's gather queue, one
//                    lookup per lane.  The reference heap's first replacement evicts its root whatever the newcomer's
//                    distance (dist2[0] is only refreshed on replacement, photonmap.h:383-397), so its result depends
//                    on the visiting order; the kernel computes that same photon set by threshold counting over
//                    fixed-radius walks of the same kd-tree plus a short walk in the reference's order (see "gather"
//                    below), from per-pixel hints where it has them.  Walk stack and bracket list live in LDS,
//                    [entry][lane], which is bank-conflict free for per-lane indices.
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
  int32_t ext_materials;   // RT_EXT_MATERIALS: this build's MtlPhong / MtlMicrofacet::GenerateSample (the reference's absorb)
  int32_t n_plights;
  int32_t accept[2];       // 0: that map already holds its n photons from earlier batches
  uint32_t cap;            // record capacity per map
};
struct PhotonOut {
  float4* rec[2];          // 2 float4 per record
  uint2* key[2];           // (photon path - first, ordinal among the path's deposits into this map)
  uint2* per_path;         // [count]: deposits of each path into the global / the caustics map
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
  // One loop of bounce steps: a lane whose path has ended starts its next one in the same step, instead of idling until the
  // longest path of the wave's current 64 is through (paths are 2.3 rays long on average and up to photon_max_bounces: with
  // a loop of paths around a loop of bounces 19 % of the lanes were active).
  uint32_t t = RT_TID;
  uint32_t e = 0;
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), c = v3(0, 0, 0);
  int prevLobe = 0, bounce = 0;
  uint32_t n_dep[2] = {0u, 0u};   // deposits of this path so far, per map: a deposit's place in insertion order follows from
                                  // (path, ordinal) and the counts of the paths before it (k_photon_place)
  auto start_path = [&]() {
    e = PP.first + t;
    const DLight& L = S.lights[O.plights[e % (uint32_t)PP.n_plights]];  // round robin, RayTracer.cpp:278,286
    // PointLight::RandomPhoton, lights.cpp:74-109
    uint32_t r[4];
    philox4x32_10(e, 0xFFFFFFFFu, 0, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r);
    float u1 = u32_to_unit(r[0]), u2 = u32_to_unit(r[1]), u3 = u32_to_unit(r[2]), u4 = u32_to_unit(r[3]);
    float phi = two_pi_times(u2);
    float cosTheta = 1.0f - 2.0f * u1;
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    V3 spherePoint = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    o = v3(L.position[0], L.position[1], L.position[2]) + (spherePoint * L.size);
    float r_hemi = sqrtf(u3);
    float phi_hemi = two_pi_times(u4);
    V3 tt, bb;
    orthonormals(spherePoint, tt, bb);
    d = ((r_hemi * cosf(phi_hemi)) * tt) + ((r_hemi * sinf(phi_hemi)) * bb) + (sqrtf(1.0f - u3) * spherePoint);
    d = normalized(d);
    c = v3(L.intensity[0], L.intensity[1], L.intensity[2]) * (float)(4 * RT_PI_D) * L.size * L.size;
    prevLobe = 0; bounce = 0;
    n_dep[0] = n_dep[1] = 0u;
  };
  bool have = t < PP.count;
  if (have) start_path();
  while (wave_any(have)) {
    if (have) {
      // RayTracer::TracePhoton (RayTracer.cpp:296-334), iterative; MtlBlinn::GenerateSample (materials.cpp:335-390)
      bool ended = true;   // (every way out of the step but its last line ends the path)
      do {
      Hit h;
      rays++;
      if (!scene_closest<CNT>(S, o, d, 1, h, stk, stride, cnt)) break;
      if (h.inst < 0) break;  // a renderable light: hInfo.node is null (RayTracer.cpp:299)
      const DMaterial& M = S.materials[S.inst[h.inst].mtl];
      const bool micro = PP.ext_materials && M.kind == RT_MTL_MICROFACET;
      if (M.kind != 1 && !(PP.ext_materials && M.kind == RT_MTL_PHONG) && !micro) break;  // MtlPhong / MtlMicrofacet::GenerateSample return false
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
      float ggx_a2 = -1.0f;
      if (micro) {  // this build's model (oracle: microfacet_params): lobe albedos from the untextured parameters, GGX facets
        const float metal = fminf(fmaxf(ksv.x, 0.0f), 1.0f);
        float a = M.glossiness.v * M.glossiness.v;
        if (a < 1e-3f) a = 1e-3f;
        ggx_a2 = a * a;
        const float rr = (1.0f - M.ior) / (1.0f + M.ior);
        const float f0d = rr * rr;
        ksv = v3(f0d, f0d, f0d) * (1.0f - metal) + kdv * metal;
        const V3 tr = rfv;
        rfv = tr * (1.0f - metal);
        kdv = (kdv * (1.0f - metal)) * (v3(1, 1, 1) - tr);
      }
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
        newDir = reflect_dir(S, sp, M.glossiness.v, u32_to_unit(r1[0]), u32_to_unit(r1[1]), ggx_a2);
        mult = ksv * dot(newDir, sp.N);
        prob = reflectProb;
        lobe = 2;
      } else if (random < diffuseProb + reflectProb + refractProb) {
        philox4x32_10(e, (uint32_t)bounce, 1, RT_TAG_PHOTON, PP.seed_lo, PP.seed_hi, r1);
        float ra[2] = {u32_to_unit(r1[0]), u32_to_unit(r1[1])}, rb[2] = {u32_to_unit(r1[2]), u32_to_unit(r1[3])};
        V3 ro;
        refract_ray(S, sp, M.ior, M.glossiness.v, ra, rb, ro, newDir, ggx_a2);  // only the direction is used (materials.cpp:380)
        mult = rfv * fabsf(dot(newDir, sp.N));
        prob = refractProb;
        lobe = 4;
      } else {
        break;  // absorbed
      }
      V3 newC = (c * mult) / prob;
      {
        // a diffuse bounce deposits: into the caustics map after a specular / transmission bounce.  The record slot comes from
        // one atomic per wave and map (the lanes that deposit in this step are counted with a ballot): one atomic per deposit
        // on the map's single counter is what made a batch that fills the denser map ten times over take 11 ms.
        const int which = (prevLobe & (4 | 2)) ? 1 : 0;
        const bool d0 = (lobe & 1) && which == 0 && PP.accept[0];
        const bool d1 = (lobe & 1) && which == 1 && PP.accept[1];
        uint32_t at = seg_push(d0, O.count, 0u, PP.cap, O.overflow);
        const uint32_t at1 = seg_push(d1, O.count + 1, 0u, PP.cap, O.overflow);
        if (d1) at = at1;
        if (d0 || d1) {
          if (at != RT_NONE) {
            float4 a, b;
            photon_pack(sp.P, -newDir, newC, a, b);
            O.rec[which][2 * (size_t)at] = a;
            O.rec[which][2 * (size_t)at + 1] = b;
            uint2 k;
            k.x = t; k.y = which ? n_dep[1] : n_dep[0];
            O.key[which][at] = k;
          }
          if (which) n_dep[1]++; else n_dep[0]++;
        }
      }
      prevLobe = lobe;
      o = h.p;
      d = newDir;
      c = newC;
      bounce++;
      ended = bounce >= PP.max_bounces;
      } while (false);
      if (ended) {
        uint2 nd;
        nd.x = n_dep[0]; nd.y = n_dep[1];
        O.per_path[t] = nd;
        t += RT_NTHREADS;
        have = t < PP.count;
        if (have) start_path();
      }
    }
  }
  if (CNT) {
    cnt.flush(O.counters);
    if (rays) rt_atomic_add_u64(&O.counters->rays_photon, rays);
  }
}

// ------------------------------------------------------------------------------------------------ gather
// PhotonMap::IrradianceEstimate<true, k, CONSTANT> (photonmap.h:268-309) over LocatePhotons<true> (:339-414).
//
// What the reference computes.  Call a photon ELIGIBLE for a query when its squared distance and (for ellipticity != 1)
// its ellipsoidal squared distance d' are below radius^2 and its direction faces the surface (dir . n < 0).
// LocatePhotons keeps the first k eligible photons it visits, heapifies them, and from the (k+1)-th on replaces the
// heap's root whenever a photon is closer than the current bound.  The bound (dist2[0]) stays radius^2 until that first
// replacement (:375-411), so the first replacement evicts the root WHATEVER the newcomer's distance.  Consequences:
//   |U| <= k                 the result is all of U, area = pi * radius^2;
//   otherwise                the result is the k nearest of U and the bound is the k-th smallest d', EXCEPT when the
//                            first k eligible photons in visiting order are exactly the k nearest: then the (k+1)-th
//                            visited (farther than all of them) evicts the k-th nearest, and the result is the k-1
//                            nearest plus the (k+1)-th nearest, with the bound at the (k+1)-th smallest d'.
// The device computes exactly that set without the per-query heap (k = 128 entries x 8 bytes per lane is what capped the
// heap kernel at two waves per CU and 8 G node visits/s): the (k+1)-th smallest d' is found by counting eligible
// photons below 16 thresholds per fixed-radius walk of the same kd-tree (the bracket narrows 16x per walk, starting
// with radius^2 / 4), the last walk sums the powers below the bracket and notes the bracket's <= 16 photons in the
// lane's LDS list (sorted after the walk), and a walk in the reference's visiting order that stops at the first
// eligible photon beyond the k-th distance decides the exception.  The photon SET is the reference's (equal d' values
// among different photons are resolved by visiting order here and by heap history there); the power sum runs in
// visiting order instead of heap order, i.e. irradiance agrees to float-sum reordering (tests: 2e-5 relative).
#ifndef RT_GATHER_STATS
#define RT_GATHER_STATS 0   // developer tallies (scripts/dev/gather_stats.sh): photon_visits counts one kind of walk only — 1 the hinted
#endif                      // collecting walk, 2 the exception / tie walks, 3 counting walks, 4 the collecting walk after them; 5: gathers = stage-2 lookups; 6 / 7: hinted collecting / counting walks over the full radius
#define RT_GS_V(kind) (RT_GATHER_STATS == 0 || RT_GATHER_STATS == 5 || RT_GATHER_STATS == (kind) || RT_GATHER_STATS == 10 + (kind) ? visits : gs_dummy)
#define RT_GS_LONG 2000   // 11..14: kinds 1..4 again, but only the walks of more than this many visits
struct HeapMem {     // per-lane views of the lookup's LDS: entry i of this lane is at [i * stride]
  uint32_t* walk;    // kd walk stack: node index << 4 | split axis << 2 | state (pre-order walks: the node index alone)
  uint32_t* list;    // RT_SEL entries: the first of the collecting walk's bracket photons, in visiting order
  uint32_t stride;
};

RT_DEV V3 photon_power(float4 pa, float4 pb) {  // PhotonData::GetPower = Color24::ToColor() * power
  uint32_t bits = __float_as_uint(pb.w);
  return v3((float)(bits & 255u) / 255.0f, (float)((bits >> 8) & 255u) / 255.0f, (float)((bits >> 16) & 255u) / 255.0f) * pa.w;
}

RT_DEV V3 photon_power_raw(float4 pa, float4 pb) {  // 255 x GetPower: the division is applied once to a sum of these
  uint32_t bits = __float_as_uint(pb.w);
  return v3((float)(bits & 255u), (float)((bits >> 8) & 255u), (float)((bits >> 16) & 255u)) * pa.w;
}

// Visits, in LocatePhotons' order, every photon of the kd-tree that is eligible under the FIXED squared radius `bound`
// and calls f(d', index, pa, pb); f returns true to stop the walk.  One `visits` per node entered (photonmap.h:340).
template <class F>
RT_DEV void photon_walk_impl(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float bound, const HeapMem& hm, uint32_t& visits, F&& f) {
  const uint32_t hs = hm.stride;
  int sp = 0;
  uint32_t index = 1;
  bool enter = true;
  for (;;) {
    float4 pa = M.ph[2 * (size_t)index];
    float4 pb = M.ph[2 * (size_t)index + 1];
    if (enter) {
      visits++;
      if ((int)index < M.half) {  // internal node: nearer child first (photonmap.h:346-355)
        int axis = (int)((__float_as_uint(pb.w) >> 24) & 3u);
        float pc = axis == 0 ? pa.x : (axis == 1 ? pa.y : pa.z);
        float qc = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
        float dist = qc - pc;
        hm.walk[sp * hs] = (index << 4) | ((uint32_t)axis << 2) | 1u;
        sp++;
        index = dist > 0.0f ? 2 * index + 1 : 2 * index;
        continue;
      }
    }
    {  // photonmap.h:357-373
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
        if (keep && f(dist2, index, pa, pb)) return;
      }
    }
    for (;;) {  // back up: second child if the splitting plane is within the radius, then the node's own photon
      if (sp == 0) return;
      uint32_t w = hm.walk[(sp - 1) * hs];
      index = w >> 4;
      if ((w & 3u) == 1u) {
        // the signed distance to the node's splitting plane, recomputed from the node's photon as on the way down
        int axis = (int)((w >> 2) & 3u);
        float pc = ((const float*)M.ph)[8 * (size_t)index + axis];
        float qc = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
        float dist = qc - pc;
        if (dist * dist < bound) {
          hm.walk[(sp - 1) * hs] = (w & ~3u) | 2u;
          index = dist > 0.0f ? 2 * index : 2 * index + 1;
          enter = true;
          break;
        }
      }
      sp--;
      enter = false;  // process photon `index` itself
      break;
    }
  }
}

// Eligibility of one photon for a query under the squared radius `bound` (photonmap.h:357-373); d2 is its (ellipsoidal)
// squared distance.  One definition, so that a distance recomputed later is the same float.
RT_DEV bool photon_eligible(float4 pa, float4 pb, V3 pos, V3 normal, float normScale, float bound, float& d2) {
  V3 dif = v3(pa.x, pa.y, pa.z) - pos;
  float dist2 = dot(dif, dif);
  if (!(dist2 < bound)) return false;
  V3 dir = v3(pb.x, pb.y, pb.z);
  if (dot(dir, normal) >= 0.0f) return false;
  if (normScale > 0.0f) {
    float perp = dot(dif, normal);
    dif = dif + normal * (perp * normScale);
    dist2 = dot(dif, dif);
    if (dist2 >= bound) return false;
  }
  d2 = dist2;
  return true;
}

// The same set of photons in pre-order (node, nearer child, farther child): with a FIXED radius the decision to enter
// the farther child does not depend on what the nearer subtree held, so a node is fetched once and the stack holds
// nothing but the farther children still to enter.  For the counting / collecting walks, whose results do not depend on
// the order.  f returns the squared radius to go on with: the same, a smaller one (what was found so far makes the rest of
// the larger ball irrelevant; nodes already on the stack are still entered, under the new radius), or a negative value
// to stop at once.  Without this a lookup whose starting radius suits a sparser place than the one it is at — a stale hint
// at the edge of a caustic, the default first guess inside one — walks every photon of that ball: tens of thousands of
// dependent fetches on one lane, milliseconds for the whole launch.
template <class F>
RT_DEV void photon_walk_pre_impl(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float bound, const HeapMem& hm, uint32_t& visits, F&& f) {
  const uint32_t hs = hm.stride;
  int sp = 0;
  uint32_t index = 1;
  float4 pa = M.ph[2], pb = M.ph[3];
  for (;;) {
    visits++;
    uint32_t next = 0;
    if ((int)index < M.half) {
      int axis = (int)((__float_as_uint(pb.w) >> 24) & 3u);
      float pc = axis == 0 ? pa.x : (axis == 1 ? pa.y : pa.z);
      float qc = axis == 0 ? pos.x : (axis == 1 ? pos.y : pos.z);
      float dist = qc - pc;
      next = dist > 0.0f ? 2 * index + 1 : 2 * index;
      if (dist * dist < bound) hm.walk[(sp++) * hs] = next ^ 1u;  // the sibling
    }
    if (next == 0 && sp > 0) next = hm.walk[(--sp) * hs];
    // the next node's photon is requested before this one is looked at: the walk is a chain of dependent fetches
    // (the last node requests itself again: an unconditional load is cheaper than a branch around it in every step)
    const uint32_t fetch = next != 0 ? next : index;
    const float4 na = M.ph[2 * (size_t)fetch];
    const float4 nb = M.ph[2 * (size_t)fetch + 1];
    // photon_eligible's tests, all evaluated and then combined: in a wave some lane passes each of them nearly always, so
    // the nested branches only added the bookkeeping of their masks (same operations, same floats)
    float dist2;
    bool ok;
    {
      V3 dif = v3(pa.x, pa.y, pa.z) - pos;
      const float d_eu = dot(dif, dif);
      const float facing = dot(v3(pb.x, pb.y, pb.z), normal);
      const float perp = dot(dif, normal);
      dif = dif + normal * (perp * normScale);
      const float d_el = dot(dif, dif);
      ok = d_eu < bound && !(facing >= 0.0f);
      dist2 = d_eu;
      if (normScale > 0.0f) { ok = ok && !(d_el >= bound); dist2 = d_el; }
    }
    if (ok) {
      bound = f(dist2, index, pa, pb);
      if (bound < 0.0f) return;
    }
    if (next == 0) return;
    index = next;
    pa = na; pb = nb;
  }
}

template <class F>
RT_DEV void photon_walk(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float bound, const HeapMem& hm, uint32_t& visits, F&& f) {
#if RT_GATHER_STATS >= 10
  uint32_t v = 0;
  photon_walk_impl(M, pos, normal, normScale, bound, hm, v, f);
  if (v > RT_GS_LONG) visits += v;
#else
  photon_walk_impl(M, pos, normal, normScale, bound, hm, visits, f);
#endif
}
template <class F>
RT_DEV void photon_walk_pre(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float bound, const HeapMem& hm, uint32_t& visits, F&& f) {
#if RT_GATHER_STATS >= 10
  uint32_t v = 0;
  photon_walk_pre_impl(M, pos, normal, normScale, bound, hm, v, f);
  if (v > RT_GS_LONG) visits += v;
#else
  photon_walk_pre_impl(M, pos, normal, normScale, bound, hm, visits, f);
#endif
}

#ifndef RT_TEST_NO_EVICT_RULE
#define RT_TEST_NO_EVICT_RULE 0   // tests build a variant with 1 to prove that the exception below is exercised
#endif
#define RT_SEL 16   // thresholds per counting walk = photons kept sorted by the collecting walk
#define RT_LIST 32  // photons of a bracket the collecting walk can note: RT_SEL in the lane's LDS list, the rest in private memory
#ifndef RT_HINT_LO
#define RT_HINT_LO 6.0f   // a hinted bracket reaches this many photons below the hint and RT_HINT_HI above it (in units of the hint /
#define RT_HINT_HI 7.0f   // (k + 1)).  C4 step with (lo, hi) = (4,4) / (5,5) / (6,6) / (6,7) / (7,6) / (6,8) / (7.5,9) / (9,11): 30.0 / 29.1 /
#endif                    // 29.0 / 28.6 / 28.7 / 29.0 / 29.4 / 31.2 ms (with the 16-entry list of round 2, +-4 was the best: wider overflowed it)

// What the collecting walk over the bracket [lo, hi) leaves behind.
struct Collected {
  float ed[RT_SEL];      // the bracket's photons sorted by d' (ties in visiting order): the RT_SEL smallest
  uint32_t ei[RT_SEL];
  V3 raw;                // sum of (8-bit colour) x power below the bracket; Color24::ToColor's / 255 is applied to the sum
  uint32_t n_below, n_all;
  float lo, hi;          // the bracket
  float max_d;           // largest d' among the photons below the bracket, and which photon that is
  uint32_t max_i;
};

// SORT_WIDE false: a bracket of more than RT_SEL photons is only counted (the hinted lookup rejects it afterwards), which
// keeps the sorted list out of the walk's registers.
// `give_up`: the walk stops as soon as more photons than that lie below the bracket (the hinted lookup's bracket is then
// too high, whatever else the ball holds).
template <bool SORT_WIDE>
RT_DEV void photon_collect(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float lo, float hi, uint32_t give_up, const HeapMem& hm,
                           uint32_t& visits, Collected& C) {
#pragma unroll
  for (int i = 0; i < RT_SEL; i++) { C.ed[i] = RT_BIG; C.ei[i] = 0; }
  C.raw = v3(0, 0, 0);
  C.n_below = 0; C.n_all = 0;
  C.lo = lo; C.hi = hi;
  C.max_d = -1.0f; C.max_i = 0;
  auto insert_sorted = [&](float nd, uint32_t ni) {
#pragma unroll
    for (int i = 0; i < RT_SEL; i++) {
      bool sw = nd < C.ed[i];
      float td = C.ed[i]; uint32_t ti = C.ei[i];
      C.ed[i] = sw ? nd : td; C.ei[i] = sw ? ni : ti;
      nd = sw ? td : nd; ni = sw ? ti : ni;
    }
  };
  // The walk only notes which photons fall into the bracket (their indices, in visiting order, in the lane's LDS list):
  // sorting them inside the walk would issue the whole insertion network whenever ANY lane of the wave meets one.  They
  // are fetched again and sorted after the walk, all lanes together.  A bracket of more than RT_SEL photons (thresholds
  // that cannot separate equal d' values) keeps its RT_SEL smallest by sorting inside the walk, as before.
  uint32_t n_list = 0;
  uint32_t more[RT_LIST - RT_SEL];   // entries RT_SEL .. RT_LIST - 1 of the list (private memory: written for the wider brackets only)
  auto list_at = [&](uint32_t i) { return i < RT_SEL ? hm.list[i * hm.stride] : more[i - RT_SEL]; };
  bool spilled = false;
  photon_walk_pre(M, pos, normal, normScale, hi, hm, visits, [&](float d2, uint32_t index, float4 pa, float4 pb) {
    C.n_all++;
    if (d2 < lo) {
      C.n_below++;
      C.raw = C.raw + photon_power_raw(pa, pb);
      if (d2 > C.max_d) { C.max_d = d2; C.max_i = index; }
      return C.n_below > give_up ? -1.0f : hi;
    }
    if (n_list < RT_LIST) {
      if (n_list < RT_SEL) hm.list[n_list * hm.stride] = index; else more[n_list - RT_SEL] = index;
      n_list++;
      return hi;
    }
    if (!SORT_WIDE) return hi;
    if (!spilled) {   // the list is full: sort what it holds, then carry on sorting in place
      spilled = true;
      for (uint32_t i = 0; i < RT_LIST; i++) {
        uint32_t pi = list_at(i);
        float4 qa = M.ph[2 * (size_t)pi], qb = M.ph[2 * (size_t)pi + 1];
        float qd = 0.0f;
        (void)photon_eligible(qa, qb, pos, normal, normScale, hi, qd);
        insert_sorted(qd, pi);
      }
    }
    insert_sorted(d2, index);
    return hi;
  });
  if (!spilled) {
    for (uint32_t i = 0; i < RT_LIST; i++) {
      if (i >= n_list) break;
      uint32_t pi = list_at(i);
      float4 qa = M.ph[2 * (size_t)pi], qb = M.ph[2 * (size_t)pi + 1];
      float qd = 0.0f;
      (void)photon_eligible(qa, qb, pos, normal, normScale, hi, qd);
      insert_sorted(qd, pi);
    }
  }
}

// The estimate from a collected bracket with C(lo) <= k < C(hi) (or, `all`, from everything within the radius).
// `hint` out: the (k+1)-th smallest d' (radius^2 when at most k photons are in range).
RT_DEV V3 photon_finish(const DPhotonMap& M, V3 pos, V3 normal, float normScale, float r2, uint32_t K, bool all, const HeapMem& hm,
                        uint32_t& visits, const Collected& C, float& guess, float& hint) {
  if (C.n_all == 0) { hint = r2; return v3(0, 0, 0); }
  uint32_t gs_dummy = 0; (void)gs_dummy;
  V3 sum = C.raw / 255.0f;
  float bound, next_hint;
  if (all) {  // |U| <= k: everything was summed (lo = radius^2), area from the search radius (dist2[0] never changes)
    bound = r2;
    next_hint = r2;
  } else {
    // ranks n_below + 1 .. in the bracket list; rank k is the last one kept, rank k + 1 decides the exception
    const uint32_t j_k = K - 1 - C.n_below, j_k1 = K - C.n_below;  // list positions of the k-th and (k+1)-th nearest
    float d_k = C.max_d, d_k1 = RT_BIG;
    uint32_t i_k = C.max_i, i_k1 = 0;
    bool have_k1 = false;
    if (C.n_below < K && j_k >= RT_SEL) {
      // The k-th nearest lies beyond the sorted list: the bracket holds more than RT_SEL photons that no threshold could
      // separate — equal d' values (photons at one position), or values a float apart.  Which of them belong to the k
      // nearest is then a matter of ties (resolved by heap history in the reference, photonmap.h:375-411): take the
      // first k - n_below of them in visiting order, with the bound at the bracket's upper end (within float spacing of
      // every one of them).  No exception test: it needs the k-th distance exactly.
      uint32_t need = K - C.n_below, taken = 0;
      V3 extra = v3(0, 0, 0);
      photon_walk_pre(M, pos, normal, normScale, C.hi, hm, RT_GS_V(2), [&](float d2, uint32_t, float4 pa, float4 pb) {
        if (d2 >= C.lo && taken < need) { extra = extra + photon_power(pa, pb); taken++; }
        return C.hi;
      });
      sum = sum + extra;
      guess = C.hi * 1.25f;
      hint = C.hi;
      float area_w = RT_PI_F * C.hi;
      if (area_w > 0.0f) sum = sum * (1.0f / area_w);
      return sum;
    }
#pragma unroll
    for (int i = 0; i < RT_SEL; i++) {
      if ((uint32_t)i <= j_k && C.n_below < K && C.ed[i] < RT_BIG) {
        sum = sum + photon_power(M.ph[2 * (size_t)C.ei[i]], M.ph[2 * (size_t)C.ei[i] + 1]);
        if ((uint32_t)i == j_k) { d_k = C.ed[i]; i_k = C.ei[i]; }
      }
      if ((uint32_t)i == j_k1 && C.ed[i] < RT_BIG) { d_k1 = C.ed[i]; i_k1 = C.ei[i]; have_k1 = true; }
    }
    bound = d_k;
    next_hint = have_k1 ? d_k1 : d_k;
    if (have_k1) {
      // the exception: the first k eligible photons in visiting order are the k nearest (none beyond the k-th distance)
      uint32_t seen = 0;
      bool beyond = false;
      photon_walk(M, pos, normal, normScale, r2, hm, RT_GS_V(2), [&](float d2, uint32_t, float4, float4) {
        if (d2 > d_k) { beyond = true; return true; }
        return ++seen == K;
      });
      if (!beyond && seen == K && !RT_TEST_NO_EVICT_RULE) {
        sum = sum - photon_power(M.ph[2 * (size_t)i_k], M.ph[2 * (size_t)i_k + 1]);
        sum = sum + photon_power(M.ph[2 * (size_t)i_k1], M.ph[2 * (size_t)i_k1 + 1]);
        bound = d_k1;
      }
    }
  }
  guess = all ? r2 : bound * 1.25f;
  hint = next_hint;
  float area = RT_PI_F * bound;
  if (area > 0.0f) sum = sum * (1.0f / area);
  return sum;
}

// The lookup from scratch.  `guess`: in, a squared radius expected to hold about k + 1 eligible photons (<= 0: none
// yet); out, what this query needed.  Only the number of walks depends on it, never the result.  `hint`: out, see
// photon_finish.
RT_DEV V3 photon_estimate(const DPhotonMap& M, V3 pos, V3 normal, float radius, float ellipticity, int k, const HeapMem& hm,
                          uint32_t& visits, float& guess, float& hint) {
  if (M.stored == 0) { hint = 0.0f; return v3(0, 0, 0); }
  const float normScale = ellipticity == 1.0f ? 0.0f : 1.0f / ellipticity - 1.0f;
  const float r2 = radius * radius;
  const uint32_t K = (uint32_t)k;
  uint32_t gs_dummy = 0; (void)gs_dummy;
  if (RT_GATHER_STATS == 16) visits += 1u;
  // ---- bracket [lo, hi) of the (k+1)-th smallest d': C(lo) <= k and C(hi) >= k + 1, C(t) = #{eligible, d' < t}
  // (a guess far below the radius — the lane's previous lookup sat in a dense cluster — is raised to a floor: every step
  // up from it costs a walk)
  float lo = 0.0f, hi = guess > 0.0f && guess < r2 ? fmaxf(guess, r2 * 1e-9f) : r2 * 0.25f;
  uint32_t c_lo = 0, c_hi = 0;
  bool hi_known = false, all = false;
  // Invariants: C(lo) = c_lo <= k always; after a counting walk that does not grow, C(hi) = c_hi >= k + 1.  Growing steps
  // are not limited: each multiplies hi by >= 1.25 and the step that reaches radius^2 ends the search either way.  Only
  // narrowing steps count against the cap (16x each: twelve of them are finer than float spacing).
  for (int narrowed = 0;;) {
    if (hi_known && c_hi - c_lo <= RT_SEL) break;
    if (hi_known && narrowed >= 12) break;
    float t[RT_SEL];
    uint32_t c[RT_SEL];
    const float step = (hi - lo) * (1.0f / RT_SEL);
#pragma unroll
    for (int i = 0; i < RT_SEL; i++) { t[i] = lo + step * (float)(i + 1); c[i] = 0; }
    t[RT_SEL - 1] = hi;
    uint32_t below = 0;
    if (RT_GATHER_STATS == 15) visits += 1u;   // 15: the number of counting walks; 16: of lookups from scratch
    float wb = hi;   // the walk's radius, shrinking
    photon_walk_pre(M, pos, normal, normScale, hi, hm, RT_GATHER_STATS == 7 && hi == r2 ? visits : RT_GS_V(3), [&](float d2, uint32_t, float4, float4) {
      if (d2 >= lo) {
#pragma unroll
        for (int i = 0; i < RT_SEL; i++) c[i] += d2 < t[i] ? 1u : 0u;
      } else {
        below++;
      }
      // more than k photons below one of the lower thresholds already: the (k+1)-th smallest d' lies below it, and the
      // rest of the walk only has to complete the counts up to there (the counts above it stay incomplete and unused:
      // the narrowing below picks the lowest threshold with more than k, which is this one or a lower one)
      float go_on = wb;
      if (below + c[RT_SEL / 2 - 1] > K) go_on = fminf(go_on, t[RT_SEL / 2 - 1]);
      if (below + c[RT_SEL / 4 - 1] > K) go_on = fminf(go_on, t[RT_SEL / 4 - 1]);
      if (below + c[0] > K) go_on = fminf(go_on, t[0]);
      wb = go_on;
      return go_on;
    });
    c_lo = below;
    c_hi = below + c[RT_SEL - 1];
    hi_known = true;
    if (c_hi < K + 1) {
      if (hi == r2) { all = true; lo = r2; break; }  // at most k eligible photons in all: the result is all of them
      lo = hi; c_lo = c_hi;
      // too small: grow by the areal extrapolation of the count (photons lie on surfaces), with a margin
      float grow = c_hi >= 8 ? 1.3f * (float)(K + 1) / (float)c_hi : 4.0f;
      hi = hi * (grow > 1.25f ? grow : 1.25f);
      if (!(hi < 0.8f * r2)) hi = r2;
      hi_known = false;
      continue;
    }
    narrowed++;
    float nlo = lo, nhi = hi;
    uint32_t nclo = c_lo, nchi = c_hi;
#pragma unroll
    for (int i = RT_SEL - 2; i >= 0; i--)
      if (below + c[i] >= K + 1 && t[i] < hi) { nhi = t[i]; nchi = below + c[i]; }
#pragma unroll
    for (int i = 0; i < RT_SEL - 1; i++)
      if (below + c[i] <= K && t[i] < nhi) { nlo = t[i]; nclo = below + c[i]; }
    if (nlo == lo && nhi == hi) break;     // thresholds no longer separate anything (equal d' values): collect as is
    lo = nlo; hi = nhi; c_lo = nclo; c_hi = nchi;
  }
  // ---- collect: powers below the bracket, the bracket's photons sorted by d' (ties in visiting order)
  Collected C;
  photon_collect<true>(M, pos, normal, normScale, lo, hi, 0xffffffffu, hm, RT_GS_V(4), C);
  return photon_finish(M, pos, normal, normScale, r2, K, all, hm, visits, C, guess, hint);
}

// The lookup from a hint: the (k+1)-th smallest d' that a lookup at (nearly) the same place found before (>= radius^2:
// it found at most k photons within the radius).  The collecting walk runs at once, on a narrow bracket around the
// hint, and the bracket is checked afterwards against what the counting walks would have established.  true: E is the
// estimate (the same photon set as photon_estimate's) at the price of one fixed-radius walk and the short walk of the
// exception test.  false: the bracket missed; `hint` then holds MINUS a squared radius to start the counting walks
// from.  Most lookups of a frame's later passes — and, within a pass, of a pixel's later samples — succeed.
RT_DEV bool photon_estimate_hinted(const DPhotonMap& M, V3 pos, V3 normal, float radius, float ellipticity, int k, const HeapMem& hm,
                                   uint32_t& visits, float& hint, V3& E) {
  if (M.stored == 0) { E = v3(0, 0, 0); return true; }
  const float normScale = ellipticity == 1.0f ? 0.0f : 1.0f / ellipticity - 1.0f;
  const float r2 = radius * radius;
  const uint32_t K = (uint32_t)k;
  const float delta_lo = fminf(0.25f, RT_HINT_LO / (float)(K + 1));   // about RT_HINT_LO + RT_HINT_HI photons in the bracket
  const float delta_hi = fminf(0.25f, RT_HINT_HI / (float)(K + 1));
  float lo, hi;
  if (hint * (1.0f + delta_hi) >= r2) {
    lo = hi = r2;        // expect at most k photons within the radius (the bracket may not reach beyond it)
  } else {
    lo = hint * (1.0f - delta_lo);
    hi = hint * (1.0f + delta_hi);
  }
  Collected C;
  uint32_t gs_dummy = 0; (void)gs_dummy;
  photon_collect<false>(M, pos, normal, normScale, lo, hi, K, hm, RT_GATHER_STATS == 6 && hi == r2 ? visits : RT_GS_V(1), C);
  if (RT_GATHER_STATS == 8 && hi == r2) visits += 1u;          // 8: the number of such walks, 9: of those that find more than k photons
  if (RT_GATHER_STATS == 9 && hi == r2 && C.n_all > K) visits += 1u;
  bool all = false;
  if (C.n_below > K) {   // (the walk gave up there) the answer lies below lo
    hint = -(hi == r2 ? 0.5f * r2 : lo);
    return false;
  }
  if (hi == r2) {
    all = true;
  } else if (!(C.n_below <= K && C.n_all >= K + 1 && C.n_all - C.n_below <= RT_LIST && K - C.n_below < RT_SEL)) {
    // (all of the bracket's photons noted, and the (k+1)-th nearest among the RT_SEL smallest of them, which is what gets sorted)
    // what this walk saw seeds the counting walks: the answer lies below hi, or above it
    float grow = C.n_all >= 8 ? 1.3f * (float)(K + 1) / (float)C.n_all : 4.0f;
    hint = -(C.n_all >= K + 1 ? hi : fminf(hi * (grow > 1.25f ? grow : 1.25f), 0.999f * r2));
    return false;
  }
  float guess_unused;
  E = photon_finish(M, pos, normal, normScale, r2, K, all, hm, visits, C, guess_unused, hint);
  return true;
}

#ifndef RT_EMU
#define RT_GATHER_BLOCK 64
// dynamic LDS: walk[depth][64], list[RT_SEL][64]
#define RT_HEAP_DECL(depth_)                                                               \
  extern __shared__ uint32_t s_heap_raw[];                                                 \
  HeapMem hm;                                                                              \
  hm.stride = RT_GATHER_BLOCK;                                                             \
  hm.walk = s_heap_raw + threadIdx.x;                                                      \
  hm.list = s_heap_raw + (size_t)(depth_) * RT_GATHER_BLOCK + threadIdx.x;
static inline size_t photon_heap_bytes(int depth) { return (size_t)(depth + RT_SEL) * RT_GATHER_BLOCK * 4; }
#else
#define RT_GATHER_BLOCK 1
#define RT_HEAP_DECL(depth_)                                                               \
  static uint32_t e_walk[RT_PHOTON_TSTACK];                                                \
  static uint32_t e_list[RT_SEL];                                                          \
  HeapMem hm;                                                                              \
  hm.stride = 1; hm.walk = e_walk; hm.list = e_list;
static inline size_t photon_heap_bytes(int) { return 0; }
#endif
// levels of a heap-ordered kd-tree over n photons (+1 spare): the walk stack holds one entry per level
static inline int photon_tree_depth(int n) {
  int d = 1;
  while ((1ll << d) <= (long long)n) d++;
  return d + 1;
}

struct GatherP {
  DPhotonMap map[2];
  float radius;
  float ellipticity[2];
  int32_t k;
  int32_t depth;   // walk stack entries per lane
  uint32_t n_pix;  // pixels of the pass (path slot -> pixel)
  uint32_t n_hint_px;  // W * H: entries per map of `hints`
  int32_t pilot;       // STAGE 1: 0 every lookup; 1 list the pilot lookups only; 2 every lookup but the pilots
  float* hints;    // [W * H][2]: per pixel and map, the (k+1)-th smallest d' of the last lookup (0: none); null without
                   // hints (STAGE 0)
  uint32_t* retry;          // STAGE 1 -> 2: queue entries still to do, segmented like the queue itself
  float* retry_seed;        // ... and, for a lookup whose hinted bracket missed, where to start from (0: nothing known)
  uint32_t* retry_counts;
  uint32_t* cursor;         // per segment (RT_CSTRIDE apart): entries of the stage's queue already claimed by some wave
};

// the shade kernel's deferred lookups (materials.cpp:315-322): indirect += (1/pi) * kd * E, weighted by the path weight
#ifndef RT_GATHER_WAVES
#define RT_GATHER_WAVES 4
#endif
#ifndef RT_HINT_CELLS
#define RT_HINT_CELLS 12.0f   // hint grid cells per search radius (secondary lookups)
#endif
// Where a lookup's hint lives.  Primary hits (global map): the pixel — its samples look up within a pixel footprint of
// each other.  Secondary hits (caustics map) land wherever the reflected / refracted ray goes, so they share hints by
// place instead: a hash of the lookup's grid cell (a twelfth of the search radius) and the side its normal faces, into
// the same table.  Collisions only cost a walk.
RT_DEV size_t gather_hint_slot(const GatherP& G, const FrameBufs& F, uint32_t slot, uint32_t which, V3 pos, V3 n) {
  if (which == 0) return 2 * (size_t)F.pixels[slot % G.n_pix];
  const float inv_cell = RT_HINT_CELLS / G.radius;
  auto cell = [&](float x) { return (int)fminf(fmaxf(floorf(x * inv_cell), -1.0e9f), 1.0e9f); };  // (any value will do far out)
  int ix = cell(pos.x), iy = cell(pos.y), iz = cell(pos.z);
  float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
  uint32_t side = ax >= ay && ax >= az ? (n.x < 0.0f ? 0u : 1u) : (ay >= az ? (n.y < 0.0f ? 2u : 3u) : (n.z < 0.0f ? 4u : 5u));
  uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u ^ (uint32_t)iz * 83492791u ^ side * 0x9e3779b9u;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return 2 * (size_t)(h % G.n_hint_px) + 1;
}

// STAGE 0: every lookup from scratch (no hint buffer).  With hints the queue is taken in two stages, because a wave is
// as slow as its slowest lane: STAGE 1 tries every hinted lookup (one walk + the exception test) and lists the ones
// without a hint or whose bracket missed; STAGE 2 runs the listed lookups from scratch, packed densely into waves.
// Lookups differ a lot in length, so the waves CLAIM the queue 64 entries at a time (one atomic per wave and claim, on the
// segment's cursor) instead of being dealt fixed shares: a wave that drew short lookups takes more of them, and the launch
// ends when the queue is empty, not when the unluckiest wave has finished its share.
template <bool CNT, int STAGE>
RT_KERNEL_W(RT_GATHER_WAVES, k_photon_gather, GatherP G, FrameBufs F, GatherQ Q) {
  RT_HEAP_DECL(G.depth)
  uint32_t visits = 0, gathers = 0;
  float guess[2] = {0.0f, 0.0f};  // per map: the squared radius the lane's previous lookup needed
  const uint32_t* seg_counts = STAGE == 2 ? G.retry_counts : Q.counts;
  struct { uint32_t s; } sw_;
  sw_.s = rt_wave_id() % Q.nseg;
  for (uint32_t segs_done = 0; segs_done < Q.nseg;) {
    uint32_t seg_n_ = seg_counts[sw_.s * RT_CSTRIDE];
    if (seg_n_ > Q.segcap) seg_n_ = Q.segcap;
    uint32_t e0_ = 0;
    if (rt_lane() == 0) e0_ = rt_atomic_add_u32(G.cursor + sw_.s * RT_CSTRIDE, RT_WAVE);
    e0_ = rt_wave_first(e0_);
    if (e0_ >= seg_n_) { sw_.s = (sw_.s + 1) % Q.nseg; segs_done++; continue; }
    const uint32_t seg_base_ = sw_.s * Q.segcap;
    const uint32_t seg_e = e0_ + rt_lane();
    const bool seg_valid = seg_e < seg_n_;
    uint32_t g = seg_base_ + (uint32_t)seg_e;
    if (STAGE == 2 && seg_valid) g = G.retry[g];
    bool todo = seg_valid;   // (every lane reaches the append below)
    float seed = 0.0f;       // STAGE 1 -> 2: a squared radius to start the counting walks from
    bool mine = seg_valid;
    if (STAGE == 1 && G.pilot != 0 && seg_valid) {
      // a frame's first pass has no hints yet: the lookups of every eighth camera sample go first, from scratch
      // (pilot 1: only they are listed), and leave hints for all the others (pilot 2: everything but them)
      const bool is_pilot = ((__float_as_uint(Q.a[g].w) / G.n_pix) & 7u) == 0u;
      mine = G.pilot == 2 ? !is_pilot : false;
      todo = G.pilot == 1 ? is_pilot : todo && mine;
    }
    if (mine) {
      float4 a = Q.a[g], b = Q.b[g], c = Q.c[g];
      uint32_t which = __float_as_uint(b.w) & 1u;
      const DPhotonMap& M = which ? G.map[1] : G.map[0];
      const float ell = which ? G.ellipticity[1] : G.ellipticity[0];
      // what the last lookup of this pixel into this map found (an earlier pass of the frame, or another sample of the
      // pixel in this pass): a plain load and store, any value is only a hint
      float* hp = STAGE == 0 ? nullptr : G.hints + gather_hint_slot(G, F, __float_as_uint(a.w), which, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z));
      V3 E = v3(0, 0, 0);
      if (STAGE == 1) {
        float hint = *hp;
        if (hint > 0.0f) {
          todo = !photon_estimate_hinted(M, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), G.radius, ell, G.k, hm, visits, hint, E);
          if (todo) seed = -hint;   // (the pixel's hint stays as it is for the other samples)
          else *hp = hint;
        }
      } else {
        float gs = which ? guess[1] : guess[0];
        bool done = false;
        float hint = 0.0f;
        if (STAGE == 2) {
          float sd = G.retry_seed[seg_base_ + (uint32_t)seg_e];
          if (sd > 0.0f) {
            gs = sd;
          } else {
            // no hint when the first stage looked (a frame's first pass): another sample of the pixel may have left one since
            hint = *hp;
            if (hint > 0.0f) {
              done = photon_estimate_hinted(M, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), G.radius, ell, G.k, hm, visits, hint, E);
              if (!done) gs = -hint;
            }
          }
        }
        if (!done) E = photon_estimate(M, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), G.radius, ell, G.k, hm, visits, gs, hint);
        if (STAGE == 2) *hp = hint;
        if (which) guess[1] = gs; else guess[0] = gs;
        todo = false;
      }
      if (!todo) {
        if (M.stored > 0 && (RT_GATHER_STATS != 5 || STAGE == 2)) gathers++;
        add_radiance(F, __float_as_uint(a.w), v3(c.x, c.y, c.z) * E);
      }
    }
    if (STAGE == 1) {
      uint32_t ri = seg_push(todo, G.retry_counts, sw_.s, Q.segcap, F.overflow);
      if (ri != RT_NONE) { G.retry[ri] = g; G.retry_seed[ri] = seed; }
    }
  }
  if (CNT) {
    if (visits) rt_atomic_add_u64(&F.counters->photon_visits, visits);
    if (gathers) rt_atomic_add_u64(&F.counters->gathers, gathers);
  }
}

// batched PhotonMap::EstimateIrradiance for the C-ABI (rt_photon_gather)
RT_KERNEL(k_api_photon_gather, DPhotonMap M, const float* P, const float* N, uint32_t n, float radius, float ellipticity, int k,
          int depth, float* out) {
  RT_HEAP_DECL(depth)
  uint32_t visits = 0;
  float guess = 0.0f;
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    float hint = 0.0f;
    V3 E = photon_estimate(M, v3(P[3 * (size_t)i], P[3 * (size_t)i + 1], P[3 * (size_t)i + 2]),
                           v3(N[3 * (size_t)i], N[3 * (size_t)i + 1], N[3 * (size_t)i + 2]), radius, ellipticity, k, hm, visits, guess, hint);
    out[3 * (size_t)i] = E.x; out[3 * (size_t)i + 1] = E.y; out[3 * (size_t)i + 2] = E.z;
  }
}

// ------------------------------------------------------------------------------------------------ host state
struct PhotonState {
  bool built = false;
  uint32_t key[8] = {0};   // parameters the maps were built with
  void* d_raw[2] = {nullptr, nullptr};    // insertion order, powers scaled (what GetPhotonMap()'s array holds before balancing)
  void* d_tree[2] = {nullptr, nullptr};   // the kd-tree in heap order, count + 2 entries (PrepareForIrradianceEstimation's result)
  size_t raw_cap[2] = {0, 0}, tree_cap[2] = {0, 0};   // photons the device buffers hold
  int32_t count[2] = {0, 0};
  uint32_t emitted = 0;
  uint32_t paths_needed[2] = {0, 0};   // paths it took to fill each map in the last build (0: unknown): sizes the next build's batches
};

static inline void photon_state_free(PhotonState& s) {
  for (int i = 0; i < 2; i++) {
    if (s.d_raw[i]) (void)hipFree(s.d_raw[i]);
    if (s.d_tree[i]) (void)hipFree(s.d_tree[i]);
    s.d_raw[i] = s.d_tree[i] = nullptr;
    s.raw_cap[i] = s.tree_cap[i] = 0;
    s.count[i] = 0;
  }
  s.emitted = 0;
  s.paths_needed[0] = s.paths_needed[1] = 0;
  s.built = false;
}
