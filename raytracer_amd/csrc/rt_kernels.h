// Kernels of the wavefront (queue-per-stage) integrator.
//
// One camera sample = one "path slot".  Stages, each a grid-stride kernel over a device-side queue whose length
// lives in device memory (no host round trip between stages):
//   k_raygen_trace   RunThread's camera-ray construction + the rays' first trace phase (RayTracer.cpp:138-174, 373-435)
//   k_trace_closest  Renderer::TraceRay for a queue of secondary rays, first phase; k_trace_mesh: second (BVH) phase
//                                                                   (RayTracer.cpp:373-435, objects.cpp)
//   k_shade          SendRay's dispatch + MtlBlinn::Shade + the light loop's unshadowed BRDF terms; appends
//                    refraction / reflection / indirect rays to the next level's queue and shadow work to the
//                    shadow queues                                  (RayTracer.cpp:227-259, materials.cpp:222-331)
//   k_shadow_*       PointLight/DirectLight::Illuminate's shadow rays (lights.cpp:19-72, lights.h:45)
//   k_accumulate     per-pixel sum / sum of squares / adaptive stop (RayTracer.cpp:176-198)
//   k_resolve        mean, sRGB, Color24                            (RayTracer.cpp:206-216)
// The recursion L = direct + reflection*L(refl) + refraction*(1-R)*L(refr) is linear in the children, so it is
// unrolled into per-ray weights; absorption is applied when the child's hit distance is known.
#pragma once
#include "rt_device.h"
#include "rtcore.h"   // RT_MTL_*, RT_EXT_*

struct RayQ {
  float4* q0;  // origin.xyz, weight.r
  float4* q1;  // dir.xyz,    weight.g
  float4* q2;  // parent absorption.rgb, weight.b
  uint4* q3;   // path slot, pixel index, camera sample index, meta
  float4* r0;  // closest hit as the trace kernels leave it: t, barycentric u, v (meshes), (instance + 1) | front << 31
  uint32_t* rf;  // ... and the face index (meshes); the hit attributes are computed by k_shade (finalize_hit)
  float4* h0;  // re-shade entries only (written by k_shade, lights.cpp:172-183): hit p.xyz, z
  float4* h1;  // hit N.xyz, instance (int bits)
  float4* h2;  // hit uvw.xyz, flags (bit0 front, bits 8.. light+1)
  uint32_t* counts;  // nseg segment lengths, RT_CSTRIDE words apart
  uint32_t segcap;   // entries per segment; entry index = segment * segcap + position
  uint32_t nseg;
};
// meta word: tree position (12 bits) | bounce << 12 (3 bits) | secondary << 15 | reshade << 16
#define RT_META(nid, bounce, sec, resh) ((uint32_t)(nid) | ((uint32_t)(bounce) << 12) | ((uint32_t)(sec) << 15) | ((uint32_t)(resh) << 16))

struct GroupQ {  // one record per (shade point, light) whose unshadowed contribution is non-zero
  float4* g0;    // P.xyz, light index
  float4* g1;    // weight * unshadowed BRDF term rgb, path slot
  float4* g2;    // disc offsets rx, ry (lights.cpp:25-26), lit-sample counter, unused
  float4* g3;    // soft groups: disc tangent (lights.cpp:32), light size     -- the same for every sample of the group,
  float4* g4;    // soft groups: disc bitangent, instances that may occlude      computed once by k_shade
  uint2* hint;   // soft groups: where the group's last occluded ray found its occluder (mesh instance, triangle slot) — an
                 // ordering hint for the second phase's walks (k_shadow_mesh), never read for a result; may hold anything
  uint32_t* counts;
  uint32_t segcap;
  uint32_t nseg;
};

struct GatherQ {  // photon-map lookups deferred to k_photon_gather (materials.cpp:315-322)
  float4* a;      // P.xyz, path slot
  float4* b;      // N.xyz, which map (0 global, 1 caustics)
  float4* c;      // weight * kd / pi rgb
  uint32_t* counts;
  uint32_t segcap;
  uint32_t nseg;
};

struct MeshQ {       // work items of the second (mesh) phase of the trace kernels
  uint4* item4;      // closest-hit: ray-queue entry, best t (bits), (best instance + 1) | front << 31, first candidate
                     // shadow:      group, sample index, first candidate, ray length t_max (bits)
  uint32_t* counts;  // segment lengths
  uint32_t* cursor;  // per segment, RT_CSTRIDE words apart: next unclaimed item (claimed 1..64 at a time by the walkers)
  uint32_t segcap;
  uint32_t nseg;
  float4* cand;           // candidates (CandSink, rt_device.h): the mesh instances of each item's ray, as a linked list
  uint32_t* cand_counts;  // their segment lengths
  uint32_t cand_segcap;
};
RT_DEV CandSink cand_sink(const MeshQ& MQ, uint32_t seg, uint32_t* overflow) {
  CandSink K;
  K.rec = MQ.cand;
  K.count = MQ.cand_counts + seg * RT_CSTRIDE;
  K.base = seg * MQ.cand_segcap;
  K.segcap = MQ.cand_segcap;
  K.overflow = overflow;
  return K;
}

struct PassP {
  // camera constants, computed on the host exactly as RunThread does (RayTracer.cpp:95-101, 15-22)
  float X[3], Y[3], Z[3], pos[3];
  float l, wh, ww, camW, camH, dof;
  int32_t W, H;
  int32_t jitter;
  uint32_t seed_lo, seed_hi;
  int32_t sample_base;   // camera-sample index of the first sample of this pass
  int32_t n_samples;     // samples per pixel in this pass
  uint32_t n_pix;        // pixels in this pass (length of `pixels`)
  int32_t max_bounces, mc_bounces, mc_samples;
  int32_t shadow_min, shadow_max;
  float light_size_scale;
  int32_t photons;
  int32_t srgb;
  int32_t aov_sample;    // write hit AOVs for this camera-sample index (-1 none)
  int32_t spp_min, spp_max, adaptive;
  float adaptive_threshold;
  int32_t discard;       // sizing pass: trace and shade, but leave the accumulator alone
  uint32_t extensions;   // rt_render_params.extensions (RT_EXT_*)
};

struct FrameBufs {
  const uint32_t* pixels;  // pixel indices of this pass, ordered in 8x8 tiles
  float* path_rad;         // n_pix * n_samples path radiance slots, one plane per colour channel (rad_stride floats apart): the
  uint32_t rad_stride;     // kernels add with one float atomic per channel, and 64 lanes' slots then share 4 cache lines, not 16
  float* accum;            // W*H*8: sum rgb, n, sum of squares rgb, done flag
  int32_t* hit_node;       // W*H AOVs (may be null)
  int32_t* hit_face;
  float* aov_depth;        // W*H, W*H*3, W*H*3: geometry buffers of the AOV sample's primary hit (may be null)
  float* aov_normal;
  float* aov_albedo;
  uint32_t* overflow;
  DCounters* counters;
};

// Three appends to ONE segment of one queue with a single atomic (k_shade's refraction / reflection / indirect children):
// pos[i] = entry index for append i or RT_NONE.  Every lane of the wave must call.
#ifndef RT_EMU
RT_DEV void seg_push3(bool p0, bool p1, bool p2, uint32_t* counts, uint32_t seg, uint32_t segcap, uint32_t* overflow, uint32_t pos[3]) {
  const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1), m2 = __ballot(p2);
  pos[0] = pos[1] = pos[2] = RT_NONE;
  if ((m0 | m1 | m2) == 0) return;
  const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
  uint32_t base = 0;
  if (rt_lane() == 0) base = atomicAdd(counts + seg * RT_CSTRIDE, n0 + n1 + n2);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
  const uint32_t r1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
  const uint32_t r2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
  const uint32_t q0 = base + r0, q1 = base + n0 + r1, q2 = base + n0 + n1 + r2;
  if (base + n0 + n1 + n2 > segcap && rt_lane() == 0) *overflow = 1u;
  if (p0 && q0 < segcap) pos[0] = seg * segcap + q0;
  if (p1 && q1 < segcap) pos[1] = seg * segcap + q1;
  if (p2 && q2 < segcap) pos[2] = seg * segcap + q2;
}
#else
RT_DEV void seg_push3(bool p0, bool p1, bool p2, uint32_t* counts, uint32_t seg, uint32_t segcap, uint32_t* overflow, uint32_t pos[3]) {
  pos[0] = seg_push(p0, counts, seg, segcap, overflow);
  pos[1] = seg_push(p1, counts, seg, segcap, overflow);
  pos[2] = seg_push(p2, counts, seg, segcap, overflow);
}
#endif

// Counts the lanes with `pred` per run of equal `key` (runs are contiguous lanes: consecutive work items of one
// shadow group) and adds the count to the run's word with ONE atomic per run instead of one per lane.  Every lane of
// the wave must call it; lanes without an item pass pred = false and key = RT_NONE.  (One atomic per lit shadow ray
// was what bound k_shadow_soft: 5 G atomics per C3 bench step.)
#ifndef RT_EMU
RT_DEV void run_count_add(bool pred, uint32_t key, uint32_t* word) {
  const uint32_t lane = rt_lane();
  uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
  unsigned long long heads = __ballot(lane == 0 || prev != key);
  unsigned long long lit = __ballot(pred);
  if (!((heads >> lane) & 1ull) || key == RT_NONE) return;
  unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
  uint32_t end = above ? lane + 1 + (uint32_t)(__ffsll((long long)above) - 1) : 64u;
  unsigned long long mask = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
  uint32_t n = (uint32_t)__popcll(lit & mask);
  if (n) atomicAdd(word, n);
}
#else
RT_DEV void run_count_add(bool pred, uint32_t key, uint32_t* word) {
  if (pred && key != RT_NONE) *word += 1u;
}
#endif

// Which segment(s) this wave walks, and with which stride.  With G >= nseg waves, wave g takes segment g % nseg
// together with the G / nseg - 1 other waves of that residue; with fewer waves each takes several segments.
struct SegWalk {
  uint32_t s, s_step, r, R;
};
RT_DEV void seg_walk_init(SegWalk& w, uint32_t nseg) {
  uint32_t G = rt_num_waves(), g = rt_wave_id();
  if (G >= nseg) {
    w.R = G / nseg;
    w.s = g % nseg;
    w.s_step = nseg;
    w.r = g / nseg;
    if (w.r >= w.R) w.s = nseg;
  } else {
    w.R = 1;
    w.s = g;
    w.s_step = G;
    w.r = 0;
  }
}
// Loop over `mult` work items per queue entry; inside: seg_e (item number within the segment), seg_valid,
// seg_base_ (index of the segment's first entry), sw_.s (segment).  All lanes of a wave run the same iterations.
#define RT_SEG_BEGIN(counts_, nseg_, segcap_, mult_)                                                        \
  {                                                                                                         \
    SegWalk sw_;                                                                                            \
    seg_walk_init(sw_, (nseg_));                                                                            \
    for (; sw_.s < (nseg_); sw_.s += sw_.s_step) {                                                          \
      uint32_t seg_n_ = (counts_)[sw_.s * RT_CSTRIDE];                                                      \
      if (seg_n_ > (segcap_)) seg_n_ = (segcap_);                                                           \
      const unsigned long long seg_items_ = (unsigned long long)seg_n_ * (unsigned long long)(mult_);       \
      const uint32_t seg_base_ = sw_.s * (segcap_);                                                         \
      for (unsigned long long e0_ = (unsigned long long)sw_.r * RT_WAVE; e0_ < seg_items_;                  \
           e0_ += (unsigned long long)sw_.R * RT_WAVE) {                                                    \
        const unsigned long long seg_e = e0_ + rt_lane();                                                   \
        const bool seg_valid = seg_e < seg_items_;
#define RT_SEG_END \
  }                \
  }                \
  }

RT_DEV float two_pi_times(float x) { return (float)(2.0 * RT_PI_D * (double)x); }  // `2.0f * M_PI * x` promotes to double

#ifndef RT_EMU
// per-lane traversal stack in dynamic LDS, [entry][lane]: the launch sizes it to the committed scene's deepest BVH (+ the
// instance BVH), so shallow scenes leave LDS for more resident blocks (kernels that never push may be launched with none)
#define RT_STACK_DECL extern __shared__ uint32_t s_stack[]; uint32_t* stk = s_stack + RT_LANE_IN_BLOCK; const uint32_t stride = RT_BLOCK;
#else
#define RT_STACK_DECL uint32_t s_stack[RT_STACK]; uint32_t* stk = s_stack; const uint32_t stride = 1;
#endif

// ------------------------------------------------------------------------------------------------ raygen
// Camera ray of path slot `idx` (RayTracer.cpp:138-174).
RT_DEV void primary_ray(const DScene& S, const PassP& P, const FrameBufs& F, uint32_t idx, V3& origin, V3& dir, uint32_t& pixel, int& i) {
  uint32_t j = idx % P.n_pix, k = idx / P.n_pix;
  pixel = F.pixels[j];
  int x = (int)(pixel % (uint32_t)P.W), y = (int)(pixel / (uint32_t)P.W);
  i = P.sample_base + (int)k;
  float hx = 0, hy = 0;
  V3 camOff = v3(0, 0, 0);
  if (P.jitter) {
    uint32_t r[4];
    philox4x32_10(pixel, 0, 0, RT_TAG_PIXEL, P.seed_lo, P.seed_hi, r);
    float ro = u32_to_unit(r[0]), roy = u32_to_unit(r[1]);
    hx = S.halton[RT_H2 + (i & 127)] + ro;   // RayTracer.cpp:138-144 (Y also uses randomOffset)
    if (hx > 1.0f) hx -= 1.0f;
    hy = S.halton[RT_H3 + (i & 127)] + ro;
    if (hy > 1.0f) hy -= 1.0f;
    if (P.dof != 0.0f) {   // (a pinhole camera: the lens offset is +-0 and changes neither origin nor direction)
      float discX = S.halton[RT_H5 + (i & 127)] + ro;   // :152-157
      float discY = S.halton[RT_H7 + (i & 127)] + roy;
      if (discX > 1.0f) discX -= 1.0f;
      if (discY > 1.0f) discY -= 1.0f;
      float r_ = sqrtf(discX);
      float angle = two_pi_times(discY);
      float lensU = r_ * P.dof * cosf(angle);
      float lensV = r_ * P.dof * sinf(angle);
      camOff = v3(lensU, lensV, 0.0f);
    }
  }
  float pixX = -(P.ww / 2.0f) + ((P.ww * ((float)x + (1.0f / 2.0f) + hx) / P.camW));   // :147
  float pixY = (P.wh / 2.0f) - ((P.wh * ((float)y + (1.0f / 2.0f) + hy) / P.camH));    // :148
  V3 pixelPos = v3(pixX, pixY, -P.l);
  V3 X = v3(P.X[0], P.X[1], P.X[2]), Y = v3(P.Y[0], P.Y[1], P.Y[2]), Z = v3(P.Z[0], P.Z[1], P.Z[2]);
  V3 worldCamera = X * camOff.x + Y * camOff.y + Z * camOff.z;      // cam2Wrld * (v, 0), :165-166
  V3 worldPixel = X * pixelPos.x + Y * pixelPos.y + Z * pixelPos.z;
  origin = v3(P.pos[0], P.pos[1], P.pos[2]) + worldCamera;
  dir = worldPixel - worldCamera;
}

// ------------------------------------------------------------------------------------------------ trace
// Closest hit in two phases.
//   k_trace_closest  every lane takes one ray through the instance table (wave-uniform loop, scalar-loaded entries):
//                    spheres and planes are intersected, mesh instances only culled against their world box.  A ray
//                    that reaches no mesh box is finished here (hit record written); the others become work items.
//   k_trace_mesh     walks the BVHs of the items' mesh instances.  Waves are persistent: a lane that finishes its
//                    ray claims the next item (one atomic per wave per refill), so the lanes of a wave stay busy
//                    although rays need very different numbers of steps.  (Measured before the split: 38 % of the
//                    lanes active in VALU instructions of the one-phase kernel, profiles/r01_v3_pmc_mem_summary.json.)
// The result is the one RayTracer::TraverseTree defines (minimum t, strict '<' in depth-first instance order): the
// second phase compares (t, instance index) against the first phase's best.
RT_DEV void write_raw(const RayQ& Q, uint32_t idx, const Best& B) {
  Q.r0[idx] = make_float4(B.t, B.mh.u, B.mh.v, __uint_as_float((uint32_t)(B.inst + 1) | (B.front ? 0x80000000u : 0u)));
  Q.rf[idx] = B.mh.face;
}
RT_DEV void unpack_raw(float4 r, uint32_t face, Best& B) {
  uint32_t w = __float_as_uint(r.w);
  B.t = r.x; B.mh.u = r.y; B.mh.v = r.z; B.mh.t = r.x; B.mh.key = 0;
  B.inst = (int)(w & 0x7fffffffu) - 1; B.front = (w >> 31) != 0;
  B.mh.face = face;
}
RT_DEV void read_raw(const RayQ& Q, uint32_t idx, Best& B) { unpack_raw(Q.r0[idx], Q.rf[idx], B); }

// Camera rays and their first phase in one kernel: the ray is generated, written to the level-0 queue (k_shade and
// k_trace_mesh read it) and taken through the instance table while still in registers (separate kernels re-read 48 bytes
// per path).  64-path tiles (8x8 pixels of one sample) are dealt round-robin to the queue segments; a wave handles one
// tile per iteration, so its appends go to that tile's segment.
template <bool CNT, bool TLAS>
RT_KERNEL(k_raygen_trace, DScene S, PassP P, FrameBufs F, RayQ Q, MeshQ MQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t traced = 0;
  const uint32_t total = P.n_pix * (uint32_t)P.n_samples;
  for (uint32_t idx = RT_TID; (idx & ~63u) < total; idx += RT_NTHREADS) {
    const bool valid = idx < total;
    const uint32_t tile = idx / 64u;
    const uint32_t seg = tile % Q.nseg;
    const uint32_t qi = seg * Q.segcap + (tile / Q.nseg) * 64u + (idx % 64u);
    uint32_t head = RT_NONE;
    Best B;
    best_init(B);
    V3 o = v3(0, 0, 0), dir = v3(1, 0, 0);
    if (valid) {
      uint32_t pixel;
      int i;
      primary_ray(S, P, F, idx, o, dir, pixel, i);
      Q.q0[qi] = make_float4(o.x, o.y, o.z, 1.0f);
      Q.q1[qi] = make_float4(dir.x, dir.y, dir.z, 1.0f);
      Q.q2[qi] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
      uint4 m;
      m.x = idx; m.y = pixel; m.z = (uint32_t)i; m.w = RT_META(1, 0, 0, 0);
      Q.q3[qi] = m;
      F.path_rad[idx] = 0.0f; F.path_rad[(size_t)F.rad_stride + idx] = 0.0f; F.path_rad[2 * (size_t)F.rad_stride + idx] = 0.0f;
      traced++;
    }
    // primary rays see front faces only (RayTracer.cpp:233)
    const CandSink K = cand_sink(MQ, seg % MQ.nseg, F.overflow);
    if (TLAS) { if (valid) closest_instances_tlas<CNT, true>(S, o, dir, 1, B, head, K, stk, stride, cnt); }
    else closest_instances<CNT, true, true>(S, o, dir, 1, B, head, K, stk, stride, cnt, valid);   // (every lane: wave-wide appends)
    if (valid && head == RT_NONE) write_raw(Q, qi, B);
    uint32_t wi = seg_push(valid && head != RT_NONE, MQ.counts, seg % MQ.nseg, MQ.segcap, F.overflow);
    if (wi != RT_NONE) {
      uint4 it;
      it.x = qi; it.y = __float_as_uint(B.t); it.z = (uint32_t)(B.inst + 1) | (B.front ? 0x80000000u : 0u); it.w = head;
      MQ.item4[wi] = it;
    }
  }
  uint32_t tiles = (total + 63u) / 64u;
  for (uint32_t sgm = RT_TID; sgm < Q.nseg; sgm += RT_NTHREADS) {
    uint32_t nt = tiles > sgm ? (tiles - sgm + Q.nseg - 1u) / Q.nseg : 0u;
    uint32_t c = nt * 64u;
    if (nt && (tiles - 1u) % Q.nseg == sgm) c -= tiles * 64u - total;  // the last tile may be partial
    Q.counts[sgm * RT_CSTRIDE] = c;
  }
  if (CNT) {
    cnt.flush(F.counters);
    cnt.flush_closest(F.counters);
    if (traced) rt_atomic_add_u64(&F.counters->rays_primary, traced);
  }
}

// TLAS = true: the instance BVH is walked per lane (scenes with many instances) instead of the wave-uniform table loop.
template <bool CNT, bool TLAS>
RT_KERNEL(k_trace_closest, DScene S, PassP P, FrameBufs F, RayQ Q, MeshQ MQ, int level) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t traced = 0;
  RT_SEG_BEGIN(Q.counts, Q.nseg, Q.segcap, 1u)
    uint32_t idx = seg_base_ + (uint32_t)seg_e;
    uint32_t head = RT_NONE;
    Best B;
    best_init(B);
    V3 o = v3(0, 0, 0), dir = v3(1, 0, 0);
    bool trace = false;
    if (seg_valid) {
      uint4 m = Q.q3[idx];
      if (!(m.w & (1u << 16))) {  // (re-shade entries carry a hit record written by k_shade, lights.cpp:172-183)
        float4 a = Q.q0[idx], b = Q.q1[idx];
        o = v3(a.x, a.y, a.z); dir = v3(b.x, b.y, b.z);
        trace = true;
        traced++;
      }
    }
    // primary rays see front faces only (RayTracer.cpp:233), secondary rays both (lights.cpp:185)
    const CandSink K = cand_sink(MQ, sw_.s % MQ.nseg, F.overflow);
    if (TLAS) { if (trace) closest_instances_tlas<CNT, true>(S, o, dir, level == 0 ? 1 : 3, B, head, K, stk, stride, cnt); }
    else closest_instances<CNT, true>(S, o, dir, level == 0 ? 1 : 3, B, head, K, stk, stride, cnt, trace);   // (every lane: wave-wide appends)
    if (trace && head == RT_NONE) write_raw(Q, idx, B);
    uint32_t wi = seg_push(trace && head != RT_NONE, MQ.counts, sw_.s % MQ.nseg, MQ.segcap, F.overflow);
    if (wi != RT_NONE) {
      uint4 it;
      it.x = idx; it.y = __float_as_uint(B.t); it.z = (uint32_t)(B.inst + 1) | (B.front ? 0x80000000u : 0u); it.w = head;
      MQ.item4[wi] = it;
    }
  RT_SEG_END
  if (CNT) {
    cnt.flush(F.counters);
    cnt.flush_closest(F.counters);
    if (traced) rt_atomic_add_u64(level == 0 ? &F.counters->rays_primary : &F.counters->rays_secondary, traced);
  }
}

// ---- item claiming for the persistent second-phase kernels
// Lanes whose `idle` is set ask for an item of segment `seg`; returns the item index or RT_NONE.  `exhausted` is
// set (wave-uniform) when the segment has no items left after this claim.
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 32u   // lanes that must wait for service (finish / claim / next instance) before a wave runs it
#endif
#ifndef RT_NODE_MIN
#define RT_NODE_MIN 8u      // see the inner-node loop of k_trace_mesh
#endif
RT_DEV uint32_t claim_items(bool idle, const MeshQ& MQ, uint32_t seg, bool& exhausted) {
  uint32_t n_seg = MQ.counts[seg * RT_CSTRIDE];
  if (n_seg > MQ.segcap) n_seg = MQ.segcap;
  exhausted = false;
#ifndef RT_EMU
  unsigned long long mask = __ballot(idle);
  uint32_t n = (uint32_t)__popcll(mask);
  if (n == 0) return RT_NONE;
  uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)rt_lane() == leader) base = atomicAdd(MQ.cursor + seg * RT_CSTRIDE, n);
  base = (uint32_t)__shfl((int)base, leader);
#else
  if (!idle) return RT_NONE;
  uint32_t n = 1, rank = 0;
  uint32_t base = MQ.cursor[seg * RT_CSTRIDE];
  MQ.cursor[seg * RT_CSTRIDE] = base + 1;
#endif
  exhausted = base + n >= n_seg;
  if (idle && base + rank < n_seg) return seg * MQ.segcap + base + rank;
  return RT_NONE;
}
RT_DEV uint32_t count_lanes(bool pred) {
#ifndef RT_EMU
  return (uint32_t)__popcll(__ballot(pred));
#else
  return pred ? 1u : 0u;
#endif
}

enum { RT_ST_IDLE = 0, RT_ST_NEXT = 1, RT_ST_WALK = 2, RT_ST_DONE = 3 };
#ifdef RT_WALK_STATS   // developer build: wave-level iteration / active-lane tallies of k_trace_mesh<true>
__device__ unsigned long long g_walk_stats[16];
#define WSTAT(i, v) do { if (CNT && rt_lane() == 0) ws[i] += (v); } while (0)
#else
#define WSTAT(i, v) do {} while (0)
#endif
// Lane states of the second-phase kernels: IDLE no ray; NEXT has a ray, wants its next mesh instance; WALK inside a BVH;
// DONE all instances walked, result not yet written.  Everything except WALK is "service" work (finish, claim, enter
// the next instance); it is batched: the wave only runs it when at least RT_REFILL_MIN lanes wait for it (or nobody
// walks), so that this code, too, is issued for many lanes at a time.

// Exit condition every wave of the persistent kernels reaches even if the step logic were wrong: far above any real
// walk (a wave of the 12.6 M-triangle stress scene takes < 2^21 steps), reported to the host through the overflow word.
#define RT_WALK_GUARD (1u << 25)
#define RT_WALK_ABORT 0xDEADu
#ifndef RT_NODE_BATCH
#define RT_NODE_BATCH 4
#endif
#ifndef RT_LEAF_MIN
#define RT_LEAF_MIN 12u     // parked leaves that trigger a leaf step while enough lanes still stand at inner nodes
                            // (24 before the leaf steps fetched one slot ahead; now 8 / 12 / 16 / 24 -> C2 20.32 / 20.24 / 20.28 / 20.45 ms,
                            //  C5 880 / - / 884 / 895 ms)
#endif
#ifndef RT_MESH_WAVES
#define RT_MESH_WAVES 5          // measured with the scene-sized stack: 4 -> 7.24 ms, 5 -> 6.58 ms, 6 -> 7.65 ms (spills) per C2 step
#endif
template <bool CNT>
RT_KERNEL_W(RT_MESH_WAVES, k_trace_mesh, DScene S, PassP P, FrameBufs F, RayQ Q, MeshQ MQ, int level) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  const uint32_t nseg = MQ.nseg;
  uint32_t seg = rt_wave_id() % nseg, segs_done = 0;  // wave-uniform: home segment first, then the others (work stealing)
  const uint32_t refill_min = RT_REFILL_MIN < RT_WAVE ? RT_REFILL_MIN : RT_WAVE;
  int state = RT_ST_IDLE;
  uint32_t ray_idx = 0;
  Best B;
  best_init(B);
  MeshWalk W;
  W.p = W.d = W.inv = v3(0, 0, 0);
  W.cur = RT_NOREF; W.pend = 0; W.node_base = W.tri_base = 0; W.sp = 0; W.next = RT_NONE; W.inst = -1; W.fast = true;
  W.hint = 0; W.onpath = false; W.minst = RT_NONE;
  W.root = RT_NOREF; W.skip = RT_NOREF; W.restart = false; W.anc_base = RT_NONE; W.anc_blocks = 0;
  MeshHit L;
  L.t = RT_BIG; L.u = L.v = 0; L.key = 0; L.face = 0;
  bool lfound = false;
  uint32_t n_items = 0, guard = 0;
#ifdef RT_WALK_STATS
  unsigned long long ws[16] = {0};
#endif
  // One step per iteration, chosen for the whole wave: service (finish rays, claim items, enter the next instance),
  // an inner-node step, or a leaf step.  A lane that reaches a leaf parks it (one per lane) and walks on from its stack,
  // so node steps keep the walking lanes busy and the parked leaves are intersected together.  Walking on before the
  // parked leaf is intersected only means pruning against a slightly older bound: the result (minimum t, tie rules by
  // key) does not depend on the order.  Service is batched: it runs when at least RT_REFILL_MIN lanes wait for it (or
  // nothing else is left to do) — checked every step, so lanes whose walk was short do not idle until the longest walk
  // of the wave ends (measured before this: 21 of 64 lanes active in a node step on the C2 bench).
#ifdef RT_WALK_STATS
  long long tprev = clock64(); int phase = 12;
#define WPHASE(ph) do { long long tn_ = clock64(); WSTAT(phase, (unsigned long long)(tn_ - tprev)); tprev = tn_; phase = (ph); } while (0)
#else
#define WPHASE(ph) do {} while (0)
#endif
  for (;;) {
    WPHASE(12);
    if (state == RT_ST_WALK && W.pend == 0 && ref_is_leaf(W.cur)) { W.pend = W.cur; W.cur = walk_pop(W, stk, stride); }
    if (state == RT_ST_WALK && W.cur == RT_NOREF && W.pend == 0) {  // this instance is walked
      // TraverseTree's strict '<' in depth-first order: an equal t wins only for an earlier instance
      if (lfound && (L.t < B.t || (L.t == B.t && W.inst < B.inst))) { B.t = L.t; B.inst = W.inst; B.mh = L; }
      state = RT_ST_NEXT;
    }
    const bool walking = state == RT_ST_WALK;
    const bool at_node = walking && ref_is_node(W.cur);
    const uint32_t n_walk = count_lanes(walking), n_node = count_lanes(at_node), n_pend = count_lanes(walking && W.pend != 0);
    // lanes that service can do something for (an idle lane only while unclaimed items may be left)
    const uint32_t n_serv = count_lanes(state == RT_ST_NEXT || (state == RT_ST_IDLE && segs_done < nseg));
    if (n_walk == 0 && n_serv == 0) break;
    if (++guard > RT_WALK_GUARD) { *F.overflow = RT_WALK_ABORT; break; }
    if (n_serv >= refill_min || n_walk == 0) {
      // ---- service: a lane whose instance is walked takes its ray's next candidate or, after the last one, writes the
      // ray's result and becomes idle; idle lanes claim new items and enter their first candidate.  One pass per candidate.
      WSTAT(2, 1); WSTAT(3, n_serv); WPHASE(13);
      uint32_t enter = RT_NONE;
      if (state == RT_ST_NEXT) {
        enter = W.next;
        if (enter == RT_NONE) {
          write_raw(Q, ray_idx, B);
          state = RT_ST_IDLE;
        }
      }
      if (segs_done < nseg) {
        bool exhausted;
        uint32_t wi = claim_items(state == RT_ST_IDLE, MQ, seg, exhausted);
        if (wi != RT_NONE) {
          uint4 it = MQ.item4[wi];
          ray_idx = it.x;
          best_init(B);
          B.t = __uint_as_float(it.y); B.inst = (int)(it.z & 0x7fffffffu) - 1; B.front = (it.z >> 31) != 0;
          enter = it.w;   // (an item always has at least one candidate)
          n_items++;
        }
        if (exhausted) { seg = (seg + 1) % nseg; segs_done++; }
      }
      if (enter != RT_NONE) {
        mesh_walk_enter<CNT>(S, MQ.cand, enter, W, cnt);
        L.t = RT_BIG; lfound = false;
        state = RT_ST_WALK;
      }
      continue;
    }
    if (n_node != 0 && (n_pend < RT_LEAF_MIN || n_node >= RT_WAVE / 2) && (n_node >= RT_NODE_MIN || n_pend < RT_NODE_MIN)) {
      WSTAT(6, 1); WSTAT(7, n_node); WPHASE(14);
      // RT_NODE_BATCH steps per decision: the wave-wide bookkeeping above costs about as much as one node step
#pragma unroll
      for (int k = 0; k < RT_NODE_BATCH; k++) {
        if (k > 0 && walking && W.pend == 0 && ref_is_leaf(W.cur)) { W.pend = W.cur; W.cur = walk_pop(W, stk, stride); }
        if (walking && ref_is_node(W.cur)) {  // (fetching inside the branch measured 10 % faster here)
          float4 n0, n1, n2, n3;
          fetch_node(S, W, true, n0, n1, n2, n3);
          mesh_walk_node<CNT, true>(W, n0, n1, n2, n3, L.t < B.t ? L.t : B.t, stk, stride, cnt);
        }
      }
    } else {
      WSTAT(8, 1); WSTAT(9, n_pend); WPHASE(15);
      if (walking && W.pend != 0) {
        mesh_walk_leaf_closest<CNT>(S, W, W.pend, L, lfound, cnt);
        W.pend = 0;
      }
    }
  }
  if (CNT) {
    cnt.flush(F.counters);
    cnt.flush_closest(F.counters);
    if (cnt.c.inst) rt_atomic_add_u64(&F.counters->mesh_inst_tests, cnt.c.inst);
    if (n_items) rt_atomic_add_u64(&F.counters->mesh_items, n_items);
#ifdef RT_WALK_STATS
    if (rt_lane() == 0) for (int i = 0; i < 16; i++) if (ws[i]) atomicAdd(&g_walk_stats[i], ws[i]);
#endif
  }
}

// ------------------------------------------------------------------------------------------------ shade
struct ShadePt {
  V3 P, N, V, uvw;
  bool front;
  uint32_t pixel, nid;
  int sample;
};

// fmodf(u, 1.0f) for 0 <= u < 2 (a table entry in [0, 1) plus an offset in [0, 1)), bit for bit: u - 1 is exact there
// (the library's fmodf is a reduction loop of about a hundred instructions)
RT_DEV float wrap01(float u) { return u >= 1.0f ? u - 1.0f : u; }

// glossy half vector, materials.cpp:35-49 / 79-94
// ggx_a2 >= 0 (RT_MTL_MICROFACET under RT_EXT_MATERIALS, this build's own model): the polar angle follows the GGX distribution
// of squared roughness ggx_a2 instead of the Blinn power lobe.
RT_DEV V3 gloss_half(const DScene& S, const ShadePt& sp, float gloss, float oPhi, float oTheta, bool renorm, float ggx_a2 = -1.0f) {
  int gi = sp.sample < S.n_gloss ? sp.sample : S.n_gloss - 1;
  float phi = two_pi_times(wrap01(S.halton[RT_HGLOSS + gi] + oPhi));
  float uTheta = wrap01(S.halton[RT_HGLOSS + S.n_gloss + gi] + oTheta);
  float cosTheta = ggx_a2 >= 0.0f ? sqrtf((1.0f - uTheta) / (1.0f + (ggx_a2 - 1.0f) * uTheta)) : powf(uTheta, 1.0f / (gloss + 1.0f));
  float sinTheta = sqrtf(1.0f - (cosTheta * cosTheta));
  float sn, cs;
  sincosf(phi, &sn, &cs);   // (one argument reduction for both)
  float x = sinTheta * cs;
  float y = sinTheta * sn;
  float z = cosTheta;
  V3 t, b;
  orthonormals(sp.N, t, b);
  V3 H = (x * t) + (y * b) + (z * sp.N);
  if (renorm) H = normalized(H);
  return H;
}
RT_DEV V3 reflect_dir(const DScene& S, const ShadePt& sp, float gloss, float oPhi, float oTheta, float ggx_a2 = -1.0f) {  // materials.cpp:30-59
  V3 H = gloss_half(S, sp, gloss, oPhi, oTheta, false, ggx_a2);
  float d = dot(H, sp.V);
  V3 dir = (2 * d) * H - sp.V;
  return normalized(dir);
}
// RefractRay, materials.cpp:70-133.  r0 = block-0 draws, r1 = block-1 draws (used on total internal reflection).
RT_DEV void refract_ray(const DScene& S, const ShadePt& sp, float ior, float gloss, const float* r0, const float* r1, V3& o, V3& d,
                        float ggx_a2 = -1.0f) {
  const float eps = 1e-4f;
  V3 view = sp.V;
  V3 H = gloss_half(S, sp, gloss, r0[0], r0[1], true, ggx_a2);
  if (sp.front) {
    float eta = 1 / ior;
    float etaSq = eta * eta;
    float NdotV = dot(view, H);
    float cosThetaT = sqrtf(1 - etaSq * (1 - NdotV * NdotV));
    V3 dir = -eta * view - (cosThetaT - eta * NdotV) * H;
    d = normalized(dir);
    float sign = dot(d, sp.N) > 0.0f ? 1.0f : -1.0f;
    o = sp.P + sp.N * (eps * sign);
  } else {
    float eta = ior;
    V3 Hb = -H;
    V3 negN = -sp.N;
    float NdotV = dot(view, Hb);
    float etaSq = eta * eta;
    float cosThetaSquared = (float)(1.0 - (double)(etaSq * (1 - NdotV * NdotV)));
    if (cosThetaSquared < 0.0f) {
      o = sp.P;
      d = reflect_dir(S, sp, gloss, r1[0], r1[1], ggx_a2);
      return;
    }
    float cosThetaT = sqrtf(cosThetaSquared);
    V3 dir = -eta * view - (cosThetaT - eta * NdotV) * Hb;
    d = normalized(dir);
    float sign = dot(d, negN) > 0.0f ? 1.0f : -1.0f;
    o = sp.P + negN * (eps * sign);
  }
}

RT_DEV void shade_draws(const PassP& P, const ShadePt& sp, uint32_t block, float out[4]) {
  uint32_t r[4];
  philox4x32_10(sp.pixel, sp.nid, block, RT_TAG_SHADE, P.seed_lo, P.seed_hi, r);
  out[0] = u32_to_unit(r[0]); out[1] = u32_to_unit(r[1]); out[2] = u32_to_unit(r[2]); out[3] = u32_to_unit(r[3]);
}

RT_DEV void add_radiance(const FrameBufs& F, uint32_t slot, V3 c) {
#ifdef RT_EXP_NO_RADIANCE   // timing experiment only: what the radiance atomics cost
  if (slot != 0xFFFFFFF0u) return;
#endif
  float* p = F.path_rad + slot;
  if (c.x != 0.0f) rt_atomic_add_f32(p, c.x);
  if (c.y != 0.0f) rt_atomic_add_f32(p + F.rad_stride, c.y);
  if (c.z != 0.0f) rt_atomic_add_f32(p + 2 * (size_t)F.rad_stride, c.z);
}

RT_DEV V3 absorb(V3 c, V3 ab, float dist) {  // materials.cpp:255-259
  if (dist > 0.0f && (ab.x > 0.0f || ab.y > 0.0f || ab.z > 0.0f)) {
    c.x *= expf(-ab.x * dist);
    c.y *= expf(-ab.y * dist);
    c.z *= expf(-ab.z * dist);
  }
  return c;
}

struct ChildRay {
  V3 o, d, w;
  uint32_t meta;
};
// this build's microfacet model (RT_EXT_MATERIALS; oracle: ggx_g1, schlick): separable Smith masking and Schlick's Fresnel
RT_DEV float ggx_g1(float x, float a2) { return (2.0f * x) / (x + sqrtf(a2 + (1.0f - a2) * (x * x))); }
RT_DEV V3 schlick_f(V3 F0, float c) {
  float m = 1.0f - c;
  float m2 = m * m;
  float m5 = (m2 * m2) * m;
  return F0 + (v3(1, 1, 1) - F0) * m5;
}
RT_DEV void write_child(const RayQ& QN, uint32_t ci, const ChildRay& c, V3 mabs, uint32_t slot, uint32_t pixel, uint32_t sample) {
  QN.q0[ci] = make_float4(c.o.x, c.o.y, c.o.z, c.w.x);
  QN.q1[ci] = make_float4(c.d.x, c.d.y, c.d.z, c.w.y);
  QN.q2[ci] = make_float4(mabs.x, mabs.y, mabs.z, c.w.z);
  uint4 cm;
  cm.x = slot; cm.y = pixel; cm.z = sample; cm.w = c.meta;
  QN.q3[ci] = cm;
}

// Every lane of a wave runs every queue-append site of an iteration (predicated), so the per-wave chunk state of
// wave_push stays uniform; see WavePush.
// MC / PH: the indirect-diffuse sampler / the photon term can be on in this pass (compiled out otherwise: the kernels are
// held back by their registers, and a frame uses one combination throughout).
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4   // measured on the C2 bench: 2 -> 7.37 ms, 3 -> 6.52 ms, 4 -> 6.26 ms per step
#endif
// One shade point: SendRay's dispatch on the closest hit `B` of the ray (q0, q1, q2, ids m) — miss / light / surface —, the hit's
// attributes, MtlBlinn::Shade's unshadowed terms, and the appends of its children (-> QN), shadow groups (-> GH / GS) and photon
// lookup (-> GQ), all into segment `oseg`.  Called by k_shade for the rays that went through a mesh BVH, and by the first-phase
// trace kernels themselves (k_raygen_shade, k_trace_shade) for the rays they finish: those never travel through a queue
// between being traced and being shaded.  A re-shade entry (lights.cpp:172-183) brings its hit record in Q.h0..h2[idx]
// instead of B.  EVERY lane of the wave must call (`valid`: this lane has a shade point).
template <bool CNT, bool MC, bool PH, bool XM>
RT_DEV void shade_point(const DScene& S, const PassP& P, const FrameBufs& F, const RayQ& Q, uint32_t idx, const RayQ& QN, const GroupQ& GH,
                        const GroupQ& GS, const GatherQ& GQ, uint32_t oseg, bool valid, uint4 m, float4 q0, float4 q1, float4 q2, const Best& B,
                        Cnt<CNT>& cnt, uint32_t& n_shades) {
  RT_UTYPE(DLight)* lights = RT_UNIFORM_PTR(DLight, S.lights);
  uint32_t slot = m.x, pixel = m.y;
  uint32_t nid = m.w & 0xFFFu, bounce = (m.w >> 12) & 7u;
  bool secondary = (m.w >> 15) & 1u, reshade = (m.w >> 16) & 1u;

  bool surface = false;
  const bool aov = bounce == 0 && F.aov_depth && (int)m.z == P.aov_sample && valid;
  V3 aov_n = v3(0, 0, 0), aov_a = v3(1, 1, 1);
  float aov_z = RT_BIG;
  ShadePt sp;
  sp.P = sp.N = sp.V = sp.uvw = v3(0, 0, 0);
  sp.front = true; sp.pixel = pixel; sp.nid = nid; sp.sample = (int)m.z;
  V3 w = v3(0, 0, 0), mabs = v3(0, 0, 0), cKd = v3(0, 0, 0), cKs = v3(0, 0, 0), kd = v3(0, 0, 0);
  float alpha = 0.0f, zs = 0.0f;
  int smode = 0;   // 0 MtlBlinn, 1 Phong lobe, 2 microfacet (cKs = F0, alpha = squared roughness): XM variants only
  V3 du0 = v3(0, 0, 0), du1 = v3(0, 0, 0);   // HitInfo::duvw of a camera ray's hit (RT_EXT_TEX_FILTER): XM variants only
  bool filt = false;
  float4 h1 = make_float4(0, 0, 0, 0), h2 = make_float4(0, 0, 0, 0);
  ChildRay c_refr, c_refl, c_mc;
  bool p_refr = false, p_refl = false, p_mc = false, p_gather = false, refl_below = false;
  c_refr.o = c_refr.d = c_refr.w = c_refl.o = c_refl.d = c_refl.w = c_mc.o = c_mc.d = c_mc.w = v3(0, 0, 0);
  c_refr.meta = c_refl.meta = c_mc.meta = 0;
  V3 wk = v3(0, 0, 0);

  if (valid) {
    float4 h0;
    if (reshade) {  // the hit record k_shade wrote for this entry (lights.cpp:172-183)
      h0 = Q.h0[idx];
      h1 = Q.h1[idx];
      h2 = Q.h2[idx];
    } else {        // hit attributes of the closest hit the trace kernels found (+ the renderable lights)
      Hit h;
      finalize_hit_from(S, S.sinst, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), B, h);
      if (XM && (P.extensions & RT_EXT_TEX_FILTER) && bounce == 0 && h.inst >= 0 && h.light < 0 && S.sinst[h.inst].uses_uvw) {
        // camera rays carry differentials (HitInfo::duvw, scene.h:82): the rays through the pixels to the right and below
        const V3 rd = v3(q1.x, q1.y, q1.z);
        const V3 dxr = rd + v3(P.X[0], P.X[1], P.X[2]) * (P.ww / P.camW), dyr = rd - v3(P.Y[0], P.Y[1], P.Y[2]) * (P.wh / P.camH);
        hit_differentials(S, S.sinst[h.inst], v3(q0.x, q0.y, q0.z), rd, dxr, dyr, B, h.uvw, du0, du1);
        filt = true;
      }
      h0 = make_float4(h.p.x, h.p.y, h.p.z, h.z);
      h1 = make_float4(h.N.x, h.N.y, h.N.z, __uint_as_float((uint32_t)h.inst));
      h2 = make_float4(h.uvw.x, h.uvw.y, h.uvw.z, __uint_as_float((h.front ? 1u : 0u) | ((uint32_t)(h.light + 1) << 8)));
      if (bounce == 0 && F.hit_node && (int)m.z == P.aov_sample) {
        F.hit_node[pixel] = h.light >= 0 ? -2 - h.light : (h.inst >= 0 ? S.sinst[h.inst].node : -1);
        F.hit_face[pixel] = h.light >= 0 ? -1 : h.face;
      }
    }
    w = v3(q0.w, q1.w, q2.w);
    V3 pabs = v3(q2.x, q2.y, q2.z);
    V3 rdir = v3(q1.x, q1.y, q1.z);
    int inst = (int)__float_as_uint(h1.w);
    uint32_t flags = __float_as_uint(h2.w);
    int light = (int)((flags >> 8) & 0xFFFFu) - 1;
    bool front = flags & 1u;
    float z = h0.w;
    if (inst < 0 && light < 0) {  // miss
      if (bounce == 0) {          // RayTracer.cpp:253-258
        float u = (float)(pixel % (uint32_t)P.W) / (float)P.W;
        float v = (float)(pixel / (uint32_t)P.W) / (float)P.H;
        V3 bg = eval_tex_color<CNT>(S, S.background, v3(u, v, 0.0f), cnt);
        aov_a = bg;
        add_radiance(F, slot, w * bg);
      } else {                    // lights.cpp:204-208, dist = BIGFLOAT
        V3 c = eval_environment<CNT>(S, S.environment, rdir, cnt);
        add_radiance(F, slot, w * absorb(c, pabs, RT_BIG));
      }
    } else if (light >= 0) {
      aov_z = z * length(rdir);
      if (bounce == 0) {          // RayTracer.cpp:243-251 / lights.h:60
        const DLight& L = S.lights[light];
        V3 c = v3(L.intensity[0], L.intensity[1], L.intensity[2]) / (RT_PI_F * L.size * L.size);
        add_radiance(F, slot, w * c);
      } else {                    // lights.cpp:199-202
        add_radiance(F, slot, w * absorb(v3(1, 1, 1), pabs, z));
      }
    } else {
      // SamplerInfo::SetHit, renderer.h:142-150
      zs = z * length(rdir);
      sp.N = normalized(v3(h1.x, h1.y, h1.z));
      V3 nd = normalized(rdir);
      sp.V = -nd;
      sp.P = v3(h0.x, h0.y, h0.z);
      sp.uvw = v3(h2.x, h2.y, h2.z);
      sp.front = front;
      if (bounce > 0) {  // lights.cpp:194 and materials.cpp:255-259 / 277-281, applied to this subtree's weight
        float dist = reshade ? 0.0f : (front ? zs : 0.0f);
        w = absorb(w, pabs, dist);
      }
      n_shades++;
      aov_z = zs; aov_n = sp.N;
      const DMaterial& M = S.sinst[inst].mtl;
      // this build's own MtlPhong / MtlMicrofacet (RT_EXT_MATERIALS; compiled into the XM variants only)
      const bool xm_on = XM && (P.extensions & RT_EXT_MATERIALS) != 0;
      const bool micro = xm_on && M.kind == RT_MTL_MICROFACET;
      if (xm_on && M.kind == RT_MTL_PHONG) smode = 1;
      if (micro) smode = 2;
      if (M.kind != 1 && smode == 0) {  // MtlPhong / MtlMicrofacet::Shade return white (materials.cpp:396-408)
        add_radiance(F, slot, w);
      } else {
        surface = true;
        // materials.cpp:231-236
        // (one copy of the texture code in a loop that is not unrolled, results routed by selects, was measured: the
        // kernel shrinks by a third and runs 5 % SLOWER on C2 — the six inlined copies stay)
        V3 reflection, refraction, ks, emission;
        if (XM && filt) {   // ShadeInfo::Eval -> TexturedValue::Eval(uvw, duvw) (renderer.h:135-136) with the derivatives filled
          reflection = eval_tex_color_filtered<CNT>(S, M.reflection, sp.uvw, du0, du1, cnt);
          refraction = eval_tex_color_filtered<CNT>(S, M.refraction, sp.uvw, du0, du1, cnt);
          kd = eval_tex_color_filtered<CNT>(S, M.diffuse, sp.uvw, du0, du1, cnt);
          ks = eval_tex_color_filtered<CNT>(S, M.specular, sp.uvw, du0, du1, cnt);
          alpha = eval_tex_float_filtered<CNT>(S, M.glossiness, sp.uvw, du0, du1, cnt);
          emission = eval_tex_color_filtered<CNT>(S, M.emission, sp.uvw, du0, du1, cnt);
        } else {
        reflection = eval_tex_color<CNT>(S, M.reflection, sp.uvw, cnt);
        refraction = eval_tex_color<CNT>(S, M.refraction, sp.uvw, cnt);
        kd = eval_tex_color<CNT>(S, M.diffuse, sp.uvw, cnt);
        ks = eval_tex_color<CNT>(S, M.specular, sp.uvw, cnt);
        alpha = eval_tex_float<CNT>(S, M.glossiness, sp.uvw, cnt);
        emission = eval_tex_color<CNT>(S, M.emission, sp.uvw, cnt);
        }
        aov_a = kd;
        float diffScalar = (float)(1 / RT_PI_D);
        float specScalar = (float)(((double)(alpha + 2)) / (8 * RT_PI_D));
        if (XM && smode == 1) specScalar = (float)(((double)(alpha + 2)) / (2 * RT_PI_D));   // the Phong lobe's normalisation
        mabs = v3(M.absorption[0], M.absorption[1], M.absorption[2]);
        bool can_bounce = (int)bounce < P.max_bounces;
        float r0[4] = {0, 0, 0, 0}, r1[4] = {0, 0, 0, 0};
        bool have_r0 = false, have_r1 = false;
        float mior = M.ior, mgloss = M.glossiness.v;
        if (XM && micro) {
          // GGX / Smith / Schlick, metallic workflow (oracle: shade_microfacet): kd = base (1 - metal) (1 - transmittance),
          // F0 = lerp(dielectric F0 of the ior, base, metal), transmission kt = transmittance (1 - metal).  For the light loop
          // below, cKs carries F0 and alpha the squared roughness.
          float metal = fminf(fmaxf(ks.x, 0.0f), 1.0f);
          float a = alpha * alpha;
          if (a < 1e-3f) a = 1e-3f;
          const float a2 = a * a;
          const float rr = (1.0f - mior) / (1.0f + mior);
          const float f0d = rr * rr;
          const V3 F0 = v3(f0d, f0d, f0d) * (1.0f - metal) + kd * metal;
          const V3 kt = refraction * (1.0f - metal);
          kd = (kd * (1.0f - metal)) * (v3(1, 1, 1) - refraction);
          cKs = F0;
          alpha = a2;
          const float NdotV = dot(sp.N, sp.V);
          if (mior > 0.0f && can_bounce && nonzero(kt)) {   // transmission through a GGX-sampled facet, Fresnel at the macro normal
            V3 wc = w * (kt * (v3(1, 1, 1) - schlick_f(F0, fabsf(NdotV))));
            if (nonzero(wc)) {
              shade_draws(P, sp, 0, r0); have_r0 = true;
              shade_draws(P, sp, 1, r1); have_r1 = true;
              refract_ray(S, sp, mior, 0.0f, r0, r1, c_refr.o, c_refr.d, a2);
              c_refr.w = wc;
              c_refr.meta = RT_META(nid * 4 + 1, bounce + 1, 1, 0);
              p_refr = true;
            }
          }
          if (can_bounce && NdotV > 0.0f) {   // reflection about a GGX-sampled facet: F G (V.H) / ((N.V) (N.H))
            if (!have_r0) { shade_draws(P, sp, 0, r0); have_r0 = true; }
            const V3 H = gloss_half(S, sp, 0.0f, r0[2], r0[3], false, a2);
            const float VdotH = dot(sp.V, H);
            const V3 dir = normalized((2 * VdotH) * H - sp.V);
            const float NdotL = dot(sp.N, dir), NdotH = dot(sp.N, H);
            if (VdotH > 0.0f && NdotL > 0.0f && NdotH > 0.0f) {
              V3 wc = w * (schlick_f(F0, VdotH) * (((ggx_g1(NdotL, a2) * ggx_g1(NdotV, a2)) * VdotH) / (NdotV * NdotH)));
              if (nonzero(wc)) {
                c_refl.d = dir;
                c_refl.o = sp.P;
                c_refl.w = wc;
                c_refl.meta = RT_META(nid * 4 + 2, bounce + 1, 1, 0);
                p_refl = true;
              }
            }
          }
          cKd = kd * diffScalar;
        } else {
        cKd = kd * diffScalar;
        cKs = ks * specScalar;
        V3 fullReflection = reflection;
        if (mior > 0.0f && can_bounce && nonzero(refraction)) {  // materials.cpp:246-269
          float iorRatio = (1.0f - mior) / (1.0f + mior);
          V3 fresnel = refraction * (iorRatio * iorRatio);
          fullReflection = fullReflection + fresnel;
          V3 wc = w * (refraction * (v3(1, 1, 1) - fullReflection));
          if (nonzero(wc)) {
            shade_draws(P, sp, 0, r0); have_r0 = true;
            shade_draws(P, sp, 1, r1); have_r1 = true;
            refract_ray(S, sp, mior, mgloss, r0, r1, c_refr.o, c_refr.d);
            c_refr.w = wc;
            c_refr.meta = RT_META(nid * 4 + 1, bounce + 1, 1, 0);
            p_refr = true;
          }
        }
        if (nonzero(fullReflection) && can_bounce) {  // materials.cpp:272-285; scaled by `reflection` only
          V3 wc = w * reflection;
          if (nonzero(wc)) {
            if (!have_r0) { shade_draws(P, sp, 0, r0); have_r0 = true; }
            c_refl.d = reflect_dir(S, sp, mgloss, r0[2], r0[3]);
            c_refl.o = sp.P;
            c_refl.w = wc;
            refl_below = dot(c_refl.d, sp.N) < 0;  // lights.cpp:174: re-shade the same point with the new view direction
            c_refl.meta = RT_META(nid * 4 + 2, bounce + 1, 1, refl_below ? 1 : 0);
            p_refl = true;
          }
        }
        }
        V3 local = emission;
        if (MC && P.mc_bounces > 0 && (int)bounce < P.mc_bounces) {  // materials.cpp:310-312 + 175-212 (mcSamples = 1)
          V3 wc = w * kd;
          if (nonzero(wc)) {
            if (!have_r1) { shade_draws(P, sp, 1, r1); have_r1 = true; }
            int hi = (sp.sample * P.mc_samples) % 1000;
            float u1 = wrap01(S.halton[RT_HPHI + hi] + r1[2]);
            float u2 = wrap01(S.halton[RT_HTHETA + hi] + r1[3]);
            float r = sqrtf(u1);
            float phi = two_pi_times(u2);
            float sn, cs;
            sincosf(phi, &sn, &cs);
            float x = r * cs, y = r * sn, zz = sqrtf(1.0f - u1);
            V3 t, b;
            orthonormals(sp.N, t, b);
            c_mc.d = normalized((x * t) + (y * b) + (zz * sp.N));
            c_mc.o = sp.P;
            c_mc.w = wc;
            c_mc.meta = RT_META(nid * 4 + 3, bounce + 1, 1, 0);
            p_mc = true;
          }
        } else if (PH && P.photons) {  // materials.cpp:315-322
          wk = w * ((float)(1.0f / RT_PI_D) * kd);
          p_gather = nonzero(wk);
        }
        add_radiance(F, slot, w * local);
      }
    }
  }
  if (aov) {
    F.aov_depth[pixel] = aov_z;
    F.aov_normal[3 * (size_t)pixel] = aov_n.x; F.aov_normal[3 * (size_t)pixel + 1] = aov_n.y; F.aov_normal[3 * (size_t)pixel + 2] = aov_n.z;
    F.aov_albedo[3 * (size_t)pixel] = aov_a.x; F.aov_albedo[3 * (size_t)pixel + 1] = aov_a.y; F.aov_albedo[3 * (size_t)pixel + 2] = aov_a.z;
  }
  // ---- queue appends (all lanes of the wave pass through here); the three kinds of children share one atomic
  {
    uint32_t cpos[3];
    seg_push3(p_refr, p_refl, MC && p_mc, QN.counts, oseg, QN.segcap, F.overflow, cpos);
    if (cpos[0] != RT_NONE) write_child(QN, cpos[0], c_refr, mabs, slot, pixel, m.z);
    if (cpos[1] != RT_NONE) {
      write_child(QN, cpos[1], c_refl, mabs, slot, pixel, m.z);
      if (refl_below) {
        QN.h0[cpos[1]] = make_float4(sp.P.x, sp.P.y, sp.P.z, zs);
        QN.h1[cpos[1]] = make_float4(sp.N.x, sp.N.y, sp.N.z, h1.w);
        QN.h2[cpos[1]] = h2;
      }
    }
    if (MC && cpos[2] != RT_NONE) write_child(QN, cpos[2], c_mc, v3(0, 0, 0), slot, pixel, m.z);  // `dist` is unused by the sampler: no absorption
  }
  // light loop, materials.cpp:288-307: unshadowed term now, visibility from the shadow kernels
  for (int li = 0; li < S.n_lights; li++) {
    RT_UTYPE(DLight)* L = lights + li;
    int kind = L->kind;
    if (kind == 1) continue;  // ambient is accumulated and never added  [uniform branch]
    bool soft = false, lit = false;
    V3 contrib = v3(0, 0, 0);
    float rx = 0, ry = 0;
    if (kind == 3) soft = (L->size * P.light_size_scale) != 0.0f;
    if (surface) {
      V3 lightDir, inten = v3(L->intensity[0], L->intensity[1], L->intensity[2]);
      if (kind == 2) {  // lights.h:45
        lightDir = -v3(L->direction[0], L->direction[1], L->direction[2]);
      } else {          // lights.cpp:62-71
        V3 dl = v3(L->position[0], L->position[1], L->position[2]) - sp.P;
        float dist = length(dl);
        lightDir = normalized(dl);
        if (L->attenuation != 0.0f) inten = inten / (dist * dist);
      }
      V3 h = normalized(lightDir + sp.V);
      float cosphi = fmaxf(dot(sp.N, h), 0.0f);
      float costheta = fmaxf(dot(lightDir, sp.N), 0.0f);
      if (XM && smode == 1) {  // Phong: the lobe is around R . V, R = the light direction mirrored about N
        V3 R = (2 * dot(sp.N, lightDir)) * sp.N - lightDir;
        cosphi = fmaxf(dot(R, sp.V), 0.0f);
      }
      if (XM && smode == 2) {  // microfacet: D G F / (4 N.V)  (the N.L of the denominator cancels the cosine)
        const float NdotL = dot(lightDir, sp.N), NdotV = dot(sp.N, sp.V), a2 = alpha;
        V3 term = cKd * costheta;
        if (NdotL > 0.0f && NdotV > 0.0f) {
          const float VdotH = fmaxf(dot(sp.V, h), 0.0f);
          const float dd = (cosphi * cosphi) * (a2 - 1.0f) + 1.0f;
          const float D = a2 / (RT_PI_F * (dd * dd));
          term = term + schlick_f(cKs, VdotH) * (((D * ggx_g1(NdotL, a2)) * ggx_g1(NdotV, a2)) / (4.0f * NdotV));
        }
        contrib = w * (inten * term);
      } else {
      // (0 * powf(..) is +0 for every finite power of a cosine in [0, 1]: a wave of purely diffuse points skips powf)
      const float spec = nonzero(cKs) ? powf(cosphi, alpha) : 0.0f;
      contrib = w * (inten * ((costheta * cKd) + (cKs * spec)));
      }
      lit = nonzero(contrib);
      if (lit && soft) {
        float rl[4];
        shade_draws(P, sp, 2 + (uint32_t)li, rl);
        rx = rl[0]; ry = rl[1];
      }
    }
    if (soft) {
      uint32_t gi = seg_push(lit, GS.counts, oseg, GS.segcap, F.overflow);
      if (gi != RT_NONE) {
        GS.g0[gi] = make_float4(sp.P.x, sp.P.y, sp.P.z, __uint_as_float((uint32_t)li));
        GS.g1[gi] = make_float4(contrib.x, contrib.y, contrib.z, __uint_as_float(slot));
        GS.g2[gi] = make_float4(rx, ry, __uint_as_float(0u), __uint_as_float(0xFFFFFFFFu));   // (.w: instances 32..63 that may occlude)
        V3 toSP = normalized(sp.P - v3(L->position[0], L->position[1], L->position[2]));  // lights.cpp:21-22, 31-32
        V3 tangent, bitangent;
        orthonormals(toSP, tangent, bitangent);
        GS.g3[gi] = make_float4(tangent.x, tangent.y, tangent.z, L->size * P.light_size_scale);
        GS.g4[gi] = make_float4(bitangent.x, bitangent.y, bitangent.z, __uint_as_float(0xFFFFFFFFu));   // (.w: instances 0..31 that may occlude, k_shadow_soft_clear)
      }
    } else {
      uint32_t gi = seg_push(lit, GH.counts, oseg, GH.segcap, F.overflow);
      if (gi != RT_NONE) {
        GH.g0[gi] = make_float4(sp.P.x, sp.P.y, sp.P.z, __uint_as_float((uint32_t)li));
        GH.g1[gi] = make_float4(contrib.x, contrib.y, contrib.z, __uint_as_float(slot));
      }
    }
  }
  if (PH && P.photons) {
    uint32_t gi = seg_push(p_gather, GQ.counts, oseg, GQ.segcap, F.overflow);
    if (gi != RT_NONE) {
      GQ.a[gi] = make_float4(sp.P.x, sp.P.y, sp.P.z, __uint_as_float(slot));
      GQ.b[gi] = make_float4(sp.N.x, sp.N.y, sp.N.z, __uint_as_float(secondary ? 1u : 0u));
      GQ.c[gi] = make_float4(wk.x, wk.y, wk.z, 0.0f);
    }
  }
}

template <bool CNT, bool MC, bool PH, bool XM>
RT_KERNEL_W(RT_SHADE_WAVES, k_shade, DScene S, PassP P, FrameBufs F, RayQ Q, RayQ QN, GroupQ GH, GroupQ GS, GatherQ GQ) {
  Cnt<CNT> cnt;
  uint32_t n_shades = 0;
  RT_SEG_BEGIN(Q.counts, Q.nseg, Q.segcap, 1u)
    const uint32_t oseg = sw_.s % QN.nseg;  // children and shadow work stay in the parent's segment
    uint32_t idx = seg_base_ + (uint32_t)seg_e;
    // the whole entry is requested at once (ids, ray, weights, raw hit): one memory round trip per iteration instead of
    // ids first, then whatever they select
    uint4 m;
    m.x = m.y = m.z = 0; m.w = 0;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, raw0 = q0;
    uint32_t raw_face = 0;
    if (seg_valid) {
      m = Q.q3[idx];
      q0 = Q.q0[idx]; q1 = Q.q1[idx]; q2 = Q.q2[idx];
      raw0 = Q.r0[idx]; raw_face = Q.rf[idx];
    }
    Best B;
    unpack_raw(raw0, raw_face, B);
    shade_point<CNT, MC, PH, XM>(S, P, F, Q, idx, QN, GH, GS, GQ, oseg, seg_valid, m, q0, q1, q2, B, cnt, n_shades);
  RT_SEG_END
  if (CNT) {
    cnt.flush(F.counters);
    if (n_shades) rt_atomic_add_u64(&F.counters->shades, n_shades);
  }
}

// ------------------------------------------------------------------------------------------------ trace + shade, fused
// The first trace phase and the shading of the rays it finishes, in one kernel.  A ray whose path reaches no mesh box has its
// closest hit after the instance table (spheres, planes, light spheres): it is shaded right there, from registers — it is never
// written to a ray queue, its raw hit never stored and never read back (84 B written + 84 B read per ray in the two-kernel
// form, for three quarters of C2's rays and all of C3's).  A ray with mesh candidates is appended to the compact queue QM
// (entry + work item at the same index, one atomic per wave), walked by k_trace_mesh and shaded by k_shade, which then
// sees nothing but mesh hits.  Children of both go to QN, in the parent's segment.
#ifndef RT_FUSED_WAVES
#define RT_FUSED_WAVES RT_SHADE_WAVES   // the fused kernels at 3 / 4 / 5 waves per SIMD: C3 k_trace_shade 20.9 / 19.3 / 21.4 ms, C2 k_raygen_shade 3.80 / 3.25 / 3.86 ms
#endif
template <bool CNT, bool TLAS, bool MC, bool PH, bool XM>
RT_KERNEL_W(RT_FUSED_WAVES, k_raygen_shade, DScene S, PassP P, FrameBufs F, uint32_t* l0_counts, RayQ QM, MeshQ MQ, RayQ QN, GroupQ GH, GroupQ GS,
            GatherQ GQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t traced = 0, n_shades = 0;
  const uint32_t total = P.n_pix * (uint32_t)P.n_samples;
  const uint32_t nseg = QN.nseg;
  for (uint32_t idx = RT_TID; (idx & ~63u) < total; idx += RT_NTHREADS) {
    const bool valid = idx < total;
    const uint32_t seg = (idx / 64u) % nseg;   // 64-path tiles (8x8 pixels of one sample) are dealt round-robin to the segments
    uint32_t head = RT_NONE;
    Best B;
    best_init(B);
    V3 o = v3(0, 0, 0), dir = v3(1, 0, 0);
    uint4 m;
    m.x = m.y = m.z = m.w = 0;
    if (valid) {
      uint32_t pixel;
      int i;
      primary_ray(S, P, F, idx, o, dir, pixel, i);
      m.x = idx; m.y = pixel; m.z = (uint32_t)i; m.w = RT_META(1, 0, 0, 0);
      F.path_rad[idx] = 0.0f; F.path_rad[(size_t)F.rad_stride + idx] = 0.0f; F.path_rad[2 * (size_t)F.rad_stride + idx] = 0.0f;
      traced++;
    }
    // primary rays see front faces only (RayTracer.cpp:233)
    const CandSink K = cand_sink(MQ, seg, F.overflow);
    if (TLAS) { if (valid) closest_instances_tlas<CNT, true>(S, o, dir, 1, B, head, K, stk, stride, cnt); }
    else closest_instances<CNT, true, true>(S, o, dir, 1, B, head, K, stk, stride, cnt, valid);   // (every lane: wave-wide appends)
    const bool to_mesh = valid && head != RT_NONE;
    const float4 q0 = make_float4(o.x, o.y, o.z, 1.0f), q1 = make_float4(dir.x, dir.y, dir.z, 1.0f), q2 = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    const uint32_t wi = seg_push(to_mesh, QM.counts, seg, QM.segcap, F.overflow);
    if (wi != RT_NONE) {
      QM.q0[wi] = q0; QM.q1[wi] = q1; QM.q2[wi] = q2; QM.q3[wi] = m;
      uint4 it;
      it.x = wi; it.y = __float_as_uint(B.t); it.z = (uint32_t)(B.inst + 1) | (B.front ? 0x80000000u : 0u); it.w = head;
      MQ.item4[wi] = it;
    }
    shade_point<CNT, MC, PH, XM>(S, P, F, QM, 0u, QN, GH, GS, GQ, seg, valid && !to_mesh, m, q0, q1, q2, B, cnt, n_shades);
  }
  // camera samples per segment: what the host measures the queues' fan-out against
  const uint32_t tiles = (total + 63u) / 64u;
  for (uint32_t sgm = RT_TID; sgm < nseg; sgm += RT_NTHREADS) {
    uint32_t nt = tiles > sgm ? (tiles - sgm + nseg - 1u) / nseg : 0u;
    uint32_t c = nt * 64u;
    if (nt && (tiles - 1u) % nseg == sgm) c -= tiles * 64u - total;  // the last tile may be partial
    l0_counts[sgm * RT_CSTRIDE] = c;
  }
  if (CNT) {
    cnt.flush(F.counters);
    cnt.flush_closest(F.counters);
    if (traced) rt_atomic_add_u64(&F.counters->rays_primary, traced);
    if (n_shades) rt_atomic_add_u64(&F.counters->shades, n_shades);
  }
}

// Secondary rays (level >= 1): the children k_shade / the fused kernels appended to Q.
template <bool CNT, bool TLAS, bool MC, bool PH, bool XM>
RT_KERNEL_W(RT_FUSED_WAVES, k_trace_shade, DScene S, PassP P, FrameBufs F, RayQ Q, RayQ QM, MeshQ MQ, RayQ QN, GroupQ GH, GroupQ GS, GatherQ GQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t traced = 0, n_shades = 0;
  RT_SEG_BEGIN(Q.counts, Q.nseg, Q.segcap, 1u)
    const uint32_t seg = sw_.s % QN.nseg;
    const uint32_t idx = seg_base_ + (uint32_t)seg_e;
    uint32_t head = RT_NONE;
    Best B;
    best_init(B);
    uint4 m;
    m.x = m.y = m.z = m.w = 0;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = make_float4(1, 0, 0, 0), q2 = q0;
    bool trace = false;
    if (seg_valid) {
      m = Q.q3[idx];
      q0 = Q.q0[idx]; q1 = Q.q1[idx]; q2 = Q.q2[idx];
      trace = !(m.w & (1u << 16));   // (re-shade entries carry a hit record written by their parent, lights.cpp:172-183)
      if (trace) traced++;
    }
    const V3 o = v3(q0.x, q0.y, q0.z), dir = v3(q1.x, q1.y, q1.z);
    // secondary rays see both sides (lights.cpp:185)
    const CandSink K = cand_sink(MQ, seg, F.overflow);
    if (TLAS) { if (trace) closest_instances_tlas<CNT, true>(S, o, dir, 3, B, head, K, stk, stride, cnt); }
    else closest_instances<CNT, true>(S, o, dir, 3, B, head, K, stk, stride, cnt, trace);   // (every lane: wave-wide appends)
    const bool to_mesh = trace && head != RT_NONE;
    const uint32_t wi = seg_push(to_mesh, QM.counts, seg, QM.segcap, F.overflow);
    if (wi != RT_NONE) {
      QM.q0[wi] = q0; QM.q1[wi] = q1; QM.q2[wi] = q2; QM.q3[wi] = m;
      uint4 it;
      it.x = wi; it.y = __float_as_uint(B.t); it.z = (uint32_t)(B.inst + 1) | (B.front ? 0x80000000u : 0u); it.w = head;
      MQ.item4[wi] = it;
    }
    shade_point<CNT, MC, PH, XM>(S, P, F, Q, idx, QN, GH, GS, GQ, seg, seg_valid && !to_mesh, m, q0, q1, q2, B, cnt, n_shades);
  RT_SEG_END
  if (CNT) {
    cnt.flush(F.counters);
    cnt.flush_closest(F.counters);
    if (traced) rt_atomic_add_u64(&F.counters->rays_secondary, traced);
    if (n_shades) rt_atomic_add_u64(&F.counters->shades, n_shades);
  }
}

// ------------------------------------------------------------------------------------------------ shadows
// Shadow rays in the same two phases as the closest-hit kernels: k_shadow_hard / k_shadow_soft1 / k_shadow_soft2 test
// spheres and planes and cull the mesh instances' world boxes; rays that reach a mesh box undecided become items of
// k_shadow_mesh, which walks the BVHs (any-hit) with persistent, refilling waves.
RT_DEV void hard_shadow_ray(const DScene& S, float4 g0, V3& Pp, V3& d, float& tmax) {
  Pp = v3(g0.x, g0.y, g0.z);
  const DLight& L = S.lights[__float_as_uint(g0.w)];
  if (L.kind == 2) {  // DirectLight::Illuminate, lights.h:45
    d = -v3(L.direction[0], L.direction[1], L.direction[2]);
    tmax = RT_BIG;
  } else {            // PointLight of size 0: the reference's 16..128 rays are identical
    V3 toLight = v3(L.position[0], L.position[1], L.position[2]) - Pp;
    tmax = length(toLight);
    d = toLight / tmax;
  }
}
// Soft shadows, PointLight::Illuminate (lights.cpp:34-60): the ray of sample j of a group.
RT_DEV void soft_shadow_ray(const DScene& S, float4 g0, float4 g2, float4 g3, float4 g4, uint32_t j, V3& Pp, V3& d, float& dist) {
  Pp = v3(g0.x, g0.y, g0.z);
  const DLight& L = S.lights[__float_as_uint(g0.w)];
  V3 position = v3(L.position[0], L.position[1], L.position[2]);
  V3 tangent = v3(g3.x, g3.y, g3.z), bitangent = v3(g4.x, g4.y, g4.z);
  float size = g3.w;
  const float twoPi = (float)(2 * RT_PI_D);
  float discX = S.halton[RT_H2 + (j & 127)] + g2.x;
  float discY = S.halton[RT_H3 + (j & 127)] + g2.y;
  discX -= (discX > 1.0f) ? 1.0f : 0.0f;
  discY -= (discY > 1.0f) ? 1.0f : 0.0f;
  float r = sqrtf(discX) * size;
  float angle = twoPi * discY;
  float sn, cs;
  sincosf(angle, &sn, &cs);
  float offsetU = r * cs;
  float offsetV = r * sn;
  V3 lsp = position + (tangent * offsetU) + (bitangent * offsetV);
  V3 toLight = lsp - Pp;
  dist = length(toLight);
  d = toLight / dist;
}

// Hard shadows: one ray per group (DirectLight, and PointLight of size 0).
#ifndef RT_SHADOW_P1_WAVES
#define RT_SHADOW_P1_WAVES 8   // C3 shadows: 5 -> 108.7 ms, 6 -> 105.3 ms, 8 -> 102.9 ms per step
#endif
template <bool CNT, bool TLAS>
RT_KERNEL_W(RT_SHADOW_P1_WAVES, k_shadow_hard, DScene S, PassP P, FrameBufs F, GroupQ G, MeshQ MQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t rays = 0;
  RT_SEG_BEGIN(G.counts, G.nseg, G.segcap, 1u)
    uint32_t g = seg_base_ + (uint32_t)seg_e;
    uint32_t head = RT_NONE;
    V3 Pp = v3(0, 0, 0), d = v3(1, 0, 0);
    float tmax = 0.0f;
    if (seg_valid) {
      hard_shadow_ray(S, G.g0[g], Pp, d, tmax);
      rays++;
    }
    const CandSink K = cand_sink(MQ, sw_.s % MQ.nseg, F.overflow);
    bool occ;
    if (TLAS) occ = seg_valid && any_instances_tlas<CNT, true>(S, Pp, d, tmax, head, K, stk, stride, cnt);
    else occ = any_instances<CNT, true>(S, Pp, d, tmax, head, K, stk, stride, cnt, seg_valid);  // (every lane: wave-wide cull and appends)
    bool need_mesh = seg_valid && !occ && head != RT_NONE;
    if (seg_valid && !occ && head == RT_NONE) {
      float4 g1 = G.g1[g];
      add_radiance(F, __float_as_uint(g1.w), v3(g1.x, g1.y, g1.z));
    }
    uint32_t wi = seg_push(need_mesh, MQ.counts, sw_.s % MQ.nseg, MQ.segcap, F.overflow);
    if (wi != RT_NONE) {
      uint4 it;
      it.x = g; it.y = 0; it.z = head; it.w = __float_as_uint(tmax);
      MQ.item4[wi] = it;
    }
  RT_SEG_END
  if (CNT) {
    cnt.flush(F.counters);
    if (rays) rt_atomic_add_u64(&F.counters->rays_shadow, rays);
  }
}

// Soft shadows, phases of PointLight::Illuminate's sample loop: `j0` .. `j0 + nj - 1` of every group (`list` = null:
// all groups of the queue; else the groups listed, i.e. those not fully lit after the first shadow_min samples).
template <bool CNT, bool TLAS>
RT_KERNEL_W(RT_SHADOW_P1_WAVES, k_shadow_soft, DScene S, PassP P, FrameBufs F, GroupQ G, const uint32_t* list, const uint32_t* list_counts, uint32_t j0,
          uint32_t nj, MeshQ MQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  uint32_t rays = 0;
  const uint32_t* counts = list ? list_counts : G.counts;
  const unsigned long long nj_magic = ((1ull << 40) + nj - 1u) / nj;   // ceil(2^40 / nj): e / nj == (e * magic) >> 40 for e < 2^32
  RT_SEG_BEGIN(counts, G.nseg, G.segcap, nj)
    uint32_t g = RT_NONE, j = 0;
    unsigned long long inst_mask = ~0ull;
    uint32_t head = RT_NONE;
    bool lit = false;
    V3 Pp = v3(0, 0, 0), d = v3(1, 0, 0);
    float dist = 0.0f;
    if (seg_valid) {
      // group and sample of work item seg_e: a 64-bit division per ray was the largest single cost of this kernel;
      // items per segment fit 32 bits and nj <= 128, for which the multiply-shift below is exact
      const uint32_t e32 = (uint32_t)seg_e;
      const uint32_t q = nj <= 128u ? (uint32_t)(((unsigned long long)e32 * nj_magic) >> 40) : e32 / nj;
      uint32_t e = seg_base_ + q;
      g = list ? list[e] : e;
      j = j0 + (e32 - q * nj);
      const float4 g4 = G.g4[g], g2 = G.g2[g];
      inst_mask = ((unsigned long long)__float_as_uint(g2.w) << 32) | (unsigned long long)__float_as_uint(g4.w);
      soft_shadow_ray(S, G.g0[g], g2, G.g3[g], g4, j, Pp, d, dist);
      rays++;
    }
    const CandSink K = cand_sink(MQ, sw_.s % MQ.nseg, F.overflow);
    bool occ;
    if (TLAS) occ = seg_valid && any_instances_tlas<CNT, true>(S, Pp, d, dist, head, K, stk, stride, cnt);
    else occ = any_instances<CNT, true, true>(S, Pp, d, dist, head, K, stk, stride, cnt, seg_valid, inst_mask);  // (every lane: wave-wide cull and appends)
    const bool need_mesh = seg_valid && !occ && head != RT_NONE;
    if (seg_valid && !occ && head == RT_NONE) lit = true;
    run_count_add(lit, g, seg_valid ? (uint32_t*)&G.g2[g].z : nullptr);
    uint32_t wi = seg_push(need_mesh, MQ.counts, sw_.s % MQ.nseg, MQ.segcap, F.overflow);
    if (wi != RT_NONE) {
      uint4 it;
      it.x = g; it.y = j; it.z = head; it.w = __float_as_uint(dist);
      MQ.item4[wi] = it;
    }
  RT_SEG_END
  if (CNT) {
    cnt.flush(F.counters);
    if (rays) rt_atomic_add_u64(&F.counters->rays_shadow, rays);
  }
}

// Second phase of the shadow kernels: any-hit walk over the mesh instances for the undecided rays.
#ifndef RT_SHADOW_ORDERED
#define RT_SHADOW_ORDERED 1   // any-hit walk: 1 = nearer child first (the answer does not depend on the order, the number of steps does)
#endif
#ifndef RT_SHADOW_MESH_WAVES
#define RT_SHADOW_MESH_WAVES 6   // 4 -> 4.19 ms, 5 -> 3.81 ms, 6 -> 3.59 ms, 8 -> 4.56 ms per C2 step
#endif
// Occluder hints.  An any-hit walk may visit the children in any order, and most shadow rays that reach a mesh ARE occluded
// (C5: 85 %, and a ray needed 59 node steps to find its occluder walking nearer-child-first: the ground under its origin
// comes first).  Rays that follow each other through a lane — samples of one soft-shadow group, shadow rays of neighbouring
// shade points — are mostly blocked by the same piece of geometry, so a walk is given the triangle slot that blocked the
// previous ray (of its group: GroupQ::hint, else of its lane) and, while it stands on the path to that slot, descends
// towards it first: a node's children cover consecutive slot ranges and the node record names where the right one begins.
// The walk tests exactly the boxes on that path, the hinted leaf, and then carries on from its stack with the leaf's
// neighbours; nothing is skipped and nothing extra is tested that a full walk would not test, so the answer is the one
// TraceBVHNodeShadow gives (objects.cpp:341-371) whatever the hint holds.  Measured (MI355X): C5 (12.6 M triangles, trees 24 deep)
// k_shadow_mesh 758 -> 520 ms per step; C2 (81 k triangles, hard shadows) 2.49 -> 2.59 ms and finalProject 13.4 -> 13.7 ms —
// there an occluder is found in ~18 steps anyway and the six extra instructions per node step are all that shows.  HINT is
// therefore a kernel variant, used for soft-shadow walks of scenes whose mesh trees are deep (rt_scene_commit, "shadow_hints").
template <bool CNT, bool SOFT, bool HINT>
RT_KERNEL_W(RT_SHADOW_MESH_WAVES, k_shadow_mesh, DScene S, PassP P, FrameBufs F, GroupQ G, MeshQ MQ) {
  RT_STACK_DECL
  Cnt<CNT> cnt;
  const uint32_t nseg = MQ.nseg;
  uint32_t seg = rt_wave_id() % nseg, segs_done = 0;
  const uint32_t refill_min = RT_REFILL_MIN < RT_WAVE ? RT_REFILL_MIN : RT_WAVE;
  int state = RT_ST_IDLE;
  uint32_t group = 0;
  float tmax = 0.0f;
  uint32_t guard = 0;
  uint32_t hint_minst = RT_NONE, hint_slot = 0;   // the lane's current hint: (position in DScene::minst_rec, triangle slot)
  MeshWalk W;
  W.p = W.d = W.inv = v3(0, 0, 0);
  W.cur = RT_NOREF; W.pend = 0; W.node_base = W.tri_base = 0; W.sp = 0; W.next = RT_NONE; W.inst = -1; W.fast = true;
  W.hint = 0; W.onpath = false; W.minst = RT_NONE;
  W.root = RT_NOREF; W.skip = RT_NOREF; W.restart = false; W.anc_base = RT_NONE; W.anc_blocks = 0;
#ifdef RT_SM_STATS   // developer build: node steps of the rays that end lit / occluded (scripts/dev/sm_stats.sh)
  uint32_t my_steps = 0; unsigned long long st_lit = 0, st_occ = 0, n_lit = 0, n_occ = 0;
#endif
  for (;;) {  // one wave-wide step per iteration, as in k_trace_mesh
    if (state == RT_ST_WALK && W.pend == 0 && ref_is_leaf(W.cur)) { W.pend = W.cur; W.cur = walk_pop(W, stk, stride); W.onpath = false; }
    if (state == RT_ST_WALK && W.cur == RT_NOREF && W.pend == 0) {
      if (HINT && W.restart) {   // nothing in the subtree around the hint: now the whole tree, without that subtree
        W.restart = false;
        W.cur = W.root;
        W.onpath = true;
      } else {
        state = RT_ST_NEXT;
      }
    }
    const bool walking = state == RT_ST_WALK;
    const bool at_node = walking && ref_is_node(W.cur);
    const uint32_t n_walk = count_lanes(walking), n_node = count_lanes(at_node), n_pend = count_lanes(walking && W.pend != 0);
    const uint32_t n_serv = count_lanes(state == RT_ST_NEXT || (state == RT_ST_IDLE && segs_done < nseg));
    if (n_walk == 0 && n_serv == 0) break;
    if (++guard > RT_WALK_GUARD) { *F.overflow = RT_WALK_ABORT; break; }
    if (n_serv >= refill_min || n_walk == 0) {
      // service: next candidate of the lane's ray, or — no mesh occludes it either — the ray is lit; then new items
      uint32_t enter = RT_NONE;
      if (state == RT_ST_NEXT) {
        enter = W.next;
        if (enter == RT_NONE) {
#ifdef RT_SM_STATS
          st_lit += my_steps; n_lit++; my_steps = 0;
#endif
          if (SOFT) {
            rt_atomic_add_u32((uint32_t*)&G.g2[group].z, 1u);
          } else {
            float4 g1 = G.g1[group];
            add_radiance(F, __float_as_uint(g1.w), v3(g1.x, g1.y, g1.z));
          }
          state = RT_ST_IDLE;
        }
      }
      if (segs_done < nseg) {
        bool exhausted;
        uint32_t wi = claim_items(state == RT_ST_IDLE, MQ, seg, exhausted);
        if (wi != RT_NONE) {
          uint4 it = MQ.item4[wi];
          group = it.x;
          enter = it.z;
          tmax = __uint_as_float(it.w);   // (the ray itself is in the candidates, already in each instance's space)
          if (SOFT && HINT) {  // the group's own hint, when it has one for a mesh (else the lane keeps its last)
            const uint2 gh = G.hint[group];
            if (gh.x < (uint32_t)S.n_minst) { hint_minst = gh.x; hint_slot = gh.y; }
          }
        }
        if (exhausted) { seg = (seg + 1) % nseg; segs_done++; }
      }
      if (enter != RT_NONE) {
        mesh_walk_enter<CNT>(S, MQ.cand, enter, W, cnt);
        W.hint = hint_slot;
        W.onpath = HINT && W.minst == hint_minst;
        if (HINT && W.onpath && W.fast && (hint_slot >> RT_ANC_SHIFT) < W.anc_blocks) {
// Around the hint first.  59 % of the occluded rays of C5 are blocked within 64 triangle slots of their hint, 72 %
          // within 256, 81 % within 1024 — and a walk from the root, even steered, spends ~22 of its 33 node steps on the way down.
          // The walk begins at the smallest inner node that holds the hint's block of 1024 slots; only if nothing occludes the ray
          // there does it start again at the root (skipping that subtree): 33 -> 22 node steps per occluded ray, lit rays unchanged.
          // The answer does not change: the slab test is monotone in the box — every quantity it compares moves one way when the
          // box grows, in floating point too — and a node's box holds its children's (checked per mesh at commit), so a ray that
          // enters a box enters every ancestor's as well: a triangle tested here is one the walk from the root tests too.  (Rays
          // with a non-finite reciprocal direction take the literal NaN-aware form of the test, which is not monotone: they
          // start at the root.)  Blocks of 64 / 256 / 1024 / 4096 slots: 561 / 545 / 544 / 553 ms per C5 step against 587; a walk
          // that starts at the 64-slot block and climbs parent by parent through the root (every node at most once) takes fewer
          // node steps still (19) and more time (661 ms): near the leaves it tests sibling leaves that a box higher up prunes.
          const uint32_t a = S.bvh_anc[W.anc_base + (hint_slot >> RT_ANC_SHIFT)];
          if (a != W.root) { W.cur = a; W.skip = a; W.restart = true; }
        }
        state = RT_ST_WALK;
      }
      continue;
    }
    if (n_node != 0 && (n_pend < RT_LEAF_MIN || n_node >= RT_WAVE / 2) && (n_node >= RT_NODE_MIN || n_pend < RT_NODE_MIN)) {
#pragma unroll
      for (int k = 0; k < RT_NODE_BATCH; k++) {
        if (k > 0 && walking && W.pend == 0 && ref_is_leaf(W.cur)) { W.pend = W.cur; W.cur = walk_pop(W, stk, stride); W.onpath = false; }
        if (walking && ref_is_node(W.cur)) {
          float4 n0, n1, n2, n3;
          fetch_node(S, W, true, n0, n1, n2, n3);
          mesh_walk_node<CNT, RT_SHADOW_ORDERED != 0, HINT>(W, n0, n1, n2, n3, tmax, stk, stride, cnt);
#ifdef RT_SM_STATS
          my_steps++;
#endif
        }
      }
    } else if (walking && W.pend != 0) {
      uint32_t at_slot = 0;
      bool occluded = mesh_walk_leaf_any<CNT>(S, W, W.pend, tmax, cnt, at_slot);
      W.pend = 0;
      if (occluded) {
        state = RT_ST_IDLE;
#ifdef RT_SM_STATS
        st_occ += my_steps; n_occ++; my_steps = 0;
#endif
        if (HINT && (hint_minst != W.minst || hint_slot != at_slot)) {
          hint_minst = W.minst; hint_slot = at_slot;
          if (SOFT) { uint2 gh; gh.x = hint_minst; gh.y = hint_slot; G.hint[group] = gh; }
        }
      }
    }
  }
#ifdef RT_SM_STATS
  if (CNT) {
    rt_atomic_add_u64(&F.counters->photon_visits, st_lit); rt_atomic_add_u64(&F.counters->gathers, st_occ);
    rt_atomic_add_u64(&F.counters->tex_evals, n_lit); rt_atomic_add_u64(&F.counters->shades, n_occ);
  }
#endif
  if (CNT) cnt.flush(F.counters);
}

// Groups with a clear view of the whole light (cone_clear) are fully lit without a ray: their term is added here, as
// k_shadow_soft_decide adds it for a group whose shadow_min rays all came through (vis = 1).  Groups in the full shadow of
// a sphere contribute vis = 0 (k_shadow_soft_final's 0 / shadow_max): nothing is added.  The others are listed for the
// sample phases.
template <bool CNT, bool TLAS>
RT_KERNEL(k_shadow_soft_clear, DScene S, PassP P, FrameBufs F, GroupQ G, uint32_t* list1, uint32_t* n1) {
  RT_STACK_DECL
  uint32_t proven = 0;
  RT_SEG_BEGIN(G.counts, G.nseg, G.segcap, 1u)
    uint32_t g = seg_base_ + (uint32_t)seg_e;
    V3 Pp = v3(0, 0, 0), C = v3(0, 0, 1);
    float R = 0.0f;
    if (seg_valid) {
      const float4 g0 = G.g0[g];
      const DLight& L = S.lights[__float_as_uint(g0.w)];
      Pp = v3(g0.x, g0.y, g0.z);
      C = v3(L.position[0], L.position[1], L.position[2]);
      R = G.g3[g].w;
    }
    unsigned long long may_occlude = ~0ull;
    int view;
    if (TLAS) view = seg_valid ? cone_clear_tlas(S, Pp, C, R, true, stk, stride) : RT_CONE_TRACE;   // (per lane: the instance BVH)
    else view = cone_clear(S, Pp, C, R, seg_valid, may_occlude);                                    // (every lane: wave-uniform table walk)
    if (seg_valid && view == RT_CONE_CLEAR) {
      const float4 g1 = G.g1[g];
      add_radiance(F, __float_as_uint(g1.w), v3(g1.x, g1.y, g1.z) * 1.0f);
    }
    if (CNT && seg_valid) proven += view == RT_CONE_CLEAR ? (uint32_t)P.shadow_min : (view == RT_CONE_UMBRA ? (uint32_t)(P.shadow_max > P.shadow_min ? P.shadow_max : P.shadow_min) : 0u);
    uint32_t li = seg_push(seg_valid && view == RT_CONE_TRACE, n1, sw_.s, G.segcap, F.overflow);
    if (li != RT_NONE) {
      list1[li] = g;
      ((uint32_t*)&G.g4[g])[3] = (uint32_t)may_occlude;   // the rays of this group are tested against these instances only (k_shadow_soft)
      ((uint32_t*)&G.g2[g])[3] = (uint32_t)(may_occlude >> 32);
    }
  RT_SEG_END
  if (CNT && proven) rt_atomic_add_u64(&F.counters->rays_shadow_proven, proven);
}

// early-out rule (lights.cpp:56-59): fully lit after shadow_min samples => done, else the group continues
// (`list1`: the groups the first sample phase ran for; null = every group of the queue)
RT_KERNEL(k_shadow_soft_decide, PassP P, FrameBufs F, GroupQ G, const uint32_t* list1, const uint32_t* n1, uint32_t* list2, uint32_t* n2) {
  RT_SEG_BEGIN((list1 ? n1 : G.counts), G.nseg, G.segcap, 1u)
    uint32_t g = seg_base_ + (uint32_t)seg_e;
    if (list1 && seg_valid) g = list1[g];
    bool more = false;
    if (seg_valid) {
      uint32_t lit = __float_as_uint(G.g2[g].z);
      if (lit == (uint32_t)P.shadow_min || P.shadow_max <= P.shadow_min) {
        float4 g1 = G.g1[g];
        float vis = (float)lit / (float)P.shadow_min;
        add_radiance(F, __float_as_uint(g1.w), v3(g1.x, g1.y, g1.z) * vis);
      } else {
        more = true;
      }
    }
    uint32_t li = seg_push(more, n2, sw_.s, G.segcap, F.overflow);
    if (li != RT_NONE) list2[li] = g;
  RT_SEG_END
}
RT_KERNEL(k_shadow_soft_final, PassP P, FrameBufs F, GroupQ G, const uint32_t* list2, const uint32_t* n2) {
  RT_SEG_BEGIN(n2, G.nseg, G.segcap, 1u)
    if (!seg_valid) continue;
    uint32_t g = list2[seg_base_ + (uint32_t)seg_e];
    float4 g1 = G.g1[g];
    float vis = (float)__float_as_uint(G.g2[g].z) / (float)P.shadow_max;  // summedLight / numSamples, lights.cpp:66
    add_radiance(F, __float_as_uint(g1.w), v3(g1.x, g1.y, g1.z) * vis);
  RT_SEG_END
}

// ------------------------------------------------------------------------------------------------ accumulate / resolve
RT_HD float t_value(int i) {  // raytracer.h:52-58, index clamped to the table
  const float T[71] = {0,      12.706f, 4.303f, 3.182f, 2.776f, 2.571f, 2.447f, 2.365f, 2.306f, 2.262f, 2.228f, 2.201f,
                       2.179f, 2.160f,  2.145f, 2.131f, 2.120f, 2.110f, 2.101f, 2.093f, 2.086f, 2.080f, 2.074f, 2.069f,
                       2.064f, 2.060f,  2.056f, 2.052f, 2.048f, 2.045f, 2.042f, 2.040f, 2.037f, 2.035f, 2.032f, 2.030f,
                       2.028f, 2.026f,  2.024f, 2.023f, 2.021f, 2.020f, 2.018f, 2.017f, 2.015f, 2.014f, 2.013f, 2.012f,
                       2.011f, 2.010f,  2.009f, 2.000f, 2.000f, 2.000f, 2.000f, 2.000f, 2.000f, 2.000f, 2.000f, 2.000f,
                       2.000f, 1.994f,  1.994f, 1.994f, 1.994f, 1.994f, 1.994f, 1.994f, 1.994f, 1.994f, 1.994f};
  return T[i > 70 ? 70 : i];
}

// RayTracer.cpp:176-198 per pixel, samples in index order; accum = sum rgb, n, sum of squares rgb, done flag
RT_KERNEL(k_accumulate, PassP P, FrameBufs F) {
  if (P.discard || *F.overflow) return;  // a pass whose queues overflowed is re-run in smaller pieces by the host
  for (uint32_t j = RT_TID; j < P.n_pix; j += RT_NTHREADS) {
    uint32_t pixel = F.pixels[j];
    float* a = F.accum + 8 * (size_t)pixel;
    float sr = a[0], sg = a[1], sb = a[2], n = a[3], qr = a[4], qg = a[5], qb = a[6], done = a[7];
    for (int k = 0; k < P.n_samples && done == 0.0f; k++) {
      const size_t slot = (size_t)k * P.n_pix + j;
      const float cx = F.path_rad[slot], cy = F.path_rad[F.rad_stride + slot], cz = F.path_rad[2 * (size_t)F.rad_stride + slot];
      sr += cx; sg += cy; sb += cz;
      qr += cx * cx; qg += cy * cy; qb += cz * cz;
      n += 1.0f;
      if (P.adaptive) {
        int i = P.sample_base + k;
        if (i >= P.spp_min) {
          float nn = (float)(i + 1);
          float vr = (qr - (sr * sr) / nn) / (nn - 1.0f);
          float vg = (qg - (sg * sg) / nn) / (nn - 1.0f);
          float vb = (qb - (sb * sb) / nn) / (nn - 1.0f);
          vr = fmaxf(vr, 0.0f); vg = fmaxf(vg, 0.0f); vb = fmaxf(vb, 0.0f);
          float t = t_value((int)nn - 1);
          float sn = sqrtf(nn);
          float pr = t * (sqrtf(vr) / sn), pg = t * (sqrtf(vg) / sn), pb = t * (sqrtf(vb) / sn);
          if (pr <= P.adaptive_threshold && pg <= P.adaptive_threshold && pb <= P.adaptive_threshold) done = 1.0f;
        }
        if (i + 1 >= P.spp_max) done = 1.0f;  // fall-through: totalSamples = maxSamples
      }
    }
    a[0] = sr; a[1] = sg; a[2] = sb; a[3] = n; a[4] = qr; a[5] = qg; a[6] = qb; a[7] = done;
  }
}

// What the host needs to know about a finished pass, in 1 KB instead of the whole table of per-segment counters:
// per (level, kind) slot the sum and the maximum of its segment lengths, then the overflow word.
RT_KERNEL(k_pass_summary, const uint32_t* counts, uint32_t n_slots, uint32_t nseg, uint32_t slot_words, uint32_t overflow_at, uint32_t* out) {
  for (uint32_t s = RT_TID; s < n_slots; s += RT_NTHREADS) {
    uint32_t sum = 0, mx = 0;
    for (uint32_t g = 0; g < nseg; g++) {
      const uint32_t v = counts[(size_t)s * slot_words + (size_t)g * RT_CSTRIDE];
      const uint32_t ns = sum + v;
      sum = ns < sum ? 0xffffffffu : ns;
      mx = v > mx ? v : mx;
    }
    out[s] = sum;
    out[n_slots + s] = mx;
  }
  if (RT_TID == 0) out[2 * n_slots] = counts[overflow_at];
}

// pixels still sampling after an adaptive pass
RT_KERNEL(k_compact_active, PassP P, FrameBufs F, uint32_t* next_pixels, uint32_t* next_count) {
  for (uint32_t j = RT_TID; j < P.n_pix; j += RT_NTHREADS) {
    uint32_t pixel = F.pixels[j];
    if (F.accum[8 * (size_t)pixel + 7] == 0.0f) next_pixels[rt_atomic_add_u32(next_count, 1u)] = pixel;
  }
}

// RayTracer.cpp:206-216
RT_KERNEL(k_resolve, int W, int H, int srgb, const float* accum, uint8_t* rgb8, float* radiance, int32_t* sample_count) {
  uint32_t total = (uint32_t)W * (uint32_t)H;
  for (uint32_t p = RT_TID; p < total; p += RT_NTHREADS) {
    const float* a = accum + 8 * (size_t)p;
    float n = a[3];
    float r = a[0] / n, g = a[1] / n, b = a[2] / n;
    if (n == 0.0f) { r = g = b = 0.0f; }  // pixel not covered by this call's tiles
    if (radiance) {
      radiance[4 * (size_t)p + 0] = r; radiance[4 * (size_t)p + 1] = g; radiance[4 * (size_t)p + 2] = b;
      radiance[4 * (size_t)p + 3] = n;
    }
    if (srgb) { r = linear_to_srgb(r); g = linear_to_srgb(g); b = linear_to_srgb(b); }
    if (rgb8) {
      rgb8[3 * (size_t)p + 0] = to_byte(r); rgb8[3 * (size_t)p + 1] = to_byte(g); rgb8[3 * (size_t)p + 2] = to_byte(b);
    }
    if (sample_count) sample_count[p] = (int32_t)n;
  }
}

// ------------------------------------------------------------------------------------------------ batched trace API
struct HitOut {  // rt_hit
  float p[3];
  float z;
  float N[3];
  float uvw[3];
  int32_t node, face, light, front;
};
RT_KERNEL(k_api_closest, DScene S, const float* rays, uint32_t n, int hitSide, HitOut* out) {
  RT_STACK_DECL
  Cnt<false> cnt;
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    const float* r = rays + 6 * (size_t)i;
    Hit h;
    bool ok = scene_closest<false>(S, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), hitSide, h, stk, stride, cnt);
    HitOut o;
    o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
    o.z = ok ? h.z : RT_BIG;
    o.N[0] = h.N.x; o.N[1] = h.N.y; o.N[2] = h.N.z;
    o.uvw[0] = h.uvw.x; o.uvw[1] = h.uvw.y; o.uvw[2] = h.uvw.z;
    o.node = (ok && h.inst >= 0) ? S.inst[h.inst].node : -1;
    o.face = (ok && h.light < 0) ? h.face : -1;
    o.light = ok ? h.light : -1;
    o.front = h.front ? 1 : 0;
    out[i] = o;
  }
}
RT_KERNEL(k_api_any, DScene S, const float* rays, const float* tmax, uint32_t n, uint8_t* out) {
  RT_STACK_DECL
  Cnt<false> cnt;
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    const float* r = rays + 6 * (size_t)i;
    out[i] = scene_any<false>(S, v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), tmax ? tmax[i] : RT_BIG, stk, stride, cnt) ? 1 : 0;
  }
}

// rt_debug_philox: the sampler's generator and its float mapping as the kernels run them (known-answer tests)
RT_KERNEL(k_debug_philox, const uint32_t* ctr, uint32_t k0, uint32_t k1, uint32_t n, uint32_t* words, float* units) {
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) {
    uint32_t r[4];
    philox4x32_10(ctr[4 * (size_t)i], ctr[4 * (size_t)i + 1], ctr[4 * (size_t)i + 2], ctr[4 * (size_t)i + 3], k0, k1, r);
    for (int k = 0; k < 4; k++) {
      words[4 * (size_t)i + k] = r[k];
      units[4 * (size_t)i + k] = u32_to_unit(r[k]);
    }
  }
}

// rt_reduce_accumulators: floats 0..3 of every pixel (sum rgb, n) to and from a packed image
RT_KERNEL(k_acc_pack, const float4* accum, float4* packed, uint32_t n) {
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) packed[i] = accum[2 * (size_t)i];
}
RT_KERNEL(k_acc_unpack, const float4* packed, float4* accum, uint32_t n) {
  for (uint32_t i = RT_TID; i < n; i += RT_NTHREADS) accum[2 * (size_t)i] = packed[i];
}
