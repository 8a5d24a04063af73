// Device-side data layout and leaf routines of the renderer core (gfx950).
//
// Arithmetic notes.  Everything is FP32, compiled with -ffp-contract=off; division and sqrt are
// IEEE (hipcc default), so intersection decisions reproduce the reference's C++ expressions
// operation for operation (reference files cited per routine, relative to RayTracer/).
#pragma once
#include <float.h>
#include <math.h>

#include "rt_platform.h"

#define RT_BIG FLT_MAX            // BIGFLOAT, scene.h:34
#define RT_MAX_CHAIN 8            // maximum scene-graph depth of an instanced primitive
#ifndef RT_STACK
#define RT_STACK 32               // BVH traversal stack entries per lane (rt_scene_commit rejects deeper trees)
#endif
#define RT_LEAF 0x80000000u
#define RT_PI_F 3.14159265358979323846f
#define RT_PI_D 3.14159265358979323846

struct V3 {
  float x, y, z;
};
RT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float length(V3 a) { return sqrtf(dot(a, a)); }
RT_HD V3 normalized(V3 a) { return a / length(a); }
RT_HD bool nonzero(V3 a) { return a.x != 0.0f || a.y != 0.0f || a.z != 0.0f; }

// ------------------------------------------------------------------ device scene

struct DInstance {      // a node that carries a primitive (64 bytes); its chain of transforms lives in DScene::xf
  int32_t prim;         // RT_PRIM_*
  int32_t mesh;
  int32_t mtl;          // material index; n_materials = the default Blinn
  int32_t node;         // node index (== rt_node index)
  int32_t depth;        // levels in the chain root .. this node
  int32_t parent_key;   // node index of the parent (identifies the chain prefix), -1 for a root-level instance
  int32_t xf_off;       // float offset into DScene::xf of this instance's chain: depth x { itm[12], tm[12] }, root first
  union {
    uint32_t shade_base;  // mesh instances: index of the mesh's first face record in DScene::shade (copied from the mesh so
                          // that the shading attributes are one fetch away from the instance record)
    float stretch;        // spheres and planes: how far a unit world length can reach in the instance's space (chain_stretch)
  };
  float wmin[3];        // padded world-space bounds (culling of mesh instances only)
  uint32_t has_uvw;     // mesh instances: the mesh has texture coordinates
  float wmax[3];
  int32_t minst_k;      // mesh instances: position in DScene::minst / minst_rec
};

struct DMesh {
  uint32_t node_base;   // float4 index of this mesh's first 64-byte BVH node in DScene::bvh
  uint32_t tri_base;    // float4 index of this mesh's first 48-byte triangle slot in DScene::tris
  uint32_t shade_base;  // index of this mesh's first 80-byte face record in DScene::shade
  uint32_t anc_base, anc_blocks;   // as DMeshInst
  uint32_t pad_[1];
  uint32_t root_ref;    // child reference of the root (leaf bit | first_slot<<3 | count, or internal node index)
  uint32_t has_uvw;
  float bounds[6];      // root box: min xyz, max xyz
};

struct DMeshInst {      // what the second-phase walk needs to enter a mesh instance, in one 80-byte record (instead of
  float wmin[3];        // the index list -> instance record -> mesh record chain of dependent fetches)
  int32_t inst;         // index into DScene::inst
  float wmax[3];        // padded world box, as DInstance
  int32_t depth;
  float bounds[6];      // the mesh's root box, as DMesh
  int32_t xf_off;
  uint32_t root_ref;
  uint32_t node_base, tri_base;
  uint32_t anc_base;    // this mesh's first entry in DScene::bvh_anc (RT_NONE: none) ...
  uint32_t anc_blocks;  // ... and how many it has
};

struct DTexColor { float v[3]; int32_t map; };
struct DTexFloat { float v; int32_t map; };
struct DTexmap {
  int32_t kind, image;
  float itm[12];
  DTexColor color[2];
};
struct DImage { uint32_t offset; int32_t width, height; int32_t pad; };
struct DMaterial {
  int32_t kind;
  DTexColor diffuse, specular, emission, reflection, refraction;
  DTexFloat glossiness;
  float absorption[3];
  float ior;
};
struct DShadeInst {     // what k_shade needs of a hit instance in ONE record: the instance header's shading fields and a copy
  int32_t prim, node;   // of its material, so that the material is not a second dependent fetch behind the instance
  int32_t depth, xf_off;
  uint32_t shade_base, has_uvw;
  uint32_t uses_uvw;    // the material reads a texture map somewhere: only then are texture coordinates worth computing
  uint32_t face_pos_base;  // meshes with texture coordinates: index of the mesh's first face in DScene::face_pos, else RT_NONE
  DMaterial mtl;
  uint32_t pad2_;
};
struct DLight {
  int32_t kind;
  float intensity[3], position[3], direction[3];
  float size, attenuation;
};

struct DScene {
  const DInstance* inst;
  const DShadeInst* sinst;  // same index as inst
  const float* xf;      // transform chains, see DInstance::xf_off
  const DMesh* meshes;
  const float4* bvh;
  const uint32_t* bvh_anc;   // per mesh and block of 2^RT_ANC_SHIFT triangle slots: the smallest inner node whose slots hold the block
  const float4* tris;
  const float4* shade;  // per face, 5 float4: the three vertex normals and texture coordinates, indices resolved at commit
                        // (n0 | t0.x) (n1 | t0.y) (n2 | t0.z) (t1 | t2.x) (t2.y t2.z - -)
  const float4* face_pos;  // per face of the meshes that have texture coordinates, 3 float4: v0, e1, e2 as DScene::tris holds
                           // them per slot — the triangle a neighbouring pixel's ray is intersected with for the texture
                           // coordinate derivatives (RT_EXT_TEX_FILTER, hit_differentials)
  const DMaterial* materials;
  const DTexmap* texmaps;
  const DImage* images;
  const uint8_t* texels;
  const DLight* lights;
  const float* halton;  // [0,128) base 2, [128,256) base 3, [256,384) base 5, [384,512) base 7, then 1000 base 2,
                        // 1000 base 3 (HaltonSeq<1000>), then n_gloss base 2 and n_gloss base 3 (Halton(i, b) direct)
  int32_t n_inst, n_lights, n_materials, n_gloss;
  DTexColor background, environment;
  const int32_t* minst;  // indices (into inst) of the mesh instances, depth-first order
  const DMeshInst* minst_rec;  // ... and their records for the second-phase walk
  int32_t n_minst;
  int32_t use_tlas;      // closest-hit rays walk the instance BVH below: the scene has enough instances for it to pay (rt_scene_commit)
  const float4* tlas;    // BVH over the instances' padded world boxes, 64-byte nodes as DScene::bvh
  const int32_t* tlas_idx;  // instance indices in leaf order
  uint32_t tlas_root;
  int32_t shadow_tlas;   // ... and so do the shadow rays of the two-phase kernels.  Their waves are coherent (samples of a few shadow
                         // groups), so the wave-uniform table walk with its 64-entries-at-a-time box cull and the proofs' masks
                         // stays ahead of per-lane BVH walks up to a few hundred instances (finalProject.xml, 70: 3.1 -> G rays/s)
};
#define RT_H2 0
#define RT_H3 128
#define RT_H5 256
#define RT_H7 384
#define RT_HPHI 512
#define RT_HTHETA 1512
#define RT_HGLOSS 2512
#define RT_DUVW_MAX 64.0f   // largest texture-coordinate derivative per pixel that hit_differentials hands on
#define RT_FILTER_TAPS 32   // Texture::Eval(uvw, duvw): sampleCount (scene.h:293); offsets behind the gloss tables: x[32], y[32]

struct DCounters {  // mirrors rt_counters
  unsigned long long rays_primary, rays_secondary, rays_shadow, rays_photon;
  unsigned long long inst_tests, sphere_tests, plane_tests, node_visits, tri_tests;
  unsigned long long shades, tex_evals, photon_visits, gathers, paths;
  unsigned long long closest_inst_tests, closest_node_visits, closest_tri_tests;
  unsigned long long mesh_inst_tests, mesh_items;
  unsigned long long rays_shadow_proven;
};

// per-lane work counters kept in registers by the counted kernel variants
struct LaneCnt {
  uint32_t inst, sphere, plane, node, tri, tex;
};
template <bool CNT> struct Cnt;
template <> struct Cnt<true> {
  LaneCnt c;
  RT_MEM Cnt() { c.inst = c.sphere = c.plane = c.node = c.tri = c.tex = 0; }
  RT_MEM void inst(uint32_t n = 1) { c.inst += n; }
  RT_MEM void sphere() { c.sphere++; }
  RT_MEM void plane() { c.plane++; }
  RT_MEM void node(uint32_t n = 1) { c.node += n; }
  RT_MEM void tri() { c.tri++; }
  RT_MEM void tex() { c.tex++; }
  RT_MEM void flush(DCounters* g) {
    if (c.inst) rt_atomic_add_u64(&g->inst_tests, c.inst);
    if (c.sphere) rt_atomic_add_u64(&g->sphere_tests, c.sphere);
    if (c.plane) rt_atomic_add_u64(&g->plane_tests, c.plane);
    if (c.node) rt_atomic_add_u64(&g->node_visits, c.node);
    if (c.tri) rt_atomic_add_u64(&g->tri_tests, c.tri);
    if (c.tex) rt_atomic_add_u64(&g->tex_evals, c.tex);
  }
  RT_MEM void flush_closest(DCounters* g) {
    if (c.inst) rt_atomic_add_u64(&g->closest_inst_tests, c.inst);
    if (c.node) rt_atomic_add_u64(&g->closest_node_visits, c.node);
    if (c.tri) rt_atomic_add_u64(&g->closest_tri_tests, c.tri);
  }
};
template <> struct Cnt<false> {
  LaneCnt c;  // never touched; lets counted-only code compile in both variants
  RT_MEM void inst(uint32_t = 1) {}
  RT_MEM void sphere() {}
  RT_MEM void plane() {}
  RT_MEM void node(uint32_t = 1) {}
  RT_MEM void tri() {}
  RT_MEM void tex() {}
  RT_MEM void flush(DCounters*) {}
  RT_MEM void flush_closest(DCounters*) {}
};

// ------------------------------------------------------------------ transforms (scene.h:170-189)
template <class M>
RT_HD V3 xf_point(M m, V3 p) {
  return v3(m[0] * p.x + m[3] * p.y + m[6] * p.z + m[9], m[1] * p.x + m[4] * p.y + m[7] * p.z + m[10],
            m[2] * p.x + m[5] * p.y + m[8] * p.z + m[11]);
}
template <class M>
RT_HD V3 xf_dir(M m, V3 p) {
  return v3(m[0] * p.x + m[3] * p.y + m[6] * p.z, m[1] * p.x + m[4] * p.y + m[7] * p.z,
            m[2] * p.x + m[5] * p.y + m[8] * p.z);
}
template <class M>
RT_HD V3 xf_tmul(M m, V3 p) {  // (3x3)^T * p : NormalTransformFrom with itm
  return v3(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[3] * p.x + m[4] * p.y + m[5] * p.z,
            m[6] * p.x + m[7] * p.y + m[8] * p.z);
}

// GetOrthonormals as this build defines it (see oracle.cpp `orthonormals`)
RT_HD void orthonormals(V3 n, V3& v0, V3& v1) {
  if (n.z >= n.y) {
    float a = 1.0f / (1.0f + n.z);
    float b = -n.x * n.y * a;
    v0 = v3(1.0f - n.x * n.x * a, b, -n.x);
    v1 = v3(b, 1.0f - n.y * n.y * a, -n.y);
  } else {
    float a = 1.0f / (1.0f + n.y);
    float b = -n.z * n.x * a;
    v0 = v3(b, -n.z, 1.0f - n.z * n.z * a);
    v1 = v3(1.0f - n.x * n.x * a, -n.x, b);
  }
}

// ------------------------------------------------------------------ Philox4x32-10 / sampling
RT_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_HD float u32_to_unit(uint32_t u) {  // RNG::RandomFloat mapping, rng.h:60-65
  const float rmax = 0x1.fffffep-1f;
  float r = (float)u * 0x1p-32f;
  return r < rmax ? r : rmax;
}
#define RT_TAG_PIXEL 0u
#define RT_TAG_SHADE 1u
#define RT_TAG_PHOTON 2u

// ------------------------------------------------------------------ primitives

// Sphere::IntersectRay, objects.cpp:20-61 — decision part; attributes are filled once at the end.
RT_DEV bool sphere_closest(V3 p, V3 d, int hitSide, float& t, bool& front) {
  const float eps = 0.002f;
  float a = dot(d, d);
  float b = 2.0f * dot(d, p);
  float c = dot(p, p) - 1.0f;
  float disc = b * b - 4.0f * a * c;
  if (disc < 0.0f) return false;
  float sq = sqrtf(disc);
  float twoA = 2.0f * a;
  float t1 = (-b - sq) / twoA;
  float t2 = (-b + sq) / twoA;
  if (t1 > eps && (hitSide & 1)) {
    t = t1; front = true; return true;
  } else if (t2 >= eps && (hitSide & 2)) {
    t = t2; front = false; return true;
  }
  return false;
}
// Sphere::ShadowRay, objects.cpp:63-82 (the reference compares against the double 0.01; for float t that is
// the same set as t > 0.01f because 0.01f < 0.01 < nextafter(0.01f))
RT_DEV bool sphere_any(V3 p, V3 d, float t_max) {
  float a = dot(d, d);
  float b = 2.0f * dot(d, p);
  float c = dot(p, p) - 1.0f;
  float disc = b * b - 4.0f * a * c;
  if (disc < 0.0f) return false;
  float sq = sqrtf(disc);
  float twoA = 2.0f * a;
  float t1 = (-b - sq) / twoA;
  float t2 = (-b + sq) / twoA;
  if (t1 > 0.01f && t1 < t_max) return true;
  if (t2 > 0.01f && t2 < t_max) return true;
  return false;
}
// Plane::IntersectRay, objects.cpp:126-150 (0.0002 double vs float: same argument as above)
RT_DEV bool plane_closest(V3 p, V3 d, int hitSide, float& t, bool& front) {
  if (fabsf(d.z) < 1e-8f) return false;
  float tt = -(p.z / d.z);
  if (tt > 0.0002f) {
    float hx = p.x + d.x * tt, hy = p.y + d.y * tt;
    if (hx < -1.0f || hx > 1.0f || hy < -1.0f || hy > 1.0f) return false;
    bool f = (d.x * 0.0f + d.y * 0.0f + d.z * 1.0f) < 0.0f;
    if ((f && (hitSide & 1)) || (!f && (hitSide & 2))) {
      t = tt; front = f; return true;
    }
  }
  return false;
}
// Plane::ShadowRay, objects.cpp:152-167
RT_DEV bool plane_any(V3 p, V3 d, float t_max) {
  float t = -(p.z / d.z);
  if (t > 0.0002f && t < t_max) {
    float hx = p.x + d.x * t, hy = p.y + d.y * t;
    if (hx < -1.0f || hx > 1.0f || hy < -1.0f || hy > 1.0f) return false;
    return true;
  }
  return false;
}

// wave-wide minimum / maximum (every lane of the wave must call)
#ifndef RT_EMU
// (the result is read back through readfirstlane so that the compiler keeps it in a scalar register)
RT_DEV float wave_min(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}
RT_DEV float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}
RT_DEV bool wave_any(bool v) { return __ballot(v) != 0ull; }
RT_DEV uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
RT_DEV unsigned long long wave_or64(unsigned long long v) {
  return ((unsigned long long)wave_or((uint32_t)(v >> 32)) << 32) | (unsigned long long)wave_or((uint32_t)v);
}
#else
RT_DEV float wave_min(float v) { return v; }
RT_DEV float wave_max(float v) { return v; }
RT_DEV bool wave_any(bool v) { return v; }
RT_DEV uint32_t wave_or(uint32_t v) { return v; }
RT_DEV unsigned long long wave_or64(unsigned long long v) { return v; }
#endif

// ---- segmented queues ------------------------------------------------------------------------------
// One returning atomic on a single counter word saturates near 90 M/s on this chip, and a kernel that appends once
// per wave per iteration is bound by exactly that.  Every queue is therefore split into nseg (64) segments with
// their own length word (128 bytes apart): wave g appends to segment g % nseg, with ONE atomic per wave per append
// site (ballot + mbcnt ranks), and consumer kernels give each wave one segment to walk.  Segments stay dense, the
// contention per word drops 64x, and because children land in their parent's segment the worst-case length of a
// segment is exactly (worst-case queue length) / nseg — no slack, no holes.
#define RT_CSTRIDE 32u
#define RT_NONE 0xFFFFFFFFu
#ifndef RT_EMU
#define RT_WAVE 64u
RT_DEV uint32_t rt_lane() { return threadIdx.x & 63u; }
RT_DEV uint32_t rt_wave_id() { return RT_TID >> 6; }
RT_DEV uint32_t rt_num_waves() { return RT_NTHREADS >> 6; }
RT_DEV uint32_t rt_wave_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }   // lane 0's value (all lanes active)
RT_DEV uint32_t seg_push(bool pred, uint32_t* counts, uint32_t seg, uint32_t segcap, uint32_t* overflow) {
  unsigned long long mask = __ballot(pred);
  if (mask == 0) return RT_NONE;
  uint32_t n = (uint32_t)__popcll(mask);
  uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if ((int)rt_lane() == leader) base = atomicAdd(counts + seg * RT_CSTRIDE, n);
  base = (uint32_t)__shfl((int)base, leader);
  if (!pred) return RT_NONE;
  uint32_t pos = base + rank;
  if (pos >= segcap) {
    *overflow = 1u;
    return RT_NONE;
  }
  return seg * segcap + pos;
}
#else
#define RT_WAVE 1u
RT_DEV uint32_t rt_lane() { return 0u; }
RT_DEV uint32_t rt_wave_id() { return 0u; }
RT_DEV uint32_t rt_num_waves() { return 1u; }
RT_DEV uint32_t rt_wave_first(uint32_t v) { return v; }
RT_DEV uint32_t seg_push(bool pred, uint32_t* counts, uint32_t seg, uint32_t segcap, uint32_t* overflow) {
  if (!pred) return RT_NONE;
  uint32_t pos = counts[seg * RT_CSTRIDE]++;
  if (pos >= segcap) {
    *overflow = 1u;
    return RT_NONE;
  }
  return seg * segcap + pos;
}
#endif

// ---- candidates: what the first phase of the two-phase trace kernels hands to the second for every mesh instance a ray
// still has to walk — the ray ALREADY IN THE INSTANCE'S SPACE (the first phase transforms it with scalar-loaded matrices, at
// full lane occupancy, and has tested the instance's world box and the mesh's root box), so that entering an instance in the
// persistent walk kernels is one 32-byte fetch instead of an instance loop with box tests and a chain of matrix fetches.
// A ray's candidates form a linked list (each record names the ray's previous one).
struct CandSink {
  float4* rec;          // 2 float4 per candidate: (local origin, index into DScene::minst_rec) (local direction, next candidate)
  uint32_t* count;      // length word of the segment this wave appends to
  uint32_t base;        // index of the segment's first record
  uint32_t segcap;
  uint32_t* overflow;
};
// wave-uniform call sites (every lane of the wave calls): one atomic per wave
RT_DEV uint32_t cand_push(const CandSink& K, bool pred) {
  uint32_t at = seg_push(pred, K.count, 0u, K.segcap, K.overflow);
  return at == RT_NONE ? RT_NONE : K.base + at;
}
// divergent call sites (the per-lane instance-BVH walk): one atomic per candidate
RT_DEV uint32_t cand_alloc(const CandSink& K) {
  uint32_t pos = rt_atomic_add_u32(K.count, 1u);
  if (pos >= K.segcap) { *K.overflow = 1u; return RT_NONE; }
  return K.base + pos;
}
RT_DEV void cand_write(const CandSink& K, uint32_t ci, V3 p, V3 d, uint32_t minst_k, uint32_t next) {
  K.rec[2 * (size_t)ci] = make_float4(p.x, p.y, p.z, __uint_as_float(minst_k));
  K.rec[2 * (size_t)ci + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(next));
}

// hitAABB, objects.cpp:94-120, returning also the entry parameter.  The ternaries keep FAST_MIN/FAST_MAX's
// behaviour for NaN operands (0 * inf when a ray lies in a slab plane).
#define RT_FMIN(a, b) ((a) < (b) ? (a) : (b))
#define RT_FMAX(a, b) ((a) > (b) ? (a) : (b))
RT_DEV bool box_test(V3 p, V3 inv, float b0, float b1, float b2, float b3, float b4, float b5, float& tmin_out) {
  float t1 = (b0 - p.x) * inv.x;
  float t2 = (b3 - p.x) * inv.x;
  float t3 = (b1 - p.y) * inv.y;
  float t4 = (b4 - p.y) * inv.y;
  float t5 = (b2 - p.z) * inv.z;
  float t6 = (b5 - p.z) * inv.z;
  float tminX = RT_FMIN(t1, t2), tmaxX = RT_FMAX(t1, t2);
  float tminY = RT_FMIN(t3, t4), tmaxY = RT_FMAX(t3, t4);
  float tminZ = RT_FMIN(t5, t6), tmaxZ = RT_FMAX(t5, t6);
  float tmin = RT_FMAX(RT_FMAX(tminX, tminY), tminZ);
  float tmax = RT_FMIN(RT_FMIN(tmaxX, tmaxY), tmaxZ);
  tmin_out = tmin;
  return tmax >= tmin && tmax >= 0.0f;
}

// Same test with v_min/v_max (min3/max3).  Bit-identical decisions whenever no slab product is NaN, i.e. whenever
// every component of inv is finite (NaN needs 0 * inf: a zero direction component AND an origin exactly on that slab
// plane); rays with a non-finite inv component take box_test above.  (-0 vs +0 results cannot change a comparison.)
RT_DEV bool box_test_fast(V3 p, V3 inv, float b0, float b1, float b2, float b3, float b4, float b5, float& tmin_out) {
  float t1 = (b0 - p.x) * inv.x;
  float t2 = (b3 - p.x) * inv.x;
  float t3 = (b1 - p.y) * inv.y;
  float t4 = (b4 - p.y) * inv.y;
  float t5 = (b2 - p.z) * inv.z;
  float t6 = (b5 - p.z) * inv.z;
  float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
  float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
  tmin_out = tmin;
  return tmax >= tmin && tmax >= 0.0f;
}
// Both children of a 64-byte node (layout per axis: left min, right min, left max, right max): the same arithmetic as
// box_test_fast per child, written pairwise so that it compiles to packed FP32 operations without register shuffles.
RT_DEV void box_test_fast2(V3 p, V3 inv, float4 nx, float4 ny, float4 nz, bool& hl, bool& hr, float& tl, float& tr) {
  float t1l = (nx.x - p.x) * inv.x, t1r = (nx.y - p.x) * inv.x;
  float t2l = (nx.z - p.x) * inv.x, t2r = (nx.w - p.x) * inv.x;
  float t3l = (ny.x - p.y) * inv.y, t3r = (ny.y - p.y) * inv.y;
  float t4l = (ny.z - p.y) * inv.y, t4r = (ny.w - p.y) * inv.y;
  float t5l = (nz.x - p.z) * inv.z, t5r = (nz.y - p.z) * inv.z;
  float t6l = (nz.z - p.z) * inv.z, t6r = (nz.w - p.z) * inv.z;
  float tminl = fmaxf(fmaxf(fminf(t1l, t2l), fminf(t3l, t4l)), fminf(t5l, t6l));
  float tminr = fmaxf(fmaxf(fminf(t1r, t2r), fminf(t3r, t4r)), fminf(t5r, t6r));
  float tmaxl = fminf(fminf(fmaxf(t1l, t2l), fmaxf(t3l, t4l)), fmaxf(t5l, t6l));
  float tmaxr = fminf(fminf(fmaxf(t1r, t2r), fmaxf(t3r, t4r)), fmaxf(t5r, t6r));
  tl = tminl; tr = tminr;
  hl = tmaxl >= tminl && tmaxl >= 0.0f;
  hr = tmaxr >= tminr && tmaxr >= 0.0f;
}
// The walk's form of the same two tests with the pruning bound folded in: a child is entered iff
// tmax >= tmin && tmax >= 0 && tmin <= bound, i.e. iff min(tmax, bound) >= max(tmin, 0) (bound >= 0).  `tl`, `tr` return
// max(tmin, 0), which orders the children exactly as tmin does except that two boxes the origin lies in tie.
RT_DEV void box_test_walk2(V3 p, V3 inv, float4 nx, float4 ny, float4 nz, float bound, bool& hl, bool& hr, float& tl, float& tr) {
  float t1l = (nx.x - p.x) * inv.x, t1r = (nx.y - p.x) * inv.x;
  float t2l = (nx.z - p.x) * inv.x, t2r = (nx.w - p.x) * inv.x;
  float t3l = (ny.x - p.y) * inv.y, t3r = (ny.y - p.y) * inv.y;
  float t4l = (ny.z - p.y) * inv.y, t4r = (ny.w - p.y) * inv.y;
  float t5l = (nz.x - p.z) * inv.z, t5r = (nz.y - p.z) * inv.z;
  float t6l = (nz.z - p.z) * inv.z, t6r = (nz.w - p.z) * inv.z;
  float tminl = fmaxf(fmaxf(fmaxf(fminf(t1l, t2l), fminf(t3l, t4l)), fminf(t5l, t6l)), 0.0f);
  float tminr = fmaxf(fmaxf(fmaxf(fminf(t1r, t2r), fminf(t3r, t4r)), fminf(t5r, t6r)), 0.0f);
  float tmaxl = fminf(fminf(fminf(fmaxf(t1l, t2l), fmaxf(t3l, t4l)), fmaxf(t5l, t6l)), bound);
  float tmaxr = fminf(fminf(fminf(fmaxf(t1r, t2r), fmaxf(t3r, t4r)), fmaxf(t5r, t6r)), bound);
  tl = tminl; tr = tminr;
  hl = tmaxl >= tminl;
  hr = tmaxr >= tminr;
}
template <bool FAST>
RT_DEV bool box_test_sel(V3 p, V3 inv, float b0, float b1, float b2, float b3, float b4, float b5, float& tmin_out) {
  if (FAST) return box_test_fast(p, inv, b0, b1, b2, b3, b4, b5, tmin_out);
  return box_test(p, inv, b0, b1, b2, b3, b4, b5, tmin_out);
}
RT_DEV bool inv_is_finite(V3 inv) {
#ifdef RT_NO_FASTBOX
  return false;
#endif
  return fabsf(inv.x) <= RT_BIG && fabsf(inv.y) <= RT_BIG && fabsf(inv.z) <= RT_BIG;
}

// Möller–Trumbore with precomputed edges — TriObj::IntersectTriangle, objects.cpp:193-231.
// (v0, e1 = v1 - v0, e2 = v2 - v0) are the same float values the reference computes per test.
RT_DEV bool tri_test(V3 p, V3 d, V3 v0, V3 e1, V3 e2, float& t, float& u, float& v) {
  const float epsilon = 0.002f;
  V3 h = cross(d, e2);
  float det = dot(e1, h);
  if (fabsf(det) < epsilon) return false;
  float inv_det = 1.0f / det;
  V3 s = p - v0;
  float uu = inv_det * dot(s, h);
  if (uu < 0.0f || uu > 1.0f) return false;
  V3 q = cross(s, e1);
  float vv = inv_det * dot(d, q);
  if (vv < 0.0f || (uu + vv) > 1.0f) return false;
  float tt = inv_det * dot(e2, q);
  t = tt; u = uu; v = vv;
  return true;
}

// Pruning slack: a BVH box is skipped only when its entry parameter exceeds the best hit by more than the
// rounding the slab test and Möller–Trumbore can disagree by; the reference prunes nothing (objects.cpp:268-339),
// so skipping must never hide a triangle that would have won.
RT_DEV float prune_bound(float t) { return t + (fabsf(t) * 1e-4f + 1e-4f); }

struct MeshHit {
  float t, u, v;
  uint32_t key;   // leaf order * 4 + (3 - slot in leaf): reproduces the reference's tie-breaks
  uint32_t face;
};

// Closest hit inside one mesh.  Equivalent to TriObj::TraceBVHNode (objects.cpp:268-339): the result is the
// minimum-t triangle among all leaves whose chain of boxes the ray's line hits; on equal t the later leaf in
// depth-first order wins (child 2 beats child 1, :331-337) and inside a leaf the earlier triangle wins (:294).
// Traversal is ordered and pruned against `t_bound` (hits at or beyond the current global best cannot win:
// the scene-graph compare is a strict <, RayTracer.cpp:387,404).
template <bool CNT, bool FAST>
RT_DEV bool mesh_closest_t(const DScene& S, const DMesh& M, V3 p, V3 d, V3 inv, float t_bound, MeshHit& out,
                           uint32_t* stk, uint32_t stride, Cnt<CNT>& cnt) {
  float tm;
  cnt.node();
  if (!box_test_sel<FAST>(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm)) return false;
  const float4* nodes = S.bvh + M.node_base;
  const float4* tris = S.tris + M.tri_base;
  float best_t = RT_BIG;
  uint32_t best_key = 0, best_face = 0;
  float best_u = 0, best_v = 0;
  bool found = false;
  uint32_t cur = M.root_ref;
  int sp = 0;
  for (;;) {
    if (cur & RT_LEAF) {
      uint32_t count = cur & 7u, first = (cur & 0x7fffffffu) >> 3;
      for (uint32_t s = 0; s < count; s++) {
        float4 a = tris[3 * (first + s)], b = tris[3 * (first + s) + 1], c = tris[3 * (first + s) + 2];
        float t, u, v;
        cnt.tri();
        if (tri_test(p, d, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v)) {
          if (t > 0.002f) {
            uint32_t key = __float_as_uint(a.w);
            if (t < best_t || (t == best_t && found && key > best_key)) {
              best_t = t; best_key = key; best_face = __float_as_uint(b.w); best_u = u; best_v = v; found = true;
            }
          }
        }
      }
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
      continue;
    }
    float4 n0 = nodes[4 * cur], n1 = nodes[4 * cur + 1], n2 = nodes[4 * cur + 2], n3 = nodes[4 * cur + 3];
    float tl, tr;
    cnt.node(2);
    bool hl = box_test_sel<FAST>(p, inv, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, tl);
    bool hr = box_test_sel<FAST>(p, inv, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, tr);
    float bound = prune_bound(best_t < t_bound ? best_t : t_bound);
    hl = hl && tl <= bound;
    hr = hr && tr <= bound;
    uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
    if (hl && hr) {
      bool left_first = tl <= tr;
      uint32_t nearc = left_first ? cl : cr, farc = left_first ? cr : cl;
      stk[(sp++) * stride] = farc;
      cur = nearc;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
    }
  }
  if (!found) return false;
  out.t = best_t; out.u = best_u; out.v = best_v; out.key = best_key; out.face = best_face;
  return true;
}

template <bool CNT>
RT_DEV bool mesh_closest(const DScene& S, const DMesh& M, V3 p, V3 d, V3 inv, float t_bound, MeshHit& out,
                         uint32_t* stk, uint32_t stride, Cnt<CNT>& cnt) {
  if (inv_is_finite(inv)) return mesh_closest_t<CNT, true>(S, M, p, d, inv, t_bound, out, stk, stride, cnt);
  return mesh_closest_t<CNT, false>(S, M, p, d, inv, t_bound, out, stk, stride, cnt);
}

// TriObj::TraceBVHNodeShadow (objects.cpp:341-371): true iff some triangle of a visited leaf has t in (0.002, t_max).
template <bool CNT, bool FAST>
RT_DEV bool mesh_any_t(const DScene& S, const DMesh& M, V3 p, V3 d, V3 inv, float t_max, uint32_t* stk, uint32_t stride,
                       Cnt<CNT>& cnt) {
  float tm;
  cnt.node();
  if (!box_test_sel<FAST>(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm)) return false;
  const float4* nodes = S.bvh + M.node_base;
  const float4* tris = S.tris + M.tri_base;
  float bound = prune_bound(t_max);
  uint32_t cur = M.root_ref;
  int sp = 0;
  for (;;) {
    if (cur & RT_LEAF) {
      uint32_t count = cur & 7u, first = (cur & 0x7fffffffu) >> 3;
      for (uint32_t s = 0; s < count; s++) {
        float4 a = tris[3 * (first + s)], b = tris[3 * (first + s) + 1], c = tris[3 * (first + s) + 2];
        float t, u, v;
        cnt.tri();
        if (tri_test(p, d, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v))
          if (t > 0.002f && t < t_max) return true;
      }
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
      continue;
    }
    float4 n0 = nodes[4 * cur], n1 = nodes[4 * cur + 1], n2 = nodes[4 * cur + 2], n3 = nodes[4 * cur + 3];
    float tl, tr;
    cnt.node(2);
    bool hl = box_test_sel<FAST>(p, inv, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, tl) && tl <= bound;
    bool hr = box_test_sel<FAST>(p, inv, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, tr) && tr <= bound;
    uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
    if (hl && hr) {
      stk[(sp++) * stride] = cr;
      cur = cl;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
    }
  }
  return false;
}

template <bool CNT>
RT_DEV bool mesh_any(const DScene& S, const DMesh& M, V3 p, V3 d, V3 inv, float t_max, uint32_t* stk, uint32_t stride,
                     Cnt<CNT>& cnt) {
  if (inv_is_finite(inv)) return mesh_any_t<CNT, true>(S, M, p, d, inv, t_max, stk, stride, cnt);
  return mesh_any_t<CNT, false>(S, M, p, d, inv, t_max, stk, stride, cnt);
}

// PointLight::IntersectRay, objects.cpp:377-413 (int radius, double arithmetic)
template <class LP>
RT_DEV bool light_sphere(LP Lp, V3 p, V3 d, float zbest, float& t_out, bool& front) {
  int r = (int)Lp->size;
  V3 oc = p - v3(Lp->position[0], Lp->position[1], Lp->position[2]);
  float eps = 0.002f;
  double a = (double)dot(d, d);
  double b = 2.0 * (double)dot(oc, d);
  double c = (double)(dot(oc, oc) - (float)(r * r));
  double disc = b * b - 4 * a * c;
  if (disc < 0) return false;
  float sq = sqrtf((float)disc);
  float twoA = (float)(2 * a);
  double t1 = (-b - (double)sq) / (double)twoA;
  double t2 = (-b + (double)sq) / (double)twoA;
  // hInfo.z of the fresh HitInfo is BIGFLOAT, so `hInfo.z > t` holds for every finite t
  if (t1 > (double)eps) {
    if ((double)RT_BIG > t1) { t_out = (float)t1; front = true; return true; }
  } else if (t2 >= (double)eps) {
    if ((double)RT_BIG > t2) { t_out = (float)t2; front = false; return true; }
  }
  return false;
}

// ------------------------------------------------------------------ scene traversal

struct Hit {            // HitInfo (scene.h:74-90) in world space before SetHit
  V3 p;
  float z;
  V3 N;
  V3 uvw;
  int32_t inst;         // instance index, -1 none
  int32_t face;
  int32_t light;        // -1 or light index
  bool front;
};

// Walks the instance table in depth-first order keeping the ray transformed to the CURRENT PARENT's space: siblings
// share it, so each instance costs one Node::ToNodeCoords (RayTracer.cpp:377) instead of one per chain level.  The
// arithmetic per level is the reference's (same matrices, same order), only repeated work is dropped.  Everything
// here is wave-uniform control flow over scalar-loaded table entries.
struct ParentRay {
  int key;
  V3 p, d;
};
template <class IP, class XP>
RT_DEV void ray_to_instance(IP I, XP xf, V3 o, V3 dir, ParentRay& pr, V3& p, V3& d) {
  int depth = I->depth, off = I->xf_off, pk = I->parent_key;
#ifdef RT_NO_PARENT_CACHE
  pr.key = -2;
#endif
  if (pk != pr.key) {
    V3 cp = o, cd = dir;
    for (int k = 0; k + 1 < depth; k++) {
      XP m = xf + off + 24 * k;
      V3 np = xf_point(m, cp);
      V3 nd = xf_dir(m, cd);
      cp = np; cd = nd;
    }
    pr.key = pk; pr.p = cp; pr.d = cd;
  }
  XP m = xf + off + 24 * (depth - 1);
  p = xf_point(m, pr.p);
  d = xf_dir(m, pr.d);
}

// Closest hit so far: ray parameter, instance, side, and for meshes the triangle.
struct Best {
  float t;
  int inst;
  bool front;
  MeshHit mh;
};
RT_DEV void best_init(Best& B) {
  B.t = RT_BIG; B.inst = -1; B.front = true;
  B.mh.t = 0; B.mh.u = 0; B.mh.v = 0; B.mh.key = 0; B.mh.face = 0;
}

// Renderer::TraceRay -> RayTracer::TraverseTree (RayTracer.cpp:373-435): closest hit over every instanced
// primitive in depth-first node order with strict '<'.  SPLIT = true is the first phase of the two-phase trace
// kernels: spheres and planes are tested, mesh instances are only culled against their world box and reported
// through `mesh_flag` (their BVHs are walked by the second phase, k_trace_mesh).
template <bool CNT, bool SPLIT, bool CULL = false>   // CULL: the wave's rays are neighbours (camera rays), see below
RT_DEV void closest_instances(const DScene& S, V3 o, V3 dir, int hitSide, Best& B, uint32_t& cand_head, const CandSink& K, uint32_t* stk,
                              uint32_t stride, Cnt<CNT>& cnt, bool active = true) {
  V3 winv = v3(0, 0, 0);
  if (CULL || S.n_minst > 0) winv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // for the world boxes
  RT_UTYPE(DInstance)* inst = RT_UNIFORM_PTR(DInstance, S.inst);
  RT_UTYPE(DMesh)* meshes = RT_UNIFORM_PTR(DMesh, S.meshes);
  RT_UTYPE(float)* xf = RT_UNIFORM_PTR(float, S.xf);
  ParentRay pr;
  pr.key = -2; pr.p = o; pr.d = dir;
  cand_head = RT_NONE;
  for (int i = 0; i < S.n_inst; i++) {
    RT_UTYPE(DInstance)* I = inst + i;
    int prim = I->prim;
    V3 p = o, d = dir;
    if (CULL && SPLIT && prim != 3) {
      // Camera rays (k_raygen_trace): spheres and planes, too, are only transformed into and intersected when some ray of
      // the wave can reach them — it has to enter the instance's padded world box, and not beyond its best hit so far (a
      // hit inside the box is no nearer than the box).  The reference tests every node for every ray
      // (RayTracer.cpp:373-435); the result is the same.  The loop is wave-uniform, so the skip is taken when NO lane
      // can reach the instance: about 15 instructions on a scalar-loaded box instead of the ~85 of transform +
      // intersection, and the camera rays of an 8x8-pixel tile miss most instances together (k_raygen_trace 1.93 ->
      // 1.74 ms per C2 step).  Secondary rays do not: for them the test costs more than it saves (k_trace_closest
      // 0.55 -> 0.62 ms on C2, 3.5 -> 4.6 ms on C3), so they go without.
      float tm;
      const bool reach = active && box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm) &&
                         !(tm > prune_bound(B.t));
      if (!wave_any(reach)) continue;
    }
    if (prim == 3) {  // world-bounds cull (the reference has none; a ray that misses the padded world box cannot
      float tm;       // hit the mesh, so the result is unchanged)
      bool want = box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm);
      if (SPLIT) {
        // first phase: the instance becomes a candidate of the second phase if the ray (a lane that has one) enters its
        // world box not beyond the best hit so far and hits the mesh's root box in the instance's space — the tests
        // TriObj::IntersectRay starts with (objects.cpp:172-183, 268-275), made here where the matrices are scalar loads
        want = want && active && !(tm > prune_bound(B.t));
        if (want) {
          ray_to_instance(I, xf, o, dir, pr, p, d);
          V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          RT_UTYPE(DMesh)* M = meshes + I->mesh;
          cnt.node();
          want = inv_is_finite(inv) ? box_test_fast(p, inv, M->bounds[0], M->bounds[1], M->bounds[2], M->bounds[3], M->bounds[4], M->bounds[5], tm)
                                    : box_test(p, inv, M->bounds[0], M->bounds[1], M->bounds[2], M->bounds[3], M->bounds[4], M->bounds[5], tm);
        }
        uint32_t ci = cand_push(K, want);   // (every lane of the wave: the loop and this branch are wave-uniform)
        if (ci != RT_NONE) { cand_write(K, ci, p, d, (uint32_t)I->minst_k, cand_head); cand_head = ci; }
        continue;
      }
      if (!want) continue;
    }
    ray_to_instance(I, xf, o, dir, pr, p, d);
    cnt.inst();
    float t;
    bool front;
    if (prim == 1) {
      cnt.sphere();
      if (sphere_closest(p, d, hitSide, t, front) && t < B.t) { B.t = t; B.inst = i; B.front = front; }
    } else if (prim == 2) {
      cnt.plane();
      if (plane_closest(p, d, hitSide, t, front) && t < B.t) { B.t = t; B.inst = i; B.front = front; }
    } else if (prim == 3) {
      V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
      MeshHit mh;
      if (mesh_closest<CNT>(S, S.meshes[I->mesh], p, d, inv, B.t, mh, stk, stride, cnt) && mh.t < B.t) {
        B.t = mh.t; B.inst = i; B.mh = mh;
      }
    }
  }
}

// ---- instance BVH (TLAS).  RayTracer::TraverseTree tests every node of the scene graph for every ray
// (RayTracer.cpp:373-435, Theta(nodes) per ray: 2 945 sphere tests per ray in scenes/volumetric_spiral.xml).  An
// instance whose padded world box the ray misses cannot be hit, so walking a BVH over those boxes (ordered, pruned
// against the best hit like the mesh walk) leaves the result unchanged; TraverseTree's "first in depth-first order wins
// an equal t" becomes an explicit compare on the instance index because the walk visits instances in spatial order.
template <class IP>
RT_DEV void ray_to_instance_full(IP I, const float* xfbase, V3 o, V3 dir, V3& p, V3& d) {
  const float* chain = xfbase + I->xf_off;
  p = o; d = dir;
  for (int k = 0; k < I->depth; k++) {
    V3 np = xf_point(chain + 24 * k, p);
    V3 nd = xf_dir(chain + 24 * k, d);
    p = np; d = nd;
  }
}

template <bool CNT, bool SPLIT>
RT_DEV void closest_instances_tlas(const DScene& S, V3 o, V3 dir, int hitSide, Best& B, uint32_t& cand_head, const CandSink& K, uint32_t* stk,
                                   uint32_t stride, Cnt<CNT>& cnt) {
  V3 winv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  cand_head = RT_NONE;
  uint32_t cur = S.tlas_root;
  int sp = 0;
  for (;;) {
    if (cur & RT_LEAF) {
      uint32_t count = cur & 7u, first = (cur & 0x7fffffffu) >> 3;
      for (uint32_t s = 0; s < count; s++) {
        int i = S.tlas_idx[first + s];
        const DInstance* I = S.inst + i;
        int prim = I->prim;
        float tm;
        if (!box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm)) continue;
        V3 p, d;
        ray_to_instance_full(I, S.xf, o, dir, p, d);
        if (prim == 3 && SPLIT) {   // a candidate of the second phase (see closest_instances); this walk is per lane
          if (tm > prune_bound(B.t)) continue;
          V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          const DMesh& M = S.meshes[I->mesh];
          cnt.node();
          bool hit = inv_is_finite(inv) ? box_test_fast(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm)
                                        : box_test(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm);
          if (!hit) continue;
          uint32_t ci = cand_alloc(K);
          if (ci != RT_NONE) { cand_write(K, ci, p, d, (uint32_t)I->minst_k, cand_head); cand_head = ci; }
          continue;
        }
        cnt.inst();
        float t;
        bool front;
        if (prim == 1) {
          cnt.sphere();
          if (sphere_closest(p, d, hitSide, t, front) && (t < B.t || (t == B.t && i < B.inst))) { B.t = t; B.inst = i; B.front = front; }
        } else if (prim == 2) {
          cnt.plane();
          if (plane_closest(p, d, hitSide, t, front) && (t < B.t || (t == B.t && i < B.inst))) { B.t = t; B.inst = i; B.front = front; }
        } else if (prim == 3) {
          V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          MeshHit mh;
          // stack entries above sp belong to the mesh walk; t_bound = B.t keeps every hit that can still tie
          if (mesh_closest<CNT>(S, S.meshes[I->mesh], p, d, inv, B.t, mh, stk + sp * stride, stride, cnt) &&
              (mh.t < B.t || (mh.t == B.t && i < B.inst))) {
            B.t = mh.t; B.inst = i; B.mh = mh;
          }
        }
      }
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
      continue;
    }
    float4 n0 = S.tlas[4 * cur], n1 = S.tlas[4 * cur + 1], n2 = S.tlas[4 * cur + 2], n3 = S.tlas[4 * cur + 3];
    float tl, tr;
    cnt.node(2);
    bool hl = box_test(o, winv, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, tl);
    bool hr = box_test(o, winv, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, tr);
    float bound = prune_bound(B.t);
    hl = hl && tl <= bound;
    hr = hr && tr <= bound;
    uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
    if (hl && hr) {
      bool left_first = tl <= tr;
      stk[(sp++) * stride] = left_first ? cr : cl;
      cur = left_first ? cl : cr;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
    }
  }
}

template <bool CNT, bool SPLIT>
RT_DEV bool any_instances_tlas(const DScene& S, V3 o, V3 dir, float t_max, uint32_t& cand_head, const CandSink& K, uint32_t* stk, uint32_t stride,
                               Cnt<CNT>& cnt) {
  V3 winv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  cand_head = RT_NONE;
  const float bound = prune_bound(t_max);
  uint32_t cur = S.tlas_root;
  int sp = 0;
  for (;;) {
    if (cur & RT_LEAF) {
      uint32_t count = cur & 7u, first = (cur & 0x7fffffffu) >> 3;
      for (uint32_t s = 0; s < count; s++) {
        int i = S.tlas_idx[first + s];
        const DInstance* I = S.inst + i;
        int prim = I->prim;
        float tm;
        if (!box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm)) continue;
        if (prim == 3 && tm > prune_bound(t_max)) continue;  // the box begins beyond the light
        V3 p, d;
        ray_to_instance_full(I, S.xf, o, dir, p, d);
        if (prim == 3 && SPLIT) {
          V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          const DMesh& M = S.meshes[I->mesh];
          cnt.node();
          bool hit = inv_is_finite(inv) ? box_test_fast(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm)
                                        : box_test(p, inv, M.bounds[0], M.bounds[1], M.bounds[2], M.bounds[3], M.bounds[4], M.bounds[5], tm);
          if (!hit) continue;
          uint32_t ci = cand_alloc(K);
          if (ci != RT_NONE) { cand_write(K, ci, p, d, (uint32_t)I->minst_k, cand_head); cand_head = ci; }
          continue;
        }
        cnt.inst();
        bool occ = false;
        if (prim == 1) {
          cnt.sphere();
          occ = sphere_any(p, d, t_max);
        } else if (prim == 2) {
          cnt.plane();
          occ = plane_any(p, d, t_max);
        } else if (prim == 3) {
          V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          occ = mesh_any<CNT>(S, S.meshes[I->mesh], p, d, inv, t_max, stk + sp * stride, stride, cnt);
        }
        if (occ) return true;
      }
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
      continue;
    }
    float4 n0 = S.tlas[4 * cur], n1 = S.tlas[4 * cur + 1], n2 = S.tlas[4 * cur + 2], n3 = S.tlas[4 * cur + 3];
    float tl, tr;
    cnt.node(2);
    bool hl = box_test(o, winv, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, tl) && tl <= bound;
    bool hr = box_test(o, winv, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, tr) && tr <= bound;
    uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
    if (hl && hr) {
      stk[(sp++) * stride] = cr;
      cur = cl;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
    }
  }
  return false;
}

// Hit attributes of the winner (objects.cpp:37-43, 135-138, 303-320), Node::FromNodeCoords back up the chain
// (scene.h:184-189), then the renderable lights (RayTracer.cpp:413-432).
// Texture coordinates are part of the hit the batched queries hand back; k_shade only reads them through a texture map
// (a sphere's cost atan2f, asinf and two double divisions, a mesh's two more 16-byte fetches per hit).
RT_DEV bool wants_uvw(const DInstance&) { return true; }
RT_DEV bool wants_uvw(const DShadeInst& I) { return I.uses_uvw != 0; }
template <class InstT>   // DInstance, or k_shade's DShadeInst (same field names)
RT_DEV bool finalize_hit_from(const DScene& S, const InstT* table, V3 o, V3 dir, const Best& B, Hit& hit) {
  const float best_t = B.t;
  const int best_inst = B.inst;
  hit.inst = -1; hit.face = -1; hit.light = -1; hit.front = true; hit.z = RT_BIG;
  hit.p = v3(0, 0, 0); hit.N = v3(0, 0, 0); hit.uvw = v3(0.5f, 0.5f, 0.5f);
  bool any = false;
  if (best_inst >= 0) {
    any = true;
    const InstT& I = table[best_inst];  // per-lane index: ordinary loads
    const float* chain = S.xf + I.xf_off;
    V3 p = o, d = dir;
    for (int k = 0; k < I.depth; k++) {
      V3 np = xf_point(chain + 24 * k, p);
      V3 nd = xf_dir(chain + 24 * k, d);
      p = np; d = nd;
    }
    V3 hp, hn, uvw;
    bool front = B.front;
    if (I.prim == 1) {  // objects.cpp:37-43
      hp = p + (d * best_t);
      hn = normalized(hp);
      uvw = v3(0.5f, 0.5f, 0.0f);
      if (wants_uvw(I)) {
        float tu = (float)(((double)atan2f(hp.y, hp.x) / (2 * RT_PI_D)) + .5);
        float tv = (float)((double)asinf(hp.z) / RT_PI_D + .5);
        uvw = v3(tu, tv, 0.0f);
      }
    } else if (I.prim == 2) {  // objects.cpp:135-138
      hp = p + (d * best_t);
      hn = v3(0.0f, 0.0f, 1.0f);
      uvw = v3((hp.x + 1.0f) * 0.5f, (hp.y + 1.0f) * 0.5f, 0.0f);
    } else {  // objects.cpp:303-320
      const float4* fr = S.shade + 5 * ((size_t)I.shade_base + B.mh.face);
      const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2];
      float u = B.mh.u, v = B.mh.v, w = 1.0f - u - v;
      uvw = v3(0, 0, 0);
      if (I.has_uvw && wants_uvw(I)) {
        const float4 f3 = fr[3], f4 = fr[4];
        uvw = (v3(f0.w, f1.w, f2.w) * w) + (v3(f3.x, f3.y, f3.z) * u) + (v3(f3.w, f4.x, f4.y) * v);
      }
      hn = normalized(w * v3(f0.x, f0.y, f0.z) + u * v3(f1.x, f1.y, f1.z) + v * v3(f2.x, f2.y, f2.z));
      hp = p + d * best_t;
      front = dot(d, hn) < 0.0f;
      hit.face = (int32_t)B.mh.face;
    }
    for (int k = I.depth - 1; k >= 0; k--) {  // Node::FromNodeCoords back up the chain (scene.h:184-189)
      hp = xf_point(chain + 24 * k + 12, hp);
      hn = xf_tmul(chain + 24 * k, hn);
    }
    hit.p = hp; hit.N = hn; hit.uvw = uvw; hit.z = best_t; hit.inst = best_inst; hit.front = front;
  }
  RT_UTYPE(DLight)* lights = RT_UNIFORM_PTR(DLight, S.lights);
  for (int l = 0; l < S.n_lights; l++) {  // RayTracer.cpp:413-432
    RT_UTYPE(DLight)* L = lights + l;
    if (L->kind == 3 && L->size > 0.0f) {
      float t;
      bool front;
      if (light_sphere(L, o, dir, RT_BIG, t, front)) {
        if (t < hit.z) {
          hit.z = t;
          hit.p = o + (dir * t);
          hit.N = v3(0, 0, 0);
          hit.uvw = v3(0.5f, 0.5f, 0.5f);
          hit.inst = -1; hit.face = -1; hit.light = l; hit.front = front;
          any = true;
        }
      }
    }
  }
  return any;
}

RT_DEV bool finalize_hit(const DScene& S, V3 o, V3 dir, const Best& B, Hit& hit) { return finalize_hit_from(S, S.inst, o, dir, B, hit); }

template <bool CNT>
RT_DEV bool scene_closest(const DScene& S, V3 o, V3 dir, int hitSide, Hit& hit, uint32_t* stk, uint32_t stride, Cnt<CNT>& cnt) {
  Best B;
  best_init(B);
  uint32_t head;
  CandSink none;
  none.rec = nullptr; none.count = nullptr; none.base = 0; none.segcap = 0; none.overflow = nullptr;
  if (S.use_tlas) closest_instances_tlas<CNT, false>(S, o, dir, hitSide, B, head, none, stk, stride, cnt);
  else closest_instances<CNT, false>(S, o, dir, hitSide, B, head, none, stk, stride, cnt);
  return finalize_hit(S, o, dir, B, hit);
}

// Renderer::TraceShadowRay -> TraverseTreeShadow (RayTracer.cpp:346-371): any occluder with t < t_max.
// SPLIT = true: first phase of the two-phase shadow kernels (see closest_instances).
// `active`: this lane has a ray (the first-phase kernels call with every lane of the wave, see the wave-wide cull below).
// MASKED: `inst_mask` names the instances (bit i: table entry i < 64; entries from 64 on always count) that can occlude this
// lane's ray at all — what cone_clear could not exclude for the ray's soft-shadow group; the wave tests the union.
template <bool CNT, bool SPLIT, bool MASKED = false>
RT_DEV bool any_instances(const DScene& S, V3 o, V3 dir, float t_max, uint32_t& cand_head, const CandSink& K, uint32_t* stk, uint32_t stride,
                          Cnt<CNT>& cnt, bool active = true, unsigned long long inst_mask = ~0ull) {
  const unsigned long long wmask = MASKED ? wave_or64(active ? inst_mask : 0ull) : ~0ull;
  V3 winv = v3(0, 0, 0);
  if (S.n_minst > 0) winv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);  // (only the mesh world boxes need it)
  RT_UTYPE(DInstance)* inst = RT_UNIFORM_PTR(DInstance, S.inst);
  RT_UTYPE(float)* xf = RT_UNIFORM_PTR(float, S.xf);
  RT_UTYPE(DMesh)* meshes = RT_UNIFORM_PTR(DMesh, S.meshes);
  ParentRay pr;
  pr.key = -2; pr.p = o; pr.d = dir;
  bool occluded = false;
  cand_head = RT_NONE;
  // Box around the segments the WAVE's rays can be occluded on, [o, o + dir * t_max] (the rays of a wave are neighbours:
  // samples of the same few shadow groups).  An instance whose padded world box does not overlap it cannot occlude any
  // of them (an occluding point lies on a segment and inside the instance), so the whole wave skips its transform and
  // test: six compares on wave-uniform values against the scalar-loaded box instead of ~100 instructions.  The reference
  // tests every node for every ray (RayTracer.cpp:346-371); the result is the same.  (A per-lane skip does not pay: the
  // wave still runs the test whenever one lane needs it.)
  V3 smin = v3(-RT_BIG, -RT_BIG, -RT_BIG), smax = v3(RT_BIG, RT_BIG, RT_BIG);
  // (with a mask that covers the whole table the mask is the cull: it is per group and exact about shapes, the box is
  // per wave and about boxes)
  const bool box_cull = SPLIT && !(MASKED && S.n_inst <= 64 && wmask != ~0ull);   // [wave-uniform]
  if (box_cull) {
    const float tseg = t_max < 1e30f ? t_max : 1e30f;
    const V3 e = o + dir * tseg;
    smin = v3(wave_min(active ? fminf(o.x, e.x) : RT_BIG), wave_min(active ? fminf(o.y, e.y) : RT_BIG), wave_min(active ? fminf(o.z, e.z) : RT_BIG));
    smax = v3(wave_max(active ? fmaxf(o.x, e.x) : -RT_BIG), wave_max(active ? fmaxf(o.y, e.y) : -RT_BIG), wave_max(active ? fmaxf(o.z, e.z) : -RT_BIG));
  }
#ifndef RT_EMU
  // The cull of up to 64 table entries at once: lane l compares entry l's box with the wave's segment box, and the loop
  // below runs over the survivors only (ascending, as before).  Entry by entry, every culled instance cost a scalar load
  // and its latency; shadow rays cull most of the table.
  unsigned long long live = ~0ull;
  int live_base = -64;
#endif
  for (int i = 0; i < S.n_inst; i++) {
#ifndef RT_EMU
    if (SPLIT) {
      if (i >= live_base + 64) {
        live_base = i;
        const int li = i + (int)(threadIdx.x & 63u);
        bool keep = false;
        if (li < S.n_inst) {
          const DInstance& J = S.inst[li];
          keep = !(smin.x > J.wmax[0] || smax.x < J.wmin[0] || smin.y > J.wmax[1] || smax.y < J.wmin[1] || smin.z > J.wmax[2] ||
                   smax.z < J.wmin[2]);
          if (MASKED && li < 64) keep = keep && ((wmask >> li) & 1ull);
        }
        live = __ballot(keep);
      }
      unsigned long long rest = live >> (unsigned)(i - live_base);
      if (rest == 0ull) { i = live_base + 63; continue; }       // nothing left in this group of 64  [wave-uniform]
      i += __ffsll((long long)rest) - 1;                         // the next survivor
    }
#endif
    RT_UTYPE(DInstance)* I = inst + i;
#ifdef RT_EMU
    if (SPLIT && (smin.x > I->wmax[0] || smax.x < I->wmin[0] || smin.y > I->wmax[1] || smax.y < I->wmin[1] || smin.z > I->wmax[2] ||
                  smax.z < I->wmin[2]))
      continue;
    if (MASKED && i < 64 && !((wmask >> i) & 1ull)) continue;
#endif
    int prim = I->prim;
    V3 p, d;
    ray_to_instance(I, xf, o, dir, pr, p, d);
    if (prim == 3 && SPLIT) {   // [wave-uniform branch] a candidate of the second phase (see closest_instances)
      float tm;
      bool want = active && !occluded && box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm);
      want = want && !(tm > prune_bound(t_max));   // the box begins beyond the light: nothing in it can occlude
      if (want) {
        V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        RT_UTYPE(DMesh)* M = meshes + I->mesh;
        cnt.node();
        want = inv_is_finite(inv) ? box_test_fast(p, inv, M->bounds[0], M->bounds[1], M->bounds[2], M->bounds[3], M->bounds[4], M->bounds[5], tm)
                                  : box_test(p, inv, M->bounds[0], M->bounds[1], M->bounds[2], M->bounds[3], M->bounds[4], M->bounds[5], tm);
      }
      uint32_t ci = cand_push(K, want);
      if (ci != RT_NONE) { cand_write(K, ci, p, d, (uint32_t)I->minst_k, cand_head); cand_head = ci; }
      continue;
    }
    if (occluded || !active) continue;  // lanes that already found an occluder idle; the loop itself stays wave-uniform
    if (prim == 3) {
      float tm;
      if (!box_test(o, winv, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2], tm)) continue;
      if (tm > prune_bound(t_max)) continue;  // the box begins beyond the light: nothing in it can occlude
    }
    cnt.inst();
    if (prim == 1) {
      cnt.sphere();
      occluded = sphere_any(p, d, t_max);
    } else if (prim == 2) {
      cnt.plane();
      occluded = plane_any(p, d, t_max);
    } else if (prim == 3) {
      V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
      occluded = mesh_any<CNT>(S, S.meshes[I->mesh], p, d, inv, t_max, stk, stride, cnt);
    }
  }
  return occluded;
}
// What can be said about the rays from P to the points within R of C without tracing them?  (The rays of one soft-shadow
// group: PointLight::Illuminate's disc samples lie within the light's size of its centre, lights.cpp:34-60; a group whose
// first shadow_min rays are all unoccluded is fully lit, :56-59, and one whose every ray is occluded sums to nothing, :66.)
//   RT_CONE_CLEAR  no ray can be occluded: the group is fully lit;
//   RT_CONE_UMBRA  every ray is occluded (by one sphere): the group contributes nothing;
//   RT_CONE_TRACE  neither is proven.
// Conservative, per instance, against what the per-ray tests would decide (any_instances).  Every point of such a ray lies
// within s R of the point at fraction s of the segment P -> C, and its direction within asin(R / |C - P|) of the segment's.
//   clear of an instance: the segment misses its padded world box grown by R;
//     a plane (z = 0 in its own space, objects.cpp:152-167) when P and every end point lie on one side of it, or the crossing
//     comes before the test's t > 0.0002 or after the light;
//     a sphere (objects.cpp:63-82) when the segment stays farther from its centre than 1 + the cone's radius there, or when
//     P lies on or outside it and every ray moves away from the centre (the sphere P itself is on: the far root is <= 0,
//     or below the test's t > 0.01 when rounding left P just inside);
//     a mesh whose grown box the segment touches is taken to occlude.
//   umbra of a sphere: every direction of the cone lies inside the sphere's outline seen from P, the sphere begins more
//     than the test's 0.01 away and before the light; or P lies on the sphere, every ray enters it and leaves it again
//     between 0.01 and the light (the far side of the sphere P itself is on).
// RT_CONE_CLEAR / RT_CONE_UMBRA are proofs.  Every lane of the wave must call (wave-uniform loop over the scalar-loaded
// instance table); `active`: this lane has a group.
// Upper bound on |M v| / |v| over a chain of world -> local 3 x 3 parts (`itm`, 24 floats apart, root first): per level the
// square root of the largest row sum of |M M^T| (Gershgorin; exact for rotation x uniform scale), a hair up for rounding.
RT_HD float chain_stretch(const float* itm, int depth) {
  float grow = 1.0f;
  for (int k = 0; k < depth; k++) {
    const float* m = itm + 24 * k;
    const float g00 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2], g11 = m[3] * m[3] + m[4] * m[4] + m[5] * m[5], g22 = m[6] * m[6] + m[7] * m[7] + m[8] * m[8];
    const float g01 = fabsf(m[0] * m[3] + m[1] * m[4] + m[2] * m[5]), g02 = fabsf(m[0] * m[6] + m[1] * m[7] + m[2] * m[8]),
                g12 = fabsf(m[3] * m[6] + m[4] * m[7] + m[5] * m[8]);
    grow *= sqrtf(fmaxf(g00 + g01 + g02, fmaxf(g01 + g11 + g12, g02 + g12 + g22))) * 1.0001f;
  }
  return grow;
}
// v_rcp_f32 / v_sqrt_f32 (1 ulp) for bounds that carry margins of 1e-3 and more: the correctly rounded forms cost ten
// instructions each, and cone_clear is bound by VALU issue
#ifndef RT_EMU
RT_DEV float rt_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV float rt_fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
RT_DEV float rt_fast_rcp(float x) { return 1.0f / x; }
RT_DEV float rt_fast_sqrt(float x) { return sqrtf(x); }
#endif
#define RT_CONE_TRACE 0
#define RT_CONE_CLEAR 1
#define RT_CONE_UMBRA 2
// The group's cone: rays from P to the points within R of C (a = C - P).
struct Cone {
  V3 P, a, ainv;
  float len, R, s_max, Rw;
  bool ainv_finite, sized;
};
RT_DEV Cone cone_of(V3 P, V3 C, float R, bool active) {
  Cone K;
  K.P = P; K.R = R;
  K.a = C - P;
  K.len = rt_fast_sqrt(dot(K.a, K.a));
  K.sized = active && K.len > 2.0f * R && K.len < 1e30f;   // (a shade point inside or next to the light itself: trace)
  K.ainv = v3(rt_fast_rcp(K.a.x), rt_fast_rcp(K.a.y), rt_fast_rcp(K.a.z));
  K.ainv_finite = fabsf(K.ainv.x) <= RT_BIG && fabsf(K.ainv.y) <= RT_BIG && fabsf(K.ainv.z) <= RT_BIG;
  K.s_max = 1.0f + 1.01f * R * rt_fast_rcp(K.len > 0.0f ? K.len : 1.0f) + 1e-3f;   // rays end at most R beyond C
  K.Rw = R * 1.002f + 1e-4f;
  return K;
}
// Does the cone's axis segment touch the box grown by the cone's radius?  (A segment parallel to an axis has an infinite
// reciprocal, and 0 x inf in a slab is NaN: no box is skipped on that.)
RT_DEV bool cone_touches_box(const Cone& K, float lx, float ly, float lz, float hx, float hy, float hz) {
  float tm;
  return !K.ainv_finite || (box_test(K.P, K.ainv, lx - K.Rw, ly - K.Rw, lz - K.Rw, hx + K.Rw, hy + K.Rw, hz + K.Rw, tm) && !(tm > K.s_max));
}
// One sphere (prim 1) or plane (prim 2) against the cone, in the instance's space: p, d = P and C - P there (the same
// transforms, in the same order, as the rays' own: P lands on the very floats the per-ray tests see), `grow` a bound on how
// far a unit world length reaches there (chain_stretch, computed at commit).  Returns true when no ray of the cone can be
// occluded by the instance; sets `umbra` when every ray is.
RT_DEV bool cone_free_of(const Cone& K, int prim, V3 p, V3 d, float grow, bool touch, bool& umbra) {
  // The one routine of the path compiled WITH floating-point contraction: every comparison below carries a margin of 1e-3 or
  // more against the per-ray tests it bounds, so an fma's missing rounding step cannot change a proof, and the kernel that runs
  // it is bound by VALU issue (98 % of the slots).  Everything that decides a hit keeps the reference's operation order.
#ifndef RT_EMU
#pragma clang fp contract(fast)
#endif
  const float len = K.len, R = K.R, s_max = K.s_max;
  const float mag = fabsf(p.x) + fabsf(p.y) + fabsf(p.z) + fabsf(d.x) + fabsf(d.y) + fabsf(d.z);
  const float slack = 1e-5f * (1.0f + mag);
  const float Rg = R * grow * 1.002f;   // the cone's radius at the light, in this space
  const float Rl = Rg + slack;
  if (prim == 2) {
    const float z0 = p.z;
    const float wlo = d.z - Rl, whi = d.z + Rl;   // z of the rays' end points relative to P's
    const float amin = wlo > 0.0f ? wlo : (whi < 0.0f ? -whi : 0.0f), amax = fmaxf(fabsf(wlo), fabsf(whi));
    const bool same_side = (z0 >= 0.0f && wlo > 0.0f) || (z0 <= 0.0f && whi < 0.0f);
    const bool too_near = amin >= fabsf(z0) * (len + R) * (1.01f / 0.0002f) + 1e-7f;   // t = |z0| / |d.z| <= 0.0002
    const bool not_reached = amax <= 0.99f * fabsf(z0);                                // t >= t_max
    // full shadow of the plane: every ray crosses z = 0 (P on one side, every end point on the other) between the test's
    // t > 0.0002 and its light sample (s* = |z0| / |w.z| in [s_lo, s_hi], t = s* |w_world|, t_max at s = 1), and every crossing
    // point p.xy + s* w.xy, w.xy within Rl of d.xy, lies inside the unit square the test accepts (objects.cpp:152-167)
    if (touch && amin > 0.0f && !same_side) {
      const float s_lo = fabsf(z0) * rt_fast_rcp(amax) * 0.999f, s_hi = fabsf(z0) * rt_fast_rcp(amin) * 1.001f;
      const bool crosses = s_hi <= 0.99f && s_lo * (len - R) > 0.000202f;
      const float xa = d.x - Rl, xb = d.x + Rl, ya = d.y - Rl, yb = d.y + Rl;
      const float x_min = p.x + fminf(s_lo * xa, s_hi * xa), x_max = p.x + fmaxf(s_lo * xb, s_hi * xb);
      const float y_min = p.y + fminf(s_lo * ya, s_hi * ya), y_max = p.y + fmaxf(s_lo * yb, s_hi * yb);
      if (crosses && x_min >= -0.999f && x_max <= 0.999f && y_min >= -0.999f && y_max <= 0.999f) umbra = true;
    }
    return same_side || too_near || not_reached;
  }
  const float dd = dot(d, d), pd = dot(p, d), pp = dot(p, p);
  const float dl = rt_fast_sqrt(dd), pl = rt_fast_sqrt(pp), c = pp - 1.0f;
  const float inv_dd = rt_fast_rcp(dd), inv_dl = rt_fast_rcp(dl);
  const float wmax = dl + Rl, wmin = dl - Rl;   // lengths of the rays here (s = 1 at their light sample)
  // the cone's radius where it could touch: no farther along than where the axis leaves the sphere grown by the
  // radius at the far end
  float sn = dd > 0.0f ? -pd * inv_dd : 0.0f;
  sn = fminf(fmaxf(sn, 0.0f), s_max);
  const float reach0 = 1.0f + s_max * Rg + slack;
  const V3 q0 = p + d * sn;
  bool beyond = dot(q0, q0) > reach0 * reach0 * 1.001f;
  if (!beyond && dd > 0.0f) {
    const float disc = pd * pd - dd * (pp - reach0 * reach0);
    float s_hi = disc > 0.0f ? (-pd + rt_fast_sqrt(disc)) * inv_dd * 1.001f + 1e-6f : s_max;
    s_hi = fminf(fmaxf(s_hi, 0.0f), s_max);
    const float reach1 = 1.0f + s_hi * Rg + slack;
    const V3 q1 = p + d * fminf(sn, s_hi);
    beyond = dot(q1, q1) > reach1 * reach1 * 1.001f;
  }
  // P on or outside the sphere and every ray moving away from its centre
  const float bmin = pd - Rl * pl * 1.01f;
  // (c within rounding of 0 — P on the sphere — is held to the same margins whatever its sign: the per-ray test's own
  //  c may differ from this one in the last ulps if the two sites are ever compiled differently)
  const bool leaving = bmin > 0.0f && (c > 4e-7f * pp || (fabsf(c) * (len + R) <= 0.004f * bmin && 4e-7f * pl * len <= 0.002f * dl));
  if (touch && wmin > 0.0f) {
    const float sin_a = Rl * 1.01f * inv_dl;
    // (b) another sphere in front of the whole light
    const V3 cr = cross(p, d);
    const float sin_f = rt_fast_sqrt(dot(cr, cr)) * inv_dl * 1.0001f;   // (x |p| below: sin f |p| against 0.99)
    const bool covers = pd < 0.0f && pl > 1.0f && sin_f + sin_a * pl < 0.99f && (pl - 1.0f) * (len - R) > 0.02f * wmax && pl < 0.99f * wmin;
    // (a) the sphere P is on, rays going in: the far root 2 |p.w| / w.w, bounded from both sides
    const float bneg = -(pd + Rl * pl * 1.01f), bpos = -pd + Rl * pl * 1.01f;
    const bool through = bneg > 0.0f && fabsf(c) * wmax * wmax <= 0.01f * bneg * bneg && 1.99f * bneg * (len - R) > 0.02f * wmax * wmax &&
                         2.01f * bpos < 0.99f * wmin * wmin;
    if (covers || through) umbra = true;
  }
  return beyond || leaving;
}
// `may_occlude`: bit i = table entry i (< 64) could not be excluded for this group's rays (any_instances<.., MASKED>).
RT_DEV int cone_clear(const DScene& S, V3 P, V3 C, float R, bool active, unsigned long long& may_occlude) {
  RT_UTYPE(DInstance)* inst = RT_UNIFORM_PTR(DInstance, S.inst);
  RT_UTYPE(float)* xf = RT_UNIFORM_PTR(float, S.xf);
  const Cone K = cone_of(P, C, R, active);
  const bool sized = K.sized;
  bool clear = sized, umbra = false;
  may_occlude = sized ? 0ull : ~0ull;
  ParentRay pr;
  pr.key = -2; pr.p = P; pr.d = K.a;
  for (int i = 0; i < S.n_inst; i++) {
    RT_UTYPE(DInstance)* I = inst + i;
    const bool touch = sized && !umbra && cone_touches_box(K, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2]);
    const int prim = I->prim;
    if (!wave_any(touch)) continue;   // [wave-uniform]
    if (prim == 3) {
      if (touch) { clear = false; if (i < 64) may_occlude |= 1ull << i; }
      continue;
    }
    V3 p, d;
    ray_to_instance(I, xf, P, K.a, pr, p, d);
    const bool free_of = cone_free_of(K, prim, p, d, I->stretch, touch, umbra);
    if (touch && !free_of) { clear = false; if (i < 64) may_occlude |= 1ull << i; }
  }
  return umbra ? RT_CONE_UMBRA : (clear ? RT_CONE_CLEAR : RT_CONE_TRACE);
}
// The same proofs for scenes whose rays walk the instance BVH (more instances than a linear table walk per group is worth):
// every lane takes its cone through the BVH over the instances' padded world boxes — a node is entered when the cone's
// axis touches its box grown by the cone's radius, exactly the per-instance box rule above, so an instance is looked at
// iff the table walk would have looked at it — and applies the per-primitive rules at the leaves.  A walk that has not
// ended after RT_CONE_STEPS node visits gives up (the group is traced): cones along a wall of instances would otherwise
// cost more than the rays they might save.
#ifndef RT_CONE_STEPS
#define RT_CONE_STEPS 96
#endif
RT_DEV int cone_clear_tlas(const DScene& S, V3 P, V3 C, float R, bool active, uint32_t* stk, uint32_t stride) {
  const Cone K = cone_of(P, C, R, active);
  if (!K.sized) return RT_CONE_TRACE;
  bool clear = true, umbra = false;
  uint32_t cur = S.tlas_root;
  int sp = 0, steps = 0;
  for (;;) {
    if (cur & RT_LEAF) {
      const uint32_t count = cur & 7u, first = (cur & 0x7fffffffu) >> 3;
      for (uint32_t s = 0; s < count && !umbra; s++) {
        const int i = S.tlas_idx[first + s];
        const DInstance* I = S.inst + i;
        if (!cone_touches_box(K, I->wmin[0], I->wmin[1], I->wmin[2], I->wmax[0], I->wmax[1], I->wmax[2])) continue;
        const int prim = I->prim;
        if (prim == 3) { clear = false; continue; }   // a mesh whose grown box the axis touches is taken to occlude
        V3 p, d;
        ray_to_instance_full(I, S.xf, P, K.a, p, d);
        if (!cone_free_of(K, prim, p, d, I->stretch, true, umbra)) clear = false;
      }
      if (umbra || sp == 0) break;
      cur = stk[(--sp) * stride];
      continue;
    }
    if (++steps > RT_CONE_STEPS) return RT_CONE_TRACE;
    const float4 n0 = S.tlas[4 * cur], n1 = S.tlas[4 * cur + 1], n2 = S.tlas[4 * cur + 2], n3 = S.tlas[4 * cur + 3];
    const bool hl = cone_touches_box(K, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z);
    const bool hr = cone_touches_box(K, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w);
    const uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
    if (hl && hr) {
      stk[(sp++) * stride] = cr;
      cur = cl;
    } else if (hl) {
      cur = cl;
    } else if (hr) {
      cur = cr;
    } else {
      if (sp == 0) break;
      cur = stk[(--sp) * stride];
    }
  }
  return umbra ? RT_CONE_UMBRA : (clear ? RT_CONE_CLEAR : RT_CONE_TRACE);
}

template <bool CNT>
RT_DEV bool scene_any(const DScene& S, V3 o, V3 dir, float t_max, uint32_t* stk, uint32_t stride, Cnt<CNT>& cnt) {
  uint32_t head;
  CandSink none;
  none.rec = nullptr; none.count = nullptr; none.base = 0; none.segcap = 0; none.overflow = nullptr;
  if (S.use_tlas) return any_instances_tlas<CNT, false>(S, o, dir, t_max, head, none, stk, stride, cnt);
  return any_instances<CNT, false>(S, o, dir, t_max, head, none, stk, stride, cnt);
}

// ------------------------------------------------------------------ second phase: per-lane walk over mesh instances
// State of one lane of k_trace_mesh / k_shadow_mesh.  Every lane walks the mesh instances in depth-first order on its
// own (instance tables are read with per-lane loads here), one BVH step per loop iteration, so that the lanes of a
// wave can be inside different instances and finished lanes can take new rays while the others keep walking.
struct MeshWalk {
  V3 p, d, inv;        // ray in the current instance's space
  uint32_t cur;        // node / leaf reference to process next, RT_NOREF when the stack has run empty
  uint32_t pend;       // a leaf reached earlier and not yet intersected (0 = none), see k_trace_mesh
  uint32_t node_base, tri_base;
  int sp;
  uint32_t next;       // the ray's next candidate (RT_NONE: this is the last one)
  int inst;            // instance being walked
  bool fast;           // reciprocal direction finite: min3/max3 slab test (bit-identical decisions, see box_test_fast)
  uint32_t minst;      // ... and its position in DScene::minst_rec
  uint32_t hint;       // any-hit walks (k_shadow_mesh): the triangle slot to head for first while `onpath`
  bool onpath;         // the walk stands on the path from the root to the hinted slot
  uint32_t root;       // the instance's root reference
  uint32_t skip;       // a subtree already searched (RT_NOREF: none), see "around the hint first" in k_shadow_mesh
  bool restart;        // the walk began at `skip` and has the rest of the tree still to do
  uint32_t anc_base, anc_blocks;
};
#ifndef RT_ANC_SHIFT
#define RT_ANC_SHIFT 10
#endif
// RT_ANC_SHIFT:   // DScene::bvh_anc has one entry per 2^RT_ANC_SHIFT triangle slots of a mesh
#define RT_NOREF 0xFFFFFFFFu
RT_DEV bool ref_is_node(uint32_t r) { return !(r & RT_LEAF); }                  // (RT_NOREF has the leaf bit)
RT_DEV bool ref_is_leaf(uint32_t r) { return (r & RT_LEAF) && r != RT_NOREF; }

// Enters the mesh instance of candidate `ci`: the ray is already in the instance's space and is known to hit the mesh's
// root box (first phase, closest_instances / any_instances); what is left is one 32-byte fetch and the walk's constants.
template <bool CNT>
RT_DEV void mesh_walk_enter(const DScene& S, const float4* cand, uint32_t ci, MeshWalk& W, Cnt<CNT>& cnt) {
  const float4 c0 = cand[2 * (size_t)ci], c1 = cand[2 * (size_t)ci + 1];
  const DMeshInst& R = S.minst_rec[__float_as_uint(c0.w)];
  cnt.inst();
  W.p = v3(c0.x, c0.y, c0.z);
  W.d = v3(c1.x, c1.y, c1.z);
  W.inv = v3(1.0f / W.d.x, 1.0f / W.d.y, 1.0f / W.d.z);
  W.fast = inv_is_finite(W.inv);
  W.next = __float_as_uint(c1.w);
  W.cur = R.root_ref; W.pend = 0; W.sp = 0; W.inst = R.inst;
  W.root = R.root_ref; W.skip = RT_NOREF; W.restart = false;
  W.anc_base = R.anc_base; W.anc_blocks = R.anc_blocks;
  W.minst = __float_as_uint(c0.w);
  W.node_base = R.node_base; W.tri_base = R.tri_base;
}

// The 64-byte node record a walking lane stands at.  (A wave-cooperative variant — wanting lanes ranked, sixteen whole
// nodes per load instruction, quarters handed to their owner through LDS — was built and measured on MI355X: correct,
// but 7 % slower on the C2 bench than these four per-lane loads; the walk is bound by VALU issue at ~36 % active lanes,
// not by the texture addresser.  DESIGN.md "What was tried".)
RT_DEV void fetch_node(const DScene& S, const MeshWalk& W, bool want, float4& n0, float4& n1, float4& n2, float4& n3) {
  if (want) {
    const float4* nodes = S.bvh + W.node_base;
    n0 = nodes[4 * W.cur]; n1 = nodes[4 * W.cur + 1]; n2 = nodes[4 * W.cur + 2]; n3 = nodes[4 * W.cur + 3];
  }
}

RT_DEV uint32_t walk_pop(MeshWalk& W, uint32_t* stk, uint32_t stride) {
  if (W.sp == 0) return RT_NOREF;
  return stk[(--W.sp) * stride];
}

// One internal-node step of the walk: both children's slab tests, pruned against `bound_t`; W.cur becomes the next
// reference (RT_NOREF when nothing is left).  ORDERED: the nearer child first (closest hit); otherwise the fixed
// child order of the any-hit walk (see mesh_any_t).  Written without nested branches: one predicated LDS write (both
// children hit), one predicated LDS read (neither hit), selects for the rest.
// HINT (any-hit walks only): while the walk is on the path to W.hint, the child whose slot range holds that slot goes first —
// n3.z is the first slot of the right child's range (the children cover consecutive ranges) — and the walk stays on the
// path iff it entered that child.
template <bool CNT, bool ORDERED, bool HINT = false>
RT_DEV void mesh_walk_node(MeshWalk& W, float4 n0, float4 n1, float4 n2, float4 n3, float bound_t, uint32_t* stk, uint32_t stride,
                           Cnt<CNT>& cnt) {
  float tl, tr;
  cnt.node(2);
  bool hl, hr;
  const float bound = prune_bound(bound_t);
  box_test_walk2(W.p, W.inv, n0, n1, n2, bound, hl, hr, tl, tr);
  if (!W.fast) {  // a zero direction component: the literal hitAABB form (NaN behaviour of FAST_MIN / FAST_MAX)
    hl = box_test(W.p, W.inv, n0.x, n1.x, n2.x, n0.z, n1.z, n2.z, tl) && tl <= bound;
    hr = box_test(W.p, W.inv, n0.y, n1.y, n2.y, n0.w, n1.w, n2.w, tr) && tr <= bound;
  }
  uint32_t cl = __float_as_uint(n3.x), cr = __float_as_uint(n3.y);
  bool left_first = ORDERED ? (tl <= tr) : true;
  if (HINT) {
    hl = hl && cl != W.skip;   // (the subtree the walk began with: found empty of occluders already)
    hr = hr && cr != W.skip;
    const bool hint_left = W.hint < __float_as_uint(n3.z);
    if (W.onpath) left_first = hint_left;
    W.onpath = W.onpath && (hint_left ? hl : hr);
  }
  bool both = hl && hr, any = hl || hr;
  uint32_t nearc = (hl && (!hr || left_first)) ? cl : cr;
  uint32_t farc = left_first ? cr : cl;
  bool pop = !any && W.sp > 0;
  int spn = W.sp - (pop ? 1 : 0);
  uint32_t next = any ? nearc : RT_NOREF;
  if (pop) next = stk[spn * stride];
  if (both) stk[spn * stride] = farc;
  W.sp = spn + (both ? 1 : 0);
  W.cur = next;
}

// Leaf step of the closest-hit walk: every triangle of leaf `ref` against the instance-local best (tie rules of
// mesh_closest_t).
template <bool CNT>
RT_DEV void mesh_walk_leaf_closest(const DScene& S, const MeshWalk& W, uint32_t ref, MeshHit& L, bool& found, Cnt<CNT>& cnt) {
  const float4* tris = S.tris + W.tri_base;
  uint32_t count = ref & 7u, first = (ref & 0x7fffffffu) >> 3;
  if (count == 0) return;
  // the next slot is requested before the current triangle is tested: the loop is not unrolled (1..4 slots), and a
  // fetch per iteration would put a whole memory latency in front of every test
  float4 a = tris[3 * first], b = tris[3 * first + 1], c = tris[3 * first + 2];
  for (uint32_t s = 0; s < count; s++) {
    float4 na = a, nb = b, nc = c;
    if (s + 1 < count) { na = tris[3 * (first + s + 1)]; nb = tris[3 * (first + s + 1) + 1]; nc = tris[3 * (first + s + 1) + 2]; }
    float t, u, v;
    cnt.tri();
    if (tri_test(W.p, W.d, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v)) {
      if (t > 0.002f) {
        uint32_t key = __float_as_uint(a.w);
        if (t < L.t || (t == L.t && found && key > L.key)) {
          L.t = t; L.key = key; L.face = __float_as_uint(b.w); L.u = u; L.v = v; found = true;
        }
      }
    }
    a = na; b = nb; c = nc;
  }
}
// Leaf step of the any-hit walk: true when a triangle of leaf `ref` has t in (0.002, t_max) (objects.cpp:341-371).
template <bool CNT>
RT_DEV bool mesh_walk_leaf_any(const DScene& S, const MeshWalk& W, uint32_t ref, float t_max, Cnt<CNT>& cnt, uint32_t& at_slot) {
  const float4* tris = S.tris + W.tri_base;
  uint32_t count = ref & 7u, first = (ref & 0x7fffffffu) >> 3;
  bool occluded = false;
  if (count == 0) return false;
  float4 a = tris[3 * first], b = tris[3 * first + 1], c = tris[3 * first + 2];  // (fetch ahead, as above)
  for (uint32_t s = 0; s < count; s++) {
    float4 na = a, nb = b, nc = c;
    if (s + 1 < count) { na = tris[3 * (first + s + 1)]; nb = tris[3 * (first + s + 1) + 1]; nc = tris[3 * (first + s + 1) + 2]; }
    float t, u, v;
    cnt.tri();
    if (tri_test(W.p, W.d, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), t, u, v))
      if (t > 0.002f && t < t_max) { occluded = true; at_slot = first + s; break; }
    a = na; b = nb; c = nc;
  }
  return occluded;
}

// ------------------------------------------------------------------ textures (texture.cpp:106-140, scene.h:312-397)
RT_DEV V3 tile_clamp(V3 uvw) {
  V3 u;
  u.x = uvw.x - (float)(int)uvw.x;
  u.y = uvw.y - (float)(int)uvw.y;
  u.z = uvw.z - (float)(int)uvw.z;
  if (u.x < 0) u.x += 1;
  if (u.y < 0) u.y += 1;
  if (u.z < 0) u.z += 1;
  return u;
}
// RGB8 texel as a colour (Color24::ToColor: each channel / 255)
RT_DEV V3 image_texel(const uint8_t* texels, int width, int x, int y) {
  const uint8_t* t = texels + 3 * ((size_t)y * (size_t)width + (size_t)x);
  return v3((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f);
}
// Tiling bilinear lookup (TextureFile::Eval, texture.cpp:106-132): texel (0, 0) is the image's first row, no flip; the four
// texels around (W u, H v) weighted by the fractional parts, neighbours wrapping to the opposite edge.  The tiled
// coordinate lies in [0, 1] — 1 itself when a tiny negative coordinate rounds up in tile_clamp's "+ 1" — so an index
// can equal the image size and wraps once; it is never negative.  The weights and the sum keep the reference's
// operation order (the result has to match it to the last bit for the 1e-4 colour gate on textured scenes).
RT_DEV V3 image_eval(const DScene& S, const DImage& im, V3 uvw) {
  const int W = im.width, H = im.height;
  if (W + H == 0) return v3(0, 0, 0);
  const V3 t = tile_clamp(uvw);
  const float x = W * t.x, y = H * t.y;
  int x0 = (int)x, y0 = (int)y;
  const float fx = x - x0, fy = y - y0;
  if (x0 >= W) x0 -= W;
  if (y0 >= H) y0 -= H;
  const int x1 = x0 + 1 < W ? x0 + 1 : x0 + 1 - W;
  const int y1 = y0 + 1 < H ? y0 + 1 : y0 + 1 - H;
  const uint8_t* texels = S.texels + im.offset;
  const float gx = 1 - fx, gy = 1 - fy;
  return image_texel(texels, W, x0, y0) * (gx * gy) + image_texel(texels, W, x1, y0) * (fx * gy) +
         image_texel(texels, W, x0, y1) * (gx * fy) + image_texel(texels, W, x1, y1) * (fx * fy);
}

// TexturedValue<Color>::Eval.  Checker colours may be textured again (xmlload.cpp:370-374); the nesting is
// walked iteratively: value_0 * value_1 * ... * leaf texture, each level evaluated at the uvw transformed so far.
template <bool CNT>
RT_DEV V3 eval_tex_color(const DScene& S, DTexColor tc, V3 uvw, Cnt<CNT>& cnt) {
  V3 acc = v3(tc.v[0], tc.v[1], tc.v[2]);
  int map = tc.map;
  for (int depth = 0; depth < 8 && map >= 0; depth++) {
    const DTexmap& tm = S.texmaps[map];
    V3 u = xf_point(tm.itm, uvw);
    cnt.tex();
    if (tm.kind == 2) {
      acc = acc * image_eval(S, S.images[tm.image], u);
      return acc;
    }
    V3 t = tile_clamp(u);
    int k = (t.x <= 0.5f) ^ (t.y <= 0.5f);
    DTexColor c = tm.color[k];
    acc = acc * v3(c.v[0], c.v[1], c.v[2]);
    map = c.map;
    uvw = u;
  }
  return acc;
}
template <bool CNT>
RT_DEV float eval_tex_float(const DScene& S, DTexFloat tf, V3 uvw, Cnt<CNT>& cnt) {
  if (tf.map < 0) return tf.v;
  DTexColor tc;
  tc.v[0] = tc.v[1] = tc.v[2] = 1.0f;
  tc.map = tf.map;
  return tf.v * eval_tex_color<CNT>(S, tc, uvw, cnt).x;
}
template <bool CNT>
RT_DEV V3 eval_environment(const DScene& S, DTexColor tc, V3 dir, Cnt<CNT>& cnt) {  // scene.h:377-384
  if (tc.map < 0) return v3(tc.v[0], tc.v[1], tc.v[2]);
  float len = length(dir);
  float z = asinf(-dir.z / len) / RT_PI_F + 0.5f;
  float x = dir.x / (fabsf(dir.x) + fabsf(dir.y));
  float y = dir.y / (fabsf(dir.x) + fabsf(dir.y));
  return eval_tex_color<CNT>(S, tc, v3(0.5f + 0.5f * z * (x - y), 0.5f + 0.5f * z * (x + y), 0.0f), cnt);
}

// ---- RT_EXT_TEX_FILTER: Texture::Eval(uvw, duvw) (scene.h:289-303) — the texture averaged over RT_FILTER_TAPS lookups placed by
// the Halton points (bases 2 and 3, wrapped to (-0.5, 0.5]) in the parallelogram the two derivatives span; TextureMap::Eval
// (scene.h:338-343) takes the derivatives through the map's transform as directions.  A checker's colours are evaluated
// unfiltered inside every tap (texture.cpp:136-140 calls color[..].Eval(uvw)).  With zero derivatives this IS the plain lookup
// (scene.h:292), which is all the reference ever reaches: it never fills HitInfo::duvw.
template <bool CNT>
RT_DEV V3 texmap_tap(const DScene& S, const DTexmap& tm, V3 u, Cnt<CNT>& cnt) {
  if (tm.kind == 2) return image_eval(S, S.images[tm.image], u);
  const V3 t = tile_clamp(u);
  const int k = (t.x <= 0.5f) ^ (t.y <= 0.5f);
  return eval_tex_color<CNT>(S, tm.color[k], u, cnt);
}
template <bool CNT>
RT_DEV V3 eval_tex_color_filtered(const DScene& S, DTexColor tc, V3 uvw, V3 d0, V3 d1, Cnt<CNT>& cnt) {
  const V3 value = v3(tc.v[0], tc.v[1], tc.v[2]);
  if (tc.map < 0) return value;
  const DTexmap& tm = S.texmaps[tc.map];
  const V3 u = xf_point(tm.itm, uvw), e0 = xf_dir(tm.itm, d0), e1 = xf_dir(tm.itm, d1);
  cnt.tex();
  V3 c = texmap_tap<CNT>(S, tm, u, cnt);
  if (dot(e0, e0) + dot(e1, e1) == 0.0f) return value * c;
  const float* tab = S.halton + RT_HGLOSS + 2 * S.n_gloss;
  for (int i = 1; i < RT_FILTER_TAPS; i++) {
    const float x = tab[i], y = tab[RT_FILTER_TAPS + i];
    c = c + texmap_tap<CNT>(S, tm, u + x * e0 + y * e1, cnt);
  }
  return value * (c / (float)RT_FILTER_TAPS);
}
template <bool CNT>
RT_DEV float eval_tex_float_filtered(const DScene& S, DTexFloat tf, V3 uvw, V3 d0, V3 d1, Cnt<CNT>& cnt) {
  if (tf.map < 0) return tf.v;
  DTexColor tc;
  tc.v[0] = tc.v[1] = tc.v[2] = 1.0f;
  tc.map = tf.map;
  return tf.v * eval_tex_color_filtered<CNT>(S, tc, uvw, d0, d1, cnt).x;
}

// HitInfo::duvw (scene.h:82) as this build fills it for camera rays: the texture coordinates where the rays through the
// neighbouring pixel to the right (dx) and below (dy), from the same origin, meet the surface's local approximation, minus
// the hit's own — in the instance's space: the plane itself; the sphere's tangent plane at the hit, projected back onto the
// sphere; the hit triangle's plane with its barycentric interpolation.  A ray parallel to that approximation gives 0.
RT_DEV V3 sphere_uvw(V3 hp) {   // objects.cpp:40-41
  return v3((float)(((double)atan2f(hp.y, hp.x) / (2 * RT_PI_D)) + .5), (float)((double)asinf(hp.z) / RT_PI_D + .5), 0.0f);
}
RT_DEV void hit_differentials(const DScene& S, const DShadeInst& I, V3 o, V3 dir, V3 dx, V3 dy, const Best& B, V3 uvw, V3& du0, V3& du1) {
  const float* chain = S.xf + I.xf_off;
  V3 p = o, d = dir, ex = dx, ey = dy;
  for (int k = 0; k < I.depth; k++) {
    const float* m = chain + 24 * k;
    V3 np = xf_point(m, p), nd = xf_dir(m, d), nx = xf_dir(m, ex), ny = xf_dir(m, ey);
    p = np; d = nd; ex = nx; ey = ny;
  }
  du0 = du1 = v3(0, 0, 0);
  for (int a = 0; a < 2; a++) {
    const V3 e = a ? ey : ex;
    V3 uv2 = uvw;
    if (I.prim == 2) {
      if (!(fabsf(e.z) < 1e-8f)) {
        const float t = -(p.z / e.z);
        const V3 hp = p + e * t;
        uv2 = v3((hp.x + 1.0f) * 0.5f, (hp.y + 1.0f) * 0.5f, 0.0f);
      }
    } else if (I.prim == 1) {
      const V3 hp0 = p + (d * B.t);
      const float den = dot(e, hp0);
      if (!(fabsf(den) < 1e-8f)) {
        const float t = dot(hp0 - p, hp0) / den;
        uv2 = sphere_uvw(normalized(p + e * t));
        const float du = uv2.x - uvw.x;   // the seam of atan2: the nearer way round
        if (du > 0.5f) uv2.x -= 1.0f;
        if (du < -0.5f) uv2.x += 1.0f;
      }
    } else if (I.has_uvw && I.face_pos_base != RT_NONE) {
      const float4* fp = S.face_pos + 3 * ((size_t)I.face_pos_base + B.mh.face);
      const float4 a0 = fp[0], a1 = fp[1], a2 = fp[2];
      const V3 v0 = v3(a0.x, a0.y, a0.z), e1 = v3(a1.x, a1.y, a1.z), e2 = v3(a2.x, a2.y, a2.z);
      const V3 h = cross(e, e2);
      const float det = dot(e1, h);
      if (!(fabsf(det) < 1e-20f)) {
        const float inv_det = 1.0f / det;
        const V3 s = p - v0;
        const float u = inv_det * dot(s, h);
        const V3 q = cross(s, e1);
        const float v = inv_det * dot(e, q);
        const float w = 1.0f - u - v;
        const float4* fr = S.shade + 5 * ((size_t)I.shade_base + B.mh.face);
        const float4 f0 = fr[0], f1 = fr[1], f2 = fr[2], f3 = fr[3], f4 = fr[4];
        uv2 = (v3(f0.w, f1.w, f2.w) * w) + (v3(f3.x, f3.y, f3.z) * u) + (v3(f3.w, f4.x, f4.y) * v);
      }
    }
    // A ray that grazes the surface meets its local approximation arbitrarily far away: the derivative is held to
    // RT_DUVW_MAX texture tiles per pixel (a footprint that size averages the texture anyway, and the taps' coordinates
    // stay where tile_clamp's integer part is defined); a NaN (a degenerate approximation) becomes the limit as well.
    V3 dd = uv2 - uvw;
    dd.x = fminf(fmaxf(dd.x, -RT_DUVW_MAX), RT_DUVW_MAX);
    dd.y = fminf(fmaxf(dd.y, -RT_DUVW_MAX), RT_DUVW_MAX);
    dd.z = fminf(fmaxf(dd.z, -RT_DUVW_MAX), RT_DUVW_MAX);
    if (a) du1 = dd; else du0 = dd;
  }
}

// Color24(Color) channel and Linear2sRGB as this build defines them (see oracle.cpp)
RT_HD uint8_t to_byte(float v) {
  if (!(v == v)) return 0;
  float s = v * 255.0f + 0.5f;
  if (s <= 0.0f) return 0;
  if (s >= 255.0f) return 255;
  return (uint8_t)(int)s;
}
RT_HD float linear_to_srgb(float v) { return v < 0.0031308f ? v * 12.92f : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f; }
