/*
 * rtcore.h — C-ABI of the MI355X-native renderer core (librtcore.so).
 *
 * This is the drop-in boundary for the hot path of devinFink/RayTracer: the host
 * (scene loader, Scene/Camera/Node model, PNG output) stays on the CPU and calls
 * these entry points instead of the reference's in-process C++ renderer
 * (`class Renderer` virtuals, reference RayTracer/renderer.h:209-238, implemented by
 * `class RayTracer`, RayTracer/raytracer.h:14-61).  Plain C, opaque handle, plain
 * pointers and sizes, int status codes (0 = ok, <0 = error, message via
 * rt_last_error).  No exceptions cross the boundary, nothing here names torch.
 *
 * The same header is the input contract of the CPU oracle (oracle/, test
 * infrastructure only): both consume one `rt_scene_desc`.
 *
 * Conventions
 *   - All matrices are 3x4, column-major: m[0..2] = column 0, m[3..5] = column 1,
 *     m[6..8] = column 2, m[9..11] = translation.  (cyMatrix34 cell order as used by
 *     the reference's Transformation, RayTracer/scene.h:152-192.)
 *   - Nodes are stored in depth-first order; node 0 is the scene root
 *     (`Scene::rootNode`, RayTracer/scene.h:484); `parent` < own index.
 *   - Every pointer inside rt_scene_desc is a HOST pointer owned by the caller;
 *     rt_scene_commit copies what it needs to the device.
 */
#ifndef RTCORE_H_INCLUDED
#define RTCORE_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* bad argument / inconsistent scene */
#define RT_ERR_DEVICE (-2)      /* HIP error */
#define RT_ERR_NOSCENE (-3)     /* call needs a committed scene */
#define RT_ERR_UNSUPPORTED (-4) /* feature not available in this build */
#define RT_ERR_OVERFLOW (-5)    /* internal queue capacity exceeded */

/* hit sides — reference RayTracer/scene.h:67-70 */
#define RT_HIT_NONE 0
#define RT_HIT_FRONT 1
#define RT_HIT_BACK 2
#define RT_HIT_FRONT_AND_BACK 3

/* ------------------------------------------------------------------ scene */
typedef struct rt_mat34 { float m[12]; } rt_mat34;

enum { RT_PRIM_NONE = 0, RT_PRIM_SPHERE = 1, RT_PRIM_PLANE = 2, RT_PRIM_MESH = 3 };

/* One scene-graph node (reference `class Node`, RayTracer/scene.h:401-441). */
typedef struct rt_node {
  int32_t parent;      /* -1 for the root */
  int32_t prim_kind;   /* RT_PRIM_* : theSphere / thePlane / TriObj (xmlload.cpp:143-161) */
  int32_t mesh_id;     /* index into rt_scene_desc.meshes when prim_kind == RT_PRIM_MESH */
  int32_t material_id; /* index into materials, -1 = none (default Blinn is substituted) */
  rt_mat34 tm;         /* local -> parent   (Transformation::tm,  scene.h:155) */
  rt_mat34 itm;        /* parent -> local   (Transformation::itm, scene.h:156) */
} rt_node;

/* 2-wide BVH node over one mesh, 32 bytes (stand-in for cyBVH's node: 6-float box +
 * child/leaf word; reference use: RayTracer/objects.cpp:268-371).
 * Internal node: a = left child index, b = right child index.
 * Leaf node:     a = first index into bvh_elements, b = 0x80000000u | element count (1..4). */
typedef struct rt_bvh_node {
  float bmin[3];
  uint32_t a;
  float bmax[3];
  uint32_t b;
} rt_bvh_node;
#define RT_BVH_LEAF_BIT 0x80000000u

/* Triangle mesh (reference `class TriObj : Object, cyTriMesh`, RayTracer/objects.h:52-75). */
typedef struct rt_mesh {
  const float* positions;   /* 3 floats per vertex            (cyTriMesh::V)  */
  const float* normals;     /* 3 floats per normal            (cyTriMesh::VN) */
  const float* uvw;         /* 3 floats per texture vertex    (cyTriMesh::VT) */
  const uint32_t* pos_idx;  /* 3 per face                     (cyTriMesh::F)  */
  const uint32_t* nrm_idx;  /* 3 per face                     (cyTriMesh::FN) */
  const uint32_t* uvw_idx;  /* 3 per face, may be NULL => uvw (0,0,0)  (cyTriMesh::FT) */
  uint32_t n_positions, n_normals, n_uvw, n_faces;
  const rt_bvh_node* bvh;   /* node 0 = root                  (BVHTriMesh, objects.h:65); NULL (with n_bvh_nodes = 0):
                             * rt_scene_commit builds the tree on the device (linear BVH, leaves <= 4) */
  const uint32_t* bvh_elements; /* face ids, grouped per leaf */
  uint32_t n_bvh_nodes, n_bvh_elements;
  float bound_min[3], bound_max[3]; /* cyTriMesh::GetBoundMin/Max (objects.h:57) */
} rt_mesh;

/* TexturedValue<Color> / TexturedValue<float> (RayTracer/scene.h:356-397) */
typedef struct rt_tex_color { float value[3]; int32_t map; } rt_tex_color; /* map: index into texmaps, -1 = none */
typedef struct rt_tex_float { float value; int32_t map; } rt_tex_float;

enum { RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2 };
/* TextureMap = Transformation + Texture* (RayTracer/scene.h:330-349) */
typedef struct rt_texmap {
  int32_t kind;          /* RT_TEX_CHECKER (texture.cpp:136-140) or RT_TEX_IMAGE (texture.cpp:106-132) */
  int32_t image;         /* index into images (RT_TEX_IMAGE) */
  rt_mat34 itm;          /* uvw -> texture space (TextureMap::Eval uses TransformTo, scene.h:337) */
  rt_tex_color color[2]; /* checker colours; may themselves be textured (xmlload.cpp:370-374) */
} rt_texmap;

typedef struct rt_image { const uint8_t* rgb; int32_t width, height; } rt_image; /* RGB8, row 0 first */

/* Material kinds.  The reference's MtlPhong::Shade and MtlMicrofacet::Shade return white and their GenerateSample absorbs
 * (stubs, materials.cpp:392-408); that is what every kind other than RT_MTL_BLINN renders as by default.  With
 * RT_EXT_MATERIALS set in rt_render_params.extensions, RT_MTL_PHONG and RT_MTL_MICROFACET are shaded by this build's own
 * models (no counterpart in the reference; the oracle restates them):
 *   RT_MTL_PHONG       MtlBlinn::Shade with the Phong lobe, ks (alpha + 2) / (2 pi) max(R.V, 0)^alpha, R = the light's mirror
 *                      direction, in place of the half-vector lobe; everything else as MtlBlinn.
 *   RT_MTL_MICROFACET  GGX / Smith / Schlick with the metallic workflow of MtlMicrofacet's parameters (materials.h:115-121),
 *                      stored in the rt_material slots:  diffuse = color (base colour), glossiness = roughness,
 *                      specular.value[0] (+ map) = metallic, refraction = transmittance, emission, absorption, ior. */
enum { RT_MTL_BLINN = 1, RT_MTL_WHITE = 2, RT_MTL_PHONG = 3, RT_MTL_MICROFACET = 4 };
/* rt_render_params.extensions (0 = the reference's behaviour) */
#define RT_EXT_MATERIALS 1u   /* real MtlPhong / MtlMicrofacet (above) */
#define RT_EXT_TEX_FILTER 2u  /* camera rays carry ray differentials into HitInfo::duvw and textures are evaluated with the
                               * reference's filtered lookup Texture::Eval(uvw, duvw) (scene.h:289-303: 32 Halton-placed
                               * lookups over the pixel's footprint); the reference never fills duvw (it stays 0) */
/* MtlBasePhongBlinn fields (RayTracer/materials.h:52-60) */
typedef struct rt_material {
  int32_t kind;
  rt_tex_color diffuse, specular, emission, reflection, refraction;
  rt_tex_float glossiness;
  float absorption[3];
  float ior;
} rt_material;

enum { RT_LIGHT_AMBIENT = 1, RT_LIGHT_DIRECT = 2, RT_LIGHT_POINT = 3 };
/* AmbientLight / DirectLight / PointLight (RayTracer/lights.h:28-77) */
typedef struct rt_light {
  int32_t kind;
  float intensity[3];
  float position[3];  /* point */
  float direction[3]; /* direct, normalised by the loader (xmlload.cpp:232) */
  float size;         /* point: sphere-light radius; 0 = hard shadows */
  float attenuation;  /* point: non-zero => 1/d^2 (lights.cpp:68-71) */
} rt_light;

/* Camera (RayTracer/scene.h:445-467), already post-processed as Camera::Load does
 * (dir = normalise(target-pos), up re-orthogonalised; xmlload.cpp:105-121). */
typedef struct rt_camera {
  float pos[3], dir[3], up[3];
  float fov, focaldist, dof;
  int32_t width, height;
  int32_t srgb;
} rt_camera;

typedef struct rt_scene_desc {
  const rt_node* nodes;         uint32_t n_nodes;
  const rt_mesh* meshes;        uint32_t n_meshes;
  const rt_material* materials; uint32_t n_materials;
  const rt_texmap* texmaps;     uint32_t n_texmaps;
  const rt_image* images;       uint32_t n_images;
  const rt_light* lights;       uint32_t n_lights;
  rt_tex_color background;      /* Scene::background  (scene.h:489) */
  rt_tex_color environment;     /* Scene::environment (scene.h:490) */
  rt_camera camera;
} rt_scene_desc;

/* ------------------------------------------------------------- rays / hits */
/* Ray (RayTracer/scene.h:55-63); invDir is derived inside. */
typedef struct rt_ray { float p[3]; float dir[3]; } rt_ray;

/* HitInfo (RayTracer/scene.h:74-90) in world space, before SamplerInfo::SetHit. */
typedef struct rt_hit {
  float p[3];
  float z;        /* ray parameter of the hit; FLT_MAX on miss */
  float N[3];     /* un-normalised shading normal in world space */
  float uvw[3];
  int32_t node;   /* node index of the hit object, -1 on miss or light */
  int32_t face;   /* face id for meshes, else -1 */
  int32_t light;  /* index of the renderable light that was hit, else -1 */
  int32_t front;  /* 1 front / 0 back */
} rt_hit;

/* ---------------------------------------------------------- render params */
enum { RT_SAMPLER_PHILOX = 0, RT_SAMPLER_PCG_HALTON = 1 /* CPU oracle only */ };

/* Every compile-time constant of the reference renderer as data
 * (raytracer.h:17-23,46; renderer.h:172-174; RayTracer.cpp:191; materials.cpp:316,320). */
typedef struct rt_render_params {
  int32_t width, height;      /* 0 = use the camera's */
  int32_t sampler;            /* RT_SAMPLER_* */
  uint32_t seed_lo, seed_hi;  /* Philox key */
  int32_t spp_min, spp_max;   /* reference 32 / 128; equal => fixed spp, no adaptive stop */
  int32_t sample_begin;       /* first camera-sample index rendered by this call (multi-GPU sample split) */
  int32_t sample_count;       /* number of samples rendered by this call; 0 = spp_max - sample_begin */
  float adaptive_threshold;   /* 0.01 */
  int32_t pixel_jitter;       /* 1 = Halton jitter + lens sampling (reference), 0 = pixel centre, no DOF */
  int32_t max_bounces;        /* bounceCount = 3 */
  int32_t mc_bounces;         /* monteCarloBounces; 0 = indirect-diffuse sampler off */
  int32_t mc_samples;         /* mcSamples = 1 */
  int32_t shadow_min, shadow_max; /* 16 / 128 */
  float light_size_scale;     /* multiplies every point light's size for shadowing; 0 => one ray to the centre */
  int32_t dedup_hard_shadows; /* 1: a size-0 light fires one shadow ray instead of 16/128 identical ones */
  int32_t photons;            /* 0 = photon term off, 1 = on */
  int32_t n_photons;          /* numPhotons = 100000 per map */
  float photon_radius;        /* 3.0 */
  int32_t photon_k;           /* 128 */
  float ellipticity_global;   /* 1.0 */
  float ellipticity_caustic;  /* 0.25 */
  int32_t skip_zero_weight;   /* 1: do not trace secondary rays whose contribution is multiplied by zero (result-identical) */
  int32_t tile_begin, tile_count, tile_stride; /* tile split: render 32x32 tiles t = tile_begin + k*tile_stride; tile_count 0 = all */
  int32_t photon_emit_limit;  /* stop emitting after this many photons even if a map is not full (the reference loops
                                 forever then, RayTracer.cpp:277); 0 = 20 * n_photons */
  int32_t photon_max_bounces; /* cap on the photon random walk (the reference has none, RayTracer.cpp:325); 0 = 64 */
  int32_t frame_mem_mb;       /* device memory the stage queues of one pass may take, MiB; 0 = 32 GB.  Larger passes
                                 amortise kernel tails; a host with less free HBM lowers it. */
  uint32_t extensions;        /* RT_EXT_* bits: capabilities beyond the reference; 0 = off */
} rt_render_params;

/* Output images.  Any pointer may be NULL.  Host memory unless `on_device` != 0. */
typedef struct rt_framebuffer {
  uint8_t* rgb8;          /* W*H*3, Color24 image (RenderImage::img, renderer.h:38) */
  float* radiance;        /* W*H*4: mean linear r,g,b and sample count */
  int32_t* sample_count;  /* W*H      (RenderImage::sampleCount, renderer.h:41) */
  int32_t* hit_node;      /* W*H: node index hit by the first camera sample's primary ray, -1 miss, -2-l light l */
  int32_t* hit_face;      /* W*H: face id for mesh hits, else -1 */
  int32_t on_device;
  int32_t reserved;
  /* Geometry buffers of the first camera sample's primary hit: what the viewer reads from RenderImage::zbuffer
   * (renderer.h:39,74; the reference's RunThread stores 0 there, RayTracer.cpp:215) and the guide images of the
   * Denoise(color, albedo, normal) overload the reference declares (denoiser.h:16-17). */
  float* depth;           /* W*H: distance from the ray origin to the hit (HitInfo::z after SetHit), BIGFLOAT on a miss */
  float* normal;          /* W*H*3: unit shading normal in world space; 0 for a miss or a light */
  float* albedo;          /* W*H*3: the material's diffuse colour at the hit (texture included); the background colour
                             on a miss, 1 for a light or a non-Blinn material */
} rt_framebuffer;

/* Algorithmic work counters. */
typedef struct rt_counters {
  uint64_t rays_primary, rays_secondary, rays_shadow, rays_photon;
  uint64_t inst_tests, sphere_tests, plane_tests, node_visits, tri_tests;
  uint64_t shades, tex_evals, photon_visits, gathers;
  uint64_t paths;
  /* the share of inst_tests / node_visits / tri_tests done by the closest-hit kernel (roofline attribution) */
  uint64_t closest_inst_tests, closest_node_visits, closest_tri_tests;
  /* mesh instances entered, mesh work items and rays finished by the second (mesh) phase of the closest-hit kernels */
  uint64_t mesh_inst_tests, mesh_items;
  /* shadow rays the reference traces (lights.cpp:34-66) and this library did not have to: soft-shadow groups with a proven
   * clear view of their light (shadow_min each) or proven to lie in full shadow (shadow_max each); not part of rays_shadow */
  uint64_t rays_shadow_proven;
} rt_counters;

/* RT_K_TRACE_CLOSEST / RT_K_SHADOW: first phase (instance table: spheres, planes, mesh world boxes);
 * RT_K_TRACE_MESH / RT_K_SHADOW_MESH: second phase (BVH walk of the rays that reach a mesh). */
enum { RT_K_RAYGEN = 0, RT_K_TRACE_CLOSEST, RT_K_SHADE, RT_K_SHADOW, RT_K_ACCUM, RT_K_PHOTON_TRACE, RT_K_PHOTON_GATHER, RT_K_DENOISE,
       RT_K_TRACE_MESH, RT_K_SHADOW_MESH,
       RT_K_PHOTON_BUILD, /* kd-tree build of the photon maps (PrepareForIrradianceEstimation): sorts + level kernels, one entry per build */
       RT_K_COUNT };
/* HIP-event timing of the library's own kernels on its own stream. */
typedef struct rt_profile {
  double ms[RT_K_COUNT];
  uint64_t launches[RT_K_COUNT];
  uint64_t items[RT_K_COUNT]; /* rays / shade points / pixels processed */
} rt_profile;

/* 32-byte photon record, layout-compatible with PhotonMap::PhotonData (photonmap.h:42-49). */
typedef struct rt_photon {
  float position[3];
  float power;
  float dir[3];
  uint8_t color[3];
  uint8_t plane;
} rt_photon;

typedef struct rt_context rt_context;

/* ------------------------------------------------------------ entry points */

/* Replaces `new RayTracer` (RayTracer/main.cpp:11).  One context drives one GPU. */
int rt_context_create(int device_id, rt_context** out);
void rt_context_destroy(rt_context* ctx);
const char* rt_last_error(const rt_context* ctx);
/* A host that drives several GPUs from ONE process (the reference starts hardware_concurrency() threads on one renderer,
 * RayTracer.cpp:41-50; its multi-GPU counterpart is one context per GPU): creates n_devices contexts, out[i] on
 * device_ids[i] (device_ids NULL = devices 0 .. n_devices-1).  Each context is single-owner with its own stream, so the host
 * may drive them from one thread (rt_render_accumulate enqueues and returns) or from one thread each; their accumulators
 * are summed with rt_reduce_accumulators or by the host.  All or nothing: on failure every context created so far is
 * destroyed and out[] is nulled.  A device id may appear once. */
int rt_context_create_multi(int n_devices, const int* device_ids, rt_context** out);
/* Page-locked host memory for the caller-owned buffers of rt_framebuffer / rt_trace_* / rt_photons_download (the
 * reference's RenderImage owns plain `new` arrays, renderer.h:35-46; with pinned memory the device-to-host copies at the
 * end of rt_render run at full PCIe rate and without a staging copy).  Freed with rt_free_pinned only. */
int rt_alloc_pinned(size_t bytes, void** out);
void rt_free_pinned(void* p);

/* Replaces the in-memory hand-over at the end of Renderer::LoadScene (xmlload.cpp:70-72):
 * the flattened scene is copied to the device, instance chains and acceleration data are built. */
int rt_scene_commit(rt_context* ctx, const rt_scene_desc* scene);

/* The BVH of mesh `mesh_id` of the committed scene in rt_mesh form (nodes in depth-first pre-order, node 0 = root;
 * per-leaf face lists): the tree the mesh came with, or the one rt_scene_commit built on the device for a mesh
 * without one (reference: BVHTriMesh inside TriObj::Load, RayTracer/objects.h:60-67).  `n_nodes` / `n_elements`
 * always receive the sizes; the arrays are filled when both capacities suffice (RT_ERR_INVALID otherwise, unless
 * `nodes` is NULL = size query). */
int rt_scene_mesh_bvh(rt_context* ctx, int mesh_id, rt_bvh_node* nodes, uint32_t node_cap, uint32_t* elements,
                      uint32_t elem_cap, uint32_t* n_nodes, uint32_t* n_elements);

/* Batched Renderer::TraceRay (RayTracer.cpp:336-339 -> TraverseTree :373-435). Host pointers. */
int rt_trace_closest(rt_context* ctx, const rt_ray* rays, size_t n, int hit_side, rt_hit* hits);
/* Batched Renderer::TraceShadowRay (RayTracer.cpp:341-344 -> TraverseTreeShadow :346-371). Host pointers. */
int rt_trace_any(rt_context* ctx, const rt_ray* rays, const float* t_max, size_t n, uint8_t* occluded);

/* Replaces BeginRender + the wait in StopRender (RayTracer.cpp:24-58), blocking:
 * photon pass (if enabled), per-pixel sampling loop, resolve (mean, sRGB, Color24). */
int rt_render(rt_context* ctx, const rt_render_params* params, rt_framebuffer* fb);

/* Asynchronous form mirroring BeginRender ("returns immediately", renderer.h:229), the
 * viewport's polling of GetNumRenderedPixels (viewport.cpp:583-586) and StopRender
 * ("should wait till rendering threads stop", renderer.h:230). */
int rt_render_begin(rt_context* ctx, const rt_render_params* params, rt_framebuffer* fb);
int rt_render_poll(rt_context* ctx, int64_t* pixels_done, int* finished);
int rt_render_wait(rt_context* ctx);

/* Multi-GPU form: add this call's samples into a DEVICE accumulator of W*H*8 floats
 * (sum r,g,b, n, sum r^2,g^2,b^2, unused) that the caller owns (e.g. a torch tensor) and may
 * all-reduce over RCCL before rt_resolve.  The per-pixel loop of RunThread (RayTracer.cpp:117-204)
 * without its final division. */
int rt_render_accumulate(rt_context* ctx, const rt_render_params* params, void* d_accum);
/* Final division, sRGB, Color24 (RayTracer.cpp:206-216) from a device accumulator. */
int rt_resolve(rt_context* ctx, const rt_render_params* params, const void* d_accum, rt_framebuffer* fb);

/* The frame's one exchange for multi-GPU hosts that do not use torch.distributed (the reference's host is C++): RCCL
 * sum-reduce of a device accumulator (W*H*8 floats, layout above) to rank `root` of `nccl_comm` (an ncclComm_t the caller
 * created, one rank per GPU), on the context's stream; blocks until it has completed.  Floats 0..3 of every pixel (sum rgb,
 * n: all a frame of partitioned samples or tiles needs) travel as a packed W*H*4 image — half the accumulator — and are
 * written back into d_accum: the root then holds the sums, every other rank its own values; floats 4..7 (sums of squares,
 * the adaptive sampler's done flag) are left alone on every rank.  `nccl_reduce_fn` is the caller's ncclReduce (pass
 * `(void*)&ncclReduce`), or NULL to look it up among the process's global symbols — librtcore does not link RCCL, so the
 * communicator and the function always come from the same library.  The reference has no counterpart (single process,
 * RayTracer.cpp:41-50); SURVEY 8e defines the exchange. */
int rt_reduce_accumulators(rt_context* ctx, void* nccl_comm, void* nccl_reduce_fn, void* d_accum, int32_t width, int32_t height, int root);

/* GeneratePhotons (RayTracer.cpp:262-294): emission, random walk, both maps, balance. */
int rt_photons_build(rt_context* ctx, const rt_render_params* params);
/* GetPhotonMap / GetCausticsMap (raytracer.h:41-42): which = 0 global, 1 caustics. */
int rt_photons_download(rt_context* ctx, int which, rt_photon* out, int32_t capacity, int32_t* n);
/* The same map after PhotonMap::PrepareForIrradianceEstimation (photonmap.h:176-198): the left-balanced kd-tree in heap
 * order, entry i of `out` = node i + 1 (what PhotonMap::GetPhotons() points at once the map is balanced, photonmap.h:109;
 * the viewer's photon display reads it, viewport.cpp:441-481).  The split axis is in the low two bits of `plane`. */
int rt_photons_download_tree(rt_context* ctx, int which, rt_photon* out, int32_t capacity, int32_t* n);
/* Replaces one map by caller-supplied photons in insertion order, powers already scaled (what a host that traced
 * its own photons through PhotonMap::AddPhoton + ScalePhotonPowers would hold, photonmap.h:82,158-172), then
 * PrepareForIrradianceEstimation (photonmap.h:176-198). */
int rt_photons_set(rt_context* ctx, int which, const rt_photon* photons, int32_t n);
/* Batched PhotonMap::EstimateIrradiance<k> (photonmap.h:98-99,268-309). Host pointers, 3 floats each. */
int rt_photon_gather(rt_context* ctx, int which, const float* P, const float* N, size_t n,
                     float radius, float ellipticity, int k, float* irradiance);

/* Denoiser::Denoise(Color*,Color*) (RayTracer/denoiser.cpp:22-64).  rgb_in/out: W*H*3 floats, host.
 * albedo/normal (W*H*3 each) may be NULL (the reference declares that overload, denoiser.h:16-17). */
int rt_denoise(rt_context* ctx, const float* rgb_in, float* rgb_out, int32_t width, int32_t height,
               const float* albedo, const float* normal);

/* rt_denoise without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot allocate device
 * memory — tight serving loops, harnesses that forbid hipMalloc, hosts whose memory is fully planned.  Same filter and kernels
 * as rt_denoise, but d_rgb_in / d_rgb_out / d_albedo / d_normal are DEVICE pointers (W*H*3 floats each, guides may be NULL)
 * and the filter's working images live in a fixed 64 MiB device array of the library (one copy per GPU, held until the
 * process ends) instead of allocated buffers.  Limit: width * height <= 1048576 pixels (RT_ERR_UNSUPPORTED beyond it).
 * Calls no hipMalloc / hipFree.  It runs on the context's own stream and blocks until the result is in d_rgb_out (so it is
 * not a call to make while a stream is being captured into a HIP graph); whatever produces d_rgb_in and the guides on
 * another stream must have finished before the call (synchronise that stream first).  Every call on one GPU shares that
 * GPU's array: calls that overlap on one GPU (from several contexts or threads) race and must be serialised by the caller. */
int rt_denoise_noalloc(rt_context* ctx, const float* d_rgb_in, float* d_rgb_out, int32_t width, int32_t height,
                       const float* d_albedo, const float* d_normal);

int rt_counters_enable(rt_context* ctx, int enable); /* counted builds of the kernels; off by default */
int rt_counters_reset(rt_context* ctx);
int rt_counters_get(rt_context* ctx, rt_counters* out);
int rt_profile_enable(rt_context* ctx, int enable);
int rt_profile_reset(rt_context* ctx);
int rt_profile_get(rt_context* ctx, rt_profile* out);

/* Developer knobs of one context (the reference's counterparts are compile-time constants, raytracer.h:17-23): "tlas_min"
 * (instances from which the instance BVH replaces the scene-graph table walk; applies at the next rt_scene_commit),
 * "gather_hints" (0: every photon lookup from scratch), "frame_gb" (stage-queue memory of one pass), "fan_scale" (scales the
 * measured queue sizing; tests force the overflow path with it), "timing", "soft_clear" (0: every soft-shadow group is traced;
 * 1, the default: groups with a proven clear view of their light, or proven to lie in a sphere's full shadow, are finished
 * without their rays — the same frame either way, rt_counters.rays_shadow_proven counts what was saved), "cone_max_inst"
 * (instances up to which those proofs walk the instance table; beyond it the instance BVH), "shadow_tlas_min" (instances from
 * which shadow rays walk the instance BVH; next commit), "fuse" (1: first trace phase and shading of the rays it finishes in
 * one kernel), "shadow_hints" (soft-shadow rays head for the triangle that blocked their group's previous ray first: 1 on,
 * 0 off, -1 — the default — by the depth of the scene's mesh trees; the same frame either way), "restart_min_depth" (mesh
 * trees at least this deep let such a walk begin in the subtree around its hint and take the rest of the tree only if that
 * holds no occluder; default 19, next commit; the same frame either way).  Their initial values
 * come from the environment variables RT_TLAS_MIN / RT_GATHER_HINTS / RT_FRAME_GB / RT_TEST_FAN_SCALE / RT_TIMING /
 * RT_SOFT_CLEAR / RT_SHADOW_HINTS, read once by rt_context_create. */
int rt_set_option(rt_context* ctx, const char* name, double value);
/* Number of device allocations (hipMalloc) made so far on behalf of this context: the entry points keep their device
 * buffers in the context and grow them only, so a steady-state caller sees this stand still (no counterpart in the
 * reference, whose renderer allocates per render, RayTracer.cpp:29-38). */
int rt_debug_alloc_count(rt_context* ctx, uint64_t* n);

/* The sampler as the kernels run it, for known-answer tests: Philox4x32-10 of `n` counters (4 words each, host) under the key
 * (key_lo, key_hi) computed ON THE DEVICE -> the 4 output words per counter and their mapping to [0, 1) (RNG::RandomFloat's
 * mapping, rng.h:60-65).  Philox replaces the reference's PCG32 streams (rng.h:29-99); a render uses counter =
 * (pixel, tree node, block, tag) under the key (seed_lo, seed_hi). */
int rt_debug_philox(rt_context* ctx, const uint32_t* counters, uint32_t key_lo, uint32_t key_hi, size_t n, uint32_t* words, float* units);

/* Fills params with the reference's defaults (raytracer.h:17-23, renderer.h:172-174). */
void rt_render_params_default(rt_render_params* params);

#ifdef __cplusplus
}
#endif
#endif
