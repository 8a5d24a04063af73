/*
 * oracle.h — C-ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference renderer's hot path
 * (devinFink/RayTracer: RayTracer.cpp, objects.cpp, materials.cpp, lights.cpp, photonmap.h, rng.h,
 * texture.cpp) used to check the HIP kernels.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; nothing under raytracer_amd/ links or calls it.
 *
 * It consumes the same flattened scene (`rt_scene_desc`, include/rtcore.h) as the renderer core.
 */
#ifndef ORACLE_H_INCLUDED
#define ORACLE_H_INCLUDED

#include "rtcore.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* flags for orc_scene_create */
#define ORC_REFERENCE_STRUCTURE 0 /* Θ(nodes) scene-graph walk, BVH visits both children, as the reference */

orc_scene* orc_scene_create(const rt_scene_desc* desc);
void orc_scene_destroy(orc_scene* s);

/* 1: the mesh BVHs are walked front to back with pruning AND the primitive nodes sit in a BVH over their world boxes, walked
 * the same way, in place of TraverseTree's test of every node (a CPU renderer with the GPU's data structures: SURVEY 8d's
 * second row) — for the second cpu_baseline row of bench.py ONLY; parity checks always run with 0, the reference's
 * visit-everything recursion.  Equal-t ties between instances may resolve differently with 1. */
void orc_set_accel(orc_scene* s, int on);

/* Renderer::TraceRay / TraceShadowRay restated (RayTracer.cpp:336-435). */
void orc_trace_closest(const orc_scene* s, const rt_ray* rays, size_t n, int hit_side, rt_hit* hits);
void orc_trace_any(const orc_scene* s, const rt_ray* rays, const float* t_max, size_t n, uint8_t* occluded);

/* BeginRender..StopRender restated (RayTracer.cpp:24-225), n_threads tile workers.
 * If params->photons != 0 and no map has been built yet, builds them first (GeneratePhotons). */
int orc_render(orc_scene* s, const rt_render_params* params, rt_framebuffer* fb, int n_threads, rt_counters* counters);
/* Per-sample radiance of one pixel (SendRay, RayTracer.cpp:227-259) for fine-grained checks: out = 3 floats. */
void orc_sample_radiance(orc_scene* s, const rt_render_params* params, int x, int y, int sample, float* out);
/* The primary ray RunThread builds for (x, y, sample) (RayTracer.cpp:138-174). */
void orc_primary_ray(const orc_scene* s, const rt_render_params* params, int x, int y, int sample, rt_ray* out);

/* GeneratePhotons / PhotonMap (RayTracer.cpp:262-334, photonmap.h). */
int orc_photons_build(orc_scene* s, const rt_render_params* params, rt_counters* counters);
int orc_photons_count(const orc_scene* s, int which);
/* Unbalanced insertion-order copy (which: 0 global, 1 caustics). */
int orc_photons_download(const orc_scene* s, int which, rt_photon* out, int32_t capacity);
/* Replace a map by caller-supplied photons (already scaled); balances the kd-tree. */
int orc_photons_set(orc_scene* s, int which, const rt_photon* photons, int32_t n);
/* The balanced map in heap order (entry i = node i + 1), and installing one that is already balanced. */
int orc_photons_tree(const orc_scene* s, int which, rt_photon* out, int32_t capacity);
int orc_photons_set_tree(orc_scene* s, int which, const rt_photon* tree, int32_t n);
void orc_photon_gather(const orc_scene* s, int which, const float* P, const float* N, size_t n, float radius,
                       float ellipticity, int k, float* irradiance);

/* This build's own denoise filter restated on the CPU (the reference's is Intel OIDN: parity unpinned).
 * rgb_in/out, albedo, normal: W*H*3 floats; albedo/normal may be NULL. */
void orc_denoise(const float* rgb_in, float* rgb_out, int width, int height, const float* albedo, const float* normal);

/* rng.h restated */
void orc_pcg32_seq(uint64_t sequence, uint64_t seed, int use_seed, int n, uint32_t* out_u32, float* out_f32);
float orc_halton(int index, int base);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* leaf routines on one ray in OBJECT space: kind = RT_PRIM_SPHERE / RT_PRIM_PLANE; returns hit flag */
int orc_intersect_prim(int kind, const rt_ray* ray, int hit_side, rt_hit* hit);
int orc_shadow_prim(int kind, const rt_ray* ray, float t_max);
int orc_hit_aabb(const rt_ray* ray, const float bounds[6]);
/* TextureFile::Eval / TextureChecker::Eval / TexturedValue::Eval(Environment) restated */
void orc_eval_tex_color(const orc_scene* s, const rt_tex_color* tc, const float uvw[3], float out[3]);
void orc_eval_environment(const orc_scene* s, const rt_tex_color* tc, const float dir[3], float out[3]);
/* Color24(Color) and Linear2sRGB as this build defines them (cyColor.h is absent from the reference) */
uint8_t orc_to_byte(float v);
float orc_linear_to_srgb(float v);
void orc_orthonormals(const float n[3], float t[3], float b[3]);

#ifdef __cplusplus
}
#endif
#endif
