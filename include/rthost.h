/*
 * rthost.h — C-ABI of the re-authored CPU host layer (librthost.so).
 *
 * The host layer is NOT accelerated; it exists so that the reference's scene files load on
 * Linux without tinyxml2.h / cyCodeBase (absent from the reference checkout) and so that the
 * flattened scene can be handed to the renderer core (rtcore.h) or to the CPU oracle.
 * It replaces, behind plain C entry points:
 *   Renderer::LoadScene           reference RayTracer/xmlload.cpp:42-77
 *   Scene::Load / LoadNode / ...   RayTracer/xmlload.cpp:81-395
 *   TriObj::Load                   RayTracer/objects.h:60-67  (OBJ read, normals, bounds, BVH leaf size 4)
 *   TextureFile::LoadFile          RayTracer/texture.cpp:68-102
 *   RenderImage::SaveImage         RayTracer/renderer.h:77
 */
#ifndef RTHOST_H_INCLUDED
#define RTHOST_H_INCLUDED

#include "rtcore.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* Loads a scene file in the reference's XML dialect.  Texture and OBJ names are resolved
 * relative to `asset_dir` first (NULL = directory of the file), then to the CWD as the
 * reference does (texture.cpp:86,95).  Mesh names of the form
 *   gen:<kind>?seed=<n>&tris=<n>[&size=<f>]   (kind: teapot_like | icosphere | terrain | blob | voxels)
 * are produced by the seeded procedural generator instead of being read from disk
 * (the reference's .obj assets are git-ignored and absent).
 * Returns NULL on error; the message is available from rth_last_error(). */
rth_scene* rth_scene_load_file(const char* path, const char* asset_dir);
rth_scene* rth_scene_load_string(const char* xml_text, const char* asset_dir);
void rth_scene_free(rth_scene* s);
const char* rth_last_error(void);

/* The flattened scene; pointers stay valid until rth_scene_free. */
const rt_scene_desc* rth_scene_desc(const rth_scene* s);
/* Node / material names for diagnostics (NULL if out of range). */
const char* rth_scene_node_name(const rth_scene* s, int node);
/* Number of warnings the loader printed (unknown tags, missing assets...). */
int rth_scene_warnings(const rth_scene* s);

/* Load flags of the *_ex entry points. */
#define RTH_LOAD_NO_BVH 1u /* meshes are handed over without a tree (rt_mesh::bvh == NULL): rt_scene_commit builds a
                              linear BVH on the device.  Default: every mesh gets its BVH here (binned SAH, as
                              TriObj::Load builds one on load, objects.h:65). */
rth_scene* rth_scene_load_file_ex(const char* path, const char* asset_dir, uint32_t flags);
rth_scene* rth_scene_load_string_ex(const char* xml_text, const char* asset_dir, uint32_t flags);

/* The same choice as a switch of the CALLING THREAD for the scenes it loads or generates afterwards through the entry
 * points without flags (1 = build here, the default; 0 = RTH_LOAD_NO_BVH).  Other threads are not affected. */
void rth_set_host_bvh(int on);

/* Synthetic stress scene (BASELINE config 5): n_instances icospheres of 20*4^subdiv triangles
 * baked into ONE world-space mesh plus a displaced ground grid, two sphere lights. */
rth_scene* rth_scene_generate_stress(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height);
rth_scene* rth_scene_generate_stress_ex(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height, uint32_t flags);

/* PNG output (8-bit RGB or grey).  0 on success. */
int rth_write_png(const char* path, const uint8_t* pixels, int width, int height, int channels);
/* PNG / PPM input decoded to RGB8; caller frees with rth_free. */
int rth_read_image_rgb(const char* path, uint8_t** rgb, int* width, int* height);
void rth_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
