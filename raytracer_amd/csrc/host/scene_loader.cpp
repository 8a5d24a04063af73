// Re-authored scene loader: the reference's XML scene dialect -> flat rt_scene_desc.
// Follows the semantics (not the code) of reference RayTracer/xmlload.cpp:42-395 and the Loader
// helpers in RayTracer/xmlload.h:57-69; see SURVEY.md Appendix B for the dialect.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "hmath.h"
#include "mesh_tools.h"
#include "rthost.h"
#include "xml_lite.h"

using namespace rth;

static thread_local std::string g_err;

struct rth_scene {
  std::vector<rt_node> nodes;
  std::vector<std::string> node_names;
  std::vector<std::string> node_mtl_names;  // material="..." until bound
  std::vector<int> node_wants_objmtl;       // obj node without material= whose mesh carries a .mtl
  std::deque<HostMesh> meshes;
  std::vector<std::string> mesh_names;
  std::vector<rt_mesh> mesh_descs;
  std::vector<rt_material> materials;
  std::vector<std::string> material_names;
  std::vector<rt_texmap> texmaps;
  std::vector<rt_image> images;
  std::vector<std::string> image_names;
  std::vector<uint8_t*> image_data;
  std::vector<rt_light> lights;
  rt_scene_desc desc;
  std::string asset_dir;
  int warnings = 0;

  ~rth_scene() {
    for (auto* p : image_data) std::free(p);
  }
};

namespace {

void warn(rth_scene& s, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::fprintf(stderr, "rthost: ");
  std::vfprintf(stderr, fmt, ap);
  std::fprintf(stderr, "\n");
  va_end(ap);
  s.warnings++;
}

// Loader::ReadFloat / ReadInt (xmlload.h:60-61): attribute must exist and parse.
bool read_float(const XmlElem* e, float& f, const char* name = "value") {
  if (!e) return false;
  const char* a = e->attr(name);
  if (!a) return false;
  char* end;
  float v = std::strtof(a, &end);
  if (end == a) return false;
  f = v;
  return true;
}
bool read_int(const XmlElem* e, int& i, const char* name = "value") {
  if (!e) return false;
  const char* a = e->attr(name);
  if (!a) return false;
  char* end;
  long v = std::strtol(a, &end, 10);
  if (end == a) return false;
  i = (int)v;
  return true;
}
// Loader::ReadVec3f (xmlload.h:62): v = def; per-component overrides; then * value.
void read_vec3(const XmlElem* e, V3& v, V3 def = V3(0, 0, 0)) {
  if (!e) return;
  v = def;
  read_float(e, v.x, "x");
  read_float(e, v.y, "y");
  read_float(e, v.z, "z");
  float f = 1;
  if (read_float(e, f)) v = v * f;
}
// Loader::ReadColor (xmlload.h:63), default white.
void read_color(const XmlElem* e, float c[3], V3 def = V3(1, 1, 1)) {
  if (!e) return;
  V3 v = def;
  read_float(e, v.x, "r");
  read_float(e, v.y, "g");
  read_float(e, v.z, "b");
  float f = 1;
  if (read_float(e, f)) v = v * f;
  c[0] = v.x; c[1] = v.y; c[2] = v.z;
}

struct Xform {  // Transformation (scene.h:152-192)
  M34 tm, itm;
  void apply(const M34& m) { tm = m * tm; itm = tm.inverse(); }
  // Transformation::Load (xmlload.cpp:176-198): children in document order
  void load(const XmlElem* e) {
    for (auto& c : e->children) {
      const char* t = c->tag.c_str();
      if (XmlElem::ieq(t, "scale")) {
        V3 s;
        read_vec3(c.get(), s, V3(1, 1, 1));
        apply(M34::scale(s));
      } else if (XmlElem::ieq(t, "rotate")) {
        V3 s;
        read_vec3(c.get(), s);
        s = normalized(s);
        float a = 0.0f;
        read_float(c.get(), a, "angle");
        const float pi = 3.14159265358979323846f;
        apply(M34::rotation(s, a * pi / 180.0f));
      } else if (XmlElem::ieq(t, "translate")) {
        V3 tr;
        read_vec3(c.get(), tr);
        apply(M34::translation(tr));
      }
    }
  }
};

std::string resolve_path(const rth_scene& s, const std::string& name) {
  auto exists = [](const std::string& p) { std::ifstream f(p); return (bool)f; };
  if (!s.asset_dir.empty()) {
    std::string p = s.asset_dir + "/" + name;
    if (exists(p)) return p;
    size_t sl = name.find_last_of("/\\");
    if (sl != std::string::npos) {
      p = s.asset_dir + "/" + name.substr(sl + 1);  // scenes name both "scenes/bricks.png" and "bricks.png"
      if (exists(p)) return p;
    }
  }
  return name;  // relative to the CWD, as the reference does (texture.cpp:86,95)
}

// ReadTextureFile (xmlload.cpp:378-395): de-duplicated by name; -1 if it cannot be loaded.
int get_image(rth_scene& s, const std::string& name) {
  for (size_t i = 0; i < s.image_names.size(); i++)
    if (s.image_names[i] == name) return (int)i;
  uint8_t* rgb = nullptr;
  int w = 0, h = 0;
  std::string p = resolve_path(s, name);
  if (rth_read_image_rgb(p.c_str(), &rgb, &w, &h) != 0) {
    warn(s, "cannot load texture file %s", name.c_str());
    return -1;
  }
  rt_image im;
  im.rgb = rgb; im.width = w; im.height = h;
  s.images.push_back(im);
  s.image_names.push_back(name);
  s.image_data.push_back(rgb);
  return (int)s.images.size() - 1;
}

void read_textured_color(rth_scene& s, const XmlElem* e, rt_tex_color& tc);

// Loader::ReadTextureMap (xmlload.cpp:347-366)
int read_texture_map(rth_scene& s, const XmlElem* e) {
  const char* tex = e->attr("texture");
  if (!tex) return -1;
  rt_texmap tm;
  std::memset(&tm, 0, sizeof tm);
  tm.image = -1;
  if (XmlElem::ieq(tex, "checkerboard")) {
    tm.kind = RT_TEX_CHECKER;
    // TextureChecker defaults (texture.h:46): black, white
    tm.color[0] = {{0, 0, 0}, -1};
    tm.color[1] = {{1, 1, 1}, -1};
    if (const XmlElem* c1 = e->child("color1")) read_textured_color(s, c1, tm.color[0]);
    if (const XmlElem* c2 = e->child("color2")) read_textured_color(s, c2, tm.color[1]);
  } else {
    int img = get_image(s, tex);
    if (img < 0) return -1;
    tm.kind = RT_TEX_IMAGE;
    tm.image = img;
  }
  Xform x;
  x.load(e);
  x.itm.store(tm.itm);
  s.texmaps.push_back(tm);
  return (int)s.texmaps.size() - 1;
}

// Loader::ReadTexturedColor / ReadTexturedFloat (xmlload.h:64-65)
void read_textured_color(rth_scene& s, const XmlElem* e, rt_tex_color& tc) {
  if (!e) return;
  float c[3] = {1, 1, 1};
  read_color(e, c);
  tc.value[0] = c[0]; tc.value[1] = c[1]; tc.value[2] = c[2];
  tc.map = read_texture_map(s, e);
}
void read_textured_float(rth_scene& s, const XmlElem* e, rt_tex_float& tf) {
  if (!e) return;
  float f = 1.0f;
  read_float(e, f);
  tf.value = f;
  tf.map = read_texture_map(s, e);
}

rt_material default_blinn() {  // MtlBasePhongBlinn defaults (materials.h:53-60)
  rt_material m;
  std::memset(&m, 0, sizeof m);
  m.kind = RT_MTL_BLINN;
  m.diffuse = {{0.5f, 0.5f, 0.5f}, -1};
  m.specular = {{0.7f, 0.7f, 0.7f}, -1};
  m.glossiness = {20.0f, -1};
  m.emission = {{0, 0, 0}, -1};
  m.reflection = {{0, 0, 0}, -1};
  m.refraction = {{0, 0, 0}, -1};
  m.ior = 1.5f;
  return m;
}

// LoadMaterial + MtlBasePhongBlinn::Load (xmlload.cpp:247-277)
void load_material(rth_scene& s, const XmlElem* e) {
  const char* type = e->attr("type");
  rt_material m = default_blinn();
  const char* name = e->attr("name");
  if (XmlElem::ieq(type, "blinn")) {
    m.kind = RT_MTL_BLINN;
  } else if (XmlElem::ieq(type, "phong")) {
    // Shade() returns white in the reference (materials.cpp:404-408) and so does the core by default; the parameters are
    // MtlBasePhongBlinn's and are kept for RT_EXT_MATERIALS
    m.kind = RT_MTL_PHONG;
  } else if (XmlElem::ieq(type, "microfacet")) {
    // MtlMicrofacet::Load (xmlload.cpp:281-290) into the slots include/rtcore.h names: color -> diffuse, roughness ->
    // glossiness, metallic -> specular.value[0] (+ map), transmittance -> refraction; defaults materials.h:115-121.
    // White by default as in the reference (materials.cpp:396-399); shaded under RT_EXT_MATERIALS.
    m.kind = RT_MTL_MICROFACET;
    m.glossiness = {1.0f, -1};
    m.specular = {{0, 0, 0}, -1};
    read_textured_color(s, e->child("color"), m.diffuse);
    read_textured_float(s, e->child("roughness"), m.glossiness);
    rt_tex_float metallic = {0.0f, -1};
    read_textured_float(s, e->child("metallic"), metallic);
    m.specular.value[0] = m.specular.value[1] = m.specular.value[2] = metallic.value;
    m.specular.map = metallic.map;
    read_textured_color(s, e->child("emission"), m.emission);
    read_textured_color(s, e->child("transmittance"), m.refraction);
    read_float(e->child("ior"), m.ior);
    if (const XmlElem* a = e->child("absorption")) read_color(a, m.absorption);
    s.materials.push_back(m);
    s.material_names.push_back(name ? name : "");
    return;
  } else {
    warn(s, "unknown material type %s", type ? type : "(null)");
    return;
  }
  read_textured_color(s, e->child("diffuse"), m.diffuse);
  read_textured_color(s, e->child("specular"), m.specular);
  read_textured_float(s, e->child("glossiness"), m.glossiness);
  read_textured_color(s, e->child("emission"), m.emission);
  read_textured_color(s, e->child("reflection"), m.reflection);
  read_textured_color(s, e->child("refraction"), m.refraction);
  read_float(e->child("refraction"), m.ior, "index");
  if (const XmlElem* a = e->child("absorption")) read_color(a, m.absorption);
  s.materials.push_back(m);
  s.material_names.push_back(name ? name : "");
}

// LoadLight + *Light::Load (xmlload.cpp:202-243)
void load_light(rth_scene& s, const XmlElem* e) {
  const char* type = e->attr("type");
  rt_light l;
  std::memset(&l, 0, sizeof l);
  if (XmlElem::ieq(type, "ambient")) l.kind = RT_LIGHT_AMBIENT;
  else if (XmlElem::ieq(type, "direct")) l.kind = RT_LIGHT_DIRECT;
  else if (XmlElem::ieq(type, "point")) l.kind = RT_LIGHT_POINT;
  else {
    warn(s, "unknown light type %s", type ? type : "(null)");
    return;
  }
  if (const XmlElem* i = e->child("intensity")) read_color(i, l.intensity);
  if (l.kind == RT_LIGHT_DIRECT) {
    V3 d;
    read_vec3(e->child("direction"), d);
    d = normalized(d);
    l.direction[0] = d.x; l.direction[1] = d.y; l.direction[2] = d.z;
  } else if (l.kind == RT_LIGHT_POINT) {
    V3 p;
    read_vec3(e->child("position"), p);
    l.position[0] = p.x; l.position[1] = p.y; l.position[2] = p.z;
    read_float(e->child("size"), l.size);
    read_float(e->child("attenuation"), l.attenuation);
  }
  s.lights.push_back(l);
}

int get_mesh(rth_scene& s, const std::string& name) {
  for (size_t i = 0; i < s.mesh_names.size(); i++)
    if (s.mesh_names[i] == name) return (int)i;  // objList.Find(name) (xmlload.cpp:146)
  HostMesh m;
  std::string err;
  bool ok;
  if (name.compare(0, 4, "gen:") == 0) ok = generate_mesh(name.substr(4), m, err);
  else ok = load_obj(resolve_path(s, name), m, err);
  if (!ok) {
    warn(s, "cannot load mesh \"%s\": %s", name.c_str(), err.c_str());
    return -1;
  }
  s.meshes.push_back(std::move(m));
  s.mesh_names.push_back(name);
  return (int)s.meshes.size() - 1;
}

// LoadNode (xmlload.cpp:125-172)
void load_node(rth_scene& s, const XmlElem* e, int parent) {
  rt_node n;
  std::memset(&n, 0, sizeof n);
  n.parent = parent;
  n.prim_kind = RT_PRIM_NONE;
  n.mesh_id = -1;
  n.material_id = -1;
  const char* name = e->attr("name");
  const char* mtl = e->attr("material");
  const char* type = e->attr("type");
  int wants_objmtl = 0;
  if (type) {
    if (XmlElem::ieq(type, "sphere")) n.prim_kind = RT_PRIM_SPHERE;
    else if (XmlElem::ieq(type, "plane")) n.prim_kind = RT_PRIM_PLANE;
    else if (XmlElem::ieq(type, "obj")) {
      int mid = name ? get_mesh(s, name) : -1;
      if (mid >= 0) {
        n.prim_kind = RT_PRIM_MESH;
        n.mesh_id = mid;
        if (!mtl && !s.meshes[mid].mtls.empty()) wants_objmtl = 1;
      }
    } else warn(s, "unknown object type %s", type);
  }
  Xform x;
  x.load(e);
  x.tm.store(n.tm);
  x.itm.store(n.itm);
  int me = (int)s.nodes.size();
  s.nodes.push_back(n);
  s.node_names.push_back(name ? name : "");
  s.node_mtl_names.push_back(mtl ? mtl : "");
  s.node_wants_objmtl.push_back(wants_objmtl);
  for (auto& c : e->children)
    if (XmlElem::ieq(c->tag.c_str(), "object")) load_node(s, c.get(), me);
}

// SetNodeMaterials + CreateMultiMtl (xmlload.cpp:294-343).  mtlID is never set by the reference's
// triangle intersection, so a MultiMtl always shades with its sub-material 0: the
// flattened scene binds that sub-material directly; an empty MultiMtl shades white (materials.h:131).
void bind_materials(rth_scene& s) {
  for (size_t i = 0; i < s.nodes.size(); i++) {
    if (s.node_wants_objmtl[i]) {
      const std::string& nm = s.node_names[i];
      int found = -1;
      for (size_t k = 0; k < s.material_names.size(); k++)
        if (s.material_names[k] == nm) { found = (int)k; break; }
      if (found < 0) {
        const HostMesh& hm = s.meshes[s.nodes[i].mesh_id];
        rt_material m = default_blinn();
        const ObjMaterial& om = hm.mtls[0];
        for (int k = 0; k < 3; k++) { m.diffuse.value[k] = om.Kd[k]; m.specular.value[k] = om.Ks[k]; }
        m.glossiness.value = om.Ns;
        m.ior = om.Ni;
        auto img_map = [&](const std::string& file) {
          int img = get_image(s, file);
          if (img < 0) return -1;
          rt_texmap tm;
          std::memset(&tm, 0, sizeof tm);
          tm.kind = RT_TEX_IMAGE;
          tm.image = img;
          M34().store(tm.itm);
          s.texmaps.push_back(tm);
          return (int)s.texmaps.size() - 1;
        };
        if (!om.map_Kd.empty()) m.diffuse.map = img_map(om.map_Kd);
        if (!om.map_Ks.empty()) m.diffuse.map = img_map(om.map_Ks);  // sic: the reference sets the DIFFUSE texture (xmlload.cpp:331)
        if (om.illum > 2 && om.illum <= 7) {
          for (int k = 0; k < 3; k++) m.reflection.value[k] = om.Ks[k];
          if (!om.map_Ks.empty()) m.reflection.map = img_map(om.map_Ks);
          if (om.illum >= 6)
            for (int k = 0; k < 3; k++) m.refraction.value[k] = 1.0f - om.Tf[k];
        }
        s.materials.push_back(m);
        s.material_names.push_back(nm);
        found = (int)s.materials.size() - 1;
      }
      s.nodes[i].material_id = found;
    } else if (!s.node_mtl_names[i].empty()) {
      int found = -1;
      for (size_t k = 0; k < s.material_names.size(); k++)
        if (s.material_names[k] == s.node_mtl_names[i]) { found = (int)k; break; }  // Find = first match
      s.nodes[i].material_id = found;  // may stay -1 (null material in the reference)
    }
  }
}

rth_scene* finish(std::unique_ptr<rth_scene> s) {
  bind_materials(*s);
  s->mesh_descs.resize(s->meshes.size());
  for (size_t i = 0; i < s->meshes.size(); i++) s->meshes[i].fill(s->mesh_descs[i]);
  rt_scene_desc& d = s->desc;
  d.nodes = s->nodes.data(); d.n_nodes = (uint32_t)s->nodes.size();
  d.meshes = s->mesh_descs.data(); d.n_meshes = (uint32_t)s->mesh_descs.size();
  d.materials = s->materials.data(); d.n_materials = (uint32_t)s->materials.size();
  d.texmaps = s->texmaps.data(); d.n_texmaps = (uint32_t)s->texmaps.size();
  d.images = s->images.data(); d.n_images = (uint32_t)s->images.size();
  d.lights = s->lights.data(); d.n_lights = (uint32_t)s->lights.size();
  return s.release();
}

void init_scene(rth_scene& s) {
  std::memset(&s.desc, 0, sizeof s.desc);
  // TexturedColor() default: value 0, no map (scene.h:363)
  s.desc.background = {{0, 0, 0}, -1};
  s.desc.environment = {{0, 0, 0}, -1};
  rt_node root;
  std::memset(&root, 0, sizeof root);
  root.parent = -1; root.prim_kind = RT_PRIM_NONE; root.mesh_id = -1; root.material_id = -1;
  M34().store(root.tm); M34().store(root.itm);
  s.nodes.push_back(root);
  s.node_names.push_back("");
  s.node_mtl_names.push_back("");
  s.node_wants_objmtl.push_back(0);
  // Camera::Init (scene.h:453-464)
  rt_camera& c = s.desc.camera;
  c.pos[0] = c.pos[1] = c.pos[2] = 0;
  c.dir[0] = 0; c.dir[1] = 0; c.dir[2] = -1;
  c.up[0] = 0; c.up[1] = 1; c.up[2] = 0;
  c.fov = 40; c.focaldist = 1; c.dof = 0; c.width = 1920; c.height = 1080; c.srgb = 0;
}

// Camera::Load (xmlload.cpp:105-121)
void load_camera(rth_scene& s, const XmlElem* e) {
  rt_camera& c = s.desc.camera;
  V3 pos(0, 0, 0), dir(0, 0, -1), up(0, 1, 0);
  read_vec3(e->child("position"), pos);
  read_vec3(e->child("target"), dir);
  read_vec3(e->child("up"), up);
  read_float(e->child("fov"), c.fov);
  read_float(e->child("focaldist"), c.focaldist);
  read_float(e->child("dof"), c.dof);
  read_int(e->child("width"), c.width);
  read_int(e->child("height"), c.height);
  dir = dir - pos;
  dir = normalized(dir);
  V3 x = cross(dir, up);
  up = normalized(cross(x, dir));
  c.pos[0] = pos.x; c.pos[1] = pos.y; c.pos[2] = pos.z;
  c.dir[0] = dir.x; c.dir[1] = dir.y; c.dir[2] = dir.z;
  c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
  c.srgb = XmlElem::ieq(e->attr("gamma"), "sRGB") ? 1 : 0;
}

rth_scene* load_text(const std::string& text, const char* asset_dir) {
  XmlParser parser;
  std::string err;
  auto doc = parser.parse(text, err);
  if (!doc) { g_err = err; return nullptr; }
  const XmlElem* xml = doc->child("xml");
  if (!xml) { g_err = "No \"xml\" tag found."; return nullptr; }
  const XmlElem* xscene = xml->child("scene");
  if (!xscene) { g_err = "No \"scene\" tag found."; return nullptr; }
  const XmlElem* xcam = xml->child("camera");
  if (!xcam) { g_err = "No \"camera\" tag found."; return nullptr; }
  auto s = std::make_unique<rth_scene>();
  s->asset_dir = asset_dir ? asset_dir : "";
  init_scene(*s);
  // Scene::Load (xmlload.cpp:81-101)
  for (auto& c : xscene->children) {
    const char* t = c->tag.c_str();
    if (XmlElem::ieq(t, "object")) load_node(*s, c.get(), 0);
    else if (XmlElem::ieq(t, "light")) load_light(*s, c.get());
    else if (XmlElem::ieq(t, "material")) load_material(*s, c.get());
    else if (XmlElem::ieq(t, "background")) read_textured_color(*s, c.get(), s->desc.background);
    else if (XmlElem::ieq(t, "environment")) read_textured_color(*s, c.get(), s->desc.environment);
    else warn(*s, "Unknown tag \"%s\"", t);
  }
  load_camera(*s, xcam);
  return finish(std::move(s));
}

}  // namespace

static rth_scene* generate_stress(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height);
namespace {
struct BvhChoice {   // one call's choice of who builds the mesh BVHs; restores the thread's previous one
  int prev;
  explicit BvhChoice(uint32_t flags) : prev(get_host_bvh()) { set_host_bvh((flags & RTH_LOAD_NO_BVH) ? 0 : 1); }
  ~BvhChoice() { set_host_bvh(prev); }
};
template <class F>
rth_scene* guarded(F f) {   // nothing is thrown through the C boundary (bad_alloc in a BVH build of a huge mesh)
  try { return f(); } catch (const std::exception& e) { g_err = std::string("scene load failed: ") + e.what(); } catch (...) { g_err = "scene load failed"; }
  return nullptr;
}
}  // namespace

extern "C" {

rth_scene* rth_scene_load_string(const char* xml_text, const char* asset_dir) {
  if (!xml_text) { g_err = "null xml text"; return nullptr; }
  return guarded([&] { return load_text(xml_text, asset_dir); });
}

rth_scene* rth_scene_load_file(const char* path, const char* asset_dir) {
  if (!path) { g_err = "null path"; return nullptr; }
  std::ifstream f(path, std::ios::binary);
  if (!f) { g_err = std::string("Failed to load the file \"") + path + "\""; return nullptr; }
  std::stringstream ss;
  ss << f.rdbuf();
  std::string dir;
  if (asset_dir) dir = asset_dir;
  else {
    std::string p(path);
    size_t sl = p.find_last_of("/\\");
    dir = sl == std::string::npos ? "." : p.substr(0, sl);
  }
  return guarded([&] { return load_text(ss.str(), dir.c_str()); });
}

rth_scene* rth_scene_load_file_ex(const char* path, const char* asset_dir, uint32_t flags) {
  BvhChoice c(flags);
  return rth_scene_load_file(path, asset_dir);
}
rth_scene* rth_scene_load_string_ex(const char* xml_text, const char* asset_dir, uint32_t flags) {
  BvhChoice c(flags);
  return rth_scene_load_string(xml_text, asset_dir);
}
rth_scene* rth_scene_generate_stress_ex(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height, uint32_t flags) {
  BvhChoice c(flags);
  return rth_scene_generate_stress(seed, n_instances, subdiv, ground_res, width, height);
}

void rth_scene_free(rth_scene* s) { delete s; }
const char* rth_last_error(void) { return g_err.c_str(); }
const rt_scene_desc* rth_scene_desc(const rth_scene* s) { return s ? &s->desc : nullptr; }
const char* rth_scene_node_name(const rth_scene* s, int node) {
  if (!s || node < 0 || node >= (int)s->node_names.size()) return nullptr;
  return s->node_names[node].c_str();
}
int rth_scene_warnings(const rth_scene* s) { return s ? s->warnings : 0; }
void rth_set_host_bvh(int on) { set_host_bvh(on); }

// BASELINE config 5 stand-in: instanced icospheres baked into one world-space mesh + displaced ground.
rth_scene* rth_scene_generate_stress(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height) {
  return guarded([&] { return generate_stress(seed, n_instances, subdiv, ground_res, width, height); });
}
}  // extern "C"

static rth_scene* generate_stress(uint32_t seed, int n_instances, int subdiv, int ground_res, int width, int height) {
  auto s = std::make_unique<rth_scene>();
  init_scene(*s);
  HostMesh ico;
  make_icosphere(ico, subdiv);
  HostMesh world;
  world.name = "stress";
  auto rnd = [&](uint32_t i, uint32_t k) {
    uint32_t x = seed * 0x9e3779b1U + i * 0x85ebca77U + k * 0xc2b2ae3dU;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (x >> 8) * (1.0f / 16777216.0f);
  };
  for (int i = 0; i < n_instances; i++) {
    float r = 0.5f + 1.5f * rnd(i, 0);
    V3 c((rnd(i, 1) - 0.5f) * 400.0f, (rnd(i, 2) - 0.5f) * 400.0f, 10.0f + rnd(i, 3) * 380.0f);
    M34 m = M34::translation(c) * M34::scale(V3(r * 6, r * 6, r * 6));
    append_transformed(world, ico, m.c);
  }
  if (ground_res > 0) {
    HostMesh g;
    std::string err;
    generate_mesh("terrain?seed=" + std::to_string(seed) + "&tris=" + std::to_string(2L * ground_res * ground_res), g, err);
    M34 m = M34::scale(V3(256.0f, 256.0f, 60.0f));
    append_transformed(world, g, m.c);
  }
  world.compute_bounds();
  world.build_bvh(4);
  s->meshes.push_back(std::move(world));
  s->mesh_names.push_back("stress");
  rt_node n;
  std::memset(&n, 0, sizeof n);
  n.parent = 0; n.prim_kind = RT_PRIM_MESH; n.mesh_id = 0; n.material_id = 0;
  M34().store(n.tm); M34().store(n.itm);
  s->nodes.push_back(n);
  s->node_names.push_back("stress");
  s->node_mtl_names.push_back("");
  s->node_wants_objmtl.push_back(0);
  rt_material m = default_blinn();
  m.diffuse = {{0.7f, 0.7f, 0.65f}, -1};
  m.specular = {{0.2f, 0.2f, 0.2f}, -1};
  s->materials.push_back(m);
  s->material_names.push_back("stone");
  for (int k = 0; k < 2; k++) {
    rt_light l;
    std::memset(&l, 0, sizeof l);
    l.kind = RT_LIGHT_POINT;
    l.intensity[0] = l.intensity[1] = l.intensity[2] = 0.6f;
    l.position[0] = k ? 300.0f : -300.0f; l.position[1] = -300.0f; l.position[2] = 600.0f;
    l.size = 8.0f;
    s->lights.push_back(l);
  }
  s->desc.background = {{0.3f, 0.4f, 0.6f}, -1};
  s->desc.environment = {{0.3f, 0.4f, 0.6f}, -1};
  rt_camera& c = s->desc.camera;
  V3 pos(0, -700, 350), target(0, 0, 100), up(0, 0, 1);
  V3 dir = normalized(target - pos);
  V3 x = cross(dir, up);
  up = normalized(cross(x, dir));
  c.pos[0] = pos.x; c.pos[1] = pos.y; c.pos[2] = pos.z;
  c.dir[0] = dir.x; c.dir[1] = dir.y; c.dir[2] = dir.z;
  c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
  c.fov = 40; c.width = width; c.height = height;
  return finish(std::move(s));
}
