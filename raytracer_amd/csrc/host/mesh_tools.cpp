#include "mesh_tools.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <map>
#include <sstream>

#include "hmath.h"

namespace rth {

// ------------------------------------------------------------------ basics

void HostMesh::compute_bounds() {
  for (int k = 0; k < 3; k++) { bmin[k] = FLT_MAX; bmax[k] = -FLT_MAX; }
  for (size_t i = 0; i + 2 < pos.size(); i += 3)
    for (int k = 0; k < 3; k++) {
      bmin[k] = std::min(bmin[k], pos[i + k]);
      bmax[k] = std::max(bmax[k], pos[i + k]);
    }
  if (pos.empty())
    for (int k = 0; k < 3; k++) bmin[k] = bmax[k] = 0;
}

void HostMesh::compute_normals() {
  size_t nv = pos.size() / 3;
  nrm.assign(nv * 3, 0.0f);
  for (size_t f = 0; f < fpos.size(); f += 3) {
    uint32_t i0 = fpos[f], i1 = fpos[f + 1], i2 = fpos[f + 2];
    V3 a(pos[3 * i0], pos[3 * i0 + 1], pos[3 * i0 + 2]);
    V3 b(pos[3 * i1], pos[3 * i1 + 1], pos[3 * i1 + 2]);
    V3 c(pos[3 * i2], pos[3 * i2 + 1], pos[3 * i2 + 2]);
    V3 n = cross(b - a, c - a);  // area-weighted
    for (uint32_t i : {i0, i1, i2}) {
      nrm[3 * i] += n.x; nrm[3 * i + 1] += n.y; nrm[3 * i + 2] += n.z;
    }
  }
  for (size_t i = 0; i < nv; i++) {
    V3 n(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
    float l = length(n);
    if (l > 0) n = n / l; else n = V3(0, 0, 1);
    nrm[3 * i] = n.x; nrm[3 * i + 1] = n.y; nrm[3 * i + 2] = n.z;
  }
  fnrm = fpos;
}

void HostMesh::fill(rt_mesh& o) const {
  std::memset(&o, 0, sizeof o);
  o.positions = pos.data();
  o.normals = nrm.data();
  o.uvw = uvw.empty() ? nullptr : uvw.data();
  o.pos_idx = fpos.data();
  o.nrm_idx = fnrm.data();
  o.uvw_idx = fuvw.empty() ? nullptr : fuvw.data();
  o.n_positions = (uint32_t)(pos.size() / 3);
  o.n_normals = (uint32_t)(nrm.size() / 3);
  o.n_uvw = (uint32_t)(uvw.size() / 3);
  o.n_faces = n_faces();
  o.bvh = bvh.empty() ? nullptr : bvh.data();   // empty: rth_set_host_bvh(0), the renderer core builds the tree
  o.bvh_elements = elems.empty() ? nullptr : elems.data();
  o.n_bvh_nodes = (uint32_t)bvh.size();
  o.n_bvh_elements = (uint32_t)elems.size();
  for (int k = 0; k < 3; k++) { o.bound_min[k] = bmin[k]; o.bound_max[k] = bmax[k]; }
}

// ------------------------------------------------------------------ BVH (binned SAH, leaf <= 4)

namespace {

struct Box3 {
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  void grow(const float* p) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
  }
  void grow(const Box3& b) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Builder {
  const HostMesh& m;
  std::vector<Box3> tbox;
  std::vector<float> cen;  // 3 per tri
  std::vector<uint32_t> idx;
  int max_leaf, max_depth;

  Builder(const HostMesh& mesh, int ml, int md) : m(mesh), max_leaf(ml), max_depth(md) {
    uint32_t nf = m.n_faces();
    tbox.resize(nf);
    cen.resize(3 * (size_t)nf);
    idx.resize(nf);
    for (uint32_t f = 0; f < nf; f++) {
      idx[f] = f;
      for (int v = 0; v < 3; v++) tbox[f].grow(&m.pos[3 * (size_t)m.fpos[3 * (size_t)f + v]]);
      for (int k = 0; k < 3; k++) cen[3 * (size_t)f + k] = 0.5f * (tbox[f].lo[k] + tbox[f].hi[k]);
    }
  }

  static bool fits(size_t n, int levels_left, int max_leaf) {
    // can n primitives be stored under a subtree of at most `levels_left` further levels?
    if (levels_left >= 40) return true;
    double cap = (double)max_leaf * std::ldexp(1.0, levels_left);
    return (double)n <= cap;
  }

  // One subtree in depth-first pre-order, node and element references relative to its own arrays.  Large subtrees are
  // built by tasks of their own and spliced in behind their parent, which keeps the node numbering and the leaf order
  // (the traversal tie keys) exactly those of a serial depth-first build.
  struct Sub {
    std::vector<rt_bvh_node> nodes;
    std::vector<uint32_t> elems;
    int depth_seen = 0;
  };
  static void splice(Sub& dst, const Sub& src) {
    const uint32_t noff = (uint32_t)dst.nodes.size(), eoff = (uint32_t)dst.elems.size();
    for (rt_bvh_node n : src.nodes) {
      if (n.b & RT_BVH_LEAF_BIT) n.a += eoff;
      else { n.a += noff; n.b += noff; }
      dst.nodes.push_back(n);
    }
    dst.elems.insert(dst.elems.end(), src.elems.begin(), src.elems.end());
    dst.depth_seen = std::max(dst.depth_seen, src.depth_seen);
  }
  std::atomic<int> tasks_left{0};   // worker tasks that may still be started

  uint32_t build(Sub& out, size_t b, size_t e, int depth) {
    std::vector<rt_bvh_node>& nodes = out.nodes;
    std::vector<uint32_t>& elems = out.elems;
    int& depth_seen = out.depth_seen;
    depth_seen = std::max(depth_seen, depth);
    uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    Box3 box, cbox;
    for (size_t i = b; i < e; i++) {
      box.grow(tbox[idx[i]]);
      cbox.grow(&cen[3 * (size_t)idx[i]]);
    }
    for (int k = 0; k < 3; k++) { nodes[me].bmin[k] = box.lo[k]; nodes[me].bmax[k] = box.hi[k]; }
    size_t n = e - b;
    if (n <= (size_t)max_leaf) {
      nodes[me].a = (uint32_t)elems.size();
      nodes[me].b = RT_BVH_LEAF_BIT | (uint32_t)n;
      for (size_t i = b; i < e; i++) elems.push_back(idx[i]);
      return me;
    }
    int levels_left = max_depth - depth - 1;  // levels available below the children
    // binned SAH over the centroid bounds
    const int NB = 16;
    int best_axis = -1, best_bin = -1;
    float best_cost = FLT_MAX;
    for (int ax = 0; ax < 3; ax++) {
      float lo = cbox.lo[ax], ext = cbox.hi[ax] - cbox.lo[ax];
      if (!(ext > 0)) continue;
      Box3 bb[NB];
      size_t cnt[NB] = {0};
      float scale = NB / ext;
      for (size_t i = b; i < e; i++) {
        int bi = (int)((cen[3 * (size_t)idx[i] + ax] - lo) * scale);
        bi = std::min(std::max(bi, 0), NB - 1);
        bb[bi].grow(tbox[idx[i]]);
        cnt[bi]++;
      }
      float la[NB - 1]; size_t lc[NB - 1];
      Box3 acc; size_t c = 0;
      for (int i = 0; i < NB - 1; i++) { acc.grow(bb[i]); c += cnt[i]; la[i] = acc.area(); lc[i] = c; }
      Box3 racc; size_t rc = 0;
      for (int i = NB - 1; i > 0; i--) {
        racc.grow(bb[i]); rc += cnt[i];
        size_t l = lc[i - 1];
        if (l == 0 || rc == 0) continue;
        if (!fits(l, levels_left, max_leaf) || !fits(rc, levels_left, max_leaf)) continue;
        float cost = la[i - 1] * (float)l + racc.area() * (float)rc;
        if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = i; }
      }
    }
    size_t mid;
    if (best_axis >= 0) {
      float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - cbox.lo[best_axis];
      float scale = NB / ext;
      auto it = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t t) {
        int bi = (int)((cen[3 * (size_t)t + best_axis] - lo) * scale);
        bi = std::min(std::max(bi, 0), NB - 1);
        return bi < best_bin;
      });
      mid = (size_t)(it - idx.begin());
    } else {
      // degenerate centroids or depth budget: median split on the widest axis
      int ax = 0;
      float ex = -1;
      for (int k = 0; k < 3; k++) {
        float d = cbox.hi[k] - cbox.lo[k];
        if (d > ex) { ex = d; ax = k; }
      }
      mid = b + n / 2;
      std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e, [&](uint32_t x, uint32_t y) {
        float cx = cen[3 * (size_t)x + ax], cy = cen[3 * (size_t)y + ax];
        return cx < cy || (cx == cy && x < y);
      });
    }
    if (n >= 65536 && tasks_left.fetch_sub(1) > 0) {
      // the right half on a task of its own (disjoint range of idx, own output arrays), spliced in behind the left half
      Sub right;
      std::exception_ptr worker_error, own_error;   // an exception (bad_alloc on a huge mesh) fails the load, not the process
      std::thread worker([&] {
        try { build(right, mid, e, depth + 1); } catch (...) { worker_error = std::current_exception(); }
      });
      uint32_t l = 0;
      try { l = build(out, b, mid, depth + 1); } catch (...) { own_error = std::current_exception(); }
      worker.join();
      tasks_left.fetch_add(1);
      if (own_error) std::rethrow_exception(own_error);
      if (worker_error) std::rethrow_exception(worker_error);
      uint32_t r = (uint32_t)nodes.size();
      splice(out, right);
      nodes[me].a = l;
      nodes[me].b = r;
      return me;
    }
    if (n >= 65536) tasks_left.fetch_add(1);   // (undo the failed claim)
    uint32_t l = build(out, b, mid, depth + 1);
    uint32_t r = build(out, mid, e, depth + 1);
    nodes[me].a = l;
    nodes[me].b = r;
    return me;
  }
};

}  // namespace

// per thread: two threads loading scenes with different choices do not see each other's
static thread_local int g_host_bvh = 1;
void set_host_bvh(int on) { g_host_bvh = on ? 1 : 0; }
int get_host_bvh() { return g_host_bvh; }

void HostMesh::build_bvh(int max_leaf, int max_depth) {
  bvh.clear();
  elems.clear();
  bvh_depth = 0;
  if (!g_host_bvh) return;   // rt_scene_commit builds the tree on the device (rt_mesh::bvh == NULL)
  if (n_faces() == 0) {
    rt_bvh_node n;
    std::memset(&n, 0, sizeof n);
    n.b = RT_BVH_LEAF_BIT | 0u;
    bvh.push_back(n);
    return;
  }
  Builder bld(*this, max_leaf, max_depth);
  Builder::Sub root;
  root.nodes.reserve(n_faces());
  root.elems.reserve(n_faces());
  unsigned hw = std::thread::hardware_concurrency();
  if (const char* e = std::getenv("RTH_BUILD_THREADS")) hw = (unsigned)std::atoi(e);
  bld.tasks_left = (int)std::min(hw > 0 ? hw : 1u, 32u) - 1;
  bld.build(root, 0, n_faces(), 0);
  bvh.swap(root.nodes);
  elems.swap(root.elems);
  bvh_depth = root.depth_seen;
}

// ------------------------------------------------------------------ OBJ

static int fix_index(long i, size_t n) {
  if (i > 0) return (int)(i - 1);
  if (i < 0) return (int)((long)n + i);
  return -1;
}

static void load_mtl(const std::string& path, std::vector<ObjMaterial>& out) {
  std::ifstream f(path);
  if (!f) return;
  std::string line;
  ObjMaterial* cur = nullptr;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string k;
    if (!(ss >> k)) continue;
    if (k == "newmtl") {
      out.emplace_back();
      cur = &out.back();
      ss >> cur->name;
    } else if (!cur) {
      continue;
    } else if (k == "Kd") ss >> cur->Kd[0] >> cur->Kd[1] >> cur->Kd[2];
    else if (k == "Ks") ss >> cur->Ks[0] >> cur->Ks[1] >> cur->Ks[2];
    else if (k == "Tf") ss >> cur->Tf[0] >> cur->Tf[1] >> cur->Tf[2];
    else if (k == "Ns") ss >> cur->Ns;
    else if (k == "Ni") ss >> cur->Ni;
    else if (k == "illum") ss >> cur->illum;
    else if (k == "map_Kd") ss >> cur->map_Kd;
    else if (k == "map_Ks") ss >> cur->map_Ks;
  }
}

bool load_obj(const std::string& path, HostMesh& mesh, std::string& err) {
  std::ifstream f(path);
  if (!f) { err = "cannot open " + path; return false; }
  mesh = HostMesh();
  mesh.name = path;
  std::string dir;
  size_t sl = path.find_last_of("/\\");
  if (sl != std::string::npos) dir = path.substr(0, sl + 1);
  std::string line;
  bool any_vn = false, any_vt = false;
  while (std::getline(f, line)) {
    const char* s = line.c_str();
    while (*s == ' ' || *s == '\t') ++s;
    if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
      float x = 0, y = 0, z = 0;
      std::sscanf(s + 2, "%f %f %f", &x, &y, &z);
      mesh.pos.insert(mesh.pos.end(), {x, y, z});
    } else if (s[0] == 'v' && s[1] == 'n') {
      float x = 0, y = 0, z = 0;
      std::sscanf(s + 3, "%f %f %f", &x, &y, &z);
      mesh.nrm.insert(mesh.nrm.end(), {x, y, z});
    } else if (s[0] == 'v' && s[1] == 't') {
      float x = 0, y = 0, z = 0;
      std::sscanf(s + 3, "%f %f %f", &x, &y, &z);
      mesh.uvw.insert(mesh.uvw.end(), {x, y, z});
    } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
      std::vector<int> vp, vt, vn;
      const char* p = s + 2;
      while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (!*p) break;
        char* q;
        long a = std::strtol(p, &q, 10), b = 0, c = 0;
        if (q == p) break;
        p = q;
        if (*p == '/') {
          ++p;
          if (*p != '/') { b = std::strtol(p, &q, 10); p = q; }
          if (*p == '/') { ++p; c = std::strtol(p, &q, 10); p = q; }
        }
        vp.push_back(fix_index(a, mesh.pos.size() / 3));
        vt.push_back(fix_index(b, mesh.uvw.size() / 3));
        vn.push_back(fix_index(c, mesh.nrm.size() / 3));
      }
      for (size_t k = 2; k < vp.size(); k++) {
        size_t tri[3] = {0, k - 1, k};
        bool ok = true;
        for (size_t t : tri) if (vp[t] < 0) ok = false;
        if (!ok) continue;
        for (size_t t : tri) {
          mesh.fpos.push_back((uint32_t)vp[t]);
          mesh.fuvw.push_back(vt[t] >= 0 ? (uint32_t)vt[t] : 0u);
          mesh.fnrm.push_back(vn[t] >= 0 ? (uint32_t)vn[t] : 0u);
          if (vt[t] >= 0) any_vt = true;
          if (vn[t] >= 0) any_vn = true;
        }
      }
    } else if (!std::strncmp(s, "mtllib", 6)) {
      std::istringstream ss(s + 6);
      std::string name;
      ss >> name;
      load_mtl(dir + name, mesh.mtls);
    }
  }
  if (mesh.fpos.empty()) { err = "no faces in " + path; return false; }
  if (!any_vn || mesh.nrm.empty()) mesh.compute_normals();  // TriObj::Load: if (!HasNormals()) ComputeNormals()
  if (!any_vt || mesh.uvw.empty()) { mesh.uvw.clear(); mesh.fuvw.clear(); }
  mesh.compute_bounds();
  mesh.build_bvh(4);
  return true;
}

// ------------------------------------------------------------------ procedural meshes

namespace {

inline uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
inline float hash01(uint32_t a, uint32_t b, uint32_t c, uint32_t seed) {
  uint32_t h = hash_u32(a * 0x9e3779b1U ^ hash_u32(b * 0x85ebca77U ^ hash_u32(c * 0xc2b2ae3dU ^ seed)));
  return (h >> 8) * (1.0f / 16777216.0f);
}
// trilinear value noise in [0,1)
float vnoise(float x, float y, float z, uint32_t seed) {
  float fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
  int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  float tx = x - fx, ty = y - fy, tz = z - fz;
  tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty); tz = tz * tz * (3 - 2 * tz);
  float r = 0;
  for (int dz = 0; dz < 2; dz++)
    for (int dy = 0; dy < 2; dy++)
      for (int dx = 0; dx < 2; dx++) {
        float w = (dx ? tx : 1 - tx) * (dy ? ty : 1 - ty) * (dz ? tz : 1 - tz);
        r += w * hash01((uint32_t)(ix + dx), (uint32_t)(iy + dy), (uint32_t)(iz + dz), seed);
      }
  return r;
}
float fbm(float x, float y, float z, uint32_t seed) {
  float a = 0.5f, s = 0;
  for (int o = 0; o < 4; o++) {
    s += a * vnoise(x, y, z, seed + 101u * o);
    x *= 2.03f; y *= 2.03f; z *= 2.03f; a *= 0.5f;
  }
  return s;
}

// Adds a (nu x nv) parametric patch; f(u,v) -> position; wraps in u and/or v if requested.
template <class F>
void add_grid(HostMesh& m, int nu, int nv, bool wrap_u, bool wrap_v, F f) {
  uint32_t base = (uint32_t)(m.pos.size() / 3);
  int pu = wrap_u ? nu : nu + 1, pv = wrap_v ? nv : nv + 1;
  for (int j = 0; j < pv; j++)
    for (int i = 0; i < pu; i++) {
      float u = (float)i / nu, v = (float)j / nv;
      V3 p = f(u, v);
      m.pos.insert(m.pos.end(), {p.x, p.y, p.z});
      m.uvw.insert(m.uvw.end(), {u, v, 0.0f});
    }
  auto id = [&](int i, int j) { return base + (uint32_t)((j % pv) * pu + (i % pu)); };
  for (int j = 0; j < nv; j++)
    for (int i = 0; i < nu; i++) {
      uint32_t a = id(i, j), b = id(i + 1, j), c = id(i + 1, j + 1), d = id(i, j + 1);
      m.fpos.insert(m.fpos.end(), {a, b, c, a, c, d});
    }
}

void finish(HostMesh& m) {
  m.fuvw = m.fpos;
  m.compute_normals();
  m.compute_bounds();
  m.build_bvh(4);
}

// A lathe body + lid knob + torus handle + bent spout: concave, self-occluding, teapot-sized
// (fits roughly in [-16,16]^2 x [0,16] like the Utah teapot the reference scenes scale by 0.3..2.5).
void make_teapot_like(HostMesh& m, uint32_t seed, int tris) {
  // budget: body 60 %, lid 10 %, handle 15 %, spout 15 %
  auto res = [&](float frac, float aspect) {
    float q = std::sqrt(std::max(8.0f, tris * frac / 2.0f) / aspect);
    int nv = std::max(3, (int)q);
    int nu = std::max(4, (int)(q * aspect));
    return std::pair<int, int>(nu, nv);
  };
  float wob = 0.15f + 0.2f * hash01(1, 2, 3, seed);
  const float PI = 3.14159265358979f;
  {
    auto r = res(0.60f, 2.0f);
    add_grid(m, r.first, r.second, true, false, [&](float u, float v) {
      float a = 2 * PI * u;
      float h = 12.0f * v;                                                     // height
      float rad = 4.0f + 7.5f * std::sin(PI * std::pow(v, 0.85f)) * (1.0f + 0.04f * std::sin(6 * a) * wob);
      if (v < 0.02f) rad *= v / 0.02f;                                          // close the bottom
      return V3(rad * std::cos(a), rad * std::sin(a), h);
    });
  }
  {
    auto r = res(0.10f, 2.0f);
    add_grid(m, r.first, r.second, true, false, [&](float u, float v) {
      float a = 2 * PI * u;
      float rad = 4.2f * (1 - v) + 1.2f * std::sin(PI * v * 3.0f) * (v > 0.55f ? 1.0f : 0.2f);
      rad = std::max(rad, 0.0f);
      return V3(rad * std::cos(a), rad * std::sin(a), 12.0f + 3.5f * v);
    });
  }
  {
    auto r = res(0.15f, 3.0f);
    add_grid(m, r.first, r.second, true, true, [&](float u, float v) {      // torus handle in the xz plane
      float a = 2 * PI * u, b = 2 * PI * v;
      float R = 4.0f, rr = 0.9f;
      float cx = -11.5f, cz = 7.0f;
      float x = cx + (R + rr * std::cos(b)) * std::cos(a) * 1.2f;
      float z = cz + (R + rr * std::cos(b)) * std::sin(a);
      return V3(x, rr * std::sin(b), z);
    });
  }
  {
    auto r = res(0.15f, 0.5f);
    add_grid(m, r.first, r.second, true, false, [&](float u, float v) {     // bent tube spout
      float a = 2 * PI * u;
      float rr = 2.0f - 1.1f * v;
      V3 c(9.0f + 8.0f * v, 0.0f, 3.0f + 9.0f * v * v);
      V3 t = normalized(V3(8.0f, 0.0f, 18.0f * v + 1e-3f));
      V3 s(0, 1, 0);
      V3 n = cross(s, t);
      return c + s * (rr * std::cos(a)) + n * (rr * std::sin(a));
    });
  }
  finish(m);
}

void make_terrain(HostMesh& m, uint32_t seed, int res, float size, float height) {
  add_grid(m, res, res, false, false, [&](float u, float v) {
    float x = (u - 0.5f) * size, y = (v - 0.5f) * size;
    float h = height * (fbm(u * 8.0f, v * 8.0f, 0.5f, seed) - 0.5f);
    return V3(x, y, h);
  });
  finish(m);
}

// n unit cubes at seeded lattice points of a 48 x 40 x 18 block region, every face a textured quad with its own four
// vertices (flat normals, uv 0..1 per face): the shape of one per-material piece of a block world (the reference's
// scenes/minecraft_world/*.obj, absent from its checkout, are such pieces).
void make_voxels(HostMesh& m, uint32_t seed, int n) {
  static const int F[6][4][3] = {{{0, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 0, 0}}, {{0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}},
                                 {{0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}}, {{0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}},
                                 {{0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {0, 1, 0}}, {{1, 0, 0}, {1, 1, 0}, {1, 1, 1}, {1, 0, 1}}};
  static const float UV[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
  for (int c = 0; c < n; c++) {
    const float ox = std::floor(hash01((uint32_t)c, 1, 7, seed) * 48.0f) - 24.0f;
    const float oy = std::floor(hash01((uint32_t)c, 2, 7, seed) * 40.0f) - 30.0f;
    const float oz = std::floor(hash01((uint32_t)c, 3, 7, seed) * 18.0f) + 14.0f;
    for (int f = 0; f < 6; f++) {
      const uint32_t base = (uint32_t)(m.pos.size() / 3);
      for (int k = 0; k < 4; k++) {
        m.pos.insert(m.pos.end(), {ox + F[f][k][0], oy + F[f][k][1], oz + F[f][k][2]});
        m.uvw.insert(m.uvw.end(), {UV[k][0], UV[k][1], 0.0f});
      }
      m.fpos.insert(m.fpos.end(), {base, base + 1, base + 2, base, base + 2, base + 3});
    }
  }
  finish(m);
}

}  // namespace

void make_icosphere(HostMesh& m, int subdiv) {
  const float t = (1.0f + std::sqrt(5.0f)) * 0.5f;
  std::vector<V3> v = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t},
                       {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
  for (auto& p : v) p = normalized(p);
  std::vector<uint32_t> f = {0, 11, 5, 0, 5, 1, 0, 1, 7, 0, 7, 10, 0, 10, 11, 1, 5, 9, 5, 11, 4, 11, 10, 2, 10, 7, 6, 7, 1, 8,
                             3, 9, 4, 3, 4, 2, 3, 2, 6, 3, 6, 8, 3, 8, 9, 4, 9, 5, 2, 4, 11, 6, 2, 10, 8, 6, 7, 9, 8, 1};
  for (int s = 0; s < subdiv; s++) {
    std::map<uint64_t, uint32_t> mid;
    auto midpoint = [&](uint32_t a, uint32_t b) {
      uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = mid.find(key);
      if (it != mid.end()) return it->second;
      v.push_back(normalized((v[a] + v[b]) * 0.5f));
      uint32_t id = (uint32_t)v.size() - 1;
      mid[key] = id;
      return id;
    };
    std::vector<uint32_t> nf;
    nf.reserve(f.size() * 4);
    for (size_t i = 0; i < f.size(); i += 3) {
      uint32_t a = f[i], b = f[i + 1], c = f[i + 2];
      uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      nf.insert(nf.end(), {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca});
    }
    f.swap(nf);
  }
  m.pos.clear(); m.nrm.clear(); m.uvw.clear();
  const float PI = 3.14159265358979f;
  for (auto& p : v) {
    m.pos.insert(m.pos.end(), {p.x, p.y, p.z});
    m.nrm.insert(m.nrm.end(), {p.x, p.y, p.z});
    m.uvw.insert(m.uvw.end(), {std::atan2(p.y, p.x) / (2 * PI) + 0.5f, std::asin(p.z) / PI + 0.5f, 0.0f});
  }
  m.fpos = f; m.fnrm = f; m.fuvw = f;
}

void append_transformed(HostMesh& dst, const HostMesh& src, const float* c) {
  M34 tm;
  std::memcpy(tm.c, c, sizeof tm.c);
  M34 itm = tm.inverse();
  uint32_t bp = (uint32_t)(dst.pos.size() / 3), bn = (uint32_t)(dst.nrm.size() / 3), bt = (uint32_t)(dst.uvw.size() / 3);
  for (size_t i = 0; i + 2 < src.pos.size(); i += 3) {
    V3 p = tm.mul_point(V3(src.pos[i], src.pos[i + 1], src.pos[i + 2]));
    dst.pos.insert(dst.pos.end(), {p.x, p.y, p.z});
  }
  for (size_t i = 0; i + 2 < src.nrm.size(); i += 3) {
    V3 n(src.nrm[i], src.nrm[i + 1], src.nrm[i + 2]);
    // inverse-transpose
    V3 o(itm.c[0] * n.x + itm.c[1] * n.y + itm.c[2] * n.z, itm.c[3] * n.x + itm.c[4] * n.y + itm.c[5] * n.z,
         itm.c[6] * n.x + itm.c[7] * n.y + itm.c[8] * n.z);
    o = normalized(o);
    dst.nrm.insert(dst.nrm.end(), {o.x, o.y, o.z});
  }
  dst.uvw.insert(dst.uvw.end(), src.uvw.begin(), src.uvw.end());
  for (uint32_t i : src.fpos) dst.fpos.push_back(bp + i);
  for (uint32_t i : src.fnrm) dst.fnrm.push_back(bn + i);
  for (uint32_t i : src.fuvw) dst.fuvw.push_back(bt + i);
}

bool generate_mesh(const std::string& spec, HostMesh& mesh, std::string& err) {
  std::string kind = spec, query;
  size_t q = spec.find('?');
  if (q != std::string::npos) { kind = spec.substr(0, q); query = spec.substr(q + 1); }
  uint32_t seed = 1;
  long tris = 4096;
  float size = 1.0f;   // uniform scale of the generated positions (stand-ins for meshes of another native size)
  std::stringstream ss(query);
  std::string kv;
  while (std::getline(ss, kv, '&')) {
    size_t eq = kv.find('=');
    if (eq == std::string::npos) continue;
    std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
    if (k == "seed") seed = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
    else if (k == "tris") tris = std::strtol(v.c_str(), nullptr, 10);
    else if (k == "size") size = std::strtof(v.c_str(), nullptr);
  }
  if (tris < 8) tris = 8;
  mesh = HostMesh();
  mesh.name = "gen:" + spec;
  if (kind == "teapot_like") {
    make_teapot_like(mesh, seed, (int)tris);
  } else if (kind == "icosphere" || kind == "blob") {
    int sub = 0;
    while (20L * (1L << (2 * (sub + 1))) <= tris && sub < 9) sub++;
    make_icosphere(mesh, sub);
    if (kind == "blob") {
      for (size_t i = 0; i + 2 < mesh.pos.size(); i += 3) {
        float x = mesh.pos[i], y = mesh.pos[i + 1], z = mesh.pos[i + 2];
        float d = 1.0f + 0.6f * (fbm(1.7f * x + 3.1f, 1.7f * y + 1.3f, 1.7f * z + 2.2f, seed) - 0.5f);
        mesh.pos[i] = x * d; mesh.pos[i + 1] = y * d; mesh.pos[i + 2] = z * d;
      }
      mesh.compute_normals();
    }
    mesh.compute_bounds();
    mesh.build_bvh(4);
  } else if (kind == "terrain") {
    int res = std::max(2, (int)std::sqrt((double)tris / 2.0));
    make_terrain(mesh, seed, res, 2.0f, 0.35f);
  } else if (kind == "voxels") {
    make_voxels(mesh, seed, std::max(1, (int)(tris / 12)));
  } else {
    err = "unknown procedural mesh kind '" + kind + "'";
    return false;
  }
  if (size != 1.0f && size > 0.0f) {
    for (float& x : mesh.pos) x *= size;
    mesh.compute_bounds();
    mesh.build_bvh(4);
  }
  return true;
}

}  // namespace rth
