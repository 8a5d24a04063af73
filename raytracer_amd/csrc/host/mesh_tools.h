// Host-side triangle mesh container, OBJ reader, seeded procedural generators and BVH builder.
// Replaces what the reference gets from cyTriMesh / cyBVH through TriObj::Load
// (reference RayTracer/objects.h:60-67): LoadFromFileObj, ComputeNormals, ComputeBoundingBox,
// BVHTriMesh::SetMesh(this, 4).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "rtcore.h"

namespace rth {

struct ObjMaterial {  // subset of cyTriMesh::Mtl used by CreateMultiMtl (xmlload.cpp:319-343)
  std::string name;
  float Kd[3] = {1, 1, 1}, Ks[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};
  float Ns = 0, Ni = 1;
  int illum = 2;
  std::string map_Kd, map_Ks;
};

struct HostMesh {
  std::string name;
  std::vector<float> pos, nrm, uvw;          // xyz triples
  std::vector<uint32_t> fpos, fnrm, fuvw;    // 3 per face
  std::vector<rt_bvh_node> bvh;
  std::vector<uint32_t> elems;
  std::vector<ObjMaterial> mtls;             // from the .mtl file, if any
  float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  int bvh_depth = 0;

  uint32_t n_faces() const { return (uint32_t)(fpos.size() / 3); }
  void compute_normals();       // area-weighted vertex normals, one per position
  void compute_bounds();
  void build_bvh(int max_leaf = 4, int max_depth = 32);
  void fill(rt_mesh& out) const;
};

// 0: meshes are handed over without a tree (rt_mesh::bvh == NULL) and the renderer core builds it on the device.
// The choice is per calling thread; the loaders' *_ex entry points set it for one call and restore it.
void set_host_bvh(int on);
int get_host_bvh();

// Wavefront OBJ subset: v / vt / vn / f (polygons fan-triangulated, negative indices), mtllib/usemtl.
bool load_obj(const std::string& path, HostMesh& mesh, std::string& err);

// Seeded procedural meshes standing in for the reference's absent .obj assets.
// spec = "<kind>?seed=<n>&tris=<n>", kind in {teapot_like, icosphere, terrain, blob, voxels}; optional &size=<f> scales the positions.
bool generate_mesh(const std::string& spec, HostMesh& mesh, std::string& err);

void make_icosphere(HostMesh& m, int subdiv);
void append_transformed(HostMesh& dst, const HostMesh& src, const float* m34 /*column-major 3x4*/);

}  // namespace rth
