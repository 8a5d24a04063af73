// rtcore — C-ABI implementation and host orchestration of the MI355X renderer core (include/rtcore.h).
//
// Built by hipcc for gfx950 into raytracer_amd/lib/librtcore.so.  (With -DRT_EMU and g++ the same file builds the
// single-threaded host emulation used by CPU-side unit tests only; see rt_platform.h.)
#include "rt_kernels.h"
#include "rt_photon.h"
#include "rt_photon_build.h"
#include "rt_bvh_build.h"
#include "rt_denoise.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "rtcore.h"
#ifndef RT_EMU
#include <dlfcn.h>
#include "rt_sort.h"
#endif

// ------------------------------------------------------------------------------------------------ helpers
#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
      return RT_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

#ifndef RT_EMU
#define LAUNCH_NL(kern, n_items, cap_, lds_, ...)                                                \
  do {                                                                                           \
    unsigned long long ni_ = (unsigned long long)(n_items);                                      \
    unsigned long long nb_ = (ni_ + RT_BLOCK - 1) / RT_BLOCK;                                    \
    if (nb_ < 1) nb_ = 1;                                                                        \
    if (nb_ > (unsigned long long)(cap_)) nb_ = (cap_);                                          \
    hipLaunchKernelGGL(kern, dim3((unsigned)nb_), dim3(RT_BLOCK), (lds_), ctx->stream, __VA_ARGS__); \
  } while (0)
#define LAUNCH_N(kern, n_items, cap_, ...) LAUNCH_NL(kern, n_items, cap_, 0, __VA_ARGS__)
// streaming kernels (raygen, shade, accumulate, ...) and traversal kernels (LDS stack, cache-bound) get their own
// persistent grid sizes
#define LAUNCH(kern, n_items, ...) LAUNCH_N(kern, n_items, ctx->max_blocks, __VA_ARGS__)
#define LAUNCH_T(kern, n_items, ...) LAUNCH_NL(kern, n_items, ctx->max_blocks_trace, ctx->trace_lds, __VA_ARGS__)
// first-phase trace kernels (instance table only, no BVH stack): streaming grid (measured: k_trace_closest 2.18 -> 1.92 ms)
#define LAUNCH_P1(kern, n_items, ...) LAUNCH_N(kern, n_items, ctx->max_blocks, __VA_ARGS__)
#else
#define LAUNCH(kern, n_items, ...) kern(__VA_ARGS__)
#define LAUNCH_T(kern, n_items, ...) kern(__VA_ARGS__)
#define LAUNCH_P1(kern, n_items, ...) kern(__VA_ARGS__)
#endif

#ifndef RT_BLOCKS_PER_CU
#define RT_BLOCKS_PER_CU 8   // streaming kernels: measured 4 -> 8 blocks per CU: k_shade 5.8 -> 4.8 ms per bench step
#endif
#ifndef RT_SHADOW_TLAS_MIN
#define RT_SHADOW_TLAS_MIN 256   // instances from which the two-phase shadow kernels walk the instance BVH, too (below it: the table walk)
#endif
#ifndef RT_CONE_MAX_INST
#define RT_CONE_MAX_INST 256   // soft-shadow proofs (cone_clear) walk the instance table linearly: ~25 instructions per entry and group,
                               // against the 16 rays through the instance BVH (~300 each) a proof may save
#endif
#ifndef RT_TLAS_MIN
#define RT_TLAS_MIN 80   // instances from which the instance BVH replaces the linear table walk.  24 until the first trace phase was
                         // fused with shading (table walks only): finalProject.xml, 70 instances, raygen + first phase + shading of
                         // their rays 4.09 ms with the BVH, 3.51 ms with the table; ~0.023 ms per instance puts the crossover near 95
#endif
#ifndef RT_SOFT_CHUNK
#define RT_SOFT_CHUNK 32   // soft-shadow samples per launch of the second sample phase when the scene has meshes (run_pass)
#endif
#ifndef RT_SHADOW_HINT_DEPTH
#define RT_SHADOW_HINT_DEPTH 19   // mesh trees at least this deep (~1 M triangles) walk their soft-shadow rays with occluder hints
#endif
#ifndef RT_TRACE_BLOCKS_PER_CU
#define RT_TRACE_BLOCKS_PER_CU 6   // traversal kernels: at most this many blocks per CU, fewer when the scene's stack needs the LDS
#endif

namespace {

struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

template <class T>
T* dptr(const DevMem& m) { return (T*)m.p; }

struct MeshInfo { uint32_t n_nodes = 0, n_slots = 0; };   // inner nodes and triangle slots a committed mesh uses

}  // namespace

// ---- rt_denoise_noalloc: the filter's working images in a fixed device array instead of allocated buffers.
// RT_NOALLOC_MAX_PIXELS pixels x 4 RGBA float images (ping, pong, albedo guide, normal guide) = 64 MiB, held by every GPU the
// library runs on until the process ends.  In host code the array's NAME is only a stand-in for the symbol: its device
// address on a given GPU comes from hipGetSymbolAddress (rt_context_create, with that device current) and is what the
// kernels and copies are given.
#define RT_NOALLOC_MAX_PIXELS (1u << 20)
#define RT_NOALLOC_BYTES ((size_t)RT_NOALLOC_MAX_PIXELS * 4u * 16u)
#ifndef RT_EMU
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
#else
alignas(256) static unsigned char scratch[RT_NOALLOC_BYTES];
#endif

// Developer knobs (tests force both sides of a threshold with them).  Read from the environment ONCE, when the context is
// created, and settable per context through rt_set_option; nothing on the render path calls getenv.
struct Options {
  double tlas_min = RT_TLAS_MIN;   // "tlas_min"     RT_TLAS_MIN      instances from which the instance BVH is used (closest-hit rays)
  double shadow_tlas_min = RT_SHADOW_TLAS_MIN;   // "shadow_tlas_min"  ... and from which the two-phase shadow kernels use it (never below tlas_min)
  double gather_hints = 1;         // "gather_hints" RT_GATHER_HINTS  0: every photon lookup from scratch
  double frame_gb = 0;             // "frame_gb"     RT_FRAME_GB      stage-queue memory of one pass (0: frame_mem_mb / 32 GB)
  double fan_scale = 1;            // "fan_scale"    RT_TEST_FAN_SCALE  scales the measured queue sizing (tests: forces overflow)
  double timing = 0;               // "timing"       RT_TIMING        host-side phases of the photon build on stderr
  double soft_clear = 1;           // "soft_clear"   RT_SOFT_CLEAR    0: trace every soft-shadow group (tests compare both)
  double fuse = 1;                 // "fuse"         RT_FUSE          1: first trace phase + shading in one kernel (k_raygen_shade / k_trace_shade);
                                   //                                  0: the separate k_raygen_trace / k_trace_closest / k_shade launches
  double shadow_hints = -1;        // "shadow_hints" RT_SHADOW_HINTS   soft-shadow mesh walks head for the previous ray's occluder first: 1 on, 0 off,
                                   //                                  -1 by the depth of the scene's mesh trees (>= RT_SHADOW_HINT_DEPTH)
  double restart_min_depth = RT_SHADOW_HINT_DEPTH;   // "restart_min_depth"  mesh trees at least this deep get the table that lets a hinted shadow walk
                                                     //                      start around its hint (k_shadow_mesh); takes effect at the next rt_scene_commit
  double summary_mapped = 1;       // "summary_mapped" RT_SUMMARY_MAPPED  1: the per-pass summary is written into pinned host memory by the kernel itself
  double cone_max_inst = RT_CONE_MAX_INST;   // "cone_max_inst"  instances up to which the soft-shadow proofs walk the instance table; beyond it
                                             //                  (with an instance BVH) every group's cone walks the BVH (tests force both)
};

// grow-only device buffers the entry points reuse from call to call (no hipMalloc / hipFree in the steady state)
enum { SC_ACCUM = 0, SC_RES_RGB, SC_RES_RAD, SC_RES_CNT, SC_API_A, SC_API_B, SC_API_C, SC_DN_RGB, SC_DN_A, SC_DN_B, SC_DN_ALB, SC_DN_NRM,
       SC_PH, SC_KD, SC_REDUCE, SC_COUNT };

struct rt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  Options opt;
  unsigned char* noalloc_ptr = nullptr;   // this GPU's copy of the file-scope `scratch` array (hipGetSymbolAddress)
  DevMem bufs[SC_COUNT];
  uint64_t n_allocs = 0;   // hipMalloc calls made on behalf of this context (rt_debug_alloc_count)
  int max_blocks = 2048, max_blocks_trace = 1024;
  int max_mesh_depth = 0;   // deepest mesh BVH of the committed scene
  int n_cus = 256;
  size_t trace_lds = (size_t)RT_STACK * RT_BLOCK * 4;   // dynamic LDS of the traversal kernels: stack entries of the scene x block
  bool has_scene = false;

  // committed scene (device)
  std::vector<DevMem> scene_mem;
  DScene ds;
  rt_camera camera;
  int n_nodes = 0, n_hard_lights = 0, n_soft_capable = 0, n_gloss = 0, n_minst = 0;
  std::vector<rt_light> lights;
  std::vector<DInstance> h_inst;
  std::vector<DMesh> h_meshes;      // as committed (rt_scene_mesh_bvh)
  std::vector<MeshInfo> h_minfo;
  std::vector<rt_material> h_materials;

  // frame resources
  std::vector<DevMem> frame_mem;
  size_t frame_key[16] = {0};
  RayQ Q[2];
  RayQ QM;     // rays waiting for the mesh phase (entry i belongs to work item i of MQ): k_trace_mesh writes their hit, k_shade shades them
  GroupQ GH, GS;
  GatherQ GQ;
  MeshQ MQ, MH, MS;
  uint32_t* d_list2 = nullptr;
  uint32_t* d_list1 = nullptr;   // soft-shadow groups that have to be traced (k_shadow_soft_clear)
  uint32_t* d_gather_cursor = nullptr;  // GatherP::cursor, one slab per stage
  uint32_t* d_gather_retry = nullptr;   // GatherP::retry, retry_seed
  float* d_gather_seed = nullptr;
  uint32_t* d_counts = nullptr;  // [levels][8] + overflow + misc
  uint32_t* d_summary = nullptr; // k_pass_summary's output (allocated with the context)
  uint32_t* h_summary = nullptr; // ... and its pinned host copy
  uint32_t* h_summary_dev = nullptr;  // the device's address of that pinned page, when the kernel can write it directly (no copy)
  float* d_path_rad = nullptr;      // three planes of rad_stride floats
  uint32_t rad_stride = 0;
  uint32_t* d_pixels = nullptr;
  uint32_t* d_pixels2 = nullptr;
  int32_t* d_hit_node = nullptr;
  int32_t* d_hit_face = nullptr;
  float *d_aov_depth = nullptr, *d_aov_normal = nullptr, *d_aov_albedo = nullptr;
  DCounters* d_counters = nullptr;
  uint32_t path_budget = 0;
  // queue sizing: entries per camera sample in the fullest segment, by queue kind (measured by a pilot pass)
  double fan[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // see FramePlan::f
  size_t fan_key[10] = {0};
  bool fan_valid = false;
  std::vector<uint32_t> h_pixels;
  size_t pixel_key[6] = {0};

  // photon maps
  PhotonState photons;
  float* d_gather_hints = nullptr;   // [W * H][2], see GatherP::hints; lives as long as the frame size
  size_t gather_hint_px = 0;
  bool gather_hints_cold = true;     // nothing in the table yet: the next pass runs its pilot lookups first

  bool counters_on = false, profile_on = false;
  uint64_t host_paths = 0;
  rt_profile prof;
  struct Ev { hipEvent_t a, b; int kind; uint64_t items_slot; };
  std::vector<hipEvent_t> ev_pool;
  std::vector<Ev> ev_used;

  // async render
  std::thread worker;
  std::atomic<long long> pixels_done{0};
  std::atomic<int> finished{1};
  int async_status = RT_OK;
};

namespace {

int alloc_dev(rt_context* ctx, std::vector<DevMem>& pool, size_t bytes, void** out) {
  DevMem m;
  if (bytes == 0) bytes = 16;
  HIP_TRY(hipMalloc(&m.p, bytes));
  ctx->n_allocs++;
  m.bytes = bytes;
  pool.push_back(m);
  *out = m.p;
  return RT_OK;
}
// The context's grow-only buffer `slot`, at least `bytes` large.  (Callers have synchronised the stream before a buffer
// can be replaced: every entry point that uses one ends with a stream synchronisation.)
int grow_dev(rt_context* ctx, int slot, size_t bytes, void** out) {
  DevMem& m = ctx->bufs[slot];
  if (bytes == 0) bytes = 16;
  if (!m.p || m.bytes < bytes) {
    m.release();
    size_t want = bytes + bytes / 8;   // headroom: a slightly larger batch next time does not reallocate
    HIP_TRY(hipMalloc(&m.p, want));
    ctx->n_allocs++;
    m.bytes = want;
  }
  *out = m.p;
  return RT_OK;
}
// An asynchronous render (rt_render_begin) owns the context's stream and frame resources until it has finished: every
// other entry point waits for it first.
void quiesce(rt_context* ctx) {
  if (ctx->worker.joinable()) ctx->worker.join();
}
template <class T>
int upload(rt_context* ctx, std::vector<DevMem>& pool, const std::vector<T>& v, const T** out) {
  void* p;
  int rc = alloc_dev(ctx, pool, v.size() * sizeof(T), &p);
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T*)p;
  return RT_OK;
}
void free_pool(std::vector<DevMem>& pool) {
  for (auto& m : pool) m.release();
  pool.clear();
}

float host_u2f(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

float halton_ref(int index, int base) {  // rng.h:103-112
  float r = 0;
  float f = 1.0f / (float)base;
  for (int i = index; i > 0; i /= base) {
    r += f * (i % base);
    f /= (float)base;
  }
  return r;
}

DTexColor conv_tc(const rt_tex_color& t) {
  DTexColor d;
  d.v[0] = t.value[0]; d.v[1] = t.value[1]; d.v[2] = t.value[2];
  d.map = t.map;
  return d;
}
DMaterial conv_mtl(const rt_material& m) {
  DMaterial d;
  d.kind = m.kind;
  d.diffuse = conv_tc(m.diffuse); d.specular = conv_tc(m.specular); d.emission = conv_tc(m.emission);
  d.reflection = conv_tc(m.reflection); d.refraction = conv_tc(m.refraction);
  d.glossiness.v = m.glossiness.value; d.glossiness.map = m.glossiness.map;
  d.absorption[0] = m.absorption[0]; d.absorption[1] = m.absorption[1]; d.absorption[2] = m.absorption[2];
  d.ior = m.ior;
  return d;
}
rt_material default_material() {  // MtlBasePhongBlinn defaults (materials.h:53-60), SURVEY Q16
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

// Converts one rt_mesh BVH (32-byte nodes, leaf element lists) to the device layout: 64-byte internal nodes that
// hold both child boxes (one fetch per visit) and 48-byte triangle slots (v0, e1, e2 + tie key + face id) stored
// in depth-first leaf order.
struct MeshConv {
  const rt_mesh& m;
  std::vector<float4>& nodes;
  std::vector<float4>& tris;
  size_t node_base, tri_base;
  uint32_t n_internal = 0, n_slots = 0, n_leaves = 0;
  size_t visited = 0;
  int max_depth = 0;
  std::string err;
  std::vector<uint32_t> anc;   // per block of 2^RT_ANC_SHIFT slots: the smallest inner node whose slot range holds the block (DScene::bvh_anc)
  bool nested = true;          // every node's box holds its children's boxes (what starting a walk below the root relies on)
  MeshConv(const rt_mesh& mm, std::vector<float4>& n, std::vector<float4>& t) : m(mm), nodes(n), tris(t) {
    node_base = n.size();
    tri_base = t.size();
  }
  bool convert(uint32_t idx, int depth, uint32_t& ref) {
    if (idx >= m.n_bvh_nodes) { err = "BVH child index out of range"; return false; }
    if (++visited > m.n_bvh_nodes) { err = "BVH is not a tree"; return false; }
    if (depth > RT_STACK) { err = "BVH deeper than 32 levels"; return false; }
    max_depth = std::max(max_depth, depth);
    const rt_bvh_node& n = m.bvh[idx];
    if (n.b & RT_BVH_LEAF_BIT) {
      uint32_t count = n.b & ~RT_BVH_LEAF_BIT;
      if (count > 7) { err = "BVH leaf with more than 7 elements"; return false; }
      if ((size_t)n.a + count > m.n_bvh_elements) { err = "BVH leaf elements out of range"; return false; }
      uint32_t first = n_slots;
      for (uint32_t s = 0; s < count; s++) {
        uint32_t face = m.bvh_elements[n.a + s];
        if (face >= m.n_faces) { err = "BVH element is not a face"; return false; }
        const uint32_t* f = &m.pos_idx[3 * (size_t)face];
        const float* v0 = &m.positions[3 * (size_t)f[0]];
        const float* v1 = &m.positions[3 * (size_t)f[1]];
        const float* v2 = &m.positions[3 * (size_t)f[2]];
        uint32_t key = n_leaves * 8u + (7u - s);
        float4 a = make_float4(v0[0], v0[1], v0[2], host_u2f(key));
        float4 b = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], host_u2f(face));  // edge1, objects.cpp:203
        float4 c = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.0f);                     // edge2, objects.cpp:204
        tris.push_back(a); tris.push_back(b); tris.push_back(c);
        n_slots++;
      }
      n_leaves++;
      if (first >= (1u << 28)) { err = "mesh too large"; return false; }
      ref = RT_LEAF | (first << 3) | count;
      return true;
    }
    uint32_t me = n_internal++;
    size_t at = node_base + 4 * (size_t)me;
    nodes.resize(std::max(nodes.size(), at + 4));
    if (n.a >= m.n_bvh_nodes || n.b >= m.n_bvh_nodes) { err = "BVH child index out of range"; return false; }
    const rt_bvh_node& l = m.bvh[n.a];
    const rt_bvh_node& r = m.bvh[n.b];
    uint32_t lref, rref;
    const uint32_t first_slot = n_slots;
    if (!convert(n.a, depth + 1, lref)) return false;
    const uint32_t right_first = n_slots;   // the children cover consecutive slot ranges: the right one's begins here (occluder hints)
    if (!convert(n.b, depth + 1, rref)) return false;
    for (int k = 0; k < 3; k++)
      if (!(l.bmin[k] >= n.bmin[k] && l.bmax[k] <= n.bmax[k] && r.bmin[k] >= n.bmin[k] && r.bmax[k] <= n.bmax[k])) nested = false;
    // blocks that lie in this node's slot range and in neither child's alone (the children came first and took theirs)
    for (uint32_t b = (first_slot + (1u << RT_ANC_SHIFT) - 1) >> RT_ANC_SHIFT; ((b + 1) << RT_ANC_SHIFT) <= n_slots; b++) {
      if (anc.size() <= b) anc.resize(b + 1, RT_NONE);
      if (anc[b] == RT_NONE) anc[b] = me;
    }
    // per axis: (left min, right min, left max, right max) — the two children's slab tests run as packed FP32 pairs
    nodes[at + 0] = make_float4(l.bmin[0], r.bmin[0], l.bmax[0], r.bmax[0]);
    nodes[at + 1] = make_float4(l.bmin[1], r.bmin[1], l.bmax[1], r.bmax[1]);
    nodes[at + 2] = make_float4(l.bmin[2], r.bmin[2], l.bmax[2], r.bmax[2]);
    nodes[at + 3] = make_float4(host_u2f(lref), host_u2f(rref), host_u2f(right_first), 0.0f);
    ref = me;
    return true;
  }
};

// BVH over the instances' padded world boxes (median split of the box centres along their widest axis, leaves of at
// most 2 instances), written in the 64-byte paired node layout of the mesh BVHs.
struct TlasBuild {
  const std::vector<DInstance>& inst;
  std::vector<float4> nodes;
  std::vector<int32_t> order;
  int max_depth = 0;
  explicit TlasBuild(const std::vector<DInstance>& i) : inst(i) {}
  void bounds(const int32_t* ids, size_t n, float* lo, float* hi) const {
    for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
    for (size_t j = 0; j < n; j++)
      for (int k = 0; k < 3; k++) {
        lo[k] = std::min(lo[k], inst[ids[j]].wmin[k]);
        hi[k] = std::max(hi[k], inst[ids[j]].wmax[k]);
      }
  }
  uint32_t build(int32_t* ids, size_t n, int depth) {
    max_depth = std::max(max_depth, depth);
    if (n <= 2) {
      uint32_t first = (uint32_t)order.size();
      order.insert(order.end(), ids, ids + n);
      return RT_LEAF | (first << 3) | (uint32_t)n;
    }
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t j = 0; j < n; j++)
      for (int k = 0; k < 3; k++) {
        float c = 0.5f * (inst[ids[j]].wmin[k] + inst[ids[j]].wmax[k]);
        clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
      }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    size_t mid = n / 2;
    std::nth_element(ids, ids + mid, ids + n, [&](int32_t a, int32_t b) {
      float ca = inst[a].wmin[axis] + inst[a].wmax[axis], cb = inst[b].wmin[axis] + inst[b].wmax[axis];
      return ca < cb || (ca == cb && a < b);
    });
    uint32_t me = (uint32_t)(nodes.size() / 4);
    nodes.resize(nodes.size() + 4);
    float llo[3], lhi[3], rlo[3], rhi[3];
    bounds(ids, mid, llo, lhi);
    bounds(ids + mid, n - mid, rlo, rhi);
    uint32_t l = build(ids, mid, depth + 1);
    uint32_t r = build(ids + mid, n - mid, depth + 1);
    nodes[4 * (size_t)me + 0] = make_float4(llo[0], rlo[0], lhi[0], rhi[0]);
    nodes[4 * (size_t)me + 1] = make_float4(llo[1], rlo[1], lhi[1], rhi[1]);
    nodes[4 * (size_t)me + 2] = make_float4(llo[2], rlo[2], lhi[2], rhi[2]);
    nodes[4 * (size_t)me + 3] = make_float4(host_u2f(l), host_u2f(r), 0.0f, 0.0f);
    return me;
  }
};

int fail(rt_context* ctx, int code, const std::string& msg) {
  ctx->err = msg;
  return code;
}

// One slab of device memory cut into 256-byte aligned pieces.
struct Carver {
  unsigned char* base = nullptr;
  size_t used = 0;
  template <class T>
  T* take(size_t count) {
    T* p = base ? (T*)(base + used) : nullptr;
    used += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
  }
};

// stable sort of (key, value) pairs on bits [0, end_bit) of the key
int sort_pairs(rt_context* ctx, void* tmp, size_t tmp_bytes, const unsigned long long* kin, unsigned long long* kout, const uint32_t* vin,
               uint32_t* vout, size_t n, int end_bit) {
#ifndef RT_EMU
  if (rt_sort_pairs_u64(tmp, &tmp_bytes, kin, kout, vin, vout, n, end_bit, (void*)ctx->stream) != 0) return fail(ctx, RT_ERR_DEVICE, "device sort failed");
#else
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
  const unsigned long long mask = end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1ull);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (kin[a] & mask) < (kin[b] & mask); });
  for (size_t i = 0; i < n; i++) { kout[i] = kin[order[i]]; vout[i] = vin[order[i]]; }
#endif
  return RT_OK;
}


// BVH of a mesh that came without one, built on the device into its ranges of the node / slot arrays (rt_bvh_build.h).
// `depth` = depth of the deepest leaf (root = 0), as MeshConv::max_depth.
int lbvh_build(rt_context* ctx, const rt_mesh& m, float4* nodes, float4* tris, DMesh& D, MeshInfo& info, int& depth) {
  const uint32_t n = m.n_faces;
  depth = 0;
  info.n_nodes = 0; info.n_slots = n;
  D.root_ref = RT_LEAF;
  for (int k = 0; k < 6; k++) D.bounds[k] = 0.0f;
  if (n == 0) return RT_OK;
  const int max_levels = RT_STACK - 4;   // inner nodes on levels 0 .. 27, leaves no deeper than 28
  size_t sort_tmp = 0;
#ifndef RT_EMU
  if (rt_sort_pairs_u64(nullptr, &sort_tmp, nullptr, nullptr, nullptr, nullptr, n, 63, nullptr) != 0) return fail(ctx, RT_ERR_DEVICE, "device sort sizing failed");
#endif
  Carver cv;
  float *pos = nullptr, *out = nullptr;
  uint32_t *fidx = nullptr, *vals_in = nullptr, *vals_out = nullptr, *total = nullptr;
  unsigned long long *keys_in = nullptr, *keys_out = nullptr;
  float4* pbox = nullptr;
  uint2 *work = nullptr, *lv = nullptr;
  uint8_t* tmp = nullptr;
  DevMem slab;
  for (int pass = 0; pass < 2; pass++) {
    cv.used = 0;
    pos = cv.take<float>(3 * (size_t)m.n_positions); fidx = cv.take<uint32_t>(3 * (size_t)n);
    keys_in = cv.take<unsigned long long>(n); keys_out = cv.take<unsigned long long>(n);
    vals_in = cv.take<uint32_t>(n); vals_out = cv.take<uint32_t>(n);
    pbox = cv.take<float4>(2 * (size_t)n);
    work = cv.take<uint2>(n);
    lv = cv.take<uint2>(max_levels + 2);
    total = cv.take<uint32_t>(4);
    out = cv.take<float>(8);
    tmp = cv.take<uint8_t>(sort_tmp + 16);
    if (pass == 0) {
      HIP_TRY(hipMalloc(&slab.p, cv.used));
      ctx->n_allocs++;
      cv.base = (unsigned char*)slab.p;
    }
  }
  struct Release { DevMem& m; ~Release() { m.release(); } } release{slab};
  HIP_TRY(hipMemcpyAsync(pos, m.positions, 3 * (size_t)m.n_positions * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(fidx, m.pos_idx, 3 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  LbvhArgs A;
  std::memset(&A, 0, sizeof A);
  A.pos = pos; A.fidx = fidx; A.n = n;
  for (int k = 0; k < 3; k++) {
    const float ext = m.bound_max[k] - m.bound_min[k];
    A.lo[k] = m.bound_min[k];
    A.scale[k] = ext > 0.0f ? 2097152.0f / ext : 0.0f;
  }
  A.keys = keys_in; A.vals = vals_in; A.tris = tris; A.nodes = nodes; A.pbox = pbox; A.work = work; A.lv = lv; A.total = total;
  A.max_levels = max_levels;
  LAUNCH(k_lbvh_keys, n, A);
  int rc = sort_pairs(ctx, tmp, sort_tmp, keys_in, keys_out, vals_in, vals_out, n, 63);
  if (rc) return rc;
  A.keys = keys_out; A.vals = vals_out;
  LAUNCH(k_lbvh_slots, n, A);
  std::vector<uint2> h_lv(max_levels + 1);
  for (auto& l : h_lv) l.x = l.y = 0;
  uint32_t root_ref = RT_LEAF | n;
  if (n > 4) {
    root_ref = 0;
    const uint2 root_run = [&] { uint2 r; r.x = 0; r.y = n; return r; }();
    const uint2 lv0 = [&] { uint2 r; r.x = 0; r.y = 1; return r; }();
    const uint32_t one = 1;
    HIP_TRY(hipMemcpyAsync(work, &root_run, sizeof root_run, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(lv, &lv0, sizeof lv0, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(total, &one, sizeof one, hipMemcpyHostToDevice, ctx->stream));
    for (int level = 0; level < max_levels; level++) {
      const unsigned long long most = std::min<unsigned long long>(1ull << level, (unsigned long long)n / 5 + 1);
      LAUNCH(k_lbvh_split, most, A, level);
      LAUNCH(k_lbvh_next, 1, A, level);
    }
    HIP_TRY(hipMemcpyAsync(h_lv.data(), lv, h_lv.size() * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (h_lv[max_levels].y != 0) return fail(ctx, RT_ERR_DEVICE, "device BVH build ran out of levels");
    for (int level = 0; level < max_levels && h_lv[level].y; level++) depth = level + 1;
    for (int level = depth - 1; level >= 0; level--) LAUNCH(k_lbvh_refit, h_lv[level].y, A, level);
    info.n_nodes = h_lv[depth - 1].x + h_lv[depth - 1].y;
  }
  LAUNCH(k_lbvh_root, 1, A, root_ref, out);
  float h_out[6];
  HIP_TRY(hipMemcpyAsync(h_out, out, sizeof h_out, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  D.root_ref = root_ref;
  for (int k = 0; k < 6; k++) D.bounds[k] = h_out[k];
  return RT_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

void rt_render_params_default(rt_render_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->sampler = RT_SAMPLER_PHILOX;
  p->seed_lo = 1;
  p->spp_min = 32;   // raytracer.h:21
  p->spp_max = 128;  // raytracer.h:20
  p->adaptive_threshold = 0.01f;  // RayTracer.cpp:191
  p->pixel_jitter = 1;
  p->max_bounces = 3;  // raytracer.h:17
  p->mc_bounces = 0;   // the indirect-diffuse sampler is commented out in the reference (materials.cpp:309-314)
  p->mc_samples = 1;   // renderer.h:172
  p->shadow_min = 16;  // renderer.h:174
  p->shadow_max = 128; // renderer.h:173
  p->light_size_scale = 1.0f;
  p->dedup_hard_shadows = 1;
  p->photons = 1;
  p->n_photons = 100000;  // raytracer.h:23
  p->photon_radius = 3.0f;  // materials.cpp:316,320
  p->photon_k = 128;
  p->ellipticity_global = 1.0f;
  p->ellipticity_caustic = 0.25f;
  p->skip_zero_weight = 1;
}

int rt_context_create(int device_id, rt_context** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_ERR_DEVICE;
  if (device_id < 0 || device_id >= n) return RT_ERR_INVALID;
  rt_context* ctx = new rt_context;
  ctx->device = device_id;
  std::memset(&ctx->prof, 0, sizeof ctx->prof);
  std::memset(&ctx->ds, 0, sizeof ctx->ds);
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
    delete ctx;
    return RT_ERR_DEVICE;
  }
#ifndef RT_EMU
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
  {
    ctx->max_blocks = prop.multiProcessorCount * RT_BLOCKS_PER_CU;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->max_blocks_trace = prop.multiProcessorCount * RT_TRACE_BLOCKS_PER_CU;  // re-sized per scene by rt_scene_commit
  }
#endif
  // every failure from here on leaves through ONE clean-up of whatever exists so far (rt_context_destroy copes with nulls)
  auto fail_create = [&]() { rt_context_destroy(ctx); return RT_ERR_DEVICE; };
  void* p = nullptr;
  if (hipMalloc(&p, sizeof(DCounters)) != hipSuccess) return fail_create();
  ctx->d_counters = (DCounters*)p;
  if (hipMemset(p, 0, sizeof(DCounters)) != hipSuccess) return fail_create();
  {  // per-pass summary (k_pass_summary): device words + their pinned host copy
    void *ds = nullptr, *hs = nullptr;
    if (hipMalloc(&ds, 4096) != hipSuccess) return fail_create();
    ctx->d_summary = (uint32_t*)ds;
    if (hipHostMalloc(&hs, 4096) != hipSuccess) return fail_create();
    ctx->h_summary = (uint32_t*)hs;
    void* hd = nullptr;   // page-locked host memory is mapped into the device's address space: the summary kernel writes it in place
    if (const char* e = std::getenv("RT_SUMMARY_MAPPED")) ctx->opt.summary_mapped = std::atof(e);   // (A/B knob: 0 = copy it)
    if (ctx->opt.summary_mapped != 0 && hipHostGetDevicePointer(&hd, hs, 0) == hipSuccess && hd) ctx->h_summary_dev = (uint32_t*)hd;
  }
  {  // this device's address of the no-allocation entry point's working array (the device is current: hipSetDevice above)
#ifndef RT_EMU
    void* sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) return fail_create();
    ctx->noalloc_ptr = (unsigned char*)sp;
#else
    ctx->noalloc_ptr = scratch;
#endif
  }
  // developer knobs: the environment is read here and nowhere else
  auto env = [](const char* name, double& v) { if (const char* e = std::getenv(name)) v = std::atof(e); };
  env("RT_TLAS_MIN", ctx->opt.tlas_min);
  env("RT_GATHER_HINTS", ctx->opt.gather_hints);
  env("RT_FRAME_GB", ctx->opt.frame_gb);
  env("RT_TEST_FAN_SCALE", ctx->opt.fan_scale);
  if (std::getenv("RT_TIMING")) ctx->opt.timing = 1;
  if (const char* e = std::getenv("RT_SOFT_CLEAR")) ctx->opt.soft_clear = std::atof(e);
  if (const char* e = std::getenv("RT_SHADOW_HINTS")) ctx->opt.shadow_hints = std::atof(e);
  env("RT_FUSE", ctx->opt.fuse);
  *out = ctx;
  return RT_OK;
}

int rt_set_option(rt_context* ctx, const char* name, double value) {
  if (!ctx || !name) return RT_ERR_INVALID;
  quiesce(ctx);
  const std::string n(name);
  if (n == "tlas_min") ctx->opt.tlas_min = value;            // takes effect at the next rt_scene_commit
  else if (n == "gather_hints") ctx->opt.gather_hints = value;
  else if (n == "frame_gb") ctx->opt.frame_gb = value;
  else if (n == "fan_scale") { ctx->opt.fan_scale = value; ctx->fan_valid = false; }
  else if (n == "timing") ctx->opt.timing = value;
  else if (n == "soft_clear") ctx->opt.soft_clear = value;
  else if (n == "cone_max_inst") ctx->opt.cone_max_inst = value;
  else if (n == "fuse") ctx->opt.fuse = value;
  else if (n == "restart_min_depth") ctx->opt.restart_min_depth = value;
  else if (n == "shadow_hints") ctx->opt.shadow_hints = value;
  else if (n == "shadow_tlas_min") ctx->opt.shadow_tlas_min = value;   // takes effect at the next rt_scene_commit
  else return fail(ctx, RT_ERR_INVALID, "unknown option: " + n);
  return RT_OK;
}

int rt_debug_alloc_count(rt_context* ctx, uint64_t* n) {
  if (!ctx || !n) return RT_ERR_INVALID;
  quiesce(ctx);
  *n = ctx->n_allocs;
  return RT_OK;
}

void rt_context_destroy(rt_context* ctx) {
  if (!ctx) return;
  if (ctx->worker.joinable()) ctx->worker.join();
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_pool(ctx->scene_mem);
  free_pool(ctx->frame_mem);
  for (auto& m : ctx->bufs) m.release();
  photon_state_free(ctx->photons);
  if (ctx->d_gather_hints) (void)hipFree(ctx->d_gather_hints);
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->d_summary) (void)hipFree(ctx->d_summary);
  if (ctx->h_summary) (void)hipHostFree(ctx->h_summary);
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  for (auto& e : ctx->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rt_context_create_multi(int n_devices, const int* device_ids, rt_context** out) {
  if (!out || n_devices <= 0 || n_devices > 64) return RT_ERR_INVALID;
  for (int i = 0; i < n_devices; i++) out[i] = nullptr;
  for (int i = 0; i < n_devices; i++)
    for (int j = 0; j < i; j++)
      if ((device_ids ? device_ids[i] : i) == (device_ids ? device_ids[j] : j)) return RT_ERR_INVALID;   // one context per GPU
  for (int i = 0; i < n_devices; i++) {
    const int rc = rt_context_create(device_ids ? device_ids[i] : i, &out[i]);
    if (rc != RT_OK) {
      for (int j = 0; j < i; j++) { rt_context_destroy(out[j]); out[j] = nullptr; }
      return rc;
    }
  }
  return RT_OK;
}

int rt_alloc_pinned(size_t bytes, void** out) {
  if (!out) return RT_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return RT_ERR_INVALID;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes) != hipSuccess || !p) return RT_ERR_DEVICE;
  *out = p;
  return RT_OK;
}
void rt_free_pinned(void* p) {
  if (p) (void)hipHostFree(p);
}

int rt_scene_commit(rt_context* ctx, const rt_scene_desc* sc) {
  if (!ctx) return RT_ERR_INVALID;
  if (!sc || !sc->nodes || sc->n_nodes == 0) return fail(ctx, RT_ERR_INVALID, "scene has no nodes");
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  free_pool(ctx->scene_mem);
  ctx->has_scene = false;
  ctx->fan_valid = false;
  photon_state_free(ctx->photons);
  if (ctx->d_gather_hints) HIP_TRY(hipMemsetAsync(ctx->d_gather_hints, 0, ctx->gather_hint_px * 2 * sizeof(float), ctx->stream));
  ctx->gather_hints_cold = true;

  // ---- nodes and instances
  for (uint32_t i = 0; i < sc->n_nodes; i++) {
    const rt_node& n = sc->nodes[i];
    if (i == 0 ? n.parent != -1 : (n.parent < 0 || n.parent >= (int)i))
      return fail(ctx, RT_ERR_INVALID, "nodes must be in depth-first order with parent < index");
  }
  std::vector<DInstance> inst;
  std::vector<float> xf;  // per instance: depth x { itm[12], tm[12] }, root level first
  for (uint32_t i = 0; i < sc->n_nodes; i++) {
    const rt_node& n = sc->nodes[i];
    if (n.prim_kind == RT_PRIM_NONE) continue;
    if (n.prim_kind != RT_PRIM_SPHERE && n.prim_kind != RT_PRIM_PLANE && n.prim_kind != RT_PRIM_MESH)
      return fail(ctx, RT_ERR_INVALID, "unknown primitive kind");
    if (n.prim_kind == RT_PRIM_MESH && (n.mesh_id < 0 || n.mesh_id >= (int)sc->n_meshes))
      return fail(ctx, RT_ERR_INVALID, "mesh index out of range");
    DInstance I;
    std::memset(&I, 0, sizeof I);
    I.node = (int)i; I.prim = n.prim_kind; I.mesh = n.mesh_id;
    I.mtl = (n.material_id >= 0 && n.material_id < (int)sc->n_materials) ? n.material_id : (int)sc->n_materials;
    int chain[64], d = 0;
    for (int k = (int)i; k >= 0 && d < 64; k = sc->nodes[k].parent) chain[d++] = k;
    if (d > RT_MAX_CHAIN) return fail(ctx, RT_ERR_UNSUPPORTED, "scene graph deeper than 8 levels");
    I.depth = d;
    I.parent_key = n.parent;
    I.xf_off = (int32_t)xf.size();
    for (int k = 0; k < d; k++) {
      const rt_node& cn = sc->nodes[chain[d - 1 - k]];
      xf.insert(xf.end(), cn.itm.m, cn.itm.m + 12);
      xf.insert(xf.end(), cn.tm.m, cn.tm.m + 12);
    }
    if (I.prim != RT_PRIM_MESH) I.stretch = chain_stretch(xf.data() + I.xf_off, d);   // (shares the word with a mesh instance's shade_base)
    // padded world bounds (only mesh instances are culled with them)
    double lo[3], hi[3];
    if (n.prim_kind == RT_PRIM_MESH) {
      const rt_mesh& m = sc->meshes[n.mesh_id];
      for (int k = 0; k < 3; k++) { lo[k] = m.bound_min[k]; hi[k] = m.bound_max[k]; }
    } else {
      lo[0] = lo[1] = -1; hi[0] = hi[1] = 1;
      lo[2] = n.prim_kind == RT_PRIM_PLANE ? 0 : -1;
      hi[2] = n.prim_kind == RT_PRIM_PLANE ? 0 : 1;
    }
    double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
    for (int c = 0; c < 8; c++) {
      double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
      for (int k = d - 1; k >= 0; k--) {
        const float* t = sc->nodes[chain[d - 1 - k]].tm.m;
        double q[3];
        for (int r = 0; r < 3; r++) q[r] = (double)t[r] * p[0] + (double)t[3 + r] * p[1] + (double)t[6 + r] * p[2] + (double)t[9 + r];
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
      }
      for (int k = 0; k < 3; k++) { wlo[k] = std::min(wlo[k], p[k]); whi[k] = std::max(whi[k], p[k]); }
    }
    for (int k = 0; k < 3; k++) {
      double pad = 1e-3 * (whi[k] - wlo[k]) + 1e-3 * std::max(std::fabs(wlo[k]), std::fabs(whi[k])) + 1e-4;
      I.wmin[k] = (float)(wlo[k] - pad);
      I.wmax[k] = (float)(whi[k] + pad);
    }
    inst.push_back(I);
  }

  // ---- meshes: trees that came with the mesh are converted here and stand first in the node / slot arrays; meshes without
  // one get their range behind them and are built on the device further down (lbvh_build)
  std::vector<DMesh> meshes(sc->n_meshes);
  std::vector<MeshInfo> minfo(sc->n_meshes);
  int max_mesh_depth = 0;
  std::vector<float4> bvh, tris;
  std::vector<float4> shade;   // DScene::shade
  std::vector<float4> face_pos(1);   // DScene::face_pos (meshes with texture coordinates only)
  std::vector<uint32_t> bvh_anc(1);  // DScene::bvh_anc
  std::vector<uint32_t> face_pos_base(sc->n_meshes, RT_NONE);
  for (uint32_t mi = 0; mi < sc->n_meshes; mi++) {
    const rt_mesh& m = sc->meshes[mi];
    if (!m.positions || !m.normals || !m.pos_idx || !m.nrm_idx || (m.bvh && (m.n_bvh_nodes == 0 || (m.n_faces && !m.bvh_elements))))
      return fail(ctx, RT_ERR_INVALID, "mesh with missing arrays");
    for (size_t k = 0; k < 3 * (size_t)m.n_faces; k++) {
      if (m.pos_idx[k] >= m.n_positions || m.nrm_idx[k] >= m.n_normals) return fail(ctx, RT_ERR_INVALID, "mesh index out of range");
      if (m.uvw_idx && m.uvw && m.uvw_idx[k] >= m.n_uvw) return fail(ctx, RT_ERR_INVALID, "mesh uvw index out of range");
    }
    DMesh& D = meshes[mi];
    std::memset(&D, 0, sizeof D);
    D.anc_base = RT_NONE;   // (meshes whose tree is built on the device: walks start at the root)
    if (!m.bvh) continue;
    MeshConv cv(m, bvh, tris);
    uint32_t root;
    if (!cv.convert(0, 0, root)) return fail(ctx, RT_ERR_INVALID, "mesh BVH: " + cv.err);
    max_mesh_depth = std::max(max_mesh_depth, cv.max_depth);
    D.node_base = (uint32_t)cv.node_base;
    D.tri_base = (uint32_t)cv.tri_base;
    D.root_ref = root;
    const rt_bvh_node& r = m.bvh[0];
    for (int k = 0; k < 3; k++) { D.bounds[k] = r.bmin[k]; D.bounds[3 + k] = r.bmax[k]; }
    minfo[mi].n_nodes = cv.n_internal; minfo[mi].n_slots = cv.n_slots;
    if (cv.nested && !(root & RT_LEAF) && cv.n_slots > 0 && cv.max_depth >= (int)ctx->opt.restart_min_depth) {
      // the last block may be partial, and a block that straddles the two halves of the root is the root's: both say "start at the root"
      cv.anc.resize(((size_t)cv.n_slots + (1u << RT_ANC_SHIFT) - 1) >> RT_ANC_SHIFT, RT_NONE);
      for (auto& a : cv.anc) if (a == RT_NONE) a = root;
      D.anc_base = (uint32_t)bvh_anc.size();
      D.anc_blocks = (uint32_t)cv.anc.size();
      bvh_anc.insert(bvh_anc.end(), cv.anc.begin(), cv.anc.end());
    }
  }
  size_t dev_nodes = 0, dev_slots = 0;   // float4s behind the converted part
  for (uint32_t mi = 0; mi < sc->n_meshes; mi++) {
    const rt_mesh& m = sc->meshes[mi];
    if (m.bvh) continue;
    if ((unsigned long long)m.n_faces > (1ull << 27)) return fail(ctx, RT_ERR_UNSUPPORTED, "mesh too large for the device BVH build");
    DMesh& D = meshes[mi];
    if ((bvh.size() + dev_nodes + 4 * (size_t)m.n_faces) >> 32 || (tris.size() + dev_slots + 3 * (size_t)m.n_faces) >> 32)
      return fail(ctx, RT_ERR_UNSUPPORTED, "mesh arrays beyond 2^32 entries");
    D.node_base = (uint32_t)(bvh.size() + dev_nodes);
    D.tri_base = (uint32_t)(tris.size() + dev_slots);
    dev_nodes += 4 * (size_t)std::max<uint32_t>(m.n_faces, 1u);
    dev_slots += 3 * (size_t)std::max<uint32_t>(m.n_faces, 1u);
  }
  for (uint32_t mi = 0; mi < sc->n_meshes; mi++) {
    const rt_mesh& m = sc->meshes[mi];
    DMesh& D = meshes[mi];
    if ((shade.size() / 5 + (size_t)m.n_faces) >> 32) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^32 faces");
    D.shade_base = (uint32_t)(shade.size() / 5);
    const bool has_uvw = m.uvw && m.uvw_idx;
    shade.resize(shade.size() + 5 * (size_t)m.n_faces);
    for (size_t f = 0; f < (size_t)m.n_faces; f++) {
      const float* n0 = m.normals + 3 * (size_t)m.nrm_idx[3 * f], *n1 = m.normals + 3 * (size_t)m.nrm_idx[3 * f + 1],
                 * n2 = m.normals + 3 * (size_t)m.nrm_idx[3 * f + 2];
      const float zero[3] = {0.0f, 0.0f, 0.0f};
      const float* t0 = has_uvw ? m.uvw + 3 * (size_t)m.uvw_idx[3 * f] : zero;
      const float* t1 = has_uvw ? m.uvw + 3 * (size_t)m.uvw_idx[3 * f + 1] : zero;
      const float* t2 = has_uvw ? m.uvw + 3 * (size_t)m.uvw_idx[3 * f + 2] : zero;
      float4* r = &shade[5 * ((size_t)D.shade_base + f)];
      r[0] = make_float4(n0[0], n0[1], n0[2], t0[0]);
      r[1] = make_float4(n1[0], n1[1], n1[2], t0[1]);
      r[2] = make_float4(n2[0], n2[1], n2[2], t0[2]);
      r[3] = make_float4(t1[0], t1[1], t1[2], t2[0]);
      r[4] = make_float4(t2[1], t2[2], 0.0f, 0.0f);
    }
    D.has_uvw = has_uvw ? 1 : 0;
    if (has_uvw && m.n_faces) {   // the triangle in face order, edges by the reference's expression (objects.cpp:203-204)
      if (((face_pos.size() / 3) + (size_t)m.n_faces) >> 32) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 2^32 faces");
      face_pos_base[mi] = (uint32_t)(face_pos.size() / 3);
      face_pos.resize(face_pos.size() + 3 * (size_t)m.n_faces);
      for (size_t f = 0; f < (size_t)m.n_faces; f++) {
        const float *a = m.positions + 3 * (size_t)m.pos_idx[3 * f], *b = m.positions + 3 * (size_t)m.pos_idx[3 * f + 1],
                    *c = m.positions + 3 * (size_t)m.pos_idx[3 * f + 2];
        float4* r = &face_pos[3 * ((size_t)face_pos_base[mi] + f)];
        r[0] = make_float4(a[0], a[1], a[2], 0.0f);
        r[1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], 0.0f);
        r[2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
      }
    }
  }
  {
    int32_t mk = 0;
    for (DInstance& I : inst)
      if (I.prim == RT_PRIM_MESH) {
        I.shade_base = meshes[I.mesh].shade_base;
        I.has_uvw = meshes[I.mesh].has_uvw;
        I.minst_k = mk++;   // position in DScene::minst / minst_rec (built below in the same order)
      }
  }

  // ---- materials, textures, lights
  std::vector<DMaterial> mtls;
  ctx->h_materials.clear();
  auto check_tc = [&](const rt_tex_color& t) { return t.map < (int)sc->n_texmaps; };
  for (uint32_t i = 0; i < sc->n_materials; i++) {
    const rt_material& m = sc->materials[i];
    if (!check_tc(m.diffuse) || !check_tc(m.specular) || !check_tc(m.emission) || !check_tc(m.reflection) ||
        !check_tc(m.refraction) || m.glossiness.map >= (int)sc->n_texmaps)
      return fail(ctx, RT_ERR_INVALID, "material texture map index out of range");
    mtls.push_back(conv_mtl(m));
    ctx->h_materials.push_back(m);
  }
  mtls.push_back(conv_mtl(default_material()));
  ctx->h_materials.push_back(default_material());
  std::vector<DTexmap> tms(sc->n_texmaps);
  for (uint32_t i = 0; i < sc->n_texmaps; i++) {
    const rt_texmap& t = sc->texmaps[i];
    if (t.kind != RT_TEX_CHECKER && t.kind != RT_TEX_IMAGE) return fail(ctx, RT_ERR_INVALID, "unknown texture kind");
    if (t.kind == RT_TEX_IMAGE && (t.image < 0 || t.image >= (int)sc->n_images)) return fail(ctx, RT_ERR_INVALID, "image index out of range");
    if (!check_tc(t.color[0]) || !check_tc(t.color[1])) return fail(ctx, RT_ERR_INVALID, "checker colour map index out of range");
    tms[i].kind = t.kind; tms[i].image = t.image;
    std::memcpy(tms[i].itm, t.itm.m, sizeof tms[i].itm);
    tms[i].color[0] = conv_tc(t.color[0]); tms[i].color[1] = conv_tc(t.color[1]);
  }
  if (!check_tc(sc->background) || !check_tc(sc->environment)) return fail(ctx, RT_ERR_INVALID, "background map index out of range");
  std::vector<DImage> imgs(sc->n_images);
  std::vector<uint8_t> texels;
  for (uint32_t i = 0; i < sc->n_images; i++) {
    const rt_image& im = sc->images[i];
    if (!im.rgb || im.width <= 0 || im.height <= 0) return fail(ctx, RT_ERR_INVALID, "empty image");
    imgs[i].offset = (uint32_t)texels.size();
    imgs[i].width = im.width; imgs[i].height = im.height; imgs[i].pad = 0;
    texels.insert(texels.end(), im.rgb, im.rgb + 3 * (size_t)im.width * im.height);
  }
  std::vector<DLight> lights(sc->n_lights);
  ctx->lights.assign(sc->lights, sc->lights + sc->n_lights);
  for (uint32_t i = 0; i < sc->n_lights; i++) {
    const rt_light& l = sc->lights[i];
    if (l.kind != RT_LIGHT_AMBIENT && l.kind != RT_LIGHT_DIRECT && l.kind != RT_LIGHT_POINT)
      return fail(ctx, RT_ERR_INVALID, "unknown light kind");
    lights[i].kind = l.kind;
    std::memcpy(lights[i].intensity, l.intensity, 12);
    std::memcpy(lights[i].position, l.position, 12);
    std::memcpy(lights[i].direction, l.direction, 12);
    lights[i].size = l.size; lights[i].attenuation = l.attenuation;
  }
  if (sc->n_lights > 250) return fail(ctx, RT_ERR_UNSUPPORTED, "more than 250 lights");

  // ---- Halton tables (values computed with the reference's float loop, rng.h:103-112)
  const int n_gloss = 4096;
  std::vector<float> hal(RT_HGLOSS + 2 * n_gloss + 2 * RT_FILTER_TAPS);
  for (int i = 0; i < RT_FILTER_TAPS; i++) {   // Texture::Eval(uvw, duvw)'s tap offsets, its own float loops (scene.h:294-299)
    float x = 0, y = 0, fx = 0.5f, fy = 1.0f / 3.0f;
    for (int ix = i; ix > 0; ix /= 2) { x += fx * (ix % 2); fx /= 2; }
    for (int iy = i; iy > 0; iy /= 3) { y += fy * (iy % 3); fy /= 3; }
    if (x > 0.5f) x -= 1;
    if (y > 0.5f) y -= 1;
    hal[RT_HGLOSS + 2 * n_gloss + i] = x;
    hal[RT_HGLOSS + 2 * n_gloss + RT_FILTER_TAPS + i] = y;
  }
  for (int i = 0; i < 128; i++) {
    hal[RT_H2 + i] = halton_ref(i, 2); hal[RT_H3 + i] = halton_ref(i, 3);
    hal[RT_H5 + i] = halton_ref(i, 5); hal[RT_H7 + i] = halton_ref(i, 7);
  }
  for (int i = 0; i < 1000; i++) { hal[RT_HPHI + i] = halton_ref(i, 2); hal[RT_HTHETA + i] = halton_ref(i, 3); }
  for (int i = 0; i < n_gloss; i++) { hal[RT_HGLOSS + i] = halton_ref(i, 2); hal[RT_HGLOSS + n_gloss + i] = halton_ref(i, 3); }

  DScene& D = ctx->ds;
  std::memset(&D, 0, sizeof D);
  int rc;
  if ((rc = upload(ctx, ctx->scene_mem, xf, &D.xf))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, inst, &D.inst))) return rc;
  {
    std::vector<DShadeInst> sinst(inst.size());
    for (size_t i = 0; i < inst.size(); i++) {
      DShadeInst& R = sinst[i];
      std::memset(&R, 0, sizeof R);
      R.prim = inst[i].prim; R.node = inst[i].node; R.depth = inst[i].depth; R.xf_off = inst[i].xf_off;
      R.shade_base = inst[i].prim == RT_PRIM_MESH ? inst[i].shade_base : 0u; R.has_uvw = inst[i].has_uvw;
      R.face_pos_base = inst[i].prim == RT_PRIM_MESH ? face_pos_base[inst[i].mesh] : RT_NONE;
      R.mtl = mtls[inst[i].mtl];
      const DMaterial& M = R.mtl;
      R.uses_uvw = (M.diffuse.map >= 0 || M.specular.map >= 0 || M.emission.map >= 0 || M.reflection.map >= 0 || M.refraction.map >= 0 ||
                    M.glossiness.map >= 0) ? 1u : 0u;
    }
    if ((rc = upload(ctx, ctx->scene_mem, sinst, &D.sinst))) return rc;
  }
  {  // node and slot arrays: the converted trees, then the trees built here
    void *d_bvh, *d_tris;
    if ((rc = alloc_dev(ctx, ctx->scene_mem, (bvh.size() + dev_nodes) * sizeof(float4), &d_bvh))) return rc;
    if ((rc = alloc_dev(ctx, ctx->scene_mem, (tris.size() + dev_slots) * sizeof(float4), &d_tris))) return rc;
    if (!bvh.empty()) HIP_TRY(hipMemcpy(d_bvh, bvh.data(), bvh.size() * sizeof(float4), hipMemcpyHostToDevice));
    if (!tris.empty()) HIP_TRY(hipMemcpy(d_tris, tris.data(), tris.size() * sizeof(float4), hipMemcpyHostToDevice));
    D.bvh = (const float4*)d_bvh;
    D.tris = (const float4*)d_tris;
    for (uint32_t mi = 0; mi < sc->n_meshes; mi++) {
      if (sc->meshes[mi].bvh) continue;
      int depth = 0;
      if ((rc = lbvh_build(ctx, sc->meshes[mi], (float4*)d_bvh + meshes[mi].node_base, (float4*)d_tris + meshes[mi].tri_base, meshes[mi], minfo[mi], depth)))
        return rc;
      if (depth > RT_STACK) return fail(ctx, RT_ERR_INVALID, "mesh BVH: BVH deeper than 32 levels");
      max_mesh_depth = std::max(max_mesh_depth, depth);
    }
  }
  if ((rc = upload(ctx, ctx->scene_mem, meshes, &D.meshes))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, shade, &D.shade))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, face_pos, &D.face_pos))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, bvh_anc, &D.bvh_anc))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, mtls, &D.materials))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, tms, &D.texmaps))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, imgs, &D.images))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, texels, &D.texels))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, lights, &D.lights))) return rc;
  if ((rc = upload(ctx, ctx->scene_mem, hal, &D.halton))) return rc;
  std::vector<int32_t> minst;
  for (size_t i = 0; i < inst.size(); i++)
    if (inst[i].prim == RT_PRIM_MESH) minst.push_back((int32_t)i);
  if ((rc = upload(ctx, ctx->scene_mem, minst, &D.minst))) return rc;
  std::vector<DMeshInst> minst_rec(minst.size());
  for (size_t k = 0; k < minst.size(); k++) {
    const DInstance& I = inst[minst[k]];
    const DMesh& M = meshes[I.mesh];
    DMeshInst& R = minst_rec[k];
    std::memset(&R, 0, sizeof R);
    for (int c = 0; c < 3; c++) { R.wmin[c] = I.wmin[c]; R.wmax[c] = I.wmax[c]; }
    for (int c = 0; c < 6; c++) R.bounds[c] = M.bounds[c];
    R.inst = minst[k]; R.depth = I.depth; R.xf_off = I.xf_off;
    R.root_ref = M.root_ref; R.node_base = M.node_base; R.tri_base = M.tri_base;
    R.anc_base = M.anc_base; R.anc_blocks = M.anc_blocks;
  }
  if ((rc = upload(ctx, ctx->scene_mem, minst_rec, &D.minst_rec))) return rc;
  D.n_minst = (int32_t)minst.size();
  ctx->n_minst = D.n_minst;
  // instance BVH: pays once the wave-uniform table loop is longer than a short divergent walk; the nested mesh walk of
  // the one-phase callers (rt_trace_closest / rt_trace_any, photons) shares the per-lane stack with it
  D.use_tlas = 0;
  D.shadow_tlas = 0;
  int tlas_depth = 0;
  const size_t tlas_min = (size_t)std::max(0.0, ctx->opt.tlas_min);   // (developer knob: tests force both paths)
  if (inst.size() >= tlas_min && inst.size() >= 2) {
    TlasBuild tb(inst);
    std::vector<int32_t> ids(inst.size());
    for (size_t i = 0; i < ids.size(); i++) ids[i] = (int32_t)i;
    uint32_t root = tb.build(ids.data(), ids.size(), 0);
    if (tb.max_depth + 1 + max_mesh_depth + 1 <= RT_STACK) {
      if ((rc = upload(ctx, ctx->scene_mem, tb.nodes, &D.tlas))) return rc;
      if ((rc = upload(ctx, ctx->scene_mem, tb.order, &D.tlas_idx))) return rc;
      D.tlas_root = root;
      D.use_tlas = 1;
      D.shadow_tlas = inst.size() >= (size_t)std::max(0.0, ctx->opt.shadow_tlas_min) ? 1 : 0;
      tlas_depth = tb.max_depth;
    }
  }
  {  // traversal stack: one entry per BVH level that can hold a deferred sibling (+ the instance BVH's when it is used,
     // whose one-phase walk nests the mesh walk above its own entries)
    ctx->max_mesh_depth = max_mesh_depth;
    int entries = max_mesh_depth + 2 + (D.use_tlas ? tlas_depth + 2 : 0);
    entries = std::max(4, std::min(entries, (int)RT_STACK));
#ifndef RT_EMU
    ctx->trace_lds = (size_t)entries * RT_BLOCK * 4;
    int fit = (int)((160u * 1024u) / ctx->trace_lds);
    ctx->max_blocks_trace = ctx->n_cus * std::max(1, std::min(fit, (int)RT_TRACE_BLOCKS_PER_CU));
#endif
  }
  D.n_inst = (int)inst.size();
  D.n_lights = (int)lights.size();
  D.n_materials = (int)sc->n_materials;
  D.n_gloss = n_gloss;
  D.background = conv_tc(sc->background);
  D.environment = conv_tc(sc->environment);
  ctx->camera = sc->camera;
  ctx->n_nodes = (int)sc->n_nodes;
  ctx->h_inst = inst;
  ctx->h_meshes = meshes;
  ctx->h_minfo = minfo;
  ctx->has_scene = true;
  return RT_OK;
}

int rt_scene_mesh_bvh(rt_context* ctx, int mesh_id, rt_bvh_node* nodes, uint32_t node_cap, uint32_t* elements, uint32_t elem_cap,
                      uint32_t* n_nodes, uint32_t* n_elements) {
  if (!ctx || !n_nodes || !n_elements) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (mesh_id < 0 || (size_t)mesh_id >= ctx->h_meshes.size()) return fail(ctx, RT_ERR_INVALID, "mesh index out of range");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const DMesh& D = ctx->h_meshes[mesh_id];
  const MeshInfo& I = ctx->h_minfo[mesh_id];
  std::vector<float4> hn(4 * (size_t)I.n_nodes), ht(3 * (size_t)I.n_slots);
  if (!hn.empty()) HIP_TRY(hipMemcpy(hn.data(), ctx->ds.bvh + D.node_base, hn.size() * sizeof(float4), hipMemcpyDeviceToHost));
  if (!ht.empty()) HIP_TRY(hipMemcpy(ht.data(), ctx->ds.tris + D.tri_base, ht.size() * sizeof(float4), hipMemcpyDeviceToHost));
  auto f2u = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  struct Item { uint32_t ref; float box[6]; int64_t parent; int right; };
  std::vector<rt_bvh_node> out;
  std::vector<uint32_t> el;
  std::vector<Item> stack;
  Item root;
  root.ref = D.root_ref; root.parent = -1; root.right = 0;
  for (int k = 0; k < 6; k++) root.box[k] = D.bounds[k];
  stack.push_back(root);
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const uint32_t me = (uint32_t)out.size();
    rt_bvh_node n;
    for (int k = 0; k < 3; k++) { n.bmin[k] = it.box[k]; n.bmax[k] = it.box[3 + k]; }
    n.a = n.b = 0;
    if (it.parent >= 0) (it.right ? out[it.parent].b : out[it.parent].a) = me;
    if (it.ref & RT_LEAF) {
      const uint32_t first = (it.ref & ~RT_LEAF) >> 3, count = it.ref & 7u;
      if ((size_t)first + count > I.n_slots) return fail(ctx, RT_ERR_DEVICE, "mesh BVH: leaf outside the slot array");
      n.a = (uint32_t)el.size();
      n.b = RT_BVH_LEAF_BIT | count;
      for (uint32_t s = 0; s < count; s++) el.push_back(f2u(ht[3 * (size_t)(first + s) + 1].w));
      out.push_back(n);
      continue;
    }
    if (it.ref >= I.n_nodes || out.size() > 2 * (size_t)I.n_slots + 2) return fail(ctx, RT_ERR_DEVICE, "mesh BVH: node outside the node array");
    out.push_back(n);
    const float4 x = hn[4 * (size_t)it.ref], y = hn[4 * (size_t)it.ref + 1], z = hn[4 * (size_t)it.ref + 2], r = hn[4 * (size_t)it.ref + 3];
    Item l, rr;
    l.ref = f2u(r.x); l.parent = me; l.right = 0;
    l.box[0] = x.x; l.box[1] = y.x; l.box[2] = z.x; l.box[3] = x.z; l.box[4] = y.z; l.box[5] = z.z;
    rr.ref = f2u(r.y); rr.parent = me; rr.right = 1;
    rr.box[0] = x.y; rr.box[1] = y.y; rr.box[2] = z.y; rr.box[3] = x.w; rr.box[4] = y.w; rr.box[5] = z.w;
    stack.push_back(rr);
    stack.push_back(l);
  }
  *n_nodes = (uint32_t)out.size();
  *n_elements = (uint32_t)el.size();
  if (!nodes) return RT_OK;
  if (node_cap < out.size() || elem_cap < el.size() || (!elements && !el.empty())) return fail(ctx, RT_ERR_INVALID, "capacity too small for the mesh BVH");
  std::memcpy(nodes, out.data(), out.size() * sizeof(rt_bvh_node));
  if (!el.empty()) std::memcpy(elements, el.data(), el.size() * sizeof(uint32_t));
  return RT_OK;
}

int rt_trace_closest(rt_context* ctx, const rt_ray* rays, size_t n, int hit_side, rt_hit* hits) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (n == 0) return RT_OK;
  if (!rays || !hits || n > 0x7fffffffu) return fail(ctx, RT_ERR_INVALID, "bad ray batch");
  static_assert(sizeof(HitOut) == sizeof(rt_hit), "rt_hit layout");
  HIP_TRY(hipSetDevice(ctx->device));
  void *d_r, *d_h;
  int rc;
  if ((rc = grow_dev(ctx, SC_API_A, n * sizeof(rt_ray), &d_r)) || (rc = grow_dev(ctx, SC_API_B, n * sizeof(rt_hit), &d_h))) return rc;
  HIP_TRY(hipMemcpyAsync(d_r, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream));
  LAUNCH_T(k_api_closest, n, ctx->ds, (const float*)d_r, (uint32_t)n, hit_side, (HitOut*)d_h);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(hits, d_h, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_debug_philox(rt_context* ctx, const uint32_t* counters, uint32_t key_lo, uint32_t key_hi, size_t n, uint32_t* words, float* units) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (n == 0) return RT_OK;
  if (!counters || !words || !units || n > 0x0fffffffu) return fail(ctx, RT_ERR_INVALID, "bad counter batch");
  HIP_TRY(hipSetDevice(ctx->device));
  void *d_c, *d_w, *d_u;
  int rc;
  if ((rc = grow_dev(ctx, SC_API_A, n * 16, &d_c)) || (rc = grow_dev(ctx, SC_API_B, n * 16, &d_w)) || (rc = grow_dev(ctx, SC_API_C, n * 16, &d_u))) return rc;
  HIP_TRY(hipMemcpyAsync(d_c, counters, n * 16, hipMemcpyHostToDevice, ctx->stream));
  LAUNCH(k_debug_philox, n, (const uint32_t*)d_c, key_lo, key_hi, (uint32_t)n, (uint32_t*)d_w, (float*)d_u);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(words, d_w, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(units, d_u, n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_trace_any(rt_context* ctx, const rt_ray* rays, const float* t_max, size_t n, uint8_t* occluded) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (n == 0) return RT_OK;
  if (!rays || !occluded || n > 0x7fffffffu) return fail(ctx, RT_ERR_INVALID, "bad ray batch");
  HIP_TRY(hipSetDevice(ctx->device));
  void *d_r, *d_t = nullptr, *d_o;
  int rc;
  if ((rc = grow_dev(ctx, SC_API_A, n * sizeof(rt_ray), &d_r)) || (rc = grow_dev(ctx, SC_API_B, n, &d_o))) return rc;
  if (t_max && (rc = grow_dev(ctx, SC_API_C, n * sizeof(float), &d_t))) return rc;
  HIP_TRY(hipMemcpyAsync(d_r, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream));
  if (t_max) HIP_TRY(hipMemcpyAsync(d_t, t_max, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  LAUNCH_T(k_api_any, n, ctx->ds, (const float*)d_r, (const float*)d_t, (uint32_t)n, (uint8_t*)d_o);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(occluded, d_o, n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

}  // extern "C"

// ================================================================================================ rendering
namespace {

// Device layout of the queue-length words: [level][kind][segment * RT_CSTRIDE]; then the overflow flag and the
// active-pixel counter.  Level l's ray queue lengths live at (l, CNT_Q); its children are counted at (l + 1, CNT_Q).
enum { CNT_Q = 0, CNT_GH = 1, CNT_GS = 2, CNT_N2 = 3, CNT_GQ = 4,
       CNT_MQ = 5, CNT_MH = 6, CNT_MS1 = 7, CNT_MS2 = 8,        // mesh-phase item queues: closest, hard shadow, soft shadow phases
       CUR_MQ = 9, CUR_MH = 10, CUR_MS1 = 11, CUR_MS2 = 12,     // their claim cursors
       CNT_GR = 13,                                             // photon lookups left to the second gather stage
       CNT_CQ = 14, CNT_CH = 15, CNT_CS1 = 16, CNT_CS2 = 17,    // candidate lists of the mesh-phase items (MeshQ::cand)
       CNT_N1 = 18,                                             // soft-shadow groups without a clear view of their light (k_shadow_soft_clear)
       CNT_KINDS = 19 };
constexpr int NFAN = 10;
constexpr int MAX_LEVELS = 8;
constexpr uint32_t NSEG = 64;
constexpr size_t CNT_SLAB = (size_t)NSEG * RT_CSTRIDE;                       // words per (level, kind)
constexpr size_t CNT_OVERFLOW = (size_t)(MAX_LEVELS + 1) * CNT_KINDS * CNT_SLAB;
constexpr size_t CNT_ACTIVE = CNT_OVERFLOW + 1;
constexpr size_t CNT_TOTAL = CNT_OVERFLOW + 8;
inline size_t cnt_at(int level, int kind) { return ((size_t)level * CNT_KINDS + kind) * CNT_SLAB; }

struct Timer {
  rt_context* ctx;
  int kind;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  Timer(rt_context* c, int k, bool allowed = true) : ctx(c), kind(k), on(c->profile_on && allowed) {
    if (!on) return;
    auto take = [&]() {
      hipEvent_t e;
      if (!ctx->ev_pool.empty()) { e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); }
      else (void)hipEventCreate(&e);
      return e;
    };
    a = take(); b = take();
    (void)hipEventRecord(a, ctx->stream);
  }
  void stop(uint64_t items_slot) {
    if (!on) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->ev_used.push_back({a, b, kind, items_slot});
  }
};

// After a pass has been synchronised: fold event timings and per-level item counts into the profile.
void collect_profile(rt_context* ctx, const std::vector<uint32_t>& counts) {
  for (auto& e : ctx->ev_used) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      ctx->prof.ms[e.kind] += ms;
      ctx->prof.launches[e.kind] += 1;
      if (e.items_slot < counts.size()) ctx->prof.items[e.kind] += counts[e.items_slot];
      else if (e.items_slot >= (1ull << 32)) ctx->prof.items[e.kind] += e.items_slot - (1ull << 32);
    }
    ctx->ev_pool.push_back(e.a);
    ctx->ev_pool.push_back(e.b);
  }
  ctx->ev_used.clear();
}

// ---- photon pass (RayTracer::GeneratePhotons, RayTracer.cpp:262-294); kernels in rt_photon.h / rt_photon_build.h

template <class Load>
void scan_u2(rt_context* ctx, Load ld, uint32_t n, uint2* tiles, uint2* out) {
  if (n == 0) return;
  const uint32_t nt = (n + RT_SCAN_TILE - 1) / RT_SCAN_TILE;
#ifndef RT_EMU
  hipLaunchKernelGGL(k_scan_sums<Load>, dim3(nt), dim3(RT_BLOCK), 0, ctx->stream, ld, n, tiles);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(RT_BLOCK), 0, ctx->stream, tiles, nt);
  hipLaunchKernelGGL(k_scan_write<Load>, dim3(nt), dim3(RT_BLOCK), 0, ctx->stream, ld, n, (const uint2*)tiles, out);
#else
  k_scan_write(ld, n, tiles, out);
#endif
}

int photon_ensure_buffers(PhotonState& ps, rt_context* ctx, int which, size_t n) {
  if (!ps.d_raw[which] || ps.raw_cap[which] < n) {
    if (ps.d_raw[which]) { (void)hipFree(ps.d_raw[which]); ps.d_raw[which] = nullptr; ps.raw_cap[which] = 0; }
    HIP_TRY(hipMalloc(&ps.d_raw[which], std::max<size_t>(n, 1) * sizeof(rt_photon)));
    ctx->n_allocs++;
    ps.raw_cap[which] = n;
  }
  if (!ps.d_tree[which] || ps.tree_cap[which] < n) {
    if (ps.d_tree[which]) { (void)hipFree(ps.d_tree[which]); ps.d_tree[which] = nullptr; ps.tree_cap[which] = 0; }
    HIP_TRY(hipMalloc(&ps.d_tree[which], (n + 2) * sizeof(rt_photon)));
    ctx->n_allocs++;
    ps.tree_cap[which] = n;
  }
  return RT_OK;
}

// PhotonMap::PrepareForIrradianceEstimation (photonmap.h:176-198) for the maps in `mask` (bit 0 global, bit 1 caustics), from
// their insertion-order arrays ps.d_raw[] to heap-ordered kd-trees in ps.d_tree[]: everything on the device, no host pass,
// no synchronisation (rt_photon_build.h).
int photon_kd_build(PhotonState& ps, rt_context* ctx, unsigned mask) {
  KdMaps M;
  std::memset(&M, 0, sizeof M);
  uint32_t max_n = 0;
  for (int w = 0; w < 2; w++) {
    if (!(mask & (1u << w))) continue;
    // the tree's unused entries (0 and count + 1) are zero, as the reference's value-initialised vector (photonmap.h:193)
    HIP_TRY(hipMemsetAsync(ps.d_tree[w], 0, ((size_t)ps.count[w] + 2) * sizeof(rt_photon), ctx->stream));
    if (ps.count[w] == 0) continue;
    const uint32_t m = M.n_maps++;
    M.raw[m] = (const float4*)ps.d_raw[w];
    M.tree[m] = (float4*)ps.d_tree[w];
    M.off[m] = M.n_total;
    M.n[m] = (uint32_t)ps.count[w];
    M.n_total += M.n[m];
    max_n = std::max(max_n, M.n[m]);
  }
  if (M.n_maps == 0) return RT_OK;
  if (M.n_maps == 1) { M.off[1] = 0xFFFFFFFFu; M.n[1] = 0; }
  const size_t N = M.n_total;
  M.cap = max_n / 2 + 1;
  const uint32_t n_tiles = (uint32_t)((3 * N + RT_SCAN_TILE - 1) / RT_SCAN_TILE);
  size_t sort_tmp = 0;
#ifndef RT_EMU
  if (rt_sort_pairs_u64(nullptr, &sort_tmp, nullptr, nullptr, nullptr, nullptr, N, 33, nullptr) != 0) return fail(ctx, RT_ERR_DEVICE, "device sort sizing failed");
#endif
  // one slab, cut twice with the same sequence: first to size it, then to place the pieces
  Carver cv;
  unsigned long long *keys_in = nullptr, *keys_out = nullptr;
  uint32_t *vals_in = nullptr, *L[2][3], *pos_seg[2], *bounds = nullptr;
  uint8_t *side = nullptr, *tmp = nullptr;
  uint2 *scan = nullptr, *tiles = nullptr;
  KdSeg* segs[2];
  for (int pass = 0; pass < 2; pass++) {
    cv.used = 0;
    keys_in = cv.take<unsigned long long>(N); keys_out = cv.take<unsigned long long>(N);
    vals_in = cv.take<uint32_t>(N);
    for (int b = 0; b < 2; b++) for (int a = 0; a < 3; a++) L[b][a] = cv.take<uint32_t>(N);
    pos_seg[0] = cv.take<uint32_t>(N); pos_seg[1] = cv.take<uint32_t>(N);
    side = cv.take<uint8_t>(N);
    scan = cv.take<uint2>(3 * N);
    tiles = cv.take<uint2>((size_t)n_tiles + 2);
    segs[0] = cv.take<KdSeg>((size_t)M.n_maps * M.cap); segs[1] = cv.take<KdSeg>((size_t)M.n_maps * M.cap);
    bounds = cv.take<uint32_t>(16);
    tmp = cv.take<uint8_t>(sort_tmp + 16);
    if (pass == 0) {
      void* slab;
      int rc = grow_dev(ctx, SC_KD, cv.used, &slab);
      if (rc) return rc;
      cv.base = (unsigned char*)slab;
    }
  }
  Timer timer(ctx, RT_K_PHOTON_BUILD);
  // box of each map's photons and the origin (photons[0] is a zeroed slot, photonmap.h:181-190)
  uint32_t h_bounds[12];
  for (int k = 0; k < 12; k++) h_bounds[k] = RT_KD_KEY_ZERO;
  HIP_TRY(hipMemcpyAsync(bounds, h_bounds, sizeof h_bounds, hipMemcpyHostToDevice, ctx->stream));
  LAUNCH(k_kd_bounds, (N + 31) / 32, M, bounds);
  LAUNCH(k_kd_roots, N, M, (const uint32_t*)bounds, segs[0], pos_seg[0]);
  for (int a = 0; a < 3; a++) {   // the three sorted orders
    LAUNCH(k_kd_keys, N, M, a, keys_in, vals_in);
    int rc = sort_pairs(ctx, tmp, sort_tmp, keys_in, keys_out, vals_in, L[0][a], N, 33);
    if (rc) return rc;
  }
  int cur = 0;
  for (uint32_t level = 0; level == 0 || (1u << level) <= max_n / 2; level++) {
    KdLevel V;
    for (int a = 0; a < 3; a++) { V.L[a] = L[cur][a]; V.Ln[a] = L[cur ^ 1][a]; }
    V.pos_seg = pos_seg[cur]; V.pos_seg_n = pos_seg[cur ^ 1];
    V.segs = segs[cur]; V.segs_n = segs[cur ^ 1];
    V.side = side; V.scan = scan;
    V.first_node = 1u << level;
    LAUNCH(k_kd_split, N, M, V);
    KdSideLoad ld;
    for (int a = 0; a < 3; a++) ld.L[a] = V.L[a];
    ld.pos_seg = V.pos_seg; ld.side = side; ld.n_total = (uint32_t)N;
    scan_u2(ctx, ld, (uint32_t)(3 * N), tiles, scan);
    LAUNCH(k_kd_scatter, 3 * N, M, V);
    cur ^= 1;
  }
  HIP_TRY(hipGetLastError());
  timer.stop((1ull << 32) + N);
  return RT_OK;
}

DPhotonMap photon_dmap(const PhotonState& ps, int which) {
  DPhotonMap m;
  m.ph = (const float4*)ps.d_tree[which];
  m.stored = ps.count[which];
  m.half = ps.count[which] / 2 - 1;
  return m;
}

int photon_ensure_built(PhotonState& ps, rt_context* ctx, const rt_render_params& RP) {
  static_assert(sizeof(rt_photon) == 32, "rt_photon must be PhotonData's 32 bytes");
  if (RP.n_photons <= 0 || RP.n_photons > (1 << 26)) return fail(ctx, RT_ERR_INVALID, "n_photons out of range");
  if (RP.photon_k <= 0 || RP.photon_k > RT_PHOTON_MAX_K) return fail(ctx, RT_ERR_UNSUPPORTED, "photon_k must be in 1..256");
  long long limit = RP.photon_emit_limit > 0 ? RP.photon_emit_limit : 20ll * RP.n_photons;
  int max_b = RP.photon_max_bounces > 0 ? RP.photon_max_bounces : 64;
  uint32_t key[8] = {RP.seed_lo, RP.seed_hi, (uint32_t)RP.n_photons, (uint32_t)limit, (uint32_t)max_b, 1u, RP.extensions & RT_EXT_MATERIALS, 0u};
  if (ps.built && !std::memcmp(key, ps.key, sizeof key)) return RT_OK;
  ps.built = false;
  const uint32_t n = (uint32_t)RP.n_photons;
  std::vector<int32_t> plights;
  for (size_t l = 0; l < ctx->lights.size(); l++)
    if (ctx->lights[l].kind == RT_LIGHT_POINT) plights.push_back((int32_t)l);  // IsPhotonSource, lights.h:63

  int rc;
  for (int w = 0; w < 2; w++)
    if ((rc = photon_ensure_buffers(ps, ctx, w, n))) return rc;
  uint32_t kept[2] = {0, 0};
  uint32_t emitted = 0;
  double needed[2] = {0.0, 0.0};   // paths after which each map was full (estimated within the batch that filled it)
  const bool timing = ctx->opt.timing != 0;   // developer aid: host-side phases of the build on stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_start = now();
  int n_batches = 0;
  if (!plights.empty()) {
    uint32_t batch = 1u << 16, cap_batch = 1u << 21;
    uint32_t cap = batch * 16;   // deposit records a batch may add (16 per path of the smallest batch) before it is retried smaller
    // A batch takes as long as its longest path (up to photon_max_bounces closest-hit queries in a row), however few
    // paths it has: the build before this one says how many paths fill the maps, so that one batch does it.  Paths past
    // the point where both maps are full deposit nothing that is kept (first n in emission order), so aiming high is safe.
    // (first up to where the first map was full: past it that map's deposits would only fill the record buffers)
    uint32_t plan[2] = {std::min(ps.paths_needed[0], ps.paths_needed[1]), std::max(ps.paths_needed[0], ps.paths_needed[1])};
    auto planned = [&](uint32_t upto) { return std::min<uint32_t>(cap_batch, (upto + upto / 32 + 4095u) & ~4095u); };
    if (plan[0] > 0) {
      // ONE batch for both maps where the record buffers can be afforded: the map that fills first goes on depositing (those
      // records are dropped when the slots are dealt: first n in emission order), about n / paths_needed[w] per path.  Two
      // batches are two launches as long as their longest paths.
      const uint32_t one = planned(plan[1]);
      const double est = (double)one * ((double)n / (double)plan[0]) * 1.25;   // the denser map's deposits, with a margin
      if (plan[1] <= cap_batch && est < 32.0e6) {
        batch = one;
        cap = std::max<uint32_t>(cap, ((uint32_t)est + 4095u) & ~4095u);
        plan[1] = 0;
      } else if (plan[0] > batch) {
        batch = planned(plan[0]);
      }
    }
    if (max_b > 255) return fail(ctx, RT_ERR_UNSUPPORTED, "photon_max_bounces must be <= 255");
    Carver cv;
    float4* d_rec[2] = {nullptr, nullptr};
    uint2 *d_key[2] = {nullptr, nullptr}, *d_per_path = nullptr, *d_path_base = nullptr, *d_tiles = nullptr;
    uint32_t* d_cnt = nullptr;
    int32_t* d_pl = nullptr;
    for (int pass = 0; pass < 2; pass++) {
      cv.used = 0;
      for (int w = 0; w < 2; w++) { d_rec[w] = cv.take<float4>((size_t)cap * 2); d_key[w] = cv.take<uint2>(cap); }
      d_per_path = cv.take<uint2>(cap_batch);
      d_path_base = cv.take<uint2>(cap_batch);
      d_tiles = cv.take<uint2>(cap_batch / RT_SCAN_TILE + 4);
      d_cnt = cv.take<uint32_t>(4);
      d_pl = cv.take<int32_t>(plights.size());
      if (pass == 0) {
        void* slab;
        if ((rc = grow_dev(ctx, SC_PH, cv.used, &slab))) return rc;
        cv.base = (unsigned char*)slab;
      }
    }
    HIP_TRY(hipMemcpyAsync(d_pl, plights.data(), plights.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    while ((kept[0] < n || kept[1] < n) && (long long)emitted < limit) {
      uint32_t count = (uint32_t)std::min<long long>(batch, limit - emitted);
      PhotonTraceP PP;
      PP.seed_lo = RP.seed_lo; PP.seed_hi = RP.seed_hi;
      PP.first = emitted; PP.count = count;
      PP.max_bounces = max_b;
      PP.ext_materials = (RP.extensions & RT_EXT_MATERIALS) ? 1 : 0;
      PP.n_plights = (int32_t)plights.size();
      PP.accept[0] = kept[0] < n; PP.accept[1] = kept[1] < n;
      PP.cap = cap;
      PhotonOut O;
      for (int w = 0; w < 2; w++) { O.rec[w] = d_rec[w]; O.key[w] = d_key[w]; }
      O.per_path = d_per_path;
      O.count = d_cnt;
      O.overflow = d_cnt + 2;
      O.plights = d_pl;
      O.counters = ctx->d_counters;
      HIP_TRY(hipMemsetAsync(d_cnt, 0, 16, ctx->stream));
      {
        Timer t(ctx, RT_K_PHOTON_TRACE);
        if (ctx->counters_on) LAUNCH_T(k_photon_trace<true>, count, ctx->ds, PP, O);
        else LAUNCH_T(k_photon_trace<false>, count, ctx->ds, PP, O);
        t.stop((1ull << 32) + count);
      }
      // deposits -> their slots in insertion order: the sequential emitter stores them path by path, bounce by bounce, and
      // stops storing into a map that is full (photonmap.h:158-172).  Enqueued before the batch's counts are known: if the
      // record buffers overflowed, the batch is redone in smaller ones and what this wrote beyond `kept` is overwritten.
      PathCountLoad pl;
      pl.per_path = d_per_path;
      scan_u2(ctx, pl, count, d_tiles, d_path_base);
      PhotonPlaceP A;
      for (int w = 0; w < 2; w++) { A.rec[w] = d_rec[w]; A.key[w] = d_key[w]; A.raw[w] = (float4*)ps.d_raw[w]; A.kept[w] = kept[w]; }
      A.n_rec = d_cnt;
      A.path_base = d_path_base;
      A.n = n; A.cap = cap;
      A.scale = 1.0f / (float)RP.n_photons;   // RayTracer.cpp:290,292: 1 / numPhotons (the map's capacity)
      LAUNCH(k_photon_place, cap, A);
      uint32_t h_cnt[4];
      HIP_TRY(hipMemcpyAsync(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      HIP_TRY(hipGetLastError());
      n_batches++;
      if (h_cnt[2]) {  // more deposits than the record buffers hold: redo these paths in smaller batches
        if (batch <= 1024) return fail(ctx, RT_ERR_OVERFLOW, "photon deposit buffer overflow");
        batch /= 4;
        cap_batch = batch;  // do not grow past a size that overflowed
        continue;
      }
      for (int w = 0; w < 2; w++)
        if (PP.accept[w]) {
          // the share of this batch's paths it took to fill map w (deposits are spread evenly over a batch's paths)
          const double share = h_cnt[w] ? std::min(1.0, (double)(n - kept[w]) / (double)h_cnt[w]) : 1.0;
          needed[w] = (double)emitted + share * (double)count;
          kept[w] = (uint32_t)std::min<unsigned long long>(n, (unsigned long long)kept[w] + h_cnt[w]);
        }
      emitted += count;
      // deposits into a map that still accepts them decide how many paths the record buffers can take per batch
      uint32_t densest = std::max(PP.accept[0] ? h_cnt[0] : 0u, PP.accept[1] ? h_cnt[1] : 0u);
      if (plan[1] > emitted + batch && (kept[0] >= n) != (kept[1] >= n)) {
        batch = planned(plan[1] - emitted);   // one map is full: the rest of what the other needed last time, in one batch
        plan[1] = 0;
      } else {
        while (batch < cap_batch && (uint64_t)densest * 8 < cap) { batch *= 2; densest *= 2; }
      }
    }
  }
  const double t_emit = now();
  ps.count[0] = (int32_t)kept[0];
  ps.count[1] = (int32_t)kept[1];
  if ((rc = photon_kd_build(ps, ctx, 3u))) return rc;
  if (timing) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::fprintf(stderr, "photon build: %.2f ms total; %d batches, %u paths; emission + placement %.2f, kd-trees %.2f\n", now() - t_start, n_batches,
                 emitted, t_emit - t_start, now() - t_emit);
  }
  if (ctx->profile_on) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> none;
    collect_profile(ctx, none);
  }
  ps.emitted = emitted;
  for (int w = 0; w < 2; w++) ps.paths_needed[w] = kept[w] >= n ? (uint32_t)std::min(needed[w], 4.0e9) : 0u;
  std::memcpy(ps.key, key, sizeof key);
  ps.built = true;
  return RT_OK;
}

// which: 0 / 1 the global / caustics map in insertion order (GetPhotonMap() before PrepareForIrradianceEstimation);
// tree: the balanced array in heap order, entries 1 .. n (what GetPhotons() points at afterwards, photonmap.h:109)
int photon_download(PhotonState& ps, rt_context* ctx, int which, bool tree, rt_photon* out, int32_t capacity, int32_t* n) {
  *n = 0;
  if (which < 0 || which > 1) return fail(ctx, RT_ERR_INVALID, "which must be 0 (global) or 1 (caustics)");
  if (!ps.built) return fail(ctx, RT_ERR_NOSCENE, "photon maps have not been built (rt_photons_build)");
  int32_t m = ps.count[which];
  *n = m;
  if (out && m > 0) {
    if (capacity < m) return fail(ctx, RT_ERR_INVALID, "photon buffer too small");
    const rt_photon* src = tree ? (const rt_photon*)ps.d_tree[which] + 1 : (const rt_photon*)ps.d_raw[which];
    HIP_TRY(hipMemcpyAsync(out, src, (size_t)m * sizeof(rt_photon), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return RT_OK;
}

int photon_set(PhotonState& ps, rt_context* ctx, int which, const rt_photon* photons, int32_t n) {
  if (which < 0 || which > 1 || n < 0 || (n && !photons)) return fail(ctx, RT_ERR_INVALID, "bad photon array");
  if (n > (1 << 26)) return fail(ctx, RT_ERR_INVALID, "more than 2^26 photons");  // the walk stack packs index << 4 | axis | state
  int rc;
  if ((rc = photon_ensure_buffers(ps, ctx, which, (size_t)n))) return rc;
  if (n) HIP_TRY(hipMemcpyAsync(ps.d_raw[which], photons, (size_t)n * sizeof(rt_photon), hipMemcpyHostToDevice, ctx->stream));
  ps.count[which] = n;
  unsigned mask = 1u << which;
  if (!ps.built && !ps.d_tree[1 - which]) {   // the other map starts out empty
    if ((rc = photon_ensure_buffers(ps, ctx, 1 - which, 0))) return rc;
    ps.count[1 - which] = 0;
    mask = 3u;
  }
  if ((rc = photon_kd_build(ps, ctx, mask))) return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));   // (the caller's array may go away)
  if (!ps.built) {
    std::memset(ps.key, 0xff, sizeof ps.key);  // caller-supplied maps: never mistaken for a built parameter set
    ps.built = true;
  }
  return RT_OK;
}

#ifndef RT_EMU
#define LAUNCH_GATHER(kern, n_items, depth_, ...)                                                          \
  do {                                                                                                     \
    unsigned long long nb_ = ((unsigned long long)(n_items) + RT_GATHER_BLOCK - 1) / RT_GATHER_BLOCK;      \
    if (nb_ < 1) nb_ = 1;                                                                                  \
    size_t lds_ = photon_heap_bytes(depth_);                                                                 \
    /* one-wave blocks, as many as are resident at once (RT_GATHER_WAVES per SIMD, LDS permitting): the queue is dealt to \
       the waves statically, so a grid larger than that runs as a full round plus a thin one */               \
    unsigned long long res_ = (unsigned long long)ctx->n_cus * std::min<unsigned long long>(4ull * RT_GATHER_WAVES, (160u * 1024u) / std::max<size_t>(lds_, 1)); \
    if (nb_ > res_) nb_ = res_;                                                                            \
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);   \
    hipLaunchKernelGGL(kern, dim3((unsigned)nb_), dim3(RT_GATHER_BLOCK), lds_, ctx->stream, __VA_ARGS__);  \
  } while (0)
#else
#define LAUNCH_GATHER(kern, n_items, depth_, ...) kern(__VA_ARGS__)
#endif

int photon_gather_api(PhotonState& ps, rt_context* ctx, int which, const float* P, const float* N, size_t n, float radius,
                      float ellipticity, int k, float* irradiance) {
  if (which < 0 || which > 1) return fail(ctx, RT_ERR_INVALID, "which must be 0 (global) or 1 (caustics)");
  if (!ps.built) return fail(ctx, RT_ERR_NOSCENE, "photon maps have not been built (rt_photons_build)");
  if (k <= 0 || k > RT_PHOTON_MAX_K) return fail(ctx, RT_ERR_UNSUPPORTED, "k must be in 1..256");
  if (n == 0) return RT_OK;
  if (n > 0x7fffffffu) return fail(ctx, RT_ERR_INVALID, "too many queries");
  void *d_p, *d_n, *d_o;
  int rc;
  if ((rc = grow_dev(ctx, SC_API_A, n * 12, &d_p)) || (rc = grow_dev(ctx, SC_API_B, n * 12, &d_n)) || (rc = grow_dev(ctx, SC_API_C, n * 12, &d_o)))
    return rc;
  HIP_TRY(hipMemcpyAsync(d_p, P, n * 12, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(d_n, N, n * 12, hipMemcpyHostToDevice, ctx->stream));
  const int depth = photon_tree_depth(ps.count[which]);
  LAUNCH_GATHER(k_api_photon_gather, n, depth, photon_dmap(ps, which), (const float*)d_p, (const float*)d_n, (uint32_t)n, radius,
                ellipticity, k, depth, (float*)d_o);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(irradiance, d_o, n * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

// Per-pixel lookup hints (GatherP::hints): allocated with the frame size, cleared when a scene is committed.  They stay
// across passes and photon rebuilds — a stale hint costs one walk, never a different result.
void ensure_gather_hints(rt_context* ctx, size_t npx) {
  if (ctx->d_gather_hints && ctx->gather_hint_px == npx) return;
  if (ctx->d_gather_hints) (void)hipFree(ctx->d_gather_hints);
  ctx->d_gather_hints = nullptr;
  ctx->gather_hint_px = 0;
  void* p = nullptr;
  if (hipMalloc(&p, npx * 2 * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return; }  // lookups run without hints
  if (hipMemsetAsync(p, 0, npx * 2 * sizeof(float), ctx->stream) != hipSuccess) { (void)hipFree(p); return; }
  ctx->d_gather_hints = (float*)p;
  ctx->gather_hint_px = npx;
  ctx->gather_hints_cold = true;
}

template <bool CNT>
void photon_gather_queue(rt_context* ctx, const GatherQ& GQ, const FrameBufs& F, const PassP& P, const rt_render_params& RP, uint64_t upper,
                         uint32_t* retry_counts) {
  GatherP G;
  G.n_pix = P.n_pix;
  ensure_gather_hints(ctx, (size_t)P.W * (size_t)P.H);
  G.hints = ctx->d_gather_hints;
  G.n_hint_px = (uint32_t)ctx->gather_hint_px;
  if (ctx->opt.gather_hints == 0) G.hints = nullptr;  // developer knob (tests run both)
  G.map[0] = photon_dmap(ctx->photons, 0);
  G.map[1] = photon_dmap(ctx->photons, 1);
  G.radius = RP.photon_radius;
  G.ellipticity[0] = RP.ellipticity_global;
  G.ellipticity[1] = RP.ellipticity_caustic;
  G.k = RP.photon_k;
  G.depth = photon_tree_depth(std::max(ctx->photons.count[0], ctx->photons.count[1]));
  G.retry = ctx->d_gather_retry;
  G.retry_seed = ctx->d_gather_seed;
  G.retry_counts = retry_counts;
  G.pilot = 0;
  // one stage-1 / stage-2 launch pair (or the single stage-0 launch); the waves claim the queues through cursors that start at 0
  auto stage = [&](int st) {
    G.cursor = ctx->d_gather_cursor + (st == 2 ? CNT_SLAB : 0);
    (void)hipMemsetAsync(G.cursor, 0, CNT_SLAB * sizeof(uint32_t), ctx->stream);
    if (st == 0) LAUNCH_GATHER((k_photon_gather<CNT, 0>), upper, G.depth, G, F, GQ);
    else if (st == 1) LAUNCH_GATHER((k_photon_gather<CNT, 1>), upper, G.depth, G, F, GQ);
    else LAUNCH_GATHER((k_photon_gather<CNT, 2>), upper, G.depth, G, F, GQ);
  };
  if (G.hints && ctx->gather_hints_cold) {
    // no hints yet (first pass after a commit): pilot lookups from scratch, then the rest from the hints they leave
    G.pilot = 1;
    stage(1);
    stage(2);
    (void)hipMemsetAsync(retry_counts, 0, CNT_SLAB * sizeof(uint32_t), ctx->stream);
    G.pilot = 2;
    stage(1);
    stage(2);
  } else if (G.hints) {
    stage(1);
    stage(2);
  } else {
    stage(0);
  }
}

// Queue capacities are entries per camera sample of a pass, by kind: f[0] ray queue (levels >= 1), f[1] hard-shadow
// groups, f[2] soft-shadow groups, f[3] photon gathers.  The worst case (every shade point spawns every child) is
// branch^(levels-1) entries per sample; real scenes stay far below it, so the capacities come from a measured pilot
// pass (accumulate_internal) with a margin, and a pass that still overflows is split and re-run.
struct FramePlan {
  int W, H;
  int levels;         // number of shading levels (max bounce depth + 1)
  int branch;
  uint32_t budget;    // camera samples per pass
  int n_hard, n_soft; // light counts by shadow kind under these params
  bool photons;
  bool meshes;        // the scene has mesh instances: the second (mesh) phase of the trace kernels runs
  double f[10];       // + f[4] closest-hit mesh items, f[5] hard-shadow mesh items, f[6] soft-shadow mesh items,
                      //   f[7] / f[8] / f[9] their candidates (mesh instances per item's ray)
};

int ensure_frame(rt_context* ctx, const FramePlan& fp) {
  size_t key[16] = {(size_t)fp.W, (size_t)fp.H, (size_t)fp.levels, (size_t)fp.branch, fp.budget, (size_t)fp.n_hard, (size_t)fp.n_soft,
                    (size_t)fp.photons, (size_t)(fp.f[0] * 4096), (size_t)(fp.f[1] * 4096), (size_t)(fp.f[2] * 4096), (size_t)(fp.f[3] * 4096),
                    (size_t)(fp.f[4] * 4096) + ((size_t)(fp.f[7] * 4096) << 32), (size_t)(fp.f[5] * 4096) + ((size_t)(fp.f[8] * 4096) << 32),
                    (size_t)(fp.f[6] * 4096) + ((size_t)(fp.f[9] * 4096) << 32), (size_t)fp.meshes};
  if (!ctx->frame_mem.empty() && !std::memcmp(key, ctx->frame_key, sizeof key)) return RT_OK;
  free_pool(ctx->frame_mem);
  std::memset(ctx->frame_key, 0, sizeof ctx->frame_key);
  // k_raygen_trace deals 64-path tiles round-robin to NSEG segments; children stay in their parent's segment
  size_t tiles = ((size_t)fp.budget + 63) / 64;
  size_t seg0 = ((tiles + NSEG - 1) / NSEG) * 64;
  auto cap_of = [&](double f) { return (size_t)std::min<double>(std::ceil((double)seg0 * f) + 64.0, 3.0e8 / NSEG); };
  size_t segcap = cap_of(std::max(1.0, fp.f[0]));
  size_t qcap = segcap * NSEG;
  int rc;
  void* p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, CNT_TOTAL * sizeof(uint32_t), &p))) return rc;
  ctx->d_counts = (uint32_t*)p;
  for (int k = 0; k < 2; k++) {
    RayQ& q = ctx->Q[k];
    q.segcap = (uint32_t)segcap;
    q.nseg = NSEG;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.q0 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.q1 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.q2 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.q3 = (uint4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.r0 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 4, &p))) return rc; q.rf = (uint32_t*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.h0 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.h1 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, qcap * 16, &p))) return rc; q.h2 = (float4*)p;
    q.counts = nullptr;
  }
  auto alloc_group = [&](GroupQ& g, double f, bool with_g2) {
    size_t sc = f > 0 ? cap_of(f) : 1;
    g.segcap = (uint32_t)sc;
    g.nseg = NSEG;
    size_t cap = sc * NSEG;
    int r;
    void* pp;
    if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 16, &pp))) return r; g.g0 = (float4*)pp;
    if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 16, &pp))) return r; g.g1 = (float4*)pp;
    g.g2 = g.g3 = g.g4 = nullptr;
    g.hint = nullptr;
    if (with_g2) {
      if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 8, &pp))) return r; g.hint = (uint2*)pp;
      HIP_TRY(hipMemsetAsync(pp, 0xff, cap * 8, ctx->stream));   // (no hint; the array is never cleared again: any value is only a hint)
      if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 16, &pp))) return r; g.g2 = (float4*)pp;
      if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 16, &pp))) return r; g.g3 = (float4*)pp;
      if ((r = alloc_dev(ctx, ctx->frame_mem, cap * 16, &pp))) return r; g.g4 = (float4*)pp;
    }
    g.counts = nullptr;
    return (int)RT_OK;
  };
  if ((rc = alloc_group(ctx->GH, fp.n_hard ? fp.f[1] : 0.0, false))) return rc;
  if ((rc = alloc_group(ctx->GS, fp.n_soft ? fp.f[2] : 0.0, true))) return rc;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, (size_t)ctx->GS.segcap * NSEG * 4, &p))) return rc;
  ctx->d_list2 = (uint32_t*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, (size_t)ctx->GS.segcap * NSEG * 4, &p))) return rc;
  ctx->d_list1 = (uint32_t*)p;
  size_t gsegcap = fp.photons ? cap_of(fp.f[3]) : 1;
  ctx->GQ.segcap = (uint32_t)gsegcap;
  ctx->GQ.nseg = NSEG;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, gsegcap * NSEG * 16, &p))) return rc; ctx->GQ.a = (float4*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, gsegcap * NSEG * 16, &p))) return rc; ctx->GQ.b = (float4*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, gsegcap * NSEG * 16, &p))) return rc; ctx->GQ.c = (float4*)p;
  ctx->GQ.counts = nullptr;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, 2 * CNT_SLAB * sizeof(uint32_t), &p))) return rc; ctx->d_gather_cursor = (uint32_t*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, gsegcap * NSEG * 4, &p))) return rc; ctx->d_gather_retry = (uint32_t*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, gsegcap * NSEG * 4, &p))) return rc; ctx->d_gather_seed = (float*)p;
  auto alloc_meshq = [&](MeshQ& q, double f, double fc) {
    size_t sc = (fp.meshes && f > 0) ? cap_of(f) : 1;
    size_t cc = (fp.meshes && f > 0) ? cap_of(std::max(f, fc)) : 1;
    q.segcap = (uint32_t)sc;
    q.cand_segcap = (uint32_t)cc;
    q.nseg = NSEG;
    q.counts = q.cursor = q.cand_counts = nullptr;
    void* pp;
    int r = alloc_dev(ctx, ctx->frame_mem, sc * NSEG * 16, &pp);
    q.item4 = (uint4*)pp;
    if (r) return r;
    r = alloc_dev(ctx, ctx->frame_mem, cc * NSEG * 32, &pp);
    q.cand = (float4*)pp;
    return r;
  };
  if ((rc = alloc_meshq(ctx->MQ, fp.f[4], fp.f[7]))) return rc;
  {  // the rays of those items (same index, same segments): ray, weights, ids, and the raw hit k_trace_mesh leaves
    RayQ& q = ctx->QM;
    const size_t mcap = (size_t)ctx->MQ.segcap * NSEG;
    q.segcap = ctx->MQ.segcap;
    q.nseg = NSEG;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 16, &p))) return rc; q.q0 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 16, &p))) return rc; q.q1 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 16, &p))) return rc; q.q2 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 16, &p))) return rc; q.q3 = (uint4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 16, &p))) return rc; q.r0 = (float4*)p;
    if ((rc = alloc_dev(ctx, ctx->frame_mem, mcap * 4, &p))) return rc; q.rf = (uint32_t*)p;
    q.h0 = q.h1 = q.h2 = nullptr;   // (re-shade entries never wait for the mesh phase)
    q.counts = nullptr;
  }
  if ((rc = alloc_meshq(ctx->MH, fp.n_hard ? fp.f[5] : 0.0, fp.f[8]))) return rc;
  if ((rc = alloc_meshq(ctx->MS, fp.n_soft ? fp.f[6] : 0.0, fp.f[9]))) return rc;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, (size_t)fp.budget * 12, &p))) return rc;
  ctx->d_path_rad = (float*)p;
  ctx->rad_stride = (uint32_t)fp.budget;
  size_t npx = (size_t)fp.W * fp.H;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, npx * 4, &p))) return rc; ctx->d_pixels = (uint32_t*)p;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, npx * 8, &p))) return rc; ctx->d_pixels2 = (uint32_t*)p;
  // the hit-id images (cleared to -1) in one allocation, the geometry buffers (cleared to 0) in another: two clears per call
  // instead of five (a C1 step is ~12 API calls long; each costs more than its kernel)
  if ((rc = alloc_dev(ctx, ctx->frame_mem, npx * 8, &p))) return rc;
  ctx->d_hit_node = (int32_t*)p; ctx->d_hit_face = (int32_t*)p + npx;
  if ((rc = alloc_dev(ctx, ctx->frame_mem, npx * 28, &p))) return rc;
  ctx->d_aov_depth = (float*)p; ctx->d_aov_normal = (float*)p + npx; ctx->d_aov_albedo = (float*)p + 4 * npx;
  std::memcpy(ctx->frame_key, key, sizeof key);
  std::memset(ctx->pixel_key, 0, sizeof ctx->pixel_key);
  ctx->path_budget = fp.budget;
  return RT_OK;
}

// Pixel list of the requested 32x32 tiles (raytracer.h:46, RayTracer.cpp:103-115), walked in 8x8 blocks so that
// one wavefront's 64 camera rays cover a compact screen patch.
void build_pixel_list(const rt_render_params& P, int W, int H, std::vector<uint32_t>& out) {
  const int tileSize = 32;
  int tilesX = (W + tileSize - 1) / tileSize, tilesY = (H + tileSize - 1) / tileSize;
  int total = tilesX * tilesY;
  int stride = P.tile_stride > 0 ? P.tile_stride : 1;
  out.clear();
  for (int t = P.tile_begin, k = 0; t < total; t += stride, k++) {
    if (P.tile_count > 0 && k >= P.tile_count) break;
    if (t < 0) continue;
    int x0 = (t % tilesX) * tileSize, y0 = (t / tilesX) * tileSize;
    int x1 = std::min(x0 + tileSize, W), y1 = std::min(y0 + tileSize, H);
    for (int by = y0; by < y1; by += 8)
      for (int bx = x0; bx < x1; bx += 8)
        for (int y = by; y < std::min(by + 8, y1); y++)
          for (int x = bx; x < std::min(bx + 8, x1); x++) out.push_back((uint32_t)(y * W + x));
  }
}

template <bool CNT>
int run_pass(rt_context* ctx, const PassP& P, const FrameBufs& F, int levels, bool any_soft, bool any_hard, bool photons,
             const rt_render_params& RP) {
  HIP_TRY(hipMemsetAsync(ctx->d_counts, 0, CNT_TOTAL * sizeof(uint32_t), ctx->stream));
  uint32_t n0 = P.n_pix * (uint32_t)P.n_samples;
  RayQ Qc = ctx->Q[0], Qn = ctx->Q[1];
  Qc.counts = ctx->d_counts + cnt_at(0, CNT_Q);
  const bool meshes = ctx->n_minst > 0;
  auto meshq = [&](const MeshQ& base, int lvl, int kind_cnt, int kind_cur) {
    MeshQ q = base;
    q.counts = ctx->d_counts + cnt_at(lvl, kind_cnt);
    q.cursor = ctx->d_counts + cnt_at(lvl, kind_cur);
    q.cand_counts = ctx->d_counts + cnt_at(lvl, kind_cnt == CNT_MQ ? CNT_CQ : (kind_cnt == CNT_MH ? CNT_CH : (kind_cnt == CNT_MS1 ? CNT_CS1 : CNT_CS2)));
    return q;
  };
  // (measured, MI355X: fused C1 0.196 -> 0.174, C3 46.4 -> 42.8, C4 43.8 -> 42.8, C2 15.94 -> 15.77 ms per step; scenes whose
  //  closest-hit rays walk the instance BVH per lane lose — 2944 spheres 7.19 -> 7.98 ms: that walk wants the trace kernels'
  //  higher occupancy — and keep the separate kernels)
  const bool fuse = ctx->opt.fuse != 0 && !ctx->ds.use_tlas;
  const bool mc = P.mc_bounces > 0, ph = P.photons != 0;
  const bool xm = P.extensions != 0;
  // occluder hints for the soft-shadow mesh walks (k_shadow_mesh): they pay where finding an occluder nearer-child-first
  // takes many more steps than a tree is deep — large meshes — and cost ~4 % where it does not
  const bool hints = ctx->opt.shadow_hints > 0 || (ctx->opt.shadow_hints < 0 && ctx->max_mesh_depth >= RT_SHADOW_HINT_DEPTH);
#define RT_SHADE(n_, Q_)                                                                                        \
  do {                                                                                                          \
    if (xm) {                                                                                                   \
      if (mc && ph) LAUNCH((k_shade<CNT, true, true, true>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);            \
      else if (mc) LAUNCH((k_shade<CNT, true, false, true>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);            \
      else if (ph) LAUNCH((k_shade<CNT, false, true, true>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);            \
      else LAUNCH((k_shade<CNT, false, false, true>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);                   \
    } else if (mc && ph) LAUNCH((k_shade<CNT, true, true, false>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);      \
    else if (mc) LAUNCH((k_shade<CNT, true, false, false>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);             \
    else if (ph) LAUNCH((k_shade<CNT, false, true, false>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);             \
    else LAUNCH((k_shade<CNT, false, false, false>), n_, ctx->ds, P, F, Q_, Qn, GH, GS, GQ);                    \
  } while (0)
  if (!fuse) {  // camera rays + their first phase (instance table)
    MeshQ MQ = meshq(ctx->MQ, 0, CNT_MQ, CUR_MQ);
    Timer t(ctx, RT_K_RAYGEN);
    if (ctx->ds.use_tlas) LAUNCH_T((k_raygen_trace<CNT, true>), n0, ctx->ds, P, F, Qc, MQ);
    else LAUNCH_P1((k_raygen_trace<CNT, false>), n0, ctx->ds, P, F, Qc, MQ);
    t.stop((1ull << 32) + n0);
  }
  double ub = n0;
  uint64_t gather_ub = 0;
  int branch = 2 + (P.mc_bounces > 0 ? 1 : 0);
  const uint64_t qcap = (uint64_t)Qc.segcap * Qc.nseg;
  for (int lvl = 0; lvl < levels; lvl++) {
    Qn.counts = ctx->d_counts + cnt_at(lvl + 1, CNT_Q);
    GroupQ GH = ctx->GH, GS = ctx->GS;
    GatherQ GQ = ctx->GQ;
    // ONE set of shadow-group queues for the pass, too: every level appends, the shadow stage runs once after the last level
    // (it only adds radiance to the paths).  A scene's deeper levels hold few groups, and a launch of the two-phase shadow
    // kernels is never shorter than its longest walk: finalProject spent 7.6 of 31 ms in the small launches of levels 1 - 3.
    GH.counts = ctx->d_counts + cnt_at(0, CNT_GH);
    GS.counts = ctx->d_counts + cnt_at(0, CNT_GS);
    // ONE lookup queue for the pass: every level appends to it and it is served once, after the last level.  A launch of the
    // gather is as long as its longest lookup (a chain of dependent fetches), so four small launches cost four such tails.
    GQ.counts = ctx->d_counts + cnt_at(0, CNT_GQ);
    uint64_t ubn = (uint64_t)std::min<double>(ub, (double)qcap);
    if (fuse) {
      // first trace phase + shading of the rays it finishes in ONE kernel; the rays that reach a mesh box go through QM / MQ
      // (entry and work item at the same index: the queue's length words are the item queue's)
      MeshQ MQ = meshq(ctx->MQ, lvl, CNT_MQ, CUR_MQ);
      RayQ QM = ctx->QM;
      QM.counts = MQ.counts;
#define RT_FUSED(kern, ...)                                                                                     \
      do {                                                                                                      \
        if (xm) {   /* RT_EXT_* variants: this build's own materials / filtered texture lookups compiled in */   \
          if (mc && ph) LAUNCH_P1((kern<CNT, false, true, true, true>), __VA_ARGS__);                           \
          else if (mc) LAUNCH_P1((kern<CNT, false, true, false, true>), __VA_ARGS__);                           \
          else if (ph) LAUNCH_P1((kern<CNT, false, false, true, true>), __VA_ARGS__);                           \
          else LAUNCH_P1((kern<CNT, false, false, false, true>), __VA_ARGS__);                                  \
        } else if (mc && ph) LAUNCH_P1((kern<CNT, false, true, true, false>), __VA_ARGS__);                     \
        else if (mc) LAUNCH_P1((kern<CNT, false, true, false, false>), __VA_ARGS__);                            \
        else if (ph) LAUNCH_P1((kern<CNT, false, false, true, false>), __VA_ARGS__);                            \
        else LAUNCH_P1((kern<CNT, false, false, false, false>), __VA_ARGS__);                                   \
      } while (0)
      if (lvl == 0) {
        Timer t(ctx, RT_K_RAYGEN);
        RT_FUSED(k_raygen_shade, n0, ctx->ds, P, F, ctx->d_counts + cnt_at(0, CNT_Q), QM, MQ, Qn, GH, GS, GQ);
        t.stop((1ull << 32) + n0);
      } else {
        Timer t(ctx, RT_K_TRACE_CLOSEST);
        RT_FUSED(k_trace_shade, ubn, ctx->ds, P, F, Qc, QM, MQ, Qn, GH, GS, GQ);
        t.stop((uint64_t)(lvl * CNT_KINDS + CNT_Q));
      }
#undef RT_FUSED
      if (meshes) {
        const uint64_t mub = std::min<uint64_t>(ubn, (uint64_t)MQ.segcap * MQ.nseg);
        {
          Timer t2(ctx, RT_K_TRACE_MESH);
          LAUNCH_T(k_trace_mesh<CNT>, mub, ctx->ds, P, F, QM, MQ, lvl);
          t2.stop((uint64_t)(lvl * CNT_KINDS + CNT_MQ));
        }
        Timer t(ctx, RT_K_SHADE);
        RT_SHADE(mub, QM);
        t.stop((uint64_t)(lvl * CNT_KINDS + CNT_MQ));
      }
    } else {
      {
        MeshQ MQ = meshq(ctx->MQ, lvl, CNT_MQ, CUR_MQ);
        if (lvl > 0) {  // (level 0 went through k_raygen_trace)
          Timer t(ctx, RT_K_TRACE_CLOSEST);
          if (ctx->ds.use_tlas) LAUNCH_T((k_trace_closest<CNT, true>), ubn, ctx->ds, P, F, Qc, MQ, lvl);
          else LAUNCH_P1((k_trace_closest<CNT, false>), ubn, ctx->ds, P, F, Qc, MQ, lvl);
          t.stop((uint64_t)(lvl * CNT_KINDS + CNT_Q));
        }
        if (meshes) {
          Timer t2(ctx, RT_K_TRACE_MESH);
          LAUNCH_T(k_trace_mesh<CNT>, std::min<uint64_t>(ubn, (uint64_t)MQ.segcap * MQ.nseg), ctx->ds, P, F, Qc, MQ, lvl);
          t2.stop((uint64_t)(lvl * CNT_KINDS + CNT_MQ));
        }
      }
      {
        Timer t(ctx, RT_K_SHADE);
        RT_SHADE(ubn, Qc);
        t.stop((uint64_t)(lvl * CNT_KINDS + CNT_Q));
      }
    }
    gather_ub += ubn;
    std::swap(Qc, Qn);
    ub *= branch;
  }
  {   // the shadow stage of the whole pass
    const int lvl = 0;   // (the count slots of the one queue set)
    GroupQ GH = ctx->GH, GS = ctx->GS;
    GH.counts = ctx->d_counts + cnt_at(0, CNT_GH);
    GS.counts = ctx->d_counts + cnt_at(0, CNT_GS);
    uint32_t* n2 = ctx->d_counts + cnt_at(0, CNT_N2);
    const uint64_t ubn = std::min<uint64_t>(gather_ub, 0xffffffffull);   // entries of all levels
    if (any_hard) {
      Timer t(ctx, RT_K_SHADOW);
      MeshQ MH = meshq(ctx->MH, lvl, CNT_MH, CUR_MH);
      uint64_t gh_ub = std::min<uint64_t>(ubn * 4, (uint64_t)GH.segcap * GH.nseg);
      if (ctx->ds.shadow_tlas) LAUNCH_T((k_shadow_hard<CNT, true>), gh_ub, ctx->ds, P, F, GH, MH);
      else LAUNCH_P1((k_shadow_hard<CNT, false>), gh_ub, ctx->ds, P, F, GH, MH);
      t.stop((uint64_t)(lvl * CNT_KINDS + CNT_GH));
      if (meshes) {
        Timer t2(ctx, RT_K_SHADOW_MESH);
        LAUNCH_T((k_shadow_mesh<CNT, false, false>), std::min<uint64_t>(gh_ub, (uint64_t)MH.segcap * MH.nseg), ctx->ds, P, F, GH, MH);
        t2.stop((uint64_t)(lvl * CNT_KINDS + CNT_MH));
      }
    }
    if (any_soft) {
      uint64_t gs_ub = std::min<uint64_t>(ubn * 4, (uint64_t)GS.segcap * GS.nseg);
      MeshQ MS1 = meshq(ctx->MS, lvl, CNT_MS1, CUR_MS1), MS2 = meshq(ctx->MS, lvl, CNT_MS2, CUR_MS2);
      const uint64_t ms_cap = (uint64_t)MS1.segcap * MS1.nseg;
      // groups with a clear view of their whole light are fully lit, groups in a sphere's full shadow dark, without a ray
      // (the proof walks the instance table — one box test per instance for most — or, in scenes of more than
      // RT_CONE_MAX_INST instances, the instance BVH with the cone in place of a ray)
      const bool cone_table = ctx->ds.n_inst <= (int)ctx->opt.cone_max_inst;
      const bool clear_first = ctx->opt.soft_clear != 0 && (cone_table || ctx->ds.use_tlas);
      const bool soft_tlas = ctx->ds.shadow_tlas != 0;
      uint32_t* n1 = ctx->d_counts + cnt_at(lvl, CNT_N1);
      const uint32_t* list1 = clear_first ? ctx->d_list1 : nullptr;
      {
        Timer t(ctx, RT_K_SHADOW);
        if (clear_first && cone_table) LAUNCH_P1((k_shadow_soft_clear<CNT, false>), gs_ub, ctx->ds, P, F, GS, ctx->d_list1, n1);
        else if (clear_first) LAUNCH_T((k_shadow_soft_clear<CNT, true>), gs_ub, ctx->ds, P, F, GS, ctx->d_list1, n1);
        if (soft_tlas)
          LAUNCH_T((k_shadow_soft<CNT, true>), gs_ub * (uint64_t)P.shadow_min, ctx->ds, P, F, GS, list1, (const uint32_t*)(list1 ? n1 : nullptr), 0u,
                   (uint32_t)P.shadow_min, MS1);
        else
          LAUNCH_P1((k_shadow_soft<CNT, false>), gs_ub * (uint64_t)P.shadow_min, ctx->ds, P, F, GS, list1, (const uint32_t*)(list1 ? n1 : nullptr), 0u,
                    (uint32_t)P.shadow_min, MS1);
        t.stop((uint64_t)(lvl * CNT_KINDS + CNT_GS));
      }
      if (meshes) {
        Timer t(ctx, RT_K_SHADOW_MESH);
        if (hints) LAUNCH_T((k_shadow_mesh<CNT, true, true>), std::min<uint64_t>(gs_ub * (uint64_t)P.shadow_min, ms_cap), ctx->ds, P, F, GS, MS1);
        else LAUNCH_T((k_shadow_mesh<CNT, true, false>), std::min<uint64_t>(gs_ub * (uint64_t)P.shadow_min, ms_cap), ctx->ds, P, F, GS, MS1);
        t.stop((uint64_t)(lvl * CNT_KINDS + CNT_MS1));
      }
      LAUNCH(k_shadow_soft_decide, gs_ub, P, F, GS, list1, (const uint32_t*)n1, ctx->d_list2, n2);
      if (P.shadow_max > P.shadow_min) {
        // The remaining samples of the groups that are not fully lit (lights.cpp:56-59).  With meshes in the scene they are
        // taken RT_SOFT_CHUNK samples at a time: every undecided ray is an item + candidates of the mesh phase (48 bytes),
        // and 112 of them per group at once decided how many camera samples fit a pass (C5: 92 passes per step instead of 5).
        // The shorter first chunk leaves a full-sized one for the pass summary's queue lengths.
        const uint32_t nr_all = (uint32_t)(P.shadow_max - P.shadow_min);
        const uint32_t chunk = meshes ? std::min<uint32_t>(nr_all, RT_SOFT_CHUNK) : nr_all;
        uint32_t j0 = (uint32_t)P.shadow_min;
        for (uint32_t done = 0; done < nr_all;) {
          const uint32_t nr = (done == 0 && nr_all % chunk) ? nr_all % chunk : chunk;
          if (done > 0) {   // the mesh phase's queues start empty again
            HIP_TRY(hipMemsetAsync(ctx->d_counts + cnt_at(lvl, CNT_MS2), 0, CNT_SLAB * sizeof(uint32_t), ctx->stream));
            HIP_TRY(hipMemsetAsync(ctx->d_counts + cnt_at(lvl, CUR_MS2), 0, CNT_SLAB * sizeof(uint32_t), ctx->stream));
            HIP_TRY(hipMemsetAsync(ctx->d_counts + cnt_at(lvl, CNT_CS2), 0, CNT_SLAB * sizeof(uint32_t), ctx->stream));
          }
          {
            Timer t(ctx, RT_K_SHADOW);
            if (soft_tlas)
              LAUNCH_T((k_shadow_soft<CNT, true>), gs_ub * (uint64_t)nr, ctx->ds, P, F, GS, (const uint32_t*)ctx->d_list2, (const uint32_t*)n2, j0, nr, MS2);
            else
              LAUNCH_P1((k_shadow_soft<CNT, false>), gs_ub * (uint64_t)nr, ctx->ds, P, F, GS, (const uint32_t*)ctx->d_list2, (const uint32_t*)n2, j0, nr, MS2);
            t.stop((uint64_t)(lvl * CNT_KINDS + CNT_N2));
          }
          if (meshes) {
            Timer t(ctx, RT_K_SHADOW_MESH);
            if (hints) LAUNCH_T((k_shadow_mesh<CNT, true, true>), std::min<uint64_t>(gs_ub * (uint64_t)nr, ms_cap), ctx->ds, P, F, GS, MS2);
            else LAUNCH_T((k_shadow_mesh<CNT, true, false>), std::min<uint64_t>(gs_ub * (uint64_t)nr, ms_cap), ctx->ds, P, F, GS, MS2);
            t.stop((uint64_t)(lvl * CNT_KINDS + CNT_MS2));
          }
          j0 += nr;
          done += nr;
        }
        LAUNCH(k_shadow_soft_final, gs_ub, P, F, GS, (const uint32_t*)ctx->d_list2, (const uint32_t*)n2);
      }
    }
  }
  if (photons) {
    GatherQ GQ = ctx->GQ;
    GQ.counts = ctx->d_counts + cnt_at(0, CNT_GQ);
    Timer t(ctx, RT_K_PHOTON_GATHER);
    photon_gather_queue<CNT>(ctx, GQ, F, P, RP, std::min<uint64_t>(gather_ub, (uint64_t)GQ.segcap * GQ.nseg), ctx->d_counts + cnt_at(0, CNT_GR));
    t.stop((uint64_t)CNT_GQ);
  }
  if (photons && !P.discard) ctx->gather_hints_cold = false;   // (a sizing pass covers a pixel subset only)
  {
    Timer t(ctx, RT_K_ACCUM);
    LAUNCH(k_accumulate, P.n_pix, P, F);
    t.stop((1ull << 32) + P.n_pix);
  }
  return RT_OK;
}

int check_params(rt_context* ctx, const rt_render_params& P) {
  if (P.sampler != RT_SAMPLER_PHILOX)
    return fail(ctx, RT_ERR_UNSUPPORTED, "the device integrator draws from Philox streams only (RT_SAMPLER_PCG_HALTON is sequential per pixel)");
  if (P.spp_max <= 0 || P.spp_min < 0 || P.spp_min > P.spp_max) return fail(ctx, RT_ERR_INVALID, "bad spp_min/spp_max");
  if (P.max_bounces < 0 || P.max_bounces > 5 || P.mc_bounces < 0 || P.mc_bounces > 5) return fail(ctx, RT_ERR_INVALID, "bounce limits must be in 0..5");
  if (P.mc_bounces > 0 && P.mc_samples != 1) return fail(ctx, RT_ERR_UNSUPPORTED, "mc_samples must be 1 (renderer.h:172)");
  if (P.shadow_min <= 0 || P.shadow_max < P.shadow_min) return fail(ctx, RT_ERR_INVALID, "bad shadow_min/shadow_max");
  if (P.sample_begin < 0 || P.sample_count < 0 || P.sample_begin + P.sample_count > 65536) return fail(ctx, RT_ERR_INVALID, "bad sample range");
  if (P.extensions & ~(RT_EXT_MATERIALS | RT_EXT_TEX_FILTER)) return fail(ctx, RT_ERR_INVALID, "unknown bits in extensions");
  return RT_OK;
}

// Adds the requested samples into d_accum (W*H*8 floats).  Writes AOVs into the context's device AOV buffers.
int accumulate_internal(rt_context* ctx, const rt_render_params& RP, float* d_accum, bool track_progress) {
  int rc = check_params(ctx, RP);
  if (rc) return rc;
  const rt_camera& c = ctx->camera;
  int W = RP.width > 0 ? RP.width : c.width, H = RP.height > 0 ? RP.height : c.height;
  if (W <= 0 || H <= 0 || (long long)W * H > (1ll << 28)) return fail(ctx, RT_ERR_INVALID, "bad image size");
  int begin = RP.sample_begin;
  int count = RP.sample_count > 0 ? RP.sample_count : RP.spp_max - begin;
  if (count <= 0) return fail(ctx, RT_ERR_INVALID, "empty sample range");
  bool adaptive = RP.spp_min < RP.spp_max && begin == 0 && count == RP.spp_max;

  int n_hard = 0, n_soft = 0;
  for (auto& l : ctx->lights) {
    if (l.kind == RT_LIGHT_DIRECT) n_hard++;
    else if (l.kind == RT_LIGHT_POINT) { if (l.size * RP.light_size_scale != 0.0f) n_soft++; else n_hard++; }
  }
  bool photons = RP.photons != 0 && RP.mc_bounces == 0;
  bool photons_mixed = RP.photons != 0;  // below the MC depth the photon branch is taken (materials.cpp:313-322)
  if (photons_mixed) {
    rc = photon_ensure_built(ctx->photons, ctx, RP);
    if (rc) return rc;
  }
  (void)photons;
  FramePlan fp;
  fp.W = W; fp.H = H;
  fp.levels = std::max(RP.max_bounces, RP.mc_bounces) + 1;
  if (fp.levels > MAX_LEVELS - 1) return fail(ctx, RT_ERR_UNSUPPORTED, "too many bounce levels");
  fp.branch = 2 + (RP.mc_bounces > 0 ? 1 : 0);
  fp.n_hard = n_hard; fp.n_soft = n_soft; fp.photons = photons_mixed;
  const double worst = std::pow((double)fp.branch, fp.levels - 1);
  const bool meshes = ctx->n_minst > 0;
  const double soft_rays = std::max<int>(RP.shadow_min, std::min<int>(RP.shadow_max - RP.shadow_min, RT_SOFT_CHUNK));   // mesh-phase items of one launch
  const double nmi = std::max(1, ctx->n_minst);   // candidates per item: at most every mesh instance
  double worst_sum = 0.0;   // the shadow groups and the photon lookups of all levels share one queue each
  for (int l = 0; l < fp.levels; l++) worst_sum += std::pow((double)fp.branch, l);
  const double fworst[NFAN] = {worst, worst_sum * n_hard, worst_sum * n_soft, photons_mixed ? worst_sum : 0.0,
                            meshes ? worst : 0.0, meshes ? worst_sum * n_hard : 0.0, meshes ? worst_sum * n_soft * soft_rays : 0.0,
                            meshes ? worst * nmi : 0.0, meshes ? worst_sum * n_hard * nmi : 0.0, meshes ? worst_sum * n_soft * soft_rays * nmi : 0.0};
  auto per_path_bytes = [&](const double* f) {
    return 2.0 * 132 * std::max(1.0, f[0]) + 32.0 * f[1] + 84.0 * f[2] + 48.0 * f[3] + 16.0 * (f[4] + f[5] + f[6]) +
           32.0 * (std::max(f[4], f[7]) + std::max(f[5], f[8]) + std::max(f[6], f[9])) + 16.0 + 64.0;
  };
  // memory the frame resources (stage queues) may take: large passes amortise the per-launch tails of the
  // traversal kernels, and HBM is 288 GB
  double frame_bytes = RP.frame_mem_mb > 0 ? (double)RP.frame_mem_mb * 1048576.0 : 32.0e9;
  if (ctx->opt.frame_gb > 0) frame_bytes = ctx->opt.frame_gb * 1e9;  // developer knob
  auto budget_for = [&](const double* f) {
    double b = frame_bytes / per_path_bytes(f);
    uint32_t v = (uint32_t)std::min<double>(std::max<double>(b, 65536.0), 201326592.0);
    return v & ~63u;
  };
  size_t fkey[10] = {(size_t)W, (size_t)H, (size_t)RP.max_bounces, (size_t)RP.mc_bounces, (size_t)RP.pixel_jitter, (size_t)n_hard,
                     (size_t)n_soft, (size_t)photons_mixed, (size_t)(RP.skip_zero_weight != 0) + 2 * (size_t)RP.shadow_min + 512 * (size_t)RP.shadow_max,
                     (size_t)RP.tile_stride + ((size_t)RP.extensions << 32)};
  if (ctx->fan_valid && std::memcmp(fkey, ctx->fan_key, sizeof fkey)) ctx->fan_valid = false;
  const bool need_pilot = !ctx->fan_valid && worst > 1.0;
  if (!need_pilot && !ctx->fan_valid) {  // no secondary rays at all: the worst case is exact
    for (int k = 0; k < NFAN; k++) ctx->fan[k] = fworst[k];
    std::memcpy(ctx->fan_key, fkey, sizeof fkey);
    ctx->fan_valid = true;
  }
  for (int k = 0; k < NFAN; k++) fp.f[k] = need_pilot ? fworst[k] : ctx->fan[k];
  fp.meshes = meshes;
  uint32_t budget = need_pilot ? 131072u : budget_for(fp.f);
  fp.budget = budget;
  if ((rc = ensure_frame(ctx, fp))) return rc;

  // pixel list
  size_t pkey[6] = {(size_t)W, (size_t)H, (size_t)RP.tile_begin + 1, (size_t)RP.tile_count, (size_t)RP.tile_stride, 1};
  if (std::memcmp(pkey, ctx->pixel_key, sizeof pkey)) {
    build_pixel_list(RP, W, H, ctx->h_pixels);
    if (!ctx->h_pixels.empty())
      HIP_TRY(hipMemcpyAsync(ctx->d_pixels, ctx->h_pixels.data(), ctx->h_pixels.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    std::memcpy(ctx->pixel_key, pkey, sizeof pkey);
  }
  uint32_t n_pix_total = (uint32_t)ctx->h_pixels.size();
  if (n_pix_total == 0) return RT_OK;
  HIP_TRY(hipMemsetAsync(ctx->d_hit_node, 0xff, (size_t)W * H * 8, ctx->stream));    // hit_node + hit_face
  HIP_TRY(hipMemsetAsync(ctx->d_aov_depth, 0, (size_t)W * H * 28, ctx->stream));     // depth + normal + albedo

  PassP P;
  std::memset(&P, 0, sizeof P);
  {  // RayTracer.cpp:15-22 and 95-101
    float Zx = -c.dir[0], Zy = -c.dir[1], Zz = -c.dir[2];
    float Yx = c.up[0], Yy = c.up[1], Yz = c.up[2];
    P.Z[0] = Zx; P.Z[1] = Zy; P.Z[2] = Zz;
    P.Y[0] = Yx; P.Y[1] = Yy; P.Y[2] = Yz;
    P.X[0] = Yy * Zz - Yz * Zy; P.X[1] = Yz * Zx - Yx * Zz; P.X[2] = Yx * Zy - Yy * Zx;
    P.pos[0] = c.pos[0]; P.pos[1] = c.pos[1]; P.pos[2] = c.pos[2];
    P.l = c.focaldist;
    P.camW = (float)W; P.camH = (float)H;
    P.wh = (float)(2.0f * P.l * std::tan(((double)c.fov * RT_PI_D / 180.0) / 2.0f));
    P.ww = P.wh * (P.camW / P.camH);
    P.dof = c.dof;
  }
  P.W = W; P.H = H;
  P.jitter = RP.pixel_jitter;
  P.seed_lo = RP.seed_lo; P.seed_hi = RP.seed_hi;
  P.max_bounces = RP.max_bounces; P.mc_bounces = RP.mc_bounces; P.mc_samples = RP.mc_samples;
  P.shadow_min = RP.shadow_min; P.shadow_max = RP.shadow_max;
  P.light_size_scale = RP.light_size_scale;
  P.photons = photons_mixed ? 1 : 0;
  P.srgb = c.srgb;
  P.aov_sample = begin;
  P.spp_min = RP.spp_min; P.spp_max = RP.spp_max; P.adaptive = adaptive ? 1 : 0;
  P.adaptive_threshold = RP.adaptive_threshold;
  P.extensions = RP.extensions;

  FrameBufs F;
  F.path_rad = ctx->d_path_rad;
  F.rad_stride = ctx->rad_stride;
  F.accum = d_accum;
  F.hit_node = ctx->d_hit_node;
  F.hit_face = ctx->d_hit_face;
  F.aov_depth = ctx->d_aov_depth;
  F.aov_normal = ctx->d_aov_normal;
  F.aov_albedo = ctx->d_aov_albedo;
  F.overflow = ctx->d_counts + CNT_OVERFLOW;
  F.counters = ctx->d_counters;

  constexpr uint32_t N_SLOTS = (MAX_LEVELS + 1) * CNT_KINDS;
  std::vector<uint32_t> counts(N_SLOTS);
  double seen[NFAN] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // fullest-segment entries per camera sample, max over the passes of this call
  bool overflowed = false;
  auto finish_pass = [&](uint64_t paths) -> int {
    // the pass in 1 KB: per (level, kind) the sum and the maximum of the segment lengths + the overflow word, folded on the
    // device and copied into pinned host memory (the full table of counters is 1 MB; copying it cost more than a C1 frame)
    // (written straight into the pinned page where the device can address it: a 1 KB copy is a launch of its own, 12 - 20 us,
    //  a tenth of a C1 step)
    uint32_t* const summary_out = ctx->h_summary_dev ? ctx->h_summary_dev : ctx->d_summary;
    LAUNCH(k_pass_summary, N_SLOTS, (const uint32_t*)ctx->d_counts, N_SLOTS, NSEG, (uint32_t)CNT_SLAB, (uint32_t)CNT_OVERFLOW, summary_out);
    if (!ctx->h_summary_dev) HIP_TRY(hipMemcpyAsync(ctx->h_summary, ctx->d_summary, (2 * N_SLOTS + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    const uint32_t* sums = ctx->h_summary;
    const uint32_t* maxs = ctx->h_summary + N_SLOTS;
    const uint32_t ovf = ctx->h_summary[2 * N_SLOTS];
    if (ovf == RT_WALK_ABORT) return fail(ctx, RT_ERR_DEVICE, "a traversal wave exceeded its step limit (internal error)");
    overflowed = ovf != 0;
    if (ctx->profile_on) {
      for (size_t s = 0; s < counts.size(); s++) counts[s] = sums[s];
      collect_profile(ctx, counts);
    }
    auto seg_max = [&](int level, int kind) { return (double)maxs[level * CNT_KINDS + kind]; };
    double l0 = std::max(64.0, seg_max(0, CNT_Q));
    const int kinds[NFAN] = {CNT_Q, CNT_GH, CNT_GS, CNT_GQ, CNT_MQ, CNT_MH, CNT_MS1, CNT_CQ, CNT_CH, CNT_CS1};
    for (int k = 0; k < NFAN; k++)
      for (int l = (k == 0 ? 1 : 0); l <= fp.levels; l++) seen[k] = std::max(seen[k], seg_max(l, kinds[k]) / l0);
    for (int l = 0; l <= fp.levels; l++) {
      seen[6] = std::max(seen[6], seg_max(l, CNT_MS2) / l0);
      seen[9] = std::max(seen[9], seg_max(l, CNT_CS2) / l0);
    }
    if (!overflowed && ctx->counters_on) ctx->host_paths += paths;  // paths = 0 for the sizing pass
    return RT_OK;
  };
  DCounters saved_counters;
  std::function<int(const uint32_t*, uint32_t, int, int, bool)> pass = [&](const uint32_t* pixels, uint32_t n_pix, int sbase, int ns,
                                                                           bool discard) -> int {
    PassP Pp = P;
    Pp.n_pix = n_pix; Pp.sample_base = sbase; Pp.n_samples = ns; Pp.discard = discard ? 1 : 0;
    FrameBufs Ff = F;
    Ff.pixels = pixels;
    if (ctx->counters_on) HIP_TRY(hipMemcpyAsync(&saved_counters, ctx->d_counters, sizeof saved_counters, hipMemcpyDeviceToHost, ctx->stream));
    int r = ctx->counters_on ? run_pass<true>(ctx, Pp, Ff, fp.levels, n_soft > 0, n_hard > 0, photons_mixed, RP)
                             : run_pass<false>(ctx, Pp, Ff, fp.levels, n_soft > 0, n_hard > 0, photons_mixed, RP);
    if (r) return r;
    if ((r = finish_pass(discard ? 0 : (uint64_t)n_pix * ns))) return r;
    if (discard && ctx->counters_on)  // the sizing pass is not part of the frame's work
      HIP_TRY(hipMemcpy(ctx->d_counters, &saved_counters, sizeof saved_counters, hipMemcpyHostToDevice));
    if (!overflowed) return RT_OK;
    // A queue filled up: k_accumulate left the accumulator alone (it checks the flag).  Undo the work counters,
    // remember to size the queues larger from now on, and redo the pass in two halves.
    if (ctx->counters_on) HIP_TRY(hipMemcpy(ctx->d_counters, &saved_counters, sizeof saved_counters, hipMemcpyHostToDevice));
    for (int k = 0; k < NFAN; k++) ctx->fan[k] = std::min(fworst[k], std::max(ctx->fan[k] * 2.0, seen[k] * 1.5));
    if (discard) return RT_OK;
    if (ns > 1) {
      if ((r = pass(pixels, n_pix, sbase, ns / 2, false))) return r;
      return pass(pixels, n_pix, sbase + ns / 2, ns - ns / 2, false);
    }
    if (n_pix > 64) {
      uint32_t h = ((n_pix / 2) + 63u) & ~63u;
      if ((r = pass(pixels, h, sbase, ns, false))) return r;
      return pass(pixels + h, n_pix - h, sbase, ns, false);
    }
    return fail(ctx, RT_ERR_OVERFLOW, "internal queue overflow");
  };

  if (need_pilot) {
    // Pilot: one sample of an evenly strided subset of the 8x8 pixel blocks, results discarded, queues sized for the
    // worst case; its queue lengths size the real passes.
    const uint32_t blocks = (n_pix_total + 63) / 64;
    const uint32_t want = std::min<uint32_t>(blocks, budget / 64);
    std::vector<uint32_t> sub;
    sub.reserve((size_t)want * 64);
    for (uint32_t k = 0; k < want; k++) {
      uint32_t blk = (uint32_t)(((uint64_t)k * blocks) / want);
      uint32_t p0 = blk * 64, p1 = std::min(p0 + 64, n_pix_total);
      sub.insert(sub.end(), ctx->h_pixels.begin() + p0, ctx->h_pixels.begin() + p1);
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_pixels2, sub.data(), sub.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = pass(ctx->d_pixels2, (uint32_t)sub.size(), begin, 1, true))) return rc;
    if (overflowed) return fail(ctx, RT_ERR_OVERFLOW, "internal queue overflow in the sizing pass");
    double margin = 1.5;
    margin *= ctx->opt.fan_scale;  // tests: force the overflow / split path
    for (int k = 0; k < NFAN; k++) ctx->fan[k] = std::min(fworst[k], seen[k] * margin + (fworst[k] > 0 ? 0.03 : 0.0));
    std::memcpy(ctx->fan_key, fkey, sizeof fkey);
    ctx->fan_valid = true;
    for (int k = 0; k < NFAN; k++) fp.f[k] = ctx->fan[k];
    budget = budget_for(fp.f);
    fp.budget = budget;
    if ((rc = ensure_frame(ctx, fp))) return rc;
    // ensure_frame re-created the buffers: pixel list, AOVs and the FrameBufs pointers
    HIP_TRY(hipMemcpyAsync(ctx->d_pixels, ctx->h_pixels.data(), ctx->h_pixels.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    std::memcpy(ctx->pixel_key, pkey, sizeof pkey);
    HIP_TRY(hipMemsetAsync(ctx->d_hit_node, 0xff, (size_t)W * H * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_aov_depth, 0, (size_t)W * H * 28, ctx->stream));
    F.path_rad = ctx->d_path_rad;
    F.rad_stride = ctx->rad_stride;
    F.hit_node = ctx->d_hit_node;
    F.hit_face = ctx->d_hit_face;
    F.aov_depth = ctx->d_aov_depth;
    F.aov_normal = ctx->d_aov_normal;
    F.aov_albedo = ctx->d_aov_albedo;
    F.overflow = ctx->d_counts + CNT_OVERFLOW;
  }

  // An asynchronous render reports progress (rt_render_poll, viewport.cpp:583-586): it takes the frame in batches of whole
  // 32x32 tiles, at least 16 per frame, instead of in the largest pass that fits — GetNumRenderedPixels then advances per
  // tile batch, as the reference's does per tile.
  uint32_t progress_chunk = 0xffffffffu;
  if (track_progress) progress_chunk = std::max<uint32_t>(8u * 1024u, (((n_pix_total + 15u) / 16u) + 1023u) & ~1023u);

  if (!adaptive) {
    // fixed sample count: chunks of pixels x chunks of samples
    uint32_t pix_chunk = std::min<uint32_t>(std::min<uint32_t>(n_pix_total, budget), progress_chunk);
    for (uint32_t p0 = 0; p0 < n_pix_total; p0 += pix_chunk) {
      uint32_t np = std::min(pix_chunk, n_pix_total - p0);
      int ns_max = std::max<int>(1, (int)(budget / np));
      for (int s = 0; s < count; s += ns_max) {
        int ns = std::min(ns_max, count - s);
        if ((rc = pass(ctx->d_pixels + p0, np, begin + s, ns, false))) return rc;
      }
      if (track_progress) ctx->pixels_done += np;
    }
  } else {
    // Adaptive: the stop test runs after every sample i >= spp_min (RayTracer.cpp:180-198).  First spp_min+1 samples
    // for every pixel, then small chunks for the pixels still active; k_accumulate applies the test sample by sample
    // and ignores samples past a pixel's stopping point, so the result equals the sequential loop's.
    uint32_t pix_chunk = std::min<uint32_t>(std::min<uint32_t>(n_pix_total, budget), progress_chunk);
    const int first = std::min(RP.spp_min + 1, RP.spp_max);
    for (uint32_t p0 = 0; p0 < n_pix_total; p0 += pix_chunk) {
      uint32_t np = std::min(pix_chunk, n_pix_total - p0);
      int ns_max = std::max<int>(1, (int)(budget / np));
      for (int s = 0; s < first; s += ns_max)
        if ((rc = pass(ctx->d_pixels + p0, np, s, std::min(ns_max, first - s), false))) return rc;
      // continue with the active subset
      const uint32_t* cur = ctx->d_pixels + p0;
      uint32_t ncur = np;
      uint32_t* bufs[2] = {ctx->d_pixels2, ctx->d_pixels2 + (size_t)W * H};
      int which = 0;
      int s = first;
      while (s < RP.spp_max) {
        HIP_TRY(hipMemsetAsync(ctx->d_counts + CNT_ACTIVE, 0, 4, ctx->stream));
        PassP Pp = P;
        Pp.n_pix = ncur;
        FrameBufs Ff = F;
        Ff.pixels = cur;
        LAUNCH(k_compact_active, ncur, Pp, Ff, bufs[which], ctx->d_counts + CNT_ACTIVE);
        uint32_t nact = 0;
        HIP_TRY(hipMemcpyAsync(&nact, ctx->d_counts + CNT_ACTIVE, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (nact == 0) break;
        cur = bufs[which];
        ncur = nact;
        which ^= 1;
        int ns = std::min<int>(std::min<int>(8, RP.spp_max - s), std::max<int>(1, (int)(budget / ncur)));
        if ((rc = pass(cur, ncur, s, ns, false))) return rc;
        s += ns;
      }
      if (track_progress) ctx->pixels_done += np;
    }
  }
  return RT_OK;
}

int resolve_internal(rt_context* ctx, const rt_render_params& RP, const float* d_accum, rt_framebuffer* fb) {
  const rt_camera& c = ctx->camera;
  int W = RP.width > 0 ? RP.width : c.width, H = RP.height > 0 ? RP.height : c.height;
  size_t npx = (size_t)W * H;
  if (fb->on_device) {
    LAUNCH(k_resolve, npx, W, H, c.srgb, d_accum, (uint8_t*)fb->rgb8, (float*)fb->radiance, (int32_t*)fb->sample_count);
    if (fb->hit_node) HIP_TRY(hipMemcpyAsync(fb->hit_node, ctx->d_hit_node, npx * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (fb->hit_face) HIP_TRY(hipMemcpyAsync(fb->hit_face, ctx->d_hit_face, npx * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (fb->depth) HIP_TRY(hipMemcpyAsync(fb->depth, ctx->d_aov_depth, npx * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (fb->normal) HIP_TRY(hipMemcpyAsync(fb->normal, ctx->d_aov_normal, npx * 12, hipMemcpyDeviceToDevice, ctx->stream));
    if (fb->albedo) HIP_TRY(hipMemcpyAsync(fb->albedo, ctx->d_aov_albedo, npx * 12, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
  }
  void *d_rgb = nullptr, *d_rad = nullptr, *d_sc = nullptr;
  int rc = RT_OK;
  if (fb->rgb8 && (rc = grow_dev(ctx, SC_RES_RGB, npx * 3, &d_rgb))) return rc;
  if (fb->radiance && (rc = grow_dev(ctx, SC_RES_RAD, npx * 16, &d_rad))) return rc;
  if (fb->sample_count && (rc = grow_dev(ctx, SC_RES_CNT, npx * 4, &d_sc))) return rc;
  LAUNCH(k_resolve, npx, W, H, c.srgb, d_accum, (uint8_t*)d_rgb, (float*)d_rad, (int32_t*)d_sc);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && fb->rgb8) e = hipMemcpyAsync(fb->rgb8, d_rgb, npx * 3, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->radiance) e = hipMemcpyAsync(fb->radiance, d_rad, npx * 16, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->sample_count) e = hipMemcpyAsync(fb->sample_count, d_sc, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->hit_node && ctx->d_hit_node) e = hipMemcpyAsync(fb->hit_node, ctx->d_hit_node, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->hit_face && ctx->d_hit_face) e = hipMemcpyAsync(fb->hit_face, ctx->d_hit_face, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->depth && ctx->d_aov_depth) e = hipMemcpyAsync(fb->depth, ctx->d_aov_depth, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->normal && ctx->d_aov_normal) e = hipMemcpyAsync(fb->normal, ctx->d_aov_normal, npx * 12, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && fb->albedo && ctx->d_aov_albedo) e = hipMemcpyAsync(fb->albedo, ctx->d_aov_albedo, npx * 12, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, RT_ERR_DEVICE, hipGetErrorString(e));
  return RT_OK;
}

// The filter passes (rt_denoise.h) over an RGBA image in `a`, ping-pong with `b`; returns the buffer that holds the result.
float4* denoise_passes(rt_context* ctx, int width, int height, float4* a, float4* b, const float4* alb, const float4* nrm, bool timed = true) {
  const size_t n = (size_t)width * height;
  DenoiseP D;
  D.W = width; D.H = height;
  D.inv_sn2 = 1.0f / (RT_DN_SIGMA_N * RT_DN_SIGMA_N);
  D.inv_sa2 = 1.0f / (RT_DN_SIGMA_A * RT_DN_SIGMA_A);
  D.has_albedo = alb ? 1 : 0; D.has_normal = nrm ? 1 : 0;
  float sc = RT_DN_SIGMA_C;
  Timer t(ctx, RT_K_DENOISE, timed);   // (profiling creates events: never on the no-allocation path)
  for (int it = 0; it < RT_DN_ITERS; it++) {
    D.step = 1 << it;
    D.inv_sc2 = 1.0f / (sc * sc);
    LAUNCH(k_dn_atrous, n, D, (const float4*)a, b, alb, nrm);
    std::swap(a, b);
    sc = sc * 0.5f;
  }
  t.stop((1ull << 32) + n);
  return a;
}

int render_blocking(rt_context* ctx, const rt_render_params& RP, rt_framebuffer* fb, bool track) {
  HIP_TRY(hipSetDevice(ctx->device));
  const rt_camera& c = ctx->camera;
  int W = RP.width > 0 ? RP.width : c.width, H = RP.height > 0 ? RP.height : c.height;
  if (W <= 0 || H <= 0) return fail(ctx, RT_ERR_INVALID, "bad image size");
  void* acc = nullptr;   // the frame's accumulator lives in the context from frame to frame
  int rc = grow_dev(ctx, SC_ACCUM, (size_t)W * H * 8 * sizeof(float), &acc);
  if (rc) return rc;
  HIP_TRY(hipMemsetAsync(acc, 0, (size_t)W * H * 8 * sizeof(float), ctx->stream));
  rc = accumulate_internal(ctx, RP, (float*)acc, track);
  if (!rc) rc = resolve_internal(ctx, RP, (const float*)acc, fb);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}

}  // namespace

extern "C" {

int rt_render(rt_context* ctx, const rt_render_params* params, rt_framebuffer* fb) {
  if (!ctx) return RT_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (!params || !fb) return fail(ctx, RT_ERR_INVALID, "null argument");
  quiesce(ctx);
  return render_blocking(ctx, *params, fb, false);
}

int rt_render_begin(rt_context* ctx, const rt_render_params* params, rt_framebuffer* fb) {
  if (!ctx) return RT_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (!params || !fb) return fail(ctx, RT_ERR_INVALID, "null argument");
  if (ctx->worker.joinable()) ctx->worker.join();
  ctx->pixels_done = 0;
  ctx->finished = 0;
  rt_render_params P = *params;
  rt_framebuffer F = *fb;
  ctx->worker = std::thread([ctx, P, F]() mutable {
    ctx->async_status = render_blocking(ctx, P, &F, true);
    ctx->finished = 1;
  });
  return RT_OK;
}

int rt_render_poll(rt_context* ctx, int64_t* pixels_done, int* finished) {
  if (!ctx) return RT_ERR_INVALID;
  if (pixels_done) *pixels_done = ctx->pixels_done.load();
  if (finished) *finished = ctx->finished.load();
  return RT_OK;
}

int rt_render_wait(rt_context* ctx) {
  if (!ctx) return RT_ERR_INVALID;
  if (ctx->worker.joinable()) ctx->worker.join();
  return ctx->async_status;
}

int rt_render_accumulate(rt_context* ctx, const rt_render_params* params, void* d_accum) {
  if (!ctx) return RT_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (!params || !d_accum) return fail(ctx, RT_ERR_INVALID, "null argument");
  if (ctx->worker.joinable()) ctx->worker.join();
  HIP_TRY(hipSetDevice(ctx->device));
  return accumulate_internal(ctx, *params, (float*)d_accum, false);
}

int rt_resolve(rt_context* ctx, const rt_render_params* params, const void* d_accum, rt_framebuffer* fb) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (!params || !d_accum || !fb) return fail(ctx, RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return resolve_internal(ctx, *params, (const float*)d_accum, fb);
}

// Multi-GPU hosts without torch (the reference is a C++ program): the frame's one exchange, an in-place RCCL sum-reduce of
// the accumulator to `root`, enqueued on the context's stream.  RCCL is not linked: ncclReduce is the caller's — passed as a
// function pointer, or looked up among the process's global symbols — so the communicator and the function that uses it
// always come from the same copy of the library.
int rt_reduce_accumulators(rt_context* ctx, void* nccl_comm, void* nccl_reduce_fn, void* d_accum, int32_t width, int32_t height, int root) {
  if (!ctx) return RT_ERR_INVALID;
  if (!nccl_comm || !d_accum || width <= 0 || height <= 0 || root < 0) return fail(ctx, RT_ERR_INVALID, "bad reduce arguments");
#ifdef RT_EMU
  return fail(ctx, RT_ERR_UNSUPPORTED, "no RCCL in the host emulation");
#else
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  // ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t, ncclRedOp_t, int root, ncclComm_t, hipStream_t)
  typedef int (*reduce_fn)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
  reduce_fn fn = (reduce_fn)nccl_reduce_fn;
  if (!fn) fn = (reduce_fn)dlsym(RTLD_DEFAULT, "ncclReduce");
  if (!fn) return fail(ctx, RT_ERR_UNSUPPORTED, "ncclReduce not found: link the host against librccl or pass the function pointer");
  const int nccl_float32 = 7, nccl_sum = 0;   // ncclFloat32 / ncclSum (nccl.h; stable ABI values)
  // Sum rgb and n (floats 0..3 of every pixel) are all a frame of partitioned samples needs: they travel as a packed W*H*4
  // image, half the accumulator.  Send and receive images are separate and the receive image starts as a copy of this rank's
  // own values: ncclReduce writes it on the root only, so unpacking it afterwards gives the root the sums and every other
  // rank its own values back — no rank has to know whether it is the root.
  const size_t npx = (size_t)width * (size_t)height;
  if (npx > 0x7fffffffu) return fail(ctx, RT_ERR_INVALID, "image too large");
  void* buf = nullptr;
  int rc = grow_dev(ctx, SC_REDUCE, 2 * npx * sizeof(float4), &buf);
  if (rc) return rc;
  float4 *send = (float4*)buf, *recv = send + npx;
  LAUNCH(k_acc_pack, npx, (const float4*)d_accum, send, (uint32_t)npx);
  HIP_TRY(hipMemcpyAsync(recv, send, npx * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
  int nrc = fn(send, recv, npx * 4u, nccl_float32, nccl_sum, root, nccl_comm, ctx->stream);
  if (nrc != 0) return fail(ctx, RT_ERR_DEVICE, "ncclReduce failed with code " + std::to_string(nrc));
  LAUNCH(k_acc_unpack, npx, (const float4*)recv, (float4*)d_accum, (uint32_t)npx);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
#endif
}

int rt_photons_build(rt_context* ctx, const rt_render_params* params) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!ctx->has_scene) return fail(ctx, RT_ERR_NOSCENE, "no scene committed");
  if (!params) return fail(ctx, RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->photons.built = false;
  return photon_ensure_built(ctx->photons, ctx, *params);
}

int rt_photons_download(rt_context* ctx, int which, rt_photon* out, int32_t capacity, int32_t* n) {
  if (!ctx || !n) return RT_ERR_INVALID;
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  return photon_download(ctx->photons, ctx, which, false, out, capacity, n);
}

int rt_photons_download_tree(rt_context* ctx, int which, rt_photon* out, int32_t capacity, int32_t* n) {
  if (!ctx || !n) return RT_ERR_INVALID;
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  return photon_download(ctx->photons, ctx, which, true, out, capacity, n);
}

int rt_photons_set(rt_context* ctx, int which, const rt_photon* photons, int32_t n) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return photon_set(ctx->photons, ctx, which, photons, n);
}

int rt_photon_gather(rt_context* ctx, int which, const float* P, const float* N, size_t n, float radius, float ellipticity,
                     int k, float* irradiance) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  if (!P || !N || !irradiance) return fail(ctx, RT_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return photon_gather_api(ctx->photons, ctx, which, P, N, n, radius, ellipticity, k, irradiance);
}

int rt_denoise(rt_context* ctx, const float* rgb_in, float* rgb_out, int32_t width, int32_t height, const float* albedo,
               const float* normal) {
  if (!ctx) return RT_ERR_INVALID;
  if (!rgb_in || !rgb_out || width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
    return fail(ctx, RT_ERR_INVALID, "bad denoise arguments");
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = (size_t)width * height;
  void *d_rgb, *d_a, *d_b, *d_alb = nullptr, *d_nrm = nullptr;
  int rc;
  if ((rc = grow_dev(ctx, SC_DN_RGB, n * 12, &d_rgb)) || (rc = grow_dev(ctx, SC_DN_A, n * 16, &d_a)) || (rc = grow_dev(ctx, SC_DN_B, n * 16, &d_b)))
    return rc;
  if (albedo && (rc = grow_dev(ctx, SC_DN_ALB, n * 16, &d_alb))) return rc;
  if (normal && (rc = grow_dev(ctx, SC_DN_NRM, n * 16, &d_nrm))) return rc;
  // host rgb -> device rgba, one image at a time through the staging buffer (stream order keeps the copies apart)
  auto stage = [&](const float* host, void* d_img) -> int {
    HIP_TRY(hipMemcpyAsync(d_rgb, host, n * 12, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH(k_dn_pack, n, (const float*)d_rgb, (float4*)d_img, (uint32_t)n);
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // (pageable host memory: the staging buffer is free again)
    return RT_OK;
  };
  if (albedo && (rc = stage(albedo, d_alb))) return rc;
  if (normal && (rc = stage(normal, d_nrm))) return rc;
  if ((rc = stage(rgb_in, d_a))) return rc;
  float4* res = denoise_passes(ctx, width, height, (float4*)d_a, (float4*)d_b, (const float4*)d_alb, (const float4*)d_nrm);
  LAUNCH(k_dn_unpack, n, (const float4*)res, (float*)d_rgb, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(rgb_out, d_rgb, n * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->profile_on) {
    std::vector<uint32_t> none;
    collect_profile(ctx, none);
  }
  return RT_OK;
}

// rt_denoise for callers that cannot allocate: device images in, device image out, the working images in this GPU's copy
// of the file-scope `scratch` array.  No hipMalloc / hipFree / allocating call on this path.
int rt_denoise_noalloc(rt_context* ctx, const float* d_rgb_in, float* d_rgb_out, int32_t width, int32_t height, const float* d_albedo,
                       const float* d_normal) {
  if (!ctx) return RT_ERR_INVALID;
  if (!d_rgb_in || !d_rgb_out || width <= 0 || height <= 0) return fail(ctx, RT_ERR_INVALID, "bad denoise arguments");
  if ((long long)width * height > (long long)RT_NOALLOC_MAX_PIXELS)
    return fail(ctx, RT_ERR_UNSUPPORTED, "rt_denoise_noalloc: width * height exceeds the fixed working array (1048576 pixels); use rt_denoise");
  if (!ctx->noalloc_ptr) return fail(ctx, RT_ERR_DEVICE, "rt_denoise_noalloc: the working array has no device address");
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = (size_t)width * height;
  // four RGBA images of RT_NOALLOC_MAX_PIXELS each; a call writes every element it later reads (pack, then each pass
  // writes all n pixels of its output), so nothing left by an earlier call is ever read
  float4* img = (float4*)ctx->noalloc_ptr;
  float4 *a = img, *b = img + RT_NOALLOC_MAX_PIXELS, *alb = nullptr, *nrm = nullptr;
  if (d_albedo) { alb = img + 2 * (size_t)RT_NOALLOC_MAX_PIXELS; LAUNCH(k_dn_pack, n, d_albedo, alb, (uint32_t)n); }
  if (d_normal) { nrm = img + 3 * (size_t)RT_NOALLOC_MAX_PIXELS; LAUNCH(k_dn_pack, n, d_normal, nrm, (uint32_t)n); }
  LAUNCH(k_dn_pack, n, d_rgb_in, a, (uint32_t)n);
  float4* res = denoise_passes(ctx, width, height, a, b, alb, nrm, false);
  LAUNCH(k_dn_unpack, n, (const float4*)res, d_rgb_out, (uint32_t)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_counters_enable(rt_context* ctx, int enable) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  ctx->counters_on = enable != 0;
  return RT_OK;
}
int rt_counters_reset(rt_context* ctx) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemset(ctx->d_counters, 0, sizeof(DCounters)));
  ctx->host_paths = 0;
  return RT_OK;
}
int rt_counters_get(rt_context* ctx, rt_counters* out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  static_assert(sizeof(DCounters) == sizeof(rt_counters), "counter layout");
  quiesce(ctx);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(out, ctx->d_counters, sizeof(rt_counters), hipMemcpyDeviceToHost));
  out->paths += ctx->host_paths;
#ifdef RT_WALK_STATS
  {
    unsigned long long ws[16];
    if (hipMemcpyFromSymbol(ws, HIP_SYMBOL(g_walk_stats), sizeof ws) == hipSuccess) {
      std::fprintf(stderr, "walk stats: outer %llu (walk lanes %.1f) | service %llu (done %.1f idle %.1f next %.1f) | node iters %llu (lanes %.1f) | leaf phases %llu (lanes %.1f) | tri iters %llu (lanes %.1f)\n",
                   ws[0], (double)ws[1] / (double)(ws[0] ? ws[0] : 1), ws[2], (double)ws[3] / (double)(ws[2] ? ws[2] : 1), (double)ws[4] / (double)(ws[2] ? ws[2] : 1),
                   (double)ws[5] / (double)(ws[2] ? ws[2] : 1), ws[6], (double)ws[7] / (double)(ws[6] ? ws[6] : 1), ws[8], (double)ws[9] / (double)(ws[8] ? ws[8] : 1), ws[10],
                   (double)ws[11] / (double)(ws[10] ? ws[10] : 1));
      double tt = (double)(ws[12] + ws[13] + ws[14] + ws[15]);
      std::fprintf(stderr, "walk cycles: classify %.1f%% service %.1f%% node %.1f%% leaf %.1f%% (total %.3g wave-cycles)\n", 100.0 * ws[12] / tt, 100.0 * ws[13] / tt,
                   100.0 * ws[14] / tt, 100.0 * ws[15] / tt, tt);
    }
  }
#endif
  return RT_OK;
}
int rt_profile_enable(rt_context* ctx, int enable) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  ctx->profile_on = enable != 0;
  return RT_OK;
}
int rt_profile_reset(rt_context* ctx) {
  if (!ctx) return RT_ERR_INVALID;
  quiesce(ctx);
  std::memset(&ctx->prof, 0, sizeof ctx->prof);
  return RT_OK;
}
int rt_profile_get(rt_context* ctx, rt_profile* out) {
  if (!ctx || !out) return RT_ERR_INVALID;
  quiesce(ctx);
  *out = ctx->prof;
  return RT_OK;
}

}  // extern "C"
