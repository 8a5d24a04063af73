// Mesh BVHs built on the device (the reference builds its BVHTriMesh on one CPU thread inside TriObj::Load,
// objects.h:60-67; its builder, cyBVH, is not part of the checkout, so the tree itself is this build's choice — any
// valid tree with leaves of at most 4 triangles gives the reference's hits, the tie keys see to that).
//
// A mesh whose rt_mesh carries no host-built tree (bvh == NULL) gets a linear BVH here at rt_scene_commit:
//   keys      one 63-bit Morton code per face from the centre of its box, 21 bits per axis inside the mesh bounds (k_lbvh_keys);
//   order     one radix sort of (code, face) pairs (rt_sort.hip); the triangle slots are written in that order
//             (k_lbvh_slots), so every subtree is a contiguous run of slots and a leaf is (first slot, count);
//   topology  top-down, one launch per level, one thread per node: a run of more than 4 slots is split where the highest
//             differing bit of its first and last code flips (binary search), or in the middle when all its codes are equal or
//             when a child would not fit under the levels that remain (the walk's stack holds RT_STACK levels).  A node's
//             index is its place in the work list, levels are contiguous in it (k_lbvh_split / k_lbvh_next);
//   boxes     bottom-up, one launch per level from the deepest: a node's record holds its two children's boxes, taken
//             from the slots' boxes (leaf child) or from the child's own record, already final (k_lbvh_refit).
// Output is the device layout the traversal kernels read (rt_device.h): 64-byte nodes, 48-byte slots with tie key
// first slot of the leaf x 8 + (7 - position in the leaf) — monotone in the depth-first leaf order, as the converter's.
#pragma once

#include "rt_device.h"

struct LbvhArgs {
  const float* pos;          // 3 per vertex
  const uint32_t* fidx;      // 3 per face
  uint32_t n;                // faces
  float lo[3], scale[3];     // quantisation of a box centre: (c - lo) * scale in [0, 2^21)
  unsigned long long* keys;  // Morton codes (unsorted in k_lbvh_keys, sorted afterwards)
  uint32_t* vals;            // face ids, same order
  float4* tris;              // this mesh's slots in DScene::tris
  float4* nodes;             // this mesh's nodes in DScene::bvh
  float4* pbox;              // 2 per slot: box min, box max
  uint2* work;               // (first slot, end slot) of node i
  uint2* lv;                 // per level: (first node, node count)
  uint32_t* total;           // nodes allocated so far
  int max_levels;            // levels that may hold inner nodes
};

RT_DEV unsigned long long lbvh_spread21(uint32_t v) {  // bit i of v -> bit 3 i
  unsigned long long x = v & 0x1FFFFFull;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
RT_DEV uint32_t lbvh_quant(float c, float lo, float scale) {
  float q = (c - lo) * scale;
  if (!(q > 0.0f)) q = 0.0f;
  if (q > 2097151.0f) q = 2097151.0f;
  return (uint32_t)q;
}
RT_DEV uint32_t lbvh_clz64(unsigned long long v) {
  uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
  return hi ? rt_clz(hi) : 32u + rt_clz(lo);
}

RT_KERNEL(k_lbvh_keys, LbvhArgs A) {
  for (uint32_t f = RT_TID; f < A.n; f += RT_NTHREADS) {
    const uint32_t* ix = A.fidx + 3 * (size_t)f;
    const float *a = A.pos + 3 * (size_t)ix[0], *b = A.pos + 3 * (size_t)ix[1], *c = A.pos + 3 * (size_t)ix[2];
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
      float mn = fminf(a[k], fminf(b[k], c[k])), mx = fmaxf(a[k], fmaxf(b[k], c[k]));
      q[k] = lbvh_quant(0.5f * (mn + mx), A.lo[k], A.scale[k]);
    }
    A.keys[f] = (lbvh_spread21(q[0]) << 2) | (lbvh_spread21(q[1]) << 1) | lbvh_spread21(q[2]);
    A.vals[f] = f;
  }
}

// slot s <- face vals[s]: v0, edge1, edge2 (objects.cpp:203-204) + face id; the tie key is written with the leaf
RT_KERNEL(k_lbvh_slots, LbvhArgs A) {
  for (uint32_t s = RT_TID; s < A.n; s += RT_NTHREADS) {
    const uint32_t f = A.vals[s];
    const uint32_t* ix = A.fidx + 3 * (size_t)f;
    const float *a = A.pos + 3 * (size_t)ix[0], *b = A.pos + 3 * (size_t)ix[1], *c = A.pos + 3 * (size_t)ix[2];
    A.tris[3 * (size_t)s + 0] = make_float4(a[0], a[1], a[2], 0.0f);
    A.tris[3 * (size_t)s + 1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], __uint_as_float(f));
    A.tris[3 * (size_t)s + 2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
    A.pbox[2 * (size_t)s + 0] = make_float4(fminf(a[0], fminf(b[0], c[0])), fminf(a[1], fminf(b[1], c[1])), fminf(a[2], fminf(b[2], c[2])), 0.0f);
    A.pbox[2 * (size_t)s + 1] = make_float4(fmaxf(a[0], fmaxf(b[0], c[0])), fmaxf(a[1], fmaxf(b[1], c[1])), fmaxf(a[2], fmaxf(b[2], c[2])), 0.0f);
  }
}

// child reference of the run [lo, hi): a leaf (its slots get their tie keys) or a new node of the next level
RT_DEV uint32_t lbvh_child(const LbvhArgs& A, uint32_t lo, uint32_t hi) {
  const uint32_t c = hi - lo;
  if (c <= 4u) {
    for (uint32_t s = 0; s < c; s++) ((float*)&A.tris[3 * (size_t)(lo + s)])[3] = __uint_as_float(lo * 8u + (7u - s));
    return RT_LEAF | (lo << 3) | c;
  }
  const uint32_t at = rt_atomic_add_u32(A.total, 1u);
  uint2 run;
  run.x = lo; run.y = hi;
  A.work[at] = run;
  return at;
}

RT_KERNEL(k_lbvh_split, LbvhArgs A, int level) {
  const uint32_t first = A.lv[level].x, count = A.lv[level].y;
  const int below = A.max_levels - level - 1;                             // levels left for inner nodes under a child
  const unsigned long long cap = below >= 40 ? ~0ull : (4ull << below);  // slots such a child can hold
  for (uint32_t i = RT_TID; i < count; i += RT_NTHREADS) {
    const uint32_t me = first + i;
    const uint2 r = A.work[me];
    const uint32_t lo = r.x, hi = r.y, n = hi - lo;
    const unsigned long long ka = A.keys[lo], kb = A.keys[hi - 1];
    uint32_t mid = lo + (n + 1) / 2;
    if (ka != kb) {
      const uint32_t common = lbvh_clz64(ka ^ kb);
      uint32_t split = lo, step = n - 1;   // the last slot of the left run: the last code that shares more than `common` bits with ka
      do {
        step = (step + 1) >> 1;
        const uint32_t t = split + step;
        if (t < hi - 1 && lbvh_clz64(ka ^ A.keys[t]) > common) split = t;
      } while (step > 1);
      const uint32_t m = split + 1;
      if ((unsigned long long)(m - lo) <= cap && (unsigned long long)(hi - m) <= cap) mid = m;
    }
    const uint32_t lref = lbvh_child(A, lo, mid), rref = lbvh_child(A, mid, hi);
    A.nodes[4 * (size_t)me + 3] = make_float4(__uint_as_float(lref), __uint_as_float(rref), __uint_as_float(mid), 0.0f);   // .z: first slot of the right child
  }
}

// the next level = the nodes allocated while this one was split
RT_KERNEL(k_lbvh_next, LbvhArgs A, int level) {
  if (RT_TID == 0) {
    uint2 nx;
    nx.x = A.lv[level].x + A.lv[level].y;
    nx.y = *A.total - nx.x;
    A.lv[level + 1] = nx;
  }
}

RT_DEV void lbvh_child_box(const LbvhArgs& A, uint32_t ref, float* mn, float* mx) {
  if (ref & RT_LEAF) {
    const uint32_t lo = (ref & ~RT_LEAF) >> 3, c = ref & 7u;
    for (int k = 0; k < 3; k++) { mn[k] = RT_BIG; mx[k] = -RT_BIG; }
    for (uint32_t s = 0; s < c; s++) {
      const float4 a = A.pbox[2 * (size_t)(lo + s)], b = A.pbox[2 * (size_t)(lo + s) + 1];
      mn[0] = fminf(mn[0], a.x); mn[1] = fminf(mn[1], a.y); mn[2] = fminf(mn[2], a.z);
      mx[0] = fmaxf(mx[0], b.x); mx[1] = fmaxf(mx[1], b.y); mx[2] = fmaxf(mx[2], b.z);
    }
  } else {   // (left min, right min, left max, right max) per axis
    const float4 x = A.nodes[4 * (size_t)ref], y = A.nodes[4 * (size_t)ref + 1], z = A.nodes[4 * (size_t)ref + 2];
    mn[0] = fminf(x.x, x.y); mx[0] = fmaxf(x.z, x.w);
    mn[1] = fminf(y.x, y.y); mx[1] = fmaxf(y.z, y.w);
    mn[2] = fminf(z.x, z.y); mx[2] = fmaxf(z.z, z.w);
  }
}

RT_KERNEL(k_lbvh_refit, LbvhArgs A, int level) {
  const uint32_t first = A.lv[level].x, count = A.lv[level].y;
  for (uint32_t i = RT_TID; i < count; i += RT_NTHREADS) {
    const uint32_t me = first + i;
    const float4 refs = A.nodes[4 * (size_t)me + 3];
    float lmn[3], lmx[3], rmn[3], rmx[3];
    lbvh_child_box(A, __float_as_uint(refs.x), lmn, lmx);
    lbvh_child_box(A, __float_as_uint(refs.y), rmn, rmx);
    A.nodes[4 * (size_t)me + 0] = make_float4(lmn[0], rmn[0], lmx[0], rmx[0]);
    A.nodes[4 * (size_t)me + 1] = make_float4(lmn[1], rmn[1], lmx[1], rmx[1]);
    A.nodes[4 * (size_t)me + 2] = make_float4(lmn[2], rmn[2], lmx[2], rmx[2]);
  }
}

// root box (6 floats), node count and the level table for the host; a mesh of at most 4 faces is one leaf
RT_KERNEL(k_lbvh_root, LbvhArgs A, uint32_t root_ref, float* out) {
  if (RT_TID == 0) {
    if (root_ref & RT_LEAF) {
      const uint32_t c = root_ref & 7u;
      for (uint32_t s = 0; s < c; s++) ((float*)&A.tris[3 * (size_t)s])[3] = __uint_as_float(7u - s);
    }
    float mn[3], mx[3];
    lbvh_child_box(A, root_ref, mn, mx);
    for (int k = 0; k < 3; k++) { out[k] = mn[k]; out[3 + k] = mx[k]; }
  }
}
