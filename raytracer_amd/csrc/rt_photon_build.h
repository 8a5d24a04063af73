// Photon maps built on the device (reference RayTracer.cpp:262-294, photonmap.h:158-264).
//
// What the reference computes and what has to come out here:
//   * AddPhoton keeps the first n deposits per map in the order a sequential emitter makes them (path by path, bounce by
//     bounce, photonmap.h:158-172) and ScalePhotonPowers multiplies their powers by 1 / numPhotons (RayTracer.cpp:290-292);
//   * PrepareForIrradianceEstimation (photonmap.h:176-198) arranges them as a LEFT-BALANCED kd-tree in heap order: the
//     tree over m photons is complete, its last level filled from the left, so the root of a segment of m photons sorted
//     along the split axis is the photon with (2^(h-1) - 1) + min(m - 2^h + 1, 2^(h-1)) photons before it (h = floor(log2 m));
//     the split axis is the widest side of the segment's box, which starts as the box of all photons AND THE ORIGIN
//     (photons[0] is a zeroed slot) and is clipped at the split coordinate on the way down; a child segment of one photon
//     is copied as it is, every other node gets its axis written into the plane byte.
//
// The reference gets there with a serial quickselect per segment; nothing of that survives here.  The device build:
//   emission   k_photon_trace notes, per deposit, its path and its ordinal among the path's deposits into that map, and per
//              path how many it made; an exclusive scan over the paths turns (path, ordinal) into the deposit's slot in
//              insertion order (k_photon_place) — no sort, no host pass;
//   kd-tree    the photons are ordered ONCE along each axis (three stable radix sorts of (coordinate, insertion index) —
//              rt_sort.hip); then, level by level and for all segments of both maps at once: the thread standing at a
//              segment's median position (in the list of the segment's split axis) writes the tree node and the two child
//              boxes (k_kd_split), every photon learns which side it is on, and a scan + scatter keeps all three lists
//              partitioned per segment AND sorted (k_kd_scatter) — five small launches per level, log2(n) levels.
// Tie rule: photons with equal coordinates along a split axis keep their insertion order (the sorts are stable), so the
// tree is a pure function of the insertion-order array.  For distinct coordinates it is exactly the reference's tree
// (median and subsets are then unique); with ties the reference's result depends on its quickselect's swap history, and
// the two trees may differ in how equal coordinates are dealt left and right — both are valid kd-trees of the same
// photons, so every lookup finds the same photons (the visiting order, which only the reference heap's first-eviction
// quirk can see, may differ).
#pragma once
#include "rt_photon.h"

// ------------------------------------------------------------------------------------------------ scans
// Exclusive prefix sums of uint2 values produced by a loader functor, over tiles of RT_SCAN_TILE positions per block:
// k_scan_sums (tile totals) -> k_scan_tiles (exclusive scan of the totals, one block) -> k_scan_write (prefix per position).
#define RT_SCAN_ITEMS 8
#define RT_SCAN_TILE (RT_BLOCK * RT_SCAN_ITEMS)

RT_HD uint2 u2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
RT_HD uint2 u2_add(uint2 a, uint2 b) { return u2(a.x + b.x, a.y + b.y); }

#ifndef RT_EMU
// block-wide exclusive scan of one uint2 per thread (256 threads); returns the block total through `total`
RT_DEV uint2 block_exclusive_scan(uint2 v, uint2& total) {
  __shared__ uint32_t sx[RT_BLOCK], sy[RT_BLOCK];
  const uint32_t t = threadIdx.x;
  sx[t] = v.x; sy[t] = v.y;
  __syncthreads();
#pragma unroll
  for (uint32_t d = 1; d < RT_BLOCK; d <<= 1) {
    uint32_t ax = 0, ay = 0;
    if (t >= d) { ax = sx[t - d]; ay = sy[t - d]; }
    __syncthreads();
    sx[t] += ax; sy[t] += ay;
    __syncthreads();
  }
  total = u2(sx[RT_BLOCK - 1], sy[RT_BLOCK - 1]);
  uint2 incl = u2(sx[t], sy[t]);
  __syncthreads();
  return u2(incl.x - v.x, incl.y - v.y);
}

template <class Load>
__global__ void __launch_bounds__(256) k_scan_sums(Load ld, uint32_t n, uint2* tiles) {
  const uint32_t base = blockIdx.x * RT_SCAN_TILE + threadIdx.x * RT_SCAN_ITEMS;
  uint2 s = u2(0, 0);
#pragma unroll
  for (int i = 0; i < RT_SCAN_ITEMS; i++)
    if (base + i < n) s = u2_add(s, ld(base + i));
  uint2 total;
  (void)block_exclusive_scan(s, total);
  if (threadIdx.x == 0) tiles[blockIdx.x] = total;
}
// one block: tiles[] -> exclusive prefix in place, grand total to tiles[n_tiles]
__global__ void __launch_bounds__(256) k_scan_tiles(uint2* tiles, uint32_t n_tiles) {
  uint2 carry = u2(0, 0);
  for (uint32_t c = 0; c < n_tiles; c += RT_BLOCK) {
    const uint32_t i = c + threadIdx.x;
    uint2 v = i < n_tiles ? tiles[i] : u2(0, 0);
    uint2 total;
    uint2 ex = block_exclusive_scan(v, total);
    if (i < n_tiles) tiles[i] = u2_add(carry, ex);
    carry = u2_add(carry, total);
  }
  if (threadIdx.x == 0) tiles[n_tiles] = carry;
}
template <class Load>
__global__ void __launch_bounds__(256) k_scan_write(Load ld, uint32_t n, const uint2* tiles, uint2* out) {
  const uint32_t base = blockIdx.x * RT_SCAN_TILE + threadIdx.x * RT_SCAN_ITEMS;
  uint2 v[RT_SCAN_ITEMS];
  uint2 s = u2(0, 0);
#pragma unroll
  for (int i = 0; i < RT_SCAN_ITEMS; i++) {
    v[i] = base + i < n ? ld(base + i) : u2(0, 0);
    s = u2_add(s, v[i]);
  }
  uint2 total;
  uint2 run = u2_add(block_exclusive_scan(s, total), tiles[blockIdx.x]);
#pragma unroll
  for (int i = 0; i < RT_SCAN_ITEMS; i++) {
    if (base + i < n) out[base + i] = run;
    run = u2_add(run, v[i]);
  }
}
#else
template <class Load>
static void k_scan_sums(Load, uint32_t, uint2*) {}
static void k_scan_tiles(uint2*, uint32_t) {}
template <class Load>
static void k_scan_write(Load ld, uint32_t n, const uint2* tiles, uint2* out) {   // (emulation: one serial pass)
  uint2 run = u2(0, 0);
  for (uint32_t q = 0; q < n; q++) { out[q] = run; run = u2_add(run, ld(q)); }
  const_cast<uint2*>(tiles)[(n + RT_SCAN_TILE - 1) / RT_SCAN_TILE] = run;
}
#endif

// ------------------------------------------------------------------------------------------------ emission -> insertion order
// Deposits of one batch of photon paths, appended unordered by k_photon_trace: record r of map w carries key
// (path - first, ordinal among that path's deposits into w).  `per_path[path - first]` = deposits per map of that path.
struct PathCountLoad {
  const uint2* per_path;
  RT_MEM uint2 operator()(uint32_t q) const { return per_path[q]; }
};

struct PhotonPlaceP {
  const float4* rec[2];     // 2 float4 per record
  const uint2* key[2];
  const uint32_t* n_rec;    // [2] records appended by the batch
  const uint2* path_base;   // exclusive scan of per_path: deposits of earlier paths of the batch, per map
  float4* raw[2];           // insertion-order arrays of the maps (n photons each)
  uint32_t kept[2];         // photons the maps hold from earlier batches
  uint32_t n;               // capacity of each map (numPhotons)
  uint32_t cap;             // record capacity per map
  float scale;              // 1 / numPhotons (RayTracer.cpp:290-292)
};
RT_KERNEL(k_photon_place, PhotonPlaceP A) {
  for (int w = 0; w < 2; w++) {
    uint32_t m = A.n_rec[w];
    if (m > A.cap) m = A.cap;
    for (uint32_t r = RT_TID; r < m; r += RT_NTHREADS) {
      const uint2 k = A.key[w][r];
      const uint2 b = A.path_base[k.x];
      const unsigned long long at = (unsigned long long)A.kept[w] + (w ? b.y : b.x) + k.y;
      if (at >= A.n) continue;   // AddPhoton on a full map (photonmap.h:160)
      float4 a = A.rec[w][2 * (size_t)r], d = A.rec[w][2 * (size_t)r + 1];
      a.w = a.w * A.scale;       // ScalePower
      A.raw[w][2 * (size_t)at] = a;
      A.raw[w][2 * (size_t)at + 1] = d;
    }
  }
}

// ------------------------------------------------------------------------------------------------ kd-tree
#define RT_KD_DEAD 0xFFFFFFFFu
struct KdSeg {           // a segment still to be split: positions [start, end] of the three lists, and its box
  float bmin[3];
  uint32_t start;
  float bmax[3];
  uint32_t end;
};
struct KdMaps {          // the maps one build handles (one or two), laid side by side in one position / element space
  const float4* raw[2];  // insertion-order photons
  float4* tree[2];       // out: heap order, 1-based; entries 0 and n + 1 zeroed by the host
  uint32_t off[2];       // first element (= first position) of the map
  uint32_t n[2];
  uint32_t n_maps, n_total;
  uint32_t cap;          // segment slots per map and level: slot = map * cap + (heap index - first heap index of the level)
};
// (the two maps' fields are picked with selects: a per-lane index into an array held in kernel-argument registers is what
// this compiler's instruction selection cannot digest)
RT_DEV uint32_t kd_map_of(const KdMaps& M, uint32_t e) { return (M.n_maps > 1 && e >= M.off[1]) ? 1u : 0u; }
RT_DEV uint32_t kd_off(const KdMaps& M, uint32_t m) { return m ? M.off[1] : M.off[0]; }
RT_DEV uint32_t kd_count(const KdMaps& M, uint32_t m) { return m ? M.n[1] : M.n[0]; }
RT_DEV float4* kd_tree(const KdMaps& M, uint32_t m) { return m ? M.tree[1] : M.tree[0]; }
RT_DEV const float4* kd_photon(const KdMaps& M, uint32_t e) {
  const uint32_t m = kd_map_of(M, e);
  return (m ? M.raw[1] : M.raw[0]) + 2 * (size_t)(e - kd_off(M, m));
}

// order-preserving float -> uint key (-0 and +0 tie, as they do under the reference's < and >)
#define RT_KD_KEY_ZERO 0x80000000u   // kd_key(0.0f)
RT_DEV uint32_t kd_key(float x) {
  if (x == 0.0f) x = 0.0f;
  uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
RT_DEV float kd_unkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// box of each map's photons and the origin, as order-preserving keys: bounds[map][0..2] min, [3..5] max (the host presets
// every word to the key of 0)
RT_KERNEL(k_kd_bounds, KdMaps M, uint32_t* bounds) {
  float lo[2][3], hi[2][3];
  for (int m = 0; m < 2; m++)
    for (int a = 0; a < 3; a++) { lo[m][a] = 0.0f; hi[m][a] = 0.0f; }
  // (the loop is wave-uniform: every lane reaches the wave-wide reductions below)
  for (uint32_t e0 = RT_TID - rt_lane(); e0 < M.n_total; e0 += RT_NTHREADS) {
    const uint32_t e = e0 + rt_lane();
    if (e < M.n_total) {
      const uint32_t m = kd_map_of(M, e);
      const float4 p = kd_photon(M, e)[0];
      lo[m][0] = fminf(lo[m][0], p.x); hi[m][0] = fmaxf(hi[m][0], p.x);
      lo[m][1] = fminf(lo[m][1], p.y); hi[m][1] = fmaxf(hi[m][1], p.y);
      lo[m][2] = fminf(lo[m][2], p.z); hi[m][2] = fmaxf(hi[m][2], p.z);
    }
  }
  for (int m = 0; m < 2; m++)
    for (int a = 0; a < 3; a++) {
      const float l = wave_min(lo[m][a]), h = wave_max(hi[m][a]);
      if (rt_lane() == 0) {
        if (l < 0.0f) rt_atomic_min_u32(bounds + 6 * m + a, kd_key(l));
        if (h > 0.0f) rt_atomic_max_u32(bounds + 6 * m + 3 + a, kd_key(h));
      }
    }
}

// sort keys of one axis: (map, coordinate) — the maps stay apart, ties keep the insertion order (stable sort on these keys)
RT_KERNEL(k_kd_keys, KdMaps M, int axis, unsigned long long* keys, uint32_t* vals) {
  for (uint32_t e = RT_TID; e < M.n_total; e += RT_NTHREADS) {
    const float4 p = kd_photon(M, e)[0];
    const float c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
    keys[e] = ((unsigned long long)kd_map_of(M, e) << 32) | kd_key(c);
    vals[e] = e;
  }
}

// root segments and the position -> segment-slot table of level 0
RT_KERNEL(k_kd_roots, KdMaps M, const uint32_t* bounds, KdSeg* segs, uint32_t* pos_seg) {
  for (uint32_t p = RT_TID; p < M.n_total; p += RT_NTHREADS) pos_seg[p] = kd_map_of(M, p) * M.cap;
  for (uint32_t m = RT_TID; m < M.n_maps; m += RT_NTHREADS) {
    if (kd_count(M, m) == 0) continue;
    uint32_t* out = (uint32_t*)(segs + (size_t)m * M.cap);   // KdSeg: bmin[3], start, bmax[3], end
    for (int a = 0; a < 3; a++) {
      out[a] = __float_as_uint(kd_unkey(bounds[6 * m + a]));
      out[4 + a] = __float_as_uint(kd_unkey(bounds[6 * m + 3 + a]));
    }
    out[3] = kd_off(M, m);
    out[7] = kd_off(M, m) + kd_count(M, m) - 1;
  }
}

// Split axis (widest side of the box; on equal sides the later axis, with z last: the reference's nested compares give
// x only if it beats both, y only if x does not beat it and it beats z) and median position of a segment.
RT_DEV void kd_split_of(const KdSeg& s, int& axis, uint32_t& median) {
  const float dx = s.bmax[0] - s.bmin[0], dy = s.bmax[1] - s.bmin[1], dz = s.bmax[2] - s.bmin[2];
  axis = (dx > dy && dx > dz) ? 0 : ((!(dx > dy) && dy > dz) ? 1 : 2);
  const uint32_t m = s.end - s.start + 1;
  uint32_t left = 0;
  if (m > 1) {   // complete tree over m nodes, last level filled from the left: size of the root's left subtree
    const uint32_t h = 31u - rt_clz(m);
    const uint32_t half = 1u << (h - 1);
    const uint32_t last = m - (1u << h) + 1u;
    left = (half - 1u) + (last < half ? last : half);
  }
  median = s.start + left;
}

struct KdLevel {
  const uint32_t* L[3];     // per axis: element ids, every live segment's range sorted along that axis
  uint32_t* Ln[3];          // the same for the next level
  const uint32_t* pos_seg;  // position -> segment slot of this level (RT_KD_DEAD: position already placed in the tree)
  uint32_t* pos_seg_n;
  const KdSeg* segs;
  KdSeg* segs_n;
  uint8_t* side;            // per element: 0 left, 1 right, 2 the segment's median
  uint2* scan;              // [3 * n_total] exclusive (left, right) counts, list after list
  uint32_t first_node;      // heap index of the level's first node: 2^level
};

// For every position of a live segment: which side the photon standing there (in the split axis' list) goes to.  The
// thread at the median position also writes the tree node and the child segments (or copies a one-photon child).
RT_KERNEL(k_kd_split, KdMaps M, KdLevel V) {
  for (uint32_t p = RT_TID; p < M.n_total; p += RT_NTHREADS) {
    const uint32_t slot = V.pos_seg[p];
    if (slot == RT_KD_DEAD) continue;
    const KdSeg s = V.segs[slot];
    int axis;
    uint32_t median;
    kd_split_of(s, axis, median);
    const uint32_t* La = axis == 0 ? V.L[0] : (axis == 1 ? V.L[1] : V.L[2]);
    const uint32_t e = La[p];
    V.side[e] = p < median ? 0 : (p == median ? 2 : 1);
    if (p != median) continue;
    const uint32_t m = slot / M.cap, idx = slot - m * M.cap;
    const size_t node = (size_t)V.first_node + idx;
    float4* tree = kd_tree(M, m);
    const float4* ph = kd_photon(M, e);
    float4 a = ph[0], b = ph[1];
    uint32_t bits = __float_as_uint(b.w);   // colour bytes | plane byte << 24: SetPlane keeps bit 3 (photonmap.h:53)
    bits = (bits & 0x08ffffffu) | ((uint32_t)axis << 24);
    tree[2 * node] = a;
    tree[2 * node + 1] = make_float4(b.x, b.y, b.z, __uint_as_float(bits));
    const float split = axis == 0 ? a.x : (axis == 1 ? a.y : a.z);
    KdSeg l = s, r = s;
    l.end = median - 1;                      // (median == start: start > end marks an empty child; never looked at)
    r.start = median + 1;
    l.bmax[axis] = split;
    r.bmin[axis] = split;
    const uint32_t nl = median - s.start, nr = s.end - median;
    if (nl == 1) {   // a child of one photon is copied as it is, plane byte untouched (photonmap.h:250-252)
      const float4* c = kd_photon(M, La[s.start]);
      tree[2 * (2 * node)] = c[0];
      tree[2 * (2 * node) + 1] = c[1];
    }
    if (nr == 1) {
      const float4* c = kd_photon(M, La[s.end]);
      tree[2 * (2 * node + 1)] = c[0];
      tree[2 * (2 * node + 1) + 1] = c[1];
    }
    if (nl > 1) V.segs_n[m * M.cap + 2 * idx] = l;
    if (nr > 1) V.segs_n[m * M.cap + 2 * idx + 1] = r;
  }
}

// scan input: position q of the concatenated lists (list = q / n_total) counts (1, 0) for a left, (0, 1) for a right photon
struct KdSideLoad {
  const uint32_t* L[3];
  const uint32_t* pos_seg;
  const uint8_t* side;
  uint32_t n_total;
  RT_MEM uint2 operator()(uint32_t q) const {
    const uint32_t b = q / n_total, p = q - b * n_total;
    if (pos_seg[p] == RT_KD_DEAD) return u2(0, 0);
    const uint32_t e = (b == 0 ? L[0] : (b == 1 ? L[1] : L[2]))[p];
    const uint32_t sd = side[e];
    return u2(sd == 0 ? 1u : 0u, sd == 1 ? 1u : 0u);
  }
};

// Stable partition of every live segment of every list: lefts to the front, the median photon at the median position,
// rights behind it — each group in its old (sorted) order.  List 0's threads also write the next level's slot table.
RT_KERNEL(k_kd_scatter, KdMaps M, KdLevel V) {
  const uint32_t total = 3u * M.n_total;
  for (uint32_t q = RT_TID; q < total; q += RT_NTHREADS) {
    const uint32_t b = q / M.n_total, p = q - b * M.n_total;
    const uint32_t* Lb = b == 0 ? V.L[0] : (b == 1 ? V.L[1] : V.L[2]);
    uint32_t* Lo = b == 0 ? V.Ln[0] : (b == 1 ? V.Ln[1] : V.Ln[2]);
    const uint32_t slot = V.pos_seg[p];
    const uint32_t e = Lb[p];
    if (slot == RT_KD_DEAD) {
      Lo[p] = e;
      if (b == 0) V.pos_seg_n[p] = RT_KD_DEAD;
      continue;
    }
    const KdSeg s = V.segs[slot];
    int axis;
    uint32_t median;
    kd_split_of(s, axis, median);
    const uint32_t m = slot / M.cap, idx = slot - m * M.cap;
    const uint2 g = V.scan[q], g0 = V.scan[b * M.n_total + s.start];
    const uint32_t sd = V.side[e];
    uint32_t np, child = RT_KD_DEAD;
    if (sd == 0) {
      np = s.start + (g.x - g0.x);
      if (median - s.start > 1) child = m * M.cap + 2 * idx;
    } else if (sd == 1) {
      np = median + 1 + (g.y - g0.y);
      if (s.end - median > 1) child = m * M.cap + 2 * idx + 1;
    } else {
      np = median;
    }
    Lo[np] = e;
    if (b == 0) V.pos_seg_n[np] = child;
  }
}
