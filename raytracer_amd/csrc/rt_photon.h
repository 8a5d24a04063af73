// Photon pass and photon-map irradiance gather (reference RayTracer.cpp:262-334, photonmap.h, materials.cpp:315-390).
#pragma once
#include "rt_kernels.h"

struct PhotonState {
  bool built = false;
  uint32_t key[8] = {0};   // parameters the maps were built with
  void* mem[16] = {nullptr};
  int n_mem = 0;
  int32_t count[2] = {0, 0};
};

static inline void photon_state_free(PhotonState& s) {
  for (int i = 0; i < s.n_mem; i++)
    if (s.mem[i]) (void)hipFree(s.mem[i]);
  s.n_mem = 0;
  s.built = false;
  s.count[0] = s.count[1] = 0;
}
